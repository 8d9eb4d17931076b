import ctypes, json, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
def ms(ptr, v, n):
    assert hip.hipMemsetAsync(ctypes.c_void_p(ptr), v, n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
def hx(t): return bytes(t.cpu().tolist()).hex()
out = []
# A: one node, tensor at different places in the segment, value 0 and 0x5a, no consumer kernel / with consumer
for shift in [0, 1, 5]:
    pads = [torch.empty(300, dtype=torch.uint8, device="cuda") for _ in range(shift)]
    for o in [0, 4, 1, 16, 256]:
        for v in [0, 0x5a]:
            for s in [4, 12, 64]:
                b = torch.full((s + o + 600,), 0xEE, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ms(b.data_ptr() + o, v, s)
                torch.cuda.synchronize()
                rec = {"shift": shift, "o": o, "v": v, "s": s, "ptr_lo": hex((b.data_ptr()) & 0xFFFFFF)}
                for r in range(3):
                    b.fill_(0xEE)
                    g.replay(); torch.cuda.synchronize()
                    good = bool((b[o:o+s] == v).all()) and bool((b[:o] == 0xEE).all()) and bool((b[o+s:] == 0xEE).all())
                    rec[f"r{r}"] = "ok" if good else hx(b[max(o-4,0):o+s+4][:24])
                if any(rec[f"r{r}"] != "ok" for r in range(3)):
                    out.append(rec)
                del b, g
    del pads
print(json.dumps({"single_bad": out[:40], "n_bad": len(out)}))
# B: memset of a tensor allocated INSIDE the capture (graph private pool), followed by a consumer; like ATen's semaphores
res = []
for keep in [0, 1, 3]:
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(64, dtype=torch.int32, device="cuda")
    with torch.cuda.graph(g):
        alive = [torch.empty(100, device="cuda") for _ in range(keep)]
        sem = torch.empty(3, dtype=torch.int32, device="cuda")
        ms(sem.data_ptr(), 0, 12)
        sem.add_(1)
        x[:3].copy_(sem)
    vals = []
    for r in range(4):
        g.replay(); torch.cuda.synchronize(); vals.append(x[:3].tolist())
    res.append({"keep": keep, "sem_lo": hex(sem.data_ptr() & 0xFFFFFF), "vals": vals})
print(json.dumps({"in_pool": res}))
