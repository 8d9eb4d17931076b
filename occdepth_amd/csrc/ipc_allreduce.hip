// Small-message all-reduce over peer-mapped device memory (the training step's SyncBatchNorm
// exchanges of occdepth/scripts/train.py:176-206 -- `Trainer(sync_batchnorm=True)`).
//
// The config-2 training step makes 532 all-reduces of (2C + 1) statistics, C <= 3840: a few hundred bytes to 60 KB each, all
// on ONE dependency chain (layer i's backward sums need layer i + 1's), so they cannot be batched, and each costs RCCL's
// launch + completion latency (~45 us inside a captured step: +16 % on one rank, 0.72 - 0.78 weak-scaling efficiency
// budgeted for 8 GPUs).  This is the latency-optimal exchange for that regime, one kernel per call, no library in the path:
//
//   every rank owns a MAILBOX in its own HBM -- 2 slots x world rows -- allocated fine-grained (coherent across XCDs and
//   over xGMI while a kernel runs) and exported with hipIpcGetMemHandle; every rank maps all peers' mailboxes
//   (hipIpcOpenMemHandle) once, at set-up.  The wire format is the "LL" one of the collective libraries: every 32-bit half of
//   the payload travels in its own 8-byte word {data : 32, sequence : 32}, written with ONE system-scope atomic store and
//   polled with system-scope atomic loads -- a word is either entirely there or not, so NO FENCE is needed anywhere
//   (a system-scope release / acquire pair costs an L2 write-back + invalidate, several microseconds each; round 5 measured
//   8.1 us per exchange with fences, and a one-launch SyncBatchNorm layer with two fences per workgroup was 35 % SLOWER than
//   the five-launch path).  all_reduce(seq):
//     1. push : write my vector, word by word, into slot seq & 1, row `rank`, of EVERY mailbox (N - 1 remote writes over the
//               N - 1 direct xGMI links of a fully connected node + one local);
//     2. pull : every thread polls the words of ITS elements in all N rows of my mailbox until they carry seq (bounded by a
//               wall-clock budget) and sums them in RANK ORDER (every rank adds the same numbers in the same order:
//               bit-identical results on all ranks, run to run -- RCCL's ring / tree order is neither).
//   Two slots suffice: a rank can only start seq + 2 (which reuses the slot) after completing seq + 1, i.e. after every peer
//   has pushed seq + 1, which a peer does only after finishing its reads of seq.  The sequence number lives in DEVICE memory
//   and is advanced by the kernel itself, so a launch captured in a hipGraph replays correctly (no host-side argument
//   changes between replays); all ranks issue the same sequence of calls, as with any collective.
// Cost: one launch, the push of 2 x N x bytes per rank and one xGMI round trip of latency -- no proxy thread, no host
// involvement, no fence, no second kernel.  world <= 16, count * elem <= max_bytes.
//
// Hardware facts used (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): per-XCD L2s are
// not coherent with each other or with another device's writes for ordinary (coarse-grained) allocations, so the mailbox is
// hipDeviceMallocFinegrained memory and every access to it is a system-scope atomic (sc0 sc1: past the caches).
#include <cstring>

#include "common.h"

namespace {

constexpr int kIpcMaxWorld = 16;
constexpr int kIpcHeader = 256;                 // bytes: [0] u64 sequence counter (owner only), [1] u64 error word

struct IpcP {
    unsigned char* mbox[kIpcMaxWorld];          // peer-mapped mailboxes, [rank] = my own
    const void* in;
    void* out;
    int count, elem;                            // elements, bytes per element (4: float, 8: double)
    int rank, world;
    long slot_bytes, row_bytes;                 // one slot = world rows; one row = 2 x max_bytes (8-byte word per 32-bit half)
    long long timeout_ticks;                    // wall_clock64 ticks (100 MHz) a poll may take; <= 0: unbounded
    int* status;                                // device int: set to 1 on timeout (sticky), untouched otherwise
};

__device__ __forceinline__ unsigned long long* ipc_row(const IpcP& q, int peer, int slot, int row) {
    return reinterpret_cast<unsigned long long*>(q.mbox[peer] + kIpcHeader + (size_t)slot * q.slot_bytes + (size_t)row * q.row_bytes);
}

template <typename T>
__global__ void __launch_bounds__(256) ipc_allreduce_kernel(const IpcP q) {
    constexpr int WPE = sizeof(T) / 4;          // LL words per element
    unsigned long long* my_hdr = reinterpret_cast<unsigned long long*>(q.mbox[q.rank]);
    // (only this kernel, one launch at a time on the stream, touches the counter: a plain uniform load)
    const unsigned long long seq = __hip_atomic_load(my_hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const unsigned seq32 = (unsigned)seq;
    const int slot = (int)(seq & 1);
    const unsigned* in32 = static_cast<const unsigned*>(q.in);
    const int nwords = q.count * WPE;
    // ---- 1. push my vector, one self-validating word per 32-bit half, into row `rank` of every mailbox
    for (int w = threadIdx.x; w < nwords; w += 256) {
        const unsigned long long word = (unsigned long long)in32[w] | ((unsigned long long)seq32 << 32);
        for (int p = 0; p < q.world; ++p)
            __hip_atomic_store(ipc_row(q, p, slot, q.rank) + w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- 2. pull: the words of my elements from all rows of MY mailbox -- all rows in flight at once (an access past the
    //         caches is a ~2 us round trip), re-polled together until complete -- summed in rank order
    T* out = static_cast<T*>(q.out);
    bool fail = false;
    for (int i = threadIdx.x; i < q.count; i += 256) {
        if (fail) {                             // a peer is gone: every remaining element of this thread is poisoned too
            out[i] = (T)__builtin_nanf("");
            continue;
        }
        unsigned long long w[kIpcMaxWorld][WPE];
        const long long t0 = wall_clock64();
        while (true) {
            bool ready = true;
#pragma unroll
            for (int r = 0; r < kIpcMaxWorld; ++r)
                if (r < q.world) {
                    const unsigned long long* src = ipc_row(q, q.rank, slot, r) + (size_t)i * WPE;
#pragma unroll
                    for (int h = 0; h < WPE; ++h) w[r][h] = __hip_atomic_load(src + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
#pragma unroll
            for (int r = 0; r < kIpcMaxWorld; ++r)
                if (r < q.world) {
#pragma unroll
                    for (int h = 0; h < WPE; ++h) ready = ready && (unsigned)(w[r][h] >> 32) == seq32;
                }
            if (ready) break;
            if (q.timeout_ticks > 0 && wall_clock64() - t0 > q.timeout_ticks) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (fail) {
            // gave up: the result must not look like a sum.  NaN poisons everything computed from it (the step's loss), the
            // sticky status word makes shard.SmallAllReduce.check() / poll() raise.
            out[i] = (T)__builtin_nanf("");
            continue;
        }
        T acc = 0;
#pragma unroll
        for (int r = 0; r < kIpcMaxWorld; ++r)
            if (r < q.world) {
                if (WPE == 1) acc += (T)__uint_as_float((unsigned)w[r][0]);
                else acc += (T)__longlong_as_double((long long)((w[r][WPE - 1] << 32) | (w[r][0] & 0xffffffffULL)));
            }
        out[i] = acc;
    }
    if (fail && q.status != nullptr) *q.status = 1;
    __syncthreads();                            // every thread has read its words of this sequence number
    if (threadIdx.x == 0) __hip_atomic_store(my_hdr, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void ipc_zero_kernel(unsigned long long* p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

}  // namespace

extern "C" {

// Bytes a mailbox needs for `world` ranks exchanging vectors of up to `max_bytes` each (64-byte aligned rows).
int64_t occd_ipc_mailbox_bytes(int32_t world, int64_t max_bytes) {
    if (world < 1 || world > kIpcMaxWorld || max_bytes < 8) return OCCD_EINVAL;
    const int64_t row = 2 * (((max_bytes + 63) / 64) * 64);          // LL: an 8-byte word per 32-bit half of the payload
    return kIpcHeader + 2 * (int64_t)world * row;
}

// Allocate this rank's mailbox (fine-grained device memory on the current device), zero it, export its IPC handle (64 bytes).
int occd_ipc_mailbox_create(int64_t bytes, void** mailbox, void* handle64) {
    if (bytes < kIpcHeader + 128 || mailbox == nullptr || handle64 == nullptr) return OCCD_EINVAL;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess || p == nullptr) {
        (void)hipGetLastError();
        return OCCD_ELAUNCH;
    }
    const long n = bytes / 8;
    hipLaunchKernelGGL(ipc_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                       static_cast<unsigned long long*>(p), n);
    if (hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return OCCD_ELAUNCH;
    }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return OCCD_ELAUNCH;
    }
    std::memcpy(handle64, &h, 64);
    *mailbox = p;
    return OCCD_OK;
}

int occd_ipc_mailbox_open(const void* handle64, void** peer) {
    if (handle64 == nullptr || peer == nullptr) return OCCD_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, 64);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || p == nullptr) {
        (void)hipGetLastError();
        return OCCD_ELAUNCH;
    }
    *peer = p;
    return OCCD_OK;
}

int occd_ipc_mailbox_close(void* peer) {
    if (peer == nullptr) return OCCD_EINVAL;
    return hipIpcCloseMemHandle(peer) == hipSuccess ? OCCD_OK : OCCD_ELAUNCH;
}

int occd_ipc_mailbox_free(void* mailbox) {
    if (mailbox == nullptr) return OCCD_EINVAL;
    return hipFree(mailbox) == hipSuccess ? OCCD_OK : OCCD_ELAUNCH;
}

// out[i] = sum over ranks of in[i] (rank order), count elements of dtype (0: float32, 1: float64); in / out device pointers of
// this rank (may alias); mailboxes: `world` peer-mapped mailbox pointers (HOST array, [rank] = this rank's own) of
// occd_ipc_mailbox_bytes(world, max_bytes) bytes each; status: optional device int set to 1 when the wait gives up after
// timeout_ms (<= 0: wait forever) -- the elements whose wait gave up are then NaN in `out`, never a partial sum.  Asynchronous on `stream`; capturable (the sequence number lives in the mailbox).
int occd_ipc_allreduce(const void* in, void* out, int64_t count, int32_t dtype, void* const* mailboxes, int32_t rank,
                       int32_t world, int64_t max_bytes, int32_t timeout_ms, int32_t* status, void* stream) {
    if (in == nullptr || out == nullptr || mailboxes == nullptr || count < 1 || (dtype != 0 && dtype != 1)) return OCCD_EINVAL;
    if (world < 1 || world > kIpcMaxWorld || rank < 0 || rank >= world) return OCCD_EINVAL;
    const int elem = dtype == 0 ? 4 : 8;
    if (count * elem > max_bytes || count > 0x7fffffffL) return OCCD_EINVAL;
    IpcP q{};
    for (int p = 0; p < world; ++p) {
        if (mailboxes[p] == nullptr) return OCCD_EINVAL;
        q.mbox[p] = static_cast<unsigned char*>(mailboxes[p]);
    }
    q.in = in; q.out = out; q.count = (int)count; q.elem = elem; q.rank = rank; q.world = world;
    q.row_bytes = 2 * (((max_bytes + 63) / 64) * 64);
    q.slot_bytes = (long)world * q.row_bytes;
    q.timeout_ticks = timeout_ms > 0 ? (long long)timeout_ms * 100000LL : 0;      // wall_clock64: 100 MHz
    q.status = status;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("ipc_allreduce", st, 0.0, (double)count * elem * (2.0 * world));
    if (dtype == 0) hipLaunchKernelGGL(ipc_allreduce_kernel<float>, dim3(1), dim3(256), 0, st, q);
    else hipLaunchKernelGGL(ipc_allreduce_kernel<double>, dim3(1), dim3(256), 0, st, q);
    return occd::check_launch();
}

}  // extern "C"
