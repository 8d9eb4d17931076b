// occd_graph_replace_memsets: rewrite every MEMSET node of a captured hipGraph as a KERNEL node (round 6).
//
// Why: on this stack (ROCm 7.2, gfx950) a hipMemsetAsync captured into a hipGraph fills with the right value on the FIRST
// launch of the instantiated graph only; from the second launch on the fill pattern is read from recycled host/kernarg memory
// (tools/probe_graph_memset*.py: a lone 4-byte memset of 0 writes 0x04040404 on replays 1, 2, ...; a 64-byte one writes
// {size, value | 0x7700, ...}).  ATen's multi-block reductions zero their semaphores with exactly such a node
// (ATen/native/hip/Reduce.cuh: launch_reduce_kernel), so a `sum` over many rows inside the captured training step
// (train_graph.py) -- e.g. the bias gradient of the full-resolution `occ_classes` convolution -- never sees "last block done"
// when the stale pattern happens to be non-zero and leaves its output unwritten: the intermittent NaN of
// tests/test_train_step.py::test_whole_step_hipgraph_matches_eager_gpu (about 1 in 12 fresh processes after any change of
// the step's allocation pattern), and the K5 statistics memset of round 4.  Kernel nodes replay
// correctly, so the node is replaced by a fill KERNEL with the same destination, value and edges before instantiation.
// The fault sits in the runtime's graph packet capture (launches after the first replay the AQL packets recorded by the
// first): with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment the same probe shows 0 of 90 bad nodes against 90 of 90
// (profiles/r06_graph_memset_probe.txt).  The packet capture is what makes replays cheap, so it stays on.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace {

// 1-D fill of `bytes` bytes at dst with a 32-bit pattern: byte (addr & 3) of `pat` is the value of the byte at `addr`.
__global__ void graph_fill_kernel(unsigned char* __restrict__ dst, unsigned pat, size_t bytes) {
    const size_t head = ((16 - ((size_t)dst & 15)) & 15) < bytes ? ((16 - ((size_t)dst & 15)) & 15) : bytes;
    const size_t body = (bytes - head) / 16;
    const size_t tail = bytes - head - body * 16;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    if (tid < head) dst[tid] = (unsigned char)(pat >> (8 * (((size_t)dst + tid) & 3)));
    uint4* b = reinterpret_cast<uint4*>(dst + head);
    const uint4 v = make_uint4(pat, pat, pat, pat);
    for (size_t i = tid; i < body; i += nthr) b[i] = v;
    unsigned char* t = dst + head + body * 16;
    if (tid < tail) t[tid] = (unsigned char)(pat >> (8 * (((size_t)t + tid) & 3)));
}

// general (2-D / pitched) form: one element per thread step
__global__ void graph_fill2d_kernel(unsigned char* __restrict__ dst, unsigned value, unsigned esz, size_t width, size_t height,
                                    size_t pitch) {
    const size_t n = width * height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned char* p = dst + (i / width) * pitch + (i % width) * esz;
        for (unsigned k = 0; k < esz; ++k) p[k] = (unsigned char)(value >> (8 * k));
    }
}

}  // namespace

extern "C" int occd_graph_replace_memsets(void* graph_handle) {
    hipGraph_t graph = static_cast<hipGraph_t>(graph_handle);
    if (graph == nullptr) return OCCD_EINVAL;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return OCCD_ELAUNCH;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return OCCD_ELAUNCH;
    int replaced = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return OCCD_ELAUNCH;
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams p;
        if (hipGraphMemsetNodeGetParams(nodes[i], &p) != hipSuccess) return OCCD_ELAUNCH;
        if (p.elementSize != 1 && p.elementSize != 2 && p.elementSize != 4) return OCCD_EINVAL;
        size_t nd = 0, nq = 0;
        if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) return OCCD_ELAUNCH;
        std::vector<hipGraphNode_t> deps(nd);
        if (nd && hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) return OCCD_ELAUNCH;
        if (hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nq) != hipSuccess) return OCCD_ELAUNCH;
        std::vector<hipGraphNode_t> users(nq);
        if (nq && hipGraphNodeGetDependentNodes(nodes[i], users.data(), &nq) != hipSuccess) return OCCD_ELAUNCH;

        hipKernelNodeParams kp = {};
        unsigned char* dst = static_cast<unsigned char*>(p.dst);
        unsigned value = p.value, esz = p.elementSize, pat;
        // debugging only (tools/probe_train_graph_memset.py): OCCD_DBG_MEMSET_XOR=<n> perturbs every fill value the way the
        // runtime's stale pattern does, to show WHAT a captured step loses when its memset nodes misfire
        if (const char* e = getenv("OCCD_DBG_MEMSET_XOR")) value ^= (unsigned)strtoul(e, nullptr, 0);
        size_t width = p.width, height = p.height ? p.height : 1, pitch = p.pitch, bytes;
        void* args1[3];
        void* args2[6];
        const bool flat = height == 1 || pitch == width * esz;
        if (flat) {
            bytes = width * esz * height;
            // the element replicated over a 32-bit word: with dst aligned to the element, pattern byte (addr & 3) lands on addr
            pat = esz == 1 ? (value & 0xff) * 0x01010101u : esz == 2 ? (value & 0xffff) * 0x00010001u : value;
            if (((size_t)dst % esz) != 0) return OCCD_EINVAL;
            args1[0] = &dst, args1[1] = &pat, args1[2] = &bytes;
            kp.func = reinterpret_cast<void*>(graph_fill_kernel);
            kp.kernelParams = args1;
            const size_t chunks = bytes / 16 + 1;
            kp.gridDim = dim3((unsigned)((chunks + 255) / 256 < 2048 ? (chunks + 255) / 256 : 2048));
        } else {
            args2[0] = &dst, args2[1] = &value, args2[2] = &esz, args2[3] = &width, args2[4] = &height, args2[5] = &pitch;
            kp.func = reinterpret_cast<void*>(graph_fill2d_kernel);
            kp.kernelParams = args2;
            const size_t cells = width * height;
            kp.gridDim = dim3((unsigned)((cells + 255) / 256 < 2048 ? (cells + 255) / 256 : 2048));
        }
        if (kp.gridDim.x == 0) kp.gridDim.x = 1;
        kp.blockDim = dim3(256);
        hipGraphNode_t fill;
        if (hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp) != hipSuccess) return OCCD_ELAUNCH;
        for (size_t u = 0; u < nq; ++u)
            if (hipGraphAddDependencies(graph, &fill, &users[u], 1) != hipSuccess) return OCCD_ELAUNCH;
        if (hipGraphDestroyNode(nodes[i]) != hipSuccess) return OCCD_ELAUNCH;
        ++replaced;
    }
    return replaced;
}
