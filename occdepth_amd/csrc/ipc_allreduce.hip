// Small-message all-reduce over peer-mapped device memory (the training step's SyncBatchNorm
// exchanges of occdepth/scripts/train.py:176-206 -- `Trainer(sync_batchnorm=True)`).
//
// The config-2 training step makes 532 all-reduces of (2C + 1) statistics, C <= 3840: a few hundred bytes to 60 KB each, all
// on ONE dependency chain (layer i's backward sums need layer i + 1's), so they cannot be batched, and each costs RCCL's
// launch + completion latency (~45 us inside a captured step: +16 % on one rank, 0.72 - 0.78 weak-scaling efficiency
// budgeted for 8 GPUs).  This is the latency-optimal exchange for that regime, one kernel per call, no library in the path:
//
//   every rank owns a MAILBOX in its own HBM -- 2 slots x world x (flag, payload) -- allocated fine-grained (coherent across
//   XCDs and over xGMI while a kernel runs) and exported with hipIpcGetMemHandle; every rank maps all peers' mailboxes
//   (hipIpcOpenMemHandle) once, at set-up.  all_reduce(seq):
//     1. push : write my vector into slot seq & 1, row `rank`, of EVERY mailbox (N - 1 remote writes over the N - 1 direct
//               xGMI links of a fully connected node + one local), system-scope release, then store seq into the row's flag;
//     2. wait : spin (system-scope acquire loads, bounded by a wall-clock budget) until all N flags of my slot hold seq;
//     3. sum  : out = sum over rows in RANK ORDER (every rank adds the same numbers in the same order: bit-identical results
//               on all ranks, run to run -- RCCL's ring / tree order is neither).
//   Two slots suffice: a rank can only start seq + 2 (which reuses the slot) after completing seq + 1, i.e. after every peer
//   has pushed seq + 1, which a peer does only after finishing its reads of seq.  The sequence number lives in DEVICE memory
//   and is advanced by the kernel itself, so a launch captured in a hipGraph replays correctly (no host-side argument
//   changes between replays); all ranks issue the same sequence of calls, as with any collective.
// Cost: one launch, the push of N x bytes per rank and one xGMI round trip of latency -- no proxy thread, no host
// involvement, no second kernel.  world <= 16, count * elem <= the slot's payload capacity.
//
// Hardware facts used (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): per-XCD L2s are
// not coherent with each other or with another device's writes for ordinary (coarse-grained) allocations, so the mailbox is
// hipDeviceMallocFinegrained memory and every flag access is a system-scope atomic; payload stores are followed by a
// system-scope fence before the flag store, payload loads preceded by one after the flag loads.
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
    long slot_bytes, row_bytes;                 // one slot = world rows; one row = 64-byte flag line + payload
    long long timeout_ticks;                    // wall_clock64 ticks (100 MHz) the wait may take; <= 0: unbounded
    int* status;                                // device int: set to 1 on timeout (sticky), untouched otherwise
};

__device__ __forceinline__ unsigned char* ipc_row(const IpcP& q, int peer, int slot, int row) {
    return q.mbox[peer] + kIpcHeader + (size_t)slot * q.slot_bytes + (size_t)row * q.row_bytes;
}

template <typename T>
__global__ void __launch_bounds__(256) ipc_allreduce_kernel(const IpcP q) {
    __shared__ unsigned long long s_seq;
    __shared__ int s_fail;
    unsigned long long* my_hdr = reinterpret_cast<unsigned long long*>(q.mbox[q.rank]);
    if (threadIdx.x == 0) {
        s_seq = my_hdr[0] + 1;                  // (only this kernel, one launch at a time on the stream, touches the counter)
        s_fail = 0;
    }
    __syncthreads();
    const unsigned long long seq = s_seq;
    const int slot = (int)(seq & 1);
    const T* in = static_cast<const T*>(q.in);
    // ---- 1. push my vector into row `rank` of every mailbox
    for (int p = 0; p < q.world; ++p) {
        T* dst = reinterpret_cast<T*>(ipc_row(q, p, slot, q.rank) + 64);
        for (int i = threadIdx.x; i < q.count; i += 256) dst[i] = in[i];
    }
    __threadfence_system();                     // payload visible system-wide before any flag
    __syncthreads();
    if (threadIdx.x < q.world) {
        unsigned long long* flag = reinterpret_cast<unsigned long long*>(ipc_row(q, threadIdx.x, slot, q.rank));
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- 2. wait for every rank's flag in MY mailbox
    if (threadIdx.x < q.world) {
        const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(ipc_row(q, q.rank, slot, threadIdx.x));
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(2);
            if (q.timeout_ticks > 0 && wall_clock64() - t0 > q.timeout_ticks) {
                s_fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();                     // acquire for the payload reads of all threads
    // ---- 3. sum the rows in rank order
    T* out = static_cast<T*>(q.out);
    if (s_fail == 0) {
        for (int i = threadIdx.x; i < q.count; i += 256) {
            T acc = 0;
            for (int r = 0; r < q.world; ++r) {
                const T* src = reinterpret_cast<const T*>(ipc_row(q, q.rank, slot, r) + 64);
                acc += __builtin_nontemporal_load(src + i);
            }
            out[i] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        my_hdr[0] = seq;
        if (s_fail != 0) {
            my_hdr[1] = seq;                    // which exchange gave up
            if (q.status != nullptr) *q.status = 1;
        }
    }
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
    const int64_t row = 64 + ((max_bytes + 63) / 64) * 64;
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
// timeout_ms (<= 0: wait forever).  Asynchronous on `stream`; capturable (the sequence number lives in the mailbox).
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
    q.row_bytes = 64 + ((max_bytes + 63) / 64) * 64;
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
