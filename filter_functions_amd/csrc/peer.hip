// peer.hip -- one-sided all-gather of the filter-function blocks over xGMI, without RCCL.
//
// The frequency-sharded step (reference path: every omega of numeric.py:846-869 is
// independent, only util.integrate couples them) ends in an all-gather of F (A x A x W/n complex
// per rank).  RCCL's all-gather is a kernel with its own LDS and register budget; the accumulate
// kernel holds 140 of every CU's 160 KiB of LDS, so the collective can only run in the gaps between
// two accumulate launches or delays the next one.  Here every rank PUSHES its
// block into the gather buffers of all ranks through IPC-mapped pointers with a plain copy kernel
// (no LDS, 32 VGPRs: co-resident with the accumulate kernel), and completion travels as sequence
// numbers in flag words that the consumer polls:
//   push(c):  wait until every peer has acknowledged the step that used this buffer set last,
//             copy F into slot `rank` of the set on every rank;
//   signal(c): (next launch: the copy kernel has completed and released its writes) store c + 1 to
//             flag[rank] on every rank, and "c steps consumed" to ack[rank] on every rank;
//   wait(c):  poll the local flags until all ranks have signalled c + 1; then the integral runs.
// Every poll loop has a wall-clock timeout (2 s by default: FFK_PEER_TIMEOUT_MS / ffk_peer_set_timeout_ms)
// that raises a STICKY error word instead of hanging; a rank whose error word is set publishes a
// poison value instead of sequence numbers from then on, so its peers fail as well (code 3) instead
// of integrating a slot that was never filled.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ffk.h"
#include "ffk_internal.h"

namespace ffk {
namespace {

constexpr long long kPoison = -1;      // a rank that failed publishes this instead of a sequence number
long long g_poll_timeout_ticks = 0;    // wall_clock64 ticks (100 MHz); 0 = not initialised yet

__device__ __forceinline__ long long load_system(const long long* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_system(long long* p, long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the first failure stays: 1 = acknowledgement timed out (push), 2 = signal timed out (wait),
// 3 = a peer reported a failure of its own
__device__ __forceinline__ void raise_error(int* error, int code) { atomicCAS(error, 0, code); }

// 0 when *p >= want within the timeout, 1 on timeout, 3 when the word holds the poison value
__device__ int poll_at_least(const long long* p, long long want, long long timeout_ticks) {
    const long long t0 = wall_clock64();
    for (;;) {
        const long long v = load_system(p);
        if (v == kPoison) return 3;
        if (v >= want) return 0;
        if (wall_clock64() - t0 > timeout_ticks) return 1;
        __builtin_amdgcn_s_sleep(8);
    }
}

// grid (nblk, world): block (b, p) copies its share of src to dst[p]; 16 bytes per thread and step
__global__ __launch_bounds__(256) void peer_push_kernel(const double2* __restrict__ src, size_t n16,
                                                        double2* const* __restrict__ dst,
                                                        const long long* __restrict__ acks,
                                                        long long need_ack, int rank,
                                                        long long timeout_ticks,
                                                        int* __restrict__ error) {
    __builtin_amdgcn_s_setprio(3);     // runs beside accumulate kernels: see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int p = blockIdx.y;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        if (p != rank && need_ack > 0) {
            const int bad = poll_at_least(acks + p, need_ack, timeout_ticks);
            if (bad) {
                ok = 0;
                raise_error(error, bad);
            }
        }
    }
    __syncthreads();
    if (!ok) return;       // the slot on rank p may still be read: nothing is written there
    double2* out = dst[p];
    for (size_t i = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x; i < n16;
         i += static_cast<size_t>(gridDim.x)*blockDim.x)
        out[i] = src[i];
}

// one wavefront: lane p tells rank p "my block of step seq - 1 is in place" and "I have consumed
// `consumed` steps".  If this rank's error word is set (a push of this or an earlier step was
// skipped, a wait timed out) it publishes the poison value instead: the peers' polls then fail at
// once with code 3 rather than integrating a slot this rank never filled, and since the error word
// is sticky every later step says the same.
__global__ __launch_bounds__(64) void peer_signal_kernel(long long* const* __restrict__ flags,
                                                         long long* const* __restrict__ acks,
                                                         int world, long long seq, long long consumed,
                                                         const int* __restrict__ error) {
    __builtin_amdgcn_s_setprio(3);
    const int p = threadIdx.x;
    if (p >= world) return;
    const bool failed = __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __threadfence_system();
    store_system(flags[p], failed ? kPoison : seq);
    store_system(acks[p], failed ? kPoison : consumed);
}

// one wavefront: lane p waits for rank p's signal
__global__ __launch_bounds__(64) void peer_wait_kernel(const long long* __restrict__ flags, int world,
                                                       long long seq, long long timeout_ticks,
                                                       int* __restrict__ error) {
    __builtin_amdgcn_s_setprio(3);     // (the poll sleeps between reads: it does not take the slots it may)
    const int p = threadIdx.x;
    if (p < world) {
        const int bad = poll_at_least(flags + p, seq, timeout_ticks);
        if (bad) raise_error(error, bad == 1 ? 2 : bad);
    }
    __threadfence_system();
}

long long poll_timeout_ticks() {
    if (g_poll_timeout_ticks <= 0) {
        double ms = 2000.0;
        if (const char* env = std::getenv("FFK_PEER_TIMEOUT_MS")) {
            const double v = std::atof(env);
            if (v > 0) ms = v;
        }
        g_poll_timeout_ticks = static_cast<long long>(ms*1e5);
    }
    return g_poll_timeout_ticks;
}

}  // namespace
}  // namespace ffk

extern "C" {

namespace {
int peer_fail(int code, const char* what, hipError_t e) {
    char message[256];
    snprintf(message, sizeof message, "%s failed: %s", what, hipGetErrorString(e));
    ffk::set_last_error(message);
    return code;
}
}  // namespace

int ffk_ipc_get_handle(const void* dptr, void* handle) {
    if (!dptr || !handle) return FFK_EINVAL;
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, const_cast<void*>(dptr));
    if (e != hipSuccess) return peer_fail(FFK_EHIP, "hipIpcGetMemHandle", e);
    static_assert(sizeof(hipIpcMemHandle_t) == FFK_IPC_HANDLE_BYTES, "handle size");
    std::memcpy(handle, &h, sizeof h);
    return FFK_OK;
}

int ffk_ipc_open_handle(const void* handle, void** dptr) {
    if (!dptr || !handle) return FFK_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    hipError_t e = hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return peer_fail(FFK_EHIP, "hipIpcOpenMemHandle", e);
    return FFK_OK;
}

int ffk_ipc_close_handle(void* dptr) {
    if (!dptr) return FFK_OK;
    hipError_t e = hipIpcCloseMemHandle(dptr);
    if (e != hipSuccess) return peer_fail(FFK_EHIP, "hipIpcCloseMemHandle", e);
    return FFK_OK;
}

int ffk_peer_push_dev(const double* src, size_t bytes, void* const* dst, const int64_t* acks,
                      int64_t need_ack, int world, int rank, int32_t* error, void* stream) {
    if (!src || !dst || !acks || !error || world < 1 || world > 64 || rank < 0 || rank >= world ||
        bytes % 16 != 0)
        return FFK_EINVAL;
    const size_t n16 = bytes/16;
    const unsigned nblk = static_cast<unsigned>(std::min<size_t>(32, (n16 + 255)/256));
    hipLaunchKernelGGL(ffk::peer_push_kernel, dim3(std::max(1u, nblk), world), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const double2*>(src), n16,
                       reinterpret_cast<double2* const*>(dst),
                       reinterpret_cast<const long long*>(acks), static_cast<long long>(need_ack), rank,
                       ffk::poll_timeout_ticks(), error);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFK_OK : peer_fail(FFK_EHIP, "peer_push_kernel", e);
}

int ffk_peer_set_timeout_ms(double ms) {
    if (!(ms > 0)) return FFK_EINVAL;
    ffk::g_poll_timeout_ticks = static_cast<long long>(ms*1e5);
    return FFK_OK;
}

int ffk_peer_signal_dev(void* const* flags, void* const* acks, int world, int64_t seq, int64_t consumed,
                        const int32_t* error, void* stream) {
    if (!flags || !acks || !error || world < 1 || world > 64) return FFK_EINVAL;
    hipLaunchKernelGGL(ffk::peer_signal_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<long long* const*>(flags),
                       reinterpret_cast<long long* const*>(acks), world, static_cast<long long>(seq),
                       static_cast<long long>(consumed), error);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFK_OK : peer_fail(FFK_EHIP, "peer_signal_kernel", e);
}

int ffk_peer_wait_dev(const int64_t* flags, int world, int64_t seq, int32_t* error, void* stream) {
    if (!flags || !error || world < 1 || world > 64) return FFK_EINVAL;
    hipLaunchKernelGGL(ffk::peer_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(flags), world, static_cast<long long>(seq),
                       ffk::poll_timeout_ticks(), error);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FFK_OK : peer_fail(FFK_EHIP, "peer_wait_kernel", e);
}

// push + signal + wait of one step in one call (three launches on `stream`; one trip through the
// binding instead of three: the host side of a sharded step is what bounds its rate)
int ffk_peer_step_dev(const double* src, size_t bytes, void* const* dst, const int64_t* own_acks,
                      int64_t need_ack, void* const* flag_at, void* const* ack_at,
                      const int64_t* own_flags, int world, int rank, int64_t step, int32_t* error,
                      void* stream) {
    if (int rc = ffk_peer_push_dev(src, bytes, dst, own_acks, need_ack, world, rank, error, stream)) return rc;
    if (int rc = ffk_peer_signal_dev(flag_at, ack_at, world, step + 1, step, error, stream)) return rc;
    return ffk_peer_wait_dev(own_flags, world, step + 1, error, stream);
}

}  // extern "C"
