// ffk_api_common.h -- what the translation units of the extern "C" surface (ffk_api*.hip) share:
// error reporting, argument checks, the bump allocator over workspaces, the process-wide staging
// arena, workspace layouts used by more than one family of entry points.
//
//   ffk_api.hip           device / stream utilities, diagonalize, control matrix, filter function,
//                         infidelity, Liouville representation, the fused pipeline pass (SURVEY 8 a)
//   ffk_api_sequence.hip  concatenation rule, decay amplitudes, cumulant function, expm (SURVEY 8 f)
//   ffk_api_resident.hip  resident results behind PulseSequence / ff.infidelity; host self test
//   ffk_api_frozen.hip    second order and gradient (outside SURVEY 8, frozen since round 2)
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ffk.h"
#include "ffk_internal.h"

#if defined(FFK_HOST_SANITIZE)
// Host-side sanitizer variant (make VARIANT=asan ...; tools/build_asan.sh): the allocation calls of
// the arena and of the block pools go to the C heap, so that their bookkeeping -- growth, reuse,
// eviction, the slicing of every workspace layout -- can run under AddressSanitizer / UBSan on a
// machine without a GPU (ffk_selftest_host below).  Never part of the shipped library.
#include <cstdlib>
namespace {
hipError_t stub_alloc(void** p, size_t n) {
    *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t stub_free(void* p) {
    std::free(p);
    return hipSuccess;
}
}  // namespace
#define hipMalloc(p, n) stub_alloc(reinterpret_cast<void**>(p), (n))
#define hipHostMalloc(p, n, flags) stub_alloc(reinterpret_cast<void**>(p), (n))
#define hipFree(p) stub_free(p)
#define hipHostFree(p) stub_free(p)
#define hipGetDevice(d) ((*(d) = 0), hipSuccess)
#define hipSetDevice(d) hipSuccess
#define hipDeviceSynchronize() hipSuccess
#endif

using ffk::align_up;
using ffk::cplx;

namespace ffk_api {
using ffk::align_up;
using ffk::cplx;

extern thread_local std::string g_error;
extern thread_local ffk_stats g_stats;
// bumped by every call that changes how a pass is enqueued (tuning knobs, instrumentation events):
// captured passes are keyed on it (resident_pass)
extern std::atomic<unsigned long long> g_knob_epoch;
extern int g_forced_chunks;
extern thread_local hipEvent_t g_ev_start, g_ev_stop, g_ev_gate;

int fail(int code, const char* fmt, ...);
// the kernels' sticky fault word (ffk::kernel_fault_word, ffk_internal.h), read AFTER a synchronisation:
// FFK_OK, or FFK_EKERNEL with the message set and the word cleared; peek: the raw word
int kernel_fault_status();
int kernel_fault_peek(bool clear);
// at the top of a host-pointer entry point: a fault left by an earlier, never checked asynchronous launch of this
// thread is reported as such (and cleared) instead of being attributed to the call that is about to run
int kernel_fault_stale();
int kernel_fault_slot_for_selftest();


#define FFK_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? FFK_ENOMEM : FFK_EHIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                     \
    } while (0)

#define FFK_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return fail(FFK_EINVAL, __VA_ARGS__); \
    } while (0)

inline bool d_ok(int d) { return d >= 2 && d <= FFK_MAX_D; }
// entry points whose kernels are compiled per dimension (see include/ffk.h)
inline bool d_templated_ok(int d) { return d >= 2 && d <= FFK_MAX_D_TEMPLATED; }

// internal flag of ffk_control_matrix_dev: the workspace already holds segtab/Tc/ops (written by
// the fused front end of ffk_pipeline_dev)
constexpr unsigned FFK_INTERNAL_PROLOGUE_DONE = 0x80000000u;
// ... and the compacted basis lists in the expansion workspace (same launch)
constexpr unsigned FFK_INTERNAL_COMPACT_DONE = 0x40000000u;
// What a caller INSIDE the library asks of the device-pointer entry points it is built from (the resident pass of
// ffk_api_resident.hip, ffk_pipeline_dev around its own stages).  Passed down explicitly to the *_impl functions
// below; the extern "C" entry points call them with the defaults.  (Rounds 3-5: seven thread_local variables.)
struct EighControls {
    const ffk::cplx* opers;      // (n_c, d, d), device
    const double* coeffs;        // (n_c, G), device
    int n_c;
};
struct PassOptions {
    // where the fidelity filter function should go if the control matrix's expansion launch can produce it too,
    // and (out) whether it did
    ffk::cplx* fuse_F = nullptr;
    bool fuse_F_done = false;
    // the eigensolver counts flagged segments into this mapped host word (no memset, no counting kernel)
    int* eigh_fail_count = nullptr;
    // the pipeline's Hamiltonian as its summands: the eigensolver kernel sums the control operators itself
    EighControls eigh_controls = {nullptr, nullptr, 0};
    // spectrum and idx of the infidelity integral are in mapped host memory (staged through LDS by the kernel)
    bool infid_spectrum_on_host = false;
};
int diagonalize_dev_impl(const double* hamiltonian, const double* dt, int G, int d, double* eigvals, double* eigvecs,
                         double* propagators, void* workspace, size_t workspace_bytes, void* stream,
                         const PassOptions& opt);
int control_matrix_dev_impl(const double* eigvals, const double* eigvecs, const double* propagators,
                            const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                            const double* n_coeffs, const double* dt, const double* t, int G, int d, unsigned flags,
                            double* control_matrix, double* noise_operators, void* workspace,
                            size_t workspace_bytes, void* stream, PassOptions& opt);
int infidelity_dev_impl(const double* filter_function, int A, int W, const double* spectrum, int s_ndim,
                        const double* omega, const int32_t* idx, int n_idx, int d, double* infid, void* workspace,
                        size_t workspace_bytes, void* stream, const PassOptions& opt);
int pipeline_dev_impl(const double* hamiltonian, const double* dt, const double* t, int G, int d,
                      const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                      const double* n_coeffs, const double* spectrum, int s_ndim, const int32_t* idx, int n_idx,
                      double* eigvals, double* eigvecs, double* propagators, double* control_matrix,
                      double* filter_function, double* infid, void* workspace, size_t workspace_bytes,
                      void* stream, PassOptions& opt);

// Scratch from the shared arena is handed to kernels on a non-blocking stream while g_arena.mu is
// held; the lock may only be dropped once that stream has drained -- on EVERY exit path, also the
// early error returns after the first enqueue: the next holder may reuse or reallocate
// the arena.  Declared after the lock_guard, so it runs before the lock is released.
struct StreamDrain {
    hipStream_t stream;
    ~StreamDrain() { (void)hipStreamSynchronize(stream); }
};

// bump allocator over a caller- or arena-provided workspace
struct Bump {
    unsigned char* base;
    size_t size, used = 0;
    Bump(void* p, size_t n) : base(static_cast<unsigned char*>(p)), size(n) {}
    template <typename T>
    T* take(size_t count) {
        const size_t bytes = align_up(count*sizeof(T));
        if (used + bytes > size) return nullptr;
        T* out = reinterpret_cast<T*>(base + used);
        used += bytes;
        return out;
    }
};

// process-wide arena for the host-pointer flavour
struct Arena {
    std::mutex mu;
    void* ptr = nullptr;
    size_t size = 0;
    int device = -1;
};
extern Arena g_arena;
int arena_reserve(size_t bytes, void** out);

// --- workspace layouts ------------------------------------------------------------------------
size_t ctrl_ws_bytes(int W, int N, int A, int G, int d, int chunks);
int max_chunks_for(int W, int A, int G, int d);
double accumulate_flops(int W, int A, int G, int d);

// diagonalize workspace: [status: G ints][seg_prop: G d^2][Qloc: (G+1) d^2][totals / scan scratch]
struct DiagWs {
    int* status;
    cplx* seg_prop;
    cplx* qloc;
    void* small;
};
DiagWs slice_diag_ws(void* workspace, size_t bytes, int G, int d);

// ffk_api_sequence.hip: one concatenation of a sequence drawn from T pulses, operands on the device
size_t sequence_scratch_bytes(int G, int d, int A, int N, int W, int which, bool hermitian, bool want_F);
int sequence_on_device(const double* dU, const double* dP, const double* dR, const int32_t* dI,
                       const double* dB, int hermitian_basis, int T, int G, int d, int A, int N, int W,
                       int which, Bump& a, double* control_matrix, double* total_propagator,
                       double* propagators_liouville, double* filter_function, hipStream_t s,
                       double* resident_R = nullptr, double* resident_F = nullptr,
                       const cplx* const* dRtab = nullptr, const double* dTau = nullptr,
                       const double* dOmega = nullptr, double* omega_copy = nullptr);
}  // namespace ffk_api

using namespace ffk_api;
