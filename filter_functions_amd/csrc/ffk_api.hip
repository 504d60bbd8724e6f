// ffk_api.hip -- the extern "C" surface declared in include/ffk.h.
//
// *_dev entry points only slice the caller's workspace and enqueue kernels.  The host-pointer
// entry points stage through a process-wide, grow-only device arena and synchronise before
// returning, so that the NumPy-facing front-end has exactly the reference's call semantics
// (borrowed inputs, freshly written outputs).
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

namespace {

thread_local std::string g_error;
thread_local ffk_stats g_stats = {};
// bumped by every call that changes how a pass is enqueued (tuning knobs, instrumentation events):
// captured passes are keyed on it (resident_pass)
std::atomic<unsigned long long> g_knob_epoch{0};
int g_forced_chunks = 0;
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr, g_ev_gate = nullptr;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

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

bool d_ok(int d) { return d >= 2 && d <= FFK_MAX_D; }
// entry points whose kernels are compiled per dimension (see include/ffk.h)
bool d_templated_ok(int d) { return d >= 2 && d <= FFK_MAX_D_TEMPLATED; }

// internal flag of ffk_control_matrix_dev: the workspace already holds segtab/Tc/ops (written by
// the fused front end of ffk_pipeline_dev)
constexpr unsigned FFK_INTERNAL_PROLOGUE_DONE = 0x80000000u;
// ... and the compacted basis lists in the expansion workspace (same launch)
constexpr unsigned FFK_INTERNAL_COMPACT_DONE = 0x40000000u;
// set by ffk_pipeline_dev around its call of ffk_control_matrix_dev: where the fidelity filter
// function should go if the expansion launch can produce it too, and whether it did
thread_local cplx* g_fuse_F = nullptr;
thread_local bool g_fuse_F_done = false;

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
} g_arena;

int arena_reserve(size_t bytes, void** out) {
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    if (g_arena.ptr && (g_arena.device != dev || g_arena.size < bytes)) {
        FFK_HIP(hipDeviceSynchronize());
        FFK_HIP(hipFree(g_arena.ptr));
        g_arena.ptr = nullptr;
        g_arena.size = 0;
    }
    if (!g_arena.ptr) {
        const size_t want = align_up(bytes + bytes/4, size_t(1) << 20);
        FFK_HIP(hipMalloc(&g_arena.ptr, want));
        g_arena.size = want;
        g_arena.device = dev;
    }
    *out = g_arena.ptr;
    return FFK_OK;
}

// --- workspace layouts ------------------------------------------------------------------------
size_t ctrl_ws_bytes(int W, int N, int A, int G, int d, int chunks) {
    size_t b = 0;
    b += align_up(sizeof(double)*G*ffk::seg_stride(d));                   // segtab
    b += align_up(sizeof(cplx)*size_t(G)*d*d);                            // Tc
    b += align_up(sizeof(cplx)*size_t(G)*(1 + A)*d*d);                    // ops
    b += align_up(sizeof(cplx)*size_t(chunks)*A*d*d*W);                   // Ypart
    b += align_up(sizeof(cplx)*size_t(A)*d*d*W);                          // Bt
    b += ffk::expand_workspace_bytes(N, d);                               // compacted basis
    return b;
}

int max_chunks_for(int W, int A, int G, int d) {
    // upper bound of what accumulate_geometry may choose, for workspace sizing
    const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, g_forced_chunks);
    return geo.chunks;
}

double accumulate_flops(int W, int A, int G, int d) {
    // FMA-counted real flops the accumulate kernels EXECUTE per (segment, frequency)
    // (DESIGN.md section 3).
    // d = 4 (ctrl_pc.hip, round 4: real tile, folded operands): per operator 16 x (zz: 2 mul + 6 fma,
    // z = psi zz: 2 mul + 2 fma, Y: 16 fma) = 832; per group of <= 3 operators the tile: 13 entries x
    // 10 (x, addition theorem 3, reciprocal 5, product 1) + 62 (two sincos and psi) + 6 (e^{ib} T of
    // the fold, one element per lane = one per frequency) and 6 per operator (Bbar times that).
    if (d == 4 && ffk::pc_accumulate_supported(d, A))
        return (838.0*A + 198.0*((A + 2)/3))*double(G)*double(W);
    // d = 8 (ctrl_pcr.hip, round 4): per operator the first product real x complex 4 d^3 = 2048, psi P
    // 6 d^2 = 384, the second product complex 8 d^3 = 4096, the fold (one (m, n) per lane = per
    // frequency) 9 complex products = 54; per group of <= 3 operators the tile: 57 entries x 10 + 62.
    if (d == 8 && ffk::pcr_accumulate_supported(d, A))
        return (6582.0*A + 632.0*((A + 2)/3))*double(G)*double(W);
    // other d: contraction (2 d^3 MAC + d^2 mul) complex per (g, w, a) = 16 d^3 + 6 d^2 flops; tile:
    // ffk_math.h::phased_integral_aa, 18 flops per distinct entry (rotation 6, addition theorem 3,
    // x 1, reciprocal 4, products 3 + 1), d(d-1)+1 entries per (g, w); two sincos and the phase: 62.
    // (Rounds 1-3 reported a direct-evaluation model, 55 flops per entry + 26, which the kernels
    // have not executed since the round-3 generator.)
    const double per_gw = (16.0*d*d*d + 6.0*d*d)*A + 18.0*(d*(d - 1) + 1) + 62.0;
    return per_gw*double(G)*double(W);
}

}  // namespace

namespace ffk {
// for the translation units that implement part of the C ABI themselves (peer.hip)
void set_last_error(const char* message) { g_error = message; }
}  // namespace ffk

extern "C" {

const char* ffk_last_error(void) { return g_error.c_str(); }
int ffk_version(void) { return FFK_VERSION; }

int ffk_device_count(int* count) {
    FFK_REQUIRE(count, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(FFK_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    return FFK_OK;
}
int ffk_set_device(int device) {
    FFK_HIP(hipSetDevice(device));
    return FFK_OK;
}
int ffk_get_device(int* device) {
    FFK_REQUIRE(device, "device is NULL");
    FFK_HIP(hipGetDevice(device));
    return FFK_OK;
}
int ffk_device_info(char* name, int len, int* compute_units, size_t* global_mem_bytes) {
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FFK_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && len > 0) {
        std::snprintf(name, len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = prop.totalGlobalMem;
    return FFK_OK;
}

int ffk_malloc(void** dptr, size_t bytes) {
    FFK_REQUIRE(dptr, "dptr is NULL");
    FFK_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return FFK_OK;
}
int ffk_malloc_finegrained(void** dptr, size_t bytes) {
    FFK_REQUIRE(dptr, "dptr is NULL");
    // fine-grained (uncached at L2 for other agents' writes): flag words polled by a running kernel
    // while a peer GPU writes them must not be served from a stale L2 line
    // (no coarse-grained substitute: the caller falls back to the RCCL collective instead)
    FFK_HIP(hipExtMallocWithFlags(dptr, bytes ? bytes : 1, hipDeviceMallocFinegrained));
    return FFK_OK;
}
int ffk_free(void* dptr) {
    FFK_HIP(hipFree(dptr));
    return FFK_OK;
}
int ffk_memset(void* dptr, int value, size_t bytes, void* stream) {
    FFK_HIP(hipMemsetAsync(dptr, value, bytes, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    FFK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    FFK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    FFK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_stream_create(void** stream) {
    FFK_REQUIRE(stream, "stream is NULL");
    hipStream_t s;
    FFK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return FFK_OK;
}
int ffk_stream_destroy(void* stream) {
    FFK_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_stream_synchronize(void* stream) {
    FFK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_device_synchronize(void) {
    FFK_HIP(hipDeviceSynchronize());
    return FFK_OK;
}
int ffk_event_create(void** event) {
    FFK_REQUIRE(event, "event is NULL");
    hipEvent_t e;
    FFK_HIP(hipEventCreate(&e));
    *event = e;
    return FFK_OK;
}
int ffk_event_destroy(void* event) {
    FFK_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return FFK_OK;
}
int ffk_event_record(void* event, void* stream) {
    FFK_HIP(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
    return FFK_OK;
}
int ffk_stream_wait_event(void* stream, void* event) {
    FFK_REQUIRE(event, "event is NULL");
    FFK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return FFK_OK;
}
int ffk_event_synchronize(void* event) {
    FFK_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return FFK_OK;
}
int ffk_event_elapsed_ms(void* start, void* stop, float* ms) {
    FFK_REQUIRE(ms, "ms is NULL");
    FFK_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return FFK_OK;
}
int ffk_release_arena(void) {
    std::lock_guard<std::mutex> lock(g_arena.mu);
    if (g_arena.ptr) {
        FFK_HIP(hipDeviceSynchronize());
        FFK_HIP(hipFree(g_arena.ptr));
        g_arena.ptr = nullptr;
        g_arena.size = 0;
    }
    return FFK_OK;
}

int ffk_set_segment_chunks(int chunks) {
    FFK_REQUIRE(chunks >= 0, "chunks must be >= 0");
    g_forced_chunks = chunks;
    ++g_knob_epoch;
    return FFK_OK;
}
int ffk_set_accumulate_variant(int variant) {
    FFK_REQUIRE(variant >= 0 && variant <= 4, "variant must be 0..4");
    ++g_knob_epoch;
    ffk::set_use_wave_kernel(variant == 1);
    ffk::set_use_gsplit(variant != 2);
    ffk::set_mfma_policy(variant == 3 ? 1 : (variant == 4 ? 2 : 0));
    return FFK_OK;
}
int ffk_set_accumulate_events(void* start, void* stop) {
    ++g_knob_epoch;
    g_ev_start = static_cast<hipEvent_t>(start);
    g_ev_stop = static_cast<hipEvent_t>(stop);
    g_ev_gate = nullptr;
    return FFK_OK;
}
int ffk_set_accumulate_gate(void* event) {
    ++g_knob_epoch;
    g_ev_gate = static_cast<hipEvent_t>(event);
    return FFK_OK;
}
int ffk_get_stats(ffk_stats* out) {
    FFK_REQUIRE(out, "out is NULL");
    *out = g_stats;
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// diagonalize
// ---------------------------------------------------------------------------------------------
// workspace layout: [status: G ints][seg_prop: G d^2][Qloc: (G+1) d^2][totals / scan scratch]
size_t ffk_diagonalize_workspace_bytes(int G, int d) {
    if (G < 1 || !d_ok(d)) return 0;
    const size_t nch = (size_t(G) + ffk::front_chunk(d) - 1)/ffk::front_chunk(d);
    return align_up(sizeof(int)*size_t(G)) + align_up(sizeof(cplx)*size_t(G)*d*d) +
           align_up(sizeof(cplx)*size_t(G + 1)*d*d) +
           std::max(ffk::scan_workspace_bytes(G, d), align_up(sizeof(cplx)*nch*d*d));
}

namespace {
struct DiagWs {
    int* status;
    cplx* seg_prop;  // (G, d, d)
    cplx* qloc;      // (G+1, d, d) chunk-local prefix products
    void* small;     // scan scratch or chunk totals
};
DiagWs slice_diag_ws(void* workspace, size_t bytes, int G, int d) {
    Bump ws(workspace, bytes);
    DiagWs out;
    out.status = ws.take<int>(G);
    out.seg_prop = ws.take<cplx>(size_t(G)*d*d);
    out.qloc = ws.take<cplx>(size_t(G + 1)*d*d);
    out.small = ws.take<unsigned char>(1);
    return out;
}
}  // namespace

int ffk_diagonalize_dev(const double* hamiltonian, const double* dt, int G, int d, double* eigvals,
                        double* eigvecs, double* propagators, void* workspace,
                        size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(G >= 1, "need at least one segment, got G=%d", G);
    FFK_REQUIRE(hamiltonian && dt && eigvals && eigvecs && propagators && workspace, "NULL argument");
    FFK_REQUIRE(workspace_bytes >= ffk_diagonalize_workspace_bytes(G, d), "workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DiagWs w = slice_diag_ws(workspace, workspace_bytes, G, d);
    const cplx* H = reinterpret_cast<const cplx*>(hamiltonian);
    FFK_HIP(ffk::launch_eigh_expm(H, dt, G, d, eigvals, reinterpret_cast<cplx*>(eigvecs), w.seg_prop,
                                  w.status, s));
    if (ffk::use_fused_front(G, d)) {
        cplx* totals = static_cast<cplx*>(w.small);
        FFK_HIP(ffk::launch_scan_local(w.seg_prop, G, d, ffk::front_chunk(d), w.qloc, totals, s));
        FFK_HIP(ffk::launch_apply_prologue(w.qloc, totals, G, d, reinterpret_cast<cplx*>(propagators),
                                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                           nullptr, nullptr, nullptr, s));
    } else {
        FFK_HIP(ffk::launch_prefix_products(w.seg_prop, G, d, reinterpret_cast<cplx*>(propagators),
                                            w.small, s));
    }
    return FFK_OK;
}

int ffk_diagonalize(const double* hamiltonian, const double* dt, int G, int d, double* eigvals,
                    double* eigvecs, double* propagators) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(G >= 1, "need at least one segment, got G=%d", G);
    FFK_REQUIRE(hamiltonian && dt && eigvals && eigvecs && propagators, "NULL argument");
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nH = sizeof(cplx)*size_t(G)*d*d, nQ = sizeof(cplx)*size_t(G + 1)*d*d;
    const size_t wsb = ffk_diagonalize_workspace_bytes(G, d);
    const size_t total = 2*align_up(nH) + align_up(nQ) + 2*align_up(sizeof(double)*G*d) + wsb;
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dH = a.take<double>(size_t(G)*d*d*2);
    double* ddt = a.take<double>(G);
    double* dD = a.take<double>(size_t(G)*d);
    double* dV = a.take<double>(size_t(G)*d*d*2);
    double* dQ = a.take<double>(size_t(G + 1)*d*d*2);
    void* ws = a.take<unsigned char>(wsb);
    FFK_REQUIRE(ws, "internal: arena too small");
    FFK_HIP(hipMemcpyAsync(dH, hamiltonian, nH, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(ddt, dt, sizeof(double)*G, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_diagonalize_dev(dH, ddt, G, d, dD, dV, dQ, ws, wsb, nullptr)) return rc;
    std::vector<int> flags(G, 0);
    const int* dstatus = slice_diag_ws(ws, wsb, G, d).status;
    FFK_HIP(hipMemcpyAsync(eigvals, dD, sizeof(double)*G*d, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipMemcpyAsync(eigvecs, dV, nH, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipMemcpyAsync(propagators, dQ, nQ, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipMemcpyAsync(flags.data(), dstatus, sizeof(int)*G, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    int status = 0;
    for (int f : flags) status += f;
    if (status != 0)
        return fail(FFK_ENOCONV, "Jacobi eigensolver did not converge for %d segment(s)", status);
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// control matrix / noise operators
// ---------------------------------------------------------------------------------------------
size_t ffk_control_matrix_workspace_bytes(int W, int N, int A, int G, int d) {
    if (W < 1 || N < 1 || A < 1 || G < 1 || !d_ok(d)) return 0;
    return ctrl_ws_bytes(W, N, A, G, d, max_chunks_for(W, A, G, d));
}

int ffk_control_matrix_dev(const double* eigvals, const double* eigvecs, const double* propagators,
                           const double* omega, int W, const double* basis, int N,
                           const double* n_opers, int A, const double* n_coeffs, const double* dt,
                           const double* t, int G, int d, unsigned flags, double* control_matrix,
                           double* noise_operators, void* workspace, size_t workspace_bytes,
                           void* stream) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && n_opers && n_coeffs && dt && t && workspace,
                "NULL argument");
    FFK_REQUIRE(control_matrix || (flags & FFK_WANT_NOISE_OPERATORS), "no output requested");
    FFK_REQUIRE(!control_matrix || basis, "basis is NULL");
    FFK_REQUIRE(!(flags & FFK_WANT_NOISE_OPERATORS) || noise_operators, "noise_operators is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, g_forced_chunks);
    FFK_REQUIRE(workspace_bytes >= ctrl_ws_bytes(W, N, A, G, d, geo.chunks), "workspace too small");
    Bump ws(workspace, workspace_bytes);
    double* segtab = ws.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = ws.take<cplx>(size_t(G)*d*d);
    cplx* ops = ws.take<cplx>(size_t(G)*(1 + A)*d*d);
    cplx* Ypart = ws.take<cplx>(size_t(geo.chunks)*A*d*d*W);
    cplx* Bt = ws.take<cplx>(size_t(A)*d*d*W);
    void* ews = ws.take<unsigned char>(ffk::expand_workspace_bytes(N, d));
    FFK_REQUIRE(Bt && ews, "workspace too small");

    if (!(flags & FFK_INTERNAL_PROLOGUE_DONE))
        FFK_HIP(ffk::launch_prologue(eigvals, reinterpret_cast<const cplx*>(eigvecs),
                                     reinterpret_cast<const cplx*>(propagators),
                                     reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, G, d, A,
                                     segtab, Tc, ops, nullptr, nullptr, s));
    if (g_ev_start && g_ev_stop) {
        // `gate`: an event of ANOTHER stream (the previous pass's accumulate kernel) that this
        // stream waits for first, so that start..stop spans this kernel's execution and not its
        // wait for the other pass's blocks to retire
        if (g_ev_gate) FFK_HIP(hipStreamWaitEvent(s, g_ev_gate, 0));
        FFK_HIP(hipEventRecord(g_ev_start, s));
    }
    FFK_HIP(ffk::launch_accumulate(omega, W, segtab, ops, G, d, A, geo, Ypart, s));
    if (g_ev_start && g_ev_stop) FFK_HIP(hipEventRecord(g_ev_stop, s));
    const size_t slab = size_t(A)*d*d*W;
    const bool want_B = (flags & FFK_WANT_NOISE_OPERATORS) != 0;
    const cplx* Bsum = Ypart;
    bool compacted = (flags & FFK_INTERNAL_COMPACT_DONE) != 0;
    if (compacted && control_matrix && !want_B && g_fuse_F && ffk::expand_ff_supported(A, N)) {
        // only R and F are wanted and the basis lists are ready: chunk sum, expansion and F in one
        FFK_HIP(ffk::launch_expand_ff(Ypart, geo.chunks, slab, A, N, d, W,
                                      reinterpret_cast<cplx*>(control_matrix), g_fuse_F, ews, s));
        g_fuse_F_done = true;
    } else if (compacted && control_matrix && !want_B && geo.chunks > 1) {
        // only R is wanted and the basis lists are ready: expand straight from the chunk partials
        FFK_HIP(ffk::launch_expand_chunks(Ypart, geo.chunks, slab, A, N, d, W,
                                          reinterpret_cast<cplx*>(control_matrix), ews, s));
    } else {
    if (geo.chunks > 1) {
        if (control_matrix && !compacted) {
            FFK_HIP(ffk::launch_reduce_and_compact(Ypart, geo.chunks, slab, Bt,
                                                   reinterpret_cast<const cplx*>(basis), N, d, ews, s));
            compacted = true;
        } else {
            FFK_HIP(ffk::launch_reduce_chunks(Ypart, geo.chunks, slab, Bt, s));
        }
        Bsum = Bt;
    }
    if (control_matrix)
        FFK_HIP(ffk::launch_expand(Bsum, reinterpret_cast<const cplx*>(basis), A, N, d, W,
                                   reinterpret_cast<cplx*>(control_matrix), ews, compacted, s));
    if (want_B)
        FFK_HIP(ffk::launch_transpose_noise_ops(Bsum, A, d, W, reinterpret_cast<cplx*>(noise_operators), s));
    }

    g_stats.accumulate_flops = accumulate_flops(W, A, G, d);
    g_stats.accumulate_bytes = double(sizeof(cplx))*(double(geo.chunks)*slab) + 8.0*W +
                               double(sizeof(cplx))*G*(double(1 + A)*d*d);
    g_stats.chunks = geo.chunks;
    g_stats.grid_x = geo.mfma ? (W + 15)/16 : (W + 63)/64;
    g_stats.grid_y = geo.task_groups;
    g_stats.grid_z = geo.chunks;
    g_stats.block = geo.nwaves*geo.gsplit*64;
    g_stats.lds_bytes = geo.lds_bytes;
    return FFK_OK;
}

int ffk_control_matrix(const double* eigvals, const double* eigvecs, const double* propagators,
                       const double* omega, int W, const double* basis, int N,
                       const double* n_opers, int A, const double* n_coeffs, const double* dt,
                       const double* t, int G, int d, unsigned flags, double* control_matrix,
                       double* noise_operators) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && n_opers && n_coeffs && dt && t,
                "NULL argument");
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t wsb = ffk_control_matrix_workspace_bytes(W, N, A, G, d);
    const bool want_R = control_matrix != nullptr;
    const bool want_B = (flags & FFK_WANT_NOISE_OPERATORS) != 0;
    size_t total = wsb;
    total += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd);
    total += align_up(8*size_t(W)) + align_up(16*size_t(N)*dd) + align_up(16*size_t(A)*dd);
    total += align_up(8*size_t(A)*G) + align_up(8*size_t(G)) + align_up(8*size_t(G + 1));
    if (want_R) total += align_up(16*size_t(A)*N*W);
    if (want_B) total += align_up(16*size_t(A)*dd*W);
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(size_t(G)*d);
    double* dV = a.take<double>(2*size_t(G)*dd);
    double* dQ = a.take<double>(2*size_t(G + 1)*dd);
    double* dom = a.take<double>(W);
    double* dbasis = a.take<double>(2*size_t(N)*dd);
    double* dnop = a.take<double>(2*size_t(A)*dd);
    double* dnc = a.take<double>(size_t(A)*G);
    double* ddt = a.take<double>(G);
    double* dtt = a.take<double>(G + 1);
    double* dR = want_R ? a.take<double>(2*size_t(A)*N*W) : nullptr;
    double* dB = want_B ? a.take<double>(2*size_t(A)*dd*W) : nullptr;
    void* ws = a.take<unsigned char>(wsb);
    FFK_REQUIRE(ws, "internal: arena too small");
    auto h2d = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, nullptr);
    };
    FFK_HIP(h2d(dD, eigvals, 8*size_t(G)*d));
    FFK_HIP(h2d(dV, eigvecs, 16*size_t(G)*dd));
    FFK_HIP(h2d(dQ, propagators, 16*size_t(G + 1)*dd));
    FFK_HIP(h2d(dom, omega, 8*size_t(W)));
    if (basis) FFK_HIP(h2d(dbasis, basis, 16*size_t(N)*dd));
    FFK_HIP(h2d(dnop, n_opers, 16*size_t(A)*dd));
    FFK_HIP(h2d(dnc, n_coeffs, 8*size_t(A)*G));
    FFK_HIP(h2d(ddt, dt, 8*size_t(G)));
    FFK_HIP(h2d(dtt, t, 8*size_t(G + 1)));
    if (int rc = ffk_control_matrix_dev(dD, dV, dQ, dom, W, basis ? dbasis : nullptr, N, dnop, A, dnc,
                                        ddt, dtt, G, d, flags, dR, dB, ws, wsb, nullptr))
        return rc;
    if (want_R)
        FFK_HIP(hipMemcpyAsync(control_matrix, dR, 16*size_t(A)*N*W, hipMemcpyDeviceToHost, nullptr));
    if (want_B)
        FFK_HIP(hipMemcpyAsync(noise_operators, dB, 16*size_t(A)*dd*W, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

static int intermediates_impl(const double* eigvals, const double* eigvecs,
                              const double* propagators, const double* omega, int W,
                              const double* basis, int N, const double* n_opers, int A,
                              const double* n_coeffs, const double* dt, const double* t, int G,
                              int d, double* n_opers_transformed, double* eigvecs_propagated,
                              double* basis_transformed, double* phase_factors,
                              double* first_order_integral, double* control_matrix_step,
                              double* noise_operators_step) {
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && n_opers && n_coeffs && dt && t,
                "NULL argument");
    FFK_REQUIRE(basis || !(basis_transformed || control_matrix_step), "basis is NULL");
    FFK_REQUIRE(size_t(G)*A <= 65535, "G*A = %zu too large for the materialising variant", size_t(G)*A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    // inputs + operands + every requested product, all resident at once (HBM is 288 GB)
    size_t total = 0;
    total += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd);
    total += align_up(8*size_t(W)) + align_up(16*size_t(N)*dd) + align_up(16*size_t(A)*dd);
    total += align_up(8*size_t(A)*G) + align_up(8*size_t(G)) + align_up(8*size_t(G + 1));
    total += align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd) +
             align_up(16*size_t(G)*(1 + A)*dd);
    total += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(G)*dd);          // nt, ep
    if (basis_transformed) total += align_up(16*size_t(G)*N*dd);
    if (phase_factors) total += align_up(16*size_t(G)*W);
    if (first_order_integral) total += align_up(16*size_t(G)*W*dd);
    if (control_matrix_step || noise_operators_step) total += align_up(16*size_t(G)*A*dd*W);
    if (control_matrix_step)
        total += align_up(16*size_t(G)*A*N*W) + ffk::expand_workspace_bytes(N, d);
    if (noise_operators_step) total += align_up(16*size_t(G)*W*A*dd);
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(size_t(G)*d);
    cplx* dV = a.take<cplx>(size_t(G)*dd);
    cplx* dQ = a.take<cplx>(size_t(G + 1)*dd);
    double* dom = a.take<double>(W);
    cplx* dbasis = a.take<cplx>(size_t(N)*dd);
    cplx* dnop = a.take<cplx>(size_t(A)*dd);
    double* dnc = a.take<double>(size_t(A)*G);
    double* ddt = a.take<double>(G);
    double* dtt = a.take<double>(G + 1);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + A)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dbt = basis_transformed ? a.take<cplx>(size_t(G)*N*dd) : nullptr;
    cplx* dph = phase_factors ? a.take<cplx>(size_t(G)*W) : nullptr;
    cplx* dint = first_order_integral ? a.take<cplx>(size_t(G)*W*dd) : nullptr;
    cplx* Ypart = (control_matrix_step || noise_operators_step) ? a.take<cplx>(size_t(G)*A*dd*W) : nullptr;
    cplx* dnstep = noise_operators_step ? a.take<cplx>(size_t(G)*W*A*dd) : nullptr;
    cplx* dstep = control_matrix_step ? a.take<cplx>(size_t(G)*A*N*W) : nullptr;
    void* dews = control_matrix_step ? a.take<unsigned char>(ffk::expand_workspace_bytes(N, d)) : nullptr;
    FFK_REQUIRE(a.used <= g_arena.size, "internal: arena too small");
    auto h2d = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, nullptr);
    };
    auto d2h = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, nullptr);
    };
    FFK_HIP(h2d(dD, eigvals, 8*size_t(G)*d));
    FFK_HIP(h2d(dV, eigvecs, 16*size_t(G)*dd));
    FFK_HIP(h2d(dQ, propagators, 16*size_t(G + 1)*dd));
    FFK_HIP(h2d(dom, omega, 8*size_t(W)));
    if (basis) FFK_HIP(h2d(dbasis, basis, 16*size_t(N)*dd));
    FFK_HIP(h2d(dnop, n_opers, 16*size_t(A)*dd));
    FFK_HIP(h2d(dnc, n_coeffs, 8*size_t(A)*G));
    FFK_HIP(h2d(ddt, dt, 8*size_t(G)));
    FFK_HIP(h2d(dtt, t, 8*size_t(G + 1)));
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dnop, dnc, ddt, dtt, G, d, A, segtab, Tc, ops, dnt, dep, nullptr));
    if (dbt) FFK_HIP(ffk::launch_basis_transformed(Tc, dbasis, G, N, d, dbt, nullptr));
    FFK_HIP(ffk::launch_phase_and_integral(dom, W, segtab, G, d, dph, dint, nullptr));
    if (Ypart) {
        // one chunk per segment: Ypart[g] is that segment's Hilbert-space step
        ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, G);
        FFK_HIP(ffk::launch_accumulate(dom, W, segtab, ops, G, d, A, geo, Ypart, nullptr));
    }
    if (dstep)    // ... expanded in the basis
        FFK_HIP(ffk::launch_expand(Ypart, dbasis, G*A, N, d, W, dstep, dews, false, nullptr));
    if (dnstep)   // ... or re-laid out as (W, A, d, d) per segment
        for (int g = 0; g < G; ++g)
            FFK_HIP(ffk::launch_transpose_noise_ops(Ypart + size_t(g)*A*dd*W, A, d, W,
                                                    dnstep + size_t(g)*W*A*dd, nullptr));
    if (n_opers_transformed) FFK_HIP(d2h(n_opers_transformed, dnt, 16*size_t(A)*G*dd));
    if (eigvecs_propagated) FFK_HIP(d2h(eigvecs_propagated, dep, 16*size_t(G)*dd));
    if (basis_transformed) FFK_HIP(d2h(basis_transformed, dbt, 16*size_t(G)*N*dd));
    if (phase_factors) FFK_HIP(d2h(phase_factors, dph, 16*size_t(G)*W));
    if (first_order_integral) FFK_HIP(d2h(first_order_integral, dint, 16*size_t(G)*W*dd));
    if (control_matrix_step) FFK_HIP(d2h(control_matrix_step, dstep, 16*size_t(G)*A*N*W));
    if (noise_operators_step) FFK_HIP(d2h(noise_operators_step, dnstep, 16*size_t(G)*W*A*dd));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_control_matrix_intermediates(const double* eigvals, const double* eigvecs,
                                     const double* propagators, const double* omega, int W,
                                     const double* basis, int N, const double* n_opers, int A,
                                     const double* n_coeffs, const double* dt, const double* t, int G,
                                     int d, double* n_opers_transformed, double* eigvecs_propagated,
                                     double* basis_transformed, double* phase_factors,
                                     double* first_order_integral, double* control_matrix_step) {
    FFK_REQUIRE(basis, "NULL argument");
    return intermediates_impl(eigvals, eigvecs, propagators, omega, W, basis, N, n_opers, A, n_coeffs,
                              dt, t, G, d, n_opers_transformed, eigvecs_propagated, basis_transformed,
                              phase_factors, first_order_integral, control_matrix_step, nullptr);
}

int ffk_noise_operators_intermediates(const double* eigvals, const double* eigvecs,
                                      const double* propagators, const double* omega, int W,
                                      const double* n_opers, int A, const double* n_coeffs,
                                      const double* dt, const double* t, int G, int d,
                                      double* n_opers_transformed, double* phase_factors,
                                      double* first_order_integral, double* noise_operators_step) {
    return intermediates_impl(eigvals, eigvecs, propagators, omega, W, nullptr, 1, n_opers, A,
                              n_coeffs, dt, t, G, d, n_opers_transformed, nullptr, nullptr,
                              phase_factors, first_order_integral, nullptr, noise_operators_step);
}

// ---------------------------------------------------------------------------------------------
// concatenation rule
// ---------------------------------------------------------------------------------------------
size_t ffk_control_matrix_from_atomic_workspace_bytes(int G, int A, int N, int W) {
    if (G < 1 || A < 1 || N < 1 || W < 1) return 0;
    return ffk::from_atomic_workspace_bytes(G, A, N, W);
}

int ffk_control_matrix_from_atomic_dev(const double* phases, const double* control_matrix_atomic,
                                       const double* propagators_liouville, int l_is_complex,
                                       int G, int A, int N, int W, int which, double* out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis: G=%d A=%d N=%d W=%d", G, A, N, W);
    FFK_REQUIRE(control_matrix_atomic && out && workspace, "NULL argument");
    FFK_REQUIRE(G == 1 || (phases && propagators_liouville), "NULL argument");
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    FFK_REQUIRE(workspace_bytes >= ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W),
                "workspace too small");
    FFK_HIP(ffk::launch_from_atomic(reinterpret_cast<const cplx*>(phases),
                                    reinterpret_cast<const cplx*>(control_matrix_atomic), nullptr,
                                    propagators_liouville, l_is_complex, G, A, N, W, which,
                                    reinterpret_cast<cplx*>(out), workspace,
                                    static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_control_matrix_from_atomic(const double* phases, const double* control_matrix_atomic,
                                   const double* propagators_liouville, int l_is_complex, int G,
                                   int A, int N, int W, int which, double* out) {
    FFK_REQUIRE(G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis: G=%d A=%d N=%d W=%d", G, A, N, W);
    FFK_REQUIRE(control_matrix_atomic && out, "NULL argument");
    FFK_REQUIRE(G == 1 || (phases && propagators_liouville), "NULL argument");
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nP = 16*size_t(G > 1 ? G - 1 : 1)*W, nR = 16*size_t(G)*A*N*W;
    const size_t nL = (l_is_complex ? 16 : 8)*size_t(G > 1 ? G - 1 : 1)*N*N;
    const size_t nO = which ? nR : 16*size_t(A)*N*W;
    const size_t wsb = ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W);
    void* base;
    if (int rc = arena_reserve(align_up(nP) + align_up(nR) + align_up(nL) + align_up(nO) + wsb, &base))
        return rc;
    Bump a(base, g_arena.size);
    double* dP = a.take<double>(nP/8);
    double* dR = a.take<double>(nR/8);
    double* dL = a.take<double>(nL/8);
    double* dO = a.take<double>(nO/8);
    void* ws = a.take<unsigned char>(wsb);
    if (G > 1) {
        FFK_HIP(hipMemcpyAsync(dP, phases, 16*size_t(G - 1)*W, hipMemcpyHostToDevice, nullptr));
        FFK_HIP(hipMemcpyAsync(dL, propagators_liouville, (l_is_complex ? 16 : 8)*size_t(G - 1)*N*N,
                               hipMemcpyHostToDevice, nullptr));
    }
    FFK_HIP(hipMemcpyAsync(dR, control_matrix_atomic, nR, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_control_matrix_from_atomic_dev(dP, dR, dL, l_is_complex, G, A, N, W, which, dO, ws,
                                                    wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(out, dO, nO, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_control_matrix_from_atomic_indexed_dev(const double* total_phases,
                                               const double* control_matrix_table,
                                               const int32_t* index,
                                               const double* propagators_liouville,
                                               int l_is_complex, int T, int G, int A, int N, int W,
                                               int which, double* out, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(T >= 1 && G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis");
    FFK_REQUIRE(total_phases && control_matrix_table && index && out && workspace, "NULL argument");
    FFK_REQUIRE(G == 1 || propagators_liouville, "NULL argument");
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    FFK_REQUIRE(workspace_bytes >= ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W),
                "workspace too small");
    FFK_HIP(ffk::launch_from_atomic(reinterpret_cast<const cplx*>(total_phases),
                                    reinterpret_cast<const cplx*>(control_matrix_table), index,
                                    propagators_liouville, l_is_complex, G, A, N, W, which,
                                    reinterpret_cast<cplx*>(out), workspace,
                                    static_cast<hipStream_t>(stream), nullptr, nullptr, T));
    return FFK_OK;
}

int ffk_control_matrix_from_atomic_indexed(const double* total_phases,
                                           const double* control_matrix_table,
                                           const int32_t* index,
                                           const double* propagators_liouville, int l_is_complex,
                                           int T, int G, int A, int N, int W, int which,
                                           double* out) {
    FFK_REQUIRE(T >= 1 && G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis");
    FFK_REQUIRE(total_phases && control_matrix_table && index && out, "NULL argument");
    FFK_REQUIRE(G == 1 || propagators_liouville, "NULL argument");
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    for (int g = 0; g < G; ++g)
        FFK_REQUIRE(index[g] >= 0 && index[g] < T, "index[%d] = %d outside [0, %d)", g, index[g], T);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nP = 16*size_t(T)*W, nR = 16*size_t(T)*A*N*W, nI = 4*size_t(G);
    const size_t nL = (l_is_complex ? 16 : 8)*size_t(G > 1 ? G - 1 : 1)*N*N;
    const size_t nO = which ? 16*size_t(G)*A*N*W : 16*size_t(A)*N*W;
    const size_t wsb = ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W);
    void* base;
    if (int rc = arena_reserve(align_up(nP) + align_up(nR) + align_up(nI) + align_up(nL) + align_up(nO) + wsb,
                               &base))
        return rc;
    Bump a(base, g_arena.size);
    double* dP = a.take<double>(nP/8);
    double* dR = a.take<double>(nR/8);
    int32_t* dI = a.take<int32_t>(G);
    double* dL = a.take<double>(nL/8);
    double* dO = a.take<double>(nO/8);
    void* ws = a.take<unsigned char>(wsb);
    FFK_HIP(hipMemcpyAsync(dP, total_phases, nP, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dR, control_matrix_table, nR, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dI, index, nI, hipMemcpyHostToDevice, nullptr));
    if (G > 1)
        FFK_HIP(hipMemcpyAsync(dL, propagators_liouville, (l_is_complex ? 16 : 8)*size_t(G - 1)*N*N,
                               hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_control_matrix_from_atomic_indexed_dev(dP, dR, dI, dL, l_is_complex, T, G, A, N, W,
                                                            which, dO, ws, wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(out, dO, nO, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

}  // extern "C"

namespace {
// phases[k, w] = exp(i omega[w] tau[k]) (pulse_sequence.py:1156, util.cexp)
__global__ void total_phases_kernel(const double* __restrict__ omega, const double* __restrict__ tau, int T,
                                    int W, cplx* __restrict__ phases) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (w >= W || k >= T) return;
    phases[static_cast<size_t>(k)*W + w] = ffk::cexp(omega[w]*tau[k]);
}
// P[g] = table[index[g]]: the per-position total propagators of a sequence drawn from T pulses
__global__ void gather_propagators_kernel(const cplx* __restrict__ table, const int32_t* __restrict__ index,
                                          int G, int dd, cplx* __restrict__ P) {
    const size_t e = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
    if (e >= static_cast<size_t>(G)*dd) return;
    P[e] = table[static_cast<size_t>(index[e / dd])*dd + e % dd];
}
}  // namespace

// ---- pulse_sequence.concatenate for a sequence drawn from T distinct pulses, in one call --------
// (pulse_sequence.py:1812-1840: the cumulative propagators, their Liouville representations, the
// cumulative phase factors and the concatenation rule).  total_propagators (T, d, d) c128,
// total_phases (T, W) c128, control_matrix_table (T, A, N, W) c128, index (G,) int32.  On the
// device: gather -> prefix products (scan.hip) -> Liouville representation of the first G - 1
// (liouville.hip) -> the gather-from-table rule (atomic.hip); nothing but the tables goes in and
// the results come out.  Outputs: control matrix ((A, N, W), or (G, A, N, W) for which = 1), the
// sequence's total propagator (d, d) and -- if not NULL -- the (G - 1, N, N) Liouville propagators
// (f64 for a Hermitian basis, else c128), and -- if not NULL, which = 0 -- the fidelity filter
// function (A, A, W) of the summed control matrix.
namespace {

// temporaries of one sequence run, in arena order
size_t sequence_scratch_bytes(int G, int d, int A, int N, int W, int which, bool hermitian, bool want_F) {
    const size_t dd = size_t(d)*d;
    const int nl = G > 1 ? G - 1 : 1;
    return align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd) +
           align_up((hermitian ? 8 : 16)*size_t(nl)*N*N) +
           align_up(which ? 16*size_t(G)*A*N*W : 16*size_t(A)*N*W) +
           align_up(ffk::scan_workspace_bytes(G, d)) + align_up(ffk::liouville_workspace_bytes(nl, d, N)) +
           align_up(ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W)) +
           (want_F ? align_up(16*size_t(A)*A*W) : 0);
}

// gather -> prefix products -> Liouville representation -> table rule (-> F) on `s`, all operands
// already on the device; results to the host pointers (asynchronously: the caller synchronises)
int sequence_on_device(const double* dU, const double* dP, const double* dR, const int32_t* dI,
                       const double* dB, int hermitian_basis, int T, int G, int d, int A, int N, int W,
                       int which, Bump& a, double* control_matrix, double* total_propagator,
                       double* propagators_liouville, double* filter_function, hipStream_t s,
                       double* resident_R = nullptr, double* resident_F = nullptr,
                       const cplx* const* dRtab = nullptr, const double* dTau = nullptr,
                       const double* dOmega = nullptr, double* omega_copy = nullptr) {
    // dRtab: device array of T pointers to the distinct control matrices (dR is then unused);
    // dTau / dOmega: durations (T) and grid (W) on the device -- the total phases are then formed
    // here (dP is the buffer they go to), by the fused front launch where it applies
    const size_t dd = size_t(d)*d;
    const int l_is_complex = hermitian_basis ? 0 : 1;
    const int nl = G > 1 ? G - 1 : 1;
    const size_t nL = (l_is_complex ? 16 : 8)*size_t(nl)*N*N;
    const size_t nO = which ? 16*size_t(G)*A*N*W : 16*size_t(A)*N*W;
    const size_t sws = ffk::scan_workspace_bytes(G, d), lws = ffk::liouville_workspace_bytes(nl, d, N);
    const size_t aws = ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W);
    const size_t nF = filter_function ? 16*size_t(A)*A*W : 0;
    cplx* dSeq = a.take<cplx>(size_t(G)*dd);
    cplx* dQ = a.take<cplx>(size_t(G + 1)*dd);
    double* dL = a.take<double>(nL/8);
    double* dO = a.take<double>(nO/8);
    void* wscan = a.take<unsigned char>(sws);
    void* wliou = a.take<unsigned char>(lws);
    void* watom = a.take<unsigned char>(aws);
    double* dF = nF ? a.take<double>(nF/8) : nullptr;
    FFK_REQUIRE(watom && (!nF || dF), "workspace too small");
    if (resident_R) dO = resident_R;          // results that stay in a handle's device block
    if (resident_F) dF = resident_F;
    if (dTau && ffk::sequence_front_supported(d, G, N)) {
        // gather + running products + Liouville representations + total phases (+ grid copy): one launch
        FFK_HIP(ffk::launch_sequence_front(reinterpret_cast<const cplx*>(dU), dI, G, d,
                                           reinterpret_cast<const cplx*>(dB), N, l_is_complex, dQ, dL, dTau,
                                           dOmega, T, W, reinterpret_cast<cplx*>(const_cast<double*>(dP)),
                                           omega_copy, s));
    } else {
        if (dTau) {
            hipLaunchKernelGGL(total_phases_kernel, dim3((W + 255)/256, T), dim3(256), 0, s, dOmega, dTau, T,
                               W, reinterpret_cast<cplx*>(const_cast<double*>(dP)));
            FFK_HIP(hipGetLastError());
            if (omega_copy) FFK_HIP(hipMemcpyAsync(omega_copy, dOmega, 8*size_t(W), hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(gather_propagators_kernel, dim3(static_cast<unsigned>((size_t(G)*dd + 255)/256)),
                           dim3(256), 0, s, reinterpret_cast<const cplx*>(dU), dI, G, d*d, dSeq);
        FFK_HIP(hipGetLastError());
        FFK_HIP(ffk::launch_prefix_products(dSeq, G, d, dQ, wscan, s));
        if (G > 1)
            FFK_HIP(ffk::launch_liouville(dQ + dd, G - 1, d, reinterpret_cast<const cplx*>(dB), N,
                                          hermitian_basis, dL, wliou, s));
    }
    // the table rule, the slab reduction and (which = 0) the filter function of the sum
    FFK_HIP(ffk::launch_from_atomic(reinterpret_cast<const cplx*>(dP), reinterpret_cast<const cplx*>(dR), dI,
                                    dL, l_is_complex, G, A, N, W, which, reinterpret_cast<cplx*>(dO), watom,
                                    s, dRtab, reinterpret_cast<cplx*>(dF), T));
    if (dF) FFK_HIP(hipMemcpyAsync(filter_function, dF, nF, hipMemcpyDeviceToHost, s));
    if (control_matrix) FFK_HIP(hipMemcpyAsync(control_matrix, dO, nO, hipMemcpyDeviceToHost, s));
    FFK_HIP(hipMemcpyAsync(total_propagator, dQ + size_t(G)*dd, 16*dd, hipMemcpyDeviceToHost, s));
    if (propagators_liouville && G > 1)
        FFK_HIP(hipMemcpyAsync(propagators_liouville, dL, (l_is_complex ? 16 : 8)*size_t(G - 1)*N*N,
                               hipMemcpyDeviceToHost, s));
    return FFK_OK;
}

}  // namespace

extern "C" {

int ffk_concatenate_sequence(const double* total_propagators, const double* total_phases,
                             const double* control_matrix_table, const int32_t* index,
                             const double* basis, int hermitian_basis, int T, int G, int d, int A,
                             int N, int W, int which, double* control_matrix,
                             double* total_propagator, double* propagators_liouville,
                             double* filter_function) {
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(T >= 1 && G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis");
    FFK_REQUIRE(!filter_function || which == 0, "the filter function needs the summed control matrix");
    FFK_REQUIRE(total_propagators && total_phases && control_matrix_table && index && basis &&
                control_matrix && total_propagator, "NULL argument");
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    for (int g = 0; g < G; ++g)
        FFK_REQUIRE(index[g] >= 0 && index[g] < T, "index[%d] = %d outside [0, %d)", g, index[g], T);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nU = 16*size_t(T)*dd, nP = 16*size_t(T)*W, nR = 16*size_t(T)*A*N*W, nI = 4*size_t(G);
    const size_t nB = 16*size_t(N)*dd;
    void* base;
    if (int rc = arena_reserve(align_up(nU) + align_up(nP) + align_up(nR) + align_up(nI) + align_up(nB) +
                               sequence_scratch_bytes(G, d, A, N, W, which, hermitian_basis != 0,
                                                      filter_function != nullptr), &base))
        return rc;
    Bump a(base, g_arena.size);
    double* dU = a.take<double>(nU/8);
    double* dP = a.take<double>(nP/8);
    double* dR = a.take<double>(nR/8);
    int32_t* dI = a.take<int32_t>(G);
    double* dB = a.take<double>(nB/8);
    FFK_HIP(hipMemcpyAsync(dU, total_propagators, nU, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dI, index, nI, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dB, basis, nB, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dP, total_phases, nP, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dR, control_matrix_table, nR, hipMemcpyHostToDevice, nullptr));
    if (int rc = sequence_on_device(dU, dP, dR, dI, dB, hermitian_basis, T, G, d, A, N, W, which, a,
                                    control_matrix, total_propagator, propagators_liouville,
                                    filter_function, nullptr))
        return rc;
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

size_t ffk_control_matrix_periodic_workspace_bytes(int A, int N, int W) {
    if (A < 1 || N < 1 || W < 1) return 0;
    return ffk::periodic_workspace_bytes(A, N, W);
}

int ffk_control_matrix_periodic_dev(const double* phases, const double* control_matrix,
                                    const double* total_propagator_liouville, int l_is_complex,
                                    int repeats, int A, int N, int W, double* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    FFK_REQUIRE(repeats >= 1, "repeats = %d: need at least one period", repeats);
    FFK_REQUIRE(phases && control_matrix && total_propagator_liouville && out && workspace, "NULL argument");
    FFK_REQUIRE(out != control_matrix, "out must not alias control_matrix");
    FFK_REQUIRE(workspace_bytes >= ffk_control_matrix_periodic_workspace_bytes(A, N, W), "workspace too small");
    FFK_HIP(ffk::launch_periodic(reinterpret_cast<const cplx*>(phases),
                                 reinterpret_cast<const cplx*>(control_matrix),
                                 total_propagator_liouville, l_is_complex, repeats, A, N, W,
                                 reinterpret_cast<cplx*>(out), workspace, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_control_matrix_periodic(const double* phases, const double* control_matrix,
                                const double* total_propagator_liouville, int l_is_complex, int repeats,
                                int A, int N, int W, double* out) {
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    FFK_REQUIRE(repeats >= 1, "repeats = %d: need at least one period", repeats);
    FFK_REQUIRE(phases && control_matrix && total_propagator_liouville && out, "NULL argument");
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nP = 16*size_t(W), nR = 16*size_t(A)*N*W, nL = (l_is_complex ? 16 : 8)*size_t(N)*N;
    const size_t wsb = ffk_control_matrix_periodic_workspace_bytes(A, N, W);
    void* base;
    if (int rc = arena_reserve(align_up(nP) + 2*align_up(nR) + align_up(nL) + wsb, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dP = a.take<double>(nP/8);
    double* dR = a.take<double>(nR/8);
    double* dL = a.take<double>(nL/8);
    double* dO = a.take<double>(nR/8);
    void* ws = a.take<unsigned char>(wsb);
    FFK_HIP(hipMemcpyAsync(dP, phases, nP, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dR, control_matrix, nR, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dL, total_propagator_liouville, nL, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_control_matrix_periodic_dev(dP, dR, dL, l_is_complex, repeats, A, N, W, dO, ws, wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(out, dO, nR, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_noise_operators_from_atomic(const double* phases, const double* noise_operators_atomic,
                                    const double* propagators, int G, int W, int A, int d,
                                    double* noise_operators) {
    FFK_REQUIRE(noise_operators_atomic && noise_operators, "NULL argument");
    FFK_REQUIRE(G == 1 || (phases && propagators), "NULL argument");
    FFK_REQUIRE(G >= 1 && W >= 1 && A >= 1, "empty axis: G=%d W=%d A=%d", G, W, A);
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nph = 16*size_t(G > 1 ? G - 1 : 1)*W, nat = 16*size_t(G)*W*A*dd;
    const size_t npr = 16*size_t(G > 1 ? G - 1 : 1)*dd, nout = 16*size_t(W)*A*dd;
    void* base;
    if (int rc = arena_reserve(align_up(nph) + align_up(nat) + align_up(npr) + align_up(nout), &base))
        return rc;
    Bump a(base, g_arena.size);
    cplx* dph = a.take<cplx>(nph/16);
    cplx* dat = a.take<cplx>(nat/16);
    cplx* dpr = a.take<cplx>(npr/16);
    cplx* dout = a.take<cplx>(nout/16);
    if (G > 1) {
        FFK_HIP(hipMemcpyAsync(dph, phases, 16*size_t(G - 1)*W, hipMemcpyHostToDevice, nullptr));
        FFK_HIP(hipMemcpyAsync(dpr, propagators, 16*size_t(G - 1)*dd, hipMemcpyHostToDevice, nullptr));
    }
    FFK_HIP(hipMemcpyAsync(dat, noise_operators_atomic, nat, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(ffk::launch_noise_ops_from_atomic(dph, dat, dpr, G, W, A, d, dout, nullptr));
    FFK_HIP(hipMemcpyAsync(noise_operators, dout, nout, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// filter function
// ---------------------------------------------------------------------------------------------
int ffk_filter_function_dev(const double* control_matrix, int A, int N, int W, int which,
                            double* filter_function, void* stream) {
    FFK_REQUIRE(control_matrix && filter_function, "NULL argument");
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    FFK_REQUIRE(which == FFK_FF_FIDELITY || which == FFK_FF_GENERALIZED, "invalid which=%d", which);
    FFK_HIP(ffk::launch_filter_function(reinterpret_cast<const cplx*>(control_matrix), A, N, W, which,
                                        reinterpret_cast<cplx*>(filter_function),
                                        static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_filter_function_weighted_dev(const double* control_matrix, int A, int N, int W,
                                     const double* weights, double scale, double* filter_function,
                                     void* stream) {
    FFK_REQUIRE(control_matrix && weights && filter_function, "NULL argument");
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    FFK_HIP(ffk::launch_filter_function_weighted(reinterpret_cast<const cplx*>(control_matrix), A, N, W,
                                                 reinterpret_cast<const cplx*>(weights), scale,
                                                 reinterpret_cast<cplx*>(filter_function),
                                                 static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_filter_function_weighted(const double* control_matrix, int A, int N, int W,
                                 const double* weights, double scale, double* filter_function) {
    FFK_REQUIRE(control_matrix && weights && filter_function, "NULL argument");
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nR = 16*size_t(A)*N*W, nM = 16*size_t(N)*N, nF = 16*size_t(A)*A*W;
    void* base;
    if (int rc = arena_reserve(align_up(nR) + align_up(nM) + align_up(nF), &base)) return rc;
    Bump a(base, g_arena.size);
    double* dR = a.take<double>(nR/8);
    double* dM = a.take<double>(nM/8);
    double* dF = a.take<double>(nF/8);
    FFK_HIP(hipMemcpyAsync(dR, control_matrix, nR, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dM, weights, nM, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_filter_function_weighted_dev(dR, A, N, W, dM, scale, dF, nullptr)) return rc;
    FFK_HIP(hipMemcpyAsync(filter_function, dF, nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_filter_function(const double* control_matrix, int A, int N, int W, int which,
                        double* filter_function) {
    FFK_REQUIRE(control_matrix && filter_function, "NULL argument");
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1, "empty axis: A=%d N=%d W=%d", A, N, W);
    FFK_REQUIRE(which == FFK_FF_FIDELITY || which == FFK_FF_GENERALIZED, "invalid which=%d", which);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nR = 16*size_t(A)*N*W;
    const size_t nF = which == FFK_FF_FIDELITY ? 16*size_t(A)*A*W : 16*size_t(A)*A*N*N*W;
    void* base;
    if (int rc = arena_reserve(align_up(nR) + align_up(nF), &base)) return rc;
    Bump a(base, g_arena.size);
    double* dR = a.take<double>(nR/8);
    double* dF = a.take<double>(nF/8);
    FFK_HIP(hipMemcpyAsync(dR, control_matrix, nR, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_filter_function_dev(dR, A, N, W, which, dF, nullptr)) return rc;
    FFK_HIP(hipMemcpyAsync(filter_function, dF, nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// infidelity
// ---------------------------------------------------------------------------------------------
size_t ffk_infidelity_workspace_bytes(int W, int n_idx, int s_ndim) {
    if (W < 1 || n_idx < 1 || s_ndim < 1 || s_ndim > 3) return 0;
    return ffk::infidelity_workspace_bytes(W, n_idx, s_ndim);
}

int ffk_infidelity_dev(const double* filter_function, int A, int W, const double* spectrum,
                       int s_ndim, const double* omega, const int32_t* idx, int n_idx, int d,
                       double* infid, void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(filter_function && spectrum && omega && idx && infid && workspace, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(A >= 1 && W >= 1 && n_idx >= 1 && d >= 1, "empty axis");
    FFK_REQUIRE(workspace_bytes >= ffk_infidelity_workspace_bytes(W, n_idx, s_ndim), "workspace too small");
    FFK_HIP(ffk::launch_infidelity(reinterpret_cast<const cplx*>(filter_function), A, W,
                                   reinterpret_cast<const cplx*>(spectrum), s_ndim, omega, idx, n_idx,
                                   d, 0, infid, workspace, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_infidelity_sharded_dev(const double* filter_function_shards, int n_shards, int shard_width,
                               int A, const double* spectrum, int s_ndim, const double* omega,
                               const int32_t* idx, int n_idx, int d, double* infid, void* stream) {
    FFK_REQUIRE(filter_function_shards && spectrum && omega && idx && infid, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(n_shards >= 1 && shard_width >= 1 && A >= 1 && n_idx >= 1 && d >= 1, "empty axis");
    FFK_HIP(ffk::launch_infidelity(reinterpret_cast<const cplx*>(filter_function_shards), A,
                                   n_shards*shard_width, reinterpret_cast<const cplx*>(spectrum),
                                   s_ndim, omega, idx, n_idx, d, shard_width, infid, nullptr,
                                   static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_infidelity(const double* filter_function, int A, int W, const double* spectrum, int s_ndim,
                   const double* omega, const int32_t* idx, int n_idx, int d, double* infid) {
    FFK_REQUIRE(filter_function && spectrum && omega && idx && infid, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(A >= 1 && W >= 1 && n_idx >= 1 && d >= 1, "empty axis");
    for (int i = 0; i < n_idx; ++i)
        FFK_REQUIRE(idx[i] >= 0 && idx[i] < A, "noise operator index %d out of range [0, %d)", idx[i], A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nout = s_ndim == 3 ? size_t(n_idx)*n_idx : n_idx;
    const size_t nS = 16*size_t(W)*(s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : size_t(n_idx)*n_idx));
    const size_t nF = 16*size_t(A)*A*W;
    const size_t wsb = ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
    void* base;
    if (int rc = arena_reserve(align_up(nF) + align_up(nS) + align_up(8*size_t(W)) + align_up(4*size_t(n_idx)) +
                                   align_up(8*nout) + wsb, &base))
        return rc;
    Bump a(base, g_arena.size);
    double* dF = a.take<double>(nF/8);
    double* dS = a.take<double>(nS/8);
    double* dom = a.take<double>(W);
    int32_t* didx = a.take<int32_t>(n_idx);
    double* dout = a.take<double>(nout);
    void* ws = a.take<unsigned char>(wsb);
    FFK_HIP(hipMemcpyAsync(dF, filter_function, nF, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dS, spectrum, nS, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dom, omega, 8*size_t(W), hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(didx, idx, 4*size_t(n_idx), hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_infidelity_dev(dF, A, W, dS, s_ndim, dom, didx, n_idx, d, dout, ws, wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(infid, dout, 8*nout, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// Decay amplitudes, cumulant function
// ---------------------------------------------------------------------------------------------
size_t ffk_decay_amplitudes_workspace_bytes(int n_pulses, int N, int W, int n_idx, int s_ndim) {
    if (n_pulses < 1 || N < 1 || W < 1 || n_idx < 1 || s_ndim < 1 || s_ndim > 3) return 0;
    return ffk::decay_amplitudes_workspace_bytes(n_pulses, N, W, n_idx, s_ndim);
}

int ffk_decay_amplitudes_shard_dev(const double* control_matrix, int n_pulses, int A, int N,
                                   int W_block, const double* spectrum, int s_ndim,
                                   const double* omega, int W, int w_offset, const int32_t* idx,
                                   int n_idx, double* decay_amplitudes, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(control_matrix && spectrum && omega && idx && decay_amplitudes && workspace,
                "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(n_pulses >= 1 && A >= 1 && N >= 1 && W_block >= 1 && n_idx >= 1, "empty axis");
    FFK_REQUIRE(w_offset >= 0 && w_offset + W_block <= W, "frequency block [%d, %d) outside [0, %d)",
                w_offset, w_offset + W_block, W);
    FFK_REQUIRE(workspace_bytes >= ffk_decay_amplitudes_workspace_bytes(n_pulses, N, W_block, n_idx, s_ndim),
                "workspace too small");
    FFK_HIP(ffk::launch_decay_amplitudes(reinterpret_cast<const cplx*>(control_matrix), n_pulses, A,
                                         N, W_block, reinterpret_cast<const cplx*>(spectrum), s_ndim,
                                         omega, W, w_offset, idx, n_idx, decay_amplitudes, workspace,
                                         static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_decay_amplitudes_dev(const double* control_matrix, int n_pulses, int A, int N, int W,
                             const double* spectrum, int s_ndim, const double* omega,
                             const int32_t* idx, int n_idx, double* decay_amplitudes,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return ffk_decay_amplitudes_shard_dev(control_matrix, n_pulses, A, N, W, spectrum, s_ndim, omega,
                                          W, 0, idx, n_idx, decay_amplitudes, workspace,
                                          workspace_bytes, stream);
}

int ffk_decay_amplitudes(const double* control_matrix, int n_pulses, int A, int N, int W,
                         const double* spectrum, int s_ndim, const double* omega,
                         const int32_t* idx, int n_idx, double* decay_amplitudes) {
    FFK_REQUIRE(control_matrix && spectrum && omega && idx && decay_amplitudes, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(n_pulses >= 1 && A >= 1 && N >= 1 && W >= 1 && n_idx >= 1, "empty axis");
    for (int i = 0; i < n_idx; ++i)
        FFK_REQUIRE(idx[i] >= 0 && idx[i] < A, "noise operator index %d out of range [0, %d)", idx[i], A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nR = 16*size_t(n_pulses)*A*N*W;
    const size_t nS = 16*size_t(W)*(s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : size_t(n_idx)*n_idx));
    const size_t nout = size_t(n_pulses)*n_pulses*n_idx*(s_ndim == 3 ? n_idx : 1)*N*N;
    const size_t wsb = ffk_decay_amplitudes_workspace_bytes(n_pulses, N, W, n_idx, s_ndim);
    void* base;
    if (int rc = arena_reserve(align_up(nR) + align_up(nS) + align_up(8*size_t(W)) +
                                   align_up(4*size_t(n_idx)) + align_up(8*nout) + wsb, &base))
        return rc;
    Bump a(base, g_arena.size);
    double* dR = a.take<double>(nR/8);
    double* dS = a.take<double>(nS/8);
    double* dom = a.take<double>(W);
    int32_t* didx = a.take<int32_t>(n_idx);
    double* dout = a.take<double>(nout);
    void* ws = a.take<unsigned char>(wsb);
    FFK_HIP(hipMemcpyAsync(dR, control_matrix, nR, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dS, spectrum, nS, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dom, omega, 8*size_t(W), hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(didx, idx, 4*size_t(n_idx), hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_decay_amplitudes_dev(dR, n_pulses, A, N, W, dS, s_ndim, dom, didx, n_idx, dout,
                                          ws, wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(decay_amplitudes, dout, 8*nout, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

size_t ffk_cumulant_function_workspace_bytes(int batch, int N, int d) {
    if (batch < 1 || N < 1 || !d_templated_ok(d)) return 0;
    return ffk::cumulant_workspace_bytes(batch, N, d);
}

int ffk_cumulant_function_dev(const double* decay_amplitudes, int batch, int N, int d,
                              const double* basis, int single_qubit, double* cumulant_function,
                              void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(decay_amplitudes && basis && cumulant_function, "NULL argument");
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis");
    FFK_REQUIRE(d_templated_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(!single_qubit || (d == 2 && N == 4), "single-qubit expression needs d = 2, N = 4");
    if (!single_qubit) {
        FFK_REQUIRE(batch <= 65535, "batch %d too large", batch);
        FFK_REQUIRE(workspace && workspace_bytes >= ffk_cumulant_function_workspace_bytes(batch, N, d),
                    "workspace too small");
    }
    FFK_HIP(ffk::launch_cumulant_function(decay_amplitudes, batch, N, d,
                                          reinterpret_cast<const cplx*>(basis), single_qubit,
                                          cumulant_function, workspace,
                                          static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_cumulant_function(const double* decay_amplitudes, int batch, int N, int d,
                          const double* basis, int single_qubit, double* cumulant_function) {
    FFK_REQUIRE(decay_amplitudes && basis && cumulant_function, "NULL argument");
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis");
    FFK_REQUIRE(d_templated_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D_TEMPLATED);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nG = 8*size_t(batch)*N*N;
    const size_t nB = 16*size_t(N)*d*d;
    const size_t wsb = single_qubit ? 0 : ffk_cumulant_function_workspace_bytes(batch, N, d);
    void* base;
    if (int rc = arena_reserve(2*align_up(nG) + align_up(nB) + wsb + 256, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dG = a.take<double>(nG/8);
    double* dK = a.take<double>(nG/8);
    double* dB = a.take<double>(nB/8);
    void* ws = a.take<unsigned char>(wsb + 16);
    FFK_HIP(hipMemcpyAsync(dG, decay_amplitudes, nG, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dB, basis, nB, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_cumulant_function_dev(dG, batch, N, d, dB, single_qubit, dK, ws, wsb, nullptr))
        return rc;
    FFK_HIP(hipMemcpyAsync(cumulant_function, dK, nG, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// second order: filter function, frequency shifts, cumulant-function contribution
// ---------------------------------------------------------------------------------------------
static int second_order_impl(const double* eigvals, const double* eigvecs,
                             const double* propagators, const double* omega, int W,
                             const double* basis, int N, const double* n_opers, int A,
                             const double* n_coeffs, const double* dt, const double* t, int G, int d,
                             double* filter_function_2, const double* spectrum, int s_ndim,
                             const int32_t* idx, int n_idx, double* frequency_shifts) {
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && basis && n_opers && n_coeffs && dt && t,
                "NULL argument");
    FFK_REQUIRE(filter_function_2 || frequency_shifts, "no output requested");
    FFK_REQUIRE(size_t(A)*N <= 65535, "A*N = %zu too large", size_t(A)*N);
    size_t nS = 0, nout = 0;
    int srows = 0;
    if (frequency_shifts) {
        FFK_REQUIRE(spectrum && idx, "NULL argument");
        FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
        FFK_REQUIRE(n_idx >= 1, "empty axis");
        for (int i = 0; i < n_idx; ++i)
            FFK_REQUIRE(idx[i] >= 0 && idx[i] < A, "noise operator index %d out of range [0, %d)", idx[i], A);
        srows = s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : n_idx*n_idx);
        nS = 16*size_t(W)*srows;
        nout = size_t(n_idx)*(s_ndim == 3 ? n_idx : 1)*N*N;
    }
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nF = size_t(A)*A*N*N*W;
    const size_t wsb = ffk::second_order_workspace_bytes(G, A, N, d);
    size_t total = 0;
    total += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd);
    total += align_up(8*size_t(W)) + align_up(16*size_t(N)*dd) + align_up(16*size_t(A)*dd);
    total += align_up(8*size_t(A)*G) + align_up(8*size_t(G)) + align_up(8*size_t(G + 1));
    total += align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd) +
             align_up(16*size_t(G)*(1 + A)*dd);
    total += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G)*N*dd);
    total += wsb + align_up(16*nF);
    total += 2*align_up(nS) + align_up(4*size_t(n_idx > 0 ? n_idx : 1)) + align_up(8*nout);
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(size_t(G)*d);
    cplx* dV = a.take<cplx>(size_t(G)*dd);
    cplx* dQ = a.take<cplx>(size_t(G + 1)*dd);
    double* dom = a.take<double>(W);
    cplx* dbasis = a.take<cplx>(size_t(N)*dd);
    cplx* dnop = a.take<cplx>(size_t(A)*dd);
    double* dnc = a.take<double>(size_t(A)*G);
    double* ddt = a.take<double>(G);
    double* dtt = a.take<double>(G + 1);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + A)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dbt = a.take<cplx>(size_t(G)*N*dd);
    void* ws = a.take<unsigned char>(wsb);
    cplx* dF = a.take<cplx>(nF);
    cplx* dS = frequency_shifts ? a.take<cplx>(nS/16) : nullptr;
    cplx* dscale = frequency_shifts ? a.take<cplx>(nS/16) : nullptr;
    int32_t* didx = frequency_shifts ? a.take<int32_t>(n_idx) : nullptr;
    double* dout = frequency_shifts ? a.take<double>(nout) : nullptr;
    FFK_REQUIRE(dF && (!frequency_shifts || dout) && a.used <= g_arena.size,
                "internal: arena too small");
    auto h2d = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, nullptr);
    };
    FFK_HIP(h2d(dD, eigvals, 8*size_t(G)*d));
    FFK_HIP(h2d(dV, eigvecs, 16*size_t(G)*dd));
    FFK_HIP(h2d(dQ, propagators, 16*size_t(G + 1)*dd));
    FFK_HIP(h2d(dom, omega, 8*size_t(W)));
    FFK_HIP(h2d(dbasis, basis, 16*size_t(N)*dd));
    FFK_HIP(h2d(dnop, n_opers, 16*size_t(A)*dd));
    FFK_HIP(h2d(dnc, n_coeffs, 8*size_t(A)*G));
    FFK_HIP(h2d(ddt, dt, 8*size_t(G)));
    FFK_HIP(h2d(dtt, t, 8*size_t(G + 1)));
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dnop, dnc, ddt, dtt, G, d, A, segtab, Tc, ops, dnt, dep, nullptr));
    FFK_HIP(ffk::launch_basis_transformed(Tc, dbasis, G, N, d, dbt, nullptr));
    FFK_HIP(ffk::launch_second_order_filter_function(dom, W, dD, ddt, dtt, dnt, dbt, G, d, A, N, dF, ws,
                                                     nullptr));
    if (frequency_shifts) {
        FFK_HIP(h2d(dS, spectrum, nS));
        FFK_HIP(h2d(didx, idx, 4*size_t(n_idx)));
        FFK_HIP(ffk::launch_spectral_weights(dS, srows, W, dom, W, 0, dscale, nullptr));
        FFK_HIP(ffk::launch_frequency_shifts(dF, A, N, W, dscale, s_ndim, didx, n_idx, dout, nullptr));
        FFK_HIP(hipMemcpyAsync(frequency_shifts, dout, 8*nout, hipMemcpyDeviceToHost, nullptr));
    }
    if (filter_function_2)
        FFK_HIP(hipMemcpyAsync(filter_function_2, dF, 16*nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_second_order_filter_function(const double* eigvals, const double* eigvecs,
                                     const double* propagators, const double* omega, int W,
                                     const double* basis, int N, const double* n_opers, int A,
                                     const double* n_coeffs, const double* dt, const double* t, int G,
                                     int d, double* filter_function_2) {
    FFK_REQUIRE(filter_function_2, "NULL argument");
    return second_order_impl(eigvals, eigvecs, propagators, omega, W, basis, N, n_opers, A, n_coeffs, dt,
                             t, G, d, filter_function_2, nullptr, 0, nullptr, 0, nullptr);
}

int ffk_frequency_shifts_from_scratch(const double* eigvals, const double* eigvecs,
                                      const double* propagators, const double* omega, int W,
                                      const double* basis, int N, const double* n_opers, int A,
                                      const double* n_coeffs, const double* dt, const double* t, int G,
                                      int d, const double* spectrum, int s_ndim, const int32_t* idx,
                                      int n_idx, double* filter_function_2, double* frequency_shifts) {
    FFK_REQUIRE(frequency_shifts, "NULL argument");
    return second_order_impl(eigvals, eigvecs, propagators, omega, W, basis, N, n_opers, A, n_coeffs, dt,
                             t, G, d, filter_function_2, spectrum, s_ndim, idx, n_idx, frequency_shifts);
}

int ffk_second_order_filter_function_from_atomic(const double* filter_function_atomic,
                                                 const double* control_matrix_step,
                                                 const double* propagators_liouville, int G, int A,
                                                 int N, int W, double* filter_function_2) {
    FFK_REQUIRE(filter_function_atomic && control_matrix_step && filter_function_2, "NULL argument");
    FFK_REQUIRE(G >= 1 && A >= 1 && N >= 1 && W >= 1, "empty axis: G=%d A=%d N=%d W=%d", G, A, N, W);
    FFK_REQUIRE(G == 1 || propagators_liouville, "NULL argument");
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nF = 16*size_t(A)*A*N*N*W, nR = 16*size_t(G)*A*N*W;
    const size_t nL = 8*size_t(G > 1 ? G - 1 : 1)*N*N;
    const size_t wsb = ffk::second_order_from_atomic_workspace_bytes(G, A, N, W);
    void* base;
    if (int rc = arena_reserve(align_up(size_t(G)*nF) + align_up(nR) + align_up(nL) + wsb +
                                   align_up(nF), &base))
        return rc;
    Bump a(base, g_arena.size);
    cplx* dFa = a.take<cplx>(size_t(G)*nF/16);
    cplx* dR = a.take<cplx>(nR/16);
    double* dL = a.take<double>(nL/8);
    void* ws = a.take<unsigned char>(wsb);
    cplx* dout = a.take<cplx>(nF/16);
    FFK_REQUIRE(dout, "internal: arena too small");
    FFK_HIP(hipMemcpyAsync(dFa, filter_function_atomic, size_t(G)*nF, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dR, control_matrix_step, nR, hipMemcpyHostToDevice, nullptr));
    if (G > 1)
        FFK_HIP(hipMemcpyAsync(dL, propagators_liouville, 8*size_t(G - 1)*N*N, hipMemcpyHostToDevice,
                               nullptr));
    FFK_HIP(ffk::launch_second_order_from_atomic(dFa, dR, dL, G, A, N, W, dout, ws, nullptr));
    FFK_HIP(hipMemcpyAsync(filter_function_2, dout, nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_frequency_shifts(const double* filter_function_2, int A, int N, int W, const double* spectrum,
                         int s_ndim, const double* omega, const int32_t* idx, int n_idx,
                         double* frequency_shifts) {
    FFK_REQUIRE(filter_function_2 && spectrum && omega && idx && frequency_shifts, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1 && n_idx >= 1, "empty axis");
    for (int i = 0; i < n_idx; ++i)
        FFK_REQUIRE(idx[i] >= 0 && idx[i] < A, "noise operator index %d out of range [0, %d)", idx[i], A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nF = 16*size_t(A)*A*N*N*W;
    const int rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : n_idx*n_idx);
    const size_t nS = 16*size_t(W)*rows;
    const size_t nout = size_t(n_idx)*(s_ndim == 3 ? n_idx : 1)*N*N;
    void* base;
    if (int rc = arena_reserve(align_up(nF) + 2*align_up(nS) + align_up(8*size_t(W)) +
                                   align_up(4*size_t(n_idx)) + align_up(8*nout), &base))
        return rc;
    Bump a(base, g_arena.size);
    cplx* dF = a.take<cplx>(nF/16);
    cplx* dS = a.take<cplx>(nS/16);
    cplx* dscale = a.take<cplx>(nS/16);
    double* dom = a.take<double>(W);
    int32_t* didx = a.take<int32_t>(n_idx);
    double* dout = a.take<double>(nout);
    FFK_REQUIRE(dout, "internal: arena too small");
    FFK_HIP(hipMemcpyAsync(dF, filter_function_2, nF, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dS, spectrum, nS, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dom, omega, 8*size_t(W), hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(didx, idx, 4*size_t(n_idx), hipMemcpyHostToDevice, nullptr));
    FFK_HIP(ffk::launch_spectral_weights(dS, rows, W, dom, W, 0, dscale, nullptr));
    FFK_HIP(ffk::launch_frequency_shifts(dF, A, N, W, dscale, s_ndim, didx, n_idx, dout, nullptr));
    FFK_HIP(hipMemcpyAsync(frequency_shifts, dout, 8*nout, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_cumulant_function_second_order(const double* frequency_shifts, int batch, int N, int d,
                                       const double* basis, double* cumulant_function) {
    FFK_REQUIRE(frequency_shifts && basis && cumulant_function, "NULL argument");
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis");
    FFK_REQUIRE(d_templated_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D_TEMPLATED);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nG = 8*size_t(batch)*N*N;
    const size_t nB = 16*size_t(N)*d*d;
    const size_t wsb = ffk::cumulant_second_order_workspace_bytes(batch, N, d);
    void* base;
    if (int rc = arena_reserve(2*align_up(nG) + align_up(nB) + align_up(wsb), &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(nG/8);
    double* dK = a.take<double>(nG/8);
    double* dB = a.take<double>(nB/8);
    void* ws = a.take<unsigned char>(wsb);
    FFK_REQUIRE(ws, "internal: arena too small");
    FFK_HIP(hipMemcpyAsync(dD, frequency_shifts, nG, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dK, cumulant_function, nG, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dB, basis, nB, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(ffk::launch_cumulant_second_order(dD, batch, N, d, reinterpret_cast<const cplx*>(dB), dK, ws,
                                              nullptr));
    FFK_HIP(hipMemcpyAsync(cumulant_function, dK, nG, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

size_t ffk_second_order_workspace_bytes(int W, int N, int A, int G, int d) {
    if (W < 1 || N < 1 || A < 1 || G < 1 || !d_templated_ok(d)) return 0;
    const size_t dd = size_t(d)*d;
    size_t b = 0;
    b += align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd);        // segtab, Tc
    b += align_up(16*size_t(G)*(1 + A)*dd);                                           // ops
    b += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(G)*dd);                     // nt, ep
    b += align_up(16*size_t(G)*N*dd);                                                 // bt
    b += ffk::second_order_workspace_bytes(G, A, N, d);                               // NB, M
    return b;
}

int ffk_second_order_filter_function_dev(const double* eigvals, const double* eigvecs,
                                         const double* propagators, const double* omega, int W,
                                         const double* basis, int N, const double* n_opers, int A,
                                         const double* n_coeffs, const double* dt, const double* t,
                                         int G, int d, double* filter_function_2, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && basis && n_opers && n_coeffs && dt && t &&
                    filter_function_2 && workspace, "NULL argument");
    FFK_REQUIRE(size_t(A)*N <= 65535, "A*N = %zu too large", size_t(A)*N);
    FFK_REQUIRE(workspace_bytes >= ffk_second_order_workspace_bytes(W, N, A, G, d), "workspace too small");
    const size_t dd = size_t(d)*d;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Bump a(workspace, workspace_bytes);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + A)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dbt = a.take<cplx>(size_t(G)*N*dd);
    void* ws = a.take<unsigned char>(ffk::second_order_workspace_bytes(G, A, N, d));
    FFK_REQUIRE(ws, "internal: workspace too small");
    FFK_HIP(ffk::launch_prologue(eigvals, reinterpret_cast<const cplx*>(eigvecs),
                                 reinterpret_cast<const cplx*>(propagators),
                                 reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, G, d, A, segtab,
                                 Tc, ops, dnt, dep, st));
    FFK_HIP(ffk::launch_basis_transformed(Tc, reinterpret_cast<const cplx*>(basis), G, N, d, dbt, st));
    FFK_HIP(ffk::launch_second_order_filter_function(omega, W, eigvals, dt, t, dnt, dbt, G, d, A, N,
                                                     reinterpret_cast<cplx*>(filter_function_2), ws, st));
    return FFK_OK;
}

size_t ffk_frequency_shifts_workspace_bytes(int W, int n_idx, int s_ndim) {
    if (W < 1 || n_idx < 1 || s_ndim < 1 || s_ndim > 3) return 0;
    return align_up(16*size_t(W)*(s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : size_t(n_idx)*n_idx)));
}

int ffk_frequency_shifts_shard_dev(const double* filter_function_2, int A, int N, int W_block,
                                   const double* spectrum, int s_ndim, const double* omega, int W,
                                   int w_offset, const int32_t* idx, int n_idx,
                                   double* frequency_shifts, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    FFK_REQUIRE(filter_function_2 && spectrum && omega && idx && frequency_shifts && workspace,
                "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(A >= 1 && N >= 1 && W_block >= 1 && n_idx >= 1, "empty axis");
    FFK_REQUIRE(w_offset >= 0 && w_offset + W_block <= W, "frequency block [%d, %d) outside [0, %d)",
                w_offset, w_offset + W_block, W);
    FFK_REQUIRE(workspace_bytes >= ffk_frequency_shifts_workspace_bytes(W_block, n_idx, s_ndim),
                "workspace too small");
    const int rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? n_idx : n_idx*n_idx);
    hipStream_t st = static_cast<hipStream_t>(stream);
    cplx* scale = static_cast<cplx*>(workspace);
    FFK_HIP(ffk::launch_spectral_weights(reinterpret_cast<const cplx*>(spectrum), rows, W_block, omega, W,
                                         w_offset, scale, st));
    FFK_HIP(ffk::launch_frequency_shifts(reinterpret_cast<const cplx*>(filter_function_2), A, N, W_block,
                                         scale, s_ndim, idx, n_idx, frequency_shifts, st));
    return FFK_OK;
}

size_t ffk_cumulant_function_second_order_workspace_bytes(int batch, int N, int d) {
    if (batch < 1 || N < 1 || !d_templated_ok(d)) return 0;
    return align_up(ffk::cumulant_second_order_workspace_bytes(batch, N, d));
}

int ffk_cumulant_function_second_order_dev(const double* frequency_shifts, int batch, int N, int d,
                                           const double* basis, double* cumulant_function,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(frequency_shifts && basis && cumulant_function && workspace, "NULL argument");
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis");
    FFK_REQUIRE(d_templated_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(workspace_bytes >= ffk_cumulant_function_second_order_workspace_bytes(batch, N, d),
                "workspace too small");
    FFK_HIP(ffk::launch_cumulant_second_order(frequency_shifts, batch, N, d,
                                              reinterpret_cast<const cplx*>(basis), cumulant_function,
                                              workspace, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// gradient: derivative of the filter function / infidelity w.r.t. the control amplitudes
// ---------------------------------------------------------------------------------------------
int ffk_filter_function_derivative(const double* eigvals, const double* eigvecs,
                                   const double* propagators, const double* omega, int W,
                                   const double* n_opers, int A, const double* n_coeffs,
                                   const double* c_opers, int H, const double* n_coeffs_ratio,
                                   const double* dt, const double* t, int G, int d,
                                   const double* spectrum, int s_ndim,
                                   double* filter_function_derivative,
                                   double* infidelity_derivative) {
    FFK_REQUIRE(d >= 2 && d <= 8, "the gradient kernels support 2 <= d <= 8, not d=%d", d);
    FFK_REQUIRE(W >= 1 && A >= 1 && H >= 1 && G >= 1, "empty axis: W=%d A=%d H=%d G=%d", W, A, H, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && n_opers && n_coeffs && c_opers && dt && t,
                "NULL argument");
    FFK_REQUIRE(filter_function_derivative || infidelity_derivative, "no output requested");
    FFK_REQUIRE(!infidelity_derivative || (spectrum && (s_ndim == 1 || s_ndim == 2)),
                "infidelity derivative needs a spectrum of shape (W,) or (A, W)");
    FFK_REQUIRE(size_t(G)*A <= 65535, "G*A = %zu too large", size_t(G)*A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nY = size_t(G)*A*dd*W, nF = size_t(A)*G*H*W;
    const int srows = s_ndim == 2 ? A : 1;
    const size_t nS = infidelity_derivative ? 16*size_t(W)*srows : 0;
    const int HA = H > A ? H : A;
    size_t total = 0;
    total += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd);
    total += align_up(8*size_t(W)) + align_up(16*size_t(A)*dd) + align_up(16*size_t(H)*dd);
    total += align_up(8*size_t(A)*G) + align_up(8*size_t(H)*G) + align_up(8*size_t(G)) + align_up(8*size_t(G + 1));
    total += align_up(8*size_t(A)*H*G);
    total += 2*(align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd) +
                align_up(16*size_t(G)*(1 + HA)*dd));
    total += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(H)*G*dd) + 2*align_up(16*size_t(G)*dd);
    total += align_up(16*size_t(H)*G*dd);                                       // E
    total += align_up(16*nY) + align_up(8*nF) + 2*align_up(nS) + align_up(8*size_t(A)*G*H);
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(size_t(G)*d);
    cplx* dV = a.take<cplx>(size_t(G)*dd);
    cplx* dQ = a.take<cplx>(size_t(G + 1)*dd);
    double* dom = a.take<double>(W);
    cplx* dnop = a.take<cplx>(size_t(A)*dd);
    cplx* dcop = a.take<cplx>(size_t(H)*dd);
    double* dnc = a.take<double>(size_t(A)*G);
    double* ddt = a.take<double>(G);
    double* dtt = a.take<double>(G + 1);
    double* dratio = a.take<double>(size_t(A)*H*G);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    double* segtab2 = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc2 = a.take<cplx>(size_t(G)*dd);
    cplx* ops2 = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dabar = a.take<cplx>(size_t(H)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dep2 = a.take<cplx>(size_t(G)*dd);
    cplx* dE = a.take<cplx>(size_t(H)*G*dd);
    cplx* Y = a.take<cplx>(nY);
    double* dF = a.take<double>(nF);
    cplx* dS = nS ? a.take<cplx>(nS/16) : nullptr;
    cplx* dscale = nS ? a.take<cplx>(nS/16) : nullptr;
    double* dI = a.take<double>(size_t(A)*G*H);
    FFK_REQUIRE(dI && a.used <= g_arena.size, "internal: arena too small");
    auto h2d = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, nullptr);
    };
    FFK_HIP(h2d(dD, eigvals, 8*size_t(G)*d));
    FFK_HIP(h2d(dV, eigvecs, 16*size_t(G)*dd));
    FFK_HIP(h2d(dQ, propagators, 16*size_t(G + 1)*dd));
    FFK_HIP(h2d(dom, omega, 8*size_t(W)));
    FFK_HIP(h2d(dnop, n_opers, 16*size_t(A)*dd));
    FFK_HIP(h2d(dcop, c_opers, 16*size_t(H)*dd));
    FFK_HIP(h2d(dnc, n_coeffs, 8*size_t(A)*G));
    FFK_HIP(h2d(ddt, dt, 8*size_t(G)));
    FFK_HIP(h2d(dtt, t, 8*size_t(G + 1)));
    if (n_coeffs_ratio) FFK_HIP(h2d(dratio, n_coeffs_ratio, 8*size_t(A)*H*G));
    // Bbar, T (noise operators) and Abar (control operators, unit coefficients)
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dnop, dnc, ddt, dtt, G, d, A, segtab, Tc, ops, dnt, dep, nullptr));
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dcop, nullptr, ddt, dtt, G, d, H, segtab2, Tc2, ops2, dabar,
                                 dep2, nullptr));
    // Hilbert-space steps of the interaction-picture noise operators, one chunk per segment, then
    // their running sums
    ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, G);
    FFK_HIP(ffk::launch_accumulate(dom, W, segtab, ops, G, d, A, geo, Y, nullptr));
    FFK_HIP(ffk::launch_segment_prefix_sum(Y, G, size_t(A)*dd*W, nullptr));
    FFK_HIP(ffk::launch_filter_function_derivative(dom, W, dD, ddt, dtt, ops, dabar, Y,
                                                   n_coeffs_ratio ? dratio : nullptr, G, d, A, H, dE, dF,
                                                   nullptr));
    if (infidelity_derivative) {
        FFK_HIP(h2d(dS, spectrum, nS));
        FFK_HIP(ffk::launch_spectral_weights(dS, srows, W, dom, W, 0, dscale, nullptr));
        FFK_HIP(ffk::launch_infidelity_derivative(dF, A, G, H, W, dscale, s_ndim, d, dI, nullptr));
        FFK_HIP(hipMemcpyAsync(infidelity_derivative, dI, 8*size_t(A)*G*H, hipMemcpyDeviceToHost, nullptr));
    }
    if (filter_function_derivative)
        FFK_HIP(hipMemcpyAsync(filter_function_derivative, dF, 8*nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_control_matrix_derivative(const double* eigvals, const double* eigvecs, const double* propagators,
                                  const double* omega, int W, const double* basis, int N,
                                  const double* n_opers, int A, const double* n_coeffs,
                                  const double* c_opers, int H, const double* n_coeffs_ratio,
                                  const double* dt, const double* t, int G, int d,
                                  double* control_matrix_derivative) {
    FFK_REQUIRE(d >= 2 && d <= 8, "the gradient kernels support 2 <= d <= 8, not d=%d", d);
    FFK_REQUIRE(W >= 1 && A >= 1 && H >= 1 && G >= 1 && N >= 1, "empty axis: W=%d A=%d H=%d G=%d N=%d", W,
                A, H, G, N);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && basis && n_opers && n_coeffs && c_opers && dt &&
                    t && control_matrix_derivative, "NULL argument");
    FFK_REQUIRE(size_t(G)*A <= 65535, "G*A = %zu too large", size_t(G)*A);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nY = size_t(G)*A*dd*W, nR = size_t(H)*W*G*A*N;
    const int HA = H > A ? H : A;
    size_t total = 0;
    total += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd);
    total += align_up(8*size_t(W)) + align_up(16*size_t(A)*dd) + align_up(16*size_t(H)*dd);
    total += align_up(16*size_t(N)*dd);
    total += align_up(8*size_t(A)*G) + align_up(8*size_t(G)) + align_up(8*size_t(G + 1));
    total += align_up(8*size_t(A)*H*G);
    total += 2*(align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd) +
                align_up(16*size_t(G)*(1 + HA)*dd));
    total += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(H)*G*dd) + 2*align_up(16*size_t(G)*dd);
    total += align_up(16*size_t(H)*G*dd);
    total += align_up(16*nY) + align_up(16*nR);
    void* base;
    if (int rc = arena_reserve(total, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dD = a.take<double>(size_t(G)*d);
    cplx* dV = a.take<cplx>(size_t(G)*dd);
    cplx* dQ = a.take<cplx>(size_t(G + 1)*dd);
    double* dom = a.take<double>(W);
    cplx* dnop = a.take<cplx>(size_t(A)*dd);
    cplx* dcop = a.take<cplx>(size_t(H)*dd);
    cplx* dbasis = a.take<cplx>(size_t(N)*dd);
    double* dnc = a.take<double>(size_t(A)*G);
    double* ddt = a.take<double>(G);
    double* dtt = a.take<double>(G + 1);
    double* dratio = a.take<double>(size_t(A)*H*G);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    double* segtab2 = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc2 = a.take<cplx>(size_t(G)*dd);
    cplx* ops2 = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dabar = a.take<cplx>(size_t(H)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dep2 = a.take<cplx>(size_t(G)*dd);
    cplx* dE = a.take<cplx>(size_t(H)*G*dd);
    cplx* Y = a.take<cplx>(nY);
    cplx* dR = a.take<cplx>(nR);
    FFK_REQUIRE(dR && a.used <= g_arena.size, "internal: arena too small");
    auto h2d = [](void* dst, const void* src, size_t n) {
        return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, nullptr);
    };
    FFK_HIP(h2d(dD, eigvals, 8*size_t(G)*d));
    FFK_HIP(h2d(dV, eigvecs, 16*size_t(G)*dd));
    FFK_HIP(h2d(dQ, propagators, 16*size_t(G + 1)*dd));
    FFK_HIP(h2d(dom, omega, 8*size_t(W)));
    FFK_HIP(h2d(dnop, n_opers, 16*size_t(A)*dd));
    FFK_HIP(h2d(dcop, c_opers, 16*size_t(H)*dd));
    FFK_HIP(h2d(dbasis, basis, 16*size_t(N)*dd));
    FFK_HIP(h2d(dnc, n_coeffs, 8*size_t(A)*G));
    FFK_HIP(h2d(ddt, dt, 8*size_t(G)));
    FFK_HIP(h2d(dtt, t, 8*size_t(G + 1)));
    if (n_coeffs_ratio) FFK_HIP(h2d(dratio, n_coeffs_ratio, 8*size_t(A)*H*G));
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dnop, dnc, ddt, dtt, G, d, A, segtab, Tc, ops, dnt, dep, nullptr));
    FFK_HIP(ffk::launch_prologue(dD, dV, dQ, dcop, nullptr, ddt, dtt, G, d, H, segtab2, Tc2, ops2, dabar,
                                 dep2, nullptr));
    ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, G);
    FFK_HIP(ffk::launch_accumulate(dom, W, segtab, ops, G, d, A, geo, Y, nullptr));
    FFK_HIP(ffk::launch_segment_prefix_sum(Y, G, size_t(A)*dd*W, nullptr));
    FFK_HIP(ffk::launch_control_matrix_derivative(dom, W, dD, ddt, dtt, ops, dabar, Y,
                                                  n_coeffs_ratio ? dratio : nullptr, dbasis, N, G, d, A, H,
                                                  dE, dR, nullptr));
    FFK_HIP(hipMemcpyAsync(control_matrix_derivative, dR, 16*nR, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

int ffk_filter_function_derivative_from_control_matrix(const double* control_matrix,
                                                       const double* control_matrix_derivative, int A,
                                                       int N, int W, int G, int H,
                                                       double* filter_function_derivative) {
    FFK_REQUIRE(control_matrix && control_matrix_derivative && filter_function_derivative, "NULL argument");
    FFK_REQUIRE(A >= 1 && N >= 1 && W >= 1 && G >= 1 && H >= 1, "empty axis: A=%d N=%d W=%d G=%d H=%d", A, N,
                W, G, H);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nR = size_t(A)*N*W, nD = size_t(H)*W*G*A*N, nF = size_t(A)*G*H*W;
    void* base;
    if (int rc = arena_reserve(align_up(16*nR) + align_up(16*nD) + align_up(8*nF), &base)) return rc;
    Bump a(base, g_arena.size);
    cplx* dR = a.take<cplx>(nR);
    cplx* dD = a.take<cplx>(nD);
    double* dF = a.take<double>(nF);
    FFK_REQUIRE(dF && a.used <= g_arena.size, "internal: arena too small");
    FFK_HIP(hipMemcpyAsync(dR, control_matrix, 16*nR, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dD, control_matrix_derivative, 16*nD, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(ffk::launch_filter_function_derivative_from_control_matrix(dR, dD, A, N, W, G, H, dF, nullptr));
    FFK_HIP(hipMemcpyAsync(filter_function_derivative, dF, 8*nF, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

size_t ffk_filter_function_derivative_workspace_bytes(int W, int A, int H, int G, int d) {
    if (W < 1 || A < 1 || H < 1 || G < 1 || d < 2 || d > 8) return 0;
    const size_t dd = size_t(d)*d;
    const int HA = H > A ? H : A;
    size_t b = 0;
    b += 2*(align_up(8*size_t(G)*ffk::seg_stride(d)) + align_up(16*size_t(G)*dd) +
            align_up(16*size_t(G)*(1 + HA)*dd));                                   // segtab, Tc, ops (x2)
    b += align_up(16*size_t(A)*G*dd) + align_up(16*size_t(H)*G*dd) + 2*align_up(16*size_t(G)*dd);
    b += align_up(16*size_t(H)*G*dd);                                              // E
    b += align_up(16*size_t(G)*A*dd*W);                                            // Y steps / Ycum
    b += align_up(16*size_t(W)*A);                                                 // spectral weights
    return b;
}

int ffk_filter_function_derivative_shard_dev(const double* eigvals, const double* eigvecs,
                                             const double* propagators, const double* omega_block,
                                             int W_block, const double* n_opers, int A,
                                             const double* n_coeffs, const double* c_opers, int H,
                                             const double* n_coeffs_ratio, const double* dt,
                                             const double* t, int G, int d, const double* spectrum,
                                             int s_ndim, const double* omega, int W, int w_offset,
                                             double* filter_function_derivative,
                                             double* infidelity_derivative, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(d >= 2 && d <= 8, "the gradient kernels support 2 <= d <= 8, not d=%d", d);
    FFK_REQUIRE(W_block >= 1 && A >= 1 && H >= 1 && G >= 1, "empty axis");
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega_block && n_opers && n_coeffs && c_opers && dt &&
                    t && filter_function_derivative && workspace, "NULL argument");
    FFK_REQUIRE(!infidelity_derivative || (spectrum && omega && (s_ndim == 1 || s_ndim == 2)),
                "infidelity derivative needs a spectrum of shape (W,) or (A, W) and the global grid");
    FFK_REQUIRE(!infidelity_derivative || (w_offset >= 0 && w_offset + W_block <= W),
                "frequency block [%d, %d) outside [0, %d)", w_offset, w_offset + W_block, W);
    FFK_REQUIRE(size_t(G)*A <= 65535, "G*A = %zu too large", size_t(G)*A);
    FFK_REQUIRE(workspace_bytes >= ffk_filter_function_derivative_workspace_bytes(W_block, A, H, G, d),
                "workspace too small");
    const size_t dd = size_t(d)*d;
    const int HA = H > A ? H : A;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Bump a(workspace, workspace_bytes);
    double* segtab = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc = a.take<cplx>(size_t(G)*dd);
    cplx* ops = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    double* segtab2 = a.take<double>(size_t(G)*ffk::seg_stride(d));
    cplx* Tc2 = a.take<cplx>(size_t(G)*dd);
    cplx* ops2 = a.take<cplx>(size_t(G)*(1 + HA)*dd);
    cplx* dnt = a.take<cplx>(size_t(A)*G*dd);
    cplx* dabar = a.take<cplx>(size_t(H)*G*dd);
    cplx* dep = a.take<cplx>(size_t(G)*dd);
    cplx* dep2 = a.take<cplx>(size_t(G)*dd);
    cplx* dE = a.take<cplx>(size_t(H)*G*dd);
    cplx* Y = a.take<cplx>(size_t(G)*A*dd*W_block);
    cplx* dscale = a.take<cplx>(size_t(W_block)*A);
    FFK_REQUIRE(dscale, "internal: workspace too small");
    const cplx* V = reinterpret_cast<const cplx*>(eigvecs);
    const cplx* Q = reinterpret_cast<const cplx*>(propagators);
    FFK_HIP(ffk::launch_prologue(eigvals, V, Q, reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, G,
                                 d, A, segtab, Tc, ops, dnt, dep, st));
    FFK_HIP(ffk::launch_prologue(eigvals, V, Q, reinterpret_cast<const cplx*>(c_opers), nullptr, dt, t, G,
                                 d, H, segtab2, Tc2, ops2, dabar, dep2, st));
    ffk::AccumGeometry geo = ffk::accumulate_geometry(W_block, A, G, d, G);
    FFK_HIP(ffk::launch_accumulate(omega_block, W_block, segtab, ops, G, d, A, geo, Y, st));
    FFK_HIP(ffk::launch_segment_prefix_sum(Y, G, size_t(A)*dd*W_block, st));
    FFK_HIP(ffk::launch_filter_function_derivative(omega_block, W_block, eigvals, dt, t, ops, dabar, Y,
                                                   n_coeffs_ratio, G, d, A, H, dE,
                                                   filter_function_derivative, st));
    if (infidelity_derivative) {
        const int srows = s_ndim == 2 ? A : 1;
        FFK_HIP(ffk::launch_spectral_weights(reinterpret_cast<const cplx*>(spectrum), srows, W_block, omega,
                                             W, w_offset, dscale, st));
        FFK_HIP(ffk::launch_infidelity_derivative(filter_function_derivative, A, G, H, W_block, dscale,
                                                  s_ndim, d, infidelity_derivative, st));
    }
    return FFK_OK;
}

int ffk_expm_real(const double* matrix, int N, double* result) {
    FFK_REQUIRE(matrix && result, "NULL argument");
    FFK_REQUIRE(N >= 1 && N <= 4096, "matrix dimension %d outside [1, 4096]", N);
    // scaling from the 1-norm (host: the matrix is N^2 <= 65536 doubles on this path)
    double norm = 0.0;
    for (int j = 0; j < N; ++j) {
        double col = 0.0;
        for (int i = 0; i < N; ++i) {
            const double v = matrix[size_t(i)*N + j];
            FFK_REQUIRE(v == v && v - v == 0.0, "matrix contains NaN or Inf");
            col += v < 0 ? -v : v;
        }
        norm = col > norm ? col : norm;
    }
    int squarings = 0;
    while (norm > 0.5 && squarings < 64) {
        norm *= 0.5;
        ++squarings;
    }
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t nb = 8*size_t(N)*N;
    void* base;
    if (int rc = arena_reserve(7*align_up(nb), &base)) return rc;
    Bump a(base, g_arena.size);
    double* dA = a.take<double>(nb/8);
    double* dO = a.take<double>(nb/8);
    double* w[5];
    for (double*& m : w) m = a.take<double>(nb/8);
    FFK_HIP(hipMemcpyAsync(dA, matrix, nb, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(ffk::launch_expm_real(dA, N, squarings, dO, w, nullptr));
    FFK_HIP(hipMemcpyAsync(result, dO, nb, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// Liouville representation
// ---------------------------------------------------------------------------------------------
size_t ffk_liouville_workspace_bytes(int batch, int d, int N) {
    if (batch < 1 || N < 1 || !d_ok(d)) return 0;
    return ffk::liouville_workspace_bytes(batch, d, N);
}

int ffk_liouville_dev(const double* U, int batch, int d, const double* basis, int N,
                      int hermitian_basis, double* liouville, void* workspace,
                      size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis: batch=%d N=%d", batch, N);
    FFK_REQUIRE(U && basis && liouville && workspace, "NULL argument");
    FFK_REQUIRE(workspace_bytes >= ffk_liouville_workspace_bytes(batch, d, N), "workspace too small");
    FFK_HIP(ffk::launch_liouville(reinterpret_cast<const cplx*>(U), batch, d,
                                  reinterpret_cast<const cplx*>(basis), N, hermitian_basis, liouville,
                                  workspace, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

int ffk_liouville(const double* U, int batch, int d, const double* basis, int N, int hermitian_basis,
                  double* liouville) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis: batch=%d N=%d", batch, N);
    FFK_REQUIRE(U && basis && liouville, "NULL argument");
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nU = 16*size_t(batch)*dd, nB = 16*size_t(N)*dd;
    const size_t nL = (hermitian_basis ? 8 : 16)*size_t(batch)*N*N;
    const size_t wsb = ffk_liouville_workspace_bytes(batch, d, N);
    void* base;
    if (int rc = arena_reserve(align_up(nU) + align_up(nB) + align_up(nL) + wsb, &base)) return rc;
    Bump a(base, g_arena.size);
    double* dU = a.take<double>(nU/8);
    double* dB = a.take<double>(nB/8);
    double* dL = a.take<double>(nL/8);
    void* ws = a.take<unsigned char>(wsb);
    FFK_HIP(hipMemcpyAsync(dU, U, nU, hipMemcpyHostToDevice, nullptr));
    FFK_HIP(hipMemcpyAsync(dB, basis, nB, hipMemcpyHostToDevice, nullptr));
    if (int rc = ffk_liouville_dev(dU, batch, d, dB, N, hermitian_basis, dL, ws, wsb, nullptr)) return rc;
    FFK_HIP(hipMemcpyAsync(liouville, dL, nL, hipMemcpyDeviceToHost, nullptr));
    FFK_HIP(hipStreamSynchronize(nullptr));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// fused device-resident pipeline
// ---------------------------------------------------------------------------------------------
size_t ffk_pipeline_workspace_bytes(int W, int N, int A, int G, int d, int n_idx, int s_ndim) {
    if (W < 1 || N < 1 || A < 1 || G < 1 || !d_templated_ok(d)) return 0;
    size_t b = ffk_diagonalize_workspace_bytes(G, d) + ffk_control_matrix_workspace_bytes(W, N, A, G, d);
    b += align_up(8*size_t(G)*d) + align_up(16*size_t(G)*d*d) + align_up(16*size_t(G + 1)*d*d);
    b += align_up(16*size_t(A)*N*W) + align_up(16*size_t(A)*A*W);
    if (n_idx > 0 && s_ndim >= 1 && s_ndim <= 3) b += ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
    return b;
}

int ffk_pipeline_dev(const double* hamiltonian, const double* dt, const double* t, int G, int d,
                     const double* omega, int W, const double* basis, int N, const double* n_opers,
                     int A, const double* n_coeffs, const double* spectrum, int s_ndim,
                     const int32_t* idx, int n_idx, double* eigvals, double* eigvecs,
                     double* propagators, double* control_matrix, double* filter_function,
                     double* infid, void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(hamiltonian && dt && t && omega && basis && n_opers && n_coeffs && workspace, "NULL argument");
    const bool want_infid = spectrum != nullptr && infid != nullptr;
    FFK_REQUIRE(!want_infid || (idx && n_idx >= 1 && s_ndim >= 1 && s_ndim <= 3), "bad spectrum arguments");
    FFK_REQUIRE(workspace_bytes >= ffk_pipeline_workspace_bytes(W, N, A, G, d, want_infid ? n_idx : 0, s_ndim),
                "workspace too small");
    Bump ws(workspace, workspace_bytes);
    const size_t dwsb = ffk_diagonalize_workspace_bytes(G, d);
    const size_t cwsb = ffk_control_matrix_workspace_bytes(W, N, A, G, d);
    void* dws = ws.take<unsigned char>(dwsb);
    void* cws = ws.take<unsigned char>(cwsb);
    double* D = eigvals ? eigvals : ws.take<double>(size_t(G)*d);
    double* V = eigvecs ? eigvecs : ws.take<double>(2*size_t(G)*d*d);
    double* Q = propagators ? propagators : ws.take<double>(2*size_t(G + 1)*d*d);
    double* R = control_matrix ? control_matrix : ws.take<double>(2*size_t(A)*N*W);
    double* F = filter_function ? filter_function : ws.take<double>(2*size_t(A)*A*W);
    unsigned cm_flags = 0;
    if (ffk::use_fused_front(G, d)) {
        // eigh, local scan, then scan fix-up + prologue writing straight into the control-matrix
        // workspace (same slicing as ffk_control_matrix_dev)
        hipStream_t s = static_cast<hipStream_t>(stream);
        const DiagWs w = slice_diag_ws(dws, dwsb, G, d);
        cplx* totals = static_cast<cplx*>(w.small);
        FFK_HIP(ffk::launch_eigh_expm(reinterpret_cast<const cplx*>(hamiltonian), dt, G, d, D,
                                      reinterpret_cast<cplx*>(V), w.seg_prop, w.status, s));
        FFK_HIP(ffk::launch_scan_local(w.seg_prop, G, d, ffk::front_chunk(d), w.qloc, totals, s));
        // same slicing as ffk_control_matrix_dev; the launch also compacts the basis (extra blocks)
        const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, g_forced_chunks);
        Bump cw(cws, cwsb);
        double* segtab = cw.take<double>(size_t(G)*ffk::seg_stride(d));
        cplx* Tc = cw.take<cplx>(size_t(G)*d*d);
        cplx* ops = cw.take<cplx>(size_t(G)*(1 + A)*d*d);
        cw.take<cplx>(size_t(geo.chunks)*A*d*d*W);     // Ypart
        cw.take<cplx>(size_t(A)*d*d*W);                // Bt
        void* ews = cw.take<unsigned char>(ffk::expand_workspace_bytes(N, d));
        FFK_REQUIRE(ews, "workspace too small");
        FFK_HIP(ffk::launch_apply_prologue_compact(
            w.qloc, totals, G, d, reinterpret_cast<cplx*>(Q), D, reinterpret_cast<const cplx*>(V),
            reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, A, segtab, Tc, ops,
            reinterpret_cast<const cplx*>(basis), N, ews, s));
        cm_flags = FFK_INTERNAL_PROLOGUE_DONE | FFK_INTERNAL_COMPACT_DONE;
    } else {
        if (int rc = ffk_diagonalize_dev(hamiltonian, dt, G, d, D, V, Q, dws, dwsb, stream)) return rc;
    }
    g_fuse_F = reinterpret_cast<cplx*>(F);
    g_fuse_F_done = false;
    const int rc_cm = ffk_control_matrix_dev(D, V, Q, omega, W, basis, N, n_opers, A, n_coeffs, dt, t,
                                             G, d, cm_flags, R, nullptr, cws, cwsb, stream);
    g_fuse_F = nullptr;
    if (rc_cm) return rc_cm;
    if (!g_fuse_F_done)
        if (int rc = ffk_filter_function_dev(R, A, N, W, FFK_FF_FIDELITY, F, stream)) return rc;
    if (want_infid) {
        const size_t iwsb = ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
        void* iws = ws.take<unsigned char>(iwsb);
        FFK_REQUIRE(iws, "workspace too small");
        if (int rc = ffk_infidelity_dev(F, A, W, spectrum, s_ndim, omega, idx, n_idx, d, infid, iws, iwsb, stream))
            return rc;
    }
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// eigensolver status of a device-resident run
// ---------------------------------------------------------------------------------------------
int ffk_eigensolver_status_dev(const void* workspace, size_t workspace_bytes, int G, int d,
                               int32_t* n_failed, void* stream) {
    FFK_REQUIRE(workspace && n_failed && G >= 1 && d_ok(d), "bad argument");
    FFK_REQUIRE(workspace_bytes >= ffk_diagonalize_workspace_bytes(G, d), "workspace too small");
    // the flags are the first slice of both the diagonalize and the pipeline workspace
    const DiagWs w = slice_diag_ws(const_cast<void*>(workspace), workspace_bytes, G, d);
    FFK_HIP(ffk::launch_count_failures(w.status, G, n_failed, static_cast<hipStream_t>(stream)));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
// resident evaluation: the user-facing PulseSequence.get_filter_function / infidelity call with
// one H2D, one pass of ffk_pipeline_dev, one D2H of the small results; R stays in HBM
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct Block {
    void* ptr;
    size_t size;
    int device = -1;      // device blocks belong to one device; pinned host blocks are portable
};

// Grow-only pools of device and pinned-host blocks: a PulseSequence is short-lived in user code
// (one per gate), hipMalloc / hipHostMalloc cost more than the whole pass at config 2.
struct BlockPool {
    std::mutex mu;
    std::vector<Block> free_blocks;
    bool pinned;
    explicit BlockPool(bool p) : pinned(p) {}
    int take(size_t bytes, int device, Block* out) {
        std::lock_guard<std::mutex> lock(mu);
        int best = -1;
        for (int i = 0; i < int(free_blocks.size()); ++i)
            if (free_blocks[i].size >= bytes && free_blocks[i].size <= 2*bytes + (1 << 16) &&
                (pinned || free_blocks[i].device == device) &&
                (best < 0 || free_blocks[i].size < free_blocks[best].size))
                best = i;
        if (best >= 0) {
            *out = free_blocks[best];
            free_blocks.erase(free_blocks.begin() + best);
            return FFK_OK;
        }
        const size_t want = align_up(bytes, size_t(1) << 16);
        void* p = nullptr;
        if (pinned)
            FFK_HIP(hipHostMalloc(&p, want, hipHostMallocPortable));
        else
            FFK_HIP(hipMalloc(&p, want));
        *out = {p, want, pinned ? -1 : device};
        return FFK_OK;
    }
    void give(Block b) {
        if (!b.ptr) return;
        std::lock_guard<std::mutex> lock(mu);
        if (free_blocks.size() >= 16) {       // bound what an idle process keeps: the OLDEST idle
            const Block old = free_blocks.front();   // block goes (a loop over fresh pulses of one
            free_blocks.erase(free_blocks.begin());  // shape must find its block again even after
            if (pinned) (void)hipHostFree(old.ptr); else (void)hipFree(old.ptr);   // other shapes filled the pool)
        }
        free_blocks.push_back(b);
    }
    int release() {
        std::lock_guard<std::mutex> lock(mu);
        for (Block& b : free_blocks) {
            if (pinned) FFK_HIP(hipHostFree(b.ptr)); else FFK_HIP(hipFree(b.ptr));
        }
        free_blocks.clear();
        return FFK_OK;
    }
};
BlockPool g_dev_pool(false), g_pin_pool(true);
// one stream per device for the resident passes, created on first use
std::mutex g_resident_stream_mu;
hipStream_t g_resident_streams[64] = {};

int resident_stream(hipStream_t* out) {
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    FFK_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_resident_stream_mu);
    if (!g_resident_streams[dev])
        FFK_HIP(hipStreamCreateWithFlags(&g_resident_streams[dev], hipStreamNonBlocking));
    *out = g_resident_streams[dev];
    return FFK_OK;
}

// byte offsets of the arrays inside the device block and (first two groups) the pinned block
struct ResidentLayout {
    size_t H, dt, t, omega, basis, n_opers, n_coeffs, inputs_end;     // one H2D
    size_t D, V, Q, F, status, outputs_end;                            // one D2H
    size_t R, S, idx, infid, end;                                      // device only (+ infid D2H)
};
ResidentLayout resident_layout(int G, int d, int W, int N, int A) {
    ResidentLayout L;
    const size_t dd = size_t(d)*d;
    size_t o = 0;
    auto put = [&o](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
    L.H = put(16*size_t(G)*dd);
    L.dt = put(8*size_t(G));
    L.t = put(8*size_t(G + 1));
    L.omega = put(8*size_t(W));
    L.basis = put(16*size_t(N)*dd);
    L.n_opers = put(16*size_t(A)*dd);
    L.n_coeffs = put(8*size_t(A)*G);
    L.inputs_end = o;
    L.D = put(8*size_t(G)*d);
    L.V = put(16*size_t(G)*dd);
    L.Q = put(16*size_t(G + 1)*dd);
    L.F = put(16*size_t(A)*A*W);
    L.status = put(sizeof(int32_t));
    L.outputs_end = o;
    L.R = put(16*size_t(A)*N*W);
    L.S = put(16*size_t(A)*A*W);          // largest spectrum: (A, A, W) c128
    L.idx = put(sizeof(int32_t)*size_t(A));
    L.infid = put(8*size_t(A)*A);
    L.end = o;
    return L;
}

}  // namespace

struct ffk_resident {
    double t_stage = 0, t_enqueue = 0, t_wait = 0;   // seconds, last pass (host clock)
    int device = -1;
    int G = 0, d = 0, W = 0, N = 0, A = 0;
    bool valid = false;
    Block dev = {nullptr, 0, -1}, pin = {nullptr, 0, -1};
    ResidentLayout L = {};
};

extern "C" {

int ffk_resident_create(ffk_resident** out) {
    FFK_REQUIRE(out, "NULL argument");
    *out = new (std::nothrow) ffk_resident();
    FFK_REQUIRE(*out, "out of host memory");
    return FFK_OK;
}

int ffk_resident_destroy(ffk_resident* r) {
    if (!r) return FFK_OK;
    g_dev_pool.give(r->dev);
    g_pin_pool.give(r->pin);
    delete r;
    return FFK_OK;
}

int ffk_resident_release_pools(void) {
    if (int rc = g_dev_pool.release()) return rc;
    return g_pin_pool.release();
}

}  // extern "C"

namespace {

// H[g] = sum_i c_coeffs[i, g] c_opers[i]  (pulse_sequence.py:1300-1302, 'ijk,il->ljk'), summed in
// operator order
__global__ void assemble_hamiltonian_kernel(const cplx* __restrict__ opers, const double* __restrict__ coeffs,
                                            int n_c, int G, int dd, cplx* __restrict__ H) {
    const size_t e = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
    if (e >= static_cast<size_t>(G)*dd) return;
    const int g = static_cast<int>(e / dd), k = static_cast<int>(e % dd);
    cplx acc = {0.0, 0.0};
    for (int i = 0; i < n_c; ++i) {
        const double c = coeffs[static_cast<size_t>(i)*G + g];
        const cplx o = opers[i*dd + k];
        acc.re = fma(c, o.re, acc.re);
        acc.im = fma(c, o.im, acc.im);
    }
    H[e] = acc;
}

// Captured resident passes.  The user-facing call builds a new PulseSequence (and handle) per pulse,
// but the block pools hand the same device / pinned blocks out again and the arena does not move:
// for a given shape the enqueue is then the SAME sequence of copies and launches on the same
// addresses, call after call.  It is captured once as a hipGraph and replayed with one launch
// (H2D of the packed inputs, up to 7 kernels, D2H of the outputs: 0.037 -> 0.012 ms of host time per
// call at config 2).  Key = everything the enqueue depends on; an entry whose addresses are no
// longer handed out simply never matches again and is evicted in turn (8 entries).
struct ResidentGraphKey {
    int dev, G, d, W, N, A, n_c, on_device;
    int s_ndim, n_idx, d_inf;                        // the integral riding in the pass (0: none)
    const void *dp, *hp, *ws;
    hipStream_t stream;
    unsigned long long epoch;
    bool operator==(const ResidentGraphKey& o) const {
        return dev == o.dev && G == o.G && d == o.d && W == o.W && N == o.N && A == o.A && n_c == o.n_c &&
               s_ndim == o.s_ndim && n_idx == o.n_idx && d_inf == o.d_inf &&
               on_device == o.on_device && dp == o.dp && hp == o.hp && ws == o.ws && stream == o.stream &&
               epoch == o.epoch;
    }
};
struct ResidentGraph {
    ResidentGraphKey key;
    hipGraphExec_t exec = nullptr;
    ffk_stats stats;
    unsigned long long used = 0;
};
constexpr int kResidentGraphs = 8;
ResidentGraph g_resident_graphs[kResidentGraphs];     // guarded by g_arena.mu (held by resident_pass)
unsigned long long g_resident_graph_clock = 0;
// A pass is captured on the SECOND sighting of its key only: the key contains the pool blocks'
// addresses, and a caller that keeps its pulses alive (a gate set, a list of pulses) never gets the
// same blocks back -- every call would pay capture + instantiate + destroy and evict the graphs
// that do repeat.  The first sighting is enqueued call by call and remembered here.
constexpr int kResidentSeen = 32;
ResidentGraphKey g_resident_seen[kResidentSeen];
bool g_resident_seen_valid[kResidentSeen] = {};
int g_resident_seen_next = 0;
bool resident_key_seen_before(const ResidentGraphKey& key) {
    for (int i = 0; i < kResidentSeen; ++i)
        if (g_resident_seen_valid[i] && g_resident_seen[i] == key) return true;
    g_resident_seen[g_resident_seen_next] = key;
    g_resident_seen_valid[g_resident_seen_next] = true;
    g_resident_seen_next = (g_resident_seen_next + 1) % kResidentSeen;
    return false;
}
bool resident_graphs_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("FFK_RESIDENT_GRAPH");
        return e == nullptr || e[0] != '0';
    }();
    return on;
}

// One resident pass; the Hamiltonian either given (G, d, d) or as control operators and
// amplitudes, in which case only the amplitudes cross PCIe (8 n_c B per segment instead of
// 16 d^2) and the sum runs on the device.
// Optionally the infidelity integral rides in the same pass (spectrum != NULL): ff.infidelity on a
// pulse with nothing cached is then ONE round trip to the device instead of two.
int resident_pass(ffk_resident* r, const double* hamiltonian, const double* c_opers, int n_c,
                  const double* c_coeffs, const double* dt, const double* t, int G, int d,
                  const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                  const double* n_coeffs, double** eigvals, double** eigvecs, double** propagators,
                  double** filter_function, const double* spectrum = nullptr, int s_ndim = 0,
                  int spectrum_is_real = 0, const int32_t* idx = nullptr, int n_idx = 0, int d_inf = 0,
                  double* infid = nullptr) {
    FFK_REQUIRE(r, "NULL handle");
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(hamiltonian || (c_opers && c_coeffs && n_c >= 1), "NULL argument");
    FFK_REQUIRE(dt && t && omega && basis && n_opers && n_coeffs, "NULL argument");
    FFK_REQUIRE(eigvals && eigvecs && propagators && filter_function, "NULL output argument");
    r->valid = false;
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    const ResidentLayout L = resident_layout(G, d, W, N, A);
    // spectrum (as c128), idx and the integrals live behind the outputs in the pinned block: the kernel
    // reads and writes them there (mapped memory), nothing extra crosses PCIe by copy
    size_t o_spec = 0, o_idx = 0, o_out = 0, pin_need = L.outputs_end, n_out = 0, s_rows = 0;
    if (spectrum) {
        FFK_REQUIRE(idx && infid && s_ndim >= 1 && s_ndim <= 3 && n_idx >= 1 && n_idx <= A && d_inf >= 1 && W >= 2,
                    "bad spectrum arguments");
        s_rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? size_t(n_idx) : size_t(n_idx)*n_idx);
        n_out = s_ndim == 3 ? size_t(n_idx)*n_idx : size_t(n_idx);
        o_spec = align_up(L.outputs_end);
        o_idx = o_spec + align_up(16*s_rows*W);
        o_out = o_idx + align_up(sizeof(int32_t)*size_t(n_idx));
        pin_need = o_out + align_up(8*n_out);
    }
    if (r->device != dev || r->dev.size < L.end || r->pin.size < pin_need) {
        g_dev_pool.give(r->dev);
        g_pin_pool.give(r->pin);
        r->dev = r->pin = Block{nullptr, 0, -1};
        if (int rc = g_dev_pool.take(L.end, dev, &r->dev)) return rc;
        if (int rc = g_pin_pool.take(pin_need, dev, &r->pin)) return rc;
        r->device = dev;
    }
    r->G = G; r->d = d; r->W = W; r->N = N; r->A = A; r->L = L;
    unsigned char* hp = static_cast<unsigned char*>(r->pin.ptr);
    unsigned char* dp = static_cast<unsigned char*>(r->dev.ptr);
    const size_t dd = size_t(d)*d;
    const auto clock0 = std::chrono::steady_clock::now();
    // controls travel in the slot of the Hamiltonian they replace (if they fit: always, but for
    // one- or two-segment pulses with many control operators, which are summed here instead)
    const size_t ctrl_opers = 16*size_t(hamiltonian ? 0 : n_c)*dd;
    const size_t ctrl_bytes = ctrl_opers + 8*size_t(hamiltonian ? 0 : n_c)*G;
    const bool on_device = !hamiltonian && ctrl_bytes <= 16*size_t(G)*dd;
    if (hamiltonian) {
        std::memcpy(hp + L.H, hamiltonian, 16*size_t(G)*dd);
    } else if (on_device) {
        std::memcpy(hp + L.H, c_opers, ctrl_opers);
        std::memcpy(hp + L.H + ctrl_opers, c_coeffs, 8*size_t(n_c)*G);
    } else {
        double* H = reinterpret_cast<double*>(hp + L.H);
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < dd; ++k) {
                double re = 0.0, im = 0.0;
                for (int i = 0; i < n_c; ++i) {
                    const double c = c_coeffs[size_t(i)*G + g];
                    re = std::fma(c, c_opers[2*(i*dd + k)], re);
                    im = std::fma(c, c_opers[2*(i*dd + k) + 1], im);
                }
                H[2*(g*dd + k)] = re;
                H[2*(g*dd + k) + 1] = im;
            }
    }
    std::memcpy(hp + L.dt, dt, 8*size_t(G));
    std::memcpy(hp + L.t, t, 8*size_t(G + 1));
    std::memcpy(hp + L.omega, omega, 8*size_t(W));
    std::memcpy(hp + L.basis, basis, 16*size_t(N)*dd);
    std::memcpy(hp + L.n_opers, n_opers, 16*size_t(A)*dd);
    std::memcpy(hp + L.n_coeffs, n_coeffs, 8*size_t(A)*G);
    if (spectrum) {
        double* hs = reinterpret_cast<double*>(hp + o_spec);
        if (spectrum_is_real) {
            for (size_t i = 0; i < s_rows*W; ++i) { hs[2*i] = spectrum[i]; hs[2*i + 1] = 0.0; }
        } else {
            std::memcpy(hs, spectrum, 16*s_rows*W);
        }
        std::memcpy(hp + o_idx, idx, sizeof(int32_t)*size_t(n_idx));
    }
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    // scratch of the pass from the shared arena (held only for the duration of this call)
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t wsb = ffk_pipeline_workspace_bytes(W, N, A, G, d, 0, 0);
    const size_t hsb = on_device ? align_up(16*size_t(G)*dd) : 0;
    const size_t iwsb = spectrum ? align_up(ffk_infidelity_workspace_bytes(W, n_idx, s_ndim)) : 0;
    void* ws;
    if (int rc = arena_reserve(wsb + hsb + iwsb, &ws)) return rc;
    StreamDrain drain{s};      // (the successful path has synchronised already: a no-op then)
    const auto clock1 = std::chrono::steady_clock::now();
    auto dptr = [dp](size_t off) { return reinterpret_cast<double*>(dp + off); };
    // copies in, kernels, copies out: on `s`, no synchronisation
    auto enqueue = [&]() -> int {
        const double* Hdev = dptr(L.H);
        if (on_device) {
            // the controls first, so that the sum runs while the rest of the inputs is still in flight
            FFK_HIP(hipMemcpyAsync(dp + L.H, hp + L.H, ctrl_bytes, hipMemcpyHostToDevice, s));
            cplx* Hsum = reinterpret_cast<cplx*>(static_cast<unsigned char*>(ws) + wsb);
            const size_t n = size_t(G)*dd;
            hipLaunchKernelGGL(assemble_hamiltonian_kernel, dim3(static_cast<unsigned>((n + 255)/256)), dim3(256),
                               0, s, reinterpret_cast<const cplx*>(dp + L.H),
                               reinterpret_cast<const double*>(dp + L.H + ctrl_opers), n_c, G, d*d, Hsum);
            FFK_HIP(hipGetLastError());
            FFK_HIP(hipMemcpyAsync(dp + L.dt, hp + L.dt, L.inputs_end - L.dt, hipMemcpyHostToDevice, s));
            Hdev = reinterpret_cast<const double*>(Hsum);
        } else {
            FFK_HIP(hipMemcpyAsync(dp, hp, L.inputs_end, hipMemcpyHostToDevice, s));
        }
        if (int rc = ffk_pipeline_dev(Hdev, dptr(L.dt), dptr(L.t), G, d, dptr(L.omega), W,
                                      dptr(L.basis), N, dptr(L.n_opers), A, dptr(L.n_coeffs), nullptr, 0,
                                      nullptr, 0, dptr(L.D), dptr(L.V), dptr(L.Q), dptr(L.R), dptr(L.F),
                                      nullptr, ws, wsb, s))
            return rc;
        if (int rc = ffk_eigensolver_status_dev(ws, wsb, G, d, reinterpret_cast<int32_t*>(dp + L.status), s))
            return rc;
        if (spectrum)
            if (int rc = ffk_infidelity_dev(dptr(L.F), A, W, reinterpret_cast<const double*>(hp + o_spec), s_ndim,
                                            dptr(L.omega), reinterpret_cast<const int32_t*>(hp + o_idx), n_idx,
                                            d_inf, reinterpret_cast<double*>(hp + o_out),
                                            static_cast<unsigned char*>(ws) + wsb + hsb, iwsb, s))
                return rc;
        FFK_HIP(hipMemcpyAsync(hp + L.D, dp + L.D, L.outputs_end - L.D, hipMemcpyDeviceToHost, s));
        return FFK_OK;
    };
    const ResidentGraphKey key{dev, G, d, W, N, A, hamiltonian ? 0 : n_c, on_device ? 1 : 0,
                               spectrum ? s_ndim : 0, spectrum ? n_idx : 0, spectrum ? d_inf : 0, dp, hp, ws, s,
                               g_knob_epoch.load()};
    ResidentGraph* hit = nullptr;
    ResidentGraph* victim = &g_resident_graphs[0];
    if (resident_graphs_enabled()) {
        for (ResidentGraph& e : g_resident_graphs) {
            if (e.exec && e.key == key) hit = &e;
            if (e.used < victim->used) victim = &e;
        }
    }
    bool enqueued = false;
    if (hit) {
        if (hipGraphLaunch(hit->exec, s) == hipSuccess) {
            hit->used = ++g_resident_graph_clock;
            g_stats = hit->stats;
            enqueued = true;
        } else {
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(hit->exec);
            hit->exec = nullptr;
            hit->used = 0;
        }
    } else if (resident_graphs_enabled() && resident_key_seen_before(key) &&
               hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
        // second pass of this shape on these blocks: capture it, then launch the capture
        const int rc = enqueue();
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc != FFK_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            return rc;
        }
        hipGraphExec_t exec = nullptr;
        if (ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess &&
            hipGraphLaunch(exec, s) == hipSuccess) {
            if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
            victim->key = key;
            victim->exec = exec;
            victim->stats = g_stats;
            victim->used = ++g_resident_graph_clock;
            enqueued = true;
        } else {
            if (exec) (void)hipGraphExecDestroy(exec);
            (void)hipGetLastError();
        }
        if (graph) (void)hipGraphDestroy(graph);
    } else {
        (void)hipGetLastError();
    }
    if (!enqueued)
        if (int rc = enqueue()) return rc;
    const auto clock2 = std::chrono::steady_clock::now();
    FFK_HIP(hipStreamSynchronize(s));
    const auto clock3 = std::chrono::steady_clock::now();
    r->t_stage = std::chrono::duration<double>(clock1 - clock0).count();
    r->t_enqueue = std::chrono::duration<double>(clock2 - clock1).count();
    r->t_wait = std::chrono::duration<double>(clock3 - clock2).count();
    const int32_t failed = *reinterpret_cast<const int32_t*>(hp + L.status);
    if (failed != 0)
        return fail(FFK_ENOCONV, "Jacobi eigensolver did not converge for %d segment(s)", int(failed));
    *eigvals = reinterpret_cast<double*>(hp + L.D);
    *eigvecs = reinterpret_cast<double*>(hp + L.V);
    *propagators = reinterpret_cast<double*>(hp + L.Q);
    *filter_function = reinterpret_cast<double*>(hp + L.F);
    if (spectrum) std::memcpy(infid, hp + o_out, 8*n_out);
    r->valid = true;
    return FFK_OK;
}

}  // namespace

extern "C" {

int ffk_resident_filter_function_infidelity(ffk_resident* r, const double* c_opers, int n_cops,
                                            const double* c_coeffs, const double* dt, const double* t, int G,
                                            int d, const double* omega, int W, const double* basis, int N,
                                            const double* n_opers, int A, const double* n_coeffs,
                                            const double* spectrum, int s_ndim, int spectrum_is_real,
                                            const int32_t* idx, int n_idx, int d_infidelity, double** eigvals,
                                            double** eigvecs, double** propagators, double** filter_function,
                                            double* infidelity) {
    FFK_REQUIRE(spectrum && idx && infidelity, "NULL argument");
    return resident_pass(r, nullptr, c_opers, n_cops, c_coeffs, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function, spectrum, s_ndim,
                         spectrum_is_real, idx, n_idx, d_infidelity, infidelity);
}

int ffk_resident_filter_function(ffk_resident* r, const double* hamiltonian, const double* dt,
                                 const double* t, int G, int d, const double* omega, int W,
                                 const double* basis, int N, const double* n_opers, int A,
                                 const double* n_coeffs, double** eigvals, double** eigvecs,
                                 double** propagators, double** filter_function) {
    FFK_REQUIRE(hamiltonian, "NULL argument");
    return resident_pass(r, hamiltonian, nullptr, 0, nullptr, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function);
}

int ffk_resident_filter_function_from_controls(ffk_resident* r, const double* c_opers, int n_cops,
                                               const double* c_coeffs, const double* dt,
                                               const double* t, int G, int d, const double* omega,
                                               int W, const double* basis, int N,
                                               const double* n_opers, int A, const double* n_coeffs,
                                               double** eigvals, double** eigvecs,
                                               double** propagators, double** filter_function) {
    FFK_REQUIRE(c_opers && c_coeffs && n_cops >= 1, "NULL or empty control Hamiltonian");
    return resident_pass(r, nullptr, c_opers, n_cops, c_coeffs, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function);
}

int ffk_resident_timing(ffk_resident* r, double* seconds) {
    FFK_REQUIRE(r && seconds, "NULL argument");
    seconds[0] = r->t_stage;
    seconds[1] = r->t_enqueue;
    seconds[2] = r->t_wait;
    return FFK_OK;
}

namespace {
int on_owning_device(const ffk_resident* r) {
    int dev = -1;
    FFK_HIP(hipGetDevice(&dev));
    FFK_REQUIRE(dev == r->device, "resident result lives on device %d, current device is %d",
                r->device, dev);
    return FFK_OK;
}
}  // namespace

}  // extern "C"


extern "C" {

// ffk_concatenate_sequence for distinct pulses whose control matrices are still resident (every one
// evaluated by ffk_resident_filter_function* on the same frequency grid): the table is assembled
// by device-to-device copies, the total propagators come from the handles' host blocks, the total
// phase factors exp(i omega tau_k) are formed on the device -- per call only index, basis and tau
// cross PCIe.  With `result` (which = 0, filter_function wanted) the summed control matrix, its filter
// function and the grid STAY in that handle (control_matrix may then be NULL): the new pulse is as
// resident as its parts -- ffk_resident_control_matrix / _infidelity serve it, and it can be an
// input of the next concatenation.
int ffk_concatenate_sequence_resident(ffk_resident* const* pulses, const double* tau,
                                      const int32_t* index, const double* basis, int hermitian_basis,
                                      int T, int G, int which, double* control_matrix,
                                      double* total_propagator, double* propagators_liouville,
                                      double* filter_function, ffk_resident* result) {
    FFK_REQUIRE(pulses && tau && index && basis && total_propagator, "NULL argument");
    FFK_REQUIRE(control_matrix || result, "NULL argument");
    FFK_REQUIRE(!result || (which == 0 && filter_function), "a resident result holds the summed control "
                "matrix and its filter function");
    for (int k = 0; result && k < T; ++k) FFK_REQUIRE(pulses[k] != result, "result must not be an input");
    FFK_REQUIRE(T >= 1 && T <= 65535 && G >= 1, "empty or oversized axis: T=%d G=%d", T, G);
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    FFK_REQUIRE(!filter_function || which == 0, "the filter function needs the summed control matrix");
    for (int k = 0; k < T; ++k) FFK_REQUIRE(pulses[k] && pulses[k]->valid, "pulse %d has no resident result", k);
    const ffk_resident* first = pulses[0];
    const int d = first->d, A = first->A, N = first->N, W = first->W;
    for (int k = 0; k < T; ++k) {
        const ffk_resident* r = pulses[k];
        FFK_REQUIRE(r->d == d && r->A == A && r->N == N && r->W == W && r->device == first->device,
                    "pulse %d: shape (d=%d, A=%d, N=%d, W=%d) or device differs from pulse 0", k, r->d,
                    r->A, r->N, r->W);
    }
    for (int g = 0; g < G; ++g)
        FFK_REQUIRE(index[g] >= 0 && index[g] < T, "index[%d] = %d outside [0, %d)", g, index[g], T);
    if (int rc = on_owning_device(first)) return rc;
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nU = 16*size_t(T)*dd, nP = 16*size_t(T)*W, nR1 = 16*size_t(A)*N*W, nI = 4*size_t(G);
    const size_t nB = 16*size_t(N)*dd, nT = 8*size_t(T), nX = 8*size_t(T);
    // host staging (propagators | tau | index | basis | pointers to the resident control matrices) in
    // one pinned block, one H2D.  The control matrices are read where they lie (round 3: assembling
    // a contiguous table cost T device-to-device copies per call, 24 of ~0.5 MB at config 3)
    const size_t oU = 0, oT = oU + align_up(nU), oI = oT + align_up(nT), oB = oI + align_up(nI);
    const size_t oX = oB + align_up(nB);
    const size_t stage = oX + align_up(nX);
    Block pin = {nullptr, 0, -1};
    if (int rc = g_pin_pool.take(stage, first->device, &pin)) return rc;
    unsigned char* hp = static_cast<unsigned char*>(pin.ptr);
    for (int k = 0; k < T; ++k) {
        const ffk_resident* r = pulses[k];
        const unsigned char* q = static_cast<const unsigned char*>(r->pin.ptr) + r->L.Q + 16*size_t(r->G)*dd;
        std::memcpy(hp + oU + 16*size_t(k)*dd, q, 16*dd);         // Q[-1]: the pulse's total propagator
    }
    std::memcpy(hp + oT, tau, nT);
    std::memcpy(hp + oI, index, nI);
    std::memcpy(hp + oB, basis, nB);
    for (int k = 0; k < T; ++k) {
        const unsigned char* rk = static_cast<const unsigned char*>(pulses[k]->dev.ptr) + pulses[k]->L.R;
        std::memcpy(hp + oX + 8*size_t(k), &rk, 8);
    }
    (void)nR1;
    void* base;
    int rc = arena_reserve(stage + align_up(nP) +
                           sequence_scratch_bytes(G, d, A, N, W, which, hermitian_basis != 0,
                                                  filter_function != nullptr), &base);
    if (rc) { g_pin_pool.give(pin); return rc; }
    Bump a(base, g_arena.size);
    unsigned char* dS = a.take<unsigned char>(stage);
    double* dP = a.take<double>(nP/8);
    // a result handle takes the layout of a one-segment pass: R, F and the grid in its device block,
    // (identity, total propagator) where the propagators of a pass sit in its host block
    ResidentLayout RL = {};
    double* keep_R = nullptr;
    double* keep_F = nullptr;
    if (result) {
        result->valid = false;
        RL = resident_layout(1, d, W, N, A);
        if (result->device != first->device || result->dev.size < RL.end || result->pin.size < RL.outputs_end) {
            g_dev_pool.give(result->dev);
            g_pin_pool.give(result->pin);
            result->dev = result->pin = Block{nullptr, 0, -1};
            rc = g_dev_pool.take(RL.end, first->device, &result->dev);
            if (!rc) rc = g_pin_pool.take(RL.outputs_end, first->device, &result->pin);
            if (rc) { g_pin_pool.give(pin); return rc; }
            result->device = first->device;
        }
        result->G = 1; result->d = d; result->W = W; result->N = N; result->A = A; result->L = RL;
        unsigned char* rp = static_cast<unsigned char*>(result->dev.ptr);
        keep_R = reinterpret_cast<double*>(rp + RL.R);
        keep_F = reinterpret_cast<double*>(rp + RL.F);
    }
    auto run = [&]() -> int {
        StreamDrain drain{s};
        FFK_HIP(hipMemcpyAsync(dS, hp, stage, hipMemcpyHostToDevice, s));
        const double* dOmega = reinterpret_cast<const double*>(
            static_cast<const unsigned char*>(first->dev.ptr) + first->L.omega);
        double* omega_copy = result ? reinterpret_cast<double*>(static_cast<unsigned char*>(result->dev.ptr) + RL.omega)
                                    : nullptr;
        if (int rc2 = sequence_on_device(reinterpret_cast<const double*>(dS + oU), dP, nullptr,
                                         reinterpret_cast<const int32_t*>(dS + oI),
                                         reinterpret_cast<const double*>(dS + oB), hermitian_basis, T, G, d,
                                         A, N, W, which, a, control_matrix, total_propagator,
                                         propagators_liouville, filter_function, s, keep_R, keep_F,
                                         reinterpret_cast<const cplx* const*>(dS + oX),
                                         reinterpret_cast<const double*>(dS + oT), dOmega, omega_copy))
            return rc2;
        FFK_HIP(hipStreamSynchronize(s));
        return FFK_OK;
    };
    rc = run();
    g_pin_pool.give(pin);
    if (!rc && result) {
        double* q = reinterpret_cast<double*>(static_cast<unsigned char*>(result->pin.ptr) + RL.Q);
        for (size_t e = 0; e < dd; ++e) {
            q[2*e] = (e / d == e % d) ? 1.0 : 0.0;
            q[2*e + 1] = 0.0;
        }
        std::memcpy(q + 2*dd, total_propagator, 16*dd);
        result->t_stage = result->t_enqueue = result->t_wait = 0.0;
        result->valid = true;
    }
    return rc;
}

int ffk_resident_control_matrix(ffk_resident* r, double* control_matrix) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    FFK_REQUIRE(control_matrix, "NULL argument");
    if (int rc = on_owning_device(r)) return rc;
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    const unsigned char* dp = static_cast<const unsigned char*>(r->dev.ptr);
    FFK_HIP(hipMemcpyAsync(control_matrix, dp + r->L.R, 16*size_t(r->A)*r->N*r->W,
                           hipMemcpyDeviceToHost, s));
    FFK_HIP(hipStreamSynchronize(s));
    return FFK_OK;
}

int ffk_resident_control_matrix_dev(ffk_resident* r, const double** control_matrix,
                                    const double** filter_function, const double** omega) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    const unsigned char* dp = static_cast<const unsigned char*>(r->dev.ptr);
    if (control_matrix) *control_matrix = reinterpret_cast<const double*>(dp + r->L.R);
    if (filter_function) *filter_function = reinterpret_cast<const double*>(dp + r->L.F);
    if (omega) *omega = reinterpret_cast<const double*>(dp + r->L.omega);
    return FFK_OK;
}

int ffk_resident_infidelity(ffk_resident* r, const double* spectrum, int s_ndim, int spectrum_is_real,
                            const int32_t* idx, int n_idx, int d, double* infid) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    FFK_REQUIRE(spectrum && idx && infid, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3 && n_idx >= 1 && n_idx <= r->A && d >= 1, "bad spectrum arguments");
    if (int rc = on_owning_device(r)) return rc;
    const int W = r->W, A = r->A;
    const size_t rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? size_t(n_idx) : size_t(n_idx)*n_idx);
    const size_t n_out = s_ndim == 3 ? size_t(n_idx)*n_idx : size_t(n_idx);
    unsigned char* hp = static_cast<unsigned char*>(r->pin.ptr);
    unsigned char* dp = static_cast<unsigned char*>(r->dev.ptr);
    const ResidentLayout& L = r->L;
    // stage spectrum (as c128), idx and the result in the pinned input region (free after the pass).
    // Round 3: the kernel reads spectrum and idx FROM that pinned block and writes the integrals INTO
    // it (pinned host memory is mapped into the device's address space): one launch and one
    // synchronisation instead of two H2D copies, the launch, a D2H copy into pageable memory and
    // the synchronisation -- the spectrum is read once (64 KB over PCIe at config 2)
    const size_t s_bytes = 16*rows*W;
    const size_t o_idx = align_up(s_bytes), o_out = o_idx + align_up(sizeof(int32_t)*size_t(n_idx));
    const size_t stage = o_out + align_up(8*n_out);
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    const bool fits = stage <= L.inputs_end;
    Block extra = {nullptr, 0, -1};
    unsigned char* stage_ptr = hp;
    if (!fits) {
        if (int rc = g_pin_pool.take(stage, r->device, &extra)) return rc;
        stage_ptr = static_cast<unsigned char*>(extra.ptr);
    }
    double* hs = reinterpret_cast<double*>(stage_ptr);
    if (spectrum_is_real) {
        for (size_t i = 0; i < rows*W; ++i) { hs[2*i] = spectrum[i]; hs[2*i + 1] = 0.0; }
    } else {
        std::memcpy(hs, spectrum, s_bytes);
    }
    std::memcpy(stage_ptr + o_idx, idx, sizeof(int32_t)*size_t(n_idx));
    int rc = FFK_OK;
    {
        std::lock_guard<std::mutex> lock(g_arena.mu);
        const size_t iwsb = ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
        void* iws;
        rc = arena_reserve(iwsb, &iws);
        if (!rc)
            rc = ffk_infidelity_dev(reinterpret_cast<const double*>(dp + L.F), A, W,
                                    reinterpret_cast<const double*>(stage_ptr), s_ndim,
                                    reinterpret_cast<const double*>(dp + L.omega),
                                    reinterpret_cast<const int32_t*>(stage_ptr + o_idx), n_idx, d,
                                    reinterpret_cast<double*>(stage_ptr + o_out), iws, iwsb, s);
        if (!rc) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) rc = fail(FFK_EHIP, "infidelity failed: %s", hipGetErrorString(e));
            else std::memcpy(infid, stage_ptr + o_out, 8*n_out);
        }
    }
    g_pin_pool.give(extra);
    return rc;
}

}  // extern "C"

#if defined(FFK_HOST_SANITIZE)
// ---------------------------------------------------------------------------------------------
// Host-logic self test for the sanitizer variant: arena growth, block-pool reuse and eviction, every
// workspace layout sliced with the size its *_workspace_bytes query reports and each slice written
// end to end (an overrun of a slice or of the reservation is a heap-buffer-overflow under ASan).
// ---------------------------------------------------------------------------------------------
#include <cstring>
#include <random>
extern "C" int ffk_selftest_host(int rounds, unsigned seed, char* report, int report_len) {
    if (rounds < 0) {
        // negative control: a deliberate one-byte overrun, which the sanitizer must report
        volatile unsigned char* p = static_cast<unsigned char*>(std::malloc(16));
        p[16] = 1;
        std::free(const_cast<unsigned char*>(p));
        return 0;
    }
    std::mt19937 rng(seed);
    auto pick = [&](int lo, int hi) { return lo + static_cast<int>(rng() % static_cast<unsigned>(hi - lo + 1)); };
    long checked = 0;
    auto touch = [&](void* p, size_t n) {
        if (p && n) {
            std::memset(p, 0xA5, n);
            ++checked;
        }
    };
    for (int r = 0; r < rounds; ++r) {
        const int d = pick(2, FFK_MAX_D_TEMPLATED), G = pick(1, 300), A = pick(1, 9), W = pick(1, 700);
        const int N = pick(1, d*d);
        // (a) arena: reserve, write all of it, grow, shrink requests
        void* base = nullptr;
        const size_t want = size_t(pick(1, 1 << 20))*pick(1, 8);
        {
            std::lock_guard<std::mutex> lock(g_arena.mu);
            if (arena_reserve(want, &base) != FFK_OK) return -1;
            touch(base, g_arena.size);
        }
        // (b) control-matrix workspace: the slices of ffk_control_matrix_dev
        {
            const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, 0);
            const size_t bytes = ffk_control_matrix_workspace_bytes(W, N, A, G, d);
            if (bytes < ctrl_ws_bytes(W, N, A, G, d, geo.chunks)) return -2;
            void* ws = std::malloc(bytes);
            Bump b(ws, bytes);
            double* segtab = b.take<double>(size_t(G)*ffk::seg_stride(d));
            cplx* Tc = b.take<cplx>(size_t(G)*d*d);
            cplx* ops = b.take<cplx>(size_t(G)*(1 + A)*d*d);
            cplx* Ypart = b.take<cplx>(size_t(geo.chunks)*A*d*d*W);
            cplx* Bt = b.take<cplx>(size_t(A)*d*d*W);
            void* ews = b.take<unsigned char>(ffk::expand_workspace_bytes(N, d));
            if (!segtab || !Tc || !ops || !Ypart || !Bt || !ews) { std::free(ws); return -3; }
            touch(segtab, sizeof(double)*size_t(G)*ffk::seg_stride(d));
            touch(Tc, sizeof(cplx)*size_t(G)*d*d);
            touch(ops, sizeof(cplx)*size_t(G)*(1 + A)*d*d);
            touch(Ypart, sizeof(cplx)*size_t(geo.chunks)*A*d*d*W);
            touch(Bt, sizeof(cplx)*size_t(A)*d*d*W);
            touch(ews, ffk::expand_workspace_bytes(N, d));
            int *nnz, *rows;
            cplx* vals;
            ffk::expand_workspace_slices(ews, N, d, &nnz, &rows, &vals);
            touch(nnz, sizeof(int)*N);
            touch(rows, sizeof(int)*size_t(N)*d*d);
            touch(vals, sizeof(cplx)*size_t(N)*d*d);
            std::free(ws);
        }
        // (c) diagonalize + pipeline workspaces
        {
            const size_t dwsb = ffk_diagonalize_workspace_bytes(G, d);
            void* ws = std::malloc(dwsb);
            const DiagWs w = slice_diag_ws(ws, dwsb, G, d);
            touch(w.status, sizeof(int)*G);
            touch(w.seg_prop, sizeof(cplx)*size_t(G)*d*d);
            touch(w.qloc, sizeof(cplx)*size_t(G + 1)*d*d);
            std::free(ws);
            const int n_idx = pick(1, A), s_ndim = pick(1, 3);
            const size_t pb = ffk_pipeline_workspace_bytes(W, N, A, G, d, n_idx, s_ndim);
            if (pb < dwsb + ffk_control_matrix_workspace_bytes(W, N, A, G, d)) return -4;
            void* pws = std::malloc(pb);
            Bump b(pws, pb);
            void* a1 = b.take<unsigned char>(dwsb);
            void* a2 = b.take<unsigned char>(ffk_control_matrix_workspace_bytes(W, N, A, G, d));
            double* D = b.take<double>(size_t(G)*d);
            double* V = b.take<double>(2*size_t(G)*d*d);
            double* Q = b.take<double>(2*size_t(G + 1)*d*d);
            double* R = b.take<double>(2*size_t(A)*N*W);
            double* F = b.take<double>(2*size_t(A)*A*W);
            void* iws = b.take<unsigned char>(ffk_infidelity_workspace_bytes(W, n_idx, s_ndim));
            if (!a1 || !a2 || !D || !V || !Q || !R || !F || !iws) { std::free(pws); return -5; }
            touch(F, 16*size_t(A)*A*W);
            touch(iws, ffk_infidelity_workspace_bytes(W, n_idx, s_ndim));
            std::free(pws);
        }
        // (d) sequence scratch of the concatenation entry points
        {
            const int T = pick(1, 30), Gs = pick(1, 1200), which = pick(0, 1);
            const bool herm = pick(0, 1) != 0, wantF = which == 0 && pick(0, 1);
            const int d2 = pick(2, 4), N2 = d2*d2, A2 = pick(1, 3), W2 = pick(1, 300);
            (void)T;
            const size_t sb = sequence_scratch_bytes(Gs, d2, A2, N2, W2, which, herm, wantF);
            void* ws = std::malloc(sb);
            Bump a(ws, sb);
            const size_t dd = size_t(d2)*d2;
            const int nl = Gs > 1 ? Gs - 1 : 1;
            cplx* dSeq = a.take<cplx>(size_t(Gs)*dd);
            cplx* dQ = a.take<cplx>(size_t(Gs + 1)*dd);
            double* dL = a.take<double>((herm ? 1 : 2)*size_t(nl)*N2*N2);
            double* dO = a.take<double>(2*(which ? size_t(Gs) : 1)*A2*N2*W2);
            void* w1 = a.take<unsigned char>(ffk::scan_workspace_bytes(Gs, d2));
            void* w2 = a.take<unsigned char>(ffk::liouville_workspace_bytes(nl, d2, N2));
            void* w3 = a.take<unsigned char>(ffk_control_matrix_from_atomic_workspace_bytes(Gs, A2, N2, W2));
            double* dF = wantF ? a.take<double>(2*size_t(A2)*A2*W2) : nullptr;
            if (!dSeq || !dQ || !dL || !dO || !w1 || !w2 || !w3 || (wantF && !dF)) { std::free(ws); return -6; }
            touch(dSeq, 16*size_t(Gs)*dd);
            touch(dQ, 16*size_t(Gs + 1)*dd);
            touch(dL, (herm ? 8 : 16)*size_t(nl)*N2*N2);
            touch(dO, 16*(which ? size_t(Gs) : 1)*A2*N2*W2);
            touch(w3, ffk_control_matrix_from_atomic_workspace_bytes(Gs, A2, N2, W2));
            if (dF) touch(dF, 16*size_t(A2)*A2*W2);
            std::free(ws);
        }
        // (e) block pools: take / write / give in random order, past the eviction bound
        {
            std::vector<Block> held;
            for (int k = 0; k < 40; ++k) {
                if (held.empty() || pick(0, 2)) {
                    Block b = {nullptr, 0, -1};
                    BlockPool& pool = pick(0, 1) ? g_dev_pool : g_pin_pool;
                    const size_t bytes = size_t(pick(1, 1 << 18));
                    if (pool.take(bytes, 0, &b) != FFK_OK || b.size < bytes) return -7;
                    touch(b.ptr, b.size);
                    b.device = (&pool == &g_dev_pool) ? 0 : -1;
                    held.push_back(b);
                } else {
                    const int i = pick(0, int(held.size()) - 1);
                    (held[i].device == 0 ? g_dev_pool : g_pin_pool).give(held[i]);
                    held.erase(held.begin() + i);
                }
            }
            for (const Block& b : held) (b.device == 0 ? g_dev_pool : g_pin_pool).give(b);
        }
        // (f) resident layout: offsets ascending, inside the block
        {
            const ResidentLayout RL = resident_layout(G, d, W, N, A);
            if (!(RL.inputs_end <= RL.D && RL.outputs_end <= RL.R && RL.R < RL.end && RL.F + 16*size_t(A)*A*W <= RL.end))
                return -8;
            ffk_resident* h = nullptr;
            if (ffk_resident_create(&h) != FFK_OK) return -9;
            if (ffk_resident_destroy(h) != FFK_OK) return -10;
        }
    }
    if (g_dev_pool.release() != FFK_OK || g_pin_pool.release() != FFK_OK) return -11;
    if (ffk_release_arena() != FFK_OK) return -12;
    if (report && report_len > 0)
        snprintf(report, report_len, "%d rounds, %ld regions written end to end, pools and arena released", rounds, checked);
    return 0;
}
#endif
