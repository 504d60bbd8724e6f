// ffk_api.hip -- the extern "C" surface declared in include/ffk.h.
//
// *_dev entry points only slice the caller's workspace and enqueue kernels.  The host-pointer
// entry points stage through a process-wide, grow-only device arena and synchronise before
// returning, so that the NumPy-facing front-end has exactly the reference's call semantics
// (borrowed inputs, freshly written outputs).
#include "ffk_api_common.h"

namespace ffk_api {

thread_local std::string g_error;
thread_local ffk_stats g_stats = {};
std::atomic<unsigned long long> g_knob_epoch{0};
int g_forced_chunks = 0;
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr, g_ev_gate = nullptr;
Arena g_arena;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

// ---- sticky fault words of the kernels that wait on flags (ctrl_pq.hip) ------------------------
// One block of ints in mapped, portable pinned host memory per process, one word per host thread (slots are handed
// out round robin on a thread's first use): a kernel whose bounded flag wait runs out stores a code in the word its
// launcher passed it as an ARGUMENT (so the word is right on whichever device the launch goes to; its results are
// garbage from then on), the host reads it with a plain load after the synchronisation it does anyway -- no copy, no
// extra launch; nothing is written on the good path.  A thread sees only the faults of launches it enqueued itself.
namespace {
constexpr int kFaultWords = 256;
std::once_flag g_fault_once;
int* g_fault_host = nullptr;      // kFaultWords ints
int* g_fault_dev = nullptr;       // the same block as the devices see it
std::atomic<unsigned> g_fault_next{0};
thread_local int g_fault_slot = -1;
int fault_slot() {
    if (g_fault_slot < 0) g_fault_slot = static_cast<int>(g_fault_next.fetch_add(1) % kFaultWords);
    return g_fault_slot;
}
}  // namespace
}  // namespace ffk_api
namespace ffk {
int* kernel_fault_word() {
    std::call_once(ffk_api::g_fault_once, [] {
        void* h = nullptr;
        const size_t bytes = sizeof(int)*ffk_api::kFaultWords;
#if defined(FFK_HOST_SANITIZE)
        if (hipHostMalloc(&h, bytes, 0) != hipSuccess) return;
        ffk_api::g_fault_dev = static_cast<int*>(h);
#else
        if (hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return;
        void* dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, h, 0) != hipSuccess) {
            (void)hipHostFree(h);
            return;
        }
        ffk_api::g_fault_dev = static_cast<int*>(dptr);
#endif
        for (int i = 0; i < ffk_api::kFaultWords; ++i) static_cast<volatile int*>(h)[i] = 0;
        ffk_api::g_fault_host = static_cast<int*>(h);
    });
    return ffk_api::g_fault_dev ? ffk_api::g_fault_dev + ffk_api::fault_slot() : nullptr;
}
}  // namespace ffk
namespace ffk_api {
int kernel_fault_peek(bool clear) {
    if (!g_fault_host) return 0;
    volatile int* word = g_fault_host + fault_slot();
    const int w = *word;
    if (w != 0 && clear) *word = 0;
    return w;
}
int kernel_fault_status() {
    const int w = kernel_fault_peek(true);
    if (w == 0) return FFK_OK;
    return fail(FFK_EKERNEL,
                "a flag wait inside the d = 4 accumulate kernel ran out (code %d): the launch's results are "
                "invalid", w);
}
int kernel_fault_stale() {
    const int w = kernel_fault_peek(true);
    if (w == 0) return FFK_OK;
    return fail(FFK_EKERNEL,
                "an EARLIER asynchronous launch of this thread reported a kernel fault (code %d) that was never "
                "collected with ffk_kernel_fault_status: its results are invalid; this call has not run", w);
}
int kernel_fault_slot_for_selftest() { return fault_slot(); }

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
    b += align_up(sizeof(cplx)*ffk::wfold_elems(d, G, A, W, chunks));     // folded W_a / padded operands (ffk_internal.h wfold), LAST
    return b;
}

int max_chunks_for(int W, int A, int G, int d) {
    // upper bound of what accumulate_geometry may choose, for workspace sizing
    const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, g_forced_chunks);
    return geo.chunks;
}

double accumulate_flops(int W, int A, int G, int d) {
    // a dimension that runs padded on the next specialised kernel executes that kernel's flops (ffk_internal.h:
    // padded_dimension; the callers of this function are the ones that hand the launch its scratch)
    if (ffk::padded_launch_pays(d, G, W, A) && std::getenv("FFK_NO_PADDED_DIMENSIONS") == nullptr)
        return accumulate_flops(W, A, G, ffk::padded_dimension(d));
    // FMA-counted real flops the accumulate kernels EXECUTE per (segment, frequency)
    // (DESIGN.md section 3).
    // d = 4: the tile per group of <= 3 operators: 13 entries x 10 (x, addition theorem 3, reciprocal 5, product
    // 1) + 62 (two sincos and psi).
    if (d == 4 && ffk::pq_accumulate_supported(d, A)) {
        // ctrl_pq.hip (round 5; the default): per operator the first product 16 x (2 mul + 6 fma) = 224,
        // zr + zi 16, the second product as THREE real 4 x 4 x 4 matrix products (Gauss) 3 x 128 = 384 = 624;
        // per group of operators (ctrl_pq.hip::pq_accumulate_groups) the tile 13 x 10 + 62 = 192 and c = psi conj(T),
        // cr + ci: 16 x 7 = 112.
        // The fold of W_a (6 per operator for Bbar times e^{ib} T, 6 per block for that product) is NOT counted:
        // this entry point has the prologue kernel do it once per segment (ffk_internal.h wfold), the kernel
        // copies it.  (Until the last change of round 5 the kernel folded it per frequency block: 630 nc + 310.)
        const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, g_forced_chunks);
        if (geo.pc) {
            // round 6: groups of three, of two and (A = 1) of one operator, each executing exactly its own
            const ffk::PqGroups gr = ffk::pq_accumulate_groups(A);
            const double per_gw = 624.0*A + 304.0*(gr.n3 + gr.n2 + gr.n1);
            return per_gw*double(G)*double(W);
        }
    }
    // d = 2 (ctrl_d2.hip, round 6): the tile's three distinct entries as below (18 x 3 + 62 = 116), twelve complex
    // multiply-adds per operator with the folded operands (96), the fold itself -- a lane's share of a segment's
    // record: two triple products and an add -- 26.
    if (d == 2 && ffk::accumulate_geometry(W, A, G, d, g_forced_chunks).d2)
        return (96.0*A + 116.0 + 26.0)*double(G)*double(W);
    // d = 8 (ctrl_pcr.hip, round 4): per operator the first product real x complex 4 d^3 = 2048, psi P
    // 6 d^2 = 384, the second product complex 8 d^3 = 4096; per group of <= 3 operators the tile: 57 entries x 10 +
    // 62.  The fold of W' (one (m, n) per lane = per frequency: 9 complex products = 54 per operator) is NOT counted
    // since round 6: the prologue does it once per segment (ffk_internal.h wfold) and the kernel copies it by LDS-DMA
    // (until then 6582 A + 632 per group).
    if (d == 8 && ffk::pcr_accumulate_supported(d, A))
        return (6528.0*A + 632.0*((A + 2)/3))*double(G)*double(W);
    // other d: contraction (2 d^3 MAC + d^2 mul) complex per (g, w, a) = 16 d^3 + 6 d^2 flops; tile:
    // ffk_math.h::phased_integral_aa, 18 flops per distinct entry (rotation 6, addition theorem 3,
    // x 1, reciprocal 4, products 3 + 1), d(d-1)+1 entries per (g, w); two sincos and the phase: 62.
    // (Rounds 1-3 reported a direct-evaluation model, 55 flops per entry + 26, which the kernels
    // have not executed since the round-3 generator.)
    const double per_gw = (16.0*d*d*d + 6.0*d*d)*A + 18.0*(d*(d - 1) + 1) + 62.0;
    return per_gw*double(G)*double(W);
}

DiagWs slice_diag_ws(void* workspace, size_t bytes, int G, int d) {
    Bump ws(workspace, bytes);
    DiagWs out;
    out.status = ws.take<int>(G);            // eigensolver flags
    out.seg_prop = ws.take<cplx>(size_t(G)*d*d);         // (G, d, d)
    out.qloc = ws.take<cplx>(size_t(G + 1)*d*d);         // (G+1, d, d) chunk-local prefix products
    out.small = ws.take<unsigned char>(1);               // scan scratch or chunk totals
    return out;
}

}  // namespace ffk_api

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

int ffk_diagonalize_dev(const double* hamiltonian, const double* dt, int G, int d, double* eigvals,
                        double* eigvecs, double* propagators, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return diagonalize_dev_impl(hamiltonian, dt, G, d, eigvals, eigvecs, propagators, workspace, workspace_bytes,
                                stream, PassOptions{});
}
}  // extern "C"
namespace ffk_api {
int diagonalize_dev_impl(const double* hamiltonian, const double* dt, int G, int d, double* eigvals, double* eigvecs,
                         double* propagators, void* workspace, size_t workspace_bytes, void* stream,
                         const PassOptions& opt) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(G >= 1, "need at least one segment, got G=%d", G);
    FFK_REQUIRE(hamiltonian && dt && eigvals && eigvecs && propagators && workspace, "NULL argument");
    FFK_REQUIRE(workspace_bytes >= ffk_diagonalize_workspace_bytes(G, d), "workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DiagWs w = slice_diag_ws(workspace, workspace_bytes, G, d);
    const cplx* H = reinterpret_cast<const cplx*>(hamiltonian);
    FFK_HIP(ffk::launch_eigh_expm(H, dt, G, d, eigvals, reinterpret_cast<cplx*>(eigvecs), w.seg_prop,
                                  w.status, s, opt.eigh_fail_count));
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
}  // namespace ffk_api
extern "C" {

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
    PassOptions opt;
    return control_matrix_dev_impl(eigvals, eigvecs, propagators, omega, W, basis, N, n_opers, A, n_coeffs, dt, t, G,
                                   d, flags, control_matrix, noise_operators, workspace, workspace_bytes, stream,
                                   opt);
}
}  // extern "C"
namespace ffk_api {
int control_matrix_dev_impl(const double* eigvals, const double* eigvecs, const double* propagators,
                            const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                            const double* n_coeffs, const double* dt, const double* t, int G, int d, unsigned flags,
                            double* control_matrix, double* noise_operators, void* workspace,
                            size_t workspace_bytes, void* stream, PassOptions& opt) {
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
    // d = 4, 8: the prologue folds W_a once per segment for the accumulate kernel (handed to both launches)
    const size_t n_fold = ffk::wfold_elems(d, G, A, W, geo.chunks);
    cplx* wfold = n_fold ? ws.take<cplx>(n_fold) : nullptr;
    FFK_REQUIRE(!n_fold || wfold, "workspace too small");

    if (!(flags & FFK_INTERNAL_PROLOGUE_DONE)) {
        FFK_HIP(ffk::launch_prologue(eigvals, reinterpret_cast<const cplx*>(eigvecs),
                                     reinterpret_cast<const cplx*>(propagators),
                                     reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, G, d, A, segtab, Tc, ops,
                                     nullptr, nullptr, s, wfold));
    }
    if (g_ev_start && g_ev_stop) {
        // `gate`: an event of ANOTHER stream (the previous pass's accumulate kernel) that this
        // stream waits for first, so that start..stop spans this kernel's execution and not its
        // wait for the other pass's blocks to retire
        if (g_ev_gate) FFK_HIP(hipStreamWaitEvent(s, g_ev_gate, 0));
        FFK_HIP(hipEventRecord(g_ev_start, s));
    }
    const bool want_B = (flags & FFK_WANT_NOISE_OPERATORS) != 0;
    bool compacted = (flags & FFK_INTERNAL_COMPACT_DONE) != 0;
    // one segment chunk, only R wanted, basis lists ready: the accumulate kernel may expand Y itself
    // (ffk_internal.h::ExpandEpilogue) and write the control matrix instead of the partial sums
    ffk::ExpandEpilogue epilogue = {};
    if (compacted && control_matrix && !want_B && geo.chunks == 1) {
        int* nnz;
        int* rows;
        cplx* vals;
        ffk::expand_workspace_slices(ews, N, d, &nnz, &rows, &vals);
        epilogue = {nnz, rows, vals, N, reinterpret_cast<cplx*>(control_matrix)};
    }
    bool expanded = false;
    // (wfold: written by the prologue above, or by the fused front's with the same slicing)
    FFK_HIP(ffk::launch_accumulate(omega, W, segtab, ops, G, d, A, geo, Ypart, s, epilogue.R ? &epilogue : nullptr,
                                   &expanded, wfold));
    if (g_ev_start && g_ev_stop) FFK_HIP(hipEventRecord(g_ev_stop, s));
    const size_t slab = size_t(A)*d*d*W;
    const cplx* Bsum = Ypart;
    if (expanded) {
        // R is written; F (if wanted through opt.fuse_F) is left to the caller's filter-function launch
    } else if (compacted && control_matrix && !want_B && opt.fuse_F && ffk::expand_ff_supported(A, N)) {
        // only R and F are wanted and the basis lists are ready: chunk sum, expansion and F in one
        FFK_HIP(ffk::launch_expand_ff(Ypart, geo.chunks, slab, A, N, d, W,
                                      reinterpret_cast<cplx*>(control_matrix), opt.fuse_F, ews, s));
        opt.fuse_F_done = true;
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
    g_stats.grid_x = geo.mfma ? (W + 15)/16 : (geo.d2 ? (W + ffk::d2_accumulate_freqs_per_block() - 1)/ffk::d2_accumulate_freqs_per_block() : (W + 63)/64);
    g_stats.grid_y = geo.task_groups;
    g_stats.grid_z = geo.chunks;
    g_stats.block = geo.nwaves*geo.gsplit*64;
    g_stats.lds_bytes = geo.lds_bytes;
    return FFK_OK;
}
}  // namespace ffk_api
extern "C" {

int ffk_control_matrix(const double* eigvals, const double* eigvecs, const double* propagators,
                       const double* omega, int W, const double* basis, int N,
                       const double* n_opers, int A, const double* n_coeffs, const double* dt,
                       const double* t, int G, int d, unsigned flags, double* control_matrix,
                       double* noise_operators) {
    FFK_REQUIRE(d_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(eigvals && eigvecs && propagators && omega && n_opers && n_coeffs && dt && t,
                "NULL argument");
    if (int rc = kernel_fault_stale()) return rc;
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
    return kernel_fault_status();
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
    if (int rc = kernel_fault_stale()) return rc;
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
    return kernel_fault_status();
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
    return infidelity_dev_impl(filter_function, A, W, spectrum, s_ndim, omega, idx, n_idx, d, infid, workspace,
                               workspace_bytes, stream, PassOptions{});
}
}  // extern "C"
namespace ffk_api {
int infidelity_dev_impl(const double* filter_function, int A, int W, const double* spectrum, int s_ndim,
                        const double* omega, const int32_t* idx, int n_idx, int d, double* infid, void* workspace,
                        size_t workspace_bytes, void* stream, const PassOptions& opt) {
    FFK_REQUIRE(filter_function && spectrum && omega && idx && infid && workspace, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3, "Expected spectrum to have < 4 dimensions, not %d", s_ndim);
    FFK_REQUIRE(A >= 1 && W >= 1 && n_idx >= 1 && d >= 1, "empty axis");
    FFK_REQUIRE(workspace_bytes >= ffk_infidelity_workspace_bytes(W, n_idx, s_ndim), "workspace too small");
    FFK_HIP(ffk::launch_infidelity(reinterpret_cast<const cplx*>(filter_function), A, W,
                                   reinterpret_cast<const cplx*>(spectrum), s_ndim, omega, idx, n_idx,
                                   d, 0, infid, workspace, static_cast<hipStream_t>(stream),
                                   opt.infid_spectrum_on_host));
    return FFK_OK;
}
}  // namespace ffk_api
extern "C" {

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
    PassOptions opt;
    return pipeline_dev_impl(hamiltonian, dt, t, G, d, omega, W, basis, N, n_opers, A, n_coeffs, spectrum, s_ndim, idx,
                             n_idx, eigvals, eigvecs, propagators, control_matrix, filter_function, infid, workspace,
                             workspace_bytes, stream, opt);
}
}  // extern "C"
namespace ffk_api {
int pipeline_dev_impl(const double* hamiltonian, const double* dt, const double* t, int G, int d,
                      const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                      const double* n_coeffs, const double* spectrum, int s_ndim, const int32_t* idx, int n_idx,
                      double* eigvals, double* eigvecs, double* propagators, double* control_matrix,
                      double* filter_function, double* infid, void* workspace, size_t workspace_bytes,
                      void* stream, PassOptions& opt) {
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
        if (opt.eigh_controls.opers != nullptr && ffk::eigh_fail_count_supported(d))
            FFK_HIP(ffk::launch_eigh_expm_controls(opt.eigh_controls.opers, opt.eigh_controls.coeffs,
                                                   opt.eigh_controls.n_c, dt, G, d, D, reinterpret_cast<cplx*>(V),
                                                   w.seg_prop, w.status, s, opt.eigh_fail_count));
        else
            FFK_HIP(ffk::launch_eigh_expm(reinterpret_cast<const cplx*>(hamiltonian), dt, G, d, D,
                                          reinterpret_cast<cplx*>(V), w.seg_prop, w.status, s, opt.eigh_fail_count));
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
        const size_t n_fold = ffk::wfold_elems(d, G, A, W, geo.chunks);
        cplx* wfold = n_fold ? cw.take<cplx>(n_fold) : nullptr;
        FFK_HIP(ffk::launch_apply_prologue_compact(
            w.qloc, totals, G, d, reinterpret_cast<cplx*>(Q), D, reinterpret_cast<const cplx*>(V),
            reinterpret_cast<const cplx*>(n_opers), n_coeffs, dt, t, A, segtab, Tc, ops,
            reinterpret_cast<const cplx*>(basis), N, ews, s, wfold));
        cm_flags = FFK_INTERNAL_PROLOGUE_DONE | FFK_INTERNAL_COMPACT_DONE;
    } else {
        if (int rc = diagonalize_dev_impl(hamiltonian, dt, G, d, D, V, Q, dws, dwsb, stream, opt)) return rc;
    }
    opt.fuse_F = reinterpret_cast<cplx*>(F);
    opt.fuse_F_done = false;
    const int rc_cm = control_matrix_dev_impl(D, V, Q, omega, W, basis, N, n_opers, A, n_coeffs, dt, t, G, d,
                                              cm_flags, R, nullptr, cws, cwsb, stream, opt);
    opt.fuse_F = nullptr;
    if (rc_cm) return rc_cm;
    if (!opt.fuse_F_done)
        if (int rc = ffk_filter_function_dev(R, A, N, W, FFK_FF_FIDELITY, F, stream)) return rc;
    if (want_infid) {
        const size_t iwsb = ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
        void* iws = ws.take<unsigned char>(iwsb);
        FFK_REQUIRE(iws, "workspace too small");
        if (int rc = infidelity_dev_impl(F, A, W, spectrum, s_ndim, omega, idx, n_idx, d, infid, iws, iwsb, stream, opt))
            return rc;
    }
    return FFK_OK;
}
}  // namespace ffk_api
extern "C" {

// ---------------------------------------------------------------------------------------------
// fault word of the flag-passing kernels, for callers of the device-pointer flavour
int ffk_kernel_fault_status(int32_t* word, int clear) {
    FFK_REQUIRE(word, "word is NULL");
    *word = kernel_fault_peek(clear != 0);
    return FFK_OK;
}

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
}  // extern "C"
