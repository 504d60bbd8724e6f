// ffk_api_sequence.hip -- the extern "C" entry points of SURVEY section 8's widening rows: the
// concatenation rule (pulse_sequence.concatenate, f1) and decay amplitudes -> cumulant function ->
// matrix exponential (numeric.calculate_decay_amplitudes / calculate_cumulant_function /
// error_transfer_matrix, f2).  Shared helpers: ffk_api_common.h.
#include "ffk_api_common.h"

extern "C" {

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
namespace ffk_api {

// temporaries of one sequence run, in arena order
size_t sequence_scratch_bytes(int G, int d, int A, int N, int W, int which, bool hermitian, bool want_F) {
    const size_t dd = size_t(d)*d;
    const int nl = G > 1 ? G - 1 : 1;
    return align_up(16*size_t(G)*dd) + align_up(16*size_t(G + 1)*dd) +
           align_up((hermitian ? 8 : 16)*size_t(nl)*N*N) +
           align_up(which ? 16*size_t(G)*A*N*W : 16*size_t(A)*N*W) +
           align_up(ffk::scan_workspace_bytes(G, d)) + align_up(ffk::liouville_workspace_bytes(nl, d, N)) +
           align_up(ffk_control_matrix_from_atomic_workspace_bytes(G, A, N, W)) +
           (want_F ? align_up(16*size_t(A)*A*W) : 0) + align_up(16*size_t(G)*N*N);      // (last: the distinct pulses' representations, T <= G)
}

// gather -> prefix products -> Liouville representation -> table rule (-> F) on `s`, all operands
// already on the device; results to the host pointers (asynchronously: the caller synchronises)
int sequence_on_device(const double* dU, const double* dP, const double* dR, const int32_t* dI,
                       const double* dB, int hermitian_basis, int T, int G, int d, int A, int N, int W,
                       int which, Bump& a, double* control_matrix, double* total_propagator,
                       double* propagators_liouville, double* filter_function, hipStream_t s,
                       double* resident_R, double* resident_F, const cplx* const* dRtab,
                       const double* dTau, const double* dOmega, double* omega_copy) {
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
    // the T distinct pulses' own Liouville representations, for the rule kernel's backward recurrence (the fused front
    // launch writes them; T <= G)
    // (Hermitian bases only: the recurrence composes representations, L(U Q) = L(U) L(Q), which holds where
    // tr(X C_k) are X's expansion coefficients -- an orthonormal HERMITIAN basis; a non-Hermitian one keeps the walk
    // along the cumulative propagators: tests/test_gpu_parity.py::test_block_rule_kernel_with_a_non_hermitian_basis)
    double* dLp = (dTau && T <= G && !l_is_complex && ffk::sequence_front_supported(d, G, N))
                      ? a.take<double>(size_t(T)*N*N) : nullptr;
    if (resident_R) dO = resident_R;          // results that stay in a handle's device block
    if (resident_F) dF = resident_F;
    if (dTau && ffk::sequence_front_supported(d, G, N)) {
        // gather + running products + Liouville representations + total phases (+ grid copy): one launch
        FFK_HIP(ffk::launch_sequence_front(reinterpret_cast<const cplx*>(dU), dI, G, d,
                                           reinterpret_cast<const cplx*>(dB), N, l_is_complex, dQ, dL, dTau,
                                           dOmega, T, W, reinterpret_cast<cplx*>(const_cast<double*>(dP)),
                                           omega_copy, s, dLp));
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
    // (ffk_set_accumulate_events: the caller times this launch on the stream it runs on)
    if (g_ev_start && g_ev_stop) FFK_HIP(hipEventRecord(g_ev_start, s));
    FFK_HIP(ffk::launch_from_atomic(reinterpret_cast<const cplx*>(dP), reinterpret_cast<const cplx*>(dR), dI,
                                    dL, l_is_complex, G, A, N, W, which, reinterpret_cast<cplx*>(dO), watom,
                                    s, dRtab, reinterpret_cast<cplx*>(dF), T, dLp));
    if (g_ev_start && g_ev_stop) FFK_HIP(hipEventRecord(g_ev_stop, s));
    if (dF) FFK_HIP(hipMemcpyAsync(filter_function, dF, nF, hipMemcpyDeviceToHost, s));
    if (control_matrix) FFK_HIP(hipMemcpyAsync(control_matrix, dO, nO, hipMemcpyDeviceToHost, s));
    FFK_HIP(hipMemcpyAsync(total_propagator, dQ + size_t(G)*dd, 16*dd, hipMemcpyDeviceToHost, s));
    if (propagators_liouville && G > 1)
        FFK_HIP(hipMemcpyAsync(propagators_liouville, dL, (l_is_complex ? 16 : 8)*size_t(G - 1)*N*N,
                               hipMemcpyDeviceToHost, s));
    return FFK_OK;
}

}  // namespace ffk_api

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
    if (batch < 1 || N < 1 || !d_ok(d)) return 0;
    return ffk::cumulant_workspace_bytes(batch, N, d);
}

int ffk_cumulant_function_dev(const double* decay_amplitudes, int batch, int N, int d,
                              const double* basis, int single_qubit, double* cumulant_function,
                              void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(decay_amplitudes && basis && cumulant_function, "NULL argument");
    FFK_REQUIRE(batch >= 1 && N >= 1, "empty axis");
    FFK_REQUIRE(d_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D);
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
    FFK_REQUIRE(d_ok(d), "dimension %d outside [2, %d]", d, FFK_MAX_D);
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
namespace {
// |B|_1 <= 1/2 after s halvings (launch_expm_real's Taylor polynomial is sized for that)
int squarings_for(double norm) {
    int squarings = 0;
    while (norm > 0.5 && squarings < 64) {
        norm *= 0.5;
        ++squarings;
    }
    return squarings;
}
}  // namespace

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
    const int squarings = squarings_for(norm);
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

size_t ffk_error_transfer_matrix_workspace_bytes(int N) {
    if (N < 1) return 0;
    return 6*align_up(8*size_t(N)*N) + align_up(16);
}

int ffk_error_transfer_matrix_dev(const double* cumulant_function, int batch, int N, double* result,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    FFK_REQUIRE(cumulant_function && result, "NULL argument");
    FFK_REQUIRE(batch >= 1, "empty axis");
    FFK_REQUIRE(N >= 1 && N <= 4096, "matrix dimension %d outside [1, 4096]", N);
    FFK_REQUIRE(workspace && workspace_bytes >= ffk_error_transfer_matrix_workspace_bytes(N), "workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    Bump a(workspace, workspace_bytes);
    const size_t nn = size_t(N)*N;
    double* dA = a.take<double>(nn);
    double* w[5];
    for (double*& m : w) m = a.take<double>(nn);
    double* d_norm = a.take<double>(2);
    FFK_HIP(ffk::launch_sum_and_one_norm(cumulant_function, batch, N, dA, d_norm, st));
    // the number of squarings decides how many products are enqueued: 16 bytes come back first
    unsigned long long words[2];
    FFK_HIP(hipMemcpyAsync(words, d_norm, sizeof(words), hipMemcpyDeviceToHost, st));
    FFK_HIP(hipStreamSynchronize(st));
    FFK_REQUIRE(words[1] == 0, "matrix contains NaN or Inf");
    double norm;
    std::memcpy(&norm, &words[0], sizeof(norm));
    FFK_HIP(ffk::launch_expm_real(dA, N, squarings_for(norm), result, w, st));
    return FFK_OK;
}

// ---------------------------------------------------------------------------------------------
}  // extern "C"
