// ffk_api_frozen.hip -- extern "C" entry points OUTSIDE SURVEY section 8's scope table, frozen since
// round 2 and exercised by the `slow` tests only: second-order filter function, frequency shifts and
// their cumulant-function contribution; the gradient of the filter function / infidelity.
#include "ffk_api_common.h"

extern "C" {

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
    return kernel_fault_status();
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
    return kernel_fault_status();
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

}  // extern "C"
