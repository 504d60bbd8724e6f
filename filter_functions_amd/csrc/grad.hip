// grad.hip -- K9: derivative of the filter function with respect to the control amplitudes
// (filter_functions/gradient.py; PulseSequence.get_filter_function_derivative,
// pulse_sequence.py:977-1054).
//
// The reference builds the derivative of the control matrix, (n_ctrl, W, G, A, d^2), from (I) the
// derivative of segment s's own contribution (_control_matrix_at_timestep_derivative, :200-381, with
// the nested integral _derivative_integral, :69-108) and (II) the derivative of every later Liouville
// propagator (_liouville_derivative, :111-197, contracted over all pairs of segments at :520 -- an
// O(G^2) tensor), and contracts it with conj(R) (calculate_filter_function_derivative, :526-556).
//
// Here everything stays in Hilbert space.  With Y_a(w) the interaction-picture noise operator
// (R_ak = tr(Y_a C_k), what the accumulate kernel sums), T = V_s^dag Q_s and bars denoting the
// eigenbasis of segment s:
//   (I)  2 Re tr(Y_a^dag Y'),  Y' = -i e^{i w t_s} T^dag G^T T,
//        G_xy = sum_n Bbar_yn Abar_nx J(w; W_yn, W_nx) - sum_q Abar_yq Bbar_qx J(w; W_qx, W_yq),
//        J(w; a, b) = int_0^dt dtau e^{i (w+a) tau} int_0^tau dtau' e^{i b tau'}
//                   = (I1(w+a+b) - I1(w+a)) / (i b)            (b != 0)
//                   = (dt e^{i (w+a) dt} - I1(w+a)) / (i (w+a))  (b == 0; dt^2/2 if w+a == 0 too),
//        so only the d^2 first-order integrals I1(w + W_mn) of the segment are needed
//        (W_yn + W_nx = W_yx): the same d^2 sincos per (segment, frequency) as the control matrix;
//   (II) a later propagator changes by Q_g -> Q_g E with the SAME anti-Hermitian generator
//        E_hs = -i T^dag (Abar_h o I1(0)) T for every g > s, so the sum over later segments collapses
//        to -2 Re tr(E_hs [Y_a^dag, Ycum_sa]) with Ycum the steps up to and including s (the part
//        proportional to the total Y_a drops out of the real part): O(G) instead of O(G^2);
//   and the explicit dependence of the noise sensitivities, (n'_ahs / n_as) 2 Re tr(Y_a^dag Ystep_sa)
//   (:376-379).
// Pass 1 (existing kernels) leaves the per-segment steps of Y in HBM; a prefix sum turns them into
// Ycum; this kernel: one lane per frequency, one block row per segment, everything that does not
// depend on the frequency broadcast from LDS.
#include "ffk_internal.h"

#ifndef FFK_GRAD_LDS_MAX_D
#define FFK_GRAD_LDS_MAX_D 4      // per-lane I1 / W_a columns in LDS up to this dimension (above: scratch
                                  // arrays win, the LDS footprint costs more occupancy than it saves)
#endif

namespace ffk {
namespace {

// E[h][s] = -i T^dag (Abar_h o I1(0)) T, one block per (s, h)
__global__ __launch_bounds__(64) void grad_generator_kernel(const cplx* __restrict__ ops, int ops_stride,
                                                            const cplx* __restrict__ abar,
                                                            const double* __restrict__ eigvals,
                                                            const double* __restrict__ dt, int G,
                                                            int d, cplx* __restrict__ E) {
    extern __shared__ unsigned char smem[];
    const int d2 = d*d;
    cplx* X = reinterpret_cast<cplx*>(smem);          // Abar o I1(0)
    cplx* Ts = X + d2;
    const int s = blockIdx.x, h = blockIdx.y;
    const cplx* T = ops + static_cast<size_t>(s)*ops_stride;
    const cplx* Ab = abar + (static_cast<size_t>(h)*G + s)*d2;
    const double dts = dt[s];
    for (int e = threadIdx.x; e < d2; e += 64) {
        const int m = e / d, n = e % d;
        const double dE = eigvals[static_cast<size_t>(s)*d + m] - eigvals[static_cast<size_t>(s)*d + n];
        X[e] = cmul(Ab[e], first_order_integral(0.0, dE, dts));
        Ts[e] = T[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < d2; e += 64) {
        const int x = e / d, y = e % d;
        cplx acc = {0.0, 0.0};
        for (int m = 0; m < d; ++m) {
            cplx row = {0.0, 0.0};
            for (int n = 0; n < d; ++n) cmac(row, X[m*d + n], Ts[n*d + y]);
            cmac_conj(acc, Ts[m*d + x], row);
        }
        E[(static_cast<size_t>(h)*G + s)*d2 + e] = {acc.im, -acc.re};      // -i acc
    }
}

template <int D>
__global__ __launch_bounds__(64) void grad_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ eigvals,
    const double* __restrict__ dt, const double* __restrict__ t, const cplx* __restrict__ ops,
    const cplx* __restrict__ abar, const cplx* __restrict__ E, const cplx* __restrict__ Ycum,
    const double* __restrict__ ratio, int G, int A, int H, double* __restrict__ out) {
    constexpr int D2 = D*D;
#if defined(FFK_GRAD_UNROLL_MAX)
    constexpr int kUnrollMax = FFK_GRAD_UNROLL_MAX;
#else
    constexpr int kUnrollMax = 4;
#endif
    constexpr int U = D <= kUnrollMax ? D : 1;  // unroll (register arrays) only where they fit
    extern __shared__ unsigned char smem[];
    double* dE = reinterpret_cast<double*>(smem);          // [D2]  W_mn
    double* inv = dE + D2;                                 // [D2]  1/W_mn, 0 where W_mn == 0
    cplx* Ts = reinterpret_cast<cplx*>(inv + D2);          // [D2]
    cplx* Bs = Ts + D2;                                    // [A][D2]
    cplx* As = Bs + A*D2;                                  // [H][D2]
    cplx* Es = As + H*D2;                                  // [H][D2]
    double* sec = reinterpret_cast<double*>(Es + H*D2);    // [H][64]  per-lane Re tr(E_h comm)
    const int s = blockIdx.x;            // segments on x: G may exceed 65535
    const int w = blockIdx.y*64 + threadIdx.x;
    for (int e = threadIdx.x; e < D2; e += 64) {
        const double v = eigvals[static_cast<size_t>(s)*D + e / D] - eigvals[static_cast<size_t>(s)*D + e % D];
        dE[e] = v;
        inv[e] = v == 0.0 ? 0.0 : 1.0/v;
        Ts[e] = ops[static_cast<size_t>(s)*(1 + A)*D2 + e];
    }
    for (int e = threadIdx.x; e < A*D2; e += 64) Bs[e] = ops[(static_cast<size_t>(s)*(1 + A) + 1)*D2 + e];
    for (int e = threadIdx.x; e < H*D2; e += 64) {
        const int h = e / D2, r = e % D2;
        As[e] = abar[(static_cast<size_t>(h)*G + s)*D2 + r];
        Es[e] = E[(static_cast<size_t>(h)*G + s)*D2 + r];
    }
    __syncthreads();
    if (w >= W) return;
    const double om = omega[w], dts = dt[s];
    const cplx ph = cexp(om*t[s]);
    // I1[m][n] = I1(w + W_mn) and, further down, Wa: per-lane columns in LDS for D <= 4 (dynamic
    // indexing without scratch), thread-local arrays (scratch) above
    constexpr bool kLds = D <= FFK_GRAD_LDS_MAX_D;
    cplx I1loc[kLds ? 1 : D2], Waloc[kLds ? 1 : D2];
    cplx* I1 = kLds ? reinterpret_cast<cplx*>(sec + H*64) + threadIdx.x : I1loc;
    cplx* Wa = kLds ? reinterpret_cast<cplx*>(sec + H*64) + D2*64 + threadIdx.x : Waloc;
    constexpr int LS = kLds ? 64 : 1;           // element stride of the two arrays
#pragma unroll 1
    for (int e = 0; e < D2; ++e) I1[e*LS] = first_order_integral(om, dE[e], dts);
    // int_0^dt tau e^{i x tau} dtau, x = w + W_e, from I1(x): the b == 0 branch of J
    auto nested = [&](cplx i1, int e) {
        const double x = om + dE[e];
        const cplx ex = {1.0 - x*i1.im, x*i1.re};                  // e^{i x dt} = 1 + i x I1
        cplx jd = {0.5*dts*dts, 0.0};
        if (x != 0.0) {
            const double rx = 1.0/x;
            jd = {(dts*ex.im - i1.im)*rx, -(dts*ex.re - i1.re)*rx};   // (dt ex - I1)/(i x)
        }
        return jd;
    };
    const size_t slab = static_cast<size_t>(A)*D2*W;       // one segment of Ycum
    for (int a = 0; a < A; ++a) {
        const cplx* Ytot = Ycum + static_cast<size_t>(G - 1)*slab + static_cast<size_t>(a)*D2*W + w;
        const cplx* Yc = Ycum + static_cast<size_t>(s)*slab + static_cast<size_t>(a)*D2*W + w;
        double tr_step = 0.0;
        {
            cplx Yd[D2], Yq[D2];                           // Ytot^dag, Ycum_s
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int y = 0; y < D; ++y) {
                    const cplx v = Ytot[static_cast<size_t>(y*D + x)*W];
                    Yd[x*D + y] = {v.re, -v.im};
                    Yq[x*D + y] = Yc[static_cast<size_t>(x*D + y)*W];
                }
            // (II): sec[h] = Re tr(E_h [Yd, Ycum])
            for (int h = 0; h < H; ++h) sec[h*64 + threadIdx.x] = 0.0;
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int y = 0; y < D; ++y) {
                    cplx c = {0.0, 0.0};                   // comm[y][x]
#pragma unroll U
                    for (int k = 0; k < D; ++k) {
                        cmac(c, Yd[y*D + k], Yq[k*D + x]);
                        const cplx p = cmul(Yq[y*D + k], Yd[k*D + x]);
                        c.re -= p.re;
                        c.im -= p.im;
                    }
                    for (int h = 0; h < H; ++h) {
                        const cplx e = Es[h*D2 + x*D + y];
                        sec[h*64 + threadIdx.x] += e.re*c.re - e.im*c.im;
                    }
                }
            // explicit sensitivity term: 2 Re tr(Yd Ystep), Ystep = Ycum_s - Ycum_{s-1}
            if (ratio) {
                const cplx* Yp = Yc - slab;
#pragma unroll U
                for (int x = 0; x < D; ++x)
#pragma unroll U
                    for (int y = 0; y < D; ++y) {
                        cplx st = Yq[y*D + x];
                        if (s > 0) {
                            const cplx pv = Yp[static_cast<size_t>(y*D + x)*W];
                            st.re -= pv.re;
                            st.im -= pv.im;
                        }
                        tr_step += Yd[x*D + y].re*st.re - Yd[x*D + y].im*st.im;
                    }
                tr_step *= 2.0;
            }
            // Wa = T Yd T^dag (Yq's registers are free from here on)
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int y = 0; y < D; ++y) {
                    cplx acc = {0.0, 0.0};
#pragma unroll U
                    for (int k = 0; k < D; ++k) {
                        const cplx ty = Ts[y*D + k];
                        cmac(acc, Yd[x*D + k], cplx{ty.re, -ty.im});       // (Yd T^dag)[x][y]
                    }
                    Yq[x*D + y] = acc;
                }
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int y = 0; y < D; ++y) {
                    cplx acc = {0.0, 0.0};
#pragma unroll U
                    for (int k = 0; k < D; ++k) cmac(acc, Ts[x*D + k], Yq[k*D + y]);
                    Wa[(x*D + y)*LS] = acc;
                }
        }
        const cplx* Bb = Bs + a*D2;
        for (int h = 0; h < H; ++h) {
            const cplx* Ab = As + h*D2;
            cplx first = {0.0, 0.0};
#pragma unroll 1
            for (int x = 0; x < D; ++x)
#pragma unroll 1
                for (int y = 0; y < D; ++y) {
                    // G_xy
                    cplx g = {0.0, 0.0};
                    const cplx iyx = I1[(y*D + x)*LS];
#pragma unroll U
                    for (int n = 0; n < D; ++n) {
                        // + Bbar_yn Abar_nx J(w; W_yn, W_nx)
                        const double r1 = inv[n*D + x];
                        const cplx iyn = I1[(y*D + n)*LS];
                        cplx j1;
                        if (r1 != 0.0) {
                            const cplx df = {iyx.re - iyn.re, iyx.im - iyn.im};
                            j1 = {df.im*r1, -df.re*r1};                     // df/(i W_nx)
                        } else {
                            j1 = nested(iyn, y*D + n);
                        }
                        cmac(g, cmul(Bb[y*D + n], Ab[n*D + x]), j1);
                        // - Abar_yn Bbar_nx J(w; W_nx, W_yn)
                        const double r2 = inv[y*D + n];
                        const cplx inx = I1[(n*D + x)*LS];
                        cplx j2;
                        if (r2 != 0.0) {
                            const cplx df = {iyx.re - inx.re, iyx.im - inx.im};
                            j2 = {df.im*r2, -df.re*r2};
                        } else {
                            j2 = nested(inx, n*D + x);
                        }
                        const cplx ab = cmul(Ab[y*D + n], Bb[n*D + x]);
                        cmac(g, cplx{-ab.re, -ab.im}, j2);
                    }
                    cmac(first, Wa[(x*D + y)*LS], g);
                }
            // 2 Re(-i ph first) = 2 Im(ph first)
            const cplx pf = cmul(ph, first);
            double val = 2.0*pf.im - 2.0*sec[h*64 + threadIdx.x];
            if (ratio) val += ratio[(static_cast<size_t>(a)*H + h)*G + s]*tr_step;
            out[((static_cast<size_t>(a)*G + s)*H + h)*W + w] = val;
        }
    }
}

// The derivative of the control matrix itself (gradient.calculate_derivative_of_control_matrix_from_scratch,
// gradient.py:384-523), out (H, W, G, A, N) = tr(dY C_k) with
//   dY_a/du_h(t_s) = Y' + [Ytot_a - Ycum_{s,a}, E_hs] + ratio_ahs Ystep_{s,a}
// (same Y', E, Ycum as above; here the part proportional to Ytot does not drop out).  One lane per
// frequency, one block row per segment; the tensor is written for callers that want it -- the
// filter-function / infidelity gradients above never form it.
template <int D>
__global__ __launch_bounds__(64) void grad_ctrlmat_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ eigvals,
    const double* __restrict__ dt, const double* __restrict__ t, const cplx* __restrict__ ops,
    const cplx* __restrict__ abar, const cplx* __restrict__ E, const cplx* __restrict__ Ycum,
    const double* __restrict__ ratio, const cplx* __restrict__ basis, int N, int G, int A, int H,
    cplx* __restrict__ out) {
    constexpr int D2 = D*D;
    constexpr int U = D <= 4 ? D : 1;
    extern __shared__ unsigned char smem[];
    double* dE = reinterpret_cast<double*>(smem);          // [D2]  W_mn
    double* inv = dE + D2;                                 // [D2]  1/W_mn, 0 where W_mn == 0
    cplx* Ts = reinterpret_cast<cplx*>(inv + D2);          // [D2]
    cplx* Bs = Ts + D2;                                    // [A][D2]
    cplx* As = Bs + A*D2;                                  // [H][D2]
    cplx* Es = As + H*D2;                                  // [H][D2]
    cplx* Cb = Es + H*D2;                                  // [N][D2]
    const int s = blockIdx.x;            // segments on x: G may exceed 65535
    const int w = blockIdx.y*64 + threadIdx.x;
    for (int e = threadIdx.x; e < D2; e += 64) {
        const double v = eigvals[static_cast<size_t>(s)*D + e / D] - eigvals[static_cast<size_t>(s)*D + e % D];
        dE[e] = v;
        inv[e] = v == 0.0 ? 0.0 : 1.0/v;
        Ts[e] = ops[static_cast<size_t>(s)*(1 + A)*D2 + e];
    }
    for (int e = threadIdx.x; e < A*D2; e += 64) Bs[e] = ops[(static_cast<size_t>(s)*(1 + A) + 1)*D2 + e];
    for (int e = threadIdx.x; e < H*D2; e += 64) {
        const int h = e / D2, r = e % D2;
        As[e] = abar[(static_cast<size_t>(h)*G + s)*D2 + r];
        Es[e] = E[(static_cast<size_t>(h)*G + s)*D2 + r];
    }
    for (int e = threadIdx.x; e < N*D2; e += 64) Cb[e] = basis[e];
    __syncthreads();
    if (w >= W) return;
    const double om = omega[w], dts = dt[s];
    const cplx ph = cexp(om*t[s]);
    cplx I1[D2];
#pragma unroll U
    for (int e = 0; e < D2; ++e) I1[e] = first_order_integral(om, dE[e], dts);
    auto nested = [&](cplx i1, int e) {                    // int_0^dt tau e^{i x tau} dtau from I1(x)
        const double x = om + dE[e];
        const cplx ex = {1.0 - x*i1.im, x*i1.re};
        cplx jd = {0.5*dts*dts, 0.0};
        if (x != 0.0) {
            const double rx = 1.0/x;
            jd = {(dts*ex.im - i1.im)*rx, -(dts*ex.re - i1.re)*rx};
        }
        return jd;
    };
    const size_t slab = static_cast<size_t>(A)*D2*W;
    for (int a = 0; a < A; ++a) {
        const cplx* Ytot = Ycum + static_cast<size_t>(G - 1)*slab + static_cast<size_t>(a)*D2*W + w;
        const cplx* Yc = Ycum + static_cast<size_t>(s)*slab + static_cast<size_t>(a)*D2*W + w;
        cplx Yr[D2], Ys[D2];                               // Ytot - Ycum_s, Ystep_s
#pragma unroll U
        for (int e = 0; e < D2; ++e) {
            const cplx tot = Ytot[static_cast<size_t>(e)*W], c = Yc[static_cast<size_t>(e)*W];
            cplx p = {0.0, 0.0};
            if (s > 0) p = (Yc - slab)[static_cast<size_t>(e)*W];
            Yr[e] = {tot.re - c.re, tot.im - c.im};
            Ys[e] = {c.re - p.re, c.im - p.im};
        }
        const cplx* Bb = Bs + a*D2;
        for (int h = 0; h < H; ++h) {
            const cplx* Ab = As + h*D2;
            const cplx* Eh = Es + h*D2;
            cplx Gm[D2];
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int y = 0; y < D; ++y) {
                    cplx g = {0.0, 0.0};
                    const cplx iyx = I1[y*D + x];
#pragma unroll U
                    for (int n = 0; n < D; ++n) {
                        const double r1 = inv[n*D + x];
                        const cplx iyn = I1[y*D + n];
                        cplx j1;
                        if (r1 != 0.0) {
                            const cplx df = {iyx.re - iyn.re, iyx.im - iyn.im};
                            j1 = {df.im*r1, -df.re*r1};
                        } else {
                            j1 = nested(iyn, y*D + n);
                        }
                        cmac(g, cmul(Bb[y*D + n], Ab[n*D + x]), j1);
                        const double r2 = inv[y*D + n];
                        const cplx inx = I1[n*D + x];
                        cplx j2;
                        if (r2 != 0.0) {
                            const cplx df = {iyx.re - inx.re, iyx.im - inx.im};
                            j2 = {df.im*r2, -df.re*r2};
                        } else {
                            j2 = nested(inx, n*D + x);
                        }
                        const cplx ab = cmul(Ab[y*D + n], Bb[n*D + x]);
                        cmac(g, cplx{-ab.re, -ab.im}, j2);
                    }
                    Gm[x*D + y] = g;
                }
            // M = G^T T, then dY = -i ph T^dag M
            cplx M[D2];
#pragma unroll U
            for (int x = 0; x < D; ++x)
#pragma unroll U
                for (int j = 0; j < D; ++j) {
                    cplx acc = {0.0, 0.0};
#pragma unroll U
                    for (int y = 0; y < D; ++y) cmac(acc, Gm[y*D + x], Ts[y*D + j]);
                    M[x*D + j] = acc;
                }
            const double r = ratio ? ratio[(static_cast<size_t>(a)*H + h)*G + s] : 0.0;
            cplx dY[D2];
#pragma unroll U
            for (int i = 0; i < D; ++i)
#pragma unroll U
                for (int j = 0; j < D; ++j) {
                    cplx acc = {0.0, 0.0};
#pragma unroll U
                    for (int x = 0; x < D; ++x) cmac_conj(acc, Ts[x*D + i], M[x*D + j]);
                    const cplx pv = cmul(ph, acc);
                    cplx v = {pv.im, -pv.re};                               // -i ph acc
                    // + [Yr, E]_ij
#pragma unroll U
                    for (int k = 0; k < D; ++k) {
                        cmac(v, Yr[i*D + k], Eh[k*D + j]);
                        const cplx q = cmul(Eh[i*D + k], Yr[k*D + j]);
                        v.re -= q.re;
                        v.im -= q.im;
                    }
                    v.re += r*Ys[i*D + j].re;
                    v.im += r*Ys[i*D + j].im;
                    dY[i*D + j] = v;
                }
            cplx* dst = out + (((static_cast<size_t>(h)*W + w)*G + s)*A + a)*N;
            for (int k = 0; k < N; ++k) {
                const cplx* C = Cb + k*D2;
                cplx acc = {0.0, 0.0};
#pragma unroll U
                for (int i = 0; i < D; ++i)
#pragma unroll U
                    for (int j = 0; j < D; ++j) cmac(acc, dY[i*D + j], C[j*D + i]);
                dst[k] = acc;
            }
        }
    }
}

// out (A,G,H,W) = 2 Re sum_k conj(R[a,k,w]) dR[h,w,s,a,k]  (gradient.py:526-556)
__global__ __launch_bounds__(256) void ctrlmat_deriv_contract_kernel(const cplx* __restrict__ R,
                                                                     const cplx* __restrict__ dR, int A,
                                                                     int N, int W, int G, int H,
                                                                     double* __restrict__ out) {
    const size_t idx = static_cast<size_t>(blockIdx.x)*256 + threadIdx.x;
    const size_t total = static_cast<size_t>(A)*G*H*W;
    if (idx >= total) return;
    const int w = static_cast<int>(idx % W);
    const int h = static_cast<int>((idx / W) % H);
    const int s = static_cast<int>((idx / W / H) % G);
    const int a = static_cast<int>(idx / W / H / G);
    const cplx* d = dR + (((static_cast<size_t>(h)*W + w)*G + s)*A + a)*N;
    const cplx* r = R + static_cast<size_t>(a)*N*W + w;
    double sum = 0.0;
    for (int k = 0; k < N; ++k) {
        const cplx rv = r[static_cast<size_t>(k)*W];
        sum += rv.re*d[k].re + rv.im*d[k].im;
    }
    out[idx] = 2.0*sum;
}

// out[row] = sum_w dF[row, w] * Re(scale[srow(row), w]) / d, rows = (a, s, h)
__global__ __launch_bounds__(256) void grad_integrate_kernel(const double* __restrict__ dF, int GH,
                                                             int W, const cplx* __restrict__ scale,
                                                             int s_ndim, double inv_d,
                                                             double* __restrict__ out) {
    const size_t row = blockIdx.x;
    const cplx* sc = scale + (s_ndim == 2 ? (row / GH)*static_cast<size_t>(W) : 0);
    const double* f = dF + row*W;
    double sum = 0.0;
    for (int w = threadIdx.x; w < W; w += 256) sum += f[w]*sc[w].re;
    __shared__ double red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = red[0]*inv_d;
}

template <int D>
hipError_t launch_grad(const double* omega, int W, const double* eigvals, const double* dt,
                       const double* t, const cplx* ops, const cplx* abar, const cplx* E,
                       const cplx* Ycum, const double* ratio, int G, int A, int H, double* out,
                       hipStream_t stream) {
    const size_t lds = 2*D*D*sizeof(double) + size_t(1 + A + 2*H)*D*D*sizeof(cplx) +
                       size_t(H)*64*sizeof(double) +
                       (D <= FFK_GRAD_LDS_MAX_D ? 2*size_t(D*D)*64*sizeof(cplx) : 0);
    if (lds > 160*1024) return hipErrorInvalidValue;
    if (lds > 48*1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(grad_kernel<D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((grad_kernel<D>), dim3(G, (W + 63)/64), dim3(64), lds, stream, omega, W, eigvals,
                       dt, t, ops, abar, E, Ycum, ratio, G, A, H, out);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_filter_function_derivative(const double* omega, int W, const double* eigvals,
                                             const double* dt, const double* t, const cplx* ops,
                                             const cplx* abar, const cplx* Ycum, const double* ratio,
                                             int G, int d, int A, int H, cplx* E, double* out,
                                             hipStream_t stream) {
    if (d < 2 || d > 8 || G > 65535 || H > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grad_generator_kernel, dim3(G, H), dim3(64), 2*d*d*sizeof(cplx), stream, ops,
                       (1 + A)*d*d, abar, eigvals, dt, G, d, E);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    switch (d) {
#define FFK_GRAD_CASE(D) \
    case D: return launch_grad<D>(omega, W, eigvals, dt, t, ops, abar, E, Ycum, ratio, G, A, H, out, stream);
        FFK_GRAD_CASE(2) FFK_GRAD_CASE(3) FFK_GRAD_CASE(4) FFK_GRAD_CASE(5)
        FFK_GRAD_CASE(6) FFK_GRAD_CASE(7) FFK_GRAD_CASE(8)
#undef FFK_GRAD_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t launch_control_matrix_derivative(const double* omega, int W, const double* eigvals,
                                            const double* dt, const double* t, const cplx* ops,
                                            const cplx* abar, const cplx* Ycum, const double* ratio,
                                            const cplx* basis, int N, int G, int d, int A, int H, cplx* E,
                                            cplx* out, hipStream_t stream) {
    if (d < 2 || d > 8 || G > 65535 || H > 65535 || N < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grad_generator_kernel, dim3(G, H), dim3(64), 2*d*d*sizeof(cplx), stream, ops,
                       (1 + A)*d*d, abar, eigvals, dt, G, d, E);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    const size_t lds = 2*size_t(d)*d*sizeof(double) + size_t(1 + A + 2*H + N)*d*d*sizeof(cplx);
    if (lds > 160*1024) return hipErrorInvalidValue;
    const dim3 grid(G, (W + 63)/64);
    switch (d) {
#define FFK_GRADC_CASE(D)                                                                                 \
    case D: {                                                                                             \
        if (lds > 48*1024) {                                                                              \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(grad_ctrlmat_kernel<D>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
            if (err != hipSuccess) return err;                                                            \
        }                                                                                                 \
        hipLaunchKernelGGL((grad_ctrlmat_kernel<D>), grid, dim3(64), lds, stream, omega, W, eigvals, dt, t, \
                           ops, abar, E, Ycum, ratio, basis, N, G, A, H, out);                            \
        return hipGetLastError();                                                                         \
    }
        FFK_GRADC_CASE(2) FFK_GRADC_CASE(3) FFK_GRADC_CASE(4) FFK_GRADC_CASE(5)
        FFK_GRADC_CASE(6) FFK_GRADC_CASE(7) FFK_GRADC_CASE(8)
#undef FFK_GRADC_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t launch_filter_function_derivative_from_control_matrix(const cplx* R, const cplx* dR, int A,
                                                                 int N, int W, int G, int H, double* out,
                                                                 hipStream_t stream) {
    const size_t total = static_cast<size_t>(A)*G*H*W;
    const size_t blocks = (total + 255)/256;
    if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ctrlmat_deriv_contract_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       stream, R, dR, A, N, W, G, H, out);
    return hipGetLastError();
}

hipError_t launch_infidelity_derivative(const double* dF, int A, int G, int H, int W, const cplx* scale,
                                        int s_ndim, int d, double* out, hipStream_t stream) {
    const size_t rows = static_cast<size_t>(A)*G*H;
    if (rows > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grad_integrate_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0, stream,
                       dF, G*H, W, scale, s_ndim, 1.0/d, out);
    return hipGetLastError();
}

}  // namespace ffk
