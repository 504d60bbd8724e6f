// second.hip -- K8: second-order filter function (the consumer of the step caches).
//
// numeric.calculate_second_order_filter_function_from_scratch (filter_functions/numeric.py:1470-1699)
// with the nested Magnus integral _second_order_integral (:170-256):
//
//   F2[a,b,k,l,w] = sum_g [ conj(G^(g)_{ak}(w)) sum_{g'<g} G^(g')_{bl}(w)                 "complete"
//                           + sum_{ij,mn} NB^(g)_{ak,ij} I^(g)_{ij,mn}(w) NB^(g)_{bl,mn} ]  "incomplete"
//   NB^(g)_{ak,ij} = Bbar^(g)_{a,ij} Cbar^(g)_{k,ji}        (:1632),
//   G^(g)_{ak}(w)  = e^{i w t_g} sum_ij NB^(g)_{ak,ij} I1^(g)_{ij}(w)   (the control-matrix step, :1649).
//
// The reference materialises I (W d^4 entries per segment) and contracts it twice.  Here the
// general branch of the integral is used in its factored form
//   I_{ij,mn} = (f(W_ij - w) - f(W_ij + W_mn)) / (w + W_mn),   f(x) = (e^{i x dt} - 1)/x,
// so that  X_{ak,mn} := sum_ij NB_{ak,ij} I_{ij,mn} = u_mn (P_ak - M_{ak,mn})  with
//   u_mn = 1/(w + W_mn)               one reciprocal per (g, w, mn),
//   P_ak = sum_ij NB_{ak,ij} f(W_ij - w)        A N d^2 MACs per (g, w),
//   M_{ak,mn} = sum_ij NB_{ak,ij} f(W_ij + W_mn)   frequency independent (so_prepare_kernel),
// and only the second contraction, F2_{ak,bl} += sum_mn X_{ak,mn} NB_{bl,mn}, is O((A N)^2 d^2) per
// (g, w): 8 (A N)^2 d^2 flops instead of 8 A N d^4 + 8 (A N)^2 d^2.  Entries with w + W_mn == 0
// exactly take the limit formulas (:186-194, :241-255) through X_{ak,mn} = sum_ij NB_{ak,ij} Isp_ij.
//
// One block: a (16 RT) x (16 RT) tile of the (A N) x (A N) output for WT frequencies; the segment
// loop runs inside the block, accumulators stay in registers, operands are staged through LDS in
// [mn][row] layout (the row index is the fast one: a wave reads 16 distinct NB rows and broadcasts
// its 4 X rows).
#include <cstdlib>

#include "ffk_internal.h"

namespace ffk {
namespace {

// (e^{i x dt} - 1)/x = (2 s / x)(-s + i c), s = sin(x dt/2), c = cos(x dt/2); i dt at x == 0
// (util.cexpm1, util.py:165-182, divided as in numeric.py:229-235)
__device__ __forceinline__ cplx frac(double x, double dt) {
    double s, c;
    sincos_pi(0.5*(x*dt), &s, &c);
    const double q = 2.0*s*rcp(x);
    cplx out = {-q*s, q*c};
    if (x == 0.0) out = {0.0, dt};
    return out;
}

// One block per (segment g, row r = a N + k):
//   NB[g][r][e = i d + j] = Bbar[a,g,i,j] Cbar[g,k,j,i]
//   M[g][r][f = m d + n]  = sum_e NB[g][r][e] f(dE_e + dE_f)
__global__ __launch_bounds__(64) void so_prepare_kernel(const cplx* __restrict__ nt,
                                                        const cplx* __restrict__ bt,
                                                        const double* __restrict__ eigvals,
                                                        const double* __restrict__ dt, int G, int A,
                                                        int N, int d, cplx* __restrict__ NB,
                                                        cplx* __restrict__ M) {
    extern __shared__ unsigned char smem[];
    const int d2 = d*d;
    cplx* row = reinterpret_cast<cplx*>(smem);                 // [d2]
    double* dE = reinterpret_cast<double*>(row + d2);          // [d2]
    const int g = blockIdx.x, r = blockIdx.y;
    const int a = r / N, k = r % N;
    const cplx* B = nt + (static_cast<size_t>(a)*G + g)*d2;
    const cplx* C = bt + (static_cast<size_t>(g)*N + k)*d2;
    cplx* NBr = NB + (static_cast<size_t>(g)*A*N + r)*d2;
    for (int e = threadIdx.x; e < d2; e += 64) {
        const int i = e / d, j = e % d;
        const cplx v = cmul(B[e], C[j*d + i]);
        row[e] = v;
        NBr[e] = v;
        dE[e] = eigvals[static_cast<size_t>(g)*d + i] - eigvals[static_cast<size_t>(g)*d + j];
    }
    __syncthreads();
    const double dtg = dt[g];
    cplx* Mr = M + (static_cast<size_t>(g)*A*N + r)*d2;
    for (int f = threadIdx.x; f < d2; f += 64) {
        cplx acc = {0.0, 0.0};
        const double df = dE[f];
        for (int e = 0; e < d2; ++e) cmac(acc, row[e], frac(dE[e] + df, dtg));
        Mr[f] = acc;
    }
}

template <int RT, int WT>
__global__ __launch_bounds__(256) void so_accumulate_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ eigvals,
    const double* __restrict__ dt, const double* __restrict__ t, const cplx* __restrict__ NB,
    const cplx* __restrict__ M, int G, int d, int A, int N, int mc, cplx* __restrict__ F2) {
    constexpr int T = 16*RT, Tp = T + 1;
    const int d2 = d*d, AN = A*N;
    extern __shared__ unsigned char smem[];
    cplx* Xs = reinterpret_cast<cplx*>(smem);   // [WT][mc][Tp]
    cplx* NBs = Xs + WT*mc*Tp;                  // [mc][Tp]
    cplx* frc1 = NBs + mc*Tp;                   // [WT][d2]   f(W_ij - w)
    cplx* I1 = frc1 + WT*d2;                    // [WT][d2]   e^{i w t_g} I1_ij(w)
    cplx* Isp = I1 + WT*d2;                     // [WT][d2]   limit integrals (w + W_mn == 0)
    cplx* P = Isp + WT*d2;                      // [WT][T]
    cplx* Xsp = P + WT*T;                       // [WT][T]
    cplx* GsA = Xsp + WT*T;                     // [WT][T]
    cplx* GsB = GsA + WT*T;                     // [WT][T]
    cplx* Gcum = GsB + WT*T;                    // [WT][T]
    double* u = reinterpret_cast<double*>(Gcum + WT*T);   // [WT][d2]   1/(w + W_mn), 0 if special
    int* spec = reinterpret_cast<int*>(u + WT*d2);        // [WT][d2]

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int w0 = blockIdx.x*WT;
    const int rowA0 = blockIdx.y*T, rowB0 = blockIdx.z*T;

    cplx acc[WT][RT][RT];
#pragma unroll
    for (int wi = 0; wi < WT; ++wi)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < RT; ++j) acc[wi][i][j] = {0.0, 0.0};
    for (int q = tid; q < WT*T; q += 256) Gcum[q] = {0.0, 0.0};

    for (int g = 0; g < G; ++g) {
        const double dtg = dt[g], tg = t[g];
        const double* D = eigvals + static_cast<size_t>(g)*d;
        // (1) the frequency-dependent scalars of this segment
        int special = 0;
        for (int q = tid; q < WT*d2; q += 256) {
            const int wi = q / d2, e = q % d2;
            const double dE = D[e / d] - D[e % d];
            const double w = omega[min(w0 + wi, W - 1)];
            const double a = -w + dE;              // np.add.outer(-E, dE), numeric.py:208
            const double b = w + dE;               // np.add.outer(E, dE),  numeric.py:207
            const cplx fa = frac(a, dtg);
            const cplx fb = frac(b, dtg);
            frc1[q] = fa;
            I1[q] = cmul(cexp(w*tg), cplx{fb.im, -fb.re});     // first-order integral = -i f(b)
            const bool sp = b == 0.0;
            special |= sp;
            spec[q] = sp;
            u[q] = sp ? 0.0 : 1.0/b;
        }
        const int any_special = __syncthreads_or(special);
        if (any_special) {      // rare: the limit integrals of the entries with w + W_mn == 0
            for (int q = tid; q < WT*d2; q += 256) {
                const int wi = q / d2, e = q % d2;
                const double a = -omega[min(w0 + wi, W - 1)] + (D[e / d] - D[e % d]);
                cplx lim = {0.5*dtg*dtg, 0.0};
                if (a != 0.0) {
                    const cplx fa = frc1[q], ph = cexp(a*dtg);
                    const double ra = 1.0/a;
                    lim = {(fa.re + dtg*ph.im)*ra, (fa.im - dtg*ph.re)*ra};   // (f(a) - i dt e^{i a dt})/a
                }
                Isp[q] = lim;
            }
            __syncthreads();
        }
        // (2) per-row contractions with the d^2 scalars: P, Xsp, G^(g) for the A rows; G^(g) for the
        // B rows
        if (tid < WT*2*T) {
            const int wi = tid / (2*T), side = (tid / T) & 1, r = tid % T;
            const int rowi = (side ? rowB0 : rowA0) + r;
            cplx p = {0.0, 0.0}, gs = {0.0, 0.0}, xs = {0.0, 0.0};
            if (rowi < AN) {
                const cplx* nb = NB + (static_cast<size_t>(g)*AN + rowi)*d2;
                const cplx* f1 = frc1 + wi*d2;
                const cplx* i1 = I1 + wi*d2;
                const cplx* is = Isp + wi*d2;
                for (int e = 0; e < d2; ++e) {
                    const cplx v = nb[e];
                    cmac(gs, v, i1[e]);
                    if (!side) {
                        cmac(p, v, f1[e]);
                        if (any_special) cmac(xs, v, is[e]);
                    }
                }
            }
            if (side) {
                GsB[wi*T + r] = gs;
            } else {
                GsA[wi*T + r] = gs;
                P[wi*T + r] = p;
                Xsp[wi*T + r] = xs;
            }
        }
        __syncthreads();
        // (3) complete intervals: conj(G^(g)_A) x cumulative G_B up to g-1   (numeric.py:1679)
        if (g > 0) {
#pragma unroll
            for (int wi = 0; wi < WT; ++wi) {
                cplx gb[RT];
#pragma unroll
                for (int j = 0; j < RT; ++j) gb[j] = Gcum[wi*T + tx*RT + j];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const cplx ga = GsA[wi*T + ty*RT + i];
                    const cplx gac = {ga.re, -ga.im};
#pragma unroll
                    for (int j = 0; j < RT; ++j) cmac(acc[wi][i][j], gac, gb[j]);
                }
            }
        }
        // (4) incomplete interval, mn in chunks of mc
        for (int f0 = 0; f0 < d2; f0 += mc) {
            const int mcur = min(mc, d2 - f0);
            for (int q = tid; q < mcur*T; q += 256) {
                const int fl = q % mcur, r = q / mcur;
                const int f = f0 + fl;
                const int rb = rowB0 + r, ra = rowA0 + r;
                cplx nbv = {0.0, 0.0};
                if (rb < AN) nbv = NB[(static_cast<size_t>(g)*AN + rb)*d2 + f];
                NBs[fl*Tp + r] = nbv;
                cplx mv = {0.0, 0.0};
                if (ra < AN) mv = M[(static_cast<size_t>(g)*AN + ra)*d2 + f];
#pragma unroll
                for (int wi = 0; wi < WT; ++wi) {
                    const cplx pv = P[wi*T + r];
                    const double uv = u[wi*d2 + f];
                    cplx x = {uv*(pv.re - mv.re), uv*(pv.im - mv.im)};
                    if (spec[wi*d2 + f]) x = Xsp[wi*T + r];
                    if (ra >= AN) x = {0.0, 0.0};
                    Xs[(wi*mc + fl)*Tp + r] = x;
                }
            }
            __syncthreads();
            if (f0 == 0 && tid < WT*T) {          // every wave is past (3): advance the cumulative sum
                Gcum[tid].re += GsB[tid].re;
                Gcum[tid].im += GsB[tid].im;
            }
            for (int fl = 0; fl < mcur; ++fl) {
                cplx nb[RT];
#pragma unroll
                for (int j = 0; j < RT; ++j) nb[j] = NBs[fl*Tp + tx*RT + j];
#pragma unroll
                for (int wi = 0; wi < WT; ++wi) {
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const cplx x = Xs[(wi*mc + fl)*Tp + ty*RT + i];
#pragma unroll
                        for (int j = 0; j < RT; ++j) cmac(acc[wi][i][j], x, nb[j]);
                    }
                }
            }
            __syncthreads();
        }
    }
    // F2[a,b,k,l,w]: row r_A = a N + k, r_B = b N + l
#pragma unroll
    for (int wi = 0; wi < WT; ++wi) {
        const int w = w0 + wi;
        if (w >= W) continue;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int ra = rowA0 + ty*RT + i;
            if (ra >= AN) continue;
            const int a = ra / N, k = ra % N;
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const int rb = rowB0 + tx*RT + j;
                if (rb >= AN) continue;
                const int b = rb / N, l = rb % N;
                F2[(((static_cast<size_t>(a)*A + b)*N + k)*N + l)*W + w] = acc[wi][i][j];
            }
        }
    }
}

// ---- the same pass on the matrix cores -----------------------------------------------------------
// One wavefront owns a (16 MT) x (16 NT) complex tile of the (A N) x (A N) output of ONE frequency
// as MT NT 2 accumulator tiles of v_mfma_f64_16x16x4 (operand maps: A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[row = (lane>>4) + 4 r][col = lane&15]).  Per segment the second
// contraction is the complex product X (16 MT x d^2) . NB^T (d^2 x 16 NT): four real MFMA chains
// over ceil(d^2/4) k-steps, with the X operand formed in registers, X = u (P - M), directly in the
// A-operand layout.  The complete-interval term conj(G_A) x Gcum_B is a rank-one update: two more
// MFMAs per tile whose k = 0, 1 slots carry (re, im) and k = 2, 3 are zero.  The four wavefronts of
// a block take four consecutive frequencies and share the LDS copy of the segment's NB and M rows.
// (Splitting the tile over wavefronts instead -- 16-row strips, 48 accumulator registers, more
// wavefronts per SIMD -- was measured 2x slower: the per-segment scalar work does not shrink with
// the tile.)
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void so_mfma_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ eigvals,
    const double* __restrict__ dt, const double* __restrict__ t, const cplx* __restrict__ NB,
    const cplx* __restrict__ M, int G, int d, int A, int N, cplx* __restrict__ F2) {
    constexpr int RA = 16*MT, RB = 16*NT;
    const int d2 = d*d, d2s = d2 + 1, AN = A*N, KS = (d2 + 3)/4;
    extern __shared__ unsigned char smem[];
    cplx* NBa = reinterpret_cast<cplx*>(smem);      // [RA][d2s]
    cplx* Ma = NBa + RA*d2s;                        // [RA][d2s]
    cplx* NBb = Ma + RA*d2s;                        // [RB][d2s]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int per_wave = 4*d2 + 3*RA + 2*RB;
    cplx* frc1 = NBb + RB*d2s + wave*per_wave;      // [d2]
    cplx* I1 = frc1 + d2;                           // [d2]
    cplx* Isp = I1 + d2;                            // [d2]
    cplx* P = Isp + d2;                             // [RA]
    cplx* Xsp = P + RA;                             // [RA]
    cplx* GsA = Xsp + RA;                           // [RA]
    cplx* GsB = GsA + RA;                           // [RB]
    cplx* Gcum = GsB + RB;                          // [RB]
    double* us = reinterpret_cast<double*>(Gcum + RB);   // [d2]  1/(w + W_mn); 0 where that is singular
    int* spec = reinterpret_cast<int*>(us + d2);         // [d2]  ... and the flag for it

    const int w = blockIdx.x*4 + wave;
    const double om = omega[min(w, W - 1)];
    const int rowA0 = blockIdx.y*RA, rowB0 = blockIdx.z*RB;
    // on a diagonal tile the B rows are the A rows: G_B = G_A
    const bool diag = rowA0 == rowB0 && RA == RB;

    f64x4 cre[MT][NT], cim[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cre[mt][nt] = {0.0, 0.0, 0.0, 0.0};
            cim[mt][nt] = {0.0, 0.0, 0.0, 0.0};
        }
    for (int q = lane; q < RB; q += 64) Gcum[q] = {0.0, 0.0};

    for (int g = 0; g < G; ++g) {
        const double dtg = dt[g], tg = t[g];
        const double* D = eigvals + static_cast<size_t>(g)*d;
        __syncthreads();                            // the previous segment's operands are consumed
        for (int q = tid; q < RA*d2; q += 256) {
            const int r = q / d2, e = q % d2;
            cplx nb = {0.0, 0.0}, m = {0.0, 0.0};
            if (rowA0 + r < AN) {
                const size_t o = (static_cast<size_t>(g)*AN + rowA0 + r)*d2 + e;
                nb = NB[o];
                m = M[o];
            }
            NBa[r*d2s + e] = nb;
            Ma[r*d2s + e] = m;
        }
        for (int q = tid; q < RB*d2; q += 256) {
            const int r = q / d2, e = q % d2;
            cplx nb = {0.0, 0.0};
            if (rowB0 + r < AN) nb = NB[(static_cast<size_t>(g)*AN + rowB0 + r)*d2 + e];
            NBb[r*d2s + e] = nb;
        }
        // (1) this wavefront's frequency-dependent scalars
        bool special = false;
        for (int e = lane; e < d2; e += 64) {
            const double dE = D[e / d] - D[e % d];
            const double a = -om + dE, b = om + dE;
            const cplx fa = frac(a, dtg), fb = frac(b, dtg);
            frc1[e] = fa;
            I1[e] = cmul(cexp(om*tg), cplx{fb.im, -fb.re});      // e^{i w t_g} (-i f(b))
            const bool sp = b == 0.0;
            special |= sp;
            spec[e] = sp;
            us[e] = sp ? 0.0 : rcp(b);
        }
        const bool any_special = __ballot(special) != 0ull;
        if (any_special) {      // rare: the limit integrals of the entries with w + W_mn == 0
            for (int e = lane; e < d2; e += 64) {
                const double a = -om + (D[e / d] - D[e % d]);
                cplx lim = {0.5*dtg*dtg, 0.0};
                if (a != 0.0) {
                    const cplx fa = frac(a, dtg), ph = cexp(a*dtg);
                    const double ra = 1.0/a;
                    lim = {(fa.re + dtg*ph.im)*ra, (fa.im - dtg*ph.re)*ra};
                }
                Isp[e] = lim;
            }
        }
        __syncthreads();
        // (2) row contractions with the d^2 scalars
        for (int job = lane; job < (diag ? RA : RA + RB); job += 64) {
            const bool side = job >= RA;
            const int r = side ? job - RA : job;
            const cplx* nb = (side ? NBb : NBa) + r*d2s;
            cplx p = {0.0, 0.0}, gs = {0.0, 0.0}, xs = {0.0, 0.0};
            for (int e = 0; e < d2; ++e) {
                const cplx v = nb[e];
                cmac(gs, v, I1[e]);
                if (!side) {
                    cmac(p, v, frc1[e]);
                    if (any_special) cmac(xs, v, Isp[e]);
                }
            }
            if (side) {
                GsB[r] = gs;
            } else {
                GsA[r] = gs;
                if (diag) GsB[r] = gs;
                P[r] = p;
                Xsp[r] = xs;
            }
        }
        wave_sync();
        // (3) complete intervals: conj(G_A) x Gcum_B as a rank-one MFMA update
        if (g > 0) {
            double b1[NT], b2[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const cplx gc = Gcum[nt*16 + li];
                b1[nt] = lk == 0 ? gc.re : (lk == 1 ? gc.im : 0.0);
                b2[nt] = lk == 0 ? gc.im : (lk == 1 ? gc.re : 0.0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const cplx ga = GsA[mt*16 + li];
                const double a1 = lk == 0 ? ga.re : (lk == 1 ? ga.im : 0.0);
                const double a2 = lk == 0 ? ga.re : (lk == 1 ? -ga.im : 0.0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[nt], cre[mt][nt], 0, 0, 0);
                    cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2[nt], cim[mt][nt], 0, 0, 0);
                }
            }
        }
        wave_sync();
        for (int q = lane; q < RB; q += 64) {
            Gcum[q].re += GsB[q].re;
            Gcum[q].im += GsB[q].im;
        }
        // (4) incomplete interval: X . NB^T over the k-steps
        cplx p[MT], xsp[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            p[mt] = P[mt*16 + li];
            xsp[mt] = Xsp[mt*16 + li];
        }
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            const int mn = ks*4 + lk;
            const bool valid = mn < d2;
            const int e = valid ? mn : 0;
            const bool sp = valid && spec[e];
            const double u = valid ? us[e] : 0.0;
            double bre[NT], bim[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const cplx v = NBb[(nt*16 + li)*d2s + e];
                bre[nt] = valid ? v.re : 0.0;
                bim[nt] = valid ? v.im : 0.0;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const cplx m = Ma[(mt*16 + li)*d2s + e];
                double xre = u*(p[mt].re - m.re), xim = u*(p[mt].im - m.im);
                if (sp) {
                    xre = xsp[mt].re;
                    xim = xsp[mt].im;
                }
                const double nxim = -xim;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xre, bre[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xre, bim[nt], cim[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nxim, bim[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xim, bre[nt], cim[mt][nt], 0, 0, 0);
            }
        }
    }
    if (w >= W) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ra = rowA0 + mt*16 + lk + 4*r, rb = rowB0 + nt*16 + li;
                if (ra >= AN || rb >= AN) continue;
                const int a = ra / N, k = ra % N, b = rb / N, l = rb % N;
                F2[(((static_cast<size_t>(a)*A + b)*N + k)*N + l)*W + w] = {cre[mt][nt][r], cim[mt][nt][r]};
            }
}

size_t so_mfma_lds_bytes(int mt, int nt, int d2) {
    const size_t ra = 16*mt, rb = 16*nt;
    return sizeof(cplx)*((2*ra + rb)*(d2 + 1) + 4*(4*size_t(d2) + 3*ra + 2*rb));
}

template <int MT, int NT>
hipError_t launch_so_mfma(const double* omega, int W, const double* eigvals, const double* dt,
                          const double* t, const cplx* NB, const cplx* M, int G, int d, int A, int N,
                          cplx* F2, hipStream_t stream) {
    const int AN = A*N;
    const size_t lds = so_mfma_lds_bytes(MT, NT, d*d);
    auto kern = so_mfma_kernel<MT, NT>;
    if (lds > 48*1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    const dim3 grid((W + 3)/4, (AN + 16*MT - 1)/(16*MT), (AN + 16*NT - 1)/(16*NT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, omega, W, eigvals, dt, t, NB, M, G, d, A,
                       N, F2);
    return hipGetLastError();
}

size_t so_lds_bytes(int rt, int wt, int mc, int d2) {
    const size_t T = 16*rt, Tp = T + 1;
    return sizeof(cplx)*((wt + 1)*mc*Tp + 3*size_t(wt)*d2 + 5*size_t(wt)*T) +
           size_t(wt)*d2*(sizeof(double) + sizeof(int));
}

template <int RT, int WT>
hipError_t launch_so(const double* omega, int W, const double* eigvals, const double* dt,
                     const double* t, const cplx* NB, const cplx* M, int G, int d, int A, int N,
                     int mc, cplx* F2, hipStream_t stream) {
    const int AN = A*N, T = 16*RT;
    const int tiles = (AN + T - 1)/T;
    const size_t lds = so_lds_bytes(RT, WT, mc, d*d);
    auto kern = so_accumulate_kernel<RT, WT>;
    if (lds > 48*1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    const dim3 grid((W + WT - 1)/WT, tiles, tiles);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, omega, W, eigvals, dt, t, NB, M, G, d, A,
                       N, mc, F2);
    return hipGetLastError();
}

// Delta[row] = sum_w Re( F2[pair(row), kl(row), w] * scale[srow, w] ), one block per output row
__global__ __launch_bounds__(256) void frequency_shifts_kernel(const cplx* __restrict__ F2, int A,
                                                               int N, int W,
                                                               const cplx* __restrict__ scale,
                                                               int s_ndim,
                                                               const int32_t* __restrict__ idx,
                                                               int n_idx, double* __restrict__ out) {
    const size_t NN = static_cast<size_t>(N)*N;
    const size_t row = blockIdx.x;
    const size_t pair = row / NN, kl = row % NN;
    int a, b, srow;
    if (s_ndim == 3) {
        a = idx[pair / n_idx];
        b = idx[pair % n_idx];
        srow = static_cast<int>(pair);
    } else {
        a = b = idx[pair];
        srow = s_ndim == 2 ? static_cast<int>(pair) : 0;
    }
    const cplx* f = F2 + ((static_cast<size_t>(a)*A + b)*NN + kl)*W;
    const cplx* s = scale + static_cast<size_t>(srow)*W;
    double sum = 0.0;
    for (int w = threadIdx.x; w < W; w += 256) sum += f[w].re*s[w].re - f[w].im*s[w].im;
    __shared__ double red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = red[0];
}

// K[b][i][j] += -1/2 Re tr(C_i [X_b, C_j]),  X_b = sum_kl (Delta_kl - Delta_lk) C_k C_l
// (numeric.py:1166-1190 without the four-element trace tensor).  One block per batch element; the
// sizes are those of one error transfer matrix (N^2 outputs).
__global__ __launch_bounds__(256) void cumulant_second_order_kernel(const double* __restrict__ delta,
                                                                    int N, int d,
                                                                    const cplx* __restrict__ basis,
                                                                    cplx* __restrict__ ws,
                                                                    double* __restrict__ K) {
    const int d2 = d*d;
    const size_t b = blockIdx.x;
    const double* Dl = delta + b*N*N;
    cplx* Dk = ws + b*(static_cast<size_t>(N)*d2 + d2 + static_cast<size_t>(N)*d2);   // [N][d2]
    cplx* X = Dk + static_cast<size_t>(N)*d2;                                          // [d2]
    cplx* Y = X + d2;                                                                  // [N][d2]
    // D_k = sum_l (Delta_kl - Delta_lk) C_l
    for (int q = threadIdx.x; q < N*d2; q += 256) {
        const int k = q / d2, e = q % d2;
        cplx acc = {0.0, 0.0};
        for (int l = 0; l < N; ++l) {
            const double c = Dl[k*N + l] - Dl[l*N + k];
            const cplx v = basis[static_cast<size_t>(l)*d2 + e];
            acc.re = fma(c, v.re, acc.re);
            acc.im = fma(c, v.im, acc.im);
        }
        Dk[q] = acc;
    }
    __syncthreads();
    // X = sum_k C_k D_k
    for (int e = threadIdx.x; e < d2; e += 256) {
        const int i = e / d, j = e % d;
        cplx acc = {0.0, 0.0};
        for (int k = 0; k < N; ++k)
            for (int x = 0; x < d; ++x)
                cmac(acc, basis[static_cast<size_t>(k)*d2 + i*d + x], Dk[k*d2 + x*d + j]);
        X[e] = acc;
    }
    __syncthreads();
    // Y_j = [X, C_j]
    for (int q = threadIdx.x; q < N*d2; q += 256) {
        const int jj = q / d2, e = q % d2, i = e / d, j = e % d;
        const cplx* C = basis + static_cast<size_t>(jj)*d2;
        cplx acc = {0.0, 0.0};
        for (int x = 0; x < d; ++x) {
            cmac(acc, X[i*d + x], C[x*d + j]);
            const cplx c = C[i*d + x], xv = X[x*d + j];
            acc.re -= c.re*xv.re - c.im*xv.im;
            acc.im -= c.re*xv.im + c.im*xv.re;
        }
        Y[q] = acc;
    }
    __syncthreads();
    // K_ij += -1/2 Re tr(C_i Y_j)
    for (int q = threadIdx.x; q < N*N; q += 256) {
        const int i = q / N, j = q % N;
        const cplx* C = basis + static_cast<size_t>(i)*d2;
        const cplx* Yj = Y + static_cast<size_t>(j)*d2;
        double acc = 0.0;
        for (int p = 0; p < d; ++p)
            for (int x = 0; x < d; ++x) {
                const cplx c = C[p*d + x], y = Yj[x*d + p];
                acc += c.re*y.re - c.im*y.im;
            }
        K[b*N*N + q] += -0.5*acc;
    }
}

// ---- concatenation rule of the second-order filter function --------------------------------------
// numeric.calculate_second_order_filter_function_from_atomic (numeric.py:1702-1818).  Both the
// complete and the incomplete steps of a pulse are bilinear in the basis elements, so pulse g's own
// F2^(g) enters the sequence's through the Liouville matrix of the preceding propagator on both
// basis indices (the absolute-time phases of the nested integral cancel), and what is left is the
// rank-one term between pulse g's summand of the control matrix and the sum of the earlier ones:
//   F2[ab,kl] = sum_g ( sum_pq L^(g-1)[p,k] F2^(g)[ab,pq] L^(g-1)[q,l]
//                       + conj(G^(g)[a,k]) sum_{g'<g} G^(g')[b,l] ),       L^(-1) = 1.
// (The reference re-evaluates the incomplete steps from its W d^4 integral caches instead; the
// rotated form needs only each pulse's F2.)  One lane per frequency, one (a, b, k, 16 values of l)
// per thread; the Liouville entries are wave-uniform (scalar loads).
__global__ __launch_bounds__(64) void so_cumulative_kernel(const cplx* step, int G, size_t slab,
                                                           cplx* cum) {      // cum may alias step
    const size_t e = static_cast<size_t>(blockIdx.x)*64 + threadIdx.x;
    if (e >= slab) return;
    cplx acc = {0.0, 0.0};
    for (int g = 0; g < G; ++g) {
        const cplx v = step[g*slab + e];
        acc.re += v.re;
        acc.im += v.im;
        cum[g*slab + e] = acc;
    }
}

__global__ __launch_bounds__(64) void so_concat_kernel(const cplx* __restrict__ F2a,
                                                       const cplx* __restrict__ step,
                                                       const cplx* __restrict__ cum,
                                                       const double* __restrict__ L, int G, int A,
                                                       int N, int W, cplx* __restrict__ out) {
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y / A, b = blockIdx.y % A;
    const int ltiles = (N + 15)/16;
    const int k = blockIdx.z / ltiles, l0 = (blockIdx.z % ltiles)*16;
    if (w >= W) return;
    const size_t NN = static_cast<size_t>(N)*N;
    cplx acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = {0.0, 0.0};
    for (int g = 0; g < G; ++g) {
        const cplx* F = F2a + ((static_cast<size_t>(g)*A*A + blockIdx.y)*NN)*W + w;
        if (g == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (l0 + i < N) {
                    const cplx v = F[(static_cast<size_t>(k)*N + l0 + i)*W];
                    acc[i].re += v.re;
                    acc[i].im += v.im;
                }
            continue;
        }
        const double* Lg = L + static_cast<size_t>(g - 1)*NN;
        for (int q = 0; q < N; ++q) {
            cplx t = {0.0, 0.0};
            for (int p = 0; p < N; ++p) {
                const double lp = Lg[p*N + k];
                const cplx v = F[(static_cast<size_t>(p)*N + q)*W];
                t.re = fma(lp, v.re, t.re);
                t.im = fma(lp, v.im, t.im);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double lq = l0 + i < N ? Lg[q*N + l0 + i] : 0.0;
                acc[i].re = fma(t.re, lq, acc[i].re);
                acc[i].im = fma(t.im, lq, acc[i].im);
            }
        }
        const cplx ga = step[((static_cast<size_t>(g)*A + a)*N + k)*W + w];
        const cplx gc = {ga.re, -ga.im};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (l0 + i < N)
                cmac(acc[i], gc, cum[((static_cast<size_t>(g - 1)*A + b)*N + l0 + i)*W + w]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (l0 + i < N)
            out[((static_cast<size_t>(blockIdx.y)*N + k)*N + l0 + i)*W + w] = acc[i];
}

}  // namespace

hipError_t launch_segment_prefix_sum(cplx* Y, int G, size_t slab, hipStream_t stream) {
    if ((slab + 63)/64 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(so_cumulative_kernel, dim3(static_cast<unsigned>((slab + 63)/64)), dim3(64), 0,
                       stream, Y, G, slab, Y);
    return hipGetLastError();
}

size_t second_order_from_atomic_workspace_bytes(int G, int A, int N, int W) {
    return align_up(sizeof(cplx)*size_t(G)*A*N*W);
}

hipError_t launch_second_order_from_atomic(const cplx* F2_atomic, const cplx* step, const double* L,
                                           int G, int A, int N, int W, cplx* out, void* ws,
                                           hipStream_t stream) {
    const size_t slab = static_cast<size_t>(A)*N*W;
    const int ltiles = (N + 15)/16;
    if (static_cast<size_t>(N)*ltiles > 65535 || static_cast<size_t>(A)*A > 65535 ||
        (slab + 63)/64 > 0x7fffffffull)
        return hipErrorInvalidValue;
    cplx* cum = static_cast<cplx*>(ws);
    hipLaunchKernelGGL(so_cumulative_kernel, dim3(static_cast<unsigned>((slab + 63)/64)), dim3(64), 0,
                       stream, step, G, slab, cum);
    hipLaunchKernelGGL(so_concat_kernel, dim3((W + 63)/64, A*A, N*ltiles), dim3(64), 0, stream,
                       F2_atomic, step, cum, L, G, A, N, W, out);
    return hipGetLastError();
}

size_t second_order_workspace_bytes(int G, int A, int N, int d) {
    return 2*align_up(sizeof(cplx)*size_t(G)*A*N*d*d);
}

hipError_t launch_second_order_filter_function(const double* omega, int W, const double* eigvals,
                                               const double* dt, const double* t, const cplx* nt,
                                               const cplx* bt, int G, int d, int A, int N, cplx* F2,
                                               void* ws, hipStream_t stream) {
    const int d2 = d*d, AN = A*N;
    if (AN > 65535) return hipErrorInvalidValue;
    cplx* NB = static_cast<cplx*>(ws);
    cplx* M = reinterpret_cast<cplx*>(static_cast<unsigned char*>(ws) +
                                      align_up(sizeof(cplx)*size_t(G)*AN*d2));
    hipLaunchKernelGGL(so_prepare_kernel, dim3(G, AN), dim3(64), d2*(sizeof(cplx) + sizeof(double)),
                       stream, nt, bt, eigvals, dt, G, A, N, d, NB, M);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    // matrix-core kernel: the largest square wave tile (<= 48 x 48) whose operands fit in LDS twice
    // per CU; FFK_TUNE_SO_MFMA=0 forces the vector kernel (cross-check in tests/)
    bool use_mfma = true;
    if (const char* e = getenv("FFK_TUNE_SO_MFMA")) use_mfma = atoi(e) != 0;
    if (use_mfma) {
        const int tiles16 = (AN + 15)/16;
        for (int mt = min(3, tiles16); mt >= 1; --mt) {
            if (so_mfma_lds_bytes(mt, mt, d2) > 78*1024) continue;
            if (mt == 3) return launch_so_mfma<3, 3>(omega, W, eigvals, dt, t, NB, M, G, d, A, N, F2, stream);
            if (mt == 2) return launch_so_mfma<2, 2>(omega, W, eigvals, dt, t, NB, M, G, d, A, N, F2, stream);
            return launch_so_mfma<1, 1>(omega, W, eigvals, dt, t, NB, M, G, d, A, N, F2, stream);
        }
    }
    const int rt = min(4, (AN + 15)/16);
    const int wt = W >= 512 ? 2 : 1;
    const int mc = min(d2, 16);
#define FFK_SO_CASE(R, Wt)                                                                       \
    if (rt == R && wt == Wt)                                                                     \
        return launch_so<R, Wt>(omega, W, eigvals, dt, t, NB, M, G, d, A, N, mc, F2, stream);
    FFK_SO_CASE(1, 1) FFK_SO_CASE(2, 1) FFK_SO_CASE(3, 1) FFK_SO_CASE(4, 1)
    FFK_SO_CASE(1, 2) FFK_SO_CASE(2, 2) FFK_SO_CASE(3, 2) FFK_SO_CASE(4, 2)
#undef FFK_SO_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_frequency_shifts(const cplx* F2, int A, int N, int W, const cplx* scale,
                                   int s_ndim, const int32_t* idx, int n_idx, double* out,
                                   hipStream_t stream) {
    const size_t rows = (s_ndim == 3 ? size_t(n_idx)*n_idx : size_t(n_idx))*N*N;
    if (rows > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(frequency_shifts_kernel, dim3(static_cast<unsigned>(rows)), dim3(256), 0,
                       stream, F2, A, N, W, scale, s_ndim, idx, n_idx, out);
    return hipGetLastError();
}

size_t cumulant_second_order_workspace_bytes(size_t batch, int N, int d) {
    return batch*sizeof(cplx)*(2*size_t(N)*d*d + size_t(d)*d);
}

hipError_t launch_cumulant_second_order(const double* delta, size_t batch, int N, int d,
                                        const cplx* basis, double* K, void* ws, hipStream_t stream) {
    if (batch > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cumulant_second_order_kernel, dim3(static_cast<unsigned>(batch)), dim3(256), 0,
                       stream, delta, N, d, basis, static_cast<cplx*>(ws), K);
    return hipGetLastError();
}

}  // namespace ffk
