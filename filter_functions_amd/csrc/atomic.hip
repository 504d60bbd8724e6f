// atomic.hip -- K6: the concatenation rule, numeric.calculate_control_matrix_from_atomic
// (filter_functions/numeric.py:621-704):
//     R[a,l,w] = R^(0)[a,l,w] + sum_{g>=1} phases[g-1,w] * sum_k R^(g)[a,k,w] L^(g-1)[k,l]
// with R^(g) the control matrix of the g-th pulse, phases the cumulated total phase factors and
// L the cumulated Liouville propagators (real for Hermitian bases, else complex).
//
// HBM-streaming: every atomic control-matrix element is read exactly once (16 B per
// (g, a, k, w) + 16/(A N) B of phase), one lane per frequency so that all accesses are 16-byte
// coalesced (1 KiB per wave instruction); the N x N propagator of the current pulse is wave
// uniform (scalar loads).  grid = (omega tiles, noise operators, column tiles of LT basis
// elements); the pulse axis can additionally be split into `gsplit` slabs whose partial sums
// are combined in fixed order by a second pass when the omega axis alone cannot fill the chip.
// which = 'correlations' writes every summand (G, A, N, W) instead of their sum.
#include "ffk_internal.h"

namespace ffk {
namespace {

// INDEXED: sequences built from few distinct pulses (randomized benchmarking draws 1000 gates from
// 24 Cliffords, examples/randomized_benchmarking.py:76-81): `Ratomic` and `phases` are then tables
// over the T distinct pulses, `index[g]` names the pulse at position g, and the cumulated phase
// factors (pulse_sequence.py:1824, a (G-1, W) cumprod on the host in the reference) become a
// running product in registers.  The tables stay resident in L2 / Infinity Cache, so the kernel
// is no longer bound by streaming G atomic control matrices from HBM.
template <int LT, bool LCPLX, bool INDEXED>
__global__ __launch_bounds__(64) void from_atomic_kernel(const cplx* __restrict__ phases,
                                                         const cplx* __restrict__ Ratomic,
                                                         const int32_t* __restrict__ index,
                                                         const double* __restrict__ L, int G, int A,
                                                         int N, int W, int glen, int correlations,
                                                         cplx* __restrict__ out) {
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y;
    const int nlt = (N + LT - 1)/LT;
    const int l0 = (blockIdx.z % nlt)*LT;
    const int slab = blockIdx.z / nlt;
    const int g0 = slab*glen, g1 = min(G, g0 + glen);
    if (w >= W) return;
    const size_t pulse_stride = static_cast<size_t>(A)*N*W;
    cplx acc[LT];
#pragma unroll
    for (int j = 0; j < LT; ++j) acc[j] = {0.0, 0.0};
    // running product of the pulses' total phase factors (plain multiply/subtract like NumPy's
    // complex cumprod); a slab that does not start at 0 first replays the product up to g0
    cplx run = {1.0, 0.0};
    auto advance = [&](int g) {   // run <- run * total_phase[pulse at position g]
        const cplx tp = phases[static_cast<size_t>(index[g])*W + w];
        const double rr = run.re*tp.re, ii = run.im*tp.im, ri = run.re*tp.im, ir = run.im*tp.re;
        run.re = rr - ii;
        run.im = ri + ir;
    };
    if (INDEXED && g0 < G)   // (an empty trailing slab must not walk index[] past its end)
        for (int g = 0; g + 1 < g0; ++g) advance(g);
    for (int g = g0; g < g1; ++g) {
        const cplx* Rg = Ratomic + (INDEXED ? index[g] : g)*pulse_stride + static_cast<size_t>(a)*N*W + w;
        cplx step[LT];
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < LT; ++j) step[j] = (l0 + j < N) ? Rg[static_cast<size_t>(l0 + j)*W] : cplx{0.0, 0.0};
        } else {
#pragma unroll
            for (int j = 0; j < LT; ++j) step[j] = {0.0, 0.0};
            cplx ph;
            if (INDEXED) {
                advance(g - 1);
                ph = run;
            } else {
                ph = phases[static_cast<size_t>(g - 1)*W + w];
            }
            const double* Lg = L + static_cast<size_t>(g - 1)*N*N*(LCPLX ? 2 : 1);
            for (int k = 0; k < N; ++k) {
                const cplx v = cmul(ph, Rg[static_cast<size_t>(k)*W]);
#pragma unroll
                for (int j = 0; j < LT; ++j) {
                    if (l0 + j < N) {
                        if (LCPLX) {
                            const cplx q = {Lg[2*(k*N + l0 + j)], Lg[2*(k*N + l0 + j) + 1]};
                            cmac(step[j], q, v);
                        } else {
                            const double q = Lg[k*N + l0 + j];
                            step[j].re = fma(q, v.re, step[j].re);
                            step[j].im = fma(q, v.im, step[j].im);
                        }
                    }
                }
            }
        }
        if (correlations) {
            cplx* o = out + g*pulse_stride + static_cast<size_t>(a)*N*W + w;
#pragma unroll
            for (int j = 0; j < LT; ++j)
                if (l0 + j < N) o[static_cast<size_t>(l0 + j)*W] = step[j];
        } else {
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                acc[j].re += step[j].re;
                acc[j].im += step[j].im;
            }
        }
    }
    if (!correlations) {
        cplx* o = out + slab*pulse_stride + static_cast<size_t>(a)*N*W + w;
#pragma unroll
        for (int j = 0; j < LT; ++j)
            if (l0 + j < N) o[static_cast<size_t>(l0 + j)*W] = acc[j];
    }
}

template <bool LCPLX, bool INDEXED>
hipError_t launch_c(const cplx* phases, const cplx* Ratomic, const int32_t* index, const double* L,
                    int G, int A, int N, int W, int gsplit, int correlations, cplx* out,
                    hipStream_t stream) {
    const int glen = (G + gsplit - 1)/gsplit;
    const unsigned tiles = (W + 63)/64;
    if (N <= 4) {
        hipLaunchKernelGGL((from_atomic_kernel<4, LCPLX, INDEXED>), dim3(tiles, A, gsplit), dim3(64),
                           0, stream, phases, Ratomic, index, L, G, A, N, W, glen, correlations, out);
    } else if (N <= 16) {
        hipLaunchKernelGGL((from_atomic_kernel<16, LCPLX, INDEXED>), dim3(tiles, A, gsplit), dim3(64),
                           0, stream, phases, Ratomic, index, L, G, A, N, W, glen, correlations, out);
    } else {
        const int nlt = (N + 15)/16;
        hipLaunchKernelGGL((from_atomic_kernel<16, LCPLX, INDEXED>), dim3(tiles, A, gsplit*nlt),
                           dim3(64), 0, stream, phases, Ratomic, index, L, G, A, N, W, glen,
                           correlations, out);
    }
    return hipGetLastError();
}

}  // namespace

// number of pulse-axis slabs used for which='total' (1 = single pass straight into the output)
int from_atomic_gsplit(int G, int A, int N, int W) {
    const long waves = static_cast<long>((W + 63)/64)*A*((N + 15)/16);
    if (waves >= 2048 || G < 16) return 1;
    long s = (2048 + waves - 1)/waves;
    if (s > G/8) s = G/8;
    if (s < 1) s = 1;
    const long glen = (G + s - 1)/s;
    return static_cast<int>((G + glen - 1)/glen);   // no empty slabs: ceil(G/glen) of them cover G
}

size_t from_atomic_workspace_bytes(int G, int A, int N, int W) {
    const int s = from_atomic_gsplit(G, A, N, W);
    return s > 1 ? align_up(sizeof(cplx)*static_cast<size_t>(s)*A*N*W) : 256;
}

hipError_t launch_from_atomic(const cplx* phases, const cplx* Ratomic, const int32_t* index,
                              const double* L, int l_is_complex, int G, int A, int N, int W,
                              int correlations, cplx* out, void* ws, hipStream_t stream) {
    if (A > 65535) return hipErrorInvalidValue;
    const int gsplit = correlations ? 1 : from_atomic_gsplit(G, A, N, W);
    cplx* target = (gsplit > 1) ? static_cast<cplx*>(ws) : out;
    hipError_t err;
    if (index) {
        err = l_is_complex ? launch_c<true, true>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                  correlations, target, stream)
                           : launch_c<false, true>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                   correlations, target, stream);
    } else {
        err = l_is_complex ? launch_c<true, false>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                   correlations, target, stream)
                           : launch_c<false, false>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                    correlations, target, stream);
    }
    if (err != hipSuccess) return err;
    if (gsplit > 1)
        return launch_reduce_chunks(target, gsplit, static_cast<size_t>(A)*N*W, out, stream);
    return hipSuccess;
}


// ---- numeric.calculate_control_matrix_periodic (numeric.py:886-954) ----------------------------
//   R_G = R_1 S_G,  S_G = sum_{g<G} T^g,  T(w) = z(w) L,  z = exp(i w tau), L the period's
//   Liouville propagator.
// The reference sums the series in closed form with one N x N solve per frequency and falls back
// to the term-by-term sum where I - T is ill conditioned.  Here the sum is built by doubling,
//   S_2n = S_n + T^n S_n,   S_n+1 = I + T S_n      (S_n commutes with T),
// applied to the rows r_n = R_1 S_n directly:
//   r_2n = r_n + z^n (r_n L^n),   r_n+1 = R_1 + z (r_n L),
// so a step is one (A W) x N by N x N product with an omega-independent matrix: ~2 log2(G)
// streaming passes over the control matrix, no inverse, no conditioning branch.  L^n (N x N) and
// z^n (W) are advanced by their own small kernels.
namespace {

// out[a,l,w] = base[a,l,w] + zf[w] * sum_k r[a,k,w] M[k,l]; one lane per frequency
template <int LT, bool LCPLX>
__global__ __launch_bounds__(64) void periodic_step_kernel(const cplx* __restrict__ base,
                                                           const cplx* __restrict__ r,
                                                           const cplx* __restrict__ zf,
                                                           const double* __restrict__ M, int N, int W,
                                                           cplx* __restrict__ out) {
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y;
    const int l0 = blockIdx.z*LT;
    if (w >= W) return;
    const size_t row0 = static_cast<size_t>(a)*N*W + w;
    const cplx z = zf[w];
    cplx acc[LT];
#pragma unroll
    for (int j = 0; j < LT; ++j) acc[j] = {0.0, 0.0};
    for (int k = 0; k < N; ++k) {
        const cplx v = cmul(z, r[row0 + static_cast<size_t>(k)*W]);
#pragma unroll
        for (int j = 0; j < LT; ++j) {
            if (l0 + j < N) {
                if (LCPLX) {
                    const cplx q = {M[2*(k*N + l0 + j)], M[2*(k*N + l0 + j) + 1]};
                    cmac(acc[j], q, v);
                } else {
                    const double q = M[k*N + l0 + j];
                    acc[j].re = fma(q, v.re, acc[j].re);
                    acc[j].im = fma(q, v.im, acc[j].im);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LT; ++j)
        if (l0 + j < N) {
            const size_t o = row0 + static_cast<size_t>(l0 + j)*W;
            const cplx b = base[o];
            out[o] = {b.re + acc[j].re, b.im + acc[j].im};
        }
}

// C = A B for N x N matrices (real, or complex interleaved); one thread per entry
template <bool LCPLX>
__global__ void small_matmul_kernel(const double* __restrict__ Am, const double* __restrict__ Bm,
                                    int N, double* __restrict__ C) {
    const int e = blockIdx.x*blockDim.x + threadIdx.x;
    if (e >= N*N) return;
    const int i = e / N, j = e % N;
    if (LCPLX) {
        cplx acc = {0.0, 0.0};
        for (int k = 0; k < N; ++k) {
            const cplx x = {Am[2*(i*N + k)], Am[2*(i*N + k) + 1]};
            const cplx y = {Bm[2*(k*N + j)], Bm[2*(k*N + j) + 1]};
            cmac(acc, x, y);
        }
        C[2*e] = acc.re;
        C[2*e + 1] = acc.im;
    } else {
        double acc = 0.0;
        for (int k = 0; k < N; ++k) acc = fma(Am[i*N + k], Bm[k*N + j], acc);
        C[e] = acc;
    }
}

// (other may be zn itself: squaring)
__global__ void phase_product_kernel(cplx* zn, const cplx* other, int W) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= W) return;
    zn[w] = cmul(zn[w], other[w]);
}

template <bool LCPLX>
void periodic_step(const cplx* base, const cplx* r, const cplx* zf, const double* M, int A, int N,
                   int W, cplx* out, hipStream_t stream) {
    const unsigned tiles = (W + 63)/64;
    if (N <= 4)
        hipLaunchKernelGGL((periodic_step_kernel<4, LCPLX>), dim3(tiles, A, 1), dim3(64), 0, stream,
                           base, r, zf, M, N, W, out);
    else
        hipLaunchKernelGGL((periodic_step_kernel<16, LCPLX>), dim3(tiles, A, (N + 15)/16), dim3(64), 0,
                           stream, base, r, zf, M, N, W, out);
}

template <bool LCPLX>
hipError_t periodic_c(const cplx* z, const cplx* R1, const double* L, int repeats, int A, int N,
                      int W, cplx* out, void* ws, hipStream_t stream) {
    const size_t nR = static_cast<size_t>(A)*N*W;
    const size_t nL = static_cast<size_t>(N)*N*(LCPLX ? 2 : 1);
    unsigned char* p = static_cast<unsigned char*>(ws);
    cplx* buf[2] = {reinterpret_cast<cplx*>(p), reinterpret_cast<cplx*>(p + align_up(nR*sizeof(cplx)))};
    p += 2*align_up(nR*sizeof(cplx));
    double* Lb[2] = {reinterpret_cast<double*>(p), reinterpret_cast<double*>(p + align_up(nL*sizeof(double)))};
    p += 2*align_up(nL*sizeof(double));
    cplx* zn = reinterpret_cast<cplx*>(p);
    hipError_t err = hipMemcpyAsync(zn, z, sizeof(cplx)*W, hipMemcpyDeviceToDevice, stream);
    if (err != hipSuccess) return err;
    int top = 30;
    while (!((repeats >> top) & 1)) --top;
    // the steps left after the current one decide whether L^n and z^n are still needed
    int steps_left = 0;
    for (int b = top - 1; b >= 0; --b) steps_left += 1 + ((repeats >> b) & 1);
    const cplx* r = R1;          // r_1
    const double* Ln = L;        // L^1
    int cur = 0, lcur = 0;
    auto target = [&]() { return steps_left == 1 ? out : buf[cur]; };   // the last step writes `out`
    auto advance_powers = [&](const double* other_L, const cplx* other_z) {
        if (steps_left == 0) return;
        hipLaunchKernelGGL((small_matmul_kernel<LCPLX>), dim3((N*N + 255)/256), dim3(256), 0, stream, Ln,
                           other_L, N, Lb[lcur]);
        Ln = Lb[lcur];
        lcur ^= 1;
        hipLaunchKernelGGL(phase_product_kernel, dim3((W + 255)/256), dim3(256), 0, stream, zn, other_z, W);
    };
    for (int b = top - 1; b >= 0; --b) {
        cplx* o = target();
        periodic_step<LCPLX>(r, r, zn, Ln, A, N, W, o, stream);        // r_2n = r_n + z^n r_n L^n
        r = o;
        cur ^= 1;
        --steps_left;
        advance_powers(Ln, zn);                                        // L^2n, z^2n
        if ((repeats >> b) & 1) {
            o = target();
            periodic_step<LCPLX>(R1, r, z, L, A, N, W, o, stream);     // r_n+1 = R_1 + z r_n L
            r = o;
            cur ^= 1;
            --steps_left;
            advance_powers(L, z);                                      // L^(n+1), z^(n+1)
        }
    }
    if (repeats == 1) {
        err = hipMemcpyAsync(out, R1, sizeof(cplx)*nR, hipMemcpyDeviceToDevice, stream);
        if (err != hipSuccess) return err;
    }
    return hipGetLastError();
}

}  // namespace

size_t periodic_workspace_bytes(int A, int N, int W) {
    return 2*align_up(sizeof(cplx)*static_cast<size_t>(A)*N*W) +
           2*align_up(2*sizeof(double)*static_cast<size_t>(N)*N) + align_up(sizeof(cplx)*static_cast<size_t>(W));
}

hipError_t launch_periodic(const cplx* phases, const cplx* R1, const double* L, int l_is_complex,
                           int repeats, int A, int N, int W, cplx* out, void* ws, hipStream_t stream) {
    if (A > 65535 || (N + 15)/16 > 65535 || repeats < 1) return hipErrorInvalidValue;
    return l_is_complex ? periodic_c<true>(phases, R1, L, repeats, A, N, W, out, ws, stream)
                        : periodic_c<false>(phases, R1, L, repeats, A, N, W, out, ws, stream);
}


// ---- Hilbert-space twin: calculate_noise_operators_from_atomic (numeric.py:377-453) ------------
//   B(w, a) = B^(0)(w, a) + sum_{g >= 1} phases[g-1, w] P_{g-1}^dag B^(g)(w, a) P_{g-1}
// One wavefront per (w, a): the d x d operator, the propagator and the half product live in LDS.
namespace {
__global__ __launch_bounds__(64) void noise_ops_from_atomic_kernel(
    const cplx* __restrict__ phases, const cplx* __restrict__ atomic, const cplx* __restrict__ props,
    int G, int W, int A, int d, cplx* __restrict__ out) {
    __shared__ cplx Bg[kMaxD*kMaxD], P[kMaxD*kMaxD], T[kMaxD*kMaxD];
    const int lane = threadIdx.x;
    const int w = blockIdx.x / A, a = blockIdx.x % A;
    const int dd = d*d;
    constexpr int kPer = (kMaxD*kMaxD + 63)/64;
    cplx acc[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int e = lane + 64*k;
        acc[k] = e < dd ? atomic[(static_cast<size_t>(w)*A + a)*dd + e] : cplx{0.0, 0.0};
    }
    for (int g = 1; g < G; ++g) {
        const cplx ph = phases[static_cast<size_t>(g - 1)*W + w];
        const cplx* src = atomic + ((static_cast<size_t>(g)*W + w)*A + a)*dd;
        __syncthreads();
        for (int e = lane; e < dd; e += 64) {
            Bg[e] = src[e];
            P[e] = props[static_cast<size_t>(g - 1)*dd + e];
        }
        __syncthreads();
        for (int e = lane; e < dd; e += 64) {     // T = B P
            const int i = e / d, j = e % d;
            cplx t = {0.0, 0.0};
            for (int k = 0; k < d; ++k) cmac(t, Bg[i*d + k], P[k*d + j]);
            T[e] = t;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {          // acc += ph P^dag T
            const int e = lane + 64*k;
            if (e < dd) {
                const int i = e / d, j = e % d;
                cplx t = {0.0, 0.0};
                for (int m = 0; m < d; ++m) cmac_conj(t, P[m*d + i], T[m*d + j]);
                cmac(acc[k], ph, t);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int e = lane + 64*k;
        if (e < dd) out[(static_cast<size_t>(w)*A + a)*dd + e] = acc[k];
    }
}
}  // namespace

hipError_t launch_noise_ops_from_atomic(const cplx* phases, const cplx* atomic, const cplx* props,
                                        int G, int W, int A, int d, cplx* out, hipStream_t stream) {
    const size_t blocks = static_cast<size_t>(W)*A;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(noise_ops_from_atomic_kernel, dim3(static_cast<unsigned>(blocks)), dim3(64), 0,
                       stream, phases, atomic, props, G, W, A, d, out);
    return hipGetLastError();
}

}  // namespace ffk
