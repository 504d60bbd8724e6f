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
#include <algorithm>
#include <cstdlib>

#include "ffk_internal.h"
#include "ffk_mfma_util.h"

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
                                                         cplx* __restrict__ out,
                                                         const cplx* const* __restrict__ Rtab) {
    // Rtab (INDEXED only, may be NULL): the distinct pulses' control matrices where they lie (one
    // device pointer each) instead of one contiguous table -- resident results are not copied
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
        const cplx* Rg = ((INDEXED && Rtab) ? Rtab[index[g]] : Ratomic + (INDEXED ? index[g] : g)*pulse_stride) +
                         static_cast<size_t>(a)*N*W + w;
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

// Slab reduction and fidelity filter function in one launch for few rows (A N <= 16): one thread per
// frequency sums the partial control matrices in slab order (the order of reduce_chunks_kernel),
// writes R and forms F[a,b] = sum_k conj(R[a,k]) R[b,k] from registers with the arithmetic of
// ff_fidelity_kernel (a <= b summed over k in order, mirrored, diagonal imaginary part 0): results
// are bit-identical to the two separate launches.
constexpr int kReduceFfRows = 16;
__global__ __launch_bounds__(128) void reduce_ff_small_kernel(const cplx* __restrict__ part, int nslab,
                                                              int A, int N, int W,
                                                              cplx* __restrict__ R,
                                                              cplx* __restrict__ F) {
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= W) return;
    const int rows = A*N;
    const size_t slab = static_cast<size_t>(rows)*W;
    cplx r[kReduceFfRows];
#pragma unroll
    for (int e = 0; e < kReduceFfRows; ++e) {
        if (e < rows) {
            cplx acc = part[static_cast<size_t>(e)*W + w];
            for (int z = 1; z < nslab; ++z) {
                const cplx v = part[static_cast<size_t>(z)*slab + static_cast<size_t>(e)*W + w];
                acc.re += v.re;
                acc.im += v.im;
            }
            r[e] = acc;
            R[static_cast<size_t>(e)*W + w] = acc;
        }
    }
    for (int a = 0; a < A; ++a)
        for (int b = a; b < A; ++b) {
            cplx acc = {0.0, 0.0};
#pragma unroll
            for (int e = 0; e < kReduceFfRows; ++e) {      // (registers: every index a compile-time constant)
                const int k = e;
                if (k < N) {
                    cplx ra = {0.0, 0.0}, rb = {0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < kReduceFfRows; ++q) {
                        if (q == a*N + k) ra = r[q];
                        if (q == b*N + k) rb = r[q];
                    }
                    cmac_conj(acc, ra, rb);
                }
            }
            if (a == b) acc.im = 0.0;
            F[(static_cast<size_t>(a)*A + b)*W + w] = acc;
            if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {acc.re, -acc.im};
        }
}

// ---- the table rule for few rows, one block per 64 frequencies ------------------------------------------
// A sequence drawn from T distinct pulses with A N <= 16 rows per control matrix (a single-qubit
// randomized-benchmarking sequence: T = 24, A N = 4) does ~60 instructions of arithmetic per position
// and frequency; in from_atomic_kernel every position costs a dependent trip to L2 (index -> table
// row), and a slab that starts at position g0 first replays g0 phase products at one such trip each.
// Here a block owns 64 frequencies and ALL positions: the T control matrices and total phase
// factors of those frequencies are staged in LDS once (T (A N + 1) KiB), sixteen wavefronts take
// one slab of positions each, the slabs' phase products are combined by a two-level product
// (slab-local products, then the prefix over slabs: the running product of NumPy's cumprod
// re-associated at slab boundaries, ~1e-16 relative), and the slab sums are added in slab order in
// LDS, from where the block writes R and -- optionally -- the fidelity filter function with the
// arithmetic of ff_fidelity_kernel.  One launch instead of rule + reduction + filter function.
template <int A, int N, bool LCPLX>
__global__ __launch_bounds__(1024) void from_atomic_block_kernel(
    const cplx* __restrict__ phases, const cplx* __restrict__ Ratomic, const cplx* const* __restrict__ Rtab,
    const int32_t* __restrict__ index, const double* __restrict__ L, int G, int T, int W, int glen,
    cplx* __restrict__ out, cplx* __restrict__ F, const double* __restrict__ Lpulse) {
    constexpr int ROWS = A*N;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nslab = blockDim.x >> 6;
    cplx* Rs = reinterpret_cast<cplx*>(lds_raw);                       // [T][ROWS][64]; later red[nslab][ROWS][64]
    const size_t big = static_cast<size_t>(max(T*ROWS, nslab*ROWS + ROWS))*64;
    cplx* Ps = Rs + big;                                               // [T][64]
    cplx* Pslab = Ps + static_cast<size_t>(T)*64;                      // [nslab][64]
    const int w = blockIdx.x*64 + lane;
    const int wc = w < W ? w : W - 1;
    const size_t pulse_stride = static_cast<size_t>(ROWS)*W;
    for (int e = wave; e < T*ROWS; e += nslab) {
        const int k = e / ROWS, r = e % ROWS;
        const cplx* src = (Rtab ? Rtab[k] : Ratomic + k*pulse_stride) + static_cast<size_t>(r)*W;
        Rs[static_cast<size_t>(e)*64 + lane] = src[wc];
    }
    for (int k = wave; k < T; k += nslab) Ps[k*64 + lane] = phases[static_cast<size_t>(k)*W + wc];
    __syncthreads();
    const int g0 = wave*glen, g1 = min(G, g0 + glen);
    auto times = [](cplx& acc, cplx tp) {           // plain multiply/subtract like NumPy's complex cumprod
        const double rr = acc.re*tp.re, ii = acc.im*tp.im, ri = acc.re*tp.im, ir = acc.im*tp.re;
        acc.re = rr - ii;
        acc.im = ri + ir;
    };
    // product of the total phases of this slab's positions, then of all earlier slabs
    // (index[] is wave uniform: scalar loads, a trip to the scalar cache per position)
    // (the slab's pulse numbers in ONE coalesced load -- lane l holds position g0 + l -- and a v_readlane per position
    // where the slab has at most 64 positions; round 5: a scalar load and a full wait per position, twice)
    const bool slab_in_lanes = glen <= 64;
    const int my_index = slab_in_lanes && g0 + lane < g1 ? index[g0 + lane] : 0;
    auto pulse_at = [&](int g) -> int {
        return slab_in_lanes ? __builtin_amdgcn_readlane(my_index, g - g0) : index[g];
    };
    cplx loc = {1.0, 0.0};
    for (int g = g0; g < g1; ++g) times(loc, Ps[pulse_at(g)*64 + lane]);
    Pslab[wave*64 + lane] = loc;
    __syncthreads();
    cplx run = {1.0, 0.0};
    for (int s = 0; s < wave; ++s) times(run, Pslab[s*64 + lane]);
    cplx acc[ROWS];
#pragma unroll
    for (int e = 0; e < ROWS; ++e) acc[e] = {0.0, 0.0};
    constexpr int LN = N*N*(LCPLX ? 2 : 1);
    if (!LCPLX && Lpulse != nullptr) {
        // Round 6 (Hermitian bases): the slab by a BACKWARD recurrence on the DISTINCT pulses' own propagators.  With v_g the control
        // matrix of the pulse at position g, p_g its total phase, L_g the Liouville representation of its propagator
        // and P_g, M_g the products of the phases / representations of the positions before g,
        //     sum_{g in slab} P_g v_g M_g = P_{g0} S_{g0} M_{g0},     S_g = v_g + p_g (S_{g+1} L_g),   S_{g1} = 0:
        // every operand of a position comes from a table of the T distinct pulses -- the representations, T x 128
        // bytes, through scalar loads that stay in the scalar cache -- and only the slab's first position needs a
        // cumulative propagator.  Rounds 2-5 read ONE CUMULATIVE propagator per position (two s_load_dwordx16 from
        // 128 KB that 16 wavefronts x 128 blocks stream through a 16-KB scalar cache, every wait lgkmcnt(0)): ~0.5 us
        // of stall per position on 0.43 us of arithmetic, 57 us for 27 us of issue (profiles/r04_m_*, r06_e_*).
        // 48 instead of 60 vector instructions per position.  The sum is re-associated (62 more orthogonal 4 x 4
        // products in a row than the reference's form): ~1e-15 relative.
        // (Measured and not kept in round 6, same arithmetic as before: the cumulative propagators through vector
        // loads, every lane the same address, 57.6 -> 80.6 us; through a per-wavefront LDS ring filled by LDS-DMA and
        // broadcast reads, 75.9 us -- 8 KB into the vector registers per position and wavefront either way.)
        cplx S[ROWS];
#pragma unroll
        for (int e = 0; e < ROWS; ++e) S[e] = {0.0, 0.0};
        int k_next = g0 < g1 ? pulse_at(g1 - 1) : 0;
        double Lnext[LN];
        auto request = [&](int k) {
            const double* Lk = Lpulse + static_cast<size_t>(k)*LN;
#pragma unroll
            for (int e = 0; e < LN; ++e) Lnext[e] = Lk[e];
        };
        if (g0 < g1) request(k_next);
#if defined(FFK_RULE_ABLATE)     /* tuning: the launch without its recurrence (what do staging, products and reduction cost?) */
        for (int g = g1 - 1; g >= g0 + (g1 - g0) - FFK_RULE_ABLATE; --g) {
#else
        for (int g = g1 - 1; g >= g0; --g) {
#endif
            const int k = k_next;
            double Lk[LN];
#pragma unroll
            for (int e = 0; e < LN; ++e) Lk[e] = Lnext[e];
            if (g > g0) {
                k_next = pulse_at(g - 1);
                request(k_next);
            }
            const cplx* Rg = Rs + static_cast<size_t>(k)*ROWS*64 + lane;
            const cplx p = Ps[k*64 + lane];
#pragma unroll
            for (int a = 0; a < A; ++a) {
                cplx t[N];
#pragma unroll
                for (int j = 0; j < N; ++j) t[j] = {0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < N; ++kk)
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        if (LCPLX) {
                            const cplx q = {Lk[2*(kk*N + j)], Lk[2*(kk*N + j) + 1]};
                            cmac(t[j], q, S[a*N + kk]);
                        } else {
                            const double q = Lk[kk*N + j];
                            t[j].re = fma(q, S[a*N + kk].re, t[j].re);
                            t[j].im = fma(q, S[a*N + kk].im, t[j].im);
                        }
                    }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const cplx v = Rg[(a*N + j)*64];
                    S[a*N + j] = {fma(p.re, t[j].re, fma(-p.im, t[j].im, v.re)),
                                  fma(p.re, t[j].im, fma(p.im, t[j].re, v.im))};
                }
            }
        }
        // the slab's sum: P_{g0} S M_{g0} (M_0 = 1)
        if (g0 < g1) {
            if (g0 == 0) {
#pragma unroll
                for (int e = 0; e < ROWS; ++e) acc[e] = S[e];
            } else {
                const double* Lg = L + static_cast<size_t>(g0 - 1)*LN;
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    cplx step[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) step[j] = {0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < N; ++kk) {
                        const cplx v = cmul(run, S[a*N + kk]);
#pragma unroll
                        for (int j = 0; j < N; ++j) {
                            if (LCPLX) {
                                const cplx q = {Lg[2*(kk*N + j)], Lg[2*(kk*N + j) + 1]};
                                cmac(step[j], q, v);
                            } else {
                                const double q = Lg[kk*N + j];
                                step[j].re = fma(q, v.re, step[j].re);
                                step[j].im = fma(q, v.im, step[j].im);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < N; ++j) acc[a*N + j] = step[j];
                }
            }
        }
    } else {
    // The position's propagator is wave uniform (scalar loads); that of position g + 1 is requested before position g
    // is worked on, so that a trip to L2 overlaps the arithmetic instead of preceding it.
    double Lnext[LN];
    int k_next = g0 < g1 ? pulse_at(g0) : 0;
    auto request = [&](int g) {
        const double* Lg = L + static_cast<size_t>(g > 0 ? g - 1 : 0)*LN;
#pragma unroll
        for (int e = 0; e < LN; ++e) Lnext[e] = Lg[e];
    };
    if (g0 < g1) request(g0);
    for (int g = g0; g < g1; ++g) {
        const int k = k_next;
        double Lg[LN];
#pragma unroll
        for (int e = 0; e < LN; ++e) Lg[e] = Lnext[e];
        if (g + 1 < g1) {
            k_next = pulse_at(g + 1);
            request(g + 1);
        }
        const cplx* Rg = Rs + static_cast<size_t>(k)*ROWS*64 + lane;
        if (g == 0) {
#pragma unroll
            for (int e = 0; e < ROWS; ++e) {
                const cplx v = Rg[e*64];
                acc[e].re += v.re;
                acc[e].im += v.im;
            }
        } else {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                cplx step[N];
#pragma unroll
                for (int j = 0; j < N; ++j) step[j] = {0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < N; ++kk) {
                    const cplx v = cmul(run, Rg[(a*N + kk)*64]);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        if (LCPLX) {
                            const cplx q = {Lg[2*(kk*N + j)], Lg[2*(kk*N + j) + 1]};
                            cmac(step[j], q, v);
                        } else {
                            const double q = Lg[kk*N + j];
                            step[j].re = fma(q, v.re, step[j].re);
                            step[j].im = fma(q, v.im, step[j].im);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    acc[a*N + j].re += step[j].re;
                    acc[a*N + j].im += step[j].im;
                }
            }
        }
        times(run, Ps[k*64 + lane]);                // run <- run * total_phase[pulse at position g]
    }
    }
    __syncthreads();                                // tables dead: the slab sums take their place
    cplx* red = Rs;
#pragma unroll
    for (int e = 0; e < ROWS; ++e) red[(static_cast<size_t>(wave)*ROWS + e)*64 + lane] = acc[e];
    __syncthreads();
    cplx* rsum = red + static_cast<size_t>(nslab)*ROWS*64;             // [ROWS][64]
    for (int e = wave; e < ROWS; e += nslab) {
        cplx v = red[static_cast<size_t>(e)*64 + lane];
        for (int s = 1; s < nslab; ++s) {
            const cplx u = red[(static_cast<size_t>(s)*ROWS + e)*64 + lane];
            v.re += u.re;
            v.im += u.im;
        }
        rsum[e*64 + lane] = v;
        if (w < W) out[static_cast<size_t>(e)*W + w] = v;
    }
    if (!F) return;
    __syncthreads();
    for (int pair = wave; pair < A*A; pair += nslab) {
        const int a = pair / A, b = pair % A;
        if (a > b) continue;
        cplx f = {0.0, 0.0};
        for (int kk = 0; kk < N; ++kk) cmac_conj(f, rsum[(a*N + kk)*64 + lane], rsum[(b*N + kk)*64 + lane]);
        if (a == b) f.im = 0.0;
        if (w < W) {
            F[(static_cast<size_t>(a)*A + b)*W + w] = f;
            if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {f.re, -f.im};
        }
    }
}

// ---- the front of a sequence concatenation for small d in ONE launch --------------------------------
// (pulse_sequence.py:1812-1840: total propagators of the positions, their running products, the
// Liouville representations of the first G - 1 of them; plus the distinct pulses' total phase factors
// exp(i omega tau_k), util.cexp, and -- for a resident result -- its copy of the grid).
// Block 0: thread g holds U_g = table[index[g]]; an inclusive Hillis-Steele scan over the positions
// in LDS (ceil(log2 G) steps, Q_{g+1} = U_g ... U_0) replaces gather + scan kernels; then thread g
// forms L_g[i,j] = tr(Q_{g+1}^dag C_i Q_{g+1} C_j) for g < G - 1 (superoperator.py:51-84) from the
// basis in LDS.  Blocks >= 1: phases (T, W) and the grid copy.  G <= 1024, d <= 4, N <= 16.
template <int D>
__global__ __launch_bounds__(1024) void sequence_front_kernel(
    const cplx* __restrict__ U, const int32_t* __restrict__ index, int G, const cplx* __restrict__ basis,
    int N, int l_is_complex, cplx* __restrict__ Q, double* __restrict__ L,
    const double* __restrict__ tau, const double* __restrict__ omega, int T, int W,
    cplx* __restrict__ phases, double* __restrict__ omega_copy, double* __restrict__ Lpulse) {
    constexpr int DD = D*D;
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x > 0) {
        const size_t e = static_cast<size_t>(blockIdx.x - 1)*blockDim.x + threadIdx.x;
        if (e < static_cast<size_t>(T)*W) {
            const int k = static_cast<int>(e / W), w = static_cast<int>(e % W);
            phases[e] = cexp(omega[w]*tau[k]);
            if (omega_copy && k == 0) omega_copy[w] = omega[w];
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* buf0 = reinterpret_cast<cplx*>(lds_raw);             // [G][DD]
    cplx* buf1 = buf0 + static_cast<size_t>(blockDim.x)*DD;    // [G][DD]
    cplx* Cs = buf1 + static_cast<size_t>(blockDim.x)*DD;      // [N][DD]
    const int g = threadIdx.x;
    for (int e = g; e < N*DD; e += blockDim.x) Cs[e] = basis[e];
    cplx M[DD];
#pragma unroll
    for (int e = 0; e < DD; ++e) M[e] = {e / D == e % D ? 1.0 : 0.0, 0.0};
    if (g < G) {
        const cplx* src = U + static_cast<size_t>(index[g])*DD;
#pragma unroll
        for (int e = 0; e < DD; ++e) M[e] = src[e];
    }
    // The scan buffers hold entry e of position g at [e][g]: consecutive threads touch consecutive
    // 16-byte slots.  ([g][e], a matrix per thread, put the 64 lanes of every LDS access 64 bytes
    // apart -- four lanes per bank group -- and a scan step cost 3.2 us, 32 of this kernel's 47 us at
    // 1000 positions: profiles/r04_m_*.)
    const int stride = blockDim.x;
    cplx* cur = buf0;
    cplx* nxt = buf1;
#pragma unroll
    for (int e = 0; e < DD; ++e) cur[e*stride + g] = M[e];
    __syncthreads();
    for (int shift = 1; shift < G; shift <<= 1) {
        if (g >= shift && g < G) {
            // M <- M (the later factors) x cur[g - shift] (the earlier ones)
            cplx E[DD], P[DD];
#pragma unroll
            for (int e = 0; e < DD; ++e) E[e] = cur[e*stride + g - shift];
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    cplx acc = {0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < D; ++k) cmac(acc, M[i*D + k], E[k*D + j]);
                    P[i*D + j] = acc;
                }
#pragma unroll
            for (int e = 0; e < DD; ++e) M[e] = P[e];
        }
#pragma unroll
        for (int e = 0; e < DD; ++e) nxt[e*stride + g] = M[e];
        __syncthreads();
        cplx* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (g == 0)
#pragma unroll
        for (int e = 0; e < DD; ++e) Q[e] = {e / D == e % D ? 1.0 : 0.0, 0.0};
    if (g < G) {
#pragma unroll
        for (int e = 0; e < DD; ++e) Q[static_cast<size_t>(g + 1)*DD + e] = M[e];
    }
    // representation of a matrix M:  CB_i = M^dag C_i M,  L[i,j] = tr(CB_i C_j)  -> dst (N, N), f64 or c128
    auto represent = [&](const cplx (&Mx)[DD], double* dst) {
        for (int i = 0; i < N; ++i) {
            const cplx* Ci = Cs + i*DD;
            cplx CM[DD], CB[DD];
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    cplx acc = {0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < D; ++k) cmac(acc, Ci[r*D + k], Mx[k*D + c]);
                    CM[r*D + c] = acc;
                }
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    cplx acc = {0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < D; ++k) cmac_conj(acc, Mx[k*D + a], CM[k*D + b]);
                    CB[a*D + b] = acc;
                }
            for (int j = 0; j < N; ++j) {
                const cplx* Cj = Cs + j*DD;
                cplx acc = {0.0, 0.0};
#pragma unroll
                for (int a = 0; a < D; ++a)
#pragma unroll
                    for (int b = 0; b < D; ++b) cmac(acc, CB[a*D + b], Cj[b*D + a]);
                const size_t o = static_cast<size_t>(i)*N + j;
                if (l_is_complex) {
                    dst[2*o] = acc.re;
                    dst[2*o + 1] = acc.im;
                } else {
                    dst[o] = acc.re;
                }
            }
        }
    };
    // L_g = representation of Q_{g+1} = M
    if (g + 1 < G) represent(M, L + static_cast<size_t>(g)*N*N*(l_is_complex ? 2 : 1));
    // round 6: the representations of the T distinct pulses' own propagators as well (the rule kernel's backward
    // recurrence reads these -- a table of a few KB that stays in the scalar cache -- instead of one cumulative
    // propagator per position)
    if (Lpulse != nullptr)
        for (int k = g; k < T; k += blockDim.x) {
            cplx Uk[DD];
#pragma unroll
            for (int e = 0; e < DD; ++e) Uk[e] = U[static_cast<size_t>(k)*DD + e];
            represent(Uk, Lpulse + static_cast<size_t>(k)*N*N*(l_is_complex ? 2 : 1));
        }
}

template <bool LCPLX, bool INDEXED>
hipError_t launch_c(const cplx* phases, const cplx* Ratomic, const int32_t* index, const double* L,
                    int G, int A, int N, int W, int gsplit, int correlations, cplx* out,
                    hipStream_t stream, const cplx* const* Rtab) {
    const int glen = (G + gsplit - 1)/gsplit;
    const unsigned tiles = (W + 63)/64;
    if (N <= 4) {
        hipLaunchKernelGGL((from_atomic_kernel<4, LCPLX, INDEXED>), dim3(tiles, A, gsplit), dim3(64),
                           0, stream, phases, Ratomic, index, L, G, A, N, W, glen, correlations, out, Rtab);
    } else if (N <= 16) {
        hipLaunchKernelGGL((from_atomic_kernel<16, LCPLX, INDEXED>), dim3(tiles, A, gsplit), dim3(64),
                           0, stream, phases, Ratomic, index, L, G, A, N, W, glen, correlations, out, Rtab);
    } else {
        const int nlt = (N + 15)/16;
        hipLaunchKernelGGL((from_atomic_kernel<16, LCPLX, INDEXED>), dim3(tiles, A, gsplit*nlt),
                           dim3(64), 0, stream, phases, Ratomic, index, L, G, A, N, W, glen,
                           correlations, out, Rtab);
    }
    return hipGetLastError();
}

}  // namespace

namespace {
int front_threads(int G) {
    int threads = 64;
    while (threads < G) threads <<= 1;
    return threads;
}
size_t front_lds_bytes(int G, int d, int N) {
    return (2*static_cast<size_t>(front_threads(G)) + N)*d*d*sizeof(cplx);
}
}  // namespace

// (the two scan buffers of block 0 must fit LDS: 1024 positions at d = 2, 512 at d = 3, 256 at d = 4)
bool sequence_front_supported(int d, int G, int N) {
    return d >= 2 && d <= 4 && G >= 1 && G <= 1024 && N <= 16 && front_lds_bytes(G, d, N) <= 144*1024;
}

hipError_t launch_sequence_front(const cplx* U, const int32_t* index, int G, int d, const cplx* basis,
                                 int N, int l_is_complex, cplx* Q, double* L, const double* tau,
                                 const double* omega, int T, int W, cplx* phases, double* omega_copy,
                                 hipStream_t stream, double* Lpulse) {
    if (!sequence_front_supported(d, G, N)) return hipErrorInvalidValue;
    const int threads = front_threads(G);
    const size_t lds = front_lds_bytes(G, d, N);
    const unsigned blocks = 1 + static_cast<unsigned>((static_cast<size_t>(T)*W + threads - 1)/threads);
#define FFK_FRONT(D)                                                                                   \
    case D: {                                                                                          \
        auto kern = sequence_front_kernel<D>;                                                          \
        if (lds > 48*1024) {                                                                           \
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                  \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                                 static_cast<int>(lds));                               \
            if (err != hipSuccess) return err;                                                         \
        }                                                                                              \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, stream, U, index, G, basis, N,      \
                           l_is_complex, Q, L, tau, omega, T, W, phases, omega_copy, Lpulse);          \
        break;                                                                                         \
    }
    switch (d) {
        FFK_FRONT(2) FFK_FRONT(3) FFK_FRONT(4)
        default: return hipErrorInvalidValue;
    }
#undef FFK_FRONT
    return hipGetLastError();
}

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
                              int correlations, cplx* out, void* ws, hipStream_t stream,
                              const cplx* const* Rtab, cplx* F, int T, const double* Lpulse) {
    if (A > 65535) return hipErrorInvalidValue;
    if (index && !correlations && T > 0 && N == 4 && A <= 4 && G >= 64) {
        // single-qubit rows, tables that fit LDS: one block per 64 frequencies does rule, reduction and F
        const int nslab = 16;
        const int glen = (G + nslab - 1)/nslab;
        const int rows = A*N;
        const size_t lds = (static_cast<size_t>(std::max(T*rows, nslab*rows + rows)) + T + nslab)*64*sizeof(cplx);
        if (lds <= 150*1024) {
            auto launch = [&](auto kern) -> hipError_t {
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    static_cast<int>(lds));
                if (e2 != hipSuccess) return e2;
                hipLaunchKernelGGL(kern, dim3((W + 63)/64), dim3(nslab*64), lds, stream, phases, Ratomic, Rtab,
                                   index, L, G, T, W, glen, out, F, Lpulse);
                return hipGetLastError();
            };
#define FFK_BLK(AA)                                                                             \
    case AA:                                                                                    \
        return l_is_complex ? launch(from_atomic_block_kernel<AA, 4, true>)                     \
                            : launch(from_atomic_block_kernel<AA, 4, false>);
            switch (A) {
                FFK_BLK(1) FFK_BLK(2) FFK_BLK(3) FFK_BLK(4)
                default: break;
            }
#undef FFK_BLK
        }
    }
    const int gsplit = correlations ? 1 : from_atomic_gsplit(G, A, N, W);
    cplx* target = (gsplit > 1) ? static_cast<cplx*>(ws) : out;
    hipError_t err;
    if (index) {
        err = l_is_complex ? launch_c<true, true>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                  correlations, target, stream, Rtab)
                           : launch_c<false, true>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                   correlations, target, stream, Rtab);
    } else {
        err = l_is_complex ? launch_c<true, false>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                   correlations, target, stream, nullptr)
                           : launch_c<false, false>(phases, Ratomic, index, L, G, A, N, W, gsplit,
                                                    correlations, target, stream, nullptr);
    }
    if (err != hipSuccess) return err;
    // F (optional, which = 'total'): the fidelity filter function of the sum.  For few rows (A N <= 16)
    // the slab reduction and F are one launch; otherwise the caller's F launch follows the reduction
    if (F && !correlations && gsplit > 1 && A*N <= kReduceFfRows) {
        hipLaunchKernelGGL(reduce_ff_small_kernel, dim3((W + 127)/128), dim3(128), 0, stream, target, gsplit,
                           A, N, W, out, F);
        return hipGetLastError();
    }
    if (gsplit > 1) {
        err = launch_reduce_chunks(target, gsplit, static_cast<size_t>(A)*N*W, out, stream);
        if (err != hipSuccess) return err;
    }
    if (F && !correlations) return launch_filter_function(out, A, N, W, 0, F, stream);
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
