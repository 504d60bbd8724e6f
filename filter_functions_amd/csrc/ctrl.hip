// ctrl.hip -- K3: the fused control-matrix accumulation, the kernel that carries ~98 % of
// PulseSequence.get_filter_function (numeric.calculate_control_matrix_from_scratch,
// filter_functions/numeric.py:707-881, hot loop :846-869; Hilbert-space twin :456-618).
//
// What it computes, per frequency w (one lane each) and noise operator a:
//     Y_a(w) = sum_g  T_g^dag [ (s_a Bbar_a^(g)) o ( e^{i w t_g} I^(g)(w) ) ] T_g ,
//     I^(g)_mn(w) = (e^{i (w + D_m - D_n) dt_g} - 1) / (i (w + D_m - D_n))
// i.e. the interaction-picture noise operator B~_a(w) of numeric.py:516-537.  The control matrix
// is its basis expansion R[a,k,w] = tr(Y_a(w) C_k), done once after the segment sum
// (post.hip) instead of inside it: d^2 + 2 d^3 complex MACs per (g, w, a) instead of the d^4 of
// the reference's 'o,jmn,omn,knm->jko' contraction, with identical results (the reference pins
// the two formulations against each other at 1e-14, tests/test_precision.py:313-353).
//
// Mapping:
//   * grid.x tiles omega in 64s, one frequency per lane; grid.z splits the segment axis into
//     chunks whose partial sums are reduced afterwards in fixed order (deterministic);
//     grid.y x waves enumerate tasks (a, column block jb of Y).
//   * Phase A: the waves of a block share the generated integral: wave w computes entries
//     w, w+nw, ... of e^{i w t_g} I^(g) (one sincos + one reciprocal each, the diagonal only
//     once) and parks them in LDS, [entry][lane] so that reads/writes are 16-byte, conflict free.
//     The same phase copies the segment's omega-independent operands (T_g and the block's
//     Bbar_a^(g), (1 + n_a) d^2 complex numbers) from HBM/L2 into LDS.
//   * Phase B, per row m:  X[m,:] = Bbar[m,:] o I'[m,:],  Z[m,:] = X[m,:] T,
//     Y[i,:] += conj(T[m,i]) Z[m,:].  Operands come from LDS as broadcast ds_read_b128 (all
//     lanes the same address): in-order LDS returns let the compiler keep many reads in flight
//     behind counted lgkmcnt waits -- the first version fed the FMAs from scalar loads and spent
//     60 % of its wave cycles in s_waitcnt lgkmcnt(0) (profiles/r01_b_*).
//   * Double-buffered LDS: one barrier per segment.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "ffk_internal.h"

namespace ffk {
namespace {

constexpr int kWaveKernelMaxD = 4;
constexpr int kLongSequenceD2 = 1024;   // d = 2 sequences from this length on take the one-wave kernel
// in-block segment split (template GS of the block kernel): instantiated for small d only
constexpr int kGsplit = 4;
constexpr int kGsplitMaxD = 4;
constexpr int kGsplitMaxNW = 4;
bool g_use_gsplit = true;
bool g_use_wave_kernel = false;   // tuning/testing: one-wave-per-block variant for d <= 4
int g_mfma_policy = 0;            // 0: matrix-core kernel for d >= 12, 1: never, 2: also d = 8

// MR = integral rows generated per LDS stage.  MR == D (one stage per segment, diagonal computed
// once, optionally double-buffered) whenever the D*D*64 tile fits the 160 KiB LDS; MR < D splits
// the rows over several single-buffered stages (D > 11).  NW = waves per block, a template
// parameter so that each wave's share of phase A is a compile-time list: straight-line code with
// immediate LDS offsets and instruction-level parallelism across entries (the run-time
// enumeration cost ~25 scalar instructions per entry and serialised the entries).
// LDS per buffer: [MR*D][64] integral tile, then [(1 + NA)][D*D] operands (T_g, Bbar of the
// block's noise operators), NA = accum_na(D, nwaves).
// Occupancy target handed to the register allocator (2nd __launch_bounds__ argument = minimum
// waves per SIMD).  Measured on MI355X (profiles/r01_c_*): for the d = 4 kernel 3 waves/SIMD
// (<= 168 VGPRs, LDS operands partly re-read instead of all kept in registers) beats both
// 2 waves/SIMD (256 VGPRs, -17 %) and 4 waves/SIMD (128 VGPRs, spills, 8x slower).  Blocks are
// therefore capped at 8 waves.  FFK_WPE overrides the target in tuning builds.
#if defined(FFK_WPE)
#define FFK_ACCUM_WPE(D) FFK_WPE
#else
#define FFK_ACCUM_WPE(D) ((D) <= 4 ? 3 : 2)
#endif

#ifndef FFK_PHASE_SKEW          /* 1: opposite phase order on the two halves of a block's waves: */
#define FFK_PHASE_SKEW 0        /* measured SLOWER at d = 8 (9.5 -> 12.3 ms), profiles/r02_d8_*    */
#endif
#ifndef FFK_TCOL_SGPR           /* 1: the wave's columns of T_g through scalar loads (d >= 6):     */
#define FFK_TCOL_SGPR 0         /* in this template the allocator answers with 752 spilled VGPRs;  */
#endif                          /* the idea lives in ctrl_pcr.hip, written around it               */
#ifndef FFK_SKEW_PRIO           /* issue priority of a wave while it generates (skewed build) */
#define FFK_SKEW_PRIO 1
#endif

// every supported dimension; a tuning build may restrict the instantiations (-DFFK_ONLY_D=4)
#ifdef FFK_ONLY_D
#define FFK_ALL_D(X) X(FFK_ONLY_D)
#else
#define FFK_ALL_D(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#endif

// Compile-time list of the integral entries (slots m*D + n) that wave WV of an NW-wave block
// generates: every NW-th slot, the diagonal (all diagonal entries coincide) only as slot 0.
template <int D, int NW, int WV>
struct EntryList {
    static constexpr int build(int* out) {
        int n = 0;
        for (int e = WV; e < D*D; e += NW) {
            if (e != 0 && e / D == e % D) continue;
            if (out) out[n] = e;
            ++n;
        }
        return n;
    }
    static constexpr int count = build(nullptr);
    struct Slots {
        int v[D*D];
    };
    static constexpr Slots make() {
        Slots s{};
        build(s.v);
        return s;
    }
    static constexpr Slots slots = make();
};

// GS > 1: the block additionally splits ITS segment chunk over GS groups of NW waves ("sub-chunks",
// each with its own LDS tiles, advancing in lock step) and adds their accumulators up through LDS
// before writing: GS x fewer partial sums in HBM (16 chunks x 3.1 MB at config 2 cost ~9 us of
// write tail here and as much again in the reduction kernel) at unchanged parallelism.
template <int D, int JB, int MR, int NBUF, int NW, int GS>
__global__ __launch_bounds__(NW*GS*64, FFK_ACCUM_WPE(D)) void ctrl_accumulate_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, int na_blk,
    cplx* __restrict__ Ypart) {
    static_assert(MR == D || NBUF == 1, "row-blocked stages are single buffered");
    static_assert(GS == 1 || (MR == D && NBUF == 2), "sub-chunks need the double-buffered layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int S = seg_stride(D);
    constexpr int NJ = D / JB;
    constexpr int NSTAGE = (D + MR - 1)/MR;
    constexpr int MUNROLL = D <= 8 ? D : 1;
    constexpr int TILE = MR*D*64;      // cplx per integral tile

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = GS == 1 ? 0 : wave_all / NW;      // sub-chunk of this wave
    const int wave = GS == 1 ? wave_all : wave_all % NW;
    const int tid = static_cast<int>(threadIdx.x) - sub*NW*64;   // thread index within the sub-chunk
    constexpr int nwaves = NW;
    constexpr int nthreads = NW*64;
    const int task0 = blockIdx.y*nwaves;
    const int task = task0 + wave;
    const bool active = task < A*NJ;
    const int alpha = active ? task / NJ : 0;
    const int jb = active ? task % NJ : 0;
    const int alpha0 = task0 / NJ;                    // first noise operator of this block
    const int n_alpha = min(na_blk, A - alpha0);      // operators staged by this block
    // SHARE: the per-segment trigonometry every entry needs -- e^{i w t_g} and sin/cos of
    // a = fl(w dt_g)/2 -- is computed ONCE per sub-chunk (by two designated waves, one segment
    // ahead of the integral generation, handed over through LDS) instead of by every wave: two
    // range-reduced sincos less on the critical path of each wave and segment.
    constexpr bool SHARE = (MR == D && NBUF == 2);
    constexpr int NTAB = SHARE ? 3 : 2;               // table rows resident in LDS
    constexpr int TRIG = SHARE ? 2*2*64 : 0;          // cplx: [2 slots][phase | (sin a, cos a)][lane]
    const int buf_stride = TILE + (1 + na_blk)*D*D;   // cplx per LDS buffer
    const int sub_stride = NBUF*buf_stride + NTAB*S/2 + TRIG;   // cplx per sub-chunk region
    cplx* lds = reinterpret_cast<cplx*>(lds_raw) + static_cast<size_t>(sub)*sub_stride;
    const int iw = blockIdx.x*64 + lane;
    const double om = omega[iw < W ? iw : W - 1];
    const int sub_len = (chunk_len + GS - 1)/GS;      // lock-step trip count of the block
    const int g0 = blockIdx.z*chunk_len + sub*sub_len;
    const int g1 = min(min(G, static_cast<int>(blockIdx.z + 1)*chunk_len), g0 + sub_len);

    cplx Y[D][JB];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) Y[i][j] = {0.0, 0.0};

    // LDS: [NBUF x (integral tile | operands)] [NTAB x table row] [shared trigonometry].  Table
    // rows are staged ahead of their use (slot = segment offset mod NTAB), so that nothing in
    // phase A waits on global or scalar memory: an earlier version fetched (dE, sin b, cos b) per
    // entry from memory and spent ~60 % of phase A in those round trips (profiles/r01_c_*).
    double* tabs = reinterpret_cast<double*>(lds + static_cast<size_t>(NBUF)*buf_stride);
    cplx* trig = reinterpret_cast<cplx*>(tabs + NTAB*S);
    constexpr int AHEAD = NTAB - 1;    // rows are fetched this many segments ahead
    auto stage_table = [&](int g, int slot) {   // all threads: global -> LDS copy of one table row
        if (g < g1) {
            const cplx* src = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S);
            cplx* dst = reinterpret_cast<cplx*>(tabs + slot*S);
            for (int e = tid; e < S/2; e += nthreads) dst[e] = src[e];
        }
    };
    // designated waves: trigonometry of the segment whose table row sits in `row_slot`
    constexpr int WAVE_P = NW - 1;
    constexpr int WAVE_S = NW >= 2 ? NW - 2 : 0;
    auto share_trig = [&](int row_slot, int trig_slot) {
        const double* st = tabs + row_slot*S;
        // (constants pinned to the point of use: hoisted out of the segment loop they would
        // occupy ~34 VGPRs across the contraction)
        if (wave == WAVE_P) {
            cplx ph;
            sincos_pi<true>(om*st[1], &ph.im, &ph.re);
            trig[trig_slot*128 + lane] = ph;
        }
        if (wave == WAVE_S) {
            double sa, ca;
            sincos_pi<true>(0.5*(om*st[0]), &sa, &ca);
            trig[trig_slot*128 + 64 + lane] = {sa, ca};
        }
    };

    // Phase A: this wave's share of e^{i w t_g} I^(g)[rows of stage][:] -> LDS, plus (stage 0)
    // the segment's operands T_g, Bbar_{alpha0..}^(g) and the table row of segment g + AHEAD -> LDS.
    // this thread's share of the staging copy, in flight across phase B.  Two doubles and an UNCONDITIONAL load (round 6):
    // as a `cplx` assigned under `if (e0 < n_ops) .. else if ..` the value lived in scratch memory, and every segment's
    // staging load was waited for on the spot -- global_load; s_waitcnt vmcnt(0); scratch_store -- in every
    // instantiation of this kernel (found through the same fault in ctrl_d2.hip's first build, profiles/r06_g_*)
    double staged_re = 0.0, staged_im = 0.0;
    // fetch: issue the staging loads; generate: compute this wave's integral entries (the two
    // halves can be called apart, so that the loads are in flight across whatever runs between)
    auto phase_a = [&](int g, int stage, int buf, int row_slot, int trig_slot, bool fetch = true,
                       bool generate = true) {
        const double* st = tabs + row_slot*S;                // LDS
        cplx* tile = lds + static_cast<size_t>(buf)*buf_stride;
        // staging copies: issue the global loads first, park them after the integral is done
        const cplx* src_ops = ops + static_cast<size_t>(g)*(1 + A)*D*D;
        const cplx* src_tab =
            reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g + AHEAD)*S);
        const int n_ops = (1 + n_alpha)*D*D;
        const int n_tab = (g + AHEAD < g1) ? S/2 : 0;
        const int e0 = tid;
        if (stage == 0 && fetch) {
            // (a thread without a share reads the first operand: a valid address, the value is not parked)
            const cplx* src = e0 < n_ops ? src_ops + (e0 < D*D ? e0 : e0 + alpha0*D*D)
                                         : (e0 < n_ops + n_tab ? src_tab + (e0 - n_ops) : src_ops);
            staged_re = src->re;
            staged_im = src->im;
        }
        if (!generate) return;
        const double dtg = st[0];
        // e^{i w t_g}, and the half-angle of the diagonal entry, a = fl(w dt)/2: every
        // off-diagonal entry follows from (sin a, cos a) and the segment's precomputed
        // (sin b, cos b) by angle addition (dE = 0, sin b = 0, cos b = 1 on the diagonal and for
        // exactly degenerate levels)
        cplx ph;
        double sa, ca;
        if constexpr (SHARE) {
            ph = trig[trig_slot*128 + lane];
            const cplx sc = trig[trig_slot*128 + 64 + lane];
            sa = sc.re;
            ca = sc.im;
        } else {
            ph = cexp(om*st[1]);
            sincos_pi(0.5*(om*dtg), &sa, &ca);
        }
        cplx* dst = tile + lane;
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        auto gen = [&](int slot, int e) {   // slot: LDS slot, e: matrix entry m*D + n
            const double* r = st + seg_rec(e);
#if defined(FFK_ABLATE) && FFK_ABLATE == 1   /* diagnostic build: no integral generation */
            dst[slot*64] = {om, dtg + ph.re + r[0]};
#else
            dst[slot*64] = phased_integral_aa(pf, r[0], r[1], r[2]);
#endif
        };
        if constexpr (MR == D) {
            // this wave's entries, fully unrolled per wave index
            auto run = [&](auto tag) {
                using EL = EntryList<D, NW, decltype(tag)::value>;
#pragma unroll
                for (int k = 0; k < EL::count; ++k) gen(EL::slots.v[k], EL::slots.v[k]);
            };
            switch (wave) {
#define FFK_WCASE(I)                                         \
    case I:                                                  \
        if constexpr (NW > I) run(std::integral_constant<int, I>{}); \
        break;
                FFK_WCASE(0) FFK_WCASE(1) FFK_WCASE(2) FFK_WCASE(3) FFK_WCASE(4) FFK_WCASE(5)
                FFK_WCASE(6) FFK_WCASE(7)
#undef FFK_WCASE
                default:
                    break;
            }
        } else {
            const int rows = min(MR, D - stage*MR);
            for (int e = wave; e < rows*D; e += nwaves) gen(e, stage*MR*D + e);
        }
    };
    // second half of the staging copy: park the loaded values in LDS (after phase B, so that the
    // global-load latency hides behind the contraction); `next_slot` receives row g + AHEAD
    auto park = [&](int g, int buf, int next_slot) {
        cplx* tile = lds + static_cast<size_t>(buf)*buf_stride;
        const cplx* src_ops = ops + static_cast<size_t>(g)*(1 + A)*D*D;
        const cplx* src_tab =
            reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g + AHEAD)*S);
        const int n_ops = (1 + n_alpha)*D*D;
        const int n_tab = (g + AHEAD < g1) ? S/2 : 0;
        const int e0 = tid;
        cplx* dst_ops = tile + TILE;
        cplx* dst_tab = reinterpret_cast<cplx*>(tabs + next_slot*S);
        if (e0 < n_ops)
            dst_ops[e0] = {staged_re, staged_im};
        else if (e0 < n_ops + n_tab)
            dst_tab[e0 - n_ops] = {staged_re, staged_im};
        for (int e = e0 + nthreads; e < n_ops + n_tab; e += nthreads) {
            if (e < n_ops)
                dst_ops[e] = src_ops[e < D*D ? e : e + alpha0*D*D];
            else
                dst_tab[e - n_ops] = src_tab[e - n_ops];
        }
    };

    // The wave's JB columns of T_g (the operand of Z = X T, the same for every row m of a segment)
    // through SCALAR loads: D*JB wave-uniform complex numbers in SGPRs feed v_fma_f64 directly
    // instead of D*JB broadcast ds_read_b128 per row.  At d = 8 (JB = 2) the LDS array is otherwise
    // as busy as the FP64 pipe: 40 ds_read_b128 (4 LDS cycles each, 8 waves per CU) against 160
    // v_fma_f64 (4 cycles each, 2 waves per SIMD) per row -- 1280 cycles both; with the columns in
    // SGPRs 24 reads remain.  Loaded for segment g + 1 at the end of segment g (the scheme of the producer/consumer kernels).
    constexpr bool TCOL = FFK_TCOL_SGPR && SHARE && D >= 6 && D*JB <= 16;
    cplx Tcol[TCOL ? D : 1][TCOL ? JB : 1];
    auto load_tcol = [&](int g) {
        if constexpr (TCOL) {
            const cplx* tg = ops + static_cast<size_t>(g)*(1 + A)*D*D + jb*JB;
#pragma unroll
            for (int n = 0; n < D; ++n)
#pragma unroll
                for (int j = 0; j < JB; ++j) Tcol[n][j] = tg[n*D + j];
        }
    };

    // Phase B, rows of one stage
    auto phase_b = [&](int g, int stage, int buf) {
        (void)g;
        const cplx* tile = lds + static_cast<size_t>(buf)*buf_stride;
        const cplx* src = tile + lane;
        const cplx* opT = tile + TILE;                               // T[n][j]
        const cplx* opB = opT + (1 + alpha - alpha0)*D*D;            // Bbar_alpha[m][n]
        const int m0 = stage*MR;
        const int m1 = min(D, m0 + MR);
#pragma unroll MUNROLL
        for (int m = m0; m < m1; ++m) {
            cplx X[D];
#pragma unroll
            for (int n = 0; n < D; ++n) {
                const int slot = (MR == D) ? ((m == n) ? 0 : m*D + n) : (m - m0)*D + n;
                X[n] = cmul(opB[m*D + n], src[slot*64]);
            }
            cplx Z[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) Z[j] = {0.0, 0.0};
#pragma unroll
            for (int n = 0; n < D; ++n)
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    if constexpr (TCOL)
                        cmac(Z[j], Tcol[n][j], X[n]);
                    else
                        cmac(Z[j], opT[n*D + jb*JB + j], X[n]);
                }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const cplx t = opT[m*D + i];
#pragma unroll
                for (int j = 0; j < JB; ++j) cmac_conj(Y[i][j], t, Z[j]);
            }
        }
    };

    if constexpr (SHARE) {
        // Software pipeline over the segments g = g0 + it of the sub-chunk; in iteration it:
        //   table row g+3 and operands g+1 : global -> registers (parked in LDS at the end),
        //   trigonometry of g+2            : designated waves -> LDS,
        //   integrals of g+1               : all waves -> LDS tile (buf ^ 1),
        //   contraction of g               : from LDS tile (buf).
        // Row slots rotate mod 3, trigonometry slots and tiles mod 2; one barrier per segment.
        stage_table(g0, 0);
        stage_table(g0 + 1, 1);
        __syncthreads();
        if (g0 < g1) share_trig(0, 0);
        __syncthreads();
        if (g0 < g1) {
            phase_a(g0, 0, 0, 0, 0);
            if (g0 + 1 < g1) share_trig(1, 1);
            park(g0, 0, 2);
        }
        __syncthreads();
        // every wave of the block runs sub_len iterations (barriers must match); a sub-chunk that
        // is shorter (last chunk) idles through the tail
        if (active && g0 < g1) load_tcol(g0);
        const int trip = GS == 1 ? g1 - g0 : sub_len;
#if FFK_PHASE_SKEW
        // waves w and w + 4 of a block share a SIMD
        const bool contract_first = NW*GS > 4 && ((wave_all >> 2) & 1) != 0;
#endif
        int r0 = 0;   // it mod 3
        for (int it = 0; it < trip; ++it) {
            const int g = g0 + it;
            const int buf = it & 1;
            const int r1 = r0 == 2 ? 0 : r0 + 1, r2 = r1 == 2 ? 0 : r1 + 1;
#if defined(FFK_ABLATE) && (FFK_ABLATE == 4 || FFK_ABLATE == 5)  /* diagnostic: contraction only */
            if (active && g < g1) phase_b(g, 0, buf);
#if FFK_ABLATE == 4
            __syncthreads();
#endif
            continue;
#endif
            if (g < g1) {
#if FFK_PHASE_SKEW
                // Half of the waves generate first and contract afterwards, the other half the
                // other way round: the latency-bound generation of one wave then overlaps the
                // FMA-bound contraction of its SIMD neighbour instead of all waves of the block
                // sitting in the same phase at the same time.  (Generation writes tile buf ^ 1,
                // contraction reads tile buf: any order is valid within the barrier interval.)
                // One copy of each phase in the instruction stream; the two half-steps run them in
                // the order of this wave's parity.  The generating wave runs at raised issue priority:
                // its dependent chains (argument reduction -> polynomial -> reciprocal) otherwise lose
                // every arbitration against the neighbour's independent FMAs (cf. ctrl_pq.hip).
                if (g + 1 < g1) phase_a(g + 1, 0, buf ^ 1, r1, buf ^ 1, true, false);
#pragma nounroll
                for (int half = 0; half < 2; ++half) {
                    if ((half == 0) != contract_first) {
                        if (g + 1 < g1) {
                            __builtin_amdgcn_s_setprio(FFK_SKEW_PRIO);
                            phase_a(g + 1, 0, buf ^ 1, r1, buf ^ 1, false, true);
                            if (g + 2 < g1) share_trig(r2, buf);
                            __builtin_amdgcn_s_setprio(0);
                        }
                    } else {
#if !(defined(FFK_ABLATE) && FFK_ABLATE == 3)
                        if (active) phase_b(g, 0, buf);
#endif
                    }
                }
                if (g + 1 < g1) park(g + 1, buf ^ 1, r0);
#else
                if (g + 1 < g1) {
                    phase_a(g + 1, 0, buf ^ 1, r1, buf ^ 1);
                    if (g + 2 < g1) share_trig(r2, buf);
                }
#if !(defined(FFK_ABLATE) && FFK_ABLATE == 3)  /* diagnostic build 3: no contraction */
                if (active) phase_b(g, 0, buf);
#endif
                if (active && g + 1 < g1) load_tcol(g + 1);
                if (g + 1 < g1) park(g + 1, buf ^ 1, r0);
#endif
            }
#if !(defined(FFK_ABLATE) && FFK_ABLATE == 6)  /* diagnostic build 6: no barrier (wrong results) */
            __syncthreads();
#endif
            r0 = r1;
        }
    } else {
        stage_table(g0, 0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int slot = (g - g0) & 1;
            for (int stage = 0; stage < NSTAGE; ++stage) {
                phase_a(g, stage, 0, slot, 0);
                if (stage == 0) park(g, 0, slot ^ 1);
                __syncthreads();
                if (active) phase_b(g, stage, 0);
                __syncthreads();
            }
        }
    }

    if constexpr (GS > 1) {
        // tree reduction of the sub-chunks' accumulators through LDS (all tiles are dead now)
        cplx* red = reinterpret_cast<cplx*>(lds_raw);
        constexpr int YSZ = D*JB*64;   // cplx per wave
#pragma unroll
        for (int stride = GS/2; stride >= 1; stride >>= 1) {
            if (sub >= stride && sub < 2*stride) {
                cplx* dst = red + static_cast<size_t>((sub - stride)*NW + wave)*YSZ + lane;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < JB; ++j) dst[(i*JB + j)*64] = Y[i][j];
            }
            __syncthreads();
            if (sub < stride) {
                const cplx* src = red + static_cast<size_t>(sub*NW + wave)*YSZ + lane;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        const cplx v = src[(i*JB + j)*64];
                        Y[i][j].re += v.re;
                        Y[i][j].im += v.im;
                    }
            }
            __syncthreads();
        }
        if (sub != 0) return;
    }

    if (active && iw < W) {
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*D*D)*W + iw;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < JB; ++j) out[static_cast<size_t>(i*D + jb*JB + j)*W] = Y[i][j];
    }
}

// ---------------------------------------------------------------------------------------------
// Small-d variant (D <= 4): ONE wavefront per block, one frequency per lane, AT noise operators
// per lane -- no barrier, no inter-wave sharing, perfect load balance.  v_fma_f64 issues back to
// back from a single wave at >90 % of the FP64 rate (tools/fp64_latency.hip), so instead of many
// thin waves the kernel runs one fat wave per SIMD (up to 512 VGPRs: the AT*D*D accumulators, all
// D*D integral entries and the operands in flight stay in registers).  The wave keeps a private,
// double-buffered copy of the segment's operands (T_g, Bbar_a^(g)) in LDS -- fetched one segment
// ahead by a single coalesced global load -- and reads them back as broadcasts.
// ---------------------------------------------------------------------------------------------
template <int D, int AT>
__global__ __launch_bounds__(64) void ctrl_accumulate_wave_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    constexpr int S = seg_stride(D);
    constexpr int NOPS = (1 + AT)*D*D;
    static_assert(NOPS <= 128, "operand block must fit two loads per lane");
    __shared__ __attribute__((aligned(16))) cplx opbuf[2][NOPS];

    const int lane = threadIdx.x;
    const int iw = blockIdx.x*64 + lane;
    const int alpha0 = blockIdx.y*AT;
    const int na = min(AT, A - alpha0);
    const int n_ops = (1 + na)*D*D;
    const double om = omega[iw < W ? iw : W - 1];
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);

    cplx Y[AT][D][D];
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Y[a][i][j] = {0.0, 0.0};

    auto src_index = [&](int e) { return e < D*D ? e : e + alpha0*D*D; };
    if (g0 < g1) {
        const cplx* src = ops + static_cast<size_t>(g0)*(1 + A)*D*D;
        if (lane < n_ops) opbuf[0][lane] = src[src_index(lane)];
        if (NOPS > 64 && lane + 64 < n_ops) opbuf[0][lane + 64] = src[src_index(lane + 64)];
    }

    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        // operands of the next segment: issue the loads now, park them at the end of the body
        cplx nxt0 = {0.0, 0.0}, nxt1 = {0.0, 0.0};
        const bool more = g + 1 < g1;
        if (more) {
            const cplx* src = ops + static_cast<size_t>(g + 1)*(1 + A)*D*D;
            if (lane < n_ops) nxt0 = src[src_index(lane)];
            if (NOPS > 64 && lane + 64 < n_ops) nxt1 = src[src_index(lane + 64)];
        }
        const double* st = segtab + static_cast<size_t>(g)*S;
        const double dtg = st[0];
        const cplx ph = cexp(om*st[1]);

        // e^{i w t_g} I^(g): the diagonal entries coincide (dE = 0); off-diagonal ones by angle
        // addition from the diagonal's half-angle a = fl(w dt)/2
        double sa, ca;
        sincos_pi(0.5*(om*dtg), &sa, &ca);
        cplx Ip[D][D];
        {
            const double q = 2.0*sa*rcp(om);
            cplx I = {q*ca, q*sa};
            if (om == 0.0) I = {dtg, 0.0};
            Ip[0][0] = cmul(ph, I);
        }
#pragma unroll
        for (int m = 0; m < D; ++m)
#pragma unroll
            for (int n = 0; n < D; ++n)
                if (m != n) {
                    const int e = m*D + n;
                    const double dE = st[seg_rec(e)];
                    Ip[m][n] = dE == 0.0 ? Ip[0][0]
                                         : cmul(ph, first_order_integral_aa(om, dE, dtg, sa, ca,
                                                                            st[seg_rec(e) + 1],
                                                                            st[seg_rec(e) + 2]));
                }

        const cplx* opT = opbuf[buf];
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            if (a < na) {
                const cplx* opB = opbuf[buf] + (1 + a)*D*D;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    cplx X[D];
#pragma unroll
                    for (int n = 0; n < D; ++n) X[n] = cmul(opB[m*D + n], m == n ? Ip[0][0] : Ip[m][n]);
                    cplx Z[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) Z[j] = {0.0, 0.0};
#pragma unroll
                    for (int n = 0; n < D; ++n)
#pragma unroll
                        for (int j = 0; j < D; ++j) cmac(Z[j], opT[n*D + j], X[n]);
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const cplx t = opT[m*D + i];
#pragma unroll
                        for (int j = 0; j < D; ++j) cmac_conj(Y[a][i][j], t, Z[j]);
                    }
                }
            }
        }
        if (more) {
            if (lane < n_ops) opbuf[buf ^ 1][lane] = nxt0;
            if (NOPS > 64 && lane + 64 < n_ops) opbuf[buf ^ 1][lane + 64] = nxt1;
        }
    }

    if (iw < W) {
#pragma unroll
        for (int a = 0; a < AT; ++a) {
            if (a < na) {
                cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha0 + a)*D*D)*W + iw;
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) out[static_cast<size_t>(i*D + j)*W] = Y[a][i][j];
            }
        }
    }
}

__host__ __device__ constexpr int accum_mr(int d) { return d <= 11 ? d : 8; }

template <typename K>
hipError_t launch_kernel(K kern, const dim3& grid, const dim3& block, const AccumGeometry& geo,
                         hipStream_t stream, const double* omega, int W, const double* segtab,
                         const cplx* ops, int G, int A, cplx* Ypart) {
    if (geo.lds_bytes > 48*1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             geo.lds_bytes);
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(kern, grid, block, geo.lds_bytes, stream, omega, W, segtab, ops, G, A,
                       geo.chunk_len, geo.na_blk, Ypart);
    return hipGetLastError();
}

template <int D, int NW>
hipError_t launch_dw(const double* omega, int W, const double* segtab, const cplx* ops,
                     int G, int A, const AccumGeometry& geo, cplx* Ypart,
                     hipStream_t stream) {
    constexpr int JB = accum_jb(D);
    constexpr int MR = accum_mr(D);
    const dim3 grid((W + 63)/64, geo.task_groups, geo.chunks);
    const dim3 block(NW*geo.gsplit*64);
    if constexpr (MR == D) {
        if constexpr (D <= kGsplitMaxD && NW <= kGsplitMaxNW) {
            if (geo.nbuf == 2 && geo.gsplit == kGsplit)
                return launch_kernel(ctrl_accumulate_kernel<D, JB, MR, 2, NW, kGsplit>, grid, block,
                                     geo, stream, omega, W, segtab, ops, G, A, Ypart);
        }
        if (geo.nbuf == 2)
            return launch_kernel(ctrl_accumulate_kernel<D, JB, MR, 2, NW, 1>, grid, block, geo,
                                 stream, omega, W, segtab, ops, G, A, Ypart);
    }
    return launch_kernel(ctrl_accumulate_kernel<D, JB, MR, 1, NW, 1>, grid, block, geo, stream,
                         omega, W, segtab, ops, G, A, Ypart);
}

template <int D, int AT>
hipError_t launch_wave(const double* omega, int W, const double* segtab, const cplx* ops, int G,
                       int A, const AccumGeometry& geo, cplx* Ypart, hipStream_t stream) {
    const dim3 grid((W + 63)/64, geo.task_groups, geo.chunks);
    hipLaunchKernelGGL((ctrl_accumulate_wave_kernel<D, AT>), grid, dim3(64), 0, stream, omega, W,
                       segtab, ops, G, A, geo.chunk_len, Ypart);
    return hipGetLastError();
}

template <int D>
hipError_t launch_d(const double* omega, int W, const double* segtab, const cplx* ops, int G,
                    int A, const AccumGeometry& geo, cplx* Ypart, hipStream_t stream) {
    if constexpr (D <= kWaveKernelMaxD) {
        if (geo.wave_kernel) {
            switch (geo.na_blk) {
                case 1:
                    return launch_wave<D, 1>(omega, W, segtab, ops, G, A, geo, Ypart, stream);
                case 2:
                    return launch_wave<D, 2>(omega, W, segtab, ops, G, A, geo, Ypart, stream);
                default:
                    return launch_wave<D, 3>(omega, W, segtab, ops, G, A, geo, Ypart, stream);
            }
        }
    }
    switch (geo.nwaves) {
#define FFK_NCASE(N) \
    case N:          \
        return launch_dw<D, N>(omega, W, segtab, ops, G, A, geo, Ypart, stream);
        FFK_NCASE(1) FFK_NCASE(2) FFK_NCASE(3) FFK_NCASE(4) FFK_NCASE(5) FFK_NCASE(6) FFK_NCASE(7)
        FFK_NCASE(8)
#undef FFK_NCASE
        default:
            return hipErrorInvalidValue;
    }
}

// resident blocks per CU of the kernel instantiation a geometry selects (occupancy API, cached)
template <int D, int NW>
int blocks_per_cu_dw(int nbuf, int lds_bytes) {
    constexpr int JB = accum_jb(D);
    constexpr int MR = accum_mr(D);
    int n = 0;
    hipError_t err;
    if constexpr (MR == D) {
        if (nbuf == 2) {
            auto kern = ctrl_accumulate_kernel<D, JB, MR, 2, NW, 1>;
            if (lds_bytes > 48*1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, NW*64, lds_bytes);
            return err == hipSuccess ? n : 0;
        }
    }
    auto kern = ctrl_accumulate_kernel<D, JB, MR, 1, NW, 1>;
    if (lds_bytes > 48*1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, NW*64, lds_bytes);
    return err == hipSuccess ? n : 0;
}

template <int D>
int blocks_per_cu_d(int nwaves, int nbuf, int lds_bytes) {
    switch (nwaves) {
#define FFK_NCASE(N) \
    case N:          \
        return blocks_per_cu_dw<D, N>(nbuf, lds_bytes);
        FFK_NCASE(1) FFK_NCASE(2) FFK_NCASE(3) FFK_NCASE(4) FFK_NCASE(5) FFK_NCASE(6) FFK_NCASE(7)
        FFK_NCASE(8)
#undef FFK_NCASE
        default:
            return 0;
    }
}

int query_blocks_per_cu(int d, int nwaves, int nbuf, int lds_bytes) {
    static int cache[kMaxD + 1][9][3] = {};
    int& slot = cache[d][nwaves][nbuf];
    if (slot > 0) return slot;
    int n = 0;
    switch (d) {
#define FFK_CASE(D)                                      \
    case D:                                              \
        n = blocks_per_cu_d<D>(nwaves, nbuf, lds_bytes); \
        break;
        FFK_ALL_D(FFK_CASE)
#undef FFK_CASE
        default:
            break;
    }
    if (n < 1) n = 1;
    slot = n;
    return n;
}

}  // namespace

int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

namespace {

}  // namespace

void set_use_wave_kernel(bool on) { g_use_wave_kernel = on; }
void set_use_gsplit(bool on) { g_use_gsplit = on; }
void set_mfma_policy(int policy) { g_mfma_policy = policy; }

AccumGeometry accumulate_geometry(int W, int A, int G, int d, int forced_chunks) {
    AccumGeometry geo;
    geo.gsplit = 1;
    geo.pc = false;
    geo.pcw = false;
    geo.generic = false;
    if (generic_dimension(d)) {
        // generic.hip: one 256-thread block per (frequency, operator, segment chunk); chunks only
        // where W A alone does not fill the chip (every chunk costs a partial sum of A d^2 W)
        geo.generic = true;
        geo.mfma = false;
        geo.wave_kernel = false;
        geo.nwaves = 4;
        geo.task_groups = A;
        geo.na_blk = 1;
        geo.nbuf = 1;
        geo.lds_bytes = static_cast<int>(2*sizeof(cplx)*d*d + 64*sizeof(double));
        int chunks = forced_chunks;
        if (chunks <= 0) {
            const long blocks = static_cast<long>(W)*A;
            chunks = static_cast<int>(std::max<long>(1, (2L*device_cu_count() + blocks - 1)/blocks));
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    geo.mfma = mfma_accumulate_supported(d) && g_mfma_policy != 1 && (d >= 12 || g_mfma_policy == 2);
    // d = 4: the producer/consumer kernel with the second product on the matrix cores (ctrl_pq.hip); the tuning
    // variants 1/2 (ffk_set_accumulate_variant) select the symmetric kernel below
    if (g_use_gsplit && !g_use_wave_kernel && pq_accumulate_supported(d, A) && !geo.mfma) {
        // groups of three operators, the remainder as groups of two (or a single operator): one launch per group
        // size, each with the same segment chunks (ctrl_pq.hip: pq_accumulate_groups)
        const PqGroups grp = pq_accumulate_groups(A);
        const int nc = grp.n3 > 0 ? 3 : (grp.n2 > 0 ? 2 : 1);
        geo.pc = true;
        geo.wave_kernel = false;
        geo.nwaves = pq_accumulate_waves(nc);
        geo.task_groups = grp.n3 + grp.n2 + grp.n1;
        geo.na_blk = nc;
        geo.nbuf = 2;
        geo.lds_bytes = pq_accumulate_lds_bytes(nc);
        const long wt = (W + 63)/64;
        const long launch_tiles[3] = {wt*grp.n3, wt*grp.n2, wt*grp.n1};
        int chunks = forced_chunks;
        if (chunks <= 0) {
            // one block per CU and round, per launch; a block's fixed cost (first tile, last hand-over) is about
            // three tiles
            const long capacity = device_cu_count();
            const int max_chunks = std::max(1, std::min((G + 15)/16, 256));
            double best = 0.0;
            chunks = 1;
            for (int c = 1; c <= max_chunks; ++c) {
                double cost = 0.0;
                for (long tiles : launch_tiles)
                    if (tiles > 0) cost += static_cast<double>((tiles*c + capacity - 1)/capacity)*((G + c - 1)/c + 3);
                if (c == 1 || cost < best*0.999) {
                    best = cost;
                    chunks = c;
                }
            }
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    // d = 2: the folded-operand kernel (ctrl_d2.hip): blocks of independent wavefronts that split the block's segment
    // chunk and add up through LDS; enough chunks for one block per CU, at least two segments per wavefront.  The tuning variants 1/2 select the kernels below.
    if (g_use_gsplit && !g_use_wave_kernel && d2_accumulate_supported(d)) {
        geo.d2 = true;
        geo.mfma = false;
        geo.wave_kernel = false;
        geo.nwaves = d2_accumulate_waves();
        geo.na_blk = d2_accumulate_ops_per_block(A);
        geo.task_groups = (A + geo.na_blk - 1)/geo.na_blk;
        geo.nbuf = 2;
        geo.lds_bytes = d2_accumulate_lds_bytes(geo.na_blk);
        int chunks = forced_chunks;
        if (chunks <= 0) {
            const int fpb = d2_accumulate_freqs_per_block();
            const long tiles = static_cast<long>((W + fpb - 1)/fpb)*geo.task_groups;
            const long want_waves = 8L*device_cu_count();      // one block of eight wavefronts per CU
            chunks = static_cast<int>(std::max<long>(1, (want_waves + tiles*geo.nwaves - 1)/(tiles*geo.nwaves)));
            chunks = std::min(chunks, std::max(1, G/(2*geo.nwaves)));
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    // d = 8: producer/consumer kernel on the matrix cores with a real integral tile (ctrl_pcr.hip);
    // the tuning variants 1/2 (ffk_set_accumulate_variant) select the symmetric kernel below
    if (g_use_gsplit && !g_use_wave_kernel && pcr_accumulate_supported(d, A) && !geo.mfma) {
        const int nc = pcr_accumulate_ops_per_block();
        geo.pcw = true;
        geo.wave_kernel = false;
        geo.nwaves = pcr_accumulate_waves();
        geo.task_groups = (A + nc - 1)/nc;
        geo.na_blk = nc;
        geo.nbuf = 2;
        geo.lds_bytes = pcr_accumulate_lds_bytes();
        const long tiles = static_cast<long>((W + 63)/64)*geo.task_groups;
        int chunks = forced_chunks;
        if (chunks <= 0) {
            const long capacity = device_cu_count();          // one 16-wave block per CU
            const int max_chunks = std::max(1, std::min((G + 7)/8, 256));
            double best = 0.0;
            chunks = 1;
            for (int c = 1; c <= max_chunks; ++c) {
                const long rounds = (tiles*c + capacity - 1)/capacity;
                const double cost = static_cast<double>(rounds)*((G + c - 1)/c + 2);
                if (c == 1 || cost < best*0.999) {
                    best = cost;
                    chunks = c;
                }
            }
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    if (geo.mfma) {
        // one wavefront per noise operator, 16 frequencies per block, one block per CU
        geo.wave_kernel = false;
        geo.nwaves = mfma_accumulate_waves(d, A);
        const int ops_per_block = mfma_accumulate_ops_per_block(d, A);
        geo.task_groups = (A + ops_per_block - 1)/ops_per_block;
        geo.na_blk = ops_per_block;
        geo.nbuf = 1;
        geo.lds_bytes = mfma_accumulate_lds_bytes(d, geo.nwaves);
        const long tiles = static_cast<long>((W + 15)/16)*geo.task_groups;
        int chunks = forced_chunks;
        if (chunks <= 0) {
            // resident blocks: one per CU for the 512-register kernel; for d = 8 two waves per SIMD
            const long per_cu = std::max(1, (geo.nwaves > 8 ? 12 : 8)/geo.nwaves);
            const long capacity = device_cu_count()*per_cu;
            const int max_chunks = std::max(1, std::min((G + 3)/4, 256));
            double best = 0.0;
            chunks = 1;
            for (int c = 1; c <= max_chunks; ++c) {
                const long rounds = (tiles*c + capacity - 1)/capacity;
                const double cost = static_cast<double>(rounds)*((G + c - 1)/c + 1);
                if (c == 1 || cost < best*0.999) {
                    best = cost;
                    chunks = c;
                }
            }
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    // d = 2, long sequences: the one-wave kernel (no barrier, nothing shared between waves) with
    // enough segment chunks for eight waves per SIMD -- every wave-segment is a dependent chain
    // (two sincos, two integrals, 80 FMAs per operator) that only other waves can hide.  The
    // reference's periodic_driving example written out (200 002 segments, 500 omega, 2 operators):
    // 1.13 ms against 1.97 ms for the block kernel at its 256 sub-chunks (profiles/r02_k_*); at
    // 256 segments the two are level (37 / 40 us), so short sequences stay on the block kernel.
    const bool long_d2 = d == 2 && G >= kLongSequenceD2 && g_use_gsplit && forced_chunks <= 0;
    geo.wave_kernel = d <= kWaveKernelMaxD && (g_use_wave_kernel || long_d2);
    if (geo.wave_kernel) {
        // one wave per block, up to 3 noise operators per lane; blocks per CU = 4 (1 wave/SIMD)
        geo.na_blk = A <= 3 ? A : (A % 3 == 0 ? 3 : (A % 2 == 0 ? 2 : 3));
        geo.nwaves = 1;
        geo.task_groups = (A + geo.na_blk - 1)/geo.na_blk;
        geo.nbuf = 2;
        geo.lds_bytes = static_cast<int>(2*(1 + geo.na_blk)*d*d*sizeof(cplx));
        const long tiles = static_cast<long>((W + 63)/64)*geo.task_groups;
        int chunks = forced_chunks;
        if (chunks <= 0) {
            // 256 CUs x 4 SIMDs: one fat wave each (d = 4: up to 512 VGPRs), eight thin ones at d = 2
            const long slots = long_d2 ? 8192 : 1024;
            chunks = static_cast<int>(std::max<long>(1, slots / tiles));
            const int max_chunks = (G + 3)/4;
            if (chunks > max_chunks) chunks = std::max(1, max_chunks);
        }
        chunks = std::max(1, std::min(chunks, G));
        geo.chunk_len = (G + chunks - 1)/chunks;
        geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
        return geo;
    }
    const int jb = accum_jb(d);
    const int ntasks = A*(d / jb);
    // waves per block: all tasks if they fit (<= 8 waves); otherwise the count in 5..8 that
    // minimises (task groups) x (per-wave share of the integral generation + one task's
    // contraction), in VALU instructions per segment: every task group regenerates the integral,
    // and wider blocks also raise the occupancy (measured at d = 8, A = 9: 8 waves x 5 groups
    // 11.3 ms, 6 x 6 12.8 ms, 4 x 9 17.4 ms).
    int nw = ntasks <= 8 ? ntasks : 8;
    if (ntasks > 8) {
        const double gen = 35.0*(d*(d - 1) + 1);
        const double contr = 4.0*d*d*(2*jb + 1);
        double best = 0.0;
        for (int c = 8; c >= 5; --c) {
            const double cost = ((ntasks + c - 1)/c)*(gen/c + contr);
            if (c == 8 || cost < best) {
                best = cost;
                nw = c;
            }
        }
    }
    geo.nwaves = nw;
    geo.task_groups = (ntasks + nw - 1)/nw;
    const int nj = d / jb;
    geo.na_blk = ntasks <= nw ? A : std::min(A, (nw - 1)/nj + 2);
    const size_t one = (static_cast<size_t>(accum_mr(d))*d*64 + static_cast<size_t>(1 + geo.na_blk)*d*d)*sizeof(cplx);
    // table rows + (double-buffered kernels) the shared trigonometry, see the kernel's LDS layout
    const size_t row = static_cast<size_t>(seg_stride(d))*sizeof(double);
    const size_t tabs2 = 3*row + 2*2*64*sizeof(cplx);
    geo.nbuf = (accum_mr(d) == d && 2*one + tabs2 <= 160*1024) ? 2 : 1;
    geo.lds_bytes = static_cast<int>(geo.nbuf == 2 ? 2*one + tabs2 : one + 2*row);
    geo.gsplit = 1;
    // Segment chunks.  Every block runs its whole chunk, so the launch is fastest when the grid is
    // a whole number of "rounds" of resident blocks (profiles/r01_a_chunk_sweep.txt: 16 chunks =
    // 1024 blocks = exactly one round beat 22 chunks by 25 %).  Model: rounds x (segments per
    // chunk + ~2 segments' worth of prologue/epilogue), minimised over the chunk count with every
    // chunk keeping >= 4 segments.
    const long tiles = static_cast<long>((W + 63)/64)*geo.task_groups;
    int chunks = forced_chunks;
    if (chunks <= 0) {
        const long capacity = static_cast<long>(device_cu_count())*
                              query_blocks_per_cu(d, nw, geo.nbuf, geo.lds_bytes);
        const int max_chunks = std::max(1, std::min((G + 3)/4, 256));  // keep >= 4 segments per chunk
        double best = 0.0;
        chunks = 1;
        for (int c = 1; c <= max_chunks; ++c) {
            const long rounds = (tiles*c + capacity - 1)/capacity;
            const double cost = static_cast<double>(rounds)*((G + c - 1)/c + 2);
            if (c == 1 || cost < best*0.999) {
                best = cost;
                chunks = c;
            }
        }
        // fold a factor kGsplit of the split into the block (same number of waves in flight,
        // kGsplit x fewer partial sums) when the block stays within 16 waves and the LDS
        // (round 6, measured with the first bench entries of d = 2, 3 -- profiles/r06_g_*: at d = 3 with 64 or more
        // frequency tiles the unfolded grid is faster, 256 segments x 3 operators x 4096 omega 92.7 -> 68.0 us, the
        // pass 132 -> 119 us; at 16 tiles the extra partial sums cost the pass 11 %; at d = 2 the kernel gains 9 % and
        // the pass loses 4 %: folded as before)
        const bool unfolded_is_faster = d == 3 && tiles >= 64;
        if (g_use_gsplit && d <= kGsplitMaxD && nw <= kGsplitMaxNW && geo.nbuf == 2 && !unfolded_is_faster &&
            chunks >= kGsplit && chunks % kGsplit == 0 &&
            static_cast<size_t>(kGsplit)*geo.lds_bytes <= 160*1024) {
            geo.gsplit = kGsplit;
            geo.lds_bytes *= kGsplit;
            chunks /= kGsplit;
        }
    }
    if (chunks > G) chunks = G;
    if (chunks < 1) chunks = 1;
    geo.chunk_len = (G + chunks - 1)/chunks;
    geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
    return geo;
}

namespace {
// ---- padded dimensions (ffk_internal.h: padded_dimension) -------------------------------------------------------
// One thread per complex number of the padded operands and table rows of a segment: T (+) 1, Bbar_a (+) 0; a table
// record (dE, sin b, cos b) of an entry that involves an added level is that of a degenerate pair (0, 0, 1) -- Bbar is
// zero there, the entry multiplies nothing.
__global__ __launch_bounds__(256) void pad_operands_kernel(const double* __restrict__ segtab, const cplx* __restrict__ ops,
                                                           int d, int p, int A, cplx* __restrict__ ops_p,
                                                           double* __restrict__ segtab_p) {
    const int g = blockIdx.x;
    const int n_ops = (1 + A)*p*p, S = seg_stride(d), Sp = seg_stride(p);
    const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*d*d;
    cplx* dst = ops_p + static_cast<size_t>(g)*n_ops;
    for (int e = threadIdx.x; e < n_ops; e += 256) {
        const int o = e/(p*p), i = (e/p) % p, j = e % p;
        cplx v = {0.0, 0.0};
        if (i < d && j < d) v = src[(o*d + i)*d + j];
        else if (o == 0 && i == j) v = {1.0, 0.0};
        dst[e] = v;
    }
    const double* row = segtab + static_cast<size_t>(g)*S;
    double* row_p = segtab_p + static_cast<size_t>(g)*Sp;
    for (int e = threadIdx.x; e < Sp; e += 256) {
        double v = 0.0;
        if (e < 4) {
            v = row[e];
        } else if (e < 4 + 4*p*p) {
            const int rec = (e - 4)/4, k = (e - 4) % 4, m = rec/p, n = rec % p;
            if (m < d && n < d) v = row[seg_rec(m*d + n) + k];
            else v = k == 2 ? 1.0 : 0.0;
        }
        row_p[e] = v;
    }
}
// Y (chunks, A, d, d, W) <- the d x d blocks of Y' (chunks', A, p, p, W); the planes of chunks the padded launch did
// not use (it chooses its own, at most the caller's) are zeros
__global__ __launch_bounds__(256) void unpad_partial_sums_kernel(const cplx* __restrict__ Yp, int d, int p, int W,
                                                                 int planes_used, cplx* __restrict__ Y) {
    const int w = blockIdx.x*256 + threadIdx.x;
    if (w >= W) return;
    const int e = blockIdx.y, ca = blockIdx.z, i = e/d, j = e % d;
    cplx v = {0.0, 0.0};
    if (ca < planes_used) v = Yp[(static_cast<size_t>(ca)*p*p + i*p + j)*W + w];
    Y[(static_cast<size_t>(ca)*d*d + e)*W + w] = v;
}
}  // namespace

hipError_t launch_accumulate(const double* omega, int W, const double* segtab, const cplx* ops,
                             int G, int d, int A, const AccumGeometry& geo, cplx* Ypart,
                             hipStream_t stream, const ExpandEpilogue* expand, bool* expanded,
                             const cplx* wfold) {
    if (expanded) *expanded = false;
    // A dimension between the specialised kernels, with scratch for it (the callers that size their workspace by
    // wfold_elems): padded to the next one.  The chunks are those of this dimension's own geometry -- the partial
    // sums the caller reads keep their shape.
    const int p = padded_dimension(d);
    if (p != 0 && padded_launch_pays(d, G, W, A) && wfold != nullptr && !geo.generic && g_use_gsplit && !g_use_wave_kernel &&
        static_cast<long>(geo.chunks)*A <= 65535 && std::getenv("FFK_NO_PADDED_DIMENSIONS") == nullptr) {
        // (the padded kernel runs the caller's segment chunks: its own choice, where smaller, measured level or worse)
        const int chunk_len = geo.chunk_len, used = geo.chunks;
        cplx* scratch = const_cast<cplx*>(wfold);
        cplx* ops_p = scratch;
        double* segtab_p = reinterpret_cast<double*>(ops_p + static_cast<size_t>(G)*(1 + A)*p*p);
        cplx* Yp = reinterpret_cast<cplx*>(segtab_p + static_cast<size_t>(G)*seg_stride(p));
        cplx* fold8 = Yp + static_cast<size_t>(geo.chunks)*A*p*p*W;
        hipLaunchKernelGGL(pad_operands_kernel, dim3(G), dim3(256), 0, stream, segtab, ops, d, p, A, ops_p, segtab_p);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
        if (p == 8) {
            err = launch_fold_w8(segtab_p, ops_p, G, A, fold8, stream);
            if (err == hipSuccess)
                err = launch_accumulate_pcr(omega, W, segtab_p, ops_p, G, p, A, used, chunk_len, Yp, fold8, stream);
        } else {
            err = launch_accumulate_mfma(omega, W, segtab_p, ops_p, G, p, A, used, chunk_len,
                                         mfma_accumulate_waves(p, A), Yp, stream, nullptr, nullptr);
        }
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(unpad_partial_sums_kernel, dim3((W + 255)/256, d*d, geo.chunks*A), dim3(256), 0, stream, Yp,
                           d, p, W, used*A, Ypart);
        return hipGetLastError();
    }
    if (geo.generic)
        return launch_accumulate_generic(omega, W, segtab, ops, G, d, A, geo.chunks, geo.chunk_len, Ypart,
                                         stream);
    if (geo.d2) return launch_accumulate_d2(omega, W, segtab, ops, G, A, geo.chunks, geo.chunk_len, Ypart, stream);
    if (geo.pc)
        return launch_accumulate_pq(omega, W, segtab, ops, G, d, A, geo.chunks, geo.chunk_len, Ypart, wfold,
                                    stream);
    if (geo.pcw)
        return launch_accumulate_pcr(omega, W, segtab, ops, G, d, A, geo.chunks, geo.chunk_len, Ypart, wfold,
                                     stream);
    if (geo.mfma)
        return launch_accumulate_mfma(omega, W, segtab, ops, G, d, A, geo.chunks, geo.chunk_len,
                                      geo.nwaves, Ypart, stream, geo.chunks == 1 ? expand : nullptr, expanded);
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_d<D>(omega, W, segtab, ops, G, A, geo, Ypart, stream);
        FFK_ALL_D(FFK_CASE)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace ffk
