// ctrl_d2.hip -- K3s: the control-matrix accumulation for d = 2 (one qubit; round 6).
//     Y_a(w) = sum_g T_g^dag [ Bbar_a o E_g(w) ] T_g                    (reference numeric.py:846-869 / :596-609)
// At d = 2 the integral tile has THREE distinct entries -- the diagonal (both entries coincide), (0, 1), (1, 0) -- and
// everything around them is frequency independent:
//     Y_a(w) = sum_g  E_d M^d_{a,g} + E_01 M^01_{a,g} + E_10 M^10_{a,g},
//     M^mn[i][j] = Bbar_a[m][n] conj(T[m][i]) T[n][j],   M^d = M^00 + M^11:
// twelve complex multiply-adds per operator and (segment, frequency) where the two 2 x 2 products of the general
// kernels take twenty-two and a Hadamard product.  The symmetric kernel of ctrl.hip, built for a tile that several
// wavefronts generate together, spends a d = 2 segment on two barriers and a hand-over through LDS (1.2 us per segment
// and block, 0.21 of the FP64 peak on the flops it executes); the one-wave kernel has no barrier but a segment is a
// serial chain of ~500 FP64 instructions per wavefront (profiles/r06_g_*).
//
// Here a block is eight wavefronts that share NOTHING while they work: lane = frequency, wavefront w = the w-th
// eighth of the block's segment chunk.  A wavefront keeps a private, double-buffered LDS record of its next segment:
// lane l forms element l of [M^d | M^01 | M^10] x operators from T_g and Bbar_a (two products per lane, a segment
// ahead, beside the current segment's arithmetic), twelve more lanes bring the table row; the record is read back as
// broadcasts.  At the end the eight partial sums are added through LDS in a fixed tree and ONE partial sum per block
// is written (the chunks of a launch are the blocks along grid.z: 8 instead of 32 partial sums of A d^2 W at the
// documentation's shapes).  164 registers: one block per CU, two wavefronts per SIMD; measured alternatives (128
// registers with spills and two blocks per CU; blocks of four wavefronts, three per CU; two frequencies per lane) were
// level or slower -- at ~350 instructions per segment of which half are not multiply-adds the kernel issues one
// instruction per ~6.7 cycles and SIMD whatever the arrangement (profiles/r06_g_*, last section).
#include <algorithm>

#include "ffk_internal.h"
#include "ffk_math.h"

namespace ffk {
namespace {

constexpr int kD2Waves = 8;
using double4_t = __attribute__((ext_vector_type(4))) double;

template <int AT>
__global__ __launch_bounds__(kD2Waves*64) void ctrl_accumulate_d2_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab, const cplx* __restrict__ ops,
    int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    constexpr int D = 2, DD = 4, S = seg_stride(2);
    constexpr int NM = 12*AT;               // complex entries of the folded operands per segment
    constexpr int NREC = NM + S/2;          // + the table row, as complex pairs
    static_assert(NREC <= 64, "a segment's record is one element per lane");
    __shared__ __attribute__((aligned(16))) cplx record[kD2Waves][2][NREC];
    __shared__ __attribute__((aligned(16))) cplx red[kD2Waves/2][AT*DD][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int iw = blockIdx.x*64 + lane;
    const double om = omega[iw < W ? iw : W - 1];
    const int alpha0 = blockIdx.y*AT;
    const int na = min(AT, A - alpha0);
    const int cbeg = blockIdx.z*chunk_len, cend = min(G, cbeg + chunk_len);
    const int sub_len = (cend - cbeg + kD2Waves - 1)/kD2Waves;
    const int g0 = cbeg + wave*sub_len, g1 = min(cend, g0 + sub_len);

    // this lane's element of a segment's record: a term or two of the folded operands, or a piece of the table row
    const int a = lane/12, which = (lane % 12)/4, e = lane % 4, i = e >> 1, j = e & 1;
    const bool is_m = lane < 12*na, is_row = lane >= NM && lane < NREC;
    const int m1 = which == 2 ? 1 : 0, n1 = which == 1 ? 1 : 0;      // M^d: (0,0) then (1,1); M^01: (0,1); M^10: (1,0)
    // Seven loads per lane and segment from fixed per-lane offsets, NO branch around them and plain variables: with the
    // loads under `if (is_m) .. else if (is_row)` (or in a struct returned from a lambda) the compiler kept the values
    // in scratch memory and waited for every load on the spot -- 4 us per segment (profiles/r06_g_*, last section).
    const int ob = is_m ? (1 + alpha0 + a)*DD : 0;
    const int o_b1 = ob + (is_m ? m1*D + n1 : 0), o_ti1 = is_m ? m1*D + i : 0, o_tj1 = is_m ? n1*D + j : 0;
    const int o_b2 = ob + (is_m ? 3 : 0), o_ti2 = is_m ? D + i : 0, o_tj2 = is_m ? D + j : 0;
    const int o_row = is_row ? lane - NM : 0;
    const double w2 = is_m && which == 0 ? 1.0 : 0.0;     // M^d has a second term
    cplx p_b1, p_ti1, p_tj1, p_b2, p_ti2, p_tj2, p_row;
    auto request = [&](int g) __attribute__((always_inline)) {
        const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*DD;
        p_b1 = src[o_b1];
        p_ti1 = src[o_ti1];
        p_tj1 = src[o_tj1];
        p_b2 = src[o_b2];
        p_ti2 = src[o_ti2];
        p_tj2 = src[o_tj2];
        p_row = reinterpret_cast<const cplx*>(segtab + static_cast<size_t>(g)*S)[o_row];
    };
    auto park = [&](int buf) __attribute__((always_inline)) {
        const cplx v1 = cmul(cmul(p_b1, cplx{p_ti1.re, -p_ti1.im}), p_tj1);
        const cplx v2 = cmul(cmul(p_b2, cplx{p_ti2.re, -p_ti2.im}), p_tj2);
        cplx v = {fma(w2, v2.re, v1.re), fma(w2, v2.im, v1.im)};
        if (!is_m) v = is_row ? p_row : cplx{0.0, 0.0};          // (operators beyond the launch's last one: zeros)
        if (lane < NREC) record[wave][buf][lane] = v;
    };

    cplx Y[AT][DD];
#pragma unroll
    for (int q = 0; q < AT; ++q)
#pragma unroll
        for (int k = 0; k < DD; ++k) Y[q][k] = {0.0, 0.0};

    if (g0 < g1) {
        request(g0);
        park(0);
    }
    for (int g = g0; g < g1; ++g) {
        const int buf = (g - g0) & 1;
        const bool more = g + 1 < g1;
        if (more) request(g + 1);
        const cplx* rec = record[wave][buf];
        const double* st = reinterpret_cast<const double*>(rec + NM);
        // the tile: e^{i w t_g} I^(g), the diagonal entry and the two off-diagonal ones by angle addition from the
        // diagonal's half-angle (the arithmetic of ctrl.hip's one-wave kernel)
        // (ffk_math.h: E = psi e^{ib} 2 sin(a + b)/x with psi = e^{i w t_g} e^{ia}; the three entries without a
        // branch between them -- the two sincos and the three entries are independent chains, and a segment's time is
        // the latency of its longest chain --, near a resonance (rare) redone under ONE branch)
        const double dtg = st[0];
        cplx ph;
        double sa, ca;
        sincos_pi(om*st[1], &ph.im, &ph.re);
        sincos_pi(0.5*(om*dtg), &sa, &ca);
        const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
        const double4_t r1 = *reinterpret_cast<const double4_t*>(st + seg_rec(1));
        const double4_t r2 = *reinterpret_cast<const double4_t*>(st + seg_rec(2));
        cplx Ed, Eo[2];
        {
            const double x0 = om, x1 = om + r1.x, x2 = om + r2.x;
            const double q0 = pf.sa2*rcp_fast(x0);
            const double q1 = fma(pf.sa2, r1.z, pf.ca2*r1.y)*rcp_fast(x1);
            const double q2 = fma(pf.sa2, r2.z, pf.ca2*r2.y)*rcp_fast(x2);
            Ed = {q0*pf.pr, q0*pf.pi};
            Eo[0] = {q1*fma(pf.pr, r1.z, -(pf.pi*r1.y)), q1*fma(pf.pr, r1.y, pf.pi*r1.z)};
            Eo[1] = {q2*fma(pf.pr, r2.z, -(pf.pi*r2.y)), q2*fma(pf.pr, r2.y, pf.pi*r2.z)};
            if (fmin(fabs(x0), fmin(fabs(x1), fabs(x2))) < pf.thr) {
                Ed = phased_integral_aa(pf, 0.0, 0.0, 1.0);
                Eo[0] = phased_integral_aa(pf, r1.x, r1.y, r1.z);
                Eo[1] = phased_integral_aa(pf, r2.x, r2.y, r2.z);
            }
        }
#pragma unroll
        for (int q = 0; q < AT; ++q) {
            if (q < na) {
#pragma unroll
                for (int k = 0; k < DD; ++k) {
                    cmac(Y[q][k], Ed, rec[q*12 + k]);
                    cmac(Y[q][k], Eo[0], rec[q*12 + 4 + k]);
                    cmac(Y[q][k], Eo[1], rec[q*12 + 8 + k]);
                }
            }
        }
        if (more) park(buf ^ 1);
    }

    // the eight partial sums in a fixed tree: (w, w + 4), (w, w + 2), (w, w + 1)
#pragma unroll
    for (int half = kD2Waves/2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2*half) {
#pragma unroll
            for (int q = 0; q < AT; ++q)
#pragma unroll
                for (int k = 0; k < DD; ++k) red[wave - half][q*DD + k][lane] = Y[q][k];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int q = 0; q < AT; ++q)
#pragma unroll
                for (int k = 0; k < DD; ++k) {
                    const cplx u = red[wave][q*DD + k][lane];
                    Y[q][k].re += u.re;
                    Y[q][k].im += u.im;
                }
        }
        __syncthreads();
    }
    if (wave == 0 && iw < W) {
#pragma unroll
        for (int q = 0; q < AT; ++q) {
            if (q < na) {
                cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha0 + q)*DD)*W + iw;
#pragma unroll
                for (int k = 0; k < DD; ++k) out[static_cast<size_t>(k)*W] = Y[q][k];
            }
        }
    }
}

}  // namespace

bool d2_accumulate_supported(int d) { return d == 2; }
int d2_accumulate_waves() { return kD2Waves; }
int d2_accumulate_freqs_per_block() { return 64; }
// operators per block: at most three (the reduction tree's LDS: 16 KB per operator), the groups of a launch as equal
// as they come (4 -> 2 + 2, 5 -> 3 + 2)
int d2_accumulate_ops_per_block(int A) {
    const int groups = (A + 2)/3;
    return (A + groups - 1)/groups;
}
int d2_accumulate_lds_bytes(int at) {
    return static_cast<int>(sizeof(cplx))*(kD2Waves*2*(12*at + seg_stride(2)/2) + (kD2Waves/2)*at*4*64);
}

hipError_t launch_accumulate_d2(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                                int chunks, int chunk_len, cplx* Ypart, hipStream_t stream) {
    const int at = d2_accumulate_ops_per_block(A);
    const dim3 grid((W + 63)/64, (A + at - 1)/at, chunks);
    switch (at) {
#define FFK_CASE(AT)                                                                                            \
    case AT:                                                                                                    \
        hipLaunchKernelGGL(ctrl_accumulate_d2_kernel<AT>, grid, dim3(kD2Waves*64), 0, stream, omega, W, segtab, \
                           ops, G, A, chunk_len, Ypart);                                                        \
        break;
        FFK_CASE(1) FFK_CASE(2) FFK_CASE(3)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ffk
