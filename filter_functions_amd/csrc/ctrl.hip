// ctrl.hip -- K3: the fused control-matrix accumulation, the kernel that carries ~98 % of
// PulseSequence.get_filter_function (numeric.calculate_control_matrix_from_scratch,
// filter_functions/numeric.py:707-881, hot loop :846-869; Hilbert-space twin :456-618).
//
// What it computes, per frequency w (one lane each) and noise operator a:
//     Y_a(w) = sum_g  T_g^dag [ (s_a Bbar_a^(g)) o ( e^{i w t_g} I^(g)(w) ) ] T_g ,
//     I^(g)_mn(w) = (e^{i (w + D_m - D_n) dt_g} - 1) / (i (w + D_m - D_n))
// i.e. the interaction-picture noise operator B~_a(w) of numeric.py:516-537.  The control matrix
// is its basis expansion R[a,k,w] = tr(Y_a(w) C_k), done once after the segment sum
// (post.hip) instead of inside it: 2 d^3 complex MACs per (g, w, a) instead of the d^4 of the
// reference's 'o,jmn,omn,knm->jko' contraction, with identical results (the reference pins the
// two formulations against each other at 1e-14, tests/test_precision.py:313-353).
//
// Mapping:
//   * grid.x tiles omega in 64s, one frequency per lane; grid.z splits the segment axis into
//     chunks whose partial sums are reduced afterwards in fixed order (deterministic);
//     grid.y x waves enumerate tasks (a, column block jb of Y).
//   * Phase A: the waves of a block share the generated integral: wave w computes entries
//     w, w+nw, ... of e^{i w t_g} I^(g) (one sincos + one reciprocal each, the diagonal only
//     once) and parks them in LDS, [entry][lane] so that reads/writes are 16-byte, conflict free.
//   * Phase B: Z[m,:] = sum_n I'[m,n] Wt[m,n,:],  Y[i,:] += conj(T[m,i]) Z[m,:].  Every
//     omega-independent operand (Wt, conj T, the segment table) is wave-uniform, so it is
//     fetched by scalar loads and enters v_fma_f64 as the SGPR source; VGPRs hold only the
//     accumulators and the current integral row.
//   * Double-buffered LDS: one barrier per segment.
#include "ffk_internal.h"

namespace ffk {
namespace {

// MR = integral rows generated per LDS stage.  MR == D (one stage per segment, diagonal computed
// once, optionally double-buffered) whenever the D*D*64 tile fits the 160 KiB LDS; MR < D splits
// the rows over several single-buffered stages (D > 12).  MAXW = upper bound of waves per block
// (launch bound: lets the register allocator use the VGPR budget the block size really leaves).
template <int D, int JB, int MR, int NBUF, int MAXW>
__global__ __launch_bounds__(MAXW*64) void ctrl_accumulate_kernel(const double* __restrict__ omega, int W,
                                       const double* __restrict__ segtab,
                                       const cplx* __restrict__ Wt, const cplx* __restrict__ Tc,
                                       int G, int A, int chunk_len, cplx* __restrict__ Ypart) {
    static_assert(MR == D || NBUF == 1, "row-blocked stages are single buffered");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* lds = reinterpret_cast<cplx*>(lds_raw);  // [NBUF][MR*D][64]
    constexpr int S = seg_stride(D);
    constexpr int NJ = D / JB;
    constexpr int NSTAGE = (D + MR - 1)/MR;
    constexpr int NE = D*(D - 1) + 1;  // distinct integral entries (all diagonal ones coincide)
    constexpr int MUNROLL = D <= 8 ? D : 1;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int task = blockIdx.y*nwaves + wave;
    const bool active = task < A*NJ;
    const int alpha = active ? task / NJ : 0;
    const int jb = active ? task % NJ : 0;
    const int iw = blockIdx.x*64 + lane;
    const double om = omega[iw < W ? iw : W - 1];
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);

    cplx Y[D][JB];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) Y[i][j] = {0.0, 0.0};

    // Phase A: this wave's share of e^{i w t_g} I^(g)[rows of stage][:] -> LDS
    auto phase_a = [&](int g, int stage, int buf) {
        const double* st = segtab + static_cast<size_t>(g)*S;
        const double dtg = st[0];
        const cplx ph = cexp(om*st[1]);
        cplx* dst = lds + static_cast<size_t>(buf)*MR*D*64 + lane;
        if (MR == D) {
            for (int ce = wave; ce < NE; ce += nwaves) {
                int slot = 0;  // the diagonal lives in slot 0
                if (ce > 0) {
                    const int o = ce - 1;
                    const int m = o/(D - 1), r = o % (D - 1);
                    slot = m*D + r + (r >= m ? 1 : 0);
                }
                const cplx I = first_order_integral(om, st[2 + slot], dtg);
                dst[slot*64] = cmul(ph, I);
            }
        } else {
            const int m0 = stage*MR;
            const int rows = min(MR, D - m0);
            for (int e = wave; e < rows*D; e += nwaves) {
                const cplx I = first_order_integral(om, st[2 + m0*D + e], dtg);
                dst[e*64] = cmul(ph, I);
            }
        }
    };

    // Phase B: Z[m,:] = sum_n I'[m,n] Wt[m,n,:];  Y[i,:] += conj(T[m,i]) Z[m,:]
    auto phase_b = [&](int g, int stage, int buf) {
        const cplx* Wg = Wt + (static_cast<size_t>(g)*A + alpha)*D*D*D + jb*JB;
        const cplx* Tg = Tc + static_cast<size_t>(g)*D*D;
        const cplx* src = lds + static_cast<size_t>(buf)*MR*D*64 + lane;
        const int m0 = stage*MR;
        const int m1 = min(D, m0 + MR);
#pragma unroll MUNROLL
        for (int m = m0; m < m1; ++m) {
            cplx Z[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) Z[j] = {0.0, 0.0};
#pragma unroll
            for (int n = 0; n < D; ++n) {
                const int slot = (MR == D) ? ((m == n) ? 0 : m*D + n) : (m - m0)*D + n;
                const cplx Iv = src[slot*64];
#pragma unroll
                for (int j = 0; j < JB; ++j) cmac(Z[j], Wg[(m*D + n)*D + j], Iv);
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) cmac(Y[i][j], Tg[m*D + i], Z[j]);
        }
    };

    if (NBUF == 2) {
        if (g0 < g1) phase_a(g0, 0, 0);
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            const int buf = (g - g0) & 1;
            if (g + 1 < g1) phase_a(g + 1, 0, buf ^ 1);
            if (active) phase_b(g, 0, buf);
            __syncthreads();
        }
    } else {
        for (int g = g0; g < g1; ++g) {
            for (int stage = 0; stage < NSTAGE; ++stage) {
                phase_a(g, stage, 0);
                __syncthreads();
                if (active) phase_b(g, stage, 0);
                __syncthreads();
            }
        }
    }

    if (active && iw < W) {
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*D*D)*W + iw;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < JB; ++j) out[static_cast<size_t>(i*D + jb*JB + j)*W] = Y[i][j];
    }
}

__host__ __device__ constexpr int accum_mr(int d) { return d <= 12 ? d : 8; }

template <typename K>
hipError_t launch_kernel(K kern, const dim3& grid, const dim3& block, const AccumGeometry& geo,
                         hipStream_t stream, const double* omega, int W, const double* segtab,
                         const cplx* Wt, const cplx* Tc, int G, int A, cplx* Ypart) {
    if (geo.lds_bytes > 48*1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             geo.lds_bytes);
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(kern, grid, block, geo.lds_bytes, stream, omega, W, segtab, Wt, Tc, G, A,
                       geo.chunk_len, Ypart);
    return hipGetLastError();
}

template <int D, int MAXW>
hipError_t launch_dw(const double* omega, int W, const double* segtab, const cplx* Wt,
                     const cplx* Tc, int G, int A, const AccumGeometry& geo, cplx* Ypart,
                     hipStream_t stream) {
    constexpr int JB = accum_jb(D);
    constexpr int MR = accum_mr(D);
    const dim3 grid((W + 63)/64, geo.task_groups, geo.chunks);
    const dim3 block(geo.nwaves*64);
    if constexpr (MR == D && 2*D*D*64*sizeof(cplx) <= 158*1024) {
        if (geo.nbuf == 2)
            return launch_kernel(ctrl_accumulate_kernel<D, JB, MR, 2, MAXW>, grid, block, geo,
                                 stream, omega, W, segtab, Wt, Tc, G, A, Ypart);
    }
    return launch_kernel(ctrl_accumulate_kernel<D, JB, MR, 1, MAXW>, grid, block, geo, stream,
                         omega, W, segtab, Wt, Tc, G, A, Ypart);
}

template <int D>
hipError_t launch_d(const double* omega, int W, const double* segtab, const cplx* Wt,
                    const cplx* Tc, int G, int A, const AccumGeometry& geo, cplx* Ypart,
                    hipStream_t stream) {
    if (geo.nwaves <= 4)
        return launch_dw<D, 4>(omega, W, segtab, Wt, Tc, G, A, geo, Ypart, stream);
    if (geo.nwaves <= 8)
        return launch_dw<D, 8>(omega, W, segtab, Wt, Tc, G, A, geo, Ypart, stream);
    return launch_dw<D, 16>(omega, W, segtab, Wt, Tc, G, A, geo, Ypart, stream);
}

}  // namespace

AccumGeometry accumulate_geometry(int W, int A, int G, int d, int forced_chunks) {
    AccumGeometry geo;
    const int jb = accum_jb(d);
    const int ntasks = A*(d / jb);
    // waves per block: all tasks if they fit (<= 16 waves), otherwise the divisor-friendly
    // largest count <= 16 so that the generated integral is shared as widely as possible.
    int nw = ntasks <= 16 ? ntasks : 16;
    if (ntasks > 16) {
        for (int c = 16; c >= 8; --c)
            if (ntasks % c == 0) {
                nw = c;
                break;
            }
    }
    geo.nwaves = nw;
    geo.task_groups = (ntasks + nw - 1)/nw;
    const size_t one = static_cast<size_t>(accum_mr(d))*d*64*sizeof(cplx);
    geo.nbuf = (accum_mr(d) == d && 2*one <= 158*1024) ? 2 : 1;
    geo.lds_bytes = static_cast<int>(geo.nbuf*one);
    // segment chunks: aim for >= ~4 waves per SIMD over the chip (256 CUs x 4 SIMDs)
    const long tiles = static_cast<long>((W + 63)/64)*geo.task_groups*nw;
    int chunks = forced_chunks;
    if (chunks <= 0) {
        const long target = 4096;
        chunks = static_cast<int>((target + tiles - 1)/tiles);
        if (chunks < 1) chunks = 1;
        const int max_chunks = (G + 3)/4;  // keep >= 4 segments per chunk
        if (chunks > max_chunks) chunks = max_chunks < 1 ? 1 : max_chunks;
    }
    if (chunks > G) chunks = G;
    if (chunks < 1) chunks = 1;
    geo.chunk_len = (G + chunks - 1)/chunks;
    geo.chunks = (G + geo.chunk_len - 1)/geo.chunk_len;
    return geo;
}

hipError_t launch_accumulate(const double* omega, int W, const double* segtab, const cplx* Wt,
                             const cplx* Tc, int G, int d, int A, const AccumGeometry& geo,
                             cplx* Ypart, hipStream_t stream) {
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_d<D>(omega, W, segtab, Wt, Tc, G, A, geo, Ypart, stream);
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace ffk
