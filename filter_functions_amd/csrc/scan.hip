// scan.hip -- K2: cumulative propagators Q[g+1] = P[g] Q[g] (util.adot, util.py:868-877, as used
// by numeric.diagonalize numeric.py:1933) as a three-phase chunked scan with the d x d complex
// matrix product as the (non-commutative) operator:
//   1. every wavefront multiplies through one chunk of L consecutive segments, writing the
//      chunk-local prefixes straight into Q and the chunk total into the workspace;
//   2. one wavefront turns the chunk totals into exclusive prefixes E_c = T_{c-1} ... T_0;
//   3. Q[g+1] <- Q_local[g+1] E_c for every segment of chunk c >= 1, fully parallel.
// The reference multiplies strictly left to right; re-association changes the rounding by
// O(G eps) on unitary factors (DESIGN.md "Numerics").
#include "ffk_internal.h"

namespace ffk {
namespace {

__host__ __device__ inline int chunk_length(int G) {
    int L = 8;
    while (L*L < G) L *= 2;   // ~sqrt(G), power of two, >= 8
    return L;
}

// out = a * b for D x D matrices held in LDS; every lane computes entries lane, lane+64, ...
template <int D>
__device__ __forceinline__ void matmul_lds(const cplx (*a)[D], const cplx (*b)[D], cplx (*out)[D],
                                           int lane) {
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, j = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) cmac(acc, a[i][k], b[k][j]);
        out[i][j] = acc;
    }
}

// Chunk-local prefix products.  All segment propagators of the chunk are fetched into LDS by one
// batch of independent loads first: with one global load per step, the serial chain paid an
// exposed memory round trip (~0.5 us) per segment.
constexpr int kScanBatch = 16;   // propagators staged per batch

template <int D>
__global__ __launch_bounds__(64) void scan_local_kernel(const cplx* __restrict__ P, int G, int L,
                                                        cplx* __restrict__ Q,
                                                        cplx* __restrict__ totals) {
    __shared__ cplx cur[2][D][D];
    __shared__ cplx pg[kScanBatch][D][D];
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int lane = threadIdx.x;
    const int c = blockIdx.x;
    const int g0 = c*L, g1 = min(G, g0 + L);
    for (int e = lane; e < D*D; e += 64) cur[0][e / D][e % D] = {(e / D == e % D) ? 1.0 : 0.0, 0.0};
    if (c == 0)
        for (int e = lane; e < D*D; e += 64) Q[e] = {(e / D == e % D) ? 1.0 : 0.0, 0.0};
    if constexpr (D <= 4) {
        // Four matrices fit one pass of the 64 lanes: the 16 segments of the chunk as four groups of
        // four -- group-local prefixes side by side (3 dependent products), the exclusive prefixes
        // of the four group totals (2), one independent fix-up per segment (1): 6 dependent
        // products instead of 16.  Segments beyond the end count as identity.
        constexpr int DD = D*D, NG = 4, GL = kScanBatch/NG;
        if (L == kScanBatch) {
            // the group-local prefixes overwrite the staged propagators in place (same layout:
            // segment k*GL + sg of the chunk): with a second 4 KiB array the kernel held 9.5 KiB of
            // static LDS, and a kernel with more than 8 KiB is not placed beside an accumulate block
            // of another pass (tools/corun.hip)
            cplx (*lp)[GL][D][D] = reinterpret_cast<cplx (*)[GL][D][D]>(&pg[0][0][0]);
            __shared__ cplx xg[NG][D][D];
            for (int e = lane; e < kScanBatch*DD; e += 64) {
                const int sg = e / DD, ent = e % DD;
                cplx v = {(ent / D == ent % D) ? 1.0 : 0.0, 0.0};
                if (g0 + sg < g1) v = P[static_cast<size_t>(g0 + sg)*DD + ent];
                (&pg[0][0][0])[e] = v;
            }
            __syncthreads();
            const int k = (lane / DD) % NG, e = lane % DD, i = e / D, j = e % D;
            const bool act = lane < NG*DD;
            for (int sg = 1; sg < GL; ++sg) {      // (lp[k][0] is pg[k*GL] already)
                cplx acc = {0.0, 0.0};
                if (act) {
#pragma unroll
                    for (int x = 0; x < D; ++x) cmac(acc, pg[k*GL + sg][i][x], lp[k][sg - 1][x][j]);
                }
                __syncthreads();                   // every read of segment (k, sg) before its overwrite
                if (act) lp[k][sg][i][j] = acc;
                __syncthreads();
            }
            if (lane < DD) {
                xg[0][i][j] = {(i == j) ? 1.0 : 0.0, 0.0};
                xg[1][i][j] = lp[0][GL - 1][i][j];
            }
            __syncthreads();
            for (int k2 = 2; k2 < NG; ++k2) {
                if (lane < DD) {
                    cplx acc = {0.0, 0.0};
#pragma unroll
                    for (int x = 0; x < D; ++x) cmac(acc, lp[k2 - 1][GL - 1][i][x], xg[k2 - 1][x][j]);
                    xg[k2][i][j] = acc;
                }
                __syncthreads();
            }
            for (int sg = 0; sg < GL; ++sg) {
                const int g = g0 + k*GL + sg;
                if (act && g < g1) {
                    cplx acc = {0.0, 0.0};
#pragma unroll
                    for (int x = 0; x < D; ++x) cmac(acc, lp[k][sg][i][x], xg[k][x][j]);
                    Q[static_cast<size_t>(g + 1)*DD + e] = acc;
                    if (g == g1 - 1) totals[static_cast<size_t>(c)*DD + e] = acc;
                }
            }
            return;
        }
    }
    int b = 0;
    for (int gb = g0; gb < g1; gb += kScanBatch) {
        const int nb = min(kScanBatch, g1 - gb);
        __syncthreads();
        for (int e = lane; e < nb*D*D; e += 64)
            (&pg[0][0][0])[e] = P[static_cast<size_t>(gb)*D*D + e];
        __syncthreads();
        for (int s2 = 0; s2 < nb; ++s2) {
            matmul_lds<D>(pg[s2], cur[b], cur[b ^ 1], lane);
            __syncthreads();
            b ^= 1;
            for (int e = lane; e < D*D; e += 64)
                Q[static_cast<size_t>(gb + s2 + 1)*D*D + e] = cur[b][e / D][e % D];
        }
    }
    for (int e = lane; e < D*D; e += 64)
        totals[static_cast<size_t>(c)*D*D + e] = cur[b][e / D][e % D];
}

// exclusive prefixes of the chunk totals, in place: totals[c] <- T_{c-1} ... T_0 (identity for c=0)
template <int D>
__global__ __launch_bounds__(64) void scan_totals_kernel(cplx* __restrict__ totals, int nchunks) {
    __shared__ cplx cur[2][D][D];
    __shared__ cplx tc[D][D];
    const int lane = threadIdx.x;
    for (int e = lane; e < D*D; e += 64) cur[0][e / D][e % D] = {(e / D == e % D) ? 1.0 : 0.0, 0.0};
    int b = 0;
    for (int c = 0; c < nchunks; ++c) {
        for (int e = lane; e < D*D; e += 64) {
            tc[e / D][e % D] = totals[static_cast<size_t>(c)*D*D + e];
        }
        __syncthreads();
        for (int e = lane; e < D*D; e += 64)
            totals[static_cast<size_t>(c)*D*D + e] = cur[b][e / D][e % D];
        matmul_lds<D>(tc, cur[b], cur[b ^ 1], lane);
        __syncthreads();
        b ^= 1;
    }
}

template <int D>
__global__ __launch_bounds__(64) void scan_apply_kernel(cplx* __restrict__ Q,
                                                        const cplx* __restrict__ excl, int G,
                                                        int L) {
    __shared__ cplx ql[D][D];
    __shared__ cplx ex[D][D];
    __shared__ cplx out[D][D];
    const int lane = threadIdx.x;
    const int g = L + blockIdx.x;  // chunk 0 needs no fix-up
    if (g >= G) return;
    const int c = g / L;
    for (int e = lane; e < D*D; e += 64) {
        ql[e / D][e % D] = Q[static_cast<size_t>(g + 1)*D*D + e];
        ex[e / D][e % D] = excl[static_cast<size_t>(c)*D*D + e];
    }
    __syncthreads();
    matmul_lds<D>(ql, ex, out, lane);
    __syncthreads();
    for (int e = lane; e < D*D; e += 64) Q[static_cast<size_t>(g + 1)*D*D + e] = out[e / D][e % D];
}

template <int D>
hipError_t launch_d(const cplx* P, int G, cplx* Q, void* ws, hipStream_t stream) {
    const int L = chunk_length(G);
    const int nchunks = (G + L - 1)/L;
    cplx* totals = static_cast<cplx*>(ws);
    hipLaunchKernelGGL(scan_local_kernel<D>, dim3(nchunks), dim3(64), 0, stream, P, G, L, Q, totals);
    if (nchunks > 1) {
        hipLaunchKernelGGL(scan_totals_kernel<D>, dim3(1), dim3(64), 0, stream, totals, nchunks);
        hipLaunchKernelGGL(scan_apply_kernel<D>, dim3(G - L), dim3(64), 0, stream, Q, totals, G, L);
    }
    return hipGetLastError();
}

}  // namespace

// phase 1 only, with a caller-chosen chunk length (the fused fix-up + prologue kernel of
// prep.hip does phases 2 and 3 itself)
hipError_t launch_scan_local(const cplx* seg_prop, int G, int d, int L, cplx* Qloc, cplx* totals,
                             hipStream_t stream) {
    const int nchunks = (G + L - 1)/L;
    switch (d) {
#define FFK_CASE(D)                                                                              \
    case D:                                                                                      \
        hipLaunchKernelGGL(scan_local_kernel<D>, dim3(nchunks), dim3(64), 0, stream, seg_prop, G, \
                           L, Qloc, totals);                                                     \
        break;
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

size_t scan_workspace_bytes(int G, int d) {
    const int L = chunk_length(G);
    const int nchunks = (G + L - 1)/L;
    return align_up(static_cast<size_t>(nchunks)*d*d*sizeof(cplx));
}

hipError_t launch_prefix_products(const cplx* seg_prop, int G, int d, cplx* Q, void* ws,
                                  hipStream_t stream) {
    if (generic_dimension(d)) return launch_prefix_products_generic(seg_prop, G, d, Q, stream);
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_d<D>(seg_prop, G, Q, ws, stream);
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace ffk
