// liouville.hip -- K5: superoperator.liouville_representation (superoperator.py:51-84 followed
// by Basis.expand basis.py:650-698):  L[b,i,j] = tr(U_b^dag C_i U_b C_j).
//
// Two steps.  (1) conjugate the basis, CB[b,i] = U_b^dag C_i U_b, sixteen elements per block, and
// lay it out as a REAL operand matrix.  (2) the contraction over the d^2 matrix entries is the
// one genuine dense GEMM of the path (N x 2d^2 by 2d^2 x N per batch element); it runs on the
// FP64 matrix cores, v_mfma_f64_16x16x4_f64, one 16 x 16 output tile per wavefront.
//   Re L[i,j] = sum_ab  Re CB_i[a,b] Re C_j[b,a] - Im CB_i[a,b] Im C_j[b,a]
//   Im L[i,j] = sum_ab  Im CB_i[a,b] Re C_j[b,a] + Re CB_i[a,b] Im C_j[b,a]
// so with K = 2 d^2 and  Bop[kk][j] = (Re C_j[b,a] ; Im C_j[b,a]),
//   AopRe[kk][i] = (Re CB_i[a,b] ; -Im CB_i[a,b]),  AopIm[kk][i] = (Im CB_i[a,b] ; Re CB_i[a,b]).
// Operands are stored K-major so that the 16 lanes of an MFMA row group read 128 contiguous bytes.
// The imaginary GEMM is skipped for Hermitian bases, where the reference returns the real part
// only (basis.py:692 `cast`).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "ffk_internal.h"

namespace ffk {
namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

// Bop[kk][j]: kk = b*d + a (Re), d*d + b*d + a (Im) of C_j[b][a]
__global__ void build_bop_kernel(const cplx* __restrict__ basis, int N, int d, int Npad, int hermitian,
                                 double* __restrict__ Bop) {
    const int j = blockIdx.x*blockDim.x + threadIdx.x;
    const int ab = blockIdx.y;  // a*d + b
    if (j >= Npad) return;
    const int a = ab / d, b = ab % d;
    cplx v = {0.0, 0.0};
    if (j < N) v = basis[(static_cast<size_t>(j)*d + b)*d + a];
    if (hermitian) {
        // (rows of hermitian_operand_row: the pair (a, b), (b, a) contributes twice the stored half)
        const int r0 = hermitian_operand_row(a, b, 0, d), r1 = hermitian_operand_row(a, b, 1, d);
        if (r0 >= 0) Bop[static_cast<size_t>(r0)*Npad + j] = a == b ? v.re : 2.0*v.re;
        if (r1 >= 0) Bop[static_cast<size_t>(r1)*Npad + j] = 2.0*v.im;
        return;
    }
    Bop[static_cast<size_t>(ab)*Npad + j] = v.re;
    Bop[static_cast<size_t>(d*d + ab)*Npad + j] = v.im;
}

// ---- the right operand's non-zeros (round 6) ------------------------------------------------------
// For the two bases the package builds (Pauli: d non-zero entries per element, GGM: one or two, d on the diagonal
// elements) a column of Bop has 8-16 resp. 1-16 non-zeros of its d^2 rows: the contraction with it is a gather of a
// handful of terms, not a GEMM.  One wavefront per column lists them in ascending row order (64 rows per step,
// positions from the ballot); a column with more than kLvNzMax raises `dense`, and the call runs as the GEMM.
#ifdef FFK_LV_TRACE   /* tuning build: when is a block of the fused conjugation in which phase? (100 MHz ticks) */
__device__ unsigned long long g_lv_trace[6*8192];
#define FFK_LV_STAMP_AT(k, T) \
    if (threadIdx.x == (T)) { \
        const unsigned lv_l = blockIdx.x + gridDim.x*blockIdx.y; \
        if (lv_l < 8192) g_lv_trace[6*lv_l + (k)] = __builtin_amdgcn_s_memrealtime(); \
    }
#define FFK_LV_STAMP(k) \
    if (threadIdx.x == 0) { \
        const unsigned lv_l = blockIdx.x + gridDim.x*blockIdx.y; \
        if (lv_l < 8192) g_lv_trace[6*lv_l + (k)] = __builtin_amdgcn_s_memrealtime(); \
    }
#define FFK_LV_HWID \
    if (threadIdx.x == 0) { \
        const unsigned lv_l = blockIdx.x + gridDim.x*blockIdx.y; \
        if (lv_l < 8192) g_lv_trace[6*lv_l + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | \
            (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11))) << 32); \
    }
#else
#define FFK_LV_STAMP(k)
#define FFK_LV_STAMP_AT(k, T)
#define FFK_LV_HWID
#endif
constexpr int kLvNzMax = 32;
struct OperandLists {
    int* dense;        // != 0: some column has more than kLvNzMax non-zeros
    int* count;        // [Npad]
    int* row;          // [Npad][kLvNzMax]  (as offsets into the fused kernel's tile, fused_tile_offset) a column's
                       //                   entries side by side (a thread fetches 16 of them with four
    double* value;     // [Npad][kLvNzMax]  resp. eight 16-byte loads); entries past the count: row 0, value 0
};
size_t operand_lists_bytes(int Npad) {
    return align_up(sizeof(int)) + align_up(sizeof(int)*Npad) + align_up(sizeof(int)*kLvNzMax*Npad) +
           align_up(sizeof(double)*kLvNzMax*Npad);
}
OperandLists slice_operand_lists(void* ws, int Npad) {
    unsigned char* p = static_cast<unsigned char*>(ws);
    OperandLists L;
    L.dense = reinterpret_cast<int*>(p);   p += align_up(sizeof(int));
    L.count = reinterpret_cast<int*>(p);   p += align_up(sizeof(int)*Npad);
    L.row = reinterpret_cast<int*>(p);     p += align_up(sizeof(int)*kLvNzMax*Npad);
    L.value = reinterpret_cast<double*>(p);
    return L;
}
// Row k of the Hermitian operand (hermitian_operand_row) -> slot of the persistent fused kernel's tile: entry (a, b),
// a <= b, keeps its real part at a d + b and (a < b) minus its imaginary part at b d + a -- slots a lane of the
// conjugation reaches with one per-lane base and an immediate.
__device__ inline int hermitian_row_to_slot(int k, int d) {
    if (k < d) return k*d + k;
    const int p = (k - d) >> 1, imag_part = (k - d) & 1;
    int a = 0;
    while ((a + 1)*(2*d - a - 2)/2 <= p) ++a;
    const int b = p - a*(2*d - a - 1)/2 + a + 1;
    return imag_part ? b*d + a : a*d + b;
}

// ... and the slot's place in the tile, in doubles: 17 per slot (16 elements + 1) and one more per d slots, so that
// BOTH neighbours of an entry -- (a, b + 1) and (a + 1, b), 1 resp. d slots away -- start 34 banks further on: a
// wavefront's 64 columns ask for rows that differ in a or in b (Pauli: eight distinct b for the real parts, eight
// distinct a for the imaginary parts); with 17 d doubles between (a, b) and (a + 1, b) every second of those met in one
// bank (SQ_LDS_BANK_CONFLICT: 55 % of the LDS cycles of the first build)
__host__ __device__ constexpr int fused_tile_offset(int slot, int d) { return slot*17 + slot/d; }
__host__ __device__ constexpr int fused_tile_doubles(int d) { return d*d*17 + d; }

__global__ __launch_bounds__(64) void operand_lists_kernel(const double* __restrict__ Bop, int K, int Npad,
                                                           OperandLists L, int slots_of_d) {
    const int j = blockIdx.x, lane = threadIdx.x;
    int* rows = L.row + static_cast<size_t>(j)*kLvNzMax;
    double* values = L.value + static_cast<size_t>(j)*kLvNzMax;
    int n = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const double v = k < K ? Bop[static_cast<size_t>(k)*Npad + j] : 0.0;
        const bool nz = v != 0.0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(nz);
        const int pos = n + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (nz && pos < kLvNzMax) {
            rows[pos] = slots_of_d > 0 ? fused_tile_offset(hermitian_row_to_slot(k, slots_of_d), slots_of_d) : k;
            values[pos] = v;
        }
        n += __builtin_popcountll(mask);
    }
    if (lane >= n && lane < kLvNzMax) {
        rows[lane] = 0;
        values[lane] = 0.0;
    }
    if (lane == 0) {
        L.count[j] = min(n, kLvNzMax);
        if (n > kLvNzMax) atomicOr(L.dense, 1);
    }
}

// The LDS tile [2 d^2][EPB + 1] of a block's EPB conjugated elements (row kk = a d + b: Re CB[a,b],
// d^2 + a d + b: -Im CB[a,b]; `transposed_slots`: the rows kernel keeps entry (a, b) in slot b d + a)
// to the K-major operand(s): EPB consecutive doubles per row.  Hermitian basis (no imaginary operand):
// only the rows of hermitian_operand_row are written, K = d^2.
template <int D, int EPB>
__device__ __forceinline__ void copy_out_operand_tile(const double* tre, const double* tim, int want_imag,
                                                      double* __restrict__ AopRe,
                                                      double* __restrict__ AopIm, int bt, int Npad, int i0,
                                                      int N, int tid, bool transposed_slots) {
    constexpr int DD = D*D, ROW = EPB + 1;
    const size_t K = liouville_operand_rows(D, want_imag);     // pad rows stay zero
    double* are = AopRe + static_cast<size_t>(bt)*K*Npad;
    double* aim = AopIm + static_cast<size_t>(bt)*K*Npad;
    for (int idx = tid; idx < 2*DD*EPB; idx += 256) {
        const int kk = idx / EPB, jj = idx % EPB;
        if (i0 + jj >= N) continue;
        const int half = kk / DD, e = kk % DD, a = e / D, b = e % D;
        const int slot = transposed_slots ? half*DD + b*D + a : kk;
        if (want_imag) {
            are[static_cast<size_t>(kk)*Npad + i0 + jj] = tre[slot*ROW + jj];
            aim[static_cast<size_t>(kk)*Npad + i0 + jj] = tim[slot*ROW + jj];
        } else {
            const int r = hermitian_operand_row(a, b, half, D);
            if (r >= 0) are[static_cast<size_t>(r)*Npad + i0 + jj] = tre[slot*ROW + jj];
        }
    }
}

// Conjugation of the basis with COALESCED operand stores.  The GEMM wants its A operand K-major,
// Aop[kk][i]: one basis element is a column, and the round-2 kernel (one wavefront per element) wrote
// each of its 2 d^2 numbers to a line of its own (8 bytes per 2 KiB at d = 16: 4.3 GB of write
// transactions for 0.5 GB of payload at batch 512 -- 862 us, more than the GEMM).  Here a 256-thread block conjugates EPB
// consecutive elements with one unitary, one matrix entry per thread, parks the results in an LDS
// tile [kk][EPB] (rows padded by one double: conflict free) and writes rows of EPB doubles -- 128
// contiguous bytes per kk for EPB = 16.
template <int D, int EPB>
__global__ __launch_bounds__(256) void conjugate_basis_tile_kernel(const cplx* __restrict__ U,
                                                                   const cplx* __restrict__ basis, int N,
                                                                   int Npad, int want_imag,
                                                                   double* __restrict__ AopRe,
                                                                   double* __restrict__ AopIm) {
    constexpr int DD = D*D, ROW = EPB + 1;
    __shared__ cplx Us[DD];
    __shared__ cplx Cs[DD];
    __shared__ cplx CUs[DD];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* tre = reinterpret_cast<double*>(lds_raw);                  // [2 DD][ROW]
    double* tim = tre + 2*DD*ROW;                                      // [2 DD][ROW] (want_imag only)
    const int bt = blockIdx.y, tid = threadIdx.x;
    const int i0 = blockIdx.x*EPB;
    for (int e = tid; e < DD; e += 256) Us[e] = U[static_cast<size_t>(bt)*DD + e];
    for (int j = 0; j < EPB; ++j) {
        const int i = i0 + j;
        __syncthreads();                 // Us staged / previous element's Cs and CUs consumed
        if (i < N)
            for (int e = tid; e < DD; e += 256) Cs[e] = basis[static_cast<size_t>(i)*DD + e];
        __syncthreads();
        if (i < N)
            for (int e = tid; e < DD; e += 256) {
                const int r = e / D, c = e % D;
                cplx acc = {0.0, 0.0};
#pragma unroll
                for (int k = 0; k < D; ++k) cmac(acc, Cs[r*D + k], Us[k*D + c]);
                CUs[e] = acc;
            }
        __syncthreads();
        for (int e = tid; e < DD; e += 256) {
            cplx acc = {0.0, 0.0};
            if (i < N) {
                const int a = e / D, b = e % D;  // CB[a][b] = sum_k conj(U[k][a]) CU[k][b]
#pragma unroll
                for (int k = 0; k < D; ++k) cmac_conj(acc, Us[k*D + a], CUs[k*D + b]);
            }
            tre[e*ROW + j] = acc.re;
            tre[(DD + e)*ROW + j] = -acc.im;
            if (want_imag) {
                tim[e*ROW + j] = acc.im;
                tim[(DD + e)*ROW + j] = acc.re;
            }
        }
    }
    __syncthreads();
    copy_out_operand_tile<D, EPB>(tre, tim, want_imag, AopRe, AopIm, bt, Npad, i0, N, tid, false);
}

// The same conjugation with the block's EPB = 256/D basis elements worked on AT ONCE: thread (j, r)
// owns row r of element j -- U^dag C_j U as two products of a register-held row with U (read from LDS,
// one address per wavefront and step: broadcast), meeting once in LDS in between.  The tile kernel
// above walks its 16 elements one after the other with three barriers each; here a block has four.
// Used at d = 8 (whole call at batch 512: 56.3 -> 49.4 us); at d = 16 it is no faster than the tile
// kernel (both are bound by reading an entry of U from LDS per complex multiply-add): see the
// matrix-core kernel below.
template <int D>
__global__ __launch_bounds__(256) void conjugate_basis_rows_kernel(const cplx* __restrict__ U,
                                                                   const cplx* __restrict__ basis, int N,
                                                                   int Npad, int want_imag,
                                                                   double* __restrict__ AopRe,
                                                                   double* __restrict__ AopIm) {
    constexpr int DD = D*D, EPB = 256/D, ROW = EPB + 1;
    constexpr int RS = D + 1;                         // row stride of a CU matrix in LDS (cplx)
    constexpr int ES = D*RS + 1;                      // element stride (cplx): rows and elements on distinct banks
    __shared__ cplx Us[DD];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* CUs = reinterpret_cast<cplx*>(lds_raw);                      // [EPB][D][RS], then reused as
    double* tre = reinterpret_cast<double*>(lds_raw);                  // [2 DD][ROW]
    double* tim = tre + 2*DD*ROW;                                      // [2 DD][ROW] (want_imag only)
    const int bt = blockIdx.y, tid = threadIdx.x;
    const int j = tid / D, r = tid % D;
    const int i0 = blockIdx.x*EPB, i = i0 + j;
    for (int e = tid; e < DD; e += 256) Us[e] = U[static_cast<size_t>(bt)*DD + e];
    cplx crow[D];
#pragma unroll
    for (int k = 0; k < D; ++k) crow[k] = i < N ? basis[static_cast<size_t>(i)*DD + r*D + k] : cplx{0.0, 0.0};
    __syncthreads();
    // CU[r][c] = sum_k C[r][k] U[k][c]
#pragma unroll
    for (int c = 0; c < D; ++c) {
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) cmac(acc, crow[k], Us[k*D + c]);
        CUs[j*ES + r*RS + c] = acc;
    }
    __syncthreads();
    // CB[a][b] = sum_k conj(U[k][a]) CU[k][b],  a = r
    cplx cb[D];
#pragma unroll
    for (int b = 0; b < D; ++b) cb[b] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const cplx ua = Us[k*D + r];
#pragma unroll
        for (int b = 0; b < D; ++b) cmac_conj(cb[b], ua, CUs[j*ES + k*RS + b]);
    }
    __syncthreads();                     // CU consumed by everybody: its space becomes the output tile
    // tile slot of entry e = a D + b: b D + a, so that the 16 threads of an element write 16
    // consecutive rows of the tile (ROW doubles apart: conflict free)
#pragma unroll
    for (int b = 0; b < D; ++b) {
        const int slot = b*D + r;
        tre[slot*ROW + j] = cb[b].re;
        tre[(DD + slot)*ROW + j] = -cb[b].im;
        if (want_imag) {
            tim[slot*ROW + j] = cb[b].im;
            tim[(DD + slot)*ROW + j] = cb[b].re;
        }
    }
    __syncthreads();
    copy_out_operand_tile<D, EPB>(tre, tim, want_imag, AopRe, AopIm, bt, Npad, i0, N, tid, true);
}

// d = 12, 16 (8 on request): the conjugation U^dag C_i U on the FP64 matrix cores, in the block-frequency form of the
// accumulate kernels (ctrl_mfma.hip) with a BASIS ELEMENT per 4 x 4 x 4 block instead of a frequency:
// v_mfma_f64_4x4x4_4b computes four independent products, lane (c, q) supplies A_b[c & 3][q] and
// B_b[q][c & 3] of block b = c >> 2 and receives D_b[q][c & 3], so the first product
// P = C_i^T conj(U) is, register for register, the transposed A operand of the second,
// Y = P^T U = U^dag C_i U.  The entries of U a lane needs (16) live in registers for the whole
// block; a wavefront conjugates four elements, a block of four wavefronts the same 16 elements as
// the tile kernel, whose LDS output tile and coalesced copy-out it shares.  The vector kernels read
// an entry of U from LDS per complex multiply-add and are bound by that (rows kernel above: no
// faster than the tile kernel at d = 16).
// HERM (Hermitian basis): only the blocks of Y on and above the diagonal are formed -- the operand
// keeps the entries a <= b (hermitian_operand_row): 10 of 16 block pairs of the second product at d = 16.
// FUSED (round 6, Hermitian basis whose operand columns are short lists, OperandLists): the block contracts its 16
// conjugated elements with the right operand's non-zeros straight from the LDS tile -- thread = column j, 16 sums
// each -- and writes 16 rows of L; the 0.25 GB operand (d = 16, batch 512) is neither written nor read back, the GEMM
// does not run.  `dense` decides on the device which of the two forms works; the other one's blocks all return.
template <int D, bool HERM, bool FUSED = false>
__global__ __launch_bounds__(256, FUSED ? 3 : 1) void conjugate_basis_mfma_kernel(
    const cplx* __restrict__ U,
                                                                     const cplx* __restrict__ basis, int N,
                                                                     int Npad, int want_imag,
                                                                     double* __restrict__ AopRe,
                                                                     double* __restrict__ AopIm,
                                                                     OperandLists lists,
                                                                     double* __restrict__ out) {
    static_assert(D % 4 == 0 && D <= 16, "d = 4, 8, 12, 16");
    static_assert(HERM || !FUSED, "the fused form keeps the Hermitian operand's rows");
    constexpr int DD = D*D, NS = D/4, EPB = 16, ROW = EPB + 1;
    // which form works is decided on the device (every block of a launch alike).  The fused form reads the flag
    // BESIDE its first operands and looks at it behind the first barrier: as the kernel's first statement it was a
    // round trip to memory of its own at the head of every block (2-3 us under this kernel's load, of 20 per block:
    // tools/tuning/trace_liouville_blocks.py)
    int dense_now = 0;
    if constexpr (FUSED) {
        FFK_LV_STAMP(0)
        FFK_LV_HWID
        dense_now = *lists.dense;
    } else if (lists.dense != nullptr && *lists.dense == 0) {
        return;
    }
    __shared__ cplx Us[DD];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* tre = reinterpret_cast<double*>(lds_raw);                  // [2 DD][ROW]
    double* tim = tre + 2*DD*ROW;                                      // [2 DD][ROW] (want_imag only)
    const int bt = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4, c4 = c & 3, b = c >> 2;
    const int i0 = blockIdx.x*EPB;
    const int j = 4*wave + b, i = i0 + j;             // this lane's basis element
    const cplx* Ci = basis + static_cast<size_t>(min(i, N - 1))*DD;
    // the element's first column block is requested before U is staged: it does not depend on U, and behind the
    // barrier its trip to L2 would be the second of two in a row at the head of every block (round 6)
    cplx x0[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) x0[s] = Ci[(4*s + q)*D + c4];
    if (tid < DD) Us[tid] = U[static_cast<size_t>(bt)*DD + tid];
    __syncthreads();
    if (FUSED && dense_now != 0) return;
    cplx tq[NS][NS];                                  // U[4 s + q][4 g + c4]
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int g = 0; g < NS; ++g) tq[s][g] = Us[(4*s + q)*D + 4*g + c4];
    if constexpr (FUSED) { FFK_LV_STAMP(1) }
    double Yr[NS][NS], Yi[NS][NS];                    // Y[4 ig + q][4 jg + c4]
#pragma unroll
    for (int ig = 0; ig < NS; ++ig)
#pragma unroll
        for (int jg = 0; jg < NS; ++jg) {
            Yr[ig][jg] = 0.0;
            Yi[ig][jg] = 0.0;
        }
#pragma unroll
    for (int ng = 0; ng < NS; ++ng) {
        double pr[NS], pi[NS];                        // P[4 ng + q][4 ig + c4]
#pragma unroll
        for (int ig = 0; ig < NS; ++ig) {
            pr[ig] = 0.0;
            pi[ig] = 0.0;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const cplx x = ng == 0 ? x0[s] : Ci[(4*s + q)*D + 4*ng + c4];           // C_i[4 s + q][4 ng + c4]
#pragma unroll
            for (int ig = 0; ig < NS; ++ig) {
                pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].re, pr[ig], 0, 0, 0);
                pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].re, pi[ig], 0, 0, 0);
            }
#pragma unroll
            for (int ig = 0; ig < NS; ++ig) {
                pr[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.im, tq[s][ig].im, pr[ig], 0, 0, 0);
                pi[ig] = __builtin_amdgcn_mfma_f64_4x4x4f64(x.re, tq[s][ig].im, pi[ig], 0, 0, 1);
            }
        }
#pragma unroll
        for (int ig = 0; ig < NS; ++ig) {
#pragma unroll
            for (int jg = HERM ? ig : 0; jg < NS; ++jg) {
                Yr[ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].re, Yr[ig][jg], 0, 0, 0);
                Yi[ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pr[ig], tq[ng][jg].im, Yi[ig][jg], 0, 0, 0);
            }
#pragma unroll
            for (int jg = HERM ? ig : 0; jg < NS; ++jg) {
                Yr[ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].im, Yr[ig][jg], 0, 0, 1);
                Yi[ig][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(pi[ig], tq[ng][jg].re, Yi[ig][jg], 0, 0, 0);
            }
        }
    }
    const bool valid = i < N;
    if constexpr (FUSED) { FFK_LV_STAMP(2) }
    if constexpr (HERM) {
        if constexpr (FUSED) {
            // entry (a, b) = (4 ig + q, 4 jg + c4), a <= b: Re at slot a D + b, -Im (a < b) at slot b D + a, slots
            // placed by fused_tile_offset: one per-lane base each and an immediate (the operand's own row numbers cost
            // ~200 integer instructions per block here)
            const int base_re = fused_tile_offset(q*D + c4, D) + j, base_im = fused_tile_offset(c4*D + q, D) + j;
#pragma unroll
            for (int ig = 0; ig < NS; ++ig)
#pragma unroll
                for (int jg = ig; jg < NS; ++jg) {
                    if (ig < jg || q <= c4) tre[base_re + (4*ig*D + 4*jg)*ROW + 4*ig] = Yr[ig][jg];
                    if (ig < jg || q < c4) tre[base_im + (4*jg*D + 4*ig)*ROW + 4*jg] = -Yi[ig][jg];
                }
        } else {
            // the tile holds the operand's rows directly (hermitian_operand_row: d^2 of them), and the
            // copy-out writes 16 bytes per lane: eight lanes per 128-byte row
#pragma unroll
            for (int ig = 0; ig < NS; ++ig)
#pragma unroll
                for (int jg = ig; jg < NS; ++jg) {
                    const int a = 4*ig + q, b2 = 4*jg + c4;
                    const int r0 = hermitian_operand_row(a, b2, 0, D), r1 = hermitian_operand_row(a, b2, 1, D);
                    if (r0 >= 0) tre[r0*ROW + j] = valid ? Yr[ig][jg] : 0.0;
                    if (r1 >= 0) tre[r1*ROW + j] = valid ? -Yi[ig][jg] : 0.0;
                }
        }
        if constexpr (FUSED) {
            // L[bt][i0 + jj][col] = sum over the non-zero rows k of column col: tile[k][jj] Bop[k][col], k ascending.
            // Sixteen list entries are requested at once -- the first batch before the barrier that completes the
            // tile, so that the trip to L2 runs beside the other wavefronts' last matrix instructions -- (a padding
            // entry reads row 0 with weight 0: a broadcast).
            constexpr int NB = 16;
            using int4_t = __attribute__((ext_vector_type(4))) int;
            using double2_t = __attribute__((ext_vector_type(2))) double;
            int4_t kk[NB/4];
            double2_t vv[NB/2];
            auto request = [&](int col, int base) __attribute__((always_inline)) {
                const int4_t* rp = reinterpret_cast<const int4_t*>(lists.row + static_cast<size_t>(col)*kLvNzMax + base);
                const double2_t* vp =
                    reinterpret_cast<const double2_t*>(lists.value + static_cast<size_t>(col)*kLvNzMax + base);
#pragma unroll
                for (int t = 0; t < NB/4; ++t) kk[t] = rp[t];
#pragma unroll
                for (int t = 0; t < NB/2; ++t) vv[t] = vp[t];
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the requests stay behind the tile's stores)
            // (the fused form is launched for complete bases only, N = d^2: a column per thread, no edge)
            const int col = tid < DD ? tid : 0;
            // the longest list among the wavefront's columns: the gather's trip count (wave-uniform; a Pauli element
            // has 8 or 16 non-zero rows, a GGM element mostly 1 or 2)
            int n_wave = lists.count[col];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) n_wave = max(n_wave, __shfl_xor(n_wave, off, 64));
            n_wave = __builtin_amdgcn_readfirstlane(n_wave);
            request(col, 0);
            __syncthreads();
            FFK_LV_STAMP(3)
            if (tid >= DD) return;
            double acc[EPB];
#pragma unroll
            for (int jj = 0; jj < EPB; ++jj) acc[jj] = 0.0;
            // A row of the tile = eight ds_read2_b64 as ONE asm statement (the list entry IS the row's offset in the
            // tile, fused_tile_offset); rows go in pairs, the second requested before the first is waited for
            // (s_waitcnt lgkmcnt(8): the eight younger reads may still fly).  The statements name the running sums as
            // inputs they do not use, so that a request stays behind the multiply-adds of the pair before -- whose
            // registers it reuses --: left to the scheduler all 256 reads of a batch are requested first and ~500
            // registers spill.
#define FFK_LV_READ_ROW(r, k)                                                                                        \
    {                                                                                                                \
        const unsigned addr_ = static_cast<unsigned>(reinterpret_cast<uintptr_t>(tre + (k)));                       \
        asm volatile("ds_read2_b64 %0, %8 offset1:1\n\t"                                                            \
                     "ds_read2_b64 %1, %8 offset0:2 offset1:3\n\t"                                                  \
                     "ds_read2_b64 %2, %8 offset0:4 offset1:5\n\t"                                                  \
                     "ds_read2_b64 %3, %8 offset0:6 offset1:7\n\t"                                                  \
                     "ds_read2_b64 %4, %8 offset0:8 offset1:9\n\t"                                                  \
                     "ds_read2_b64 %5, %8 offset0:10 offset1:11\n\t"                                                \
                     "ds_read2_b64 %6, %8 offset0:12 offset1:13\n\t"                                                \
                     "ds_read2_b64 %7, %8 offset0:14 offset1:15"                                                     \
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]),    \
                       "=&v"(r[7])                                                                                   \
                     : "v"(addr_), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]),     \
                       "v"(acc[6]), "v"(acc[7]), "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), \
                       "v"(acc[13]), "v"(acc[14]), "v"(acc[15])                                                      \
                     : "memory");                                                                                    \
    }
#define FFK_LV_WAIT_ROW(r, younger)                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(" #younger ")"                                                                   \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])    \
                 :                                                                                                   \
                 : "memory");
            for (int base = 0;;) {
                double2_t ra[EPB/2], rb[EPB/2];
#pragma unroll
                for (int t = 0; t < NB; t += 2) {
                    if (base + t >= n_wave) break;
                    FFK_LV_READ_ROW(ra, kk[t/4][t%4])
                    FFK_LV_READ_ROW(rb, kk[(t + 1)/4][(t + 1)%4])
                    FFK_LV_WAIT_ROW(ra, 8)
                    {
                        const double v = vv[t/2][t%2];
#pragma unroll
                        for (int jj = 0; jj < EPB; ++jj) acc[jj] = fma(ra[jj/2][jj%2], v, acc[jj]);
                    }
                    FFK_LV_WAIT_ROW(rb, 0)
                    {
                        const double v = vv[(t + 1)/2][(t + 1)%2];
#pragma unroll
                        for (int jj = 0; jj < EPB; ++jj) acc[jj] = fma(rb[jj/2][jj%2], v, acc[jj]);
                    }
                }
                base += NB;
                if (base >= n_wave) break;
                request(col, base);
            }
#undef FFK_LV_READ_ROW
#undef FFK_LV_WAIT_ROW
            double* o = out + (static_cast<size_t>(bt)*DD + i0)*DD + col;
#pragma unroll
            for (int jj = 0; jj < EPB; ++jj) o[jj*DD] = acc[jj];
            FFK_LV_STAMP(4)
            return;
        }
        __syncthreads();
        using double2_t = __attribute__((ext_vector_type(2))) double;
        double* are = AopRe + static_cast<size_t>(bt)*liouville_operand_rows(D, 0)*Npad + i0;
        for (int idx = tid; idx < DD*(EPB/2); idx += 256) {
            const int r = idx/(EPB/2), cp = 2*(idx % (EPB/2));
            const double2_t v = {tre[r*ROW + cp], tre[r*ROW + cp + 1]};
            double* dst = are + static_cast<size_t>(r)*Npad + cp;
            if (i0 + cp + 1 < N) *reinterpret_cast<double2_t*>(dst) = v;
            else if (i0 + cp < N) dst[0] = v.x;
        }
        return;
    }
#pragma unroll
    for (int ig = 0; ig < NS; ++ig)
#pragma unroll
        for (int jg = 0; jg < NS; ++jg) {
            const int e = (4*ig + q)*D + 4*jg + c4;
            const double re = valid ? Yr[ig][jg] : 0.0, im = valid ? Yi[ig][jg] : 0.0;
            tre[e*ROW + j] = re;
            tre[(DD + e)*ROW + j] = -im;
            if (want_imag) {
                tim[e*ROW + j] = im;
                tim[(DD + e)*ROW + j] = re;
            }
        }
    __syncthreads();
    copy_out_operand_tile<D, EPB>(tre, tim, want_imag, AopRe, AopIm, bt, Npad, i0, N, tid, false);
}

// One wavefront per (16 TM) x (16 TN) tile of L = Aop^T Bop.  v_mfma_f64_16x16x4_f64 operand maps
// (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4) + 4 r][col = lane&15] for result register r = 0..3.  With one 16 x 16 tile per
// wavefront every MFMA (2048 flops) waits for 1 KiB of operands from L2: 25.8 TFLOP/s, MFMA pipe
// 34 % busy at d = 16 (profiles/r02_d_*).  TM x TN tiles per wavefront re-use each operand TN resp. TM
// times from registers and keep TM*TN independent accumulators in flight (a dependent
// v_mfma_f64_16x16x4 waits out the 16 passes of its predecessor).
// IMAG is a template parameter: for Hermitian bases (the common case) the imaginary accumulators do
// not exist, which halves the register count -- 4 x 4 tiles then fit 3 wavefronts per SIMD instead of
// one, and a wavefront's operand fetches (straight from L2, nothing prefetched) are covered by the
// others' matrix instructions.
template <int TM, int TN, bool IMAG>
__global__ __launch_bounds__(64, (TM == 4 && !IMAG) ? 3 : 1) void liouville_gemm_kernel(const double* __restrict__ AopRe,
                                                            const double* __restrict__ AopIm,
                                                            const double* __restrict__ Bop, int N,
                                                            int Npad, int K,
                                                            double* __restrict__ out,
                                                            const int* __restrict__ only_if_dense) {
    if (only_if_dense != nullptr && *only_if_dense == 0) return;       // (served by the fused conjugation)
    const int lane = threadIdx.x;
    const int ti = blockIdx.x, tj = blockIdx.y, bt = blockIdx.z;
    const int l15 = lane & 15, lk = lane >> 4;
    // tile columns beyond Npad (partial tile groups) are clamped: their results are not stored
    const double* are[TM];
    const double* aim[TM];
    const double* bop[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int col = min((ti*TM + m)*16 + l15, Npad - 1);
        are[m] = AopRe + static_cast<size_t>(bt)*K*Npad + col;
        aim[m] = AopIm + static_cast<size_t>(bt)*K*Npad + col;
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) bop[n] = Bop + min((tj*TN + n)*16 + l15, Npad - 1);
    f64x4 cre[TM][TN], cim[IMAG ? TM : 1][IMAG ? TN : 1];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            cre[m][n] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (IMAG) cim[m][n] = {0.0, 0.0, 0.0, 0.0};
        }
    for (int k0 = 0; k0 < K; k0 += 4) {
        const size_t row = static_cast<size_t>(k0 + lk)*Npad;
        double a[TM], ai[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            a[m] = are[m][row];
            if constexpr (IMAG) ai[m] = aim[m][row];
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = bop[n][row];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                cre[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], cre[m][n], 0, 0, 0);
        if constexpr (IMAG) {
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    cim[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[m], b[n], cim[m][n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = (tj*TN + n)*16 + l15;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = (ti*TM + m)*16 + lk + 4*r;
                if (rowi >= N) continue;
                const size_t o = (static_cast<size_t>(bt)*N + rowi)*N + col;
                if constexpr (IMAG) {
                    out[2*o] = cre[m][n][r];
                    out[2*o + 1] = cim[m][n][r];
                } else {
                    out[o] = cre[m][n][r];
                }
            }
        }
}

// ---- the same product with operands shared through LDS (Hermitian bases, Npad a multiple of 256) ----
// The register-fed kernel above reads 64 columns of Aop and 64 of Bop per 64 x 64 tile straight from
// L2: 8 flops per byte, 4 GB per call at d = 16 / batch 512, and it runs at the rate L2 delivers them
// (5.5 TB/s, 0.58 of the matrix peak).  Here a workgroup of four wavefronts owns 128 rows (one batch
// element's basis elements i) x 128 columns (j): all batch elements stacked, the product is
// (batch N) x K by K x N with the SAME right operand for every row block.  K is walked 16 rows at a
// time: every thread fetches its share of the next 16 rows of both operands into registers while the
// matrix cores work on the current ones from LDS (two buffers, one barrier per 16 rows = per 64
// matrix instructions of a wavefront); a wavefront's 64 x 64 tile is 16 accumulators of 16 x 16.
// d = 16, batch 512: 750 -> 600 us, 57 TFLOP/s = 0.73 of the matrix peak (profiles/r04_l_*).
// LDS rows are padded by 16 doubles: the four k rows of an operand fragment then start 32 banks
// apart, so the two 16-lane groups of a half-wavefront never meet in a bank.
constexpr int kLbMT = 128, kLbKS = 16, kLbPad = 16;
constexpr int kLbSA = kLbMT + kLbPad;
// NTW: 16-column tiles per wavefront; the workgroup (2 x 2 wavefronts) owns 128 x 32 NTW.  Built with
// NTW = 4: 128 x 128 and TWO workgroups per CU, so that one's barrier, LDS waits, first fetch and
// result stores are covered by the other's matrix instructions (16 flops per byte from L2).  The
// 128 x 256 form (NTW = 8, 256 accumulator registers, one workgroup per CU, 22 flops per byte) was
// 4 % slower at d = 16 / batch 512 (profiles/r04_l_*).
template <int NTW>
constexpr size_t lb_lds_bytes() { return sizeof(double)*2*kLbKS*(kLbSA + 32*NTW + kLbPad); }

template <int NTW>
__global__ __launch_bounds__(256, NTW >= 8 ? 1 : 2) void liouville_gemm_block_kernel(
    const double* __restrict__ Aop, const double* __restrict__ Bop, int N, int Npad, int K,
    double* __restrict__ out, const int* __restrict__ only_if_dense) {
    if (only_if_dense != nullptr && *only_if_dense == 0) return;       // (served by the fused conjugation)
    using double2_t = __attribute__((ext_vector_type(2))) double;
    constexpr int NT = 32*NTW, SB = NT + kLbPad;
    constexpr int BT = NT/2;                      // threads per row of the right operand's 16-row slab
    constexpr int BR = 256/BT, BP = kLbKS/BR;     // rows per pass, passes
    extern __shared__ __attribute__((aligned(16))) double lb_lds[];
    double* As = lb_lds;                               // [2][16][128 + 16]
    double* Bs = lb_lds + 2*kLbKS*kLbSA;               // [2][16][NT + 16]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    const int mt_per = Npad/kLbMT;
    const int bt = blockIdx.x/mt_per, i0 = (blockIdx.x % mt_per)*kLbMT, j0 = blockIdx.y*NT;
    const double* Ag = Aop + static_cast<size_t>(bt)*K*Npad + i0 + 2*(t & 63) + static_cast<size_t>(t >> 6)*Npad;
    const double* Bg = Bop + j0 + 2*(t % BT) + static_cast<size_t>(t / BT)*Npad;
    double* Aw = As + (t >> 6)*kLbSA + 2*(t & 63);
    double* Bw = Bs + (t / BT)*SB + 2*(t % BT);
    double2_t ra[4], rb[BP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2)
            ra[p2] = *reinterpret_cast<const double2_t*>(Ag + static_cast<size_t>(k0 + 4*p2)*Npad);
#pragma unroll
        for (int p2 = 0; p2 < BP; ++p2)
            rb[p2] = *reinterpret_cast<const double2_t*>(Bg + static_cast<size_t>(k0 + BR*p2)*Npad);
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2)
            *reinterpret_cast<double2_t*>(Aw + (buf*kLbKS + 4*p2)*kLbSA) = ra[p2];
#pragma unroll
        for (int p2 = 0; p2 < BP; ++p2)
            *reinterpret_cast<double2_t*>(Bw + (buf*kLbKS + BR*p2)*SB) = rb[p2];
    };
    f64x4 acc[4][NTW];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = {0.0, 0.0, 0.0, 0.0};
    const int wi = (wave & 1)*64, wj = (wave >> 1)*16*NTW;
    fetch(0);
    park(0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += kLbKS) {
        const int buf = (k0/kLbKS) & 1;
        const bool more = k0 + kLbKS < K;
        if (more) fetch(k0 + kLbKS);
        const double* Ab = As + (buf*kLbKS + lk)*kLbSA + wi + l15;
        const double* Bb = Bs + (buf*kLbKS + lk)*SB + wj + l15;
#pragma unroll
        for (int kk = 0; kk < kLbKS/4; ++kk) {
            double a[4], b[NTW];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = Ab[4*kk*kLbSA + 16*m];
#pragma unroll
            for (int n = 0; n < NTW; ++n) b[n] = Bb[4*kk*SB + 16*n];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (more) park(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int col = j0 + wj + 16*n + l15;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = i0 + wi + 16*m + lk + 4*r;
                if (rowi < N) out[(static_cast<size_t>(bt)*N + rowi)*N + col] = acc[m][n][r];
            }
        }
}

bool liouville_block_gemm_applies(int Npad, int K, int want_imag) {
    return !want_imag && Npad % 128 == 0 && K % kLbKS == 0;
}

}  // namespace

size_t liouville_workspace_bytes(int batch, int d, int N) {
    const size_t Npad = (static_cast<size_t>(N) + 15)/16*16;
    const size_t K = (2*static_cast<size_t>(d)*d + 3)/4*4;
    return align_up(K*Npad*sizeof(double)) + 2*align_up(static_cast<size_t>(batch)*K*Npad*sizeof(double));
}

hipError_t launch_liouville(const cplx* U, int batch, int d, const cplx* basis, int N,
                            int hermitian, double* out, void* ws, hipStream_t stream) {
    const int Npad = (N + 15)/16*16;
    const int want_imag = hermitian ? 0 : 1;
    const int K = liouville_operand_rows(d, want_imag);
    unsigned char* p = static_cast<unsigned char*>(ws);
    double* Bop = reinterpret_cast<double*>(p);
    p += align_up(static_cast<size_t>(K)*Npad*sizeof(double));
    double* AopRe = reinterpret_cast<double*>(p);
    p += align_up(static_cast<size_t>(batch)*K*Npad*sizeof(double));
    double* AopIm = reinterpret_cast<double*>(p);
    if (d*d > 65535) return hipErrorInvalidValue;

    // padding rows (K) and columns (N -> Npad) must contribute nothing: zero the operands once --
    // unless there is no padding at all (d^2 a multiple of 16, e.g. d = 4, 8, 16)
    if (Npad != N || K != (want_imag ? 2 : 1)*d*d) {
        hipError_t err = hipMemsetAsync(ws, 0, liouville_workspace_bytes(batch, d, N), stream);
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(build_bop_kernel, dim3((Npad + 63)/64, d*d), dim3(64), 0, stream, basis, N, d,
                       Npad, hermitian ? 1 : 0, Bop);
    // d = 12, 16 with a Hermitian basis: the fused form for operands of short columns (the lists live in the unused
    // imaginary operand's space); both forms are enqueued, `dense` picks one on the device
    OperandLists lists = {nullptr, nullptr, nullptr, nullptr};
    const bool may_fuse = hermitian && (d == 16 || d == 12) && N == d*d && std::getenv("FFK_LIOUVILLE_GEMM") == nullptr &&
                          operand_lists_bytes(Npad) <= align_up(static_cast<size_t>(batch)*K*Npad*sizeof(double));
    if (may_fuse) {
        lists = slice_operand_lists(AopIm, Npad);
        hipError_t err = hipMemsetAsync(lists.dense, 0, sizeof(int), stream);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(operand_lists_kernel, dim3(Npad), dim3(64), 0, stream, Bop, K, Npad, lists, d);
    }
    // the batch axis rides on grid.y / grid.z (at most 65535 blocks): longer batches (the
    // propagators of a 200 000-segment pulse) go in slabs
    const int tiles = Npad/16;
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = std::min(65535, batch - b0);
        const cplx* Us = U + static_cast<size_t>(b0)*d*d;
        double* are = AopRe + static_cast<size_t>(b0)*K*Npad;
        double* aim = AopIm + static_cast<size_t>(b0)*K*Npad;
        double* o = out + static_cast<size_t>(b0)*N*N*(want_imag ? 2 : 1);
        // d = 8: all elements of a block at once
        constexpr bool rows_form = true;
        bool done = false;
        if (generic_dimension(d)) {
            const hipError_t eg = launch_conjugate_basis_generic(Us, nb, d, basis, N, Npad, K, want_imag, are,
                                                                 aim, stream);
            if (eg != hipSuccess) return eg;
            done = true;
        }
        // d = 12, 16: the conjugation on the matrix cores
        if (d == 16 || d == 12) {
            // (Hermitian basis: the tile holds the operand's d^2 rows only)
            const size_t lds = static_cast<size_t>(want_imag ? 4 : 1)*fused_tile_doubles(d)*sizeof(double);   // (d more than d^2 x 17: the fused form's padding)
            auto go = [&](auto kern) -> hipError_t {
                if (lds > 40*1024) {
                    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(lds));
                    if (e2 != hipSuccess) return e2;
                }
                if (std::getenv("FFK_DEBUG_OCCUPANCY")) {
                    int per_cu = -1;
                    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256,
                                                                       lds);
                    fprintf(stderr, "conjugate_basis_mfma: %zu bytes of LDS -> %d blocks per CU\n", lds, per_cu);
                }
                hipLaunchKernelGGL(kern, dim3((N + 15)/16, nb), dim3(256), lds, stream, Us, basis, N, Npad,
                                   want_imag, are, aim, lists, o);
                return hipGetLastError();
            };
            const hipError_t e3 = d == 16 ? (want_imag ? go(conjugate_basis_mfma_kernel<16, false>)
                                                       : go(conjugate_basis_mfma_kernel<16, true>))
                                            : (want_imag ? go(conjugate_basis_mfma_kernel<12, false>)
                                                         : go(conjugate_basis_mfma_kernel<12, true>));
            if (e3 != hipSuccess) return e3;
            if (may_fuse) {
                const hipError_t e6 = d == 16 ? go(conjugate_basis_mfma_kernel<16, true, true>)
                                              : go(conjugate_basis_mfma_kernel<12, true, true>);
                if (e6 != hipSuccess) return e6;
            }
            done = true;
        }
        if (!done && rows_form && d == 8) {
            const int epb = 256/d, dd = d*d;
            const size_t tile = static_cast<size_t>(want_imag ? 2 : 1)*2*dd*(epb + 1)*sizeof(double);
            const size_t cus = (static_cast<size_t>(epb)*(d*(d + 1) + 1))*sizeof(cplx);
            const size_t lds = std::max(tile, cus);
            auto go = [&](auto kern) -> hipError_t {
                if (lds > 40*1024) {
                    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        static_cast<int>(lds));
                    if (e2 != hipSuccess) return e2;
                }
                hipLaunchKernelGGL(kern, dim3((N + epb - 1)/epb, nb), dim3(256), lds, stream, Us, basis, N,
                                   Npad, want_imag, are, aim);
                return hipGetLastError();
            };
            const hipError_t e3 = d == 16 ? go(conjugate_basis_rows_kernel<16>) : go(conjugate_basis_rows_kernel<8>);
            if (e3 != hipSuccess) return e3;
            done = true;
        }
        if (!done) switch (d) {
#define FFK_CASE(D)                                                                              \
    case D: {                                                                                    \
        constexpr int EPB = 16;                                                                  \
        const size_t lds = static_cast<size_t>(want_imag ? 2 : 1)*2*D*D*(EPB + 1)*sizeof(double); \
        auto kern = conjugate_basis_tile_kernel<D, EPB>;                                         \
        if (lds > 40*1024) {                                                                     \
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),             \
                                                hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                                static_cast<int>(lds));                          \
            if (e2 != hipSuccess) return e2;                                                     \
        }                                                                                        \
        hipLaunchKernelGGL(kern, dim3((N + EPB - 1)/EPB, nb), dim3(256), lds, stream, Us, basis, \
                           N, Npad, want_imag, are, aim);                                        \
        break;                                                                                   \
    }
            FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
            FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
            FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
            default:
                return hipErrorInvalidValue;
        }
        // tiles per wavefront by problem size: enough wavefronts to fill 1024 SIMDs first
        const long waves4 = static_cast<long>((tiles + 3)/4)*((tiles + 3)/4)*nb;
        const long waves2 = static_cast<long>((tiles + 1)/2)*((tiles + 1)/2)*nb;
        auto launch = [&](auto kern, int t) {
            hipLaunchKernelGGL(kern, dim3((tiles + t - 1)/t, (tiles + t - 1)/t, nb), dim3(64), 0, stream, are,
                               aim, Bop, N, Npad, K, o, lists.dense);
        };
        if (liouville_block_gemm_applies(Npad, K, want_imag) && static_cast<long>(nb)*(Npad/128)*(Npad/128) >= 512) {
            auto go = [&](auto kern, size_t lds, int nt) -> hipError_t {
                hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    static_cast<int>(lds));
                if (e4 != hipSuccess) return e4;
                // (grid.x <= 65535 batch elements x Npad/128 row blocks: 2^31 - 1 is the limit there)
                hipLaunchKernelGGL(kern, dim3(nb*(Npad/kLbMT), Npad/nt), dim3(256), lds, stream, are, Bop, N,
                                   Npad, K, o, lists.dense);
                return hipGetLastError();
            };
            const hipError_t e5 = go(liouville_gemm_block_kernel<4>, lb_lds_bytes<4>(), 128);
            if (e5 != hipSuccess) return e5;
        } else if (tiles >= 4 && waves4 >= 2048) {
            if (want_imag) launch(liouville_gemm_kernel<4, 4, true>, 4);
            else launch(liouville_gemm_kernel<4, 4, false>, 4);
        } else if (tiles >= 2 && waves2 >= 2048) {
            if (want_imag) launch(liouville_gemm_kernel<2, 2, true>, 2);
            else launch(liouville_gemm_kernel<2, 2, false>, 2);
        } else {
            if (want_imag) launch(liouville_gemm_kernel<1, 1, true>, 1);
            else launch(liouville_gemm_kernel<1, 1, false>, 1);
        }
    }
    return hipGetLastError();
}

}  // namespace ffk

#ifdef FFK_LV_TRACE
// (tuning build only, not in include/ffk.h) per block of the last fused launch: five stamps and the hardware id
extern "C" int ffk_debug_lv_trace(unsigned long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_lv_trace), sizeof(unsigned long long)*6*n_blocks) != hipSuccess;
}
#endif
