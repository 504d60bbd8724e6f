// post.hip -- everything downstream of the segment sum: chunk reduction, basis expansion
// R = tr(B~ C_k), filter function F = R^dag R and the trapezoid integrals of the infidelity.
// All of these are single-pass, HBM-streaming kernels with omega as the fastest (lane) axis.
#include <algorithm>

#include <cstdlib>

#include "ffk_internal.h"

namespace ffk {
namespace {

// Bt[e] = sum_c Ypart[c*slab + e]   (fixed order -> deterministic)
__global__ void reduce_chunks_kernel(const cplx* __restrict__ Ypart, int chunks, size_t slab,
                                     cplx* __restrict__ Bt) {
    const size_t e = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
    if (e >= slab) return;
    cplx acc = Ypart[e];
    for (int c = 1; c < chunks; ++c) {
        const cplx v = Ypart[static_cast<size_t>(c)*slab + e];
        acc.re += v.re;
        acc.im += v.im;
    }
    Bt[e] = acc;
}

// Basis.expand (basis.py:650-698: tensordot(M, basis, axes=[(-2,-1),(-1,-2)])):
//     R[a,k,w] = sum_ij Bt[a,i,j,w] C_k[j,i].
// Operator bases are sparse (a Pauli element has d non-zeros of d^2, a GGM element 1, 2 or up to
// d), so a first tiny kernel compacts every C_k into (count, row index i*d+j, value) lists and
// the expansion only touches the non-zeros: d^3 instead of d^4 work per (a, w) for Pauli bases,
// ~d^2 for GGM -- the same saving the reference gets from its closed-form ggm_expand
// (basis.py:701-787), without special-casing the basis type.  Dense bases cost what they did.
__global__ __launch_bounds__(64) void basis_compact_kernel(const cplx* __restrict__ basis, int d,
                                                           int* __restrict__ nnz,
                                                           int* __restrict__ rows,
                                                           cplx* __restrict__ vals) {
    basis_compact_one(basis, d, blockIdx.x, threadIdx.x, nnz, rows, vals);
}

// Chunk reduction and basis compaction in ONE launch (they are independent and both precede the
// expansion; every launch saved is ~4-5 us of the config-2 step): blocks [0, nred) reduce,
// blocks [nred, nred + N) compact basis element k = blockIdx.x - nred with their first wavefront.
__global__ __launch_bounds__(256) void reduce_compact_kernel(const cplx* __restrict__ Ypart,
                                                             int chunks, size_t slab,
                                                             cplx* __restrict__ Bt, int nred,
                                                             const cplx* __restrict__ basis, int d,
                                                             int* __restrict__ nnz,
                                                             int* __restrict__ rows,
                                                             cplx* __restrict__ vals) {
    if (static_cast<int>(blockIdx.x) >= nred) {
        if (threadIdx.x < 64)
            basis_compact_one(basis, d, static_cast<int>(blockIdx.x) - nred, threadIdx.x, nnz, rows, vals);
        return;
    }
    const size_t e = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
    if (e >= slab) return;
    cplx acc = Ypart[e];
    for (int c = 1; c < chunks; ++c) {
        const cplx v = Ypart[static_cast<size_t>(c)*slab + e];
        acc.re += v.re;
        acc.im += v.im;
    }
    Bt[e] = acc;
}

// Large d, which expansion?  A basis whose elements have few non-zeros (GGM: 2.5 d^2 in all) is
// expanded fastest by the per-element kernel with 64 frequencies per wavefront (1-KiB row runs, every
// entry of Y read ~2.5 times, from L2 after the first); a dense-ish one (Pauli: d^3 non-zeros) by the
// LDS-staged kernel that reads Y exactly once in 256-byte runs.  The non-zero count lives on the
// device (the compaction runs there), so BOTH kernels are launched for d >= 8 and each wavefront
// decides from the count whether it is the one to work (`want_sparse`: 1 / 0 = this kernel is the
// sparse / the dense form, -1 = unconditional).  Config 5 (GGM, d = 16): 0.97 -> 0.76 ms.
__device__ __forceinline__ bool basis_is_sparse(const int* __restrict__ nnz, int N, int dd) {
    int part = 0;
    for (int k = threadIdx.x & 63; k < N; k += 64) part += nnz[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    return part <= 3*dd;
}

// one lane per omega; blockIdx.y = a; blockIdx.z = basis element group of KT
template <int KT>
__global__ __launch_bounds__(64) void expand_sparse_kernel(const cplx* __restrict__ Bt,
                                                           const int* __restrict__ nnz,
                                                           const int* __restrict__ rows,
                                                           const cplx* __restrict__ vals, int N,
                                                           int dd, int W, cplx* __restrict__ R,
                                                           int want_sparse) {
    if (want_sparse >= 0 && basis_is_sparse(nnz, N, dd) != (want_sparse != 0)) return;
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y;
    if (w >= W) return;
    const cplx* b = Bt + static_cast<size_t>(a)*dd*W + w;
    for (int kk = 0; kk < KT; ++kk) {
        const int k = blockIdx.z*KT + kk;
        if (k >= N) break;
        const int n = nnz[k];
        const int* rk = rows + static_cast<size_t>(k)*dd;
        const cplx* vk = vals + static_cast<size_t>(k)*dd;
        cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
        int q = 0;
        for (; q + 1 < n; q += 2) {
            cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
            cmac(acc1, vk[q + 1], b[static_cast<size_t>(rk[q + 1])*W]);
        }
        if (q < n) cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
        R[(static_cast<size_t>(a)*N + k)*W + w] = {acc0.re + acc1.re, acc0.im + acc1.im};
    }
}

// The same with ALL basis elements of an (operator, 64 frequencies) tile in one block of four
// wavefronts, each walking a contiguous quarter of the elements in step with the others.  Elements
// that share their support are then read close together in time and the second reader finds the
// rows of Y in L2: GGM's symmetric element of a pair (i, j) is number 1 + p, the antisymmetric one
// d(d-1)/2 + 1 + p -- a quarter and a bit apart, i.e. a few steps.  With the elements spread over
// grid.z (above) the second read came from HBM: config 5 0.80 -> 0.69 ms.  (Two elements per trip, their four
// rows requested together: no further gain.)
__global__ __launch_bounds__(256) void expand_sparse_quarters_kernel(const cplx* __restrict__ Bt,
                                                                     const int* __restrict__ nnz,
                                                                     const int* __restrict__ rows,
                                                                     const cplx* __restrict__ vals,
                                                                     int N, int dd, int W,
                                                                     cplx* __restrict__ R, int want_sparse) {
    if (want_sparse >= 0 && basis_is_sparse(nnz, N, dd) != (want_sparse != 0)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x*64 + lane;
    const int a = blockIdx.y;
    if (w >= W) return;
    const cplx* b = Bt + static_cast<size_t>(a)*dd*W + w;
    const int quarter = (N + 3)/4;
    const int k1 = min(N, (wave + 1)*quarter);
    for (int k = wave*quarter; k < k1; ++k) {
        const int n = nnz[k];
        const int* rk = rows + static_cast<size_t>(k)*dd;
        const cplx* vk = vals + static_cast<size_t>(k)*dd;
        cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
        int q = 0;
        for (; q + 1 < n; q += 2) {
            cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
            cmac(acc1, vk[q + 1], b[static_cast<size_t>(rk[q + 1])*W]);
        }
        if (q < n) cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
        R[(static_cast<size_t>(a)*N + k)*W + w] = {acc0.re + acc1.re, acc0.im + acc1.im};
    }
}

// The same, with elements that share their support computed TOGETHER: GGM's symmetric and
// antisymmetric element of a pair (i, j) both read Y_ij and Y_ji, so the thread that expands one
// expands the other from the same two loads (Y is then read once, not twice: the expansion is a
// streaming kernel, 2.4 GB instead of 3.6 GB at config 5).  The grouping is found by the block
// itself, in LDS, from the compacted lists -- no basis-specific code and no extra workspace: an
// element with at most two non-zeros gets the key (count, row 0, row 1); the first element of every
// key is a leader and chains the others behind it; leaders are dealt to the four wavefronts round
// robin.  Elements with more non-zeros (GGM's diagonal ones) are leaders without followers.
__global__ __launch_bounds__(256) void expand_sparse_groups_kernel(const cplx* __restrict__ Bt,
                                                                   const int* __restrict__ nnz,
                                                                   const int* __restrict__ rows,
                                                                   const cplx* __restrict__ vals, int N,
                                                                   int dd, int W, cplx* __restrict__ R,
                                                                   int want_sparse) {
    if (want_sparse >= 0 && basis_is_sparse(nnz, N, dd) != (want_sparse != 0)) return;
    constexpr int kMaxN = 256;                        // d <= 16
    __shared__ int keys[kMaxN], next[kMaxN], leaders[kMaxN], n_leaders;
    const int tid = threadIdx.x;
    if (tid == 0) n_leaders = 0;
    if (tid < N) {
        const int n = nnz[tid];
        const int* rk = rows + static_cast<size_t>(tid)*dd;
        keys[tid] = (n >= 1 && n <= 2) ? ((n << 28) | (rk[0] << 14) | (n == 2 ? rk[1] : 0x3fff)) : -1 - tid;
    }
    __syncthreads();
    if (tid < N) {
        const int key = keys[tid];
        bool leader = true;
        for (int k = 0; k < tid; ++k) leader &= keys[k] != key;
        int nx = -1;
        for (int k = N - 1; k > tid; --k)
            if (keys[k] == key) nx = k;
        next[tid] = nx;
        // (order of the leader list does not matter for the result: every element is written once)
        if (leader) leaders[atomicAdd(&n_leaders, 1)] = tid;
    }
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = blockIdx.x*64 + lane;
    const int a = blockIdx.y;
    if (w >= W) return;
    const cplx* b = Bt + static_cast<size_t>(a)*dd*W + w;
    cplx* out = R + static_cast<size_t>(a)*N*W + w;
    const int nl = n_leaders;
    for (int idx = wave; idx < nl; idx += 4) {
        const int k = __builtin_amdgcn_readfirstlane(leaders[idx]);
        const int n = nnz[k];
        const int* rk = rows + static_cast<size_t>(k)*dd;
        if (n >= 1 && n <= 2) {
            const cplx y0 = b[static_cast<size_t>(rk[0])*W];
            const cplx y1 = n == 2 ? b[static_cast<size_t>(rk[1])*W] : cplx{0.0, 0.0};
            for (int m = k; m >= 0; m = __builtin_amdgcn_readfirstlane(next[m])) {
                const cplx* vm = vals + static_cast<size_t>(m)*dd;
                // the same operation order as the per-element kernels: acc0 = v0 y0, acc1 = v1 y1, sum
                cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
                cmac(acc0, vm[0], y0);
                if (n == 2) cmac(acc1, vm[1], y1);
                out[static_cast<size_t>(m)*W] = {acc0.re + acc1.re, acc0.im + acc1.im};
            }
        } else {
            const cplx* vk = vals + static_cast<size_t>(k)*dd;
            cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
            int q = 0;
            for (; q + 1 < n; q += 2) {
                cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
                cmac(acc1, vk[q + 1], b[static_cast<size_t>(rk[q + 1])*W]);
            }
            if (q < n) cmac(acc0, vk[q], b[static_cast<size_t>(rk[q])*W]);
            out[static_cast<size_t>(k)*W] = {acc0.re + acc1.re, acc0.im + acc1.im};
        }
    }
}

// out[w,a,i,j] = Bt[a,i,j,w]; a (A*d*d) x W transpose through LDS, 64 x 64 tiles
__global__ __launch_bounds__(256) void transpose_kernel(const cplx* __restrict__ in, int rows,
                                                        int cols, cplx* __restrict__ out) {
    __shared__ cplx tile[64][65];
    const int c0 = blockIdx.x*64, r0 = blockIdx.y*64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = in[static_cast<size_t>(r0 + r)*cols + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (c0 + c < cols && r0 + tx < rows) out[static_cast<size_t>(c0 + c)*rows + r0 + tx] = tile[tx][c];
}

// F[a,b,w] = sum_k conj(R[a,k,w]) R[b,k,w]      ('ako,bko->abo', numeric.py:1462).
// Only a <= b is summed; F[b,a] = conj(F[a,b]) is mirrored so that F is EXACTLY Hermitian in
// (a,b) like NumPy's result (the reference asserts infidelity matrices equal their own
// conjugate transpose bit for bit, tests/test_precision.py:549).
__global__ void ff_fidelity_kernel(const cplx* __restrict__ R, int A, int N, int W,
                                   cplx* __restrict__ F) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= W) return;
    // operator pairs on grid.y, strided: A*A may exceed the 65535 blocks a grid axis holds
    // (pulse-correlation filter functions of long sequences: A = n_pulses * n_nops)
    for (long long pair = blockIdx.y; pair < static_cast<long long>(A)*A; pair += gridDim.y) {
        const int a = static_cast<int>(pair / A), b = static_cast<int>(pair % A);
        if (a > b) continue;
        const cplx* ra = R + static_cast<size_t>(a)*N*W + w;
        const cplx* rb = R + static_cast<size_t>(b)*N*W + w;
        cplx acc = {0.0, 0.0};
        for (int k = 0; k < N; ++k)
            cmac_conj(acc, ra[static_cast<size_t>(k)*W], rb[static_cast<size_t>(k)*W]);
        if (a == b) acc.im = 0.0;   // sum_k |R|^2: the imaginary parts cancel term by term
        F[(static_cast<size_t>(a)*A + b)*W + w] = acc;
        if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {acc.re, -acc.im};
    }
}

// The same for many noise operators: one thread per frequency and per TB x TB block of operator
// pairs (only blocks on or above the diagonal), so that a row of R is re-read A/TB times instead
// of A times (A = 18 at config 5: 23 GB of L2 traffic in the pairwise kernel).  Identical
// summation order per (a, b), hence bit-identical results.
template <int TB>
__global__ __launch_bounds__(128) void ff_fidelity_blocked_kernel(const cplx* __restrict__ R, int A,
                                                                  int N, int W,
                                                                  cplx* __restrict__ F) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    const int nb = (A + TB - 1)/TB;
    if (w >= W) return;
    for (long long tile = blockIdx.y; tile < static_cast<long long>(nb)*nb; tile += gridDim.y) {
    const int ba = static_cast<int>(tile / nb), bb = static_cast<int>(tile % nb);
    if (ba > bb) continue;
    const int a0 = ba*TB, b0 = bb*TB;
    cplx acc[TB][TB];
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = {0.0, 0.0};
    for (int k = 0; k < N; ++k) {
        cplx ra[TB], rb[TB];
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int a = min(a0 + i, A - 1), b = min(b0 + i, A - 1);
            ra[i] = R[(static_cast<size_t>(a)*N + k)*W + w];
            rb[i] = R[(static_cast<size_t>(b)*N + k)*W + w];
        }
#pragma unroll
        for (int i = 0; i < TB; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) cmac_conj(acc[i][j], ra[i], rb[j]);
    }
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int a = a0 + i, b = b0 + j;
            if (a >= A || b >= A || a > b) continue;
            cplx v = acc[i][j];
            if (a == b) v.im = 0.0;
            F[(static_cast<size_t>(a)*A + b)*W + w] = v;
            if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {v.re, -v.im};
        }
    }
}

// The same again with R read ONCE.  In ff_fidelity_blocked_kernel every TB x TB tile of operator
// pairs re-reads its rows of R: 54 row sets for 18 operators instead of 18, 3.6 GB through L2 /
// Infinity Cache at config 5 (0.39 ms).  Here a block owns 64 frequencies and ALL operators: the rows
// of R for a few k at a time are staged in LDS by everybody (1 KiB runs; the next chunk's loads fly
// while this one is worked on, two LDS buffers), and wavefront t forms the pairs of tile t (of the
// tiles on and above the diagonal) from LDS.  Same summation order per pair: bit-identical F.
template <int TB, int NT>
__global__ __launch_bounds__(NT*64) void ff_fidelity_lds_kernel(const cplx* __restrict__ R, int A, int N, int W,
                                                             int kc, cplx* __restrict__ F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ffl_raw[];
    cplx* buf = reinterpret_cast<cplx*>(ffl_raw);                  // [2][kc][A][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int nb = (A + TB - 1)/TB;
    int ba = 0, rem = wave;                                       // wave -> tile (ba, bb), ba <= bb
    while (rem >= nb - ba) {
        rem -= nb - ba;
        ++ba;
    }
    const int bb = ba + rem;
    const int a0 = ba*TB, b0 = bb*TB;
    const int w0 = blockIdx.x*64;
    const int w = w0 + lane;
    const int wc = min(w, W - 1);
    const int rows = kc*A;                        // rows of 64 frequencies per chunk
    constexpr int kMaxRows = 12;                  // per wavefront and chunk (host: rows <= kMaxRows*nwaves)
    // row r of this wavefront's share of a chunk: operator and k offset (the same for every chunk)
    const cplx* src[kMaxRows];
    int dst[kMaxRows];
#pragma unroll
    for (int r = 0; r < kMaxRows; ++r) {
        // (rows past the end repeat the last one: the same values fetched and parked again)
        const int row = min(wave + r*nwaves, rows - 1);
        src[r] = R + (static_cast<size_t>(row % A)*N + row / A)*W + wc;
        dst[r] = row*64 + lane;
    }
    cplx acc[TB][TB];
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = {0.0, 0.0};
    // chunk 0 straight into buffer 0
#pragma unroll
    for (int r = 0; r < kMaxRows; ++r) buf[dst[r]] = src[r][0];
    __syncthreads();
    int b = 0;
    for (int k0 = 0; k0 < N; k0 += kc) {
        // The next chunk's rows are requested before this chunk is worked on and parked after it --
        // unconditionally, as plain locals: under `if (more chunks)` / in a lambda the staged rows
        // lived in scratch memory and every load was waited for where it was requested (0.93 ms
        // instead of 0.39).  The trip after the last chunk re-reads clamped rows and parks them
        // where nobody looks.
        cplx staged[kMaxRows];
        const size_t knext = static_cast<size_t>(min(k0 + kc, N - kc < 0 ? 0 : N - kc))*W;
#pragma unroll
        for (int r = 0; r < kMaxRows; ++r) staged[r] = src[r][knext];
        const int kn = min(kc, N - k0);
        for (int kk = 0; kk < kn; ++kk) {
            const cplx* rowp = buf + (static_cast<size_t>(b)*rows + kk*A)*64 + lane;
            cplx ra[TB], rb[TB];
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                ra[i] = rowp[min(a0 + i, A - 1)*64];
                rb[i] = rowp[min(b0 + i, A - 1)*64];
            }
#pragma unroll
            for (int i = 0; i < TB; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) cmac_conj(acc[i][j], ra[i], rb[j]);
        }
#pragma unroll
        for (int r = 0; r < kMaxRows; ++r) buf[static_cast<size_t>(b ^ 1)*rows*64 + dst[r]] = staged[r];
        __syncthreads();
        b ^= 1;
    }
    if (w >= W) return;
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int a = a0 + i, bq = b0 + j;
            if (a >= A || bq >= A || a > bq) continue;
            cplx v = acc[i][j];
            if (a == bq) v.im = 0.0;
            F[(static_cast<size_t>(a)*A + bq)*W + w] = v;
            if (a != bq) F[(static_cast<size_t>(bq)*A + a)*W + w] = {v.re, -v.im};
        }
}

// F[a,b,w] = scale * sum_kl conj(R[a,k,w]) M[k,l] R[b,l,w]   ('ako,blo,kl->abo', the filter
// function of a basis that is not traceless, numeric.py:2295-2305; M from the four-element traces)
__global__ __launch_bounds__(128) void ff_weighted_kernel(const cplx* __restrict__ R, int A, int N,
                                                          int W, const cplx* __restrict__ M,
                                                          double scale, cplx* __restrict__ F) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= W) return;
    for (long long pair = blockIdx.y; pair < static_cast<long long>(A)*A; pair += gridDim.y) {
        const int a = static_cast<int>(pair / A), b = static_cast<int>(pair % A);
        const cplx* ra = R + static_cast<size_t>(a)*N*W + w;
        const cplx* rb = R + static_cast<size_t>(b)*N*W + w;
        cplx acc = {0.0, 0.0};
        for (int k = 0; k < N; ++k) {
            cplx u = {0.0, 0.0};                       // sum_l M[k,l] R[b,l,w]
            for (int l = 0; l < N; ++l) cmac(u, M[k*N + l], rb[static_cast<size_t>(l)*W]);
            cmac_conj(acc, ra[static_cast<size_t>(k)*W], u);
        }
        F[(static_cast<size_t>(a)*A + b)*W + w] = {scale*acc.re, scale*acc.im};
    }
}

// F[a,b,k,l,w] = conj(R[a,k,w]) R[b,l,w]        ('ako,blo->abklo', numeric.py:1465).
// Plain multiply / subtract (no FMA contraction across the two products) so that
// F[b,a,l,k] == conj(F[a,b,k,l]) holds exactly, as it does for NumPy's complex multiply.
__global__ void ff_generalized_kernel(const cplx* __restrict__ R, int A, int N, int W,
                                      cplx* __restrict__ F) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    if (w >= W) return;
    // basis pairs on grid.y and operator pairs on grid.z, both strided (d = 16: N*N = 65536 > 65535)
    for (long long ab = blockIdx.z; ab < static_cast<long long>(A)*A; ab += gridDim.z) {
        const int a = static_cast<int>(ab / A), b = static_cast<int>(ab % A);
        for (long long kl = blockIdx.y; kl < static_cast<long long>(N)*N; kl += gridDim.y) {
            const int k = static_cast<int>(kl / N), l = static_cast<int>(kl % N);
            const cplx x = R[(static_cast<size_t>(a)*N + k)*W + w];
            const cplx y = R[(static_cast<size_t>(b)*N + l)*W + w];
            const double rr = x.re*y.re;
            const double ii = x.im*y.im;
            const double ri = x.re*y.im;
            const double ir = x.im*y.re;
            cplx v;
            v.re = rr + ii;
            v.im = ri - ir;
            F[((((static_cast<size_t>(a)*A + b)*N + k)*N + l))*W + w] = v;
        }
    }
}

// ---- infidelity ------------------------------------------------------------------------------
// integrand_p[w] = Re(F[ia, ib, w] S_p[w]); partial[p, blk] = sum over the block's omega tile of
// (f[w+1] + f[w]) (omega[w+1] - omega[w])   (util.integrate, util.py:903-906, before the /2).
// One block per output element p: the whole omega axis is reduced over 1024 slots in fixed order
// (slot v sums intervals v, v+1024, ...; then a tree over the slots): deterministic, one launch.
// The block has 256 threads, four slots each (bit-identical to one thread per slot): a
// 1024-thread block cannot be placed on a CU that an accumulate block of another pass occupies
// (16 + 16 waves fit, their registers do not), and a workgroup that cannot be placed anywhere
// stalls the dispatcher for every queue -- the scan and prologue of the next pass then waited for
// the accumulate kernel to retire as well (profiles/r02_q_*).
// shard_width > 0: F is the raw all-gather buffer (n_shards, A, A, shard_width) of omega blocks
// (one block per rank) instead of (A, A, W): global frequency w lives in shard w / shard_width.
constexpr int kInfidSlots = 1024, kInfidThreads = 256;
constexpr int kInfidPer = kInfidSlots/kInfidThreads;       // slots per thread: v = thread + 256 j

// One batch of 1024 intervals [base, base + 1024): thread t owns the frequencies w = base + t + 256 j, forms the
// integrand there ONCE and takes integrand and frequency at w + 1 from its neighbour lane (the last lane of a
// wavefront fetches them itself).  All of a thread's loads of a batch are in flight together: the loop this
// replaces made one trip to memory per interval -- sixteen dependent round trips per thread at config 2, most of
// the kernel's 5 us above its launch floor (profiles/r05_s_*).  Same terms, same order per slot: same bits.
// (J0, JN: the slots taken in one go -- all four where registers are free, two and two in the kernel that has to fit
// beside another pass's accumulate blocks: 56 registers)
// acc: the slots' running sums -- registers, or (red != nullptr) the slots of the LDS array the final tree reduces,
// where the registers are needed for the loads in flight.
template <int J0, int JN, typename LoadF, typename LoadS>
__device__ __forceinline__ void infid_batch(int base, int W, const double* __restrict__ omega, LoadF Fat, LoadS Sat,
                                            double (&acc)[kInfidPer], double* red = nullptr) {
    const int lane = threadIdx.x & 63;
    cplx f[kInfidPer], sp[kInfidPer];
    double om[kInfidPer];
#pragma unroll
    for (int j = J0; j < J0 + JN; ++j) {
        const int w = base + static_cast<int>(threadIdx.x) + kInfidThreads*j;
        const int wc = w < W ? w : W - 1;
        f[j] = Fat(wc);
        sp[j] = Sat(wc);
        om[j] = omega[wc];
    }
#pragma unroll
    for (int j = J0; j < J0 + JN; ++j) {
        const int w = base + static_cast<int>(threadIdx.x) + kInfidThreads*j;
        const double i0 = f[j].re*sp[j].re - f[j].im*sp[j].im;
        double i1 = __shfl_down(i0, 1, 64), o1 = __shfl_down(om[j], 1, 64);
        if (lane == 63 && w + 1 < W) {
            const cplx f1 = Fat(w + 1), s1 = Sat(w + 1);
            i1 = f1.re*s1.re - f1.im*s1.im;
            o1 = omega[w + 1];
        }
        if (w < W - 1) {
            if (red != nullptr) red[threadIdx.x + kInfidThreads*j] += (i1 + i0)*(o1 - om[j]);
            else acc[j] += (i1 + i0)*(o1 - om[j]);
        }
    }
}

// The tree over the 1024 slots, pairing (v, v + s) for s = 512, 256, .., 1 as a loop over LDS with a barrier per level
// would: the levels 512 and 256 pair slots of ONE thread (registers), 128 and 64 cross wavefronts (two barriers), the
// last six stay inside wavefront 0 (lane shifts).  Same association, same bits, eight barriers fewer.
template <bool FROM_REGISTERS = true>
__device__ __forceinline__ void infid_finish(double* red, const double (&acc)[kInfidPer], int d, double* out) {
    static_assert(kInfidPer == 4 && kInfidThreads == 256, "the tree below is written for 4 slots per thread, 4 wavefronts");
    double sl[kInfidPer];
#pragma unroll
    for (int j = 0; j < kInfidPer; ++j) sl[j] = FROM_REGISTERS ? acc[j] : red[threadIdx.x + kInfidThreads*j];
    double x = (sl[0] + sl[2]) + (sl[1] + sl[3]);
    __syncthreads();                                   // (every thread has read its own slots)
    red[threadIdx.x] = x;
    __syncthreads();
    if (threadIdx.x < 128) red[threadIdx.x] = x = x + red[threadIdx.x + 128];
    __syncthreads();
    if (threadIdx.x < 64) {
        x = x + red[threadIdx.x + 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x = x + __shfl_down(x, off, 64);
        if (threadIdx.x == 0) *out = (x/2.0)/(2.0*3.141592653589793*d);
    }
}

template <bool SHARDED>
__global__ __launch_bounds__(kInfidThreads) void infid_kernel(const cplx* __restrict__ F, int A, int W,
                                                              const cplx* __restrict__ S, int s_ndim,
                                                              const double* __restrict__ omega,
                                                              const int32_t* __restrict__ idx, int n_idx,
                                                              int d, int shard_width,
                                                              double* __restrict__ infid) {
    __shared__ double red[kInfidSlots];
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int p = blockIdx.x;  // output element
    int ia, ib;
    const cplx* Sp;
    if (s_ndim == 3) {
        ia = idx[p / n_idx];
        ib = idx[p % n_idx];
        Sp = S + static_cast<size_t>(p)*W;
    } else {
        ia = ib = idx[p];
        Sp = S + (s_ndim == 2 ? static_cast<size_t>(p)*W : 0);
    }
    const cplx* Fp = F + (static_cast<size_t>(ia)*A + ib)*(SHARDED ? shard_width : W);
    auto Fat = [&](int w) -> cplx {
        if (SHARDED) {
            const int r = w / shard_width, wl = w - r*shard_width;
            return Fp[static_cast<size_t>(r)*A*A*shard_width + wl];
        }
        return Fp[w];
    };
    auto Sat = [&](int w) -> cplx { return Sp[w]; };
    // this kernel runs beside another pass's accumulate blocks: 56 registers (tests/test_kernel_resources.py) -- the
    // running sums live in the LDS slots themselves (each thread its own four), the registers hold the loads in flight
    double acc[kInfidPer];
#pragma unroll
    for (int j = 0; j < kInfidPer; ++j) red[threadIdx.x + kInfidThreads*j] = 0.0;
    for (int base = 0; base < W - 1; base += kInfidSlots) {
        infid_batch<0, kInfidPer>(base, W, omega, Fat, Sat, acc, red);
        asm volatile("" ::: "memory");      // (keeps the next batch's loads behind this batch's sums)
    }
    infid_finish<false>(red, acc, d, infid + p);
}

// The same integral with the spectrum in mapped pinned HOST memory (the resident API path hands it over
// without a copy): every read of it is a trip over PCIe, so a stage of 4096 spectrum values goes through LDS
// first, every thread's 17 reads in flight together (27 us -> 14 us with the interval-by-interval loop,
// profiles/r05_r_*); then the same batches as above: bit-identical results.
constexpr int kInfidStage = 4096;
__global__ __launch_bounds__(kInfidThreads) void infid_host_spectrum_kernel(
    const cplx* __restrict__ F, int A, int W, const cplx* __restrict__ S, int s_ndim,
    const double* __restrict__ omega, const int32_t* __restrict__ idx, int n_idx, int d,
    double* __restrict__ infid) {
    __shared__ double red[kInfidSlots];
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];
    cplx* sl = reinterpret_cast<cplx*>(stage_raw);       // [kInfidStage + 1]
    const int p = blockIdx.x;
    int ia, ib;
    const cplx* Sp;
    if (s_ndim == 3) {
        ia = idx[p / n_idx];
        ib = idx[p % n_idx];
        Sp = S + static_cast<size_t>(p)*W;
    } else {
        ia = ib = idx[p];
        Sp = S + (s_ndim == 2 ? static_cast<size_t>(p)*W : 0);
    }
    const cplx* Fp = F + (static_cast<size_t>(ia)*A + ib)*W;
    constexpr int kLoads = kInfidStage/kInfidThreads + 1;
    double acc[kInfidPer];
#pragma unroll
    for (int j = 0; j < kInfidPer; ++j) acc[j] = 0.0;
    for (int s0 = 0; s0 < W - 1; s0 += kInfidStage) {
        const int n = min(kInfidStage + 1, W - s0);
        cplx v[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int i = threadIdx.x + kInfidThreads*k;
            v[k] = i < n ? Sp[s0 + i] : cplx{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < kLoads; ++k) {
            const int i = threadIdx.x + kInfidThreads*k;
            if (i < n) sl[i] = v[k];
        }
        __syncthreads();
        auto Fat = [&](int w) -> cplx { return Fp[w]; };
        auto Sat = [&](int w) -> cplx { return sl[w - s0]; };
        const int bend = min(W - 1, s0 + kInfidStage);
        for (int base = s0; base < bend; base += kInfidSlots) infid_batch<0, kInfidPer>(base, W, omega, Fat, Sat, acc);
        __syncthreads();
    }
    infid_finish(red, acc, d, infid + p);
}

}  // namespace

hipError_t launch_reduce_chunks(const cplx* Ypart, int chunks, size_t slab, cplx* Bt,
                                hipStream_t stream) {
    const int block = 256;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(static_cast<unsigned>((slab + block - 1)/block)),
                       dim3(block), 0, stream, Ypart, chunks, slab, Bt);
    return hipGetLastError();
}

size_t expand_workspace_bytes(int N, int d) {
    const size_t dd = static_cast<size_t>(d)*d;
    return align_up(sizeof(int)*N) + align_up(sizeof(int)*N*dd) + align_up(sizeof(cplx)*N*dd);
}

namespace {
struct CompactWs {
    int* nnz;
    int* rows;
    cplx* vals;
};
CompactWs slice_compact_ws(void* ws, int N, int d) {
    const size_t dd = static_cast<size_t>(d)*d;
    unsigned char* p = static_cast<unsigned char*>(ws);
    CompactWs out;
    out.nnz = reinterpret_cast<int*>(p);
    p += align_up(sizeof(int)*N);
    out.rows = reinterpret_cast<int*>(p);
    p += align_up(sizeof(int)*N*dd);
    out.vals = reinterpret_cast<cplx*>(p);
    return out;
}
}  // namespace

hipError_t launch_reduce_and_compact(const cplx* Ypart, int chunks, size_t slab, cplx* Bt,
                                     const cplx* basis, int N, int d, void* ws, hipStream_t stream) {
    const CompactWs c = slice_compact_ws(ws, N, d);
    const int block = 256;
    const int nred = static_cast<int>((slab + block - 1)/block);
    hipLaunchKernelGGL(reduce_compact_kernel, dim3(nred + N), dim3(block), 0, stream, Ypart, chunks,
                       slab, Bt, nred, basis, d, c.nnz, c.rows, c.vals);
    return hipGetLastError();
}

void expand_workspace_slices(void* ws, int N, int d, int** nnz, int** rows, cplx** vals) {
    const CompactWs cw = slice_compact_ws(ws, N, d);
    *nnz = cw.nnz;
    *rows = cw.rows;
    *vals = cw.vals;
}

// Expansion straight from the segment-chunk partials: R[a,k,w] = sum_nz C_k[j,i] (sum_z
// Ypart[z,a,i,j,w]) with the chunk sum taken on the fly in chunk order (the same value
// reduce_chunks produces), so that the reduction launch and the round trip of the summed Y through
// HBM disappear when only the control matrix is wanted.  Each entry of Y is summed once per basis
// element that touches it (d times for a Pauli basis): more L2 reads, one launch less.
__global__ __launch_bounds__(64) void expand_chunks_kernel(const cplx* __restrict__ Ypart, int chunks,
                                                           size_t slab, const int* __restrict__ nnz,
                                                           const int* __restrict__ rows,
                                                           const cplx* __restrict__ vals, int N,
                                                           int dd, int W, cplx* __restrict__ R) {
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y, k = blockIdx.z;
    if (w >= W) return;
    const cplx* b = Ypart + static_cast<size_t>(a)*dd*W + w;
    const int n = nnz[k];
    const int* rk = rows + static_cast<size_t>(k)*dd;
    const cplx* vk = vals + static_cast<size_t>(k)*dd;
    auto summed = [&](int e) {
        cplx acc = b[static_cast<size_t>(e)*W];
        for (int z = 1; z < chunks; ++z) {
            const cplx v = b[z*slab + static_cast<size_t>(e)*W];
            acc.re += v.re;
            acc.im += v.im;
        }
        return acc;
    };
    cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
    int q = 0;
    for (; q + 1 < n; q += 2) {
        cmac(acc0, vk[q], summed(rk[q]));
        cmac(acc1, vk[q + 1], summed(rk[q + 1]));
    }
    if (q < n) cmac(acc0, vk[q], summed(rk[q]));
    R[(static_cast<size_t>(a)*N + k)*W + w] = {acc0.re + acc1.re, acc0.im + acc1.im};
}

// The same expansion followed, in the same launch, by the fidelity filter function of the block's
// 16 frequencies: F[a,b,w] = sum_k conj(R[a,k,w]) R[b,k,w] (a <= b, mirrored; the summation order
// of ff_fidelity_kernel).  Block = 16 frequencies x `kt` basis-element lanes; R passes through LDS.
__global__ __launch_bounds__(256) void expand_ff_kernel(const cplx* __restrict__ Ypart, int chunks,
                                                         size_t slab, const int* __restrict__ nnz,
                                                         const int* __restrict__ rows,
                                                         const cplx* __restrict__ vals, int N, int dd,
                                                         int W, int A, int kt, cplx* __restrict__ R,
                                                         cplx* __restrict__ F) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* rl = reinterpret_cast<cplx*>(lds_raw);       // [A][N][16]
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    cplx* ys = rl + static_cast<size_t>(A)*N*16;        // [A][dd][16]: chunk partials summed
    // thread = (frequency wl, basis-element lane kl, noise operator lane al)
    const int wl = threadIdx.x & 15;
    const int kl = (threadIdx.x >> 4) % kt, al = (threadIdx.x >> 4) / kt;
    const int at = (blockDim.x >> 4) / kt;
    const int w = blockIdx.x*16 + wl;
    const int wc = w < W ? w : W - 1;
    // every partial sum is read from memory once (in chunk order) and expanded from LDS: a matrix
    // element takes part in several basis elements (four for the Pauli basis of d = 4)
    for (int idx = threadIdx.x >> 4; idx < A*dd; idx += blockDim.x >> 4) {
        const cplx* b = Ypart + static_cast<size_t>(idx)*W + wc;
        cplx acc = b[0];
        for (int z = 1; z < chunks; ++z) {
            const cplx v = b[z*slab];
            acc.re += v.re;
            acc.im += v.im;
        }
        ys[static_cast<size_t>(idx)*16 + wl] = acc;
    }
    __syncthreads();
    for (int a = al; a < A; a += at) {
        const cplx* ya = ys + static_cast<size_t>(a)*dd*16 + wl;
        for (int k = kl; k < N; k += kt) {
            const int n = nnz[k];
            const int* rk = rows + static_cast<size_t>(k)*dd;
            const cplx* vk = vals + static_cast<size_t>(k)*dd;
            auto summed = [&](int e) { return ya[e*16]; };
            cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
            int q = 0;
            for (; q + 1 < n; q += 2) {
                cmac(acc0, vk[q], summed(rk[q]));
                cmac(acc1, vk[q + 1], summed(rk[q + 1]));
            }
            if (q < n) cmac(acc0, vk[q], summed(rk[q]));
            const cplx r = {acc0.re + acc1.re, acc0.im + acc1.im};
            rl[(static_cast<size_t>(a)*N + k)*16 + wl] = r;
            if (w < W) R[(static_cast<size_t>(a)*N + k)*W + w] = r;
        }
    }
    __syncthreads();
    if (w >= W) return;
    const int npairs = A*(A + 1)/2;
    const int lanes = blockDim.x >> 4;
    for (int p = threadIdx.x >> 4; p < npairs; p += lanes) {
        int a = 0, rem = p;                  // p -> (a, b), a <= b, row-major over the upper triangle
        while (rem >= A - a) {
            rem -= A - a;
            ++a;
        }
        const int b = a + rem;
        const cplx* ra = rl + static_cast<size_t>(a)*N*16 + wl;
        const cplx* rb = rl + static_cast<size_t>(b)*N*16 + wl;
        cplx acc = {0.0, 0.0};
        for (int k = 0; k < N; ++k) cmac_conj(acc, ra[k*16], rb[k*16]);
        if (a == b) acc.im = 0.0;
        F[(static_cast<size_t>(a)*A + b)*W + w] = acc;
        if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {acc.re, -acc.im};
    }
}

bool expand_ff_supported(int A, int N) {
    // R tile [A][N][16] + summed partials [A][d^2][16], d^2 <= N for a complete basis; N bounds both
    return static_cast<size_t>(A)*2*N*16*sizeof(cplx) <= 128*1024;
}

hipError_t launch_expand_ff(const cplx* Ypart, int chunks, size_t slab, int A, int N, int d, int W,
                            cplx* R, cplx* F, void* ws, hipStream_t stream) {
    if (!expand_ff_supported(A, N)) return hipErrorInvalidValue;
    const CompactWs cw = slice_compact_ws(ws, N, d);
    const int kt = N >= 16 ? 16 : (N >= 8 ? 8 : 4);
    // block = 16 x kt x at <= 256 threads (one wave per SIMD): a 768-thread block of this kernel does
    // not fit beside an accumulate block of another pass (3 + 4 waves per SIMD would, their registers
    // do not) and then waits for that whole kernel -- with the front of the next-but-one pass queued
    // behind it on the same stream
    const int at = std::max(1, std::min(A, 16/kt));
    const size_t lds = static_cast<size_t>(A)*(N + d*d)*16*sizeof(cplx);
    if (lds > 160*1024) return hipErrorInvalidValue;
    if (lds > 48*1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(expand_ff_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(expand_ff_kernel, dim3((W + 15)/16), dim3(16*kt*at), lds, stream, Ypart,
                       chunks, slab, cw.nnz, cw.rows, cw.vals, N, d*d, W, A, kt, R, F);
    return hipGetLastError();
}

// Large d: one block per (noise operator, 16 frequencies) stages Y[a,:,:,w-tile] in LDS once
// (summing the segment chunks on the way) and expands it for every basis element from there.  Y is
// read from HBM exactly once whatever the sparsity of the basis (a Pauli element of d = 16 has 16
// non-zeros: the per-element kernel re-read Y 16 times through L2, 2.7 ms at config-5 size).
__global__ __launch_bounds__(256) void expand_lds_kernel(const cplx* __restrict__ Ypart, int chunks,
                                                         size_t slab, const int* __restrict__ nnz,
                                                         const int* __restrict__ rows,
                                                         const cplx* __restrict__ vals, int N, int dd,
                                                         int W, cplx* __restrict__ R, int want_sparse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (want_sparse >= 0 && basis_is_sparse(nnz, N, dd) != (want_sparse != 0)) return;   // (block uniform)
    cplx* yl = reinterpret_cast<cplx*>(lds_raw);       // [dd][16]
    const int wl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int a = blockIdx.y;
    const int w = blockIdx.x*16 + wl;
    const int wc = w < W ? w : W - 1;
    const cplx* b = Ypart + static_cast<size_t>(a)*dd*W + wc;
    for (int e = kl; e < dd; e += 16) {
        cplx acc = b[static_cast<size_t>(e)*W];
        for (int z = 1; z < chunks; ++z) {
            const cplx v = b[z*slab + static_cast<size_t>(e)*W];
            acc.re += v.re;
            acc.im += v.im;
        }
        yl[e*16 + wl] = acc;
    }
    __syncthreads();
    if (w >= W) return;
    for (int k = kl; k < N; k += 16) {
        const int n = nnz[k];
        const int* rk = rows + static_cast<size_t>(k)*dd;
        const cplx* vk = vals + static_cast<size_t>(k)*dd;
        cplx acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
        int q = 0;
        for (; q + 1 < n; q += 2) {
            cmac(acc0, vk[q], yl[rk[q]*16 + wl]);
            cmac(acc1, vk[q + 1], yl[rk[q + 1]*16 + wl]);
        }
        if (q < n) cmac(acc0, vk[q], yl[rk[q]*16 + wl]);
        R[(static_cast<size_t>(a)*N + k)*W + w] = {acc0.re + acc1.re, acc0.im + acc1.im};
    }
}

hipError_t launch_expand_lds(const cplx* Ypart, int chunks, size_t slab, int A, int N, int d, int W,
                             cplx* R, void* ws, hipStream_t stream, int want_sparse = -1) {
    if (A > 65535) return hipErrorInvalidValue;
    const CompactWs cw = slice_compact_ws(ws, N, d);
    const size_t lds = static_cast<size_t>(d)*d*16*sizeof(cplx);
    if (lds > 48*1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(expand_lds_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL(expand_lds_kernel, dim3((W + 15)/16, A), dim3(256), lds, stream, Ypart, chunks,
                       slab, cw.nnz, cw.rows, cw.vals, N, d*d, W, R, want_sparse);
    return hipGetLastError();
}


hipError_t launch_expand_chunks(const cplx* Ypart, int chunks, size_t slab, int A, int N, int d,
                                int W, cplx* R, void* ws, hipStream_t stream) {
    if (A > 65535 || N > 65535) return hipErrorInvalidValue;
    if (d >= 8 && d <= kMaxD)      // (d > 16: the tile would not fit LDS)
        return launch_expand_lds(Ypart, chunks, slab, A, N, d, W, R, ws, stream);
    const CompactWs cw = slice_compact_ws(ws, N, d);
    hipLaunchKernelGGL(expand_chunks_kernel, dim3((W + 63)/64, A, N), dim3(64), 0, stream, Ypart,
                       chunks, slab, cw.nnz, cw.rows, cw.vals, N, d*d, W, R);
    return hipGetLastError();
}

hipError_t launch_expand(const cplx* Bt, const cplx* basis, int A2, int N, int d, int W, cplx* R,
                         void* ws, bool compacted, hipStream_t stream) {
    if (A2 > 65535) return hipErrorInvalidValue;
    const size_t dd = static_cast<size_t>(d)*d;
    const CompactWs cw = slice_compact_ws(ws, N, d);
    int* nnz = cw.nnz;
    int* rows = cw.rows;
    cplx* vals = cw.vals;
    if (!compacted)
        hipLaunchKernelGGL(basis_compact_kernel, dim3(N), dim3(64), 0, stream, basis, d, nnz, rows, vals);
    if (d >= 8 && d <= kMaxD) {
        // both forms, each deciding on the device whether the basis is its kind (see basis_is_sparse)
        hipError_t err = launch_expand_lds(Bt, 1, 0, A2, N, d, W, R, ws, stream, 0);
        if (err != hipSuccess) return err;
        if (N >= 64 && static_cast<long>((W + 63)/64)*A2 >= 1024) {
            if (N <= 256 && dd <= 0x3fff)
                hipLaunchKernelGGL(expand_sparse_groups_kernel, dim3((W + 63)/64, A2), dim3(256), 0, stream, Bt,
                                   nnz, rows, vals, N, static_cast<int>(dd), W, R, 1);
            else
                hipLaunchKernelGGL(expand_sparse_quarters_kernel, dim3((W + 63)/64, A2), dim3(256), 0, stream, Bt,
                                   nnz, rows, vals, N, static_cast<int>(dd), W, R, 1);
            return hipGetLastError();
        }
        constexpr int KT = 8;
        hipLaunchKernelGGL(expand_sparse_kernel<KT>, dim3((W + 63)/64, A2, (N + KT - 1)/KT), dim3(64),
                           0, stream, Bt, nnz, rows, vals, N, static_cast<int>(dd), W, R, 1);
        return hipGetLastError();
    }
    // few basis elements per thread when the grid would otherwise be small
    const long blocks1 = static_cast<long>((W + 63)/64)*A2;
    if (blocks1*N <= 16384 || N <= 16) {
        hipLaunchKernelGGL(expand_sparse_kernel<1>, dim3((W + 63)/64, A2, N), dim3(64), 0, stream, Bt,
                           nnz, rows, vals, N, static_cast<int>(dd), W, R, -1);
    } else {
        constexpr int KT = 8;
        hipLaunchKernelGGL(expand_sparse_kernel<KT>, dim3((W + 63)/64, A2, (N + KT - 1)/KT), dim3(64),
                           0, stream, Bt, nnz, rows, vals, N, static_cast<int>(dd), W, R, -1);
    }
    return hipGetLastError();
}

hipError_t launch_transpose_noise_ops(const cplx* Bt, int A, int d, int W, cplx* out,
                                      hipStream_t stream) {
    const int rows = A*d*d;
    hipLaunchKernelGGL(transpose_kernel, dim3((W + 63)/64, (rows + 63)/64), dim3(256), 0, stream, Bt,
                       rows, W, out);
    return hipGetLastError();
}

// blocks on grid.y / grid.z: at most 65535; the kernels stride over what does not fit
static unsigned grid_axis(long long count) { return static_cast<unsigned>(count < 65535 ? count : 65535); }

hipError_t launch_filter_function(const cplx* R, int A, int N, int W, int which, cplx* F,
                                  hipStream_t stream) {
    const int block = 128;
    if (which == 0 && A > 4) {
        constexpr int TB = 6;
        const int nb = (A + TB - 1)/TB;
        // many frequencies, up to 24 operators: R read once through LDS (one wavefront per pair tile)
        const int ntiles = nb*(nb + 1)/2;
        int kc = std::min(8, (72*1024)/(A*1024));
        while (kc > 1 && kc*A > 12*ntiles) --kc;            // at most 12 staged rows per wavefront and chunk
        if (ntiles <= 6 && kc >= 1 && kc*A <= 12*ntiles && (W + 63)/64 >= 128) {
            const size_t lds = sizeof(cplx)*2*static_cast<size_t>(kc)*A*64;
            auto go = [&](auto kern) -> hipError_t {
                hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(lds));
                if (err != hipSuccess) return err;
                hipLaunchKernelGGL(kern, dim3((W + 63)/64), dim3(64*ntiles), lds, stream, R, A, N, W, kc, F);
                return hipGetLastError();
            };
            // (A <= 6: one tile; <= 12: three; <= 18: six wavefronts = pair tiles per block)
            return ntiles <= 3 ? go(ff_fidelity_lds_kernel<TB, 3>) : go(ff_fidelity_lds_kernel<TB, 6>);
        }
        hipLaunchKernelGGL(ff_fidelity_blocked_kernel<TB>,
                           dim3((W + block - 1)/block, grid_axis(static_cast<long long>(nb)*nb)),
                           dim3(block), 0, stream, R, A, N, W, F);
    } else if (which == 0) {
        hipLaunchKernelGGL(ff_fidelity_kernel,
                           dim3((W + block - 1)/block, grid_axis(static_cast<long long>(A)*A)),
                           dim3(block), 0, stream, R, A, N, W, F);
    } else {
        hipLaunchKernelGGL(ff_generalized_kernel,
                           dim3((W + block - 1)/block, grid_axis(static_cast<long long>(N)*N),
                                grid_axis(static_cast<long long>(A)*A)),
                           dim3(block), 0, stream, R, A, N, W, F);
    }
    return hipGetLastError();
}

hipError_t launch_filter_function_weighted(const cplx* R, int A, int N, int W, const cplx* M,
                                           double scale, cplx* F, hipStream_t stream) {
    const int block = 128;
    hipLaunchKernelGGL(ff_weighted_kernel,
                       dim3((W + block - 1)/block, grid_axis(static_cast<long long>(A)*A)),
                       dim3(block), 0, stream, R, A, N, W, M, scale, F);
    return hipGetLastError();
}

size_t infidelity_workspace_bytes(int W, int n_idx, int s_ndim) {
    (void)W;
    (void)n_idx;
    (void)s_ndim;
    return 256;  // none needed any more; kept non-zero so callers can always pass a buffer
}

hipError_t launch_infidelity(const cplx* F, int A, int W, const cplx* S, int s_ndim,
                             const double* omega, const int32_t* idx, int n_idx, int d,
                             int shard_width, double* infid, void* ws, hipStream_t stream,
                             bool spectrum_on_host) {
    (void)ws;
    const int nout = s_ndim == 3 ? n_idx*n_idx : n_idx;
    if (spectrum_on_host && (shard_width <= 0 || shard_width >= W)) {
        const int lds = static_cast<int>(sizeof(cplx))*(kInfidStage + 1);
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(infid_host_spectrum_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(infid_host_spectrum_kernel, dim3(nout), dim3(kInfidThreads), lds, stream, F, A, W, S,
                           s_ndim, omega, idx, n_idx, d, infid);
        return hipGetLastError();
    }
    // (one shard as wide as the grid -- the single-GPU bench -- IS the plain layout: no index division per load)
    if (shard_width > 0 && shard_width < W)
        hipLaunchKernelGGL(infid_kernel<true>, dim3(nout), dim3(kInfidThreads), 0, stream, F, A, W, S, s_ndim, omega,
                           idx, n_idx, d, shard_width, infid);
    else
        hipLaunchKernelGGL(infid_kernel<false>, dim3(nout), dim3(kInfidThreads), 0, stream, F, A, W, S, s_ndim, omega,
                           idx, n_idx, d, shard_width, infid);
    return hipGetLastError();
}

}  // namespace ffk
