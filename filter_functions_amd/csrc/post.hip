// post.hip -- everything downstream of the segment sum: chunk reduction, basis expansion
// R = tr(B~ C_k), filter function F = R^dag R and the trapezoid integrals of the infidelity.
// All of these are single-pass, HBM-streaming kernels with omega as the fastest (lane) axis.
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

// R[a,k,w] = sum_ij Bt[a,i,j,w] C_k[j,i]  -- Basis.expand (basis.py:650-698:
// tensordot(M, basis, axes=[(-2,-1),(-1,-2)])).  One lane per omega, KT basis elements per
// thread; the basis is wave-uniform (scalar loads).
template <int D, int KT>
__global__ __launch_bounds__(64) void expand_kernel(const cplx* __restrict__ Bt,
                                                    const cplx* __restrict__ basis, int N, int W,
                                                    cplx* __restrict__ R) {
    const int w = blockIdx.x*64 + threadIdx.x;
    const int a = blockIdx.y;
    const int k0 = blockIdx.z*KT;
    if (w >= W) return;
    const cplx* b = Bt + static_cast<size_t>(a)*D*D*W + w;
    cplx acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = {0.0, 0.0};
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const cplx v = b[static_cast<size_t>(i*D + j)*W];
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const int k = k0 + kk;
                if (k < N) cmac(acc[kk], basis[(static_cast<size_t>(k)*D + j)*D + i], v);
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int k = k0 + kk;
        if (k < N) R[(static_cast<size_t>(a)*N + k)*W + w] = acc[kk];
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
    const int a = blockIdx.y / A, b = blockIdx.y % A;
    if (w >= W || a > b) return;
    const cplx* ra = R + static_cast<size_t>(a)*N*W + w;
    const cplx* rb = R + static_cast<size_t>(b)*N*W + w;
    cplx acc = {0.0, 0.0};
    for (int k = 0; k < N; ++k) cmac_conj(acc, ra[static_cast<size_t>(k)*W], rb[static_cast<size_t>(k)*W]);
    if (a == b) acc.im = 0.0;   // sum_k |R|^2: the imaginary parts cancel term by term
    F[(static_cast<size_t>(a)*A + b)*W + w] = acc;
    if (a != b) F[(static_cast<size_t>(b)*A + a)*W + w] = {acc.re, -acc.im};
}

// F[a,b,k,l,w] = conj(R[a,k,w]) R[b,l,w]        ('ako,blo->abklo', numeric.py:1465).
// Plain multiply / subtract (no FMA contraction across the two products) so that
// F[b,a,l,k] == conj(F[a,b,k,l]) holds exactly, as it does for NumPy's complex multiply.
__global__ void ff_generalized_kernel(const cplx* __restrict__ R, int A, int N, int W,
                                      cplx* __restrict__ F) {
    const int w = blockIdx.x*blockDim.x + threadIdx.x;
    const int l = blockIdx.y % N, k = blockIdx.y / N;
    const int a = blockIdx.z / A, b = blockIdx.z % A;
    if (w >= W) return;
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

// ---- infidelity ------------------------------------------------------------------------------
// integrand_p[w] = Re(F[ia, ib, w] S_p[w]); partial[p, blk] = sum over the block's omega tile of
// (f[w+1] + f[w]) (omega[w+1] - omega[w])   (util.integrate, util.py:903-906, before the /2).
// One block per output element p: the whole omega axis is reduced by 1024 threads in fixed order
// (thread t sums intervals t, t+1024, ...; then a tree over the block): deterministic, one launch.
__global__ __launch_bounds__(1024) void infid_kernel(const cplx* __restrict__ F, int A, int W,
                                                     const cplx* __restrict__ S, int s_ndim,
                                                     const double* __restrict__ omega,
                                                     const int32_t* __restrict__ idx, int n_idx,
                                                     int d, double* __restrict__ infid) {
    __shared__ double red[1024];
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
    const cplx* Fp = F + (static_cast<size_t>(ia)*A + ib)*W;
    double acc = 0.0;
    for (int w = threadIdx.x; w < W - 1; w += 1024) {
        const cplx f0 = Fp[w], f1 = Fp[w + 1], s0 = Sp[w], s1 = Sp[w + 1];
        const double i0 = f0.re*s0.re - f0.im*s0.im;
        const double i1 = f1.re*s1.re - f1.im*s1.im;
        acc += (i1 + i0)*(omega[w + 1] - omega[w]);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) infid[p] = (red[0]/2.0)/(2.0*3.141592653589793*d);
}

template <int D>
hipError_t launch_expand_d(const cplx* Bt, const cplx* basis, int A2, int N, int W, cplx* R,
                           hipStream_t stream) {
    constexpr int KT = 4;
    const dim3 grid((W + 63)/64, A2, (N + KT - 1)/KT);
    hipLaunchKernelGGL((expand_kernel<D, KT>), grid, dim3(64), 0, stream, Bt, basis, N, W, R);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_reduce_chunks(const cplx* Ypart, int chunks, size_t slab, cplx* Bt,
                                hipStream_t stream) {
    const int block = 256;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(static_cast<unsigned>((slab + block - 1)/block)),
                       dim3(block), 0, stream, Ypart, chunks, slab, Bt);
    return hipGetLastError();
}

hipError_t launch_expand(const cplx* Bt, const cplx* basis, int A2, int N, int d, int W, cplx* R,
                         hipStream_t stream) {
    if (A2 > 65535) return hipErrorInvalidValue;
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_expand_d<D>(Bt, basis, A2, N, W, R, stream);
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t launch_transpose_noise_ops(const cplx* Bt, int A, int d, int W, cplx* out,
                                      hipStream_t stream) {
    const int rows = A*d*d;
    hipLaunchKernelGGL(transpose_kernel, dim3((W + 63)/64, (rows + 63)/64), dim3(256), 0, stream, Bt,
                       rows, W, out);
    return hipGetLastError();
}

hipError_t launch_filter_function(const cplx* R, int A, int N, int W, int which, cplx* F,
                                  hipStream_t stream) {
    const int block = 128;
    if (which == 0) {
        hipLaunchKernelGGL(ff_fidelity_kernel, dim3((W + block - 1)/block, A*A), dim3(block), 0,
                           stream, R, A, N, W, F);
    } else {
        if (N*N > 65535 || A*A > 65535) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ff_generalized_kernel, dim3((W + block - 1)/block, N*N, A*A), dim3(block),
                           0, stream, R, A, N, W, F);
    }
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
                             double* infid, void* ws, hipStream_t stream) {
    (void)ws;
    const int nout = s_ndim == 3 ? n_idx*n_idx : n_idx;
    hipLaunchKernelGGL(infid_kernel, dim3(nout), dim3(1024), 0, stream, F, A, W, S, s_ndim, omega,
                       idx, n_idx, d, infid);
    return hipGetLastError();
}

}  // namespace ffk
