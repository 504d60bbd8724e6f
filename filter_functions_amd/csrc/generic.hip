// generic.hip -- the path for Hilbert-space dimensions 17 <= d <= 64 (FFK_MAX_D), where the kernels
// of eigh.hip / scan.hip / prep.hip / ctrl*.hip (compile-time D <= 16: matrices in registers or
// wave-private LDS) do not reach.  Same mathematics, same operand layouts and outputs as those
// kernels, runtime d, one 256-thread workgroup per matrix-sized job with the d x d operands in LDS
// (d = 64: 2 x 64 KiB) and a register tile per thread.  Correctness first: the reference computes
// any d (numeric.py:1886-1935, :707-881, superoperator.py:51-84; a 5-qubit register is d = 32) and
// the drop-in must not raise where it does; speed comes second here (a few TFLOP/s on vector FMAs:
// still three to four orders of magnitude above the NumPy path at these sizes).
//
//   eigh_expm_generic_kernel      numpy.linalg.eigh + eigen-expm       numeric.py:1919-1929
//   prefix_generic_kernel         util.adot (prefix products)          util.py:868-877
//   prologue_generic_kernel       _propagate_eigenvectors, _transform_hamiltonian  numeric.py:93-141
//   accumulate_generic_kernel     the hot loop of calculate_control_matrix_from_scratch /
//                                 calculate_noise_operators_from_scratch  numeric.py:846-869, :596-609
//   conjugate_generic_kernel      U^dag C_i U for liouville_representation  superoperator.py:51-84
//
// Register tile: thread (r, c) = (tid / 16, tid % 16) of the 16 x 16 thread grid owns the entries
// (r + 16 a, c + 16 b), a, b < T = ceil(d / 16) <= 4, of every d x d result -- cyclic, so that a
// wavefront's LDS reads are 4 broadcast rows or 16 consecutive columns.
#include <algorithm>

#include "ffk_internal.h"

namespace ffk {
namespace {

constexpr int kGenThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// C = op(A) B for d x d matrices in LDS (row-major, leading dimension d), this thread's tile into
// acc[a][b].  CONJT: op(A) = A^dag, i.e. C[i][j] = sum_k conj(A[k][i]) B[k][j]; else C = A B.
template <int T, bool CONJT>
__device__ __forceinline__ void tile_matmul(const cplx* __restrict__ A, const cplx* __restrict__ B, int d,
                                            cplx (&acc)[T][T]) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = {0.0, 0.0};
    for (int k = 0; k < d; ++k) {
        cplx av[T], bv[T];
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int i = min(r + 16*a, d - 1);
            av[a] = CONJT ? A[k*d + i] : A[i*d + k];
        }
#pragma unroll
        for (int b = 0; b < T; ++b) bv[b] = B[k*d + min(c + 16*b, d - 1)];
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) {
                if (CONJT) cmac_conj(acc[a][b], av[a], bv[b]);
                else cmac(acc[a][b], av[a], bv[b]);
            }
    }
}

// tile -> row-major matrix (LDS or global), entries outside d x d dropped; scale optional
template <int T>
__device__ __forceinline__ void tile_store(cplx* __restrict__ M, int d, const cplx (&acc)[T][T],
                                           double scale = 1.0) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const int i = r + 16*a, j = c + 16*b;
            if (i < d && j < d) M[i*d + j] = {scale*acc[a][b].re, scale*acc[a][b].im};
        }
}

__device__ __forceinline__ void stage(cplx* __restrict__ dst, const cplx* __restrict__ src, int n) {
    for (int e = threadIdx.x; e < n; e += kGenThreads) dst[e] = src[e];
}

// ---- eigensolver -------------------------------------------------------------------------------
// Cyclic Jacobi with the d/2 disjoint rotations of a round-robin step applied together, like
// eigh.hip's wavefront version; lower triangle read (LAPACK UPLO = 'L'), eigenvalues ascending.
__device__ __forceinline__ bool jacobi_rotation_g(cplx apq, double alpha, double* c, cplx* w) {
    const double mag2 = apq.re*apq.re + apq.im*apq.im;
    const double r2 = fma(alpha, alpha, mag2);
    if (!(mag2 > 0.0) || !(r2 < 1e300) || r2 < 2.3e-308) return false;
    const double ri = rsqrt(r2);
    const double c2 = fma(0.5*fabs(alpha), ri, 0.5);
    const double ci = rsqrt(c2);
    const double k = (alpha >= 0.0 ? 0.5 : -0.5)*ri*ci;
    *c = c2*ci;
    *w = {k*apq.re, k*apq.im};
    return true;
}

__global__ __launch_bounds__(kGenThreads) void eigh_expm_generic_kernel(
    const cplx* __restrict__ H, const double* __restrict__ dt, int G, int d,
    double* __restrict__ eigvals, cplx* __restrict__ eigvecs, cplx* __restrict__ seg_prop,
    int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* A = reinterpret_cast<cplx*>(lds_raw);            // [d][d]
    cplx* V = A + d*d;                                     // [d][d]
    cplx* rot_w = V + d*d;                                 // [32]
    cplx* phase = rot_w + 32;                              // [64]
    double* rot_c = reinterpret_cast<double*>(phase + 64); // [32]
    double* red = rot_c + 32;                              // [4]
    int* rot_p = reinterpret_cast<int*>(red + 4);          // [32]
    int* rot_q = rot_p + 32;                               // [32]
    const int g = blockIdx.x, tid = threadIdx.x;
    const int dd = d*d, DP = d + (d & 1), NP = DP/2;
    const cplx* Hg = H + static_cast<size_t>(g)*dd;
    for (int e = tid; e < dd; e += kGenThreads) {
        const int i = e / d, j = e % d;
        cplx h;
        if (i >= j) {
            h = Hg[i*d + j];
        } else {
            h = Hg[j*d + i];
            h.im = -h.im;
        }
        if (i == j) h.im = 0.0;
        A[e] = h;
        V[e] = {i == j ? 1.0 : 0.0, 0.0};
    }
    __syncthreads();
    const double tol2 = static_cast<double>(dd)*4.930380657631324e-32;  // (d eps)^2
    auto off_and_total = [&](double* off, double* tot) {
        double o = 0.0, t = 0.0;
        for (int e = tid; e < dd; e += kGenThreads) {
            const cplx a = A[e];
            const double m2 = a.re*a.re + a.im*a.im;
            t += m2;
            if (e / d != e % d) o += m2;
        }
        *off = block_sum(o, red);
        *tot = block_sum(t, red);
    };
    bool converged = false;
    constexpr int kMaxSweeps = 60;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        if (sweep >= 3) {
            double off, tot;
            off_and_total(&off, &tot);
            if (off <= tol2*tot) {
                converged = true;
                break;
            }
        }
        for (int step = 0; step < DP - 1; ++step) {
            if (tid < NP) {
                int p, q;
                if (tid == 0) {
                    p = DP - 1;
                    q = step;
                } else {
                    p = (step + tid) % (DP - 1);
                    q = (step + (DP - 1) - tid) % (DP - 1);
                }
                if (p > q) {
                    const int tmp = p;
                    p = q;
                    q = tmp;
                }
                double c = 1.0;
                cplx w = {0.0, 0.0};
                bool valid = q < d;
                if (valid) valid = jacobi_rotation_g(A[p*d + q], 0.5*(A[q*d + q].re - A[p*d + p].re), &c, &w);
                rot_p[tid] = p;
                rot_q[tid] = valid ? q : -1;
                rot_c[tid] = c;
                rot_w[tid] = w;
            }
            __syncthreads();
            // column update of A and V:  (x_p, x_q) <- (c x_p - conj(w) x_q, w x_p + c x_q)
            for (int it = tid; it < NP*d*2; it += kGenThreads) {
                const int pr = it/(2*d), r = it % (2*d);
                const int q = rot_q[pr];
                if (q < 0) continue;
                const int p = rot_p[pr];
                const double c = rot_c[pr];
                const cplx w = rot_w[pr];
                cplx* M = (r >= d) ? V : A;
                const int row = r % d;
                const cplx xp = M[row*d + p], xq = M[row*d + q];
                M[row*d + p] = {c*xp.re - (w.re*xq.re + w.im*xq.im), c*xp.im - (w.re*xq.im - w.im*xq.re)};
                M[row*d + q] = {c*xq.re + (w.re*xp.re - w.im*xp.im), c*xq.im + (w.re*xp.im + w.im*xp.re)};
            }
            __syncthreads();
            // row update of A:  (x_p, x_q) <- (c x_p - w x_q, conj(w) x_p + c x_q)
            for (int it = tid; it < NP*d; it += kGenThreads) {
                const int pr = it / d, col = it % d;
                const int q = rot_q[pr];
                if (q < 0) continue;
                const int p = rot_p[pr];
                const double c = rot_c[pr];
                const cplx w = rot_w[pr];
                const cplx xp = A[p*d + col], xq = A[q*d + col];
                A[p*d + col] = {c*xp.re - (w.re*xq.re - w.im*xq.im), c*xp.im - (w.re*xq.im + w.im*xq.re)};
                A[q*d + col] = {c*xq.re + (w.re*xp.re + w.im*xp.im), c*xq.im + (w.re*xp.im - w.im*xp.re)};
            }
            __syncthreads();
            if (tid < NP && rot_q[tid] >= 0) {
                const int p = rot_p[tid], q = rot_q[tid];
                A[p*d + q] = {0.0, 0.0};
                A[q*d + p] = {0.0, 0.0};
                A[p*d + p].im = 0.0;
                A[q*d + q].im = 0.0;
            }
            __syncthreads();
        }
    }
    bool ok = true;
    if (!converged) {
        double off, tot;
        off_and_total(&off, &tot);
        ok = off <= 1e-24*tot;
    }
    // ascending order by rank (stable), eigenvalues / eigenvectors out
    double* ev = eigvals + static_cast<size_t>(g)*d;
    cplx* vg = eigvecs + static_cast<size_t>(g)*dd;
    if (tid < d) {
        const double li = A[tid*d + tid].re;
        int rank = 0;
        for (int j = 0; j < d; ++j) {
            const double lj = A[j*d + j].re;
            rank += (lj < li || (lj == li && j < tid)) ? 1 : 0;
        }
        ev[rank] = li;
        for (int row = 0; row < d; ++row) vg[row*d + rank] = V[row*d + tid];
        // exp(-i lambda dt): argument rounded exactly like util.cexp(-dt*eigvals), numeric.py:1929
        phase[tid] = cexp(-(dt[g]*li));
    }
    __syncthreads();
    // P = V diag(phase) V^dag  (the column pairing is order independent)
    cplx* P = seg_prop + static_cast<size_t>(g)*dd;
    for (int e = tid; e < dd; e += kGenThreads) {
        const int i = e / d, k = e % d;
        cplx acc = {0.0, 0.0};
        for (int j = 0; j < d; ++j) {
            const cplx vp = cmul(V[i*d + j], phase[j]);
            cmac_conj(acc, V[k*d + j], vp);
        }
        P[e] = acc;
    }
    if (tid == 0) status[g] = ok ? 0 : 1;
}

size_t eigh_generic_lds(int d) {
    return sizeof(cplx)*(2*static_cast<size_t>(d)*d + 32 + 64) + sizeof(double)*(32 + 4) + sizeof(int)*64;
}

// ---- prefix products: Q[0] = 1, Q[g+1] = P[g] Q[g] ------------------------------------------------
// One workgroup walks the segments (G is small where d is large; a chunked scan like scan.hip's is
// the next step if that changes): Q[g] stays in LDS, P[g] is staged beside it.
template <int T>
__global__ __launch_bounds__(kGenThreads) void prefix_generic_kernel(const cplx* __restrict__ P, int G,
                                                                     int d, cplx* __restrict__ Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* Qc = reinterpret_cast<cplx*>(lds_raw);
    cplx* Pg = Qc + d*d;
    const int dd = d*d;
    for (int e = threadIdx.x; e < dd; e += kGenThreads) {
        const cplx one = {e / d == e % d ? 1.0 : 0.0, 0.0};
        Qc[e] = one;
        Q[e] = one;
    }
    for (int g = 0; g < G; ++g) {
        __syncthreads();
        stage(Pg, P + static_cast<size_t>(g)*dd, dd);
        __syncthreads();
        cplx acc[T][T];
        tile_matmul<T, false>(Pg, Qc, d, acc);
        __syncthreads();
        tile_store<T>(Qc, d, acc);
        tile_store<T>(Q + static_cast<size_t>(g + 1)*dd, d, acc);
    }
}

// ---- prologue -----------------------------------------------------------------------------------
// ops[g][0] = T_g = V_g^dag Q_g, ops[g][1+a] = s_a(g) V_g^dag B_a V_g; Tc[g] = conj(T_g); table row
// (dt, t, 0, 0, D_0 .. D_{d-1}) -- the generic accumulate kernel evaluates the integral entries
// directly and needs no per-entry records; optional reference intermediates.
template <int T>
__global__ __launch_bounds__(kGenThreads) void prologue_generic_kernel(
    const double* __restrict__ eigvals, const cplx* __restrict__ eigvecs,
    const cplx* __restrict__ propagators, const cplx* __restrict__ n_opers,
    const double* __restrict__ n_coeffs, const double* __restrict__ dt, const double* __restrict__ t, int G,
    int d, int A, double* __restrict__ segtab, cplx* __restrict__ Tc, cplx* __restrict__ ops,
    cplx* __restrict__ n_opers_transformed, cplx* __restrict__ eigvecs_propagated) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* Vl = reinterpret_cast<cplx*>(lds_raw);
    cplx* Ml = Vl + d*d;
    const int g = blockIdx.x, tid = threadIdx.x, dd = d*d;
    const int r = tid >> 4, c = tid & 15;
    double* row = segtab + static_cast<size_t>(g)*seg_stride(d);
    if (tid == 0) {
        row[0] = dt[g];
        row[1] = t[g];
        row[2] = 0.0;
        row[3] = 0.0;
    }
    for (int m = tid; m < d; m += kGenThreads) row[4 + m] = eigvals[static_cast<size_t>(g)*d + m];
    stage(Vl, eigvecs + static_cast<size_t>(g)*dd, dd);
    stage(Ml, propagators + static_cast<size_t>(g)*dd, dd);
    __syncthreads();
    cplx acc[T][T];
    tile_matmul<T, true>(Vl, Ml, d, acc);                 // T = V^dag Q
    cplx* og = ops + static_cast<size_t>(g)*(1 + A)*dd;
    tile_store<T>(og, d, acc);
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const int i = r + 16*a, j = c + 16*b;
            if (i < d && j < d) {
                if (Tc) Tc[static_cast<size_t>(g)*dd + i*d + j] = {acc[a][b].re, -acc[a][b].im};
                // Q^dag V = T^dag
                if (eigvecs_propagated)
                    eigvecs_propagated[static_cast<size_t>(g)*dd + j*d + i] = {acc[a][b].re, -acc[a][b].im};
            }
        }
    for (int al = 0; al < A; ++al) {
        __syncthreads();
        stage(Ml, n_opers + static_cast<size_t>(al)*dd, dd);
        __syncthreads();
        tile_matmul<T, false>(Ml, Vl, d, acc);            // B V
        __syncthreads();
        tile_store<T>(Ml, d, acc);
        __syncthreads();
        tile_matmul<T, true>(Vl, Ml, d, acc);             // V^dag (B V)
        const double s = n_coeffs[static_cast<size_t>(al)*G + g];
        tile_store<T>(og + static_cast<size_t>(1 + al)*dd, d, acc, s);
        if (n_opers_transformed)
            tile_store<T>(n_opers_transformed + (static_cast<size_t>(al)*G + g)*dd, d, acc, s);
    }
}

// ---- accumulate ---------------------------------------------------------------------------------
// One workgroup per (frequency, noise operator, segment chunk):
//   Y += T_g^dag [Bbar_g o E_g(w)] T_g,   E_g(w)[m][n] = e^{i w t_g} I(w, D_m - D_n, dt_g),
// the integral entries by the direct, reference-exact evaluation (first_order_integral: the
// argument fl(fl(w + dE) dt) as in numeric.py:155-163).  T_g and X = Bbar o E (then Z = X T in its
// place) in LDS, this thread's tile of Z and of Y in registers.
template <int T>
__global__ __launch_bounds__(kGenThreads) void accumulate_generic_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int d, int A, int chunk_len, cplx* __restrict__ Ypart) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* Tl = reinterpret_cast<cplx*>(lds_raw);
    cplx* Xl = Tl + d*d;
    double* Dl = reinterpret_cast<double*>(Xl + d*d);      // [d]
    const int iw = blockIdx.x, al = blockIdx.y, z = blockIdx.z, tid = threadIdx.x, dd = d*d;
    const int r = tid >> 4, c = tid & 15;
    const double om = omega[iw];
    const int g0 = z*chunk_len, g1 = min(G, g0 + chunk_len);
    const int S = seg_stride(d);
    cplx Y[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) Y[a][b] = {0.0, 0.0};
    for (int g = g0; g < g1; ++g) {
        const double* row = segtab + static_cast<size_t>(g)*S;
        const cplx* og = ops + static_cast<size_t>(g)*(1 + A)*dd;
        const cplx* Bg = og + static_cast<size_t>(1 + al)*dd;
        __syncthreads();                                  // previous segment's reads of Tl / Xl done
        stage(Tl, og, dd);
        for (int m = tid; m < d; m += kGenThreads) Dl[m] = row[4 + m];
        __syncthreads();
        const double dtg = row[0];
        const cplx ph = cexp(om*row[1]);                   // e^{i w t_g}, argument rounded like numeric.py:865
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const int m = r + 16*a, n = c + 16*b;
                if (m < d && n < d) {
                    const cplx I = first_order_integral(om, Dl[m] - Dl[n], dtg);
                    Xl[m*d + n] = cmul(Bg[m*d + n], cmul(ph, I));
                }
            }
        __syncthreads();
        cplx Zt[T][T];
        tile_matmul<T, false>(Xl, Tl, d, Zt);             // Z = X T
        __syncthreads();
        tile_store<T>(Xl, d, Zt);
        __syncthreads();
        cplx Yt[T][T];
        tile_matmul<T, true>(Tl, Xl, d, Yt);              // T^dag Z
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b) {
                Y[a][b].re += Yt[a][b].re;
                Y[a][b].im += Yt[a][b].im;
            }
    }
    cplx* out = Ypart + ((static_cast<size_t>(z)*A + al)*dd)*W + iw;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const int i = r + 16*a, j = c + 16*b;
            if (i < d && j < d) out[static_cast<size_t>(i*d + j)*W] = Y[a][b];
        }
}

// ---- basis conjugation for the Liouville representation -------------------------------------------
// CB_i = U^dag C_i U, written into the GEMM's K-major real operands (layout: liouville.hip).
template <int T>
__global__ __launch_bounds__(kGenThreads) void conjugate_generic_kernel(
    const cplx* __restrict__ U, const cplx* __restrict__ basis, int N, int d, int Npad, int K,
    int want_imag, double* __restrict__ AopRe, double* __restrict__ AopIm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cplx* Ul = reinterpret_cast<cplx*>(lds_raw);
    cplx* Ml = Ul + d*d;
    const int i = blockIdx.x, bt = blockIdx.y, dd = d*d;
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    stage(Ul, U + static_cast<size_t>(bt)*dd, dd);
    stage(Ml, basis + static_cast<size_t>(i)*dd, dd);
    __syncthreads();
    cplx acc[T][T];
    tile_matmul<T, false>(Ml, Ul, d, acc);                // C U
    __syncthreads();
    tile_store<T>(Ml, d, acc);
    __syncthreads();
    tile_matmul<T, true>(Ul, Ml, d, acc);                 // U^dag (C U)
    double* are = AopRe + static_cast<size_t>(bt)*K*Npad;
    double* aim = AopIm + static_cast<size_t>(bt)*K*Npad;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) {
            const int ra = r + 16*a, cb = c + 16*b;
            if (ra < d && cb < d) {
                const size_t e = static_cast<size_t>(ra)*d + cb;
                if (want_imag) {
                    are[e*Npad + i] = acc[a][b].re;
                    are[(dd + e)*Npad + i] = -acc[a][b].im;
                    aim[e*Npad + i] = acc[a][b].im;
                    aim[(dd + e)*Npad + i] = acc[a][b].re;
                } else {
                    // Hermitian basis: K = d^2 rows (ffk_internal.h::hermitian_operand_row)
                    const int r0 = hermitian_operand_row(ra, cb, 0, d), r1 = hermitian_operand_row(ra, cb, 1, d);
                    if (r0 >= 0) are[static_cast<size_t>(r0)*Npad + i] = acc[a][b].re;
                    if (r1 >= 0) are[static_cast<size_t>(r1)*Npad + i] = -acc[a][b].im;
                }
            }
        }
}

template <typename K>
hipError_t allow_lds(K kern, size_t bytes) {
    if (bytes <= 48*1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(bytes));
}

// dispatch on the register tile T = ceil(d / 16)
#define FFK_GEN_DISPATCH(d, CALL)      \
    switch (((d) + 15)/16) {           \
        case 2: { CALL(2) } break;     \
        case 3: { CALL(3) } break;     \
        case 4: { CALL(4) } break;     \
        default: return hipErrorInvalidValue; \
    }

}  // namespace

bool generic_dimension(int d) { return d > kMaxD && d <= kMaxDGeneric; }

hipError_t launch_eigh_expm_generic(const cplx* H, const double* dt, int G, int d, double* eigvals,
                                    cplx* eigvecs, cplx* seg_prop, int* status, hipStream_t stream) {
    const size_t lds = eigh_generic_lds(d);
    hipError_t err = allow_lds(eigh_expm_generic_kernel, lds);
    if (err != hipSuccess) return err;
    for (int g0 = 0; g0 < G; g0 += 65535) {
        const int n = std::min(65535, G - g0);
        hipLaunchKernelGGL(eigh_expm_generic_kernel, dim3(n), dim3(kGenThreads), lds, stream,
                           H + static_cast<size_t>(g0)*d*d, dt + g0, n, d, eigvals + static_cast<size_t>(g0)*d,
                           eigvecs + static_cast<size_t>(g0)*d*d, seg_prop + static_cast<size_t>(g0)*d*d,
                           status + g0);
    }
    return hipGetLastError();
}

hipError_t launch_prefix_products_generic(const cplx* seg_prop, int G, int d, cplx* Q, hipStream_t stream) {
    const size_t lds = 2*sizeof(cplx)*static_cast<size_t>(d)*d;
#define FFK_CALL(T)                                                                                  \
    hipError_t err = allow_lds(prefix_generic_kernel<T>, lds);                                       \
    if (err != hipSuccess) return err;                                                               \
    hipLaunchKernelGGL(prefix_generic_kernel<T>, dim3(1), dim3(kGenThreads), lds, stream, seg_prop, G, d, Q);
    FFK_GEN_DISPATCH(d, FFK_CALL)
#undef FFK_CALL
    return hipGetLastError();
}

hipError_t launch_prologue_generic(const double* eigvals, const cplx* eigvecs, const cplx* propagators,
                                   const cplx* n_opers, const double* n_coeffs, const double* dt,
                                   const double* t, int G, int d, int A, double* segtab, cplx* Tc, cplx* ops,
                                   cplx* n_opers_transformed, cplx* eigvecs_propagated,
                                   hipStream_t stream) {
    if (G > 65535*32) return hipErrorInvalidValue;
    const size_t lds = 2*sizeof(cplx)*static_cast<size_t>(d)*d;
#define FFK_CALL(T)                                                                                  \
    hipError_t err = allow_lds(prologue_generic_kernel<T>, lds);                                     \
    if (err != hipSuccess) return err;                                                               \
    hipLaunchKernelGGL(prologue_generic_kernel<T>, dim3(G), dim3(kGenThreads), lds, stream, eigvals,  \
                       eigvecs, propagators, n_opers, n_coeffs, dt, t, G, d, A, segtab, Tc, ops,     \
                       n_opers_transformed, eigvecs_propagated);
    FFK_GEN_DISPATCH(d, FFK_CALL)
#undef FFK_CALL
    return hipGetLastError();
}

hipError_t launch_accumulate_generic(const double* omega, int W, const double* segtab, const cplx* ops,
                                     int G, int d, int A, int chunks, int chunk_len, cplx* Ypart,
                                     hipStream_t stream) {
    if (A > 65535 || chunks > 65535) return hipErrorInvalidValue;
    const size_t lds = 2*sizeof(cplx)*static_cast<size_t>(d)*d + sizeof(double)*64;
#define FFK_CALL(T)                                                                                  \
    hipError_t err = allow_lds(accumulate_generic_kernel<T>, lds);                                   \
    if (err != hipSuccess) return err;                                                               \
    hipLaunchKernelGGL(accumulate_generic_kernel<T>, dim3(W, A, chunks), dim3(kGenThreads), lds, stream, \
                       omega, W, segtab, ops, G, d, A, chunk_len, Ypart);
    FFK_GEN_DISPATCH(d, FFK_CALL)
#undef FFK_CALL
    return hipGetLastError();
}

hipError_t launch_conjugate_basis_generic(const cplx* U, int batch, int d, const cplx* basis, int N,
                                          int Npad, int K, int want_imag, double* AopRe, double* AopIm,
                                          hipStream_t stream) {
    if (batch > 65535) return hipErrorInvalidValue;
    const size_t lds = 2*sizeof(cplx)*static_cast<size_t>(d)*d;
#define FFK_CALL(T)                                                                                  \
    hipError_t err = allow_lds(conjugate_generic_kernel<T>, lds);                                    \
    if (err != hipSuccess) return err;                                                               \
    hipLaunchKernelGGL(conjugate_generic_kernel<T>, dim3(N, batch), dim3(kGenThreads), lds, stream, U, \
                       basis, N, d, Npad, K, want_imag, AopRe, AopIm);
    FFK_GEN_DISPATCH(d, FFK_CALL)
#undef FFK_CALL
    return hipGetLastError();
}

}  // namespace ffk
