// eigh.hip -- K1: batched Hermitian eigendecomposition + eigen-expm, one wavefront per segment.
//
// Replaces numpy.linalg.eigh + the 'lij,jl,lkj->lik' einsum of numeric.diagonalize
// (filter_functions/numeric.py:1919, 1928-1929).  The d x d matrix and its eigenvector
// accumulator live in LDS; the D/2 disjoint rotations of one round-robin step of the cyclic
// Jacobi method are computed and applied by the 64 lanes in parallel (complex Hermitian
// rotations, rows then columns).  Only the lower triangle of the input is read, like
// LAPACK's UPLO='L' default; eigenvalues are returned ascending.
#include <algorithm>

#include "ffk_internal.h"

namespace ffk {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Wave-private LDS state of one eigensolve
template <int D>
struct EighState {
    static constexpr int DP = D + (D & 1);  // round-robin players (a bye when D is odd)
    static constexpr int NP = DP/2;         // rotations per step
    cplx A[D][D];
    cplx V[D][D];
    cplx rot_w[NP];
    double rot_c[NP];
    int rot_p[NP], rot_q[NP];
    cplx phase[D];
};

// Intra-wavefront hand-off through LDS: the LDS unit executes one wave's instructions in order,
// so only the compiler needs telling that other lanes wrote memory.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Complex Jacobi rotation annihilating a_pq:  (x_p, x_q) <- (c x_p - conj(w) x_q, w x_p + c x_q) on
// columns, with alpha = (a_qq - a_pp)/2, r = sqrt(alpha^2 + |a_pq|^2):
//     cos 2t = |alpha|/r,  c = sqrt((1 + cos 2t)/2),  |w| = sin t = (|a_pq|/r)/(2c)   (|t| <= pi/4),
//     w = sign(alpha) a_pq/(2 r c).
// Two reciprocal square roots in the dependent chain, no division, no square root (this scalar
// section is the critical path of every Jacobi step; the tan-based form needed four).  c^2 lies in
// [1/2, 1]: no cancellation.  Returns false where there is nothing to rotate.
__device__ __forceinline__ bool jacobi_rotation(cplx apq, double alpha, double* c, cplx* w) {
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

// One wavefront: eigendecomposition of Hg (lower triangle), eigenvalues/eigenvectors to global
// memory in ascending order, segment propagator P = V exp(-i D dt) V^dag to `P` (LDS or global).
// Returns false if the Jacobi iteration clearly failed to converge.
template <int D>
__device__ __forceinline__ bool eigh_expm_wave(EighState<D>& st, const cplx* __restrict__ Hg, double dtg, int lane,
                               double* __restrict__ eigvals_g, cplx* __restrict__ eigvecs_g,
                               cplx* P) {
    constexpr int DP = EighState<D>::DP;
    constexpr int NP = EighState<D>::NP;
    constexpr int kMaxSweeps = 40;
    constexpr bool kRegisterRotations = NP*2*D <= 64;    // D <= 8
    auto& A = st.A;
    auto& V = st.V;

    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, j = e % D;
        cplx h;
        if (i >= j) {
            h = Hg[i*D + j];
        } else {
            h = Hg[j*D + i];
            h.im = -h.im;
        }
        if (i == j) h.im = 0.0;
        A[i][j] = h;
        V[i][j] = {i == j ? 1.0 : 0.0, 0.0};
    }
    wave_sync();

    const double tol2 = static_cast<double>(D*D)*4.930380657631324e-32;  // (D eps)^2
    bool converged = false;
    // A dense matrix needs at least four sweeps to reach (D eps)^2: the convergence test (an LDS
    // pass and two wave reductions) starts after the third; on an already diagonal matrix the
    // untested sweeps are no-ops (nothing to rotate).
    constexpr int kFirstTest = D >= 3 ? 3 : 1;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        if (sweep >= kFirstTest) {
            double off = 0.0, tot = 0.0;
            for (int e = lane; e < D*D; e += 64) {
                const int i = e / D, j = e % D;
                const cplx a = A[i][j];
                const double m2 = a.re*a.re + a.im*a.im;
                tot += m2;
                if (i != j) off += m2;
            }
            off = wave_sum(off);
            tot = wave_sum(tot);
            if (off <= tol2*tot) {
                converged = true;
                break;
            }
        }
        for (int step = 0; step < DP - 1; ++step) {
            if constexpr (kRegisterRotations) {
                // One lane per (rotation, matrix, row): every lane derives its rotation's angle
                // itself (no hand-off through LDS) and keeps it in registers for the column update
                // (A and V) and the row update (A; lanes of the A half, their row index as the
                // column).  Two dependent LDS round trips per step instead of four.
                const int pr = lane/(2*D), r = lane % (2*D);
                int p, q;
                if (pr == 0) {
                    p = DP - 1;
                    q = step;
                } else {
                    p = (step + pr) % (DP - 1);
                    q = (step + (DP - 1) - pr) % (DP - 1);
                }
                if (p > q) {
                    const int tmp = p;
                    p = q;
                    q = tmp;
                }
                bool valid = lane < NP*2*D && q < D;
                double c = 1.0;
                cplx w = {0.0, 0.0};
                cplx(*M)[D] = (r >= D) ? V : A;
                const int row = r % D;
                cplx xp = {0.0, 0.0}, xq = {0.0, 0.0};
                if (valid) {
                    const cplx apq = A[p][q];
                    const double alpha = 0.5*(A[q][q].re - A[p][p].re);
                    xp = M[row][p];
                    xq = M[row][q];
                    valid = jacobi_rotation(apq, alpha, &c, &w);
                }
                if (valid) {
                    M[row][p] = {c*xp.re - (w.re*xq.re + w.im*xq.im), c*xp.im - (w.re*xq.im - w.im*xq.re)};
                    M[row][q] = {c*xq.re + (w.re*xp.re - w.im*xp.im), c*xq.im + (w.re*xp.im + w.im*xp.re)};
                }
                wave_sync();
                if (valid && r < D) {
                    const int col = r;
                    xp = A[p][col];
                    xq = A[q][col];
                    cplx np = {c*xp.re - (w.re*xq.re - w.im*xq.im), c*xp.im - (w.re*xq.im + w.im*xq.re)};
                    cplx nq = {c*xq.re + (w.re*xp.re + w.im*xp.im), c*xq.im + (w.re*xp.im - w.im*xp.re)};
                    // the annihilated pair is exactly zero, the diagonal exactly real
                    if (col == q) {
                        np = {0.0, 0.0};
                        nq.im = 0.0;
                    }
                    if (col == p) {
                        nq = {0.0, 0.0};
                        np.im = 0.0;
                    }
                    A[p][col] = np;
                    A[q][col] = nq;
                }
                wave_sync();
                continue;
            }
            if (lane < NP) {
                int p, q;
                if (lane == 0) {
                    p = DP - 1;
                    q = step;
                } else {
                    p = (step + lane) % (DP - 1);
                    q = (step + (DP - 1) - lane) % (DP - 1);
                }
                if (p > q) {
                    const int tmp = p;
                    p = q;
                    q = tmp;
                }
                double c = 1.0;
                cplx w = {0.0, 0.0};
                bool valid = q < D;
                if (valid) valid = jacobi_rotation(A[p][q], 0.5*(A[q][q].re - A[p][p].re), &c, &w);
                st.rot_p[lane] = p;
                st.rot_q[lane] = valid ? q : -1;
                st.rot_c[lane] = c;
                st.rot_w[lane] = w;
            }
            wave_sync();
            // column update of A and V:  (x_p, x_q) <- (c x_p - conj(w) x_q, w x_p + c x_q)
            for (int it = lane; it < NP*D*2; it += 64) {
                const int pr = it/(2*D), r = it % (2*D);
                const int q = st.rot_q[pr];
                if (q < 0) continue;
                const int p = st.rot_p[pr];
                const double c = st.rot_c[pr];
                const cplx w = st.rot_w[pr];
                cplx(*M)[D] = (r >= D) ? V : A;
                const int row = r % D;
                const cplx xp = M[row][p], xq = M[row][q];
                M[row][p] = {c*xp.re - (w.re*xq.re + w.im*xq.im), c*xp.im - (w.re*xq.im - w.im*xq.re)};
                M[row][q] = {c*xq.re + (w.re*xp.re - w.im*xp.im), c*xq.im + (w.re*xp.im + w.im*xp.re)};
            }
            wave_sync();
            // row update of A:  (x_p, x_q) <- (c x_p - w x_q, conj(w) x_p + c x_q)
            for (int it = lane; it < NP*D; it += 64) {
                const int pr = it / D, col = it % D;
                const int q = st.rot_q[pr];
                if (q < 0) continue;
                const int p = st.rot_p[pr];
                const double c = st.rot_c[pr];
                const cplx w = st.rot_w[pr];
                const cplx xp = A[p][col], xq = A[q][col];
                A[p][col] = {c*xp.re - (w.re*xq.re - w.im*xq.im), c*xp.im - (w.re*xq.im + w.im*xq.re)};
                A[q][col] = {c*xq.re + (w.re*xp.re + w.im*xp.im), c*xq.im + (w.re*xp.im - w.im*xp.re)};
            }
            wave_sync();
            if (lane < NP && st.rot_q[lane] >= 0) {
                const int p = st.rot_p[lane], q = st.rot_q[lane];
                A[p][q] = {0.0, 0.0};
                A[q][p] = {0.0, 0.0};
                A[p][p].im = 0.0;
                A[q][q].im = 0.0;
            }
            wave_sync();
        }
    }
    bool ok = true;
    if (!converged) {
        // accept a stall just above the threshold, flag a genuine failure
        double off = 0.0, tot = 0.0;
        for (int e = lane; e < D*D; e += 64) {
            const int i = e / D, j = e % D;
            const cplx a = A[i][j];
            const double m2 = a.re*a.re + a.im*a.im;
            tot += m2;
            if (i != j) off += m2;
        }
        off = wave_sum(off);
        tot = wave_sum(tot);
        ok = off <= 1e-24*tot;
    }

    // ascending order by rank (stable), write eigenvalues / eigenvectors
    if (lane < D) {
        const double li = A[lane][lane].re;
        int rank = 0;
        for (int j = 0; j < D; ++j) {
            const double lj = A[j][j].re;
            rank += (lj < li || (lj == li && j < lane)) ? 1 : 0;
        }
        eigvals_g[rank] = li;
        for (int row = 0; row < D; ++row) eigvecs_g[row*D + rank] = V[row][lane];
        // exp(-i lambda dt): argument rounded exactly like util.cexp(-dt*eigvals), numeric.py:1929
        st.phase[lane] = cexp(-(dtg*li));
    }
    wave_sync();
    // P = V diag(phase) V^dag  (column pairing is order independent)
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, k = e % D;
        cplx acc = {0.0, 0.0};
        for (int j = 0; j < D; ++j) {
            const cplx vp = cmul(V[i][j], st.phase[j]);
            cmac_conj(acc, V[k][j], vp);  // += conj(V[k][j]) * vp
        }
        P[e] = acc;
    }
    return ok;
}

// Stand-alone K1: one wavefront (= one block) per segment.
template <int D>
__global__ __launch_bounds__(64) void eigh_expm_kernel(const cplx* __restrict__ H,
                                                       const double* __restrict__ dt, int G,
                                                       double* __restrict__ eigvals,
                                                       cplx* __restrict__ eigvecs,
                                                       cplx* __restrict__ seg_prop,
                                                       int* __restrict__ status,
                                                       int* __restrict__ fail_count) {
    __shared__ EighState<D> st;
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const bool ok = eigh_expm_wave<D>(st, H + static_cast<size_t>(g)*D*D, dt[g], lane,
                                      eigvals + static_cast<size_t>(g)*D,
                                      eigvecs + static_cast<size_t>(g)*D*D,
                                      seg_prop + static_cast<size_t>(g)*D*D);
    if (lane == 0) {
        status[g] = ok ? 0 : 1;
        // (resident API path: a word in mapped host memory that the host zeroed before the launch -- the count of
        // flagged segments without a memset, a counting kernel and their two launch floors; never taken when all is well)
        if (!ok && fail_count != nullptr)
            __hip_atomic_fetch_add(fail_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same with the Hamiltonian summed in the kernel: H[g] = sum_i coeffs[i, g] opers[i] (pulse_sequence.py:1300-1302,
// 'ijk,il->ljk', in operator order -- the order and the FMAs of the resident path's assemble_hamiltonian_kernel,
// whose launch and whose separate copy of the controls this saves there).  A kernel of its own: the plain one's
// registers and LDS are budgeted for running beside another pass's accumulate blocks.
template <int D>
__global__ __launch_bounds__(64) void eigh_expm_controls_kernel(const cplx* __restrict__ opers,
                                                                const double* __restrict__ coeffs, int n_c,
                                                                const double* __restrict__ dt, int G,
                                                                double* __restrict__ eigvals,
                                                                cplx* __restrict__ eigvecs,
                                                                cplx* __restrict__ seg_prop,
                                                                int* __restrict__ status,
                                                                int* __restrict__ fail_count) {
    __shared__ EighState<D> st;
    __shared__ cplx Hs[D*D];
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < D*D; e += 64) {
        cplx acc = {0.0, 0.0};
        for (int i = 0; i < n_c; ++i) {
            const double c = coeffs[static_cast<size_t>(i)*G + g];
            const cplx o = opers[i*D*D + e];
            acc.re = fma(c, o.re, acc.re);
            acc.im = fma(c, o.im, acc.im);
        }
        Hs[e] = acc;
    }
    wave_sync();
    const bool ok = eigh_expm_wave<D>(st, Hs, dt[g], lane, eigvals + static_cast<size_t>(g)*D,
                                      eigvecs + static_cast<size_t>(g)*D*D,
                                      seg_prop + static_cast<size_t>(g)*D*D);
    if (lane == 0) {
        status[g] = ok ? 0 : 1;
        if (!ok && fail_count != nullptr)
            __hip_atomic_fetch_add(fail_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int D>
hipError_t launch_d(const cplx* H, const double* dt, int G, double* eigvals, cplx* eigvecs,
                    cplx* seg_prop, int* status, int* fail_count, hipStream_t stream) {
    hipLaunchKernelGGL(eigh_expm_kernel<D>, dim3(G), dim3(64), 0, stream, H, dt, G, eigvals,
                       eigvecs, seg_prop, status, fail_count);
    return hipGetLastError();
}

// Number of segments flagged by the eigensolver, as one device integer (integer sum: the order of
// the atomics does not matter).  One block per 64 Ki segments: a single block took 207 us over the
// 200 002 flags of a long sequence.
__global__ __launch_bounds__(256) void count_failures_kernel(const int* __restrict__ status, int G,
                                                            int32_t* __restrict__ out) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int mine = 0;
    for (int g = blockIdx.x*blockDim.x + threadIdx.x; g < G; g += gridDim.x*blockDim.x) mine += status[g] != 0;
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && total) atomicAdd(out, total);
}

}  // namespace

hipError_t launch_count_failures(const int* status, int G, int32_t* out, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(out, 0, sizeof(int32_t), stream);
    if (err != hipSuccess) return err;
    const int blocks = std::max(1, std::min(256, (G + 65535)/65536*16));
    hipLaunchKernelGGL(count_failures_kernel, dim3(blocks), dim3(256), 0, stream, status, G, out);
    return hipGetLastError();
}

bool eigh_fail_count_supported(int d) { return !generic_dimension(d); }

hipError_t launch_eigh_expm_controls(const cplx* opers, const double* coeffs, int n_c, const double* dt, int G,
                                     int d, double* eigvals, cplx* eigvecs, cplx* seg_prop, int* status,
                                     hipStream_t stream, int* fail_count) {
    switch (d) {
#define FFK_CASE(D)                                                                                          \
    case D:                                                                                                  \
        hipLaunchKernelGGL(eigh_expm_controls_kernel<D>, dim3(G), dim3(64), 0, stream, opers, coeffs, n_c,   \
                           dt, G, eigvals, eigvecs, seg_prop, status, fail_count);                           \
        return hipGetLastError();
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

hipError_t launch_eigh_expm(const cplx* H, const double* dt, int G, int d, double* eigvals,
                            cplx* eigvecs, cplx* seg_prop, int* status, hipStream_t stream, int* fail_count) {
    if (generic_dimension(d))
        return launch_eigh_expm_generic(H, dt, G, d, eigvals, eigvecs, seg_prop, status, stream);
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_d<D>(H, dt, G, eigvals, eigvecs, seg_prop, status, fail_count, stream);
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace ffk
