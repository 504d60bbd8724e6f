// eigh.hip -- K1: batched Hermitian eigendecomposition + eigen-expm, one wavefront per segment.
//
// Replaces numpy.linalg.eigh + the 'lij,jl,lkj->lik' einsum of numeric.diagonalize
// (filter_functions/numeric.py:1919, 1928-1929).  The d x d matrix and its eigenvector
// accumulator live in LDS; the D/2 disjoint rotations of one round-robin step of the cyclic
// Jacobi method are computed and applied by the 64 lanes in parallel (complex Hermitian
// rotations, rows then columns).  Only the lower triangle of the input is read, like
// LAPACK's UPLO='L' default; eigenvalues are returned ascending.
#include "ffk_internal.h"

namespace ffk {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int D>
__global__ __launch_bounds__(64) void eigh_expm_kernel(const cplx* __restrict__ H,
                                                       const double* __restrict__ dt, int G,
                                                       double* __restrict__ eigvals,
                                                       cplx* __restrict__ eigvecs,
                                                       cplx* __restrict__ seg_prop,
                                                       int* __restrict__ status) {
    constexpr int DP = D + (D & 1);  // round-robin players (a bye when D is odd)
    constexpr int NP = DP/2;         // rotations per step
    constexpr int kMaxSweeps = 40;
    __shared__ cplx A[D][D];
    __shared__ cplx V[D][D];
    __shared__ double rot_c[NP];
    __shared__ cplx rot_w[NP];
    __shared__ int rot_p[NP], rot_q[NP];
    __shared__ double lam[D];
    __shared__ cplx phase[D];

    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const cplx* Hg = H + static_cast<size_t>(g)*D*D;

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
    __syncthreads();

    const double tol2 = static_cast<double>(D*D)*4.930380657631324e-32;  // (D eps)^2
    bool converged = false;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
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
        for (int step = 0; step < DP - 1; ++step) {
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
                if (valid) {
                    const cplx apq = A[p][q];
                    const double mag2 = apq.re*apq.re + apq.im*apq.im;
                    if (mag2 > 0.0) {
                        const double mag = sqrt(mag2);
                        const double tau = (A[q][q].re - A[p][p].re)/(2.0*mag);
                        const double sgn = tau >= 0.0 ? 1.0 : -1.0;
                        const double tt = sgn/(fabs(tau) + sqrt(fma(tau, tau, 1.0)));
                        c = 1.0/sqrt(fma(tt, tt, 1.0));
                        const double s = tt*c;
                        w = {s*apq.re/mag, s*apq.im/mag};
                    } else {
                        valid = false;
                    }
                }
                rot_p[lane] = p;
                rot_q[lane] = valid ? q : -1;
                rot_c[lane] = c;
                rot_w[lane] = w;
            }
            __syncthreads();
            // column update of A and V:  (x_p, x_q) <- (c x_p - conj(w) x_q, w x_p + c x_q)
            for (int it = lane; it < NP*D*2; it += 64) {
                const int pr = it/(2*D), r = it % (2*D);
                const int q = rot_q[pr];
                if (q < 0) continue;
                const int p = rot_p[pr];
                const double c = rot_c[pr];
                const cplx w = rot_w[pr];
                cplx(*M)[D] = (r >= D) ? V : A;
                const int row = r % D;
                const cplx xp = M[row][p], xq = M[row][q];
                M[row][p] = {c*xp.re - (w.re*xq.re + w.im*xq.im), c*xp.im - (w.re*xq.im - w.im*xq.re)};
                M[row][q] = {c*xq.re + (w.re*xp.re - w.im*xp.im), c*xq.im + (w.re*xp.im + w.im*xp.re)};
            }
            __syncthreads();
            // row update of A:  (x_p, x_q) <- (c x_p - w x_q, conj(w) x_p + c x_q)
            for (int it = lane; it < NP*D; it += 64) {
                const int pr = it / D, col = it % D;
                const int q = rot_q[pr];
                if (q < 0) continue;
                const int p = rot_p[pr];
                const double c = rot_c[pr];
                const cplx w = rot_w[pr];
                const cplx xp = A[p][col], xq = A[q][col];
                A[p][col] = {c*xp.re - (w.re*xq.re - w.im*xq.im), c*xp.im - (w.re*xq.im + w.im*xq.re)};
                A[q][col] = {c*xq.re + (w.re*xp.re + w.im*xp.im), c*xq.im + (w.re*xp.im - w.im*xp.re)};
            }
            __syncthreads();
            if (lane < NP && rot_q[lane] >= 0) {
                const int p = rot_p[lane], q = rot_q[lane];
                A[p][q] = {0.0, 0.0};
                A[q][p] = {0.0, 0.0};
                A[p][p].im = 0.0;
                A[q][q].im = 0.0;
            }
            __syncthreads();
        }
    }
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
        if (!(off <= 1e-24*tot) && lane == 0) atomicAdd(status, 1);
    }

    // ascending order by rank (stable), write eigenvalues / eigenvectors
    if (lane < D) {
        const double li = A[lane][lane].re;
        int rank = 0;
        for (int j = 0; j < D; ++j) {
            const double lj = A[j][j].re;
            rank += (lj < li || (lj == li && j < lane)) ? 1 : 0;
        }
        lam[rank] = li;
        eigvals[static_cast<size_t>(g)*D + rank] = li;
        rot_p[0] = 0;  // (keeps rot_p live; no effect)
        // column `lane` of V goes to column `rank`
        for (int row = 0; row < D; ++row)
            eigvecs[(static_cast<size_t>(g)*D + row)*D + rank] = V[row][lane];
        // exp(-i lambda dt): argument rounded exactly like util.cexp(-dt*eigvals), numeric.py:1929
        phase[lane] = cexp(-(dt[g]*li));
    }
    __syncthreads();
    // P = V diag(phase) V^dag  (column pairing is order independent)
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, k = e % D;
        cplx acc = {0.0, 0.0};
        for (int j = 0; j < D; ++j) {
            const cplx vp = cmul(V[i][j], phase[j]);
            cmac_conj(acc, V[k][j], vp);  // += conj(V[k][j]) * vp
        }
        seg_prop[static_cast<size_t>(g)*D*D + e] = acc;
    }
}

template <int D>
hipError_t launch_d(const cplx* H, const double* dt, int G, double* eigvals, cplx* eigvecs,
                    cplx* seg_prop, int* status, hipStream_t stream) {
    hipLaunchKernelGGL(eigh_expm_kernel<D>, dim3(G), dim3(64), 0, stream, H, dt, G, eigvals,
                       eigvecs, seg_prop, status);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_eigh_expm(const cplx* H, const double* dt, int G, int d, double* eigvals,
                            cplx* eigvecs, cplx* seg_prop, int* status, hipStream_t stream) {
    switch (d) {
#define FFK_CASE(D) \
    case D:         \
        return launch_d<D>(H, dt, G, eigvals, eigvecs, seg_prop, status, stream);
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace ffk
