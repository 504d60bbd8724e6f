// prep.hip -- omega-independent operands of the control-matrix accumulation, one wavefront per
// segment, plus the two "cache_intermediates" side products that are cheap element-wise maps.
//
// Reference arithmetic restated here:
//   T_g = V_g^dag Q_g                         (_propagate_eigenvectors, numeric.py:93-95, :577)
//   Bbar_a^(g) = s_a(g) V_g^dag B_a V_g       (_transform_hamiltonian, numeric.py:98-141)
//   dE^(g)[m,n] = D_m - D_n                   (np.subtract.outer, numeric.py:155)
// packed per segment as ops[g] = (T_g, Bbar_0^(g), ..., Bbar_{A-1}^(g)), the block of operands the
// accumulate kernel stages in LDS for segment g.
#include "ffk_internal.h"

namespace ffk {
namespace {

// Prologue arithmetic for one segment by one wavefront; V (eigenvectors) and Q (propagator
// before the segment) already sit in LDS.
template <int D>
__device__ void prologue_segment(const cplx (*V)[D], cplx (*Q)[D], cplx (*T)[D], cplx (*BV)[D],
                                 int g, int lane, const double* __restrict__ eigvals,
                                 const cplx* __restrict__ n_opers,
                                 const double* __restrict__ n_coeffs,
                                 const double* __restrict__ dt, const double* __restrict__ t, int G,
                                 int A, double* __restrict__ segtab, cplx* __restrict__ Tc,
                                 cplx* __restrict__ ops, cplx* __restrict__ n_opers_transformed,
                                 cplx* __restrict__ eigvecs_propagated, bool with_noise_ops = true,
                                 cplx* __restrict__ wfold = nullptr) {
    constexpr int S = seg_stride(D);
    double* st = segtab + static_cast<size_t>(g)*S;
    if (lane == 0) {
        st[0] = dt[g];
        st[1] = t[g];
    }
    cplx eb = {1.0, 0.0};                              // e^{i b_mn} of this lane's entry (d = 4: one entry per lane < 16)
    for (int e = lane; e < D*D; e += 64) {
        const double dE = eigvals[static_cast<size_t>(g)*D + e / D] - eigvals[static_cast<size_t>(g)*D + e % D];
        double sb, cb;
        sincos_pi(0.5*(dE*dt[g]), &sb, &cb);
        st[seg_rec(e)] = dE;
        st[seg_rec(e) + 1] = sb;
        st[seg_rec(e) + 2] = cb;
        st[seg_rec(e) + 3] = 0.0;
        eb = {cb, sb};
    }
    if (lane < 2) st[2 + lane] = 0.0;
    for (int e = 4 + 4*D*D + lane; e < S; e += 64) st[e] = 0.0;

    // T = V^dag Q
    for (int e = lane; e < D*D; e += 64) {
        const int m = e / D, j = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) cmac_conj(acc, V[k][m], Q[k][j]);
        T[m][j] = acc;
        Tc[static_cast<size_t>(g)*D*D + e] = {acc.re, -acc.im};
        ops[static_cast<size_t>(g)*(1 + A)*D*D + e] = acc;
        // eigvecs_propagated = Q^dag V = T^dag:  [i][j] = conj(T[j][i])
        if (eigvecs_propagated)
            eigvecs_propagated[static_cast<size_t>(g)*D*D + j*D + m] = {acc.re, -acc.im};
    }
    __syncthreads();

    if (!with_noise_ops) return;        // (large d: noise_ops_kernel, one block per operator)
    // d = 4 with a buffer for it (ffk_internal.h wfold): W_a[n][m][j] = Bbar_a[m][n] e^{i b_mn} T[n][j], the
    // operand the accumulate kernel's tiles hold, folded here once per segment.  Q is free by now: it keeps e^{ib}.
    const bool fold = D == 4 && wfold != nullptr;
    if (fold) {
        if (lane < D*D) Q[lane / D][lane % D] = eb;
        __syncthreads();
    }
    for (int a = 0; a < A; ++a) {
        const cplx* B = n_opers + static_cast<size_t>(a)*D*D;
        const double s = n_coeffs ? n_coeffs[static_cast<size_t>(a)*G + g] : 1.0;   // NULL: unit
        // BV = B V
        for (int e = lane; e < D*D; e += 64) {
            const int i = e / D, n = e % D;
            cplx acc = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < D; ++k) cmac(acc, B[i*D + k], V[k][n]);
            BV[i][n] = acc;
        }
        __syncthreads();
        // Bbar = s V^dag BV
        cplx bbar = {0.0, 0.0};
        for (int e = lane; e < D*D; e += 64) {
            const int m = e / D, n = e % D;
            cplx acc = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < D; ++k) cmac_conj(acc, V[k][m], BV[k][n]);
            acc.re *= s;
            acc.im *= s;
            ops[(static_cast<size_t>(g)*(1 + A) + 1 + a)*D*D + e] = acc;
            if (n_opers_transformed)
                n_opers_transformed[(static_cast<size_t>(a)*G + g)*D*D + e] = acc;
            if (fold) bbar = acc;
        }
        __syncthreads();
        if (fold) {
            if (lane < D*D) BV[lane / D][lane % D] = bbar;      // (every lane has its products out of BV: the barrier above)
            __syncthreads();
            if constexpr (D == 4) {
                // lane = (m, n, j), the accumulate kernel's order of products: (e^{ib} T) first, then Bbar times that
                const int m = lane >> 4, n = (lane >> 2) & 3, j = lane & 3;
                const cplx et = cmul(Q[m][n], T[n][j]);
                const cplx w = cmul(BV[m][n], et);
                wfold[(static_cast<size_t>(g)*A + a)*64 + n*16 + m*4 + j] = w;
            }
            __syncthreads();
        }
    }
}

template <int D>
__global__ __launch_bounds__(64) void prologue_kernel(
    const double* __restrict__ eigvals, const cplx* __restrict__ eigvecs,
    const cplx* __restrict__ propagators, const cplx* __restrict__ n_opers,
    const double* __restrict__ n_coeffs, const double* __restrict__ dt,
    const double* __restrict__ t, int G, int A, double* __restrict__ segtab,
    cplx* __restrict__ Tc, cplx* __restrict__ ops, cplx* __restrict__ n_opers_transformed,
    cplx* __restrict__ eigvecs_propagated, bool with_noise_ops, cplx* __restrict__ wfold) {
    __shared__ cplx V[D][D];
    __shared__ cplx Q[D][D];
    __shared__ cplx T[D][D];
    __shared__ cplx BV[D][D];
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    for (int e = lane; e < D*D; e += 64) {
        V[e / D][e % D] = eigvecs[static_cast<size_t>(g)*D*D + e];
        Q[e / D][e % D] = propagators[static_cast<size_t>(g)*D*D + e];
    }
    __syncthreads();
    prologue_segment<D>(V, Q, T, BV, g, lane, eigvals, n_opers, n_coeffs, dt, t, G, A, segtab, Tc,
                        ops, n_opers_transformed, eigvecs_propagated, with_noise_ops, wfold);
}

// Bbar_a^(g) = s_a(t_g) V_g^dag B_a V_g for one (segment, noise operator) per block.  For large d the
// two d^3 products per operator dominate the prologue (d = 16, 18 operators: 172 us for 13 blocks
// of one wavefront each); spread over G x A blocks they take as long as one operator.
template <int D>
__global__ __launch_bounds__(64) void noise_ops_kernel(const cplx* __restrict__ eigvecs,
                                                       const cplx* __restrict__ n_opers,
                                                       const double* __restrict__ n_coeffs, int G,
                                                       int A, cplx* __restrict__ ops,
                                                       cplx* __restrict__ n_opers_transformed) {
    __shared__ cplx V[D][D];
    __shared__ cplx BV[D][D];
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    const int g = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    for (int e = lane; e < D*D; e += 64) V[e / D][e % D] = eigvecs[static_cast<size_t>(g)*D*D + e];
    __syncthreads();
    const cplx* B = n_opers + static_cast<size_t>(a)*D*D;
    const double s = n_coeffs ? n_coeffs[static_cast<size_t>(a)*G + g] : 1.0;   // NULL: unit
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, n = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) cmac(acc, B[i*D + k], V[k][n]);
        BV[i][n] = acc;
    }
    __syncthreads();
    for (int e = lane; e < D*D; e += 64) {
        const int m = e / D, n = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < D; ++k) cmac_conj(acc, V[k][m], BV[k][n]);
        acc.re *= s;
        acc.im *= s;
        ops[(static_cast<size_t>(g)*(1 + A) + 1 + a)*D*D + e] = acc;
        if (n_opers_transformed)
            n_opers_transformed[(static_cast<size_t>(a)*G + g)*D*D + e] = acc;
    }
}

// noise operators by their own launch from this dimension on (and only with several of them)
constexpr int kNoiseOpsKernelMinD = 12;
inline bool split_noise_ops(int d, int A) { return d >= kNoiseOpsKernelMinD && A >= 2 && A <= 65535; }

// Fused scan fix-up + prologue, one wavefront per segment:
//   E_c = T_{c-1} ... T_0 rebuilt serially from the chunk totals (<= 63 small products, done
//   redundantly by every block instead of a third launch), Q[g+1] = Qloc[g+1] E_c written out,
//   Q[g] = Qloc[g] E_c kept in LDS and fed straight into the prologue.
template <int D>
__global__ __launch_bounds__(64) void apply_prologue_kernel(
    const cplx* __restrict__ Qloc, const cplx* __restrict__ totals, int G, int L,
    cplx* __restrict__ Qout, const double* __restrict__ eigvals, const cplx* __restrict__ eigvecs,
    const cplx* __restrict__ n_opers, const double* __restrict__ n_coeffs,
    const double* __restrict__ dt, const double* __restrict__ t, int A,
    double* __restrict__ segtab, cplx* __restrict__ Tc, cplx* __restrict__ ops,
    const cplx* __restrict__ basis, int* __restrict__ nnz, int* __restrict__ rows,
    cplx* __restrict__ vals, bool with_noise_ops, cplx* __restrict__ wfold) {
    // extra blocks (launch_apply_prologue_compact): basis compaction
    if (static_cast<int>(blockIdx.x) >= G) {
        basis_compact_one(basis, D, static_cast<int>(blockIdx.x) - G, threadIdx.x, nnz, rows, vals);
        return;
    }
    __builtin_amdgcn_s_setprio(3);     // see ffk_internal.h FFK_SMALL_KERNEL_PRIORITY
    __shared__ cplx E[2][D][D];
    __shared__ cplx M[D][D];
    __shared__ cplx V[D][D];
    __shared__ cplx Q[D][D];
    __shared__ cplx T[D][D];
    __shared__ cplx BV[D][D];
    constexpr int kBatch = 16;
    constexpr bool kTree = D <= 8;                 // a whole matrix fits one pass of the 64 lanes
    // the chunk totals: for the tree as many as the launch has chunks (dynamic LDS: 4 KiB at config 2;
    // a static array for the 64 chunks use_fused_front allows took 16 KiB, and a kernel with more
    // than 8 KiB of static LDS is not placed beside an accumulate block of another pass,
    // tools/corun.hip), else one batch
    extern __shared__ __attribute__((aligned(16))) unsigned char tot_raw[];
    cplx (*tot)[D][D] = reinterpret_cast<cplx (*)[D][D]>(tot_raw);
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int c = g / L;
    for (int e = lane; e < D*D; e += 64) E[0][e / D][e % D] = {(e / D == e % D) ? 1.0 : 0.0, 0.0};
    int b = 0;
    if constexpr (kTree) {
        // exclusive prefix E_c = T_{c-1} ... T_0 of the chunk totals as an ordered pairwise tree
        // (newer factor on the left), several products per pass: ceil(log2 c) dependent levels
        // instead of c - 1 dependent products.  In place: pass j writes slot j after reading slots
        // 2j and 2j+1, which no later pass of the level reads.
        constexpr int DD = D*D, PER = 64/DD;
        for (int e = lane; e < c*DD; e += 64) (&tot[0][0][0])[e] = totals[e];
        __syncthreads();
        int n = c;
        while (n > 1) {
            const int pairs = n >> 1;
            for (int j0 = 0; j0 < pairs; j0 += PER) {
                const int jj = j0 + lane/DD, e = lane % DD, i = e / D, k = e % D;
                const bool act = lane < PER*DD && jj < pairs;
                cplx acc = {0.0, 0.0};
                if (act) {
#pragma unroll
                    for (int x = 0; x < D; ++x) cmac(acc, tot[2*jj + 1][i][x], tot[2*jj][x][k]);
                }
                __syncthreads();
                if (act) tot[jj][i][k] = acc;
                __syncthreads();
            }
            if (n & 1) {                            // the newest factor moves up unpaired
                // (two doubles read by every lane from a clamped index: as a struct copy the value was kept in SCRATCH
                // memory across the barrier -- a trip to memory per odd level of the tree, in every pass; round 6)
                const cplx* from = &tot[n - 1][0][0] + (lane < DD ? lane : 0);
                const double carry_re = from->re, carry_im = from->im;      // (two doubles, not a struct copy)
                __syncthreads();
                if (lane < DD) (&tot[pairs][0][0])[lane] = {carry_re, carry_im};
                __syncthreads();
            }
            n = pairs + (n & 1);
        }
        if (c > 0) {
            for (int e = lane; e < DD; e += 64) E[0][e / D][e % D] = tot[0][e / D][e % D];
            __syncthreads();
        }
    } else {
    // exclusive prefix of the chunk totals; the totals are staged in batches of independent loads
    // (one dependent global load per step cost ~0.5 us each)
    for (int kb = 0; kb < c; kb += kBatch) {
        const int nb = min(kBatch, c - kb);
        __syncthreads();
        for (int e = lane; e < nb*D*D; e += 64)
            (&tot[0][0][0])[e] = totals[static_cast<size_t>(kb)*D*D + e];
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            for (int e = lane; e < D*D; e += 64) {
                const int i = e / D, j = e % D;
                cplx acc = {0.0, 0.0};
#pragma unroll
                for (int x = 0; x < D; ++x) cmac(acc, tot[k][i][x], E[b][x][j]);
                E[b ^ 1][i][j] = acc;
            }
            __syncthreads();
            b ^= 1;
        }
    }
    }
    // Q[g] (into LDS) and Q[g+1] (to memory)
    for (int e = lane; e < D*D; e += 64) M[e / D][e % D] = Qloc[static_cast<size_t>(g + 1)*D*D + e];
    if (g == 0)
        for (int e = lane; e < D*D; e += 64) Qout[e] = {(e / D == e % D) ? 1.0 : 0.0, 0.0};
    __syncthreads();
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, j = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int x = 0; x < D; ++x) cmac(acc, M[i][x], E[b][x][j]);
        Qout[static_cast<size_t>(g + 1)*D*D + e] = acc;
    }
    if (segtab == nullptr) return;
    __syncthreads();
    if (g % L == 0) {
        for (int e = lane; e < D*D; e += 64) Q[e / D][e % D] = E[b][e / D][e % D];
    } else {
        for (int e = lane; e < D*D; e += 64) M[e / D][e % D] = Qloc[static_cast<size_t>(g)*D*D + e];
        __syncthreads();
        for (int e = lane; e < D*D; e += 64) {
            const int i = e / D, j = e % D;
            cplx acc = {0.0, 0.0};
#pragma unroll
            for (int x = 0; x < D; ++x) cmac(acc, M[i][x], E[b][x][j]);
            Q[i][j] = acc;
        }
    }
    for (int e = lane; e < D*D; e += 64) V[e / D][e % D] = eigvecs[static_cast<size_t>(g)*D*D + e];
    __syncthreads();
    prologue_segment<D>(V, Q, T, BV, g, lane, eigvals, n_opers, n_coeffs, dt, t, G, A, segtab, Tc,
                        ops, nullptr, nullptr, with_noise_ops, wfold);
}

// out[g,k] = T_g C_k T_g^dag with T_g = conj(Tc[g])   (= (Q^dag V)^dag C_k (Q^dag V))
template <int D>
__global__ __launch_bounds__(64) void basis_transformed_kernel(const cplx* __restrict__ Tc,
                                                               const cplx* __restrict__ basis,
                                                               int N, cplx* __restrict__ out) {
    __shared__ cplx T[D][D];
    __shared__ cplx C[D][D];
    __shared__ cplx TC[D][D];
    const int g = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;   // g on x: G may exceed 65535
    for (int e = lane; e < D*D; e += 64) {
        const cplx v = Tc[static_cast<size_t>(g)*D*D + e];
        T[e / D][e % D] = {v.re, -v.im};
        C[e / D][e % D] = basis[static_cast<size_t>(k)*D*D + e];
    }
    __syncthreads();
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, j = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int a = 0; a < D; ++a) cmac(acc, T[i][a], C[a][j]);
        TC[i][j] = acc;
    }
    __syncthreads();
    for (int e = lane; e < D*D; e += 64) {
        const int i = e / D, j = e % D;
        cplx acc = {0.0, 0.0};
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const cplx tj = T[j][a];  // (T^dag)[a][j] = conj(T[j][a])
            cmac_conj(acc, tj, TC[i][a]);
        }
        out[(static_cast<size_t>(g)*N + k)*D*D + e] = acc;
    }
}

// phase_factors[g,w] = exp(i w t_g); integral[g,w,m,n] per numeric.py:144-167
__global__ void phase_integral_kernel(const double* __restrict__ omega, int W,
                                      const double* __restrict__ segtab, int d, int S,
                                      cplx* __restrict__ phase_factors,
                                      cplx* __restrict__ integral) {
    const int g = blockIdx.x;   // segments on x: G may exceed 65535
    const int w = blockIdx.y*blockDim.x + threadIdx.x;
    if (w >= W) return;
    const double* st = segtab + static_cast<size_t>(g)*S;
    const double om = omega[w];
    if (phase_factors) phase_factors[static_cast<size_t>(g)*W + w] = cexp(om*st[1]);
    if (integral) {
        cplx* out = integral + (static_cast<size_t>(g)*W + w)*d*d;
        for (int e = 0; e < d*d; ++e) out[e] = first_order_integral(om, st[seg_rec(e)], st[0]);
    }
}

}  // namespace

// d = 8: W'_a[m][n][i] = Bbar_a[m][n] e^{i b_mn} conj(T[m][i]), the frequency-independent A operand of the accumulate
// kernel's first product (ctrl_pcr.hip), once per (segment, operator) instead of once per 64-frequency block: the
// kernel's producers then copy it into their tiles by LDS-DMA.  Thread l <-> (m, n) = (l >> 3, l & 7); the products,
// their order and the layout [(s, ng, ig)][consumer lane] are those of the producers' own fold (same bits).
__global__ __launch_bounds__(64) void fold_w8_kernel(const double* __restrict__ segtab, const cplx* __restrict__ ops,
                                                     int A, cplx* __restrict__ wfold) {
    constexpr int D = 8, DD = 64, S = seg_stride(8);
    const int g = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*DD;
    const double* r = segtab + static_cast<size_t>(g)*S + seg_rec(lane);
    const cplx x = src[static_cast<size_t>(1 + a)*DD + lane];
    const double sb = r[1], cb = r[2];
    const int m = lane >> 3, n = lane & 7;
    const int base = ((m >> 2)*2 + (n >> 2))*2*64 + 4*(n & 3) + 16*(m & 3);
    const cplx bt = cmul(x, cplx{cb, sb});
    cplx* out = wfold + (static_cast<size_t>(g)*A + a)*512;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const cplx t = src[m*D + i];
        out[base + (i >> 2)*64 + (i & 3)] = cmul(bt, cplx{t.re, -t.im});
    }
}

hipError_t launch_fold_w8(const double* segtab, const cplx* ops, int G, int A, cplx* wfold, hipStream_t stream) {
    hipLaunchKernelGGL(fold_w8_kernel, dim3(G, A), dim3(64), 0, stream, segtab, ops, A, wfold);
    return hipGetLastError();
}

hipError_t launch_prologue(const double* eigvals, const cplx* eigvecs, const cplx* propagators,
                           const cplx* n_opers, const double* n_coeffs, const double* dt,
                           const double* t, int G, int d, int A, double* segtab, cplx* Tc,
                           cplx* ops, cplx* n_opers_transformed, cplx* eigvecs_propagated,
                           hipStream_t stream, cplx* wfold) {
    if (generic_dimension(d))
        return launch_prologue_generic(eigvals, eigvecs, propagators, n_opers, n_coeffs, dt, t, G, d, A, segtab,
                                       Tc, ops, n_opers_transformed, eigvecs_propagated, stream);
    switch (d) {
#define FFK_CASE(D)                                                                             \
    case D:                                                                                     \
        hipLaunchKernelGGL(prologue_kernel<D>, dim3(G), dim3(64), 0, stream, eigvals, eigvecs,  \
                           propagators, n_opers, n_coeffs, dt, t, G, A, segtab, Tc, ops,        \
                           n_opers_transformed, eigvecs_propagated, !split_noise_ops(D, A),     \
                           D == 4 ? wfold : nullptr);                                           \
        if (split_noise_ops(D, A))                                                              \
            hipLaunchKernelGGL(noise_ops_kernel<D>, dim3(G, A), dim3(64), 0, stream, eigvecs,   \
                               n_opers, n_coeffs, G, A, ops, n_opers_transformed);              \
        break;
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
    if (d == 8 && wfold != nullptr) return launch_fold_w8(segtab, ops, G, A, wfold, stream);
    return hipGetLastError();
}

hipError_t launch_apply_prologue_compact(const cplx* Qloc, const cplx* totals, int G, int d, cplx* Q,
                                         const double* eigvals, const cplx* eigvecs,
                                         const cplx* n_opers, const double* n_coeffs,
                                         const double* dt, const double* t, int A, double* segtab,
                                         cplx* Tc, cplx* ops, const cplx* basis, int N, void* ews,
                                         hipStream_t stream, cplx* wfold) {
    const int L = front_chunk(d);
    int* nnz = nullptr;
    int* rows = nullptr;
    cplx* vals = nullptr;
    int extra = 0;
    if (basis) {
        expand_workspace_slices(ews, N, d, &nnz, &rows, &vals);
        extra = N;
    }
    switch (d) {
#define FFK_CASE(D)                                                                              \
    case D:                                                                                      \
        if (D > 8) {   /* (static 7 D^2 + dynamic 16 D^2 complex numbers: beyond the default limit) */ \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(apply_prologue_kernel<D>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                               static_cast<int>(sizeof(cplx)*D*D*16));           \
            if (e != hipSuccess) return e;                                                       \
        }                                                                                        \
        hipLaunchKernelGGL(apply_prologue_kernel<D>, dim3(G + extra), dim3(64),                  \
                           sizeof(cplx)*D*D*(D <= 8 ? (G + L - 1)/L : 16), stream, Qloc,         \
                           totals, G, L, Q, eigvals, eigvecs, n_opers, n_coeffs, dt, t, A,       \
                           segtab, Tc, ops, basis, nnz, rows, vals, !split_noise_ops(D, A),      \
                           D == 4 ? wfold : nullptr);                                            \
        if (split_noise_ops(D, A))                                                               \
            hipLaunchKernelGGL(noise_ops_kernel<D>, dim3(G, A), dim3(64), 0, stream, eigvecs,    \
                               n_opers, n_coeffs, G, A, ops, nullptr);                           \
        break;
        FFK_CASE(2) FFK_CASE(3) FFK_CASE(4) FFK_CASE(5) FFK_CASE(6) FFK_CASE(7) FFK_CASE(8)
        FFK_CASE(9) FFK_CASE(10) FFK_CASE(11) FFK_CASE(12) FFK_CASE(13) FFK_CASE(14)
        FFK_CASE(15) FFK_CASE(16)
#undef FFK_CASE
        default:
            return hipErrorInvalidValue;
    }
    if (d == 8 && wfold != nullptr) return launch_fold_w8(segtab, ops, G, A, wfold, stream);
    return hipGetLastError();
}

hipError_t launch_apply_prologue(const cplx* Qloc, const cplx* totals, int G, int d, cplx* Q,
                                 const double* eigvals, const cplx* eigvecs, const cplx* n_opers,
                                 const double* n_coeffs, const double* dt, const double* t, int A,
                                 double* segtab, cplx* Tc, cplx* ops, hipStream_t stream) {
    return launch_apply_prologue_compact(Qloc, totals, G, d, Q, eigvals, eigvecs, n_opers, n_coeffs,
                                         dt, t, A, segtab, Tc, ops, nullptr, 0, nullptr, stream);
}

hipError_t launch_basis_transformed(const cplx* Tc, const cplx* basis, int G, int N, int d,
                                    cplx* out, hipStream_t stream) {
    if (N > 65535) return hipErrorInvalidValue;
    switch (d) {
#define FFK_CASE(D)                                                                            \
    case D:                                                                                    \
        hipLaunchKernelGGL(basis_transformed_kernel<D>, dim3(G, N), dim3(64), 0, stream, Tc,   \
                           basis, N, out);                                                     \
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

hipError_t launch_phase_and_integral(const double* omega, int W, const double* segtab, int G,
                                     int d, cplx* phase_factors, cplx* integral,
                                     hipStream_t stream) {
    if (!phase_factors && !integral) return hipSuccess;
    const int block = 128;
    if ((W + block - 1)/block > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(phase_integral_kernel, dim3(G, (W + block - 1)/block), dim3(block), 0,
                       stream, omega, W, segtab, d, seg_stride(d), phase_factors, integral);
    return hipGetLastError();
}

}  // namespace ffk
