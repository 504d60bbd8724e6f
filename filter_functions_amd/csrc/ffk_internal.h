// ffk_internal.h -- launcher declarations shared between the kernel translation units and the
// C-ABI layer (ffk_api*.hip).  Everything here is device-pointer based and asynchronous on the
// given stream; nothing allocates.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "ffk_math.h"

namespace ffk {

constexpr int kMaxD = 16;          // compile-time-D kernels (registers / wave-private LDS)
constexpr int kMaxDGeneric = 64;   // runtime-d kernels of generic.hip (FFK_MAX_D)
constexpr int kWave = 64;

// Per-segment uniform table row (doubles): [0] = dt_g, [1] = t_g, [2..3] pad, then one 4-double
// record per matrix entry e = m*d + n:  (dE = D_m - D_n, sin b, cos b, 0) with b = fl(dE*dt)/2,
// the frequency-independent half-angle of first_order_integral_aa.  A record is one 32-byte LDS
// read.  Rows are padded to a multiple of 8 doubles (64-byte aligned).
// d > kMaxD (generic.hip): [0] = dt_g, [1] = t_g, [2..3] pad, then the d eigenvalues (the generic
// accumulate kernel evaluates the integral entries directly and needs no per-entry records).
__host__ __device__ constexpr int seg_stride(int d) {
    return d > kMaxD ? ((4 + d + 7)/8)*8 : ((4 + 4*d*d + 7)/8)*8;
}
__host__ __device__ constexpr int seg_rec(int e) { return 4 + 4*e; }

// Columns of the Hilbert-space accumulator kept per thread in ctrl_accumulate.
__host__ __device__ constexpr int accum_jb(int d) {
#if defined(FFK_JB4)  /* tuning builds */
    if (d == 4) return FFK_JB4;
#endif
#if defined(FFK_JB8)
    if (d == 8) return FFK_JB8;
#endif
#if defined(FFK_JB16)
    if (d == 16) return FFK_JB16;
#endif
    // largest column block whose accumulators + row temporaries stay in registers (no scratch)
    // at the 2-waves/SIMD budget: survey in profiles/r01_e_register_survey.txt
    return d <= 5 ? d : (d == 6 ? 3 : ((d == 8 || d == 10) ? 2 : 1));
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1)/a*a; }

// message behind ffk_last_error() (thread local; ffk_api.hip)
void set_last_error(const char* message);

// ---- eigh.hip --------------------------------------------------------------------------------
// eigvals (G,d), eigvecs (G,d,d), seg_prop (G,d,d) = V exp(-i D dt) V^dag; status (G) ints:
// 1 for every segment whose Jacobi iteration failed to converge, else 0.
// fail_count (optional, compiled dimensions only: eigh_fail_count_supported): a counter in memory the kernel can
// reach with a system-scope atomic, incremented per flagged segment (the caller zeroes it).
hipError_t launch_eigh_expm(const cplx* H, const double* dt, int G, int d, double* eigvals,
                            cplx* eigvecs, cplx* seg_prop, int* status, hipStream_t stream,
                            int* fail_count = nullptr);
bool eigh_fail_count_supported(int d);
// the same from control operators (n_c, d, d) and amplitudes (n_c, G): H[g] = sum_i coeffs[i, g] opers[i] is formed
// in the kernel (compiled dimensions only)
hipError_t launch_eigh_expm_controls(const cplx* opers, const double* coeffs, int n_c, const double* dt, int G,
                                     int d, double* eigvals, cplx* eigvecs, cplx* seg_prop, int* status,
                                     hipStream_t stream, int* fail_count = nullptr);
// out[0] = number of non-zero entries of status (G): the device-resident paths' convergence check
hipError_t launch_count_failures(const int* status, int G, int32_t* out, hipStream_t stream);
// Chunk length of the two-kernel scan used by the fused front end (scan_local + fix-up fused
// with the prologue): both serial parts are ~sqrt(G) long at G = 256.
// FFK_SMALL_KERNEL_PRIORITY: the latency-bound kernels of a pass (eigensolver, scans, prologue,
// expansion, integral: a few hundred waves) raise their issue priority with s_setprio 3 on entry.
// Passes are kept in flight side by side so that these kernels run beside another pass's
// accumulate kernel -- but the SIMD's arbiter serves the OLDEST ready wave first, and an FMA-dense
// kernel always has one: at equal priority a younger wave on the same SIMD does not get a single
// issue slot until the dense kernel retires (tools/corun.hip: a 64-thread kernel launched beside a
// 16-wave FMA loop finished 3 us after the loop ended, 230 us later; with s_setprio 3 after 7 us).
__host__ __device__ constexpr int front_chunk(int d) { return d <= 8 ? 16 : 8; }
// The fused front end is used while the serial part of its second kernel stays short.
inline bool use_fused_front(int G, int d) {
    return d <= kMaxD && (G + front_chunk(d) - 1)/front_chunk(d) <= 64;
}

// ---- liouville.hip / generic.hip: rows of the Liouville GEMM's operands for a Hermitian basis -----
// tr(CB_i C_j) with CB_i = U^dag C_i U and C_j both Hermitian (CB_i is, whatever U):
//   sum_a Re CB[a,a] Re C[a,a] + sum_{a<b} 2 (Re CB[a,b] Re C[b,a] - Im CB[a,b] Im C[b,a]),
// i.e. K = d^2 real rows instead of 2 d^2: row a for the diagonal, rows d + 2 p(a,b) (real parts) and
// d + 2 p(a,b) + 1 (imaginary parts) for the pair a < b, p(a,b) = a (2d - a - 1)/2 + b - a - 1.
// Returns -1 for the entries that are not stored (a > b; the imaginary part of the diagonal).
__host__ __device__ inline int hermitian_operand_row(int a, int b, int imag_part, int d) {
    if (a == b) return imag_part ? -1 : a;
    if (a > b) return -1;
    return d + 2*(a*(2*d - a - 1)/2 + b - a - 1) + imag_part;
}
__host__ __device__ inline int liouville_operand_rows(int d, int want_imag) {
    return ((want_imag ? 2*d*d : d*d) + 3)/4*4;       // padded to the MFMA k-step
}

// ---- scan.hip --------------------------------------------------------------------------------
size_t scan_workspace_bytes(int G, int d);
// chunk-local prefix products Qloc (G+1,d,d) (Qloc[0] = 1) and chunk totals (nchunks,d,d)
hipError_t launch_scan_local(const cplx* seg_prop, int G, int d, int L, cplx* Qloc, cplx* totals,
                             hipStream_t stream);
// Q (G+1,d,d): Q[0] = 1, Q[g+1] = P[g] Q[g]
hipError_t launch_prefix_products(const cplx* seg_prop, int G, int d, cplx* Q, void* ws,
                                  hipStream_t stream);

// ---- prep.hip --------------------------------------------------------------------------------
// Fills the omega-independent operands of the accumulate kernel:
//   segtab (G, seg_stride(d)); Tc (G,d,d) = conj(V^dag Q_g); ops (G, 1+A, d, d) with
//   ops[g,0] = T_g = V_g^dag Q_g and ops[g,1+a] = Bbar_a^(g) = s_a(g) V_g^dag B_a V_g.
// Optional reference intermediates (may be NULL): n_opers_transformed (A,G,d,d),
// eigvecs_propagated (G,d,d) = Q_g^dag V_g.
hipError_t launch_prologue(const double* eigvals, const cplx* eigvecs, const cplx* propagators,
                           const cplx* n_opers, const double* n_coeffs, const double* dt,
                           const double* t, int G, int d, int A, double* segtab, cplx* Tc,
                           cplx* ops, cplx* n_opers_transformed, cplx* eigvecs_propagated,
                           hipStream_t stream, cplx* wfold = nullptr);
// wfold (d = 4 and d = 8, may be NULL): the frequency-independent operand of the accumulate kernel's first product per
// (segment, operator), in the order the kernel's tile holds it, wfold_elems(d, G, A, W, chunks) complex numbers in all -- d = 4:
// W_a[n][m][j] = Bbar_a[m][n] e^{i b_mn} T[n][j] (64 per segment and operator); d = 8: W'_a[m][n][i] = Bbar_a[m][n]
// e^{i b_mn} conj(T[m][i]) (512, ctrl_pcr.hip).  It
// does not depend on the frequency: a caller that owns a buffer for it (ffk_control_matrix_dev, ffk_pipeline_dev)
// has the prologue kernels write it once per segment and hands it to launch_accumulate, whose producers then copy
// it into their tiles instead of each of the W/64 frequency blocks folding it again (same products in the same
// order: same bits).  NULL: the accumulate kernel folds it itself.
// Fused scan fix-up (+ prologue): every block rebuilds the exclusive chunk prefix E_c from
// `totals`, writes Q[g+1] = Qloc[g+1] E_c (Q[0] = 1 by block 0) and, if segtab != NULL, runs the
// prologue for its segment with Q[g] without a round trip through memory.
hipError_t launch_apply_prologue(const cplx* Qloc, const cplx* totals, int G, int d, cplx* Q,
                                 const double* eigvals, const cplx* eigvecs, const cplx* n_opers,
                                 const double* n_coeffs, const double* dt, const double* t, int A,
                                 double* segtab, cplx* Tc, cplx* ops, hipStream_t stream);
// The same launch with N extra blocks that compact the basis into the expansion workspace `ews`
// (what launch_expand_chunks needs) -- no separate launch for it.
hipError_t launch_apply_prologue_compact(const cplx* Qloc, const cplx* totals, int G, int d, cplx* Q,
                                         const double* eigvals, const cplx* eigvecs,
                                         const cplx* n_opers, const double* n_coeffs,
                                         const double* dt, const double* t, int A, double* segtab,
                                         cplx* Tc, cplx* ops, const cplx* basis, int N, void* ews,
                                         hipStream_t stream, cplx* wfold = nullptr);
// basis_transformed (G,N,d,d) = (Q^dag V)^dag C_k (Q^dag V)   (numeric.py:863-864)
hipError_t launch_basis_transformed(const cplx* Tc, const cplx* basis, int G, int N, int d,
                                    cplx* out, hipStream_t stream);
// phase_factors (G,W) and first_order_integral (G,W,d,d)  (numeric.py:865-866)
hipError_t launch_phase_and_integral(const double* omega, int W, const double* segtab, int G,
                                     int d, cplx* phase_factors, cplx* integral,
                                     hipStream_t stream);

// ---- ctrl.hip --------------------------------------------------------------------------------
struct AccumGeometry {
    int chunks;       // segment chunks (grid.z)
    int chunk_len;    // segments per chunk
    int nwaves;       // waves per block
    int task_groups;  // grid.y
    int lds_bytes;
    int nbuf;
    int na_blk;       // noise operators whose Bbar one block stages in LDS
    bool wave_kernel; // small-d one-wave-per-block variant
    int gsplit;       // sub-chunks per block (in-block segment split), 1 = none
    bool mfma;        // large-d matrix-core kernel (ctrl_mfma.hip): 16 frequencies per block
    bool pc;          // producer/consumer kernel with the second product on the matrix cores (ctrl_pq.hip), d = 4
    bool pcw;         // producer/consumer kernel on the matrix cores (ctrl_pcr.hip), d = 8
    bool generic;     // runtime-d kernel (generic.hip), d > 16: one block per (frequency, operator, chunk)
    bool d2 = false;  // folded-operand kernel for d = 2 (ctrl_d2.hip): independent wavefronts, one partial sum per block
};
void set_use_wave_kernel(bool on);
void set_use_gsplit(bool on);
void set_mfma_policy(int policy);   // 0 default (d >= 12), 1 never, 2 wherever supported (d = 8 too)
AccumGeometry accumulate_geometry(int W, int A, int G, int d, int forced_chunks);
// Ypart (chunks, A, d, d, W): partial Hilbert-space sums, omega fastest
// `expand` (optional): with ONE segment chunk the block that owns an (operator, frequency tile) holds
// the complete Y of it when its last segment is done; a kernel that supports it (the d = 12, 16
// matrix-core kernel) then expands Y in the basis from LDS and writes the control matrix R (A, N, W)
// INSTEAD of Ypart -- the expansion launch and the round trip of Y through HBM disappear (config 5:
// 1.2 GB written and read back).  *expanded is set to whether that happened.
struct ExpandEpilogue {
    const int* nnz;        // compacted basis lists (post.hip: expand_workspace_slices)
    const int* rows;
    const cplx* vals;
    int N;
    cplx* R;               // NULL: no epilogue
};
hipError_t launch_accumulate(const double* omega, int W, const double* segtab, const cplx* ops,
                             int G, int d, int A, const AccumGeometry& geo, cplx* Ypart,
                             hipStream_t stream, const ExpandEpilogue* expand = nullptr,
                             bool* expanded = nullptr, const cplx* wfold = nullptr);

// ---- ctrl_mfma.hip ---------------------------------------------------------------------------
int device_cu_count();   // compute units of the current device (ctrl.hip)
bool mfma_accumulate_supported(int d);
int mfma_accumulate_waves(int d, int A);
int mfma_accumulate_ops_per_block(int d, int A);
int mfma_accumulate_lds_bytes(int d, int nw);
hipError_t launch_accumulate_mfma(const double* omega, int W, const double* segtab, const cplx* ops,
                                  int G, int d, int A, int chunks, int chunk_len, int nw,
                                  cplx* Ypart, hipStream_t stream,
                                  const ExpandEpilogue* expand = nullptr, bool* expanded = nullptr);

// ---- sticky fault words (ffk_api.hip) ----------------------------------------------------------
// Device pointer of the CALLING THREAD's int in mapped pinned host memory (one block of words per process,
// created on first use -- never inside a stream capture: ffk_graph_capture_begin and the resident pass touch it
// first; portable memory, so the pointer is valid on every device).  Launchers pass it to their kernels as an
// ARGUMENT; a kernel whose bounded flag wait runs out stores a non-zero code there.  One word per host thread: a
// thread reads only the faults of launches it enqueued itself.  NULL only if the runtime refused the allocation.
int* kernel_fault_word();
constexpr int kFaultPcProducerWait = 1, kFaultPcConsumerWait = 2;
inline int kernel_fault_code_for_selftest() { return kFaultPcConsumerWait; }

// ---- ctrl_pq.hip (d = 4, second product on the matrix cores) -------------------------------------
int pq_accumulate_lds_bytes(int nc);
int pq_accumulate_waves(int nc);
bool pq_accumulate_supported(int d, int A);
struct PqGroups {
    int n3, n2, n1;     // blocks of three, two, one operator(s): 3 n3 + 2 n2 + n1 = A, one launch per size
};
PqGroups pq_accumulate_groups(int A);
// Dimensions between the specialised kernels run PADDED with decoupled levels on the next specialised kernel
// (7 -> 8: ctrl_pcr.hip; 11 -> 12 and 13, 14, 15 -> 16: ctrl_mfma.hip): operands T (+) 1 and Bbar (+) 0, the
// d x d block of Y copied out (ctrl.hip: launch_accumulate).  0: d is not padded (d = 5, 6, 9, 10 do not gain: d = 10 on the d = 12
// kernel measured 577 against 517 us; profiles/r06_p_*).
constexpr int padded_dimension(int d) {
    return d == 7 ? 8 : (d == 11 ? 12 : ((d >= 13 && d <= 15) ? 16 : 0));
}
// (small problems lose to the dimension's own kernel -- d = 7: 100 segments x 2 operators x 1000 frequencies 115
// against 95 us, d = 11: 20 x 2 x 200 61 against 54 us; d = 13-15 gain at every size tried: profiles/r06_p_*)
inline bool padded_launch_pays(int d, int G, int W, int A) {
    const double work = static_cast<double>(G)*W*A;
    return padded_dimension(d) != 0 && (d == 7 ? work >= 1.0e6 : (d == 11 ? work >= 5.0e4 : true));
}
// Complex numbers of scratch the accumulate launch wants beside its operands (`wfold`): the folded operand of the
// d = 4 / d = 8 kernels; for a padded dimension the padded operands, table rows and partial sums (and the d = 8
// kernel's folded operand).
constexpr size_t wfold_elems(int d, int G, int A, int W, int chunks) {
    if (d == 4) return static_cast<size_t>(G)*A*64;
    if (d == 8) return static_cast<size_t>(G)*A*512;
    const int p = padded_dimension(d);
    if (p == 0) return 0;
    return static_cast<size_t>(G)*(1 + A)*p*p + static_cast<size_t>(G)*seg_stride(p)/2 +
           static_cast<size_t>(chunks)*A*p*p*W + (p == 8 ? static_cast<size_t>(G)*A*512 : 0);
}
hipError_t launch_accumulate_pq(const double* omega, int W, const double* segtab, const cplx* ops,
                                int G, int d, int A, int chunks, int chunk_len, cplx* Ypart,
                                const cplx* wfold, hipStream_t stream);

// ---- ctrl_d2.hip (d = 2, folded operands) ------------------------------------------------------
bool d2_accumulate_supported(int d);
int d2_accumulate_waves();
int d2_accumulate_freqs_per_block();
int d2_accumulate_ops_per_block(int A);
int d2_accumulate_lds_bytes(int ops_per_block);
hipError_t launch_accumulate_d2(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                                int chunks, int chunk_len, cplx* Ypart, hipStream_t stream);

// ---- ctrl_pcr.hip (d = 8, real integral tile) --------------------------------------------------
bool pcr_accumulate_supported(int d, int A);
int pcr_accumulate_ops_per_block();
int pcr_accumulate_waves();
int pcr_accumulate_lds_bytes();
// d = 8: fills wfold from the prologue's outputs (segtab, ops): one 64-thread block per (segment, operator)
hipError_t launch_fold_w8(const double* segtab, const cplx* ops, int G, int A, cplx* wfold, hipStream_t stream);
hipError_t launch_accumulate_pcr(const double* omega, int W, const double* segtab, const cplx* ops,
                                 int G, int d, int A, int chunks, int chunk_len, cplx* Ypart,
                                 const cplx* wfold, hipStream_t stream);

// ---- post.hip --------------------------------------------------------------------------------
// Bt (A,d,d,W) = sum over chunks of Ypart
hipError_t launch_reduce_chunks(const cplx* Ypart, int chunks, size_t slab, cplx* Bt,
                                hipStream_t stream);
// R (A2,N,W) with R[a,k,w] = sum_ij Bt[a,i,j,w] C_k[j,i];  A2 = any leading batch
size_t expand_workspace_bytes(int N, int d);
void expand_workspace_slices(void* ws, int N, int d, int** nnz, int** rows, cplx** vals);
// R from the chunk partials directly (basis lists already compacted in ws)
hipError_t launch_expand_chunks(const cplx* Ypart, int chunks, size_t slab, int A, int N, int d,
                                int W, cplx* R, void* ws, hipStream_t stream);
// ... and the fidelity filter function F (A,A,W) in the same launch (small A*N)
bool expand_ff_supported(int A, int N);
hipError_t launch_expand_ff(const cplx* Ypart, int chunks, size_t slab, int A, int N, int d, int W,
                            cplx* R, cplx* F, void* ws, hipStream_t stream);
// `compacted`: the basis lists in ws were already produced by launch_reduce_and_compact
hipError_t launch_expand(const cplx* Bt, const cplx* basis, int A2, int N, int d, int W, cplx* R,
                         void* ws, bool compacted, hipStream_t stream);
// reduce_chunks + basis compaction in one launch
hipError_t launch_reduce_and_compact(const cplx* Ypart, int chunks, size_t slab, cplx* Bt,
                                     const cplx* basis, int N, int d, void* ws, hipStream_t stream);
// out (W,A,d,d) from Bt (A,d,d,W)
hipError_t launch_transpose_noise_ops(const cplx* Bt, int A, int d, int W, cplx* out,
                                      hipStream_t stream);
hipError_t launch_filter_function(const cplx* R, int A, int N, int W, int which, cplx* F,
                                  hipStream_t stream);
hipError_t launch_filter_function_weighted(const cplx* R, int A, int N, int W, const cplx* M,
                                           double scale, cplx* F, hipStream_t stream);
size_t infidelity_workspace_bytes(int W, int n_idx, int s_ndim);
// spectrum_on_host: S (and idx) live in mapped pinned HOST memory: the spectrum is staged through LDS with all of
// a thread's reads in flight at once (every read is a trip over PCIe); same sums in the same order.
hipError_t launch_infidelity(const cplx* F, int A, int W, const cplx* S, int s_ndim,
                             const double* omega, const int32_t* idx, int n_idx, int d,
                             int shard_width, double* infid, void* ws, hipStream_t stream,
                             bool spectrum_on_host = false);

// ---- atomic.hip -----------------------------------------------------------------------------
size_t from_atomic_workspace_bytes(int G, int A, int N, int W);
// out: (A,N,W) for the sum, (G,A,N,W) for correlations != 0; L is (G-1,N,N) f64 or c128.
// index == NULL: phases (G-1,W) cumulated, Ratomic (G,A,N,W).  index != NULL (G int32): phases
// (T,W) total phase factors and Ratomic (T,A,N,W) tables over the distinct pulses.
// Rtab (indexed form only, may be NULL): T device pointers to the distinct control matrices where
// they lie, instead of the contiguous table Ratomic.  F (may be NULL; sum only): the fidelity filter
// function (A,A,W) of the result -- for A N <= 16 formed by the slab-reduction launch itself.
hipError_t launch_from_atomic(const cplx* phases, const cplx* Ratomic, const int32_t* index,
                              const double* L, int l_is_complex, int G, int A, int N, int W,
                              int correlations, cplx* out, void* ws, hipStream_t stream,
                              const cplx* const* Rtab = nullptr, cplx* F = nullptr, int T = 0,
                              const double* Lpulse = nullptr);
// Lpulse (optional, with index and T): the Liouville representations (T, N, N) of the T distinct pulses' own total
// propagators (f64, or c128 with l_is_complex): the single-qubit block kernel then walks a slab by a backward
// recurrence on these instead of reading one cumulative propagator per position (atomic.hip)
// (T: number of distinct pulses of the indexed form; with it, few rows and tables that fit LDS the
// rule, its slab reduction and F are ONE launch, from_atomic_block_kernel)
// The front of a sequence concatenation in one launch (d <= 4, G <= 1024, N <= 16): Q (G+1,d,d)
// running products of U[index[g]], L (G-1,N,N) their Liouville representations, phases (T,W) =
// exp(i omega tau_k), and optionally a copy of omega.
bool sequence_front_supported(int d, int G, int N);
hipError_t launch_sequence_front(const cplx* U, const int32_t* index, int G, int d, const cplx* basis,
                                 int N, int l_is_complex, cplx* Q, double* L, const double* tau,
                                 const double* omega, int T, int W, cplx* phases, double* omega_copy,
                                 hipStream_t stream, double* Lpulse = nullptr);

// ---- decay.hip -------------------------------------------------------------------------------
// Gamma (Gp,Gp,n_idx[,n_idx],N,N) f64 from R (Gp,A,N,W) c128 (Gp = 1: the total control matrix),
// S c128 (W,), (n_idx,W) or (n_idx,n_idx,W); omega (Wg,) is the global grid of which R and S hold
// the block [w_offset, w_offset + W)
size_t decay_amplitudes_workspace_bytes(int Gp, int N, int W, int n_idx, int s_ndim);
hipError_t launch_decay_amplitudes(const cplx* R, int Gp, int A, int N, int W, const cplx* S,
                                   int s_ndim, const double* omega, int Wg, int w_offset,
                                   const int32_t* idx, int n_idx, double* gamma, void* ws,
                                   hipStream_t stream);
// K (batch,N,N) f64 from Gamma (batch,N,N) f64 and the basis (N,d,d)
size_t cumulant_workspace_bytes(size_t batch, int N, int d);
hipError_t launch_cumulant_function(const double* gamma, size_t batch, int N, int d,
                                    const cplx* basis, int single_qubit, double* K, void* ws,
                                    hipStream_t stream);

// scale (rows,W) = trapezoid weight(w_offset + w) S[row,w] / 2 pi over the global grid omega (Wg,)
hipError_t launch_spectral_weights(const cplx* S, int rows, int W, const double* omega, int Wg,
                                   int w_offset, cplx* scale, hipStream_t stream);

// ---- second.hip ------------------------------------------------------------------------------
// F2 (A,A,N,N,W) from nt = n_opers_transformed (A,G,d,d) and bt = basis_transformed (G,N,d,d)
size_t second_order_workspace_bytes(int G, int A, int N, int d);
hipError_t launch_second_order_filter_function(const double* omega, int W, const double* eigvals,
                                               const double* dt, const double* t, const cplx* nt,
                                               const cplx* bt, int G, int d, int A, int N, cplx* F2,
                                               void* ws, hipStream_t stream);
// concatenation rule: F2_atomic (G,A,A,N,N,W), step (G,A,N,W) = summands of the sequence's control
// matrix, L (G-1,N,N) f64 = Liouville matrices of the cumulative propagators -> out (A,A,N,N,W)
size_t periodic_workspace_bytes(int A, int N, int W);
hipError_t launch_periodic(const cplx* phases, const cplx* R1, const double* L, int l_is_complex,
                           int repeats, int A, int N, int W, cplx* out, void* ws, hipStream_t stream);
size_t second_order_from_atomic_workspace_bytes(int G, int A, int N, int W);
hipError_t launch_second_order_from_atomic(const cplx* F2_atomic, const cplx* step, const double* L,
                                           int G, int A, int N, int W, cplx* out, void* ws,
                                           hipStream_t stream);
// Delta (n_idx[,n_idx],N,N) = sum_w Re(F2[idx,idx] scale); scale from launch_spectral_weights
hipError_t launch_frequency_shifts(const cplx* F2, int A, int N, int W, const cplx* scale,
                                   int s_ndim, const int32_t* idx, int n_idx, double* out,
                                   hipStream_t stream);
// K (batch,N,N) += second-order contribution of Delta (batch,N,N)
size_t cumulant_second_order_workspace_bytes(size_t batch, int N, int d);
hipError_t launch_cumulant_second_order(const double* delta, size_t batch, int N, int d,
                                        const cplx* basis, double* K, void* ws, hipStream_t stream);

// in-place inclusive prefix sum over the G leading slabs of Y (G, slab)
hipError_t launch_segment_prefix_sum(cplx* Y, int G, size_t slab, hipStream_t stream);

// ---- grad.hip --------------------------------------------------------------------------------
// dF (A,G,H,W) f64 from ops (G,1+A,d,d) [T_s, Bbar], abar (H,G,d,d) = V^dag A_h V, Ycum (G,A,d,d,W)
// = prefix sums of the Hilbert-space steps, ratio (A,H,G) = n'_ahs/n_as or NULL; E (H,G,d,d) scratch
hipError_t launch_filter_function_derivative(const double* omega, int W, const double* eigvals,
                                             const double* dt, const double* t, const cplx* ops,
                                             const cplx* abar, const cplx* Ycum, const double* ratio,
                                             int G, int d, int A, int H, cplx* E, double* out,
                                             hipStream_t stream);
// dR (H,W,G,A,N) c128 = d R_ak / d u_h(t_s), same operands plus the basis (N,d,d)
hipError_t launch_control_matrix_derivative(const double* omega, int W, const double* eigvals,
                                            const double* dt, const double* t, const cplx* ops,
                                            const cplx* abar, const cplx* Ycum, const double* ratio,
                                            const cplx* basis, int N, int G, int d, int A, int H, cplx* E,
                                            cplx* out, hipStream_t stream);
// dF (A,G,H,W) = 2 Re sum_k conj(R[a,k,w]) dR[h,w,s,a,k]
hipError_t launch_filter_function_derivative_from_control_matrix(const cplx* R, const cplx* dR, int A,
                                                                 int N, int W, int G, int H, double* out,
                                                                 hipStream_t stream);
// out (A,G,H) = sum_w dF Re(scale)/d, scale from launch_spectral_weights with rows = 1 or A
hipError_t launch_infidelity_derivative(const double* dF, int A, int G, int H, int W, const cplx* scale,
                                        int s_ndim, int d, double* out, hipStream_t stream);

// B (W,A,d,d) = B^(0) + sum_g phases[g-1] P_{g-1}^dag B^(g) P_{g-1}; atomic (G,W,A,d,d), props (G-1,d,d)
hipError_t launch_noise_ops_from_atomic(const cplx* phases, const cplx* atomic, const cplx* props,
                                        int G, int W, int A, int d, cplx* out, hipStream_t stream);

// sum = K summed over its leading axis (batch, N, N) -> (N, N); norm_and_bad[0] = |sum|_1 (a double), [1] = count
// of NaN / Inf entries (a 64-bit integer; columns holding one do not enter the norm)
hipError_t launch_sum_and_one_norm(const double* K, int batch, int N, double* sum, double* norm_and_bad,
                                   hipStream_t stream);
// exp of a real N x N matrix (device pointers; w: five N*N scratch matrices); see decay.hip
hipError_t launch_expm_real(const double* A, int N, int squarings, double* out, double* const w[5],
                            hipStream_t stream);

// ---- generic.hip (17 <= d <= kMaxDGeneric) -----------------------------------------------------
bool generic_dimension(int d);
hipError_t launch_eigh_expm_generic(const cplx* H, const double* dt, int G, int d, double* eigvals,
                                    cplx* eigvecs, cplx* seg_prop, int* status, hipStream_t stream);
hipError_t launch_prefix_products_generic(const cplx* seg_prop, int G, int d, cplx* Q, hipStream_t stream);
hipError_t launch_prologue_generic(const double* eigvals, const cplx* eigvecs, const cplx* propagators,
                                   const cplx* n_opers, const double* n_coeffs, const double* dt,
                                   const double* t, int G, int d, int A, double* segtab, cplx* Tc, cplx* ops,
                                   cplx* n_opers_transformed, cplx* eigvecs_propagated,
                                   hipStream_t stream);
hipError_t launch_accumulate_generic(const double* omega, int W, const double* segtab, const cplx* ops,
                                     int G, int d, int A, int chunks, int chunk_len, cplx* Ypart,
                                     hipStream_t stream);
// U^dag C_i U into the Liouville GEMM's K-major operands (liouville.hip)
hipError_t launch_conjugate_basis_generic(const cplx* U, int batch, int d, const cplx* basis, int N,
                                          int Npad, int K, int want_imag, double* AopRe, double* AopIm,
                                          hipStream_t stream);

// ---- liouville.hip ---------------------------------------------------------------------------
size_t liouville_workspace_bytes(int batch, int d, int N);
hipError_t launch_liouville(const cplx* U, int batch, int d, const cplx* basis, int N,
                            int hermitian, double* out, void* ws, hipStream_t stream);


#if defined(__HIPCC__)
// One wavefront compacts basis element k into (count, entry index e = i*d + j, value C_k[j][i])
// lists: the expansion R[a,k,w] = sum_ij Y[a,i,j,w] C_k[j,i] then only touches the non-zeros
// (d per element for Pauli bases, <= 2 for the off-diagonal GGM elements).
__device__ __forceinline__ void basis_compact_one(const cplx* __restrict__ basis, int d, int k,
                                                  int lane, int* __restrict__ nnz,
                                                  int* __restrict__ rows,
                                                  cplx* __restrict__ vals) {
    const int dd = d*d;
    const cplx* C = basis + static_cast<size_t>(k)*dd;
    int count = 0;
    for (int base = 0; base < dd; base += 64) {
        const int e = base + lane;            // e = i*d + j  (row index into Bt)
        cplx v = {0.0, 0.0};
        if (e < dd) v = C[(e % d)*d + e / d];  // C_k[j][i]
        const bool nz = v.re != 0.0 || v.im != 0.0;
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
            rows[static_cast<size_t>(k)*dd + pos] = e;
            vals[static_cast<size_t>(k)*dd + pos] = v;
        }
        count += __popcll(mask);
    }
    if (lane == 0) nnz[k] = count;
}
#endif

}  // namespace ffk
