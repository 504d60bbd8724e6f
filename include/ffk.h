/* ffk.h -- C ABI of libffk.so: the MI355X (gfx950) filter-function hot path.
 *
 * This is the drop-in boundary for the numeric hot path of qutech/filter_functions
 * (reference v1.2.1).  The reference has no FFI of its own: the path sits behind Python
 * free functions in filter_functions/numeric.py and superoperator.py.  Every entry point
 * below replaces exactly one of those functions; the citation names the reference
 * interface (file:line relative to the reference root) whose arguments, array layouts and
 * results it reproduces.  INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - All arrays are C-contiguous.  "c128" is complex128 stored interleaved (re, im) as two
 *     doubles, identical to NumPy's layout; such arrays are passed as `const double*` /
 *     `double*` and have 2x the element count in doubles.
 *   - Plain pointers and sizes only.  No torch / numpy types.
 *   - Every function returns FFK_OK (0) or a negative FFK_E* code; ffk_last_error() returns
 *     a thread-local message for the last failure.
 *   - Two flavours per operation:
 *       ffk_<op>      host pointers in/out; the library stages through its own device
 *                     arena (H2D, kernels, D2H, synchronous).  This is what the NumPy-
 *                     facing Python front-end calls.
 *       ffk_<op>_dev  device pointers in/out, asynchronous on `stream` (a hipStream_t cast
 *                     to void*, NULL = default stream); scratch memory is supplied by the
 *                     caller (size from the matching *_workspace_bytes query).  No
 *                     allocation, no synchronisation: safe to capture in a hipGraph.  This
 *                     is what bench.py and the multi-GPU driver call with HBM-resident data.
 *   - d (Hilbert-space dimension) must satisfy 2 <= d <= FFK_MAX_D (64) for the path the reference's
 *     get_filter_function / infidelity / liouville_representation walk: ffk_diagonalize*,
 *     ffk_control_matrix* (control matrix and noise operators), ffk_filter_function*, ffk_infidelity*,
 *     ffk_decay_amplitudes*, ffk_cumulant_function*, ffk_expm_real, ffk_error_transfer_matrix_dev (the one
 *     `_dev` call that synchronises its stream, see there), ffk_control_matrix_from_atomic*, ffk_liouville*.  Up to
 *     FFK_MAX_D_TEMPLATED (16) the kernels are compiled per dimension (operands in registers or
 *     wave-private LDS); above it one runtime-d kernel set serves (csrc/generic.hip, workgroup-wide
 *     LDS tiles).  The remaining entry points (intermediates, second order, gradients, fused
 *     pipeline and resident passes, sequence concatenation in one call) accept
 *     d <= FFK_MAX_D_TEMPLATED only and return FFK_EINVAL above it.
 *   - Thread-safety: calls on one device are serialised by the caller; the library keeps one
 *     arena per process and device.
 */
#ifndef FFK_H
#define FFK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFK_VERSION 100 /* 0.1.0 */
#define FFK_MAX_D 64
#define FFK_MAX_D_TEMPLATED 16

#define FFK_OK 0
#define FFK_EINVAL -1   /* bad argument (shape, NULL, unsupported d) -> ValueError        */
#define FFK_EHIP -2     /* a HIP runtime call failed -> RuntimeError                       */
#define FFK_ENOMEM -3   /* device allocation failed / workspace too small -> MemoryError   */
#define FFK_ENOCONV -4  /* Jacobi eigensolver did not converge -> numpy.linalg.LinAlgError */
#define FFK_EKERNEL -5  /* a kernel reported an internal fault (a bounded wait between its       */
                        /* wavefronts ran out): the launch's results are invalid -> RuntimeError */

/* flags for ffk_control_matrix* */
#define FFK_WANT_NOISE_OPERATORS 0x1 /* also return B~(w) laid out (W, A, d, d)           */

/* which for ffk_filter_function* (numeric.py:1414 `which`) */
#define FFK_FF_FIDELITY 0
#define FFK_FF_GENERALIZED 1

/* ---- library / device management ------------------------------------------------------ */
const char* ffk_last_error(void);
int ffk_version(void);
int ffk_device_count(int* count);
int ffk_set_device(int device);
int ffk_get_device(int* device);
/* name (<= len-1 chars), compute units, total global memory bytes of the current device */
int ffk_device_info(char* name, int len, int* compute_units, size_t* global_mem_bytes);

/* ---- device memory, streams, events (so a host language without a HIP binding can keep
 *      data resident and time kernels on the stream they run on) --------------------------- */
int ffk_malloc(void** dptr, size_t bytes);
/* device memory that other agents (peer GPUs) may write while a local kernel polls it: fine-grained
 * coherence (hipExtMallocWithFlags).  FFK_EHIP if the runtime refuses: there is no silent
 * coarse-grained substitute (flag words polled by a running kernel would not be coherent) */
int ffk_malloc_finegrained(void** dptr, size_t bytes);
int ffk_free(void* dptr);
int ffk_memset(void* dptr, int value, size_t bytes, void* stream);
int ffk_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int ffk_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int ffk_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int ffk_stream_create(void** stream);
int ffk_stream_destroy(void* stream);
int ffk_stream_synchronize(void* stream);
int ffk_device_synchronize(void);
int ffk_event_create(void** event);
int ffk_event_destroy(void* event);
int ffk_event_record(void* event, void* stream);
int ffk_event_synchronize(void* event);
int ffk_event_elapsed_ms(void* start, void* stop, float* ms);
/* release the library's cached device arena (host-pointer flavour) */
int ffk_release_arena(void);

/* ---- numeric.diagonalize (filter_functions/numeric.py:1886-1935; caller
 *      pulse_sequence.py:577-586) ----------------------------------------------------------
 * hamiltonian (G, d, d) c128 -- only the LOWER triangle is read (numpy.linalg.eigh default);
 * dt (G,) f64.
 * -> eigvals (G, d) f64 ascending; eigvecs (G, d, d) c128, columns are eigenvectors;
 *    propagators (G+1, d, d) c128 with propagators[0] = 1 and
 *    propagators[g+1] = V_g exp(-i D_g dt_g) V_g^dag propagators[g].                        */
int ffk_diagonalize(const double* hamiltonian, const double* dt, int G, int d,
                    double* eigvals, double* eigvecs, double* propagators);
size_t ffk_diagonalize_workspace_bytes(int G, int d);
int ffk_diagonalize_dev(const double* hamiltonian, const double* dt, int G, int d,
                        double* eigvals, double* eigvecs, double* propagators,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- numeric.calculate_control_matrix_from_scratch (numeric.py:707-881; callers
 *      pulse_sequence.py:625, 1843, 2595) and its Hilbert-space twin
 *      numeric.calculate_noise_operators_from_scratch (numeric.py:456-618) -----------------
 * eigvals (G, d) f64, eigvecs (G, d, d) c128, propagators (G+1, d, d) c128, omega (W,) f64,
 * basis (N, d, d) c128, n_opers (A, d, d) c128, n_coeffs (A, G) f64, dt (G,) f64,
 * t (G+1,) f64 (absolute segment times; t[0] is usually 0).
 * -> control_matrix (A, N, W) c128, omega fastest (may be NULL if only the noise operators
 *    are wanted); with FFK_WANT_NOISE_OPERATORS also noise_operators (W, A, d, d) c128,
 *    omega slowest -- both exactly the reference's layouts.                                */
int ffk_control_matrix(const double* eigvals, const double* eigvecs, const double* propagators,
                       const double* omega, int W, const double* basis, int N,
                       const double* n_opers, int A, const double* n_coeffs, const double* dt,
                       const double* t, int G, int d, unsigned flags, double* control_matrix,
                       double* noise_operators);
size_t ffk_control_matrix_workspace_bytes(int W, int N, int A, int G, int d);
int ffk_control_matrix_dev(const double* eigvals, const double* eigvecs,
                           const double* propagators, const double* omega, int W,
                           const double* basis, int N, const double* n_opers, int A,
                           const double* n_coeffs, const double* dt, const double* t, int G,
                           int d, unsigned flags, double* control_matrix,
                           double* noise_operators, void* workspace, size_t workspace_bytes,
                           void* stream);

/* The `cache_intermediates=True` products of numeric.py:828-833, 871-878 (materialising,
 * HBM-bound variant).  Any output pointer may be NULL to skip it.
 *   n_opers_transformed (A, G, d, d), eigvecs_propagated (G, d, d), basis_transformed
 *   (G, N, d, d), phase_factors (G, W), first_order_integral (G, W, d, d),
 *   control_matrix_step (G, A, N, W); all c128.  (control_matrix_step_cumulative is the
 *   running sum of control_matrix_step and is formed by the host wrapper.)                 */
int ffk_control_matrix_intermediates(
    const double* eigvals, const double* eigvecs, const double* propagators,
    const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
    const double* n_coeffs, const double* dt, const double* t, int G, int d,
    double* n_opers_transformed, double* eigvecs_propagated, double* basis_transformed,
    double* phase_factors, double* first_order_integral, double* control_matrix_step);
/* The cache_intermediates products of calculate_noise_operators_from_scratch
 * (numeric.py:586-615): n_opers_transformed (A, G, d, d), phase_factors (G, W),
 * first_order_integral (G, W, d, d), noise_operators_step (G, W, A, d, d) c128 (any may be NULL). */
int ffk_noise_operators_intermediates(const double* eigvals, const double* eigvecs,
                                      const double* propagators, const double* omega, int W,
                                      const double* n_opers, int A, const double* n_coeffs,
                                      const double* dt, const double* t, int G, int d,
                                      double* n_opers_transformed, double* phase_factors,
                                      double* first_order_integral, double* noise_operators_step);

/* ---- numeric.calculate_control_matrix_from_atomic (numeric.py:621-704; caller
 *      pulse_sequence.concatenate pulse_sequence.py:1858) -- the concatenation rule -----------
 * phases (G-1, W) c128 cumulated total phase factors; control_matrix_atomic (G, A, N, W) c128;
 * propagators_liouville (G-1, N, N), f64 if l_is_complex == 0 else c128.
 * which 0 ('total'): out (A, N, W) = R^(0) + sum_g phases[g-1] R^(g) L^(g-1);
 * which 1 ('correlations'): out (G, A, N, W), every summand.                                */
int ffk_control_matrix_from_atomic(const double* phases, const double* control_matrix_atomic,
                                   const double* propagators_liouville, int l_is_complex, int G,
                                   int A, int N, int W, int which, double* out);
size_t ffk_control_matrix_from_atomic_workspace_bytes(int G, int A, int N, int W);
int ffk_control_matrix_from_atomic_dev(const double* phases, const double* control_matrix_atomic,
                                       const double* propagators_liouville, int l_is_complex,
                                       int G, int A, int N, int W, int which, double* out,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* The same rule for sequences drawn from T distinct pulses (pulse_sequence.concatenate called with
 * repeated PulseSequence objects, e.g. examples/randomized_benchmarking.py:76-81):
 * total_phases (T, W) c128 = exp(i omega tau) of every distinct pulse, control_matrix_table
 * (T, A, N, W) c128, index (G,) int32 = which distinct pulse sits at position g.  The cumulated
 * phases of pulse_sequence.py:1824 are formed in the kernel.                                   */
int ffk_control_matrix_from_atomic_indexed(const double* total_phases,
                                           const double* control_matrix_table,
                                           const int32_t* index,
                                           const double* propagators_liouville, int l_is_complex,
                                           int T, int G, int A, int N, int W, int which,
                                           double* out);
int ffk_control_matrix_from_atomic_indexed_dev(const double* total_phases,
                                               const double* control_matrix_table,
                                               const int32_t* index,
                                               const double* propagators_liouville,
                                               int l_is_complex, int T, int G, int A, int N, int W,
                                               int which, double* out, void* workspace,
                                               size_t workspace_bytes, void* stream);

/* ---- pulse_sequence.concatenate (pulse_sequence.py:1668-1887) for a sequence of G positions drawn
 * from T distinct pulses, the whole rule in one call: total_propagators (T, d, d) c128 of the
 * distinct pulses, their total_phases (T, W) c128 and control_matrix_table (T, A, N, W) c128,
 * index (G,) int32, basis (N, d, d) c128.  The cumulative propagators (util.adot, :1812), their
 * Liouville representations (:1827), the cumulative phases (:1824) and the sum (:1836,
 * numeric.calculate_control_matrix_from_atomic) all run on the device.  Returns the control matrix
 * ((A, N, W); (G, A, N, W) for which = 1), the total propagator (d, d) and, if
 * propagators_liouville is not NULL, the (G - 1, N, N) cumulative Liouville propagators (f64 if
 * hermitian_basis, else c128); if filter_function is not NULL (which = 0 only), also the fidelity
 * filter function (A, A, W) of the result (numeric.calculate_filter_function).                    */
int ffk_concatenate_sequence(const double* total_propagators, const double* total_phases,
                             const double* control_matrix_table, const int32_t* index,
                             const double* basis, int hermitian_basis, int T, int G, int d, int A,
                             int N, int W, int which, double* control_matrix,
                             double* total_propagator, double* propagators_liouville,
                             double* filter_function);

/* The same for distinct pulses whose control matrices are still resident in HBM (each evaluated by
 * ffk_resident_filter_function[_from_controls] on the same frequency grid, same device): handles
 * (T,), tau (T,) f64 total durations (the total phases exp(i omega tau) are formed on the device
 * from the resident grid), index (G,) int32, basis (N, d, d) c128.  The table of control matrices
 * is assembled by device-to-device copies; shapes come from the handles.  With `result` (another
 * handle; which = 0 and filter_function required, control_matrix may be NULL) the summed control
 * matrix, its filter function and the grid stay resident in it: ffk_resident_control_matrix and
 * ffk_resident_infidelity serve the concatenated pulse, and it can be an input of a further call. */
struct ffk_resident;
int ffk_concatenate_sequence_resident(struct ffk_resident* const* pulses, const double* tau,
                                      const int32_t* index, const double* basis, int hermitian_basis,
                                      int T, int G, int which, double* control_matrix,
                                      double* total_propagator, double* propagators_liouville,
                                      double* filter_function, struct ffk_resident* result);

/* ---- numeric.calculate_control_matrix_periodic (numeric.py:886-954) ----------------------
 * phases (W,) c128 = exp(i omega T) of one period, control_matrix (A, N, W) c128 of one period,
 * total_propagator_liouville (N, N) f64 (or c128 if l_is_complex) of one period -> the control
 * matrix (A, N, W) c128 of `repeats` periods.  The reference evaluates the geometric series in
 * closed form (one inverse per frequency, with a term-by-term fallback where it is ill
 * conditioned); here it is summed by doubling in ~2 log2(repeats) streaming passes, no inverse. */
int ffk_control_matrix_periodic(const double* phases, const double* control_matrix,
                                const double* total_propagator_liouville, int l_is_complex, int repeats,
                                int A, int N, int W, double* out);
size_t ffk_control_matrix_periodic_workspace_bytes(int A, int N, int W);
int ffk_control_matrix_periodic_dev(const double* phases, const double* control_matrix,
                                    const double* total_propagator_liouville, int l_is_complex,
                                    int repeats, int A, int N, int W, double* out, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* ---- numeric.calculate_filter_function (numeric.py:1413-1467) --------------------------
 * control_matrix (A, N, W) c128 -> fidelity: (A, A, W) c128,
 *                                  generalized: (A, A, N, N, W) c128.                       */
int ffk_filter_function(const double* control_matrix, int A, int N, int W, int which,
                        double* filter_function);
int ffk_filter_function_dev(const double* control_matrix, int A, int N, int W, int which,
                            double* filter_function, void* stream);

/* The filter function of a basis that is not traceless (numeric.py:2295-2305):
 * F[a,b,w] = scale * sum_kl conj(R[a,k,w]) M[k,l] R[b,l,w], weights M (N, N) c128 built by the
 * caller from the four-element traces of the basis, scale = 1/d.                               */
int ffk_filter_function_weighted(const double* control_matrix, int A, int N, int W,
                                 const double* weights, double scale, double* filter_function);
int ffk_filter_function_weighted_dev(const double* control_matrix, int A, int N, int W,
                                     const double* weights, double scale, double* filter_function,
                                     void* stream);

/* ---- numeric.infidelity, filter-function branch (numeric.py:2307-2320 with _get_integrand
 *      :323-325, :351-352, :374 and util.integrate util.py:880-906) ------------------------
 * filter_function (A, A, W) c128; omega (W,) f64; idx (n_idx,) int32 noise-operator indices;
 * spectrum c128 with s_ndim in {1, 2, 3}: (W,), (n_idx, W) or (n_idx, n_idx, W) (already
 * validated / broadcast by the caller as util.parse_spectrum util.py:214-227 does).
 * -> infid f64: (n_idx,) for s_ndim 1 or 2, (n_idx, n_idx) for s_ndim 3:
 *    Re-part trapezoid of S*F over omega, divided by 2 pi d.                                 */
int ffk_infidelity(const double* filter_function, int A, int W, const double* spectrum,
                   int s_ndim, const double* omega, const int32_t* idx, int n_idx, int d,
                   double* infid);
size_t ffk_infidelity_workspace_bytes(int W, int n_idx, int s_ndim);
int ffk_infidelity_dev(const double* filter_function, int A, int W, const double* spectrum,
                       int s_ndim, const double* omega, const int32_t* idx, int n_idx, int d,
                       double* infid, void* workspace, size_t workspace_bytes, void* stream);

/* The same integral on the raw buffer of an all-gather over omega blocks: filter_function_shards
 * (n_shards, A, A, shard_width) c128, block r holding frequencies [r, r+1) * shard_width of the
 * global grid omega (n_shards * shard_width,).  Saves the re-layout pass on the multi-GPU path. */
int ffk_infidelity_sharded_dev(const double* filter_function_shards, int n_shards, int shard_width,
                               int A, const double* spectrum, int s_ndim, const double* omega,
                               const int32_t* idx, int n_idx, int d, double* infid, void* stream);

/* ---- numeric.calculate_noise_operators_from_atomic (numeric.py:377-453) -------------------
 * phases (G-1, W) c128, noise_operators_atomic (G, W, A, d, d) c128, propagators (G-1, d, d) c128
 * -> noise_operators (W, A, d, d):  B = B^(0) + sum_{g>=1} phases[g-1] P_{g-1}^dag B^(g) P_{g-1}. */
int ffk_noise_operators_from_atomic(const double* phases, const double* noise_operators_atomic,
                                    const double* propagators, int G, int W, int A, int d,
                                    double* noise_operators);

/* ---- numeric.calculate_decay_amplitudes (numeric.py:1194-1337; integrand _get_integrand
 *      :310-374 'generalized' with the control matrix; util.integrate util.py:880-906) -------
 * control_matrix (n_pulses, A, N, W) c128: n_pulses = 1 is the total control matrix
 * (which='total'), n_pulses = G the pulse-correlation control matrix (which='correlations').
 * spectrum / s_ndim / idx / n_idx as in ffk_infidelity.
 * -> decay_amplitudes f64: (n_pulses, n_pulses, n_idx, N, N) for s_ndim 1 or 2,
 *    (n_pulses, n_pulses, n_idx, n_idx, N, N) for s_ndim 3:
 *    Gamma[g,h,a,b,k,l] = int dw/2pi Re(R*[g,a,k,w] S_ab(w) R[h,b,l,w])  (trapezoid).          */
size_t ffk_decay_amplitudes_workspace_bytes(int n_pulses, int N, int W, int n_idx, int s_ndim);
int ffk_decay_amplitudes_dev(const double* control_matrix, int n_pulses, int A, int N, int W,
                             const double* spectrum, int s_ndim, const double* omega,
                             const int32_t* idx, int n_idx, double* decay_amplitudes,
                             void* workspace, size_t workspace_bytes, void* stream);
int ffk_decay_amplitudes(const double* control_matrix, int n_pulses, int A, int N, int W,
                         const double* spectrum, int s_ndim, const double* omega,
                         const int32_t* idx, int n_idx, double* decay_amplitudes);
/* The contribution of one frequency block to the same integral (multi-GPU path):
 * control_matrix (n_pulses, A, N, W_block) and spectrum ([[n_idx,] n_idx,] W_block) hold the
 * frequencies [w_offset, w_offset + W_block) of the global grid omega (W,); the trapezoid
 * weights are those of the global grid, so the blocks' results add up to ffk_decay_amplitudes
 * of the whole grid.                                                                           */
int ffk_decay_amplitudes_shard_dev(const double* control_matrix, int n_pulses, int A, int N,
                                   int W_block, const double* spectrum, int s_ndim,
                                   const double* omega, int W, int w_offset, const int32_t* idx,
                                   int n_idx, double* decay_amplitudes, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---- numeric.calculate_cumulant_function, first order (numeric.py:957-1191) ----------------
 * decay_amplitudes (batch, N, N) f64 (any leading axes flattened into batch), basis (N, d, d)
 * c128 -> cumulant_function (batch, N, N) f64,
 *    K_ij = -1/2 sum_kl Gamma_kl (T_klji - T_kjli - T_kilj + T_kijl),  T = Basis.four_element_traces
 * (basis.py:330-348), evaluated without forming T.  single_qubit != 0 selects the simplified
 * expression the reference uses for d = 2 with a Pauli or GGM basis (:1119-1141).             */
size_t ffk_cumulant_function_workspace_bytes(int batch, int N, int d);
int ffk_cumulant_function_dev(const double* decay_amplitudes, int batch, int N, int d,
                              const double* basis, int single_qubit, double* cumulant_function,
                              void* workspace, size_t workspace_bytes, void* stream);
int ffk_cumulant_function(const double* decay_amplitudes, int batch, int N, int d,
                          const double* basis, int single_qubit, double* cumulant_function);

/* ---- second order: numeric.calculate_second_order_filter_function_from_scratch
 *      (numeric.py:1470-1699, nested integral :170-256), numeric.calculate_frequency_shifts
 *      (:1340-1410) and the frequency-shift terms of calculate_cumulant_function (:1139-1141,
 *      :1166-1190) ---------------------------------------------------------------------------
 * Inputs as ffk_control_matrix.  filter_function_2 (A, A, N, N, W) c128, omega fastest.        */
int ffk_second_order_filter_function(const double* eigvals, const double* eigvecs,
                                     const double* propagators, const double* omega, int W,
                                     const double* basis, int N, const double* n_opers, int A,
                                     const double* n_coeffs, const double* dt, const double* t, int G,
                                     int d, double* filter_function_2);
/* numeric.calculate_second_order_filter_function_from_atomic (numeric.py:1702-1818), the
 * concatenation rule: filter_function_atomic (G, A, A, N, N, W) c128 = each pulse's own F2,
 * control_matrix_step (G, A, N, W) c128 = the summands of the sequence's control matrix
 * (ffk_control_matrix_from_atomic with correlations != 0), propagators_liouville (G-1, N, N) f64
 * (Hermitian basis) -> filter_function_2 (A, A, N, N, W).                                       */
int ffk_second_order_filter_function_from_atomic(const double* filter_function_atomic,
                                                 const double* control_matrix_step,
                                                 const double* propagators_liouville, int G, int A,
                                                 int N, int W, double* filter_function_2);
/* Delta = int dw/2pi Re(S F2): spectrum/s_ndim/idx as in ffk_decay_amplitudes; frequency_shifts
 * (n_idx, N, N) f64 for s_ndim 1, 2 and (n_idx, n_idx, N, N) for s_ndim 3.                      */
int ffk_frequency_shifts(const double* filter_function_2, int A, int N, int W, const double* spectrum,
                         int s_ndim, const double* omega, const int32_t* idx, int n_idx,
                         double* frequency_shifts);
/* Both in one device-resident pass (F2 never crosses PCIe unless filter_function_2 != NULL, in
 * which case it is returned as well, for the caller's cache).                                   */
int ffk_frequency_shifts_from_scratch(const double* eigvals, const double* eigvecs,
                                      const double* propagators, const double* omega, int W,
                                      const double* basis, int N, const double* n_opers, int A,
                                      const double* n_coeffs, const double* dt, const double* t, int G,
                                      int d, const double* spectrum, int s_ndim, const int32_t* idx,
                                      int n_idx, double* filter_function_2, double* frequency_shifts);
/* cumulant_function (batch, N, N) f64, IN/OUT: the first-order result of ffk_cumulant_function, to
 * which -1/2 sum_kl Delta_kl (T_klji - T_lkji - T_klij + T_lkij) is added, evaluated as the
 * commutator -1/2 Re tr(C_i [X, C_j]), X = sum_kl (Delta_kl - Delta_lk) C_k C_l (which is also the
 * reference's single-qubit expression -(Delta - Delta^T) on the traceless block).              */
int ffk_cumulant_function_second_order(const double* frequency_shifts, int batch, int N, int d,
                                       const double* basis, double* cumulant_function);

/* Device-resident flavours (device pointers, caller-provided workspace, asynchronous on `stream`);
 * ffk_frequency_shifts_shard_dev integrates the frequency block [w_offset, w_offset + W_block) of
 * the global grid omega (W,) with the global trapezoid weights (multi-GPU: the per-rank results
 * add up to the unsharded integral).                                                            */
size_t ffk_second_order_workspace_bytes(int W, int N, int A, int G, int d);
int ffk_second_order_filter_function_dev(const double* eigvals, const double* eigvecs,
                                         const double* propagators, const double* omega, int W,
                                         const double* basis, int N, const double* n_opers, int A,
                                         const double* n_coeffs, const double* dt, const double* t,
                                         int G, int d, double* filter_function_2, void* workspace,
                                         size_t workspace_bytes, void* stream);
size_t ffk_frequency_shifts_workspace_bytes(int W, int n_idx, int s_ndim);
int ffk_frequency_shifts_shard_dev(const double* filter_function_2, int A, int N, int W_block,
                                   const double* spectrum, int s_ndim, const double* omega, int W,
                                   int w_offset, const int32_t* idx, int n_idx,
                                   double* frequency_shifts, void* workspace, size_t workspace_bytes,
                                   void* stream);
size_t ffk_cumulant_function_second_order_workspace_bytes(int batch, int N, int d);
int ffk_cumulant_function_second_order_dev(const double* frequency_shifts, int batch, int N, int d,
                                           const double* basis, double* cumulant_function,
                                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- gradient (filter_functions/gradient.py; PulseSequence.get_filter_function_derivative,
 *      pulse_sequence.py:977-1054; gradient.infidelity_derivative, gradient.py:559-676) ----------
 * Derivative of the fidelity filter function of each noise operator with respect to the amplitude
 * of control operator h during segment s: filter_function_derivative (A, G, H, W) f64 (the
 * reference's (n_nops, n_dt, n_ctrl, n_omega)); infidelity_derivative (A, G, H) f64 =
 * int dw/(2 pi d) S_a(w) dF_a/du_h(t_s) for a spectrum (W,) or (A, W) c128 (s_ndim 1, 2).  Either
 * output may be NULL.  c_opers (H, d, d) c128: the control operators to differentiate by (the
 * eigensystem is that of the full control Hamiltonian); n_coeffs_ratio (A, H, G) f64 or NULL:
 * (d n_coeffs[a, s] / d u_h(t_s)) / n_coeffs[a, s], the explicit dependence of the noise
 * sensitivities on the controls (gradient.py:376-379).  Supports 2 <= d <= 8.                  */
int ffk_filter_function_derivative(const double* eigvals, const double* eigvecs,
                                   const double* propagators, const double* omega, int W,
                                   const double* n_opers, int A, const double* n_coeffs,
                                   const double* c_opers, int H, const double* n_coeffs_ratio,
                                   const double* dt, const double* t, int G, int d,
                                   const double* spectrum, int s_ndim,
                                   double* filter_function_derivative,
                                   double* infidelity_derivative);

/* The two tensor-level gradient functions of the reference.
 * ffk_control_matrix_derivative: gradient.calculate_derivative_of_control_matrix_from_scratch
 * (gradient.py:384-523): control_matrix_derivative (H, W, G, A, N) c128 = d R_ak(w) / d u_h(t_s)
 * [the reference's (n_ctrl, n_omega, n_dt, n_nops, d**2)]; basis (N, d, d) c128; the other arguments
 * as for ffk_filter_function_derivative.  2 <= d <= 8.
 * ffk_filter_function_derivative_from_control_matrix: gradient.calculate_filter_function_derivative
 * (gradient.py:526-556): (A, G, H, W) f64 = 2 Re sum_k conj(R[a,k,w]) dR[h,w,s,a,k] from
 * control_matrix (A, N, W) c128 and control_matrix_derivative (H, W, G, A, N) c128.              */
int ffk_control_matrix_derivative(const double* eigvals, const double* eigvecs, const double* propagators,
                                  const double* omega, int W, const double* basis, int N,
                                  const double* n_opers, int A, const double* n_coeffs,
                                  const double* c_opers, int H, const double* n_coeffs_ratio,
                                  const double* dt, const double* t, int G, int d,
                                  double* control_matrix_derivative);
int ffk_filter_function_derivative_from_control_matrix(const double* control_matrix,
                                                       const double* control_matrix_derivative, int A,
                                                       int N, int W, int G, int H,
                                                       double* filter_function_derivative);

/* Device-resident flavour on one block [w_offset, w_offset + W_block) of the global grid omega (W,):
 * omega_block (W_block,) are the block's frequencies, spectrum (W_block,) or (A, W_block) its part of
 * the spectrum; the trapezoid weights are those of the global grid, so the per-rank
 * infidelity_derivative results add up to the unsharded one (multi-GPU).
 * filter_function_derivative (A, G, H, W_block) is required (device scratch if not wanted).       */
size_t ffk_filter_function_derivative_workspace_bytes(int W, int A, int H, int G, int d);
int ffk_filter_function_derivative_shard_dev(const double* eigvals, const double* eigvecs,
                                             const double* propagators, const double* omega_block,
                                             int W_block, const double* n_opers, int A,
                                             const double* n_coeffs, const double* c_opers, int H,
                                             const double* n_coeffs_ratio, const double* dt,
                                             const double* t, int G, int d, const double* spectrum,
                                             int s_ndim, const double* omega, int W, int w_offset,
                                             double* filter_function_derivative,
                                             double* infidelity_derivative, void* workspace,
                                             size_t workspace_bytes, void* stream);

/* ---- exp of the summed cumulant function (numeric.error_transfer_matrix, numeric.py:2049-2053;
 *      the reference calls scipy.linalg.expm) ---------------------------------------------------
 * matrix (N, N) f64 row-major -> result (N, N) = exp(matrix): scaling and squaring with a Taylor
 * polynomial of degree 18, every product on v_mfma_f64_16x16x4.                                */
int ffk_expm_real(const double* matrix, int N, double* result);

/* The same for a cumulant function that is already in HBM (round 6): cumulant_function (batch, N, N) f64 is summed
 * over its leading axis in order (numeric.py:2049: `cumulant_function.sum(axis=...)`), the 1-norm that sizes the
 * scaling is taken on the device (16 bytes cross to the host, one stream synchronisation: the number of squarings
 * decides how many products are enqueued), result (N, N) f64 stays in HBM.  NaN / Inf in the sum: FFK_EINVAL.  */
size_t ffk_error_transfer_matrix_workspace_bytes(int N);
int ffk_error_transfer_matrix_dev(const double* cumulant_function, int batch, int N, double* result,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- superoperator.liouville_representation (superoperator.py:51-84 + Basis.expand
 *      basis.py:350-371, 650-698) --------------------------------------------------------
 * U (batch, d, d) c128, basis (N, d, d) c128 -> liouville (batch, N, N):
 * L_ij = tr(U^dag C_i U C_j), written as f64 (real part) if `hermitian_basis` != 0 (the
 * reference casts to real iff basis.isherm), else c128.                                     */
int ffk_liouville(const double* U, int batch, int d, const double* basis, int N,
                  int hermitian_basis, double* liouville);
size_t ffk_liouville_workspace_bytes(int batch, int d, int N);
int ffk_liouville_dev(const double* U, int batch, int d, const double* basis, int N,
                      int hermitian_basis, double* liouville, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- fused path: PulseSequence.get_filter_function + infidelity in one device-resident
 *      pass (pulse_sequence.py:577-586, 588-636, 691-902 and numeric.py:2062-2334) ---------
 * hamiltonian (G, d, d) c128 (lower triangle), dt (G,), omega (W,), basis (N, d, d),
 * n_opers (A, d, d), n_coeffs (A, G); spectrum/idx/s_ndim as in ffk_infidelity (spectrum
 * may be NULL: no infidelity).  Outputs (any may be NULL): eigvals, eigvecs, propagators,
 * control_matrix (A, N, W), filter_function (A, A, W), infid.
 * `t` is formed on the host by the caller (t = [0, cumsum(dt)]) exactly like the reference
 * (pulse_sequence.py:534) so the phase arguments omega*t_g round identically.               */
size_t ffk_pipeline_workspace_bytes(int W, int N, int A, int G, int d, int n_idx, int s_ndim);
int ffk_pipeline_dev(const double* hamiltonian, const double* dt, const double* t, int G, int d,
                     const double* omega, int W, const double* basis, int N,
                     const double* n_opers, int A, const double* n_coeffs,
                     const double* spectrum, int s_ndim, const int32_t* idx, int n_idx,
                     double* eigvals, double* eigvecs, double* propagators,
                     double* control_matrix, double* filter_function, double* infid,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Number of segments whose eigensolver iteration did not converge (or met NaN/Inf) in the last
 * ffk_diagonalize_dev / ffk_pipeline_dev run that used `workspace`, written to the device integer
 * `n_failed` on `stream`: the device-resident counterpart of the FFK_ENOCONV return of
 * ffk_diagonalize (numpy.linalg.LinAlgError of numeric.py:1919).                               */
int ffk_eigensolver_status_dev(const void* workspace, size_t workspace_bytes, int G, int d,
                               int32_t* n_failed, void* stream);

/* Fault word of the kernels whose wavefronts hand tiles to one another through flags in LDS (the d = 4
 * accumulation behind ffk_control_matrix*, ffk_pipeline_dev and the resident passes; reference loop
 * numeric.py:846-869).  Their waits are bounded; a wait that runs out stores a non-zero code in a word of mapped
 * host memory that the launcher hands the kernel as an ARGUMENT (so it is right on whichever device the launch goes
 * to), and the launch's results are invalid.  ONE WORD PER HOST THREAD: a thread sees the faults of the launches it
 * enqueued itself, and only those (a captured graph reports to the thread that captured it).  The host-pointer
 * entry points read the calling thread's word after their own synchronisation and return FFK_EKERNEL; a fault that
 * an EARLIER, never checked asynchronous launch of the thread left behind is reported by them on entry, as such,
 * before anything runs.  Callers of the `_dev` flavour call this AFTER synchronising the stream, on the thread that
 * enqueued: *word = 0 means every launch of this thread since the last clearing was sound.  Sticky until read with
 * clear != 0.                                                                                                    */
int ffk_kernel_fault_status(int32_t* word, int clear);

/* ---- resident evaluation: the user-facing call PulseSequence.get_filter_function(omega)
 *      followed by ff.infidelity(pulse, S, omega) (pulse_sequence.py:691-805, 577-677;
 *      numeric.py:2062-2334) on host arrays with the minimum of PCIe traffic ------------------
 * ffk_resident_filter_function stages all inputs (same arrays as ffk_pipeline_dev, host pointers)
 * through one pinned block with ONE H2D copy, runs the fused pass, and returns the small results
 * -- eigvals (G, d), eigvecs (G, d, d), propagators (G+1, d, d), filter_function (A, A, W) -- by ONE
 * D2H copy as pointers into pinned host memory owned by the handle (valid until the handle runs
 * another pass or is destroyed; the binding wraps them as arrays without copying).  The control
 * matrix (A, N, W) -- 5x the bytes of F at config 2 -- stays in HBM and is fetched only when the
 * caller asks for it (ffk_resident_control_matrix), which is what the reference's cache
 * (`pulse._frequency_data['control_matrix']`) needs it for.  ffk_resident_infidelity integrates
 * the resident F against a host spectrum ((W,), (n_idx, W) or (n_idx, n_idx, W); real f64 if
 * spectrum_is_real, else c128) on the resident frequency grid, normalised by 1/(2 pi d) with the
 * caller's d (pulse.d, which a user may override: tests/test_precision.py:300).  Returns FFK_ENOCONV like
 * ffk_diagonalize.  Device and pinned blocks come from grow-only pools
 * (ffk_resident_release_pools frees what is idle).                                             */
typedef struct ffk_resident ffk_resident;
int ffk_resident_create(ffk_resident** handle);
int ffk_resident_destroy(ffk_resident* handle);
int ffk_resident_release_pools(void);
int ffk_resident_filter_function(ffk_resident* handle, const double* hamiltonian, const double* dt,
                                 const double* t, int G, int d, const double* omega, int W,
                                 const double* basis, int N, const double* n_opers, int A,
                                 const double* n_coeffs, double** eigvals, double** eigvecs,
                                 double** propagators, double** filter_function);
/* The same pass with the control Hamiltonian given as PulseSequence holds it -- c_opers
 * (n_cops, d, d) c128 and c_coeffs (n_cops, G) f64 -- instead of the summed (G, d, d) array of
 * numeric.diagonalize (pulse_sequence.py:1300-1302, einsum 'ijk,il->ljk'): the sum runs on the
 * device, 8 n_cops bytes per segment cross PCIe instead of 16 d^2.                               */
int ffk_resident_filter_function_from_controls(ffk_resident* handle, const double* c_opers, int n_cops,
                                               const double* c_coeffs, const double* dt,
                                               const double* t, int G, int d, const double* omega,
                                               int W, const double* basis, int N,
                                               const double* n_opers, int A, const double* n_coeffs,
                                               double** eigvals, double** eigvecs,
                                               double** propagators, double** filter_function);
/* The same pass with the infidelity integral of numeric.infidelity (numeric.py:2063-2334, `which='total'`,
 * traceless basis) riding in it: `ff.infidelity(pulse, S, omega)` on a pulse with nothing cached is one
 * round trip to the device instead of two.  spectrum: (W,), (n_idx, W) or (n_idx, n_idx, W), f64 if
 * spectrum_is_real else c128, already validated (util.parse_spectrum, util.py:214-227); idx: the noise
 * operators' indices; d_infidelity: the dimension in 1/(2 pi d); infidelity: (n_idx,) resp. (n_idx, n_idx). */
int ffk_resident_filter_function_infidelity(ffk_resident* handle, const double* c_opers, int n_cops,
                                            const double* c_coeffs, const double* dt, const double* t,
                                            int G, int d, const double* omega, int W, const double* basis,
                                            int N, const double* n_opers, int A, const double* n_coeffs,
                                            const double* spectrum, int s_ndim, int spectrum_is_real,
                                            const int32_t* idx, int n_idx, int d_infidelity,
                                            double** eigvals, double** eigvecs, double** propagators,
                                            double** filter_function, double* infidelity);
/* host-clock seconds of the last pass: [0] packing the inputs into the pinned block, [1] enqueueing
 * the H2D copy, the kernels and the D2H copy, [2] waiting for the stream                         */
int ffk_resident_timing(ffk_resident* handle, double* seconds);
int ffk_resident_control_matrix(ffk_resident* handle, double* control_matrix);
/* device pointers of the resident control matrix, filter function and frequencies (any may be NULL) */
int ffk_resident_control_matrix_dev(ffk_resident* handle, const double** control_matrix,
                                    const double** filter_function, const double** omega);
int ffk_resident_infidelity(ffk_resident* handle, const double* spectrum, int s_ndim,
                            int spectrum_is_real, const int32_t* idx, int n_idx, int d,
                            double* infid);

/* ---- one-sided all-gather of the F blocks over xGMI (frequency-sharded step; the
 *      reference has no multi-device path: numeric.py:846-869 is embarrassingly parallel in omega
 *      and this is the exchange that reassembles F(omega) for util.integrate, util.py:880-906) ----
 * Every rank pushes its block into slot `rank` of a gather buffer on every rank through pointers
 * obtained with ffk_ipc_open_handle, with a copy kernel that needs no LDS (it shares the CUs with
 * the accumulate kernel, which an RCCL all-gather kernel cannot), and completion travels as
 * sequence numbers in 64-bit flag words polled with a timeout (error word, sticky -- the first
 * failure stays: 1 = acknowledgement timed out in push, the copy to that peer was skipped; 2 =
 * signal timed out in wait; 3 = a peer published the poison value).  A rank whose error word is set
 * signals the poison value (-1) instead of sequence numbers from then on, so that no peer
 * integrates a slot that was never filled: the failure reaches every rank within one step.  dst / flags / acks are DEVICE arrays of
 * `world` device pointers (addresses on rank p of: the slot of this rank in the buffer set, the flag
 * word of this rank, the acknowledgement word of this rank); `acks` / `flags` of push / wait are
 * this rank's own words, one per peer.  filter_functions_amd/parallel.py holds the protocol.     */
#define FFK_IPC_HANDLE_BYTES 64
int ffk_ipc_get_handle(const void* dptr, void* handle);
int ffk_ipc_open_handle(const void* handle, void** dptr);
int ffk_ipc_close_handle(void* dptr);
int ffk_peer_push_dev(const double* src, size_t bytes, void* const* dst, const int64_t* acks,
                      int64_t need_ack, int world, int rank, int32_t* error, void* stream);
int ffk_peer_signal_dev(void* const* flags, void* const* acks, int world, int64_t seq,
                        int64_t consumed, const int32_t* error, void* stream);
/* timeout of every poll of the protocol (default 2000 ms, or the environment variable
 * FFK_PEER_TIMEOUT_MS); legitimate host stalls on a peer (first-launch compilation, a profiler, a
 * garbage collection) must fit into it */
int ffk_peer_set_timeout_ms(double ms);
int ffk_peer_wait_dev(const int64_t* flags, int world, int64_t seq, int32_t* error, void* stream);
/* the three of step `step` in one call: push (after acknowledgements >= need_ack), signal
 * (flags = step + 1, acknowledging `step` buffers consumed), wait (flags of every peer >= step + 1) */
int ffk_peer_step_dev(const double* src, size_t bytes, void* const* dst, const int64_t* own_acks,
                      int64_t need_ack, void* const* flag_at, void* const* ack_at,
                      const int64_t* own_flags, int world, int rank, int64_t step, int32_t* error,
                      void* stream);

/* ---- hipGraph capture of device-pointer calls ------------------------------------------------
 * The reference's user-facing call (PulseSequence.get_filter_function, pulse_sequence.py:691-805 ->
 * numeric.calculate_control_matrix_from_scratch, numeric.py:707-881) is one Python call; here a
 * pass is 6 kernel launches whose host-side enqueue cost rivals their device time.  Any sequence
 * of `_dev` calls issued on `stream` between ffk_graph_capture_begin and ffk_graph_capture_end is
 * recorded instead of executed and comes back as one graph; ffk_graph_launch replays it on any
 * stream with a single runtime call.  Work forked onto other streams inside the capture
 * (ffk_event_record on the captured stream, a wait for that event on the other stream) is captured
 * too, provided it is joined back the same way before the end.  Arguments are frozen at capture:
 * every buffer a captured call names (inputs, outputs, workspace) must stay allocated and in place
 * while the graph lives; their CONTENTS may change between replays.  `stream` must be a created
 * stream (ffk_stream_create or any hipStream_t), not the null stream.  ffk_graph_capture_abort
 * leaves capture mode after a failed call. */
typedef struct ffk_graph ffk_graph;
int ffk_graph_capture_begin(void* stream);
int ffk_graph_capture_end(void* stream, ffk_graph** graph);
int ffk_graph_capture_abort(void* stream);
int ffk_graph_launch(ffk_graph* graph, void* stream);
int ffk_graph_node_count(const ffk_graph* graph, int* nodes);
int ffk_graph_destroy(ffk_graph* graph);
int ffk_stream_wait_event(void* stream, void* event);

/* ---- tuning / introspection ------------------------------------------------------------ */
/* Number of segment chunks the control-matrix kernel splits G into (0 = automatic).        */
int ffk_set_segment_chunks(int chunks);
/* Kernel variant of the control-matrix accumulation: 0 = default (multi-wave blocks sharing the
 * generated integral through LDS), 1 = one-wave-per-block variant for d <= 4 (kept for tuning;
 * measured slower on MI355X because its 48 accumulators per lane spill into AGPRs),
 * 2 = default kernel without the in-block segment split (tuning), 3 = never use the matrix-core
 * kernel (d >= 12 use it by default), 4 = use the matrix-core kernel wherever it exists (d = 8 too). */
int ffk_set_accumulate_variant(int variant);
/* Per-call statistics of the last ffk_control_matrix*_dev launch on this thread:
 * algorithmic FP64 flops of the accumulate kernel, its grid/block geometry, chunks used.   */
typedef struct ffk_stats {
    double accumulate_flops;   /* FMA-counted real flops ffk::ctrl_accumulate executes      */
    double accumulate_bytes;   /* HBM bytes it must move (inputs + partial sums)            */
    int chunks;                /* segment chunks                                            */
    int grid_x, grid_y, grid_z, block;
    int lds_bytes;
} ffk_stats;
int ffk_get_stats(ffk_stats* out);
/* Profiling hook: when both are non-NULL hipEvent_t handles, the next ffk_control_matrix_dev /
 * ffk_pipeline_dev calls on this thread record `start` immediately before and `stop`
 * immediately after the accumulate kernel, on the stream it is launched on (so that a caller
 * can time the dominant kernel inside its own timed region); the concatenation entry points
 * (ffk_concatenate_sequence*) do the same around their rule kernel.  Pass NULLs to switch off. */
int ffk_set_accumulate_events(void* start, void* stop);
/* With several passes in flight on several streams, the accumulate kernel of one pass waits for the
 * blocks of the other pass's to retire, and a start event recorded on its own stream would include
 * that wait.  `event` (recorded on another stream; NULL = none) is waited for on the launch stream
 * right before `start` is recorded; it applies to the calls that follow until reset, and
 * ffk_set_accumulate_events resets it.                                                          */
int ffk_set_accumulate_gate(void* event);

#ifdef __cplusplus
}
#endif
#endif /* FFK_H */
