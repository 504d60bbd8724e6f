// ctrl_pq.hip -- K3q: the control-matrix accumulation for d = 4 (BASELINE config 2, the headline) with
// SPECIALISED wavefronts and the second product on the matrix cores (round 5; replaces ctrl_pc.hip).
//     Y_a(w) = sum_g T_g^dag [ Bbar_a o E_g(w) ] T_g,   E = psi e^{ib} q,  q = 2 sin(a + b)/x REAL   (ffk_math.h)
//     Z_a[m][j] = sum_n q[m][n] W_a[m][n][j]     W_a = Bbar_a e^{ib} T, frequency independent, folded by the producers
//     Y_a[i][j] += sum_m c[m][i] Z_a[m][j]       c = psi conj(T): psi is per frequency, T per segment
// Replaces the reference's hot loop numeric.py:846-869 / :596-609.
//
// A block owns 64 frequencies, a chunk of the segments and NC = 1, 2 or 3 operators: 12 wavefronts.
//   * FOUR PRODUCERS (one per SIMD, lane = frequency) generate the tile of segment p, p + 4, ...: the 13 distinct
//     q, psi, the folded operands W_a and (Tr, Ti) -- into a ring of eight LDS slots.  The table row of the NEXT
//     tile and its operands are requested a tile ahead and parked in a private LDS row, whose records are then
//     read in two batches (through scalar loads the row's 41 doubles do not fit the SGPR file: ten batches with
//     a full wait each, 3.1 us per tile; profiles/r05_b_*).
//   * EIGHT CONSUMERS (two per SIMD) own eight frequencies each -- two sets of four -- and ALL operators, and
//     walk every tile.  A lane is (row m, frequency f of the set, column j): the B-operand layout of
//     v_mfma_f64_4x4x4_4b with ONE FREQUENCY PER 4x4x4 BLOCK.  The first product stays on the vector ALU in that
//     layout (8 instructions per operator and set, all 64 lanes busy, W_a[m][.][j] in registers for the whole
//     tile), so Z lands where the matrix instruction wants its B operand: nothing moves between the products.
//     The second product is THREE matrix instructions (Gauss: P1 = cr^T zr, P2 = ci^T zi, P3 = (cr + ci)^T
//     (zr + zi); Re Y = P1 - P2, Im Y = P3 - P1 - P2; each linear in the segment's data, so P1..P3 are summed
//     over the segments and combined once at the end).  psi multiplies the A operand -- one complex product per
//     lane and set, shared by the operators -- instead of every Z.  Per operator and four frequencies: 9 vector +
//     3 matrix instructions + 5/NC shared, on a mix that holds 2.33 GHz where the pure v_fma_f64 stream is power
//     capped at 2.03 (tools/fp64_mix_probe.hip, profiles/r05_a_*).  6 accumulator registers per lane, operator
//     and set: no segment split inside the block, no tree reduction at the end.
//   * The consumer's WHOLE TILE LOOP is one generated asm block per operator count (ctrl_pq_consumer.inc,
//     tools/gen_pq_consumer.py: PqConsumer<NC>::run), unrolled over the ring's eight slots so that every LDS
//     address is a per-lane base register plus an immediate: the next tile's operands are requested from inside
//     the last set, when their registers are dead, and fly during its matrix instructions and the hand-over; the
//     flag of the tile after that and the SIMD partner's progress are read a tile ahead; every s_waitcnt carries
//     the exact count of younger LDS operations (a wavefront does not issue in the shadow of its own matrix
//     instructions: tools/fp64_issue_probe.py, profiles/r05_l_*, r05_m_*).  142 / 108 / 74 VGPRs for NC = 3 / 2 /
//     1: three wavefronts of the NC = 3 form per SIMD leave room for a wavefront of another pass's small kernels
//     (56): the two-pass schedule of the bench keeps overlapping (tests/test_kernel_resources.py).
//     Round 6: NC = 1 and 2 run the generated loop too (round 5: a compiler-scheduled C++ consumer), and a launch
//     serves A operators as groups of three plus groups of two (A = 4: 2 + 2, 5: 3 + 2, 7: 3 + 2 + 2) instead of
//     a zero-padded three-operator block (profiles/r06_*).
//   * W_a comes folded from the prologue kernel where the caller owns a buffer for it (PRE: wfold != nullptr,
//     ffk_control_matrix_dev / ffk_pipeline_dev), else the producers fold it per tile.
//   * Flags in LDS: ready[slot] written by the slot's producer, done[slot] counted up by the consumers with
//     ds_add, progress[consumer]; the lagging consumer of a SIMD raises its priority (the arbiter serves the
//     oldest wavefront first and would let one run ahead until the ring stops it: soft lockstep).  Every wait is
//     bounded; a wait that runs out is a reported FAULT (the launch's fault word -- a kernel argument, so it is
//     right on every device -- ffk_internal.h::kernel_fault_word -> FFK_EKERNEL), the wavefront stops waiting for
//     the rest of the launch and runs to the end so that the grid drains.
//     -DFFK_PC_SPIN_LIMIT=n -DFFK_PC_FAULT_INJECT: the test build whose producers stop publishing after eight
//     tiles (tests/test_gpu_parity.py::test_flag_wait_timeout_is_an_error).
// History with measurements: profiles/r05_b_d4_matrix_core_kernel_steps.txt, r05_l_*; where the time is:
// DESIGN.md section 6.1.  The instrumented round-5 source (clocks, ablations, the four-set and per-tile forms)
// is csrc/tuning/ctrl_pq_r5.hip, built only by `make VARIANT=...`.
#include <algorithm>
#include <cstdlib>

#include "ffk_internal.h"

namespace ffk {
namespace {

constexpr int kPqProducers = 4;   // wavefronts 0..3, one per SIMD
constexpr int kPqSets = 2;        // a consumer owns this many sets of four of the block's 64 frequencies, all operators
constexpr int kPqConsumers = 16/kPqSets;   // wavefronts 4..11: two per SIMD
constexpr int kPqRing = 8;        // tile slots: two per producer
#ifndef FFK_PC_SPIN_LIMIT
#define FFK_PC_SPIN_LIMIT (1 << 21)
#endif
constexpr int kPqSpinLimit = FFK_PC_SPIN_LIMIT;

__device__ __forceinline__ void pq_report_fault(int* fault, int code) {
    if (fault != nullptr && (threadIdx.x & 63) == 0)
        __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double8_t __attribute__((ext_vector_type(8)));
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lds_int_t;

template <int NC>
struct PqConsumer;                // the consumer's tile loop as generated assembly
#include "ctrl_pq_consumer.inc"   // generated: tools/gen_pq_consumer.py

// ---- one tile slot, in doubles ----------------------------------------------------------------------
//   [0, 1024)      q: two planes h (columns n = 2h, 2h + 1), per plane [frequency 0..63][4 slots of
//                  (q[m][2h], q[m][2h+1])], row m of frequency w in slot (m ^ (w >> 2)) & 3: the
//                  producer's lanes (= frequencies, 64 bytes apart) store to all bank groups, a
//                  consumer's set of four frequencies reads 256 contiguous bytes
//   [1024, 1152)   psi[frequency] (re, im)
//   [1152, ..)     W_a[n][m][j] complex, NC x 64: the consumer lane (m, ., j) reads column n with its
//                  n-th 16-byte read, every read 256 contiguous bytes
//   then           (Tr, Ti)[16], index m*4 + i, 16 bytes each: the A operands before psi (same lane offset as W)
// behind the ring: one private copy of the current segment's table row per producer, the flags
constexpr int kPqQ = 0, kPqPsi = 1024, kPqW = 1152;
constexpr int kPqRow = seg_stride(4);             // 72 doubles
__host__ __device__ constexpr int pq_tile_doubles(int nc) { return kPqW + nc*128 + 32; }
__host__ __device__ constexpr int pq_lds_bytes_for(int nc) {
    return (kPqRing*pq_tile_doubles(nc) + kPqProducers*kPqRow)*8 + 3*kPqRing*4;
}

__device__ __forceinline__ int lds_peek(const int* flag) {
    return __builtin_amdgcn_readfirstlane(*(const volatile lds_int_t*)(flag));
}

// PRE: W_a comes folded from the prologue kernel (wfold) instead of being folded here.  The launch serves the
// operators [alpha_base, alpha_end) in blocks of NC.
template <int NC, bool PRE>
__global__ __launch_bounds__((kPqProducers + kPqConsumers)*64, 3) void ctrl_accumulate_pq_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart,
    const cplx* __restrict__ wfold, int alpha_base, int alpha_end, int* __restrict__ fault_word) {
    constexpr int D = 4, DD = 16, S = kPqRow, TILE = pq_tile_doubles(NC);
    constexpr int TOP = kPqW + NC*128;                  // the A operands of a slot
    static_assert(PqConsumer<NC>::kTileBytes == TILE*8, "tools/gen_pq_consumer.py and pq_tile_doubles() disagree");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* ring = reinterpret_cast<double*>(lds_raw);
    double* rows = ring + kPqRing*TILE;
    int* ready = reinterpret_cast<int*>(rows + kPqProducers*S);
    int* done = ready + kPqRing;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int alpha0 = alpha_base + blockIdx.y*NC;
    const int n_alpha = min(NC, alpha_end - alpha0);
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);
    const int n_it = max(0, g1 - g0);                  // tiles of this block
    int spin_limit = kPqSpinLimit;                     // 0 after a wait of this wavefront has run out

    if (threadIdx.x < 3*kPqRing) *(volatile lds_int_t*)(ready + threadIdx.x) = 0;
    __syncthreads();

    if (wave < kPqProducers) {
        // ---- producer p: tiles p, p + 4, ... -------------------------------------------------------
        // The operands of tile it + 4 (one element of [T | Bbar_a] and two doubles of the table row per
        // lane) are requested before tile it is worked on and parked in this producer's LDS row at the top
        // of their own iteration: every record of the row is then one broadcast ds_read away.  (Through
        // scalar loads the 41 doubles of a row do not fit the SGPR file at once; the compiler fetched them
        // in ten batches with a full wait after each -- 3.1 us per tile, profiles/r05_b_*.)
        // Priority 0: the producers are the oldest wavefronts of their SIMDs and win the arbiter anyway, except against
        // a consumer that is behind its partner.  (Priority 3, as the small kernels have it: kernel the same, step
        // +1.3 %; consumers always above the producers: 71 us -- the producers alone make a tile per 1600 cycles, the
        // consumers want one per 1430, and a starved ring costs more than it saves.  profiles/r05_l_*.)
        __builtin_amdgcn_s_setprio(0);
        const int iw = blockIdx.x*64 + lane;
        const double om = omega[iw < W ? iw : W - 1];
        const int n_ops = (1 + n_alpha)*DD;            // <= 64: one staged element per lane
        double* row = rows + wave*S;
        struct Staged {
            cplx o;            // lane l: element l of [T | Bbar_0 | Bbar_1 ..] of the segment
            double r0, r1;     // doubles l and (l < 8) 64 + l of the table row
        };
        auto request = [&](int it) __attribute__((always_inline)) -> Staged {
            const int g = g0 + it;
            const cplx* src = ops + static_cast<size_t>(g)*(1 + A)*DD;
            const double* st = segtab + static_cast<size_t>(g)*S;
            Staged t;
            t.o = lane < n_ops ? src[lane < DD ? lane : lane + alpha0*DD] : cplx{0.0, 0.0};
            t.r0 = st[lane];
            t.r1 = lane < S - 64 ? st[64 + lane] : 0.0;
            return t;
        };
        Staged cur = {};
        if (wave < n_it) cur = request(wave);
        for (int it = wave; it < n_it; it += kPqProducers) {
            const int slot = it & (kPqRing - 1);
            const int gen = it/kPqRing;
            double* buf = ring + slot*TILE;
            row[lane] = cur.r0;
            if (lane < S - 64) row[64 + lane] = cur.r1;
            const cplx o = cur.o;
            if (it + kPqProducers < n_it) cur = request(it + kPqProducers);
            // The records of the row are requested in two batches, each all at once (the compiler otherwise
            // fetches them in pairs with a full LDS round trip between the pairs; all 13 at once take 104 registers).
            const double* st = row;
            const double2_t sbcb = *reinterpret_cast<const double2_t*>(st + seg_rec(lane >> 2) + 1);
            const double2_t head = *reinterpret_cast<const double2_t*>(st);          // dt_g, t_g
            const double sb = sbcb.x, cb = sbcb.y;
            // the tile: q of the 13 distinct entries (all diagonal entries coincide), psi
            const double dtg = head.x;
            cplx ph;
            sincos_pi<false>(om*head.y, &ph.im, &ph.re);
            double sa, ca;
            sincos_pi<false>(0.5*(om*dtg), &sa, &ca);
            const PhasedFrequency pf = phased_frequency(om, dtg, ph, sa, ca);
            double q[DD];
            auto batch = [&](int e0, int e1) __attribute__((always_inline)) {
                double4_t rec[DD];
#pragma unroll
                for (int e = 0; e < DD; ++e) {
                    if (e < e0 || e >= e1 || (e != 0 && e/D == e%D)) continue;
                    rec[e] = *reinterpret_cast<const double4_t*>(st + seg_rec(e));
                }
                asm volatile("" ::: "memory");
                // near a resonance (|x| < thr, rare) the entry is redone under one branch; whether any of the
                // batch's entries is, is ONE running minimum of |x| (a compare, a select and an or per entry
                // were a twelfth of the producer's instructions)
                double xmin = pf.thr;
#pragma unroll
                for (int e = 0; e < DD; ++e) {
                    if (e < e0 || e >= e1 || (e != 0 && e/D == e%D)) continue;
                    const double x = om + rec[e].x;
                    q[e] = fma(pf.sa2, rec[e].z, pf.ca2*rec[e].y)*rcp_fast(x);
                    asm("v_min_f64 %0, |%1|, %0" : "+v"(xmin) : "v"(x));
                }
                if (xmin < pf.thr) {
#pragma unroll
                    for (int e = 0; e < DD; ++e) {
                        if (e < e0 || e >= e1 || (e != 0 && e/D == e%D)) continue;
                        if (fabs(om + rec[e].x) < pf.thr) q[e] = phased_q(pf, rec[e].x, rec[e].y, rec[e].z);
                    }
                }
            };
            batch(0, 8);
            batch(8, 16);
            q[5] = q[0];
            q[10] = q[0];
            q[15] = q[0];
            // the pre-folded W_a of the segment (element `lane` of each operator's 64): requested HERE, with the tile's
            // records dead, and used behind the wait for the slot -- staged a tile ahead like the row it cost 12
            // registers across the whole tile: 168 with spills where the consumers' loop needs 146
            cplx wpre[NC];
            if constexpr (PRE) {
#pragma unroll
                for (int a = 0; a < NC; ++a)
                    wpre[a] = a < n_alpha ? wfold[(static_cast<size_t>(g0 + it)*A + alpha0 + a)*64 + lane] : cplx{0.0, 0.0};
            }
            // the slot's previous tenant (tile it - 8) has been read by every consumer?
            if (gen > 0) {
                int spin = 0;
                for (; spin < spin_limit; ++spin) {
                    if (lds_peek(done + slot) >= gen*kPqConsumers) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (spin == spin_limit && spin_limit != 0) {
                    pq_report_fault(fault_word, kFaultPcProducerWait);
                    spin_limit = 0;
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const double2_t v = {q[m*D + 2*h], q[m*D + 2*h + 1]};
                    *reinterpret_cast<double2_t*>(buf + kPqQ + h*512 + lane*8 + (((m ^ (lane >> 2)) & 3) << 1)) = v;
                }
            {
                const double2_t v = {pf.pr, pf.pi};
                *reinterpret_cast<double2_t*>(buf + kPqPsi + lane*2) = v;
            }
            // W_a[m][n][j] = Bbar_a[m][n] e^{i b_mn} T[n][j], (m, n, j) = this lane's index: folded once per segment
            // by the prologue kernel where the caller gave it a buffer (wfold), else here
            if constexpr (PRE) {
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    const double2_t v = {wpre[a].re, wpre[a].im};
                    *reinterpret_cast<double2_t*>(buf + kPqW + a*128 + lane*2) = v;
                }
            } else {
                const int src_t = lane & (DD - 1);                          // T[n][j]
                const cplx tv = {__shfl(o.re, src_t, 64), __shfl(o.im, src_t, 64)};
                const cplx et = cmul(cplx{cb, sb}, tv);
                const int wslot = ((lane >> 2) & 3)*16 + (lane >> 4)*4 + (lane & 3);   // [n][m][j]
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    const int src_b = DD + a*DD + (lane >> 2);              // Bbar_a[m][n]
                    const cplx bv = {__shfl(o.re, src_b, 64), __shfl(o.im, src_b, 64)};
                    const cplx w = cmul(bv, et);
                    const double2_t v = {w.re, w.im};
                    *reinterpret_cast<double2_t*>(buf + kPqW + a*128 + wslot*2) = v;
                }
            }
            if (lane < DD) {
                const double2_t v = {o.re, o.im};
                *reinterpret_cast<double2_t*>(buf + TOP + lane*2) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FFK_PC_FAULT_INJECT
            if (it < 2*kPqProducers)
#endif
            if (lane == 0) *(volatile lds_int_t*)(ready + slot) = it + 1;
        }
        return;
    }

    // ---- consumer: eight frequencies (two sets of four), all operators, every tile of the block ----------
    // The flag of a later tile and the progress of the SIMD partner (the other consumer of this SIMD) are
    // read a tile ahead, beside the tile's operands: in the usual case -- the producers are ahead -- the loop
    // finds the flag in a register instead of paying an LDS round trip (~600 cycles under load,
    // profiles/r05_b_*).  The SIMD's arbiter serves the older wavefront first, which lets one consumer run
    // ahead until the ring stops it and leaves the other to finish alone: whoever is behind its partner
    // raises its priority (soft lockstep).
    const int octant = wave - kPqProducers;            // frequencies 8 octant .. 8 octant + 7 of the block
    const int m = lane >> 4, f = (lane >> 2) & 3, j = lane & 3;
    // this lane's offsets inside a slot, in doubles
    const int mj = m*4 + j;                                            // W, T: (m, j) = (m, i)
    const int wf = octant*4*kPqSets + f;                               // the lane's frequency of set 0; set s: + 4 s
    const int o_w = kPqW + mj*2;                                       // + a*128 + n*32; T at + NC*128
    const int o_p = kPqPsi + wf*2;                                     // set 1: + 8
    const int o_q0 = kPqQ + wf*8 + ((m ^ ((kPqSets*octant) & 3)) << 1);   // row m in 16-byte slot m ^ ((w >> 2) & 3)
    const int o_q1 = (o_q0 ^ 2) + 32;                                  // set 1; plane h = 1: + 512
    double P[6*NC];                                                    // P_k of (operator a, set s) at 3 (2 a + s) + k
#pragma unroll
    for (int k = 0; k < 6*NC; ++k) P[k] = 0.0;
    if (n_it > 0) {
        // tile 0 published?  (the generated loop waits for every later tile itself)
        int spin = 0;
        for (; spin < spin_limit; ++spin) {
            if (lds_peek(ready) >= 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (spin == spin_limit) {
            pq_report_fault(fault_word, kFaultPcConsumerWait);
            spin_limit = 0;
        }
        asm volatile("" ::: "memory");
        const unsigned ring_b = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring));
        const unsigned flags_b = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ready));
        const int fault_code = PqConsumer<NC>::run(ring_b + o_w*8, ring_b + o_q0*8, ring_b + o_q1*8, ring_b + o_p*8, n_it,
                                                   spin_limit, flags_b, octant, P);
        if (fault_code != 0) pq_report_fault(fault_word, fault_code);
    }
    // D[i = lane >> 4][column lane & 15]: Y_f[i][j] of frequency f = (lane >> 2) & 3 of the set
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        const int alpha = alpha0 + a;
        if (alpha >= alpha_end) break;
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD + mj)*W;
#pragma unroll
        for (int s = 0; s < kPqSets; ++s) {
            const int iw = blockIdx.x*64 + wf + s*4;
            const double p1 = P[3*(kPqSets*a + s)], p2 = P[3*(kPqSets*a + s) + 1], p3 = P[3*(kPqSets*a + s) + 2];
            if (iw < W) out[iw] = cplx{p1 - p2, p3 - (p1 + p2)};
        }
    }
}

template <int NC, bool PRE>
hipError_t launch_pq_as(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                        int chunks, int chunk_len, cplx* Ypart, const cplx* wfold, int alpha_base, int alpha_end,
                        int* fault_word, hipStream_t stream) {
    const int lds = pq_lds_bytes_for(NC);
    auto kern = ctrl_accumulate_pq_kernel<NC, PRE>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const int n = alpha_end - alpha_base;
    const dim3 grid((W + 63)/64, (n + NC - 1)/NC, chunks);
    hipLaunchKernelGGL(kern, grid, dim3((kPqProducers + kPqConsumers)*64), lds, stream, omega, W, segtab,
                       ops, G, A, chunk_len, Ypart, wfold, alpha_base, alpha_end, fault_word);
    return hipGetLastError();
}

template <int NC>
hipError_t launch_pq(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, cplx* Ypart, const cplx* wfold, int alpha_base, int alpha_end,
                     int* fault_word, hipStream_t stream) {
    return wfold != nullptr ? launch_pq_as<NC, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, wfold,
                                                     alpha_base, alpha_end, fault_word, stream)
                            : launch_pq_as<NC, false>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, nullptr,
                                                      alpha_base, alpha_end, fault_word, stream);
}

}  // namespace

int pq_accumulate_lds_bytes(int nc) { return pq_lds_bytes_for(nc); }
int pq_accumulate_waves(int) { return kPqProducers + kPqConsumers; }
bool pq_accumulate_supported(int d, int A) { return d == 4 && A >= 1; }

// Operators in groups of three; what is left over as groups of two (A = 3 k + 1 >= 4: two of them, taking one of
// the threes apart) or, for A = 1, a single one -- never a zero-padded three-operator block (round 5: A = 4 cost
// what A = 6 costs, profiles/r05_c_*).  One launch per group size, the threes first.
PqGroups pq_accumulate_groups(int A) {
    if (A <= 0) return {0, 0, 0};
    if (A == 1) return {0, 0, 1};
    switch (A % 3) {
        case 0: return {A/3, 0, 0};
        case 1: return {(A - 4)/3, 2, 0};
        default: return {(A - 2)/3, 1, 0};
    }
}

hipError_t launch_accumulate_pq(const double* omega, int W, const double* segtab, const cplx* ops,
                                int G, int d, int A, int chunks, int chunk_len, cplx* Ypart,
                                const cplx* wfold, hipStream_t stream) {
    if (d != 4) return hipErrorInvalidValue;
    int* fault_word = kernel_fault_word();
    const PqGroups g = pq_accumulate_groups(A);
    hipError_t err = hipSuccess;
    int base = 0;
    if (g.n3 > 0) {
        err = launch_pq<3>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, wfold, base, base + 3*g.n3,
                           fault_word, stream);
        base += 3*g.n3;
    }
    if (err == hipSuccess && g.n2 > 0) {
        err = launch_pq<2>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, wfold, base, base + 2*g.n2,
                           fault_word, stream);
        base += 2*g.n2;
    }
    if (err == hipSuccess && g.n1 > 0)
        err = launch_pq<1>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, wfold, base, base + g.n1,
                           fault_word, stream);
    return err;
}

}  // namespace ffk
