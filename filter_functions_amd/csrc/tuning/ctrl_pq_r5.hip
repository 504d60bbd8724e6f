// ctrl_pq.hip -- K3q: the control-matrix accumulation for d = 4 (BASELINE config 2, the headline) with
// SPECIALISED wavefronts and the second product on the matrix cores (round 5; replaces ctrl_pc.hip).
//     Y_a(w) = sum_g T_g^dag [ Bbar_a o E_g(w) ] T_g,   E = psi e^{ib} q,  q = 2 sin(a + b)/x REAL   (ffk_math.h)
//     Z_a[m][j] = sum_n q[m][n] W_a[m][n][j]     W_a = Bbar_a e^{ib} T, frequency independent, folded by the producers
//     Y_a[i][j] += sum_m c[m][i] Z_a[m][j]       c = psi conj(T): psi is per frequency, T per segment
// Replaces the reference's hot loop numeric.py:846-869 / :596-609.
//
// A block owns 64 frequencies, a chunk of the segments and up to three operators: 12 wavefronts.
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
//     3 matrix instructions + 5/3 shared = 90.7 issue cycles against the 112 of the round-4 kernel's 28 v_fma_f64,
//     on a mix that holds 2.33 GHz where the pure v_fma_f64 stream is power capped at 2.03
//     (tools/fp64_mix_probe.hip, profiles/r05_a_*).  18 accumulator registers per lane instead of 64: no
//     segment split inside the block, no tree reduction at the end.
//   * The consumer's WHOLE TILE LOOP is one generated asm block (ctrl_pq_consumer.inc, tools/gen_pq_consumer.py),
//     unrolled over the ring's eight slots so that every LDS address is a per-lane base register plus an
//     immediate: the next tile's operands are requested from inside the last set, when their registers are dead,
//     and fly during its matrix instructions and the hand-over; the flag of the tile after that and the SIMD
//     partner's progress are read a tile ahead; every s_waitcnt carries the exact count of younger LDS
//     operations; 16 bookkeeping instructions per tile (a wavefront does not issue in the shadow of its own
//     matrix instructions: each of the 48 of a rolled loop lengthened the consumer's chain,
//     tools/fp64_issue_probe.py, profiles/r05_l_*, r05_m_*).  hipcc moved the last set's vector work behind the
//     requests (168 VGPRs, accumulators spilled in the loop) or serialised the reads.  At 146 VGPRs three of these
//     wavefronts per SIMD leave room for a wavefront of another pass's small kernels (56): the two-pass schedule
//     of the bench keeps overlapping (tests/test_kernel_resources.py).
//   * W_a comes folded from the prologue kernel where the caller owns a buffer for it (PRE, ffk_internal.h
//     g_d4_wfold: ffk_control_matrix_dev / ffk_pipeline_dev), else the producers fold it per tile.
//   * Flags in LDS: ready[slot] written by the slot's producer, done[slot] counted up by the consumers with
//     ds_add, progress[consumer]; the lagging consumer of a SIMD raises its priority (the arbiter serves the
//     oldest wavefront first and would let one run ahead until the ring stops it: soft lockstep).  Every wait is
//     bounded; a wait that runs out is a reported FAULT (ffk_internal.h::kernel_fault_word -> FFK_EKERNEL),
//     the wavefront stops waiting for the rest of the launch and runs to the end so that the grid drains.
//     -DFFK_PC_SPIN_LIMIT=n -DFFK_PC_FAULT_INJECT: the test build whose producers stop publishing after eight
//     tiles (tests/test_gpu_parity.py::test_flag_wait_timeout_is_an_error).
// Same box, bench schedule: round-4 kernel 67.6-68.7 us per step, the per-tile asm block 59.6-59.8, this one 58.0
// (kernel alone 72.6-73.5 -> 63.2-63.7 -> 60.9); the steps in between with their measurements:
// profiles/r05_b_d4_matrix_core_kernel_steps.txt, r05_l_*; where the time is: DESIGN.md section 6.1.
// One or two operators per block (A < 3) run the same arithmetic through a C++ consumer.
#include <algorithm>
#include <cstdlib>

#include "ffk_internal.h"
#ifndef FFK_PQ_SETS            /* sets of four frequencies per consumer: 2 (shipped) or 4 (tuning: profiles/r05_b_*, ab_pq18; needs GEN_PQ_SETS=4 python tools/gen_pq_consumer.py > ctrl_pq_consumer4.inc) */
#define FFK_PQ_SETS 2
#endif
#ifndef FFK_PQ_CONSUMER_INC   /* tuning builds: a block with parts left out (GEN_PQ_DROP) */
#if FFK_PQ_SETS == 4
#define FFK_PQ_CONSUMER_INC "ctrl_pq_consumer4.inc"
#else
#define FFK_PQ_CONSUMER_INC "ctrl_pq_consumer.inc"
#endif
#endif
#include FFK_PQ_CONSUMER_INC   // generated: tools/gen_pq_consumer.py

namespace ffk {

thread_local cplx* g_d4_wfold = nullptr;     // ffk_internal.h

namespace {

constexpr int kPqProducers = 4;   // wavefronts 0..3, one per SIMD
constexpr int kPqSets = FFK_PQ_SETS;   // a consumer owns this many sets of four of the block's 64 frequencies, all operators
constexpr int kPqConsumers = 16/kPqSets;   // wavefronts 4..11: two per SIMD (four sets: 4..7, one per SIMD)
constexpr int kPqRing = 8;        // tile slots: two per producer
#ifndef FFK_PC_SPIN_LIMIT
#define FFK_PC_SPIN_LIMIT (1 << 21)
#endif
constexpr int kPqSpinLimit = FFK_PC_SPIN_LIMIT;

__device__ int* g_pq_fault_word = nullptr;
__device__ __forceinline__ void pq_report_fault(int code) {
    int* fault = g_pq_fault_word;
    if (fault != nullptr && (threadIdx.x & 63) == 0)
        __hip_atomic_store(fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifdef FFK_PQ_CLOCK   /* tuning build: s_memtime stamps of block (0, 0, 0), tools/trace_pq.py */
constexpr int kPqTraceTiles = 64, kPqTraceStamps = 12, kPqTraceWaves = 16;   // 4 C++ stamps + 8 inside the asm block
__device__ unsigned long long g_pq_trace[kPqTraceWaves*(2 + kPqTraceTiles*kPqTraceStamps)];
__device__ unsigned long long g_pq_blocks[3*1024];   // per block: start, end of its last wavefront (100 MHz ticks), HW_ID | XCC_ID << 32
// every wavefront of every launch: start, end (100 MHz ticks), HW_ID | XCC_ID << 32, block | wave << 32 -- a ring over
// the last ~21 launches, for the turn-around of a CU between the blocks of consecutive launches (tools/trace_pq.py cu)
constexpr unsigned kPqWaveRing = 1u << 16;
__device__ unsigned long long g_pq_waves[4*kPqWaveRing];
__device__ unsigned g_pq_waves_head = 0;
#define FFK_PQ_STAMP(it, k)                                                                          \
    do {                                                                                              \
        if (pq_tr != nullptr && (it) < kPqTraceTiles) pq_tr[2 + (it)*kPqTraceStamps + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FFK_PQ_STAMP(it, k)
#endif

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lds_int_t;

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

// PRE: W_a comes folded from the prologue kernel (wfold, ffk_internal.h g_d4_wfold) instead of being folded here
template <int NC, bool PRE>
__global__ __launch_bounds__((kPqProducers + kPqConsumers)*64, kPqSets == 2 ? 3 : 2) void ctrl_accumulate_pq_kernel(
    const double* __restrict__ omega, int W, const double* __restrict__ segtab,
    const cplx* __restrict__ ops, int G, int A, int chunk_len, cplx* __restrict__ Ypart,
    const cplx* __restrict__ wfold) {
    constexpr int D = 4, DD = 16, S = kPqRow, TILE = pq_tile_doubles(NC);
    constexpr int TOP = kPqW + NC*128;                  // the A operands of a slot
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* ring = reinterpret_cast<double*>(lds_raw);
    double* rows = ring + kPqRing*TILE;
    int* ready = reinterpret_cast<int*>(rows + kPqProducers*S);
    int* done = ready + kPqRing;
    int* progress = done + kPqRing;                    // per consumer: the tile it will take next

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int alpha0 = blockIdx.y*NC;
    const int n_alpha = min(NC, A - alpha0);
    const int g0 = blockIdx.z*chunk_len;
    const int g1 = min(G, g0 + chunk_len);
    const int n_it = max(0, g1 - g0);                  // tiles of this block
    int spin_limit = kPqSpinLimit;                     // 0 after a wait of this wavefront has run out

    if (threadIdx.x < 3*kPqRing) *(volatile lds_int_t*)(ready + threadIdx.x) = 0;
    __syncthreads();
#if defined(FFK_PQ_ABLATE) && FFK_PQ_ABLATE == 4   /* tuning: what does the launch cost with no work in it? */
    if (n_it >= 0) return;
#endif
#ifdef FFK_PQ_CLOCK
    unsigned long long* pq_tr = nullptr;
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && wave < kPqTraceWaves) {
        pq_tr = g_pq_trace + wave*(2 + kPqTraceTiles*kPqTraceStamps);
        pq_tr[0] = __builtin_amdgcn_s_memtime();
        pq_tr[1] = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned pq_block = blockIdx.x + gridDim.x*(blockIdx.y + gridDim.y*blockIdx.z);
    const unsigned long long pq_wave_start = __builtin_amdgcn_s_memrealtime();
    auto pq_wave_record = [&]() {
        if (lane != 0) return;
        // (one atomic per wavefront on one address serialises the 3072 wavefronts of a launch: +27 us.)  Slot by launch
        // number -- block 0 counts the launches -- block and wavefront; a wavefront that ends after the next launch's
        // first block has started lands in that launch's slot and is overwritten there: one record lost
        const unsigned seq = __hip_atomic_load(&g_pq_waves_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned at = ((seq & 15u)*4096u + (pq_block*12u + static_cast<unsigned>(wave))) & (kPqWaveRing - 1);
        g_pq_waves[4*at] = pq_wave_start;
        g_pq_waves[4*at + 1] = __builtin_amdgcn_s_memrealtime();
        g_pq_waves[4*at + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                               (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11))) << 32);
        g_pq_waves[4*at + 3] = pq_block | (static_cast<unsigned long long>(wave) << 32);
    };
    if (threadIdx.x == 0 && pq_block == 0) atomicAdd(&g_pq_waves_head, 1u);
    if (threadIdx.x == 0 && pq_block < 1024) {
        g_pq_blocks[3*pq_block] = __builtin_amdgcn_s_memrealtime();
        g_pq_blocks[3*pq_block + 1] = 0;
        g_pq_blocks[3*pq_block + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                      (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11))) << 32);
    }
#endif

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
#ifndef FFK_PQ_PRODUCER_PRIO          /* tuning; FFK_PQ_PRODUCER_PRIO_LATER: from a producer's second tile on */
#define FFK_PQ_PRODUCER_PRIO 0
#endif
#ifndef FFK_PQ_PRODUCER_PRIO_LATER
#define FFK_PQ_PRODUCER_PRIO_LATER FFK_PQ_PRODUCER_PRIO
#endif
        __builtin_amdgcn_s_setprio(FFK_PQ_PRODUCER_PRIO);
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
#if defined(FFK_PQ_ABLATE) && FFK_PQ_ABLATE == 1   /* tuning: only the first round of tiles is generated */
            if (it >= kPqRing) {
                if (lane == 0) *(volatile lds_int_t*)(ready + (it & (kPqRing - 1))) = it + 1;
                continue;
            }
#endif
            const int slot = it & (kPqRing - 1);
            const int gen = it/kPqRing;
            double* buf = ring + slot*TILE;
            FFK_PQ_STAMP(it, 0);
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
            FFK_PQ_STAMP(it, 1);
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
                    pq_report_fault(kFaultPcProducerWait);
                    spin_limit = 0;
                }
                asm volatile("" ::: "memory");
            }
            FFK_PQ_STAMP(it, 2);
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
            // by the prologue kernel where the caller gave it a buffer (ffk_internal.h g_d4_wfold), else here
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
            if (FFK_PQ_PRODUCER_PRIO_LATER != FFK_PQ_PRODUCER_PRIO && it == wave)
                __builtin_amdgcn_s_setprio(FFK_PQ_PRODUCER_PRIO_LATER);
            FFK_PQ_STAMP(it, 3);
        }
#ifdef FFK_PQ_CLOCK
        pq_wave_record();
#endif
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
    const int me = octant, partner = kPqSets == 2 ? me ^ 4 : me;      // (four sets: one consumer per SIMD, no partner)
    int flag_v = 0, partner_v = 0;
    int prio = 0;
    auto await = [&](int it) __attribute__((always_inline)) {     // tile `it` published?
        if (__builtin_amdgcn_readfirstlane(flag_v) < it + 1) {
            int spin = 0;
            for (; spin < spin_limit; ++spin) {
                if (lds_peek(ready + (it & (kPqRing - 1))) >= it + 1) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin == spin_limit && spin_limit != 0) {
                pq_report_fault(kFaultPcConsumerWait);
                spin_limit = 0;
            }
        }
        asm volatile("" ::: "memory");
    };
    auto set_priority = [&](int it) __attribute__((always_inline)) {
        const int want = __builtin_amdgcn_readfirstlane(partner_v) > it + 1 ? 1 : 0;
        if (want != prio) {
            prio = want;
            if (want) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    };
    // this lane's offsets inside a slot, in doubles
    const int mj = m*4 + j;                                            // W, T: (m, j) = (m, i)
    const int wf = octant*4*kPqSets + f;                               // the lane's frequency of set 0; set s: + 4 s
    const int o_w = kPqW + mj*2;                                       // + a*128 + n*32; T at + NC*128
    const int o_p = kPqPsi + wf*2;                                     // set 1: + 8
    const int o_q0 = kPqQ + wf*8 + ((m ^ ((kPqSets*octant) & 3)) << 1);   // row m in 16-byte slot m ^ ((w >> 2) & 3)
    auto o_q = [&](int s) { return (o_q0 ^ (s << 1)) + s*32; };        // set s; plane h = 1: + 512
    cplx y[NC][kPqSets];                                               // the block's sums, combined

    if constexpr (NC == 3) {
        // ---- the tile loop as generated assembly (tools/gen_pq_consumer.py has the register map) ---------
        typedef double double8_t __attribute__((ext_vector_type(8)));
        double8_t W0v, W1v, W2v;        // v[24:39], v[40:55], v[56:71]: W_a[n] = (re, im), n = 0..3
        double8_t Qv;                   // v[72:87]: q01, q23, psi, (tr, ti)
        // accumulators P_k of (a, s), index 3 (kPqSets a + s) + k: two sets v[88:123], four sets v[88:159]
        double8_t A0 = 0.0, A1 = 0.0;   // v[88:103], v[104:119]
#if FFK_PQ_SETS == 4
        double8_t A2 = 0.0, A3 = 0.0;   // v[120:135], v[136:151]
        double4_t A4 = 0.0;             // v[152:159]
#else
        double2_t A2 = 0.0;             // v[120:123]
#endif
        const unsigned ring_b = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring));
        const unsigned flags_b = static_cast<unsigned>(reinterpret_cast<uintptr_t>(ready));
        const unsigned b_w = ring_b + o_w*8, b_p = ring_b + o_p*8, b_q0 = ring_b + o_q0*8, b_q1 = ring_b + o_q(1)*8;
#if FFK_PQ_SETS == 4
        const unsigned b_q2 = ring_b + o_q(2)*8, b_q3 = ring_b + o_q(3)*8;
#endif
        const unsigned a_partner = flags_b + (2*kPqRing + partner)*4, a_prog = flags_b + (2*kPqRing + me)*4;
        if (n_it > 0) {
            await(0);
            asm volatile(FFK_PQ_CONSUMER_PROLOGUE_ASM
                         : "=&{v[24:39]}"(W0v), "=&{v[40:55]}"(W1v), "=&{v[56:71]}"(W2v), "=&{v[72:87]}"(Qv)
                         : [a_w] "v"(b_w), [a_p] "v"(b_p), [a_q0] "v"(b_q0)
                         : "memory");
        }
#if defined(FFK_PQ_CONSUMER_LOOP_ASM) && (!defined(FFK_PQ_CLOCK) || defined(FFK_PQ_LOOP_CLOCK)) && !defined(FFK_PQ_TILE_BLOCKS) && FFK_PQ_SETS == 2
        // ---- the whole tile loop as ONE block (FFK_PQ_TILE_BLOCKS: the per-tile form below, for A/B runs) ----
        static_assert(FFK_PQ_TILE_BYTES == TILE*8, "tools/gen_pq_consumer.py and pq_tile_doubles() disagree");
        if (n_it > 0) {
            typedef int int4_t __attribute__((ext_vector_type(4)));
            int4_t sarg = {n_it, spin_limit, static_cast<int>(flags_b), me};
            const int4_t varg = {static_cast<int>(b_w), static_cast<int>(b_q0), static_cast<int>(b_q1),
                                 static_cast<int>(b_p)};
            int fault_code;
#ifdef FFK_PQ_LOOP_CLOCK   /* tuning (GEN_PQ_LOOP_CLOCK=1 block via -DFFK_PQ_CONSUMER_INC): s_memtime around the loop */
            unsigned long long loop_t0, loop_t1;
#define FFK_PQ_LOOP_STAMPS , "={s[52:53]}"(loop_t0), "={s[54:55]}"(loop_t1)
#else
#define FFK_PQ_LOOP_STAMPS
#endif
            asm volatile(FFK_PQ_CONSUMER_LOOP_ASM
                         : "+{v[24:39]}"(W0v), "+{v[40:55]}"(W1v), "+{v[56:71]}"(W2v), "+{v[72:87]}"(Qv),
                           "+{v[88:103]}"(A0), "+{v[104:119]}"(A1), "+{v[120:123]}"(A2), "+{s[36:39]}"(sarg),
                           "={s48}"(fault_code) FFK_PQ_LOOP_STAMPS
                         : "{v[138:141]}"(varg)
                         : FFK_PQ_LOOP_CLOBBERS);
#undef FFK_PQ_LOOP_STAMPS
#ifdef FFK_PQ_LOOP_CLOCK
            if (pq_tr != nullptr) {
                pq_tr[2] = loop_t0;
                pq_tr[3] = loop_t1;
                pq_tr[4] = static_cast<unsigned long long>(n_it);
                pq_tr[5] = __builtin_amdgcn_s_memtime();
                pq_tr[6] = __builtin_amdgcn_s_memrealtime();
            }
#endif
            if (fault_code != 0) pq_report_fault(fault_code);
            (void)a_partner; (void)a_prog; (void)flag_v; (void)partner_v; (void)prio;
        }
#else
        for (int it = 0; it < n_it; ++it) {
            FFK_PQ_STAMP(it, 0);
            const bool last = it + 1 == n_it;
            // the block requests tile it + 1's operands: its flag (read by the block of tile it - 1) must be up.
            // (Waiting here holds back nothing the producers need: they wait for tile it - 7 at most.)
            if (!last) await(it + 1);
            set_priority(it);
            const unsigned cur = static_cast<unsigned>((it & (kPqRing - 1))*TILE*8);
            // (the last tile requests its own slot once more: one asm statement in the loop -- with a second
            // variant behind a branch the compiler parks the loop-carried operands elsewhere and copies all 100
            // registers in front of every block)
            const unsigned nxt = last ? cur : static_cast<unsigned>(((it + 1) & (kPqRing - 1))*TILE*8);
            const unsigned a_w = b_w + nxt, a_p = b_p + nxt, a_q0 = b_q0 + nxt;
            const unsigned a_q1 = b_q1 + cur, a_p1 = b_p + cur + 64;
#if FFK_PQ_SETS == 4
            const unsigned a_q2 = b_q2 + cur, a_q3 = b_q3 + cur;
#endif
            const unsigned a_flag = flags_b + ((it + 2) & (kPqRing - 1))*4;
            const unsigned a_done = flags_b + (kPqRing + (it & (kPqRing - 1)))*4;
            const int progress_v = it + 1, one = 1;
            FFK_PQ_STAMP(it, 1);
#if FFK_PQ_SETS == 4
#define FFK_PQ_ASM_ACC "+{v[120:135]}"(A2), "+{v[136:151]}"(A3), "+{v[152:159]}"(A4)
#define FFK_PQ_ASM_SETS , [a_q2] "v"(a_q2), [a_q3] "v"(a_q3)
#else
#define FFK_PQ_ASM_ACC "+{v[120:123]}"(A2)
#define FFK_PQ_ASM_SETS
#endif
#ifdef FFK_PQ_CLOCK   /* build with -DFFK_PQ_CONSUMER_INC pointing at a GEN_PQ_STAMPS=1 block */
            unsigned long long ts[8];
#define FFK_PQ_ASM_STAMPS , [t0] "=s"(ts[0]), [t1] "=s"(ts[1]), [t2] "=s"(ts[2]), [t3] "=s"(ts[3]), [t4] "=s"(ts[4]), \
                            [t5] "=s"(ts[5]), [t6] "=s"(ts[6]), [t7] "=s"(ts[7])
#else
#define FFK_PQ_ASM_STAMPS
#endif
#define FFK_PQ_ASM_OPERANDS                                                                                    \
    : "+{v[24:39]}"(W0v), "+{v[40:55]}"(W1v), "+{v[56:71]}"(W2v), "+{v[72:87]}"(Qv), "+{v[88:103]}"(A0),       \
      "+{v[104:119]}"(A1), FFK_PQ_ASM_ACC, [flag] "=&v"(flag_v), [partner] "=&v"(partner_v) FFK_PQ_ASM_STAMPS    \
    : [a_w] "v"(a_w), [a_p] "v"(a_p), [a_q0] "v"(a_q0), [a_q1] "v"(a_q1), [a_p1] "v"(a_p1), [a_flag] "v"(a_flag), \
      [a_partner] "v"(a_partner), [a_done] "v"(a_done), [a_prog] "v"(a_prog), [progress] "v"(progress_v),     \
      [one] "v"(one) FFK_PQ_ASM_SETS                                                                           \
    : FFK_PQ_CONSUMER_CLOBBERS
            asm volatile(FFK_PQ_CONSUMER_ASM FFK_PQ_ASM_OPERANDS);
#undef FFK_PQ_ASM_OPERANDS
#undef FFK_PQ_ASM_ACC
#undef FFK_PQ_ASM_SETS
#undef FFK_PQ_ASM_STAMPS
#ifdef FFK_PQ_CLOCK
            if (pq_tr != nullptr && it < kPqTraceTiles)
                for (int k = 0; k < 8; ++k) pq_tr[2 + it*kPqTraceStamps + 4 + k] = ts[k];
#endif
            FFK_PQ_STAMP(it, 2);
            FFK_PQ_STAMP(it, 3);
        }
#endif
        // (the compiler does not know that matrix instructions wrote the accumulators: keep the vector
        // instructions that read them next out of their shadow)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                     : "+{v[24:39]}"(W0v), "+{v[40:55]}"(W1v), "+{v[56:71]}"(W2v), "+{v[72:87]}"(Qv)
                     :
                     : "memory");
#if FFK_PQ_SETS == 4
        const double P[36] = {A0[0], A0[1], A0[2], A0[3], A0[4], A0[5], A0[6], A0[7], A1[0], A1[1], A1[2], A1[3],
                              A1[4], A1[5], A1[6], A1[7], A2[0], A2[1], A2[2], A2[3], A2[4], A2[5], A2[6], A2[7],
                              A3[0], A3[1], A3[2], A3[3], A3[4], A3[5], A3[6], A3[7], A4[0], A4[1], A4[2], A4[3]};
#else
        const double P[18] = {A0[0], A0[1], A0[2], A0[3], A0[4], A0[5], A0[6], A0[7], A1[0], A1[1], A1[2], A1[3],
                              A1[4], A1[5], A1[6], A1[7], A2[0], A2[1]};
#endif
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int s = 0; s < kPqSets; ++s) {
                const double p1 = P[3*(kPqSets*a + s)], p2 = P[3*(kPqSets*a + s) + 1], p3 = P[3*(kPqSets*a + s) + 2];
                y[a][s] = cplx{p1 - p2, p3 - (p1 + p2)};
            }
    } else {
        // ---- one or two operators per block: the same arithmetic in C++ (not the headline's path) --------
        double acc[NC][kPqSets][3];
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int s = 0; s < kPqSets; ++s) acc[a][s][0] = acc[a][s][1] = acc[a][s][2] = 0.0;
        for (int it = 0; it < n_it; ++it) {
            const int slot = it & (kPqRing - 1);
            await(it);
            set_priority(it);
            const double* buf = ring + slot*TILE;
            const double2_t t = *reinterpret_cast<const double2_t*>(buf + o_w + NC*128);
            double2_t w[NC][4];
#pragma unroll
            for (int a = 0; a < NC; ++a)
#pragma unroll
                for (int n = 0; n < 4; ++n) w[a][n] = *reinterpret_cast<const double2_t*>(buf + o_w + a*128 + n*32);
            flag_v = *(const volatile lds_int_t*)(ready + ((it + 1) & (kPqRing - 1)));
            partner_v = *(const volatile lds_int_t*)(progress + partner);
#pragma unroll
            for (int s = 0; s < kPqSets; ++s) {
                const int oq = o_q(s);
                const double2_t q01 = *reinterpret_cast<const double2_t*>(buf + oq);
                const double2_t q23 = *reinterpret_cast<const double2_t*>(buf + oq + 512);
                const double2_t psi = *reinterpret_cast<const double2_t*>(buf + o_p + s*8);
                // c = psi conj(T[m][i]) = (pr tr + pi ti) + i (pi tr - pr ti)
                const double cr = fma(psi.x, t.x, psi.y*t.y), ci = fma(-psi.x, t.y, psi.y*t.x);
                const double cs = cr + ci;
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    double zr = q01.x*w[a][0].x, zi = q01.x*w[a][0].y;
                    zr = fma(q01.y, w[a][1].x, zr);
                    zi = fma(q01.y, w[a][1].y, zi);
                    zr = fma(q23.x, w[a][2].x, zr);
                    zi = fma(q23.x, w[a][2].y, zi);
                    zr = fma(q23.y, w[a][3].x, zr);
                    zi = fma(q23.y, w[a][3].y, zi);
                    const double zs = zr + zi;
                    acc[a][s][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, zr, acc[a][s][0], 0, 0, 0);
                    acc[a][s][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ci, zi, acc[a][s][1], 0, 0, 0);
                    acc[a][s][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(cs, zs, acc[a][s][2], 0, 0, 0);
                }
            }
            // done with the slot (LDS operations of a wavefront execute in order: no wait needed)
            asm volatile("" ::: "memory");
            if (lane == 0) {
                *(volatile lds_int_t*)(progress + me) = it + 1;
                __hip_atomic_fetch_add(done + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int s = 0; s < kPqSets; ++s)
                y[a][s] = cplx{acc[a][s][0] - acc[a][s][1], acc[a][s][2] - (acc[a][s][0] + acc[a][s][1])};
    }
    // D[i = lane >> 4][column lane & 15]: Y_f[i][j] of frequency f = (lane >> 2) & 3 of the set
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        const int alpha = alpha0 + a;
        if (alpha >= A) break;
        cplx* out = Ypart + ((static_cast<size_t>(blockIdx.z)*A + alpha)*DD + mj)*W;
#pragma unroll
        for (int s = 0; s < kPqSets; ++s) {
            const int iw = blockIdx.x*64 + wf + s*4;
#if defined(FFK_PQ_ABLATE) && FFK_PQ_ABLATE == 3   /* tuning: the block's results are not stored (a never-true condition keeps them alive) */
            if (iw < W && y[a][s].re == 1.2345e300) out[iw] = y[a][s];
#else
            if (iw < W) out[iw] = y[a][s];
#endif
        }
    }
#ifdef FFK_PQ_CLOCK
    if (lane == 0 && pq_block < 1024) atomicMax(&g_pq_blocks[3*pq_block + 1], __builtin_amdgcn_s_memrealtime());
    pq_wave_record();
#endif
}

template <int NC, bool PRE>
hipError_t launch_pq_as(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                        int chunks, int chunk_len, cplx* Ypart, const cplx* wfold, hipStream_t stream) {
    const int lds = pq_lds_bytes_for(NC);
    (void)kernel_fault_word();
    auto kern = ctrl_accumulate_pq_kernel<NC, PRE>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    const dim3 grid((W + 63)/64, (A + NC - 1)/NC, chunks);
    hipLaunchKernelGGL(kern, grid, dim3((kPqProducers + kPqConsumers)*64), lds, stream, omega, W, segtab,
                       ops, G, A, chunk_len, Ypart, wfold);
    return hipGetLastError();
}

template <int NC>
hipError_t launch_pq(const double* omega, int W, const double* segtab, const cplx* ops, int G, int A,
                     int chunks, int chunk_len, cplx* Ypart, hipStream_t stream) {
    const cplx* wfold = g_d4_wfold;
    return wfold != nullptr
               ? launch_pq_as<NC, true>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, wfold, stream)
               : launch_pq_as<NC, false>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, nullptr, stream);
}

}  // namespace

hipError_t pq_bind_fault_word(int* device_pointer) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pq_fault_word), &device_pointer, sizeof device_pointer);
}
int pq_accumulate_lds_bytes(int nc) { return pq_lds_bytes_for(nc); }
int pq_accumulate_waves(int) { return kPqProducers + kPqConsumers; }
bool pq_accumulate_supported(int d, int A) { return d == 4 && A >= 1; }
int pq_accumulate_ops_per_block(int A) { return A >= 3 ? 3 : A; }

hipError_t launch_accumulate_pq(const double* omega, int W, const double* segtab, const cplx* ops,
                                int G, int d, int A, int nc, int chunks, int chunk_len, cplx* Ypart,
                                hipStream_t stream) {
    if (d != 4) return hipErrorInvalidValue;
    switch (nc) {
        case 1: return launch_pq<1>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 2: return launch_pq<2>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        case 3: return launch_pq<3>(omega, W, segtab, ops, G, A, chunks, chunk_len, Ypart, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ffk

#ifdef FFK_PQ_CLOCK
// (tuning build only, not in include/ffk.h) the last launch's stamps of block (0, 0, 0)
extern "C" int ffk_debug_pq_trace(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_pq_trace), sizeof(ffk::g_pq_trace)) != hipSuccess;
}
extern "C" int ffk_debug_pq_blocks(unsigned long long* out, int n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_pq_blocks), sizeof(unsigned long long)*3*n_blocks) != hipSuccess;
}
extern "C" int ffk_debug_pq_waves(unsigned long long* out, unsigned* head) {
    if (hipMemcpyFromSymbol(head, HIP_SYMBOL(ffk::g_pq_waves_head), sizeof(unsigned)) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ffk::g_pq_waves), sizeof(ffk::g_pq_waves)) != hipSuccess;
}
extern "C" int ffk_debug_pq_trace_words(void) { return static_cast<int>(sizeof(ffk::g_pq_trace)/8); }
#endif
