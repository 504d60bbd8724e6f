#!/usr/bin/env python3
"""What does ONE FP64 instruction cost to issue on an MI355X, and does the choice of registers matter?

tools/fp64_mix_probe.hip found that whole mixes of the d = 4 consumer reach ~0.8 of the nominal issue rate
(4 cycles per v_fma_f64, 16 per v_mfma_f64_4x4x4_4b) even on synthetic operands.  This probe takes the mix
apart: straight-line blocks of one instruction kind with FIXED physical registers (inline asm), no memory
traffic inside the timed loop, s_memtime around it, 1 / 2 / 3 / 4 wavefronts per SIMD.  Register patterns:
the accumulator, the two sources on the same or on different register pairs modulo 4 (the vector register
file of earlier GCN parts has four banks by register number), and the exact operand patterns of the shipped
consumer block (the round-5 generator, tools/tuning/gen_pq_consumer_r5.py).

    python3 tools/fp64_issue_probe.py            # writes build/probe/fp64_issue.hip, compiles, runs, prints a table
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'build', 'probe')


def v(r):
    return f'v[{r}:{r + 1}]'


def block_fma(acc_base, acc_stride, a, b, n=32, nacc=8):
    return [f'v_fma_f64 {v(acc_base + acc_stride*(i % nacc))}, {v(a)}, {v(b)}, {v(acc_base + acc_stride*(i % nacc))}'
            for i in range(n)]


def block_mul(dst_base, stride, a, b, n=32, nd=8):
    return [f'v_mul_f64 {v(dst_base + stride*(i % nd))}, {v(a)}, {v(b)}' for i in range(n)]


def block_mfma(acc_base, acc_stride, a, b, n=18, nacc=9):
    return [f'v_mfma_f64_4x4x4_4b_f64 {v(acc_base + acc_stride*(i % nacc))}, {v(a)}, {v(b)}, '
            f'{v(acc_base + acc_stride*(i % nacc))}' for i in range(n)]


def consumer_vector():
    """the 32 vector instructions of one set of the shipped block, its registers (operands preloaded)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'tuning'))
    import gen_pq_consumer_r5 as g
    st = g.Stream([])
    g.vector_part(st, 0, None)
    return [l for l in st.lines if l.startswith('v_')]


def consumer_set():
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'tuning'))
    import gen_pq_consumer_r5 as g
    st = g.Stream([])
    g.vector_part(st, 0, None)
    g.matrix_part(st, 0)
    return [l for l in st.lines if l.startswith('v_')]


def consumer_tile(drop=()):
    """The shipped per-tile block with its operands in fixed registers (the whole-loop form's map): every LDS
    address points into a 32 KB window of the block's LDS, every flag is up -- the block as the kernel runs it, minus
    the producers and minus any waiting for a tile.  `drop`: 'reads' (no ds_read_b128 and no waits), 'handover'
    (no flag reads, no ds_add / ds_write), 'mfma', 'vector'."""
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'tuning'))
    import gen_pq_consumer_r5 as g
    g.OPS = dict(g.LOOP_V)
    lines = g.build(False).lines
    g.OPS = dict(g.TILE_OPERANDS)
    out = []
    for l in lines:
        if 'reads' in drop and (l.startswith('ds_read_b128') or l.startswith('s_waitcnt')):
            continue
        if 'handover' in drop and (l.startswith('ds_read_b32') or l.startswith('ds_add') or l.startswith('ds_write')
                                   or 'exec' in l):
            continue
        if 'mfma' in drop and 'mfma' in l:
            continue
        if 'vector' in drop and l.startswith('v_') and 'mfma' not in l:
            continue
        out.append(l)
    if 'reads' in drop and 'handover' not in drop:
        out.append('s_waitcnt lgkmcnt(4)')
    return out


def consumer_loop(lockstep=0):
    """the shipped whole-loop block (flags all up: no tile is ever waited for)"""
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'tuning'))
    import gen_pq_consumer_r5 as g
    g.LOCKSTEP = lockstep
    lines = g.build_loop().lines
    g.LOCKSTEP = 0
    return lines


def interleave(x, y):
    out = []
    for i in range(max(len(x), len(y))):
        if i < len(x):
            out.append(x[i])
        if i < len(y):
            out.append(y[i])
    return out


def patterns():
    p = []
    # accumulators v0, v4, .. (pair 0 mod 4) unless said otherwise; sources at v[64..79]
    p.append(('fma  acc%4=0 a%4=0 b%4=0', block_fma(0, 4, 64, 68)))
    p.append(('fma  acc%4=0 a%4=2 b%4=0', block_fma(0, 4, 66, 68)))
    p.append(('fma  acc%4=0 a%4=0 b%4=2', block_fma(0, 4, 64, 70)))
    p.append(('fma  acc%4=0 a%4=2 b%4=2', block_fma(0, 4, 66, 70)))
    p.append(('fma  acc%4=2 a%4=0 b%4=0', block_fma(2, 4, 64, 68)))
    p.append(('fma  acc alternating pairs, a%4=0 b%4=2', block_fma(0, 2, 64, 70)))
    p.append(('fma  a == b (two distinct operands)', block_fma(0, 4, 64, 64)))
    p.append(('mul  dst%4=0 a%4=0 b%4=0', block_mul(0, 4, 64, 68)))
    p.append(('mul  dst%4=0 a%4=0 b%4=2', block_mul(0, 4, 64, 70)))
    p.append(('mfma acc%4=0 a%4=0 b%4=0', block_mfma(0, 4, 64, 68)))
    p.append(('mfma acc%4=0 a%4=2 b%4=0', block_mfma(0, 4, 66, 68)))
    p.append(('mfma acc%4=2 a%4=0 b%4=0', block_mfma(2, 4, 64, 68)))
    p.append(('mfma acc alternating pairs a%4=0 b%4=2', block_mfma(0, 2, 64, 70)))
    p.append(('mfma 9 + fma 9 interleaved (independent)', interleave(block_mfma(0, 4, 64, 70, 9), block_fma(40, 2, 66, 68, 9))))
    p.append(('mfma 9 then fma 32', block_mfma(0, 4, 64, 70, 9) + block_fma(40, 2, 66, 68, 32)))
    p.append(('consumer: vector part of one set (32)', consumer_vector()))
    p.append(('consumer: one set (32 vector + 3 add + 9 matrix)', consumer_set()))
    return p


def tile_patterns():
    return [('tile: the shipped block', consumer_tile()),
            ('tile: no operand reads, no waits', consumer_tile(('reads',))),
            ('tile: no flags, no hand-over', consumer_tile(('handover',))),
            ('tile: neither', consumer_tile(('reads', 'handover'))),
            ('tile: no matrix instructions', consumer_tile(('mfma',))),
            ('tile: no vector instructions', consumer_tile(('vector',)))]


SOURCE_HEAD = r'''// GENERATED by tools/fp64_issue_probe.py
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Stamp { unsigned long long cycles, ticks, start, end; };
#define ALL_V "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
  "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
  "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59", \
  "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
  "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99", \
  "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116", \
  "v117","v118","v119","v120","v121","v122","v123"
'''

KERNEL = r'''
__global__ __launch_bounds__(1024) void probe_%(k)d(Stamp* stamps, int iters, double seed) {
    // every register of the block holds a small finite number (the sums stay finite: |x| < 1e-3, products shrink)
    asm volatile(%(init)s ::"v"(seed) : ALL_V);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) asm volatile(%(body)s ::: ALL_V);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x*16 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0, r0, r1};
}
'''


TILE_KERNEL = r'''
__global__ __launch_bounds__(768) void tile_%(k)d(Stamp* stamps, int iters, double seed) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    for (int i = threadIdx.x; i < 12800; i += blockDim.x) lds[i] = 1e-3*((i*37) %% 101);
    __syncthreads();
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // operands of the block: W/T at [0, 3.1 KB) + lane, q / psi behind, flags at 64 KB (all of them "published")
    const unsigned a_w = base + (lane & 15)*16, a_q = base + 8192 + wave*256 + (lane >> 2)*16, a_p = base + 16384 + wave*128 + (lane & 12)*4;
    const unsigned flags = base + 65536 + wave*64;
    asm volatile(%(init)s
                 "v_mov_b32 v124, %%1\n\tv_mov_b32 v125, %%2\n\tv_mov_b32 v126, %%3\n\tv_mov_b32 v127, %%2\n\tv_mov_b32 v128, %%3\n\t"
                 "v_mov_b32 v129, %%4\n\tv_add_u32 v130, 4, %%4\n\tv_mov_b32 v131, 1\n\tv_add_u32 v132, 8, %%4\n\t"
                 "v_add_u32 v133, 12, %%4\n\tv_mov_b32 v134, 1\n\t"
                 ::"v"(seed), "v"(a_w), "v"(a_q), "v"(a_p), "v"(flags) : ALL_V, "v124", "v125", "v126", "v127", "v128", "v129",
                   "v130", "v131", "v132", "v133", "v134", "v135", "v136");
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(%(body)s ::: ALL_V, "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133",
                     "v134", "v135", "v136", "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x*16 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0, r0, r1};
}
'''


LOOP_KERNEL = r'''
__global__ __launch_bounds__(768) void loop_%(k)d(Stamp* stamps, int iters, double seed) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int* flags = reinterpret_cast<int*>(lds + 12544);                 // behind the eight slots
    for (int i = threadIdx.x; i < 12544; i += blockDim.x) lds[i] = 1e-3*((i*37) %% 101);
    if (threadIdx.x < 32) flags[threadIdx.x] = threadIdx.x < 8 ? (1 << 30) : 0;
    __syncthreads();
    const int base = static_cast<int>(reinterpret_cast<uintptr_t>(lds));
    const int lane = threadIdx.x & 63, octant = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 7;
    const int m = lane >> 4, f = (lane >> 2) & 3, j = lane & 3, mj = m*4 + j, wf = octant*8 + f;
    const int o_q0 = wf*8 + ((m ^ ((2*octant) & 3)) << 1);
    typedef int int4_t __attribute__((ext_vector_type(4)));
    int4_t sarg = {iters, 1 << 20, base + 12544*8, octant};
    const int4_t varg = {base + (1152 + mj*2)*8, base + o_q0*8, base + ((o_q0 ^ 2) + 32)*8, base + (1024 + wf*2)*8};
    int fault;
    asm volatile(%(init)s ::"v"(seed) : ALL_V);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile(%(body)s : "+{s[36:39]}"(sarg), "={s48}"(fault) : "{v[138:141]}"(varg)
                 : ALL_V, "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136",
                   "v137", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s49", "s50", "s51", "scc", "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x*16 + (threadIdx.x >> 6)] = Stamp{c1 - c0 + (fault ? 1ull << 40 : 0), r1 - r0, r0, r1};
}
'''


def c_string(lines):
    return '\n        '.join('"' + l + '\\n\\t"' for l in lines)


def main():
    os.makedirs(OUT, exist_ok=True)
    pats = [(name, body*4) for name, body in patterns()]        # (loop overhead below 1 %)
    src = [SOURCE_HEAD]
    init = []
    for r in range(0, 124, 2):
        init.append(f'v_mov_b32 v{r}, {r + 3}')
        init.append(f'v_cvt_f64_i32 v[{r}:{r + 1}], v{r}')
        init.append(f'v_mul_f64 v[{r}:{r + 1}], v[{r}:{r + 1}], %0')
    for k, (name, body) in enumerate(pats):
        src.append(KERNEL % dict(k=k, init=c_string(init), body=c_string(body)))
    tiles = tile_patterns()
    for k, (name, body) in enumerate(tiles):
        src.append(TILE_KERNEL % dict(k=k, init=c_string(init), body=c_string(body)))
    loops = [('loop: the shipped whole-loop block', consumer_loop(0)), ('loop: priority never raised', consumer_loop(1 << 28))]
    for k, (name, body) in enumerate(loops):
        src.append(LOOP_KERNEL % dict(k=k, init=c_string(init), body=c_string(body)))
    src.append('struct Pattern { const char* name; void (*kernel)(Stamp*, int, double); int n; };\n')
    src.append('static Pattern patterns[] = {\n' + ''.join(
        f'    {{"{name}", probe_{k}, {len(body)}}},\n' for k, (name, body) in enumerate(pats)) + '};\n')
    src.append('static Pattern loops[] = {\n' + ''.join(
        f'    {{"{name}", loop_{k}, 1}},\n' for k, (name, body) in enumerate(loops)) + '};\n')
    src.append('static Pattern tiles[] = {\n' + ''.join(
        f'    {{"{name}", tile_{k}, 1}},\n' for k, (name, body) in enumerate(tiles)) + '};\n')
    src.append(r'''
int main() {
    Stamp* stamps;
    const int blocks = 256, iters = 2000;
    CHECK(hipMalloc(&stamps, blocks*16*sizeof(Stamp)));
    std::vector<Stamp> h(blocks*16);
    printf("%-52s %5s | cycles per instruction at 1, 2, 3, 4 wavefronts per SIMD (median wavefront; issue cycles of the SIMD = x / waves)\n", "pattern", "instr");
    for (auto& p : patterns) {
        printf("%-52s %5d |", p.name, p.n);
        for (int waves = 1; waves <= 4; ++waves) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(p.kernel, dim3(blocks), dim3(256*waves), 0, 0, stamps, iters, 1e-4);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), stamps, h.size()*sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc; double ghz = 0; int cnt = 0;
            unsigned long long first = ~0ull, last = 0, longest = 0;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4*waves; ++w) {
                const Stamp& s = h[b*16 + w];
                cyc.push_back(double(s.cycles)/(double(iters)*p.n));
                ghz += double(s.cycles)/double(s.ticks)*0.1; ++cnt;
                first = std::min(first, s.start); last = std::max(last, s.end); longest = std::max(longest, s.ticks);
            }
            std::sort(cyc.begin(), cyc.end());
            // span: the launch from the first wavefront's start to the last one's end over the longest wavefront
            // (1.0 = every block ran at the same time); chip: instructions per microsecond of the whole launch
            printf("  %6.2f (/w %5.2f, %.2f GHz, span %.2f, %.0f/us)", cyc[cyc.size()/2], cyc[cyc.size()/2]/waves, ghz/cnt,
                   double(last - first)/double(longest), double(blocks)*4*waves*iters*p.n/(double(last - first)*0.01));
        }
        printf("\n");
    }
    // the consumer's tile: cycles per TILE and wavefront; 12 wavefronts = the kernel's block (4 of them here do the same work
    // instead of producing), 8 = its consumers alone
    printf("\n%-40s | cycles per tile at 4, 8, 12 wavefronts per CU (median wavefront)\n", "the consumer's tile (issue: 2 x 283 = 566)");
    for (auto& p : tiles) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(p.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 102400));
        printf("%-40s |", p.name);
        for (int waves = 1; waves <= 3; ++waves) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(p.kernel, dim3(blocks), dim3(256*waves), 102400, 0, stamps, 4000, 1e-4);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), stamps, h.size()*sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc; double ghz = 0; int cnt = 0;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4*waves; ++w) {
                const Stamp& s = h[b*16 + w];
                cyc.push_back(double(s.cycles)/4000.0);
                ghz += double(s.cycles)/double(s.ticks)*0.1; ++cnt;
            }
            std::sort(cyc.begin(), cyc.end());
            printf("  %7.0f [%6.0f .. %6.0f] (%.2f GHz)", cyc[cyc.size()/2], cyc.front(), cyc.back(), ghz/cnt);
        }
        printf("\n");
    }
    printf("\n%-40s | cycles per tile at 4, 8 wavefronts per CU (median wavefront) -- 8 = the kernel's consumers, partners on one SIMD\n", "the consumer's loop");
    for (auto& p : loops) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(p.kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 102400));
        printf("%-40s |", p.name);
        for (int waves = 1; waves <= 2; ++waves) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(p.kernel, dim3(blocks), dim3(256*waves), 102400, 0, stamps, 4000, 1e-4);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), stamps, h.size()*sizeof(Stamp), hipMemcpyDeviceToHost));
            std::vector<double> cyc; double ghz = 0; int cnt = 0;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4*waves; ++w) {
                const Stamp& s = h[b*16 + w];
                cyc.push_back(double(s.cycles)/4000.0);
                ghz += double(s.cycles & ((1ull << 40) - 1))/double(s.ticks)*0.1; ++cnt;
            }
            std::sort(cyc.begin(), cyc.end());
            printf("  %7.0f [%6.0f .. %6.0f] (%.2f GHz)", cyc[cyc.size()/2], cyc.front(), cyc.back(), ghz/cnt);
        }
        printf("\n");
    }
    return 0;
}
''')
    path = os.path.join(OUT, 'fp64_issue.hip')
    with open(path, 'w') as f:
        f.write(''.join(src))
    exe = os.path.join(OUT, 'fp64_issue')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O2', path, '-o', exe])
    if '--build-only' in sys.argv:
        return
    subprocess.check_call([exe])


if __name__ == '__main__':
    main()
