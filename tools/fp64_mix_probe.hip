// Which instruction mix does an MI355X sustain for the d = 4 accumulation, and at what clock?
// Sixteen wavefronts per CU (one 1024-thread block per CU, like the accumulate kernel), every wavefront
// loops over "segments"; per segment it handles NSETS sets of four frequencies.  Three mixes per set, all
// on pseudo-random operands read from LDS (the operands of the real kernel have no reuse either):
//   mix 0  the vector-only consumer of the round-4 kernel:    28 v_fma_f64 per set, 69/16 ds_read_b128
//   mix 1  first product + psi on the vector ALU (12 instructions), second product as FOUR
//          v_mfma_f64_4x4x4_4b (one frequency per block), 3 ds_read_b128
//   mix 2  the same with the three-product complex multiplication: 13 vector + THREE matrix instructions
// Every block stamps s_memtime / s_memrealtime around its loop, so issue cycles and clock can be told
// apart (tools/fp64_ceiling_probe.hip).  Printed: time per (segment, set) per SIMD in cycles against the
// sum of the instruction times, the clock held, and sets per microsecond for the whole chip.
//   hipcc --offload-arch=gfx950 -O2 tools/fp64_mix_probe.hip -o build/probe/fp64_mix
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));

struct Stamp {
    unsigned long long cycles, ticks;
};

__device__ inline double rnd(unsigned seed) {
    unsigned x = seed*2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    unsigned y = x*1664525u + 1013904223u;
    return ((x & 0xfffffu)*4294967296.0 + y)/(1048576.0*4294967296.0) - 0.5;
}

constexpr int kTileDoubles = 64*18;   // per segment buffer: q[64 frequencies][16] | psi[64][2]

template <int MIX, int NSETS>
__global__ __launch_bounds__(1024) void mix(double* out, Stamp* stamps, int segments) {
    __shared__ __attribute__((aligned(16))) double tile[2][kTileDoubles];
    for (int i = threadIdx.x; i < 2*kTileDoubles; i += 1024) (&tile[0][0])[i] = rnd(i + 77*blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m = lane >> 4, f = (lane >> 2) & 3;
    // frequency-independent operands of the segment (W_a[m][n][j] for this lane, T as the A operand)
    double wr[4], wi[4], tr, ti, td;
    for (int n = 0; n < 4; ++n) {
        wr[n] = rnd(lane*8 + n + 1000);
        wi[n] = rnd(lane*8 + n + 2000);
    }
    tr = rnd(lane + 3000);
    ti = rnd(lane + 4000);
    td = tr - ti;
    const double nti = -ti;
    double acc[NSETS][MIX == 0 ? 7 : 3];
    for (int s = 0; s < NSETS; ++s)
        for (int k = 0; k < (MIX == 0 ? 7 : 3); ++k) acc[s][k] = 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int g = 0; g < segments; ++g) {
        const double* t = tile[g & 1];
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const int fr = ((wave*NSETS + s)*4 + f) & 63;
            const double2_t q01 = *reinterpret_cast<const double2_t*>(t + fr*16 + m*4);
            const double2_t q23 = *reinterpret_cast<const double2_t*>(t + fr*16 + m*4 + 2);
            const double2_t psi = *reinterpret_cast<const double2_t*>(t + 64*16 + fr*2);
            if (MIX == 0) {
                // 28 multiply-adds on the same operands: 7 chains of 4
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    double a = acc[s][k];
                    a = fma(q01.x, wr[k & 3], a);
                    a = fma(q01.y, wi[k & 3], a);
                    a = fma(q23.x, k < 4 ? psi.x : tr, a);
                    a = fma(q23.y, k < 4 ? psi.y : ti, a);
                    acc[s][k] = a;
                }
            } else {
                double zr = q01.x*wr[0], zi = q01.x*wi[0];
                zr = fma(q01.y, wr[1], zr);
                zi = fma(q01.y, wi[1], zi);
                zr = fma(q23.x, wr[2], zr);
                zi = fma(q23.x, wi[2], zi);
                zr = fma(q23.y, wr[3], zr);
                zi = fma(q23.y, wi[3], zi);
                const double yr = fma(psi.x, zr, -psi.y*zi), yi = fma(psi.x, zi, psi.y*zr);
                if (MIX == 1) {
                    acc[s][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(tr, yr, acc[s][0], 0, 0, 0);
                    acc[s][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti, yi, acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(tr, yi, acc[s][1], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(nti, yr, acc[s][1], 0, 0, 0);
                } else {
                    const double ys = yr + yi;
                    acc[s][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(tr, yr, acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ti, yi, acc[s][1], 0, 0, 0);
                    acc[s][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(td, ys, acc[s][2], 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double sum = 0;
    for (int s = 0; s < NSETS; ++s)
        for (int k = 0; k < (MIX == 0 ? 7 : 3); ++k) sum += acc[s][k];
    out[blockIdx.x*1024 + threadIdx.x] = sum;
    if (threadIdx.x == 0) stamps[blockIdx.x] = {c1 - c0, r1 - r0};
}

template <typename K>
void run(const char* name, K kern, int nsets, int segments, double ideal_cycles_per_set) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    double* out;
    Stamp* stamps;
    (void)hipMalloc(&out, sizeof(double)*blocks*1024);
    (void)hipMalloc(&stamps, sizeof(Stamp)*blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 12; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, stamps, segments);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, out, stamps, segments);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<Stamp> h(blocks);
    (void)hipMemcpy(h.data(), stamps, sizeof(Stamp)*blocks, hipMemcpyDeviceToHost);
    std::vector<double> mhz(blocks), cyc(blocks);
    for (int i = 0; i < blocks; ++i) {
        mhz[i] = double(h[i].cycles)/double(h[i].ticks)*100.0;
        cyc[i] = double(h[i].cycles);
    }
    std::nth_element(mhz.begin(), mhz.begin() + blocks/2, mhz.end());
    std::nth_element(cyc.begin(), cyc.begin() + blocks/2, cyc.end());
    // four wavefronts per SIMD, nsets sets each per segment
    const double per_set = cyc[blocks/2]/(4.0*nsets*segments);
    const double sets_per_us = double(blocks)*16*nsets*segments/(best*1e3);
    printf("%-46s %8.3f ms  clock %5.0f MHz  %6.1f cycles per set and SIMD (instruction times add up to %5.1f)  %8.0f sets/us\n",
           name, best, mhz[blocks/2], per_set, ideal_cycles_per_set, sets_per_us);
    (void)hipFree(out);
    (void)hipFree(stamps);
}

int main() {
    const int seg = 4000;
    run("vector only, 4 sets per wavefront", mix<0, 4>, 4, seg, 112.0);
    run("12 vector + 4 matrix, 4 sets", mix<1, 4>, 4, seg, 112.0);
    run("12 vector + 4 matrix, 8 sets", mix<1, 8>, 8, seg/2, 112.0);
    run("13 vector + 3 matrix, 4 sets", mix<2, 4>, 4, seg, 100.0);
    run("13 vector + 3 matrix, 8 sets", mix<2, 8>, 8, seg/2, 100.0);
    return 0;
}
