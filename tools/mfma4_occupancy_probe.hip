// v_mfma_f64_4x4x4_4b issue rate against the number of wavefronts per SIMD (random operands, 8
// independent accumulators per wavefront, and the same with only 2: dependent chains).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_occupancy_probe.hip -o build/probe/mfma4_occ
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ inline double lane_value(unsigned seed) {
    unsigned x = seed*2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    unsigned y = x*1664525u + 1013904223u;
    return ((x & 0xfffffu)*4294967296.0 + y)/(1048576.0*4294967296.0) - 0.5;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4(double* out, int iters) {
    double acc[NACC], a[4], b[4];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    for (int i = 0; i < 4; ++i) {
        a[i] = lane_value(threadIdx.x*8 + i + blockIdx.x*4096);
        b[i] = lane_value(threadIdx.x*8 + 4 + i + blockIdx.x*4096);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8/NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + r) & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount*waves_per_simd;     // 256 threads = one wave per SIMD
    const int iters = 40000/waves_per_simd;
    double* out;
    (void)hipMalloc(&out, sizeof(double)*blocks*256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = double(blocks)*4*iters*8*512.0;          // 4 waves per block, 8 MFMAs per iteration
    printf("%-34s %d wave(s)/SIMD  %8.3f ms  %6.1f TFLOP/s\n", name, waves_per_simd, best, flops/best/1e9);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 4, 8}) run("8 independent accumulators", mfma4<8>, w);
    for (int w : {1, 2, 3, 4, 8}) run("4 independent accumulators", mfma4<4>, w);
    for (int w : {1, 2, 3, 4, 8}) run("2 independent accumulators", mfma4<2>, w);
    return 0;
}
