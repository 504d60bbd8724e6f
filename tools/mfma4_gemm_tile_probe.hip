// Does v_mfma_f64_4x4x4_4b keep its rate with a GEMM-shaped register tile?  One wavefront per SIMD,
// operands resident in registers (no memory in the loop):
//   (a) 8 accumulators, 4 + 4 operand registers (tools/mfma4_occupancy_probe.hip);
//   (b) 64 accumulators acc[16][4], operands a[16], b[4], each accumulator used twice per pass with a
//       distance of four instructions (the inner loop of decay_gemm_lds_kernel<true>).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4_gemm_tile_probe.hip -o build/probe/mfma4_tile
#include <hip/hip_runtime.h>

#include <cstdio>

__device__ inline double lane_value(unsigned seed) {
    unsigned x = seed*2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    return (x & 0xffffff)/double(0x1000000) - 0.5;
}

__global__ __launch_bounds__(256) void small_tile(double* out, int iters) {
    double acc[8], a[4], b[4];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    for (int i = 0; i < 4; ++i) { a[i] = lane_value(threadIdx.x*8 + i); b[i] = lane_value(threadIdx.x*8 + 4 + i); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + r) & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void gemm_tile(double* out, int iters) {
    double acc[16][4], ar[16], ai[16], br[4], bi[4];
    for (int r = 0; r < 16; ++r) {
        ar[r] = lane_value(threadIdx.x*64 + r);
        ai[r] = lane_value(threadIdx.x*64 + 16 + r);
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    }
    for (int c = 0; c < 4; ++c) { br[c] = lane_value(threadIdx.x*64 + 32 + c); bi[c] = lane_value(threadIdx.x*64 + 40 + c); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
                acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[rg], br[cg], acc[rg][cg], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
                acc[rg][cg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai[rg], bi[cg], acc[rg][cg], 0, 0, 0);
        }
    }
    double s = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 4; ++c) s += acc[r][c];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int iters, double mfma_per_iter) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    double* out;
    (void)hipMalloc(&out, sizeof(double)*blocks*256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
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
    printf("%-46s %8.3f ms  %6.1f TFLOP/s  (%.1f ns per instruction and SIMD)\n", name, best,
           double(blocks)*4*iters*mfma_per_iter*512.0/best/1e9, best*1e6/(iters*mfma_per_iter));
    (void)hipFree(out);
}

int main() {
    run("8 accumulators, 4 + 4 operands", small_tile, 4000, 128.0);
    run("64 accumulators acc[16][4], a[16], b[4] (re, im)", gemm_tile, 4000, 128.0);
    return 0;
}
