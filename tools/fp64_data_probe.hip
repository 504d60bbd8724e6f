// Attainable FP64 rate as a function of the operand DATA (MI355X throttles under bit toggling):
// v_fma_f64 and v_mfma_f64_16x16x4 with (a) constant operands, (b) per-lane pseudo-random operands.
//   hipcc --offload-arch=gfx950 -O2 tools/fp64_data_probe.hip -o build/probe/fp64_data && ./build/probe/fp64_data
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;

__device__ inline double lane_value(unsigned seed) {
    unsigned x = seed*2654435761u + 12345u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    return (x & 0xffffff)/double(0x1000000) - 0.5;     // uniform in [-0.5, 0.5), full mantissa noise below
}

template <bool RANDOM>
__global__ __launch_bounds__(256) void valu(double* out, int iters) {
    double a[8], b[8], acc[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = RANDOM ? lane_value(threadIdx.x*16 + i) + 1e-9*lane_value(blockIdx.x + i) : 1.0000001;
        b[i] = RANDOM ? lane_value(threadIdx.x*16 + 8 + i) : 0.9999999;
        acc[i] = 0.0;
    }
    for (int it = 0; it < iters; ++it) {     // static operand choice: no extra instructions
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(a[i], b[(i + 1) & 7], acc[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(-a[(i + 3) & 7], b[i], acc[i]);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <bool RANDOM>
__global__ __launch_bounds__(256) void mfma(double* out, int iters) {
    f64x4 acc[8];
    double a[4], b[4];
    for (int i = 0; i < 8; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 4; ++i) {
        a[i] = RANDOM ? lane_value(threadIdx.x*8 + i) : 1.0000001;
        b[i] = RANDOM ? lane_value(threadIdx.x*8 + 4 + i) : 0.9999999;
    }
    double na[4];
    for (int i = 0; i < 4; ++i) na[i] = -a[i];
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(na[(i + 2) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int iters, double flops_per_thread_iter) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount*8;
    double* out; (void)hipMalloc(&out, sizeof(double)*blocks*256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 100);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.1f TFLOP/s\n", name, best, double(blocks)*256*iters*flops_per_thread_iter/best/1e9);
    (void)hipFree(out);
}

int main() {
    run("v_fma_f64, constant operands", valu<false>, 20000, 32.0);
    run("v_fma_f64, pseudo-random operands", valu<true>, 20000, 32.0);
    run("v_mfma_f64_16x16x4, constant operands", mfma<false>, 4000, 8*2048.0/64);
    run("v_mfma_f64_16x16x4, pseudo-random operands", mfma<true>, 4000, 8*2048.0/64);
    return 0;
}
