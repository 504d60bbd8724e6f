// How many wavefronts per SIMD does v_mfma_f64_16x16x4 need to saturate the matrix pipe?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_occupancy_probe.hip -o build/probe/mfma_occ && ./build/probe/mfma_occ
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, int block) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus*4*waves_per_simd*64/block;
    double* out; (void)hipMalloc(&out, sizeof(double)*blocks*block);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(block), 0, 0, out, 10, 1.0, 2.0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(block), 0, 0, out, iters, 1.0, 2.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = double(blocks)*(block/64)*iters*NACC;
    printf("acc=%2d waves/SIMD=%d block=%3d: %7.3f ms  %6.1f TFLOP/s  %6.1f cycles/MFMA/SIMD @2.4GHz\n", NACC,
           waves_per_simd, block, ms, mfmas*2048/ms/1e9, ms*1e-3*2.4e9*cus*4/mfmas);
    (void)hipFree(out);
}

int main() {
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256); run<16>(1, 256);
    run<8>(2, 256); run<8>(4, 256); run<2>(4, 256); run<1>(8, 256);
    return 0;
}
