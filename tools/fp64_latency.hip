// fp64_latency.hip -- issue interval vs dependent latency of v_fma_f64 on gfx950:
// cycles per instruction seen by ONE wave running NCHAIN independent accumulation chains,
// with 1, 2, 3, 4 and 8 such waves resident per SIMD.  (DESIGN.md "Why the accumulate kernel
// wants >= 4 waves per SIMD".)
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_latency.hip -o build/fp64_latency
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 8192;

template <int NCHAIN>
__global__ void chain_kernel(double* out, long long* cycles, double a, double b) {
    double acc[NCHAIN];
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) acc[i] = threadIdx.x*1e-9 + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NCHAIN; ++i) acc[i] = fma(acc[i], a, b);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NCHAIN>
void run(double* out, long long* cyc, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        // blocks of 256 threads (one wave per SIMD each); w blocks per CU -> w waves per SIMD
        const dim3 grid(cus*waves_per_simd), block(256);
        hipLaunchKernelGGL(chain_kernel<NCHAIN>, grid, block, 0, 0, out, cyc, 0.999999, 1e-9);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel<NCHAIN>, grid, block, 0, 0, out, cyc, 0.999999, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long c = 0;
        hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
        const double n_inst = double(kIters)*8*NCHAIN;          // per wave
        const double ns_per_inst_wave = ms*1e6/n_inst;
        std::printf("chains=%2d waves/SIMD=%d : kernel %7.3f ms, %6.2f ns per v_fma_f64 per wave, SIMD issues one per %5.2f ns "
                    "(= %5.2f cycles at 2.4 GHz); s_memtime delta/inst %6.2f\n",
                    NCHAIN, waves_per_simd, ms, ns_per_inst_wave, ns_per_inst_wave/waves_per_simd,
                    ns_per_inst_wave/waves_per_simd*2.4, double(c)/n_inst);
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    double* out;
    long long* cyc;
    hipMalloc(&out, sizeof(double)*prop.multiProcessorCount*2*1024);
    hipMalloc(&cyc, sizeof(long long));
    run<1>(out, cyc, prop.multiProcessorCount);
    run<2>(out, cyc, prop.multiProcessorCount);
    run<4>(out, cyc, prop.multiProcessorCount);
    run<8>(out, cyc, prop.multiProcessorCount);
    return 0;
}
