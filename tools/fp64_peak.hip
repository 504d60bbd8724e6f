// fp64_peak.hip -- microbenchmark: sustained FP64 rate of (a) the vector pipe (v_fma_f64),
// (b) the matrix pipe (v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64) and (c) both issued
// from co-resident waves.  Evidence for the `peak` used in bench.py's roofline (the MI355X guide
// lists no FP64 row); results are recorded in DESIGN.md and profiles/.
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_peak.hip -o build/fp64_peak && build/fp64_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using f64x4 = __attribute__((ext_vector_type(4))) double;

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e));              \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int kIters = 4096;

__global__ __launch_bounds__(256) void valu_kernel(double* out, double a, double b) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x*1e-9 + i;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

// three VGPR sources per FMA (acc += x*y with per-lane x, y): the operand pattern of a
// contraction whose operands were broadcast from LDS into vector registers
__global__ __launch_bounds__(256) void valu3_kernel(double* out, double a, double b) {
    double acc[8], x[4], y[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x*1e-9 + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = out[(threadIdx.x + 64*i) & 1023]*1e-30 + a;   // per-lane values -> VGPRs
        y[i] = out[(threadIdx.x + 64*i + 7) & 1023]*1e-30 + b;
    }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(x[i & 3], y[(i + (i >> 2)) & 3], acc[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(x[(i + 1) & 3], y[i & 3], acc[i]);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

// two VGPR sources + one SGPR source (acc += s*y)
__global__ __launch_bounds__(256) void valu2s_kernel(double* out, double a, double b) {
    double acc[8], y[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x*1e-9 + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = out[(threadIdx.x + 64*i + 7) & 1023]*1e-30 + b;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(a, y[i & 3], acc[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(b, y[(i + 1) & 3], acc[i]);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma16_kernel(double* out, double a, double b) {
    f64x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    const double av = a + threadIdx.x*1e-9, bv = b;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma4_kernel(double* out, double a, double b) {
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    const double av = a + threadIdx.x*1e-9, bv = b;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

// waves 0,1 of each 4-wave block run MFMA, waves 2,3 run VALU FMA (same SIMDs host both kinds)
__global__ __launch_bounds__(512) void mixed_kernel(double* out, double a, double b) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave & 1) {
        f64x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
        const double av = a + threadIdx.x*1e-9, bv = b;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x*1e-9 + i;
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <typename K>
int time_kernel(const char* name, K kern, int block, double flops_per_thread_iter, double* out) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount*8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, 0.999999, 1e-9);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, out, 0.999999, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = flops_per_thread_iter*double(kIters)*double(grid)*block*reps;
    std::printf("%-28s %8.3f ms  %8.2f TFLOP/s\n", name, ms/reps, flops/(ms*1e-3)/1e12);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    std::printf("device: %s (%s), %d CUs, clock %d MHz\n", prop.name, prop.gcnArchName,
                prop.multiProcessorCount, prop.clockRate/1000);
    double* out;
    CHECK(hipMalloc(&out, sizeof(double)*prop.multiProcessorCount*8*512));
    // per thread and iteration: 16 FMA = 32 flop
    if (time_kernel("v_fma_f64 (VALU)", valu_kernel, 256, 32.0, out)) return 1;
    if (time_kernel("v_fma_f64 3 VGPR sources", valu3_kernel, 256, 32.0, out)) return 1;
    if (time_kernel("v_fma_f64 2 VGPR + 1 SGPR", valu2s_kernel, 256, 32.0, out)) return 1;
    // per wave and iteration: 4 MFMA x 16*16*4*2 flop = 8192 flop -> /64 lanes
    if (time_kernel("v_mfma_f64_16x16x4", mfma16_kernel, 256, 4*2048.0/64, out)) return 1;
    // 4x4x4 with 4 blocks: 4*4*4*4*2 = 512 flop per instruction
    if (time_kernel("v_mfma_f64_4x4x4_4b", mfma4_kernel, 256, 8*512.0/64, out)) return 1;
    // mixed: half the waves 128 flop/lane/iter (mfma), half 32 (valu)
    if (time_kernel("mixed MFMA16 + VALU", mixed_kernel, 512, (4*2048.0/64 + 32.0)/2, out)) return 1;
    CHECK(hipFree(out));
    return 0;
}
