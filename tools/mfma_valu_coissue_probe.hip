// Do v_mfma_f64_4x4x4_4b and ordinary vector instructions of ANOTHER wavefront on the same SIMD
// overlap?  Blocks of 512 threads = 2 wavefronts per SIMD; wavefronts 0-3 (one per SIMD) run a
// stream of independent MFMAs, wavefronts 4-7 a stream of (a) nothing, (b) v_add_u32 (integer),
// (c) v_permlane16_swap, (d) v_fma_f64, (e) ds_read_b128.  Reported: time of the MFMA-only run, of
// the partner-only run, and of both together -- "max" means they overlap, "sum" that they do not.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_coissue_probe.hip -o build/probe/coissue
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void probe(double* out, int mfma_iters, int partner_iters) {
    __shared__ double4 lds[1024];
    const int wave = threadIdx.x >> 6;
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    const double a = 1.0 + threadIdx.x*1e-3, b = 0.5 - threadIdx.x*1e-4;
    lds[threadIdx.x] = make_double4(a, b, a, b);
    lds[threadIdx.x + 512] = make_double4(b, a, b, a);
    __syncthreads();
    if (wave < 4) {
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else {
        unsigned u0 = threadIdx.x, u1 = threadIdx.x*3, u2 = 7, u3 = 11;
        double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
        for (int it = 0; it < partner_iters; ++it) {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(u0) : "v"(u1));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(u2) : "v"(u3));
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u0), "+v"(u1));
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u2), "+v"(u3));
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f0 = fma(f0, a, b);
                    f1 = fma(f1, b, a);
                    f2 = fma(f2, a, a);
                    f3 = fma(f3, b, b);
                }
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double4 v = lds[(threadIdx.x + 16*i + it) & 1023];
                    f0 += v.x;
                    f1 += v.w;
                }
            }
        }
        acc[0] = u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <typename K>
float time_it(K kern, int blocks, double* out, int mi, int pi) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, mi, pi);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, mi, pi);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <int MODE>
void run(const char* name, int partner_iters) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;      // one 8-wave block per CU
    double* out;
    (void)hipMalloc(&out, sizeof(double)*blocks*512);
    const int mi = 20000;
    const float t_m = time_it(probe<MODE>, blocks, out, mi, 0);
    const float t_p = time_it(probe<MODE>, blocks, out, 0, partner_iters);
    const float t_b = time_it(probe<MODE>, blocks, out, mi, partner_iters);
    printf("%-28s MFMA alone %7.3f ms   partner alone %7.3f ms   together %7.3f ms   (max %.3f, sum %.3f)\n", name,
           t_m, t_p, t_b, t_m > t_p ? t_m : t_p, t_m + t_p);
    (void)hipFree(out);
}

int main() {
    run<1>("partner: v_add_u32", 20000);
    run<2>("partner: v_permlane*_swap", 20000);
    run<3>("partner: v_fma_f64", 20000);
    run<4>("partner: ds_read_b128 + add", 20000);
    return 0;
}
