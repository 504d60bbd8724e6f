// corun.hip -- when does a small kernel run BESIDE a 16-wave, 140 KiB-LDS, FMA-dense block (the
// shape of the d = 4 accumulate kernel) instead of after it?
// Kernel A ("hog"): one 1024-thread block per CU, 143872 B of dynamic LDS, ~100 VGPRs, dependent
// FMA chains for `spin` ticks.  Kernel B: a few 64-thread blocks launched on another stream ~75 us
// later, with dynamic or static LDS, with or without a barrier per iteration, at wave priority
// BPRIO (default: not raised).  Reported: when B finished relative to A's start.
//   hipcc -O2 --offload-arch=gfx950 [-DBPRIO=3] [-DNACC=46] tools/corun.hip -o /tmp/corun && /tmp/corun
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

#ifndef NACC
#define NACC 46      /* doubles carried through A's loop: 46 -> 102 VGPRs */
#endif

__global__ __launch_bounds__(1024, 4) void hog(long long spin_ticks, double* sink) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = lds[(threadIdx.x + i) & 1023];
    while (wall_clock64() - t0 < spin_ticks) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = acc[i]*1.0000001 + acc[(i + 1) % NACC]*1e-9;
    }
    double total = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) total += acc[i];
    if (total == 12345.678) sink[0] = total;
}

__device__ __forceinline__ void raise_priority() {
#ifdef BPRIO
    __builtin_amdgcn_s_setprio(BPRIO);
#endif
}

__global__ __launch_bounds__(64) void small_dynamic(double* out) {
    extern __shared__ double lds[];
    raise_priority();
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double acc = 0;
    for (int i = 0; i < 200; ++i) acc += lds[(threadIdx.x + i) & 63];
    out[blockIdx.x*64 + threadIdx.x] = acc;
}

// NB doubles of static LDS; indices wrap at MASK + 1 (a power of two <= NB: cheap) or, MASK = 0, at NB
// itself (an integer division per index: ~40 instructions)
template <int NB, bool BARRIER, int MASK = 0>
__global__ __launch_bounds__(64) void small_static(double* out) {
    __shared__ double b[NB];
    raise_priority();
    for (int i = threadIdx.x; i < NB; i += 64) b[i] = i;
    __syncthreads();
    double acc = 0;
    for (int i = 0; i < 200; ++i) {
        if (MASK)
            acc += b[(threadIdx.x*3 + i) & MASK]*b[(i*7) & MASK];
        else
            acc += b[(threadIdx.x*3 + i) % NB]*b[(i*7) % NB];
        if (BARRIER) __syncthreads();
    }
    out[blockIdx.x*64 + threadIdx.x] = acc;
}

struct Rig {
    int cus;
    long long spin;
    double *sink, *out;
    hipStream_t sa, sb;
    hipEvent_t a0, a1, b0, b1;

    template <typename Launch>
    void run(const char* what, Launch launch_b) {
        float ta = 0, tb0 = 0, tb1 = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(a0, sa);
            hipLaunchKernelGGL(hog, dim3(cus), dim3(1024), 143872, sa, spin, sink);
            hipEventRecord(a1, sa);
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            hipEventRecord(b0, sb);
            launch_b();
            hipEventRecord(b1, sb);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ta, a0, a1);
            hipEventElapsedTime(&tb0, a0, b0);
            hipEventElapsedTime(&tb1, a0, b1);
        }
        printf("A: %4.0f us | B: %-58s launched at %4.0f us, done at %4.0f us -> %s\n", ta*1e3, what,
               tb0*1e3, tb1*1e3, tb1 < ta - 0.003f ? "ran beside A" : "WAITED for A");
    }
};

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    Rig r;
    r.cus = prop.multiProcessorCount;
    r.spin = 300*100;   // wall_clock64 ticks at 100 MHz: 300 us
    (void)hipMalloc(&r.sink, 8);
    (void)hipMalloc(&r.out, 8*64*4096);
    (void)hipStreamCreateWithFlags(&r.sa, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&r.sb, hipStreamNonBlocking);
    (void)hipEventCreate(&r.a0); (void)hipEventCreate(&r.a1);
    (void)hipEventCreate(&r.b0); (void)hipEventCreate(&r.b1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small_dynamic),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 64*1024);
#ifdef BPRIO
    printf("B raises its wave priority to %d\n", BPRIO);
#else
    printf("B at default wave priority\n");
#endif
    double* out = r.out;
    hipStream_t sb = r.sb;
    r.run("dynamic LDS 1024 B x 16 blocks", [&] { hipLaunchKernelGGL(small_dynamic, dim3(16), dim3(64), 1024, sb, out); });
    r.run("dynamic LDS 9728 B x 16 blocks", [&] { hipLaunchKernelGGL(small_dynamic, dim3(16), dim3(64), 9728, sb, out); });
    r.run("dynamic LDS 17920 B x 272 blocks", [&] { hipLaunchKernelGGL(small_dynamic, dim3(272), dim3(64), 17920, sb, out); });
    r.run("static LDS 4096 B x 16, no barrier in loop", [&] { hipLaunchKernelGGL((small_static<512, false>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 4096 B x 16, barrier per iteration", [&] { hipLaunchKernelGGL((small_static<512, true>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 8192 B x 16, no barrier in loop", [&] { hipLaunchKernelGGL((small_static<1024, false>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 8192 B x 16, barrier per iteration", [&] { hipLaunchKernelGGL((small_static<1024, true>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 9728 B x 16, no barrier in loop", [&] { hipLaunchKernelGGL((small_static<1216, false>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 9728 B x 16, barrier per iteration", [&] { hipLaunchKernelGGL((small_static<1216, true>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 9728 B x 16, barrier, cheap indices", [&] { hipLaunchKernelGGL((small_static<1216, true, 1023>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 17920 B x 16, barrier, cheap indices", [&] { hipLaunchKernelGGL((small_static<2240, true, 1023>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 8000 B x 16, barrier, division per index", [&] { hipLaunchKernelGGL((small_static<1000, true>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 17920 B x 16, no barrier in loop", [&] { hipLaunchKernelGGL((small_static<2240, false>), dim3(16), dim3(64), 0, sb, out); });
    r.run("static LDS 17920 B x 16, barrier per iteration", [&] { hipLaunchKernelGGL((small_static<2240, true>), dim3(16), dim3(64), 0, sb, out); });
    return 0;
}
