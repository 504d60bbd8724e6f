// What does a launch COST by its shape?  Empty kernels (every wavefront returns after one barrier), launched
// back to back on one stream, timed with HIP events over 500 launches and singly (event, launch, event):
// the d = 4 accumulate kernel's shape (256 blocks x 768 threads, 101 KB of dynamic LDS) against smaller ones.
//   hipcc --offload-arch=gfx950 -O2 tools/launch_shape_probe.hip -o build/probe/launch_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(int* out) {
    extern __shared__ int lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    __syncthreads();
    if (out != nullptr && lds[0] == 2) out[0] = 1;
}

int main() {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Shape { int blocks, threads, lds; } shapes[] = {
        {256, 64, 0}, {256, 256, 0}, {256, 768, 0}, {256, 1024, 0}, {256, 768, 32*1024}, {256, 768, 64*1024},
        {256, 768, 101*1024}, {256, 768, 160*1024}, {256, 256, 101*1024}, {64, 768, 101*1024}, {512, 768, 101*1024},
        {1024, 768, 101*1024}, {2048, 64, 0}, {16384, 64, 0}};
    printf("%8s %8s %8s | %12s %12s\n", "blocks", "threads", "LDS KB", "us/launch", "us single");
    for (const Shape& sh : shapes) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.blocks), dim3(sh.threads), sh.lds, s, nullptr);
        CHECK(hipStreamSynchronize(s));
        const int n = 500;
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(sh.blocks), dim3(sh.threads), sh.lds, s, nullptr);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        float single = 0.f;
        for (int i = 0; i < 20; ++i) {
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(empty_kernel, dim3(sh.blocks), dim3(sh.threads), sh.lds, s, nullptr);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float m1 = 0.f;
            CHECK(hipEventElapsedTime(&m1, e0, e1));
            single += m1;
        }
        printf("%8d %8d %8d | %12.2f %12.2f\n", sh.blocks, sh.threads, sh.lds/1024, ms*1000.f/n, single*1000.f/20);
    }
    return 0;
}
