#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    out[128 + threadIdx.x] = r2[0];
    out[192 + threadIdx.x] = r2[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256*4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int s = 0; s < 4; ++s) { for (int i = 0; i < 64; ++i) printf("%u ", h[s*64+i]); printf("\n"); }
    return 0;
}
