// Operand layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4), derived empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb) {
        const double a = lane + 1, b = lane == lb ? 1.0 : 0.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[lb*64 + lane] = d;
    }
}
int main() {
    double* d; (void)hipMalloc(&d, 64*64*8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static double h[64*64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int lb = 0; lb < 64; ++lb) {
        printf("B one-hot at lane %2d -> D:", lb);
        for (int l = 0; l < 64; ++l) if (h[lb*64 + l] != 0) printf(" [%d]=A@%d", l, int(h[lb*64 + l]) - 1);
        printf("\n");
    }
    return 0;
}
