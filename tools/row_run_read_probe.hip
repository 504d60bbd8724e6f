// Read bandwidth for the access pattern of the decay-amplitude GEMM: a block repeatedly fetches a
// step image of ROWS rows x RUN bytes, the rows 262144 bytes apart ((A, N, W) layout, W = 16384 complex),
// advancing RUN bytes per step; 18 "operators" of 256 rows x 256 KiB = 1.2 GB in all.  RUN = 256 (what
// the GEMM does: 16 frequencies per step), 512, 1024, 4096.
//   hipcc --offload-arch=gfx950 -O2 tools/row_run_read_probe.hip -o build/probe/row_run
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void reader(const double2* __restrict__ data, int rows_per_block, int run16,
                                              int steps, size_t row_stride16, double2* __restrict__ sink) {
    // block b: operator a = b / (256 / rows_per_block ... ) ; simply: strip = blockIdx.x, chunk = blockIdx.y
    const size_t strip0 = static_cast<size_t>(blockIdx.x)*rows_per_block;
    const size_t w0 = static_cast<size_t>(blockIdx.y)*steps*run16;
    double2 acc = {0.0, 0.0};
    const int per_row = run16;                       // 16-byte elements per row and step
    for (int s = 0; s < steps; ++s) {
        for (int e = threadIdx.x; e < rows_per_block*per_row; e += 256) {
            const int r = e / per_row, c = e % per_row;
            const double2 v = data[(strip0 + r)*row_stride16 + w0 + static_cast<size_t>(s)*run16 + c];
            acc.x += v.x;
            acc.y += v.y;
        }
    }
    if (acc.x == 12345.678) sink[0] = acc;
}

int main() {
    const size_t W = 16384, rows = 18*256;
    const size_t n16 = rows*W;
    double2 *data, *sink;
    (void)hipMalloc(&data, n16*16);
    (void)hipMalloc(&sink, 16);
    (void)hipMemset(data, 0, n16*16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int run_bytes : {256, 512, 1024, 4096}) {
        for (int rows_per_block : {128, 32}) {
            const int run16 = run_bytes/16;
            const int chunk = 2048;                      // frequencies per block (k-chunk)
            const int steps = chunk/run16;
            const dim3 grid(static_cast<unsigned>(rows/rows_per_block), static_cast<unsigned>(W/chunk));
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(reader, grid, dim3(256), 0, 0, data, rows_per_block, run16, steps, W, sink);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("run %5d B, %3d rows per block: %7.3f ms  %6.2f TB/s\n", run_bytes, rows_per_block, best,
                   n16*16.0/best/1e9);
        }
    }
    return 0;
}
