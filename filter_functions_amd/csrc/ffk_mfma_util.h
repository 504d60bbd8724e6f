// ffk_mfma_util.h -- lane-row transposes for the v_mfma_f64_4x4x4_4b operand layouts (gfx950:
// v_permlane16_swap / v_permlane32_swap) and the LDS-DMA copy, shared by the matrix-core accumulate kernels.
// Layout of the instruction (tools/mfma4_layout_probe.hip), c = lane & 15, q = lane >> 4:
//   A[i = c & 3][k = q] (the same 4 x 4 matrix in each of the four blocks c >> 2),
//   B[k = q][column c],  D[i = q][column c]:  four rows, sixteen columns per instruction.
#pragma once

#include <hip/hip_runtime.h>

namespace ffk {

// 2 x 2 transposes between two registers and the 16-lane rows (bit 0 / bit 1 of the row index)
__device__ __forceinline__ void swap_rows16(double& a, double& b) {
    unsigned alo = static_cast<unsigned>(__double2loint(a)), ahi = static_cast<unsigned>(__double2hiint(a));
    unsigned blo = static_cast<unsigned>(__double2loint(b)), bhi = static_cast<unsigned>(__double2hiint(b));
    auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
    b = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}
__device__ __forceinline__ void swap_rows32(double& a, double& b) {
    unsigned alo = static_cast<unsigned>(__double2loint(a)), ahi = static_cast<unsigned>(__double2hiint(a));
    unsigned blo = static_cast<unsigned>(__double2loint(b)), bhi = static_cast<unsigned>(__double2hiint(b));
    auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
    b = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}
// v[k] in row q  <-  v[q] in row k   (rows = lane >> 4)
__device__ __forceinline__ void transpose_rows(double (&v)[4]) {
    swap_rows16(v[0], v[1]);
    swap_rows16(v[2], v[3]);
    swap_rows32(v[0], v[2]);
    swap_rows32(v[1], v[3]);
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no register, no ds_write -- a
// ds_write_b128 costs a wavefront 35-45 cycles of issue, tools/lds_issue_probe.py): the destination is wave-uniform
// base + 16 lane, the source per lane (inactive lanes copy nothing).  As an asm statement: through the builtin hipcc,
// which cannot tell what the copy writes, waits vmcnt(0) in front of the next LDS read.  The CALLER waits
// (s_waitcnt vmcnt(0)) before the barrier that publishes the copies.
__device__ __forceinline__ void lds_dma16(const void* lane_source, void* uniform_destination) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(uniform_destination)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_source), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace ffk
