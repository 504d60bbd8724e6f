// ffk_mfma_util.h -- lane-row transposes for the v_mfma_f64_4x4x4_4b operand layouts (gfx950:
// v_permlane16_swap / v_permlane32_swap), shared by the matrix-core accumulate kernels.
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

}  // namespace ffk
