#!/bin/bash
# Register / spill / scratch metadata of the accumulate-kernel instantiations of one dimension.
#   tools/kmeta.sh 4 [extra hipcc flags]      (writes build/asm/ctrl_d<D>.s)
D=$1; shift
mkdir -p build/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Iinclude \
  -Ifilter_functions_amd/csrc -DFFK_ONLY_D=$D "$@" --cuda-device-only -S \
  filter_functions_amd/csrc/ctrl.hip -o build/asm/ctrl_d$D.s 2>/dev/null
awk '/^\s*\.amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_private_segment_fixed_size/{p=$2}
     /\.end_amdhsa_kernel/{print k, "vgpr=" v, "scratch=" p}' build/asm/ctrl_d$D.s | sed 's/_ZN3ffk12_GLOBAL__N_1//' | grep accumulate_kernelILi
grep -E "vgpr_spill_count|\.name:" build/asm/ctrl_d$D.s | paste - - | grep accumulate_kernelILi | awk '{print $2, $4}' | sed 's/_ZN3ffk12_GLOBAL__N_1//'
