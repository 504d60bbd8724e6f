#!/bin/bash
# Compile one kernel source to gfx950 assembly and print resources plus an instruction histogram of
# the lines between two labels:   tools/isa_loop.sh ctrl_pcr.hip "<flags>" [.LBB0_4 .LBB0_6]
src=$1; flags=$2; l0=${3:-}; l1=${4:-}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/isa/$(basename $src .hip)$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_').s
mkdir -p $root/build/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -I$root/include \
    -I$root/filter_functions_amd/csrc -S --cuda-device-only $flags $root/filter_functions_amd/csrc/$src -o $out 2>&1 | grep -v "hip-link"
echo "== $src $flags -> $out"
grep "\.vgpr_count\|\.private_segment_fixed_size\|\.name: \|\.vgpr_spill_count\|\.sgpr_spill_count" $out | paste - - - - - | sed 's/  */ /g'
if [ -n "$l0" ]; then
  s=$(grep -n "^$l0:" $out | head -1 | cut -d: -f1); e=$(grep -n "^$l1:" $out | head -1 | cut -d: -f1)
  echo "lines $s..$e"
  awk -v s=$s -v e=$e 'NR>=s && NR<=e' $out | awk '{print $1}' | grep -v "^;\|^\." | sort | uniq -c | sort -rn | head -${5:-16}
fi
