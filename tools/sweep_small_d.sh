#!/bin/bash
# d = 2, 3 (and 5, 6): the symmetric kernel with and without the in-block segment split (tuning variants 0 / 2) on a few shapes
for d in 2 3 5 6; do for shape in "256 3 4096" "64 2 1024" "1000 3 512" "32 6 300"; do set -- $shape; for v in 0 2; do printf "d=%s G=%-4s A=%s W=%-4s variant=%s  " $d $1 $2 $3 $v; python tools/tune_accumulate.py --d $d --G $1 --A $2 --W $3 --variant $v --reps 20 --chunks 0 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*grid=/grid=/'; done; done; done
