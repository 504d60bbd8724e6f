#!/bin/bash
# d = 2, 3: the one-wave kernel (tuning variant 1) against the block kernel (0 / 2) by chunk count; round 6
for d in 2 3; do for shape in "256 3 4096" "64 2 1024" "1000 3 512" "32 6 300"; do set -- $shape
  for v in 0 2; do printf "d=%s G=%-4s A=%s W=%-4s variant=%s chunks=auto " $d $1 $2 $3 $v; python tools/tune_accumulate.py --d $d --G $1 --A $2 --W $3 --variant $v --reps 20 --chunks 0 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*grid=/grid=/'; done
  for c in 0 8 16 32 64; do printf "d=%s G=%-4s A=%s W=%-4s variant=1 chunks=%-4s " $d $1 $2 $3 $c; python tools/tune_accumulate.py --d $d --G $1 --A $2 --W $3 --variant 1 --reps 20 --chunks $c 2>&1 | grep -v amdgpu | tail -1 | sed 's/.*grid=/grid=/'; done
done; done
