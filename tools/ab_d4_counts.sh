#!/bin/bash
# d = 4 accumulate kernel alone for every operator count (config 2's segments and frequencies), over several builds:
#   tools/ab_d4_counts.sh "<A values>" <library> [<library> ...]      ("-" = the product library)
for A in $1; do
  for lib in "${@:2}"; do
    if [ "$lib" = "-" ]; then unset FFK_LIBRARY; else export FFK_LIBRARY=$PWD/$lib; fi
    printf "A=%-2s %-28s " "$A" "$lib"
    python3 tools/tune_accumulate.py --d 4 --G 256 --A $A --W 4096 --reps 30 --chunks 0 2>&1 | tail -1
  done
done
