#!/bin/bash
# Same-box A/B of the accumulate kernel alone over several builds of the library:
#   tools/ab_accumulate.sh <rounds> "<tune_accumulate.py arguments>" <library> [<library> ...]
# ("-" = the product library).  Prints tune_accumulate.py's line per build and round.
rounds=$1; args=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset FFK_LIBRARY; else export FFK_LIBRARY=$PWD/$lib; fi
    printf "%-32s " "$lib"
    python3 tools/tune_accumulate.py $args --chunks 0 2>&1 | tail -1
  done
done
