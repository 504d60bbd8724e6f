#!/bin/bash
# A/B of the headline step under bench.py's own schedule (pre-warmed clocks, passes back to back):
#   tools/ab_bench.sh <rounds> "<env A>" "<env B>" [...]        e.g.  tools/ab_bench.sh 2 "FFK_TUNE_PC_SYNC=0" "FFK_TUNE_PC_SYNC=1"
# prints ms_per_step, the accumulate kernel's average launch time and the one-pass latency per run.
rounds=$1; shift
for r in $(seq $rounds); do
  for e in "$@"; do
    line=$(env $e python3 bench.py --no-configs --no-pmc --no-cpu-baseline 2>/dev/null | tail -1)
    python3 - "$e" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:40s} ms_per_step {d['ms_per_step']*1e3:7.2f} us  kernel {d['roofline']['avg_launch_ms']*1e3:7.2f} us "
      f"[{d['roofline']['launch_ms_min_max'][0]*1e3:.1f}..{d['roofline']['launch_ms_min_max'][1]*1e3:.1f}]  "
      f"one pass {d['single_stream_ms_per_step']*1e3:7.2f} us  frac {d['roofline']['frac']:.3f}")
PY
  done
done
