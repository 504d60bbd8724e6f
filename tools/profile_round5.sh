#!/bin/bash
# Round-5 evidence in one GPU call: rocprofv3 --kernel-trace --stats of the driver's bench command (two passes
# in flight) and of the same with --streams 1 (isolated kernel durations), the PMC passes over the d = 4
# accumulate kernel, and the 2- and 4-rank rehearsals of `bench.py --gpus N` on one GPU (gloo control plane).
#   gpurun -- 'bash tools/profile_round5.sh > gpurun_out/prof_r05.log 2>&1'
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05/prof
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof/two_stream -- python3 $R/bench.py --steps 20 --warmup 5 --no-configs --no-pmc --no-cpu-baseline > $R/gpurun_out/r05/prof/bench_two_stream.json 2> $R/gpurun_out/r05/prof/two_stream.err
python3 $R/tools/kstats.py $R/gpurun_out/r05/prof/two_stream > $R/gpurun_out/r05/kernel_stats_two_stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof/one_stream -- python3 $R/bench.py --steps 200 --warmup 20 --streams 1 --no-configs --no-pmc --no-cpu-baseline > $R/gpurun_out/r05/prof/bench_one_stream.json 2> $R/gpurun_out/r05/prof/one_stream.err
python3 $R/tools/kstats.py $R/gpurun_out/r05/prof/one_stream > $R/gpurun_out/r05/kernel_stats_one_stream.txt 2>&1
cd $R
bash tools/pmc_accumulate.sh r05 4 256 3 4096 > gpurun_out/r05/pmc_d4_accumulate.txt 2>&1
rm -rf gpurun_out/pmc gpurun_out/r05/prof/two_stream gpurun_out/r05/prof/one_stream
FFK_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 --warmup 10 --no-configs --no-pmc --no-cpu-baseline > gpurun_out/r05/rehearsal_2_ranks_one_gpu.json 2> gpurun_out/r05/rehearsal_2.err
FFK_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 50 --warmup 10 --no-configs --no-pmc --no-cpu-baseline > gpurun_out/r05/rehearsal_4_ranks_one_gpu.json 2> gpurun_out/r05/rehearsal_4.err
tail -c 300 gpurun_out/r05/rehearsal_4.err
head -8 gpurun_out/r05/kernel_stats_two_stream.txt; head -8 gpurun_out/r05/kernel_stats_one_stream.txt; tail -30 gpurun_out/r05/pmc_d4_accumulate.txt
