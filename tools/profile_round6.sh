#!/bin/bash
# Round-6 evidence in one GPU call: rocprofv3 --kernel-trace --stats of the driver's bench command (two passes in
# flight) and of the same with --streams 1 (isolated kernel durations), the PMC passes over the d = 4, d = 8 and
# d = 16 accumulate kernels, and the 2- and 4-rank rehearsals of `bench.py --gpus N` on one GPU.
#   gpurun -- 'bash tools/profile_round6.sh > gpurun_out/prof_r06.log 2>&1'
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06/prof
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/prof/two_stream -- python3 $R/bench.py --steps 20 --warmup 5 --no-configs --no-pmc --no-cpu-baseline > $R/gpurun_out/r06/bench_line_under_rocprofv3.json 2> $R/gpurun_out/r06/prof/two_stream.err
python3 $R/tools/kstats.py $R/gpurun_out/r06/prof/two_stream > $R/gpurun_out/r06/kernel_stats_two_stream.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/prof/one_stream -- python3 $R/bench.py --steps 200 --warmup 20 --streams 1 --no-configs --no-pmc --no-cpu-baseline > $R/gpurun_out/r06/prof/bench_one_stream.json 2> $R/gpurun_out/r06/prof/one_stream.err
python3 $R/tools/kstats.py $R/gpurun_out/r06/prof/one_stream > $R/gpurun_out/r06/kernel_stats_one_stream.txt 2>&1
# config 4 shard and config 5 passes: per-kernel split
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/prof/cfg4 -- python3 $R/tools/tune_accumulate.py --d 8 --G 512 --A 9 --W 8192 --reps 20 --chunks 0 > $R/gpurun_out/r06/prof/cfg4.txt 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/r06/prof/cfg4 > $R/gpurun_out/r06/kernel_stats_config4_shard.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/prof/cfg5 -- python3 $R/tools/time_config5.py --reps 20 > $R/gpurun_out/r06/prof/cfg5.txt 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/r06/prof/cfg5 > $R/gpurun_out/r06/kernel_stats_config5.txt 2>&1
grep "config 5" $R/gpurun_out/r06/prof/cfg5.txt >> $R/gpurun_out/r06/kernel_stats_config5.txt
rm -rf $R/gpurun_out/r06/prof/two_stream $R/gpurun_out/r06/prof/one_stream $R/gpurun_out/r06/prof/cfg4 $R/gpurun_out/r06/prof/cfg5
cd $R
bash tools/pmc_accumulate.sh r06 4 256 3 4096 > gpurun_out/r06/pmc_d4_accumulate.txt 2>&1
bash tools/pmc_accumulate.sh r06 8 512 9 8192 > gpurun_out/r06/pmc_d8_accumulate.txt 2>&1
bash tools/pmc_accumulate.sh r06 16 13 18 16384 > gpurun_out/r06/pmc_d16_accumulate.txt 2>&1
rm -rf gpurun_out/pmc
FFK_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 50 --warmup 10 --no-configs --no-pmc --no-cpu-baseline > gpurun_out/r06/rehearsal_2_ranks_one_gpu.json 2> gpurun_out/r06/rehearsal_2.err
FFK_BENCH_REHEARSE=1 timeout -k 10 300 python3 bench.py --gpus 4 --steps 50 --warmup 10 --no-configs --no-pmc --no-cpu-baseline > gpurun_out/r06/rehearsal_4_ranks_one_gpu.json 2> gpurun_out/r06/rehearsal_4.err
tail -c 300 gpurun_out/r06/rehearsal_4.err
head -8 gpurun_out/r06/kernel_stats_two_stream.txt; head -8 gpurun_out/r06/kernel_stats_one_stream.txt; tail -30 gpurun_out/r06/pmc_d8_accumulate.txt
