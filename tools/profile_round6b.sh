#!/bin/bash
# Round-6 evidence for the two kernels added late in the round (one GPU call):
#   r06_m: decay amplitudes of config 5 -- the symmetric-block kernel against the register-fed one, its in-kernel clock
#          and block trace (build/libffk_dgclock.so = make VARIANT=dgclock VSRCS=decay.hip VFLAGS=-DFFK_DG_CLOCK),
#          kernel split of the config-5 pass, the exp tail in HBM against through the host
#   r06_l: liouville_representation d = 16 / batch 512 -- fused sparse contraction against the GEMM form, dense basis
#   gpurun -- 'bash tools/profile_round6b.sh > gpurun_out/prof_r06b.log 2>&1'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06b
mkdir -p $O/p
cd $R
{
echo "== decay amplitudes, config-5 shape (18 operators, N = 256, 16384 omega): tools/bench_etm.py"
echo "-- symmetric-block kernel (default)"; python3 tools/bench_etm.py 2>&1 | grep "decay\|operator"
echo "-- register-fed kernel (FFK_DECAY_REGISTER_FED=1: round 5's)"; FFK_DECAY_REGISTER_FED=1 python3 tools/bench_etm.py 2>&1 | grep "decay\|operator"
echo "-- ragged frequency count, 5 operators"; python3 tools/bench_etm.py --W 16389 --A 5 2>&1 | grep "decay\|operator"
echo "-- below a chip's worth of chunks (3 operators x 1030 omega): the register-fed kernel serves"; python3 tools/bench_etm.py --W 1030 --A 3 2>&1 | grep "decay\|operator"
echo "-- in-kernel clock and block trace (-DFFK_DG_CLOCK build)"; FFK_LIBRARY=build/libffk_dgclock.so python3 tools/bench_etm.py 2>&1 | grep -v amdgpu.ids | sed -n 1,28p
} > $O/r06_m_decay_symmetric_block.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p/c5 -- python3 $R/tools/time_config5.py --reps 20 > $O/p/c5.txt 2>&1
{ echo "== config-5 pass, kernel by kernel (rocprofv3 --kernel-trace --stats -- python3 tools/time_config5.py)"; python3 $R/tools/kstats.py $O/p/c5 | cut -c1-160 | head -24; grep median $O/p/c5.txt; } >> $O/r06_m_decay_symmetric_block.txt 2>&1
{
echo "== superoperator.liouville_representation, d = 16, Pauli basis, batch 512: tools/time_liouville.py --d 16"
echo "-- fused (default)"; python3 $R/tools/time_liouville.py --d 16 2>&1 | tail -1
echo "-- conjugation + GEMM (FFK_LIOUVILLE_GEMM=1: round 5's)"; FFK_LIOUVILLE_GEMM=1 python3 $R/tools/time_liouville.py --d 16 2>&1 | tail -1
echo "-- a basis without zeros (--dense-basis): the device-side switch picks the GEMM form"; python3 $R/tools/time_liouville.py --d 16 --dense-basis 2>&1 | tail -1
} > $O/r06_l_liouville_fused.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p/lv -- python3 $R/tools/time_liouville.py --d 16 > $O/p/lv.txt 2>&1
{ echo "== kernel split, fused"; python3 $R/tools/kstats.py $O/p/lv | cut -c1-160 | head -8; } >> $O/r06_l_liouville_fused.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p/lvd -- python3 $R/tools/time_liouville.py --d 16 --dense-basis > $O/p/lvd.txt 2>&1
{ echo "== kernel split, basis without zeros"; python3 $R/tools/kstats.py $O/p/lvd | cut -c1-160 | head -8; } >> $O/r06_l_liouville_fused.txt 2>&1
rm -rf $O/p
cat $O/r06_l_liouville_fused.txt
