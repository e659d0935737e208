#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench + rocprofv3 passes for the round's profile evidence.
#   tools/profile_round.sh <tag> [plies-per-step] [extra bench.py arguments ...]
#   e.g. config 2:  LIGHT=1 tools/profile_round.sh r05c2 256 --size 9 --games-per-gpu 4096 --desync 128
# LIGHT=1: no CPU baseline / extras in the bench line, no counter calibration, no per-entry-point table (a second launch
# shape of a round whose main pass has them).
# kernel-trace/stats and each PMC group are separate runs (never --pmc together with sys/hip traces).
# tools/summarize_profiles.py <tag> then condenses gpurun_out/<tag>/ into tracked files under profiles/.
TAG=${1:-r02}; F=${2:-256}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --plies-per-step $F $*"
if [ -n "$LIGHT" ]; then timeout 900 $BENCH --no-cpu-baseline --no-also > $O/bench.json 2> $O/bench.err
else timeout 900 $BENCH > $O/bench.json 2> $O/bench.err; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH --no-cpu-baseline --no-also > $O/kt.log 2>&1
SHORT="python $R/bench.py --steps 6 --warmup 0 --burn-in 2 --plies-per-step $F --no-cpu-baseline --no-also $*"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $SHORT > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $SHORT > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_inst -o p -- $SHORT > $O/pmc_inst.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_act -o p -- $SHORT > $O/pmc_act.log 2>&1
timeout 600 rocprofv3 --pmc VALUBusy SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 --kernel-trace --output-format csv -d $O/pmc_busy -o p -- $SHORT > $O/pmc_busy.log 2>&1
if [ -n "$LIGHT" ]; then find $O -name '*_agent_info.csv' -delete 2>/dev/null; cat $O/bench.json; exit 0; fi
# calibration of FETCH_SIZE / WRITE_SIZE on launches whose byte counts are known (the guide: WRITE_SIZE is uncalibrated)
CAL="python $R/tools/calib_traffic.py"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -o p -- $CAL > $O/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -o p -- $CAL > $O/cal_write.log 2>&1
# the per-ply / children / areas entry points: throughput table + kernel stats of the same script
timeout 300 python $R/tools/bench_ops.py > $O/ops.json 2> $O/ops.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ops -o kt -- python $R/tools/bench_ops.py > $O/kt_ops.log 2>&1
# drop the bulky raw traces that the summary does not read (gpurun_out is capped at 64 MiB)
find $O -name '*_agent_info.csv' -delete 2>/dev/null
cat $O/bench.json
