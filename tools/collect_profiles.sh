#!/bin/bash
# Measurement artefacts of one round, collected on the GPU box in ONE gpurun call (each profiler pass is its own run of the same bench command;
# counters never share a run with a trace domain other than --kernel-trace):  bash tools/collect_profiles.sh r03
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-batch16"
echo "[collect] kernel trace + stats"
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch16 > $O/bench_under_rocprof.json 2> $O/stats.stderr.txt || exit 1
echo "[collect] FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- $BENCH > /dev/null 2> $O/pmc_f.stderr.txt || exit 1
echo "[collect] WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- $BENCH > /dev/null 2> $O/pmc_w.stderr.txt || exit 1
echo "[collect] MFMA / VALU utilisation"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-include-regex "igemm_kernel|attn_kernel|astat_regs_kernel" \
    -d $O/pmc_m -o p --output-format csv -- $BENCH > /dev/null 2> $O/pmc_m.stderr.txt || exit 1
cd $R
python3 tools/pmc_bench_traffic.py $O/pmc_f $O/pmc_w $O/igemm_traffic.json > $O/traffic_summary.txt || exit 1
python3 tools/pmc_mfma_util.py $O/pmc_m $O/mfma_utilisation.json > /dev/null || exit 1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv
# the big raw traces stay on the box (gpurun_out merge limit): keep the summaries only
rm -rf $O/stats $O/pmc_f $O/pmc_w $O/pmc_m
cp $O/igemm_traffic.json $R/profiles/igemm_traffic.json            # so that the plain bench below reports the measured traffic
echo "[collect] plain bench (driver form)"
python3 bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err || exit 1
tail -c 400 $O/traffic_summary.txt
