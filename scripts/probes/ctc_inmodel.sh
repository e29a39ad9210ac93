#!/bin/bash
# kernel trace of the benchmark (optionally another workload), CTC rows only
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-aishell_b32_t1000}
OUT=$R/gpurun_out/ctc_inmodel_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1
python3 $R/scripts/prof_summary.py $(ls $OUT/trace/*.db | head -1) 29 > $OUT/summary.txt
grep -E "ctc_" $OUT/summary.txt | cut -c1-50,105-175
