#!/bin/bash
# Profiles of the benchmark on the GPU box: kernel trace (rocprofv3 results database) and the two HBM
# counter passes (one counter per pass, csv).  Results land under gpurun_out/prof_bench/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1
ls -la $OUT/*/
