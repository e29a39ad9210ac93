#!/bin/bash
# Profiles of the benchmark on the GPU box: kernel trace (rocprofv3 results database -> per-kernel
# summary), the two HBM counter passes (one counter per pass, csv -> traffic per launch and family),
# the benchmark lines of the headline workload and of the two CTC-stress workloads.  Results land
# under gpurun_out/prof_bench/ with the names they are committed under in profiles/.
# usage: scripts/run_profiles.sh [prefix, default r3]
set -e
P=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=24; WARM=5
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-logit-error > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logit-error > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logit-error > $OUT/write.log 2>&1
cd $R
# the number of profiled steps = launches of the optimizer kernel
NSTEP=$(python3 scripts/prof_query.py $OUT/trace/bench_results.db clip_adam | sed -n 's/.* n= *\([0-9]*\) .*/\1/p' | head -1)
python3 scripts/prof_summary.py $OUT/trace/bench_results.db ${NSTEP:-1} > $OUT/${P}_bench_kernel_stats.txt
python3 scripts/prof_sequence.py $OUT/trace/bench_results.db step_increment 8 > $OUT/${P}_step_sequence.txt
python3 scripts/pmc_traffic.py $(ls $OUT/fetch/*counter_collection.csv | head -1) $(ls $OUT/write/*counter_collection.csv | head -1) $OUT/${P}_pmc_traffic.json > /dev/null
timeout -k 10 280 python3 bench.py > $OUT/${P}_bench.json 2> $OUT/bench.err
timeout -k 10 280 python3 bench.py --workload librispeech_b8_t3000 --no-cpu-baseline > $OUT/${P}_bench_librispeech_b8_t3000.json 2> $OUT/bench_libri.err
timeout -k 10 280 python3 bench.py --workload ctc_heavy_b64 --no-cpu-baseline > $OUT/${P}_bench_ctc_heavy_b64.json 2> $OUT/bench_ctc.err
rm -rf $OUT/trace $OUT/fetch $OUT/write
ls -la $OUT
head -3 $OUT/${P}_bench_kernel_stats.txt
