# Kernel sequences of one plain and one forced data-parallel step of the benchmark (gpurun_out/seq/).
set -e
R=${GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/seq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/plain -o bench -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-logit-error > $OUT/plain.log 2>&1
export ATHENA_FORCE_DP=1
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/dp -o bench -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-logit-error > $OUT/dp.log 2>&1
cd $R
python3 scripts/prof_sequence.py $OUT/plain/bench_results.db step_increment ${WHICH:-8} > $OUT/plain_seq.txt
python3 scripts/prof_sequence.py $OUT/dp/bench_results.db step_increment ${WHICH:-8} > $OUT/dp_seq.txt
tail -1 $OUT/plain.log | cut -c1-300; tail -1 $OUT/dp.log | cut -c1-300
rm -rf $OUT/plain $OUT/dp
