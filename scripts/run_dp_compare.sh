# Plain step against the forced single-rank data-parallel step (gpurun_out/dpcmp/).
set -e
R=${GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/dpcmp
mkdir -p $OUT
cd $R
timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-logit-error > $OUT/plain.json 2> $OUT/plain.err
ATHENA_FORCE_DP=1 timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-logit-error > $OUT/dp.json 2> $OUT/dp.err
ATHENA_FORCE_DP=1 ATHENA_DP_SIDE_TABLES=1 timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-logit-error > $OUT/dp_side.json 2> $OUT/dp_side.err
ATHENA_FORCE_DP=1 ATHENA_WGRAD_SPLIT=0 timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-logit-error > $OUT/dp_nosplit.json 2> $OUT/dp_nosplit.err
for f in plain dp dp_side dp_nosplit; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d.get('step_kind'), d['roofline_parts']['dense_wgrad'])
"; done
