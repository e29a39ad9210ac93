# Plain step against the forced single-rank data-parallel step, with and without the step-owned tables,
# alternating on one box (300 steps each, twice); then the model and data-parallel tests.
set -e
cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-logit-error"
for i in 1 2; do
 echo plain new; timeout -k 10 200 $B | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
 echo plain old; ATHENA_STEP_TABLES=0 timeout -k 10 200 $B | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
 echo dp new; ATHENA_FORCE_DP=1 timeout -k 10 200 $B | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
 echo dp old; ATHENA_FORCE_DP=1 ATHENA_STEP_TABLES=0 timeout -k 10 200 $B | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
timeout -k 10 600 python3 -u -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
