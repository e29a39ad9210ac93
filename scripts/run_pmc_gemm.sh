#!/bin/bash
# Counters of the large-tile GEMM core's main loop (4096^3 product and the conv2-shaped product) and
# the shader clock beside a queue of such products -> gpurun_out/pmc_gemm/ (summary: pmc_gemm_loop.txt).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -o p -- python3 $R/scripts/pmc_gemm_probe.py > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o p -- python3 $R/scripts/pmc_gemm_probe.py > $OUT/grbm.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python3 $R/scripts/pmc_gemm_probe.py > $OUT/trace.log 2>&1
cd $R
hipcc --offload-arch=gfx950 -O3 -I include scripts/probes/clock_under_gemm.hip -L athena_amd/lib -lathena_hip -Wl,-rpath,$R/athena_amd/lib -o /tmp/clock_under_gemm 2>/dev/null
timeout -k 10 120 /tmp/clock_under_gemm > $OUT/clock.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "gemm_big" not in r["Kernel_Name"]:
                continue
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), c in per.items():
            for n, v in c.items():
                acc[k][n].append(v)
    return acc
lines = []
for d in ("sq", "grbm"):
    for k, c in sorted(counters(d).items()):
        lines.append(k[:90])
        for n, v in sorted(c.items()):
            lines.append("   %-28s %.4g   (mean of %d launches)" % (n, sum(v) / len(v), len(v)))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_big" in r["Name"]:
            lines.append("%s: calls %s, average %.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
lines.append(open(out + "/clock.txt").read())
open(out + "/pmc_gemm_loop.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/sq $OUT/grbm $OUT/trace
