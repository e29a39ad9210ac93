#!/bin/bash
# Compact per-kernel register / scratch / occupancy table for one .hip file (gfx950).
# usage: scripts/kernel_resources.sh athena_amd/csrc/gemm.hip [name filter]
src="$1"; filt="${2:-.}"
cd "$(dirname "$src")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$(basename "$src")" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    if "error" in line: print(line.rstrip())
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.split()[0]] = v
print("%-100s %5s %5s %7s %4s %7s" % ("kernel", "vgpr", "agpr", "scratch", "occ", "lds"))
for r in rows:
    print("%-100s %5s %5s %7s %4s %7s" % (r["name"][:100], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
' | grep -E "kernel |$filt"
