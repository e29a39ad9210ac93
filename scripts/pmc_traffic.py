"""HBM traffic per launch of the benchmark's kernel families from two rocprofv3 --pmc passes.

usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

The number of training steps in a pass is the number of clip_adam_kernel dispatches.

The passes are ``rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py ...`` and the same
with WRITE_SIZE.  FETCH_SIZE is in KiB-like units of 1024 B on this ROCm (the CSV states the value per
dispatch); on gfx950 it reports half of the bytes of a wide streaming read, so it is doubled.
"""
import csv, json, sys
from collections import defaultdict

FAMILIES = {
    "gemm": ("gemm_mfma", "gemm_big", "layer_tail", "panel_gemm", "panel_k", "splitk_reduce"),
    "attention": ("mha_",),
    "ctc": ("ctc_", "rescale_unless"),
}


def family_of(name):
    for fam, keys in FAMILIES.items():
        if any(k in name for k in keys):
            return fam
    return None


def load(path, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            if "clip_adam" in row["Kernel_Name"]:
                calls["_steps"] += 1
            fam = family_of(row["Kernel_Name"])
            if fam is None:
                continue
            tot[fam] += float(row["Counter_Value"])
            calls[fam] += 1
    return tot, calls


def main(fetch_csv, write_csv, out):
    f_tot, f_calls = load(fetch_csv, "FETCH_SIZE")
    w_tot, w_calls = load(write_csv, "WRITE_SIZE")
    steps = f_calls["_steps"]
    res = {"workload": "aishell_b32_t1000", "unit": "bytes", "steps_profiled": steps,
           "note": "FETCH_SIZE and WRITE_SIZE are reported in KB (x1024); FETCH_SIZE doubled (gfx950 counts a "
                   "128-byte request as 64 bytes); separate passes", "families": {}}
    for fam in FAMILIES:
        if not f_calls.get(fam):
            continue
        rd = 2.0 * f_tot[fam] * 1024.0
        wr = w_tot.get(fam, 0.0) * 1024.0
        n = f_calls[fam]
        res["families"][fam] = {"launches": n, "launches_per_step": n / steps,
                                "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / max(w_calls.get(fam, n), 1),
                                "hbm_bytes_per_launch": rd / n + wr / max(w_calls.get(fam, n), 1),
                                "hbm_bytes_per_step": (rd + wr) / steps}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
