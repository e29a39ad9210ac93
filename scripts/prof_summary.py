"""Summarise a rocprofv3 results database: per-kernel calls, total and average duration."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path, steps):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms over %d kernels (%.3f ms per step at %d steps)" %
          (tot / 1e6, sum(r[1] for r in rows), tot / 1e6 / steps, steps))
    print("%-110s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for n, cnt, s, a, mn, mx in rows:
        print("%-110s %8d %12.3f %10.2f %10.2f %10.2f %6.2f" % (short(n), cnt, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * s / tot))
    # kernel families as bench.py groups them (C-ABI entry points)
    fams = [("GEMM cores (gemm_mfma*, gemm_big*, layer_tail*, panel_*, splitk_reduce*)", ("gemm_mfma", "gemm_big", "layer_tail", "panel_gemm", "panel_k", "splitk_reduce")),
            ("attention (mha_*)", ("mha_",)),
            ("CTC (ctc_*, rescale_unless*)", ("ctc_", "rescale_unless"))]
    print()
    print("%-50s %10s %14s %12s" % ("family", "calls/step", "ms per step", "avg_us"))
    for label, keys in fams:
        sel = [r for r in rows if any(k in r[0] for k in keys)]
        cnt = sum(r[1] for r in sel)
        t = sum(r[2] for r in sel)
        if cnt:
            print("%-50s %10.1f %14.3f %12.2f" % (label, cnt / steps, t / 1e6 / steps, t / 1e3 / cnt))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
