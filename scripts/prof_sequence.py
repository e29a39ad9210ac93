"""Kernel sequence of ONE step out of a rocprofv3 results database: the dispatches between the last
two launches of a marker kernel (default: the step counter's increment), in start order, with the
queue each ran on, its start relative to the first, its duration and the gap to the end of the
dispatch in front of it on the same queue.

    python scripts/prof_sequence.py <results.db> [marker substring] [which]

``which`` counts the marker's launches from the end (default 1: the last complete step; the steps of
a benchmark's timed region lie further back than those of its per-kernel timing passes).  The
number of dispatches of every step is printed first.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
marker = sys.argv[2] if len(sys.argv) > 2 else "step_increment_kernel"
rows = list(cur.execute(
    "select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from %s d join %s s "
    "on d.kernel_id = s.id order by d.start" % (kd, ks)))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
if len(marks) < 2:
    sys.exit("marker %r seen %d times" % (marker, len(marks)))
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print("dispatches per step:", [marks[i + 1] - marks[i] for i in range(len(marks) - 1)])
a, b = marks[-1 - which], marks[-which]
t0 = rows[a][1]
last_end = {}


def short(name):
    name = name.split("(")[0] if not name.startswith("(") else name
    for junk in ("void ", "at::native::", "(anonymous namespace)::"):
        name = name.replace(junk, "")
    return name[:70]


print("%d dispatches in the step, %.3f ms from first start to last end" %
      (b - a, (max(r[2] for r in rows[a:b]) - t0) / 1e6))
for r in rows[a:b]:
    q = r[3]
    gap = (r[1] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r[2]
    print("q%-3d %9.1f us  dur %7.1f  gap %6.1f  grid %7d  %s" %
          (q, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[4] // max(r[5], 1), short(r[0])))
