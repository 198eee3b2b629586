"""Per-kernel statistics over ALL steady-state steps of a rocprofv3 --kernel-trace CSV of bench.py (tools/step_profile.py shows one step):
mean / median / p90 duration per kernel name inside the timed replays, and the mean step period.
usage: python tools/trace_stats.py <kernel_trace.csv> [--skip N] (skip the first N steps: capture / warm-up)"""
import collections
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 40
first = "frame_prep_fwd"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
starts = starts[skip:-5]
per = collections.defaultdict(list)
periods = []
for a, b in zip(starts[:-1], starts[1:]):
    periods.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    for r in rows[a:b]:
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])
        n = n.split("(")[0][:60]
        per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
good = [p for p in periods if p < 2 * statistics.median(periods)]
print("steps %d  period mean %.1f us  median %.1f us" % (len(good), statistics.mean(good), statistics.median(good)))
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%8.1f mean %8.1f med %8.1f p90  x%-4d %s" % (statistics.mean(v), statistics.median(v), v2[int(0.9 * (len(v2) - 1))], len(v), n))
