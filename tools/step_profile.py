"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py: one replayed step (frame_prep_fwd to the kernel before the next one), kernels in launch order
grouped by name (count, total us).  usage: python tools/step_profile.py <kernel_trace.csv | results.db> [--seq | --timeline]
(rocprofv3 --kernel-trace writes a rocpd .db unless --output-format csv is given; both are read)
(--timeline: start offset, duration, HW queue and name of every kernel of the step -- shows what overlaps with what)"""
import collections
import csv
import re
import sys

if sys.argv[1].endswith(".db"):
    # rocprofv3's default output (ROCm 7.2) is a rocpd SQLite database, not CSV: read its `kernels` view
    import sqlite3
    con = sqlite3.connect(sys.argv[1])
    rows = [{"Kernel_Name": n, "Start_Timestamp": str(s), "End_Timestamp": str(e), "Queue_Id": str(q)}
            for n, s, e, q in con.execute("select name, start, end, queue_id from kernels")]
else:
    rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = "frame_prep_fwd" if any("frame_prep_fwd" in r["Kernel_Name"] for r in rows) else "flame_skin_fwd"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else len(starts) // 2
s, e = starts[k], starts[k + 1]            # one step well inside the timed graph replays (bench.py issues a few eager steps at the end)
seq = rows[s:e]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:90]


print("kernels/step %d  wall %.1f us  busy %.1f us" % (len(seq), (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3,
                                                       sum(dur(r) for r in seq)))
if "--timeline" in sys.argv:
    t0 = int(seq[0]["Start_Timestamp"])
    qs = {}
    for r in seq:
        q = qs.setdefault(r.get("Queue_Id", "0"), len(qs))
        print("%8.1f %8.1f  q%d %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur(r), q, short(r["Kernel_Name"])[:64]))
elif "--seq" in sys.argv:
    for r in seq:
        print("%8.1f %s" % (dur(r), short(r["Kernel_Name"])))
else:
    agg = collections.OrderedDict()
    for r in seq:
        k = short(r["Kernel_Name"])
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1
        agg[k][1] += dur(r)
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%8.1f us  x%-3d %s" % (t, n, k))
