"""Per-kernel SQ counters of one optimiser step (rocprofv3 --pmc SQ_... -- python tools/step_pmc.py): where a kernel's wave-cycles go.
WAIT_ANY (waves parked at s_waitcnt / barriers) + WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES (MI355X_MICROARCH.md,
"rocprofv3 PMC slots").   python tools/sq_pmc_report.py <dir with *counter_collection.csv>  -> JSON"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    out = {}
    for k, c in acc.items():
        n = max(len(calls[k]), 1)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        row = {"launches": n}
        for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"):
            if name in c:
                row[name + "_per_launch"] = c[name] / n
        if wc:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if name in c:
                    row[name.replace("SQ_", "").lower() + "_frac_of_wave_cycles"] = round(c[name] / wc, 4)
        out[k] = row
    order = sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES_per_launch", 0))
    print(json.dumps({k: out[k] for k in order}, indent=1))


if __name__ == "__main__":
    main()
