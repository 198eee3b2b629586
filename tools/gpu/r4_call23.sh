#!/bin/bash
# round 4, GPU call 23: texture-addresser / L1 counters of the step's kernels (is the gather path, not HBM, what the pixel kernels wait on?)
set +e
O=gpurun_out/r4c23
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
timeout 120 rocprofv3 -L > $R/$O/avail.txt 2>&1
grep -o "\bTA_[A-Z_a-z]*\|\bTCP_[A-Z_a-z]*\|\bTD_[A-Z_a-z]*\|GRBM_[A-Z_a-z]*" $R/$O/avail.txt | sort -u > $R/$O/avail_ta_tcp.txt; wc -l $R/$O/avail_ta_tcp.txt
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TD_TC_STALL_sum TD_STORE_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -- python $R/tools/step_pmc.py > $R/$O/p$i.log 2>&1 ; echo "set $i rc=$?"
done
python - <<P
import csv, glob, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("$R/$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / n[k][c] for c in acc[k]} for k in acc if "at::" not in k and "Cijk" not in k and "rocprim" not in k and "rocclr" not in k}
json.dump(out, open("$R/$O/r04_step_ta_tcp_pmc.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("GRBM_GUI_ACTIVE", 0))[:14]:
    print(k[:36].ljust(36), {c.replace("_sum", ""): int(v) for c, v in out[k].items()})
P
rm -rf $R/$O/p?
