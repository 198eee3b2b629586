#!/bin/bash
# round 4, GPU call 22: SQ counters of every kernel of the step: parked at waitcnt vs issue-stalled vs issuing
set +e
O=gpurun_out/r4c22
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/$O/sq -- python $R/tools/step_pmc.py > $R/$O/sq.log 2>&1 ; echo rc=$?
python $R/tools/sq_pmc_report.py $R/$O/sq > $R/$O/r04_step_sq_pmc.json 2> $R/$O/sq_report.err; echo rc=$?
python -c "
import json; d=json.load(open('$R/$O/r04_step_sq_pmc.json'))
for k in list(d): print(k[:44].ljust(44), {a.replace('_frac_of_wave_cycles',''): b for a, b in d[k].items() if 'frac' in a}, int(d[k].get('SQ_WAVES_per_launch',0)))" | grep -v "at::\|Cijk\|rocprim\|rocclr"
rm -rf $R/$O/sq
