#!/bin/bash
# round 3, GPU call 12: MFMA counters of the FLAME contractions (the four-counter pass of call 11 came back empty)
set +e
O=gpurun_out/r3c12
mkdir -p $O
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/$O/st_mfma -- python $R/tools/step_pmc.py > $R/$O/st_mfma.log 2>&1 ; echo rc=$?
ls -la $R/$O/st_mfma/*/ | head; wc -l $R/$O/st_mfma/*/*counter_collection.csv
python $R/tools/step_pmc.py --report-mfma $R/$O/st_mfma > $R/$O/r03_flame_mfma_pmc.json 2> $R/$O/mfma_report.err ; echo rc=$?
cat $R/$O/r03_flame_mfma_pmc.json | head -40
cp "$R/$O"/st_mfma/*/*counter_collection.csv "$R/$O/mfma_counter_collection.csv"
