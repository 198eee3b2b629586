#!/bin/bash
# round 5, GPU call 17: after the advisor items (AccessLog: unknown owner for pointer-less calls; ingest pointers recorded; multi-view feed de-duplicated): plan / ingest / native tests, the stage loop's rate
set +e
O=gpurun_out/r5c17
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_ingest.py tests/test_native_gpu.py tests/test_energy_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo rc=$?; tail -4 $O/pytest.log | cut -c1-200
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],4), 'stage', d['stage_fps']['value'], d['stage_fps']['runs'], 'deferred', d['config']['deferred_join'])"
