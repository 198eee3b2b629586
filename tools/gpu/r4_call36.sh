#!/bin/bash
# round 4, GPU call 36: bench.py after the last edits (median-of-three stage timing), shortened
set +e
O=gpurun_out/r4c36
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity > $O/bench.json 2> $O/bench.err; echo rc=$?
python -c "import json; d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],4), d['value'], {k: d['stage_fps'][k] for k in ('value','runs','steps')})"
tail -2 $O/bench.err | cut -c1-200
