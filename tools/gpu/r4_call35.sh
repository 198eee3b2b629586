#!/bin/bash
# round 4, GPU call 35: stage_fps with and without the supervising parent (call 34 measured 13.9 k frames/s against 17.6 k in call 30)
set +e
O=gpurun_out/r4c35
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
VHAP_BENCH_CHILD=1 timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_direct.json 2> $O/bench.err; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_direct.json')); print('direct', round(d['ms_per_step'],4), d['value'], d.get('stage_fps'))" | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --no-parity > $O/bench_sup.json 2> $O/bench.err; echo rc=$?
python -c "import json; d=json.load(open('$O/bench_sup.json')); print('supervised', round(d['ms_per_step'],4), d['value'], d.get('stage_fps'))" | cut -c1-400
nproc; uptime
