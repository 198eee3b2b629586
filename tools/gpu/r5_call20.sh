#!/bin/bash
# round 5, GPU call 20: after the AccessLog change (strided views -> whole storage): does the shipped step still defer its join?
set +e
O=gpurun_out/r5c20
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 150 python bench.py --steps 100 --no-cpu-baseline --no-parity --no-stage > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],4), 'deferred', d['config']['deferred_join'])"
