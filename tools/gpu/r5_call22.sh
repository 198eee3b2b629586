#!/bin/bash
# round 5, GPU call 22: bench.py through the sharded step once more (world-size-1 RCCL group) after the last host-side changes
set +e
mkdir -p gpurun_out/r5c22
cd "$GRAFT_REPO_ROOT"
VHAP_FORCE_DIST=1 timeout 120 python bench.py --steps 50 --no-cpu-baseline --no-parity --no-stage > gpurun_out/r5c22/bench.json 2> gpurun_out/r5c22/bench.err; echo rc=$?
python -c "
import json
d=json.load(open('gpurun_out/r5c22/bench.json')); print(round(d['ms_per_step'],4), d['config']['sharded_step'], d['config']['tex_sharded'], d['roofline']['frac_isolated'], d.get('supervisor'))"
wc -l gpurun_out/r5c22/bench.json
