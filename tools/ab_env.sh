#!/bin/bash
# A/B of step-executor switches on one box: tools/ab_env.sh "VAR=val VAR2=val" "VAR3=val" ...   ("-" = defaults); prints ms/step per variant.
# Every variant runs `bench.py --no-cpu-baseline` twice (box noise is ~1 %).
STEPS=${STEPS:-300}
CONFIG=${CONFIG:-2}
for v in "$@"; do
    for rep in 1 2; do
        if [ "$v" = "-" ]; then e=""; else e="$v"; fi
        out=$(env $e python bench.py --config $CONFIG --steps $STEPS --warmup 30 --no-cpu-baseline 2>/dev/null | tail -n 1)
        ms=$(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f ms  ri_in_step %s' % (d['ms_per_step'], d['roofline'].get('us_in_step_deferred')))" "$out" 2>/dev/null || echo "FAILED: $out")
        echo "[$v] $ms"
    done
done
