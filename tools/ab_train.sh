#!/bin/bash
# Dev tool (GPU box): same-box A/B of bench.py --mode train (config 4) under environment settings.   tools/ab_train.sh OUTDIR "NAME:VAR=VAL ..." ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline > "$out/$name.json" 2> "$out/$name.err"
    python - "$out/$name.json" "$name" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} {d['ms_per_step']:8.1f} ms/it  wgrad frac {d.get('roofline', {}).get('frac')}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
P
done
