#!/bin/bash
# Dev tool (GPU box): same-box A/B of bench.py --mode bf16_1024 (config 5) under environment settings.
#   tools/ab_cfg5.sh OUTDIR "NAME1:VAR=VAL VAR=VAL" "NAME2:..."      (a trailing '+csv' on a name also writes the per-shape table)
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    csv=""
    case "$name" in *+csv) name=${name%+csv}; csv="--conv-breakdown $out/$name.csv";; esac
    env $envs python bench.py --mode bf16_1024 --no-cpu-baseline $csv > "$out/$name.json" 2> "$out/$name.err"
    python - "$out/$name.json" "$name" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline', {})
    print(f"{sys.argv[2]:28s} {d['value']:9.1f} img/s {d['ms_per_step']:7.3f} ms  hbm_frac {r.get('frac')} mfma_frac {r.get('mfma_frac')}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
P
done
