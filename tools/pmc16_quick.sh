# usage (GPU box): bash tools/pmc16_quick.sh "<dbg values>" N CIN COUT H [K]
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc16
mkdir -p $O
DBGS="$1"; shift
for d in $DBGS; do
  export PG_CONV16_DBG=$d
  echo "== dbg $d"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    i=$((i+1)); rm -rf /tmp/p16
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p16 -- python3 $R/tools/conv16_probe.py "$@" > /tmp/p16.log 2>&1
    f=$(find /tmp/p16 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'conv2d_mfma16' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('  '.join(f'{c}={sum(v)/len(v)/1e6:.2f}M' for c, v in sorted(agg.items())))
PY
  done
done
