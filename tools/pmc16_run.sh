# usage (GPU box): bash tools/pmc16_run.sh N CIN COUT H [K]   -> gpurun_out/pmc16/summary_<shape>.txt
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc16
mkdir -p $O
TAG="n$1_c$2_o$3_h$4_k${5:-3}"
: > $O/summary_$TAG.txt
python3 $R/tools/conv16_probe.py "$@" >> $O/summary_$TAG.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/p16
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/p16 -- python3 $R/tools/conv16_probe.py "$@" > /tmp/p16.log 2>&1
  f=$(find /tmp/p16 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$i" >> $O/summary_$TAG.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'conv2d_mfma16' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()):
    print(f'pass{sys.argv[2]} {c} mean={sum(v)/len(v):.1f} n={len(v)}')
PY
done
cat $O/summary_$TAG.txt
