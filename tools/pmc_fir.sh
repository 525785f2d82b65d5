# usage (GPU box): bash tools/pmc_fir.sh    -> gpurun_out/pmc_fir/summary.txt: PMC counters of the blur kernels on [8,64,513(516),513] -> 512^2
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fir
mkdir -p $O
: > $O/summary.txt
for mode in 1 0; do
  export PG_FIR_BLUR4=$mode
  echo "== PG_FIR_BLUR4=$mode (1 = separable 16-byte blur kernel, 0 = generic tiled kernel)" >> $O/summary.txt
  python3 $R/tools/fir_probe.py 2>&1 | grep -v amdgpu >> $O/summary.txt
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/pfir
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pfir -- python3 $R/tools/fir_probe.py > /tmp/pfir.log 2>&1
    f=$(find /tmp/pfir -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$i" >> $O/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'upfirdn2d' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    print(f'pass{sys.argv[2]} {k} {c} mean={sum(v)/len(v):.1f} n={len(v)}')
PY
  done
done
cat $O/summary.txt
