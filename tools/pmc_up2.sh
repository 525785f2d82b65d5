# GPU box: counters of the `up = 2` transposed convolution, bf16x3 form (conv2d_up2x3) and fp32 form (conv2d_up2), separate --pmc passes.
#   bash tools/pmc_up2.sh   -> gpurun_out/pmc_up2/summary.txt
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_up2
mkdir -p $O; rm -f $O/summary.txt
for form in x3 fp32; do
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p_${form}_$i -- python3 $R/tools/pmc_up2_probe.py $form > $O/p_${form}_$i.log 2>&1
    f=$(find $O/p_${form}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$form" "$i" >> $O/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'conv2d_up2' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, g, c), v in sorted(agg.items()):
    print(f'{sys.argv[2]} pass{sys.argv[3]} {k} grid={g} {c} mean={sum(v)/len(v):.1f} n={len(v)}')
PY
    rm -rf $O/p_${form}_$i
  done
  # wall time of the same launches (kernel trace only)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$form -- python3 $R/tools/pmc_up2_probe.py $form > $O/t_$form.log 2>&1
  f=$(find $O/t_$form -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -i "conv2d_up2" "$f" | sed "s/^/$form stats /" >> $O/summary.txt
  rm -rf $O/t_$form
done
cat $O/summary.txt
