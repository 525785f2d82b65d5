# usage (GPU box): bash tools/pmc_copy.sh   -> gpurun_out/pmc_copy/summary.txt: the memory-side counters of tools/pmc_fir.sh on torch.copy_ of the blur's bytes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_copy
mkdir -p $O
python3 $R/tools/copy_probe.py 2>&1 | grep -v amdgpu > $O/summary.txt
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pcopy
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pcopy -- python3 $R/tools/copy_probe.py > /tmp/pcopy.log 2>&1
  f=$(find /tmp/pcopy -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$i" >> $O/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'copy' in r['Kernel_Name'].lower() or 'elementwise' in r['Kernel_Name']:
        agg[(r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
for (g, c), v in sorted(agg.items()):
    print(f'pass{sys.argv[2]} grid={g} {c} mean={sum(v)/len(v):.1f} n={len(v)}')
PY
done
cat $O/summary.txt
