# GPU box: HBM traffic of the dominant kernel (conv2d_wino) -- rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over
# bench.py (1 timed step), as MI355X_MICROARCH.md prescribes; writes gpurun_out/traffic/{fetch,write}.csv (wino launches only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/$c.log 2>&1
  f=$(find /tmp/tr_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$O/$c.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv2d_wino' in r['Kernel_Name']]
rows = rows[len(rows) // 2:]          # the timed step (second of the two forward passes)
with open(sys.argv[2], 'w') as f:
    f.write('dispatch,kernel,grid_threads,counter,value_KiB\n')
    for r in rows:
        f.write(f'{r["Dispatch_Id"]},"{r["Kernel_Name"].split("(")[0]}",{r["Grid_Size"]},{r["Counter_Name"]},{float(r["Counter_Value"]):.3f}\n')
print(len(rows), 'launches', sum(float(r['Counter_Value']) for r in rows) / max(len(rows), 1), 'KiB/launch')
PY
done
