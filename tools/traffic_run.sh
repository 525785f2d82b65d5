# GPU box: HBM traffic of a mode's dominant kernel -- rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over bench.py
# (1 timed step), as MI355X_MICROARCH.md prescribes.  usage: bash tools/traffic_run.sh <tag> <kernel substring> <bench.py args...>
#   -> gpurun_out/traffic_<tag>/{FETCH_SIZE,WRITE_SIZE}.csv (that kernel's launches of the timed step only)
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; TAG=${1:-cfg2}; KSUB=${2:-conv2d_wino}; shift; shift
O=$R/gpurun_out/traffic_$TAG
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_$c -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 "$@" > $O/$c.log 2>&1
  f=$(find /tmp/tr_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$O/$c.csv" "$KSUB" <<'PY'
import csv, sys
import re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[3], r['Kernel_Name'])]      # <kernel substring> is a regular expression
rows = rows[len(rows) // 2:]          # the timed step (second of the two passes)
with open(sys.argv[2], 'w') as f:
    f.write('dispatch,kernel,grid_threads,counter,value_KiB\n')
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").removeprefix("void ").split("(")[0]
        f.write(f'{r["Dispatch_Id"]},"{name}",{r["Grid_Size"]},{r["Counter_Name"]},{float(r["Counter_Value"]):.3f}\n')
print(len(rows), 'launches', sum(float(r['Counter_Value']) for r in rows) / max(len(rows), 1), 'KiB/launch')
PY
done
