cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_out -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $O/log.txt 2>&1
f=$(find /tmp/tr_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/last_step.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 3
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
def short(nm):
    nm = re.sub(r'^void ', '', nm).replace('(anonymous namespace)::', '').replace('pgconv::', '')
    if nm.startswith('at::native::'):
        m = re.search(r'(MulFunctor|CUDAFunctor_add|CUDAFunctorOnOther_add|DivFunctor|direct_copy_kernel|compare_scalar|CompareEq|AbsFunctor|ArgMaxOps|sum_functor|and_kernel|index_kernel|neg|sub|FillFunctor|where)', nm)
        return 'aten ' + nm.split('<')[0].replace('at::native::', '') + ' [' + (m.group(1) if m else '?') + ']'
    return re.sub(r'\(.*$', '', nm)[:70]
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}")
PY
wc -l $O/last_step.txt
