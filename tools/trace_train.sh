# Kernel sequence of the LAST training iteration (config 4, one GPU) with durations -- which torch kernels still sit between the native ones.
#   gpurun -- 'bash tools/trace_train.sh'  ->  gpurun_out/trace_train/{last_step.txt, by_name.txt}
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_train
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_out -- python3 $R/bench.py --mode train --no-cpu-baseline --steps 2 --warmup 2 > $O/log.txt 2>&1
f=$(find /tmp/tt_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" $O <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last iteration = the kernels after the last-but-one Adam multi_tensor burst of the G phase; simpler: the last quarter by count is NOT exact
# (warm-up iterations carry first-call work), so cut by the EMA update: its `lerp` multi-tensor kernels end every iteration
ends = [i for i, r in enumerate(rows) if 'LerpFunctor' in r['Kernel_Name'] or 'lerp' in r['Kernel_Name'].lower()]
cuts = [e for j, e in enumerate(ends) if j + 1 == len(ends) or ends[j + 1] - e > 1000]
last = rows[cuts[-6] + 1:cuts[-1] + 1] if len(cuts) >= 6 else rows[-len(rows) // 4:]      # 4 Adam steps (their lerp) + the EMA update per iteration
t0 = int(last[0]['Start_Timestamp'])
def short(nm):
    nm = re.sub(r'^void ', '', nm).replace('(anonymous namespace)::', '').replace('pgconv::', '')
    if nm.startswith('at::native::'):
        m = re.search(r'(MulFunctor|CUDAFunctor_add|CUDAFunctorOnSelf_add|CUDAFunctorOnOther_add|DivFunctor|direct_copy_kernel|float16|compare_scalar|CompareEq|AbsFunctor|ArgMaxOps|sum_functor|and_kernel|index_kernel|neg|sub|FillFunctor|where|threshold|clamp|addcmul|sqrt|rsqrt|pow|batch_norm\w+|CatArray\w+|indexSelect\w+|Lerp|addcdiv|mean|var|std|exp|sigmoid|softplus|log)', nm)
        return 'aten ' + nm.split('<')[0].replace('at::native::', '') + ' [' + (m.group(1) if m else '?') + ']'
    return re.sub(r'\(.*$', '', nm)[:90]
busy = 0
agg = collections.OrderedDict()
with open(sys.argv[2] + '/last_step.txt', 'w') as out:
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        busy += e - s
        k = short(r['Kernel_Name'])
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
        out.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {k}\n")
span = int(last[-1]['End_Timestamp']) - t0
with open(sys.argv[2] + '/by_name.txt', 'w') as out:
    out.write(f"# last iteration: {len(last)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f"{t / 1e6:9.3f} ms {c:6d}  {k}\n")
PY
head -60 $O/by_name.txt
