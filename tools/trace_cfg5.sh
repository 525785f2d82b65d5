# Kernel table of ONE replayed step of config 5 (bf16 1024^2 stack): gpurun -- 'bash tools/trace_cfg5.sh' -> gpurun_out/trace_cfg5/by_name.txt
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_cfg5
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/t5_out -- python3 $R/bench.py --mode bf16_1024 --no-cpu-baseline --steps 4 --warmup 2 > $O/log.txt 2>&1
f=$(find /tmp/t5_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" $O <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the timed steps are graph replays of identical kernel sequences: find the period of the tail
names = [r['Kernel_Name'] for r in rows]
per = None
for L in range(20, 400):
    if names[-L:] == names[-2 * L:-L] == names[-3 * L:-2 * L]:
        per = L
        break
last = rows[-per:] if per else rows[-100:]
def short(nm):
    nm = re.sub(r'^void ', '', nm).replace('(anonymous namespace)::', '').replace('pgconv::', '').replace('pgconv16::', '')
    return re.sub(r'\(.*$', '', nm)[:110]
agg = collections.OrderedDict()
t0 = int(last[0]['Start_Timestamp'])
with open(sys.argv[2] + '/last_step.txt', 'w') as out:
    for r in last:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = short(r['Kernel_Name'])
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
        out.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {k}\n")
span = int(last[-1]['End_Timestamp']) - t0
busy = sum(v[1] for v in agg.values())
with open(sys.argv[2] + '/by_name.txt', 'w') as out:
    out.write(f"# one step: {len(last)} kernels (period {per}), span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write(f"{t / 1e3:9.1f} us {c:4d}  {k}\n")
PY
cat $O/by_name.txt
