"""GPU box helper: kernel statistics of the TIMED region only.  rocprofv3's --stats covers the whole process -- for the training
step that is dominated by MIOpen's one-off solver search during warm-up.  This reads the kernel trace, keeps the kernels that
started in the last `steps * ms_per_step` milliseconds before the end of the run's last convolution kernel, and writes a stats CSV in the
--stats format.   python tools/window_stats.py <kernel_trace.csv> <bench log with the JSON line> <out.csv>"""
import csv
import json
import sys


def main(trace, log, out):
    line = None
    for l in open(log):
        l = l.strip()
        if l.startswith('{') and '"ms_per_step"' in l:
            line = json.loads(l)
    window_ns = line['steps'] * line['ms_per_step'] * 1e6
    rows = list(csv.DictReader(open(trace)))
    conv = [r for r in rows if 'conv2d_' in r['Kernel_Name']]              # the timed region ends with this package's last convolution
    t_end = max(int(r['End_Timestamp']) for r in (conv or rows))           # (what follows are the bench's own finiteness checks)
    keep = [r for r in rows if t_end - window_ns <= int(r['Start_Timestamp']) <= t_end]
    agg = {}
    for r in keep:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg.setdefault(r['Kernel_Name'], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    with open(out, 'w') as f:
        f.write(f'# kernels that started within the timed region ({line["steps"]} steps x {line["ms_per_step"]} ms); busy {total / 1e6:.1f} ms of {window_ns / 1e6:.1f} ms\n')
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{name}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / max(total, 1):.2f},{a[2]},{a[3]}\n')
    print(f'{len(keep)} of {len(rows)} kernels in the window; busy {total / 1e6:.1f} ms of {window_ns / 1e6:.1f} ms')


if __name__ == '__main__':
    main(*sys.argv[1:4])
