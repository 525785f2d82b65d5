"""Build container, after `gpurun -- 'bash tools/profile_round.sh'`: copy what that call measured from gpurun_out/ into
profiles/rNN_* with the commit it was measured on in every header.   python tools/collect_profiles.py 02 [commit the call ran on]"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def main(rnd, commit=None):
    commit = commit or subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(['git', 'status', '--porcelain', '--', 'pasta-gan-plusplus_amd', 'bench.py'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    stamp = f'commit {commit}' if len(sys.argv) > 2 else f'commit {commit}{" + uncommitted changes" if dirty else ""}'
    pre = os.path.join(P, f'r{rnd}_')

    def copy_csv(src, dst, header):
        if not os.path.isfile(src):
            print('missing', src)
            return
        with open(src) as f, open(dst, 'w') as o:
            o.write(f'# {header} [{stamp}]\n')
            o.write(f.read())
        print('wrote', os.path.relpath(dst, ROOT))

    for tag, what in (('cfg2', 'python bench.py --no-cpu-baseline --steps 7 --warmup 2 (config 2, fp32 512^2 N=8)'),
                      ('cfg5', 'python bench.py --mode bf16_1024 --no-cpu-baseline --steps 7 --warmup 2 (config 5, bf16 1024^2 N=4)'),
                      ('train', 'python bench.py --mode train --no-cpu-baseline --steps 3 --warmup 1 (config 4, one GPU, batch 4)')):
        copy_csv(os.path.join(G, f'prof_{tag}', 'kernel_stats.csv'), pre + f'{tag}_kernel_stats.csv', f'rocprofv3 --kernel-trace --stats -- {what}')
        copy_csv(os.path.join(G, f'prof_{tag}', 'kernel_stats_timed.csv'), pre + f'{tag}_kernel_stats_timed.csv',
                 f'the same run, kernels of the TIMED region only (tools/window_stats.py over the kernel trace): {what}')
        copy_csv(os.path.join(G, f'prof_{tag}', 'by_shape.csv'), pre + f'{tag}_conv_by_shape.csv',
                 f'bench.py --conv-breakdown under the same command: conv launches of one step grouped by geometry / algorithm / shape')

    # HBM traffic of the dominant kernel, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)
    for tag, kernel in (('cfg2', 'conv2d_wino4<MODE,TAIL>'), ('cfg5', 'conv2d_mfma16<T,...> + conv2d_up2f16<T,...> (every 16-bit convolution launch of a step)')):
        d = os.path.join(G, f'traffic_{tag}')
        try:
            rd = [float(r['value_KiB']) for r in csv.DictReader(open(os.path.join(d, 'FETCH_SIZE.csv')))]
            wr = [float(r['value_KiB']) for r in csv.DictReader(open(os.path.join(d, 'WRITE_SIZE.csv')))]
        except OSError:
            print('missing traffic', tag)
            continue
        algo = per_step = None
        bj = os.path.join(G, f'bench_{tag}.json')
        if os.path.isfile(bj):
            try:
                line = json.loads(open(bj).read().strip().splitlines()[-1])
                algo = line['roofline'].get('algorithmic_bytes_per_launch')
                per_step = line['roofline'].get('launches_per_step')
            except (ValueError, KeyError, IndexError):
                pass
        fetch, write = 2.0 * 1024 * sum(rd) / max(len(rd), 1), 1024 * sum(wr) / max(len(wr), 1)
        out = dict(kernel=kernel, measured_on=stamp, launches_per_step=per_step, launches_sampled=len(rd), fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
                   hbm_bytes_per_launch=fetch + write, algorithmic_bytes_per_launch=algo,
                   note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 (tools/traffic_run.sh); FETCH_SIZE '
                        'doubled per the gfx950 correction of MI355X_MICROARCH.md; per launch = mean over that kernel\'s launches in the second half of the profiled process (launches_sampled; config 5 runs an eager and a graph pass there); launches_per_step is bench.py\'s count')
        with open(pre + f'traffic_{tag}.json', 'w') as f:
            json.dump(out, f, indent=1)
        print('wrote', os.path.relpath(pre + f'traffic_{tag}.json', ROOT), f'{(fetch + write) / 1e6:.1f} MB/launch')

    for src, dst, what in ((os.path.join(G, 'pmc', 'summary.txt'), pre + 'pmc_wino.txt', 'tools/pmc_run.sh: SQ counters of conv2d_wino4 (tools/pmc_probe.py winograd4: N8 128->128 256^2 = grid 196608 x ..., N8 64->64 512^2)'),
                           (os.path.join(G, 'pmc16', 'summary_n4_c64_o64_h512_k3.txt'), pre + 'pmc_mfma16.txt', 'tools/pmc16_run.sh 4 64 64 512: counters of conv2d_mfma16 (bf16 3x3 64->64 at 512^2, N=4)')):
        copy_csv(src, dst, what)
    for tag in ('cfg2', 'cfg5', 'train'):
        src = os.path.join(G, f'bench_{tag}.json')
        if os.path.isfile(src):
            with open(src) as f, open(pre + f'bench_{tag}.json', 'w') as o:
                o.write(f.read().strip().splitlines()[-1] + '\n')
            print('wrote', os.path.relpath(pre + f'bench_{tag}.json', ROOT))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else '02', sys.argv[2] if len(sys.argv) > 2 else None)
