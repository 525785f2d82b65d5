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


def train_traffic(pre, stamp):
    # config 4: the weight-gradient kernels of one training iteration by family (bench.py --mode train prices two of them: the bf16x3 group = the 16-bit
    # kernel's launches + the splitting passes, and the fp32 stride-1 3x3 kernel)
    d = os.path.join(G, 'traffic_train')
    try:
        fam = {}
        for cname, scale in (('FETCH_SIZE', 2.0 * 1024), ('WRITE_SIZE', 1024.0)):
            for r in csv.DictReader(open(os.path.join(d, cname + '.csv'))):
                k = r['kernel']
                key = ('bf16x3_wgrad' if 'conv2d16_wgrad_x3k' in k or ('conv2d16_wgrad<' in k and k.rstrip('>').endswith('true')) else
                       'bf16x3_split' if 'split3_bf16' in k else 'fp32_wgrad_3x3_s1' if 'conv2d_wgrad<3, 3, 1' in k else
                       'fp16_wgrad' if 'conv2d16_wgrad<' in k else 'fp32_wgrad_other' if 'conv2d_wgrad<' in k else 'reduce' if 'wgrad_reduce' in k else 'other')
                e = fam.setdefault(key, dict(launches=0, fetch_bytes=0.0, write_bytes=0.0))
                e['fetch_bytes' if cname == 'FETCH_SIZE' else 'write_bytes'] += scale * float(r['value_KiB'])
                if cname == 'FETCH_SIZE':
                    e['launches'] += 1
        for e in fam.values():
            e['hbm_bytes_per_launch'] = (e['fetch_bytes'] + e['write_bytes']) / max(e['launches'], 1)
        x3n = fam.get('bf16x3_wgrad', dict(launches=0))['launches']
        x3 = sum(fam[k]['fetch_bytes'] + fam[k]['write_bytes'] for k in ('bf16x3_wgrad', 'bf16x3_split') if k in fam) / max(x3n, 1)
        out = dict(kernel='weight-gradient kernels of one training iteration (config 4, one GPU, batch 4)', measured_on=stamp, families=fam,
                   hbm_bytes_per_launch=x3, hbm_bytes_per_launch_is='bf16x3 group: (16-bit weight-gradient launches + their two splitting passes) / weight-gradient launches',
                   fp32_hbm_bytes_per_launch=fam.get('fp32_wgrad_3x3_s1', {}).get('hbm_bytes_per_launch'),
                   note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --mode train --steps 1 --warmup 1 (tools/traffic_run.sh), the second (timed) iteration; '
                        'FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; the fixed-order reductions (wgrad_reduce) are listed, not added')
        with open(pre + 'traffic_train.json', 'w') as f:
            json.dump(out, f, indent=1)
        print('wrote', os.path.relpath(pre + 'traffic_train.json', ROOT), f'{x3 / 1e6:.1f} MB per bf16x3 weight gradient')
    except OSError:
        print('missing traffic train')


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

    train_traffic(pre, stamp)

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
