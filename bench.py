#!/usr/bin/env python3
"""Headline benchmark: 512-res try-on images/s through the SynthesisNetwork forward
(BASELINE.json config 2: SynthesisNetworkFull_v18, 512^2, N=8 per GPU, fp32, eval,
random latents + style maps, noise_mode='const', argmax parsing path as test.py runs it).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Started through torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS
a rank; started plainly as `python bench.py --gpus N` it is the launcher: before importing torch or touching the GPU it
starts N fresh child processes of itself (training/launch.py, what the reference's train.py:563-568 does with
torch.multiprocessing.spawn) and waits.  Every rank asserts WORLD_SIZE == --gpus.  Every rank runs the same workload on
its own batch (independent images, no data-path collective: "weak" scaling); RCCL is used only for
the barrier and the max-over-ranks of the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- dominant kernel = conv2d_wino4 (Winograd F(4x4,3x3) convolution, csrc/conv2d_wino4.h:
                  the stride-1 3x3 layers with Cin >= 64, ~58 % of the step): sum of the FLOPs it executes
                  on the matrix pipe (1/4 of the direct-convolution FLOPs) over its launches in the timed region /
                  sum of their durations measured with HIP events on the launch stream, vs 157.3 TFLOP/s;
                  `traffic` = measured HBM bytes per launch, from the newest committed
                  profiles/rNN_traffic_cfg2.json (rocprofv3 --pmc passes over this same command;
                  PMC counters cannot be read in-process) -- `traffic_source` names the file.
                  --mode bf16_1024 (config 5): dominant kernel = conv2d_mfma16, HBM-bound.
  cpu_baseline -- the CPU oracle (oracle/network_ref.py, a port) timed on this host at N=1 on a
                  bounded sample (rank 0, --gpus 1 only).
  secondary    -- (--gpus 1, headline mode only) BASELINE configs 3, 5 and 4 (one-GPU share of the training step: batch 4, `--mode train`) run as
                  short child processes of the same script after the headline's timed region: their value / ms_per_step / roofline, so that
                  those numbers are driver-timed too (skipped under a profiler).
                  Headline fields are untouched by it.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
sys.path.insert(0, ROOT)

torch = None                     # imported by main() AFTER the launcher branch: the launcher parent must not touch the GPU runtime

CFG2 = dict(w_dim=512, img_resolution=512, img_channels=3, channel_base=32768, channel_max=512, conv_clamp=256)
BATCH_PER_GPU = 8
GFLOP_PER_IMAGE = 962.2          # SURVEY.md section 8d (conv FLOPs of one 512^2 image)
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def committed_traffic(tag):
    """HBM bytes per launch of a mode's dominant kernel: PMC counters cannot be read in-process, so the number comes from the
    newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/traffic_run.sh ->
    profiles/rNN_traffic_<tag>.json; round 1's file is profiles/r01_traffic.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_traffic_{tag}.json')))
    if not files and tag == 'cfg2':
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r01_traffic.json')))
    for path in reversed(files):
        try:
            with open(path) as f:
                return round(json.load(f)['hbm_bytes_per_launch']), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def make_inputs(n, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    inp = dict(
        ws=r(n, 14, 512), pose_feat=r(n, 512, 8, 8),
        cat_feat={str(k): r(n, 64, k, k) for k in (512, 256, 128, 64)},
        du=torch.rand(n, 3, 512, 512, generator=g) * 2 - 1, dl=torch.rand(n, 3, 512, 512, generator=g) * 2 - 1,
        mu=(torch.rand(n, 1, 512, 512, generator=g) > 0.5).float(), ml=(torch.rand(n, 1, 512, 512, generator=g) > 0.5).float(),
    )
    mv = lambda t: t.to(device)
    return dict(ws=mv(inp['ws']), pose_feat=mv(inp['pose_feat']), cat_feat={k: mv(v) for k, v in inp['cat_feat'].items()},
                du=mv(inp['du']), dl=mv(inp['dl']), mu=mv(inp['mu']), ml=mv(inp['ml']))


def run_net(net, inp):
    return net(inp['ws'], inp['pose_feat'], inp['cat_feat'], inp['du'], inp['dl'], inp['mu'], inp['ml'], None, noise_mode='const')


def init_weights(net):
    from training.synthetic import fill_module_
    return fill_module_(net, 'cfg2.')      # name-keyed N(0,1) weights, noise_strength 0.1


def cpu_baseline(max_seconds=40.0):
    """Oracle network on the host cores, N=1, fp32, bounded: one cold image, then up to 2 more while
    the budget lasts; the median is reported.  Threads are capped at 16 (the box reports 256 logical
    CPUs, where oneDNN's small convolutions oversubscribe badly)."""
    from oracle import network_ref as NR
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    torch.set_num_threads(threads)
    net = init_weights(NR.SynthesisNetworkFull_v18(**CFG2)).eval()
    inp = make_inputs(1, 'cpu', seed=0)
    times = []
    t_start = time.perf_counter()
    with torch.no_grad():
        while len(times) < 3 and (not times or time.perf_counter() - t_start + min(times) < max_seconds):
            t0 = time.perf_counter()
            run_net(net, inp)
            times.append(time.perf_counter() - t0)
    best = sorted(times)[len(times) // 2]          # median
    return dict(value=round(1.0 / best, 5), unit='images/s', cores=threads, kind='port',
                sample=f'oracle/network_ref.py SynthesisNetworkFull_v18 fwd, N=1, 512^2, fp32, {len(times)} image(s) timed one by one '
                       f'({", ".join(f"{t:.1f}s" for t in times)}), median reported; host reports {avail} logical CPUs')


def conv_roofline(timeline, elapsed, steps, images_per_s_per_gpu, traffic_tag, gflop_per_image=None):
    """`roofline` of an fp32 run from the per-launch HIP events of its timed region: the dominant kernel = the convolution algorithm with the largest share
    of the region.  The MFMA roofline is priced on the flops that algorithm EXECUTES on the matrix pipe -- Winograd F(4x4,3x3) (csrc/conv2d_wino4.h):
    36 multiplies per 16 outputs = 1/4 of the direct-convolution count; F(2x2,3x3) (csrc/conv2d_wino.h): 4/9; the direct implicit GEMM: all of it --
    and the direct-convolution-equivalent rate of the same launches is reported next to it (it can exceed the matrix peak)."""
    WORK = {'winograd4x3': 0.25, 'winograd4': 0.25, 'winograd': 4.0 / 9.0, 'direct': 1.0}
    BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md, dense bf16 matrix peak
    KERNEL = {'winograd4x3': 'conv2d_wino4<MODE,TAIL,X3> (Winograd F(4x4,3x3) stride-1 3x3 convolution; the 36 transform-domain GEMMs as six bf16 x bf16 products of exact three-term '
                             'operand splits, fp32 accumulation, v_mfma_f32_32x32x16_bf16)',
              'winograd4': 'conv2d_wino4<MODE,TAIL> (Winograd F(4x4,3x3) stride-1 3x3 convolution, v_mfma_f32_32x32x2_f32)',
              'winograd': 'conv2d_wino<MODE,VEC> (Winograd F(2x2,3x3) stride-1 3x3 convolution, v_mfma_f32_32x32x2_f32)',
              'direct': 'conv2d_mfma<KH,KW,S,BM,KC,XF> (implicit GEMM, v_mfma_f32_32x32x2_f32)'}
    by_algo = {}
    for geo, fl, e0, e1, by in timeline:
        if geo[3] in WORK:
            by_algo.setdefault(geo[3], []).append((fl, e0.elapsed_time(e1) * 1e-3, by))
    if not by_algo:
        return None
    allk = [ft[:2] for v in by_algo.values() for ft in v]
    algo = max(by_algo, key=lambda a_: sum(ft[1] for ft in by_algo[a_]))
    dom, work = [ft[:2] for ft in by_algo[algo]], WORK[algo]
    dom_bytes = sum(ft[2] for ft in by_algo[algo])
    dom_flops, dom_time = sum(f for f, _ in dom), sum(t for _, t in dom)
    achieved = work * dom_flops / dom_time / 1e12 if dom_time > 0 else 0.0
    traffic, traffic_src = committed_traffic(traffic_tag) if traffic_tag else (None, None)
    out = dict(bound='mfma', kernel=KERNEL[algo],
               achieved=round(achieved, 2), peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
               traffic=traffic, traffic_source=traffic_src, algorithmic_bytes_per_launch=round(dom_bytes / max(len(dom), 1)),
               flops_counted=f'flops the kernel executes on the matrix pipe = {work:.4g} x the direct-convolution flops of SURVEY 8d',
               direct_equivalent_tflops=round(dom_flops / max(dom_time, 1e-12) / 1e12, 2),
               direct_equivalent_frac=round(dom_flops / max(dom_time, 1e-12) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),    # SURVEY 8d count / peak (> 1: fewer multiplies than counted)
               launches_per_step=len(dom) // max(steps, 1), avg_launch_ms=round(1e3 * dom_time / max(len(dom), 1), 4),
               time_frac_of_step=round(dom_time / elapsed, 4),
               other_algorithms={a_: dict(launches_per_step=len(v) // max(steps, 1), ms_per_step=round(1e3 * sum(ft[1] for ft in v) / steps, 3),
                                          executed_tflops=round(WORK[a_] * sum(ft[0] for ft in v) / max(sum(ft[1] for ft in v), 1e-12) / 1e12, 2))
                                 for a_, v in by_algo.items() if a_ != algo},
               all_conv_direct_equivalent_tflops=round(sum(f for f, _ in allk) / max(sum(t for _, t in allk), 1e-12) / 1e12, 2),
               conv_time_frac_of_step=round(sum(t for _, t in allk) / elapsed, 4))
    if algo == 'winograd4x3':
        # the matrix pipe executes SIX bf16 multiply-adds per float32 one of the Winograd-domain GEMM: the roofline is priced on that count against the dense bf16 peak
        # (as config 4's bf16x3 weight gradients are), with the float32-equivalent rate -- what the fp32-MFMA form of the same kernel is priced on -- beside it
        out.update(achieved=round(6 * achieved, 2), peak=BF16_MFMA_PEAK_TFLOPS, frac=round(6 * achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                   flops_counted=f'bf16 multiply-adds the kernel executes = 6 x {work:.4g} x the direct-convolution flops of SURVEY 8d (six plane products per float32 product)',
                   fp32_equivalent_tflops=round(achieved, 2), fp32_equivalent_frac_of_fp32_peak=round(achieved / F32_MFMA_PEAK_TFLOPS, 4))
    if gflop_per_image:
        out['end_to_end_direct_equivalent_frac'] = round(images_per_s_per_gpu * gflop_per_image / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)
    return out


def routed_sample_inputs(n, dev, seed):
    """Synthetic inputs of the loader's patch-routing step (reference dataset.py:2555-2700 as test.py:117-160 runs it per sample), in the reference's formats:
    512 x 512 x 3 uint8 garment images and 0 / 255 masks resident on the GPU, OpenPose-18 key points (x, y, confidence) on the host."""
    import numpy as np
    from training import patch_routing as P
    rng = np.random.default_rng(seed)
    joints = dict(cnose=(256, 60), cneck=(256, 110), rshoulder=(200, 120), relbow=(180, 200), rwrist=(170, 270), lshoulder=(312, 120), lelbow=(335, 200),
                  lwrist=(345, 270), rhip=(220, 290), rknee=(215, 390), rankle=(212, 480), lhip=(292, 290), lknee=(297, 390), lankle=(300, 480),
                  reye=(246, 50), leye=(266, 50), rear=(236, 55), lear=(276, 55))

    def kps():
        kp = np.zeros((18, 3))
        for k, (x, y) in joints.items():
            kp[P.ORDER.index(k)] = (x + rng.normal(0, 8.0), y + rng.normal(0, 8.0), 1.0)
        return kp
    samples = []
    for _ in range(n):
        up, lo = (torch.from_numpy(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)).to(dev) for _ in range(2))
        um = torch.zeros(512, 512, 3, dtype=torch.uint8, device=dev); um[90:310, 150:370] = 255
        lm = torch.zeros(512, 512, 3, dtype=torch.uint8, device=dev); lm[270:505, 190:330] = 255
        samples.append((up * (um > 0), lo * (lm > 0), um, lm, kps(), kps()))
    return samples


def route_batch(samples):
    """HIP patch routing of the whole batch (training/patch_routing.py `normalize_batch`: pg_warp_perspective_u8 x 2 + pg_patch_compose_ordered_u8, three launches for
    all samples) -> the generator's routed inputs, assembled on the GPU the way the loader (training/dataset.py) and test.py:126-147 do: c [N, 45, 128, 128],
    de-normalised garments + masks."""
    from training import patch_routing as P
    unit = lambda t: t.permute(0, 3, 1, 2).to(torch.float32) / 127.5 - 1
    norm_img, norm_lower, den_up, _, _ = P.normalize_batch([(up, lo, um, lm, None, ckp, pkp) for up, lo, um, lm, ckp, pkp in samples], 2, device=samples[0][0].device)
    den_lo = torch.stack([s[1] for s in samples])                    # (the loader keeps the person's own lower garment, edge eroded: dataset.py `denorm_lower`)
    mask = lambda t: (t.to(torch.int32).sum(dim=3, keepdim=True) > 0).permute(0, 3, 1, 2).float()
    return dict(c=torch.cat([unit(norm_img), unit(norm_lower)], dim=1), denorm_upper_input=unit(den_up), denorm_lower_input=unit(den_lo),
                denorm_upper_mask=mask(den_up), denorm_lower_mask=mask(den_lo))


def run_generator(args, rank, world, dev, dist):
    """BASELINE config 3 (secondary, --mode generator): full GeneratorFull_v20 inference -- pose encoder, garment-part style encoder
    with its feature pyramid, mapping MLP, then the synthesis network -- at N=16 per GPU on synthetic tensors of the shapes test.py
    feeds it (the patch-routing warp upstream is not part of this path)."""
    from training import networks, replicas
    from torch_utils.ops import conv2d_mfma
    routed = args.mode == 'generator_routed'
    n = args.batch if args.batch != BATCH_PER_GPU else 16
    G = networks.GeneratorFull_v20(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
                                   synthesis_kwargs=dict(channel_base=32768, channel_max=512, conv_clamp=256))
    G = init_weights(G).to(dev).eval()
    g = torch.Generator(device='cpu').manual_seed(200 + rank)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    inp = dict(z=torch.zeros([n, 0], device=dev), c=u(n, 45, 128, 128), retain=u(n, 6, 512, 512), pose=u(n, 5, 512, 512),
               denorm_upper_input=u(n, 3, 512, 512), denorm_lower_input=u(n, 3, 512, 512),
               denorm_upper_mask=(u(n, 1, 512, 512) > 0).float(), denorm_lower_mask=(u(n, 1, 512, 512) > 0).float())
    samples = routed_sample_inputs(n, dev, 300 + rank) if routed else None
    route_events = []

    def step():
        if routed:          # BASELINE config 3 as written: the patch-routing warp is on the critical path of every batch (test.py:117-160)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inp.update(route_batch(samples))
            e1.record()
            route_events.append((e0, e1))
        return G(**inp, noise_mode='const')

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        barrier()
        route_events.clear()
        if routed:
            from training import patch_routing
            patch_routing.traffic_counter = dict(bytes=0, launches=0)
        timeline = conv2d_mfma.start_timeline()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        conv2d_mfma.stop_timeline()
    assert all(torch.isfinite(o).all() for o in out)
    elapsed = replicas.max_over_ranks(elapsed, device=dev)
    if rank == 0:
        extra = {}
        roofline = conv_roofline(timeline, elapsed, args.steps, args.steps * n / elapsed, 'cfg2')
        if roofline:
            roofline['traffic_note'] = 'per launch of the same kernel in the config-2 PMC passes (the synthesis network is the same; N = 8 there)'
            extra['roofline'] = roofline
        if routed:
            from training import patch_routing
            t_route = sum(e0.elapsed_time(e1) for e0, e1 in route_events) * 1e-3
            tc = patch_routing.traffic_counter
            extra['routing'] = dict(kernels='pg_warp_perspective_u8 (image -> patch, patch -> canvas: two launches over every job of the batch) + pg_patch_compose_ordered_u8 (erode + paste of every canvas of the batch, one launch)',
                                    ms_per_step=round(1e3 * t_route / args.steps, 3), ms_per_sample=round(1e3 * t_route / args.steps / n, 3),
                                    time_frac_of_step=round(t_route / elapsed, 4), launches_per_step=tc['launches'] // max(args.steps, 1),
                                    bound='hbm', algorithmic_bytes_per_step=tc['bytes'] // max(args.steps, 1),
                                    achieved_gbs=round(tc['bytes'] / max(t_route, 1e-12) / 1e9, 1), peak_gbs=HBM_PEAK_GBS, frac=round(tc['bytes'] / max(t_route, 1e-12) / 1e9 / HBM_PEAK_GBS, 5),
                                    note='stage time by events around the whole routing of a batch: it includes the host key-point geometry (the quadrilaterals per sample, one batched solve for '
                                         'the homographies of the batch), the two job-table uploads and the batched tensor bookkeeping behind the three native launches',
                                    parity='UNPINNED: HIP == own oracle bit for bit; the oracle restates OpenCV\'s published fixed-point algorithm, cv2 is not available offline (DESIGN.md 6d)')
        print(json.dumps(dict(metric='512-res try-on images/sec (full generator: ' + ('patch routing + ' if routed else '') + 'encoders + mapping + synthesis)', value=round(args.steps * n * world / elapsed, 3),
                              unit='images/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 2),
                              higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic', **extra,
                              config=dict(workload='BASELINE config 3: ' + ('HIP patch routing of every sample (synthetic garments / masks / OpenPose key points in the reference formats) -> ' if routed else '') +
                                                   'GeneratorFull_v20 forward (ConstEncoderNetwork + StyleEncoderNetworkV18 + MappingNetwork + '
                                                   'SynthesisNetworkFull_v18), 512x512, fp32, eval, noise_mode=const, argmax parsing, random-init weights',
                                          images_per_gpu_per_step=n, global_batch=n * world, parallelism=f'replicas x{world}'))), flush=True)


CFG5 = dict(w_dim=512, img_resolution=1024, img_channels=3, channel_base=32768, conv_clamp=256)
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md, "HBM3E peak BW" (spec; 6.29 TB/s measured copy rate)


def cpu_baseline_stack(channel_max, max_seconds=60.0):
    """The float32 oracle stack (oracle/network_ref.py SynthesisStack, a port) on the host cores, N=1: the blocks up to 256^2 of
    the same 1024^2 network (a bounded sample: the two top blocks alone are ~60 % of the FLOPs and minutes of CPU time)."""
    from oracle import network_ref as NR
    from training.synthetic import fill_module_
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    torch.set_num_threads(threads)
    sample_res = 256
    net = fill_module_(NR.SynthesisStack(**dict(CFG5, img_resolution=sample_res, channel_max=channel_max)), 'cfg5.').eval()
    ws = torch.randn([1, net.num_ws, 512], generator=torch.Generator().manual_seed(0))
    t0 = time.perf_counter()
    with torch.no_grad():
        net(ws, noise_mode='const')
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 5), unit=f'{sample_res}^2-prefix images/s', cores=threads, kind='port',
                sample=f'oracle/network_ref.py SynthesisStack fwd, N=1, fp32: blocks 8^2..{sample_res}^2 of the 1024^2 stack (same channel widths), '
                       f'one image, {dt:.1f} s; host reports {avail} logical CPUs')


def run_stack(args, rank, world, dev, dist):
    """BASELINE config 5 (--mode bf16_1024): the StyleGAN2 block stack (SynthesisLayer x2 + ToRGB + skip-image upsample per
    resolution, no SPADE) at 1024^2, N=4 per GPU, every block in bf16 with fp32 accumulation.  The 16-bit
    convolution kernel is near the machine's ridge, so both rooflines of its launches are reported: HBM (algorithmic bytes =
    x + y + packed weights per launch) as `roofline`, the matrix rate next to it."""
    from training import networks, replicas
    from torch_utils.ops import conv2d_mfma
    from training.synthetic import fill_module_
    n = args.batch if args.batch != BATCH_PER_GPU else 4
    net = networks.SynthesisStack(num_fp16_res=8, half_dtype=torch.bfloat16, channel_max=args.channel_max, **CFG5)
    net = fill_module_(net, 'cfg5.').to(dev).eval()
    ws = torch.randn([n, net.num_ws, 512], generator=torch.Generator().manual_seed(rank)).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(args.warmup):
            img = net(ws, noise_mode='const')
        barrier()
        timeline = conv2d_mfma.start_timeline()          # per-launch HIP events (for `roofline`) on an eager pass of the same steps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img = net(ws, noise_mode='const')
        barrier()
        elapsed = eager_elapsed = time.perf_counter() - t0
        conv2d_mfma.stop_timeline()
        if not args.no_graph:
            # the step is launch-bound on the host (~95 kernels, 2.4 ms of GPU time): replay it as one hipGraph (training/graphed.py).
            # Events cannot be recorded inside a replayed graph, hence the eager pass above for the per-kernel durations.
            from training.graphed import GraphedForward
            fwd = GraphedForward(lambda w_: net(w_, noise_mode='const'), [ws], warmup=1)
            for _ in range(args.warmup):
                img = fwd(ws)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                img = fwd(ws)
            barrier()
            elapsed = time.perf_counter() - t0
    assert torch.isfinite(img).all() and img.shape == (n, 3, 1024, 1024)
    elapsed = replicas.max_over_ranks(elapsed, device=dev)
    if rank != 0:
        return
    rows = [(geo, fl, e0.elapsed_time(e1) * 1e-3, by) for geo, fl, e0, e1, by in timeline]
    execution = ('eager launches' if args.no_graph else
                 f'one hipGraph replay per step (eager, with per-launch events: {1e3 * eager_elapsed / args.steps:.3f} ms/step)')
    dom = [r for r in rows if r[0][3] == 'mfma16']
    t_dom = sum(r[2] for r in dom)
    gbs = sum(r[3] for r in dom) / max(t_dom, 1e-12) / 1e9
    tfl = sum(r[1] for r in dom) / max(t_dom, 1e-12) / 1e12
    # SURVEY 8d counts an up = 2 layer as the transposed convolution it is in the reference (2 N Cin Hin Win Cout 9); the launches execute 2x that (the fused-x
    # form: the y half of the FIR in the weights, 18 tap-products per position) or 4x (the composite four-phase form: 36)
    survey = lambda r: r[1] / (2.0 if 'fused-x' in r[0][4] else (4.0 if '4 phases' in r[0][4] else 1.0))
    tfl_survey = sum(survey(r) for r in dom) / max(t_dom, 1e-12) / 1e12
    top = [r for r in dom if '1024x1024' in r[0][4] or '512x512' in r[0][4]]
    line = dict(metric='1024-res bf16 StyleGAN2-stack images/sec (SynthesisStack fwd)', value=round(args.steps * n * world / elapsed, 3), unit='images/s',
                n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True, scaling='weak',
                vs_baseline=None, dtype='bf16', data='synthetic',
                config=dict(workload=f'BASELINE config 5: StyleGAN2 block stack 8^2..1024^2 (SynthesisLayer x2 + ToRGB + skip upsample per resolution), '
                                     f'channel_base 32768, channel_max {args.channel_max}, all blocks bf16 (fp32 accumulate), eval, noise_mode=const, random-init weights',
                            images_per_gpu_per_step=n, global_batch=n * world, parallelism=f'replicas x{world}', execution=execution),
                roofline=dict(bound='hbm', kernel='conv2d_mfma16<bf16,...> (channels-last implicit GEMM) + conv2d_up2f16<bf16,...> (the up = 2 layers in one launch), both v_mfma_f32_32x32x16_bf16; all 16-bit convolution launches of a step',
                              achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                              traffic=committed_traffic('cfg5')[0], traffic_source=committed_traffic('cfg5')[1],
                              algorithmic_bytes_per_launch=round(sum(r[3] for r in dom) / max(len(dom), 1)),
                              bytes_counted='numel(x) + numel(y) + packed weights, in bytes, per launch',
                              mfma_tflops=round(tfl, 1), mfma_frac=round(tfl / BF16_MFMA_PEAK_TFLOPS, 4), mfma_flops_counted='executed on the matrix pipe',
                              mfma_tflops_survey_count=round(tfl_survey, 1), mfma_frac_survey_count=round(tfl_survey / BF16_MFMA_PEAK_TFLOPS, 4),
                              launches_per_step=len(dom) // max(args.steps, 1),
                              measured_in='the EAGER pass of the same steps (per-launch HIP events cannot be recorded inside a replayed graph): every figure of this object, '
                                          'including conv_time_frac_of_step, is of that pass; `value` / `ms_per_step` are of the hipGraph replay' if not args.no_graph else 'the timed (eager) pass',
                              eager_ms_per_step=round(1e3 * eager_elapsed / args.steps, 3),
                              conv_time_frac_of_step=round(t_dom / eager_elapsed, 4),   # its launches' event time / the wall time of the pass they were measured in
                              top_res_gbs=round(sum(r[3] for r in top) / max(sum(r[2] for r in top), 1e-12) / 1e9, 1)))
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline_stack(args.channel_max)
    if args.conv_breakdown:
        groups = {}
        for geo, fl, tm, by in rows:
            g = groups.setdefault(geo, [0, 0.0, 0.0, 0.0])
            g[0] += 1; g[1] += tm; g[2] += fl; g[3] += by
        with open(args.conv_breakdown, 'w') as f:
            f.write('kh,kw,stride,algorithm,shape,launches_per_step,avg_us,ms_per_step,algorithmic_tflops,algorithmic_gbs\n')
            for geo, (cnt, tm, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                f.write(f'{geo[0]},{geo[1]},{geo[2]},{geo[3]},{geo[4]},{cnt / args.steps:g},{1e6 * tm / cnt:.1f},{1e3 * tm / args.steps:.3f},{fl / tm / 1e12:.1f},{by / tm / 1e9:.0f}\n')
    print(json.dumps(line), flush=True)


def run_train(args, rank, world, dev, dist):
    """BASELINE config 4 (secondary, --mode train): iterations/s of the 8-phase fullbody G+D step incl. lazy R1, batch 4 per
    GPU, flat-bucket gradient exchange over RCCL overlapped with the last backward (training/ddp.py).  VGG/contextual losses
    omitted (weights unavailable offline).  Every convolution of the step is this package's: input gradients on the forward MFMA kernels (flipped, O<->I
    transposed packs), weight gradients on csrc/conv2d_wgrad.hip (fp32 3x3 / 1x1 / transposed / the 7x7 stem; fp16 on channels-last operands); the training
    route runs bias_act in the convolution epilogues, the SPADE combine and the modulated convolutions as native launches.  `roofline` = the largest kernel of the step,
    conv2d_wgrad<3,3,1>, timed with HIP events around its launches."""
    from training import networks, replicas
    from training.loss import StyleGAN2Loss
    from training.training_step import TrainingStep
    from training import ddp
    torch.manual_seed(0)                                       # same initial weights on every rank
    G = networks.GeneratorFull_v20(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
                                   synthesis_kwargs=dict(channel_base=32768, channel_max=512, conv_clamp=256)).to(dev).train()
    dkw = dict(c_dim=512, img_resolution=512, channel_base=32768, channel_max=512, conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=4))
    dkw['num_fp16_res'] = args.d_fp16_res                      # train.py:196: fp16 for the 3 highest resolutions of both discriminators
    D = networks.Discriminator(img_channels=6, **dkw).to(dev).train()
    DP = networks.Discriminator(img_channels=10, **dkw).to(dev).train()
    ddp.broadcast_parameters([G, D, DP])
    parts = dict(G_mapping=G.mapping, G_synthesis=G.synthesis, G_const_encoding=G.const_encoding, G_style_encoding=G.style_encoding)
    loss = StyleGAN2Loss(device=dev, **parts, D=D, D_parsing=DP, style_mixing_prob=0.9, r1_gamma=10, l1_weight=50, mask_weight=1.0)
    n = args.batch if args.batch != BATCH_PER_GPU else 4       # batch_gpu 4 (global 32 on 8 GPUs, train.py:174)
    step = TrainingStep(parts, D, DP, loss, batch_size=n * world, graphs=(world == 1 and args.train_graphs))
    g = torch.Generator(device='cpu').manual_seed(100 + rank)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    batch = dict(real_img=u(n, 3, 512, 512), gen_z=torch.zeros([n, 0], device=dev), style_input=u(n, 45, 128, 128), retain=u(n, 6, 512, 512),
                 pose=u(n, 5, 512, 512), denorm_upper_input=u(n, 3, 512, 512), denorm_lower_input=u(n, 3, 512, 512),
                 denorm_upper_mask=(u(n, 1, 512, 512) > 0).float(), denorm_lower_mask=(u(n, 1, 512, 512) > 0).float(),
                 gt_parsing=torch.randint(0, 7, [n, 1, 512, 512], generator=g).float().to(dev))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    from torch_utils.ops import conv2d_mfma
    for _ in range(args.warmup):
        step.run([batch])
    barrier()
    wtl = conv2d_mfma.start_wgrad_timeline() if (rank == 0 and not step.graphs) else None      # events cannot be recorded inside a replayed graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step.run([batch])
    barrier()
    elapsed = replicas.max_over_ranks(time.perf_counter() - t0, device=dev)
    conv2d_mfma.stop_wgrad_timeline()
    roofline = None
    if wtl:
        # largest single kernel of the step: the fp32 weight gradient of the stride-1 3x3 layers (csrc/conv2d_wgrad.hip: a GEMM over pixels on the fp32
        # MFMA + its fixed-order split-K reduction, both inside the event pair); flops = the forward count of SURVEY 8d for the same layer
        dom = [(fl, e0.elapsed_time(e1) * 1e-3, by) for geo, fl, e0, e1, by in wtl if geo[:4] == (3, 3, 1, 'wgrad')]
        # round 5: the 3x3 layers with >= 64 channels take the bf16 matrix pipe (three-term operand split, six products, fp32 accumulation: conv2d_mfma.
        # _weight_gradient_bf16x3); event pairs span the two splitting passes, the 6N-image launch of conv2d16_wgrad and its reduction
        x3 = [(fl, e0.elapsed_time(e1) * 1e-3, by) for geo, fl, e0, e1, by in wtl if geo[3] == 'wgrad_bf16x3']
        allw = sum(e0.elapsed_time(e1) * 1e-3 for _, _, e0, e1, _ in wtl)
        fp32_part = x3_part = None
        if dom:
            fl, tm = sum(d[0] for d in dom), sum(d[1] for d in dom)
            fp32_part = dict(bound='mfma', kernel='conv2d_wgrad<3,3,1> + wgrad_reduce (fp32 weight gradient of the stride-1 3x3 layers, v_mfma_f32_32x32x2_f32)',
                             achieved=round(fl / tm / 1e12, 2), peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(fl / tm / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), traffic=None, traffic_note='not collected: no PMC pass was made over the training step this round',
                             launches_per_step=round(len(dom) / args.steps, 1), avg_launch_ms=round(1e3 * tm / len(dom), 4), time_frac_of_step=round(tm / elapsed, 4),
                             algorithmic_bytes_per_launch=round(sum(d[2] for d in dom) / len(dom)))
        if x3:
            fl3, tm3 = sum(d[0] for d in x3), sum(d[1] for d in x3)
            x3_part = dict(bound='mfma', kernel='conv2d16_wgrad<3,3,s,bf16> over the six products of three-term operand splits (+ pg_split3_bf16_cl x 2 and wgrad_reduce inside each event pair): '
                                                'the float32 weight gradients of the 3x3 layers with >= 64 channels, fp32 accumulation; PG_WGRAD_BF16X3=0 sends them to conv2d_wgrad<3,3,s>',
                           achieved=round(6 * fl3 / tm3 / 1e12, 1), peak=BF16_MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(6 * fl3 / tm3 / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), traffic=None, traffic_note='not collected: no PMC pass was made over the training step this round',
                           flops_counted='executed on the bf16 matrix pipe = 6 x the forward count of SURVEY 8d', fp32_equivalent_tflops=round(fl3 / tm3 / 1e12, 2),
                           fp32_equivalent_frac_of_fp32_peak=round(fl3 / tm3 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                           launches_per_step=round(len(x3) / args.steps, 1), avg_launch_ms=round(1e3 * tm3 / len(x3), 4), time_frac_of_step=round(tm3 / elapsed, 4),
                           algorithmic_bytes_per_launch=round(sum(d[2] for d in x3) / len(x3)))
        # the object describes the group the step spends more time in; the other one rides along under its own key
        if x3_part and (not fp32_part or x3_part['time_frac_of_step'] >= fp32_part['time_frac_of_step']):
            roofline = dict(x3_part, **(dict(fp32_weight_gradients=fp32_part) if fp32_part else {}))
        elif fp32_part:
            roofline = dict(fp32_part, **(dict(bf16x3_weight_gradients=x3_part) if x3_part else {}))
        if roofline:
            roofline['all_native_wgrad_time_frac_of_step'] = round(allw / elapsed, 4)
    if rank == 0:
        print(json.dumps(dict(metric='fullbody G+D training iterations/sec (8-phase step incl. lazy R1)', value=round(args.steps / elapsed, 4), unit='it/s',
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 1), higher_is_better=True,
                              scaling='weak', vs_baseline=None, dtype='f32', data='synthetic', images_per_sec=round(args.steps * n * world / elapsed, 3),
                              config=dict(workload='BASELINE config 4: fullbody G+D step, lazy R1 (gamma 10), L1 + parsing CE, no VGG; ' + (f'discriminators fp16 at their {args.d_fp16_res} highest resolutions (train.py:196)' if args.d_fp16_res else 'discriminators in fp32'),
                                          batch_per_gpu=n, global_batch=n * world, parallelism=f'dp{world}, one flat fp32 gradient bucket per phase, segment all_reduce (RCCL) launched from autograd hooks on a side stream',
                                          weight_gradients=('float32; 3x3 layers with >= 64 channels as six bf16 x bf16 products of exact three-term operand splits with fp32 accumulation (fp32-class: tests/test_hip_parity.py::test_weight_gradient_bf16x3), the rest on the fp32 MFMA'
                                                            if os.environ.get('PG_WGRAD_BF16X3', 'auto') != '0' else 'float32 on the fp32 MFMA (PG_WGRAD_BF16X3=0)'),
                                          first_batch_idx=0, note='steps start at batch_idx = warmup; reg phases fire every 4th (G) / 16th (D) iteration',
                                          execution=(f'one hipGraph replay per phase once captured (captured so far: {step.graphed_phases()}); a phase runs eagerly the first time it is due and is captured the second time' if step.graphs else 'eager launches')),
                              **(dict(roofline=roofline) if roofline else {}))), flush=True)


def run_selftest(args, rank, world):
    """Launcher / rank plumbing without a GPU (tests/test_bench_launcher.py): gloo, a stub forward, the same barrier -> timed
    steps -> barrier -> MAX over ranks -> one JSON line from rank 0 protocol as the real modes.  Not a measurement."""
    import torch.distributed as dist
    from training import replicas
    if world > 1 or 'RANK' in os.environ:
        dist.init_process_group('gloo')
    stub = torch.nn.Conv2d(3, 4, 3, padding=1)
    x = torch.randn([args.batch, 3, 16, 16], generator=torch.Generator().manual_seed(rank))

    def barrier():
        if dist.is_initialized():
            dist.barrier()
    with torch.no_grad():
        for _ in range(args.warmup):
            stub(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = stub(x)
        barrier()
    elapsed = replicas.max_over_ranks(time.perf_counter() - t0)
    ranks_seen = torch.zeros([world])
    ranks_seen[rank] = 1
    if dist.is_initialized():
        dist.all_reduce(ranks_seen)
    if rank == 0:
        print(json.dumps(dict(metric='launcher selftest (stub forward on CPU, gloo)', value=round(args.steps * args.batch * world / elapsed, 3), unit='images/s',
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True,
                              scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                              config=dict(workload='stub', images_per_gpu_per_step=args.batch, global_batch=args.batch * world, parallelism=f'replicas x{world}',
                                          ranks_seen=int(ranks_seen.sum()), out_shape=list(y.shape)))), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def under_profiler():
    """True when a rocprofv3 / roctracer tool library is preloaded into this process: children would inherit it, write into the same
    output directory and spend minutes under tracing -- the secondary runs are skipped then."""
    return any('rocprof' in os.environ.get(k, '').lower() or 'roctracer' in os.environ.get(k, '').lower()
               for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB', 'ROCPROFILER_LIBRARY_PATH'))


def secondary_runs(budget_s=240.0):
    """BASELINE configs 3, 5 and 4 as child processes of this script (fresh processes: the parent's GPU memory pool and plugin state do not
    leak into them; the parent only waits -- it never execs), a few seconds each; their JSON lines are attached to the headline
    line as `secondary`.  Failures are recorded, never raised: the headline must not depend on them."""
    import subprocess
    out = {}
    t_start = time.perf_counter()
    for key, extra in (('config3_generator', ['--mode', 'generator', '--steps', '8', '--warmup', '3']),
                       ('config3_routed', ['--mode', 'generator_routed', '--steps', '6', '--warmup', '2']),
                       ('config5_bf16_1024', ['--mode', 'bf16_1024', '--steps', '30', '--warmup', '10', '--no-cpu-baseline']),
                       ('config4_train_step', ['--mode', 'train', '--steps', '10', '--warmup', '3', '--no-cpu-baseline'])):
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20:
            out[key] = dict(error='skipped: time budget of the secondary runs used up')
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', *extra], capture_output=True, text=True, timeout=left,
                               env={k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')})
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not line:
                out[key] = dict(error=f'exit {r.returncode}: {(r.stderr or r.stdout)[-300:]}')
                continue
            j = json.loads(line[-1])
            out[key] = {k: j[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'images_per_sec', 'dtype', 'config', 'roofline', 'routing') if k in j}
        except Exception as e:                                # noqa: BLE001 -- recorded, the headline line still prints
            out[key] = dict(error=f'{type(e).__name__}: {e}'[:300])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)         # warm-up 5, time 20 iterations
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='images per GPU per step (config 2: 8)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='config 5: time eager launches instead of hipGraph replays')
    ap.add_argument('--train-graphs', action='store_true', help='config 4: replay each phase as one hipGraph (TrainingStep(graphs=True)); measured 312 vs 304 ms eager -- the step is GPU-bound')
    ap.add_argument('--d-fp16-res', type=int, default=3, help='config 4: discriminator resolutions in fp16 (train.py:196: 3; 0 = fp32)')
    ap.add_argument('--conv-breakdown', default=None, metavar='CSV', help='also write the per-shape conv launch timeline of the timed steps')
    ap.add_argument('--mode', choices=['synthesis', 'generator', 'generator_routed', 'train', 'bf16_1024', 'selftest'], default='synthesis',
                    help="'synthesis' = the headline (config 2); 'generator' = config 3 (encoders + mapping + synthesis, N=16) on synthetic tensors, 'generator_routed' = the same behind the HIP patch routing of every sample; 'train' = config 4 step; "
                         "'bf16_1024' = config 5 (StyleGAN2 stack at 1024^2 in bf16, N=4); 'selftest' = launcher / rank plumbing on CPU over gloo with a stub forward (no measurement)")
    ap.add_argument('--no-secondary', action='store_true', help='headline mode: skip the config 3 / config 5 child runs')
    ap.add_argument('--channel-max', type=int, default=1024, help="config 5: widest layer (SURVEY 8d: 'channels 1024 -> 32')")
    args = ap.parse_args()

    from training import launch
    if args.gpus > 1 and not launch.launched_as_rank():
        # Plain `python bench.py --gpus N`: this process is the launcher.  It has imported neither torch nor the plugins and never
        # touches the GPU; the N ranks are fresh children of it (reference: train.py:563-568).
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__), *sys.argv[1:]], args.gpus))

    global torch
    import torch as _torch
    torch = _torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it plainly (it launches its own ranks) or with --nproc-per-node {args.gpus}')
    if args.mode == 'selftest':
        return run_selftest(args, rank, world)
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: the product has no CPU path')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or launch.launched_as_rank():              # one rank per GPU
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)      # RCCL over xGMI

    from torch_utils import custom_ops
    custom_ops.verbosity = 'none'
    from torch_utils.ops import conv2d_mfma
    from training import networks, replicas

    if args.mode in ('train', 'generator', 'generator_routed', 'bf16_1024'):
        dict(train=run_train, generator=run_generator, generator_routed=run_generator, bf16_1024=run_stack)[args.mode](args, rank, world, dev, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    net = init_weights(networks.SynthesisNetworkFull_v18(**CFG2)).to(dev).eval()
    inp = make_inputs(args.batch, dev, seed=rank)           # inputs resident in HBM before the timed region

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            run_net(net, inp)
        barrier()
        timeline = conv2d_mfma.start_timeline()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run_net(net, inp)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        conv2d_mfma.stop_timeline()
    assert all(torch.isfinite(o).all() for o in out)

    elapsed = replicas.max_over_ranks(elapsed, device=dev)   # slowest rank defines the step time

    if rank == 0:
        images = args.batch * args.steps * world
        value = images / elapsed
        roofline = conv_roofline(timeline, elapsed, args.steps, value / world, 'cfg2', GFLOP_PER_IMAGE)
        line = dict(metric='512-res try-on images/sec (SynthesisNetwork fwd)', value=round(value, 3), unit='images/s', n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True,
                    scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                    config=dict(workload='BASELINE config 2: SynthesisNetworkFull_v18 forward, 512x512, channel_base 32768, fp32, eval, '
                                         'noise_mode=const, argmax parsing, random-init weights', images_per_gpu_per_step=args.batch,
                                global_batch=args.batch * world, parallelism=f'replicas x{world}'),
                    roofline=roofline)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        if world == 1 and not args.no_secondary and under_profiler():
            line['secondary'] = dict(skipped='a profiler tool library is preloaded into this process')
        elif world == 1 and not args.no_secondary:
            del net, inp, out                                  # the children get the whole GPU
            torch.cuda.empty_cache()
            line['secondary'] = secondary_runs()
        if args.conv_breakdown:                                # per (geometry, algorithm, shape, fused stages): launches, time, rate
            groups = {}
            for geo, fl, e0, e1, _ in timeline:
                g = groups.setdefault(geo, [0, 0.0, 0.0])
                g[0] += 1; g[1] += e0.elapsed_time(e1); g[2] += fl
            with open(args.conv_breakdown, 'w') as f:
                f.write('kh,kw,stride,algorithm,shape,launches_per_step,avg_us,ms_per_step,algorithmic_tflops\n')
                for geo, (cnt, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                    f.write(f'{geo[0]},{geo[1]},{geo[2]},{geo[3]},{geo[4]},{cnt / args.steps:g},{1e3 * ms / cnt:.1f},{ms / args.steps:.3f},{fl / ms / 1e9:.1f}\n')
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
