"""Where does the config-4 training iteration spend its GPU time, by operator and operand shape?  (dev tool, GPU box)

    python tools/train_attrib.py [--top 70] [--steps 2]

Runs the bench's training step (bench.run_train's set-up) under torch.profiler and prints (a) device time per phase, (b) the operators
(aten ops and this package's autograd Functions) ranked by self device time, grouped by input shapes.  Output goes to stdout; redirect it
under gpurun_out/.
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
sys.path.insert(0, ROOT)

import torch                                                      # noqa: E402


def build(dev, n=4, d_fp16_res=3):
    from training import networks
    from training.loss import StyleGAN2Loss
    from training.training_step import TrainingStep
    torch.manual_seed(0)
    G = networks.GeneratorFull_v20(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
                                   synthesis_kwargs=dict(channel_base=32768, channel_max=512, conv_clamp=256)).to(dev).train()
    dkw = dict(c_dim=512, img_resolution=512, channel_base=32768, channel_max=512, conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=4),
               num_fp16_res=d_fp16_res)
    D = networks.Discriminator(img_channels=6, **dkw).to(dev).train()
    DP = networks.Discriminator(img_channels=10, **dkw).to(dev).train()
    parts = dict(G_mapping=G.mapping, G_synthesis=G.synthesis, G_const_encoding=G.const_encoding, G_style_encoding=G.style_encoding)
    loss = StyleGAN2Loss(device=dev, **parts, D=D, D_parsing=DP, style_mixing_prob=0.9, r1_gamma=10, l1_weight=50, mask_weight=1.0)
    step = TrainingStep(parts, D, DP, loss, batch_size=n)
    g = torch.Generator(device='cpu').manual_seed(100)
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
    batch = dict(real_img=u(n, 3, 512, 512), gen_z=torch.zeros([n, 0], device=dev), style_input=u(n, 45, 128, 128), retain=u(n, 6, 512, 512),
                 pose=u(n, 5, 512, 512), denorm_upper_input=u(n, 3, 512, 512), denorm_lower_input=u(n, 3, 512, 512),
                 denorm_upper_mask=(u(n, 1, 512, 512) > 0).float(), denorm_lower_mask=(u(n, 1, 512, 512) > 0).float(),
                 gt_parsing=torch.randint(0, 7, [n, 1, 512, 512], generator=g).float().to(dev))
    return step, batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--top', type=int, default=70)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument("--d-fp16-res", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    step, batch = build(dev, d_fp16_res=args.d_fp16_res)
    for _ in range(2):
        step.run([batch])
    torch.cuda.synchronize()

    # (a) per phase, with events
    marks = []
    def observer(event, ph):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((event, ph.name, e))
    step.observer = observer
    for _ in range(args.steps):
        step.run([batch])
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    step.observer = None
    per = collections.OrderedDict()
    begins = [(n, e) for ev, n, e in marks if ev == 'begin'] + [('end', end)]
    for (n, e0), (_, e1) in zip(begins[:-1], begins[1:]):
        per.setdefault(n, []).append(e0.elapsed_time(e1))
    print('phase (begin -> next begin), ms per occurrence:')
    for n, v in per.items():
        print(f'  {n:16s} {sum(v) / len(v):8.2f}  x{len(v) / args.steps:.1f} per iteration')

    # (b) by operator and shape
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(args.steps):
            step.run([batch])
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, 'self_device_time_total', None)
        if t is None:
            t = e.self_cuda_time_total
        if t > 0:
            rows.append((t / args.steps / 1e3, e.count / args.steps, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f'\nself device time by operator and input shapes, ms per iteration (total {total:.1f}):')
    for t, c, k, s in rows[:args.top]:
        print(f'  {t:8.3f} {c:7.1f}  {k[:48]:48s} {s}')
    by_op = collections.Counter()
    for t, c, k, s in rows:
        by_op[k] += t
    print('\nby operator:')
    for k, t in by_op.most_common(45):
        print(f'  {t:8.3f}  {k[:90]}')


if __name__ == '__main__':
    main()
