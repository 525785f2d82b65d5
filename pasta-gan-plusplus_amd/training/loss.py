"""Losses of the full-body training step, restated from the reference's ``StyleGAN2Loss``
(training/loss_fullbody.py:29-330): non-saturating logistic GAN loss on three discriminator views
(coarse image, fine-tuned image, predicted parsing), L1 to the real image, class-weighted cross-entropy
on the parsing head, and lazy R1 on both discriminators.

Not restated: the VGG perceptual and contextual terms (:336-618) -- their weights
(``./checkpoints/vgg19*.pth``) are not available offline (``train.sh`` runs with contextual_weight 0);
pass ``vgg_weight=0`` semantics are the only ones supported.  Path-length regularisation is commented out
in the reference (:200-221), so a ``Greg`` phase only runs the style encoder.

Gradient synchronisation is NOT done here (the reference toggles DDP's hooks through ``ddp_sync``, letting only the
LAST backward of the last accumulation round reduce): ``on_last_backward``, when set, is called right before that
backward so that ``training.ddp.GradBucket`` can arm its exchange hooks (segments of the flat bucket are then reduced
over RCCL while the rest of that backward still runs).
"""

import torch
import torch.nn.functional as F

from torch_utils.ops import conv2d_gradfix

_G_PARTS = ('G_mapping', 'G_synthesis', 'G_const_encoding', 'G_style_encoding')


class StyleGAN2Loss:
    def __init__(self, device, G_mapping, G_synthesis, G_const_encoding, G_style_encoding, D, D_parsing, augment_pipe=None,
                 style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=0, l1_weight=50, vgg_weight=0,
                 contextual_weight=0, mask_weight=1.0, report=None):
        if vgg_weight or contextual_weight:
            raise NotImplementedError('VGG / contextual terms need checkpoints that are not available offline')
        self.device = device
        self.G_mapping, self.G_synthesis, self.G_const_encoding, self.G_style_encoding = G_mapping, G_synthesis, G_const_encoding, G_style_encoding
        self.D, self.D_parsing, self.augment_pipe = D, D_parsing, augment_pipe
        self.style_mixing_prob, self.r1_gamma, self.pl_weight = style_mixing_prob, r1_gamma, pl_weight
        self.l1_weight, self.mask_weight = l1_weight, mask_weight
        self.class_weight = torch.tensor([1, 3, 4, 4, 4, 4, 4], dtype=torch.float32, device=device)   # loss_fullbody.py:53
        self.report = report or (lambda name, value: None)
        self.on_last_backward = None          # callable; fired once per accumulate_gradients(sync=True), before its last backward

    def phase_is_empty(self, phase):
        """True for a phase that runs no backward whatever the data (the same answer on every rank): 'Greg', whose only term --
        path-length regularisation -- is commented out in the reference (loss_fullbody.py:200-221); an R1 phase with gamma 0."""
        return phase == 'Greg' or (phase in ('Dreg', 'D_parsingreg') and self.r1_gamma == 0)

    # ------------------------------------------------------------------ forward helpers (loss_fullbody.py:75-114)
    def run_G(self, z, c, pose, const_feats, denorm_upper_mask, denorm_lower_mask, denorm_upper_input, denorm_lower_input, gt_parsing):
        cat_feats = {str(f.shape[2]): f for f in const_feats}
        pose_feat = self.G_const_encoding(pose)
        ws = self.G_mapping(z, c)
        if self.style_mixing_prob > 0:
            cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
            cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
            # ws[:, cutoff:] = mapping(...)[:, cutoff:] (loss_fullbody.py:90) without reading `cutoff` on the host (a 0-dim tensor used as a slice
            # bound synchronises, and cannot be captured into a hipGraph): the same selection as a mask
            mixed = self.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)
            ws = torch.where((torch.arange(ws.shape[1], device=ws.device) >= cutoff)[None, :, None], mixed, ws)
        img, finetune_img, pred_parsing = self.G_synthesis(ws, pose_feat, cat_feats, denorm_upper_input, denorm_lower_input,
                                                           denorm_upper_mask, denorm_lower_mask, gt_parsing)
        return img, finetune_img, pred_parsing, ws

    def run_D(self, img, pose, c):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        return self.D(torch.cat([img, pose[:, 0:3]], dim=1), c)

    def run_D_parsing(self, parsing, pose, c):
        return self.D_parsing(torch.cat([parsing, pose[:, 0:3]], dim=1), c)

    # ------------------------------------------------------------------ one phase, one accumulation round
    def accumulate_gradients(self, phase, real_img, gen_z, style_input, retain, pose, denorm_upper_input, denorm_lower_input,
                             denorm_upper_mask, denorm_lower_mask, gt_parsing, sync=True, gain=1):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth', 'D_parsingmain', 'D_parsingreg', 'D_parsingboth']
        do_Gmain = phase in ('Gmain', 'Gboth')
        do_Dmain = phase in ('Dmain', 'Dboth')
        do_Dr1 = phase in ('Dreg', 'Dboth') and self.r1_gamma != 0
        do_DPmain = phase in ('D_parsingmain', 'D_parsingboth')
        do_DPr1 = phase in ('D_parsingreg', 'D_parsingboth') and self.r1_gamma != 0

        # the backward calls of this round, in program order; with sync=True the last one is announced (loss_fullbody.py
        # passes sync=False to every earlier one: "gets synced by loss_Dreal")
        slots = [do_Gmain, do_Dmain, do_Dmain or do_Dr1, do_DPmain, do_DPmain or do_DPr1]
        last_slot = max([i for i, on in enumerate(slots) if on], default=-1)

        def before_backward(slot):
            if sync and slot == last_slot and self.on_last_backward is not None:
                self.on_last_backward()

        real_c, cat_feats = self.G_style_encoding(style_input, retain)
        gen_c = real_c                                               # the style code conditions both G and D (loss_fullbody.py:129)
        g_args = (gen_z, gen_c, pose, cat_feats, denorm_upper_mask, denorm_lower_mask, denorm_upper_input, denorm_lower_input, gt_parsing)
        nonsat = lambda logits: F.softplus(-logits)                  # -log(sigmoid(x))
        sat = lambda logits: F.softplus(logits)                      # -log(1 - sigmoid(x))

        if do_Gmain:                                                 # G: make all three discriminator views say "real"
            gen_img, gen_fine, pred_parsing, _ = self.run_G(*g_args)
            parsing_prob = torch.softmax(pred_parsing, dim=1)
            adv = (nonsat(self.run_D(gen_img, pose, gen_c)).mean() + nonsat(self.run_D(gen_fine, pose, gen_c)).mean()) / 2
            adv_parsing = nonsat(self.run_D_parsing(parsing_prob, pose, gen_c)).mean()
            l1 = 0
            if self.l1_weight > 0:
                l1 = ((gen_img - real_img).abs().mean() + (gen_fine - real_img).abs().mean()) / 2 * self.l1_weight
            ce = 0
            if self.mask_weight > 0:
                ce = F.cross_entropy(pred_parsing, gt_parsing.long()[:, 0], weight=self.class_weight, ignore_index=255) * self.mask_weight
            loss_G = adv + l1 + ce + adv_parsing
            self.report('Loss/G/loss', adv)
            self.report('Loss/G/L1', l1)
            self.report('Loss/G/mask_loss', ce)
            self.report('Loss/G/loss_parsing', adv_parsing)
            before_backward(0)
            loss_G.mul(gain).backward()

        loss_Dgen_fine = 0
        if do_Dmain:                                                 # D on generated images
            gen_img, gen_fine, _, _ = self.run_G(*g_args)
            loss_Dgen = sat(self.run_D(gen_img, pose, gen_c))
            loss_Dgen_fine = sat(self.run_D(gen_fine, pose, gen_c))
            before_backward(1)
            ((loss_Dgen.mean() + loss_Dgen_fine.mean()) / 2).mul(gain).backward()

        if do_Dmain or do_Dr1:                                       # D on real images (+ lazy R1)
            real_tmp = real_img.detach().requires_grad_(do_Dr1)
            real_logits = self.run_D(real_tmp, pose, real_c)
            self._real_and_r1(real_logits, real_tmp, do_Dmain, do_Dr1, gain, 'D', lambda: before_backward(2))

        loss_DPgen = 0
        if do_DPmain:                                                # parsing discriminator on the predicted parsing
            _, _, pred_parsing, _ = self.run_G(*g_args)
            loss_DPgen = sat(self.run_D_parsing(torch.softmax(pred_parsing, dim=1), pose, gen_c))
            before_backward(3)
            loss_DPgen.mean().mul(gain).backward()

        if do_DPmain or do_DPr1:                                     # ... and on the one-hot ground-truth parsing (+ lazy R1)
            onehot = torch.cat([(gt_parsing == k).to(gt_parsing.dtype) for k in range(7)], dim=1).detach().requires_grad_(do_DPr1)
            real_logits = self.run_D_parsing(onehot, pose, real_c)
            self._real_and_r1(real_logits, onehot, do_DPmain, do_DPr1, gain, 'D_parsing', lambda: before_backward(4))

    def _real_and_r1(self, real_logits, real_input, do_main, do_r1, gain, tag, announce=lambda: None):
        loss_real = F.softplus(-real_logits) if do_main else 0
        loss_r1 = 0
        if do_r1:                                                    # R1: gamma/2 * |d logits / d input|^2, without weight gradients of the inner pass
            with conv2d_gradfix.no_weight_gradients():
                r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_input], create_graph=True, only_inputs=True)
            penalty = r1_grads.square().sum([1, 2, 3])
            loss_r1 = penalty * (self.r1_gamma / 2)
            self.report(f'Loss/{tag}/r1_penalty', penalty)
        self.report(f'Loss/{tag}/real', loss_real)
        announce()
        (real_logits * 0 + loss_real + loss_r1).mean().mul(gain).backward()
