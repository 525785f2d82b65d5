"""CPU restatement of the PASTA-GAN++ synthesis stack (TEST INFRASTRUCTURE ONLY).

Follows training/networks.py of the reference for structure and arithmetic
order; module/parameter names equal the reference's so a reference
``state_dict`` loads here (that is how ``tests/golden/make_golden.py`` pins this
file against the real classes).  All tensor maths goes through
``oracle.ops_ref`` -- never through the product.

``SynthesisLayer`` is NOT in the reference tree: it is
used at networks.py:2006-2011, 2121-2126 but defined nowhere.  The class below
is the build's own statement of the layer contract recovered from the call
sites (ctor kwargs networks.py:2006-2011, forward kwargs :2054-2062, parameter
names legacy.py:178-195): affine -> modulated_conv2d(+noise) -> bias_act.
"""

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops_ref as R

SQRT_HALF = math.sqrt(0.5)


class FullyConnectedLayer(nn.Module):
    """networks.py:99-128 (equalised-LR dense layer)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.to(x.dtype) * self.weight_gain
        y = x @ w.t()
        b = None if self.bias is None else self.bias.to(x.dtype) * self.bias_gain
        if self.activation == 'linear':
            return y if b is None else y + b[None]
        return R.bias_act(y, b, act=self.activation)


class Conv2dLayer(nn.Module):
    """networks.py:133-179: conv2d_resample -> bias_act."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = R.ACTIVATIONS[activation][1]
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels])) if bias else None

    def forward(self, x, gain=1):
        w = self.weight * self.weight_gain
        x = R.conv2d_resample(x, w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                              padding=self.padding, flip_weight=(self.up == 1))
        clamp = None if self.conv_clamp is None else self.conv_clamp * gain
        b = None if self.bias is None else self.bias.to(x.dtype)
        return R.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class ResBlock(nn.Module):
    """networks.py:287-316 (kernel_size argument is ignored by the reference too)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        kw = dict(resample_filter=resample_filter, conv_clamp=conv_clamp)
        self.conv0 = Conv2dLayer(in_channels, out_channels, 3, activation=activation, up=up, down=down, bias=bias, **kw)
        self.conv1 = Conv2dLayer(out_channels, out_channels, 3, activation=activation, bias=bias, **kw)
        self.skip = Conv2dLayer(in_channels, out_channels, 1, bias=False, up=up, down=down, **kw)

    def forward(self, x):
        y = self.skip(x, gain=SQRT_HALF)
        x = self.conv0(x)
        x = self.conv1(x, gain=SQRT_HALF)
        return y + x


class Spade_Conv2dLayer(nn.Module):
    """networks.py:1586-1635: bias_act FIRST (unless no_act), then the conv."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='relu', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = R.ACTIVATIONS[activation][1]
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels])) if bias else None

    def forward(self, x, gain=1, no_act=False):
        w = self.weight * self.weight_gain
        if not no_act:
            clamp = None if self.conv_clamp is None else self.conv_clamp * gain
            b = None if self.bias is None else self.bias.to(x.dtype)
            x = R.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)
        return R.conv2d_resample(x, w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                 padding=self.padding, flip_weight=(self.up == 1))


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False) as used at networks.py:1713: biased variance."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = (x - mean).square().mean(dim=(2, 3), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


class Spade_Norm_Block(nn.Module):
    """networks.py:1702-1723: norm(x) * (1 + gamma(feat)) + beta(feat)."""

    def __init__(self, in_channels, norm_channels):
        super().__init__()
        self.conv_mlp = Spade_Conv2dLayer(in_channels, norm_channels, 3, bias=False)
        self.conv_gamma = Spade_Conv2dLayer(norm_channels, norm_channels, 3, bias=False)
        self.conv_beta = Spade_Conv2dLayer(norm_channels, norm_channels, 3, bias=False)

    def forward(self, x, denorm_feats):
        actv = torch.relu(self.conv_mlp(denorm_feats, no_act=True))
        gamma = self.conv_gamma(actv, no_act=True)
        beta = self.conv_beta(actv, no_act=True)
        return instance_norm(x) * (1 + gamma) + beta


class Spade_ResBlockV4_512(nn.Module):
    """networks.py:1859-1904."""

    def __init__(self, in_channels, out_channels, spade_channels, resample_filter=(1, 3, 3, 1), conv_clamp=None):
        super().__init__()
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        kw = dict(bias=False, resample_filter=resample_filter, conv_clamp=conv_clamp)
        self.conv = Spade_Conv2dLayer(in_channels, in_channels, 3, **kw)
        self.conv0 = Spade_Conv2dLayer(in_channels, out_channels, 3, **kw)
        self.conv1 = Spade_Conv2dLayer(out_channels, out_channels, 3, **kw)
        self.skip = Spade_Conv2dLayer(in_channels, out_channels, 1, **kw)
        self.spade_skip = Spade_Norm_Block(spade_channels, in_channels)
        self.spade0 = Spade_Norm_Block(spade_channels, in_channels)
        self.spade1 = Spade_Norm_Block(spade_channels, out_channels)

    def forward(self, x, denorm_feat):
        x = self.conv(x, no_act=True)
        y = self.skip(self.spade_skip(x, denorm_feat), gain=SQRT_HALF)
        x = self.conv0(self.spade0(x, denorm_feat))
        x = self.conv1(self.spade1(x, denorm_feat), gain=SQRT_HALF)
        return y + x


class SynthesisLayer(nn.Module):
    """The build's own statement of the missing layer (see module docstring)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation='lrelu', resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False):
        super().__init__()
        self.resolution, self.up, self.use_noise, self.activation, self.conv_clamp = resolution, up, use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.act_gain = R.ACTIVATIONS[activation][1]
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1):
        assert noise_mode in ('random', 'const', 'none')
        styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution]) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = R.modulated_conv2d(x, self.weight, styles, noise=noise, up=self.up, padding=self.padding,
                               resample_filter=self.resample_filter, flip_weight=(self.up == 1), fused_modconv=fused_modconv)
        clamp = None if self.conv_clamp is None else self.conv_clamp * gain
        return R.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class ToRGBLayerFull(nn.Module):
    """networks.py:1941-1967 (v1_v5: 7 parsing classes) / :1910-1936 (v1_v4: 6)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, is_last=False, is_style=False, parsing_channels=7):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.has_parsing = is_last and is_style
        if self.has_parsing:
            self.m_weight1 = nn.Parameter(torch.randn([parsing_channels, in_channels, kernel_size, kernel_size]))
            self.m_bias1 = nn.Parameter(torch.zeros([parsing_channels]))

    def forward(self, x, w, fused_modconv=True):
        styles = self.affine(w) * self.weight_gain
        pred_parsing = None
        if self.has_parsing:
            pred_parsing = R.modulated_conv2d(x, self.m_weight1, styles, demodulate=False, fused_modconv=fused_modconv)
            pred_parsing = R.bias_act(pred_parsing, self.m_bias1.to(x.dtype), clamp=self.conv_clamp)
        y = R.modulated_conv2d(x, self.weight, styles, demodulate=False, fused_modconv=fused_modconv)
        return R.bias_act(y, self.bias.to(x.dtype), clamp=self.conv_clamp), pred_parsing


class SynthesisBlockFull(nn.Module):
    """networks.py:2086-2194 (v1_v6, style branch) and :1971-2082 (v1_v4, texture
    branch: adds spade_b512 conditioned on the 1-channel parsing map)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, is_style=False,
                 texture=False, resample_filter=(1, 3, 3, 1), conv_clamp=None, **layer_kwargs):
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.texture = is_last, texture
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = nn.Parameter(torch.randn([out_channels, resolution, resolution]))  # unused (networks.py:2157-2161)
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution,
                                    conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        self.torgb = ToRGBLayerFull(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, is_last=is_last,
                                    is_style=is_style, parsing_channels=(6 if texture else 7))
        self.num_torgb += 1
        if resolution > 32:
            self.merge_conv = Conv2dLayer(out_channels + 64, out_channels, kernel_size=1, resample_filter=resample_filter)
        if texture:
            self.spade_b512 = Spade_ResBlockV4_512(out_channels, out_channels, spade_channels=1)

    def forward(self, x, img, ws, pose_feature, cat_feat, parsing=None, fused_modconv=None, **layer_kwargs):
        assert ws.shape[1] == self.num_conv + self.num_torgb
        w_iter = iter(ws.unbind(dim=1))
        if fused_modconv is None:
            fused_modconv = not self.training  # fp32 path of networks.py:2152-2154
        if self.in_channels == 0:
            x = pose_feature.to(torch.float32)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        else:
            x = x.to(torch.float32)
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            if x.shape[2] > 32:
                x = torch.cat([x, cat_feat[str(x.shape[2])].to(torch.float32)], dim=1)
                x = self.merge_conv(x)
            if self.texture:
                x = self.spade_b512(x, parsing)
        if img is not None:
            img = R.upsample2d(img, self.resample_filter)
        y, pred_parsing = self.torgb(x, next(w_iter), fused_modconv=fused_modconv)
        img = y if img is None else img + y
        return x, img, pred_parsing


class SynthesisStack(nn.Module):
    """float32 statement of the plain StyleGAN2 block stack of BASELINE config 5 (SynthesisLayer x2 +
    ToRGB + skip-image upsample per resolution, no SPADE): the reference's style-branch block (networks.py:2147-2194)
    without the pose / garment-feature inputs, learned constant input, at any power-of-two resolution.  The reference class
    is hard-wired to 512^2, so this is the build's own composition of the pinned layers above."""

    def __init__(self, w_dim, img_resolution, img_channels=3, channel_base=32768, channel_max=512, conv_clamp=None, **_):
        super().__init__()
        log2 = int(math.log2(img_resolution))
        self.block_resolutions = [1 << e for e in range(3, log2 + 1)]
        width = lambda res: min(channel_base // res, channel_max)
        self.num_ws = 0
        for res in self.block_resolutions:
            blk = nn.Module()
            blk.register_buffer('resample_filter', R.setup_filter([1, 3, 3, 1]))
            if res == 8:
                blk.const = nn.Parameter(torch.randn([width(res), res, res]))
            else:
                blk.conv0 = SynthesisLayer(width(res // 2), width(res), w_dim=w_dim, resolution=res, up=2, conv_clamp=conv_clamp)
            blk.conv1 = SynthesisLayer(width(res), width(res), w_dim=w_dim, resolution=res, conv_clamp=conv_clamp)
            blk.torgb = ToRGBLayerFull(width(res), img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
            blk.num_conv = 1 if res == 8 else 2
            setattr(self, f'b{res}', blk)
            self.num_ws += blk.num_conv + (1 if res == img_resolution else 0)

    def forward(self, ws, noise_mode='const'):
        x = img = None
        start = 0
        for res in self.block_resolutions:
            blk = getattr(self, f'b{res}')
            if res == 8:
                x = blk.const[None].expand(ws.shape[0], -1, -1, -1)
                x = blk.conv1(x, ws[:, start], noise_mode=noise_mode)
            else:
                x = blk.conv0(x, ws[:, start], noise_mode=noise_mode)
                x = blk.conv1(x, ws[:, start + 1], noise_mode=noise_mode)
            y, _ = blk.torgb(x, ws[:, start + blk.num_conv])
            img = y if img is None else R.upsample2d(img, blk.resample_filter) + y
            start += blk.num_conv
        return img


def nearest_half(x):
    """F.interpolate(x, scale_factor=0.5) in its default 'nearest' mode (networks.py:2255-2256)."""
    return x[:, :, ::2, ::2]


class SynthesisNetworkFull_v18(nn.Module):
    """networks.py:2198-2327."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, **block_kwargs):
        super().__init__()
        assert img_resolution >= 8 and img_resolution & (img_resolution - 1) == 0
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.block_resolutions = [2 ** i for i in range(3, int(math.log2(img_resolution)) + 1)]
        ch = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            is_last = res == img_resolution
            block = SynthesisBlockFull(ch[res // 2] if res > 8 else 0, ch[res], w_dim=w_dim, resolution=res,
                                       img_channels=img_channels, is_last=is_last, is_style=True, **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if is_last else 0)
            setattr(self, f'b{res}', block)
        r2, r1 = self.block_resolutions[-2], self.block_resolutions[-1]
        self.spade_b256_1 = Spade_ResBlockV4_512(ch[r2], ch[r2], spade_channels=128)
        self.spade_b256_2 = Spade_ResBlockV4_512(ch[r2], ch[r2], spade_channels=128)
        self.texture_b512 = SynthesisBlockFull(ch[r1 // 2], ch[r1], w_dim=w_dim, resolution=r1, img_channels=img_channels,
                                               is_last=True, is_style=False, texture=True, **block_kwargs)
        ngf = 64
        self.spade_encoder = nn.Sequential(
            Conv2dLayer(3, ngf, kernel_size=7, activation='relu'),
            ResBlock(ngf, ngf, activation='relu'),
            ResBlock(ngf, ngf * 2, activation='relu', down=2),
        )

    def get_spade_feat(self, mask_512, denorm_mask, denorm_input):
        """networks.py:2253-2276: encode the warped garment, fill the uncovered part of
        the predicted region with the masked mean feature."""
        mask_512 = (mask_512 > 0.9).to(mask_512.dtype)
        mask_256 = (nearest_half(mask_512) > 0.9).to(mask_512.dtype)
        denorm_mask_256 = (nearest_half(denorm_mask) > 0.9).to(mask_512.dtype)
        valid = ((mask_256 + denorm_mask_256) == 2.0).to(mask_512.dtype)
        rest = mask_256 - valid
        feat = self.spade_encoder(denorm_input * mask_512 - (1 - mask_512))
        feat_sum = (feat * valid).sum(dim=(2, 3), keepdim=True)
        cnt = valid.sum(dim=(2, 3), keepdim=True)
        ok = (cnt > 10).to(mask_512.dtype)
        cnt = cnt * ok + (256 * 256) * (1 - ok)
        return feat * (1 - rest) + (feat_sum / cnt) * rest

    def forward(self, ws, pose_feat, cat_feat, denorm_upper_input, denorm_lower_input, denorm_upper_mask,
                denorm_lower_mask, gt_parsing=None, **block_kwargs):
        assert ws.shape[1] == self.num_ws
        ws = ws.to(torch.float32)
        block_ws, idx = [], 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            block_ws.append(ws.narrow(1, idx, block.num_conv + block.num_torgb))
            idx += block.num_conv
        x = img = None
        for res, cur in zip(self.block_resolutions, block_ws):
            x, img, pred_parsing = getattr(self, f'b{res}')(x, img, cur, pose_feat, cat_feat, **block_kwargs)
            if res == 256:
                x_256, img_256 = x.clone(), img.clone()
        if gt_parsing is not None:
            parsing_index = gt_parsing
        else:
            parsing_index = torch.argmax(torch.softmax(pred_parsing.detach(), dim=1), dim=1)[:, None].float()
        upper = (parsing_index == 1).float() + (parsing_index == 4).float()
        lower = (parsing_index == 2).float() + (parsing_index == 3).float()
        feat_u = self.get_spade_feat(upper, denorm_upper_mask, denorm_upper_input)
        feat_l = self.get_spade_feat(lower, denorm_lower_mask, denorm_lower_input)
        upper_256 = (nearest_half(upper) > 0.9).to(upper.dtype)
        lower_256 = (nearest_half(lower) > 0.9).to(upper.dtype)
        spade_feat = feat_u * upper_256 + feat_l * lower_256
        y = self.spade_b256_1(x_256, spade_feat)
        y = self.spade_b256_2(y, spade_feat)
        _, finetune_img, _ = self.texture_b512(y, img_256, block_ws[-1], pose_feat, cat_feat, parsing=parsing_index, **block_kwargs)
        return img, finetune_img, pred_parsing


# ==========================================================================
# "next" row f1: what runs immediately upstream of synthesis in test.py:151-153.


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """networks.py:31-33."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class MappingNetwork(nn.Module):
    """networks.py:184-259 (z and/or c -> embed/normalise -> FC stack -> broadcast to num_ws -> truncation)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed_features = (w_dim if embed_features is None else embed_features) if c_dim > 0 else 0
        layer_features = w_dim if layer_features is None else layer_features
        feats = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(feats[i], feats[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = y if x is None else torch.cat([x, y], dim=1)
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class ConstEncoderNetwork(nn.Module):
    """networks.py:357-375: 1x1 stem then stride-2 3x3 Conv2dLayers down to the 8x8 pose feature."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4):
        super().__init__()
        mult_ins, mult_outs = [1, 2, 4, 4, 4, 8], [2, 4, 4, 4, 8, 8]
        layers = [Conv2dLayer(input_nc, ngf, kernel_size=1)]
        for i in range(n_downsampling):
            layers.append(Conv2dLayer(ngf * mult_ins[i], ngf * mult_outs[i], kernel_size=3, down=2))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class Dense(nn.Module):
    """networks.py:391-408: per-pixel Linear -> InstanceNorm2d -> LeakyReLU(0.01)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels)

    def forward(self, x):
        y = F.linear(x.permute(0, 2, 3, 1), self.linear.weight, self.linear.bias).permute(0, 3, 1, 2)
        return F.leaky_relu(instance_norm(y), 0.01)


class StyleEncoderNetworkV18(nn.Module):
    """networks.py:1727-1774: garment-part encoder -> style vector, plus the 4-scale 64-channel feature pyramid of
    the retained-region image that the synthesis blocks concatenate."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=4):
        super().__init__()
        enc = [Conv2dLayer(input_nc, ngf, kernel_size=1)]
        for mi, mo in zip([1, 2, 4], [2, 4, 8]):
            enc += [Dense(ngf * mi, ngf * mi), Conv2dLayer(ngf * mi, ngf * mo, kernel_size=3, down=2)]
        for _ in range(3):
            enc += [Dense(ngf * 8, ngf * 8), Conv2dLayer(ngf * 8, ngf * 8, kernel_size=3)]
        enc += [nn.AdaptiveAvgPool2d(1)]
        self.model = nn.Sequential(*enc)
        self.fc = FullyConnectedLayer(output_nc, output_nc)
        feat = [Conv2dLayer(6, ngf, kernel_size=3)]
        for _ in range(3):
            feat.append(Conv2dLayer(ngf, ngf, kernel_size=3, down=2))
        self.feat_enc = nn.Sequential(*feat)

    def forward(self, x, const_input):
        feats = []
        for m in self.feat_enc:
            const_input = m(const_input)
            feats.append(const_input)
        x = self.model(x)
        return self.fc(x.view(x.size(0), -1)), feats


class GeneratorFull_v20(nn.Module):
    """networks.py:2330-2366."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.synthesis = SynthesisNetworkFull_v18(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)
        self.const_encoding = ConstEncoderNetwork(input_nc=3 + 2, output_nc=512, ngf=64, n_downsampling=6)
        self.style_encoding = StyleEncoderNetworkV18(input_nc=(10 * 3 + 5 * 3), output_nc=512, ngf=64, n_downsampling=6)

    def forward(self, z, c, retain, pose, denorm_upper_input, denorm_lower_input, denorm_upper_mask, denorm_lower_mask,
                gt_parsing=None, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        pose_feat = self.const_encoding(pose)
        stylecode, feats = self.style_encoding(c, retain)
        ws = self.mapping(z, stylecode, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        cat_feats = {str(f.shape[2]): f for f in feats}
        return self.synthesis(ws, pose_feat, cat_feats, denorm_upper_input, denorm_lower_input, denorm_upper_mask,
                              denorm_lower_mask, gt_parsing, **synthesis_kwargs)


# ==========================================================================
# "next" row f2: the discriminators of the training step (networks.py:444-666).


class DiscriminatorBlock(nn.Module):
    """networks.py:444-523 ('resnet' architecture; fp32)."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, architecture='resnet',
                 activation='lrelu', resample_filter=(1, 3, 3, 1), conv_clamp=None, **_unused):
        super().__init__()
        assert architecture == 'resnet' and in_channels in (0, tmp_channels)
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.register_buffer('resample_filter', R.setup_filter(list(resample_filter)))
        self.num_layers = 0
        if in_channels == 0:
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation, conv_clamp=conv_clamp)
            self.num_layers += 1
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2, resample_filter=resample_filter, conv_clamp=conv_clamp)
        self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, resample_filter=resample_filter)
        self.num_layers += 3

    def forward(self, x, img):
        if self.in_channels == 0:
            y = self.fromrgb(img.to(torch.float32))
            x = y if x is None else x + y
            img = None
        y = self.skip(x, gain=SQRT_HALF)
        x = self.conv0(x)
        x = self.conv1(x, gain=SQRT_HALF)
        return y + x, img


class MinibatchStdLayer(nn.Module):
    """networks.py:528-549."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        n, c, h, w = x.shape
        g = min(self.group_size, n) if self.group_size is not None else n
        f = self.num_channels
        y = x.reshape(g, -1, f, c // f, h, w)
        y = y - y.mean(dim=0)
        y = (y.square().mean(dim=0) + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1).repeat(g, 1, h, w)
        return torch.cat([x, y], dim=1)


class DiscriminatorEpilogue(nn.Module):
    """networks.py:554-607 ('resnet')."""

    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4, mbstd_num_channels=1,
                 activation='lrelu', conv_clamp=None):
        super().__init__()
        self.cmap_dim = cmap_dim
        self.mbstd = MinibatchStdLayer(mbstd_group_size, mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * resolution ** 2, in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap):
        x = x.to(torch.float32)
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / math.sqrt(self.cmap_dim))
        return x


class Discriminator(nn.Module):
    """networks.py:612-666."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=0,
                 conv_clamp=None, cmap_dim=None, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}):
        super().__init__()
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.block_resolutions = [2 ** i for i in range(int(math.log2(img_resolution)), 2, -1)]
        ch = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        if cmap_dim is None:
            cmap_dim = ch[4]
        if c_dim == 0:
            cmap_dim = 0
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        idx = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(ch[res] if res < img_resolution else 0, ch[res], ch[res // 2], resolution=res, first_layer_idx=idx, **block_kwargs, **common)
            setattr(self, f'b{res}', block)
            idx += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(ch[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **common)

    def forward(self, img, c, **_):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img)
        cmap = self.mapping(None, c) if self.c_dim > 0 else None
        return self.b4(x, img, cmap)
