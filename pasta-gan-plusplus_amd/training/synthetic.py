"""Deterministic, name-keyed synthetic tensors: the weights and inputs of `bench.py` (random-init networks of the named architecture,
no checkpoints offline), shared with the golden-vector generator and
the tests (so weights and inputs need not be stored in the fixtures).

Uses NumPy's legacy ``RandomState`` (stream frozen by NumPy's compatibility
policy) seeded with crc32(name).
"""

import zlib

import numpy as np
import torch


def det_array(name, shape, kind='normal', scale=1.0):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if kind == 'normal':
        a = rs.standard_normal(shape)
    elif kind == 'uniform':       # U(-1, 1)
        a = rs.uniform(-1.0, 1.0, shape)
    elif kind == 'mask':          # {0, 1}
        a = (rs.uniform(0.0, 1.0, shape) > 0.5).astype(np.float64)
    elif kind == 'blockmask':     # {0,1} constant over 16x16 blocks (a plausible garment mask)
        h, w = shape[-2], shape[-1]
        coarse = rs.uniform(0.0, 1.0, shape[:-2] + ((h + 15) // 16, (w + 15) // 16)) > 0.5
        a = np.repeat(np.repeat(coarse, 16, axis=-2), 16, axis=-1)[..., :h, :w].astype(np.float64)
    elif kind == 'labels7':       # integer labels 0..6 constant over 32x32 blocks
        h, w = shape[-2], shape[-1]
        coarse = rs.randint(0, 7, shape[:-2] + ((h + 31) // 32, (w + 31) // 32))
        a = np.repeat(np.repeat(coarse, 32, axis=-2), 32, axis=-1)[..., :h, :w].astype(np.float64)
    else:
        raise KeyError(kind)
    return (a * scale).astype(np.float32)


def det_tensor(name, shape, kind='normal', scale=1.0, dtype=torch.float32):
    return torch.from_numpy(det_array(name, shape, kind, scale)).to(dtype)


def fill_module_(module, prefix='', noise_strength=0.1, bias_scale=0.1):
    """Overwrite every parameter / noise buffer of `module` with name-keyed values.

    weights ~ N(0,1); biases ~ 0.1*N(0,1) (non-zero so bias paths are exercised);
    affine.bias = 1 + 0.1*N(0,1); noise_strength = `noise_strength`;
    noise_const ~ N(0,1).  FIR buffers (`resample_filter`) are left alone.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            full = prefix + name
            if name.endswith('noise_strength'):
                p.fill_(noise_strength)
            elif name.endswith('affine.bias'):
                p.copy_(1.0 + det_tensor(full, p.shape, scale=0.1))
            elif name.endswith('bias') or name.endswith('m_bias1'):
                p.copy_(det_tensor(full, p.shape, scale=bias_scale))
            else:
                p.copy_(det_tensor(full, p.shape))
        for name, b in module.named_buffers():
            if name.endswith('noise_const'):
                b.copy_(det_tensor(prefix + name, b.shape))
    return module


def synthesis_inputs(n, w_dim=512, num_ws=14, feat_ch=512, seed_tag='cfg2', labels=True):
    """Synthetic inputs of BASELINE config 2's shapes."""
    t = seed_tag
    inp = dict(
        ws=det_tensor(f'{t}.ws', [n, num_ws, w_dim]),
        pose_feat=det_tensor(f'{t}.pose_feat', [n, feat_ch, 8, 8]),
        cat_feat={str(r): det_tensor(f'{t}.cat{r}', [n, 64, r, r]) for r in (512, 256, 128, 64)},
        denorm_upper_input=det_tensor(f'{t}.du', [n, 3, 512, 512], 'uniform'),
        denorm_lower_input=det_tensor(f'{t}.dl', [n, 3, 512, 512], 'uniform'),
        denorm_upper_mask=det_tensor(f'{t}.mu', [n, 1, 512, 512], 'blockmask'),
        denorm_lower_mask=det_tensor(f'{t}.ml', [n, 1, 512, 512], 'blockmask'),
        gt_parsing=det_tensor(f'{t}.parsing', [n, 1, 512, 512], 'labels7') if labels else None,
    )
    return inp
