"""Case tables shared by ``make_golden.py`` (which runs the REFERENCE on them) and
the tests (which run the oracle and the HIP path on them).  Inputs come from
``detgen.det_tensor(<case name>.<tensor>)``; only outputs live in the .npz files.
"""

import math

FIR_1331 = [1, 3, 3, 1]
# 12 arbitrary taps standing in for augment.py's sym6 wavelet low-pass (augment.py:27-43):
# only the length (>= 8 => separable path, upfirdn2d.py:103-104) matters to the op.
FIR_12 = [0.015, -0.035, -0.11, 0.03, 0.41, 0.79, 0.79, 0.41, 0.03, -0.11, -0.035, 0.015]

# name, x shape, filter spec, up, down, padding, flip_filter, gain
#   filter spec: ('taps', list) -> setup_filter(list); ('rand2d', [fh, fw]) -> det 2-D filter; None -> identity
UPFIRDN2D_CASES = [
    ('blur_after_upconv', [2, 3, 17, 17], ('taps', FIR_1331), 1, 1, [1, 1, 1, 1], False, 4),          # conv2d_resample.py:140
    ('skip_img_up2',      [2, 3, 9, 9],   ('taps', FIR_1331), 2, 1, [2, 1, 2, 1], False, 4),          # networks.py:2186
    ('down2_1x1_skip',    [2, 3, 16, 16], ('taps', FIR_1331), 1, 2, [1, 1, 1, 1], False, 1),          # conv2d_resample.py:108
    ('blur_before_s2',    [2, 3, 16, 16], ('taps', FIR_1331), 1, 1, [2, 2, 2, 2], False, 1),          # conv2d_resample.py:120
    ('sym_up2_sep',       [1, 2, 10, 13], ('taps', FIR_12),   2, 1, [6, 5, 6, 5], False, 4),          # augment.py:290
    ('sym_down2_sep',     [1, 2, 40, 44], ('taps', FIR_12),   1, 2, [-7, -7, -7, -7], True, 1),       # augment.py:301
    ('rand_nonsquare',    [1, 5, 33, 20], ('rand2d', [3, 5]), (2, 1), (1, 2), [3, -1, 0, 2], False, 1.5),
    ('rand_up3_down2',    [2, 2, 11, 7],  ('rand2d', [5, 4]), 3, 2, [2, 3, 4, 1], True, 0.7),
    ('crop_only',         [1, 3, 12, 12], ('rand2d', [2, 2]), 1, 1, [-2, -1, -3, 0], False, 1),
    ('identity_none',     [1, 2, 6, 5],   None,               1, 1, 0, False, 2.0),
    ('one_pixel',         [1, 1, 1, 1],   ('taps', FIR_1331), 2, 1, [2, 1, 2, 1], False, 4),
    ('big_filter_large',  [1, 2, 40, 37], ('rand2d', [25, 27]), 1, 1, [12, 14, 13, 11], False, 1),    # > 24 taps: "large" kernel territory
]
UPFIRDN2D_DTYPE_CASES = ['blur_after_upconv', 'skip_img_up2', 'down2_1x1_skip']   # also run in fp16 / bf16 / fp64

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
# name suffix, has bias, gain, clamp, dim, x shape
BIAS_ACT_VARIANTS = [
    ('plain',       False, None,            None, 1, [2, 5, 7, 3]),
    ('bias',        True,  None,            None, 1, [2, 5, 7, 3]),
    ('bias_clamp',  True,  None,            1.5,  1, [2, 5, 7, 3]),
    ('gain_clamp',  False, math.sqrt(0.5),  0.75, 1, [2, 5, 7, 3]),
    ('bias_dim0',   True,  2.0,             None, 0, [6, 4]),
    ('bias_last',   True,  None,            256., 2, [3, 2, 9]),
]

# name, x shape, w shape, filter taps or None, up, down, padding, groups, flip_weight
CONV2D_RESAMPLE_CASES = [
    ('plain3x3',       [2, 4, 9, 9],   [6, 4, 3, 3], FIR_1331, 1, 1, 1, 1, True),     # conv2d_resample.py:145-147
    ('plain3x3_noflip', [2, 4, 9, 9],  [6, 4, 3, 3], FIR_1331, 1, 1, 1, 1, False),
    ('up2_3x3',        [2, 4, 8, 8],   [6, 4, 3, 3], FIR_1331, 2, 1, 1, 1, False),    # :125-142 (SynthesisLayer up=2)
    ('down2_3x3',      [2, 4, 16, 16], [6, 4, 3, 3], FIR_1331, 1, 2, 1, 1, True),     # :119-122
    ('down2_1x1',      [2, 4, 16, 16], [6, 4, 1, 1], FIR_1331, 1, 2, 0, 1, True),     # :107-110
    ('up2_1x1',        [2, 4, 8, 8],   [6, 4, 1, 1], FIR_1331, 2, 1, 0, 1, False),    # :113-116
    ('plain7x7',       [1, 3, 20, 20], [8, 3, 7, 7], FIR_1331, 1, 1, 3, 1, True),     # spade_encoder[0]
    ('grouped3x3',     [1, 8, 9, 9],   [12, 4, 3, 3], FIR_1331, 1, 1, 1, 2, True),    # fused modconv, N=2
    ('grouped_up2',    [1, 8, 8, 8],   [12, 4, 3, 3], FIR_1331, 2, 1, 1, 2, False),
    ('odd_pad_generic', [1, 3, 10, 10], [4, 3, 3, 3], FIR_1331, 1, 1, [1, 0, 2, 1], 1, True),  # :150-154
    ('up2_down2',      [1, 3, 8, 8],   [4, 3, 3, 3], FIR_1331, 2, 2, 1, 1, True),
    # wide enough (Cout > 32, Cin >= 16) for the product's Winograd F(2x2,3x3) kernel; ragged width (W % 4 != 0) takes its scalar-DMA form
    ('plain3x3_wide',  [2, 32, 20, 24], [80, 32, 3, 3], FIR_1331, 1, 1, 1, 1, True),
    ('plain3x3_wide_noflip_ragged', [1, 16, 13, 19], [64, 16, 3, 3], FIR_1331, 1, 1, 1, 1, False),
]

# name, N, Cin, Cout, k, H, up, demodulate, fused, noise kind ('none' | 'const' | 'per_sample')
MODCONV_CASES = [
    ('fused_demod',            2, 5, 6, 3, 9, 1, True,  True,  'none'),
    ('fused_demod_noise',      2, 5, 6, 3, 9, 1, True,  True,  'const'),
    ('nonfused_demod_noise',   2, 5, 6, 3, 9, 1, True,  False, 'per_sample'),
    ('nonfused_demod',         2, 5, 6, 3, 9, 1, True,  False, 'none'),
    ('fused_up2_noise',        2, 5, 6, 3, 8, 2, True,  True,  'const'),
    ('nonfused_up2_noise',     2, 5, 6, 3, 8, 2, True,  False, 'const'),
    ('torgb_fused',            2, 5, 3, 1, 9, 1, False, True,  'none'),
    ('torgb_nonfused',         2, 5, 3, 1, 9, 1, False, False, 'none'),
    ('nodemod_noise_nonfused', 2, 5, 6, 3, 9, 1, False, False, 'const'),
    # wide enough for the product's Winograd kernel (modulated form: style scale in the prologue, demodulation + noise in the tail)
    ('fused_demod_noise_wide', 2, 32, 72, 3, 16, 1, True,  True,  'const'),
    ('nonfused_demod_wide',    2, 16, 64, 3, 12, 1, True,  False, 'per_sample'),
]

# Reduced synthesis network for G6 (full 512^2 spatial size, small channel counts).
G6_KW = dict(w_dim=32, img_resolution=512, img_channels=3, channel_base=2048, channel_max=32, conv_clamp=256)
G6_FEAT_CH = 32      # channels of pose_feat = channels at res 8
G6_SUB = 4           # outputs stored at [..., ::4, ::4]
G6_CROP = (192, 256, 160, 224)   # plus this full-resolution crop (y0, y1, x0, x1)
