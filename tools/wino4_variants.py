"""Dev tool: build (here) and time (GPU box, one process, round-robin, median) variants of the Winograd F(4x4,3x3) kernel (csrc/conv2d_wino4.h) compiled with
-DWINO4_EXP=<mask>.  Ablation bits (results wrong by design): 1 no U loads in the K loop, 2 no transform phase, 4 no tail,
8 no halo DMA (see conv2d_wino4.h)."""
import sys, os, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
VARIANTS = [int(v) for v in os.environ.get('WINO_VARIANTS', '0').split(',')]
DEFINE = os.environ.get('WINO_DEFINE', 'WINO4_EXP')       # WINO_DEFINE=W4X_EXP WINO_ALGO=4: the X3 form's switches
ALGO = int(os.environ.get('WINO_ALGO', '2'))
TAG = 'wino4_exp' if DEFINE == 'WINO4_EXP' else DEFINE.lower()
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
plugins = {v: custom_ops.get_plugin(f'{TAG}{v}', sources=SRC, extra_hipcc_flags=[f'-D{DEFINE}={v}'], build_only=True) for v in VARIANTS}
if sys.argv[1] == 'build':
    print(plugins)
    sys.exit(0)

import torch
from torch_utils.ops import conv2d_mfma
from torch_utils.ops import _native as nat
libs = {}
if os.environ.get('WINO_PREV'):                  # a prebuilt csrc/wino_prev.so (older revision of the sources) as variant -1
    prev = custom_ops.NativePlugin('wino_prev', os.path.join(custom_ops.CSRC_DIR, 'wino_prev.so'), 'conv2d_plugin')
    _orig0 = custom_ops.get_plugin
    custom_ops.get_plugin = lambda name, **kw: prev
    conv2d_mfma._plugin = None
    libs[-1] = conv2d_mfma._init('wino_prev')
    custom_ops.get_plugin = _orig0
for v in VARIANTS:
    conv2d_mfma._plugin = None
    custom_ops.PLUGIN_SOURCES[f'{TAG}{v}'] = SRC
    _orig = custom_ops.get_plugin
    custom_ops.get_plugin = lambda name, _v=v, **kw: _orig(name, extra_hipcc_flags=[f'-D{DEFINE}={_v}'], abi_name='conv2d_plugin', **kw)
    libs[v] = conv2d_mfma._init(f'{TAG}{v}')
    custom_ops.get_plugin = _orig

SHAPES = [(8, 256, 128, 128), (8, 256, 64, 128), (8, 512, 64, 64), (8, 64, 512, 512), (8, 128, 256, 256)]
if os.environ.get('WINO_SHAPES'):               # e.g. WINO_SHAPES=4,256,128,128;4,16,512,512  (N,H,Cin,Cout)
    SHAPES = [tuple(int(v) for v in sh.split(',')) for sh in os.environ['WINO_SHAPES'].split(';')]
rounds = int(os.environ.get('WINO_ROUNDS', '7'))
for (N, H, cin, cout) in SHAPES:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    ALL = ([-1] if -1 in libs else []) + VARIANTS
    times = {v: [] for v in ALL}
    packed = {}
    for v in ALL:
        conv2d_mfma._plugin = libs[v]
        packed[v] = conv2d_mfma.pack_weight(w, winograd=ALGO)
    outs = {}
    for v in ALL:
        conv2d_mfma._plugin = libs[v]
        outs[v] = conv2d_mfma.conv2d_forward(x, packed[v], cout, 3, 3, pad=(1, 1), winograd=ALGO).clone()
    print('   max |variant - first|: ' + '  '.join(f'[{v}] {float((outs[v] - outs[ALL[0]]).abs().max()):.1e}' for v in ALL), flush=True)
    if os.environ.get('WINO_CHECK'):            # every variant against aten's convolution, three launches each (a race shows as a changing / huge figure)
        ref = torch.nn.functional.conv2d(x, w, padding=1)
        for v in ALL:
            conv2d_mfma._plugin = libs[v]
            errs = [float((conv2d_mfma.conv2d_forward(x, packed[v], cout, 3, 3, pad=(1, 1), winograd=ALGO) - ref).abs().max()) for _ in range(3)]
            print(f'   [{v}] max |y - aten| over three launches: ' + ' '.join(f'{e:.2e}' for e in errs), flush=True)
    for r in range(rounds + 1):
        for v in ALL:
            conv2d_mfma._plugin = libs[v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                conv2d_mfma.conv2d_forward(x, packed[v], cout, 3, 3, pad=(1, 1), winograd=ALGO)
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    fl = 2.0 * N * cout * H * H * cin * 9
    print(f'N{N} H{H} {cin}->{cout}: ' + '  '.join(f'[{v}] {statistics.median(times[v]):7.1f}us {fl / statistics.median(times[v]) / 1e6:5.1f}TF' for v in ALL), flush=True)
