"""f16x3 accuracy vs operand magnitude (GPU box probe): fp16 pieces have an absolute floor (subnormal spacing 6e-8)."""
import sys, os, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffusion_models_dsdiff_amd import ops
from util import rel_l2
g = torch.Generator().manual_seed(1)
x0 = torch.randn(8, 128, 64, 64, generator=g); w = torch.randn(320, 128, 3, 3, generator=g) / (128 * 9) ** 0.5   # 512 tiles: the split kernels are really used
for scale in (1e3, 1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
    x = x0 * scale
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    row = {}
    for prec in ("f32", "bf16x6", "f16x3", "bf16x3"):
        y = ops.conv2d(ops.to_nhwc(x).cuda(), w.cuda(), None, precision=prec)
        row[prec] = f"{rel_l2(ops.to_nchw(y), ref):.2e}"
    print(f"|x| ~ {scale:g}", row)
