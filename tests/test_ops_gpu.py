"""Kernel-level parity (GPU): each hand-written HIP kernel vs the CPU oracle / reference-generated fixtures.

Tolerances (fp32 path, stated per test): convolution / linear rel-L2 <= 2e-6, GroupNorm+SiLU <= 2e-6,
attention <= 3e-6, timestep embedding abs <= 1e-4 (sin/cos of arguments up to 1e3 amplify a 1-ulp
difference in the frequency table; see misc.hip).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet as O
from util import golden, fixture_params, rel_l2, randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from diffusion_models_dsdiff_amd import ops as m, _lib
    _lib.require_gpu(0)
    return m


def cu(t):
    return t.cuda().contiguous()


def attn_ref64(qkv, heads, new_order):
    """float64 evaluation of QKVAttention / QKVAttentionLegacy (same association as oracle.unet.qkv_attention)."""
    import math
    bs, width, length = qkv.shape
    ch = width // (3 * heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    qkv = qkv.double()
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q, k, v = [t.reshape(bs * heads, ch, length) for t in (q, k, v)]
    else:
        q, k, v = qkv.reshape(bs * heads, ch * 3, length).split(ch, dim=1)
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)


CONV_CASES = [
    # N, H, W, Cin, Cout, ks, stride, ups
    (2, 16, 16, 32, 64, 3, 1, False),
    (2, 16, 16, 64, 64, 3, 2, False),
    (2, 8, 8, 96, 32, 1, 1, False),
    (2, 32, 32, 1, 32, 3, 1, False),      # first layer (Cin = 1): direct kernel
    (1, 8, 8, 32, 32, 3, 1, True),        # nearest x2 folded into the gather
    (3, 12, 20, 64, 160, 3, 1, False),    # ragged M (720 rows), NT = 5
    (1, 32, 32, 320, 320, 3, 1, False),   # hot shape (small spatial)
    (2, 8, 8, 960, 480, 1, 1, False),
    (1, 16, 16, 320, 1, 3, 1, False),     # out conv: Cout = 1, masked N tile
    (1, 5, 7, 36, 20, 3, 1, False),       # Cin % 32 != 0 (masked K chunk), odd sizes
    (1, 4, 4, 6, 5, 3, 2, False),         # scalar fallback
    (2, 64, 64, 128, 96, 3, 1, False),
    (2, 128, 128, 64, 320, 3, 1, False),  # >= 512 tiles: the shape class where the library itself picks the A-direct structure
    (1, 8, 8, 960, 960, 3, 1, False),     # batch-1 bottleneck layer: 30 output tiles, 270 k-tiles -> split-K x16
    (1, 8, 8, 320, 100, 3, 1, False),     # split-K x11 with ragged M (64 rows) and ragged N (100 columns)
]


# arithmetic modes of the convolution: fp32 MFMA (exact fmaf chain), bf16x6 (fp32 split into 3 bf16 pieces, 6 products:
# fp32-grade), bf16x3 (2 pieces, 3 products: ~2^-16 per product).  Tolerances are rel-L2 vs float64.
PREC_TOL = {"f32": 2e-6, "bf16x6": 2e-6, "f16x3": 3e-6, "bf16x3": 3e-5}


@pytest.mark.parametrize("prec", ["f32", "bf16x6", "f16x3", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_vs_torch_cpu(ops, case, prec):
    N, H, W, Cin, Cout, ks, stride, ups = case
    g = torch.Generator().manual_seed(sum(case[:6]) + 7)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=ks // 2)
    y = ops.conv2d(cu(ops.to_nhwc(x)), cu(w), cu(b), stride=stride, upsample=ups, precision=prec)
    assert rel_l2(ops.to_nchw(y), ref) < PREC_TOL[prec]


@pytest.mark.parametrize("structure", ["adirect", "adirect256", "staged"])
@pytest.mark.parametrize("prec", ["bf16x6", "f16x3", "bf16x3"])
def test_conv2d_split_structures(ops, prec, structure):
    """Both kernel structures of the split-bf16 convolution (the library picks per shape) on every eligible case,
    incl. the epilogue fusions, stride 2, folded upsample and ragged M / N tiles."""
    for case in CONV_CASES:
        N, H, W, Cin, Cout, ks, stride, ups = case
        if Cin % 32:
            continue
        g = torch.Generator().manual_seed(sum(case[:6]) + 11)
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
        b = torch.randn(Cout, generator=g)
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
        ref = F.conv2d(xin.double(), w.double(), b.double(), stride=stride, padding=ks // 2)
        emb = torch.randn(N, Cout, generator=g)
        res = torch.randn(*ref.shape, generator=g)
        ref = ref + emb.double()[:, :, None, None] + res.double()
        y = ops.conv2d(cu(ops.to_nhwc(x)), cu(w), cu(b), stride=stride, upsample=ups, emb=cu(emb), res=cu(ops.to_nhwc(res)),
                       precision=prec, structure=structure)
        assert rel_l2(ops.to_nchw(y), ref) < PREC_TOL[prec], (case, prec, structure)


WINO_CASES = [
    # N, H, W, Cin, Cout        (3x3, stride 1; >= 4096 output pixels, W a power of two <= 256, Cin % 32 == 0, Cout = 0 or 64 mod 128)
    (1, 64, 64, 64, 128),        # one full N tile
    (2, 32, 64, 32, 320),        # 128 + 128 + 64: the narrow last N tile; non-square
    (1, 64, 64, 96, 64),         # a single narrow tile, 18 k-tiles
    (3, 44, 32, 64, 192),        # ragged last block (2112 tiles = 16.5 blocks), blocks straddle samples
    (16, 16, 16, 320, 960),      # a 16x16 layer of the network at batch 16 (7 full N tiles + 1 narrow)
    (1, 128, 256, 32, 128),      # full-width rows: one block = one image row; exercises the folded-upsample variant too
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd_f23(ops, case):
    """conv_wino.hip (F(2,3) along the width, bf16x6 operands) against float64, with and without the epilogue fusions, next to
    the direct bf16x6 kernel on the same data: same tolerance (2e-6), and the printed ratio shows what the fp32 input /
    output transforms cost in accuracy."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    xc, wc, bc = cu(ops.to_nhwc(x)), cu(w), cu(b)
    yw = ops.conv2d(xc, wc, bc, precision="bf16x6", structure="winograd")
    yd = ops.conv2d(xc, wc, bc, precision="bf16x6")
    ew, ed = rel_l2(ops.to_nchw(yw), ref), rel_l2(ops.to_nchw(yd), ref)
    print(f"winograd F(2,3) {case}: rel-L2 vs fp64 {ew:.3e} (direct bf16x6: {ed:.3e}, ratio {ew / ed:.2f})")
    assert ew < PREC_TOL["bf16x6"]
    emb = torch.randn(N, Cout, generator=g)
    res = torch.randn(*ref.shape, generator=g)
    ref2 = ref + emb.double()[:, :, None, None] + res.double()
    y2 = ops.conv2d(xc, wc, bc, emb=cu(emb), res=cu(ops.to_nhwc(res)), precision="bf16x6", structure="winograd")
    assert rel_l2(ops.to_nchw(y2), ref2) < PREC_TOL["bf16x6"]
    assert torch.equal(y2, ops.conv2d(xc, wc, bc, emb=cu(emb), res=cu(ops.to_nhwc(res)), precision="bf16x6", structure="winograd"))
    # nearest x2 folded in front of the convolution (Upsample): the loader reads pixel (h >> 1, w >> 1)
    if H * W * N >= 4 * 4096:
        xs = x[:, :, : H // 2, : W // 2].contiguous()
        refu = F.conv2d(F.interpolate(xs.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
        yu = ops.conv2d(cu(ops.to_nhwc(xs)), wc, bc, upsample=True, precision="bf16x6", structure="winograd")
        assert rel_l2(ops.to_nchw(yu), refu) < PREC_TOL["bf16x6"]


def test_conv2d_winograd_rejects_ineligible(ops):
    from diffusion_models_dsdiff_amd import _lib
    g = torch.Generator().manual_seed(1)
    for shp, cout, stride in (((1, 64, 16, 16), 128, 1), ((1, 64, 64, 63), 128, 1), ((1, 64, 114, 36), 128, 1), ((1, 64, 64, 64), 96, 1),
                              ((1, 64, 64, 64), 128, 2)):
        x = torch.randn(*shp, generator=g)
        w = torch.randn(cout, shp[1], 3, 3, generator=g)
        with pytest.raises(_lib.DsdError):
            ops.conv2d(cu(ops.to_nhwc(x)), cu(w), cu(torch.zeros(cout)), stride=stride, precision="bf16x6", structure="winograd")


def test_conv2d_split_extreme_magnitudes(ops):
    """bf16 pieces keep fp32's exponent range: tiny and huge operands, exact zeros, and denormal-scale residuals."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 8, 8, generator=g)
    x[:, :16] *= 1e-20
    x[:, 16:32] *= 1e18
    x[:, 32:40] = 0.0
    w = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    w[:8] *= 1e-10
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    for prec in ("bf16x6", "bf16x3"):
        for structure in ("staged", "adirect", "adirect256"):
            y = ops.conv2d(cu(ops.to_nhwc(x)), cu(w), None, precision=prec, structure=structure)
            assert bool(torch.isfinite(y).all())
            assert rel_l2(ops.to_nchw(y), ref) < PREC_TOL[prec], (prec, structure)
    # fp16 pieces do NOT have that range: the call must fail loudly instead of returning inf / NaN
    from diffusion_models_dsdiff_amd import _lib
    for structure in ("staged", "adirect"):
        with pytest.raises(_lib.DsdError, match="fp16 range"):
            ops.conv2d(cu(ops.to_nhwc(x)), cu(w), None, precision="f16x3", structure=structure)
    small = x.clone()
    small[:, 16:32] *= 1e-18                      # back inside the range: fine again (tiny values lose nothing that matters)
    y = ops.conv2d(cu(ops.to_nhwc(small)), cu(w), None, precision="f16x3")
    assert rel_l2(ops.to_nchw(y), F.conv2d(small.double(), w.double(), None, padding=1)) < PREC_TOL["f16x3"]


@pytest.mark.parametrize("prec", ["f32", "bf16x6"])
def test_conv2d_epilogue_emb_and_residual(ops, prec):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 64, 8, 8, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(96, generator=g)
    emb = torch.randn(3, 96, generator=g)
    res = torch.randn(3, 96, 8, 8, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + emb.double()[:, :, None, None] + res.double()
    y = ops.conv2d(cu(ops.to_nhwc(x)), cu(w), cu(b), emb=cu(emb), res=cu(ops.to_nhwc(res)), precision=prec)
    assert rel_l2(ops.to_nchw(y), ref) < 2e-6


def test_conv2d_golden(ops):
    g = golden("ops")
    for key, xs, shp, kw in [("conv3x3_s1", 13, (2, 32, 16, 16), {}), ("conv3x3_s2", 14, (2, 64, 16, 16), {"stride": 2}),
                             ("conv1x1", 15, (2, 96, 8, 8), {}), ("conv3x3_c1", 16, (2, 1, 32, 32), {})]:
        sd = fixture_params(g, key)
        y = ops.conv2d(cu(ops.to_nhwc(randn(shp, xs))), cu(sd["weight"]), cu(sd["bias"]), **kw)
        assert rel_l2(ops.to_nchw(y), g[key + "_y"]) < 2e-6, key
    sd = fixture_params(g, "upsample")
    y = ops.conv2d(cu(ops.to_nhwc(randn((2, 32, 8, 8), 40))), cu(sd["conv.weight"]), cu(sd["conv.bias"]), upsample=True)
    assert rel_l2(ops.to_nchw(y), g["upsample_y"]) < 2e-6
    sd = fixture_params(g, "downsample")
    y = ops.conv2d(cu(ops.to_nhwc(randn((2, 32, 16, 16), 41))), cu(sd["op.weight"]), cu(sd["op.bias"]), stride=2)
    assert rel_l2(ops.to_nchw(y), g["downsample_y"]) < 2e-6


@pytest.mark.parametrize("shape,silu", [((2, 320, 16, 16), True), ((2, 960, 8, 8), True), ((3, 32, 5, 7), False),
                                        ((1, 1920, 4, 4), True), ((2, 64, 64, 64), True), ((1, 2880, 2, 2), False)])
def test_group_norm_vs_oracle(ops, shape, silu):
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(*shape, generator=g) * 3 + 1.5
    gamma, beta = torch.randn(shape[1], generator=g), torch.randn(shape[1], generator=g)
    ref = F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)
    if silu:
        ref = F.silu(ref)
    y = ops.group_norm(cu(ops.to_nhwc(x)), cu(gamma), cu(beta), silu=silu)
    assert rel_l2(ops.to_nchw(y), ref) < 2e-6


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 37, 45), (1, 128, 16, 32), (3, 192, 5, 7), (2, 256, 33, 64), (2, 320, 64, 64),
                                     (1, 320, 256, 256)])
def test_gn_silu_conv_out1_vs_fp64(ops, N, C, H, W):
    """The network's last layer (GroupNorm32 + SiLU + Conv3x3 to ONE channel, model.py:511-515) as one memory-bound pass
    (conv_out1.hip): tile edges (sizes that are no multiples of the 16 x 32 tile), every channel count it takes, the headline
    size.  fp32 FMAs, hardware exp2 / reciprocal in the SiLU: <= 2e-6 of the float64 evaluation."""
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    w, b = torch.randn(1, C, 3, 3, generator=g) / (3 * C ** 0.5), torch.randn(1, generator=g)
    ref = F.conv2d(F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)), w.double(), b.double(), padding=1)
    y = ops.gn_silu_conv_out1(cu(ops.to_nhwc(x)), cu(gamma), cu(beta), cu(w), cu(b))
    err = rel_l2(y.unsqueeze(1), ref)
    print(f"gn_silu_conv_out1 N={N} C={C} {H}x{W}: rel-L2 vs fp64 {err:.3e}")
    assert err < 2e-6


def test_gn_silu_conv_out1_refuses_other_widths(ops):
    x = torch.zeros(1, 8, 8, 96).cuda()
    with pytest.raises(RuntimeError, match="input channels unsupported"):
        ops.gn_silu_conv_out1(x, torch.ones(96).cuda(), torch.zeros(96).cuda(), torch.zeros(1, 96, 3, 3).cuda(), torch.zeros(1).cuda())


def test_group_norm_golden(ops):
    g = golden("ops")
    for key, xs, shp, sc, of in [("gn_silu_320", 11, (2, 320, 16, 16), 2.0, 0.5), ("gn_silu_960", 12, (2, 960, 8, 8), 3.0, -1.0)]:
        sd = fixture_params(g, key)
        x = randn(shp, xs) * sc + of
        y = ops.group_norm(cu(ops.to_nhwc(x)), cu(sd["0.weight"]), cu(sd["0.bias"]), silu=True)
        assert rel_l2(ops.to_nchw(y), g[key + "_y"]) < 2e-6, key


@pytest.mark.parametrize("N,T,C,heads,new", [(2, 64, 64, 4, True), (2, 64, 64, 2, False), (1, 4, 64, 2, True),
                                             (1, 1024, 128, 4, True), (1, 144, 96, 2, True), (2, 200, 128, 2, True),
                                             (1, 256, 96, 3, False), (1, 70, 256, 2, True), (2, 330, 144, 2, True),
                                             (1, 4096, 128, 2, True)])
def test_qkv_attention_vs_oracle(ops, N, T, C, heads, new):
    g = torch.Generator().manual_seed(T + C)
    qkv = torch.randn(N, 3 * C, T, generator=g)
    ref = attn_ref64(qkv, heads, new)                                  # [N, C, T]
    assert rel_l2(O.qkv_attention(qkv, heads, new), ref) < 1e-6        # the oracle agrees with the fp64 evaluation
    for split in (False, True):                                        # fp32 MFMA kernel / bf16x6 kernel (3 bf16 pieces, 6 products)
        a = ops.qkv_attention(cu(qkv.permute(0, 2, 1)), heads, new, split=split)        # [N, T, C]
        err = rel_l2(a.permute(0, 2, 1), ref)
        print(f"attention N={N} T={T} C={C} heads={heads} split={split}: rel-L2 vs fp64 {err:.3e}")
        assert err < 3e-6, split


def test_attention_online_softmax_rescale(ops):
    """Force the running-max rescale: one key far above the rest late in the sequence."""
    g = torch.Generator().manual_seed(1)
    N, T, C, heads = 1, 256, 32, 1
    qkv = torch.randn(N, 3 * C, T, generator=g)
    qkv[0, C:2 * C, 200] *= 30.0      # spike key 200 (4th LDS stage)
    ref = attn_ref64(qkv, heads, True)
    for split in (False, True):
        a = ops.qkv_attention(cu(qkv.permute(0, 2, 1)), heads, True, split=split)
        assert rel_l2(a.permute(0, 2, 1), ref) < 3e-6, split


def test_timestep_embedding(ops):
    """Two paths.  (1) With the frequency table of the host that generated the fixture (tests/golden/temb.npz; the shim
    installs its own host's table through dsd_set_timestep_freqs, UNet_DS_Diff/model.py) the arguments t*f are
    bit-identical to the reference's, so only sinf / cosf can differ from ATen's: bound 2 ulp of 1.0 (2.4e-7 absolute on
    values in [-1, 1]; the measured maximum is printed).  torch's fp32 exp differs by 1 ulp between CPU ISAs, which
    sin/cos(t*f) at t ~ 1e3 amplify to ~6e-5 — so (2) the paths whose table is NOT the fixture host's (this host's
    torch.exp; the library's own fp64 exp when no table is given) are held to 1e-4 only."""
    import math
    g, go = golden("temb"), golden("ops")
    ulp2 = 2 * 2.0 ** -23
    worst = 0.0
    for dim in (320, 32, 96, 1152):
        fr = torch.from_numpy(g[f"freqs_{dim}"]).cuda()
        for kind in ("int", "float"):
            t = torch.from_numpy(g["t_" + kind]).cuda()
            y = ops.timestep_embedding(t, dim, fr).cpu().numpy()
            err = float(np.abs(y - g[f"{kind}_{dim}"]).max())
            worst = max(worst, err)
            assert err <= ulp2, (dim, kind, err)
            assert np.abs(ops.timestep_embedding(t, dim).cpu().numpy() - g[f"{kind}_{dim}"]).max() < 1e-4
            half = dim // 2
            here = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half).cuda()
            assert np.abs(ops.timestep_embedding(t, dim, here).cpu().numpy() - g[f"{kind}_{dim}"]).max() < 1e-4
    print(f"timestep_embedding with the fixture host's table: max abs err {worst:.3e} ({worst / 2.0 ** -23:.2f} ulp of 1.0)")
    y = ops.timestep_embedding(torch.from_numpy(go["temb_t_int"]).cuda(), 320).cpu().numpy()   # round-1 fixtures, no table
    assert np.abs(y - go["temb_int_320"]).max() < 1e-4


@pytest.mark.parametrize("N,K,O_,act", [(16, 1280, 640, 1), (2, 128, 64, 0), (3, 60, 17, 1), (9, 320, 1280, 0),
                                        # the LDS-staged kernel (9..16 rows, >= 512 outputs): DiT-B's adaLN width, a ragged last
                                        # workgroup (O % 32 != 0); 5 and 17 rows take the wave-per-row kernel
                                        (16, 768, 55296, 1), (9, 1280, 1000, 1), (12, 256, 513, 0), (5, 1280, 1000, 1), (17, 512, 1024, 1)])
def test_linear(ops, N, K, O_, act):
    g = torch.Generator().manual_seed(K)
    x, w, b = torch.randn(N, K, generator=g), torch.randn(O_, K, generator=g) / K ** 0.5, torch.randn(O_, generator=g)
    xin = F.silu(x.double()) if act else x.double()
    ref = F.linear(xin, w.double(), b.double())
    assert rel_l2(ops.linear(cu(x), cu(w), cu(b), act), ref) < 2e-6


def test_philox_normal_moments(ops):
    z = ops.philox_normal(1 << 20, 1234, 7).cpu().double()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and abs((z ** 4).mean() - 3) < 5e-2
    z2 = ops.philox_normal(1 << 20, 1234, 8).cpu().double()
    assert abs((z * z2).mean()) < 5e-3                       # different steps are uncorrelated
    assert torch.equal(ops.philox_normal(4096, 99, 3), ops.philox_normal(4096, 99, 3))   # deterministic


def test_diagnostic_entry_points():
    """The measurement helpers behind bench.py's roofline.sustained and tools/conv_stamps.py: bare-MFMA loop rates are
    physical (below the 2.5 PF nominal peak, zeros faster than random data), and the stamped instantiation of the dominant
    convolution reports ordered clock stamps for every workgroup."""
    import ctypes as C
    from diffusion_models_dsdiff_amd import _lib
    L = _lib.lib()
    rates = {}
    for v in (0, 1, 2):
        ms, tf = C.c_float(), C.c_double()
        _lib.check(L.dsd_bench_mfma_peak(v, 4, 2.0, 3, C.byref(ms), C.byref(tf)))
        rates[v] = tf.value
        assert ms.value > 0 and 300.0 < tf.value < 2600.0, (v, ms.value, tf.value)
    assert rates[2] > rates[0] > 0.9 * rates[1]      # zeros hold a higher clock; LDS-fed is not faster than register-fed
    ms, tf = C.c_float(), C.c_double()
    _lib.check(L.dsd_bench_mfma_peak(4, 4, 2.0, 3, C.byref(ms), C.byref(tf)))       # round 3: variants 4-7 = the 16x16x32 shape
    assert 300.0 < tf.value < 2600.0
    assert L.dsd_bench_mfma_peak(8, 4, 2.0, 3, C.byref(ms), C.byref(tf)) != 0       # beyond the table: refused
    import numpy as np
    buf = np.zeros((512, 8), dtype=np.int64)
    n = C.c_int()
    _lib.check(L.dsd_bench_conv2d_stamps(1, 256, 256, 320, 320, 2, 0, buf.ctypes.data_as(C.POINTER(C.c_longlong)), 512, C.byref(n)))
    assert n.value == 512                             # 65536 output pixels / 256 rows x 2 column tiles of 160
    st = buf[: n.value]
    assert (st > 0).all()
    for c in (0, 1):                                  # core clock and 100 MHz clock: entry <= loop start < loop end <= exit
        t = st[:, c::2]
        assert (t[:, 0] <= t[:, 1]).all() and (t[:, 1] < t[:, 2]).all() and (t[:, 2] <= t[:, 3]).all()
    assert L.dsd_bench_conv2d_stamps(1, 256, 256, 320, 320, 0, 5, buf.ctypes.data_as(C.POINTER(C.c_longlong)), 512, C.byref(n)) != 0   # not instantiated
    assert L.dsd_bench_conv2d_stamps(1, 256, 256, 320, 320, 0, 0, buf.ctypes.data_as(C.POINTER(C.c_longlong)), 8, C.byref(n)) != 0     # too little room
    assert L.dsd_bench_conv2d_stamps(1, 64, 64, 320, 320, 0, 0, buf.ctypes.data_as(C.POINTER(C.c_longlong)), 512, C.byref(n)) != 0     # split-K shape: no diagnostic build


TR_CASES = [
    # N, H, W, Cin, Cout: 3x3 stride-1 layers whose 256-row tiles are whole image rows and whose grid keeps the 160-column
    # tile (the shape class the tap-reuse instantiation of the A-direct kernel takes: conv_split.hip, TR)
    (1, 256, 256, 32, 320),      # one image row per tile; first / last rows and columns are the zero padding
    (2, 128, 128, 64, 320),      # two rows per tile
    (8, 64, 64, 32, 160),        # four rows per tile, a single column tile
    (32, 32, 32, 96, 320),       # eight rows per tile = a quarter of a sample; 27 k-tiles, three channel chunks
    (4, 64, 128, 64, 320),       # non-square
    (2, 64, 64, 32, 320, True),  # nearest x2 folded into the gather: 128 x 128 output from a 64 x 64 input
    (1, 128, 64, 64, 160, True), # the same, non-square, 256 x 128 output
    # 128-column tiles (round 3: the tap-reuse kernel's second width — 128 / 256 / 512 output channels)
    # (sizes at which the planner picks the 256-row tile: >= 160 row tiles)
    (1, 256, 256, 64, 128),      # one column tile
    (1, 256, 256, 96, 256),      # two column tiles, three channel chunks
    (4, 128, 128, 128, 256),     # two rows per tile
    (16, 64, 64, 32, 512),       # four rows per tile, four column tiles
    (1, 128, 128, 128, 128, True),   # nearest x2 folded in: 256 x 256 output
]


@pytest.mark.parametrize("case", TR_CASES)
def test_conv2d_row_tiles(ops, case):
    """bf16x6 on the large 3x3 layers, every epilogue fusion on, against float64 — the kernel structure is the library's
    choice (with tap reuse enabled these shapes run the TR instantiation)."""
    N, H, W, Cin, Cout = case[:5]
    ups = len(case) > 5 and case[5]
    g = torch.Generator().manual_seed(sum(case[:5]) + 5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    emb = torch.randn(N, Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    H, W = xin.shape[2:]
    res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1) + emb.double()[:, :, None, None] + res.double()
    y = ops.conv2d(cu(ops.to_nhwc(x)), cu(w), cu(b), upsample=bool(ups), emb=cu(emb), res=cu(ops.to_nhwc(res)), precision="bf16x6")
    assert rel_l2(ops.to_nchw(y), ref) < PREC_TOL["bf16x6"], case
    # border pixels exactly where the padding matters
    yy, rr = ops.to_nchw(y).double().cpu(), ref
    for sl in ((..., 0, slice(None)), (..., H - 1, slice(None)), (..., slice(None), 0), (..., slice(None), W - 1)):
        assert rel_l2(yy[sl], rr[sl]) < PREC_TOL["bf16x6"], (case, sl)


@pytest.mark.parametrize("case", TR_CASES)
def test_conv2d_row_tiles_mfma16(ops, case):
    """The tap-reuse kernel with its matrix work issued as v_mfma_f32_16x16x32_bf16 (conv_tr16.hip, dsd_set_conv_mfma16): the
    same cases, the same float64 bar, every epilogue fusion on; and the two MFMA shapes agree to fp32 summation order."""
    from diffusion_models_dsdiff_amd import _lib
    L = _lib.lib()
    N, H, W, Cin, Cout = case[:5]
    ups = len(case) > 5 and case[5]
    g = torch.Generator().manual_seed(sum(case[:5]) + 5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    emb = torch.randn(N, Cout, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    H, W = xin.shape[2:]
    res = torch.randn(N, Cout, H, W, generator=g)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1) + emb.double()[:, :, None, None] + res.double()
    args = (cu(ops.to_nhwc(x)), cu(w), cu(b))
    kw = dict(upsample=bool(ups), emb=cu(emb), res=cu(ops.to_nhwc(res)), precision="bf16x6")
    prev = L.dsd_set_conv_mfma16(1)
    try:
        y16 = ops.conv2d(*args, **kw)
    finally:
        L.dsd_set_conv_mfma16(prev)
    y32 = ops.conv2d(*args, **kw)
    assert rel_l2(ops.to_nchw(y16), ref) < PREC_TOL["bf16x6"], case
    assert rel_l2(y16, y32) < 1e-6
    if case[:5] == (1, 256, 256, 32, 320):
        assert not torch.equal(y16, y32)          # a different kernel really ran (the planner gives this shape the 256 x 160 tile)
    yy = ops.to_nchw(y16).double().cpu()
    for sl in ((..., 0, slice(None)), (..., H - 1, slice(None)), (..., slice(None), 0), (..., slice(None), W - 1)):
        assert rel_l2(yy[sl], ref[sl]) < PREC_TOL["bf16x6"], (case, sl)
