"""The single-product half-precision kernels (DSD_PREC_F16 / DSD_PREC_BF16, BASELINE configs[4] "DiT 512x512 fp16") through
the C ABI: gemm16.hip (nn.Linear + the DiTBlock epilogues) and attention16.hip (timm Attention core), and the DiT handle
in those modes.

The kernel checks compare with float64 arithmetic on THE SAME 16-bit-rounded operands, so the bars are those of the kernel
(fp32 accumulation order, one rounding of the result): 1.5 ulp of the 16-bit type per element, and a few 1e-4 rel-L2.
The network checks compare with oracle/dit.py twice: with its restatement of the mode's roundings (tight) and with the fp32
oracle (the mode's own accuracy, an autocast-grade bar written in the test).  PARITY UNPINNED BY THE REFERENCE for all of
this: DiT_models.py needs timm (absent), and torch.autocast on a GPU is what the reference would run."""
import math

import pytest
import torch

from oracle import dit as OD
from oracle.synth import synth_params
from util import rel_l2, randn

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16}
EPS = {"f16": 2.0 ** -11, "bf16": 2.0 ** -8}     # half an ulp relative


def r16(x, dt):
    return x.to(DT[dt]).double()


def gelu_tanh64(v):
    return 0.5 * v * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (v + 0.044715 * v ** 3)))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 192, 64), (300, 288, 96), (77, 104, 40), (512, 768, 768), (1024, 3072, 768),
                                    (640, 768, 3072), (256, 256, 8)])
def test_gemm_half_store_vs_fp64(dt, M, N, K):
    from diffusion_models_dsdiff_amd import ops
    x, w, b = randn((M, K), 1), randn((N, K), 2) * (K ** -0.5), randn((N,), 3) * 0.1
    want = r16(x, dt) @ r16(w, dt).T + b.double()
    got = ops.gemm_half(x.cuda(), w.cuda(), b.cuda(), dtype=dt).double().cpu()
    # one rounding of an fp32-accumulated sum: within 1.5 half-ulps of the exact value everywhere
    assert float(((got - want).abs() / (want.abs() + 1e-3)).max()) < 3.2 * EPS[dt]
    assert rel_l2(got, want) < 1.2 * EPS[dt]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_gemm_half_gelu_and_gated_epilogues(dt):
    from diffusion_models_dsdiff_amd import ops
    M, N, K, T = 384, 320, 192, 128     # 3 samples of 128 tokens
    x, w, b = randn((M, K), 4), randn((N, K), 5) * (K ** -0.5), randn((N,), 6) * 0.1
    lin = r16(x, dt) @ r16(w, dt).T + b.double()
    want = gelu_tanh64(r16(lin.float(), dt))
    got = ops.gemm_half(x.cuda(), w.cuda(), b.cuda(), dtype=dt, epi="gelu").double().cpu()
    assert rel_l2(got, want) < 3 * EPS[dt]
    # gated residual: y += gate[m // T] * round16(x w^T + b), fp32 stream
    y0, gate = randn((M, N), 7), randn((M // T, N), 8)
    want = y0.double() + gate.double().repeat_interleave(T, dim=0) * r16(lin.float(), dt)
    y = y0.clone().cuda()
    ops.gemm_half(x.cuda(), w.cuda(), b.cuda(), dtype=dt, epi="gated", gate=gate.cuda(), T=T, y=y)
    assert rel_l2(y.cpu(), want) < 2 * EPS[dt]
    # no bias
    got = ops.gemm_half(x.cuda(), w.cuda(), None, dtype=dt).double().cpu()
    assert rel_l2(got, r16(x, dt) @ r16(w, dt).T) < 1.2 * EPS[dt]


def attn64(qkv, heads, dt):
    """float64 softmax(q k^T d^-1/2) v on the 16-bit-rounded q (already scaled by d^-1/2 log2 e), k, v."""
    N, T, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    q = (qkv[..., :C].to(DT[dt]).float() * (math.log2(math.e) / math.sqrt(d))).to(DT[dt]).double()   # the entry's two roundings
    k, v = r16(qkv[..., C:2 * C], dt), r16(qkv[..., 2 * C:], dt)
    sp = lambda z: z.reshape(N, T, heads, d).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) * math.log(2.0)
    return (torch.softmax(s, dim=-1) @ sp(v)).permute(0, 2, 1, 3).reshape(N, T, C)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("N,T,C,heads", [(2, 64, 64, 4), (2, 64, 96, 3), (1, 1024, 768, 12), (2, 100, 144, 2), (1, 333, 128, 1),
                                          (1, 4096, 128, 2), (3, 31, 48, 2), (1, 200, 48, 6),
                                          # head dim 64 = the LDS-DMA kernel (three-stage ring): one, two, three and many tiles,
                                          # ragged last tiles (keys beyond Tk come as out-of-range DMA lanes), query blocks < 128
                                          (3, 31, 128, 2), (1, 64, 64, 1), (1, 65, 64, 1), (2, 100, 128, 2), (1, 129, 192, 3),
                                          (2, 200, 64, 1), (1, 333, 64, 1), (1, 1000, 128, 2)])
def test_attention_half_vs_fp64(dt, N, T, C, heads):
    from diffusion_models_dsdiff_amd import ops
    qkv = randn((N, T, 3 * C), 11)
    qkv[..., :C] *= 2.0      # logits with a spread of several units
    want = attn64(qkv, heads, dt)
    got = ops.attention_half(qkv.cuda(), heads, dtype=dt).double().cpu()
    # P and the output are rounded to 16 bits once each
    assert rel_l2(got, want) < 3.0 * EPS[dt], rel_l2(got, want)
    assert bool(torch.isfinite(got).all())


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_attention_half_running_maximum_branch(dt):
    """The rescale of O / l / the pending tile only happens when a score exceeds the running maximum by more than 2^thr: spike
    single keys so that the branch fires at chosen tiles (first sub-tile, mid-sequence, last tile), for some queries of a
    wave only; thr = 0 (move the maximum at every increase), 3 and the default (8)
    must agree with float64 and with each other to rounding."""
    from diffusion_models_dsdiff_amd import ops
    N, T, C, heads = 1, 512, 128, 2
    d = C // heads
    qkv = randn((N, T, 3 * C), 21) * 0.5
    for key, qsel, amp in ((5, slice(0, 40), 6.0), (200, slice(17, 23), 9.0), (259, slice(100, 300), 12.0), (511, slice(0, 512, 7), 15.0)):
        # key `key` of head 0 lines up with the selected queries: score jumps by ~amp^2 natural units there
        dirn = torch.nn.functional.normalize(randn((d,), 100 + key), dim=0)
        qkv[0, key, C:C + d] = dirn * amp
        qkv[0, qsel, 0:d] += dirn * amp * math.sqrt(d) / 2
    want = attn64(qkv, heads, dt)
    outs = {thr: ops.attention_half(qkv.cuda(), heads, dtype=dt, thr=thr).double().cpu() for thr in (0.0, -1.0, 3.0)}
    for thr, got in outs.items():
        assert bool(torch.isfinite(got).all()) and rel_l2(got, want) < 3.0 * EPS[dt], (thr, rel_l2(got, want))
    assert rel_l2(outs[0.0], outs[-1.0]) < 2.0 * EPS[dt] and rel_l2(outs[3.0], outs[-1.0]) < 2.0 * EPS[dt]


def make(**kw):
    from diffusion_models_dsdiff_amd.UNet_DS_Diff.DiT_models import DiT
    m = DiT(**kw)
    sd = synth_params([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 901)
    m.load_state_dict(sd, strict=True)
    return m, sd


@pytest.mark.parametrize("kw,labels", [
    (dict(input_size=16, patch_size=2, in_channels=4, hidden_size=64, depth=2, num_heads=4, num_classes=10), True),
    (dict(input_size=32, patch_size=4, in_channels=3, hidden_size=96, depth=3, num_heads=3, num_classes=5, learn_sigma=False,
          class_dropout_prob=0.0), False),
    (dict(input_size=64, patch_size=8, in_channels=4, hidden_size=768, depth=2, num_heads=12, num_classes=1000), True),   # DiT-B/8 blocks
    (dict(input_size=32, patch_size=2, in_channels=4, hidden_size=144, depth=2, num_heads=2, num_classes=0), False),     # head dim 72 (DiT-XL's)
])
def test_dit_half_modes_vs_oracle(kw, labels):
    """Autocast-grade bars (written here, measured values printed): against the oracle's restatement of the mode's own
    roundings 1e-4 (fp16) / 8e-4 (bf16) rel-L2 (fp32 summation order, a lazily moved softmax maximum and values that sit on
    a rounding boundary are all that differs; measured 1.7-2.1e-5 / 1.4-1.7e-4); against the fp32 oracle 2e-4 (fp16) / 1.5e-3
    (bf16) for two or three blocks (measured 3.2-3.7e-5 / 2.6-2.9e-4) — operands carry 11 / 8 significant bits."""
    m, sd = make(**kw)
    N = 2
    x = randn((N, kw["in_channels"] - 1, kw["input_size"], kw["input_size"]), 5)
    cond = randn((N, 1, kw["input_size"], kw["input_size"]), 6)
    t = torch.tensor([17.0, 999.0])
    y = torch.tensor([1, 4]) if labels else None
    okw = dict(patch_size=kw["patch_size"], num_heads=kw["num_heads"], out_channels=m.out_channels)
    full = OD.dit_forward(sd, torch.cat([x, cond], 1), t, y, **okw)
    for prec, bar_emu, bar32 in (("f16", 1e-4, 2e-4), ("bf16", 8e-4, 1.5e-3)):
        emu = OD.dit_forward(sd, torch.cat([x, cond], 1), t, y, half=DT[prec], **okw)
        m.set_precision(prec)
        assert m.precision == prec
        got = m(x.cuda(), t.cuda(), None if y is None else y.cuda(), cond=cond.cuda())
        e_emu, e_full = rel_l2(got, emu), rel_l2(got, full)
        print(f"DiT {kw['hidden_size']}x{kw['depth']} p{kw['patch_size']} {prec}: vs emulated roundings {e_emu:.3e}, vs fp32 oracle "
              f"{e_full:.3e} (the emulation itself: {rel_l2(emu, full):.3e})")
        assert got.shape == full.shape and bool(torch.isfinite(got).all())
        assert e_emu < bar_emu, prec
        assert e_full < bar32, prec
        assert torch.equal(got, m(x.cuda(), t.cuda(), None if y is None else y.cuda(), cond=cond.cuda()))   # deterministic
    m.set_precision("bf16x6")
    assert rel_l2(m(x.cuda(), t.cuda(), None if y is None else y.cuda(), cond=cond.cuda()), full) < 1e-5   # and back


def test_dit_b8_at_512_half():
    """BASELINE config 5's shape in its own arithmetic: DiT-B/8 on 512x512 (4096 tokens, 12 heads of 64), two blocks, fp16:
    finite, deterministic, batch rows independent, one slice against the fp32 oracle at the autocast-grade bar."""
    kw = dict(input_size=512, patch_size=8, in_channels=4, hidden_size=768, depth=2, num_heads=12, num_classes=1000)
    m, sd = make(**kw)
    x = randn((2, 4, 512, 512), 11)
    t = torch.tensor([500.0, 3.0])
    m.set_precision("f16")
    got = m(x.cuda(), t.cuda())
    assert bool(torch.isfinite(got).all()) and torch.equal(got, m(x.cuda(), t.cuda()))
    alone = m(x[1:].cuda(), t[1:].cuda())
    assert torch.equal(alone, got[1:])          # same tiles, same order: bit-identical
    want = OD.dit_forward(sd, x[:1], t[:1], None, patch_size=8, num_heads=12, out_channels=m.out_channels)
    err = rel_l2(got[:1], want)
    print(f"DiT-B/8 blocks @512, fp16 vs fp32 oracle: {err:.3e}")
    assert err < 2e-4


def test_half_modes_are_dit_only():
    from diffusion_models_dsdiff_amd import _lib
    from diffusion_models_dsdiff_amd.blocks import AttentionBlock
    blk = AttentionBlock(64, num_head_channels=32, use_new_attention_order=True)
    with pytest.raises(_lib.DsdError, match="DSD_BLOCK_DIT"):
        blk.set_precision("f16")
