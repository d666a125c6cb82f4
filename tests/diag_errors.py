"""Print measured rel-L2 errors of every parity case (GPU box diagnostic, not a test)."""
import json, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import golden, fixture_params, rel_l2, randn
from diffusion_models_dsdiff_amd import ops, blocks as B
from diffusion_models_dsdiff_amd.UNet_DS_Diff.model import DSUnetModel
from oracle import unet as O

g = golden("ops")
for dim, key in ((320, "temb_int_320"), (32, "temb_int_32")):
    y = ops.timestep_embedding(torch.from_numpy(g["temb_t_int"]).cuda(), dim).cpu().numpy()
    print("temb", dim, "max abs", np.abs(y - g[key]).max())
y = ops.timestep_embedding(torch.from_numpy(g["temb_t_float"]).cuda(), 320).cpu().numpy()
print("temb float max abs", np.abs(y - g["temb_float_320"]).max())
emb = randn((2, 128), 20).cuda()
for key, kw, shp, xs in [("res_same", dict(channels=64, out_channels=64), (2, 64, 16, 16), 21),
                         ("res_skip", dict(channels=32, out_channels=64), (2, 32, 16, 16), 22),
                         ("res_film", dict(channels=32, out_channels=64, use_scale_shift_norm=True), (2, 32, 16, 16), 23)]:
    m = B.ResBlock(emb_channels=128, dropout=0.0, **kw); m.load_state_dict(fixture_params(g, key))
    print(key, rel_l2(m(randn(shp, xs).cuda(), emb), g[key + "_y"]))
for key, kw, shp, xs in [("attn_new_c64_t64", dict(channels=64, num_head_channels=16, use_new_attention_order=True), (2, 64, 8, 8), 30),
                         ("attn_new_c128_t1024", dict(channels=128, num_head_channels=32, use_new_attention_order=True), (1, 128, 32, 32), 33)]:
    m = B.AttentionBlock(**kw); m.load_state_dict(fixture_params(g, key))
    print(key, rel_l2(m(randn(shp, xs).cuda()), g[key + "_y"]))
m = B.FeatureDisentangle(64, 32); m.load_state_dict(fixture_params(g, "disentangle"))
print("disentangle", rel_l2(m(randn((2, 64, 4, 4), 51).cuda()), g["disentangle_y"]))
for key, xs, shp, sc, of in [("gn_silu_320", 11, (2, 320, 16, 16), 2.0, 0.5)]:
    sd = fixture_params(g, key); x = randn(shp, xs) * sc + of
    y = ops.group_norm(ops.to_nhwc(x).cuda(), sd["0.weight"].cuda(), sd["0.bias"].cuda(), silu=True)
    print(key, rel_l2(ops.to_nchw(y), g[key + "_y"]))
for key, xs, shp in [("conv3x3_s1", 13, (2, 32, 16, 16))]:
    sd = fixture_params(g, key)
    y = ops.conv2d(ops.to_nhwc(randn(shp, xs)).cuda(), sd["weight"].cuda(), sd["bias"].cuda())
    print(key, rel_l2(ops.to_nchw(y), g[key + "_y"]))
gm = golden("model")
for key in ("tiny", "tinyfilm"):
    params = json.loads(str(gm[key + "_cfg"]))
    m = DSUnetModel(**params); m.load_state_dict(fixture_params(gm, key))
    for C, xs in ((2, 70), (4, 71)):
        x = randn((2, C, 32, 32), xs)
        for tk, t in (("int", torch.tensor([999, 17])), ("float", torch.tensor([499.5, 20.0]))):
            y, feats = m(x.cuda(), t.cuda())
            print(key, C, tk, "out", rel_l2(y, gm[f"{key}_c{C}_{tk}_y"]),
                  {k: round(rel_l2(torch.stack(v), gm[f"{key}_c{C}_feat_{k}"]), 9) for k, v in feats.items()} if tk == "int" else "")
