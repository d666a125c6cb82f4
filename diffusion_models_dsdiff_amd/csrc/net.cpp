// net.cpp — builds the launch plan of the DSUnetModel forward (and of single blocks) over the hand-written
// kernels, owns the parameter slab and the liveness-planned workspace arena.
//
// Graph structure follows UNet_DS_Diff/model.py:282-515 (ctor) and :629-756 (forward); blocks follow
// ldm/modules/diffusionmodules/openaimodel.py (ResBlock :264-284, AttentionBlock :467-473, Upsample :111-121,
// Downsample :162-164), UNet_DS_Diff/model.py:152-168 (FeatureDisentangle), Disc_diff/guided_diffusion/unet.py:82-109
// (SE_Attention) and ldm/modules/attention.py (CrossAttention :164-193, BasicTransformerBlock :326-330,
// SpatialTransformer :411-428).
#include "net.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace dsd;

// =============================================================================================== arena
size_t ArenaPlanner::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    int best = -1;
    for (int i = 0; i < (int)free_list.size(); ++i)
        if (free_list[i].size >= bytes && (best < 0 || free_list[i].size < free_list[best].size)) best = i;
    if (best >= 0) {
        const size_t off = free_list[best].off;
        free_list[best].off += bytes;
        free_list[best].size -= bytes;
        if (free_list[best].size == 0) free_list.erase(free_list.begin() + best);
        return off;
    }
    // extend the top (absorbing a free block that touches it)
    if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
        const size_t off = free_list.back().off;
        free_list.pop_back();
        top = off + bytes;
        peak = std::max(peak, top);
        return off;
    }
    const size_t off = top;
    top += bytes;
    peak = std::max(peak, top);
    return off;
}

void ArenaPlanner::release(size_t off, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    auto it = std::lower_bound(free_list.begin(), free_list.end(), off, [](const Blk& b, size_t o) { return b.off < o; });
    it = free_list.insert(it, Blk{off, bytes});
    // coalesce with next / previous
    if (it + 1 != free_list.end() && it->off + it->size == (it + 1)->off) {
        it->size += (it + 1)->size;
        free_list.erase(it + 1);
    }
    if (it != free_list.begin() && (it - 1)->off + (it - 1)->size == it->off) {
        (it - 1)->size += it->size;
        free_list.erase(it);
    }
}

// =============================================================================================== spec
namespace {

enum { L_CONV, L_RES, L_ATTN, L_DOWN, L_UP };
struct Layer {
    int kind = 0, cin = 0, cout = 0, ch = 0, heads = 0;
    bool up = false, down = false;
    int dh = 0;          // L_ATTN as SpatialTransformer (UNetModel(use_spatial_transformer=True)): d_head (> 0), heads = n_heads
};
// SpatialTransformer options of a UNetModel handle (openaimodel.py:623-631): trailing integers of its iargs
struct StOpt {
    int on = 0, depth = 1, ctx_dim = 0, linear = 0;
};
StOpt st_opt_from_iargs(const std::vector<int32_t>& a) {
    StOpt o;
    if (a.size() < 12) return o;
    const int nl = a[10];
    const size_t base = 12 + 2 * (size_t)nl + (size_t)a[11 + 2 * nl];
    if (a.size() >= base + 4) {
        o.on = a[base]; o.depth = a[base + 1]; o.ctx_dim = a[base + 2]; o.linear = a[base + 3];
    }
    return o;
}
struct Spec {
    std::vector<std::vector<Layer>> input_blocks, output_blocks;
    std::vector<Layer> middle;
    int conv_ch = 0, half = 0, final_ch = 0, ted = 0;
};

Layer mk_res(int cin, int cout, bool up = false, bool down = false) {
    Layer l;
    l.kind = L_RES; l.cin = cin; l.cout = cout; l.up = up; l.down = down;
    return l;
}

// model.py:282-515
Spec build_spec(const dsd_config& c, bool plain = false, bool st = false) {
    Spec s;
    const int mc = c.model_channels;
    DSD_CHECK(c.n_levels >= 1 && c.n_levels <= DSD_MAX_LEVELS, "channel_mult must have 1..%d entries", DSD_MAX_LEVELS);
    DSD_CHECK(plain || c.in_channels == 1, "in_channels must be 1: every stream of DSUnetModel.forward receives one plane (model.py:654-663)");
    DSD_CHECK(c.in_channels >= 1 && mc >= 32 && mc % 32 == 0 && c.out_channels >= 1, "bad channel counts");
    int num_heads = c.num_heads, num_heads_upsample = c.num_heads_upsample;
    if (num_heads_upsample == -1) num_heads_upsample = num_heads;
    const int nhc = c.num_head_channels;
    DSD_CHECK(!(num_heads == -1 && nhc == -1), "Either num_heads or num_head_channels has to be set");
    auto in_attn = [&](int ds) {
        for (int i = 0; i < c.n_attention_resolutions; ++i)
            if (c.attention_resolutions[i] == ds) return true;
        return false;
    };
    auto attn = [&](int ch, int heads_arg) {
        int dim_head;
        if (nhc == -1) {
            dim_head = ch / num_heads;
        } else {
            num_heads = ch / nhc;
            dim_head = nhc;
        }
        if (c.legacy) dim_head = st ? ch / num_heads : nhc;   // (openaimodel.py:745-747)
        const int h_arg = heads_arg < 0 ? num_heads : heads_arg;
        Layer l;
        l.kind = L_ATTN; l.ch = ch;
        if (st) {   // SpatialTransformer(ch, num_heads, dim_head, ...) — also in the decoder (num_heads, not num_heads_upsample)
            DSD_CHECK(num_heads > 0 && dim_head > 0, "spatial transformer: heads %d x dim_head %d", num_heads, dim_head);
            l.heads = num_heads;
            l.dh = dim_head;
            return l;
        }
        if (dim_head == -1) {
            l.heads = h_arg;
        } else {
            DSD_CHECK(ch % dim_head == 0, "q,k,v channels %d is not divisible by num_head_channels %d", ch, dim_head);
            l.heads = ch / dim_head;
        }
        return l;
    };
    Layer first;
    first.kind = L_CONV; first.cin = c.in_channels; first.cout = mc;
    s.input_blocks.push_back({first});
    std::vector<int> chans{mc};
    int ch = mc, ds = 1;
    for (int level = 0; level < c.n_levels; ++level) {
        const int mult = c.channel_mult[level];
        for (int nr = 0; nr < c.num_res_blocks[level]; ++nr) {
            std::vector<Layer> layers{mk_res(ch, mult * mc)};
            ch = mult * mc;
            if (in_attn(ds)) layers.push_back(attn(ch, -1));
            s.input_blocks.push_back(layers);
            chans.push_back(ch);
        }
        if (level != c.n_levels - 1) {
            if (c.resblock_updown) {
                s.input_blocks.push_back({mk_res(ch, ch, false, true)});
            } else {
                Layer d;
                d.kind = L_DOWN; d.ch = ch;
                s.input_blocks.push_back({d});
            }
            chans.push_back(ch);
            ds *= 2;
        }
    }
    s.middle = {mk_res(ch, ch), attn(ch, -1), mk_res(ch, ch)};
    for (int level = c.n_levels - 1; level >= 0; --level) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i < c.num_res_blocks[level] + 1; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            std::vector<Layer> layers{mk_res(ch + ich, mc * mult)};
            ch = mc * mult;
            if (in_attn(ds)) layers.push_back(attn(ch, num_heads_upsample));
            if (level && i == c.num_res_blocks[level]) {
                if (c.resblock_updown) {
                    layers.push_back(mk_res(ch, ch, true, false));
                } else {
                    Layer u;
                    u.kind = L_UP; u.ch = ch;
                    layers.push_back(u);
                }
                ds /= 2;
            }
            s.output_blocks.push_back(layers);
        }
    }
    s.conv_ch = c.channel_mult[0] * mc * c.channel_mult[c.n_levels - 1];
    s.half = s.conv_ch / 2;
    s.final_ch = ch;
    s.ted = mc * 4;
    DSD_CHECK(s.final_ch == mc, "out.0 normalises %d channels but out.2 expects model_channels=%d", s.final_ch, mc);
    return s;
}

// ------------------------------------------------------------------------------------------- params
void add_param(dsd_handle* h, const std::string& name, std::vector<int64_t> shape, bool pack3x3 = false, int region = 0) {
    DSD_CHECK(!h->pidx.count(name), "duplicate parameter %s", name.c_str());
    Param p;
    p.name = name; p.shape = shape; p.pack3x3 = pack3x3; p.region = region;
    p.numel = 1;
    for (auto d : shape) p.numel *= d;
    h->pidx[name] = (int)h->params.size();
    h->params.push_back(p);
}
void p_lin(dsd_handle* h, const std::string& n, int cin, int cout, bool bias = true, int wregion = 0) {
    add_param(h, n + ".weight", {cout, cin}, false, wregion);
    if (bias) add_param(h, n + ".bias", {cout}, false, wregion ? wregion + 1 : 0);
}
void p_conv(dsd_handle* h, const std::string& n, int cin, int cout, int k) {
    add_param(h, n + ".weight", {cout, cin, k, k}, k == 3);
    add_param(h, n + ".bias", {cout});
}
void p_conv1d(dsd_handle* h, const std::string& n, int cin, int cout) {
    add_param(h, n + ".weight", {cout, cin, 1});
    add_param(h, n + ".bias", {cout});
}
void p_norm(dsd_handle* h, const std::string& n, int c) {
    add_param(h, n + ".weight", {c});
    add_param(h, n + ".bias", {c});
}
void p_res(dsd_handle* h, const std::string& p, int cin, int cout, int ted, bool film, bool emb_region) {
    p_norm(h, p + ".in_layers.0", cin);
    p_conv(h, p + ".in_layers.2", cin, cout, 3);
    p_lin(h, p + ".emb_layers.1", ted, film ? 2 * cout : cout, true, emb_region ? 1 : 0);
    p_norm(h, p + ".out_layers.0", cout);
    p_conv(h, p + ".out_layers.3", cout, cout, 3);
    if (cin != cout) p_conv(h, p + ".skip_connection", cin, cout, 1);
}
void p_attn(dsd_handle* h, const std::string& p, int ch) {
    p_norm(h, p + ".norm", ch);
    p_conv1d(h, p + ".qkv", ch, 3 * ch);
    p_conv1d(h, p + ".proj_out", ch, ch);
}
void p_btb(dsd_handle* h, const std::string& p, int dim, int heads, int dh, int cd);
// SpatialTransformer (ldm/modules/attention.py:365-428)
void p_spatial_transformer(dsd_handle* h, const std::string& p, int in_ch, int heads, int dh, int depth, int ctx_dim, int linear) {
    const int inner = heads * dh;
    p_norm(h, p + "norm", in_ch);
    if (linear) p_lin(h, p + "proj_in", in_ch, inner); else p_conv(h, p + "proj_in", in_ch, inner, 1);
    for (int d = 0; d < depth; ++d) p_btb(h, p + "transformer_blocks." + std::to_string(d), inner, heads, dh, ctx_dim);
    if (linear) p_lin(h, p + "proj_out", in_ch, inner); else p_conv(h, p + "proj_out", inner, in_ch, 1);
}
void p_layer(dsd_handle* h, const std::string& p, const Layer& L, int ted, bool film) {
    switch (L.kind) {
        case L_CONV: p_conv(h, p, L.cin, L.cout, 3); break;
        case L_RES: p_res(h, p, L.cin, L.cout, ted, film, true); break;
        case L_ATTN:
            if (L.dh > 0) {
                const StOpt o = st_opt_from_iargs(h->iargs);
                p_spatial_transformer(h, p + ".", L.ch, L.heads, L.dh, o.depth, o.ctx_dim, o.linear);
            } else {
                p_attn(h, p, L.ch);
            }
            break;
        case L_DOWN: p_conv(h, p + ".op", L.ch, L.ch, 3); break;
        case L_UP: p_conv(h, p + ".conv", L.ch, L.ch, 3); break;
    }
}
void p_disentangle(dsd_handle* h, const std::string& n, int cc, int half) {
    p_norm(h, n + ".conv_1.0", cc);
    p_conv(h, n + ".conv_1.2", cc, cc, 3);
    p_norm(h, n + ".conv_2.0", cc);
    p_conv(h, n + ".conv_2.2", cc, half, 1);
}
void p_xattn(dsd_handle* h, const std::string& p, int qd, int cd, int heads, int dh) {
    const int inner = heads * dh;
    p_lin(h, p + ".to_q", qd, inner, false);
    p_lin(h, p + ".to_k", cd, inner, false);
    p_lin(h, p + ".to_v", cd, inner, false);
    p_lin(h, p + ".to_out.0", inner, qd);
}
void p_ff(dsd_handle* h, const std::string& p, int dim, int mult) {
    p_lin(h, p + ".net.0.proj", dim, dim * mult * 2);
    p_lin(h, p + ".net.2", dim * mult, dim);
}
void p_btb(dsd_handle* h, const std::string& p, int dim, int heads, int dh, int cd) {
    p_xattn(h, p + ".attn1", dim, dim, heads, dh);
    p_ff(h, p + ".ff", dim, 4);
    p_xattn(h, p + ".attn2", dim, cd, heads, dh);
    p_norm(h, p + ".norm1", dim);
    p_norm(h, p + ".norm2", dim);
    p_norm(h, p + ".norm3", dim);
}

std::string pre(const std::string& p, const std::string& n) { return p.empty() ? n : p + "." + n; }

// ---- KL-VAE (ldm/modules/diffusionmodules/model.py:452-655, ldm/models/autoencoder.py:26-147)
// iargs: ch, out_ch, in_channels, resolution, z_channels, double_z, embed_dim, num_res_blocks, with_quant, n_mult, mult[..],
//        n_attn, attn_resolutions[..]   (the ddconfig of configs/autoencoder_kl_64x64x3.yaml:14-24 + embed_dim)
struct VaeCfg {
    int ch, out_ch, in_ch, resolution, z, double_z, embed, nrb, with_quant;
    std::vector<int> mult, attn_res;
    bool attn_at(int res) const { return std::find(attn_res.begin(), attn_res.end(), res) != attn_res.end(); }
};
VaeCfg vae_cfg(const std::vector<int32_t>& a) {
    DSD_CHECK(a.size() >= 11, "VAE handle needs >= 11 integer arguments");
    VaeCfg c;
    c.ch = a[0]; c.out_ch = a[1]; c.in_ch = a[2]; c.resolution = a[3]; c.z = a[4]; c.double_z = a[5]; c.embed = a[6];
    c.nrb = a[7]; c.with_quant = a[8];
    const int nm = a[9];
    DSD_CHECK(nm >= 1 && nm <= 8 && (int)a.size() >= 11 + nm, "VAE: bad ch_mult");
    for (int i = 0; i < nm; ++i) c.mult.push_back(a[10 + i]);
    const int na = a[10 + nm];
    DSD_CHECK(na >= 0 && (int)a.size() >= 11 + nm + na, "VAE: bad attn_resolutions");
    for (int i = 0; i < na; ++i) c.attn_res.push_back(a[11 + nm + i]);
    DSD_CHECK(c.ch % 32 == 0 && c.ch >= 32, "VAE: ch = %d must be a multiple of 32 (GroupNorm(32))", c.ch);
    DSD_CHECK(c.nrb >= 1 && c.z >= 1 && c.in_ch >= 1 && c.out_ch >= 1 && c.embed >= 1, "VAE: bad configuration");
    return c;
}
// ---- DiT (UNet_DS_Diff/DiT_models.py:145-262).  iargs: input_size, patch_size, in_channels, hidden_size, depth, num_heads,
//      mlp_hidden (= int(hidden_size * mlp_ratio)), num_classes, learn_sigma, use_cfg_embedding (class_dropout_prob > 0)
struct DitCfg {
    int input, p, cin, D, depth, heads, mlp, classes, learn_sigma, cfg_emb;
    int cout() const { return learn_sigma ? cin / 3 * 2 : cin; }   // DiT.__init__ :163 (sic)
    int T() const { return (input / p) * (input / p); }
};
DitCfg dit_cfg(const std::vector<int32_t>& a) {
    DSD_CHECK(a.size() >= 10, "DiT handle needs 10 integer arguments");
    DitCfg c{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]};
    DSD_CHECK(c.p >= 1 && c.input % c.p == 0 && c.D % c.heads == 0 && (c.D / c.heads) % 4 == 0 && c.D / c.heads <= 128 && c.D % 4 == 0,
              "DiT: unsupported geometry (input %d, patch %d, hidden %d, heads %d)", c.input, c.p, c.D, c.heads);
    DSD_CHECK(c.depth >= 1 && c.cout() >= 1 && c.mlp % 4 == 0, "DiT: bad configuration");
    return c;
}
void p_dit(dsd_handle* h, const DitCfg& c) {
    add_param(h, "x_embedder.proj.weight", {c.D, c.cin, c.p, c.p});
    add_param(h, "x_embedder.proj.bias", {c.D});
    p_lin(h, "t_embedder.mlp.0", 256, c.D);
    p_lin(h, "t_embedder.mlp.2", c.D, c.D);
    if (c.classes > 0) add_param(h, "y_embedder.embedding_table.weight", {c.classes + (c.cfg_emb ? 1 : 0), c.D});
    add_param(h, "pos_embed", {1, c.T(), c.D});
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = "blocks." + std::to_string(i);
        p_lin(h, b + ".attn.qkv", c.D, 3 * c.D);
        p_lin(h, b + ".attn.proj", c.D, c.D);
        p_lin(h, b + ".mlp.fc1", c.D, c.mlp);
        p_lin(h, b + ".mlp.fc2", c.mlp, c.D);
        p_lin(h, b + ".adaLN_modulation.1", c.D, 6 * c.D, true, /*wregion=*/1);   // all modulations = one GEMM per forward
    }
    p_lin(h, "final_layer.linear", c.D, c.p * c.p * c.cout());
    p_lin(h, "final_layer.adaLN_modulation.1", c.D, 2 * c.D, true, 1);
}

// ---- plain single-stream UNetModel (ldm/modules/diffusionmodules/openaimodel.py:571-958), the denoiser of the latent path.
// iargs: in_channels, model_channels, out_channels, num_heads, num_head_channels, num_heads_upsample, use_scale_shift_norm,
//        resblock_updown, use_new_attention_order, legacy, n_levels, channel_mult[..], num_res_blocks[..] (per level),
//        n_attention_resolutions, attention_resolutions[..]
dsd_config unet_cfg_from_iargs(const std::vector<int32_t>& a) {
    DSD_CHECK(a.size() >= 12, "UNetModel handle needs >= 12 integer arguments");
    dsd_config c{};
    c.in_channels = a[0]; c.model_channels = a[1]; c.out_channels = a[2];
    c.num_heads = a[3]; c.num_head_channels = a[4]; c.num_heads_upsample = a[5];
    c.use_scale_shift_norm = a[6]; c.resblock_updown = a[7]; c.use_new_attention_order = a[8]; c.legacy = a[9];
    c.n_levels = a[10];
    DSD_CHECK(c.n_levels >= 1 && c.n_levels <= DSD_MAX_LEVELS && (int)a.size() >= 12 + 2 * c.n_levels, "UNetModel: bad channel_mult");
    for (int i = 0; i < c.n_levels; ++i) {
        c.channel_mult[i] = a[11 + i];
        c.num_res_blocks[i] = a[11 + c.n_levels + i];
    }
    c.n_attention_resolutions = a[11 + 2 * c.n_levels];
    DSD_CHECK(c.n_attention_resolutions >= 0 && c.n_attention_resolutions <= DSD_MAX_LEVELS &&
              (int)a.size() >= 12 + 2 * c.n_levels + c.n_attention_resolutions, "UNetModel: bad attention_resolutions");
    for (int i = 0; i < c.n_attention_resolutions; ++i) c.attention_resolutions[i] = a[12 + 2 * c.n_levels + i];
    return c;
}
void p_plain_unet(dsd_handle* h) {
    const StOpt so = st_opt_from_iargs(h->iargs);
    DSD_CHECK(!so.on || (so.depth == 1 && so.ctx_dim > 0), "UNetModel: spatial transformer needs transformer_depth 1 and a context_dim");
    const Spec s = build_spec(h->cfg, true, so.on != 0);
    const bool film = h->cfg.use_scale_shift_norm;
    p_lin(h, "time_embed.0", h->cfg.model_channels, s.ted);
    p_lin(h, "time_embed.2", s.ted, s.ted);
    for (size_t bi = 0; bi < s.input_blocks.size(); ++bi)
        for (size_t li = 0; li < s.input_blocks[bi].size(); ++li)
            p_layer(h, "input_blocks." + std::to_string(bi) + "." + std::to_string(li), s.input_blocks[bi][li], s.ted, film);
    for (size_t li = 0; li < s.middle.size(); ++li) p_layer(h, "middle_block." + std::to_string(li), s.middle[li], s.ted, film);
    for (size_t bi = 0; bi < s.output_blocks.size(); ++bi)
        for (size_t li = 0; li < s.output_blocks[bi].size(); ++li)
            p_layer(h, "output_blocks." + std::to_string(bi) + "." + std::to_string(li), s.output_blocks[bi][li], s.ted, film);
    p_norm(h, "out.0", s.final_ch);
    p_conv(h, "out.2", h->cfg.model_channels, h->cfg.out_channels, 3);
}

void p_vae_res(dsd_handle* h, const std::string& p, int cin, int cout) {
    p_norm(h, p + ".norm1", cin);
    p_conv(h, p + ".conv1", cin, cout, 3);
    p_norm(h, p + ".norm2", cout);
    p_conv(h, p + ".conv2", cout, cout, 3);
    if (cin != cout) p_conv(h, p + ".nin_shortcut", cin, cout, 1);
}
void p_vae_attn(dsd_handle* h, const std::string& p, int c) {
    p_norm(h, p + ".norm", c);
    for (const char* n : {"q", "k", "v", "proj_out"}) p_conv(h, p + "." + n, c, c, 1);
}
void p_vae_encoder(dsd_handle* h, const VaeCfg& c) {   // model.py:452-517
    const std::string e = "encoder";
    p_conv(h, e + ".conv_in", c.in_ch, c.ch, 3);
    int res = c.resolution, block_in = c.ch;
    for (size_t l = 0; l < c.mult.size(); ++l) {
        block_in = c.ch * (l == 0 ? 1 : c.mult[l - 1]);
        const int block_out = c.ch * c.mult[l];
        for (int j = 0; j < c.nrb; ++j) {
            p_vae_res(h, e + ".down." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out);
            block_in = block_out;
            if (c.attn_at(res)) p_vae_attn(h, e + ".down." + std::to_string(l) + ".attn." + std::to_string(j), block_in);
        }
        if (l + 1 != c.mult.size()) {
            p_conv(h, e + ".down." + std::to_string(l) + ".downsample.conv", block_in, block_in, 3);
            res /= 2;
        }
    }
    p_vae_res(h, e + ".mid.block_1", block_in, block_in);
    p_vae_attn(h, e + ".mid.attn_1", block_in);
    p_vae_res(h, e + ".mid.block_2", block_in, block_in);
    p_norm(h, e + ".norm_out", block_in);
    p_conv(h, e + ".conv_out", block_in, c.double_z ? 2 * c.z : c.z, 3);
    if (c.with_quant) p_conv(h, "quant_conv", 2 * c.z, 2 * c.embed, 1);   // autoencoder.py:53
}
void p_vae_decoder(dsd_handle* h, const VaeCfg& c) {   // model.py:546-616
    const std::string d = "decoder";
    const int L = (int)c.mult.size();
    if (c.with_quant) p_conv(h, "post_quant_conv", c.embed, c.z, 1);      // autoencoder.py:54
    int block_in = c.ch * c.mult[L - 1];
    int res = c.resolution >> (L - 1);
    p_conv(h, d + ".conv_in", c.z, block_in, 3);
    p_vae_res(h, d + ".mid.block_1", block_in, block_in);
    p_vae_attn(h, d + ".mid.attn_1", block_in);
    p_vae_res(h, d + ".mid.block_2", block_in, block_in);
    for (int l = L - 1; l >= 0; --l) {
        const int block_out = c.ch * c.mult[l];
        for (int j = 0; j < c.nrb + 1; ++j) {
            p_vae_res(h, d + ".up." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out);
            block_in = block_out;
            if (c.attn_at(res)) p_vae_attn(h, d + ".up." + std::to_string(l) + ".attn." + std::to_string(j), block_in);
        }
        if (l != 0) {
            p_conv(h, d + ".up." + std::to_string(l) + ".upsample.conv", block_in, block_in, 3);
            res *= 2;
        }
    }
    p_norm(h, d + ".norm_out", block_in);
    p_conv(h, d + ".conv_out", block_in, c.out_ch, 3);
}

}  // namespace

float* dsd_handle::P(const std::string& name) const {
    auto it = pidx.find(name);
    DSD_CHECK(it != pidx.end(), "unknown parameter %s", name.c_str());
    return reinterpret_cast<float*>(slab + params[it->second].off);
}
const Param& dsd_handle::PP(const std::string& name) const {
    auto it = pidx.find(name);
    DSD_CHECK(it != pidx.end(), "unknown parameter %s", name.c_str());
    return params[it->second];
}

void dsd::net_declare_params(dsd_handle* h) {
    if (!h->is_block) {
        const Spec s = build_spec(h->cfg);
        const bool film = h->cfg.use_scale_shift_norm;
        p_lin(h, "time_embed.0", h->cfg.model_channels, s.ted);
        p_lin(h, "time_embed.2", s.ted, s.ted);
        for (const char* sfx : {"", "_a", "_al", "_l"})
            for (size_t bi = 0; bi < s.input_blocks.size(); ++bi)
                for (size_t li = 0; li < s.input_blocks[bi].size(); ++li)
                    p_layer(h, "input_blocks" + std::string(sfx) + "." + std::to_string(bi) + "." + std::to_string(li),
                            s.input_blocks[bi][li], s.ted, film);
        for (size_t li = 0; li < s.middle.size(); ++li) p_layer(h, "middle_block." + std::to_string(li), s.middle[li], s.ted, film);
        for (size_t bi = 0; bi < s.output_blocks.size(); ++bi)
            for (size_t li = 0; li < s.output_blocks[bi].size(); ++li)
                p_layer(h, "output_blocks." + std::to_string(bi) + "." + std::to_string(li), s.output_blocks[bi][li], s.ted, film);
        p_norm(h, "out.0", s.final_ch);
        p_conv(h, "out.2", h->cfg.model_channels, h->cfg.out_channels, 3);
        for (const char* nm : {"conv_style", "conv_content", "conv_anatomy", "conv_lesion"}) p_disentangle(h, nm, s.conv_ch, s.half);
        for (const char* nm : {"style_proj", "share_content_proj", "anatomy_proj", "lesion_proj"}) {
            p_lin(h, std::string(nm) + ".0.se.0", s.half, s.half / 8, false);
            p_lin(h, std::string(nm) + ".0.se.2", s.half / 8, s.half, false);
            p_conv(h, std::string(nm) + ".1", s.half, s.half, 3);
        }
        p_conv(h, "all_proj.1", s.half * 6, s.conv_ch, 1);
    } else {
        const auto& a = h->iargs;
        auto need = [&](size_t n) { DSD_CHECK(a.size() >= n, "block kind %d needs %zu integer arguments", h->block_kind, n); };
        switch (h->block_kind) {
            case DSD_BLOCK_RES: need(6); p_res(h, "", a[0], a[1], a[2], a[3] != 0, false); break;
            case DSD_BLOCK_ATTN: need(3); p_attn(h, "", a[0]); break;
            case DSD_BLOCK_UPSAMPLE: need(1); p_conv(h, "conv", a[0], a[0], 3); break;
            case DSD_BLOCK_DOWNSAMPLE: need(1); p_conv(h, "op", a[0], a[0], 3); break;
            case DSD_BLOCK_DISENTANGLE: need(2); p_disentangle(h, "", a[0], a[1]); break;
            case DSD_BLOCK_SE:
                need(2);
                p_lin(h, "se.0", a[0], a[0] / a[1], false);
                p_lin(h, "se.2", a[0] / a[1], a[0], false);
                break;
            case DSD_BLOCK_CROSSATTN: need(4); p_xattn(h, "", a[0], a[1], a[2], a[3]); break;
            case DSD_BLOCK_FF_GEGLU: need(2); p_ff(h, "", a[0], a[1]); break;
            case DSD_BLOCK_BASIC_TRANSFORMER: need(4); p_btb(h, "", a[0], a[1], a[2], a[3]); break;
            case DSD_BLOCK_SPATIAL_TRANSFORMER: need(6); p_spatial_transformer(h, "", a[0], a[1], a[2], a[3], a[4], a[5]); break;
            case DSD_BLOCK_DIT: p_dit(h, dit_cfg(a)); break;
            case DSD_BLOCK_UNET:
                h->cfg = unet_cfg_from_iargs(a);
                p_plain_unet(h);
                break;
            case DSD_BLOCK_VAE_ENCODER: p_vae_encoder(h, vae_cfg(a)); break;
            case DSD_BLOCK_VAE_DECODER: p_vae_decoder(h, vae_cfg(a)); break;
            default: fail("unknown block kind %d", h->block_kind);
        }
        // block parameter names carry no leading '.'
        for (auto& p : h->params)
            if (!p.name.empty() && p.name[0] == '.') p.name = p.name.substr(1);
        h->pidx.clear();
        for (size_t i = 0; i < h->params.size(); ++i) h->pidx[h->params[i].name] = (int)i;
    }
    // lay the slab out: general region, then the contiguous emb_layers weight and bias regions
    size_t off = 0;
    for (int region = 0; region < 3; ++region) {
        if (region == 1) h->emb_w_off = off;
        if (region == 2) h->emb_b_off = off;
        for (auto& p : h->params) {
            if (p.region != region) continue;
            p.off = off;
            size_t b = (size_t)p.numel * sizeof(float);
            if (region == 0) b = (b + 255) & ~(size_t)255;  // emb regions stay densely packed (one GEMM over all of them)
            off += b;
            if (region == 2) h->emb_total += p.numel;
        }
        off = (off + 255) & ~(size_t)255;
    }
    h->slab_bytes = off + 256;
    if (h->device >= 0) DSD_HIP(hipMalloc((void**)&h->slab, h->slab_bytes));  // device < 0: table-only handle (host-side checks)
}

void dsd::net_set_param(dsd_handle* h, const char* name, const float* src, const int64_t* shape, int ndim, int src_is_device,
                        hipStream_t s) {
    DSD_CHECK(h->device >= 0 && h->slab, "this handle was created without a device (table only)");
    auto it = h->pidx.find(name);
    DSD_CHECK(it != h->pidx.end(), "unexpected parameter '%s'", name);
    Param& p = h->params[it->second];
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    bool same = (int)p.shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = p.shape[i] == shape[i];
    // a Linear weight [O,I] may also arrive as a 1x1 conv weight [O,I,1,1] / [O,I,1] and vice versa
    if (!same && n == p.numel && ndim >= 2 && p.shape.size() >= 2 && shape[0] == p.shape[0] && shape[1] == p.shape[1]) same = true;
    if (!same) {
        std::string want, got;
        for (auto d : p.shape) want += std::to_string(d) + ",";
        for (int i = 0; i < ndim; ++i) got += std::to_string(shape[i]) + ",";
        fail("size mismatch for %s: expected [%s] got [%s]", name, want.c_str(), got.c_str());
    }
    float* dst = reinterpret_cast<float*>(h->slab + p.off);
    const size_t bytes = (size_t)p.numel * sizeof(float);
    const hipMemcpyKind kind = src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // the repack staging buffer is shared by every upload: an upload on ANOTHER stream than the previous one first waits for it
    if (h->any_param_upload && h->last_param_stream != s) {
        auto pe = h->param_evs.find(h->last_param_stream);
        if (pe != h->param_evs.end()) DSD_HIP(hipStreamWaitEvent(s, pe->second, 0));
    }
    if (p.pack3x3) {
        if (h->staging_bytes < bytes) {
            if (h->staging) {
                DSD_HIP(hipStreamSynchronize(s));
                DSD_HIP(hipFree(h->staging));
            }
            h->staging_bytes = std::max(bytes, (size_t)64 << 20);
            DSD_HIP(hipMalloc((void**)&h->staging, h->staging_bytes));
        }
        DSD_HIP(hipMemcpyAsync(h->staging, src, bytes, kind, s));
        pack_ohwi(h->staging, dst, (int)p.shape[0], (int)p.shape[1], 3, s);
        if (!src_is_device) DSD_HIP(hipStreamSynchronize(s));  // the caller may free/reuse its host buffer
    } else {
        DSD_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
        if (!src_is_device) DSD_HIP(hipStreamSynchronize(s));
    }
    p.set = true;
    // plan-time consumers of the slab (weight splitting) may run on another stream: they wait on this event
    {
        hipEvent_t& ev = h->param_evs[s];
        if (!ev) DSD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        DSD_HIP(hipEventRecord(ev, s));
        h->last_param_stream = s;
        h->any_param_upload = true;
    }
    const std::string conv_name = p.name.size() > 7 ? p.name.substr(0, p.name.size() - 7) : p.name;   // "<conv>.weight" -> "<conv>"
    for (const std::string& key : {conv_name, conv_name + "#f16", conv_name + "#wino", p.name + "#h16", p.name + "#b16"}) {
        auto sp = h->wsplit.find(key);
        if (sp != h->wsplit.end()) {
            DSD_HIP(hipDeviceSynchronize());
            (void)hipFree(sp->second);
            h->wsplit.erase(sp);
            h->wsplit_bytes.erase(key);
            h->plan.valid = false;
            net_drop_graph(h);
        }
    }
}

void dsd::net_drop_graph(dsd_handle* h) {
    if (h->gexec) {
        (void)hipDeviceSynchronize();
        (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
    }
    h->gkey = GraphKey{};
}

void dsd::net_drop_other_pieces(dsd_handle* h, int precision) {
    if (precision == PREC_F32) return;   // the exact mode needs none, but switching back should not re-split everything
    // which family a cached entry belongs to, by the suffix of its key: "#f16" two fp16 pieces (f16x3), "#h16" / "#b16" the
    // single fp16 / bf16 copy of the half-precision modes, anything else the bf16 pieces (bf16x3 / bf16x6, also what the
    // half-precision modes keep for the layers that stay fp32-grade)
    auto ends = [](const std::string& k, const char* sfx) { return k.size() > 4 && k.compare(k.size() - 4, 4, sfx) == 0; };
    bool any = false;
    for (auto it = h->wsplit.begin(); it != h->wsplit.end();) {
        bool keep;
        if (ends(it->first, "#f16")) keep = precision == PREC_F16X3;
        else if (ends(it->first, "#h16")) keep = precision == PREC_F16;
        else if (ends(it->first, "#b16")) keep = precision == PREC_BF16;
        else keep = precision != PREC_F16X3;
        if (!keep) {
            if (!any) DSD_HIP(hipDeviceSynchronize());
            any = true;
            (void)hipFree(it->second);
            h->wsplit_bytes.erase(it->first);
            it = h->wsplit.erase(it);
        } else {
            ++it;
        }
    }
    if (any) net_drop_graph(h);
}

size_t dsd::net_piece_bytes(const dsd_handle* h) {
    size_t b = 0;
    for (const auto& kv : h->wsplit_bytes) b += kv.second;
    return b;
}

void dsd::net_free(dsd_handle* h) {
    if (h->slab) (void)hipFree(h->slab);
    if (h->staging) (void)hipFree(h->staging);
    if (h->arena) (void)hipFree(h->arena);
    if (h->tbuf) (void)hipFree(h->tbuf);
    if (h->mout) (void)hipFree(h->mout);
    if (h->zplane) (void)hipFree(h->zplane);
    if (h->dpm_m) (void)hipFree(h->dpm_m);
    if (h->freqs) (void)hipFree(h->freqs);
    if (h->ovf) (void)hipFree(h->ovf);
    if (h->slice_ids) (void)hipFree(h->slice_ids);
    for (auto& kv : h->param_evs) (void)hipEventDestroy(kv.second);
    h->param_evs.clear();
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->lane_fork) (void)hipEventDestroy(h->lane_fork);
    for (int l = 0; l < 3; ++l) {
        if (h->lane_join[l]) (void)hipEventDestroy(h->lane_join[l]);
        if (h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    }
    for (auto& kv : h->wsplit) (void)hipFree(kv.second);
    h->wsplit.clear();
    h->wsplit_bytes.clear();
    for (auto e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
}

// =============================================================================================== builder
namespace {

struct EmbRef {
    size_t arena_off = 0;  // emb_all tensor
    int64_t col = 0;       // first column of this block
    int stride = 0;
    bool valid = false;
};

struct Builder {
    dsd_handle* hd;
    Plan& plan;
    ArenaPlanner ar;
    int B;
    hipStream_t ps;          // plan-time device work (weight pieces) goes on the caller's stream
    bool split_any = false;
    Builder(dsd_handle* h, Plan& p, int b, hipStream_t s) : hd(h), plan(p), B(b), ps(s) {}
    // ---- stream lanes: between begin_parallel() and end_parallel() ops carry the current lane and NOTHING is recycled (lanes run
    // concurrently: a buffer freed by one must not be handed to another while the first may still be reading it)
    int lane = -1;
    bool parallel = false;
    std::vector<std::pair<size_t, size_t>> deferred;
    void begin_parallel() { parallel = true; }
    // how many launches of the same shape run side by side where the op being built will run (the planner's hint, ConvArgs::lanes)
    // Only for layers of >= 16384 output pixels: there four launches really run side by side (measured, tools/ab_lanes.py:
    // batch 1, 128 x 128 level on the wide tap-reuse kernel instead of narrow tiles: 36.9 -> 35.1 ms per step); deeper levels are
    // short dependent chains of small kernels whose lanes rarely meet in the same layer, and split-K sized for the whole chip
    // stays the better plan for them (with the hint on every lane level: batch 2 -1.8 % instead of -3.0 %, batch 16 +0.3 %).
    int plan_lanes(int64_t out_pixels) const { return parallel && hd->use_lanes && out_pixels >= 16384 ? 4 : 1; }
    void set_lane(int l) { lane = l; }
    void end_parallel() {
        parallel = false;
        lane = -1;
        for (auto& d : deferred) ar.release(d.first, d.second);
        deferred.clear();
    }
    void free_bytes(size_t off, size_t bytes) {
        if (parallel) deferred.emplace_back(off, bytes); else ar.release(off, bytes);
    }

    Tn alloc(int n, int h, int w, int c, int esz = 4) {
        Tn t;
        t.n = n; t.h = h; t.w = w; t.c = c; t.esz = esz;
        t.off = ar.alloc(t.bytes());
        return t;
    }
    // arithmetic of the convolution kernels: the half-precision modes (DiT) keep their few non-transformer layers fp32-grade
    int conv_prec() const { return hd->precision >= PREC_F16 ? PREC_BF16X6 : hd->precision; }
    bool half_mode() const { return hd->precision == PREC_F16 || hd->precision == PREC_BF16; }
    // 16-bit copy of a Linear weight [N][K] (made once per mode at plan time, on the caller's stream)
    const void* w16(const std::string& name) {
        const bool bf = hd->precision == PREC_BF16;
        const std::string key = name + (bf ? "#b16" : "#h16");
        auto it = hd->wsplit.find(key);
        if (it == hd->wsplit.end()) {
            const Param& pw = hd->PP(name);
            void* buf = nullptr;
            DSD_HIP(hipMalloc(&buf, (size_t)pw.numel * 2));
            cast16(W(name), pw.numel, buf, bf ? 1 : 0, ps);
            split_any = true;
            it = hd->wsplit.emplace(key, buf).first;
            hd->wsplit_bytes[key] = (size_t)pw.numel * 2;
        }
        return it->second;
    }
    // nn.Linear on 16-bit operands (gemm16.hip).  x: 16-bit tokens [n, h*w, K].  epi EPI16_STORE / EPI16_GELU: returns the
    // 16-bit result; EPI16_GATED: xres (fp32) += gate * result, returns xres.
    Tn linear16(const std::string& name, const Tn& x, int cout, int epi, const Tn* xres = nullptr, size_t gate_off = 0,
                int gate_col = 0, int gate_stride = 0, int qcols = 0, float qscale = 1.f) {
        DSD_CHECK(x.esz == 2, "linear16 %s: 16-bit input expected", name.c_str());
        const Param& pw = hd->PP(name + ".weight");
        const int K = x.c, M = x.n * x.hw(), T = x.hw();
        DSD_CHECK(pw.numel == (int64_t)cout * K, "linear %s: weight has %lld elements, graph expects %dx%d", name.c_str(),
                  (long long)pw.numel, cout, K);
        Gemm16Args a;
        a.w = w16(name + ".weight");
        a.bias = W(name + ".bias");
        a.M = M; a.N = cout; a.K = K; a.ldx = K; a.bf16 = hd->precision == PREC_BF16; a.epi = epi;
        a.qcols = qcols; a.qscale = qscale;
        Tn y;
        if (epi == EPI16_GATED) {
            DSD_CHECK(xres && xres->esz == 4 && xres->c == cout && xres->n * xres->hw() == M, "linear16 %s: residual stream mismatch", name.c_str());
            y = *xres;
            a.ldx32 = cout; a.gate_stride = gate_stride; a.T = T;
        } else {
            y = alloc(x.n, x.h, x.w, cout, 2);
            a.ldy = cout;
        }
        const size_t xoff = x.off, yoff = y.off;
        dsd_handle* h = hd;
        const double fl = 2.0 * M * (double)cout * K;
        plan.flops += fl;
        op([=](hipStream_t s) {
            Gemm16Args c = a;
            c.x = h->arena + xoff;
            if (epi == EPI16_GATED) {
                c.x32 = reinterpret_cast<float*>(h->arena + yoff);
                c.gate = reinterpret_cast<const float*>(h->arena + gate_off) + gate_col;
            } else {
                c.y16 = h->arena + yoff;
            }
            gemm16(c, s);
        }, 1, epi == EPI16_GATED ? "gemm16_gated" : (epi == EPI16_GELU ? "gemm16_gelu" : "gemm16"), fl,
           2.0 * ((double)M * K + (double)cout * K) + (epi == EPI16_GATED ? 8.0 : 2.0) * M * cout);
        return y;
    }
    void release(Tn& t) {
        if (t.valid()) free_bytes(t.off, t.bytes());
        t.off = (size_t)-1;
        for (auto& st : t.st) {   // the statistics that travelled with the tensor die with it
            if (st.valid()) free_bytes(st.off, st.bytes(t.n));
            st = StatRef{};
        }
    }
    size_t alloc_raw(size_t bytes) { return ar.alloc(bytes); }
    void release_raw(size_t off, size_t bytes) { free_bytes(off, bytes); }
    void op(std::function<void(hipStream_t)> f, int launches = 1, const std::string& kind = "misc", double flops = 0.0,
            double bytes = 0.0) {
        plan.ops.push_back(std::move(f));
        plan.op_name.push_back(cur_name);
        plan.op_lane.push_back((signed char)(parallel ? lane : -1));
        plan.launches += launches;
        int k = -1;
        for (size_t i = 0; i < plan.kind_names.size(); ++i)
            if (plan.kind_names[i] == kind) k = (int)i;
        if (k < 0) {
            k = (int)plan.kind_names.size();
            plan.kind_names.push_back(kind);
        }
        plan.op_kind.push_back(k);
        plan.op_flops.push_back(flops);
        plan.op_bytes.push_back(bytes);
    }
    std::string cur_name;   // the last parameter looked up: names the ops emitted next (profiles only)
    float* W(const std::string& n) {
        cur_name = n;
        return hd->P(n);
    }
    bool has(const std::string& n) { return hd->pidx.count(n) != 0; }

    // ---- convolution: src is an arena tensor, or (plane >= 0) one of the caller's input planes
    // dst != nullptr: the result is written into channels [0, cout) of the wider NHWC tensor *dst (row stride dst->c) instead
    // of a tensor of its own — the decoder's cat([h, skip]) without the copy of h; the returned Tn is then a view of *dst.
    Tn conv(const std::string& name, const Tn& x, int cout, int ks, int stride = 1, bool ups = false,
            const EmbRef* emb = nullptr, const Tn* res = nullptr, int plane = -1, bool to_out = false, bool bias = true,
            const Tn* dst = nullptr, bool want_stats = false, int pad_lo = -1, int pad_total = -1, const GnRef* gn = nullptr) {
        ConvArgs a;
        a.N = x.n; a.H = x.h; a.W = x.w; a.Cin = x.c; a.Cout = cout; a.ks = ks; a.stride = stride; a.ups = ups ? 1 : 0;
        a.pad_lo = pad_lo; a.pad_total = pad_total;
        {
            int oh_, ow_;
            conv_out_hw(a, &oh_, &ow_);
            a.lanes = plan_lanes((int64_t)x.n * oh_ * ow_);
        }
        a.w = W(name + ".weight");
        a.bias = bias ? W(name + ".bias") : nullptr;
        const Param& pw = hd->PP(name + ".weight");
        DSD_CHECK(pw.numel == (int64_t)cout * x.c * ks * ks, "conv %s: weight has %lld elements, graph expects %dx%dx%dx%d",
                  name.c_str(), (long long)pw.numel, cout, x.c, ks, ks);
        int OH, OW;
        conv_out_hw(a, &OH, &OW);
        Tn y;
        if (dst) {
            DSD_CHECK(!to_out && dst->n == x.n && dst->h == OH && dst->w == OW && dst->c >= cout && dst->c % 4 == 0,
                      "conv %s: destination view does not fit", name.c_str());
            y = *dst;
            y.c = cout;
            y.st[0] = y.st[1] = StatRef{};
        } else if (!to_out) {
            y = alloc(x.n, OH, OW, cout);
        }
        const int y_ld = dst ? dst->c : 0;
        if (res) DSD_CHECK(res->n == x.n && res->h == OH && res->w == OW && res->c == cout, "conv %s: residual shape mismatch", name.c_str());
        const int prec = conv_prec();
        if (prec != PREC_F32 && x.c % 32 == 0 && plane < 0) {   // split-bf16 arithmetic: pieces of the (packed) weight
            const bool f16 = prec == PREC_F16X3;
            const std::string key = f16 ? name + "#f16" : name;
            auto it = hd->wsplit.find(key);
            if (it == hd->wsplit.end()) {
                void* planes = nullptr;
                DSD_HIP(hipMalloc(&planes, (size_t)pw.numel * 2 * 3));
                if (f16 && !hd->ovf) {
                    DSD_HIP(hipMalloc((void**)&hd->ovf, sizeof(int)));
                    DSD_HIP(hipMemset(hd->ovf, 0, sizeof(int)));
                }
                // on the caller's stream: ordered after the dsd_set_param copies / repacks enqueued there
                split_weights(a.w, pw.numel, 3, planes, ps, f16, hd->ovf);
                split_any = true;
                it = hd->wsplit.emplace(key, planes).first;
                hd->wsplit_bytes[key] = (size_t)pw.numel * 2 * 3;
            }
            a.w_split = it->second;
            a.precision = prec;
            a.ovf = f16 ? hd->ovf : nullptr;
            // 3x3 stride-1 layers with enough tiles: F(2,3)-along-W kernel (conv_wino.hip), transformed weights packed once
            if (hd->use_winograd && conv2d_wino_worthwhile(a)) {
                const std::string wkey = name + "#wino";
                auto wi = hd->wsplit.find(wkey);
                if (wi == hd->wsplit.end()) {
                    void* packed = nullptr;
                    const size_t wb = wino_packed_bytes(cout, x.c);
                    DSD_HIP(hipMalloc(&packed, wb));
                    wino_pack_weights(a.w, cout, x.c, packed, ps);
                    split_any = true;
                    wi = hd->wsplit.emplace(wkey, packed).first;
                    hd->wsplit_bytes[wkey] = wb;
                }
                a.w_wino = wi->second;
            }
        }
        if (gn) {   // the kernel applies GroupNorm + SiLU to x itself (the caller asked can_fuse_gn first)
            a.gn_scale = a.gn_shift = reinterpret_cast<const float*>(this);   // non-null placeholders for the plan-time queries
            DSD_CHECK(gn->act == ACT_SILU && conv2d_fuses_gn(a), "conv %s: cannot take the GroupNorm of its input", name.c_str());
        }
        const size_t gsc = gn ? gn->scoff : 0, gsh = gn ? gn->shoff : 0;
        const bool has_gn = gn != nullptr;
        const size_t skb = conv2d_scratch_bytes(a);            // split-K partial tiles of the small-grid layers
        const size_t skoff = skb ? alloc_raw(skb) : 0;
        // GroupNorm statistics of the output from the epilogue, when the kernel this problem gets can emit them
        size_t stoff = 0;
        int stchunks = 0;
        if (want_stats && !to_out && hd->fuse_gn_stats) {
            ConvArgs q = a;
            q.x_bs = -1;
            stchunks = conv2d_stats_chunks(q);
            if (stchunks > 0) {
                StatRef sr;
                sr.chunks = stchunks; sr.c0 = 0; sr.c = cout;
                sr.off = stoff = alloc_raw(sr.bytes(x.n));
                y.st[0] = sr;
            }
        }
        const size_t xoff = x.off, yoff = y.off, roff = res ? res->off : 0;
        const bool has_res = res != nullptr;
        EmbRef e;
        if (emb) e = *emb;
        dsd_handle* h = hd;
        const bool nchw = to_out && cout > 1;
        plan.flops += conv2d_flops(a);
        op([=](hipStream_t s) {
            ConvArgs c = a;
            if (plane >= 0) {
                c.x = h->io.plane[plane];
                c.x_bs = h->io.plane_bs[plane];
            } else {
                c.x = reinterpret_cast<const float*>(h->arena + xoff);
                c.x_bs = -1;
            }
            c.y = to_out ? h->io.out : reinterpret_cast<float*>(h->arena + yoff);
            c.y_ld = y_ld;
            c.out_nchw = nchw ? 1 : 0;
            if (e.valid) {
                c.emb = reinterpret_cast<const float*>(h->arena + e.arena_off) + e.col;
                c.emb_stride = e.stride;
            }
            if (has_res) c.res = reinterpret_cast<const float*>(h->arena + roff);
            if (skb) {
                c.scratch = reinterpret_cast<float*>(h->arena + skoff);
                c.scratch_bytes = skb;
            }
            if (stchunks > 0) {
                c.stats = reinterpret_cast<double*>(h->arena + stoff);
                c.stats_chunks = stchunks;
            }
            if (has_gn) {
                c.gn_scale = reinterpret_cast<const float*>(h->arena + gsc);
                c.gn_shift = reinterpret_cast<const float*>(h->arena + gsh);
            }
            conv2d(c, s);
        }, skb ? 2 : 1, conv2d_variant(a), conv2d_flops(a),
           4.0 * ((double)x.n * x.h * x.w * x.c + (double)cout * x.c * ks * ks + (double)x.n * OH * OW * cout * (has_res ? 2 : 1)));
        if (skb) release_raw(skoff, skb);
        return y;
    }

    // ---- GroupNorm32 (+SiLU, + FiLM) : stats, finalize, apply
    Tn gn_act(const std::string& name, const Tn& x, int act, float eps = 1e-5f, const EmbRef* film = nullptr) {
        const int HW = x.hw(), C = x.c, N = x.n;
        if (hd->fuse_gn_stats && gn_small_ok(HW, C)) {   // small map: statistics + finalize + apply in ONE launch
            Tn y = alloc(x.n, x.h, x.w, x.c);
            const float* gamma = W(name + ".weight");
            const float* beta = W(name + ".bias");
            DSD_CHECK(hd->PP(name + ".weight").numel == C, "norm %s: %lld channels, graph expects %d", name.c_str(),
                      (long long)hd->PP(name + ".weight").numel, C);
            const size_t xoff = x.off, yoff = y.off;
            EmbRef e;
            if (film) e = *film;
            dsd_handle* h = hd;
            op([=](hipStream_t s) {
                const float* fp = e.valid ? reinterpret_cast<const float*>(h->arena + e.arena_off) + e.col : nullptr;
                gn_small(reinterpret_cast<const float*>(h->arena + xoff), N, HW, C, gamma, beta, eps, fp, e.stride, act,
                         reinterpret_cast<float*>(h->arena + yoff), s);
            }, 1, act == ACT_SILU ? "gn_silu_small" : "gn_small", 0.0, 12.0 * N * HW * C);
            return y;
        }
        GnRef g = gn_prepare(name, x, act, eps, film);
        Tn y = gn_apply(g);
        gn_release(g);
        return y;
    }

    // statistics (those that came with the tensor — convolution epilogue / concat kernel — else a pass of our own) + finalize:
    // the per-(sample, channel) scale / shift of the normalisation stay allocated until gn_release()
    GnRef gn_prepare(const std::string& name, const Tn& x, int act, float eps = 1e-5f, const EmbRef* film = nullptr) {
        const int HW = x.hw(), C = x.c, N = x.n;
        const bool have = x.st[0].valid() && x.st[0].c0 == 0 &&
                          (x.st[0].c == C || (x.st[1].valid() && x.st[1].c0 == x.st[0].c && x.st[0].c + x.st[1].c == C));
        const StatRef r0 = x.st[0], r1 = (have && x.st[0].c < C) ? x.st[1] : StatRef{};
        const int nchunk = gn_nchunks(HW, C);
        const size_t pbytes = have ? 0 : (size_t)N * nchunk * C * 2 * sizeof(double);
        const size_t sbytes = (size_t)N * C * sizeof(float);
        const size_t poff = have ? 0 : alloc_raw(pbytes), scoff = alloc_raw(sbytes), shoff = alloc_raw(sbytes);
        const float* gamma = W(name + ".weight");
        const float* beta = W(name + ".bias");
        DSD_CHECK(hd->PP(name + ".weight").numel == C, "norm %s: %lld channels, graph expects %d", name.c_str(),
                  (long long)hd->PP(name + ".weight").numel, C);
        const size_t xoff = x.off;
        EmbRef e;
        if (film) e = *film;
        dsd_handle* h = hd;
        const double tbytes = 4.0 * N * HW * C;
        if (!have)
            op([=](hipStream_t s) {
                gn_stats(reinterpret_cast<const float*>(h->arena + xoff), N, HW, C, reinterpret_cast<double*>(h->arena + poff), nchunk, s);
            }, 1, "gn_stats", 0.0, tbytes);
        op([=](hipStream_t s) {
            const float* fp = e.valid ? reinterpret_cast<const float*>(h->arena + e.arena_off) + e.col : nullptr;
            GnSrc s0, s1;
            if (have) {
                s0.p = reinterpret_cast<const double*>(h->arena + r0.off); s0.chunks = r0.chunks; s0.c0 = 0; s0.c = r0.c;
                if (r1.valid()) {
                    s1.p = reinterpret_cast<const double*>(h->arena + r1.off); s1.chunks = r1.chunks; s1.c0 = r1.c0; s1.c = r1.c;
                }
            } else {
                s0.p = reinterpret_cast<const double*>(h->arena + poff); s0.chunks = nchunk; s0.c0 = 0; s0.c = C;
            }
            gn_finalize(s0, s1, N, HW, C, gamma, beta, eps, fp, e.stride,
                        reinterpret_cast<float*>(h->arena + scoff), reinterpret_cast<float*>(h->arena + shoff), s);
        }, 1, "gn_finalize");
        if (!have) release_raw(poff, pbytes);
        GnRef g;
        g.x = x; g.scoff = scoff; g.shoff = shoff; g.sbytes = sbytes; g.act = act;
        return g;
    }
    // the apply pass: y = act(x * scale + shift) as a tensor of its own (one read + one write through HBM)
    Tn gn_apply(const GnRef& g) {
        const Tn& x = g.x;
        const int HW = x.hw(), C = x.c, N = x.n, act = g.act;
        Tn y = alloc(x.n, x.h, x.w, x.c);
        const size_t xoff = x.off, yoff = y.off, scoff = g.scoff, shoff = g.shoff;
        dsd_handle* h = hd;
        op([=](hipStream_t s) {
            affine_act(reinterpret_cast<const float*>(h->arena + xoff), N, HW, C, reinterpret_cast<float*>(h->arena + scoff),
                       reinterpret_cast<float*>(h->arena + shoff), act, reinterpret_cast<float*>(h->arena + yoff), s);
        }, 1, act == ACT_SILU ? "gn_silu_apply" : "gn_apply", 0.0, 8.0 * N * HW * C);
        return y;
    }
    void gn_release(const GnRef& g) {
        release_raw(g.scoff, g.sbytes);
        release_raw(g.shoff, g.sbytes);
    }
    // out = Conv3x3(SiLU(GroupNorm(x))) with one output channel written to the caller's buffer: one memory-bound pass
    // (conv_out1.hip) when the shape allows and the GroupNorm fusion is on, else normalisation pass + generic convolution
    void out_conv1(const std::string& norm, const std::string& cv, const Tn& x, int cout) {
        if (!(hd->fuse_gn_apply && conv_out1_ok(x.c, cout, 3, 1) && !gn_small_ok(x.hw(), x.c))) {
            Tn a = gn_act(norm, x, ACT_SILU);
            conv(cv, a, cout, 3, 1, false, nullptr, nullptr, -1, /*to_out=*/true);
            release(a);
            return;
        }
        const Param& pw = hd->PP(cv + ".weight");
        DSD_CHECK(pw.numel == (int64_t)9 * x.c, "conv %s: weight has %lld elements, graph expects 1x%dx3x3", cv.c_str(),
                  (long long)pw.numel, x.c);
        GnRef g = gn_prepare(norm, x, ACT_SILU);
        const float* w = W(cv + ".weight");
        const float* bias = W(cv + ".bias");
        const size_t xoff = x.off, scoff = g.scoff, shoff = g.shoff;
        const int N = x.n, H = x.h, Wd = x.w, C = x.c;
        dsd_handle* h = hd;
        const double fl = 2.0 * N * H * Wd * 9.0 * C;
        plan.flops += fl;
        op([=](hipStream_t s) {
            ConvOut1Args a;
            a.x = reinterpret_cast<const float*>(h->arena + xoff);
            a.N = N; a.H = H; a.W = Wd; a.C = C;
            a.scale = reinterpret_cast<const float*>(h->arena + scoff);
            a.shift = reinterpret_cast<const float*>(h->arena + shoff);
            a.w = w;
            a.bias = bias;
            a.y = h->io.out;
            conv_out1(a, s);
        }, 1, "gn_silu_conv_out1", fl, 4.0 * N * H * Wd * (C + 1.0));
        gn_release(g);
    }

    // would conv(name, x, cout, 3) apply the GroupNorm + SiLU of its input by itself?  (the tap-reuse kernel, bf16x6 only)
    bool can_fuse_gn(const Tn& x, int cout) {
        if (!hd->fuse_gn_apply || conv_prec() != PREC_BF16X6 || x.c % 32 != 0) return false;
        ConvArgs a;
        a.N = x.n; a.H = x.h; a.W = x.w; a.Cin = x.c; a.Cout = cout; a.ks = 3; a.stride = 1;
        a.precision = PREC_BF16X6;
        a.lanes = plan_lanes((int64_t)x.n * x.h * x.w);
        a.w_split = this;   // (non-null: only the shape matters here)
        if (hd->use_winograd && conv2d_wino_worthwhile(a)) return false;
        return conv2d_fuses_gn(a);
    }

    Tn resample(const Tn& x, bool up) {
        Tn y = up ? alloc(x.n, x.h * 2, x.w * 2, x.c) : alloc(x.n, x.h / 2, x.w / 2, x.c);
        const size_t xoff = x.off, yoff = y.off;
        dsd_handle* h = hd;
        const Tn xx = x;
        op([=](hipStream_t s) {
            const float* xp = reinterpret_cast<const float*>(h->arena + xoff);
            float* yp = reinterpret_cast<float*>(h->arena + yoff);
            if (up) upsample2(xp, xx.n, xx.h, xx.w, xx.c, yp, s); else avgpool2(xp, xx.n, xx.h, xx.w, xx.c, yp, s);
        });
        return y;
    }

    // ResBlock._forward, openaimodel.py:264-284.  x stays owned by the caller.
    Tn res_block(const std::string& p, const Tn& x, int cin, int cout, bool up, bool down, const EmbRef& emb,
                 const Tn* dst = nullptr) {
        DSD_CHECK(x.c == cin, "ResBlock %s: input has %d channels, expected %d", p.c_str(), x.c, cin);
        const bool film = hd->PP(pre(p, "emb_layers.1.weight")).shape[0] == 2 * cout;
        // GroupNorm + SiLU in front of a 3x3 convolution: the large layers apply it while they stage their input (no apply pass)
        Tn xs = x;
        bool xs_owned = false;
        Tn h;
        const bool gn_small_in = hd->fuse_gn_stats && gn_small_ok(x.hw(), x.c);
        if (!up && !down && !gn_small_in && can_fuse_gn(x, cout)) {
            GnRef g = gn_prepare(pre(p, "in_layers.0"), x, ACT_SILU);
            h = conv(pre(p, "in_layers.2"), x, cout, 3, 1, false, film ? nullptr : &emb, nullptr, -1, false, true, nullptr,
                     /*want_stats=*/true, -1, -1, &g);
            gn_release(g);
        } else {
            Tn a = gn_act(pre(p, "in_layers.0"), x, ACT_SILU);
            if (up || down) {
                Tn a2 = resample(a, up);
                release(a);
                a = a2;
                xs = resample(x, up);
                xs_owned = true;
            }
            h = conv(pre(p, "in_layers.2"), a, cout, 3, 1, false, film ? nullptr : &emb, nullptr, -1, false, true, nullptr,
                     /*want_stats=*/true);   // out_layers.0 normalises exactly this tensor
            release(a);
        }
        Tn skip = xs;
        bool skip_owned = false;
        if (cin != cout) {
            skip = conv(pre(p, "skip_connection"), xs, cout, 1);
            skip_owned = true;
        }
        Tn out;
        const bool gn_small_out = hd->fuse_gn_stats && gn_small_ok(h.hw(), h.c);
        if (!gn_small_out && can_fuse_gn(h, cout)) {
            GnRef g = gn_prepare(pre(p, "out_layers.0"), h, ACT_SILU, 1e-5f, film ? &emb : nullptr);
            out = conv(pre(p, "out_layers.3"), h, cout, 3, 1, false, nullptr, &skip, -1, false, true, dst, true, -1, -1, &g);
            gn_release(g);
            release(h);
        } else {
            Tn a3 = gn_act(pre(p, "out_layers.0"), h, ACT_SILU, 1e-5f, film ? &emb : nullptr);
            release(h);
            out = conv(pre(p, "out_layers.3"), a3, cout, 3, 1, false, nullptr, &skip, -1, false, true, dst, true);
            release(a3);
        }
        if (skip_owned) release(skip);
        if (xs_owned) release(xs);
        return out;
    }

    // AttentionBlock._forward, openaimodel.py:467-473
    Tn attn_block(const std::string& p, const Tn& x, int heads, bool new_order, const Tn* dst = nullptr) {
        const int C = x.c, T = x.hw();
        DSD_CHECK(C % heads == 0, "attention %s: %d channels not divisible by %d heads", p.c_str(), C, heads);
        const int d = C / heads;
        Tn n = gn_act(pre(p, "norm"), x, ACT_NONE);
        Tn qkv = conv(pre(p, "qkv"), n, 3 * C, 1);
        release(n);
        Tn a = alloc(x.n, x.h, x.w, C);
        {
            AttnArgs aa;
            aa.N = x.n; aa.Tq = T; aa.Tk = T; aa.heads = heads; aa.d = d;
            aa.ldq = aa.ldk = aa.ldv = 3 * C; aa.ldo = C;
            aa.split = hd->precision != PREC_F32;
            const float scale = 1.f / std::sqrt(std::sqrt((float)d));  // openaimodel.py:547
            aa.scale_q = aa.scale_k = scale;
            int qo, ko, vo;
            if (new_order) {  // QKVAttention: qkv.chunk(3) then heads
                qo = 0; ko = C; vo = 2 * C;
                aa.q_hs = aa.k_hs = aa.v_hs = d;
            } else {          // QKVAttentionLegacy: heads then split(ch)
                qo = 0; ko = d; vo = 2 * d;
                aa.q_hs = aa.k_hs = aa.v_hs = 3 * d;
            }
            const size_t qoff = qkv.off, aoff = a.off;
            dsd_handle* h = hd;
            plan.flops += 4.0 * x.n * heads * (double)T * T * d;
            op([=](hipStream_t s) {
                AttnArgs r = aa;
                const float* base = reinterpret_cast<const float*>(h->arena + qoff);
                r.q = base + qo; r.k = base + ko; r.v = base + vo;
                r.out = reinterpret_cast<float*>(h->arena + aoff);
                attention(r, s);
            }, 1, "attention", 4.0 * x.n * heads * (double)T * T * d);
        }
        release(qkv);
        Tn out = conv(pre(p, "proj_out"), a, C, 1, 1, false, nullptr, &x, -1, false, true, dst, true);
        release(a);
        return out;
    }

    // FeatureDisentangle.forward, model.py:165-168
    Tn disentangle(const std::string& p, const Tn& x, int half) {
        Tn a = gn_act(pre(p, "conv_1.0"), x, ACT_SILU);
        Tn o = conv(pre(p, "conv_1.2"), a, x.c, 3, 1, false, nullptr, &x, -1, false, true, nullptr, true);
        release(a);
        Tn a2 = gn_act(pre(p, "conv_2.0"), o, ACT_SILU);
        release(o);
        Tn y = conv(pre(p, "conv_2.2"), a2, half, 1);
        release(a2);
        return y;
    }

    Tn se(const std::string& p, const Tn& x) {
        Tn y = alloc(x.n, x.h, x.w, x.c);
        const float* w1 = W(pre(p, "se.0.weight"));
        const float* w2 = W(pre(p, "se.2.weight"));
        const int Cr = (int)hd->PP(pre(p, "se.0.weight")).shape[0];
        const size_t xoff = x.off, yoff = y.off;
        const Tn xx = x;
        dsd_handle* h = hd;
        op([=](hipStream_t s) {
            se_scale(reinterpret_cast<const float*>(h->arena + xoff), xx.n, xx.hw(), xx.c, w1, w2, Cr,
                     reinterpret_cast<float*>(h->arena + yoff), s);
        });
        return y;
    }

    // dst[:, coff:coff+C] = act((a+b+c+d)/div)
    // want_stats: also emit the GroupNorm statistics of the written channels (they become dst.st[coff ? 1 : 0])
    void avg(const Tn* srcs, int nsrc, float div, Tn& dst, int coff, int act, bool want_stats = false) {
        size_t o[4] = {0, 0, 0, 0};
        int bmask = 0;
        for (int i = 0; i < nsrc; ++i) {
            o[i] = srcs[i].off;
            DSD_CHECK(srcs[i].n == dst.n || srcs[i].n == 1, "avg: batch mismatch");
            if (srcs[i].n == 1 && dst.n > 1) bmask |= 1 << i;   // one shared sample, broadcast over the batch
        }
        const int C = srcs[0].c;
        const int64_t pixels = (int64_t)dst.n * dst.hw();
        const int64_t per_sample = dst.hw();
        const size_t doff = dst.off;
        const int dstC = dst.c;
        dsd_handle* h = hd;
        if (want_stats && hd->fuse_gn_stats && C % 32 == 0 && (coff == 0 || dst.st[0].valid())) {
            StatRef sr;
            sr.chunks = gn_nchunks((int)per_sample, C); sr.c0 = coff; sr.c = C;
            sr.off = alloc_raw(sr.bytes(dst.n));
            dst.st[coff ? 1 : 0] = sr;
            const int N = dst.n, HW = (int)per_sample, nchunk = sr.chunks;
            const size_t soff = sr.off;
            op([=](hipStream_t s) {
                const float* p[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int i = 0; i < nsrc; ++i) p[i] = reinterpret_cast<const float*>(h->arena + o[i]);
                avg_into_stats(p[0], p[1], p[2], p[3], div, N, HW, C, reinterpret_cast<float*>(h->arena + doff), dstC, coff, act,
                               bmask, reinterpret_cast<double*>(h->arena + soff), nchunk, s);
            }, 1, nsrc == 4 ? "skip_avg4_concat+stats" : "concat_copy+stats", 0.0, 4.0 * pixels * C * (nsrc + 1));
            return;
        }
        op([=](hipStream_t s) {
            const float* p[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int i = 0; i < nsrc; ++i) p[i] = reinterpret_cast<const float*>(h->arena + o[i]);
            avg_into(p[0], p[1], p[2], p[3], div, pixels, C, reinterpret_cast<float*>(h->arena + doff), dstC, coff, act, s,
                     per_sample, bmask);
        }, 1, nsrc == 4 ? "skip_avg4_concat" : "concat_copy", 0.0, 4.0 * pixels * C * (nsrc + 1));
    }

    // TimestepEmbedSequential.forward, openaimodel.py:80-90.  Consumes (releases) x unless keep_input.
    Tn block(const std::string& prefix, const std::vector<Layer>& layers, Tn x, bool keep_input,
             const std::vector<EmbRef>& embs, size_t& emb_i, int plane = -1, const Tn* dst = nullptr) {
        Tn cur = x;
        bool cur_owned = !keep_input;
        for (size_t li = 0; li < layers.size(); ++li) {
            const Layer& L = layers[li];
            const std::string nm = prefix + "." + std::to_string(li);
            Tn nxt;
            const Tn* d = li + 1 == layers.size() ? dst : nullptr;   // only the block's last layer writes into the view
            switch (L.kind) {
                case L_CONV: nxt = conv(nm, cur, L.cout, 3, 1, false, nullptr, nullptr, plane, false, true, d, true); break;
                case L_RES: nxt = res_block(nm, cur, L.cin, L.cout, L.up, L.down, embs.at(emb_i++), d); break;
                case L_ATTN:
                    nxt = L.dh > 0 ? spatial_transformer(nm + ".", cur, L.heads, L.dh, 1, &st_ctx, d)
                                   : attn_block(nm, cur, L.heads, hd->cfg.use_new_attention_order != 0, d);
                    break;
                case L_DOWN: nxt = conv(nm + ".op", cur, L.ch, 3, 2, false, nullptr, nullptr, -1, false, true, d, true); break;
                case L_UP: nxt = conv(nm + ".conv", cur, L.ch, 3, 1, true, nullptr, nullptr, -1, false, true, d, true); break;
            }
            if (cur_owned && plane < 0) release(cur);
            plane = -1;
            cur = nxt;
            cur_owned = true;
        }
        return cur;
    }

    // ---------------------------------------------------------------- token blocks (cross-attention variant)
    Tn linear_tok(const std::string& name, const Tn& x, int cout, bool bias, const Tn* res = nullptr) {
        return conv(name, x, cout, 1, 1, false, nullptr, res, -1, false, bias);
    }
    Tn lnorm(const std::string& name, const Tn& x) {
        Tn y = alloc(x.n, x.h, x.w, x.c);
        const float* g = W(name + ".weight");
        const float* b = W(name + ".bias");
        const size_t xoff = x.off, yoff = y.off;
        const int64_t rows = (int64_t)x.n * x.hw();
        const int C = x.c;
        dsd_handle* h = hd;
        op([=](hipStream_t s) {
            layer_norm(reinterpret_cast<const float*>(h->arena + xoff), rows, C, g, b, 1e-5f,
                       reinterpret_cast<float*>(h->arena + yoff), s);
        });
        return y;
    }
    // CrossAttention.forward (attention.py:164-193); ctx == nullptr -> self attention.  + res if given.
    Tn xattn(const std::string& p, const Tn& x, const Tn* ctx, int heads, const Tn* res) {
        const Tn& c = ctx ? *ctx : x;
        const int inner = (int)hd->PP(pre(p, "to_q.weight")).shape[0];
        const int d = inner / heads;
        Tn q = linear_tok(pre(p, "to_q"), x, inner, false);
        Tn k = linear_tok(pre(p, "to_k"), c, inner, false);
        Tn v = linear_tok(pre(p, "to_v"), c, inner, false);
        Tn a = alloc(x.n, x.h, x.w, inner);
        AttnArgs aa;
        aa.N = x.n; aa.Tq = x.hw(); aa.Tk = c.hw(); aa.heads = heads; aa.d = d;
        aa.ldq = aa.ldk = aa.ldv = aa.ldo = inner;
        aa.q_hs = aa.k_hs = aa.v_hs = d;
        aa.split = hd->precision != PREC_F32;
        aa.scale_s = 1.f / std::sqrt((float)d);
        const size_t qo = q.off, ko = k.off, vo = v.off, ao = a.off;
        dsd_handle* h = hd;
        plan.flops += 4.0 * x.n * heads * (double)aa.Tq * aa.Tk * d;
        op([=](hipStream_t s) {
            AttnArgs r = aa;
            r.q = reinterpret_cast<const float*>(h->arena + qo);
            r.k = reinterpret_cast<const float*>(h->arena + ko);
            r.v = reinterpret_cast<const float*>(h->arena + vo);
            r.out = reinterpret_cast<float*>(h->arena + ao);
            attention(r, s);
        });
        release(q); release(k); release(v);
        Tn y = linear_tok(pre(p, "to_out.0"), a, x.c, true, res);
        release(a);
        return y;
    }
    Tn ff_geglu(const std::string& p, const Tn& x, const Tn* res) {
        const int inner2 = (int)hd->PP(pre(p, "net.0.proj.weight")).shape[0];
        Tn hh = linear_tok(pre(p, "net.0.proj"), x, inner2, true);
        Tn g = alloc(x.n, x.h, x.w, inner2 / 2);
        const size_t ho = hh.off, go = g.off;
        const int64_t rows = (int64_t)x.n * x.hw();
        dsd_handle* h = hd;
        op([=](hipStream_t s) {
            geglu(reinterpret_cast<const float*>(h->arena + ho), rows, inner2 / 2, reinterpret_cast<float*>(h->arena + go), s);
        });
        release(hh);
        Tn y = linear_tok(pre(p, "net.2"), g, x.c, true, res);
        release(g);
        return y;
    }
    // BasicTransformerBlock._forward (attention.py:326-330); x stays owned by the caller
    // SpatialTransformer.forward (attention.py:409-428): x + proj_out(blocks(proj_in(GroupNorm(x)))); in NHWC the conv1x1 and
    // Linear flavours of proj_in / proj_out are the same GEMM.  ctx[d] = context of block d (invalid Tn: self-attention).
    Tn st_ctx;   // the context every SpatialTransformer of a UNetModel receives (openaimodel.py:946-952)
    Tn spatial_transformer(const std::string& p, const Tn& x, int heads, int dh, int depth, const Tn* ctx, const Tn* dst = nullptr) {
        Tn n = gn_act(p + "norm", x, ACT_NONE, 1e-6f);
        Tn t = conv(p + "proj_in", n, heads * dh, 1);
        release(n);
        for (int d = 0; d < depth; ++d) {
            Tn t2 = btb(p + "transformer_blocks." + std::to_string(d), t, ctx[d].valid() ? &ctx[d] : nullptr, heads);
            release(t);
            t = t2;
        }
        Tn y = conv(p + "proj_out", t, x.c, 1, 1, false, nullptr, &x, -1, false, true, dst, dst != nullptr);
        release(t);
        return y;
    }
    Tn btb(const std::string& p, const Tn& x, const Tn* ctx, int heads) {
        Tn n1 = lnorm(pre(p, "norm1"), x);
        Tn x1 = xattn(pre(p, "attn1"), n1, nullptr, heads, &x);
        release(n1);
        Tn n2 = lnorm(pre(p, "norm2"), x1);
        Tn x2 = xattn(pre(p, "attn2"), n2, ctx, heads, &x1);
        release(n2); release(x1);
        Tn n3 = lnorm(pre(p, "norm3"), x2);
        Tn x3 = ff_geglu(pre(p, "ff"), n3, &x2);
        release(n3); release(x2);
        return x3;
    }

    // ---------------------------------------------------------------- KL-VAE blocks (ldm/modules/diffusionmodules/model.py)
    // ResnetBlock.forward with temb = None (:121-149); GroupNorm eps 1e-6 (Normalize :41-42).  x stays owned by the caller.
    Tn vae_res(const std::string& p, const Tn& x, int cout) {
        // (as in res_block: where the tap-reuse kernel takes the 3x3 convolution — 128- and 160-column tiles — it applies the
        // GroupNorm + swish of its input itself and the apply pass disappears)
        Tn h;
        if (can_fuse_gn(x, cout)) {
            GnRef g = gn_prepare(p + ".norm1", x, ACT_SILU, 1e-6f);
            h = conv(p + ".conv1", x, cout, 3, 1, false, nullptr, nullptr, -1, false, true, nullptr, true, -1, -1, &g);
            gn_release(g);
        } else {
            Tn a = gn_act(p + ".norm1", x, ACT_SILU, 1e-6f);
            h = conv(p + ".conv1", a, cout, 3, 1, false, nullptr, nullptr, -1, false, true, nullptr, true);
            release(a);
        }
        Tn skip = x;
        const bool proj = x.c != cout;
        if (proj) skip = conv(p + ".nin_shortcut", x, cout, 1);
        Tn out;
        if (can_fuse_gn(h, cout)) {
            GnRef g = gn_prepare(p + ".norm2", h, ACT_SILU, 1e-6f);
            out = conv(p + ".conv2", h, cout, 3, 1, false, nullptr, &skip, -1, false, true, nullptr, true, -1, -1, &g);
            gn_release(g);
            release(h);
        } else {
            Tn a2 = gn_act(p + ".norm2", h, ACT_SILU, 1e-6f);
            release(h);
            out = conv(p + ".conv2", a2, cout, 3, 1, false, nullptr, &skip, -1, false, true, nullptr, true);
            release(a2);
        }
        if (proj) release(skip);
        return out;
    }
    // AttnBlock.forward (:185-209): ONE head over all C channels, softmax(q k^T C^-1/2) v.  C is 512 at the yaml's size, far
    // beyond the flash kernel's head dims, so the score matrix is materialised per sample and both products run as GEMMs
    // on the convolution kernels (a 1x1 convolution whose "weights" are the other operand: y[m][n] = sum_k A[m][k] B[n][k]).
    Tn vae_attn(const std::string& p, const Tn& x) {
        const int C = x.c, T = x.hw(), Bn = x.n;
        Tn n = gn_act(p + ".norm", x, ACT_NONE, 1e-6f);
        Tn q = conv(p + ".q", n, C, 1), k = conv(p + ".k", n, C, 1), v = conv(p + ".v", n, C, 1);
        release(n);
        Tn a = alloc(Bn, x.h, x.w, C);
        const bool split = hd->precision != PREC_F32 && C % 32 == 0 && T % 32 == 0;
        const bool f16 = hd->precision == PREC_F16X3;
        const size_t sb = (size_t)T * T * sizeof(float), vb = (size_t)C * T * sizeof(float), pb = split ? (size_t)T * C * 6 : 0;
        const size_t soff = alloc_raw(sb), vtoff = alloc_raw(vb), poff = pb ? alloc_raw(pb) : 0;
        if (split && f16 && !hd->ovf) {
            DSD_HIP(hipMalloc((void**)&hd->ovf, sizeof(int)));
            DSD_HIP(hipMemset(hd->ovf, 0, sizeof(int)));
        }
        const size_t qo = q.off, ko = k.off, vo = v.off, ao = a.off;
        const int prec = hd->precision;
        const float scale = 1.f / std::sqrt((float)C);   // int(c)**(-0.5), :196
        dsd_handle* h = hd;
        const double fl = 4.0 * Bn * (double)T * T * C;
        plan.flops += fl;
        op([=](hipStream_t s) {
            float* S = reinterpret_cast<float*>(h->arena + soff);
            float* vt = reinterpret_cast<float*>(h->arena + vtoff);
            void* planes = pb ? static_cast<void*>(h->arena + poff) : nullptr;
            auto gemm_nt = [&](const float* A, const float* Bm, float* Y, int M, int N, int K) {
                ConvArgs c;
                c.x = A; c.N = 1; c.H = M; c.W = 1; c.Cin = K; c.w = Bm; c.Cout = N; c.ks = 1; c.y = Y;
                if (split) {
                    split_weights(Bm, (int64_t)N * K, 3, planes, s, f16, f16 ? h->ovf : nullptr);
                    c.w_split = planes;
                    c.precision = prec;
                    c.ovf = f16 ? h->ovf : nullptr;
                }
                conv2d(c, s);
            };
            for (int b = 0; b < Bn; ++b) {
                const float* qb = reinterpret_cast<const float*>(h->arena + qo) + (size_t)b * T * C;
                const float* kb = reinterpret_cast<const float*>(h->arena + ko) + (size_t)b * T * C;
                const float* vb_ = reinterpret_cast<const float*>(h->arena + vo) + (size_t)b * T * C;
                float* ab = reinterpret_cast<float*>(h->arena + ao) + (size_t)b * T * C;
                gemm_nt(qb, kb, S, T, T, C);            // w_[i][j] = sum_c q[i][c] k[j][c]                     :194-195
                softmax_rows(S, T, T, scale, s);         // * C^-1/2, softmax over j                             :196-197
                nhwc_to_nchw(vb_, 1, C, T, vt, s);       // v^T [C][T]
                gemm_nt(S, vt, ab, T, C, T);            // h_[i][c] = sum_j w_[i][j] v[j][c]                    :200-203
            }
        }, Bn * (split ? 6 : 4), "vae_attention", fl);
        release_raw(soff, sb);
        release_raw(vtoff, vb);
        if (pb) release_raw(poff, pb);
        release(q); release(k); release(v);
        Tn out = conv(p + ".proj_out", a, C, 1, 1, false, nullptr, &x, -1, false, true, nullptr, true);
        release(a);
        return out;
    }

    // copy an external device buffer (io.*) into the arena, optionally NCHW -> NHWC
    Tn import_ext(int which, int n, int h, int w, int c, bool from_nchw) {
        Tn t = alloc(n, h, w, c);
        const size_t off = t.off;
        dsd_handle* hh = hd;
        op([=](hipStream_t s) {
            const float* src = which == 0 ? hh->io.x_nchw : (which == 1 ? hh->io.aux : hh->io.aux2);
            float* dst = reinterpret_cast<float*>(hh->arena + off);
            if (from_nchw && c > 1 && h * w > 1)
                nchw_to_nhwc(src, n, c, h * w, dst, s);
            else
                DSD_HIP(hipMemcpyAsync(dst, src, (size_t)n * h * w * c * sizeof(float), hipMemcpyDeviceToDevice, s));
        });
        return t;
    }
    void export_out(const Tn& t, bool to_nchw, int feat_idx = -1) {
        const size_t off = t.off;
        const Tn tt = t;
        dsd_handle* hh = hd;
        op([=](hipStream_t s) {
            float* dst = feat_idx >= 0 ? hh->io.feats[feat_idx] : hh->io.out;
            const float* src = reinterpret_cast<const float*>(hh->arena + off);
            if (to_nchw && tt.c > 1 && tt.hw() > 1)
                nhwc_to_nchw(src, tt.n, tt.c, tt.hw(), dst, s);
            else
                DSD_HIP(hipMemcpyAsync(dst, src, tt.bytes(), hipMemcpyDeviceToDevice, s));
        });
    }
};

// --------------------------------------------------------------------------------- DSUnetModel.forward
void build_unet(Builder& b, int H, int W, bool zero_al_l, bool want_feats, bool share) {
    dsd_handle* hd = b.hd;
    const dsd_config& cfg = hd->cfg;
    const Spec sp = build_spec(cfg);
    const int B = b.B;
    const int nds = cfg.n_levels - 1;
    DSD_CHECK(H % (1 << nds) == 0 && W % (1 << nds) == 0, "H=%d, W=%d must be multiples of %d (down/up-sampling + skip concat)", H, W, 1 << nds);
    // share: the al / l streams see the same (all-zero) plane and the same timestep for every slice -> batch of ONE
    DSD_CHECK(!share || (zero_al_l && !want_feats), "zero-stream sharing needs the 2-channel branch and no feature outputs");

    // ---- timestep embedding MLP + all 68 emb_layers as ONE GEMM (model.py:645-646; openaimodel.py:222-228,273)
    const int mc = cfg.model_channels, ted = sp.ted;
    Tn temb = b.alloc(B, 1, 1, mc), e1 = b.alloc(B, 1, 1, ted), emb = b.alloc(B, 1, 1, ted);
    Tn emb_all = b.alloc(B, 1, 1, (int)hd->emb_total);
    {
        const size_t to = temb.off, e1o = e1.off, eo = emb.off, ao = emb_all.off;
        const float *w0 = b.W("time_embed.0.weight"), *b0 = b.W("time_embed.0.bias");
        const float *w2 = b.W("time_embed.2.weight"), *b2 = b.W("time_embed.2.bias");
        const float* wall = reinterpret_cast<const float*>(hd->slab + hd->emb_w_off);
        const float* ball = reinterpret_cast<const float*>(hd->slab + hd->emb_b_off);
        const int etot = (int)hd->emb_total;
        b.plan.flops += 2.0 * B * ((double)mc * ted + (double)ted * ted + (double)ted * etot);
        b.op([=](hipStream_t s) {
            float* tp = reinterpret_cast<float*>(hd->arena + to);
            float* e1p = reinterpret_cast<float*>(hd->arena + e1o);
            float* ep = reinterpret_cast<float*>(hd->arena + eo);
            float* ap = reinterpret_cast<float*>(hd->arena + ao);
            timestep_embedding(hd->io.t, hd->io.t_is_float, B, mc, tp, s, hd->freqs);
            linear(tp, B, mc, mc, w0, b0, ted, ACT_NONE, e1p, ted, s);
            linear(e1p, B, ted, ted, w2, b2, ted, ACT_SILU, ep, ted, s);
            linear(ep, B, ted, ted, wall, ball, etot, ACT_SILU, ap, etot, s);
        }, 4, "time_embed_mlp", 2.0 * B * ((double)mc * ted + (double)ted * ted + (double)ted * etot));
    }
    b.release(temb); b.release(e1); b.release(emb);
    // column of each ResBlock inside emb_all = position of its bias in the contiguous bias region (declaration order)
    std::unordered_map<std::string, int64_t> emb_col;
    {
        int64_t col = 0;
        for (const auto& p : hd->params)
            if (p.region == 2) {
                emb_col[p.name.substr(0, p.name.size() - std::strlen(".emb_layers.1.bias"))] = col;
                col += p.numel;
            }
    }
    auto embs_for = [&](const std::string& prefix, const std::vector<std::vector<Layer>>& blocks) {
        std::vector<std::vector<EmbRef>> out(blocks.size());
        for (size_t bi = 0; bi < blocks.size(); ++bi)
            for (size_t li = 0; li < blocks[bi].size(); ++li)
                if (blocks[bi][li].kind == L_RES) {
                    EmbRef e;
                    e.arena_off = emb_all.off;
                    e.col = emb_col.at(prefix + "." + std::to_string(bi) + "." + std::to_string(li));
                    e.stride = (int)hd->emb_total;
                    e.valid = true;
                    out[bi].push_back(e);
                }
        return out;
    };

    // ---- four encoder streams (model.py:674-686): stream order n, a, al, l ; planes io.plane[0..3]
    const char* sfx[4] = {"", "_a", "_al", "_l"};
    // Emission order: first the blocks whose layers fill the chip, stream after stream; then — from the first block whose
    // input has few enough pixels (dsd_handle::lane_pixels) — the rest of the four encoders as four LANES that the executor
    // runs concurrently (net_launch_ops): small grids of independent chains share the 256 CUs instead of queueing.
    std::vector<Tn> hs[4];
    Tn cur4[4];
    size_t split = sp.input_blocks.size();
    {
        int hh = H, ww = W;
        for (size_t bi = 0; bi < sp.input_blocks.size() && split == sp.input_blocks.size(); ++bi) {
            if (hd->use_lanes && (int64_t)B * hh * ww <= hd->lane_pixels && bi > 0) split = bi;
            for (const Layer& L : sp.input_blocks[bi])
                if (L.kind == L_DOWN || (L.kind == L_RES && L.down)) { hh /= 2; ww /= 2; }
        }
    }
    auto run_blocks = [&](int s, size_t from, size_t to) {
        const std::string base = std::string("input_blocks") + sfx[s];
        auto embs = embs_for(base, sp.input_blocks);
        for (size_t bi = from; bi < to; ++bi) {
            size_t ei = 0;
            Tn nxt = b.block(base + "." + std::to_string(bi), sp.input_blocks[bi], cur4[s], /*keep_input=*/true, embs[bi], ei,
                             bi == 0 ? s : -1);
            hs[s].push_back(nxt);
            cur4[s] = nxt;
        }
    };
    for (int s = 0; s < 4; ++s) {
        cur4[s].n = (share && s >= 2) ? 1 : B; cur4[s].h = H; cur4[s].w = W; cur4[s].c = 1;  // the caller's plane
        run_blocks(s, 0, split);
    }
    if (split < sp.input_blocks.size()) {
        b.begin_parallel();
        for (int s = 0; s < 4; ++s) {
            b.set_lane(s);
            run_blocks(s, split, sp.input_blocks.size());
        }
        b.end_parallel();
    }
    // ---- middle block on the noise stream only (model.py:688)
    Tn h_n;
    {
        std::vector<EmbRef> me;
        for (size_t li = 0; li < sp.middle.size(); ++li)
            if (sp.middle[li].kind == L_RES) {
                EmbRef e;
                e.arena_off = emb_all.off;
                e.col = emb_col.at("middle_block." + std::to_string(li));
                e.stride = (int)hd->emb_total;
                e.valid = true;
                me.push_back(e);
            }
        size_t ei = 0;
        h_n = b.block("middle_block", sp.middle, hs[0].back(), /*keep_input=*/true, me, ei);
    }
    // ---- disentangle heads (model.py:695-725)
    const Tn &h_a = hs[1].back(), &h_al = hs[2].back(), &h_l = hs[3].back();
    const int half = sp.half;
    Tn st[3] = {b.disentangle("conv_style", h_a, half), b.disentangle("conv_style", h_al, half), b.disentangle("conv_style", h_l, half)};
    Tn ct[3] = {b.disentangle("conv_content", h_a, half), b.disentangle("conv_content", h_al, half), b.disentangle("conv_content", h_l, half)};
    Tn an[2] = {b.disentangle("conv_anatomy", h_a, half), b.disentangle("conv_anatomy", h_al, half)};
    Tn le[2] = {b.disentangle("conv_lesion", h_al, half), b.disentangle("conv_lesion", h_l, half)};
    Tn n_style, n_content;
    if (want_feats) {  // dead for sampling: only returned in the dict (model.py:695-696,729)
        n_style = b.disentangle("conv_style", h_n, half);
        n_content = b.disentangle("conv_content", h_n, half);
    }
    auto proj = [&](const std::string& nm, Tn* list, int n) {
        int nb = 1;
        for (int i = 0; i < n; ++i) nb = std::max(nb, list[i].n);
        Tn m = b.alloc(nb, list[0].h, list[0].w, half);
        b.avg(list, n, (float)n, m, 0, ACT_NONE);           // ht.mean(ht.stack(list), dim=0)
        Tn se = b.se(nm + ".0", m);
        b.release(m);
        Tn y = b.conv(nm + ".1", se, half, 3);
        b.release(se);
        return y;
    };
    Tn h_style = proj("style_proj", st, 3);
    Tn h_share = proj("share_content_proj", ct, 3);
    Tn h_anat = proj("anatomy_proj", an, 2);
    Tn h_les = proj("lesion_proj", le, 2);
    if (want_feats) {
        int fi = 0;
        for (int i = 0; i < 3; ++i) b.export_out(st[i], true, fi++);
        for (int i = 0; i < 3; ++i) b.export_out(ct[i], true, fi++);
        for (int i = 0; i < 2; ++i) b.export_out(an[i], true, fi++);
        for (int i = 0; i < 2; ++i) b.export_out(le[i], true, fi++);
        b.export_out(h_style, true, fi++);
        b.export_out(n_style, true, fi++);
        b.export_out(h_share, true, fi++);
        b.export_out(n_content, true, fi++);
        b.release(n_style); b.release(n_content);
    }
    for (auto& t : st) b.release(t);
    for (auto& t : ct) b.release(t);
    for (auto& t : an) b.release(t);
    for (auto& t : le) b.release(t);
    // ---- h = all_proj(cat[h_n, share_content, style, anatomy, lesion])  (model.py:734-738): SiLU fused into the concat
    Tn cat = b.alloc(B, h_n.h, h_n.w, sp.conv_ch + 4 * half);
    b.avg(&h_n, 1, 1.f, cat, 0, ACT_SILU);
    b.avg(&h_share, 1, 1.f, cat, sp.conv_ch, ACT_SILU);
    b.avg(&h_style, 1, 1.f, cat, sp.conv_ch + half, ACT_SILU);
    b.avg(&h_anat, 1, 1.f, cat, sp.conv_ch + 2 * half, ACT_SILU);
    b.avg(&h_les, 1, 1.f, cat, sp.conv_ch + 3 * half, ACT_SILU);
    b.release(h_n); b.release(h_share); b.release(h_style); b.release(h_anat); b.release(h_les);
    Tn h = b.conv("all_proj.1", cat, sp.conv_ch, 1);
    b.release(cat);
    // ---- decoder (model.py:743-746): cat[h, (hs+hs_a+hs_al+hs_l)/4]
    auto dembs = embs_for("output_blocks", sp.output_blocks);
    // The concat buffer of block bi+1 is allocated before block bi runs, and block bi's last convolution writes h straight
    // into its first channels (row stride = concat width): torch.cat([h, skip]) costs no copy of h.
    auto block_out = [&](const std::vector<Layer>& layers, int c, int hh, int ww, int* oc, int* oh, int* ow) {
        *oc = c; *oh = hh; *ow = ww;
        for (const Layer& L : layers) {
            if (L.kind == L_RES || L.kind == L_CONV) *oc = L.cout;
            if (L.kind == L_UP || (L.kind == L_RES && L.up)) { *oh *= 2; *ow *= 2; }
            if (L.kind == L_DOWN || (L.kind == L_RES && L.down)) { *oh /= 2; *ow /= 2; }
        }
    };
    Tn c2 = b.alloc(B, h.h, h.w, h.c + hs[0].back().c);
    b.avg(&h, 1, 1.f, c2, 0, ACT_NONE, /*want_stats=*/true);
    b.release(h);
    for (size_t bi = 0; bi < sp.output_blocks.size(); ++bi) {
        Tn sk[4];
        for (int s = 0; s < 4; ++s) {
            sk[s] = hs[s].back();
            hs[s].pop_back();
        }
        DSD_CHECK(sk[0].h == c2.h && sk[0].w == c2.w, "decoder skip shape mismatch at output_blocks.%zu", bi);
        const int hc = c2.c - sk[0].c;
        b.avg(sk, 4, 4.f, c2, hc, ACT_NONE, /*want_stats=*/true);   // only taken when the h part brought its statistics
        for (auto& t : sk) b.release(t);
        const bool last = bi + 1 == sp.output_blocks.size();
        Tn next;
        if (!last) {
            int oc, oh, ow;
            block_out(sp.output_blocks[bi], c2.c, c2.h, c2.w, &oc, &oh, &ow);
            next = b.alloc(B, oh, ow, oc + hs[0].back().c);
        }
        size_t ei = 0;
        Tn out = b.block("output_blocks." + std::to_string(bi), sp.output_blocks[bi], c2, /*keep_input=*/false, dembs[bi], ei,
                         -1, last ? nullptr : &next);
        if (last) {
            h = out;
        } else {
            next.st[0] = out.st[0];   // the block's last convolution wrote h (and its statistics) into the next concat buffer
            c2 = next;
        }
    }
    // ---- out = Conv3x3(SiLU(GN(h)))  (model.py:511-515,751)
    b.out_conv1("out.0", "out.2", h, cfg.out_channels);
    b.release(h);
    b.release(emb_all);
}

// --------------------------------------------------------------------------------- UNetModel.forward (openaimodel.py:926-958)
// x: NCHW [B, in_channels, H, W] (latents, with the `concat` conditioning already appended), aux = timesteps [B] fp32.
void build_plain_unet(Builder& b, int C, int H, int W, int aux_len, int aux_len2) {
    dsd_handle* hd = b.hd;
    const dsd_config& cfg = hd->cfg;
    const StOpt so = st_opt_from_iargs(hd->iargs);
    const Spec sp = build_spec(cfg, true, so.on != 0);
    DSD_CHECK(!so.on || aux_len2 > 0, "UNetModel(use_spatial_transformer=True): context [B, tokens, %d] missing (aux2)", so.ctx_dim);
    DSD_CHECK(so.on || aux_len2 == 0, "UNetModel without spatial transformer takes no context");
    const int B = b.B;
    const int nds = cfg.n_levels - 1;
    DSD_CHECK(C == cfg.in_channels, "UNetModel: input has %d channels, expected %d", C, cfg.in_channels);
    DSD_CHECK(H % (1 << nds) == 0 && W % (1 << nds) == 0, "H=%d, W=%d must be multiples of %d (down/up-sampling + skip concat)", H, W, 1 << nds);
    DSD_CHECK(aux_len == 1, "UNetModel: timesteps missing");
    // ---- timestep embedding MLP + all emb_layers as ONE GEMM (openaimodel.py:939-940; :222-228,273)
    const int mc = cfg.model_channels, ted = sp.ted, etot = (int)hd->emb_total;
    Tn temb = b.alloc(B, 1, 1, mc), e1 = b.alloc(B, 1, 1, ted), emb = b.alloc(B, 1, 1, ted), emb_all = b.alloc(B, 1, 1, etot);
    {
        const size_t to = temb.off, e1o = e1.off, eo = emb.off, ao = emb_all.off;
        const float *w0 = b.W("time_embed.0.weight"), *b0 = b.W("time_embed.0.bias");
        const float *w2 = b.W("time_embed.2.weight"), *b2 = b.W("time_embed.2.bias");
        const float* wall = reinterpret_cast<const float*>(hd->slab + hd->emb_w_off);
        const float* ball = reinterpret_cast<const float*>(hd->slab + hd->emb_b_off);
        const double fl = 2.0 * B * ((double)mc * ted + (double)ted * ted + (double)ted * etot);
        b.plan.flops += fl;
        b.op([=](hipStream_t s) {
            float* tp = reinterpret_cast<float*>(hd->arena + to);
            float* e1p = reinterpret_cast<float*>(hd->arena + e1o);
            float* ep = reinterpret_cast<float*>(hd->arena + eo);
            float* ap = reinterpret_cast<float*>(hd->arena + ao);
            timestep_embedding(hd->io.aux, 1, B, mc, tp, s, hd->freqs);
            linear(tp, B, mc, mc, w0, b0, ted, ACT_NONE, e1p, ted, s);
            linear(e1p, B, ted, ted, w2, b2, ted, ACT_SILU, ep, ted, s);
            linear(ep, B, ted, ted, wall, ball, etot, ACT_SILU, ap, etot, s);
        }, 4, "time_embed_mlp", fl);
    }
    b.release(temb); b.release(e1); b.release(emb);
    std::unordered_map<std::string, int64_t> emb_col;
    {
        int64_t col = 0;
        for (const auto& p : hd->params)
            if (p.region == 2) {
                emb_col[p.name.substr(0, p.name.size() - std::strlen(".emb_layers.1.bias"))] = col;
                col += p.numel;
            }
    }
    auto embs_of = [&](const std::string& prefix, const std::vector<Layer>& layers) {
        std::vector<EmbRef> out;
        for (size_t li = 0; li < layers.size(); ++li)
            if (layers[li].kind == L_RES) {
                EmbRef e;
                e.arena_off = emb_all.off;
                e.col = emb_col.at(prefix + "." + std::to_string(li));
                e.stride = etot;
                e.valid = true;
                out.push_back(e);
            }
        return out;
    };
    // ---- encoder: h = module(h, emb, context); hs.append(h)   (:946-948)
    Tn x = b.import_ext(0, B, H, W, C, /*from_nchw=*/true);
    if (so.on) b.st_ctx = b.import_ext(2, B, aux_len2, 1, so.ctx_dim, false);
    std::vector<Tn> hs;
    Tn cur = x;
    for (size_t bi = 0; bi < sp.input_blocks.size(); ++bi) {
        const std::string pre_ = "input_blocks." + std::to_string(bi);
        auto embs = embs_of(pre_, sp.input_blocks[bi]);
        size_t ei = 0;
        Tn nxt = b.block(pre_, sp.input_blocks[bi], cur, /*keep_input=*/true, embs, ei);
        hs.push_back(nxt);
        cur = nxt;
    }
    b.release(x);
    // ---- middle (:949)
    Tn h;
    {
        auto me = embs_of("middle_block", sp.middle);
        size_t ei = 0;
        h = b.block("middle_block", sp.middle, hs.back(), /*keep_input=*/true, me, ei);
    }
    // ---- decoder: h = cat([h, hs.pop()]); h = module(h, emb)   (:950-952); a block's last convolution writes h straight into
    // the first channels of the next concat buffer
    auto block_out = [&](const std::vector<Layer>& layers, int c, int hh, int ww, int* oc, int* oh, int* ow) {
        *oc = c; *oh = hh; *ow = ww;
        for (const Layer& L : layers) {
            if (L.kind == L_RES || L.kind == L_CONV) *oc = L.cout;
            if (L.kind == L_UP || (L.kind == L_RES && L.up)) { *oh *= 2; *ow *= 2; }
            if (L.kind == L_DOWN || (L.kind == L_RES && L.down)) { *oh /= 2; *ow /= 2; }
        }
    };
    Tn c2 = b.alloc(B, h.h, h.w, h.c + hs.back().c);
    b.avg(&h, 1, 1.f, c2, 0, ACT_NONE, /*want_stats=*/true);
    b.release(h);
    for (size_t bi = 0; bi < sp.output_blocks.size(); ++bi) {
        Tn sk = hs.back();
        hs.pop_back();
        DSD_CHECK(sk.h == c2.h && sk.w == c2.w, "decoder skip shape mismatch at output_blocks.%zu", bi);
        const int hc = c2.c - sk.c;
        b.avg(&sk, 1, 1.f, c2, hc, ACT_NONE, /*want_stats=*/true);
        b.release(sk);
        const bool last = bi + 1 == sp.output_blocks.size();
        Tn next;
        if (!last) {
            int oc, oh, ow;
            block_out(sp.output_blocks[bi], c2.c, c2.h, c2.w, &oc, &oh, &ow);
            next = b.alloc(B, oh, ow, oc + hs.back().c);
        }
        const std::string pre_ = "output_blocks." + std::to_string(bi);
        auto de = embs_of(pre_, sp.output_blocks[bi]);
        size_t ei = 0;
        Tn out = b.block(pre_, sp.output_blocks[bi], c2, /*keep_input=*/false, de, ei, -1, last ? nullptr : &next);
        if (last) {
            h = out;
        } else {
            next.st[0] = out.st[0];
            c2 = next;
        }
    }
    // ---- out = Conv3x3(SiLU(GN(h)))   (:901-905,957)
    Tn a = b.gn_act("out.0", h, ACT_SILU);
    b.release(h);
    Tn y = b.conv("out.2", a, cfg.out_channels, 3);
    b.release(a);
    b.export_out(y, /*to_nchw=*/true);
    b.release(y);
    if (so.on) b.release(b.st_ctx);
    b.release(emb_all);
}

// --------------------------------------------------------------------------------- DiT.forward (DiT_models.py:224-243)
// x: NCHW [B, in_channels, S, S] (the caller concatenates `cond`), aux = t [B] fp32, aux2 = y [B] int64 (optional).
void build_dit(Builder& b, int C, int H, int W, int aux_len, int aux_len2) {
    dsd_handle* hd = b.hd;
    const DitCfg c = dit_cfg(hd->iargs);
    const int B = b.B, T = c.T(), D = c.D, p = c.p, K = c.cin * p * p;
    DSD_CHECK(C == c.cin, "DiT: input has %d channels, x_embedder expects %d (x and cond concatenated)", C, c.cin);
    DSD_CHECK(H == c.input && W == c.input, "DiT: input is %dx%d, pos_embed was built for %dx%d", H, W, c.input, c.input);
    DSD_CHECK(aux_len == 1, "DiT: timesteps missing");
    DSD_CHECK(aux_len2 == 0 || c.classes > 0, "DiT: labels given but the model has no label embedding");
    const int etot = (int)hd->emb_total;   // depth * 6D + 2D: every adaLN modulation of the forward
    // ---- patch embedding: Conv2d(k = stride = p) = patch gather + GEMM, + bias, + fixed sin-cos position table (:232)
    Tn tok = b.alloc(B, T, 1, K);
    {
        const size_t to = tok.off;
        b.op([=](hipStream_t s) { patchify(hd->io.x_nchw, B, C, H, W, p, reinterpret_cast<float*>(hd->arena + to), s); }, 1, "patchify");
    }
    Tn x = b.conv("x_embedder.proj", tok, D, 1);
    b.release(tok);
    {
        const size_t xo = x.off;
        const float* pos = b.W("pos_embed");
        b.op([=](hipStream_t s) { add_rows_broadcast(reinterpret_cast<float*>(hd->arena + xo), pos, B, (int64_t)T * D, s); }, 1, "pos_embed_add");
    }
    // ---- conditioning vector c = t_embedder(t) [+ y_embedder(y)] and ALL adaLN modulations as one GEMM (:233-238, :113-116)
    Tn tf = b.alloc(B, 1, 1, 256), e1 = b.alloc(B, 1, 1, D), cvec = b.alloc(B, 1, 1, D), mod = b.alloc(B, 1, 1, etot);
    {
        const size_t tfo = tf.off, e1o = e1.off, co = cvec.off, mo = mod.off;
        const float *w0 = b.W("t_embedder.mlp.0.weight"), *b0 = b.W("t_embedder.mlp.0.bias");
        const float *w2 = b.W("t_embedder.mlp.2.weight"), *b2 = b.W("t_embedder.mlp.2.bias");
        const float* table = c.classes > 0 ? b.W("y_embedder.embedding_table.weight") : nullptr;
        const float* wall = reinterpret_cast<const float*>(hd->slab + hd->emb_w_off);
        const float* ball = reinterpret_cast<const float*>(hd->slab + hd->emb_b_off);
        const bool labels = aux_len2 > 0;
        const double fl = 2.0 * B * (256.0 * D + (double)D * D + (double)D * etot);
        b.plan.flops += fl;
        b.op([=](hipStream_t s) {
            float* tfp = reinterpret_cast<float*>(hd->arena + tfo);
            float* e1p = reinterpret_cast<float*>(hd->arena + e1o);
            float* cp = reinterpret_cast<float*>(hd->arena + co);
            float* mp = reinterpret_cast<float*>(hd->arena + mo);
            timestep_embedding(hd->io.aux, 1, B, 256, tfp, s, hd->freqs);
            linear(tfp, B, 256, 256, w0, b0, D, ACT_NONE, e1p, D, s);
            linear(e1p, B, D, D, w2, b2, D, ACT_SILU, cp, D, s);
            if (labels) embed_add(cp, table, reinterpret_cast<const long long*>(hd->io.aux2), B, D, cp, s);
            linear(cp, B, D, D, wall, ball, etot, ACT_SILU, mp, etot, s);
        }, 5, "dit_conditioning", fl);
    }
    b.release(tf); b.release(e1); b.release(cvec);
    // column of each modulation inside `mod` = position of its bias in the contiguous bias region (declaration order)
    std::unordered_map<std::string, int> col;
    {
        int cc = 0;
        for (const auto& pr : hd->params)
            if (pr.region == 2) {
                col[pr.name.substr(0, pr.name.size() - std::strlen(".bias"))] = cc;
                cc += (int)pr.numel;
            }
    }
    // half-precision modes: the four Linears and the attention of every block take 16-bit operands (gemm16.hip,
    // attention16.hip); shapes those kernels do not take (widths not multiples of 8, head dims beyond 128) keep bf16x6
    const int hdim0 = D / c.heads;
    const bool h16 = b.half_mode() && gemm16_shape_ok(B * T, D, D) && gemm16_shape_ok(B * T, D, c.mlp) && c.mlp % 8 == 0 &&
                     D % c.heads == 0 && attention16_shape_ok(hdim0);
    const int bf = hd->precision == PREC_BF16 ? 1 : 0;
    auto ln_mod = [&](const Tn& src, int mcol, int shift_off, int scale_off, bool out16 = false) {
        Tn y = b.alloc(src.n, src.h, src.w, src.c, out16 ? 2 : 4);
        const size_t so = src.off, yo = y.off, mo = mod.off;
        if (out16) {
            b.op([=](hipStream_t s) {
                ln_modulate16(reinterpret_cast<const float*>(hd->arena + so), B, T, D, reinterpret_cast<const float*>(hd->arena + mo) + mcol,
                              etot, shift_off, scale_off, 1e-6f, hd->arena + yo, bf, s);
            }, 1, "ln_modulate16", 0.0, 6.0 * B * T * D);
            return y;
        }
        b.op([=](hipStream_t s) {
            ln_modulate(reinterpret_cast<const float*>(hd->arena + so), B, T, D, reinterpret_cast<const float*>(hd->arena + mo) + mcol, etot,
                        shift_off, scale_off, 1e-6f, reinterpret_cast<float*>(hd->arena + yo), s);
        }, 1, "ln_modulate", 0.0, 8.0 * B * T * D);
        return y;
    };
    auto gated = [&](const Tn& xx, const Tn& yy, int mcol, int gate_off) {
        const size_t xo = xx.off, yo = yy.off, mo = mod.off;
        b.op([=](hipStream_t s) {
            gated_residual(reinterpret_cast<float*>(hd->arena + xo), reinterpret_cast<const float*>(hd->arena + yo), B, T, D,
                           reinterpret_cast<const float*>(hd->arena + mo) + mcol, etot, gate_off, s);
        }, 1, "gated_residual", 0.0, 12.0 * B * T * D);
    };
    const int hdim = D / c.heads;
    for (int i = 0; i < c.depth; ++i) {   // DiTBlock.forward :118-122
        const std::string bp = "blocks." + std::to_string(i);
        const int mc = col.at(bp + ".adaLN_modulation.1");   // chunk(6): shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        if (h16) {
            // x = x + gate_msa * proj(attn(qkv(modulate(norm1(x)))));  x = x + gate_mlp * fc2(gelu(fc1(modulate(norm2(x)))))
            // q leaves the qkv GEMM multiplied by hd^-1/2 log2(e) (one rounding): the attention kernel works on base-2 logits
            Tn n1 = ln_mod(x, mc, 0, D, true);
            Tn qkv = b.linear16(bp + ".attn.qkv", n1, 3 * D, EPI16_STORE, nullptr, 0, 0, 0, D,
                                1.4426950408889634f / std::sqrt((float)hdim0));
            b.release(n1);
            Tn a = b.alloc(B, T, 1, D, 2);
            {
                Attn16Args aa;
                aa.N = B; aa.Tq = T; aa.Tk = T; aa.heads = c.heads; aa.d = hdim0;
                aa.ldq = aa.ldk = aa.ldv = 3 * D; aa.ldo = D;
                aa.q_hs = aa.k_hs = aa.v_hs = hdim0;
                aa.bf16 = bf;
                const size_t qo = qkv.off, ao = a.off;
                const double fl = 4.0 * B * c.heads * (double)T * T * hdim0;
                b.plan.flops += fl;
                b.op([=](hipStream_t s) {
                    Attn16Args r = aa;
                    const char* base = hd->arena + qo;
                    r.q = base; r.k = base + (size_t)D * 2; r.v = base + (size_t)2 * D * 2;
                    r.out = hd->arena + ao;
                    attention16(r, s);
                }, 1, "attention16", fl, 2.0 * B * T * 4.0 * D);
            }
            b.release(qkv);
            b.linear16(bp + ".attn.proj", a, D, EPI16_GATED, &x, mod.off, mc + 2 * D, etot);
            b.release(a);
            Tn n2 = ln_mod(x, mc, 3 * D, 4 * D, true);
            Tn h1 = b.linear16(bp + ".mlp.fc1", n2, c.mlp, EPI16_GELU);
            b.release(n2);
            b.linear16(bp + ".mlp.fc2", h1, D, EPI16_GATED, &x, mod.off, mc + 5 * D, etot);
            b.release(h1);
            continue;
        }
        Tn n1 = ln_mod(x, mc, 0, D);
        Tn qkv = b.conv(bp + ".attn.qkv", n1, 3 * D, 1);
        b.release(n1);
        Tn a = b.alloc(B, T, 1, D);
        {   // timm Attention: qkv.reshape(B, N, 3, heads, hd); q * hd^-0.5; softmax(q k^T) v
            AttnArgs aa;
            aa.N = B; aa.Tq = T; aa.Tk = T; aa.heads = c.heads; aa.d = hdim;
            aa.ldq = aa.ldk = aa.ldv = 3 * D; aa.ldo = D;
            aa.q_hs = aa.k_hs = aa.v_hs = hdim;
            aa.split = hd->precision != PREC_F32;
            aa.scale_q = 1.f / std::sqrt((float)hdim);
            const size_t qo = qkv.off, ao = a.off;
            const double fl = 4.0 * B * c.heads * (double)T * T * hdim;
            b.plan.flops += fl;
            b.op([=](hipStream_t s) {
                AttnArgs r = aa;
                const float* base = reinterpret_cast<const float*>(hd->arena + qo);
                r.q = base; r.k = base + D; r.v = base + 2 * D;
                r.out = reinterpret_cast<float*>(hd->arena + ao);
                attention(r, s);
            }, 1, "attention", fl);
        }
        b.release(qkv);
        Tn y1 = b.conv(bp + ".attn.proj", a, D, 1);
        b.release(a);
        gated(x, y1, mc, 2 * D);
        b.release(y1);
        Tn n2 = ln_mod(x, mc, 3 * D, 4 * D);
        Tn h1 = b.conv(bp + ".mlp.fc1", n2, c.mlp, 1);
        b.release(n2);
        {
            const size_t ho = h1.off;
            const int64_t cnt = (int64_t)B * T * c.mlp;
            b.op([=](hipStream_t s) { gelu_tanh(reinterpret_cast<float*>(hd->arena + ho), cnt, s); }, 1, "gelu_tanh", 0.0, 8.0 * cnt);
        }
        Tn y2 = b.conv(bp + ".mlp.fc2", h1, D, 1);
        b.release(h1);
        gated(x, y2, mc, 5 * D);
        b.release(y2);
    }
    // ---- FinalLayer.forward :138-142 + unpatchify :209-222
    Tn nf = ln_mod(x, col.at("final_layer.adaLN_modulation.1"), 0, D);
    b.release(x);
    const int co = c.cout();
    Tn o = b.conv("final_layer.linear", nf, p * p * co, 1);
    b.release(nf);
    {
        const size_t oo = o.off;
        const int hw = c.input / p;
        b.op([=](hipStream_t s) { unpatchify(reinterpret_cast<const float*>(hd->arena + oo), B, co, hw, hw, p, hd->io.out, s); }, 1, "unpatchify");
    }
    b.release(o);
    b.release(mod);
}

// --------------------------------------------------------------------------------- single blocks
void build_block(Builder& b, int C, int H, int W, int aux_len, int aux_len2) {
    dsd_handle* hd = b.hd;
    const auto& a = hd->iargs;
    const int B = b.B;
    const int kind = hd->block_kind;
    const bool token = kind == DSD_BLOCK_CROSSATTN || kind == DSD_BLOCK_FF_GEGLU || kind == DSD_BLOCK_BASIC_TRANSFORMER;
    if (kind == DSD_BLOCK_DIT) {
        build_dit(b, C, H, W, aux_len, aux_len2);
        return;
    }
    if (kind == DSD_BLOCK_UNET) {
        build_plain_unet(b, C, H, W, aux_len, aux_len2);
        return;
    }
    Tn x = b.import_ext(0, B, H, W, C, !token);
    Tn y;
    switch (kind) {
        case DSD_BLOCK_RES: {
            DSD_CHECK(aux_len == a[2], "ResBlock: emb has %d channels, expected %d", aux_len, a[2]);
            const int eo = (int)hd->PP("emb_layers.1.weight").shape[0];
            Tn e = b.import_ext(1, B, 1, 1, aux_len, false);
            Tn eout = b.alloc(B, 1, 1, eo);
            const size_t ei = e.off, oo = eout.off;
            const float *w = b.W("emb_layers.1.weight"), *bb = b.W("emb_layers.1.bias");
            b.op([=](hipStream_t s) {
                linear(reinterpret_cast<const float*>(hd->arena + ei), B, aux_len, aux_len, w, bb, eo, ACT_SILU,
                       reinterpret_cast<float*>(hd->arena + oo), eo, s);
            });
            EmbRef er;
            er.arena_off = eout.off; er.col = 0; er.stride = eo; er.valid = true;
            y = b.res_block("", x, a[0], a[1], a[4] != 0, a[5] != 0, er);
            b.release(e); b.release(eout);
            break;
        }
        case DSD_BLOCK_ATTN: y = b.attn_block("", x, a[1], a[2] != 0); break;
        case DSD_BLOCK_UPSAMPLE: y = b.conv("conv", x, a[0], 3, 1, true); break;
        case DSD_BLOCK_DOWNSAMPLE: y = b.conv("op", x, a[0], 3, 2); break;
        case DSD_BLOCK_DISENTANGLE: y = b.disentangle("", x, a[1]); break;
        case DSD_BLOCK_SE: y = b.se("", x); break;
        case DSD_BLOCK_CROSSATTN: {
            Tn ctx;
            if (aux_len > 0) ctx = b.import_ext(1, B, aux_len, 1, a[1], false);
            y = b.xattn("", x, aux_len > 0 ? &ctx : nullptr, a[2], nullptr);
            if (aux_len > 0) b.release(ctx);
            break;
        }
        case DSD_BLOCK_FF_GEGLU: y = b.ff_geglu("", x, nullptr); break;
        case DSD_BLOCK_BASIC_TRANSFORMER: {
            Tn ctx;
            if (aux_len > 0) ctx = b.import_ext(1, B, aux_len, 1, a[3], false);
            y = b.btb("", x, aux_len > 0 ? &ctx : nullptr, a[1]);
            if (aux_len > 0) b.release(ctx);
            break;
        }
        case DSD_BLOCK_SPATIAL_TRANSFORMER: {
            const int depth = a[3];
            DSD_CHECK(depth >= 1 && depth <= 2, "spatial transformer block handle supports depth 1..2 (two context inputs)");
            Tn ctx[2];
            if (aux_len > 0) ctx[0] = b.import_ext(1, B, aux_len, 1, a[4], false);
            if (depth > 1 && aux_len2 > 0) ctx[1] = b.import_ext(2, B, aux_len2, 1, a[4], false);
            y = b.spatial_transformer("", x, a[1], a[2], depth, ctx);
            for (auto& c : ctx) b.release(c);
            break;
        }
        case DSD_BLOCK_VAE_ENCODER: {   // Encoder.forward (model.py:519-543) [+ quant_conv, autoencoder.py:138-142]
            const VaeCfg c = vae_cfg(a);
            const int L = (int)c.mult.size();
            DSD_CHECK(C == c.in_ch, "VAE encoder: input has %d channels, expected %d", C, c.in_ch);
            DSD_CHECK(H % (1 << (L - 1)) == 0 && W % (1 << (L - 1)) == 0, "VAE encoder: H=%d, W=%d must be multiples of %d", H, W, 1 << (L - 1));
            const std::string e = "encoder";
            Tn hcur = b.conv(e + ".conv_in", x, c.ch, 3, 1, false, nullptr, nullptr, -1, false, true, nullptr, true);
            int res = c.resolution;
            for (int l = 0; l < L; ++l) {
                for (int j = 0; j < c.nrb; ++j) {
                    Tn t = b.vae_res(e + ".down." + std::to_string(l) + ".block." + std::to_string(j), hcur, c.ch * c.mult[l]);
                    b.release(hcur);
                    hcur = t;
                    if (c.attn_at(res)) {
                        Tn t2 = b.vae_attn(e + ".down." + std::to_string(l) + ".attn." + std::to_string(j), hcur);
                        b.release(hcur);
                        hcur = t2;
                    }
                }
                if (l != L - 1) {   // Downsample :78-83: zero row/column AFTER the last one, stride 2, no padding before
                    Tn t = b.conv(e + ".down." + std::to_string(l) + ".downsample.conv", hcur, hcur.c, 3, 2, false, nullptr, nullptr,
                                  -1, false, true, nullptr, true, /*pad_lo=*/0, /*pad_total=*/1);
                    b.release(hcur);
                    hcur = t;
                    res /= 2;
                }
            }
            Tn t = b.vae_res(e + ".mid.block_1", hcur, hcur.c);
            b.release(hcur);
            Tn t2 = b.vae_attn(e + ".mid.attn_1", t);
            b.release(t);
            Tn t3 = b.vae_res(e + ".mid.block_2", t2, t2.c);
            b.release(t2);
            Tn n = b.gn_act(e + ".norm_out", t3, ACT_SILU, 1e-6f);
            b.release(t3);
            y = b.conv(e + ".conv_out", n, c.double_z ? 2 * c.z : c.z, 3);
            b.release(n);
            if (c.with_quant) {
                Tn m = b.conv("quant_conv", y, 2 * c.embed, 1);
                b.release(y);
                y = m;
            }
            break;
        }
        case DSD_BLOCK_VAE_DECODER: {   // [post_quant_conv, autoencoder.py:144-147 +] Decoder.forward (model.py:618-655)
            const VaeCfg c = vae_cfg(a);
            const int L = (int)c.mult.size();
            DSD_CHECK(C == (c.with_quant ? c.embed : c.z), "VAE decoder: input has %d channels, expected %d", C, c.with_quant ? c.embed : c.z);
            const std::string d = "decoder";
            Tn zin = x;
            bool zin_owned = false;
            if (c.with_quant) {
                zin = b.conv("post_quant_conv", x, c.z, 1);
                zin_owned = true;
            }
            Tn hcur = b.conv(d + ".conv_in", zin, c.ch * c.mult[L - 1], 3, 1, false, nullptr, nullptr, -1, false, true, nullptr, true);
            if (zin_owned) b.release(zin);
            Tn t = b.vae_res(d + ".mid.block_1", hcur, hcur.c);
            b.release(hcur);
            Tn t2 = b.vae_attn(d + ".mid.attn_1", t);
            b.release(t);
            hcur = b.vae_res(d + ".mid.block_2", t2, t2.c);
            b.release(t2);
            int res = c.resolution >> (L - 1);
            for (int l = L - 1; l >= 0; --l) {
                for (int j = 0; j < c.nrb + 1; ++j) {
                    Tn r = b.vae_res(d + ".up." + std::to_string(l) + ".block." + std::to_string(j), hcur, c.ch * c.mult[l]);
                    b.release(hcur);
                    hcur = r;
                    if (c.attn_at(res)) {
                        Tn r2 = b.vae_attn(d + ".up." + std::to_string(l) + ".attn." + std::to_string(j), hcur);
                        b.release(hcur);
                        hcur = r2;
                    }
                }
                if (l != 0) {
                    Tn r = b.conv(d + ".up." + std::to_string(l) + ".upsample.conv", hcur, hcur.c, 3, 1, true, nullptr, nullptr, -1,
                                  false, true, nullptr, true);
                    b.release(hcur);
                    hcur = r;
                    res *= 2;
                }
            }
            Tn n = b.gn_act(d + ".norm_out", hcur, ACT_SILU, 1e-6f);
            b.release(hcur);
            y = b.conv(d + ".conv_out", n, c.out_ch, 3);
            b.release(n);
            break;
        }
        default: fail("unknown block kind");
    }
    b.export_out(y, !token);
    b.release(y);
    b.release(x);
}

}  // namespace

void dsd::net_plan(dsd_handle* h, int B, int C, int H, int W, int zero_al_l, int want_feats, int aux_len, int aux_len2,
                   int share, hipStream_t s) {
    Plan& p = h->plan;
    if (p.valid && p.B == B && p.C == C && p.H == H && p.W == W && p.zero_al_l == zero_al_l && p.want_feats == want_feats &&
        p.aux_len == aux_len && p.aux_len2 == aux_len2 && p.share == share)
        return;
    DSD_CHECK(h->device >= 0, "this handle was created without a device (table only)");
    for (const auto& prm : h->params) DSD_CHECK(prm.set, "parameter '%s' has not been set", prm.name.c_str());
    DSD_CHECK(B >= 1 && H >= 1 && W >= 1, "empty input");
    const int gen = p.gen + 1;
    net_drop_graph(h);
    p = Plan{};
    p.gen = gen;
    p.B = B; p.C = C; p.H = H; p.W = W; p.zero_al_l = zero_al_l; p.want_feats = want_feats;
    p.aux_len = aux_len; p.aux_len2 = aux_len2; p.share = share;
    for (auto& kv : h->param_evs) DSD_HIP(hipStreamWaitEvent(s, kv.second, 0));   // uploads enqueued on ANY stream have landed first
    Builder b(h, p, B, s);
    try {
        if (h->is_block)
            build_block(b, C, H, W, aux_len, aux_len2);
        else
            build_unet(b, H, W, zero_al_l != 0, want_feats != 0, share != 0);
    } catch (...) {
        // a plan that fails half-way may have enqueued weight-piece kernels on s: let them finish before anybody can free or
        // overwrite what they read, and leave no half-built plan behind
        if (b.split_any) (void)hipStreamSynchronize(s);
        p = Plan{};
        p.gen = gen;
        throw;
    }
    p.arena_bytes = b.ar.peak + 256;
    if (p.arena_bytes > h->arena_cap) {
        DSD_HIP(hipDeviceSynchronize());
        if (h->arena) DSD_HIP(hipFree(h->arena));
        h->arena = nullptr;
        h->arena_cap = 0;
        DSD_HIP(hipMalloc((void**)&h->arena, p.arena_bytes));
        h->arena_cap = p.arena_bytes;
    }
    if (b.split_any) DSD_HIP(hipStreamSynchronize(s));   // plan time only: the pieces exist before any stream may use them
    p.valid = true;
}

void dsd::net_launch_ops(dsd_handle* h, hipStream_t s) {
    Plan& p = h->plan;
    bool any_lane = false;
    if (h->use_lanes == 1)
        for (signed char l : p.op_lane) any_lane |= l > 0;
    if (!any_lane) {
        for (auto& f : p.ops) f(s);
        return;
    }
    if (!h->lane_fork) {
        DSD_HIP(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
        for (int l = 0; l < 3; ++l) {
            DSD_HIP(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
            DSD_HIP(hipEventCreateWithFlags(&h->lane_join[l], hipEventDisableTiming));
        }
    }
    unsigned active = 0;      // bit l: lane l (1..3) has work in flight since the last join
    bool forked = false;      // the fork event of the current region has been recorded
    auto join = [&]() {
        for (int l = 1; l <= 3; ++l)
            if (active & (1u << l)) {
                DSD_HIP(hipEventRecord(h->lane_join[l - 1], h->lane_stream[l - 1]));
                DSD_HIP(hipStreamWaitEvent(s, h->lane_join[l - 1], 0));
            }
        active = 0;
        forked = false;
    };
    for (size_t i = 0; i < p.ops.size(); ++i) {
        const int L = p.op_lane[i];
        if (L < 0) {                       // sequential op: everything the lanes produced is its potential input
            if (active) join();
            forked = false;
            p.ops[i](s);
            continue;
        }
        if (!forked) {                     // first op of a region: lanes may start once everything before it is done
            DSD_HIP(hipEventRecord(h->lane_fork, s));
            forked = true;
        }
        if (L == 0) {                      // lane 0 is the caller's stream itself
            p.ops[i](s);
            continue;
        }
        if (!(active & (1u << L))) {
            DSD_HIP(hipStreamWaitEvent(h->lane_stream[L - 1], h->lane_fork, 0));
            active |= 1u << L;
        }
        p.ops[i](h->lane_stream[L - 1]);
    }
    if (active) join();
}

void dsd::net_run(dsd_handle* h, hipStream_t s) {
    DSD_CHECK(h->plan.valid, "no plan");
    Plan& p = h->plan;
    if (!h->profiling) {
        net_launch_ops(h, s);
        ++p.eager_runs;
        return;
    }
    // profiling pass: one event pair per op on the launch stream, read back after a stream sync
    const size_t n = p.ops.size();
    while (h->ev.size() < 2 * n) {
        hipEvent_t e;
        DSD_HIP(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    for (size_t i = 0; i < n; ++i) {
        DSD_HIP(hipEventRecord(h->ev[2 * i], s));
        p.ops[i](s);
        DSD_HIP(hipEventRecord(h->ev[2 * i + 1], s));
    }
    DSD_HIP(hipStreamSynchronize(s));
    if (h->prof_names != p.kind_names) {
        h->prof_names = p.kind_names;
        h->prof_ms.assign(p.kind_names.size(), 0.0);
        h->prof_flops.assign(p.kind_names.size(), 0.0);
        h->prof_bytes.assign(p.kind_names.size(), 0.0);
        h->prof_calls.assign(p.kind_names.size(), 0);
        h->prof_runs = 0;
    }
    h->prof_op_ms.assign(n, 0.f);
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        DSD_HIP(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        h->prof_op_ms[i] = ms;
        const int k = p.op_kind[i];
        h->prof_ms[k] += ms;
        h->prof_flops[k] += p.op_flops[i];
        h->prof_bytes[k] += p.op_bytes[i];
        h->prof_calls[k] += 1;
    }
    h->prof_runs += 1;
}

void dsd::net_check_overflow(dsd_handle* h, hipStream_t s) {
    if (h->precision != PREC_F16X3 || !h->ovf) return;
    int flag = 0;
    DSD_HIP(hipMemcpyAsync(&flag, h->ovf, sizeof(int), hipMemcpyDeviceToHost, s));
    DSD_HIP(hipStreamSynchronize(s));
    if (flag) {
        DSD_HIP(hipMemsetAsync(h->ovf, 0, sizeof(int), s));
        fail("f16x3: a convolution operand exceeded the fp16 range (|x| > 65504); the result is invalid - use bf16x6 or f32");
    }
}

// --------------------------------------------------------------------------------------------- whole-forward hipGraph
// The plan is a fixed list of launches whose arguments depend only on the plan and on the pointers bound in io, so a
// sampling loop (same buffers every step) replays ONE captured graph per step instead of ~1000 host launches (a22: the
// reference's Python loop, gaussian_diffusion.py:569-616).  Capture runs the same closures on a library-owned stream (the
// caller's may be the legacy NULL stream, which cannot be captured) after the plan's first host-launched forward, so lazy
// code-object loading and any first-use initialisation have happened; the instantiated graph is launched on the caller's
// stream.  Profiling and the f16x3 overflow check keep working: profiling bypasses the graph, the flag is read after it.
static GraphKey graph_key(const dsd_handle* h) {
    GraphKey k;
    k.plan_gen = h->plan.gen;
    for (int i = 0; i < 4; ++i) {
        k.ptr[i] = h->io.plane[i];
        k.bs[i] = h->io.plane_bs[i];
    }
    k.ptr[4] = h->io.t; k.ptr[5] = h->io.out; k.ptr[6] = h->io.feats; k.ptr[7] = h->io.x_nchw; k.ptr[8] = h->io.aux;
    k.ptr[9] = h->io.aux2; k.ptr[10] = h->arena; k.ptr[11] = h->freqs;
    k.t_is_float = h->io.t_is_float;
    return k;
}

void dsd::net_run_cached(dsd_handle* h, hipStream_t s) {
    DSD_CHECK(h->plan.valid, "no plan");
    if (!h->use_graph || h->profiling || h->io.feats) {
        net_run(h, s);
        return;
    }
    const GraphKey key = graph_key(h);
    if (h->gexec && key == h->gkey) {
        DSD_HIP(hipGraphLaunch(h->gexec, s));
        ++h->graph_launches;
        return;
    }
    if (h->plan.eager_runs == 0) {   // first forward of a plan: from the host (loads code objects, warms the caches)
        net_run(h, s);
        return;
    }
    net_drop_graph(h);
    if (!h->cap_stream) DSD_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    DSD_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    try {
        net_launch_ops(h, h->cap_stream);   // (lane streams join the capture through the fork event and leave it at the join)
    } catch (...) {
        (void)hipStreamEndCapture(h->cap_stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    DSD_HIP(hipStreamEndCapture(h->cap_stream, &graph));
    hipError_t e = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        h->gexec = nullptr;
        fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    h->gkey = key;
    ++h->graph_captures;
    DSD_HIP(hipGraphLaunch(h->gexec, s));
    ++h->graph_launches;
}
