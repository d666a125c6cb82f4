// Launcher declarations for the hand-written gfx950 kernels of libdsdiff.
#pragma once
#include "common.h"

namespace dsd {

// ---------------------------------------------------------------- conv.hip
// Implicit-GEMM convolution on NHWC fp32, f32 MFMA (v_mfma_f32_32x32x2_f32), M = N*OH*OW, N = Cout, K = ks*ks*Cin.
struct ConvArgs {
    const float* x = nullptr;   // input, NHWC within a sample
    int N = 0, H = 0, W = 0, Cin = 0;
    int64_t x_bs = -1;          // batch stride of x in elements (<0 -> H*W*Cin; 0 = one plane shared by all samples)
    const float* w = nullptr;   // packed OHWI: [Cout][ks*ks*Cin]
    const float* bias = nullptr;
    int Cout = 0, ks = 3, stride = 1;
    int pad_lo = -1, pad_total = -1;   // padding before the first row/column and in total per axis (-1: ks/2 and 2*(ks/2))
    int ups = 0;                // nearest x2 folded into the gather (Upsample, openaimodel.py:111-121)
    const float* emb = nullptr; // per-(n,co) add: emb[n*emb_stride + co]   (ResBlock h + emb_out, openaimodel.py:282)
    int emb_stride = 0;
    const float* res = nullptr; // residual NHWC [N,OH,OW,Cout] added in the epilogue (skip_connection(x) + h)
    float* y = nullptr;
    int y_ld = 0;               // row stride of y in elements (0 -> Cout): lets a conv write a channel slice of a wider
                                // NHWC tensor, e.g. the decoder's concat buffer (torch.cat([h, skip]) never copied)
    int out_nchw = 0;           // store as [N,Cout,OH,OW] (final layer with out_channels > 1)
    int OH = 0, OW = 0;         // filled by conv2d()
    int variant = -1;           // kernel variant override (-1 = default / env DSD_CONV_VARIANT)
    int precision = 0;          // PREC_F32 | PREC_BF16X3 | PREC_BF16X6 (conv_split.hip)
    const void* w_split = nullptr;  // [3][Cout][ks*ks*Cin] bf16 pieces of w (needed for the split precisions)
    const void* w_wino = nullptr;   // transformed + split + packed weights of the F(2,3) kernel (conv_wino.hip), optional
    int* ovf = nullptr;             // device flag set by the f16x3 kernels when an operand exceeds the fp16 range
    float* scratch = nullptr;       // split-K partial sums (conv2d_scratch_bytes() bytes); without it small grids run unsplit
    size_t scratch_bytes = 0;
    // GroupNorm statistics of the OUTPUT, accumulated in the epilogue: stats[n][chunk][co][2] doubles (GnSrc layout),
    // stats_chunks = conv2d_stats_chunks() chunks per sample; nullptr = not wanted
    double* stats = nullptr;
    int stats_chunks = 0;
    // diagnostic build of the dominant kernel only (tools/conv_stamps.py): 8 int64 per workgroup — (s_memtime,
    // s_memrealtime) at entry, in front of the k-loop, behind it, and after the epilogue
    long long* stamps = nullptr;
    int diag = 0;   // what-if bits (conv_split.hip, DIAG)
    // GroupNorm + SiLU of the INPUT applied while the filter rows are staged (tap-reuse kernel only: conv2d_fuses_gn()):
    // x is then the tensor BEFORE the normalisation and gn_scale / gn_shift are gn_finalize's per-(sample, channel)
    // coefficients [N][Cin]; the separate apply pass over HBM disappears
    const float* gn_scale = nullptr;
    const float* gn_shift = nullptr;
    // launches of this shape that run side by side on other HIP streams (the four encoder streams' lanes, net.cpp): the planner
    // sizes tile width / split-K for 1/lanes of the chip — a grid that would leave three quarters of the CUs idle alone fills them
    // together, so the wide tap-reuse kernel (with the fused GroupNorm) is the right choice where a lone launch takes narrow tiles
    int lanes = 1;
};
// Output size.  Default padding is ks/2 on every side (the U-Net's convolutions); pad_lo / pad_total describe the VAE's
// Downsample (ldm/modules/diffusionmodules/model.py:78-83: F.pad (0,1,0,1) then a stride-2 conv with padding 0), i.e. no
// padding before the first row / column and one zero row / column after the last.
inline void conv_out_hw(const ConvArgs& a, int* OH, int* OW) {
    const int IHg = a.ups ? a.H * 2 : a.H, IWg = a.ups ? a.W * 2 : a.W;
    const int pt = a.pad_total >= 0 ? a.pad_total : 2 * (a.ks / 2);
    *OH = (IHg + pt - a.ks) / a.stride + 1;
    *OW = (IWg + pt - a.ks) / a.stride + 1;
}
// chunks per sample the kernel conv2d() will launch for these arguments can emit output statistics with (0 = it cannot:
// the caller runs gn_stats on the output instead)
int conv2d_stats_chunks(const ConvArgs& a);
// out = Conv3x3(SiLU(x * scale + shift)) with ONE output channel (conv_out1.hip): x NHWC [N,H,W,C] BEFORE the normalisation,
// scale / shift = gn_finalize's coefficients [N][C], w = [3][3][C] (the packed OHWI weight of a 1-channel conv), y = [N,H,W]
struct ConvOut1Args {
    const float* x = nullptr;
    int N = 0, H = 0, W = 0, C = 0;
    const float* scale = nullptr;
    const float* shift = nullptr;
    const float* w = nullptr;
    const float* bias = nullptr;
    float* y = nullptr;
};
bool conv_out1_ok(int C, int cout, int ks, int stride);
void conv_out1(const ConvOut1Args& a, hipStream_t s);
enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_BF16X6 = 2, PREC_F16X3 = 3, PREC_F16 = 4, PREC_BF16 = 5 };
void conv2d(ConvArgs a, hipStream_t s);
// conv_split.hip: fp32 operands as sums of bf16 pieces on the bf16 matrix cores
void split_weights(const float* w, int64_t n, int np, void* planes, hipStream_t s, bool f16 = false, int* ovf = nullptr);
bool conv2d_split_eligible(const ConvArgs& a);
bool conv2d_split_tr(const ConvArgs& a, int nt, int ksplit, int ad);   // the tap-reuse instantiation would take it
// process-wide switch of the tap-reuse kernel's MFMA shape (0: 32x32x16, 1: 16x16x32 — conv_tr16.hip); A/B runs in one process
void conv2d_set_mfma16(int on);
int conv2d_get_mfma16();
bool conv2d_fuses_gn(const ConvArgs& a);   // conv2d(a) can apply GroupNorm + SiLU to its input itself (a.gn_scale / a.gn_shift)
void conv2d_split(const ConvArgs& a, int nt, int ksplit, int structure, hipStream_t s);
double conv2d_flops(const ConvArgs& a);
size_t conv2d_scratch_bytes(const ConvArgs& a);   // workspace conv2d() can use for these arguments (0 = none)
// N-tile width (in 32-column units) and split-K factor the split-precision path runs this problem with; nt_default is the
// width the generic cost model (conv.hip pick_nt) would take
void conv2d_split_plan(const ConvArgs& a, int nt_default, int* nt, int* ksplit, int* structure = nullptr, bool allow_split = true);
const char* conv2d_variant(const ConvArgs& a);
void conv2d_plan_query(const ConvArgs& a, int* structure, int* nt, int* ksplit);   // what conv2d() would launch   // name of the kernel conv2d() will launch for these arguments
void pack_ohwi(const float* w_oihw, float* w_ohwi, int Cout, int Cin, int ks, hipStream_t s);
// conv_wino.hip: 3x3 stride-1 convolution as F(2,3) along the width (1.5x fewer MFMAs), bf16x6 arithmetic
bool conv2d_wino_shape_ok(const ConvArgs& a);      // could run there if it had packed weights
bool conv2d_wino_eligible(const ConvArgs& a);      // shape_ok and a.w_wino present
bool conv2d_wino_worthwhile(const ConvArgs& a);    // shape_ok and a grid big enough to beat the direct kernel (the planner's rule)
size_t wino_packed_bytes(int Cout, int Cin);
// peak.hip: bare MFMA loops (diagnostic)
int64_t mfma_peak_src_bytes();
void mfma_peak_fill(void* src, bool zero, hipStream_t s);
double mfma_peak_launch(int variant, const void* src, float* sink, int workgroups, int loops, hipStream_t s);
void wino_pack_weights(const float* w_ohwi, int Cout, int Cin, void* packed, hipStream_t s);
void conv2d_wino(const ConvArgs& a, hipStream_t s);
int conv2d_wino_stats_chunks(const ConvArgs& a);

// ---------------------------------------------------------------- norm.hip
int gn_nchunks(int HW, int C);
// Per-column statistics of the channels [c0, c0+c) of a tensor: p[n][chunk][col][2] doubles (sum, sumsq over the chunk's
// pixels), `chunks` chunks per sample.  Written by gn_stats, by avg_into_stats and by the convolution epilogues.
struct GnSrc {
    const double* p = nullptr;
    int chunks = 0, c0 = 0, c = 0;
};
// partial: [N][nchunk][C][2] doubles
void gn_stats(const float* x, int N, int HW, int C, double* partial, int nchunk, hipStream_t s);
// scale/shift: [N][C] so that GN(x) = x*scale + shift; film (optional, [N][film_stride] = scale|shift halves):
// (GN(x))*(1+fs)+fsh folded in (ResBlock use_scale_shift_norm, openaimodel.py:276-280).  s0 covers channels [0, s0.c),
// s1 (optional: p == nullptr) the rest — a concatenated tensor has one source per part.
void gn_finalize(const GnSrc& s0, const GnSrc& s1, int N, int HW, int C, const float* gamma, const float* beta,
                 float eps, const float* film, int film_stride, float* scale, float* shift, hipStream_t s);
// whole GroupNorm32 (+ FiLM, + activation) of a small map in one launch: one workgroup per (group, sample)
bool gn_small_ok(int HW, int C);
void gn_small(const float* x, int N, int HW, int C, const float* gamma, const float* beta, float eps, const float* film,
              int film_stride, int act, float* y, hipStream_t s);
// avg_into (misc.hip) on [N][HW][C] sources + the per-column statistics of what it wrote: partial [N][gn_nchunks(HW,C)][C][2]
void avg_into_stats(const float* a, const float* b, const float* c, const float* d, float div, int N, int HW, int C,
                    float* dst, int dstC, int coff, int act, int bmask, double* partial, int nchunk, hipStream_t s);
enum { ACT_NONE = 0, ACT_SILU = 1 };
void affine_act(const float* x, int N, int HW, int C, const float* scale, const float* shift, int act, float* y,
                hipStream_t s);
void layer_norm(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps, float* y,
                hipStream_t s);

// ---------------------------------------------------------------- attention.hip
struct AttnArgs {
    const float *q = nullptr, *k = nullptr, *v = nullptr;  // rows of ld* floats, head h at +h*hs*
    int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    int q_hs = 0, k_hs = 0, v_hs = 0;
    int N = 0, Tq = 0, Tk = 0, heads = 0, d = 0;
    float scale_q = 1.f, scale_k = 1.f, scale_s = 1.f;
    float* out = nullptr;  // [N][Tq][ldo], head h at +h*d
    int split = 0;         // 1: both products on the bf16 matrix cores with operands split into 3 bf16 pieces (bf16x6), 0: fp32 MFMA
};
void attention(const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- gemm16.hip / attention16.hip
// Single-product half-precision arithmetic of the transformer backbone (PREC_F16 / PREC_BF16: 16-bit operands, one MFMA per
// product, fp32 accumulation) — what the reference's DiT runs under fp16 autocast (BASELINE configs[4]).
enum { EPI16_STORE = 0, EPI16_GELU = 1, EPI16_GATED = 2 };
struct Gemm16Args {
    const void* x = nullptr;   // [M][ldx] 16-bit activations
    int ldx = 0;
    const void* w = nullptr;   // [N][K] 16-bit weights (nn.Linear layout)
    const float* bias = nullptr;
    int M = 0, N = 0, K = 0;
    int bf16 = 0;              // 0: fp16, 1: bf16
    int epi = EPI16_STORE;
    void* y16 = nullptr;       // EPI16_STORE / EPI16_GELU: [M][ldy] 16-bit
    int ldy = 0;
    float* x32 = nullptr;      // EPI16_GATED: x32[m][n] += gate[(m / T) * gate_stride + n] * round16(acc + bias)
    int ldx32 = 0;
    const float* gate = nullptr;
    int gate_stride = 0, T = 1;
    int qcols = 0;             // EPI16_STORE: columns < qcols are multiplied by qscale in fp32 before the rounding
    float qscale = 1.f;
};
bool gemm16_shape_ok(int M, int N, int K);
void gemm16(const Gemm16Args& a, hipStream_t s);
void gemm16_whatif(const Gemm16Args& a, int whatif, hipStream_t s);   // diagnostic instantiations (timing only), -1 = the product kernel
void cast16(const float* x, int64_t n, void* y, int bf16, hipStream_t s);      // fp32 -> fp16 / bf16, round to nearest even
void uncast16(const void* x, int64_t n, float* y, int bf16, hipStream_t s);
// ln_modulate (misc.hip) with a 16-bit result: the A operand of the following Linear
void ln_modulate16(const float* x, int N, int T, int C, const float* mod, int mod_stride, int shift_off, int scale_off, float eps,
                   void* y, int bf16, hipStream_t s);
struct Attn16Args {
    const void *q = nullptr, *k = nullptr, *v = nullptr;   // 16-bit rows of ld* elements, head h at +h*hs*
    int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    int q_hs = 0, k_hs = 0, v_hs = 0;
    int N = 0, Tq = 0, Tk = 0, heads = 0, d = 0;
    float scale_q = 1.f;   // q * scale_q (fp32, rounded back) — 1 when q arrives pre-scaled by hd^-1/2 * log2(e); scores are base-2 logits
    float thr = -1.f;      // running-maximum threshold in log2 units (< 0: default 8; 0: move the maximum on every increase)
    int bf16 = 0;
    void* out = nullptr;   // [N][Tq][ldo] 16-bit, head h at +h*d
};
bool attention16_shape_ok(int d);
void attention16(const Attn16Args& a, hipStream_t s);
void attention16_whatif(const Attn16Args& a, int whatif, hipStream_t s);   // diagnostic instantiations (timing only), -1 = the product kernel

// ---------------------------------------------------------------- misc.hip
void timestep_embedding(const void* t, int t_is_float, int N, int dim, float* y, hipStream_t s,
                        const float* freqs = nullptr);
void linear(const float* x, int N, int K, int ldx, const float* w, const float* bias, int O, int act_in, float* y,
            int ldy, hipStream_t s);
void se_scale(const float* x, int N, int HW, int C, const float* w1, const float* w2, int Cr, float* y, hipStream_t s);
// dst[p, coff:coff+C] = act((a+b+c+d) * scale), sources NHWC [pixels][C]
// bmask bit k: source k is ONE sample ([per_sample_pixels][C]) broadcast over the batch of dst
void avg_into(const float* a, const float* b, const float* c, const float* d, float scale, int64_t pixels, int C,
              float* dst, int dstC, int coff, int act, hipStream_t s, int64_t per_sample_pixels = 0, int bmask = 0);
void nchw_to_nhwc(const float* x, int N, int C, int HW, float* y, hipStream_t s);
void nhwc_to_nchw(const float* x, int N, int C, int HW, float* y, hipStream_t s);
void avgpool2(const float* x, int N, int H, int W, int C, float* y, hipStream_t s);
void upsample2(const float* x, int N, int H, int W, int C, float* y, hipStream_t s);
void geglu(const float* x, int64_t rows, int inner, float* y, hipStream_t s);
// ---- DiT pieces (UNet_DS_Diff/DiT_models.py)
// LayerNorm without affine (eps) then modulate: y[n,t,:] = ((x - mean) * rstd) * (1 + scale[n,:]) + shift[n,:]; mod rows have
// stride mod_stride, shift at +shift_off, scale at +scale_off   (DiTBlock.forward :119-121, FinalLayer.forward :139-141)
void ln_modulate(const float* x, int N, int T, int C, const float* mod, int mod_stride, int shift_off, int scale_off, float eps,
                 float* y, hipStream_t s);
// x[n,t,:] += gate[n,:] * y[n,t,:]   (gate = mod + gate_off, the adaLN-Zero residual :120-121); in place on x
void gated_residual(float* x, const float* y, int N, int T, int C, const float* mod, int mod_stride, int gate_off, hipStream_t s);
// in place: GELU(approximate="tanh")
void gelu_tanh(float* x, int64_t n, hipStream_t s);
// x [N,C,H,W] -> tokens [N, (H/p)*(W/p), C*p*p] in Conv2d weight order (c, ph, pw)   (timm PatchEmbed = Conv2d(k = stride = p))
void patchify(const float* x, int N, int C, int H, int W, int p, float* y, hipStream_t s);
// tokens [N, h*w, p*p*c] (order ph, pw, c) -> [N, c, h*p, w*p]   (DiT.unpatchify :209-222)
void unpatchify(const float* x, int N, int c, int h, int w, int p, float* y, hipStream_t s);
// x[n,t,:] += pos[t,:]
void add_rows_broadcast(float* x, const float* pos, int N, int64_t TC, hipStream_t s);
// y[n,:] = (a ? a[n,:] : 0) + table[idx[n], :]   (LabelEmbedder: c = t_emb + y_emb :241-244); idx int64 on device
void embed_add(const float* a, const float* table, const long long* idx, int N, int C, float* y, hipStream_t s);
// in place: s[r][:] = softmax(s[r][:] * scale) over `cols` contiguous fp32 (one workgroup per row; max-subtracted, fp32)
void softmax_rows(float* s, int64_t rows, int cols, float scale, hipStream_t st);
void add2(const float* a, const float* b, int64_t n, float* y, hipStream_t s);

// ---------------------------------------------------------------- sampler.hip
struct StepCoef {
    float c[8];
    int mode, pred, learned_range, clip, nonzero;
    float eta;
};
// model_out: [B,Cm,HW] (Cm = 1, or 2 with learned_range); x in/out [B,1,HW]; noise [B,1,HW] or null (Philox)
// slice_ids (optional, device [B]): global slice index of every batch row — the Philox counter of element p of row b is
// slice_ids[b]*HW + p instead of b*HW + p, so a slice's noise does not depend on how the volume was sharded or batched
void sampler_update(const StepCoef& sc, const float* model_out, float* x, const float* noise, uint64_t seed,
                    uint64_t step, int B, int HW, hipStream_t s, float* x0_out = nullptr, const int64_t* slice_ids = nullptr);
// DPM-Solver(++) multistep: coefficients of one network evaluation + update (host tables, include/dsdiff.h dsd_dpm_schedule)
struct DpmCoef {
    float alpha, sigma;      // marginal alpha_t, sigma_t at the evaluation time
    float cx, cm, cd, ir0;   // x <- (cx*x - cm*m0) - cd*(ir0*(m0 - m1))
    int order;               // 0: x <- m0 (denoise to zero), 1, 2
    int pred;                // DSD_PRED_* of the network
    int data_pred, thresh;
};
void dpm_step(const DpmCoef& c, const float* model_out, int Cm, float* x, float* m_cur, const float* m_prev, float* s_buf,
              float ratio, float max_val, int B, int HW, hipStream_t s);
void dpm_threshold(const float* x0, float* y, float* s_buf, float ratio, float max_val, int B, int n, hipStream_t s);
void philox_normal(float* y, int64_t n, uint64_t seed, uint64_t step, hipStream_t s);
// DiagonalGaussianDistribution.sample (ldm/modules/distributions/distributions.py:24-37): moments [B,2E,HW] (NCHW) ->
// z = mean + exp(0.5 * clamp(logvar, -30, 20)) * eps, eps = noise[B,E,HW] or Philox normals (noise == nullptr)
void gaussian_sample(const float* moments, const float* noise, uint64_t seed, int B, int E, int HW, float* z, hipStream_t s);
void fill_t(float* t, int B, float v, hipStream_t s);

}  // namespace dsd
