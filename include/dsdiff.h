/*
 * dsdiff.h — C ABI of libdsdiff.so: the MI355X (gfx950) implementation of the conditional-DDPM
 * sampling hot path of larrybb626/diffusion_models_dsdiff.
 *
 * Plain C: pointers, sizes, ints.  No torch types.  Every tensor pointer is a DEVICE pointer to
 * contiguous fp32 unless stated otherwise; the caller owns inputs/outputs, the library owns the
 * packed weights and the workspace.  Every entry point returns 0 on success, <0 on error
 * (dsd_last_error() gives the message, thread-local).  All work is enqueued on the caller's
 * hipStream_t (passed as void*; NULL = default stream) and is asynchronous w.r.t. the host.
 * Calls on one handle are not re-entrant.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   dsd_create / dsd_set_param     <- DSUnetModel.__init__           UNet_DS_Diff/model.py:172-611
 *                                     + nn.Module.load_state_dict (names = reference state_dict keys)
 *   dsd_forward                    <- DSUnetModel.forward            UNet_DS_Diff/model.py:629-756
 *   dsd_sample                     <- GaussianDiffusion.p_sample_loop / ddim_sample_loop
 *                                        Disc_diff/guided_diffusion/gaussian_diffusion.py:524-616,705-786
 *                                     DDPMModel.p_sample_loop        trainers/trainer_ddpm.py:447-482
 *                                     DDIMSampler.ddim_sampling      ldm/models/diffusion/ddim.py:128-261
 *   dsd_block_*                    <- ResBlock / AttentionBlock / Upsample / Downsample
 *                                        ldm/modules/diffusionmodules/openaimodel.py:93-164,167-284,426-473
 *                                     FeatureDisentangle             UNet_DS_Diff/model.py:152-168
 *                                     SE_Attention                   Disc_diff/guided_diffusion/unet.py:82-109
 *                                     CrossAttention / BasicTransformerBlock / SpatialTransformer
 *                                        ldm/modules/attention.py:145-194,302-331,366-428
 *   dsd_op_*                       <- the stock torch ops those modules call (kernel-level tests)
 */
#ifndef DSDIFF_H
#define DSDIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsd_handle dsd_handle;

#define DSD_MAX_LEVELS 8

/* ctor kwargs of DSUnetModel that shape the graph (UNet_DS_Diff/model.py:172-202). */
typedef struct dsd_config {
    int32_t in_channels;        /* per-stream input channels (yaml: 1) */
    int32_t model_channels;
    int32_t out_channels;
    int32_t n_levels;           /* len(channel_mult) */
    int32_t channel_mult[DSD_MAX_LEVELS];
    int32_t num_res_blocks[DSD_MAX_LEVELS];
    int32_t n_attention_resolutions;
    int32_t attention_resolutions[DSD_MAX_LEVELS];
    int32_t num_heads;          /* -1 if unset */
    int32_t num_head_channels;  /* -1 if unset */
    int32_t num_heads_upsample; /* -1 if unset */
    int32_t use_scale_shift_norm;
    int32_t resblock_updown;
    int32_t use_new_attention_order;
    int32_t legacy;
} dsd_config;

/* Error text of the last failing call on this thread ("" if none). */
const char* dsd_last_error(void);
/* Library / device probe: returns 0 and fills name (<=255 chars) if a gfx950 device is usable. */
int dsd_device_info(int device, char* name, int name_len, int* n_cu, int64_t* hbm_bytes);

/* ---- model handle ------------------------------------------------------------------------ */
/* device < 0 creates a table-only handle (parameter names/shapes, no GPU needed; compute calls fail). */
int dsd_create(const dsd_config* cfg, int device, dsd_handle** out);
void dsd_destroy(dsd_handle* h);

/* Parameter table (names/shapes identical to the reference state_dict, OIHW conv weights). */
int dsd_param_count(dsd_handle* h);
int dsd_param_info(dsd_handle* h, int idx, const char** name, int64_t shape[4], int* ndim);
/* Upload one parameter. src is fp32 in the reference layout; src_is_device selects the copy kind.
 * 3x3 conv weights are re-packed OIHW -> OHWI on device.  With src_is_device the copy is asynchronous on `stream`;
 * later plan-time work of the library (splitting the weights into bf16 / fp16 pieces, on whatever stream the planning
 * call is given) waits on an event recorded here, so uploads on a non-blocking stream need no extra synchronisation. */
int dsd_set_param(dsd_handle* h, const char* name, const float* src, const int64_t* shape, int ndim,
                  int src_is_device, void* stream);
/* Optional: the sinusoidal-embedding frequency table exp(-ln(1e4)*k/half), k < model_channels/2
 * (util.py:172-174), as evaluated by the caller's own fp32 exp.  sin/cos of t*f with t up to 1e3 amplify a
 * 1-ulp difference in f to ~6e-5, so a host that wants bit-identical arguments supplies its table; without
 * it the library evaluates exp in fp64 and rounds once. */
int dsd_set_timestep_freqs(dsd_handle* h, const float* freqs_host, int n);
/* Arithmetic of the convolutions (99 % of the FLOPs): DSD_PREC_F32 = v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma
 * chain); DSD_PREC_BF16X6 = every fp32 operand split exactly into three bf16 pieces, six bf16 MFMA products with fp32
 * accumulation (drops only terms <= 2^-24: fp32-grade, ~1e-6 on the network output); DSD_PREC_BF16X3 = two pieces, three
 * products (~1.5e-5 on the network output, still inside the 1e-4 bar of the sampled image).
 * DSD_PREC_F16X3 = two fp16 pieces (11 + 11 bits), three products: ~1.4x the fp32 kernel's error at 1.5x the speed of
 * bf16x6, but fp16's range: if a conv operand exceeds 65504 the call FAILS (device flag, checked with one stream
 * synchronisation at the end of dsd_forward / dsd_sample) instead of returning inf/NaN, and operands that are uniformly
 * tiny lose accuracy (absolute floor 3e-8: |x|~1e-3 -> 2e-5 relative).  Meant for GroupNorm-normalised activations.
 * Default: DSD_PREC_BF16X6 (all parity tests hold at the fp32 tolerances); shapes the split kernel cannot take
 * (Cin % 32 != 0, operands >= 4 GiB) use the fp32 kernels in every mode.
 * DSD_PREC_F16 / DSD_PREC_BF16 (DSD_BLOCK_DIT handles only; any other handle refuses them): the reference's autocast
 * arithmetic for the transformer backbone (BASELINE configs[4] "512x512 fp16"; UNet_DS_Diff/DiT_models.py:101-122 under
 * torch.autocast) — the four Linears of every DiTBlock and both attention products take 16-bit operands rounded ONCE
 * (round to nearest even) and run ONE MFMA per product with fp32 accumulation; LayerNorm, adaLN modulation, softmax
 * statistics, the gated residual stream, patch embedding, conditioning MLPs and the final layer stay fp32 as autocast keeps
 * them.  fp16 has fp16's range (activations beyond 65504 become inf exactly as in the reference); bf16 does not. */
enum { DSD_PREC_F32 = 0, DSD_PREC_BF16X3 = 1, DSD_PREC_BF16X6 = 2, DSD_PREC_F16X3 = 3, DSD_PREC_F16 = 4, DSD_PREC_BF16 = 5 };
int dsd_set_precision(dsd_handle* h, int precision);
int dsd_get_precision(dsd_handle* h);
/* dsd_sample only, OFF by default.  In the 2-channel branch (model.py:654-658) the `al` and `l` encoder streams get
 * zeros_like(x) as input and, inside a sampling loop, the same timestep for every slice: their activations are identical
 * across the batch.  With this option they are evaluated once per step (batch of one) and broadcast where they join the
 * other streams — the same output (bit-identical when the per-layer kernel choice does not change with the batch,
 * otherwise equal to fp32 rounding, ~1e-7), 28 % fewer FLOPs at batch 16.  It is work the reference performs redundantly, so
 * it is never enabled silently and bench.py reports it as a separate line; dsd_plan_flops() counts what is executed. */
int dsd_set_share_zero_streams(dsd_handle* h, int on);
/* GroupNorm statistics (ON by default): every tensor a GroupNorm32 normalises is written by a convolution, or is the
 * concatenation of a convolution output and the skip average (model.py:743-746), so the per-channel sums are accumulated
 * in the epilogue of the kernel that writes the tensor and the separate statistics pass over HBM disappears
 * (openaimodel.py:264-284: "fused GroupNorm" of the north star).  0 = always run the standalone pass (A/B, tests). */
int dsd_set_fuse_gn_stats(dsd_handle* h, int on);
/* GroupNorm + SiLU in front of a 3x3 convolution (ResBlock in_layers / out_layers, openaimodel.py:264-284), ON by default in
 * the bf16x6 mode: the large layers — those the tap-reuse kernel takes — apply the normalisation while they stage their input
 * rows into LDS, so the apply pass (one read + one write of the tensor through HBM per GroupNorm) disappears.  An element is
 * then activated once per filter row and column tile (6x on a 320-channel layer) in VALU slots of the convolution, with the
 * hardware exp2 / reciprocal (~1 ulp each) instead of expf and an IEEE division: the outputs agree with the apply-pass route
 * to fp32 rounding (measured 1e-7 relative on the network output), not bit for bit.  0 = always the separate pass. */
int dsd_set_fuse_gn_apply(dsd_handle* h, int on);
/* The four encoder streams of DSUnetModel (noise, a, al, l: UNet_DS_Diff/model.py:674-686) are independent chains until the
 * skip average.  ON by default: from the first encoder block whose input has at most `max_pixels` pixels (batch x H x W;
 * <= 0 keeps the current value, default 16384) the four streams are launched on four HIP streams (the caller's + three of
 * the library's, forked and joined through events inside dsd_forward / dsd_sample, also under hipGraph capture), so that
 * four grids too small to fill 256 CUs share the chip instead of queueing.  The convolutions of those levels are PLANNED for
 * a quarter of the chip each (tile width / split-K: a layer that would take narrow tiles alone takes the wide tap-reuse kernel
 * with the fused GroupNorm when its three siblings run beside it), so on = 1 and on = 0 agree to fp32 rounding, not bit for
 * bit; on = 2 keeps the lanes' plan and launches it on the caller's stream alone — bit-identical to on = 1 (the check that no
 * dependency between the lanes is missing).  The caller sees one stream: work submitted to it after the call is ordered
 * after the join. */
int dsd_set_stream_lanes(dsd_handle* h, int on, int max_pixels);
/* bf16x6 mode only, OFF by default: 3x3 stride-1 convolutions whose grid fills the chip at least twice (>= 512 workgroups),
 * with an output width that is a power of two <= 256, Cin % 32 == 0 and Cout = 0 or 64 (mod 128), run as Winograd F(2,3)
 * ALONG THE WIDTH — 4 instead of 6 products per pair of outputs and filter row, i.e. 1.5x fewer MFMAs.  The input transform
 * (sums / differences of neighbouring pixels) is done in fp32, the weight transform once in fp64, and both operands are then
 * split into three bf16 pieces exactly as in the direct kernel, so every operand still carries 24 significant bits; the
 * result differs from the direct form by an fp32 re-association of the convolution sum (measured: 0.75x the direct
 * kernel's error against fp64, tests/test_ops_gpu.py).  Opt-in because it is only 6-20 % faster than the direct kernel on
 * the layers it takes (its operand feeding costs what the saved MFMAs gain: conv_wino.hip header) — 2 % on a whole
 * denoising step — and stays below the 300 TF/s go / no-go set for it. */
int dsd_set_winograd(dsd_handle* h, int on);
/* 0 if every parameter has been set, else -1 with the first missing name in dsd_last_error(). */
int dsd_params_ready(dsd_handle* h);

/* Build (or re-use) the launch plan + workspace for an input shape.  Allocation happens here,
 * never inside dsd_forward/dsd_sample once the plan for that shape exists. */
int dsd_plan(dsd_handle* h, int B, int C, int H, int W);
int64_t dsd_workspace_bytes(dsd_handle* h);
/* Everything the handle holds on the device right now: parameter slab, upload staging, workspace arena, the bf16 / fp16
 * weight pieces of the split arithmetic modes (6 B per convolution weight; only the family of the current mode is kept),
 * sampling scratch. */
int64_t dsd_device_bytes(dsd_handle* h);
/* Number of kernel launches in the current plan's forward. */
int dsd_plan_launches(dsd_handle* h);
/* Algorithmic FLOPs (2*MAC of conv/linear/attention matmuls) of one forward of the current plan:
 * executed = what the kernels really compute. */
double dsd_plan_flops(dsd_handle* h);

/* One denoising-network evaluation.  x: [B,C,H,W] (C in {2,4}, NCHW as the reference passes it),
 * t: [B] int64 (t_is_float=0) or fp32 (t_is_float=1) on device, out: [B,out_channels,H,W].
 * feats: NULL, or 14 device pointers to [B,half,H/2^(L-1),W/2^(L-1)] NCHW buffers in the order
 * style[3], content[3], anatomy[2], lesion[2], n_style_content[4] (model.py:751-756). */
int dsd_forward(dsd_handle* h, const float* x, const void* t, int t_is_float, int B, int C, int H, int W,
                float* out, float* const* feats, void* stream);

/* Per-kernel timing of the plan, measured with hipEvents on the stream the kernels are launched on (one
 * pair per op; forces a stream sync per forward, so never leave it on in a throughput run).  Totals
 * accumulate over forwards until re-enabled.  kind = kernel name, flops / bytes = the algorithmic work of
 * those launches (2*MAC; fp32 read+write bytes for the memory-bound kernels). */
int dsd_profile_enable(dsd_handle* h, int on);
int dsd_profile_count(dsd_handle* h);
int dsd_profile_get(dsd_handle* h, int idx, const char** kind, double* total_ms, double* flops, double* bytes,
                    int64_t* calls, int* runs);
/* The same per op of the plan, in launch order, for the last profiled forward (kind = the op's kernel kind). */
int dsd_profile_op_count(dsd_handle* h);
int dsd_profile_op_get(dsd_handle* h, int idx, const char** kind, double* ms, double* flops, double* bytes);
/* The layer op `idx` of the current plan belongs to: the name of the last parameter the graph builder looked up before it
 * emitted the op ("input_blocks.4.0.in_layers.2.bias", ...; "" for ops in front of the first parameter). */
int dsd_profile_op_name(dsd_handle* h, int idx, const char** name);

/* ---- sampling loop ----------------------------------------------------------------------- */
/* Whole-forward hipGraph replay inside dsd_sample / dsd_sample_dpm (OFF by default): the network evaluation of a step is
 * a fixed list of ~800 launches whose arguments do not change while the loop runs on the same buffers, so it can be captured
 * once (after the plan's first host-launched forward) and replayed per step; the reference's Python loop
 * (gaussian_diffusion.py:569-616) pays the launches every step.  Results are bit-identical either way.  Measured on MI355X:
 * the asynchronous host launches already keep the GPU busy (batch 16: 443.7 vs 445.4 ms per step, batch 1: 39.0 vs 40.3), so
 * replay only takes the launch work off the host thread; hence opt-in.  dsd_forward and dsd_profile_* always launch from
 * the host. */
int dsd_set_graph(dsd_handle* h, int on);
int dsd_graph_stats(dsd_handle* h, int* captures, int* launches);
/* Optional: the global slice index of every row of the next sampling batches (host array, n = batch size; n = 0 clears).
 * With it the on-device Philox noise of a slice is keyed by (seed, step, slice index, pixel) instead of its position in
 * the batch, so a volume gives the same samples however its slices are sharded over GPUs or grouped into batches. */
int dsd_set_slice_ids(dsd_handle* h, const int64_t* ids_host, int n);
enum { DSD_MODE_A_DDPM = 0, DSD_MODE_A_DDIM = 1, DSD_MODE_B_DDPM = 2, DSD_MODE_B_DDIM = 3 };
enum { DSD_PRED_EPS = 0, DSD_PRED_X0 = 1, DSD_PRED_V = 2 };
#define DSD_NCOEF 8
/* Host-side schedule for `steps` loop iterations, iteration k = 0 is the FIRST executed (largest t).
 * coef[k*DSD_NCOEF + j] fp32, meaning per mode (tables are produced in float64 on the host exactly as
 * the reference does and rounded to fp32 once, gaussian_diffusion.py:1003 / ddpm.py:155-178):
 *   A (guided-diffusion):  0 sqrt_acp  1 sqrt_1m_acp  2 sqrt_recip_acp  3 sqrt_recipm1_acp
 *                          4 post_coef1  5 post_coef2  6 log_var(fixed) or min_log(learned range)  7 max_log
 *        A_DDIM re-uses 0-3 and:  4 alpha_bar  5 alpha_bar_prev
 *   B_DDPM (ldm):          0 sqrt_acp 1 sqrt_1m_acp 2 sqrt_recip_acp 3 sqrt_recipm1_acp 4 coef1 5 coef2 6 post_log_var
 *   B_DDIM (ldm):          0 sqrt_acp 1 sqrt_1m_acp 4 a_t 5 a_prev 6 sigma_t 7 sqrt_1m_at
 * t_model[k]: the timestep value handed to the network at iteration k (timestep_map[t], optionally
 * rescaled, respace.py:123-128); nonzero[k]: 0 only where the reference masks the noise (t == 0). */
typedef struct dsd_schedule {
    int32_t steps;
    int32_t mode;          /* DSD_MODE_* */
    int32_t pred;          /* DSD_PRED_* */
    int32_t learned_range; /* model emits 2*C channels, LEARNED_RANGE variance (gaussian_diffusion.py:280-294) */
    int32_t clip_denoised;
    float eta;             /* DDIM eta (A_DDIM computes sigma on device from alpha_bar, alpha_bar_prev) */
    const float* coef;     /* host, steps*DSD_NCOEF */
    const float* t_model;  /* host, steps */
    const int32_t* nonzero;/* host, steps */
} dsd_schedule;

/* Runs the whole loop on the device.  x: [B,1,H,W] in = x_T, out = x_0 (in place).  cond: [B,Cc,H,W]
 * (Cc in {1,3}) concatenated after x every step (DiffusionWrapper 'concat', ddpm.py:1331-1333).
 * noise: NULL -> on-device Philox4x32-10 N(0,1) from philox_seed; else [steps,B,1,H,W] pre-drawn
 * normals, noise[k] used at iteration k (parity mode).  first_step/n_steps select a sub-range of
 * iterations (n_steps<=0 = all) so a caller can time or checkpoint part of the chain. */
int dsd_sample(dsd_handle* h, const dsd_schedule* sched, const float* cond, int Cc, float* x, const float* noise,
               uint64_t philox_seed, int B, int H, int W, int first_step, int n_steps, void* stream);
/* The fused sampler update alone (iteration k): model_out [B,Cm,H,W]; x updated in place; pred_xstart
 * (optional, [B,1,H,W]) receives the clipped x_0 prediction (the reference's out["pred_xstart"]). */
int dsd_op_sampler_update(const dsd_schedule* sched, int k, const float* model_out, float* x, const float* noise,
                          uint64_t philox_seed, int B, int H, int W, float* pred_xstart, void* stream);

/* ---- DPM-Solver(++) multistep sampler -----------------------------------------------------
 * Replaces DPM_Solver(...).sample(method="multistep", order<=2) of Disc_diff/guided_diffusion/sampler.py:1017-1222
 * (call site GaussianDiffusion.dpm_solver_sample_loop, gaussian_diffusion.py:467-522) and of its twin
 * ldm/models/diffusion/dpm_solver_new/dpm_solver_pytorch.py (call site DPMSolverSampler.sample, sampler.py:86-101).
 * Iteration k evaluates the network once at model time t_input[k] and applies one update:
 *   m_k  = data prediction (x - sigma_k*eps)/alpha_k [data_pred=1, "dpmsolver++"] or eps [data_pred=0, "dpmsolver"],
 *          eps from the network output per `pred` (model_wrapper, sampler.py:247-265); with `thresholding` the data
 *          prediction is dynamically thresholded per sample (sampler.py:379-388)
 *   order[k] = 1:  x <- cx*x - cm*m_k                                   (dpm_solver_first_update :509-553)
 *   order[k] = 2:  x <- (cx*x - cm*m_k) - cd*(ir0*(m_k - m_{k-1}))      (multistep_dpm_solver_second_update :760-816)
 *   order[k] = 0:  x <- thresholded data prediction                     (denoise_to_zero_fn :503-507; forces data_pred)
 * coef[k*DSD_NCOEF + j]: 0 alpha_k  1 sigma_k  2 cx  3 cm  4 cd  5 ir0 (= 1/r0), all produced on the host with the
 * reference's fp32 torch expressions (NoiseScheduleVP, sampler.py:76-149). */
typedef struct dsd_dpm_schedule {
    int32_t steps;            /* network evaluations (= updates) */
    int32_t pred;             /* DSD_PRED_* : what the network predicts */
    int32_t data_pred;        /* 1 = dpmsolver++ (data prediction), 0 = dpmsolver (noise prediction) */
    int32_t thresholding;     /* dynamic thresholding of the data prediction */
    float threshold_ratio;    /* 0.995 */
    float threshold_max;      /* 1.0 */
    const float* coef;        /* host, steps*DSD_NCOEF */
    const float* t_input;     /* host, steps: fp32 model time (t - 1/N)*1000, sampler.py:236-245 */
    const int32_t* order;     /* host, steps */
} dsd_dpm_schedule;
/* x: [B,1,H,W] in = x_T, out = sample (in place); cond as in dsd_sample.  A 2-channel (learned-sigma) network
 * contributes its first channel only (gaussian_diffusion.py:484-485). */
int dsd_sample_dpm(dsd_handle* h, const dsd_dpm_schedule* sched, const float* cond, int Cc, float* x, int B, int H, int W,
                   void* stream);
/* Iteration k's post-network part alone: model_out [B,Cm,H,W], x updated in place, m_cur [B,1,H,W] receives m_k,
 * m_prev = m_{k-1} (may be NULL when order[k] < 2). */
int dsd_op_dpm_step(const dsd_dpm_schedule* sched, int k, const float* model_out, int Cm, float* x, float* m_cur,
                    const float* m_prev, int B, int H, int W, void* stream);
/* Dynamic thresholding alone: y = clamp(x0,-s,s)/s with s_b = max(quantile_ratio(|x0_b|), max_val); x0,y [B,n], s [B]. */
int dsd_op_dpm_threshold(const float* x0, int B, int n, float ratio, float max_val, float* y, float* s_out, void* stream);

/* ---- single blocks (own parameter namespace, names relative to the block) ----------------- */
enum { DSD_BLOCK_RES = 0, DSD_BLOCK_ATTN = 1, DSD_BLOCK_UPSAMPLE = 2, DSD_BLOCK_DOWNSAMPLE = 3,
       DSD_BLOCK_DISENTANGLE = 4, DSD_BLOCK_SE = 5, DSD_BLOCK_CROSSATTN = 6, DSD_BLOCK_FF_GEGLU = 7,
       DSD_BLOCK_BASIC_TRANSFORMER = 8, DSD_BLOCK_SPATIAL_TRANSFORMER = 9,
       /* latent path (SURVEY f-3): the KL-VAE of ldm/models/autoencoder.py:26-147 — Encoder / Decoder of
        * ldm/modules/diffusionmodules/model.py:452-655 with quant_conv / post_quant_conv; parameter names are the
        * AutoencoderKL state_dict's ("encoder.down.0.block.0.norm1.weight", "quant_conv.weight", ...) */
       DSD_BLOCK_VAE_ENCODER = 10, DSD_BLOCK_VAE_DECODER = 11,
       /* transformer backbone (SURVEY f-4): DiT of UNet_DS_Diff/DiT_models.py:145-262 (adaLN-Zero blocks, timm-style
        * PatchEmbed / Attention / Mlp); parameter names are the DiT state_dict's */
       DSD_BLOCK_DIT = 12,
       /* the plain single-stream UNetModel of ldm/modules/diffusionmodules/openaimodel.py:571-958 (no class embedding): the
        * denoiser that consumes the VAE latents in the latent path; optionally with a SpatialTransformer on `context` in
        * every attention slot (use_spatial_transformer=True, :761-765,818-822,872-876) */
       DSD_BLOCK_UNET = 13 };
/* iargs by kind:
 *   RES: cin, cout, emb_ch, use_scale_shift_norm, up, down      ATTN: ch, heads, new_order
 *   UPSAMPLE/DOWNSAMPLE: ch     DISENTANGLE: ch, half_ch       SE: ch, reduction
 *   CROSSATTN: query_dim, context_dim, heads, dim_head          FF_GEGLU: dim, mult
 *   BASIC_TRANSFORMER: dim, heads, dim_head, context_dim
 *   SPATIAL_TRANSFORMER: in_ch, heads, dim_head, depth, context_dim, use_linear
 *   VAE_ENCODER / VAE_DECODER: ch, out_ch, in_channels, resolution, z_channels, double_z, embed_dim, num_res_blocks,
 *        with_quant (1: AutoencoderKL.encode / decode incl. quant_conv / post_quant_conv, 0: bare Encoder / Decoder),
 *        len(ch_mult), ch_mult..., len(attn_resolutions), attn_resolutions...    (configs/autoencoder_kl_64x64x3.yaml:14-24)
 *        encoder: x [B,in_channels,H,W] -> moments [B,2*embed_dim,H/f,W/f];  decoder: z [B,embed_dim,h,w] -> [B,out_ch,h*f,w*f]
 *   DIT: input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_hidden, num_classes, learn_sigma,
 *        use_cfg_embedding.  x [B,in_channels,S,S] (x and cond already concatenated), aux = timesteps [B] fp32 (aux_len 1),
 *        aux2 = class labels [B] int64 or NULL (aux_len2 1 / 0); out [B, out_channels, S, S].  dsd_set_timestep_freqs
 *        (128 entries) installs the caller's frequency table as for the U-Net.
 *   UNET: in_channels, model_channels, out_channels, num_heads, num_head_channels, num_heads_upsample, use_scale_shift_norm,
 *        resblock_updown, use_new_attention_order, legacy, len(channel_mult), channel_mult..., num_res_blocks per level...,
 *        len(attention_resolutions), attention_resolutions...  (the ctor kwargs of openaimodel.py:601-633), optionally followed by
 *        use_spatial_transformer, transformer_depth (1), context_dim, use_linear_in_transformer.
 *        x [B,in_channels,H,W], aux = timesteps [B] fp32 (aux_len 1), aux2 = context [B,tokens,context_dim] with
 *        aux_len2 = tokens (spatial-transformer variant only, else NULL / 0); out [B,out_channels,H,W]. */
int dsd_block_create(int kind, const int32_t* iargs, int n_iargs, int device, dsd_handle** out);
/* x: NCHW [B,C,H,W] (token blocks: [B,N,C] passed as H=N, W=1 "NHWC"), aux: emb [B,emb_ch] for RES,
 * context [B,Nc,Cc] for cross-attention kinds (aux2/aux_len2 = second context for depth-2 spatial
 * transformer), else NULL.  out sized as the block's output. */
int dsd_block_forward(dsd_handle* h, const float* x, int B, int C, int H, int W, const float* aux, int aux_len,
                      const float* aux2, int aux_len2, float* out, void* stream);

/* ---- kernel-level ops (NHWC activations; used by tests and micro-benchmarks) -------------- */
/* y[N,OH,OW,Cout] = conv(x[N,H,W,Cin], w OIHW ks x ks, pad ks/2, stride) + bias (+ emb[N,Cout]) (+ res) ;
 * upsample=1 folds a nearest x2 in front of the conv (Upsample, openaimodel.py:111-121). */
int dsd_op_conv2d(const float* x, int N, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                  int ks, int stride, int upsample, const float* emb, const float* res, float* y, void* stream);
/* Same with an explicit arithmetic mode: 0 = fp32 MFMA, 1 = bf16x3, 2 = bf16x6 (fp32 operands split into bf16 pieces,
 * fp32 accumulation; conv_split.hip).  Shapes the split kernel cannot take fall back to fp32.  OR-ing 16 / 32 / 64
 * into `precision` forces the A-direct / fully staged / 256-row A-direct kernel structure (tests); otherwise the library
 * chooses; OR-ing 128 (with precision 2) runs the F(2,3) kernel of dsd_set_winograd and fails if the shape cannot take it. */
int dsd_op_conv2d_prec(const float* x, int N, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                       int ks, int stride, int upsample, const float* emb, const float* res, int precision, float* y,
                       void* stream);
/* Micro-benchmark of the convolution kernel on random data (library-owned buffers): average ms per launch over
 * `iters` back-to-back launches (hipEvents) and the algorithmic FLOPs of one launch.  variant: -1/0 default fp32
 * kernel, 1 flat-load fp32 kernel, 10 bf16x3, 11 bf16x6, 12 f16x3 (library's choice of structure), 20/21 both operands staged
 * through LDS, 30/31 activations read straight into registers (128-row tile), 40/41/42 the same on the 256-row tile,
 * 50 bf16x6 as F(2,3) along the width (FLOPs reported are those of the direct form: "fp32-equivalent"). */
int dsd_bench_conv2d(int N, int H, int W, int Cin, int Cout, int ks, int stride, int variant, int iters, float* avg_ms,
                     double* flops);
/* Diagnostic build of the dominant kernel (bf16x6, 256-row tile, Cout = 320 -> 160-column tiles): `warm` untimed launches
 * of the product kernel, then ONE launch of an instantiation that stamps (s_memtime, s_memrealtime) at entry, in front of
 * the k-loop, behind it and after the epilogue.  out: 8 int64 per workgroup (host memory, room for max_wgs workgroups);
 * *n_wgs = workgroups launched.  tools/conv_stamps.py turns this into prologue / loop / epilogue / gap times and the
 * in-kernel clock.  whatif != 0 runs an instantiation with one cost of the k-loop removed (output garbage; timing only):
 * 2 no activation loads, 4 no weight loads / LDS writes, 8 no barrier, 16 no weight fragment reads, 31 all of them and no
 * activation split (the bare MFMA stream of this kernel), 32 loads issued in bursts of eight (the schedule before round 2;
 * results stay correct), 256 activations read as pre-split bf16 planes (12 piece loads per tile, no split VALU), 512 the tap-reuse
 * instantiation the library uses for these layers by default (0 stamps the plain A-direct kernel).  Other values fail (conv_split.hip lists what else was measured). */
int dsd_bench_conv2d_stamps(int N, int H, int W, int Cin, int Cout, int warm, int whatif, long long* out, int max_wgs,
                            int* n_wgs);
/* What the bf16 matrix pipes of this device sustain: a bare v_mfma_f32_32x32x16_bf16 loop on the dominant convolution's
 * accumulator tile (one wave per SIMD, 160 accumulators, six products per group), `workgroups_per_cu` x 256 workgroups
 * (<= 0: 8), about `ms_target` milliseconds per launch, average of `iters` launches (hipEvents).  variant 0: operands in
 * registers (nothing but MFMAs: the upper bound at the clock the chip holds); 1: weight fragments re-read from LDS as the
 * convolution does; 2 / 3: the same two on all-zero operands; 4 .. 7: the same four with the work issued as
 * v_mfma_f32_16x16x32_bf16 (the same output tile per wave, the same matrix-pipe cycles: the two shapes can hold different
 * clocks under load).  *tflops = issued bf16 MFMA TFLOP/s.  (peak.hip) */
int dsd_bench_mfma_peak(int variant, int workgroups_per_cu, float ms_target, int iters, float* avg_ms, double* tflops);
/* How the library would run a convolution (host-side query, no GPU work): kernel structure (0 both operands staged
 * through LDS, 1 / 2 activations read straight into registers with a 128 / 256-row tile; -1 for the fp32 kernel), N-tile
 * width in 32-column units, split-K factor and the scratch bytes the split needs.  precision as in dsd_op_conv2d_prec;
 * + 256: planned as inside a stream-lane region (dsd_set_stream_lanes), i.e. for a quarter of the chip. */
/* Process-wide switch (experiment): the dominant convolution kernel (256 x 160 tile, bf16x6, tap reuse,
 * optional fused GroupNorm) with its matrix work issued as v_mfma_f32_16x16x32_bf16 (conv_tr16.hip) instead of
 * v_mfma_f32_32x32x16_bf16.  Same results up to fp32 summation order.  Default: the environment variable
 * DSD_CONV_MFMA16 (0).  Returns the previous setting. */
int dsd_set_conv_mfma16(int on);
int dsd_conv_plan(int N, int H, int W, int Cin, int Cout, int ks, int stride, int precision, int* structure, int* nt,
                  int* ksplit, uint64_t* scratch_bytes);
/* GroupNorm(32, C, eps) [+ SiLU] on x[N,HW,C]. */
int dsd_op_group_norm(const float* x, int N, int HW, int C, const float* gamma, const float* beta, float eps,
                      int silu, float* y, void* stream);
/* The model's last layer as one pass: y[N,H,W] = Conv3x3(SiLU(GroupNorm32(x))) with ONE output channel, x NHWC [N,H,W,C],
 * C a multiple of 64 up to 320, w_oihw [1,C,3,3] (UNet_DS_Diff/model.py:511-515 `self.out`, applied at :751). */
int dsd_op_gn_silu_conv_out1(const float* x, int N, int H, int W, int C, const float* gamma, const float* beta, float eps,
                             const float* w_oihw, const float* bias, float* y, void* stream);
/* QKVAttention / QKVAttentionLegacy (openaimodel.py:496-555) on qkv[N,T,3C] -> a[N,T,C].  split = 0: both products on
 * the fp32 matrix cores (what DSD_PREC_F32 runs); 1: operands split exactly into three bf16 pieces, six bf16 MFMA products
 * each (what every other mode runs); the softmax is fp32 either way. */
int dsd_op_qkv_attention(const float* qkv, int N, int T, int C, int heads, int new_order, int split, float* a, void* stream);
/* The half-precision kernels of DSD_PREC_F16 / DSD_PREC_BF16 on fp32 device buffers (the entry rounds the inputs to 16 bits
 * once, runs the kernel, and widens the 16-bit result): y[M,N] = x[M,K] w[N,K]^T + bias (nn.Linear; timm Mlp / Attention
 * projections, DiT_models.py:101-122).  epi 0: y = round16(acc + bias); 1: y = round16(gelu_tanh(round16(acc + bias)));
 * 2: y (fp32, in/out) += gate[(m / T), :] * round16(acc + bias) — the adaLN-Zero gated residual (:120-121). */
int dsd_op_gemm_half(const float* x, const float* w, const float* bias, int M, int N, int K, int bf16, int epi,
                     const float* gate, int T, float* y, void* stream);
/* Timing of the half-precision GEMM kernel on random operands (hipEvents, average of `iters` launches after one warm-up).
 * whatif < 0: the product kernel with epilogue `epi`; whatif >= 0: a diagnostic instantiation with costs removed (bits: 1 no
 * LDS-DMA staging in the loop, 2 fragments read from LDS only once, 4 no epilogue, 8 no barrier; instantiated: 0, 1, 2, 3, 4,
 * 7, 15) whose results are garbage — tools/gemm_whatif.py. */
int dsd_bench_gemm_half(int M, int N, int K, int bf16, int epi, int whatif, int iters, float* avg_ms);
/* The same for the half-precision attention kernel on a random qkv[N,T,3C].  whatif bits: 1 no softmax arithmetic, 2 no K / V
 * staging after the first tile, 4 no second product, 8 no first product (instantiated for fp16, head dim 64: 0, 1, 2, 3, 4, 8,
 * 13) — tools/attn_whatif.py. */
int dsd_bench_attention_half(int N, int T, int C, int heads, int bf16, int whatif, int iters, float* avg_ms);
/* softmax(q k^T * d^-1/2) v of timm Attention on qkv[N,T,3C] (q | k | v, heads inside each) -> a[N,T,C], 16-bit operands,
 * fp32 softmax statistics; thr: running-maximum threshold in log2 units (< 0: the library's default). */
int dsd_op_attention_half(const float* qkv, int N, int T, int C, int heads, int bf16, float thr, float* a, void* stream);
/* timestep_embedding (util.py:161-181): t[N] (int64 or fp32) -> [N,dim].  freqs (device, [dim/2], may be NULL): the
 * frequency table as the caller's own fp32 exp evaluates it (what dsd_set_timestep_freqs installs in a model handle);
 * with it the sin/cos arguments are bit-identical to the reference's. */
int dsd_op_timestep_embedding(const void* t, int t_is_float, int N, int dim, const float* freqs, float* y, void* stream);
/* y[N,O] = act_in(x[N,K]) @ w[O,K]^T + bias ; act_in: 0 none, 1 SiLU. */
int dsd_op_linear(const float* x, int N, int K, const float* w, const float* bias, int O, int act_in, float* y,
                  void* stream);
/* DiagonalGaussianDistribution(moments).sample() (ldm/modules/distributions/distributions.py:24-37): moments [B,2E,H,W]
 * -> z [B,E,H,W] = mean + exp(0.5*clamp(logvar,-30,20)) * eps; eps = noise [B,E,H,W] or on-device Philox normals. */
int dsd_op_gaussian_sample(const float* moments, const float* noise_or_null, uint64_t philox_seed, int B, int E, int H, int W,
                           float* z, void* stream);
/* Philox4x32-10 + Box-Muller stream used by dsd_sample when noise == NULL: fills n normals. */
int dsd_op_philox_normal(float* y, int64_t n, uint64_t seed, uint64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSDIFF_H */
