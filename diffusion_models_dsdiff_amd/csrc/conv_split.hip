// conv_split.hip — the same implicit-GEMM convolution on the bf16 matrix cores with fp32 operands SPLIT into bf16 pieces.
//
// gfx950 has no TF32/xf32; its fp32-input MFMA runs at 1/16 of the bf16 rate.  An fp32 value is written exactly as a sum
// of three bf16 values (8 + 8 + 8 significand bits): a = a1 + a2 + a3, w = w1 + w2 + w3.  With fp32 accumulation
//   bf16x6:  a1w1 + a1w2 + a2w1 + a2w2 + a1w3 + a3w1   drops only terms <= 2^-24 |a w|  -> fp32-grade result
//            (measured on the CPU oracle: 9e-7 rel-L2 on the full network's output, the same as fp32 re-ordering noise)
//   bf16x3:  a1w1 + a1w2 + a2w1                         drops terms ~2^-16..2^-17 |a w| -> 1.5e-5 on the network output
// at 6 (resp. 3) v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 v_mfma_f32_32x32x2_f32: 2.67x (5.3x) fewer matrix-pipe
// cycles.  Weights are split once at upload (split_weights); activations are split while they are staged into LDS.
//
// Structures measured (tools/bench_conv.py, tools/ab_conv.sh; 16x256x256x320->320, bf16x6 / bf16x3 TF/s of fp32-equivalent
// work): both operands staged through LDS 196 / 333 (conv_split_kernel); activations read straight into registers
// (conv_split_ad_kernel): single LDS stage, two barriers per k-tile 199 / 322 -> two weight stages, one barrier 205 ->
// staging software-pipelined into the MFMA stream, branch-free loop body 216-222 / 350 -> 256-row block tile, one workgroup
// per CU 221-229 / 378-384 (the default on the large layers).  Measured and removed: producer/consumer wave specialisation
// 165 / 326; a schedule pinned group by group without the pipelining 196; the 256-row kernel on the 16x16x32 MFMA shape
// 223 / 355 (no clock gain here, MI355X_MICROARCH.md "DVFS give-back" 7 notwithstanding); a third weight stage that moves
// the first fragment read of a tile in front of the barrier 220 / 355; a 128 x 320 tile (one activation load and split
// per k-tile for the whole output width, twice the LDS reads per MFMA) 210 / 368.
// PMC (profiles/r01_pmc.json): MFMA pipe busy 71 %, clock 1.80 GHz, no LDS bank conflicts; 6 x 226 = 1.36 PF/s issued =
// 54 % of the nominal bf16 peak, 72 % of what the matrix pipes deliver at the clock the chip holds under this load.
// Round 2, by in-kernel clock stamps (DIAG instantiations below, tools/conv_stamps.py; 16x256x256x320->320): a workgroup
// lives 254 us = prologue 4.3 + k-loop 245 + epilogue 4.7, successor starts 0.5 us after it; the k-loop runs 4945 cycles per
// k-tile against 3840 of MFMA issue (78 % dense) at 1.82 GHz.  One cost removed at a time (cycles per k-tile / loop us):
// no activation loads 4551 / 209, no weight loads + LDS writes 4506 / 227, no barrier 4604 / 238, no weight fragment reads
// 5002 / 245, all of them 3962 / 182; the activation loads made contiguous and cache-hot 4890 / 236 (the access pattern is
// not the cost, a vector-memory instruction is: ~50 cycles of the wave's only issue stream each).  The same tile as a BARE
// MFMA loop (peak.hip) runs 1.87-1.92 PF/s on random operands and 2.47 on zeros on the same device: the chip is power-
// bound, and a cycle saving comes back as clock — loads issued one at a time instead of in bursts of eight: 4946 -> 4763
// cycles (-3.7 %), clock 1.818 -> 1.766 GHz, loop time -1.3 % (kept: SPREAD).  Weight tile by LDS-DMA instead of through
// registers (DMA = true, DSD_CONV_DMA): no difference (228 vs 228 TF/s), the DMA piece costs the issue slots the
// register path's load + ds_write did.  Weight-tile loads non-temporal (what-if 128, so that they do not push the
// activation lines the next filter tap re-reads out of the 32 KB L1): clock 1.81 -> 1.78 GHz, +1.5 % time (the tiles then
// miss L2 as well).  Activations pre-split in memory as three bf16 planes (what-if 256: 12 piece loads per tile and lane
// instead of 8, no split VALU): 6187 cycles per k-tile, +15 % time — the VALU it saves was hidden, the loads it adds are not.
//
// Tiling is the fp32 kernel's (conv.hip): 256 threads, 128 x 32*NT x 32 block tile, wave = 32 rows x NT column tiles,
// buffer loads with hardware range checks, register prefetch of the next tile, fused bias/embedding/residual epilogue.
// LDS: one plane per piece, rows of 32 bf16 padded to 80 B (ds_read_b128 fragment reads conflict-free: lane = row,
// 8 consecutive k = 16 B, slot (5*row + const) mod 16).
#include "conv_split_kernels.inc"

namespace dsd {

// y = sum over the k-chunks of the partial tiles (fixed order: deterministic) + bias + embedding + residual
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int Cout,
                                                            const float* __restrict__ bias, const float* __restrict__ emb,
                                                            int emb_stride, int ohw, const float* __restrict__ res,
                                                            float* __restrict__ y, int y_ld, int out_nchw) {
    const int64_t total4 = (int64_t)M * Cout / 4;
    const int64_t plane = (int64_t)M * Cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int m = (int)(e / Cout), n = (int)(e - (int64_t)m * Cout);
        f32x4 v = *reinterpret_cast<const f32x4*>(partial + e);
        for (int c = 1; c < S; ++c) v += *reinterpret_cast<const f32x4*>(partial + c * plane + e);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        const int nb = m / ohw;
        if (emb) v += *reinterpret_cast<const f32x4*>(emb + (int64_t)nb * emb_stride + n);
        if (res) v += *reinterpret_cast<const f32x4*>(res + e);
        if (out_nchw) {
            const int r = m - nb * ohw;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[((int64_t)nb * Cout + n + k) * ohw + r] = v[k];
        } else {
            *reinterpret_cast<f32x4*>(y + (int64_t)m * y_ld + n) = v;
        }
    }
}

// w (fp32, [rows][K], K contiguous) -> planes[NP][rows][K] bf16, w = sum of the planes up to 2^-24 relative
__global__ void split_weights_kernel(const float* __restrict__ w, int64_t n, int np, int f16, unsigned short* __restrict__ planes,
                                     int* ovf) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float r = w[i];
        if (f16 && ovf && !(fabsf(r) <= 65504.f)) *ovf = 1;
        for (int q = 0; q < np; ++q) {
            if (f16) {
                const _Float16 b = (_Float16)r;
                planes[(int64_t)q * n + i] = __builtin_bit_cast(unsigned short, b);
                r -= (float)b;
            } else {
                const __bf16 b = (__bf16)r;
                planes[(int64_t)q * n + i] = __builtin_bit_cast(unsigned short, b);
                r -= (float)b;
            }
        }
    }
}

void split_weights(const float* w, int64_t n, int np, void* planes, hipStream_t s, bool f16, int* ovf) {
    if (!n) return;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, w, n, np,
                       f16 ? 1 : 0, (unsigned short*)planes, ovf);
    check_launch("split_weights");
}

bool conv2d_split_eligible(const ConvArgs& a) {
    if (a.Cin % SBK != 0 || a.w_split == nullptr) return false;
    const int64_t x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    const int64_t xb = ((int64_t)(a.N - 1) * x_bs + (int64_t)a.H * a.W * a.Cin) * 4;
    const int64_t wb = (int64_t)a.Cout * a.ks * a.ks * a.Cin * 2 * 3;
    return xb < 0xFFFFFF00ll && wb < 0xFFFFFF00ll;
}

// experiment: LDS-DMA weight staging in the dominant kernel (DSD_CONV_DMA=1)
static bool conv_dma_enabled() {
    static const bool on = getenv("DSD_CONV_DMA") != nullptr;
    return on;
}

// tap reuse (TR instantiation): 3x3, stride 1, "same" padding (the folded nearest-x2 upsample included), unsplit K; the 256-row tile is a whole
// number of image rows of one sample (W = 32 .. 256 a power of two, H * W a multiple of 256)
static bool tr_enabled() {
    static const int mode = getenv("DSD_CONV_TR") ? atoi(getenv("DSD_CONV_TR")) : 1;   // DSD_CONV_TR=0: the plain A-direct kernel (A/B)
    return mode > 0;
}
// the tap-reuse kernel with its matrix work issued as v_mfma_f32_16x16x32_bf16 (conv_tr16.hip): DSD_CONV_MFMA16=1 / 0
void launch_split_tr16(const SplitP& p, dim3 grid, hipStream_t s);
static int g_mfma16 = -1;
static bool mfma16_enabled() {
    if (g_mfma16 < 0) g_mfma16 = getenv("DSD_CONV_MFMA16") ? atoi(getenv("DSD_CONV_MFMA16")) : 0;
    return g_mfma16 > 0;
}
void conv2d_set_mfma16(int on) { g_mfma16 = on ? 1 : 0; }
int conv2d_get_mfma16() { return mfma16_enabled() ? 1 : 0; }
static bool tr_shape_ok(int ks, int stride, int pad, int OW, int IWg, int OH, int IHg, int ksplit, int out_nchw, int ohw, int64_t M, int Cin) {
    return ks == 3 && stride == 1 && pad == 1 && OW == IWg && OH == IHg && ksplit == 1 && !out_nchw &&
           (OW == 32 || OW == 64 || OW == 128 || OW == 256) && ohw % (2 * SBM) == 0 && M % (2 * SBM) == 0 && Cin % SBK == 0;
}
// tile widths the tap-reuse kernel is instantiated for: 160 columns (the 320 / 640 / 1280-channel networks) and, round 3, 128
// columns (128 / 256 / 512 channels: the KL-VAE, narrower U-Nets); DSD_CONV_TR_NT4=0 switches the latter off for A/B runs
static bool tr_nt_ok(int nt) {
    static const bool nt4 = !(getenv("DSD_CONV_TR_NT4") && atoi(getenv("DSD_CONV_TR_NT4")) == 0);
    return nt == 5 || (nt == 4 && nt4);
}
static bool conv_tr_ok(const SplitP& p) {
    return tr_enabled() && !p.stamps && tr_shape_ok(p.ks, p.stride, p.pad, p.OW, p.IWg, p.OH, p.IHg, p.ksplit, p.out_nchw, p.ohw, p.M, p.Cin);
}
// would conv2d_split(a, nt, ksplit, ad) run the tap-reuse instantiation?  (conv2d_variant: its launches are a kind of their own)
bool conv2d_split_tr(const ConvArgs& a, int nt, int ksplit, int ad) {
    if (!(tr_enabled() && !a.stamps && a.precision == PREC_BF16X6 && ad == 2 && tr_nt_ok(nt))) return false;
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int IHg = a.ups ? a.H * 2 : a.H, IWg = a.ups ? a.W * 2 : a.W;
    return tr_shape_ok(a.ks, a.stride, a.pad_lo >= 0 ? a.pad_lo : a.ks / 2, OW, IWg, OH, IHg, ksplit, a.out_nchw, OH * OW, (int64_t)a.N * OH * OW, a.Cin);
}

template <int NP, bool F16>
static void launch_split(const SplitP& p, int nt, hipStream_t s, int ad) {   // ad: 0 = staged, 1 / 2 = A-direct with RB row blocks
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n * (ad >= 1 ? p.ksplit : 1)));
    if (p.stamps) {
        if (!(ad == 2 && nt == 5 && NP == 3 && !F16 && p.ksplit == 1)) fail("conv stamps: only the dominant kernel (256-row tile, 160 columns, bf16x6) has the diagnostic build");
        launch_split_diag(p, grid, s);
        check_launch("conv_split_ad2_stamped");
        return;
    }
    if (ad == 2 && nt == 5 && NP == 3 && !F16 && conv_tr_ok(p) && mfma16_enabled()) {   // the same kernel on v_mfma_f32_16x16x32_bf16 (conv_tr16.hip)
        launch_split_tr16(p, grid, s);
        check_launch("conv_split_tr16");
        return;
    }
    if (ad == 2 && nt == 5 && NP == 3 && !F16 && conv_tr_ok(p)) {
        if (p.gn_scale)
            hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 0, true, true>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 0, true>), grid, dim3(256), 0, s, p);
        check_launch("conv_split_ad2_tr");
        return;
    }
    if (ad == 2 && nt == 4 && tr_nt_ok(4) && NP == 3 && !F16 && conv_tr_ok(p)) {
        if (p.gn_scale)
            hipLaunchKernelGGL((conv_split_ad_kernel<4, 3, false, 2, false, 0, true, true>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((conv_split_ad_kernel<4, 3, false, 2, false, 0, true>), grid, dim3(256), 0, s, p);
        check_launch("conv_split_ad2_tr4");
        return;
    }
    if (p.gn_scale) fail("conv2d: GroupNorm coefficients given for a problem the tap-reuse kernel does not take (ask conv2d_fuses_gn first)");
    if (ad == 2 && conv_dma_enabled() && nt == 5 && NP == 3 && !F16) {
        hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, true>), grid, dim3(256), 0, s, p);
        check_launch("conv_split_ad2_dma");
        return;
    }
    if (ad == 2) {
        switch (nt) {
            case 1: hipLaunchKernelGGL((conv_split_ad_kernel<1, NP, F16, 2>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((conv_split_ad_kernel<2, NP, F16, 2>), grid, dim3(256), 0, s, p); break;
            case 3: hipLaunchKernelGGL((conv_split_ad_kernel<3, NP, F16, 2>), grid, dim3(256), 0, s, p); break;
            case 4: hipLaunchKernelGGL((conv_split_ad_kernel<4, NP, F16, 2>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((conv_split_ad_kernel<5, NP, F16, 2>), grid, dim3(256), 0, s, p); break;
        }
        check_launch("conv_split_ad2");
        return;
    }
    if (ad) {
        switch (nt) {
            case 1: hipLaunchKernelGGL((conv_split_ad_kernel<1, NP, F16, 1>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((conv_split_ad_kernel<2, NP, F16, 1>), grid, dim3(256), 0, s, p); break;
            case 3: hipLaunchKernelGGL((conv_split_ad_kernel<3, NP, F16, 1>), grid, dim3(256), 0, s, p); break;
            case 4: hipLaunchKernelGGL((conv_split_ad_kernel<4, NP, F16, 1>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((conv_split_ad_kernel<5, NP, F16, 1>), grid, dim3(256), 0, s, p); break;
        }
        check_launch("conv_split_ad");
        return;
    }
    switch (nt) {
        case 1: hipLaunchKernelGGL((conv_split_kernel<1, NP, F16>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_split_kernel<2, NP, F16>), grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((conv_split_kernel<3, NP, F16>), grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_split_kernel<4, NP, F16>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv_split_kernel<5, NP, F16>), grid, dim3(256), 0, s, p); break;
    }
    check_launch("conv_split");
}

// structure of the split-precision convolution for a problem (0 staged, 1 / 2 A-direct with 128 / 256-row tiles)
static int split_structure(const ConvArgs& a, int M) {
    if (a.variant == 30) return 1;
    if (a.variant == 31) return 0;
    if (a.variant == 32) return 2;
    if (a.precision == PREC_BF16X6) return M >= 4096 ? 2 : 1;
    if (a.precision == PREC_F16X3) return M >= 16384 ? 2 : 0;
    return M >= 16384 ? 2 : (M >= 4096 ? 0 : 1);
}

// Split-K: a layer whose output has fewer tiles than the chip has CUs (8x8 and 16x16 maps, everything at small batch)
// runs its long k-loop (up to 540 tiles) serially in a handful of workgroups, and the generic cost model then narrows the
// N tile to get more workgroups, which re-reads and re-splits the activations once per N tile.  The A-direct kernels
// instead divide the k-tiles over `ksplit` workgroups per output tile (a second kernel adds the partial tiles in a fixed
// order), and tile width and split factor are chosen together from a small cycle model:
//   workgroup = O + nt*E + ceil(KT/ksplit) * (A + nt*B)      time = max(1, workgroups / 256) * workgroup (+ reduction)
// (in bf16x6 the 128-row / 256-row structure is part of the choice).
// Measured (tools/bench_conv.py): batch 1, 8x8 960->960: 0.096 -> 0.027 ms, whole step 52.0 -> 40 ms; batch 16, 8x8
// 960->960: 0.161 (nt 1, x2) -> 0.109 ms (nt 3, x6).  Grids that fill the chip anyway keep nt_default and ksplit = 1.
void conv2d_split_plan(const ConvArgs& a, int nt_default, int* nt_out, int* ks_out, int* ad_out, bool allow_split) {
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int64_t M = (int64_t)a.N * OH * OW;
    int ad0 = split_structure(a, (int)std::min<int64_t>(M, 1 << 30));
    *nt_out = nt_default;
    *ks_out = 1;
    if (ad_out) *ad_out = ad0;
    static const bool off = getenv("DSD_NO_SPLITK") != nullptr;      // experiments only
    static const bool forced_nt = getenv("DSD_FORCE_NT") != nullptr;
    if (off || !allow_split || M >= (1 << 20) || a.Cout % 4 != 0 || ad0 == 0) return;
    const int t32 = cdiv(a.Cout, 32);
    const int KT = a.ks * a.ks * (a.Cin / SBK);
    // only grids that leave CUs idle are re-planned; everything else keeps the measured structure and nt_default
    const int lanes = std::max(1, a.lanes);   // launches of this shape running side by side: each gets 1 / lanes of the chip
    if ((int64_t)cdiv((int)M, ad0 * SBM) * cdiv(t32, nt_default) * lanes > (ad0 == 1 ? 320 : 160) || KT < 32) return;
    const bool free_structure = a.precision == PREC_BF16X6 && !(a.variant >= 30 && a.variant <= 32);
    double best_t = 1e300;
    for (int ad = 1; ad <= 2; ++ad) {
        if (!free_structure && ad != ad0) continue;
        const int tiles_m = cdiv((int)M, ad * SBM);
        const double mf = (a.precision == PREC_BF16X6 ? 384.0 : 192.0) * ad;   // MFMA cycles per k-tile per 32 columns
        const double fix = ad == 1 ? 700.0 : 1000.0;                            // per k-tile: activation loads, split, barrier
        for (int nt = forced_nt ? nt_default : 5; nt >= (forced_nt ? nt_default : 1); --nt) {
            const int blocks = tiles_m * cdiv(t32, nt);
            for (int ks = 1; ks <= 16; ++ks) {
                if (ks > 1 && KT / ks < 8) break;
                if (ks > 1 && (int64_t)blocks * ks > 768) break;
                const double wg = 3000.0 + nt * 600.0 + cdiv(KT, ks) * (fix + nt * mf);
                // a CU holds one 256-row workgroup or two 128-row ones.  Two residents of a small-M layer (weight streaming,
                // latency-bound) overlap well; on mid-size M they compete for the matrix pipes and the second one buys
                // nothing (measured: 8x8 960->960 x16: 480 workgroups 0.098 ms vs 240 0.125; 64x64 640->640 x1: 512
                // workgroups 0.218 ms vs 256 0.154)
                const double over = std::max(1.0, (double)blocks * ks * lanes / 256.0);
                double t = (ad == 1 && M <= 2048 ? std::pow(over, 0.6) : over) * wg;
                if (ks > 1) t += 9000.0 + (double)ks * M * a.Cout * 4.0 / 2.5e12 * 1.8e9;   // the reduction kernel
                if (t < best_t * 0.999) {
                    best_t = t;
                    *nt_out = nt;
                    *ks_out = ks;
                    if (ad_out) *ad_out = ad;
                }
            }
        }
    }
}

void conv2d_split(const ConvArgs& a, int nt, int ksplit, int ad, hipStream_t s) {
    SplitP p{};
    p.x = a.x; p.w = a.w_split; p.bias = a.bias; p.emb = a.emb; p.res = a.res; p.y = a.y;
    p.N = a.N; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.ks = a.ks; p.stride = a.stride;
    p.pad = a.pad_lo >= 0 ? a.pad_lo : a.ks / 2; p.ups = a.ups; p.emb_stride = a.emb_stride; p.out_nchw = a.out_nchw;
    p.y_ld = a.y_ld > 0 ? a.y_ld : a.Cout;
    p.x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    p.IHg = a.ups ? a.H * 2 : a.H;
    p.IWg = a.ups ? a.W * 2 : a.W;
    conv_out_hw(a, &p.OH, &p.OW);
    p.ohw = p.OH * p.OW;
    p.M = (int)((int64_t)a.N * p.ohw);
    p.Ktot = a.ks * a.ks * a.Cin;
    p.cchunks = a.Cin / SBK;
    p.tiles_m = cdiv(p.M, SBM);
    p.tiles_n = cdiv(a.Cout, nt * 32);
    const int np = a.precision == PREC_BF16X6 ? 3 : 2;
    p.x_bytes = (unsigned)(((int64_t)(a.N - 1) * p.x_bs + (int64_t)a.H * a.W * a.Cin) * 4);
    p.w_plane_bytes = (unsigned)((int64_t)a.Cout * p.Ktot * 2);
    p.ovf = a.ovf;
    p.w_bytes = p.w_plane_bytes * 3u;   // planes are always stored 3 deep; bf16x3 reads the first two
    // Structure by measurement (tools/bench_conv.py 21 31 41 20 30 40 12 42, batch-16 layer shapes, TF/s of fp32-equivalent
    // work, staged / A-direct 128-row / A-direct 256-row):
    //   bf16x6  M >= 16384: 196-202 / 217-226 / 224-237    M = 4096: 131-156 / 160-175 / 164-182    M = 1024: 65 / 90 / 63
    //   bf16x3  M >= 16384: 322-344 / 340-364 / 373-413    M = 4096: 224-259 / 195-239 / 209-255    M = 1024: 81 / 110 / 64
    //   f16x3   M >= 16384: 302-331 /    -    / 331-365    M = 4096: 226-261 /    -    / 189-222
    if (ad == 2) p.tiles_m = cdiv(p.M, 2 * SBM);   // 256-row block tile, one workgroup per CU
    p.ksplit = 1;
    if (ad >= 1 && ksplit > 1) {
        DSD_CHECK(a.scratch && (size_t)ksplit * p.M * a.Cout * sizeof(float) <= a.scratch_bytes,
                  "conv2d: split-K x%d needs %zu bytes of scratch (conv2d_scratch_bytes), got %zu", ksplit,
                  (size_t)ksplit * p.M * a.Cout * sizeof(float), a.scratch_bytes);
        p.ksplit = ksplit;
        p.partial = a.scratch;
    }
    p.stats = nullptr;
    p.stats_chunks = 0;
    p.stamps = a.stamps;
    p.diag = a.diag;
    p.gn_scale = a.gn_scale;
    p.gn_shift = a.gn_shift;
    DSD_CHECK((a.gn_scale == nullptr) == (a.gn_shift == nullptr), "conv2d: gn_scale and gn_shift come together");
    if (a.stats) {
        const int rows = ad == 2 ? 2 * SBM : SBM;
        DSD_CHECK(p.ksplit == 1 && !a.out_nchw && p.ohw % rows == 0 && p.ohw / rows == a.stats_chunks && !(ad == 1 && nt >= 4),
                  "conv2d: output statistics requested with %d chunks but the kernel (tile %d rows, split-K x%d, ohw %d) cannot "
                  "emit them (conv2d_stats_chunks)", a.stats_chunks, rows, p.ksplit, p.ohw);
        p.stats = a.stats;
        p.stats_chunks = a.stats_chunks;
    }
    if (a.precision == PREC_F16X3)
        launch_split<2, true>(p, nt, s, ad);
    else if (np == 2)
        launch_split<2, false>(p, nt, s, ad);
    else
        launch_split<3, false>(p, nt, s, ad);
    if (p.ksplit > 1) {
        const int64_t total4 = (int64_t)p.M * a.Cout / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>((total4 + 255) / 256, 2048)), dim3(256), 0, s,
                           p.partial, p.ksplit, p.M, a.Cout, a.bias, a.emb, a.emb_stride, p.ohw, a.res, a.y, p.y_ld, a.out_nchw);
        check_launch("splitk_reduce");
    }
}

}  // namespace dsd
