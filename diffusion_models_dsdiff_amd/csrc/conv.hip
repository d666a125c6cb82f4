// conv.hip — NHWC fp32 convolution as an implicit GEMM on the gfx950 f32 matrix cores.
//
// Replaces every nn.Conv2d / nn.Conv1d(k=1) call site on the hot path
// (ldm/modules/diffusionmodules/util.py:229-239 conv_nd; openaimodel.py:109,155,208,234,245,452,460;
//  UNet_DS_Diff/model.py:158,163,285,514,569-601).
//
//   M = N*OH*OW (output pixels), Ngemm = Cout, K = ks*ks*Cin, A gathered on the fly (im2col never
//   materialised), B = weights packed OHWI so both operands are K-contiguous.
//   Block = 256 threads = 4 waves; block tile 128 (M) x 32*NT (N) x 32 (K); each wave owns 32 rows and
//   all NT 32-wide column tiles -> NT accumulators of v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
//   LDS tiles are [row][36] floats (pad 4): the ds_read_b128 fragment reads (lane = row, 4 k's) and the
//   ds_write_b128 staging writes are both bank-conflict free with that stride.
//   K order inside a group of 8 is permuted (lane half h takes k = 8g+4h+j for MFMA j) identically for A
//   and B, so one 16-byte LDS read feeds four MFMAs.
//   Global -> register prefetch of tile kt+1 is issued before the MFMAs of tile kt (single LDS buffer,
//   2 blocks per CU hide the barriers).  Epilogue fuses bias, the per-(sample,channel) timestep-embedding
//   add and the residual add.
//   Two kernels share the tile code: conv_mfma_buf_kernel (default; buffer loads, hardware range check for
//   padding) and conv_mfma_kernel (flat loads + masks; any Cin % 4 == 0, operands >= 4 GiB).
//   Measured on the way here (tools/bench_conv.py, 16x256x256x320->320): flat loads 125 TF/s; + fragment
//   reads issued one group ahead and a branch-free epilogue 129; MFMA + LDS reads alone 151, + barriers and
//   LDS writes 147, i.e. the flat kernel lost 12 % to address arithmetic / branches in the load phase;
//   buffer loads 143 (91 % of the 157.3 TF/s fp32 matrix peak).  A double-buffered BK=16 / one-barrier
//   variant and an accumulator-interleaved MFMA order were tried and are not faster (142 / 127).
#include "kernels.h"

#include <string>

#include <cstdlib>

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr size_t DIRECT_LDS_MAX = 128 * 1024;   // weights of the small-K direct kernel, [K][Cout] fp32 in LDS
static constexpr int BM = 128;
static constexpr int BK = 32;
static constexpr int LDS_STRIDE = 36;

struct ConvP {
    const float* x;
    const float* w;
    const float* bias;
    const float* emb;
    const float* res;
    float* y;
    int64_t x_bs;
    int N, H, W, Cin, Cout, OH, OW, ks, stride, pad, ups, emb_stride, out_nchw;
    int M, Ktot, cchunks, IHg, IWg, tiles_m, tiles_n, ohw;
    unsigned x_bytes, w_bytes;  // buffer-descriptor extents (buffer-load kernel only)
    int y_ld;                   // row stride of y (>= Cout)
};


// ---- accumulator tile -> memory.  C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Interior tiles (the common case) take a branch-free path: bias once per column, then per group of 4 rows all
// residual / embedding loads are issued together before the adds and stores (one vmcnt wait per group instead of
// one dependent global load per element).
template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, const f32x16 (&acc)[NT], int m0, int n0, int wave, int lrow,
                                              int half) {
    const bool interior = (m0 + BM <= p.M) && (n0 + NT * 32 <= p.Cout) && !p.out_nchw;
    if (interior) {
        float bj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bj[j] = p.bias ? p.bias[n0 + j * 32 + lrow] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int mb = m0 + wave * 32 + 8 * g + 4 * half;  // rows mb .. mb+3  <->  regs 4g .. 4g+3
            float ev[4][NT], rv[4][NT];
            if (p.emb) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float* er = p.emb + (int64_t)((mb + rr) / p.ohw) * p.emb_stride + n0 + lrow;
#pragma unroll
                    for (int j = 0; j < NT; ++j) ev[rr][j] = er[j * 32];
                }
            }
            if (p.res) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float* rp = p.res + (int64_t)(mb + rr) * p.Cout + n0 + lrow;
#pragma unroll
                    for (int j = 0; j < NT; ++j) rv[rr][j] = rp[j * 32];
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float* yp = p.y + (int64_t)(mb + rr) * p.y_ld + n0 + lrow;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float v = acc[j][4 * g + rr] + bj[j];
                    if (p.emb) v += ev[rr][j];
                    if (p.res) v += rv[rr][j];
                    yp[j * 32] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.M) continue;
        const int nb = m / p.ohw;
        const float* embrow = p.emb ? p.emb + (int64_t)nb * p.emb_stride : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 32 + lrow;
            if (n >= p.Cout) continue;
            float v = acc[j][r];
            if (p.bias) v += p.bias[n];
            if (embrow) v += embrow[n];
            if (p.res) v += p.res[(int64_t)m * p.Cout + n];
            if (p.out_nchw) {
                const int pix = m - nb * p.ohw;
                p.y[((int64_t)nb * p.Cout + n) * p.ohw + pix] = v;
            } else {
                p.y[(int64_t)m * p.y_ld + n] = v;
            }
        }
    }
}

// ---- MFMAs of NKG k-groups (8 k each) of one staged tile, fragment reads software-pipelined ONE group ahead:
// the ds_read_b128 of group g+1 is issued before the four MFMAs of group g, so its latency hides behind 256 cycles
// of matrix work instead of being waited for in front of every group.
template <int NT, int NKG, int LSTRIDE>
__device__ __forceinline__ void conv_tile_mfma(const float* a_frag, const float* b_frag, f32x16 (&acc)[NT]) {
    float4 a_cur = *reinterpret_cast<const float4*>(a_frag);
    float4 b_cur = *reinterpret_cast<const float4*>(b_frag);
    float4 a_nxt = a_cur, b_nxt = b_cur;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j + 1 < NT) {
                b_nxt = *reinterpret_cast<const float4*>(b_frag + (j + 1) * 32 * LSTRIDE + 8 * kg);
            } else if (kg + 1 < NKG) {
                b_nxt = *reinterpret_cast<const float4*>(b_frag + 8 * (kg + 1));
                a_nxt = *reinterpret_cast<const float4*>(a_frag + 8 * (kg + 1));
            }
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc[j], 0, 0, 0);
            b_cur = b_nxt;
            if (j == NT - 1) a_cur = a_nxt;
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvP p) {
    __shared__ __attribute__((aligned(16))) float lds[(BM + NT * 32) * LDS_STRIDE];
    float* As = lds;
    float* Bs = lds + BM * LDS_STRIDE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // XCD-aware tile order: blocks that share A rows / the weight panel run on one XCD (one L2).
    const int nwg = gridDim.x;
    int L = blockIdx.x;
    {
        const int cpx = nwg >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const int tile_n = L % p.tiles_n;
    const int tile_m = L / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * (NT * 32);

    // ---- staging map: thread -> (row = tid>>3 (+32 i), 4 consecutive k = 4*(tid&7))
    const int col4 = tid & 7;
    const int srow = tid >> 3;
    int a_h[4], a_w[4];
    int64_t a_nb[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        m = a_ok[i] ? m : 0;
        const int n = m / p.ohw;
        const int r = m - n * p.ohw;
        const int oh = r / p.OW;
        const int ow = r - oh * p.OW;
        a_h[i] = oh * p.stride - p.pad;
        a_w[i] = ow * p.stride - p.pad;
        a_nb[i] = (int64_t)n * p.x_bs;
    }
    const float* wrow[NT];
    bool b_ok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + srow + 32 * j;
        b_ok[j] = n < p.Cout;
        wrow[j] = p.w + (int64_t)(b_ok[j] ? n : 0) * p.Ktot;
    }

    float4 ra[4], rb[NT];
    auto load_tile = [&](int kh, int kw, int cc) {
        const int c = cc * BK + col4 * 4;
        const bool cok = c < p.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ih = a_h[i] + kh, iw = a_w[i] + kw;
            const bool ok = a_ok[i] && cok && (unsigned)ih < (unsigned)p.IHg && (unsigned)iw < (unsigned)p.IWg;
            if (p.ups) {
                ih >>= 1;
                iw >>= 1;
            }
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) ra[i] = *reinterpret_cast<const float4*>(p.x + a_nb[i] + ((int64_t)ih * p.W + iw) * p.Cin + c);
        }
        const int kofs = (kh * p.ks + kw) * p.Cin + c;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_ok[j] && cok) rb[j] = *reinterpret_cast<const float4*>(wrow[j] + kofs);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int lrow = lane & 31;
    const int half = lane >> 5;
    const float* a_frag = As + (wave * 32 + lrow) * LDS_STRIDE + 4 * half;
    const float* b_frag = Bs + lrow * LDS_STRIDE + 4 * half;
    float* a_st = As + srow * LDS_STRIDE + col4 * 4;
    float* b_st = Bs + srow * LDS_STRIDE + col4 * 4;

    const int KT = p.ks * p.ks * p.cchunks;
    int kh = 0, kw = 0, cc = 0;
    load_tile(kh, kw, cc);
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();  // all waves finished reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(a_st + 32 * i * LDS_STRIDE) = ra[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(b_st + 32 * j * LDS_STRIDE) = rb[j];
        __syncthreads();
        if (kt + 1 < KT) {  // prefetch next tile into registers; lands while the MFMAs below run
            if (++cc == p.cchunks) {
                cc = 0;
                if (++kw == p.ks) {
                    kw = 0;
                    ++kh;
                }
            }
            load_tile(kh, kw, cc);
        }
        conv_tile_mfma<NT, 4, LDS_STRIDE>(a_frag, b_frag, acc);
    }

    conv_epilogue<NT>(p, acc, m0, n0, wave, lrow, half);
}

// Default kernel: same tiling/loop as conv_mfma_kernel, but the operand gathers use BUFFER loads: a 32-bit per-lane
// byte offset that is recomputed only when the filter tap changes (every Cin/32 tiles), a scalar per-tile offset, and
// the hardware range check instead of exec-masked branches for padding / ragged rows (an out-of-range offset returns
// zeros).  The flat-load kernel spends ~12 % of its time issuing address arithmetic and branches (tools/bench_conv.py
// diagnostics); this one issues 9 loads and two scalar adds per tile.  Needs Cin % 32 == 0 and < 4 GiB operands.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned OOB = 0xFFFFFFF0u;

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_mfma_buf_kernel(ConvP p) {
    __shared__ __attribute__((aligned(16))) float lds[(BM + NT * 32) * LDS_STRIDE];
    float* As = lds;
    float* Bs = lds + BM * LDS_STRIDE;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwg = gridDim.x;
    int L = blockIdx.x;
    {
        const int cpx = nwg >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const int tile_n = L % p.tiles_n;
    const int tile_m = L / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * (NT * 32);

    const int col4 = tid & 7;
    const int srow = tid >> 3;
    int a_h[4], a_w[4];
    unsigned a_nb[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        m = a_ok[i] ? m : 0;
        const int n = m / p.ohw;
        const int r = m - n * p.ohw;
        const int oh = r / p.OW;
        const int ow = r - oh * p.OW;
        a_h[i] = oh * p.stride - p.pad;
        a_w[i] = ow * p.stride - p.pad;
        a_nb[i] = (unsigned)n * (unsigned)p.x_bs + (unsigned)(col4 * 4);
    }
    unsigned b_voff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + srow + 32 * j;
        b_voff[j] = n < p.Cout ? ((unsigned)n * (unsigned)p.Ktot + (unsigned)(col4 * 4)) * 4u : OOB;
    }
    unsigned a_voff[4];
    auto tap_offsets = [&](int kh, int kw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ih = a_h[i] + kh, iw = a_w[i] + kw;
            const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.IHg && (unsigned)iw < (unsigned)p.IWg;
            if (p.ups) {
                ih >>= 1;
                iw >>= 1;
            }
            a_voff[i] = ok ? (a_nb[i] + (unsigned)(ih * p.W + iw) * (unsigned)p.Cin) * 4u : OOB;
        }
    };
    f32x4 ra[4], rb[NT];
    auto load_tile = [&](int soff_a, int soff_b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[i], soff_a, 0));
#pragma unroll
        for (int j = 0; j < NT; ++j) rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[j], soff_b, 0));
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int lrow = lane & 31;
    const int half = lane >> 5;
    const float* a_frag = As + (wave * 32 + lrow) * LDS_STRIDE + 4 * half;
    const float* b_frag = Bs + lrow * LDS_STRIDE + 4 * half;
    float* a_st = As + srow * LDS_STRIDE + col4 * 4;
    float* b_st = Bs + srow * LDS_STRIDE + col4 * 4;

    const int KT = p.ks * p.ks * p.cchunks;
    int kh = 0, kw = 0, cc = 0, tap = 0;
    tap_offsets(0, 0);
    load_tile(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(a_st + 32 * i * LDS_STRIDE) = ra[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(b_st + 32 * j * LDS_STRIDE) = rb[j];
        __syncthreads();
        if (kt + 1 < KT) {
            if (++cc == p.cchunks) {
                cc = 0;
                ++tap;
                if (++kw == p.ks) {
                    kw = 0;
                    ++kh;
                }
                tap_offsets(kh, kw);
            }
            load_tile(cc * (BK * 4), (tap * p.Cin + cc * BK) * 4);
        }
        conv_tile_mfma<NT, 4, LDS_STRIDE>(a_frag, b_frag, acc);
    }
    conv_epilogue<NT>(p, acc, m0, n0, wave, lrow, half);
}

// Direct kernel for tiny K (first layer: Cin = 1, K = 9): output-write bound, weights staged in LDS as [k][Cout].
__global__ __launch_bounds__(256) void conv_direct_lds_kernel(ConvP p) {
    extern __shared__ float wl[];  // [Ktot][Cout]
    const int tid = threadIdx.x;
    for (int i = tid; i < p.Ktot * p.Cout; i += 256) {
        const int k = i / p.Cout, co = i - k * p.Cout;
        wl[i] = p.w[(int64_t)co * p.Ktot + k];
    }
    __syncthreads();
    const int c4n = p.Cout >> 2;
    const int PIX = 64;
    const int64_t pix0 = (int64_t)blockIdx.x * PIX;
    for (int idx = tid; idx < PIX * c4n; idx += 256) {
        const int pl = idx / c4n, c4 = idx - pl * c4n;
        const int64_t m = pix0 + pl;
        if (m >= p.M) break;
        const int nb = (int)(m / p.ohw);
        const int r = (int)(m - (int64_t)nb * p.ohw);
        const int oh = r / p.OW, ow = r - oh * p.OW;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (int kh = 0; kh < p.ks; ++kh)
            for (int kw = 0; kw < p.ks; ++kw) {
                int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                const bool ok = (unsigned)ih < (unsigned)p.IHg && (unsigned)iw < (unsigned)p.IWg;
                if (p.ups) {
                    ih >>= 1;
                    iw >>= 1;
                }
                const float* src = p.x + (int64_t)nb * p.x_bs + ((int64_t)ih * p.W + iw) * p.Cin;
                for (int ci = 0; ci < p.Cin; ++ci, ++k) {
                    const float xv = ok ? src[ci] : 0.f;
                    const float4 w4 = *reinterpret_cast<const float4*>(wl + k * p.Cout + c4 * 4);
                    acc.x = fmaf(xv, w4.x, acc.x);
                    acc.y = fmaf(xv, w4.y, acc.y);
                    acc.z = fmaf(xv, w4.z, acc.z);
                    acc.w = fmaf(xv, w4.w, acc.w);
                }
            }
        float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = c4 * 4 + q;
            if (p.bias) v[q] += p.bias[n];
            if (p.emb) v[q] += p.emb[(int64_t)nb * p.emb_stride + n];
            if (p.res) v[q] += p.res[m * p.Cout + n];
        }
        if (p.out_nchw) {
            for (int q = 0; q < 4; ++q) p.y[((int64_t)nb * p.Cout + c4 * 4 + q) * p.ohw + r] = v[q];
        } else {
            *reinterpret_cast<float4*>(p.y + m * p.y_ld + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// First layer (Cin = 1, K = 9 -> 320 channels at 256x256): pure output-write traffic.  A thread owns four fixed output
// channels, keeps their K x 4 weights and bias in registers and walks over pixels: per pixel K broadcast loads, 4K FMAs and
// one 16-byte store, no LDS and no per-element index divisions (the LDS version above reaches 1.9 TB/s, this one the
// store rate of the elementwise kernels).
// STATS [r3]: the workgroup's pixels are one chunk of ONE sample (ppb divides OH*OW) and it also leaves that chunk's per-channel
// (sum, sum of squares) in stats[n][chunk][co][2] (fp64, the GnSrc layout gn_finalize reads): the GroupNorm that follows the
// first convolution of every encoder stream no longer re-reads the 1.3 GB it just wrote.
template <int KMAX, bool STATS>
__global__ __launch_bounds__(256) void conv_direct_cols_kernel(ConvP p, int W4, int ppb, double* __restrict__ stats, int chunks) {
    __shared__ double sm[STATS ? 1024 : 1][2];   // [row][co] partials of rows 1.. (rpi * Cout <= 1024)
    const int rpi = blockDim.x / W4;
    const int row = threadIdx.x / W4;
    const int c4 = threadIdx.x - row * W4;
    double ss[4] = {0.0, 0.0, 0.0, 0.0}, qq[4] = {0.0, 0.0, 0.0, 0.0};
    float4 w[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < p.Ktot) {
            w[k].x = p.w[(int64_t)(c4 * 4 + 0) * p.Ktot + k];
            w[k].y = p.w[(int64_t)(c4 * 4 + 1) * p.Ktot + k];
            w[k].z = p.w[(int64_t)(c4 * 4 + 2) * p.Ktot + k];
            w[k].w = p.w[(int64_t)(c4 * 4 + 3) * p.Ktot + k];
        } else {
            w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    int dh[KMAX], dw[KMAX], dc[KMAX];   // filter-tap offsets of k, once per thread (no divisions in the pixel loop)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int tap = k / p.Cin;
        dc[k] = k - tap * p.Cin;
        dh[k] = tap / p.ks - p.pad;
        dw[k] = tap - (tap / p.ks) * p.ks - p.pad;
    }
    const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t m_end = min((int64_t)p.M, ((int64_t)blockIdx.x + 1) * ppb);
    int64_t m = (int64_t)blockIdx.x * ppb + row;
    int nb = (int)(m / p.ohw);
    int oh = (int)(m - (int64_t)nb * p.ohw) / p.OW;
    int ow = (int)(m - (int64_t)nb * p.ohw) - oh * p.OW;
    for (; m < m_end; m += rpi, ow += rpi) {
        while (ow >= p.OW) {   // pixel coordinates advance incrementally: no division per pixel
            ow -= p.OW;
            if (++oh == p.OH) {
                oh = 0;
                ++nb;
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xb = p.x + (int64_t)nb * p.x_bs;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            int ih = oh * p.stride + dh[k], iw = ow * p.stride + dw[k];
            const bool ok = (k < p.Ktot) & ((unsigned)ih < (unsigned)p.IHg) & ((unsigned)iw < (unsigned)p.IWg);
            ih >>= p.ups;
            iw >>= p.ups;
            const float xv = ok ? xb[((int64_t)ih * p.W + iw) * p.Cin + dc[k]] : 0.f;
            acc.x = fmaf(xv, w[k].x, acc.x);
            acc.y = fmaf(xv, w[k].y, acc.y);
            acc.z = fmaf(xv, w[k].z, acc.z);
            acc.w = fmaf(xv, w[k].w, acc.w);
        }
        acc.x += b4.x; acc.y += b4.y; acc.z += b4.z; acc.w += b4.w;   // bias after the fma chain, like every other epilogue
        if (p.emb) {
            const float4 e = *reinterpret_cast<const float4*>(p.emb + (int64_t)nb * p.emb_stride + c4 * 4);
            acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
        }
        if (p.res) {
            const float4 e = *reinterpret_cast<const float4*>(p.res + m * p.Cout + c4 * 4);
            acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
        }
        *reinterpret_cast<float4*>(p.y + m * p.y_ld + c4 * 4) = acc;
        if (STATS) {
            const double a = acc.x, b = acc.y, c = acc.z, d = acc.w;
            ss[0] += a; qq[0] = fma(a, a, qq[0]);
            ss[1] += b; qq[1] = fma(b, b, qq[1]);
            ss[2] += c; qq[2] = fma(c, c, qq[2]);
            ss[3] += d; qq[3] = fma(d, d, qq[3]);
        }
    }
    if (STATS) {   // rows 1.. hand their sums to row 0, which adds them in row order (deterministic)
        if (row > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sm[row * p.Cout + c4 * 4 + e][0] = ss[e];
                sm[row * p.Cout + c4 * 4 + e][1] = qq[e];
            }
        }
        __syncthreads();
        if (row == 0) {
            for (int r = 1; r < rpi; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ss[e] += sm[r * p.Cout + c4 * 4 + e][0];
                    qq[e] += sm[r * p.Cout + c4 * 4 + e][1];
                }
            const int64_t m0 = (int64_t)blockIdx.x * ppb;
            const int n0 = (int)(m0 / p.ohw), chunk = (int)((m0 - (int64_t)n0 * p.ohw) / ppb);
            double* o = stats + (((int64_t)n0 * chunks + chunk) * p.Cout + c4 * 4) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] = ss[e];
                o[2 * e + 1] = qq[e];
            }
        }
    }
}

// Fully generic scalar fallback (odd channel counts; never on the hot configs).
__global__ __launch_bounds__(256) void conv_scalar_kernel(ConvP p) {
    const int64_t total = (int64_t)p.M * p.Cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / p.Cout;
        const int n = (int)(i - m * p.Cout);
        const int nb = (int)(m / p.ohw);
        const int r = (int)(m - (int64_t)nb * p.ohw);
        const int oh = r / p.OW, ow = r - oh * p.OW;
        float acc = 0.f;
        const float* wr = p.w + (int64_t)n * p.Ktot;
        for (int kh = 0; kh < p.ks; ++kh)
            for (int kw = 0; kw < p.ks; ++kw) {
                int ih = oh * p.stride - p.pad + kh, iw = ow * p.stride - p.pad + kw;
                if (!((unsigned)ih < (unsigned)p.IHg && (unsigned)iw < (unsigned)p.IWg)) continue;
                if (p.ups) {
                    ih >>= 1;
                    iw >>= 1;
                }
                const float* src = p.x + (int64_t)nb * p.x_bs + ((int64_t)ih * p.W + iw) * p.Cin;
                const float* wk = wr + (kh * p.ks + kw) * p.Cin;
                for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(src[ci], wk[ci], acc);
            }
        if (p.bias) acc += p.bias[n];
        if (p.emb) acc += p.emb[(int64_t)nb * p.emb_stride + n];
        if (p.res) acc += p.res[m * p.Cout + n];
        if (p.out_nchw)
            p.y[((int64_t)nb * p.Cout + n) * p.ohw + r] = acc;
        else
            p.y[m * p.y_ld + n] = acc;
    }
}

__global__ void pack_ohwi_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int kk) {
    const int64_t total = (int64_t)Cout * Cin * kk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // dst index: ((co*kk + t)*Cin + ci)
        const int ci = (int)(i % Cin);
        const int64_t r = i / Cin;
        const int t = (int)(r % kk);
        const int co = (int)(r / kk);
        dst[i] = src[((int64_t)co * Cin + ci) * kk + t];
    }
}

void pack_ohwi(const float* w_oihw, float* w_ohwi, int Cout, int Cin, int ks, hipStream_t s) {
    const int64_t total = (int64_t)Cout * Cin * ks * ks;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(pack_ohwi_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, w_ohwi, Cout, Cin, ks * ks);
    check_launch("pack_ohwi");
}

double conv2d_flops(const ConvArgs& a) {
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    return 2.0 * a.N * OH * OW * (double)a.Cout * a.ks * a.ks * a.Cin;
}

// kernel variant: 0 = default (buffer-load kernel when eligible), 1 = force the flat-load kernel.
// ConvArgs.variant >= 0 wins, else env DSD_CONV_VARIANT, else 0.
static int conv_variant(const ConvArgs& a) {
    if (a.variant >= 0) return a.variant;
    static const int env = [] {
        const char* e = getenv("DSD_CONV_VARIANT");
        return e ? atoi(e) : 0;
    }();
    return env;
}

// Column-tile count NT (block tile = 128 x 32*NT) by a small cost model fitted to tools/bench_conv.py measurements:
//   time ~ ceil(blocks / (256 CUs x 2 resident)) x K-tiles x (t_fix + NT x t_mfma)
// t_fix = per-tile staging/barrier cycles that do not shrink with NT, t_mfma = matrix-pipe cycles per 32 columns per
// tile (fp32: 16 x 64, bf16x6: 12 x 32, bf16x3: 6 x 32).  Big layers get NT = 5 (weights and A staging amortised over the
// widest tile); small-M layers get narrower tiles so the grid still fills the chip.
static int pick_nt(int Cout, int tiles_m, int precision, int lanes = 1) {
    static const int force = [] {
        const char* e = getenv("DSD_FORCE_NT");   // experiments only
        return e ? atoi(e) : 0;
    }();
    if (force >= 1 && force <= 5) return force;
    const int t32 = cdiv(Cout, 32);
    const double t_fix = precision == PREC_F32 ? 500.0 : 700.0;
    const double t_mf = precision == PREC_F32 ? 1024.0 : (precision == PREC_BF16X6 ? 384.0 : 192.0);   // bf16x3 = f16x3
    int best = 1;
    double best_t = 1e300;
    for (int nt = 5; nt >= 1; --nt) {
        const int64_t blocks = (int64_t)tiles_m * cdiv(t32, nt) * std::max(1, lanes);   // (what shares the chip with this launch)
        const double t = (double)cdiv(blocks, 512) * (t_fix + nt * t_mf);
        if (t < best_t * 0.999) {
            best_t = t;
            best = nt;
        }
    }
    return best;
}

// split-bf16 arithmetic requested and usable for this problem?  (Until the A-direct kernel was software-pipelined, grids
// of at most one block per CU were faster on the fp32 kernel and stayed there; now the 128-row A-direct kernel ties it in
// bf16x6 and beats it in the 3-product modes on those layers — tools/bench_conv.py 0 31 30 with DSD_SHAPES=small.)
static int effective_precision(const ConvArgs& a, int tiles_m) {
    (void)tiles_m;
    if (a.precision == PREC_F32 || !conv2d_split_eligible(a)) return PREC_F32;
    return a.precision;
}

size_t conv2d_scratch_bytes(const ConvArgs& a) {
    if (a.precision == PREC_F32 || a.Cin % 32 != 0 || a.Cin % 4 != 0) return 0;
    if (conv2d_wino_eligible(a)) return 0;
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int64_t M = (int64_t)a.N * OH * OW;
    const int tm = cdiv(M, BM);
    if (effective_precision(a, tm) == PREC_F32) return 0;
    int nt, ks;
    conv2d_split_plan(a, pick_nt(a.Cout, tm, a.precision, a.lanes), &nt, &ks);
    return ks > 1 ? (size_t)ks * M * a.Cout * sizeof(float) : 0;
}

// pixels per workgroup of the first-layer kernel; with_stats: a divisor of OH*OW (one chunk of one sample), 0 = none suitable
static int direct_cols_ppb(const ConvArgs& a, int64_t M, int ohw, bool with_stats) {
    const int W4 = a.Cout / 4, rpi = 256 / W4;
    if (!with_stats) return std::max(rpi * 8, cdiv(cdiv(M, 8192), rpi) * rpi);
    for (int ppb = 128; ppb >= 16; ppb >>= 1)
        if (ohw % ppb == 0) return ppb;
    return 0;
}
static bool direct_cols_ok(const ConvArgs& a) {
    return a.Cout % 4 == 0 && a.Cout / 4 <= 256 && a.ks * a.ks * a.Cin <= 9 && !a.out_nchw && (!a.emb || a.emb_stride % 4 == 0);
}

int conv2d_stats_chunks(const ConvArgs& a) {
    // the split-precision implicit-GEMM kernels carry the statistics epilogue (not their split-K form, whose output is
    // written by the reduction kernel), and so does the first-layer kernel
    if (a.Cin % 4 != 0 || a.ks * a.ks * a.Cin < 32) {
        if (!direct_cols_ok(a)) return 0;
        int OH, OW;
        conv_out_hw(a, &OH, &OW);
        const int ppb = direct_cols_ppb(a, (int64_t)a.N * OH * OW, OH * OW, true);
        return ppb ? OH * OW / ppb : 0;
    }
    if (a.out_nchw) return 0;
    if (conv2d_wino_eligible(a)) return conv2d_wino_stats_chunks(a);
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int64_t M = (int64_t)a.N * OH * OW;
    const int tm = cdiv(M, BM);
    const int pr = effective_precision(a, tm);
    if (pr == PREC_F32) return 0;
    int nt = pick_nt(a.Cout, tm, pr, a.lanes), ks = 1, ad = 0;
    conv2d_split_plan(a, nt, &nt, &ks, &ad);
    if (ks > 1 || (ad == 1 && nt >= 4)) return 0;   // (128-row kernels with >= 4 column tiles have no registers for it)
    const int rows = ad == 2 ? 2 * BM : BM;
    const int ohw = OH * OW;
    return ohw % rows == 0 ? ohw / rows : 0;
}

bool conv2d_fuses_gn(const ConvArgs& a) {
    if (a.Cin % 4 != 0 || a.ks * a.ks * a.Cin < 32 || a.precision != PREC_BF16X6 || a.w_split == nullptr) return false;
    static const bool off = getenv("DSD_NO_GN_FUSE") != nullptr;   // A/B
    if (off || conv2d_wino_eligible(a)) return false;
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int tm = cdiv((int64_t)a.N * OH * OW, BM);
    if (effective_precision(a, tm) != PREC_BF16X6) return false;
    int nt = pick_nt(a.Cout, tm, PREC_BF16X6, a.lanes), ks = 1, ad = 0;
    conv2d_split_plan(a, nt, &nt, &ks, &ad, true);
    return conv2d_split_tr(a, nt, ks, ad);
}

void conv2d_plan_query(const ConvArgs& a, int* structure, int* nt_out, int* ks_out) {
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    const int tm = cdiv((int64_t)a.N * OH * OW, BM);
    const int pr = effective_precision(a, tm);
    int nt = pick_nt(a.Cout, tm, pr, a.lanes), ks = 1, ad = -1;
    if (pr != PREC_F32) conv2d_split_plan(a, nt, &nt, &ks, &ad);
    *structure = ad;
    *nt_out = nt;
    *ks_out = ks;
}

const char* conv2d_variant(const ConvArgs& a) {
    const int Ktot = a.ks * a.ks * a.Cin;
    if (a.Cin % 4 != 0 || Ktot < 32) {
        if (a.Cout % 4 == 0 && a.Cout / 4 <= 256 && Ktot <= 9 && !a.out_nchw && (!a.emb || a.emb_stride % 4 == 0)) return "conv_direct_cols";
        return (a.Cout % 4 == 0 && (size_t)Ktot * a.Cout * 4 <= DIRECT_LDS_MAX) ? "conv_direct_lds" : "conv_scalar";
    }
    int OH, OW;
    conv_out_hw(a, &OH, &OW);
    static const char* names[4][6] = {{"", "conv_mfma<1>", "conv_mfma<2>", "conv_mfma<3>", "conv_mfma<4>", "conv_mfma<5>"},
                                      {"", "conv_bf16x3<1>", "conv_bf16x3<2>", "conv_bf16x3<3>", "conv_bf16x3<4>", "conv_bf16x3<5>"},
                                      {"", "conv_bf16x6<1>", "conv_bf16x6<2>", "conv_bf16x6<3>", "conv_bf16x6<4>", "conv_bf16x6<5>"},
                                      {"", "conv_f16x3<1>", "conv_f16x3<2>", "conv_f16x3<3>", "conv_f16x3<4>", "conv_f16x3<5>"}};
    const int tm = cdiv((int64_t)a.N * OH * OW, BM);
    const int pr = effective_precision(a, tm);
    int nt = pick_nt(a.Cout, tm, pr, a.lanes), ks = 1, ad = 2;
    if (pr == PREC_BF16X6 && conv2d_wino_eligible(a)) return "conv_wino_bf16x6";
    if (pr == PREC_F32) return names[pr][nt];
    conv2d_split_plan(a, nt, &nt, &ks, &ad);
    // the 256-row A-direct kernel keeps the plain name; the other structures and split-K runs are separate kinds, so that a
    // kind's average launch time is one kernel's (bench.py roofline vs the rocprofv3 kernel trace)
    static std::string pool[4][6][3][2][2];
    const bool tr = conv2d_split_tr(a, nt, ks, ad);   // tap reuse through LDS: an instantiation (kernel symbol) of its own
    std::string& n = pool[pr][nt][ad][ks > 1][tr];
    if (n.empty())
        n = std::string(names[pr][nt]) + (ad == 2 ? "" : (ad == 1 ? "/r128" : "/staged")) + (ks > 1 ? "+splitk" : "") + (tr ? "/tr" : "");
    if (tr && a.gn_scale) {   // the instantiation that also applies GroupNorm + SiLU to its input: a kind (kernel symbol) of its own
        static std::string gnpool[6];   // per tile width (the 128- and the 160-column tile are different kernels)
        std::string& gn = gnpool[nt];
        if (gn.empty()) gn = n + "+gn";
        return gn.c_str();
    }
    return n.c_str();
}

void conv2d(ConvArgs a, hipStream_t s) {
    DSD_CHECK(a.ks == 1 || a.ks == 3, "conv2d: kernel size %d unsupported", a.ks);
    DSD_CHECK(a.gn_scale == nullptr || conv2d_fuses_gn(a), "conv2d: GroupNorm coefficients given, but this problem does not run on the kernel that applies them");
    DSD_CHECK(a.stride == 1 || a.stride == 2, "conv2d: stride %d unsupported", a.stride);
    ConvP p{};
    p.x = a.x; p.w = a.w; p.bias = a.bias; p.emb = a.emb; p.res = a.res; p.y = a.y;
    p.N = a.N; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.ks = a.ks; p.stride = a.stride;
    p.y_ld = a.y_ld > 0 ? a.y_ld : a.Cout;
    p.pad = a.pad_lo >= 0 ? a.pad_lo : a.ks / 2; p.ups = a.ups; p.emb_stride = a.emb_stride; p.out_nchw = a.out_nchw;
    p.x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    p.IHg = a.ups ? a.H * 2 : a.H;
    p.IWg = a.ups ? a.W * 2 : a.W;
    conv_out_hw(a, &p.OH, &p.OW);
    p.ohw = p.OH * p.OW;
    const int64_t M64 = (int64_t)a.N * p.ohw;
    DSD_CHECK(M64 < (1ll << 31) && M64 * std::max(a.Cout, a.Cin) < (1ll << 40), "conv2d: problem too large");
    p.M = (int)M64;
    p.Ktot = a.ks * a.ks * a.Cin;
    p.cchunks = cdiv(a.Cin, BK);
    p.tiles_m = cdiv(p.M, BM);
    if (p.M == 0 || a.Cout == 0) return;

    if (a.Cin % 4 != 0 || p.Ktot < 32) {
        const size_t lds = (size_t)p.Ktot * a.Cout * sizeof(float);
        if (direct_cols_ok(a)) {
            const int W4 = a.Cout / 4, rpi = 256 / W4;
            if (a.stats) {
                const int ppb = direct_cols_ppb(a, p.M, p.ohw, true);
                DSD_CHECK(ppb > 0 && p.ohw / ppb == a.stats_chunks, "conv2d: statistics chunks %d do not match this launch", a.stats_chunks);
                hipLaunchKernelGGL((conv_direct_cols_kernel<9, true>), dim3(p.M / ppb), dim3(rpi * W4), 0, s, p, W4, ppb, a.stats, a.stats_chunks);
            } else {
                const int ppb = direct_cols_ppb(a, p.M, p.ohw, false);
                hipLaunchKernelGGL((conv_direct_cols_kernel<9, false>), dim3(cdiv(p.M, ppb)), dim3(rpi * W4), 0, s, p, W4, ppb, nullptr, 0);
            }
            check_launch("conv_direct_cols");
        } else if (a.Cout % 4 == 0 && lds <= DIRECT_LDS_MAX) {
            // (beyond the 64 KB default the dynamic LDS size has to be allowed once per process; the latent U-Net's first layer,
            // 6 -> 320 channels, needs 69 KB and used to fall to the scalar kernel: ~1 ms per evaluation)
            static const hipError_t lds_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_direct_lds_kernel),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIRECT_LDS_MAX);
            DSD_CHECK(lds_attr == hipSuccess, "hipFuncSetAttribute(conv_direct_lds_kernel): %s", hipGetErrorString(lds_attr));
            hipLaunchKernelGGL(conv_direct_lds_kernel, dim3(cdiv(p.M, 64)), dim3(256), lds, s, p);
            check_launch("conv_direct_lds");
        } else {
            const int blocks = (int)std::min<int64_t>(((int64_t)p.M * a.Cout + 255) / 256, 1 << 20);
            hipLaunchKernelGGL(conv_scalar_kernel, dim3(blocks), dim3(256), 0, s, p);
            check_launch("conv_scalar");
        }
        return;
    }
    const int prec = effective_precision(a, p.tiles_m);
    int nt = pick_nt(a.Cout, p.tiles_m, prec, a.lanes);
    if (prec == PREC_BF16X6 && conv2d_wino_eligible(a)) {
        conv2d_wino(a, s);
        return;
    }
    if (prec != PREC_F32) {
        int ks = 1, ad = 0;
        // a caller without scratch (never the planned graph) gets the unsplit configuration
        conv2d_split_plan(a, nt, &nt, &ks, &ad, a.scratch != nullptr);
        conv2d_split(a, nt, ks, ad, s);
        return;
    }
    p.tiles_n = cdiv(a.Cout, nt * 32);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    {
        // default path: buffer-load kernel when the operands are addressable with 32-bit byte offsets
        const int64_t xb = ((int64_t)(a.N - 1) * p.x_bs + (int64_t)a.H * a.W * a.Cin) * 4;
        const int64_t wb = (int64_t)a.Cout * p.Ktot * 4;
        const bool buf_ok = a.Cin % BK == 0 && xb < 0xFFFFFF00ll && wb < 0xFFFFFF00ll && p.x_bs >= 0;
        if (buf_ok && (conv_variant(a) == 0)) {
            p.x_bytes = (unsigned)xb;
            p.w_bytes = (unsigned)wb;
            switch (nt) {
                case 1: hipLaunchKernelGGL(conv_mfma_buf_kernel<1>, grid, dim3(256), 0, s, p); break;
                case 2: hipLaunchKernelGGL(conv_mfma_buf_kernel<2>, grid, dim3(256), 0, s, p); break;
                case 3: hipLaunchKernelGGL(conv_mfma_buf_kernel<3>, grid, dim3(256), 0, s, p); break;
                case 4: hipLaunchKernelGGL(conv_mfma_buf_kernel<4>, grid, dim3(256), 0, s, p); break;
                default: hipLaunchKernelGGL(conv_mfma_buf_kernel<5>, grid, dim3(256), 0, s, p); break;
            }
            check_launch("conv_mfma_buf");
            return;
        }
    }
    switch (nt) {
        case 1: hipLaunchKernelGGL(conv_mfma_kernel<1>, grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(conv_mfma_kernel<2>, grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(conv_mfma_kernel<3>, grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL(conv_mfma_kernel<4>, grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL(conv_mfma_kernel<5>, grid, dim3(256), 0, s, p); break;
    }
    check_launch("conv_mfma");
}

}  // namespace dsd
