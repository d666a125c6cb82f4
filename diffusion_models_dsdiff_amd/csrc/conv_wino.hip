// conv_wino.hip — 3x3 stride-1 convolution with HALF the Winograd trick: F(2,3) along the image width only, on the bf16
// matrix cores with fp32 operands split exactly into three bf16 pieces (the bf16x6 arithmetic of conv_split.hip).
//
// 97 % of the network's FLOPs are 3x3 stride-1 convolutions.  Writing two horizontally adjacent outputs of one filter row as
//     m0 = (d0-d2) g0        m1 = (d1+d2) (g0+g1+g2)/2        m2 = (d2-d1) (g0-g1+g2)/2        m3 = (d1-d3) g2
//     y0 = m0 + m1 + m2      y1 = m1 - m2 - m3                                  (d = 4 input pixels, g = 3 filter taps)
// takes 4 multiplications where the direct form takes 6, for all three filter rows alike: 1.5x fewer MFMAs.  The four
// products are four GEMMs   Y_p[tile][co] = sum_{kh,c} U_p[tile][kh,c] * V_p[co][kh,c]   over M/2 two-pixel tiles with
// K = 3*Cin: U_p are sums / differences of activations formed in fp32 (one rounding, like any fp32 re-association of the
// convolution sum), V_p are formed once from the weights in fp64 and rounded to fp32; both are then split into bf16 pieces
// exactly as in the direct kernel, so no operand is narrower than fp32.  The output transform is two adds per output in
// the epilogue.  The full 2-D F(2x2,3x3) was analysed and dropped: its 16 positions need 16 accumulator sets per wave (or a
// cross-wave reduction of partial tiles through HBM); the 1-D form keeps all four positions of a tile in one
// wave, so it fits the "A-direct" structure of conv_split_ad_kernel:
//
//   * a lane owns one tile (two output pixels) and 8 consecutive channels: it loads its 4 input pixels x 32 B straight
//     from HBM/L2 (buffer loads, hardware range check = zero padding), forms U_0..U_3 in registers and splits them;
//   * the transformed weights are packed at plan time in exactly the order the kernel stages them —
//     [N-tile][k-tile = (16-channel chunk, kh)][piece][position][row][16 k] — so one k-tile is 48 KB contiguous and goes
//     global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write); the LDS image is lane-linear,
//     the bank-conflict-free XOR swizzle of the 32-byte rows is applied on the SOURCE address and on the fragment reads
//     (cdna_hip_programming.md rule 21);
//   * block tile = 128 tiles (256 output pixels) x 128 output channels x 4 positions = 256 accumulator registers per lane,
//     one workgroup per CU, one barrier per k-tile (16 k of one filter row: 96 MFMAs per wave), two LDS stages.
//   * k-tile order: filter rows inner, channel chunks outer (a 128-byte line of a pixel is re-read while hot).
#include "kernels.h"

#include <cstdlib>

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr unsigned OOB = 0xFFFFFFF0u;
// 4 positions x 4 column tiles x 16 = 256 accumulator registers = the whole AGPR file (the compiler keeps every MFMA
// accumulator of a kernel in AGPRs once it uses them at all, so 5 column tiles = 320 spill); a layer whose Cout is not a
// multiple of 128 gets a narrower LAST N tile whose missing column tiles are simply not multiplied (Cout = 320: 128+128+64)
static constexpr int WNT = 4;
static constexpr int WBROWS = WNT * 32;
static constexpr int WPLANE = WBROWS * 32;        // bytes of one (piece, position) plane of a k-tile: 128 rows x 16 bf16
static constexpr int WSTAGE = 12 * WPLANE;        // 3 pieces x 4 positions = 49152 B
static constexpr int WTILES = 128;                // two-pixel tiles per block

struct WinoP {
    const float* x;
    const unsigned char* wp;   // packed transformed weight pieces (wino_pack_weights)
    const float* bias;
    const float* emb;
    const float* res;
    float* y;
    int64_t x_bs;
    int N, H, W, Cin, Cout, emb_stride, y_ld;   // H, W: OUTPUT size (= input size, or twice it with the folded nearest x2)
    int ups, IW;                                  // ups: the input is [N, H >> 1, W >> 1, Cin], pixel (h, w) reads (h >> 1, w >> 1)
    int M, T, halfW, ohw, tps;   // output pixels, tiles (= M/2), tiles per image row, pixels / tiles per sample
    int KT, cchunks;             // k-tiles = 3 * Cin/16
    int tiles_m, tiles_n;
    unsigned x_bytes;
    double* stats;
    int stats_chunks;
    int whatif;   // experiments (env DSD_WINO_WHATIF): 1 = no activation loads in the k-loop, 2 = no weight DMA, 4 = no transform/split
};

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 t;
    t[0] = (__bf16)a;
    t[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// 8 fp32 (two float4) -> three bf16x8 pieces, a = p0 + p1 + p2 up to 2^-24 |a|
__device__ __forceinline__ void split8x3(f32x4 lo, f32x4 hi, bf16x8 (&out)[3]) {
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pk = pk_bf16(v[2 * i], v[2 * i + 1]);
            w[i] = pk;
            if (q < 2) {
                v[2 * i] -= bf_lo(pk);
                v[2 * i + 1] -= bf_hi(pk);
            }
        }
        out[q] = __builtin_bit_cast(bf16x8, w);
    }
}

// Measured and removed (round 2): giving the narrow last N tile of TWO neighbouring M tiles to one block, so that every
// block carries the same MFMA work and the blocks sharing activations stay in step: 228.9 / 240.1 TF/s against 232.1 / 247.4
// unpaired on 16x256x256x{320,640}->320 — the faster 640-channel layers (267-277) are not faster because their N tiles are
// equal.  What separates this kernel from its MFMA skeleton is vector-memory ISSUE: per k-tile and wave 4 activation + 12
// weight LDS-DMA instructions at ~60 cycles each beside 96 MFMAs (3072 cycles), twice the direct kernel's ratio.
//
// NT = column tiles (of 32 output channels) this block really has: 4, or 2 in the last N tile of a layer whose Cout is
// 64 (mod 128).  A compile-time count: a run-time `if (j < ntv)` around the MFMAs makes every accumulator a phi of
// "multiplied / not multiplied", which doubles the AGPR demand and spills the whole accumulator file.
//
// Version 2 of the activation path.  Version 1 had every lane fetch its four pixels x 32 B straight from global memory
// (8 buffer loads per k-tile, each touching 64 different 128-byte lines): measured, the kernel ran no faster than the direct
// one although it issues 1.5x fewer MFMAs, and removing those loads alone (what-if build) recovered the difference — the
// texture-address path, not the matrix pipe, was the limit.  Now the block's pixels come in by LDS-DMA as 64-byte entries
// (16 channels of one pixel = half a cache line, 16 entries per wave-instruction), even and odd image columns in two
// arrays so that the tiles' neighbours are neighbouring entries, XOR-swizzled through the SOURCE address so that the
// fragment reads are bank-conflict-free; lanes then read their four pixels from LDS.  Three activation stages: the tile
// after next is in flight while the next one is already readable, because the first product of a tile needs its
// activations transformed BEFORE its barrier.
static constexpr int AENT = 64;                       // bytes of one pixel entry: 16 channels fp32
static constexpr int ASTAGE = 2 * WTILES * AENT;      // even-column entries [128] | odd-column entries [128] = 16 KB
static constexpr int A_BASE = 2 * WSTAGE;             // activation stages follow the two weight stages
static constexpr int Z_OFF = A_BASE + 3 * ASTAGE;     // 64 zero bytes: every padding pixel reads here
static constexpr int WLDS = Z_OFF + 64;

template <int NT>
__device__ __forceinline__ void wino_body(const WinoP& p, unsigned char* Bs) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 31;
    const int half = lane >> 5;
    int L = blockIdx.x;
    {
        const int cpx = (int)gridDim.x >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);   // consecutive tiles on one XCD (its L2)
    }
    // whatif & 8: N tiles outermost (every CU of an XCD streams the same 2.4 MB of packed weights: L2-resident)
    const int tile_n = (p.whatif & 8) ? L / p.tiles_m : L % p.tiles_n;
    const int tile_m = (p.whatif & 8) ? L % p.tiles_m : L / p.tiles_n;
    const int t0 = tile_m * WTILES;
    const int n0 = tile_n * WBROWS;
    if (tid < 4) *reinterpret_cast<f32x4*>(Bs + Z_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- compute role: this lane's tile = output pixels (n, oh, 2 tw), (n, oh, 2 tw + 1); its input pixels of filter row
    // kh are columns 2 tw - 1 .. 2 tw + 2 of image row oh - 1 + kh = entries O[tl-1], E[tl], O[tl], E[tl+1] (tl = tile index
    // inside the block; a block starts at an image-row start, so the entries outside 0..127 are exactly the zero padding)
    const int tl = wave * 32 + lrow;
    const int t = t0 + tl;
    const bool a_ok = t < p.T;
    const int tt = a_ok ? t : p.T - 1;
    const int ns = tt / p.tps;
    const int rem = tt - ns * p.tps;
    const int oh = rem / p.halfW;
    const int tw = rem - oh * p.halfW;
    // stage-relative byte offsets of the two 16-byte slots (this lane's 8 channels) of its four pixels; slot s of entry i
    // lives at physical slot s ^ ((i >> 2) & 3)
    int aoff[4][2];
    {
        const int idx[4] = {tl - 1, tl, tl, tl + 1};
        const int arr[4] = {1, 0, 1, 0};   // O, E, O, E
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                aoff[j][sl] = arr[j] * (WTILES * AENT) + idx[j] * AENT + (((half * 2 + sl) ^ ((idx[j] >> 2) & 3)) << 4);
    }
    const bool padl = tw == 0, padr = tw == p.halfW - 1;
    const unsigned okmask = ((unsigned)(oh - 1) < (unsigned)p.H ? 1u : 0u) | 2u | ((unsigned)(oh + 1) < (unsigned)p.H ? 4u : 0u);

    // ---- loader role: LDS-DMA chunk q = (wave*4 + i)*64 + lane of an activation stage: entry e = q >> 2 (array e >> 7,
    // index e & 127), physical slot q & 3 holding logical slot (q & 3) ^ ((index >> 2) & 3) of that pixel's 64 bytes
    unsigned dcol[4];
    int doh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (wave * 4 + i) * 64 + lane;
        const int e = q >> 2, arr = e >> 7, idx = e & 127;
        const int ls = (q & 3) ^ ((idx >> 2) & 3);
        const int tg = min(t0 + idx, p.T - 1);
        const int gn = tg / p.tps;
        const int gr = tg - gn * p.tps;
        const int goh = gr / p.halfW;
        const int gtw = gr - goh * p.halfW;
        doh[i] = goh;
        // byte offset of (row 0, source column of output column 2 gtw + arr, channel 4 ls) of sample gn
        dcol[i] = ((unsigned)gn * (unsigned)p.x_bs + (unsigned)((2 * gtw + arr) >> p.ups) * (unsigned)p.Cin + (unsigned)(ls * 4)) * 4u;
    }
    const unsigned rowpitch = (unsigned)p.IW * (unsigned)p.Cin * 4u;
    auto dma_a = [&](int kt, int stage_off) {
        if (kt >= p.KT) return;
        // k-tile order: 32-channel chunk, filter row, 16-channel half — the two 64-byte halves of a pixel's 128-byte line are
        // fetched by CONSECUTIVE tiles (the first brings the line into L2, the second hits it)
        const int c32 = kt / 6, sub = kt - c32 * 6, kh = sub >> 1, cc = c32 * 2 + (sub & 1);
        const int soff = __builtin_amdgcn_readfirstlane(cc * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = doh[i] - 1 + kh;
            const unsigned v = (unsigned)ih < (unsigned)p.H ? dcol[i] + (unsigned)(ih >> p.ups) * rowpitch : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (void __attribute__((address_space(3)))*)(Bs + stage_off + (wave * 4 + i) * 1024),
                                                     16, v, soff, 0, 0);
        }
    };

    // ---- weights: the k-tile kt of this N tile is WSTAGE contiguous bytes; LDS-DMA chunk q = i*256 + tid lands at LDS byte
    // q*16 (lane-linear) and must hold the logical chunk q ^ ((q >> 4) & 1)  ((q>>4)&1 = (row>>3)&1 of the chunk's row)
    const unsigned char* wt = p.wp + (size_t)tile_n * p.KT * WSTAGE + (size_t)((tid ^ ((tid >> 4) & 1)) * 16);
    auto dma_b = [&](int kt, int so) {
        if (kt >= p.KT) return;
        if (NT == 2 && wave >= 2) return;   // narrow N tile: rows 64..127 of every plane (waves 2, 3 of each 4 KB round) are never read
        const unsigned char* src = wt + (size_t)kt * WSTAGE;
#pragma unroll
        for (int i = 0; i < WSTAGE / 4096; ++i)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 4096),
                                             (void __attribute__((address_space(3)))*)(Bs + so + i * 4096 + wave * 1024), 16, 0, 0);
    };
    // fragment of (plane pp, column tile j): row j*32 + lrow, 16-byte chunk half ^ ((row >> 3) & 1)
    const int frag_off = lrow * 32 + ((half ^ ((lrow >> 3) & 1)) << 4);

    f32x16 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][j][e] = 0.f;

    bf16x8 af[4][3];
    // pixel j of filter row kh from the activation stage at byte offset `st` (zero entry for padding)
    auto rd = [&](int j, int st, int kh, f32x4& lo, f32x4& hi) {
        const bool pad = (j == 0 && padl) || (j == 3 && padr) || !((okmask >> kh) & 1u);
        const int a0 = pad ? Z_OFF : st + aoff[j][0];
        const int a1 = pad ? Z_OFF + 16 : st + aoff[j][1];
        lo = *reinterpret_cast<const f32x4*>(Bs + a0);
        hi = *reinterpret_cast<const f32x4*>(Bs + a1);
    };
    auto mfma6 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& c) {   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    };

    // ---- prologue: weights of tile 0, activations of tiles 0 and 1; after the barrier U_0 of tile 0 is formed (the only
    // transform that is not hidden behind MFMAs)
    dma_b(0, 0);
    dma_a(0, A_BASE);
    dma_a(1, A_BASE + ASTAGE);
    __syncthreads();
    f32x4 d1l, d1h, d2l, d2h, xl, xh, yl, yh;
    rd(0, A_BASE, 0, xl, xh);
    rd(2, A_BASE, 0, yl, yh);
    split8x3(xl - yl, xh - yh, af[0]);

    // One k-tile = 4 positions x NT units of 6 MFMAs, position-major; the transform of each position runs one position ahead
    // of its products, spread over the units of the position before:
    //   during position 0:  d1, d2 of this tile -> U_1 = d1 + d2 -> af[1]
    //   during position 1:  U_2 = d2 - d1 -> af[2];  d3 is read
    //   during position 2:  U_3 = d1 - d3 -> af[3];  d0, d2 of the NEXT tile are read (its stage is already visible)
    //   during position 3:  U_0 = d0 - d2 of the next tile -> af[0]
    // and at the top of the tile the DMAs of the weights of tile kt+1 and of the activations of tile kt+2 go out.
    int sub6 = 0, ast = 0;   // position of tile kt inside its group of 6 (kh = sub6 >> 1); index (0..2) of its activation stage
    for (int kt = 0; kt < p.KT; ++kt) {
        const int so = (kt & 1) * WSTAGE;
        const int st_cur = A_BASE + ast * ASTAGE;
        const int ast1 = ast == 2 ? 0 : ast + 1, ast2 = ast1 == 2 ? 0 : ast1 + 1;
        const int st_nxt = A_BASE + ast1 * ASTAGE;
        const int sub61 = sub6 == 5 ? 0 : sub6 + 1;
        const int kh = sub6 >> 1, kh1 = sub61 >> 1;
        if (!(p.whatif & 2)) dma_b(kt + 1, WSTAGE - so);
        if (!(p.whatif & 1)) dma_a(kt + 2, A_BASE + ast2 * ASTAGE);
        const unsigned char* bf = Bs + so + frag_off;
        bf16x8 b_cur[3], b_nxt[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b_nxt[q] = b_cur[q] = *reinterpret_cast<const bf16x8*>(bf + (q * 4) * WPLANE);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4 * NT; ++u) {
            const int pos = u / NT, j = u % NT;
            if (u + 1 < 4 * NT) {
                const int pos1 = (u + 1) / NT, j1 = (u + 1) % NT;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    b_nxt[q] = *reinterpret_cast<const bf16x8*>(bf + (q * 4 + pos1) * WPLANE + j1 * 32 * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(p.whatif & 4)) {
                if (u == 0) {
                    rd(1, st_cur, kh, d1l, d1h);
                    rd(2, st_cur, kh, d2l, d2h);
                }
                if (u == NT - 1) split8x3(d1l + d2l, d1h + d2h, af[1]);
                if (u == NT) rd(3, st_cur, kh, xl, xh);
                if (u == 2 * NT - 1) split8x3(d2l - d1l, d2h - d1h, af[2]);
                if (u == 2 * NT) {
                    rd(0, st_nxt, kh1, yl, yh);
                    rd(2, st_nxt, kh1, d2l, d2h);
                }
                if (u == 3 * NT - 1) split8x3(d1l - xl, d1h - xh, af[3]);
                if (u == 4 * NT - 1) split8x3(yl - d2l, yh - d2h, af[0]);
            }
            mfma6(af[pos], b_cur, acc[pos][j]);
#pragma unroll
            for (int q = 0; q < 3; ++q) b_cur[q] = b_nxt[q];
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // DMAs issued at the top have landed (vmcnt(0) before the barrier); every wave left this tile's stages
        sub6 = sub61;
        ast = ast1;
    }

    // ---- epilogue: y0 = Y0 + Y1 + Y2, y1 = Y1 - Y2 - Y3, + bias + per-(sample, channel) embedding + residual
    // GroupNorm statistics as shifted fp32 partials per lane and column (see StatAcc in conv_split.hip for why shifted)
    float cs[NT], cq[NT], cref[NT];
    int ccnt[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        cs[j] = cq[j] = cref[j] = 0.f;
        ccnt[j] = 0;
    }
    const bool want_stats = p.stats != nullptr;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 32 + lrow;
        const bool n_ok = n < p.Cout;
        const float bj = (p.bias && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tr = (e & 3) + 8 * (e >> 2) + 4 * half;
            const int te = t0 + wave * 32 + tr;
            if (te >= p.T || !n_ok) continue;
            const int64_t m = 2 * (int64_t)te;
            const float y0 = (acc[0][j][e] + acc[1][j][e]) + acc[2][j][e];
            const float y1 = (acc[1][j][e] - acc[2][j][e]) - acc[3][j][e];
            float v0 = y0 + bj, v1 = y1 + bj;
            if (p.emb) {
                const float ev = p.emb[(int64_t)(m / p.ohw) * p.emb_stride + n];
                v0 += ev;
                v1 += ev;
            }
            if (p.res) {
                v0 += p.res[m * p.Cout + n];
                v1 += p.res[(m + 1) * p.Cout + n];
            }
            p.y[m * p.y_ld + n] = v0;
            p.y[(m + 1) * p.y_ld + n] = v1;
            if (want_stats) {
                if (ccnt[j] == 0) cref[j] = v0;
                const float d0 = v0 - cref[j], d1 = v1 - cref[j];
                cs[j] += d0 + d1;
                cq[j] = fmaf(d0, d0, fmaf(d1, d1, cq[j]));
                ccnt[j] += 2;
            }
        }
    }
    if (want_stats) {   // GroupNorm statistics of the block tile's 256 pixels (GnSrc layout; see conv_split.hip stats_reduce)
        double* red = reinterpret_cast<double*>(Bs);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const double r = (double)cref[j], S = (double)cs[j], nn = (double)ccnt[j];
            double s = S + nn * r, q = (double)cq[j] + 2.0 * r * S + nn * r * r;
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (half == 0) {
                red[((wave * WBROWS) + j * 32 + lrow) * 2 + 0] = s;
                red[((wave * WBROWS) + j * 32 + lrow) * 2 + 1] = q;
            }
        }
        __syncthreads();
        const int m0 = 2 * t0;
        const int nb = m0 / p.ohw;
        const int chunk = (m0 - nb * p.ohw) / (2 * WTILES);
        for (int c = tid; c < WBROWS; c += 256) {
            const int n = n0 + c;
            if (n >= p.Cout) continue;
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += red[((w * WBROWS) + c) * 2 + 0];
                q += red[((w * WBROWS) + c) * 2 + 1];
            }
            double* o = p.stats + (((int64_t)nb * p.stats_chunks + chunk) * p.Cout + n) * 2;
            o[0] = s;
            o[1] = q;
        }
    }
}

__global__ __launch_bounds__(256, 1) void conv_wino_kernel(WinoP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char Bs[WLDS];
    int L = blockIdx.x;
    {
        const int cpx = (int)gridDim.x >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const int tn = (p.whatif & 8) ? L / p.tiles_m : L % p.tiles_n;
    const bool narrow = tn == p.tiles_n - 1 && (p.Cout % WBROWS) != 0;   // uniform: last N tile of a Cout = 64 (mod 128) layer
    if (narrow)
        wino_body<2>(p, Bs);
    else
        wino_body<WNT>(p, Bs);
}

// w (OHWI fp32: [Cout][3][3][Cin]) -> transformed, split and packed:
//   out[tile_n][kt = (cc/2)*6 + kh*2 + cc%2][piece][pos][row][16]  bf16,   co = tile_n*128 + row,  ci = cc*16 + k
//   V_0 = g0, V_1 = (g0+g1+g2)/2, V_2 = (g0-g1+g2)/2, V_3 = g2   (g_kw = w[co][kh][kw][ci]; fp64, one rounding to fp32)
__global__ void wino_pack_kernel(const float* __restrict__ w, int Cout, int Cin, unsigned short* __restrict__ out) {
    const int64_t total = (int64_t)Cout * 3 * Cin;
    const int KT = 3 * (Cin / 16);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        const int kh = (int)((i / Cin) % 3);
        const int co = (int)(i / ((int64_t)Cin * 3));
        const float* g = w + (((int64_t)co * 3 + kh) * 3) * Cin + ci;
        const double g0 = g[0], g1 = g[Cin], g2 = g[2 * (int64_t)Cin];
        const float v[4] = {(float)g0, (float)((g0 + g1 + g2) * 0.5), (float)((g0 - g1 + g2) * 0.5), (float)g2};
        const int tile_n = co / WBROWS, row = co - tile_n * WBROWS;
        const int cc = ci / 16, k = ci - cc * 16;
        const int kt = (cc >> 1) * 6 + kh * 2 + (cc & 1);   // the kernel's k-tile order: (32-channel chunk, kh, 16-channel half)
        unsigned short* base = out + ((int64_t)tile_n * KT + kt) * (WSTAGE / 2);
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            float r = v[pos];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const __bf16 b = (__bf16)r;
                base[((q * 4 + pos) * WBROWS + row) * 16 + k] = __builtin_bit_cast(unsigned short, b);
                r -= (float)b;
            }
        }
    }
}

size_t wino_packed_bytes(int Cout, int Cin) { return (size_t)cdiv(Cout, WBROWS) * (3 * (Cin / 16)) * WSTAGE; }

void wino_pack_weights(const float* w_ohwi, int Cout, int Cin, void* packed, hipStream_t s) {
    if (Cout % WBROWS != 0) DSD_HIP(hipMemsetAsync(packed, 0, wino_packed_bytes(Cout, Cin), s));   // rows past Cout: never multiplied
    const int64_t total = (int64_t)Cout * 3 * Cin;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, w_ohwi, Cout,
                       Cin, (unsigned short*)packed);
    check_launch("wino_pack");
}

// 3x3, stride 1, symmetric padding, no folded upsample, width a power of two <= 256, Cin % 32 == 0, Cout = 0 or 64 (mod 128), bf16x6,
// operands addressable with 32-bit byte offsets, and enough tiles to fill the chip (small grids keep the split-K kernels)
bool conv2d_wino_eligible(const ConvArgs& a) { return a.w_wino != nullptr && conv2d_wino_shape_ok(a); }

bool conv2d_wino_shape_ok(const ConvArgs& a) {
    static const bool off = getenv("DSD_NO_WINOGRAD") != nullptr;   // experiments only
    if (off || a.ks != 3 || a.stride != 1 || a.pad_lo >= 0 || a.pad_total >= 0 || a.out_nchw) return false;
    if (a.precision != PREC_BF16X6) return false;
    const int OW = a.ups ? 2 * a.W : a.W, OH = a.ups ? 2 * a.H : a.H;
    if (OW % 2 != 0 || a.Cin % 32 != 0 || (a.Cout % WBROWS != 0 && a.Cout % WBROWS != 64)) return false;
    if (WTILES % (OW / 2) != 0) return false;   // a block (128 tiles) must start at an image-row start: output W = 2, 4, ..., 256
    const int64_t x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    const int64_t xb = ((int64_t)(a.N - 1) * x_bs + (int64_t)a.H * a.W * a.Cin) * 4;
    if (xb >= 0xFFFFFF00ll) return false;
    const int64_t M = (int64_t)a.N * OH * OW;
    return M >= 4096 && M < (1ll << 31);
}

// At least two full rounds of workgroups over the 256 CUs; smaller grids stay on the direct / split-K kernels (measured:
// 16 x 32x32 x 640->640 = 320 workgroups runs at 185 TF/s here against 224 direct, 16 x 64x64 = 1280 workgroups 275 vs 224).
bool conv2d_wino_worthwhile(const ConvArgs& a) {
    if (!conv2d_wino_shape_ok(a)) return false;
    static const int min_blocks = getenv("DSD_WINO_MIN_BLOCKS") ? atoi(getenv("DSD_WINO_MIN_BLOCKS")) : 512;
    const int64_t M = (int64_t)a.N * a.H * a.W * (a.ups ? 4 : 1);
    const int64_t blocks = ((M / 2 + WTILES - 1) / WTILES) * ((a.Cout + WBROWS - 1) / WBROWS);
    return blocks >= min_blocks;
}

int conv2d_wino_stats_chunks(const ConvArgs& a) {
    const int ohw = (a.ups ? 4 : 1) * a.H * a.W;
    return ohw % (2 * WTILES) == 0 ? ohw / (2 * WTILES) : 0;
}

void conv2d_wino(const ConvArgs& a, hipStream_t s) {
    WinoP p{};
    p.x = a.x; p.wp = (const unsigned char*)a.w_wino; p.bias = a.bias; p.emb = a.emb; p.res = a.res; p.y = a.y;
    p.N = a.N; p.Cin = a.Cin; p.Cout = a.Cout; p.emb_stride = a.emb_stride;
    p.ups = a.ups ? 1 : 0;
    p.H = a.H << p.ups; p.W = a.W << p.ups; p.IW = a.W;
    p.y_ld = a.y_ld > 0 ? a.y_ld : a.Cout;
    p.x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    p.ohw = p.H * p.W;
    p.M = (int)((int64_t)a.N * p.ohw);
    p.T = p.M / 2;
    p.halfW = p.W / 2;
    p.tps = p.ohw / 2;
    p.cchunks = a.Cin / 16;
    p.KT = 3 * p.cchunks;
    p.tiles_m = cdiv(p.T, WTILES);
    p.tiles_n = cdiv(a.Cout, WBROWS);
    p.x_bytes = (unsigned)(((int64_t)(a.N - 1) * p.x_bs + (int64_t)a.H * a.W * a.Cin) * 4);
    static const int whatif = getenv("DSD_WINO_WHATIF") ? atoi(getenv("DSD_WINO_WHATIF")) : 0;
    p.whatif = whatif;
    p.stats = nullptr;
    if (a.stats) {
        DSD_CHECK(p.ohw % (2 * WTILES) == 0 && p.ohw / (2 * WTILES) == a.stats_chunks,
                  "conv2d_wino: output statistics requested with %d chunks, the kernel emits %d", a.stats_chunks,
                  p.ohw % (2 * WTILES) == 0 ? p.ohw / (2 * WTILES) : 0);
        p.stats = a.stats;
        p.stats_chunks = a.stats_chunks;
    }
    hipLaunchKernelGGL(conv_wino_kernel, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), 0, s, p);
    check_launch("conv_wino");
}

}  // namespace dsd
