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
    int N, H, W, Cin, Cout, emb_stride, y_ld;
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

// NT = column tiles (of 32 output channels) this block really has: 4, or 2 in the last N tile of a layer whose Cout is
// 64 (mod 128).  A compile-time count: a run-time `if (j < ntv)` around the MFMAs makes every accumulator a phi of
// "multiplied / not multiplied", which doubles the AGPR demand and spills the whole accumulator file.
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
    const int tile_n = L % p.tiles_n;
    const int tile_m = L / p.tiles_n;
    const int t0 = tile_m * WTILES;
    const int n0 = tile_n * WBROWS;

    // ---- this lane's tile: output pixels (n, oh, 2 tw) and (n, oh, 2 tw + 1); input pixels (oh - 1 + kh, 2 tw - 1 + j)
    const int t = t0 + wave * 32 + lrow;
    const bool a_ok = t < p.T;
    const int tt = a_ok ? t : 0;
    const int ns = tt / p.tps;
    const int rem = tt - ns * p.tps;
    const int oh = rem / p.halfW;
    const int tw = rem - oh * p.halfW;
    unsigned rowoff[3], coloff[4];
    bool rowok[3], colok[4];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh - 1 + kh;
        rowok[kh] = a_ok & ((unsigned)ih < (unsigned)p.H);
        rowoff[kh] = ((unsigned)ns * (unsigned)p.x_bs + (unsigned)(ih * p.W) * (unsigned)p.Cin + (unsigned)(half * 8)) * 4u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int iw = 2 * tw - 1 + j;
        colok[j] = (unsigned)iw < (unsigned)p.W;
        coloff[j] = (unsigned)iw * (unsigned)p.Cin * 4u;
    }

    // ---- weights: the k-tile kt of this N tile is WSTAGE contiguous bytes; LDS-DMA chunk q = i*256 + tid lands at LDS byte
    // q*16 (lane-linear) and must hold the logical chunk q ^ ((q >> 4) & 1)  ((q>>4)&1 = (row>>3)&1 of the chunk's row)
    const unsigned char* wt = p.wp + (size_t)tile_n * p.KT * WSTAGE + (size_t)((tid ^ ((tid >> 4) & 1)) * 16);
    auto dma_tile = [&](int kt, int so) {
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

    // activations: raw[set][pixel j][16-byte half]; after the in-place transform raw[set][q] holds U_q
    f32x4 raw[2][4][2];
    bf16x8 af[4][3];
    auto load_raw = [&](int set, int kt) {
        // clamped: the look-ahead past the last k-tile re-reads the last one (its data is never multiplied)
        const int k = min(kt, p.KT - 1);
        const int cc = k / 3, kh = k - cc * 3;
        const int soff = __builtin_amdgcn_readfirstlane(cc * 64);
        const unsigned ro = kh == 0 ? rowoff[0] : (kh == 1 ? rowoff[1] : rowoff[2]);
        const bool rk = kh == 0 ? rowok[0] : (kh == 1 ? rowok[1] : rowok[2]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rk & colok[j];
            const unsigned v0 = ok ? ro + coloff[j] : OOB;
            const unsigned v1 = ok ? ro + coloff[j] + 16u : OOB;
            raw[set][j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v0, soff, 0));
            raw[set][j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, v1, soff, 0));
        }
    };
    auto transform = [&](int set) {   // d0..d3 -> U0 = d0-d2, U1 = d1+d2, U2 = d2-d1, U3 = d1-d3 (in place)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 d0 = raw[set][0][h], d1 = raw[set][1][h], d2 = raw[set][2][h], d3 = raw[set][3][h];
            raw[set][0][h] = d0 - d2;
            raw[set][1][h] = d1 + d2;
            raw[set][2][h] = d2 - d1;
            raw[set][3][h] = d1 - d3;
        }
    };
    auto mfma6 = [&](const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& c) {   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    };

    // ---- prologue: weights of tile 0 -> stage 0; activations of tile 0 -> af[0..2] and U_3 (left in raw[0][3], split at
    // position 0 of tile 0 like in every other tile), of tile 1 -> raw[1] (in flight)
    dma_tile(0, 0);
    load_raw(0, 0);
    transform(0);
#pragma unroll
    for (int q = 0; q < 3; ++q) split8x3(raw[0][q][0], raw[0][q][1], af[q]);
    load_raw(1, 1);

    // One k-tile = 4 positions x NT units of 6 MFMAs.  While tile kt is multiplied (position-major):
    //   start         the LDS-DMA of the weights of tile kt+1 into the other stage (a whole tile to land)
    //   position 0    af[3] of THIS tile from the U_3 kept since the previous tile
    //   after pos. 1  the activations of tile kt+1 (loaded during tile kt-1) are transformed in place; U_0, U_1 -> af[0], af[1]
    //                 (dead for this tile), then the loads of tile kt+2 go out into the other register set
    //   after pos. 2  U_2 -> af[2];   U_3 stays in its registers until position 0 of the next tile
    // so the activation fragments need no second buffer.  `cur` = register set holding tile kt+1.
    // The two register sets alternate by tile parity, so the loop body is TWO k-tiles of straight-line code (KT = 3 * Cin/16
    // is even: Cin % 32 == 0).  No branch may surround an MFMA: the 256 accumulators fill the AGPR file, and a control-flow
    // join would need copies of them.
#define DSD_WINO_TILE(KT_, SO, CUR, OTH)                                                                                     \
    {                                                                                                                        \
        __syncthreads(); /* DMA of this tile has landed (vmcnt(0) before the barrier); every wave left the other stage */    \
        if ((KT_) + 1 < p.KT && !(p.whatif & 2)) dma_tile((KT_) + 1, WSTAGE - (SO));                                         \
        const unsigned char* bf = Bs + (SO) + frag_off;                                                                      \
        bf16x8 b_cur[3], b_nxt[3];                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) b_nxt[q] = b_cur[q] = *reinterpret_cast<const bf16x8*>(bf + (q * 4) * WPLANE); \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 4 * NT; ++u) {                                                                 \
            const int pos = u / NT, j = u % NT;                                                                              \
            if (u + 1 < 4 * NT) {                                                                                            \
                const int pos1 = (u + 1) / NT, j1 = (u + 1) % NT;                                                            \
                _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                \
                    b_nxt[q] = *reinterpret_cast<const bf16x8*>(bf + (q * 4 + pos1) * WPLANE + j1 * 32 * 32);                \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            if (!(p.whatif & 4)) {                                                                                           \
                if (u == 0) split8x3(raw[OTH][3][0], raw[OTH][3][1], af[3]);                                                 \
                if (u == 2 * NT) {                                                                                           \
                    transform(CUR);                                                                                          \
                    split8x3(raw[CUR][0][0], raw[CUR][0][1], af[0]);                                                         \
                }                                                                                                            \
                if (u == 2 * NT + 1) split8x3(raw[CUR][1][0], raw[CUR][1][1], af[1]);                                        \
                if (u == 3 * NT) split8x3(raw[CUR][2][0], raw[CUR][2][1], af[2]);                                            \
            }                                                                                                                \
            if (u == 2 * NT + 2 && !(p.whatif & 1)) load_raw(OTH, (KT_) + 2);                                                \
            mfma6(af[pos], b_cur, acc[pos][j]);                                                                              \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) b_cur[q] = b_nxt[q];                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                    \
    }
    for (int kt = 0; kt < p.KT; kt += 2) {
        DSD_WINO_TILE(kt, 0, 1, 0)
        DSD_WINO_TILE(kt + 1, WSTAGE, 0, 1)
    }
#undef DSD_WINO_TILE

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
    __shared__ __attribute__((aligned(1024))) unsigned char Bs[2 * WSTAGE];
    int L = blockIdx.x;
    {
        const int cpx = (int)gridDim.x >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const bool narrow = (L % p.tiles_n) == p.tiles_n - 1 && (p.Cout % WBROWS) != 0;   // uniform: last N tile of a Cout = 64 (mod 128) layer
    if (narrow)
        wino_body<2>(p, Bs);
    else
        wino_body<WNT>(p, Bs);
}

// w (OHWI fp32: [Cout][3][3][Cin]) -> transformed, split and packed:
//   out[tile_n][kt = cc*3 + kh][piece][pos][row][16]  bf16,   co = tile_n*128 + row,  ci = cc*16 + k
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
        const int kt = cc * 3 + kh;
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

// 3x3, stride 1, symmetric padding, no folded upsample, even output width, Cin % 16 == 0, Cout % 32 == 0, bf16x6 arithmetic,
// operands addressable with 32-bit byte offsets, and enough tiles to fill the chip (small grids keep the split-K kernels)
bool conv2d_wino_eligible(const ConvArgs& a) { return a.w_wino != nullptr && conv2d_wino_shape_ok(a); }

bool conv2d_wino_shape_ok(const ConvArgs& a) {
    static const bool off = getenv("DSD_NO_WINOGRAD") != nullptr;   // experiments only
    if (off || a.ks != 3 || a.stride != 1 || a.ups || a.pad_lo >= 0 || a.pad_total >= 0 || a.out_nchw) return false;
    if (a.precision != PREC_BF16X6) return false;
    if (a.W % 2 != 0 || a.Cin % 32 != 0 || (a.Cout % WBROWS != 0 && a.Cout % WBROWS != 64)) return false;
    const int64_t x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    const int64_t xb = ((int64_t)(a.N - 1) * x_bs + (int64_t)a.H * a.W * a.Cin) * 4;
    if (xb >= 0xFFFFFF00ll) return false;
    const int64_t M = (int64_t)a.N * a.H * a.W;
    return M >= 4096 && M < (1ll << 31);
}

int conv2d_wino_stats_chunks(const ConvArgs& a) {
    const int ohw = a.H * a.W;
    return ohw % (2 * WTILES) == 0 ? ohw / (2 * WTILES) : 0;
}

void conv2d_wino(const ConvArgs& a, hipStream_t s) {
    WinoP p{};
    p.x = a.x; p.wp = (const unsigned char*)a.w_wino; p.bias = a.bias; p.emb = a.emb; p.res = a.res; p.y = a.y;
    p.N = a.N; p.H = a.H; p.W = a.W; p.Cin = a.Cin; p.Cout = a.Cout; p.emb_stride = a.emb_stride;
    p.y_ld = a.y_ld > 0 ? a.y_ld : a.Cout;
    p.x_bs = a.x_bs >= 0 ? a.x_bs : (int64_t)a.H * a.W * a.Cin;
    p.ohw = a.H * a.W;
    p.M = (int)((int64_t)a.N * p.ohw);
    p.T = p.M / 2;
    p.halfW = a.W / 2;
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
