// gemm16.hip — single-product half-precision GEMM for the transformer backbone (DSD_PREC_F16 / DSD_PREC_BF16).
//
// BASELINE configs[4] names the arithmetic of the reference's DiT (UNet_DS_Diff/DiT_models.py:101-122 under fp16 autocast):
// every nn.Linear takes fp16 operands and accumulates in fp32.  This file is that arithmetic on the gfx950 matrix cores:
//     Y[m][n] = sum_k X[m][k] W[n][k] + bias[n]        X, W 16-bit (fp16 or bf16), fp32 accumulation,
// ONE v_mfma_f32_16x16x32_{f16,bf16} per 16x16x32 block of products (the split modes of conv_split.hip issue 3 or 6).
//
// Tile 256 (n) x 256 (m) x 64 (k), 512 threads: 8 waves as 2 (n) x 4 (m), a wave owns 128 n x 64 m = 8 x 4 accumulator
// tiles.  The product is computed TRANSPOSED (MFMA A operand = W rows, B operand = X rows), so a lane ends up with FOUR
// CONSECUTIVE n of one row m: the epilogue stores 8 bytes (16-bit output) or a float4 read-modify-write (gated residual) per
// lane and tile instead of 2-byte scatters.
// Operands reach LDS by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, no VGPR round trip), two 64 KB stages; a DMA piece
// is lane-linear in LDS, so the bank swizzle sits on the SOURCE address (cdna_hip_programming.md rule 21): tile rows are
// 128 B (64 k), 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7), and the fragment reads (ds_read_b128, lane =
// row, 8 consecutive k) apply the same XOR — conflict-free for the 16x16x32 operand pattern (checked lane group by lane
// group against MI355X_MICROARCH.md §LDS).  The DMA uses buffer addressing: rows beyond M / N and chunks beyond K get an
// out-of-range offset (zeros), and the k-tile advances through the scalar offset — no vector address work in the loop.
// Loop: stage tile t+1, multiply tile t, one barrier (+ vmcnt(0)) per k-tile; the fragments of the tile's second k-step are
// read while the first one multiplies.
//
// Fused epilogues (what DiTBlock.forward does around each Linear):
//   EPI_STORE   y16 = round16(acc + bias)             columns < qcols first multiplied by qscale (the attention's q scale)
//   EPI_GELU    y16 = round16(gelu_tanh(round16(acc + bias)))                                   (Mlp: fc1 -> GELU)
//   EPI_GATED   x32[m][n] += gate[m / T][n] * round16(acc + bias)                               (x + gate * f(...), fp32 stream)
#include <cstdlib>

#include "kernels.h"

namespace dsd {

namespace g16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 256, GBN = 256, GBK = 64;
constexpr int G_TILE_BYTES = GBN * GBK * 2;   // 32 KB per operand and stage
constexpr int G_EROW = 128 * 2 + 16;          // epilogue staging: a wave's 128 n of one output row + 16 B (bank spread, 16-B aligned)

template <typename T>
struct Frag;
template <>
struct Frag<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Frag<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// F.gelu(approximate="tanh") = 0.5 v (1 + tanh(u)), u = sqrt(2/pi) (v + 0.044715 v^3); 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)):
// one v_exp_f32 and one v_rcp_f32 (the result is rounded to 16 bits right after)
__device__ __forceinline__ float gelu_tanh_f(float v) {
    const float u2 = -2.f * 0.7978845608028654f * 1.4426950408889634f * (v + 0.044715f * v * v * v);   // -2u log2(e)
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u2));
}

// 16 B per lane from a buffer resource straight into LDS (lane-linear from the wave-uniform `lds`).  A __device__ function of
// its own: the target builtin directly inside a __global__ template makes the HOST pass drop the instantiation silently.
static __device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds, 16, voff, soff, 0, 0);
}

struct G16P {
    const void* x;      // [M][ldx] 16-bit
    const void* w;      // [N][K] 16-bit
    const float* bias;  // [N] or null
    void* y16;          // [M][ldy] 16-bit (EPI_STORE / EPI_GELU)
    float* x32;         // [M][ldx32] fp32 (EPI_GATED)
    const float* gate;  // [M / T][gate_stride]
    int M, N, K, ldx, ldy, ldx32, gate_stride, T, qcols;
    float qscale;
    int tiles_m, tiles_n;
};

// WI: what-if bits of the diagnostic instantiations (dsd_bench_gemm_half; results are then garbage): 1 no LDS-DMA staging in the
// loop, 2 fragments read from LDS once (not per k-tile), 4 no epilogue, 8 no barrier, 32 the DMA is issued but never waited for (barrier only),
// 16 the output of every m tile stored over tile 0's rows (stays in L2)
// M32: the same tile on v_mfma_f32_32x32x16 (a wave's 128 n x 64 m = 4 x 2 accumulator tiles of 32 x 32, four k-steps of 16 per
// k-tile).  Same matrix-pipe cycles; an MFMA of this shape holds the wave's issue port for 8 of its 32 cycles instead of 8 of 16,
// which leaves the DMA / ds_read / address instructions of the two waves of a SIMD twice the issue slots.
// LD4: only waves 0-3 — one per SIMD (a workgroup's waves w and w + 4 share a SIMD) — issue the LDS-DMA, each for itself and for
// its SIMD-mate.  What-if table, round 3: with the DMA flowing but never waited for and the fragments kept in registers the loop
// still runs 1.44x the time of the bare MFMA loop — a DMA instruction holds its wave's in-order issue until the texture
// addresser has taken its 64 lanes (16 cycles each, 64 instructions per k-tile and CU), and with all eight waves issuing their
// pieces right behind the barrier both waves of every SIMD are stuck at the same time.  With one loader per SIMD its mate
// keeps the matrix pipe busy meanwhile.
template <typename T16, int EPI, int WI = 0, bool M32 = false, int LD4 = 0>   // LD4: 1 = loaders issue in a burst, 2 = one piece per 4 MFMAs
__global__ __launch_bounds__(512, 2) void gemm16_kernel(G16P p) {
    using F = Frag<T16>;
    // [stage][W tile | X tile]; after the k-loop the same memory transposes the output tile (8 waves x 64 rows x 272 B)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[(4 * G_TILE_BYTES > 8 * 64 * G_EROW) ? 4 * G_TILE_BYTES : 8 * 64 * G_EROW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;

    // XCD-aware tile order: workgroups b and b + 8 share an XCD (and its L2); give each XCD a contiguous run of tiles, the n
    // tiles of one m tile next to each other, so the X rows of an m tile are fetched from HBM once per XCD
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
    const int bm0 = tile_m * GBM, bn0 = tile_n * GBN;

    // ---- staging: per k-tile 4 + 4 LDS-DMA pieces per thread.  Piece j of wave w covers tile rows 64 j + 8 w .. + 8; lane ->
    // row + (lane >> 3), physical chunk lane & 7, logical chunk (lane & 7) ^ ((row >> 1) & 7).  Buffer addressing: the
    // per-lane byte offset of a piece is fixed for the whole k-loop (rows beyond M / N get an out-of-range offset: the
    // hardware delivers zeros), the k-tile advances through the SCALAR offset — no vector address arithmetic in the loop.
    constexpr int NV = LD4 != 0 ? 2 : 1;                                  // virtual waves staged by this wave (LD4: itself and wave + 4)
    const int rows_w = min(p.N - bn0, GBN), rows_x = min(p.M - bm0, GBM);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const T16*>(p.w) + (int64_t)bn0 * p.K), 0, (unsigned)rows_w * (unsigned)p.K * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const T16*>(p.x) + (int64_t)bm0 * p.ldx), 0,
        ((unsigned)(rows_x - 1) * (unsigned)p.ldx + (unsigned)p.K) * 2u, 0x00020000);
    unsigned wvo[NV][4], xvo[NV][4];
    int schunk[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int srow = (wave + 4 * v) * 8 + (lane >> 3);            // + 64 j
        schunk[v] = (lane & 7) ^ ((srow >> 1) & 7);                   // ((64 j + srow) >> 1) & 7 == (srow >> 1) & 7
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = j * 64 + srow;
            wvo[v][j] = r < rows_w ? ((unsigned)r * (unsigned)p.K + (unsigned)schunk[v] * 8u) * 2u : 0xFFFFFFF0u;
            xvo[v][j] = r < rows_x ? ((unsigned)r * (unsigned)p.ldx + (unsigned)schunk[v] * 8u) * 2u : 0xFFFFFFF0u;
        }
    }
    // one DMA piece (q = 0..3: W rows 64 q .., q = 4..7: X rows) of virtual wave v of the k-tile at k0 into stage `buf`
    auto stage_piece = [&](int buf, int k0, int q, int v, bool tail) {
        unsigned char* base = lds + buf * (2 * G_TILE_BYTES) + (wave + 4 * v) * 1024;
        const int so = __builtin_amdgcn_readfirstlane(k0 * 2);
        // last, partial k-tile: chunks beyond K are zeros (a row's tail would otherwise read its neighbour's head)
        const bool kok = !tail || k0 + schunk[v] * 8 < p.K;
        if (q < 4) dma16(rw, base + q * 8192, kok ? wvo[v][q] : 0xFFFFFFF0u, so);
        else dma16(rxb, base + G_TILE_BYTES + (q - 4) * 8192, kok ? xvo[v][q - 4] : 0xFFFFFFF0u, so);
    };
    auto stage = [&](int buf, int k0) {
        if (LD4 != 0 && wave >= 4) return;                                 // (wave-uniform)
        const bool tail = k0 + GBK > p.K;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int q = 0; q < 8; ++q) stage_piece(buf, k0, q, v, tail);
    };

    // ---- fragment read offsets: lane -> row (lane & 15) of a 16-row tile, k chunk 4 ks + (lane >> 4), XOR ((row >> 1) & 7)
    const int frow = lane & 15;
    const int foff0 = frow * 128 + (((lane >> 4) ^ ((frow >> 1) & 7)) << 4);   // ks = 0; ks = 1: ^ 64
    const int aoff = wn * (128 * 128) + foff0;                                 // W tile: rows wn * 128 + 16 i
    const int boff = G_TILE_BYTES + wm * (64 * 128) + foff0;                   // X tile: rows wm * 64 + 16 j

    if constexpr (M32) {
        // ---- the 32x32x16 build: fragment = row (lane & 31) of a 32-row tile, 16-byte chunk 2 ks + (lane >> 5), ks = 0..3; the
        // same XOR ((row >> 1) & 7): 16 consecutive lanes = 16 consecutive rows of one chunk = 16 different slots of the bank row
        const int frow32 = lane & 31, half = lane >> 5;
        const int fsw = (frow32 >> 1) & 7;
        const int abase = wn * (128 * 128) + frow32 * 128;                  // + i * 4096 (32 rows)
        const int bbase = G_TILE_BYTES + wm * (64 * 128) + frow32 * 128;    // + j * 4096
        f32x16 c32[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) c32[i][j][r] = 0.f;
        const int nkt32 = (p.K + GBK - 1) / GBK;
        stage(0, 0);
        __syncthreads();
        typename F::v8 a32[2][4], b32[2][2];
        auto read32 = [&](const unsigned char* sb, int ks, typename F::v8 (&a)[4], typename F::v8 (&b)[2]) {
            const int ch = ((2 * ks + half) ^ fsw) << 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const typename F::v8*>(sb + bbase + j * 4096 + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const typename F::v8*>(sb + abase + i * 4096 + ch);
        };
        for (int t = 0; t < nkt32; ++t) {
            const int buf = t & 1;
            const unsigned char* sb = lds + buf * (2 * G_TILE_BYTES);
            read32(sb, 0, a32[0], b32[0]);
            if (t + 1 < nkt32) stage(buf ^ 1, (t + 1) * GBK);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) read32(sb, ks + 1, a32[(ks + 1) & 1], b32[(ks + 1) & 1]);   // the next k-step's fragments are on their way
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) c32[i][j] = F::mfma32(a32[ks & 1][i], b32[ks & 1][j], c32[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
        }
        // ---- epilogue.  c32[i][j][r]: n = bn0 + wn*128 + 32 i + 8 (r >> 2) + 4 (lane >> 5) + (r & 3), m = bm0 + wm*64 + 32 j + (lane & 31):
        // again four consecutive n of one row per register quad
        const int nl32 = bn0 + wn * 128 + half * 4;     // + 32 i + 8 g
        const int ml32 = bm0 + wm * 64 + frow32;        // + 32 j
        const int rows_left32 = min(p.M - bm0, GBM);
        if (EPI == 2) {
            float* xb = p.x32 + (int64_t)bm0 * p.ldx32;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)rows_left32 * (unsigned)p.ldx32 * 4u, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = ml32 + j * 32;
                const int mc = min(m, p.M - 1);
                const float* gp = p.gate + (int64_t)(mc / p.T) * p.gate_stride;
                const unsigned rowoff = (unsigned)(m - bm0) * (unsigned)p.ldx32 * 4u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 xv[4], gv[4], bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nl32 + i * 32 + g * 8;
                        const unsigned off = (m < p.M && n < p.N) ? rowoff + (unsigned)n * 4u : 0xFFFFFFF0u;
                        xv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
                        gv[g] = *reinterpret_cast<const f32x4*>(gp + min(n, p.N - 4));
                        bq[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + min(n, p.N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nl32 + i * 32 + g * 8;
                        const unsigned off = (m < p.M && n < p.N) ? rowoff + (unsigned)n * 4u : 0xFFFFFFF0u;
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = fmaf(gv[g][e], (float)(T16)(c32[i][j][4 * g + e] + bq[g][e]), xv[g][e]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), rx, off, 0, 0);
                    }
                }
            }
        } else {
            unsigned char* wb = lds + wave * (64 * G_EROW);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nl32 + i * 32 + g * 8;
                    const f32x4 bq = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + min(n, p.N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = c32[i][j][4 * g + e] + bq[e];
                        if (EPI == 0 && n < p.qcols) v *= p.qscale;
                        typename F::v4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (T16)v[e];
                        if (EPI == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = (T16)gelu_tanh_f((float)h[e]);
                        }
                        *reinterpret_cast<typename F::v4*>(wb + (j * 32 + frow32) * G_EROW + (i * 32 + g * 8 + half * 4) * 2) = h;
                    }
                }
            }
            T16* yb = reinterpret_cast<T16*>(p.y16) + (int64_t)bm0 * p.ldy;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (unsigned)rows_left32 * (unsigned)p.ldy * 2u, 0x00020000);
            const int nc = bn0 + wn * 128 + (lane & 15) * 8;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int m = bm0 + wm * 64 + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(wb + row * G_EROW + (lane & 15) * 16);
                const unsigned off = (m < p.M && nc < p.N) ? (unsigned)(m - bm0) * (unsigned)p.ldy * 2u + (unsigned)nc * 2u : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, ry, off, 0, 0);
            }
        }
        return;
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (p.K + GBK - 1) / GBK;
    stage(0, 0);
    __syncthreads();   // (emits vmcnt(0): the first tile has landed)
    typename F::v8 af[2][8], bf[2][4];
    auto read_frags = [&](const unsigned char* sb, int ks, typename F::v8 (&a)[8], typename F::v8 (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const typename F::v8*>(sb + ((boff + j * 2048) ^ (ks * 64)));
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const typename F::v8*>(sb + ((aoff + i * 2048) ^ (ks * 64)));
    };
    // (Measured and dropped, round 3: the 8 DMA pieces of the next k-tile spread over the MFMA stream, one per group of 4 MFMAs,
    // waves 0-3 during the first k-step and waves 4-7 during the second — 740 instead of 825 TF/s: with vmcnt(0) in front of
    // the tile's barrier a late piece is a late tile.  A spread issue needs the counted-vmcnt, three-stage structure of
    // cdna_hip_programming.md's 8-phase template, which half-tile LDS stages would have to make room for.)
    for (int t = 0; t < nkt; ++t) {
        const int buf = t & 1;
        const unsigned char* sb = lds + buf * (2 * G_TILE_BYTES);
        if (!(WI & 2) || t == 0) read_frags(sb, 0, af[0], bf[0]);
        if (t + 1 < nkt && !(WI & 1) && LD4 != 2) stage(buf ^ 1, (t + 1) * GBK);
        if (!(WI & 2) || t == 0) read_frags(sb, 1, af[1], bf[1]);   // the second k-step's fragments are on their way while the first one multiplies
        const bool spread = LD4 == 2 && wave < 4 && t + 1 < nkt;    // (wave-uniform)
        const bool stail = (t + 1) * GBK + GBK > p.K;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = F::mfma(af[ks][i], bf[ks][j], acc[i][j]);
                if (LD4 == 2) {
                    if (spread) stage_piece(buf ^ 1, (t + 1) * GBK, i, ks, stail);   // piece i of virtual wave ks: 16 pieces over the tile's 16 MFMA groups
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (WI & 32) {   // (diagnostic) barrier WITHOUT the wait for the DMA: what the exposed latency of the next tile costs
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
            __builtin_amdgcn_s_barrier();
        } else if (!(WI & 8)) {
            __syncthreads();   // vmcnt(0) + barrier: tile t+1 has landed, every wave is done reading tile t
        }
    }
    if (WI & 4) {   // (diagnostic) keep the accumulators alive without the stores
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.y16)[0] = t;
        return;
    }

    // ---- epilogue.  acc[i][j][r]: n = bn0 + wn*128 + 16 i + 4 (lane >> 4) + r, m = bm0 + wm*64 + 16 j + (lane & 15).
    // No lane-dependent branch: loads go to clamped addresses, stores are buffer stores whose offset is pushed out of range
    // for rows >= M / columns >= N (the hardware drops them) — a store under a branch costs a vmcnt(0) round trip per tile.
    const int nl = bn0 + wn * 128 + (lane >> 4) * 4;
    const int ml = bm0 + wm * 64 + (lane & 15);
    f32x4 bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = min(nl + i * 16, p.N - 4);
        bv[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int rows_left = min(p.M - bm0, GBM);
    if (EPI == 2) {
        float* xb = p.x32 + (int64_t)bm0 * p.ldx32;   // descriptor over this tile's rows only
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (unsigned)rows_left * (unsigned)p.ldx32 * 4u, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = ml + j * 16;
            const int mc = min(m, p.M - 1);
            const float* gp = p.gate + (int64_t)(mc / p.T) * p.gate_stride;
            const unsigned rowoff = (unsigned)(m - bm0) * (unsigned)p.ldx32 * 4u;
            f32x4 xv[8], gv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = nl + i * 16;
                const unsigned off = (m < p.M && n < p.N) ? rowoff + (unsigned)n * 4u : 0xFFFFFFF0u;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));   // out of range -> 0
                gv[i] = *reinterpret_cast<const f32x4*>(gp + min(n, p.N - 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = nl + i * 16;
                const unsigned off = (m < p.M && n < p.N) ? rowoff + (unsigned)n * 4u : 0xFFFFFFF0u;
                const f32x4 v = acc[i][j] + bv[i];
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaf(gv[i][e], (float)(T16)v[e], xv[i][e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), rx, off, 0, 0);
            }
        }
    } else {
        // 16-bit output.  A lane holds 4 consecutive n (8 B) of 16 different rows: stored as they stand, every store instruction
        // would touch 16 rows with 32 contiguous bytes each (what-if table, round 3: the epilogue was 35 % of a K = 768 launch).
        // The wave's 128 n x 64 m sub-tile goes through LDS instead (the stages are idle now; every wave has passed the loop's
        // last barrier): written [m][n] with ds_write_b64, read back 16 B per lane with 16 lanes along a row, so that a store
        // instruction writes 4 rows x 256 contiguous bytes.  LDS operations of one wave execute in order: no barrier.
        unsigned char* wb = lds + wave * (64 * G_EROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = nl + i * 16;
                f32x4 v = acc[i][j] + bv[i];
                if (EPI == 0 && n < p.qcols) v *= p.qscale;
                typename F::v4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (T16)v[e];
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (T16)gelu_tanh_f((float)h[e]);
                }
                *reinterpret_cast<typename F::v4*>(wb + (j * 16 + (lane & 15)) * G_EROW + (i * 16 + (lane >> 4) * 4) * 2) = h;
            }
        }
        // (what-if 16: every m tile stores into the rows of tile 0 — the same instruction stream, but the output stays in L2)
        T16* yb = reinterpret_cast<T16*>(p.y16) + ((WI & 16) ? (int64_t)0 : (int64_t)bm0 * p.ldy);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (unsigned)rows_left * (unsigned)p.ldy * 2u, 0x00020000);
        const int nc = bn0 + wn * 128 + (lane & 15) * 8;          // this lane's 8 columns of a row (N % 8 == 0: valid together)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4);
            const int m = bm0 + wm * 64 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wb + row * G_EROW + (lane & 15) * 16);
            const unsigned off = (m < p.M && nc < p.N) ? (unsigned)(m - bm0) * (unsigned)p.ldy * 2u + (unsigned)nc * 2u : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(v, ry, off, 0, 0);
        }
    }
}

template <typename T16>
__global__ void cast16_kernel(const float* __restrict__ x, int64_t n, T16* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (T16)x[i];
}
template <typename T16>
__global__ void uncast16_kernel(const T16* __restrict__ x, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (float)x[i];
}

// LayerNorm (no affine) + adaLN modulate of one token row per wave, 16-bit output; the row is read once (kept in registers)
template <typename T16, int NV>   // NV float4 per lane: C <= 256 NV
__global__ __launch_bounds__(256) void ln_modulate16_kernel(const float* __restrict__ x, int64_t rows, int T, int C,
                                                            const float* __restrict__ mod, int mod_stride, int shift_off,
                                                            int scale_off, float eps, T16* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    const float* mr = mod + (row / T) * mod_stride;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < C ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q = fmaf(d, d, q);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < C) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(mr + scale_off + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(mr + shift_off + c);
            typename Frag<T16>::v4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (T16)(((v[i][r] - mean) * rstd) * (1.f + sc[r]) + sh[r]);
            *reinterpret_cast<typename Frag<T16>::v4*>(y + row * C + c) = h;
        }
    }
}

template <typename T16, int EPI>
void launch_gemm16(const G16P& p, hipStream_t s) {
    static const bool m32 = getenv("DSD_GEMM16_M32") != nullptr;   // A/B: the 32x32x16 build
    static const bool ld4 = getenv("DSD_GEMM16_LD4") != nullptr;   // A/B: one loader wave per SIMD
    if (ld4)
        hipLaunchKernelGGL((gemm16_kernel<T16, EPI, 0, false, 1>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), 0, s, p);
    else if (m32)
        hipLaunchKernelGGL((gemm16_kernel<T16, EPI, 0, true>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL((gemm16_kernel<T16, EPI>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), 0, s, p);
}
void launch_gemm16_whatif(const G16P& p, int wi, hipStream_t s) {
    const dim3 g((unsigned)(p.tiles_m * p.tiles_n)), b(512);
    switch (wi) {
        case 0: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 0>), g, b, 0, s, p); break;
        case 1: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 1>), g, b, 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 2>), g, b, 0, s, p); break;
        case 3: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 3>), g, b, 0, s, p); break;
        case 4: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 4>), g, b, 0, s, p); break;
        case 7: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 7>), g, b, 0, s, p); break;
        case 15: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 15>), g, b, 0, s, p); break;
        case 16: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 16>), g, b, 0, s, p); break;
        case 32: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 32>), g, b, 0, s, p); break;
        case 34: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 34>), g, b, 0, s, p); break;
        case 64: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 0, true>), g, b, 0, s, p); break;   // the 32x32x16 build
        case 128: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 0, false, 1>), g, b, 0, s, p); break;   // one loader wave per SIMD
        case 129: hipLaunchKernelGGL((gemm16_kernel<_Float16, 0, 0, false, 2>), g, b, 0, s, p); break;   // ... its pieces spread over the MFMA stream
        default: fail("gemm16 what-if %d is not instantiated (0, 1, 2, 3, 4, 7, 15, 16, 32)", wi);
    }
}

}  // namespace g16
using namespace g16;

bool gemm16_shape_ok(int M, int N, int K) { return M >= 1 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0; }

void gemm16(const Gemm16Args& a, hipStream_t s) { gemm16_whatif(a, -1, s); }

void gemm16_whatif(const Gemm16Args& a, int whatif, hipStream_t s) {
    DSD_CHECK(gemm16_shape_ok(a.M, a.N, a.K), "gemm16: M=%d N=%d K=%d unsupported (N %% 8, K %% 8)", a.M, a.N, a.K);
    DSD_CHECK(a.ldx % 8 == 0 && a.ldx >= a.K, "gemm16: ldx=%d must be a multiple of 8 and >= K", a.ldx);
    G16P p{};
    p.x = a.x; p.w = a.w; p.bias = a.bias; p.y16 = a.y16; p.x32 = a.x32; p.gate = a.gate;
    p.M = a.M; p.N = a.N; p.K = a.K; p.ldx = a.ldx; p.ldy = a.ldy; p.ldx32 = a.ldx32; p.gate_stride = a.gate_stride;
    p.T = a.T > 0 ? a.T : 1;
    p.qcols = a.qcols; p.qscale = a.qscale;
    p.tiles_m = cdiv(a.M, GBM);
    p.tiles_n = cdiv(a.N, GBN);
    if (a.epi == EPI16_GATED) {
        DSD_CHECK(a.x32 && a.gate && a.ldx32 % 4 == 0 && a.gate_stride % 4 == 0, "gemm16: gated epilogue needs x32 / gate (strides %% 4)");
    } else {
        DSD_CHECK(a.y16 && a.ldy % 8 == 0, "gemm16: 16-bit output missing or ldy %% 8 != 0 (rows are stored in 16-byte pieces)");
    }
    if (whatif >= 0) {
        launch_gemm16_whatif(p, whatif, s);
        check_launch("gemm16_whatif");
        return;
    }
    if (a.bf16) {
        switch (a.epi) {
            case EPI16_STORE: launch_gemm16<__bf16, 0>(p, s); break;
            case EPI16_GELU: launch_gemm16<__bf16, 1>(p, s); break;
            default: launch_gemm16<__bf16, 2>(p, s); break;
        }
    } else {
        switch (a.epi) {
            case EPI16_STORE: launch_gemm16<_Float16, 0>(p, s); break;
            case EPI16_GELU: launch_gemm16<_Float16, 1>(p, s); break;
            default: launch_gemm16<_Float16, 2>(p, s); break;
        }
    }
    check_launch("gemm16");
}

void cast16(const float* x, int64_t n, void* y, int bf16, hipStream_t s) {
    if (!n) return;
    const dim3 g((unsigned)std::min<int64_t>((n + 255) / 256, 65535));
    if (bf16) hipLaunchKernelGGL(cast16_kernel<__bf16>, g, dim3(256), 0, s, x, n, (__bf16*)y);
    else hipLaunchKernelGGL(cast16_kernel<_Float16>, g, dim3(256), 0, s, x, n, (_Float16*)y);
    check_launch("cast16");
}
void uncast16(const void* x, int64_t n, float* y, int bf16, hipStream_t s) {
    if (!n) return;
    const dim3 g((unsigned)std::min<int64_t>((n + 255) / 256, 65535));
    if (bf16) hipLaunchKernelGGL(uncast16_kernel<__bf16>, g, dim3(256), 0, s, (const __bf16*)x, n, y);
    else hipLaunchKernelGGL(uncast16_kernel<_Float16>, g, dim3(256), 0, s, (const _Float16*)x, n, y);
    check_launch("uncast16");
}

void ln_modulate16(const float* x, int N, int T, int C, const float* mod, int mod_stride, int shift_off, int scale_off, float eps,
                   void* y, int bf16, hipStream_t s) {
    const int64_t rows = (int64_t)N * T;
    if (rows == 0) return;
    DSD_CHECK(C % 4 == 0 && C <= 2048 && mod_stride % 4 == 0 && shift_off % 4 == 0 && scale_off % 4 == 0,
              "ln_modulate16: C=%d must be a multiple of 4 and <= 2048 (strides %% 4)", C);
    const dim3 g(cdiv(rows, 4)), b(256);
    const int nv = cdiv(C, 256);
#define LN16(TT, NV) hipLaunchKernelGGL((ln_modulate16_kernel<TT, NV>), g, b, 0, s, x, rows, T, C, mod, mod_stride, shift_off, scale_off, eps, (TT*)y)
    if (bf16) {
        if (nv <= 1) LN16(__bf16, 1); else if (nv <= 2) LN16(__bf16, 2); else if (nv <= 3) LN16(__bf16, 3);
        else if (nv <= 4) LN16(__bf16, 4); else if (nv <= 5) LN16(__bf16, 5); else LN16(__bf16, 8);
    } else {
        if (nv <= 1) LN16(_Float16, 1); else if (nv <= 2) LN16(_Float16, 2); else if (nv <= 3) LN16(_Float16, 3);
        else if (nv <= 4) LN16(_Float16, 4); else if (nv <= 5) LN16(_Float16, 5); else LN16(_Float16, 8);
    }
#undef LN16
    check_launch("ln_modulate16");
}

}  // namespace dsd
