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
#include "kernels.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int SBM = 128;
static constexpr int SBK = 32;
static constexpr int RSB = 80;  // LDS row stride in bytes (64 B of bf16 + 16 B pad)
static constexpr unsigned OOB = 0xFFFFFFF0u;

struct SplitP {
    const float* x;
    const void* w;  // [NP][Cout][Ktot] bf16
    const float* bias;
    const float* emb;
    const float* res;
    float* y;
    int64_t x_bs;
    int N, H, W, Cin, Cout, OH, OW, ks, stride, pad, ups, emb_stride, out_nchw;
    int M, Ktot, cchunks, IHg, IWg, tiles_m, tiles_n, ohw;
    unsigned x_bytes, w_bytes, w_plane_bytes;
    int* ovf;   // f16x3 only: set to 1 when an operand magnitude exceeds the fp16 range (caller reports it)
    int y_ld;         // row stride of y (>= Cout)
    int ksplit;       // split-K: the k-tiles are divided over ksplit workgroups per output tile (A-direct 128-row kernel)
    float* partial;   // [ksplit][M][Cout] raw partial sums, reduced (+ bias / embedding / residual) by splitk_reduce_kernel
    double* stats;    // GroupNorm statistics of the output, [sample][chunk][Cout][2] (GnSrc layout), or nullptr
    int stats_chunks; // chunks per sample = ohw / block-tile rows (a block tile never straddles two samples then)
    long long* stamps;   // diagnostic instantiations only (DIAG > 0), else unused
    int diag;            // what-if bits of the diagnostic instantiation
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Piece element type: bf16 (8 significand bits, fp32's exponent range) or fp16 (11 bits, range +-65504).
template <bool F16> struct Elt;
template <> struct Elt<false> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        bf16x2 t;
        t[0] = (__bf16)a;
        t[1] = (__bf16)b;
        return __builtin_bit_cast(unsigned, t);
    }
    static __device__ __forceinline__ float lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
    static __device__ __forceinline__ float hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Elt<true> {
    typedef f16x8 v8;
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        f16x2 t;
        t[0] = (_Float16)a;
        t[1] = (_Float16)b;
        return __builtin_bit_cast(unsigned, t);
    }
    static __device__ __forceinline__ float lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
    static __device__ __forceinline__ float hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// split 4 fp32 into NP bf16 pieces each; out[p] = 4 packed bf16 (8 bytes)
template <int NP, bool F16 = false>
__device__ __forceinline__ void split4(f32x4 v, u32x2 (&out)[NP], int* ovf = nullptr) {
    float a = v.x, b = v.y, c = v.z, d = v.w;
    if (F16 && ovf) {   // fp16 pieces cannot hold |x| > 65504: flag it instead of silently producing inf
        const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
        if (!(m <= 65504.f)) *ovf = 1;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const unsigned p01 = Elt<F16>::pk(a, b), p23 = Elt<F16>::pk(c, d);
        out[p].x = p01;
        out[p].y = p23;
        if (p + 1 < NP) {
            a -= Elt<F16>::lo(p01);
            b -= Elt<F16>::hi(p01);
            c -= Elt<F16>::lo(p23);
            d -= Elt<F16>::hi(p23);
        }
    }
}

// ---- accumulators -> memory: bias + per-(sample,channel) embedding + residual (same fusion as the fp32 kernel)
// STATS: the GroupNorm statistics of the tensor being written.  Every stored value v of this lane's (up to) 16 rows of a
// column goes into SHIFTED fp32 partials  cs = sum(v - ref), cq = sum((v - ref)^2)  with ref = the lane's first value of
// that column (StatAcc); the caller turns them into plain fp64 (sum, sum of squares) per 32-row block (stat_flush) and
// everything above is fp64.  The shift matters: var = E[x^2] - mean^2 cancels, and unshifted fp32 partials would carry
// their 1e-7 relative error on E[x^2] (measured: 9e-5 on the network output for zero-variance groups, rstd = 316); shifted,
// the fp32 error is relative to the VARIANCE of 16 neighbouring values, and a constant channel gives exactly zero.
template <int NT>
struct StatAcc {
    float ref[NT], cs[NT], cq[NT];
    int rows;        // values per column added by this lane (the same for every column it owns at all)
    unsigned first;  // bit j: column j has no reference value yet (ragged tiles only; interior tiles know it statically)
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int j = 0; j < NT; ++j) ref[j] = cs[j] = cq[j] = 0.f;
        rows = 0;
        first = ~0u;
    }
    // first: compile-time "this is the lane's first value of column j" (the loops around the calls are fully unrolled)
    __device__ __forceinline__ void add(int j, float v, bool first) {
        if (first) ref[j] = v;
        const float d = v - ref[j];
        cs[j] += d;
        cq[j] = fmaf(d, d, cq[j]);
    }
    // plain (sum, sum of squares) of the values added so far, in fp64:  sum = S + n r,  sumsq = Q + 2 r S + n r^2
    // (a column the lane never stored to has ref = cs = cq = 0 and contributes 0 whatever `rows` says)
    __device__ __forceinline__ void flush(double (&ds)[NT], double (&dq)[NT]) {
        const double n = (double)rows;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const double r = (double)ref[j], S = (double)cs[j];
            ds[j] += S + n * r;
            dq[j] += (double)cq[j] + 2.0 * r * S + n * r * r;
        }
        clear();
    }
};

template <int NT, bool STATS = false>
__device__ __forceinline__ void split_epilogue(const SplitP& p, const f32x16 (&acc)[NT], int m0, int n0, int wave, int lrow,
                                               int half, int tile_rows, StatAcc<NT>& st, bool first_block = true) {
    constexpr int BROWS = NT * 32;
    const bool interior = (m0 + tile_rows <= p.M) && (n0 + BROWS <= p.Cout) && !p.out_nchw;
    if (interior) {
        float bj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bj[j] = p.bias ? p.bias[n0 + j * 32 + lrow] : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int mb = m0 + wave * 32 + 8 * g + 4 * half;
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
                if (STATS && g == 3 && rr == 3) st.rows += 16;
                float* yp = p.y + (int64_t)(mb + rr) * p.y_ld + n0 + lrow;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float v = acc[j][4 * g + rr] + bj[j];
                    if (p.emb) v += ev[rr][j];
                    if (p.res) v += rv[rr][j];
                    yp[j * 32] = v;
                    if (STATS) st.add(j, v, first_block && g == 0 && rr == 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.M) continue;
        if (STATS) st.rows += 1;
        const int nb = m / p.ohw;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 32 + lrow;
            if (n >= p.Cout) continue;
            float v = acc[j][r];
            if (p.bias) v += p.bias[n];
            if (p.emb) v += p.emb[(int64_t)nb * p.emb_stride + n];
            if (p.res) v += p.res[(int64_t)m * p.Cout + n];
            if (STATS) {   // ragged tile: only stored values reach this point; the shift is the first of them
                st.add(j, v, (st.first >> j) & 1u);
                st.first &= ~(1u << j);
            }
            if (p.out_nchw)
                p.y[((int64_t)nb * p.Cout + n) * p.ohw + (m - nb * p.ohw)] = v;
            else
                p.y[(int64_t)m * p.y_ld + n] = v;
        }
    }
}

template <int NT>
__device__ __forceinline__ void split_epilogue(const SplitP& p, const f32x16 (&acc)[NT], int m0, int n0, int wave, int lrow,
                                               int half, int tile_rows = SBM) {
    StatAcc<NT> st;
    split_epilogue<NT, false>(p, acc, m0, n0, wave, lrow, half, tile_rows, st);
}

// Column sums of one block tile -> p.stats.  Lanes l and l+32 hold the two row halves of a column, the four waves the row
// blocks: combined through the (now idle) LDS in a fixed order, so the result is deterministic.  One (sum, sumsq) pair per
// column and block tile; the tile lies inside one sample (host guarantees ohw % tile rows == 0).
template <int NT>
__device__ __forceinline__ void stats_reduce(const SplitP& p, const double (&cs)[NT], const double (&cq)[NT], unsigned char* lds,
                                             int m0, int n0, int tile_rows, int tid, int wave, int lrow, int half) {
    constexpr int BROWS = NT * 32;
    double* red = reinterpret_cast<double*>(lds);   // [4 waves][BROWS][2]
    __syncthreads();                                 // every wave is done with the operand tiles in LDS
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        double s = cs[j], q = cq[j];
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (half == 0) {
            red[((wave * BROWS) + j * 32 + lrow) * 2 + 0] = s;
            red[((wave * BROWS) + j * 32 + lrow) * 2 + 1] = q;
        }
    }
    __syncthreads();
    const int nb = m0 / p.ohw;
    const int chunk = (m0 - nb * p.ohw) / tile_rows;
    for (int c = tid; c < BROWS; c += 256) {
        const int n = n0 + c;
        if (n >= p.Cout) continue;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s += red[((w * BROWS) + c) * 2 + 0];
            q += red[((w * BROWS) + c) * 2 + 1];
        }
        double* o = p.stats + (((int64_t)nb * p.stats_chunks + chunk) * p.Cout + n) * 2;
        o[0] = s;
        o[1] = q;
    }
}

template <int NT, int NP, bool F16>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(SplitP p) {
    typedef typename Elt<F16>::v8 bf16x8;   // (name kept: 8 packed 16-bit pieces, bf16 or fp16)
    constexpr int BROWS = NT * 32;
    constexpr int A_PLANE = SBM * RSB, B_PLANE = BROWS * RSB;
    constexpr int NBL = (BROWS * 4 * NP + 255) / 256;  // 16-byte weight loads per thread per tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[NP * (A_PLANE + B_PLANE)];
    unsigned char* As = lds;
    unsigned char* Bs = lds + NP * A_PLANE;
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
    const int m0 = tile_m * SBM;
    const int n0 = tile_n * BROWS;

    // ---- A staging: thread -> (row = tid>>3 (+32 i), 4 consecutive k = 4*(tid&7)), fp32 in HBM, split on the fly
    const int col4 = tid & 7;
    // 16 consecutive lanes (one ds_write_b64 bank group) cover rows r and r+4 (see the B mapping below): conflict-free
    const int srow = ((tid >> 6) << 3) + (((tid >> 3) & 1) << 2) + ((tid >> 4) & 3);
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
    // ---- B staging: pre-split bf16 planes; 16-byte chunk q = tid + 256 i  ->  (piece, row, chunk of 8 k)
    unsigned b_voff[NBL];
    int b_lds[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int q = tid + 256 * i;
        const int piece = q / (BROWS * 4);
        const int rem = q - piece * (BROWS * 4);
        // 8 consecutive lanes (one ds_write_b128 bank group) write rows r and r+4: with the 80-byte row stride their two
        // 64-byte segments are 320 B = 64 (mod 128) apart, i.e. all 32 banks exactly once (rows r, r+1 overlap on 4 banks)
        const int ch = rem & 3;
        const int row = (rem >> 5) * 8 + ((rem >> 3) & 3) + 4 * ((rem >> 2) & 1);
        const int n = n0 + row;
        const bool ok = piece < NP && n < p.Cout;
        b_voff[i] = ok ? (unsigned)piece * p.w_plane_bytes + ((unsigned)n * (unsigned)p.Ktot) * 2u + (unsigned)(ch * 16) : OOB;
        b_lds[i] = piece < NP ? piece * B_PLANE + row * RSB + ch * 16 : -1;
    }

    f32x4 ra[4];
    u32x4 rb[NBL];
    u32x2 pa[4][NP];   // bf16 pieces of the staged A rows (split during the previous tile's MFMAs)
    auto load_tile = [&](int soff_a, int soff_b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[i], soff_a, 0));
#pragma unroll
        for (int i = 0; i < NBL; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff_b, 0);
    };
    auto split_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) split4<NP, F16>(ra[i], pa[i], p.ovf);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                *reinterpret_cast<u32x2*>(As + q * A_PLANE + (srow + 32 * i) * RSB + col4 * 8) = pa[i][q];
#pragma unroll
        for (int i = 0; i < NBL; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<u32x4*>(Bs + b_lds[i]) = rb[i];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int lrow = lane & 31;
    const int half = lane >> 5;
    const unsigned char* a_frag = As + (wave * 32 + lrow) * RSB + half * 16;
    const unsigned char* b_frag = Bs + lrow * RSB + half * 16;

    auto mfma_group = [&](const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16& c) {
        // smallest terms first
        if (NP == 3) {
            c = Elt<F16>::mfma(a[2], b[0], c);
            c = Elt<F16>::mfma(a[0], b[2], c);
            c = Elt<F16>::mfma(a[1], b[1], c);
        }
        c = Elt<F16>::mfma(a[1], b[0], c);
        c = Elt<F16>::mfma(a[0], b[1], c);
        c = Elt<F16>::mfma(a[0], b[0], c);
    };

    // one k-step (16 k) of the staged tile: fragments of group j+1 are read before the MFMAs of group j
    auto kstep = [&](int s) {
        bf16x8 a_cur[NP], b_cur[NP], b_nxt[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            a_cur[q] = *reinterpret_cast<const bf16x8*>(a_frag + q * A_PLANE + s * 32);
            b_cur[q] = *reinterpret_cast<const bf16x8*>(b_frag + q * B_PLANE + s * 32);
            b_nxt[q] = b_cur[q];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j + 1 < NT) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    b_nxt[q] = *reinterpret_cast<const bf16x8*>(b_frag + q * B_PLANE + (j + 1) * 32 * RSB + s * 32);
            }
            mfma_group(a_cur, b_cur, acc[j]);
#pragma unroll
            for (int q = 0; q < NP; ++q) b_cur[q] = b_nxt[q];
        }
    };

    const int KT = p.ks * p.ks * p.cchunks;
    int kh = 0, kw = 0, cc = 0, tap = 0;
    tap_offsets(0, 0);
    load_tile(0, 0);
    split_tile();
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const bool more = kt + 1 < KT;
        if (more) {
            if (++cc == p.cchunks) {
                cc = 0;
                ++tap;
                if (++kw == p.ks) {
                    kw = 0;
                    ++kh;
                }
                tap_offsets(kh, kw);
            }
            load_tile(cc * (SBK * 4), (tap * p.Cin + cc * SBK) * 2);
        }
        kstep(0);
        if (more) split_tile();   // VALU work of the NEXT tile, issued in the shadow of this tile's MFMAs
        kstep(1);
    }

    if (p.stats) {
        StatAcc<NT> st;
        double ds[NT], dq[NT];
        st.clear();
#pragma unroll
        for (int j = 0; j < NT; ++j) ds[j] = dq[j] = 0.0;
        split_epilogue<NT, true>(p, acc, m0, n0, wave, lrow, half, SBM, st);
        st.flush(ds, dq);
        stats_reduce<NT>(p, ds, dq, lds, m0, n0, SBM, tid, wave, lrow, half);
        return;
    }
    split_epilogue<NT>(p, acc, m0, n0, wave, lrow, half);
}

// ------------------------------------------------------------------------------------------------------------------
// "A-direct" variant (256 threads): the activation operand never touches LDS.  In the MFMA A layout a lane owns ONE row
// (output pixel) and 8 consecutive k, which is 32 contiguous bytes of the NHWC fp32 tensor — so every lane loads its own
// fragment straight from global memory (2 x 16 B per k-step), splits it into bf16 pieces in registers and feeds the MFMA.
// Only the weight tile (shared by the four waves) is staged through LDS.  This removes 24 KB of LDS writes and 24 KB of LDS
// reads per block tile (LDS was 42 % busy with 30 % of that in bank conflicts in the staged kernel, PMC).
template <int NP, bool F16>
__device__ __forceinline__ void split8(f32x4 lo, f32x4 hi, typename Elt<F16>::v8 (&out)[NP], int* ovf) {
    typedef typename Elt<F16>::v8 bf16x8;
    u32x2 pl[NP], ph[NP];
    split4<NP, F16>(lo, pl, ovf);
    split4<NP, F16>(hi, ph, ovf);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        u32x4 v;
        v.x = pl[q].x; v.y = pl[q].y; v.z = ph[q].x; v.w = ph[q].y;
        out[q] = __builtin_bit_cast(bf16x8, v);
    }
}

// (target builtins behind __device__ functions: see lds_dma16)
static __device__ __forceinline__ long long clock_core() { return (long long)__builtin_amdgcn_s_memtime(); }
static __device__ __forceinline__ long long clock_100mhz() { return (long long)__builtin_amdgcn_s_memrealtime(); }
static __device__ __forceinline__ void wait_all() { __builtin_amdgcn_s_waitcnt(0); }
// 16 B per lane from a buffer resource straight into LDS (lane-linear from the wave-uniform `lds`).  A __device__ function of
// its own: the target builtin directly inside a __global__ template makes the HOST pass drop the instantiation silently.
static __device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds, 16, voff, soff, 0, 0);
}

// RB = 32-row blocks per wave.  RB = 1: block tile 128 x 32*NT, two workgroups per CU.  RB = 2: block tile 256 x 32*NT,
// ONE workgroup per CU (accumulators alone are 2*NT*16 registers): every weight fragment read from LDS feeds two MFMA
// groups, which halves the LDS traffic per MFMA, and a k-tile carries twice the MFMAs per barrier.
// DMA = the weight tile goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) instead of through registers: no staging
// registers (32) and no ds_write_b128 (8 per thread and tile) in the MFMA stream.  The LDS image is then lane-linear — rows
// of 64 B without padding — and the bank-conflict-free order is an XOR swizzle of the four 16-byte chunks of a row with
// (row >> 2) & 3, applied on the source address and on the fragment reads (as in conv_wino.hip).
// DIAG > 0 = diagnostic instantiations (never on the product path): clock stamps around prologue / k-loop / epilogue, and
// with what-if bits DIAG - 1 one cost of the loop removed (results are then garbage): 1 no activation split (raw bits),
// 2 no activation loads, 4 no weight loads / LDS writes, 8 no barrier, 16 no weight fragment reads from LDS.
// TR ("tap reuse", 3x3 stride-1 layers on the 256-row tile): the activations of ONE filter row — the tile's 256 pixels of
// input row oh + kh - 1, 32 channels, fp32 — are staged in LDS once (global -> registers -> LDS, like the weights, a whole
// tile ahead) and serve the three taps kw = 0, 1, 2, which differ only by a one-pixel shift of the LDS read address (a lane
// whose shifted pixel falls off its image row reads a row of zeros).  Per k-tile and wave that is 2.7 instead of 8
// activation load instructions (each costs ~50 cycles of the wave's only issue stream, what-if table above) and a third
// of the activation bytes from L2; lanes fetch their 16-byte pieces from LDS one unit ahead of the split that consumes
// them (128-byte rows, the eight 16-byte chunks XOR-swizzled with (row >> 1) & 7: conflict-free reads and writes).
// Measured (16x256x256x320->320 and the other large layers, same box): 230.6 vs 226.1, 237.5 vs 233.0, 231.0 vs 226.8 TF/s;
// whole step 435.1 vs 441.1 ms (-1.4 %).  The cycle count per k-tile stays (4757 vs 4766) — the issue slots of the saved
// loads go to the extra LDS traffic — the gain is the clock (1.78 vs 1.755 GHz: fewer bytes from L2).  What-ifs on THIS
// kernel (profiles/r02_conv_stamps_tr_whatif.txt): no weight loads / LDS writes 4383 cycles, -7.5 % time at the same clock;
// no barrier 4625 cycles but 1.745 GHz, -1 %; no weight fragment reads 4733 cycles at 1.82 GHz, -2 %.
// Two things that did NOT work on the way: staging by LDS-DMA (the stage then needs padding entries, and every barrier
// waits for the DMA: +15 % cycles, profiles/r02_conv_stamps_tr_dma.txt), and uniform branches ("first tile of a filter
// row?") inside the unit loop — the IR then has several blocks per tile and the compiler sinks the whole split into the last
// one (5564 cycles); the loop is therefore unrolled by three with the tap column a compile-time constant.
template <int NT, int NP, bool F16, int RB, bool DMA = false, int DIAG = 0, bool TR = false>
__global__ __launch_bounds__(256, RB == 1 ? 2 : 1) void conv_split_ad_kernel(SplitP p) {
    constexpr bool STAMP = DIAG > 0;
    constexpr int WI = DIAG > 0 ? DIAG - 1 : 0;
    // vector-memory loads issued one at a time, each right after the register it refills has been consumed, instead of in
    // bursts of eight (stamps: 4946 -> 4763 cycles per k-tile, k-loop 246.6 -> 243.3 us; what-if bit 32 = the burst schedule)
    constexpr bool SPREAD = RB == 2 && !DMA && !(WI & 32);
    int soff_bs = 0, soff_as = 0;
    // (what-if bit 128) weight-tile loads non-temporal: a tile is read once per CU and need not displace the activation
    // lines, which the next filter tap re-reads, from the 32 KB L1
    constexpr int B_AUX = (WI & 128) ? 2 : 0;
    // (what-if bit 256) VERDICT r1 item 5a, the convolution's side of it: activations arrive PRE-SPLIT as three bf16 planes
    // (6 B per element) — per tile and lane 12 loads of 16 B instead of 8 and no split VALU in the loop.  Timing only: the
    // planes are faked inside the fp32 tensor (same pixel stride in elements, a third of the buffer apart).
    constexpr bool PRESPLIT = (WI & 256) != 0;
    const unsigned pplane = (p.x_bytes / 4u) & ~255u;
    u32x4 rp[RB][6];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < 6; ++q) rp[r][q] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    auto stamp = [&](int i) {
        if (STAMP && threadIdx.x == 0) {
            p.stamps[(size_t)blockIdx.x * 8 + 2 * i] = clock_core();
            p.stamps[(size_t)blockIdx.x * 8 + 2 * i + 1] = clock_100mhz();
        }
    };
    stamp(0);
    typedef typename Elt<F16>::v8 bf16x8;
    constexpr int BROWS = NT * 32;
    constexpr int RS = DMA ? 64 : RSB;              // LDS row stride of a weight plane
    constexpr int B_PLANE = BROWS * RS;
    constexpr int NBL = (BROWS * 4 * NP + 255) / 256;
    // Two LDS stages for the weight tile: tile kt+1 is written into stage (kt+1)&1 while tile kt is read from the other
    // one, so ONE barrier per k-tile is enough (stage (kt+1)&1 was last read for tile kt-1, and every wave has passed the
    // barrier of tile kt since).  The last 16 B x 256 are per-thread dummy slots: the threads of the final, partly filled
    // staging round write there instead of branching on the exec mask.
    constexpr int STAGE = NP * B_PLANE;
    constexpr int A_STAGE = 257 * 128;               // one filter row of the tile: [pixel][32 channels fp32] + a row of zeros
    constexpr int A_OFF = 2 * STAGE + 256 * 16;
    static_assert(!TR || (RB == 2 && !DMA && NT == 5), "tap reuse is built for the 256 x 160 tile with the register weight path");
    static_assert((A_OFF + (TR ? 2 * A_STAGE : 0)) * (RB == 1 ? 2 : 1) <= 160 * 1024, "the stages of the resident workgroups must fit the LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char Bs[A_OFF + (TR ? 2 * A_STAGE : 0)];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 31;
    const int half = lane >> 5;
    const int nwg = gridDim.x;
    int L = blockIdx.x;
    {
        const int cpx = nwg >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const int ntiles = p.tiles_m * p.tiles_n;
    const int chunk = L / ntiles;   // split-K chunk of this workgroup (0 when ksplit == 1)
    L -= chunk * ntiles;
    const int tile_n = L % p.tiles_n;
    const int tile_m = L / p.tiles_n;
    const int m0 = tile_m * (SBM * RB);
    const int n0 = tile_n * BROWS;

    // ---- A: this lane's RB output pixels (rows of the implicit GEMM) and its 8-float slot inside a 16-k step
    int a_h[RB], a_w[RB];
    unsigned a_nb[RB];
    bool a_ok[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        int m = m0 + (wave * RB + r) * 32 + lrow;
        a_ok[r] = m < p.M;
        m = a_ok[r] ? m : 0;
        const int n = m / p.ohw;
        const int rem = m - n * p.ohw;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_h[r] = oh * p.stride - p.pad;
        a_w[r] = ow * p.stride - p.pad;
        a_nb[r] = (unsigned)n * (unsigned)p.x_bs + (unsigned)(half * 8);
    }
    unsigned a_voff[RB][4];   // the four 16-byte pieces of a row's two 8-float slots; OOB must not wrap when offset
    auto tap_offsets = [&](int kh, int kw) {   // branch-free: this runs inside the MFMA stream
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int ih = a_h[r] + kh, iw = a_w[r] + kw;
            const bool ok = a_ok[r] & ((unsigned)ih < (unsigned)p.IHg) & ((unsigned)iw < (unsigned)p.IWg);
            ih >>= p.ups;
            iw >>= p.ups;
            const unsigned base = (a_nb[r] + (unsigned)(ih * p.W + iw) * (unsigned)p.Cin) * 4u;
            a_voff[r][0] = ok ? base : OOB;
            a_voff[r][1] = ok ? base + 16u : OOB;
            a_voff[r][2] = ok ? base + 64u : OOB;
            a_voff[r][3] = ok ? base + 80u : OOB;
            if (WI & 64) {   // (diagnostic) the same number of loads, but each one 1 KB contiguous per wave and cache-hot
#pragma unroll
                for (int q = 0; q < 4; ++q) a_voff[r][q] = (unsigned)(tid * 16 + (r * 4 + q) * 4096);
            }
        }
    };
    // ---- B staging (as in the staged kernel)
    unsigned b_voff[NBL];
    int b_lds[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int q = tid + 256 * i;
        const int piece = q / (BROWS * 4);
        const int rem = q - piece * (BROWS * 4);
        // 8 consecutive lanes (one ds_write_b128 bank group) write rows r and r+4: with the 80-byte row stride their two
        // 64-byte segments are 320 B = 64 (mod 128) apart, i.e. all 32 banks exactly once (rows r, r+1 overlap on 4 banks)
        // DMA: chunk q lands at LDS byte 16 q = (piece, row = rem >> 2, physical chunk rem & 3), which holds the logical
        // chunk (rem & 3) ^ ((row >> 2) & 3) of that row
        const int ch = DMA ? ((rem & 3) ^ ((rem >> 4) & 3)) : (rem & 3);
        const int row = DMA ? (rem >> 2) : (rem >> 5) * 8 + ((rem >> 3) & 3) + 4 * ((rem >> 2) & 1);
        const int n = n0 + row;
        const bool ok = piece < NP && n < p.Cout;
        b_voff[i] = ok ? (unsigned)piece * p.w_plane_bytes + ((unsigned)n * (unsigned)p.Ktot) * 2u + (unsigned)(ch * 16) : OOB;
        b_lds[i] = piece < NP ? piece * B_PLANE + row * RS + ch * 16 : -1;
    }
    const int b_dummy = 2 * STAGE + tid * 16;
    // ---- tap reuse: this lane's pixels inside the tile, and the staging role of this thread
    int a_lm[RB], a_ow[RB];
    unsigned g_off[8];      // byte offset of (sample, row 0, source column, the 16-byte chunk this slot holds)
    int g_oh[8];            // row of the pixel above this slot's output pixel, in (upsampled) image coordinates
    const unsigned rowpitch = (unsigned)p.W * (unsigned)p.Cin * 4u;
    if (TR) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            a_lm[r] = (wave * RB + r) * 32 + lrow;
            a_ow[r] = a_lm[r] % p.OW;
        }
        const int nb = m0 / p.ohw;
        const int oh0 = (m0 - nb * p.ohw) / p.OW;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = i * 256 + tid;        // LDS slot: pixel q >> 3, physical chunk q & 7
            const int px = q >> 3;
            const int lc = (q & 7) ^ ((px >> 1) & 7);
            const int sg = px / p.OW, col = px - sg * p.OW;
            g_oh[i] = oh0 + sg - 1;
            g_off[i] = ((unsigned)nb * (unsigned)p.x_bs + (unsigned)(col >> p.ups) * (unsigned)p.Cin) * 4u + (unsigned)(lc * 16);
        }
    }
    u32x4 stg[8];           // (tap reuse) the next filter row on its way global -> LDS
    // loads of filter row (c2, h2); a pixel whose input row lies outside the image gets an out-of-range address, i.e. zeros
    // (with the nearest-x2 upsample folded in, image row ih and column col read source row ih >> 1, column col >> 1)
    auto load_row = [&](int i, int c2, int h2) {
        const int ih = g_oh[i] + h2;
        const unsigned v = (unsigned)ih < (unsigned)p.IHg ? g_off[i] + (unsigned)(ih >> p.ups) * rowpitch : OOB;
        stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, v, __builtin_amdgcn_readfirstlane(min(c2, p.cchunks - 1) * (SBK * 4)), 0);
    };
    auto store_row = [&](int i, int c2, int h2) {
        *reinterpret_cast<u32x4*>(Bs + A_OFF + ((c2 + h2) & 1) * A_STAGE + (i * 256 + tid) * 16) = stg[i];
    };
    f32x4 ra[RB][4];   // [row block][k-step lo/hi 4 floats]
    u32x4 rb[NBL];

    f32x16 acc[RB][NT];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
    // fragment of k-step s: row lrow (+ 32 j), chunk 2 s + half — at its swizzled place in the DMA image
    const int fsw = (lrow >> 2) & 3;
    const int foff[2] = {DMA ? lrow * RS + ((half ^ fsw) << 4) : lrow * RS + half * 16,
                         DMA ? lrow * RS + (((2 + half) ^ fsw) << 4) : lrow * RS + half * 16 + 32};
    auto mfma_group = [&](const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16& c) {
        if (NP == 3) {
            c = Elt<F16>::mfma(a[2], b[0], c);
            c = Elt<F16>::mfma(a[0], b[2], c);
            c = Elt<F16>::mfma(a[1], b[1], c);
        }
        c = Elt<F16>::mfma(a[1], b[0], c);
        c = Elt<F16>::mfma(a[0], b[1], c);
        c = Elt<F16>::mfma(a[0], b[0], c);
    };

    // ---- software pipeline over the k-tiles (one barrier per tile, nothing but that barrier and the first fragment
    // read outside the MFMA stream).  While tile kt is multiplied:
    //   first k-step  (units 0..NT-1):   the weight tile kt+1 (already in registers) is written into the OTHER LDS stage,
    //                                    then the loads of weight tile kt+2 are issued
    //   second k-step (units NT..2NT-1): the activations of tile kt+1 (already in registers) are split into pieces — the
    //                                    first k-step's straight into af[0], whose old value is dead by then — and the loads
    //                                    of the activations of tile kt+2 are issued row block by row block
    // A unit = the RB*2*NP MFMAs of one k-step on one accumulator column; each unit first issues the LDS reads of the NEXT
    // unit's weight fragments (sched_barriers pin that order: left alone, the compiler waits on LDS in front of most MFMAs).
    constexpr int U = 2 * NT;
    constexpr int NTASK = 4 * RB;   // split4 calls per tile
    const int KT_all = p.ks * p.ks * p.cchunks;
    const int kt_first = (int)((int64_t)chunk * KT_all / p.ksplit);
    const int KT = (int)((int64_t)(chunk + 1) * KT_all / p.ksplit) - kt_first;   // k-tiles of this workgroup
    const int taps = p.ks * p.ks;
    int cc = kt_first / taps, tap = kt_first - cc * taps;
    int kh = tap / p.ks, kw = tap - kh * p.ks;
    // counters of the next tile to LOAD, branch-free (scalar selects): the loop body below is one basic block so that the
    // compiler can interleave the staging work with the MFMAs.  Past the last tile the counters run on harmlessly: the
    // loads are range-checked by the buffer resources (or hit valid activations) and their data is never multiplied.
    // k-tile order: the filter taps are the INNER loop and the 32-channel chunk the outer one, so nine consecutive tiles
    // re-read the same 128-byte lines of a pixel neighbourhood while they are hot in L2 instead of coming back to them
    // a whole channel sweep later (HBM reads per launch 3.9 GB -> see profiles/; less HBM traffic also means a higher
    // clock under the MFMA load, MI355X_MICROARCH.md "DVFS give-back").
    auto advance = [&]() {
        const bool wrapw = kw + 1 == p.ks;
        const bool wraph = wrapw && (kh + 1 == p.ks);
        kw = wrapw ? 0 : kw + 1;
        kh = wrapw ? (wraph ? 0 : kh + 1) : kh;
        cc += wraph ? 1 : 0;
        tap = kh * p.ks + kw;
    };
    auto load_a = [&](int r) {
        // uniform by construction (readfirstlane keeps it scalar); clamped so that the two look-ahead tiles past the end
        // of the k-loop re-read the last channel chunk instead of the bytes behind the pixel
        const int soff_a = __builtin_amdgcn_readfirstlane(min(cc, p.cchunks - 1) * (SBK * 4));
        if (WI & 2) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ra[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[r][i], soff_a, 0));
    };
    auto load_b = [&]() {
        // clamped: the two look-ahead loads past the last tile re-read the last tile instead of running off the planes
        const int soff_b = __builtin_amdgcn_readfirstlane(min((tap * p.Cin + cc * SBK) * 2, (p.Ktot - SBK) * 2));
        if (WI & 4) return;
#pragma unroll
        for (int i = 0; i < NBL; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff_b, B_AUX);
    };
    // LDS-DMA of the weight tile the counters point at into the stage at byte offset `so`: round i of the 4 waves covers LDS
    // bytes [4096 i, 4096 i + 4096); the last round is half empty for NT = 5 (whole waves: NT * 384 chunks is a multiple of 64)
    auto dma_b = [&](int so) {
        const int soff_b = __builtin_amdgcn_readfirstlane(min((tap * p.Cin + cc * SBK) * 2, (p.Ktot - SBK) * 2));
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
            if (256 * i + 64 * wave < BROWS * 4 * NP)
                lds_dma16(rw, Bs + so + i * 4096 + wave * 1024, b_voff[i], soff_b);
        }
    };
    auto store_b = [&](int i, int so) {
        if (WI & 4) return;
        if (256 * (i + 1) <= BROWS * 4 * NP)   // compile-time: this staging round is full
            *reinterpret_cast<u32x4*>(Bs + so + b_lds[i]) = rb[i];
        else
            *reinterpret_cast<u32x4*>(Bs + (b_lds[i] >= 0 ? so + b_lds[i] : b_dummy)) = rb[i];
    };
    auto join = [&](const u32x2 (&lo)[NP], const u32x2 (&hi)[NP], bf16x8 (&out)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            u32x4 v;
            v.x = lo[q].x; v.y = lo[q].y; v.z = hi[q].x; v.w = hi[q].y;
            out[q] = __builtin_bit_cast(bf16x8, v);
        }
    };

    auto sp4 = [&](f32x4 v, u32x2 (&out)[NP]) {
        if (WI & 1) {   // (diagnostic) raw bits instead of pieces: no VALU work
            const u32x4 b = __builtin_bit_cast(u32x4, v);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                out[q].x = b[q & 3];
                out[q].y = b[(q + 1) & 3];
            }
        } else {
            split4<NP, F16>(v, out, p.ovf);
        }
    };
    // tap reuse: 16-byte piece i of row block r under tap kw2 from the filter row with stage parity par
    auto lds_piece = [&](int r, int i, int par, int kw2) -> f32x4 {
        const bool in = (unsigned)(a_ow[r] + kw2 - 1) < (unsigned)p.OW;
        const int row = in ? a_lm[r] + kw2 - 1 : 256;   // 256 = the row of zeros
        const int ch = (i < 2 ? 0 : 4) + half * 2 + (i & 1);
        return *reinterpret_cast<const f32x4*>(Bs + A_OFF + par * A_STAGE + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
    };
    int kh_cur = 0, cc_cur = 0;               // (tap reuse) filter row of the tile being multiplied
    f32x4 a_nx[1];                            // (tap reuse) the piece read from LDS one unit ahead of its split
    bf16x8 af[2][RB][NP];   // A fragments of the current tile (both k-steps), in registers
    bf16x8 afn1[RB][NP];    // second k-step of the next tile (af[1] is live until the last unit)
    // prologue: tile 0 -> af / LDS stage 0, tile 1 -> ra / rb in flight
    if (TR) {   // filter row (0, 0) -> stage 0, the zero rows of both stages; its successors follow from inside the loop
#pragma unroll
        for (int i = 0; i < 8; ++i) load_row(i, 0, 0);
        if (DMA) dma_b(0); else load_b();
        if (tid < 16) *reinterpret_cast<u32x4*>(Bs + A_OFF + (tid >> 3) * A_STAGE + 256 * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 8; ++i) store_row(i, 0, 0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[r][i] = lds_piece(r, i, 0, 0);
    } else {
        tap_offsets(kh, kw);
#pragma unroll
        for (int r = 0; r < RB; ++r) load_a(r);
        if (DMA) dma_b(0); else load_b();
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        split8<NP, F16>(ra[r][0], ra[r][1], af[0][r], p.ovf);
        split8<NP, F16>(ra[r][2], ra[r][3], af[1][r], p.ovf);
    }
    if (!DMA) {
#pragma unroll
        for (int i = 0; i < NBL; ++i) store_b(i, 0);
    }
    advance();
    if (!TR) {
        tap_offsets(kh, kw);
#pragma unroll
        for (int r = 0; r < RB; ++r) load_a(r);
    }
    if (!DMA) load_b();

    stamp(1);
    // One k-tile.  KWC: the tap column of the tile as a compile-time constant in the tap-reuse instantiation, whose loop is
    // unrolled by three so that "first / second tile of a filter row" costs no branch (a branch inside the MFMA stream lets the
    // compiler sink the split work of a whole tile into its last block: measured 5564 cycles per k-tile); -1 otherwise.
    auto tile = [&](auto KWC, const int kt) __attribute__((always_inline)) {
        constexpr int kw_cur = TR ? decltype(KWC)::value : 0;
        const int so = (kt & 1) * STAGE;
        if (!(WI & 8)) __syncthreads();   // weight tile kt is visible; every wave is done with the other stage (tile kt-1)
        if (DMA && kt + 1 < KT) dma_b(STAGE - so);   // the counters are at tile kt+1 here: a whole tile for the DMA to land
        bf16x8 b_cur[NP], b_nxt[NP];
        const unsigned char* bf = Bs + so;
        u32x2 pl[NP], ph[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) b_nxt[q] = b_cur[q] = *reinterpret_cast<const bf16x8*>(bf + q * B_PLANE + foff[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = u / NT, j = u % NT;
            if (u + 1 < U) {
                const int s1 = (u + 1) / NT, j1 = (u + 1) % NT;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    if (!(WI & 16)) b_nxt[q] = *reinterpret_cast<const bf16x8*>(bf + q * B_PLANE + j1 * 32 * RS + foff[s1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TR) {   // the NEXT filter row: loaded during the first tile of this one, written to LDS during the second
                const bool wrap = kh_cur == 2;
                const int c2 = wrap ? cc_cur + 1 : cc_cur, h2 = wrap ? 0 : kh_cur + 1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (s == 1 && i * NT / 8 == j && kw_cur == 0) load_row(i, c2, h2);
                    if (s == 0 && i * NT / 8 == j && kw_cur == 1) store_row(i, c2, h2);
                }
                // activations of tile kt+1 -> pieces: the filter row is in LDS already, so the split is spread over BOTH k-steps
                // (one 16-byte piece per unit, fetched one unit ahead): units 1..4 make the second k-step's fragments (afn1),
                // units 5..8 the first k-step's (af[0], free once this tile's first k-step is done)
                if (u >= 1 && u <= 8) {
                    const int t = u - 1, r = (t & 3) >> 1, i = (t < 4 ? 2 : 0) + (t & 1);
                    if (i == 0 || i == 2) sp4(a_nx[0], pl);
                    if (i == 1) {
                        sp4(a_nx[0], ph);
                        join(pl, ph, af[0][r]);
                    }
                    if (i == 3) {
                        sp4(a_nx[0], ph);
                        join(pl, ph, afn1[r]);
                    }
                }
                if (u <= 7) {
                    const int t = u, r = (t & 3) >> 1, i = (t < 4 ? 2 : 0) + (t & 1);
                    const int kw_n = kw_cur == 2 ? 0 : kw_cur + 1;
                    const int par_n = kw_cur == 2 ? (c2 + h2) & 1 : (cc_cur + kh_cur) & 1;
                    a_nx[0] = lds_piece(r, i, par_n, kw_n);
                }
            }
            if (s == 0 && SPREAD) {   // as below, but each staging register is re-loaded (tile kt+2) right after its LDS write
                if (j == 0) {
                    advance();
                    soff_bs = __builtin_amdgcn_readfirstlane(min((tap * p.Cin + cc * SBK) * 2, (p.Ktot - SBK) * 2));
                }
#pragma unroll
                for (int i = 0; i < NBL; ++i)
                    if (i * NT / NBL == j) {
                        store_b(i, STAGE - so);
                        rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff_bs, B_AUX);
                    }
            } else if (s == 0) {   // weight tile kt+1 -> the other stage, spread over the first k-step's units
                if (!DMA) {
#pragma unroll
                    for (int i = 0; i < NBL; ++i)
                        if (i * NT / NBL == j) store_b(i, STAGE - so);
                }
                if (j == NT - 1) {
                    advance();
                    if (!DMA) load_b();
                }
            } else if (TR) {   // (done above)
            } else {        // activations of tile kt+1 -> pieces, task t = (row block, 4-float slot) in unit t*NT/NTASK
                if (j == 0) tap_offsets(kh, kw);   // of the tile advance() moved to; its loads follow once ra is free
#pragma unroll
                for (int t = 0; t < NTASK; ++t) {
                    if (t * NT / NTASK != j) continue;
                    const int r = t / 4, i = t % 4;
                    if (PRESPLIT) {   // (diagnostic) the pieces come from memory: 6 loads of 16 B instead of 4 + the split
                        if (t == 0) soff_as = __builtin_amdgcn_readfirstlane(min(cc, p.cchunks - 1) * (SBK * 2));
                        if (i == 1 || i == 3) {
                            bf16x8 (&dst)[NP] = i == 1 ? af[0][r] : afn1[r];
#pragma unroll
                            for (int q = 0; q < NP; ++q) dst[q] = __builtin_bit_cast(bf16x8, rp[r][(i >> 1) * 3 + q]);
#pragma unroll
                            for (int q = 0; q < NP; ++q)   // piece plane q, k-step i >> 1: the fp32 element offset halved
                                rp[r][(i >> 1) * 3 + q] = __builtin_amdgcn_raw_buffer_load_b128(
                                    rx, a_voff[r][0] == OOB ? OOB : (a_voff[r][0] >> 1) + (unsigned)q * pplane + (unsigned)(i >> 1) * 32u, soff_as, 0);
                        }
                        continue;
                    }
                    if (i == 0) sp4(ra[r][0], pl);
                    if (i == 1) {
                        sp4(ra[r][1], ph);
                        join(pl, ph, af[0][r]);
                    }
                    if (i == 2) sp4(ra[r][2], pl);
                    if (i == 3) {
                        sp4(ra[r][3], ph);
                        join(pl, ph, afn1[r]);
                        if (!SPREAD) load_a(r);
                    }
                    if (SPREAD) {   // each 16-byte piece is re-loaded (tile kt+2) as soon as it has been split
                        if (t == 0) soff_as = __builtin_amdgcn_readfirstlane(min(cc, p.cchunks - 1) * (SBK * 4));
                        ra[r][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[r][i], soff_as, 0));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) mfma_group(af[s][r], b_cur, acc[r][j]);
#pragma unroll
            for (int q = 0; q < NP; ++q) b_cur[q] = b_nxt[q];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int q = 0; q < NP; ++q) af[1][r][q] = afn1[r][q];
        if (TR && kw_cur == 2) {   // last tap of a filter row -> the next filter row (scalar selects)
            const bool wh = kh_cur == 2;
            kh_cur = wh ? 0 : kh_cur + 1;
            cc_cur += wh ? 1 : 0;
        }
    };
    if (TR) {   // (K unsplit and 3 x 3: the tile count is a multiple of three and the first tile has kw = 0)
        for (int kt = 0; kt < KT; kt += 3) {
            tile(std::integral_constant<int, 0>{}, kt);
            tile(std::integral_constant<int, 1>{}, kt + 1);
            tile(std::integral_constant<int, 2>{}, kt + 2);
        }
    } else {
        for (int kt = 0; kt < KT; ++kt) tile(std::integral_constant<int, -1>{}, kt);
    }
    stamp(2);
    if (p.ksplit > 1) {   // raw partial sums; bias / embedding / residual are added once, by the reduction
        SplitP q = p;
        q.y = p.partial + (int64_t)chunk * p.M * p.Cout;
        q.bias = q.emb = q.res = nullptr;
        q.out_nchw = 0;
        q.y_ld = p.Cout;
#pragma unroll
        for (int r = 0; r < RB; ++r) split_epilogue<NT>(q, acc[r], m0, n0, wave * RB + r, lrow, half, SBM * RB);
        return;
    }
    // (the 128-row kernels with 4 or 5 column tiles run two workgroups per CU at 256 registers each and have none to spare:
    // the statistics epilogue would spill values that live across the k-loop, so those keep the standalone pass)
    constexpr bool STATS_OK = RB == 2 || NT <= 3;
    if (STATS_OK && p.stats) {
        // one shifted fp32 partial per lane and column over its 16 * RB rows (the fp64 conversion after the last store keeps
        // the register pressure of the epilogue where it was: anything more spills values that live across the k-loop)
        StatAcc<NT> st;
        st.clear();
#pragma unroll
        for (int r = 0; r < RB; ++r) split_epilogue<NT, true>(p, acc[r], m0, n0, wave * RB + r, lrow, half, SBM * RB, st, r == 0);
        double ds[NT], dq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) ds[j] = dq[j] = 0.0;
        st.flush(ds, dq);
        stats_reduce<NT>(p, ds, dq, Bs, m0, n0, SBM * RB, tid, wave, lrow, half);
        if (STAMP) wait_all();   // (diagnostic) the stores have left the wave
        stamp(3);
        return;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) split_epilogue<NT>(p, acc[r], m0, n0, wave * RB + r, lrow, half, SBM * RB);
    if (STAMP) wait_all();
    stamp(3);
}

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
static bool tr_shape_ok(int ks, int stride, int pad, int OW, int IWg, int OH, int IHg, int ksplit, int out_nchw, int ohw, int64_t M, int Cin) {
    return ks == 3 && stride == 1 && pad == 1 && OW == IWg && OH == IHg && ksplit == 1 && !out_nchw &&
           (OW == 32 || OW == 64 || OW == 128 || OW == 256) && ohw % (2 * SBM) == 0 && M % (2 * SBM) == 0 && Cin % SBK == 0;
}
static bool conv_tr_ok(const SplitP& p) {
    return tr_enabled() && !p.stamps && tr_shape_ok(p.ks, p.stride, p.pad, p.OW, p.IWg, p.OH, p.IHg, p.ksplit, p.out_nchw, p.ohw, p.M, p.Cin);
}
// would conv2d_split(a, nt, ksplit, ad) run the tap-reuse instantiation?  (conv2d_variant: its launches are a kind of their own)
bool conv2d_split_tr(const ConvArgs& a, int nt, int ksplit, int ad) {
    if (!(tr_enabled() && !a.stamps && a.precision == PREC_BF16X6 && ad == 2 && nt == 5)) return false;
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
        switch (p.diag) {
            case 0: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 1>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 3>), grid, dim3(256), 0, s, p); break;
            case 4: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 5>), grid, dim3(256), 0, s, p); break;
            case 8: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 9>), grid, dim3(256), 0, s, p); break;
            case 16: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 17>), grid, dim3(256), 0, s, p); break;
            case 31: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 32>), grid, dim3(256), 0, s, p); break;
            case 32: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 33>), grid, dim3(256), 0, s, p); break;
            case 256: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 257>), grid, dim3(256), 0, s, p); break;
            case 512: hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 1, true>), grid, dim3(256), 0, s, p); break;
            default: fail("conv stamps: what-if %d is not instantiated in this build (0, 2, 4, 8, 16, 31, 32, 256, 512 are; 1, 3, 64, 128 were measured in round 2 and their cases removed to keep the build short: add the case back to re-measure)", p.diag);
        }
        check_launch("conv_split_ad2_stamped");
        return;
    }
    if (ad == 2 && nt == 5 && NP == 3 && !F16 && conv_tr_ok(p)) {
        hipLaunchKernelGGL((conv_split_ad_kernel<5, 3, false, 2, false, 0, true>), grid, dim3(256), 0, s, p);
        check_launch("conv_split_ad2_tr");
        return;
    }
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
    if ((int64_t)cdiv((int)M, ad0 * SBM) * cdiv(t32, nt_default) > (ad0 == 1 ? 320 : 160) || KT < 32) return;
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
                const double over = std::max(1.0, (double)blocks * ks / 256.0);
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
