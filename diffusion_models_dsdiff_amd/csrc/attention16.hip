// attention16.hip — flash attention on 16-bit operands (fp16 or bf16), fp32 accumulation and softmax statistics:
// the attention of the transformer backbone in the arithmetic BASELINE configs[4] names (timm Attention under fp16 autocast,
// UNet_DS_Diff/DiT_models.py:101-122: q k^T and attn v are half-precision matmuls, the softmax is fp32).
//
// One workgroup = (sample, head, 128 queries), 4 waves x 32 queries, 64 keys per LDS stage, two stages (one barrier per tile).
//   * S^T = K Q^T (v_mfma_f32_32x32x16): the lane owns ONE query column, so the row maximum / sum are in-lane plus one
//     exchange with the other lane half, and exp2(S^T) is already the B operand of O^T = V^T P^T: P never leaves registers.
//     The accumulator gives a lane the keys 4h + (r & 3) + 8 (r >> 2) of a 32-key sub-tile; k-step s of the second product
//     takes registers 8s .. 8s+7 as they are and the V^T fragment is fetched in the matching key order.
//   * K is staged k-step-major: plane s holds [key][d = 16 s .. 16 s + 15] (32 B per key, 16-byte halves XOR-swizzled by
//     (key >> 3) & 1): the A-operand read (lane = key, 16 B) is a conflict-free ds_read_b128 for every head dim.
//   * V is staged d-tile-major: plane t holds [key][d = 32 t .. 32 t + 31] (64 B per key); the A operand of the second
//     product needs V TRANSPOSED (lane = d, 8 keys) and comes from ds_read_b64_tr_b16, the hardware transpose read: four
//     consecutive keys x 16 d per 16-lane group, 256 contiguous bytes per 32 lanes — conflict-free, and no 2-byte scatter
//     stores (what made the split kernel's V^T staging conflict on 48 % of its LDS cycles).
//   * Softmax in base 2: q arrives pre-multiplied by hd^-1/2 log2(e) (the qkv GEMM's epilogue does it in fp32 before the one
//     rounding to 16 bits), the NEGATED running maximum is the C operand of the first S^T MFMA, so the score tile comes out
//     as s - m and p = exp2(that) costs one v_exp_f32.  The maximum is only moved when a score exceeds it by more than 2^THR
//     (a wave-uniform branch): O, l and the pending tile are then rescaled together, before any of the tile is exponentiated
//     (cdna_hip_programming.md T13's safe order).  p <= 2^8 in the 16-bit P, sums in fp32.
// Head dims: multiples of 8 up to 128 (padding chunks are zeroed once); keys beyond Tk are masked in the last tile.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace dsd {

namespace a16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct AF;
template <>
struct AF<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <>
struct AF<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

struct A16P {
    const void *q, *k, *v;
    void* out;
    int ldq, ldk, ldv, ldo, q_hs, k_hs, v_hs;
    int N, Tq, Tk, heads, d;
    float scale_q;   // applied to q in fp32 before it is rounded back (1 = q arrives pre-scaled)
    float thr;       // rescale threshold in log2 units
};

// WI: what-if bits of the diagnostic instantiations (dsd_bench_attention_half; results are then garbage): 1 no softmax arithmetic
// (P = the raw scores), 2 no K / V staging after the first tile (no global loads, LDS writes or barriers in the loop), 4 no
// second product (no V reads, no P V MFMAs), 8 no first product (no K reads, no Q K^T MFMAs)
template <typename T16, int NKS, int WI = 0>
__global__ __launch_bounds__(256, 2) void attention16_kernel(A16P a) {
    using F = AF<T16>;
    constexpr int KEYS = 64;
    constexpr int DT = (NKS + 1) / 2;        // 32-wide d tiles of O^T
    constexpr int CH = 2 * NKS;              // 16-byte chunk slots per key row (K side)
    constexpr int KPL = KEYS * 32 + (NKS == 2 ? 64 : NKS == 3 ? 96 : 32);      // bytes per K plane; the pad spreads the planes that 8 consecutive staging lanes write (1-2 key rows) over all 32 banks
    constexpr int VPL = KEYS * 64 + 64;      // bytes per V plane
    constexpr int NLD = (KEYS * CH + 255) / 256;
    constexpr int STG = NKS * KPL + DT * VPL;   // one stage (K planes | V planes); two stages: ONE barrier per 64-key tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    // block -> (query block, head, sample).  The 32 query blocks of one (sample, head) re-read the same K / V (1 MB at 4096
    // tokens of 64): with the hardware's round-robin of consecutive blocks over the 8 XCDs every XCD's L2 would fetch every
    // head (PMC, round 3: 4.5x the algorithmic bytes from beyond L2); here workgroups b and b + 8 — the same XCD — walk the
    // query blocks of the same head, so a head's K / V is fetched by ONE L2.  Placement only: any mapping is correct.
    const int QB = (a.Tq + 127) / 128, HB = a.heads * a.N;
    int bid = blockIdx.x, hb, qb;
    if ((HB & 7) == 0) {
        const int slot = bid >> 3;
        hb = (slot / QB) * 8 + (bid & 7);
        qb = slot - (slot / QB) * QB;
    } else {
        hb = bid / QB;
        qb = bid - hb * QB;
    }
    const int n = hb / a.heads, head = hb - n * a.heads;
    const int q = qb * 128 + wave * 32 + lrow;
    const bool q_ok = q < a.Tq;
    const int hch = a.d >> 3;   // valid 16-byte chunks per row

    // zero the whole stage once: padding chunks (d >= head dim) are never written again
    for (int i = tid * 16; i < 2 * STG; i += 256 * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    // Q^T fragments (B operand of S^T = K Q^T): this lane's query, k-step s covers d = 16 s + 8 half .. + 8
    typename F::v8 qf[NKS];
    {
        const T16* qp = reinterpret_cast<const T16*>(a.q) + ((int64_t)n * a.Tq + (q_ok ? q : 0)) * a.ldq + (int64_t)head * a.q_hs;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int c = 2 * s + half;
            typename F::v8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (T16)0.f;
            if (q_ok && c < hch) t = *reinterpret_cast<const typename F::v8*>(qp + c * 8);
            if (a.scale_q != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (T16)((float)t[e] * a.scale_q);
            }
            qf[s] = t;
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    f32x16 negm;   // -m in every register: the C operand of the first S^T MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;   // l_run: this lane's 16 keys of every sub-tile only (the halves are added at the end)

    const T16* kbase = reinterpret_cast<const T16*>(a.k) + (int64_t)n * a.Tk * a.ldk + (int64_t)head * a.k_hs;
    const T16* vbase = reinterpret_cast<const T16*>(a.v) + (int64_t)n * a.Tk * a.ldv + (int64_t)head * a.v_hs;
    u32x4 kreg[NLD], vreg[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int kr = i / CH, c = i - kr * CH;
            const int key = k0 + kr;
            u32x4 kv = u32x4{0u, 0u, 0u, 0u}, vv = kv;
            if (kr < KEYS && key < a.Tk && c < hch) {
                kv = *reinterpret_cast<const u32x4*>(kbase + (int64_t)key * a.ldk + c * 8);
                vv = *reinterpret_cast<const u32x4*>(vbase + (int64_t)key * a.ldv + c * 8);
            }
            kreg[j] = kv;
            vreg[j] = vv;
        }
    };
    // fragment read offsets
    const int kofs = (lrow * 32 + half * 16) ^ (((lrow >> 3) & 1) << 4);                      // + s * KPL + sub * 1024
    const int g = lane >> 4, li = lane & 15;
    const int vofs = (4 * half + (li >> 2)) * 64 + (16 * (g & 1) + 4 * (li & 3)) * 2;         // + t * VPL + sub * 2048 + s2 * 1024 + j * 512

    auto stage_write = [&](int st) {
        unsigned char* Kw = lds + st * STG;
        unsigned char* Vw = Kw + NKS * KPL;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int kr = i / CH, c = i - kr * CH;
            if (kr < KEYS && c < hch) {
                *reinterpret_cast<u32x4*>(Kw + (c >> 1) * KPL + ((kr * 32 + (c & 1) * 16) ^ (((kr >> 3) & 1) << 4))) = kreg[j];
                *reinterpret_cast<u32x4*>(Vw + (c >> 2) * VPL + kr * 64 + (c & 3) * 16) = vreg[j];
            }
        }
    };
    fetch(0);
    __syncthreads();       // the zero fill has landed
    stage_write(0);
    __syncthreads();
    bool first = true;
    int st = 0;
    for (int k0 = 0; k0 < a.Tk; k0 += KEYS, st ^= 1) {
        // tile k0 is in stage st (visible: barrier below / above); the next tile travels global -> registers while this one is
        // multiplied and is written to the OTHER stage afterwards — every wave left that stage before the last barrier
        const unsigned char* Kl = lds + st * STG;
        const unsigned char* Vl = Kl + NKS * KPL;
        if (k0 + KEYS < a.Tk && !(WI & 2)) fetch(k0 + KEYS);
#pragma unroll
        for (int sub = 0; sub < KEYS / 32; ++sub) {
            if (k0 + sub * 32 >= a.Tk) break;
            // S^T[key][q] - m = sum_d K[key][d] Q[q][d] + (-m)
            typename F::v8 kf[NKS];   // all fragment reads first: their latencies overlap instead of queueing in front of each MFMA
#pragma unroll
            for (int s = 0; s < NKS; ++s)
                if (!(WI & 8)) kf[s] = *reinterpret_cast<const typename F::v8*>(Kl + s * KPL + sub * 1024 + kofs);
            f32x16 sacc = negm;
            if (!(WI & 8)) {
                sacc = F::mfma(kf[0], qf[0], negm);
#pragma unroll
                for (int s = 1; s < NKS; ++s) sacc = F::mfma(kf[s], qf[s], sacc);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = (float)(r + lane) * 0.01f + m_run;
            }
            if (k0 + sub * 32 + 32 > a.Tk) {   // last, partial sub-tile: keys beyond Tk take no part
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    sacc[r] = key < a.Tk ? sacc[r] : -INFINITY;
                }
            }
            if (!(WI & 1)) {
                float tmax = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
    #pragma unroll
                for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[r]), sacc[r + 1]);
                tmax = fmaxf(tmax, sacc[15]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                // the maximum moves only when some query of the wave saw a score more than 2^thr above it (always on the first tile)
                const bool need = first || !(tmax <= a.thr);
                if (__any(need)) {
                    const float delta = need ? tmax : 0.f;          // first tile: may be negative — m becomes the tile's maximum
                    const float corr = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
    #pragma unroll
                    for (int t = 0; t < DT; ++t)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) o[t][r] *= corr;
                    l_run *= corr;
                    m_run += delta;
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        negm[r] = -m_run;
                        sacc[r] -= delta;
                    }
                    first = false;
                }
                float psum = 0.f;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = __builtin_amdgcn_exp2f(sacc[r]);
                    psum += sacc[r];
                }
                l_run += psum;
            }
            typename F::v8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[s2][e] = (T16)sacc[8 * s2 + e];
            // O^T[d][q] += sum_key V[key][d] P[key][q]; element j of k-step s2 is key 16 s2 + 8 (j >> 2) + 4 half + (j & 3)
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                if (WI & 4) {
                    o[t][0] += (float)pf[0][0] + (float)pf[1][7];
                    continue;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const unsigned char* vp = Vl + t * VPL + sub * 2048 + s2 * 1024 + vofs;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 512));
                    const typename F::v8 vf = __builtin_bit_cast(typename F::v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[t] = F::mfma(vf, pf[s2], o[t]);
                }
            }
        }
        if (k0 + KEYS < a.Tk && !(WI & 2)) {
            stage_write(st ^ 1);
            __syncthreads();
        }
        if (WI & 2) st ^= 1;   // (keep reading the first tile)
    }
    if (!q_ok) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    T16* op = reinterpret_cast<T16*>(a.out) + ((int64_t)n * a.Tq + q) * a.ldo + (int64_t)head * a.d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = t * 32 + 8 * r4 + 4 * half;
            if (d < a.d) {   // head dim % 8 == 0 and d % 4 == 0: the four values are valid together
                typename F::v4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (T16)(o[t][4 * r4 + e] * inv);
                *reinterpret_cast<typename F::v4*>(op + d) = h;
            }
        }
}

// ---- head dim 64 (every DiT size but XL): K / V by LDS-DMA, three stages --------------------------------------------------------
// What-if table of the kernel above at 16 x 4096 tokens x 12 heads (tools/attn_whatif.py, round 3): 1294 us; without the
// softmax arithmetic 910; WITHOUT THE STAGING (global -> registers -> ds_write, barrier) 828; with neither 634.  Staging through
// registers costs a third of the kernel: 32 registers of loads in flight per lane, 8 ds_write_b128 per tile, and a tile that is
// requested only ONE tile ahead of its use.  Here a tile goes global -> LDS directly (buffer_load ... lds, 16 bytes per lane, 1 KB
// of LDS per wave-instruction, 4 instructions per wave and tile), TWO tiles ahead, behind a counted vmcnt, into a ring of three
// 16 KB stages; still one barrier per tile.
//   stage = K [64 keys][128 B] | V [64 keys][128 B], rows whole: 8 consecutive lanes fetch one 128-byte row of one key.
//   K row: 16-byte chunk c stored at c ^ ((key >> 1) & 7): the A-operand read of S^T (16 consecutive lanes = 16 consecutive
//   keys, one chunk) covers the 16 slots of the 256-byte bank row once.  V row: 64-byte half t stored at t ^ ((key >> 1) & 1):
//   the transposed read (4 keys x 64 B per 32 lanes) covers them once as well.  Both swizzles are applied to the SOURCE
//   address of the DMA (the LDS side of a DMA instruction is fixed: lane i -> byte 16 i of the piece).
//   Keys beyond Tk: the lane's offset is sent out of range and the hardware delivers zeros (the scores are masked as above).
// One LDS-DMA instruction, HIDDEN from hipcc's wait-count bookkeeping (inline asm): with the builtin, hipcc assumes every later
// LDS read may alias a DMA still in flight and waits for the older tile at the top of each tile — the counted vmcnt below is the
// only wait these need.  M0 (the LDS destination) is written in the same statement and restored; rsrc / soffset / M0 come from
// readfirstlane, hence the leading s_nop (cdna_hip_programming.md "What hipcc does not do").
typedef int i32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void dma16h(i32x4 rsrc, unsigned lds_addr, unsigned voff, int soff) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
        : "memory");
}
static __device__ __forceinline__ i32x4 make_rsrc16(const void* base, unsigned bytes) {
    const uint64_t b = (uint64_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));   // (stride 0, no swizzle)
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

typedef int v2i __attribute__((ext_vector_type(2)));
// eight ds_read_b64_tr_b16 (d tile t = 0, 1 x k-step s2 = 0, 1 x key octet j = 0, 1) at a0 / a1 + OFF + (16 s2 + 8 j) * 128 AND the
// wait for them, in ONE statement.  (The first version issued the reads in front of the softmax and waited in a second statement
// in front of the second product, to hide their latency behind the softmax of the same wave.  hipcc does not know that the
// destination registers of an asm load are in flight: under the 128-register cap of the 8-wave build it spilled four of them
// between the two statements and reused them for v_exp results — non-finite output, caught by tests/test_half_gpu.py.  One
// statement leaves no live range to tamper with; the latency is covered by the other three waves of the SIMD.)
template <int OFF>
static __device__ __forceinline__ void tr_read8(v2i (&r)[2][2][2], unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %4, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0][0][0]), "=&v"(r[0][0][1]), "=&v"(r[0][1][0]), "=&v"(r[0][1][1]), "=&v"(r[1][0][0]), "=&v"(r[1][0][1]),
          "=&v"(r[1][1][0]), "=&v"(r[1][1][1])
        : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 8 * 128), "i"(OFF + 16 * 128), "i"(OFF + 24 * 128)
        : "memory");
}

// NW waves (4 or 8) share the stages, each wave owns QW blocks of 32 queries (1 or 2).  8 waves = 256 queries per workgroup halve
// the K / V bytes and barriers per FLOP: SLOWER (1185 vs 1045 us) while the kernel needed 146 registers — one workgroup per CU,
// nobody runs while eight waves meet at the barrier — and the FASTEST variant (886 vs 977 us) once it fitted 128 registers and two
// such workgroups share a CU: the product configuration.  QW = 2 gets the same 256 queries from FOUR waves (a K / V fragment is
// read from LDS once and multiplied with two query blocks; two independent MFMA chains and softmax streams per wave; 228
// registers): 944 us, between the two.
template <typename T16, int WI = 0, int NW = 4, int QW = 1>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void attention16_dma_kernel(A16P a) {   // (second argument = waves per SIMD: 8 waves in 128 registers, two workgroups per CU)
    using F = AF<T16>;
    constexpr int KEYS = 64, NKS = 4, DT = 2, QPB = 32 * NW * QW, PJ = 8 / NW;   // queries per block; K (and V) pieces per wave and tile
    constexpr int STG = 2 * KEYS * 128;   // K | V
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    const int QB = (a.Tq + QPB - 1) / QPB, HB = a.heads * a.N;
    int bid = blockIdx.x, hb, qb;
    if ((HB & 7) == 0) {   // (placement only, as above: the query blocks of a head share an XCD)
        const int slot = bid >> 3;
        hb = (slot / QB) * 8 + (bid & 7);
        qb = slot - (slot / QB) * QB;
    } else {
        hb = bid / QB;
        qb = bid - hb * QB;
    }
    const int n = hb / a.heads, head = hb - n * a.heads;
    int q[QW];
    bool q_ok[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        q[w] = qb * QPB + (wave * QW + w) * 32 + lrow;
        q_ok[w] = q[w] < a.Tq;
    }

    const T16* kbase = reinterpret_cast<const T16*>(a.k) + (int64_t)n * a.Tk * a.ldk + (int64_t)head * a.k_hs;
    const T16* vbase = reinterpret_cast<const T16*>(a.v) + (int64_t)n * a.Tk * a.ldv + (int64_t)head * a.v_hs;
    const i32x4 rk = make_rsrc16(kbase, ((unsigned)(a.Tk - 1) * (unsigned)a.ldk + 64u) * 2u);
    const i32x4 rv = make_rsrc16(vbase, ((unsigned)(a.Tk - 1) * (unsigned)a.ldv + 64u) * 2u);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // this wave's pieces of a tile: K keys 8 wave + (lane >> 3) (+ 32 with four waves), V the same keys
    const int pkey = wave * 8 + (lane >> 3), pc = lane & 7;
    unsigned kvo[PJ], vvo[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int key = pkey + 32 * j;
        kvo[j] = ((unsigned)key * (unsigned)a.ldk + 8u * (unsigned)(pc ^ ((key >> 1) & 7))) * 2u;
        vvo[j] = ((unsigned)key * (unsigned)a.ldv + 8u * (unsigned)((((pc >> 2) ^ ((key >> 1) & 1)) << 2) | (pc & 3))) * 2u;
    }
    auto issue = [&](int st, int k0) {
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + st * STG + wave * 1024));
        const int sk = __builtin_amdgcn_readfirstlane(k0 * a.ldk * 2), sv = __builtin_amdgcn_readfirstlane(k0 * a.ldv * 2);
        const bool tail = k0 + KEYS > a.Tk;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const bool ok = !tail || k0 + pkey + 32 * j < a.Tk;
            dma16h(rk, base + j * 4096, ok ? kvo[j] : 0xFFFFFFF0u, sk);
            dma16h(rv, base + KEYS * 128 + j * 4096, ok ? vvo[j] : 0xFFFFFFF0u, sv);
        }
    };
    const int NT = (a.Tk + KEYS - 1) / KEYS;
    issue(0, 0);
    if (NT > 1) issue(1, KEYS);

    // Q^T fragments (B operand of S^T = K Q^T): this lane's query, k-step s covers d = 16 s + 8 half .. + 8.  The scale is
    // applied unconditionally (x 1 is exact): hipcc then waits for these loads HERE, once — left pending on one path, their
    // vmcnt(0) lands in front of the first MFMA of the loop and waits for the tiles in flight in every iteration.
    typename F::v8 qf[QW][NKS];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        const T16* qp = reinterpret_cast<const T16*>(a.q) + ((int64_t)n * a.Tq + (q_ok[w] ? q[w] : 0)) * a.ldq + (int64_t)head * a.q_hs;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            typename F::v8 t = *reinterpret_cast<const typename F::v8*>(qp + (2 * s + half) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = q_ok[w] ? (T16)((float)t[e] * a.scale_q) : (T16)0.f;
            qf[w][s] = t;
        }
    }
    f32x16 o[QW][DT];
    f32x16 negm[QW];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int w = 0; w < QW; ++w) {
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[w][t][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[w][r] = 0.f;
        m_run[w] = l_run[w] = 0.f;
    }

    // fragment read offsets inside a stage
    int kofs[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) kofs[s] = lrow * 128 + (((2 * s + half) ^ ((lrow >> 1) & 7)) << 4);           // + sub * 4096
    const int g = lane >> 4, li = lane & 15;
    const int vkey = 4 * half + (li >> 2);                                                                       // + 32 sub + 16 s2 + 8 j
    const int vb = (li >> 3) & 1;                                                                                // (key >> 1) & 1 of that key
    int vofs[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) vofs[t] = KEYS * 128 + vkey * 128 + ((t ^ vb) << 6) + (16 * (g & 1) + 4 * (li & 3)) * 2;

    bool first = true;
    const float pbound = 16.f * __builtin_amdgcn_exp2f(fminf(a.thr, 11.f));   // (2^15 at most: P is 16-bit)
    for (int it = 0; it < NT; ++it) {
        const int k0 = it * KEYS;
        // this wave's pieces of tile `it` have landed (the younger tile's may still fly); after the barrier everybody's
        // have, and every wave has left tile it - 1, whose stage takes tile it + 2
        if (!(WI & 2) || it == 0) {
            if (it + 1 < NT) __builtin_amdgcn_s_waitcnt(NW == 4 ? 0x0074 : 0x0072);   // vmcnt(2 PJ) lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            if (it + 2 < NT && !(WI & 2)) issue((it + 2) % 3, k0 + 2 * KEYS);
        }
        const unsigned char* Sl = lds + ((WI & 2) ? 0 : (it % 3)) * STG;
        const unsigned sl_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)const_cast<unsigned char*>(Sl);
        auto sub_tile = [&](auto SUB) {
            constexpr int sub = decltype(SUB)::value;
            // scores of this sub-tile minus the running maximum: S^T[key][q] - m = sum_d K[key][d] Q[q][d] + (-m); keys beyond Tk -> -inf
            f32x16 sacc[QW];
            auto scores = [&]() {
                typename F::v8 kf[NKS];
#pragma unroll
                for (int s = 0; s < NKS; ++s)
                    if (!(WI & 8)) kf[s] = *reinterpret_cast<const typename F::v8*>(Sl + sub * 4096 + kofs[s]);
                if (!(WI & 8)) {
#pragma unroll
                    for (int w = 0; w < QW; ++w) sacc[w] = F::mfma(kf[0], qf[w][0], negm[w]);
#pragma unroll
                    for (int s = 1; s < NKS; ++s)
#pragma unroll
                        for (int w = 0; w < QW; ++w) sacc[w] = F::mfma(kf[s], qf[w][s], sacc[w]);
                } else {
#pragma unroll
                    for (int w = 0; w < QW; ++w)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[w][r] = (float)(r + lane) * 0.01f + m_run[w];
                }
                if (k0 + sub * 32 + 32 > a.Tk) {   // last, partial sub-tile: keys beyond Tk take no part
#pragma unroll
                    for (int w = 0; w < QW; ++w)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            sacc[w][r] = key < a.Tk ? sacc[w][r] : -INFINITY;
                        }
                }
            };
            scores();
            // OPTIMISTIC softmax: exponentiate first, look at the maximum only if that went wrong.  p = exp2(s - m) is safe while
            // s - m stays below the 16-bit range; a lane's sum of its 16 p bounds each of them, so "psum <= 16 * 2^thr" (one compare,
            // false for inf / NaN too) replaces the 15 v_max + cross-half exchange + compare of the tile maximum on every sub-tile.
            // When it fails for any query of the wave (and on the first sub-tile, whose m is still 0) the scores are computed again
            // — K is still in its stage — and the maximum is moved the exact way before they are exponentiated.
            typename F::v8 pf[QW][2];
            if (!(WI & 1)) {
                float psum[QW];
                bool redo = first;
                if (!redo) {
                    bool bad = false;
#pragma unroll
                    for (int w = 0; w < QW; ++w) {
                        psum[w] = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            sacc[w][r] = __builtin_amdgcn_exp2f(sacc[w][r]);
                            psum[w] += sacc[w][r];
                        }
                        bad |= !(psum[w] <= pbound);
                    }
                    redo = __any(bad);
                }
                if (redo) {
                    if (!first) scores();
#pragma unroll
                    for (int w = 0; w < QW; ++w) {
                        float tmax = fmaxf(fmaxf(sacc[w][0], sacc[w][1]), sacc[w][2]);
#pragma unroll
                        for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[w][r]), sacc[w][r + 1]);
                        tmax = fmaxf(tmax, sacc[w][15]);
                        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                        const bool need = first || !(tmax <= a.thr);
                        if (__any(need)) {
                            const float delta = need ? tmax : 0.f;          // first tile: may be negative — m becomes the tile's maximum
                            const float corr = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                            for (int t = 0; t < DT; ++t)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[w][t][r] *= corr;
                            l_run[w] *= corr;
                            m_run[w] += delta;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                negm[w][r] = -m_run[w];
                                sacc[w][r] -= delta;
                            }
                        }
                        psum[w] = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            sacc[w][r] = __builtin_amdgcn_exp2f(sacc[w][r]);
                            psum[w] += sacc[w][r];
                        }
                    }
                }
#pragma unroll
                for (int w = 0; w < QW; ++w) l_run[w] += psum[w];
            }
#pragma unroll
            for (int w = 0; w < QW; ++w)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[w][s2][e] = (T16)sacc[w][8 * s2 + e];
            first = false;
            // the eight transposed V fragments of this sub-tile (inline asm: hipcc puts s_waitcnt vmcnt(0) in front of the
            // ds_read_tr BUILTIN when LDS-DMA is in flight — it cannot tell the stages apart — which would serialise the tile
            // behind the two tiles being fetched)
            v2i vr[DT][2][2];
            if (!(WI & 4)) tr_read8<sub * 32 * 128>(vr, sl_addr + vofs[0], sl_addr + vofs[1]);
            if (WI & 4) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
#pragma unroll
                    for (int w = 0; w < QW; ++w) o[w][t][0] += (float)pf[w][0][0] + (float)pf[w][1][7];
            } else {
                // k-step outside, d tile inside: consecutive MFMAs go to DIFFERENT accumulators (a dependent pair would wait
                // for the first one's result)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int t = 0; t < DT; ++t) {
                        const typename F::v8 vf = __builtin_bit_cast(typename F::v8, __builtin_shufflevector(vr[t][s2][0], vr[t][s2][1], 0, 1, 2, 3));
#pragma unroll
                        for (int w = 0; w < QW; ++w) o[w][t] = F::mfma(vf, pf[w][s2], o[w][t]);
                    }
            }
        };
        sub_tile(std::integral_constant<int, 0>{});
        if (k0 + 32 < a.Tk) sub_tile(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        if (!q_ok[w]) continue;
        const float l_tot = l_run[w] + __shfl_xor(l_run[w], 32);
        const float inv = 1.f / l_tot;
        T16* op = reinterpret_cast<T16*>(a.out) + ((int64_t)n * a.Tq + q[w]) * a.ldo + (int64_t)head * a.d;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = t * 32 + 8 * r4 + 4 * half;
                typename F::v4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (T16)(o[w][t][4 * r4 + e] * inv);
                *reinterpret_cast<typename F::v4*>(op + d) = h;
            }
    }
}

// the DMA kernel takes head dim 64 with K / V addressable in 32-bit byte offsets from the (sample, head) base
static bool attn16_dma_ok(const A16P& p) {
    static const bool off = getenv("DSD_ATTN16_NO_DMA") != nullptr;   // A/B
    return !off && p.d == 64 && (int64_t)p.Tk * p.ldk * 2 < (1ll << 31) && (int64_t)p.Tk * p.ldv * 2 < (1ll << 31);
}

template <typename T16>
void launch_attn16(const A16P& p, int nks, hipStream_t s) {
    const dim3 grid((unsigned)(cdiv(p.Tq, 128) * p.heads * p.N)), block(256);
    if (attn16_dma_ok(p)) {
        // 256 queries per workgroup when the sequence has them (the K / V tile is fetched once for twice the work).  Same-box
        // A/B inside DiT-B/8 @512 (fp16 / bf16, ms per 12 calls): 8 waves x 32 queries 10.18 / 9.84, 4 waves x 64 queries
        // 10.29-10.79 / 9.86-9.96, 4 waves x 32 queries 11.09 / 10.43 (and 11.56 / 10.80 before the kernel fitted 128 registers)
        // (the 4 x 64 instantiation is no longer built: with the optimistic softmax it needs more than 256 registers)
        static const int variant = getenv("DSD_ATTN16_VARIANT") ? atoi(getenv("DSD_ATTN16_VARIANT")) : 0;   // A/B: 1 = 4 x 32
        const dim3 grid256((unsigned)(cdiv(p.Tq, 256) * p.heads * p.N));
        if (p.Tq < 256 || variant == 1)
            hipLaunchKernelGGL((attention16_dma_kernel<T16>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((attention16_dma_kernel<T16, 0, 8>), grid256, dim3(512), 0, s, p);
        return;
    }
    switch (nks) {
        case 1: hipLaunchKernelGGL((attention16_kernel<T16, 1>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((attention16_kernel<T16, 2>), grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL((attention16_kernel<T16, 3>), grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL((attention16_kernel<T16, 4>), grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL((attention16_kernel<T16, 5>), grid, block, 0, s, p); break;
        case 6: hipLaunchKernelGGL((attention16_kernel<T16, 6>), grid, block, 0, s, p); break;
        case 7: hipLaunchKernelGGL((attention16_kernel<T16, 7>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((attention16_kernel<T16, 8>), grid, block, 0, s, p); break;
    }
}

}  // namespace a16
using namespace a16;

void attention16_whatif(const Attn16Args& a, int wi, hipStream_t s);

bool attention16_shape_ok(int d) { return d >= 8 && d <= 128 && d % 8 == 0; }

void attention16(const Attn16Args& a, hipStream_t s) { attention16_whatif(a, -1, s); }

void attention16_whatif(const Attn16Args& a, int wi, hipStream_t s) {
    DSD_CHECK(attention16_shape_ok(a.d), "attention16: head dim %d unsupported (multiple of 8, <= 128)", a.d);
    DSD_CHECK(a.Tk >= 1 && a.Tq >= 1, "attention16: empty sequence");
    DSD_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.q_hs % 8 == 0 && a.k_hs % 8 == 0 && a.v_hs % 8 == 0 && a.ldo % 4 == 0,
              "attention16: rows must be 16-byte aligned");
    A16P p{};
    p.q = a.q; p.k = a.k; p.v = a.v; p.out = a.out;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo; p.q_hs = a.q_hs; p.k_hs = a.k_hs; p.v_hs = a.v_hs;
    p.N = a.N; p.Tq = a.Tq; p.Tk = a.Tk; p.heads = a.heads; p.d = a.d;
    p.scale_q = a.scale_q;
    p.thr = a.thr >= 0.f ? a.thr : 8.f;
    const int nks = cdiv(a.d, 16);
    if (wi >= 0) {   // diagnostic instantiations: fp16, head dim 64
        DSD_CHECK(nks == 4 && !a.bf16, "attention16 what-if: fp16, head dim 64 only");
        const dim3 grid((unsigned)(cdiv(p.Tq, 128) * p.heads * p.N)), block(256);
        switch (wi) {
            case 0: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 0>), grid, block, 0, s, p); break;
            case 1: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 1>), grid, block, 0, s, p); break;
            case 2: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 2>), grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 4>), grid, block, 0, s, p); break;
            case 8: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 8>), grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 3>), grid, block, 0, s, p); break;
            case 13: hipLaunchKernelGGL((attention16_kernel<_Float16, 4, 13>), grid, block, 0, s, p); break;
            case 100: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 0>), grid, block, 0, s, p); break;   // 100 + bits: the LDS-DMA kernel
            case 101: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 1>), grid, block, 0, s, p); break;
            case 102: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 2>), grid, block, 0, s, p); break;
            case 103: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 3>), grid, block, 0, s, p); break;
            case 200: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 0, 8>), dim3((unsigned)(cdiv(p.Tq, 256) * p.heads * p.N)), dim3(512), 0, s, p); break;   // 8 waves
            case 202: hipLaunchKernelGGL((attention16_dma_kernel<_Float16, 2, 8>), dim3((unsigned)(cdiv(p.Tq, 256) * p.heads * p.N)), dim3(512), 0, s, p); break;
            default: fail("attention16 what-if %d is not instantiated (0, 1, 2, 3, 4, 8, 13)", wi);
        }
        check_launch("attention16_whatif");
        return;
    }
    if (a.bf16) launch_attn16<__bf16>(p, nks, s); else launch_attn16<_Float16>(p, nks, s);
    check_launch("attention16");
}

}  // namespace dsd
