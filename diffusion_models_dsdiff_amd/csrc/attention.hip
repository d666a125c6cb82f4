// attention.hip — fp32 softmax(Q K^T) V with online softmax on the gfx950 f32 matrix cores.
//
// Replaces QKVAttention / QKVAttentionLegacy (ldm/modules/diffusionmodules/openaimodel.py:496-555; the two
// einsums + fp32 softmax) and the einsum/softmax core of CrossAttention (ldm/modules/attention.py:164-193).
//
// One workgroup = (sample, head, 128 queries); each of the 4 waves owns 32 queries.  The score tile is
// computed TRANSPOSED, S^T = K Q^T (A = K rows from LDS, B = Q^T held in registers), so a lane owns ONE
// query column and 16 of the 32 keys of the tile: the softmax row reductions are in-lane plus a single
// exchange with lane^32 (wavefront-level, no LDS).  exp(S^T) is already in the B-operand layout of the
// second product O^T = V^T P^T, so P never leaves registers; V^T fragments are conflict-free ds_read_b32.
// The head dim is zero-padded to a multiple of 32 (DT tiles); keys beyond Tk are masked to -inf.
#include "kernels.h"

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DT>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    constexpr int KEYS = DT <= 2 ? 64 : 32;  // keys per LDS stage (32-key MFMA sub-tiles); keeps LDS <= 64 KB
    constexpr int DP = DT * 32;   // padded head dim
    constexpr int DH = DP / 2;    // per lane-half k range of the QK^T product
    constexpr int LS = DP + 4;    // LDS row stride (floats)
    __shared__ __attribute__((aligned(16))) float Ks[KEYS * LS];
    __shared__ __attribute__((aligned(16))) float Vs[KEYS * LS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    const int n = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool q_ok = q < a.Tq;

    // Q^T fragment (B operand): this lane's query, d in [half*DH, half*DH+DH), pre-scaled like the reference
    float qf[DH];
    {
        const float* qp = a.q + ((int64_t)n * a.Tq + (q_ok ? q : 0)) * a.ldq + (int64_t)head * a.q_hs;
#pragma unroll
        for (int i = 0; i < DH; i += 4) {
            const int d = half * DH + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q_ok && d < a.d) v = *reinterpret_cast<const float4*>(qp + d);
            qf[i] = v.x * a.scale_q;
            qf[i + 1] = v.y * a.scale_q;
            qf[i + 2] = v.z * a.scale_q;
            qf[i + 3] = v.w * a.scale_q;
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float* kbase = a.k + (int64_t)n * a.Tk * a.ldk + (int64_t)head * a.k_hs;
    const float* vbase = a.v + (int64_t)n * a.Tk * a.ldv + (int64_t)head * a.v_hs;
    constexpr int C4 = DP / 4;  // float4 columns per row
    constexpr int NLD = KEYS * C4 / 256;   // float4 (K, V) pairs each thread stages per tile
    static_assert(NLD * 256 == KEYS * C4, "the staging loop covers the tile exactly");
    // The next K / V tile is fetched into registers while the current one is multiplied (the loads used to sit, fully
    // exposed, between the two barriers of every tile).
    float4 kreg[NLD], vreg[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int kr = i / C4, c4 = i - kr * C4;
            const int key = k0 + kr, d = c4 * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (key < a.Tk && d < a.d) {
                kv = *reinterpret_cast<const float4*>(kbase + (int64_t)key * a.ldk + d);
                vv = *reinterpret_cast<const float4*>(vbase + (int64_t)key * a.ldv + d);
            }
            kreg[j] = kv;
            vreg[j] = vv;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < a.Tk; k0 += KEYS) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int kr = i / C4, d = (i - kr * C4) * 4;
            float4 kv = kreg[j];
            kv.x *= a.scale_k; kv.y *= a.scale_k; kv.z *= a.scale_k; kv.w *= a.scale_k;
            *reinterpret_cast<float4*>(Ks + kr * LS + d) = kv;
            *reinterpret_cast<float4*>(Vs + kr * LS + d) = vreg[j];
        }
        __syncthreads();
        if (k0 + KEYS < a.Tk) fetch(k0 + KEYS);
#pragma unroll
        for (int sub = 0; sub < KEYS / 32; ++sub) {
            if (k0 + sub * 32 >= a.Tk) break;
            // S^T[key][q] = sum_d K[key][d] Q[q][d]
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float* kf = Ks + (sub * 32 + lrow) * LS + half * DH;
#pragma unroll
            for (int i = 0; i < DH; i += 4) {
                const float4 k4 = *reinterpret_cast<const float4*>(kf + i);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qf[i], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qf[i + 1], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qf[i + 2], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qf[i + 3], sacc, 0, 0, 0);
            }
            // mask + online softmax; this lane holds keys (r&3)+8*(r>>2)+4*half of the sub-tile for query lrow
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float sv = sacc[r] * a.scale_s;
                sv = key < a.Tk ? sv : -INFINITY;
                sacc[r] = sv;
                tmax = fmaxf(tmax, sv);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            const float corr = expf(m_run - m_new);  // m_run = -inf on the first tile -> 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = expf(sacc[r] - m_new);
                sacc[r] = pv;
                psum += pv;
            }
            psum += __shfl_xor(psum, 32);
            l_run = l_run * corr + psum;
            m_run = m_new;
            // O^T[d][q] = corr * O^T + sum_key V[key][d] P[key][q]
#pragma unroll
            for (int t = 0; t < DT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= corr;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int kr = sub * 32 + (s & 3) + 8 * (s >> 2) + 4 * half;
                    const float vf = Vs[kr * LS + t * 32 + lrow];
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sacc[s], o[t], 0, 0, 0);
                }
            }
        }
    }
    if (!q_ok) return;
    const float inv = 1.f / l_run;
    float* op = a.out + ((int64_t)n * a.Tq + q) * a.ldo + (int64_t)head * a.d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (d < a.d) op[d] = o[t][r] * inv;
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The same flash attention on the bf16 matrix cores with the fp32 operands split EXACTLY into three bf16 pieces (the
// bf16x6 arithmetic of conv_split.hip: a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1, fp32 accumulation, dropped terms <= 2^-24):
// 24 v_mfma_f32_32x32x16_bf16 (768 matrix-pipe cycles) per 32 keys at d = 32 instead of 32 v_mfma_f32_32x32x2_f32 (2048).
//   * Q (pre-scaled, base-2 logits) is split once into registers in the B-operand layout (lane = query, 8 consecutive d per
//     k-step).
//   * K and V are split ONCE per workgroup while they are staged, 8 floats -> three 16-byte bf16 chunks, into the LDS image
//     of attention16.hip, one image per piece: K k-step-major ([k-step][key][32 B], halves XOR-swizzled by (key >> 3) & 1:
//     conflict-free ds_read_b128 A-operand reads), V d-tile-major ([d tile][key][64 B]) and read TRANSPOSED by
//     ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group, 256 contiguous bytes per 32 lanes: conflict-free).  [r3] This
//     replaces a V^T image filled by 2-byte scatter stores, which conflicted on 48 % of the kernel's LDS cycles
//     (profiles/r02_pmc.json); every staging store is now a 16-byte ds_write_b128.
//   * The accumulator layout of S^T gives a lane the keys 4h + (r & 3) + 8 (r >> 2) of a 32-key sub-tile (h = lane half,
//     r = register); k-step s of the second product takes registers 8s..8s+7 as they are and the transposed V reads fetch
//     their keys in the matching order — P never leaves registers.
//   * Softmax as in attention16.hip: base-2 logits, the negated running maximum is the C operand of the first S^T MFMA
//     (p = exp2(acc), one v_exp_f32), the maximum only moves when a score exceeds it by more than 2^8 (wave-uniform branch;
//     O, l and the pending tile are rescaled together before the tile is exponentiated).
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));
typedef short as16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned a_pk(float x, float y) {
    abf16x2 t;
    t[0] = (__bf16)x;
    t[1] = (__bf16)y;
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float a_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float a_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
// 8 fp32 -> three bf16x8 pieces
__device__ __forceinline__ void a_split8(const float (&v0)[8], abf16x8 (&out)[3]) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v0[i];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        au32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pk = a_pk(v[2 * i], v[2 * i + 1]);
            w[i] = pk;
            if (q < 2) {
                v[2 * i] -= a_lo(pk);
                v[2 * i + 1] -= a_hi(pk);
            }
        }
        out[q] = __builtin_bit_cast(abf16x8, w);
    }
}
__device__ __forceinline__ void a_mfma6(const abf16x8 (&a)[3], const abf16x8 (&b)[3], f32x16& c) {   // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
}

template <int NKS>
__global__ __launch_bounds__(256, NKS <= 4 ? 2 : 1) void attention_split_kernel(AttnArgs a) {
    constexpr int KEYS = 64;                 // keys per LDS stage = two 32-key sub-tiles
    constexpr int DT = (NKS + 1) / 2;        // 32-wide d tiles of O^T
    constexpr int CH = 2 * NKS;              // 8-float chunk slots per key row
    constexpr int KPL = KEYS * 32 + (NKS == 2 ? 64 : NKS == 3 ? 96 : 32);      // bytes per (piece, k-step) plane of K (+32: the planes of a key row land on different banks)
    constexpr int VPL = KEYS * 64 + 64;      // bytes per (piece, d-tile) plane of V
    constexpr int KPC = NKS * KPL, VPC = DT * VPL;   // one piece
    constexpr int NLD = (KEYS * CH + 255) / 256;
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * KPC + 3 * VPC];
    unsigned char* Kl = lds;
    unsigned char* Vl = lds + 3 * KPC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    const int n = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool q_ok = q < a.Tq;
    const int hch = (a.d + 7) >> 3;          // chunk slots that hold data (head dim % 4 == 0: a chunk may be half valid)

    for (int i = tid * 16; i < 3 * KPC + 3 * VPC; i += 256 * 16) *reinterpret_cast<au32x4*>(lds + i) = au32x4{0u, 0u, 0u, 0u};

    // Q pieces (B operand): this lane's query, k-step s covers d = 16 s + 8 half .. + 8; scores come out as base-2 logits
    abf16x8 qf[NKS][3];
    {
        const float qs = a.scale_q * a.scale_s * LOG2E;
        const float* qp = a.q + ((int64_t)n * a.Tq + (q_ok ? q : 0)) * a.ldq + (int64_t)head * a.q_hs;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                const int d = s * 16 + half * 8 + i;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q_ok && d < a.d) t = *reinterpret_cast<const float4*>(qp + d);
                v[i] = t.x * qs; v[i + 1] = t.y * qs; v[i + 2] = t.z * qs; v[i + 3] = t.w * qs;
            }
            a_split8(v, qf[s]);
        }
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    float m_run = 0.f, l_run = 0.f;   // l_run: this lane's 16 keys of every sub-tile (the halves are added at the end)

    const float* kbase = a.k + (int64_t)n * a.Tk * a.ldk + (int64_t)head * a.k_hs;
    const float* vbase = a.v + (int64_t)n * a.Tk * a.ldv + (int64_t)head * a.v_hs;
    float4 kreg[NLD][2], vreg[NLD][2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * 256;
            const int kr = i / CH, c = i - kr * CH;
            const int key = k0 + kr;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = c * 8 + e * 4;
                float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
                if (kr < KEYS && key < a.Tk && d < a.d) {
                    kv = *reinterpret_cast<const float4*>(kbase + (int64_t)key * a.ldk + d);
                    vv = *reinterpret_cast<const float4*>(vbase + (int64_t)key * a.ldv + d);
                }
                kreg[j][e] = kv;
                vreg[j][e] = vv;
            }
        }
    };
    const int kofs = (lrow * 32 + half * 16) ^ (((lrow >> 3) & 1) << 4);                      // + pq * KPC + s * KPL + sub * 1024
    const int g = lane >> 4, li = lane & 15;
    const int vofs = (4 * half + (li >> 2)) * 64 + (16 * (g & 1) + 4 * (li & 3)) * 2;         // + pq * VPC + t * VPL + sub * 2048 + s2 * 1024 + j * 512
    constexpr float THR = 8.f;

    fetch(0);
    bool first = true;
    for (int k0 = 0; k0 < a.Tk; k0 += KEYS) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NLD; ++j) {   // split once per workgroup: 8 floats -> three 16-byte chunks each for K and V
            const int i = tid + j * 256;
            const int kr = i / CH, c = i - kr * CH;
            if (kr < KEYS && c < hch) {
                const float kv[8] = {kreg[j][0].x * a.scale_k, kreg[j][0].y * a.scale_k, kreg[j][0].z * a.scale_k, kreg[j][0].w * a.scale_k,
                                     kreg[j][1].x * a.scale_k, kreg[j][1].y * a.scale_k, kreg[j][1].z * a.scale_k, kreg[j][1].w * a.scale_k};
                const float vv[8] = {vreg[j][0].x, vreg[j][0].y, vreg[j][0].z, vreg[j][0].w, vreg[j][1].x, vreg[j][1].y, vreg[j][1].z, vreg[j][1].w};
                abf16x8 kp[3], vp[3];
                a_split8(kv, kp);
                a_split8(vv, vp);
                const int ko = (c >> 1) * KPL + ((kr * 32 + (c & 1) * 16) ^ (((kr >> 3) & 1) << 4));
                const int vo = (c >> 2) * VPL + kr * 64 + (c & 3) * 16;
#pragma unroll
                for (int pq = 0; pq < 3; ++pq) {
                    *reinterpret_cast<abf16x8*>(Kl + pq * KPC + ko) = kp[pq];
                    *reinterpret_cast<abf16x8*>(Vl + pq * VPC + vo) = vp[pq];
                }
            }
        }
        __syncthreads();
        if (k0 + KEYS < a.Tk) fetch(k0 + KEYS);
#pragma unroll
        for (int sub = 0; sub < KEYS / 32; ++sub) {
            if (k0 + sub * 32 >= a.Tk) break;
            // S^T[key][q] - m = sum_d K[key][d] Q[q][d] + (-m)
            f32x16 sacc = negm;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                abf16x8 kp[3];
#pragma unroll
                for (int pq = 0; pq < 3; ++pq) kp[pq] = *reinterpret_cast<const abf16x8*>(Kl + pq * KPC + s * KPL + sub * 1024 + kofs);
                a_mfma6(kp, qf[s], sacc);
            }
            if (k0 + sub * 32 + 32 > a.Tk) {   // last, partial sub-tile: keys beyond Tk take no part
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    sacc[r] = key < a.Tk ? sacc[r] : -INFINITY;
                }
            }
            float tmax = fmaxf(fmaxf(sacc[0], sacc[1]), sacc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, sacc[r]), sacc[r + 1]);
            tmax = fmaxf(tmax, sacc[15]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const bool need = first || !(tmax <= THR);
            if (__any(need)) {
                const float delta = need ? tmax : 0.f;
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
            // O^T[d][q] += sum_key V[key][d] P[key][q];  k-step s2 = registers 8 s2 .. 8 s2 + 7 of P, split exactly into 3 pieces
            abf16x8 pf[2][3];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sacc[8 * s2 + e];
                a_split8(v, pf[s2]);
            }
#pragma unroll
            for (int t = 0; t < DT; ++t) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    abf16x8 vp[3];
#pragma unroll
                    for (int pq = 0; pq < 3; ++pq) {
                        const unsigned char* vq = Vl + pq * VPC + t * VPL + sub * 2048 + s2 * 1024 + vofs;
                        const as16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4*)(vq));
                        const as16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4*)(vq + 512));
                        vp[pq] = __builtin_bit_cast(abf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    a_mfma6(vp, pf[s2], o[t]);
                }
            }
        }
    }
    if (!q_ok) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    float* op = a.out + ((int64_t)n * a.Tq + q) * a.ldo + (int64_t)head * a.d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = t * 32 + 8 * r4 + 4 * half;
            if (d < a.d)   // head dim % 4 == 0 and d % 4 == 0: the four values are valid together
                *reinterpret_cast<float4*>(op + d) = make_float4(o[t][4 * r4] * inv, o[t][4 * r4 + 1] * inv, o[t][4 * r4 + 2] * inv, o[t][4 * r4 + 3] * inv);
        }
}

void attention(const AttnArgs& a, hipStream_t s) {
    DSD_CHECK(a.d % 4 == 0 && a.d >= 4 && a.d <= 128, "attention: head dim %d unsupported (need multiple of 4, <=128)", a.d);
    DSD_CHECK(a.Tk >= 1 && a.Tq >= 1, "attention: empty sequence");
    DSD_CHECK(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.q_hs % 4 == 0 && a.k_hs % 4 == 0 && a.v_hs % 4 == 0,
              "attention: rows must be 16-byte aligned");
    const dim3 grid(cdiv(a.Tq, 128), a.heads, a.N), block(256);
    const int dt = cdiv(a.d, 32);
    if (a.split) {   // bf16x6 arithmetic (every mode but the exact-fp32 one)
        DSD_CHECK(a.ldo % 4 == 0, "attention: output rows must be 16-byte aligned");
        switch (cdiv(a.d, 16)) {
            case 1: hipLaunchKernelGGL(attention_split_kernel<1>, grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL(attention_split_kernel<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(attention_split_kernel<3>, grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL(attention_split_kernel<4>, grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL(attention_split_kernel<5>, grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL(attention_split_kernel<6>, grid, block, 0, s, a); break;
            case 7: hipLaunchKernelGGL(attention_split_kernel<7>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(attention_split_kernel<8>, grid, block, 0, s, a); break;
        }
        check_launch("attention_split");
        return;
    }
    switch (dt) {
        case 1: hipLaunchKernelGGL(attention_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(attention_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(attention_kernel<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(attention_kernel<4>, grid, block, 0, s, a); break;
    }
    check_launch("attention");
}

}  // namespace dsd
