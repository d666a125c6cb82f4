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
//   * Q (pre-scaled) is split once into registers in the B-operand layout (lane = query, 8 consecutive d per k-step).
//   * K and V are split ONCE per workgroup while they are staged: K as three row-major bf16 planes [key][d] (A operand of
//     S^T = K Q^T), V TRANSPOSED as three planes [d][key] (A operand of O^T = V^T P^T).  The k index of the second product
//     is the key; the accumulator layout of S^T gives a lane the keys 4h + (r & 3) + 8 (r >> 2) of a 32-key sub-tile
//     (h = lane half, r = register), so k-step s takes registers 8s..8s+7 as they are, and the V^T planes store their keys in
//     the matching order  position(16 s + 8 g + 4 h + m) = 16 s + 8 h + 4 g + m  — P never leaves registers.
//   * LDS rows are padded by 16 B (row strides 80 / 144 / 208 / 272 B): ds_read_b128 fragment reads are conflict-free.
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));

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

template <int DT>
__global__ __launch_bounds__(256) void attention_split_kernel(AttnArgs a) {
    constexpr int KEYS = 64;            // keys per LDS stage = two 32-key sub-tiles
    constexpr int DP = DT * 32;         // padded head dim
    constexpr int NKS = DP / 16;        // k-steps of the QK^T product
    constexpr int KROW = DP * 2 + 16;   // bytes per key row of a K plane
    constexpr int VROW = KEYS * 2 + 16; // bytes per d row of a V^T plane
    constexpr int KPL = KEYS * KROW, VPL = DP * VROW;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[3 * KPL];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[3 * VPL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, half = lane >> 5;
    const int n = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + lrow;
    const bool q_ok = q < a.Tq;

    // Q pieces (B operand): this lane's query, k-step s covers d = 16 s + 8 half .. + 8, pre-scaled like the reference
    abf16x8 qf[NKS][3];
    {
        const float* qp = a.q + ((int64_t)n * a.Tq + (q_ok ? q : 0)) * a.ldq + (int64_t)head * a.q_hs;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                const int d = s * 16 + half * 8 + i;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q_ok && d < a.d) t = *reinterpret_cast<const float4*>(qp + d);
                v[i] = t.x * a.scale_q; v[i + 1] = t.y * a.scale_q; v[i + 2] = t.z * a.scale_q; v[i + 3] = t.w * a.scale_q;
            }
            a_split8(v, qf[s]);
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
    constexpr int C4 = DP / 4;
    constexpr int NLD = KEYS * C4 / 256;   // float4 (K, V) pairs each thread stages per tile
    static_assert(NLD * 256 == KEYS * C4, "the staging loop covers the tile exactly");
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
        for (int j = 0; j < NLD; ++j) {   // split once per workgroup; K row-major, V transposed with the keys in fragment order
            const int i = tid + j * 256;
            const int kr = i / C4, d = (i - kr * C4) * 4;
            float kv[4] = {kreg[j].x * a.scale_k, kreg[j].y * a.scale_k, kreg[j].z * a.scale_k, kreg[j].w * a.scale_k};
            float vv[4] = {vreg[j].x, vreg[j].y, vreg[j].z, vreg[j].w};
            // key kr = 32 sub + 16 s + 8 g + 4 h + m  ->  position 32 sub + 16 s + 8 h + 4 g + m
            const int pos = (kr & ~15) | ((kr & 4) << 1) | ((kr & 8) >> 1) | (kr & 3);
#pragma unroll
            for (int pq = 0; pq < 3; ++pq) {
                const unsigned k01 = a_pk(kv[0], kv[1]), k23 = a_pk(kv[2], kv[3]);
                *reinterpret_cast<uint2*>(Ks + pq * KPL + kr * KROW + d * 2) = make_uint2(k01, k23);
                const unsigned v01 = a_pk(vv[0], vv[1]), v23 = a_pk(vv[2], vv[3]);
                unsigned short* vp = reinterpret_cast<unsigned short*>(Vs + pq * VPL + pos * 2);
                vp[(d + 0) * (VROW / 2)] = (unsigned short)(v01 & 0xFFFFu);
                vp[(d + 1) * (VROW / 2)] = (unsigned short)(v01 >> 16);
                vp[(d + 2) * (VROW / 2)] = (unsigned short)(v23 & 0xFFFFu);
                vp[(d + 3) * (VROW / 2)] = (unsigned short)(v23 >> 16);
                if (pq < 2) {
                    kv[0] -= a_lo(k01); kv[1] -= a_hi(k01); kv[2] -= a_lo(k23); kv[3] -= a_hi(k23);
                    vv[0] -= a_lo(v01); vv[1] -= a_hi(v01); vv[2] -= a_lo(v23); vv[3] -= a_hi(v23);
                }
            }
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
            const unsigned char* kf = Ks + (sub * 32 + lrow) * KROW + half * 16;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                abf16x8 kp[3];
#pragma unroll
                for (int pq = 0; pq < 3; ++pq) kp[pq] = *reinterpret_cast<const abf16x8*>(kf + pq * KPL + s * 32);
                a_mfma6(kp, qf[s], sacc);
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
            // O^T[d][q] = corr * O^T + sum_key V[key][d] P[key][q];  k-step s2 = registers 8 s2 .. 8 s2 + 7 of P
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
                for (int r = 0; r < 16; ++r) o[t][r] *= corr;
                const unsigned char* vf = Vs + (t * 32 + lrow) * VROW + (sub * 32 + half * 8) * 2;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    abf16x8 vp[3];
#pragma unroll
                    for (int pq = 0; pq < 3; ++pq) vp[pq] = *reinterpret_cast<const abf16x8*>(vf + pq * VPL + s2 * 32);
                    a_mfma6(vp, pf[s2], o[t]);
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

void attention(const AttnArgs& a, hipStream_t s) {
    DSD_CHECK(a.d % 4 == 0 && a.d >= 4 && a.d <= 128, "attention: head dim %d unsupported (need multiple of 4, <=128)", a.d);
    DSD_CHECK(a.Tk >= 1 && a.Tq >= 1, "attention: empty sequence");
    DSD_CHECK(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.q_hs % 4 == 0 && a.k_hs % 4 == 0 && a.v_hs % 4 == 0,
              "attention: rows must be 16-byte aligned");
    const dim3 grid(cdiv(a.Tq, 128), a.heads, a.N), block(256);
    const int dt = cdiv(a.d, 32);
    if (a.split) {   // bf16x6 arithmetic (every mode but the exact-fp32 one)
        switch (dt) {
            case 1: hipLaunchKernelGGL(attention_split_kernel<1>, grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL(attention_split_kernel<2>, grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL(attention_split_kernel<3>, grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL(attention_split_kernel<4>, grid, block, 0, s, a); break;
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
