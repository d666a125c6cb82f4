// peak.hip — what the bf16 matrix pipes of THIS device sustain: bare v_mfma_f32_32x32x16_bf16 loops on random operands.
//
// The nominal 2.5 PFLOP/s is 256 CUs x 4 SIMDs x 1024 FLOP/clk at 2.4 GHz.  Under a dense MFMA stream the chip holds a much
// lower clock (MI355X_MICROARCH.md "DVFS give-back": 1.5-1.95 GHz on random data, device to device), so the rate a real
// kernel can be held against is the rate of a loop that does NOTHING but the MFMAs.  Two loops with the dominant
// convolution's accumulator tile (one wave per SIMD, 2 row blocks x 5 column tiles x 16 accumulators, six products per
// group as in conv_split.hip):
//   variant 0: every operand stays in registers — no LDS, no memory, no barrier: the upper bound
//   variant 1: the weight fragments are re-read from LDS in front of every unit as the convolution does (ds_read_b128,
//              conflict-free 80-byte rows), activations in registers: the bound of an MFMA + LDS skeleton
//   variant 2 / 3: the same two on ALL-ZERO operands (the clock the chip holds without data toggling)
//   variant 4 .. 7: the same four on the v_mfma_f32_16x16x32_bf16 shape (below)
// bench.py runs variants 0 / 1 / 4 / 5 beside the network and reports them as roofline.sustained; DESIGN.md §5.
#include "kernels.h"

namespace dsd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int PK_NT = 5, PK_RB = 2, PK_NP = 3;
static constexpr int PK_RS = 80;                       // LDS row stride (bytes) of a 32-k weight row, as in conv_split.hip
static constexpr int PK_PLANE = PK_NT * 32 * PK_RS;    // one piece of a 160-row weight tile
static constexpr int PK_STAGE = PK_NP * PK_PLANE;

// src: >= 256 * 64 + PK_STAGE bytes of operand bits (random bf16 or zeros); sink: one float per thread (keeps the
// accumulators alive); loops: k-tiles (2 k-steps x 5 units x 12 MFMAs per wave each)
template <bool LDS>
__global__ __launch_bounds__(256, 1) void mfma_peak_kernel(const unsigned char* __restrict__ src, float* __restrict__ sink, int loops) {
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * PK_STAGE + 4096];   // the convolution's LDS footprint: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int lrow = lane & 31, half = lane >> 5;
    for (int i = tid * 16; i < 2 * PK_STAGE; i += 256 * 16)
        *reinterpret_cast<u32x4*>(Bs + i) = *reinterpret_cast<const u32x4*>(src + 256 * 64 + (i % PK_STAGE));
    bf16x8 af[2][PK_RB][PK_NP], bw[PK_NP];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < PK_RB; ++r)
#pragma unroll
            for (int q = 0; q < PK_NP; ++q)
                af[s][r][q] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(tid * 4 + ((s * 6 + r * 3 + q) & 3)) * 16));
#pragma unroll
    for (int q = 0; q < PK_NP; ++q) bw[q] = *reinterpret_cast<const bf16x8*>(src + 256 * 64 + (size_t)(q * 64 + lane) * 16);
    f32x16 acc[PK_RB][PK_NT];
#pragma unroll
    for (int r = 0; r < PK_RB; ++r)
#pragma unroll
        for (int j = 0; j < PK_NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][j][e] = 0.f;
    __syncthreads();
    const unsigned char* frag = Bs + lrow * PK_RS + half * 16;
    for (int kt = 0; kt < loops; ++kt) {
        const unsigned char* bf = frag + (kt & 1) * PK_STAGE;
#pragma unroll
        for (int u = 0; u < 2 * PK_NT; ++u) {
            const int s = u / PK_NT, j = u % PK_NT;
            bf16x8 b[PK_NP];
#pragma unroll
            for (int q = 0; q < PK_NP; ++q)
                b[q] = LDS ? *reinterpret_cast<const bf16x8*>(bf + q * PK_PLANE + j * 32 * PK_RS + s * 32) : bw[q];
#pragma unroll
            for (int r = 0; r < PK_RB; ++r) {
                f32x16 c = acc[r][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][r][0], b[0], c, 0, 0, 0);
                acc[r][j] = c;
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < PK_RB; ++r)
#pragma unroll
        for (int j = 0; j < PK_NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[r][j][e];
    sink[(size_t)blockIdx.x * 256 + tid] = t;
}

// The same work on the v_mfma_f32_16x16x32_bf16 shape (MI355X_MICROARCH.md "DVFS give-back" item 7: at equal cycles per FLOP
// the chip can hold a higher clock on one shape than on the other): the wave's 64 x 160 output tile as 4 x 10 tiles of
// 16 x 16 (the same 160 accumulator registers), one 32-deep k-step per k-tile, six products per tile: 240 MFMAs of 16 cycles
// per k-tile = the 3840 matrix-pipe cycles of the 32x32x16 loop.  LDS variant: the 10 x 3 weight fragments of a k-tile come
// from LDS ([k chunk][row][16 B]: conflict-free ds_read_b128 for this operand pattern).
typedef float f32x4p __attribute__((ext_vector_type(4)));
template <bool LDS>
__global__ __launch_bounds__(256, 1) void mfma_peak16_kernel(const unsigned char* __restrict__ src, float* __restrict__ sink, int loops) {
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * PK_STAGE + 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid * 16; i < 2 * PK_STAGE; i += 256 * 16)
        *reinterpret_cast<u32x4*>(Bs + i) = *reinterpret_cast<const u32x4*>(src + 256 * 64 + (i % PK_STAGE));
    bf16x8 af[4][PK_NP], bw[PK_NP];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < PK_NP; ++q) af[r][q] = *reinterpret_cast<const bf16x8*>(src + ((size_t)(tid * 4 + ((r * 3 + q) & 3)) * 16));
#pragma unroll
    for (int q = 0; q < PK_NP; ++q) bw[q] = *reinterpret_cast<const bf16x8*>(src + 256 * 64 + (size_t)(q * 64 + lane) * 16);
    f32x4p acc[4][2 * PK_NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2 * PK_NT; ++j) acc[r][j] = f32x4p{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    constexpr int PLANE16 = PK_NT * 32 * 64;     // one piece of a 160-row x 32-k tile, [k chunk][row][16 B]
    const unsigned char* frag = Bs + (lane >> 4) * (PK_NT * 32 * 16) + (lane & 15) * 16;
    for (int kt = 0; kt < loops; ++kt) {
        const unsigned char* bf = frag + (kt & 1) * PK_STAGE;
#pragma unroll
        for (int j = 0; j < 2 * PK_NT; ++j) {
            bf16x8 b[PK_NP];
#pragma unroll
            for (int q = 0; q < PK_NP; ++q) b[q] = LDS ? *reinterpret_cast<const bf16x8*>(bf + q * PLANE16 + j * 256) : bw[q];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4p c = acc[r][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[r][0], b[0], c, 0, 0, 0);
                acc[r][j] = c;
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2 * PK_NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) t += acc[r][j][e];
    sink[(size_t)blockIdx.x * 256 + tid] = t;
}

__global__ void peak_fill_kernel(unsigned short* p, int n, int zero) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        // bf16 bit patterns of values in +-[0.5, 2): random sign, exponent 126..127 and significand (no inf / nan / denormal)
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const unsigned short v = (unsigned short)(((h & 1u) << 15) | ((126u + ((h >> 1) & 1u)) << 7) | ((h >> 2) & 0x7Fu));
        p[i] = zero ? (unsigned short)0 : v;
    }
}

int64_t mfma_peak_src_bytes() { return 256 * 64 + PK_STAGE; }

// FLOPs of one launch: workgroups x 4 waves x loops x 120 MFMAs x 32768
double mfma_peak_launch(int variant, const void* src, float* sink, int workgroups, int loops, hipStream_t s) {
    // variants 0..3: the 32x32x16 shape; 4..7: the same work on the 16x16x32 shape (bit 0: weights from LDS, bit 1: zeros)
    if (variant >= 4) {
        if (variant & 1)
            hipLaunchKernelGGL((mfma_peak16_kernel<true>), dim3(workgroups), dim3(256), 0, s, (const unsigned char*)src, sink, loops);
        else
            hipLaunchKernelGGL((mfma_peak16_kernel<false>), dim3(workgroups), dim3(256), 0, s, (const unsigned char*)src, sink, loops);
    } else if (variant & 1)
        hipLaunchKernelGGL((mfma_peak_kernel<true>), dim3(workgroups), dim3(256), 0, s, (const unsigned char*)src, sink, loops);
    else
        hipLaunchKernelGGL((mfma_peak_kernel<false>), dim3(workgroups), dim3(256), 0, s, (const unsigned char*)src, sink, loops);
    check_launch("mfma_peak");
    return (double)workgroups * 4.0 * loops * (2.0 * PK_NT * PK_RB * 6) * 32768.0;   // (240 x 16384 = 120 x 32768 per k-tile and wave)
}

void mfma_peak_fill(void* src, bool zero, hipStream_t s) {
    const int n = (int)(mfma_peak_src_bytes() / 2);
    hipLaunchKernelGGL(peak_fill_kernel, dim3(64), dim3(256), 0, s, (unsigned short*)src, n, zero ? 1 : 0);
    check_launch("peak_fill");
}

}  // namespace dsd
