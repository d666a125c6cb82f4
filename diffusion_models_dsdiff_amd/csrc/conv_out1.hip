// The U-Net's last layer, out = Conv3x3(SiLU(GroupNorm32(h))) with ONE output channel (UNet_DS_Diff/model.py:511-515,751), as one
// memory-bound pass: 320 channels in, one out — 0.4 FLOP per input byte after the trick below, so the floor is reading h once
// (1.34 GB at batch 16 x 256^2), not the matrix pipes the implicit-GEMM kernel spent 1.7 ms on (one column of a 32-wide MFMA
// tile used) after a separate normalisation pass had read and re-written the tensor.
//
// For every INPUT pixel the nine dot products d_k = sum_c w[k][c] * silu(scale[c] * h[c] + shift[c]) are formed once (the
// activation once per element, not once per tap) and parked in LDS; an output pixel is then the sum of nine of them taken from
// its 3x3 neighbourhood.  A workgroup owns a 16 x 32 output tile (18 x 34 input pixels, 1.2x halo); 16 lanes share a pixel,
// each holding 4 consecutive channels of every 64 (16-byte loads, 256 contiguous bytes per pixel and instruction), the nine
// sums cross the 16 lanes in DPP adds (no LDS traffic), weights are broadcast reads from LDS.  fp32 FMAs throughout: this
// layer is computed in plain fp32 in every arithmetic mode.
#include "common.h"
#include "kernels.h"

namespace dsd {
namespace {

constexpr int O1_TH = 16, O1_TW = 32, O1_IH = O1_TH + 2, O1_IW = O1_TW + 2;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, o);
}
// sum over the 16 lanes of a DPP row, the result in all of them
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    return v;
}

__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// (3 workgroups per CU: without the register cap hipcc keeps all 9 x C / 16 weight registers of a lane resident — 370 VGPRs at
// C = 320, one wave per SIMD and nothing to hide the loads behind: 619 us at the headline size against the 270 us of the read)
template <int NJ>   // C = 64 * NJ
__global__ __launch_bounds__(256, 3) void conv_out1_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                        int tiles_x, int tiles_y) {
    constexpr int C = 64 * NJ;
    __shared__ float4 wl[9 * C / 4];
    __shared__ float dots[9][O1_IH][O1_IW + 1];
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int y0 = ty * O1_TH - 1, x0 = tx * O1_TW - 1;   // image coordinates of the tile's first input pixel
    for (int i = tid; i < 9 * C / 4; i += 256) wl[i] = reinterpret_cast<const float4*>(w)[i];
    float4 sc[NJ], sh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        sc[j] = *reinterpret_cast<const float4*>(scale + (int64_t)n * C + 64 * j + 4 * l16);
        sh[j] = *reinterpret_cast<const float4*>(shift + (int64_t)n * C + 64 * j + 4 * l16);
    }
    __syncthreads();
    const float* xn = x + (int64_t)n * H * W * C + 4 * l16;
    constexpr int NPIX = O1_IH * O1_IW, U = 2;   // U pixels per trip and 16-lane group: 2 * NJ loads in flight per lane
    for (int p0 = grp; p0 < NPIX; p0 += 16 * U) {
        float4 v[U][NJ];
        bool inb[U];
        int iy[U], ix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + 16 * u;
            iy[u] = p / O1_IW;
            ix[u] = p - iy[u] * O1_IW;
            const int gy = y0 + iy[u], gx = x0 + ix[u];
            inb[u] = p < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const float* px = xn + ((int64_t)(inb[u] ? gy : 0) * W + (inb[u] ? gx : 0)) * C;
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[u][j] = *reinterpret_cast<const float4*>(px + 64 * j);
        }
        int wlane = l16;                     // opaque per trip: the weight reads stay LDS reads inside the loop (hoisted, they
        asm volatile("" : "+v"(wlane));      // are 9 x C / 16 registers per lane and spill)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float4 a;
                a.x = silu_f(fmaf(v[u][j].x, sc[j].x, sh[j].x));
                a.y = silu_f(fmaf(v[u][j].y, sc[j].y, sh[j].y));
                a.z = silu_f(fmaf(v[u][j].z, sc[j].z, sh[j].z));
                a.w = silu_f(fmaf(v[u][j].w, sc[j].w, sh[j].w));
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float4 wk = wl[k * (C / 4) + 16 * j + wlane];
                    acc[k] = fmaf(a.x, wk.x, acc[k]);
                    acc[k] = fmaf(a.y, wk.y, acc[k]);
                    acc[k] = fmaf(a.z, wk.z, acc[k]);
                    acc[k] = fmaf(a.w, wk.w, acc[k]);
                }
            }
            float mine = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float t = row_sum16(acc[k]);
                if (l16 == k) mine = t;
            }
            // zero padding applies to the ACTIVATED tensor: a pixel outside the image contributes nothing
            if (l16 < 9 && p0 + 16 * u < NPIX) dots[l16][iy[u]][ix[u]] = inb[u] ? mine : 0.f;
        }
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
    for (int o = tid; o < O1_TH * O1_TW; o += 256) {
        const int oy = o / O1_TW, ox = o - oy * O1_TW;
        const int gy = y0 + 1 + oy, gx = x0 + 1 + ox;
        float r = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) r += dots[kh * 3 + kw][oy + kh][ox + kw];
        if (gy < H && gx < W) y[((int64_t)n * H + gy) * W + gx] = r + b0;
    }
}

}  // namespace

bool conv_out1_ok(int C, int cout, int ks, int stride) { return cout == 1 && ks == 3 && stride == 1 && C % 64 == 0 && C >= 64 && C <= 320; }

void conv_out1(const ConvOut1Args& a, hipStream_t s) {
    DSD_CHECK(conv_out1_ok(a.C, 1, 3, 1), "conv_out1: %d input channels unsupported (a multiple of 64 up to 320)", a.C);
    if (a.N == 0 || a.H == 0 || a.W == 0) return;
    const int tx = cdiv(a.W, O1_TW), ty = cdiv(a.H, O1_TH);
    const int64_t blocks = (int64_t)a.N * tx * ty;
    DSD_CHECK(blocks < (1ll << 31), "conv_out1: problem too large");
    const dim3 grid((unsigned)blocks), block(256);
    switch (a.C / 64) {
        case 1: hipLaunchKernelGGL(conv_out1_kernel<1>, grid, block, 0, s, a.x, a.scale, a.shift, a.w, a.bias, a.y, a.H, a.W, tx, ty); break;
        case 2: hipLaunchKernelGGL(conv_out1_kernel<2>, grid, block, 0, s, a.x, a.scale, a.shift, a.w, a.bias, a.y, a.H, a.W, tx, ty); break;
        case 3: hipLaunchKernelGGL(conv_out1_kernel<3>, grid, block, 0, s, a.x, a.scale, a.shift, a.w, a.bias, a.y, a.H, a.W, tx, ty); break;
        case 4: hipLaunchKernelGGL(conv_out1_kernel<4>, grid, block, 0, s, a.x, a.scale, a.shift, a.w, a.bias, a.y, a.H, a.W, tx, ty); break;
        default: hipLaunchKernelGGL(conv_out1_kernel<5>, grid, block, 0, s, a.x, a.scale, a.shift, a.w, a.bias, a.y, a.H, a.W, tx, ty); break;
    }
    check_launch("conv_out1");
}

}  // namespace dsd
