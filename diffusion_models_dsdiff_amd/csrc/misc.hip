// misc.hip — the small kernels around the convolutions: timestep embedding, tiny linears, SE gate,
// skip-average / concat, layout changes, resampling, GEGLU.
#include "kernels.h"

namespace dsd {

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }

// timestep_embedding, ldm/modules/diffusionmodules/util.py:161-181: cat[cos(t*f), sin(t*f)], f_k = exp(-ln(1e4)*k/half)
// evaluated like the reference in fp32 (the fp32 argument is formed with fp32 ops, exp itself is taken in fp64 and rounded once).
__global__ void timestep_embedding_kernel(const void* __restrict__ t, int t_is_float, int N, int dim,
                                          const float* __restrict__ freqs, float* __restrict__ y) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half) return;
    const int n = i / half, k = i - n * half;
    const float tv = t_is_float ? ((const float*)t)[n] : (float)((const long long*)t)[n];
    const float neg_log = -9.210340371976184f;  // (float)(-math.log(10000))
    const float arg = (neg_log * (float)k) / (float)half;
    const float f = freqs ? freqs[k] : (float)exp((double)arg);   // host table (bit-identical to the caller's torch.exp) if given
    const float a = tv * f;
    y[(int64_t)n * dim + k] = cosf(a);
    y[(int64_t)n * dim + half + k] = sinf(a);
    if ((dim & 1) && k == 0) y[(int64_t)n * dim + dim - 1] = 0.f;
}

void timestep_embedding(const void* t, int t_is_float, int N, int dim, float* y, hipStream_t s, const float* freqs) {
    const int total = N * (dim / 2);
    if (total == 0) return;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t, t_is_float, N, dim, freqs, y);
    check_launch("timestep_embedding");
}

// y[n][o] = bias[o] + sum_k act(x[n][k]) * w[o][k].  One wave per output row o, 8 batch rows at a time.
template <int ACT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int N, int K, int ldx,
                                                     const float* __restrict__ w, const float* __restrict__ bias, int O,
                                                     float* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wr = w + (int64_t)o * K;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0);
    for (int n0 = 0; n0 < N; n0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (vec) {
            for (int k = lane * 4; k < K; k += 256) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (n0 + j < N) {
                        float4 v = *reinterpret_cast<const float4*>(x + (int64_t)(n0 + j) * ldx + k);
                        if (ACT == ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                        acc[j] = fmaf(v.x, w4.x, acc[j]);
                        acc[j] = fmaf(v.y, w4.y, acc[j]);
                        acc[j] = fmaf(v.z, w4.z, acc[j]);
                        acc[j] = fmaf(v.w, w4.w, acc[j]);
                    }
                }
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float wv = wr[k];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (n0 + j < N) {
                        float v = x[(int64_t)(n0 + j) * ldx + k];
                        if (ACT == ACT_SILU) v = silu_f(v);
                        acc[j] = fmaf(v, wv, acc[j]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && n0 + j < N) y[(int64_t)(n0 + j) * ldy + o] = v + (bias ? bias[o] : 0.f);
        }
    }
}

// The same for at most 16 batch rows whose (activated) x fits LDS — the conditioning GEMMs of the networks: [B, 1280] x all 68
// emb_layers of the U-Net, [B, 768] x the 12 x 6 adaLN modulations of DiT-B (55296 outputs).  The kernel above gives a wave ONE
// output row and lets it activate all of x again for it (N K SiLUs per output row: 256 us for DiT-B's conditioning at batch 16,
// 680 M SiLUs) and walks the batch in groups of 8 (each weight row read twice).  Here a workgroup activates x once into LDS and
// its four waves take 8 output rows each, all batch rows at a time: the weights stream through once.
template <int ACT>
__global__ __launch_bounds__(256) void linear_lds_kernel(const float* __restrict__ x, int N, int K, int ldx,
                                                         const float* __restrict__ w, const float* __restrict__ bias, int O,
                                                         float* __restrict__ y, int ldy) {
    extern __shared__ float xs[];   // [16][K], activated; rows >= N are zeros (the product loop has no row condition)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 4; i < 16 * K; i += 1024) {
        const int n = i / K, k = i - n * K;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            v = *reinterpret_cast<const float4*>(x + (int64_t)n * ldx + k);
            if (ACT == ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        }
        *reinterpret_cast<float4*>(xs + i) = v;
    }
    __syncthreads();
    // this wave's 8 output rows x 16 batch rows = 128 partial sums per lane (each lane took every 64th float4 of k)
    const int o0 = blockIdx.x * 32 + wave * 8;
    float acc[128];
#pragma unroll
    for (int v = 0; v < 128; ++v) acc[v] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        float4 w4[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)   // 8 independent 16-byte loads in flight (rows beyond O read row O - 1: never stored)
            w4[r] = *reinterpret_cast<const float4*>(w + (int64_t)min(o0 + r, O - 1) * K + k);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const float4 v = *reinterpret_cast<const float4*>(xs + n * K + k);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = acc[r * 16 + n];
                a = fmaf(v.x, w4[r].x, a);
                a = fmaf(v.y, w4[r].y, a);
                a = fmaf(v.z, w4[r].z, a);
                a = fmaf(v.w, w4[r].w, a);
                acc[r * 16 + n] = a;
            }
        }
    }
    // reduce-scatter over the 64 lanes: at distance d a lane keeps one half of its values and receives the other lanes' share of
    // that half — 64 + 32 + ... + 2 = 126 exchanges for 128 sums (an all-reduce per sum would be 768); lane L ends with the two
    // sums v = 2 L, 2 L + 1, i.e. output row o0 + L / 8, batch rows 2 (L % 8) and 2 (L % 8) + 1
    int cnt = 128;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const bool up = (lane & d) != 0;
        cnt >>= 1;
#pragma unroll
        for (int jx = 0; jx < 64; ++jx) {
            if (jx < cnt) {
                const float lo = acc[jx], hi = acc[jx + cnt];
                const float recv = __shfl_xor(up ? lo : hi, d);
                acc[jx] = (up ? hi : lo) + recv;
            }
        }
    }
    const int o = o0 + (lane >> 3), n0 = 2 * (lane & 7);
    if (o < O) {
        const float b = bias ? bias[o] : 0.f;
        if (n0 < N) y[(int64_t)n0 * ldy + o] = acc[0] + b;
        if (n0 + 1 < N) y[(int64_t)(n0 + 1) * ldy + o] = acc[1] + b;
    }
}

void linear(const float* x, int N, int K, int ldx, const float* w, const float* bias, int O, int act_in, float* y,
            int ldy, hipStream_t s) {
    if (N == 0 || O == 0) return;
    const size_t xs_bytes = (size_t)16 * K * sizeof(float);   // (rows beyond N are zero-filled)
    // (9..16 rows: with fewer the wave-per-row kernel wins — it reads each weight row once anyway and does not pay for 16 batch
    // rows per output; batch 1, the U-Net's conditioning: 79 us there, 412 us with the first version of this kernel)
    if (N > 8 && N <= 16 && O >= 512 && K % 4 == 0 && ldx % 4 == 0 && xs_bytes <= 96 * 1024) {
        static const hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_lds_kernel<ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_lds_kernel<ACT_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        DSD_CHECK(a0 == hipSuccess && a1 == hipSuccess, "hipFuncSetAttribute(linear_lds_kernel) failed");
        const dim3 grid(cdiv(O, 32)), block(256);
        if (act_in == ACT_SILU)
            hipLaunchKernelGGL(linear_lds_kernel<ACT_SILU>, grid, block, xs_bytes, s, x, N, K, ldx, w, bias, O, y, ldy);
        else
            hipLaunchKernelGGL(linear_lds_kernel<ACT_NONE>, grid, block, xs_bytes, s, x, N, K, ldx, w, bias, O, y, ldy);
        check_launch("linear_lds");
        return;
    }
    const dim3 grid(cdiv(O, 4)), block(256);
    if (act_in == ACT_SILU)
        hipLaunchKernelGGL(linear_kernel<ACT_SILU>, grid, block, 0, s, x, N, K, ldx, w, bias, O, y, ldy);
    else
        hipLaunchKernelGGL(linear_kernel<ACT_NONE>, grid, block, 0, s, x, N, K, ldx, w, bias, O, y, ldy);
    check_launch("linear");
}

// SE_Attention (Disc_diff/guided_diffusion/unet.py:82-109): y = x * sigmoid(W2 relu(W1 avgpool(x))). One block per sample.
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ x, int HW, int C,
                                                       const float* __restrict__ w1, const float* __restrict__ w2, int Cr,
                                                       float* __restrict__ y) {
    extern __shared__ float sm[];  // pooled[C], hid[Cr], gate[C]
    float* pooled = sm;
    float* hid = sm + C;
    float* gate = hid + Cr;
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (int64_t)n * HW * C;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += xb[(int64_t)p * C + c];
        pooled[c] = s / (float)HW;
    }
    __syncthreads();
    for (int j = tid; j < Cr; j += 256) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(w1[(int64_t)j * C + c], pooled[c], s);
        hid[j] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s = fmaf(w2[(int64_t)c * Cr + j], hid[j], s);
        gate[c] = 1.f / (1.f + expf(-s));
    }
    __syncthreads();
    float* yb = y + (int64_t)n * HW * C;
    for (int64_t i = tid; i < (int64_t)HW * C; i += 256) yb[i] = xb[i] * gate[i % C];
}

void se_scale(const float* x, int N, int HW, int C, const float* w1, const float* w2, int Cr, float* y, hipStream_t s) {
    if (N == 0) return;
    const size_t lds = (size_t)(2 * C + Cr) * sizeof(float);
    hipLaunchKernelGGL(se_scale_kernel, dim3(N), dim3(256), lds, s, x, HW, C, w1, w2, Cr, y);
    check_launch("se_scale");
}

// dst[p, coff:coff+C] = act((a [+b] [+c] [+d]) / div)   — skip average (model.py:745), mean(stack) (:722-725), concat slices
template <int ACT>
__global__ __launch_bounds__(256) void avg_into_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                       const float4* __restrict__ c, const float4* __restrict__ d,
                                                       float div, int64_t total4, int cols, float* __restrict__ dst,
                                                       int dstC, int coff, int64_t per_sample4, int bmask) {
    // bmask bit k set: source k holds ONE sample that is shared by (broadcast over) the whole batch
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t is = bmask ? i % per_sample4 : i;
        float4 v = a[(bmask & 1) ? is : i];
        if (b) { const float4 t = b[(bmask & 2) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (c) { const float4 t = c[(bmask & 4) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (d) { const float4 t = d[(bmask & 8) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (div != 1.f) { v.x /= div; v.y /= div; v.z /= div; v.w /= div; }
        if (ACT == ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        const int64_t p = i / cols;
        const int c4 = (int)(i - p * cols);
        *reinterpret_cast<float4*>(dst + p * dstC + coff + c4 * 4) = v;
    }
}

void avg_into(const float* a, const float* b, const float* c, const float* d, float div, int64_t pixels, int C, float* dst,
              int dstC, int coff, int act, hipStream_t s, int64_t per_sample_pixels, int bmask) {
    DSD_CHECK(C % 4 == 0 && dstC % 4 == 0 && coff % 4 == 0, "avg_into: channel counts must be multiples of 4");
    const int cols = C / 4;
    const int64_t total = pixels * cols;
    if (total == 0) return;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    if (act == ACT_SILU)
        hipLaunchKernelGGL(avg_into_kernel<ACT_SILU>, dim3(blocks), dim3(256), 0, s, (const float4*)a, (const float4*)b,
                           (const float4*)c, (const float4*)d, div, total, cols, dst, dstC, coff, per_sample_pixels * cols, bmask);
    else
        hipLaunchKernelGGL(avg_into_kernel<ACT_NONE>, dim3(blocks), dim3(256), 0, s, (const float4*)a, (const float4*)b,
                           (const float4*)c, (const float4*)d, div, total, cols, dst, dstC, coff, per_sample_pixels * cols, bmask);
    check_launch("avg_into");
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int C, int HW, int64_t total, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int p = (int)(r % HW);
        const int64_t n = r / HW;
        y[i] = x[(n * C + c) * HW + p];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int C, int HW, int64_t total, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t r = i / HW;
        const int c = (int)(r % C);
        const int64_t n = r / C;
        y[i] = x[(n * HW + p) * C + c];
    }
}
void nchw_to_nhwc(const float* x, int N, int C, int HW, float* y, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    if (!total) return;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, C, HW, total, y);
    check_launch("nchw_to_nhwc");
}
void nhwc_to_nchw(const float* x, int N, int C, int HW, float* y, hipStream_t s) {
    const int64_t total = (int64_t)N * C * HW;
    if (!total) return;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, C, HW, total, y);
    check_launch("nhwc_to_nchw");
}

// AvgPool2d(2,2) (Downsample without conv, openaimodel.py:160) / nearest x2 (Upsample without conv, :118), NHWC
__global__ void avgpool2_kernel(const float4* __restrict__ x, int H, int W, int cols, int64_t total4, float4* __restrict__ y) {
    const int OH = H / 2, OW = W / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % cols);
        int64_t r = i / cols;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const int64_t n = r / OH;
        const float4* b = x + ((n * H + 2 * oh) * W + 2 * ow) * cols + c4;
        const float4 v0 = b[0], v1 = b[cols], v2 = b[(int64_t)W * cols], v3 = b[(int64_t)W * cols + cols];
        float4 o;
        o.x = (v0.x + v1.x + v2.x + v3.x) * 0.25f;
        o.y = (v0.y + v1.y + v2.y + v3.y) * 0.25f;
        o.z = (v0.z + v1.z + v2.z + v3.z) * 0.25f;
        o.w = (v0.w + v1.w + v2.w + v3.w) * 0.25f;
        y[i] = o;
    }
}
__global__ void upsample2_kernel(const float4* __restrict__ x, int H, int W, int cols, int64_t total4, float4* __restrict__ y) {
    const int OH = H * 2, OW = W * 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % cols);
        int64_t r = i / cols;
        const int ow = (int)(r % OW); r /= OW;
        const int oh = (int)(r % OH);
        const int64_t n = r / OH;
        y[i] = x[((n * H + (oh >> 1)) * W + (ow >> 1)) * cols + c4];
    }
}
void avgpool2(const float* x, int N, int H, int W, int C, float* y, hipStream_t s) {
    DSD_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "avgpool2: bad shape");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    if (!total) return;
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, (const float4*)x, H, W, C / 4, total, (float4*)y);
    check_launch("avgpool2");
}
void upsample2(const float* x, int N, int H, int W, int C, float* y, hipStream_t s) {
    DSD_CHECK(C % 4 == 0, "upsample2: bad shape");
    const int64_t total = (int64_t)N * H * 2 * W * 2 * (C / 4);
    if (!total) return;
    hipLaunchKernelGGL(upsample2_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, (const float4*)x, H, W, C / 4, total, (float4*)y);
    check_launch("upsample2");
}

// GEGLU (ldm/modules/attention.py:47-55): y = a * gelu(gate), [a | gate] = x row halves; exact erf GELU.
__global__ void geglu_kernel(const float* __restrict__ x, int64_t rows, int inner, float* __restrict__ y) {
    const int64_t total = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / inner;
        const int c = (int)(i - r * inner);
        const float a = x[r * 2 * inner + c];
        const float g = x[r * 2 * inner + inner + c];
        y[i] = a * (0.5f * g * (1.f + erff(g * 0.70710678118654752440f)));
    }
}
void geglu(const float* x, int64_t rows, int inner, float* y, hipStream_t s) {
    const int64_t total = rows * inner;
    if (!total) return;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, rows, inner, y);
    check_launch("geglu");
}

// row softmax for the materialised score matrix of the VAE's single-head attention (model.py:196-198)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int cols, float scale) {
    __shared__ float red[4];
    float* row = s + (int64_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, row[c] * scale);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = tid; c < cols; c += 256) {
        const float e = expf(row[c] * scale - m);
        row[c] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.f / sum;
    for (int c = tid; c < cols; c += 256) row[c] *= inv;
}
void softmax_rows(float* s, int64_t rows, int cols, float scale, hipStream_t st) {
    if (rows == 0 || cols == 0) return;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols, scale);
    check_launch("softmax_rows");
}

// ------------------------------------------------------------------------------------------------ DiT (DiT_models.py)
// one wave per token row: two-pass LayerNorm (mean, then centred variance) in fp32, no affine, then adaLN modulate
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, int64_t rows, int T, int C,
                                                          const float* __restrict__ mod, int mod_stride, int shift_off,
                                                          int scale_off, float eps, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    const float* mr = mod + (row / T) * mod_stride;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        q = fmaf(d, d, q);
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / C + eps);
    for (int c = lane; c < C; c += 64)
        y[row * C + c] = ((xr[c] - mean) * rstd) * (1.f + mr[scale_off + c]) + mr[shift_off + c];
}
void ln_modulate(const float* x, int N, int T, int C, const float* mod, int mod_stride, int shift_off, int scale_off, float eps,
                 float* y, hipStream_t s) {
    const int64_t rows = (int64_t)N * T;
    if (rows == 0) return;
    hipLaunchKernelGGL(ln_modulate_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, T, C, mod, mod_stride, shift_off, scale_off,
                       eps, y);
    check_launch("ln_modulate");
}

__global__ void gated_residual_kernel(float4* __restrict__ x, const float4* __restrict__ y, int64_t total4, int64_t tc4, int c4,
                                      const float* __restrict__ mod, int mod_stride, int gate_off) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / tc4;
        const int c = (int)(i % c4) * 4;
        const float4 g = *reinterpret_cast<const float4*>(mod + n * mod_stride + gate_off + c);
        float4 a = x[i];
        const float4 b = y[i];
        a.x = fmaf(g.x, b.x, a.x); a.y = fmaf(g.y, b.y, a.y); a.z = fmaf(g.z, b.z, a.z); a.w = fmaf(g.w, b.w, a.w);
        x[i] = a;
    }
}
void gated_residual(float* x, const float* y, int N, int T, int C, const float* mod, int mod_stride, int gate_off, hipStream_t s) {
    DSD_CHECK(C % 4 == 0 && mod_stride % 4 == 0 && gate_off % 4 == 0, "gated_residual: widths must be multiples of 4");
    const int64_t total4 = (int64_t)N * T * C / 4;
    if (!total4) return;
    hipLaunchKernelGGL(gated_residual_kernel, dim3((unsigned)std::min<int64_t>((total4 + 255) / 256, 65535)), dim3(256), 0, s, (float4*)x,
                       (const float4*)y, total4, (int64_t)T * C / 4, C / 4, mod, mod_stride, gate_off);
    check_launch("gated_residual");
}

// torch's tanh approximation: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__global__ void gelu_tanh_kernel(float* __restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        const float inner = 0.7978845608028654f * (v + 0.044715f * v * v * v);
        x[i] = 0.5f * v * (1.f + tanhf(inner));
    }
}
void gelu_tanh(float* x, int64_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(gelu_tanh_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, x, n);
    check_launch("gelu_tanh");
}

__global__ void patchify_kernel(const float* __restrict__ x, int C, int H, int W, int p, int64_t total, float* __restrict__ y) {
    const int h = H / p, w = W / p, K = C * p * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % K);
        int64_t r = i / K;
        const int tw = (int)(r % w); r /= w;
        const int th = (int)(r % h);
        const int64_t n = r / h;
        const int pw = k % p, ph = (k / p) % p, c = k / (p * p);
        y[i] = x[((n * C + c) * H + th * p + ph) * W + tw * p + pw];
    }
}
void patchify(const float* x, int N, int C, int H, int W, int p, float* y, hipStream_t s) {
    const int64_t total = (int64_t)N * C * H * W;
    if (!total) return;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, C, H, W, p, total, y);
    check_launch("patchify");
}

__global__ void unpatchify_kernel(const float* __restrict__ x, int c, int h, int w, int p, int64_t total, float* __restrict__ y) {
    const int Hh = h * p, Ww = w * p;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {   // i over the output NCHW
        const int ow = (int)(i % Ww);
        int64_t r = i / Ww;
        const int oh = (int)(r % Hh); r /= Hh;
        const int ch = (int)(r % c);
        const int64_t n = r / c;
        const int th = oh / p, ph = oh - th * p, tw = ow / p, pw = ow - tw * p;
        y[i] = x[((n * h + th) * w + tw) * (int64_t)(p * p * c) + (ph * p + pw) * c + ch];
    }
}
void unpatchify(const float* x, int N, int c, int h, int w, int p, float* y, hipStream_t s) {
    const int64_t total = (int64_t)N * c * h * p * w * p;
    if (!total) return;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, c, h, w, p, total, y);
    check_launch("unpatchify");
}

__global__ void add_rows_broadcast_kernel(float* __restrict__ x, const float* __restrict__ pos, int64_t total, int64_t tc) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) x[i] += pos[i % tc];
}
void add_rows_broadcast(float* x, const float* pos, int N, int64_t TC, hipStream_t s) {
    const int64_t total = (int64_t)N * TC;
    if (!total) return;
    hipLaunchKernelGGL(add_rows_broadcast_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, x, pos, total, TC);
    check_launch("add_rows_broadcast");
}

__global__ void embed_add_kernel(const float* __restrict__ a, const float* __restrict__ table, const long long* __restrict__ idx, int N,
                                 int C, float* __restrict__ y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    y[i] = (a ? a[i] : 0.f) + table[idx[n] * C + c];
}
void embed_add(const float* a, const float* table, const long long* idx, int N, int C, float* y, hipStream_t s) {
    if (N * C == 0) return;
    hipLaunchKernelGGL(embed_add_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, s, a, table, idx, N, C, y);
    check_launch("embed_add");
}

__global__ void add2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = a[i] + b[i];
}
void add2(const float* a, const float* b, int64_t n, float* y, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, a, b, n, y);
    check_launch("add2");
}

}  // namespace dsd
