// sampler.hip — the per-step sampler update as ONE fused elementwise kernel (+ on-device Philox noise).
//
// Mode A  = guided-diffusion: p_mean_variance / p_sample / ddim_sample
//           (Disc_diff/guided_diffusion/gaussian_diffusion.py:244-350, 422-465, 618-665)
// Mode B  = LDM: DDPMModel.p_sample / p_mean_variance (trainers/trainer_ddpm.py:461-482, with
//           ldm/models/diffusion/ddpm.py:284-311) and DDIMSampler.p_sample_ddim (ldm/models/diffusion/ddim.py:187-261).
// The arithmetic is written in the reference's fp32 operation order; coefficient tables come from the host
// (float64 there, rounded to fp32 once, like _extract_into_tensor(...).float() / the registered fp32 buffers).
#include "kernels.h"
#include "../../include/dsdiff.h"

// The reference evaluates these updates as separate fp32 torch ops, so nothing in this file may be contracted into an
// FMA: plain operators under contract(off) (HIP's __fmul_rn/__fsub_rn are header inlines that stay contractable);
// fmaf() is written where torch itself fuses.
#pragma clang fp contract(off)

namespace dsd {

// ---- Philox4x32-10 (Salmon et al. 2011): counter = (idx, step), key = seed
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                               uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// element i of the normal stream for (seed, step): Box-Muller on the Philox block i/4... each block yields 4 normals
__device__ __forceinline__ float philox_normal_at(int64_t i, uint64_t seed, uint64_t step) {
    uint32_t r[4];
    const uint64_t blk = (uint64_t)i >> 2;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const int lane = (int)(i & 3);
    const uint32_t a = r[lane & 2], b = r[(lane & 2) + 1];
    const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float u2 = (float)b * 2.3283064365386963e-10f;           // [0,1)
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    return (lane & 1) ? rad * sinf(ang) : rad * cosf(ang);
}

__global__ void philox_normal_kernel(float* __restrict__ y, int64_t n, uint64_t seed, uint64_t step) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = philox_normal_at(i, seed, step);
}

void philox_normal(float* y, int64_t n, uint64_t seed, uint64_t step, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, y, n, seed, step);
    check_launch("philox_normal");
}

__global__ __launch_bounds__(256) void sampler_update_kernel(StepCoef sc, const float* __restrict__ mo,
                                                             float* __restrict__ x, const float* __restrict__ noise,
                                                             uint64_t seed, uint64_t step, int B, int HW,
                                                             float* __restrict__ x0_out, const int64_t* __restrict__ slice_ids) {
    const int64_t total = (int64_t)B * HW;
    const int Cm = sc.learned_range ? 2 : 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW;
        const int64_t p = i - b * HW;
        const float out = mo[(b * Cm) * HW + p];
        const float xt = x[i];
        const float z = noise ? noise[i] : philox_normal_at(slice_ids ? slice_ids[b] * HW + p : i, seed, step);
        const float* c = sc.c;
        float x0, res;
        if (sc.mode == DSD_MODE_B_DDIM) {
            // ddim.py:222-260
            float e_t;
            if (sc.pred == DSD_PRED_V) {
                e_t = c[0] * out + c[1] * xt;   // predict_eps_from_z_and_v ddpm.py:298-302
                x0 = c[0] * xt - c[1] * out;    // predict_start_from_z_and_v ddpm.py:290-296
            } else {
                e_t = out;
                x0 = (xt - c[7] * e_t) / sqrtf(c[4]);
            }
            if (sc.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            const float dir = sqrtf(1.f - c[5] - c[6] * c[6]) * e_t;
            res = sqrtf(c[5]) * x0 + dir + c[6] * z;
        } else {
            if (sc.pred == DSD_PRED_V)
                x0 = c[0] * xt - c[1] * out;
            else if (sc.pred == DSD_PRED_EPS)
                x0 = c[2] * xt - c[3] * out;
            else
                x0 = out;
            if (sc.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
            const float nz = sc.nonzero ? 1.f : 0.f;
            if (sc.mode == DSD_MODE_A_DDIM) {
                // gaussian_diffusion.py:646-664
                const float eps = (c[2] * xt - x0) / c[3];
                const float ab = c[4], abp = c[5];
                const float sigma = sc.eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
                const float mean_pred = x0 * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * eps;
                res = mean_pred + nz * sigma * z;
            } else {
                // DDPM: q_posterior mean + exp(0.5 logvar) z   (gaussian_diffusion.py:220-223,464; trainer_ddpm.py:467)
                const float mean = c[4] * x0 + c[5] * xt;
                float logvar = c[6];
                if (sc.learned_range) {  // gaussian_diffusion.py:287-293
                    const float v = mo[(b * Cm + 1) * HW + p];
                    const float frac = (v + 1.f) / 2.f;
                    logvar = frac * c[7] + (1.f - frac) * c[6];
                }
                res = mean + nz * expf(0.5f * logvar) * z;
            }
        }
        x[i] = res;
        if (x0_out) x0_out[i] = x0;
    }
}

void sampler_update(const StepCoef& sc, const float* model_out, float* x, const float* noise, uint64_t seed,
                    uint64_t step, int B, int HW, hipStream_t s, float* x0_out, const int64_t* slice_ids) {
    const int64_t total = (int64_t)B * HW;
    if (!total) return;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(sampler_update_kernel, dim3(blocks), dim3(256), 0, s, sc, model_out, x, noise, seed, step, B, HW, x0_out, slice_ids);
    check_launch("sampler_update");
}

// ------------------------------------------------------------------------------------------------------------------
// DPM-Solver(++) multistep (Disc_diff/guided_diffusion/sampler.py; twin ldm/models/diffusion/dpm_solver_new/
// dpm_solver_pytorch.py).  Per network evaluation k:
//   dpm_model   m_k = data prediction x0 (dpmsolver++, sampler.py:396-405) or noise prediction (:390-394) from the
//               network output, for model types noise / x_start / v (model_wrapper :247-265)
//   dpm_quantile  s_b = max(quantile_0.995(|x0_b|), max_val) per sample (dynamic thresholding :379-388, torch.quantile
//               'linear': sorted[floor(r)] lerp sorted[ceil(r)], r = q*(n-1) in fp32) by an exact 4-pass radix select
//   dpm_update  m_k <- clamp(m_k,-s,s)/s ; x <- first-order (:509-553) or second-order multistep update (:760-816)
// Every product/sum is a separately rounded fp32 op in the reference's order (file-wide contract(off)).
__global__ __launch_bounds__(256) void dpm_model_kernel(DpmCoef c, const float* __restrict__ mo, int Cm,
                                                        const float* __restrict__ x, float* __restrict__ m, int B,
                                                        int HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW;
        const float out = mo[(b * Cm) * HW + (i - b * HW)];   // a learned-sigma model: first channel only (gaussian_diffusion.py:484-485)
        const float xt = x[i];
        float eps;
        if (c.pred == DSD_PRED_EPS)
            eps = out;
        else if (c.pred == DSD_PRED_X0)
            eps = (xt - c.alpha * out) / c.sigma;
        else
            eps = c.alpha * out + c.sigma * xt;
        m[i] = c.data_pred ? (xt - c.sigma * eps) / c.alpha : eps;
    }
}

// One block per sample.  |v| bit patterns of non-negative floats order like unsigned integers, so the element of rank k
// is found digit by digit (8 bits per pass) with an LDS histogram; the rank k+1 element is either the same value or the
// smallest value above it.
__global__ __launch_bounds__(1024) void dpm_quantile_kernel(const float* __restrict__ m, int n, float ratio, float max_val,
                                                            float* __restrict__ s_out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_prefix, sh_k, sh_cnt, sh_min;
    const float* v = m + (int64_t)blockIdx.x * n;
    const float rank = ratio * (float)(n - 1);
    const float fl = floorf(rank);
    const uint32_t k_below = (uint32_t)fl;
    const bool need_above = ceilf(rank) != fl;
    const float w = rank - fl;
    if (threadIdx.x == 0) { sh_prefix = 0; sh_k = k_below; sh_min = 0xFFFFFFFFu; }
    uint32_t mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = sh_prefix;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const uint32_t u = __float_as_uint(v[i]) & 0x7FFFFFFFu;
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t k = sh_k, d = 0;
            while (d < 255 && k >= hist[d]) { k -= hist[d]; ++d; }
            sh_k = k;
            sh_cnt = hist[d];
            sh_prefix = prefix | (d << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const uint32_t ubelow = sh_prefix;
    uint32_t uabove = ubelow;
    if (need_above && sh_k + 1 >= sh_cnt) {      // rank k+1 lies beyond the run of equal values
        uint32_t loc = 0xFFFFFFFFu;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const uint32_t u = __float_as_uint(v[i]) & 0x7FFFFFFFu;
            if (u > ubelow && u < loc) loc = u;
        }
        atomicMin(&sh_min, loc);
        __syncthreads();
        uabove = sh_min;
    }
    if (threadIdx.x == 0) {
        const float a = __uint_as_float(ubelow), b = __uint_as_float(uabove);
        // at::lerp (ATen/native/Lerp.h, vectorised CPU form): fma(w<0.5 ? w : w-1, b-a, w<0.5 ? a : b)
        const float diff = b - a;
        const float q = (fabsf(w) < 0.5f) ? fmaf(w, diff, a) : fmaf(w - 1.0f, diff, b);
        s_out[blockIdx.x] = fmaxf(q, max_val);
    }
}

__global__ __launch_bounds__(256) void dpm_update_kernel(DpmCoef c, float* __restrict__ m0, const float* __restrict__ m1,
                                                         const float* __restrict__ s_thr, float* __restrict__ x, int B,
                                                         int HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float m = m0[i];
        if (s_thr) {
            const float s = s_thr[i / HW];
            m = fminf(fmaxf(m, -s), s) / s;
            m0[i] = m;
        }
        if (!x) continue;                                     // thresholding only (dsd_op_dpm_threshold)
        float res;
        if (c.order == 0) {
            res = m;                                          // denoise_to_zero_fn :503-507
        } else {
            res = c.cx * x[i] - c.cm * m;
            if (c.order == 2) res = res - c.cd * (c.ir0 * (m - m1[i]));
        }
        x[i] = res;
    }
}

void dpm_step(const DpmCoef& c, const float* model_out, int Cm, float* x, float* m_cur, const float* m_prev, float* s_buf,
              float ratio, float max_val, int B, int HW, hipStream_t s) {
    const int64_t total = (int64_t)B * HW;
    if (!total) return;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(dpm_model_kernel, dim3(blocks), dim3(256), 0, s, c, model_out, Cm, x, m_cur, B, HW);
    check_launch("dpm_model");
    const bool thr = c.thresh && c.data_pred;
    if (thr) {
        hipLaunchKernelGGL(dpm_quantile_kernel, dim3(B), dim3(1024), 0, s, m_cur, HW, ratio, max_val, s_buf);
        check_launch("dpm_quantile");
    }
    hipLaunchKernelGGL(dpm_update_kernel, dim3(blocks), dim3(256), 0, s, c, m_cur, m_prev, thr ? s_buf : nullptr, x, B, HW);
    check_launch("dpm_update");
}

void dpm_threshold(const float* x0, float* y, float* s_buf, float ratio, float max_val, int B, int n, hipStream_t s) {
    if (!B || !n) return;
    DSD_HIP(hipMemcpyAsync(y, x0, (size_t)B * n * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(dpm_quantile_kernel, dim3(B), dim3(1024), 0, s, y, n, ratio, max_val, s_buf);
    check_launch("dpm_quantile");
    DpmCoef c{};
    c.order = 0;
    hipLaunchKernelGGL(dpm_update_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)B * n + 255) / 256, 4096)), dim3(256), 0, s, c,
                       y, (const float*)nullptr, s_buf, (float*)nullptr, B, n);
    check_launch("dpm_update");
    DSD_HIP(hipStreamSynchronize(s));
}

__global__ void gaussian_sample_kernel(const float* __restrict__ mo, const float* __restrict__ noise, uint64_t seed, int B, int E,
                                       int HW, float* __restrict__ z) {
    const int64_t total = (int64_t)B * E * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / ((int64_t)E * HW);
        const int64_t r = i - b * E * HW;                       // (e, p) inside the sample
        const float mean = mo[b * 2 * E * HW + r];
        float logvar = mo[b * 2 * E * HW + (int64_t)E * HW + r];
        logvar = fminf(fmaxf(logvar, -30.f), 20.f);             // distributions.py:28
        const float std = expf(0.5f * logvar);
        const float eps = noise ? noise[i] : philox_normal_at(i, seed, 0);
        z[i] = mean + std * eps;                                // :36
    }
}
void gaussian_sample(const float* moments, const float* noise, uint64_t seed, int B, int E, int HW, float* z, hipStream_t s) {
    const int64_t total = (int64_t)B * E * HW;
    if (!total) return;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, moments,
                       noise, seed, B, E, HW, z);
    check_launch("gaussian_sample");
}

__global__ void fill_t_kernel(float* t, int B, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) t[i] = v;
}
void fill_t(float* t, int B, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_t_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, t, B, v);
    check_launch("fill_t");
}

}  // namespace dsd
