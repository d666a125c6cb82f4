// norm.hip — GroupNorm(32) statistics, affine+SiLU apply, LayerNorm.  HBM-bound kernels.
//
// Replaces GroupNorm32 (+ nn.SiLU) (ldm/modules/diffusionmodules/util.py:209-226; call sites
// openaimodel.py:205-209,229-236,451; UNet_DS_Diff/model.py:155-163,511-513) and nn.LayerNorm in
// ldm/modules/attention.py:316-318.
//
// Statistics are kept as PER-COLUMN partial sums  partial[n][chunk][c] = (sum, sumsq) in fp64 over a chunk of a sample's
// pixels.  Three producers write that layout: the epilogue of the convolution that produces the tensor (conv_split.hip:
// the values are in registers there, so the statistics cost no pass over HBM), the skip-average / concat kernel
// (avg_into_stats below) and, for tensors neither of them covers, the standalone gn_stats pass (NHWC: a thread owns fixed
// float4 columns and streams whole pixel rows with 16-byte loads, fp64 per-channel partials in registers).  A
// concatenated tensor simply has two sources (one per part), so group boundaries need not align with the parts.
// gn_finalize reduces columns x chunks of a group in a fixed order (deterministic) and folds mean/rstd/gamma/beta (and
// the FiLM scale/shift of use_scale_shift_norm ResBlocks) into one per-(sample,channel) scale/shift pair, so the apply
// pass is y = act(x*scale + shift): 1 read + 1 write of the tensor.
#include "kernels.h"

#include <cstdlib>

namespace dsd {

static constexpr int GN_GROUPS = 32;

// (A hardware exp2 / reciprocal SiLU was measured: 18.9 -> 18.4 ms per step for ~3e-7 of extra error; not taken.)
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }

struct GnGeom {
    int threads;  // block size
    int k;        // float4 columns per thread
    int rpi;      // pixel rows per iteration
    int ppc;      // pixels per chunk
    int nchunk;
};

static GnGeom gn_geom(int HW, int C) {
    GnGeom g{};
    const int cols = C / 4;
    if (cols <= 256) {
        g.k = 1;
        g.rpi = 256 / cols;
        g.threads = g.rpi * cols;
    } else {
        g.k = 2;
        while (cols % g.k != 0 || cols / g.k > 256) ++g.k;
        g.rpi = 1;
        g.threads = cols / g.k;
    }
    int ppc = cdiv(HW, 256);
    if (ppc < 32) ppc = 32;
    ppc = cdiv(ppc, g.rpi) * g.rpi;
    g.ppc = ppc;
    g.nchunk = cdiv(HW, ppc);
    return g;
}

int gn_nchunks(int HW, int C) { return gn_geom(HW, C).nchunk; }

template <int K>
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int HW, int C, int rpi, int ppc,
                                                       double* __restrict__ partial) {
    extern __shared__ double sm[];  // [rpi][C][2]
    const int cols = C >> 2;
    const int colsk = cols / K;
    const int tid = threadIdx.x;
    const int row = tid / colsk;
    const int col0 = tid - row * colsk;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * ppc;
    const int p1 = min(HW, p0 + ppc);
    double s[K][4], q[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[k][e] = q[k][e] = 0.0;
    const float* base = x + (int64_t)n * HW * C;
    auto accum = [&](int k, const float4 v) {
        const double a = v.x, b = v.y, c = v.z, d = v.w;
        s[k][0] += a; q[k][0] = fma(a, a, q[k][0]);
        s[k][1] += b; q[k][1] = fma(b, b, q[k][1]);
        s[k][2] += c; q[k][2] = fma(c, c, q[k][2]);
        s[k][3] += d; q[k][3] = fma(d, d, q[k][3]);
    };
    // 4 pixel rows per trip: 4*K independent 16-byte loads in flight per thread before the fp64 accumulation
    int p = p0 + row;
    for (; p + 3 * rpi < p1; p += 4 * rpi) {
        float4 v[4][K];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k)
                v[u][k] = reinterpret_cast<const float4*>(base + (int64_t)(p + u * rpi) * C)[col0 + k * colsk];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k) accum(k, v[u][k]);
    }
    for (; p < p1; p += rpi) {
        const float4* rp = reinterpret_cast<const float4*>(base + (int64_t)p * C);
#pragma unroll
        for (int k = 0; k < K; ++k) accum(k, rp[col0 + k * colsk]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (col0 + k * colsk) * 4 + e;
            sm[((int64_t)row * C + c) * 2 + 0] = s[k][e];
            sm[((int64_t)row * C + c) * 2 + 1] = q[k][e];
        }
    __syncthreads();
    // rows -> one (sum, sumsq) per column; consecutive threads take consecutive columns (16-byte LDS stride)
    for (int c = tid; c < C; c += blockDim.x) {
        double ss = 0.0, qq = 0.0;
        for (int r = 0; r < rpi; ++r) {
            ss += sm[((int64_t)r * C + c) * 2 + 0];
            qq += sm[((int64_t)r * C + c) * 2 + 1];
        }
        double* o = partial + (((int64_t)n * gridDim.x + chunk) * C + c) * 2;
        o[0] = ss;
        o[1] = qq;
    }
}

void gn_stats(const float* x, int N, int HW, int C, double* partial, int nchunk, hipStream_t s) {
    DSD_CHECK(C % GN_GROUPS == 0 && C % 4 == 0, "GroupNorm32: C=%d must be a multiple of 32", C);
    const GnGeom g = gn_geom(HW, C);
    DSD_CHECK(g.nchunk == nchunk, "gn_stats: chunk count mismatch");
    const size_t lds = (size_t)g.rpi * C * 2 * sizeof(double);
    DSD_CHECK(lds <= 64 * 1024, "gn_stats: C=%d too large", C);
    const dim3 grid(g.nchunk, N), block(g.threads);
    switch (g.k) {
        case 1: hipLaunchKernelGGL(gn_stats_kernel<1>, grid, block, lds, s, x, HW, C, g.rpi, g.ppc, partial); break;
        case 2: hipLaunchKernelGGL(gn_stats_kernel<2>, grid, block, lds, s, x, HW, C, g.rpi, g.ppc, partial); break;
        case 3: hipLaunchKernelGGL(gn_stats_kernel<3>, grid, block, lds, s, x, HW, C, g.rpi, g.ppc, partial); break;
        case 4: hipLaunchKernelGGL(gn_stats_kernel<4>, grid, block, lds, s, x, HW, C, g.rpi, g.ppc, partial); break;
        default: fail("gn_stats: C=%d needs %d columns per thread (unsupported)", C, g.k);
    }
    check_launch("gn_stats");
}

// One workgroup per (group, sample): sums the group's columns over all chunks of their source (fixed order: thread t takes
// items t, t+256, ...; then a fixed LDS tree), then writes scale/shift of the group's channels.
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnSrc s0, GnSrc s1, int HW, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          const float* __restrict__ film, int film_stride,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[256][2];
    __shared__ float mean_s, rstd_s;
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / GN_GROUPS;
    const int g0 = g * cpg, g1 = g0 + cpg;
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const GnSrc& sr = k == 0 ? s0 : s1;
        if (!sr.p) continue;
        const int lo = max(g0, sr.c0), hi = min(g1, sr.c0 + sr.c);   // this group's columns inside the source
        const int ncol = hi - lo;
        if (ncol <= 0) continue;
        const int items = sr.chunks * ncol;
        const double* base = sr.p + (int64_t)n * sr.chunks * sr.c * 2;
        for (int it = tid; it < items; it += 256) {
            const int ch = it / ncol, c = lo + (it - ch * ncol) - sr.c0;
            const double* o = base + ((int64_t)ch * sr.c + c) * 2;
            ss += o[0];
            qq += o[1];
        }
    }
    red[tid][0] = ss;
    red[tid][1] = qq;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            red[tid][0] += red[tid + st][0];
            red[tid][1] += red[tid + st][1];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)HW * cpg;
        const double mean = red[0][0] / cnt;
        double var = red[0][1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s = (float)mean;
        rstd_s = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = g0 + tid; c < g1; c += 256) {
        float sc = rstd_s * gamma[c];
        float sh = beta[c] - mean_s * sc;
        if (film) {  // (GN(x)) * (1 + fscale) + fshift, openaimodel.py:278-279
            const float f = 1.f + film[(int64_t)n * film_stride + c];
            sc *= f;
            sh = sh * f + film[(int64_t)n * film_stride + C + c];
        }
        scale[(int64_t)n * C + c] = sc;
        shift[(int64_t)n * C + c] = sh;
    }
}

void gn_finalize(const GnSrc& s0, const GnSrc& s1, int N, int HW, int C, const float* gamma, const float* beta,
                 float eps, const float* film, int film_stride, float* scale, float* shift, hipStream_t s) {
    DSD_CHECK(s0.p && s0.c0 == 0 && s0.c + (s1.p ? s1.c : 0) == C && (!s1.p || s1.c0 == s0.c),
              "gn_finalize: the statistic sources do not cover the %d channels", C);
    if (N == 0) return;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_GROUPS, N), dim3(256), 0, s, s0, s1, HW, C, gamma, beta, eps, film,
                       film_stride, scale, shift);
    check_launch("gn_finalize");
}

// Small maps (a group's HW x C/32 values fit a workgroup's reach): ONE launch per GroupNorm instead of statistics +
// finalize + apply — one workgroup per (group, sample) sums its values in fp64, then normalises them (second read from L2).
// An experiment for the batch-1 latency (the three-launch form of the 8x8 ... 32x32 layers looked like pure launch latency);
// measured no gain, see gn_small_ok.
// V = values per load (2 when the group width is even: 8-byte loads; else 1).  Loads are issued U at a time before they are
// consumed (a dependent chain of single loads made the first version latency-bound: 27 us per launch at batch 1).
template <int ACT, int V>
__global__ __launch_bounds__(256) void gn_small_kernel(const float* __restrict__ x, int HW, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, const float* __restrict__ film,
                                                       int film_stride, float* __restrict__ y) {
    constexpr int U = 8;
    __shared__ double red[4][2];
    __shared__ float mean_s, rstd_s;
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = C / GN_GROUPS, cv = cpg / V;
    const float* xb = x + (int64_t)n * HW * C + g * cpg;
    float* yb = y + (int64_t)n * HW * C + g * cpg;
    const int items = HW * cv;                      // V-wide items: (pixel, channel pair)
    auto off = [&](int i) {
        const int p = i / cv, c = i - p * cv;
        return (int64_t)p * C + c * V;
    };
    double s = 0.0, q = 0.0;
    for (int i0 = tid; i0 < items; i0 += 256 * U) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
#pragma unroll
            for (int k = 0; k < V; ++k) v[u][k] = 0.f;
            if (i < items) {
                if (V == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(xb + off(i));
                    v[u][0] = t.x;
                    v[u][V - 1] = t.y;
                } else {
                    v[u][0] = xb[off(i)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const double d = v[u][k];     // padding items add exact zeros
                s += d;
                q = fma(d, d, q);
            }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
    if (lane == 0) {
        red[wave][0] = s;
        red[wave][1] = q;
    }
    __syncthreads();
    if (tid == 0) {
        const double a = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        const double b = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        const double cnt = (double)HW * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s = (float)mean;
        rstd_s = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = mean_s, rstd = rstd_s;
    auto coef = [&](int ch, float& sc, float& sh) {   // the same scale / shift form as gn_finalize + affine_act
        sc = rstd * gamma[ch];
        sh = beta[ch] - mean * sc;
        if (film) {
            const float f = 1.f + film[(int64_t)n * film_stride + ch];
            sc *= f;
            sh = sh * f + film[(int64_t)n * film_stride + C + ch];
        }
    };
    for (int i0 = tid; i0 < items; i0 += 256 * U) {
        float v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            if (i < items) {
                if (V == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(xb + off(i));
                    v[u][0] = t.x;
                    v[u][V - 1] = t.y;
                } else {
                    v[u][0] = xb[off(i)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            if (i >= items) continue;
            const int p = i / cv, c = (i - p * cv) * V;
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float sc, sh;
                coef(g * cpg + c + k, sc, sh);
                o[k] = fmaf(v[u][k], sc, sh);
                if (ACT == ACT_SILU) o[k] = silu_f(o[k]);
            }
            if (V == 2)
                *reinterpret_cast<float2*>(yb + (int64_t)p * C + c) = make_float2(o[0], o[V - 1]);
            else
                yb[(int64_t)p * C + c] = o[0];
        }
    }
}

// Measured (bench.py kernels table, same kernels otherwise): batch 1 — 38.6 ms per step with it, 38.0 without; batch 16 — the
// small maps' GroupNorm work 3.0 ms with it against ~2.5 ms as statistics-from-epilogue + finalize + apply.  672 instead of
// 854 launches per step, but the step is not launch-bound, so it stays OFF: DSD_GN_SMALL=1 enables it.
bool gn_small_ok(int HW, int C) {
    static const bool on = getenv("DSD_GN_SMALL") != nullptr;
    return on && C % GN_GROUPS == 0 && (int64_t)HW * (C / GN_GROUPS) <= 32768;
}

void gn_small(const float* x, int N, int HW, int C, const float* gamma, const float* beta, float eps, const float* film,
              int film_stride, int act, float* y, hipStream_t s) {
    if ((int64_t)N * HW * C == 0) return;
    const bool v2 = (C / GN_GROUPS) % 2 == 0;   // group start and pixel stride are then multiples of 8 bytes
    const dim3 grid(GN_GROUPS, N), block(256);
#define DSD_GNS(A, VV) hipLaunchKernelGGL((gn_small_kernel<A, VV>), grid, block, 0, s, x, HW, C, gamma, beta, eps, film, film_stride, y)
    if (act == ACT_SILU) {
        if (v2) DSD_GNS(ACT_SILU, 2); else DSD_GNS(ACT_SILU, 1);
    } else {
        if (v2) DSD_GNS(ACT_NONE, 2); else DSD_GNS(ACT_NONE, 1);
    }
#undef DSD_GNS
    check_launch("gn_small");
}

// dst[:, coff:coff+C] = act((a [+b] [+c] [+d]) / div)  (the job of avg_into, misc.hip) AND the per-column statistics of what
// it wrote, in the geometry of the statistics pass: grid (chunk, sample), a thread owns K fixed float4 columns.
template <int K, int ACT>
__global__ __launch_bounds__(256) void avg_stats_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                        const float4* __restrict__ c, const float4* __restrict__ d,
                                                        float div, int HW, int C, int rpi, int ppc, float* __restrict__ dst,
                                                        int dstC, int coff, int bmask, double* __restrict__ partial) {
    extern __shared__ double sm[];  // [rpi][C][2]
    const int cols = C >> 2;
    const int colsk = cols / K;
    const int tid = threadIdx.x;
    const int row = tid / colsk;
    const int col0 = tid - row * colsk;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * ppc;
    const int p1 = min(HW, p0 + ppc);
    double s[K][4], q[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[k][e] = q[k][e] = 0.0;
    const int64_t sb = (int64_t)n * HW * cols;   // sample base (float4 units); broadcast sources have one sample only
    for (int p = p0 + row; p < p1; p += rpi) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c4 = col0 + k * colsk;
            const int64_t is = (int64_t)p * cols + c4, i = sb + is;
            float4 v = a[(bmask & 1) ? is : i];
            if (b) { const float4 t = b[(bmask & 2) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (c) { const float4 t = c[(bmask & 4) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (d) { const float4 t = d[(bmask & 8) ? is : i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (div != 1.f) { v.x /= div; v.y /= div; v.z /= div; v.w /= div; }
            if (ACT == ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            *reinterpret_cast<float4*>(dst + ((int64_t)n * HW + p) * dstC + coff + c4 * 4) = v;
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            s[k][0] += x0; q[k][0] = fma(x0, x0, q[k][0]);
            s[k][1] += x1; q[k][1] = fma(x1, x1, q[k][1]);
            s[k][2] += x2; q[k][2] = fma(x2, x2, q[k][2]);
            s[k][3] += x3; q[k][3] = fma(x3, x3, q[k][3]);
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = (col0 + k * colsk) * 4 + e;
            sm[((int64_t)row * C + cc) * 2 + 0] = s[k][e];
            sm[((int64_t)row * C + cc) * 2 + 1] = q[k][e];
        }
    __syncthreads();
    for (int cc = tid; cc < C; cc += blockDim.x) {
        double ss = 0.0, qq = 0.0;
        for (int r = 0; r < rpi; ++r) {
            ss += sm[((int64_t)r * C + cc) * 2 + 0];
            qq += sm[((int64_t)r * C + cc) * 2 + 1];
        }
        double* o = partial + (((int64_t)n * gridDim.x + chunk) * C + cc) * 2;
        o[0] = ss;
        o[1] = qq;
    }
}

void avg_into_stats(const float* a, const float* b, const float* c, const float* d, float div, int N, int HW, int C,
                    float* dst, int dstC, int coff, int act, int bmask, double* partial, int nchunk, hipStream_t s) {
    DSD_CHECK(C % 4 == 0 && dstC % 4 == 0 && coff % 4 == 0, "avg_into_stats: channel counts must be multiples of 4");
    if ((int64_t)N * HW * C == 0) return;
    const GnGeom g = gn_geom(HW, C);
    DSD_CHECK(g.nchunk == nchunk, "avg_into_stats: chunk count mismatch");
    const size_t lds = (size_t)g.rpi * C * 2 * sizeof(double);
    DSD_CHECK(lds <= 64 * 1024, "avg_into_stats: C=%d too large", C);
    const dim3 grid(g.nchunk, N), block(g.threads);
#define DSD_AVG(KK)                                                                                                          \
    if (act == ACT_SILU)                                                                                                     \
        hipLaunchKernelGGL((avg_stats_kernel<KK, ACT_SILU>), grid, block, lds, s, (const float4*)a, (const float4*)b,        \
                           (const float4*)c, (const float4*)d, div, HW, C, g.rpi, g.ppc, dst, dstC, coff, bmask, partial);  \
    else                                                                                                                     \
        hipLaunchKernelGGL((avg_stats_kernel<KK, ACT_NONE>), grid, block, lds, s, (const float4*)a, (const float4*)b,        \
                           (const float4*)c, (const float4*)d, div, HW, C, g.rpi, g.ppc, dst, dstC, coff, bmask, partial);
    switch (g.k) {
        case 1: DSD_AVG(1) break;
        case 2: DSD_AVG(2) break;
        case 3: DSD_AVG(3) break;
        case 4: DSD_AVG(4) break;
        default: fail("avg_into_stats: C=%d needs %d columns per thread (unsupported)", C, g.k);
    }
#undef DSD_AVG
    check_launch("avg_into_stats");
}


// y = act(x * scale[n][c] + shift[n][c]).  Like the statistics pass a thread owns fixed float4 columns, so its scale / shift
// live in registers and the loop body has no index arithmetic beyond one add (the first version recomputed (sample, column)
// from a flat 64-bit index with two integer divisions per 16 bytes and was VALU-bound at 4.9 TB/s).
template <int ACT>
__global__ __launch_bounds__(256) void affine_act_kernel(const float4* __restrict__ x, int HW, int cols, int W, int ppc,
                                                         const float4* __restrict__ scale, const float4* __restrict__ shift,
                                                         float4* __restrict__ y) {
    const int n = blockIdx.y;
    const int rpi = blockDim.x / W;
    const int row = threadIdx.x / W;
    const int cw = threadIdx.x - row * W;
    const int p0 = blockIdx.x * ppc;
    const int p1 = min(HW, p0 + ppc);
    const float4* xs = x + (int64_t)n * HW * cols;
    float4* ys = y + (int64_t)n * HW * cols;
    auto act = [](float4 v, const float4 sc, const float4 sh) {
        float4 o;
        o.x = fmaf(v.x, sc.x, sh.x);
        o.y = fmaf(v.y, sc.y, sh.y);
        o.z = fmaf(v.z, sc.z, sh.z);
        o.w = fmaf(v.w, sc.w, sh.w);
        if (ACT == ACT_SILU) {
            o.x = silu_f(o.x);
            o.y = silu_f(o.y);
            o.z = silu_f(o.z);
            o.w = silu_f(o.w);
        }
        return o;
    };
    for (int c4 = cw; c4 < cols; c4 += W) {   // one pass when cols <= 256
        const float4 sc = scale[(int64_t)n * cols + c4];
        const float4 sh = shift[(int64_t)n * cols + c4];
        int p = p0 + row;
        for (; p + 3 * rpi < p1; p += 4 * rpi) {   // 4 independent 16-byte loads in flight per thread
            const int64_t i0 = (int64_t)p * cols + c4, st = (int64_t)rpi * cols;
            const float4 v0 = xs[i0], v1 = xs[i0 + st], v2 = xs[i0 + 2 * st], v3 = xs[i0 + 3 * st];
            ys[i0] = act(v0, sc, sh);
            ys[i0 + st] = act(v1, sc, sh);
            ys[i0 + 2 * st] = act(v2, sc, sh);
            ys[i0 + 3 * st] = act(v3, sc, sh);
        }
        for (; p < p1; p += rpi) {
            const int64_t i0 = (int64_t)p * cols + c4;
            ys[i0] = act(xs[i0], sc, sh);
        }
    }
}

void affine_act(const float* x, int N, int HW, int C, const float* scale, const float* shift, int act, float* y,
                hipStream_t s) {
    const int cols = C / 4;
    if ((int64_t)N * HW * cols == 0) return;
    const int W = std::min(cols, 256);
    const int rpi = 256 / W, threads = rpi * W;
    // ~4096 workgroups in all, at least 4 row passes each
    int nchunk = std::max(1, std::min(4096 / std::max(N, 1), HW / (4 * rpi)));
    const int ppc = cdiv(cdiv(HW, nchunk), rpi) * rpi;
    nchunk = cdiv(HW, ppc);
    const dim3 grid((unsigned)nchunk, (unsigned)N);
    if (act == ACT_SILU)
        hipLaunchKernelGGL(affine_act_kernel<ACT_SILU>, grid, dim3(threads), 0, s, (const float4*)x, HW, cols, W, ppc,
                           (const float4*)scale, (const float4*)shift, (float4*)y);
    else
        hipLaunchKernelGGL(affine_act_kernel<ACT_NONE>, grid, dim3(threads), 0, s, (const float4*)x, HW, cols, W, ppc,
                           (const float4*)scale, (const float4*)shift, (float4*)y);
    check_launch("affine_act");
}

// LayerNorm over the last dim: one wave per row, two-pass (mean, then centred variance) in fp32.
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
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
    for (int c = lane; c < C; c += 64) y[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

void layer_norm(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps, float* y,
                hipStream_t s) {
    if (rows == 0) return;
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, C, gamma, beta, eps, y);
    check_launch("layer_norm");
}

}  // namespace dsd
