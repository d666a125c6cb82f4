// conv_tr16.hip — the dominant convolution kernel (256-row x 160-column tile, bf16x6, tap reuse through LDS, optional fused
// GroupNorm + SiLU of the input: conv_split_kernels.inc, TR / GN instantiation) with its matrix work issued as
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why: the kernel is power-bound, and bare MFMA loops on this chip hold a higher clock on the 16x16x32 shape
// at the same matrix-pipe cycles — csrc/peak.hip, same accumulator tile, random operands: 2144-2154 vs 1902-1912 TF/s with
// the operands in registers (1.127x), 1900-1906 vs 1789-1795 with the weight fragments re-read from LDS (1.062x)
// (profiles/r03_mfma_peak.txt; MI355X_MICROARCH.md "DVFS give-back" item 7).  VERDICT r2 item 5 asks for that shape in the
// TR+GN instantiation, measured.
//
// What changes against the 32x32x16 kernel (everything else — staging of the filter row and of the weight tile, the
// one-barrier software pipeline, k-tile order, epilogue fusions — is the same design):
//   * a wave still owns 64 pixels x 160 channels, now as 4 row blocks x 10 column tiles of 16 x 16 (the same 160 accumulator
//     registers); a k-tile (32 channels of one filter tap) is ONE k-step: 10 units of 4 x 6 MFMAs of 16 cycles = the 3840
//     matrix-pipe cycles of before;
//   * lane = (pixel l & 15 of a row block, k quarter l >> 4): its 8 consecutive k are two 16-byte pieces of the staged
//     filter row.  The row's 16-byte chunks are XOR-swizzled with (px >> 1 & 1) | (px >> 2 & 1) << 2 (found by exhaustive
//     search over linear swizzles: zero bank conflicts for this operand pattern under all three tap shifts; the 32x32
//     kernel's (px >> 1) & 7 is 2-way here);
//   * weight fragments: lane = (channel l & 15 of a column tile, k quarter): plane rows of 64 B, chunk c of row r stored at
//     c ^ ((r >> 1) & 3): conflict-free ds_read_b128 reads AND conflict-free ds_write_b128 staging (8 lanes = two whole
//     rows); no padding (30 KB per stage instead of 37.5 KB);
//   * accumulator layout: lane = channel, registers = 4 consecutive pixels -> the epilogue stores 64 contiguous bytes per
//     16 lanes and pixel; bias / embedding / residual / GroupNorm statistics are re-stated for that layout.
// Selected by DSD_CONV_MFMA16=1 (default: see conv_split.hip's planner note for the measured outcome).
#include "conv_split_kernels.inc"

namespace dsd {

typedef float f32x4a __attribute__((ext_vector_type(4)));

template <bool GN>
__global__ __launch_bounds__(256, 1) void conv_split_tr16_kernel(SplitP p) {
    constexpr int NT = 5, NP = 3, CT = 10;           // CT column tiles of 16
    constexpr int BROWS = NT * 32;
    constexpr int RS = 64;                            // LDS row stride of a weight plane (no padding: XOR swizzle)
    constexpr int B_PLANE = BROWS * RS;
    constexpr int NBL = (BROWS * 4 * NP + 255) / 256; // 16-byte weight loads per thread per tile (8, the last half empty)
    constexpr int STAGE = NP * B_PLANE;
    constexpr int A_STAGE = 257 * 128;                // one filter row of the tile: [pixel][32 channels fp32] + a row of zeros
    constexpr int A_OFF = 2 * STAGE + 256 * 16;
    static_assert(A_OFF + 2 * A_STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char Bs[A_OFF + 2 * A_STAGE];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, quart = lane >> 4;
    const int nwg = gridDim.x;
    int L = blockIdx.x;
    {
        const int cpx = nwg >> 3;
        if (L < (cpx << 3)) L = (L & 7) * cpx + (L >> 3);
    }
    const int tile_n = L % p.tiles_n, tile_m = L / p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * BROWS;

    // ---- weight staging: slot q = tid + 256 i -> (piece, row = rem >> 2, chunk = rem & 3); 8 consecutive lanes = two rows
    unsigned b_voff[NBL];
    int b_lds[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        const int q = tid + 256 * i;
        const int piece = q / (BROWS * 4);
        const int rem = q - piece * (BROWS * 4);
        const int row = rem >> 2, ch = rem & 3;
        const int n = n0 + row;
        const bool ok = piece < NP && n < p.Cout;
        b_voff[i] = ok ? (unsigned)piece * p.w_plane_bytes + ((unsigned)n * (unsigned)p.Ktot) * 2u + (unsigned)(ch * 16) : OOB;
        b_lds[i] = piece < NP ? piece * B_PLANE + row * RS + ((ch ^ ((row >> 1) & 3)) << 4) : -1;
    }
    const int b_dummy = 2 * STAGE + tid * 16;
    // ---- tap reuse: this lane's pixels inside the tile (4 row blocks of 16), and the staging role of this thread
    int a_lm[4], a_ow[4];
    unsigned g_off[8];
    int g_oh[8];
    const unsigned rowpitch = (unsigned)p.W * (unsigned)p.Cin * 4u;
    auto aswz = [](int px) { return ((px >> 1) & 1) | (((px >> 2) & 1) << 2); };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a_lm[r] = wave * 64 + r * 16 + l16;
        a_ow[r] = a_lm[r] % p.OW;
    }
    const int g_nb = m0 / p.ohw;
    {
        const int oh0 = (m0 - g_nb * p.ohw) / p.OW;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = i * 256 + tid;        // LDS slot: pixel q >> 3, physical chunk q & 7
            const int px = q >> 3;
            const int lc = (q & 7) ^ aswz(px);
            const int sg = px / p.OW, col = px - sg * p.OW;
            g_oh[i] = oh0 + sg - 1;
            g_off[i] = ((unsigned)g_nb * (unsigned)p.x_bs + (unsigned)(col >> p.ups) * (unsigned)p.Cin) * 4u + (unsigned)(lc * 16);
        }
    }
    u32x4 stg[8];
    f32x4 gsc = {0.f, 0.f, 0.f, 0.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    const int g_lc = (tid & 7) ^ aswz(tid >> 3);   // the logical chunk of every slot of this thread (32 i pixels leave bits 1, 2 alone)
    auto load_coef = [&](int c2) {
        if (!GN) return;
        const int c = min(c2, p.cchunks - 1) * SBK + g_lc * 4;
        gsc = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)g_nb * p.Cin + c);
        gsh = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)g_nb * p.Cin + c);
    };
    auto load_row = [&](int i, int c2, int h2) {
        const int ih = g_oh[i] + h2;
        const unsigned v = (unsigned)ih < (unsigned)p.IHg ? g_off[i] + (unsigned)(ih >> p.ups) * rowpitch : OOB;
        stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, v, __builtin_amdgcn_readfirstlane(min(c2, p.cchunks - 1) * (SBK * 4)), 0);
    };
    auto store_row = [&](int i, int c2, int h2) {
        u32x4 w = stg[i];
        if (GN) {
            const bool in = (unsigned)(g_oh[i] + h2) < (unsigned)p.IHg;
            const f32x4 v = __builtin_bit_cast(f32x4, w);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = fmaf(v[e], gsc[e], gsh[e]);
                o[e] = in ? t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * t)) : 0.f;
            }
            w = __builtin_bit_cast(u32x4, o);
        }
        *reinterpret_cast<u32x4*>(Bs + A_OFF + ((c2 + h2) & 1) * A_STAGE + (i * 256 + tid) * 16) = w;
    };
    u32x4 rb[NBL];
    f32x4a acc[4][CT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[r][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
    const int foff = l16 * RS + ((quart ^ ((l16 >> 1) & 3)) << 4);   // + piece * B_PLANE + j * 16 * RS
    auto mfma6 = [&](const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x4a& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    };

    const int KT = p.ks * p.ks * p.cchunks;
    int cc = 0, tap = 0, kh = 0, kw = 0;
    auto advance = [&]() {
        const bool wrapw = kw + 1 == p.ks;
        const bool wraph = wrapw && (kh + 1 == p.ks);
        kw = wrapw ? 0 : kw + 1;
        kh = wrapw ? (wraph ? 0 : kh + 1) : kh;
        cc += wraph ? 1 : 0;
        tap = kh * p.ks + kw;
    };
    auto load_b = [&]() {
        const int soff_b = __builtin_amdgcn_readfirstlane(min((tap * p.Cin + cc * SBK) * 2, (p.Ktot - SBK) * 2));
#pragma unroll
        for (int i = 0; i < NBL; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff_b, 0);
    };
    auto store_b = [&](int i, int so) {
        if (256 * (i + 1) <= BROWS * 4 * NP)
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
    // 16-byte piece i (0 / 1) of row block r under tap kw2 from the filter row with stage parity par
    auto lds_piece = [&](int r, int i, int par, int kw2) -> f32x4 {
        const bool in = (unsigned)(a_ow[r] + kw2 - 1) < (unsigned)p.OW;
        const int row = in ? a_lm[r] + kw2 - 1 : 256;   // 256 = the row of zeros
        const int ch = quart * 2 + i;
        return *reinterpret_cast<const f32x4*>(Bs + A_OFF + par * A_STAGE + row * 128 + ((ch ^ aswz(row)) << 4));
    };
    int kh_cur = 0, cc_cur = 0;
    f32x4 a_nx;
    bf16x8 af[4][NP], afn[4][NP];
    // prologue: filter row (0, 0) -> stage 0 + the zero rows of both stages; weight tile 0 -> stage 0, tile 1 in flight
    load_coef(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) load_row(i, 0, 0);
    load_b();
    if (tid < 16) *reinterpret_cast<u32x4*>(Bs + A_OFF + (tid >> 3) * A_STAGE + 256 * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) store_row(i, 0, 0);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) split8<NP, false>(lds_piece(r, 0, 0, 0), lds_piece(r, 1, 0, 0), af[r], nullptr);
#pragma unroll
    for (int i = 0; i < NBL; ++i) store_b(i, 0);
    advance();
    load_b();

    int soff_bs = 0;
    auto tile = [&](auto KWC, const int kt) __attribute__((always_inline)) {
        constexpr int kw_cur = decltype(KWC)::value;
        const int so = (kt & 1) * STAGE;
        __syncthreads();   // weight tile kt is visible; every wave is done with the other stage (tile kt-1)
        bf16x8 b_cur[NP], b_nxt[NP];
        const unsigned char* bf = Bs + so;
        u32x2 pl[NP], ph[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) b_nxt[q] = b_cur[q] = *reinterpret_cast<const bf16x8*>(bf + q * B_PLANE + foff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CT; ++u) {
            if (u + 1 < CT) {
#pragma unroll
                for (int q = 0; q < NP; ++q) b_nxt[q] = *reinterpret_cast<const bf16x8*>(bf + q * B_PLANE + (u + 1) * 16 * RS + foff);
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // the NEXT filter row: loaded during the first tile of this one, written to LDS during the second
                const bool wrap = kh_cur == 2;
                const int c2 = wrap ? cc_cur + 1 : cc_cur, h2 = wrap ? 0 : kh_cur + 1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (u >= NT && i * NT / 8 == u - NT && kw_cur == 0) {
                        if (i == 0) load_coef(c2);
                        load_row(i, c2, h2);
                    }
                    if ((GN ? u == i : (u < NT && i * NT / 8 == u)) && kw_cur == 1) store_row(i, c2, h2);
                }
                // activations of tile kt+1 -> pieces: one 16-byte piece per unit, fetched one unit ahead (piece t: row block t >> 1)
                if (u >= 1 && u <= 8) {
                    const int t = u - 1, r = t >> 1;
                    if ((t & 1) == 0) {
                        split4<NP, false>(a_nx, pl);
                    } else {
                        split4<NP, false>(a_nx, ph);
                        join(pl, ph, afn[r]);
                    }
                }
                if (u <= 7) {
                    const int kw_n = kw_cur == 2 ? 0 : kw_cur + 1;
                    const int par_n = kw_cur == 2 ? (c2 + h2) & 1 : (cc_cur + kh_cur) & 1;
                    a_nx = lds_piece(u >> 1, u & 1, par_n, kw_n);
                }
            }
            if (u < NT) {   // weight tile kt+1 -> the other stage, each staging register re-loaded (tile kt+2) right after its LDS write
                if (u == 0) {
                    advance();
                    soff_bs = __builtin_amdgcn_readfirstlane(min((tap * p.Cin + cc * SBK) * 2, (p.Ktot - SBK) * 2));
                }
#pragma unroll
                for (int i = 0; i < NBL; ++i)
                    if (i * NT / NBL == u) {
                        store_b(i, STAGE - so);
                        rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff_bs, 0);
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mfma6(af[r], b_cur, acc[r][u]);
#pragma unroll
            for (int q = 0; q < NP; ++q) b_cur[q] = b_nxt[q];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < NP; ++q) af[r][q] = afn[r][q];
        if (kw_cur == 2) {
            const bool wh = kh_cur == 2;
            kh_cur = wh ? 0 : kh_cur + 1;
            cc_cur += wh ? 1 : 0;
        }
    };
    for (int kt = 0; kt < KT; kt += 3) {
        tile(std::integral_constant<int, 0>{}, kt);
        tile(std::integral_constant<int, 1>{}, kt + 1);
        tile(std::integral_constant<int, 2>{}, kt + 2);
    }

    // ---- epilogue.  acc[r][j][e]: pixel m0 + wave*64 + 16 r + 4 quart + e, channel n0 + 16 j + l16; the tile lies inside ONE
    // sample (ohw % 256 == 0) and is full in M (M % 256 == 0): bias + embedding are per channel here
    float bj[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int n = min(n0 + j * 16 + l16, p.Cout - 1);
        bj[j] = (p.bias ? p.bias[n] : 0.f) + (p.emb ? p.emb[(int64_t)g_nb * p.emb_stride + n] : 0.f);
    }
    const bool stats = p.stats != nullptr;
    float ref[CT], cs[CT], cq[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) ref[j] = cs[j] = cq[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int mb = m0 + wave * 64 + r * 16 + quart * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* yp = p.y + (int64_t)(mb + e) * p.y_ld + n0 + l16;
            const float* rp = p.res ? p.res + (int64_t)(mb + e) * p.Cout + n0 + l16 : nullptr;
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (n0 + j * 16 + l16 >= p.Cout) continue;   // (lane-dependent only in the last, ragged column tile)
                float v = acc[r][j][e] + bj[j];
                if (rp) v += rp[j * 16];
                yp[j * 16] = v;
                if (r == 0 && e == 0) ref[j] = v;
                const float d = v - ref[j];
                cs[j] += d;
                cq[j] = fmaf(d, d, cq[j]);
            }
        }
    }
    if (!stats) return;
    // GroupNorm statistics of the output: shifted fp32 partials of this lane's 16 pixels -> plain fp64 (sum, sum of squares),
    // then lanes (the four k quarters hold different pixels of the same channel) and waves in a fixed order through LDS
    double* red = reinterpret_cast<double*>(Bs);   // [4 waves][BROWS][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const double rr = (double)ref[j], S = (double)cs[j];
        double s = S + 16.0 * rr, q = (double)cq[j] + 2.0 * rr * S + 16.0 * rr * rr;
        s += __shfl_xor(s, 16);
        q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (quart == 0) {
            red[((wave * BROWS) + j * 16 + l16) * 2 + 0] = s;
            red[((wave * BROWS) + j * 16 + l16) * 2 + 1] = q;
        }
    }
    __syncthreads();
    const int chunk = (m0 - g_nb * p.ohw) / 256;
    for (int c = tid; c < BROWS; c += 256) {
        const int n = n0 + c;
        if (n >= p.Cout) continue;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s += red[((w * BROWS) + c) * 2 + 0];
            q += red[((w * BROWS) + c) * 2 + 1];
        }
        double* o = p.stats + (((int64_t)g_nb * p.stats_chunks + chunk) * p.Cout + n) * 2;
        o[0] = s;
        o[1] = q;
    }
}

void launch_split_tr16(const SplitP& p, dim3 grid, hipStream_t s) {
    if (p.gn_scale)
        hipLaunchKernelGGL((conv_split_tr16_kernel<true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_split_tr16_kernel<false>), grid, dim3(256), 0, s, p);
}

}  // namespace dsd
