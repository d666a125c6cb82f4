// api.cpp — the extern "C" surface of libdsdiff.so (see include/dsdiff.h).
#include <cstring>
#include <vector>

#include "net.h"

using namespace dsd;

static thread_local std::string g_err;

#define DSD_TRY try {
#define DSD_CATCH                          \
    }                                      \
    catch (const std::exception& e) {      \
        g_err = e.what();                  \
        return -1;                         \
    }                                      \
    catch (...) {                          \
        g_err = "unknown error";           \
        return -1;                         \
    }                                      \
    return 0;

namespace {

void ensure_buf(float** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return;
    if (*p) {
        DSD_HIP(hipDeviceSynchronize());
        DSD_HIP(hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    DSD_HIP(hipMalloc((void**)p, bytes));
    *cap = bytes;
}

// temp device buffer for the dsd_op_* test entry points
struct Tmp {
    void* p = nullptr;
    explicit Tmp(size_t bytes) { DSD_HIP(hipMalloc(&p, bytes ? bytes : 256)); }
    ~Tmp() {
        if (p) {
            (void)hipDeviceSynchronize();
            (void)hipFree(p);
        }
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

void set_device(int device) {
    if (device >= 0) DSD_HIP(hipSetDevice(device));
}

void bind_planes(dsd_handle* h, const float* x, int C, int H, int W, hipStream_t s) {
    const int64_t hw = (int64_t)H * W;
    DSD_CHECK(C == 2 || C == 4, "x must have 2 or 4 channels (noise + 1 or 3 conditions), got %d", C);
    h->io.plane[0] = x;
    h->io.plane[1] = x + hw;
    h->io.plane_bs[0] = h->io.plane_bs[1] = (int64_t)C * hw;
    if (C == 2) {  // model.py:654-658: al = l = zeros_like(n)
        ensure_buf(&h->zplane, &h->zplane_cap, (size_t)hw * sizeof(float));
        DSD_HIP(hipMemsetAsync(h->zplane, 0, (size_t)hw * sizeof(float), s));
        h->io.plane[2] = h->io.plane[3] = h->zplane;
        h->io.plane_bs[2] = h->io.plane_bs[3] = 0;
    } else {       // model.py:660-663
        h->io.plane[2] = x + 2 * hw;
        h->io.plane[3] = x + 3 * hw;
        h->io.plane_bs[2] = h->io.plane_bs[3] = (int64_t)C * hw;
    }
}

}  // namespace

extern "C" {

const char* dsd_last_error(void) { return g_err.c_str(); }

int dsd_device_info(int device, char* name, int name_len, int* n_cu, int64_t* hbm_bytes) {
    DSD_TRY
    int cnt = 0;
    DSD_HIP(hipGetDeviceCount(&cnt));
    DSD_CHECK(device >= 0 && device < cnt, "device %d not present (%d visible)", device, cnt);
    hipDeviceProp_t prop;
    DSD_HIP(hipGetDeviceProperties(&prop, device));
    DSD_CHECK(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, "device %d is %s; libdsdiff is built for gfx950 only", device,
              prop.gcnArchName);
    if (name && name_len > 0) {
        std::strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    DSD_CATCH
}

int dsd_create(const dsd_config* cfg, int device, dsd_handle** out) {
    DSD_TRY
    DSD_CHECK(cfg && out, "null argument");
    set_device(device);
    auto* h = new dsd_handle();
    h->device = device;
    h->cfg = *cfg;
    try {
        net_declare_params(h);
    } catch (...) {
        net_free(h);
        delete h;
        throw;
    }
    *out = h;
    DSD_CATCH
}

int dsd_block_create(int kind, const int32_t* iargs, int n_iargs, int device, dsd_handle** out) {
    DSD_TRY
    DSD_CHECK(out && (iargs || n_iargs == 0), "null argument");
    set_device(device);
    auto* h = new dsd_handle();
    h->device = device;
    h->is_block = true;
    h->block_kind = kind;
    h->iargs.assign(iargs, iargs + n_iargs);
    h->cfg.use_new_attention_order = 1;
    try {
        net_declare_params(h);
    } catch (...) {
        net_free(h);
        delete h;
        throw;
    }
    *out = h;
    DSD_CATCH
}

void dsd_destroy(dsd_handle* h) {
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
    }
    net_free(h);
    delete h;
}

int dsd_param_count(dsd_handle* h) { return h ? (int)h->params.size() : -1; }

int dsd_param_info(dsd_handle* h, int idx, const char** name, int64_t shape[4], int* ndim) {
    DSD_TRY
    DSD_CHECK(h && idx >= 0 && idx < (int)h->params.size(), "parameter index out of range");
    const Param& p = h->params[idx];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int)p.shape.size();
    if (shape)
        for (size_t i = 0; i < 4; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
    DSD_CATCH
}

int dsd_set_param(dsd_handle* h, const char* name, const float* src, const int64_t* shape, int ndim, int src_is_device,
                  void* stream) {
    DSD_TRY
    DSD_CHECK(h && name && src && shape, "null argument");
    set_device(h->device);
    net_set_param(h, name, src, shape, ndim, src_is_device, (hipStream_t)stream);
    DSD_CATCH
}

int dsd_set_timestep_freqs(dsd_handle* h, const float* freqs_host, int n) {
    DSD_TRY
    DSD_CHECK(h && freqs_host && (!h->is_block || h->block_kind == DSD_BLOCK_DIT || h->block_kind == DSD_BLOCK_UNET),
              "null argument / handle without a timestep embedding");
    const int want = h->block_kind == DSD_BLOCK_DIT ? 128 : h->cfg.model_channels / 2;   // DiT: frequency_embedding_size 256 (DiT_models.py:31)
    DSD_CHECK(n == want, "expected %d frequencies, got %d", want, n);
    set_device(h->device);
    if (!h->freqs) DSD_HIP(hipMalloc((void**)&h->freqs, (size_t)n * sizeof(float)));
    DSD_HIP(hipMemcpy(h->freqs, freqs_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    DSD_CATCH
}

int dsd_set_precision(dsd_handle* h, int precision) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    DSD_CHECK(precision >= PREC_F32 && precision <= PREC_BF16, "unknown precision %d", precision);
    DSD_CHECK(precision <= PREC_F16X3 || (h->is_block && h->block_kind == DSD_BLOCK_DIT),
              "DSD_PREC_F16 / DSD_PREC_BF16 (single-product autocast arithmetic) exist for DSD_BLOCK_DIT handles only: the other "
              "networks are checked against the fp32 CPU path at 1e-4 and keep fp32-grade products");
    if (h->precision != precision) {
        set_device(h->device);
        h->precision = precision;
        h->plan.valid = false;   // the plan bakes the kernel choice in
        net_drop_graph(h);
        net_drop_other_pieces(h, precision);   // bf16 and fp16 pieces (6 B per weight each) are never both resident
    }
    DSD_CATCH
}

int dsd_get_precision(dsd_handle* h) { return h ? h->precision : -1; }

int dsd_set_share_zero_streams(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block, "needs a model handle");
    h->share_zero_streams = on != 0;
    DSD_CATCH
}

int dsd_params_ready(dsd_handle* h) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    for (const auto& p : h->params) DSD_CHECK(p.set, "parameter '%s' has not been set", p.name.c_str());
    DSD_CATCH
}

int dsd_plan(dsd_handle* h, int B, int C, int H, int W) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block, "dsd_plan needs a model handle");
    set_device(h->device);
    DSD_CHECK(C == 2 || C == 4, "x must have 2 or 4 channels, got %d", C);
    net_plan(h, B, C, H, W, C == 2, 0, 0, 0);
    DSD_CATCH
}

int64_t dsd_workspace_bytes(dsd_handle* h) { return h && h->plan.valid ? (int64_t)h->plan.arena_bytes : -1; }

int64_t dsd_device_bytes(dsd_handle* h) {
    if (!h) return -1;
    return (int64_t)(h->slab_bytes + h->staging_bytes + h->arena_cap + net_piece_bytes(h) + h->tbuf_cap + h->mout_cap +
                     h->zplane_cap + h->dpm_m_cap);
}

int dsd_set_graph(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    h->use_graph = on != 0;
    if (!on) {
        set_device(h->device);
        net_drop_graph(h);
    }
    DSD_CATCH
}

int dsd_set_fuse_gn_stats(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    if (h->fuse_gn_stats != (on != 0)) {
        h->fuse_gn_stats = on != 0;
        h->plan.valid = false;
    }
    DSD_CATCH
}

int dsd_set_fuse_gn_apply(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    if (h->fuse_gn_apply != (on != 0)) {
        h->fuse_gn_apply = on != 0;
        h->plan.valid = false;
    }
    DSD_CATCH
}

int dsd_set_stream_lanes(dsd_handle* h, int on, int max_pixels) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    const int px = max_pixels > 0 ? max_pixels : h->lane_pixels;
    const int mode = on == 2 ? 2 : (on != 0);   // 2: the plan of the lanes (tile choices included), launched on ONE stream (tests)
    if (h->use_lanes != mode || h->lane_pixels != px) {
        h->use_lanes = mode;
        h->lane_pixels = px;
        h->plan.valid = false;   // the emission order of the plan (and what the arena may recycle) depends on it
        net_drop_graph(h);
    }
    DSD_CATCH
}

int dsd_set_winograd(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    if (h->use_winograd != (on != 0)) {
        h->use_winograd = on != 0;
        h->plan.valid = false;
    }
    DSD_CATCH
}

int dsd_graph_stats(dsd_handle* h, int* captures, int* launches) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    if (captures) *captures = h->graph_captures;
    if (launches) *launches = h->graph_launches;
    DSD_CATCH
}

int dsd_set_slice_ids(dsd_handle* h, const int64_t* ids_host, int n) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block && n >= 0 && (ids_host || n == 0), "bad argument");
    set_device(h->device);
    if (h->slice_ids) {
        DSD_HIP(hipDeviceSynchronize());
        DSD_HIP(hipFree(h->slice_ids));
        h->slice_ids = nullptr;
    }
    h->n_slice_ids = 0;
    if (n > 0) {
        for (int i = 0; i < n; ++i) DSD_CHECK(ids_host[i] >= 0, "slice id %d is negative", i);
        DSD_HIP(hipMalloc((void**)&h->slice_ids, (size_t)n * sizeof(int64_t)));
        DSD_HIP(hipMemcpy(h->slice_ids, ids_host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
        h->n_slice_ids = n;
    }
    DSD_CATCH
}
int dsd_plan_launches(dsd_handle* h) { return h && h->plan.valid ? h->plan.launches : -1; }
double dsd_plan_flops(dsd_handle* h) { return h && h->plan.valid ? h->plan.flops : -1.0; }

int dsd_forward(dsd_handle* h, const float* x, const void* t, int t_is_float, int B, int C, int H, int W, float* out,
                float* const* feats, void* stream) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block && x && t && out, "null argument");
    set_device(h->device);
    hipStream_t s = (hipStream_t)stream;
    net_plan(h, B, C, H, W, C == 2, feats != nullptr, 0, 0, 0, s);
    bind_planes(h, x, C, H, W, s);
    h->io.t = t;
    h->io.t_is_float = t_is_float;
    h->io.out = out;
    h->io.feats = feats;
    net_run(h, s);
    net_check_overflow(h, s);
    DSD_CATCH
}

int dsd_profile_enable(dsd_handle* h, int on) {
    DSD_TRY
    DSD_CHECK(h, "null handle");
    h->profiling = on != 0;
    if (on) {
        h->prof_names.clear();
        h->prof_runs = 0;
    }
    DSD_CATCH
}

int dsd_profile_count(dsd_handle* h) { return h ? (int)h->prof_names.size() : -1; }

int dsd_profile_get(dsd_handle* h, int idx, const char** kind, double* total_ms, double* flops, double* bytes,
                    int64_t* calls, int* runs) {
    DSD_TRY
    DSD_CHECK(h && idx >= 0 && idx < (int)h->prof_names.size(), "profile index out of range");
    if (kind) *kind = h->prof_names[idx].c_str();
    if (total_ms) *total_ms = h->prof_ms[idx];
    if (flops) *flops = h->prof_flops[idx];
    if (bytes) *bytes = h->prof_bytes[idx];
    if (calls) *calls = h->prof_calls[idx];
    if (runs) *runs = h->prof_runs;
    DSD_CATCH
}

int dsd_profile_op_count(dsd_handle* h) { return h ? (int)h->prof_op_ms.size() : -1; }

int dsd_profile_op_get(dsd_handle* h, int idx, const char** kind, double* ms, double* flops, double* bytes) {
    DSD_TRY
    DSD_CHECK(h && idx >= 0 && idx < (int)h->prof_op_ms.size() && h->prof_op_ms.size() == h->plan.ops.size(),
              "op index out of range (or the plan changed since the profiled forward)");
    if (kind) *kind = h->plan.kind_names[h->plan.op_kind[idx]].c_str();
    if (ms) *ms = h->prof_op_ms[idx];
    if (flops) *flops = h->plan.op_flops[idx];
    if (bytes) *bytes = h->plan.op_bytes[idx];
    DSD_CATCH
}

int dsd_profile_op_name(dsd_handle* h, int idx, const char** name) {
    DSD_TRY
    DSD_CHECK(h && name && idx >= 0 && idx < (int)h->plan.op_name.size(), "op index out of range");
    *name = h->plan.op_name[idx].c_str();
    DSD_CATCH
}

int dsd_block_forward(dsd_handle* h, const float* x, int B, int C, int H, int W, const float* aux, int aux_len,
                      const float* aux2, int aux_len2, float* out, void* stream) {
    DSD_TRY
    DSD_CHECK(h && h->is_block && x && out, "null argument / not a block handle");
    set_device(h->device);
    net_plan(h, B, C, H, W, 0, 0, aux ? aux_len : 0, aux2 ? aux_len2 : 0, 0, (hipStream_t)stream);
    h->io.x_nchw = x;
    h->io.aux = aux;
    h->io.aux2 = aux2;
    h->io.out = out;
    net_run(h, (hipStream_t)stream);
    net_check_overflow(h, (hipStream_t)stream);
    DSD_CATCH
}

static StepCoef step_coef(const dsd_schedule* sc, int k) {
    StepCoef c{};
    for (int j = 0; j < DSD_NCOEF; ++j) c.c[j] = sc->coef[(size_t)k * DSD_NCOEF + j];
    c.mode = sc->mode; c.pred = sc->pred; c.learned_range = sc->learned_range; c.clip = sc->clip_denoised;
    c.nonzero = sc->nonzero ? sc->nonzero[k] : 1;
    c.eta = sc->eta;
    return c;
}

static void check_schedule(const dsd_schedule* sc) {
    DSD_CHECK(sc && sc->coef && sc->t_model && sc->steps >= 1, "bad schedule");
    DSD_CHECK(sc->mode >= DSD_MODE_A_DDPM && sc->mode <= DSD_MODE_B_DDIM, "unknown sampler mode %d", sc->mode);
    DSD_CHECK(sc->pred >= DSD_PRED_EPS && sc->pred <= DSD_PRED_V, "unknown prediction type %d", sc->pred);
    DSD_CHECK(!(sc->learned_range && sc->mode >= DSD_MODE_B_DDPM), "learned-range variance exists only in the guided-diffusion family");
}

// DiffusionWrapper 'concat' (ddpm.py:1331-1333) without materialising the cat: streams read planes in place
static void bind_sampling_io(dsd_handle* h, float* x, const float* cond, int Cc, int64_t hw, hipStream_t s) {
    h->io.plane[0] = x;
    h->io.plane_bs[0] = hw;
    h->io.plane[1] = cond;
    h->io.plane_bs[1] = (int64_t)Cc * hw;
    if (Cc == 1) {
        ensure_buf(&h->zplane, &h->zplane_cap, (size_t)hw * sizeof(float));
        DSD_HIP(hipMemsetAsync(h->zplane, 0, (size_t)hw * sizeof(float), s));
        h->io.plane[2] = h->io.plane[3] = h->zplane;
        h->io.plane_bs[2] = h->io.plane_bs[3] = 0;
    } else {
        h->io.plane[2] = cond + hw;
        h->io.plane[3] = cond + 2 * hw;
        h->io.plane_bs[2] = h->io.plane_bs[3] = (int64_t)Cc * hw;
    }
    h->io.t = h->tbuf;
    h->io.t_is_float = 1;
    h->io.out = h->mout;
    h->io.feats = nullptr;
}

int dsd_sample(dsd_handle* h, const dsd_schedule* sc, const float* cond, int Cc, float* x, const float* noise,
               uint64_t philox_seed, int B, int H, int W, int first_step, int n_steps, void* stream) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block && cond && x, "null argument");
    check_schedule(sc);
    DSD_CHECK(Cc == 1 || Cc == 3, "cond must have 1 or 3 channels, got %d", Cc);
    const int out_ch = h->cfg.out_channels;
    DSD_CHECK(out_ch == (sc->learned_range ? 2 : 1), "model has %d output channels but the schedule expects %d", out_ch,
              sc->learned_range ? 2 : 1);
    set_device(h->device);
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)H * W;
    net_plan(h, B, Cc + 1, H, W, Cc == 1, 0, 0, 0, (Cc == 1 && B > 1) ? h->share_zero_streams : 0, s);
    ensure_buf(&h->tbuf, &h->tbuf_cap, (size_t)B * sizeof(float));
    ensure_buf(&h->mout, &h->mout_cap, (size_t)B * out_ch * hw * sizeof(float));
    bind_sampling_io(h, x, cond, Cc, hw, s);
    DSD_CHECK(h->n_slice_ids == 0 || h->n_slice_ids == B, "dsd_set_slice_ids gave %d ids but the batch has %d slices", h->n_slice_ids, B);
    const int64_t* ids = h->n_slice_ids == B ? h->slice_ids : nullptr;
    const int k0 = first_step < 0 ? 0 : first_step;
    const int k1 = n_steps <= 0 ? sc->steps : std::min(sc->steps, k0 + n_steps);
    for (int k = k0; k < k1; ++k) {
        fill_t(h->tbuf, B, sc->t_model[k], s);
        net_run_cached(h, s);
        const StepCoef c = step_coef(sc, k);
        sampler_update(c, h->mout, x, noise ? noise + (size_t)k * B * hw : nullptr, philox_seed, (uint64_t)k, B, (int)hw, s,
                       nullptr, ids);
    }
    net_check_overflow(h, s);
    DSD_CATCH
}

int dsd_op_sampler_update(const dsd_schedule* sc, int k, const float* model_out, float* x, const float* noise,
                          uint64_t philox_seed, int B, int H, int W, float* pred_xstart, void* stream) {
    DSD_TRY
    check_schedule(sc);
    DSD_CHECK(k >= 0 && k < sc->steps && model_out && x, "bad argument");
    sampler_update(step_coef(sc, k), model_out, x, noise, philox_seed, (uint64_t)k, B, H * W, (hipStream_t)stream, pred_xstart);
    DSD_CATCH
}

// ------------------------------------------------------------------------------------------- DPM-Solver(++)
static void check_dpm_schedule(const dsd_dpm_schedule* sc) {
    DSD_CHECK(sc && sc->steps > 0 && sc->coef && sc->t_input && sc->order, "bad DPM schedule");
    DSD_CHECK(sc->pred >= DSD_PRED_EPS && sc->pred <= DSD_PRED_V, "bad pred %d", sc->pred);
    for (int k = 0; k < sc->steps; ++k) {
        DSD_CHECK(sc->order[k] >= 0 && sc->order[k] <= 2, "order[%d] = %d: the multistep solver is built for order <= 2", k,
                  sc->order[k]);
        DSD_CHECK(!(k == 0 && sc->order[k] == 2), "the first update cannot be second order");
    }
    DSD_CHECK(!sc->thresholding || (sc->threshold_ratio >= 0.f && sc->threshold_ratio <= 1.f), "threshold_ratio outside [0,1]");
}

static DpmCoef dpm_coef(const dsd_dpm_schedule* sc, int k) {
    const float* c = sc->coef + (size_t)k * DSD_NCOEF;
    DpmCoef d;
    d.alpha = c[0]; d.sigma = c[1]; d.cx = c[2]; d.cm = c[3]; d.cd = c[4]; d.ir0 = c[5];
    d.order = sc->order[k];
    d.pred = sc->pred;
    d.data_pred = sc->data_pred || d.order == 0;
    d.thresh = sc->thresholding;
    return d;
}

int dsd_sample_dpm(dsd_handle* h, const dsd_dpm_schedule* sc, const float* cond, int Cc, float* x, int B, int H, int W,
                   void* stream) {
    DSD_TRY
    DSD_CHECK(h && !h->is_block && cond && x, "null argument");
    check_dpm_schedule(sc);
    DSD_CHECK(Cc == 1 || Cc == 3, "cond must have 1 or 3 channels, got %d", Cc);
    const int out_ch = h->cfg.out_channels;
    DSD_CHECK(out_ch == 1 || out_ch == 2, "model has %d output channels; the solver takes 1 (or 2 with a learned sigma)", out_ch);
    set_device(h->device);
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)H * W;
    net_plan(h, B, Cc + 1, H, W, Cc == 1, 0, 0, 0, (Cc == 1 && B > 1) ? h->share_zero_streams : 0, s);
    ensure_buf(&h->tbuf, &h->tbuf_cap, (size_t)B * sizeof(float));
    ensure_buf(&h->mout, &h->mout_cap, (size_t)B * out_ch * hw * sizeof(float));
    ensure_buf(&h->dpm_m, &h->dpm_m_cap, ((size_t)2 * B * hw + B) * sizeof(float));
    bind_sampling_io(h, x, cond, Cc, hw, s);
    float* m_cur = h->dpm_m;
    float* m_prev = h->dpm_m + (size_t)B * hw;
    float* s_buf = h->dpm_m + (size_t)2 * B * hw;
    for (int k = 0; k < sc->steps; ++k) {
        fill_t(h->tbuf, B, sc->t_input[k], s);
        net_run_cached(h, s);
        dpm_step(dpm_coef(sc, k), h->mout, out_ch, x, m_cur, m_prev, s_buf, sc->threshold_ratio, sc->threshold_max, B, (int)hw, s);
        std::swap(m_cur, m_prev);
    }
    net_check_overflow(h, s);
    DSD_CATCH
}

int dsd_op_dpm_step(const dsd_dpm_schedule* sc, int k, const float* model_out, int Cm, float* x, float* m_cur,
                    const float* m_prev, int B, int H, int W, void* stream) {
    DSD_TRY
    check_dpm_schedule(sc);
    DSD_CHECK(k >= 0 && k < sc->steps && model_out && x && m_cur && (Cm == 1 || Cm == 2), "bad argument");
    DSD_CHECK(sc->order[k] < 2 || m_prev, "a second-order update needs m_prev");
    Tmp sb((size_t)B * sizeof(float));
    dpm_step(dpm_coef(sc, k), model_out, Cm, x, m_cur, m_prev, sb.as<float>(), sc->threshold_ratio, sc->threshold_max, B, H * W,
             (hipStream_t)stream);
    DSD_HIP(hipStreamSynchronize((hipStream_t)stream));
    DSD_CATCH
}

int dsd_op_dpm_threshold(const float* x0, int B, int n, float ratio, float max_val, float* y, float* s_out, void* stream) {
    DSD_TRY
    DSD_CHECK(x0 && y && s_out && B >= 0 && n >= 1 && ratio >= 0.f && ratio <= 1.f, "bad argument");
    dpm_threshold(x0, y, s_out, ratio, max_val, B, n, (hipStream_t)stream);
    DSD_CATCH
}

// ------------------------------------------------------------------------------------------- kernel-level ops
int dsd_op_conv2d(const float* x, int N, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout, int ks,
                  int stride, int upsample, const float* emb, const float* res, float* y, void* stream) {
    DSD_TRY
    hipStream_t s = (hipStream_t)stream;
    Tmp wp((size_t)Cout * Cin * ks * ks * sizeof(float));
    pack_ohwi(w_oihw, wp.as<float>(), Cout, Cin, ks, s);
    ConvArgs a;
    a.x = x; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.w = wp.as<float>(); a.bias = bias; a.Cout = Cout; a.ks = ks;
    a.stride = stride; a.ups = upsample; a.emb = emb; a.emb_stride = Cout; a.res = res; a.y = y;
    conv2d(a, s);
    DSD_HIP(hipStreamSynchronize(s));
    DSD_CATCH
}

int dsd_set_conv_mfma16(int on) {
    const int prev = conv2d_get_mfma16();
    conv2d_set_mfma16(on);
    return prev;
}

int dsd_conv_plan(int N, int H, int W, int Cin, int Cout, int ks, int stride, int precision, int* structure, int* nt,
                  int* ksplit, uint64_t* scratch_bytes) {
    DSD_TRY
    DSD_CHECK(structure && nt && ksplit && scratch_bytes, "null argument");
    ConvArgs a;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ks = ks; a.stride = stride;
    a.precision = precision & 3;
    if (precision & 16) a.variant = 30;
    if (precision & 32) a.variant = 31;
    if (precision & 64) a.variant = 32;
    if (precision & 256) a.lanes = 4;   // as inside a stream-lane region: three launches of the same shape run beside this one
    static const int dummy = 0;
    a.w_split = a.precision != PREC_F32 ? &dummy : nullptr;   // only its presence matters to the eligibility test
    conv2d_plan_query(a, structure, nt, ksplit);
    *scratch_bytes = conv2d_scratch_bytes(a);
    DSD_CATCH
}

int dsd_bench_conv2d(int N, int H, int W, int Cin, int Cout, int ks, int stride, int variant, int iters, float* avg_ms,
                     double* flops) {
    DSD_TRY
    DSD_CHECK(iters >= 1 && avg_ms, "bad argument");
    hipStream_t s = nullptr;
    const int pad = ks / 2;
    const int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1;
    const size_t nx = (size_t)N * H * W * Cin, nw = (size_t)Cout * Cin * ks * ks, ny = (size_t)N * OH * OW * Cout;
    Tmp x(nx * 4), w(nw * 4), b((size_t)Cout * 4), y(ny * 4);
    philox_normal(x.as<float>(), (int64_t)nx, 1, 0, s);
    philox_normal(w.as<float>(), (int64_t)nw, 2, 0, s);
    philox_normal(b.as<float>(), Cout, 3, 0, s);
    ConvArgs a;
    a.x = x.as<float>(); a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.w = w.as<float>(); a.bias = b.as<float>();
    a.Cout = Cout; a.ks = ks; a.stride = stride; a.y = y.as<float>(); a.variant = variant >= 10 ? -1 : variant;
    Tmp planes(nw * 2 * 3);
    const bool wino = variant == 50;   // 50 = bf16x6, F(2,3) along the width (conv_wino.hip)
    if (wino) variant = 11;
    if (variant >= 40) {  // 40 = bf16x3, 41 = bf16x6, 42 = f16x3 on the 256-row A-direct tile (forced)
        a.variant = 32;
        variant -= 30;
    } else if (variant >= 30) {  // 30 = bf16x3 A-direct, 31 = bf16x6 A-direct (forced)
        a.variant = 30;
        variant -= 20;
    } else if (variant >= 20) {  // 20 = bf16x3 staged, 21 = bf16x6 staged (forced)
        a.variant = 31;
        variant -= 10;
    }
    if (variant >= 10) {  // 10 = bf16x3, 11 = bf16x6, 12 = f16x3
        a.precision = variant == 10 ? PREC_BF16X3 : (variant == 11 ? PREC_BF16X6 : PREC_F16X3);
        split_weights(w.as<float>(), (int64_t)nw, 3, planes.p, s, a.precision == PREC_F16X3);
        a.w_split = planes.p;
    }
    Tmp wpk(wino ? wino_packed_bytes(Cout, Cin) : 256);
    if (wino) {
        DSD_CHECK(conv2d_wino_shape_ok(a), "this shape cannot run on the F(2,3) kernel");
        // the library keeps weights OHWI: random data is layout-agnostic here (timing only)
        wino_pack_weights(w.as<float>(), Cout, Cin, wpk.p, s);
        a.w_wino = wpk.p;
    }
    Tmp scratch(conv2d_scratch_bytes(a));
    a.scratch = scratch.as<float>();
    a.scratch_bytes = conv2d_scratch_bytes(a);
    conv2d(a, s);  // warm-up
    hipEvent_t e0, e1;
    DSD_HIP(hipEventCreate(&e0));
    DSD_HIP(hipEventCreate(&e1));
    DSD_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) conv2d(a, s);
    DSD_HIP(hipEventRecord(e1, s));
    DSD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    if (flops) *flops = conv2d_flops(a);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DSD_CATCH
}

int dsd_bench_conv2d_stamps(int N, int H, int W, int Cin, int Cout, int warm, int whatif, long long* out, int max_wgs,
                            int* n_wgs) {
    DSD_TRY
    DSD_CHECK(out && n_wgs && max_wgs > 0, "bad argument");
    hipStream_t s = nullptr;
    const size_t nx = (size_t)N * H * W * Cin, nw = (size_t)Cout * Cin * 9, ny = (size_t)N * H * W * Cout;
    Tmp x(nx * 4), w(nw * 4), b((size_t)Cout * 4), y(ny * 4), planes(nw * 2 * 3);
    philox_normal(x.as<float>(), (int64_t)nx, 1, 0, s);
    philox_normal(w.as<float>(), (int64_t)nw, 2, 0, s);
    philox_normal(b.as<float>(), Cout, 3, 0, s);
    ConvArgs a;
    a.x = x.as<float>(); a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.w = w.as<float>(); a.bias = b.as<float>();
    a.Cout = Cout; a.ks = 3; a.stride = 1; a.y = y.as<float>(); a.variant = 32; a.precision = PREC_BF16X6;
    split_weights(w.as<float>(), (int64_t)nw, 3, planes.p, s, false);
    a.w_split = planes.p;
    const int wgs = (int)(((int64_t)N * H * W + 255) / 256) * ((Cout + 159) / 160);
    DSD_CHECK(wgs <= max_wgs, "room for %d workgroups needed", wgs);
    for (int i = 0; i < warm; ++i) conv2d(a, s);
    Tmp st((size_t)wgs * 8 * sizeof(long long));
    DSD_HIP(hipMemsetAsync(st.p, 0, (size_t)wgs * 8 * sizeof(long long), s));
    a.stamps = st.as<long long>();
    a.diag = whatif;
    for (int i = 0; i < (whatif ? 20 : 0); ++i) conv2d(a, s);   // the what-if build's own steady state
    conv2d(a, s);
    DSD_HIP(hipMemcpyAsync(out, st.p, (size_t)wgs * 8 * sizeof(long long), hipMemcpyDeviceToHost, s));
    DSD_HIP(hipStreamSynchronize(s));
    *n_wgs = wgs;
    DSD_CATCH
}

int dsd_bench_mfma_peak(int variant, int workgroups_per_cu, float ms_target, int iters, float* avg_ms, double* tflops) {
    DSD_TRY
    DSD_CHECK(variant >= 0 && variant <= 7 && iters >= 1 && avg_ms && tflops, "bad argument");
    hipStream_t s = nullptr;
    const int wgs = 256 * (workgroups_per_cu > 0 ? std::min(workgroups_per_cu, 64) : 8);
    Tmp src((size_t)mfma_peak_src_bytes()), sink((size_t)wgs * 256 * sizeof(float));
    mfma_peak_fill(src.p, (variant & 2) != 0, s);
    hipEvent_t e0, e1;
    DSD_HIP(hipEventCreate(&e0));
    DSD_HIP(hipEventCreate(&e1));
    auto timed = [&](int loops, int n, double* fl) {
        DSD_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) *fl = mfma_peak_launch(variant, src.p, sink.as<float>(), wgs, loops, s);
        DSD_HIP(hipEventRecord(e1, s));
        DSD_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        DSD_HIP(hipEventElapsedTime(&ms, e0, e1));
        return ms / n;
    };
    double fl = 0.0;
    int loops = 64;
    const float probe = timed(loops, 1, &fl);   // also the warm-up; then size the loop for ms_target per launch
    const float want = ms_target > 0.f ? std::min(ms_target, 200.f) : 5.f;
    loops = (int)std::max(16.0, std::min(1.0e6, loops * (double)want / std::max(probe, 1e-3f)));
    (void)timed(loops, 1, &fl);
    const float ms = timed(loops, iters, &fl);
    *avg_ms = ms;
    *tflops = fl / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DSD_CATCH
}

int dsd_op_conv2d_prec(const float* x, int N, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout, int ks,
                       int stride, int upsample, const float* emb, const float* res, int precision, float* y, void* stream) {
    DSD_TRY
    hipStream_t s = (hipStream_t)stream;
    const size_t nw = (size_t)Cout * Cin * ks * ks;
    Tmp wp(nw * sizeof(float)), planes(nw * 2 * 3);
    pack_ohwi(w_oihw, wp.as<float>(), Cout, Cin, ks, s);
    ConvArgs a;
    a.x = x; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.w = wp.as<float>(); a.bias = bias; a.Cout = Cout; a.ks = ks;
    a.stride = stride; a.ups = upsample; a.emb = emb; a.emb_stride = Cout; a.res = res; a.y = y;
    Tmp ovf(sizeof(int));
    DSD_HIP(hipMemsetAsync(ovf.p, 0, sizeof(int), s));
    if ((precision & 3) != PREC_F32) {
        const bool f16 = (precision & 3) == PREC_F16X3;
        split_weights(wp.as<float>(), (int64_t)nw, 3, planes.p, s, f16, f16 ? ovf.as<int>() : nullptr);
        a.w_split = planes.p;
        a.precision = precision & 3;
        a.ovf = f16 ? ovf.as<int>() : nullptr;
        if (precision & 16) a.variant = 30;   // force the A-direct structure
        if (precision & 32) a.variant = 31;   // force the staged structure
        if (precision & 64) a.variant = 32;   // force the 256-row A-direct tile
    }
    Tmp wpk((precision & 128) ? wino_packed_bytes(Cout, Cin) + 256 : 256);
    if (precision & 128) {                    // F(2,3)-along-W kernel (conv_wino.hip); fails loudly if the shape cannot take it
        DSD_CHECK(conv2d_wino_shape_ok(a), "conv2d: this problem cannot run on the F(2,3) kernel (3x3, stride 1, even width, "
                                           "Cin %% 16 == 0, Cout %% 32 == 0, >= 4096 output pixels, bf16x6)");
        wino_pack_weights(wp.as<float>(), Cout, Cin, wpk.p, s);
        a.w_wino = wpk.p;
    }
    Tmp scratch(conv2d_scratch_bytes(a));
    a.scratch = scratch.as<float>();
    a.scratch_bytes = conv2d_scratch_bytes(a);
    conv2d(a, s);
    int flag = 0;
    DSD_HIP(hipMemcpyAsync(&flag, ovf.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DSD_HIP(hipStreamSynchronize(s));
    DSD_CHECK(!flag, "f16x3: a convolution operand exceeded the fp16 range (|x| > 65504); the result is invalid - use bf16x6 or f32");
    DSD_CATCH
}

int dsd_op_group_norm(const float* x, int N, int HW, int C, const float* gamma, const float* beta, float eps, int silu,
                      float* y, void* stream) {
    DSD_TRY
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = gn_nchunks(HW, C);
    Tmp part((size_t)N * nchunk * C * 2 * sizeof(double)), sc((size_t)N * C * sizeof(float)), sh((size_t)N * C * sizeof(float));
    gn_stats(x, N, HW, C, part.as<double>(), nchunk, s);
    GnSrc s0;
    s0.p = part.as<double>(); s0.chunks = nchunk; s0.c0 = 0; s0.c = C;
    gn_finalize(s0, GnSrc{}, N, HW, C, gamma, beta, eps, nullptr, 0, sc.as<float>(), sh.as<float>(), s);
    affine_act(x, N, HW, C, sc.as<float>(), sh.as<float>(), silu ? ACT_SILU : ACT_NONE, y, s);
    DSD_HIP(hipStreamSynchronize(s));
    DSD_CATCH
}

int dsd_op_gn_silu_conv_out1(const float* x, int N, int H, int W, int C, const float* gamma, const float* beta, float eps,
                             const float* w_oihw, const float* bias, float* y, void* stream) {
    DSD_TRY
    hipStream_t s = (hipStream_t)stream;
    DSD_CHECK(conv_out1_ok(C, 1, 3, 1), "gn_silu_conv_out1: %d input channels unsupported (a multiple of 64 up to 320)", C);
    const int HW = H * W, nchunk = gn_nchunks(HW, C);
    Tmp part((size_t)N * nchunk * C * 2 * sizeof(double)), sc((size_t)N * C * sizeof(float)), sh((size_t)N * C * sizeof(float));
    Tmp wp((size_t)C * 9 * sizeof(float));
    pack_ohwi(w_oihw, wp.as<float>(), 1, C, 3, s);
    gn_stats(x, N, HW, C, part.as<double>(), nchunk, s);
    GnSrc s0;
    s0.p = part.as<double>(); s0.chunks = nchunk; s0.c0 = 0; s0.c = C;
    gn_finalize(s0, GnSrc{}, N, HW, C, gamma, beta, eps, nullptr, 0, sc.as<float>(), sh.as<float>(), s);
    ConvOut1Args a;
    a.x = x; a.N = N; a.H = H; a.W = W; a.C = C; a.scale = sc.as<float>(); a.shift = sh.as<float>(); a.w = wp.as<float>();
    a.bias = bias; a.y = y;
    conv_out1(a, s);
    DSD_HIP(hipStreamSynchronize(s));
    DSD_CATCH
}

int dsd_op_qkv_attention(const float* qkv, int N, int T, int C, int heads, int new_order, int split, float* out, void* stream) {
    DSD_TRY
    DSD_CHECK(heads > 0 && C % heads == 0, "C=%d not divisible by heads=%d", C, heads);
    const int d = C / heads;
    AttnArgs a;
    a.N = N; a.Tq = a.Tk = T; a.heads = heads; a.d = d;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
    a.scale_q = a.scale_k = 1.f / std::sqrt(std::sqrt((float)d));
    if (new_order) {
        a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C;
        a.q_hs = a.k_hs = a.v_hs = d;
    } else {
        a.q = qkv; a.k = qkv + d; a.v = qkv + 2 * d;
        a.q_hs = a.k_hs = a.v_hs = 3 * d;
    }
    a.out = out;
    a.split = split != 0;
    attention(a, (hipStream_t)stream);
    DSD_CATCH
}

int dsd_op_gemm_half(const float* x, const float* w, const float* bias, int M, int N, int K, int bf16, int epi, const float* gate,
                     int T, float* y, void* stream) {
    DSD_TRY
    DSD_CHECK(x && w && y && epi >= 0 && epi <= 2 && (epi != 2 || (gate && T >= 1)), "bad argument");
    hipStream_t s = (hipStream_t)stream;
    Tmp x16((size_t)M * K * 2), w16((size_t)N * K * 2), y16((size_t)M * N * 2);
    cast16(x, (int64_t)M * K, x16.p, bf16, s);
    cast16(w, (int64_t)N * K, w16.p, bf16, s);
    Gemm16Args a;
    a.x = x16.p; a.ldx = K; a.w = w16.p; a.bias = bias; a.M = M; a.N = N; a.K = K; a.bf16 = bf16; a.epi = epi;
    a.y16 = y16.p; a.ldy = N;
    if (epi == EPI16_GATED) {
        a.x32 = y; a.ldx32 = N; a.gate = gate; a.gate_stride = N; a.T = T;
    }
    gemm16(a, s);
    if (epi != EPI16_GATED) uncast16(y16.p, (int64_t)M * N, y, bf16, s);
    DSD_CATCH
}

int dsd_bench_gemm_half(int M, int N, int K, int bf16, int epi, int whatif, int iters, float* avg_ms) {
    DSD_TRY
    DSD_CHECK(iters >= 1 && avg_ms && M > 0 && N > 0 && K > 0, "bad argument");
    hipStream_t s = nullptr;
    Tmp xf((size_t)M * K * 4), wf((size_t)N * K * 4), x16((size_t)M * K * 2), w16((size_t)N * K * 2), y16((size_t)M * N * 2), b((size_t)N * 4);
    Tmp x32(epi == EPI16_GATED ? (size_t)M * N * 4 : 256), gate(epi == EPI16_GATED ? (size_t)N * 4 : 256);
    philox_normal(xf.as<float>(), (int64_t)M * K, 1, 0, s);
    philox_normal(wf.as<float>(), (int64_t)N * K, 2, 0, s);
    philox_normal(b.as<float>(), N, 3, 0, s);
    cast16(xf.as<float>(), (int64_t)M * K, x16.p, bf16, s);
    cast16(wf.as<float>(), (int64_t)N * K, w16.p, bf16, s);
    if (epi == EPI16_GATED) {
        DSD_HIP(hipMemsetAsync(x32.p, 0, (size_t)M * N * 4, s));
        philox_normal(gate.as<float>(), N, 4, 0, s);
    }
    Gemm16Args a;
    a.x = x16.p; a.ldx = K; a.w = w16.p; a.bias = b.as<float>(); a.M = M; a.N = N; a.K = K; a.bf16 = bf16; a.epi = epi;
    a.y16 = y16.p; a.ldy = N; a.x32 = x32.as<float>(); a.ldx32 = N; a.gate = gate.as<float>(); a.gate_stride = 0; a.T = M;
    gemm16_whatif(a, whatif, s);
    hipEvent_t e0, e1;
    DSD_HIP(hipEventCreate(&e0));
    DSD_HIP(hipEventCreate(&e1));
    DSD_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) gemm16_whatif(a, whatif, s);
    DSD_HIP(hipEventRecord(e1, s));
    DSD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DSD_CATCH
}

int dsd_bench_attention_half(int N, int T, int C, int heads, int bf16, int whatif, int iters, float* avg_ms) {
    DSD_TRY
    DSD_CHECK(iters >= 1 && avg_ms && heads > 0 && C % heads == 0, "bad argument");
    hipStream_t s = nullptr;
    const int d = C / heads;
    Tmp qf((size_t)N * T * 3 * C * 4), q16((size_t)N * T * 3 * C * 2), o16((size_t)N * T * C * 2);
    philox_normal(qf.as<float>(), (int64_t)N * T * 3 * C, 5, 0, s);
    cast16(qf.as<float>(), (int64_t)N * T * 3 * C, q16.p, bf16, s);
    Attn16Args a;
    a.N = N; a.Tq = a.Tk = T; a.heads = heads; a.d = d;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
    a.q_hs = a.k_hs = a.v_hs = d;
    a.q = q16.p;
    a.k = (const char*)q16.p + (size_t)C * 2;
    a.v = (const char*)q16.p + (size_t)2 * C * 2;
    a.bf16 = bf16;
    a.out = o16.p;
    a.scale_q = 1.4426950408889634f / std::sqrt((float)d);   // scores of unit variance in base 2, as the qkv GEMM's epilogue leaves them
    attention16_whatif(a, whatif, s);
    hipEvent_t e0, e1;
    DSD_HIP(hipEventCreate(&e0));
    DSD_HIP(hipEventCreate(&e1));
    DSD_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) attention16_whatif(a, whatif, s);
    DSD_HIP(hipEventRecord(e1, s));
    DSD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DSD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DSD_CATCH
}

int dsd_op_attention_half(const float* qkv, int N, int T, int C, int heads, int bf16, float thr, float* out, void* stream) {
    DSD_TRY
    DSD_CHECK(qkv && out && heads > 0 && C % heads == 0, "C=%d not divisible by heads=%d", C, heads);
    hipStream_t s = (hipStream_t)stream;
    const int d = C / heads;
    Tmp q16((size_t)N * T * 3 * C * 2), o16((size_t)N * T * C * 2);
    cast16(qkv, (int64_t)N * T * 3 * C, q16.p, bf16, s);
    Attn16Args a;
    a.N = N; a.Tq = a.Tk = T; a.heads = heads; a.d = d;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
    a.q_hs = a.k_hs = a.v_hs = d;
    a.q = q16.p;
    a.k = (const char*)q16.p + (size_t)C * 2;
    a.v = (const char*)q16.p + (size_t)2 * C * 2;
    a.scale_q = 1.4426950408889634f / std::sqrt((float)d);   // scores as base-2 logits
    a.thr = thr;
    a.bf16 = bf16;
    a.out = o16.p;
    attention16(a, s);
    uncast16(o16.p, (int64_t)N * T * C, out, bf16, s);
    DSD_CATCH
}

int dsd_op_timestep_embedding(const void* t, int t_is_float, int N, int dim, const float* freqs, float* y, void* stream) {
    DSD_TRY
    timestep_embedding(t, t_is_float, N, dim, y, (hipStream_t)stream, freqs);
    DSD_CATCH
}

int dsd_op_linear(const float* x, int N, int K, const float* w, const float* bias, int O, int act_in, float* y, void* stream) {
    DSD_TRY
    linear(x, N, K, K, w, bias, O, act_in, y, O, (hipStream_t)stream);
    DSD_CATCH
}

int dsd_op_gaussian_sample(const float* moments, const float* noise, uint64_t philox_seed, int B, int E, int H, int W, float* z,
                           void* stream) {
    DSD_TRY
    DSD_CHECK(moments && z && B >= 0 && E >= 1 && H >= 1 && W >= 1, "bad argument");
    gaussian_sample(moments, noise, philox_seed, B, E, H * W, z, (hipStream_t)stream);
    DSD_CATCH
}

int dsd_op_philox_normal(float* y, int64_t n, uint64_t seed, uint64_t step, void* stream) {
    DSD_TRY
    philox_normal(y, n, seed, step, (hipStream_t)stream);
    DSD_CATCH
}

}  // extern "C"
