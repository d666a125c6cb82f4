// net.h — parameter store, workspace planner and launch-plan builder of libdsdiff.
#pragma once
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dsdiff.h"
#include "kernels.h"

namespace dsd {

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    int64_t numel = 0;
    size_t off = 0;      // byte offset into the parameter slab
    bool pack3x3 = false;  // stored OHWI on device
    bool set = false;
    int region = 0;      // 0 general, 1 emb_layers weights (contiguous), 2 emb_layers biases (contiguous)
};

// Activation living in the workspace arena (NHWC).
struct Tn {
    size_t off = (size_t)-1;
    int n = 0, h = 0, w = 0, c = 0;
    bool valid() const { return off != (size_t)-1; }
    int64_t numel() const { return (int64_t)n * h * w * c; }
    size_t bytes() const { return (size_t)numel() * sizeof(float); }
    int hw() const { return h * w; }
};

struct ArenaPlanner {
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;  // sorted by offset
    size_t top = 0, peak = 0;
    size_t alloc(size_t bytes);
    void release(size_t off, size_t bytes);
};

// Per-call bindings read by the plan's closures.
struct IO {
    const float* plane[4] = {nullptr, nullptr, nullptr, nullptr};  // noise, a, al, l input planes ([B,1,H,W] each)
    int64_t plane_bs[4] = {0, 0, 0, 0};
    const void* t = nullptr;
    int t_is_float = 0;
    float* out = nullptr;
    float* const* feats = nullptr;
    const float* x_nchw = nullptr;  // block handles
    const float* aux = nullptr;
    const float* aux2 = nullptr;
};

struct Plan {
    int B = 0, H = 0, W = 0, C = 0;
    int zero_al_l = 0, want_feats = 0, share = 0;
    int aux_len = 0, aux_len2 = 0;
    bool valid = false;
    std::vector<std::function<void(hipStream_t)>> ops;
    std::vector<int> op_kind;        // index into kind_names
    std::vector<double> op_flops;    // algorithmic FLOPs of the op (0 for memory-bound ops)
    std::vector<double> op_bytes;    // algorithmic bytes (read+write) of the op (0 if not tracked)
    std::vector<std::string> kind_names;
    size_t arena_bytes = 0;
    double flops = 0.0;
    int launches = 0;
};

struct dsd_net;
using Net = dsd_net;

}  // namespace dsd

struct dsd_handle {
    int device = 0;
    bool is_block = false;
    int block_kind = -1;
    std::vector<int32_t> iargs;
    dsd_config cfg{};
    // parameters
    std::vector<dsd::Param> params;
    std::unordered_map<std::string, int> pidx;
    char* slab = nullptr;
    size_t slab_bytes = 0;
    float* staging = nullptr;
    size_t staging_bytes = 0;
    size_t emb_w_off = 0, emb_b_off = 0;  // start of the contiguous emb_layers regions
    int64_t emb_total = 0;                 // sum of emb_layers output widths
    // workspace + plan
    char* arena = nullptr;
    size_t arena_cap = 0;
    dsd::Plan plan;
    dsd::IO io;
    // sampling scratch
    float* tbuf = nullptr;     // [B] timesteps (fp32)
    float* mout = nullptr;     // [B,out_ch,H,W]
    float* zplane = nullptr;   // [H*W] zeros
    float* dpm_m = nullptr;    // dsd_sample_dpm: m_k, m_{k-1} [B,H*W] each + thresholds [B]
    float* freqs = nullptr;    // [model_channels/2] optional timestep-embedding frequency table (host-supplied)
    // arithmetic mode of the convolutions (PREC_*): bf16 pieces of each conv weight are made lazily at plan time
    int precision = dsd::PREC_BF16X6;
    // dsd_sample only: evaluate the two all-zero-input streams of the C_in = 2 branch ONCE per step instead of once per
    // slice (same input, same timestep for every slice of the batch -> same result).  Off by default.
    int share_zero_streams = 0;
    std::unordered_map<std::string, void*> wsplit;   // parameter name -> [3][numel] bf16 planes
    int* ovf = nullptr;                              // device flag: fp16 range exceeded in f16x3 mode
    // per-kernel profiling (dsd_profile_*): hipEvents around every op of the plan on the caller's stream
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<double> prof_ms, prof_flops, prof_bytes;   // per kind
    std::vector<int64_t> prof_calls;
    std::vector<std::string> prof_names;
    int prof_runs = 0;
    size_t tbuf_cap = 0, mout_cap = 0, zplane_cap = 0, dpm_m_cap = 0;

    float* P(const std::string& name) const;
    const dsd::Param& PP(const std::string& name) const;
};

namespace dsd {
void net_declare_params(dsd_handle* h);
void net_set_param(dsd_handle* h, const char* name, const float* src, const int64_t* shape, int ndim, int src_is_device,
                   hipStream_t s);
void net_plan(dsd_handle* h, int B, int C, int H, int W, int zero_al_l, int want_feats, int aux_len, int aux_len2,
              int share = 0);
void net_run(dsd_handle* h, hipStream_t s);
// f16x3 only: synchronises the stream and throws if an operand left the fp16 range during the work enqueued so far
void net_check_overflow(dsd_handle* h, hipStream_t s);
void net_free(dsd_handle* h);
}  // namespace dsd
