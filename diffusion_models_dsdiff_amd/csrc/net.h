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

// GroupNorm statistics that travel with a tensor: per-column partial sums (kernels.h GnSrc) of its channels [c0, c0+c),
// written by the op that produced those channels.
struct StatRef {
    size_t off = (size_t)-1;   // arena offset of [n][chunks][c][2] doubles
    int chunks = 0, c0 = 0, c = 0;
    bool valid() const { return off != (size_t)-1; }
    size_t bytes(int n) const { return (size_t)n * chunks * c * 2 * sizeof(double); }
};

// Activation living in the workspace arena (NHWC).
struct Tn {
    size_t off = (size_t)-1;
    int n = 0, h = 0, w = 0, c = 0;
    int esz = 4;     // bytes per element: 4 (fp32) everywhere but the 16-bit activations of DSD_PREC_F16 / DSD_PREC_BF16
    StatRef st[2];   // st[0] covers channels [0, st[0].c); st[1] (concatenated tensors) the rest
    bool valid() const { return off != (size_t)-1; }
    int64_t numel() const { return (int64_t)n * h * w * c; }
    size_t bytes() const { return (size_t)numel() * esz; }
    int hw() const { return h * w; }
};

// A GroupNorm whose statistics are finalised (per-(sample, channel) scale / shift in the arena) but not yet applied: either
// the apply pass makes a tensor of it, or the consuming convolution applies it while staging its input (conv2d_fuses_gn).
struct GnRef {
    Tn x;
    size_t scoff = 0, shoff = 0, sbytes = 0;
    int act = 0;
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
    std::vector<signed char> op_lane;   // -1: the caller's stream after joining every lane; 0..3: encoder stream lanes (run concurrently)
    std::vector<std::string> op_name;   // the last parameter looked up before the op was emitted = the layer it belongs to (dsd_profile_op_name)
    std::vector<int> op_kind;        // index into kind_names
    std::vector<double> op_flops;    // algorithmic FLOPs of the op (0 for memory-bound ops)
    std::vector<double> op_bytes;    // algorithmic bytes (read+write) of the op (0 if not tracked)
    std::vector<std::string> kind_names;
    size_t arena_bytes = 0;
    double flops = 0.0;
    int launches = 0;
    int gen = 0;          // bumped every time the plan is rebuilt (a captured graph of an older plan is stale)
    int eager_runs = 0;   // forwards launched from the host since the plan was built (capture waits for the first one)
};

// What a captured forward bakes in besides the plan: every pointer the closures read from dsd_handle::io.
struct GraphKey {
    int plan_gen = -1;
    const void* ptr[12] = {};
    int64_t bs[4] = {};
    int t_is_float = 0;
    bool operator==(const GraphKey& o) const {
        if (plan_gen != o.plan_gen || t_is_float != o.t_is_float) return false;
        for (int i = 0; i < 12; ++i) if (ptr[i] != o.ptr[i]) return false;
        for (int i = 0; i < 4; ++i) if (bs[i] != o.bs[i]) return false;
        return true;
    }
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
    int64_t* slice_ids = nullptr;  // [n_slice_ids] global slice index of every batch row (Philox counter base), optional
    int n_slice_ids = 0;
    // whole-forward hipGraph (sampling loops): captured on cap_stream after the first host-launched forward of a plan,
    // replayed on the caller's stream while plan + bound pointers stay the same
    // one event per stream dsd_set_param has been called on, re-recorded after every upload on that stream: plan-time
    // consumers of the slab (weight splitting, on the planning call's stream) wait on ALL of them, so uploads spread over
    // several non-blocking streams need no synchronisation by the caller
    std::unordered_map<hipStream_t, hipEvent_t> param_evs;
    hipStream_t last_param_stream = nullptr;
    bool any_param_upload = false;
    int use_graph = 0;
    int use_winograd = 0;    // bf16x6 only, opt-in: 3x3 stride-1 convolutions as F(2,3) along the width (conv_wino.hip)
    int fuse_gn_stats = 1;   // GroupNorm statistics from the producing kernel's epilogue (0: always the standalone pass)
    int fuse_gn_apply = 1;   // GroupNorm + SiLU applied by the consuming 3x3 convolution while it stages its input (0: apply pass)
    // The four encoder streams are independent chains (UNet_DS_Diff/model.py:674-686).  Where their layers are too small to fill
    // 256 CUs (pixels per layer <= lane_pixels), the plan tags them with a lane and the executor runs lanes 1..3 on streams of
    // its own between a fork and a join event: four under-filled grids share the chip instead of queueing.  Same kernels,
    // same arithmetic: bit-identical to the sequential order.  dsd_set_stream_lanes.
    int use_lanes = 1;
    int lane_pixels = 16384;
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t lane_fork = nullptr, lane_join[3] = {nullptr, nullptr, nullptr};
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    dsd::GraphKey gkey;
    int graph_launches = 0, graph_captures = 0;
    // arithmetic mode of the convolutions (PREC_*): bf16 pieces of each conv weight are made lazily at plan time
    int precision = dsd::PREC_BF16X6;
    // dsd_sample only: evaluate the two all-zero-input streams of the C_in = 2 branch ONCE per step instead of once per
    // slice (same input, same timestep for every slice of the batch -> same result).  Off by default.
    int share_zero_streams = 0;
    std::unordered_map<std::string, void*> wsplit;   // parameter name (+"#f16") -> [3][numel] bf16 / fp16 planes
    std::unordered_map<std::string, size_t> wsplit_bytes;
    int* ovf = nullptr;                              // device flag: fp16 range exceeded in f16x3 mode
    // per-kernel profiling (dsd_profile_*): hipEvents around every op of the plan on the caller's stream
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<double> prof_ms, prof_flops, prof_bytes;   // per kind
    std::vector<int64_t> prof_calls;
    std::vector<float> prof_op_ms;                         // per op of the plan, last profiled forward
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
// s: the caller's stream — plan-time device work (splitting the weights into bf16 / fp16 pieces) is ordered after the
// parameter uploads the caller enqueued on it
void net_plan(dsd_handle* h, int B, int C, int H, int W, int zero_al_l, int want_feats, int aux_len, int aux_len2,
              int share = 0, hipStream_t s = nullptr);
void net_run(dsd_handle* h, hipStream_t s);
// every op of the plan in order; lane-tagged ops on the lane streams (fork / join through events) unless lanes are off
void net_launch_ops(dsd_handle* h, hipStream_t s);
// the same forward through the captured hipGraph when one is valid for the current plan and bindings (sampling loops)
void net_run_cached(dsd_handle* h, hipStream_t s);
void net_drop_graph(dsd_handle* h);
// frees the weight pieces of the arithmetic family (bf16 / fp16) that `precision` does not use
void net_drop_other_pieces(dsd_handle* h, int precision);
size_t net_piece_bytes(const dsd_handle* h);
// f16x3 only: synchronises the stream and throws if an operand left the fp16 range during the work enqueued so far
void net_check_overflow(dsd_handle* h, hipStream_t s);
void net_free(dsd_handle* h);
}  // namespace dsd
