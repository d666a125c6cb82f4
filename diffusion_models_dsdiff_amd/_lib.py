"""ctypes binding of libdsdiff.so (the C ABI in include/dsdiff.h).

There is NO fallback: if the shared library is missing or no gfx950 device is usable the product
raises.  The oracle under /oracle is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSD_LIBRARY") or os.path.join(_HERE, "libdsdiff.so")   # DSD_LIBRARY: A/B builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

DSD_MAX_LEVELS = 8
DSD_NCOEF = 8
MODE_A_DDPM, MODE_A_DDIM, MODE_B_DDPM, MODE_B_DDIM = 0, 1, 2, 3
PRED_EPS, PRED_X0, PRED_V = 0, 1, 2
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3, "f16": 4, "bf16": 5}
(BLOCK_RES, BLOCK_ATTN, BLOCK_UPSAMPLE, BLOCK_DOWNSAMPLE, BLOCK_DISENTANGLE, BLOCK_SE, BLOCK_CROSSATTN,
 BLOCK_FF_GEGLU, BLOCK_BASIC_TRANSFORMER, BLOCK_SPATIAL_TRANSFORMER, BLOCK_VAE_ENCODER, BLOCK_VAE_DECODER,
 BLOCK_DIT, BLOCK_UNET) = range(14)

# every symbol include/dsdiff.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = [
    "dsd_last_error", "dsd_device_info", "dsd_create", "dsd_destroy", "dsd_param_count", "dsd_param_info",
    "dsd_set_param", "dsd_set_timestep_freqs", "dsd_set_precision", "dsd_get_precision", "dsd_set_share_zero_streams", "dsd_params_ready", "dsd_plan", "dsd_workspace_bytes", "dsd_device_bytes", "dsd_set_graph", "dsd_graph_stats", "dsd_set_fuse_gn_stats", "dsd_set_fuse_gn_apply", "dsd_set_stream_lanes", "dsd_set_winograd", "dsd_set_slice_ids", "dsd_plan_launches", "dsd_plan_flops",
    "dsd_profile_enable", "dsd_profile_count", "dsd_profile_get", "dsd_profile_op_count", "dsd_profile_op_get", "dsd_profile_op_name", "dsd_forward", "dsd_sample", "dsd_op_sampler_update", "dsd_sample_dpm", "dsd_op_dpm_step", "dsd_op_dpm_threshold", "dsd_block_create", "dsd_block_forward", "dsd_bench_conv2d", "dsd_bench_conv2d_stamps", "dsd_bench_mfma_peak", "dsd_conv_plan", "dsd_set_conv_mfma16", "dsd_op_conv2d", "dsd_op_conv2d_prec", "dsd_op_gn_silu_conv_out1",
    "dsd_op_gaussian_sample", "dsd_op_group_norm", "dsd_op_qkv_attention", "dsd_op_gemm_half", "dsd_bench_gemm_half", "dsd_bench_attention_half", "dsd_op_attention_half", "dsd_op_timestep_embedding", "dsd_op_linear", "dsd_op_philox_normal",
]


class DsdConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("model_channels", C.c_int32), ("out_channels", C.c_int32),
        ("n_levels", C.c_int32), ("channel_mult", C.c_int32 * DSD_MAX_LEVELS),
        ("num_res_blocks", C.c_int32 * DSD_MAX_LEVELS), ("n_attention_resolutions", C.c_int32),
        ("attention_resolutions", C.c_int32 * DSD_MAX_LEVELS), ("num_heads", C.c_int32),
        ("num_head_channels", C.c_int32), ("num_heads_upsample", C.c_int32), ("use_scale_shift_norm", C.c_int32),
        ("resblock_updown", C.c_int32), ("use_new_attention_order", C.c_int32), ("legacy", C.c_int32),
    ]


class DsdSchedule(C.Structure):
    _fields_ = [
        ("steps", C.c_int32), ("mode", C.c_int32), ("pred", C.c_int32), ("learned_range", C.c_int32),
        ("clip_denoised", C.c_int32), ("eta", C.c_float), ("coef", C.POINTER(C.c_float)),
        ("t_model", C.POINTER(C.c_float)), ("nonzero", C.POINTER(C.c_int32)),
    ]


class DsdDpmSchedule(C.Structure):
    _fields_ = [
        ("steps", C.c_int32), ("pred", C.c_int32), ("data_pred", C.c_int32), ("thresholding", C.c_int32),
        ("threshold_ratio", C.c_float), ("threshold_max", C.c_float), ("coef", C.POINTER(C.c_float)),
        ("t_input", C.POINTER(C.c_float)), ("order", C.POINTER(C.c_int32)),
    ]


class DsdError(RuntimeError):
    pass


_lib = None


def build(verbose: bool = False) -> str:
    """Compile libdsdiff.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise DsdError("building libdsdiff.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or make -C {CSRC}); there is no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load torch FIRST so both share that one HIP runtime
    # (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available").
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    L.dsd_last_error.restype = C.c_char_p
    L.dsd_device_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.dsd_create.argtypes = [C.POINTER(DsdConfig), i32, C.POINTER(vp)]
    L.dsd_destroy.argtypes = [vp]
    L.dsd_destroy.restype = None
    L.dsd_param_count.argtypes = [vp]
    L.dsd_param_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.dsd_set_param.argtypes = [vp, C.c_char_p, f32p, C.POINTER(C.c_int64), i32, i32, vp]
    L.dsd_set_timestep_freqs.argtypes = [vp, f32p, i32]
    L.dsd_set_precision.argtypes = [vp, i32]
    L.dsd_get_precision.argtypes = [vp]
    L.dsd_set_share_zero_streams.argtypes = [vp, i32]
    L.dsd_params_ready.argtypes = [vp]
    L.dsd_plan.argtypes = [vp, i32, i32, i32, i32]
    L.dsd_workspace_bytes.argtypes = [vp]
    L.dsd_workspace_bytes.restype = i64
    L.dsd_device_bytes.argtypes = [vp]
    L.dsd_device_bytes.restype = i64
    L.dsd_set_graph.argtypes = [vp, i32]
    L.dsd_set_fuse_gn_stats.argtypes = [vp, i32]
    L.dsd_set_fuse_gn_apply.argtypes = [vp, i32]
    L.dsd_set_stream_lanes.argtypes = [vp, i32, i32]
    L.dsd_set_winograd.argtypes = [vp, i32]
    L.dsd_graph_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dsd_set_slice_ids.argtypes = [vp, C.POINTER(C.c_int64), i32]
    L.dsd_plan_launches.argtypes = [vp]
    L.dsd_plan_flops.argtypes = [vp]
    L.dsd_plan_flops.restype = C.c_double
    L.dsd_profile_enable.argtypes = [vp, i32]
    L.dsd_profile_count.argtypes = [vp]
    L.dsd_op_gn_silu_conv_out1.argtypes = [vp, i32, i32, i32, i32, vp, vp, C.c_float, vp, vp, vp, vp]
    L.dsd_profile_op_count.argtypes = [vp]
    L.dsd_profile_op_name.argtypes = [vp, i32, C.POINTER(C.c_char_p)]
    L.dsd_profile_op_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dsd_profile_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    L.dsd_forward.argtypes = [vp, f32p, vp, i32, i32, i32, i32, i32, f32p, C.POINTER(vp), vp]
    L.dsd_sample.argtypes = [vp, C.POINTER(DsdSchedule), f32p, i32, f32p, f32p, C.c_uint64, i32, i32, i32, i32, i32, vp]
    L.dsd_op_sampler_update.argtypes = [C.POINTER(DsdSchedule), i32, f32p, f32p, f32p, C.c_uint64, i32, i32, i32, f32p, vp]
    L.dsd_sample_dpm.argtypes = [vp, C.POINTER(DsdDpmSchedule), f32p, i32, f32p, i32, i32, i32, vp]
    L.dsd_op_dpm_step.argtypes = [C.POINTER(DsdDpmSchedule), i32, f32p, i32, f32p, f32p, f32p, i32, i32, i32, vp]
    L.dsd_op_dpm_threshold.argtypes = [f32p, i32, i32, C.c_float, C.c_float, f32p, f32p, vp]
    L.dsd_block_create.argtypes = [i32, C.POINTER(C.c_int32), i32, i32, C.POINTER(vp)]
    L.dsd_block_forward.argtypes = [vp, f32p, i32, i32, i32, i32, f32p, i32, f32p, i32, f32p, vp]
    L.dsd_op_conv2d.argtypes = [f32p, i32, i32, i32, i32, f32p, f32p, i32, i32, i32, i32, f32p, f32p, f32p, vp]
    L.dsd_bench_conv2d.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    L.dsd_bench_conv2d_stamps.argtypes = [i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_longlong), i32, C.POINTER(C.c_int)]
    L.dsd_bench_mfma_peak.argtypes = [i32, i32, C.c_float, i32, C.POINTER(C.c_float), C.POINTER(C.c_double)]
    L.dsd_conv_plan.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.dsd_op_conv2d_prec.argtypes = [f32p, i32, i32, i32, i32, f32p, f32p, i32, i32, i32, i32, f32p, f32p, i32, f32p, vp]
    L.dsd_op_group_norm.argtypes = [f32p, i32, i32, i32, f32p, f32p, C.c_float, i32, f32p, vp]
    L.dsd_op_qkv_attention.argtypes = [f32p, i32, i32, i32, i32, i32, i32, f32p, vp]
    L.dsd_set_conv_mfma16.argtypes = [i32]
    L.dsd_op_gemm_half.argtypes = [f32p, f32p, f32p, i32, i32, i32, i32, i32, f32p, i32, f32p, vp]
    L.dsd_bench_gemm_half.argtypes = [i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float)]
    L.dsd_bench_attention_half.argtypes = [i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float)]
    L.dsd_op_attention_half.argtypes = [f32p, i32, i32, i32, i32, i32, C.c_float, f32p, vp]
    L.dsd_op_timestep_embedding.argtypes = [vp, i32, i32, i32, f32p, f32p, vp]
    L.dsd_op_linear.argtypes = [f32p, i32, i32, f32p, f32p, i32, i32, f32p, vp]
    L.dsd_op_gaussian_sample.argtypes = [f32p, f32p, C.c_uint64, i32, i32, i32, i32, f32p, vp]
    L.dsd_op_philox_normal.argtypes = [f32p, i64, C.c_uint64, C.c_uint64, vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise DsdError(lib().dsd_last_error().decode("utf-8", "replace"))


def require_gpu(device: int = 0):
    """Fail loudly unless a gfx950 device is usable (no silent eager/CPU fallback)."""
    name = C.create_string_buffer(256)
    ncu, mem = C.c_int(0), C.c_int64(0)
    check(lib().dsd_device_info(device, name, 256, C.byref(ncu), C.byref(mem)))
    return name.value.decode(), ncu.value, mem.value


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    """Device pointer of a contiguous fp32 CUDA tensor (or None)."""
    if t is None:
        return None
    import torch
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous CUDA tensor"
    return C.c_void_p(t.data_ptr())
