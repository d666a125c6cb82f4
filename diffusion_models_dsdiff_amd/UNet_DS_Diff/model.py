"""DSUnetModel — drop-in for the reference's UNet_DS_Diff/model.py:171-756, executed by libdsdiff.so.

Same constructor keyword arguments (UNet_DS_Diff/model.py:172-202), same ``state_dict`` names and
shapes (OIHW conv weights), same ``forward(x, timesteps, context=None, y=None)`` returning
``(out, dict)`` (model.py:629-663,751-756).  The nn.Module only *holds* the parameters (so Lightning /
``load_state_dict`` / ``.to()`` behave as usual); every FLOP runs in hand-written gfx950 kernels behind
the C ABI (include/dsdiff.h).  No CPU path: construction fails if the library or the GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib
from .._lib import DsdConfig, check, dptr, lib, stream_ptr

FEAT_KEYS = (("style", 3), ("content", 3), ("anatomy", 2), ("lesion", 2), ("n_style_content", 4))


class _Holder(nn.Module):
    """Nameless container so parameter paths match the reference module tree."""


def _is_zero_init(name: str) -> bool:
    """zero_module() sites of the reference: ResBlock.out_layers[-1] (openaimodel.py:233-235),
    AttentionBlock.proj_out (:460), DSUnetModel.out[-1] (model.py:514)."""
    return (".out_layers.3." in name or name.startswith("out_layers.3.") or ".proj_out." in name
            or name.startswith("proj_out.") or name.startswith("out.2."))


class NativeModule(nn.Module):
    """An nn.Module whose parameters mirror a libdsdiff handle's parameter table."""

    _skip_init = False     # parallel.empty_init(): parameter storage is allocated but not initialised (values arrive by broadcast)
    _zero_sites = True     # apply the reference's zero_module() initialisation sites (U-Net blocks); the VAE has none
    _strip = ""            # prefix of the C-side parameter names that this module's own tree does not carry

    def __init__(self):
        super().__init__()
        self._h = C.c_void_p()
        self._uploaded: Dict[str, tuple] = {}
        self._cname: Dict[str, str] = {}      # python parameter name -> name in the library's table
        self._remote: set = set()              # parameters whose values live in the library only (upload_param)
        self._device_index = 0

    # ---- parameter tree from the C-side table (names identical to the reference state_dict)
    def _build_params(self):
        L = lib()
        n = L.dsd_param_count(self._h)
        name, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        fan_in: Dict[str, int] = {}
        for i in range(n):
            check(L.dsd_param_info(self._h, i, C.byref(name), shape, C.byref(ndim)))
            cname = name.value.decode()
            nm = cname[len(self._strip):] if self._strip and cname.startswith(self._strip) else cname
            self._cname[nm] = cname
            shp = tuple(int(shape[k]) for k in range(ndim.value))
            p = nn.Parameter(torch.empty(shp, dtype=torch.float32), requires_grad=False)
            base = nm.rsplit(".", 1)[0]
            with torch.no_grad():
                if NativeModule._skip_init:
                    pass
                elif self._zero_sites and _is_zero_init(nm):
                    p.zero_()
                elif len(shp) >= 2:                      # nn.Conv*/nn.Linear default: kaiming_uniform(a=sqrt(5))
                    fi = int(torch.tensor(shp[1:]).prod())
                    fan_in[base] = fi
                    p.uniform_(-1.0 / math.sqrt(fi), 1.0 / math.sqrt(fi))
                elif nm.endswith(".bias") and base in fan_in:
                    p.uniform_(-1.0 / math.sqrt(fan_in[base]), 1.0 / math.sqrt(fan_in[base]))
                elif nm.endswith(".weight"):            # GroupNorm / LayerNorm
                    p.fill_(1.0)
                else:
                    p.zero_()
            mod = self
            parts = nm.split(".")
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Holder())
                mod = mod._modules[part]
            mod.register_parameter(parts[-1], p)

    def set_precision(self, precision: str):
        """Arithmetic of the convolutions: "f32" (fp32 MFMA), "bf16x6" (fp32-grade split-bf16), "bf16x3" (see dsdiff.h).
        The environment variable DSD_PRECISION sets the default of new modules."""
        check(lib().dsd_set_precision(self._h, _lib.PRECISIONS[precision]))
        return self

    def share_zero_streams(self, on: bool = True):
        """Sampling loops only (see include/dsdiff.h: dsd_set_share_zero_streams); off by default."""
        check(lib().dsd_set_share_zero_streams(self._h, int(on)))
        return self

    def use_graph(self, on: bool = True):
        """Replay the captured hipGraph of a denoising step inside sampling loops (default) or launch every kernel from
        the host (include/dsdiff.h: dsd_set_graph)."""
        check(lib().dsd_set_graph(self._h, int(on)))
        return self

    def winograd(self, on: bool = True):
        """bf16x6 only: F(2,3)-along-W kernel for the large 3x3 stride-1 convolutions (include/dsdiff.h: dsd_set_winograd)."""
        check(lib().dsd_set_winograd(self._h, int(on)))
        return self

    def fuse_gn_stats(self, on: bool = True):
        """GroupNorm statistics from the producing kernel's epilogue (default) or from the standalone pass."""
        check(lib().dsd_set_fuse_gn_stats(self._h, int(on)))
        return self

    def fuse_gn_apply(self, on: bool = True):
        """GroupNorm + SiLU applied by the consuming 3x3 convolution while it stages its input (default) or by the apply pass."""
        check(lib().dsd_set_fuse_gn_apply(self._h, int(on)))
        return self

    def stream_lanes(self, on=True, max_pixels: int = 0):
        """Run the four encoder streams concurrently on four HIP streams where their layers are too small to fill the chip
        (default) or one after the other (include/dsdiff.h: dsd_set_stream_lanes).  The lanes' plan sizes those layers for a
        quarter of the chip each: equal to the sequential plan to fp32 rounding; ``on=2`` runs the lanes' plan on one stream
        (bit-identical to ``on=True``: the dependency check of the tests)."""
        check(lib().dsd_set_stream_lanes(self._h, 2 if on == 2 else int(bool(on)), int(max_pixels)))
        return self

    def graph_stats(self):
        c, l = C.c_int(), C.c_int()
        check(lib().dsd_graph_stats(self._h, C.byref(c), C.byref(l)))
        return {"captures": c.value, "launches": l.value}

    def set_slice_ids(self, ids=None):
        """Global slice index of every row of the next sampling batches: keys the on-device noise by slice, not by batch
        position (include/dsdiff.h: dsd_set_slice_ids).  None clears."""
        ids = [] if ids is None else [int(i) for i in ids]
        arr = (C.c_int64 * max(1, len(ids)))(*ids)
        check(lib().dsd_set_slice_ids(self._h, arr, len(ids)))
        return self

    @property
    def precision(self) -> str:
        code = lib().dsd_get_precision(self._h)
        return {v: k for k, v in _lib.PRECISIONS.items()}[code]

    # ---- change detection ---------------------------------------------------------------------------
    # The contract: (data_ptr, _version, device) is compared on every forward — it catches load_state_dict, .to(), optimiser
    # steps and every in-place op on the parameter itself, and costs no device work.  Edits through ``p.data``
    # (``p.data.copy_(ema)``, ``p.data.normal_()`` — the idiom of the reference's LitEma.copy_to / ema_scope weight swaps,
    # ldm/modules/ema.py) bump neither, so they must be announced with mark_dirty() / sync_params(force=True).
    # As a safety net for HOST-resident parameters a sparse value fingerprint (three elements per parameter, read through
    # their address: no tensor op) is compared as well while ``detect_data_edits`` is True (default): a swap or
    # re-initialisation changes every element and cannot hide from it; an edit that leaves those three elements untouched
    # (a masked / sliced write) is NOT seen — the contract above stands.  CUDA-resident parameters are never
    # fingerprinted: reading them back is a device synchronisation on every forward.
    detect_data_edits = True

    @staticmethod
    def _probe_index(numel):
        return (0, numel // 2, numel - 1) if numel > 0 else ()

    def _fingerprints(self):
        return self._fingerprints_of(self.named_parameters())

    def _fingerprints_of(self, named):
        """{name: (v0, v_mid, v_last)} for host-resident fp32 parameters (read through their address), () for everything else
        (CUDA-resident parameters, or detect_data_edits switched off)."""
        out = {}
        for nm, p in named:
            n = p.numel()
            if self.detect_data_edits and n and p.device.type == "cpu" and p.dtype == torch.float32 and p.is_contiguous():
                base = p.data_ptr()
                out[nm] = tuple(C.c_float.from_address(base + 4 * i).value for i in self._probe_index(n))
            else:
                out[nm] = ()
        return out

    def upload_param(self, nm: str, t: torch.Tensor):
        """Upload ONE parameter from `t` (any device) into the library without touching the module's own storage, and
        remember the module-side state as uploaded — the receiving end of parallel.broadcast_params_bucketed: on ranks
        other than the source the torch-side tensors stay uninitialised, the library holds the values."""
        p = dict(self.named_parameters())[nm]
        assert tuple(t.shape) == tuple(p.shape), (nm, tuple(t.shape), tuple(p.shape))
        t = t.detach().float().contiguous()
        shp = (C.c_int64 * max(1, t.dim()))(*t.shape)
        check(lib().dsd_set_param(self._h, self._cname.get(nm, nm).encode(), C.c_void_p(t.data_ptr()), shp, t.dim(), int(t.is_cuda),
                                  stream_ptr() if t.is_cuda else None))
        # the module-side storage may be uninitialised memory (parallel.empty_init): it is identified by pointer + version
        # only, never by its values
        self._uploaded[nm] = (p.data_ptr(), p._version, p.device.type, "remote")
        self._remote.add(nm)

    def mark_dirty(self, names=None):
        """Force the next sync_params() to re-upload `names` (default: every parameter)."""
        for nm in (list(self._uploaded) if names is None else names):
            self._uploaded.pop(nm, None)

    def sync_params(self, force: bool = False):
        """Upload parameters that changed since the last upload: load_state_dict, .to(), in-place ops on the parameter
        (version counter); whole-tensor ``.data`` rewrites of host parameters are caught by the fingerprint safety net, any
        other ``.data`` edit needs mark_dirty() first (see above).  ``force=True`` re-uploads everything unconditionally."""
        L = lib()
        fps = self._fingerprints_of([(nm, p) for nm, p in self.named_parameters() if nm not in self._remote])
        for nm, p in self.named_parameters():
            if nm in self._remote:
                if self._uploaded.get(nm) == (p.data_ptr(), p._version, p.device.type, "remote"):
                    continue                   # (also under force: the module-side storage holds no values to upload)
                self._remote.discard(nm)       # the module-side tensor was written after all: it is the truth again
                fps.update(self._fingerprints_of([(nm, p)]))
            key = (p.data_ptr(), p._version, p.device.type, fps[nm])
            old = self._uploaded.get(nm)
            # a side without a fingerprint (CUDA-resident, or detect_data_edits switched off / on in between) compares
            # pointer + version + device only: toggling the safety net must not by itself re-upload anything
            if not force and old is not None and old[:3] == key[:3] and (old[3] == key[3] or old[3] == () or key[3] == ()):
                if key[3] != ():
                    self._uploaded[nm] = key
                continue
            t = p.detach()
            if t.dtype != torch.float32:
                t = t.float()
            t = t.contiguous()
            shp = (C.c_int64 * max(1, t.dim()))(*t.shape)
            check(L.dsd_set_param(self._h, self._cname.get(nm, nm).encode(), C.c_void_p(t.data_ptr()), shp, t.dim(), int(t.is_cuda),
                                  stream_ptr() if t.is_cuda else None))
            self._uploaded[nm] = key
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()

    # ---- introspection used by bench.py / tests
    def plan_info(self):
        L = lib()
        return {"workspace_bytes": int(L.dsd_workspace_bytes(self._h)), "launches": int(L.dsd_plan_launches(self._h)),
                "flops": float(L.dsd_plan_flops(self._h)), "device_bytes": int(L.dsd_device_bytes(self._h))}

    def profile(self, on: bool):
        """Per-kernel hipEvent timing of every op of the plan (dsd_profile_enable)."""
        check(lib().dsd_profile_enable(self._h, int(on)))

    def profile_report(self):
        """{kernel: {ms, flops, bytes, calls}} summed over the forwards run since profile(True); 'runs' = #forwards."""
        L = lib()
        out, runs = {}, 0
        kind, ms, fl, by = C.c_char_p(), C.c_double(), C.c_double(), C.c_double()
        calls, r = C.c_int64(), C.c_int()
        for i in range(L.dsd_profile_count(self._h)):
            check(L.dsd_profile_get(self._h, i, C.byref(kind), C.byref(ms), C.byref(fl), C.byref(by), C.byref(calls),
                                    C.byref(r)))
            out[kind.value.decode()] = {"ms": ms.value, "flops": fl.value, "bytes": by.value, "calls": calls.value}
            runs = r.value
        return out, runs

    def profile_ops(self):
        """[(kind, ms, flops, bytes, layer)] per op of the plan in launch order, for the last profiled forward."""
        L = lib()
        kind, ms, fl, by, nm = C.c_char_p(), C.c_double(), C.c_double(), C.c_double(), C.c_char_p()
        out = []
        for i in range(L.dsd_profile_op_count(self._h)):
            check(L.dsd_profile_op_get(self._h, i, C.byref(kind), C.byref(ms), C.byref(fl), C.byref(by)))
            check(L.dsd_profile_op_name(self._h, i, C.byref(nm)))
            out.append((kind.value.decode(), ms.value, fl.value, by.value, (nm.value or b"").decode()))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().dsd_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class DSUnetModel(NativeModule):
    def __init__(self,
                 image_size=None,
                 in_channels=1,
                 model_channels=320,
                 out_channels=1,
                 num_res_blocks=2,
                 attention_resolutions=(),
                 dropout=0,
                 channel_mult=(1, 2, 4, 8),
                 conv_resample=True,
                 dims=2,
                 num_classes=None,
                 use_checkpoint=False,
                 use_fp16=False,
                 use_bf16=False,
                 num_heads=-1,
                 num_head_channels=-1,
                 num_heads_upsample=-1,
                 use_scale_shift_norm=False,
                 resblock_updown=False,
                 use_new_attention_order=False,
                 use_spatial_transformer=False,
                 transformer_depth=1,
                 context_dim=None,
                 n_embed=None,
                 legacy=True,
                 disable_self_attentions=None,
                 num_attention_blocks=None,
                 disable_middle_self_attn=False,
                 use_linear_in_transformer=False,
                 adm_in_channels=None,
                 device_index: int = 0):
        super().__init__()
        # the same argument checks as the reference ctor (model.py:204-219,225-231)
        if use_spatial_transformer:
            assert context_dim is not None, "context_dim must be provided for spatial transformer"
            raise NotImplementedError("use_spatial_transformer=True raises NameError in the reference too "
                                      "(UNet_DS_Diff/model.py:20 imports only SpatialTransformer_fft)")
        if context_dim is not None:
            assert use_spatial_transformer, "context_dim is only meaningful with use_spatial_transformer=True"
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        if num_classes is not None or n_embed is not None:
            raise NotImplementedError("class-conditional / codebook heads are not on the sampling hot path")
        if dims != 2 or not conv_resample or dropout != 0 or use_fp16 or use_bf16:
            raise NotImplementedError("hot path is dims=2, conv_resample=True, dropout=0, fp32 (SURVEY.md 8)")
        if disable_self_attentions is not None or num_attention_blocks is not None:
            raise NotImplementedError("disable_self_attentions / num_attention_blocks are unused by every shipped yaml")
        channel_mult = list(channel_mult)
        if isinstance(num_res_blocks, int):
            nrb = len(channel_mult) * [num_res_blocks]
        else:
            nrb = list(num_res_blocks)
            if len(nrb) != len(channel_mult):
                raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                                 "as a list/tuple (per-level) with the same length as channel_mult")
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = nrb
        self.attention_resolutions = list(attention_resolutions)
        self.channel_mult = channel_mult
        self.num_classes = None
        self.dtype = torch.float32
        self._device_index = device_index

        _lib.require_gpu(device_index)
        cfg = DsdConfig()
        cfg.in_channels, cfg.model_channels, cfg.out_channels = in_channels, model_channels, out_channels
        cfg.n_levels = len(channel_mult)
        for i, (m, r) in enumerate(zip(channel_mult, nrb)):
            cfg.channel_mult[i], cfg.num_res_blocks[i] = int(m), int(r)
        cfg.n_attention_resolutions = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            cfg.attention_resolutions[i] = int(a)
        cfg.num_heads, cfg.num_head_channels, cfg.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        cfg.use_scale_shift_norm = int(bool(use_scale_shift_norm))
        cfg.resblock_updown = int(bool(resblock_updown))
        cfg.use_new_attention_order = int(bool(use_new_attention_order))
        cfg.legacy = int(bool(legacy))
        self._cfg = cfg
        check(lib().dsd_create(C.byref(cfg), device_index, C.byref(self._h)))
        import os
        if os.environ.get("DSD_PRECISION"):
            self.set_precision(os.environ["DSD_PRECISION"])
        self._build_params()
        # timestep_embedding's frequency table exactly as the reference evaluates it (util.py:172-174, torch CPU fp32 exp)
        half = model_channels // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half).contiguous()
        check(lib().dsd_set_timestep_freqs(self._h, C.c_void_p(freqs.data_ptr()), half))
        self._half = int(int(channel_mult[0] * model_channels) * channel_mult[-1] / 2)

    # ---- reference API ---------------------------------------------------------------------------
    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """model.py:629-756.  x: [B,C,H,W] fp32 CUDA with C in {2,4}; timesteps: [B] int64 or float."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        out, feats = self._run(x, timesteps, want_feats=True)
        return out, feats

    @torch.no_grad()
    def _run(self, x, timesteps, want_feats: bool):
        if not x.is_cuda:
            raise _lib.DsdError("DSUnetModel runs on the MI355X only: input tensor is on the CPU (no CPU fallback)")
        self.sync_params()
        x = x.float().contiguous()
        B, Cc, H, W = x.shape
        t = timesteps.to(x.device)
        t_is_float = t.dtype.is_floating_point
        t = (t.float() if t_is_float else t.long()).contiguous()
        assert t.shape == (B,)
        out = torch.empty((B, self.out_channels, H, W), device=x.device, dtype=torch.float32)
        feats_t: List[torch.Tensor] = []
        fptr = None
        if want_feats:
            ds = 2 ** (len(self.channel_mult) - 1)
            feats_t = [torch.empty((B, self._half, H // ds, W // ds), device=x.device, dtype=torch.float32)
                       for _ in range(14)]
            fptr = (C.c_void_p * 14)(*[f.data_ptr() for f in feats_t])
        check(lib().dsd_forward(self._h, dptr(x), C.c_void_p(t.data_ptr()), int(t_is_float), B, Cc, H, W, dptr(out),
                                fptr, stream_ptr()))
        feats = {}
        if want_feats:
            i = 0
            for k, n in FEAT_KEYS:
                feats[k] = feats_t[i:i + n]
                i += n
        return out, feats

    def convert_to_fp16(self):
        raise NotImplementedError("the hot path is fp32 end-to-end (SURVEY.md 9, quirk 8)")

    def convert_to_fp32(self):
        return None
