"""MI355X-native conditional-DDPM sampling hot path (drop-in for larrybb626/diffusion_models_dsdiff).

Sub-packages mirror the reference's module paths so yaml ``target:`` strings resolve unchanged:
  UNet_DS_Diff.model.DSUnetModel, ldm.util.instantiate_from_config,
  Disc_diff.guided_diffusion.{gaussian_diffusion,respace,script_util}, ldm.models.diffusion.{ddpm,ddim},
  trainers.trainer_ddpm.  All arithmetic runs in libdsdiff.so (hand-written gfx950 HIP kernels).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
