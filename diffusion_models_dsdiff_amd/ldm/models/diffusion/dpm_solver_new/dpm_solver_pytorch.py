"""ldm/models/diffusion/dpm_solver_new/dpm_solver_pytorch.py of the reference is the same solver as
Disc_diff/guided_diffusion/sampler.py; the native build keeps one implementation."""
from .....Disc_diff.guided_diffusion.sampler import (NoiseScheduleVP, model_wrapper, DPM_Solver,  # noqa: F401
                                                    interpolate_fn, expand_dims)
