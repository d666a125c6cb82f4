from .sampler import DPMSolverSampler  # noqa: F401  (trainers/trainer_ddpm.py:19 imports it from the package)
