"""instantiate_from_config / get_obj_from_str — drop-in for ldm/util.py:82-97.

The yaml ``target:`` strings of the reference (e.g. ``UNet_DS_Diff.model.DSUnetModel``) resolve to the
native classes of this package, so the reference's model yamls work unchanged.
"""
import importlib

_PKG = __name__.split(".")[0]
# reference dotted paths served natively
_NATIVE_PREFIXES = ("UNet_DS_Diff.model", "ldm.util", "ldm.models.diffusion.ddpm", "ldm.models.diffusion.ddim",
                    "Disc_diff.guided_diffusion", "trainers.trainer_ddpm")


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) and not isinstance(d, type) else d


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if any(module == p or module.startswith(p + ".") for p in _NATIVE_PREFIXES):
        module = _PKG + "." + module
    module_imp = importlib.import_module(module)
    if reload:
        importlib.reload(module_imp)
    return getattr(module_imp, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_unet_from_yaml(path, **overrides):
    """Build the U-Net named by ``model.params.unet_config`` of a reference model yaml (PyYAML; the hot-path
    yamls use no OmegaConf interpolation — SURVEY.md 5)."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    uc = dict(cfg["model"]["params"]["unet_config"])
    uc["params"] = {**uc.get("params", {}), **overrides}
    return instantiate_from_config(uc), cfg
