"""End-to-end entry point (GPU): python -m diffusion_models_dsdiff_amd.infer_2d with the reference's yaml keys, a
checkpoint carrying the Lightning prefix, .npy slices in / out (ragged last batch), checked against the oracle's DDIM loop."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import samplers as OS, unet as O
from util import golden, fixture_params, rel_l2, cond_image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_infer_2d_ddim_end_to_end(tmp_path):
    gm = golden("model")
    params = json.loads(str(gm["tiny_cfg"]))
    sd = fixture_params(gm, "tiny")
    # model yaml / inference yaml with the reference's keys
    model_yaml = {"model": {"params": {"parameterization": "v", "diffusion_steps": 1000, "noise_schedule": "linear",
                                       "learn_sigma": False, "predict_xstart": False, "rescale_timesteps": False,
                                       "timestep_respacing": "", "clip_denoised": True,
                                       "unet_config": {"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}}}}
    infer_yaml = {"cuda_idx": 0, "test_batch_size": 3, "seed": 2024,
                  "sampler_setting": {"sampler": "ddim", "sample_steps": 10, "ddim_eta": 0, "ddim_use_original_steps": False}}
    (tmp_path / "m.yaml").write_text(yaml.safe_dump(model_yaml))
    (tmp_path / "i.yaml").write_text(yaml.safe_dump(infer_yaml))
    torch.save({"state_dict": {"model.diffusion_model." + k: v for k, v in sd.items()}}, tmp_path / "ckpt.pt")
    n = 5                                                   # ragged: batches of 3 + 2
    cond = cond_image((n, 1, 32, 32), 321)
    xT = torch.randn(n, 1, 32, 32, generator=torch.Generator().manual_seed(5))
    np.save(tmp_path / "in.npy", cond.numpy())
    np.save(tmp_path / "xT.npy", xT.numpy())
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "diffusion_models_dsdiff_amd.infer_2d", "--model-yaml", str(tmp_path / "m.yaml"),
                        "--infer-yaml", str(tmp_path / "i.yaml"), "--input", str(tmp_path / "in.npy"), "--x-T", str(tmp_path / "xT.npy"),
                        "--output", str(tmp_path / "out.npy"), "--ckpt", str(tmp_path / "ckpt.pt")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = np.load(tmp_path / "out.npy")
    assert out.shape == (n, 1, 32, 32) and np.isfinite(out).all()
    # on_predict_start semantics: sample_steps (10) != training steps (1000) -> respacing "10" + rescale_timesteps, DDIM eta 0
    cfg = O.UNetConfig.from_params(params)
    od = OS.DiffusionA(steps=1000, timestep_respacing="10", rescale_timesteps=True, parameterization="v")
    yo = od.ddim_sample_loop(lambda xx, tt: O.unet_forward(cfg, sd, xx, tt)[0], xT, torch.zeros((10, n, 1, 32, 32)), [cond])
    assert rel_l2(out, yo) < 1e-4


def test_infer_2d_h5_slices_to_nifti_volumes(tmp_path):
    """The reference's own file formats either side of the path (SURVEY f-2): <id>/layer_<z>.h5 slice files in
    (preprocess/to_h5.py layout, read by h5lite), one NIfTI volume per id out, slices at their index; the volumes equal what
    the .npy route gives for the same slices (x_T is keyed by the global slice number in both)."""
    from diffusion_models_dsdiff_amd import host_io
    gm = golden("model")
    params = json.loads(str(gm["tiny_cfg"]))
    sd = fixture_params(gm, "tiny")
    model_yaml = {"model": {"params": {"parameterization": "v", "diffusion_steps": 1000, "noise_schedule": "linear",
                                       "unet_config": {"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}}}}
    infer_yaml = {"cuda_idx": 0, "test_batch_size": 4, "seed": 7,
                  "sampler_setting": {"sampler": "ddim", "sample_steps": 4, "ddim_eta": 0}}
    (tmp_path / "m.yaml").write_text(yaml.safe_dump(model_yaml))
    (tmp_path / "i.yaml").write_text(yaml.safe_dump(infer_yaml))
    torch.save(sd, tmp_path / "ckpt.pt")
    depth = {"vol_a": 3, "vol_b": 2}
    cond = cond_image((5, 1, 32, 32), 99).numpy()
    k = 0
    for id_, d in depth.items():
        os.makedirs(tmp_path / "h5" / id_)
        for z in range(d):
            host_io.write_h5(str(tmp_path / "h5" / id_ / f"layer_{z}.h5"), {"F_Data1": cond[k, 0], "S_Data1": np.zeros((32, 32), np.float32)})
            k += 1
    np.save(tmp_path / "in.npy", cond)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, "-m", "diffusion_models_dsdiff_amd.infer_2d", "--model-yaml", str(tmp_path / "m.yaml"),
            "--infer-yaml", str(tmp_path / "i.yaml"), "--ckpt", str(tmp_path / "ckpt.pt")]
    r = subprocess.run(base + ["--input", str(tmp_path / "h5"), "--input-keys", "F_Data1", "--output", str(tmp_path / "pred")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run(base + ["--input", str(tmp_path / "in.npy"), "--output", str(tmp_path / "out.npy")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    flat = np.load(tmp_path / "out.npy")
    k = 0
    for id_, d in depth.items():
        vol, _hdr = host_io.read_nifti(str(tmp_path / "pred" / id_ / "pred.nii.gz"))
        assert vol.shape == (d, 32, 32)
        assert np.array_equal(vol, flat[k:k + d, 0])
        k += d


def test_infer_2d_templates_task_naming_and_metric_csv(tmp_path):
    """ADVICE r2 + f-2 remainder: with --template-dir the volumes take the template's depth (trailing unpredicted slices stay
    zero), dtype and header bytes and --task-id names them <task>_<id>_pred.nii.gz (trainer_use_gaussian_diff.py:639-649);
    --gt-dir runs the metric pass of inference/get_metric_BraTs.py over them and writes the table as CSV.  Sizes: 192x192
    slices (the MS-SSIM of the "ssim" column needs > 160 px; also a width the yamls use that is not a power of two)."""
    import csv
    from diffusion_models_dsdiff_amd import host_io
    gm = golden("model")
    params = json.loads(str(gm["tiny_cfg"]))
    sd = fixture_params(gm, "tiny")
    model_yaml = {"model": {"params": {"parameterization": "v", "diffusion_steps": 1000, "noise_schedule": "linear",
                                       "unet_config": {"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}}}}
    infer_yaml = {"cuda_idx": 0, "test_batch_size": 3, "seed": 11,
                  "sampler_setting": {"sampler": "ddim", "sample_steps": 3, "ddim_eta": 0}}
    (tmp_path / "m.yaml").write_text(yaml.safe_dump(model_yaml))
    (tmp_path / "i.yaml").write_text(yaml.safe_dump(infer_yaml))
    torch.save(sd, tmp_path / "ckpt.pt")
    S = 192
    cond = cond_image((4, 1, S, S), 77).numpy()
    depth_template = {"p01": 4, "p02": 3}                       # predicted: p01 slices 0, 1 ; p02 slices 0, 2
    pred_slices = {"p01": [0, 1], "p02": [0, 2]}
    k = 0
    for id_, zs in pred_slices.items():
        os.makedirs(tmp_path / "h5" / id_)
        os.makedirs(tmp_path / "tpl" / id_)
        for z in zs:
            host_io.write_h5(str(tmp_path / "h5" / id_ / f"t1_{z}.h5"), {"F_Data1": cond[k, 0]})
            k += 1
        host_io.write_h5(str(tmp_path / "h5" / id_ / "stats.h5"), {"F_Data1": np.zeros((2, 2), np.float32)})   # stray file: skipped
        gt = np.random.default_rng(len(id_) + k).uniform(-1, 1, (depth_template[id_], S, S)).astype(np.float32)
        host_io.write_nifti(str(tmp_path / "tpl" / id_ / "F_Data1.nii.gz"), gt, spacing=(3.0, 0.5, 0.5))
        host_io.write_nifti(str(tmp_path / "tpl" / id_ / "ce.nii.gz"), gt, spacing=(3.0, 0.5, 0.5))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "diffusion_models_dsdiff_amd.infer_2d", "--model-yaml", str(tmp_path / "m.yaml"),
                        "--infer-yaml", str(tmp_path / "i.yaml"), "--ckpt", str(tmp_path / "ckpt.pt"), "--input", str(tmp_path / "h5"),
                        "--input-keys", "F_Data1", "--output", str(tmp_path / "pred"), "--template-dir", str(tmp_path / "tpl"),
                        "--task-id", "Task9", "--gt-dir", str(tmp_path / "tpl")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for id_, zs in pred_slices.items():
        vol, hdr = host_io.read_nifti(str(tmp_path / "pred" / f"Task9_{id_}_pred.nii.gz"))
        _, thdr = host_io.read_nifti(str(tmp_path / "tpl" / id_ / "F_Data1.nii.gz"))
        assert vol.shape == (depth_template[id_], S, S) and vol.dtype == np.float32
        assert hdr[76:108] == thdr[76:108] and hdr[252:344] == thdr[252:344]                 # the template's geometry
        for z in range(depth_template[id_]):
            assert bool(np.any(vol[z])) == (z in zs)                                         # unpredicted slices stay zero
    rows = list(csv.reader(open(str(tmp_path / "pred") + "_metric.csv")))
    assert rows[0] == ["ids"] + host_io.METRIC_COLUMNS and [r[0] for r in rows[1:]] == ["0", "p01", "p02"]
    assert all(np.isfinite(float(v)) for r in rows[1:] for i, v in enumerate(r[1:]) if host_io.METRIC_COLUMNS[i] != "mi")


def test_infer_2d_three_ranks_on_one_gpu_match_the_single_process_run(tmp_path):
    """--gpus N for real device memory without N GPUs (the 8-GPU path, rehearsed): three ranks over gloo, all on device 0
    (DSD_INFER_BACKEND / DSD_INFER_SINGLE_DEVICE), a ragged 7-slice input: rank 0 initialises the weights, ranks 1-2 build their
    modules empty and receive them through the bucketed broadcast straight into the library, every rank samples its shard with
    slice-keyed noise, rank 0 gathers.  The result must equal the single-process run bit for bit (the same kernels on the same
    slices: the sharding may not change a value)."""
    gm = golden("model")
    params = json.loads(str(gm["tiny_cfg"]))
    sd = fixture_params(gm, "tiny")
    model_yaml = {"model": {"params": {"parameterization": "v", "diffusion_steps": 1000, "noise_schedule": "linear",
                                       "learn_sigma": False, "predict_xstart": False, "rescale_timesteps": False,
                                       "timestep_respacing": "", "clip_denoised": True,
                                       "unet_config": {"target": "UNet_DS_Diff.model.DSUnetModel", "params": params}}}}
    infer_yaml = {"cuda_idx": 0, "test_batch_size": 2, "seed": 2024,
                  "sampler_setting": {"sampler": "ddpm", "sample_steps": 6}}
    (tmp_path / "m.yaml").write_text(yaml.safe_dump(model_yaml))
    (tmp_path / "i.yaml").write_text(yaml.safe_dump(infer_yaml))
    torch.save({"state_dict": {"model.diffusion_model." + k: v for k, v in sd.items()}}, tmp_path / "ckpt.pt")
    n = 7
    np.save(tmp_path / "in.npy", cond_image((n, 1, 32, 32), 322).numpy())
    base = ["--model-yaml", str(tmp_path / "m.yaml"), "--infer-yaml", str(tmp_path / "i.yaml"), "--input", str(tmp_path / "in.npy"),
            "--ckpt", str(tmp_path / "ckpt.pt")]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "diffusion_models_dsdiff_amd.infer_2d"] + base + ["--output", str(tmp_path / "one.npy")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env3 = dict(env, DSD_INFER_BACKEND="gloo", DSD_INFER_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "diffusion_models_dsdiff_amd.infer_2d"] + base + ["--output", str(tmp_path / "three.npy"), "--gpus", "3"],
                       env=env3, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    one, three = np.load(tmp_path / "one.npy"), np.load(tmp_path / "three.npy")
    assert one.shape == three.shape == (n, 1, 32, 32) and np.isfinite(three).all()
    assert np.array_equal(one, three)
