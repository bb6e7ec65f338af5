"""Evaluation loop of the reference's experiment driver on the MI355X backend
(`AD/experiments/main.py:251-314`, mode "eval"): for every test batch build the condition with the likelihood, draw x_T,
run the conditional sampler, save sample / condition / ground-truth images, collect per-sample metrics and write
`results.json` with the reference's keys (`mse_mean`, `mse_median`, `mse_std`, `lpips_*`, `fid`).

LPIPS and FID need downloaded network weights (`lpips.LPIPS(net='vgg')`, torchmetrics' Inception: SURVEY 8c lists both as
unavailable offline), so they are hooks: pass `lpips_fn(x0, batch) -> [B,1,1,1]` and / or a `fid` object with
`update(img_uint8ish, real=bool)` / `compute()`; without them the keys are written as `None`.  The MSE is the HIP kernel
`mi355_mse_per_sample`; the sampler is whatever `get_conditional_sample_fn` returned (HIP loop).
"""
from __future__ import annotations

import json
import os

import torch

from mi355.imageio import save_image
from mi355.ops import default_ops


def to_img(x):
    """[-1, 1] -> [0, 1] (the reference's `to_img` helper used for image dumps and FID updates)."""
    return default_ops.to_unit_range(x.float().contiguous()) if x.is_cuda else x.float().clip(-1, 1) / 2 + 0.5


def evaluate(cond_sample_fn, likelihood, batches, experiment_dir, num_batches=None, lpips_fn=None, fid=None, save_images=True,
             noise_fn=torch.randn_like):
    """batches: iterable of image tensors [B,C,H,W] in [-1,1] already on the sampling device (or (images, labels) pairs).
    Returns the `results` dict that is also written to `<experiment_dir>/results.json`."""
    gen_dir = os.path.join(str(experiment_dir), "generated")
    gt_dir = os.path.join(str(experiment_dir), "generated_groundtruth")
    os.makedirs(gen_dir, exist_ok=True)
    os.makedirs(gt_dir, exist_ok=True)
    metrics = {"mse": [], "lpips": []}
    idx = 0
    for k, batch in enumerate(batches):
        if num_batches is not None and k >= num_batches:
            break
        if isinstance(batch, (tuple, list)):
            batch = batch[0]
        batch = batch.float().contiguous()
        test_condition = likelihood.sample(batch)
        pad = getattr(likelihood, "pad_value", None)
        cond_plt = test_condition if pad is None else torch.where(test_condition == pad, 1.0, test_condition)
        xT = noise_fn(batch)
        x0 = cond_sample_fn(xT, test_condition)
        if fid is not None:
            fid.update(to_img(x0), real=False)
        if save_images:
            imgs, conds, trues = to_img(x0).cpu(), to_img(cond_plt).cpu(), to_img(batch).cpu()
            for im_sample, cond, im_true in zip(imgs, conds, trues):
                save_image(im_sample, os.path.join(gen_dir, f"image_{str(idx).zfill(3)}.png"), nrow=1, padding=0)
                save_image(cond, os.path.join(gt_dir, f"image_gt_{str(idx).zfill(3)}.png"), nrow=1, padding=0)
                save_image(im_true, os.path.join(gt_dir, f"image_gt2_{str(idx).zfill(3)}.png"), nrow=1, padding=0)
                idx += 1
        metrics["mse"].append(default_ops.mse_per_sample(x0.float().contiguous(), batch))
        if lpips_fn is not None:
            metrics["lpips"].append(lpips_fn(x0, batch).reshape(-1).float())
    results = {}
    for name, vals in metrics.items():
        if vals:
            v = torch.cat(vals, dim=0)
            results.update({f"{name}_mean": torch.mean(v).item(), f"{name}_median": torch.median(v).item(), f"{name}_std": torch.std(v).item()})
        else:
            results.update({f"{name}_mean": None, f"{name}_median": None, f"{name}_std": None})
    results["fid"] = fid.compute().item() if fid is not None else None
    # key order of the reference: means, medians, stds, fid
    ordered = {k: results[k] for suffix in ("_mean", "_median", "_std") for k in (f"mse{suffix}", f"lpips{suffix}")}
    ordered["fid"] = results["fid"]
    with open(os.path.join(str(experiment_dir), "results.json"), "w") as f:
        json.dump(ordered, f)
    return ordered


# ---- offline Frechet proxy -------------------------------------------------------------------------
# The reference scores samples with cleanfid (cifar10/compute_fid.py:92-100), which downloads Inception weights and CIFAR
# statistics - unavailable offline.  The stand-in keeps FID's formula and replaces the feature extractor by a FIXED, seeded
# random convolutional network shipped here (three 3x3 conv + ReLU + 2x2 average-pool stages, global mean and global max pooling
# -> 256 features).  It measures how far two SAMPLE SETS are apart in that feature space (e.g. bf16-mode vs fp32-mode samples
# from identical x0); its scale is not comparable with a true FID, so it is always reported next to a same-distribution floor
# (two independent sets of the same sampler).  Metric-side host code (PyTorch-CPU), not part of the sampling hot path.

def random_conv_features(images_u8: torch.Tensor, seed: int = 0, chunk: int = 2048) -> torch.Tensor:
    """uint8 [N, C, H, W] -> float64 [N, 256] features of the seeded random conv net (CPU)."""
    import numpy as np
    import torch.nn.functional as F

    x = images_u8.detach().to("cpu")
    C = x.shape[1]
    rs = np.random.RandomState(seed)
    widths = [C, 32, 64, 128]
    ws = [torch.from_numpy((rs.standard_normal((widths[i + 1], widths[i], 3, 3)) * np.sqrt(2.0 / (9 * widths[i]))).astype(np.float32))
          for i in range(3)]
    feats = []
    with torch.no_grad():
        for s in range(0, x.shape[0], chunk):
            h = x[s:s + chunk].float() / 127.5 - 1.0
            for i, w in enumerate(ws):
                h = F.relu(F.conv2d(h, w, padding=1))
                if i < 2:
                    h = F.avg_pool2d(h, 2)
            feats.append(torch.cat((h.mean(dim=(2, 3)), h.amax(dim=(2, 3))), dim=1).double())
    return torch.cat(feats, dim=0)


def frechet_distance(f1: torch.Tensor, f2: torch.Tensor) -> float:
    """|mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2)) of two feature sets [N, D] (the FID formula)."""
    import numpy as np

    a, b = f1.double().numpy(), f2.double().numpy()
    mu1, mu2 = a.mean(0), b.mean(0)
    s1, s2 = np.cov(a, rowvar=False), np.cov(b, rowvar=False)
    # Tr((S1 S2)^(1/2)) = sum of sqrt of the eigenvalues of S1^(1/2) S2 S1^(1/2) (symmetric PSD: numerically stable)
    w, v = np.linalg.eigh(s1)
    r1 = (v * np.sqrt(np.clip(w, 0, None))) @ v.T
    ev = np.linalg.eigvalsh(r1 @ s2 @ r1)
    tr = np.sqrt(np.clip(ev, 0, None)).sum()
    return float(((mu1 - mu2) ** 2).sum() + np.trace(s1) + np.trace(s2) - 2.0 * tr)


def frechet_proxy(images_a_u8: torch.Tensor, images_b_u8: torch.Tensor, seed: int = 0) -> float:
    return frechet_distance(random_conv_features(images_a_u8, seed), random_conv_features(images_b_u8, seed))
