"""`LocalWriter` of the reference's `AD/image_diffusion/writers.py:291-369` (its copy `mnist/writers.py` is what
`mnist/train_mnist*.py` and `AD/experiments/main.py` log through): the local-disk half of the evaluation harness
(SURVEY.md 8f rank 4).  Same constructor, same methods, same on-disk artefacts:

    <logdir>/<filename>.csv      one row per write_scalars call: pandas frame of {"step": step, **scalars}, flushed every
                                 `flush_every_n` calls and on close(); a flush re-reads the existing file (index_col=0) and
                                 appends, so the index column restarts at every flush exactly like the reference's
    <logdir>/config.yaml         log_hparams: yaml.dump of the mapping
    <logdir>/images/<key>_<step>.png   write_images ((N)CHW arrays in [0,1] or uint8; a 4-D batch becomes a grid) and
                                 write_figures (caller-made matplotlib figures, saved with bbox_inches="tight" and closed)

Image files are encoded directly with PIL (mi355.imageio) rather than drawn through a matplotlib axes at dpi 300: the pixel
content is the array itself.  TensorBoard / Aim / async writers of the reference are out of scope (training-side logging).
Host-side file I/O only; nothing here touches the GPU.
"""
from __future__ import annotations

import os
from typing import Any, Mapping

import numpy as np
import torch


class LocalWriter:
    """MetricWriter that writes files to local disk."""

    def __init__(self, logdir: str, flush_every_n: int = 100, filename: str = "metrics"):
        if not os.path.exists(logdir):
            os.mkdir(logdir)          # like the reference: one level only, a missing parent is the caller's error
        self._logdir = str(logdir)
        self._flush_every_n = int(flush_every_n)
        self._metrics_path = f"{self._logdir}/{filename}.csv"
        self._config_path = f"{self._logdir}/config.yaml"
        self._pending = []
        self._count = 0

    # -- hyper-parameters ------------------------------------------------------------------------------------------
    def log_hparams(self, hparams: Mapping[str, Any]):
        import yaml

        with open(self._config_path, "w") as f:
            f.write(yaml.dump(hparams))

    # -- scalars ---------------------------------------------------------------------------------------------------
    def write_scalars(self, step: int, scalars: Mapping[str, Any]):
        row = {"step": step}
        row.update(scalars)
        self._pending.append(row)
        # the reference's order (writers.py:309-313): test (count + 1) % n first, then count the call; flush() resets the count, so the
        # first flush comes after n calls and every later one after n - 1 (the index column of metrics.csv shows that cadence)
        if (self._count + 1) % self._flush_every_n == 0:
            self.flush()
        self._count += 1

    def flush(self):
        if not self._pending:
            return
        import pandas as pd

        frame = pd.DataFrame(self._pending)
        if os.path.exists(self._metrics_path):
            frame = pd.concat([pd.read_csv(self._metrics_path, index_col=0), frame], axis=0)
        frame.to_csv(self._metrics_path)
        self._pending = []
        self._count = 0

    # -- images ----------------------------------------------------------------------------------------------------
    def _image_dir(self):
        path = f"{self._logdir}/images"
        if not os.path.exists(path):
            os.mkdir(path)
        return path

    def write_images(self, step: int, images: Mapping[str, Any]):
        """(N)CHW with C = 1 or 3, float in [0, 1] or uint8 in [0, 255]; a 4-D batch is laid out as a grid first."""
        from mi355.imageio import make_grid, save_image

        path = self._image_dir()
        for key, value in images.items():
            t = torch.as_tensor(np.asarray(value.detach().cpu()) if isinstance(value, torch.Tensor) else np.asarray(value))
            if t.dtype == torch.uint8:
                t = t.float() / 255.0
            t = t.float()
            if t.dim() == 4:
                t = make_grid(t)
            save_image(t, f"{path}/{key}_{step}.png", nrow=1, padding=0)

    def write_figures(self, step: int, figures: Mapping[str, Any]):
        """Caller-made matplotlib figures (the reference's plots(): mnist/train_mnist.py:290-300)."""
        path = self._image_dir()
        for key, fig in figures.items():
            fig.savefig(f"{path}/{key}_{step}.png", bbox_inches="tight")
            try:
                import matplotlib.pyplot as plt

                plt.close(fig)
            except (ImportError, TypeError):   # no matplotlib, or a duck-typed object that is not a pyplot figure: nothing to close
                pass

    def close(self):
        self.flush()
