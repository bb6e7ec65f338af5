"""CPU: the drop-in boundary against the reference's REAL call sites (constructor kwargs, module names, signatures, return
tuples, checkpoint layouts, rank-consistent x0 draws).  No compute runs here (no GPU): these tests pin the host logic."""
import inspect

import numpy as np
import os
import subprocess
import sys

import pytest
import torch

from tests.conftest import PKG


def test_wrappers_accept_the_reference_constructor_kwargs():
    """mnist/train_mnist.py:262-267, train_mnist2.py:350-355 (InPaintModelWrapper) and train_mnist_hy.py:312-318,
    train_mnist_hy2.py:313-318 (SuperResModelWrapper) pass class_cond=True with num_classes=None: an unconditional net."""
    from torchcfm_compat import InPaintModelWrapper, SuperResModelWrapper, UNetModelWrapper

    m = InPaintModelWrapper(dim=(1, 28, 28), num_channels=32, num_res_blocks=1, num_classes=None, class_cond=True)
    assert m.in_channels == 2 and m.out_channels == 1 and m.channel_mult == (1, 2, 2)
    s = SuperResModelWrapper(dim=(1, 28, 28), num_channels=32, num_res_blocks=1, num_classes=None, class_cond=True)
    assert s.in_channels == 2 and s.out_channels == 1
    s64 = SuperResModelWrapper(dim=(3, 64, 64), num_channels=128, num_res_blocks=1, num_classes=None, class_cond=True)
    assert s64.in_channels == 6 and s64.channel_mult == (1, 2, 3, 4) and s64.attention_resolutions == (4,)
    u = UNetModelWrapper(dim=(2, 28, 28), num_channels=32, num_res_blocks=1, num_classes=None, class_cond=True)   # train_mnist.py:256
    assert u.in_channels == 2 and u.out_channels == 2
    # cifar10/train_cifar10.py:92-101 / compute_fid.py:39-48
    c = UNetModelWrapper(dim=(3, 32, 32), num_res_blocks=2, num_channels=128, channel_mult=[1, 2, 2, 2], num_heads=4, num_head_channels=64,
                         attention_resolutions="16", dropout=0.1)
    assert sum(p.numel() for p in c.parameters()) == 35_746_307   # the known torchcfm / DDPM CIFAR size (SURVEY finding 2)
    with pytest.raises(NotImplementedError):
        UNetModelWrapper(dim=(1, 28, 28), num_channels=32, num_res_blocks=1, num_classes=10, class_cond=True)


REF_SIGNATURES = {
    # module: (positional parameters of generate_samples_eval as in the reference file, length of the returned tuple)
    "utils_mnist": (["model", "test_images", "savedir", "batch_size", "step", "net_"], 2),     # mnist/utils_mnist.py:90,135
    "utils_mnist2": (["model", "test_images", "batch_size", "step", "net_"], 3),               # mnist/utils_mnist2.py:118,138
    "utils_mnist_hy": (["model", "test_images", "batch_size", "step", "net_"], 3),             # mnist/utils_mnist_hy.py:76,98
    "utils_mnist_hy2": (["model", "test_images", "batch_size", "step", "net_"], 3),            # mnist/utils_mnist_hy2.py:148,169
}


@pytest.mark.parametrize("mod", sorted(REF_SIGNATURES))
def test_mnist_modules_export_the_reference_names_and_signatures(mod):
    m = __import__(mod)
    for name in ("ema", "generate_samples", "infiniteloop", "generate_samples_eval"):   # mnist/train_mnist*.py:17
        assert callable(getattr(m, name)), (mod, name)
    sig = inspect.signature(m.generate_samples_eval)
    positional = [p.name for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert positional == REF_SIGNATURES[mod][0]
    d = {p.name: p.default for p in sig.parameters.values()}
    assert d["batch_size"] == 8 and d["step"] == 0 and d["net_"] == "normal"
    gs = [p.name for p in inspect.signature(m.generate_samples).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert gs[:5] == ["model", "parallel", "savedir", "step", "net_"]


def test_mnist2_patch_is_20_pixels_and_mnist_is_14():
    import utils_mnist
    import utils_mnist2

    torch.manual_seed(1)
    x = torch.rand(3, 1, 64, 64)
    for mod, size in ((utils_mnist, 14), (utils_mnist2, 20)):
        c = mod.sample(x)
        for k in range(3):
            assert int((c[k] == -2).sum()) == size * size


class _FakeModel(torch.nn.Module):
    """Records the conditions it is called with; returns a constant field (CPU: only the host plumbing is exercised)."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, x, t, con=None, low_res=None):
        self.seen.append((float(t), (con if con is not None else low_res).clone()))
        return torch.ones_like(x)


class _CpuOps:
    def euler_step_(self, x, v, dt):
        x.add_(v * dt) if v is not x else x.mul_(1 + dt)
        return x

    def clip_(self, x, lo, hi):
        return x.clamp_(lo, hi)


def test_euler_conditional_lets_the_condition_drift_like_the_reference(monkeypatch):
    """mnist/utils_mnist2.py:118-138: the ODE state is cat(x, con) and d(con)/dt = con, so under Euler the model sees
    con_k = con * (1 + dt)^k.  (the condition used to be held constant.)"""
    import utils_mnist

    monkeypatch.setattr(utils_mnist, "default_ops", _CpuOps())
    monkeypatch.setattr(utils_mnist, "device", torch.device("cpu"))
    m = _FakeModel()
    x0 = torch.zeros(2, 1, 4, 4)
    con = torch.full((2, 1, 4, 4), -2.0)
    x, nfe = utils_mnist._euler_conditional(m, x0, con, steps=10)
    assert nfe == 10 and len(m.seen) == 10
    for k, (t, c) in enumerate(m.seen):
        assert abs(t - k / 10) < 1e-6
        torch.testing.assert_close(c, con * (1.1 ** k), rtol=1e-5, atol=1e-6)
    assert torch.equal(con, torch.full((2, 1, 4, 4), -2.0))          # the caller's tensor is untouched
    torch.testing.assert_close(x, torch.ones_like(x0), rtol=1e-5, atol=1e-6)


def test_generate_samples_eval_return_tuples(monkeypatch):
    import utils_mnist
    import utils_mnist2
    import utils_mnist_hy2

    monkeypatch.setattr(utils_mnist, "default_ops", _CpuOps())
    monkeypatch.setattr(utils_mnist, "device", torch.device("cpu"))
    for mod in (utils_mnist2, utils_mnist_hy2):
        monkeypatch.setattr(mod, "device", torch.device("cpu"))
    imgs = torch.rand(8, 1, 28, 28)
    m = _FakeModel()
    m.train()
    out = utils_mnist.generate_samples_eval(m, imgs, "/unused/", step=3, net_="net_model", solver="euler", steps=3)
    assert len(out) == 2 and out[0].shape == (8, 1, 28, 28) and out[1].shape == imgs.shape and m.training
    assert float(out[0].max()) <= 1.0 and int((out[1] == -2).sum()) == 8 * 14 * 14
    with pytest.raises(RuntimeError):   # the reference's own behaviour: a 20-pixel patch cannot be placed in 28 pixels (randint(5, 3))
        utils_mnist2.generate_samples_eval(m, imgs, step=3, net_="net_model", steps=4)
    big = torch.rand(8, 1, 64, 64)
    out = utils_mnist2.generate_samples_eval(m, big, step=3, net_="net_model", steps=4, image_shape=(1, 64, 64))   # active form: Euler
    assert len(out) == 3 and out[2] == 4 and int((out[1] == -2).sum()) == 8 * 20 * 20
    out = utils_mnist_hy2.generate_samples_eval(m, imgs, solver="euler", steps=2)
    assert len(out) == 3 and out[1].shape == (8, 1, 7, 7) and out[0].shape == (8, 1, 28, 28)


def test_ema_invalidates_the_packed_engine():
    """ema() writes through .data / a raw-pointer kernel, which bumps no version counter; the packed-weight cache
    must be dropped explicitly or `ema(net, ema_model, d); generate_samples(ema_model, ...)` samples from stale weights."""
    import utils_cifar
    import utils_mnist
    from torchcfm_compat import UNetModelWrapper

    kw = dict(dim=(1, 16, 16), num_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions="8")
    src, tgt = UNetModelWrapper(**kw), UNetModelWrapper(**kw)
    for ema in (utils_cifar.ema, utils_mnist.ema):
        tgt._engine = object()   # stands for a packed engine
        gen = tgt._weights_generation
        before = tgt.state_dict()["time_embed.0.weight"].clone()
        ema(src, tgt, 0.5)
        assert tgt._engine is None and tgt._weights_generation == gen + 1
        assert not torch.equal(tgt.state_dict()["time_embed.0.weight"], before)


def test_checkpoint_layouts(tmp_path, capsys):
    """SURVEY 8(b) checkpoints: (i) {"net_model","ema_model",...} with optional "module." prefix (cifar10/compute_fid.py:52-64),
    (ii) {"step","ema","network"} with `ema` keys prefixed "ema_model." (AD/image_diffusion/unet.py:107-123), (iii) bare state-dict;
    all read with weights_only=True."""
    import compute_fid
    from image_diffusion.unet import create_model
    from torchcfm_compat import UNetModelWrapper

    kw = dict(dim=(3, 32, 32), num_channels=32, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions="16")
    donor = UNetModelWrapper(**kw)
    with torch.no_grad():
        for p in donor.parameters():
            p.uniform_(-0.3, 0.3)
    sd = donor.state_dict()
    # (i) torchcfm layout, plain and DataParallel-prefixed
    for prefix in ("", "module."):
        path = tmp_path / f"ckpt_i_{len(prefix)}.pt"
        ema_sd = {prefix + k: v for k, v in sd.items()}
        torch.save({"net_model": ema_sd, "ema_model": ema_sd, "sched": {}, "optim": {}, "step": 7}, path)
        net = compute_fid.load_checkpoint(UNetModelWrapper(**kw), str(path))
        assert not net.training
        for k, v in net.state_dict().items():
            assert torch.equal(v, sd[k]), k
    # (ii) image_diffusion layout through create_model(model_path=...)
    ckw = dict(image_size=32, in_channels=3, out_channels=3, num_channels=32, num_res_blocks=1, channel_mult="1,2", attention_resolutions="16")
    path = tmp_path / "ckpt_ii.pt"
    torch.save({"step": 3, "ema": {**{"ema_model." + k: v for k, v in sd.items()}, "initted": torch.tensor(True), "step": torch.tensor(3)},
                "network": sd}, path)
    m = create_model(model_path=str(path), **ckw)
    assert "successfully" in capsys.readouterr().out
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # (iii) bare state-dict
    path = tmp_path / "ckpt_iii.pt"
    torch.save(sd, path)
    m = create_model(model_path=str(path), **ckw)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # partial-match fallback (unet.py:22-40,113-123): a checkpoint of a different width -> shape-matched tensors only, rest re-initialised
    other = UNetModelWrapper(dim=(3, 32, 32), num_channels=64, num_res_blocks=1, channel_mult=(1, 2), attention_resolutions="16").state_dict()
    path = tmp_path / "ckpt_other.pt"
    torch.save(other, path)
    m = create_model(model_path=str(path), **ckw)
    out = capsys.readouterr().out
    assert "Could not load" in out and "matching weights" in out
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    # create_model's size/string mapping (unet.py:63-84)
    m128 = create_model(image_size=128, in_channels=6, out_channels=3, num_channels=32, num_res_blocks=1, attention_resolutions="32,16,8")
    assert m128.channel_mult == (1, 1, 2, 3, 4) and m128.attention_resolutions == (4, 8, 16)


_FID_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
import torch
from mi355 import dist as mdist
import compute_fid
rank, world, local = mdist.init_from_env("gloo")
full = []
for call in range(2):
    mine = compute_fid.draw_x0_shard(10, 5, call, torch.device("cpu"))
    lo, hi = mdist.shard_range(10)
    g = torch.Generator(); g.manual_seed(5 + call)
    ref = torch.randn(10, 3, 32, 32, generator=g)
    assert torch.equal(mine, ref[lo:hi]), (rank, call)
    assert torch.equal(mdist.all_gather_batch(mine, 10), ref)
mdist.barrier()
print("rank", rank, "ok")
'''


def test_compute_fid_x0_is_the_one_rank_batch_resharded(tmp_path):
    """cifar10/compute_fid.py:75 draws ONE batch; with N ranks each takes its slice of the same seeded draw."""
    script = tmp_path / "w.py"
    script.write_text(_FID_WORKER.format(pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_local_writer_files(tmp_path):
    """AD/image_diffusion/writers.py:291-369 (LocalWriter): metrics.csv (pandas frame, flushed every n and on close, the index
    column restarting at every flush like the reference's read-concat-write), config.yaml, images/<key>_<step>.png."""
    import pandas as pd
    import yaml
    from PIL import Image

    from image_diffusion.writers import LocalWriter

    w = LocalWriter(str(tmp_path / "run"), flush_every_n=2)
    w.log_hparams({"lr": 1e-3, "net": {"num_channels": 32}})
    assert yaml.safe_load(open(tmp_path / "run" / "config.yaml")) == {"lr": 1e-3, "net": {"num_channels": 32}}
    w.write_scalars(0, {"loss": 1.5})
    assert not (tmp_path / "run" / "metrics.csv").exists()            # not flushed yet
    w.write_scalars(1, {"loss": 1.25})
    df = pd.read_csv(tmp_path / "run" / "metrics.csv", index_col=0)
    assert list(df.columns) == ["step", "loss"] and df["loss"].tolist() == [1.5, 1.25] and df.index.tolist() == [0, 1]
    w.write_scalars(2, {"loss": 1.0, "test_conditional_mse": 0.3})   # mnist/train_mnist.py:283-288
    w.close()
    df = pd.read_csv(tmp_path / "run" / "metrics.csv", index_col=0)
    assert df["step"].tolist() == [0, 1, 2] and df.index.tolist() == [0, 1, 0]
    assert np.isnan(df["test_conditional_mse"].iloc[0]) and df["test_conditional_mse"].iloc[2] == 0.3
    # flush cadence of the reference (writers.py:309-313, 354-365): (count + 1) % n is tested before the call is counted and flush()
    # resets the count, so with n = 3 the file is rewritten after calls 3, 5, 7, ... and the index column restarts there
    w3 = LocalWriter(str(tmp_path / "run3"), flush_every_n=3)
    sizes = []
    for k in range(7):
        w3.write_scalars(k, {"v": float(k)})
        p3 = tmp_path / "run3" / "metrics.csv"
        sizes.append(len(pd.read_csv(p3, index_col=0)) if p3.exists() else 0)
    assert sizes == [0, 0, 3, 3, 5, 5, 7]
    assert pd.read_csv(tmp_path / "run3" / "metrics.csv", index_col=0).index.tolist() == [0, 1, 2, 0, 1, 0, 1]
    w.write_images(7, {"samples": torch.rand(6, 1, 8, 8), "one": (torch.rand(3, 5, 4) * 255).to(torch.uint8)})
    g = Image.open(tmp_path / "run" / "images" / "samples_7.png")
    assert g.size == (6 * 10 + 2, 10 + 2)                              # make_grid: nrow 8, padding 2
    assert Image.open(tmp_path / "run" / "images" / "one_7.png").size == (4, 5)

    class Fig:
        def savefig(self, path, bbox_inches=None):
            assert bbox_inches == "tight"
            open(path, "wb").write(b"png")

    w.write_figures(9, {"condition": Fig()})
    assert (tmp_path / "run" / "images" / "condition_9.png").exists()


def test_frechet_proxy_properties():
    """SURVEY 8(d)(ii) stand-in for FID: zero on identical sets, the closed form on Gaussians, grows with a distribution shift."""
    import evaluation

    g = torch.Generator().manual_seed(0)
    a = torch.randn(4000, 5, generator=g, dtype=torch.float64) * torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0]) + 0.3
    b = torch.randn(4000, 5, generator=g, dtype=torch.float64) * torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0]) + 0.3
    assert abs(evaluation.frechet_distance(a, a)) < 1e-9
    assert evaluation.frechet_distance(a, b) < 0.02                      # same distribution: sampling noise only
    shifted = b + torch.tensor([1.0, 0, 0, 0, 0])
    assert abs(evaluation.frechet_distance(a, shifted) - 1.0) < 0.1        # mean shift 1 in one axis -> |dmu|^2 = 1
    scaled = b * 2.0                                                       # N(0.6, 4 S): Tr(S + 4S - 4S) = Tr(S) = 8.5, |dmu|^2 = 0.45
    assert abs(evaluation.frechet_distance(a, scaled) - (8.5 + 0.45)) < 0.5
    imgs = (torch.rand(600, 3, 32, 32, generator=g) * 255).to(torch.uint8)
    f = evaluation.random_conv_features(imgs, seed=0)
    assert f.shape == (600, 256) and torch.equal(f, evaluation.random_conv_features(imgs, seed=0))
    darker = (imgs.float() * 0.8).to(torch.uint8)
    assert evaluation.frechet_proxy(imgs[:300], imgs[300:]) < evaluation.frechet_proxy(imgs[:300], darker[300:])


_DOPRI_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
import torch
from mi355 import dist as mdist
from mi355.ode import Dopri5


class CpuOps:
    """torch stand-in for the three HIP stage kernels (host logic under test: the controller and its ONE exchange)."""
    def rk_combine(self, out, y0, ks, coeffs):
        acc = torch.zeros_like(ks[0])
        for k, c in zip(ks, coeffs):
            acc = acc + k * float(c)
        out.copy_(acc if y0 is None else y0 + acc)
        return out
    def rk_sqnorm(self, acc, a, sub=None, b=None, b2=None, atol=1.0, rtol=0.0):
        num = a if sub is None else a - sub
        den = torch.full_like(a, atol)
        if b is not None:
            m = b.abs() if b2 is None else torch.max(b.abs(), b2.abs())
            den = atol + rtol * m
        acc += (num / den).double().pow(2).sum()
        return acc
    def rk_interp(self, out, y0, y1, ym, f0, f1, dt, x):
        a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * ym
        b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * ym
        c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * ym
        out.copy_((((a * x + b) * x + c) * x + dt * f0) * x + y0)
        return out


rank, world, local = mdist.init_from_env("gloo")
g = torch.Generator().manual_seed(3)
full = torch.randn(6, 5, generator=g)
A = torch.randn(5, 5, generator=g) * 0.7
# rows decay at very different rates, so a per-shard error norm would pick different steps on the two ranks
scale = torch.tensor([0.1, 0.3, 1.0, 3.0, 9.0, 27.0]).reshape(6, 1)
f = lambda t, y, s: [torch.tanh(y[0] @ A) * s - 0.5 * y[0] * s]
lo, hi = mdist.shard_range(6)
single = Dopri5(lambda t, y: f(t, y, scale), 1e-5, 1e-5, ops=CpuOps(), sync_norm=False)
want = single.integrate([full], 0.0, 1.0)[0]
sharded = Dopri5(lambda t, y: f(t, y, scale[lo:hi]), 1e-5, 1e-5, ops=CpuOps(), sync_norm=True)
got = sharded.integrate([full[lo:hi].clone()], 0.0, 1.0)[0]
assert sharded.n_steps == single.n_steps and sharded.nfe == single.nfe, (rank, sharded.n_steps, single.n_steps)
assert torch.allclose(got, want[lo:hi], rtol=1e-6, atol=1e-7), (rank, (got - want[lo:hi]).abs().max())
mdist.barrier()
print("rank", rank, "ok", sharded.n_steps)
'''


def test_dopri5_error_norm_allreduce_world2_gloo(tmp_path):
    """The adaptive path's one real exchange (mi355/ode.py Dopri5._norm): with the batch sharded over two ranks the error norm is
    all-reduced, so both ranks take the SAME accept / reject decisions and step sizes as the single-rank solve of the full batch
    (torchdiffeq's norm runs over the whole batch: cifar10/compute_fid.py:80-85)."""
    script = tmp_path / "w.py"
    script.write_text(_DOPRI_WORKER.format(pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_ema_invalidates_engine_behind_dataparallel():
    """cifar10/train_cifar10.py:112-113,154-159: under --parallel the EMA model is an nn.DataParallel whose U-Net sits at `.module`;
    ema() must drop THAT module's packed-weight engine, or generate_samples(ema_model, True, ...) samples stale weights."""
    import utils_cifar
    import utils_mnist
    from torchcfm_compat import UNetModelWrapper

    def mk():
        return UNetModelWrapper(dim=(1, 16, 16), num_res_blocks=1, num_channels=32, channel_mult=(1, 2), num_heads=1, num_head_channels=-1,
                                attention_resolutions="8", dropout=0.0)

    for mod in (utils_cifar, utils_mnist):
        src, tgt = mk(), mk()
        wrapped = torch.nn.DataParallel(tgt)
        calls = []
        tgt.invalidate_engine = lambda calls=calls: calls.append(1)
        mod.ema(torch.nn.DataParallel(src), wrapped, 0.5)   # both are wrapped in the reference: the "module." keys line up
        assert calls == [1], mod.__name__
        mod.ema(src, tgt, 0.5)           # the bare module still works, once per call
        assert calls == [1, 1]
