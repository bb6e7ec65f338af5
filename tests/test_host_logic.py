"""CPU: host-side logic of the product (no GPU, no oracle in the product path): DDPM tables, likelihoods,
state-dict layout, sampler call order with a recording op double, batch sharding / gather over gloo."""
import math
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.conftest import PKG, REPO


def test_ddpm_tables_bit_exact(golden):
    from image_diffusion.sde_diffusion import DDPM, TABLE_NAMES

    g = golden("ddpm_tables")
    assert list(g.json("buffer_order")) == list(TABLE_NAMES)
    for Ns in (21, 25, 50, 100, 1000):
        d = DDPM(Ns)
        assert [n for n, _ in d.named_buffers()] == list(TABLE_NAMES)
        for n in TABLE_NAMES:
            torch.testing.assert_close(getattr(d, n), g.t(f"Ns{Ns}/{n}"), rtol=0, atol=0, equal_nan=True)
        torch.testing.assert_close(d.ts, g.t(f"Ns{Ns}/ts"), rtol=0, atol=0)
    for Ns in (19, 20):
        d = DDPM(Ns)
        for n in TABLE_NAMES:
            assert np.array_equal(torch.isfinite(getattr(d, n)).numpy(), g[f"Ns{Ns}/{n}/isfinite"])


def test_likelihoods_match_reference(golden):
    from image_diffusion.likelihoods import HyperResolution, InPainting, OutPainting, get_likelihood

    g = golden("likelihoods")
    img = g.t("img")
    for name, cls in (("inpainting", InPainting), ("outpainting", OutPainting)):
        lik = cls(patch_size=int(g[f"{name}/patch"]), pad_value=-2)
        torch.manual_seed(77)
        cond = lik.sample(img)  # same torch.randint stream as the reference run
        torch.testing.assert_close(cond, g.t(f"{name}/cond"), rtol=0, atol=0)
        torch.testing.assert_close(lik.none_like(img[:1]), g.t(f"{name}/none_like"), rtol=0, atol=0)
        from mi355.synth import randn
        torch.testing.assert_close(lik.loss(randn(5002, 3, 3, 32, 32), cond), g.t(f"{name}/loss"), rtol=1e-6, atol=1e-5)
    hr = HyperResolution(16, 16)
    torch.testing.assert_close(hr.sample(g.t("hyper/img")), g.t("hyper/cond"), rtol=1e-6, atol=1e-6)
    assert get_likelihood("InPainting") is InPainting
    with pytest.raises(NotImplementedError):
        get_likelihood("nope")


def test_state_dict_layout_matches_reference(golden):
    from image_diffusion.unet import UNetModel
    from tests.test_oracle_golden import cfg_from_json

    keys = golden("unet_keys").json("keys")
    for name, ref in keys.items():
        cfg = cfg_from_json(golden("unet_" + name).json("config"))
        net = UNetModel(image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                        out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks, attention_resolutions=cfg.attention_resolutions,
                        channel_mult=cfg.channel_mult, conv_resample=cfg.conv_resample, num_heads=cfg.num_heads,
                        num_head_channels=cfg.num_head_channels, use_scale_shift_norm=cfg.use_scale_shift_norm,
                        resblock_updown=cfg.resblock_updown, use_new_attention_order=cfg.use_new_attention_order)
        mine = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert mine == ref, name
    # zero-initialised modules as in the reference (zero_module): 109 all-zero tensors in the CIFAR config
    cfg = cfg_from_json(golden("unet_cifar").json("config"))
    net = UNetModel(32, 3, 128, 3, 2, (2,), dropout=0.1, channel_mult=(1, 2, 2, 2), num_heads=4, num_head_channels=64)
    assert sum(int((v == 0).all()) for v in net.state_dict().values()) == 109
    assert sum(p.numel() for p in net.parameters()) == 35746307


def test_cpu_forward_fails_loudly():
    from image_diffusion.unet import UNetModel
    from mi355._lib import MI355BackendError

    net = UNetModel(16, 1, 32, 1, 1, (2,), channel_mult=(1, 2), num_heads=2)
    with pytest.raises(MI355BackendError):
        net(torch.zeros(1, 1, 16, 16), torch.zeros(1))
    from mi355.ops import default_ops
    with pytest.raises(MI355BackendError):
        default_ops.clip_(torch.zeros(4))


class RecordingOps:
    """Test double for mi355.ops.Ops: records the call sequence, computes nothing."""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def f(*a, **k):
            scal = [round(float(v), 6) if isinstance(v, (int, float)) else None for v in a]
            self.calls.append((name, [v for v in scal if v is not None], [id(v) for v in a if isinstance(v, torch.Tensor)]))
            return a[0]
        return f


def test_sampler_call_order_and_noise_accounting():
    """sampling.py:50-75,80-133,209-260 loop structure, checked without a GPU through the op double."""
    from image_diffusion import sampling
    from image_diffusion.conditioning import Amortized, Replacement
    from image_diffusion.likelihoods import InPainting
    from image_diffusion.sde_diffusion import DDPM

    Ns = 25
    ddpm = DDPM(Ns)
    lik = InPainting(4, -2)
    seen_t, seen_c = [], []

    def eps_model(xi, i):
        assert i.dtype == torch.long and i.shape == (xi.shape[0],)
        seen_t.append(int(i[0]))
        seen_c.append(xi.shape[1])
        return torch.zeros(xi.shape[0], 1, *xi.shape[2:])

    xT = torch.zeros(2, 1, 8, 8)
    cond = torch.zeros(2, 1, 8, 8)
    draws = [torch.zeros(2, 1, 8, 8) for _ in range(200)]

    rec = RecordingOps()
    with sampling.use_ops(rec), sampling.injected_noise(draws):
        sampling.get_prior_sample_fn(eps_model, ddpm, Replacement(0.1, 1.0, True, 0), lik)(xT)
    names = [c[0] for c in rec.calls]
    assert names == ["ddpm_step_"] * Ns + ["clip_"]
    assert seen_t == list(reversed(range(Ns))) and set(seen_c) == {1}
    T = ddpm.host_tables()
    first = rec.calls[0][1]
    assert first[0] == pytest.approx(float(T["sqrt_recip_alphas_cumprod"][Ns - 1]), rel=1e-5)
    assert first[4] == pytest.approx(math.exp(0.5 * float(T["posterior_log_variance_clipped"][Ns - 1])), rel=1e-4)

    # amortized with 2 corrector steps: per step predictor + 2 x (net call, corrector); net sees 2C channels
    rec = RecordingOps(); seen_t.clear(); seen_c.clear()
    with sampling.use_ops(rec), sampling.injected_noise(draws):
        sampling.get_conditional_sample_fn(eps_model, ddpm, Amortized(0.9, 2, 0.1), lik)(xT, cond)
    names = [c[0] for c in rec.calls]
    assert names == (["ddpm_step_", "corrector_step_", "corrector_step_"] * Ns) + ["clip_"]
    assert len(seen_t) == 3 * Ns and set(seen_c) == {2}

    # replacement with start_fraction 0.5: mask applied only while i < int(Ns*0.5) = 12
    rec = RecordingOps(); seen_t.clear()
    with sampling.use_ops(rec), sampling.injected_noise(draws):
        sampling.get_conditional_sample_fn(eps_model, ddpm, Replacement(0.1, 0.5, True, 0), lik)(xT, cond)
    names = [c[0] for c in rec.calls]
    assert names.count("replace_mask_") == int(Ns * 0.5)
    assert names[: Ns - 12] == ["ddpm_step_"] * (Ns - 12) and names[Ns - 12: Ns - 10] == ["replace_mask_", "ddpm_step_"]

    # noise accounting: prior = Ns-1 draws; replacement(noise) = Ns-1 + int(Ns*sf)
    class Counting(RecordingOps):
        pass
    few = [torch.zeros(2, 1, 8, 8) for _ in range(Ns - 1)]
    with sampling.use_ops(RecordingOps()), sampling.injected_noise(few):
        sampling.get_prior_sample_fn(eps_model, ddpm, Replacement(0.1, 1.0, True, 0), lik)(xT)  # exactly enough
    with sampling.use_ops(RecordingOps()), sampling.injected_noise(few[:-1]):
        with pytest.raises(RuntimeError, match="exhausted"):
            sampling.get_prior_sample_fn(eps_model, ddpm, Replacement(0.1, 1.0, True, 0), lik)(xT)

    from image_diffusion.conditioning import ReconstructionGuidance, get_conditioning
    with pytest.raises(NotImplementedError):
        sampling.get_conditional_sample_fn(eps_model, ddpm, ReconstructionGuidance(10.0, 1.0, "before", 0, 0.1), lik)
    assert get_conditioning("replacement") is Replacement


def test_shard_ranges_partition():
    from mi355.dist import shard_range

    for total in (1, 7, 256, 2048, 50000):
        for w in (1, 2, 3, 8):
            rs = [shard_range(total, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
import torch
from mi355 import dist as mdist
rank, world, local = mdist.init_from_env("gloo")
assert world == 2
for total in (8, 7):
    lo, hi = mdist.shard_range(total)
    full = torch.arange(total * 6, dtype=torch.float32).reshape(total, 2, 3)
    out = mdist.all_gather_batch(full[lo:hi].clone(), total)
    assert torch.equal(out, full), (rank, total)
    u8 = (full % 251).to(torch.uint8)
    assert torch.equal(mdist.all_gather_batch(u8[lo:hi].clone(), total), u8)
mdist.barrier()
print("rank", rank, "ok")
'''


def test_all_gather_world2_gloo(tmp_path):
    """N > 1 path on CPU: two processes, gloo, equal and ragged shards, one collective each."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(pkg=PKG))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` with no launcher environment (the driver's command shape): the parent starts two ranks of itself,
    both reach the process group (gloo here) and agree on the shard ranges; rank 0's JSON line is relayed, the exit code is the worst
    child's.  --dry-run stops before the first GPU call."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["dry_run"] and rec["ok"] and rec["n_gpus"] == 2 and rec["master"] == "127.0.0.1"
    assert rec["shards"] == [[0, 0, 256], [1, 256, 512]]
    # a mismatching launcher environment is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    # a rank that dies before the rendezvous: the parent notices, ends the sibling (which would sit in the rendezvous until the process-group
    # timeout), reports the exit codes with the dead rank's output and returns its code - within seconds
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, MI355_BENCH_TEST_DIE_RANK="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 7 and time.time() - t0 < 120, (r.returncode, time.time() - t0, r.stderr[-500:])
    assert "child exit codes" in r.stderr and "dying on request" in r.stderr


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package, bench's product leg excepted, may import it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(root, f)).read()), f
    bench = open(os.path.join(REPO, "bench.py")).read()
    head, _, tail = bench.partition("def cpu_baseline")
    assert not pat.search(head)                      # only the cpu_baseline leg touches the oracle
    assert "from oracle" in tail.split("def main")[0]
    assert not pat.search(tail.split("def main")[1])


def test_png_grid_writer(tmp_path):
    from mi355.imageio import make_grid, save_image

    x = torch.rand(64, 3, 32, 32)
    g = make_grid(x, nrow=8, padding=2)
    assert g.shape == (3, 8 * 34 + 2, 8 * 34 + 2)
    save_image(x, str(tmp_path / "g.png"), nrow=8)
    from PIL import Image
    assert Image.open(tmp_path / "g.png").size == (274, 274)


def test_mnist_patch_sampler():
    import utils_mnist

    torch.manual_seed(0)
    x = torch.rand(5, 1, 28, 28)
    c = utils_mnist.sample(x)
    assert c.shape == x.shape
    for k in range(5):
        assert int((c[k] == -2).sum()) == 14 * 14
        ys, xs = torch.where(c[k, 0] == -2)
        assert ys.min() >= 5 and xs.min() >= 5 and ys.max() - ys.min() == 13 and xs.max() - xs.min() == 13
    with pytest.raises(NotImplementedError):
        utils_mnist.generate_samples(torch.nn.Identity(), False, "/tmp/", 0, solver="rk4")


def test_infiniteloop_and_ema_cpu_plumbing():
    """cifar10/utils_cifar.py:47-59: `infiniteloop` yields images only, forever; `ema` on CPU tensors is the eager expression."""
    import utils_cifar

    data = [(torch.full((2,), float(k)), torch.tensor([k])) for k in range(3)]
    it = utils_cifar.infiniteloop(data)
    got = [next(it)[0].item() for _ in range(7)]
    assert got == [0.0, 1.0, 2.0, 0.0, 1.0, 2.0, 0.0]
    src, tgt = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    want = {k: v * 0.9 + src.state_dict()[k] * (1 - 0.9) for k, v in tgt.state_dict().items()}
    utils_cifar.ema(src, tgt, 0.9)
    for k, v in tgt.state_dict().items():
        assert torch.equal(v, want[k])
