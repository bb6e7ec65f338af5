"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mi355_sampler.h declares."""
import os
import re

import pytest

from tests.conftest import REPO


def test_build_and_symbols():
    import __graft_entry__ as ge

    ge.build()
    from mi355 import _lib

    L = _lib.lib()
    header = open(os.path.join(REPO, "include", "mi355_sampler.h")).read()
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", header))
    declared -= {"mi355_unet_config", "mi355_unet_stats", "mi355_ddpm_tables", "mi355_ddpm_options"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert set(_lib.SIGNATURES) == declared, (set(_lib.SIGNATURES) ^ declared)
    assert L.mi355_version() >= 100


def test_plan_builder_on_cpu():
    """Host-only entry points (no GPU): parameter inventory / weight size for every golden config, and argument checks."""
    import ctypes as C

    from image_diffusion.unet import param_shapes
    from mi355 import _lib
    from mi355.engine import param_inventory
    from tests.conftest import Golden
    from tests.test_oracle_golden import UNETS, cfg_from_json

    for name in UNETS:
        cfg = cfg_from_json(Golden("unet_" + name).json("config"))
        for dt in (_lib.MI355_F32, _lib.MI355_BF16):
            c = _lib.make_config(image_size=cfg.image_size, in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                                 out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks,
                                 attention_ds=cfg.attention_resolutions, channel_mult=cfg.channel_mult, conv_resample=cfg.conv_resample,
                                 num_heads=cfg.num_heads, num_head_channels=cfg.num_head_channels,
                                 use_scale_shift_norm=cfg.use_scale_shift_norm, resblock_updown=cfg.resblock_updown,
                                 use_new_attention_order=cfg.use_new_attention_order, dtype=dt)
            assert param_inventory(c) == list(param_shapes(cfg).items())
            wb = _lib.lib().mi355_unet_weight_bytes(C.byref(c))
            nparams = sum(int(__import__("numpy").prod(s)) for s in param_shapes(cfg).values())
            assert wb >= nparams * (4 if dt == 0 else 2) * 0.9
    bad = _lib.make_config(image_size=32, in_channels=3, model_channels=48, out_channels=3, num_res_blocks=1, attention_ds=(2,),
                           channel_mult=(1, 2))
    assert _lib.lib().mi355_unet_param_count(C.byref(bad)) < 0
    assert b"multiple of 32" in _lib.lib().mi355_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mi355 import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MI355BackendError):
        _lib.lib()
