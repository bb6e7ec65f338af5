"""Counterpart of `AD/image_diffusion/unet.py` (UNetModel / create_model) for inference on MI355X.

`UNetModel` keeps the reference's constructor signature and its `state_dict()` key layout
(`time_embed.{0,2}`, `input_blocks.i.j.(in_layers|emb_layers|out_layers|skip_connection|norm|qkv|
proj_out|op|conv)`, `middle_block.{0,1,2}`, `output_blocks.i.j`, `out.{0,2}`; unet.py:564-706) so
reference / torchcfm checkpoints load unchanged, but it holds NO compute modules: `forward` hands the
parameters to the HIP engine (csrc/unet_engine.hip), which runs the whole network as fused kernels.
There is no CPU path and no autograd through `forward` (training is out of scope).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from mi355._lib import MI355BackendError
from mi355.engine import UNetEngine


def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Parameter names and shapes in the reference's state_dict order (unet.py:564-706).

    `cfg` is any object with the UNetModel constructor attributes (image_size, in_channels,
    model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult, conv_resample,
    use_scale_shift_norm, resblock_updown)."""
    mc = cfg.model_channels
    E = 4 * mc
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(p, co, ci, k):
        out[p + ".weight"] = (co, ci, k, k)
        out[p + ".bias"] = (co,)

    def res(p, cin, cout):
        out[p + ".in_layers.0.weight"] = (cin,)
        out[p + ".in_layers.0.bias"] = (cin,)
        conv(p + ".in_layers.2", cout, cin, 3)
        ew = 2 * cout if cfg.use_scale_shift_norm else cout
        out[p + ".emb_layers.1.weight"] = (ew, E)
        out[p + ".emb_layers.1.bias"] = (ew,)
        out[p + ".out_layers.0.weight"] = (cout,)
        out[p + ".out_layers.0.bias"] = (cout,)
        conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cout, cin, 1)

    def attn(p, c):
        out[p + ".norm.weight"] = (c,)
        out[p + ".norm.bias"] = (c,)
        out[p + ".qkv.weight"] = (3 * c, c, 1)
        out[p + ".qkv.bias"] = (3 * c,)
        out[p + ".proj_out.weight"] = (c, c, 1)
        out[p + ".proj_out.bias"] = (c,)

    out["time_embed.0.weight"] = (E, mc)
    out["time_embed.0.bias"] = (E,)
    out["time_embed.2.weight"] = (E, E)
    out["time_embed.2.bias"] = (E,)
    mult = tuple(cfg.channel_mult)
    ch = input_ch = int(mult[0] * mc)
    conv("input_blocks.0.0", ch, cfg.in_channels, 3)
    chans = [ch]
    ds, idx = 1, 1
    conv_resample = getattr(cfg, "conv_resample", True)
    for level, m in enumerate(mult):
        for _ in range(cfg.num_res_blocks):
            res(f"input_blocks.{idx}.0", ch, int(m * mc))
            ch = int(m * mc)
            if ds in cfg.attention_resolutions:
                attn(f"input_blocks.{idx}.1", ch)
            chans.append(ch)
            idx += 1
        if level != len(mult) - 1:
            if cfg.resblock_updown:
                res(f"input_blocks.{idx}.0", ch, ch)
            elif conv_resample:
                conv(f"input_blocks.{idx}.0.op", ch, ch, 3)
            chans.append(ch)
            ds *= 2
            idx += 1
    res("middle_block.0", ch, ch)
    attn("middle_block.1", ch)
    res("middle_block.2", ch, ch)
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            res(f"output_blocks.{idx}.0", ch + ich, int(mc * m))
            ch = int(mc * m)
            j = 1
            if ds in cfg.attention_resolutions:
                attn(f"output_blocks.{idx}.{j}", ch)
                j += 1
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    res(f"output_blocks.{idx}.{j}", ch, ch)
                elif conv_resample:
                    conv(f"output_blocks.{idx}.{j}.conv", ch, ch, 3)
                ds //= 2
            idx += 1
    out["out.0.weight"] = (ch,)
    out["out.0.bias"] = (ch,)
    conv("out.2", cfg.out_channels, input_ch, 3)
    return out


class _Node(nn.Module):
    """Parameter-tree node: gives `state_dict()` the reference's dotted keys without any compute module."""


def _is_zero_init(name: str) -> bool:
    # zero_module sites: ResBlock.out_layers[-1], AttentionBlock.proj_out, UNetModel.out[-1] (unet.py:310,389,705)
    return ".out_layers.3." in name or ".proj_out." in name or name.startswith("out.2.")


def _is_norm(name: str) -> bool:
    return ".norm." in name or ".in_layers.0." in name or ".out_layers.0." in name or name.startswith("out.0.")


DEFAULT_PRECISION = os.environ.get("MI355_PRECISION", "bf16")


class UNetModel(nn.Module):
    """The full UNet with attention and timestep embedding (unet.py:490-728), MI355X inference build."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, precision: Optional[str] = None):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("the MI355X build supports dims=2 only (every reference config is 2-D)")
        if num_classes is not None:
            raise NotImplementedError("class-conditional label_emb is not used by the reference's samplers and is not built")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.dropout, self.channel_mult, self.conv_resample = dropout, tuple(channel_mult), conv_resample
        self.num_classes, self.use_checkpoint = num_classes, use_checkpoint
        self.dtype = torch.float32  # boundary dtype (the reference's use_fp16 would raise, SURVEY finding 6)
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.use_scale_shift_norm, self.resblock_updown = use_scale_shift_norm, resblock_updown
        self.use_new_attention_order = use_new_attention_order
        self.precision = precision or ("fp16" if use_fp16 else DEFAULT_PRECISION)   # use_fp16: the reference runs its torso in float16 (unet.py:559-563)
        self._shapes = param_shapes(self)
        gen_bound = lambda fan_in: 1.0 / math.sqrt(fan_in)
        for name, shape in self._shapes.items():
            p = nn.Parameter(torch.empty(shape))
            with torch.no_grad():
                if _is_norm(name):
                    p.fill_(1.0 if name.endswith("weight") else 0.0)
                elif _is_zero_init(name):
                    p.zero_()
                else:  # PyTorch's default conv / linear init: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    fan_in = int(torch.tensor(shape[1:]).prod()) if len(shape) > 1 else None
                    if fan_in is None:
                        wshape = self._shapes[name[:-4] + "weight"]
                        fan_in = int(torch.tensor(wshape[1:]).prod())
                    p.uniform_(-gen_bound(fan_in), gen_bound(fan_in))
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], p)
        self._engine: Optional[UNetEngine] = None
        self._engine_key = None
        self._dengine: Optional[UNetEngine] = None   # differentiable plan (reconstruction guidance): built on first use
        self._dengine_key = None
        self._weights_generation = 0

    # ---- engine management -------------------------------------------------------------------
    def set_precision(self, precision: str):
        """'bf16' (bf16 storage + bf16 MFMA, fp32 accumulate / GN / softmax), 'bf16x2' (the same with every conv / qkv weight as hi + lo bf16
        halves: no weight rounding, twice the MFMAs), 'fp16' (fp16 storage + fp16 MFMA: the reference's use_fp16 mode, bf16's speed with three more
        mantissa bits) or 'fp32' (exact f32 MFMA; the reference is fp32 end to end, unet.py:559,719)."""
        self.precision = precision
        self._engine = None
        self._dengine = None
        return self

    def invalidate_engine(self):
        """The parameters were rewritten through a path autograd's version counters do not see (`.data.copy_`, a raw-pointer
        kernel such as the fused EMA update): drop the packed copy so the next call re-packs the current values."""
        self._weights_generation += 1
        self._engine = None
        self._dengine = None
        return self

    def _cfg_kwargs(self):
        return dict(image_size=self.image_size, in_channels=self.in_channels, model_channels=self.model_channels,
                    out_channels=self.out_channels, num_res_blocks=self.num_res_blocks, attention_ds=self.attention_resolutions,
                    channel_mult=self.channel_mult, conv_resample=self.conv_resample, num_heads=self.num_heads,
                    num_head_channels=self.num_head_channels, num_heads_upsample=self.num_heads_upsample,
                    use_scale_shift_norm=self.use_scale_shift_norm, resblock_updown=self.resblock_updown,
                    use_new_attention_order=self.use_new_attention_order)

    def engine(self, device=None, differentiable: bool = False) -> UNetEngine:
        """The packed HIP engine for the CURRENT parameter values (re-packed when any parameter changed).
        differentiable=True: the plan that keeps what `UNetEngine.vjp` needs (its own handle and weight copy)."""
        params = list(self.parameters())
        device = torch.device(device) if device is not None else params[0].device
        if device.type != "cuda":
            raise MI355BackendError(
                f"UNetModel.forward needs the model / inputs on an MI355X device (got {device}); this build has no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (str(device), self.precision, self._weights_generation, tuple(p._version for p in params),
               tuple(p.data_ptr() for p in params))
        if differentiable:
            if self._dengine is None or key != self._dengine_key:
                self._dengine = UNetEngine(self._cfg_kwargs(), self.state_dict(), device, self.precision, differentiable=True,
                                           debug=getattr(self, "debug", None))
                self._dengine_key = key
            return self._dengine
        if self._engine is None or key != self._engine_key:
            self._engine = UNetEngine(self._cfg_kwargs(), self.state_dict(), device, self.precision, debug=getattr(self, "debug", None))
            self._engine_key = key
        return self._engine

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._engine = None
        self._dengine = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # ---- reference API ---------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, timesteps):
        """unet.py:708-728: x [N, Cin, H, W], timesteps [N] (may be fractional) -> [N, Cout, H, W]."""
        if not x.is_cuda:
            raise MI355BackendError("UNetModel.forward: x is a CPU tensor; this build only runs on the MI355X HIP backend")
        t = torch.as_tensor(timesteps, device=x.device).float()
        if t.dim() == 0:
            t = t.repeat(x.shape[0])
        eng = self.engine(x.device)
        return eng.forward(x.float().contiguous(), t.contiguous())


def create_model(*, image_size: int, in_channels: int, out_channels: int, num_channels: int, num_res_blocks,
                 channel_mult="", use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False,
                 use_new_attention_order=False, model_path=""):
    """unet.py:43-125: string/size -> channel_mult & attention_ds mapping and checkpoint loading
    ((i) bare state-dict, (ii) {"ema": {"ema_model.<key>": ...}} as written by experiments/main.py)."""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(c) for c in channel_mult.split(","))
    attention_ds = []
    if isinstance(attention_resolutions, int):
        attention_ds.append(image_size // attention_resolutions)
    elif isinstance(attention_resolutions, str):
        for res in attention_resolutions.split(","):
            attention_ds.append(image_size // int(res))
    else:
        raise NotImplementedError
    model = UNetModel(image_size=image_size, in_channels=in_channels, model_channels=num_channels, out_channels=out_channels,
                      num_res_blocks=num_res_blocks, attention_resolutions=tuple(attention_ds), dropout=dropout,
                      channel_mult=channel_mult, num_classes=None, use_checkpoint=use_checkpoint, use_fp16=use_fp16,
                      num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                      use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                      use_new_attention_order=use_new_attention_order)
    if model_path:
        try:
            state = torch.load(model_path, map_location="cpu", weights_only=True)
            if "ema" in state.keys():
                state = {k[len("ema_model."):]: v for k, v in state["ema"].items() if "ema_model" in str(k)}
            model.load_state_dict(state, strict=False)
            print(f"Loaded {model_path} successfully.")
        except Exception:
            print(f"Could not load {model_path}.")
            try:
                print("Trying to load matching weights only.")
                model = load_matching_weights(model, state)
            except Exception:
                print("Could not load matching parameters. Initializing randomly.")
    return model


def load_matching_weights(model, pretrained_state_dict):
    """unet.py:22-40: copy shape-matched tensors, re-initialise the rest."""
    sd = model.state_dict()
    for name, param in sd.items():
        if name in pretrained_state_dict and param.shape == pretrained_state_dict[name].shape:
            sd[name] = pretrained_state_dict[name]
        else:
            print("Init random", name)
            if len(param.shape) == 1:
                param.data.uniform_(-0.1, 0.1)
            else:
                nn.init.xavier_uniform_(param.data)
    model.load_state_dict(sd)
    return model
