"""fp32 PyTorch-CPU restatement of the reference U-Net forward (TEST ORACLE).

Functional: walks a plain state-dict (reference key layout) with a config; no
nn.Module objects.  Reference: `amortised diffusion/image_diffusion/unet.py`
(abbreviated AD/…) and `AD/image_diffusion/nn.py`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """Mirror of the UNetModel ctor arguments that matter at inference (unet.py:521-541)."""

    image_size: int
    in_channels: int
    model_channels: int
    out_channels: int
    num_res_blocks: int
    attention_resolutions: Tuple[int, ...]  # downsample rates ds at which attention is used
    channel_mult: Tuple[int, ...] = (1, 2, 4, 8)
    conv_resample: bool = True
    num_heads: int = 1
    num_head_channels: int = -1
    num_heads_upsample: int = -1
    use_scale_shift_norm: bool = False
    resblock_updown: bool = False
    use_new_attention_order: bool = False

    def heads_for(self, ch: int, upsample: bool = False) -> int:
        # unet.py:370-378
        if self.num_head_channels == -1:
            nh = self.num_heads_upsample if (upsample and self.num_heads_upsample != -1) else self.num_heads
            return nh
        assert ch % self.num_head_channels == 0
        return ch // self.num_head_channels


def config_from_create_model(*, image_size, in_channels, out_channels, num_channels, num_res_blocks,
                             channel_mult="", attention_resolutions="16", num_heads=1, num_head_channels=-1,
                             num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                             use_new_attention_order=False, **_unused) -> UNetConfig:
    """`create_model` argument mapping (unet.py:43-105)."""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        cm = table[image_size]
    else:
        cm = tuple(int(c) for c in channel_mult.split(","))
    if isinstance(attention_resolutions, int):
        ads = (image_size // attention_resolutions,)
    elif isinstance(attention_resolutions, str):
        ads = tuple(image_size // int(r) for r in attention_resolutions.split(","))
    else:
        raise NotImplementedError
    return UNetConfig(image_size=image_size, in_channels=in_channels, model_channels=num_channels,
                      out_channels=out_channels, num_res_blocks=num_res_blocks, attention_resolutions=ads,
                      channel_mult=cm, num_heads=num_heads, num_head_channels=num_head_channels,
                      num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                      resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order)


# ----------------------------------------------------------------------------------------------
# structure (unet.py:571-706): a list of blocks, each a list of layer descriptors
# ----------------------------------------------------------------------------------------------

def build_plan(cfg: UNetConfig):
    mc = cfg.model_channels
    ch = input_ch = int(cfg.channel_mult[0] * mc)
    input_blocks: List[List[tuple]] = [[("conv", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            out = int(mult * mc)
            layers = [("res", ch, out, False, False)]
            ch = out
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch, cfg.heads_for(ch)))
            input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            if cfg.resblock_updown:
                input_blocks.append([("res", ch, ch, False, True)])
            else:
                input_blocks.append([("down", ch, cfg.conv_resample)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch, False, False), ("attn", ch, cfg.heads_for(ch)), ("res", ch, ch, False, False)]
    output_blocks: List[List[tuple]] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            out = int(mc * mult)
            layers = [("res", ch + ich, out, False, False)]
            ch = out
            if ds in cfg.attention_resolutions:
                layers.append(("attn", ch, cfg.heads_for(ch, upsample=True)))
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    layers.append(("res", ch, ch, True, False))
                else:
                    layers.append(("up", ch, cfg.conv_resample))
                ds //= 2
            output_blocks.append(layers)
    return input_blocks, middle, output_blocks, input_ch


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------

def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """nn.py:97-115: [cos(t f_k), sin(t f_k)], f_k = exp(-ln(max_period) k / half), zero-pad if dim is odd."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """nn.py:11-13,87-94: GroupNorm(32, C) in fp32, eps = torch default 1e-5."""
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def res_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, emb: torch.Tensor, cin: int, cout: int,
              up: bool, down: bool, film: bool) -> torch.Tensor:
    """ResBlock._forward (unet.py:331-351)."""
    h = F.silu(group_norm32(x, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"]))
    if up:  # Upsample(use_conv=False): nearest x2 on both h and x (unet.py:289-291, 332-337)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:  # Downsample(use_conv=False): AvgPool2d(2) (unet.py:292-294, 236)
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(sd, p + ".in_layers.2", h)
    emb_out = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[..., None, None]
    if film:  # unet.py:343-347
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = group_norm32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"]) * (1 + scale) + shift
        h = F.silu(h)
    else:  # unet.py:349-350
        h = h + emb_out
        h = F.silu(group_norm32(h, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"]))
    h = _conv(sd, p + ".out_layers.3", h)  # Dropout is identity in eval mode
    if cout == cin:
        skip = x
    else:
        w = sd[p + ".skip_connection.weight"]
        skip = F.conv2d(x, w, sd[p + ".skip_connection.bias"], padding=1 if w.shape[-1] == 3 else 0)
    return skip + h


def qkv_attention(qkv: torch.Tensor, n_heads: int, new_order: bool) -> torch.Tensor:
    """QKVAttentionLegacy.forward (unet.py:433-448) / QKVAttention.forward (unet.py:464-483)."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = q.reshape(bs * n_heads, ch, length)
        k = k.reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def attention_block(sd, p: str, x: torch.Tensor, n_heads: int, new_order: bool) -> torch.Tensor:
    """AttentionBlock._forward (unet.py:395-401)."""
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(xf, sd[p + ".norm.weight"], sd[p + ".norm.bias"]), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    h = qkv_attention(qkv, n_heads, new_order)
    h = F.conv1d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


def _run_layers(sd, cfg: UNetConfig, prefix: str, layers: Sequence[tuple], h: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    for j, layer in enumerate(layers):
        p = f"{prefix}.{j}"
        kind = layer[0]
        if kind == "conv":
            h = _conv(sd, p, h)
        elif kind == "res":
            _, cin, cout, up, down = layer
            h = res_block(sd, p, h, emb, cin, cout, up, down, cfg.use_scale_shift_norm)
        elif kind == "attn":
            h = attention_block(sd, p, h, layer[2], cfg.use_new_attention_order)
        elif kind == "down":  # Downsample (unet.py:215-240)
            h = _conv(sd, p + ".op", h, stride=2) if layer[2] else F.avg_pool2d(h, 2, 2)
        elif kind == "up":  # Upsample (unet.py:185-212)
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if layer[2]:
                h = _conv(sd, p + ".conv", h)
        else:
            raise ValueError(kind)
    return h


def unet_forward_diff(sd: Dict[str, torch.Tensor], cfg: UNetConfig, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    """UNetModel.forward (unet.py:708-728), fp32, transparent to torch.autograd (the reconstruction-guidance oracle
    differentiates through it; `unet_forward` below is the same function under no_grad)."""
    input_blocks, middle, output_blocks, _ = build_plan(cfg)
    temb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(temb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    hs = []
    h = x.float()
    for i, layers in enumerate(input_blocks):
        h = _run_layers(sd, cfg, f"input_blocks.{i}", layers, h, emb)
        hs.append(h)
    h = _run_layers(sd, cfg, "middle_block", middle, h, emb)
    for i, layers in enumerate(output_blocks):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, cfg, f"output_blocks.{i}", layers, h, emb)
    h = F.silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
    return _conv(sd, "out.2", h)


@torch.no_grad()
def unet_forward(sd: Dict[str, torch.Tensor], cfg: UNetConfig, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
    """UNetModel.forward (unet.py:708-728), fp32, inference (no autograd graph)."""
    return unet_forward_diff(sd, cfg, x, timesteps)


def model_fn(sd, cfg: UNetConfig):
    """torchcfm call convention model(t, x) with scalar or [B] t (cifar10/train_cifar10.py:148)."""

    def f(t, x, *args, **kwargs):
        t = torch.as_tensor(t, dtype=torch.float32)
        if t.dim() == 0:
            t = t.repeat(x.shape[0])
        return unet_forward(sd, cfg, x, t)

    return f
