// Vector-Jacobian product of the U-Net w.r.t. its image input: the plan of unet_engine.hip walked in reverse.
//
// Reference: the reconstruction-guidance sampler differentiates the x0 model through the network,
// `vmap(grad(constraint, argnums=0))(xi, i, condition)` (AD/image_diffusion/sampling.py:154-163); every sample's loss depends on
// its own image only (GroupNorm and attention are per sample), so vmap(grad) over the batch IS one batched backward pass.
// Only data gradients exist here (no weight gradients: inference), so each forward op has a short adjoint:
//   conv (+ residual, + emb)     residual: grad[res] += G (or its 2x2 block sums for the nearest-x2 residual of ResBlock(up));
//                                data: the same implicit-GEMM kernels on transposed / tap-flipped weights (conv_pack_weights_dgrad);
//                                stride 2: zero-stuff G first; nearest-x2 input: 2x2 block sums of the result;
//                                two sources (skip concat): one conv, the halves of its output go to the two gradients
//   GroupNorm32 (+SiLU, +FiLM)   gn_silu_bwd (backward.hip) from the site's kept (a, b, mean, rstd)
//   attention core               attention_bwd (attention_bwd.hip) from the kept qkv tensor
//   avg-pool / nearest-up        gathers (backward.hip)
// Every activation of the forward is still in the workspace (the plan never reuses an arena slot); gradients live in a second arena
// of the same layout and are written by their first contributor, added to by the others.
#include "unet_engine.h"

int unet_backward(const mi355_unet* net, const float* grad_out, float* grad_x, int Cx, int B, void* workspace, int64_t workspace_bytes,
                  hipStream_t stream) {
  MI355_REQUIRE(net && grad_out && grad_x && workspace, -1, "unet_vjp: null argument");
  MI355_REQUIRE(net->cfg.differentiable, -4, "unet_vjp: the handle was not created with cfg.differentiable = 1");
  MI355_REQUIRE(B > 0 && Cx > 0 && Cx <= net->cfg.in_channels, -2, "unet_vjp: bad batch / channel count");
  const WsLayout l = unet_ws_layout(net, B);
  MI355_REQUIRE((int64_t)l.total <= workspace_bytes, -2, "unet_vjp: workspace too small");
  const int dtype = net->cfg.dtype, esz = dtype == 0 ? 4 : 2, CH = dtype == 0 ? 16 : 32;
  char* ws = reinterpret_cast<char*>(workspace);
  const char* W = net->dev_weights;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  auto TP = [&](int id) -> void* { return ws + l.arena + net->tensors[id].offset_per_image * (size_t)B * esz; };
  auto GP = [&](int id) -> void* { return ws + l.grads + net->tensors[id].offset_per_image * (size_t)B * esz; };
  void* du = ws + l.du; void* tmp = ws + l.tmp; void* zbuf = ws + l.z; void* dyp = ws + l.dy;
  std::vector<char> written(net->tensors.size(), 0);
  int rc;
  auto site = [&](int k, const float*& a, const float*& b, const float*& mean, const float*& rstd) {
    const float* sp = F(l.sites) + net->site_off[k] * (size_t)B;
    const size_t Cs = (size_t)net->site_C[k];
    a = sp; b = sp + (size_t)B * Cs; mean = sp + (size_t)2 * B * Cs; rstd = mean + (size_t)B * 32;
  };
  // grad[id] (+)= gather(src)
  auto accumulate = [&](int id, const void* src, int Hs, int Ws, int cs, int coff, int mode, float scale) -> int {
    const PlanTensor& t = net->tensors[id];
    const int r = grad_gather_launch(dtype, GP(id), src, B, t.H, t.W, t.C, Hs, Ws, cs, coff, mode, written[id], scale, stream);
    written[id] = 1;
    return r;
  };
  const int S = net->cfg.image_size;
  for (int oi = (int)net->ops.size() - 1; oi >= 0; --oi) {
    const PlanOp& op = net->ops[oi];
    const PlanTensor& s0 = net->tensors[op.src0];
    const int C1 = op.src1 >= 0 ? net->tensors[op.src1].C : 0;
    if (op.kind == OP_GN) continue;   // folded into its consumer's adjoint
    if (op.kind == OP_CONV) {
      // ---- G = gradient of the conv's output ----
      const void* G; int Hg, Wg, Cg;
      if (op.dst < 0) {   // network output (NCHW fp32): pack the caller's cotangent, channel-padded to one chunk
        if ((rc = pack_nhwc_launch(dtype, grad_out, op.Cout, nullptr, 0, B, S * S, CH, dyp, stream))) return rc;
        G = dyp; Hg = S; Wg = S; Cg = CH;
      } else {
        MI355_REQUIRE(written[op.dst], -4, "unet_vjp: a conv output has no gradient (plan order)");
        const PlanTensor& d = net->tensors[op.dst];
        G = GP(op.dst); Hg = d.H; Wg = d.W; Cg = d.C;
      }
      // ---- residual operand ----
      if (op.res >= 0) {
        if (op.res_mode == RES_SAME) rc = accumulate(op.res, G, Hg, Wg, Cg, 0, GATHER_SAME, 1.0f);
        else rc = accumulate(op.res, G, Hg, Wg, Cg, 0, GATHER_POOL, 1.0f);   // forward added nearest-x2(res): sum the 2x2 blocks
        if (rc) return rc;
      }
      // ---- data gradient through the conv ----
      ConvDesc c; c.dtype = dtype; c.N = B; c.ks = op.ks; c.mode = CONV_UNIT;
      c.w = W + op.wT_off; c.Cout = op.cin_pad; c.out = du; c.out_mode = OUT_NHWC;
      int Hd, Wd;   // resolution of the data-gradient conv's output
      if (op.mode == CONV_STRIDE2) {   // zero insertion back to the input resolution, then a stride-1 conv with the flipped taps
        if ((rc = grad_gather_launch(dtype, zbuf, G, B, s0.H, s0.W, Cg, Hg, Wg, Cg, 0, GATHER_STUFF, 0, 1.0f, stream))) return rc;
        c.src0 = zbuf; c.C0 = Cg; c.Hs = s0.H; c.Ws = s0.W; Hd = s0.H; Wd = s0.W;
      } else {
        c.src0 = G; c.C0 = Cg; c.Hs = Hg; c.Ws = Wg; Hd = Hg; Wd = Wg;
      }
      if ((rc = conv_launch(c, stream))) return rc;
      // nearest-x2 input (Upsample / ResBlock(up)): the conv saw up2(u), so du(u) = 2x2 block sums of the conv's data gradient
      const void* dU = du; int cs = op.cin_pad;
      const bool up = op.mode == CONV_UP2;
      if (op.use_pro) {
        if (up) {
          if ((rc = grad_gather_launch(dtype, tmp, du, B, s0.H, s0.W, op.cin_pad, Hd, Wd, op.cin_pad, 0, GATHER_POOL, 0, 1.0f, stream))) return rc;
          dU = tmp;
        }
        GnBwdDesc g; g.dtype = dtype; g.x0 = TP(op.src0); g.C0 = s0.C; g.x1 = op.src1 >= 0 ? TP(op.src1) : nullptr; g.C1 = C1;
        g.du = dU; g.du_stride = cs; g.N = B; g.HW = s0.H * s0.W; g.silu = op.pro_silu;
        site(op.gn_site, g.a, g.b, g.mean, g.rstd);
        g.g0 = GP(op.src0); g.acc0 = written[op.src0];
        if (op.src1 >= 0) { g.g1 = GP(op.src1); g.acc1 = written[op.src1]; }
        if ((rc = gn_silu_bwd_launch(g, stream))) return rc;
        written[op.src0] = 1;
        if (op.src1 >= 0) written[op.src1] = 1;
      } else {
        const int mode = up ? GATHER_POOL : GATHER_SAME;
        if ((rc = accumulate(op.src0, dU, Hd, Wd, cs, 0, mode, 1.0f))) return rc;
        if (op.src1 >= 0 && (rc = accumulate(op.src1, dU, Hd, Wd, cs, s0.C, mode, 1.0f))) return rc;
      }
    } else if (op.kind == OP_ATTN) {
      MI355_REQUIRE(written[op.dst], -4, "unet_vjp: an attention output has no gradient (plan order)");
      AttnBwdDesc a; a.dtype = dtype; a.qkv = TP(op.src0); a.a = TP(op.dst); a.da = GP(op.dst); a.dqkv = GP(op.src0);
      a.N = B; a.T = s0.H * s0.W; a.heads = op.heads; a.ch = op.ch; a.new_order = net->cfg.use_new_attention_order;
      a.L = F(l.ld); a.D = a.L + (size_t)B * op.heads * a.T;
      if ((rc = attention_bwd_launch(a, stream))) return rc;
      written[op.src0] = 1;   // q, k and v parts are all written: the qkv tensor has this one consumer
    } else if (op.kind == OP_POOLAFF) {
      // forward: out = avgpool2(silu(a x + b)): each input position got 1/4 of its block's gradient
      MI355_REQUIRE(written[op.dst], -4, "unet_vjp: a pooled tensor has no gradient (plan order)");
      const PlanTensor& d = net->tensors[op.dst];
      if ((rc = grad_gather_launch(dtype, tmp, GP(op.dst), B, s0.H, s0.W, s0.C, d.H, d.W, d.C, 0, GATHER_UP, 0, 0.25f, stream))) return rc;
      GnBwdDesc g; g.dtype = dtype; g.x0 = TP(op.src0); g.C0 = s0.C; g.du = tmp; g.du_stride = s0.C; g.N = B; g.HW = s0.H * s0.W;
      g.silu = op.pro_silu;
      site(op.gn_site, g.a, g.b, g.mean, g.rstd);
      g.g0 = GP(op.src0); g.acc0 = written[op.src0];
      if ((rc = gn_silu_bwd_launch(g, stream))) return rc;
      written[op.src0] = 1;
    } else if (op.kind == OP_RESAMPLE) {
      MI355_REQUIRE(written[op.dst], -4, "unet_vjp: a resampled tensor has no gradient (plan order)");
      const PlanTensor& d = net->tensors[op.dst];
      if (op.mode == CONV_POOL2) rc = accumulate(op.src0, GP(op.dst), d.H, d.W, d.C, 0, GATHER_UP, 0.25f);
      else rc = accumulate(op.src0, GP(op.dst), d.H, d.W, d.C, 0, GATHER_POOL, 1.0f);
      if (rc) return rc;
    } else {
      mi355_set_error("unet_vjp: op kind without an adjoint in a differentiable plan");
      return -4;
    }
  }
  MI355_REQUIRE(written[net->in_tensor], -4, "unet_vjp: the input received no gradient");
  return unpack_channels_launch(dtype, GP(net->in_tensor), B, S * S, net->in_pad, 0, Cx, grad_x, stream);
}
