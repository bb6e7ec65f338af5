// U-Net layer plan, weight packer and executor.
//
// Rebuilds the wiring of UNetModel.__init__/forward (AD/image_diffusion/unet.py:521-728; identical to
// torchcfm's UNetModelWrapper used by cifar10/ and mnist/) as a flat list of fused device ops:
//   ResBlock  (unet.py:331-351) = GN-stats, conv3x3[GN+SiLU prologue, +emb epilogue], GN-stats(+FiLM),
//                                  (1x1 skip conv), conv3x3[GN+SiLU prologue, +skip epilogue]
//   Attention (unet.py:395-401) = GN-stats, 1x1 conv[GN prologue] -> qkv, fused attention, 1x1 conv[+x]
//   Down/Upsample, skip concat, nearest-up / avg-pool are gather modes of the consuming conv.
// Parameters arrive in the reference's state_dict order and layouts and are repacked once.
#include "unet_engine.h"

#include <algorithm>
#include <cstring>
#include <map>

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Walker {
  mi355_unet_config cfg;
  int dtype, esz, CH;
  bool dry;
  const float* const* host = nullptr;
  std::vector<ParamInfo> params;
  std::map<std::string, int> pidx;
  std::vector<char> blob;
  size_t cursor = 0;
  mi355_unet* net = nullptr;
  std::string err;
  struct EmbPart { std::string name; int off, width; };
  std::vector<EmbPart> emb_parts;
  int emb_total = 0;
  int last_site = -1;   // differentiable plans: the GroupNorm site the next prologue consumer applies

  bool has_attn(int ds) const {
    for (int i = 0; i < cfg.n_attention_ds; ++i) if (cfg.attention_ds[i] == ds) return true;
    return false;
  }
  int heads_for(int ch, bool upsample) const {  // unet.py:370-378
    if (cfg.num_head_channels == -1) {
      int nh = (upsample && cfg.num_heads_upsample != -1) ? cfg.num_heads_upsample : cfg.num_heads;
      return nh;
    }
    return ch / cfg.num_head_channels;
  }

  size_t alloc(size_t bytes) {
    size_t o = cursor;
    cursor = align_up(cursor + bytes, 256);
    if (!dry) blob.resize(cursor, 0);
    return o;
  }
  const float* P(const std::string& name, std::vector<int64_t> expect) {
    auto it = pidx.find(name);
    if (it == pidx.end()) { err = "missing parameter " + name; return nullptr; }
    if (params[it->second].shape != expect) { err = "shape mismatch for " + name; return nullptr; }
    return dry ? nullptr : host[it->second];
  }
  size_t put_f32(const std::string& name, std::vector<int64_t> shape) {
    const float* p = P(name, shape);
    int64_t n = 1; for (auto s : shape) n *= s;
    size_t o = alloc((size_t)n * 4);
    if (!dry && p) memcpy(blob.data() + o, p, (size_t)n * 4);
    return o;
  }
  size_t put_conv(const std::string& name, int Cout, int Cin, int ks, bool conv1d) {
    std::vector<int64_t> shape = conv1d ? std::vector<int64_t>{Cout, Cin, 1} : std::vector<int64_t>{Cout, Cin, ks, ks};
    const float* p = P(name, shape);
    size_t bytes = conv_packed_weight_bytes(dtype, Cout, Cin, ks, net->wsplit);
    size_t o = alloc(bytes);
    if (!dry && p) conv_pack_weights(dtype, p, Cout, Cin, ks, blob.data() + o, net->wsplit);
    net->weight_bytes += (double)Cout * Cin * ks * ks * esz;
    return o;
  }
  size_t put_linear_t(const std::string& name, int J, int K) {  // W [J][K] -> Wt [K][J]
    const float* p = P(name, {J, K});
    size_t o = alloc((size_t)J * K * 4);
    if (!dry && p) {
      float* d = reinterpret_cast<float*>(blob.data() + o);
      for (int j = 0; j < J; ++j) for (int k = 0; k < K; ++k) d[(size_t)k * J + j] = p[(size_t)j * K + k];
    }
    return o;
  }

  int tensor(int C, int H, int W) {
    PlanTensor t{C, H, W, false, net->act_elems_per_image};
    net->act_elems_per_image += align_up((size_t)C * H * W, 128);
    net->tensors.push_back(t);
    return (int)net->tensors.size() - 1;
  }
  const PlanTensor& T(int id) const { return net->tensors[id]; }

  // Returns -1 (the consumer conv applies a*x + b (+SiLU) in its staging prologue) or, for small images, the id of a tensor
  // that already holds silu?(GN(x)): at 8x8 and below the convs are latency-bound 64-pixel-tile launches whose prologue math
  // and per-image (a, b) loads sit on the critical path, while the GN kernel has the whole image in L2 anyway.
  int add_gn(int s0, int s1, const std::string& wname, const std::string& bname, int film_off, int apply_silu, bool may_apply = true) {
    const int max_hw = net->knobs.gn_apply_max_hw;
    const int C = T(s0).C + (s1 >= 0 ? T(s1).C : 0);
    PlanOp op; op.kind = OP_GN; op.src0 = s0; op.src1 = s1;
    op.gamma_off = put_f32(wname, {C}); op.beta_off = put_f32(bname, {C}); op.film_emb_off = film_off;
    if (cfg.differentiable) {
      // the backward pass re-derives everything from (x, a, b, mean, rstd) of each site: no applied copies, no shared (a, b) buffer
      op.gn_site = last_site = (int)net->site_C.size();
      net->site_C.push_back(C);
      net->site_off.push_back(net->site_floats_per_image);
      net->site_floats_per_image += (size_t)2 * C + 64;
    } else if (may_apply && T(s0).H * T(s0).W <= max_hw) { op.dst = tensor(C, T(s0).H, T(s0).W); op.pro_silu = apply_silu; }
    else {
      // larger images: take the statistics from the partial sums the producing convs leave in their epilogues when the groups are
      // whole channel quads of each source (C multiple of 128 for GroupNorm32); decided per launch (a producer that cannot
      // provide them, e.g. a resample pass, leaves the site on the statistics kernel)
      const int fuse = net->knobs.gn_fuse;
      const int C1 = s1 >= 0 ? T(s1).C : 0;
      if (fuse && (C / 32) % 4 == 0 && T(s0).C % 4 == 0 && C1 % 4 == 0) {
        op.fin_ok = 1;
        for (int s : {s0, s1}) {
          if (s < 0 || net->tensors[s].stats_cap) continue;
          PlanTensor& t = net->tensors[s];
          t.stats_cap = 4 * ((t.H * t.W + 63) / 64) + 8;   // >= slots of every kernel variant: at most one slot per 32 pixels
          t.stats_off_per_image = net->stats_floats_per_image;
          net->stats_floats_per_image += (size_t)t.stats_cap * (t.C / 4) * 2;
        }
      }
    }
    net->ops.push_back(op);
    if (C > net->max_gn_c) net->max_gn_c = C;
    return op.dst;
  }
  // returns dst tensor (or -1 for the NCHW fp32 network output)
  int add_conv(const std::string& prefix, int s0, int s1, int Cin_logical, int Cout, int ks, int mode, bool conv1d, int use_pro,
               int pro_silu, int emb_off, int res, int res_mode, int out_mode) {
    PlanOp op; op.kind = OP_CONV; op.src0 = s0; op.src1 = s1; op.mode = mode; op.ks = ks; op.Cout = Cout;
    op.w_off = put_conv(prefix + ".weight", Cout, Cin_logical, ks, conv1d);
    op.bias_off = put_f32(prefix + ".bias", {Cout});
    op.use_pro = use_pro; op.pro_silu = pro_silu; op.emb_off = emb_off; op.res = res; op.res_mode = res_mode; op.out_mode = out_mode;
    int Ho = T(s0).H, Wo = T(s0).W;
    if (mode == CONV_UP2) { Ho *= 2; Wo *= 2; }
    else if (mode == CONV_POOL2) { Ho /= 2; Wo /= 2; }
    else if (mode == CONV_STRIDE2) { Ho = (Ho - 1) / 2 + 1; Wo = (Wo - 1) / 2 + 1; }
    op.dst = out_mode == OUT_NHWC ? tensor(Cout, Ho, Wo) : -1;
    if (cfg.differentiable) {
      if (use_pro) op.gn_site = last_site;
      op.cin_pad = (int)align_up(T(s0).C + (s1 >= 0 ? T(s1).C : 0), 32);   // NHWC conv outputs come in whole 32-channel tiles
      const float* pw = P(prefix + ".weight", conv1d ? std::vector<int64_t>{Cout, Cin_logical, 1} : std::vector<int64_t>{Cout, Cin_logical, ks, ks});
      op.wT_off = alloc(conv_packed_weight_bytes_dgrad(dtype, Cout, Cin_logical, ks, op.cin_pad));
      if (!dry && pw) conv_pack_weights_dgrad(dtype, pw, Cout, Cin_logical, ks, op.cin_pad, blob.data() + op.wT_off);
      // scratch of the backward pass: the data-gradient conv's output (at the conv's own input resolution, before any pooling back),
      // the zero-stuffed output gradient of a stride-2 conv, the pooled gradient of an up-sampling conv with a prologue
      const size_t hw_dgrad = mode == CONV_UP2 ? (size_t)Ho * Wo : (size_t)T(s0).H * T(s0).W;
      net->bwd_du_elems = std::max(net->bwd_du_elems, hw_dgrad * op.cin_pad);
      if (mode == CONV_STRIDE2) net->bwd_z_elems = std::max(net->bwd_z_elems, (size_t)T(s0).H * T(s0).W * Cout);
      net->bwd_tmp_elems = std::max(net->bwd_tmp_elems, (size_t)T(s0).H * T(s0).W * op.cin_pad);
    }
    net->ops.push_back(op);
    const double in_elems = (double)(T(s0).C + (s1 >= 0 ? T(s1).C : 0)) * T(s0).H * T(s0).W;
    net->conv_flops += 2.0 * Ho * Wo * (double)Cout * Cin_logical * ks * ks;
    net->act_bytes += (in_elems + (double)Cout * Ho * Wo) * esz;
    return op.dst;
  }

  int res_block(const std::string& p, int s0, int s1, int cin, int cout, bool up, bool down) {
    const bool film = cfg.use_scale_shift_norm != 0;
    const int ew = film ? 2 * cout : cout;
    const int eoff = emb_total;
    emb_parts.push_back({p + ".emb_layers.1", eoff, ew});
    emb_total += ew;
    P(p + ".emb_layers.1.weight", {ew, 4 * cfg.model_channels});
    P(p + ".emb_layers.1.bias", {ew});
    const int y1 = add_gn(s0, s1, p + ".in_layers.0.weight", p + ".in_layers.0.bias", -1, 1, !down);
    int h1;
    int res = s0, res_mode = up ? RES_UP2 : RES_SAME;
    if (down) {
      // ResBlock(down=True): h = conv(AvgPool(SiLU(GN(x)))), x -> AvgPool(x) (unet.py:332-337): both pools are small HBM-bound
      // pre-passes (the first one fused with the GN affine + SiLU), so the conv itself stays a plain stride-1 conv.
      if (s1 >= 0) { err = "ResBlock(down) over a channel concat is not supported"; return -1; }
      PlanOp op; op.kind = OP_POOLAFF; op.src0 = s0; op.pro_silu = 1;
      op.dst = tensor(T(s0).C, T(s0).H / 2, T(s0).W / 2);
      if (cfg.differentiable) { op.gn_site = last_site; net->bwd_tmp_elems = std::max(net->bwd_tmp_elems, (size_t)T(s0).H * T(s0).W * T(s0).C); }
      net->ops.push_back(op);
      h1 = add_conv(p + ".in_layers.2", op.dst, -1, cin, cout, 3, CONV_UNIT, false, 0, 0, film ? -1 : eoff, -1, RES_NONE, OUT_NHWC);
      res = resample(s0, CONV_POOL2);
    } else {
      h1 = y1 >= 0 ? add_conv(p + ".in_layers.2", y1, -1, cin, cout, 3, up ? CONV_UP2 : CONV_UNIT, false, 0, 0, film ? -1 : eoff, -1, RES_NONE, OUT_NHWC)
                   : add_conv(p + ".in_layers.2", s0, s1, cin, cout, 3, up ? CONV_UP2 : CONV_UNIT, false, 1, 1, film ? -1 : eoff, -1, RES_NONE, OUT_NHWC);
    }
    const int y2 = add_gn(h1, -1, p + ".out_layers.0.weight", p + ".out_layers.0.bias", film ? eoff : -1, 1);
    if (cin != cout) {
      if (up || down) { err = "ResBlock(up/down) with a channel change is not supported"; return -1; }
      auto it = pidx.find(p + ".skip_connection.weight");
      const int sks = (it != pidx.end() && params[it->second].shape.size() == 4) ? (int)params[it->second].shape[3] : 1;
      res = add_conv(p + ".skip_connection", s0, s1, cin, cout, sks, CONV_UNIT, false, 0, 0, -1, -1, RES_NONE, OUT_NHWC);
      res_mode = RES_SAME;
    } else if (s1 >= 0) {
      err = "identity skip over a channel concat is not supported";
      return -1;
    }
    const int skip_idx = cin != cout ? (int)net->ops.size() - 1 : -1;
    if (y2 >= 0) {
      const int out = add_conv(p + ".out_layers.3", y2, -1, cout, cout, 3, CONV_UNIT, false, 0, 0, -1, res, res_mode, OUT_NHWC);
      // Small levels (apply-type norms: the second conv reads one already-activated tensor): the 1x1 skip_connection can ride in the
      // second conv as centre-tap K chunks of the raw block input (unet.py:312-317, 351: return skip_connection(x) + h), if the launch
      // agrees (conv_fused_skip_ok: the small-level kernel, conv_small bit 3).  Both weight images are kept.
      if (out >= 0 && skip_idx >= 0 && net->ops[skip_idx].ks == 1 && !cfg.differentiable && !net->wsplit && cout % 128 == 0) {
        const int conv2_idx = (int)net->ops.size() - 1;
        PlanOp& c2 = net->ops[conv2_idx];
        const float* w3 = P(p + ".out_layers.3.weight", {cout, cout, 3, 3});
        const float* w1 = P(p + ".skip_connection.weight", {cout, cin, 1, 1});
        const float* b3 = P(p + ".out_layers.3.bias", {cout});
        const float* b1 = P(p + ".skip_connection.bias", {cout});
        c2.wf_off = alloc(conv_packed_weight_bytes_skip(dtype, cout, cout, cin));
        c2.bf_off = alloc((size_t)cout * 4);
        if (!dry && w3 && w1 && b3 && b1) {
          conv_pack_weights_skip(dtype, w3, w1, cout, cout, cin, blob.data() + c2.wf_off);
          float* bs = reinterpret_cast<float*>(blob.data() + c2.bf_off);
          for (int i = 0; i < cout; ++i) bs[i] = b3[i] + b1[i];
        }
        c2.skip_op = skip_idx;
        net->ops[skip_idx].carrier = conv2_idx;
      }
      return out;
    }
    return add_conv(p + ".out_layers.3", h1, -1, cout, cout, 3, CONV_UNIT, false, 1, 1, -1, res, res_mode, OUT_NHWC);
  }

  int attn_block(const std::string& p, int x, int C, int heads) {
    if (heads <= 0 || C % heads != 0) { err = "attention: bad head count"; return -1; }
    const int ch = C / heads;
    {
      bool ok = false;
      for (int v : {32, 64, 96, 128, 192, 256, 384, 512}) ok = ok || ch == v;
      if (!ok) { err = "attention: head channels must be one of 32, 64, 96, 128, 192, 256, 384, 512 (got " + std::to_string(ch) + ")"; return -1; }
    }
    if (cfg.differentiable && ch > 256) {   // attention_bwd.hip instantiates head sizes up to 256: fail at build, not in the middle of a guidance loop
      err = "attention: differentiable plans support head channels up to 256 (got " + std::to_string(ch) + ")";
      return -1;
    }
    const int yn = add_gn(x, -1, p + ".norm.weight", p + ".norm.bias", -1, 0);
    const double Tn_ = (double)T(x).H * T(x).W;
    if (yn < 0 && !cfg.differentiable && !net->wsplit && attn_fused_eligible(dtype, T(x).H * T(x).W, C, heads, ch, &net->knobs)) {
      // norm-apply + qkv 1x1 + attention in one kernel (attn_fused.hip): the [T, 3C] qkv tensor never exists
      PlanOp op; op.kind = OP_ATTN_FUSED; op.src0 = x; op.heads = heads; op.ch = ch; op.Cout = 3 * C;
      op.w_off = put_conv(p + ".qkv.weight", 3 * C, C, 1, true);
      op.bias_off = put_f32(p + ".qkv.bias", {3 * C});
      op.dst = tensor(C, T(x).H, T(x).W);
      net->ops.push_back(op);
      net->conv_flops += 2.0 * Tn_ * 3.0 * C * C;
      net->attn_flops += 4.0 * Tn_ * Tn_ * C;
      net->act_bytes += (2.0 * C * Tn_) * esz;   // x in, attention output out: the qkv tensor is not algorithmic traffic any more
      return add_conv(p + ".proj_out", op.dst, -1, C, C, 1, CONV_UNIT, true, 0, 0, -1, x, RES_SAME, OUT_NHWC);
    }
    const int qkv = yn >= 0 ? add_conv(p + ".qkv", yn, -1, C, 3 * C, 1, CONV_UNIT, true, 0, 0, -1, -1, RES_NONE, OUT_NHWC)
                            : add_conv(p + ".qkv", x, -1, C, 3 * C, 1, CONV_UNIT, true, 1, 0, -1, -1, RES_NONE, OUT_NHWC);
    PlanOp op; op.kind = OP_ATTN; op.src0 = qkv; op.heads = heads; op.ch = ch;
    op.dst = tensor(C, T(x).H, T(x).W);
    net->ops.push_back(op);
    if (cfg.differentiable) net->bwd_ld_floats = std::max(net->bwd_ld_floats, (size_t)2 * heads * T(x).H * T(x).W);
    const double Tn = (double)T(x).H * T(x).W;
    net->attn_flops += 4.0 * Tn * Tn * C;
    net->act_bytes += (4.0 * C * Tn) * esz;
    return add_conv(p + ".proj_out", op.dst, -1, C, C, 1, CONV_UNIT, true, 0, 0, -1, x, RES_SAME, OUT_NHWC);
  }

  int resample(int x, int mode) {
    PlanOp op; op.kind = OP_RESAMPLE; op.src0 = x; op.mode = mode;
    const int Ho = mode == CONV_UP2 ? T(x).H * 2 : T(x).H / 2, Wo = mode == CONV_UP2 ? T(x).W * 2 : T(x).W / 2;
    op.dst = tensor(T(x).C, Ho, Wo);
    net->ops.push_back(op);
    return op.dst;
  }

  int walk() {
    const int mc = cfg.model_channels, nl = cfg.n_channel_mult;
    const int S = cfg.image_size;
    net->in_pad = (int)align_up(cfg.in_channels, CH);
    net->te_w0 = put_linear_t("time_embed.0.weight", 4 * mc, mc);
    net->te_b0 = put_f32("time_embed.0.bias", {4 * mc});
    net->te_w2 = put_linear_t("time_embed.2.weight", 4 * mc, 4 * mc);
    net->te_b2 = put_f32("time_embed.2.bias", {4 * mc});
    int ch = cfg.channel_mult[0] * mc;
    const int input_ch = ch;
    net->in_tensor = tensor(net->in_pad, S, S);
    int h = add_conv("input_blocks.0.0", net->in_tensor, -1, cfg.in_channels, ch, 3, CONV_UNIT, false, 0, 0, -1, -1, RES_NONE, OUT_NHWC);
    std::vector<int> hs{h};
    int ds = 1, idx = 1;
    for (int level = 0; level < nl && err.empty(); ++level) {
      const int mult = cfg.channel_mult[level];
      for (int r = 0; r < cfg.num_res_blocks && err.empty(); ++r, ++idx) {
        const std::string p = "input_blocks." + std::to_string(idx);
        h = res_block(p + ".0", h, -1, ch, mult * mc, false, false);
        ch = mult * mc;
        if (has_attn(ds) && err.empty()) h = attn_block(p + ".1", h, ch, heads_for(ch, false));
        hs.push_back(h);
      }
      if (level != nl - 1 && err.empty()) {
        const std::string p = "input_blocks." + std::to_string(idx) + ".0";
        if (cfg.resblock_updown) h = res_block(p, h, -1, ch, ch, false, true);
        else if (cfg.conv_resample) h = add_conv(p + ".op", h, -1, ch, ch, 3, CONV_STRIDE2, false, 0, 0, -1, -1, RES_NONE, OUT_NHWC);
        else h = resample(h, CONV_POOL2);
        hs.push_back(h);
        ds *= 2; ++idx;
      }
    }
    if (!err.empty()) return -1;
    h = res_block("middle_block.0", h, -1, ch, ch, false, false);
    if (err.empty()) h = attn_block("middle_block.1", h, ch, heads_for(ch, false));
    if (err.empty()) h = res_block("middle_block.2", h, -1, ch, ch, false, false);
    idx = 0;
    for (int level = nl - 1; level >= 0 && err.empty(); --level) {
      const int mult = cfg.channel_mult[level];
      for (int i = 0; i <= cfg.num_res_blocks && err.empty(); ++i, ++idx) {
        const std::string p = "output_blocks." + std::to_string(idx);
        const int skip = hs.back(); hs.pop_back();
        const int ich = T(skip).C;
        h = res_block(p + ".0", h, skip, ch + ich, mc * mult, false, false);
        ch = mc * mult;
        int j = 1;
        if (has_attn(ds) && err.empty()) { h = attn_block(p + "." + std::to_string(j), h, ch, heads_for(ch, true)); ++j; }
        if (level && i == cfg.num_res_blocks && err.empty()) {
          const std::string q = p + "." + std::to_string(j);
          if (cfg.resblock_updown) h = res_block(q, h, -1, ch, ch, true, false);
          else if (cfg.conv_resample) h = add_conv(q + ".conv", h, -1, ch, ch, 3, CONV_UP2, false, 0, 0, -1, -1, RES_NONE, OUT_NHWC);
          else h = resample(h, CONV_UP2);
          ds /= 2;
        }
      }
    }
    if (!err.empty()) return -1;
    add_gn(h, -1, "out.0.weight", "out.0.bias", -1, 1, false);
    add_conv("out.2", h, -1, input_ch, cfg.out_channels, 3, CONV_UNIT, false, 1, 1, -1, -1, RES_NONE, OUT_NCHW_F32);
    if (!err.empty()) return -1;
    // batched emb_layers: Wt [4mc][emb_total], bias [emb_total]
    const int K = 4 * mc;
    net->emb_total = emb_total;
    net->emb_w = alloc((size_t)K * emb_total * 4);
    net->emb_b = alloc((size_t)emb_total * 4);
    if (!dry) {
      float* W = reinterpret_cast<float*>(blob.data() + net->emb_w);
      float* Bv = reinterpret_cast<float*>(blob.data() + net->emb_b);
      for (auto& e : emb_parts) {
        const float* w = host[pidx[e.name + ".weight"]];
        const float* b = host[pidx[e.name + ".bias"]];
        for (int j = 0; j < e.width; ++j) {
          Bv[e.off + j] = b[j];
          for (int k = 0; k < K; ++k) W[(size_t)k * emb_total + e.off + j] = w[(size_t)j * K + k];
        }
      }
    }
    net->conv_flops += 2.0 * (double)K * emb_total + 2.0 * (double)mc * K + 2.0 * (double)K * K;
    net->out_channels = cfg.out_channels;
    net->launches = 5 + (int64_t)net->ops.size();
    return 0;
  }
};

int check_cfg(const mi355_unet_config& c) {
  MI355_REQUIRE(c.dtype == MI355_F32 || c.dtype == MI355_BF16 || c.dtype == MI355_BF16X2 || c.dtype == MI355_F16, -1,
                "unet: dtype must be MI355_F32, MI355_BF16, MI355_BF16X2 or MI355_F16");
  MI355_REQUIRE(!((c.dtype == MI355_BF16X2 || c.dtype == MI355_F16) && c.differentiable), -4, "unet: MI355_BF16X2 / MI355_F16 plans have no backward pass");
  MI355_REQUIRE(c.n_channel_mult >= 1 && c.n_channel_mult <= 8 && c.n_attention_ds >= 0 && c.n_attention_ds <= 8, -1, "unet: bad config arrays");
  MI355_REQUIRE(c.model_channels % 32 == 0 && c.model_channels > 0, -4, "unet: model_channels must be a multiple of 32 (GroupNorm32 + 64-byte channel chunks)");
  MI355_REQUIRE(c.in_channels > 0 && c.in_channels <= 32 && c.out_channels > 0 && c.out_channels <= 32, -4, "unet: in/out channels must be in 1..32");
  MI355_REQUIRE(c.image_size > 0 && c.num_res_blocks > 0, -1, "unet: bad sizes");
  return 0;
}

}  // namespace

int unet_enumerate_params(const mi355_unet_config& cfg, std::vector<ParamInfo>& out) {
  if (int rc = check_cfg(cfg)) return rc;
  out.clear();
  const int mc = cfg.model_channels, E = 4 * mc, nl = cfg.n_channel_mult;
  auto add = [&](const std::string& n, std::vector<int64_t> s) { out.push_back({n, s}); };
  auto conv = [&](const std::string& p, int co, int ci, int k) { add(p + ".weight", {co, ci, k, k}); add(p + ".bias", {co}); };
  auto res = [&](const std::string& p, int cin, int cout) {
    add(p + ".in_layers.0.weight", {cin}); add(p + ".in_layers.0.bias", {cin});
    conv(p + ".in_layers.2", cout, cin, 3);
    const int ew = cfg.use_scale_shift_norm ? 2 * cout : cout;
    add(p + ".emb_layers.1.weight", {ew, E}); add(p + ".emb_layers.1.bias", {ew});
    add(p + ".out_layers.0.weight", {cout}); add(p + ".out_layers.0.bias", {cout});
    conv(p + ".out_layers.3", cout, cout, 3);
    if (cin != cout) conv(p + ".skip_connection", cout, cin, 1);
  };
  auto attn = [&](const std::string& p, int C) {
    add(p + ".norm.weight", {C}); add(p + ".norm.bias", {C});
    add(p + ".qkv.weight", {3 * C, C, 1}); add(p + ".qkv.bias", {3 * C});
    add(p + ".proj_out.weight", {C, C, 1}); add(p + ".proj_out.bias", {C});
  };
  auto has_attn = [&](int ds) { for (int i = 0; i < cfg.n_attention_ds; ++i) if (cfg.attention_ds[i] == ds) return true; return false; };
  add("time_embed.0.weight", {E, mc}); add("time_embed.0.bias", {E});
  add("time_embed.2.weight", {E, E}); add("time_embed.2.bias", {E});
  int ch = cfg.channel_mult[0] * mc;
  const int input_ch = ch;
  conv("input_blocks.0.0", ch, cfg.in_channels, 3);
  std::vector<int> chans{ch};
  int ds = 1, idx = 1;
  for (int level = 0; level < nl; ++level) {
    const int mult = cfg.channel_mult[level];
    for (int r = 0; r < cfg.num_res_blocks; ++r, ++idx) {
      const std::string p = "input_blocks." + std::to_string(idx);
      res(p + ".0", ch, mult * mc);
      ch = mult * mc;
      if (has_attn(ds)) attn(p + ".1", ch);
      chans.push_back(ch);
    }
    if (level != nl - 1) {
      const std::string p = "input_blocks." + std::to_string(idx) + ".0";
      if (cfg.resblock_updown) res(p, ch, ch);
      else if (cfg.conv_resample) conv(p + ".op", ch, ch, 3);
      chans.push_back(ch);
      ds *= 2; ++idx;
    }
  }
  res("middle_block.0", ch, ch); attn("middle_block.1", ch); res("middle_block.2", ch, ch);
  idx = 0;
  for (int level = nl - 1; level >= 0; --level) {
    const int mult = cfg.channel_mult[level];
    for (int i = 0; i <= cfg.num_res_blocks; ++i, ++idx) {
      const std::string p = "output_blocks." + std::to_string(idx);
      const int ich = chans.back(); chans.pop_back();
      res(p + ".0", ch + ich, mc * mult);
      ch = mc * mult;
      int j = 1;
      if (has_attn(ds)) { attn(p + "." + std::to_string(j), ch); ++j; }
      if (level && i == cfg.num_res_blocks) {
        const std::string q = p + "." + std::to_string(j);
        if (cfg.resblock_updown) res(q, ch, ch);
        else if (cfg.conv_resample) conv(q + ".conv", ch, ch, 3);
        ds /= 2;
      }
    }
  }
  add("out.0.weight", {ch}); add("out.0.bias", {ch});
  conv("out.2", cfg.out_channels, input_ch, 3);
  return 0;
}

static int run_walker(const mi355_unet_config& cfg, const float* const* host, mi355_unet* net, Walker& w) {
  // MI355_BF16X2 = bf16 storage and MFMAs with every conv / qkv weight held as hi + lo bf16 halves: from here on the plan is a bf16 plan with wsplit set
  // and MI355_F16 -> DT_F16: net->cfg.dtype holds the INTERNAL element-type code (ops.h) from here on
  net->wsplit = cfg.dtype == MI355_BF16X2 ? 1 : 0;
  w.cfg = cfg; w.cfg.dtype = cfg.dtype == MI355_F16 ? DT_F16 : (net->wsplit ? DT_BF16 : cfg.dtype);
  w.dtype = w.cfg.dtype; w.esz = w.dtype == 0 ? 4 : 2; w.CH = w.dtype == 0 ? 16 : 32;
  w.dry = host == nullptr; w.host = host; w.net = net;
  if (int rc = unet_enumerate_params(cfg, w.params)) return rc;
  for (size_t i = 0; i < w.params.size(); ++i) w.pidx[w.params[i].name] = (int)i;
  net->cfg = w.cfg;
  net->knobs = cfg.debug ? *cfg.debug : mi355_default_debug();
  net->cfg.debug = nullptr;
  if (w.walk() != 0 || !w.err.empty()) { mi355_set_error("unet plan: " + w.err); return -4; }
  return 0;
}

int64_t unet_weight_bytes(const mi355_unet_config& cfg) {
  mi355_unet tmp; Walker w;
  if (int rc = run_walker(cfg, nullptr, &tmp, w)) return rc;
  return (int64_t)w.cursor;
}

int unet_build(const mi355_unet_config& cfg, const float* const* params_host, int n_params, void* dev_weights,
               int64_t dev_weights_bytes, hipStream_t stream, mi355_unet** out) {
  MI355_REQUIRE(params_host && dev_weights && out, -1, "unet_create: null argument");
  mi355_unet* net = new mi355_unet();
  Walker w;
  int rc = run_walker(cfg, params_host, net, w);
  if (rc == 0 && n_params != (int)w.params.size()) { mi355_set_error("unet_create: parameter count mismatch"); rc = -2; }
  if (rc == 0 && (int64_t)w.cursor > dev_weights_bytes) { mi355_set_error("unet_create: device weight buffer too small"); rc = -2; }
  if (rc == 0) {
    hipError_t e = hipMemcpyAsync(dev_weights, w.blob.data(), w.cursor, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // blob is a temporary: creation is a one-off, not a hot path
    if (e != hipSuccess) { mi355_set_error(std::string("unet_create: weight upload: ") + hipGetErrorString(e)); rc = -3; }
  }
  if (rc == 0) {
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&net->err_host), 64, hipHostMallocMapped);
    if (e == hipSuccess) { *net->err_host = 0u; e = hipHostGetDevicePointer(reinterpret_cast<void**>(&net->err_dev), net->err_host, 0); }
    if (e != hipSuccess) { (void)hipGetLastError(); mi355_set_error(std::string("unet_create: pinned error word: ") + hipGetErrorString(e)); rc = -3; }
  }
  if (rc) { delete net; return rc; }
  net->params = w.params;
  net->tensor_state_n = net->tensors.size();
  net->tensor_state.reset(new std::atomic<char>[net->tensor_state_n]);
  for (size_t i = 0; i < net->tensor_state_n; ++i) net->tensor_state[i].store(0, std::memory_order_relaxed);
  net->dev_weights = reinterpret_cast<char*>(dev_weights);
  net->dev_weights_bytes = (int64_t)w.cursor;
  *out = net;
  return 0;
}

mi355_unet::~mi355_unet() {
  if (err_host) (void)hipHostFree(err_host);
  for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (capture_stream) (void)hipStreamDestroy(capture_stream);
}

int unet_status(const mi355_unet* net, int clear) {
  if (!net || !net->err_host) return 0;
  const uint32_t v = *reinterpret_cast<volatile uint32_t*>(net->err_host);
  if (clear) *reinterpret_cast<volatile uint32_t*>(net->err_host) = 0u;
  if (v == 0u) return 0;
  mi355_set_error("a launch of this handle gave up a bounded counter wait of the persistent conv (hand-over stalled): its output is invalid"
                  " [error word " + std::to_string(v) + "]");
  return MI355_ERR_TIMEOUT;
}

WsLayout unet_ws_layout(const mi355_unet* net, int B) {
  const int mc = net->cfg.model_channels, esz = net->cfg.dtype == 0 ? 4 : 2, CH = net->cfg.dtype == 0 ? 16 : 32;
  WsLayout l{}; size_t c = 0;
  auto take = [&](size_t bytes) { size_t o = c; c = align_up(c + bytes, 256); return o; };
  l.temb = take((size_t)B * mc * 4);
  l.emb1 = take((size_t)B * 4 * mc * 4);
  l.emb2 = take((size_t)B * 4 * mc * 4);
  l.embp = take((size_t)B * net->emb_total * 4);
  l.gna = take((size_t)B * net->max_gn_c * 4);
  l.gnb = take((size_t)B * net->max_gn_c * 4);
  l.stats = take((size_t)B * net->stats_floats_per_image * 4);
  l.sites = take((size_t)B * net->site_floats_per_image * 4);
  l.arena = take(net->act_elems_per_image * (size_t)B * esz);
  if (net->cfg.differentiable) {   // backward scratch: one gradient per activation tensor + the temporaries sized by the plan
    const size_t S = (size_t)net->cfg.image_size * net->cfg.image_size;
    l.grads = take(net->act_elems_per_image * (size_t)B * esz);
    l.du = take(net->bwd_du_elems * (size_t)B * esz);
    l.tmp = take(net->bwd_tmp_elems * (size_t)B * esz);
    l.z = take(net->bwd_z_elems * (size_t)B * esz);
    l.dy = take(S * CH * (size_t)B * esz);
    l.ld = take(net->bwd_ld_floats * (size_t)B * 4);
  }
  l.total = c;
  return l;
}
static WsLayout ws_layout(const mi355_unet* net, int B) { return unet_ws_layout(net, B); }

int unet_embedding_table(const mi355_unet* net, const float* t_dev, int n, float* table, float* scratch, hipStream_t stream) {
  const int mc = net->cfg.model_channels;
  const char* W = net->dev_weights;
  auto WF = [&](size_t off) { return reinterpret_cast<const float*>(W + off); };
  float* temb = scratch; float* e1 = scratch + (size_t)n * mc; float* e2 = e1 + (size_t)n * 4 * mc;
  int rc;
  // emb2 = silu(time_embed(timestep_embedding(t))): the SiLU that opens every emb_layers is applied once here
  if ((rc = timestep_embedding_launch(t_dev, n, mc, 10000.f, temb, stream))) return rc;
  if ((rc = linear_launch(temb, WF(net->te_w0), WF(net->te_b0), e1, n, mc, 4 * mc, 0, 1, stream))) return rc;
  if ((rc = linear_launch(e1, WF(net->te_w2), WF(net->te_b2), e2, n, 4 * mc, 4 * mc, 0, 1, stream))) return rc;
  return linear_launch(e2, WF(net->emb_w), WF(net->emb_b), table, n, 4 * mc, net->emb_total, 0, 0, stream);
}

int64_t unet_workspace_bytes(const mi355_unet* net, int batch) { return (int64_t)ws_layout(net, batch).total; }

int unet_forward(const mi355_unet* net, const float* x, int Cx, const float* cond, int Cc, const float* t, float* out, int B,
                 void* workspace, int64_t workspace_bytes, hipStream_t stream, const UnetRun& run) {
  MI355_REQUIRE(net && x && t && out && workspace, -1, "unet_forward: null argument");
  MI355_REQUIRE(B > 0, -1, "unet_forward: batch must be positive");
  if (int rc = unet_status(net, 0)) return rc;   // an earlier launch of this handle gave up a counter wait
  MI355_REQUIRE(Cx + (cond ? Cc : 0) == net->cfg.in_channels, -2, "unet_forward: x/cond channels do not add up to in_channels");
  const WsLayout l = ws_layout(net, B);
  MI355_REQUIRE((int64_t)l.total <= workspace_bytes, -2, "unet_forward: workspace too small");
  MI355_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, -1, "unet_forward: workspace must be 256-byte aligned");
  const int dtype = net->cfg.dtype, esz = dtype == 0 ? 4 : 2;
  char* ws = reinterpret_cast<char*>(workspace);
  char* W = net->dev_weights;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
  auto WF = [&](size_t off) { return reinterpret_cast<const float*>(W + off); };
  auto TP = [&](int id) -> void* { return id < 0 ? nullptr : ws + l.arena + net->tensors[id].offset_per_image * (size_t)B * esz; };
  auto SP = [&](int id) -> float* { return F(l.stats) + net->tensors[id].stats_off_per_image * (size_t)B; };
  std::vector<int> gn_slots(net->tensors.size(), 0);   // partial-statistics slots each tensor's producer filled in THIS forward
  // GroupNorm sites the producing conv applied in its epilogue (small levels: ConvDesc::act_out); a tensor's raw copy is written only
  // if an op other than that site reads it
  std::vector<char> gn_done(net->ops.size(), 0);
  int euler_done = 0;   // the last conv's epilogue applied the sampler's Euler update (UnetRun::euler_x)
  std::vector<char> skip_fused(net->ops.size(), 0);   // second convs of small-level ResBlocks that carry the block's 1x1 skip conv in THIS forward
  std::vector<char> pro_off(net->ops.size(), 0);   // convs whose input arrives already normalised (16x16 level: applied IN PLACE by the producer)
  std::vector<int> readers(net->tensors.size(), 0);
  // apply-type GroupNorm sites (small images) by the tensors they read, and how many of a site's sources their producers have already applied
  std::vector<std::vector<int>> apply_sites(net->tensors.size());
  std::vector<char> site_parts(net->ops.size(), 0);
  for (size_t j = 0; j < net->ops.size(); ++j) {
    const PlanOp& o = net->ops[j];
    if (o.src0 >= 0) ++readers[o.src0];
    if (o.src1 >= 0) ++readers[o.src1];
    if (o.kind == OP_CONV && o.res >= 0) ++readers[o.res];
    if (o.kind == OP_GN && o.dst >= 0 && o.gn_site < 0) {
      apply_sites[o.src0].push_back((int)j);
      if (o.src1 >= 0) apply_sites[o.src1].push_back((int)j);
    }
  }
  int rc;
  auto mark = [&](const mi355_op_profile& r) {
    if (!run.prof) return;
    hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, stream);
    run.prof_events->push_back(e); run.prof->push_back(r);
  };
  if (run.prof) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, stream); run.prof_events->push_back(e); }
  // time embedding path (fp32): emb2 = silu(time_embed(timestep_embedding(t))) ; embp = all emb_layers linears
  // in the sampler loops every image shares the step time: one embedding row, broadcast with stride 0
  const int Be = run.t_uniform ? 1 : B, estride = run.t_uniform ? 0 : net->emb_total;
  const float* embp = run.emb_row ? run.emb_row : F(l.embp);
  if (!run.emb_row && (rc = unet_embedding_table(net, t, Be, F(l.embp), F(l.temb), stream))) return rc;
  const int S = net->cfg.image_size;
  // the first conv reads the caller's fp32 NCHW tensors itself where its kernel can (conv_edge bit 2): no packed copy, no pack launch
  bool in_direct = false;
  if (!net->cfg.differentiable && readers[net->in_tensor] == 1) {
    for (const PlanOp& o : net->ops) {
      if (o.kind != OP_CONV || o.src0 != net->in_tensor) continue;
      const PlanTensor& ti = net->tensors[net->in_tensor];
      ConvDesc c; c.dtype = dtype; c.src0 = TP(o.src0); c.C0 = ti.C; c.N = B; c.Hs = ti.H; c.Ws = ti.W; c.mode = o.mode; c.ks = o.ks; c.wsplit = net->wsplit;
      c.w = W + o.w_off; c.bias = WF(o.bias_off); c.Cout = o.Cout; c.out_mode = o.out_mode; c.out = TP(o.dst); c.knobs = &net->knobs;
      c.cin_real = net->cfg.in_channels;
      if (o.use_pro || o.emb_off >= 0 || o.res >= 0 || o.src1 >= 0) break;
      c.nchw0 = x; c.nchw_c0 = Cx; c.nchw1 = cond; c.nchw_c1 = cond ? Cc : 0;
      in_direct = conv_in_reads_nchw(c) == 0;
      break;
    }
  }
  if (in_direct) { if ((size_t)net->in_tensor < net->tensor_state_n) net->tensor_state[net->in_tensor].store((char)1, std::memory_order_relaxed); }
  else if ((rc = pack_nhwc_launch(dtype, x, Cx, cond, cond ? Cc : 0, B, S * S, net->in_pad, TP(net->in_tensor), stream))) return rc;
  { mi355_op_profile r{}; r.kind = MI355_OP_PRELUDE; mark(r); }
  // the conv a GroupNorm pass feeds is the next op of the plan: the pass warms the L2s with its weights (common.h l2_warm_wave)
  const int warm_mask = net->knobs.l2_warm;   // 1 = statistics / apply passes, 2 = finalize passes (measured: no gain, off)
  auto warm_next = [&](const PlanOp& op, const void*& wp, uint32_t& wb, int bit) {
    const size_t oi = (size_t)(&op - net->ops.data());
    if (!(warm_mask & bit) || oi + 1 >= net->ops.size() || net->ops[oi + 1].kind != OP_CONV) return;
    const PlanOp& nx = net->ops[oi + 1];
    const int cin = net->tensors[nx.src0].C + (nx.src1 >= 0 ? net->tensors[nx.src1].C : 0);
    wp = W + nx.w_off; wb = (uint32_t)conv_packed_weight_bytes(dtype, nx.Cout, cin, nx.ks, net->wsplit);
  };
  for (const PlanOp& op : net->ops) {
    mi355_op_profile r{};
    const PlanTensor& s0 = net->tensors[op.src0];
    const int C1 = op.src1 >= 0 ? net->tensors[op.src1].C : 0;
    if (op.kind == OP_GN && gn_done[(size_t)(&op - net->ops.data())]) {
      rc = 0;   // applied by the producing conv's epilogue
      r.kind = MI355_OP_GN; r.cin = s0.C + C1; r.h = s0.H; r.w = s0.W; r.bytes = 0; r.tile_m = r.tile_n = -1;   // (tile -1: nothing was launched for this op)
    } else if (op.kind == OP_GN && op.fin_ok && op.dst < 0 && gn_slots[op.src0] > 0 && (op.src1 < 0 || gn_slots[op.src1] > 0)) {
      GnFinDesc g; g.stats0 = SP(op.src0); g.slots0 = gn_slots[op.src0]; g.C0 = s0.C;
      if (op.src1 >= 0) { g.stats1 = SP(op.src1); g.slots1 = gn_slots[op.src1]; g.C1 = C1; }
      g.N = B; g.HW = s0.H * s0.W; g.gamma = WF(op.gamma_off); g.beta = WF(op.beta_off);
      if (op.film_emb_off >= 0) { g.film = embp + op.film_emb_off; g.film_stride = estride; }
      g.a = F(l.gna); g.b = F(l.gnb);
      warm_next(op, g.warm, g.warm_bytes, 2);
      rc = gn_finalize_launch(g, stream);
      r.kind = MI355_OP_GN; r.cin = s0.C + C1; r.h = s0.H; r.w = s0.W;
      r.bytes = 0;   // no activation traffic: the statistics came with the producers' epilogues
    } else if (op.kind == OP_GN) {
      GnDesc g; g.dtype = dtype; g.src0 = TP(op.src0); g.C0 = s0.C; g.src1 = TP(op.src1); g.C1 = C1;
      g.N = B; g.HW = s0.H * s0.W; g.gamma = WF(op.gamma_off); g.beta = WF(op.beta_off);
      if (op.film_emb_off >= 0) { g.film = embp + op.film_emb_off; g.film_stride = estride; }
      g.a = F(l.gna); g.b = F(l.gnb);
      if (op.gn_site >= 0) {   // differentiable plan: this site's own (a, b, mean, rstd)
        float* sp = F(l.sites) + net->site_off[op.gn_site] * (size_t)B;
        const size_t Cs = (size_t)net->site_C[op.gn_site];
        g.a = sp; g.b = sp + (size_t)B * Cs; g.mean = sp + (size_t)2 * B * Cs; g.rstd = g.mean + (size_t)B * 32;
      }
      if (op.dst >= 0) { g.y = TP(op.dst); g.y_silu = op.pro_silu; }
      warm_next(op, g.warm, g.warm_bytes, 1);
      rc = gn_affine_launch(g, stream);
      r.kind = MI355_OP_GN; r.cin = s0.C + C1; r.h = s0.H; r.w = s0.W;
      r.bytes = (double)B * s0.H * s0.W * (s0.C + C1) * esz * (op.dst >= 0 ? 2 : 1);
    } else if (op.kind == OP_CONV && op.carrier >= 0 && [&]() {
                 // 1x1 skip_connection of a small-level ResBlock: does the second conv's launch take it along?  (same description as below,
                 // as far as eligibility looks: shapes, batch, precision, knobs)
                 const PlanOp& c2 = net->ops[op.carrier];
                 const PlanTensor& y2 = net->tensors[c2.src0];
                 ConvDesc c; c.dtype = dtype; c.src0 = TP(c2.src0); c.C0 = y2.C; c.N = B; c.Hs = y2.H; c.Ws = y2.W; c.mode = c2.mode; c.ks = c2.ks;
                 c.wsplit = net->wsplit; c.w = W + c2.wf_off; c.bias = WF(c2.bf_off); c.Cout = c2.Cout; c.out_mode = c2.out_mode; c.out = TP(c2.dst);
                 c.knobs = &net->knobs; c.err = net->err_dev;
                 c.skip_src0 = TP(op.src0); c.skip_C0 = s0.C; c.skip_src1 = TP(op.src1); c.skip_C1 = C1;
                 return conv_fused_skip_ok(c) == 0;
               }()) {
      skip_fused[op.carrier] = 1;   // nothing to launch: the tensor is never written
      gn_done[(size_t)(&op - net->ops.data())] = 1;   // (counts as a launch that did not happen)
      if (op.dst >= 0 && (size_t)op.dst < net->tensor_state_n) net->tensor_state[op.dst].store((char)1, std::memory_order_relaxed);
      rc = 0;
      r.kind = MI355_OP_CONV; r.ks = op.ks; r.cin = s0.C + C1; r.cout = op.Cout; r.h = s0.H; r.w = s0.W; r.tile_m = r.tile_n = -1;
    } else if (op.kind == OP_CONV) {
      ConvDesc c; c.dtype = dtype; c.src0 = TP(op.src0); c.C0 = s0.C; c.src1 = TP(op.src1); c.C1 = C1;
      c.N = B; c.Hs = s0.H; c.Ws = s0.W; c.mode = op.mode; c.ks = op.ks; c.wsplit = net->wsplit;
      const size_t oi = (size_t)(&op - net->ops.data());
      if (op.use_pro && !pro_off[oi]) { c.pro_a = F(l.gna); c.pro_b = F(l.gnb); c.pro_silu = op.pro_silu; }
      if (op.use_pro && !pro_off[oi] && op.gn_site >= 0) {
        float* sp = F(l.sites) + net->site_off[op.gn_site] * (size_t)B;
        c.pro_a = sp; c.pro_b = sp + (size_t)B * net->site_C[op.gn_site];
      }
      c.w = W + op.w_off; c.bias = WF(op.bias_off); c.Cout = op.Cout;
      if (op.src0 == net->in_tensor) {
        c.cin_real = net->cfg.in_channels;
        if (in_direct) { c.nchw0 = x; c.nchw_c0 = Cx; c.nchw1 = cond; c.nchw_c1 = cond ? Cc : 0; }
      }
      if (op.out_mode == OUT_NCHW_F32 && run.euler_x) { c.axpy_x = run.euler_x; c.axpy_scale = run.euler_dt; c.axpy_done = &euler_done; }
      if (op.emb_off >= 0) { c.emb = embp + op.emb_off; c.emb_stride = estride; }
      if (op.res >= 0) { c.res = TP(op.res); c.res_mode = op.res_mode; }
      c.out_mode = op.out_mode;
      c.knobs = &net->knobs; c.err = net->err_dev;
      c.out = op.out_mode == OUT_NHWC ? TP(op.dst) : (void*)out;
      if (skip_fused[oi]) {   // the ResBlock's 1x1 skip conv rides in this launch (its op was skipped above)
        const PlanOp& sk = net->ops[op.skip_op];
        c.skip_src0 = TP(sk.src0); c.skip_C0 = net->tensors[sk.src0].C;
        c.skip_src1 = TP(sk.src1); c.skip_C1 = sk.src1 >= 0 ? net->tensors[sk.src1].C : 0;
        c.w = W + op.wf_off; c.bias = WF(op.bf_off); c.res = nullptr; c.res_mode = RES_NONE;
      }
      int slots = 0, act_done = 0;
      if (op.dst >= 0 && net->tensors[op.dst].stats_cap) { c.gn_stats = SP(op.dst); c.gn_slots_cap = net->tensors[op.dst].stats_cap; }
      bool try_act = false;
      size_t act_consumer = 0;
      if (op.dst >= 0 && op.out_mode == OUT_NHWC && op.res < 0 && oi + 2 < net->ops.size() && readers[op.dst] == 2) {
        // statistics-type site (larger images) read by exactly one prologue conv: where the persistent kernel's tile is the whole
        // image (16x16) it normalises its own output in place, the site's launch disappears and the consumer runs prologue-free
        const PlanOp& g = net->ops[oi + 1];
        if (g.kind == OP_GN && g.fin_ok && g.dst < 0 && g.src0 == op.dst && g.src1 < 0 && g.gn_site < 0) {
          for (size_t j = oi + 2; j < net->ops.size() && j <= oi + 3; ++j) {
            const PlanOp& cn = net->ops[j];
            if (cn.kind == OP_CONV && cn.use_pro && cn.src0 == op.dst && cn.src1 < 0 && cn.gn_site < 0) { act_consumer = j; break; }
          }
          if (act_consumer) {
            c.act_out = c.out; c.act_raw = 0; c.act_gamma = WF(g.gamma_off); c.act_beta = WF(g.beta_off);
            if (g.film_emb_off >= 0) { c.act_film = embp + g.film_emb_off; c.act_film_stride = estride; }
            c.act_silu = net->ops[act_consumer].pro_silu;
            try_act = true;
          }
        }
      }
      int fused_site[2] = {-1, -1};
      if (!try_act && op.dst >= 0 && op.out_mode == OUT_NHWC) {
        // The apply-type GroupNorm sites (small images) that read this conv's output: the one that follows it (in_layers / out_layers norm of
        // the next conv, unet.py:196-212) and, for a skip connection, the norm of the up path's concat (unet.py:650), whose groups are whole
        // inside each source when both channel counts are multiples of the group width: each producer then applies its own channels.
        for (int gi : apply_sites[op.dst]) {
          const PlanOp& g = net->ops[gi];
          const bool cat = g.src1 >= 0;
          const int Cg = net->tensors[g.src0].C + (cat ? net->tensors[g.src1].C : 0);
          const int coff = g.src0 == op.dst ? 0 : net->tensors[g.src0].C;
          if (cat) {
            const int cpg = Cg / 32;
            if (!(net->knobs.gn_epilogue & 4) || g.film_emb_off >= 0 || g.src0 == g.src1 || Cg % 32 || net->tensors[g.src0].C % cpg || net->tensors[g.src1].C % cpg) continue;
          }
          if (!c.act_out) {
            c.act_out = TP(g.dst); c.act_gamma = WF(g.gamma_off) + coff; c.act_beta = WF(g.beta_off) + coff;
            if (g.film_emb_off >= 0) { c.act_film = embp + g.film_emb_off; c.act_film_stride = estride; }
            c.act_silu = g.pro_silu; c.act_stride = Cg; c.act_coff = coff; c.act_cpg = Cg / 32;
            if (!cat || coff == 0) warm_next(g, c.warm, c.warm_bytes, 1);
            fused_site[0] = gi;
          } else if (!c.act2_out && g.film_emb_off < 0) {
            c.act2_out = TP(g.dst); c.act2_gamma = WF(g.gamma_off) + coff; c.act2_beta = WF(g.beta_off) + coff;
            c.act2_silu = g.pro_silu; c.act2_stride = Cg; c.act2_coff = coff; c.act2_cpg = Cg / 32;
            fused_site[1] = gi;
          }
        }
        if (c.act_out) {
          // the raw tensor is written unless the one site asked for is its only reader (with two sites asked for the launch may still take one)
          c.act_raw = c.act2_out ? 1 : readers[op.dst] > 1;
          try_act = true;
        }
      }
      rc = conv_launch(c, stream, &slots, try_act ? &act_done : nullptr);
      if (act_done && act_consumer) { gn_done[oi + 1] = 1; pro_off[act_consumer] = 1; }
      else if (act_done) {
        for (int k = 0; k < 2; ++k) {
          if (!(act_done & (1 << k)) || fused_site[k] < 0) continue;
          const PlanOp& g = net->ops[fused_site[k]];
          if (++site_parts[fused_site[k]] == (g.src1 >= 0 ? 2 : 1)) gn_done[fused_site[k]] = 1;
        }
      }
      if (op.dst >= 0 && (size_t)op.dst < net->tensor_state_n) net->tensor_state[op.dst].store((char)(!act_done ? 0 : (act_consumer ? 2 : (c.act_raw ? 0 : 1))), std::memory_order_relaxed);
      if (op.dst >= 0) gn_slots[op.dst] = slots;
      if (run.prof) {
        const ConvGeom cg = conv_geometry(c);
        const int cin = s0.C + C1;
        r.kind = MI355_OP_CONV; r.ks = op.ks; r.cin = cin; r.cout = op.Cout; r.h = cg.Ho; r.w = cg.Wo; r.tile_m = cg.BM; r.tile_n = cg.BN;
        r.flops = 2.0 * B * cg.Ho * cg.Wo * (double)op.Cout * cin * op.ks * op.ks;
        r.bytes = ((double)B * s0.H * s0.W * cin + (double)B * cg.Ho * cg.Wo * op.Cout) * esz + (double)op.Cout * cin * op.ks * op.ks * esz;
        if (skip_fused[oi]) {   // the ResBlock's 1x1 skip conv this launch carried
          r.flops += 2.0 * B * cg.Ho * cg.Wo * (double)op.Cout * (c.skip_C0 + c.skip_C1);
          r.bytes += ((double)B * cg.Ho * cg.Wo + (double)op.Cout) * (c.skip_C0 + c.skip_C1) * esz;
        }
      }
    } else if (op.kind == OP_ATTN) {
      AttnDesc a; a.dtype = dtype; a.qkv = TP(op.src0); a.out = TP(op.dst); a.N = B; a.T = s0.H * s0.W;
      a.heads = op.heads; a.ch = op.ch; a.new_order = net->cfg.use_new_attention_order;
      rc = attention_launch(a, stream);
      r.kind = MI355_OP_ATTN; r.cin = 3 * op.heads * op.ch; r.cout = op.heads * op.ch; r.h = s0.H; r.w = s0.W;
      r.flops = 4.0 * B * (double)a.T * a.T * op.heads * op.ch;
      r.bytes = 4.0 * B * a.T * op.heads * op.ch * esz;
    } else if (op.kind == OP_ATTN_FUSED) {
      AttnFusedDesc a; a.dtype = dtype; a.x = TP(op.src0); a.ga = F(l.gna); a.gb = F(l.gnb); a.w = W + op.w_off; a.bias = WF(op.bias_off);
      a.out = TP(op.dst); a.N = B; a.T = s0.H * s0.W; a.C = s0.C; a.heads = op.heads; a.ch = op.ch;
      a.new_order = net->cfg.use_new_attention_order; a.knobs = &net->knobs;
      rc = attn_fused_launch(a, stream);
      r.kind = MI355_OP_ATTN; r.cin = s0.C; r.cout = s0.C; r.h = s0.H; r.w = s0.W; r.ks = 1;   // ks = 1 marks the fused form
      r.flops = 2.0 * B * (double)a.T * 3.0 * s0.C * s0.C + 4.0 * B * (double)a.T * a.T * s0.C;
      r.bytes = 2.0 * B * a.T * (double)s0.C * esz + 3.0 * s0.C * s0.C * esz;
    } else if (op.kind == OP_POOLAFF) {
      const float* pa = F(l.gna); const float* pb = F(l.gnb);
      if (op.gn_site >= 0) { pa = F(l.sites) + net->site_off[op.gn_site] * (size_t)B; pb = pa + (size_t)B * net->site_C[op.gn_site]; }
      rc = affine_pool_launch(dtype, TP(op.src0), pa, pb, op.pro_silu, TP(op.dst), B, s0.H, s0.W, s0.C, stream);
      r.kind = MI355_OP_RESAMPLE; r.cin = s0.C; r.h = s0.H; r.w = s0.W;
      r.bytes = 1.25 * B * s0.H * s0.W * (double)s0.C * esz;
    } else {
      rc = resample_launch(dtype, TP(op.src0), TP(op.dst), B, s0.H, s0.W, s0.C, op.mode, stream);
      r.kind = MI355_OP_RESAMPLE; r.cin = s0.C; r.h = s0.H; r.w = s0.W;
    }
    if (rc) return rc;
    mark(r);
  }
  if (run.euler_x && !euler_done) {
    const int64_t n_out = (int64_t)B * net->cfg.out_channels * S * S;
    if ((rc = euler_step_launch(run.euler_x, out, run.euler_dt, n_out, stream))) return rc;
  }
  {
    int64_t skipped = in_direct ? 1 : 0;   // (the Euler update is the sampler's launch, not the forward's: not counted either way)
    for (char d : gn_done) skipped += d;
    net->last_launches = net->launches - skipped - (run.emb_row ? 4 : 0);
  }
  return 0;
}
