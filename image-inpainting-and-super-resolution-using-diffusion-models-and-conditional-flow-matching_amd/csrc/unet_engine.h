// U-Net layer plan + executor (host C++).  See unet_engine.hip.
#pragma once
#include <string>
#include <vector>

#include "../../include/mi355_sampler.h"
#include <atomic>
#include <memory>
#include <mutex>

#include "ops.h"

struct ParamInfo {
  std::string name;
  std::vector<int64_t> shape;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

// Parameter inventory in reference state_dict order (UNetModel.__init__, unet.py:564-706).
int unet_enumerate_params(const mi355_unet_config& cfg, std::vector<ParamInfo>& out);

struct PlanTensor {
  int C, H, W; bool f32; size_t offset_per_image;   // element type T unless f32
  // fused GroupNorm statistics (common.h GnPartial): set when a GroupNorm site that takes its (a, b) from partial sums reads this
  // tensor; the producing conv fills stats[N][slots][C/4][2] at stats_off_per_image * N floats into the statistics arena
  int stats_cap = 0; size_t stats_off_per_image = 0;
};

enum OpKind { OP_GN = 0, OP_CONV = 1, OP_ATTN = 2, OP_RESAMPLE = 3, OP_POOLAFF = 4, OP_ATTN_FUSED = 5 };

struct PlanOp {
  int kind;
  // common
  int src0 = -1, src1 = -1, dst = -1;
  // GN
  size_t gamma_off = 0, beta_off = 0; int film_emb_off = -1;
  int fin_ok = 0;   // this site's groups are whole channel quads of each source: (a, b) may come from the producers' partial sums
  int gn_site = -1; // differentiable plans: OP_GN = its site; OP_CONV / OP_POOLAFF with a prologue = the site whose (a, b) they apply
  size_t wT_off = 0; int cin_pad = 0;   // differentiable plans: data-gradient weights (conv_pack_weights_dgrad) and their output channels
  // conv
  int mode = 0, ks = 3, Cout = 0; size_t w_off = 0, bias_off = 0;
  int use_pro = 0, pro_silu = 0; int emb_off = -1; int res = -1, res_mode = 0; int out_mode = 0;
  // attention
  int heads = 0, ch = 0;
  // small levels: a ResBlock's 1x1 skip_connection op names the second conv that can carry it (carrier), and that conv names the skip op
  // (skip_op) and holds the fused weight image / summed bias (conv_pack_weights_skip); decided per launch (conv_fused_skip_ok)
  int carrier = -1, skip_op = -1; size_t wf_off = 0, bf_off = 0;
};

struct mi355_unet {
  mi355_unet_config cfg;
  mi355_debug_config knobs;        // diagnostic switches, copied at creation (cfg.debug is not kept)
  // error word of the handle's launches: pinned host memory the kernels write through (mi355_unet_status); dev = its device address
  uint32_t* err_host = nullptr; uint32_t* err_dev = nullptr;
  ~mi355_unet();
  std::vector<ParamInfo> params;
  std::vector<PlanTensor> tensors;
  std::vector<PlanOp> ops;
  char* dev_weights = nullptr;
  int64_t dev_weights_bytes = 0;
  // fp32 side tables in the weight blob
  size_t te_w0 = 0, te_b0 = 0, te_w2 = 0, te_b2 = 0, emb_w = 0, emb_b = 0;
  int emb_total = 0;   // concatenated emb_layers output channels of all ResBlocks
  int in_pad = 0;      // first conv's padded input channels
  int max_gn_c = 0;
  size_t act_elems_per_image = 0;  // activation arena (elements of T) per image
  size_t stats_floats_per_image = 0;   // partial GroupNorm statistics arena (fp32) per image
  // differentiable plans (reconstruction guidance): per-site (a, b, mean, rstd) kept by the forward, and the backward's scratch sizes
  std::vector<int> site_C; std::vector<size_t> site_off;   // site k: a[C] | b[C] | mean[32] | rstd[32] floats per image at site_off[k]
  size_t site_floats_per_image = 0;
  size_t bwd_du_elems = 0, bwd_tmp_elems = 0, bwd_z_elems = 0, bwd_ld_floats = 0;   // per image
  int in_tensor = -1, out_channels = 0;
  // stats per image
  double conv_flops = 0, attn_flops = 0, act_bytes = 0, weight_bytes = 0;
  int wsplit = 0;         // MI355_BF16X2: conv / qkv weights packed as hi | lo bf16 halves along K (ConvDesc::wsplit)
  int64_t launches = 0;   // device launches of one forward as planned (an upper bound: a GroupNorm pass a conv epilogue absorbed is not launched)
  mutable int64_t last_launches = 0;   // what the most recent forward really launched (0 before the first)
  // what the most recent forward left in each activation tensor (diagnostics: mi355_unet_read_tensor): 0 = the tensor as the reference
  // defines it, 1 = never written (its only reader, a GroupNorm site, was fused into the producing conv's epilogue), 2 = overwritten in
  // place by silu?(GroupNorm(.)) (16x16 level).  Sized at build.  The record is PER HANDLE (of whichever forward wrote it last, on any workspace or
  // batch size): relaxed atomic bytes, so concurrent forwards of one handle from several threads are race-free and only blur the diagnostic.
  mutable std::unique_ptr<std::atomic<char>[]> tensor_state;
  size_t tensor_state_n = 0;
  // knobs.sampler_graph: the flow-matching Euler loop of mi355_cfm_euler_sample as instantiated hipGraphs, one per (workspace, batch, schedule,
  // condition) the handle has been driven with (a few entries, least recently used replaced).  Guarded by graph_mu: the cache is the one piece of
  // handle state a sampler call changes.
  struct SamplerGraph { uint64_t key[8] = {0}; hipGraphExec_t exec = nullptr; uint64_t stamp = 0; int64_t launches = 0; };
  mutable std::mutex graph_mu;
  mutable std::vector<SamplerGraph> graphs;
  mutable hipStream_t capture_stream = nullptr;
  mutable uint64_t graph_clock = 0;
};

// Per-call options of unet_forward.  They are arguments, not handle state: a handle is immutable after unet_build, so one
// handle may be driven from several host threads / streams (each with its own workspace).
struct UnetRun {
  int t_uniform = 0;   // sampler loops: t[0] holds for the whole batch (one embedding row, stride-0 broadcast)
  const float* emb_row = nullptr;   // sampler loops: this step's row of a precomputed table of all emb_layers outputs
                                    // (unet_embedding_table): the four time-embedding launches are skipped
  // sampler loops (flow-matching Euler): x += euler_dt * (network output) in the last conv's epilogue where that conv runs on the streaming kernel
  // (the output tensor is then not written); else unet_forward adds the step launch itself.  euler_x may be the network input x.
  float* euler_x = nullptr; float euler_dt = 0.f;
  // optional per-op profiling (mi355_unet_profile)
  std::vector<mi355_op_profile>* prof = nullptr;
  std::vector<hipEvent_t>* prof_events = nullptr;
};

int unet_build(const mi355_unet_config& cfg, const float* const* params_host, int n_params, void* dev_weights,
               int64_t dev_weights_bytes, hipStream_t stream, mi355_unet** out);
int64_t unet_weight_bytes(const mi355_unet_config& cfg);
int unet_status(const mi355_unet* net, int clear);   // 0 or MI355_ERR_TIMEOUT (+ message)
int64_t unet_workspace_bytes(const mi355_unet* net, int batch);
struct WsLayout { size_t temb, emb1, emb2, embp, gna, gnb, stats, sites, arena, grads, du, tmp, z, dy, ld, total; };
WsLayout unet_ws_layout(const mi355_unet* net, int B);
// (d out / d x)^T grad_out of the last unet_forward on this workspace (differentiable plans only; unet_backward.hip)
int unet_backward(const mi355_unet* net, const float* grad_out, float* grad_x, int Cx, int batch, void* workspace, int64_t workspace_bytes,
                  hipStream_t stream);
// All emb_layers outputs (UNetModel.time_embed + every ResBlock's emb_layers, unet.py:564-569,293-299) for n step times at once:
// table [n][emb_total] fp32; scratch = n * 9 * model_channels floats.  The sampler loops know every step time in advance.
int unet_embedding_table(const mi355_unet* net, const float* t_dev, int n, float* table, float* scratch, hipStream_t stream);
int unet_forward(const mi355_unet* net, const float* x, int Cx, const float* cond, int Cc, const float* t, float* out, int batch,
                 void* workspace, int64_t workspace_bytes, hipStream_t stream, const UnetRun& run = UnetRun());
