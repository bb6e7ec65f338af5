// extern "C" boundary of libmi355_sampler.so (see include/mi355_sampler.h) and the sampler loops.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "unet_engine.h"

static thread_local std::string g_err;
void mi355_set_error(const std::string& msg) { g_err = msg; }

namespace {
inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace

// The sampler loops know every step time in advance: all emb_layers outputs of up to EMB_TABLE_STEPS steps are computed by four
// launches before the loop instead of four launches per step (longer schedules fall back to the per-step path).
constexpr int EMB_TABLE_STEPS = 1024;
static size_t emb_table_bytes(const mi355_unet* net) {
  return al256((size_t)EMB_TABLE_STEPS * 4) + al256((size_t)EMB_TABLE_STEPS * ((size_t)net->emb_total + 9 * (size_t)net->cfg.model_channels) * 4);
}

static UnetRun uniform_t_run();

const mi355_debug_config& mi355_default_debug() {
  static const mi355_debug_config d = [] { mi355_debug_config c; mi355_debug_defaults(&c); return c; }();
  return d;
}

extern "C" {

int mi355_version(void) { return 105; }
void mi355_debug_defaults(mi355_debug_config* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->conv_ws = 1; c->conv_small = 15; c->conv_min_wgs = 512; c->conv_stagger = 0; c->conv_ablate = 0; c->conv_spin_limit = 1 << 22;
  c->conv_time_reps = 0; c->gn_apply_max_hw = 64; c->gn_fuse = 1; c->l2_warm = 1; c->attn_fused = 1; c->gn_epilogue = 7; c->conv_pp = 45; c->conv_edge = 15;
}
int mi355_unet_status(mi355_unet* net, int clear) {
  if (!net) { mi355_set_error("null handle"); return -1; }
  return unet_status(net, clear);
}
const char* mi355_last_error(void) { return g_err.c_str(); }

int mi355_unet_param_count(const mi355_unet_config* cfg) {
  if (!cfg) { mi355_set_error("null config"); return -1; }
  std::vector<ParamInfo> p;
  int rc = unet_enumerate_params(*cfg, p);
  return rc ? rc : (int)p.size();
}

int mi355_unet_param_info(const mi355_unet_config* cfg, int index, char* name, int name_cap, int64_t shape[4], int* ndim) {
  if (!cfg || !name || !shape || !ndim) { mi355_set_error("null argument"); return -1; }
  std::vector<ParamInfo> p;
  if (int rc = unet_enumerate_params(*cfg, p)) return rc;
  MI355_REQUIRE(index >= 0 && index < (int)p.size(), -1, "param index out of range");
  MI355_REQUIRE((int)p[index].name.size() + 1 <= name_cap, -1, "name buffer too small");
  std::strcpy(name, p[index].name.c_str());
  *ndim = (int)p[index].shape.size();
  for (int i = 0; i < 4; ++i) shape[i] = i < *ndim ? p[index].shape[i] : 1;
  return 0;
}

int64_t mi355_unet_weight_bytes(const mi355_unet_config* cfg) {
  if (!cfg) { mi355_set_error("null config"); return -1; }
  return unet_weight_bytes(*cfg);
}

int mi355_unet_create(const mi355_unet_config* cfg, const float* const* params_host, int n_params, void* dev_weights,
                      int64_t dev_weights_bytes, void* stream, mi355_unet** out) {
  if (!cfg) { mi355_set_error("null config"); return -1; }
  return unet_build(*cfg, params_host, n_params, dev_weights, dev_weights_bytes, S(stream), out);
}

void mi355_unet_destroy(mi355_unet* net) { delete net; }

int64_t mi355_unet_workspace_bytes(const mi355_unet* net, int batch) {
  if (!net || batch <= 0) { mi355_set_error("bad argument"); return -1; }
  // + sampler scratch: t[B], eps/v [B,Cout,H,W], none_like [B,Cin,H,W], and the per-step time-embedding table of the sampler loops
  const size_t hw = (size_t)net->cfg.image_size * net->cfg.image_size;
  // (+ the graph form's resident state [B,Cout,H,W], mi355_debug_config::sampler_graph)
  const size_t scratch = al256((size_t)batch * 4) + 2 * al256((size_t)batch * 32 * hw * 4) + emb_table_bytes(net) +
                         al256((size_t)batch * net->cfg.out_channels * hw * 4);
  return unet_workspace_bytes(net, batch) + (int64_t)scratch;
}

int mi355_unet_forward(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels, const float* t,
                       float* out, int batch, void* workspace, int64_t workspace_bytes, void* stream) {
  return unet_forward(net, x, x_channels, cond, cond_channels, t, out, batch, workspace, workspace_bytes, S(stream));
}

int mi355_unet_forward_t(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels, float t, float* out,
                         int batch, void* workspace, int64_t workspace_bytes, void* stream) {
  MI355_REQUIRE(net && workspace && batch > 0, -1, "unet_forward_t: bad argument");
  const WsLayout l = unet_ws_layout(net, batch);
  MI355_REQUIRE((int64_t)l.total <= workspace_bytes, -2, "unet_forward_t: workspace too small");
  // the step time lives in the first word of the workspace's emb2 region until the embedding kernels have read it (they write
  // temb -> emb1 -> emb2 in that order, t is read by the first one only); every image shares it: ONE embedding row
  float* tdev = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + l.emb2);
  if (int rc = fill_launch(tdev, t, 1, S(stream))) return rc;
  return unet_forward(net, x, x_channels, cond, cond_channels, tdev, out, batch, workspace, workspace_bytes, S(stream), uniform_t_run());
}

int mi355_unet_vjp(mi355_unet* net, const float* grad_out, float* grad_x, int x_channels, int batch, void* workspace, int64_t workspace_bytes,
                   void* stream) {
  MI355_REQUIRE(net && workspace, -1, "unet_vjp: null argument");
  // same layout as mi355_unet_forward: the engine workspace starts at `workspace` (only the sampler loops carve their scratch in front)
  const int64_t engine_bytes = unet_workspace_bytes(net, batch);
  MI355_REQUIRE(workspace_bytes >= engine_bytes, -2, "unet_vjp: workspace too small");
  return unet_backward(net, grad_out, grad_x, x_channels, batch, workspace, workspace_bytes, S(stream));
}

int mi355_unet_plan_op(const mi355_unet* net, int index, int32_t fields[16]) {
  MI355_REQUIRE(net && fields, -1, "plan_op: null argument");
  if (index < 0 || index >= (int)net->ops.size()) return -1;
  const PlanOp& o = net->ops[index];
  const int32_t v[16] = {o.kind, o.src0, o.src1, o.dst, o.mode, o.ks, o.Cout, o.use_pro, o.pro_silu, o.res, o.res_mode, o.gn_site, o.heads, o.ch,
                         o.dst >= 0 ? net->tensors[o.dst].C : 0, o.dst >= 0 ? net->tensors[o.dst].H : 0};
  for (int i = 0; i < 16; ++i) fields[i] = v[i];
  return (int)net->ops.size();
}

int mi355_unet_read_tensor(const mi355_unet* net, int tensor, int gradient, float* out, int batch, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  MI355_REQUIRE(net && out && workspace, -1, "read_tensor: null argument");
  MI355_REQUIRE(tensor >= 0 && tensor < (int)net->tensors.size(), -1, "read_tensor: tensor index out of range");
  MI355_REQUIRE(!gradient || net->cfg.differentiable, -4, "read_tensor: gradients exist in differentiable plans only");
  const WsLayout l = unet_ws_layout(net, batch);
  MI355_REQUIRE((int64_t)l.total <= workspace_bytes, -2, "read_tensor: workspace too small");
  const PlanTensor& t = net->tensors[tensor];
  const int tstate = (!gradient && (size_t)tensor < net->tensor_state_n) ? (int)net->tensor_state[tensor].load(std::memory_order_relaxed) : 0;
  if (tstate) {
    mi355_set_error(tstate == 1
                        ? "read_tensor: the last forward did not materialise this tensor (a conv output whose only reader, a GroupNorm site, ran in the conv's "
                          "epilogue; a 1x1 skip conv that rode in the next 3x3 conv; the packed network input the first conv read from the caller's tensor): "
                          "create the handle with debug.gn_epilogue = 0, conv_small = 7, conv_edge = 3 to inspect it"
                        : "read_tensor: the last forward normalised this conv output in place (16x16 level: it holds silu(GroupNorm(.)), not the conv's result): "
                          "create the handle with debug.gn_epilogue = 0 (or 1) to inspect it");
    return MI355_ERR_UNSUPPORTED;
  }
  const int esz = net->cfg.dtype == 0 ? 4 : 2;
  const char* p = reinterpret_cast<const char*>(workspace) + (gradient ? l.grads : l.arena) + t.offset_per_image * (size_t)batch * esz;
  return unpack_nchw_launch(net->cfg.dtype, p, batch, t.H * t.W, t.C, out, S(stream));
}

int mi355_unet_get_stats(const mi355_unet* net, int batch, mi355_unet_stats* out) {
  if (!net || !out) { mi355_set_error("null argument"); return -1; }
  out->launches = net->last_launches > 0 ? net->last_launches : net->launches;   // the most recent forward's real count once there was one
  out->conv_flops = net->conv_flops * batch;
  out->attn_flops = net->attn_flops * batch;
  out->act_bytes = net->act_bytes * batch;
  out->weight_bytes = net->weight_bytes;
  return 0;
}

int mi355_unet_profile(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels, const float* t,
                       float* out, int batch, void* workspace, int64_t workspace_bytes, void* stream, mi355_op_profile* recs,
                       int cap) {
  MI355_REQUIRE(net && recs && cap > 0, -1, "unet_profile: bad argument");
  std::vector<mi355_op_profile> prof;
  std::vector<hipEvent_t> ev;
  UnetRun run; run.prof = &prof; run.prof_events = &ev;
  int rc = unet_forward(net, x, x_channels, cond, cond_channels, t, out, batch, workspace, workspace_bytes, S(stream), run);
  // An event record is itself a packet the queue has to retire: an interval between two events holds one such gap besides the
  // op.  Calibrate it on this stream (back-to-back records with nothing in between) and take it off every interval, so the
  // per-op times agree with rocprofv3's kernel durations.
  constexpr int NCAL = 17;
  hipEvent_t cal[NCAL];
  for (int i = 0; i < NCAL; ++i) { (void)hipEventCreate(&cal[i]); (void)hipEventRecord(cal[i], S(stream)); }
  hipError_t e = hipStreamSynchronize(S(stream));
  if (rc == 0 && e == hipSuccess) {
    float gap = 0.f;
    for (int i = 0; i + 1 < NCAL; ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, cal[i], cal[i + 1]); gap += ms; }
    gap /= (float)(NCAL - 1);
    for (size_t i = 0; i + 1 < ev.size() && i < prof.size(); ++i) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      prof[i].ms = ms > 2.f * gap ? ms - gap : 0.5f * ms;
    }
  }
  for (int i = 0; i < NCAL; ++i) (void)hipEventDestroy(cal[i]);
  for (auto h : ev) (void)hipEventDestroy(h);
  if (rc) return rc;
  if (e != hipSuccess) { mi355_set_error(hipGetErrorString(e)); return -3; }
  const int n = (int)prof.size();
  for (int i = 0; i < n && i < cap; ++i) recs[i] = prof[i];
  return n;
}

// ---- sampler loops --------------------------------------------------------------------------------

struct Scratch { float* t; float* v; float* none; float* tsteps; float* embtab; float* xstate; char* unet_ws; int64_t unet_bytes; };
// every image of a sampler step shares the step time: the engine computes ONE embedding row (stride-0 broadcast)
static UnetRun uniform_t_run() { UnetRun r; r.t_uniform = 1; return r; }
// table[k] = emb_layers outputs for step time t_host[k] (k < n <= EMB_TABLE_STEPS); returns the table or null (per-step path)
static int make_emb_table(const mi355_unet* net, const Scratch& sc, const float* t_host, int n, hipStream_t s, const float** table) {
  *table = nullptr;
  if (n <= 0 || n > EMB_TABLE_STEPS) return 0;
  MI355_CHECK_HIP(hipMemcpyAsync(sc.tsteps, t_host, (size_t)n * 4, hipMemcpyHostToDevice, s));
  float* scratch = sc.embtab + (size_t)EMB_TABLE_STEPS * net->emb_total;
  if (int rc = unet_embedding_table(net, sc.tsteps, n, sc.embtab, scratch, s)) return rc;
  *table = sc.embtab;
  return 0;
}
static int carve(mi355_unet* net, int B, void* workspace, int64_t workspace_bytes, Scratch& sc) {
  MI355_REQUIRE(net && workspace, -1, "null argument");
  MI355_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, -1, "workspace must be 256-byte aligned");
  MI355_REQUIRE(workspace_bytes >= mi355_unet_workspace_bytes(net, B), -2, "workspace too small");
  const size_t hw = (size_t)net->cfg.image_size * net->cfg.image_size;
  char* p = reinterpret_cast<char*>(workspace);
  sc.t = reinterpret_cast<float*>(p); p += al256((size_t)B * 4);
  sc.v = reinterpret_cast<float*>(p); p += al256((size_t)B * 32 * hw * 4);
  sc.none = reinterpret_cast<float*>(p); p += al256((size_t)B * 32 * hw * 4);
  sc.tsteps = reinterpret_cast<float*>(p); p += al256((size_t)EMB_TABLE_STEPS * 4);
  sc.embtab = reinterpret_cast<float*>(p); p += al256((size_t)EMB_TABLE_STEPS * ((size_t)net->emb_total + 9 * (size_t)net->cfg.model_channels) * 4);
  sc.xstate = reinterpret_cast<float*>(p); p += al256((size_t)B * net->cfg.out_channels * hw * 4);
  sc.unet_ws = p;
  sc.unet_bytes = workspace_bytes - (p - reinterpret_cast<char*>(workspace));
  return 0;
}

// The loop proper: every launch of every step on stream s (the caller's stream, or the handle's capture stream while a graph is recorded).
static int cfm_euler_loop(mi355_unet* net, const Scratch& sc, float* x, int x_channels, const float* cond, int cond_channels, float* cdrift,
                          const float* t_span_host, int n_t, const float* emb_table, float* traj, int batch, int64_t n, int64_t nc, hipStream_t s) {
  UnetRun run = uniform_t_run();
  for (int k = 0; k + 1 < n_t; ++k) {
    const float t = t_span_host[k], dt = t_span_host[k + 1] - t_span_host[k];
    int rc;
    if (emb_table) run.emb_row = emb_table + (size_t)k * net->emb_total;
    else if ((rc = fill_launch(sc.t, t, batch, s))) return rc;
    run.euler_x = x; run.euler_dt = dt;   // x += dt * v: in the last conv's epilogue, or as a launch of unet_forward's own behind it
    if ((rc = unet_forward(net, x, x_channels, cond, cond_channels, sc.t, sc.v, batch, sc.unet_ws, sc.unet_bytes, s, run))) return rc;
    if (cdrift && (rc = euler_step_launch(cdrift, cdrift, dt, nc, s))) return rc;
    if (traj) MI355_CHECK_HIP(hipMemcpyAsync(traj + (size_t)(k + 1) * n, x, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// knobs.sampler_graph: the loop as ONE graph launch.  The graph works on the workspace's resident state (sc.xstate) so that it does not depend on
// the caller's x / u8 pointers (a fresh tensor per call is the normal use); it does depend on the workspace, the batch, the schedule (dt is a kernel
// argument, the embedding rows are table addresses) and the condition pointer: those are its key.  The embedding table is rebuilt on every call,
// outside the graph (the workspace is the caller's: another sampler may have used it in between).
static int cfm_euler_graph(mi355_unet* net, const Scratch& sc, float* x, int x_channels, const float* cond, int cond_channels, const float* t_span_host,
                           int n_t, const float* emb_table, int batch, int64_t n, void* workspace, hipStream_t s) {
  uint64_t h = 1469598103934665603ull;
  for (int k = 0; k < n_t; ++k) { uint32_t b; std::memcpy(&b, &t_span_host[k], 4); h = (h ^ b) * 1099511628211ull; }
  const uint64_t key[8] = {(uint64_t)reinterpret_cast<uintptr_t>(workspace), (uint64_t)batch, (uint64_t)n_t, h, (uint64_t)reinterpret_cast<uintptr_t>(cond),
                           (uint64_t)cond_channels, (uint64_t)x_channels, (uint64_t)reinterpret_cast<uintptr_t>(emb_table)};
  std::lock_guard<std::mutex> lock(net->graph_mu);
  mi355_unet::SamplerGraph* g = nullptr;
  for (auto& e : net->graphs) if (e.exec && std::memcmp(e.key, key, sizeof(key)) == 0) { g = &e; break; }
  if (!g) {
    if (!net->capture_stream) MI355_CHECK_HIP(hipStreamCreateWithFlags(&net->capture_stream, hipStreamNonBlocking));
    hipStream_t cs = net->capture_stream;
    MI355_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    int rc = cfm_euler_loop(net, sc, sc.xstate, x_channels, cond, cond_channels, nullptr, t_span_host, n_t, emb_table, nullptr, batch, n, 0, cs);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) { mi355_set_error(std::string("cfm_euler_sample: graph capture failed: ") + hipGetErrorString(e)); return -3; }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { mi355_set_error(std::string("cfm_euler_sample: graph instantiation failed: ") + hipGetErrorString(e)); return -3; }
    constexpr size_t CAP = 4;
    if (net->graphs.size() < CAP) { net->graphs.emplace_back(); g = &net->graphs.back(); }
    else {
      g = &net->graphs[0];
      for (auto& c : net->graphs) if (c.stamp < g->stamp) g = &c;
      // the replaced graph may still be running on a stream of the caller's: wait for the device before its nodes are freed (rare: a fifth key)
      (void)hipDeviceSynchronize();
      (void)hipGraphExecDestroy(g->exec);
    }
    std::memcpy(g->key, key, sizeof(key));
    g->exec = exec;
    g->launches = net->last_launches;
  }
  g->stamp = ++net->graph_clock;
  MI355_CHECK_HIP(hipMemcpyAsync(sc.xstate, x, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  MI355_CHECK_HIP(hipGraphLaunch(g->exec, s));
  MI355_CHECK_HIP(hipMemcpyAsync(x, sc.xstate, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  net->last_launches = g->launches;
  return 0;
}

int mi355_cfm_euler_sample(mi355_unet* net, float* x, int x_channels, const float* cond, int cond_channels, int cond_drift,
                           const float* t_span_host, int n_t, float* traj, uint8_t* u8_out, int batch, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  MI355_REQUIRE(net && x && t_span_host && n_t >= 1, -1, "cfm_euler_sample: bad argument");
  MI355_REQUIRE(x_channels == net->cfg.out_channels, -2, "cfm_euler_sample: the vector field must have the state's channel count");
  Scratch sc;
  if (int rc = carve(net, batch, workspace, workspace_bytes, sc)) return rc;
  hipStream_t s = S(stream);
  const float* emb_table = nullptr;
  if (int rc = make_emb_table(net, sc, t_span_host, n_t - 1, s, &emb_table)) return rc;
  const int64_t n = (int64_t)batch * x_channels * net->cfg.image_size * net->cfg.image_size;
  const int64_t nc = (int64_t)batch * cond_channels * net->cfg.image_size * net->cfg.image_size;
  if (traj) MI355_CHECK_HIP(hipMemcpyAsync(traj, x, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
  // cond_drift: the reference integrates the CONCATENATED state [x, con] whose second half has derivative con itself
  // (mnist/utils_mnist2.py:120-124), so under Euler the condition the model sees is con_{k+1} = con_k + dt * con_k.
  // The drifting copy lives in the sampler scratch (the caller's tensor is not modified).
  float* cdrift = nullptr;
  if (cond && cond_drift) {
    MI355_REQUIRE(cond_channels <= 32, -2, "cfm_euler_sample: condition has more than 32 channels");
    cdrift = sc.none;
    MI355_CHECK_HIP(hipMemcpyAsync(cdrift, cond, (size_t)nc * 4, hipMemcpyDeviceToDevice, s));
    cond = cdrift;
  }
  int rc;
  if (net->knobs.sampler_graph && emb_table && !traj && !cdrift && n_t > 1)
    rc = cfm_euler_graph(net, sc, x, x_channels, cond, cond_channels, t_span_host, n_t, emb_table, batch, n, workspace, s);
  else
    rc = cfm_euler_loop(net, sc, x, x_channels, cond, cond_channels, cdrift, t_span_host, n_t, emb_table, traj, batch, n, nc, s);
  if (rc) return rc;
  if (u8_out) return quantize_u8_launch(x, u8_out, n, s);
  return 0;
}

int mi355_ddpm_sample(mi355_unet* net, float* x, int channels, const float* cond, const mi355_ddpm_tables* tb,
                      const mi355_ddpm_options* opt, const float* noise, int64_t n_noise_draws, int batch, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  MI355_REQUIRE(net && x && tb && opt, -1, "ddpm_sample: null argument");
  MI355_REQUIRE(channels == net->cfg.out_channels, -2, "ddpm_sample: eps model must output the state's channel count");
  const int mode = opt->mode, Ns = tb->Ns;
  const bool amortized = net->cfg.in_channels == 2 * channels;
  MI355_REQUIRE(amortized || net->cfg.in_channels == channels, -2, "ddpm_sample: network in_channels must be C or 2C");
  MI355_REQUIRE(mode != MI355_DDPM_REPLACEMENT || (!amortized && cond), -2, "ddpm_sample: replacement needs an unconditional net and a condition");
  MI355_REQUIRE(mode != MI355_DDPM_AMORTIZED || (amortized && cond), -2, "ddpm_sample: amortized needs a 2C-input net and a condition");
  Scratch sc;
  if (int rc = carve(net, batch, workspace, workspace_bytes, sc)) return rc;
  UnetRun run = uniform_t_run();
  hipStream_t s = S(stream);
  const int64_t n = (int64_t)batch * channels * net->cfg.image_size * net->cfg.image_size;
  const int64_t n_al = (n + 3) / 4 * 4;
  int rc;
  const float* emb_table = nullptr;   // row i = step i (time i / Ns)
  if (Ns <= EMB_TABLE_STEPS) {
    std::vector<float> th((size_t)Ns);
    for (int i = 0; i < Ns; ++i) th[i] = (float)i / (float)Ns;
    if ((rc = make_emb_table(net, sc, th.data(), Ns, s, &emb_table))) return rc;
    MI355_CHECK_HIP(hipStreamSynchronize(s));   // `th` is a temporary host buffer (once per sample() call)
  }
  if (amortized && (rc = fill_launch(sc.none, opt->none_value, n, s))) return rc;
  // the net's condition input on predictor steps / on corrector steps (the reference's corrector calls
  // x0_model without the condition, sampling.py:116 -> none_like)
  const float* cond_pred = !amortized ? nullptr : ((mode == MI355_DDPM_AMORTIZED || (mode == MI355_DDIM && cond)) ? cond : sc.none);
  const float* cond_corr = amortized ? sc.none : nullptr;
  int64_t draw = 0;
  auto next_noise = [&](const float*& zptr, int& philox, uint64_t& off) -> int {
    if (noise) {
      MI355_REQUIRE(draw < n_noise_draws, -2, "ddpm_sample: injected noise exhausted");
      zptr = noise + (size_t)draw * n; philox = 0; off = 0;
    } else { zptr = nullptr; philox = 1; off = (uint64_t)draw * (uint64_t)n_al; }
    ++draw;
    return 0;
  };
  for (int i = Ns - 1; i >= 0; --i) {
    const float tval = (float)i / (float)Ns;  // eps_model(xi, i) = network(xi, 1.0*i/Ns)  loss_functions.py:18-19
    if (mode == MI355_DDPM_REPLACEMENT && i < (int)(Ns * opt->start_fraction)) {
      const float* z = nullptr; int ph = 0; uint64_t off = 0;
      if (opt->noise_condition && (rc = next_noise(z, ph, off))) return rc;
      if ((rc = replace_mask_launch(x, cond, z, opt->pad_value, opt->noise_condition, tb->sqrt_alphas_cumprod[i],
                                    tb->sqrt_one_minus_alphas_cumprod[i], ph, opt->seed, off, n, s))) return rc;
    }
    if (emb_table) run.emb_row = emb_table + (size_t)i * net->emb_total;
    else if ((rc = fill_launch(sc.t, tval, batch, s))) return rc;
    if ((rc = unet_forward(net, x, channels, cond_pred, channels, sc.t, sc.v, batch, sc.unet_ws, sc.unet_bytes, s, run))) return rc;
    if (mode == MI355_DDIM) {
      if ((rc = ddim_step_launch(x, sc.v, tb->sqrt_recip_alphas_cumprod[i], tb->sqrt_recipm1_alphas_cumprod[i],
                                 tb->alphas_cumprod_prev[i], n, s))) return rc;
      continue;
    }
    const float* z = nullptr; int ph = 0; uint64_t off = 0;
    if (i > 0 && (rc = next_noise(z, ph, off))) return rc;
    const float sigma = expf(0.5f * tb->posterior_log_variance_clipped[i]);
    if ((rc = ddpm_step_launch(x, sc.v, z, tb->sqrt_recip_alphas_cumprod[i], tb->sqrt_recipm1_alphas_cumprod[i],
                               tb->posterior_mean_coef1[i], tb->posterior_mean_coef2[i], sigma, ph, opt->seed, off, n, s))) return rc;
    for (int c = 0; c < opt->n_corrector; ++c) {
      if ((rc = unet_forward(net, x, channels, cond_corr, channels, sc.t, sc.v, batch, sc.unet_ws, sc.unet_bytes, s, run))) return rc;
      const float* z2 = nullptr; int ph2 = 0; uint64_t off2 = 0;
      if ((rc = next_noise(z2, ph2, off2))) return rc;
      const float dt = (opt->tmax - opt->tmin) / (float)Ns;
      if ((rc = corrector_step_launch(x, sc.v, z2, tb->sqrt_recip_alphas_cumprod[i], tb->sqrt_recipm1_alphas_cumprod[i],
                                      tb->recip_sqrt_m1_alphas_cumprod[i], dt, opt->delta, ph2, opt->seed, off2, n, s))) return rc;
    }
  }
  return clip_launch(x, -1.f, 1.f, n, s);
}

// ---- single ops -------------------------------------------------------------------------------------

int mi355_timestep_embedding(const float* t, int batch, int dim, float max_period, float* out, void* stream) {
  MI355_REQUIRE(t && out && batch > 0 && dim > 0, -1, "timestep_embedding: bad argument");
  return timestep_embedding_launch(t, batch, dim, max_period, out, S(stream));
}
int mi355_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int batch, int channels, int hw, int groups,
                    float eps, int silu, void* stream) {
  MI355_REQUIRE(x && gamma && beta && y, -1, "groupnorm: null argument");
  return groupnorm_nchw_launch(x, gamma, beta, y, batch, channels, hw, groups, eps, silu, S(stream));
}
int mi355_euler_step(float* x, const float* v, float dt, int64_t n, void* stream) { return euler_step_launch(x, v, dt, n, S(stream)); }
int mi355_ddpm_step(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float coef1, float coef2,
                    float sigma, int use_philox, uint64_t seed, uint64_t offset, int64_t n, void* stream) {
  return ddpm_step_launch(x, eps, z, c_recip, c_recipm1, coef1, coef2, sigma, use_philox, seed, offset, n, S(stream));
}
int mi355_corrector_step(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float recip_sqrt_m1, float dt,
                         float delta, int use_philox, uint64_t seed, uint64_t offset, int64_t n, void* stream) {
  return corrector_step_launch(x, eps, z, c_recip, c_recipm1, recip_sqrt_m1, dt, delta, use_philox, seed, offset, n, S(stream));
}
int mi355_ddim_step(float* x, const float* eps, float c_recip, float c_recipm1, float acp_prev, int64_t n, void* stream) {
  return ddim_step_launch(x, eps, c_recip, c_recipm1, acp_prev, n, S(stream));
}
int mi355_replace_mask(float* x, const float* cond, const float* z, float pad_value, int noisy, float sa, float sb, int use_philox,
                       uint64_t seed, uint64_t offset, int64_t n, void* stream) {
  return replace_mask_launch(x, cond, z, pad_value, noisy, sa, sb, use_philox, seed, offset, n, S(stream));
}
int mi355_clip(float* x, float lo, float hi, int64_t n, void* stream) { return clip_launch(x, lo, hi, n, S(stream)); }
int mi355_guidance_seed(const float* x, const float* eps, const float* cond, float c_recip, float c_recipm1, int mode, float pad_value,
                        int64_t elems_per_sample, float* g_eps, float* g_x, int64_t n, void* stream) {
  return guidance_seed_launch(x, eps, cond, c_recip, c_recipm1, mode, pad_value, elems_per_sample, g_eps, g_x, n, S(stream));
}
int mi355_guidance_update(float* x, const float* g_x, const float* vjp, float scale, int apply, float* update, int64_t n, void* stream) {
  return guidance_update_launch(x, g_x, vjp, scale, apply, update, n, S(stream));
}
int mi355_ema_update(float* target, const float* source, float decay, float one_minus_decay, int64_t n, void* stream) {
  MI355_REQUIRE(target && source, -1, "ema_update: null argument");
  return ema_update_launch(target, source, decay, one_minus_decay, n, S(stream));
}
int mi355_mse_per_sample(const float* a, const float* b, float* out, int batch, int64_t elems_per_sample, void* stream) {
  MI355_REQUIRE(a && b && out, -1, "mse_per_sample: null argument");
  return mse_per_sample_launch(a, b, out, batch, elems_per_sample, S(stream));
}
int mi355_lincomb_per_sample(float* out, const float* x, const float* y, const float* a, const float* b, int batch,
                             int64_t elems_per_sample, void* stream) {
  return lincomb_per_sample_launch(out, x, y, a, b, batch, elems_per_sample, S(stream));
}
int mi355_resize_bilinear(const float* in, float* out, int64_t planes, int h_in, int w_in, int h_out, int w_out, void* stream) {
  return resize_bilinear_launch(in, out, planes, h_in, w_in, h_out, w_out, S(stream));
}
int mi355_paint_patch(const float* images, const int32_t* top, const int32_t* left, int patch_size, float pad_value, int outpaint, float* out,
                      int batch, int channels, int h, int w, void* stream) {
  return paint_patch_launch(images, top, left, patch_size, pad_value, outpaint, out, batch, channels, h, w, S(stream));
}
int mi355_quantize_u8(const float* x, uint8_t* out, int64_t n, void* stream) { return quantize_u8_launch(x, out, n, S(stream)); }
int mi355_to_unit_range(const float* x, float* out, int64_t n, void* stream) { return to_unit_range_launch(x, out, n, S(stream)); }
int mi355_randn(float* out, uint64_t seed, uint64_t offset, int64_t n, void* stream) { return randn_launch(out, seed, offset, n, S(stream)); }

int mi355_rk_combine(float* out, const float* y0, const float* k0, const float* k1, const float* k2, const float* k3, const float* k4,
                     const float* k5, const float* k6, const float* coeff_host, int nk, int64_t n, void* stream) {
  MI355_REQUIRE(coeff_host || nk == 0, -1, "rk_combine: null coefficients");
  const float* k[7] = {k0, k1, k2, k3, k4, k5, k6};
  return rk_combine_launch(out, y0, k, coeff_host, nk, n, S(stream));
}
int mi355_rk_sqnorm(const float* a, const float* sub, const float* b, const float* b2, float atol, float rtol, int64_t n, double* out,
                    void* stream) {
  return rk_sqnorm_launch(a, sub, b, b2, atol, rtol, n, out, S(stream));
}
int mi355_rk_interp(float* out, const float* y0, const float* y1, const float* y_mid, const float* f0, const float* f1, float dt, float x,
                    int64_t n, void* stream) {
  MI355_REQUIRE(out && y0 && y1 && y_mid && f0 && f1, -1, "rk_interp: null argument");
  return rk_interp_launch(out, y0, y1, y_mid, f0, f1, dt, x, n, S(stream));
}

int64_t mi355_box_probe_workspace_bytes(void) { return box_probe_workspace_bytes(); }
int mi355_box_probe(int reps, void* workspace, int64_t workspace_bytes, void* stream, float* us_per_launch, float* clock_mhz, float* tflop) {
  if (tflop) *tflop = (float)(box_probe_flops() * 1e-12);
  return box_probe_run(reps, workspace, workspace_bytes, S(stream), us_per_launch, clock_mhz);
}
int64_t mi355_box_probe_hbm_workspace_bytes(void) { return box_probe_hbm_workspace_bytes(); }
int mi355_box_probe_hbm(int reps, void* workspace, int64_t workspace_bytes, void* stream, float* us_per_launch, float* gbytes) {
  if (gbytes) *gbytes = (float)(box_probe_hbm_bytes() * 1e-9);
  return box_probe_hbm_run(reps, workspace, workspace_bytes, S(stream), us_per_launch);
}

int64_t mi355_op_workspace_bytes(int batch, int max_channels, int hw) {
  const size_t c = (size_t)max_channels + 32;
  return (int64_t)(2 * al256((size_t)batch * hw * 4 * c * 4) + al256(c * c * 9 * 4 * 2) + 4 * al256((size_t)batch * c * 4) + (1 << 20));
}

namespace {
// device scratch of mi355_conv2d_ex's extras (a test op: plain hipMalloc, freed when the call returns)
struct ExScratch {
  std::vector<void*> ptrs;
  ~ExScratch() { for (void* q : ptrs) (void)hipFree(q); }
  void* get(size_t bytes) { void* q = nullptr; if (hipMalloc(&q, bytes ? bytes : 256) != hipSuccess) return nullptr; ptrs.push_back(q); return q; }
};
}  // namespace

static int conv2d_impl(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                       int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                       const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                       int64_t workspace_bytes, void* stream, mi355_conv_extras* ex);

int mi355_conv2d(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                 int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                 const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                 int64_t workspace_bytes, void* stream) {
  return conv2d_impl(x, x1, cin1, w_host, bias_host, y, batch, cin, h, w, cout, ksize, stride, resample, gn_gamma, gn_beta, gn_silu, emb, res, res_mode,
                     dtype, debug, workspace, workspace_bytes, stream, nullptr);
}
int mi355_conv2d_ex(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                    int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                    const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                    int64_t workspace_bytes, void* stream, mi355_conv_extras* extras) {
  MI355_REQUIRE(extras, -1, "conv2d_ex: null extras");
  return conv2d_impl(x, x1, cin1, w_host, bias_host, y, batch, cin, h, w, cout, ksize, stride, resample, gn_gamma, gn_beta, gn_silu, emb, res, res_mode,
                     dtype, debug, workspace, workspace_bytes, stream, extras);
}

static int conv2d_impl(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                       int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                       const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                       int64_t workspace_bytes, void* stream, mi355_conv_extras* ex) {
  const mi355_debug_config& K = debug ? *debug : mi355_default_debug();
  MI355_REQUIRE(x && w_host && y && workspace, -1, "conv2d: null argument");
  MI355_REQUIRE(dtype == MI355_F32 || dtype == MI355_BF16 || dtype == MI355_BF16X2 || dtype == MI355_F16, -1, "conv2d: bad dtype");
  const int wsplit = dtype == MI355_BF16X2 ? 1 : 0;   // bf16 storage, weights as hi | lo bf16 halves along K
  dtype = dtype == MI355_F16 ? DT_F16 : (wsplit ? DT_BF16 : dtype);   // the internal element-type code from here on (ops.h)
  MI355_REQUIRE(stride == 1 || stride == 2, -1, "conv2d: stride must be 1 or 2");
  MI355_REQUIRE(!(stride == 2 && resample), -1, "conv2d: stride 2 cannot be combined with resampling");
  MI355_REQUIRE((x1 != nullptr) == (cin1 > 0), -1, "conv2d: x1 and cin1 go together");
  MI355_REQUIRE(res == nullptr || res_mode == RES_SAME || res_mode == RES_UP2, -1, "conv2d: res_mode must be 1 (same size) or 2 (nearest x2)");
  hipStream_t s = S(stream);
  const int CH = dtype == 0 ? 16 : 32, esz = dtype == 0 ? 4 : 2;
  MI355_REQUIRE(!x1 || (cin % CH == 0 && cin1 % CH == 0), -2, "conv2d: a two-source conv needs both channel counts to be multiples of the 64-byte chunk");
  const int cpad = (cin + CH - 1) / CH * CH;
  const int ctot = cin + cin1, ctot_pad = cpad + cin1;
  ConvDesc d; d.dtype = dtype; d.N = batch; d.Hs = h; d.Ws = w; d.C0 = cpad; d.C1 = cin1; d.ks = ksize; d.Cout = cout;
  d.knobs = &K; d.wsplit = wsplit;
  if (!x1 && cin <= 8) d.cin_real = cin;   // the padding channels pack_nhwc adds are zero: conv3x3_in_kernel contracts over the first slot only
  const bool pool = resample == 3;   // 2x2 average pool of the (normalised) input: a pre-pass, then a plain conv
  MI355_REQUIRE(!(pool && x1), -4, "conv2d: pooling over a channel concat is not supported");
  if (pool) { d.Hs = h / 2; d.Ws = w / 2; }
  d.mode = stride == 2 ? CONV_STRIDE2 : (resample == 2 ? CONV_UP2 : CONV_UNIT);
  const ConvGeom g = conv_geometry(d);
  const bool nhwc = cout % 32 == 0;
  MI355_REQUIRE(nhwc || (!emb && !res), -4, "conv2d: emb / residual epilogues need an NHWC output (cout % 32 == 0)");
  const int Hr = res_mode == RES_UP2 ? g.Ho / 2 : g.Ho, Wr = res_mode == RES_UP2 ? g.Wo / 2 : g.Wo;
  char* p = reinterpret_cast<char*>(workspace);
  char* end = p + workspace_bytes;
  void* xin = p; p += al256((size_t)batch * h * w * cpad * esz);
  void* xin1 = p; if (x1) p += al256((size_t)batch * h * w * cin1 * esz);
  void* wdev = p; const size_t wbytes = conv_packed_weight_bytes(dtype, cout, ctot, ksize, wsplit); p += al256(wbytes);
  float* bdev = reinterpret_cast<float*>(p); p += al256((size_t)cout * 4);
  float* ga = reinterpret_cast<float*>(p); p += al256((size_t)batch * ctot_pad * 4);
  float* gb = reinterpret_cast<float*>(p); p += al256((size_t)batch * ctot_pad * 4);
  void* yout = p; p += al256((size_t)batch * g.Ho * g.Wo * cout * esz);
  void* rin = p; if (res) p += al256((size_t)batch * Hr * Wr * cout * esz);
  void* xpool = p; if (pool) p += al256((size_t)batch * (h / 2) * (w / 2) * cpad * esz);
  uint32_t* errw = reinterpret_cast<uint32_t*>(p); p += 256;
  MI355_REQUIRE(p <= end, -2, "conv2d: workspace too small");
  MI355_CHECK_HIP(hipMemsetAsync(errw, 0, 256, s));
  d.err = errw;
  int rc;
  if ((rc = pack_nhwc_launch(dtype, x, cin, nullptr, 0, batch, h * w, cpad, xin, s))) return rc;
  if (x1 && (rc = pack_nhwc_launch(dtype, x1, cin1, nullptr, 0, batch, h * w, cin1, xin1, s))) return rc;
  if (res && (rc = pack_nhwc_launch(dtype, res, cout, nullptr, 0, batch, Hr * Wr, cout, rin, s))) return rc;
  std::vector<char> packed(wbytes);
  conv_pack_weights(dtype, w_host, cout, ctot, ksize, packed.data(), wsplit);
  MI355_CHECK_HIP(hipMemcpyAsync(wdev, packed.data(), wbytes, hipMemcpyHostToDevice, s));
  if (bias_host) MI355_CHECK_HIP(hipMemcpyAsync(bdev, bias_host, (size_t)cout * 4, hipMemcpyHostToDevice, s));
  if (gn_gamma) {
    MI355_REQUIRE(ctot % 32 == 0 && cin % 32 == 0, -2, "conv2d: the GroupNorm32 prologue needs channels % 32 == 0");
    GnDesc gd; gd.dtype = dtype; gd.src0 = xin; gd.C0 = cpad; gd.src1 = x1 ? xin1 : nullptr; gd.C1 = cin1; gd.N = batch; gd.HW = h * w;
    gd.gamma = gn_gamma; gd.beta = gn_beta;
    gd.a = ga; gd.b = gb;
    if ((rc = gn_affine_launch(gd, s))) return rc;
    if (!pool) { d.pro_a = ga; d.pro_b = gb; d.pro_silu = gn_silu; }
  }
  if (x1) d.src1 = xin1;
  if (emb) { d.emb = emb; d.emb_stride = cout; }
  if (res) { d.res = rin; d.res_mode = res_mode; }
  d.src0 = xin;
  if (pool) {
    if ((rc = affine_pool_launch(dtype, xin, gn_gamma ? ga : nullptr, gn_gamma ? gb : nullptr, gn_silu, xpool, batch, h, w, cpad, s))) return rc;
    d.src0 = xpool;
  } d.w = wdev; d.bias = bias_host ? bdev : nullptr;
  d.out_mode = nhwc ? OUT_NHWC : OUT_NCHW_F32;
  d.out = nhwc ? yout : (void*)y;
#ifdef CONV_STAMPS
  const size_t nwaves = (size_t)g.grid_m * g.grid_n * 4;
  MI355_REQUIRE(p + 65536 + nwaves * 64 <= end, -2, "conv2d: workspace too small (stamps)");
  MI355_CHECK_HIP(hipMemsetAsync(p, 0, 65536 + nwaves * 64, s));
  unsigned long long* clk = reinterpret_cast<unsigned long long*>(p);   // [4096 waves][shader cycles, 100-MHz ticks]: CLK_FLUSH
  p += 65536;
  d.dbg = p;
#endif
  // ---- extras (mi355_conv2d_ex): the small-level kernel's fused forms, as the engine's walker asks for them ----
  ExScratch xs;
  void* act_dev[2] = {nullptr, nullptr};
  std::vector<char> packed_f; std::vector<float> bias_f;
  int act_done = 0;
  if (ex) {
    ex->act_done = 0; ex->skip_done = 0;
    MI355_REQUIRE(nhwc && !pool, -4, "conv2d_ex: extras need an NHWC output and no pooling");
    if (ex->skip_x0) {
      MI355_REQUIRE(ksize == 3 && stride == 1 && !resample && !x1 && !res && !wsplit && ex->skip_w_host && ex->skip_c0 > 0, -1, "conv2d_ex: the fused skip conv goes with a plain 3x3 conv of one source");
      MI355_REQUIRE((ex->skip_x1 != nullptr) == (ex->skip_c1 > 0), -1, "conv2d_ex: skip_x1 and skip_c1 go together");
      const int cs = ex->skip_c0 + ex->skip_c1;
      void* k0 = xs.get((size_t)batch * h * w * ex->skip_c0 * esz);
      void* k1 = ex->skip_x1 ? xs.get((size_t)batch * h * w * ex->skip_c1 * esz) : nullptr;
      const size_t fb = conv_packed_weight_bytes_skip(dtype, cout, ctot, cs);
      void* wf = xs.get(fb);
      MI355_REQUIRE(k0 && wf && (!ex->skip_x1 || k1), -2, "conv2d_ex: out of device memory");
      if ((rc = pack_nhwc_launch(dtype, ex->skip_x0, ex->skip_c0, nullptr, 0, batch, h * w, ex->skip_c0, k0, s))) return rc;
      if (k1 && (rc = pack_nhwc_launch(dtype, ex->skip_x1, ex->skip_c1, nullptr, 0, batch, h * w, ex->skip_c1, k1, s))) return rc;
      packed_f.resize(fb);
      conv_pack_weights_skip(dtype, w_host, ex->skip_w_host, cout, ctot, cs, packed_f.data());
      MI355_CHECK_HIP(hipMemcpyAsync(wf, packed_f.data(), fb, hipMemcpyHostToDevice, s));
      bias_f.assign((size_t)cout, 0.f);
      for (int i = 0; i < cout; ++i) bias_f[i] = (bias_host ? bias_host[i] : 0.f) + (ex->skip_bias_host ? ex->skip_bias_host[i] : 0.f);
      MI355_CHECK_HIP(hipMemcpyAsync(bdev, bias_f.data(), (size_t)cout * 4, hipMemcpyHostToDevice, s));
      d.w = wf; d.bias = bdev;
      d.skip_src0 = k0; d.skip_C0 = ex->skip_c0; d.skip_src1 = k1; d.skip_C1 = ex->skip_c1;
      if (conv_fused_skip_ok(d) != 0) { mi355_set_error("conv2d_ex: this launch cannot carry the fused skip conv (shape, batch or knobs)"); return MI355_ERR_UNSUPPORTED; }
      ex->skip_done = 1;
    }
    for (int k = 0; k < 2; ++k) {
      if (!ex->act_out[k]) continue;
      MI355_REQUIRE(k == 0 || ex->act_out[0], -1, "conv2d_ex: site 1 without site 0");
      MI355_REQUIRE(ex->act_gamma[k] && ex->act_beta[k] && ex->act_ctotal[k] % 32 == 0 && ex->act_coff[k] >= 0 && ex->act_coff[k] + cout <= ex->act_ctotal[k], -1, "conv2d_ex: bad GroupNorm site");
      const size_t ab = (size_t)batch * g.Ho * g.Wo * ex->act_ctotal[k] * esz;
      act_dev[k] = xs.get(ab);
      MI355_REQUIRE(act_dev[k], -2, "conv2d_ex: out of device memory");
      MI355_CHECK_HIP(hipMemsetAsync(act_dev[k], 0, ab, s));
    }
    if (act_dev[0]) {
      d.act_out = act_dev[0]; d.act_gamma = ex->act_gamma[0] + ex->act_coff[0]; d.act_beta = ex->act_beta[0] + ex->act_coff[0];
      d.act_silu = ex->act_silu[0]; d.act_stride = ex->act_ctotal[0]; d.act_coff = ex->act_coff[0]; d.act_cpg = ex->act_ctotal[0] / 32; d.act_raw = 1;
      if (ex->act_film) { d.act_film = ex->act_film; d.act_film_stride = 2 * cout; }
    }
    if (act_dev[1]) {
      d.act2_out = act_dev[1]; d.act2_gamma = ex->act_gamma[1] + ex->act_coff[1]; d.act2_beta = ex->act_beta[1] + ex->act_coff[1];
      d.act2_silu = ex->act_silu[1]; d.act2_stride = ex->act_ctotal[1]; d.act2_coff = ex->act_coff[1]; d.act2_cpg = ex->act_ctotal[1] / 32;
    }
  }
  if ((rc = conv_launch(d, s, nullptr, ex ? &act_done : nullptr))) return rc;
  if (ex) {
    ex->act_done = act_done;
    for (int k = 0; k < 2; ++k)
      if (act_dev[k] && (act_done & (1 << k)) && (rc = unpack_nchw_launch(dtype, act_dev[k], batch, g.Ho * g.Wo, ex->act_ctotal[k], ex->act_out[k], s))) return rc;
  }
  if (K.conv_time_reps > 0) {   // diagnostic: average duration of the conv launch alone
    const int reps = K.conv_time_reps;
    hipEvent_t e0, e1;
    MI355_CHECK_HIP(hipEventCreate(&e0)); MI355_CHECK_HIP(hipEventCreate(&e1));
    MI355_CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) if ((rc = conv_launch(d, s))) return rc;
    MI355_CHECK_HIP(hipEventRecord(e1, s));
    MI355_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MI355_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, fl = 2.0 * batch * g.Ho * g.Wo * (double)cout * ctot * ksize * ksize;
    fprintf(stderr, "[conv time] %d launches, %.1f us each, %.0f TFLOP/s\n", reps, us, fl / us * 1e-6);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  if (nhwc && (rc = unpack_nchw_launch(dtype, yout, batch, g.Ho * g.Wo, cout, y, s))) return rc;
  MI355_CHECK_HIP(hipStreamSynchronize(s));  // `packed` is a temporary host buffer
  {
    uint32_t ev = 0;
    MI355_CHECK_HIP(hipMemcpy(&ev, errw, 4, hipMemcpyDeviceToHost));
    if (ev) { mi355_set_error("conv2d: the persistent kernel gave up a bounded counter wait (hand-over stalled): the output is invalid"); return MI355_ERR_TIMEOUT; }
  }
#ifdef CONV_STAMPS
  {
    std::vector<unsigned long long> hv(nwaves * 8);
    MI355_CHECK_HIP(hipMemcpy(hv.data(), p, nwaves * 64, hipMemcpyDeviceToHost));
    static const char* names_plain[8] = {"setup", "commit_patch", "barrier_A", "commit_w(+vmcnt)", "barrier_B", "prefetch_issue", "mma", "epilogue"};
    // warp-specialised kernel: slots 0-4 are written by loader waves only, 5-7 by consumer waves only (half the waves each)
    static const char* names_ws[8] = {"L:fill", "L:commit_w", "L:commit_frag", "L:issue", "L:barrier", "C:barrier", "C:mma_row", "C:epilogue+setup"};
    const bool ws_names = K.conv_ws != 0;
    const char** names = ws_names ? names_ws : names_plain;
    double h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
    for (size_t w = 0; w < nwaves; ++w) for (int k = 0; k < 8; ++k) h[k] += (double)hv[w * 8 + k];
    for (int k = 0; k < 8; ++k) tot += h[k];
    {   // in-kernel clock of the LAST launch (after conv_time_reps back-to-back launches: the sustained clock), median over waves
      std::vector<unsigned long long> hc(8192);
      MI355_CHECK_HIP(hipMemcpy(hc.data(), clk, 65536, hipMemcpyDeviceToHost));
      std::vector<double> mhz;
      for (size_t w = 0; w < 4096; ++w) if (hc[2 * w + 1] > 0) mhz.push_back(100.0 * (double)hc[2 * w] / (double)hc[2 * w + 1]);
      if (!mhz.empty()) {
        std::sort(mhz.begin(), mhz.end());
        fprintf(stderr, "[conv clock] %zu waves: in-kernel clock min %.0f median %.0f max %.0f MHz\n", mhz.size(), mhz.front(), mhz[mhz.size() / 2], mhz.back());
      }
    }
    if (g.BM == 256 && g.BN == 256 && nwaves * 8 >= 2048 * 8 + 160) {   // timeline of workgroup 0, taps 8 .. 11 (PP_TRACE): cycles relative to wave 0's first stamp
      const unsigned long long* tr = hv.data() + 2048 * 8;
      const unsigned long long t0 = tr[0];
      static const char* ev[5] = {"issued", "vmcnt", "barL", "mfma", "barM"};
      for (int w = 0; w < 8; ++w) {
        fprintf(stderr, "[pp trace] wave %d:", w);
        for (int i = 0; i < 20; ++i) fprintf(stderr, " %s%d=%lld", ev[i % 5], 8 + i / 5, (long long)(int32_t)(uint32_t)(tr[w * 20 + i] - t0));
        fprintf(stderr, "\n");
      }
    }
    if (g.BM == 256 && g.BN == 256) {   // ping-pong kernel (conv_pp.inc.h): 8 waves per workgroup, waves 0-3 = group 0, 4-7 = group 1 (one tick behind)
      static const char* names_pp[8] = {"L:reads+dma", "L:vmcnt", "L:barrier", "M:mfma", "M:barrier", "E:epilogue", "E:rejoin", "tile_head"};
      for (int grp = 0; grp < 2; ++grp) {
        double hp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0; size_t nw = 0;
        for (size_t w = 0; w < nwaves; ++w) {
          if (((w >> 2) & 1) != (size_t)grp) continue;
          double tw = 0;
          for (int k = 0; k < 8; ++k) tw += (double)hv[w * 8 + k];
          if (tw == 0) continue;
          ++nw;
          for (int k = 0; k < 8; ++k) hp[k] += (double)hv[w * 8 + k];
          tp += tw;
        }
        if (!nw) continue;
        fprintf(stderr, "[conv stamps] ping-pong group %d, waves %zu, cycles/wave %.0f:", grp, nw, tp / nw);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.0f (%.1f%%)", names_pp[k], hp[k] / nw, 100.0 * hp[k] / tp);
        fprintf(stderr, "\n");
      }
    } else if (tot > 0) {
      fprintf(stderr, "[conv stamps] waves %zu, cycles/wave %.0f:", nwaves, tot / nwaves);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.0f (%.1f%%)", names[k], h[k] / nwaves, 100.0 * h[k] / tot);
      fprintf(stderr, "\n");
    }
  }
#endif
  return 0;
}

int mi355_qkv_attention(const float* qkv, float* out, int batch, int heads, int head_channels, int length, int new_order, int dtype,
                        void* workspace, int64_t workspace_bytes, void* stream) {
  MI355_REQUIRE(qkv && out && workspace, -1, "qkv_attention: null argument");
  hipStream_t s = S(stream);
  MI355_REQUIRE(dtype == MI355_F32 || dtype == MI355_BF16 || dtype == MI355_F16, -1, "qkv_attention: bad dtype");
  dtype = dtype == MI355_F16 ? DT_F16 : dtype;
  const int esz = dtype == 0 ? 4 : 2, C = heads * head_channels;
  char* p = reinterpret_cast<char*>(workspace);
  void* qin = p; p += al256((size_t)batch * length * 3 * C * esz);
  void* o = p; p += al256((size_t)batch * length * C * esz);
  MI355_REQUIRE(p <= reinterpret_cast<char*>(workspace) + workspace_bytes, -2, "qkv_attention: workspace too small");
  int rc;
  if ((rc = pack_nhwc_launch(dtype, qkv, 3 * C, nullptr, 0, batch, length, 3 * C, qin, s))) return rc;
  AttnDesc a; a.dtype = dtype; a.qkv = qin; a.out = o; a.N = batch; a.T = length; a.heads = heads; a.ch = head_channels; a.new_order = new_order;
  if ((rc = attention_launch(a, s))) return rc;
  return unpack_nchw_launch(dtype, o, batch, length, C, out, s);
}

}  // extern "C"
