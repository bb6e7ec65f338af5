/*
 * mi355_sampler.h - C ABI of libmi355_sampler.so (gfx950 / MI355X only).
 *
 * The reference (VladimirRadenkovic/Image-inpainting-and-Super-Resolution-...) is pure Python and has
 * no FFI layer; the drop-in boundary is a set of Python call signatures.  This
 * header is the C ABI that the package's Python mirror of those signatures binds with ctypes
 * (see INTEGRATION.md).  Each entry point cites the reference code it replaces, paths relative to
 * /root/reference, "AD/" = "amortised diffusion/".
 *
 * Conventions
 *   - every `const float* x`-style data pointer is a DEVICE pointer owned by the caller (the PyTorch
 *     allocator), contiguous, NCHW fp32 at the boundary;  host pointers are named *_host;
 *   - `stream` is a hipStream_t passed as void*; functions only enqueue work on it, they never
 *     synchronise and never allocate device memory (workspaces are sized by *_bytes() and passed in);
 *   - return value 0 = ok, negative = error; the message is in mi355_last_error() (thread local);
 *   - no C++ exceptions cross the ABI; no hidden globals besides the error string.
 */
#ifndef MI355_SAMPLER_H
#define MI355_SAMPLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_ARG (-1)
#define MI355_ERR_SHAPE (-2)
#define MI355_ERR_HIP (-3)
#define MI355_ERR_UNSUPPORTED (-4)
#define MI355_ERR_TIMEOUT (-5) /* a kernel's bounded counter wait expired (see mi355_unet_status): results of that launch are invalid */

/* compute type of the contraction path */
#define MI355_F32 0  /* fp32 storage, exact f32 MFMA (v_mfma_f32_16x16x4_f32): tight-parity mode   */
#define MI355_BF16 1 /* bf16 storage + bf16 MFMA, fp32 accumulate; GN stats / softmax / x state fp32 */
#define MI355_BF16X2 2 /* as MI355_BF16 with every conv / qkv weight held as two bf16 halves, hi = bf16(w) and lo = bf16(w - hi), multiplied by the
                        * same bf16 activations and accumulated in fp32 (twice the MFMAs): the weight rounding - 94 % of bf16 mode's distance to
                        * fp32 mode (profiles/r4_quality_delta.json) - is gone; mi355_unet_config::dtype and the test ops' dtype accept it */
#define MI355_F16 3 /* fp16 storage + fp16 MFMA (v_mfma_f32_16x16x32_f16), fp32 accumulate; GN stats / softmax / x state fp32: the reference's own reduced-
                     * precision mode (UNetModel(use_fp16=True) runs its torso in float16, AD/image_diffusion/unet.py:559-563): bf16's speed, three more
                     * mantissa bits on every stored activation and weight; values beyond +-65504 overflow to inf as they do in the reference */

/* ABI version = 100 * major + minor.  The minor number counts additive changes; 105 (round 5, last): mi355_debug_config::sampler_graph (carved out of the
 * reserved tail), mi355_box_probe_hbm.  104 (round 5, later): mi355_conv2d_ex (the small-level conv's fused forms as a
 * test op), gn_epilogue bit 2, conv_small bit 3, conv_edge bits 2-3, conv_pp bit 5, mi355_op_profile::tile_m = -1 for plan ops that launched nothing.
 * 103 (round 5): conv_pp became a bit mask (bits 2, 3, 4: the
 * prologue and narrow forms of the ping-pong kernel), mi355_box_probe, MI355_BF16X2 and MI355_F16 added.  102 (round 4): mi355_debug_config gained conv_pp and
 * conv_edge (carved out of its reserved tail: the struct's size is unchanged), attn_fused became a bit mask, mi355_unet_read_tensor
 * returns MI355_ERR_UNSUPPORTED for a tensor the plan did not materialise as stored.  Callers that fill a mi355_debug_config must start
 * from mi355_debug_defaults() (or zero the struct and set every field): a field this header does not know yet is then at its shipped
 * value, and the reserved words must stay 0. */
int mi355_version(void);
const char* mi355_last_error(void);

/* ---- diagnostic switches -------------------------------------------------------------------------------------------------------
 * Every kernel-path choice the library makes can be overridden here, for experiments and tests only; NULL (or a struct filled by
 * mi355_debug_defaults) = the shipped behaviour.  The struct is COPIED into the handle at mi355_unet_create and passed per call to
 * the test ops: the library reads no environment variable and keeps no process-global switch. */
typedef struct mi355_debug_config {
  int32_t conv_ws;         /* 1: 3x3 convs of the large levels run on the persistent warp-specialised kernel (conv_ws.inc.h); 0: plain tiles */
  int32_t conv_small;      /* 15: bit 0: 8x8 / 4x4 levels run on the LDS-resident-patch kernel (conv_small.inc.h); bit 1: its whole-chip 8x8 launches use eight
                            *    waves of 32 channels (two per SIMD) instead of four of 64; bit 2: the stride-2 Downsample convs 16 -> 8 and 8 -> 4 too;
                            *    bit 3: a ResBlock's 1x1 skip_connection rides in the block's second 3x3 conv as centre-tap K chunks (no launch, no tensor) */
  int32_t conv_min_wgs;    /* 512: the plain kernel takes the largest tile that still yields this many workgroups */
  int32_t conv_stagger;    /* 0: plain kernel: start delay of the odd workgroup slot; persistent kernel: wave priorities (consumer | loader << 2) */
  int32_t conv_ablate;     /* 0: timing experiments, results become wrong: 1 no output stores, 2 no prologue math, 32 the loaders never
                            *    publish kernel row 5 (the consumers' counter wait then expires: exercises MI355_ERR_TIMEOUT) */
  int32_t conv_spin_limit; /* 4194304: polls of an LDS counter before a wait of the persistent kernel gives up and flags the launch */
  int32_t conv_time_reps;  /* 0: mi355_conv2d only: re-launch the conv this many times between two events and print the average */
  int32_t gn_apply_max_hw; /* 64: GroupNorm sites on images up to this many pixels write silu(a x + b) themselves (prologue-free conv) */
  int32_t gn_fuse;         /* 1: GroupNorm statistics come from partial sums in the producing convs' epilogues where possible */
  int32_t l2_warm;         /* 1: bit 0: statistics / apply passes touch the next conv's weights; bit 1: finalize passes too */
  int32_t attn_fused;      /* bit 0: GroupNorm-apply + qkv + attention in one kernel where the shape allows; bit 1: never its persistent form; bits 8..: image lanes of the persistent form (tests) */
  int32_t gn_epilogue;     /* 7: bit 0: at the 8x8 / 4x4 levels a GroupNorm (+SiLU) site whose only source is a small-level conv's output is applied in
                            *    that conv's epilogue (no pass); bit 1: at the 16x16 level (a persistent-conv tile = a whole image) the first conv of
                            *    a ResBlock normalises its own output in place (its own template instantiation), the site's finalize launch
                            *    disappears and the second conv runs prologue-free; bit 2 (with bit 0): at the 8x8 / 4x4 levels the norm of a CONCAT consumer
                            *    (unet.py:650; groups whole inside each source) is applied by the two producers, each its own channels - a skip
                            *    connection's conv then serves two sites in one epilogue; 0: gn_affine pass / finalize launch + prologue */
  int32_t conv_pp;         /* 45: the ping-pong 3x3 kernel (conv_pp.inc.h: 8 MFMA waves in two groups that alternate LDS-read / DMA segments with MFMA
                            *    segments).  Bits 0-1: 1 = it takes an eligible conv when the launch has at least one tile per CU, 2 = whenever the
                            *    shape is eligible (tests), 0 = never.  Bit 2 (4): wide-geometry convs (Cout % 256 == 0) with a GroupNorm + SiLU input
                            *    prologue too (applied in LDS, in place; otherwise such convs stay on the warp-specialised kernel).  Bit 3 (8): the narrow
                            *    geometry (512 pixels x 128 channels) for Cout % 256 != 0, Cout % 128 == 0 on images at least 32 wide.  Bit 4 (16): the
                            *    prologue form of the narrow geometry (off: it only ties the warp-specialised kernel).  The 1x1 ping-pong kernel reads bits 0-1
                            *    and bit 5 (32): 128-pixel x 256-channel tiles where the 256 x 256 walk gives a CU fewer than two tiles */
  int32_t conv_edge;       /* bit 0: the network's last conv (GroupNorm + SiLU -> 3x3 -> <= 4 channels, NCHW fp32) runs on the streaming kernel of
                            *    conv_edge.hip instead of the generic MFMA tile kernel; bit 1: the first conv (<= 8 real input channels -> 128,
                            *    bf16) on conv3x3_in_kernel of the same file (contraction over tap x 8 channels instead of tap x padded chunk); bit 2: that kernel
                            *    reads the caller's fp32 NCHW x (and condition) itself - no packed NHWC copy, no pack launch; bit 3: in
                            *    mi355_cfm_euler_sample the update x += dt * v happens in the last conv's epilogue (v is not stored).  Default 15 */
  int32_t sampler_graph;   /* 0: mi355_cfm_euler_sample enqueues its launches one by one; 1: it captures the whole loop (every network evaluation of every
                            *    step) into ONE hipGraph per (workspace, batch, schedule, condition), instantiated once and kept on the handle, and each
                            *    call is a copy-in, one graph launch, a copy-out.  Carved out of the reserved tail (size unchanged) */
  int32_t reserved[1];
} mi355_debug_config;
void mi355_debug_defaults(mi355_debug_config* out);

/* ---- U-Net (AD/image_diffusion/unet.py:490-728 UNetModel; == torchcfm UNetModelWrapper) ---------- */
/* ZERO-INITIALISE this struct (`mi355_unet_config cfg = {0};`, `memset`) before filling it: fields are only ever added at its end, a zero
 * field means "the behaviour before the field existed" (debug = NULL: shipped kernel paths), and mi355_unet_create dereferences `debug`
 * when it is not NULL - a caller compiled against an older header that leaves the tail uninitialised hands it a garbage pointer.
 * Compare mi355_version() / 100 with the major version the caller was built for before the first call. */

typedef struct mi355_unet_config {
  int32_t image_size;
  int32_t in_channels;
  int32_t model_channels;
  int32_t out_channels;
  int32_t num_res_blocks;
  int32_t n_attention_ds;
  int32_t attention_ds[8];  /* downsample rates with attention (UNetModel.attention_resolutions)   */
  int32_t n_channel_mult;
  int32_t channel_mult[8];  /* integer multipliers only                                            */
  int32_t conv_resample;
  int32_t num_heads;
  int32_t num_head_channels;
  int32_t num_heads_upsample;
  int32_t use_scale_shift_norm;
  int32_t resblock_updown;
  int32_t use_new_attention_order;
  int32_t dtype; /* MI355_F32 | MI355_BF16 | MI355_BF16X2 | MI355_F16 */
  int32_t differentiable; /* 1: keep what mi355_unet_vjp needs (per-site GroupNorm statistics, the qkv tensors, transposed weights) */
  const mi355_debug_config* debug; /* NULL = defaults; copied at mi355_unet_create */
} mi355_unet_config;

typedef struct mi355_unet mi355_unet; /* opaque */

/* Parameter inventory in the reference's state_dict order (unet.py:564-706).  Returns the count,
 * or fills name/shape for parameter `index`.  The Python side feeds tensors in exactly this order. */
int mi355_unet_param_count(const mi355_unet_config* cfg);
int mi355_unet_param_info(const mi355_unet_config* cfg, int index, char* name, int name_cap, int64_t shape[4], int* ndim);

/* Bytes of device memory the packed weights need (caller allocates, 256-byte aligned). */
int64_t mi355_unet_weight_bytes(const mi355_unet_config* cfg);

/* Build the layer plan and pack `params_host[i]` (fp32, reference layouts: conv [Co,Ci,k,k],
 * qkv/proj [Co,Ci,1], linear [out,in]) into kernel layouts inside `dev_weights` (async on stream). */
int mi355_unet_create(const mi355_unet_config* cfg, const float* const* params_host, int n_params, void* dev_weights,
                      int64_t dev_weights_bytes, void* stream, mi355_unet** out);
void mi355_unet_destroy(mi355_unet* net);

int64_t mi355_unet_workspace_bytes(const mi355_unet* net, int batch);

/* 0, or MI355_ERR_TIMEOUT when a launch of this handle gave up a bounded counter wait (the persistent conv never hangs: a stalled
 * hand-over ends after conv_spin_limit polls, the launch's output is then invalid).  The flag is one word of pinned host memory
 * the kernels write through; it is also checked at the start of every call that takes the handle, so a failure is reported by the
 * next call at the latest - synchronise the stream first to learn about the launches already queued.  `clear` resets it. */
int mi355_unet_status(mi355_unet* net, int clear);

/* UNetModel.forward(x, timesteps) (unet.py:708-728).  x: [B, Cx, H, W]; cond: NULL or [B, Cc, H, W]
 * with Cx + Cc == in_channels (the Amortized sampler's channel concat, AD/image_diffusion/sampling.py:39,
 * is folded into the first conv's gather); t: float[B]; out: [B, out_channels, H, W]. */
int mi355_unet_forward(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels,
                       const float* t, float* out, int batch, void* workspace, int64_t workspace_bytes, void* stream);

/* The same with ONE host-side time for the whole batch (the ODE solvers call the vector field as f(t, x) with a scalar t:
 * cifar10/utils_cifar.py:34-39, mnist/utils_mnist.py:96-97): one time-embedding row instead of B, no device tensor for t. */
int mi355_unet_forward_t(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels, float t, float* out,
                         int batch, void* workspace, int64_t workspace_bytes, void* stream);

/* Vector-Jacobian product of the last mi355_unet_forward on this workspace w.r.t. its image input: grad_x = (d out / d x)^T grad_out
 * (and, when cond was given, nothing for cond).  This is the `grad(constraint)` through the x0 model of the reconstruction-guidance
 * sampler (AD/image_diffusion/sampling.py:154-163; per-sample losses make vmap(grad) one batched backward pass).
 * Needs a handle created with cfg.differentiable = 1 and the SAME workspace (it holds the forward's activations).
 * grad_out: [B, out_channels, H, W]; grad_x: [B, x_channels, H, W] (fp32, NCHW). */
int mi355_unet_vjp(mi355_unet* net, const float* grad_out, float* grad_x, int x_channels, int batch, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* Diagnostics: the plan's ops (kind, src0, src1, dst, mode, ks, Cout, use_pro, pro_silu, res, res_mode, gn_site, heads, ch, dst C, dst H;
 * returns the op count) and an activation (or, differentiable plans, its gradient from the last mi355_unet_vjp) as NCHW fp32
 * [B, C, H, W] - the per-tensor view the reference gives through forward hooks / autograd on AD/image_diffusion/unet.py.
 * A non-differentiable plan does not materialise a conv output that nothing but a GroupNorm site fused into that conv's epilogue
 * reads (8x8 / 4x4 levels; its activated copy is the site's dst tensor), and at the 16x16 level the first conv of a ResBlock overwrites its
 * output with silu(GroupNorm(.)) in place: for such a tensor mi355_unet_read_tensor returns MI355_ERR_UNSUPPORTED after the forward that
 * did so (the record is per handle, of its most recent forward); create the handle with debug.gn_epilogue = 0 to inspect them. */
int mi355_unet_plan_op(const mi355_unet* net, int index, int32_t fields[16]);
int mi355_unet_read_tensor(const mi355_unet* net, int tensor, int gradient, float* out, int batch, void* workspace, int64_t workspace_bytes,
                           void* stream);

/* Launch/traffic counters of the last forward (host side bookkeeping, for bench.py's roofline). */
typedef struct mi355_unet_stats {
  int64_t launches; /* device launches of the most recent forward (the planned count before the first one) */
  double conv_flops;      /* 2*MAC of all conv/1x1/linear contractions        */
  double attn_flops;      /* 2*MAC of QK^T and PV                             */
  double act_bytes;       /* algorithmic activation bytes (in + out of each contraction op) */
  double weight_bytes;    /* packed weight bytes                              */
} mi355_unet_stats;
int mi355_unet_get_stats(const mi355_unet* net, int batch, mi355_unet_stats* out);

/* One forward with a HIP event pair around every device op of the plan (synchronises the stream at the end).
 * Fills up to `cap` records in launch order and returns the number of ops; used by bench.py to measure the
 * dominant kernel's average launch duration live, beside its algorithmic FLOPs / bytes. */
#define MI355_OP_PRELUDE 0 /* timestep embedding + time/emb linears + NCHW->NHWC pack (5 launches) */
#define MI355_OP_GN 1
#define MI355_OP_CONV 2
#define MI355_OP_ATTN 3
#define MI355_OP_RESAMPLE 4
typedef struct mi355_op_profile {
  int32_t kind, ks, cin, cout, h, w; /* conv: kernel size, in/out channels, OUTPUT height/width */
  int32_t tile_m, tile_n;            /* conv: workgroup tile; -1, -1: the plan op launched nothing in this forward (a GroupNorm site its producers applied, a 1x1 skip conv
                                      * that rode in the next 3x3 conv): its ms is the event pair's own cost */
  float ms;
  double flops; /* algorithmic 2*MAC */
  double bytes; /* algorithmic bytes: activations in + out (+ weights once) */
} mi355_op_profile;
int mi355_unet_profile(mi355_unet* net, const float* x, int x_channels, const float* cond, int cond_channels, const float* t,
                       float* out, int batch, void* workspace, int64_t workspace_bytes, void* stream,
                       mi355_op_profile* recs, int cap);

/* ---- sampler loops ------------------------------------------------------------------------------ */

/* Fixed-step Euler CFM sampler: x_{k+1} = x_k + (t_{k+1}-t_k) * model(t_k, x_k)
 * (torchdyn NeuralODE(solver="euler").trajectory as called at cifar10/compute_fid.py:69-79,
 *  cifar10/utils_cifar.py:34-39, mnist/utils_mnist2.py:125-134).
 * x: in/out [B, Cx, H, W].  traj: NULL or [n_t, B, Cx, H, W] (all states, as the reference returns).
 * u8_out: NULL or uint8 [B, Cx, H, W] = (x*127.5+128).clip(0,255) (cifar10/compute_fid.py:87). */
/* cond_drift != 0: the condition is part of the integrated state with derivative `cond` itself, as in the reference's
 * concatenated-state Euler sampler (mnist/utils_mnist2.py:118-138: ode_func returns cat(x_t, x[:,1])), i.e. the model sees
 * cond_{k+1} = cond_k + dt*cond_k; the caller's `cond` tensor is left untouched (a scratch copy drifts). */
int mi355_cfm_euler_sample(mi355_unet* net, float* x, int x_channels, const float* cond, int cond_channels, int cond_drift,
                           const float* t_span_host, int n_t, float* traj, uint8_t* u8_out, int batch,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* per-step scalars of the DDPM tables (AD/image_diffusion/sde_diffusion.py:127-167), host arrays of length Ns */
typedef struct mi355_ddpm_tables {
  int32_t Ns;
  const float* sqrt_recip_alphas_cumprod;
  const float* sqrt_recipm1_alphas_cumprod;
  const float* posterior_mean_coef1;
  const float* posterior_mean_coef2;
  const float* posterior_log_variance_clipped;
  const float* sqrt_alphas_cumprod;
  const float* sqrt_one_minus_alphas_cumprod;
  const float* recip_sqrt_m1_alphas_cumprod;
  const float* alphas_cumprod_prev;
} mi355_ddpm_tables;

#define MI355_DDPM_PRIOR 0       /* get_prior_sample_fn                 sampling.py:50-75   */
#define MI355_DDPM_AMORTIZED 1   /* get_conditional_sample_fn[Amortized] sampling.py:80-133  */
#define MI355_DDPM_REPLACEMENT 2 /* get_conditional_sample_fn[Replacement] sampling.py:209-260 */
#define MI355_DDIM 3             /* build-defined deterministic DDIM(eta=0) on the same tables */

typedef struct mi355_ddpm_options {
  int32_t mode;
  int32_t n_corrector;      /* Langevin corrector steps per predictor step (sampling.py:113-121) */
  float delta;              /* corrector step size                                                */
  float tmin, tmax;         /* DDPM.tmin / tmax (sde_diffusion.py:130-132)                        */
  float start_fraction;     /* Replacement: replace while i < int(Ns*start_fraction)              */
  int32_t noise_condition;  /* Replacement: q_sample the condition (1) or use it as is (0)        */
  float pad_value;          /* Replacement: mask sentinel (-2)  likelihoods.py:55-56              */
  float none_value;         /* Amortized: value of likelihood.none_like (-2 painting, 0 hyperres) */
  int32_t cond_is_none;     /* Amortized prior (cond == NULL): feed none_like as the condition    */
  uint64_t seed;            /* device Philox seed when noise == NULL                              */
} mi355_ddpm_options;

/* Reverse-denoising loop i = Ns-1..0.  x: in/out [B, C, H, W] (xT -> x0, clipped to [-1,1]).
 * cond: NULL or [B, C, H, W].  noise: NULL (device Philox) or the injected draws, one [B,C,H,W]
 * tensor per torch.randn_like call of the reference, in call order. */
int mi355_ddpm_sample(mi355_unet* net, float* x, int channels, const float* cond, const mi355_ddpm_tables* tables,
                      const mi355_ddpm_options* opt, const float* noise, int64_t n_noise_draws, int batch,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- single ops (used by the Python mirror for arbitrary eps_model callables, and by the tests) --- */

/* timestep_embedding (AD/image_diffusion/nn.py:97-115): t[B] -> out[B, dim] */
int mi355_timestep_embedding(const float* t, int batch, int dim, float max_period, float* out, void* stream);

/* GroupNorm32 (+ optional SiLU) on NCHW fp32 (nn.py:11-13,87-94), standalone op for parity tests */
int mi355_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int batch, int channels, int hw,
                    int groups, float eps, int silu, void* stream);

/* x += dt * v  (Euler update) */
int mi355_euler_step(float* x, const float* v, float dt, int64_t n, void* stream);

/* DDPM ancestral step (sampling.py:59-67 + sde_diffusion.py:220-237), elementwise over n values:
 *   x0 = clip(c_recip*x - c_recipm1*eps, -1, 1); mean = coef1*x0 + coef2*x; x <- mean + sigma*z
 * z == NULL and use_philox == 0 means "no noise" (i == 0). */
int mi355_ddpm_step(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float coef1, float coef2,
                    float sigma, int use_philox, uint64_t seed, uint64_t offset, int64_t n, void* stream);

/* Langevin corrector (sampling.py:113-121): x += 0.5*dt*delta*score + sqrt(dt*delta)*z,
 * score = -recip_sqrt_m1 * clip(c_recip*x - c_recipm1*eps, -1, 1) */
int mi355_corrector_step(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float recip_sqrt_m1,
                         float dt, float delta, int use_philox, uint64_t seed, uint64_t offset, int64_t n, void* stream);

/* DDIM(eta=0) step on the same tables (build-defined extension) */
int mi355_ddim_step(float* x, const float* eps, float c_recip, float c_recipm1, float acp_prev, int64_t n, void* stream);

/* Replacement mask (sampling.py:225-232): x = where(cond == pad, x, noisy ? sa*cond + sb*z : cond) */
int mi355_replace_mask(float* x, const float* cond, const float* z, float pad_value, int noisy, float sa, float sb,
                       int use_philox, uint64_t seed, uint64_t offset, int64_t n, void* stream);

/* Reconstruction guidance, elementwise parts (sampling.py:148-172 + likelihoods.py:58-66,138-143 + sde_diffusion.py:220-224):
 *   seed  : x0 = clip(c_recip x - c_recipm1 eps); g = d loss_n / d x0 (mode 0: Painting.loss, entries with cond == pad_value masked;
 *           mode 1: HyperResolution.loss = mean squared error over the sample's `per` entries), through the clip;
 *           g_eps = -c_recipm1 g (cotangent for mi355_unet_vjp), g_x = c_recip g (direct path of predict_start_from_noise)
 *   update: u = -scale (g_x + vjp); update <- u; if apply: x += u   ("before" rule; "after" adds `update` to the next state) */
int mi355_guidance_seed(const float* x, const float* eps, const float* cond, float c_recip, float c_recipm1, int mode, float pad_value,
                        int64_t elems_per_sample, float* g_eps, float* g_x, int64_t n, void* stream);
int mi355_guidance_update(float* x, const float* g_x, const float* vjp, float scale, int apply, float* update, int64_t n, void* stream);

/* clip(x, lo, hi) in place, NaN-propagating like torch.clip (sampling.py:13-14) */
int mi355_clip(float* x, float lo, float hi, int64_t n, void* stream);
/* EMA of one parameter tensor, in place: target = target*decay + source*(1-decay)  (cifar10/utils_cifar.py:47-53, called per
 * state-dict entry at cifar10/train_cifar10.py:154).  one_minus_decay is passed separately because the reference rounds
 * (1 - decay) from a Python double. */
int mi355_ema_update(float* target, const float* source, float decay, float one_minus_decay, int64_t n, void* stream);
/* out[n] = mean over (C,H,W) of (a - b)^2: the per-sample MSE metric of the evaluation loop
 * (AD/experiments/main.py:299 `torch.mean((x0 - batch)**2, dim=(1, 2, 3))`) */
int mi355_mse_per_sample(const float* a, const float* b, float* out, int batch, int64_t elems_per_sample, void* stream);
/* out[n,:] = a[n]*x[n,:] + b[n]*y[n,:] with per-sample device coefficients a, b (y and b may both be NULL: out = a*x).
 * The arithmetic of DDPM.predict_start_from_noise / q_posterior / q_sample / score_from_x0
 * (AD/image_diffusion/sde_diffusion.py:214-244), whose coefficients are `extract`ed per sample (sde_diffusion.py:101-104). */
int mi355_lincomb_per_sample(float* out, const float* x, const float* y, const float* a, const float* b, int batch,
                             int64_t elems_per_sample, void* stream);
/* Condition builders (once per batch / per solve, never per step).
 *   resize_bilinear: F.interpolate(x, size=(h_out, w_out), mode="bilinear", align_corners=False) on `planes` = N*C fp32 planes:
 *                    downsample_images (mnist/utils_mnist_hy.py:18-28), HyperResolution._sample / .loss
 *                    (AD/image_diffusion/likelihoods.py:119-126,138-143), the SuperRes wrapper's up-sampling of `low_res`
 *                    (mnist/utils_mnist_hy.py:82).
 *   paint_patch    : InPainting._sample / OutPainting._sample (likelihoods.py:78-87,95-104) for the whole batch: image n's
 *                    patch_size x patch_size window at (top[n], left[n]) (device int32 arrays, drawn by the caller in the
 *                    reference's order, likelihoods.py:22-27,49-53) becomes pad_value (outpaint = 0) or is the only part kept
 *                    (outpaint = 1). */
int mi355_resize_bilinear(const float* in, float* out, int64_t planes, int h_in, int w_in, int h_out, int w_out, void* stream);
int mi355_paint_patch(const float* images, const int32_t* top, const int32_t* left, int patch_size, float pad_value, int outpaint,
                      float* out, int batch, int channels, int h, int w, void* stream);
/* (x*127.5+128).clip(0,255).to(uint8)  cifar10/compute_fid.py:87 */
int mi355_quantize_u8(const float* x, uint8_t* out, int64_t n, void* stream);
/* x.clip(-1,1)/2 + 0.5  cifar10/utils_cifar.py:40-41 */
int mi355_to_unit_range(const float* x, float* out, int64_t n, void* stream);
/* N(0,1) fill from the device Philox4x32-10 stream (seed, offset) */
int mi355_randn(float* out, uint64_t seed, uint64_t offset, int64_t n, void* stream);

/* Adaptive Dormand-Prince 5(4) building blocks (torchdiffeq.odeint(method="dopri5"): cifar10/compute_fid.py:80-85,
 * mnist/utils_mnist.py:101-108; un-vendored, algorithm restated).  The step-size controller runs on the host.
 *   rk_combine: out = y0 + sum_j coeff_host[j] * k_j   (coefficients already multiplied by dt; y0 may be NULL)
 *   rk_sqnorm : *out += sum_i ((a_i - sub_i) / (atol + rtol * max(|b_i|, |b2_i|)))^2   (sub, b, b2 may be NULL; fp64)
 *   rk_interp : torchdiffeq's quartic dense output at x = (t - t0) / dt                                   */
int mi355_rk_combine(float* out, const float* y0, const float* k0, const float* k1, const float* k2, const float* k3, const float* k4,
                     const float* k5, const float* k6, const float* coeff_host, int nk, int64_t n, void* stream);
int mi355_rk_sqnorm(const float* a, const float* sub, const float* b, const float* b2, float atol, float rtol, int64_t n, double* out,
                    void* stream);
int mi355_rk_interp(float* out, const float* y0, const float* y1, const float* y_mid, const float* f0, const float* f1, float dt, float x,
                    int64_t n, void* stream);

/* ---- measurement: box calibration probe (bench.py roofline.box; no reference counterpart) ------------------------------------------
 * A FROZEN kernel (csrc/box_probe.hip: 1.0995 TFLOP of bf16 v_mfma_f32_16x16x32 on seeded pseudo-random register operands, no memory
 * traffic in its loop) that no round edits: boxes of the pool differ by several per cent for one build and the chip lowers its clock in
 * MFMA-dense kernels, so throughput lines of different boxes / rounds are compared through this launch's duration.  Runs `reps` warm
 * launches, then `reps` timed ones (HIP events on `stream`), synchronises; *us_per_launch = mean duration, *clock_mhz = median in-kernel
 * clock of the last launch (s_memtime / s_memrealtime), *tflop = the work of one launch.  workspace: mi355_box_probe_workspace_bytes(). */
int64_t mi355_box_probe_workspace_bytes(void);
int mi355_box_probe(int reps, void* workspace, int64_t workspace_bytes, void* stream, float* us_per_launch, float* clock_mhz, float* tflop);
/* The memory side of the same calibration (csrc/box_probe_hbm.hip, FROZEN as well; ABI 105): one launch copies 512 MiB to another 512 MiB (each twice
 * the Infinity Cache) with 16-byte accesses; two warm launches, then `reps` timed ones.  *us_per_launch = mean duration, *gbytes = bytes moved per
 * launch (read + written) / 1e9.  workspace: mi355_box_probe_hbm_workspace_bytes() (1 GiB; contents are irrelevant and overwritten). */
int64_t mi355_box_probe_hbm_workspace_bytes(void);
int mi355_box_probe_hbm(int reps, void* workspace, int64_t workspace_bytes, void* stream, float* us_per_launch, float* gbytes);

/* Standalone conv / attention ops on NCHW fp32 tensors for parity tests of the HIP kernels
 * (pack -> implicit-GEMM MFMA kernel -> unpack; `workspace` from mi355_op_workspace_bytes). */
int64_t mi355_op_workspace_bytes(int batch, int max_channels, int hw);
/* Conv2d k in {1,3}, stride in {1,2}, padding k/2 - w_host [Co,Ci+Ci1,k,k], bias_host [Co] are HOST pointers.
 * x1 (NULL or [B, cin1, H, W]): second source of a never-materialised channel concat th.cat([x, x1], 1) (unet.py:725).
 * resample: 0 none, 2 nearest x2 upsample of the input (unet.py:185-212), 3 2x2 average pool of the input.
 * Optional fused prologue GroupNorm32(gamma, beta) [+ SiLU] on the (concatenated) input (gn_gamma/gn_beta device
 * pointers or NULL), i.e. the ResBlock in_layers / out_layers (unet.py:283-286,306-311).
 * Optional fused epilogue: + emb[B, Co] (ResBlock emb_layers, unet.py:349) and + res (NULL or [B, Co, Hr, Wr]; res_mode 1:
 * Hr = Ho (skip + h, unet.py:351,401), 2: nearest x2 of a half-size tensor (ResBlock(up=True), unet.py:332-334)).
 * Synchronises the stream (test op: the packed weights are staged from a temporary host buffer); returns MI355_ERR_TIMEOUT if the
 * launch flagged an expired counter wait.  debug: NULL = defaults. */
int mi355_conv2d(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                 int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                 const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                 int64_t workspace_bytes, void* stream);
/* The same op with the fused forms the engine asks of the 8x8 / 4x4-level conv (csrc/conv_small.inc.h), for op-level parity tests:
 *  - a ResBlock's 1x1 skip_connection (unet.py:312-317, 351) over cat(skip_x0, skip_x1) accumulated into the same output as centre-tap K chunks
 *    (skip_w_host [Co, c0 + c1], skip_bias_host [Co]: HOST pointers; needs ksize 3, stride 1, one source, no residual).  MI355_ERR_UNSUPPORTED
 *    if the launch cannot carry it;
 *  - up to two GroupNorm32 (+ SiLU) sites applied in the epilogue (unet.py:196-212, 650; nn.py:87-94): site k normalises this conv's Co output
 *    channels as channels act_coff[k] .. of a tensor of act_ctotal[k] channels (groups of act_ctotal[k] / 32; gamma / beta [act_ctotal[k]],
 *    device) and writes them into act_out[k] ([B, act_ctotal[k], Ho, Wo] fp32, device; the other channels are written as zeros); FiLM
 *    (act_film [B, 2 Co] = scale | shift, unet.py:343-347) on site 0 only.  act_done reports which sites the launch applied (bit k); a
 *    site it did not apply leaves its act_out untouched.  y is the conv's own output as in mi355_conv2d. */
typedef struct mi355_conv_extras {
  const float* skip_x0; const float* skip_x1; int32_t skip_c0, skip_c1;
  const float* skip_w_host; const float* skip_bias_host;
  float* act_out[2]; const float* act_gamma[2]; const float* act_beta[2];
  int32_t act_ctotal[2], act_coff[2], act_silu[2];
  const float* act_film;
  int32_t act_done, skip_done;   /* out */
} mi355_conv_extras;
int mi355_conv2d_ex(const float* x, const float* x1, int cin1, const float* w_host, const float* bias_host, float* y, int batch, int cin,
                    int h, int w, int cout, int ksize, int stride, int resample, const float* gn_gamma, const float* gn_beta, int gn_silu,
                    const float* emb, const float* res, int res_mode, int dtype, const mi355_debug_config* debug, void* workspace,
                    int64_t workspace_bytes, void* stream, mi355_conv_extras* extras);
/* QKVAttentionLegacy / QKVAttention (unet.py:424-487): qkv [B, 3*H*ch, T] -> out [B, H*ch, T] */
int mi355_qkv_attention(const float* qkv, float* out, int batch, int heads, int head_channels, int length, int new_order,
                        int dtype, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SAMPLER_H */
