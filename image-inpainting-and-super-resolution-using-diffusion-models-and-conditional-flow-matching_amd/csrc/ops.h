// Internal C++ launch API shared by the U-Net executor, the sampler loops and the C ABI.
#pragma once
#include "common.h"
#include "../../include/mi355_sampler.h"

const mi355_debug_config& mi355_default_debug();   // the shipped behaviour (capi.hip)

// INTERNAL element-type codes of every descriptor below (the C ABI's MI355_* dtype values are translated at the boundary, capi.hip: MI355_BF16X2 ->
// DT_BF16 + ConvDesc::wsplit, MI355_F16 -> DT_F16).  Sizes and chunking depend only on "4-byte or 2-byte": dtype == 0 ? 4 : 2.
enum { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
// run f(T()) for the element type of an internal dtype code
template <typename F> auto dispatch_dtype(int dtype, F&& f) {
  if (dtype == DT_F32) return f(float());
  if (dtype == DT_F16) return f(f16());
  return f(bf16());
}

// ---- implicit-GEMM convolution -------------------------------------------------------------------
// Activations are NHWC in HBM ([N][H][W][C], element type float or bf16).  out[n, y, x, co] =
//   bias[co] + emb[n, co] + res[...] + sum_{tap, ci} W[co, ci, tap] * P(in)[n, y*s + ky - pad, x*s + kx - pad, ci]
// where P is the optional fused prologue  P(v) = silu?(a[n, ci] * v + b[n, ci])  (GroupNorm affine,
// FiLM folded into a/b) evaluated while the input patch is staged into LDS, and `in` is read through
// a gather: identity | nearest x2 upsample | 2x2 average pool (both applied AFTER P, as the reference
// does: ResBlock up/down resamples between SiLU and the conv, unet.py:332-337).
enum ConvMode { CONV_UNIT = 0, CONV_STRIDE2 = 1, CONV_UP2 = 2, CONV_POOL2 = 3 };
enum ResMode { RES_NONE = 0, RES_SAME = 1, RES_UP2 = 2, RES_POOL2 = 3 };
enum OutMode { OUT_NHWC = 0, OUT_NCHW_F32 = 1 };

struct ConvDesc {
  int dtype;                 // MI355_F32 / MI355_BF16
  const void* src0 = nullptr; int C0 = 0;   // first source (channels = row stride of its NHWC tensor)
  const void* src1 = nullptr; int C1 = 0;   // optional second source (skip concat, unet.py:725), never materialised
  int N = 0, Hs = 0, Ws = 0;                // source spatial size
  int mode = CONV_UNIT;
  int ks = 3;
  const float* pro_a = nullptr;             // [N][C0+C1] or null
  const float* pro_b = nullptr;
  int pro_silu = 0;
  const void* w = nullptr;                  // packed by conv_pack_weights
  const float* bias = nullptr;              // [Cout]
  int Cout = 0;
  const float* emb = nullptr; int emb_stride = 0;   // per-(n, co) additive term (ResBlock emb_layers)
  const void* res = nullptr; int res_mode = RES_NONE;  // residual NHWC tensor with Cout channels
  void* out = nullptr; int out_mode = OUT_NHWC;
  // optional fused GroupNorm statistics of the OUTPUT (common.h GnPartial): [N][slots][Cout/4][2] fp32, room for gn_slots_cap slots
  // per image; conv_launch reports the slots it filled (0 = this launch cannot produce them: the caller runs the stats kernel)
  float* gn_stats = nullptr; int gn_slots_cap = 0;
  void* dbg = nullptr;                      // diagnostic builds only (-DCONV_STAMPS): 9 x u64 phase-cycle sums
  const mi355_debug_config* knobs = nullptr; // diagnostic switches (null = defaults)
  uint32_t* err = nullptr;                  // device-visible error word: the persistent kernel ORs 1 into it when a counter wait expires
  // Optional: the GroupNorm (+ SiLU) site that reads this conv's output, applied in the conv's epilogue where the kernel holds whole
  // images x whole groups per wave (conv_small.inc.h): act_out = silu?(GN(out)) in out's layout; out itself only if act_raw; one touch
  // of the next conv's packed weights (warm).  conv_launch reports through *act_done whether the launch did it (0: run the pass).
  void* act_out = nullptr; const float* act_gamma = nullptr; const float* act_beta = nullptr;
  const float* act_film = nullptr; int act_film_stride = 0; float act_eps = 1e-5f; int act_silu = 0, act_raw = 1;
  // Where the site is the GroupNorm of a CONCAT consumer (unet.py:650: h = cat([h, hs.pop()])) and its groups are whole inside each
  // source, this conv's output is one source of it: act_out then has act_stride channels per pixel (0 = Cout) of which this conv fills
  // act_coff ..., in groups of act_cpg channels (0 = Cout / 32); act_gamma / act_beta already point at channel act_coff.
  int act_stride = 0, act_coff = 0, act_cpg = 0;
  // A skip connection is read by TWO such sites (the next ResBlock's in_layers norm now, the up path's concat norm later): the second
  // one, same conventions, never FiLM.  *act_done: bit 0 = the first site was applied, bit 1 = the second.
  void* act2_out = nullptr; const float* act2_gamma = nullptr; const float* act2_beta = nullptr;
  int act2_silu = 0, act2_stride = 0, act2_coff = 0, act2_cpg = 0;
  const void* warm = nullptr; uint32_t warm_bytes = 0;
  // Optional (small-level kernel only; ask conv_fused_skip_ok first): out += conv1x1(cat(skip_src0, skip_src1)) - a ResBlock's skip_connection
  // (unet.py:312-317, 351) inside its second conv: centre-tap K chunks of the raw block input at the output resolution.  w then is the image
  // conv_pack_weights_skip made, bias the sum of both biases, res none, src1 none.
  const void* skip_src0 = nullptr; const void* skip_src1 = nullptr; int skip_C0 = 0, skip_C1 = 0;
  // Optional, the network's two edge convs (conv_edge.hip), sampler loops: (i) first conv: the caller's fp32 NCHW tensors x (nchw_c0 channels)
  // and condition (nchw_c1) read directly while the patch is staged - src0's packed NHWC copy (pack_nhwc) is then never made
  // (conv_in_reads_nchw says whether the launch would); (ii) last conv (NCHW fp32 output v): axpy_x += axpy_scale * v instead of storing
  // v - the Euler update x <- x + dt v of the flow-matching sampler (mnist/utils_mnist2.py:118-138) in the epilogue; *axpy_done = 1 if done
  const float* nchw0 = nullptr; const float* nchw1 = nullptr; int nchw_c0 = 0, nchw_c1 = 0;
  float* axpy_x = nullptr; float axpy_scale = 0.f; int* axpy_done = nullptr;
  int cin_real = 0;                         // > 0: only the first cin_real channels of src0 are non-zero (the network's first conv: in_channels padded to a chunk)
  int wsplit = 0;                           // 1 (bf16x2 precision): w holds [bf16(w) | bf16(w - bf16(w))] along K (twice the chunks): the contraction runs over the
                                            // input channels twice, once against each half - fp32 accumulation of both, activations read (not stored) twice
};

struct ConvGeom {
  int Ho, Wo, BM, BN;
  size_t lds_bytes;
  int grid_m, grid_n;
};
ConvGeom conv_geometry(const ConvDesc& d);
// bytes of the packed weight image for a [Cout][Cin][ks][ks] filter
size_t conv_packed_weight_bytes(int dtype, int Cout, int Cin, int ks, int split = 0);
int conv_tile_n(int Cout);
// host-side packing: w_host [Cout][Cin][ks][ks] fp32 (Cin = logical input channels; padded to a chunk)
// split = 1 (bf16 only): [hi | lo] halves along K, hi = bf16(w), lo = bf16(w - hi); the chunk count doubles
void conv_pack_weights(int dtype, const float* w_host, int Cout, int Cin, int ks, void* dst_host, int split = 0);
int conv_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used = nullptr, int* act_done = nullptr);
int conv_in_reads_nchw(const ConvDesc& d);    // 0 = conv_launch(d) with nchw0 set would take the first-conv kernel and read the fp32 NCHW tensors itself
int conv_fused_skip_ok(const ConvDesc& d);   // 0 = conv_launch(d) would launch with its fused skip conv (nothing is launched here)
// [3x3 filter w3 [Cout][Cin][3][3] | 1x1 filter w1 [Cout][Cskip]] per 128-channel pack tile: the 3x3 tiles, then one tile per skip chunk
size_t conv_packed_weight_bytes_skip(int dtype, int Cout, int Cin, int Cskip);
void conv_pack_weights_skip(int dtype, const float* w3, const float* w1, int Cout, int Cin, int Cskip, void* dst_host);
// 1x1 GEMM with a stationary activation tile (conv1x1.hip): 0 = launched, 1 = not eligible, <0 = error
int conv1x1_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used = nullptr);
// 1x1 GEMM in the ping-pong structure (conv_pp1.inc.h: 256 pixels x 256 channels, both operands streamed by DMA): same return convention
int conv1x1_pp_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used = nullptr);
int conv_in_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used = nullptr);   // conv_edge.hip: the network's first conv
// the network's last conv (GN + SiLU prologue, <= 4 output channels, NCHW fp32) as a streaming kernel (conv_edge.hip): same convention
int conv_out_try_launch(const ConvDesc& d, hipStream_t stream);

// ---- GroupNorm statistics -> per-(n, channel) affine ----------------------------------------------
// a[n,c] = rstd * gamma[c] (* (1 + film_scale)), b[n,c] = beta[c] - mean * rstd * gamma[c] (FiLM folded)
struct GnDesc {
  int dtype;
  const void* src0 = nullptr; int C0 = 0;
  const void* src1 = nullptr; int C1 = 0;
  int N = 0, HW = 0;
  int groups = 32;
  float eps = 1e-5f;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* film = nullptr; int film_stride = 0;  // [N][2C]: scale | shift (use_scale_shift_norm)
  float* a = nullptr;
  float* b = nullptr;
  void* y = nullptr; int y_silu = 0;   // optional: also write silu?(a*x + b), NHWC [N][HW][C0 + C1] (the consumer conv then has no prologue)
  float* mean = nullptr; float* rstd = nullptr;   // optional [N][groups]: kept for the backward pass (reconstruction guidance)
  const void* warm = nullptr; uint32_t warm_bytes = 0;   // optional: packed weights of the conv this pass feeds (common.h l2_warm_issue)
};
int gn_affine_launch(const GnDesc& d, hipStream_t stream);
// The same (a, b) from the partial sums the producing convs left (ConvDesc::gn_stats): no pass over the activation.
struct GnFinDesc {
  const float* stats0 = nullptr; int slots0 = 0, C0 = 0;
  const float* stats1 = nullptr; int slots1 = 0, C1 = 0;
  int N = 0, HW = 0, groups = 32;
  float eps = 1e-5f;
  const float* gamma = nullptr; const float* beta = nullptr;
  const float* film = nullptr; int film_stride = 0;
  float* a = nullptr; float* b = nullptr;
  const void* warm = nullptr; uint32_t warm_bytes = 0;   // as GnDesc::warm
};
int gn_finalize_launch(const GnFinDesc& d, hipStream_t stream);
// out = avgpool2x2(silu?(a * in + b)) on NHWC tensors (a, b per (n, c), may be null): the ResBlock(down=True) input path
int affine_pool_launch(int dtype, const void* in, const float* a, const float* b, int silu, void* out, int N, int Hs, int Ws, int C,
                       hipStream_t s);

// ---- attention --------------------------------------------------------------------------------------
// qkv: NHWC [N][T][3*C] with the reference channel order (legacy: per head [q|k|v]; new: [q heads|k heads|v heads])
// out: NHWC [N][T][C]
struct AttnDesc {
  int dtype;
  const void* qkv = nullptr;
  void* out = nullptr;
  int N = 0, T = 0, heads = 0, ch = 0;
  int new_order = 0;
};
int attention_launch(const AttnDesc& d, hipStream_t stream);
// GroupNorm-apply -> qkv 1x1 -> attention of one AttentionBlock in one kernel (attn_fused.hip): x NHWC [N][T][C], (a, b) [N][C],
// w / bias = the qkv conv's packed weights (conv_pack_weights, ks 1, Cout 3C) / bias [3C]; out NHWC [N][T][C]
struct AttnFusedDesc {
  int dtype;
  const void* x = nullptr; const float* ga = nullptr; const float* gb = nullptr;
  const void* w = nullptr; const float* bias = nullptr;
  void* out = nullptr;
  int N = 0, T = 0, C = 0, heads = 0, ch = 0, new_order = 0;
  const mi355_debug_config* knobs = nullptr;
};
bool attn_fused_eligible(int dtype, int T, int C, int heads, int ch, const mi355_debug_config* knobs = nullptr);
int attn_fused_launch(const AttnFusedDesc& d, hipStream_t stream);

// ---- backward (data gradient only: reconstruction guidance, AD/image_diffusion/sampling.py:136-206) --------------------------------
void conv_pack_weights_dgrad(int dtype, const float* w_host, int Cout, int Cin, int ks, int cin_pad, void* dst_host);
size_t conv_packed_weight_bytes_dgrad(int dtype, int Cout, int Cin, int ks, int cin_pad);
struct GnBwdDesc {
  int dtype;
  const void* x0 = nullptr; const void* x1 = nullptr; int C0 = 0, C1 = 0;
  const void* du = nullptr; int du_stride = 0;
  int N = 0, HW = 0, groups = 32, silu = 0;
  const float* a = nullptr; const float* b = nullptr; const float* mean = nullptr; const float* rstd = nullptr;
  void* g0 = nullptr; void* g1 = nullptr; int acc0 = 0, acc1 = 0;
};
int gn_silu_bwd_launch(const GnBwdDesc& d, hipStream_t stream);
enum { GATHER_SAME = 0, GATHER_POOL = 1, GATHER_UP = 2, GATHER_STUFF = 3 };
int grad_gather_launch(int dtype, void* dst, const void* src, int N, int Hd, int Wd, int Cd, int Hs, int Ws, int src_cstride, int src_coff,
                       int mode, int accumulate, float scale, hipStream_t s);
struct AttnBwdDesc {
  int dtype;
  const void* qkv = nullptr; const void* a = nullptr; const void* da = nullptr; void* dqkv = nullptr;
  float* L = nullptr; float* D = nullptr;
  int N = 0, T = 0, heads = 0, ch = 0, new_order = 0;
};
int attention_bwd_launch(const AttnBwdDesc& d, hipStream_t stream);
int guidance_seed_launch(const float* x, const float* eps, const float* cond, float c_recip, float c_recipm1, int mode, float pad,
                         int64_t per, float* g_eps, float* g_x, int64_t n, hipStream_t s);
int guidance_update_launch(float* x, const float* g_x, const float* vjp, float scale, int apply, float* update, int64_t n, hipStream_t s);
// out[n][c][hw] (NCHW fp32, c < count) = in[n][hw][c0 + c] (NHWC T, row stride `stride`)
int unpack_channels_launch(int dtype, const void* in, int N, int HW, int stride, int c0, int count, float* out, hipStream_t s);

// ---- box calibration probe (box_probe.hip: frozen) ------------------------------------------------------------------------------
int64_t box_probe_workspace_bytes();
double box_probe_flops();
int box_probe_run(int reps, void* workspace, int64_t workspace_bytes, hipStream_t s, float* us_per_launch, float* clock_mhz);
// memory side (box_probe_hbm.hip: frozen): one launch copies 512 MiB to another 512 MiB
int64_t box_probe_hbm_workspace_bytes();
double box_probe_hbm_bytes();
int box_probe_hbm_run(int reps, void* workspace, int64_t workspace_bytes, hipStream_t s, float* us_per_launch);

// ---- small fp32 ops -----------------------------------------------------------------------------------
int timestep_embedding_launch(const float* t, int B, int dim, float max_period, float* out, hipStream_t s);
// out[b][j] = bias[j] + sum_k act(in[b][k]) * Wt[k][j]   (Wt = transposed weights, [K][J]); act: 0 none, 1 silu(in)
// out_act: 0 none, 1 silu(out)
int linear_launch(const float* in, const float* Wt, const float* bias, float* out, int B, int K, int J, int in_act,
                  int out_act, hipStream_t s);

// NCHW fp32 (x | cond) -> NHWC T with channels padded to Cpad (zero fill)
int pack_nhwc_launch(int dtype, const float* x, int Cx, const float* cond, int Cc, int N, int HW, int Cpad, void* out,
                     hipStream_t s);
// NHWC T [N][HW][C] -> NCHW fp32
int unpack_nchw_launch(int dtype, const void* in, int N, int HW, int C, float* out, hipStream_t s);
// plain resampling of NHWC tensors (conv_resample=False paths): mode CONV_UP2 / CONV_POOL2
int resample_launch(int dtype, const void* in, void* out, int N, int Hs, int Ws, int C, int mode, hipStream_t s);

// elementwise sampler steps (steps.hip)
int euler_step_launch(float* x, const float* v, float dt, int64_t n, hipStream_t s);
int ddpm_step_launch(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float coef1, float coef2,
                     float sigma, int use_philox, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s);
int corrector_step_launch(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float rsm1, float dt,
                          float delta, int use_philox, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s);
int ddim_step_launch(float* x, const float* eps, float c_recip, float c_recipm1, float acp_prev, int64_t n, hipStream_t s);
int replace_mask_launch(float* x, const float* cond, const float* z, float pad, int noisy, float sa, float sb,
                        int use_philox, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s);
int clip_launch(float* x, float lo, float hi, int64_t n, hipStream_t s);
int mse_per_sample_launch(const float* a, const float* b, float* out, int batch, int64_t per, hipStream_t s);
int lincomb_per_sample_launch(float* out, const float* x, const float* y, const float* a, const float* b, int batch, int64_t per,
                              hipStream_t s);
int ema_update_launch(float* target, const float* source, float decay, float one_minus_decay, int64_t n, hipStream_t s);
int quantize_u8_launch(const float* x, uint8_t* out, int64_t n, hipStream_t s);
int to_unit_range_launch(const float* x, float* out, int64_t n, hipStream_t s);
int randn_launch(float* out, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s);
int fill_launch(float* x, float v, int64_t n, hipStream_t s);
// condition builders (steps.hip): bilinear resize of NCHW fp32 planes (align_corners = False), square-patch painting of a batch
int resize_bilinear_launch(const float* in, float* out, int64_t planes, int Hi, int Wi, int Ho, int Wo, hipStream_t s);
int paint_patch_launch(const float* img, const int* top, const int* left, int patch, float pad, int outpaint, float* out, int N, int C,
                       int H, int W, hipStream_t s);
int groupnorm_nchw_launch(const float* x, const float* gamma, const float* beta, float* y, int N, int C, int HW, int groups,
                          float eps, int silu, hipStream_t s);

// adaptive RK45 helpers (ode.hip)
int rk_combine_launch(float* out, const float* y0, const float* const* k, const float* c, int nk, int64_t n, hipStream_t s);
int rk_sqnorm_launch(const float* a, const float* sub, const float* b, const float* b2, float atol, float rtol, int64_t n, double* out,
                     hipStream_t s);
int rk_interp_launch(float* out, const float* y0, const float* y1, const float* ym, const float* f0, const float* f1, float dt, float x,
                     int64_t n, hipStream_t s);
