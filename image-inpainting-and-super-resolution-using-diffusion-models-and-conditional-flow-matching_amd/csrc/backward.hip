// Data-gradient (vector-Jacobian product w.r.t. the INPUT image) building blocks of the U-Net, for the reconstruction-guidance
// sampler: the reference differentiates the x0 model through the network with vmap(grad(constraint))
// (AD/image_diffusion/sampling.py:136-163).  Weight gradients are never needed (inference only), so the backward pass is
//   conv  -> the SAME implicit-GEMM kernels on transposed / tap-flipped packed weights (conv_pack_weights_dgrad, unet_backward.hip)
//   GroupNorm32 (+ SiLU, + FiLM) prologue -> gn_silu_bwd_kernel (this file), statistics in fp32 like the forward
//   attention core -> attention_bwd.hip
//   nearest-up / avg-pool / stride-2 / residual adds / channel concat -> the small gather kernels below.
// Every kernel is HBM-bound elementwise / reduction work on NHWC tensors with 16-byte channel fragments.
#include "ops.h"

namespace {

template <typename T> __device__ __forceinline__ void ldfrag(const T* p, float (&f)[Elem<T>::VEC]) {
  frag_to_float(*reinterpret_cast<const u32x4*>(p), f, T());
}
template <typename T> __device__ __forceinline__ void stfrag(T* p, const float (&f)[Elem<T>::VEC]) {
  *reinterpret_cast<u32x4*>(p) = float_to_frag(f, T());
}

// ---- GroupNorm32 (+ SiLU) backward -------------------------------------------------------------------------------------------
// forward (gn_stats.hip + conv prologue): y = a x + b with a = rstd_g * gamma_eff, b = beta_eff - mean_g * a (FiLM folded), u = silu?(y).
// Given dU: dy = dU * silu'(y); w = dy * a (= rstd * dy * gamma_eff); xh = (x - mean) * rstd;
//   dx = w - mean_group(w) - xh * mean_group(w * xh)            (means over the group's cpg * HW elements of the image)
// One workgroup per image, two sweeps over (x, dU): per-channel partial sums -> groups -> apply (second sweep from L2).
struct GnBwdArgs {
  const void* x0; const void* x1; int C0, C1;
  const void* du; int du_stride;          // [N][HW][du_stride] elements, the C0 + C1 channels of interest start at channel 0
  int N, HW, groups, silu;
  const float* a; const float* b; const float* mean; const float* rstd;   // [N][C], [N][C], [N][groups], [N][groups]
  void* g0; void* g1; int acc0, acc1;     // gradient tensors of the two sources (NHWC, own channel counts); acc = add to what is there
};
constexpr int GB_THREADS = 512;

template <typename T>
__global__ void __launch_bounds__(GB_THREADS) gn_silu_bwd_kernel(GnBwdArgs p) {
  constexpr int V = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int C = p.C0 + p.C1, CV = C / V;
  const int ppi = GB_THREADS / CV;
  float* red_a = red;                       // [ppi][C]
  float* red_b = red + (size_t)ppi * C;     // [ppi][C]
  float* ch_a = red_b + (size_t)ppi * C;    // [C]
  float* ch_b = ch_a + C;                   // [C]
  float* g_a = ch_b + C;                    // [groups] mean(w)
  float* g_b = g_a + p.groups;              // [groups] mean(w * xh)
  const int tid = threadIdx.x, n = blockIdx.x;
  const int frag = tid % CV, prow = tid / CV;
  const int cb = frag * V, cpg = C / p.groups;
  const bool active = prow < ppi;
  const bool from0 = cb < p.C0;
  const int Cs = from0 ? p.C0 : p.C1, cl = from0 ? cb : cb - p.C0;
  const T* xp = (from0 ? reinterpret_cast<const T*>(p.x0) : reinterpret_cast<const T*>(p.x1)) + (size_t)n * p.HW * Cs + cl;
  const T* dp = reinterpret_cast<const T*>(p.du) + (size_t)n * p.HW * p.du_stride + cb;
  float av[V], bv[V], mu[V], rs[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = min(cb + j, C - 1), g = c / cpg;
    av[j] = p.a[(size_t)n * C + c]; bv[j] = p.b[(size_t)n * C + c];
    mu[j] = p.mean[(size_t)n * p.groups + g]; rs[j] = p.rstd[(size_t)n * p.groups + g];
  }
  auto wxh = [&](const float (&x)[V], const float (&d)[V], float (&w)[V], float (&xh)[V]) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float y = av[j] * x[j] + bv[j];
      float dy = d[j];
      if (p.silu) {
        const float s = 1.0f / (1.0f + expf(-y));
        dy *= s * (1.0f + y * (1.0f - s));
      }
      w[j] = dy * av[j];
      xh[j] = (x[j] - mu[j]) * rs[j];
    }
  };
  float sa[V], sb[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { sa[j] = 0.f; sb[j] = 0.f; }
  if (active) {
    for (int pix = prow; pix < p.HW; pix += ppi) {
      float x[V], d[V], w[V], xh[V];
      ldfrag(xp + (size_t)pix * Cs, x);
      ldfrag(dp + (size_t)pix * p.du_stride, d);
      wxh(x, d, w, xh);
#pragma unroll
      for (int j = 0; j < V; ++j) { sa[j] += w[j]; sb[j] += w[j] * xh[j]; }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red_a[prow * C + cb + j] = sa[j]; red_b[prow * C + cb + j] = sb[j]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GB_THREADS) {
    float ta = 0.f, tb = 0.f;
    for (int r = 0; r < ppi; ++r) { ta += red_a[r * C + c]; tb += red_b[r * C + c]; }
    ch_a[c] = ta; ch_b[c] = tb;
  }
  __syncthreads();
  if (tid < p.groups) {
    float ta = 0.f, tb = 0.f;
    for (int j = 0; j < cpg; ++j) { ta += ch_a[tid * cpg + j]; tb += ch_b[tid * cpg + j]; }
    const float inv = 1.0f / ((float)cpg * (float)p.HW);
    g_a[tid] = ta * inv; g_b[tid] = tb * inv;
  }
  __syncthreads();
  if (!active) return;
  float ma[V], mb[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { const int g = min(cb + j, C - 1) / cpg; ma[j] = g_a[g]; mb[j] = g_b[g]; }
  T* gp = (from0 ? reinterpret_cast<T*>(p.g0) : reinterpret_cast<T*>(p.g1)) + (size_t)n * p.HW * Cs + cl;
  const bool acc = from0 ? p.acc0 != 0 : p.acc1 != 0;
  for (int pix = prow; pix < p.HW; pix += ppi) {
    float x[V], d[V], w[V], xh[V], o[V];
    ldfrag(xp + (size_t)pix * Cs, x);
    ldfrag(dp + (size_t)pix * p.du_stride, d);
    wxh(x, d, w, xh);
    if (acc) ldfrag(gp + (size_t)pix * Cs, o);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float dx = w[j] - ma[j] - xh[j] * mb[j];
      o[j] = acc ? o[j] + dx : dx;
    }
    stfrag(gp + (size_t)pix * Cs, o);
  }
}

// ---- gather / accumulate ---------------------------------------------------------------------------------------------------------
// dst[n, y, x, c] (+)= scale * G(src)[n, y, x, coff + c], dst NHWC [N][Hd][Wd][Cd]; src row stride cs elements:
//   GA_SAME  G = identity                       (residual adds, skip-concat split, dgrad without prologue)
//   GA_POOL  G = sum of the 2x2 block           (backward of nearest x2: F.interpolate(..., mode="nearest"), unet.py:209)
//   GA_UP    G = src[y/2, x/2]                  (backward of AvgPool2d(2) with scale 1/4, unet.py:236)
//   GA_STUFF G = src[y/2, x/2] if y, x even and inside src, else 0 (zero insertion: stride-2 conv backward)
enum { GA_SAME = 0, GA_POOL = 1, GA_UP = 2, GA_STUFF = 3 };
struct GaArgs {
  void* dst; const void* src; int N, Hd, Wd, Cd, Hs, Ws, cs, coff, mode, accumulate; float scale;
};
template <typename T>
__global__ void __launch_bounds__(256) grad_gather_kernel(GaArgs p) {
  constexpr int V = Elem<T>::VEC;
  const int CV = p.Cd / V;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)p.N * p.Hd * p.Wd * CV) return;
  const int f = idx % CV;
  size_t r = idx / CV;
  const int x = r % p.Wd; r /= p.Wd;
  const int y = r % p.Hd;
  const size_t n = r / p.Hd;
  const T* s = reinterpret_cast<const T*>(p.src) + p.coff + f * V;
  float v[V];
#pragma unroll
  for (int j = 0; j < V; ++j) v[j] = 0.f;
  auto at = [&](int yy, int xx) { return s + ((n * p.Hs + yy) * p.Ws + xx) * p.cs; };
  if (p.mode == GA_SAME) {
    ldfrag(at(y, x), v);
  } else if (p.mode == GA_POOL) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t[V];
      ldfrag(at(2 * y + (k >> 1), 2 * x + (k & 1)), t);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += t[j];
    }
  } else if (p.mode == GA_UP) {
    ldfrag(at(y >> 1, x >> 1), v);
  } else {
    if (!(y & 1) && !(x & 1) && (y >> 1) < p.Hs && (x >> 1) < p.Ws) ldfrag(at(y >> 1, x >> 1), v);
  }
  T* d = reinterpret_cast<T*>(p.dst) + ((n * p.Hd + y) * p.Wd + x) * p.Cd + f * V;
  float o[V];
  if (p.accumulate) ldfrag(d, o);
#pragma unroll
  for (int j = 0; j < V; ++j) o[j] = p.accumulate ? o[j] + p.scale * v[j] : p.scale * v[j];
  stfrag(d, o);
}

// ---- reconstruction-guidance seed (sampling.py:154-163 + likelihoods.py:58-66,138-143 + sde_diffusion.py:220-224) --------------
// x0_pre = c_recip x - c_recipm1 eps; x0 = clip(x0_pre, -1, 1); g = dloss/dx0 of the PER-SAMPLE constraint:
//   painting   loss_n = sum over unmasked entries (x0 - cond)^2        -> g = 2 (x0 - cond) [cond != pad]
//   hyperres   loss_n = mean over the sample's entries (cond - x0)^2    -> g = 2 (x0 - cond) / per
// chain rule through the clip (pass-through on [-1, 1], like torch.clamp) and predict_start_from_noise:
//   g_eps = -c_recipm1 g_pre  (the cotangent fed to the U-Net VJP),  g_x = c_recip g_pre  (the direct path)
__global__ void __launch_bounds__(256) guidance_seed_kernel(const float* x, const float* eps, const float* cond, float c_recip, float c_recipm1,
                                                          int mode, float pad, float inv_per, float* g_eps, float* g_x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float pre = c_recip * x[i] - c_recipm1 * eps[i];
  const float x0 = clip_nan(pre, -1.f, 1.f);
  const float c = cond[i];
  float g = 2.0f * (x0 - c);
  if (mode == 0) g = c == pad ? 0.f : g;
  else g *= inv_per;
  if (!(pre >= -1.f && pre <= 1.f)) g = 0.f;
  g_eps[i] = -c_recipm1 * g;
  g_x[i] = c_recip * g;
}

// x_grad = g_x + vjp; update = -scale * x_grad; optionally x += update ("before"); update is kept for the "after" rule
__global__ void __launch_bounds__(256) guidance_update_kernel(float* x, const float* g_x, const float* vjp, float scale, int apply, float* update,
                                                            int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float u = -scale * (g_x[i] + vjp[i]);
  update[i] = u;
  if (apply) x[i] = x[i] + u;
}

template <typename T>
__global__ void unpack_channels_kernel(const T* in, int N, int HW, int stride, int c0, int count, float* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * count * HW) return;
  const int hw = idx % HW;
  const size_t r = idx / HW;
  const int c = r % count;
  const size_t n = r / count;
  out[idx] = (float)in[(n * HW + hw) * stride + c0 + c];
}

}  // namespace

int unpack_channels_launch(int dtype, const void* in, int N, int HW, int stride, int c0, int count, float* out, hipStream_t s) {
  const size_t total = (size_t)N * count * HW;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == 0) hipLaunchKernelGGL(unpack_channels_kernel<float>, grid, dim3(256), 0, s, (const float*)in, N, HW, stride, c0, count, out);
  else hipLaunchKernelGGL(unpack_channels_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, N, HW, stride, c0, count, out);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int gn_silu_bwd_launch(const GnBwdDesc& d, hipStream_t stream) {
  const int C = d.C0 + d.C1;
  const int V = d.dtype == 0 ? 4 : 8;
  MI355_REQUIRE(C % d.groups == 0 && d.C0 % V == 0 && d.C1 % V == 0 && d.du_stride % V == 0, -2, "gn backward: channel fragments");
  MI355_REQUIRE(C / V <= GB_THREADS && d.groups <= GB_THREADS, -4, "gn backward: too many channels");
  GnBwdArgs a{d.x0, d.x1, d.C0, d.C1, d.du, d.du_stride, d.N, d.HW, d.groups, d.silu, d.a, d.b, d.mean, d.rstd, d.g0, d.g1, d.acc0, d.acc1};
  const int ppi = GB_THREADS / (C / V);
  const size_t lds = ((size_t)2 * ppi * C + 2 * C + 2 * d.groups) * sizeof(float);
  if (d.dtype == 0) {
    if (lds > 64 * 1024) { if (int rc = mi355_allow_big_lds(gn_silu_bwd_kernel<float>, "gn backward")) return rc; }
    hipLaunchKernelGGL(gn_silu_bwd_kernel<float>, dim3(d.N), dim3(GB_THREADS), lds, stream, a);
  } else {
    if (lds > 64 * 1024) { if (int rc = mi355_allow_big_lds(gn_silu_bwd_kernel<bf16>, "gn backward")) return rc; }
    hipLaunchKernelGGL(gn_silu_bwd_kernel<bf16>, dim3(d.N), dim3(GB_THREADS), lds, stream, a);
  }
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int grad_gather_launch(int dtype, void* dst, const void* src, int N, int Hd, int Wd, int Cd, int Hs, int Ws, int cs, int coff, int mode,
                       int accumulate, float scale, hipStream_t s) {
  const int V = dtype == 0 ? 4 : 8;
  MI355_REQUIRE(Cd % V == 0 && cs % V == 0 && coff % V == 0, -2, "grad gather: channel fragments");
  GaArgs a{dst, src, N, Hd, Wd, Cd, Hs, Ws, cs, coff, mode, accumulate, scale};
  const size_t total = (size_t)N * Hd * Wd * (Cd / V);
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == 0) hipLaunchKernelGGL(grad_gather_kernel<float>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(grad_gather_kernel<bf16>, grid, dim3(256), 0, s, a);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int guidance_seed_launch(const float* x, const float* eps, const float* cond, float c_recip, float c_recipm1, int mode, float pad,
                         int64_t per, float* g_eps, float* g_x, int64_t n, hipStream_t s) {
  MI355_REQUIRE(x && eps && cond && g_eps && g_x && n > 0 && per > 0, -1, "guidance_seed: bad argument");
  hipLaunchKernelGGL(guidance_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, eps, cond, c_recip, c_recipm1, mode, pad,
                     1.0f / (float)per, g_eps, g_x, n);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int guidance_update_launch(float* x, const float* g_x, const float* vjp, float scale, int apply, float* update, int64_t n, hipStream_t s) {
  MI355_REQUIRE(x && g_x && vjp && update && n > 0, -1, "guidance_update: bad argument");
  hipLaunchKernelGGL(guidance_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, g_x, vjp, scale, apply, update, n);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
