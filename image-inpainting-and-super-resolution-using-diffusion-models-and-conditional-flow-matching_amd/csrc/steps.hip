// Per-step sampler updates: HBM-bound fused elementwise kernels over the fp32 NCHW state.
//
// Reference (each is 8-10 separate eager kernels there):
//   Euler update                x += dt*v                  torchdyn fixed-step euler (cifar10/compute_fid.py:79)
//   DDPM ancestral step         sampling.py:59-67 + sde_diffusion.py:220-237
//   Langevin corrector          sampling.py:113-121 + sde_diffusion.py:214-217
//   Replacement mask            sampling.py:225-232 + sde_diffusion.py:239-244
//   clip / uint8 / unit range   sampling.py:13-14, cifar10/compute_fid.py:87, cifar10/utils_cifar.py:40-41
// Noise is either injected (a pre-drawn tensor, for parity with the reference's torch.randn_like
// stream) or generated in-kernel with Philox4x32-10 + Box-Muller keyed by (seed, element index).
#include "ops.h"

// torch evaluates a*x - b*e as mul, mul, sub (three roundings); keep that shape so the ill-conditioned
// x0 = c_recip*x - c_recipm1*eps (c ~ 1e3 at high noise levels) tracks the reference bit-for-bit where possible.
#pragma clang fp contract(off)

namespace {

struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c[0] = hi1 ^ c[1] ^ k0; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k1; c[3] = lo0;
  }
  __device__ static inline void gen(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) { round(c, k0, k1); k0 += W0; k1 += W1; }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
};

// 4 standard normals for counter `ctr` (elements 4*ctr .. 4*ctr+3 of the stream)
__device__ inline void randn4(uint64_t seed, uint64_t ctr, float (&z)[4]) {
  uint32_t r[4];
  Philox::gen(seed, ctr, r);
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r[0] + 0.5f) * k, u1 = ((float)r[1] + 0.5f) * k;
  const float u2 = ((float)r[2] + 0.5f) * k, u3 = ((float)r[3] + 0.5f) * k;
  const float ra = sqrtf(-2.0f * logf(fminf(fmaxf(u0, 1e-12f), 1.0f))), rb = sqrtf(-2.0f * logf(fminf(fmaxf(u2, 1e-12f), 1.0f)));
  const float ta = 6.283185307179586f * u1, tb = 6.283185307179586f * u3;
  z[0] = ra * cosf(ta); z[1] = ra * sinf(ta); z[2] = rb * cosf(tb); z[3] = rb * sinf(tb);
}

// Every kernel below handles 4 consecutive elements per thread (16-B accesses) with a scalar tail.
template <typename F>
__global__ void __launch_bounds__(256) ew4_kernel(int64_t n, F f) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x);
  const int64_t i = i4 * 4;
  if (i >= n) return;
  f(i, i4, (int)((n - i) < 4 ? (n - i) : 4));
}

template <typename F>
int launch_ew4(int64_t n, hipStream_t s, F f) {
  if (n <= 0) return 0;
  const int64_t nthreads = (n + 3) / 4;
  hipLaunchKernelGGL(ew4_kernel<F>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, n, f);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

__device__ inline void ld4(const float* p, int64_t i, int cnt, float (&v)[4]) {
  if (cnt == 4 && ((reinterpret_cast<uintptr_t>(p + i) & 15) == 0)) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p + i);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    for (int j = 0; j < 4; ++j) v[j] = j < cnt ? p[i + j] : 0.f;
  }
}
__device__ inline void st4(float* p, int64_t i, int cnt, const float (&v)[4]) {
  if (cnt == 4 && ((reinterpret_cast<uintptr_t>(p + i) & 15) == 0)) {
    *reinterpret_cast<f32x4*>(p + i) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    for (int j = 0; j < cnt; ++j) p[i + j] = v[j];
  }
}
__device__ inline void noise4(const float* z, int use_philox, uint64_t seed, uint64_t offset4, int64_t i, int64_t i4, int cnt,
                              float (&zz)[4]) {
  if (z) ld4(z, i, cnt, zz);
  else if (use_philox) randn4(seed, offset4 + (uint64_t)i4, zz);
  else { zz[0] = zz[1] = zz[2] = zz[3] = 0.f; }
}

}  // namespace

int euler_step_launch(float* x, const float* v, float dt, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float a[4], b[4];
    ld4(x, i, cnt, a); ld4(v, i, cnt, b);
    {
#pragma clang fp contract(off)   // product and sum rounded separately, as `x + dt * v` in eager PyTorch (and as conv_edge.hip's fused form of this step)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = a[j] + dt * b[j];
    }
    st4(x, i, cnt, a);
  });
}

int ddpm_step_launch(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float coef1, float coef2,
                     float sigma, int use_philox, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s) {
  const uint64_t off4 = offset / 4;
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t i4, int cnt) {
    float xv[4], ev[4], zz[4];
    ld4(x, i, cnt, xv); ld4(eps, i, cnt, ev);
    noise4(z, use_philox, seed, off4, i, i4, cnt, zz);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = clip_nan(c_recip * xv[j] - c_recipm1 * ev[j], -1.f, 1.f);  // predict_start_from_noise + process_x0
      const float mean = coef1 * x0 + coef2 * xv[j];                                // q_posterior
      xv[j] = mean + sigma * zz[j];                                                 // exp(0.5*logvar) * noise
    }
    st4(x, i, cnt, xv);
  });
}

int corrector_step_launch(float* x, const float* eps, const float* z, float c_recip, float c_recipm1, float rsm1, float dt,
                          float delta, int use_philox, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s) {
  const uint64_t off4 = offset / 4;
  const float kd = 0.5f * dt * delta, kn = sqrtf(dt * delta);
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t i4, int cnt) {
    float xv[4], ev[4], zz[4];
    ld4(x, i, cnt, xv); ld4(eps, i, cnt, ev);
    noise4(z, use_philox, seed, off4, i, i4, cnt, zz);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = clip_nan(c_recip * xv[j] - c_recipm1 * ev[j], -1.f, 1.f);
      const float score = -rsm1 * x0;  // score_from_x0
      xv[j] = xv[j] + (kd * score + kn * zz[j]);
    }
    st4(x, i, cnt, xv);
  });
}

int ddim_step_launch(float* x, const float* eps, float c_recip, float c_recipm1, float acp_prev, int64_t n, hipStream_t s) {
  const float sa = sqrtf(acp_prev), sb = sqrtf(1.0f - acp_prev);
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float xv[4], ev[4];
    ld4(x, i, cnt, xv); ld4(eps, i, cnt, ev);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = clip_nan(c_recip * xv[j] - c_recipm1 * ev[j], -1.f, 1.f);
      const float e2 = (c_recip * xv[j] - x0) / c_recipm1;
      xv[j] = sa * x0 + sb * e2;
    }
    st4(x, i, cnt, xv);
  });
}

int replace_mask_launch(float* x, const float* cond, const float* z, float pad, int noisy, float sa, float sb, int use_philox,
                        uint64_t seed, uint64_t offset, int64_t n, hipStream_t s) {
  const uint64_t off4 = offset / 4;
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t i4, int cnt) {
    float xv[4], cv[4], zz[4] = {0.f, 0.f, 0.f, 0.f};
    ld4(x, i, cnt, xv); ld4(cond, i, cnt, cv);
    if (noisy) noise4(z, use_philox, seed, off4, i, i4, cnt, zz);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float nc = noisy ? sa * cv[j] + sb * zz[j] : cv[j];  // q_sample(condition, i)
      xv[j] = cv[j] == pad ? xv[j] : nc;                         // torch.where(condition == pad_value, xi, noised)
    }
    st4(x, i, cnt, xv);
  });
}

int clip_launch(float* x, float lo, float hi, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float a[4];
    ld4(x, i, cnt, a);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = clip_nan(a[j], lo, hi);
    st4(x, i, cnt, a);
  });
}

// Per-sample mean squared error, torch.mean((a - b)**2, dim=(1, 2, 3)) of the evaluation loop (AD/experiments/main.py:299):
// one workgroup per sample, fp32 differences, fp64 accumulation.
__global__ void __launch_bounds__(256) mse_per_sample_kernel(const float* a, const float* b, float* out, int64_t per) {
  __shared__ double sh[4];
  const float* pa = a + (int64_t)blockIdx.x * per;
  const float* pb = b + (int64_t)blockIdx.x * per;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < per; i += 256) { const float d = pa[i] - pb[i]; acc += (double)d * (double)d; }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)((sh[0] + sh[1] + sh[2] + sh[3]) / (double)per);
}
int mse_per_sample_launch(const float* a, const float* b, float* out, int batch, int64_t per, hipStream_t s) {
  MI355_REQUIRE(batch > 0 && per > 0, -2, "mse_per_sample: empty input");
  hipLaunchKernelGGL(mse_per_sample_kernel, dim3((unsigned)batch), dim3(256), 0, s, a, b, out, per);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

// EMA of a parameter tensor: target = target * decay + source * (1 - decay)  (cifar10/utils_cifar.py:47-53), two separately
// rounded products like the reference's eager expression (this file is compiled with fp contraction off).
int ema_update_launch(float* target, const float* source, float decay, float one_minus_decay, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float t[4], v[4];
    ld4(target, i, cnt, t);
    ld4(source, i, cnt, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = t[j] * decay + v[j] * one_minus_decay;
    st4(target, i, cnt, t);
  });
}

int quantize_u8_launch(const float* x, uint8_t* out, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float a[4];
    ld4(x, i, cnt, a);
    uint8_t q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = clip_nan(a[j] * 127.5f + 128.f, 0.f, 255.f);
      q[j] = (uint8_t)(int)v;  // .to(uint8): truncation toward zero
    }
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(out + i) & 3) == 0)) {
      *reinterpret_cast<uint32_t*>(out + i) = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    } else {
      for (int j = 0; j < cnt; ++j) out[i + j] = q[j];
    }
  });
}

int to_unit_range_launch(const float* x, float* out, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float a[4];
    ld4(x, i, cnt, a);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = clip_nan(a[j], -1.f, 1.f) / 2.f + 0.5f;
    st4(out, i, cnt, a);
  });
}

int randn_launch(float* out, uint64_t seed, uint64_t offset, int64_t n, hipStream_t s) {
  const uint64_t off4 = offset / 4;
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t i4, int cnt) {
    float zz[4];
    randn4(seed, off4 + (uint64_t)i4, zz);
    st4(out, i, cnt, zz);
  });
}

int fill_launch(float* x, float v, int64_t n, hipStream_t s) {
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float a[4] = {v, v, v, v};
    st4(x, i, cnt, a);
  });
}

// out[n, :] = a[n] * x[n, :] + b[n] * y[n, :] with per-sample coefficients (device arrays gathered from the DDPM tables by
// `extract`, sde_diffusion.py:101-104): predict_start_from_noise / q_posterior mean / q_sample / score_from_x0
// (sde_diffusion.py:214-244).  Two separately rounded products and one add, like the reference's eager expression; y (and b)
// may be null (out = a * x).  `per` = elements per sample.
int lincomb_per_sample_launch(float* out, const float* x, const float* y, const float* a, const float* b, int batch, int64_t per,
                              hipStream_t s) {
  MI355_REQUIRE(out && x && a && batch > 0 && per > 0, -1, "lincomb_per_sample: bad argument");
  MI355_REQUIRE((y == nullptr) == (b == nullptr), -1, "lincomb_per_sample: y and b go together");
  const int64_t n = (int64_t)batch * per;
  return launch_ew4(n, s, [=] __device__(int64_t i, int64_t, int cnt) {
    float xv[4], yv[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
    ld4(x, i, cnt, xv);
    if (y) ld4(y, i, cnt, yv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t smp = (i + j) / per;   // a 4-element group may straddle two samples when per % 4 != 0
      const int64_t sc = smp < batch ? smp : batch - 1;
      o[j] = y ? a[sc] * xv[j] + b[sc] * yv[j] : a[sc] * xv[j];
    }
    st4(out, i, cnt, o);
  });
}

namespace {
// ---- condition builders (once per batch / once per solve; AD/image_diffusion/likelihoods.py, mnist/utils_mnist_hy.py:18-28) ----------
// Bilinear resize of NCHW fp32 planes, align_corners = False, no antialiasing: torch.nn.functional.interpolate(mode="bilinear")
// as the reference calls it for the low-res condition (downsample_images, HyperResolution._sample, the SuperRes wrapper's
// upsampling).  Source index = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out in fp32; the four taps are combined as
// h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11), the operation order of ATen's kernel (no fused multiply-adds here).
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* in, float* out, int64_t planes, int Hi, int Wi, int Ho, int Wo,
                                                               float sh, float sw) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= planes * Ho * Wo) return;
  const int ox = (int)(idx % Wo);
  const int oy = (int)((idx / Wo) % Ho);
  const int64_t pl = idx / ((int64_t)Wo * Ho);
  float fy = sh * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sw * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  const float h1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), w1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
  const float h0 = 1.f - h1, w0 = 1.f - w1;
  const float* p = in + pl * (int64_t)Hi * Wi;
  const float v00 = p[(int64_t)y0 * Wi + x0], v01 = p[(int64_t)y0 * Wi + x1], v10 = p[(int64_t)y1 * Wi + x0], v11 = p[(int64_t)y1 * Wi + x1];
  out[idx] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
}

// InPainting._sample / OutPainting._sample (likelihoods.py:78-87, 95-104) for a whole batch in one launch: image n's square patch
// has its top-left corner at (top[n], left[n]) (drawn on the host in the reference's order); inside the patch the result is
// pad_value (in-painting) or the image (out-painting), outside it the other one.
__global__ void __launch_bounds__(256) paint_patch_kernel(const float* img, const int* top, const int* left, int patch, float pad,
                                                           int outpaint, float* out, int64_t n_elems, int C, int H, int W) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_elems) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H);
  const int64_t n = idx / ((int64_t)W * H * C);
  const int t = top[n], l = left[n];
  const bool inside = y >= t && y < t + patch && x >= l && x < l + patch;
  out[idx] = (inside != (outpaint != 0)) ? pad : img[idx];
}

}  // namespace

int resize_bilinear_launch(const float* in, float* out, int64_t planes, int Hi, int Wi, int Ho, int Wo, hipStream_t s) {
  MI355_REQUIRE(in && out && planes > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, -1, "resize_bilinear: bad argument");
  const int64_t n = planes * Ho * Wo;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, planes, Hi, Wi, Ho, Wo,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int paint_patch_launch(const float* img, const int* top, const int* left, int patch, float pad, int outpaint, float* out, int N, int C,
                       int H, int W, hipStream_t s) {
  MI355_REQUIRE(img && top && left && out && N > 0 && C > 0 && H > 0 && W > 0 && patch > 0, -1, "paint_patch: bad argument");
  const int64_t n = (int64_t)N * C * H * W;
  hipLaunchKernelGGL(paint_patch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, top, left, patch, pad, outpaint, out, n, C, H, W);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
