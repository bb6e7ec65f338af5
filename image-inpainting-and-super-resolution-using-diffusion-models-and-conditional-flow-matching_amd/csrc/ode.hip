// Fused elementwise / reduction kernels of the adaptive Dormand-Prince 5(4) integrator (the reference's default FID
// solver: torchdiffeq.odeint(..., method="dopri5"), cifar10/compute_fid.py:80-85, mnist/utils_mnist.py:101-108).
// The step-size controller itself runs on the host (it needs one scalar per step anyway); these kernels replace the
// dozens of eager elementwise launches torchdiffeq issues per step: stage combination, error norm, dense output.
#include "ops.h"

namespace {

struct KPtrs { const float* k[7]; float c[7]; };

__global__ void __launch_bounds__(256) rk_combine_kernel(float* out, const float* y0, KPtrs kp, int nk, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nk; ++j) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kp.k[j] + i);
      const float c = kp.c[j];
      acc = f32x4{acc[0] + kv[0] * c, acc[1] + kv[1] * c, acc[2] + kv[2] * c, acc[3] + kv[3] * c};
    }
    f32x4 y = y0 ? *reinterpret_cast<const f32x4*>(y0 + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(out + i) = f32x4{y[0] + acc[0], y[1] + acc[1], y[2] + acc[2], y[3] + acc[3]};
  } else {
    for (int64_t e = i; e < n; ++e) {
      float acc = 0.f;
      for (int j = 0; j < nk; ++j) acc += kp.k[j][e] * kp.c[j];
      out[e] = (y0 ? y0[e] : 0.f) + acc;
    }
  }
}

// sum_i ((a_i - sub_i) / (atol + rtol * max(|b_i|, |b2_i|)))^2  accumulated into *out (fp64 atomics, one per block)
__global__ void __launch_bounds__(256) rk_sqnorm_kernel(const float* a, const float* sub, const float* b, const float* b2, float atol,
                                                      float rtol, int64_t n, double* out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float num = a[i] - (sub ? sub[i] : 0.f);
    float mag = b ? fabsf(b[i]) : 0.f;
    if (b2) mag = fmaxf(mag, fabsf(b2[i]));
    const float r = num / (atol + rtol * mag);
    acc += (double)r * (double)r;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// torchdiffeq dense output (interp.py _interp_fit / _interp_evaluate): quartic through y0, y_mid, y1 with end slopes f0, f1
__global__ void __launch_bounds__(256) rk_interp_kernel(float* out, const float* y0, const float* y1, const float* ym, const float* f0,
                                                      const float* f1, float dt, float x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = 2.f * dt * (f1[i] - f0[i]) - 8.f * (y1[i] + y0[i]) + 16.f * ym[i];
  const float b = dt * (5.f * f0[i] - 3.f * f1[i]) + 18.f * y0[i] + 14.f * y1[i] - 32.f * ym[i];
  const float c = dt * (f1[i] - 4.f * f0[i]) - 11.f * y0[i] - 5.f * y1[i] + 16.f * ym[i];
  const float d = dt * f0[i], e = y0[i];
  out[i] = (((a * x + b) * x + c) * x + d) * x + e;
}

}  // namespace

int rk_combine_launch(float* out, const float* y0, const float* const* k, const float* c, int nk, int64_t n, hipStream_t s) {
  MI355_REQUIRE(out && nk >= 0 && nk <= 7, -1, "rk_combine: bad argument");
  KPtrs kp;
  for (int j = 0; j < 7; ++j) { kp.k[j] = j < nk ? k[j] : nullptr; kp.c[j] = j < nk ? c[j] : 0.f; }
  const int64_t nth = (n + 3) / 4;
  hipLaunchKernelGGL(rk_combine_kernel, dim3((unsigned)((nth + 255) / 256)), dim3(256), 0, s, out, y0, kp, nk, n);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int rk_sqnorm_launch(const float* a, const float* sub, const float* b, const float* b2, float atol, float rtol, int64_t n, double* out,
                     hipStream_t s) {
  MI355_REQUIRE(a && out, -1, "rk_sqnorm: null argument");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rk_sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, sub, b, b2, atol, rtol, n, out);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int rk_interp_launch(float* out, const float* y0, const float* y1, const float* ym, const float* f0, const float* f1, float dt, float x,
                     int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(rk_interp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, y0, y1, ym, f0, f1, dt, x, n);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
