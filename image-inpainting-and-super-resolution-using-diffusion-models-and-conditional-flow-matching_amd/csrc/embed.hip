// Sinusoidal timestep embedding + the small fp32 linears around it, and NCHW<->NHWC boundary packing.
//
// Reference: timestep_embedding (AD/image_diffusion/nn.py:97-115), UNetModel.time_embed
// (unet.py:564-569: Linear, SiLU, Linear) and every ResBlock's emb_layers (unet.py:297-305: SiLU,
// Linear) - the latter are batched into ONE linear over the concatenated output channels of all
// ResBlocks per forward.  All fp32 (the reference forces fp32 here, nn.py:111).
#include "ops.h"

namespace {

__global__ void timestep_embedding_kernel(const float* t, int B, int dim, float max_period, float* out) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * dim) return;
  const int b = idx / dim, j = idx % dim;
  float v = 0.f;  // zero pad when dim is odd (nn.py:113-114)
  if (j < 2 * half) {
    const int k = j < half ? j : j - half;
    // freqs = exp(-ln(max_period) * k / half), fp32 like the reference
    const float freq = expf(-logf(max_period) * (float)k / (float)half);
    const float arg = t[b] * freq;
    v = j < half ? cosf(arg) : sinf(arg);
  }
  out[idx] = v;
}

// out[b][j] = bias[j] + sum_k act(in[b][k]) * Wt[k][j].  One thread per (j, 8-row batch slab): the 8 input rows
// are staged in LDS (broadcast reads), Wt reads are coalesced across lanes, 4 independent loads in flight.
constexpr int LB = 8;
__global__ void __launch_bounds__(128) linear_kernel(const float* __restrict__ in, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B, int K, int J,
                                                     int in_act, int out_act) {
  extern __shared__ float xin[];  // [LB][K]
  const int j = blockIdx.x * 128 + threadIdx.x;
  const int b0 = blockIdx.y * LB;
  for (int i = threadIdx.x; i < LB * K; i += 128) {
    const int r = i / K, k = i - r * K;
    float x = b0 + r < B ? in[(size_t)(b0 + r) * K + k] : 0.f;
    xin[i] = in_act ? silu_f<false>(x) : x;
  }
  __syncthreads();
  if (j >= J) return;
  float acc[LB];
#pragma unroll
  for (int r = 0; r < LB; ++r) acc[r] = 0.f;
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    const float w0 = Wt[(size_t)k * J + j], w1 = Wt[(size_t)(k + 1) * J + j];
    const float w2 = Wt[(size_t)(k + 2) * J + j], w3 = Wt[(size_t)(k + 3) * J + j];
#pragma unroll
    for (int r = 0; r < LB; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xin + r * K + k);
      acc[r] = fmaf(x[3], w3, fmaf(x[2], w2, fmaf(x[1], w1, fmaf(x[0], w0, acc[r]))));
    }
  }
  for (; k < K; ++k) {
    const float w = Wt[(size_t)k * J + j];
#pragma unroll
    for (int r = 0; r < LB; ++r) acc[r] = fmaf(xin[r * K + k], w, acc[r]);
  }
  const float bj = bias ? bias[j] : 0.f;
#pragma unroll
  for (int r = 0; r < LB; ++r) {
    const int b = b0 + r;
    if (b < B) {
      float v = acc[r] + bj;
      out[(size_t)b * J + j] = out_act ? silu_f<false>(v) : v;
    }
  }
}

// Small-batch form (the sampler loops share one step time, so B = 1): a GEMV is latency-bound, so the K range is split
// over the 4 waves of a workgroup (lanes own 64 consecutive outputs, 8 independent loads in flight per lane) and the
// partial sums meet in LDS.
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ in, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias, float* __restrict__ out, int B, int K, int J,
                                                           int in_act, int out_act) {
  __shared__ float part[4][LB][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int kq = K / 4, k0 = wave * kq;
  float acc[LB];
#pragma unroll
  for (int r = 0; r < LB; ++r) acc[r] = 0.f;
  if (j < J) {
    for (int k = k0; k < k0 + kq; k += 8) {
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = Wt[(size_t)(k + u) * J + j];
#pragma unroll
      for (int r = 0; r < LB; ++r) {
        if (r < B) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float x = in[(size_t)r * K + k + u];
            if (in_act) x = silu_f<false>(x);
            acc[r] = fmaf(x, w[u], acc[r]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LB; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave == 0 && j < J) {
    const float bj = bias ? bias[j] : 0.f;
    for (int r = 0; r < B; ++r) {
      const float v = part[0][r][lane] + part[1][r][lane] + part[2][r][lane] + part[3][r][lane] + bj;
      out[(size_t)r * J + j] = out_act ? silu_f<false>(v) : v;
    }
  }
}

// One thread = one 16-byte channel fragment of one pixel (a pixel's fragments are adjacent threads: full-line NHWC stores; the
// NCHW reads of a channel are contiguous across the pixels of neighbouring thread groups).  Cpad is a multiple of the fragment.
template <typename T>
__global__ void __launch_bounds__(256) pack_nhwc_kernel(const float* x, int Cx, const float* cond, int Cc, int N, int HW, int Cpad, T* out) {
  constexpr int V = Elem<T>::VEC;
  const int G = Cpad / V;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * HW * G) return;
  const int g = idx % G;
  const size_t pix = idx / G;
  const int hw = pix % HW;
  const size_t n = pix / HW;
  float v[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int c = g * V + j;
    v[j] = c < Cx ? x[(n * Cx + c) * HW + hw] : (c < Cx + Cc ? cond[(n * Cc + (c - Cx)) * HW + hw] : 0.f);
  }
  *reinterpret_cast<u32x4*>(out + pix * Cpad + g * V) = float_to_frag(v, T());
}

template <typename T>
__global__ void unpack_nchw_kernel(const T* in, int N, int HW, int C, float* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * HW * C;
  if (idx >= total) return;
  const int hw = idx % HW;
  const size_t r = idx / HW;
  const int c = r % C;
  const size_t n = r / C;
  out[idx] = (float)in[(n * HW + hw) * C + c];
}

template <typename T>
__global__ void resample_kernel(const T* in, T* out, int N, int Hs, int Ws, int C, int mode) {
  const int Ho = mode == CONV_UP2 ? Hs * 2 : Hs / 2, Wo = mode == CONV_UP2 ? Ws * 2 : Ws / 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * Ho * Wo * C;
  if (idx >= total) return;
  const int c = idx % C;
  size_t r = idx / C;
  const int x = r % Wo; r /= Wo;
  const int y = r % Ho;
  const size_t n = r / Ho;
  float v;
  if (mode == CONV_UP2) {
    v = (float)in[((n * Hs + (y >> 1)) * Ws + (x >> 1)) * C + c];
  } else {
    const size_t b = ((n * Hs + 2 * y) * Ws + 2 * x) * C + c;
    v = 0.25f * (((float)in[b] + (float)in[b + C]) + ((float)in[b + (size_t)Ws * C] + (float)in[b + (size_t)Ws * C + C]));
  }
  out[idx] = (T)v;
}

inline dim3 grid1d(size_t total, int block) { return dim3((unsigned)((total + block - 1) / block)); }

}  // namespace

int timestep_embedding_launch(const float* t, int B, int dim, float max_period, float* out, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embedding_kernel, grid1d((size_t)B * dim, 256), dim3(256), 0, s, t, B, dim, max_period, out);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int linear_launch(const float* in, const float* Wt, const float* bias, float* out, int B, int K, int J, int in_act, int out_act,
                  hipStream_t s) {
  if (B <= LB && K % 32 == 0) {
    hipLaunchKernelGGL(linear_small_kernel, dim3((J + 63) / 64), dim3(256), 0, s, in, Wt, bias, out, B, K, J, in_act, out_act);
    MI355_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 grid((J + 127) / 128, (B + LB - 1) / LB);
  MI355_REQUIRE(K % 4 == 0 && (size_t)LB * K * 4 <= 64 * 1024, -4, "linear: K must be a multiple of 4 and <= 2048");
  hipLaunchKernelGGL(linear_kernel, grid, dim3(128), (size_t)LB * K * 4, s, in, Wt, bias, out, B, K, J, in_act, out_act);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int pack_nhwc_launch(int dtype, const float* x, int Cx, const float* cond, int Cc, int N, int HW, int Cpad, void* out,
                     hipStream_t s) {
  const int V = dtype == 0 ? 4 : 8;
  MI355_REQUIRE(Cpad % V == 0, -2, "pack_nhwc: padded channels must be whole 16-byte fragments");
  const size_t total = (size_t)N * HW * (Cpad / V);
  dispatch_dtype(dtype, [&](auto t) { using T = decltype(t); hipLaunchKernelGGL(pack_nhwc_kernel<T>, grid1d(total, 256), dim3(256), 0, s, x, Cx, cond, Cc, N, HW, Cpad, (T*)out); return 0; });
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int unpack_nchw_launch(int dtype, const void* in, int N, int HW, int C, float* out, hipStream_t s) {
  const size_t total = (size_t)N * HW * C;
  dispatch_dtype(dtype, [&](auto t) { using T = decltype(t); hipLaunchKernelGGL(unpack_nchw_kernel<T>, grid1d(total, 256), dim3(256), 0, s, (const T*)in, N, HW, C, out); return 0; });
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int resample_launch(int dtype, const void* in, void* out, int N, int Hs, int Ws, int C, int mode, hipStream_t s) {
  const int Ho = mode == CONV_UP2 ? Hs * 2 : Hs / 2, Wo = mode == CONV_UP2 ? Ws * 2 : Ws / 2;
  const size_t total = (size_t)N * Ho * Wo * C;
  dispatch_dtype(dtype, [&](auto t) { using T = decltype(t); hipLaunchKernelGGL(resample_kernel<T>, grid1d(total, 256), dim3(256), 0, s, (const T*)in, (T*)out, N, Hs, Ws, C, mode); return 0; });
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
