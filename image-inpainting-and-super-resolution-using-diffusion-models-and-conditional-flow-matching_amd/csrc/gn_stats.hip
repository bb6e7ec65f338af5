// GroupNorm32 statistics -> per-(image, channel) affine  y = a*x + b  consumed by the conv prologue.
//
// Reference: GroupNorm32 / normalization (AD/image_diffusion/nn.py:11-13,87-94): nn.GroupNorm(32, C),
// eps 1e-5, statistics in fp32.  FiLM (use_scale_shift_norm, unet.py:343-347) folds into a, b:
//   GN(x)*(1+scale)+shift = (a*(1+scale))*x + (b*(1+scale)+shift).
// The input may be the never-materialised channel concat of two NHWC tensors (unet.py:725).
//
// HBM-bound: one pass over the activation, 16 B per lane, a workgroup per image; lanes keep
// per-channel fp32 partial sums (a 16-B fragment = 4 or 8 fixed channels), reduced through LDS.
#include "ops.h"

namespace {

struct GnKArgs {
  const void* src0; const void* src1; int C0, C1;
  int N, HW, groups; float eps;
  const float* gamma; const float* beta; const float* film; int film_stride;
  float* a; float* b;
  void* y; int y_silu;   // optional: also write silu?(a*x + b) as one NHWC tensor of C0 + C1 channels (small images: see gn_affine_launch)
  float* mean; float* rstd;   // optional [N][groups]
  const void* warm; uint32_t warm_bytes;   // L2 warm-up of the consumer conv's weights by one extra wave (common.h l2_warm_wave)
};

constexpr int GN_THREADS = 512;

// NL > 0: the small-image form - the image is exactly NL 16-byte fragments per thread (HW * C / V == GN_THREADS * NL), all of them
// are loaded at once and STAY in registers, so the apply pass needs no second read (one memory latency instead of four in a row;
// the 8x8 / 4x4 concat sites: 12.5 -> ~6 us).  NL == 0: any size, the apply pass re-reads the image from L2.
template <typename T, int NL>
__global__ void __launch_bounds__(GN_THREADS + 64) gn_affine_kernel(GnKArgs p) {
  constexpr int V = Elem<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int C = p.C0 + p.C1, CV = C / V;
  const int ppi = GN_THREADS / CV;  // pixels handled per sweep
  float* red_s = red;                       // [ppi][C]
  float* red_q = red + (size_t)ppi * C;     // [ppi][C]
  float* ch_s = red_q + (size_t)ppi * C;    // [C]
  float* ch_q = ch_s + C;                   // [C]
  float* g_mean = ch_q + C;                 // [groups]
  float* g_rstd = g_mean + p.groups;
  const int tid = threadIdx.x, n = blockIdx.x;
  if (tid >= GN_THREADS) { l2_warm_wave(p.warm, p.warm_bytes); return; }   // the extra wave (launched only when there is something to warm)
  const int frag = tid % CV, prow = tid / CV;
  const int cb = frag * V;
  // this thread's channel parameters travel with the activation reads (they do not depend on the statistics)
  float pg, pbt, psc = 0.f, psh = 0.f;
  {
    const int c = min(tid, C - 1);
    pg = p.gamma[c]; pbt = p.beta[c];
    if (p.film) { psc = p.film[(size_t)n * p.film_stride + c]; psh = p.film[(size_t)n * p.film_stride + C + c]; }
  }
  float s[V], q[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s[j] = 0.f; q[j] = 0.f; }
  u32x4 keep[NL > 0 ? NL : 1];
  if constexpr (NL > 0) {   // every thread has work (prow < ppi always): NL fragments at pixels prow + i ppi
    const bool from0 = cb < p.C0;
    const T* sp = from0 ? reinterpret_cast<const T*>(p.src0) + cb : reinterpret_cast<const T*>(p.src1) + (cb - p.C0);
    const int Cs = from0 ? p.C0 : p.C1;
    sp += (size_t)n * p.HW * Cs;
#pragma unroll
    for (int i = 0; i < NL; ++i) keep[i] = *reinterpret_cast<const u32x4*>(sp + (size_t)(prow + i * ppi) * Cs);
#pragma unroll
    for (int i = 0; i + 3 < NL; i += 4) {   // the summation order of the general form
      float f0[V], f1[V], f2[V], f3[V];
      frag_to_float(keep[i], f0, T()); frag_to_float(keep[i + 1], f1, T()); frag_to_float(keep[i + 2], f2, T()); frag_to_float(keep[i + 3], f3, T());
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
        q[j] += (f0[j] * f0[j] + f1[j] * f1[j]) + (f2[j] * f2[j] + f3[j] * f3[j]);
      }
    }
#pragma unroll
    for (int i = NL / 4 * 4; i < NL; ++i) {
      float f0[V];
      frag_to_float(keep[i], f0, T());
#pragma unroll
      for (int j = 0; j < V; ++j) { s[j] += f0[j]; q[j] += f0[j] * f0[j]; }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red_s[prow * C + cb + j] = s[j]; red_q[prow * C + cb + j] = q[j]; }
  } else if (prow < ppi) {
    const bool from0 = cb < p.C0;
    const T* sp = from0 ? reinterpret_cast<const T*>(p.src0) + cb : reinterpret_cast<const T*>(p.src1) + (cb - p.C0);
    const int Cs = from0 ? p.C0 : p.C1;
    sp += (size_t)n * p.HW * Cs;
    int pix = prow;
    for (; pix + 3 * ppi < p.HW; pix += 4 * ppi) {  // 4 independent 16-B loads in flight
      u32x4 r0 = *reinterpret_cast<const u32x4*>(sp + (size_t)pix * Cs);
      u32x4 r1 = *reinterpret_cast<const u32x4*>(sp + (size_t)(pix + ppi) * Cs);
      u32x4 r2 = *reinterpret_cast<const u32x4*>(sp + (size_t)(pix + 2 * ppi) * Cs);
      u32x4 r3 = *reinterpret_cast<const u32x4*>(sp + (size_t)(pix + 3 * ppi) * Cs);
      float f0[V], f1[V], f2[V], f3[V];
      frag_to_float(r0, f0, T()); frag_to_float(r1, f1, T()); frag_to_float(r2, f2, T()); frag_to_float(r3, f3, T());
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
        q[j] += (f0[j] * f0[j] + f1[j] * f1[j]) + (f2[j] * f2[j] + f3[j] * f3[j]);
      }
    }
    for (; pix < p.HW; pix += ppi) {
      u32x4 r0 = *reinterpret_cast<const u32x4*>(sp + (size_t)pix * Cs);
      float f0[V];
      frag_to_float(r0, f0, T());
#pragma unroll
      for (int j = 0; j < V; ++j) { s[j] += f0[j]; q[j] += f0[j] * f0[j]; }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red_s[prow * C + cb + j] = s[j]; red_q[prow * C + cb + j] = q[j]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GN_THREADS) {
    float ts = 0.f, tq = 0.f;
    for (int r = 0; r < ppi; ++r) { ts += red_s[r * C + c]; tq += red_q[r * C + c]; }
    ch_s[c] = ts; ch_q[c] = tq;
  }
  __syncthreads();
  const int cpg = C / p.groups;
  if (tid < p.groups) {
    float ts = 0.f, tq = 0.f;
    for (int j = 0; j < cpg; ++j) { ts += ch_s[tid * cpg + j]; tq += ch_q[tid * cpg + j]; }
    const float inv = 1.0f / ((float)cpg * (float)p.HW);
    const float mean = ts * inv;
    const float var = fmaxf(tq * inv - mean * mean, 0.f);
    g_mean[tid] = mean;
    g_rstd[tid] = 1.0f / sqrtf(var + p.eps);
    if (p.mean) { p.mean[(size_t)n * p.groups + tid] = mean; p.rstd[(size_t)n * p.groups + tid] = g_rstd[tid]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += GN_THREADS) {
    const int g = c / cpg;
    const bool mine = c == tid;   // the first sweep uses the prefetched parameters
    float a = g_rstd[g] * (mine ? pg : p.gamma[c]);
    float b = (mine ? pbt : p.beta[c]) - g_mean[g] * a;
    if (p.film) {
      const float sc = 1.0f + (mine ? psc : p.film[(size_t)n * p.film_stride + c]);
      const float sh = mine ? psh : p.film[(size_t)n * p.film_stride + C + c];
      a *= sc;
      b = b * sc + sh;
    }
    p.a[(size_t)n * C + c] = a;
    p.b[(size_t)n * C + c] = b;
    ch_s[c] = a; ch_q[c] = b;
  }
  if (p.y == nullptr) return;
  // ---- apply pass (small images): the image was just read, so this second read comes from L2 ----
  __syncthreads();
  if constexpr (NL > 0) {
    constexpr bool FAST = Elem<T>::DTYPE == 1;
    T* yp = reinterpret_cast<T*>(p.y) + (size_t)n * p.HW * C + cb;
    float av[V], bv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { av[j] = ch_s[cb + j]; bv[j] = ch_q[cb + j]; }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float f[V];
      frag_to_float(keep[i], f, T());
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float v = av[j] * f[j] + bv[j];
        f[j] = p.y_silu ? (FAST ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)) : v / (1.0f + expf(-v))) : v;
      }
      *reinterpret_cast<u32x4*>(yp + (size_t)(prow + i * ppi) * C) = float_to_frag(f, T());
    }
  } else if (prow < ppi) {
    constexpr bool FAST = Elem<T>::DTYPE == 1;
    const bool from0 = cb < p.C0;
    const T* sp = from0 ? reinterpret_cast<const T*>(p.src0) + cb : reinterpret_cast<const T*>(p.src1) + (cb - p.C0);
    const int Cs = from0 ? p.C0 : p.C1;
    sp += (size_t)n * p.HW * Cs;
    T* yp = reinterpret_cast<T*>(p.y) + (size_t)n * p.HW * C + cb;
    float av[V], bv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { av[j] = ch_s[cb + j]; bv[j] = ch_q[cb + j]; }
    for (int pix = prow; pix < p.HW; pix += ppi) {
      const u32x4 r0 = *reinterpret_cast<const u32x4*>(sp + (size_t)pix * Cs);
      float f[V];
      frag_to_float(r0, f, T());
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float v = av[j] * f[j] + bv[j];
        f[j] = p.y_silu ? (FAST ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)) : v / (1.0f + expf(-v))) : v;
      }
      *reinterpret_cast<u32x4*>(yp + (size_t)pix * C) = float_to_frag(f, T());
    }
  }
}

// (a, b) of one GroupNorm site from the per-wave partial sums the producing convs left in their epilogues (common.h GnPartial:
// stats[N][slots][C/4][2] = sum, sum of squares per channel quad): no pass over the activation.  A workgroup per image; the
// partials are summed in a fixed order (bitwise reproducible).  The input may be the channel concat of two tensors, each with
// its own partial buffer; a group may straddle the two (e.g. 256 + 128 channels: groups of 12).
struct GnFinArgs {
  const float* st0; const float* st1; int slots0, slots1, C0, C1;
  int HW, groups; float eps;
  const float* gamma; const float* beta; const float* film; int film_stride;
  float* a; float* b;
  const void* warm; uint32_t warm_bytes;   // as in GnKArgs: one extra wave touches the consumer conv's weights
};
__global__ void __launch_bounds__(320) gn_finalize_kernel(GnFinArgs p) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int C = p.C0 + p.C1, Q = C >> 2, Q0 = p.C0 >> 2, Q1 = p.C1 >> 2;
  float* qs = fsm;            // [Q] sums
  float* qq = fsm + Q;        // [Q] sums of squares
  float* g_mean = qq + Q;     // [groups]
  float* g_rstd = g_mean + p.groups;
  const int tid = threadIdx.x, n = blockIdx.x;
  if (tid >= 256) { l2_warm_wave(p.warm, p.warm_bytes); return; }
  // the per-channel parameters of this thread's (up to two) channels are requested FIRST: they do not depend on the statistics, and
  // the kernel is nothing but dependent round trips (partials -> group statistics -> (a, b)), so they travel with the partials
  float pg[2], pbt[2], psc[2] = {0.f, 0.f}, psh[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = min(tid + 256 * k, C - 1);
    pg[k] = p.gamma[c]; pbt[k] = p.beta[c];
    if (p.film) { psc[k] = p.film[(size_t)n * p.film_stride + c]; psh[k] = p.film[(size_t)n * p.film_stride + C + c]; }
  }
  for (int q = tid; q < Q; q += 256) {
    const bool first = q < Q0;
    const float* base = first ? p.st0 + ((size_t)n * p.slots0 * Q0 + q) * 2 : p.st1 + ((size_t)n * p.slots1 * Q1 + (q - Q0)) * 2;
    const int slots = first ? p.slots0 : p.slots1, stride = (first ? Q0 : Q1) * 2;
    // slots summed in a fixed order, eight independent loads in flight (a dependent chain of L2 round trips is all this kernel is)
    float s = 0.f, sq = 0.f;
    int k = 0;
    for (; k + 8 <= slots; k += 8) {
      f32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x2*>(base + (size_t)(k + u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += v[u][0]; sq += v[u][1]; }
    }
    for (; k < slots; ++k) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(base + (size_t)k * stride);
      s += v[0]; sq += v[1];
    }
    qs[q] = s; qq[q] = sq;
  }
  __syncthreads();
  const int cpg = C / p.groups, qpg = cpg >> 2;
  if (tid < p.groups) {
    float ts = 0.f, tq = 0.f;
    for (int j = 0; j < qpg; ++j) { ts += qs[tid * qpg + j]; tq += qq[tid * qpg + j]; }
    const float inv = 1.0f / ((float)cpg * (float)p.HW);
    const float mean = ts * inv;
    const float var = fmaxf(tq * inv - mean * mean, 0.f);
    g_mean[tid] = mean;
    g_rstd[tid] = 1.0f / sqrtf(var + p.eps);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    if (c < C) {
      const int g = c / cpg;
      float a = g_rstd[g] * pg[k];
      float b = pbt[k] - g_mean[g] * a;
      if (p.film) {
        const float sc = 1.0f + psc[k];
        a *= sc;
        b = b * sc + psh[k];
      }
      p.a[(size_t)n * C + c] = a;
      p.b[(size_t)n * C + c] = b;
    }
  }
  for (int c = tid + 512; c < C; c += 256) {   // more than 512 channels: the rest the slow way
    const int g = c / cpg;
    float a = g_rstd[g] * p.gamma[c];
    float b = p.beta[c] - g_mean[g] * a;
    if (p.film) {
      const float sc = 1.0f + p.film[(size_t)n * p.film_stride + c];
      const float sh = p.film[(size_t)n * p.film_stride + C + c];
      a *= sc;
      b = b * sc + sh;
    }
    p.a[(size_t)n * C + c] = a;
    p.b[(size_t)n * C + c] = b;
  }
}

// Standalone GroupNorm(+SiLU) on NCHW fp32 (parity-test op; one workgroup per (n, group)).
__global__ void __launch_bounds__(256) groupnorm_nchw_kernel(const float* x, const float* gamma, const float* beta, float* y,
                                                           int C, int HW, int groups, float eps, int silu) {
  __shared__ float sh[16];
  const int n = blockIdx.x / groups, g = blockIdx.x % groups, cpg = C / groups;
  const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
  const int cnt = cpg * HW;
  float s = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) { float v = x[base + i]; s += v; q += v * v; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }  // wave64 shuffle reduction
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = q; }
  __syncthreads();
  s = sh[0] + sh[1] + sh[2] + sh[3];
  q = sh[4] + sh[5] + sh[6] + sh[7];
  const float mean = s / cnt, var = fmaxf(q / cnt - mean * mean, 0.f), rstd = 1.0f / sqrtf(var + eps);
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int c = g * cpg + i / HW;
    float v = (x[base + i] - mean) * rstd * gamma[c] + beta[c];
    y[base + i] = silu ? silu_f<false>(v) : v;
  }
}

// out[n, y, x, c] = mean over the 2x2 block of  silu?(a[n,c] * in + b[n,c])   (ResBlock(down=True): GN, SiLU, AvgPool2d(2),
// AD/image_diffusion/unet.py:332-337,236).  HBM-bound, 16-byte fragments; the pooled tensor feeds the conv unchanged.
template <typename T>
__global__ void __launch_bounds__(256) affine_pool_kernel(const T* in, const float* a, const float* b, int silu, T* out, int N, int Hs,
                                                        int Ws, int C) {
  constexpr int V = Elem<T>::VEC;
  constexpr bool FAST = Elem<T>::DTYPE == 1;
  const int Ho = Hs / 2, Wo = Ws / 2, CV = C / V;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)N * Ho * Wo * CV) return;
  const int f = idx % CV;
  size_t r = idx / CV;
  const int x = r % Wo; r /= Wo;
  const int y = r % Ho;
  const size_t n = r / Ho;
  float av[V], bv[V], acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { av[j] = a ? a[n * C + f * V + j] : 1.f; bv[j] = a ? b[n * C + f * V + j] : 0.f; acc[j] = 0.f; }
  const T* base = in + ((n * Hs + 2 * y) * Ws + 2 * x) * C + f * V;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const T* q = base + ((size_t)(k >> 1) * Ws + (k & 1)) * C;
    float v[V];
    frag_to_float(*reinterpret_cast<const u32x4*>(q), v, T());
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = av[j] * v[j] + bv[j];
      acc[j] += silu ? silu_f<FAST>(t) : t;
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] *= 0.25f;
  *reinterpret_cast<u32x4*>(out + (((n * Ho + y) * Wo + x) * C + f * V)) = float_to_frag(acc, T());
}

}  // namespace

int affine_pool_launch(int dtype, const void* in, const float* a, const float* b, int silu, void* out, int N, int Hs, int Ws, int C,
                       hipStream_t s) {
  const int V = dtype == 0 ? 4 : 8;
  MI355_REQUIRE(C % V == 0 && Hs % 2 == 0 && Ws % 2 == 0, -2, "affine_pool: needs even size and 16-byte channel fragments");
  const size_t total = (size_t)N * (Hs / 2) * (Ws / 2) * (C / V);
  dim3 grid((unsigned)((total + 255) / 256));
  dispatch_dtype(dtype, [&](auto t) { using T = decltype(t); hipLaunchKernelGGL(affine_pool_kernel<T>, grid, dim3(256), 0, s, (const T*)in, a, b, silu, (T*)out, N, Hs, Ws, C); return 0; });
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int gn_affine_launch(const GnDesc& d, hipStream_t stream) {
  const int C = d.C0 + d.C1;
  const int V = d.dtype == 0 ? 4 : 8;
  MI355_REQUIRE(C % d.groups == 0, -2, "groupnorm: channels not divisible by groups");
  MI355_REQUIRE(d.C0 % V == 0 && d.C1 % V == 0, -2, "groupnorm: channels must be a multiple of the 16-byte fragment");
  MI355_REQUIRE(C / V <= GN_THREADS, -4, "groupnorm: too many channels");
  MI355_REQUIRE(d.groups <= GN_THREADS, -4, "groupnorm: too many groups");
  GnKArgs a{d.src0, d.src1, d.C0, d.C1, d.N, d.HW, d.groups, d.eps, d.gamma, d.beta, d.film, d.film_stride, d.a, d.b, d.y, d.y_silu, d.mean, d.rstd, d.warm, d.warm_bytes};
  const int ppi = GN_THREADS / (C / V);
  const size_t lds = ((size_t)2 * ppi * C + 2 * C + 2 * d.groups) * sizeof(float);
  const int nthreads = GN_THREADS + (d.warm && d.warm_bytes ? 64 : 0);
  // small-image form: the image is a whole number NL of fragments per thread and the launch writes y
  int nl = 0;
  if (d.y && GN_THREADS % (C / V) == 0 && ((size_t)d.HW * (C / V)) % GN_THREADS == 0) {
    const size_t q = (size_t)d.HW * (C / V) / GN_THREADS;
    if (q == 1 || q == 2 || q == 4 || q == 8) nl = (int)q;
  }
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(d.N), dim3(nthreads), lds, stream, a); };
  dispatch_dtype(d.dtype, [&](auto t) {
    using T = decltype(t);
    switch (nl) { case 1: go(gn_affine_kernel<T, 1>); break; case 2: go(gn_affine_kernel<T, 2>); break; case 4: go(gn_affine_kernel<T, 4>); break;
                  case 8: go(gn_affine_kernel<T, 8>); break; default: go(gn_affine_kernel<T, 0>); }
    return 0;
  });
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int gn_finalize_launch(const GnFinDesc& d, hipStream_t stream) {
  const int C = d.C0 + d.C1;
  MI355_REQUIRE(d.stats0 && d.slots0 > 0 && (d.C1 == 0 || (d.stats1 && d.slots1 > 0)), -1, "gn_finalize: missing partial statistics");
  MI355_REQUIRE(C % d.groups == 0 && (C / d.groups) % 4 == 0 && d.C0 % 4 == 0 && d.C1 % 4 == 0, -2,
                "gn_finalize: groups must be whole channel quads");
  MI355_REQUIRE(d.groups <= 256, -4, "gn_finalize: too many groups");
  GnFinArgs a{d.stats0, d.stats1, d.slots0, d.slots1, d.C0, d.C1, d.HW, d.groups, d.eps, d.gamma, d.beta, d.film, d.film_stride, d.a, d.b, d.warm, d.warm_bytes};
  const size_t lds = ((size_t)2 * (C / 4) + 2 * d.groups) * sizeof(float);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(d.N), dim3(d.warm && d.warm_bytes ? 320 : 256), lds, stream, a);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}

int groupnorm_nchw_launch(const float* x, const float* gamma, const float* beta, float* y, int N, int C, int HW, int groups,
                          float eps, int silu, hipStream_t s) {
  MI355_REQUIRE(C % groups == 0, -2, "groupnorm: channels not divisible by groups");
  hipLaunchKernelGGL(groupnorm_nchw_kernel, dim3(N * groups), dim3(256), 0, s, x, gamma, beta, y, C, HW, groups, eps, silu);
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
