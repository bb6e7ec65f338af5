// Fused QK^T -> fp32 softmax -> PV (flash-style, online softmax) for the U-Net's AttentionBlock.
//
// Reference: QKVAttentionLegacy.forward (AD/image_diffusion/unet.py:433-448) and QKVAttention.forward
// (:464-483): w = softmax_s((q*s)^T (k*s)), s = ch^-1/4, softmax in fp32; a = w v.  The reference
// materialises the [B*heads, T, T] weight matrix in HBM; here it never leaves registers.
//
// Layout: qkv is NHWC [N][T][3C] (the qkv 1x1 conv's output), channel order as in the reference
// (legacy: per head [q|k|v]; new: [all q | all k | all v]).  out is NHWC [N][T][C].
// One workgroup = 64 query rows of one (image, head); 4 waves x 16 rows.  Per 64-key tile:
//   S^T = K Q^T   (keys on MFMA rows, queries on the lane -> column softmax is lane-local + 2 shuffles)
//   O^T += V^T P^T (P^T is already the B operand, straight from the S^T accumulators; V is transposed
//                   once while it is staged into LDS).
#include "ops.h"

namespace {

struct AttnKArgs {
  const void* qkv; void* out;
  int N, T, heads, C;
  int qoff_h, koff, voff;  // channel offsets: q of head h at h*qoff_h, k at koff + h*qoff_h, v at voff + h*qoff_h
  float scale2;
};

template <typename T, int CH>
__global__ void __launch_bounds__(256) attention_kernel(AttnKArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, SZ = sizeof(T);
  constexpr int KST = CH / CHUNK;           // k-steps over channels for S^T
  constexpr int CI = CH / 16;               // 16-channel row tiles of O^T
  constexpr int KROW = CH * SZ + 32;        // K tile row stride (bytes)
  constexpr int VROW = 64 * SZ + 16;        // V^T tile row stride (bytes), 64 keys per row
  __shared__ __attribute__((aligned(16))) char klds[64 * KROW];
  __shared__ __attribute__((aligned(16))) char vlds[CH * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const size_t rowstride = (size_t)3 * p.C;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)n * p.T * rowstride;
  const int qc = h * p.qoff_h, kc = p.koff + h * p.qoff_h, vc = p.voff + h * p.qoff_h;

  // Q^T fragments (B operand): lane holds Q[q0+lr][ks*CHUNK + lq*V .. +V)
  u32x4 qf[KST];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    qf[ks] = u32x4{0u, 0u, 0u, 0u};
    if (q0 + lr < p.T) qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)(q0 + lr) * rowstride + qc + ks * CHUNK + lq * V);
  }

  f32x4 o[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) o[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.T + 63) / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    // ---- stage K [64][CH] and V^T [CH][64] ----
    constexpr int FPR = CH / V;  // 16-B fragments per row
    for (int e = tid; e < 64 * FPR; e += 256) {
      const int s = e / FPR, f = e % FPR;
      const int key = kt * 64 + s;
      u32x4 kv = u32x4{0u, 0u, 0u, 0u}, vv = u32x4{0u, 0u, 0u, 0u};
      if (key < p.T) {
        const T* rp = base + (size_t)key * rowstride;
        kv = *reinterpret_cast<const u32x4*>(rp + kc + f * V);
        vv = *reinterpret_cast<const u32x4*>(rp + vc + f * V);
      }
      *reinterpret_cast<u32x4*>(klds + s * KROW + f * 16) = kv;
      T tmp[V];
      *reinterpret_cast<u32x4*>(tmp) = vv;
#pragma unroll
      for (int j = 0; j < V; ++j) *reinterpret_cast<T*>(vlds + (f * V + j) * VROW + s * SZ) = tmp[j];
    }
    __syncthreads();

    // ---- S^T = K Q^T : 4 row tiles of 16 keys ----
    f32x4 sacc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      sacc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        u32x4 kf = *reinterpret_cast<const u32x4*>(klds + (mi * 16 + lr) * KROW + (ks * 4 + lq) * 16);
        mma16(sacc[mi], kf, qf[ks], T());
      }
    }
    // ---- online softmax over keys (column = query = lane&15; rows spread over regs and lane>>4) ----
    float mx = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 64 + mi * 16 + lq * 4 + r;
        float v = sacc[mi][r] * p.scale2;
        v = key < p.T ? v : -INFINITY;
        sacc[mi][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);  // 0 on the first tile (m_run = -inf)
    float psum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(sacc[mi][r] - m_new);
        sacc[mi][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ci][r] *= alpha;

    // ---- O^T += V^T P^T ----
    if constexpr (E::DTYPE == 1) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {  // 32 keys per MFMA
        bf16x8 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pb[r] = (bf16)sacc[2 * s2][r]; pb[4 + r] = (bf16)sacc[2 * s2 + 1][r]; }
        const u32x4 pfrag = __builtin_bit_cast(u32x4, pb);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const char* vr = vlds + (ci * 16 + lr) * VROW + (32 * s2 + 4 * lq) * SZ;
          u32x2 lo = *reinterpret_cast<const u32x2*>(vr);
          u32x2 hi = *reinterpret_cast<const u32x2*>(vr + 16 * SZ);
          mma16(o[ci], u32x4{lo[0], lo[1], hi[0], hi[1]}, pfrag, T());
        }
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {  // 16 keys per fragment pair
        const u32x4 pfrag = __builtin_bit_cast(u32x4, sacc[mi]);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          u32x4 vf = *reinterpret_cast<const u32x4*>(vlds + (ci * 16 + lr) * VROW + (16 * mi + 4 * lq) * SZ);
          mma16(o[ci], vf, pfrag, T());
        }
      }
    }
  }
  // ---- normalise and store: lane holds channels ci*16 + 4*lq + r of query q0 + lr ----
  float l = l_run;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  if (q0 + lr < p.T) {
    T* op = reinterpret_cast<T*>(p.out) + ((size_t)n * p.T + q0 + lr) * p.C + h * CH;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      T* dst = op + ci * 16 + 4 * lq;
      if constexpr (E::DTYPE == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[ci][0] * inv, o[ci][1] * inv, o[ci][2] * inv, o[ci][3] * inv};
      } else {
        bf16x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = (bf16)(o[ci][r] * inv);
        *reinterpret_cast<bf16x4*>(dst) = t;
      }
    }
  }
}

template <typename T>
int launch_attn(const AttnKArgs& a, int ch, hipStream_t s) {
  dim3 grid((a.T + 63) / 64, a.N * a.heads);
  if (ch == 32) hipLaunchKernelGGL((attention_kernel<T, 32>), grid, dim3(256), 0, s, a);
  else if (ch == 64) hipLaunchKernelGGL((attention_kernel<T, 64>), grid, dim3(256), 0, s, a);
  else if (ch == 128) hipLaunchKernelGGL((attention_kernel<T, 128>), grid, dim3(256), 0, s, a);
  else { mi355_set_error("attention: head channels must be 32, 64 or 128"); return -4; }
  return 0;
}

}  // namespace

int attention_launch(const AttnDesc& d, hipStream_t stream) {
  AttnKArgs a;
  a.qkv = d.qkv; a.out = d.out; a.N = d.N; a.T = d.T; a.heads = d.heads; a.C = d.heads * d.ch;
  if (d.new_order) { a.qoff_h = d.ch; a.koff = a.C; a.voff = 2 * a.C; }
  else { a.qoff_h = 3 * d.ch; a.koff = d.ch; a.voff = 2 * d.ch; }
  a.scale2 = 1.0f / sqrtf((float)d.ch);
  int rc = d.dtype == 0 ? launch_attn<float>(a, d.ch, stream) : launch_attn<bf16>(a, d.ch, stream);
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
