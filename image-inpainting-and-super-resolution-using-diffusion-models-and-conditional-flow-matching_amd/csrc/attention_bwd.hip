// Backward of the attention core (QKVAttentionLegacy / QKVAttention, AD/image_diffusion/unet.py:433-448, 464-483) for the
// reconstruction-guidance data gradient:   S = scale2 Q K^T, P = softmax_rows(S), A = P V
//   dV = P^T dA,  dP = dA V^T,  dS = P o (dP - D),  D_q = dA_q . A_q,  dQ = scale2 dS K,  dK = scale2 dS^T Q
// Flash-style, nothing of size T x T is stored; two kernels that mirror the forward's structure (attention.hip):
//   attention_bwd_q   workgroup = 64 queries of one (image, head): pass 1 over the key tiles rebuilds the softmax statistics
//                     L = m + log2(l) (log2 domain), pass 2 forms dS^T = P^T o (V dA^T - D) and accumulates dQ^T += K^T dS^T;
//                     also writes L and D per query for the second kernel;
//   attention_bwd_kv  workgroup = 64 keys: streams (Q, dA) tiles, S = Q K^T with the keys on the lane, dV^T += dA^T P,
//                     dK^T += Q^T dS - the accumulator tiles are the next MFMA's B operand as they stand (their rows, the
//                     queries, are the summed index), Q^T / dA^T come through the transposed LDS read.
// qkv / dqkv: NHWC [N][T][3C] in the reference's channel order; a, da: [N][T][C]; L, D: fp32 [N * heads][T].
#include "ops.h"

namespace {

struct AttnBwdArgs {
  const void* qkv; const void* a; const void* da; void* dqkv;
  float* L; float* D;
  int N, T, heads, C;
  int qoff_h, koff, voff;
  float scale2;
};
typedef short s16x4 __attribute__((ext_vector_type(4)));

// A^T-operand fragment of a row-major LDS tile ([row][channel], row stride ROW bytes): channels 16 ci + (lane & 15) as MFMA rows,
// 8 (bf16) / 4 (fp32) tile rows as the k index, in the k order of the B fragments built from two accumulator tiles
// (bf16: rows r0 + 4 lq + {0..3} and r0 + 16 + 4 lq + {0..3}; fp32: rows r0 + 4 lq + {0..3})
template <typename T, int ROW>
__device__ __forceinline__ u32x4 tr_frag(const char* tile, int r0, int ci, int lr, int lq) {
  if constexpr (Elem<T>::DTYPE == 1) {
    const char* vrow = tile + (r0 + 4 * lq + (lr >> 2)) * ROW + 8 * (lr & 3) + ci * 32;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 16 * ROW));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2[0], l2[1], h2[0], h2[1]};
  } else {
    const char* vp = tile + (r0 + 4 * lq) * ROW + (16 * ci + lr) * 4;
    return u32x4{*reinterpret_cast<const uint32_t*>(vp), *reinterpret_cast<const uint32_t*>(vp + ROW),
                 *reinterpret_cast<const uint32_t*>(vp + 2 * ROW), *reinterpret_cast<const uint32_t*>(vp + 3 * ROW)};
  }
}
// B fragment (k = the accumulator tiles' row index) from one (fp32) or two (bf16) 16x16 accumulator tiles
template <typename T>
__device__ __forceinline__ u32x4 acc_frag(const f32x4& t0, const f32x4& t1) {
  if constexpr (Elem<T>::DTYPE == 1) {
    bf16x8 pb;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pb[r] = (bf16)t0[r]; pb[4 + r] = (bf16)t1[r]; }
    return __builtin_bit_cast(u32x4, pb);
  } else {
    return __builtin_bit_cast(u32x4, t0);
  }
}

// Register budget as in attention.hip (attn_min_blocks): two workgroups per CU = a 256-register budget = MFMAs in their VGPR form, no
// v_accvgpr copies around the softmax; the head sizes whose live set needs more keep the full budget.
template <typename T, int CH> constexpr int bwd_q_min_blocks() { return (sizeof(T) == 2 ? CH <= 192 : CH <= 96) ? 2 : 1; }
template <typename T, int CH> constexpr int bwd_kv_min_blocks() { return (sizeof(T) == 2 ? CH <= 128 : CH <= 64) ? 2 : 1; }

// ---- dQ (+ L, D) ----------------------------------------------------------------------------------------------------------------
template <typename T, int CH>
__global__ void __launch_bounds__(256, (bwd_q_min_blocks<T, CH>())) attention_bwd_q_kernel(AttnBwdArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, SZ = sizeof(T);
  constexpr bool BF = E::DTYPE == 1;
  constexpr int KST = CH / CHUNK, CI = CH / 16, KT = 64, MT = 4;
  constexpr int ROW = CH * SZ + 32, TILE = KT * ROW, FPR = CH / V, NF = (KT * FPR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K tile | V tile
  char* klds = smem; char* vlds = smem + TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
  const int q = blockIdx.x * 64 + wave * 16 + lr;
  const bool qok = q < p.T;
  const size_t rs = (size_t)3 * p.C;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)n * p.T * rs;
  const int qc = h * p.qoff_h, kc = p.koff + h * p.qoff_h, vc = p.voff + h * p.qoff_h;
  const T* arow = reinterpret_cast<const T*>(p.a) + ((size_t)n * p.T + (qok ? q : 0)) * p.C + h * CH;
  const T* drow = reinterpret_cast<const T*>(p.da) + ((size_t)n * p.T + (qok ? q : 0)) * p.C + h * CH;

  u32x4 qf[KST], df[KST];
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    qf[ks] = u32x4{0u, 0u, 0u, 0u}; df[ks] = u32x4{0u, 0u, 0u, 0u};
    if (qok) {
      qf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)q * rs + qc + ks * CHUNK + lq * V);
      df[ks] = *reinterpret_cast<const u32x4*>(drow + ks * CHUNK + lq * V);
      float fa[V], fd[V];
      frag_to_float(*reinterpret_cast<const u32x4*>(arow + ks * CHUNK + lq * V), fa, T());
      frag_to_float(df[ks], fd, T());
#pragma unroll
      for (int j = 0; j < V; ++j) dsum += fa[j] * fd[j];
    }
  }
  dsum += __shfl_xor(dsum, 16);
  dsum += __shfl_xor(dsum, 32);            // D_q = dA_q . A_q, identical in the four lanes of a query column

  auto stage = [&](int kt) {
    u32x4 kreg[NF], vreg[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR, key = kt * KT + s;
      kreg[u] = u32x4{0u, 0u, 0u, 0u}; vreg[u] = u32x4{0u, 0u, 0u, 0u};
      if (e < KT * FPR && key < p.T) {
        kreg[u] = *reinterpret_cast<const u32x4*>(base + (size_t)key * rs + kc + f * V);
        vreg[u] = *reinterpret_cast<const u32x4*>(base + (size_t)key * rs + vc + f * V);
      }
    }
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR;
      if (e < KT * FPR) { *reinterpret_cast<u32x4*>(klds + s * ROW + f * 16) = kreg[u]; *reinterpret_cast<u32x4*>(vlds + s * ROW + f * 16) = vreg[u]; }
    }
  };
  auto scores = [&](int kt, f32x4 (&sacc)[MT]) {   // S^T tile (raw q.k), -inf on keys past the end
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      sacc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks)
        mma16(sacc[mi], *reinterpret_cast<const u32x4*>(klds + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16), qf[ks], T());
#pragma unroll
      for (int r = 0; r < 4; ++r) if (kt * KT + mi * 16 + lq * 4 + r >= p.T) sacc[mi][r] = -INFINITY;
    }
  };
  const int ntiles = (p.T + KT - 1) / KT;
  const float c2 = p.scale2 * 1.4426950408889634f;
  // ---- pass 1: softmax statistics ----
  float m_run = -INFINITY, l_run = 0.f;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage(kt);
    __syncthreads();
    f32x4 sacc[MT];
    scores(kt, sacc);
    float mx = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[mi][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * c2);
    float psum = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) psum += __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[mi][r], c2, -m_new));
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + psum;
    m_run = m_new;
  }
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  const float Lq = m_run + log2f(l_run);
  if (qok && lq == 0) { p.L[(size_t)blockIdx.y * p.T + q] = Lq; p.D[(size_t)blockIdx.y * p.T + q] = dsum; }
  // ---- pass 2: dQ^T += K^T dS^T ----
  f32x4 dq[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) dq[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage(kt);
    __syncthreads();
    f32x4 sacc[MT], dp[MT];
    scores(kt, sacc);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      dp[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks)
        mma16(dp[mi], *reinterpret_cast<const u32x4*>(vlds + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16), df[ks], T());   // dP^T = V dA^T
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[mi][r], c2, -Lq));   // 0 on masked keys
        sacc[mi][r] = pr * (dp[mi][r] - dsum) * p.scale2;                                  // dS^T (the S scale folded in)
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < (BF ? MT / 2 : MT); ++s2) {
      const u32x4 sf = BF ? acc_frag<T>(sacc[2 * s2], sacc[2 * s2 + 1]) : acc_frag<T>(sacc[s2], sacc[s2]);
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) mma16(dq[ci], tr_frag<T, ROW>(klds, (BF ? 32 : 16) * s2, ci, lr, lq), sf, T());
    }
  }
  if (qok) {
    T* op = reinterpret_cast<T*>(p.dqkv) + ((size_t)n * p.T + q) * rs + qc;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      float o4[4] = {dq[ci][0], dq[ci][1], dq[ci][2], dq[ci][3]};
      if constexpr (!BF) *reinterpret_cast<f32x4*>(op + ci * 16 + 4 * lq) = f32x4{o4[0], o4[1], o4[2], o4[3]};
      else { bf16x4 t; for (int r = 0; r < 4; ++r) t[r] = (bf16)o4[r]; *reinterpret_cast<bf16x4*>(op + ci * 16 + 4 * lq) = t; }
    }
  }
}

// ---- dK, dV -------------------------------------------------------------------------------------------------------------------------
template <typename T, int CH>
__global__ void __launch_bounds__(256, (bwd_kv_min_blocks<T, CH>())) attention_bwd_kv_kernel(AttnBwdArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, SZ = sizeof(T);
  constexpr bool BF = E::DTYPE == 1;
  constexpr int KST = CH / CHUNK, CI = CH / 16, QT = 64, MT = 4;
  constexpr int ROW = CH * SZ + 32, TILE = QT * ROW, FPR = CH / V, NF = (QT * FPR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // Q tile | dA tile | L[64] | D[64]
  char* qlds = smem; char* dlds = smem + TILE;
  float* Ll = reinterpret_cast<float*>(smem + 2 * TILE); float* Dl = Ll + QT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
  const int key = blockIdx.x * 64 + wave * 16 + lr;
  const bool kok = key < p.T;
  const size_t rs = (size_t)3 * p.C;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)n * p.T * rs;
  const T* dab = reinterpret_cast<const T*>(p.da) + (size_t)n * p.T * p.C + h * CH;
  const int qc = h * p.qoff_h, kc = p.koff + h * p.qoff_h, vc = p.voff + h * p.qoff_h;
  u32x4 kf[KST], vf[KST];
#pragma unroll
  for (int ks = 0; ks < KST; ++ks) {
    kf[ks] = u32x4{0u, 0u, 0u, 0u}; vf[ks] = u32x4{0u, 0u, 0u, 0u};
    if (kok) {
      kf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)key * rs + kc + ks * CHUNK + lq * V);
      vf[ks] = *reinterpret_cast<const u32x4*>(base + (size_t)key * rs + vc + ks * CHUNK + lq * V);
    }
  }
  f32x4 dk[CI], dv[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) { dk[ci] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ci] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float c2 = p.scale2 * 1.4426950408889634f;
  const int ntiles = (p.T + QT - 1) / QT;
  for (int qt = 0; qt < ntiles; ++qt) {
    __syncthreads();
    {   // stage the query tile: Q and dA rows (zero past the end), L (+inf past the end: P = 0) and D
      u32x4 qreg[NF], dreg[NF];
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR, q = qt * QT + s;
        qreg[u] = u32x4{0u, 0u, 0u, 0u}; dreg[u] = u32x4{0u, 0u, 0u, 0u};
        if (e < QT * FPR && q < p.T) {
          qreg[u] = *reinterpret_cast<const u32x4*>(base + (size_t)q * rs + qc + f * V);
          dreg[u] = *reinterpret_cast<const u32x4*>(dab + (size_t)q * p.C + f * V);
        }
      }
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR;
        if (e < QT * FPR) { *reinterpret_cast<u32x4*>(qlds + s * ROW + f * 16) = qreg[u]; *reinterpret_cast<u32x4*>(dlds + s * ROW + f * 16) = dreg[u]; }
      }
      if (tid < QT) {
        const int q = qt * QT + tid;
        Ll[tid] = q < p.T ? p.L[(size_t)blockIdx.y * p.T + q] : INFINITY;
        Dl[tid] = q < p.T ? p.D[(size_t)blockIdx.y * p.T + q] : 0.f;
      }
    }
    __syncthreads();
    // S = Q K^T and dP = dA V^T: rows = queries (registers / lane >> 4), columns = this wave's keys (lane & 15)
    f32x4 pt[MT], ds[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        mma16(s, *reinterpret_cast<const u32x4*>(qlds + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16), kf[ks], T());
        mma16(dp, *reinterpret_cast<const u32x4*>(dlds + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16), vf[ks], T());
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mi * 16 + lq * 4 + r;
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -Ll[row]));
        pt[mi][r] = pr;
        ds[mi][r] = pr * (dp[r] - Dl[row]) * p.scale2;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < (BF ? MT / 2 : MT); ++s2) {
      const u32x4 pf = BF ? acc_frag<T>(pt[2 * s2], pt[2 * s2 + 1]) : acc_frag<T>(pt[s2], pt[s2]);
      const u32x4 sf = BF ? acc_frag<T>(ds[2 * s2], ds[2 * s2 + 1]) : acc_frag<T>(ds[s2], ds[s2]);
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        mma16(dv[ci], tr_frag<T, ROW>(dlds, (BF ? 32 : 16) * s2, ci, lr, lq), pf, T());   // dV^T += dA^T P
        mma16(dk[ci], tr_frag<T, ROW>(qlds, (BF ? 32 : 16) * s2, ci, lr, lq), sf, T());   // dK^T += Q^T dS
      }
    }
  }
  if (kok) {
    T* okp = reinterpret_cast<T*>(p.dqkv) + ((size_t)n * p.T + key) * rs + kc;
    T* ovp = reinterpret_cast<T*>(p.dqkv) + ((size_t)n * p.T + key) * rs + vc;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      if constexpr (!BF) {
        *reinterpret_cast<f32x4*>(okp + ci * 16 + 4 * lq) = dk[ci];
        *reinterpret_cast<f32x4*>(ovp + ci * 16 + 4 * lq) = dv[ci];
      } else {
        bf16x4 tk, tv;
        for (int r = 0; r < 4; ++r) { tk[r] = (bf16)dk[ci][r]; tv[r] = (bf16)dv[ci][r]; }
        *reinterpret_cast<bf16x4*>(okp + ci * 16 + 4 * lq) = tk;
        *reinterpret_cast<bf16x4*>(ovp + ci * 16 + 4 * lq) = tv;
      }
    }
  }
}

template <typename T, int CH>
int launch_bwd(const AttnBwdArgs& a, hipStream_t s) {
  constexpr size_t tile = (size_t)64 * (CH * sizeof(T) + 32);
  constexpr size_t lds_q = 2 * tile, lds_kv = 2 * tile + 2 * 64 * sizeof(float);
  static_assert(lds_kv <= 160 * 1024, "attention backward tile does not fit the LDS");
  auto kq = attention_bwd_q_kernel<T, CH>;
  auto kkv = attention_bwd_kv_kernel<T, CH>;
  if (lds_q > 64 * 1024) { if (int rc = mi355_allow_big_lds(kq, "attention backward")) return rc; }
  if (lds_kv > 64 * 1024) { if (int rc = mi355_allow_big_lds(kkv, "attention backward")) return rc; }
  dim3 grid((a.T + 63) / 64, a.N * a.heads);
  hipLaunchKernelGGL(kq, grid, dim3(256), lds_q, s, a);
  hipLaunchKernelGGL(kkv, grid, dim3(256), lds_kv, s, a);
  return 0;
}

template <typename T>
int launch_bwd_ch(const AttnBwdArgs& a, int ch, hipStream_t s) {
  switch (ch) {
    case 32: return launch_bwd<T, 32>(a, s);
    case 64: return launch_bwd<T, 64>(a, s);
    case 96: return launch_bwd<T, 96>(a, s);
    case 128: return launch_bwd<T, 128>(a, s);
    case 192: return launch_bwd<T, 192>(a, s);
    case 256: return launch_bwd<T, 256>(a, s);
    default: break;
  }
  mi355_set_error("attention backward: head channels must be one of 32, 64, 96, 128, 192, 256 (got " + std::to_string(ch) + ")");
  return -4;
}

}  // namespace

int attention_bwd_launch(const AttnBwdDesc& d, hipStream_t stream) {
  MI355_REQUIRE(d.qkv && d.a && d.da && d.dqkv && d.L && d.D, -1, "attention backward: null argument");
  AttnBwdArgs a;
  a.qkv = d.qkv; a.a = d.a; a.da = d.da; a.dqkv = d.dqkv; a.L = d.L; a.D = d.D;
  a.N = d.N; a.T = d.T; a.heads = d.heads; a.C = d.heads * d.ch;
  if (d.new_order) { a.qoff_h = d.ch; a.koff = a.C; a.voff = 2 * a.C; }
  else { a.qoff_h = 3 * d.ch; a.koff = d.ch; a.voff = 2 * d.ch; }
  a.scale2 = 1.0f / sqrtf((float)d.ch);
  int rc = d.dtype == 0 ? launch_bwd_ch<float>(a, d.ch, stream) : launch_bwd_ch<bf16>(a, d.ch, stream);
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
