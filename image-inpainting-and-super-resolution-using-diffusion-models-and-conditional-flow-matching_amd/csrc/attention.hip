// Fused QK^T -> fp32 softmax -> PV (flash-style, online softmax) for the U-Net's AttentionBlock.
//
// Reference: QKVAttentionLegacy.forward (AD/image_diffusion/unet.py:433-448) and QKVAttention.forward
// (:464-483): w = softmax_s((q*s)^T (k*s)), s = ch^-1/4, softmax in fp32; a = w v.  The reference
// materialises the [B*heads, T, T] weight matrix in HBM; here it never leaves registers.
//
// Layout: qkv is NHWC [N][T][3C] (the qkv 1x1 conv's output), channel order as in the reference
// (legacy: per head [q|k|v]; new: [all q | all k | all v]).  out is NHWC [N][T][C].
// One workgroup = 4 waves x QB blocks of 16 query rows of one (image, head).  Per KT-key tile:
//   S^T = K Q^T    (keys on MFMA rows, queries on the lane -> the softmax column is lane-local + 2 shuffles)
//   O^T += V^T P^T (P^T is already the B operand, straight from the S^T accumulators)
// K and V tiles are staged ROW-MAJOR ([key][channel], 16-byte copies, rows padded by 32 B): K is read by rows
// (ds_read_b128), V^T comes out of the same kind of image through the hardware transpose read ds_read_b64_tr_b16
// (bf16; 4 keys x 16 channels per 16-lane group, conflict-free with the 32-byte row pad) - no transposing stores.
// Tiles are double-buffered (NBUF = 2): the next tile's global loads are issued right after the barrier that publishes
// the current one and written to the other buffer after the tile's MFMAs, so there is ONE barrier per key tile and the
// load latency hides behind the compute.  Head sizes up to 512 channels (torchcfm's default single head at 384 / 512
// channels, mnist/train_mnist_hy.py:312-318) use 32-key tiles (and, in fp32, a single buffer) to stay inside 160 KB.
#include "ops.h"

namespace {

struct AttnKArgs {
  const void* qkv; void* out;
  int N, T, heads, C;
  int qoff_h, koff, voff;  // channel offsets: q of head h at h*qoff_h, k at koff + h*qoff_h, v at voff + h*qoff_h
  float scale2;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// Register budget: with a 512-register budget (one workgroup per CU) the compiler puts the MFMA accumulators into AccVGPRs, and the
// softmax / rescale between the two GEMMs then costs a v_accvgpr_read or _write per accumulator register and key tile (192 of them
// beside 32 MFMAs at 64 channels).  Promising two workgroups per CU caps the budget at 256 = the VGPR form of the MFMAs, no copies;
// the head sizes whose accumulators need more than that keep the full budget.
template <typename T, int CH, int QB>
constexpr int attn_min_blocks() { return (sizeof(T) == 2 ? CH * QB <= 192 : CH <= 96) ? 2 : 1; }

template <typename T, int CH, int QB, int KT, int NBUF>
__global__ void __launch_bounds__(256, (attn_min_blocks<T, CH, QB>())) attention_kernel(AttnKArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, SZ = sizeof(T);
  constexpr bool BF = E::DTYPE == 1;
  constexpr int KST = CH / CHUNK;           // k-steps over channels for S^T
  constexpr int CI = CH / 16;               // 16-channel row tiles of O^T
  constexpr int MT = KT / 16;               // 16-key row tiles of S^T per key tile
  constexpr int ROW = CH * SZ + 32;         // K / V tile row stride (bytes)
  constexpr int TILE = KT * ROW;            // bytes of one K (or V) tile
  constexpr int FPR = CH / V;               // 16-B fragments per row
  constexpr int NF = (KT * FPR + 255) / 256;   // fragments per thread per tile (K and V each)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][K tile | V tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
  const int q0 = blockIdx.x * (64 * QB) + wave * (16 * QB);
  const size_t rowstride = (size_t)3 * p.C;
  const T* base = reinterpret_cast<const T*>(p.qkv) + (size_t)n * p.T * rowstride;
  const int qc = h * p.qoff_h, kc = p.koff + h * p.qoff_h, vc = p.voff + h * p.qoff_h;
  const float c2 = p.scale2 * 1.4426950408889634f;

  // Q^T fragments (B operand): lane holds Q[q0 + 16 qb + lr][ks*CHUNK + lq*V .. +V)
  u32x4 qf[QB][KST];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const int q = q0 + 16 * qb + lr;
      qf[qb][ks] = u32x4{0u, 0u, 0u, 0u};
      if (q < p.T) qf[qb][ks] = *reinterpret_cast<const u32x4*>(base + (size_t)q * rowstride + qc + ks * CHUNK + lq * V);
      // q carries ch^-1/2 * log2(e) from here on: S^T comes out of the MFMAs in the log2 domain and the softmax needs no multiply
      // per element (fp32: exact to rounding; bf16: one more rounding of q, the size of the one the qkv conv's store already made)
      float f[V];
      frag_to_float(qf[qb][ks], f, T());
#pragma unroll
      for (int j = 0; j < V; ++j) f[j] *= c2;
      qf[qb][ks] = float_to_frag(f, T());
    }

  f32x4 o[QB][CI];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) o[qb][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: thread owns fragments e = tid + 256 u of a tile (row e / FPR, 16-B slot e % FPR), K and V alike ----
  u32x4 kreg[NF], vreg[NF];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR;
      const int key = kt * KT + s;
      kreg[u] = u32x4{0u, 0u, 0u, 0u}; vreg[u] = u32x4{0u, 0u, 0u, 0u};
      if (e < KT * FPR && key < p.T) {
        const T* rp = base + (size_t)key * rowstride;
        kreg[u] = *reinterpret_cast<const u32x4*>(rp + kc + f * V);
        vreg[u] = *reinterpret_cast<const u32x4*>(rp + vc + f * V);
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* kb = smem + buf * (2 * TILE);
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int e = tid + 256 * u, s = e / FPR, f = e - s * FPR;
      if (e < KT * FPR) {
        *reinterpret_cast<u32x4*>(kb + s * ROW + f * 16) = kreg[u];
        *reinterpret_cast<u32x4*>(kb + TILE + s * ROW + f * 16) = vreg[u];
      }
    }
  };

  const int ntiles = (p.T + KT - 1) / KT;
  // the column maximum over the four lanes that share a query: two VALU row swaps instead of two LDS round trips
  auto colmax = [&](float v) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, v), false, false);
    v = fmaxf(__builtin_bit_cast(float, (uint32_t)s16[0]), __builtin_bit_cast(float, (uint32_t)s16[1]));
    const auto s32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)s32[0]), __builtin_bit_cast(float, (uint32_t)s32[1]));
  };
  load_tile(0);
  store_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = NBUF == 2 ? (kt & 1) : 0;
    __syncthreads();                        // tile kt is staged (and, NBUF == 2, every wave is done with tile kt - 1's buffer)
    if (NBUF == 2 && kt + 1 < ntiles) load_tile(kt + 1);   // in flight during this tile's MFMAs
    const char* klds = smem + buf * (2 * TILE);
    const char* vlds = klds + TILE;

    // ---- S^T = K Q^T : MT row tiles of 16 keys; a K fragment is read once and used by all QB query blocks ----
    f32x4 sacc[QB][MT];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) sacc[qb][mi] = f32x4{-m_run[qb], -m_run[qb], -m_run[qb], -m_run[qb]};   // the column's reference offset
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(klds + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) mma16(sacc[qb][mi], kf, qf[qb][ks], T());
      }
    // ---- online softmax over keys (column = query = lane&15; rows spread over regs and lane>>4), in the log2 domain: the accumulators
    //      hold s - m_ref, so p = exp2(sacc) (max, exp2, add per element).  The reference offset of a column moves only when a tile's
    //      maximum exceeds it by more than 8 (p <= 2^8): then the tile is shifted and O / l rescaled on a wave-uniform slow path; the
    //      first tile always takes it (its offset becomes the tile maximum, whatever its sign) ----
    if ((kt + 1) * KT > p.T) {              // only a ragged last tile has keys to mask (wave-uniform)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * KT + mi * 16 + lq * 4 + r >= p.T) sacc[qb][mi][r] = -INFINITY;
    }
    const bool first = kt == 0;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qb][mi][r]);
      mx = colmax(mx);
      const float delta = first ? mx : (mx > 8.0f ? mx : 0.0f);
      if (first || __builtin_amdgcn_ballot_w64(delta != 0.0f) != 0) {
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        m_run[qb] += delta;
        l_run[qb] *= alpha;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[qb][mi][r] -= delta;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[qb][ci][r] *= alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sacc[qb][mi][r]);
          sacc[qb][mi][r] = pv;
          psum += pv;
        }
      l_run[qb] += psum;
    }

    // ---- O^T += V^T P^T ----
    if constexpr (BF) {
#pragma unroll
      for (int s2 = 0; s2 < MT / 2; ++s2) {  // 32 keys per MFMA: keys 32 s2 + 4 lq + {0..3} and + 16
        u32x4 pfrag[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const u32x2 p0 = pack4(sacc[qb][2 * s2], T()), p1 = pack4(sacc[qb][2 * s2 + 1], T());
          pfrag[qb] = u32x4{p0[0], p0[1], p1[0], p1[1]};
        }
        // transposed read: lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of a 4-key x 16-channel block and
        // receives column (lane & 15) of its 4 rows = V^T[channel 16 ci + lr][4 consecutive keys]  (EXEC is all ones here)
        const char* vrow = vlds + (32 * s2 + 4 * lq + (lr >> 2)) * ROW + 8 * (lr & 3);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + ci * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 16 * ROW + ci * 32));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          const u32x4 vf = u32x4{l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) mma16(o[qb][ci], vf, pfrag[qb], T());
        }
      }
    } else {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {  // 16 keys per fragment pair: MFMA r of mma16 multiplies key 16 mi + 4 lq + r
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const char* vp = vlds + (16 * mi + 4 * lq) * ROW + (16 * ci + lr) * 4;
          const u32x4 vf = u32x4{*reinterpret_cast<const uint32_t*>(vp), *reinterpret_cast<const uint32_t*>(vp + ROW),
                                 *reinterpret_cast<const uint32_t*>(vp + 2 * ROW), *reinterpret_cast<const uint32_t*>(vp + 3 * ROW)};
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) mma16(o[qb][ci], vf, __builtin_bit_cast(u32x4, sacc[qb][mi]), T());
        }
      }
    }
    if (kt + 1 < ntiles) {
      if (NBUF == 2) store_tile((kt + 1) & 1);   // the other buffer: last read during tile kt - 1, before this tile's barrier
      else { __syncthreads(); load_tile(kt + 1); store_tile(0); }
    }
  }
  // ---- normalise and store: lane holds channels ci*16 + 4*lq + r of query q0 + 16 qb + lr ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int q = q0 + 16 * qb + lr;
    if (q < p.T) {
      T* op = reinterpret_cast<T*>(p.out) + ((size_t)n * p.T + q) * p.C + h * CH;
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        T* dst = op + ci * 16 + 4 * lq;
        if constexpr (!BF) {
          *reinterpret_cast<f32x4*>(dst) = f32x4{o[qb][ci][0] * inv, o[qb][ci][1] * inv, o[qb][ci][2] * inv, o[qb][ci][3] * inv};
        } else {
          *reinterpret_cast<u32x2*>(dst) = pack4(f32x4{o[qb][ci][0] * inv, o[qb][ci][1] * inv, o[qb][ci][2] * inv, o[qb][ci][3] * inv}, T());
        }
      }
    }
  }
}

template <typename T, int CH, int QB, int KT, int NBUF>
int launch_one(const AttnKArgs& a, hipStream_t s) {
  auto kern = attention_kernel<T, CH, QB, KT, NBUF>;
  constexpr size_t lds = (size_t)NBUF * 2 * KT * (CH * sizeof(T) + 32);
  static_assert(lds <= 160 * 1024, "attention tile does not fit the LDS");
  if (lds > 64 * 1024) { if (int rc = mi355_allow_big_lds(kern, "attention")) return rc; }
  dim3 grid((a.T + 64 * QB - 1) / (64 * QB), a.N * a.heads);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  return 0;
}

template <typename T>
int launch_attn(const AttnKArgs& a, int ch, hipStream_t s) {
  constexpr bool F32 = sizeof(T) == 4;
  // two query blocks per wave (128 queries per workgroup: half the K/V staging per query) once the sequence is long enough
  const bool wide = a.T >= 256;
  switch (ch) {
    case 32: return wide ? launch_one<T, 32, 2, 64, 2>(a, s) : launch_one<T, 32, 1, 64, 2>(a, s);
    case 64: return wide ? launch_one<T, 64, 2, 64, 2>(a, s) : launch_one<T, 64, 1, 64, 2>(a, s);
    case 96: return launch_one<T, 96, 1, 64, 2>(a, s);
    case 128: return launch_one<T, 128, 1, 64, 2>(a, s);
    case 192: return launch_one<T, 192, 1, 32, 2>(a, s);
    case 256: return launch_one<T, 256, 1, 32, 2>(a, s);
    case 384: return launch_one<T, 384, 1, 32, F32 ? 1 : 2>(a, s);
    case 512: return launch_one<T, 512, 1, 32, F32 ? 1 : 2>(a, s);
    default: break;
  }
  mi355_set_error("attention: head channels must be one of 32, 64, 96, 128, 192, 256, 384, 512 (got " + std::to_string(ch) + ")");
  return -4;
}

}  // namespace

int attention_launch(const AttnDesc& d, hipStream_t stream) {
  AttnKArgs a;
  a.qkv = d.qkv; a.out = d.out; a.N = d.N; a.T = d.T; a.heads = d.heads; a.C = d.heads * d.ch;
  if (d.new_order) { a.qoff_h = d.ch; a.koff = a.C; a.voff = 2 * a.C; }
  else { a.qoff_h = 3 * d.ch; a.koff = d.ch; a.voff = 2 * d.ch; }
  a.scale2 = 1.0f / sqrtf((float)d.ch);
  int rc = dispatch_dtype(d.dtype, [&](auto t) { return launch_attn<decltype(t)>(a, d.ch, stream); });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
