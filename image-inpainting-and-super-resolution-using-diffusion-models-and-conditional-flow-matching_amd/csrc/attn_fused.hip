// AttentionBlock front half in ONE kernel: GroupNorm-apply -> qkv 1x1 conv -> QK^T / fp32 softmax / PV.
//
// Reference: AttentionBlock._forward (AD/image_diffusion/unet.py:395-401): qkv = self.qkv(self.norm(x)); h = self.attention(qkv)
// with QKVAttentionLegacy / QKVAttention (:433-448, :464-483).  Unfused, the [B, T, 3C] qkv tensor makes a round trip through
// memory between a 1x1-conv launch and the attention launch (100 MB at B = 256, T = 256, C = 256); here it never exists:
// one workgroup = one (image, head), 8 waves, each wave owns T/8 tokens:
//   phase 1  D[out channel][token] = Wqkv_h (192 x C) . GN(x)^T: ALL of the head's q, k, v weight rows (192 x C: 96 KB at C = 256) are
//            staged once, up front, so the K loop over 64-byte channel chunks runs without a barrier or a weight wait (a
//            chunk is only 24 MFMAs per wave: a one-chunk-ahead prefetch cannot cover an L2 round trip); the x operand goes
//            straight from global memory to registers (a wave's tokens are its own), two chunks ahead, with the GroupNorm
//            affine applied in flight;
//   phase 2  q stays in registers - an accumulator tile is the next MFMA's B operand as it stands (rows = channels = the summed
//            index); k and v are written ROW-MAJOR into LDS (they alias the dead weight buffers), k with the channel order the
//            q fragments have, so a K fragment is one ds_read_b128;
//   phase 3  S^T = K Q^T, online softmax, O^T += V^T P^T over the resident K / V with no barrier at all (V^T through the
//            hardware transpose read ds_read_b64_tr_b16, as in attention.hip).
// Shapes: head size 64, T = 128 or 256 tokens, C <= 512 (the 16x16 / 8x16 attention levels of every mc = 128 configuration);
// anything else runs the unfused ops.  The four heads of an image are dispatched 8 workgroups apart (same XCD under the
// round-robin placement, speed only), so x is fetched into one L2 once.
#include "ops.h"

namespace {

struct AttnFuseArgs {
  const void* x; const float* ga; const float* gb;
  const void* w; const float* bias;
  void* out;
  int N, T, C, heads, nchunks, new_order;
  int stage_chunks;   // weight chunks resident per stage (all of them when the head's rows fit the LDS: one stage)
  float scale2;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int I> struct IC3 { static constexpr int value = I; };

template <typename T, int QB>
__global__ void __launch_bounds__(512) attn_fused_kernel(AttnFuseArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, SZ = sizeof(T), CH = 64;
  constexpr bool BF = E::DTYPE == 1;
  constexpr int KST = CH / CHUNK, CI = CH / 16, NCT = 12;   // k-steps of S^T, channel tiles of O^T, 16-row tiles of the head's q|k|v rows
  constexpr int ROW = CH * SZ + 32;                          // K / V row stride in LDS (bytes)
  constexpr int TT = 128 * QB;                               // tokens
  constexpr int WBUF = 192 * 64;                             // one chunk of the head's weight rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* klds = smem;                       // [TT][ROW]                       (phase 2 on)
  char* vlds = smem + TT * ROW;            // [TT][ROW]
  char* wlds = smem;                       // [nchunks][192 rows][64 B], aliases K / V    (phase 1)
  float* ablds = reinterpret_cast<float*>(smem + (size_t)p.stage_chunks * WBUF);   // a[C] | b[C]  (phase 1)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  // workgroup -> (image, head): heads of one image are 8 workgroups apart
  const int w = blockIdx.x, grp = w >> 3;
  const int n = (w & 7) + 8 * (grp / p.heads), h = grp % p.heads;
  if (n >= p.N) return;                    // whole workgroup (no barrier was reached)
  const int C = p.C;
  const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)n * TT * C;
  const int tok0 = wave * (16 * QB);

  // rows of this head in the [3C] output channels of the qkv conv: legacy = [h*192, h*192 + 192), new = q | k | v blocks of C
  auto grow = [&](int j) { return p.new_order ? (j >> 6) * C + h * CH + (j & 63) : h * (3 * CH) + j; };

  // x fragments (B operand): lane holds x[tok0 + 16 qb + lr][c*CHUNK + lq*V .. +V); a ring two chunks deep
  u32x4 xr[3][QB];
  auto load_x = [&](int c, u32x4 (&dst)[QB]) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) dst[qb] = *reinterpret_cast<const u32x4*>(xb + (size_t)(tok0 + 16 * qb + lr) * C + c * CHUNK + lq * V);
  };
  load_x(0, xr[0]);
  if (p.nchunks > 1) load_x(1, xr[1]);

  // ---- the head's weights are staged stage_chunks chunks at a time: per chunk, fragment e = tid + 512 u < 768 is (row e >> 2, 16-B
  //      slot e & 3); 64-byte rows are copied verbatim (the packed image's XOR swizzle depends on (row >> 1) & 3 only, and every
  //      head's row base is a multiple of 8); the per-thread source / destination offsets are chunk-invariant ----
  uint32_t wso[2]; int wdo[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 512 * u, j = min(e >> 2, 191), r = grow(j);
    wso[u] = ((uint32_t)(r >> 7) * p.nchunks) * 8192u + (uint32_t)(r & 127) * 64u + (e & 3) * 16;
    wdo[u] = j * 64 + (e & 3) * 16;
  }
  const bool second = tid + 512 < 768;
  auto stage_weights = [&](int c0, int nc) {
    const char* wsrc = reinterpret_cast<const char*>(p.w) + (size_t)c0 * 8192;
    for (int cl = 0; cl < nc; cl += 4) {   // four chunks in flight per thread (8 x 16 B)
      u32x4 t[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = min(cl + k, nc - 1);
        t[k][0] = *reinterpret_cast<const u32x4*>(wsrc + wso[0] + (size_t)cc * 8192);
        t[k][1] = *reinterpret_cast<const u32x4*>(wsrc + wso[1] + (size_t)cc * 8192);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (cl + k < nc) {
          *reinterpret_cast<u32x4*>(wlds + (cl + k) * WBUF + wdo[0]) = t[k][0];
          if (second) *reinterpret_cast<u32x4*>(wlds + (cl + k) * WBUF + wdo[1]) = t[k][1];
        }
      }
    }
  };
  for (int c = tid; c < C; c += 512) { ablds[c] = p.ga[(size_t)n * C + c]; ablds[C + c] = p.gb[(size_t)n * C + c]; }

  f32x4 acc[NCT][QB];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) acc[ct][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---------------- phase 1: q | k | v rows of this head for the wave's tokens (no barrier inside a stage) ----------------
  int c0 = 0;                              // first chunk of the resident stage
  auto chunk = [&](int c, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
    if (c + 2 < p.nchunks) load_x(c + 2, xr[(slot + 2) % 3]);
    // GroupNorm affine on the x fragments (the attention norm has no SiLU, unet.py:379,397)
    u32x4 xf[QB];
    {
      float av[V], bv[V];
#pragma unroll
      for (int j = 0; j < V; j += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ablds + c * CHUNK + lq * V + j);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ablds + C + c * CHUNK + lq * V + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) { av[j + k] = a4[k]; bv[j + k] = b4[k]; }
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float f[V];
        frag_to_float(xr[slot][qb], f, T());
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = av[j] * f[j] + bv[j];
        xf[qb] = float_to_frag(f, T());
      }
    }
    const char* wb = wlds + (c - c0) * WBUF + lr * 64 + 16 * (lq ^ ((lr >> 1) & 3));
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const u32x4 wf = *reinterpret_cast<const u32x4*>(wb + ct * 1024);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) mma16(acc[ct][qb], wf, xf[qb], T());   // D rows = output channels, cols = tokens
    }
  };
  // stage_chunks is a multiple of 3 (or all chunks): three chunks per trip keep the ring slot a compile-time index
  for (c0 = 0; c0 < p.nchunks; c0 += p.stage_chunks) {
    const int nc = min(p.stage_chunks, p.nchunks - c0);
    if (c0 > 0) __syncthreads();           // every wave is done with the previous stage's weights
    stage_weights(c0, nc);
    __syncthreads();                       // this stage's weights (and, first time, the (a, b) table) are staged
    for (int c = c0; c < c0 + nc; c += 3) {
      chunk(c, IC3<0>());
      if (c + 1 < c0 + nc) chunk(c + 1, IC3<1>());
      if (c + 2 < c0 + nc) chunk(c + 2, IC3<2>());
    }
  }
  // bias of the qkv conv: lane's rows of tile ct are channels ct*16 + 4 lq .. + 3
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + grow(ct * 16 + 4 * lq));
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ct][qb][r] += b4[r];
  }

  // ---------------- phase 2: q -> B-operand fragments; k, v -> LDS ----------------
  u32x4 qf[QB][KST];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      if constexpr (BF) {   // 32-channel k-step = tiles 2 ks and 2 ks + 1: elements 0-3 = channels 32 ks + 4 lq + r, 4-7 = + 16
        const u32x2 t0 = pack4(acc[2 * ks][qb], T()), t1 = pack4(acc[2 * ks + 1][qb], T());
        qf[qb][ks] = u32x4{t0[0], t0[1], t1[0], t1[1]};
      } else {
        qf[qb][ks] = __builtin_bit_cast(u32x4, acc[ks][qb]);
      }
    }
  __syncthreads();                         // every wave has finished reading the weight buffers K / V are about to overwrite
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int tok = tok0 + 16 * qb + lr;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const f32x4 kv = acc[4 + ct][qb], vv = acc[8 + ct][qb];
      if constexpr (BF) {
        // K: the 16-byte slot (ks, lq) holds tile 2 ks (first 8 bytes) and tile 2 ks + 1 (last 8) = the q fragments' channel order
        *reinterpret_cast<u32x2*>(klds + tok * ROW + (ct >> 1) * 64 + lq * 16 + (ct & 1) * 8) = pack4(kv, T());
        *reinterpret_cast<u32x2*>(vlds + tok * ROW + (16 * ct + 4 * lq) * 2) = pack4(vv, T());   // V: natural [key][channel]
      } else {
        *reinterpret_cast<f32x4*>(klds + tok * ROW + ct * 64 + lq * 16) = kv;
        *reinterpret_cast<f32x4*>(vlds + tok * ROW + (16 * ct + 4 * lq) * 4) = vv;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 3: attention over the resident K / V (no barrier) ----------------
  f32x4 o[QB][CI];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) o[qb][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float c2 = p.scale2 * 1.4426950408889634f;
  for (int kt = 0; kt < TT / 64; ++kt) {
    const char* kb = klds + kt * 64 * ROW;
    const char* vb = vlds + kt * 64 * ROW;
    f32x4 sacc[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) sacc[qb][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(kb + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) mma16(sacc[qb][mi], kf, qf[qb][ks], T());
      }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {   // log2-domain online softmax: p = exp2(s * c2 - m2), c2 = ch^-1/2 * log2(e)
      float mx = -INFINITY;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qb][mi][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // deferred rescale: the reference maximum only moves when the tile's maximum exceeds it by more than 8 (log2 domain), so
      // p <= 2^8 and most tiles skip the O / l rescale (alpha == 1 exactly); the decision is per query column and identical in
      // the four lanes that share it (mx is already reduced over them)
      const float mt = mx * c2;
      const float m_new = mt > m_run[qb] + 8.0f ? mt : m_run[qb];
      const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qb][mi][r], c2, -m_new)); sacc[qb][mi][r] = pv; psum += pv; }
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // wave-uniform: some column's maximum moved
        l_run[qb] *= alpha;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[qb][ci][r] *= alpha;
      }
      l_run[qb] += psum;
      m_run[qb] = m_new;
    }
    if constexpr (BF) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pfrag[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const u32x2 p0 = pack4(sacc[qb][2 * s2], T()), p1 = pack4(sacc[qb][2 * s2 + 1], T());
          pfrag[qb] = u32x4{p0[0], p0[1], p1[0], p1[1]};
        }
        const char* vrow = vb + (32 * s2 + 4 * lq + (lr >> 2)) * ROW + 8 * (lr & 3);
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
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const char* vp = vb + (16 * mi + 4 * lq) * ROW + (16 * ci + lr) * 4;
          const u32x4 vf = u32x4{*reinterpret_cast<const uint32_t*>(vp), *reinterpret_cast<const uint32_t*>(vp + ROW),
                                 *reinterpret_cast<const uint32_t*>(vp + 2 * ROW), *reinterpret_cast<const uint32_t*>(vp + 3 * ROW)};
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) mma16(o[qb][ci], vf, __builtin_bit_cast(u32x4, sacc[qb][mi]), T());
        }
    }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    T* op = reinterpret_cast<T*>(p.out) + ((size_t)n * TT + tok0 + 16 * qb + lr) * C + h * CH;
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

// ---- persistent form (bf16, T = 256, C = 32 NCH <= 256): one workgroup per CU keeps ITS head's q|k|v weight rows in LDS for every image
// it visits, so the 96-KB staging, its barrier and its L2 round trips happen once per workgroup instead of once per (image, head), and
// the next image's x fragments and (a, b) table are in flight while this image's attention runs.  The weights now stay resident, so K / V
// cannot alias them: the keys go through a 128-key (40-KB) buffer in two rounds - waves 0-3 own keys 0-127 and write them first, every
// wave attends to them, then waves 4-7 write keys 128-255 (their k / v rows wait in accumulators meanwhile); the online softmax does not
// care about the key order.  Four barriers per image, none inside a phase.
// Workgroup b -> (xcd = b & 7, head = (b >> 3) % heads, j = (b >> 3) / heads): image lane = xcd + 8 j, images lane, lane + lanes, ...;
// the heads of one image run at the same time on the same XCD (x is fetched into that L2 once).
#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0   // experiments only (make variant_src): compile-time ablation mask of the persistent kernel
#endif
template <int NCH, typename T>   // T: bf16 or f16 (two-byte elements)
__global__ void __launch_bounds__(512) attn_fused_pers_kernel(AttnFuseArgs p, int lanes) {
  constexpr int V = 8, CHUNK = 32, CH = 64, QB = 2, TT = 256;
  constexpr int KST = 2, CI = 4;
  constexpr int ROW = CH * 2 + 32;
  constexpr int WBUF = 192 * 64;
  constexpr int C = NCH * CHUNK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wlds = smem;                                   // [NCH][192 rows][64 B]: resident for the whole kernel
  char* klds = smem + NCH * WBUF;                      // [128 keys][ROW]
  char* vlds = klds + 128 * ROW;                       // [128 keys][ROW]
  float* ablds = reinterpret_cast<float*>(vlds + 128 * ROW);   // [2][a[C] | b[C]]: this image's and the next one's

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr0 = lane & 15, lq0 = lane >> 4;
  const int b = blockIdx.x, kq = b >> 3;
  const int h = kq % p.heads, ilane = (b & 7) + 8 * (kq / p.heads);
  if (ilane >= lanes || ilane >= p.N) return;          // whole workgroup, before any barrier
  const int tok0 = wave * (16 * QB);
  if constexpr ((ATTN_ABLATE & 512) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }   // experiment: the younger wave of each SIMD first
  auto grow = [&](int j) { return p.new_order ? (j >> 6) * C + h * CH + (j & 63) : h * (3 * CH) + j; };

  u32x4 xr[NCH][QB];
  // quarter j of an image's x fragments (chunks j NCH/4 ...): the next image's quarters are issued one per key tile of the attention
  // phase - all 16 loads of all 8 waves at once queue up behind the texture addresser (row stride 2C bytes: 16 segments each) and
  // every wave sat 2-5k cycles in the issue
  auto load_x = [&](int n, int lr, int lq, auto jc) {
    constexpr int J = decltype(jc)::value;
    const T* xb = reinterpret_cast<const T*>(p.x) + (size_t)n * TT * C;
#pragma unroll
    for (int c = J * (NCH / 4); c < (J + 1) * (NCH / 4); ++c)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) xr[c][qb] = *reinterpret_cast<const u32x4*>(xb + (size_t)(tok0 + 16 * qb + lr) * C + c * CHUNK + lq * V);
  };
  load_x(ilane, lr0, lq0, IC3<0>()); load_x(ilane, lr0, lq0, IC3<1>()); load_x(ilane, lr0, lq0, IC3<2>()); load_x(ilane, lr0, lq0, IC3<3>());
  {   // the head's weight rows, once (copied verbatim: see attn_fused_kernel)
    uint32_t wso[2]; int wdo[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, j = min(e >> 2, 191), r = grow(j);
      wso[u] = ((uint32_t)(r >> 7) * NCH) * 8192u + (uint32_t)(r & 127) * 64u + (e & 3) * 16;
      wdo[u] = j * 64 + (e & 3) * 16;
    }
    const bool second = tid + 512 < 768;
    const char* wsrc = reinterpret_cast<const char*>(p.w);
#pragma unroll
    for (int cl = 0; cl < NCH; cl += 4) {
      u32x4 t[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t[k][0] = *reinterpret_cast<const u32x4*>(wsrc + wso[0] + (size_t)(cl + k) * 8192);
        t[k][1] = *reinterpret_cast<const u32x4*>(wsrc + wso[1] + (size_t)(cl + k) * 8192);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<u32x4*>(wlds + (cl + k) * WBUF + wdo[0]) = t[k][0];
        if (second) *reinterpret_cast<u32x4*>(wlds + (cl + k) * WBUF + wdo[1]) = t[k][1];
      }
    }
  }
  if (tid < 2 * C) ablds[tid] = tid < C ? p.ga[(size_t)ilane * C + tid] : p.gb[(size_t)ilane * C + tid - C];
  float* biaslds = ablds + 4 * C;                      // the head's 192 bias values, row order
  if (tid < 192) biaslds[tid] = p.bias[grow(tid)];
  __syncthreads();

  const float c2 = p.scale2 * 1.4426950408889634f;
  int it = 0;
  // experiments only: cycle stamps of one wave (ATTN_ABLATE & 256), printed at the end
  constexpr bool STAMP = (ATTN_ABLATE & 256) != 0;
  unsigned long long ts[16] = {};
  auto stamp = [&](int i) { if constexpr (STAMP) { if (it == 1) { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } };
  for (int n = ilane; n < p.N; n += lanes, ++it) {
    const int nn = n + lanes;
    const bool more = nn < p.N;
    // every per-lane LDS / global offset below derives from these two, re-made opaque per image: hoisted out of the loop the two dozen
    // loop-invariant addresses do not fit the register file next to the fragments and come back as scratch reloads inside phase 1
    int lr = lr0, lq = lq0;
    asm volatile("" : "+v"(lr), "+v"(lq));
    float abn = 0.f;
    if (more && tid < 2 * C) abn = tid < C ? p.ga[(size_t)nn * C + tid] : p.gb[(size_t)nn * C + tid - C];
    const float* ab = ablds + (it & 1) * 2 * C;
    stamp(0);

    // ---------------- phase 1: q | k | v rows of this head for the wave's tokens; weights resident, x already in registers ----------------
    // (a) GroupNorm affine on all x fragments (the attention norm has no SiLU, unet.py:379,397); (b) three passes over the head's rows -
    // q, k, v (4 row tiles each) - so only 8 accumulator quads are live and the weight fragments run a whole chunk (4
    // ds_read_b128) ahead of their MFMAs: the reads of chunk c + 1 are issued between the MFMA pairs of chunk c (explicitly
    // software-pipelined + sched_group_barrier: left alone, the scheduler keeps two reads in flight and every MFMA pair waits for LDS).
    u32x4 xf[NCH][QB];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float av[V], bv[V];
#pragma unroll
      for (int j = 0; j < V; j += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ab + c * CHUNK + lq * V + j);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(ab + C + c * CHUNK + lq * V + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) { av[j + k] = a4[k]; bv[j + k] = b4[k]; }
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float f[V];
        frag_to_float(xr[c][qb], f, T());
#pragma unroll
        for (int j = 0; j < V; ++j) f[j] = av[j] * f[j] + bv[j];
        xf[c][qb] = (ATTN_ABLATE & 64) ? xr[c][qb] : float_to_frag(f, T());
      }
    }
    stamp(1);
    const char* wb0 = wlds + lr * 64 + 16 * (lq ^ ((lr >> 1) & 3));
    u32x4 qf[QB][KST];
    u32x2 kpk[QB][4], vpk[QB][4];
    auto rows_pass = [&](auto t0c, auto ntc, auto&& done) {
      constexpr int T0 = decltype(t0c)::value, NT = decltype(ntc)::value;
      __builtin_amdgcn_sched_barrier(0);     // the pass is one scheduling region: the groups below count ITS reads and MFMAs only
      f32x4 acc[NT][QB];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) acc[ct][qb] = *reinterpret_cast<const f32x4*>(biaslds + (T0 + ct) * 16 + 4 * lq);
      u32x4 wcur[NT], wnxt[NT];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) wcur[ct] = *reinterpret_cast<const u32x4*>(wb0 + (T0 + ct) * 1024);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);   // bias rows + the first chunk's fragments
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          if (c + 1 < NCH) wnxt[ct] = *reinterpret_cast<const u32x4*>(wb0 + (c + 1) * WBUF + (T0 + ct) * 1024);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) mma16(acc[ct][qb], wcur[ct], xf[c][qb], T());
        }
        if (c + 1 < NCH) {
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read ...
            __builtin_amdgcn_sched_group_barrier(0x008, QB, 0);  // ... then the MFMAs of one row tile
          }
#pragma unroll
          for (int ct = 0; ct < NT; ++ct) wcur[ct] = wnxt[ct];
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, QB * NT, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      done(acc);
    };
    rows_pass(IC3<0>(), IC3<4>(), [&](f32x4 (&acc)[4][QB]) {
      // q -> B-operand fragments, scaled by ch^-1/2 * log2(e): S^T comes out of the MFMAs in the log2 domain
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
          const f32x4 s0 = acc[2 * ks][qb], s1 = acc[2 * ks + 1][qb];
          const u32x2 t0 = pack4(f32x4{s0[0] * c2, s0[1] * c2, s0[2] * c2, s0[3] * c2}, T()), t1 = pack4(f32x4{s1[0] * c2, s1[1] * c2, s1[2] * c2, s1[3] * c2}, T());
          qf[qb][ks] = u32x4{t0[0], t0[1], t1[0], t1[1]};
        }
    });
    stamp(2);
    rows_pass(IC3<4>(), IC3<4>(), [&](f32x4 (&acc)[4][QB]) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          kpk[qb][ct] = pack4(acc[ct][qb], T());
    });
    stamp(3);
    rows_pass(IC3<8>(), IC3<4>(), [&](f32x4 (&acc)[4][QB]) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          vpk[qb][ct] = pack4(acc[ct][qb], T());
    });
    stamp(4);
    if (more && tid < 2 * C) ablds[((it + 1) & 1) * 2 * C + tid] = abn;   // read two barriers later at the earliest
    auto write_kv = [&](int key0) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int key = key0 + 16 * qb + lr;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          // K: the 16-byte slot (ks, lq) holds tile 2 ks (first 8 bytes) and tile 2 ks + 1 (last 8) = the q fragments' channel order
          *reinterpret_cast<u32x2*>(klds + key * ROW + (ct >> 1) * 64 + lq * 16 + (ct & 1) * 8) = kpk[qb][ct];
          *reinterpret_cast<u32x2*>(vlds + key * ROW + (16 * ct + 4 * lq) * 2) = vpk[qb][ct];   // V: natural [key][channel]
        }
      }
    };
    // ---------------- phase 3: attention over the key buffer, two rounds of 128 keys ----------------
    f32x4 o[QB][CI];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) o[qb][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The reference offset m_run of a query column enters as the START value of its S^T accumulators, so p = exp2(sacc) with no
    // multiply-add per element; it moves only when a tile's maximum exceeds it by more than 8 (p <= 2^8), and then the tile is shifted
    // and O / l rescaled on a wave-uniform slow path.  The very first tile always takes that path (its offset is the tile maximum).
    // The column maximum is reduced over the four lanes that share a query with two VALU row swaps (v_permlane16_swap /
    // v_permlane32_swap) instead of two LDS round trips.
    auto colmax = [&](float v) {
      const auto s16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, v), false, false);
      v = fmaxf(__builtin_bit_cast(float, (uint32_t)s16[0]), __builtin_bit_cast(float, (uint32_t)s16[1]));
      const auto s32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, v), __builtin_bit_cast(uint32_t, v), false, false);
      return fmaxf(__builtin_bit_cast(float, (uint32_t)s32[0]), __builtin_bit_cast(float, (uint32_t)s32[1]));
    };
    auto attend = [&](auto roundc) {
      constexpr int RND = decltype(roundc)::value;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const char* kb = klds + kt * 64 * ROW;
        const char* vb = vlds + kt * 64 * ROW;
        const bool first = RND == 0 && kt == 0;
        if (more && !(ATTN_ABLATE & 4)) { if (kt == 0) load_x(nn, lr, lq, IC3<2 * RND>()); else load_x(nn, lr, lq, IC3<2 * RND + 1>()); }   // x fragments are dead since phase 1
        f32x4 sacc[QB][4];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) sacc[qb][mi] = f32x4{-m_run[qb], -m_run[qb], -m_run[qb], -m_run[qb]};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ks = 0; ks < KST; ++ks) {
            const u32x4 kf = (ATTN_ABLATE & 16) ? u32x4{(uint32_t)lane << 8, 0x3c003c00u, (uint32_t)(mi + kt), 0x3c003c00u} : *reinterpret_cast<const u32x4*>(kb + (mi * 16 + lr) * ROW + (ks * 4 + lq) * 16);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) mma16(sacc[qb][mi], kf, qf[qb][ks], T());
          }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          float mx = fmaxf(fmaxf(sacc[qb][0][0], sacc[qb][0][1]), fmaxf(sacc[qb][0][2], sacc[qb][0][3]));
#pragma unroll
          for (int mi = 1; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qb][mi][r]);
          mx = colmax(mx);
          const float delta = first ? mx : (mx > 8.0f ? mx : 0.0f);
          if (first || __builtin_amdgcn_ballot_w64(delta != 0.0f) != 0) {   // wave-uniform: some column's offset moves
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            m_run[qb] += delta;
            l_run[qb] *= alpha;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) sacc[qb][mi][r] -= delta;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
              for (int r = 0; r < 4; ++r) o[qb][ci][r] *= alpha;
          }
          float psum = 0.f;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float pv = (ATTN_ABLATE & 8) ? sacc[qb][mi][r] : __builtin_amdgcn_exp2f(sacc[qb][mi][r]); sacc[qb][mi][r] = pv; psum += pv; }
          l_run[qb] += psum;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          u32x4 pfrag[QB];
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const u32x2 p0 = pack4(sacc[qb][2 * s2], T()), p1 = pack4(sacc[qb][2 * s2 + 1], T());
            pfrag[qb] = u32x4{p0[0], p0[1], p1[0], p1[1]};
          }
          const char* vrow = vb + (32 * s2 + 4 * lq + (lr >> 2)) * ROW + 8 * (lr & 3);
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + ci * 32));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 16 * ROW + ci * 32));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            const u32x4 vf = (ATTN_ABLATE & 16) ? u32x4{(uint32_t)lane << 8, 0x3c003c00u, (uint32_t)(ci + kt), 0x3c003c00u} : u32x4{l2[0], l2[1], h2[0], h2[1]};
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) mma16(o[qb][ci], vf, pfrag[qb], T());
          }
        }
      }
    };
    stamp(5);
    __syncthreads();                       // D: every wave is done with the previous image's second round
    stamp(6);
    if (wave < 4) write_kv(tok0);
    __syncthreads();                       // A: keys 0-127 are in place
    stamp(7);
    if constexpr (!(ATTN_ABLATE & 1)) attend(IC3<0>());
    stamp(8);
    __syncthreads();                       // B: every wave is done with keys 0-127
    stamp(9);
    if (wave >= 4) write_kv(tok0 - 128);
    __syncthreads();                       // C: keys 128-255 are in place
    stamp(10);
    if constexpr (!(ATTN_ABLATE & 1)) attend(IC3<1>());
    stamp(11);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float l = l_run[qb];
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = 1.0f / l;
      T* op = reinterpret_cast<T*>(p.out) + ((size_t)n * TT + tok0 + 16 * qb + lr) * C + h * CH;
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        *reinterpret_cast<u32x2*>(op + ci * 16 + 4 * lq) = pack4(f32x4{o[qb][ci][0] * inv, o[qb][ci][1] * inv, o[qb][ci][2] * inv, o[qb][ci][3] * inv}, T());
      }
    }
    stamp(12);
  }
  if constexpr (STAMP) {
    if (blockIdx.x == 8 && (tid == 0 || tid == 320))
      printf("[attn stamps] wave %d: affine %llu q %llu k %llu v %llu pack %llu barD %llu kvw+barA %llu att1 %llu barB %llu kvw+barC %llu att2 %llu store %llu | item %llu cycles\n",
             tid >> 6, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[8] - ts[7], ts[9] - ts[8],
             ts[10] - ts[9], ts[11] - ts[10], ts[12] - ts[11], ts[12] - ts[0]);
  }
}

template <int NCH, typename T>
int launch_fused_pers(const AttnFuseArgs& a, int lanes, hipStream_t s) {
  auto kern = attn_fused_pers_kernel<NCH, T>;
  constexpr size_t lds = (size_t)NCH * 192 * 64 + 2 * 128 * (64 * 2 + 32) + (size_t)4 * NCH * 32 * 4 + 192 * 4;
  static_assert(lds <= 160 * 1024, "persistent attention block: LDS budget");
  if (int rc = mi355_allow_big_lds(kern, "attention block (persistent)")) return rc;
  const int grid = 8 * a.heads * ((lanes + 7) / 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, lanes);
  return 0;
}

template <typename T, int QB>
int launch_fused(const AttnFuseArgs& a, hipStream_t s) {
  auto kern = attn_fused_kernel<T, QB>;
  constexpr size_t kv = (size_t)2 * 128 * QB * (64 * sizeof(T) + 32);
  const size_t ph1 = (size_t)a.stage_chunks * 192 * 64 + (size_t)2 * a.C * 4;   // the resident weight chunks + the (a, b) table
  const size_t lds = kv > ph1 ? kv : ph1;
  if (lds > 160 * 1024) { mi355_set_error("attention block: LDS budget exceeded"); return -4; }
  if (lds > 64 * 1024) { if (int rc = mi355_allow_big_lds(kern, "attention block")) return rc; }
  const int grid = ((a.N + 7) / 8) * 8 * a.heads;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
  return 0;
}

}  // namespace

bool attn_fused_eligible(int dtype, int T, int C, int heads, int ch, const mi355_debug_config* knobs) {
  const int enabled = (knobs ? knobs : &mi355_default_debug())->attn_fused;
  const int CHUNK = dtype == 0 ? 16 : 32;
  (void)CHUNK;
  return (enabled & 1) && ch == 64 && heads * ch == C && (T == 128 || T == 256) && C % 128 == 0 && C <= 512;   // 3C % 128 == 0: 128-row weight tiles
}

int attn_fused_launch(const AttnFusedDesc& d, hipStream_t stream) {
  MI355_REQUIRE(attn_fused_eligible(d.dtype, d.T, d.C, d.heads, d.ch, d.knobs), -4, "attention block: shape not supported by the fused kernel");
  MI355_REQUIRE(d.x && d.ga && d.gb && d.w && d.bias && d.out, -1, "attention block: null argument");
  AttnFuseArgs a;
  a.x = d.x; a.ga = d.ga; a.gb = d.gb; a.w = d.w; a.bias = d.bias; a.out = d.out;
  a.N = d.N; a.T = d.T; a.C = d.C; a.heads = d.heads; a.nchunks = d.C / (d.dtype == 0 ? 16 : 32); a.new_order = d.new_order;
  a.scale2 = 1.0f / sqrtf((float)d.ch);
  // resident weight chunks per stage: all of them if they fit beside the (a, b) table, else the largest multiple of 3 that does
  {
    const int fit = (int)((160 * 1024 - (size_t)8 * d.C) / (192 * 64));
    a.stage_chunks = a.nchunks <= fit ? a.nchunks : fit / 3 * 3;
  }
  int rc;
  // persistent form: bf16, 256 tokens, the head's weight rows + a 128-key buffer fit the LDS (C <= 256), and enough (image, head) pairs
  // that a workgroup visits more than one image (knob attn_fused: bit 1 = off, bits 8.. = image lanes override, tests only)
  const int knob = (d.knobs ? d.knobs : &mi355_default_debug())->attn_fused;
  int lanes = 0;
  if (d.dtype != DT_F32 && d.T == 256 && (d.C == 128 || d.C == 256) && !(knob & 2)) {
    static const int cus = [] { int dev = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount; return n; }();
    lanes = std::max(8, cus / d.heads / 8 * 8);
    if (knob >> 8) lanes = knob >> 8;
    if (!(knob >> 8) && d.N < 2 * lanes) lanes = 0;   // fewer than two images per workgroup: nothing to amortise
    if (lanes > d.N) lanes = d.N;
  }
  if (lanes > 0 && d.dtype == DT_F16) rc = d.C == 256 ? launch_fused_pers<8, f16>(a, lanes, stream) : launch_fused_pers<4, f16>(a, lanes, stream);
  else if (lanes > 0) rc = d.C == 256 ? launch_fused_pers<8, bf16>(a, lanes, stream) : launch_fused_pers<4, bf16>(a, lanes, stream);
  else rc = dispatch_dtype(d.dtype, [&](auto t) { using T = decltype(t); return d.T == 256 ? launch_fused<T, 2>(a, stream) : launch_fused<T, 1>(a, stream); });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
