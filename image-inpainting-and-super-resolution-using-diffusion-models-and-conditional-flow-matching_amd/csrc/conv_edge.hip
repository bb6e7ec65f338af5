// The U-Net's last convolution as a streaming kernel: out = conv3x3(SiLU(GroupNorm32(h))), C_in -> out_channels <= 4, NCHW fp32 output
// (UNetModel.out, AD/image_diffusion/unet.py:702-706; `zero_module(conv_nd(...))`).
//
// With N = 3 this is not a GEMM: 1.8 GFLOP against 67 MB of activations at cfg 2 (27 FLOP per byte) - an HBM-bound pass with a little
// arithmetic.  The generic implicit-GEMM kernel ran it as 2048 workgroups of 128 pixels x 32 (padded) channels with a barrier per kernel
// row and the prologue in its staging path: 47 us = 1.46 TB/s (128-px configuration: 611 us per evaluation).  Here:
//   * a workgroup (4 waves) owns 8 rows x 16 columns of one image and ALL input channels: the haloed patch (10 x 18 pixels x C_in) goes
//     global -> LDS in one DMA burst (46 KB at 128 bf16 channels: every byte of the tile in flight at once), three workgroups per CU, so one's
//     burst overlaps another's arithmetic and a third's stores;
//   * GroupNorm affine + SiLU is applied IN PLACE in LDS, once per patch element (a thread owns one 16-byte channel fragment, its (a, b) in
//     registers); out-of-image pixels stay the zeros the DMA wrote (zero padding applies to the activated tensor);
//   * the contraction is 72 MFMAs per wave (rows = output channels, of which <= 4 are real; columns = 16 pixels of a row): A fragments from
//     the patch (16-byte slots XOR-swizzled by the pixel's column: conflict-free for every tap, row shifts are plain offsets), B fragments
//     from the first rows of the packed weight tiles, copied once into LDS;
//   * lanes 0-15 of every wave hold the real output channels of 16 consecutive pixels: 64-byte NCHW fp32 segments, bias added in flight.
#include "ops.h"

namespace {

constexpr int OT_W = 16, OT_H = 8, OP_W = OT_W + 2, OP_H = OT_H + 2, OP_N = OP_W * OP_H;   // 180 patch pixels

struct OutArgs {
  const void* src; int C, N, H, W;
  const float* pro_a; const float* pro_b; int pro_silu;
  const void* w; const float* bias; int Cout;
  float* out;
  uint32_t bytes_src, wbytes;
  int tiles_x, tiles_y, ntiles;
  float* axpy_x; float axpy_scale;   // non-null: x += scale * v instead of storing v (same element order as out)
};

// GN affine (+ SiLU) of one 16-byte fragment.  bf16: fp32 math on element PAIRS (v_pk_fma / v_pk_mul / v_pk_add_f32: two elements per VALU
// issue; the pass is VALU-bound), the operation sequence of conv_ws.inc.h's ws_pro_frag.
typedef __bf16 ebf16x2 __attribute__((ext_vector_type(2)));
template <typename T2>   // bf16 / f16
__device__ __forceinline__ u32x4 edge_pro_frag(const u32x4& raw, const f32x2 (&a2)[4], const f32x2 (&b2)[4], bool silu, T2) {
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = raw[i];
    f32x2 x;
    { float xl, xh; unpack2(w, xl, xh, T2()); x = f32x2{xl, xh}; }
    f32x2 v = a2[i] * x + b2[i];
    if (silu) {
      const f32x2 sc = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
      const f32x2 d = f32x2{__builtin_amdgcn_exp2f(sc[0]), __builtin_amdgcn_exp2f(sc[1])} + f32x2{1.0f, 1.0f};
      v = v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    }
    out[i] = pack2(v[0], v[1], T2());
  }
  return out;
}
__device__ __forceinline__ u32x4 edge_pro_frag(const u32x4& raw, const f32x2 (&a2)[2], const f32x2 (&b2)[2], bool silu, float) {
  const f32x4 x = __builtin_bit_cast(f32x4, raw);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = a2[j >> 1][j & 1] * x[j] + b2[j >> 1][j & 1];
    o[j] = silu ? u / (1.0f + expf(-u)) : u;
  }
  return __builtin_bit_cast(u32x4, o);
}

// FIXED: 16 fragments = 256 bytes per pixel (128 bf16 / 64 fp32 channels: the U-Nets of every BASELINE configuration): the kernel is bound by
// its VALU instruction count (address arithmetic), and as literals the divisions, row offsets and loop bounds fold into immediates.
// Measured against it (profiles/r4_experiments.md; tools/experiments/r4_conv_out_persistent_triple_buffer_variant.hip.txt): one persistent
// 8-wave workgroup per CU with a triple-buffered patch ring and counted vmcnt - 42 us instead of 36 at cfg 2, 289 instead of 240 at 128 px:
// two waves per SIMD do not hide the LDS round trips that twelve do.
template <typename T, bool FIXED>
__global__ void __launch_bounds__(256) conv3x3_out_kernel(OutArgs p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, ESZ = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int CC = FIXED ? 16 * V : p.C;       // input channels
  const int FPP = FIXED ? 16 : p.C / V;      // 16-byte fragments per pixel (16, 32 or 64: the launcher)
  const int PXB = FIXED ? 256 : p.C * ESZ;   // bytes per pixel
  const int nch = FIXED ? 4 : p.C / CHUNK;
  char* patch = smem;                        // [180 px][PXB], slot s of a pixel holds fragment s ^ (column & 15)
  char* wl = smem + OP_N * PXB;              // [chunk][tap][NR rows][64 B]: the first rows of the packed weight tiles (their slot swizzle kept)
  const int NR = p.Cout <= 3 ? 3 : 4;        // (three rows at out_channels <= 3: 53 KB per workgroup, three workgroups per CU)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware tile order: workgroups b, b + 8, ... share an L2, and take consecutive tiles (the tiles of one image share halo rows)
  int t;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  }
  const int tpi = p.tiles_x * p.tiles_y;
  const int n0 = t / tpi, rem = t - n0 * tpi;
  const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
  const int y0 = tyi * OT_H, x0 = txi * OT_W;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, p.bytes_src, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- the whole haloed patch by DMA: a piece = 1 KB = 64 / FPP pixels; lane l -> pixel l / FPP of the piece, slot l % FPP ----
  const int ppp = FIXED ? 4 : 64 / FPP;      // pixels per piece; 180 pixels = a whole number of pieces for 1, 2 and 4
  const int npieces = OP_N / ppp;
  for (int j = wave; j < npieces; j += 4) {
    const int px = j * ppp + lane / FPP, s = lane % FPP;
    const int py = (int)(((float)px + 0.5f) * (1.0f / (float)OP_W)), pc = px - py * OP_W;   // exact: px < 180
    const int y = y0 - 1 + py, x = x0 - 1 + pc;
    const int f = (s & ~15) | ((s ^ pc) & 15);
    const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
    const uint32_t vo = ok ? (uint32_t)((n0 * p.H + y) * p.W + x) * (uint32_t)PXB + (uint32_t)f * 16u : p.bytes_src;   // out of range: the DMA writes zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + j * 1024), 16, vo, 0, 0, 0);
  }
  // ---- weights: rows 0 .. NR - 1 of every (chunk, tap) tile of the packed image ([chunk][tap][32 rows][64 B]) ----
  for (int i = tid; i < nch * 9 * NR * 4; i += 256) {
    const int tile = i / (NR * 4), rr = i - tile * (NR * 4);
    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (uint32_t)(tile * 2048 + rr * 16), 0, 0));
    *reinterpret_cast<u32x4*>(wl + i * 16) = v;
  }
  // ---- this thread's (a, b): it owns channel fragment fo (+ 16 k) of every pixel it transforms (fetched beside the burst) ----
  const int fo = tid & 15;
  f32x2 a2[V / 2], b2[V / 2];
#pragma unroll
  for (int j = 0; j < V / 2; ++j) {
    a2[j] = *reinterpret_cast<const f32x2*>(p.pro_a + (size_t)n0 * CC + fo * V + 2 * j);
    b2[j] = *reinterpret_cast<const f32x2*>(p.pro_b + (size_t)n0 * CC + fo * V + 2 * j);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- GroupNorm affine + SiLU in place ----
  for (int fb = 0; fb < FPP; fb += 16) {
    if (fb > 0) {
#pragma unroll
      for (int j = 0; j < V / 2; ++j) {
        a2[j] = *reinterpret_cast<const f32x2*>(p.pro_a + (size_t)n0 * CC + (fb + fo) * V + 2 * j);
        b2[j] = *reinterpret_cast<const f32x2*>(p.pro_b + (size_t)n0 * CC + (fb + fo) * V + 2 * j);
      }
    }
    // twelve fragments per thread (pixels tid / 16 + 16 m), three at a time: independent LDS round trips overlap inside the wave
    for (int m0 = 0; m0 < 12; m0 += 3) {
      char* q[3]; bool ok[3]; u32x4 v[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int px = (tid >> 4) + 16 * (m0 + m);
        const int py = (int)(((float)px + 0.5f) * (1.0f / (float)OP_W)), pc = px - py * OP_W;   // exact: px < 192
        const int y = y0 - 1 + py, x = x0 - 1 + pc;
        ok[m] = px < OP_N && y >= 0 && y < p.H && x >= 0 && x < p.W;      // zero padding stays zero
        q[m] = patch + (px < OP_N ? px : 0) * PXB + (fb + ((fo ^ pc) & 15)) * 16;
        v[m] = *reinterpret_cast<const u32x4*>(q[m]);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const u32x4 o = edge_pro_frag(v[m], a2, b2, p.pro_silu != 0, T());
        if (ok[m]) *reinterpret_cast<u32x4*>(q[m]) = o;
      }
    }
  }
  __syncthreads();

  // ---- contraction: wave w owns tile rows 2 w, 2 w + 1; D rows = output channels (4 lq + j), columns = the 16 pixels of a row; a chunk's
  //      27 fragments are read before its 18 MFMAs ----
  const int lr = lane & 15, lq = lane >> 4;
  // fused Euler update: this lane's state values are requested now and used after the contraction (no other workgroup touches these pixels)
  float xs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (p.axpy_x && lq == 0) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int y = y0 + 2 * wave + mi, x = x0 + lr;
      if (y < p.H && x < p.W) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < p.Cout) xs[mi][j] = p.axpy_x[(((size_t)n0 * p.Cout + j) * p.H + y) * p.W + x];
      }
    }
  }
  const int co = lr < p.Cout ? lr : 0;       // lanes of unused accumulator rows read row 0 (their results are never stored)
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  int abase[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) abase[kx] = (2 * wave * OP_W + lr + kx) * PXB;
  const int bbase = co * 64 + 16 * (lq ^ ((co >> 1) & 3));
  for (int c = 0; c < nch; ++c) {           // (FIXED: four trips, unrolled by the compiler)
    u32x4 bf[9], af[2][9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int fA = c * 4 + lq;             // fragment index of (chunk c, k-slot lq)
      const int slot = (fA & ~15) | ((fA ^ (lr + kx)) & 15);
      bf[tap] = *reinterpret_cast<const u32x4*>(wl + (c * 9 + tap) * NR * 64 + bbase);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi][tap] = *reinterpret_cast<const u32x4*>(patch + abase[kx] + (mi + ky) * OP_W * PXB + slot * 16);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) mma16(acc[mi], bf[tap], af[mi][tap], T());
  }
  // ---- NCHW fp32 stores: lanes lq == 0 hold channels 0 .. 3 of pixel lr ----
  if (lq == 0) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int y = y0 + 2 * wave + mi, x = x0 + lr;
      if (y < p.H && x < p.W) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < p.Cout) {
            const size_t oi = (((size_t)n0 * p.Cout + j) * p.H + y) * p.W + x;
            const float v = acc[mi][j] + (p.bias ? p.bias[j] : 0.f);
            if (p.axpy_x) {
#pragma clang fp contract(off)   // (the expression of steps.hip euler_step_launch, rounded the same way)
              p.axpy_x[oi] = xs[mi][j] + p.axpy_scale * v;
            } else p.out[oi] = v;
          }
      }
    }
  }
}


// ---- the network's FIRST conv (unet.py:575: conv_nd(dims, in_channels, model_channels, 3, padding=1)): K = 9 taps x 3 ... 8 real channels.
// The input tensor is NHWC with its channels padded to one 64-byte chunk (pack_nhwc); only the first 16-byte slot of a pixel holds real
// channels (cin_real <= 8) and the packed weights of the other slots are zero, so the contraction runs over (tap, 8 channels): 72 -> 96
// = three 32-deep MFMA steps of four taps each (taps 9-11 are zero rows) instead of nine steps that are 90 % padding.  One workgroup =
// one 16x16 pixel tile x 128 output channels, 4 waves x 4 tile rows; what is left is the 256-byte-per-pixel output stream (16-byte
// stores via v_permlane16_swap) and the GroupNorm partial sums of the output (common.h GnPartial; slot = one wave's 64 pixels).
struct InArgs {
  const void* src; const void* w; const float* bias; void* out;
  float* gn_stats; int gn_slots;
  int N, H, W, tiles_x, tiles_y;
  const float* x0; const float* x1; int c0, c1;   // x0 non-null: the source is cat(x0, x1) in fp32 NCHW (c0 + c1 <= 8 channels), src unused
};

template <bool GN, typename T = bf16>   // T: bf16 or f16
__global__ void __launch_bounds__(256, 4) conv3x3_in_kernel(InArgs p) {
  constexpr int NT = 8;                       // 16-channel tiles of the 128 output channels
  constexpr int PW = 18, NPX = PW * PW;       // haloed patch
  __shared__ __attribute__((aligned(16))) char wl[12 * 128 * 16];       // [tap (9 real + 3 zero)][row][channels 0-7]
  __shared__ __attribute__((aligned(16))) char xl[(NPX + 1) * 16];      // [patch pixel][channels 0-7] + one zero slot
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr0 = lane & 15, lq0 = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int n = blockIdx.x / tpi, trem = blockIdx.x - n * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;

  // weights: slot 0 of every (tap, row) of the packed image ([tap][128 rows][64 B], slots swizzled by (row >> 1) & 3)
  for (int e = tid; e < 12 * 128; e += 256) {
    const int tap = e >> 7, row = e & 127;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (tap < 9) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w) + (size_t)tap * 8192 + row * 64 + 16 * ((row >> 1) & 3));
    *reinterpret_cast<u32x4*>(wl + e * 16) = v;
  }
  // patch: slot 0 of every pixel of the haloed tile (zero outside the image)
  for (int q = tid; q < NPX + 1; q += 256) {
    const int py = q / PW, px = q - py * PW;
    const int gy = ty * 16 + py - 1, gx = tx * 16 + px - 1;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (q < NPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
      if (p.x0) {   // what pack_nhwc would have written to this pixel's first slot: the real channels, rounded to T, zeros behind them
        float f[8];
        const size_t hw = (size_t)p.H * p.W, px0 = (size_t)gy * p.W + gx;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          f[c] = c < p.c0 ? p.x0[((size_t)n * p.c0 + c) * hw + px0] : (c < p.c0 + p.c1 ? p.x1[((size_t)n * p.c1 + (c - p.c0)) * hw + px0] : 0.f);
        v = float_to_frag(f, T());
      } else {
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.src) + (((size_t)n * p.H + gy) * p.W + gx) * 64);
      }
    }
    *reinterpret_cast<u32x4*>(xl + q * 16) = v;
  }
  __syncthreads();

  float gs[NT], gq[NT];     // GroupNorm partial sums of this wave's 64 pixels, per channel tile (GnPartial's layout, 16 registers instead of 32)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { gs[nt] = 0.f; gq[nt] = 0.f; }
  constexpr bool do_gn = GN;   // (a run-time flag costs a branch and a phi per store group)
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int r0 = wave * 4 + half * 2;                 // two tile rows at a time: 64 accumulator registers, four workgroups per CU
    // lane ids re-made opaque per half: hoisted out of this loop, the 24 weight fragments (96 registers) and a dozen offsets would
    // halve the occupancy or come back as scratch reloads whose vmcnt waits also wait for the previous half's stores
    int lr = lr0, lq = lq0;
    asm volatile("" : "+v"(lr), "+v"(lq));
    const int wofs = (lq * 128 + lr) * 16;
    const int co_s = (lq & 1) * 16 + (lq >> 1) * 8;     // channel of this lane's 16-byte store within a pair of channel tiles
    // lane (lr, lq) of k-step ks holds tap 4 ks + lq: weights of row nt * 16 + lr (A operand), pixel (r + ky, lr + kx) (B operand)
    int boff[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int tap = 4 * ks + lq, ky = tap / 3, kx = tap - 3 * ky;
      boff[ks] = tap < 9 ? (ky * PW + lr + kx) * 16 : NPX * 16;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + nt * 16 + 4 * lq);
      acc[0][nt] = b4; acc[1][nt] = b4;
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      u32x4 bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) bf[mi] = *reinterpret_cast<const u32x4*>(xl + boff[ks] + (boff[ks] == NPX * 16 ? 0 : (r0 + mi) * PW * 16));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 af = *reinterpret_cast<const u32x4*>(wl + wofs + (4 * ks * 128 + nt * 16) * 16);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) mma16(acc[mi][nt], af, bf[mi], T());
      }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int gy = ty * 16 + r0 + mi, gx = tx * 16 + lr;
      char* op = reinterpret_cast<char*>(p.out) + ((((size_t)n * p.H + gy) * p.W + gx) * 128 + co_s) * 2;
#pragma unroll
      for (int k = 0; k < NT / 2; ++k) {
        const f32x4 va = acc[mi][2 * k], vb = acc[mi][2 * k + 1];
        if constexpr (do_gn) {
          gs[2 * k] += (va[0] + va[1]) + (va[2] + va[3]);
          gq[2 * k] = __builtin_fmaf(va[0], va[0], __builtin_fmaf(va[1], va[1], __builtin_fmaf(va[2], va[2], __builtin_fmaf(va[3], va[3], gq[2 * k]))));
          gs[2 * k + 1] += (vb[0] + vb[1]) + (vb[2] + vb[3]);
          gq[2 * k + 1] = __builtin_fmaf(vb[0], vb[0], __builtin_fmaf(vb[1], vb[1], __builtin_fmaf(vb[2], vb[2], __builtin_fmaf(vb[3], vb[3], gq[2 * k + 1]))));
        }
        const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
        const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
        const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
        *reinterpret_cast<u32x4*>(op + k * 64) = u32x4{w0[0], w1[0], w0[1], w1[1]};
      }
    }
  }
  if constexpr (do_gn) {   // slot = (tile of the image, wave): this wave's 64 pixels x all 128 channels
    const int slot = trem * 4 + wave;
    // lane (lq, lr = nt) stores the pair of quad 4 nt + lq (GnPartial::store): 16-lane DPP row sums over the pixels
    float ms = 0.f, mq = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float ts = GnPartial<1>::row_sum(gs[nt]), tq = GnPartial<1>::row_sum(gq[nt]);
      if (lr0 == nt) { ms = ts; mq = tq; }
    }
    if (lr0 < NT) *reinterpret_cast<f32x2*>(p.gn_stats + (((size_t)n * p.gn_slots + slot) * (size_t)(128 >> 2) + (lr0 * 4 + lq0)) * 2) = f32x2{ms, mq};
  }
}

}  // namespace

// 0 = launched, 1 = not eligible (the caller goes on to the generic kernel), < 0 = error.  Switch: mi355_debug_config::conv_edge.
int conv_out_try_launch(const ConvDesc& d, hipStream_t stream) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  if (!(K.conv_edge & 1) || d.wsplit) return 1;   // (hi / lo split weights: the generic kernel)
  if (d.ks != 3 || d.mode != CONV_UNIT || d.out_mode != OUT_NCHW_F32 || d.src1 || d.res || d.emb || !d.pro_a || d.Cout > 4 || d.Cout < 1) return 1;
  const int V = d.dtype == 0 ? 4 : 8, esz = d.dtype == 0 ? 4 : 2;
  const int FPP = d.C0 / V;
  if (d.C0 % (16 * V) != 0 || FPP > 64 || (64 % FPP) != 0 || conv_tile_n(d.Cout) != 32) return 1;   // whole 16-fragment swizzle groups per pixel, whole pixels per 1-KB DMA piece
  const size_t lds = (size_t)OP_N * d.C0 * esz + (size_t)(d.C0 / (d.dtype == 0 ? 16 : 32)) * 9 * (d.Cout <= 3 ? 3 : 4) * 64;
  if (lds > 160 * 1024) return 1;
  OutArgs a;
  a.src = d.src0; a.C = d.C0; a.N = d.N; a.H = d.Hs; a.W = d.Ws;
  a.pro_a = d.pro_a; a.pro_b = d.pro_b; a.pro_silu = d.pro_silu;
  a.w = d.w; a.bias = d.bias; a.Cout = d.Cout; a.out = reinterpret_cast<float*>(d.out);
  a.axpy_x = (K.conv_edge & 8) ? d.axpy_x : nullptr; a.axpy_scale = d.axpy_scale;
  const size_t bs = (size_t)d.N * d.Hs * d.Ws * d.C0 * esz, wb = conv_packed_weight_bytes(d.dtype, d.Cout, d.C0, 3);
  MI355_REQUIRE(bs < 0xFFFF0000ull, -4, "conv (out): the source tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.bytes_src = (uint32_t)bs; a.wbytes = (uint32_t)wb;
  a.tiles_x = (d.Ws + OT_W - 1) / OT_W; a.tiles_y = (d.Hs + OT_H - 1) / OT_H; a.ntiles = d.N * a.tiles_x * a.tiles_y;
  const bool fixed = FPP == 16;
  auto go = [&](auto kern) -> int {
    if (int r = mi355_allow_big_lds(kern, "conv (out)")) return r;
    hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(256), lds, stream, a);
    return 0;
  };
  int rc;
  rc = dispatch_dtype(d.dtype, [&](auto t) { using T = decltype(t); return fixed ? go(conv3x3_out_kernel<T, true>) : go(conv3x3_out_kernel<T, false>); });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  if (a.axpy_x && d.axpy_done) *d.axpy_done = 1;
  return 0;
}

// The first conv (see conv3x3_in_kernel).  0 = launched, 1 = not eligible, < 0 = error.  Switch: mi355_debug_config::conv_edge bit 1.
static bool conv_in_eligible(const ConvDesc& d);
int conv_in_reads_nchw(const ConvDesc& d) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  return (d.nchw0 && (K.conv_edge & 4) && d.nchw_c0 >= 1 && d.nchw_c0 + d.nchw_c1 == d.cin_real && (d.nchw_c1 == 0 || d.nchw1) && conv_in_eligible(d)) ? 0 : 1;
}
static bool conv_in_eligible(const ConvDesc& d) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  if (!(K.conv_edge & 2) || d.wsplit) return false;
  if (d.dtype == DT_F32 || d.ks != 3 || d.mode != CONV_UNIT || d.out_mode != OUT_NHWC || d.src1 || d.res || d.emb || d.pro_a || d.act_out) return false;
  if (d.cin_real < 1 || d.cin_real > 8 || d.C0 != 32 || d.Cout != 128 || conv_tile_n(d.Cout) != 128 || !d.bias) return false;
  if (d.Hs % 16 != 0 || d.Ws % 16 != 0) return false;
  return true;
}
int conv_in_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  if (d.nchw0) MI355_REQUIRE(conv_in_reads_nchw(d) == 0, -5, "conv (in): this launch cannot read the fp32 NCHW input itself (ask conv_in_reads_nchw first)");
  if (!(K.conv_edge & 2) || d.wsplit) return 1;
  if (d.dtype == DT_F32 || d.ks != 3 || d.mode != CONV_UNIT || d.out_mode != OUT_NHWC || d.src1 || d.res || d.emb || d.pro_a || d.act_out) return 1;
  if (d.cin_real < 1 || d.cin_real > 8 || d.C0 != 32 || d.Cout != 128 || conv_tile_n(d.Cout) != 128 || !d.bias) return 1;
  if (d.Hs % 16 != 0 || d.Ws % 16 != 0) return 1;
  InArgs a;
  a.src = d.src0; a.w = d.w; a.bias = d.bias; a.out = d.out;
  a.x0 = d.nchw0; a.x1 = d.nchw_c1 ? d.nchw1 : nullptr; a.c0 = d.nchw_c0; a.c1 = d.nchw1 ? d.nchw_c1 : 0;
  a.N = d.N; a.H = d.Hs; a.W = d.Ws; a.tiles_x = d.Ws / 16; a.tiles_y = d.Hs / 16;
  const int slots = a.tiles_x * a.tiles_y * 4;
  a.gn_stats = nullptr; a.gn_slots = 0;
  if (d.gn_stats && slots <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = slots; }
  const dim3 gr(d.N * a.tiles_x * a.tiles_y);
  if (d.dtype == DT_F16) { if (a.gn_stats) hipLaunchKernelGGL((conv3x3_in_kernel<true, f16>), gr, dim3(256), 0, stream, a); else hipLaunchKernelGGL((conv3x3_in_kernel<false, f16>), gr, dim3(256), 0, stream, a); }
  else if (a.gn_stats) hipLaunchKernelGGL((conv3x3_in_kernel<true, bf16>), gr, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv3x3_in_kernel<false, bf16>), gr, dim3(256), 0, stream, a);
  MI355_CHECK_HIP(hipGetLastError());
  if (gn_slots_used) *gn_slots_used = a.gn_slots;
  return 0;
}
