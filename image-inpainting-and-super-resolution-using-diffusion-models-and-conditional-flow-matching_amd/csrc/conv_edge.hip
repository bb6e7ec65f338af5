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
};

// GN affine (+ SiLU) of one 16-byte fragment.  bf16: fp32 math on element PAIRS (v_pk_fma / v_pk_mul / v_pk_add_f32: two elements per VALU
// issue; the pass is VALU-bound), the operation sequence of conv_ws.inc.h's ws_pro_frag.
typedef __bf16 ebf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 edge_pro_frag(const u32x4& raw, const f32x2 (&a2)[4], const f32x2 (&b2)[4], bool silu, bf16) {
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = raw[i];
    const f32x2 x = f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
    f32x2 v = a2[i] * x + b2[i];
    if (silu) {
      const f32x2 sc = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
      const f32x2 d = f32x2{__builtin_amdgcn_exp2f(sc[0]), __builtin_amdgcn_exp2f(sc[1])} + f32x2{1.0f, 1.0f};
      v = v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    }
    const ebf16x2 h = ebf16x2{(bf16)v[0], (bf16)v[1]};
    out[i] = __builtin_bit_cast(uint32_t, h);
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
          if (j < p.Cout) p.out[(((size_t)n0 * p.Cout + j) * p.H + y) * p.W + x] = acc[mi][j] + (p.bias ? p.bias[j] : 0.f);
      }
    }
  }
}

}  // namespace

// 0 = launched, 1 = not eligible (the caller goes on to the generic kernel), < 0 = error.  Switch: mi355_debug_config::conv_edge.
int conv_out_try_launch(const ConvDesc& d, hipStream_t stream) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  if (!(K.conv_edge & 1)) return 1;
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
  if (d.dtype == 0) rc = fixed ? go(conv3x3_out_kernel<float, true>) : go(conv3x3_out_kernel<float, false>);
  else rc = fixed ? go(conv3x3_out_kernel<bf16, true>) : go(conv3x3_out_kernel<bf16, false>);
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  return 0;
}
