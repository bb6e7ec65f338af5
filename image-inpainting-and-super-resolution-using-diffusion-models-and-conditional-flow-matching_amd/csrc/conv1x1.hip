// 1x1 convolution (GEMM) with a stationary activation tile: qkv / proj_out of AttentionBlock and the ResBlock
// skip_connection (AD/image_diffusion/unet.py:381,389,318).
//
// K (input channels) is small here (<= 512), so each workgroup stages its BM-pixel activation tile for ALL
// channel chunks once (GroupNorm prologue applied once, not once per output-channel tile), keeps it in LDS, and
// then walks the output-channel tiles of its N range streaming only weights (register-prefetched, double-buffered
// in LDS, one barrier per 3-chunk group).  Epilogue straight from the accumulators (rows = channels, cols = pixels).
#include "ops.h"

namespace {

constexpr int NT1 = 256;

struct K1Args {
  const void* src0; const void* src1;
  int C0, C1, Cin, nchunks;
  int N, H, W;
  const float* pro_a; const float* pro_b; int pro_silu;
  const void* w; const float* bias; int Cout;
  const float* emb; int emb_stride;
  const void* res;
  void* out;
  float* gn_stats; int gn_slots;     // fused GroupNorm partial sums of the output (common.h GnPartial), or null
  int lvw, lth, G, tiles_x, tiles_y, ntn;
  uint32_t bytes0, bytes1, wbytes, obytes;
};

template <bool FAST> __device__ __forceinline__ float silu1(float v) {
  if (FAST) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  return v / (1.0f + expf(-v));
}
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// GC: channel chunks per weight group (one barrier per group).  GC = 1 shrinks the weight double buffer to 16 KB so that a 128-pixel
// tile of 256 input channels still fits twice per CU: half the weight bytes streamed from L2 per FLOP of the 64-pixel tile.
template <typename T, int BM, int GC>
__global__ void __launch_bounds__(NT1, 2) conv1x1_kernel(K1Args p) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, ESZ = sizeof(T);
  constexpr bool FAST = (E::DTYPE == 1);
  constexpr int BN = 128, WN = 2, WTM = BM / 2, WTN = 64, MI = WTM / 16, NI = 4;
  constexpr int WTILE = BN * 64, WIT = GC * WTILE / (NT1 * 16), PIT = BM / 64, APL = BM * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* alds = smem;                          // [nchunks][BM][64 B], 16-B slots XOR-swizzled by (row >> 1) & 3
  char* wlds = smem + p.nchunks * APL;        // two buffers of GC weight tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int ng = blockIdx.x / tpi, rem = blockIdx.x - ng * tpi;
  const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
  const int n0 = ng * p.G, y0 = tyi << p.lth, x0 = txi << p.lvw;
  const int VWm = (1 << p.lvw) - 1, THm = (1 << p.lth) - 1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.obytes, 0x00020000);

  // ---- stage the activation tile, all chunks ----
  const int fq = tid & 3, frow = tid >> 2;
  int spix[PIT], nimg[PIT], adst[PIT];
#pragma unroll
  for (int u = 0; u < PIT; ++u) {
    const int m = frow + 64 * u;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, g = m >> (p.lvw + p.lth);
    const int n = n0 + g, y = y0 + ty, x = x0 + tx;
    spix[u] = (n < p.N && y < p.H && x < p.W) ? (n * p.H + y) * p.W + x : -1;
    nimg[u] = n < p.N ? n : 0;
    adst[u] = m * 64 + 16 * (fq ^ ((m >> 1) & 3));
  }
  const int pro = p.pro_a == nullptr ? 0 : (p.pro_silu ? 2 : 1);
  {
    u32x4 cur[PIT], nxt[PIT];
    auto load_chunk = [&](int c, u32x4 (&dst)[PIT]) {
      const int cb = c * CHUNK;
      const bool from0 = cb < p.C0;
      const uint32_t Cs = (from0 ? p.C0 : p.C1) * ESZ, so = (from0 ? cb : cb - p.C0) * ESZ, lim = from0 ? p.bytes0 : p.bytes1;
#pragma unroll
      for (int u = 0; u < PIT; ++u) {
        const uint32_t vo = spix[u] >= 0 ? (uint32_t)spix[u] * Cs + fq * 16 : lim;
        dst[u] = from0 ? bload16(rs0, vo, so) : bload16(rs1, vo, so);
      }
    };
    load_chunk(0, cur);
    for (int c = 0; c < p.nchunks; ++c) {
      if (c + 1 < p.nchunks) load_chunk(c + 1, nxt);
      char* plane = alds + c * APL;
#pragma unroll
      for (int u = 0; u < PIT; ++u) {
        u32x4 outv = cur[u];
        if (pro) {
          const float* ap = p.pro_a + (size_t)nimg[u] * p.Cin + c * CHUNK + fq * V;
          const float* bp = p.pro_b + (size_t)nimg[u] * p.Cin + c * CHUNK + fq * V;
          float f[V];
          frag_to_float(cur[u], f, T());
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float v = ap[j] * f[j] + bp[j];
            f[j] = pro == 2 ? silu1<FAST>(v) : v;
          }
          outv = float_to_frag(f, T());
          if (spix[u] < 0) outv = u32x4{0u, 0u, 0u, 0u};
        }
        *reinterpret_cast<u32x4*>(plane + adst[u]) = outv;
      }
#pragma unroll
      for (int u = 0; u < PIT; ++u) cur[u] = nxt[u];
    }
  }

  int arow[MI], brow[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + lr;
    arow[mi] = m * 64 + 16 * (lq ^ ((m >> 1) & 3));
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * WTN + ni * 16 + lr;
    brow[ni] = row * 64 + 16 * (lq ^ ((row >> 1) & 3));
  }
  uint32_t woff[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) woff[i] = (i * NT1 + tid) * 16;   // three consecutive 8 KB tiles
  u32x4 wreg[WIT];
  const int ngroups = (p.nchunks + GC - 1) / GC;
  const int nt0 = blockIdx.y * p.ntn;
  const int ntn = min(p.ntn, p.Cout / BN - nt0);
  const int total = ntn * ngroups;
  auto prefetch_w = [&](int gidx) {
    const int nti = gidx / ngroups, g = gidx - nti * ngroups;
    const uint32_t so = ((uint32_t)(nt0 + nti) * p.nchunks + GC * g) * WTILE;
#pragma unroll
    for (int i = 0; i < WIT; ++i) wreg[i] = bload16(rsw, woff[i], so);   // past-the-end tiles of a short last group read as 0
  };
  prefetch_w(0);

  // output pixel bookkeeping for the epilogue
  uint32_t opix[MI]; int on[MI]; bool ovalid[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + lr;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, g = m >> (p.lvw + p.lth);
    const int n = n0 + g, y = y0 + ty, x = x0 + tx;
    ovalid[mi] = n < p.N && y < p.H && x < p.W;
    on[mi] = n;
    opix[mi] = (uint32_t)((n * p.H + y) * p.W + x);
  }

  int gi = 0;
  for (int nti = 0; nti < ntn; ++nti) {
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // epilogue operands of this output-channel tile are fetched NOW (bias) / before the last group (residual), so their
    // global-memory latency hides behind the K loop instead of stalling every tile's epilogue
    constexpr bool PAIR = E::DTYPE == 1;
    constexpr int NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;
    const int co_t = (nt0 + nti) * BN + wn * WTN;
    const int co_w = co_t + 4 * lq;
    const int co_s = PAIR ? co_t + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
    f32x4 bias4[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bias4[ni] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co_w + ni * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 rr[MI][NP2];
    for (int g = 0; g < ngroups; ++g, ++gi) {
      if (g == ngroups - 1 && p.res) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int k = 0; k < NP2; ++k)
            rr[mi][k] = bload16(rsr, ovalid[mi] ? (opix[mi] * (uint32_t)p.Cout + co_s + k * PSTEP) * ESZ : p.obytes, 0);
      }
      char* wb = wlds + (gi & 1) * (GC * WTILE);
#pragma unroll
      for (int i = 0; i < WIT; ++i) *reinterpret_cast<u32x4*>(wb + (i * NT1 + tid) * 16) = wreg[i];
      __syncthreads();   // weights of this group (and, the first time, the activation tile) visible; orders buffer reuse
      if (gi + 1 < total) prefetch_w(gi + 1);
      const int nc = min(GC, p.nchunks - GC * g);
      const char* ab = alds + GC * g * APL;
#pragma unroll
      for (int j = 0; j < GC; ++j) {
        if (j < nc) {
          u32x4 a[MI], b[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(ab + j * APL + arow[mi]);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(wb + j * WTILE + brow[ni]);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], b[ni], a[mi], T());
        }
      }
    }
    // ---- epilogue of this output-channel tile: 16-byte stores (bf16: tile pairs via permlane16_swap) ----
    {
      GnPartial<NI> gp;
      const bool do_gn = p.gn_stats != nullptr;
      const bool gn_mask = ((p.W & VWm) | (p.H & THm)) != 0;   // partial tiles exist: out-of-image pixels must not count
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f32x4 ad[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          ad[ni] = bias4[ni];
          if (p.emb) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(p.emb + (size_t)min(on[mi], p.N - 1) * p.emb_stride + co_w + ni * 16);
            ad[ni] = f32x4{ad[ni][0] + ev[0], ad[ni][1] + ev[1], ad[ni][2] + ev[2], ad[ni][3] + ev[3]};
          }
        }
        const uint32_t obase = ovalid[mi] ? (opix[mi] * (uint32_t)p.Cout + co_s) * ESZ : p.obytes;
        if constexpr (!PAIR) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4 o = f32x4{acc[mi][ni][0] + ad[ni][0], acc[mi][ni][1] + ad[ni][1], acc[mi][ni][2] + ad[ni][2], acc[mi][ni][3] + ad[ni][3]};
            if (p.res) { const f32x4 t = __builtin_bit_cast(f32x4, rr[mi][ni]); o = f32x4{o[0] + t[0], o[1] + t[1], o[2] + t[2], o[3] + t[3]}; }
            if (do_gn) gp.add(ni, o[0], o[1], o[2], o[3], gn_mask, ovalid[mi] ? 1.f : 0.f);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, obase + ni * 16 * ESZ, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP2; ++k) {
            float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.res) {
              const auto s0 = __builtin_amdgcn_permlane16_swap(rr[mi][k][0], rr[mi][k][2], false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(rr[mi][k][1], rr[mi][k][3], false, false);
              const uint32_t xa[2] = {s0[0], s1[0]}, xb[2] = {s0[1], s1[1]};
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                unpack2(xa[j], ra[2 * j], ra[2 * j + 1], T());
                unpack2(xb[j], rb[2 * j], rb[2 * j + 1], T());
              }
            }
            float va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              va[j] = acc[mi][2 * k][j] + ad[2 * k][j] + ra[j];
              vb[j] = acc[mi][2 * k + 1][j] + ad[2 * k + 1][j] + rb[j];
            }
            if (do_gn) {
              const float vm = ovalid[mi] ? 1.f : 0.f;
              gp.add(2 * k, va[0], va[1], va[2], va[3], gn_mask, vm);
              gp.add(2 * k + 1, vb[0], vb[1], vb[2], vb[3], gn_mask, vm);
            }
            const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, obase + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
      if (do_gn)   // slot = (pixel tile of the image, pixel half wm); G == 1 (launcher), so n0 is the image
        gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * 2 + wm) * (size_t)(p.Cout >> 2) + (co_t >> 2)) * 2, lq, lr);
    }
  }
}

template <typename T, int BM, int GC>
int launch1(const K1Args& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv1x1_kernel<T, BM, GC>;
  if (int rc = mi355_allow_big_lds(kern, "conv1x1")) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(NT1), lds, s, a);
  return 0;
}

}  // namespace

// Returns 1 if this conv is not eligible (caller falls back to the generic implicit-GEMM kernel), 0 on launch, <0 on error.
int conv1x1_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used) {
  if (d.ks != 1 || d.mode != CONV_UNIT || d.out_mode != OUT_NHWC || d.wsplit) return 1;   // (hi / lo split weights: the ping-pong / generic kernels)
  {   // prologue-free, 256-channel output tiles, whole 256-pixel tiles per image: the streaming ping-pong kernel (conv_pp1.inc.h)
    const int r = conv1x1_pp_try_launch(d, stream, gn_slots_used);
    if (r <= 0) return r;
  }
  if (d.Cout % 128 != 0 || (d.res && d.res_mode != RES_SAME)) return 1;
  const int CH = d.dtype == 0 ? 16 : 32, esz = d.dtype == 0 ? 4 : 2;
  const int Cin = d.C0 + d.C1, nchunks = Cin / CH;
  if (d.C0 % CH || d.C1 % CH) return 1;
  const long M = (long)d.N * d.Hs * d.Ws;
  // tile choice: 128 pixels while two workgroups fit a CU (80 KB each) - with the one-chunk weight groups if the three-chunk
  // double buffer would not fit - else 64 pixels
  int BM = 64, GC = 3;
  if ((M + 127) / 128 >= 512) {
    if ((size_t)nchunks * 128 * 64 + 2 * 3 * 8192 <= 80 * 1024) { BM = 128; GC = 3; }
    else if ((size_t)nchunks * 128 * 64 + 2 * 1 * 8192 <= 80 * 1024) { BM = 128; GC = 1; }
  }
  // few pixels (8x8 / 4x4 levels) and many input channels (skip convs over a concat): with the one-chunk weight groups the stationary
  // tile still fits (512 -> 256 at 8x8: 17.8 -> 14.2 us); with many pixels the plain kernel's 128 x 128 tiles stay faster (measured)
  if ((M + 127) / 128 < 512 && (size_t)nchunks * 64 * 64 + 2 * 3 * 8192 > 80 * 1024 && (size_t)nchunks * 64 * 64 + 2 * 1 * 8192 <= 80 * 1024) GC = 1;
  const size_t wl = 2 * (size_t)GC * 8192;
  if ((size_t)nchunks * BM * 64 + wl > 160 * 1024) return 1;
  // A stationary tile only pays when it is reused by several output-channel tiles or when two workgroups still fit a CU
  if ((size_t)nchunks * BM * 64 + wl > 80 * 1024 && d.Cout / 128 < 3) return 1;
  K1Args a;
  a.src0 = d.src0; a.src1 = d.src1; a.C0 = d.C0; a.C1 = d.C1; a.Cin = Cin; a.nchunks = nchunks;
  a.N = d.N; a.H = d.Hs; a.W = d.Ws;
  a.pro_a = d.pro_a; a.pro_b = d.pro_b; a.pro_silu = d.pro_silu;
  a.w = d.w; a.bias = d.bias; a.Cout = d.Cout;
  a.emb = d.emb; a.emb_stride = d.emb_stride;
  a.res = d.res_mode == RES_NONE ? nullptr : d.res;
  a.out = d.out;
  a.gn_stats = nullptr; a.gn_slots = 0;
  const int lw = ilog2_ceil(d.Ws);
  a.lvw = (1 << lw) > BM ? ilog2_ceil(BM) : lw;
  const int VW = 1 << a.lvw, thfull = BM / VW;
  if (d.Hs >= thfull) { a.lth = ilog2_ceil(thfull); a.G = 1; }
  else { a.lth = ilog2_ceil(d.Hs); a.G = thfull >> a.lth; }
  a.tiles_x = (d.Ws + VW - 1) / VW;
  a.tiles_y = (d.Hs + (1 << a.lth) - 1) >> a.lth;
  const int groups = (d.N + a.G - 1) / a.G;
  const int mt = groups * a.tiles_x * a.tiles_y;
  if (d.gn_stats && a.G == 1 && 2 * a.tiles_x * a.tiles_y <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = 2 * a.tiles_x * a.tiles_y; }
  const int ntiles = d.Cout / 128;
  // all output-channel tiles in one workgroup unless that leaves CUs idle
  int ntn = ntiles;
  while (ntn > 1 && (long)mt * ((ntiles + ntn - 1) / ntn) < 512) ntn = (ntn + 1) / 2;
  a.ntn = ntn;
  const size_t b0 = (size_t)d.N * d.Hs * d.Ws * d.C0 * esz, b1 = (size_t)d.N * d.Hs * d.Ws * d.C1 * esz;
  const size_t wb = conv_packed_weight_bytes(d.dtype, d.Cout, Cin, 1);
  MI355_REQUIRE(b0 < 0xFFFF0000ull && b1 < 0xFFFF0000ull && wb < 0xFFFF0000ull, -4,
                "conv1x1: a source tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.bytes0 = (uint32_t)b0; a.bytes1 = d.src1 ? (uint32_t)b1 : 0u; a.wbytes = (uint32_t)wb;
  const size_t ob = (size_t)d.N * d.Hs * d.Ws * d.Cout * esz;
  MI355_REQUIRE(ob < 0xFFFF0000ull, -4, "conv1x1: output tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.obytes = (uint32_t)ob;
  dim3 grid(mt, (ntiles + ntn - 1) / ntn);
  const size_t lds = (size_t)nchunks * BM * 64 + wl;
  int rc;
  rc = dispatch_dtype(d.dtype, [&](auto t) {
    using T = decltype(t);
    if (BM == 128 && GC == 3) return launch1<T, 128, 3>(a, grid, lds, stream);
    if (BM == 128) return launch1<T, 128, 1>(a, grid, lds, stream);
    if (GC == 1) return launch1<T, 64, 1>(a, grid, lds, stream);
    return launch1<T, 64, 3>(a, grid, lds, stream);
  });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  if (gn_slots_used) *gn_slots_used = a.gn_slots;
  return 0;
}
