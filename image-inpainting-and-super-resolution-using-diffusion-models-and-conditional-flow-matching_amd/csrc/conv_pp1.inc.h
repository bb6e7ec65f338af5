// 1x1 convolution in the ping-pong structure of conv_pp.inc.h (included by conv_igemm.hip behind it): ResBlock skip_connection over a channel
// concat, AttentionBlock proj_out (AD/image_diffusion/unet.py:318,389,401).
//
// These launches are HBM-bound GEMMs (M = 65,536 pixels, N = 256, K = 128 .. 512: 140-340 FLOP per byte of activation traffic); round 3 left
// them at 2.0-3.8 TB/s of algorithmic bytes: the wide-input ones (K = 384 / 512) on the generic kernel, whose two 128-channel tiles each fetch
// the input, the others on a stationary-tile kernel whose phases (stage the tile, multiply, fetch the residual, store) run one after the
// other in both workgroups of a CU at once.  Here:
//   * tile = 256 consecutive pixels x 256 output channels: the input is fetched once; eight MFMA waves in two groups, half a step apart (L
//     segment: fragment reads + DMA issue; M segment: 32 MFMAs), exactly as conv3x3_pp_kernel;
//   * a step is one 64-byte channel chunk; BOTH operands go global -> LDS by DMA through four-deep rings (16 KB per chunk each), three
//     chunks ahead: 48 KB of activations in flight per CU (12 MB on the chip = HBM latency x bandwidth), counted waits, nothing drained;
//   * persistent walk with a continuous stream: the loads of the next tile run during the epilogue of this one, so the residual fetch and the
//     stores of tile i overlap the input stream of tile i + 1 (with one tile per CU - B = 256 at 16x16 - they overlap its own tail only).
// Ordering rules and the tick diagram are conv_pp.inc.h's (T = chunks per tile).
namespace pp1 {
constexpr int NRING = 4, AHEAD = 3;
// WIDE = 1: tile 256 pixels x 256 channels, waves 2 (pixels) x 4 (channels); WIDE = 0: 512 pixels x 128 channels, waves 4 x 2 (the 128-channel
// outputs of the 32x32 level).  A wave tile is 128 pixels x 64 channels either way.
// WIDE = 2 (round 5): 128 pixels x 256 channels, waves 2 x 4 with a wave tile of 64 pixels x 64 channels - for launches of fewer than two wide tiles
// per CU (B = 256 at 16x16: one), where the stream had nothing to overlap a tile's residual fetch and stores with: twice the tiles, so the second
// tile's input streams under the first one's epilogue (the weights are fetched from L2 once more per tile).
template <int WIDE> struct Cfg {
  static constexpr int BM = WIDE == 2 ? 128 : (WIDE ? 256 : 512), BN = WIDE ? 256 : 128;
  static constexpr int MIW = WIDE == 2 ? 4 : 8, WPX = 16 * MIW;         // a wave's pixel tiles / pixels
  static constexpr int TAPA = BM * 64, TAPB = BN * 64;                  // one chunk of the activations / of the weights
  static constexpr int PA = BM / 16 / 8, PB = BN / 16 / 8;              // 1-KB DMA pieces per wave and step: 2 + 2 / 4 + 1
  static constexpr int OPS = PA + PB;
  static constexpr int OFF_A = 0, OFF_B = NRING * TAPA;
  static constexpr size_t LDS_BYTES = NRING * ((size_t)TAPA + TAPB);    // 131,072 / 163,840 B
};
}  // namespace pp1

template <typename T, int WIDE>
__global__ void __launch_bounds__(512, 2) conv1x1_pp_kernel(ConvKArgs p, int n_mt, int n_nt) {
  using namespace pp1;
  using G = Cfg<WIDE>;
  constexpr int MIW = G::MIW, WPX = G::WPX;
  constexpr int BM = G::BM, BN = G::BN, TAPA = G::TAPA, TAPB = G::TAPB, PA = G::PA, PB = G::PB, OPS = G::OPS, OFF_A = G::OFF_A, OFF_B = G::OFF_B;
  using E = Elem<T>;
  constexpr int CHUNK = E::CHUNK, ESZ = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2;                 // 0: waves 0-3, 1: waves 4-7 (the second wave of each SIMD), one tick behind
  const int wm = WIDE ? grp : wave8 >> 1, wn = WIDE ? wave8 & 3 : wave8 & 1;   // pixels WPX wm .. WPX wm + WPX - 1 of the tile, channels 64 wn .. 64 wn + 63
  const int lr = lane & 15, lq = lane >> 4;
  const int TT = p.nchunks;                   // steps per tile (a multiple of 4: the launcher)
  const int HW = p.Ho * p.Wo;                 // a multiple of 256 (the launcher): a tile never straddles two images
  // Tile walk: 8 consecutive workgroups (one per XCD) take 8 consecutive pixel tiles, the workgroup 8 further on (same XCD, same L2) the
  // next 256-channel tile of the same pixels.
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt;
  auto decode = [&](int t, int& mt, int& nt) {
    const int per = 8 * n_nt, blk = t / per, r = t - blk * per;
    nt = r >> 3; mt = blk * 8 + (r & 7);
  };
  auto next_valid = [&](int t) {
    for (t += gridDim.x; t < ntp; t += gridDim.x) { int mt, nt; decode(t, mt, nt); if (mt < n_mt) break; }
    return t;
  };
  int t_first = (int)blockIdx.x - (int)gridDim.x;
  t_first = next_valid(t_first);
  if (t_first >= ntp) return;                 // the whole workgroup leaves together

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.ablate & 1) ? 0u : p.obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);

  // ---------------- DMA streams (three steps ahead of the multiplication, continuous across tiles) ----------------
  // Weights: chunk c of a 256-channel tile = tile (2 nt + g, c) of the packed image ([nt128][chunk][128 rows][64 B], rows already XOR-swizzled:
  // the LDS image) for g = 0, 1: 16 pieces of 1 KB; wave w moves pieces 2 w, 2 w + 1 (group g: half g); piece j lands at ring slot + 1024 j.
  // (WIDE = 0: the 128-channel tile is one packed tile of 8 pieces, wave w moves piece w)
  auto ws_base = [&](int t) { int mt, nt; decode(t, mt, nt); return (uint32_t)((WIDE ? 2 * nt + grp : nt) * TT) * 8192u; };
  uint32_t ws_soff = ws_base(t_first);
  const uint32_t wvo0 = (uint32_t)((WIDE ? (2 * wave8) & 7 : wave8) * 1024 + lane * 16);
  // Activations: piece j = pixels 16 j .. 16 j + 15 of the tile x 64 B; lane l lands at slot l & 3 of pixel 16 j + (l >> 2), so the slot swizzle
  // (by the pixel index) goes into the per-lane SOURCE address.  Wave w moves pieces 2 w, 2 w + 1.
  // (sized by a literal, PA <= 4 used: as `uint32_t pvo0[PA]` - a size that depends on the template parameter - hipcc 7.2's HOST pass silently
  //  drops the kernel's stub once the DMA builtin reads the array inside a lambda, and the library fails to load with an undefined symbol;
  //  tests/test_abi.py loads the library on the CPU)
  uint32_t pvo0[4], pvo1[4];             // per-lane source offsets of this wave's pieces, for the tile whose chunks are being streamed
  auto ps_setup = [&](int t) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));   // (opaque: see conv_pp.inc.h ps_setup)
    asm volatile("" : "+s"(t));
    int mt, nt;
    decode(t, mt, nt);
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int px = 16 * (PA * wave8 + q) + (ln >> 2);
      const uint32_t pix = (uint32_t)(mt * BM + px);
      const uint32_t fq = (uint32_t)(((ln & 3) ^ ((px >> 1) & 3)) * 16);
      pvo0[q] = pix * (uint32_t)(p.C0 * ESZ) + fq;
      pvo1[q] = pix * (uint32_t)(p.C1 * ESZ) + fq;
    }
  };
  ps_setup(t_first);
  int s_chunk = 0;                       // chunk of the streamed tile the next L segment fetches
  auto issue = [&](auto ringc) {         // the stream's next step: weights, then activations
    constexpr int ring = decltype(ringc)::value;
    char* dw = smem + OFF_B + ring * TAPB + (PB * wave8) * 1024;
#pragma unroll
    for (int q = 0; q < PB; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(dw + q * 1024), 16, wvo0 + q * 1024u, ws_soff, 0, 0);
    ws_soff += 8192u;
    const int cb = src_chunk(p, s_chunk) * CHUNK;
    const bool first = cb < p.C0;
    const uint32_t so = (uint32_t)((first ? cb : cb - p.C0) * ESZ);
    char* da = smem + OFF_A + ring * TAPA + (PA * wave8) * 1024;
#pragma unroll
    for (int q = 0; q < PA; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)(da + q * 1024), 16, first ? pvo0[q] : pvo1[q], so, 0, 0);
    ++s_chunk;
  };

  // ---------------- fragment addresses (bases made opaque per step: see conv_pp.inc.h) ----------------
  int a_base = OFF_A + (wm * WPX + lr) * 64 + 16 * (lq ^ ((lr >> 1) & 3));   // + ring * TAPA (added per step: beyond the 16-bit immediate when WIDE = 0) + mi * 1024
  int b_base = OFF_B + (wn * 64 + lr) * 64 + 16 * (lq ^ ((lr >> 1) & 3));    // + ring * TAPB + ni * 1024

  constexpr bool PAIR = E::DTYPE == 1;
  constexpr int NI = 4, NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;
  f32x4 acc[MIW][NI];
  f32x4 cin[NI];                           // bias (+ per-image embedding) of this lane's channels: the accumulators start from it
  auto cinit_load = [&](int t) {
    int mt, nt;
    decode(t, mt, nt);
    const int n0 = (mt * BM) / HW;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int co = nt * BN + wn * 64 + ni * 16 + 4 * lq;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co);
      if (p.emb) { const f32x4 e = *reinterpret_cast<const f32x4*>(p.emb + (size_t)n0 * p.emb_stride + co); v = f32x4{v[0] + e[0], v[1] + e[1], v[2] + e[2], v[3] + e[3]}; }
      cin[ni] = v;
    }
  };
  auto acc_init = [&]() {
#pragma unroll
    for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = cin[ni];
  };

  // ---------------- pipeline fill: steps 0 .. 2 of the first tile -> ring slots 0 .. 2 ----------------
  cinit_load(t_first);
  issue(IC<0>()); issue(IC<1>()); issue(IC<2>());
  acc_init();
  pp_wait_vm<2 * OPS>();                   // this wave's pieces of step 0 have landed (steps 1, 2 may fly)
  pp_barrier();

  // E's vector-memory operations sit inside the window of the next tile's first wait (conv_pp.inc.h): stores (+ residual loads) counted
  // exactly, everything else not at all; fp32 would overflow the 6-bit counter and drains instead.
  constexpr int EPI_STORES = MIW * NP2;
  const int extra0 = PAIR ? EPI_STORES * (p.res_mode != RES_NONE ? 2 : 1) : 0;
  int extra = 0;

  for (int t = t_first; t < ntp;) {
    const int t_next = next_valid(t);
    const int t_nextc = t_next < ntp ? t_next : t;   // the streams' next tile (clamped at the end of the walk)
    int mt, nt;
    decode(t, mt, nt);
    if (grp == 1) pp_barrier();            // one tick behind group 0
    // one step: S = step mod 4 = ring slot; lastq: the group of four is the tile's last
    auto step = [&](auto Sc, bool first_q, bool lastq) {
      constexpr int S = decltype(Sc)::value, ringn = (S + AHEAD) % NRING;
      const bool last_of_tile = lastq && S == 3;
      u32x4 af[MIW], bf[NI];
      {
        int ab = a_base + S * TAPA, bb = b_base;
        asm volatile("" : "+v"(ab), "+v"(bb));
        const char* ap = smem + ab;
        const char* bp = smem + bb;
#pragma unroll
        for (int mi = 0; mi < MIW; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(ap + mi * 1024);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(bp + S * TAPB + ni * 1024);
      }
      // the stream reaches the walk's next tile three steps before the multiplication does (past the end of the walk it re-fetches the
      // last tile into slots nobody reads)
      if constexpr (S == 1) { if (lastq) { ps_setup(t_nextc); ws_soff = ws_base(t_nextc); s_chunk = 0; } }
      issue(IC<ringn>());
      if constexpr (S == 0 && PAIR) {      // the first step of a tile: the previous tile's epilogue is inside the window
        if (!first_q || extra == 0) pp_wait_vm<2 * OPS>();
        else if (extra == EPI_STORES) pp_wait_vm<2 * OPS + EPI_STORES>();
        else pp_wait_vm<2 * OPS + 2 * EPI_STORES>();
      } else {
        pp_wait_vm<2 * OPS>();             // everything up to and including the pieces of step s + 1 (issued in L(s - 2)) has landed
      }
      pp_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], bf[ni], af[mi], T());   // D rows = channels, cols = pixels
      __builtin_amdgcn_s_setprio(0);
      if (!(last_of_tile && grp == 1)) pp_barrier();
    };
    for (int c = 0; c < TT; c += 4) {
      const bool lastq = c + 4 >= TT;
      step(IC<0>(), c == 0, lastq); step(IC<1>(), false, lastq); step(IC<2>(), false, lastq); step(IC<3>(), false, lastq);
    }

    // ---------------- epilogue (conv_pp.inc.h's: MFMA rows are channels, columns are pixels) ----------------
    int t_nx = t_nextc;
    asm volatile("" : "+s"(t_nx), "+s"(nt), "+s"(mt));   // (opaque copies: the epilogue's address arithmetic is invariant in the step loop)
    cinit_load(t_nx);
    const int m0 = mt * BM, n0 = m0 / HW;
    const int co_w = nt * BN + wn * 64 + 4 * lq;
    const int co_s = PAIR ? nt * BN + wn * 64 + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
    GnPartial<NI> gp;
    const bool do_gn = p.gn_stats != nullptr;
    auto epi_half = [&](auto hc, auto resc, auto gnc) {
      constexpr int h = decltype(hc)::value, GNM = decltype(gnc)::value;
      constexpr bool HAS_RES = decltype(resc)::value != 0;
      uint32_t ovo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ovo[j] = ((uint32_t)(m0 + wm * WPX + (h * 4 + j) * 16 + lr) * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ;
      u32x4 rr[4][NP2];
      if constexpr (HAS_RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < NP2; ++k)
            rr[j][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, ovo[j] + k * PSTEP * ESZ, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mi = h * 4 + j;
        if constexpr (!PAIR) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4 o = acc[mi][ni];
            if constexpr (HAS_RES) {
              const f32x4 tt = __builtin_bit_cast(f32x4, rr[j][ni]);
              o = f32x4{o[0] + tt[0], o[1] + tt[1], o[2] + tt[2], o[3] + tt[3]};
            }
            if constexpr (GNM != 0) gp.add(ni, o[0], o[1], o[2], o[3], false, 1.f);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, ovo[j] + ni * 16 * ESZ, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP2; ++k) {
            float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HAS_RES) {   // un-swap the 8-channel residual piece back to the accumulator layout
              const auto s0 = __builtin_amdgcn_permlane16_swap(rr[j][k][0], rr[j][k][2], false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(rr[j][k][1], rr[j][k][3], false, false);
              const uint32_t xa[2] = {s0[0], s1[0]}, xb[2] = {s0[1], s1[1]};
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                unpack2(xa[q], ra[2 * q], ra[2 * q + 1], T());
                unpack2(xb[q], rb[2 * q], rb[2 * q + 1], T());
              }
            }
            float va[4], vb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              va[q] = HAS_RES ? acc[mi][2 * k][q] + ra[q] : acc[mi][2 * k][q];
              vb[q] = HAS_RES ? acc[mi][2 * k + 1][q] + rb[q] : acc[mi][2 * k + 1][q];
            }
            if constexpr (GNM != 0) {
              gp.add(2 * k, va[0], va[1], va[2], va[3], false, 1.f);
              gp.add(2 * k + 1, vb[0], vb[1], vb[2], vb[3], false, 1.f);
            }
            const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo[j] + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
    };
    auto epi = [&](auto resc, auto gnc) { epi_half(IC<0>(), resc, gnc); if constexpr (MIW == 8) epi_half(IC<1>(), resc, gnc); };
    if (p.res_mode != RES_NONE) { if (!do_gn) epi(IC<1>(), IC<0>()); else epi(IC<1>(), IC<1>()); }
    else { if (!do_gn) epi(IC<0>(), IC<0>()); else epi(IC<0>(), IC<1>()); }
    if (do_gn) {   // slot = (pixel tile of the image, a wave's pixel part); quads of this wave's 64 channels
      const int rem = (m0 - n0 * HW) / BM;
      gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * (BM / WPX) + wm) * (size_t)(p.Cout >> 2) + ((nt * BN + wn * 64) >> 2)) * 2, lq, lr);
    }
    acc_init();
    if constexpr (!PAIR) pp_wait_vm<0>();
    extra = PAIR ? extra0 : 0;
    pp_barrier();                          // both groups have stored their tile: the next tile starts with group 0's L(0)
    t = t_next;
  }
  pp_wait_vm<0>();                         // no DMA piece may still be in flight towards LDS when the workgroup retires
}

// 0 = launched, 1 = not eligible (the caller goes on to the stationary-tile / generic kernels), < 0 = error.  mode: mi355_debug_config::conv_pp.
template <typename T, int WIDE>
static int pp1_launch(const ConvKArgs& a, int n_mt, int n_nt, hipStream_t stream) {
  if (int rc = mi355_allow_big_lds(conv1x1_pp_kernel<T, WIDE>, "conv1x1 (ping-pong)")) return rc;
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt, ncu = ws_num_cus();
  const int grid = ntp < ncu ? ntp : ncu;   // one persistent workgroup per CU
  hipLaunchKernelGGL((conv1x1_pp_kernel<T, WIDE>), dim3(grid), dim3(512), pp1::Cfg<WIDE>::LDS_BYTES, stream, a, n_mt, n_nt);
  return 0;
}

// 0 = launched, 1 = not eligible (the caller goes on to the stationary-tile / generic kernels), < 0 = error.  mode: mi355_debug_config::conv_pp.
static int pp1_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used) {
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  const int CH = d.dtype == 0 ? 16 : 32, esz = d.dtype == 0 ? 4 : 2;
  if (!K.conv_pp || d.ks != 1 || d.mode != CONV_UNIT || d.out_mode != OUT_NHWC || d.pro_a) return 1;
  if (d.Cout % 128 != 0 || d.C0 % CH != 0 || d.C1 % CH != 0 || conv_tile_n(d.Cout) != 128) return 1;
  const int wide = d.Cout % 256 == 0;
  const int HW0 = d.Hs * d.Ws;
  // half-height wide tiles (conv_pp bit 5) where the wide walk would give a CU fewer than two tiles
  const bool half = wide && (K.conv_pp & 32) && HW0 % 256 == 0 && (long)d.N * HW0 / 256 * (d.Cout / 256) < 2L * ws_num_cus();
  const int BM = half ? 128 : (wide ? 256 : 512), BN = wide ? 256 : 128;
  const int Cin = d.C0 + d.C1, nreal = Cin / CH, nchunks = nreal * (d.wsplit ? 2 : 1), HW = d.Hs * d.Ws;
  // (four chunks: the old stationary-tile kernel ties - 128 -> 256 at 16x16: 13.4 vs 13.9 us - and stays)
  if (nchunks < 8 || (nchunks & 3) || (HW % BM) != 0) return 1;
  if (d.res && d.res_mode != RES_SAME) return 1;
  const int n_mt = (int)((long)d.N * HW / BM), n_nt = d.Cout / BN;
  if ((K.conv_pp & 3) < 2 && n_mt * n_nt < ws_num_cus()) return 1;
  ConvKArgs a{};
  a.src0 = d.src0; a.src1 = d.src1; a.C0 = d.C0; a.C1 = d.C1; a.Cin = Cin; a.nchunks = nchunks; a.nreal = nreal;
  a.N = d.N; a.Hs = d.Hs; a.Ws = d.Ws; a.Hc = d.Hs; a.Wc = d.Ws; a.Ho = d.Hs; a.Wo = d.Ws;
  a.w = d.w; a.bias = d.bias; a.Cout = d.Cout; a.bn_pack = 128;
  a.emb = d.emb; a.emb_stride = d.emb_stride;
  a.res = d.res; a.res_mode = d.res ? RES_SAME : RES_NONE; a.Hr = d.Hs; a.Wr = d.Ws;
  a.out = d.out; a.out_mode = OUT_NHWC;
  const size_t b0 = (size_t)d.N * HW * d.C0 * esz, b1 = (size_t)d.N * HW * d.C1 * esz, ob = (size_t)d.N * HW * d.Cout * esz;
  const size_t wb = conv_packed_weight_bytes(d.dtype, d.Cout, Cin, 1, d.wsplit);
  MI355_REQUIRE(b0 < 0xFFFF0000ull && b1 < 0xFFFF0000ull && wb < 0xFFFF0000ull && ob < 0xFFFF0000ull, -4,
                "conv1x1: a tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.bytes0 = (uint32_t)b0; a.bytes1 = d.src1 ? (uint32_t)b1 : 0u; a.wbytes = (uint32_t)wb; a.obytes = (uint32_t)ob; a.rbytes = d.res ? (uint32_t)ob : 0u;
  a.ablate = K.conv_ablate; a.err = d.err; a.spin_limit = 1;
  const int slots = HW / (half ? 64 : 128);   // (pixel tile of the image, a wave's pixel part)
  if (d.gn_stats && slots <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = slots; }
  int rc;
  rc = dispatch_dtype(d.dtype, [&](auto t) { using T = decltype(t); return half ? pp1_launch<T, 2>(a, n_mt, n_nt, stream) : (wide ? pp1_launch<T, 1>(a, n_mt, n_nt, stream) : pp1_launch<T, 0>(a, n_mt, n_nt, stream)); });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  if (gn_slots_used) *gn_slots_used = a.gn_slots;
  return 0;
}
