// 3x3 convolution of the 8x8 and 4x4 levels (included by conv_igemm.hip; same args, same packed weights, same epilogue).
//
// At these levels a batch holds few pixels (B = 256: 16k / 4k), so the GEMM is short and wide: every 64-pixel tile needs ALL of
// the layer's weights (1.2-2.4 MB) and there are only as many tiles as CUs.  The plain kernel's per-row pipeline (stage the
// patch, stage a weight row, barrier, 24 MFMAs) is then latency-bound in every phase (stamps: DESIGN.md section 6), although the
// bound itself - the CU's 64 B/clk L1 path streaming the weights, equal to the MFMA time of a 64-pixel tile - is 3x lower.  Here:
//   * one 4-wave workgroup per 64-pixel tile (one 8x8 image or four 4x4 images, whole images: no halo exchange, zero padding
//     only); its haloed input patch for ALL channel chunks is staged into LDS once (one barrier), then the K loop has NO barrier;
//   * the weights never touch LDS: a wave owns 64 output channels, nobody else in the workgroup needs its weight rows, so its B
//     fragments go global -> VGPR directly (the host-packed tile is already in fragment order: one 1-KB coalesced load per 16
//     channels) through a register ring one chunk (9 taps) deep - the loads of chunk c + 1 are issued as chunk c's taps retire;
//   * few tiles (4x4 level): the four waves split K instead of N (KSPLIT = 2 / 4: more workgroups of fewer channels), partial
//     accumulators meet in LDS and every wave finishes (epilogue) a quarter / half of the pixels;
//   * LDS image of the patch: dense 64-B pixel rows, 16-B slots XOR-swizzled by the pixel index, row pitch 16 (8x8) / 8 (4x4)
//     pixels: conflict-free ds_read_b128 for every tap shift (tools/lds_bank_sim.py; the plain kernel's 96-B rows are 2-way
//     conflicted on 8-pixel-wide images).
namespace sm {
// sphase x schp: centre-tap chunk phases of a fused 1x1 skip conv (0: none); sup: its planes are staged with the main patch instead (one phase, dense
// 64-pixel x 64-B planes behind the main ones: LDS permitting)
struct Args { int pwp, pimg, sh, plane, nphase, chp, red_bytes, sphase, schp, sup; };
constexpr int SPLANE = 4096, SDMAX = 16;   // dense skip plane bytes; at most this many of them
}  // namespace sm

// NI = 4: four waves of 64 output channels (one per SIMD).  NI = 2 (round 5, whole-chip launches of the 8x8 level): EIGHT waves of 32 channels, two per SIMD - the
// same bytes through the CU's L1 path, but while one wave of a SIMD waits for its weight fragments the other multiplies (in-kernel stamps of the four-wave
// form: weight stream alone 9.1 us, MFMAs alone 9.8 us, together 20: they added up instead of overlapping).
template <typename T, int KSPLIT, int NCHP, int PITU, int NI = 4, int W8_ = (NCHP == 16), int SKIP = 0>   // W8_: 8x8 output images (one per tile); else four 4x4 images; SKIP: carries a fused 1x1 skip conv
__global__ void __launch_bounds__(NI == 4 ? 256 : 512, NI == 4 ? 1 : 2) conv3x3_small_kernel(ConvKArgs p, sm::Args g) {
  using E = Elem<T>;
  constexpr int CHUNK = E::CHUNK, ESZ = sizeof(T);
  constexpr int NW = (NI == 4 ? 4 : 8) / KSPLIT;   // waves along the output channels (16 NI each); KSPLIT waves share a channel block
  constexpr int NT = NI == 4 ? 256 : 512;          // threads
  constexpr int MI = 4, MPW = MI / KSPLIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = wave / NW, wn = wave - kk * NW;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = blockIdx.x * p.G;
  const int co0 = ((int)blockIdx.y * NW + wn) * (16 * NI);
  const int VWm = (1 << p.lvw) - 1, THm = (1 << p.lth) - 1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
  // fused 1x1 skip conv (ResBlock skip_connection, unet.py:312-317,351: out = skip(x) + conv2(h)): x = cat(sk0, sk1), contracted at the centre tap
  const __amdgpu_buffer_rsrc_t rk0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sk0 ? p.sk0 : p.src0), 0, p.sk0 ? p.skbytes0 : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.sk1 ? p.sk1 : p.src0), 0, p.sk1 ? p.skbytes1 : 0u, 0x00020000);

  // ---- weight stream of this wave: chunk sequence s = 0 .. nphase * chw - 1 ----
  const int chw = g.chp / KSPLIT;     // chunks per wave and phase: chunk (ph, j) of this wave = ph * chp + kk * chw + j
  uint32_t boff[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (co0 & 127) + ni * 16 + lr;
    boff[ni] = row * 64 + 16 * (lq ^ ((row >> 1) & 3));
  }
  // 8-KB tiles of this wave's 128-channel pack tile: (chunk c, tap) at 9 c + tap, then - fused skip conv - one tile per skip chunk
  const uint32_t wt0 = (uint32_t)(co0 >> 7) * p.wstride;
  u32x4 bring[9][NI] = {};
  auto issue_b = [&](auto tapc, int tile9) {   // ring slot tap <- tile tile9 + tap
    constexpr int tap = decltype(tapc)::value;
    const uint32_t so = (wt0 + (uint32_t)tile9 + tap) * 8192u;
    if constexpr (WS_ABLATE & 8) return;   // diagnostic builds only (Makefile: variant ABL=<mask>): 4 = no LDS reads / MFMAs, 8 = no weight loads, 16 = no patch staging
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bring[tap][ni] = buf_load16(rsw, boff[ni], so);
  };
  {
    const int c = kk * chw * 9;
    issue_b(IC<0>(), c); issue_b(IC<1>(), c); issue_b(IC<2>(), c); issue_b(IC<3>(), c); issue_b(IC<4>(), c);
    issue_b(IC<5>(), c); issue_b(IC<6>(), c); issue_b(IC<7>(), c); issue_b(IC<8>(), c);
  }
  const int schw = g.schp / KSPLIT;   // skip chunks per wave and skip phase: chunk (sp, j) of this wave = sp * schp + kk * schw + j, tile 9 nchunks + that

  // ---- patch staging geometry: fragment u of this thread = patch pixel (tid + 256 u) >> 2, 16-B quarter (tid + 256 u) & 3 ----
  uint32_t voff0[PITU], voff1[PITU]; int ldst[PITU];
  auto patch_geometry = [&](int CA, int CB, uint32_t bytesA, uint32_t bytesB) {   // per-pixel byte offsets into two sources of CA / CB channels
    const int ppi = p.PH * p.PW, npv = p.G * ppi;
    const float inv_ppi = 1.0f / (float)ppi, inv_pw = 1.0f / (float)p.PW;
#pragma unroll
    for (int u = 0; u < PITU; ++u) {
      const int i = tid + NT * u, pp = i >> 2, q = i & 3;
      int s = -1, dst = -1;
      if (pp < npv) {
        const int gi = (int)(((float)pp + 0.5f) * inv_ppi), r = pp - gi * ppi;
        const int py = (int)(((float)r + 0.5f) * inv_pw), px = r - py * p.PW;
        const int n = n0 + gi, cy = py - 1, cx = px - 1;
        if (n < p.N && cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc)
          s = p.mode == CONV_UP2 ? (n * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1) : (n * p.Hs + cy) * p.Ws + cx;
        const int lin = gi * g.pimg + py * g.pwp + px;
        dst = lin * 64 + 16 * (q ^ ((lin >> g.sh) & 3));
      }
      ldst[u] = dst;
      voff0[u] = s >= 0 ? (uint32_t)s * (uint32_t)(CA * ESZ) + q * 16 : bytesA;
      voff1[u] = s >= 0 ? (uint32_t)s * (uint32_t)(CB * ESZ) + q * 16 : bytesB;
    }
  };
  patch_geometry(p.C0, p.C1, p.bytes0, p.bytes1);
  // planes 0 .. nlive - 1 <- channel chunks c0 .. of cat(A, B) (A has CA channels), NB planes' loads in flight at a time
  auto stage_from = [&](auto nbc, int c0, int nlive, bool wrap, const __amdgpu_buffer_rsrc_t& rA, const __amdgpu_buffer_rsrc_t& rB, int CA, uint32_t bytesA, uint32_t bytesB) {
    constexpr int NB = decltype(nbc)::value;
    if constexpr (WS_ABLATE & 16) return;
#pragma unroll
    for (int b0 = 0; b0 < NCHP; b0 += NB) {
      if (b0 >= nlive) break;
      u32x4 raw[NB][PITU];
#pragma unroll
      for (int cl = 0; cl < NB; ++cl) {
        const int cb = (wrap ? src_chunk(p, c0 + b0 + cl) : c0 + b0 + cl) * CHUNK;
        const bool live = b0 + cl < nlive, first = cb < CA;
#pragma unroll
        for (int u = 0; u < PITU; ++u) {
          const uint32_t vo = live ? (first ? voff0[u] : voff1[u]) : (first ? bytesA : bytesB);
          raw[cl][u] = buf_load16(first ? rA : rB, vo, (uint32_t)(first ? cb : cb - CA) * ESZ);
        }
      }
#pragma unroll
      for (int cl = 0; cl < NB; ++cl) {
        if (b0 + cl < nlive) {
#pragma unroll
          for (int u = 0; u < PITU; ++u)
            if (ldst[u] >= 0) *reinterpret_cast<u32x4*>(smem + (b0 + cl) * g.plane + ldst[u]) = raw[cl][u];
        }
      }
    }
  };
  auto stage = [&](int ph) { stage_from(IC<NCHP>(), ph * g.chp, g.chp, true, rs0, rs1, p.C0, p.bytes0, p.bytes1); };

  // ---- A fragment addresses of the 9 taps (the swizzle follows the shifted pixel, so every tap has its own) ----
  int aaddr[9][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mi * 16 + lr;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, gi = m >> (p.lvw + p.lth);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int lin = gi * g.pimg + (ty * p.stride + t / 3) * g.pwp + tx * p.stride + t % 3;
      aaddr[t][mi] = lin * 64 + 16 * (lq ^ ((lin >> g.sh) & 3));
    }
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fused skip conv, planes staged up front: the 64 output pixels' rows of the raw block input, chunk by chunk, dense (rows of 64 B, 16-B
  // slots XOR-swizzled by the pixel pair: conflict-free ds_read_b128, tools/lds_bank_sim.py); loads issued here, stored behind the main patch's ----
  constexpr int SDPT = sm::SDMAX / (NT / 256);   // dense planes per thread (a plane is 256 16-B fragments)
  u32x4 rawd[SKIP ? SDPT : 1];
  int dstd = 0, saddr[MI] = {};
  if constexpr (SKIP != 0) {
    if (g.sup) {
      const int pp = (tid & 255) >> 2, q = tid & 3, half = tid >> 8;
      const int tx = pp & VWm, ty = (pp >> p.lvw) & THm, gi = pp >> (p.lvw + p.lth);
      const int n = n0 + gi;
      const bool ok = n < p.N && ty < p.Ho && tx < p.Wo;
      const uint32_t spx = (uint32_t)((n * p.Hs + ty) * p.Ws + tx);
      const uint32_t oA = ok ? spx * (uint32_t)(p.SC0 * ESZ) + q * 16 : p.skbytes0, oB = ok ? spx * (uint32_t)(p.SC1 * ESZ) + q * 16 : p.skbytes1;
      dstd = g.chp * g.plane + pp * 64 + 16 * (q ^ ((pp >> 1) & 3));
#pragma unroll
      for (int k = 0; k < SDPT; ++k) {
        const int cl = k * (NT / 256) + half, cb = cl * CHUNK;
        const bool live = cl < g.schp, first = cb < p.SC0;
        rawd[k] = buf_load16(first ? rk0 : rk1, live ? (first ? oA : oB) : (first ? p.skbytes0 : p.skbytes1), (uint32_t)(first ? cb : cb - p.SC0) * ESZ);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = mi * 16 + lr;
        saddr[mi] = m * 64 + 16 * (lq ^ ((m >> 1) & 3));
      }
    }
  }

  for (int ph = 0; ph < g.nphase; ++ph) {
    if (ph > 0) __syncthreads();         // every wave has finished reading the previous phase's patch
    stage(ph);
    if constexpr (SKIP != 0) {
      if (g.sup && ph == 0) {
        const int half = tid >> 8;
#pragma unroll
        for (int k = 0; k < SDPT; ++k) {
          const int cl = k * (NT / 256) + half;
          if (cl < g.schp) *reinterpret_cast<u32x4*>(smem + cl * sm::SPLANE + dstd) = rawd[k];
        }
      }
    }
    __syncthreads();
    for (int j = 0; j < chw; ++j) {
      const int c = ph * g.chp + kk * chw + j;
      // the chunk whose weights replace this one's in the ring; after the last chunk: this wave's first skip chunks, or the same chunk again (never used)
      const int cn = (j + 1 < chw ? c + 1 : (ph + 1 < g.nphase ? c + 1 + g.chp - chw : c)) * 9;
      const int cn9 = (SKIP != 0 && j + 1 == chw && ph + 1 == g.nphase) ? p.nchunks * 9 + kk * schw : cn;
      const char* pl = smem + (kk * chw + j) * g.plane;
      u32x4 af[2][MI] = {};
      auto read_a = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (WS_ABLATE & 4) return;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[t & 1][mi] = *reinterpret_cast<const u32x4*>(pl + aaddr[t][mi]);
      };
      auto tap = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t < 8) read_a(IC<t + 1>());
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WS_ABLATE & 4) { asm volatile("" ::"v"(bring[t][0]), "v"(bring[t][1]), "v"(bring[t][2]), "v"(bring[t][3])); }
        else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], bring[t][ni], af[t & 1][mi], T());   // D rows = channels, cols = pixels
        __builtin_amdgcn_sched_barrier(0);
        issue_b(tc, cn9);
      };
      read_a(IC<0>());
      tap(IC<0>()); tap(IC<1>()); tap(IC<2>()); tap(IC<3>()); tap(IC<4>()); tap(IC<5>()); tap(IC<6>()); tap(IC<7>()); tap(IC<8>());
    }
  }

  // ---- fused skip conv: phases of centre-tap chunks of the raw block input, nine chunks per pass of the weight ring (the ring slot that was a
  // tap is now a chunk: tile 9 nchunks + chunk); a slot beyond the wave's chunks holds a tile nobody multiplies ----
  if constexpr (SKIP != 0) {
    if (!g.sup) {
      __syncthreads();                   // main patch dead
      patch_geometry(p.SC0, p.SC1, p.skbytes0, p.skbytes1);
    }
    const int pstep = g.sup ? sm::SPLANE : g.plane;
    int caddr[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) caddr[mi] = g.sup ? g.chp * g.plane + saddr[mi] : aaddr[4][mi];
    for (int sp = 0; sp < g.sphase; ++sp) {
      if (!g.sup) {
        if (sp > 0) __syncthreads();
        stage_from(IC<(NCHP >= 8 ? NCHP / 2 : NCHP)>(), sp * g.schp, g.schp, false, rk0, rk1, p.SC0, p.skbytes0, p.skbytes1);
        __syncthreads();
      }
      for (int j0 = 0; j0 < schw; j0 += 9) {
        const int nvalid = min(9, schw - j0);
        const int here = sp * g.schp + kk * schw + j0;   // first skip chunk of this pass
        const int next9 = p.nchunks * 9 + (j0 + 9 < schw ? here + 9 : (sp + 1 < g.sphase ? (sp + 1) * g.schp + kk * schw : here));
        const char* pl0 = smem + (kk * schw + j0) * pstep;
        u32x4 af[2][MI] = {};
        auto read_s = [&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (WS_ABLATE & 4) return;
          const char* pl = pl0 + (t < nvalid ? t : 0) * pstep;   // (a slot beyond the wave's chunks reads a plane that exists; nobody multiplies it)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[t & 1][mi] = *reinterpret_cast<const u32x4*>(pl + caddr[mi]);
        };
        auto stap = [&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (t < 8) read_s(IC<t + 1>());
          __builtin_amdgcn_sched_barrier(0);
          if (t < nvalid) {
            if constexpr (WS_ABLATE & 4) { asm volatile("" ::"v"(bring[t][0]), "v"(bring[t][1])); }
            else
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], bring[t][ni], af[t & 1][mi], T());
          }
          __builtin_amdgcn_sched_barrier(0);
          issue_b(tc, next9);
        };
        read_s(IC<0>());
        stap(IC<0>()); stap(IC<1>()); stap(IC<2>()); stap(IC<3>()); stap(IC<4>()); stap(IC<5>()); stap(IC<6>()); stap(IC<7>()); stap(IC<8>());
      }
    }
  }

  // ---- split K: partial accumulators meet in LDS; wave kk finishes pixels 16 * MPW * kk .. of the tile ----
  if constexpr (KSPLIT > 1) {
    __syncthreads();                     // the patch is dead
    char* red = smem + ((wn * KSPLIT + kk) * (MI * NI)) * 1024 + lane * 16;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (mi / MPW != kk) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x4*>(red + (mi * NI + ni) * 1024) = acc[mi][ni];
      }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      if (mi / MPW == kk) {
#pragma unroll
        for (int k2 = 0; k2 < KSPLIT; ++k2)
          if (k2 != kk) {
            const char* src = smem + ((wn * KSPLIT + k2) * (MI * NI)) * 1024 + lane * 16;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + (mi * NI + ni) * 1024);
              acc[mi][ni] = f32x4{acc[mi][ni][0] + v[0], acc[mi][ni][1] + v[1], acc[mi][ni][2] + v[2], acc[mi][ni][3] + v[3]};
            }
          }
      }
  }

  // ---- epilogue (as conv_igemm_kernel's): lane (lr, lq) holds channels 4 lq .. 4 lq + 3 of pixel lr per 16x16 tile ----
  constexpr bool PAIR = E::DTYPE == 1;
  constexpr int NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.ablate & 1) ? 0u : p.obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);
  const int co_w = co0 + 4 * lq;
  const int co_s = PAIR ? co0 + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
  f32x4 bias4[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bias4[ni] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co_w + ni * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
  // final values of pixel tile mi (accumulator + bias + emb + residual), and where they go.  The loads (residual, emb) are issued for
  // every tile of the wave first (final_issue) and consumed afterwards (final_o): whatever is issued in between - the weight touches
  // of the fused-GroupNorm epilogue - then queues BEHIND them (a wave's loads return in order)
  u32x4 rr_all[MI][NP2]; f32x4 ev_all[MI][NI]; uint32_t opx_all[MI]; int n_all[MI];   // opx: output pixel index (one past the tensor's last for a pixel outside it: every offset built from it is out of range)
  auto final_issue = [&](auto mic) {
    constexpr int mi = decltype(mic)::value;
    const int m = mi * 16 + lr;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, gi = m >> (p.lvw + p.lth);
    const int n = n0 + gi;
    n_all[mi] = n;
    const bool ok = n < p.N && ty < p.Ho && tx < p.Wo;
    const uint32_t opix = ok ? (uint32_t)((n * p.Ho + ty) * p.Wo + tx) : (uint32_t)(p.N * p.Ho * p.Wo);
    opx_all[mi] = opix;
    const uint32_t rvo = (ok && p.res_mode != RES_NONE) ? (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.rbytes;
    if (p.res_mode != RES_NONE) {
#pragma unroll
      for (int k = 0; k < NP2; ++k) rr_all[mi][k] = buf_load16(rsr, rvo + k * PSTEP * ESZ, 0);
    }
    if (p.emb) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) ev_all[mi][ni] = *reinterpret_cast<const f32x4*>(p.emb + (size_t)min(n, p.N - 1) * p.emb_stride + co_w + ni * 16);
    }
  };
  auto final_o = [&](auto mic, f32x4 (&o)[NI], uint32_t& opix, int& n) {
    constexpr int mi = decltype(mic)::value;
    opix = opx_all[mi]; n = n_all[mi];
    const u32x4 (&rr)[NP2] = rr_all[mi];
    f32x4 ad[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      ad[ni] = bias4[ni];
      if (p.emb) {
        const f32x4 ev = ev_all[mi][ni];
        ad[ni] = f32x4{ad[ni][0] + ev[0], ad[ni][1] + ev[1], ad[ni][2] + ev[2], ad[ni][3] + ev[3]};
      }
    }
    if constexpr (!PAIR) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        o[ni] = f32x4{acc[mi][ni][0] + ad[ni][0], acc[mi][ni][1] + ad[ni][1], acc[mi][ni][2] + ad[ni][2], acc[mi][ni][3] + ad[ni][3]};
        if (p.res_mode != RES_NONE) { const f32x4 t = __builtin_bit_cast(f32x4, rr[ni]); o[ni] = f32x4{o[ni][0] + t[0], o[ni][1] + t[1], o[ni][2] + t[2], o[ni][3] + t[3]}; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.res_mode != RES_NONE) {   // un-swap the 8-channel residual piece back to the accumulator layout
          const auto s0 = __builtin_amdgcn_permlane16_swap(rr[k][0], rr[k][2], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(rr[k][1], rr[k][3], false, false);
          const uint32_t xa[2] = {s0[0], s1[0]}, xb[2] = {s0[1], s1[1]};
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unpack2(xa[j], ra[2 * j], ra[2 * j + 1], T());
            unpack2(xb[j], rb[2 * j], rb[2 * j + 1], T());
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[2 * k][j] = acc[mi][2 * k][j] + ad[2 * k][j] + ra[j];
          o[2 * k + 1][j] = acc[mi][2 * k + 1][j] + ad[2 * k + 1][j] + rb[j];
        }
      }
    }
  };
  auto store_vals = [&](const f32x4 (&o)[NI], const __amdgpu_buffer_rsrc_t& rs, uint32_t ovo) {
    if constexpr (!PAIR) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[ni]), rs, ovo + ni * 16 * ESZ, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < NP2; ++k) {
        const u32x2 pa2 = pack4(o[2 * k], T()), pb2 = pack4(o[2 * k + 1], T());
        const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
        const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rs, ovo + k * PSTEP * ESZ, 0, 0);
      }
    }
  };
  auto for_mine = [&](auto f) {           // the pixel tiles this wave finishes
    if constexpr (KSPLIT == 1) { f(IC<0>()); f(IC<1>()); f(IC<2>()); f(IC<3>()); }
    else if constexpr (KSPLIT == 2) {
      if (kk == 0) { f(IC<0>()); f(IC<1>()); } else { f(IC<2>()); f(IC<3>()); }
    } else {
      if (kk == 0) f(IC<0>()); else if (kk == 1) f(IC<1>()); else if (kk == 2) f(IC<2>()); else f(IC<3>());
    }
  };
  // (arrays written and read under DIFFERENT runtime branches end up in scratch memory: with K-sharing waves a tile's loads and its
  // arithmetic stay inside one branch)
  constexpr bool W8 = W8_ != 0;
  constexpr bool ACT_OK = !(W8 && KSPLIT > 1);   // an 8x8 image split over K-sharing waves has no wave-local statistics (the host never asks)
  auto plain_tile = [&](auto mic) {
    f32x4 o[NI]; uint32_t opix; int n;
    final_o(mic, o, opix, n);
    store_vals(o, rso, (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ);
  };
  if (!ACT_OK || p.act_out == nullptr) {
    if constexpr (KSPLIT == 1) { for_mine(final_issue); for_mine(plain_tile); }
    else for_mine([&](auto mic) { final_issue(mic); plain_tile(mic); });
    return;
  }
  if constexpr (ACT_OK) {
  // ---- epilogue with the consumers' GroupNorm (+ SiLU) applied: GroupNorm32 (AD/image_diffusion/nn.py:11-13,87-94) over the conv's
  // final values, fp32 statistics; a wave holds whole images (8x8: its four pixel tiles are one image; 4x4: pixel tile mi is image
  // mi) x 64 (32) channels = whole groups (cpg = 4 / 8 / 16: a lane's quad, two or four lq rows), so the statistics are wave-local:
  // a DPP row sum over the 16 pixels of a tile row + one or two shuffles.  FiLM as in gn_affine_kernel.  Up to two sites read the
  // tensor (ops.h ConvDesc::act2_out): the row sums are shared, the group width, (gamma, beta) and destination are each site's own. ----
  constexpr bool FAST = E::DTYPE == 1;
  struct ActSite { __amdgpu_buffer_rsrc_t rs; const float* gamma; const float* beta; const float* film; int film_stride, silu, stride, coff, cpg; float inv_cnt; };
  const float npx = (float)(p.Ho * p.Wo);
  const ActSite st1{__builtin_amdgcn_make_buffer_rsrc(p.act_out, 0, (p.ablate & 1) ? 0u : p.abytes, 0x00020000), p.act_gamma, p.act_beta, p.act_film,
                    p.act_film_stride, p.act_silu, p.act_stride, p.act_coff, p.act_cpg, 1.0f / ((float)p.act_cpg * npx)};
  const bool two = p.act2_out != nullptr;
  const ActSite st2{__builtin_amdgcn_make_buffer_rsrc(two ? p.act2_out : p.act_out, 0, (two && !(p.ablate & 1)) ? p.a2bytes : 0u, 0x00020000), p.act2_gamma, p.act2_beta,
                    nullptr, 0, p.act2_silu, p.act2_stride, p.act2_coff, two ? p.act2_cpg : 4, 1.0f / ((float)(two ? p.act2_cpg : 4) * npx)};
  // Every parameter load of the epilogue is ISSUED before the weight touches below (a wave's loads return in order: a load issued behind the
  // cold touches waits for HBM): (gamma, beta) of both sites here, FiLM (scale, shift) rows with the residual / emb loads of their tile.
  f32x4 gam1[NI], bet1[NI], gam2[NI], bet2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    gam1[ni] = *reinterpret_cast<const f32x4*>(p.act_gamma + co_w + ni * 16);
    bet1[ni] = *reinterpret_cast<const f32x4*>(p.act_beta + co_w + ni * 16);
    gam2[ni] = *reinterpret_cast<const f32x4*>((two ? p.act2_gamma : p.act_gamma) + co_w + ni * 16);
    bet2[ni] = *reinterpret_cast<const f32x4*>((two ? p.act2_beta : p.act_beta) + co_w + ni * 16);
  }
  auto film_issue = [&](int n, f32x4 (&fsc)[NI], f32x4 (&fsh)[NI]) {
    if (!p.act_film) return;
    const float* fp = p.act_film + (size_t)min(n, p.N - 1) * p.act_film_stride + co_w;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fsc[ni] = *reinterpret_cast<const f32x4*>(fp + ni * 16);
      fsh[ni] = *reinterpret_cast<const f32x4*>(fp + p.Cout + ni * 16);
    }
  };
  // (a, b) of one image's channels co_w + 16 ni + j for one site, from the row sums (rs, rq) of the values and their squares
  auto site_ab = [&](const ActSite& st, const f32x4 (&gam)[NI], const f32x4 (&bet)[NI], const f32x4 (&fsc)[NI], const f32x4 (&fsh)[NI],
                     const float (&rs)[NI], const float (&rq)[NI], f32x4 (&A)[NI], f32x4 (&Bv)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float gs = rs[ni], gq = rq[ni];
      if (st.cpg >= 8) { gs += __shfl_xor(gs, 16); gq += __shfl_xor(gq, 16); }
      if (st.cpg >= 16) { gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32); }
      const float mean = gs * st.inv_cnt;
      const float var = fmaxf(gq * st.inv_cnt - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + p.act_eps);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = rstd * gam[ni][j];
        float b = bet[ni][j] - mean * a;
        if (st.film) {
          const float sc = 1.0f + fsc[ni][j];
          a *= sc;
          b = b * sc + fsh[ni][j];
        }
        A[ni][j] = a; Bv[ni][j] = b;
      }
    }
  };
  // y = silu?(a o + b) of pixel tile values o, to the site's tensor
  auto apply_store = [&](const ActSite& st, const f32x4 (&o)[NI], const f32x4 (&A)[NI], const f32x4 (&Bv)[NI], uint32_t opix) {
    f32x4 y[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = A[ni][j] * o[ni][j] + Bv[ni][j];
        y[ni][j] = st.silu ? (FAST ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)) : v / (1.0f + expf(-v))) : v;
      }
    store_vals(y, st.rs, (opix * (uint32_t)st.stride + (uint32_t)(st.coff + co_s)) * ESZ);
  };
  // Touch of the NEXT conv's packed weights (the pass that used to warm the L2s with them is gone; as common.h l2_warm_wave, spread
  // over all waves: the workgroups of an XCD cover the range once).  A wave's loads return in order, so the cold touches go behind
  // the last load the epilogue waits for, and their latency overlaps the normalisation; measured alternatives: before the epilogue's
  // arithmetic (same), behind the first chunk's weights at the start of the kernel (the weight stream stalls: 2-4 us slower).
  uint32_t wv[4] = {0u, 0u, 0u, 0u};
  auto warm_issue = [&]() {
    if (p.warm_bytes == 0) return;
    const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    const uint32_t rank = wg >> 3, per = (nwg + 7) >> 3, step = per * (uint32_t)NT * 128u;
    uint32_t off = (rank * (uint32_t)NT + tid) * 128u;
    const char* base = reinterpret_cast<const char*>(p.warm);
#pragma unroll
    for (int i = 0; i < 4; ++i, off += step) wv[i] = *reinterpret_cast<const uint32_t*>(base + (off < p.warm_bytes ? off : 0u));
  };
  if constexpr (W8) {                     // KSPLIT == 1: the wave's four pixel tiles are one image
    for_mine(final_issue);
    f32x4 fsc[NI] = {}, fsh[NI] = {};
    film_issue(n0, fsc, fsh);
    float ls[NI] = {}, lq2[NI] = {};
    for_mine([&](auto mic) {
      constexpr int mi = decltype(mic)::value;
      f32x4 o[NI]; uint32_t opix; int n;
      final_o(mic, o, opix, n);
      if (p.act_raw) store_vals(o, rso, (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        acc[mi][ni] = o[ni];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ls[ni] += o[ni][j]; lq2[ni] += o[ni][j] * o[ni][j]; }
      }
    });
    warm_issue();
    float rs[NI], rq[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { rs[ni] = GnPartial<1>::row_sum(ls[ni]); rq[ni] = GnPartial<1>::row_sum(lq2[ni]); }
    auto one_site = [&](const ActSite& st, const f32x4 (&gam)[NI], const f32x4 (&bet)[NI]) {
      f32x4 A[NI], Bv[NI];
      site_ab(st, gam, bet, fsc, fsh, rs, rq, A, Bv);
      for_mine([&](auto mic) {
        constexpr int mi = decltype(mic)::value;
        apply_store(st, acc[mi], A, Bv, opx_all[mi]);
      });
    };
    one_site(st1, gam1, bet1);
    if (two) one_site(st2, gam2, bet2);
  } else {                                // pixel tile mi is image mi
    bool warmed = false;
    f32x4 fsc_all[MI][NI] = {}, fsh_all[MI][NI] = {};
    auto film_tile = [&](auto mic) {
      constexpr int mi = decltype(mic)::value;
      film_issue(n_all[mi], fsc_all[mi], fsh_all[mi]);
    };
    auto act_tile = [&](auto mic) {
      constexpr int mi = decltype(mic)::value;
      f32x4 o[NI]; uint32_t opix; int n;
      final_o(mic, o, opix, n);
      if (p.act_raw) store_vals(o, rso, (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ);
      if (!warmed) { warm_issue(); warmed = true; }
      float rs[NI], rq[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += o[ni][j]; s2 += o[ni][j] * o[ni][j]; }
        rs[ni] = GnPartial<1>::row_sum(s1); rq[ni] = GnPartial<1>::row_sum(s2);
      }
      f32x4 A[NI], Bv[NI];
      site_ab(st1, gam1, bet1, fsc_all[mi], fsh_all[mi], rs, rq, A, Bv);
      apply_store(st1, o, A, Bv, opix);
      if (two) {
        site_ab(st2, gam2, bet2, fsc_all[mi], fsh_all[mi], rs, rq, A, Bv);
        apply_store(st2, o, A, Bv, opix);
      }
    };
    if constexpr (KSPLIT == 1) { for_mine(final_issue); for_mine(film_tile); for_mine(act_tile); }
    else for_mine([&](auto mic) { final_issue(mic); film_tile(mic); act_tile(mic); });
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wv[i]));   // the touches have landed before the wave ends
  }
}

// Shapes this kernel takes: 3x3, whole 8x8 / 4x4 OUTPUT images, stride 1 (plain or nearest-x2 input) or - round 5 - stride 2 (the Downsample convs
// 16 -> 8 and 8 -> 4, AD/image_diffusion/unet.py:217-240: the staged patch is the (2 Ho + 1)^2 input window, the fragment addresses step two
// pixels per output pixel: 2-way bank conflicts on reads that are a tenth of this kernel's LDS budget), no GroupNorm prologue (these
// levels normalise in a pass of their own), NHWC output with C_out % 128 == 0, residual at the output resolution or none.
// 0 = launched, 1 = not eligible (the caller goes on to the plain kernel), < 0 = error
// dry: eligibility only (the walker decides about a fused skip conv before it reaches the conv that would carry it)
template <typename T>
int launch_small(const ConvKArgs& a0, int enabled, int ks, hipStream_t s, int* act_done, bool dry = false) {
  ConvKArgs a = a0;
  if (!enabled || ks != 3 || a.out_mode != OUT_NHWC || a.pro_a) return 1;
  const bool s2 = a.stride == 2;
  const bool skip = a.sk0 != nullptr;
  if (skip && (!(enabled & 8) || s2 || a.mode != CONV_UNIT || a.res_mode != RES_NONE || a.src1 || a.nchunks != a.nreal)) return 1;
  if (s2 && (!(enabled & 4) || a.mode != CONV_STRIDE2 || a.Hc != 2 * a.Ho || a.Wc != 2 * a.Wo)) return 1;
  if (a.Ho != a.Wo || (a.Ho != 8 && a.Ho != 4) || a.Cout % 128 != 0) return 1;
  if (a.res_mode != RES_NONE && a.res_mode != RES_SAME) return 1;
  constexpr int CH = Elem<T>::CHUNK;
  if (a.C0 % CH || a.C1 % CH) return 1;
  const bool w8 = a.Ho == 8;
  sm::Args g;
  a.lvw = a.lth = w8 ? 3 : 2;
  a.G = w8 ? 1 : 4;
  a.PW = a.PH = s2 ? 2 * a.Ho + 1 : a.Ho + 2;
  g.pwp = s2 ? a.PW : (w8 ? 16 : 8); g.sh = w8 ? 1 : 2;
  g.pimg = a.PH * g.pwp;
  g.plane = (a.G * g.pimg - (g.pwp - a.PW)) * 64;
  const int nchp = s2 ? (w8 ? 8 : 4) : (w8 ? 16 : 8);
  g.nphase = (a.nchunks + nchp - 1) / nchp;
  g.chp = a.nchunks / g.nphase;
  if (g.chp * g.nphase != a.nchunks) return 1;
  const int tiles = (a.N + a.G - 1) / a.G;
  // waves along N: as many as keep every CU busy (each halving doubles the workgroups and splits K between wave pairs)
  const int ncu = ws_num_cus();
  int nw = 4;
  while (nw > 1 && (long)tiles * (a.Cout / (64 * nw)) < ncu) nw >>= 1;
  if (a.Cout % (64 * nw)) return 1;
  const int ksplit = 4 / nw;
  if (g.chp % ksplit) return 1;
  g.red_bytes = ksplit > 1 ? 4 * 16 * 1024 : 0;
  g.sphase = g.schp = g.sup = 0;
  a.wstride = (uint32_t)a.nchunks * 9;
  if (skip) {
    if (a.SC0 % CH || a.SC1 % CH) return 1;
    const int ns = (a.SC0 + a.SC1) / CH;
    g.sup = g.nphase == 1 && ns <= sm::SDMAX && ns % ksplit == 0 && (size_t)g.chp * g.plane + (size_t)ns * sm::SPLANE <= 160 * 1024;
    g.sphase = g.sup ? 1 : (ns + nchp - 1) / nchp;
    g.schp = ns / g.sphase;
    if (g.schp * g.sphase != ns || g.schp % ksplit) return 1;
    a.wstride += (uint32_t)ns;
  }
  const size_t lds = g.sup ? std::max((size_t)g.chp * g.plane + (size_t)g.schp * sm::SPLANE, (size_t)g.red_bytes)
                           : std::max((size_t)std::max(g.chp, g.schp) * g.plane, (size_t)g.red_bytes);
  if (lds > 160 * 1024) return 1;
  if (dry) return 0;
  a.gn_stats = nullptr; a.gn_slots = 0;
  {   // GroupNorm of the output in the epilogue: whole images per wave (an 8x8 image split over K-sharing waves is not), 4 / 8 / 16 channels per group
    auto cpg_ok = [&](int cpg, int coff) { return (cpg == 4 || cpg == 8 || cpg == 16) && a.Cout % cpg == 0 && coff % cpg == 0; };
    const bool ok = a.act_out && cpg_ok(a.act_cpg, a.act_coff) && (!w8 || ksplit == 1);
    if (!ok) { a.act_out = nullptr; a.warm = nullptr; a.warm_bytes = 0; }
    const bool ok2 = ok && a.act2_out && cpg_ok(a.act2_cpg, a.act2_coff);
    if (!ok2) a.act2_out = nullptr;
    if (act_done) *act_done = (ok ? 1 : 0) | (ok2 ? 2 : 0);
  }
  // whole-chip launches of the 8x8 level (one image per workgroup, no K split): eight waves of 32 channels (knob conv_small bit 1)
  const bool w8x2 = w8 && !s2 && !skip && ksplit == 1 && (enabled & 2) && a.Cout % 256 == 0;
  dim3 grid(tiles, a.Cout / (64 * nw));
  int rc = 0;
  auto go = [&](auto kern, int threads) {
    rc = mi355_allow_big_lds(kern, "conv3x3 (small levels)");
    if (rc == 0) hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, a, g);
  };
  if (s2) {   // (2 Ho + 1)^2 patch pixels per image: 289 x 4 / 256 -> 5 fragments per thread (8x8), 4 x 81 x 4 / 256 -> 6 (4x4)
    if (w8) {
      if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 8, 5, 4, 1>, 256);
      else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 8, 5, 4, 1>, 256);
      else go(conv3x3_small_kernel<T, 4, 8, 5, 4, 1>, 256);
    } else {
      if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 4, 6, 4, 0>, 256);
      else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 4, 6, 4, 0>, 256);
      else go(conv3x3_small_kernel<T, 4, 4, 6, 4, 0>, 256);
    }
  } else if (skip) {
    if (w8) {
      if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 16, 2, 4, 1, 1>, 256);   // (the eight-wave form has no registers left for the skip planes' staging: 240 B of scratch)
      else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 16, 2, 4, 1, 1>, 256);
      else go(conv3x3_small_kernel<T, 4, 16, 2, 4, 1, 1>, 256);
    } else {
      if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 8, 3, 4, 0, 1>, 256);
      else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 8, 3, 4, 0, 1>, 256);
      else go(conv3x3_small_kernel<T, 4, 8, 3, 4, 0, 1>, 256);
    }
  } else if (w8) {
    if (w8x2) go(conv3x3_small_kernel<T, 1, 16, 1, 2>, 512);
    else if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 16, 2>, 256);
    else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 16, 2>, 256);
    else go(conv3x3_small_kernel<T, 4, 16, 2>, 256);
  } else {
    if (ksplit == 1) go(conv3x3_small_kernel<T, 1, 8, 3>, 256);
    else if (ksplit == 2) go(conv3x3_small_kernel<T, 2, 8, 3>, 256);
    else go(conv3x3_small_kernel<T, 4, 8, 3>, 256);
  }
  return rc;
}
