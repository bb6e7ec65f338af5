// Warp-specialised, persistent 3x3 implicit-GEMM (included by conv_igemm.hip; same args, same LDS images).
//
// Measured on the plain kernel (in-kernel stamps + ablations): its phases (patch staging + GN/SiLU
// prologue, weight staging, MFMA rows, epilogue stores) simply ADD UP - two co-resident workgroups run them in lockstep
// and a wave is in-order, so the MFMA pipe idles ~70 % of the time.  Here the overlap is built in:
//   * one 8-wave workgroup per CU; waves 0-3 are LOADERS (global -> registers -> prologue -> LDS), waves 4-7 are
//     CONSUMERS (LDS -> MFMA -> epilogue).  Each SIMD hosts one of each: VALU / VMEM / LDS-write work next to MFMA work;
//   * persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; the loaders treat the whole walk as ONE stream of
//     (tile, chunk, kernel row) items and run ahead of the consumers across tile boundaries:
//        global loads  : 2 chunks (6 kernel rows) ahead, in register rings (patch fragments, weight rows, GN (a, b))
//        LDS commits   : patch chunk q+1 is transformed and written (one fragment per kernel row) while the consumers
//                        multiply chunk q (3 patch planes); weight row g+1 while they multiply row g (2 buffers)
//     so HBM/L2 latency, the prologue VALU work and the LDS writes all sit beside the MFMA rows, ONE barrier per row;
//   * the tile is 8 x 16 pixels (10 x 18 = 180 patch pixels = exactly 3 fragments per loader thread, 1.4x halo)
//     instead of the plain kernel's 4 x 32 (204 pixels -> 4 fragments, 1.6x halo): 25 % less prologue work.
// Every wave reaches every barrier of the schedule (both roles execute exactly one barrier per kernel row of every tile).
#ifndef WS_ABLATE
#define WS_ABLATE 0   // diagnostic builds only: 4 = consumers skip LDS reads + MFMA, 8 = no weight loads, 16 = no patch loads, 64 = no weight LDS writes
#endif
template <typename T, int PIT, bool PRO>
__global__ void __launch_bounds__(512, 2) conv3x3_ws_kernel(ConvKArgs p, int n_mt, int n_nt) {
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, ESZ = sizeof(T);
  constexpr bool FAST = (E::DTYPE == 1);
  constexpr int BN = 128, WN = 2, WTM = 64, WTN = 64, MI = 4, NI = 4;
  constexpr int WTILE = BN * 64, WIT = 3 * WTILE / (256 * 16), FR = 64, PLANE = PIT * FR * PROW;
  static_assert(PIT == 3, "one patch fragment per kernel row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pbuf = smem;                 // 3 patch planes
  char* wbuf = smem + 3 * PLANE;     // 2 x (3 weight tiles)

  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave8 < 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int ngr = p.nchunks * 3;     // kernel rows (barrier intervals) per tile
  // Tile walk: items t = 0 .. ntp-1; 8 consecutive workgroups (one per XCD) take 8 consecutive pixel tiles, and the
  // workgroup 8 further on (same XCD, same L2) takes the next channel tile of the same pixels.
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt;
  auto decode = [&](int t, int& mt, int& nt) {
    const int per = 8 * n_nt, blk = t / per, r = t - blk * per;
    nt = r >> 3; mt = blk * 8 + (r & 7);
  };
  auto next_valid = [&](int t) {     // next item of this workgroup's walk that is a real tile (or >= ntp)
    for (t += gridDim.x; t < ntp; t += gridDim.x) { int mt, nt; decode(t, mt, nt); if (mt < n_mt) break; }
    return t;
  };
  auto origin = [&](int mt, int& n0, int& y0, int& x0) {
    const int ng = mt / tpi, rem = mt - ng * tpi;
    const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
    n0 = ng; y0 = tyi << p.lth; x0 = txi << p.lvw;
  };
  int t_first = (int)blockIdx.x - (int)gridDim.x;
  t_first = next_valid(t_first);
  if (t_first >= ntp) return;        // whole workgroup leaves together: no barrier is ever reached

  if (loader) {
    // ================================= LOADER waves =================================
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
    const int fq = tid & 3, frow = tid >> 2;
    const int pro = (!PRO || (p.ablate & 2)) ? 0 : (p.pro_silu ? 2 : 1);
    const int pimg = p.PH * p.PW;
    const float inv_pw = 1.0f / (float)p.PW;
    uint32_t woff[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) woff[i] = (i * 256 + tid) * 16;

    // ---- load stream: geometry of the tile the global loads currently address ----
    uint32_t voff0[PIT], voff1[PIT], vmask_ld = 0; int n0_ld = 0;
    auto setup = [&](int mt) {
      int n0, y0, x0;
      origin(mt, n0, y0, x0);
      n0_ld = n0; vmask_ld = 0;
      const int cy0 = y0 - 1, cx0 = x0 - 1;
#pragma unroll
      for (int u = 0; u < PIT; ++u) {
        const int i = frow + u * FR;
        int s = -1;
        if (i < pimg) {
          const int py = (int)(((float)i + 0.5f) * inv_pw), px = i - py * p.PW;   // exact: i < 256, PW <= 18
          const int cy = cy0 + py, cx = cx0 + px;
          if (cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
            if (p.mode == CONV_UP2) s = (n0 * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1);
            else s = (n0 * p.Hs + cy) * p.Ws + cx;
          }
        }
        voff0[u] = s >= 0 ? (uint32_t)s * (uint32_t)(p.C0 * ESZ) + fq * 16 : p.bytes0;
        voff1[u] = s >= 0 ? (uint32_t)s * (uint32_t)(p.C1 * ESZ) + fq * 16 : p.bytes1;
        if (s >= 0) vmask_ld |= 1u << u;
      }
    };
    int ld_t = t_first, ld_c = 0;
    { int mt, nt; decode(ld_t, mt, nt); setup(mt); }
    auto ld_advance = [&]() {
      if (++ld_c == p.nchunks) {
        ld_c = 0;
        ld_t = next_valid(ld_t);
        if (ld_t < ntp) { int mt, nt; decode(ld_t, mt, nt); setup(mt); }
      }
    };
    u32x4 raw[2][PIT] = {}; float pa[2][V] = {}, pb[2][V] = {};
    // Every load below is issued unconditionally (a finished stream keeps re-reading its last valid addresses and the data
    // is never committed): a load under a branch makes the compiler's s_waitcnt bookkeeping assume the shortest queue,
    // i.e. vmcnt(0) at every use, which would drain the whole run-ahead.
    auto issue_frag = [&](auto slotc, auto uc) {
      constexpr int slot = decltype(slotc)::value, u = decltype(uc)::value;
      const int cb = ld_c * CHUNK;
      const bool first = cb < p.C0;
      if constexpr (!(WS_ABLATE & 16)) raw[slot][u] = buf_load16(first ? rs0 : rs1, first ? voff0[u] : voff1[u], (first ? cb : cb - p.C0) * ESZ);
    };
    auto issue_ab = [&](auto slotc) {
      constexpr int slot = decltype(slotc)::value;
      if constexpr (PRO) {
        const float* ap = p.pro_a + (size_t)n0_ld * p.Cin + ld_c * CHUNK + fq * V;
        const float* bp = p.pro_b + (size_t)n0_ld * p.Cin + ld_c * CHUNK + fq * V;
#pragma unroll
        for (int j = 0; j < V; ++j) { pa[slot][j] = ap[j]; pb[slot][j] = bp[j]; }
      }
    };
    auto commit_frag = [&](auto slotc, auto uc, int plane, uint32_t mask) {
      constexpr int slot = decltype(slotc)::value, u = decltype(uc)::value;
      u32x4 outv = raw[slot][u];
      if (pro) {
        float f[V];
        frag_to_float(raw[slot][u], f, T());
        if (pro == 2) {
#pragma unroll
          for (int j = 0; j < V; ++j) f[j] = silu_fast<FAST>(pa[slot][j] * f[j] + pb[slot][j]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) f[j] = pa[slot][j] * f[j] + pb[slot][j];
        }
        outv = float_to_frag(f, T());
        if (!((mask >> u) & 1u)) outv = u32x4{0u, 0u, 0u, 0u};   // zero padding applies AFTER the prologue
      }
      *reinterpret_cast<u32x4*>(pbuf + plane * PLANE + (frow + u * FR) * PROW + fq * 16) = outv;
    };

    // ---- weight stream: 6 kernel rows ahead of the row being committed ----
    int wl_t = t_first, wl_row = 0, wl_nt;
    { int mt; decode(wl_t, mt, wl_nt); }
    u32x4 wreg[6][WIT] = {};
    auto prefetch_w = [&](auto slotc) {
      constexpr int slot = decltype(slotc)::value;
      const uint32_t so = ((uint32_t)wl_nt * p.nchunks * 9 + (uint32_t)wl_row * 3) * WTILE;
      if constexpr (!(WS_ABLATE & 8)) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) wreg[slot][i] = buf_load16(rsw, woff[i], so);
      }
      if (++wl_row == ngr) {
        wl_row = 0;
        wl_t = next_valid(wl_t);
        if (wl_t < ntp) { int mt; decode(wl_t, mt, wl_nt); }
      }
    };
    auto commit_w = [&](auto slotc, int buf) {
      constexpr int slot = decltype(slotc)::value;
      char* dst = wbuf + buf * (3 * WTILE) + tid * 16;
      if constexpr (!(WS_ABLATE & 64)) {
#pragma unroll
        for (int i = 0; i < WIT; ++i) *reinterpret_cast<u32x4*>(dst + i * 256 * 16) = wreg[slot][i];
      }
    };

    STAMP_DECL
    // ---- fill the pipeline: chunks 0, 1 and rows 0..5 in flight, chunk 0 committed ----
    issue_frag(IC<0>(), IC<0>()); issue_frag(IC<0>(), IC<1>()); issue_frag(IC<0>(), IC<2>()); issue_ab(IC<0>());
    ld_advance();
    issue_frag(IC<1>(), IC<0>()); issue_frag(IC<1>(), IC<1>()); issue_frag(IC<1>(), IC<2>()); issue_ab(IC<1>());
    uint32_t vmask_cm = vmask_ld;      // chunks 0 and 1 belong to the first tile (nchunks >= 2)
    prefetch_w(IC<0>()); prefetch_w(IC<1>()); prefetch_w(IC<2>()); prefetch_w(IC<3>()); prefetch_w(IC<4>()); prefetch_w(IC<5>());
    ld_advance();
    commit_frag(IC<0>(), IC<0>(), 0, vmask_cm); issue_frag(IC<0>(), IC<0>());
    commit_frag(IC<0>(), IC<1>(), 0, vmask_cm); issue_frag(IC<0>(), IC<1>());
    commit_frag(IC<0>(), IC<2>(), 0, vmask_cm); issue_frag(IC<0>(), IC<2>());
    issue_ab(IC<0>());

    STAMP(0)
    int gcnt = 0, plane = 1;           // next weight buffer parity; plane the NEXT chunk is committed into
    for (int t = t_first; t < ntp;) {
      const int t_next = next_valid(t);
      // one chunk of the consumer stream: its three weight rows + the commit of the chunk AFTER it
      auto body = [&](int c, auto ws0c, auto psc) {
        constexpr int WS0 = decltype(ws0c)::value, PS = decltype(psc)::value;
        bool nx_ok = true;
        if (c + 1 == p.nchunks) {      // the next chunk opens the next tile; the load stream is at its chunk 1
          nx_ok = t_next < ntp;
          vmask_cm = vmask_ld;
        }
        ld_advance();                  // -> two chunks after the chunk committed below
        auto interval = [&](auto kyc) {
          constexpr int ky = decltype(kyc)::value;
          commit_w(IC<WS0 + ky>(), gcnt & 1); ++gcnt;
          STAMP(1)
          prefetch_w(IC<WS0 + ky>());
          STAMP(3)
          if (nx_ok) commit_frag(IC<PS>(), IC<ky>(), plane, vmask_cm);
          STAMP(2)
          issue_frag(IC<PS>(), IC<ky>());
          if (ky == 2) issue_ab(IC<PS>());
          STAMP(3)
          __syncthreads();             // kernel row (t, c, ky) is in LDS; after ky == 2 so is the next chunk's patch
          STAMP(4)
        };
        interval(IC<0>()); interval(IC<1>()); interval(IC<2>());
        plane = plane == 2 ? 0 : plane + 1;
      };
      for (int c = 0; c < p.nchunks; c += 2) { body(c, IC<0>(), IC<1>()); body(c + 1, IC<3>(), IC<0>()); }
      t = t_next;
    }
    STAMP_FLUSH
  } else {
    // ================================= CONSUMER waves =================================
    __builtin_amdgcn_s_setprio(2);     // the MFMA stream goes first when both waves of a SIMD are ready
    const int wave = wave8 - 4;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lq = lane >> 4;
    const int VWm = (1 << p.lvw) - 1, THm = (1 << p.lth) - 1;
    int arow[MI], a1[MI], a2[MI], brow[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = wm * WTM + mi * 16 + lr;
      const int tx = m & VWm, ty = (m >> p.lvw) & THm;
      arow[mi] = (ty * p.PW + tx) * PROW + lq * 16;
      a1[mi] = arow[mi] + p.PW * PROW; a2[mi] = arow[mi] + 2 * p.PW * PROW;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = wn * WTN + ni * 16 + lr;
      brow[ni] = row * 64 + 16 * (lq ^ ((row >> 1) & 3));
    }
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.ablate & 1) ? 0u : p.obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);
    constexpr bool PAIR = E::DTYPE == 1;
    constexpr int NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;

    int gcnt = 0, plane = 0;
    STAMP_DECL
    for (int t = t_first; t < ntp; t = next_valid(t)) {
      int mt, nt, n0, y0, x0;
      decode(t, mt, nt);
      origin(mt, n0, y0, x0);
      f32x4 acc[MI][NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < p.nchunks; ++c) {
        const char* patch = pbuf + plane * PLANE;
        auto row = [&](const int (&ao)[MI]) {
          STAMP(7)
          __syncthreads();                   // the loaders have finished staging this kernel row
          STAMP(5)
          const char* wt = wbuf + (gcnt & 1) * (3 * WTILE);
          ++gcnt;
          if constexpr (WS_ABLATE & 4) return;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            u32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(patch + ao[mi] + kx * PROW);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(wt + kx * WTILE + brow[ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], b[ni], a[mi], T());
          }
        };
        row(arow); STAMP(6) row(a1); STAMP(6) row(a2); STAMP(6)
        plane = plane == 2 ? 0 : plane + 1;
      }
      // ---- epilogue (same as the plain kernel): all loads first, 16-byte stores ----
      const int co_w = nt * BN + wn * WTN + 4 * lq;
      const int co_s = PAIR ? nt * BN + wn * WTN + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
      uint32_t ovo[MI], rvo[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = wm * WTM + mi * 16 + lr;
        const int tx = m & VWm, ty = (m >> p.lvw) & THm;
        const int y = y0 + ty, x = x0 + tx;
        const bool ok = y < p.Ho && x < p.Wo && co_s < p.Cout;
        const uint32_t opix = (uint32_t)((n0 * p.Ho + y) * p.Wo + x);
        ovo[mi] = ok ? (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.obytes;
        uint32_t rpix = opix;
        if (p.res_mode == RES_UP2) rpix = (uint32_t)((n0 * p.Hr + (y >> 1)) * p.Wr + (x >> 1));
        rvo[mi] = (ok && p.res_mode != RES_NONE) ? (rpix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.rbytes;
      }
      u32x4 rr[MI][NP2];
      if (p.res_mode != RES_NONE) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int k = 0; k < NP2; ++k)
            rr[mi][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, rvo[mi] + k * PSTEP * ESZ, 0, 0));
      }
      f32x4 add4[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int co = co_w + ni * 16;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (co < p.Cout) {
          if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co);
          if (p.emb) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(p.emb + (size_t)n0 * p.emb_stride + co);
            v = f32x4{v[0] + ev[0], v[1] + ev[1], v[2] + ev[2], v[3] + ev[3]};
          }
        }
        add4[ni] = v;
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if constexpr (!PAIR) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4 o = f32x4{acc[mi][ni][0] + add4[ni][0], acc[mi][ni][1] + add4[ni][1], acc[mi][ni][2] + add4[ni][2], acc[mi][ni][3] + add4[ni][3]};
            if (p.res_mode != RES_NONE) {
              const f32x4 tt = __builtin_bit_cast(f32x4, rr[mi][ni]);
              o = f32x4{o[0] + tt[0], o[1] + tt[1], o[2] + tt[2], o[3] + tt[3]};
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, ovo[mi] + ni * 16 * ESZ, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP2; ++k) {
            float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.res_mode != RES_NONE) {   // un-swap the 8-channel residual piece back to the accumulator layout
              const auto s0 = __builtin_amdgcn_permlane16_swap(rr[mi][k][0], rr[mi][k][2], false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(rr[mi][k][1], rr[mi][k][3], false, false);
              const uint32_t xa[2] = {s0[0], s1[0]}, xb[2] = {s0[1], s1[1]};
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                ra[2 * j] = __builtin_bit_cast(float, xa[j] << 16); ra[2 * j + 1] = __builtin_bit_cast(float, xa[j] & 0xffff0000u);
                rb[2 * j] = __builtin_bit_cast(float, xb[j] << 16); rb[2 * j + 1] = __builtin_bit_cast(float, xb[j] & 0xffff0000u);
              }
            }
            bf16x4 ta, tb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              ta[j] = (bf16)(acc[mi][2 * k][j] + add4[2 * k][j] + ra[j]);
              tb[j] = (bf16)(acc[mi][2 * k + 1][j] + add4[2 * k + 1][j] + rb[j]);
            }
            const u32x2 pa2 = __builtin_bit_cast(u32x2, ta), pb2 = __builtin_bit_cast(u32x2, tb);
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo[mi] + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
    }
    STAMP(7)
    STAMP_FLUSH
  }
}

// 0 = launched, 1 = not eligible (caller uses the plain kernel)
template <typename T>
int launch_ws(ConvKArgs a, int BM, int BN, int ks, hipStream_t s) {
  static const int enabled = getenv("MI355_CONV_WS") ? atoi(getenv("MI355_CONV_WS")) : 1;
  if (!enabled || ks != 3 || BM != 128 || BN != 128 || a.G != 1 || a.bn_pack != 128 || a.out_mode != OUT_NHWC) return 1;
  if (a.stride != 1 || a.nchunks < 2 || (a.nchunks & 1)) return 1;   // the register rings are unrolled over chunk pairs
  // 8 x 16 (or 16 x 8) pixel tiles: 180 patch pixels
  const int lw = ilog2_ceil(a.Wo), lh = ilog2_ceil(a.Ho);
  a.lvw = lw < 4 ? lw : 4;
  a.lth = 7 - a.lvw;
  if (a.lth > lh || a.lth > 4) return 1;
  const int VW = 1 << a.lvw, TH = 1 << a.lth;
  a.PW = VW + 2; a.PH = TH + 2; a.NP = a.PW * a.PH;
  if (a.NP > 3 * 64) return 1;
  a.tiles_x = (a.Wo + VW - 1) / VW; a.tiles_y = (a.Ho + TH - 1) / TH;
  const int n_mt = a.N * a.tiles_x * a.tiles_y, n_nt = (a.Cout + 127) / 128;
  constexpr int PIT = 3;
  auto kern = a.pro_a ? conv3x3_ws_kernel<T, PIT, true> : conv3x3_ws_kernel<T, PIT, false>;
  static bool attr_done = false;
  if (!attr_done) {
    for (auto k : {conv3x3_ws_kernel<T, PIT, true>, conv3x3_ws_kernel<T, PIT, false>}) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) (void)hipGetLastError();
    }
    attr_done = true;
  }
  const size_t lds = 3 * (size_t)PIT * 64 * PROW + 2 * 3 * (size_t)128 * 64;
  static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; } return n; }();
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt;
  const int grid = ntp < ncu ? ntp : ncu;   // one persistent workgroup per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, n_mt, n_nt);
  return 0;
}
