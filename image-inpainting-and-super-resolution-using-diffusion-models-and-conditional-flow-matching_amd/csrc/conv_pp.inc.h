// Ping-pong 3x3 implicit GEMM for prologue-free inputs (included by conv_igemm.hip; same ConvKArgs, same packed weights, same NHWC tensors).
//
// Why a second large-level kernel (round 4).  conv3x3_ws_kernel splits a CU into four DMA / prologue waves and four MFMA waves; three rounds
// of stamps and ablations ended at: the MFMA waves alone, operands resident in LDS, take 78 % of the kernel; they sit
// at the 256-register limit, nothing hides their LDS latency, hand-over polls or epilogue, and a loader wave costs the SIMD as many
// registers as an MFMA wave.  This kernel is the structure of cdna_hip_programming.md's 256 x 256 8-phase GEMM instead:
//   * tile = 256 pixels (16 x 16) x 256 output channels; ALL eight waves multiply (wave tile 128 pixels x 64 channels, 2 x 4 waves);
//     the patch of a 64-byte channel chunk is staged once per 256 output channels (the warp-specialised kernel: once per 128);
//   * the waves form two groups (waves 0-3 / 4-7 = the two waves of every SIMD) that run half a step apart: while one group issues its
//     LDS fragment reads and its share of the DMA (an "L" segment), the other runs the 32 MFMAs of a tap (an "M" segment), then they
//     swap - the matrix pipe of a SIMD always has one wave feeding it, and a wave's own reads never sit in front of its own MFMAs, so
//     one fragment register set is enough (176 VGPRs);
//   * operands go global -> LDS by DMA (`buffer_load ... lds`) issued by the MFMA waves themselves: per tap every wave moves 2 of the
//     16 weight pieces (1 KB each) five taps ahead into a 6-tap ring, and on taps 0 / 2 / 4 of a chunk one of the 24 patch pieces of
//     the NEXT chunk into the other patch plane (a plane is 324 pixels x 64 B padded to 24 whole pieces so that every wave issues the
//     same number: the `s_waitcnt vmcnt(N)` that publishes a tap is an immediate).  Nothing is ever drained: the wait at the end of L(s)
//     leaves exactly the pieces of taps s+2 .. s+5 in flight;
//   * ordering is the guide's rule: the issuing wave's counted vmcnt, then a workgroup barrier, then the read one tick later (RAW); a
//     buffer is re-filled only behind the barrier that follows its last reader's segment (WAR).  Two raw s_barriers per tap and wave;
//   * persistent: a workgroup walks tiles like the warp-specialised kernel; the DMA stream is continuous across tile boundaries (the
//     last taps of a tile already fetch the next tile's first weights and patch chunk), the epilogues of the two groups overlap each
//     other and the second group's last MFMA segment; past the end of the walk the stream re-fetches valid addresses into free
//     buffers (never read), so the waits' immediates hold everywhere.
// Tick diagram (T = taps per tile, one raw barrier between ticks):
//   group 0:  L0 | M0 | L1 | M1 | ... | M(T-1) | E ............ | L0' | M0' ...
//   group 1:  -- | L0 | M0 | L1 | ... | L(T-1) | M(T-1) E ..... | --  | L0' ...
// Diagnostic build only (-DCONV_STAMPS): absolute stamp times of taps 8 .. 11 of workgroup 0's first tile, kept in LDS (a global store would
// count in vmcnt and move the counted waits), copied behind the stamp sums at the end: the timeline of the two groups.
#ifdef CONV_STAMPS
#define PP_TRACE(k) if (tr_on) { *reinterpret_cast<volatile unsigned long long*>(smem + pp::LDS_BYTES + ((wave8 * 4 + (tr_tap - 8)) * 5 + (k)) * 8) = st_prev; }
#define PP_TRACE_BYTES 2048
#else
#define PP_TRACE(k)
#define PP_TRACE_BYTES 0
#endif
namespace pp {
constexpr int VW = 16, TH = 16, PW = VW + 2, PH = TH + 2, NPX = PW * PH;   // 18 x 18 = 324 patch pixels
constexpr int NPIECE = 24, PLANE = NPIECE * 1024, AROWB = PW * 64;          // a plane: 384 pixel rows of 64 B (324 real), 24,576 B
constexpr int BN = 256, WTAP = BN * 64;                                     // one tap of one chunk: 256 rows x 64 B = 16 KB
constexpr int NRING = 6, AHEAD = 5;                                         // weight ring (taps) / taps the weight stream runs ahead
constexpr int OFF_PLANE = NRING * WTAP;                                     // 98,304
constexpr size_t LDS_BYTES = OFF_PLANE + 2 * (size_t)PLANE;                 // 147,456 B
// VMEM operations a wave issues in the L segment of tap s of a chunk: 2 weight pieces, + 1 patch piece on taps 0 / 2 / 4
constexpr int ops_of(int s) { return 2 + ((s == 0 || s == 2 || s == 4) ? 1 : 0); }
// operations younger than the pieces of tap s + 1 (issued in L(s - 4)) at the end of L(s): those of L(s - 3) .. L(s)
constexpr int younger(int s) { int n = 0; for (int u = s - 3; u <= s; ++u) n += ops_of(u < 0 ? u + 9 : u); return n; }
static_assert(younger(0) == 9 && younger(2) == 10 && younger(5) == 10 && younger(6) == 9 && younger(8) == 8, "window sums");
}  // namespace pp

template <int N> __device__ __forceinline__ void pp_wait_vm() {   // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14)
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_barrier() {
  asm volatile("" ::: "memory");          // no LDS access of the compiler's moves across (the builtin alone carries no memory semantics)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

template <typename T>
__global__ void __launch_bounds__(512, 2) conv3x3_pp_kernel(ConvKArgs p, int n_mt, int n_nt) {
  using namespace pp;
  using E = Elem<T>;
  constexpr int CHUNK = E::CHUNK, ESZ = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2;                 // 0: waves 0-3, 1: waves 4-7 (the second wave of each SIMD), one tick behind
  const int wm = grp, wn = wave8 & 3;         // pixel rows 8 wm .. 8 wm + 7 of the tile, channels 64 wn .. 64 wn + 63
  const int lr = lane & 15, lq = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int T9 = p.nchunks * 9;               // taps per tile (a multiple of 18: the launcher requires an even chunk count)
  // Tile walk (as conv3x3_ws_kernel): 8 consecutive workgroups (one per XCD) take 8 consecutive pixel tiles, the workgroup 8 further on
  // (same XCD, same L2) the next 256-channel tile of the same pixels.
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt;
  auto decode = [&](int t, int& mt, int& nt) {
    const int per = 8 * n_nt, blk = t / per, r = t - blk * per;
    nt = r >> 3; mt = blk * 8 + (r & 7);
  };
  auto next_valid = [&](int t) {
    for (t += gridDim.x; t < ntp; t += gridDim.x) { int mt, nt; decode(t, mt, nt); if (mt < n_mt) break; }
    return t;
  };
  auto origin = [&](int mt, int& n0, int& y0, int& x0) {
    const int ng = mt / tpi, rem = mt - ng * tpi;
    const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
    n0 = ng; y0 = tyi * TH; x0 = txi * VW;
  };
  int t_first = (int)blockIdx.x - (int)gridDim.x;
  t_first = next_valid(t_first);
  if (t_first >= ntp) return;                 // the whole workgroup leaves together

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.ablate & 1) ? 0u : p.obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);

  // ---------------- DMA streams (ahead of the multiplication, continuous across tiles) ----------------
  // Weights: a tap of a 256-channel tile = the same (chunk, tap) tile of two consecutive 128-row tiles of the packed image
  // ([nt128][chunk][tap][128 rows][64 B], rows already XOR-swizzled: the LDS image), 16 pieces of 1 KB; wave w moves pieces 2w, 2w + 1,
  // so group g moves half g.  Piece j lands at ring slot + 1024 j: row r of the 256 at 64 r.
  auto ws_base = [&](int t) { int mt, nt; decode(t, mt, nt); return (uint32_t)((2 * nt + grp) * T9) * 8192u; };
  uint32_t ws_soff = ws_base(t_first);   // the stream's next tap; moves to the next tile of the walk five taps before the multiplication does
  const uint32_t wvo0 = (uint32_t)(((2 * wave8) & 7) * 1024 + lane * 16), wvo1 = wvo0 + 1024u;
  auto issue_w = [&](auto ringc) {
    constexpr int ring = decltype(ringc)::value;
    char* dst = smem + ring * WTAP + (2 * wave8) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)dst, 16, wvo0, ws_soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(dst + 1024), 16, wvo1, ws_soff, 0, 0);
    ws_soff += 8192u;
  };
  // Patch: piece j = 16 pixel rows of 64 B (pixels 16 j .. 16 j + 15 of the 18 x 18 patch, row-major); lane l lands at slot l & 3 of pixel
  // 16 j + (l >> 2), so the slot swizzle (by the pixel's column) goes into the per-lane SOURCE address; zero padding and the 60 pad pixels
  // are out-of-range offsets (the DMA writes zeros for them: tools/probe/lds_dma_oob_probe.cpp).  Wave w moves pieces w, 8 + w, 16 + w.
  uint32_t pvo0[3], pvo1[3];             // per-lane source offsets of this wave's three pieces, for the tile whose chunks are being streamed
  auto ps_setup = [&](int t) {
    int mt, nt, n0, y0, x0;
    decode(t, mt, nt);
    origin(mt, n0, y0, x0);
    const int cy0 = y0 - 1, cx0 = x0 - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = 16 * (8 * i + wave8) + (lane >> 2);
      const int py = (int)(((float)idx + 0.5f) * (1.0f / (float)PW)), px = idx - py * PW;   // exact: idx < 384
      const int fqx = (lane & 3) ^ ((px >> 1) & 3);
      const int cy = cy0 + py, cx = cx0 + px;
      int sp = -1;
      if (idx < NPX && cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
        if (p.mode == CONV_UP2) sp = (n0 * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1);
        else sp = (n0 * p.Hs + cy) * p.Ws + cx;
      }
      pvo0[i] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C0 * ESZ) + fqx * 16 : p.bytes0;
      pvo1[i] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C1 * ESZ) + fqx * 16 : p.bytes1;
    }
  };
  ps_setup(t_first);
  auto issue_patch = [&](auto ic, auto planec, int sc) {   // piece i of chunk sc (of the streamed tile) -> plane
    constexpr int i = decltype(ic)::value, plane = decltype(planec)::value;
    const int cb = sc * CHUNK;
    const bool first = cb < p.C0;
    const uint32_t so = (uint32_t)((first ? cb : cb - p.C0) * ESZ);
    char* dst = smem + OFF_PLANE + plane * PLANE + (8 * i + wave8) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)dst, 16, first ? pvo0[i] : pvo1[i], so, 0, 0);
  };

  // ---------------- fragment addresses ----------------
  // Every address below is loop-invariant.  Left to itself hipcc hoists all 54 + 24 derived addresses out of the tile loop and spills
  // them (one scratch reload + `s_waitcnt vmcnt(0)` in front of every fragment read: the DMA pipeline drained per tap); an empty asm
  // makes the five bases opaque inside each tap, so the constants fold into the ds_read `offset:` field.
  int a_base[3];                           // per tap column kx (the slot swizzle follows the pixel column); + plane, + (mi + ky) patch rows as immediates
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) a_base[kx] = OFF_PLANE + (wm * 8 * PW + lr + kx) * 64 + 16 * (lq ^ (((lr + kx) >> 1) & 3));
  int b_base0 = (wn * 64 + lr) * 64 + 16 * (lq ^ ((lr >> 1) & 3));   // ring slots 0-2 (+ ni * 1024 + slot * WTAP fit the 16-bit immediate)
  int b_base1 = b_base0 + 3 * WTAP;                                   // ring slots 3-5

  constexpr bool PAIR = E::DTYPE == 1;
  constexpr int NI = 4, NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;
  f32x4 acc[8][NI];
  f32x4 cin[NI];                           // bias + timestep embedding of this lane's channels: the accumulators start from it
  auto cinit_load = [&](int t) {
    int mt, nt, n0, y0, x0;
    decode(t, mt, nt);
    origin(mt, n0, y0, x0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int co = nt * BN + wn * 64 + ni * 16 + 4 * lq;   // Cout % 256 == 0: always in range
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co);
      if (p.emb) { const f32x4 e = *reinterpret_cast<const f32x4*>(p.emb + (size_t)n0 * p.emb_stride + co); v = f32x4{v[0] + e[0], v[1] + e[1], v[2] + e[2], v[3] + e[3]}; }
      cin[ni] = v;
    }
  };
  auto acc_init = [&]() {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = cin[ni];
  };

  const int abl = p.ablate;   // timing experiments (results become wrong): 4 no MFMA, 8 no weight DMA, 16 no patch DMA, 128 no fragment reads; 64 drain instead of counting
  STAMP_DECL
  CLK_DECL
  // ---------------- pipeline fill: chunk 0 of the first tile -> plane 0, weight taps 0 .. 4 -> ring slots 0 .. 4 ----------------
  cinit_load(t_first);
  issue_patch(IC<0>(), IC<0>(), 0); issue_patch(IC<1>(), IC<0>(), 0); issue_patch(IC<2>(), IC<0>(), 0);
  issue_w(IC<0>()); issue_w(IC<1>()); issue_w(IC<2>()); issue_w(IC<3>()); issue_w(IC<4>());
  acc_init();
  pp_wait_vm<8>();                         // this wave's pieces of the patch and of tap 0 have landed (taps 1 .. 4 may fly)
  pp_barrier();

  // E's vector-memory operations sit between the weight pieces issued before it and the waits of the next tile's first four taps:
  // stores (+ residual loads) counted exactly, the GroupNorm store and the next tile's bias / emb loads not at all (an over-wait by
  // operations issued a whole epilogue earlier); fp32 would overflow the 6-bit counter and drains instead.
  constexpr int EPI_STORES = 8 * NP2;
  const int extra0 = PAIR ? EPI_STORES * (p.res_mode != RES_NONE ? 2 : 1) : 0;
  int extra = 0;

  for (int t = t_first; t < ntp;) {
    const int t_next = next_valid(t);
    const int t_nextc = t_next < ntp ? t_next : t;   // the streams' next tile (clamped at the end of the walk)
    int mt, nt, n0, y0, x0;
    decode(t, mt, nt);
    origin(mt, n0, y0, x0);
    if (grp == 1) pp_barrier();            // one tick behind group 0
    STAMP(7)
    // one tap: P = chunk parity (patch plane), S = tap of the chunk
    // c: the chunk being multiplied (runtime), P = its parity (patch plane), S = tap of the chunk; lastp: the chunk pair is the tile's last
    auto tap = [&](auto Pc, auto Sc, int c, bool lastp) {
      constexpr int P = decltype(Pc)::value, S = decltype(Sc)::value;
      constexpr int ky = S / 3, kx = S % 3, ring = (P * 9 + S) % NRING, ringn = (ring + AHEAD) % NRING;
      const bool last_of_tile = lastp && P == 1 && S == 8;
      // ---- L segment: this tap's fragments; this wave's share of the DMA ----
      u32x4 af[8], bf[NI];
      {
        int ab = a_base[kx], bb = ring < 3 ? b_base0 : b_base1;
        asm volatile("" : "+v"(ab), "+v"(bb));
        const char* ap = smem + ab;
        const char* bp = smem + bb;
        if (!(abl & 128)) {
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(ap + P * PLANE + (mi + ky) * AROWB);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const u32x4*>(bp + (ring % 3) * WTAP + ni * 1024);
        } else {   // timing experiment: MFMAs on whatever the registers hold
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) asm volatile("" : "=v"(af[mi]));
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) asm volatile("" : "=v"(bf[ni]));
        }
      }
      // the streams: the patch one chunk ahead (the last chunk of a tile fetches chunk 0 of the walk's next tile: its geometry is switched
      // in once, in front of that chunk), the weights five taps ahead (tap S = 4 of a tile's last chunk is the first to fetch the next
      // tile's weights).  Past the end of the walk both re-fetch the last tile's addresses into buffers nobody reads.
      if constexpr (P == 1 && S == 0) { if (lastp) ps_setup(t_nextc); }
      if constexpr (P == 1 && S == 4) { if (lastp) ws_soff = ws_base(t_nextc); }
      if constexpr (S == 0 || S == 2 || S == 4) { if (!(abl & 16)) issue_patch(IC<S / 2>(), IC<1 - P>(), P == 0 ? c + 1 : (lastp ? 0 : c + 1)); }
      if (!(abl & 8)) issue_w(IC<ringn>());
#ifdef CONV_STAMPS
      const int tr_tap = c * 9 + S;
      const bool tr_on = blockIdx.x == 0 && t == t_first && tr_tap >= 8 && tr_tap < 12 && lane == 0;
#endif
      STAMP(0) PP_TRACE(0)
      if constexpr (P == 0 && S < 4 && PAIR) {     // the first taps of a tile: the previous tile's epilogue is inside the window
        if (extra == 0) pp_wait_vm<younger(S)>();
        else if (extra == EPI_STORES) pp_wait_vm<younger(S) + EPI_STORES>();
        else pp_wait_vm<younger(S) + 2 * EPI_STORES>();
      } else {
        pp_wait_vm<younger(S)>();
      }
      STAMP(1) PP_TRACE(1)
      pp_barrier();
      STAMP(2) PP_TRACE(2)
      // ---- M segment ----
      __builtin_amdgcn_s_setprio(1);
      if (!(abl & 4)) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], bf[ni], af[mi], T());   // D rows = channels, cols = pixels
      } else {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) asm volatile("" :: "v"(af[mi]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" :: "v"(bf[ni]));
      }
      __builtin_amdgcn_s_setprio(0);
      STAMP(3) PP_TRACE(3)
      if (!(last_of_tile && grp == 1)) pp_barrier();
      STAMP(4) PP_TRACE(4)
    };
    for (int c = 0; c < p.nchunks; c += 2) {
      const bool lastp = c + 2 >= p.nchunks;
      tap(IC<0>(), IC<0>(), c, lastp); tap(IC<0>(), IC<1>(), c, lastp); tap(IC<0>(), IC<2>(), c, lastp);
      tap(IC<0>(), IC<3>(), c, lastp); tap(IC<0>(), IC<4>(), c, lastp); tap(IC<0>(), IC<5>(), c, lastp);
      tap(IC<0>(), IC<6>(), c, lastp); tap(IC<0>(), IC<7>(), c, lastp); tap(IC<0>(), IC<8>(), c, lastp);
      tap(IC<1>(), IC<0>(), c + 1, lastp); tap(IC<1>(), IC<1>(), c + 1, lastp); tap(IC<1>(), IC<2>(), c + 1, lastp);
      tap(IC<1>(), IC<3>(), c + 1, lastp); tap(IC<1>(), IC<4>(), c + 1, lastp); tap(IC<1>(), IC<5>(), c + 1, lastp);
      tap(IC<1>(), IC<6>(), c + 1, lastp); tap(IC<1>(), IC<7>(), c + 1, lastp); tap(IC<1>(), IC<8>(), c + 1, lastp);
    }

    // ---------------- epilogue (as conv3x3_ws_kernel: MFMA rows are channels, columns are pixels; lane (lr, lq) holds 4 consecutive
    // channels of pixel lr per 16 x 16 tile; bf16 pairs of channel tiles merge into 16-byte stores by two v_permlane16_swap) ----------------
    cinit_load(t_nextc);                     // the next tile's start values travel while this tile is stored
    const int co_w = nt * BN + wn * 64 + 4 * lq;
    const int co_s = PAIR ? nt * BN + wn * 64 + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
    GnPartial<NI> gp;
    const bool do_gn = p.gn_stats != nullptr;
    const bool gn_mask = ((p.Wo | p.Ho) & 15) != 0;
    auto epi_half = [&](auto hc, auto resc, auto gnc) {
      constexpr int h = decltype(hc)::value, GNM = decltype(gnc)::value;
      constexpr bool HAS_RES = decltype(resc)::value != 0;
      uint32_t ovo[4], rvo[4];
      float vm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = y0 + wm * 8 + h * 4 + j, x = x0 + lr;
        const bool ok = y < p.Ho && x < p.Wo;
        vm[j] = ok ? 1.f : 0.f;
        const uint32_t opix = (uint32_t)((n0 * p.Ho + y) * p.Wo + x);
        ovo[j] = ok ? (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.obytes;
        uint32_t rpix = opix;
        if (p.res_mode == RES_UP2) rpix = (uint32_t)((n0 * p.Hr + (y >> 1)) * p.Wr + (x >> 1));
        rvo[j] = (ok && HAS_RES) ? (rpix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.rbytes;
      }
      u32x4 rr[4][NP2];
      if constexpr (HAS_RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < NP2; ++k)
            rr[j][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, rvo[j] + k * PSTEP * ESZ, 0, 0));
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
            if constexpr (GNM != 0) gp.add(ni, o[0], o[1], o[2], o[3], GNM == 2, vm[j]);
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
                ra[2 * q] = __builtin_bit_cast(float, xa[q] << 16); ra[2 * q + 1] = __builtin_bit_cast(float, xa[q] & 0xffff0000u);
                rb[2 * q] = __builtin_bit_cast(float, xb[q] << 16); rb[2 * q + 1] = __builtin_bit_cast(float, xb[q] & 0xffff0000u);
              }
            }
            bf16x4 ta, tb;
            float va[4], vb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              va[q] = HAS_RES ? acc[mi][2 * k][q] + ra[q] : acc[mi][2 * k][q];
              vb[q] = HAS_RES ? acc[mi][2 * k + 1][q] + rb[q] : acc[mi][2 * k + 1][q];
              ta[q] = (bf16)va[q];
              tb[q] = (bf16)vb[q];
            }
            if constexpr (GNM != 0) {
              gp.add(2 * k, va[0], va[1], va[2], va[3], GNM == 2, vm[j]);
              gp.add(2 * k + 1, vb[0], vb[1], vb[2], vb[3], GNM == 2, vm[j]);
            }
            const u32x2 pa2 = __builtin_bit_cast(u32x2, ta), pb2 = __builtin_bit_cast(u32x2, tb);
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo[j] + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
    };
    auto epi = [&](auto resc, auto gnc) { epi_half(IC<0>(), resc, gnc); epi_half(IC<1>(), resc, gnc); };
    if (p.res_mode != RES_NONE) {
      if (!do_gn) epi(IC<1>(), IC<0>()); else if (!gn_mask) epi(IC<1>(), IC<1>()); else epi(IC<1>(), IC<2>());
    } else {
      if (!do_gn) epi(IC<0>(), IC<0>()); else if (!gn_mask) epi(IC<0>(), IC<1>()); else epi(IC<0>(), IC<2>());
    }
    if (do_gn) {   // slot = (pixel tile of the image, 8-row half); quads of this wave's 64 channels
      const int rem = mt - n0 * tpi;
      gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * 2 + wm) * (size_t)(p.Cout >> 2) + ((nt * BN + wn * 64) >> 2)) * 2, lq, lr);
    }
    acc_init();
    if constexpr (!PAIR) pp_wait_vm<0>();
    else if (p.ablate & 64) pp_wait_vm<0>();   // diagnostic: drain instead of counting the epilogue's operations (results must not change)
    extra = (!PAIR || (p.ablate & 64)) ? 0 : extra0;
    STAMP(5)
    pp_barrier();                          // both groups have stored their tile: the next tile starts with group 0's L(0)
    STAMP(6)
    t = t_next;
  }
  pp_wait_vm<0>();                         // no DMA piece may still be in flight towards LDS when the workgroup retires
  STAMP_FLUSH
  CLK_FLUSH
#ifdef CONV_STAMPS
  if (p.dbg && blockIdx.x == 0 && lane < 20) p.dbg[2048 * 8 + wave8 * 20 + lane] = *reinterpret_cast<volatile unsigned long long*>(smem + pp::LDS_BYTES + (wave8 * 20 + lane) * 8);
#endif
}

// Shapes the ping-pong kernel takes: 3x3 / stride 1 / NHWC output without input prologue, 256-channel output tiles over the 128-row packed
// weight tiles, an even number of 64-byte channel chunks per source switch (two chunks are unrolled; a source boundary may fall anywhere),
// images of at least one 16 x 16 tile.  mode 1: only when every CU gets a tile; mode 2: always (tests).
static bool pp_eligible(int mode, int ks, int G, int bn_pack, int out_mode, int stride, int nchunks, bool has_pro, int N, int Ho, int Wo, int Cout) {
  if (!mode || ks != 3 || G != 1 || bn_pack != 128 || out_mode != OUT_NHWC || stride != 1 || has_pro) return false;
  if (Cout % pp::BN != 0 || nchunks < 2 || (nchunks & 1)) return false;
  if (Wo < pp::VW || Ho < pp::TH) return false;
  if (mode >= 2) return true;
  const int n_mt = N * ((Wo + pp::VW - 1) / pp::VW) * ((Ho + pp::TH - 1) / pp::TH), n_nt = Cout / pp::BN;
  return n_mt * n_nt >= ws_num_cus();
}

// 0 = launched, 1 = not eligible, < 0 = error
template <typename T>
int launch_pp(ConvKArgs a, int mode, int ks, hipStream_t s) {
  if (!pp_eligible(mode, ks, a.G, a.bn_pack, a.out_mode, a.stride, a.nchunks, a.pro_a != nullptr, a.N, a.Ho, a.Wo, a.Cout)) return 1;
  a.lvw = 4; a.lth = 4; a.PW = pp::PW; a.PH = pp::PH; a.NP = pp::NPX;
  a.tiles_x = (a.Wo + pp::VW - 1) / pp::VW; a.tiles_y = (a.Ho + pp::TH - 1) / pp::TH;
  const int n_mt = a.N * a.tiles_x * a.tiles_y, n_nt = a.Cout / pp::BN;
  if (int rc = mi355_allow_big_lds(conv3x3_pp_kernel<T>, "conv3x3 (ping-pong)")) return rc;
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt, ncu = ws_num_cus();
  const int grid = ntp < ncu ? ntp : ncu;   // one persistent workgroup per CU
  hipLaunchKernelGGL(conv3x3_pp_kernel<T>, dim3(grid), dim3(512), pp::LDS_BYTES + PP_TRACE_BYTES, s, a, n_mt, n_nt);
  return 0;
}
