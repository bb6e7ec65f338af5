// Ping-pong 3x3 implicit GEMM (included by conv_igemm.hip; same ConvKArgs, same packed weights, same NHWC tensors).
//
// Why a second large-level kernel (round 4).  conv3x3_ws_kernel splits a CU into four DMA / prologue waves and four MFMA waves; three rounds
// of stamps and ablations ended at: the MFMA waves alone, operands resident in LDS, take 78 % of the kernel; they sit
// at the 256-register limit, nothing hides their LDS latency, hand-over polls or epilogue, and a loader wave costs the SIMD as many
// registers as an MFMA wave.  This kernel is the structure of cdna_hip_programming.md's 256 x 256 8-phase GEMM instead:
//   * ALL eight waves multiply (wave tile 128 pixels x 64 channels = 8 x 4 MFMA tiles); two tile geometries (template CFG, namespace pp):
//       CFG 0 "wide"  : 16 x 16 pixels x 256 output channels, waves 2 (8-row halves) x 4 (64 channels); the patch of a 64-byte channel
//                       chunk (18 x 18 pixels) is staged once per 256 output channels;
//       CFG 1 "narrow": 16 rows x 32 columns x 128 output channels, waves 4 (4 rows x 32 columns) x 2 (round 5): every Cout % 128 == 0
//                       layer on images at least 32 wide - the ten 32x32-level convs of the CIFAR net; halo 34 x 18 / 512 = 1.20;
//   * the waves form two groups (waves 0-3 / 4-7 = the two waves of every SIMD) that run half a step apart: while one group issues its
//     LDS fragment reads and its share of the DMA (an "L" segment), the other runs the 32 MFMAs of a tap (an "M" segment), then they
//     swap - the matrix pipe of a SIMD always has one wave feeding it, and a wave's own reads never sit in front of its own MFMAs, so
//     one fragment register set is enough;
//   * operands go global -> LDS by DMA (`buffer_load ... lds`) issued by the MFMA waves themselves: per tap every wave moves its share of
//     the tap's weight pieces (1 KB each) five taps ahead into a 6-tap ring, and the patch pieces of the NEXT chunk into the other patch
//     plane (a plane is padded to 8 x NPW whole pieces so that every wave issues the same number: the `s_waitcnt vmcnt(N)` that
//     publishes a tap is an immediate).  Nothing is ever drained: the wait at the end of L(s) leaves the pieces of taps s+2 .. s+5 in flight;
//   * ordering is the guide's rule: the issuing wave's counted vmcnt, then a workgroup barrier, then the read one tick later (RAW); a
//     buffer is re-filled only behind the barrier that follows its last reader's segment (WAR: the readers' ds_reads are issued before that
//     barrier and the DMA that overwrites the slot is issued after it, so the order rests on an LDS read returning (~100 cycles) before a DMA
//     issued later lands (> 500 cycles through L2): a latency argument, not a counter - the PP_ABLATE builds without MFMAs shorten the
//     distance and are timing experiments only);
//   * persistent: a workgroup walks tiles like the warp-specialised kernel; the DMA stream is continuous across tile boundaries (the
//     last taps of a tile already fetch the next tile's first weights and patch chunk), the epilogues of the two groups overlap each
//     other and the second group's last MFMA segment; past the end of the walk the stream re-fetches valid addresses into free
//     buffers (never read), so the waits' immediates hold everywhere.
// PRO = 2 (round 5): the GroupNorm affine + SiLU (+ FiLM, folded into (a, b)) of the conv's INPUT (AD/image_diffusion/unet.py:281-285,
// 305-310, 343-347; the skip concat of :725 is the two-source chunk stream) applied IN LDS, IN PLACE, by the MFMA waves themselves: all
// pieces of the next chunk are issued in L(0) behind two 4-byte-per-lane DMAs of that chunk's (a, b) rows into a private 512-byte buffer
// of the wave; in the L segments of later taps (pp::S::trn) a wave reads back one of ITS OWN pieces (its own counted vmcnt is all the
// ordering that needs - no barrier), applies silu(a x + b) and writes it back where the zero padding allows (a padded pixel keeps the zeros
// the DMA wrote for its out-of-range source offset); the barrier at the end of L(8) publishes the plane as before.  The lane <-> 16-byte
// slot assignment of this pass undoes the plane's XOR swizzle, so a lane works on the same 8 (bf16) / 4 (fp32) channels in every piece.
// Measured against the wide form on the same box (profiles/r4_experiments.md): a variant whose step is a tap COLUMN (three taps of one kx: 10
// patch-row fragments instead of 24, one barrier pair per 96 MFMAs; tools/experiments/r4_conv_pp_column_steps_variant.inc.h.txt) reaches 79 %
// MFMA issue density instead of 59 % - and the same wall time: the in-kernel clock falls from 2.28 to 1.8 GHz.
// Tick diagram (T = taps per tile, one raw barrier between ticks):
//   group 0:  L0 | M0 | L1 | M1 | ... | M(T-1) | E ............ | L0' | M0' ...
//   group 1:  -- | L0 | M0 | L1 | ... | L(T-1) | M(T-1) E ..... | --  | L0' ...
// Diagnostic build only (-DCONV_STAMPS): absolute stamp times of taps 8 .. 11 of workgroup 0's first tile, kept in LDS (a global store would
// count in vmcnt and move the counted waits), copied behind the stamp sums at the end: the timeline of the two groups.
#ifndef PP_ABLATE
#define PP_ABLATE 0
#endif
#ifndef PP_TRACE_TAP0
#define PP_TRACE_TAP0 8   // first of the four traced taps of workgroup 0's first tile
#endif
#ifdef CONV_STAMPS
#define PP_TRACE(k) if (tr_on) { const uint32_t ta = (uint32_t)(LDS_END + ((wave8 * 4 + (tr_tap - PP_TRACE_TAP0)) * 5 + (k)) * 4); uint32_t tv = (uint32_t)st_prev; asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(tr_a), "=&v"(tr_v) : "s"(ta), "s"(tv) : "memory"); }
#define PP_TRACE_BYTES 2048
#else
#define PP_TRACE(k)
#define PP_TRACE_BYTES 0
#endif
namespace pp {
constexpr int NRING = 6, AHEAD = 5;           // weight ring (taps) / taps the weight stream runs ahead
constexpr int ABUF = 512;                     // PRO: a wave's private (a, b) rows of one chunk: a at +0, b at +256 (fp32, <= 32 channels each)
template <int CFG> struct G;
template <> struct G<0> {                     // wide: 16 x 16 pixels x 256 channels
  static constexpr int VW = 16, TH = 16, BN = 256, WM = 2, WROWS = 8, NWP = 2, NPW = 3;
  static constexpr int row(int mi) { return mi; }
  static constexpr int col(int mi) { (void)mi; return 0; }
};
template <> struct G<1> {                     // narrow: 16 rows x 32 columns x 128 channels
  static constexpr int VW = 32, TH = 16, BN = 128, WM = 4, WROWS = 4, NWP = 1, NPW = 5;
  static constexpr int row(int mi) { return mi >> 1; }
  static constexpr int col(int mi) { return 16 * (mi & 1); }
};
// derived sizes.  WM: pixel waves (x 8 / WM channel waves); WROWS: tile rows per pixel wave; NWP: weight pieces per wave and tap;
// NPW: patch pieces per wave and chunk (a plane = 8 NPW pieces of 16 pixel rows of 64 B, the real PW x PH pixels first)
template <int CFG> struct D : G<CFG> {
  using G<CFG>::VW; using G<CFG>::TH; using G<CFG>::BN; using G<CFG>::NPW;
  static constexpr int PW = VW + 2, PH = TH + 2, NPX = PW * PH, NPIECE = 8 * NPW, PLANE = NPIECE * 1024, AROWB = PW * 64;
  static constexpr int WTAP = BN * 64;                           // one tap of one chunk: BN rows x 64 B
  static constexpr int OFF_PLANE = NRING * WTAP, OFF_AB = OFF_PLANE + 2 * PLANE;
  static constexpr int XBUF = 1024;                              // ACT: the two 8-row halves of a tile exchange their group sums: [8 waves][16 quads] x (sum, sum of squares)
  static constexpr size_t lds_bytes(int pro, int act = 0) { return (size_t)OFF_AB + (pro ? 8 * ABUF : 0) + (act ? XBUF : 0); }
  static_assert(NPX <= NPIECE * 16, "plane too small");
};
// Issue / wait schedule of a chunk (9 taps).  Operations a wave issues in L(s), in this order: [PRO, s = 0: the two (a, b) DMAs] [patch
// pieces of the next chunk whose issue tap is s] [NWP weight pieces of tap s + 5].
template <int CFG, int PRO> struct S {
  static constexpr int NWP = G<CFG>::NWP, NPW = G<CFG>::NPW, NAB = PRO ? 2 : 0;
  static constexpr int iss(int i) { return PRO ? 0 : (CFG == 0 ? 2 * i : i); }                       // tap whose L segment issues patch piece i
  // PRO: piece i is transformed in the L segment of tap trn(i): wide in taps 3 / 5 / 7, narrow in taps 3 .. 7 (tap 1 would wait for pieces issued one L
  // segment earlier; tap 8's barrier publishes the plane).  Same-box A/Bs (profiles/r5_experiments.md; the kernel with those switches is kept as
  // tools/experiments/r5_conv_pp_with_experiment_switches.inc.h.txt): the prologue costs its VALU issue time wherever it runs - whole pieces, half
  // pieces over twice the taps, operands read early or late, between the wave's own MFMAs of the M segment, either wave priority, the exponent as a
  // second multiply-add from prescaled tables: 256 -> 256 at 16x16 68-72 us against 62 without the arithmetic and 77 on the warp-specialised kernel.
  static constexpr int trn(int i) { return CFG == 0 ? 3 + 2 * i : 3 + i; }
  static constexpr int piece_of(int s) { for (int i = 0; i < NPW; ++i) if (trn(i) == s) return i; return -1; }   // the piece tap s transforms, -1 = none
  static constexpr int npatch(int s) { int n = 0; for (int i = 0; i < NPW; ++i) n += iss(i) == s ? 1 : 0; return n; }
  static constexpr int ops_of(int s) { return (s == 0 ? NAB : 0) + npatch(s) + NWP; }
  // operations younger than the weight pieces of tap s + 1 (issued last in L(s - 4)) at the end of L(s): those of L(s - 3) .. L(s)
  static constexpr int yw(int s) { int n = 0; for (int u = s - 3; u <= s; ++u) n += ops_of(u < 0 ? u + 9 : u); return n; }
  // PRO: operations younger than the piece that L(s + 1) transforms (all pieces are issued in L(0) of this chunk, in index order); 99 = none is due
  static constexpr int yp(int s) {
    if (!PRO || piece_of(s + 1) < 0) return 99;
    int n = (NPW - 1 - piece_of(s + 1)) + NWP;
    for (int v = 1; v <= s; ++v) n += ops_of(v);
    return n;
  }
  static constexpr int younger(int s) { return yw(s) < yp(s) ? yw(s) : yp(s); }
  static constexpr bool trn_ok() { for (int i = 0; i < NPW; ++i) if (trn(i) < 2 || trn(i) > 7) return false; return true; }
  static_assert(!PRO || trn_ok(), "a piece is transformed in L(2) .. L(7): after its own wait, before the wave's last lgkmcnt(0) in front of the publishing barrier");
};
static_assert(S<0, 0>::younger(0) == 9 && S<0, 0>::younger(2) == 10 && S<0, 0>::younger(5) == 10 && S<0, 0>::younger(6) == 9 && S<0, 0>::younger(8) == 8, "window sums of the round-4 kernel");
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

// silu(a x + b) of NV 32-bit words of a fragment (bf16: 2 NV channels, fp32: NV channels): the same operation sequence as the other conv kernels'
// prologues, written stage by stage (the transcendental results are used a stage later: no dependent back-to-back issue)
template <int NV, typename T>
__device__ __forceinline__ void pp_pro_words(const uint32_t (&raw)[NV], const float* a, const float* b, uint32_t (&out)[NV], T) {   // T = bf16 / f16
  float x[2 * NV], v[2 * NV], e[2 * NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { unpack2(raw[i], x[2 * i], x[2 * i + 1], T()); }
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) v[j] = a[j] * x[j] + b[j];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) e[j] = -1.4426950408889634f * v[j];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) e[j] = __builtin_amdgcn_exp2f(e[j]);
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) e[j] = 1.0f + e[j];
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
#pragma unroll
  for (int j = 0; j < 2 * NV; ++j) v[j] = v[j] * e[j];
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = pack2(v[2 * i], v[2 * i + 1], T());
}
template <int NV>
__device__ __forceinline__ void pp_pro_words(const uint32_t (&raw)[NV], const float* a, const float* b, uint32_t (&out)[NV], float) {
#pragma unroll
  for (int j = 0; j < NV; ++j) { const float v = a[j] * __builtin_bit_cast(float, raw[j]) + b[j]; out[j] = __builtin_bit_cast(uint32_t, v / (1.0f + expf(-v))); }
}

// ACT = 1 (wide only): the GroupNorm (+ SiLU, FiLM) site that READS this conv's output is applied in the epilogue, in place (launcher: one 16 x 16 tile
// per image, no residual, 8 or 16 channels per group) - as conv3x3_ws_kernel<T, PRO, 1> did; a template parameter so that the other forms keep their registers.
template <typename T, int CFG, int PRO, int ACT = 0>
__global__ void __launch_bounds__(512, 2) conv3x3_pp_kernel(ConvKArgs p, int n_mt, int n_nt) {
  using Dg = pp::D<CFG>;
  using Sc = pp::S<CFG, PRO>;
  using E = Elem<T>;
  constexpr int NRING = pp::NRING, AHEAD = pp::AHEAD;
  constexpr int VW = Dg::VW, TH = Dg::TH, PW = Dg::PW, NPX = Dg::NPX, PLANE = Dg::PLANE, AROWB = Dg::AROWB, BN = Dg::BN, WTAP = Dg::WTAP;
  constexpr int OFF_PLANE = Dg::OFF_PLANE, OFF_AB = Dg::OFF_AB, NPW = Dg::NPW, NWP = Dg::NWP, WMN = Dg::WM, WROWS = Dg::WROWS;
  constexpr int CHUNK = E::CHUNK, ESZ = sizeof(T);
  constexpr int LDS_END = (int)Dg::lds_bytes(PRO, ACT), OFF_X = (int)Dg::lds_bytes(PRO);
  static_assert(ACT == 0 || CFG == 0, "the fused output GroupNorm needs whole images per tile: wide form at 16 x 16");
  (void)LDS_END; (void)NPX; (void)OFF_AB; (void)OFF_X;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2;                 // 0: waves 0-3, 1: waves 4-7 (the second wave of each SIMD), one tick behind
  // wide: pixel rows 8 wm .. + 7, channels 64 wn .. + 63 (wm = group); narrow: pixel rows 4 wm .. + 3 (all 32 columns), channels 64 wn .. + 63
  const int wm = CFG == 0 ? grp : wave8 >> 1, wn = CFG == 0 ? wave8 & 3 : wave8 & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;
  const int T9 = p.nchunks * 9;               // taps per tile (a multiple of 18: the launcher requires an even chunk count)
  // Tile walk (as conv3x3_ws_kernel): 8 consecutive workgroups (one per XCD) take 8 consecutive pixel tiles, the workgroup 8 further on
  // (same XCD, same L2) the next channel tile of the same pixels.
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
  const uint32_t abytes = PRO ? (uint32_t)p.N * (uint32_t)p.Cin * 4u : 0u;   // the (a, b) tables: [N][Cin] fp32
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO ? p.pro_a : p.bias), 0, abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRO ? p.pro_b : p.bias), 0, abytes, 0x00020000);

  // ---------------- DMA streams (ahead of the multiplication, continuous across tiles) ----------------
  // Weights: packed image [nt128][chunk][tap][128 rows][64 B], rows already XOR-swizzled: the LDS image.  Wide: a tap of a 256-channel
  // tile = the same (chunk, tap) tile of two consecutive 128-row tiles, 16 pieces of 1 KB; wave w moves pieces 2w, 2w + 1, so group g
  // moves half g.  Narrow: one 128-row tile, 8 pieces, wave w moves piece w.  Piece j lands at ring slot + 1024 j: row r at 64 r.
  auto ws_base = [&](int t) { int mt, nt; decode(t, mt, nt); return (uint32_t)((CFG == 0 ? 2 * nt + grp : nt) * T9) * 8192u; };
  uint32_t ws_soff = ws_base(t_first);   // the stream's next tap; moves to the next tile of the walk five taps before the multiplication does
  const uint32_t wvo0 = (uint32_t)((CFG == 0 ? ((2 * wave8) & 7) : wave8) * 1024 + lane * 16), wvo1 = wvo0 + 1024u;
  auto issue_w = [&](auto ringc) {
    constexpr int ring = decltype(ringc)::value;
    char* dst = smem + ring * WTAP + (NWP * wave8) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)dst, 16, wvo0, ws_soff, 0, 0);
    if constexpr (NWP == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(dst + 1024), 16, wvo1, ws_soff, 0, 0);
    ws_soff += 8192u;
  };
  // Patch: piece j = 16 pixel rows of 64 B (pixels 16 j .. 16 j + 15 of the PW x PH patch, row-major); lane l lands at slot l & 3 of pixel
  // 16 j + (l >> 2), so the slot swizzle (by the pixel's column) goes into the per-lane SOURCE address; zero padding and the pad pixels
  // are out-of-range offsets (the DMA writes zeros for them: tools/probe/lds_dma_oob_probe.cpp).  Wave w moves pieces w, 8 + w, 16 + w, ...
  // Wide without prologue (the round-4 kernel, unchanged): the per-lane byte offsets of both sources are kept (6 registers).  Every other form keeps ONE
  // key per piece, source pixel * 4 + swizzled slot (-1: zero padding / pad pixel), and forms the byte offset when the piece is issued
  // (a multiply-add per DMA; the narrow form's ten offsets did not fit the register budget: two lived in scratch and every L(0) reloaded
  // them behind `s_waitcnt vmcnt(0)`).  The key's sign is also the prologue's "inside the image" bit.
  constexpr bool KEYED = !(CFG == 0 && PRO == 0);
  uint32_t pvo0[6], pvo1[6];             // per-lane source offsets of this wave's pieces (NPW <= 6 used), for the tile whose chunks are being streamed
  int pkey[6];
  int ps_n0 = 0;                         // PRO: image of the streamed tile (row of the (a, b) tables)
  // (t and the lane id are made opaque: everything here is invariant in the loops that call it, and hipcc would otherwise compute the next
  //  tile's offsets at the top of every tile and the per-lane (py, px) at the top of the kernel, keep them alive through the tap loop, spill
  //  them and reload them behind `s_waitcnt vmcnt(0)` - the DMA pipeline drained once per chunk pair)
  auto ps_setup = [&](int t) {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    asm volatile("" : "+s"(t));
    int mt, nt, n0, y0, x0;
    decode(t, mt, nt);
    origin(mt, n0, y0, x0);
    const int cy0 = y0 - 1, cx0 = x0 - 1;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = 16 * (8 * i + wave8) + (ln >> 2);
      const int py = (int)(((float)idx + 0.5f) * (1.0f / (float)PW)), px = idx - py * PW;   // exact: idx < 1024
      const int fqx = (ln & 3) ^ ((px >> 1) & 3);
      const int cy = cy0 + py, cx = cx0 + px;
      int sp = -1;
      if (idx < NPX && cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
        if (p.mode == CONV_UP2) sp = (n0 * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1);
        else sp = (n0 * p.Hs + cy) * p.Ws + cx;
      }
      if constexpr (KEYED) {
        pkey[i] = sp >= 0 ? sp * 4 + fqx : -1;
      } else {
        pvo0[i] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C0 * ESZ) + fqx * 16 : p.bytes0;
        pvo1[i] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C1 * ESZ) + fqx * 16 : p.bytes1;
      }
    }
    if constexpr (PRO != 0) ps_n0 = n0;
  };
  ps_setup(t_first);
  auto issue_patch = [&](auto ic, auto planec, int sc) {   // piece i of chunk sc (of the streamed tile) -> plane
    constexpr int i = decltype(ic)::value, plane = decltype(planec)::value;
    const int cb = src_chunk(p, sc) * CHUNK;
    const bool first = cb < p.C0;
    const uint32_t so = (uint32_t)((first ? cb : cb - p.C0) * ESZ);
    char* dst = smem + OFF_PLANE + plane * PLANE + (8 * i + wave8) * 1024;
    uint32_t vo;
    if constexpr (KEYED) {
      const int k = pkey[i];
      const uint32_t rowb = (uint32_t)((first ? p.C0 : p.C1) * ESZ);
      vo = k >= 0 ? (uint32_t)(k >> 2) * rowb + (uint32_t)(k & 3) * 16u : (first ? p.bytes0 : p.bytes1);
    } else {
      vo = first ? pvo0[i] : pvo1[i];
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
  };
  auto issue_patch_at = [&](auto sc_tap, auto planec, int sc) {   // the pieces whose issue tap is S
    constexpr int S = decltype(sc_tap)::value;
    if constexpr (NPW > 0 && Sc::iss(0) == S) issue_patch(IC<0>(), planec, sc);
    if constexpr (NPW > 1 && Sc::iss(1 < NPW ? 1 : 0) == S) issue_patch(IC<1>(), planec, sc);
    if constexpr (NPW > 2 && Sc::iss(2 < NPW ? 2 : 0) == S) issue_patch(IC<2>(), planec, sc);
    if constexpr (NPW > 3 && Sc::iss(3 < NPW ? 3 : 0) == S) issue_patch(IC<3>(), planec, sc);
    if constexpr (NPW > 4 && Sc::iss(4 < NPW ? 4 : 0) == S) issue_patch(IC<4>(), planec, sc);
    if constexpr (NPW > 5 && Sc::iss(5 < NPW ? 5 : 0) == S) issue_patch(IC<5>(), planec, sc);
  };
  // PRO: the (a, b) rows of chunk sc of the streamed tile's image -> this wave's private buffer, 4 bytes per lane (lanes >= CHUNK read out of range: zeros)
  const uint32_t abvo = PRO ? ((uint32_t)lane < (uint32_t)CHUNK ? (uint32_t)lane * 4u : abytes) : 0u;
  auto issue_ab = [&](int sc) {
    if constexpr (PRO != 0) {
      const uint32_t so = (uint32_t)(ps_n0 * p.Cin + src_chunk(p, sc) * CHUNK) * 4u;
      char* dst = smem + OFF_AB + wave8 * pp::ABUF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (__attribute__((address_space(3))) void*)dst, 4, abvo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(dst + 256), 4, abvo, so, 0, 0);
    }
  };
  // PRO: piece i of this wave in `plane`: silu(a x + b) in place.  Lane l works on pixel 16 (8 i + wave) + (l >> 2), LOGICAL slot l & 3; the physical
  // slot of the lane's logical slot is the low two bits of the piece's key (ps_setup: key = pixel * 4 + ((lane & 3) ^ swizzle)), so a lane's channels are
  // the same in every piece; a lane whose pixel is padding (key < 0) stores nothing.  ds_read_b128 / ds_write_b128 of whole pixels (64 B) per lane quad:
  // conflict-free.  The reads (tr_load) are issued in FRONT of the tap's fragment reads (LDS returns in order: the arithmetic starts while the fragments travel).
  struct TrOps { uint32_t raw[4]; float a[8], b[8]; int off; };
  auto tr_load = [&](auto ic, auto planec, TrOps& o) {
    constexpr int i = decltype(ic)::value, plane = decltype(planec)::value;
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    o.off = OFF_PLANE + plane * PLANE + (8 * i + wave8) * 1024 + ((ln >> 2) << 6) + ((pkey[i] & 3) << 4);
    const char* abp = smem + OFF_AB + wave8 * pp::ABUF + (ln & 3) * (E::VEC * 4);
    const u32x4 r = *reinterpret_cast<const u32x4*>(smem + o.off);
    o.raw[0] = r[0]; o.raw[1] = r[1]; o.raw[2] = r[2]; o.raw[3] = r[3];
#pragma unroll
    for (int k = 0; k < E::VEC; k += 4) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(abp + k * 4), bv = *reinterpret_cast<const f32x4*>(abp + 256 + k * 4);
      o.a[k] = av[0]; o.a[k + 1] = av[1]; o.a[k + 2] = av[2]; o.a[k + 3] = av[3];
      o.b[k] = bv[0]; o.b[k + 1] = bv[1]; o.b[k + 2] = bv[2]; o.b[k + 3] = bv[3];
    }
  };
  auto tr_finish = [&](auto ic, const TrOps& o) {
    constexpr int i = decltype(ic)::value;
    uint32_t ov[4];
    pp_pro_words<4>(o.raw, o.a, o.b, ov, T());
    if (pkey[i] >= 0) *reinterpret_cast<u32x4*>(smem + o.off) = u32x4{ov[0], ov[1], ov[2], ov[3]};
  };
  auto transform_piece = [&](auto ic, auto planec) {
    if constexpr (PRO != 0) { TrOps o; tr_load(ic, planec, o); tr_finish(ic, o); }
  };
  // ---------------- fragment addresses ----------------
  // Every address below is loop-invariant.  Left to itself hipcc hoists all derived addresses out of the tile loop and spills
  // them (one scratch reload + `s_waitcnt vmcnt(0)` in front of every fragment read: the DMA pipeline drained per tap); an empty asm
  // makes the five bases opaque inside each tap, so the constants fold into the ds_read `offset:` field.
  int a_base[3];                           // per tap column kx (the slot swizzle follows the pixel column); + plane, + rows / column halves as immediates
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) a_base[kx] = OFF_PLANE + (wm * WROWS * PW + lr + kx) * 64 + 16 * (lq ^ (((lr + kx) >> 1) & 3));
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
      const int co = nt * BN + wn * 64 + ni * 16 + 4 * lq;   // Cout % BN == 0: always in range
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

  constexpr int abl = PP_ABLATE;   // timing experiments, compile-time (make variant PPABL=<mask>; results become wrong): 4 no MFMA, 8 no weight DMA, 16 no patch DMA, 128 no fragment reads, 256 no prologue math
  STAMP_DECL
  CLK_DECL
  // ---------------- pipeline fill: chunk 0 of the first tile -> plane 0, weight taps 0 .. 4 -> ring slots 0 .. 4 ----------------
  cinit_load(t_first);
  issue_ab(0);
  issue_patch(IC<0>(), IC<0>(), 0); issue_patch(IC<1>(), IC<0>(), 0); issue_patch(IC<2>(), IC<0>(), 0);
  if constexpr (NPW > 3) issue_patch(IC<3>(), IC<0>(), 0);
  if constexpr (NPW > 4) issue_patch(IC<4>(), IC<0>(), 0);
  if constexpr (NPW > 5) issue_patch(IC<5>(), IC<0>(), 0);
  issue_w(IC<0>()); issue_w(IC<1>()); issue_w(IC<2>()); issue_w(IC<3>()); issue_w(IC<4>());
  acc_init();
  pp_wait_vm<4 * NWP>();                   // this wave's pieces of the patch (and (a, b)) and of tap 0 have landed (taps 1 .. 4 may fly)
  if constexpr (PRO != 0) {
    transform_piece(IC<0>(), IC<0>()); transform_piece(IC<1>(), IC<0>()); transform_piece(IC<2>(), IC<0>());
    if constexpr (NPW > 3) transform_piece(IC<3>(), IC<0>());
    if constexpr (NPW > 4) transform_piece(IC<4>(), IC<0>());
    if constexpr (NPW > 5) transform_piece(IC<5>(), IC<0>());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
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
    // one tap: c = the chunk being multiplied (runtime), P = its parity (patch plane), S = tap of the chunk; lastp: the chunk pair is the tile's last
    auto tap = [&](auto Pc, auto Sc_, int c, bool lastp) {
      constexpr int P = decltype(Pc)::value, S = decltype(Sc_)::value;
      constexpr int ky = S / 3, kx = S % 3, ring = (P * 9 + S) % NRING, ringn = (ring + AHEAD) % NRING;
      const bool last_of_tile = lastp && P == 1 && S == 8;
      // ---- L segment: this tap's fragments; this wave's share of the DMA; PRO: one of its pieces of the next chunk in place ----
      u32x4 af[8], bf[NI];
      constexpr int TRP = PRO != 0 && !(abl & 256) ? Sc::piece_of(S) : -1;   // the piece of the next chunk this tap transforms
      TrOps tro;
      if constexpr (TRP >= 0) {
        tr_load(IC<(TRP < 0 ? 0 : TRP)>(), IC<1 - P>(), tro);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        int ab = a_base[kx], bb = ring < 3 ? b_base0 : b_base1;
        asm volatile("" : "+v"(ab), "+v"(bb));
        const char* ap = smem + ab;
        const char* bp = smem + bb;
        if constexpr (!(abl & 128)) {
#pragma unroll
          for (int mi = 0; mi < 8; ++mi) af[mi] = *reinterpret_cast<const u32x4*>(ap + P * PLANE + (Dg::row(mi) + ky) * AROWB + Dg::col(mi) * 64);
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
      {
        const int scn = P == 0 ? c + 1 : (lastp ? 0 : c + 1);
        if constexpr (S == 0 && !(abl & 16)) issue_ab(scn);
        if constexpr (!(abl & 16)) issue_patch_at(IC<S>(), IC<1 - P>(), scn);
      }
      if constexpr (!(abl & 8)) issue_w(IC<ringn>());
      if constexpr (TRP >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        tr_finish(IC<(TRP < 0 ? 0 : TRP)>(), tro);
      }
#ifdef CONV_STAMPS
      const int tr_tap = c * 9 + S;
      const bool tr_on = blockIdx.x == 0 && t == t_first && tr_tap >= PP_TRACE_TAP0 && tr_tap < PP_TRACE_TAP0 + 4;   // every lane writes the same word
      uint32_t tr_a, tr_v;
#endif
      STAMP(0) PP_TRACE(0)
      constexpr int YW = Sc::yw(S), YP = Sc::yp(S), YY = YW < YP ? YW : YP;
      if constexpr (P == 0 && S < 4 && PAIR) {     // the first taps of a tile: the previous tile's epilogue is inside the weight window
        if (extra == 0) pp_wait_vm<YY>();
        else if (extra == EPI_STORES) pp_wait_vm<(YW + EPI_STORES < YP ? YW + EPI_STORES : YP)>();
        else pp_wait_vm<(YW + 2 * EPI_STORES < YP ? YW + 2 * EPI_STORES : YP)>();
      } else {
        pp_wait_vm<YY>();
      }
      STAMP(1) PP_TRACE(1)
      pp_barrier();
      STAMP(2) PP_TRACE(2)
      // ---- M segment ----
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!(abl & 4)) {
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
    const bool gn_mask = ((p.Wo & (VW - 1)) | (p.Ho & (TH - 1))) != 0;
    auto epi_half = [&](auto hc, auto resc, auto gnc) {
      constexpr int h = decltype(hc)::value, GNM = decltype(gnc)::value;
      constexpr bool HAS_RES = decltype(resc)::value != 0;
      uint32_t ovo[4], rvo[4];
      float vm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = y0 + wm * WROWS + Dg::row(h * 4 + j), x = x0 + Dg::col(h * 4 + j) + lr;
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
              gp.add(2 * k, va[0], va[1], va[2], va[3], GNM == 2, vm[j]);
              gp.add(2 * k + 1, vb[0], vb[1], vb[2], vb[3], GNM == 2, vm[j]);
            }
            const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo[j] + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
    };
    auto epi = [&](auto resc, auto gnc) { epi_half(IC<0>(), resc, gnc); epi_half(IC<1>(), resc, gnc); };
    if constexpr (ACT != 0) {
      // ---- GroupNorm32 (AD/image_diffusion/nn.py:11-13,87-94) of the tile = the whole image, fp32 statistics of the fp32 accumulators: every wave sums its
      //      128 pixels per channel quad, the two 8-row halves (wm = 0, 1: the two groups) meet in LDS behind one extra barrier (both groups run it: the
      //      barrier count per tile stays equal), the sums are added in the fixed order half 0 + half 1, and the wave stores silu?(a o + b) IN PLACE of the
      //      raw tensor: the consumer conv runs prologue-free, no statistics launch, no second copy. ----
      constexpr bool FASTA = E::DTYPE == 1;
      const int cpg = p.Cout >> 5;
      float gs[NI], gq[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) { s1 += acc[mi][ni][r4]; s2 += acc[mi][ni][r4] * acc[mi][ni][r4]; }
        gs[ni] = GnPartial<1>::row_sum(s1); gq[ni] = GnPartial<1>::row_sum(s2);
      }
      char* xb = smem + OFF_X;
      if (lr == 0) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x2*>(xb + (wn * 2 + wm) * 128 + (ni * 4 + lq) * 8) = f32x2{gs[ni], gq[ni]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_barrier();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const f32x2 h0 = *reinterpret_cast<const f32x2*>(xb + (wn * 2 + 0) * 128 + (ni * 4 + lq) * 8);
        const f32x2 h1 = *reinterpret_cast<const f32x2*>(xb + (wn * 2 + 1) * 128 + (ni * 4 + lq) * 8);
        float ts = h0[0] + h1[0], tq = h0[1] + h1[1];
        if (cpg >= 8) { ts += __shfl_xor(ts, 16); tq += __shfl_xor(tq, 16); }
        if (cpg >= 16) { ts += __shfl_xor(ts, 32); tq += __shfl_xor(tq, 32); }
        gs[ni] = ts; gq[ni] = tq;
      }
      const float inv_cnt = 1.0f / ((float)cpg * (float)(p.Ho * p.Wo));
      f32x4 ga[NI], gb[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float mean = gs[ni] * inv_cnt;
        const float var = fmaxf(gq[ni] * inv_cnt - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + p.act_eps);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.act_gamma + co_w + ni * 16);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.act_beta + co_w + ni * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = rstd * g4[j];
          float b = b4[j] - mean * a;
          if (p.act_film) {
            const float* fpt = p.act_film + (size_t)n0 * p.act_film_stride + co_w + ni * 16 + j;
            const float sc = 1.0f + fpt[0], sh = fpt[p.Cout];
            a *= sc;
            b = b * sc + sh;
          }
          ga[ni][j] = a; gb[ni][j] = b;
        }
      }
      auto actv = [&](float v) {
        return p.act_silu ? (FASTA ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)) : v / (1.0f + expf(-v))) : v;
      };
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int y = y0 + wm * WROWS + Dg::row(mi), x = x0 + Dg::col(mi) + lr;
        const uint32_t opix = (uint32_t)((n0 * p.Ho + y) * p.Wo + x);
        const uint32_t ovo = (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ;
        if constexpr (!PAIR) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = actv(ga[ni][j] * acc[mi][ni][j] + gb[ni][j]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, ovo + ni * 16 * ESZ, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP2; ++k) {
            float va[4], vb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              va[q] = actv(ga[2 * k][q] * acc[mi][2 * k][q] + gb[2 * k][q]);
              vb[q] = actv(ga[2 * k + 1][q] * acc[mi][2 * k + 1][q] + gb[2 * k + 1][q]);
            }
            const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
            const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
            const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo + k * PSTEP * ESZ, 0, 0);
          }
        }
      }
    } else if (p.res_mode != RES_NONE) {
      if (!do_gn) epi(IC<1>(), IC<0>()); else if (!gn_mask) epi(IC<1>(), IC<1>()); else epi(IC<1>(), IC<2>());
    } else {
      if (!do_gn) epi(IC<0>(), IC<0>()); else if (!gn_mask) epi(IC<0>(), IC<1>()); else epi(IC<0>(), IC<2>());
    }
    if (do_gn && ACT == 0) {   // slot = (pixel tile of the image, pixel wave); quads of this wave's 64 channels
      const int rem = mt - n0 * tpi;
      gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * WMN + wm) * (size_t)(p.Cout >> 2) + ((nt * BN + wn * 64) >> 2)) * 2, lq, lr);
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
  if (p.dbg && blockIdx.x == 0 && lane < 20) p.dbg[2048 * 8 + wave8 * 20 + lane] = *reinterpret_cast<volatile uint32_t*>(smem + LDS_END + (wave8 * 20 + lane) * 4);
#endif
}

// Shapes the ping-pong kernel takes: 3x3 / stride 1 / NHWC output, input either as it is or through the GroupNorm affine + SiLU prologue
// (PRO = 2; an affine-only prologue stays on the other kernels), an even number of 64-byte channel chunks per source switch (two chunks
// are unrolled; a source boundary may fall anywhere), images of at least one tile.  Geometry: Cout % 256 == 0 -> wide (16 x 16 x 256);
// otherwise Cout % 128 == 0 and Wo >= 32 -> narrow (16 x 32 x 128).  mode = mi355_debug_config::conv_pp: bits 0-1: 1 = only when every CU
// gets a tile, 2 = always (tests); bit 2: the prologue form of the wide geometry; bit 3: the narrow geometry; bit 4: the prologue form of the
// narrow geometry too (off by default: same-box it ties the warp-specialised kernel at 256 / 384 input channels and loses 3-5 % at 128,
// profiles/r5_experiments.md - the prologue's arithmetic is amortised over 128 output channels instead of 256).
// Returns the geometry (0 wide, 1 narrow) or -1.
static int pp_config(int mode, int ks, int G, int bn_pack, int out_mode, int stride, int nchunks, bool has_pro, bool pro_silu, int N, int Ho, int Wo, int Cout) {
  if (!(mode & 3) || ks != 3 || G != 1 || bn_pack != 128 || out_mode != OUT_NHWC || stride != 1) return -1;
  if (has_pro && !pro_silu) return -1;
  if (nchunks < 2 || (nchunks & 1)) return -1;
  int cfg = -1;
  if (Cout % 256 == 0 && Wo >= 16 && Ho >= 16) cfg = 0;
  else if (Cout % 128 == 0 && (mode & 8) && Wo >= 32 && Ho >= 16) cfg = 1;
  if (cfg < 0) return -1;
  if (has_pro && !(mode & (cfg == 0 ? 4 : 16))) return -1;
  if ((mode & 3) >= 2) return cfg;
  const int vw = cfg == 0 ? 16 : 32, bn = cfg == 0 ? 256 : 128;
  const int n_mt = N * ((Wo + vw - 1) / vw) * ((Ho + 15) / 16), n_nt = Cout / bn;
  return n_mt * n_nt >= ws_num_cus() ? cfg : -1;
}

template <typename T, int CFG, int PRO, int ACT = 0>
int launch_pp_k(ConvKArgs a, hipStream_t s) {
  using Dg = pp::D<CFG>;
  a.lvw = CFG == 0 ? 4 : 5; a.lth = 4; a.PW = Dg::PW; a.PH = Dg::PH; a.NP = Dg::NPX;
  a.tiles_x = (a.Wo + Dg::VW - 1) / Dg::VW; a.tiles_y = (a.Ho + Dg::TH - 1) / Dg::TH;
  const int n_mt = a.N * a.tiles_x * a.tiles_y, n_nt = a.Cout / Dg::BN;
  if (int rc = mi355_allow_big_lds(conv3x3_pp_kernel<T, CFG, PRO, ACT>, "conv3x3 (ping-pong)")) return rc;
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt, ncu = ws_num_cus();
  const int grid = ntp < ncu ? ntp : ncu;   // one persistent workgroup per CU
  hipLaunchKernelGGL((conv3x3_pp_kernel<T, CFG, PRO, ACT>), dim3(grid), dim3(512), Dg::lds_bytes(PRO, ACT) + PP_TRACE_BYTES, s, a, n_mt, n_nt);
  return 0;
}

// statistics slots per image the launch fills (pixel tile x pixel wave), for a geometry pp_config returned
static int pp_gn_slots(int cfg, int Ho, int Wo) {
  return cfg == 0 ? 2 * ((Wo + 15) / 16) * ((Ho + 15) / 16) : 4 * ((Wo + 31) / 32) * ((Ho + 15) / 16);
}

// 0 = launched, 1 = not eligible, < 0 = error.  a.act_out set: the caller asks for the fused output GroupNorm (in place); *act_done reports whether this
// launch did it (wide form, one 16 x 16 tile per image, no residual, 8 or 16 channels per group) - otherwise the conv runs without it.
template <typename T>
int launch_pp(ConvKArgs a, int mode, int ks, hipStream_t s, int* act_done = nullptr) {
  if (act_done) *act_done = 0;
  const int cfg = pp_config(mode, ks, a.G, a.bn_pack, a.out_mode, a.stride, a.nchunks, a.pro_a != nullptr, a.pro_silu != 0, a.N, a.Ho, a.Wo, a.Cout);
  if (cfg < 0) return 1;
  if (a.act_out) {
    const int cpg = a.Cout / 32;
    const bool ok = cfg == 0 && a.Ho == 16 && a.Wo == 16 && (cpg == 8 || cpg == 16) && a.res_mode == RES_NONE && a.act_out == a.out && !a.act_raw;
    if (!ok) a.act_out = nullptr;
  }
  if (a.act_out) {
    a.gn_stats = nullptr; a.gn_slots = 0;
    if (act_done) *act_done = 1;
    return a.pro_a ? launch_pp_k<T, 0, 2, 1>(a, s) : launch_pp_k<T, 0, 0, 1>(a, s);
  }
  if (a.pro_a) return cfg == 0 ? launch_pp_k<T, 0, 2>(a, s) : launch_pp_k<T, 1, 2>(a, s);
  return cfg == 0 ? launch_pp_k<T, 0, 0>(a, s) : launch_pp_k<T, 1, 0>(a, s);
}
