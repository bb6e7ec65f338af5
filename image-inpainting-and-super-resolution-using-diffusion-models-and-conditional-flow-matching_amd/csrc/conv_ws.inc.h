// Warp-specialised, persistent 3x3 implicit-GEMM (included by conv_igemm.hip; same args, same LDS images).
//
// Measured on the plain kernel (in-kernel stamps + ablations): its phases (patch staging + GN/SiLU
// prologue, weight staging, MFMA rows, epilogue stores) simply ADD UP - two co-resident workgroups run them in lockstep
// and a wave is in-order, so the MFMA pipe idles ~70 % of the time.  Ablations of the first warp-specialised version
// (tools/ws_ablate.sh) then showed what a 128-pixel tile cannot hide: 24 KB of weights per kernel row per workgroup
// through `ds_write_b128` (13 cycles per wave-instruction) and through the CU's 64 B/clk L1 path.  Hence:
//   * tile = 256 pixels (16 x 16) x 128 output channels: weight bytes per MFMA halve; halo 324 / 256 = 1.27x;
//   * one 8-wave workgroup per CU; waves 0-3 are LOADERS (global -> registers -> prologue -> LDS), waves 4-7 are
//     CONSUMERS (LDS -> MFMA -> epilogue), 128 pixels x 64 channels each.  Every SIMD hosts one of each;
//   * persistent: a workgroup walks tiles blockIdx.x, +gridDim.x, ...; the loaders treat the whole walk as ONE stream of
//     (tile, chunk, kernel row) items and run ahead of the consumers across tile boundaries:
//        global loads  : patch fragments and GN (a, b) one chunk (3 rows) ahead in a register ring
//        LDS commits   : two patch fragments per kernel row, transformed and written one chunk ahead of the consumers
//                        (3 patch planes); weight rows go global -> LDS by DMA one row ahead (4 row buffers);
//   * NO workgroup barrier in the stream (round 2; round 1 had one s_barrier per kernel row, and the stamps showed both roles
//     waiting at it - 750 / 1080 cycles of a 3.3k-cycle row - because a barrier runs the two streams in lockstep: every row
//     costs max(loader row, consumer row), and the consumers' epilogue (5.3k cycles per tile) stalls the loaders).  Monotonic
//     row counters in LDS instead, one per wave (a shared counter would let a fast wave's next row stand in for a slow wave's
//     current one): a loader wave adds 1 to its `prod` when its share of a kernel row (patch fragments and DMA'd weights) has
//     landed, a consumer wave adds 1 to its `cons` when its reads of a row have been issued (LDS executes a wave's operations
//     in order, so the add is performed after them); a consumer starts row R once min(prod) >= R + 1, a loader may overwrite
//     the buffers of row R - 3 once min(cons) >= R - 2 (one ds_read_b128 fetches all four).  With 4 weight buffers and 3 patch planes the
//     loaders run up to two rows further ahead than lockstep allowed: row-time variance and the epilogue are absorbed;
//   * the consumers run a register-double-buffered pipeline of half-taps (16 MFMAs each): the LDS reads of step s+1 are
//     issued before the MFMAs of step s, and the hand-over to the next row sits between the last reads and the last
//     MFMAs of the current row, so neither LDS latency nor the counter poll stalls the matrix pipe.
// Every wait is a poll of a counter that the other role advances without waiting for the poller (a consumer at row R has
// released R - 1, which is all the loaders need for row R + 1), bounded by a spin limit as a last resort: no hang, and the launch is
// flagged (error word -> MI355_ERR_TIMEOUT at the C boundary).
#ifndef WS_ABLATE
#define WS_ABLATE 0   // diagnostic builds only: 4 = consumers skip LDS reads + MFMA, 8 = no weight loads, 16 = no patch loads, 64 = no weight LDS writes
#endif
// poll back-off of the counter waits, in units of 64 cycles: wait_ge (the loaders' buffer-free wait, the consumers' once-per-tile wait)
// and release_acquire (the consumers' row hand-over).  The loaders run ahead and wait for a free buffer most of the time; every poll
// costs the MFMA wave on the same SIMD issue slots, and a late wake-up costs nothing while the ring holds a row of slack: same box,
// 1 / 2 / 8 / 16 / 32 -> 1250-1260 / +0.1 / +0.2-0.3 / +0.3 / +0.15 % end to end; 0 for the consumers' hand-over -0.1 %.
#ifndef WS_LSLEEP
#define WS_LSLEEP 8
#endif
#ifndef WS_CSLEEP
#define WS_CSLEEP 2   // consumers' hand-over (with WS_LSLEEP 8): 1 / 2 / 4 -> 1262.2 / 1265.3 / 1264.7 images/s
#endif
#define WS_STR2(x) #x
#define WS_STR(x) WS_STR2(x)
namespace ws {
constexpr int VW = 16, TH = 16, PW = VW + 2, PH = TH + 2, NPX = PW * PH;   // 18 x 18 = 324 patch pixels
constexpr int PIT = 6, FR = 64, PLANE = NPX * 64;                           // 20,736 B per patch plane: dense 64-B pixel rows, the 16-B slots
                                                                            // XOR-swizzled by the pixel's COLUMN ((px >> 1) & 3): conflict-free b128
                                                                            // reads for every tap shift, and row shifts are plain address offsets
constexpr int AROWB = PW * 64;                                              // bytes between patch rows
constexpr int BN = 128, WTILE = BN * 64, WIT = 3 * WTILE / (256 * 16);      // 6 x 16 B per loader thread per kernel row
constexpr int CBUF = BN * 4;                                                // bias + emb of one tile's channels (f32)
constexpr int NPLANES = 3, NWBUF = 4;                                        // patch planes / weight row buffers
constexpr size_t LDS_BYTES = NPLANES * (size_t)PLANE + NWBUF * 3 * (size_t)WTILE + 2 * CBUF + 64;   // 160,576 B (the eight counters, the give-up flag)
// Counter polls are bounded (p.spin_limit, default 1 << 22 polls of >= 64 cycles: seconds): a protocol bug or a wave that never arrives
// ends in MI355_ERR_TIMEOUT instead of a hung GPU.  A wait that gives up stores 1 into the launch's error word (p.err; cold path inside
// the poll's asm block), every later wait of that wave polls once, and the wave goes on: the tile is wrong, the grid drains at once.
}  // namespace ws

// GN affine (+ SiLU) of one 16-byte fragment.  bf16: fp32 math on element PAIRS (v_pk_fma/mul/add_f32: two elements per VALU
// issue; the loaders' VALU stream competes with the consumers' MFMA issue on the same SIMD, so every instruction counts).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <typename T2>   // bf16 / f16
__device__ __forceinline__ u32x4 ws_pro_frag(const u32x4& raw, const f32x2 (&a2)[4], const f32x2 (&b2)[4], bool silu, T2) {
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t w = raw[i];
    f32x2 x;
    { float xl, xh; unpack2(w, xl, xh, T2()); x = f32x2{xl, xh}; }
    f32x2 v = a2[i] * x + b2[i];
    if (silu && (WS_ABLATE & 128)) {   // diagnostic only: same instruction count without the two transcendentals
      const f32x2 sc = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
      const f32x2 d = sc * sc + f32x2{1.0f, 1.0f};
      v = v * (d * sc + d);
    } else if (silu) {   // v * rcp(1 + exp2(-log2(e) * v)): the same operation sequence as silu_fast<true>
      const f32x2 sc = v * f32x2{-1.4426950408889634f, -1.4426950408889634f};
      const f32x2 d = f32x2{__builtin_amdgcn_exp2f(sc[0]), __builtin_amdgcn_exp2f(sc[1])} + f32x2{1.0f, 1.0f};
      v = v * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    }
    out[i] = pack2(v[0], v[1], T2());
  }
  return out;
}
__device__ __forceinline__ u32x4 ws_pro_frag(const u32x4& raw, const f32x2 (&a2)[2], const f32x2 (&b2)[2], bool silu, float) {
  const f32x4 x = __builtin_bit_cast(f32x4, raw);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = a2[j >> 1][j & 1] * x[j] + b2[j >> 1][j & 1];
    o[j] = silu ? silu_fast<false>(v) : v;
  }
  return __builtin_bit_cast(u32x4, o);
}

template <typename T, int PRO>   // PRO: 0 = no prologue, 1 = GN affine, 2 = GN affine + SiLU
__global__ void __launch_bounds__(512, 2) conv3x3_ws_kernel(ConvKArgs p, int n_mt, int n_nt) {
  using namespace ws;
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, ESZ = sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pbuf = smem;                       // NPLANES patch planes
  char* wbuf = smem + NPLANES * PLANE;     // NWBUF x (3 weight tiles of one kernel row)
  char* cbuf = wbuf + NWBUF * 3 * WTILE; // 2 x accumulator start values (bias + timestep embedding) of a tile's 128 channels
  uint32_t* c_prod = reinterpret_cast<uint32_t*>(cbuf + 2 * CBUF);   // [4]: kernel rows staged by loader wave w
  uint32_t* c_cons = c_prod + 4;                                     // [4]: kernel rows read by consumer wave w
  // c_prod[12]: the workgroup's give-up flag (see wait_ge); c_prod[13 .. 15]: pad

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
    n0 = ng; y0 = tyi * TH; x0 = txi * VW;
  };
  int t_first = (int)blockIdx.x - (int)gridDim.x;
  t_first = next_valid(t_first);
  if (t_first >= ntp) return;        // whole workgroup leaves together: no barrier is ever reached
  if (threadIdx.x < 16) c_prod[threadIdx.x] = 0u;
  if constexpr ((WS_ABLATE & 2048) != 0) {   // consumers-alone experiment: random bf16 operands in [0.5, 1) of either sign (zeros would raise the clock)
    for (int i = threadIdx.x; i < (NPLANES * PLANE + NWBUF * 3 * WTILE) / 16; i += 512) {
      uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
      u32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[k] = (h & 0x807f807fu) | 0x3f003f00u; }
      *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = v;
    }
  }
  __syncthreads();                   // the only barrier of the kernel
  // poll until min(cp[0..3]) >= target; the other role never waits for this wave to get there
  // Both helpers are single asm blocks: as C++ (a spin loop, an `if (lane == 0)`) they put control flow into the consumers' row loop,
  // and the register allocator, already at the 256-VGPR limit there, answered with 16-byte scratch spills per row.  EXEC is all
  // ones at every call site (wave-uniform control flow only) and is restored; s_waitcnt lgkmcnt(0) also retires the wave's own
  // outstanding fragment reads, which it would have to wait for before the next row's MFMAs anyway.
  // the launch's error word (conv_launch always passes one: the caller's, or a scratch word of the workspace)
  const __attribute__((address_space(1))) uint32_t* errp = (const __attribute__((address_space(1))) uint32_t*)p.err;
  // The give-up is STICKY per workgroup: the first wait that expires sets a flag word in LDS (c_prod[12]); every poll loop looks at it before
  // it sleeps (slow path only) and leaves at once when it is set, so a flagged launch drains in microseconds (the default limit is about
  // a second per wait, a workgroup runs ~190 waits per tile, and sampler loops have dozens of launches queued behind).  FO = byte
  // offset of the flag from the counter row the wait polls (48 from c_prod, 32 from c_cons): an immediate, no register.
  // (two plain lambdas from one macro: as a generic lambda over the offset hipcc rejects the capture of the address-space-qualified errp)
  // (in the slow path: the flag read = another wait of this workgroup has given up, leave at once; label 4 = gave up: set the workgroup's flag and
  //  flag the launch - one lane, a plain system-scope store: the word is host memory and only bit 0 is ever set, no PCIe AtomicOps needed - then go on)
#define WS_WAIT_GE_BODY(FO_) \
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)cp; \
    uint32_t v0, v1, v2, v3; int sv, spins = p.spin_limit; \
    asm volatile( \
        "1:\n\t" \
        "ds_read_b32 %[v0], %[a]\n\t" \
        "ds_read_b32 %[v1], %[a] offset:4\n\t" \
        "ds_read_b32 %[v2], %[a] offset:8\n\t" \
        "ds_read_b32 %[v3], %[a] offset:12\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_min_u32 %[v0], %[v0], %[v1]\n\t" \
        "v_min_u32 %[v2], %[v2], %[v3]\n\t" \
        "v_min_u32 %[v0], %[v0], %[v2]\n\t" \
        "s_nop 0\n\t" \
        "v_readfirstlane_b32 %[sv], %[v0]\n\t" \
        "s_cmp_ge_u32 %[sv], %[tg]\n\t" \
        "s_cbranch_scc1 2f\n\t" \
        "s_sub_u32 %[sp], %[sp], 1\n\t" \
        "s_cmp_eq_u32 %[sp], 0\n\t" \
        "s_cbranch_scc1 4f\n\t" \
        "ds_read_b32 %[v1], %[a] offset:%[fo]\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "v_readfirstlane_b32 %[sv], %[v1]\n\t" \
        "s_cmp_lg_u32 %[sv], 0\n\t" \
        "s_cbranch_scc1 2f\n\t" \
        "s_sleep " WS_STR(WS_LSLEEP) "\n\t" \
        "s_branch 1b\n\t" \
        "4:\n\t" \
        "s_mov_b64 exec, 1\n\t" \
        "v_mov_b32 %[v1], 1\n\t" \
        "ds_write_b32 %[a], %[v1] offset:%[fo]\n\t" \
        "s_cmp_eq_u64 %[err], 0\n\t" \
        "s_cbranch_scc1 5f\n\t" \
        "v_mov_b32 %[v0], 0\n\t" \
        "global_store_dword %[v0], %[v1], %[err] sc0 sc1\n\t" \
        "5:\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "2:" \
        : [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [sv] "=&s"(sv), [sp] "+s"(spins) \
        : [a] "v"(a), [tg] "s"(target), [err] "s"(errp), [fo] "i"(FO_) \
        : "scc", "memory");
  auto wait_ge_prod = [&](const uint32_t* cp, uint32_t target) { WS_WAIT_GE_BODY(48) };   // polls c_prod
  auto wait_ge_cons = [&](const uint32_t* cp, uint32_t target) { WS_WAIT_GE_BODY(32) };   // polls c_cons
#undef WS_WAIT_GE_BODY

  // The consumers' row hand-over, split so that the poll's LDS round trip hides behind a half-tap of MFMAs: `poll_issue` reads the
  // four producer counters (before the MFMAs of step 4), `release_acquire` (step 5) releases the finished row and only spins if
  // the early values were not there yet.
  // (the early read is a C++ load, so the compiler itself waits for it before the values are used or moved)
  // (an LDS-address-space pointer: through a generic `volatile` pointer this compiled to flat_load_dwordx4 sc0 sc1 followed by
  //  s_waitcnt vmcnt(0) lgkmcnt(0) in front of the step's MFMAs - address-space inference skips volatile accesses, and a flat
  //  load returns out of order with LDS reads - i.e. the "early" poll drained the fragment pipeline once per kernel row)
  auto poll_issue = [&](const uint32_t* cp, uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3) {
    const u32x4 v = *reinterpret_cast<const volatile __attribute__((address_space(3))) u32x4*>((const __attribute__((address_space(3))) uint32_t*)cp);
    v0 = v[0]; v1 = v[1]; v2 = v[2]; v3 = v[3];
  };
  auto release_acquire = [&](uint32_t* mine, const uint32_t* cp, uint32_t target, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    const uint32_t am = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)mine;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)cp;
    const uint32_t one = 1u;
    int sv, spins = p.spin_limit;
    asm volatile(
        "s_mov_b64 exec, 1\n\t"
        "ds_add_u32 %[am], %[one]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 3f\n\t"
        "1:\n\t"
        "ds_read_b32 %[v0], %[a]\n\t"
        "ds_read_b32 %[v1], %[a] offset:4\n\t"
        "ds_read_b32 %[v2], %[a] offset:8\n\t"
        "ds_read_b32 %[v3], %[a] offset:12\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "3:\n\t"
        "v_min_u32 %[v0], %[v0], %[v1]\n\t"
        "v_min_u32 %[v2], %[v2], %[v3]\n\t"
        "v_min_u32 %[v0], %[v0], %[v2]\n\t"
        "s_nop 0\n\t"
        "v_readfirstlane_b32 %[sv], %[v0]\n\t"
        "s_cmp_ge_u32 %[sv], %[tg]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sub_u32 %[sp], %[sp], 1\n\t"
        "s_cmp_eq_u32 %[sp], 0\n\t"
        "s_cbranch_scc1 4f\n\t"
        "ds_read_b32 %[v1], %[a] offset:48\n\t"   // (cp = c_prod here) another wait of this workgroup has given up: leave at once
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[sv], %[v1]\n\t"
        "s_cmp_lg_u32 %[sv], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_sleep " WS_STR(WS_CSLEEP) "\n\t"
        "s_branch 1b\n\t"
        "4:\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b32 %[v1], 1\n\t"
        "ds_write_b32 %[a], %[v1] offset:48\n\t"
        "s_cmp_eq_u64 %[err], 0\n\t"
        "s_cbranch_scc1 5f\n\t"
        "v_mov_b32 %[v0], 0\n\t"
        "global_store_dword %[v0], %[v1], %[err] sc0 sc1\n\t"
        "5:\n\t"
        "s_mov_b64 exec, -1\n\t"
        "2:"
        : [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3), [sv] "=&s"(sv), [sp] "+s"(spins)
        : [a] "v"(a), [am] "v"(am), [one] "v"(one), [tg] "s"(target), [err] "s"(errp)
        : "scc", "memory");
  };
  auto bump = [&](uint32_t* cp) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cp;
    const uint32_t one = 1u;
    asm volatile(
        "s_mov_b64 exec, 1\n\t"
        "ds_add_u32 %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(a), "v"(one)
        : "memory");
  };

  if (loader) {
    // ================================= LOADER waves =================================
    if constexpr ((WS_ABLATE & 2048) != 0) return;   // timing experiment: consumers alone (no staging, no hand-over; they multiply whatever LDS holds)
    switch (p.stagger > 0 ? ((p.stagger >> 2) & 3) : 2) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; case 3: __builtin_amdgcn_s_setprio(3); break; default: break; }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);
    if constexpr (PRO == 0 && !(WS_ABLATE & 1024)) {
    // ---------------- prologue-free input: the loaders issue DMA and counter updates only (round 3) ----------------
    // Round-2 finding: whatever a loader wave issues is paid for by the MFMA stream of the consumer wave on its SIMD (same-box A/B in
    // profiles/r3_experiments.md: removing a stall from the consumers alone slowed the loaders by as much).  Without a prologue nothing
    // has to pass through registers: the patch goes global -> LDS by DMA like the weights.  A plane is 324 dense 64-byte pixel rows
    // = 21 pieces of 1 KB (16 pixels; piece 20 = pixels 308-323, re-writing 12 pixels of piece 19 with the same bytes: no partial
    // piece, no padding).  Lane l of a piece lands at piece base + 16 l = slot l & 3 of pixel (l >> 2), so the slot swizzle goes
    // into the per-lane SOURCE address (fragment (l & 3) ^ ((px >> 1) & 3) of that pixel), zero padding = an out-of-range offset
    // (the DMA then writes zeros: tools/probe/lds_dma_oob_probe.cpp).  Chunk c + 1's 21 pieces are issued during chunk c's three
    // intervals: interval ky moves pieces 7 ky .. 7 ky + 6, wave w taking 7 ky + w and (w < 3) 7 ky + 4 + w, in front of the six
    // weight pieces of the next kernel row; the counted wait at the end of an interval leaves exactly that interval's own pieces
    // in flight, so everything a published row needs (its weights and ALL pieces of its chunk) was issued at least one interval ago.
    constexpr int NPU = 6;                       // pieces per wave and chunk: u = 2 ky + h
    const bool has_h1 = wave8 < 3;
    uint32_t pvo0[NPU], pvo1[NPU];               // per-lane source offsets of piece u in source 0 / 1 (tile-dependent)
    auto piece_first_pixel = [&](int u) { const int j = 7 * (u >> 1) + 4 * (u & 1) + wave8; return j == 20 ? NPX - 16 : 16 * j; };
    auto setup = [&](int mt) {
      int n0, y0, x0;
      origin(mt, n0, y0, x0);
      const int cy0 = y0 - 1, cx0 = x0 - 1;
#pragma unroll
      for (int u = 0; u < NPU; ++u) {
        const int i = piece_first_pixel(u) + (lane >> 2);
        const int py = (int)(((float)i + 0.5f) * (1.0f / (float)PW)), px = i - py * PW;   // exact: i < 384
        const int fqx = (lane & 3) ^ ((px >> 1) & 3);
        const int cy = cy0 + py, cx = cx0 + px;
        int sp = -1;
        if (cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
          if (p.mode == CONV_UP2) sp = (n0 * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1);
          else sp = (n0 * p.Hs + cy) * p.Ws + cx;
        }
        pvo0[u] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C0 * ESZ) + fqx * 16 : p.bytes0;
        pvo1[u] = sp >= 0 ? (uint32_t)sp * (uint32_t)(p.C1 * ESZ) + fqx * 16 : p.bytes1;
      }
    };
    uint32_t woff[WIT];
#pragma unroll
    for (int i = 0; i < WIT; ++i) woff[i] = (i * 256 + tid) * 16;
    // patch stream: the chunk whose pieces are being issued (one chunk ahead of the row stream)
    int pt_t = t_first, pt_c = 0, pt_plane = 0;
    { int mt, nt; decode(pt_t, mt, nt); setup(mt); }
    auto pt_advance = [&]() {
      pt_plane = pt_plane == NPLANES - 1 ? 0 : pt_plane + 1;
      if (++pt_c == p.nchunks) {
        pt_c = 0;
        pt_t = next_valid(pt_t);
        if (pt_t < ntp) { int mt, nt; decode(pt_t, mt, nt); setup(mt); }
      }
    };
    // this wave's pieces of slot ky for the patch stream's chunk -> number issued (wave-uniform)
    auto issue_pieces = [&](auto kyc) -> int {
      constexpr int ky = decltype(kyc)::value;
      if (pt_t >= ntp) return 0;
      if constexpr (WS_ABLATE & 16) return 0;
      const int cb = src_chunk(p, pt_c) * CHUNK;
      const bool first = cb < p.C0;
      const uint32_t so = (uint32_t)((first ? cb : cb - p.C0) * ESZ);
      char* pl = pbuf + pt_plane * PLANE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)(pl + piece_first_pixel(2 * ky) * 64), 16,
                                               first ? pvo0[2 * ky] : pvo1[2 * ky], so, 0, 0);
      if (!has_h1) return 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)(pl + piece_first_pixel(2 * ky + 1) * 64), 16,
                                               first ? pvo0[2 * ky + 1] : pvo1[2 * ky + 1], so, 0, 0);
      return 2;
    };
    int wl_t = t_first, wl_row = 0, wl_nt, wl_buf = 0;
    { int mt; decode(wl_t, mt, wl_nt); }
    auto dma_w = [&]() {               // next row of the weight stream -> wbuf[wl_buf] (6 pieces per wave)
      const uint32_t so = ((uint32_t)wl_nt * p.nchunks * 9 + (uint32_t)wl_row * 3) * WTILE;
      if constexpr (!(WS_ABLATE & 8)) {
        char* dst = wbuf + wl_buf * (3 * WTILE) + wave8 * 1024;
#pragma unroll
        for (int i = 0; i < WIT; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, woff[i], so, 0, 0);
      }
      wl_buf = wl_buf == NWBUF - 1 ? 0 : wl_buf + 1;
      if (++wl_row == ngr) {
        wl_row = 0;
        wl_t = next_valid(wl_t);
        if (wl_t < ntp) { int mt; decode(wl_t, mt, wl_nt); }
      }
    };
    // all but this interval's own DMA pieces (6 weight pieces + k patch pieces) have landed; lgkmcnt(0): this wave's cbuf write
    auto wait_landed = [&](int k) {
      asm volatile("" ::: "memory");
      constexpr int WK = (WS_ABLATE & 8) ? 0 : WIT;
      if (k == 2) __builtin_amdgcn_s_waitcnt((WK + 2) | (7 << 4) | (0 << 8));
      else if (k == 1) __builtin_amdgcn_s_waitcnt((WK + 1) | (7 << 4) | (0 << 8));
      else __builtin_amdgcn_s_waitcnt(WK | (7 << 4) | (0 << 8));
    };
    int R = 0;
    auto acquire_free = [&]() {
      if (R >= NWBUF - 1) wait_ge_cons(c_cons, (uint32_t)(R - (NWBUF - 2)));
    };
    // accumulator start values (bias + timestep embedding) of a tile's 128 channels, staged in LDS for the consumers: loaded at the
    // start of a tile's LAST chunk, committed at its end (the compiler's wait for the two loads then finds them long complete)
    f32x4 cv_b = f32x4{0.f, 0.f, 0.f, 0.f}, cv_e = f32x4{0.f, 0.f, 0.f, 0.f};
    int cst_par = 0;
    auto cinit_load = [&](int t) {
      int mt, nt, n0, y0, x0;
      decode(t, mt, nt);
      origin(mt, n0, y0, x0);
      const int co = min(nt * BN + (tid & 31) * 4, p.Cout - 4);
      const float* dummy = reinterpret_cast<const float*>(p.w);
      cv_b = *reinterpret_cast<const f32x4*>(p.bias ? p.bias + co : dummy);
      cv_e = *reinterpret_cast<const f32x4*>(p.emb ? p.emb + (size_t)n0 * p.emb_stride + co : dummy);
    };
    auto cinit_commit = [&]() {
      if (tid < 32) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (p.bias ? cv_b[j] : 0.0f) + (p.emb ? cv_e[j] : 0.0f);
        *reinterpret_cast<f32x4*>(cbuf + cst_par * CBUF + tid * 16) = v;
      }
      cst_par ^= 1;
    };
    STAMP_DECL
    // ---- fill: accumulator start values of the first tile, all pieces of its chunk 0, weight row 0 ----
    cinit_load(t_first);
    cinit_commit();
    (void)issue_pieces(IC<0>()); (void)issue_pieces(IC<1>()); (void)issue_pieces(IC<2>());
    pt_advance();
    dma_w();                           // row 0
    STAMP(0)
    for (int t = t_first; t < ntp;) {
      const int t_next = next_valid(t);
      for (int c = 0; c < p.nchunks; ++c) {
        const bool last_c = c + 1 == p.nchunks && t_next < ntp;
        auto tail = [&](int k) {
          dma_w();
          STAMP(3)
          wait_landed(k);
          STAMP(1)
          if (!((p.ablate & 32) && R == 5)) bump(c_prod + wave8);        // this wave's share of kernel row (t, c, ky), and of every piece issued before this interval, is in LDS
          ++R;
          acquire_free();              // before anything of the next interval is issued
          STAMP(4)
        };
        if (last_c) cinit_load(t_next);
        const int k0 = issue_pieces(IC<0>());
        tail(k0);
        const int k1 = issue_pieces(IC<1>());
        tail(k1);
        const int k2 = issue_pieces(IC<2>());
        pt_advance();
        if (last_c) cinit_commit();
        tail(k2);
      }
      t = t_next;
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // no DMA piece may still be in flight towards LDS when the workgroup retires
    STAMP_FLUSH
    } else {
    // patch fragment u of a thread: pixel frow + 64 u (u < 5: 320 of the 324 patch pixels, 80 per wave); the last four pixels are
    // fragment 5 of lanes 0-15 of ONE wave, which changes with every chunk (`xw`): all four loader waves do 5.25 commits per chunk
    // on average instead of 6 / 5 / 5 / 5 (the slowest wave sets the pace of a row)
    const int fq = tid & 3, frow = tid >> 2;
    const bool pok5 = (lane >> 2) < 4;
    const bool pro = PRO && !(p.ablate & 2);
    int pdst[PIT];                   // LDS offset of fragment u inside a plane (tile-independent)
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
      const int i = u < 5 ? frow + u * FR : 320 + (lane >> 2);
      const int py = (int)(((float)i + 0.5f) * (1.0f / (float)PW)), px = i - py * PW;
      pdst[u] = i * 64 + 16 * (fq ^ ((px >> 1) & 3));
    }
    int xw = 0;                      // the wave that commits fragment 5 of the chunk being completed
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
        const int i = u < 5 ? frow + u * FR : 320 + (lane >> 2);
        int s = -1;
        if (u < 5 || pok5) {
          const int py = (int)(((float)i + 0.5f) * (1.0f / (float)PW)), px = i - py * PW;   // exact: i < 384
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
    u32x4 raw[PIT] = {};                                  // fragment u of the chunk that is committed next
    f32x2 pa[V / 2] = {}, pb[V / 2] = {}, pan[V / 2] = {}, pbn[V / 2] = {};   // GN (a, b) of that chunk / of the one after it, as element pairs
    // Every load below is issued unconditionally (a finished stream keeps re-reading its last valid addresses and the data
    // is never committed): a load under a branch makes the compiler's s_waitcnt bookkeeping assume the shortest queue,
    // i.e. vmcnt(0) at every use, which would drain the whole run-ahead.
    auto issue_frag = [&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int cb = src_chunk(p, ld_c) * CHUNK;
      const bool first = cb < p.C0;
      if constexpr (!(WS_ABLATE & 16)) raw[u] = buf_load16(first ? rs0 : rs1, first ? voff0[u] : voff1[u], (first ? cb : cb - p.C0) * ESZ);
    };
    auto issue_ab = [&]() {
      if constexpr (PRO) {
        const float* ap = p.pro_a + (size_t)n0_ld * p.Cin + src_chunk(p, ld_c) * CHUNK + fq * V;
        const float* bp = p.pro_b + (size_t)n0_ld * p.Cin + src_chunk(p, ld_c) * CHUNK + fq * V;
#pragma unroll
        for (int j = 0; j < V / 2; ++j) { pan[j] = *reinterpret_cast<const f32x2*>(ap + 2 * j); pbn[j] = *reinterpret_cast<const f32x2*>(bp + 2 * j); }
      }
    };
    auto take_ab = [&]() {
#pragma unroll
      for (int j = 0; j < V / 2; ++j) { pa[j] = pan[j]; pb[j] = pbn[j]; }
    };
    auto commit_frag = [&](auto uc, int plane, uint32_t mask) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u == 5) { if (wave8 != xw) return; }
      u32x4 outv = raw[u];
      if (pro) {
        outv = ws_pro_frag(raw[u], pa, pb, PRO == 2, T());
        if (!((mask >> u) & 1u)) outv = u32x4{0u, 0u, 0u, 0u};   // zero padding applies AFTER the prologue
      }
      if (u < 5 || pok5) *reinterpret_cast<u32x4*>(pbuf + plane * PLANE + pdst[u]) = outv;
    };

    // ---- weight stream: one kernel row ahead, moved by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR round trip, no
    //      ds_write; lane l of a wave lands at M0 base + 16 l, probed in tools/probe/lds_dma_probe.cpp).  The host packer
    //      stores weights already in their LDS image, so a row (3 taps x 8 KB) is 6 linear 1-KB pieces per wave. ----
    int wl_t = t_first, wl_row = 0, wl_nt, wl_buf = 0;
    { int mt; decode(wl_t, mt, wl_nt); }
    auto dma_w = [&]() {               // next row of the stream -> wbuf[wl_buf]
      const uint32_t so = ((uint32_t)wl_nt * p.nchunks * 9 + (uint32_t)wl_row * 3) * WTILE;
      if constexpr (!(WS_ABLATE & 8)) {
        char* dst = wbuf + wl_buf * (3 * WTILE) + wave8 * 1024;
#pragma unroll
        for (int i = 0; i < WIT; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, woff[i], so, 0, 0);
      }
      wl_buf = wl_buf == NWBUF - 1 ? 0 : wl_buf + 1;
      if (++wl_row == ngr) {
        wl_row = 0;
        wl_t = next_valid(wl_t);
        if (wl_t < ntp) { int mt; decode(wl_t, mt, wl_nt); }
      }
    };
    // All but the 8 youngest vector-memory operations of this wave are complete: at the end of an interval those 8 are the
    // interval's two patch-fragment loads and the 6 DMA pieces of the NEXT row, so the row published by the coming barrier
    // (DMA issued one interval ago) has landed while nothing younger is waited for.  (gfx9 s_waitcnt: vmcnt[3:0] | expcnt<<4 | lgkmcnt<<8 | vmcnt[5:4]<<14)
    // Together with the wave's own LDS writes (lgkmcnt 0) this is all the coming barrier has to order, so the loaders use the bare
    // s_barrier: __syncthreads()'s workgroup release fence would drain vmcnt to 0 (the DMA counts as an LDS write), i.e. wait for
    // the next row's DMA and the next chunk's patch fragments at every kernel row.
    auto wait_landed = [&]() {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(8 | (7 << 4) | (0 << 8));   // vmcnt(8) lgkmcnt(0)
    };
    // Row R of the stream (all tiles): weights in buffer R % NWBUF, patch in plane (R / 3) % NPLANES.  The interval that ends by
    // publishing row R writes patch fragments for rows >= R (planes last read by row R - 5 or earlier) and queues the DMA of row
    // R + 1 into the buffer row R + 1 - NWBUF = R - 3 was read from: both are free once the consumers have released row R - 3.
    int R = 0;
    auto acquire_free = [&]() {
      if (R >= NWBUF - 1) wait_ge_cons(c_cons, (uint32_t)(R - (NWBUF - 2)));
    };

    // ---- accumulator start values of a tile: 128 channels, 4 per thread of the first half-wave ----
    f32x4 cv_b = f32x4{0.f, 0.f, 0.f, 0.f}, cv_e = f32x4{0.f, 0.f, 0.f, 0.f};
    int cst_par = 0;
    auto cinit_load = [&](int t) {     // loads only (no wait, no branch: see issue_frag), consumed by cinit_commit two intervals later
      int mt, nt, n0, y0, x0;
      decode(t, mt, nt);
      origin(mt, n0, y0, x0);
      const int co = min(nt * BN + (tid & 31) * 4, p.Cout - 4);   // Cout % 128 == 0 for this kernel; the clamp only guards the address
      const float* dummy = reinterpret_cast<const float*>(p.w);           // any readable 16 bytes: an absent operand is zeroed at commit
      cv_b = *reinterpret_cast<const f32x4*>(p.bias ? p.bias + co : dummy);
      cv_e = *reinterpret_cast<const f32x4*>(p.emb ? p.emb + (size_t)n0 * p.emb_stride + co : dummy);
    };
    auto cinit_commit = [&]() {
      const float fb = p.bias ? 1.0f : 0.0f, fe = p.emb ? 1.0f : 0.0f;
      if (tid < 32) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (p.bias ? cv_b[j] : 0.0f) + (p.emb ? cv_e[j] : 0.0f);
        *reinterpret_cast<f32x4*>(cbuf + cst_par * CBUF + tid * 16) = v;
      }
      (void)fb; (void)fe;
      cst_par ^= 1;
    };
    cinit_load(t_first);
    STAMP_DECL
    // ---- fill the pipeline: fragments 0-3 of chunk 0 committed, 4-5 still in registers, chunk 1 and row 0 in flight ----
    issue_frag(IC<0>()); issue_frag(IC<1>()); issue_frag(IC<2>()); issue_frag(IC<3>()); issue_frag(IC<4>()); issue_frag(IC<5>());
    issue_ab();
    dma_w();                           // row 0
    uint32_t vmask_cm = vmask_ld;      // chunks 0 and 1 belong to the first tile (nchunks >= 2)
    take_ab();
    ld_advance();                      // -> chunk 1
    issue_ab();
    commit_frag(IC<0>(), 0, vmask_cm); issue_frag(IC<0>());
    commit_frag(IC<1>(), 0, vmask_cm); issue_frag(IC<1>());
    commit_frag(IC<2>(), 0, vmask_cm); issue_frag(IC<2>());
    commit_frag(IC<3>(), 0, vmask_cm); issue_frag(IC<3>());
    cinit_commit();

    STAMP(0)
    int plane = 0;                     // plane of the chunk the consumers multiply during this body
    for (int t = t_first; t < ntp;) {
      const int t_next = next_valid(t);
      // One chunk (t, c) of the consumer stream = three intervals, each ending in the barrier that publishes one kernel row:
      //   ky = 0: the last two fragments of chunk c itself (its plane was still being read one chunk ago until now)
      //   ky = 1, 2: fragments 0-3 of the chunk AFTER c into the other plane (free since the barrier that ended ky = 0)
      // every committed fragment's registers are refilled at once with the same fragment of the following chunk, and every
      // interval queues the DMA of the next kernel row behind its fragment loads.
      auto body = [&](int c) {
        bool nx_ok = true;
        auto tail = [&]() {
          dma_w();
          STAMP(3)
          wait_landed();
          STAMP(1)                     // (diagnostic build: slot 1 = time spent in the counted wait)
          if (!((p.ablate & 32) && R == 5)) bump(c_prod + wave8);        // this wave's share of kernel row (t, c, ky) and of its patch is in LDS
          ++R;
          acquire_free();              // before anything of the next interval is written
          STAMP(4)
        };
        // ---- ky = 0 ----
        cinit_load(t_next < ntp ? t_next : t);   // next tile's accumulator start values (tiny, L2-resident; loaded every chunk so that no load sits under a branch)
        commit_frag(IC<4>(), plane, vmask_cm); issue_frag(IC<4>());
        commit_frag(IC<5>(), plane, vmask_cm); issue_frag(IC<5>());
        xw = (xw + 1) & 3;
        STAMP(2)
        tail();
        // ---- switch to the next chunk ----
        if (c + 1 == p.nchunks) {      // it opens the next tile; the load stream is at that chunk (its geometry is current)
          nx_ok = t_next < ntp;
          vmask_cm = vmask_ld;
        }
        take_ab();
        ld_advance();                  // -> the chunk after the next one
        issue_ab();
        plane = plane == NPLANES - 1 ? 0 : plane + 1;
        // ---- ky = 1 ----
        if (nx_ok) commit_frag(IC<0>(), plane, vmask_cm);
        issue_frag(IC<0>());
        if (nx_ok) commit_frag(IC<1>(), plane, vmask_cm);
        issue_frag(IC<1>());
        STAMP(2)
        tail();
        // ---- ky = 2 ----
        if (nx_ok) commit_frag(IC<2>(), plane, vmask_cm);
        issue_frag(IC<2>());
        if (nx_ok) commit_frag(IC<3>(), plane, vmask_cm);
        issue_frag(IC<3>());
        if (c + 1 == p.nchunks && nx_ok) cinit_commit();
        STAMP(2)
        tail();
      };
      for (int c = 0; c < p.nchunks; c += 2) { body(c); body(c + 1); }   // nchunks is even (launcher); two chunks per trip keep the in-flight fragment registers free of loop-carried copies
      t = t_next;
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // no DMA piece may still be in flight towards LDS when the workgroup retires
    STAMP_FLUSH
    }   // register-staged loaders (the prologue variants)
  } else {
    // ================================= CONSUMER waves =================================
    // the MFMA stream goes first when both waves of a SIMD are ready (experiment knob: MI355_CONV_STAGGER = consumer | loader << 2)
    switch (p.stagger > 0 ? (p.stagger & 3) : 1) { case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; case 3: __builtin_amdgcn_s_setprio(3); break; default: break; }
    const int wave = wave8 - 4;
    const int wm = wave >> 1, wn = wave & 1;   // pixel rows 8*wm .. 8*wm+7 of the tile, channels 64*wn .. 64*wn+63
    const int lr = lane & 15, lq = lane >> 4;
    int a_cur[3];                                                           // per tap column kx (the slot swizzle follows the pixel column); + mi * AROWB
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a_cur[kx] = (wm * 8 * PW + lr + kx) * 64 + 16 * (lq ^ (((lr + kx) >> 1) & 3));
    int b_cur = (wn * 64 + lr) * 64 + 16 * (lq ^ ((lr >> 1) & 3));          // + ni * 1024 (+ kx * WTILE)
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (p.ablate & 1) ? 0u : p.obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);
    constexpr bool PAIR = E::DTYPE == 1;
    constexpr int NI = 4, NP2 = PAIR ? NI / 2 : NI, PSTEP = PAIR ? 32 : 16;

    f32x4 acc[8][NI];
    u32x4 af[2][4], bf[2][4];
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    f32x16_t accb[8];   // WS_ABLATE & 256 only
    // bias + timestep embedding of a tile's channels: the accumulators START from it (srcC of the tile's first MFMAs), so the
    // epilogue has no add.  The loaders stage the 128 values of the NEXT tile in LDS (cbuf) during the current tile's last chunk.
    f32x4 cin[NI];
    int cpar = 0;
    auto read_cinit = [&]() {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) cin[ni] = *reinterpret_cast<const f32x4*>(cbuf + cpar * CBUF + (wn * 64 + ni * 16 + lq * 4) * 4);
      cpar ^= 1;
    };
    STAMP_DECL
    CLK_DECL
    // row state of the stream of kernel rows (continuous across chunks and tiles)
    int ky = 0, plane = 0, sel = 0;   // the fragment addresses move by wave-uniform steps (no per-lane base registers kept)
    uint32_t rowc = 0;
    uint32_t pq0 = 0, pq1 = 0, pq2 = 0, pq3 = 0;   // producer counters read one half-tap before they are needed                // row of the stream this wave multiplies
    const bool gate512 = p.N > 0;   // always true, opaque to the compiler (WS_ABLATE & 512)
    auto advance_row = [&]() {
      int bstep = 3 * WTILE, astep = AROWB;
      if (sel == NWBUF - 1) { sel = 0; bstep = -(NWBUF - 1) * 3 * WTILE; } else ++sel;
      if (++ky == 3) {
        ky = 0;
        if (plane == NPLANES - 1) { plane = 0; astep = -(NPLANES - 1) * PLANE - 2 * AROWB; } else { ++plane; astep = PLANE - 2 * AROWB; }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) a_cur[kx] += astep;
      b_cur += bstep;
    };
    auto read_a = [&](auto bufc, auto halfc, auto kxc) {
      constexpr int buf = decltype(bufc)::value, half = decltype(halfc)::value, kx = decltype(kxc)::value;
      if constexpr (WS_ABLATE & 4) return;
      if constexpr (WS_ABLATE & 512) { if (gate512) return; }   // timing experiment: MFMAs on stale fragments, no LDS reads
#pragma unroll
      for (int j = 0; j < 4; ++j) af[buf][j] = *reinterpret_cast<const u32x4*>(pbuf + a_cur[kx] + (half * 4 + j) * AROWB);
    };
    auto read_b = [&](auto bufc, auto kxc) {
      constexpr int buf = decltype(bufc)::value, kx = decltype(kxc)::value;
      if constexpr (WS_ABLATE & 4) return;
      if constexpr (WS_ABLATE & 512) { if (gate512) return; }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[buf][ni] = *reinterpret_cast<const u32x4*>(wbuf + b_cur + kx * WTILE + ni * 1024);
    };
    // One half-tap: issue the LDS reads of the NEXT step, then the 16 MFMAs of this one.
    //   R: row parity inside the unrolled row pair, s = 2 * kx + half, ZERO: first tap of a tile (accumulators start at bias + emb)
    auto step = [&](auto Rc, auto sc, auto zeroc, bool more_rows) {
      constexpr int R = decltype(Rc)::value, s = decltype(sc)::value;
      constexpr bool ZERO = decltype(zeroc)::value != 0;
      constexpr int half = s & 1, kx = s >> 1, bpar = (R + kx) & 1;
      if constexpr (s < 5) {
        constexpr int s1 = s + 1, half1 = s1 & 1, kx1 = s1 >> 1;
        read_a(IC<half1>(), IC<half1>(), IC<kx1>());
        if constexpr (half1 == 0) read_b(IC<(R + kx1) & 1>(), IC<kx1>());
        if constexpr (s == 4 && !(WS_ABLATE & 2048)) poll_issue(c_prod, pq0, pq1, pq2, pq3);
      } else {
        if (more_rows) {
          STAMP(6)
          // every read of this row has been issued (LDS performs the add after them); the next kernel row is staged
          if constexpr (!(WS_ABLATE & 2048)) release_acquire(c_cons + wave, c_prod, rowc + 2u, pq0, pq1, pq2, pq3);
          ++rowc;
          STAMP(5)
          advance_row();
          read_a(IC<0>(), IC<0>(), IC<0>());
          read_b(IC<(R + 1) & 1>(), IC<0>());
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((WS_ABLATE & 256) != 0 && Elem<T>::DTYPE == 1) {
        // timing experiment: the same fragment traffic feeding half as many v_mfma_f32_32x32x16_bf16 (8 per step, 32 cycles each,
        // holding the issue port 8 of 32 cycles instead of 8 of 16); the numbers produced are meaningless
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            if constexpr (ZERO) accb[half * 4 + j] = f32x16_t{};
            accb[half * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[bpar][2 * h2]), __builtin_bit_cast(bf16x8, af[half][j]), accb[half * 4 + j], 0, 0, 0);
          }
      } else if constexpr (!(WS_ABLATE & 4)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (ZERO) acc[half * 4 + j][ni] = cin[ni];
            mma16(acc[half * 4 + j][ni], bf[bpar][ni], af[half][j], T());   // D rows = channels, cols = pixels
          }
      } else if constexpr (ZERO) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[half * 4 + j][ni] = cin[ni];
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (!(WS_ABLATE & 2048)) wait_ge_prod(c_prod, 1u);               // kernel row 0 of the first tile is staged
    STAMP(5)
    read_a(IC<0>(), IC<0>(), IC<0>());
    read_b(IC<0>(), IC<0>());
    for (int t = t_first; t < ntp;) {
      const int t_next = next_valid(t);
      int mt, nt, n0, y0, x0;
      decode(t, mt, nt);
      origin(mt, n0, y0, x0);
      const int co_w = nt * BN + wn * 64 + 4 * lq;
      for (int r = 0; r < ngr; r += 2) {
        const bool last_pair = r + 2 >= ngr;
        if (r == 0) { read_cinit(); step(IC<0>(), IC<0>(), IC<1>(), true); step(IC<0>(), IC<1>(), IC<1>(), true); }
        else { step(IC<0>(), IC<0>(), IC<0>(), true); step(IC<0>(), IC<1>(), IC<0>(), true); }
        step(IC<0>(), IC<2>(), IC<0>(), true); step(IC<0>(), IC<3>(), IC<0>(), true);
        step(IC<0>(), IC<4>(), IC<0>(), true); step(IC<0>(), IC<5>(), IC<0>(), true);
        step(IC<1>(), IC<0>(), IC<0>(), true); step(IC<1>(), IC<1>(), IC<0>(), true);
        step(IC<1>(), IC<2>(), IC<0>(), true); step(IC<1>(), IC<3>(), IC<0>(), true);
        step(IC<1>(), IC<4>(), IC<0>(), true); step(IC<1>(), IC<5>(), IC<0>(), !last_pair);
      }
      // End of a tile: the last row is released before the epilogue (the loaders keep running through it), but the first fragments
      // of the next tile are read AFTER it - held across the epilogue they were 32 more live registers at the kernel's pressure peak
      // (scratch spills inside the row loop); a tile start now exposes one LDS round trip per 12-48 rows instead.
      if (t_next < ntp) bump(c_cons + wave);
      if constexpr ((WS_ABLATE & 256) != 0) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{accb[mi][4 * ni], accb[mi][4 * ni + 1], accb[mi][4 * ni + 2], accb[mi][4 * ni + 3]};
      }
      STAMP(6)
      // ---- epilogue: MFMA rows are channels and columns are pixels, so lane (lr, lq) holds 4 consecutive channels of
      // pixel lr per 16x16 tile; bf16 pairs of channel tiles are merged into 16-byte stores by two v_permlane16_swap ----
      const int co_s = PAIR ? nt * BN + wn * 64 + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;
      GnPartial<NI> gp;
      const bool do_gn = p.gn_stats != nullptr;
      const bool gn_mask = ((p.Wo | p.Ho) & 15) != 0;   // partial 16x16 tiles exist: out-of-image pixels must not count
      // HAS_RES / GNM (0 = no statistics, 1 = statistics, 2 = statistics with out-of-image pixels masked) are compile-time inside one
      // copy of the epilogue and chosen by wave-uniform branches outside it: as run-time flags inside the loops they became selects
      // and zero-adds (44 VALU instructions per 16-byte store instead of about 20; the epilogue is 13 % of a 12-row tile)
      auto epi_half = [&](auto hc, auto resc, auto gnc) {
        constexpr int h = decltype(hc)::value, GNM = decltype(gnc)::value;
        constexpr bool HAS_RES = decltype(resc)::value != 0;
        uint32_t ovo[4], rvo[4];
        float vm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int y = y0 + wm * 8 + h * 4 + j, x = x0 + lr;
          const bool ok = y < p.Ho && x < p.Wo && co_s < p.Cout;
          vm[j] = (y < p.Ho && x < p.Wo) ? 1.f : 0.f;
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
      if (p.res_mode != RES_NONE) {
        if (!do_gn) epi(IC<1>(), IC<0>()); else if (!gn_mask) epi(IC<1>(), IC<1>()); else epi(IC<1>(), IC<2>());
      } else {
        if (!do_gn) epi(IC<0>(), IC<0>()); else if (!gn_mask) epi(IC<0>(), IC<1>()); else epi(IC<0>(), IC<2>());
      }
      if (do_gn) {   // slot = (pixel tile of the image, 8-row half); quads of this wave's 64 channels
        const int rem = mt - n0 * tpi;
        gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * 2 + wm) * (size_t)(p.Cout >> 2) + ((nt * BN + wn * 64) >> 2)) * 2, lq, lr);
      }
      STAMP(7)
      if (t_next < ntp) {
        if constexpr (!(WS_ABLATE & 2048)) wait_ge_prod(c_prod, rowc + 2u);
        ++rowc;
        advance_row();
        read_a(IC<0>(), IC<0>(), IC<0>());
        read_b(IC<0>(), IC<0>());
      }
      t = t_next;
    }
    STAMP_FLUSH
    CLK_FLUSH
  }
}

// Shapes the warp-specialised kernel takes (everything else stays on the plain kernel): the plain tile choice already was the
// largest one (>= 512 workgroups of 128 x 128), 3x3 / stride 1 / NHWC output, an even number of 64-byte channel chunks
// (row pairs and chunk pairs are unrolled), images of at least one 16 x 16 tile, and at least one tile per CU.
static int ws_num_cus() {
  static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; } return n; }();
  return ncu;
}
static bool ws_eligible(int enabled, int ks, int BM, int BN, int G, int bn_pack, int out_mode, int stride, int nchunks, int N, int Ho, int Wo, int Cout) {
  if (!enabled || ks != 3 || BM != 128 || BN != 128 || G != 1 || bn_pack != 128 || out_mode != OUT_NHWC) return false;
  if (stride != 1 || nchunks < 2 || (nchunks & 1)) return false;
  if (Wo < ws::VW || Ho < ws::TH) return false;
  const int n_mt = N * ((Wo + ws::VW - 1) / ws::VW) * ((Ho + ws::TH - 1) / ws::TH), n_nt = (Cout + 127) / 128;
  return n_mt * n_nt >= ws_num_cus();   // fewer tiles than CUs: the plain kernel's smaller tiles fill the chip better
}

// 0 = launched, 1 = not eligible (caller uses the plain kernel), < 0 = error
template <typename T>
int launch_ws(ConvKArgs a, int enabled, int BM, int BN, int ks, hipStream_t s) {
  if (!ws_eligible(enabled, ks, BM, BN, a.G, a.bn_pack, a.out_mode, a.stride, a.nchunks, a.N, a.Ho, a.Wo, a.Cout)) return 1;
  a.lvw = 4; a.lth = 4; a.PW = ws::PW; a.PH = ws::PH; a.NP = ws::NPX;
  a.tiles_x = (a.Wo + ws::VW - 1) / ws::VW; a.tiles_y = (a.Ho + ws::TH - 1) / ws::TH;
  const int n_mt = a.N * a.tiles_x * a.tiles_y, n_nt = (a.Cout + 127) / 128;
  const int ncu = ws_num_cus();
  using KernT = void (*)(ConvKArgs, int, int);
  KernT kern = !a.pro_a ? (KernT)conv3x3_ws_kernel<T, 0> : (a.pro_silu ? (KernT)conv3x3_ws_kernel<T, 2> : (KernT)conv3x3_ws_kernel<T, 1>);
  int rc;
  if (!a.pro_a) rc = mi355_allow_big_lds(conv3x3_ws_kernel<T, 0>, "conv3x3 (persistent)");
  else if (a.pro_silu) rc = mi355_allow_big_lds(conv3x3_ws_kernel<T, 2>, "conv3x3 (persistent)");
  else rc = mi355_allow_big_lds(conv3x3_ws_kernel<T, 1>, "conv3x3 (persistent)");
  if (rc) return rc;
  const int ntp = ((n_mt + 7) / 8) * 8 * n_nt;
  const int grid = ntp < ncu ? ntp : ncu;   // one persistent workgroup per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ws::LDS_BYTES, s, a, n_mt, n_nt);
  return 0;
}
