// Implicit-GEMM convolution (3x3 / 1x1, stride 1 / 2, fused nearest-up / avg-pool gather) on MFMA.
//
// Replaces, for inference, every conv2d / conv1d(1) of the reference U-Net together with the ops
// the reference runs around it as separate eager kernels (AD/image_diffusion/unet.py):
//   GroupNorm32-apply + SiLU (+ FiLM)   unet.py:283-286,306-311,343-350  -> prologue while staging
//   F.interpolate(nearest x2) / AvgPool unet.py:209,236,332-337          -> gather mode
//   th.cat([h, hs.pop()], 1)            unet.py:725                      -> two-source K loop
//   h + emb_out, skip + h, x + proj     unet.py:349,351,401              -> epilogue
//
// GEMM view: M = output pixels (BM per workgroup: a rectangle of one image, or several whole small
// images), N = output channels (BN per workgroup), K = taps x input channels, walked as
// (64-byte channel chunk) x (kernel row) x (kernel column).
//   * Per chunk the haloed input patch is staged ONCE into LDS ([patch pixel][64 B], rows padded to
//     96 B so the 16-lane ds_read_b128 groups of every tap shift are bank-conflict free) and re-read by
//     all 9 taps at shifted row addresses.  1x1 convs stage three chunks as three "taps".
//   * Weights arrive pre-tiled / pre-swizzled from the host packer, so staging is a linear copy.
//   * Software pipeline (register staging, T14 style): the NEXT chunk's patch fragments and the NEXT
//     kernel row's weight tiles are loaded into registers while the current row's MFMAs run; weight
//     tiles are double-buffered in LDS, so there is one barrier per kernel row (48 MFMAs per wave).
// 4 waves per workgroup, v_mfma_f32_16x16x32_bf16 (or v_mfma_f32_16x16x4_f32 in the fp32 build).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ops.h"

namespace {

constexpr int PROW = 96;   // bytes per patch pixel row in LDS (64 data + 32 pad)

struct ConvKArgs {
  const void* src0; const void* src1;
  int C0, C1, Cin, nchunks;          // nchunks: 64-byte K chunks of the packed weights (2 x nreal with hi / lo split weights, else nreal)
  int nreal;                         // 64-byte channel chunks of the activations: weight chunk c contracts with activation chunk c mod nreal
  int N, Hs, Ws, Hc, Wc, Ho, Wo;
  int mode, pad, stride;
  const float* pro_a; const float* pro_b; int pro_silu;
  const void* w; const float* bias; int Cout; int bn_pack;
  const float* emb; int emb_stride;
  const void* res; int res_mode; int Hr, Wr;
  void* out; int out_mode;
  float* gn_stats; int gn_slots;     // fused GroupNorm partial sums of the output (common.h GnPartial), or null
  int lvw, lth, G, PW, PH, NP, tiles_x, tiles_y;
  uint32_t bytes0, bytes1, wbytes, obytes, rbytes;   // buffer sizes (raw buffer descriptors: out-of-range loads return 0, stores drop)
  unsigned long long* dbg;           // diagnostic build only (-DCONV_STAMPS): per-phase cycle sums
  int ablate;                        // diagnostic build only: 1 = no output stores, 2 = no prologue math, 4 = no MFMA
  int stagger;                       // odd-slot workgroup start delay in units of s_sleep(127) (~8k cycles each)
  uint32_t* err;                     // error word (device-visible): bit 0 = a counter wait of the persistent kernel expired
  int spin_limit;                    // polls before such a wait gives up
  // small-level kernel only (conv_small.inc.h): the GroupNorm site that reads this conv's output is applied in the epilogue - a wave
  // holds whole images x 64 channels = whole groups, so act_out = silu?(GN(out)) needs no pass of its own; out itself is written
  // only if somebody else reads it (act_raw); one touch of the NEXT conv's packed weights (the pass's L2 warm-up moves here too)
  void* act_out; const float* act_gamma; const float* act_beta; const float* act_film; int act_film_stride; float act_eps;
  int act_silu, act_raw;
  int act_stride, act_coff, act_cpg; uint32_t abytes;   // act_out: channels per pixel, first channel this conv fills, channels per group, bytes
  void* act2_out; const float* act2_gamma; const float* act2_beta; int act2_silu, act2_stride, act2_coff, act2_cpg; uint32_t a2bytes;   // a second site (no FiLM)
  const void* warm; uint32_t warm_bytes;
  // small-level kernel only: a 1x1 conv of cat(sk0, sk1) (SC0 + SC1 channels at the output resolution) accumulated into the same output - the
  // ResBlock's skip_connection inside its second conv; w then holds, per 128-channel pack tile, the 3x3 tiles followed by one tile per skip chunk
  const void* sk0; const void* sk1; int SC0, SC1; uint32_t skbytes0, skbytes1;
  uint32_t wstride;                  // 8-KB weight tiles per 128-channel pack tile (9 nchunks without a fused skip conv)
};

#ifdef CONV_STAMPS
// In-kernel phase stamps (cdna_hip_programming.md section 7): diagnostic build only, never the timed build.
__device__ __forceinline__ unsigned long long conv_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP_DECL unsigned long long st_prev = conv_stamp(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k) { const unsigned long long st_now = conv_stamp(); st_acc[k] += st_now - st_prev; st_prev = st_now; }
#define STAMP_FLUSH if (p.dbg && (threadIdx.x & 63) == 0) { unsigned long long* q = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6)) * 8; for (int k = 0; k < 8; ++k) q[k] = st_acc[k]; }
// In-kernel clock of a wave (MI355X_MICROARCH.md, DVFS give-back item 6): shader-clock cycles per 100-MHz reference tick over the wave's
// life time; two u64 per wave in the 64 KB in FRONT of the stamp buffer (the persistent conv's consumer waves only).
#define CLK_DECL const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#define CLK_FLUSH if (p.dbg && (threadIdx.x & 63) == 0) { unsigned long long* q = p.dbg - 8192 + ((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 2; q[0] = __builtin_amdgcn_s_memtime() - ck_t0; q[1] = __builtin_amdgcn_s_memrealtime() - ck_r0; }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#define CLK_DECL
#define CLK_FLUSH
#endif

template <int I> struct IC { static constexpr int value = I; };

// activation chunk of weight chunk c (hi / lo split weights: the second half of the K loop runs over the same activations again)
__device__ __forceinline__ int src_chunk(const ConvKArgs& p, int c) { return c >= p.nreal ? c - p.nreal : c; }

template <bool FAST> __device__ __forceinline__ float silu_fast(float v) {
  if (FAST) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  return v / (1.0f + expf(-v));
}

__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// KS: 3 (3x3) or 1 (1x1: three channel chunks play the role of the three taps of a kernel row)
// PIT: 16-B patch fragments per thread per plane (compile-time bound of the staging loops)
// MULTI: the tile spans several (small) images, so the prologue's (a, b) differ per patch fragment
template <typename T, int BM, int BN, int WM, int WN, int KS, int PIT, bool MULTI>
__global__ void __launch_bounds__(64 * WM * WN, WM * WN / 2) conv_igemm_kernel(ConvKArgs p) {
  constexpr int NTHREADS = 64 * WM * WN;     // 4 or 8 waves; two workgroups per CU either way
  constexpr int FR = NTHREADS / 4;          // patch pixels covered by one fragment sweep of the workgroup
  using E = Elem<T>;
  constexpr int V = E::VEC, CHUNK = E::CHUNK, ESZ = sizeof(T);
  constexpr bool FAST = (E::DTYPE == 1);
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int WTILE = BN * 64;             // bytes of one (chunk, tap) weight tile of this workgroup
  constexpr int WIT = (3 * WTILE + NTHREADS * 16 - 1) / (NTHREADS * 16);  // 16-B weight fragments per thread per kernel row
  constexpr int NPL = KS == 1 ? 3 : 1;       // patch planes
  constexpr int WD = (KS == 3 && BM == 64) ? 3 : 1;   // weight prefetch distance in kernel rows (register ring): a 64-pixel tile's row is
                                                      // 24 MFMAs per wave, far shorter than an L2 round trip (PMC: 42 % of wave cycles parked)
  constexpr int PLANE = PIT * FR * PROW;     // bytes per plane (every thread owns PIT fragment slots: no bounds checks)
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* wlds = smem + NPL * PLANE;           // two buffers of 3*WTILE

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lq = lane >> 4;

  const int nt = blockIdx.y;
  const int tpi = p.tiles_x * p.tiles_y;
  const int ng = blockIdx.x / tpi, rem = blockIdx.x - ng * tpi;
  const int tyi = rem / p.tiles_x, txi = rem - tyi * p.tiles_x;
  const int n0 = ng * p.G, y0 = tyi << p.lth, x0 = txi << p.lvw;
  const int VWm = (1 << p.lvw) - 1, THm = (1 << p.lth) - 1;
  const int pimg = p.PH * p.PW;

  // wave-uniform buffer descriptors (kernel arguments only): 32-bit per-lane offsets, OOB -> 0, no 64-bit address math
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src0), 0, p.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src1 ? p.src1 : p.src0), 0, p.bytes1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- per-thread patch fragments: fragment u of this thread is patch pixel (tid>>2) + 64u, 16-B slot tid&3 ----
  const int fq = tid & 3, frow = tid >> 2;
  int sidx[PIT];        // source pixel index, -1 = zero padding / out of range
  uint32_t voff[PIT];   // byte offset of the fragment in the CURRENT source tensor (>= size when invalid)
  uint32_t vmask = 0;
  bool vsrc0 = true;    // voff currently addresses the first source
  {
    const int cy0 = y0 * p.stride - p.pad, cx0 = x0 * p.stride - p.pad;
    const float inv_pimg = 1.0f / (float)pimg, inv_pw = 1.0f / (float)p.PW;
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
      const int i = frow + u * FR;
      int s = -1;
      if (i < p.NP) {
        // i < 65536 and the divisors are small: floor((i + 0.5) / d) in fp32 is exact and costs 3 instructions
        const int g = (int)(((float)i + 0.5f) * inv_pimg), r = i - g * pimg;
        const int py = (int)(((float)r + 0.5f) * inv_pw), px = r - py * p.PW;
        const int n = n0 + g, cy = cy0 + py, cx = cx0 + px;
        if (n < p.N && cy >= 0 && cy < p.Hc && cx >= 0 && cx < p.Wc) {
          if (p.mode == CONV_UP2) s = (n * p.Hs + (cy >> 1)) * p.Ws + (cx >> 1);
          else s = (n * p.Hs + cy) * p.Ws + cx;
        }
      }
      sidx[u] = s;
      voff[u] = s >= 0 ? (uint32_t)s * (uint32_t)(p.C0 * ESZ) + fq * 16 : p.bytes0;
      if (s >= 0) vmask |= 1u << u;
    }
  }

  int arow[MI], brow[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + lr;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, g = m >> (p.lvw + p.lth);
    arow[mi] = ((g * p.PH + ty * p.stride) * p.PW + tx * p.stride) * PROW + lq * 16;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * WTN + ni * 16 + lr;
    brow[ni] = row * 64 + 16 * (lq ^ ((row >> 1) & 3));
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // packed weights: [nt_pack][chunk][tap][bn_pack rows][64 B]; this workgroup owns rows sub*BN .. of nt_pack
  const int per = p.bn_pack / BN;
  const uint32_t wtile_pack = p.bn_pack * 64;
  const uint32_t wbase = (uint32_t)(nt / per) * p.nchunks * (KS * KS) * wtile_pack + (uint32_t)(nt % per) * WTILE;
  uint32_t woff[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const uint32_t off = (i * NTHREADS + tid) * 16;
    const uint32_t tile = off / WTILE, inner = off - tile * WTILE;
    woff[i] = off < 3 * WTILE ? tile * wtile_pack + inner : p.wbytes;
  }
#ifdef CONV_STAMPS
  const int pro = (p.pro_a == nullptr || (p.ablate & 2)) ? 0 : (p.pro_silu ? 2 : 1);
#else
  const int pro = p.pro_a == nullptr ? 0 : (p.pro_silu ? 2 : 1);
#endif

  u32x4 raw[NPL][PIT];
  u32x4 wreg[WD][WIT];
  constexpr int NAB = MULTI ? PIT : 1;   // (a, b) register sets per plane
  float pa[NPL][NAB][V], pb[NPL][NAB][V];

  // issue the global loads of chunk c into plane register set PL (no waits).  PL is a compile-time constant and
  // no pointer to a register array is ever formed: either would push the arrays into scratch memory.
  auto prefetch_patch = [&](int c, auto plc) {
    constexpr int pl = decltype(plc)::value;
#ifdef CONV_STAMPS
    if (p.ablate & 16) return;
#endif
    const int cb = src_chunk(p, c) * CHUNK;
    {
      const bool first = cb < p.C0;
      if (first != vsrc0) {   // the chunk stream enters the other source (skip concat; with hi / lo split weights it comes back to the first): re-base the fragment offsets
        vsrc0 = first;
#pragma unroll
        for (int u = 0; u < PIT; ++u) voff[u] = sidx[u] >= 0 ? (uint32_t)sidx[u] * (uint32_t)((first ? p.C0 : p.C1) * ESZ) + fq * 16 : (first ? p.bytes0 : p.bytes1);
      }
      const uint32_t so = (first ? cb : cb - p.C0) * ESZ;
      if (first) {
#pragma unroll
        for (int u = 0; u < PIT; ++u) raw[pl][u] = buf_load16(rs0, voff[u], so);
      } else {
#pragma unroll
        for (int u = 0; u < PIT; ++u) raw[pl][u] = buf_load16(rs1, voff[u], so);
      }
    }
    if (pro) {   // prologue affine of this chunk's channels, prefetched with the data (never loaded inside commit_patch:
                 // a load there would force the compiler to drain vmcnt, i.e. to wait for the weight prefetch too)
#pragma unroll
      for (int k = 0; k < NAB; ++k) {
        int n = n0;
        if (MULTI) n = min(n0 + (int)(((float)(frow + k * FR) + 0.5f) * (1.0f / (float)pimg)), p.N - 1);
        const float* ap = p.pro_a + (size_t)n * p.Cin + cb + fq * V;
        const float* bp = p.pro_b + (size_t)n * p.Cin + cb + fq * V;
#pragma unroll
        for (int j = 0; j < V; ++j) { pa[pl][k][j] = ap[j]; pb[pl][k][j] = bp[j]; }
      }
    }
  };
  // transform + write chunk c (register set PL) into LDS plane PL
  auto commit_patch = [&](int c, auto plc) {
    constexpr int pl = decltype(plc)::value;
#ifdef CONV_STAMPS
    if (p.ablate & 16) return;
#endif
    char* dst = patch + pl * PLANE + frow * PROW + fq * 16;
    if (pro == 0) {
#pragma unroll
      for (int u = 0; u < PIT; ++u) *reinterpret_cast<u32x4*>(dst + u * FR * PROW) = raw[pl][u];
      return;
    }
#pragma unroll
    for (int u = 0; u < PIT; ++u) {
      float av[V], bv[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { av[j] = pa[pl][MULTI ? u : 0][j]; bv[j] = pb[pl][MULTI ? u : 0][j]; }
      float f[V];
      auto xform = [&](const u32x4& rw, float (&o)[V]) {
        frag_to_float(rw, o, T());
        if (pro == 2) {
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = silu_fast<FAST>(av[j] * o[j] + bv[j]);
        } else if (pro == 1) {
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = av[j] * o[j] + bv[j];
        }
      };
      xform(raw[pl][u], f);
      u32x4 outv = float_to_frag(f, T());
      if (!((vmask >> u) & 1u)) outv = u32x4{0u, 0u, 0u, 0u};   // zero padding applies AFTER the prologue
      *reinterpret_cast<u32x4*>(dst + u * FR * PROW) = outv;
    }
  };
  // weight tiles of one kernel row: three consecutive (chunk, tap) tiles starting at linear tile index t0
  auto prefetch_w = [&](int t0, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
#ifdef CONV_STAMPS
    if (p.ablate & 8) return;
#endif
    const uint32_t so = wbase + (uint32_t)t0 * wtile_pack;
#pragma unroll
    for (int i = 0; i < WIT; ++i) wreg[slot][i] = buf_load16(rsw, woff[i], so);
  };
  auto commit_w = [&](int buf, auto slotc) {
    constexpr int slot = decltype(slotc)::value;
#ifdef CONV_STAMPS
    if (p.ablate & 8) return;
#endif
    char* dst = wlds + buf * (3 * WTILE) + tid * 16;
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * NTHREADS * 16 <= 3 * WTILE || (i * NTHREADS + tid) * 16 < 3 * WTILE)
        *reinterpret_cast<u32x4*>(dst + i * NTHREADS * 16) = wreg[slot][i];
  };
  auto mma_tap = [&](const char* pa_, const int (&ao)[MI], int aimm, const char* wt, const int (&bo)[NI]) {
    u32x4 a[MI], b[NI];
#ifdef CONV_STAMPS
    if (p.ablate & 32) return;
#endif
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(pa_ + ao[mi] + aimm);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(wt + bo[ni]);
#ifdef CONV_STAMPS
    if (p.ablate & 4) { asm volatile("" ::"v"(a[0]), "v"(b[0])); return; }
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], b[ni], a[mi], T());  // D rows = channels, cols = pixels
    __builtin_amdgcn_s_setprio(0);
  };

  // The three taps of one kernel row with the fragment reads of tap kx + 1 issued before the MFMAs of tap kx (register double
  // buffer): at one or two waves per SIMD nothing else hides the LDS latency between a tap's reads and its MFMAs.
  auto mma_row3 = [&](const char* pa_, const int (&ao)[MI], const char* wt, const int (&bo)[NI]) {
#ifdef CONV_STAMPS
    if (p.ablate & (32 | 4)) { mma_tap(pa_, ao, 0, wt, bo); mma_tap(pa_, ao, PROW, wt + WTILE, bo); mma_tap(pa_, ao, 2 * PROW, wt + 2 * WTILE, bo); return; }
#endif
    u32x4 a[2][MI], b[2][NI];
    auto rd = [&](auto bufc, auto kxc) {
      constexpr int buf = decltype(bufc)::value, kx = decltype(kxc)::value;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[buf][mi] = *reinterpret_cast<const u32x4*>(pa_ + ao[mi] + kx * PROW);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[buf][ni] = *reinterpret_cast<const u32x4*>(wt + kx * WTILE + bo[ni]);
    };
    auto mm = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma16(acc[mi][ni], b[buf][ni], a[buf][mi], T());
      __builtin_amdgcn_sched_barrier(0);
    };
    rd(IC<0>(), IC<0>());
    rd(IC<1>(), IC<1>());
    mm(IC<0>());
    rd(IC<0>(), IC<2>());
    mm(IC<1>());
    mm(IC<0>());
  };

  int gi = 0;  // kernel-row counter (weight buffer parity)
  // Two workgroups share a CU and would otherwise run their staging / MFMA / store phases in lockstep (measured: the
  // phase costs simply add up).  Delaying the workgroup in the odd threadgroup slot of the CU (HW_ID.TG_ID, bits 19:16)
  // by about half a chunk period puts one workgroup's MFMA rows beside the other's VALU/LDS/memory phases.
  if (p.stagger > 0 && ((__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (16 << 6) | 4) & 1) != 0)) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  STAMP_DECL
  if constexpr (KS == 3) {
    int a1[MI], a2[MI];   // fragment row offsets of kernel rows 1 and 2 (row 0 = arow); kx is an immediate offset
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { a1[mi] = arow[mi] + p.PW * PROW; a2[mi] = arow[mi] + 2 * p.PW * PROW; }
    prefetch_patch(0, IC<0>());
    prefetch_w(0, IC<0>());
    if constexpr (WD == 3) { prefetch_w(3, IC<1>()); prefetch_w(6, IC<2>()); }
    STAMP(0)
    for (int c = 0; c < p.nchunks; ++c) {
      if (c > 0) __syncthreads();           // every wave has finished reading the previous chunk's patch
      STAMP(2)
      commit_patch(c, IC<0>());
      STAMP(1)
      if (c + 1 < p.nchunks) prefetch_patch(c + 1, IC<0>());
      STAMP(5)
      auto row = [&](auto kyc) {
        constexpr int ky = decltype(kyc)::value;
        constexpr int slot = WD == 3 ? ky : 0;
        commit_w(gi & 1, IC<slot>());
        STAMP(3)
        __syncthreads();                    // patch + this row's weights visible; also orders weight-buffer reuse
        STAMP(4)
        const int nxt = WD == 3 ? (c + 1) * 9 + ky * 3 : c * 9 + (ky + 1) * 3;
        if (nxt < p.nchunks * 9) prefetch_w(nxt, IC<slot>());
        STAMP(5)
        const char* wt = wlds + (gi & 1) * (3 * WTILE);
        if (ky == 0) mma_row3(patch, arow, wt, brow);
        else if (ky == 1) mma_row3(patch, a1, wt, brow);
        else mma_row3(patch, a2, wt, brow);
        STAMP(6)
        ++gi;
      };
      row(IC<0>()); row(IC<1>()); row(IC<2>());
    }
  } else {
    const int ngroups = (p.nchunks + 2) / 3;
    auto group_n = [&](int g) { return min(3, p.nchunks - 3 * g); };
    auto prefetch_group = [&](int g) {
      const int nn = group_n(g);
      prefetch_patch(3 * g, IC<0>());
      if (nn > 1) prefetch_patch(3 * g + 1, IC<1>());
      if (nn > 2) prefetch_patch(3 * g + 2, IC<2>());
      prefetch_w(3 * g, IC<0>());
    };
    prefetch_group(0);
    for (int g = 0; g < ngroups; ++g, ++gi) {
      const int nc = group_n(g);
      if (g > 0) __syncthreads();
      commit_patch(3 * g, IC<0>());
      if (nc > 1) commit_patch(3 * g + 1, IC<1>());
      if (nc > 2) commit_patch(3 * g + 2, IC<2>());
      commit_w(gi & 1, IC<0>());
      __syncthreads();
      if (g + 1 < ngroups) prefetch_group(g + 1);
      const char* wt = wlds + (gi & 1) * (3 * WTILE);
      mma_tap(patch, arow, 0, wt, brow);
      if (nc > 1) mma_tap(patch + PLANE, arow, 0, wt + WTILE, brow);
      if (nc > 2) mma_tap(patch + 2 * PLANE, arow, 0, wt + 2 * WTILE, brow);
    }
  }

  // ---------------- epilogue, straight from the accumulators ----------------
  // MFMA rows are output channels and columns are pixels, so lane (lr, lq) holds, per 16x16 tile, the 4 consecutive
  // channels 4*lq..4*lq+3 of pixel lr: one 8-byte (bf16) / 16-byte (fp32) NHWC store per tile, no LDS round trip.
  // Every load of the epilogue (bias, emb, residual) is issued before the first use, and invalid pixels are steered
  // out of the buffer range (loads return 0, stores are dropped), so there is no branch and no per-tile latency chain.
  const int co_w = nt * BN + wn * WTN + 4 * lq;   // first output channel of this lane
  int pn[MI], py_[MI], px_[MI]; bool pvalid[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = wm * WTM + mi * 16 + lr;
    const int tx = m & VWm, ty = (m >> p.lvw) & THm, g = m >> (p.lvw + p.lth);
    pn[mi] = n0 + g; py_[mi] = y0 + ty; px_[mi] = x0 + tx;
    pvalid[mi] = pn[mi] < p.N && py_[mi] < p.Ho && px_[mi] < p.Wo;
  }
  if (p.out_mode == OUT_NHWC) {
    // bf16: the natural piece is 8 bytes per lane; two v_permlane16_swap per pair of channel tiles turn it into 16 bytes
    // (8 consecutive channels) per lane, halving the store / residual-load instruction count (the store tail is
    // issue-bound).  After swap(X = tile a, Y = tile b): 16-lane row 0 holds tile a ch 0-7, row 1 tile b ch 0-7,
    // row 2 tile a ch 8-15, row 3 tile b ch 8-15 (probed on gfx950: tools/probe/permlane_probe.cpp); the swap is an involution.
    constexpr bool PAIR = E::DTYPE == 1;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.out), 0, p.rbytes, 0x00020000);
    const int co_s = PAIR ? nt * BN + wn * WTN + (lq & 1) * 16 + (lq >> 1) * 8 : co_w;   // first channel this lane stores
    uint32_t ovo[MI], rvo[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const uint32_t opix = (uint32_t)((pn[mi] * p.Ho + py_[mi]) * p.Wo + px_[mi]);
      const bool ok = pvalid[mi] && co_s < p.Cout;
      ovo[mi] = ok ? (opix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.obytes;
      uint32_t rpix = opix;
      if (p.res_mode == RES_UP2) rpix = (uint32_t)((pn[mi] * p.Hr + (py_[mi] >> 1)) * p.Wr + (px_[mi] >> 1));
      rvo[mi] = (ok && p.res_mode != RES_NONE) ? (rpix * (uint32_t)p.Cout + (uint32_t)co_s) * ESZ : p.rbytes;
    }
    constexpr int NP2 = PAIR ? NI / 2 : NI;     // 16-byte pieces per pixel row of this wave
    constexpr int PSTEP = PAIR ? 32 : 16;       // channels between consecutive pieces
    u32x4 rr[MI][NP2];
    if (p.res_mode != RES_NONE) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int k = 0; k < NP2; ++k)
          rr[mi][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, rvo[mi] + k * PSTEP * ESZ, 0, 0));
    }
    GnPartial<NI> gp;
    const bool do_gn = !MULTI && p.gn_stats != nullptr;
    const bool gn_mask = ((p.Wo & VWm) | (p.Ho & THm)) != 0;   // partial tiles exist: out-of-image pixels must not count
    f32x4 add4[MULTI ? MI : 1][NI];   // bias + emb per (image, channel quad), original (unswapped) layout
#pragma unroll
    for (int k = 0; k < (MULTI ? MI : 1); ++k)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int co = co_w + ni * 16;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (co < p.Cout) {
          if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + co);
          if (p.emb) {
            const int n = MULTI ? min(pn[k], p.N - 1) : n0;
            const f32x4 ev = *reinterpret_cast<const f32x4*>(p.emb + (size_t)n * p.emb_stride + co);
            v = f32x4{v[0] + ev[0], v[1] + ev[1], v[2] + ev[2], v[3] + ev[3]};
          }
        }
        add4[k][ni] = v;
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if constexpr (!PAIR) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const f32x4 ad = add4[MULTI ? mi : 0][ni];
          f32x4 o = f32x4{acc[mi][ni][0] + ad[0], acc[mi][ni][1] + ad[1], acc[mi][ni][2] + ad[2], acc[mi][ni][3] + ad[3]};
          if (p.res_mode != RES_NONE) {
            const f32x4 t = __builtin_bit_cast(f32x4, rr[mi][ni]);
            o = f32x4{o[0] + t[0], o[1] + t[1], o[2] + t[2], o[3] + t[3]};
          }
          if (do_gn) gp.add(ni, o[0], o[1], o[2], o[3], gn_mask, pvalid[mi] ? 1.f : 0.f);
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
              unpack2(xa[j], ra[2 * j], ra[2 * j + 1], T());
              unpack2(xb[j], rb[2 * j], rb[2 * j + 1], T());
            }
          }
          const f32x4 ada = add4[MULTI ? mi : 0][2 * k], adb = add4[MULTI ? mi : 0][2 * k + 1];
          float va[4], vb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            va[j] = acc[mi][2 * k][j] + ada[j] + ra[j];
            vb[j] = acc[mi][2 * k + 1][j] + adb[j] + rb[j];
          }
          if (do_gn) {
            const float vm = pvalid[mi] ? 1.f : 0.f;
            gp.add(2 * k, va[0], va[1], va[2], va[3], gn_mask, vm);
            gp.add(2 * k + 1, vb[0], vb[1], vb[2], vb[3], gn_mask, vm);
          }
          const u32x2 pa2 = pack4(va, T()), pb2 = pack4(vb, T());
          const auto w0 = __builtin_amdgcn_permlane16_swap(pa2[0], pb2[0], false, false);
          const auto w1 = __builtin_amdgcn_permlane16_swap(pa2[1], pb2[1], false, false);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{w0[0], w1[0], w0[1], w1[1]}, rso, ovo[mi] + k * PSTEP * ESZ, 0, 0);
        }
      }
    }
    if (do_gn)
      gp.store(p.gn_stats + (((size_t)n0 * p.gn_slots + rem * WM + wm) * (size_t)(p.Cout >> 2) + ((nt * BN + wn * WTN) >> 2)) * 2, lq, lr);
  } else {  // OUT_NCHW_F32 (network output): lanes lr are 16 consecutive pixels of a row -> 64-byte fp32 segments
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (!pvalid[mi]) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = co_w + ni * 16 + r;
          if (c < p.Cout)
            reinterpret_cast<float*>(p.out)[(((size_t)pn[mi] * p.Cout + c) * p.Ho + py_[mi]) * p.Wo + px_[mi]] = acc[mi][ni][r] + (p.bias ? p.bias[c] : 0.f);
        }
    }
  }
  STAMP(7)
  STAMP_FLUSH
}

template <typename T, int BM, int BN, int WM, int WN, int KS, int PIT, bool MULTI>
int launch_m(const ConvKArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, KS, PIT, MULTI>;
  if (int rc = mi355_allow_big_lds(kern, "conv")) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, s, a);
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int KS, int PIT>
int launch_t(const ConvKArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  if constexpr (BM == 64) { if (a.G > 1) return launch_m<T, BM, BN, WM, WN, KS, PIT, true>(a, grid, lds, s); }
  if (a.G > 1) { mi355_set_error("conv: multi-image tiles need the 64-pixel tile"); return -4; }
  return launch_m<T, BM, BN, WM, WN, KS, PIT, false>(a, grid, lds, s);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_ks(const ConvKArgs& a, int ks, int pit_t, dim3 grid, size_t lds, hipStream_t s) {
  constexpr int W8 = WM * WN == 8;
  if (ks == 1) return launch_t<T, BM, BN, WM, WN, 1, W8 ? 1 : 2>(a, grid, lds, s);
  if (pit_t == (W8 ? 2 : 4)) return launch_t<T, BM, BN, WM, WN, 3, W8 ? 2 : 4>(a, grid, lds, s);
  if (pit_t == (W8 ? 4 : 7)) return launch_t<T, BM, BN, WM, WN, 3, W8 ? 4 : 7>(a, grid, lds, s);
  if (pit_t == (W8 ? 6 : 11)) return launch_t<T, BM, BN, WM, WN, 3, W8 ? 6 : 11>(a, grid, lds, s);
  mi355_set_error("conv: patch too large");
  return -4;
}

template <typename T>
int launch_cfg(const ConvKArgs& a, int BM, int BN, int ks, int pit, dim3 grid, size_t lds, hipStream_t s) {
  if (BM == 128 && BN == 128) return launch_ks<T, 128, 128, 2, 2>(a, ks, pit, grid, lds, s);
  if (BM == 128 && BN == 64) return launch_ks<T, 128, 64, 2, 2>(a, ks, pit, grid, lds, s);
  if (BM == 128 && BN == 32) return launch_ks<T, 128, 32, 4, 1>(a, ks, pit, grid, lds, s);
  if (BM == 64 && BN == 128) return launch_ks<T, 64, 128, 2, 2>(a, ks, pit, grid, lds, s);
  if (BM == 64 && BN == 64) return launch_ks<T, 64, 64, 2, 2>(a, ks, pit, grid, lds, s);
  if (BM == 64 && BN == 32) return launch_ks<T, 64, 32, 4, 1>(a, ks, pit, grid, lds, s);
  mi355_set_error("conv: unsupported tile");
  return -4;
}

#include "conv_ws.inc.h"
#ifndef PP_IMPL   // experiments: make variant VARFLAGS='-DPP_IMPL="\"../../tools/experiments/<file>\""' VARTAG=_x builds a library around another version of the kernel
#define PP_IMPL "conv_pp.inc.h"
#endif
#include PP_IMPL
#include "conv_pp1.inc.h"
#include "conv_small.inc.h"

struct Geo {
  int Hc, Wc, Ho, Wo, BM, BN, bn_pack, lvw, lth, G, PW, PH, NP, tiles_x, tiles_y, groups, pad, stride, plane_bytes, pit, pit_t;
  size_t lds;
};

int compute_geo(const ConvDesc& d, Geo& g) {
  g.pad = d.ks / 2;
  g.stride = d.mode == CONV_STRIDE2 ? 2 : 1;
  if (d.mode == CONV_UP2) { g.Hc = d.Hs * 2; g.Wc = d.Ws * 2; }
  else if (d.mode == CONV_POOL2) { g.Hc = d.Hs / 2; g.Wc = d.Ws / 2; }
  else { g.Hc = d.Hs; g.Wc = d.Ws; }
  if (d.mode == CONV_STRIDE2) { g.Ho = (g.Hc + 2 * g.pad - d.ks) / 2 + 1; g.Wo = (g.Wc + 2 * g.pad - d.ks) / 2 + 1; }
  else { g.Ho = g.Hc; g.Wo = g.Wc; }
  g.bn_pack = conv_tile_n(d.Cout);
  // Tile choice: the largest tile that still gives every CU two workgroups; otherwise the smallest tile.
  const long M = (long)d.N * g.Ho * g.Wo;
  const int P = g.bn_pack, Q = P > 64 ? 64 : P;
  int cand[4][2]; int nc = 0;
  cand[nc][0] = 128; cand[nc++][1] = P;
  cand[nc][0] = 64; cand[nc++][1] = P;
  if (Q != P) { cand[nc][0] = 64; cand[nc++][1] = Q; }
  int bestBM = cand[nc - 1][0], bestBN = cand[nc - 1][1];
  const long img_slots = (long)(1 << ilog2_ceil(g.Wo)) * (1 << ilog2_ceil(g.Ho));
  const long min_wgs = (d.knobs ? d.knobs : &mi355_default_debug())->conv_min_wgs;
  for (int i = 0; i < nc; ++i) {
    if (cand[i][0] == 128 && img_slots < 128) continue;   // several images per tile: only the 64-pixel tile has that variant
    const long wgs = ((M + cand[i][0] - 1) / cand[i][0]) * ((d.Cout + cand[i][1] - 1) / cand[i][1]);
    if (wgs >= min_wgs) { bestBM = cand[i][0]; bestBN = cand[i][1]; break; }
  }
  g.BM = bestBM; g.BN = bestBN;
  const int lw = ilog2_ceil(g.Wo);
  g.lvw = (1 << lw) > g.BM ? ilog2_ceil(g.BM) : lw;
  const int VW = 1 << g.lvw;
  const int thfull = g.BM / VW;
  if (g.Ho >= thfull) { g.lth = ilog2_ceil(thfull); g.G = 1; }
  else { g.lth = ilog2_ceil(g.Ho); g.G = thfull >> g.lth; }
  const int THp = 1 << g.lth;
  g.tiles_x = (g.Wo + VW - 1) / VW;
  g.tiles_y = (g.Ho + THp - 1) / THp;
  g.groups = (d.N + g.G - 1) / g.G;
  if (g.stride == 2) { g.PW = 2 * VW + 1; g.PH = 2 * THp + 1; }
  else { g.PW = VW + 2 * g.pad; g.PH = THp + 2 * g.pad; }
  g.NP = g.G * g.PH * g.PW;
  const bool w8 = false;   // 8-wave (512-thread) tiles are supported by the kernel template but lose to 4 waves: 128 VGPRs spill
  const int fr = w8 ? 128 : 64;                 // patch pixels per fragment sweep
  g.pit = (g.NP + fr - 1) / fr;
  if (d.ks == 1) g.pit_t = w8 ? 1 : 2;          // template PIT actually launched
  else if (w8) g.pit_t = g.pit <= 2 ? 2 : (g.pit <= 4 ? 4 : 6);
  else g.pit_t = g.pit <= 4 ? 4 : (g.pit <= 7 ? 7 : 11);
  g.plane_bytes = g.pit_t * fr * PROW;
  const int npl = d.ks == 1 ? 3 : 1;
  size_t main_lds = (size_t)npl * g.plane_bytes + 2 * 3 * (size_t)g.BN * 64;
  g.lds = (main_lds + 15) / 16 * 16;
  return 0;
}

}  // namespace

int conv1x1_pp_try_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used) { return pp1_try_launch(d, stream, gn_slots_used); }

int conv_tile_n(int Cout) {
  if (Cout % 128 == 0) return 128;
  if (Cout % 64 == 0) return 64;
  return 32;
}

static inline int chunk_of(int dtype) { return dtype == 0 ? 16 : 32; }

size_t conv_packed_weight_bytes(int dtype, int Cout, int Cin, int ks, int split) {
  const int BN = conv_tile_n(Cout), CH = chunk_of(dtype);
  const size_t nt = (Cout + BN - 1) / BN, nc = (size_t)((Cin + CH - 1) / CH) * (split ? 2 : 1);
  return nt * nc * ks * ks * (size_t)BN * 64;
}

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

void conv_pack_weights(int dtype, const float* w, int Cout, int Cin, int ks, void* dst, int split) {
  const int BN = conv_tile_n(Cout), CH = chunk_of(dtype), V = CH / 4, esz = dtype == 0 ? 4 : 2;
  const int nt = (Cout + BN - 1) / BN, nr = (Cin + CH - 1) / CH, nc = nr * (split && dtype == 1 ? 2 : 1), ntaps = ks * ks;
  char* out = reinterpret_cast<char*>(dst);
  memset(out, 0, conv_packed_weight_bytes(dtype, Cout, Cin, ks, split));
  for (int t = 0; t < nt; ++t)
    for (int c = 0; c < nc; ++c)
      for (int tap = 0; tap < ntaps; ++tap) {
        char* tile = out + (((size_t)t * nc + c) * ntaps + tap) * (size_t)BN * 64;
        const bool lo = c >= nr;               // second half of the K loop: the rounding residual of the first half's weights
        for (int row = 0; row < BN; ++row) {
          const int co = t * BN + row;
          if (co >= Cout) break;
          for (int kl = 0; kl < CH; ++kl) {
            const int ci = (lo ? c - nr : c) * CH + kl;
            if (ci >= Cin) break;
            float v = w[((size_t)co * Cin + ci) * ntaps + tap];
            if (lo) v = v - bf2f(f2bf(v));
            const int q = kl / V, e = kl % V;
            char* dstp = tile + row * 64 + 16 * (q ^ ((row >> 1) & 3)) + e * esz;
            if (dtype == DT_F32) memcpy(dstp, &v, 4);
            else if (dtype == DT_F16) { const _Float16 h = (_Float16)v; memcpy(dstp, &h, 2); }   // RNE
            else { uint16_t h = f2bf(v); memcpy(dstp, &h, 2); }
          }
        }
      }
}

size_t conv_packed_weight_bytes_skip(int dtype, int Cout, int Cin, int Cskip) {
  return conv_packed_weight_bytes(dtype, Cout, Cin, 3, 0) + conv_packed_weight_bytes(dtype, Cout, Cskip, 1, 0);
}
void conv_pack_weights_skip(int dtype, const float* w3, const float* w1, int Cout, int Cin, int Cskip, void* dst) {
  const size_t b3 = conv_packed_weight_bytes(dtype, Cout, Cin, 3, 0), b1 = conv_packed_weight_bytes(dtype, Cout, Cskip, 1, 0);
  std::vector<char> t3(b3), t1(b1);
  conv_pack_weights(dtype, w3, Cout, Cin, 3, t3.data(), 0);
  conv_pack_weights(dtype, w1, Cout, Cskip, 1, t1.data(), 0);
  const size_t nt = (Cout + conv_tile_n(Cout) - 1) / conv_tile_n(Cout), s3 = b3 / nt, s1 = b1 / nt;
  char* out = reinterpret_cast<char*>(dst);
  for (size_t t = 0; t < nt; ++t) {
    memcpy(out + t * (s3 + s1), t3.data() + t * s3, s3);
    memcpy(out + t * (s3 + s1) + s3, t1.data() + t * s1, s1);
  }
}

// Weights of the data-gradient conv (backward w.r.t. the input) of a [Cout][Cin][ks][ks] forward filter: the transposed conv of a
// stride-1, padding ks/2 convolution is the same convolution with input / output channels swapped and the taps flipped,
// W'[ci][co][ky][kx] = W[co][ci][ks-1-ky][ks-1-kx]; rows ci >= Cin (channel padding of the forward input) are zero.
size_t conv_packed_weight_bytes_dgrad(int dtype, int Cout, int Cin, int ks, int cin_pad) { (void)Cin; return conv_packed_weight_bytes(dtype, cin_pad, Cout, ks); }
void conv_pack_weights_dgrad(int dtype, const float* w, int Cout, int Cin, int ks, int cin_pad, void* dst) {
  const int nt = ks * ks;
  std::vector<float> t((size_t)cin_pad * Cout * nt, 0.f);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int tap = 0; tap < nt; ++tap) t[((size_t)ci * Cout + co) * nt + (nt - 1 - tap)] = w[((size_t)co * Cin + ci) * nt + tap];
  conv_pack_weights(dtype, t.data(), cin_pad, Cout, ks, dst);
}

ConvGeom conv_geometry(const ConvDesc& d) {
  Geo g; compute_geo(d, g);
  ConvGeom r;
  r.Ho = g.Ho; r.Wo = g.Wo; r.BM = g.BM; r.BN = g.BN; r.lds_bytes = g.lds;
  r.grid_m = g.groups * g.tiles_x * g.tiles_y; r.grid_n = (d.Cout + g.BN - 1) / g.BN;
  const int CH = d.dtype == 0 ? 16 : 32;
  const mi355_debug_config& Kg = d.knobs ? *d.knobs : mi355_default_debug();
  if (d.C0 % CH == 0 && d.C1 % CH == 0) {
    const int pc = pp_config(Kg.conv_pp, d.ks, g.G, g.bn_pack, d.out_mode, g.stride, (d.C0 + d.C1) / CH * (d.wsplit ? 2 : 1), d.pro_a != nullptr, d.pro_silu != 0, d.N, g.Ho, g.Wo, d.Cout);
    if (pc >= 0) {   // ping-pong kernel (conv_pp.inc.h): 256 px x 256 ch or 512 px x 128 ch tiles
      r.BM = pc == 0 ? 256 : 512; r.BN = pc == 0 ? 256 : 128;
      r.lds_bytes = pc == 0 ? pp::D<0>::lds_bytes(d.pro_a != nullptr) : pp::D<1>::lds_bytes(d.pro_a != nullptr);
      return r;
    }
  }
  if (d.C0 % CH == 0 && d.C1 % CH == 0 &&
      ws_eligible((d.knobs ? d.knobs : &mi355_default_debug())->conv_ws, d.ks, g.BM, g.BN, g.G, g.bn_pack, d.out_mode, g.stride, (d.C0 + d.C1) / CH * (d.wsplit ? 2 : 1), d.N, g.Ho, g.Wo, d.Cout)) {
    r.BM = 256; r.lds_bytes = ws::LDS_BYTES;   // warp-specialised persistent kernel: 16 x 16 pixel tiles (grid_m / grid_n stay the plain launch's: workspace sizing)
  }
  return r;
}

static int conv_launch_impl(const ConvDesc& d, hipStream_t stream, int* gn_slots_used, int* act_done, bool dry);
int conv_launch(const ConvDesc& d, hipStream_t stream, int* gn_slots_used, int* act_done) { return conv_launch_impl(d, stream, gn_slots_used, act_done, false); }
// 0: a conv with a fused skip conv (ConvDesc::skip_src0) of this description would launch; nothing is launched
int conv_fused_skip_ok(const ConvDesc& d) { return d.skip_src0 ? conv_launch_impl(d, nullptr, nullptr, nullptr, true) : 1; }

static int conv_launch_impl(const ConvDesc& d, hipStream_t stream, int* gn_slots_used, int* act_done, bool dry) {
  if (gn_slots_used) *gn_slots_used = 0;
  if (act_done) *act_done = 0;
  const bool fskip = d.skip_src0 != nullptr;   // only the small-level kernel carries a fused skip conv
  if (!fskip) {
    int r = conv1x1_try_launch(d, stream, gn_slots_used);
    if (r <= 0) return r;
    r = conv_out_try_launch(d, stream);
    if (r <= 0) return r;
    r = conv_in_try_launch(d, stream, gn_slots_used);
    if (r <= 0) return r;
  }
  const int CH = chunk_of(d.dtype);
  const int Cin = d.C0 + d.C1;
  MI355_REQUIRE(d.ks == 1 || d.ks == 3, -1, "conv: kernel size must be 1 or 3");
  MI355_REQUIRE(d.C0 % CH == 0 && d.C1 % CH == 0 && Cin > 0, -2, "conv: source channels must be multiples of the 64-byte chunk");
  MI355_REQUIRE(d.mode == CONV_UNIT || d.ks == 3, -1, "conv: resampling modes need a 3x3 kernel");
  MI355_REQUIRE(d.out_mode == OUT_NCHW_F32 || d.Cout % 32 == 0, -2, "conv: NHWC output needs Cout % 32 == 0 (a wave stores whole 32-channel tiles)");
  MI355_REQUIRE(d.mode != CONV_POOL2 && d.res_mode != RES_POOL2, -4, "conv: average pooling is a separate pass (affine_pool / resample), not a gather mode");
  Geo g; compute_geo(d, g);
  MI355_REQUIRE(g.lds <= 160 * 1024, -4, "conv: LDS budget exceeded");
  MI355_REQUIRE(g.pit <= g.pit_t, -4, "conv: input patch too large for the staging loops");
  ConvKArgs a{};
  MI355_REQUIRE(!d.wsplit || d.dtype == DT_BF16, -1, "conv: hi / lo split weights are a bf16 mode");
  a.src0 = d.src0; a.src1 = d.src1; a.C0 = d.C0; a.C1 = d.C1; a.Cin = Cin; a.nreal = Cin / CH; a.nchunks = a.nreal * (d.wsplit ? 2 : 1);
  a.N = d.N; a.Hs = d.Hs; a.Ws = d.Ws; a.Hc = g.Hc; a.Wc = g.Wc; a.Ho = g.Ho; a.Wo = g.Wo;
  a.mode = d.mode; a.pad = g.pad; a.stride = g.stride;
  a.pro_a = d.pro_a; a.pro_b = d.pro_b; a.pro_silu = d.pro_silu;
  a.w = d.w; a.bias = d.bias; a.Cout = d.Cout; a.bn_pack = g.bn_pack;
  const size_t esz = d.dtype == 0 ? 4 : 2;
  const size_t b0 = (size_t)d.N * d.Hs * d.Ws * d.C0 * esz, b1 = (size_t)d.N * d.Hs * d.Ws * d.C1 * esz;
  const size_t wb = conv_packed_weight_bytes(d.dtype, d.Cout, Cin, d.ks, d.wsplit) + (fskip ? conv_packed_weight_bytes(d.dtype, d.Cout, d.skip_C0 + d.skip_C1, 1, 0) : 0);
  MI355_REQUIRE(b0 < 0xFFFF0000ull && b1 < 0xFFFF0000ull && wb < 0xFFFF0000ull, -4,
                "conv: a source tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.bytes0 = (uint32_t)b0; a.bytes1 = d.src1 ? (uint32_t)b1 : 0u; a.wbytes = (uint32_t)wb;
  a.Hr = d.res_mode == RES_UP2 ? g.Ho / 2 : g.Ho;
  a.Wr = d.res_mode == RES_UP2 ? g.Wo / 2 : g.Wo;
  const size_t ob = (size_t)d.N * g.Ho * g.Wo * d.Cout * esz, rb = (size_t)d.N * a.Hr * a.Wr * d.Cout * esz;
  MI355_REQUIRE(ob < 0xFFFF0000ull && rb < 0xFFFF0000ull, -4, "conv: output tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
  a.obytes = (uint32_t)ob; a.rbytes = d.res ? (uint32_t)rb : 0u;
  a.dbg = reinterpret_cast<unsigned long long*>(d.dbg);
  const mi355_debug_config& K = d.knobs ? *d.knobs : mi355_default_debug();
  a.stagger = K.conv_stagger; a.ablate = K.conv_ablate;
  a.err = d.err; a.spin_limit = K.conv_spin_limit > 0 ? K.conv_spin_limit : 1;
  a.emb = d.emb; a.emb_stride = d.emb_stride;
  a.res = d.res; a.res_mode = d.res ? d.res_mode : RES_NONE;
  a.out = d.out; a.out_mode = d.out_mode;
  a.gn_stats = nullptr; a.gn_slots = 0;
  a.lvw = g.lvw; a.lth = g.lth; a.G = g.G; a.PW = g.PW; a.PH = g.PH; a.NP = g.NP;
  a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
  if (fskip) {
    const size_t k0 = (size_t)d.N * g.Ho * g.Wo * d.skip_C0 * esz, k1 = (size_t)d.N * g.Ho * g.Wo * d.skip_C1 * esz;
    MI355_REQUIRE(k0 < 0xFFFF0000ull && k1 < 0xFFFF0000ull, -4, "conv: a source tensor exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
    a.sk0 = d.skip_src0; a.sk1 = d.skip_src1; a.SC0 = d.skip_C0; a.SC1 = d.skip_src1 ? d.skip_C1 : 0;
    a.skbytes0 = (uint32_t)k0; a.skbytes1 = d.skip_src1 ? (uint32_t)k1 : 0u;
  }
  dim3 grid(g.groups * g.tiles_x * g.tiles_y, (d.Cout + g.BN - 1) / g.BN);
  const bool gn_ok = d.gn_stats && d.out_mode == OUT_NHWC && g.G == 1 && d.Cout % g.BN == 0 && d.Cout % 4 == 0;
  if (!fskip) {   // ping-pong kernel (conv_pp.inc.h): 256- or 128-channel output tiles, input as it is or through the in-LDS GroupNorm + SiLU prologue
    const int pc = pp_config(K.conv_pp, d.ks, g.G, g.bn_pack, d.out_mode, g.stride, a.nchunks, d.pro_a != nullptr, d.pro_silu != 0, d.N, g.Ho, g.Wo, d.Cout);
    const int pp_slots = pc >= 0 ? pp_gn_slots(pc, g.Ho, g.Wo) : 0;
    if (pc >= 0 && gn_ok && pp_slots <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = pp_slots; }
    int pp_act = 0;
    if (d.act_out && act_done && (K.gn_epilogue & 2) && d.act_out == d.out && !d.act_raw) {   // in place: the activated tensor replaces the raw one
      a.act_out = d.act_out; a.act_gamma = d.act_gamma; a.act_beta = d.act_beta; a.act_film = d.act_film; a.act_film_stride = d.act_film_stride;
      a.act_eps = d.act_eps; a.act_silu = d.act_silu; a.act_raw = 0;
    }
    const int r = dispatch_dtype(d.dtype, [&](auto t) { return launch_pp<decltype(t)>(a, K.conv_pp, d.ks, stream, &pp_act); });
    if (r == 0) {
      MI355_CHECK_HIP(hipGetLastError());
      if (pp_act) { if (act_done) *act_done = 1; if (gn_slots_used) *gn_slots_used = 0; }
      else if (gn_slots_used) *gn_slots_used = a.gn_slots;
      return 0;
    }
    if (r < 0) return r;
    a.act_out = nullptr;
    a.gn_stats = nullptr; a.gn_slots = 0;
  }
  if (!fskip) {   // dominant shapes: warp-specialised persistent kernel (conv_ws.inc.h)
    const int ws_slots = 2 * ((g.Wo + ws::VW - 1) / ws::VW) * ((g.Ho + ws::TH - 1) / ws::TH);   // (16x16 pixel tile, 8-row half) per image
    if (gn_ok && ws_slots <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = ws_slots; }
    const int r = dispatch_dtype(d.dtype, [&](auto t) { return launch_ws<decltype(t)>(a, K.conv_ws, g.BM, g.BN, d.ks, stream); });
    if (r == 0) {
      MI355_CHECK_HIP(hipGetLastError());
      if (gn_slots_used) *gn_slots_used = a.gn_slots;
      return 0;
    }
    if (r < 0) return r;
    a.gn_stats = nullptr; a.gn_slots = 0;
  }
  {   // 8x8 / 4x4 levels: barrier-free K loop over an LDS-resident patch, weights straight into registers (conv_small.inc.h)
    if (d.act_out && act_done && (K.gn_epilogue & 1) && d.act_out != d.out) {
      a.act_out = d.act_out; a.act_gamma = d.act_gamma; a.act_beta = d.act_beta; a.act_film = d.act_film; a.act_film_stride = d.act_film_stride;
      a.act_eps = d.act_eps; a.act_silu = d.act_silu; a.act_raw = d.act_raw;
      a.act_stride = d.act_stride ? d.act_stride : d.Cout; a.act_coff = d.act_coff; a.act_cpg = d.act_cpg ? d.act_cpg : d.Cout / 32;
      const size_t ab = (size_t)d.N * g.Ho * g.Wo * a.act_stride * esz;
      MI355_REQUIRE(ab < 0xFFFF0000ull, -4, "conv: activated output exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
      a.abytes = (uint32_t)ab;
      if (d.act2_out && (K.gn_epilogue & 4)) {
        a.act2_out = d.act2_out; a.act2_gamma = d.act2_gamma; a.act2_beta = d.act2_beta; a.act2_silu = d.act2_silu;
        a.act2_stride = d.act2_stride ? d.act2_stride : d.Cout; a.act2_coff = d.act2_coff; a.act2_cpg = d.act2_cpg ? d.act2_cpg : d.Cout / 32;
        const size_t a2b = (size_t)d.N * g.Ho * g.Wo * a.act2_stride * esz;
        MI355_REQUIRE(a2b < 0xFFFF0000ull, -4, "conv: activated output exceeds 4 GiB (32-bit buffer offsets): run the batch in slices");
        a.a2bytes = (uint32_t)a2b;
      }
      a.warm = (K.l2_warm & 1) ? d.warm : nullptr; a.warm_bytes = a.warm ? d.warm_bytes : 0u;
    }
    const int r = dispatch_dtype(d.dtype, [&](auto t) { return launch_small<decltype(t)>(a, K.conv_small, d.ks, stream, act_done, dry); });
    if (dry) return r;
    if (r == 0) { MI355_CHECK_HIP(hipGetLastError()); return 0; }
    MI355_REQUIRE(!fskip || r < 0, -5, "conv: this launch cannot carry the fused skip conv (ask conv_fused_skip_ok first)");
    if (act_done) *act_done = 0;
    a.act_out = nullptr; a.act2_out = nullptr; a.warm = nullptr; a.warm_bytes = 0;
    if (r < 0) return r;
  }
  {
    const int slots = g.tiles_x * g.tiles_y * (g.BN == 32 ? 4 : 2);   // (pixel tile, pixel-wave) per image
    if (gn_ok && slots <= d.gn_slots_cap) { a.gn_stats = d.gn_stats; a.gn_slots = slots; }
  }
  int rc = dispatch_dtype(d.dtype, [&](auto t) { return launch_cfg<decltype(t)>(a, g.BM, g.BN, d.ks, g.pit_t, grid, g.lds, stream); });
  if (rc) return rc;
  MI355_CHECK_HIP(hipGetLastError());
  if (gn_slots_used) *gn_slots_used = a.gn_slots;
  return 0;
}
