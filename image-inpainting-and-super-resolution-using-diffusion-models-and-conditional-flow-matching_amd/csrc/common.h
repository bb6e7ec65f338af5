// Shared device/host helpers for libmi355_sampler (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstring>
#include <string>

typedef __bf16 bf16;
typedef _Float16 f16;      // MI355_F16 (round 5): fp16 storage + fp16 MFMA, the reference's own reduced-precision mode (use_fp16, unet.py:559)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// One "K chunk" is 64 bytes of channels per pixel in both precisions, so every LDS image (rows of
// 64 B split into four 16-B fragments) is byte-identical between the fp32 and the bf16 build of a
// kernel; only the number of channels per chunk differs.
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int CHUNK = 16;  // channels per 64-B chunk
  static constexpr int VEC = 4;     // channels per 16-B fragment
  static constexpr int DTYPE = 0;
};
template <> struct Elem<bf16> {
  static constexpr int CHUNK = 32;
  static constexpr int VEC = 8;
  static constexpr int DTYPE = 1;
};
template <> struct Elem<f16> {   // every layout decision follows the element SIZE (DTYPE 1 = two bytes): fp16 shares all of bf16's images
  static constexpr int CHUNK = 32;
  static constexpr int VEC = 8;
  static constexpr int DTYPE = 1;
};

// Two-byte element helpers: the 32-bit word w holds elements (lo, hi) of consecutive channels.
//   bf16: a shift / a mask (the value IS the upper half of an fp32); f16: v_cvt_f32_f16 (+ SDWA for the upper half)
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi, bf16) {
  lo = __builtin_bit_cast(float, w << 16); hi = __builtin_bit_cast(float, w & 0xffff0000u);
}
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi, f16) {
  const f16x2 h = __builtin_bit_cast(f16x2, w);
  lo = (float)h[0]; hi = (float)h[1];
}
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi, float) { lo = __builtin_bit_cast(float, w); hi = 0.f; }   // (never executed: fp32 images hold one element per word)
typedef __bf16 bf16x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi, bf16) { return __builtin_bit_cast(uint32_t, bf16x2_c{(bf16)lo, (bf16)hi}); }   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
__device__ __forceinline__ uint32_t pack2(float lo, float hi, f16) { return __builtin_bit_cast(uint32_t, f16x2{(f16)lo, (f16)hi}); }          // v_cvt_pk_f16_f32 .. (RNE; |x| > 65504 -> inf)
__device__ __forceinline__ uint32_t pack2(float lo, float, float) { return __builtin_bit_cast(uint32_t, lo); }
// four consecutive channels -> 8 bytes
template <typename T> __device__ __forceinline__ u32x2 pack4(const float (&v)[4], T) { return u32x2{pack2(v[0], v[1], T()), pack2(v[2], v[3], T())}; }
template <typename T> __device__ __forceinline__ u32x2 pack4(const f32x4& v, T) { return u32x2{pack2(v[0], v[1], T()), pack2(v[2], v[3], T())}; }

// 16-B fragment <-> float conversions
// NB: never __builtin_bit_cast a single element of an ext-vector lvalue (`bit_cast(float, v[i])` reads element 0
// for every i with hipcc 7.2); cast the whole vector and index the result.
__device__ __forceinline__ void frag_to_float(const u32x4& v, float (&f)[4], float) {
  const f32x4 t = __builtin_bit_cast(f32x4, v);
  f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
}
__device__ __forceinline__ void frag_to_float(const u32x4& v, float (&f)[8], bf16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void frag_to_float(const u32x4& v, float (&f)[8], f16) {
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2(v[i], f[2 * i], f[2 * i + 1], f16());
}
__device__ __forceinline__ u32x4 float_to_frag(const float (&f)[4], float) {
  return __builtin_bit_cast(u32x4, f32x4{f[0], f[1], f[2], f[3]});
}
__device__ __forceinline__ u32x4 float_to_frag(const float (&f)[8], bf16) {
  bf16x8 b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (bf16)f[i];  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return __builtin_bit_cast(u32x4, b);
}
__device__ __forceinline__ u32x4 float_to_frag(const float (&f)[8], f16) {
  f16x8 b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (f16)f[i];
  return __builtin_bit_cast(u32x4, b);
}

// one 16-B x 16-B fragment pair -> 16x16 f32 accumulator.
//   bf16: one v_mfma_f32_16x16x32_bf16 (K = 32 across the 4 lane quads)
//   f32 : four v_mfma_f32_16x16x4_f32 (K = 16), exact f32 (k-ordered fmaf chain)
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b, bf16) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b, f16) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b, float) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
}

template <bool FAST> __device__ __forceinline__ float silu_f(float v) {
  if (FAST) return __fdividef(v, 1.0f + __expf(-v));
  return v / (1.0f + expf(-v));
}

// torch.clip semantics: NaN propagates (fminf/fmaxf would drop it; DDPM(Ns<=20) relies on NaN).
__device__ __forceinline__ float clip_nan(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// ---- error plumbing (host) ----------------------------------------------------------------------
void mi355_set_error(const std::string& msg);
#define MI355_CHECK_HIP(expr)                                                                       \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      mi355_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
      return -3;                                                                                    \
    }                                                                                               \
  } while (0)
#define MI355_REQUIRE(cond, code, msg)                                                              \
  do {                                                                                              \
    if (!(cond)) {                                                                                  \
      mi355_set_error(std::string(msg) + " [" #cond "]");                                         \
      return code;                                                                                  \
    }                                                                                               \
  } while (0)

// Allow a kernel more than 64 KB of dynamic LDS (the CU has 160 KB).  The attribute is per (function, device): one bit per device
// per kernel instantiation (the static lives in this template's instantiation), set once, thread-safe, and a failure is reported
// instead of surfacing later as an opaque launch error.
template <typename K>
int mi355_allow_big_lds(K kern, const char* what) {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  MI355_CHECK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    mi355_set_error(std::string(what) + ": hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KB) failed: " + hipGetErrorString(e));
    return -3;
  }
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// ---- L2 warm-up of the next kernel's weights -----------------------------------------------------------------------------------
// A layer's packed weights (0.3-4.7 MB) are read by every workgroup of its conv at the same moment, cold from HBM / Infinity Cache:
// at the 8x8 / 4x4 levels that first touch is 20 % of the launch (in-network 28 us vs 22 us with warm weights).  The memory-bound
// GroupNorm pass that runs right before such a conv gets one EXTRA wave per workgroup that only touches those weights and exits
// (loads return in order within a wave, so a working wave would wait for the cold lines before its own data; a wave that has ended
// does not take part in later barriers).  Workgroup b runs on XCD b % 8 (one L2 each): the extra waves of every XCD cover the whole
// range once, one 128-B line per lane and step.
// NB (the HIP programming model does not promise this): the extra wave RETURNS before the barriers its workgroup runs later.  On
// gfx9-family hardware s_barrier counts the waves of the workgroup that are still alive, so a wave that has ended is not waited
// for (the same property every `if (tid >= n) return;` tail in front of a __syncthreads() relies on); the HIP specification calls a
// barrier not reached by all threads undefined.  This library targets gfx950 only; if that ever changes, give the warm-up its own
// launch (it was measured as such in round 2: the extra launch costs more than the warm-up gains).
__device__ __forceinline__ void l2_warm_wave(const void* p, uint32_t bytes) {
  const uint32_t rank = blockIdx.x >> 3, per = (gridDim.x + 7) >> 3, step = per * 64u * 128u;
  const char* base = reinterpret_cast<const char*>(p);
  uint32_t off = (rank * 64u + (threadIdx.x & 63)) * 128u;
  // 16 loads in flight per lane (4 MB per XCD at 256 workgroups; a longer range is warmed partially, a shorter one re-touches line 0).
  // The destinations stay live until all 16 have landed: a register reused while its load is in flight would be overwritten by it.
  uint32_t v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i, off += step) v[i] = *reinterpret_cast<const uint32_t*>(base + (off < bytes ? off : 0u));
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(v[i]));
}

// ---- fused GroupNorm statistics: per-wave partial sums from a conv epilogue -------------------------------------------------
// The reference normalises in a separate pass (GroupNorm32, AD/image_diffusion/nn.py:11-13,87-94); here the conv that PRODUCES a
// tensor also leaves, per (image, slot, channel quad), the fp32 sum and sum of squares of its final output values
// (`stats[N][slots][C/4][2]`), and the consumer's statistics pass shrinks to summing those partials (gn_finalize_kernel).  A slot
// is one wave's share of an image's pixels; every (image, slot, quad) is written by exactly one wave (plain stores: no atomics, no
// zero-fill, and the sums are bitwise reproducible).  Epilogue layout of every conv kernel here: a lane holds, per 16x16 MFMA
// tile (mi, ni), the 4 consecutive channels 16 ni + 4 lq .. + 3 of pixel lr = one channel quad.
template <int NI>
struct GnPartial {
  f32x2 s2[NI], q2[NI];   // element pairs: the epilogue is on the consumer waves' critical path, so v_pk_add / v_pk_fma (4 per quad)
  __device__ __forceinline__ GnPartial() {
#pragma unroll
    for (int i = 0; i < NI; ++i) { s2[i] = f32x2{0.f, 0.f}; q2[i] = f32x2{0.f, 0.f}; }
  }
  // valid = 1 for an in-image pixel, 0 otherwise; `masked` is wave-uniform (false when the tiling covers the image exactly)
  __device__ __forceinline__ void add(int ni, float v0, float v1, float v2, float v3, bool masked, float valid) {
    f32x2 lo = f32x2{v0, v1}, hi = f32x2{v2, v3};
    if (masked) { lo *= f32x2{valid, valid}; hi *= f32x2{valid, valid}; }
    s2[ni] += lo + hi;
    q2[ni] = lo * lo + q2[ni];
    q2[ni] = hi * hi + q2[ni];
  }
  // all-reduce over the 16 lanes of a DPP row (lr), four v_add_f32 with a DPP operand each: row_mirror, row_half_mirror,
  // quad_perm [2,3,0,1], quad_perm [1,0,3,2]
  static __device__ __forceinline__ float row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    return v;
  }
  // dst = &stats[((n * slots + slot) * (C / 4) + quad0) * 2], quad0 = channel quad of (ni = 0, lq = 0); lane (lq, lr = ni) stores
  // the pair of quad quad0 + 4 ni + lq: one store instruction per wave, 16 lanes x 8 bytes
  __device__ __forceinline__ void store(float* dst, int lq, int lr) {
    float ms = 0.f, mq = 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float ts = row_sum(s2[ni][0] + s2[ni][1]), tq = row_sum(q2[ni][0] + q2[ni][1]);
      if (lr == ni) { ms = ts; mq = tq; }
    }
    if (lr < NI) *reinterpret_cast<f32x2*>(dst + (size_t)(lr * 4 + lq) * 2) = f32x2{ms, mq};
  }
};
