// Probe: do a wave's MFMA stream and ANOTHER wave's VALU stream on the same SIMD overlap on gfx950?
// One workgroup of 8 waves per CU (two per SIMD).  Waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 a VALU loop; each role is
// timed alone and together (s_memtime per wave, max over waves; all 256 CUs busy so the clock sees a realistic load).
//   mode 0: MFMA only     mode 1: VALU only      mode 2: both
// VALU kinds: 0 = v_pk_fma_f32 chain x4 independent, 1 = v_exp_f32 (transcendental), 2 = ds_read_b128 stream, 3 = v_fma_f32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(512) probe(int mode, int iters_m, int iters_v, int prio_m, int prio_v, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mf = wave < 4;
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, t1 = 0;
  if (mf) {
    if (mode == 1) return;
    if (prio_m == 1) __builtin_amdgcn_s_setprio(1); else if (prio_m == 2) __builtin_amdgcn_s_setprio(2); else if (prio_m == 3) __builtin_amdgcn_s_setprio(3);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a = u32x4{(uint32_t)lane, 1u, 2u, 3u}, b = u32x4{4u, (uint32_t)lane, 6u, 7u};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.f) sink[0] = s;
  } else {
    if (mode == 0) return;
    if (prio_v == 1) __builtin_amdgcn_s_setprio(1); else if (prio_v == 2) __builtin_amdgcn_s_setprio(2); else if (prio_v == 3) __builtin_amdgcn_s_setprio(3);
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{(float)lane * 0.001f + i, 0.5f};
    const f32x2 m = f32x2{0.999f, 1.001f}, c = f32x2{0.001f, -0.001f};
    u32x4 r[4] = {};
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters_v; ++it) {
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * m + c;
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i][0] = __builtin_amdgcn_exp2f(x[i][0] * 0.5f);
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<volatile u32x4*>(lds + ((lane * 16 + i * 1024 + it * 64) & 16368));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i][0] = x[i][0] * 0.999f + 0.001f;
      }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    s += (float)(r[0][0] + r[1][1] + r[2][2] + r[3][3]);
    if (s == 12345.f) sink[1] = s;
  }
  if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int im, int iv) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 64);
  unsigned long long h[256 * 8];
  const int prios[3][2] = {{0, 0}, {2, 0}, {0, 2}};
  for (int pp = 0; pp < 3; ++pp)
    for (int mode = 0; mode < 3; ++mode) {
      if (pp > 0 && mode != 2) continue;
      hipMemset(d, 0, sizeof(h));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, im, iv, prios[pp][0], prios[pp][1], d, sink);   // warm
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, mode, im, iv, prios[pp][0], prios[pp][1], d, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      double sm = 0, sv = 0; int nm = 0, nv = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { if (w < 4 && mode != 1) { sm += h[b * 8 + w]; ++nm; } if (w >= 4 && mode != 0) { sv += h[b * 8 + w]; ++nv; } }
      printf("%-14s prio(m,v)=(%d,%d) mode %d: kernel %.1f us | mfma wave %.0f ticks (%d MFMA) | valu wave %.0f ticks (%d iters)\n", name, prios[pp][0], prios[pp][1], mode,
             ms * 1e3, nm ? sm / nm : 0.0, im * 8, nv ? sv / nv : 0.0, iv);
    }
}
int main() {
  // iteration counts chosen so each role alone lasts about the same time
  run<0>("pk_fma", 4000, 16000);
  run<1>("exp2", 4000, 4000);
  run<2>("ds_read_b128", 4000, 8000);
  run<3>("fma", 4000, 16000);
  return 0;
}
