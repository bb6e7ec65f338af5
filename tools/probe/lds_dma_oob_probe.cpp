// Probe (gfx950): what `buffer_load_dwordx4 ... lds` does for (1) lanes whose offset is out of the descriptor's range,
// (2) lanes masked off by EXEC, (3) per-lane gathered source addresses.  The DMA patch loader of the persistent conv relies on:
// out-of-range lanes WRITE ZEROS (zero padding for free), masked lanes leave LDS untouched, lane l lands at M0 base + 16 l.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k(const void* src, uint32_t bytes, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, bytes, 0x00020000);
  const int lane = threadIdx.x;
  // poison 3 KB of LDS
  for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(smem + i * 1024 + lane * 16) = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
  __syncthreads();
  // piece 0: gather: lane l reads fragment (63 - l); odd lanes point out of range
  const uint32_t off0 = (lane & 1) ? bytes + 64u * lane : (63 - lane) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, off0, 0, 0, 0);
  // piece 1: only lanes < 16 active
  if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, lane * 16, 1024, 0, 0);
  // piece 2: soffset pushes every lane out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 2048), 16, lane * 16, bytes, 0, 0);
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
  __syncthreads();
  for (int i = 0; i < 3; ++i) out[i * 64 + lane] = *reinterpret_cast<u32x4*>(smem + i * 1024 + lane * 16);
}
int main() {
  const int n = 4096;
  uint32_t* h = (uint32_t*)malloc(n); for (int i = 0; i < n / 4; ++i) h[i] = i * 3 + 1;
  void* d; u32x4* o; hipMalloc(&d, n); hipMalloc((void**)&o, 3072); hipMemcpy(d, h, n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, (uint32_t)n, o);
  uint32_t r[768]; hipMemcpy(r, o, 3072, hipMemcpyDeviceToHost);
  int bad_gather = 0, oob_zero = 0, oob_kept = 0, oob_other = 0, mask_ok = 0, mask_bad = 0, act_bad = 0, so_zero = 0, so_other = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    const uint32_t v = r[l * 4 + j];
    if (l & 1) { if (v == 0) ++oob_zero; else if (v == 0xdeadbeefu) ++oob_kept; else ++oob_other; }
    else if (v != h[(63 - l) * 4 + j]) ++bad_gather;
    const uint32_t w = r[256 + l * 4 + j];
    if (l < 16) { if (w != h[256 + l * 4 + j]) ++act_bad; } else { if (w == 0xdeadbeefu) ++mask_ok; else ++mask_bad; }
    const uint32_t z = r[512 + l * 4 + j];
    if (z == 0) ++so_zero; else ++so_other;
  }
  printf("gather per-lane source: %s (%d bad)\n", bad_gather ? "FAIL" : "OK", bad_gather);
  printf("out-of-range lanes (voffset): zero-written %d, left untouched %d, other %d of 128 dwords\n", oob_zero, oob_kept, oob_other);
  printf("EXEC-masked lanes: untouched %d, overwritten %d of 192 dwords; active lanes bad %d\n", mask_ok, mask_bad, act_bad);
  printf("out-of-range via soffset: zero-written %d, other %d of 256 dwords\n", so_zero, so_other);
  return 0;
}
