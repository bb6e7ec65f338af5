#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(const void* src, uint32_t bytes, u32x4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, bytes, 0x00020000);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // each wave moves 1 KB: lane l's 16 bytes land at lds_base + l*16
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, tid * 16, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 4096 + wave * 1024), 16, tid * 16, 4096, 0, 0);
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
  __syncthreads();
  out[tid] = *reinterpret_cast<u32x4*>(smem + tid * 16);
  out[256 + tid] = *reinterpret_cast<u32x4*>(smem + 4096 + tid * 16);
}
int main() {
  const int n = 8192;
  uint32_t* h = (uint32_t*)malloc(n); for (int i = 0; i < n / 4; ++i) h[i] = i * 3 + 1;
  void* d; u32x4* o; hipMalloc(&d, n); hipMalloc((void**)&o, n); hipMemcpy(d, h, n, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, (uint32_t)n, o);
  uint32_t* r = (uint32_t*)malloc(n); hipMemcpy(r, o, n, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n / 4; ++i) if (r[i] != h[i]) { if (bad < 5) printf("mismatch %d: %u vs %u\n", i, r[i], h[i]); ++bad; }
  printf("lds dma b128: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
