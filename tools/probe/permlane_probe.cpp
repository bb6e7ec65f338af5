// Probe of v_permlane16_swap / v_permlane32_swap lane semantics on gfx950 (prints which lane's value each lane ends up with).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 vdst", "p16 vsrc", "p32 vdst", "p32 vsrc"};
  for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[t * 64 + l]); printf("\n"); }
  return 0;
}
