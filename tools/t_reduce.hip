#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../llama_cu_awq_amd/csrc/q4_device.h"
using namespace q4;
__device__ __forceinline__ float swap32_add(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v0 = 1.f, v1 = 10.f, v2 = 100.f, v3 = 1000.f;
  float u0 = swap32_add(v0, v2), u1 = swap32_add(v1, v3);
  float w = swap16_add(u0, u1);
  out[l] = u0; out[64 + l] = u1; out[128 + l] = w; out[192 + l] = row16_sum(w);
  auto r = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
  out[256 + l] = r[0]; out[320 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
  out[384 + l] = q[0]; out[448 + l] = q[1];
}
int main() {
  float* d; hipMalloc(&d, 512 * 4); float h[512];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 512 * 4, hipMemcpyDeviceToHost);
  const char* names[] = {"u0", "u1", "w", "rowsum", "p32.r0", "p32.r1", "p16.r0", "p16.r1"};
  for (int a = 0; a < 8; a++) { printf("%-7s:", names[a]); for (int i = 0; i < 64; i += (a < 4 ? 8 : 1)) printf(" %g", h[a * 64 + i]); printf("\n"); }
}
