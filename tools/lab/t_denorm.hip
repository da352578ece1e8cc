#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, unsigned wa, unsigned xb) {
  h2 a = __builtin_bit_cast(h2, wa), b = __builtin_bit_cast(h2, xb);
  float r0 = __builtin_amdgcn_fdot2(a, b, 0.f, false);
  float r1; asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r1) : "v"(wa), "v"(xb));
  float r2 = 0.f; asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r2) : "v"(wa), "v"(xb));
  out[0] = r0; out[1] = r1; out[2] = r2;
  h2 p = a * b; out[3] = (float)p.x; out[4] = (float)p.y;
}
int main() {
  float* d; hipMalloc(&d, 64); float h[8];
  // a = (5 * 2^-24 denormal, 0x00F0 = 240*2^-24), b = (1.0, 2.0)
  unsigned wa = 0x00F00005u, xb = 0x40003C00u;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, wa, xb); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  double expect = (5.0 + 2.0 * 240.0) / 16777216.0;
  printf("expect %.9g | fdot2 builtin %.9g | v_dot2_f32_f16 %.9g | v_dot2c %.9g | pk_mul lanes %.9g %.9g\n", expect, h[0], h[1], h[2], h[3], h[4]);
  return 0;
}
