// How does v_dot2c_f32_f16 (the int4 GEMVs' multiply-accumulate: D = a.x * b.x + a.y * b.y + D) round? For random operands -- a = denormal
// nibble pairs as in the dequant-free GEMV body, b = fp16 activations, c = a running fp32 sum -- the device result is held against the exact
// sum (double: the two products are exact, the three-term sum is exact in double here) rounded to fp32 three ways: to nearest even, toward
// zero, toward minus infinity. Round 5 (tools/bias_probe.py) found a signed error of the GEMVs that the restatement does not have; this
// names its source.      hipcc --offload-arch=gfx950 -O2 tools/lab/t_dot2_round.hip -o tools/lab/t_dot2_round && tools/lab/t_dot2_round
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const unsigned* a, const unsigned* b, const float* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a[i]), __builtin_bit_cast(h2, b[i]), c[i], false);
}
static double h2d(unsigned short h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v = e == 0 ? ldexp((double)m, -24) : ldexp((double)(m | 1024), e - 25);
    return s ? -v : v;
}
static float round_dir(double v, int dir) {      // dir 0 nearest even, 1 toward zero, 2 toward -inf
    float f = (float)v;                          // nearest even (the host's default mode)
    if (dir == 0 || (double)f == v) return f;
    if (dir == 1) { if (fabs((double)f) > fabs(v)) f = nextafterf(f, 0.f); return f; }
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
}
int main() {
    const int n = 1 << 20;
    std::vector<unsigned> a(n), b(n);
    std::vector<float> c(n), r(n);
    srand(7);
    for (int mode = 0; mode < 3; mode++) {
        for (int i = 0; i < n; i++) {
            const unsigned q0 = rand() & 15, q1 = rand() & 15;
            a[i] = mode == 2 ? (((rand() & 0x7BFF) | (rand() & 1 ? 0x8000 : 0)) | (unsigned)((rand() & 0x7BFF) | (rand() & 1 ? 0x8000 : 0)) << 16)   // any finite fp16 pair
                             : (q0 | (q1 << 16)) << (mode == 1 ? 4 : 0);                                                                        // even / odd nibbles in place
            unsigned short x0 = (unsigned short)(0x3000 + (rand() % 0x1400)) | (rand() & 1 ? 0x8000 : 0), x1 = (unsigned short)(0x3000 + (rand() % 0x1400)) | (rand() & 1 ? 0x8000 : 0);
            b[i] = x0 | ((unsigned)x1 << 16);
            const double scale = mode == 2 ? 1.0 : ldexp(1.0, -20);
            c[i] = (float)(((rand() / (double)RAND_MAX) * 2.0 - 1.0) * 40.0 * scale);     // a running sum of the size a unit's chain reaches
        }
        unsigned *da, *db; float *dc, *dr;
        hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dr, n * 4);
        hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dr, da, db, dc, n);
        hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
        long eq[3] = {0, 0, 0}, inexact = 0;
        double sum_err_ulp = 0.0, sum_abs_ulp = 0.0;
        for (int i = 0; i < n; i++) {
            const double ex = h2d(a[i] & 0xFFFF) * h2d(b[i] & 0xFFFF) + h2d(a[i] >> 16) * h2d(b[i] >> 16) + (double)c[i];
            const float rn = round_dir(ex, 0);
            if ((double)rn == ex) continue;            // representable: every mode agrees
            inexact++;
            for (int d = 0; d < 3; d++) eq[d] += r[i] == round_dir(ex, d);
            const double ulp = fabs((double)nextafterf(rn, INFINITY) - (double)rn);
            sum_err_ulp += ((double)r[i] - ex) / ulp;
            sum_abs_ulp += fabs((double)r[i] - ex) / ulp;
        }
        printf("%s: %ld inexact sums of %d: device == nearest-even %.4f, == toward-zero %.4f, == toward-minus-infinity %.4f; mean signed error %+.4f ulp, mean |error| %.4f ulp\n",
               mode == 0 ? "even nibbles (q * 2^-24) x fp16" : mode == 1 ? "odd nibbles (16 q * 2^-24) x fp16" : "fp16 x fp16", inexact, n,
               eq[0] / (double)inexact, eq[1] / (double)inexact, eq[2] / (double)inexact, sum_err_ulp / inexact, sum_abs_ulp / inexact);
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dr);
    }
    return 0;
}
