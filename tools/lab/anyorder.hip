// Does a dispatch without the AQL barrier bit (hipExtAnyOrderLaunch) overlap its predecessor on gfx950, and what does
// two-stream concurrency look like?  hipcc --offload-arch=gfx950 -O2 tools/lab/anyorder.hip -o tools/lab/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(unsigned long long* rec, int ticks) {
    if (threadIdx.x == 0) {
        unsigned long long t0 = wall_clock64();
        atomicMin(&rec[0], t0);
        while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
        atomicMax(&rec[1], wall_clock64());
    }
}

int main() {
    unsigned long long* rec;
    CHK(hipMalloc(&rec, 64));
    hipStream_t s1, s2;
    CHK(hipStreamCreate(&s1));
    CHK(hipStreamCreate(&s2));
    const int blocks = 256, ticks = 3000;   // 30 us at 100 MHz
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            unsigned long long init[4] = {~0ull, 0, ~0ull, 0};
            CHK(hipMemcpy(rec, init, sizeof(init), hipMemcpyHostToDevice));
            CHK(hipDeviceSynchronize());
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, rec, ticks);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, rec + 2, ticks);
            } else if (mode == 1) {
                hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, rec, ticks);
                hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, rec + 2, ticks);
            } else if (mode == 2) {
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, rec, ticks);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s2, rec + 2, ticks);
            } else {
                hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, nullptr, nullptr, 0, rec, ticks);
                hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, rec + 2, ticks);
            }
            CHK(hipDeviceSynchronize());
            unsigned long long r[4];
            CHK(hipMemcpy(r, rec, sizeof(r), hipMemcpyDeviceToHost));
            const char* names[] = {"same stream, plain", "same stream, AnyOrder both", "two streams", "same stream, AnyOrder on B"};
            printf("%-28s A: %.2f us   B starts %.2f us after A starts, ends %.2f\n", names[mode], (r[1] - r[0]) * 0.01,
                   ((long long)r[2] - (long long)r[0]) * 0.01, ((long long)r[3] - (long long)r[0]) * 0.01);
        }
    }
    return 0;
}
