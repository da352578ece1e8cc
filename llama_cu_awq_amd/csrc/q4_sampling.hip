// q4_sampling.hip -- temperature / top-p sampling (sampler.h:51-81, gpu_kernels.h:499-584).
#include <hip/hip_runtime.h>
#include "q4_device.h"
#include "q4_internal.h"
using namespace q4;

extern "C" int q4_sample_topp_device(Sampler* sampler, RunState* s, float coin) {
    (void)sampler; (void)s; (void)coin;
    snprintf(g_last_error, sizeof(g_last_error), "temperature/top-p sampling is not built yet (use -t 0)");
    return Q4_ERR_ARG;
}
