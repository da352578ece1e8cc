// exp/lds_flags.h -- LABORATORY: flag words in LDS for blocks whose waves must not meet at a hardware barrier (the loader / consumer engine,
// the paced strips). Not used by the product.
#pragma once
#include <hip/hip_runtime.h>

namespace q4 {

// Flag words in LDS. A wave's LDS operations execute in program order, so "write data, then bump the flag" and "see the flag, then
// read data" need no hardware fence inside a workgroup; the relaxed forms + compiler barriers keep hipcc from re-ordering them AND
// from attaching its own waits: for an acquire / release at workgroup scope it emits s_waitcnt vmcnt(0) here (it cannot see the
// asm LDS-DMA pieces, but it counts the x loads at the kernel's entry), which would drain the loader's stream at every flag.
__device__ __forceinline__ unsigned lds_peek(unsigned* p) {
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void lds_post(unsigned* p, unsigned v, unsigned lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_bump(unsigned* p, unsigned lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// Every wait is on a wave of the SAME block (all resident by construction: no scheduling order can wedge it), and still bounded
// (~10 s): a wait that runs out -- a logic error, not a race -- raises the block's fail word, and the block then stores NaN, so the
// failure is loud in every consumer of the result instead of a hung GPU or plausible garbage.
constexpr unsigned ENG_SPIN_LIMIT = 1u << 27;
__device__ __forceinline__ unsigned lds_wait_ge(unsigned* p, unsigned target, unsigned* fail) {
    unsigned v = lds_peek(p);
    for (unsigned n = 0; v < target; n++) {
        if (n >= ENG_SPIN_LIMIT || ((n & 1023u) == 1023u && lds_peek(fail) != 0u)) { lds_post(fail, 1u, 0u); break; }
        __builtin_amdgcn_s_sleep(1);
        v = lds_peek(p);
    }
    return v;
}

}  // namespace q4
