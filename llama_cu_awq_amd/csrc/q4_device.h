// q4_device.h -- wave64 / gfx950 device helpers shared by all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q4 {

typedef _Float16 f16_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int as_i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float as_f(int v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ unsigned as_u(h2 v) { return __builtin_bit_cast(unsigned, v); }

__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(f16_t, b); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (f16_t)f); }   // RNE
__device__ __forceinline__ float round_h(float f) { return (float)(f16_t)f; }

// DPP cross-lane move (wave64, 16-lane rows). CTRL: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return as_f(__builtin_amdgcn_update_dpp(0, as_i(v), CTRL, 0xF, 0xF, true));
}

// all-reduce inside each 16-lane DPP row: every lane ends with its row's sum
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return as_f(__builtin_amdgcn_readlane(as_i(v), lane));
}

// full 64-lane sum, result wave-uniform. Fixed order: 4 DPP steps per row, then (r0+r1)+(r2+r3).
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

// streamed-once weights: non-temporal 128-bit load (global_load_dwordx4 ... nt)
__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

// sum of squares of 8 packed halves: four v_dot2c_f32_f16 (x.x*x.x + x.y*x.y + acc, fp32) in element order -- the
// canonical per-chunk order of every rmsnorm in this library (stand-alone kernel and fused staging share this function)
__device__ __forceinline__ float sumsq8(u32x4 v, float acc) {
#pragma unroll
    for (int d = 0; d < 4; d++) acc = __builtin_amdgcn_fdot2(as_h2(v[d]), as_h2(v[d]), acc, false);
    return acc;
}

// Canonical rmsnorm reduction (block-size independent, so the standalone rmsnorm kernel and every fused
// consumer produce the same bits): per-16B-chunk partials in LDS, then EVERY wave sums them lane-strided by 64
// and tree-reduces on its own -- redundant, but it needs no second barrier and no serial LDS chain on one wave.
// `part` holds `nchunks` floats (NCH > 0: compile-time count). Returns 1/sqrt(mean + eps).
// Caller: every thread has written part[] for its chunks and called __syncthreads() before.
template <int NCH>
__device__ __forceinline__ float rms_scale_from_partials(const float* part, int nchunks, int size) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if constexpr (NCH > 0) {
        constexpr int R = (NCH + 63) / 64;
        float v[R];
#pragma unroll
        for (int i = 0; i < R; i++) v[i] = (lane + 64 * i < NCH) ? part[lane + 64 * i] : 0.f;   // independent LDS reads
#pragma unroll
        for (int i = 0; i < R; i++) s += v[i];
    } else {
        for (int u = lane; u < nchunks; u += 64) s += part[u];
    }
    s = wave_sum(s);
    float ss = s / (float)size;        // gpu_kernels.h:88
    ss += 1e-5f;                       // :89
    return 1.0f / sqrtf(ss);           // :90
}

// RoPE angle of pair index i at position pos: RoPERotation_kernel gpu_kernels.h:338-342
__device__ __forceinline__ void rope_angle(int i, int head_size, int pos, float rope_theta, float& fcr, float& fci) {
    const int head_dim = (i * 2) % head_size;
    const float freq = 1.0f / powf(rope_theta, head_dim / (float)head_size);
    const float val = pos * freq;
    fcr = cosf(val);
    fci = sinf(val);
}

// normalise 8 halves: half(x * (ss * w))   gpu_kernels.h:100-102
// Mixed-precision FMAs read the fp16 halves directly (no v_cvt) and round the fp32 product once to fp16:
//   t = ss * float(w)            v_fma_mix_f32      (fp32)
//   r = half(float(x) * t)       v_fma_mixlo/hi_f16 (fp32 product, one rounding to fp16) == the reference's arithmetic
// 4 VALU per pair instead of the 7 hipcc emits for the C expression (cvt x2, cvt w2, 2 pk_mul, cvt_pk).
__device__ __forceinline__ u32x4 rms_apply8(u32x4 xv, u32x4 wv, float ss) {
    u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float tl, th;
        unsigned r;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(tl) : "v"(wv[d]), "v"(ss));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(th) : "v"(wv[d]), "v"(ss));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xv[d]), "v"(tl));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(xv[d]), "v"(th));
        o[d] = r;
    }
    return o;
}

}  // namespace q4
