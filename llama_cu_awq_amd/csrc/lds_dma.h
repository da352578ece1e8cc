// lds_dma.h -- gfx950 LDS-DMA primitives shared by the strips GEMVs (gemv_strip*.h) and the split-context attention role
// (attention.h): a wave streams 1 KiB pieces from HBM straight into LDS (`buffer_load_dwordx4 ... lds`: no VGPR on the way in), waits for
// its oldest piece with s_waitcnt vmcnt (a wave's loads return in order: no flags, no polling), reads it back with ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>

namespace q4 {

// one LDS-DMA piece: 64 lanes x 16 B from (descriptor, soffset + lane's voff) to LDS bytes [lds_dst + lane * 16, + 16): the LDS address
// follows the LANE, the memory address the offset. M0 (the LDS destination) is written in the statement that uses it; hipcc neither
// counts these loads nor waits for them: every wait is an explicit wait_vmcnt below.
__device__ __forceinline__ void dma_piece(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_dst), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// the same without the "memory" clobber: for requests into LDS that the code around them neither reads nor writes -- the compiler may then move ITS loads
// (the LDS reads of a dot-product step) across the request instead of serialising every step behind it. (Volatile: the requests keep their order among
// themselves and with the barriers.)
__device__ __forceinline__ void dma_piece_free(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" ::"s"(lds_dst), "v"(voff), "s"(r), "s"(soff));
}
__device__ __forceinline__ void dma_piece_default_free(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(r), "s"(soff));
}
__device__ __forceinline__ void dma_piece_default(unsigned lds_dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// Descriptor for a block's share of a tensor: based at the block's first byte and bounded at the tensor's end. The range check of a raw
// buffer covers the lane's voffset (+ the immediate) only, NOT soffset: with the tensor's base in the descriptor and the block's start in
// soffset, a 1 KiB piece that begins near the tensor's end reads up to 1 KiB past it (harmless inside a model's weight slab, a latent
// fault for a caller whose tensor ends with its allocation). Based like this, lanes past the end are out of range: no memory request.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_from(const void* tensor, unsigned first_byte, unsigned tensor_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(tensor) + first_byte), 0, (int)(tensor_bytes - first_byte), 0x00020000);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the same with a count that is a constant only after unrolling (0 .. 15; anything larger waits for 15: stronger, never wrong)
__device__ __forceinline__ void wait_vmcnt_upto15(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        default: wait_vmcnt<15>(); break;
    }
}
// block barrier that waits for the wave's LDS operations only (hipcc's __syncthreads may add s_waitcnt vmcnt(0): it would drain the stream)
__device__ __forceinline__ void block_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace q4
