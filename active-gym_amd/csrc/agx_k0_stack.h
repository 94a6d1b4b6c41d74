// agx_k0_stack.h - K0: ring <-> stack order (k_stack_u8, k_set_stack, k_full).
#pragma once
#include "agx_common.h"

namespace agx {

// ---------------------------------------------------------------------------------------------
// K0: stack-order views of the ring
// ---------------------------------------------------------------------------------------------
struct StackParams {
    uint8_t *ring;
    int32_t *head;           // current head (read), or written by k_set_stack
    const uint8_t *in_u8;
    uint8_t *out_u8;
    float *out_f32;
    int32_t words, fs;       // words = oh*ow/4
};

// grid = (ceil(words/256), fs, N)
__global__ __launch_bounds__(kThreads) void k_stack_u8(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= p.words) return;
    int slot = p.head[n] + j;
    if (slot >= p.fs) slot -= p.fs;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.ring) + ((size_t)n * p.fs + slot) * p.words;
    reinterpret_cast<uint32_t *>(p.out_u8)[((size_t)n * p.fs + j) * p.words + i] = src[i];
}

__global__ __launch_bounds__(kThreads) void k_set_stack(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i == 0 && j == 0) p.head[n] = 0;
    if (i >= p.words) return;
    const size_t o = ((size_t)n * p.fs + j) * p.words + i;
    reinterpret_cast<uint32_t *>(p.ring)[o] = reinterpret_cast<const uint32_t *>(p.in_u8)[o];
}

__global__ __launch_bounds__(kThreads) void k_full(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= p.words) return;
    int slot = p.head[n] + j;
    if (slot >= p.fs) slot -= p.fs;
    const uint32_t v = (reinterpret_cast<const uint32_t *>(p.ring) + ((size_t)n * p.fs + slot) * p.words)[i];
    float4 o;
    o.x = unit(v & 0xFF);
    o.y = unit((v >> 8) & 0xFF);
    o.z = unit((v >> 16) & 0xFF);
    o.w = unit(v >> 24);
    // write-once observation stream: written through (sc1) like the fovea kernels' (store_obs, agx_k2_fixed.h); the frame of
    // (n, j) is the buffer - wave-uniform by construction
    const uintptr_t a = reinterpret_cast<uintptr_t>(reinterpret_cast<float4 *>(p.out_f32) + ((size_t)n * p.fs + j) * p.words);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, p.words * 16, 0x00027000);
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v w = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, i * 16, 0, 16 /* sc1 */);
}

}  // namespace agx
