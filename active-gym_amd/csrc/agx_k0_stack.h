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
    reinterpret_cast<float4 *>(p.out_f32)[((size_t)n * p.fs + j) * p.words + i] = o;
}

}  // namespace agx
