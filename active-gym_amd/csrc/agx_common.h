// agx_common.h - constants, geometry descriptors and the small gfx950 helpers shared by every kernel family
// (scalar-cache loads, 24-bit multiplies, diagnostic stamps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "agx.h"
#include "agx_taps.h"

namespace agx {

constexpr int kRawH = 210;
constexpr int kRawW = 160;
constexpr int kRawRowBytes = kRawW * 3;
constexpr int kRawFrameBytes = kRawH * kRawRowBytes;
constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// geometry: compile-time for the headline 84x84 / 30x30 configuration, run-time otherwise
// ---------------------------------------------------------------------------------------------
template <int OH, int OW, int FH, int FW>
struct GeomS {
    __host__ __device__ constexpr int oh() const { return OH; }
    __host__ __device__ constexpr int ow() const { return OW; }
    __host__ __device__ constexpr int fh() const { return FH; }
    __host__ __device__ constexpr int fw() const { return FW; }
};
struct GeomR {
    int oh_, ow_, fh_, fw_;
    __host__ __device__ int oh() const { return oh_; }
    __host__ __device__ int ow() const { return ow_; }
    __host__ __device__ int fh() const { return fh_; }
    __host__ __device__ int fw() const { return fw_; }
};

// float32(k)/255 exactly as numpy's `state.astype(np.float32) / 255.` (atari_env.py:75):
// IEEE correctly-rounded single division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt;
// tests/test_gpu_parity.py checks all 256 values bit for bit).
__device__ __forceinline__ float unit(uint32_t k) { return (float)k / 255.0f; }



// float32(k) / 255 correctly rounded without the division sequence: q = k * y, r = k - 255 q (exact in an FMA),
// q + r * y with y = RN(1/255) (Markstein).  tests/host_tables_harness.cpp and the GPU parity tests check all 256
// values bit for bit against the IEEE quotient.
__host__ __device__ __forceinline__ float unit_fast(float k) {
    const float y = 1.0f / 255.0f;
    const float q = k * y;
    const float r = fmaf(-q, 255.0f, k);
    return fmaf(r, y, q);
}

// Wave-uniform byte through the scalar cache.  hipcc emits a VECTOR load + s_waitcnt vmcnt(0) for
// `p.cmd[n]` (it cannot prove the buffer read-only), i.e. a full memory round trip in front of the
// first frame load of every workgroup; s_load_dword is counted on lgkmcnt and served by the scalar
// cache.  Reads the aligned dword that contains the byte (same 4-byte word, never crosses a page).
__device__ __forceinline__ uint32_t uniform_load_u8(const uint8_t *ptr) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    const uint32_t *aligned = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    uint32_t w;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(aligned) : "memory");
    return (w >> (8 * (uint32_t)(a & 3))) & 0xFFu;
}
// 24-bit multiply at full rate.  hipcc lowers __mul24 / __umul24 to the quarter-rate v_mul_lo_u32
// whenever it cannot prove the operand ranges itself; every product on this path fits (operands < 2^24,
// result < 2^32).
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a[23:0] * b[23:0]) >> 32
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int32_t uniform_load_i32(const int32_t *ptr) {
    int32_t w;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(ptr) : "memory");
    return w;
}


// i / d for 0 <= i < 2^20 and 1 <= d <= 1024 (agx_create caps obs at 1024 x 1024), exact: floor((i + 0.5) * (1/d)) in f32 - three full-rate VALU operations
// where a run-time 32-bit integer division is ~35 (K4 spent most of its 1.5 k VALU instructions per wave on them).
// The distance of (i + 0.5)/d from an integer is >= 0.5/d, the f32 error of the product is < (i + 0.5) * 2^-22 / d:
// exact for i < 2^21; verified exhaustively over the stated range, also with a reciprocal 1 ulp off (v_rcp_f32).
struct FastDiv {
    float rcp;
    int d;
    __device__ __forceinline__ explicit FastDiv(int d_) : rcp(__builtin_amdgcn_rcpf((float)d_)), d(d_) {}
    __device__ __forceinline__ int div(int i) const { return (int)(((float)i + 0.5f) * rcp); }
    __device__ __forceinline__ void divmod(int i, int &q, int &r) const {
        q = div(i);
        r = i - q * d;
    }
};

#ifdef AGX_STAMPS
// slot 5 of every wave's record holds where it ran: XCC_ID | HW_ID << 8 (se/cu/simd/wave slot)
#define AGX_STAMP(i)                                                                              \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (p.stamps && (threadIdx.x & 63) == 0) {                                                \
            unsigned long long *rec_ = p.stamps +                                                 \
                (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (T / 64) + (threadIdx.x >> 6)) * 8; \
            rec_[(i)] = t_;                                                                       \
            if ((i) == 0) {                                                                       \
                unsigned xcc_, hw_;                                                               \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));               \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                 \
                unsigned long long rt_;                                                           \
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory");  \
                rec_[5] = ((unsigned long long)hw_ << 8) | (xcc_ & 0xFF);                         \
                rec_[6] = rt_;                                                                    \
            }                                                                                     \
            if ((i) == 4) {                                                                       \
                unsigned long long rt_;                                                           \
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory");  \
                rec_[7] = rt_;                                                                    \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#else
#define AGX_STAMP(i) do {} while (0)
#endif

}  // namespace agx
