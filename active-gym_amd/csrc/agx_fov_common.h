// agx_fov_common.h - what K2, K3 and K4 share: FovParams, sensory action -> fov_loc, window staging, fov state reset.
#pragma once
#include "agx_common.h"

namespace agx {

// ---------------------------------------------------------------------------------------------
// sensory action -> fov_loc   (fov_env.py:166-170,187-199; flexible: :270-271,300-324)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double load_action(const void *p, int dt, size_t i) {
    switch (dt) {
        case AGX_DT_F32: return (double)static_cast<const float *>(p)[i];
        case AGX_DT_F64: return static_cast<const double *>(p)[i];
        case AGX_DT_I32: return (double)static_cast<const int32_t *>(p)[i];
        default: return (double)static_cast<const int64_t *>(p)[i];
    }
}

// np.rint(np.clip(x, lo, hi)).astype(int); NaN is normalised to lo (the reference is undefined there)
__device__ __forceinline__ int clip_rint(double x, double lo, double hi) {
    x = fmax(x, lo);
    x = fmin(x, hi);
    return (int)rint(x);
}

struct FovParams {
    const uint8_t *ring;
    const int32_t *head;
    const int32_t *loc_in;
    int32_t *loc_out;
    const int32_t *res_in;      // flexible only
    int32_t *res_out;
    const void *action;         // [N][2] or nullptr
    const int32_t *action_type; // flexible only, may be nullptr
    const uint8_t *mask;        // [N] or nullptr
    float *obs;
    int32_t *user_loc;          // may be nullptr
    int32_t *user_res;          // may be nullptr
    const Tap *xtab;            // fixed/resize: [ow] lin2 taps fov_w -> obs_w
    const Tap *ytab;            // fixed/resize: [oh] lin2 taps fov_h -> obs_h
    double sas_lo, sas_hi;
    int32_t action_dt;
    int32_t relative;
    int32_t fs;
    int32_t out_mode;
    int32_t antialias;
    int32_t per_h, per_w;
    int32_t buf1_floats;        // generic kernels: size of the second LDS buffer (multiple of 4)
    // fused step (agx_step_fixed): the fovea work of one step is split around the ingest it rides with
    //   phase 0: stand-alone launch, `head` is the ring head after the ingest
    //   phase 1: same launch as the ingest: `head` is the head BEFORE it; only slots the ingest does not
    //            touch are processed (sl != written slot, env not cleared)
    //   phase 2: after the ingest: the written slot (all slots of a cleared env)
    const uint8_t *cmd;         // ingest command bytes (phases 1 and 2)
    int32_t phase;
    unsigned long long *stamps; // diagnostic builds only (AGX_STAMPS)
    // packed ragged output of the flexible raw-crop mode (agx_fovea_flexible_packed): env n's crops [fs][rh][rw], tight,
    // start at packed + packed_off[n]; an env whose crops would end past packed_cap floats is not written
    float *packed;
    const int64_t *packed_off;
    int64_t packed_cap;
};

// Raw inputs of the fov_loc update.  Kept separate from the arithmetic so that a kernel can issue
// these (vector) loads BEFORE its bulk loads: vmcnt retires in order, so waiting for them later does
// not drain the younger bulk loads.
struct LocIn {
    int r, c;
    uint32_t w[4];       // raw bits of the two action elements (4- or 8-byte each), converted later
};
__device__ __forceinline__ LocIn load_loc_inputs(const FovParams &p, int n) {
    LocIn in;
    const int2 rc = *reinterpret_cast<const int2 *>(p.loc_in + 2 * n);
    in.r = rc.x;
    in.c = rc.y;
    // two unconditional 8-byte loads, no branch and no use of the bits here, so no wait is forced:
    // 4-byte elements: a0 holds both; 8-byte elements: a0, a1 hold one each.  A null action reads
    // loc_in instead (ignored later).
    const bool wide = p.action_dt == AGX_DT_F64 || p.action_dt == AGX_DT_I64;
    const char *base = p.action ? static_cast<const char *>(p.action) + (size_t)n * (wide ? 16 : 8)
                                : reinterpret_cast<const char *>(p.loc_in + 2 * n);
    const uint2 a0 = *reinterpret_cast<const uint2 *>(base);
    const uint2 a1 = *reinterpret_cast<const uint2 *>(base + ((wide && p.action) ? 8 : 0));
    in.w[0] = a0.x;
    in.w[1] = wide ? a0.y : 0u;
    in.w[2] = wide ? a1.x : a0.y;
    in.w[3] = wide ? a1.y : 0u;
    return in;
}
// CONTRACT of the scalar forms below: (1) everything they read - fov_loc, fov_res, the ring head, the action, the action type - was
// written by an EARLIER launch (or the host): the scalar cache is only made coherent at a kernel boundary, so a kernel form that
// produces loc / head in the same launch that consumes them (the experiments' single-launch steps) must use load_loc_inputs (vector
// loads) instead; every shipped launch qualifies because the state is double-buffered and the head comes from the ingest launch
// before.  (2) every scalar element is a naturally aligned 4-byte type: s_load ignores the two low address bits.
static_assert(sizeof(int32_t) == 4 && sizeof(((FovParams *)nullptr)->action_type[0]) == 4 && sizeof(((FovParams *)nullptr)->head[0]) == 4 &&
              sizeof(((FovParams *)nullptr)->loc_in[0]) == 4 && sizeof(((FovParams *)nullptr)->res_in[0]) == 4,
              "s_load_dword reads aligned dwords: the scalar state loads need 4-byte elements");
// The same inputs + the ring head through the SCALAR cache (the env index is workgroup-uniform): four s_loads, one wait.  A
// scalar round trip is shorter than a vector one, and this hop sits in front of every workgroup's window fetch (K2: removing it
// altogether was worth 1.9 us of 20.5, profiles/r03_k2_ablation.txt).  `head` may be null (returns 0).
__device__ __forceinline__ LocIn load_loc_inputs_scalar(const FovParams &p, int n, const int32_t *head_ptr, int &head) {
    const bool wide = p.action_dt == AGX_DT_F64 || p.action_dt == AGX_DT_I64;
    const int32_t *loc = p.loc_in + 2 * n;
    const char *base = p.action ? static_cast<const char *>(p.action) + (size_t)n * (wide ? 16 : 8)
                                : reinterpret_cast<const char *>(loc);
    const char *base1 = base + ((wide && p.action) ? 8 : 0);
    const int32_t *hp = head_ptr ? head_ptr + n : loc;
    int2 rc;
    uint2 a0, a1;
    int32_t h;
    asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\ts_load_dwordx2 %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(rc), "=&s"(a0), "=&s"(a1), "=&s"(h)
                 : "s"(loc), "s"(base), "s"(base1), "s"(hp)
                 : "memory");
    head = head_ptr ? h : 0;
    LocIn in;
    in.r = rc.x;
    in.c = rc.y;
    in.w[0] = a0.x;
    in.w[1] = wide ? a0.y : 0u;
    in.w[2] = wide ? a1.x : a0.y;
    in.w[3] = wide ? a1.y : 0u;
    return in;
}
// ... and the flexible env's extra state (fov_res, sensory_action_type) the same way: six s_loads, one wait (one asm statement:
// no register of an s_load in flight may be visible to the compiler before the wait)
__device__ __forceinline__ LocIn load_flex_inputs_scalar(const FovParams &p, int n, int &head, int2 &res_old, int &type) {
    const bool wide = p.action_dt == AGX_DT_F64 || p.action_dt == AGX_DT_I64;
    const int32_t *loc = p.loc_in + 2 * n;
    const char *base = p.action ? static_cast<const char *>(p.action) + (size_t)n * (wide ? 16 : 8)
                                : reinterpret_cast<const char *>(loc);
    const char *base1 = base + ((wide && p.action) ? 8 : 0);
    const int32_t *hp = p.head + n;
    const int32_t *resp = p.res_in + 2 * n;
    const int32_t *typp = (p.action && p.action_type) ? p.action_type + n : loc;
    int2 rc, rs;
    uint2 a0, a1;
    int32_t h, ty;
    asm volatile("s_load_dwordx2 %0, %6, 0x0\n\ts_load_dwordx2 %1, %7, 0x0\n\ts_load_dwordx2 %2, %8, 0x0\n\ts_load_dword %3, %9, 0x0\n\t"
                 "s_load_dwordx2 %4, %10, 0x0\n\ts_load_dword %5, %11, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(rc), "=&s"(a0), "=&s"(a1), "=&s"(h), "=&s"(rs), "=&s"(ty)
                 : "s"(loc), "s"(base), "s"(base1), "s"(hp), "s"(resp), "s"(typp)
                 : "memory");
    head = h;
    res_old = rs;
    type = (p.action && p.action_type) ? ty : AGX_FOV_LOC;
    LocIn in;
    in.r = rc.x;
    in.c = rc.y;
    in.w[0] = a0.x;
    in.w[1] = wide ? a0.y : 0u;
    in.w[2] = wide ? a1.x : a0.y;
    in.w[3] = wide ? a1.y : 0u;
    return in;
}
__device__ __forceinline__ double action_value(int dt, uint32_t lo, uint32_t hi) {
    switch (dt) {
        case AGX_DT_F32: return (double)__uint_as_float(lo);
        case AGX_DT_F64: return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        case AGX_DT_I32: return (double)(int32_t)lo;
        default: return (double)(int64_t)(((uint64_t)hi << 32) | lo);
    }
}
__device__ __forceinline__ void compute_loc(const FovParams &p, const LocIn &in, int bound_r, int bound_c, int &r,
                                            int &c) {
    r = in.r;
    c = in.c;
    if (p.action) {
        const double ar = action_value(p.action_dt, in.w[0], in.w[1]);
        const double ac = action_value(p.action_dt, in.w[2], in.w[3]);
        if (p.relative) {
            const int dr = clip_rint(ar, p.sas_lo, p.sas_hi);
            const int dc = clip_rint(ac, p.sas_lo, p.sas_hi);
            r = clip_rint((double)(r + dr), 0.0, (double)bound_r);
            c = clip_rint((double)(c + dc), 0.0, (double)bound_c);
        } else {
            r = clip_rint(ar, 0.0, (double)bound_r);
            c = clip_rint(ac, 0.0, (double)bound_c);
        }
    }
}
__device__ __forceinline__ void next_loc(const FovParams &p, int n, int bound_r, int bound_c, int &r, int &c) {
    const LocIn in = load_loc_inputs(p, n);
    compute_loc(p, in, bound_r, bound_c, r, c);
}

// Stage the window [r, r+h) x [c, c+w) of one u8 frame (row pitch ow, ow % 4 == 0) into LDS as
// float32 k/255, tight pitch w.  Aligned dword loads; each thread peels the bytes it owns.
__device__ __forceinline__ void stage_window(const uint8_t *frame, int ow, int r, int c, int h, int w,
                                             float *dst, int tid) {
    const int c4 = c & ~3;
    const int wpr = ((c - c4) + w + 3) >> 2;          // dwords per row
    const int ntask = h * wpr;
    for (int task = tid; task < ntask; task += kThreads) {
        const int y = task / wpr, q = task - y * wpr;
        const int col = c4 + 4 * q;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(frame + (size_t)(r + y) * ow + col);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = col + b - c;
            if (x >= 0 && x < w) dst[y * w + x] = unit((v >> (8 * b)) & 0xFF);
        }
    }
}


// fov_loc / fov_res (re)initialisation for masked envs (fov_env.py:149-150,250-251)
struct FovResetParams {
    const uint8_t *mask;
    int32_t *loc;
    int32_t *res;     // may be nullptr
    int32_t init_r, init_c, fh, fw, n;
};
__global__ __launch_bounds__(kThreads) void k_fovea_reset(FovResetParams p) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= p.n) return;
    if (p.mask && !p.mask[n]) return;
    p.loc[2 * n] = p.init_r;
    p.loc[2 * n + 1] = p.init_c;
    if (p.res) {
        p.res[2 * n] = p.fh;
        p.res[2 * n + 1] = p.fw;
    }
}

}  // namespace agx
