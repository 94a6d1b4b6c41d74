// agx_k2_fixed.h - K2: FixedFovealEnv (k_fovea_fixed, the two-slot form, and the fused step launches).
#pragma once
#include "agx_fov_common.h"
#include "agx_k1_ingest.h"

namespace agx {

// ---------------------------------------------------------------------------------------------
// K2: FixedFovealEnv
// grid = (fs, N): one workgroup per (env, stacked frame); block = 256
//   MODE = AGX_OUT_RESIZE: LDS s[fh][fw] -> H[fh][ow] (horizontal lerp) -> float4 rows of the
//          84x84 output = vertical lerp of two ds_read_b128; every store is 16 B/lane, lane-linear.
// ---------------------------------------------------------------------------------------------
// The observation is a write-once stream of 115 MB per launch.  Rounds 1 / 2 wrote it with nontemporal stores, so that the
// next launch (K1) does not queue behind ~115 MB of dirty L2 / Infinity-Cache lines (K1 ran ~6 us slower after plain stores).
// Round 3 measured the cache-policy bits on this very store shape (tools/storebench.hip, 115.6 MB, same box): nt 20.1 us
// (5.76 TB/s), plain 18.0 (6.41), sc0 18.0, **sc1 17.0 us (6.80 TB/s)**, sc1 nt 19.6.  An agent-scope (sc1) store is written
// through the XCD's L2 towards memory at once - nothing stays dirty behind the launch either.  In the real step (same box,
// bench.py's kernel events / us per step): K2 22.0 -> 19.9 / 52.5-53.1 -> 51.0-51.2; K3 24.5-25.0 -> 22.6-23.8 / 55.4-56.0 ->
// 53.4-55.2; K4 24.2-25.0 -> 22.0-22.4 / 55.1-56.3 -> 53.7-53.8; K1 behind them unchanged.
// A raw buffer store carries the bit (aux 16 = sc1 on gfx940+) and stays an ordinary store for the compiler (an inline-asm store
// does not: the hazard recogniser cannot see that its four data VGPRs must not be overwritten by the very next VALU
// instruction, and the first sc1 build produced a few hundred wrong observation values per launch that way).  One buffer
// resource per workgroup = its output frame: the pointer is wave-uniform by construction, out-of-range offsets are dropped.
struct ObsOut {
    __amdgpu_buffer_rsrc_t rs;
#ifdef AGX_CANARY_ASM_OBS_STORE
    float4 *base;
#endif
};
__device__ __forceinline__ ObsOut obs_out(float4 *frame, int n_float4) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(frame);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    void *p = reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo);
    ObsOut o;
    o.rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, n_float4 * 16, 0x00027000);
#ifdef AGX_CANARY_ASM_OBS_STORE
    o.base = frame;
#endif
    return o;
}
__device__ __forceinline__ void store_obs(const ObsOut &o, int q, const float4 &v) {
#ifdef AGX_CANARY_ASM_OBS_STORE
    // The KNOWN-BAD store of commit 327a14a, kept as a canary for the tests only (build.py: build_canary() ->
    // lib/libagx_canary.so, never loaded by the product): an inline-asm store is invisible to the compiler's hazard
    // recogniser, and on gfx940+ the data VGPRs of a store of more than 64 bits must not be overwritten by the VALU
    // instructions right behind it.  tests/test_gpu_lowocc.py must FAIL on this build (tools/canary_probe.py shows it).
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o.base + q), "v"(w));
#else
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(w, o.rs, q * 16, 0, 16 /* sc1 */);
#endif
}

// one float of a raw crop (fixed: [fh][fw]; flexible, packed: [rh][rw] at an arbitrary 4-byte aligned offset - hence dword stores),
// written through like the full-size observations: the crop of one stacked frame is the buffer
struct PackedOut {
    __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ PackedOut packed_out(float *crop, int n_floats) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(crop);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    void *q = reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo);
    PackedOut o;
    o.rs = __builtin_amdgcn_make_buffer_rsrc(q, 0, crop ? n_floats * 4 : 0, 0x00027000);
    return o;
}
__device__ __forceinline__ void store_packed(const PackedOut &o, int i, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o.rs, i * 4, 0, 16 /* sc1 */);
}
// grid = (fs, N): workgroup (sl, n) owns PHYSICAL ring slot sl of env n, block = 256.
// Prologue: the small state loads (action, fov_loc, head) and this thread's taps go out first; the scalar chain action ->
// rint(clip(..)) -> (r, c) runs as soon as the state has arrived and moves to SGPRs; then ONLY the window of the slot is
// fetched, for every MODE: its fh rows (LDS image u8 [fh][ow]).  u8 -> float32 k/255 is unit_fast (3 FMAs, the correctly
// rounded quotient).
//   RESIZE: H[fh][ow] = horizontal lerp of the window rows (thread = fixed column x, rows y = yb+3k),
//           then each output float4 is the vertical lerp of two ds_read_b128; stores are 16 B per
//           lane, lane-linear, 1 KiB per wave at 1-KiB steps, written through (sc1, store_obs above).
// (ablation of the previous serial version at N=1024: loc chain 5.8 us, loc-dependent window load
//  6.3 us, H pass with a tap load per iteration 7.3 us, row-tap loads 2.1 us of a 32.7 us launch.)
// COHERENT: the frame is read with agent-scope loads (global_load_dword sc1: served by L2, never by this CU's L1) - for a
// slot that the same launch has just written (k_step_env); phase 3: `head` is the pre-ingest head, every slot is processed.
template <class G, int MODE, bool COHERENT = false>
__device__ __forceinline__ void fovea_fixed_body(const G g, const FovParams &p, const int sl, const int n,
                                                 unsigned char *smem, const int tid) {
    constexpr int T = kThreads;
    (void)T;
    AGX_STAMP(0);
    const int oh = g.oh(), ow = g.ow(), fh = g.fh(), fw = g.fw();
    if (p.mask && !p.mask[n]) {
        if (sl == 0 && tid < 2) p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
        return;
    }
    int head_fixup = 0;                      // what to add to p.head[n] to get the post-ingest head
    if (p.phase != 0) {
        const uint32_t cmd = uniform_load_u8(p.cmd + n);
        const bool skip = (cmd & AGX_CMD_SKIP) != 0, clear = (cmd & AGX_CMD_CLEAR) != 0 && !skip;
        const int h = uniform_load_i32(p.head + n);
        // slot the ingest writes: the pre-ingest head (fs-1 after a clear, which also zeroes the others)
        int wslot;
        if (p.phase == 1 || p.phase == 3) {
            wslot = h;
            head_fixup = skip ? 0 : (clear ? -h : (h + 1 == p.fs ? 1 - p.fs : 1));
        } else {
            wslot = skip ? h : (h == 0 ? p.fs - 1 : h - 1);
        }
        const bool touched = clear || sl == wslot;
        if (p.phase != 3 && (p.phase == 1) == touched) return;   // phase 1 takes the untouched slots, phase 2 the rest
    }
    // LDS carve: window rows u8 [fh][ow] (16-B padded) | ytab[oh] | H[fh][ow]     (agx_api.hip: fixed_lds)
    unsigned char *raw = smem;
    const int fbytes = oh * ow;                                       // multiple of 4 (ow % 4 == 0)
    const int raw_pad = (fh * ow + 15) & ~15;
    Tap *ytab_s = reinterpret_cast<Tap *>(raw + raw_pad);
    float *H = reinterpret_cast<float *>(ytab_s + oh);

    // ---- State first (vmcnt retires in order), then only the fh window rows of the slot (2.5 KB of the 7 KB frame), for every
    // MODE: every resident workgroup of the launch starts with this burst, and a third of the bytes returns sooner than the
    // extra dependent round trip costs (round 1 fetched the whole frame to avoid that dependency: 3.6 us of a wave's 6.7 us
    // life were the load chain; K2 23.1 -> 21.9 us).  Round 3 also built the narrower form - of each row only the
    // dword-aligned column span that holds [c, c + fw), 1.1 KB - and measured a tie (23.4-23.5 us both, same box): the rows of
    // a window are 84 bytes apart, so the span touches the same cache lines as the whole rows; not kept.
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    int r, c, j;
    int4 xt = make_int4(0, 0, 0, 0);                                  // this thread's column taps {lo, aux, a, b}
    const int wp = ow;                                                // row pitch of the LDS image
    {
        // this thread's taps go out first (vector loads, in flight through the scalar wait below), then the state through the
        // scalar cache
        int4 yt0 = make_int4(0, 0, 0, 0);
        if (MODE == AGX_OUT_RESIZE) {
            xt = *reinterpret_cast<const int4 *>(p.xtab + tid % ow);
            yt0 = *reinterpret_cast<const int4 *>(p.ytab + min(tid, oh - 1));
        }
        int head0;
        const LocIn lin = load_loc_inputs_scalar(p, n, p.head, head0);
        const int head = head0 + head_fixup;
        compute_loc(p, lin, oh - fh, ow - fw, r, c);
        r = __builtin_amdgcn_readfirstlane(r);
        c = __builtin_amdgcn_readfirstlane(c);
        j = sl - __builtin_amdgcn_readfirstlane(head);
        if (j < 0) j += p.fs;
        const uint32_t *wsrc = fsrc + r * (ow >> 2);
        const int wwords = (fh * ow) >> 2;
        constexpr int kW = 3;
        uint32_t ww[kW];
#pragma unroll
        for (int k = 0; k < kW; ++k) {
            const uint32_t *q = wsrc + min(tid + k * kThreads, wwords - 1);
            ww[k] = COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
        if (sl == 0 && tid == 0) {
            p.loc_out[2 * n] = r;
            p.loc_out[2 * n + 1] = c;
            if (p.user_loc) {
                p.user_loc[2 * n] = r;
                p.user_loc[2 * n + 1] = c;
            }
        }
        if (MODE == AGX_OUT_RESIZE) {
            if (tid < oh) *reinterpret_cast<int4 *>(ytab_s + tid) = yt0;
            for (int i = tid + kThreads; i < oh; i += kThreads) ytab_s[i] = p.ytab[i];
        }
#pragma unroll
        for (int k = 0; k < kW; ++k)
            if (tid + k * kThreads < wwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = ww[k];
        for (int i = tid + kW * kThreads; i < wwords; i += kThreads)
            reinterpret_cast<uint32_t *>(raw)[i] = COHERENT ? __hip_atomic_load(wsrc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : wsrc[i];
        AGX_STAMP(1);
        __syncthreads();
        AGX_STAMP(2);
    }
    const int xcol = tid % ow, yb = tid / ow;                         // phase-C column / first row
    const unsigned char *win = raw + c;                               // window origin inside the LDS image (row r of the frame = row 0)
    if (MODE == AGX_OUT_RAW) {
        const PackedOut cout = packed_out(p.obs + ((size_t)n * p.fs + j) * (size_t)(fh * fw), fh * fw);
        for (int i = tid; i < fh * fw; i += kThreads) {
            const int y = i / fw, x = i - y * fw;
            store_packed(cout, i, unit_fast((float)win[y * wp + x]));
        }
        return;
    }
    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);
    if (MODE == AGX_OUT_MASK) {
        for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
            const int q = tid + k_ * kThreads;
            if (q >= oh * ow4) break;
            const int row = q / ow4, x = (q - row * ow4) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row >= r && row < r + fh && x + 3 >= c && x < c + fw) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(raw + (row - r) * wp + x);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k >= c && x + k < c + fw) v[k] = unit_fast((float)((w >> (8 * k)) & 0xFF));
            }
            store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
        }
        return;
    }
    // ---- RESIZE, phase C: thread owns column xcol (taps in registers), rows yb, yb + rstep, ...
    const int rstep = kThreads / ow;                                  // 3 for ow = 84
    if (rstep > 0) {
        if (yb < rstep) {
            const unsigned char *c0 = win + xt.x, *c1 = win + xt.y;
            const float wa = __int_as_float(xt.z), wb = __int_as_float(xt.w);
#pragma unroll 10
            for (int y = yb; y < fh; y += rstep)
                // u8 -> float32 k/255 by unit_fast (3 FMAs, the correctly rounded quotient) rather than through the LDS table:
                // one LDS round trip less in the chain byte -> value -> lerp (23.0-23.4 vs 23.2-23.8 us, a tie at worst)
                H[y * ow + xcol] = fmaf(wb, unit_fast((float)c1[y * wp]), wa * unit_fast((float)c0[y * wp]));
        }
    } else {                                                          // ow > 256: generic striding
        for (int i = tid; i < fh * ow; i += kThreads) {
            const int y = i / ow, x = i - y * ow;
            const Tap t = p.xtab[x];
            H[i] = fmaf(t.b, unit_fast((float)win[y * wp + t.aux]), t.a * unit_fast((float)win[y * wp + t.lo]));
        }
    }
    __syncthreads();
    AGX_STAMP(3);
    // ---- phase D
    const float4 *H4 = reinterpret_cast<const float4 *>(H);
    // (a uniform trip count with the bound tested inside: inline asm is convergent, and a loop whose trip count differs per
    //  thread cannot be unrolled around it - with the compile-time geometry all 7 passes unroll and their LDS reads batch up)
    const int nq = oh * ow4, passes = (nq + kThreads - 1) / kThreads;
#pragma unroll 7
    for (int k = 0; k < passes; ++k) {
        const int q = tid + k * kThreads;
        if (q >= nq) break;
        const int row = q / ow4, x4 = q - row * ow4;
        const Tap t = ytab_s[row];
        const float4 a = H4[t.lo * ow4 + x4];
        const float4 b = H4[t.aux * ow4 + x4];
        // the lerp is written as mul + fma explicitly: every instantiation of this body (stand-alone, pair, fused, per-env
        // step) then rounds the same way whatever contraction the optimiser would pick in its context
        float4 o;
        o.x = fmaf(t.b, b.x, t.a * a.x);
        o.y = fmaf(t.b, b.y, t.a * a.y);
        o.z = fmaf(t.b, b.z, t.a * a.z);
        o.w = fmaf(t.b, b.w, t.a * a.w);
        store_obs(oout, q, o);
    }
    AGX_STAMP(4);
}

template <class G, int MODE>
__device__ __forceinline__ void fovea_fixed_body(const G g, const FovParams &p, const int sl, const int n,
                                                 unsigned char *smem) {
    fovea_fixed_body<G, MODE, false>(g, p, sl, n, smem, (int)threadIdx.x);
}

template <class G, int MODE>
__global__ __launch_bounds__(kThreads) void k_fovea_fixed(G g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fovea_fixed_body<G, MODE>(g, p, blockIdx.x, blockIdx.y, smem);
}

}  // namespace agx
