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
template <class T4>
__device__ __forceinline__ void store_obs(T4 *dst, const T4 &v) {
#ifndef AGX_K2_PLAIN_STORES
    // write-once stream: nontemporal, so the next launch (K1) does not queue behind ~115 MB of dirty
    // L2 / Infinity-Cache lines (measured: K1 is ~6 us faster after nontemporal obs stores)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f4v *>(dst));
#else
    *dst = v;
#endif
}

// grid = (fs, N): workgroup (sl, n) owns PHYSICAL ring slot sl of env n, block = 256.
// Every stage that costs a memory round trip is started at once:
//   * the u8 frame of that slot -> registers -> LDS (resize_to_full: only its fh window rows, requested as soon as the
//     env's state has given the window's first row; mask-out / raw crop: the whole frame, address known at launch),
//   * the scalar chain action / fov_loc / head -> (r, c) and the stack position j of this slot,
//   * this thread's column taps (registers) and one row-tap entry (-> LDS).
// u8 -> float32 k/255 goes through a 256-entry LDS table (one exact division per thread).
//   RESIZE: H[fh][ow] = horizontal lerp of the window rows (thread = fixed column x, rows y = yb+3k),
//           then each output float4 is the vertical lerp of two ds_read_b128; stores are 16 B per
//           lane, lane-linear, 1 KiB per wave at 1-KiB steps, nontemporal.
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
    // LDS carve: lut[256] | raw[oh*ow] u8 | ytab[oh] | H[fh][ow]
    float *lut = reinterpret_cast<float *>(smem);
    unsigned char *raw = smem + 1024;
    const int fbytes = oh * ow;                                       // multiple of 4 (ow % 4 == 0)
    const int raw_pad = (fbytes + 15) & ~15;
    Tap *ytab_s = reinterpret_cast<Tap *>(raw + raw_pad);
    float *H = reinterpret_cast<float *>(ytab_s + oh);

    // ---- every round trip starts now: the frame, the taps, then the small state loads.  (The first
    // use of the state waits for everything older too, which is fine: all of it is needed before the
    // LDS image can be written; what matters is that nothing waits before everything is issued.)
    // RESIZE: only the fh window rows of the slot are fetched, after the state (see below); -DAGX_K2_FULL_FRAME restores the
    // whole-frame prologue of round 1 (same box, N = 1024: K2 23.0-23.2 -> 21.7-22.1 us, step 60.7 -> 59.4 us)
#if defined(AGX_K2_FULL_FRAME)
    constexpr bool kWindowOnly = false;
#else
    constexpr bool kWindowOnly = true;
#endif
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    int r, c, j;
    int4 xt = make_int4(0, 0, 0, 0);                                  // this thread's column taps {lo, aux, a, b}
    if (kWindowOnly) {
        // State first (vmcnt retires in order), then only the fh window rows of the slot (2.5 KB of the 7 KB frame): every
        // resident workgroup of the launch starts with this burst, and a third of the bytes returns sooner than the extra
        // dependent round trip costs (round 1 fetched the whole frame to avoid that dependency: 3.6 us of a wave's 6.7 us
        // life were the load chain)
        const LocIn lin = load_loc_inputs(p, n);
        const int head = p.head[n] + head_fixup;
        int4 yt0 = make_int4(0, 0, 0, 0);
        if (MODE == AGX_OUT_RESIZE) {
            xt = *reinterpret_cast<const int4 *>(p.xtab + tid % ow);
            yt0 = *reinterpret_cast<const int4 *>(p.ytab + min(tid, oh - 1));
        }
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
        __syncthreads();
    }
    const int r_img = kWindowOnly ? r : 0;                            // frame row held by row 0 of the LDS image
    const int xcol = tid % ow, yb = tid / ow;                         // phase-C column / first row
    if (!kWindowOnly) {
    const int fwords = fbytes >> 2;
    constexpr int kFW = 7;                                            // 7 * 256 dwords cover 84x84; loop beyond
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k) {
        const uint32_t *q = fsrc + min(tid + k * kThreads, fwords - 1);
        fw_[k] = COHERENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
    int4 yt = make_int4(0, 0, 0, 0);                                  // raw Tap bits {lo, aux, a, b}
    if (MODE == AGX_OUT_RESIZE) {
        xt = *reinterpret_cast<const int4 *>(p.xtab + xcol);
        yt = *reinterpret_cast<const int4 *>(p.ytab + min(tid, oh - 1));
    }
    const LocIn lin = load_loc_inputs(p, n);
    const int head = p.head[n] + head_fixup;
    lut[tid] = unit((uint32_t)tid);
    compute_loc(p, lin, oh - fh, ow - fw, r, c);
#ifndef AGX_K2_VECTOR_STATE
    // workgroup-uniform: scalar registers from here on (window base, output base)
    r = __builtin_amdgcn_readfirstlane(r);
    c = __builtin_amdgcn_readfirstlane(c);
    j = sl - __builtin_amdgcn_readfirstlane(head);                    // stack position of this slot
#else
    j = sl - head;
#endif
    if (j < 0) j += p.fs;
    if (sl == 0 && tid == 0) {
        p.loc_out[2 * n] = r;
        p.loc_out[2 * n + 1] = c;
        if (p.user_loc) {
            p.user_loc[2 * n] = r;
            p.user_loc[2 * n + 1] = c;
        }
    }
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (tid + k * kThreads < fwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = fw_[k];
    for (int i = tid + kFW * kThreads; i < fwords; i += kThreads)
        reinterpret_cast<uint32_t *>(raw)[i] = COHERENT ? __hip_atomic_load(fsrc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : fsrc[i];
    if (MODE == AGX_OUT_RESIZE) {
        if (tid < oh) *reinterpret_cast<int4 *>(ytab_s + tid) = yt;
        for (int i = tid + kThreads; i < oh; i += kThreads) ytab_s[i] = p.ytab[i];
    }
    AGX_STAMP(1);
    __syncthreads();
    AGX_STAMP(2);

    }
    const unsigned char *win = raw + (r - r_img) * ow + c;            // window origin inside the LDS image
    if (MODE == AGX_OUT_RAW) {
        float *out = p.obs + ((size_t)n * p.fs + j) * (size_t)(fh * fw);
        for (int i = tid; i < fh * fw; i += kThreads) {
            const int y = i / fw, x = i - y * fw;
            out[i] = unit_fast((float)win[y * ow + x]);
        }
        return;
    }
    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    if (MODE == AGX_OUT_MASK) {
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = q / ow4, x = (q - row * ow4) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row >= r && row < r + fh && x + 3 >= c && x < c + fw) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(raw + (row - r_img) * ow + x);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k >= c && x + k < c + fw) v[k] = unit_fast((float)((w >> (8 * k)) & 0xFF));
            }
            store_obs(&out4[q], make_float4(v[0], v[1], v[2], v[3]));
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
                H[y * ow + xcol] = fmaf(wb, unit_fast((float)c1[y * ow]), wa * unit_fast((float)c0[y * ow]));
        }
    } else {                                                          // ow > 256: generic striding
        for (int i = tid; i < fh * ow; i += kThreads) {
            const int y = i / ow, x = i - y * ow;
            const Tap t = p.xtab[x];
            H[i] = fmaf(t.b, unit_fast((float)win[y * ow + t.aux]), t.a * unit_fast((float)win[y * ow + t.lo]));
        }
    }
    __syncthreads();
    AGX_STAMP(3);
    // ---- phase D
    const float4 *H4 = reinterpret_cast<const float4 *>(H);
#pragma unroll 7
    for (int q = tid; q < oh * ow4; q += kThreads) {
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
        store_obs(&out4[q], o);
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

// env-range part of a split step (agx_step_fixed): same body, a name of its own in kernel traces
template <class G, int MODE>
__global__ __launch_bounds__(kThreads) void k_fovea_fixed_part(G g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fovea_fixed_body<G, MODE>(g, p, blockIdx.x, blockIdx.y, smem);
}

// ---------------------------------------------------------------------------------------------
// K2, two slots per workgroup (resize_to_full, stand-alone launch): grid = (fs/2, N), block = 256.
// The occupancy timeline of the one-slot form shows two synchronized rounds of workgroups, each wave
// spending 54 % of its life on the load chain.  Here a workgroup requests BOTH of its frames up front
// and keeps the second in registers while the first goes LDS -> H -> stores, so the second frame's load
// latency is hidden and the whole launch is resident at once (2048 workgroups x 4 waves at N=1024).
// ---------------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(kThreads) void k_fovea_fixed2(G g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh(), fw = g.fw();
    const int sl0 = 2 * blockIdx.x;
    if (p.mask && !p.mask[n]) {
        if (sl0 == 0 && tid < 2) p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
        return;
    }
    float *lut = reinterpret_cast<float *>(smem);
    unsigned char *raw = smem + 1024;
    const int fbytes = oh * ow, fwords = fbytes >> 2;
    const int raw_pad = (fbytes + 15) & ~15;
    Tap *ytab_s = reinterpret_cast<Tap *>(raw + raw_pad);
    float *H = reinterpret_cast<float *>(ytab_s + oh);
    constexpr int kFW = 7;
    uint32_t fa[kFW], fb[kFW];
    const uint32_t *src0 = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl0) * (size_t)fbytes);
    const uint32_t *src1 = src0 + fwords;
#pragma unroll
    for (int k = 0; k < kFW; ++k) fa[k] = src0[min(tid + k * kThreads, fwords - 1)];
#pragma unroll
    for (int k = 0; k < kFW; ++k) fb[k] = src1[min(tid + k * kThreads, fwords - 1)];
    const int xcol = tid % ow, yb = tid / ow;
    const int4 xt = *reinterpret_cast<const int4 *>(p.xtab + xcol);
    const int4 yt = *reinterpret_cast<const int4 *>(p.ytab + min(tid, oh - 1));
    const LocIn lin = load_loc_inputs(p, n);
    const int head = p.head[n];
    lut[tid] = unit((uint32_t)tid);
    int r, c;
    compute_loc(p, lin, oh - fh, ow - fw, r, c);
    if (sl0 == 0 && tid == 0) {
        p.loc_out[2 * n] = r;
        p.loc_out[2 * n + 1] = c;
        if (p.user_loc) {
            p.user_loc[2 * n] = r;
            p.user_loc[2 * n + 1] = c;
        }
    }
    if (tid < oh) *reinterpret_cast<int4 *>(ytab_s + tid) = yt;
    for (int i = tid + kThreads; i < oh; i += kThreads) ytab_s[i] = p.ytab[i];
    const int ow4 = ow >> 2;
    const int rstep = kThreads / ow;
    const float wa = __int_as_float(xt.z), wb = __int_as_float(xt.w);
    const float4 *H4 = reinterpret_cast<const float4 *>(H);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                                    // raw / H of the first frame are consumed
#pragma unroll
        for (int k = 0; k < kFW; ++k)
            if (tid + k * kThreads < fwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = half ? fb[k] : fa[k];
        if (kFW * kThreads < fwords) {
            const uint32_t *src = half ? src1 : src0;
            for (int i = tid + kFW * kThreads; i < fwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = src[i];
        }
        __syncthreads();
        const unsigned char *win = raw + r * ow + c;
        if (rstep > 0) {
            if (yb < rstep) {
                const unsigned char *c0 = win + xt.x, *c1 = win + xt.y;
#pragma unroll 10
                for (int y = yb; y < fh; y += rstep) H[y * ow + xcol] = fmaf(wb, lut[c1[y * ow]], wa * lut[c0[y * ow]]);
            }
        } else {
            for (int i = tid; i < fh * ow; i += kThreads) {
                const int y = i / ow, x = i - y * ow;
                const Tap t = p.xtab[x];
                H[i] = fmaf(t.b, lut[win[y * ow + t.aux]], t.a * lut[win[y * ow + t.lo]]);
            }
        }
        __syncthreads();
        int j = sl0 + half - head;
        if (j < 0) j += p.fs;
        float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
#pragma unroll 7
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = q / ow4, x4 = q - row * ow4;
            const Tap t = ytab_s[row];
            const float4 a = H4[t.lo * ow4 + x4];
            const float4 b = H4[t.aux * ow4 + x4];
            float4 o;
            o.x = fmaf(t.b, b.x, t.a * a.x);
            o.y = fmaf(t.b, b.y, t.a * a.y);
            o.z = fmaf(t.b, b.z, t.a * a.z);
            o.w = fmaf(t.b, b.w, t.a * a.w);
            store_obs(&out4[q], o);
        }
    }
}

// Fused step, second launch: grid = (1, N).  One workgroup per env processes the ring slot the ingest
// has just written; for the rare cleared env (full reset: every slot changed) it walks all of them.
template <class G>
__global__ __launch_bounds__(kThreads) void k_step_fixed_tail(G g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y;
    if (p.mask && !p.mask[n]) return;        // (the fused step never passes a mask; kept for symmetry)
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0, clear = (cmd & AGX_CMD_CLEAR) != 0 && !skip;
    const int h = uniform_load_i32(p.head + n);
    const int wslot = skip ? h : (h == 0 ? p.fs - 1 : h - 1);
    FovParams q = p;
    q.phase = 0;                             // `head` is already the post-ingest head
    if (!clear) {
        fovea_fixed_body<G, AGX_OUT_RESIZE>(g, q, wslot, n, smem);
        // slot 0 is the one that publishes fov_loc; it ran in the first launch unless it is the written slot
        return;
    }
    for (int sl = 0; sl < p.fs; ++sl) {
        if (sl) __syncthreads();             // the LDS image of the previous slot has been consumed
        fovea_fixed_body<G, AGX_OUT_RESIZE>(g, q, sl, n, smem);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused step, first launch: grid = (bands + fs, N), block = 256.  Workgroups x < bands ingest band x
// of env n; workgroups x >= bands run the resize_to_full fovea of ring slot x - bands, but only for
// the slots this step's ingest does not touch (phase 1).  The two kinds of workgroup are independent
// (disjoint ring slots, double-buffered head / fov_loc), so the store-bound fovea work fills the
// issue slots and the drain of the load/ALU-bound ingest (K1 alone: CUs run dry for its last 9 us).
// The written slot follows in a second, small launch of k_fovea_fixed with phase 2.
// ---------------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(kThreads) void k_step_fixed(G g, IngestParams pi, FovParams pf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int x = blockIdx.x, n = blockIdx.y;
    if (x < pi.nbands)
        ingest_band<kThreads>(pi, x, n, smem);
    else
        fovea_fixed_body<G, AGX_OUT_RESIZE>(g, pf, x - pi.nbands, n, smem);
}

}  // namespace agx
