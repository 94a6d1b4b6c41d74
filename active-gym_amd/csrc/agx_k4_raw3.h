// agx_k4_raw3.h - K4, raw-crop / mask-out / packed-ragged forms: FlexibleFovealEnv._fov_step + _get_fov_state
// (fov_env.py:270-330) without the final Resize(obs_size), on the composed-operator plan of agx_k4_flex3.h.
//
//   reference, per env and stacked frame:   crop[rh][rw]
//       iff rh > fov_h:  Resize(fov_size) -> Resize(fov_res)                 fov_env.py:276-287 (rows-only test)
//       mask_out: zeros[oh][ow] with the crop pasted at (r, c)   :289-293
//       raw     : the ragged crop [rh][rw] itself                             :296-298
//   here:  rh <= fov_h :  out = crop / 255 (exact k/255)
//          rh >  fov_h :  D[fh][rw] = Hdwn(rh) . crop            <= 4 / 8 taps, reads the u8 window (1/255 in the weights)
//                         E[fh][rw] = D . (Wbck Wdwn)(rw)^T      composed on the host: <= 4 / 8 / 16 taps
//                         out[rh][rw] = Hbck(rh) . E             2 taps (an up-scale)
//
// Output forms (OUT): kRawPacked - env n's crops [fs][rh][rw] tight at packed + offsets[n] (agx_fovea_flexible_packed);
// AGX_OUT_RAW - the crop at the origin of a zeroed [oh][ow] frame (padded batch); AGX_OUT_MASK - pasted at (r, c).
//
// Packed layout: offsets[] is an exclusive scan of fs * rh * rw over the envs' NEW resolutions, so the state update runs
// first, as k_flex_state_scan: grid = ceil(N / 256) workgroups, one env per thread; each writes the env-local exclusive
// offsets of its 256 envs and its block total.  The crop launch then only reads the final state; a workgroup of a later
// block adds the totals of the blocks before it (<= 255 values: four per lane and one wave reduction).  Two launches, no allocation,
// any N <= 65,535.
#pragma once
#include "agx_fov_common.h"
#include "agx_k2_fixed.h"
#include "agx_k4_flex3.h"

namespace agx {

constexpr int kRawPacked = 100;

constexpr int kScanEnvsPerBlock = kThreads;      // one env per thread of a 256-thread scan workgroup

struct FlexRawParams {
    const int2 *wb_meta;      // [ow + 1]       {T, first float of that size's table in wb_w}, T in {4, 8, 16}
    const int32_t *wb_lo;     // [ow + 1][ow]   first D column of the composed (Wbck Wdwn) operator (0 beyond rw)
    const float *wb_w;        // per size: [ow][T] (zero rows beyond rw)
    const int2 *hd_meta;      // [oh + 1]       {T, first float in hd_w}, T in {4, 8}
    const int32_t *hd_lo;     // [oh + 1][fh]   first window row of the H squeeze
    const float *hd_w;        // per size: [fh][T]; weights carry the 1/255
    const Tap *hb;            // [oh + 1][oh]   Hbck(rh): rows of E -> rows of the crop, {i0, i1, w0, w1}
    int32_t r0_bytes, r1_bytes;
    int32_t dp;               // pitch of D in floats (multiple of 8, >= ow)
    // packed form
    const int64_t *local_off; // [N]  exclusive offset of env n inside its scan block (kScanEnvsPerBlock envs)
    const int64_t *block_tot; // [ceil(N / kScanEnvsPerBlock)]
    int64_t *offsets;         // [N + 1] out (written by the crop launch)
    int32_t n_envs;
};

// ---- launch 1 of the packed form: state update of every env (fov_env.py:300-324) + two-level exclusive scan, level 1
struct FlexScanParams {
    FovParams f;
    int64_t *local_off;
    int64_t *block_tot;
    int32_t n, oh, ow;
};
// one scan block (256 envs, one per thread); wave_tot: kThreads / 64 int64 of this workgroup's LDS
__device__ __forceinline__ void flex_state_scan_block(const FlexScanParams &q, const int block, int64_t *wave_tot) {
    const FovParams &p = q.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = block * kScanEnvsPerBlock + tid;
    int64_t mine = 0;
    if (n < q.n) {
        const LocIn lin = load_loc_inputs(p, n);
        const int2 res_old = *reinterpret_cast<const int2 *>(p.res_in + 2 * n);
        const int type = (p.action && p.action_type) ? p.action_type[n] : AGX_FOV_LOC;
        int rh = min(max(res_old.x, 1), q.oh), rw = min(max(res_old.y, 1), q.ow), r, c;
        if (p.action && type == AGX_FOV_RES) {
            rh = clip_rint(action_value(p.action_dt, lin.w[0], lin.w[1]), 1.0, (double)q.oh);
            rw = clip_rint(action_value(p.action_dt, lin.w[2], lin.w[3]), 1.0, (double)q.ow);
            r = clip_rint((double)lin.r, 0.0, (double)(q.oh - rh));
            c = clip_rint((double)lin.c, 0.0, (double)(q.ow - rw));
        } else {
            compute_loc(p, lin, q.oh - rh, q.ow - rw, r, c);
        }
        *reinterpret_cast<int2 *>(p.loc_out + 2 * n) = make_int2(r, c);
        *reinterpret_cast<int2 *>(p.res_out + 2 * n) = make_int2(rh, rw);
        if (p.user_loc) *reinterpret_cast<int2 *>(p.user_loc + 2 * n) = make_int2(r, c);
        if (p.user_res) *reinterpret_cast<int2 *>(p.user_res + 2 * n) = make_int2(rh, rw);
        mine = (int64_t)p.fs * rh * rw;
    }
    int64_t incl = mine;                                        // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int64_t before = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w)
        if (w < wave) before += wave_tot[w];
    if (n < q.n) q.local_off[n] = before + incl - mine;
    if (tid == kThreads - 1) q.block_tot[block] = before + incl;
}
__global__ __launch_bounds__(kThreads) void k_flex_state_scan(FlexScanParams q) {
    __shared__ int64_t wave_tot[kThreads / 64];
    flex_state_scan_block(q, blockIdx.x, wave_tot);
}

// fallback of the packed form for geometries outside the raw3 plan: offsets[] from the two scan levels (grid = ceil((N+1)/256))
__global__ __launch_bounds__(kThreads) void k_flex_finish_offsets(const int64_t *local_off, const int64_t *block_tot,
                                                                  int64_t *offsets, int n) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i > n) return;
    const int b = min(i, n - 1) / kScanEnvsPerBlock;
    int64_t base = 0;
    for (int k = 0; k < (i == n ? b + 1 : b); ++k) base += block_tot[k];
    offsets[i] = i == n ? base : base + local_off[i];
}

template <class G, int OUT>
__global__ __launch_bounds__(kThreads) void k_fovea_flexible_raw3(G g, FlexRawParams t, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool PACKED = OUT == kRawPacked;
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh();
    if (!PACKED && p.mask && !p.mask[n]) {
        if (sl == 0 && tid < 2) {
            p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
            p.res_out[2 * n + tid] = p.res_in[2 * n + tid];
        }
        return;
    }
    const int fbytes = oh * ow;
    unsigned char *raw = smem;
    float *R0 = reinterpret_cast<float *>(smem);
    float *R1 = reinterpret_cast<float *>(smem + t.r0_bytes);
    Tap *ytab_s = reinterpret_cast<Tap *>(smem + t.r0_bytes + t.r1_bytes);

    // ---- the env's state.  Packed form: final already (k_flex_state_scan wrote it), read through the scalar cache.
    int rh, rw, r, c, head;
    int64_t poff = 0;
    if (PACKED) {
        // one batch of (wave-uniform) loads, one wait: final state, ring head, this env's block-local offset ...
        const int2 rc = *(reinterpret_cast<const int2 *>(p.loc_in) + n);
        const int2 hw = *(reinterpret_cast<const int2 *>(p.res_in) + n);
        const int hd = p.head[n];
        int64_t off = t.local_off[n];
        // ... and the totals of the scan blocks before this env's (at most 255: four per lane, one wave reduction)
        const int b = n / kScanEnvsPerBlock;
        if (b > 0) {
            int64_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = (tid & 63) + 64 * k;
                if (i < b) v += t.block_tot[i];
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            off += v;
        }
        r = __builtin_amdgcn_readfirstlane(rc.x), c = __builtin_amdgcn_readfirstlane(rc.y);
        rh = __builtin_amdgcn_readfirstlane(hw.x), rw = __builtin_amdgcn_readfirstlane(hw.y);
        head = __builtin_amdgcn_readfirstlane(hd);
        poff = ((int64_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
        if (sl == 0 && tid == 0) {
            t.offsets[n] = poff;
            if (n == t.n_envs - 1) t.offsets[n + 1] = poff + (int64_t)p.fs * rh * rw;
        }
    } else {
        int type;
        int2 res_old;
        const LocIn lin = load_flex_inputs_scalar(p, n, head, res_old, type);
        rh = min(max(res_old.x, 1), oh), rw = min(max(res_old.y, 1), ow);
        if (p.action && type == AGX_FOV_RES) {
            rh = clip_rint(action_value(p.action_dt, lin.w[0], lin.w[1]), 1.0, (double)oh);
            rw = clip_rint(action_value(p.action_dt, lin.w[2], lin.w[3]), 1.0, (double)ow);
            r = clip_rint((double)lin.r, 0.0, (double)(oh - rh));
            c = clip_rint((double)lin.c, 0.0, (double)(ow - rw));
        } else {
            compute_loc(p, lin, oh - rh, ow - rw, r, c);
        }
        rh = __builtin_amdgcn_readfirstlane(rh);
        rw = __builtin_amdgcn_readfirstlane(rw);
        r = __builtin_amdgcn_readfirstlane(r);
        c = __builtin_amdgcn_readfirstlane(c);
        head = __builtin_amdgcn_readfirstlane(head);
        if (sl == 0 && tid == 0) {
            *reinterpret_cast<int2 *>(p.loc_out + 2 * n) = make_int2(r, c);
            *reinterpret_cast<int2 *>(p.res_out + 2 * n) = make_int2(rh, rw);
            if (p.user_loc) *reinterpret_cast<int2 *>(p.user_loc + 2 * n) = make_int2(r, c);
            if (p.user_res) *reinterpret_cast<int2 *>(p.user_res + 2 * n) = make_int2(rh, rw);
        }
    }
    int j = sl - head;
    if (j < 0) j += p.fs;
    const int cnt = rh * rw;
    if (PACKED && poff + (int64_t)p.fs * cnt > p.packed_cap) return;   // the caller's buffer is too small for this env
    const bool squeeze = rh > fh;                                     // rows only, fov_env.py:286
    const uint8_t *frame = p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes;
    const int ow4 = ow >> 2;
    float *pdst = PACKED ? p.packed + poff + (int64_t)j * cnt : nullptr;
    // packed crops are a write-once stream like the observations: written through (sc1), one buffer resource per crop
    const PackedOut pout = packed_out(pdst, cnt);
    float4 *out4 = PACKED ? nullptr : reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);
    const int pr = OUT == AGX_OUT_MASK ? r : 0, pc = OUT == AGX_OUT_MASK ? c : 0;   // where the crop lands in a full frame
    const FastDiv dv_rw(rw);

    // ---- the window -> LDS: rows [r, r + rh) (squeeze path: + 8 rows of slack, read with zero weights, clipped to the frame),
    // of each row the dword-aligned column span that holds [c, c + rw); one round trip for the whole workgroup
    const int wrows = squeeze ? min(rh + 8, oh - r) : rh;
    const int span = ((c & 3) + rw + 3) >> 2;
    const int wp = span * 4;
    const int wwords = wrows * span;
    const uint32_t *wsrc = reinterpret_cast<const uint32_t *>(frame) + r * ow4 + (c >> 2);
    const int wlimit = (fbytes >> 2) - 1 - (r * ow4 + (c >> 2));
    const FastDiv dv_span(span);
    auto src_of = [&](int i) {
        const int y = dv_span.div(i);
        return min(y * ow4 + (i - y * span), wlimit);
    };
    constexpr int kFW = 7;
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (k * kThreads < wwords) fw_[k] = wsrc[src_of(min(tid + k * kThreads, wwords - 1))];
    if (!squeeze) {
        // ---- the crop itself, exact k/255
#pragma unroll
        for (int k = 0; k < kFW; ++k)
            if (tid + k * kThreads < wwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = fw_[k];
        for (int i = tid + kFW * kThreads; i < wwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = wsrc[src_of(i)];
        __syncthreads();
        const unsigned char *win = raw + (c & 3);
        if (PACKED) {
            for (int i = tid; i < cnt; i += kThreads) {
                const int y = dv_rw.div(i), x = i - y * rw;
                store_packed(pout, i, unit_fast((float)win[y * wp + x]));
            }
        } else {
            for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
                const int q = tid + k_ * kThreads;
                if (q >= oh * ow4) break;
                const int row = q / ow4, x = (q - row * ow4) * 4;
                const int y = row - pr;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (y >= 0 && y < rh) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xx = x + k - pc;
                        if (xx >= 0 && xx < rw) v[k] = unit_fast((float)win[y * wp + xx]);
                    }
                }
                store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
        return;
    }

    const int rstep = kThreads / ow;
    const int xcol = tid % ow, yb = tid / ow;
    const int yf = tid >> 3, xl = tid & 7;
    const int4 yt = *reinterpret_cast<const int4 *>(t.hb + rh * oh + min(tid, oh - 1));
    const int2 mw = uniform_load_i32x2(t.wb_meta + rw), mh = uniform_load_i32x2(t.hd_meta + rh);
    const int Tw = mw.x, Th = mh.x;
    float wc[16], hw[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) wc[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) hw[q] = 0.f;
    const int yfc = min(yf, fh - 1);
    const int hlo = t.hd_lo[rh * fh + yfc];
    {
        const float4 *hs = reinterpret_cast<const float4 *>(t.hd_w + mh.y + yfc * Th);
        const float4 b0 = hs[0];
        hw[0] = b0.x, hw[1] = b0.y, hw[2] = b0.z, hw[3] = b0.w;
        if (Th > 4) {
            const float4 b1 = hs[1];
            hw[4] = b1.x, hw[5] = b1.y, hw[6] = b1.z, hw[7] = b1.w;
        }
    }
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (tid + k * kThreads < wwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = fw_[k];
    for (int i = tid + kFW * kThreads; i < wwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = wsrc[src_of(i)];
    if (tid < oh) *reinterpret_cast<int4 *>(ytab_s + tid) = yt;
    for (int i = tid + kThreads; i < rh; i += kThreads) ytab_s[i] = t.hb[rh * oh + i];
    // the composed W taps are needed only after the first pass: requested now that the window's registers are free
    const int wlo = t.wb_lo[rw * ow + xcol];
    {
        const float4 *ws = reinterpret_cast<const float4 *>(t.wb_w + mw.y + xcol * Tw);
        const float4 a0 = ws[0];
        wc[0] = a0.x, wc[1] = a0.y, wc[2] = a0.z, wc[3] = a0.w;
        if (Tw > 4) {
            const float4 a1 = ws[1];
            wc[4] = a1.x, wc[5] = a1.y, wc[6] = a1.z, wc[7] = a1.w;
        }
        if (Tw > 8) {
            const float4 a2 = ws[2], a3 = ws[3];
            wc[8] = a2.x, wc[9] = a2.y, wc[10] = a2.z, wc[11] = a2.w;
            wc[12] = a3.x, wc[13] = a3.y, wc[14] = a3.z, wc[15] = a3.w;
        }
    }
    __syncthreads();

    // ---- D = Hdwn . crop   (columns up to max(rw, Tw): every D element the W pass reads is finite)
    if (yf < fh) {
        const int kmax = (max(rw, Tw) + 7) >> 3;
        const unsigned char *src = raw + (c & 3) + hlo * wp + xl;
        float *dst = R1 + yf * t.dp + xl;
        if (Th <= 4) flex3_hdwn<4>(src, dst, hw, wp, kmax);
        else flex3_hdwn<8>(src, dst, hw, wp, kmax);
    }
    __syncthreads();
    // ---- E = D . (Wbck Wdwn)^T into R0, pitch ow (the raw bytes are dead)
    if (yb < rstep) {
        const int kmax = (fh + rstep - 1) / rstep;
        const float *src = R1 + yb * t.dp + wlo;
        float *dst = R0 + yb * ow + xcol;
        if (Tw <= 4) flex3_wcomp<4>(src, dst, wc, t.dp, ow, rstep, kmax);
        else if (Tw <= 8) flex3_wcomp<8>(src, dst, wc, t.dp, ow, rstep, kmax);
        else flex3_wcomp<16>(src, dst, wc, t.dp, ow, rstep, kmax);
    }
    __syncthreads();
    // ---- out = Hbck . E
    const float *E = R0;
    if (PACKED) {
        for (int i = tid; i < cnt; i += kThreads) {
            const int y = dv_rw.div(i), x = i - y * rw;
            const Tap tp = ytab_s[y];
            store_packed(pout, i, fmaf(tp.b, E[tp.aux * ow + x], tp.a * E[tp.lo * ow + x]));
        }
    } else {
        for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
            const int q = tid + k_ * kThreads;
            if (q >= oh * ow4) break;
            const int row = q / ow4, x = (q - row * ow4) * 4;
            const int y = row - pr;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < rh) {
                const Tap tp = ytab_s[y];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int xx = x + k - pc;
                    if (xx >= 0 && xx < rw) v[k] = fmaf(tp.b, E[tp.aux * ow + xx], tp.a * E[tp.lo * ow + xx]);
                }
            }
            store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}

}  // namespace agx
