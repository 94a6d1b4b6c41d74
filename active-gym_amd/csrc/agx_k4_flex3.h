// agx_k4_flex3.h - K4, resize_to_full form: FlexibleFovealEnv._fov_step + _get_fov_state (fov_env.py:270-330) with the
// per-axis Resize chains composed on the host (agx_rows.h).
//
//   reference, per env and stacked frame:   crop[rh][rw]
//       iff rh > fov_h:  Resize(fov_size) -> Resize(fov_res)                 fov_env.py:276-287 (rows-only test)
//       Resize(obs_size)                                                      fov_env.py:295
//   here (W and H passes commute; banded operators from agx_create):
//       rh <= fov_h :  E[rh][ow]   = crop . Wfin(rw)^T                        2 taps, reads the u8 window
//                      out[oh][ow] = Hfin(rh) . E                             2 taps
//       rh >  fov_h :  D[fh][rw]   = Hdwn(rh) . crop                          <= 4 / 8 taps (antialiased squeeze)
//                      E[fh][ow]   = D . (Wfin Wbck Wdwn)(rw)^T               <= 4 / 8 / 16 taps, composed
//                      out[oh][ow] = (Hfin Hbck)(rh) . E                      3 taps, composed
//
// grid = (fs, N): workgroup (sl, n) owns PHYSICAL ring slot sl of env n (address known at launch); block = 256.
// Everything that depends on the env's resolution (rh, rw) is wave-uniform and lives in SGPRs (readfirstlane), so
// every loop has a scalar trip count and every thread a fixed role per pass:
//   Hdwn : thread = (output row yf = tid / 8, column phase tid % 8), its row's taps in registers
//   W    : thread = output column tid % ow (its taps in registers), rows tid / ow + k * (256 / ow)
//   final: thread = float4 q = tid + 256 k of the frame, row taps from a small LDS table; 16-B written-through (sc1) stores
// The tables a thread needs and the window's rows of the slot are requested as soon as the env's state has arrived.
// LDS: R0 = raw frame (+ slack rows read with zero weights) aliased by E on the squeeze path | R1 = D, or E on the
// plain path | row taps: 21.5 KB for 84x84 / 30x30 -> 7 workgroups per CU.
#pragma once
#include "agx_fov_common.h"
#include "agx_k2_fixed.h"

namespace agx {

struct Flex3Params {
    const Tap *wf;            // [ow + 1][ow]   plain final W taps r -> ow; weights carry the 1/255 (they multiply u8)
    const int2 *wc_meta;      // [ow + 1]       {T, first float of that size's table in wc_w}, T in {4, 8, 16}
    const int32_t *wc_lo;     // [ow + 1][ow]   first D column of the composed W operator
    const float *wc_w;        // per size: [ow][T]
    const int2 *hd_meta;      // [oh + 1]       {T, first float in hd_w}, T in {4, 8}
    const int32_t *hd_lo;     // [oh + 1][fh]   first window row of the H squeeze
    const float *hd_w;        // per size: [fh][T]; weights carry the 1/255
    const int4 *hy;           // [oh + 1][oh]   final H taps {i0 | i1 << 8 | i2 << 16, w0, w1, w2} over the rows of E
    int32_t r0_bytes, r1_bytes;
    int32_t dp;               // pitch of D in floats (multiple of 8, >= ow)
};

__device__ __forceinline__ int2 uniform_load_i32x2(const int2 *ptr) {
    int2 w;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(ptr) : "memory");
    return w;
}

// D[yf][x] = sum_t hw[t] * u8 window[(lo + t)][x], x = xl + 8 k
template <int T>
__device__ __forceinline__ void flex3_hdwn(const unsigned char *src, float *dst, const float (&hw)[8], int pitch, int kmax) {
    for (int k = 0; k < kmax; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < T; ++q) acc = fmaf(hw[q], (float)src[q * pitch + 8 * k], acc);
        dst[8 * k] = acc;
    }
}

// The same four columns at a time: x = 4 (xl + 8 k) ... + 3.  The T window bytes of a column quad come as ONE pair of aligned
// dwords per tap row (ds_read2_b32) + v_alignbyte_b32 (the window starts at byte c of the image row: any alignment) and four
// v_cvt_f32_ubyteN, and the four results leave as one ds_write_b128 - a quarter of the LDS instructions of the byte form.
// Columns up to 4 * ceil(cols / 4) are produced (the W pass reads only finite values either way: bytes are finite).
template <int T>
__device__ __forceinline__ void flex3_hdwn4(const unsigned char *img, int c, int row0, int yf_dp, float *R1, const float (&hw)[8],
                                            int pitch, int xl, int cols) {
    struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; };
    const int quads = (cols + 3) >> 2;
    const uint32_t sh = (uint32_t)(c & 3);
    for (int g = xl; g < quads; g += 8) {
        const unsigned char *s = img + row0 * pitch + ((c + 4 * g) & ~3);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int q = 0; q < T; ++q) {
            const U2 d = *reinterpret_cast<const U2 *>(s + q * pitch);
            const uint32_t v = __builtin_amdgcn_alignbyte(d.y, d.x, sh);
            a0 = fmaf(hw[q], (float)(v & 0xFFu), a0);
            a1 = fmaf(hw[q], (float)((v >> 8) & 0xFFu), a1);
            a2 = fmaf(hw[q], (float)((v >> 16) & 0xFFu), a2);
            a3 = fmaf(hw[q], (float)(v >> 24), a3);
        }
        *reinterpret_cast<float4 *>(R1 + yf_dp + 4 * g) = make_float4(a0, a1, a2, a3);
    }
}

// E[y][xcol] = sum_t w[t] * D[y][lo + t], y = yb + rstep * k
template <int T>
__device__ __forceinline__ void flex3_wcomp(const float *src, float *dst, const float (&w)[16], int dp, int ow, int rstep,
                                            int kmax) {
#pragma unroll 2
    for (int k = 0; k < kmax; ++k) {
        const float *s = src + k * rstep * dp;
        float v[T];
#pragma unroll
        for (int q = 0; q < T; ++q) v[q] = s[q];
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < T; ++q) acc = fmaf(w[q], v[q], acc);
        dst[k * rstep * ow] = acc;
    }
}

// Round 4's two K4 probes, kept as compile-time switches (profiles/r04_k4_ab.txt; same box, N = 1024, shipped form 20.6-20.9 us):
// -DAGX_K4_WAVES8 (amdgpu_waves_per_eu(8, 8): 76 -> 57 VGPRs, no spill) 21.5-22.1 us; -DAGX_K4_HDWN_QUADS (flex3_hdwn4: 200 -> 124
// static LDS instructions, 82 VGPRs) 21.8-22.1 us; both 21.35-21.5 us.  All slower: the LDS (21.5 KB -> 7 workgroups per CU) caps the
// occupancy either way, and the byte-wise squeeze pass is not what the launch waits for - it is store-bound like K2.
#ifdef AGX_K4_WAVES8
#define AGX_K4_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define AGX_K4_OCC
#endif
template <class G>
__global__ __launch_bounds__(kThreads) AGX_K4_OCC void k_fovea_flexible3(G g, Flex3Params t, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh();
    if (p.mask && !p.mask[n]) {
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
    int4 *ytab_s = reinterpret_cast<int4 *>(smem + t.r0_bytes + t.r1_bytes);

    // ---- the env's state first (vmcnt retires in order: waiting for these leaves the frame loads in flight) ...
    int head, type;
    int2 res_old;
    const LocIn lin = load_flex_inputs_scalar(p, n, head, res_old, type);     // through the scalar cache: a shorter first hop
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);

    // ---- state update (fov_env.py:300-324); res from agx_set_fov_state is clamped for memory safety only
    int rh = min(max(res_old.x, 1), oh), rw = min(max(res_old.y, 1), ow), r, c;
    if (p.action && type == AGX_FOV_RES) {
        rh = clip_rint(action_value(p.action_dt, lin.w[0], lin.w[1]), 1.0, (double)oh);
        rw = clip_rint(action_value(p.action_dt, lin.w[2], lin.w[3]), 1.0, (double)ow);
        r = clip_rint((double)lin.r, 0.0, (double)(oh - rh));
        c = clip_rint((double)lin.c, 0.0, (double)(ow - rw));
    } else {
        compute_loc(p, lin, oh - rh, ow - rw, r, c);
    }
    // workgroup-uniform from here on: scalar registers, scalar branches, scalar trip counts
    rh = __builtin_amdgcn_readfirstlane(rh);
    rw = __builtin_amdgcn_readfirstlane(rw);
    r = __builtin_amdgcn_readfirstlane(r);
    c = __builtin_amdgcn_readfirstlane(c);
    int j = sl - __builtin_amdgcn_readfirstlane(head);
    if (j < 0) j += p.fs;
    if (sl == 0 && tid == 0) {
        p.loc_out[2 * n] = r;
        p.loc_out[2 * n + 1] = c;
        p.res_out[2 * n] = rh;
        p.res_out[2 * n + 1] = rw;
        if (p.user_loc) {
            p.user_loc[2 * n] = r;
            p.user_loc[2 * n + 1] = c;
        }
        if (p.user_res) {
            p.user_res[2 * n] = rh;
            p.user_res[2 * n + 1] = rw;
        }
    }
    const bool squeeze = rh > fh;                                     // rows only, fov_env.py:286
    // ---- ... then the rows of this slot the window needs: [r, r + rh + 8) clipped to the frame (the 8 rows of slack are read
    // with zero weights; what lies past the frame's end stays whatever the LDS held - bytes, hence finite as floats).  On
    // average half of the 7 KB frame: the burst every resident workgroup starts with is halved.  (Round 3 built two further
    // cuts, measured them on the same box against this form and kept neither: only the dword-aligned column span of each
    // row - a fifth of the frame - 26.9-27.1 vs 26.2-26.4 us: rows are 84 bytes apart, the span touches the same cache lines and
    // pays a division per dword; and the composed W taps requested only after the window had left its registers - 64
    // VGPRs instead of 76, eight waves per SIMD instead of six - 26.7 us: their L2 round trip no longer hides under the
    // window load.)
    const int wrows = min(rh + 8, oh - r);
    const int wp = ow;                                                // row pitch of the LDS image
    const int wwords = (wrows * ow) >> 2;
    const uint32_t *wsrc = fsrc + r * (ow >> 2);
    constexpr int kFW = 7;
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (k * kThreads < wwords) fw_[k] = wsrc[min(tid + k * kThreads, wwords - 1)];

    // ---- the taps this thread will use, requested now (L2 hits; they land under the window load)
    const int rstep = kThreads / ow;                                  // rows per sweep of the W passes (3 for ow = 84)
    const int xcol = tid % ow, yb = tid / ow;
    const int yf = tid >> 3, xl = tid & 7;                            // H-squeeze role
    const int4 yt = t.hy[rh * oh + min(tid, oh - 1)];
    int4 xt = make_int4(0, 0, 0, 0);
    int wlo = 0, hlo = 0, Tw = 0, Th = 0, wc_off = 0;
    float wc[16], hw[8];
#pragma unroll
    for (int q = 0; q < 16; ++q) wc[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) hw[q] = 0.f;
    if (squeeze) {
        const int2 mw = uniform_load_i32x2(t.wc_meta + rw), mh = uniform_load_i32x2(t.hd_meta + rh);
        Tw = mw.x;
        Th = mh.x;
        wc_off = mw.y;
        const int yfc = min(yf, fh - 1);
        hlo = t.hd_lo[rh * fh + yfc];
        const float4 *hs = reinterpret_cast<const float4 *>(t.hd_w + mh.y + yfc * Th);
        const float4 b0 = hs[0];
        hw[0] = b0.x, hw[1] = b0.y, hw[2] = b0.z, hw[3] = b0.w;
        if (Th > 4) {
            const float4 b1 = hs[1];
            hw[4] = b1.x, hw[5] = b1.y, hw[6] = b1.z, hw[7] = b1.w;
        }
    } else {
        xt = *reinterpret_cast<const int4 *>(t.wf + rw * ow + xcol);
    }
    auto load_wc = [&]() {
        if (squeeze) {
            wlo = t.wc_lo[rw * ow + xcol];
            const float4 *ws = reinterpret_cast<const float4 *>(t.wc_w + wc_off + xcol * Tw);
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
    };
    load_wc();
    // ---- LDS image: the window rows (row r of the frame is row 0 of the image), the row taps
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (tid + k * kThreads < wwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = fw_[k];
    for (int i = tid + kFW * kThreads; i < wwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = wsrc[i];
    if (tid < oh) ytab_s[tid] = yt;
    for (int i = tid + kThreads; i < oh; i += kThreads) ytab_s[i] = t.hy[rh * oh + i];
    __syncthreads();

    const unsigned char *win = raw + c;                               // window origin inside the LDS image
    const float *E;
    if (squeeze) {
        // ---- D = Hdwn . crop   (columns up to max(rw, Tw) so that every D element the W pass reads is finite)
        if (yf < fh) {
#ifdef AGX_K4_HDWN_QUADS
            if (Th <= 4) flex3_hdwn4<4>(raw, c, hlo, yf * t.dp, R1, hw, wp, xl, max(rw, Tw));
            else flex3_hdwn4<8>(raw, c, hlo, yf * t.dp, R1, hw, wp, xl, max(rw, Tw));
#else
            const int kmax = (max(rw, Tw) + 7) >> 3;
            const unsigned char *src = win + hlo * wp + xl;
            float *dst = R1 + yf * t.dp + xl;
            if (Th <= 4) flex3_hdwn<4>(src, dst, hw, wp, kmax);
            else flex3_hdwn<8>(src, dst, hw, wp, kmax);
#endif
        }
        __syncthreads();
        // ---- E = D . Wcomp^T, into R0 (the raw bytes are dead)
        if (yb < rstep) {
            const int kmax = (fh + rstep - 1) / rstep;
            const float *src = R1 + yb * t.dp + wlo;
            float *dst = R0 + yb * ow + xcol;
            if (Tw <= 4) flex3_wcomp<4>(src, dst, wc, t.dp, ow, rstep, kmax);
            else if (Tw <= 8) flex3_wcomp<8>(src, dst, wc, t.dp, ow, rstep, kmax);
            else flex3_wcomp<16>(src, dst, wc, t.dp, ow, rstep, kmax);
        }
        E = R0;
    } else {
        // ---- E = crop . Wfin^T, straight from the u8 window
        if (yb < rstep) {
            const int kmax = (rh + rstep - 1) / rstep;
            const unsigned char *c0 = win + yb * wp + xt.x, *c1 = win + yb * wp + xt.y;
            const float wa = __int_as_float(xt.z), wb = __int_as_float(xt.w);
            float *dst = R1 + yb * ow + xcol;
#pragma unroll 2
            for (int k = 0; k < kmax; ++k)
                dst[k * rstep * ow] = fmaf(wb, (float)c1[k * rstep * wp], wa * (float)c0[k * rstep * wp]);
        }
        E = R1;
    }
    __syncthreads();

    // ---- out = Hfinal . E: each float4 is the 2- or 3-tap vertical blend of ds_read_b128 rows; lane-linear stores
    const int ow4 = ow >> 2;
    const float4 *E4 = reinterpret_cast<const float4 *>(E);
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);
    if (squeeze) {
#pragma unroll 7
        for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
            const int q = tid + k_ * kThreads;
            if (q >= oh * ow4) break;
            const int row = q / ow4, x4 = q - row * ow4;
            const int4 e = ytab_s[row];
            const float w0 = __int_as_float(e.y), w1 = __int_as_float(e.z), w2 = __int_as_float(e.w);
            const float4 a = E4[(e.x & 0xFF) * ow4 + x4];
            const float4 b = E4[((e.x >> 8) & 0xFF) * ow4 + x4];
            const float4 d = E4[(e.x >> 16) * ow4 + x4];
            float4 o;
            o.x = fmaf(w2, d.x, fmaf(w1, b.x, w0 * a.x));
            o.y = fmaf(w2, d.y, fmaf(w1, b.y, w0 * a.y));
            o.z = fmaf(w2, d.z, fmaf(w1, b.z, w0 * a.z));
            o.w = fmaf(w2, d.w, fmaf(w1, b.w, w0 * a.w));
            store_obs(oout, q, o);
        }
    } else {
#pragma unroll 7
        for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
            const int q = tid + k_ * kThreads;
            if (q >= oh * ow4) break;
            const int row = q / ow4, x4 = q - row * ow4;
            const int4 e = ytab_s[row];
            const float w0 = __int_as_float(e.y), w1 = __int_as_float(e.z);
            const float4 a = E4[(e.x & 0xFF) * ow4 + x4];
            const float4 b = E4[((e.x >> 8) & 0xFF) * ow4 + x4];
            float4 o;
            o.x = fmaf(w1, b.x, w0 * a.x);
            o.y = fmaf(w1, b.y, w0 * a.y);
            o.z = fmaf(w1, b.z, w0 * a.z);
            o.w = fmaf(w1, b.w, w0 * a.w);
            store_obs(oout, q, o);
        }
    }
}

}  // namespace agx
