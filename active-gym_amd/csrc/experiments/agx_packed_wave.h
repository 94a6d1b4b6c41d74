// experiments/agx_packed_wave.h - packed ragged crops with ONE WAVE per (slot, env) item (round 4).
// Same arithmetic, tables and LDS plan as k_fovea_flexible_raw3<G, kRawPacked> (agx_k4_raw3.h), element for element - so the results
// are bit-identical - but the item is walked by a 64-thread workgroup: every pass strides by 64 instead of 256, the H squeeze by 8
// rows at a time, the W pass by 64 columns, and the three `__syncthreads()` are barriers of a single wave.  Built only with
// -DAGX_EXPERIMENTS (lib/libagx_exp.so), selected by AGX_PACKED_WAVE=1 in agx_fovea_flexible_packed.
// Measured (profiles/r04_packed_wave_ab.txt): see docs/HISTORY.md "Round 4".
#pragma once
#include "../agx_k4_raw3.h"

namespace agx {

template <class G>
__global__ __launch_bounds__(64) void k_fovea_flexible_raw3_wave(G g, FlexRawParams t, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = 64;
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh();
    const int fbytes = oh * ow;
    unsigned char *raw = smem;
    float *R0 = reinterpret_cast<float *>(smem);
    float *R1 = reinterpret_cast<float *>(smem + t.r0_bytes);
    Tap *ytab_s = reinterpret_cast<Tap *>(smem + t.r0_bytes + t.r1_bytes);

    // ---- final state (k_flex_state_scan wrote it), ring head, packed offset: as the shipped kernel
    const int2 rc = *(reinterpret_cast<const int2 *>(p.loc_in) + n);
    const int2 hw_ = *(reinterpret_cast<const int2 *>(p.res_in) + n);
    const int hd = p.head[n];
    int64_t off = t.local_off[n];
    const int b = n / kScanEnvsPerBlock;
    if (b > 0) {
        int64_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 64 * k;
            if (i < b) v += t.block_tot[i];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        off += v;
    }
    const int r = __builtin_amdgcn_readfirstlane(rc.x), c = __builtin_amdgcn_readfirstlane(rc.y);
    const int rh = __builtin_amdgcn_readfirstlane(hw_.x), rw = __builtin_amdgcn_readfirstlane(hw_.y);
    const int head = __builtin_amdgcn_readfirstlane(hd);
    const int64_t poff = ((int64_t)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
    if (sl == 0 && tid == 0) {
        t.offsets[n] = poff;
        if (n == t.n_envs - 1) t.offsets[n + 1] = poff + (int64_t)p.fs * rh * rw;
    }
    int j = sl - head;
    if (j < 0) j += p.fs;
    const int cnt = rh * rw;
    if (poff + (int64_t)p.fs * cnt > p.packed_cap) return;
    const bool squeeze = rh > fh;
    const uint8_t *frame = p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes;
    const int ow4 = ow >> 2;
    float *pdst = p.packed + poff + (int64_t)j * cnt;
    const PackedOut pout = packed_out(pdst, cnt);
    const FastDiv dv_rw(rw);

    // ---- the window -> LDS (same image as the shipped kernel: rows [r, r + rh (+ 8)), dword-aligned column span)
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
    // eight loads in flight per lane, then the stores; up to 92 x 22 dwords / 64 lanes = 32 per lane
    for (int base = 0; base < wwords; base += 8 * T) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = wsrc[src_of(min(base + tid + k * T, wwords - 1))];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + tid + k * T < wwords) reinterpret_cast<uint32_t *>(raw)[base + tid + k * T] = v[k];
    }
    if (squeeze)
        for (int i = tid; i < rh; i += T) ytab_s[i] = t.hb[rh * oh + i];
    __syncthreads();
    const unsigned char *win = raw + (c & 3);
    if (!squeeze) {
        for (int i = tid; i < cnt; i += T) {
            const int y = dv_rw.div(i), x = i - y * rw;
            store_packed(pout, i, unit_fast((float)win[y * wp + x]));
        }
        return;
    }
    const int2 mw = uniform_load_i32x2(t.wb_meta + rw), mh = uniform_load_i32x2(t.hd_meta + rh);
    const int Tw = mw.x, Th = mh.x;
    // ---- D = Hdwn . crop: 8 rows of D at a time (yf = tid / 8 + 8 k), each lane the columns xl + 8 m of its row
    {
        const int xl = tid & 7;
        const int kmax = (max(rw, Tw) + 7) >> 3;
        for (int yf = tid >> 3; yf < fh; yf += T >> 3) {
            float hw[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) hw[q] = 0.f;
            const int hlo = t.hd_lo[rh * fh + yf];
            const float4 *hs = reinterpret_cast<const float4 *>(t.hd_w + mh.y + yf * Th);
            const float4 b0 = hs[0];
            hw[0] = b0.x, hw[1] = b0.y, hw[2] = b0.z, hw[3] = b0.w;
            if (Th > 4) {
                const float4 b1 = hs[1];
                hw[4] = b1.x, hw[5] = b1.y, hw[6] = b1.z, hw[7] = b1.w;
            }
            const unsigned char *src = win + hlo * wp + xl;
            float *dst = R1 + yf * t.dp + xl;
            if (Th <= 4) flex3_hdwn<4>(src, dst, hw, wp, kmax);
            else flex3_hdwn<8>(src, dst, hw, wp, kmax);
        }
    }
    __syncthreads();
    // ---- E = D . (Wbck Wdwn)^T into R0 (pitch ow): a lane owns column xcol = tid + 64 m (only the rw columns the last pass reads)
    for (int xcol = tid; xcol < rw; xcol += T) {
        float wc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) wc[q] = 0.f;
        const int wlo = t.wb_lo[rw * ow + xcol];
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
        const float *src = R1 + wlo;
        float *dst = R0 + xcol;
        // rows one after the other (rstep = 1): the same sums, in the same order, as flex3_wcomp
        if (Tw <= 4) flex3_wcomp<4>(src, dst, wc, t.dp, ow, 1, fh);
        else if (Tw <= 8) flex3_wcomp<8>(src, dst, wc, t.dp, ow, 1, fh);
        else flex3_wcomp<16>(src, dst, wc, t.dp, ow, 1, fh);
    }
    __syncthreads();
    // ---- out = Hbck . E
    const float *E = R0;
    for (int i = tid; i < cnt; i += T) {
        const int y = dv_rw.div(i), x = i - y * rw;
        const Tap tp = ytab_s[y];
        store_packed(pout, i, fmaf(tp.b, E[tp.aux * ow + x], tp.a * E[tp.lo * ow + x]));
    }
}

}  // namespace agx
