// agx_k1_direct.h - K1g, direct form (round 4): the grayscale-screen ingest of the headline geometry without LDS.
//
// The band12 form (agx_k1_ingest.h) stages the source rows of a 12-row band in LDS behind a barrier.  With RGB screens the
// 23 KB a workgroup requests keep HBM busy while other workgroups compute; with ALE's own grayscale screens (a third of the
// bytes) the same structure is latency-sized: 8 resident workgroups x 7.7 KB is all a CU ever has in flight, and every
// workgroup spends half its life behind its barrier (K1g 17.8 us for 62 MB).  A gray source needs no staging at all: a pixel is a
// byte, the two horizontal taps of an output pixel are adjacent bytes, and the taps of two adjacent output pixels lie within
// 4 bytes of each other (any down-scaling geometry: checked at agx_create) - so one thread, owning 4 adjacent output pixels of
// one row as in phase 2 of the band form, fetches for each of the 4 source rows it needs (top / bottom x two frames) two 8-byte
// windows at dword-aligned addresses, picks every tap pair out of its window with ONE v_perm_b32 (selector from the tap's
// offset in the window), and runs the same fixed-point arithmetic.  No LDS, no barrier, waves independent of one another;
// the window addresses come from arithmetic (x0 = (dx * x_mul + x_add) >> x_shift, rows as in the band form), so the pixel
// loads are issued before any table has arrived.
#pragma once
#include "agx_k1_ingest.h"

namespace agx {

struct U2 { uint32_t x, y; };
__device__ __forceinline__ U2 load_window8(const uint8_t *ptr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
#ifdef AGX_K1D_PLAIN
    const u32x2 v = *reinterpret_cast<const u32x2_a4 *>(ptr);
#else
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_a4 *>(ptr));
#endif
    U2 r;
    r.x = v.x, r.y = v.y;
    return r;
}

// grid = (oh / 12, N), block = 256 (12 * ow / 4 = 252 threads own an output quad each)
template <bool COMPACT>
__device__ __forceinline__ void ingest_direct_gray(const IngestParams &p, const int band, const int n, const int tid) {
    const uint32_t kFrameB = (COMPACT ? (uint32_t)p.src_rows : (uint32_t)kRawH) * kRawW;
    const int ow4 = p.ow >> 2;
    const int t = min(tid, kB12Rows * ow4 - 1);                           // the 4 spare threads repeat the last quad's loads
    const int dyl = (int)(mul_u24((uint32_t)t, (uint32_t)p.ow4_inv16) >> 16);
    const int xq = t - dyl * ow4;
    const uint32_t dy = (uint32_t)(band * kB12Rows + dyl);
    const int y0 = COMPACT ? (int)(2u * dy) : (int)(mul_u24(dy, (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
    const int y1 = COMPACT ? y0 + 1 : min(y0 + 1, kRawH - 1);
    // window bases of the output pairs (4 xq, 4 xq + 1) and (4 xq + 2, 4 xq + 3): the dword that holds x0 of the pair's first
    // pixel, pulled back where 8 bytes from there would leave the row
    uint32_t xb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t x0 = (mul_u24((uint32_t)(4 * xq + 2 * j), (uint32_t)p.x_mul) + (uint32_t)p.x_add) >> p.x_shift;
        xb[j] = min(x0 & ~3u, (uint32_t)kRawW - 8u);
    }
    const uint8_t *f0 = p.frames + (size_t)n * 2 * kFrameB;
    const uint32_t r0 = mul_u24((uint32_t)y0, kRawW), r1 = mul_u24((uint32_t)y1, kRawW);
    U2 w[2][2][2];                                                        // [frame][top | bottom][pair]
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w[f][0][j] = load_window8(f0 + f * kFrameB + r0 + xb[j]);
            w[f][1][j] = load_window8(f0 + f * kFrameB + r1 + xb[j]);
        }
    const int2 yt = p.ytab12[dy];
    const int4 xt01 = *reinterpret_cast<const int4 *>(p.xtab12 + xq * 4);
    const int4 xt23 = *reinterpret_cast<const int4 *>(p.xtab12 + xq * 4 + 2);
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const int head = uniform_load_i32(p.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (band == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip || tid >= kB12Rows * ow4) return;
    const int nvalid = min((int)(cmd & AGX_CMD_NVALID_MASK), 2);
    const int slot = clear ? p.fs - 1 : head;
    uint32_t packed = 0;
    if (nvalid > 0) {
        const uint32_t b0s = (uint32_t)yt.x, b1s = (uint32_t)yt.y;
        const uint32_t xo[4] = {(uint32_t)xt01.x, (uint32_t)xt01.z, (uint32_t)xt23.x, (uint32_t)xt23.z};   // 2 * x0
        const uint32_t xa[4] = {(uint32_t)xt01.y, (uint32_t)xt01.w, (uint32_t)xt23.y, (uint32_t)xt23.w};
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        uint32_t sel[4];                                                  // bytes (o, o + 1) of the window -> u16 lanes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t o = (xo[k] >> 1) - xb[k >> 1];
            sel[k] = 0x0C010C00u + o * 0x00010001u;
        }
        auto vsum = [&](uint32_t tp, uint32_t bt, int k) {
            const u16x2 aa = __builtin_bit_cast(u16x2, xa[k]);
            const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, tp), aa, 0u, false);
            const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, bt), aa, 0u, false);
            return mul_hi_u24(b0s, h0 & 0xFFFFFF00u) + mul_hi_u24(b1s, h1 & 0xFFFFFF00u) + 2u;
        };
        auto px = [&](int f, int k) {
            const U2 &a = w[f][0][k >> 1], &b = w[f][1][k >> 1];
            return vsum(__builtin_amdgcn_perm(a.y, a.x, sel[k]), __builtin_amdgcn_perm(b.y, b.x, sel[k]), k);
        };
        if (nvalid > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) packed |= (max(px(0, k), px(1, k)) >> 2) << (8 * k);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) packed |= (px(0, k) >> 2) << (8 * k);
        }
    }
    const uint32_t fsz = (uint32_t)p.oh * p.ow;
    uint8_t *env = p.ring + (size_t)n * p.fs * fsz;
    const uint32_t off = mad_u24(dy, (uint32_t)p.ow, (uint32_t)xq * 4);
    *reinterpret_cast<uint32_t *>(env + (slot * fsz + off)) = packed;
    if (clear)
        for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + (s * fsz + off)) = 0u;
}

__global__ __launch_bounds__(kThreads) void k_ingest_grayraw_direct(IngestParams p) {
    ingest_direct_gray<false>(p, blockIdx.x, blockIdx.y, (int)threadIdx.x);
}
__global__ __launch_bounds__(kThreads) void k_ingest_grayraw_direct_compact(IngestParams p) {
    ingest_direct_gray<true>(p, blockIdx.x, blockIdx.y, (int)threadIdx.x);
}

}  // namespace agx
