// agx_k1_ingest.h - K1: raw frames -> gray -> obs-sized -> ring slot (k_ingest and its opt-in forms, k_ingest_gray,
// k_ingest_rgb).
#pragma once
#include "agx_common.h"

namespace agx {

// ---------------------------------------------------------------------------------------------
// K1: ingest
// ---------------------------------------------------------------------------------------------

// ALE ColourPalette luminance: (uint8) round(r*0.2989 + g*0.5870 + b*0.1140) in C double.
// The rational value (2989r+5870g+1140b)/10000 decides everything except exact .5 ties, where
// the double evaluation sometimes lands below the tie (292 of 2^24 inputs); those are replayed
// in double with the same operation order and no fused multiply-add.
//
// Instruction diet (K1 is issue-bound, not HBM-bound, unless this is tight):
//   t   = 2989r + 5870g + 1140b + 5000 via two v_dot4_u32_u8 on the pixel dword
//         (weights split as 256*(11,22,4) + (173,238,116)) and one v_lshl_add_u32;
//   q   = floor(t / 10000) = v_mul_hi_u32_u24(t, ceil(2^37/1e4)) >> 5, exact for t < 2^22
//         (t * eps / 2^37 < 1.9e-5 < the 1e-4 granularity of t/10000);
//   tie = (q * 10000 == t), one v_mul_u32_u24 + v_cmp.
constexpr uint32_t kLumWLo = 173u | (238u << 8) | (116u << 16);
constexpr uint32_t kLumWHi = 11u | (22u << 8) | (4u << 16);

__device__ __forceinline__ uint32_t ale_lum_px(uint32_t px /* r | g<<8 | b<<16 | any<<24 */, bool &tie) {
    const uint32_t hi = __builtin_amdgcn_udot4(px, kLumWHi, 0u, false);
    const uint32_t t = __builtin_amdgcn_udot4(px, kLumWLo, (hi << 8) + 5000u, false);
    const uint32_t q = (uint32_t)(((uint64_t)(t & 0xFFFFFFu) * 13743896ull) >> 32) >> 5;
    tie |= mul_u24(q, 10000u) == t;
    return q;
}

// 12 bytes = 4 RGB pixels -> 4 luminance bytes packed little-endian; `tie` is raised when any of
// them sits on an exact .5 tie (the caller re-does that piece with ale_lum_exact)
__device__ __forceinline__ uint32_t lum4(uint32_t a, uint32_t b, uint32_t c, bool &tie) {
    const uint32_t q0 = ale_lum_px(a, tie);
    const uint32_t q1 = ale_lum_px(__builtin_amdgcn_alignbyte(b, a, 3), tie);
    const uint32_t q2 = ale_lum_px(__builtin_amdgcn_alignbyte(c, b, 2), tie);
    const uint32_t q3 = ale_lum_px(c >> 8, tie);
    return q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
}

// The same with a cheaper tie test for the default kernel: the remainder r = t - 10000 q comes from ONE signed 24-bit
// multiply-add (v_mad_i32_i24), the remainders of a piece are folded with v_min3_u32, and a single compare per row job
// asks whether any of them is 0 - instead of a multiply, a compare and a scalar OR per pixel.
template <int SH = 0>      // SH = 1: the pixel sits in bytes 1..3 of `px` (weights moved up a byte, no data shift)
__device__ __forceinline__ uint32_t ale_lum_px_r(uint32_t px, uint32_t &rmin) {
    const uint32_t hi = __builtin_amdgcn_udot4(px, kLumWHi << (8 * SH), 0u, false);
    const uint32_t t = __builtin_amdgcn_udot4(px, kLumWLo << (8 * SH), (hi << 8) + 5000u, false);
    const uint32_t q = (uint32_t)(((uint64_t)(t & 0xFFFFFFu) * 13743896ull) >> 32) >> 5;
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(-10000), "v"(t));     // 0 <= r < 10000
    rmin = min(rmin, r);
    return q;
}
__device__ __forceinline__ uint32_t lum4_r(uint32_t a, uint32_t b, uint32_t c, uint32_t &rmin) {
    const uint32_t q0 = ale_lum_px_r(a, rmin);
    const uint32_t q1 = ale_lum_px_r(__builtin_amdgcn_alignbyte(b, a, 3), rmin);
    const uint32_t q2 = ale_lum_px_r(__builtin_amdgcn_alignbyte(c, b, 2), rmin);
    const uint32_t q3 = ale_lum_px_r<1>(c, rmin);
    return q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
}

// exact-tie replay of one pixel in C double, ALE's operation order, no fused multiply-add
__device__ __forceinline__ uint32_t ale_lum_exact(uint32_t r, uint32_t g, uint32_t b) {
    const uint32_t t = 2989u * r + 5870u * g + 1140u * b + 5000u;
    uint32_t q = t / 10000u;
    if (t - q * 10000u == 0u) {
#pragma clang fp contract(off)
        const double x = ((double)r * 0.2989 + (double)g * 0.5870) + (double)b * 0.1140;
        const double fl = floor(x);
        q = (uint32_t)fl + (((x - fl) >= 0.5) ? 1u : 0u);
    }
    return q;
}

struct __attribute__((aligned(4))) U3 { uint32_t x, y, z; };
// one 12-byte piece of a raw screen (4 RGB pixels), nontemporal: the screens are read exactly once, and keeping them out
// of L2 / Infinity Cache leaves those to the ring (re-read four times) - same box, N=1024: K1 38.2 -> 37.3 us, K2
// 24.4 -> 24.0 us, step 62.6 -> 61.4 us.  (-DAGX_K1_PLAIN_LOADS restores default-policy loads.)
__device__ __forceinline__ U3 load_piece(const uint8_t *ptr) {
#ifndef AGX_K1_PLAIN_LOADS
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
    const u32x3 v = __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4 *>(ptr));
    U3 r;
    r.x = v.x, r.y = v.y, r.z = v.z;
    return r;
#else
    return *reinterpret_cast<const U3 *>(ptr);
#endif
}

struct IngestParams {
    const uint8_t *frames;   // [N][2][210][160][3]
    const uint8_t *cmd;      // [N]
    uint8_t *ring;           // [N][fs][oh][ow]
    const int32_t *head_in;  // [N]
    int32_t *head_out;       // [N]
    const int2 *xtab;        // [ow]  {x0 | x1<<16, a0 | a1<<16}
    const int4 *ytab;        // [oh]  {y0, y1, b0, b1}
    int32_t oh, ow, fs;
    int32_t band_rows;       // output rows per workgroup (band_rows * ow/4 <= 256, band_rows <= 12)
    int32_t nbands;          // ceil(oh / band_rows)
    // y0(dy) == (dy * y_mul + y_add) >> y_shift and y1 == min(y0 + 1, raw_h - 1) for every dy (checked
    // exhaustively against ytab at agx_create); lets the frame loads start without a table round trip.
    int32_t y_affine, y_mul, y_add, y_shift;
    // band12 form (ingest_band12): phase-2 tables prepared on the host and the thread -> (row, column quad) divider
    const int2 *xtab12;      // [ow]  {2 * x0 (byte offset of the tap pair in a u16 gray row), (a0 | a1 << 16) << 4}
    const int2 *ytab12;      // [oh]  {b0 << 8, b1 << 8}
    int32_t ow4_inv16;       // ceil(65536 / (ow / 4)): tid / (ow / 4) == (tid * ow4_inv16) >> 16 for tid < 256
    // COMPACT source screens (agx_ingest_compact): only the src_rows source rows the vertical resize reads, packed in
    // ascending order - u8 [N][2][src_rows][160][3] (gray: [N][2][src_rows][160]).  `ytab` then holds PACKED row indices;
    // the band12 form needs every (y0, y1) pair disjoint and ascending, so that output row dy reads packed rows 2 dy, 2 dy + 1.
    int32_t src_rows;
    unsigned long long *stamps;   // diagnostic builds only (AGX_STAMPS): [workgroup][wave][8] records
};


// grid = (bands, N), block = T threads (T = 128 or 256).  Per workgroup: the two source rows of each
// of its output rows, for both frames, go HBM -> registers (12-byte / 4-pixel pieces, lane-contiguous)
// -> luminance -> LDS; then each thread produces 4 adjacent output pixels and stores one dword.
// LDS gray layout: [frame][dyl][x][2] — the vertical pair (row y0, row y1) of one source column is
// one aligned u16, so the bilinear taps of an output pixel are two ds_read_u16.
// Measured floor of this access shape with no arithmetic at all: ~30 us at N=1024 (tools/membench.hip).
// GRAY: the source frames are already ALE's own grayscale screens, u8 [N][2][210][160] (`getScreenGrayscale`, what the
// reference itself reads, atari_env.py:74): one dword = 4 pixels per lane, no luminance arithmetic, a third of the bytes.
// FBR > 0: compile-time band height with every band full (oh % FBR == 0, 2 * FBR == (T / 40) * 4) and the affine source
// row form: the row job of iteration `it` is (frame it / 2, row rg + RG * (it % 2)) with no clamping against a ragged
// last band, so the second frame's offsets are the first's plus a constant and the index arithmetic folds away.
// COMPACT: the screens hold only the p.src_rows rows the resize reads (packed); p.ytab holds packed row indices and
// p.y_affine is 0 (rows always come from the table).
template <int T, bool GRAY = false, int FBR = 0, bool COMPACT = false>
__device__ __forceinline__ void ingest_band(const IngestParams &p, const int band, const int n, unsigned char *smem,
                                            const int tid) {
    AGX_STAMP(0);
    static_assert(!(COMPACT && FBR > 0), "the compile-time band form reads full screens");
    constexpr uint32_t kRowB = GRAY ? kRawW : kRawRowBytes;               // source row / frame pitch in bytes
    const uint32_t kFrameB = (COMPACT ? (uint32_t)p.src_rows : (uint32_t)kRawH) * kRowB;
    const int BR = FBR > 0 ? FBR : p.band_rows;
    const int dy0 = band * BR;
    const int rows = FBR > 0 ? FBR : min(BR, p.oh - dy0);
    int4 *ytab_s = reinterpret_cast<int4 *>(smem);                      // [BR]
    int2 *xtab_s = reinterpret_cast<int2 *>(smem + sizeof(int4) * BR);    // [ow]
    unsigned char *gray = smem + sizeof(int4) * BR + sizeof(int2) * p.ow; // [2][BR][160][2]
    const int ow4 = p.ow >> 2;
    if (FBR == 0 && (COMPACT || !p.y_affine)) {           // general geometry / compact screens: source rows come from the table
        if (tid < rows) ytab_s[tid] = p.ytab[dy0 + tid];
        __syncthreads();
    }

    // phase 1: thread = (piece g4 of 40, row group rg of T/40); row job rj = rg + RG*it is (frame,
    // output row); it loads both source rows of that output row, 4 pixels wide -> 8 gray bytes in LDS.
    // The loads go out FIRST and unconditionally, as if both frames were wanted (stamps showed 40 % of
    // a wave's life spent waiting for the per-env command byte before its first frame load): the
    // command / ring-head scalar loads then complete underneath them; `skip` and `nvalid` only gate
    // what is written.  (A skipped env costs its reads; sparse launches are rare and host-bound.)
    constexpr int G4 = kRawW / 4;                                         // 40 four-pixel pieces per row
    constexpr int RG = T / G4;                                            // row groups: 6 (T=256) / 3 (T=128)
    constexpr int kIter = 4;                                              // 2 frames * band_rows / RG
    const int rg = tid / G4, g4 = tid - rg * G4;
    const uint8_t *fbase = p.frames + (size_t)n * 2 * kFrameB;            // wave-uniform base
    const uint32_t col = g4 * (GRAY ? 4 : 12);
    int nvalid = 2;                                                       // speculative until cmd arrives
    auto row_offsets = [&](int it, uint32_t &o0, uint32_t &o1, int &d) {
        if (FBR > 0) {
            constexpr int half = kIter / 2;
            const int f = it / half;                                      // compile-time in the unrolled loops
            const int dyl = min(rg + RG * (it - f * half), FBR - 1);      // (idle threads rg >= RG stay in bounds)
            const int y0 = (int)(mul_u24((uint32_t)(dy0 + dyl), (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
            const int y1 = min(y0 + 1, kRawH - 1);
            const uint32_t fo = f * kFrameB + col;
            o0 = mad_u24((uint32_t)y0, kRowB, fo);
            o1 = mad_u24((uint32_t)y1, kRowB, fo);
            d = (f < nvalid && rg < RG) ? ((f * FBR + dyl) * kRawW + g4 * 4) * 2 : -1;
            return;
        }
        const int nrj = max(nvalid, 1) * rows;
        const int rj_raw = rg + RG * it;
        const int rj = min(rj_raw, nrj - 1);
        const int f = rj >= rows ? 1 : 0;                                 // nvalid <= 2
        const int dyl = rj - f * rows;
        int y0, y1;
        if (!COMPACT && p.y_affine) {
            y0 = (int)(mul_u24((uint32_t)(dy0 + dyl), (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
            y1 = min(y0 + 1, kRawH - 1);
        } else {
            const int4 yt = ytab_s[dyl];
            y0 = yt.x;
            y1 = yt.y;
        }
        const uint32_t fo = f * kFrameB + col;                            // 32-bit lane offsets
        o0 = mad_u24((uint32_t)y0, kRowB, fo);
        o1 = mad_u24((uint32_t)y1, kRowB, fo);
        d = (rj_raw < nvalid * rows && rg < RG) ? ((f * BR + dyl) * kRawW + g4 * 4) * 2 : -1;
    };
    U3 w0[kIter], w1[kIter];                                              // GRAY uses .x only
    int dst[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        uint32_t o0, o1;
        row_offsets(it, o0, o1, dst[it]);
        if (GRAY) {
            w0[it].x = *reinterpret_cast<const uint32_t *>(fbase + o0);
            w1[it].x = *reinterpret_cast<const uint32_t *>(fbase + o1);
        } else {
            w0[it] = load_piece(fbase + o0);
            w1[it] = load_piece(fbase + o1);
        }
    }
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const int head = uniform_load_i32(p.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (band == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip) return;
    nvalid = min((int)(cmd & AGX_CMD_NVALID_MASK), 2);
    const int slot = clear ? p.fs - 1 : head;
    const int nrj = nvalid * rows;
    if (nrj > 0) {
#pragma unroll
        for (int it = 0; it < kIter; ++it)                                // frame-1 jobs are void when nvalid == 1
            if (FBR > 0 ? (it / (kIter / 2) >= nvalid) : (rg + RG * it >= nrj)) dst[it] = -1;
        AGX_STAMP(1);
        // the phase-2 tables are requested AFTER the frame pieces (vmcnt retires in order, so waiting
        // for them later costs nothing) and parked in LDS once the luminance work is done
        const int4 yt_own = p.ytab[dy0 + min(tid, rows - 1)];
        const int2 xt_own = p.xtab[min(tid, p.ow - 1)];
        uint32_t tie_its = 0;
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            uint32_t rmin = 1u;                                           // smallest remainder of the 8 pixels (GRAY: none)
            const uint32_t top = GRAY ? w0[it].x : lum4_r(w0[it].x, w0[it].y, w0[it].z, rmin);
            const uint32_t bot = GRAY ? w1[it].x : lum4_r(w1[it].x, w1[it].y, w1[it].z, rmin);
            const bool tie = rmin == 0u;
            if (dst[it] >= 0) {
                uint2 v;                                                  // t0 b0 t1 b1 | t2 b2 t3 b3
                v.x = __builtin_amdgcn_perm(bot, top, 0x05010400u);
                v.y = __builtin_amdgcn_perm(bot, top, 0x07030602u);
                *reinterpret_cast<uint2 *>(gray + dst[it]) = v;
                tie_its |= tie ? (1u << it) : 0u;
            }
        }
        if (!GRAY && __builtin_expect(tie_its != 0, 0)) {
            // about 1e-4 of random pixels sit on an exact .5 tie: redo those pieces byte by byte with
            // the exact rule.  The source bytes are re-read (L2 hits) so that the fast path does not
            // have to keep 24 registers alive for this branch.
#pragma nounroll
            for (int it = 0; it < kIter; ++it) {
                if (!((tie_its >> it) & 1u)) continue;
                uint32_t o0, o1;
                int d;
                row_offsets(it, o0, o1, d);
                const U3 a = *reinterpret_cast<const U3 *>(fbase + o0);   // one round trip, then registers only
                const U3 b = *reinterpret_cast<const U3 *>(fbase + o1);
#pragma nounroll
                for (int j = 0; j < 8; ++j) {
                    const bool which = j & 1;
                    const int k = j >> 1;
                    const uint32_t x = which ? b.x : a.x, y = which ? b.y : a.y, z = which ? b.z : a.z;
                    const uint64_t lo = (uint64_t)x | ((uint64_t)y << 32);
                    const uint64_t hi = (uint64_t)y | ((uint64_t)z << 32);
                    const uint32_t px = (uint32_t)(k < 2 ? (lo >> (24 * k)) : (hi >> (24 * k - 32)));
                    gray[d + j] = (unsigned char)ale_lum_exact(px & 0xFF, (px >> 8) & 0xFF, (px >> 16) & 0xFF);
                }
            }
        }
        if (tid < rows) ytab_s[tid] = yt_own;
        if (tid < p.ow) xtab_s[tid] = xt_own;
        for (int i = tid + T; i < p.ow; i += T) xtab_s[i] = p.xtab[i];
    }
    AGX_STAMP(2);
    __syncthreads();
    AGX_STAMP(3);

    // phase 2: OpenCV fixed-point bilinear + max over the sampled frames
    if (tid < rows * ow4) {
        const int dyl = FastDiv(ow4).div(tid), xq = tid - dyl * ow4;
        const int dy = dy0 + dyl;
        uint32_t b0 = 0, b1 = 0;
        int4 xt01 = make_int4(0, 0, 0, 0), xt23 = xt01;
        if (nvalid) {
            const int4 yt = ytab_s[dyl];
            b0 = (uint32_t)yt.z;
            b1 = (uint32_t)yt.w;
            xt01 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4);
            xt23 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4 + 2);
        }
        const int xi[4] = {xt01.x, xt01.z, xt23.x, xt23.z};
        const int xa[4] = {xt01.y, xt01.w, xt23.y, xt23.w};
        const unsigned char *row0 = gray + mul_u24((uint32_t)dyl, kRawW * 2);      // frame 0, this output row
        const uint32_t fstride = (uint32_t)BR * kRawW * 2;                         // wave-uniform
        // both frames' taps are read up front (16 ds_read_u16 in flight, no per-frame loop); a frame that was not
        // sampled this step is masked out of the max afterwards - its LDS bytes are stale but never used
        uint32_t pp[4][2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const uint16_t *row = reinterpret_cast<const uint16_t *>(row0 + f * fstride);
                pp[k][f][0] = row[xi[k] & 0xFFFF];                                 // lo byte: row y0, hi byte: row y1
                pp[k][f][1] = row[(uint32_t)xi[k] >> 16];
            }
        const uint32_t keep0 = nvalid > 0 ? 0xFFu : 0u, keep1 = nvalid > 1 ? 0xFFu : 0u;
        const uint32_t b0s = b0 << 8, b1s = b1 << 8;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // horizontal pass as two v_dot2_u32_u16: the table already stores the coefficient pair as a0 | a1 << 16, and
            // one v_perm_b32 puts the two taps of a source row side by side as 16-bit lanes
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            // coefficients * 16 (a <= 2048 keeps each half inside its 16 bits): the dot2 then yields h << 4, so that
            // (h >> 4) << 8 is one AND and (b * (h >> 4)) >> 16 one v_mul_hi_u32_u24 of (b << 8) and that value
            const u16x2 aa = __builtin_bit_cast(u16x2, (uint32_t)xa[k] << 4);
            uint32_t v[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const uint32_t p0 = pp[k][f][0], p1 = pp[k][f][1];
                const uint32_t top = __builtin_amdgcn_perm(p1, p0, 0x0C040C00u);   // p0.byte0 | p1.byte0 << 16  (row y0)
                const uint32_t bot = __builtin_amdgcn_perm(p1, p0, 0x0C050C01u);   // p0.byte1 | p1.byte1 << 16  (row y1)
                const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, top), aa, 0u, false);
                const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, bot), aa, 0u, false);
                v[f] = (mul_hi_u24(b0s, h0 & 0xFFFFFF00u) + mul_hi_u24(b1s, h1 & 0xFFFFFF00u) + 2) >> 2;
            }
            packed |= max(v[0] & keep0, v[1] & keep1) << (8 * k);
        }
        const uint32_t fsz = (uint32_t)p.oh * p.ow;
        uint8_t *env = p.ring + (size_t)n * p.fs * fsz;                            // wave-uniform
        const uint32_t off = mad_u24((uint32_t)dy, (uint32_t)p.ow, (uint32_t)xq * 4);
        *reinterpret_cast<uint32_t *>(env + (slot * fsz + off)) = packed;
        if (clear)
            for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + (s * fsz + off)) = 0u;
    }
    AGX_STAMP(4);
}
template <int T, bool GRAY = false, int FBR = 0, bool COMPACT = false>
__device__ __forceinline__ void ingest_band(const IngestParams &p, const int band, const int n, unsigned char *smem) {
    ingest_band<T, GRAY, FBR, COMPACT>(p, band, n, smem, (int)threadIdx.x);
}


// ---------------------------------------------------------------------------------------------
// K1, band12 form (round 3) - the kernel of the headline geometry.  Same work split as ingest_band<256, GRAY, 12>
// (12-row bands, all full; affine source rows; 240 loader threads x 8 pieces) with a shorter instruction stream:
//
// * luminance: T8 = 8 t = 8 (2989 r + 5870 g + 1140 b + 5000) straight from the two v_dot4_u32_u8 (weights
//   8 * (2989, 5870, 1140) = 256 * (93, 183, 35) + (104, 112, 160), constant 40000 = 256 * 156 + 64), then ONE
//   v_mul_hi_u32 by ceil(2^45 / 10^4): X = floor(t * 65536 / 10^4 + e), 0 <= e < 2^-7, so byte 2 of X is
//   q = floor(t / 10^4) exactly (the largest fraction, 9999 / 10^4 * 65536 + e, stays below 65536), byte 3 is 0 and the
//   low 16 bits are 0 exactly when t is a multiple of 10^4 - the exact .5 ties, the only inputs where ALE's double
//   expression can differ (a remainder m > 0 leaves floor(6.55 m) >= 6 there).  All 2^24 triples checked on the host
//   (tests/test_oracle_resize.py::test_band12_luminance_arithmetic) and on the device.  No shift, no remainder
//   multiply: q is already byte-aligned and the tie test is a packed 16-bit minimum over the X of a row job.
// * gray goes to LDS as u16 [frame][row][top | bottom][160]: one v_perm_b32 per two pixels builds the 16-bit lanes that
//   v_dot2_u32_u16 wants, so phase 2 reads every horizontal tap pair (x0, x0 + 1 - adjacent for every down-scaling
//   geometry, checked at agx_create) with ONE ds_read_b32 at a 2-byte-aligned address and needs no byte shuffling.
// * everything phase 1 writes is written unconditionally (a frame that was not sampled is left out in phase 2, by a
//   wave-uniform branch on nvalid): no per-job destination predicate.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kLum8Lo = 104u | (112u << 8) | (160u << 16);
constexpr uint32_t kLum8Hi = 93u | (183u << 8) | (35u << 16);
constexpr uint32_t kLumM13 = 3518437209u;                                 // ceil(2^45 / 10^4)

template <int SH = 0>      // SH = 1: the pixel sits in bytes 1..3 of `px`
__device__ __forceinline__ uint32_t lum_x(uint32_t px) {
    const uint32_t hi = __builtin_amdgcn_udot4(px, kLum8Hi << (8 * SH), 156u, false);
    const uint32_t t8 = __builtin_amdgcn_udot4(px, kLum8Lo << (8 * SH), (hi << 8) + 64u, false);
    return __umulhi(t8, kLumM13);
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// ALE's double expression for a pixel on an exact tie (operation order of ColourPalette, no fused multiply-add)
__device__ __forceinline__ uint32_t ale_lum_tie(uint32_t r, uint32_t g, uint32_t b) {
#pragma clang fp contract(off)
    const double x = ((double)r * 0.2989 + (double)g * 0.5870) + (double)b * 0.1140;
    const double fl = floor(x);
    return (uint32_t)fl + (((x - fl) >= 0.5) ? 1u : 0u);
}
struct __attribute__((packed, aligned(4))) U2a4 { uint32_t x, y; };      // two dwords at a 4-byte-aligned LDS address

constexpr int kB12Rows = 12;
constexpr uint32_t kB12RowB = kRawW * 2;                                  // one u16 gray row
constexpr uint32_t kB12JobB = 2 * kB12RowB;                               // top + bottom source row of one output row
constexpr uint32_t kB12FrameB = kB12Rows * kB12JobB;
constexpr uint32_t kB12GrayB = 2 * kB12FrameB;                            // 15,360 B

// LDS: ytab12[12] int2 | xtab12[ow] int2 | gray16 [2][12][2][160] u16
// COMPACT: packed source rows (see IngestParams::src_rows): output row dy reads packed rows 2 dy and 2 dy + 1 - no row
// arithmetic at all, and every byte of every 128-B line the kernel touches is used.
template <bool GRAY, bool COMPACT = false>
__device__ __forceinline__ void ingest_band12(const IngestParams &p, const int band, const int n, unsigned char *smem,
                                              const int tid) {
    constexpr int T = kThreads;
    (void)T;
    AGX_STAMP(0);
    constexpr uint32_t kRowB = GRAY ? kRawW : kRawRowBytes;
    const uint32_t kFrameB = (COMPACT ? (uint32_t)p.src_rows : (uint32_t)kRawH) * kRowB;
    constexpr int G4 = kRawW / 4, RG = kThreads / G4;                     // 40 pieces per row, 6 row groups
    const int dy0 = band * kB12Rows;
    int2 *ytab_s = reinterpret_cast<int2 *>(smem);
    int2 *xtab_s = ytab_s + kB12Rows;
    unsigned char *gray = smem + sizeof(int2) * (kB12Rows + p.ow);
    const uint8_t *f0 = p.frames + (size_t)n * 2 * kFrameB;               // wave-uniform bases of the two sampled screens
    const uint8_t *f1 = f0 + kFrameB;
    const int rg = tid / G4, g4 = tid - rg * G4;
    const bool loader = rg < RG;                                          // 240 of the 256 threads
    // phase 1 loads: row job `it` = (frame it / 2, output row rg + 6 * (it % 2)); both source rows, 4 pixels wide.
    // Issued before the command byte is known, as if both frames were sampled.
    U3 w0[4], w1[4];                                                      // GRAY uses .x only
    uint32_t dstb = 0;
    if (loader) {
        const uint32_t col = g4 * (GRAY ? 4 : 12);
        uint32_t o0[2], o1[2];
#pragma unroll
        for (int itl = 0; itl < 2; ++itl) {
            const uint32_t dy = (uint32_t)(dy0 + rg + RG * itl);
            const int y0 = COMPACT ? (int)(2u * dy) : (int)(mul_u24(dy, (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
            const int y1 = COMPACT ? y0 + 1 : min(y0 + 1, kRawH - 1);
            o0[itl] = mad_u24((uint32_t)y0, kRowB, col);
            o1[itl] = mad_u24((uint32_t)y1, kRowB, col);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint8_t *fb = it < 2 ? f0 : f1;
            if (GRAY) {
                w0[it].x = *reinterpret_cast<const uint32_t *>(fb + o0[it & 1]);
                w1[it].x = *reinterpret_cast<const uint32_t *>(fb + o1[it & 1]);
            } else {
                w0[it] = load_piece(fb + o0[it & 1]);
                w1[it] = load_piece(fb + o1[it & 1]);
            }
        }
        dstb = mad_u24((uint32_t)rg, kB12JobB, (uint32_t)g4 * 8u);
    }
    // phase-2 tables of this band: requested after the pieces (vmcnt retires in order), parked in LDS after the luminance
    const int2 yt_own = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(p.ytab12 + dy0) + 8u * (uint32_t)min(tid, kB12Rows - 1));
    const int2 xt_own = *reinterpret_cast<const int2 *>(reinterpret_cast<const char *>(p.xtab12) + 8u * (uint32_t)min(tid, p.ow - 1));
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const int head = uniform_load_i32(p.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (band == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip) return;
    const int nvalid = min((int)(cmd & AGX_CMD_NVALID_MASK), 2);
    const int slot = clear ? p.fs - 1 : head;
    AGX_STAMP(1);
    if (nvalid > 0) {
        if (loader) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                // LDS byte offset of this job's top row: frame it / 2, output row rg + 6 * (it % 2), column 4 g4
                const uint32_t d = dstb + (uint32_t)(it >> 1) * kB12FrameB + (uint32_t)(it & 1) * (RG * kB12JobB);
                uint2 top, bot;
                if (GRAY) {
                    top.x = __builtin_amdgcn_perm(0u, w0[it].x, 0x0C010C00u);
                    top.y = __builtin_amdgcn_perm(0u, w0[it].x, 0x0C030C02u);
                    bot.x = __builtin_amdgcn_perm(0u, w1[it].x, 0x0C010C00u);
                    bot.y = __builtin_amdgcn_perm(0u, w1[it].x, 0x0C030C02u);
                } else {
                    // the 8 pixels of the job, stage by stage (8 independent chains: a v_dot4 result needs three wait
                    // states before the next VALU instruction may read it, and the chains fill them for one another)
                    const uint32_t px[8] = {w0[it].x, __builtin_amdgcn_alignbyte(w0[it].y, w0[it].x, 3),
                                            __builtin_amdgcn_alignbyte(w0[it].z, w0[it].y, 2), w0[it].z,
                                            w1[it].x, __builtin_amdgcn_alignbyte(w1[it].y, w1[it].x, 3),
                                            __builtin_amdgcn_alignbyte(w1[it].z, w1[it].y, 2), w1[it].z};
                    uint32_t X[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        X[j] = __builtin_amdgcn_udot4(px[j], (j & 3) == 3 ? kLum8Hi << 8 : kLum8Hi, 156u, false);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        X[j] = __builtin_amdgcn_udot4(px[j], (j & 3) == 3 ? kLum8Lo << 8 : kLum8Lo, (X[j] << 8) + 64u, false);
#pragma unroll
                    for (int j = 0; j < 8; ++j) X[j] = __umulhi(X[j], kLumM13);
                    // low halves: the smallest 16-bit fraction of the 8 pixels; 0 <=> one of them is an exact tie
                    // (1,703 of the 2^24 colours: 1e-4 of random pixels, i.e. every other workgroup meets one).  The tied
                    // pixels are re-done right here with ALE's double expression, from the piece registers of this very
                    // iteration: a wave whose 64 lanes hold no tie (95 % of the jobs) skips the block on s_cbranch_execz.
                    // (Round 2 flagged the job and re-read its 24 source bytes after the loop: a global round trip in
                    // front of the barrier of half of all workgroups.)
                    const uint32_t m = pk_min_u16(pk_min_u16(pk_min_u16(X[0], X[1]), pk_min_u16(X[2], X[3])),
                                                  pk_min_u16(pk_min_u16(X[4], X[5]), pk_min_u16(X[6], X[7])));
                    if (__builtin_expect((m & 0xFFFFu) == 0u, 0)) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if ((X[j] & 0xFFFFu) == 0u) {
                                const uint32_t v = (j & 3) == 3 ? px[j] >> 8 : px[j];
                                X[j] = ale_lum_tie(v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF) << 16;
                            }
                    }
                    top.x = __builtin_amdgcn_perm(X[1], X[0], 0x0C060C02u);   // q(px 0) | q(px 1) << 16
                    top.y = __builtin_amdgcn_perm(X[3], X[2], 0x0C060C02u);
                    bot.x = __builtin_amdgcn_perm(X[5], X[4], 0x0C060C02u);
                    bot.y = __builtin_amdgcn_perm(X[7], X[6], 0x0C060C02u);
                }
                *reinterpret_cast<uint2 *>(gray + d) = top;
                *reinterpret_cast<uint2 *>(gray + d + kB12RowB) = bot;
            }
        }
        if (tid < kB12Rows) ytab_s[tid] = yt_own;
        if (tid < p.ow) xtab_s[tid] = xt_own;
    }
    AGX_STAMP(2);
    __syncthreads();
    AGX_STAMP(3);

    // phase 2: OpenCV fixed-point bilinear of 4 adjacent output pixels + max over the sampled frames -> one ring dword
    const int ow4 = p.ow >> 2;
    if (tid < kB12Rows * ow4) {
        const int dyl = (int)(mul_u24((uint32_t)tid, (uint32_t)p.ow4_inv16) >> 16);
        const int xq = tid - dyl * ow4;
        uint32_t packed = 0;
        if (nvalid > 0) {
            const int2 yt = ytab_s[dyl];
            const uint32_t b0s = (uint32_t)yt.x, b1s = (uint32_t)yt.y;
            const int4 xt01 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4);
            const int4 xt23 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4 + 2);
            const uint32_t xo[4] = {(uint32_t)xt01.x, (uint32_t)xt01.z, (uint32_t)xt23.x, (uint32_t)xt23.z};
            const uint32_t xa[4] = {(uint32_t)xt01.y, (uint32_t)xt01.w, (uint32_t)xt23.y, (uint32_t)xt23.w};
            const unsigned char *row = gray + mul_u24((uint32_t)dyl, kB12JobB);
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            // vertical pass of one frame at tap pair k: ((b0 * (h_top >> 4)) >> 16) + ((b1 * (h_bot >> 4)) >> 16) + 2,
            // the coefficient pair pre-scaled by 16 so that (h >> 4) << 8 is one AND and each term one v_mul_hi_u32_u24
            auto vsum = [&](uint32_t t, uint32_t b, int k) {
                const u16x2 aa = __builtin_bit_cast(u16x2, xa[k]);
                const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, t), aa, 0u, false);
                const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, b), aa, 0u, false);
                return mul_hi_u24(b0s, h0 & 0xFFFFFF00u) + mul_hi_u24(b1s, h1 & 0xFFFFFF00u) + 2u;
            };
            // all tap pairs are requested before the first is used.  The pair {x0, x0 + 1} as u16 lanes is the dword at byte
            // 2 * x0 of the row - 2-byte aligned only, and a misaligned ds_read_b32 is slow on gfx950 (measured: the kernel
            // took 59 us with them) - so the two ALIGNED dwords around it are read (ds_read2_b32) and v_alignbit_b32 picks
            // the pair: shift 0 for an even x0, 16 for an odd one.
            uint32_t tp[2][4], bt[2][4];
            auto pair = [&](const unsigned char *rowp, int k) {
                const U2a4 d = *reinterpret_cast<const U2a4 *>(rowp + (xo[k] & ~3u));
                return __builtin_amdgcn_alignbit(d.y, d.x, (xo[k] & 2u) << 3);
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tp[0][k] = pair(row, k);
                bt[0][k] = pair(row + kB12RowB, k);
            }
            if (nvalid > 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    tp[1][k] = pair(row + kB12FrameB, k);
                    bt[1][k] = pair(row + kB12FrameB + kB12RowB, k);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    packed |= (max(vsum(tp[0][k], bt[0][k], k), vsum(tp[1][k], bt[1][k], k)) >> 2) << (8 * k);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) packed |= (vsum(tp[0][k], bt[0][k], k) >> 2) << (8 * k);
            }
        }
        const uint32_t fsz = (uint32_t)p.oh * p.ow;
        uint8_t *env = p.ring + (size_t)n * p.fs * fsz;                            // wave-uniform
        const uint32_t off = mad_u24((uint32_t)(dy0 + dyl), (uint32_t)p.ow, (uint32_t)xq * 4);
        *reinterpret_cast<uint32_t *>(env + (slot * fsz + off)) = packed;
        if (clear)
            for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + (s * fsz + off)) = 0u;
    }
    AGX_STAMP(4);
}

template <int T>
__global__ __launch_bounds__(T) void k_ingest(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<T>(p, blockIdx.x, blockIdx.y, smem);
}

// the headline geometry's form: 12-row bands, all full (84 = 7 * 12), affine source rows
__global__ __launch_bounds__(kThreads) void k_ingest_full12(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band12<false>(p, blockIdx.x, blockIdx.y, smem, (int)threadIdx.x);
}

// compact source screens (agx_ingest_compact / agx_ingest_gray_raw_compact)
__global__ __launch_bounds__(kThreads) void k_ingest_full12_compact(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band12<false, true>(p, blockIdx.x, blockIdx.y, smem, (int)threadIdx.x);
}
__global__ __launch_bounds__(kThreads) void k_ingest_compact(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<kThreads, false, 0, true>(p, blockIdx.x, blockIdx.y, smem);
}
__global__ __launch_bounds__(kThreads) void k_ingest_grayraw_full12_compact(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band12<true, true>(p, blockIdx.x, blockIdx.y, smem, (int)threadIdx.x);
}
__global__ __launch_bounds__(kThreads) void k_ingest_grayraw_compact(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<kThreads, true, 0, true>(p, blockIdx.x, blockIdx.y, smem);
}

// K1g: the same from ALE grayscale screens u8 [N][2][210][160] (agx_ingest_gray_raw)
__global__ __launch_bounds__(kThreads) void k_ingest_grayraw(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<kThreads, true>(p, blockIdx.x, blockIdx.y, smem);
}
__global__ __launch_bounds__(kThreads) void k_ingest_grayraw_full12(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band12<true>(p, blockIdx.x, blockIdx.y, smem, (int)threadIdx.x);
}

struct IngestGrayParams {
    const uint8_t *small;    // [N][2][oh][ow]
    const uint8_t *cmd;
    uint8_t *ring;
    const int32_t *head_in;
    int32_t *head_out;
    int32_t oh, ow, fs;
};

// grid = (ceil(oh*ow/4 / 256), N)
__global__ __launch_bounds__(kThreads) void k_ingest_gray(IngestGrayParams p) {
    const int n = blockIdx.y;
    const uint32_t cmd = p.cmd[n];
    const int head = p.head_in[n];
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip) return;
    int nvalid = cmd & AGX_CMD_NVALID_MASK;
    if (nvalid > 2) nvalid = 2;
    const int slot = clear ? p.fs - 1 : head;
    const int words = (p.oh * p.ow) >> 2;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= words) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.small) + (size_t)n * 2 * words;
    uint32_t v = 0;
    if (nvalid >= 1) v = src[i];
    if (nvalid >= 2) {
        const uint32_t u = src[words + i];
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= max((v >> (8 * k)) & 0xFF, (u >> (8 * k)) & 0xFF) << (8 * k);
        v = m;
    }
    uint32_t *env = reinterpret_cast<uint32_t *>(p.ring) + (size_t)n * p.fs * words;
    env[(size_t)slot * words + i] = v;
    if (clear)
        for (int s = 0; s < p.fs - 1; ++s) env[(size_t)s * words + i] = 0u;
}

// K1b (DMC pixel front end, reference dmc_env.py:175-186): frames are obs-sized RGB renders
// u8[N][oh][ow][3]; gray = cv2.cvtColor(obs, COLOR_BGR2GRAY) - OpenCV's fixed-point weights with channel 0
// taken as blue, exactly what the reference does to an RGB render - appended to the ring, no max-pool, no
// resize.  One thread = 4 output pixels = 12 source bytes (three dwords, lane-contiguous).
struct IngestRgbParams {
    const uint8_t *frames;   // [N][oh][ow][3]
    const uint8_t *cmd;      // [N]
    uint8_t *ring;
    const int32_t *head_in;
    int32_t *head_out;
    int32_t oh, ow, fs;
    uint32_t k0, k1, k2;     // weights of channels 0,1,2; k0 + k1 + k2 == 1 << shift
    uint32_t shift;
};

// grid = (ceil(oh*ow/4 / 256), N)
__global__ __launch_bounds__(kThreads) void k_ingest_rgb(IngestRgbParams p) {
    const int n = blockIdx.y;
    const int words = (p.oh * p.ow) >> 2;
    const int i = min((int)(blockIdx.x * kThreads + threadIdx.x), words - 1);
    // the pixel loads go out before the per-env command / head loads they do not depend on
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.frames) + ((size_t)n * words + i) * 3;
    const uint32_t a = src[0], b = src[1], c = src[2];
    const uint32_t cmd = p.cmd[n];
    const int head = p.head_in[n];
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip || (int)(blockIdx.x * kThreads + threadIdx.x) >= words) return;
    const uint32_t rnd = 1u << (p.shift - 1);
    auto g = [&](uint32_t c0, uint32_t c1, uint32_t c2) {
        return (mad_u24(c0, p.k0, mad_u24(c1, p.k1, mad_u24(c2, p.k2, rnd))) >> p.shift) & 0xFFu;
    };
    // bytes: a = c0 c1 c2 c0' | b = c1' c2' c0" c1" | c = c2" c0"' c1"' c2"'
    uint32_t v = 0;
    if ((cmd & AGX_CMD_NVALID_MASK) != 0) {
        v = g(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF);
        v |= g(a >> 24, b & 0xFF, (b >> 8) & 0xFF) << 8;
        v |= g((b >> 16) & 0xFF, b >> 24, c & 0xFF) << 16;
        v |= g((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24) << 24;
    }
    const int slot = clear ? p.fs - 1 : head;
    uint32_t *env = reinterpret_cast<uint32_t *>(p.ring) + (size_t)n * p.fs * words;
    env[(size_t)slot * words + i] = v;
    if (clear)
        for (int s = 0; s < p.fs - 1; ++s) env[(size_t)s * words + i] = 0u;
}

}  // namespace agx
