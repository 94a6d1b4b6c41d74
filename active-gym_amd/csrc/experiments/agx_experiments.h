// experiments/agx_experiments.h - kernel forms that were built, verified bit-identical to the default kernels and
// measured EQUAL OR SLOWER (DESIGN.md section 3 has each one's numbers).  They are not part of libagx.so: this header is
// compiled only with -DAGX_EXPERIMENTS (python active-gym_amd/build.py --experiments -> lib/libagx_exp.so), the library
// tools/ and tests/test_gpu_variants.py load through AGX_LIB to re-measure them or to hold them against the default
// kernels bit for bit.  Knobs (read per context in agx_create, experiments build only): AGX_INGEST_T, AGX_INGEST_BAND_ROWS,
// AGX_INGEST_PIPE, AGX_INGEST_WAVE, AGX_INGEST_PAIR12, AGX_FOVEA_PAIR, AGX_STEP_FUSED, AGX_STEP_SPLIT, AGX_STEP_AUX_PRIO,
// AGX_STEP_ENV.
#pragma once
#include "../agx_k1_ingest.h"
#include "../agx_k2_fixed.h"

namespace agx {

// =============================================================================================
// K1 forms
// =============================================================================================
// The same two kernels under names of their own, for the env-range parts of a split step (agx_step_fixed): a kernel
// trace then tells the concurrently running part launches from the stand-alone full-batch launches.
__global__ __launch_bounds__(kThreads) void k_ingest_part(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<kThreads>(p, blockIdx.x, blockIdx.y, smem);
}
__global__ __launch_bounds__(kThreads) void k_ingest_full12_part(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<kThreads, false, 12>(p, blockIdx.x, blockIdx.y, smem);
}

// ---------------------------------------------------------------------------------------------
// K1, two bands per workgroup with the second band's loads under the first band's tail (opt-in, AGX_INGEST_PAIR12=1:
// measured a tie with k_ingest_full12): grid = (7 bands, ceil(N / 2)), block = 256.  Workgroup (x, m) ingests band x of env 2m, then band x of env
// 2m + 1.  A band's life is  [pieces in flight] -> luminance -> barrier -> resize -> ring store; at 8 workgroups per CU
// (the hardware cap) about a third of it has no load in flight.  Here the second env's 8 pieces per thread are requested
// as soon as the first env's luminance has consumed its registers, so they fly under the first env's barrier, resize
// and store: no second register set (the same 24 VGPRs are re-used), a second 7.7 KB gray buffer in LDS, and the two envs
// share the row offsets and the resize tables (same band).  Bit-identical to k_ingest_full12.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 8) void k_ingest_pair12(IngestParams p, int n_envs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = kThreads, FBR = 12, G4 = kRawW / 4, RG = T / G4, kIter = 4;
    constexpr uint32_t kRowB = kRawRowBytes, kFrameB = kRawH * kRowB;
    const int tid = threadIdx.x, band = blockIdx.x, n0 = 2 * blockIdx.y;
    const bool two = n0 + 1 < n_envs;
    const int dy0 = band * FBR;
    int4 *ytab_s = reinterpret_cast<int4 *>(smem);                        // [12]
    int2 *xtab_s = reinterpret_cast<int2 *>(smem + sizeof(int4) * FBR);   // [ow]
    unsigned char *gray0 = smem + sizeof(int4) * FBR + sizeof(int2) * p.ow;
    constexpr uint32_t kGrayB = 2u * FBR * kRawW * 2u;                    // [2 frames][12 rows][160][2]
    unsigned char *gray1 = gray0 + kGrayB;
    const int ow4 = p.ow >> 2;
    const int rg = tid / G4, g4 = tid - rg * G4;
    const bool loader = rg < RG;
    const uint32_t col = g4 * 12;
    // lane offsets of the 8 pieces and their LDS destinations: the same for both envs; recomputed where needed (a handful
    // of VALU operations) rather than kept alive across the whole kernel
    auto offsets = [&](int it, uint32_t &a, uint32_t &b) {
        const int f = it / 2;
        const int dyl = min(rg + RG * (it - 2 * f), FBR - 1);
        const int y0 = (int)(mul_u24((uint32_t)(dy0 + dyl), (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
        const int y1 = min(y0 + 1, kRawH - 1);
        const uint32_t fo = f * kFrameB + col;
        a = mad_u24((uint32_t)y0, kRowB, fo);
        b = mad_u24((uint32_t)y1, kRowB, fo);
    };
    auto dpos = [&](int it) {
        const int f = it / 2;
        const int dyl = min(rg + RG * (it - 2 * f), FBR - 1);
        return ((f * FBR + dyl) * kRawW + g4 * 4) * 2;
    };
    const uint8_t *fb0 = p.frames + (size_t)n0 * 2 * kFrameB;
    const uint8_t *fb1 = fb0 + (two ? 2 * (size_t)kFrameB : 0);
    // the resize tables of this band (L2 hits, requested first so that they arrive first and leave their registers for
    // LDS before the luminance starts), then the first env's 8 pieces
    const int4 yt_own = p.ytab[dy0 + min(tid, FBR - 1)];
    const int2 xt_own = p.xtab[min(tid, p.ow - 1)];
    U3 w0[kIter], w1[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        uint32_t a, b;
        offsets(it, a, b);
        w0[it] = load_piece(fb0 + a);
        w1[it] = load_piece(fb0 + b);
    }
    if (tid < FBR) ytab_s[tid] = yt_own;
    if (tid < p.ow) xtab_s[tid] = xt_own;
    for (int i = tid + T; i < p.ow; i += T) xtab_s[i] = p.xtab[i];
    const int nB = __builtin_amdgcn_readfirstlane(n0 + (two ? 1 : 0));
    const uint32_t cmdA = uniform_load_u8(p.cmd + n0), cmdB = uniform_load_u8(p.cmd + nB);
    const int headA = uniform_load_i32(p.head_in + n0), headB = uniform_load_i32(p.head_in + nB);

    auto flags = [&](uint32_t cmd, int head, int n, bool &skip, bool &clear, int &nvalid, int &slot) {
        skip = (cmd & AGX_CMD_SKIP) != 0;
        clear = (cmd & AGX_CMD_CLEAR) != 0;
        if (band == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
        nvalid = min((int)(cmd & AGX_CMD_NVALID_MASK), 2);
        slot = clear ? p.fs - 1 : head;
    };
    // luminance of the 8 pieces in w0 / w1 -> gray bytes [frame][row][x][2] (the vertical tap pair of a column is one u16)
    auto luminance = [&](const uint8_t *fbase, unsigned char *gray, int nvalid) {
        uint32_t tie_its = 0;
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            uint32_t rmin = 1u;
            const uint32_t top = lum4_r(w0[it].x, w0[it].y, w0[it].z, rmin);
            const uint32_t bot = lum4_r(w1[it].x, w1[it].y, w1[it].z, rmin);
            if (loader && it / 2 < nvalid) {
                uint2 v;                                                  // t0 b0 t1 b1 | t2 b2 t3 b3
                v.x = __builtin_amdgcn_perm(bot, top, 0x05010400u);
                v.y = __builtin_amdgcn_perm(bot, top, 0x07030602u);
                *reinterpret_cast<uint2 *>(gray + dpos(it)) = v;
                tie_its |= rmin == 0u ? (1u << it) : 0u;
            }
        }
        if (__builtin_expect(tie_its != 0, 0)) {
            // an exact .5 tie (about 1e-4 of random pixels): that piece again, byte by byte, with ALE's double expression
#pragma nounroll
            for (int it = 0; it < kIter; ++it) {
                if (!((tie_its >> it) & 1u)) continue;
                uint32_t oa, ob;
                offsets(it, oa, ob);
                unsigned char *g = gray + dpos(it);
#pragma nounroll
                for (int which = 0; which < 2; ++which) {                 // row y0, then row y1: bytes g[0,2,4,6] / g[1,3,5,7]
                    const U3 a = *reinterpret_cast<const U3 *>(fbase + (which ? ob : oa));
                    const uint32_t px[4] = {a.x, __builtin_amdgcn_alignbyte(a.y, a.x, 3), __builtin_amdgcn_alignbyte(a.z, a.y, 2),
                                            a.z >> 8};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        g[2 * k + which] = (unsigned char)ale_lum_exact(px[k] & 0xFF, (px[k] >> 8) & 0xFF, (px[k] >> 16) & 0xFF);
                }
            }
        }
    };
    // OpenCV fixed-point bilinear of this band from its gray bytes + max over the sampled frames -> ring slot.  The four
    // pixels of a thread are made two at a time (8 taps in flight, not 16): the other env's 24 piece registers are live here
    auto resize_store = [&](const unsigned char *gray, int n, int nvalid, int slot, bool clear) {
        if (tid >= FBR * ow4) return;
        const int dyl = FastDiv(ow4).div(tid), xq = tid - dyl * ow4;
        const int dy = dy0 + dyl;
        uint32_t b0s = 0, b1s = 0;
        if (nvalid) {
            const int4 yt = ytab_s[dyl];
            b0s = (uint32_t)yt.z << 8;
            b1s = (uint32_t)yt.w << 8;
        }
        const unsigned char *row0 = gray + mul_u24((uint32_t)dyl, kRawW * 2);
        constexpr uint32_t fstride = (uint32_t)FBR * kRawW * 2;
        const uint32_t keep0 = nvalid > 0 ? 0xFFu : 0u, keep1 = nvalid > 1 ? 0xFFu : 0u;
        uint32_t packed = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int4 xt = make_int4(0, 0, 0, 0);
            if (nvalid) xt = *reinterpret_cast<const int4 *>(xtab_s + xq * 4 + 2 * h);
            const int xi[2] = {xt.x, xt.z};
            const int xa[2] = {xt.y, xt.w};
            uint32_t pp[2][2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const uint16_t *row = reinterpret_cast<const uint16_t *>(row0 + f * fstride);
                    pp[k][f][0] = row[xi[k] & 0xFFFF];
                    pp[k][f][1] = row[(uint32_t)xi[k] >> 16];
                }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                const u16x2 aa = __builtin_bit_cast(u16x2, (uint32_t)xa[k] << 4);
                uint32_t v[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const uint32_t p0 = pp[k][f][0], p1 = pp[k][f][1];
                    const uint32_t top = __builtin_amdgcn_perm(p1, p0, 0x0C040C00u);
                    const uint32_t bot = __builtin_amdgcn_perm(p1, p0, 0x0C050C01u);
                    const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, top), aa, 0u, false);
                    const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, bot), aa, 0u, false);
                    v[f] = (mul_hi_u24(b0s, h0 & 0xFFFFFF00u) + mul_hi_u24(b1s, h1 & 0xFFFFFF00u) + 2) >> 2;
                }
                packed |= max(v[0] & keep0, v[1] & keep1) << (8 * (2 * h + k));
            }
            if (h == 0) __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t fsz = (uint32_t)p.oh * p.ow;
        uint8_t *env = p.ring + (size_t)n * p.fs * fsz;
        const uint32_t off = mad_u24((uint32_t)dy, (uint32_t)p.ow, (uint32_t)xq * 4);
        *reinterpret_cast<uint32_t *>(env + (slot * fsz + off)) = packed;
        if (clear)
            for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + (s * fsz + off)) = 0u;
    };

    bool skipA, clearA, skipB = true, clearB = false;
    int nvA, slotA, nvB = 0, slotB = 0;
    flags(cmdA, headA, n0, skipA, clearA, nvA, slotA);
    if (two) flags(cmdB, headB, n0 + 1, skipB, clearB, nvB, slotB);
    if (!skipA && nvA > 0) luminance(fb0, gray0, nvA);
    // the second env's pieces, into the registers the first env's luminance has just released (the scheduling barriers
    // keep the compiler from hoisting them above that luminance - which would need a second register set - or sinking
    // them below the resize they are meant to fly under)
    __builtin_amdgcn_sched_barrier(0);
    if (two) {
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            uint32_t a, b;
            offsets(it, a, b);
            w0[it] = load_piece(fb1 + a);
            w1[it] = load_piece(fb1 + b);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (!skipA) resize_store(gray0, n0, nvA, slotA, clearA);
    if (two && !skipB) {
        if (nvB > 0) luminance(fb1, gray1, nvB);
        __syncthreads();
        resize_store(gray1, n0 + 1, nvB, slotB, clearB);
    }
}

// ---------------------------------------------------------------------------------------------
// K1, wave-private form (opt-in, AGX_INGEST_WAVE=1): grid = (bands, N), block = 256, but
// the 4 waves of a workgroup never meet.  Wave w owns RPW = band_rows/4 output rows end to end:
// it loads their source rows for both frames (60 of its 64 lanes x 4 pieces = 240 twelve-byte
// pieces = 3 rows x 2 frames x 40), turns them into gray bytes in ITS slice of LDS, and produces its
// own 3 x ow/4 (= 63) output dwords.  No __syncthreads: LDS traffic of one wave is ordered by the
// hardware, so only a wavefront-scope fence separates the phases.  (s_memtime stamps of the
// barrier version: 16 % of a wave's life waiting at the barrier, on top of inter-wave skew.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_ingest_wave(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = kThreads;                                           // (AGX_STAMP uses T)
    (void)T;
    const int n = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    AGX_STAMP(0);
    const int BR = p.band_rows, RPW = BR >> 2;                            // rows per wave (3 for 84x84)
    const int dy0 = band * BR + wave * RPW;                               // first output row of this wave
    const int rows = max(0, min(RPW, p.oh - dy0));
    const int ow4 = p.ow >> 2;
    // per-wave LDS slice: xtab[ow] int2 | gray[2][RPW][160][2]
    const int slice = (int)sizeof(int2) * p.ow + 2 * RPW * kRawW * 2;
    unsigned char *mine = smem + wave * ((slice + 15) & ~15);
    int2 *xtab_s = reinterpret_cast<int2 *>(mine);
    unsigned char *gray = mine + sizeof(int2) * p.ow;

    constexpr int G4 = kRawW / 4, LPI = 60, kIter = 4;                    // 60 lanes x 4 = 240 pieces
    const uint8_t *fbase = p.frames + (size_t)n * 2 * kRawFrameBytes;
    int nvalid = 2;                                                       // speculative until cmd arrives
    auto piece = [&](int it, uint32_t &o0, uint32_t &o1, int &d) {
        const int ntask = max(nvalid, 1) * max(rows, 1) * G4;
        const int t_raw = it * LPI + lane;
        const int task = min(t_raw, ntask - 1);
        const int rj = task / G4, g4 = task - rj * G4;                    // rj = f * rows + dl
        const int f = rj >= rows ? 1 : 0;
        const int dl = rj - f * rows;
        const int dy = min(dy0 + dl, p.oh - 1);
        const int y0 = (int)(mul_u24((uint32_t)dy, (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
        const int y1 = min(y0 + 1, kRawH - 1);
        const uint32_t fo = f * kRawFrameBytes + g4 * 12;
        o0 = mad_u24((uint32_t)y0, kRawRowBytes, fo);
        o1 = mad_u24((uint32_t)y1, kRawRowBytes, fo);
        d = (lane < LPI && t_raw < nvalid * rows * G4) ? ((f * RPW + dl) * kRawW + g4 * 4) * 2 : -1;
    };
    U3 w0[kIter], w1[kIter];
    int dst[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        uint32_t o0, o1;
        piece(it, o0, o1, dst[it]);
        w0[it] = *reinterpret_cast<const U3 *>(fbase + o0);
        w1[it] = *reinterpret_cast<const U3 *>(fbase + o1);
    }
    // phase-2 taps: requested after the frame pieces, parked in this wave's LDS slice
    const int dl2 = lane / ow4, xq = lane - dl2 * ow4;
    const bool p2 = lane < rows * ow4;
    const int4 yt2 = p.ytab[min(dy0 + dl2, p.oh - 1)];
    int2 xt_own[2];
    xt_own[0] = p.xtab[min(lane, p.ow - 1)];
    xt_own[1] = p.xtab[min(lane + 64, p.ow - 1)];
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const int head = uniform_load_i32(p.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (band == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip || rows == 0) return;
    nvalid = min((int)(cmd & AGX_CMD_NVALID_MASK), 2);
    const int slot = clear ? p.fs - 1 : head;
    AGX_STAMP(1);
    if (nvalid > 0) {
        uint32_t tie_its = 0;
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            bool tie = false;
            const uint32_t top = lum4(w0[it].x, w0[it].y, w0[it].z, tie);
            const uint32_t bot = lum4(w1[it].x, w1[it].y, w1[it].z, tie);
            if (it * LPI + lane >= nvalid * rows * G4) dst[it] = -1;      // frame-1 pieces are void when nvalid == 1
            if (dst[it] >= 0) {
                uint2 v;                                                  // t0 b0 t1 b1 | t2 b2 t3 b3
                v.x = __builtin_amdgcn_perm(bot, top, 0x05010400u);
                v.y = __builtin_amdgcn_perm(bot, top, 0x07030602u);
                *reinterpret_cast<uint2 *>(gray + dst[it]) = v;
                tie_its |= tie ? (1u << it) : 0u;
            }
        }
        if (__builtin_expect(tie_its != 0, 0)) {                          // exact .5 luminance ties, ~1e-4 of pixels
#pragma nounroll
            for (int it = 0; it < kIter; ++it) {
                if (!((tie_its >> it) & 1u)) continue;
                uint32_t o0, o1;
                int d;
                piece(it, o0, o1, d);
                const U3 a = *reinterpret_cast<const U3 *>(fbase + o0);
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
        if (lane < p.ow) xtab_s[lane] = xt_own[0];
        if (lane + 64 < p.ow) xtab_s[lane + 64] = xt_own[1];
        for (int i = lane + 128; i < p.ow; i += 64) xtab_s[i] = p.xtab[i];
    }
    AGX_STAMP(2);
    // this wave's LDS writes are consumed by other lanes of the SAME wave only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    AGX_STAMP(3);
    if (p2) {
        const int dy = dy0 + dl2;
        uint32_t packed = 0;
        if (nvalid > 0) {
            const uint32_t b0 = (uint32_t)yt2.z, b1 = (uint32_t)yt2.w;
            const int4 xt01 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4);
            const int4 xt23 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4 + 2);
            const int xi[4] = {xt01.x, xt01.z, xt23.x, xt23.z};
            const int xa[4] = {xt01.y, xt01.w, xt23.y, xt23.w};
            const unsigned char *row0 = gray + mul_u24((uint32_t)dl2, kRawW * 2);
            const uint32_t fstride = (uint32_t)RPW * kRawW * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x0 = xi[k] & 0xFFFF, x1 = (uint32_t)xi[k] >> 16;
                const uint32_t a0 = xa[k] & 0xFFFF, a1 = (uint32_t)xa[k] >> 16;
                uint32_t best = 0;
                for (int f = 0; f < nvalid; ++f) {
                    const uint16_t *row = reinterpret_cast<const uint16_t *>(row0 + f * fstride);
                    const uint32_t p0 = row[x0], p1 = row[x1];           // lo byte: row y0, hi byte: row y1
                    const uint32_t h0 = mad_u24(p1 & 0xFF, a1, mul_u24(p0 & 0xFF, a0));
                    const uint32_t h1 = mad_u24(p1 >> 8, a1, mul_u24(p0 >> 8, a0));
                    const uint32_t v = (((mul_u24(b0, h0 >> 4) >> 16) + (mul_u24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xFF;
                    best = max(best, v);
                }
                packed |= best << (8 * k);
            }
        }
        const uint32_t fsz = (uint32_t)p.oh * p.ow;
        uint8_t *env = p.ring + (size_t)n * p.fs * fsz;
        const uint32_t off = mad_u24((uint32_t)dy, (uint32_t)p.ow, (uint32_t)xq * 4);
        *reinterpret_cast<uint32_t *>(env + (slot * fsz + off)) = packed;
        if (clear)
            for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + (s * fsz + off)) = 0u;
    }
    AGX_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// K1, pipelined form: grid = (P, N), block = 256.  Workgroup (part, n) walks bands part, part+P, ...
// of env n.  The NEXT band's source pieces are requested (registers B) before the current band's
// luminance (registers A) is computed, so every wave has loads in flight for its whole life instead
// of once per workgroup; gray bytes are double-buffered in LDS, one barrier per band.  All loads
// are unconditional: the prefetch past the last band re-reads the last band (L2 hits, never used).
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_ingest_pipe(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y;
    const int part = blockIdx.x, P = gridDim.x;
    const int tid = threadIdx.x;
    const uint32_t cmd = uniform_load_u8(p.cmd + n);
    const int head = uniform_load_i32(p.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0;
    const bool clear = (cmd & AGX_CMD_CLEAR) != 0;
    if (part == 0 && tid == 0) p.head_out[n] = skip ? head : (clear ? 0 : (head + 1 == p.fs ? 0 : head + 1));
    if (skip) return;
    int nvalid = cmd & AGX_CMD_NVALID_MASK;
    if (nvalid > 2) nvalid = 2;
    const int slot = clear ? p.fs - 1 : head;

    constexpr int G4 = kRawW / 4, RG = T / G4, kIter = 4;
    const int BR = p.band_rows;
    const int gray_bytes = 2 * BR * kRawW * 2;
    int4 *ytab_s = reinterpret_cast<int4 *>(smem);                        // [oh]  {y0, y1, b0, b1}
    int2 *xtab_s = reinterpret_cast<int2 *>(smem + sizeof(int4) * p.oh);    // [ow]
    unsigned char *gray0 = smem + sizeof(int4) * p.oh + sizeof(int2) * p.ow;
    unsigned char *gray1 = gray0 + gray_bytes;
    for (int i = tid; i < p.oh; i += T) ytab_s[i] = p.ytab[i];
    for (int i = tid; i < p.ow; i += T) xtab_s[i] = p.xtab[i];
    const int ow4 = p.ow >> 2;
    const int rg = tid / G4, g4 = tid - rg * G4;
    const bool loader = rg < RG;
    const uint8_t *fbase = p.frames + (size_t)n * 2 * kRawFrameBytes;
    const uint32_t col = g4 * 12;
    const size_t fsz = (size_t)p.oh * p.ow;
    uint8_t *env = p.ring + (size_t)n * p.fs * fsz;
    const int last_band = p.nbands - 1;
    __syncthreads();

    auto offsets = [&](int band, int it, uint32_t &o0, uint32_t &o1, int &d) {
        const int dy0 = band * BR;
        const int rows = min(BR, p.oh - dy0);
        const int nrj = max(nvalid, 1) * rows;
        const int rj_raw = rg + RG * it;
        const int rj = min(rj_raw, nrj - 1);
        const int f = rj >= rows ? 1 : 0;
        const int dyl = rj - f * rows;
        const int4 yt = ytab_s[dy0 + dyl];
        const uint32_t fo = f * kRawFrameBytes + col;
        o0 = mad_u24((uint32_t)yt.x, kRawRowBytes, fo);
        o1 = mad_u24((uint32_t)yt.y, kRawRowBytes, fo);
        d = (rj_raw < nvalid * rows && loader) ? ((f * BR + dyl) * kRawW + g4 * 4) * 2 : -1;
    };
    auto issue = [&](U3 (&w0)[kIter], U3 (&w1)[kIter], int band) {
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            uint32_t o0, o1;
            int d;
            offsets(band, it, o0, o1, d);
            w0[it] = *reinterpret_cast<const U3 *>(fbase + o0);
            w1[it] = *reinterpret_cast<const U3 *>(fbase + o1);
        }
    };
    auto lum_to_lds = [&](const U3 (&w0)[kIter], const U3 (&w1)[kIter], int band, unsigned char *gray) {
        uint32_t tie_its = 0;
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            bool tie = false;
            const uint32_t top = lum4(w0[it].x, w0[it].y, w0[it].z, tie);
            const uint32_t bot = lum4(w1[it].x, w1[it].y, w1[it].z, tie);
            uint32_t o0, o1;
            int d;
            offsets(band, it, o0, o1, d);
            if (d >= 0) {
                uint2 v;
                v.x = __builtin_amdgcn_perm(bot, top, 0x05010400u);
                v.y = __builtin_amdgcn_perm(bot, top, 0x07030602u);
                *reinterpret_cast<uint2 *>(gray + d) = v;
                tie_its |= tie ? (1u << it) : 0u;
            }
        }
        if (__builtin_expect(tie_its != 0, 0)) {
#pragma nounroll
            for (int it = 0; it < kIter; ++it) {
                if (!((tie_its >> it) & 1u)) continue;
                uint32_t o0, o1;
                int d;
                offsets(band, it, o0, o1, d);
                const U3 a = *reinterpret_cast<const U3 *>(fbase + o0);
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
    };
    auto finish = [&](int band, const unsigned char *gray) {
        const int dy0 = band * BR;
        const int rows = min(BR, p.oh - dy0);
        if (tid < rows * ow4) {
            const int dyl = tid / ow4, xq = tid - dyl * ow4;
            uint32_t packed = 0;
            if (nvalid) {
                const int4 yt = ytab_s[dy0 + dyl];
                const int b0 = yt.z, b1 = yt.w;
                const int4 xt01 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4);
                const int4 xt23 = *reinterpret_cast<const int4 *>(xtab_s + xq * 4 + 2);
                const int xi[4] = {xt01.x, xt01.z, xt23.x, xt23.z};
                const int xa[4] = {xt01.y, xt01.w, xt23.y, xt23.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x0 = xi[k] & 0xFFFF, x1 = xi[k] >> 16;
                    const int a0 = xa[k] & 0xFFFF, a1 = xa[k] >> 16;
                    int best = 0;
                    for (int f = 0; f < nvalid; ++f) {
                        const uint16_t *row = reinterpret_cast<const uint16_t *>(gray + (size_t)(f * BR + dyl) * kRawW * 2);
                        const uint32_t p0 = row[x0], p1 = row[x1];
                        const uint32_t h0 = mad_u24(p1 & 0xFF, a1, mul_u24(p0 & 0xFF, a0));
                        const uint32_t h1 = mad_u24(p1 >> 8, a1, mul_u24(p0 >> 8, a0));
                        const int v = (int)((((mul_u24(b0, h0 >> 4) >> 16) + (mul_u24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xFF);
                        best = max(best, v);
                    }
                    packed |= (uint32_t)best << (8 * k);
                }
            }
            const size_t off = (size_t)(dy0 + dyl) * p.ow + xq * 4;
            *reinterpret_cast<uint32_t *>(env + slot * fsz + off) = packed;
            if (clear)
                for (int s = 0; s < p.fs - 1; ++s) *reinterpret_cast<uint32_t *>(env + s * fsz + off) = 0u;
        }
    };

    U3 a0[kIter], a1[kIter], b0[kIter], b1[kIter];
    int band = part;
    if (band > last_band) return;
    issue(a0, a1, band);
    while (true) {
        issue(b0, b1, min(band + P, last_band));
        lum_to_lds(a0, a1, band, gray0);
        __syncthreads();
        finish(band, gray0);
        band += P;
        if (band > last_band) break;
        issue(a0, a1, min(band + P, last_band));
        lum_to_lds(b0, b1, band, gray1);
        __syncthreads();
        finish(band, gray1);
        band += P;
        if (band > last_band) break;
    }
}


// =============================================================================================
// K2 forms and the fused step launches
// =============================================================================================
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
        const ObsOut oout = obs_out(out4, oh * ow4);
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
            store_obs(oout, q, o);
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


// the same with round 3's band12 ingest body (round 4 re-test: AGX_STEP_FUSED=2)
template <class G>
__global__ __launch_bounds__(kThreads) void k_step_fixed12(G g, IngestParams pi, FovParams pf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int x = blockIdx.x, n = blockIdx.y;
    if (x < pi.nbands)
        ingest_band12<false>(pi, x, n, smem, (int)threadIdx.x);
    else
        fovea_fixed_body<G, AGX_OUT_RESIZE>(g, pf, x - pi.nbands, n, smem);
}
// ... and with the fovea workgroups FIRST in the grid (x < fs), so that the store-bound work is resident from the start of the launch
template <class G>
__global__ __launch_bounds__(kThreads) void k_step_fixed12_ff(G g, IngestParams pi, FovParams pf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int x = blockIdx.x, n = blockIdx.y;
    if (x >= pf.fs)
        ingest_band12<false>(pi, x - pf.fs, n, smem, (int)threadIdx.x);
    else
        fovea_fixed_body<G, AGX_OUT_RESIZE>(g, pf, x, n, smem);
}

}  // namespace agx

#include "agx_step_env.h"
