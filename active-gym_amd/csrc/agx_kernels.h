// agx_kernels.h — gfx950 device code of the observation pipeline (HBM-bound
// byte/float streaming work: no MFMA anywhere, there is no dense contraction).
//
//   k_ingest            K1  RGB -> ALE luminance -> OpenCV fixed-point bilinear -> 2-frame max -> u8 ring slot
//   k_ingest_gray       K1' same append from already obs-sized gray frames
//   k_ingest_rgb        K1b DMC front end: obs-sized RGB -> cv2 BGR2GRAY fixed point -> ring slot (no max, no resize)
//   k_stack_u8 / k_full K0  ring -> stack order (u8 / f32 k/255)
//   k_fovea_fixed       K2  clip/rint sensory action, crop, {raw | mask-out | bilinear upsample}
//   k_fovea_generic     K3/K4 peripheral squeeze-expand + paste, flexible (ragged) fovea
//
// Persistent per-env state (owned by the context, see agx_api.hip):
//   ring  u8 [N][fs][oh][ow]   numerators k of the reference's float32 k/255 frames
//   head  i32[2][N]            next slot to write == oldest frame; double-buffered so that the
//                              several workgroups of one env all read the pre-launch value
//   loc   i32[2][N][2], res i32[2][N][2]   fov_loc / fov_res, double-buffered for the same reason
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

// ---------------------------------------------------------------------------------------------
// K1: ingest
// ---------------------------------------------------------------------------------------------

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
    unsigned long long *stamps;   // diagnostic builds only (AGX_STAMPS): [workgroup][wave][8] records
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

// grid = (bands, N), block = T threads (T = 128 or 256).  Per workgroup: the two source rows of each
// of its output rows, for both frames, go HBM -> registers (12-byte / 4-pixel pieces, lane-contiguous)
// -> luminance -> LDS; then each thread produces 4 adjacent output pixels and stores one dword.
// LDS gray layout: [frame][dyl][x][2] — the vertical pair (row y0, row y1) of one source column is
// one aligned u16, so the bilinear taps of an output pixel are two ds_read_u16.
// Measured floor of this access shape with no arithmetic at all: ~30 us at N=1024 (tools/membench.hip).
template <int T>
__device__ __forceinline__ void ingest_band(const IngestParams &p, const int band, const int n, unsigned char *smem) {
    const int tid = threadIdx.x;
    AGX_STAMP(0);
    const int BR = p.band_rows;
    const int dy0 = band * BR;
    const int rows = min(BR, p.oh - dy0);
    int4 *ytab_s = reinterpret_cast<int4 *>(smem);                      // [BR]
    int2 *xtab_s = reinterpret_cast<int2 *>(smem + sizeof(int4) * BR);    // [ow]
    unsigned char *gray = smem + sizeof(int4) * BR + sizeof(int2) * p.ow; // [2][BR][160][2]
    const int ow4 = p.ow >> 2;
    if (!p.y_affine) {                       // general geometry: source rows come from the table
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
    const uint8_t *fbase = p.frames + (size_t)n * 2 * kRawFrameBytes;     // wave-uniform base
    const uint32_t col = g4 * 12;
    int nvalid = 2;                                                       // speculative until cmd arrives
    auto row_offsets = [&](int it, uint32_t &o0, uint32_t &o1, int &d) {
        const int nrj = max(nvalid, 1) * rows;
        const int rj_raw = rg + RG * it;
        const int rj = min(rj_raw, nrj - 1);
        const int f = rj >= rows ? 1 : 0;                                 // nvalid <= 2
        const int dyl = rj - f * rows;
        int y0, y1;
        if (p.y_affine) {
            y0 = (int)(mul_u24((uint32_t)(dy0 + dyl), (uint32_t)p.y_mul) + (uint32_t)p.y_add) >> p.y_shift;
            y1 = min(y0 + 1, kRawH - 1);
        } else {
            const int4 yt = ytab_s[dyl];
            y0 = yt.x;
            y1 = yt.y;
        }
        const uint32_t fo = f * kRawFrameBytes + col;                     // 32-bit lane offsets
        o0 = mad_u24((uint32_t)y0, kRawRowBytes, fo);
        o1 = mad_u24((uint32_t)y1, kRawRowBytes, fo);
        d = (rj_raw < nvalid * rows && rg < RG) ? ((f * BR + dyl) * kRawW + g4 * 4) * 2 : -1;
    };
    U3 w0[kIter], w1[kIter];
    int dst[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
        uint32_t o0, o1;
        row_offsets(it, o0, o1, dst[it]);
        w0[it] = *reinterpret_cast<const U3 *>(fbase + o0);
        w1[it] = *reinterpret_cast<const U3 *>(fbase + o1);
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
            if (rg + RG * it >= nrj) dst[it] = -1;
        AGX_STAMP(1);
        // the phase-2 tables are requested AFTER the frame pieces (vmcnt retires in order, so waiting
        // for them later costs nothing) and parked in LDS once the luminance work is done
        const int4 yt_own = p.ytab[dy0 + min(tid, rows - 1)];
        const int2 xt_own = p.xtab[min(tid, p.ow - 1)];
        uint32_t tie_its = 0;
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            bool tie = false;
            const uint32_t top = lum4(w0[it].x, w0[it].y, w0[it].z, tie);
            const uint32_t bot = lum4(w1[it].x, w1[it].y, w1[it].z, tie);
            if (dst[it] >= 0) {
                uint2 v;                                                  // t0 b0 t1 b1 | t2 b2 t3 b3
                v.x = __builtin_amdgcn_perm(bot, top, 0x05010400u);
                v.y = __builtin_amdgcn_perm(bot, top, 0x07030602u);
                *reinterpret_cast<uint2 *>(gray + dst[it]) = v;
                tie_its |= tie ? (1u << it) : 0u;
            }
        }
        if (__builtin_expect(tie_its != 0, 0)) {
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
        const int dyl = tid / ow4, xq = tid - dyl * ow4;
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
        uint32_t packed = 0;
        const int xi[4] = {xt01.x, xt01.z, xt23.x, xt23.z};
        const int xa[4] = {xt01.y, xt01.w, xt23.y, xt23.w};
        const unsigned char *row0 = gray + mul_u24((uint32_t)dyl, kRawW * 2);      // frame 0, this output row
        const uint32_t fstride = (uint32_t)BR * kRawW * 2;                         // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x0 = xi[k] & 0xFFFF, x1 = (uint32_t)xi[k] >> 16;
            const uint32_t a0 = xa[k] & 0xFFFF, a1 = (uint32_t)xa[k] >> 16;
            uint32_t best = 0;
            for (int f = 0; f < nvalid; ++f) {
                const uint16_t *row = reinterpret_cast<const uint16_t *>(row0 + f * fstride);
                const uint32_t p0 = row[x0], p1 = row[x1];               // lo byte: row y0, hi byte: row y1
                const uint32_t h0 = mad_u24(p1 & 0xFF, a1, mul_u24(p0 & 0xFF, a0));
                const uint32_t h1 = mad_u24(p1 >> 8, a1, mul_u24(p0 >> 8, a0));
                const uint32_t v = (((mul_u24(b0, h0 >> 4) >> 16) + (mul_u24(b1, h1 >> 4) >> 16) + 2) >> 2) & 0xFF;
                best = max(best, v);
            }
            packed |= best << (8 * k);
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

template <int T>
__global__ __launch_bounds__(T) void k_ingest(IngestParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ingest_band<T>(p, blockIdx.x, blockIdx.y, smem);
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

// ---------------------------------------------------------------------------------------------
// K0: stack-order views of the ring
// ---------------------------------------------------------------------------------------------
struct StackParams {
    uint8_t *ring;
    int32_t *head;           // current head (read), or written by k_set_stack
    const uint8_t *in_u8;
    uint8_t *out_u8;
    float *out_f32;
    int32_t words, fs;       // words = oh*ow/4
};

// grid = (ceil(words/256), fs, N)
__global__ __launch_bounds__(kThreads) void k_stack_u8(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= p.words) return;
    int slot = p.head[n] + j;
    if (slot >= p.fs) slot -= p.fs;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.ring) + ((size_t)n * p.fs + slot) * p.words;
    reinterpret_cast<uint32_t *>(p.out_u8)[((size_t)n * p.fs + j) * p.words + i] = src[i];
}

__global__ __launch_bounds__(kThreads) void k_set_stack(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i == 0 && j == 0) p.head[n] = 0;
    if (i >= p.words) return;
    const size_t o = ((size_t)n * p.fs + j) * p.words + i;
    reinterpret_cast<uint32_t *>(p.ring)[o] = reinterpret_cast<const uint32_t *>(p.in_u8)[o];
}

__global__ __launch_bounds__(kThreads) void k_full(StackParams p) {
    const int n = blockIdx.z, j = blockIdx.y;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= p.words) return;
    int slot = p.head[n] + j;
    if (slot >= p.fs) slot -= p.fs;
    const uint32_t v = (reinterpret_cast<const uint32_t *>(p.ring) + ((size_t)n * p.fs + slot) * p.words)[i];
    float4 o;
    o.x = unit(v & 0xFF);
    o.y = unit((v >> 8) & 0xFF);
    o.z = unit((v >> 16) & 0xFF);
    o.w = unit(v >> 24);
    reinterpret_cast<float4 *>(p.out_f32)[((size_t)n * p.fs + j) * p.words + i] = o;
}

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
//   * the whole u8 frame of that slot (address known at launch) -> registers -> LDS,
//   * the scalar chain action / fov_loc / head -> (r, c) and the stack position j of this slot,
//   * this thread's column taps (registers) and one row-tap entry (-> LDS).
// u8 -> float32 k/255 goes through a 256-entry LDS table (one exact division per thread).
//   RESIZE: H[fh][ow] = horizontal lerp of the window rows (thread = fixed column x, rows y = yb+3k),
//           then each output float4 is the vertical lerp of two ds_read_b128; stores are 16 B per
//           lane, lane-linear, 1 KiB per wave at 1-KiB steps, nontemporal.
// (ablation of the previous serial version at N=1024: loc chain 5.8 us, loc-dependent window load
//  6.3 us, H pass with a tap load per iteration 7.3 us, row-tap loads 2.1 us of a 32.7 us launch.)
template <class G, int MODE>
__device__ __forceinline__ void fovea_fixed_body(const G g, const FovParams &p, const int sl, const int n,
                                                 unsigned char *smem) {
    const int tid = threadIdx.x;
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
        if (p.phase == 1) {
            wslot = h;
            head_fixup = skip ? 0 : (clear ? -h : (h + 1 == p.fs ? 1 - p.fs : 1));
        } else {
            wslot = skip ? h : (h == 0 ? p.fs - 1 : h - 1);
        }
        const bool touched = clear || sl == wslot;
        if ((p.phase == 1) == touched) return;            // phase 1 takes the untouched slots, phase 2 the rest
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
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    const int fwords = fbytes >> 2;
    constexpr int kFW = 7;                                            // 7 * 256 dwords cover 84x84; loop beyond
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k) fw_[k] = fsrc[min(tid + k * kThreads, fwords - 1)];
    const int xcol = tid % ow, yb = tid / ow;                         // phase-C column / first row
    int4 xt = make_int4(0, 0, 0, 0), yt = xt;                         // raw Tap bits {lo, aux, a, b}
    if (MODE == AGX_OUT_RESIZE) {
        xt = *reinterpret_cast<const int4 *>(p.xtab + xcol);
        yt = *reinterpret_cast<const int4 *>(p.ytab + min(tid, oh - 1));
    }
    const LocIn lin = load_loc_inputs(p, n);
    const int head = p.head[n] + head_fixup;
    lut[tid] = unit((uint32_t)tid);
    int r, c;
    compute_loc(p, lin, oh - fh, ow - fw, r, c);
    int j = sl - head;                                                // stack position of this slot
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
    for (int i = tid + kFW * kThreads; i < fwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = fsrc[i];
    if (MODE == AGX_OUT_RESIZE) {
        if (tid < oh) *reinterpret_cast<int4 *>(ytab_s + tid) = yt;
        for (int i = tid + kThreads; i < oh; i += kThreads) ytab_s[i] = p.ytab[i];
    }
    AGX_STAMP(1);
    __syncthreads();
    AGX_STAMP(2);

    const unsigned char *win = raw + r * ow + c;                      // window origin inside the frame
    if (MODE == AGX_OUT_RAW) {
        float *out = p.obs + ((size_t)n * p.fs + j) * (size_t)(fh * fw);
        for (int i = tid; i < fh * fw; i += kThreads) {
            const int y = i / fw, x = i - y * fw;
            out[i] = lut[win[y * ow + x]];
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
                const uint32_t w = *reinterpret_cast<const uint32_t *>(raw + row * ow + x);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k >= c && x + k < c + fw) v[k] = lut[(w >> (8 * k)) & 0xFF];
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
                H[y * ow + xcol] = wa * lut[c0[y * ow]] + wb * lut[c1[y * ow]];
        }
    } else {                                                          // ow > 256: generic striding
        for (int i = tid; i < fh * ow; i += kThreads) {
            const int y = i / ow, x = i - y * ow;
            const Tap t = p.xtab[x];
            H[i] = t.a * lut[win[y * ow + t.lo]] + t.b * lut[win[y * ow + t.aux]];
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
        float4 o;
        o.x = t.a * a.x + t.b * b.x;
        o.y = t.a * a.y + t.b * b.y;
        o.z = t.a * a.z + t.b * b.z;
        o.w = t.a * a.w + t.b * b.w;
        store_obs(&out4[q], o);
    }
    AGX_STAMP(4);
}

template <class G, int MODE>
__global__ __launch_bounds__(kThreads) void k_fovea_fixed(G g, FovParams p) {
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
                for (int y = yb; y < fh; y += rstep) H[y * ow + xcol] = wa * lut[c0[y * ow]] + wb * lut[c1[y * ow]];
            }
        } else {
            for (int i = tid; i < fh * ow; i += kThreads) {
                const int y = i / ow, x = i - y * ow;
                const Tap t = p.xtab[x];
                H[i] = t.a * lut[win[y * ow + t.lo]] + t.b * lut[win[y * ow + t.aux]];
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
            o.x = t.a * a.x + t.b * b.x;
            o.y = t.a * a.y + t.b * b.y;
            o.z = t.a * a.z + t.b * b.z;
            o.w = t.a * a.w + t.b * b.w;
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

// ---------------------------------------------------------------------------------------------
// generic separable resample pass inside one workgroup (K3 / K4)
// ---------------------------------------------------------------------------------------------
struct PassDesc {
    int n_in, n_out;
    bool aa;         // antialiased down-scale (n_in > n_out and antialias on)
    float inv;       // aa: 1/scale
};

__device__ __forceinline__ PassDesc make_pass(int n_in, int n_out, int antialias) {
    PassDesc d;
    d.n_in = n_in;
    d.n_out = n_out;
    d.aa = antialias && n_in > n_out;
    d.inv = d.aa ? (float)((double)n_out / (double)n_in) : 1.f;
    return d;
}

__device__ __forceinline__ void build_taps(const PassDesc &d, Tap *tab, int tid) {
    for (int i = tid; i < d.n_out; i += kThreads) {
        float inv;
        tab[i] = d.aa ? make_tap_aa(i, d.n_in, d.n_out, &inv) : make_tap_lin2(i, d.n_in, d.n_out);
    }
}

// element of a pass: src walks with `stride` floats between consecutive taps
__device__ __forceinline__ float apply_tap(const PassDesc &d, const Tap &t, const float *src, int stride) {
    if (!d.aa) return t.a * src[t.lo * stride] + t.b * src[t.aux * stride];
    float acc = 0.f;
    const float *q = src + t.lo * stride;
    for (int k = 0; k < t.aux; ++k) {
        float x = ((float)k - t.a + 0.5f) * d.inv;
        x = fabsf(x);
        const float w = x < 1.f ? 1.f - x : 0.f;
        acc += w * q[k * stride];
    }
    return acc * t.b;
}

// dst[rows][n_out] = resample along W of src[rows][n_in]
__device__ __forceinline__ void pass_w(const PassDesc &d, const Tap *tab, const float *src, float *dst,
                                       int rows, int tid) {
    const int total = rows * d.n_out;
    for (int i = tid; i < total; i += kThreads) {
        const int y = i / d.n_out, x = i - y * d.n_out;
        dst[i] = apply_tap(d, tab[x], src + y * d.n_in, 1);
    }
}

// dst[n_out][cols] = resample along H of src[n_in][cols]
__device__ __forceinline__ void pass_h(const PassDesc &d, const Tap *tab, const float *src, float *dst,
                                       int cols, int tid) {
    const int total = d.n_out * cols;
    for (int i = tid; i < total; i += kThreads) {
        const int y = i / cols, x = i - y * cols;
        dst[i] = apply_tap(d, tab[y], src + x, cols);
    }
}

// ---------------------------------------------------------------------------------------------
// K3: FixedFovealPeripheralEnv, K4: FlexibleFovealEnv.  grid = (fs, N), block = 256.
// LDS: buf0, buf1 (oh*ow floats each), tab (max(oh,ow,..) taps)
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(kThreads) void k_fovea_generic(GeomR g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int j = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh(), fw = g.fw();
    const bool flex = KIND == AGX_KIND_FLEXIBLE;
    if (p.mask && !p.mask[n]) {
        if (j == 0 && tid < 2) {
            p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
            if (flex) p.res_out[2 * n + tid] = p.res_in[2 * n + tid];
        }
        return;
    }
    // ---- state update
    int rh = fh, rw = fw, r, c;
    if (flex) {
        rh = p.res_in[2 * n];
        rw = p.res_in[2 * n + 1];
        const int type = (p.action && p.action_type) ? p.action_type[n] : AGX_FOV_LOC;
        if (p.action && type == AGX_FOV_RES) {
            rh = clip_rint(load_action(p.action, p.action_dt, 2 * (size_t)n), 1.0, (double)oh);
            rw = clip_rint(load_action(p.action, p.action_dt, 2 * (size_t)n + 1), 1.0, (double)ow);
            r = clip_rint((double)p.loc_in[2 * n], 0.0, (double)(oh - rh));
            c = clip_rint((double)p.loc_in[2 * n + 1], 0.0, (double)(ow - rw));
        } else {
            next_loc(p, n, oh - rh, ow - rw, r, c);
        }
    } else {
        next_loc(p, n, oh - fh, ow - fw, r, c);
    }
    if (j == 0 && tid == 0) {
        p.loc_out[2 * n] = r;
        p.loc_out[2 * n + 1] = c;
        if (p.user_loc) {
            p.user_loc[2 * n] = r;
            p.user_loc[2 * n + 1] = c;
        }
        if (flex) {
            p.res_out[2 * n] = rh;
            p.res_out[2 * n + 1] = rw;
            if (p.user_res) {
                p.user_res[2 * n] = rh;
                p.user_res[2 * n + 1] = rw;
            }
        }
    }
    int slot = p.head[n] + j;
    if (slot >= p.fs) slot -= p.fs;
    const size_t fsz = (size_t)oh * ow;
    const uint8_t *frame = p.ring + ((size_t)n * p.fs + slot) * fsz;
    const int cap = (oh * ow + 3) & ~3;
    float *buf0 = reinterpret_cast<float *>(smem);
    float *buf1 = buf0 + cap;
    Tap *tab = reinterpret_cast<Tap *>(buf1 + p.buf1_floats);
    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);

    if (KIND == AGX_KIND_PERIPHERAL) {
        // S = full frame; periphery = expand(squeeze(S)); fovea pasted at full resolution
        const int ph = p.per_h, pw = p.per_w;
        float *S = buf0;
        stage_window(frame, ow, 0, 0, oh, ow, S, tid);
        // the three intermediates share buf1: A[oh][pw] | B[ph][pw] | C[ph][ow]
        float *A = buf1;
        float *B = A + oh * pw;
        float *C = B + ph * pw;
        const bool same = (ph == oh && pw == ow);           // torchvision returns the input unchanged
        PassDesc d = make_pass(ow, pw, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_w(d, tab, S, A, oh, tid);
        __syncthreads();
        d = make_pass(oh, ph, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_h(d, tab, A, B, pw, tid);
        __syncthreads();
        d = make_pass(pw, ow, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_w(d, tab, B, C, ph, tid);
        __syncthreads();
        d = make_pass(ph, oh, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = q / ow4, x = (q - row * ow4) * 4;
            const Tap t = tab[row];
            const bool in_r = row >= r && row < r + fh;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + k;
                if (same || (in_r && xx >= c && xx < c + fw))
                    v[k] = S[row * ow + xx];
                else
                    v[k] = apply_tap(d, t, C + xx, ow);
            }
            out4[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
        return;
    }

    // ---- flexible
    float *cur = buf0, *oth = buf1;
    stage_window(frame, ow, r, c, rh, rw, cur, tid);
    __syncthreads();
    if (rh > fh) {                                           // rows only, fov_env.py:286
        // Resize(fov_size) then Resize(fov_res): [rh][rw] -> [rh][fw] -> [fh][fw] -> [fh][rw] -> [rh][rw]
        PassDesc d = make_pass(rw, fw, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_w(d, tab, cur, oth, rh, tid);
        __syncthreads();
        d = make_pass(rh, fh, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_h(d, tab, oth, cur, fw, tid);
        __syncthreads();
        d = make_pass(fw, rw, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_w(d, tab, cur, oth, fh, tid);
        __syncthreads();
        d = make_pass(fh, rh, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        pass_h(d, tab, oth, cur, rw, tid);
        __syncthreads();
    }
    if (p.out_mode == AGX_OUT_RESIZE && !(rh == oh && rw == ow)) {
        PassDesc d = make_pass(rw, ow, p.antialias);         // res <= obs: never a down-scale
        build_taps(d, tab, tid);
        __syncthreads();
        pass_w(d, tab, cur, oth, rh, tid);                   // [rh][ow]
        __syncthreads();
        d = make_pass(rh, oh, p.antialias);
        build_taps(d, tab, tid);
        __syncthreads();
        const float4 *H4 = reinterpret_cast<const float4 *>(oth);
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = q / ow4, x4 = q - row * ow4;
            const Tap t = tab[row];
            const float4 a = H4[t.lo * ow4 + x4];
            const float4 b = H4[t.aux * ow4 + x4];
            out4[q] = make_float4(t.a * a.x + t.b * b.x, t.a * a.y + t.b * b.y,
                                  t.a * a.z + t.b * b.z, t.a * a.w + t.b * b.w);
        }
        return;
    }
    // mask-out paste at (r, c); raw (padded, window at the origin); resize with res == obs (identity)
    const int pr = (p.out_mode == AGX_OUT_MASK) ? r : 0;
    const int pc = (p.out_mode == AGX_OUT_MASK) ? c : 0;
    for (int q = tid; q < oh * ow4; q += kThreads) {
        const int row = q / ow4, x = (q - row * ow4) * 4;
        const int y = row - pr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < rh) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + k - pc;
                if (xx >= 0 && xx < rw) v[k] = cur[y * rw + xx];
            }
        }
        out4[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// K3, tuned form: FixedFovealPeripheralEnv with the context's fixed geometry.
// grid = (fs, N): workgroup (sl, n) owns physical ring slot sl; block = 256.
// The four separable passes use tap tables built on the HOST at agx_create (ATen's arithmetic in
// double, weights normalised, narrowed to f32): per output index {lo, n} and n weights.
//   raw u8 frame --W squeeze--> A[oh][pw] --H squeeze--> B[ph][pw] --W expand--> C[ph][ow]
//   --H expand, fused with the full-resolution fovea paste and the nontemporal store.
// u8 -> f32 through the 256-entry LDS table.  ~20 KB LDS for 84/20 (C aliases A): 8 workgroups per CU
// (the generic kernel it replaces needed 43 KB and built its taps in f64 on the device).
// ---------------------------------------------------------------------------------------------
struct AxisTab {            // device pointers, one per pass
    const int2 *ln;         // [n_out] {lo, n}
    const float *w;         // [n_out][maxt]
    int32_t n_out, maxt;
};
struct PerParams {
    AxisTab t[4];           // 0: W squeeze (ow->pw), 1: H squeeze (oh->ph), 2: W expand (pw->ow), 3: H expand (ph->oh)
    int32_t oh, ow, fh, fw, ph, pw, same;
};

// MT = compile-time bound of the squeeze passes' tap count (tables are zero-padded to it by the host);
// MT == 0 keeps run-time trip counts.  With a fixed bound every LDS read of an output is issued before
// the first FMA instead of one dependent read pair per tap.
template <int MT>
__global__ __launch_bounds__(kThreads) void k_fovea_peripheral2(PerParams g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh, ow = g.ow, fh = g.fh, fw = g.fw, ph = g.ph, pw = g.pw;
    if (p.mask && !p.mask[n]) {
        if (sl == 0 && tid < 2) p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
        return;
    }
    constexpr int MTR = MT > 0 ? MT : 1;
    const int fbytes = oh * ow, fwords = fbytes >> 2;
    const int mt1 = g.t[1].maxt, mt3 = g.t[3].maxt;
    // LDS: lut[256] | raw[oh*ow] | A[oh][pw] aliased by C[ph][ow] | B[ph][pw] | pass-1 table | pass-3 table
    float *lut = reinterpret_cast<float *>(smem);
    unsigned char *raw = smem + 1024;
    float *A = reinterpret_cast<float *>(raw + ((fbytes + 15) & ~15));             // 16-B aligned
    float *C = A;                  // C reuses A's floats: A is dead once pass 1 has produced B (a barrier lies between)
    float *B = A + ((max(oh * pw, ph * ow) + 3) & ~3);
    int2 *ln1_s = reinterpret_cast<int2 *>(B + ((ph * pw + 3) & ~3));              // [ph]   (layout as per2_lds)
    float *w1_s = reinterpret_cast<float *>(ln1_s + ph);                           // [ph][mt1]
    int2 *ln3_s = reinterpret_cast<int2 *>(w1_s + ph * mt1);                       // [oh]
    float *w3_s = reinterpret_cast<float *>(ln3_s + oh);                           // [oh][mt3]

    // ---- every round trip starts now: the frame, this thread's pass-0 / pass-2 taps (registers), the
    // pass-1 / pass-3 tables (-> LDS), then the small state loads
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    constexpr int kFW = 7;
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k) fw_[k] = fsrc[min(tid + k * kThreads, fwords - 1)];
    const int xp0 = tid % pw, y00 = tid / pw, per0 = kThreads / pw;                // pass 0: column xp0, rows y00 + per0*i
    const int2 ln0 = g.t[0].ln[xp0];
    float wr0[MTR];
    if (MT > 0) {
#pragma unroll
        for (int k = 0; k < MT; ++k) wr0[k] = g.t[0].w[xp0 * g.t[0].maxt + k];
    }
    const int per2 = kThreads / ow;                                                // pass 2: column x2, rows y20 + per2*i
    const int x2 = per2 > 0 ? tid % ow : 0, y20 = per2 > 0 ? tid / ow : 0;
    const int2 ln2 = g.t[2].ln[x2];
    const float w2a = g.t[2].w[x2 * g.t[2].maxt], w2b = g.t[2].maxt > 1 ? g.t[2].w[x2 * g.t[2].maxt + 1] : 0.f;
    for (int i = tid; i < ph; i += kThreads) ln1_s[i] = g.t[1].ln[i];
    for (int i = tid; i < ph * mt1; i += kThreads) w1_s[i] = g.t[1].w[i];
    for (int i = tid; i < oh; i += kThreads) ln3_s[i] = g.t[3].ln[i];
    for (int i = tid; i < oh * mt3; i += kThreads) w3_s[i] = g.t[3].w[i];
    const LocIn lin = load_loc_inputs(p, n);
    const int head = p.head[n];
    lut[tid] = unit((uint32_t)tid);
    int r, c;
    compute_loc(p, lin, oh - fh, ow - fw, r, c);
    int j = sl - head;
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
    for (int i = tid + kFW * kThreads; i < fwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = fsrc[i];
    __syncthreads();

    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    if (!g.same) {
        // pass 0: A[y][xp] = sum_k (w0[xp][k] / 255) * raw[y][lo + k]
        // The pass-0 weights carry the 1/255 (host side), so bytes convert with v_cvt_f32_ubyteN and no
        // table lookup: sum_k (w_k/255) * b_k differs from sum_k w_k * f32(b_k/255) by < 1e-7, far inside
        // the 1e-5 bar of the float resize path (the pasted fovea keeps the exact table).
        if (y00 < per0) {
            if (MT > 0) {
                constexpr int NDW = (MTR + 6) / 4;                   // aligned dwords covering (lo & 3) + MT bytes
                for (int y = y00; y < oh; y += per0) {
                    const int off = y * ow + ln0.x;
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(raw + (off & ~3));
                    const uint32_t sh = off & 3;
                    uint32_t d[NDW];
#pragma unroll
                    for (int k = 0; k < NDW; ++k) d[k] = src[k];     // reads past n hit zero weights
                    float acc = 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < MT / 4; ++q4) {
                        const uint32_t v = __builtin_amdgcn_alignbyte(d[q4 + 1], d[q4], sh);
                        acc = fmaf(wr0[4 * q4 + 0], (float)(v & 0xFF), acc);
                        acc = fmaf(wr0[4 * q4 + 1], (float)((v >> 8) & 0xFF), acc);
                        acc = fmaf(wr0[4 * q4 + 2], (float)((v >> 16) & 0xFF), acc);
                        acc = fmaf(wr0[4 * q4 + 3], (float)(v >> 24), acc);
                    }
                    if (MT % 4) {
                        const uint32_t v = __builtin_amdgcn_alignbyte(d[MT / 4 + 1], d[MT / 4], sh);
#pragma unroll
                        for (int k = 0; k < MT % 4; ++k) acc = fmaf(wr0[(MT / 4) * 4 + k], (float)((v >> (8 * k)) & 0xFF), acc);
                    }
                    A[y * pw + xp0] = acc;
                }
            } else {
                const float *w = g.t[0].w + xp0 * g.t[0].maxt;
                for (int y = y00; y < oh; y += per0) {
                    const unsigned char *src = raw + y * ow + ln0.x;
                    float acc = 0.f;
                    for (int k = 0; k < ln0.y; ++k) acc = fmaf(w[k], (float)src[k], acc);
                    A[y * pw + xp0] = acc;
                }
            }
        }
        __syncthreads();
        // pass 1: B[yp][xp] = sum_k w1[yp][k] * A[lo + k][xp]
        for (int i = tid; i < ph * pw; i += kThreads) {
            const int yp = i / pw, xp = i - yp * pw;
            const int2 ln = ln1_s[yp];
            const float *w = w1_s + yp * mt1;
            float acc = 0.f;
            if (MT > 0) {
                float v[MTR];
#pragma unroll
                for (int k = 0; k < MT; ++k) v[k] = A[min(ln.x + k, oh - 1) * pw + xp];   // clamped: weight is 0 there
#pragma unroll
                for (int k = 0; k < MT; ++k) acc = fmaf(w[k], v[k], acc);
            } else {
                for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], A[(ln.x + k) * pw + xp], acc);
            }
            B[i] = acc;
        }
        __syncthreads();
        // pass 2: C[yp][x] = sum_k w2[x][k] * B[yp][lo + k]     (expansion: at most 2 taps when pw <= ow)
        if (per2 > 0 && g.t[2].maxt <= 2) {
            if (y20 < per2) {
                const int i1 = ln2.y > 1 ? ln2.x + 1 : ln2.x;
                for (int yp = y20; yp < ph; yp += per2)
                    C[yp * ow + x2] = fmaf(w2b, B[yp * pw + i1], w2a * B[yp * pw + ln2.x]);
            }
        } else {
            const AxisTab &t = g.t[2];
            for (int i = tid; i < ph * ow; i += kThreads) {
                const int yp = i / ow, x = i - yp * ow;
                const int2 ln = t.ln[x];
                const float *w = t.w + x * t.maxt;
                float acc = 0.f;
                for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], B[yp * pw + ln.x + k], acc);
                C[i] = acc;
            }
        }
        __syncthreads();
    }
    // pass 3 fused with paste + store: out[row][x..x+3]
    const float4 *C4 = reinterpret_cast<const float4 *>(C);
    for (int q = tid; q < oh * ow4; q += kThreads) {
        const int row = q / ow4, x4 = q - row * ow4, x = x4 * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in_r = row >= r && row < r + fh;
        const bool all_fov = g.same || (in_r && x >= c && x + 3 < c + fw);
        if (!all_fov) {
            const int2 ln = ln3_s[row];
            const float *w = w3_s + row * mt3;
            for (int k = 0; k < ln.y; ++k) {
                const float4 v = C4[(ln.x + k) * ow4 + x4];
                o.x = fmaf(w[k], v.x, o.x);
                o.y = fmaf(w[k], v.y, o.y);
                o.z = fmaf(w[k], v.z, o.z);
                o.w = fmaf(w[k], v.w, o.w);
            }
        }
        if (g.same || (in_r && x + 3 >= c && x < c + fw)) {
            const uint32_t wv = *reinterpret_cast<const uint32_t *>(raw + row * ow + x);
            if (g.same || (x >= c && x < c + fw)) o.x = lut[wv & 0xFF];
            if (g.same || (x + 1 >= c && x + 1 < c + fw)) o.y = lut[(wv >> 8) & 0xFF];
            if (g.same || (x + 2 >= c && x + 2 < c + fw)) o.z = lut[(wv >> 16) & 0xFF];
            if (g.same || (x + 3 >= c && x + 3 < c + fw)) o.w = lut[wv >> 24];
        }
        store_obs(&out4[q], o);
    }
}

// ---------------------------------------------------------------------------------------------
// K4, tuned form: FlexibleFovealEnv (per-env ragged window rh x rw).  grid = (fs, N), block = 256,
// workgroup (sl, n) owns physical ring slot sl.
// Tap tables for every window size r come from the HOST (agx_create): per axis three families,
//   dwn[r]: r -> fov (the squeeze, antialiased when r > fov and antialias is on)
//   bck[r]: fov -> r (the expansion back; an antialiased DOWN-scale when r < fov)
//   fin[r]: r -> obs (the final resize_to_full, always an up-scale)
// The reference's chain  crop -> Resize(fov_size) -> Resize(fov_res) -> Resize(obs_size)
// (fov_env.py:276-298) is evaluated without its two largest intermediates:
//   A[rh][fw] = Wdwn(crop)   B[fh][fw] = Hdwn(A)   C[fh][rw] = Wbck(B)
//   resize: E[fh][ow] = Wfin(C), out[y] = sum_a Hfin[y][a] * sum_b Hbck[i_a][b] * E[j_ab]   (H passes composed)
//   mask / raw: out[y][x] = sum_b Hbck[y][b] * C[j_b][x]
// (W and H passes act on different axes and commute; only float rounding differs, ~1e-7.)
// ---------------------------------------------------------------------------------------------
struct TabFamily {
    const int2 *ln;      // {lo, n} entries of all sizes, concatenated
    const float *w;      // weights, pitch meta[r].z per entry
    const int4 *meta;    // [rmax + 1]: {first entry, first weight, maxt, entry count} of size r
};
struct FlexParams {
    TabFamily wd, wb, wf, hd, hb, hf;
    int32_t oh, ow, fh, fw;
};

struct LdsTab {          // one staged table
    const int2 *ln;
    const float *w;
    int maxt;
};
// copy the table of size r into LDS at float offset `off` (kept a multiple of 4 floats)
__device__ __forceinline__ LdsTab stage_tab(const TabFamily &f, int r, float *base, int &off, int tid) {
    const int4 m = f.meta[r];
    int2 *ln = reinterpret_cast<int2 *>(base + off);
    float *w = base + off + 2 * m.w;
    for (int i = tid; i < m.w; i += kThreads) ln[i] = f.ln[m.x + i];
    for (int i = tid; i < m.w * m.z; i += kThreads) w[i] = f.w[m.y + i];
    off = (off + 2 * m.w + m.w * m.z + 3) & ~3;
    LdsTab t{ln, w, m.z};
    return t;
}
__device__ __forceinline__ float tap_dot(const LdsTab &t, int i, const float *src, int stride) {
    const int2 ln = t.ln[i];
    const float *w = t.w + i * t.maxt;
    float acc = 0.f;
    for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], src[(ln.x + k) * stride], acc);
    return acc;
}

__global__ __launch_bounds__(kThreads) void k_fovea_flexible2(FlexParams g, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh, ow = g.ow, fh = g.fh, fw = g.fw;
    if (p.mask && !p.mask[n]) {
        if (sl == 0 && tid < 2) {
            p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
            p.res_out[2 * n + tid] = p.res_in[2 * n + tid];
        }
        return;
    }
    const int fbytes = oh * ow, fwords = fbytes >> 2;
    // LDS: lut[256] | raw[oh*ow] | AE[max(oh*fw, fh*ow)] | B[fh*fw] | C[fh*ow] | tables
    float *lut = reinterpret_cast<float *>(smem);
    unsigned char *raw = smem + 1024;
    float *AE = reinterpret_cast<float *>(raw + ((fbytes + 15) & ~15));
    const int ae_floats = (max(oh * fw, fh * ow) + 3) & ~3;
    float *B = AE + ae_floats;
    float *C = B + ((fh * fw + 3) & ~3);
    float *tabs = C + ((fh * ow + 3) & ~3);

    // ---- round trips start now
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    constexpr int kFW = 7;
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k) fw_[k] = fsrc[min(tid + k * kThreads, fwords - 1)];
    const LocIn lin = load_loc_inputs(p, n);
    const int2 res_old = *reinterpret_cast<const int2 *>(p.res_in + 2 * n);
    const int type = (p.action && p.action_type) ? p.action_type[n] : AGX_FOV_LOC;
    const int head = p.head[n];
    lut[tid] = unit((uint32_t)tid);
    // ---- state update (fov_env.py:300-324)
    int rh = res_old.x, rw = res_old.y, r, c;
    if (p.action && type == AGX_FOV_RES) {
        rh = clip_rint(action_value(p.action_dt, lin.w[0], lin.w[1]), 1.0, (double)oh);
        rw = clip_rint(action_value(p.action_dt, lin.w[2], lin.w[3]), 1.0, (double)ow);
        r = clip_rint((double)lin.r, 0.0, (double)(oh - rh));
        c = clip_rint((double)lin.c, 0.0, (double)(ow - rw));
    } else {
        compute_loc(p, lin, oh - rh, ow - rw, r, c);
    }
    int j = sl - head;
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
    const bool squeeze = rh > fh;                                 // rows only, fov_env.py:286
    const bool resize = p.out_mode == AGX_OUT_RESIZE;
    // ---- stage the tables this window needs (L2 hits; their latency hides under the frame load)
    int toff = 0;
    LdsTab wd{}, hd{}, wb{}, hb{}, wf{}, hf{};
    if (squeeze) {
        wd = stage_tab(g.wd, rw, tabs, toff, tid);
        hd = stage_tab(g.hd, rh, tabs, toff, tid);
        wb = stage_tab(g.wb, rw, tabs, toff, tid);
        hb = stage_tab(g.hb, rh, tabs, toff, tid);
    }
    if (resize) {
        wf = stage_tab(g.wf, rw, tabs, toff, tid);
        hf = stage_tab(g.hf, rh, tabs, toff, tid);
    }
#pragma unroll
    for (int k = 0; k < kFW; ++k)
        if (tid + k * kThreads < fwords) reinterpret_cast<uint32_t *>(raw)[tid + k * kThreads] = fw_[k];
    for (int i = tid + kFW * kThreads; i < fwords; i += kThreads) reinterpret_cast<uint32_t *>(raw)[i] = fsrc[i];
    __syncthreads();

    const unsigned char *win = raw + r * ow + c;
    const float kInv255 = 1.0f / 255.0f;          // resampling inputs only (<= 1 ulp from k/255); pasted pixels use lut
    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);

    if (squeeze) {
        // P1: A[y][xf] = Wdwn(crop)      y < rh, xf < fw
        for (int i = tid; i < rh * fw; i += kThreads) {
            const int y = i / fw, xf = i - y * fw;
            const int2 ln = wd.ln[xf];
            const float *w = wd.w + xf * wd.maxt;
            const unsigned char *src = win + y * ow + ln.x;
            float acc = 0.f;
            for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], (float)src[k], acc);
            AE[i] = acc * kInv255;
        }
        __syncthreads();
        // P2: B[yf][xf] = Hdwn(A)
        for (int i = tid; i < fh * fw; i += kThreads) {
            const int yf = i / fw, xf = i - yf * fw;
            B[i] = tap_dot(hd, yf, AE + xf, fw);
        }
        __syncthreads();
        // P3: C[yf][x] = Wbck(B)         x < rw
        for (int i = tid; i < fh * rw; i += kThreads) {
            const int yf = i / rw, x = i - yf * rw;
            C[yf * ow + x] = tap_dot(wb, x, B + yf * fw, 1);
        }
        __syncthreads();
    }

    if (resize) {
        // E[y][xo] = Wfin(src rows): src = C (fh rows) after a squeeze, else the crop itself (rh rows)
        const int erows = squeeze ? fh : rh;
        for (int i = tid; i < erows * ow; i += kThreads) {
            const int y = i / ow, xo = i - y * ow;
            const int2 ln = wf.ln[xo];
            const float *w = wf.w + xo * wf.maxt;
            float acc = 0.f;
            if (squeeze) {
                for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], C[y * ow + ln.x + k], acc);
            } else {
                const unsigned char *src = win + y * ow + ln.x;
                for (int k = 0; k < ln.y; ++k) acc = fmaf(w[k], (float)src[k], acc);
                acc *= kInv255;
            }
            AE[i] = acc;
        }
        __syncthreads();
        const float4 *E4 = reinterpret_cast<const float4 *>(AE);
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = q / ow4, x4 = q - row * ow4;
            const int2 lnf = hf.ln[row];
            const float *wfv = hf.w + row * hf.maxt;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int a = 0; a < lnf.y; ++a) {
                const int ya = lnf.x + a;                      // row of the (virtual) rh-row image
                if (squeeze) {
                    const int2 lnb = hb.ln[ya];
                    const float *wbv = hb.w + ya * hb.maxt;
                    for (int b = 0; b < lnb.y; ++b) {
                        const float ww = wfv[a] * wbv[b];
                        const float4 v = E4[(lnb.x + b) * ow4 + x4];
                        o.x = fmaf(ww, v.x, o.x);
                        o.y = fmaf(ww, v.y, o.y);
                        o.z = fmaf(ww, v.z, o.z);
                        o.w = fmaf(ww, v.w, o.w);
                    }
                } else {
                    const float4 v = E4[ya * ow4 + x4];
                    o.x = fmaf(wfv[a], v.x, o.x);
                    o.y = fmaf(wfv[a], v.y, o.y);
                    o.z = fmaf(wfv[a], v.z, o.z);
                    o.w = fmaf(wfv[a], v.w, o.w);
                }
            }
            store_obs(&out4[q], o);
        }
        return;
    }
    // mask-out paste at (r, c) / raw crop at the origin of the obs-pitched buffer
    const int pr = (p.out_mode == AGX_OUT_MASK) ? r : 0;
    const int pc = (p.out_mode == AGX_OUT_MASK) ? c : 0;
    for (int q = tid; q < oh * ow4; q += kThreads) {
        const int row = q / ow4, x = (q - row * ow4) * 4;
        const int y = row - pr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < rh) {
            int2 lnb = make_int2(0, 0);
            const float *wbv = nullptr;
            if (squeeze) {
                lnb = hb.ln[y];
                wbv = hb.w + y * hb.maxt;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + k - pc;
                if (xx >= 0 && xx < rw) {
                    if (squeeze) {
                        float acc = 0.f;
                        for (int b = 0; b < lnb.y; ++b) acc = fmaf(wbv[b], C[(lnb.x + b) * ow + xx], acc);
                        v[k] = acc;
                    } else {
                        v[k] = lut[win[y * ow + xx]];
                    }
                }
            }
        }
        store_obs(&out4[q], make_float4(v[0], v[1], v[2], v[3]));
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
