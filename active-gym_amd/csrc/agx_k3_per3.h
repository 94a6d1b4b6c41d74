// agx_k3_per3.h - K3, tuned form 3: FixedFovealPeripheralEnv._get_fov_state (fov_env.py:375-388):
//   Sequential(Resize(peripheral_res), Resize(obs_size)) on the whole stack, the full-resolution fovea pasted.
//
// grid = (fs, N): workgroup (sl, n) owns PHYSICAL ring slot sl of env n; block = 256.  Same four separable passes as
// k_fovea_peripheral2 (agx_k34_resample.h) - the W squeeze reads the u8 frame through aligned dwords, taps from
// agx_create in registers / LDS - but with everything that depends on the env (fov_loc, ring head) in SGPRs, compile-
// time geometry for the 84x84 / 30x30 / 20x20 headline (divisions fold, loops unroll to fixed trip counts), a fixed
// thread role per pass and no per-lane trip counts:
//   raw u8 --W squeeze--> A[oh][pw] --H squeeze--> B[ph][pw] --W expand--> C[ph][ow] --H expand + paste--> out
// The pasted fovea pixels are the exact float32 k/255 of the reference (unit_fast: 3 FMAs, correctly rounded);
// the periphery carries the 1/255 in the W-squeeze weights.  LDS 18 KB for the headline -> 8 workgroups per CU.
#pragma once
#include "agx_fov_common.h"
#include "agx_k2_fixed.h"

namespace agx {

template <int OH, int OW, int FH, int FW, int PH, int PW>
struct PGeomS {
    __host__ __device__ constexpr int oh() const { return OH; }
    __host__ __device__ constexpr int ow() const { return OW; }
    __host__ __device__ constexpr int fh() const { return FH; }
    __host__ __device__ constexpr int fw() const { return FW; }
    __host__ __device__ constexpr int ph() const { return PH; }
    __host__ __device__ constexpr int pw() const { return PW; }
};
struct PGeomR {
    int oh_, ow_, fh_, fw_, ph_, pw_;
    __host__ __device__ int oh() const { return oh_; }
    __host__ __device__ int ow() const { return ow_; }
    __host__ __device__ int fh() const { return fh_; }
    __host__ __device__ int fw() const { return fw_; }
    __host__ __device__ int ph() const { return ph_; }
    __host__ __device__ int pw() const { return pw_; }
};

struct Per3Params {
    const int32_t *lo0;   // [pw]        first frame column of the W squeeze
    const float *w0;      // [pw][MT]    its weights / 255 (they multiply u8 numerators), zero-padded
    const int32_t *lo1;   // [ph]        first A row of the H squeeze (lo + MT - 1 <= oh - 1)
    const float *w1;      // [ph][MT]
    const Tap *x2;        // [ow]        W expand {lo, i1, wa, wb} over the columns of B
    const int4 *y3;       // [oh]        H expand {i0 | i1 << 8, w0, w1, -} over the rows of C
    int32_t same;         // peripheral_res == obs_size: torchvision returns the input unchanged
};

template <class G, int MT>
__global__ __launch_bounds__(kThreads) void k_fovea_peripheral3(G g, Per3Params t, FovParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sl = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int oh = g.oh(), ow = g.ow(), fh = g.fh(), fw = g.fw(), ph = g.ph(), pw = g.pw();
    if (p.mask && !p.mask[n]) {
        if (sl == 0 && tid < 2) p.loc_out[2 * n + tid] = p.loc_in[2 * n + tid];
        return;
    }
    const int fbytes = oh * ow, fwords = fbytes >> 2;
    const int per2 = kThreads / ow;                                    // rows per sweep of the W expand (3)
    const int crows = (ph + per2 - 1) / per2 * per2;                   // C / B rows incl. the sweep's overhang
    // LDS: raw[oh*ow] (+16 B the W squeeze may read past the end) | A[oh][pw] aliased by C[crows][ow] | B[crows][pw]
    //      | w1[ph][MT] | lo1[ph] | y3[oh]
    unsigned char *raw = smem;
    float *A = reinterpret_cast<float *>(smem + ((fbytes + 16 + 15) & ~15));
    float *C = A;
    float *B = A + ((max(oh * pw, crows * ow) + 3) & ~3);
    float *w1_s = B + ((crows * pw + 3) & ~3);
    int32_t *lo1_s = reinterpret_cast<int32_t *>(w1_s + ph * MT);
    int4 *y3_s = reinterpret_cast<int4 *>(lo1_s + ((ph + 3) & ~3));

    // ---- every round trip starts now: state, the frame, this thread's taps
    int head;
    const LocIn lin = load_loc_inputs_scalar(p, n, p.head, head);
    const uint32_t *fsrc = reinterpret_cast<const uint32_t *>(p.ring + ((size_t)n * p.fs + sl) * (size_t)fbytes);
    constexpr int kFW = 7;
    uint32_t fw_[kFW];
#pragma unroll
    for (int k = 0; k < kFW; ++k) fw_[k] = fsrc[min(tid + k * kThreads, fwords - 1)];
    const int per0 = kThreads / pw;                                    // rows per sweep of the W squeeze (12)
    const int xp0 = tid % pw, y00 = tid / pw;
    const int lo0 = t.lo0[xp0];
    float wr0[MT];
#pragma unroll
    for (int k = 0; k < MT; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(t.w0 + xp0 * MT + k);
        wr0[k] = v.x, wr0[k + 1] = v.y, wr0[k + 2] = v.z, wr0[k + 3] = v.w;
    }
    const int x2 = tid % ow, y20 = tid / ow;
    const int4 xt = *reinterpret_cast<const int4 *>(t.x2 + x2);
    const int4 yt = t.y3[min(tid, oh - 1)];
    const float w1v = t.w1[min(tid, ph * MT - 1)];
    const int lo1v = t.lo1[min(tid, ph - 1)];

    int r, c;
    compute_loc(p, lin, oh - fh, ow - fw, r, c);
    r = __builtin_amdgcn_readfirstlane(r);
    c = __builtin_amdgcn_readfirstlane(c);
    int j = sl - __builtin_amdgcn_readfirstlane(head);
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
    if (tid < 4) reinterpret_cast<uint32_t *>(raw)[fwords + tid] = 0u;  // the 16 B past the frame: zero weights, finite bytes
    if (tid < ph * MT) w1_s[tid] = w1v;
    for (int i = tid + kThreads; i < ph * MT; i += kThreads) w1_s[i] = t.w1[i];
    if (tid < ph) lo1_s[tid] = lo1v;
    for (int i = tid + kThreads; i < ph; i += kThreads) lo1_s[i] = t.lo1[i];
    if (tid < oh) y3_s[tid] = yt;
    for (int i = tid + kThreads; i < oh; i += kThreads) y3_s[i] = t.y3[i];
    __syncthreads();

    const int ow4 = ow >> 2;
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);
    if (!t.same) {
        // ---- pass 0: A[y][xp] = sum_k (w0[xp][k] / 255) * raw[y][lo + k]; bytes through aligned dwords + alignbyte
        if (y00 < per0) {
            constexpr int NDW = (MT + 6) / 4;                           // aligned dwords covering (lo & 3) + MT bytes
            for (int y = y00; y < oh; y += per0) {
                const int off = y * ow + lo0;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(raw + (off & ~3));
                const uint32_t sh = off & 3;
                uint32_t d[NDW];
#pragma unroll
                for (int k = 0; k < NDW; ++k) d[k] = src[k];
                float acc = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < MT / 4; ++q4) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(d[q4 + 1], d[q4], sh);
                    acc = fmaf(wr0[4 * q4 + 0], (float)(v & 0xFF), acc);
                    acc = fmaf(wr0[4 * q4 + 1], (float)((v >> 8) & 0xFF), acc);
                    acc = fmaf(wr0[4 * q4 + 2], (float)((v >> 16) & 0xFF), acc);
                    acc = fmaf(wr0[4 * q4 + 3], (float)(v >> 24), acc);
                }
                A[y * pw + xp0] = acc;
            }
        }
        __syncthreads();
        // ---- pass 1: B[yp][xp] = sum_k w1[yp][k] * A[lo + k][xp]
        for (int i = tid; i < ph * pw; i += kThreads) {
            const int yp = i / pw, xp = i - yp * pw;
            const float *a = A + lo1_s[yp] * pw + xp;
            const float4 *w = reinterpret_cast<const float4 *>(w1_s + yp * MT);
            float v[MT];
#pragma unroll
            for (int k = 0; k < MT; ++k) v[k] = a[k * pw];
            float acc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < MT / 4; ++k4) {
                const float4 ww = w[k4];
                acc = fmaf(ww.x, v[4 * k4], acc);
                acc = fmaf(ww.y, v[4 * k4 + 1], acc);
                acc = fmaf(ww.z, v[4 * k4 + 2], acc);
                acc = fmaf(ww.w, v[4 * k4 + 3], acc);
            }
            B[i] = acc;
        }
        __syncthreads();
        // ---- pass 2: C[yp][x] = wa * B[yp][lo] + wb * B[yp][i1]   (row ph .. crows-1: overhang, never read back)
        if (y20 < per2) {
            const float wa = __int_as_float(xt.z), wb = __int_as_float(xt.w);
            const float *b0 = B + y20 * pw + xt.x, *b1 = B + y20 * pw + xt.y;
            float *dst = C + y20 * ow + x2;
            const int kmax = crows / per2;
#pragma unroll 7
            for (int k = 0; k < kmax; ++k) dst[k * per2 * ow] = fmaf(wb, b1[k * per2 * pw], wa * b0[k * per2 * pw]);
        }
        __syncthreads();
    }
    // ---- pass 3: H expand, the fovea pasted at full resolution, 16-B written-through (sc1) stores
    const float4 *C4 = reinterpret_cast<const float4 *>(C);
    const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(raw);
#pragma unroll 7
    for (int k_ = 0; k_ < (oh * ow4 + kThreads - 1) / kThreads; ++k_) {
        const int q = tid + k_ * kThreads;
        if (q >= oh * ow4) break;
        const int row = q / ow4, x4 = q - row * ow4, x = x4 * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!t.same) {
            const int4 e = y3_s[row];
            const float w0 = __int_as_float(e.y), w1 = __int_as_float(e.z);
            const float4 a = C4[(e.x & 0xFF) * ow4 + x4];
            const float4 b = C4[(e.x >> 8) * ow4 + x4];
            o.x = fmaf(w1, b.x, w0 * a.x);
            o.y = fmaf(w1, b.y, w0 * a.y);
            o.z = fmaf(w1, b.z, w0 * a.z);
            o.w = fmaf(w1, b.w, w0 * a.w);
        }
        const bool paste = t.same || ((unsigned)(row - r) < (unsigned)fh && x + 3 >= c && x < c + fw);
        if (paste) {
            const uint32_t wv = raw32[q];
            const unsigned d0 = (unsigned)(x - c);                      // column inside the window (wraps when left of it)
            if (t.same || d0 < (unsigned)fw) o.x = unit_fast((float)(wv & 0xFF));
            if (t.same || d0 + 1 < (unsigned)fw) o.y = unit_fast((float)((wv >> 8) & 0xFF));
            if (t.same || d0 + 2 < (unsigned)fw) o.z = unit_fast((float)((wv >> 16) & 0xFF));
            if (t.same || d0 + 3 < (unsigned)fw) o.w = unit_fast((float)(wv >> 24));
        }
        store_obs(oout, q, o);
    }
}

}  // namespace agx
