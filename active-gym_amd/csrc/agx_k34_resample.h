// agx_k34_resample.h - K3 / K4: peripheral squeeze-expand + paste and the flexible (ragged) fovea; generic fallback
// and the tuned host-table forms.
#pragma once
#include "agx_fov_common.h"
#include "agx_k2_fixed.h"

namespace agx {

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
    const FastDiv dv(d.n_out);
    for (int i = tid; i < total; i += kThreads) {
        int y, x;
        dv.divmod(i, y, x);
        dst[i] = apply_tap(d, tab[x], src + y * d.n_in, 1);
    }
}

// dst[n_out][cols] = resample along H of src[n_in][cols]
__device__ __forceinline__ void pass_h(const PassDesc &d, const Tap *tab, const float *src, float *dst,
                                       int cols, int tid) {
    const int total = d.n_out * cols;
    const FastDiv dv(cols);
    for (int i = tid; i < total; i += kThreads) {
        int y, x;
        dv.divmod(i, y, x);
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
    const FastDiv dv_ow4(ow4);
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);

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
            const int row = dv_ow4.div(q), x = (q - row * ow4) * 4;
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
            store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
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
            const int row = dv_ow4.div(q), x4 = q - row * ow4;
            const Tap t = tab[row];
            const float4 a = H4[t.lo * ow4 + x4];
            const float4 b = H4[t.aux * ow4 + x4];
            store_obs(oout, q, make_float4(t.a * a.x + t.b * b.x, t.a * a.y + t.b * b.y,
                                           t.a * a.z + t.b * b.z, t.a * a.w + t.b * b.w));
        }
        return;
    }
    if (p.packed) {                                             // raw crops, packed: [j][rh][rw] tight
        const int64_t off = p.packed_off[n], cnt = (int64_t)rh * rw;
        if (off + (int64_t)p.fs * cnt > p.packed_cap) return;
        float *dst = p.packed + off + (int64_t)j * cnt;
        for (int i = tid; i < rh * rw; i += kThreads) dst[i] = cur[i];
        return;
    }
    // mask-out paste at (r, c); raw (padded, window at the origin); resize with res == obs (identity)
    const int pr = (p.out_mode == AGX_OUT_MASK) ? r : 0;
    const int pc = (p.out_mode == AGX_OUT_MASK) ? c : 0;
    for (int q = tid; q < oh * ow4; q += kThreads) {
        const int row = dv_ow4.div(q), x = (q - row * ow4) * 4;
        const int y = row - pr;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < rh) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + k - pc;
                if (xx >= 0 && xx < rw) v[k] = cur[y * rw + xx];
            }
        }
        store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// ---------------------------------------------------------------------------------------------
// K3, tuned form: FixedFovealPeripheralEnv with the context's fixed geometry.
// grid = (fs, N): workgroup (sl, n) owns physical ring slot sl; block = 256.
// The four separable passes use tap tables built on the HOST at agx_create (ATen's arithmetic in
// double, weights normalised, narrowed to f32): per output index {lo, n} and n weights.
//   raw u8 frame --W squeeze--> A[oh][pw] --H squeeze--> B[ph][pw] --W expand--> C[ph][ow]
//   --H expand, fused with the full-resolution fovea paste and the written-through (sc1) store.
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
    const FastDiv dv_ow4(ow4), dv_pw(pw), dv_ow(ow);
    float4 *out4 = reinterpret_cast<float4 *>(p.obs) + ((size_t)n * p.fs + j) * (size_t)(oh * ow4);
    const ObsOut oout = obs_out(out4, oh * ow4);
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
            const int yp = dv_pw.div(i), xp = i - yp * pw;
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
                const int yp = dv_ow.div(i), x = i - yp * ow;
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
        const int row = dv_ow4.div(q), x4 = q - row * ow4, x = x4 * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in_r = row >= r && row < r + fh;
        const bool all_fov = g.same || (in_r && x >= c && x + 3 < c + fw);
        if (!all_fov) {
            const int2 ln = ln3_s[row];
            const float *w = w3_s + row * mt3;
            if (mt3 <= 2) {
                // the expansion is an up-scale (peripheral_res <= obs): one or two taps; unrolled with a zero second
                // weight (the table is zero-padded to mt3), no per-lane trip count
                const float w0 = w[0], w1 = mt3 > 1 ? w[1] : 0.f;
                const float4 v0 = C4[ln.x * ow4 + x4];
                const float4 v1 = C4[(ln.y > 1 ? ln.x + 1 : ln.x) * ow4 + x4];
                o.x = fmaf(w1, v1.x, fmaf(w0, v0.x, o.x));
                o.y = fmaf(w1, v1.y, fmaf(w0, v0.y, o.y));
                o.z = fmaf(w1, v1.z, fmaf(w0, v0.z, o.z));
                o.w = fmaf(w1, v1.w, fmaf(w0, v0.w, o.w));
            } else {
                for (int k = 0; k < ln.y; ++k) {
                    const float4 v = C4[(ln.x + k) * ow4 + x4];
                    o.x = fmaf(w[k], v.x, o.x);
                    o.y = fmaf(w[k], v.y, o.y);
                    o.z = fmaf(w[k], v.z, o.z);
                    o.w = fmaf(w[k], v.w, o.w);
                }
            }
        }
        if (g.same || (in_r && x + 3 >= c && x < c + fw)) {
            const uint32_t wv = *reinterpret_cast<const uint32_t *>(raw + row * ow + x);
            if (g.same || (x >= c && x < c + fw)) o.x = lut[wv & 0xFF];
            if (g.same || (x + 1 >= c && x + 1 < c + fw)) o.y = lut[(wv >> 8) & 0xFF];
            if (g.same || (x + 2 >= c && x + 2 < c + fw)) o.z = lut[(wv >> 16) & 0xFF];
            if (g.same || (x + 3 >= c && x + 3 < c + fw)) o.w = lut[wv >> 24];
        }
        store_obs(oout, q, o);
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

// RESIZE: compile-time out_mode == AGX_OUT_RESIZE (two instantiations: each keeps only the table families it reads in SGPRs)
template <bool RESIZE>
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
    // LDS: lut[256] | raw[oh*ow] aliased by C[fh*ow] | AE[max(oh*fw, fh*ow)] | B[fh*fw] | tables
    // (C is written by P3, two barriers after P1's last read of the raw bytes; only the squeeze path has a C, and
    //  it never looks at the raw frame again - the paste / crop outputs read C there)
    float *lut = reinterpret_cast<float *>(smem);
    unsigned char *raw = smem + 1024;
    float *C = reinterpret_cast<float *>(raw);
    const int rc_bytes = max((fbytes + 15) & ~15, ((fh * ow + 3) & ~3) * 4);
    float *AE = reinterpret_cast<float *>(raw + rc_bytes);
    const int ae_floats = (max(oh * fw, fh * ow) + 3) & ~3;
    float *B = AE + ae_floats;
    float *tabs = B + ((fh * fw + 3) & ~3);

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
    constexpr bool resize = RESIZE;
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
    const ObsOut oout = obs_out(out4, oh * ow4);

    if (squeeze) {
        // P1: A[y][xf] = Wdwn(crop)      y < rh, xf < fw
        const FastDiv dv_fw(fw);
        for (int i = tid; i < rh * fw; i += kThreads) {
            const int y = dv_fw.div(i), xf = i - y * fw;
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
            const int yf = dv_fw.div(i), xf = i - yf * fw;
            B[i] = tap_dot(hd, yf, AE + xf, fw);
        }
        __syncthreads();
        // P3: C[yf][x] = Wbck(B)         x < rw
        const FastDiv dv_rw(rw);
        for (int i = tid; i < fh * rw; i += kThreads) {
            const int yf = dv_rw.div(i), x = i - yf * rw;
            C[yf * ow + x] = tap_dot(wb, x, B + yf * fw, 1);
        }
        __syncthreads();
    }

    if (resize) {
        // E[y][xo] = Wfin(src rows): src = C (fh rows) after a squeeze, else the crop itself (rh rows)
        const int erows = squeeze ? fh : rh;
        const FastDiv dv_ow(ow), dv_ow4(ow4);
        for (int i = tid; i < erows * ow; i += kThreads) {
            const int y = dv_ow.div(i), xo = i - y * ow;
            // the final resize is never a down-scale (r <= obs): one or two taps, unrolled with a zero second weight
            const int2 ln = wf.ln[xo];
            const float *w = wf.w + xo * wf.maxt;
            const bool two = ln.y > 1;
            const float w0 = w[0], w1 = two ? w[1] : 0.f;
            const int i1 = two ? ln.x + 1 : ln.x;
            float acc;
            if (squeeze) {
                acc = fmaf(w1, C[y * ow + i1], w0 * C[y * ow + ln.x]);
            } else {
                const unsigned char *src = win + y * ow;
                acc = fmaf(w1, (float)src[i1], w0 * (float)src[ln.x]) * kInv255;
            }
            AE[i] = acc;
        }
        __syncthreads();
        const float4 *E4 = reinterpret_cast<const float4 *>(AE);
        for (int q = tid; q < oh * ow4; q += kThreads) {
            const int row = dv_ow4.div(q), x4 = q - row * ow4;
            // both H passes of this output are up-scales here (r <= obs; fov < r on the squeeze path): at most two
            // taps each, unrolled with zero weights for a missing second tap - no per-lane trip counts
            const int2 lnf = hf.ln[row];
            const float *wfv = hf.w + row * hf.maxt;
            const bool twof = lnf.y > 1;
            const float fa[2] = {wfv[0], twof ? wfv[1] : 0.f};
            const int ya[2] = {lnf.x, twof ? lnf.x + 1 : lnf.x};          // rows of the (virtual) rh-row image
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (squeeze) {
                    const int2 lnb = hb.ln[ya[a]];
                    const float *wbv = hb.w + ya[a] * hb.maxt;
                    const bool twob = lnb.y > 1;
                    const float wb0 = fa[a] * wbv[0], wb1 = twob ? fa[a] * wbv[1] : 0.f;
                    const float4 v0 = E4[lnb.x * ow4 + x4];
                    const float4 v1 = E4[(twob ? lnb.x + 1 : lnb.x) * ow4 + x4];
                    o.x = fmaf(wb1, v1.x, fmaf(wb0, v0.x, o.x));
                    o.y = fmaf(wb1, v1.y, fmaf(wb0, v0.y, o.y));
                    o.z = fmaf(wb1, v1.z, fmaf(wb0, v0.z, o.z));
                    o.w = fmaf(wb1, v1.w, fmaf(wb0, v0.w, o.w));
                } else {
                    const float4 v = E4[ya[a] * ow4 + x4];
                    o.x = fmaf(fa[a], v.x, o.x);
                    o.y = fmaf(fa[a], v.y, o.y);
                    o.z = fmaf(fa[a], v.z, o.z);
                    o.w = fmaf(fa[a], v.w, o.w);
                }
            }
            store_obs(oout, q, o);
        }
        return;
    }
    if (p.packed) {                                             // raw crops, packed: [j][rh][rw] tight
        const int64_t off = p.packed_off[n], cnt = (int64_t)rh * rw;
        if (off + (int64_t)p.fs * cnt > p.packed_cap) return;
        float *dst = p.packed + off + (int64_t)j * cnt;
        const FastDiv dv_rw(rw);
        for (int i = tid; i < rh * rw; i += kThreads) {
            const int y = dv_rw.div(i), x = i - y * rw;
            float v;
            if (squeeze) {
                const int2 lnb = hb.ln[y];
                const float *wbv = hb.w + y * hb.maxt;
                v = 0.f;
                for (int b = 0; b < lnb.y; ++b) v = fmaf(wbv[b], C[(lnb.x + b) * ow + x], v);
            } else {
                v = lut[win[y * ow + x]];
            }
            dst[i] = v;
        }
        return;
    }
    // mask-out paste at (r, c) / raw crop at the origin of the obs-pitched buffer
    const int pr = (p.out_mode == AGX_OUT_MASK) ? r : 0;
    const int pc = (p.out_mode == AGX_OUT_MASK) ? c : 0;
    const FastDiv dv_ow4(ow4);
    for (int q = tid; q < oh * ow4; q += kThreads) {
        const int row = dv_ow4.div(q), x = (q - row * ow4) * 4;
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
        store_obs(oout, q, make_float4(v[0], v[1], v[2], v[3]));
    }
}

}  // namespace agx
