// agx_host_tables.h - HOST side: the per-context operator tables of the composed-operator kernels
// (k_fovea_flexible3, agx_k4_flex3.h), built from agx_rows.h.  Included by agx_api.hip and by the CPU test
// harness tests/host_tables_harness.cpp, which replays the kernels' arithmetic from these very tables.
#pragma once
#include <cstring>
#include <vector>

#include "agx.h"
#include "agx_common.h"
#include "agx_k3_per3.h"
#include "agx_k4_flex3.h"
#include "agx_k4_raw3.h"
#include "agx_rows.h"

namespace agx {

// ---- K4 resize_to_full form: composed operators per window size (agx_k4_flex3.h, agx_rows.h)
struct Flex3Host {
    bool ok = false;
    std::vector<Tap> wf;
    std::vector<int2> wc_meta, hd_meta;
    std::vector<int32_t> wc_lo, hd_lo;
    std::vector<float> wc_w, hd_w;
    std::vector<int4> hy;
    int r0_bytes = 0, r1_bytes = 0, dp = 0;
    size_t lds() const { return (size_t)r0_bytes + r1_bytes; }
};

inline int flex3_bucket(int n, int cap) { return n <= 4 ? 4 : (n <= 8 ? 8 : (n <= 16 && cap >= 16 ? 16 : 0)); }

inline Flex3Host build_flex3(const agx_config &c) {
    using namespace agx::rows;
    Flex3Host h;
    const int oh = c.obs_h, ow = c.obs_w, fh = c.fov_h, fw = c.fov_w;
    const bool aa = c.antialias != 0;
    // thread-per-column W passes, 8 lanes per squeezed row, byte-sized row indices
    if (c.out_mode != AGX_OUT_RESIZE || ow > kThreads || fh > kThreads / 8 || oh > 1024) return h;
    const int rstep = kThreads / ow;
    const int erows = (fh + rstep - 1) / rstep * rstep;
    h.dp = (ow + 7) & ~7;
    h.r0_bytes = (int)((std::max((size_t)oh * ow + 8 * (size_t)ow + 32, (size_t)erows * ow * 4) + 15) & ~(size_t)15);
    h.r1_bytes = (int)(((size_t)erows * h.dp * 4 + 15) & ~(size_t)15);
    h.wf.assign((size_t)(ow + 1) * ow, Tap{0, 0, 0.f, 0.f});
    h.wc_meta.assign(ow + 1, make_int2(4, 0));
    h.wc_lo.assign((size_t)(ow + 1) * ow, 0);
    h.hd_meta.assign(oh + 1, make_int2(4, 0));
    h.hd_lo.assign((size_t)(oh + 1) * fh, 0);
    h.hy.assign((size_t)(oh + 1) * oh, make_int4(0, 0, 0, 0));
    for (int rw = 1; rw <= ow; ++rw) {
        const Op fin = resize_axis(rw, ow, aa);
        for (int x = 0; x < ow; ++x) {
            Row r = fin[x];
            trim(r);
            if (r.w.size() > 2) return h;
            const bool two = r.w.size() > 1;
            h.wf[(size_t)rw * ow + x] = Tap{r.lo, two ? r.lo + 1 : r.lo, (float)(r.w[0] / 255.0), two ? (float)(r.w[1] / 255.0) : 0.f};
        }
        // squeeze path: crop -> Resize(fov) -> Resize(res) -> Resize(obs) along W
        Op comp = compose(fin, compose(resize_axis(fw, rw, aa), resize_axis(rw, fw, aa)));
        const int T = flex3_bucket(max_taps(comp), 16);
        if (!T) return h;
        fit(comp, rw, T);
        h.wc_meta[rw] = make_int2(T, (int)h.wc_w.size());
        for (int x = 0; x < ow; ++x) {
            h.wc_lo[(size_t)rw * ow + x] = comp[x].lo;
            for (int q = 0; q < T; ++q) h.wc_w.push_back(q < (int)comp[x].w.size() ? (float)comp[x].w[q] : 0.f);
        }
    }
    for (int rh = 1; rh <= oh; ++rh) {
        const bool squeeze = rh > fh;                                  // rows only, fov_env.py:286
        const int er = squeeze ? fh : rh;                              // rows of E
        Op fin = resize_axis(rh, oh, aa);
        if (squeeze) {
            Op d = resize_axis(rh, fh, aa);
            const int T = flex3_bucket(max_taps(d), 8);
            if (!T) return h;
            fit(d, rh, T);
            h.hd_meta[rh] = make_int2(T, (int)h.hd_w.size());
            for (int y = 0; y < fh; ++y) {
                h.hd_lo[(size_t)rh * fh + y] = d[y].lo;
                for (int q = 0; q < T; ++q) h.hd_w.push_back(q < (int)d[y].w.size() ? (float)(d[y].w[q] / 255.0) : 0.f);
            }
            fin = compose(fin, resize_axis(fh, rh, aa));
        }
        for (int y = 0; y < oh; ++y) {
            Row r = fin[y];
            trim(r);
            if (r.w.size() > 3 || er > 256) return h;
            int idx[3];
            float w[3];
            for (int q = 0; q < 3; ++q) {
                idx[q] = std::min(r.lo + q, er - 1);
                w[q] = q < (int)r.w.size() ? (float)r.w[q] : 0.f;
            }
            int4 e;
            e.x = idx[0] | (idx[1] << 8) | (idx[2] << 16);
            memcpy(&e.y, &w[0], 4);
            memcpy(&e.z, &w[1], 4);
            memcpy(&e.w, &w[2], 4);
            h.hy[(size_t)rh * oh + y] = e;
        }
    }
    if (h.wc_w.empty()) h.wc_w.push_back(0.f);
    if (h.hd_w.empty()) h.hd_w.assign(4, 0.f);
    h.ok = true;
    return h;
}

inline size_t flex3_lds(const Flex3Host &h, const agx_config &c) { return h.lds() + (size_t)c.obs_h * sizeof(int4); }


// ---- K4 raw-crop / mask-out / packed forms (agx_k4_raw3.h): the squeeze-and-expand-back chain of fov_env.py:276-287
// composed per axis: H squeeze (shared shape with Flex3Host), W (Wbck Wdwn)(rw): rw -> rw, H expand back fh -> rh
struct FlexRawHost {
    bool ok = false;
    std::vector<int2> wb_meta, hd_meta;
    std::vector<int32_t> wb_lo, hd_lo;
    std::vector<float> wb_w, hd_w;
    std::vector<Tap> hb;
    int r0_bytes = 0, r1_bytes = 0, dp = 0;
    size_t lds(const agx_config &c) const { return (size_t)r0_bytes + r1_bytes + (size_t)c.obs_h * sizeof(Tap); }
};

inline FlexRawHost build_flexraw(const agx_config &c) {
    using namespace agx::rows;
    FlexRawHost h;
    const int oh = c.obs_h, ow = c.obs_w, fh = c.fov_h, fw = c.fov_w;
    const bool aa = c.antialias != 0;
    if (c.out_mode == AGX_OUT_RESIZE || ow > kThreads || fh > kThreads / 8 || oh > 1024) return h;
    const int rstep = kThreads / ow;
    const int erows = (fh + rstep - 1) / rstep * rstep;
    h.dp = (ow + 7) & ~7;
    h.r0_bytes = (int)((std::max((size_t)oh * ow + 8 * (size_t)ow + 32, (size_t)erows * ow * 4) + 15) & ~(size_t)15);
    h.r1_bytes = (int)(((size_t)erows * h.dp * 4 + 15) & ~(size_t)15);
    h.wb_meta.assign(ow + 1, make_int2(4, 0));
    h.wb_lo.assign((size_t)(ow + 1) * ow, 0);
    h.hd_meta.assign(oh + 1, make_int2(4, 0));
    h.hd_lo.assign((size_t)(oh + 1) * fh, 0);
    h.hb.assign((size_t)(oh + 1) * oh, Tap{0, 0, 0.f, 0.f});
    for (int rw = 1; rw <= ow; ++rw) {
        Op comp = compose(resize_axis(fw, rw, aa), resize_axis(rw, fw, aa));      // rw -> fw -> rw
        const int T = flex3_bucket(max_taps(comp), 16);
        if (!T) return h;
        fit(comp, rw, T);
        h.wb_meta[rw] = make_int2(T, (int)h.wb_w.size());
        for (int x = 0; x < ow; ++x) {
            h.wb_lo[(size_t)rw * ow + x] = x < rw ? comp[x].lo : 0;
            for (int q = 0; q < T; ++q) h.wb_w.push_back(x < rw && q < (int)comp[x].w.size() ? (float)comp[x].w[q] : 0.f);
        }
    }
    for (int rh = fh + 1; rh <= oh; ++rh) {
        Op d = resize_axis(rh, fh, aa);
        const int T = flex3_bucket(max_taps(d), 8);
        if (!T) return h;
        fit(d, rh, T);
        h.hd_meta[rh] = make_int2(T, (int)h.hd_w.size());
        for (int y = 0; y < fh; ++y) {
            h.hd_lo[(size_t)rh * fh + y] = d[y].lo;
            for (int q = 0; q < T; ++q) h.hd_w.push_back(q < (int)d[y].w.size() ? (float)(d[y].w[q] / 255.0) : 0.f);
        }
        const Op back = resize_axis(fh, rh, aa);                                  // an up-scale: one or two taps
        for (int y = 0; y < rh; ++y) {
            Row r = back[y];
            trim(r);
            if (r.w.size() > 2) return h;
            const bool two = r.w.size() > 1;
            h.hb[(size_t)rh * oh + y] = Tap{r.lo, two ? r.lo + 1 : r.lo, (float)r.w[0], two ? (float)r.w[1] : 0.f};
        }
    }
    if (h.wb_w.empty()) h.wb_w.push_back(0.f);
    if (h.hd_w.empty()) h.hd_w.assign(4, 0.f);
    h.ok = true;
    return h;
}

// ---- K3 tuned form 3 (agx_k3_per3.h): the four passes' taps, the squeeze tables padded to one compile-time bound
struct Per3Host {
    bool ok = false;
    int mt = 0;
    std::vector<int32_t> lo0, lo1;
    std::vector<float> w0, w1;
    std::vector<Tap> x2;
    std::vector<int4> y3;
    size_t lds = 0;
};

inline Per3Host build_per3(const agx_config &c) {
    using namespace agx::rows;
    Per3Host h;
    const int oh = c.obs_h, ow = c.obs_w, ph = c.per_h, pw = c.per_w;
    const bool aa = c.antialias != 0;
    if (ow > kThreads || pw > kThreads || ph > 256 || pw < 1 || ph < 1) return h;
    Op s0 = resize_axis(ow, pw, aa), s1 = resize_axis(oh, ph, aa);
    const Op e2 = resize_axis(pw, ow, aa), e3 = resize_axis(ph, oh, aa);
    const int m = std::max(max_taps(s0), max_taps(s1));
    h.mt = m <= 4 ? 4 : (m <= 8 ? 8 : (m <= 12 ? 12 : (m <= 16 ? 16 : 0)));
    if (!h.mt || oh < h.mt || ow < h.mt || max_taps(e2) > 2 || max_taps(e3) > 2) return h;
    fit(s0, ow, h.mt);
    fit(s1, oh, h.mt);
    for (int x = 0; x < pw; ++x) {
        h.lo0.push_back(s0[x].lo);
        for (int q = 0; q < h.mt; ++q) h.w0.push_back(q < (int)s0[x].w.size() ? (float)(s0[x].w[q] / 255.0) : 0.f);
    }
    for (int y = 0; y < ph; ++y) {
        h.lo1.push_back(s1[y].lo);
        for (int q = 0; q < h.mt; ++q) h.w1.push_back(q < (int)s1[y].w.size() ? (float)s1[y].w[q] : 0.f);
    }
    for (int x = 0; x < ow; ++x) {
        Row r = e2[x];
        trim(r);
        const bool two = r.w.size() > 1;
        h.x2.push_back(Tap{r.lo, two ? r.lo + 1 : r.lo, (float)r.w[0], two ? (float)r.w[1] : 0.f});
    }
    for (int y = 0; y < oh; ++y) {
        Row r = e3[y];
        trim(r);
        const bool two = r.w.size() > 1;
        const float w0 = (float)r.w[0], w1 = two ? (float)r.w[1] : 0.f;
        int4 e;
        e.x = r.lo | ((two ? r.lo + 1 : r.lo) << 8);
        memcpy(&e.y, &w0, 4);
        memcpy(&e.z, &w1, 4);
        e.w = 0;
        h.y3.push_back(e);
    }
    // LDS plan: must match the carve in k_fovea_peripheral3
    const int per2 = kThreads / ow, crows = (ph + per2 - 1) / per2 * per2;
    const size_t raw = ((size_t)oh * ow + 16 + 15) & ~(size_t)15;
    const size_t ac = ((size_t)std::max(oh * pw, crows * ow) + 3) & ~(size_t)3;
    const size_t b = ((size_t)crows * pw + 3) & ~(size_t)3;
    h.lds = raw + (ac + b + (size_t)ph * h.mt + (((size_t)ph + 3) & ~(size_t)3)) * 4 + (size_t)oh * sizeof(int4);
    h.ok = true;
    return h;
}

}  // namespace agx
