// CPU harness for the host-built operator tables of k_fovea_flexible3 (active-gym_amd/csrc/agx_host_tables.h).
// Replays the kernel's arithmetic (same tables, same float32 operation order, same index rules) on the host and
// compares it with the reference chain  crop -> [Resize(fov) -> Resize(res) iff rh > fov_h] -> Resize(obs)
// (fov_env.py:276-298) evaluated pass by pass in double with the plain per-axis operators.  Also checks the memory
// rules the kernel relies on (every LDS index inside the regions agx_create sizes).
// Build: hipcc -std=c++17 -I include -I active-gym_amd/csrc tests/host_tables_harness.cpp -o <out>   (host code only)
// Usage: harness oh ow fh fw antialias          (k_fovea_flexible3 tables) -> "max_err <e> cases <n>" | "unsupported"
//        harness per oh ow ph pw antialias      (k_fovea_peripheral3 tables + unit_fast over all 256 values)
//        harness raw oh ow fh fw antialias      (k_fovea_flexible_raw3 tables: raw-crop / mask-out / packed forms)
#include <cstdio>
#include <cstdlib>
#include <random>

#include "agx_host_tables.h"

using namespace agx;
using namespace agx::rows;

static std::vector<double> apply_w(const Op &op, const std::vector<double> &x, int rows_, int n_in) {
    std::vector<double> y((size_t)rows_ * op.size());
    for (int r = 0; r < rows_; ++r)
        for (size_t i = 0; i < op.size(); ++i) {
            double a = 0;
            for (size_t k = 0; k < op[i].w.size(); ++k) a += op[i].w[k] * x[(size_t)r * n_in + op[i].lo + k];
            y[(size_t)r * op.size() + i] = a;
        }
    return y;
}
static std::vector<double> apply_h(const Op &op, const std::vector<double> &x, int cols, int n_in) {
    (void)n_in;
    std::vector<double> y(op.size() * (size_t)cols);
    for (size_t i = 0; i < op.size(); ++i)
        for (int c = 0; c < cols; ++c) {
            double a = 0;
            for (size_t k = 0; k < op[i].w.size(); ++k) a += op[i].w[k] * x[(size_t)(op[i].lo + k) * cols + c];
            y[i * cols + c] = a;
        }
    return y;
}
// torchvision Resize(size)(x[h][w]) in double: W pass then H pass (identity when the size matches)
static std::vector<double> resize(const std::vector<double> &x, int h, int w, int nh, int nw, bool aa) {
    if (h == nh && w == nw) return x;
    return apply_h(resize_axis(h, nh, aa), apply_w(resize_axis(w, nw, aa), x, h, w), nw, h);
}

static int per_main(int argc, char **argv) {
    if (argc < 7) return 2;
    for (int k = 0; k < 256; ++k)
        if (unit_fast((float)k) != (float)k / 255.0f) { printf("unit_fast(%d) is not the IEEE quotient\n", k); return 1; }
    agx_config c{};
    c.obs_h = atoi(argv[2]); c.obs_w = atoi(argv[3]); c.per_h = atoi(argv[4]); c.per_w = atoi(argv[5]);
    c.antialias = atoi(argv[6]);
    const Per3Host h = build_per3(c);
    if (!h.ok) { printf("unsupported\n"); return 0; }
    const int oh = c.obs_h, ow = c.obs_w, ph = c.per_h, pw = c.per_w, MT = h.mt;
    const bool aa = c.antialias != 0;
    std::mt19937 rng(11);
    std::vector<unsigned char> raw((size_t)oh * ow + 16, 0);
    for (size_t i = 0; i < (size_t)oh * ow; ++i) raw[i] = (unsigned char)(rng() & 0xFF);
    std::vector<float> A((size_t)oh * pw), B((size_t)ph * pw), C((size_t)ph * ow);
    for (int y = 0; y < oh; ++y)
        for (int x = 0; x < pw; ++x) {
            const int off = y * ow + h.lo0[x];
            if (((off & ~3) + 4 * ((MT + 6) / 4)) > oh * ow + 16) { printf("pass-0 read past the pad\n"); return 1; }
            float acc = 0.f;
            for (int q = 0; q < MT; ++q) acc = fmaf(h.w0[(size_t)x * MT + q], (float)raw[off + q], acc);
            A[(size_t)y * pw + x] = acc;
        }
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x) {
            if (h.lo1[y] + MT > oh) { printf("pass-1 row past A\n"); return 1; }
            float acc = 0.f;
            for (int q = 0; q < MT; ++q) acc = fmaf(h.w1[(size_t)y * MT + q], A[(size_t)(h.lo1[y] + q) * pw + x], acc);
            B[(size_t)y * pw + x] = acc;
        }
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < ow; ++x) {
            const Tap t = h.x2[x];
            if (t.lo >= pw || t.aux >= pw) { printf("pass-2 column\n"); return 1; }
            C[(size_t)y * ow + x] = fmaf(t.b, B[(size_t)y * pw + t.aux], t.a * B[(size_t)y * pw + t.lo]);
        }
    std::vector<double> x((size_t)oh * ow);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (double)raw[i] / 255.0;
    const std::vector<double> ref = resize(resize(x, oh, ow, ph, pw, aa), ph, pw, oh, ow, aa);
    double worst = 0;
    for (int y = 0; y < oh; ++y) {
        const int4 e = h.y3[y];
        const int i0 = e.x & 0xFF, i1 = e.x >> 8;
        if (i0 >= ph || i1 >= ph) { printf("pass-3 row\n"); return 1; }
        float w0, w1;
        memcpy(&w0, &e.y, 4); memcpy(&w1, &e.z, 4);
        for (int xx = 0; xx < ow; ++xx) {
            const float o = fmaf(w1, C[(size_t)i1 * ow + xx], w0 * C[(size_t)i0 * ow + xx]);
            const double err = std::fabs((double)o - ref[(size_t)y * ow + xx]);
            if (!(err <= worst)) worst = err;
        }
    }
    printf("max_err %.3e mt %d lds %zu\n", worst, MT, h.lds);
    return 0;
}


// k_fovea_flexible_raw3 (agx_k4_raw3.h): the squeeze-and-expand-back chain of the raw-crop / mask-out / packed forms.
// Replays the kernel on its LDS image of the window (rows [r, r + rh + 8) clipped to the frame, of each row the dword-aligned
// column span holding [c, c + rw), a dword past the frame's end clamped to its last dword) with stale-LDS NaNs everywhere
// else, and compares with  crop -> Resize(fov_size) -> Resize(fov_res)  (fov_env.py:276-287) in double.
static int raw_main(int argc, char **argv) {
    if (argc < 7) return 2;
    agx_config c{};
    c.obs_h = atoi(argv[2]); c.obs_w = atoi(argv[3]); c.fov_h = atoi(argv[4]); c.fov_w = atoi(argv[5]);
    c.antialias = atoi(argv[6]);
    c.out_mode = AGX_OUT_RAW;
    const FlexRawHost h = build_flexraw(c);
    if (!h.ok) { printf("unsupported\n"); return 0; }
    const int oh = c.obs_h, ow = c.obs_w, fh = c.fov_h, fw = c.fov_w;
    const bool aa = c.antialias != 0;
    const int rstep = kThreads / ow, erows = (fh + rstep - 1) / rstep * rstep;
    std::mt19937 rng(11);
    double worst = 0;
    long cases = 0;
    std::vector<unsigned char> R0(h.r0_bytes);
    std::vector<float> D((size_t)h.r1_bytes / 4), E((size_t)h.r0_bytes / 4);
    for (int rh = fh + 1; rh <= oh; ++rh)
        for (int rw = 1 + (rh * 5) % 3; rw <= ow; rw += 3) {
            const int r = (int)(rng() % (unsigned)(oh - rh + 1)), cc = (int)(rng() % (unsigned)(ow - rw + 1));
            std::vector<unsigned char> F((size_t)oh * ow);
            for (auto &b : F) b = (unsigned char)(rng() & 0xFF);
            for (auto &b : R0) b = (unsigned char)(rng() & 0xFF);
            for (auto &v : D) v = NAN;
            for (auto &v : E) v = NAN;
            const int wrows = std::min(rh + 8, oh - r), span = ((cc & 3) + rw + 3) >> 2, wp = span * 4;
            if ((size_t)wrows * wp > (size_t)h.r0_bytes || wp > ow) { printf("image too large rh=%d rw=%d\n", rh, rw); return 1; }
            for (int i = 0; i < wrows * span; ++i) {
                const int y = i / span, q = i - y * span;
                const int src = std::min((r + y) * (ow / 4) + (cc >> 2) + q, oh * ow / 4 - 1);
                memcpy(&R0[(size_t)i * 4], &F[(size_t)src * 4], 4);
            }
            const unsigned char *win = R0.data() + (cc & 3);
            auto in_r0 = [&](const unsigned char *p) { return p >= R0.data() && p < R0.data() + h.r0_bytes; };
            const int Tw = h.wb_meta[rw].x, Th = h.hd_meta[rh].x;
            const int kmax = (std::max(rw, Tw) + 7) >> 3;
            if (8 * kmax > h.dp) { printf("D pitch too small rw=%d\n", rw); return 1; }
            for (int yf = 0; yf < fh; ++yf)
                for (int x = 0; x < 8 * kmax; ++x) {
                    const int lo = h.hd_lo[(size_t)rh * fh + yf];
                    const float *w = &h.hd_w[h.hd_meta[rh].y + (size_t)yf * Th];
                    float acc = 0.f;
                    for (int q = 0; q < Th; ++q) {
                        const unsigned char *p = win + (lo + q) * wp + x;
                        if (!in_r0(p)) { printf("raw read outside R0 rh=%d rw=%d\n", rh, rw); return 1; }
                        acc = fmaf(w[q], (float)*p, acc);
                    }
                    D[(size_t)yf * h.dp + x] = acc;
                }
            if (erows * ow * 4 > h.r0_bytes) { printf("E does not fit R0\n"); return 1; }
            for (int y = 0; y < fh; ++y)
                for (int x = 0; x < ow; ++x) {
                    const int lo = h.wb_lo[(size_t)rw * ow + x];
                    const float *w = &h.wb_w[h.wb_meta[rw].y + (size_t)x * Tw];
                    if (lo < 0 || lo + Tw > 8 * kmax) { printf("D read outside the written columns rw=%d x=%d\n", rw, x); return 1; }
                    float acc = 0.f;
                    for (int q = 0; q < Tw; ++q) acc = fmaf(w[q], D[(size_t)y * h.dp + lo + q], acc);
                    E[(size_t)y * ow + x] = acc;
                }
            std::vector<double> crop((size_t)rh * rw);
            for (int y = 0; y < rh; ++y)
                for (int x = 0; x < rw; ++x) {
                    if (win[y * wp + x] != F[(size_t)(r + y) * ow + cc + x]) { printf("image != frame window rh=%d rw=%d\n", rh, rw); return 1; }
                    crop[(size_t)y * rw + x] = (double)F[(size_t)(r + y) * ow + cc + x] / 255.0;
                }
            const std::vector<double> ref = resize(resize(crop, rh, rw, fh, fw, aa), fh, fw, rh, rw, aa);
            for (int y = 0; y < rh; ++y) {
                const Tap tp = h.hb[(size_t)rh * oh + y];
                if (tp.lo < 0 || tp.aux >= fh) { printf("row tap rh=%d y=%d\n", rh, y); return 1; }
                for (int x = 0; x < rw; ++x) {
                    const float o = fmaf(tp.b, E[(size_t)tp.aux * ow + x], tp.a * E[(size_t)tp.lo * ow + x]);
                    const double err = std::fabs((double)o - ref[(size_t)y * rw + x]);
                    if (!(err <= worst)) worst = err;      // NaN-propagating max
                }
            }
            ++cases;
        }
    printf("max_err %.3e cases %ld lds %zu\n", worst, cases, h.lds(c));
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "per")) return per_main(argc, argv);
    if (argc > 1 && !strcmp(argv[1], "raw")) return raw_main(argc, argv);
    if (argc < 6) return 2;
    agx_config c{};
    c.obs_h = atoi(argv[1]); c.obs_w = atoi(argv[2]); c.fov_h = atoi(argv[3]); c.fov_w = atoi(argv[4]);
    c.antialias = atoi(argv[5]);
    c.out_mode = AGX_OUT_RESIZE;
    const Flex3Host h = build_flex3(c);
    if (!h.ok) { printf("unsupported\n"); return 0; }
    const int oh = c.obs_h, ow = c.obs_w, fh = c.fov_h, fw = c.fov_w;
    const bool aa = c.antialias != 0;
    const int rstep = kThreads / ow, erows = (fh + rstep - 1) / rstep * rstep;
    std::mt19937 rng(7);
    double worst = 0;
    long cases = 0;
    std::vector<unsigned char> R0(h.r0_bytes);
    std::vector<float> D((size_t)h.r1_bytes / 4), E((size_t)std::max(h.r0_bytes, h.r1_bytes) / 4);
    for (int rh = 1; rh <= oh; ++rh)
        for (int rw = 1 + (rh * 7) % 3; rw <= ow; rw += 3) {
            const int r = (int)(rng() % (unsigned)(oh - rh + 1)), cc = (int)(rng() % (unsigned)(ow - rw + 1));
            for (auto &b : R0) b = (unsigned char)(rng() & 0xFF);          // frame + slack: arbitrary bytes
            for (auto &v : D) v = NAN;                                    // stale LDS may hold anything
            for (auto &v : E) v = NAN;
            const unsigned char *win = R0.data() + r * ow + cc;
            const bool squeeze = rh > fh;
            const int er = squeeze ? fh : rh;
            auto in_r0 = [&](const unsigned char *p) { return p >= R0.data() && p < R0.data() + h.r0_bytes; };
            if (squeeze) {
                const int Tw = h.wc_meta[rw].x, Th = h.hd_meta[rh].x;
                const int kmax = (std::max(rw, Tw) + 7) >> 3;
                if (8 * kmax > h.dp) { printf("D pitch too small rw=%d\n", rw); return 1; }
                for (int yf = 0; yf < fh; ++yf)
                    for (int x = 0; x < 8 * kmax; ++x) {
                        const int lo = h.hd_lo[(size_t)rh * fh + yf];
                        const float *w = &h.hd_w[h.hd_meta[rh].y + (size_t)yf * Th];
                        float acc = 0.f;
                        for (int q = 0; q < Th; ++q) {
                            const unsigned char *p = win + (lo + q) * ow + x;
                            if (!in_r0(p)) { printf("raw read outside R0 rh=%d rw=%d\n", rh, rw); return 1; }
                            acc = fmaf(w[q], (float)*p, acc);
                        }
                        D[(size_t)yf * h.dp + x] = acc;
                    }
                for (int y = 0; y < fh; ++y)
                    for (int x = 0; x < ow; ++x) {
                        const int lo = h.wc_lo[(size_t)rw * ow + x];
                        const float *w = &h.wc_w[h.wc_meta[rw].y + (size_t)x * Tw];
                        if (lo < 0 || lo + Tw > 8 * kmax) { printf("D read outside the written columns rw=%d x=%d\n", rw, x); return 1; }
                        float acc = 0.f;
                        for (int q = 0; q < Tw; ++q) acc = fmaf(w[q], D[(size_t)y * h.dp + lo + q], acc);
                        E[(size_t)y * ow + x] = acc;
                    }
            } else {
                const int kmax = (rh + rstep - 1) / rstep;
                if (kmax * rstep > erows) { printf("E rows\n"); return 1; }
                for (int y = 0; y < kmax * rstep; ++y)
                    for (int x = 0; x < ow; ++x) {
                        const Tap t = h.wf[(size_t)rw * ow + x];
                        const unsigned char *p0 = win + y * ow + t.lo, *p1 = win + y * ow + t.aux;
                        if (!in_r0(p0) || !in_r0(p1) || t.aux >= rw) { printf("window read rh=%d rw=%d\n", rh, rw); return 1; }
                        E[(size_t)y * ow + x] = fmaf(t.b, (float)*p1, t.a * (float)*p0);
                    }
            }
            // reference chain in double
            std::vector<double> crop((size_t)rh * rw);
            for (int y = 0; y < rh; ++y)
                for (int x = 0; x < rw; ++x) crop[(size_t)y * rw + x] = (double)win[y * ow + x] / 255.0;
            std::vector<double> ref = crop;
            if (squeeze) ref = resize(resize(ref, rh, rw, fh, fw, aa), fh, fw, rh, rw, aa);
            ref = resize(ref, rh, rw, oh, ow, aa);
            for (int y = 0; y < oh; ++y) {
                const int4 e = h.hy[(size_t)rh * oh + y];
                const int i0 = e.x & 0xFF, i1 = (e.x >> 8) & 0xFF, i2 = e.x >> 16;
                if (i0 >= er || i1 >= er || i2 >= er) { printf("row index rh=%d\n", rh); return 1; }
                float w0, w1, w2;
                memcpy(&w0, &e.y, 4); memcpy(&w1, &e.z, 4); memcpy(&w2, &e.w, 4);
                for (int x = 0; x < ow; ++x) {
                    float o = fmaf(w1, E[(size_t)i1 * ow + x], w0 * E[(size_t)i0 * ow + x]);
                    if (squeeze) o = fmaf(w2, E[(size_t)i2 * ow + x], o);
                    const double err = std::fabs((double)o - ref[(size_t)y * ow + x]);
                    if (!(err <= worst)) worst = err;      // NaN-propagating max
                }
            }
            ++cases;
        }
    printf("max_err %.3e cases %ld lds %zu\n", worst, cases, flex3_lds(h, c));
    return 0;
}
