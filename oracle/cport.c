/* oracle/cport.c — plain-C restatement of the per-env CPU work of the reference's hot path.
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline); never linked into the product.
 *
 *   agxo_get_state   AtariEnv._get_state integer part   reference atari_env.py:73-75
 *                    ALE luminance round(.2989r+.5870g+.1140b) + OpenCV 8-bit INTER_LINEAR (11-bit fixed point)
 *   agxo_ingest      AtariEnv._step / _reset image part (max over the sampled screens, _reset_buffer, deque.append)
 *                    reference atari_env.py:80-82,111-112,121-133,143
 *   agxo_fovea_fixed FixedFovealEnv._fov_step + _get_fov_state (absolute, resize_to_full)   reference fov_env.py:166-183,193-195
 *   agxo_step_fixed  the two in a row: one env step of the headline config
 * Arithmetic follows oracle/oracle.py function for function; tests/test_oracle_cport.py checks they agree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RAW_H 210
#define RAW_W 160

static inline uint8_t lum(const uint8_t *p) {
    double x = ((double)p[0] * 0.2989 + (double)p[1] * 0.5870) + (double)p[2] * 0.1140;
    return (uint8_t)round(x);
}

typedef struct { int i0, i1, c0, c1; } cvtap;

static void cv_axis(int src, int dst, int is_x, cvtap *t) {
    double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (is_x) {
            if (s < 0) { s = 0; f = 0.f; }
            if (s >= src - 1) { s = src - 1; f = 0.f; }
            t[d].i0 = s; t[d].i1 = s + 1 < src ? s + 1 : src - 1;
        } else {
            t[d].i0 = s < 0 ? 0 : (s > src - 1 ? src - 1 : s);
            t[d].i1 = s + 1 < 0 ? 0 : (s + 1 > src - 1 ? src - 1 : s + 1);
        }
        t[d].c0 = (int)nearbyintf((1.f - f) * 2048.f);
        t[d].c1 = (int)nearbyintf(f * 2048.f);
    }
}

/* rgb u8[210][160][3] -> out u8[oh][ow]; dsize = (ow, oh) */
void agxo_get_state(const uint8_t *rgb, int oh, int ow, uint8_t *out) {
    uint8_t *gray = (uint8_t *)malloc(RAW_H * RAW_W);      /* (per call: callers may step envs from several threads) */
    cvtap *tx = (cvtap *)malloc(sizeof(cvtap) * ow), *ty = (cvtap *)malloc(sizeof(cvtap) * oh);
    int *h0 = (int *)malloc(sizeof(int) * ow), *h1 = (int *)malloc(sizeof(int) * ow);
    cv_axis(RAW_W, ow, 1, tx);
    cv_axis(RAW_H, oh, 0, ty);
    for (int i = 0; i < RAW_H * RAW_W; ++i) gray[i] = lum(rgb + 3 * i);
    for (int y = 0; y < oh; ++y) {
        const uint8_t *r0 = gray + ty[y].i0 * RAW_W, *r1 = gray + ty[y].i1 * RAW_W;
        for (int x = 0; x < ow; ++x) {
            h0[x] = r0[tx[x].i0] * tx[x].c0 + r0[tx[x].i1] * tx[x].c1;
            h1[x] = r1[tx[x].i0] * tx[x].c0 + r1[tx[x].i1] * tx[x].c1;
        }
        for (int x = 0; x < ow; ++x)
            out[y * ow + x] = (uint8_t)(((((ty[y].c0 * (h0[x] >> 4)) >> 16) + ((ty[y].c1 * (h1[x] >> 4)) >> 16) + 2) >> 2));
    }
    free(tx); free(ty); free(h0); free(h1); free(gray);
}

/* AtariEnv._step / _reset image part for one env (reference atari_env.py:80-82,111-112,121-133): `clear` = _reset_buffer (the
 * stack is zero-filled first), then observation = max over the first nvalid (0..2) sampled screens (zeros for 0), deque.append.
 *   frames u8[2][210][160][3], ring u8[fs][oh][ow] (oldest..newest, shifted in place) */
void agxo_ingest(const uint8_t *frames, int nvalid, int clear, uint8_t *ring, int fs, int oh, int ow) {
    const int px = oh * ow;
    uint8_t *a = (uint8_t *)calloc(px, 1), *b = (uint8_t *)malloc(px);
    for (int f = 0; f < nvalid && f < 2; ++f) {
        agxo_get_state(frames + (size_t)f * RAW_H * RAW_W * 3, oh, ow, b);
        for (int i = 0; i < px; ++i) if (b[i] > a[i]) a[i] = b[i];
    }
    if (clear) memset(ring, 0, (size_t)fs * px);
    memmove(ring, ring + px, (size_t)(fs - 1) * px);       /* deque.append */
    memcpy(ring + (size_t)(fs - 1) * px, a, px);
    free(a); free(b);
}

/* FixedFovealEnv._fov_step + _get_fov_state, absolute mode, resize_to_full (reference fov_env.py:166-183,193-195):
 *   ring u8[fs][oh][ow], action (row, col) doubles -> obs double[fs][oh][ow] (the reference computes the resize in float64 on
 *   float32 k/255 values), fov_loc out. */
void agxo_fovea_fixed(const uint8_t *ring, int fs, int oh, int ow, int fh, int fw, const double *action, double *obs, int *fov_loc) {
    const int px = oh * ow;
    /* rint(clip(action, 0, obs - fov)) */
    int loc[2];
    const int bound[2] = {oh - fh, ow - fw};
    for (int k = 0; k < 2; ++k) {
        double v = action[k];
        if (!(v > 0.0)) v = 0.0;
        if (v > bound[k]) v = bound[k];
        loc[k] = (int)nearbyint(v);
        fov_loc[k] = loc[k];
    }
    /* bilinear align_corners=False, float64, on float32 k/255 values */
    for (int j = 0; j < fs; ++j) {
        const uint8_t *fr = ring + (size_t)j * px;
        for (int y = 0; y < oh; ++y) {
            double fy = ((double)fh / oh) * (y + 0.5) - 0.5; if (fy < 0) fy = 0;
            int y0 = (int)fy; if (y0 > fh - 1) y0 = fh - 1;
            int y1 = y0 + (y0 < fh - 1); double ly = fy - y0;
            for (int x = 0; x < ow; ++x) {
                double fx = ((double)fw / ow) * (x + 0.5) - 0.5; if (fx < 0) fx = 0;
                int x0 = (int)fx; if (x0 > fw - 1) x0 = fw - 1;
                int x1 = x0 + (x0 < fw - 1); double lx = fx - x0;
#define P(yy, xx) ((double)((float)fr[(loc[0] + (yy)) * ow + loc[1] + (xx)] / 255.0f))
                obs[((size_t)j * oh + y) * ow + x] =
                    (1 - ly) * ((1 - lx) * P(y0, x0) + lx * P(y0, x1)) + ly * ((1 - lx) * P(y1, x0) + lx * P(y1, x1));
#undef P
            }
        }
    }
}

/* One env step of the headline config on the CPU, the way the reference does it per env: agxo_ingest, then agxo_fovea_fixed. */
void agxo_step_fixed(const uint8_t *frames, int nvalid, uint8_t *ring, int fs, int oh, int ow, int fh, int fw,
                     const double *action, double *obs, int *fov_loc) {
    agxo_ingest(frames, nvalid, 0, ring, fs, oh, ow);
    agxo_fovea_fixed(ring, fs, oh, ow, fh, fw, action, obs, fov_loc);
}
