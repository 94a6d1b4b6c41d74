// agx_api.hip — the C ABI declared in include/agx.h: context, table builders, launches.
// Built for gfx950 only (see ../build.py): hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "agx.h"
#include "agx_kernels.h"
#include "agx_host_tables.h"
#include "agx_device_guard.h"

using namespace agx;

struct agx_ctx {
    agx_config cfg;
    uint8_t *ring = nullptr;
    int32_t *head[2] = {nullptr, nullptr};
    int32_t *loc[2] = {nullptr, nullptr};
    int32_t *res[2] = {nullptr, nullptr};
    int cur_head = 0;
    int cur_fov = 0;
    int2 *in_xtab = nullptr;   // K1 tables
    int4 *in_ytab = nullptr;
    int2 *in_xtab12 = nullptr; // K1 band12 form: {2 * x0, (a0 | a1 << 16) << 4} / {b0 << 8, b1 << 8}
    int2 *in_ytab12 = nullptr;
    bool band12_ok = false;    // 12-row bands all full, affine source rows, every x tap pair adjacent
    // compact source screens (agx_ingest_compact): the source rows the vertical resize reads, ascending, and the y table with
    // PACKED row indices; compact12_ok: band12_ok and every (y0, y1) pair disjoint and ascending (packed rows 2 dy, 2 dy + 1)
    std::vector<int32_t> src_rows;
    int4 *in_ytab_c = nullptr;
    bool compact12_ok = false;
    Tap *fx_xtab = nullptr;    // K2 tables
    Tap *fx_ytab = nullptr;
    int2 *per_ln[4] = {nullptr, nullptr, nullptr, nullptr};   // K3 tables
    float *per_w[4] = {nullptr, nullptr, nullptr, nullptr};
    int per_maxt[4] = {0, 0, 0, 0};
    // K4 table families (wd, wb, wf, hd, hb, hf): entries, weights, per-size meta
    int2 *flex_ln[6] = {};
    float *flex_w[6] = {};
    int4 *flex_meta[6] = {};
    size_t flex_tab_floats = 0;   // worst-case LDS floats of the staged tables
    // K4 resize_to_full form (k_fovea_flexible3): composed per-axis operators, see agx_k4_flex3.h
    Flex3Params f3{};
    bool f3_ok = false;
    // K4 raw-crop / mask-out / packed forms (k_fovea_flexible_raw3, agx_k4_raw3.h)
    FlexRawParams fr{};
    bool fr_ok = false;
    size_t fr_lds = 0;
    int64_t *pack_local = nullptr;   // agx_fovea_flexible_packed: [N] block-local exclusive offsets, [ceil(N/256)] block totals
    int64_t *pack_block = nullptr;   // (both allocated in agx_create for flexible raw-crop contexts: no allocation in a step call)
    // K3 tuned form 3 (k_fovea_peripheral3)
    Per3Params p3{};
    int p3_mt = 0;
    size_t p3_lds = 0;
    std::vector<void *> owned;    // further device allocations freed by agx_destroy
#ifdef AGX_EXPERIMENTS
    // split step (agx_step_fixed): env-range parts 1.. run on these internal streams, forked from / joined to the caller's
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
#endif
    int band_rows = 0;
    int ingest_t = 256;
    int rows_touched = 0;
    int y_affine = 0, y_mul = 0, y_add = 0, y_shift = 0;   // see IngestParams
    int init_r = 0, init_c = 0;
    hipEvent_t prof[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // agx_profile_next: [ingest | fovea][start | stop]
    // Testing knobs, read from the environment ONCE PER CONTEXT in agx_create (so one process can hold contexts of
    // several forms and compare them).  The shipped library has only the five that select a FALLBACK kernel, i.e. the
    // kernel (or launch sequence) other geometries get anyway (tests/test_gpu_parity.py::test_generic_fallback_kernel_matches_tuned); the rest
    // exist in the experiments build (-DAGX_EXPERIMENTS, experiments/agx_experiments.h) only.
    struct Tune {
        int generic = 0;         // AGX_FOVEA_GENERIC     K3 / K4 through the generic fallback kernel
        int no_full = 0;         // AGX_INGEST_NO_FULL    general k_ingest<256> even where k_ingest_full12 applies
        int flex_v2 = 0;         // AGX_FLEX_V2           K4 through k_fovea_flexible2 (pass-by-pass form)
        int per_v2 = 0;          // AGX_PER_V2            K3 through k_fovea_peripheral2
        int packed_unfused = 0;  // AGX_STEP_PACKED_UNFUSED  agx_step_flexible_packed as the three stand-alone launches
        // ---- experiments build only (always 0 in libagx.so)
        int ingest_t = 0;        // AGX_INGEST_T          128 | 256 threads per ingest workgroup
        int band_rows = 0;       // AGX_INGEST_BAND_ROWS  output rows per ingest workgroup (<= the default)
        int pipe_parts = 0;      // AGX_INGEST_PIPE       k_ingest_pipe with this many workgroups per env
        int wave = 0;            // AGX_INGEST_WAVE       wave-private (barrier-free) ingest
        int pair = 0;            // AGX_FOVEA_PAIR        two ring slots per K2 workgroup
        int fused = 0;           // AGX_STEP_FUSED        agx_step_fixed as one heterogeneous launch + tail
        int pair12 = 0;          // AGX_INGEST_PAIR12     k_ingest_pair12: two envs' bands per workgroup
        int step_env = 0;        // AGX_STEP_ENV          agx_step_fixed as ONE launch, one workgroup per env (k_step_env)
        int split = 0;           // AGX_STEP_SPLIT        env-range parts of agx_step_fixed on internal streams
        int aux_prio = 0;        // AGX_STEP_AUX_PRIO     -1 | 0 | 1: priority of the internal streams relative to normal
        int packed_wave = 0;     // AGX_PACKED_WAVE       packed ragged crops with one wave (64-thread workgroup) per (slot, env) item
    } tune;
    std::string err;
};

// Launch of a benchmarked kernel.  When agx_profile_next armed a start / stop event pair for this kernel family the
// launch goes through hipExtLaunchKernelGGL, which stamps the two events with the dispatch's own begin / end times (the
// times rocprofv3's kernel trace reports) instead of bracketing it with two more packets on the stream.
#define AGX_LAUNCH(which, kernel, grid, block, lds, stream, ...)                                              \
    do {                                                                                                      \
        hipEvent_t e0_ = ctx->prof[which][0], e1_ = ctx->prof[which][1];                                      \
        if (e0_ && e1_) {                                                                                     \
            ctx->prof[which][0] = ctx->prof[which][1] = nullptr;                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)(lds), stream, e0_, e1_, 0, __VA_ARGS__);    \
        } else {                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
        }                                                                                                     \
    } while (0)

namespace {

int env_int(const char *name, int dflt_if_set_empty = 1) {
    const char *e = getenv(name);
    if (!e) return 0;
    return *e ? atoi(e) : dflt_if_set_empty;
}

thread_local std::string g_create_err;

int fail(agx_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_err = buf;
    return code;
}

#define AGX_HIP(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail((ctx), AGX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct HipDeviceApi {
    static int get(int *dev) { return hipGetDevice(dev) == hipSuccess ? 0 : 1; }
    static int set(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : 1; }
};
using DeviceGuard = agx::DeviceGuardT<HipDeviceApi>;      // agx_device_guard.h (unit-tested with a mocked runtime)

inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }

// ---- OpenCV 8-bit INTER_LINEAR tables (imgproc/src/resize.cpp), see oracle/oracle.py for the
// restated algorithm: inv_scale = dst/src, scale = 1/inv_scale, f = (float)((d+.5)*scale-.5),
// s = floor(f), f -= s, coefficients = rint-half-even(float * 2048).
inline int cv_round_f(float v) { return (int)std::nearbyintf(v); }   // default mode: half-to-even

void cv_axis(int src, int dst, bool is_x, std::vector<int> &i0, std::vector<int> &i1,
             std::vector<int> &c0, std::vector<int> &c1) {
    const double inv_scale = (double)dst / (double)src;
    const double scale = 1.0 / inv_scale;
    i0.resize(dst); i1.resize(dst); c0.resize(dst); c1.resize(dst);
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (is_x) {                       // x axis: clamp index and zero the fraction
            if (s < 0) { s = 0; f = 0.f; }
            if (s >= src - 1) { s = src - 1; f = 0.f; }
            i0[d] = s;
            i1[d] = std::min(s + 1, src - 1);
        } else {                          // y axis: keep coefficients, clip the row indices
            i0[d] = std::min(std::max(s, 0), src - 1);
            i1[d] = std::min(std::max(s + 1, 0), src - 1);
        }
        c0[d] = cv_round_f((1.f - f) * 2048.f);
        c1[d] = cv_round_f(f * 2048.f);
    }
}

// One axis of torchvision Resize (ATen upsample_bilinear2d, align_corners=False) as explicit taps:
// antialiased triangle filter when down-scaling with antialias on (_compute_indices_min_size_weights_aa),
// plain bilinear otherwise (area_pixel_compute_source_index); all in double, weights normalised.
// compile-time tap bounds the tuned kernels are instantiated for (0 = run-time loops)
int tap_bucket(int n) { return n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : n <= 12 ? 12 : n <= 16 ? 16 : n; }

void axis_taps(int n_in, int n_out, bool antialias, std::vector<int2> &ln, std::vector<float> &w, int &maxt) {
    std::vector<std::vector<double>> ws(n_out);
    ln.resize(n_out);
    const double scale = (double)n_in / (double)n_out;
    const bool aa = antialias && n_in > n_out;
    maxt = 1;
    for (int i = 0; i < n_out; ++i) {
        if (aa) {
            const double support = scale, invscale = 1.0 / scale, center = scale * (i + 0.5);
            long long xmin = (long long)(center - support + 0.5);
            if (xmin < 0) xmin = 0;
            long long xmax = (long long)(center + support + 0.5);
            if (xmax > n_in) xmax = n_in;
            double total = 0.0;
            for (long long jx = xmin; jx < xmax; ++jx) {
                double x = ((double)jx - center + 0.5) * invscale;
                if (x < 0) x = -x;
                const double wv = x < 1.0 ? 1.0 - x : 0.0;
                ws[i].push_back(wv);
                total += wv;
            }
            if (total != 0.0)
                for (double &v : ws[i]) v /= total;
            ln[i] = make_int2((int)xmin, (int)ws[i].size());
        } else {
            double f = scale * (i + 0.5) - 0.5;
            if (f < 0.0) f = 0.0;
            int i0 = (int)f;
            if (i0 > n_in - 1) i0 = n_in - 1;
            const int i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
            const double l1 = f - i0;
            if (i1 == i0) ws[i] = {1.0};              // (1-l)*p + l*p
            else ws[i] = {1.0 - l1, l1};
            ln[i] = make_int2(i0, (int)ws[i].size());
        }
        maxt = std::max(maxt, (int)ws[i].size());
    }
    maxt = tap_bucket(maxt);                          // zero-padded to the kernels' compile-time bounds
    w.assign((size_t)n_out * maxt, 0.f);
    for (int i = 0; i < n_out; ++i)
        for (size_t k = 0; k < ws[i].size(); ++k) w[(size_t)i * maxt + k] = (float)ws[i][k];
}

size_t per2_tables(const agx_config &c) {
    std::vector<int2> ln;
    std::vector<float> w;
    int m1 = 0, m3 = 0;
    axis_taps(c.obs_h, c.per_h, c.antialias != 0, ln, w, m1);
    axis_taps(c.per_h, c.obs_h, c.antialias != 0, ln, w, m3);
    return (size_t)c.per_h * (sizeof(int2) + m1 * sizeof(float)) + (size_t)c.obs_h * (sizeof(int2) + m3 * sizeof(float));
}

size_t per2_lds(const agx_config &c) {
    const size_t raw = ((size_t)c.obs_h * c.obs_w + 15) & ~(size_t)15;
    // A[oh][pw] and C[ph][ow] share one region (C is written after A's last read), then B[ph][pw]
    const size_t ac = (std::max((size_t)c.obs_h * c.per_w, (size_t)c.per_h * c.obs_w) + 3) & ~(size_t)3;
    const size_t b = ((size_t)c.per_h * c.per_w + 3) & ~(size_t)3;
    // + the pass-1 and pass-3 tap tables ({lo,n} + zero-padded weights; bounded by the bucketed tap counts,
    //   which per2_tables() below computes the same way agx_create does)
    return 1024 + raw + (ac + b) * sizeof(float) + per2_tables(c);
}

// K4: one family = the taps of every window size r in [1, rmax] along one axis.
//   which = 0: r -> fov (squeeze)   1: fov -> r (expand back)   2: r -> obs (final resize)
struct HostFamily {
    std::vector<int2> ln;
    std::vector<float> w;
    std::vector<int4> meta;                       // [rmax + 1]
    std::vector<size_t> floats;                   // LDS floats of the staged table of size r
};
HostFamily build_family(int which, int rmax, int fov, int obs, bool antialias) {
    HostFamily f;
    f.meta.assign(rmax + 1, make_int4(0, 0, 1, 0));
    f.floats.assign(rmax + 1, 0);
    for (int r = 1; r <= rmax; ++r) {
        std::vector<int2> ln;
        std::vector<float> w;
        int maxt = 0;
        const int n_in = which == 0 ? r : (which == 1 ? fov : r);
        const int n_out = which == 0 ? fov : (which == 1 ? r : obs);
        axis_taps(n_in, n_out, antialias, ln, w, maxt);
        f.meta[r] = make_int4((int)f.ln.size(), (int)f.w.size(), maxt, n_out);
        f.floats[r] = (((size_t)2 * n_out + (size_t)n_out * maxt) + 3) & ~(size_t)3;
        f.ln.insert(f.ln.end(), ln.begin(), ln.end());
        f.w.insert(f.w.end(), w.begin(), w.end());
    }
    return f;
}

size_t flex2_lds(const agx_config &c, size_t tab_floats) {
    const size_t raw = ((size_t)c.obs_h * c.obs_w + 15) & ~(size_t)15;
    const size_t ae = (std::max((size_t)c.obs_h * c.fov_w, (size_t)c.fov_h * c.obs_w) + 3) & ~(size_t)3;
    const size_t b = ((size_t)c.fov_h * c.fov_w + 3) & ~(size_t)3;
    const size_t cc = ((size_t)c.fov_h * c.obs_w + 3) & ~(size_t)3;      // C aliases the raw frame bytes
    return 1024 + std::max(raw, cc * sizeof(float)) + (ae + b + tab_floats) * sizeof(float);
}

template <class T>
int upload(agx_ctx *ctx, T **dptr, const std::vector<T> &h) {
    AGX_HIP(ctx, hipMalloc(reinterpret_cast<void **>(dptr), h.size() * sizeof(T)));
    AGX_HIP(ctx, hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return AGX_OK;
}

template <class T>
int upload_owned(agx_ctx *ctx, const T **dptr, const std::vector<T> &h) {
    T *d = nullptr;
    const int rc = upload(ctx, &d, h);
    if (rc == AGX_OK) {
        ctx->owned.push_back(d);
        *dptr = d;
    }
    return rc;
}

bool has_fovea(const agx_config &c) { return c.kind != AGX_KIND_BASE; }

size_t fixed_lds(const agx_config &c) {
    // window rows u8 [fh][ow] (16-B padded) | ytab[oh] | H[fh][ow]   (the carve of fovea_fixed_body)
    const size_t raw = ((size_t)c.fov_h * c.obs_w + 15) & ~(size_t)15;
    size_t b = raw;
    if (c.out_mode == AGX_OUT_RESIZE) b += (size_t)c.obs_h * sizeof(Tap) + (size_t)c.fov_h * c.obs_w * sizeof(float);
    return b;
}
#ifdef AGX_EXPERIMENTS
size_t fixed2_lds(const agx_config &c) {     // k_fovea_fixed2: lut[256] f32 | raw frame u8 (16-B padded) | ytab[oh] | H[fh][ow]
    const size_t raw = ((size_t)c.obs_h * c.obs_w + 15) & ~(size_t)15;
    return 1024 + raw + (size_t)c.obs_h * sizeof(Tap) + (size_t)c.fov_h * c.obs_w * sizeof(float);
}
#endif

// second LDS buffer of the generic kernels, in floats: flexible ping-pongs two full frames,
// peripheral keeps A[oh][pw] | B[ph][pw] | C[ph][ow] there
size_t generic_buf1(const agx_config &c) {
    const size_t cap = ((size_t)c.obs_h * c.obs_w + 3) & ~(size_t)3;
    if (c.kind != AGX_KIND_PERIPHERAL) return cap;
    const size_t abc = (size_t)c.obs_h * c.per_w + (size_t)c.per_h * c.per_w + (size_t)c.per_h * c.obs_w;
    return (abc + 3) & ~(size_t)3;
}

size_t generic_lds(const agx_config &c) {
    const size_t cap = ((size_t)c.obs_h * c.obs_w + 3) & ~(size_t)3;
    int tmax = std::max(std::max(c.obs_h, c.obs_w), std::max(c.fov_h, c.fov_w));
    if (c.kind == AGX_KIND_PERIPHERAL) tmax = std::max(tmax, std::max(c.per_h, c.per_w));
    return (cap + generic_buf1(c)) * sizeof(float) + (size_t)tmax * sizeof(Tap);
}

constexpr size_t kMaxLds = 160 * 1024;   // gfx950: a workgroup may take the whole 160 KiB of its CU

}  // namespace

extern "C" {

int agx_abi_version(void) { return AGX_ABI_VERSION; }

#ifndef AGX_SRC_HASH
#define AGX_SRC_HASH "unknown"
#endif
#define AGX_STR2(x) #x
#define AGX_STR(x) AGX_STR2(x)
const char *agx_build_info(void) { return "libagx abi " AGX_STR(AGX_ABI_VERSION) " src " AGX_SRC_HASH; }

const char *agx_last_error(const agx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int agx_device_pci_bus_id(int device, char *buf, int len) {
    if (!buf || len < 13) return fail(nullptr, AGX_E_INVALID, "agx_device_pci_bus_id: buffer of at least 13 bytes needed");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();                 // the runtime's last-error slot is sticky: do not leave it to the caller's next HIP call
        return fail(nullptr, AGX_E_HIP, "no HIP device");
    }
    if (device < 0 || device >= ndev) return fail(nullptr, AGX_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    const hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, AGX_E_HIP, "hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e));
    }
    return AGX_OK;
}

int agx_destroy(agx_ctx *ctx) {
    if (!ctx) return AGX_OK;
    DeviceGuard g(ctx->cfg.device);
    void *ptrs[] = {ctx->ring, ctx->head[0], ctx->head[1], ctx->loc[0], ctx->loc[1], ctx->res[0], ctx->res[1],
                    ctx->in_xtab, ctx->in_ytab, ctx->in_ytab_c, ctx->in_xtab12, ctx->in_ytab12, ctx->fx_xtab, ctx->fx_ytab,
                    ctx->per_ln[0], ctx->per_ln[1], ctx->per_ln[2], ctx->per_ln[3],
                    ctx->per_w[0], ctx->per_w[1], ctx->per_w[2], ctx->per_w[3],
                    ctx->flex_ln[0], ctx->flex_ln[1], ctx->flex_ln[2], ctx->flex_ln[3], ctx->flex_ln[4], ctx->flex_ln[5],
                    ctx->flex_w[0], ctx->flex_w[1], ctx->flex_w[2], ctx->flex_w[3], ctx->flex_w[4], ctx->flex_w[5],
                    ctx->flex_meta[0], ctx->flex_meta[1], ctx->flex_meta[2], ctx->flex_meta[3], ctx->flex_meta[4],
                    ctx->flex_meta[5]};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : ctx->owned)
        if (p) (void)hipFree(p);
#ifdef AGX_EXPERIMENTS
    for (hipStream_t st : ctx->aux)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (hipEvent_t e : ctx->ev_join)
        if (e) (void)hipEventDestroy(e);
#endif
    delete ctx;
    return AGX_OK;
}

int agx_create(const agx_config *cfg, agx_ctx **out) {
    if (!cfg || !out) return fail(nullptr, AGX_E_INVALID, "agx_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(agx_config))
        return fail(nullptr, AGX_E_INVALID, "agx_create: struct_size %d != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(agx_config));
    const agx_config &c = *cfg;
    if (c.num_envs < 1 || c.num_envs > 65535)   // env index rides on gridDim.y / gridDim.z
        return fail(nullptr, AGX_E_INVALID, "num_envs must be in [1, 65535] per context (shard larger batches)");
    if (c.raw_h != kRawH || c.raw_w != kRawW)
        return fail(nullptr, AGX_E_INVALID, "raw screen must be %dx%d (ALE), got %dx%d", kRawH, kRawW, c.raw_h, c.raw_w);
    if (c.obs_h < 4 || c.obs_w < 4 || (c.obs_w & 3) || c.obs_w > 1024 || c.obs_h > 1024)
        return fail(nullptr, AGX_E_INVALID, "obs_size (%d,%d): need 4 <= h,w <= 1024 and w %% 4 == 0", c.obs_h, c.obs_w);
    if (c.frame_stack < 1 || c.frame_stack > 16) return fail(nullptr, AGX_E_INVALID, "frame_stack must be in [1,16]");
    if (c.kind < AGX_KIND_BASE || c.kind > AGX_KIND_PERIPHERAL) return fail(nullptr, AGX_E_INVALID, "unknown kind %d", c.kind);
    if (has_fovea(c)) {
        // assert (np.array(self.fov_size) < np.array(self.obs_size)).all()   fov_env.py:112
        if (c.fov_h < 1 || c.fov_w < 1 || c.fov_h >= c.obs_h || c.fov_w >= c.obs_w)
            return fail(nullptr, AGX_E_INVALID, "fov_size (%d,%d) must be >= 1 and < obs_size (%d,%d)", c.fov_h, c.fov_w,
                        c.obs_h, c.obs_w);
        if (c.out_mode < AGX_OUT_RAW || c.out_mode > AGX_OUT_MASK) return fail(nullptr, AGX_E_INVALID, "bad out_mode");
        if (c.action_mode != AGX_MODE_ABSOLUTE && c.action_mode != AGX_MODE_RELATIVE)
            return fail(nullptr, AGX_E_INVALID, "bad action_mode");
        if (c.action_mode == AGX_MODE_RELATIVE && !(c.sas_lo <= c.sas_hi))
            return fail(nullptr, AGX_E_INVALID, "relative mode needs sensory_action_space lo <= hi");
        if (!std::isfinite(c.init_loc[0]) || !std::isfinite(c.init_loc[1]))
            return fail(nullptr, AGX_E_INVALID, "fov_init_loc must be finite");
        if (c.kind == AGX_KIND_PERIPHERAL && (c.per_h < 1 || c.per_w < 1 || c.per_h > 1024 || c.per_w > 1024))
            return fail(nullptr, AGX_E_INVALID, "peripheral_res (%d,%d) out of range", c.per_h, c.per_w);
        // the LDS of the kernel that will actually run: the tuned peripheral kernel when its plan fits (and the
        // testing knob does not force the fallback), otherwise the generic one
        const bool per_tuned = env_int("AGX_FOVEA_GENERIC") == 0 && per2_lds(c) <= kMaxLds && c.per_w <= kThreads;
        const size_t lds = c.kind == AGX_KIND_FIXED ? fixed_lds(c)
                           : (c.kind == AGX_KIND_PERIPHERAL && per_tuned ? per2_lds(c) : generic_lds(c));
        if (lds > kMaxLds)
            return fail(nullptr, AGX_E_INVALID, "geometry needs %zu B of LDS per workgroup (limit %zu)", lds, kMaxLds);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, AGX_E_HIP, "no HIP device: libagx has no CPU path");
    if (c.device < 0 || c.device >= ndev) return fail(nullptr, AGX_E_INVALID, "device %d out of range (%d visible)", c.device, ndev);

    agx_ctx *ctx = new (std::nothrow) agx_ctx;
    if (!ctx) return fail(nullptr, AGX_E_NOMEM, "out of host memory");
    ctx->cfg = c;
    ctx->tune.generic = env_int("AGX_FOVEA_GENERIC");
    ctx->tune.no_full = env_int("AGX_INGEST_NO_FULL");
    ctx->tune.flex_v2 = env_int("AGX_FLEX_V2");
    ctx->tune.per_v2 = env_int("AGX_PER_V2");
    ctx->tune.packed_unfused = env_int("AGX_STEP_PACKED_UNFUSED");
#ifdef AGX_EXPERIMENTS
    ctx->tune.ingest_t = env_int("AGX_INGEST_T");
    ctx->tune.band_rows = env_int("AGX_INGEST_BAND_ROWS");
    ctx->tune.pipe_parts = env_int("AGX_INGEST_PIPE");
    ctx->tune.wave = env_int("AGX_INGEST_WAVE");
    ctx->tune.pair = env_int("AGX_FOVEA_PAIR");
    ctx->tune.fused = env_int("AGX_STEP_FUSED");
    ctx->tune.pair12 = env_int("AGX_INGEST_PAIR12");
    ctx->tune.step_env = env_int("AGX_STEP_ENV");
    ctx->tune.split = env_int("AGX_STEP_SPLIT");
    ctx->tune.aux_prio = env_int("AGX_STEP_AUX_PRIO", 0);
    ctx->tune.packed_wave = env_int("AGX_PACKED_WAVE");
#endif
    DeviceGuard g(c.device);
    int rc = AGX_OK;
    auto bail = [&](int code) {
        g_create_err = ctx->err;
        agx_destroy(ctx);
        return code;
    };
    const size_t N = c.num_envs, fsz = (size_t)c.obs_h * c.obs_w;
#define TRY(expr)                                  \
    do {                                           \
        hipError_t e_ = (expr);                    \
        if (e_ != hipSuccess) {                    \
            fail(ctx, AGX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
            return bail(AGX_E_HIP);                \
        }                                          \
    } while (0)
    TRY(hipMalloc(reinterpret_cast<void **>(&ctx->ring), N * c.frame_stack * fsz));
    TRY(hipMemset(ctx->ring, 0, N * c.frame_stack * fsz));
    for (int b = 0; b < 2; ++b) {
        TRY(hipMalloc(reinterpret_cast<void **>(&ctx->head[b]), N * sizeof(int32_t)));
        TRY(hipMemset(ctx->head[b], 0, N * sizeof(int32_t)));
    }
    // K1 tables (only meaningful for square obs; built anyway, agx_ingest checks)
    {
        std::vector<int> x0, x1, a0, a1, y0, y1, b0, b1;
        cv_axis(kRawW, c.obs_w, true, x0, x1, a0, a1);
        cv_axis(kRawH, c.obs_h, false, y0, y1, b0, b1);
        std::vector<int2> xt(c.obs_w);
        std::vector<int4> yt(c.obs_h);
        for (int i = 0; i < c.obs_w; ++i) xt[i] = make_int2(x0[i] | (x1[i] << 16), (a0[i] & 0xFFFF) | (a1[i] << 16));
        std::set<int> touched;
        for (int i = 0; i < c.obs_h; ++i) {
            yt[i] = make_int4(y0[i], y1[i], b0[i], b1[i]);
            touched.insert(y0[i]);
            touched.insert(y1[i]);
        }
        ctx->rows_touched = (int)touched.size();
        // compact screens: packed index of every touched source row, the y table in packed indices
        ctx->src_rows.assign(touched.begin(), touched.end());
        std::vector<int> packed_of(kRawH, -1);
        for (size_t k = 0; k < ctx->src_rows.size(); ++k) packed_of[ctx->src_rows[k]] = (int)k;
        std::vector<int4> ytc(c.obs_h);
        bool pairs = true;
        for (int i = 0; i < c.obs_h; ++i) {
            ytc[i] = make_int4(packed_of[y0[i]], packed_of[y1[i]], b0[i], b1[i]);
            pairs = pairs && ytc[i].x == 2 * i && ytc[i].y == 2 * i + 1;
        }
        if ((rc = upload(ctx, &ctx->in_ytab_c, ytc)) != AGX_OK) return bail(rc);
        // look for an exact integer form of the row table: y0 = (dy*mul + add) >> shift, y1 = min(y0+1, H-1)
        for (int sh = 0; sh <= 12 && !ctx->y_affine; ++sh) {
            const long mul = std::lround((double)kRawH / c.obs_h * (double)(1 << sh));
            for (long add = -(1L << sh); add <= (1L << (sh + 1)) && !ctx->y_affine; ++add) {
                bool ok = true;
                for (int i = 0; i < c.obs_h && ok; ++i) {
                    const long v = (i * mul + add) >> sh;
                    ok = v >= 0 && v == y0[i] && std::min<long>(v + 1, kRawH - 1) == y1[i];
                }
                if (ok) {
                    ctx->y_affine = 1;
                    ctx->y_mul = (int)mul;
                    ctx->y_add = (int)add;
                    ctx->y_shift = sh;
                }
            }
        }
        if ((rc = upload(ctx, &ctx->in_xtab, xt)) != AGX_OK) return bail(rc);
        if ((rc = upload(ctx, &ctx->in_ytab, yt)) != AGX_OK) return bail(rc);
        // band12 form: the same coefficients in the shape its phase 2 consumes
        std::vector<int2> xt12(c.obs_w), yt12(c.obs_h);
        bool adjacent = true;
        for (int i = 0; i < c.obs_w; ++i) {
            adjacent = adjacent && x1[i] == x0[i] + 1;
            xt12[i] = make_int2(2 * x0[i], (int)((((uint32_t)a0[i] & 0xFFFFu) | ((uint32_t)a1[i] << 16)) << 4));
        }
        for (int i = 0; i < c.obs_h; ++i) yt12[i] = make_int2(b0[i] << 8, b1[i] << 8);
        if ((rc = upload(ctx, &ctx->in_xtab12, xt12)) != AGX_OK) return bail(rc);
        if ((rc = upload(ctx, &ctx->in_ytab12, yt12)) != AGX_OK) return bail(rc);
        ctx->band12_ok = adjacent && ctx->y_affine && c.obs_h % 12 == 0 && (c.obs_w / 4) * 12 <= kThreads;
        // phase 2 of the band12 form reads 8 bytes at byte (2 x0) & ~3 of a 320-byte gray row: never past the row + the slack above
        for (int i = 0; i < c.obs_w && ctx->band12_ok; ++i)
            if (((2 * x0[i]) & ~3) + 8 > 2 * kRawW + 8) ctx->band12_ok = false;
        ctx->compact12_ok = ctx->band12_ok && pairs;
        // ingest workgroup: T threads produce band_rows output rows (band_rows * ow/4 <= T and the
        // 2 * band_rows row jobs fit the T/40 loader groups x 4 iterations).  128-thread workgroups give
        // 16 independent workgroups per CU whose load / compute phases interleave (AGX_INGEST_T tunes).
        const int ow4 = c.obs_w / 4;
        const int forced_t = ctx->tune.ingest_t;
        ctx->ingest_t = (forced_t == 128 || forced_t == 256) ? forced_t : 256;   // 8 WGs/CU whatever T: 256 fills the wave slots
        if (ow4 > 128) ctx->ingest_t = 256;
        ctx->band_rows = std::max(1, std::min(2 * (ctx->ingest_t / 40), ctx->ingest_t / ow4));
        // tuning knob (bands per env).  Measured at N=1024, same box: 12 rows x 7 bands (3.5 rounds of 2048 resident
        // workgroups) 46.0-46.7 us; 11 x 8 (4.0 rounds) 50-52; 10 x 9 53; and with wider workgroups whose grids are
        // exact rounds - 320 thr x 14 rows, 384 x 18, 512 x 21 - 50.2 / 49.2 / 48.5 us: the half-empty last round is
        // not what limits this kernel.
        const int forced_br = ctx->tune.band_rows;
        if (forced_br >= 1 && forced_br <= ctx->band_rows) ctx->band_rows = forced_br;
    }
    if (has_fovea(c)) {
        // _init_fov_loc: np.rint(fov_init_loc).astype(np.int32)  (not clipped)   fov_env.py:149-150
        ctx->init_r = (int)std::nearbyint(c.init_loc[0]);
        ctx->init_c = (int)std::nearbyint(c.init_loc[1]);
        if (ctx->init_r < 0 || ctx->init_c < 0 || ctx->init_r > c.obs_h - c.fov_h || ctx->init_c > c.obs_w - c.fov_w) {
            fail(ctx, AGX_E_INVALID, "fov_init_loc (%g,%g) puts the %dx%d window outside the %dx%d frame", c.init_loc[0],
                 c.init_loc[1], c.fov_h, c.fov_w, c.obs_h, c.obs_w);
            return bail(AGX_E_INVALID);
        }
        std::vector<int32_t> loc(2 * N), res(2 * N);
        for (size_t i = 0; i < N; ++i) {
            loc[2 * i] = ctx->init_r;
            loc[2 * i + 1] = ctx->init_c;
            res[2 * i] = c.fov_h;
            res[2 * i + 1] = c.fov_w;
        }
        for (int b = 0; b < 2; ++b) {
            if ((rc = upload(ctx, &ctx->loc[b], loc)) != AGX_OK) return bail(rc);
            if ((rc = upload(ctx, &ctx->res[b], res)) != AGX_OK) return bail(rc);
        }
        if (c.kind == AGX_KIND_FLEXIBLE) {
            size_t worst_w = 0, worst_h = 0;
            HostFamily fam[6];
            for (int k = 0; k < 6; ++k) {
                const bool is_w = k < 3;
                fam[k] = build_family(k % 3, is_w ? c.obs_w : c.obs_h, is_w ? c.fov_w : c.fov_h, is_w ? c.obs_w : c.obs_h,
                                      c.antialias != 0);
                if ((rc = upload(ctx, &ctx->flex_ln[k], fam[k].ln)) != AGX_OK) return bail(rc);
                if ((rc = upload(ctx, &ctx->flex_w[k], fam[k].w)) != AGX_OK) return bail(rc);
                if ((rc = upload(ctx, &ctx->flex_meta[k], fam[k].meta)) != AGX_OK) return bail(rc);
            }
            for (int r = 1; r <= c.obs_w; ++r) worst_w = std::max(worst_w, fam[0].floats[r] + fam[1].floats[r] + fam[2].floats[r]);
            for (int r = 1; r <= c.obs_h; ++r) worst_h = std::max(worst_h, fam[3].floats[r] + fam[4].floats[r] + fam[5].floats[r]);
            ctx->flex_tab_floats = worst_w + worst_h;
            // resize_to_full: the composed-operator kernel where its plan applies (thread-per-column, <= 16 taps)
            const Flex3Host f3 = build_flex3(c);
            if (f3.ok && flex3_lds(f3, c) <= kMaxLds) {
                Flex3Params &q = ctx->f3;
                if ((rc = upload_owned(ctx, &q.wf, f3.wf)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.wc_meta, f3.wc_meta)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.wc_lo, f3.wc_lo)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.wc_w, f3.wc_w)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_meta, f3.hd_meta)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_lo, f3.hd_lo)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_w, f3.hd_w)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hy, f3.hy)) != AGX_OK) return bail(rc);
                q.r0_bytes = f3.r0_bytes;
                q.r1_bytes = f3.r1_bytes;
                q.dp = f3.dp;
                ctx->f3_ok = true;
            }
            // raw-crop / mask-out: the composed squeeze-and-back form where its plan applies
            const FlexRawHost fr = build_flexraw(c);
            if (fr.ok && fr.lds(c) <= kMaxLds) {
                FlexRawParams &q = ctx->fr;
                if ((rc = upload_owned(ctx, &q.wb_meta, fr.wb_meta)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.wb_lo, fr.wb_lo)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.wb_w, fr.wb_w)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_meta, fr.hd_meta)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_lo, fr.hd_lo)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hd_w, fr.hd_w)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.hb, fr.hb)) != AGX_OK) return bail(rc);
                q.r0_bytes = fr.r0_bytes;
                q.r1_bytes = fr.r1_bytes;
                q.dp = fr.dp;
                q.n_envs = c.num_envs;
                ctx->fr_lds = fr.lds(c);
                ctx->fr_ok = true;
            }
            if (c.out_mode == AGX_OUT_RAW) {       // the packed form's scan buffers
                const size_t nb = (N + kScanEnvsPerBlock - 1) / kScanEnvsPerBlock;
                TRY(hipMalloc(reinterpret_cast<void **>(&ctx->pack_local), N * sizeof(int64_t)));
                ctx->owned.push_back(ctx->pack_local);
                TRY(hipMalloc(reinterpret_cast<void **>(&ctx->pack_block), nb * sizeof(int64_t)));
                ctx->owned.push_back(ctx->pack_block);
            }
        }
        if (c.kind == AGX_KIND_PERIPHERAL) {
            const int nin[4] = {c.obs_w, c.obs_h, c.per_w, c.per_h};
            const int nout[4] = {c.per_w, c.per_h, c.obs_w, c.obs_h};
            for (int k = 0; k < 4; ++k) {
                std::vector<int2> ln;
                std::vector<float> w;
                axis_taps(nin[k], nout[k], c.antialias != 0, ln, w, ctx->per_maxt[k]);
                if (k == 0)                          // the first pass reads u8 numerators: fold the /255 into its weights
                    for (float &v : w) v = (float)((double)v / 255.0);
                if ((rc = upload(ctx, &ctx->per_ln[k], ln)) != AGX_OK) return bail(rc);
                if ((rc = upload(ctx, &ctx->per_w[k], w)) != AGX_OK) return bail(rc);
            }
        }
        if (c.kind == AGX_KIND_PERIPHERAL) {
            const Per3Host h3 = build_per3(c);
            if (h3.ok && h3.lds <= kMaxLds) {
                Per3Params &q = ctx->p3;
                if ((rc = upload_owned(ctx, &q.lo0, h3.lo0)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.w0, h3.w0)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.lo1, h3.lo1)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.w1, h3.w1)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.x2, h3.x2)) != AGX_OK) return bail(rc);
                if ((rc = upload_owned(ctx, &q.y3, h3.y3)) != AGX_OK) return bail(rc);
                q.same = (c.per_h == c.obs_h && c.per_w == c.obs_w) ? 1 : 0;
                ctx->p3_mt = h3.mt;
                ctx->p3_lds = h3.lds;
            }
        }
        if (c.kind == AGX_KIND_FIXED && c.out_mode == AGX_OUT_RESIZE) {
            std::vector<Tap> xt(c.obs_w), yt(c.obs_h);
            for (int i = 0; i < c.obs_w; ++i) xt[i] = make_tap_lin2(i, c.fov_w, c.obs_w);
            for (int i = 0; i < c.obs_h; ++i) yt[i] = make_tap_lin2(i, c.fov_h, c.obs_h);
            if ((rc = upload(ctx, &ctx->fx_xtab, xt)) != AGX_OK) return bail(rc);
            if ((rc = upload(ctx, &ctx->fx_ytab, yt)) != AGX_OK) return bail(rc);
        }
    }
#undef TRY
    *out = ctx;
    return AGX_OK;
}

int agx_obs_shape(const agx_ctx *ctx, int32_t dims[4]) {
    if (!ctx || !dims) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    dims[0] = c.num_envs;
    dims[1] = c.frame_stack;
    const bool crop = c.kind == AGX_KIND_FIXED && c.out_mode == AGX_OUT_RAW;
    dims[2] = crop ? c.fov_h : c.obs_h;
    dims[3] = crop ? c.fov_w : c.obs_w;
    return AGX_OK;
}

int agx_profile_next(agx_ctx *ctx, int kernel_id, void *start_event, void *stop_event) {
    if (!ctx) return AGX_E_INVALID;
    const int which = kernel_id == AGX_K_INGEST ? 0 : (kernel_id == AGX_K_FOVEA ? 1 : -1);
    if (which < 0) return fail(ctx, AGX_E_INVALID, "agx_profile_next: kernel_id must be AGX_K_INGEST or AGX_K_FOVEA");
    if ((start_event == nullptr) != (stop_event == nullptr))
        return fail(ctx, AGX_E_INVALID, "agx_profile_next: pass both events, or neither to disarm");
    ctx->prof[which][0] = static_cast<hipEvent_t>(start_event);
    ctx->prof[which][1] = static_cast<hipEvent_t>(stop_event);
    return AGX_OK;
}

int64_t agx_algorithmic_bytes(const agx_ctx *ctx, int kernel_id) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    const int64_t N = c.num_envs, fs = c.frame_stack, px = (int64_t)c.obs_h * c.obs_w;
    switch (kernel_id) {
        case AGX_K_INGEST:   // two frames, only the source rows the vertical resize touches + one u8 slot
            return N * (2 * (int64_t)ctx->rows_touched * kRawRowBytes + px);
        case AGX_K_INGEST_GRAY_RAW:   // two gray frames, only the touched source rows, + one u8 slot
            return N * (2 * (int64_t)ctx->rows_touched * kRawW + px);
        case AGX_K_INGEST_RGB:   // one obs-sized RGB render in, one u8 slot out
            return N * px * 4;
        case AGX_K_FULL:
            return N * fs * px * 5;
        case AGX_K_FOVEA: {
            if (!has_fovea(c)) return AGX_E_STATE;
            const int64_t win = (int64_t)c.fov_h * c.fov_w;
            if (c.kind == AGX_KIND_PERIPHERAL) return N * fs * px * 5;
            if (c.kind == AGX_KIND_FIXED && c.out_mode == AGX_OUT_RAW) return N * fs * win * 5;
            return N * fs * (win + px * 4);
        }
        default:
            return AGX_E_INVALID;
    }
}

// ---------------------------------------------------------------- K1
static IngestParams ingest_params(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd) {
    const agx_config &c = ctx->cfg;
    IngestParams p;
    p.frames = d_frames;
    p.cmd = d_cmd;
    p.ring = ctx->ring;
    p.head_in = ctx->head[ctx->cur_head];
    p.head_out = ctx->head[ctx->cur_head ^ 1];
    p.xtab = ctx->in_xtab;
    p.ytab = ctx->in_ytab;
    p.oh = c.obs_h;
    p.ow = c.obs_w;
    p.fs = c.frame_stack;
    p.band_rows = ctx->band_rows;
    p.y_affine = ctx->y_affine;
    p.y_mul = ctx->y_mul;
    p.y_add = ctx->y_add;
    p.y_shift = ctx->y_shift;
    p.nbands = (c.obs_h + ctx->band_rows - 1) / ctx->band_rows;
    p.xtab12 = ctx->in_xtab12;
    p.ytab12 = ctx->in_ytab12;
    p.ow4_inv16 = (65536 + c.obs_w / 4 - 1) / (c.obs_w / 4);
    p.src_rows = 0;
    p.stamps = nullptr;
#ifdef AGX_STAMPS
    if (const char *e = getenv("AGX_DBG_PTR")) p.stamps = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0));
#endif
    return p;
}

static size_t ingest_lds(const agx_ctx *ctx) {
    return sizeof(int4) * ctx->band_rows + sizeof(int2) * ctx->cfg.obs_w + (size_t)2 * ctx->band_rows * 2 * kRawW;
}
// + 8 bytes of slack: phase 2 reads the two ALIGNED dwords around every tap pair, and for the last pair of a row (x0 = 158 at
// 160 -> 84) the second dword lies past the row's 320 bytes - past the allocation for the very last row (its value is shifted
// out, but the read must stay inside the workgroup's LDS)
static size_t band12_lds(const agx_ctx *ctx) { return sizeof(int2) * (kB12Rows + ctx->cfg.obs_w) + kB12GrayB + 8; }

int agx_ingest(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_frames || !d_cmd) return fail(ctx, AGX_E_INVALID, "agx_ingest: null buffer");
    const agx_config &c = ctx->cfg;
    if (c.obs_h != c.obs_w)
        return fail(ctx, AGX_E_INVALID,
                    "agx_ingest: obs_size (%d,%d) is not square; the reference hands obs_size to cv2.resize as "
                    "(width,height) and fails on non-square sizes (atari_env.py:74,126)", c.obs_h, c.obs_w);
    DeviceGuard g(c.device);
    const IngestParams p = ingest_params(ctx, d_frames, d_cmd);
    const int bands = p.nbands;
    const size_t lds = ingest_lds(ctx);
#ifdef AGX_EXPERIMENTS
    bool launched = true;
    const int pipe_parts = ctx->tune.pipe_parts;
    // wave-private form: needs the affine row form, band_rows = 4 * RPW with RPW * ow/4 <= 64 lanes and
    // 2 frames * RPW rows * 40 pieces <= 240 (RPW <= 3)
    // (measured equal to the barrier form at N=1024 - 46.5 vs 45.6 us - so it is opt-in: AGX_INGEST_WAVE=1)
    const bool want_wave = ctx->tune.wave != 0;
    const int rpw = ctx->band_rows / 4;
    const bool wave_ok = want_wave && pipe_parts == 0 && ctx->ingest_t == 256 && ctx->y_affine && ctx->band_rows % 4 == 0 &&
                         rpw >= 1 && rpw <= 3 && rpw * (c.obs_w / 4) <= 64;
    if (wave_ok) {
        const size_t slice = ((sizeof(int2) * c.obs_w + (size_t)2 * rpw * kRawW * 2) + 15) & ~(size_t)15;
        hipLaunchKernelGGL(k_ingest_wave, dim3(bands, c.num_envs), dim3(256), 4 * slice, S(stream), p);
    } else if (pipe_parts > 0 && ctx->ingest_t == 256) {
        const int parts = std::min(pipe_parts, bands);
        const size_t lds2 = sizeof(int4) * c.obs_h + sizeof(int2) * c.obs_w + (size_t)2 * (2 * ctx->band_rows * 2 * kRawW);
        hipLaunchKernelGGL(k_ingest_pipe<256>, dim3(parts, c.num_envs), dim3(256), lds2, S(stream), p);
    } else if (ctx->ingest_t == 128)
        hipLaunchKernelGGL(k_ingest<128>, dim3(bands, c.num_envs), dim3(128), lds, S(stream), p);
    // (same box, N=1024: 37.5-37.9 us against 37.9-38.2 for the one-band form - K1 is VALU-issue- and HBM-limited, not
    //  limited by the load-free tail of a workgroup - so it stays opt-in)
    else if (ctx->tune.no_full == 0 && ctx->tune.pair12 != 0 && ctx->tune.band_rows == 0 && ctx->tune.ingest_t == 0 &&
             ctx->y_affine && ctx->band_rows == 12 && c.obs_h % 12 == 0 && (c.obs_w / 4) * 12 <= kThreads)
        AGX_LAUNCH(0, k_ingest_pair12, dim3(bands, (c.num_envs + 1) / 2), dim3(256), lds + (size_t)2 * 12 * 2 * kRawW, S(stream), p,
                   (int)c.num_envs);
    else launched = false;
    if (!launched)
#endif
    {
        // the headline form where its plan applies (12-row bands all full, affine source rows, adjacent x taps), the general
        // band kernel otherwise
        if (ctx->tune.no_full == 0 && ctx->band12_ok && ctx->band_rows == 12)
            AGX_LAUNCH(0, k_ingest_full12, dim3(bands, c.num_envs), dim3(256), band12_lds(ctx), S(stream), p);
        else
            AGX_LAUNCH(0, k_ingest<256>, dim3(bands, c.num_envs), dim3(256), lds, S(stream), p);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    return AGX_OK;
}

int agx_ingest_gray_raw(agx_ctx *ctx, const uint8_t *d_gray, const uint8_t *d_cmd, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_gray || !d_cmd) return fail(ctx, AGX_E_INVALID, "agx_ingest_gray_raw: null buffer");
    const agx_config &c = ctx->cfg;
    if (c.obs_h != c.obs_w)
        return fail(ctx, AGX_E_STATE, "agx_ingest_gray_raw needs a square obs_size (cv2.resize takes (width, height): atari_env.py:74)");
    DeviceGuard g(c.device);
    IngestParams p = ingest_params(ctx, d_gray, d_cmd);
    // always the default 256-thread band form (the opt-in variants are RGB-only)
    const int br = std::max(1, std::min(2 * (kThreads / 40), kThreads / (c.obs_w / 4)));
    p.band_rows = std::min(br, ctx->band_rows > 0 && ctx->ingest_t == 256 ? ctx->band_rows : br);
    p.nbands = (c.obs_h + p.band_rows - 1) / p.band_rows;
    const size_t lds = sizeof(int4) * p.band_rows + sizeof(int2) * c.obs_w + (size_t)2 * p.band_rows * 2 * kRawW;
    if (ctx->tune.no_full == 0 && ctx->band12_ok && p.band_rows == 12)
        AGX_LAUNCH(0, k_ingest_grayraw_full12, dim3(p.nbands, c.num_envs), dim3(kThreads), band12_lds(ctx), S(stream), p);
    else
        AGX_LAUNCH(0, k_ingest_grayraw, dim3(p.nbands, c.num_envs), dim3(kThreads), lds, S(stream), p);
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    return AGX_OK;
}

int agx_source_rows(const agx_ctx *ctx, int32_t *rows, int32_t *n) {
    if (!ctx || !n) return AGX_E_INVALID;
    *n = (int32_t)ctx->src_rows.size();
    if (rows) std::copy(ctx->src_rows.begin(), ctx->src_rows.end(), rows);
    return AGX_OK;
}

// agx_ingest / agx_ingest_gray_raw from compact screens: the same band kernels with packed source rows
static int ingest_compact(agx_ctx *ctx, const uint8_t *d_rows, const uint8_t *d_cmd, void *stream, bool gray, const char *who) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_rows || !d_cmd) return fail(ctx, AGX_E_INVALID, "%s: null buffer", who);
    const agx_config &c = ctx->cfg;
    if (c.obs_h != c.obs_w)
        return fail(ctx, AGX_E_INVALID, "%s: obs_size (%d,%d) is not square (cv2.resize takes (width, height): atari_env.py:74)", who,
                    c.obs_h, c.obs_w);
    DeviceGuard g(c.device);
    IngestParams p = ingest_params(ctx, d_rows, d_cmd);
    p.src_rows = (int32_t)ctx->src_rows.size();
    p.ytab = ctx->in_ytab_c;             // packed row indices
    p.y_affine = 0;
    const int br = std::max(1, std::min(2 * (kThreads / 40), kThreads / (c.obs_w / 4)));
    p.band_rows = br;
    p.nbands = (c.obs_h + br - 1) / br;
    const size_t lds = sizeof(int4) * br + sizeof(int2) * c.obs_w + (size_t)2 * br * 2 * kRawW;
    const dim3 grid(p.nbands, c.num_envs), block(kThreads);
    if (ctx->tune.no_full == 0 && ctx->compact12_ok && br == 12) {
        if (gray) AGX_LAUNCH(0, k_ingest_grayraw_full12_compact, grid, block, band12_lds(ctx), S(stream), p);
        else AGX_LAUNCH(0, k_ingest_full12_compact, grid, block, band12_lds(ctx), S(stream), p);
    } else {
        if (gray) AGX_LAUNCH(0, k_ingest_grayraw_compact, grid, block, lds, S(stream), p);
        else AGX_LAUNCH(0, k_ingest_compact, grid, block, lds, S(stream), p);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    return AGX_OK;
}

int agx_ingest_compact(agx_ctx *ctx, const uint8_t *d_rows, const uint8_t *d_cmd, void *stream) {
    return ingest_compact(ctx, d_rows, d_cmd, stream, false, "agx_ingest_compact");
}
int agx_ingest_gray_raw_compact(agx_ctx *ctx, const uint8_t *d_rows, const uint8_t *d_cmd, void *stream) {
    return ingest_compact(ctx, d_rows, d_cmd, stream, true, "agx_ingest_gray_raw_compact");
}

int agx_ingest_gray(agx_ctx *ctx, const uint8_t *d_small, const uint8_t *d_cmd, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_small || !d_cmd) return fail(ctx, AGX_E_INVALID, "agx_ingest_gray: null buffer");
    const agx_config &c = ctx->cfg;
    DeviceGuard g(c.device);
    IngestGrayParams p;
    p.small = d_small;
    p.cmd = d_cmd;
    p.ring = ctx->ring;
    p.head_in = ctx->head[ctx->cur_head];
    p.head_out = ctx->head[ctx->cur_head ^ 1];
    p.oh = c.obs_h;
    p.ow = c.obs_w;
    p.fs = c.frame_stack;
    const int words = c.obs_h * c.obs_w / 4;
    hipLaunchKernelGGL(k_ingest_gray, dim3((words + kThreads - 1) / kThreads, c.num_envs), dim3(kThreads), 0,
                       S(stream), p);
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    return AGX_OK;
}

int agx_ingest_rgb(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, int gray_mode, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_frames || !d_cmd) return fail(ctx, AGX_E_INVALID, "agx_ingest_rgb: null buffer");
    const agx_config &c = ctx->cfg;
    DeviceGuard g(c.device);
    IngestRgbParams p;
    p.frames = d_frames;
    p.cmd = d_cmd;
    p.ring = ctx->ring;
    p.head_in = ctx->head[ctx->cur_head];
    p.head_out = ctx->head[ctx->cur_head ^ 1];
    p.oh = c.obs_h;
    p.ow = c.obs_w;
    p.fs = c.frame_stack;
    // cv2.cvtColor(rgb, COLOR_BGR2GRAY): channel 0 gets the blue weight (dmc_env.py:181-182 hands it an RGB render)
    switch (gray_mode) {
        case AGX_GRAY_CV15: p.k0 = 3735, p.k1 = 19235, p.k2 = 9798, p.shift = 15; break;   // OpenCV 4.x: BY15 GY15 RY15
        case AGX_GRAY_CV14: p.k0 = 1868, p.k1 = 9617, p.k2 = 4899, p.shift = 14; break;    // OpenCV <= 3.x: B2Y G2Y R2Y
        default: return fail(ctx, AGX_E_INVALID, "agx_ingest_rgb: unknown gray_mode %d", gray_mode);
    }
    const int words = c.obs_h * c.obs_w / 4;
    AGX_LAUNCH(0, k_ingest_rgb, dim3((words + kThreads - 1) / kThreads, c.num_envs), dim3(kThreads), 0, S(stream), p);
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    return AGX_OK;
}

// ---------------------------------------------------------------- K0
static int stack_launch(agx_ctx *ctx, int which, const uint8_t *in_u8, uint8_t *out_u8, float *out_f32, void *stream) {
    const agx_config &c = ctx->cfg;
    DeviceGuard g(c.device);
    StackParams p;
    p.ring = ctx->ring;
    p.head = ctx->head[ctx->cur_head];
    p.in_u8 = in_u8;
    p.out_u8 = out_u8;
    p.out_f32 = out_f32;
    p.words = c.obs_h * c.obs_w / 4;
    p.fs = c.frame_stack;
    const dim3 grid((p.words + kThreads - 1) / kThreads, c.frame_stack, c.num_envs);
    if (which == 0)
        hipLaunchKernelGGL(k_stack_u8, grid, dim3(kThreads), 0, S(stream), p);
    else if (which == 1)
        hipLaunchKernelGGL(k_set_stack, grid, dim3(kThreads), 0, S(stream), p);
    else
        hipLaunchKernelGGL(k_full, grid, dim3(kThreads), 0, S(stream), p);
    AGX_HIP(ctx, hipGetLastError());
    return AGX_OK;
}

int agx_observe_full(agx_ctx *ctx, float *d_obs, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_obs) return fail(ctx, AGX_E_INVALID, "agx_observe_full: null buffer");
    return stack_launch(ctx, 2, nullptr, nullptr, d_obs, stream);
}
int agx_get_stack_u8(agx_ctx *ctx, uint8_t *d_out, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_out) return fail(ctx, AGX_E_INVALID, "agx_get_stack_u8: null buffer");
    return stack_launch(ctx, 0, nullptr, d_out, nullptr, stream);
}
int agx_set_stack_u8(agx_ctx *ctx, const uint8_t *d_in, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    if (!d_in) return fail(ctx, AGX_E_INVALID, "agx_set_stack_u8: null buffer");
    return stack_launch(ctx, 1, d_in, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------- fovea state
int agx_fovea_reset(agx_ctx *ctx, const uint8_t *d_mask, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (!has_fovea(c)) return fail(ctx, AGX_E_STATE, "agx_fovea_reset: context has no fovea (AGX_KIND_BASE)");
    DeviceGuard g(c.device);
    FovResetParams p;
    p.mask = d_mask;
    p.loc = ctx->loc[ctx->cur_fov];
    p.res = ctx->res[ctx->cur_fov];
    p.init_r = ctx->init_r;
    p.init_c = ctx->init_c;
    p.fh = c.fov_h;
    p.fw = c.fov_w;
    p.n = c.num_envs;
    hipLaunchKernelGGL(k_fovea_reset, dim3((c.num_envs + kThreads - 1) / kThreads), dim3(kThreads), 0, S(stream), p);
    AGX_HIP(ctx, hipGetLastError());
    return AGX_OK;
}

int agx_get_fov_state(agx_ctx *ctx, int32_t *d_fov_loc, int32_t *d_fov_res, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (!has_fovea(c)) return fail(ctx, AGX_E_STATE, "agx_get_fov_state: context has no fovea");
    DeviceGuard g(c.device);
    const size_t b = (size_t)c.num_envs * 2 * sizeof(int32_t);
    if (d_fov_loc) AGX_HIP(ctx, hipMemcpyAsync(d_fov_loc, ctx->loc[ctx->cur_fov], b, hipMemcpyDeviceToDevice, S(stream)));
    if (d_fov_res) AGX_HIP(ctx, hipMemcpyAsync(d_fov_res, ctx->res[ctx->cur_fov], b, hipMemcpyDeviceToDevice, S(stream)));
    return AGX_OK;
}

int agx_set_fov_state(agx_ctx *ctx, const int32_t *d_fov_loc, const int32_t *d_fov_res, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (!has_fovea(c)) return fail(ctx, AGX_E_STATE, "agx_set_fov_state: context has no fovea");
    DeviceGuard g(c.device);
    const size_t b = (size_t)c.num_envs * 2 * sizeof(int32_t);
    if (d_fov_loc) AGX_HIP(ctx, hipMemcpyAsync(ctx->loc[ctx->cur_fov], d_fov_loc, b, hipMemcpyDeviceToDevice, S(stream)));
    if (d_fov_res) AGX_HIP(ctx, hipMemcpyAsync(ctx->res[ctx->cur_fov], d_fov_res, b, hipMemcpyDeviceToDevice, S(stream)));
    return AGX_OK;
}

// ---------------------------------------------------------------- K2/K3/K4
static int check_dt(agx_ctx *ctx, const void *d_action, int dt) {
    if (d_action && (dt < AGX_DT_F32 || dt > AGX_DT_I64)) return fail(ctx, AGX_E_INVALID, "unknown action dtype %d", dt);
    return AGX_OK;
}

static FovParams fov_params(agx_ctx *ctx, const void *d_action, int dt, const int32_t *d_type, const uint8_t *d_mask,
                            float *d_obs, int32_t *d_loc, int32_t *d_res) {
    const agx_config &c = ctx->cfg;
    FovParams p;
    p.ring = ctx->ring;
    p.head = ctx->head[ctx->cur_head];
    p.loc_in = ctx->loc[ctx->cur_fov];
    p.loc_out = ctx->loc[ctx->cur_fov ^ 1];
    p.res_in = ctx->res[ctx->cur_fov];
    p.res_out = ctx->res[ctx->cur_fov ^ 1];
    p.action = d_action;
    p.action_type = d_type;
    p.mask = d_mask;
    p.obs = d_obs;
    p.user_loc = d_loc;
    p.user_res = d_res;
    p.xtab = ctx->fx_xtab;
    p.ytab = ctx->fx_ytab;
    p.sas_lo = c.sas_lo;
    p.sas_hi = c.sas_hi;
    p.action_dt = dt;
    p.relative = c.action_mode == AGX_MODE_RELATIVE;
    p.fs = c.frame_stack;
    p.out_mode = c.out_mode;
    p.antialias = c.antialias != 0;
    p.per_h = c.per_h;
    p.per_w = c.per_w;
    p.buf1_floats = (int32_t)generic_buf1(c);
    p.cmd = nullptr;
    p.phase = 0;
    p.packed = nullptr;
    p.packed_off = nullptr;
    p.packed_cap = 0;
    p.stamps = nullptr;
#ifdef AGX_STAMPS
    if (const char *e = getenv("AGX_DBG_PTR2")) p.stamps = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0));
#endif
    return p;
}

int agx_fovea_fixed(agx_ctx *ctx, const void *d_action, int action_dtype, const uint8_t *d_mask, float *d_obs,
                    int32_t *d_fov_loc, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_FIXED) return fail(ctx, AGX_E_STATE, "agx_fovea_fixed on a context of kind %d", c.kind);
    if (!d_obs) return fail(ctx, AGX_E_INVALID, "agx_fovea_fixed: null obs buffer");
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    DeviceGuard g(c.device);
    const FovParams p = fov_params(ctx, d_action, action_dtype, nullptr, d_mask, d_obs, d_fov_loc, nullptr);
    const size_t lds = fixed_lds(c);
    const bool headline = c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30;
    using GS = GeomS<84, 84, 30, 30>;
    const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
    const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
#define LAUNCH(MODE)                                                                                  \
    do {                                                                                              \
        if (headline)                                                                                 \
            AGX_LAUNCH(1, (k_fovea_fixed<GS, MODE>), grid, block, lds, S(stream), GS{}, p);      \
        else                                                                                          \
            AGX_LAUNCH(1, (k_fovea_fixed<GeomR, MODE>), grid, block, lds, S(stream), gr, p);     \
    } while (0)
#ifdef AGX_EXPERIMENTS
    // two physical slots per workgroup (whole launch resident at once, second frame's load hidden): measured a tie
    // with the one-slot form at N=1024 (26.3 vs 25.8 us) - the launch is store-limited
    if (c.out_mode == AGX_OUT_RESIZE && c.frame_stack % 2 == 0 && ctx->tune.pair == 1) {
        const dim3 grid2(c.frame_stack / 2, c.num_envs);
        if (headline)
            hipLaunchKernelGGL((k_fovea_fixed2<GS>), grid2, block, fixed2_lds(c), S(stream), GS{}, p);
        else
            hipLaunchKernelGGL((k_fovea_fixed2<GeomR>), grid2, block, fixed2_lds(c), S(stream), gr, p);
    } else
#endif
        switch (c.out_mode) {
            case AGX_OUT_RAW: LAUNCH(AGX_OUT_RAW); break;
            case AGX_OUT_MASK: LAUNCH(AGX_OUT_MASK); break;
            default: LAUNCH(AGX_OUT_RESIZE); break;
        }
#undef LAUNCH
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_fov ^= 1;
    return AGX_OK;
}

// ---------------------------------------------------------------- fused step (K1 + K2)
int agx_step_fixed(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, const void *d_action, int action_dtype,
                   float *d_obs, int32_t *d_fov_loc, void *mid_event, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_FIXED) return fail(ctx, AGX_E_STATE, "agx_step_fixed on a context of kind %d", c.kind);
    if (!d_frames || !d_cmd || !d_obs) return fail(ctx, AGX_E_INVALID, "agx_step_fixed: null buffer");
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    // The step as the library ships it: the two stand-alone launches (ingest, then fovea), the fastest form measured.  The
    // other forms of this call that were built and measured slower or equal (heterogeneous fused launch + tail, env-range
    // parts on internal streams, one workgroup per env) live in the experiments build only.
#ifdef AGX_EXPERIMENTS
    const bool headline = c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30;
    // The heterogeneous launch (ingest bands + fovea of the untouched slots in one grid, written slot after) is
    // bit-identical and measured a tie at N=1024 (69.3 vs 67.9 us per step: it fills the ingest's drain but its
    // second launch is one latency chain long), so the default is the two stand-alone launches.
    const bool fused = ctx->tune.fused != 0;                                    // tuning / testing knob
    // ---- split step: the batch as P env ranges, range 0 on the caller's stream, the others on internal streams forked
    // from it and joined back before the call returns its work to the caller's stream order.  One range's fovea stores
    // and ingest drain then run under another range's ingest loads (reads and writes of the same step overlap), with
    // the results of one launch pair bit for bit (same kernels, disjoint env ranges, no shared state).
    const agx_ctx::Tune &tn = ctx->tune;
    const bool default_forms = !fused && !tn.ingest_t && !tn.band_rows && !tn.pipe_parts && !tn.wave && !tn.pair && !tn.no_full;
    // Measured at N=1024 (same box, bench.py --steps 600): one launch pair 60.9 us per step; 2 parts 72.7 (69.3 with
    // low-priority internal streams, 75.1 with high), 3 parts 86.6, 4 parts 105: every cross-stream event edge costs more
    // than the overlap returns (round 1's +6-10 % came from two independent contexts that never join).  So it is opt-in.
    int parts = tn.split > 0 ? tn.split : 1;
    parts = std::min(std::min(parts, 4), c.num_envs);
    // ---- one launch, one workgroup per env (agx_step_env.h): the headline geometry's resize_to_full path
    if (tn.step_env != 0 && default_forms && !mid_event && c.out_mode == AGX_OUT_RESIZE && c.obs_h == 84 && c.obs_w == 84 &&
        c.fov_h == 30 && c.fov_w == 30 && ctx->y_affine && ctx->band_rows == 12 && c.frame_stack >= 1) {
        DeviceGuard g(c.device);
        const IngestParams pi = ingest_params(ctx, d_frames, d_cmd);
        FovParams pf = fov_params(ctx, d_action, action_dtype, nullptr, nullptr, d_obs, d_fov_loc, nullptr);
        pf.cmd = d_cmd;
        pf.phase = 3;                                        // `head` is the pre-ingest head, every slot is processed
        pf.head = ctx->head[ctx->cur_head];
        const size_t team_lds = (std::max(ingest_lds(ctx), fixed_lds(c)) + 15) & ~(size_t)15;
        using GS = GeomS<84, 84, 30, 30>;
        StepEnvArgs sa;
        sa.pi = pi;
        sa.pf = pf;
        sa.team_lds = (int32_t)team_lds;
        sa.debug = env_int("AGX_STEP_ENV_DEBUG", 0);
        hipLaunchKernelGGL((k_step_env<GS>), dim3(c.num_envs), dim3(2 * kThreads), 2 * team_lds, S(stream), sa);
        AGX_HIP(ctx, hipGetLastError());
        ctx->cur_head ^= 1;
        ctx->cur_fov ^= 1;
        return AGX_OK;
    }
    if (parts > 1 && default_forms && c.obs_h == c.obs_w && !mid_event) {
        DeviceGuard g(c.device);
        if (!ctx->ev_fork) {
            int lo_p = 0, hi_p = 0;
            AGX_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));        // lo_p: least urgent (largest number)
            const int prio = tn.aux_prio > 0 ? hi_p : (tn.aux_prio < 0 ? lo_p : 0);
            for (int k = 0; k < 3; ++k) {
                AGX_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux[k], hipStreamNonBlocking, prio));
                AGX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
            }
            AGX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        }
        hipStream_t st[4] = {S(stream), ctx->aux[0], ctx->aux[1], ctx->aux[2]};
        AGX_HIP(ctx, hipEventRecord(ctx->ev_fork, st[0]));
        for (int k = 1; k < parts; ++k) AGX_HIP(ctx, hipStreamWaitEvent(st[k], ctx->ev_fork, 0));
        const size_t fsz = (size_t)c.obs_h * c.obs_w;
        const bool crop = c.out_mode == AGX_OUT_RAW;
        const size_t obs_env = (size_t)c.frame_stack * (crop ? (size_t)c.fov_h * c.fov_w : fsz);
        const bool wide = action_dtype == AGX_DT_F64 || action_dtype == AGX_DT_I64;
        const bool full12 = ctx->y_affine && ctx->band_rows == 12 && c.obs_h % 12 == 0;
        const bool headline = c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30;
        using GS = GeomS<84, 84, 30, 30>;
        const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
        int n0[5];
        for (int k = 0; k <= parts; ++k) n0[k] = (int)((long long)c.num_envs * k / parts);
        const IngestParams pi0 = ingest_params(ctx, d_frames, d_cmd);
        for (int k = 0; k < parts; ++k) {
            IngestParams q = pi0;
            const size_t b = n0[k];
            q.frames += b * 2 * (size_t)kRawFrameBytes;
            q.cmd += b;
            q.ring += b * c.frame_stack * fsz;
            q.head_in += b;
            q.head_out += b;
            const dim3 grid(q.nbands, n0[k + 1] - n0[k]);
            if (full12) hipLaunchKernelGGL(k_ingest_full12_part, grid, dim3(kThreads), ingest_lds(ctx), st[k], q);
            else hipLaunchKernelGGL(k_ingest_part, grid, dim3(kThreads), ingest_lds(ctx), st[k], q);
        }
        AGX_HIP(ctx, hipGetLastError());
        ctx->cur_head ^= 1;
        const FovParams pf0 = fov_params(ctx, d_action, action_dtype, nullptr, nullptr, d_obs, d_fov_loc, nullptr);
        for (int k = 0; k < parts; ++k) {
            FovParams q = pf0;
            const size_t b = n0[k];
            q.ring += b * c.frame_stack * fsz;
            q.head += b;
            q.loc_in += 2 * b;
            q.loc_out += 2 * b;
            q.res_in += 2 * b;
            q.res_out += 2 * b;
            if (q.action) q.action = static_cast<const char *>(q.action) + b * (wide ? 16 : 8);
            q.obs += b * obs_env;
            if (q.user_loc) q.user_loc += 2 * b;
            const dim3 grid(c.frame_stack, n0[k + 1] - n0[k]);
            const size_t lds = fixed_lds(c);
#define LAUNCH_PART(MODE)                                                                                              \
    do {                                                                                                               \
        if (headline) hipLaunchKernelGGL((k_fovea_fixed_part<GS, MODE>), grid, dim3(kThreads), lds, st[k], GS{}, q);   \
        else hipLaunchKernelGGL((k_fovea_fixed_part<GeomR, MODE>), grid, dim3(kThreads), lds, st[k], gr, q);           \
    } while (0)
            switch (c.out_mode) {
                case AGX_OUT_RAW: LAUNCH_PART(AGX_OUT_RAW); break;
                case AGX_OUT_MASK: LAUNCH_PART(AGX_OUT_MASK); break;
                default: LAUNCH_PART(AGX_OUT_RESIZE); break;
            }
#undef LAUNCH_PART
        }
        AGX_HIP(ctx, hipGetLastError());
        ctx->cur_fov ^= 1;
        for (int k = 1; k < parts; ++k) {
            AGX_HIP(ctx, hipEventRecord(ctx->ev_join[k - 1], st[k]));
            AGX_HIP(ctx, hipStreamWaitEvent(st[0], ctx->ev_join[k - 1], 0));
        }
        return AGX_OK;
    }
    if (!(c.out_mode != AGX_OUT_RESIZE || ctx->ingest_t != 256 || !fused || c.obs_h != c.obs_w)) {
        DeviceGuard g(c.device);
        const IngestParams pi = ingest_params(ctx, d_frames, d_cmd);
        FovParams pf = fov_params(ctx, d_action, action_dtype, nullptr, nullptr, d_obs, d_fov_loc, nullptr);
        pf.cmd = d_cmd;
        pf.phase = 1;
        pf.head = ctx->head[ctx->cur_head];                  // the head BEFORE this step's ingest
        const bool b12 = ctx->tune.fused >= 2 && ctx->band12_ok && ctx->band_rows == 12;      // AGX_STEP_FUSED=2 / 3: band12 ingest body
        const size_t lds = std::max(b12 ? band12_lds(ctx) : ingest_lds(ctx), fixed_lds(c));
        const dim3 grid1(pi.nbands + c.frame_stack, c.num_envs), grid2(1, c.num_envs), block(kThreads);
        using GS = GeomS<84, 84, 30, 30>;
        const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
        if (b12 && headline && ctx->tune.fused == 3)
            hipLaunchKernelGGL((k_step_fixed12_ff<GS>), grid1, block, lds, S(stream), GS{}, pi, pf);
        else if (b12 && headline)
            hipLaunchKernelGGL((k_step_fixed12<GS>), grid1, block, lds, S(stream), GS{}, pi, pf);
        else if (headline)
            hipLaunchKernelGGL((k_step_fixed<GS>), grid1, block, lds, S(stream), GS{}, pi, pf);
        else
            hipLaunchKernelGGL((k_step_fixed<GeomR>), grid1, block, lds, S(stream), gr, pi, pf);
        AGX_HIP(ctx, hipGetLastError());
        ctx->cur_head ^= 1;
        if (mid_event) AGX_HIP(ctx, hipEventRecord(static_cast<hipEvent_t>(mid_event), S(stream)));
        pf.phase = 2;
        pf.head = ctx->head[ctx->cur_head];                  // the head AFTER the ingest
        if (headline)
            hipLaunchKernelGGL((k_step_fixed_tail<GS>), grid2, block, fixed_lds(c), S(stream), GS{}, pf);
        else
            hipLaunchKernelGGL((k_step_fixed_tail<GeomR>), grid2, block, fixed_lds(c), S(stream), gr, pf);
        AGX_HIP(ctx, hipGetLastError());
        ctx->cur_fov ^= 1;
        return AGX_OK;
    }
#endif
    rc = agx_ingest(ctx, d_frames, d_cmd, stream);
    if (rc) return rc;
    if (mid_event) AGX_HIP(ctx, hipEventRecord(static_cast<hipEvent_t>(mid_event), S(stream)));
    return agx_fovea_fixed(ctx, d_action, action_dtype, nullptr, d_obs, d_fov_loc, stream);
}

int agx_fovea_peripheral(agx_ctx *ctx, const void *d_action, int action_dtype, const uint8_t *d_mask, float *d_obs,
                         int32_t *d_fov_loc, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_PERIPHERAL) return fail(ctx, AGX_E_STATE, "agx_fovea_peripheral on a context of kind %d", c.kind);
    if (!d_obs) return fail(ctx, AGX_E_INVALID, "agx_fovea_peripheral: null obs buffer");
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    DeviceGuard g(c.device);
    const FovParams p = fov_params(ctx, d_action, action_dtype, nullptr, d_mask, d_obs, d_fov_loc, nullptr);
    const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
    const bool generic_only = ctx->tune.generic != 0;                               // tuning / testing knob
    if (!generic_only && ctx->p3_mt && ctx->tune.per_v2 == 0) {
        const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
        const size_t lds = ctx->p3_lds;
        using GS = PGeomS<84, 84, 30, 30, 20, 20>;
        const PGeomR pg{c.obs_h, c.obs_w, c.fov_h, c.fov_w, c.per_h, c.per_w};
        const bool headline = c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30 && c.per_h == 20 && c.per_w == 20;
        if (headline && ctx->p3_mt == 12) AGX_LAUNCH(1, (k_fovea_peripheral3<GS, 12>), grid, block, lds, S(stream), GS{}, ctx->p3, p);
        else if (headline && ctx->p3_mt == 4) AGX_LAUNCH(1, (k_fovea_peripheral3<GS, 4>), grid, block, lds, S(stream), GS{}, ctx->p3, p);
        else if (ctx->p3_mt == 4) AGX_LAUNCH(1, (k_fovea_peripheral3<PGeomR, 4>), grid, block, lds, S(stream), pg, ctx->p3, p);
        else if (ctx->p3_mt == 8) AGX_LAUNCH(1, (k_fovea_peripheral3<PGeomR, 8>), grid, block, lds, S(stream), pg, ctx->p3, p);
        else if (ctx->p3_mt == 12) AGX_LAUNCH(1, (k_fovea_peripheral3<PGeomR, 12>), grid, block, lds, S(stream), pg, ctx->p3, p);
        else AGX_LAUNCH(1, (k_fovea_peripheral3<PGeomR, 16>), grid, block, lds, S(stream), pg, ctx->p3, p);
    } else
    // the tuned kernel keeps A | B | C with C 16-byte aligned and one row sweep per 256 threads
    if (!generic_only && per2_lds(c) <= kMaxLds && c.per_w <= kThreads) {
        PerParams g;
        for (int k = 0; k < 4; ++k) {
            g.t[k].ln = ctx->per_ln[k];
            g.t[k].w = ctx->per_w[k];
            g.t[k].maxt = ctx->per_maxt[k];
        }
        g.t[0].n_out = c.per_w; g.t[1].n_out = c.per_h; g.t[2].n_out = c.obs_w; g.t[3].n_out = c.obs_h;
        g.oh = c.obs_h; g.ow = c.obs_w; g.fh = c.fov_h; g.fw = c.fov_w; g.ph = c.per_h; g.pw = c.per_w;
        g.same = (c.per_h == c.obs_h && c.per_w == c.obs_w) ? 1 : 0;                // torchvision returns the input
        const int mt = std::max(ctx->per_maxt[0], ctx->per_maxt[1]);
        const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
        const size_t lds = per2_lds(c);
        // both squeeze tables are padded to their own bucket; the kernel bound must not exceed either row pitch
        const bool same_bucket = ctx->per_maxt[0] == ctx->per_maxt[1];
        if (same_bucket && mt == 2) AGX_LAUNCH(1, k_fovea_peripheral2<2>, grid, block, lds, S(stream), g, p);
        else if (same_bucket && mt == 4) AGX_LAUNCH(1, k_fovea_peripheral2<4>, grid, block, lds, S(stream), g, p);
        else if (same_bucket && mt == 8) AGX_LAUNCH(1, k_fovea_peripheral2<8>, grid, block, lds, S(stream), g, p);
        else if (same_bucket && mt == 12) AGX_LAUNCH(1, k_fovea_peripheral2<12>, grid, block, lds, S(stream), g, p);
        else if (same_bucket && mt == 16) AGX_LAUNCH(1, k_fovea_peripheral2<16>, grid, block, lds, S(stream), g, p);
        else AGX_LAUNCH(1, k_fovea_peripheral2<0>, grid, block, lds, S(stream), g, p);
    } else {
        hipLaunchKernelGGL((k_fovea_generic<AGX_KIND_PERIPHERAL>), dim3(c.frame_stack, c.num_envs), dim3(kThreads),
                           generic_lds(c), S(stream), gr, p);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_fov ^= 1;
    return AGX_OK;
}

int agx_fovea_flexible(agx_ctx *ctx, const void *d_action, int action_dtype, const int32_t *d_action_type,
                       const uint8_t *d_mask, float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_FLEXIBLE) return fail(ctx, AGX_E_STATE, "agx_fovea_flexible on a context of kind %d", c.kind);
    if (!d_obs) return fail(ctx, AGX_E_INVALID, "agx_fovea_flexible: null obs buffer");
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    DeviceGuard g(c.device);
    const FovParams p = fov_params(ctx, d_action, action_dtype, d_action_type, d_mask, d_obs, d_fov_loc, d_fov_res);
    const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
    const bool generic_only = ctx->tune.generic != 0;                               // tuning / testing knob
    const size_t lds2 = flex2_lds(c, ctx->flex_tab_floats);
    if (!generic_only && ctx->f3_ok && ctx->tune.flex_v2 == 0) {
        const size_t lds3 = (size_t)ctx->f3.r0_bytes + ctx->f3.r1_bytes + (size_t)c.obs_h * sizeof(int4);
        const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
        using GS = GeomS<84, 84, 30, 30>;
        if (c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30)
            AGX_LAUNCH(1, (k_fovea_flexible3<GS>), grid, block, lds3, S(stream), GS{}, ctx->f3, p);
        else
            AGX_LAUNCH(1, (k_fovea_flexible3<GeomR>), grid, block, lds3, S(stream), gr, ctx->f3, p);
    } else if (!generic_only && ctx->fr_ok && ctx->tune.flex_v2 == 0 && c.out_mode != AGX_OUT_RESIZE) {
        const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
        using GS = GeomS<84, 84, 30, 30>;
        const bool headline = c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30;
        if (c.out_mode == AGX_OUT_MASK) {
            if (headline) AGX_LAUNCH(1, (k_fovea_flexible_raw3<GS, AGX_OUT_MASK>), grid, block, ctx->fr_lds, S(stream), GS{}, ctx->fr, p);
            else AGX_LAUNCH(1, (k_fovea_flexible_raw3<GeomR, AGX_OUT_MASK>), grid, block, ctx->fr_lds, S(stream), gr, ctx->fr, p);
        } else {
            if (headline) AGX_LAUNCH(1, (k_fovea_flexible_raw3<GS, AGX_OUT_RAW>), grid, block, ctx->fr_lds, S(stream), GS{}, ctx->fr, p);
            else AGX_LAUNCH(1, (k_fovea_flexible_raw3<GeomR, AGX_OUT_RAW>), grid, block, ctx->fr_lds, S(stream), gr, ctx->fr, p);
        }
    } else if (!generic_only && lds2 <= kMaxLds) {
        FlexParams g;
        TabFamily *fam[6] = {&g.wd, &g.wb, &g.wf, &g.hd, &g.hb, &g.hf};
        for (int k = 0; k < 6; ++k) {
            fam[k]->ln = ctx->flex_ln[k];
            fam[k]->w = ctx->flex_w[k];
            fam[k]->meta = ctx->flex_meta[k];
        }
        g.oh = c.obs_h; g.ow = c.obs_w; g.fh = c.fov_h; g.fw = c.fov_w;
        if (c.out_mode == AGX_OUT_RESIZE) AGX_LAUNCH(1, k_fovea_flexible2<true>, dim3(c.frame_stack, c.num_envs), dim3(kThreads), lds2, S(stream), g, p);
        else AGX_LAUNCH(1, k_fovea_flexible2<false>, dim3(c.frame_stack, c.num_envs), dim3(kThreads), lds2, S(stream), g, p);
    } else {
        hipLaunchKernelGGL((k_fovea_generic<AGX_KIND_FLEXIBLE>), dim3(c.frame_stack, c.num_envs), dim3(kThreads),
                           generic_lds(c), S(stream), gr, p);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_fov ^= 1;
    return AGX_OK;
}

// the state / scan launch of the packed form (fov_env.py:300-324 + level 1 of the exclusive scan of the crop sizes; the scratch
// belongs to the context since agx_create)
static FlexScanParams packed_scan_params(agx_ctx *ctx, const void *d_action, int action_dtype, const int32_t *d_action_type,
                                         int32_t *d_fov_loc, int32_t *d_fov_res) {
    FlexScanParams q;
    q.f = fov_params(ctx, d_action, action_dtype, d_action_type, nullptr, nullptr, d_fov_loc, d_fov_res);
    q.local_off = ctx->pack_local;
    q.block_tot = ctx->pack_block;
    q.n = ctx->cfg.num_envs;
    q.oh = ctx->cfg.obs_h;
    q.ow = ctx->cfg.obs_w;
    return q;
}
static bool packed_raw3_ok(const agx_ctx *ctx) { return ctx->fr_ok && ctx->tune.generic == 0 && ctx->tune.flex_v2 == 0; }

// the crop launch of the packed form on the raw3 plan (the state is final: cur_fov has been flipped by the caller)
static int packed_crops_raw3(agx_ctx *ctx, float *d_packed, int64_t capacity_floats, int64_t *d_offsets, void *stream) {
    const agx_config &c = ctx->cfg;
    FovParams p = fov_params(ctx, nullptr, 0, nullptr, nullptr, d_packed, nullptr, nullptr);
    p.packed = d_packed;
    p.packed_off = d_offsets;
    p.packed_cap = capacity_floats;
    const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
    FlexRawParams fr = ctx->fr;
    fr.local_off = ctx->pack_local;
    fr.block_tot = ctx->pack_block;
    fr.offsets = d_offsets;
    const dim3 grid(c.frame_stack, c.num_envs), block(kThreads);
    using GS = GeomS<84, 84, 30, 30>;
#ifdef AGX_EXPERIMENTS
    if (ctx->tune.packed_wave != 0) {              // one wave per (slot, env) item: measured slower (docs/HISTORY.md, round 4)
        if (c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30)
            AGX_LAUNCH(1, (k_fovea_flexible_raw3_wave<GS>), grid, dim3(64), ctx->fr_lds, S(stream), GS{}, fr, p);
        else
            AGX_LAUNCH(1, (k_fovea_flexible_raw3_wave<GeomR>), grid, dim3(64), ctx->fr_lds, S(stream), gr, fr, p);
    } else
#endif
    if (c.obs_h == 84 && c.obs_w == 84 && c.fov_h == 30 && c.fov_w == 30)
        AGX_LAUNCH(1, (k_fovea_flexible_raw3<GS, kRawPacked>), grid, block, ctx->fr_lds, S(stream), GS{}, fr, p);
    else
        AGX_LAUNCH(1, (k_fovea_flexible_raw3<GeomR, kRawPacked>), grid, block, ctx->fr_lds, S(stream), gr, fr, p);
    AGX_HIP(ctx, hipGetLastError());
    return AGX_OK;
}

int agx_fovea_flexible_packed(agx_ctx *ctx, const void *d_action, int action_dtype, const int32_t *d_action_type,
                              float *d_packed, int64_t capacity_floats, int64_t *d_offsets, int32_t *d_fov_loc,
                              int32_t *d_fov_res, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_FLEXIBLE || c.out_mode != AGX_OUT_RAW)
        return fail(ctx, AGX_E_STATE, "agx_fovea_flexible_packed needs a flexible context in raw-crop mode (kind %d, out_mode %d)",
                    c.kind, c.out_mode);
    if (!d_packed || !d_offsets || capacity_floats < 0) return fail(ctx, AGX_E_INVALID, "agx_fovea_flexible_packed: null buffer");
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    DeviceGuard g(c.device);
    // launch 1: every env's new fov_loc / fov_res (fov_env.py:300-324), its crop size, and level 1 of the exclusive scan
    // (block-local offsets + block totals)
    const FlexScanParams q = packed_scan_params(ctx, d_action, action_dtype, d_action_type, d_fov_loc, d_fov_res);
    const int nb = (c.num_envs + kScanEnvsPerBlock - 1) / kScanEnvsPerBlock;
    hipLaunchKernelGGL(k_flex_state_scan, dim3(nb), dim3(kThreads), 0, S(stream), q);
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_fov ^= 1;                   // the state is final from here on; the crop launch only reads it
    // launch 2: the crops (squeezed to fov_size and back iff rows > fov rows, fov_env.py:283-287) at their offsets
    if (packed_raw3_ok(ctx)) return packed_crops_raw3(ctx, d_packed, capacity_floats, d_offsets, stream);
    FovParams p = fov_params(ctx, nullptr, 0, nullptr, nullptr, d_packed, nullptr, nullptr);
    p.packed = d_packed;
    p.packed_off = d_offsets;
    p.packed_cap = capacity_floats;
    const GeomR gr{c.obs_h, c.obs_w, c.fov_h, c.fov_w};
    // geometries outside the raw3 plan: offsets as a launch of their own, then the pass-by-pass crop kernel, which writes
    // the (unchanged) state through into the other half of the double buffer
    hipLaunchKernelGGL(k_flex_finish_offsets, dim3((c.num_envs + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, S(stream),
                       ctx->pack_local, ctx->pack_block, d_offsets, (int)c.num_envs);
    const size_t lds2 = flex2_lds(c, ctx->flex_tab_floats);
    if (ctx->tune.generic == 0 && lds2 <= kMaxLds) {
        FlexParams fp;
        TabFamily *fam[6] = {&fp.wd, &fp.wb, &fp.wf, &fp.hd, &fp.hb, &fp.hf};
        for (int k = 0; k < 6; ++k) {
            fam[k]->ln = ctx->flex_ln[k];
            fam[k]->w = ctx->flex_w[k];
            fam[k]->meta = ctx->flex_meta[k];
        }
        fp.oh = c.obs_h; fp.ow = c.obs_w; fp.fh = c.fov_h; fp.fw = c.fov_w;
        AGX_LAUNCH(1, k_fovea_flexible2<false>, dim3(c.frame_stack, c.num_envs), dim3(kThreads), lds2, S(stream), fp, p);
    } else {
        hipLaunchKernelGGL((k_fovea_generic<AGX_KIND_FLEXIBLE>), dim3(c.frame_stack, c.num_envs), dim3(kThreads),
                           generic_lds(c), S(stream), gr, p);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_fov ^= 1;
    return AGX_OK;
}

int agx_step_flexible_packed(agx_ctx *ctx, const uint8_t *d_screens, int screens, const uint8_t *d_cmd, const void *d_action,
                             int action_dtype, const int32_t *d_action_type, float *d_packed, int64_t capacity_floats,
                             int64_t *d_offsets, int32_t *d_fov_loc, int32_t *d_fov_res, void *stream) {
    if (!ctx) return AGX_E_INVALID;
    const agx_config &c = ctx->cfg;
    if (c.kind != AGX_KIND_FLEXIBLE || c.out_mode != AGX_OUT_RAW)
        return fail(ctx, AGX_E_STATE, "agx_step_flexible_packed needs a flexible context in raw-crop mode (kind %d, out_mode %d)",
                    c.kind, c.out_mode);
    if (!d_screens || !d_cmd || !d_packed || !d_offsets || capacity_floats < 0)
        return fail(ctx, AGX_E_INVALID, "agx_step_flexible_packed: null buffer");
    if (screens & ~(AGX_SCREENS_GRAY | AGX_SCREENS_COMPACT))
        return fail(ctx, AGX_E_INVALID, "agx_step_flexible_packed: unknown screen layout bits 0x%x", screens);
    int rc = check_dt(ctx, d_action, action_dtype);
    if (rc) return rc;
    const bool gray = (screens & AGX_SCREENS_GRAY) != 0, compact = (screens & AGX_SCREENS_COMPACT) != 0;
    // the two-launch form needs the band12 ingest plan for this layout and the raw3 crop plan; everything else (and
    // AGX_STEP_PACKED_UNFUSED=1, for A/B runs) is the three launches of the stand-alone entry points, same results
    const int br_def = std::max(1, std::min(2 * (kThreads / 40), kThreads / std::max(1, c.obs_w / 4)));
    bool band12;
    if (compact) band12 = ctx->compact12_ok && br_def == 12;
    else if (gray) band12 = ctx->band12_ok && std::min(br_def, ctx->band_rows > 0 && ctx->ingest_t == 256 ? ctx->band_rows : br_def) == 12;
    else band12 = ctx->band12_ok && ctx->band_rows == 12;
    const int nb = (c.num_envs + kScanEnvsPerBlock - 1) / kScanEnvsPerBlock;
    const bool fused = c.obs_h == c.obs_w && band12 && ctx->tune.no_full == 0 && packed_raw3_ok(ctx) && c.num_envs + nb <= 65535 &&
                       ctx->tune.packed_unfused == 0 && ctx->tune.packed_wave == 0 &&
                       ctx->tune.pipe_parts == 0 && ctx->tune.wave == 0 && ctx->tune.pair12 == 0 && ctx->ingest_t == 256;
    if (!fused) {
        if (compact) rc = gray ? agx_ingest_gray_raw_compact(ctx, d_screens, d_cmd, stream) : agx_ingest_compact(ctx, d_screens, d_cmd, stream);
        else rc = gray ? agx_ingest_gray_raw(ctx, d_screens, d_cmd, stream) : agx_ingest(ctx, d_screens, d_cmd, stream);
        if (rc) return rc;
        return agx_fovea_flexible_packed(ctx, d_action, action_dtype, d_action_type, d_packed, capacity_floats, d_offsets, d_fov_loc,
                                         d_fov_res, stream);
    }
    DeviceGuard g(c.device);
    IngestParams p = ingest_params(ctx, d_screens, d_cmd);
    p.band_rows = 12;
    p.nbands = c.obs_h / 12;
    if (compact) {
        p.src_rows = (int32_t)ctx->src_rows.size();
        p.ytab = ctx->in_ytab_c;
        p.y_affine = 0;
    }
    const FlexScanParams q = packed_scan_params(ctx, d_action, action_dtype, d_action_type, d_fov_loc, d_fov_res);
    // launch 1: the scan blocks (first rows of the grid) + the ingest bands
    const dim3 grid(p.nbands, nb + c.num_envs), block(kThreads);
    const size_t lds = band12_lds(ctx);
    if (compact) {
        if (gray) AGX_LAUNCH(0, (k_ingest_full12_flexscan<true, true>), grid, block, lds, S(stream), p, q, nb);
        else AGX_LAUNCH(0, (k_ingest_full12_flexscan<false, true>), grid, block, lds, S(stream), p, q, nb);
    } else {
        if (gray) AGX_LAUNCH(0, (k_ingest_full12_flexscan<true, false>), grid, block, lds, S(stream), p, q, nb);
        else AGX_LAUNCH(0, (k_ingest_full12_flexscan<false, false>), grid, block, lds, S(stream), p, q, nb);
    }
    AGX_HIP(ctx, hipGetLastError());
    ctx->cur_head ^= 1;
    ctx->cur_fov ^= 1;
    // launch 2: the crops
    return packed_crops_raw3(ctx, d_packed, capacity_floats, d_offsets, stream);
}

}  // extern "C"

#include "agx_loop_impl.h"
