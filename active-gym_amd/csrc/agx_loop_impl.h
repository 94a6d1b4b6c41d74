// agx_loop_impl.h - the native step loop declared in include/agx_loop.h (included at the end of agx_api.hip: it is built on
// the public entry points of this very library, plus two row-copy kernels and the HIP runtime's copy / event calls).
#pragma once
#include "agx_loop.h"

namespace agx {

// rows of `row16` 16-byte words: dst[j] = src[idx[j]] (gather) / dst[idx[j]] = src[j] (scatter); grid = (ceil(row16 / 256), k)
struct RowCopyParams {
    const uint4 *src;
    uint4 *dst;
    const int32_t *idx;      // [k]
    int64_t src_pitch16, dst_pitch16;
    int32_t row16;
    int32_t gather;
};
__global__ __launch_bounds__(kThreads) void k_row_copy(RowCopyParams p) {
    const int q = blockIdx.x * kThreads + threadIdx.x;
    if (q >= p.row16) return;
    const int j = blockIdx.y;
    const int64_t e = p.idx[j];
    const int64_t s = p.gather ? e : j, d = p.gather ? j : e;
    p.dst[d * p.dst_pitch16 + q] = p.src[s * p.src_pitch16 + q];
}
// the same gather in 4-byte words, for observation rows that are not a multiple of 16 bytes (a raw-crop context with an odd
// fov size: fs * fov_h * fov_w floats per env); grid = (ceil(row32 / 256), k)
__global__ __launch_bounds__(kThreads) void k_row_gather32(const uint32_t *src, const int32_t *idx, uint32_t *dst, int64_t row32) {
    const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (q >= row32) return;
    const int j = blockIdx.y;
    dst[(int64_t)j * row32 + q] = src[(int64_t)idx[j] * row32 + q];
}
// 8-byte rows (fov_loc / fov_res): dst[j] = src[idx[j]]
__global__ __launch_bounds__(kThreads) void k_gather_int2(const int2 *src, const int32_t *idx, int2 *dst, int k) {
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j < k) dst[j] = src[idx[j]];
}

}  // namespace agx

struct agx_loop {
    agx_ctx *ctx = nullptr;
    agx_host_source src{};
    agx_loop_config cfg{};
    int N = 0;
    size_t screen_bytes = 0;          // one staged screen
    size_t obs_row_floats = 0;        // one env's observation
    bool fovea = false, flexible = false;
    // step screens: two pinned sets and two device sets, alternating; the copy stream and its event edges
    uint8_t *h_frames[2] = {nullptr, nullptr}, *h_cmd[2] = {nullptr, nullptr};
    uint8_t *d_frames[2] = {nullptr, nullptr}, *d_cmd[2] = {nullptr, nullptr};
    hipEvent_t ev_copy[2] = {nullptr, nullptr};      // copy stream: this pinned set has left the host / these screens are on the device
    hipEvent_t ev_free[2] = {nullptr, nullptr};      // launch stream: the kernels that read this device set have been enqueued and run
    int stage_i = 0, dset_i = 0;
    hipStream_t copy_stream = nullptr;
    // resets of a subset: packed screens + {idx i32 [N] | cmd u8 [N] | mask u8 [N]} in one pinned block, two sets
    uint8_t *h_rframes[2] = {nullptr, nullptr}, *h_rmeta[2] = {nullptr, nullptr};
    hipEvent_t ev_r[2] = {nullptr, nullptr};
    hipEvent_t ev_rfree = nullptr;
    int rset_i = 0;
    uint8_t *d_rframes = nullptr, *d_rmeta = nullptr;
    // terminal observations / fov state of the envs that ended an episode
    float *d_final_obs = nullptr;
    int32_t *d_final_loc = nullptr, *d_final_res = nullptr;
    std::vector<double> reward, raw;
    std::vector<uint8_t> done;
    std::vector<int32_t> done_idx, noops;
    std::string err;
};

namespace {

int lfail(agx_loop *l, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (l) l->err = buf;
    else g_create_err = buf;
    return code;
}
#define LOOP_HIP(l, expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return lfail((l), AGX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)
#define LOOP_AGX(l, expr)                                                                             \
    do {                                                                                              \
        const int rc_ = (expr);                                                                       \
        if (rc_ != AGX_OK) return lfail((l), rc_, "%s: %s", #expr, agx_last_error((l)->ctx));        \
    } while (0)

int32_t *rmeta_idx(uint8_t *m) { return reinterpret_cast<int32_t *>(m); }
uint8_t *rmeta_cmd(uint8_t *m, int N) { return m + 4 * (size_t)N; }
uint8_t *rmeta_mask(uint8_t *m, int N) { return m + 5 * (size_t)N; }

int loop_ingest(agx_loop *l, const uint8_t *d_frames, const uint8_t *d_cmd, void *stream) {
    if (l->cfg.compact) return l->cfg.gray ? agx_ingest_gray_raw_compact(l->ctx, d_frames, d_cmd, stream) : agx_ingest_compact(l->ctx, d_frames, d_cmd, stream);
    return l->cfg.gray ? agx_ingest_gray_raw(l->ctx, d_frames, d_cmd, stream) : agx_ingest(l->ctx, d_frames, d_cmd, stream);
}

int loop_observe(agx_loop *l, const void *d_action, int dt, const int32_t *d_type, const uint8_t *d_mask, float *d_obs, int32_t *d_loc,
                 int32_t *d_res, void *stream) {
    switch (l->ctx->cfg.kind) {
        case AGX_KIND_BASE: return agx_observe_full(l->ctx, d_obs, stream);
        case AGX_KIND_FIXED: return agx_fovea_fixed(l->ctx, d_action, dt, d_mask, d_obs, d_loc, stream);
        case AGX_KIND_PERIPHERAL: return agx_fovea_peripheral(l->ctx, d_action, dt, d_mask, d_obs, d_loc, stream);
        default: return agx_fovea_flexible(l->ctx, d_action, dt, d_type, d_mask, d_obs, d_loc, d_res, stream);
    }
}

// The reset of the k envs in l->done_idx (ascending, already filled) with l->noops: emulators, upload of the packed screens and
// of {idx, cmd, mask}, scatter into slot 0 of the current device screens, ingest, fov state reset, masked re-observation.
// (active_gym/vector.py: _reset_subset + the autoreset tail of step().)
// gather: first copy the terminal rows of those envs (observation, fov_loc, fov_res as the step's kernels left them) to the
// side buffers - in stream order in front of everything that overwrites them.
int loop_reset_subset(agx_loop *l, int k, float *d_obs, int32_t *d_loc, int32_t *d_res, hipStream_t st, int64_t *h2d, bool gather) {
    const int N = l->N;
    l->rset_i ^= 1;
    const int rs = l->rset_i;
    LOOP_HIP(l, hipEventSynchronize(l->ev_r[rs]));                      // the reset before the previous one has left this pinned set
    uint8_t *meta = l->h_rmeta[rs];
    if (l->src.reset_packed(l->src.self, l->done_idx.data(), k, l->noops.data(), l->h_rframes[rs], (int64_t)l->screen_bytes,
                            rmeta_cmd(meta, N)) != 0)
        return lfail(l, AGX_E_STATE, "host source: reset_packed failed");
    std::memcpy(rmeta_idx(meta), l->done_idx.data(), sizeof(int32_t) * (size_t)k);
    uint8_t *mask = rmeta_mask(meta, N);
    std::memset(mask, 0, (size_t)N);
    for (int j = 0; j < k; ++j) mask[l->done_idx[j]] = 1;
    // on the COPY stream, queued between this step's screens and the next step's (a small copy issued on the launch stream would
    // reach the DMA engine behind the next step's screens and stall this step's reset kernels for a whole copy time)
    LOOP_HIP(l, hipStreamWaitEvent(l->copy_stream, l->ev_rfree, 0));   // the previous reset's kernels have read d_rframes / d_rmeta
    LOOP_HIP(l, hipMemcpyAsync(l->d_rmeta, meta, 6 * (size_t)N, hipMemcpyHostToDevice, l->copy_stream));
    LOOP_HIP(l, hipMemcpyAsync(l->d_rframes, l->h_rframes[rs], (size_t)k * l->screen_bytes, hipMemcpyHostToDevice, l->copy_stream));
    LOOP_HIP(l, hipEventRecord(l->ev_r[rs], l->copy_stream));
    LOOP_HIP(l, hipStreamWaitEvent(st, l->ev_r[rs], 0));
    if (h2d) *h2d += (int64_t)(6 * (size_t)N + (size_t)k * l->screen_bytes);
    if (gather) {
        if (l->obs_row_floats % 4 == 0) {
            agx::RowCopyParams q;
            q.src = reinterpret_cast<const uint4 *>(d_obs);
            q.dst = reinterpret_cast<uint4 *>(l->d_final_obs);
            q.idx = rmeta_idx(l->d_rmeta);
            q.src_pitch16 = q.dst_pitch16 = (int64_t)(l->obs_row_floats / 4);
            q.row16 = (int32_t)(l->obs_row_floats / 4);
            q.gather = 1;
            hipLaunchKernelGGL(agx::k_row_copy, dim3((q.row16 + kThreads - 1) / kThreads, k), dim3(kThreads), 0, st, q);
        } else {
            const int64_t row32 = (int64_t)l->obs_row_floats;
            hipLaunchKernelGGL(agx::k_row_gather32, dim3((unsigned)((row32 + kThreads - 1) / kThreads), k), dim3(kThreads), 0, st,
                               reinterpret_cast<const uint32_t *>(d_obs), rmeta_idx(l->d_rmeta), reinterpret_cast<uint32_t *>(l->d_final_obs), row32);
        }
        if (l->fovea && d_loc)
            hipLaunchKernelGGL(agx::k_gather_int2, dim3((k + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                               reinterpret_cast<const int2 *>(d_loc), rmeta_idx(l->d_rmeta), reinterpret_cast<int2 *>(l->d_final_loc), k);
        if (l->flexible && d_res)
            hipLaunchKernelGGL(agx::k_gather_int2, dim3((k + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                               reinterpret_cast<const int2 *>(d_res), rmeta_idx(l->d_rmeta), reinterpret_cast<int2 *>(l->d_final_res), k);
    }
    // the j-th packed screen -> slot 0 of env idx[j]'s screens
    agx::RowCopyParams p;
    p.src = reinterpret_cast<const uint4 *>(l->d_rframes);
    p.dst = reinterpret_cast<uint4 *>(l->d_frames[l->dset_i]);
    p.idx = rmeta_idx(l->d_rmeta);
    p.src_pitch16 = (int64_t)(l->screen_bytes / 16);
    p.dst_pitch16 = (int64_t)(2 * l->screen_bytes / 16);
    p.row16 = (int32_t)(l->screen_bytes / 16);
    p.gather = 0;
    hipLaunchKernelGGL(agx::k_row_copy, dim3((p.row16 + kThreads - 1) / kThreads, k), dim3(kThreads), 0, st, p);
    LOOP_HIP(l, hipGetLastError());
    LOOP_AGX(l, loop_ingest(l, l->d_frames[l->dset_i], rmeta_cmd(l->d_rmeta, N), st));
    const uint8_t *d_mask = rmeta_mask(l->d_rmeta, N);
    if (l->fovea) {
        LOOP_AGX(l, agx_fovea_reset(l->ctx, d_mask, st));
        LOOP_AGX(l, loop_observe(l, nullptr, 0, nullptr, d_mask, d_obs, d_loc, d_res, st));
    } else {
        LOOP_AGX(l, loop_observe(l, nullptr, 0, nullptr, nullptr, d_obs, nullptr, nullptr, st));
    }
    LOOP_HIP(l, hipEventRecord(l->ev_rfree, st));
    return AGX_OK;
}

}  // namespace

extern "C" {

const char *agx_loop_last_error(const agx_loop *loop) { return loop ? loop->err.c_str() : g_create_err.c_str(); }

int agx_loop_destroy(agx_loop *l) {
    if (!l) return AGX_OK;
    DeviceGuard g(l->ctx->cfg.device);
    (void)hipDeviceSynchronize();                      // copies from the pinned sets may still be in flight
    for (int b = 0; b < 2; ++b) {
        if (l->h_frames[b]) (void)hipHostFree(l->h_frames[b]);
        if (l->h_cmd[b]) (void)hipHostFree(l->h_cmd[b]);
        if (l->d_frames[b]) (void)hipFree(l->d_frames[b]);
        if (l->d_cmd[b]) (void)hipFree(l->d_cmd[b]);
        if (l->h_rframes[b]) (void)hipHostFree(l->h_rframes[b]);
        if (l->h_rmeta[b]) (void)hipHostFree(l->h_rmeta[b]);
        if (l->ev_copy[b]) (void)hipEventDestroy(l->ev_copy[b]);
        if (l->ev_free[b]) (void)hipEventDestroy(l->ev_free[b]);
        if (l->ev_r[b]) (void)hipEventDestroy(l->ev_r[b]);
    }
    if (l->ev_rfree) (void)hipEventDestroy(l->ev_rfree);
    if (l->copy_stream) (void)hipStreamDestroy(l->copy_stream);
    void *dev[] = {l->d_rframes, l->d_rmeta, l->d_final_obs, l->d_final_loc, l->d_final_res};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    delete l;
    return AGX_OK;
}

int agx_loop_create(agx_ctx *ctx, const agx_host_source *src, const agx_loop_config *cfg, agx_loop **out) {
    if (!ctx || !src || !cfg || !out) return lfail(nullptr, AGX_E_INVALID, "agx_loop_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(agx_loop_config))
        return lfail(nullptr, AGX_E_INVALID, "agx_loop_create: struct_size %d != %zu", cfg->struct_size, sizeof(agx_loop_config));
    if (!src->step || !src->reset_packed) return lfail(nullptr, AGX_E_INVALID, "agx_loop_create: the host source needs step and reset_packed");
    const agx_config &c = ctx->cfg;
    // (the flexible raw-crop mode is stepped in its PADDED form here - agx_fovea_flexible into [N][fs][obs_h][obs_w]; a caller who
    //  wants the packed ragged crops runs agx_fovea_flexible_packed itself: active_gym/vector.py keeps that on its Python loop)
    agx_loop *l = new (std::nothrow) agx_loop;
    if (!l) return lfail(nullptr, AGX_E_NOMEM, "out of host memory");
    l->ctx = ctx;
    l->src = *src;
    l->cfg = *cfg;
    l->N = c.num_envs;
    const size_t rows = cfg->compact ? ctx->src_rows.size() : (size_t)kRawH;
    l->screen_bytes = rows * kRawW * (cfg->gray ? 1 : 3);
    int32_t dims[4];
    agx_obs_shape(ctx, dims);
    l->obs_row_floats = (size_t)dims[1] * dims[2] * dims[3];
    l->fovea = c.kind != AGX_KIND_BASE;
    l->flexible = c.kind == AGX_KIND_FLEXIBLE;
    const size_t N = (size_t)l->N;
    l->reward.assign(N, 0.0);
    l->raw.assign(N, 0.0);
    l->done.assign(N, 0);
    l->done_idx.reserve(N);
    l->noops.assign(N, 0);
    DeviceGuard g(c.device);
    auto bail = [&](int code) {
        g_create_err = l->err;
        agx_loop_destroy(l);
        return code;
    };
#define TRYL(expr)                                                                      \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            lfail(l, AGX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));               \
            return bail(AGX_E_HIP);                                                     \
        }                                                                               \
    } while (0)
    // pinned staging: allocated by the calling thread (bind it to the GPU's NUMA node first: first touch - hostplan.bound_to)
    for (int b = 0; b < 2; ++b) {
        TRYL(hipHostMalloc(reinterpret_cast<void **>(&l->h_frames[b]), N * 2 * l->screen_bytes, hipHostMallocDefault));
        TRYL(hipHostMalloc(reinterpret_cast<void **>(&l->h_cmd[b]), N, hipHostMallocDefault));
        TRYL(hipHostMalloc(reinterpret_cast<void **>(&l->h_rframes[b]), N * l->screen_bytes, hipHostMallocDefault));
        TRYL(hipHostMalloc(reinterpret_cast<void **>(&l->h_rmeta[b]), 6 * N, hipHostMallocDefault));
        TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_frames[b]), N * 2 * l->screen_bytes + 2 * l->screen_bytes));   // + slack: K1 loads both screens speculatively
        TRYL(hipMemset(l->d_frames[b], 0, N * 2 * l->screen_bytes + 2 * l->screen_bytes));
        TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_cmd[b]), N));
        TRYL(hipEventCreateWithFlags(&l->ev_copy[b], hipEventDisableTiming));
        TRYL(hipEventCreateWithFlags(&l->ev_free[b], hipEventDisableTiming));
        TRYL(hipEventCreateWithFlags(&l->ev_r[b], hipEventDisableTiming));
    }
    TRYL(hipEventCreateWithFlags(&l->ev_rfree, hipEventDisableTiming));
    TRYL(hipStreamCreateWithFlags(&l->copy_stream, hipStreamNonBlocking));
    TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_rframes), N * l->screen_bytes));
    TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_rmeta), 6 * N));
    TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_final_obs), N * l->obs_row_floats * sizeof(float)));
    if (l->fovea) {
        TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_final_loc), N * 2 * sizeof(int32_t)));
        TRYL(hipMalloc(reinterpret_cast<void **>(&l->d_final_res), N * 2 * sizeof(int32_t)));
    }
#undef TRYL
    *out = l;
    return AGX_OK;
}

int agx_loop_reset(agx_loop *l, const int32_t *noops, float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res, void *stream) {
    if (!l) return AGX_E_INVALID;
    if (!d_obs) return lfail(l, AGX_E_INVALID, "agx_loop_reset: null obs buffer");
    DeviceGuard g(l->ctx->cfg.device);
    hipStream_t st = S(stream);
    const int N = l->N;
    // everything in flight from earlier steps is drained first: a full reset is rare and rewrites every staging set
    LOOP_HIP(l, hipStreamSynchronize(l->copy_stream));
    LOOP_HIP(l, hipStreamSynchronize(st));
    l->done_idx.resize(N);
    for (int i = 0; i < N; ++i) {
        l->done_idx[i] = i;
        l->noops[i] = noops ? noops[i] : 0;
    }
    const int rc = loop_reset_subset(l, N, d_obs, d_fov_loc, d_fov_res, st, nullptr, false);
    if (rc != AGX_OK) return rc;
    LOOP_HIP(l, hipEventRecord(l->ev_free[l->dset_i], st));
    return AGX_OK;
}

int agx_loop_reset_envs(agx_loop *l, const int32_t *idx, int32_t k, const int32_t *noops, float *d_obs, int32_t *d_fov_loc,
                        int32_t *d_fov_res, void *stream) {
    if (!l) return AGX_E_INVALID;
    if (!d_obs || !idx || k < 0 || k > l->N) return lfail(l, AGX_E_INVALID, "agx_loop_reset_envs: bad argument");
    for (int j = 0; j < k; ++j)
        if (idx[j] < 0 || idx[j] >= l->N) return lfail(l, AGX_E_INVALID, "agx_loop_reset_envs: env index %d out of range", idx[j]);
    if (k == 0) return AGX_OK;
    DeviceGuard g(l->ctx->cfg.device);
    hipStream_t st = S(stream);
    l->done_idx.assign(idx, idx + k);
    for (int j = 0; j < k; ++j) l->noops[j] = noops ? noops[j] : 0;
    const int rc = loop_reset_subset(l, k, d_obs, d_fov_loc, d_fov_res, st, nullptr, false);
    if (rc != AGX_OK) return rc;
    LOOP_HIP(l, hipEventRecord(l->ev_free[l->dset_i], st));
    return AGX_OK;
}

int agx_loop_step(agx_loop *l, const int32_t *motor, const void *d_action, int action_dtype, const int32_t *d_action_type,
                  float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res, agx_loop_result *res, void *stream) {
    if (!l) return AGX_E_INVALID;
    if (!motor || !d_obs || !res) return lfail(l, AGX_E_INVALID, "agx_loop_step: null argument");
    if (l->fovea && !d_fov_loc) return lfail(l, AGX_E_INVALID, "agx_loop_step: a fovea context needs d_fov_loc (final_info is gathered from it)");
    if (l->flexible && !d_fov_res) return lfail(l, AGX_E_INVALID, "agx_loop_step: a flexible context needs d_fov_res");
    DeviceGuard g(l->ctx->cfg.device);
    hipStream_t st = S(stream);
    const int N = l->N;
    const size_t env_bytes = 2 * l->screen_bytes;
    int64_t h2d = 0;
    // the other pinned set (its previous screens left it two steps ago) and the other device set
    l->stage_i ^= 1;
    l->dset_i ^= 1;
    const int sg = l->stage_i, ds = l->dset_i;
    LOOP_HIP(l, hipEventSynchronize(l->ev_copy[sg]));
    if (l->src.step(l->src.self, motor, l->h_frames[sg], l->h_cmd[sg], l->reward.data(), l->raw.data(), l->done.data()) != 0)
        return lfail(l, AGX_E_STATE, "host source: step failed");
    // screens + command bytes on the copy stream: behind the kernels that read this device set two steps ago, in front of this
    // step's kernels
    LOOP_HIP(l, hipStreamWaitEvent(l->copy_stream, l->ev_free[ds], 0));
    LOOP_HIP(l, hipMemcpyAsync(l->d_cmd[ds], l->h_cmd[sg], (size_t)N, hipMemcpyHostToDevice, l->copy_stream));
    LOOP_HIP(l, hipMemcpyAsync(l->d_frames[ds], l->h_frames[sg], (size_t)N * env_bytes, hipMemcpyHostToDevice, l->copy_stream));
    LOOP_HIP(l, hipEventRecord(l->ev_copy[sg], l->copy_stream));
    LOOP_HIP(l, hipStreamWaitEvent(st, l->ev_copy[sg], 0));
    h2d += (int64_t)N + (int64_t)N * (int64_t)env_bytes;
    LOOP_AGX(l, loop_ingest(l, l->d_frames[ds], l->d_cmd[ds], st));
    LOOP_AGX(l, loop_observe(l, l->fovea ? d_action : nullptr, action_dtype, d_action_type, nullptr, d_obs, d_fov_loc, d_fov_res, st));
    l->done_idx.clear();
    for (int i = 0; i < N; ++i)
        if (l->done[i]) l->done_idx.push_back(i);
    const int k = (int)l->done_idx.size();
    res->reward = l->reward.data();
    res->raw = l->raw.data();
    res->done = l->done.data();
    res->n_done = k;
    res->done_idx = l->done_idx.data();
    res->d_final_obs = nullptr;
    res->d_final_loc = res->d_final_res = nullptr;
    if (k > 0 && l->cfg.autoreset) {
        if (l->src.draw_noops) {
            if (l->src.draw_noops(l->src.noops_user, l->done_idx.data(), k, l->noops.data()) != 0)
                return lfail(l, AGX_E_STATE, "host source: draw_noops failed");
        } else {
            std::fill(l->noops.begin(), l->noops.begin() + k, 0);
        }
        // terminal rows first (gathers), then the reset pass - all inside loop_reset_subset, in stream order
        const int rc = loop_reset_subset(l, k, d_obs, d_fov_loc, d_fov_res, st, &h2d, true);
        res->d_final_obs = l->d_final_obs;
        if (l->fovea) res->d_final_loc = l->d_final_loc;
        if (l->flexible) res->d_final_res = l->d_final_res;

        if (rc != AGX_OK) return rc;
    }
    LOOP_HIP(l, hipEventRecord(l->ev_free[ds], st));
    res->h2d_bytes = h2d;
    return AGX_OK;
}

}  // extern "C"
