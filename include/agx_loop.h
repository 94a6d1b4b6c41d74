/*
 * agx_loop.h — the native step loop of libagx.so: one C call per vector step.
 *
 * What it replaces: the per-step orchestration around the kernels — what gymnasium's SyncVectorEnv does for the
 * reference's envs (reference atari_env.py:241: step every env, reset the done ones inside the same call and hand
 * their last observation back as `final_observation`) and what AtariEnv._step / _reset do around the image work
 * (atari_env.py:84-148).  active_gym/vector.py does the same in Python (a dozen torch calls per step); with the
 * emulators, the PCIe copy and the kernels each under a millisecond at N = 1024, that Python is a third of a
 * gray-screen step.  agx_loop_step owns the whole sequence:
 *
 *   emulators (host source callback) -> pinned staging (two sets, alternating) -> hipMemcpyAsync on a copy stream
 *   (two device screen sets, event-ordered against the launch stream) -> agx_ingest* -> the context's fovea kernel
 *   -> for the envs that ended an episode: terminal observations / fov state gathered to side buffers, emulator
 *   reset (callback), packed reset screens uploaded and scattered, agx_ingest* (CLEAR), agx_fovea_reset, masked
 *   re-observation.
 *
 * Nothing here synchronises the device or the caller's stream: the call returns when everything is enqueued; host
 * outputs (reward, done, ...) are complete at return, device outputs are ordered on `stream`.  The only host waits
 * are on the loop's own copies out of a pinned staging set before that set is overwritten - copies issued two steps
 * (two resets) earlier - which is also what bounds how far the host can run ahead of the device: two steps.
 * The host source is a table of C callbacks, so libagx.so does not link the runner: active_gym/native_loop.py
 * fills it with the entry points of libagx_runner.so (agxr_step, agxr_reset_packed have exactly these signatures).
 */
#ifndef AGX_LOOP_H
#define AGX_LOOP_H

#include "agx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agx_loop agx_loop;

typedef struct agx_host_source {
    void *self;          /* first argument of step / reset_packed (an agxr_runner*) */
    /* One AtariEnv._step per env (atari_env.py:119-148): the two sampled screens of env i to frames + i * 2 * screen_bytes,
     * cmd[i] = nvalid, reward / raw / done per env.  Signature of agxr_step. */
    int (*step)(void *self, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward, double *raw, uint8_t *done);
    /* One AtariEnv._reset for the k envs idx[] (atari_env.py:84-117): the j-th env's single reset screen to
     * frames + j * row_stride, cmd[i] = 1 | AGX_CMD_CLEAR (full reset) or 1 (life-loss reset), AGX_CMD_SKIP for every
     * other env.  Signature of agxr_reset_packed. */
    int (*reset_packed)(void *self, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames, int64_t row_stride,
                        uint8_t *cmd);
    /* The reference draws random.randrange(30) no-ops per full reset from Python's global `random` (atari_env.py:96):
     * the caller keeps that stream.  Called with the k envs about to be reset; fills noops[k].  NULL: no no-ops. */
    int (*draw_noops)(void *user, const int32_t *idx, int32_t k, int32_t *noops);
    void *noops_user;
} agx_host_source;

typedef struct agx_loop_config {
    int32_t struct_size;
    int32_t gray;        /* 1: the source writes grayscale screens (agx_ingest_gray_raw*), 0: RGB (agx_ingest*)  */
    int32_t compact;     /* 1: screens hold only the agx_source_rows() rows (agx_ingest*_compact)                */
    int32_t autoreset;   /* 1: gymnasium<1.0 SyncVectorEnv semantics - done envs are reset inside agx_loop_step  */
} agx_loop_config;

/* What one step hands back.  Host arrays and device buffers belong to the loop and stay valid until the next
 * agx_loop_step / agx_loop_reset_envs on it. */
typedef struct agx_loop_result {
    const double *reward;        /* [N] sign(raw) if the source clips, else raw                                   */
    const double *raw;           /* [N]                                                                           */
    const uint8_t *done;         /* [N] incl. life-loss terminals                                                 */
    int32_t n_done;              /* envs that ended an episode in this step                                       */
    const int32_t *done_idx;     /* [n_done] ascending                                                            */
    const float *d_final_obs;    /* device [n_done][obs row]: their last observations (autoreset), else NULL      */
    const int32_t *d_final_loc;  /* device [n_done][2] fov_loc before the reset (fovea kinds), else NULL          */
    const int32_t *d_final_res;  /* device [n_done][2] fov_res before the reset (flexible kind), else NULL        */
    int64_t h2d_bytes;           /* bytes this call put on the copy stream                                        */
} agx_loop_result;

AGX_API int agx_loop_create(agx_ctx *ctx, const agx_host_source *src, const agx_loop_config *cfg, agx_loop **out);
AGX_API int agx_loop_destroy(agx_loop *loop);
AGX_API const char *agx_loop_last_error(const agx_loop *loop);

/* env.reset() of every env: reset_packed(all envs, noops[N]) -> upload -> ingest (CLEAR) -> agx_fovea_reset ->
 * observation.  d_obs as the context's agx_obs_shape (agx_observe_full's for AGX_KIND_BASE); d_fov_loc / d_fov_res
 * may be NULL. */
AGX_API int agx_loop_reset(agx_loop *loop, const int32_t *noops, float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res,
                           void *stream);

/* One vector step.  motor i32[N] (host); d_action / action_dtype / d_action_type as for agx_fovea_* (ignored for
 * AGX_KIND_BASE); d_obs receives the observation of every env - for an env that ended an episode (autoreset) the reset
 * observation, its terminal one is in res->d_final_obs. */
AGX_API int agx_loop_step(agx_loop *loop, const int32_t *motor, const void *d_action, int action_dtype,
                          const int32_t *d_action_type, float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res,
                          agx_loop_result *res, void *stream);

/* Reset only the k envs idx[] (what a caller without autoreset does after `done`): the other envs' rows of d_obs and
 * their state are left as they are.  noops[k]. */
AGX_API int agx_loop_reset_envs(agx_loop *loop, const int32_t *idx, int32_t k, const int32_t *noops, float *d_obs,
                                int32_t *d_fov_loc, int32_t *d_fov_res, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AGX_LOOP_H */
