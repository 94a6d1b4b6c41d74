/*
 * agx_runner.h — C ABI of libagx_runner.so: the native host half of N Atari envs.
 *
 * One emulator per env on a pool of worker threads (one per host core), raw RGB screens written
 * straight into the caller's (pinned) staging buffer for agx_ingest.  It reproduces, per env, the
 * emulator-facing control flow of the reference's AtariEnv._step / _reset (reference
 * active_gym/atari_env.py:84-148) — the same contract as the Python runner
 * (active-gym_amd/active_gym/runner.py), against which tests/test_native_runner_cpu.py checks it.
 *
 * Backends
 *   "scripted"  deterministic event script + arithmetic screens (splitmix64), mirrored in Python
 *               (tests/lcg_ale.py) so that both runners can be compared bit for bit; also the
 *               emulator of end-to-end benchmarks on machines without ALE.
 *   "ale_c"     real ALE through atari_py's C wrapper (libale_c.so: ALE_new, loadROM, act,
 *               getScreenRGB, ...), dlopen'ed from `ale_lib`; configured as atari_env.py:44-50.
 *               Not exercised in the build image (no ALE there).
 */
#ifndef AGX_RUNNER_H
#define AGX_RUNNER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#define AGXR_API __attribute__((visibility("default")))
#else
#define AGXR_API
#endif

#define AGXR_OK 0
#define AGXR_E_INVALID -1
#define AGXR_E_BACKEND -2
#define AGXR_E_STATE -3

typedef struct agxr_runner agxr_runner;

typedef struct agxr_config {
    int32_t struct_size;
    int32_t num_envs;
    int32_t env_offset;        /* global index of local env 0 (emulator seed = seed + env_offset + i) */
    int32_t action_repeat;     /* args.action_repeat */
    int32_t clip_reward;       /* args.clip_reward: returned reward = sign(raw) */
    int32_t num_threads;       /* 0 = one per host core, capped at num_envs */
    int64_t seed;              /* args.seed */
    int32_t max_episode_frames;/* args.max_episode_length (108e3) */
    int32_t scripted_actions;  /* "scripted": size of the action set */
    int32_t scripted_lives;    /* "scripted": lives per episode */
    int32_t scripted_p_life;   /* "scripted": per-frame life-loss probability, per mille */
    int32_t scripted_p_over;   /* "scripted": per-frame game-over probability, per mille */
    const char *backend;       /* "scripted" | "ale_c" */
    const char *ale_lib;       /* path of libale_c.so ("ale_c") */
    const char *rom_path;      /* ROM file ("ale_c") */
    int32_t gray_frames;       /* 0: RGB screens u8[N][2][210][160][3] (agx_ingest); 1: ALE grayscale screens
                                  u8[N][2][210][160] (getScreenGrayscale, what the reference reads; agx_ingest_gray_raw) */
    /* Compact staging: stage only the n_src_rows screen rows listed in src_rows (ascending; agx_source_rows() of the
     * observation context gives them: the rows cv2.resize reads, 168 of 210 for 84 x 84).  Every screen in `frames` then has
     * n_src_rows rows instead of 210 - u8 [N][2][n_src_rows][160][3] - for agx_ingest_compact; reset screens likewise.
     * n_src_rows = 0: whole screens.  The list is copied at agxr_create. */
    int32_t n_src_rows;
    const int32_t *src_rows;
    /* Worker placement: worker w is pinned to CPU cpu_list[w % n_cpus] (pthread_setaffinity_np); n_cpus = 0: not pinned.
     * The caller derives the list from the topology (active_gym/hostplan.py: the CPUs of the NUMA node the rank's GPU hangs
     * off, split between the ranks that share the node).  The list is copied at agxr_create. */
    const int32_t *cpu_list;
    int32_t n_cpus;
    int32_t reserved;
} agxr_config;

/* Worker threads agxr_create starts when num_threads = 0: the CPUs this process may use (scheduler affinity and the
 * cgroup CPU quota) divided by LOCAL_WORLD_SIZE (one process per GPU; 1 when unset), at least 1, at most 64. */
AGXR_API int agxr_default_threads(void);
/* What that figure is made of: out[0] = CPUs in the affinity mask, out[1] = cgroup quota in CPUs (0 = none),
 * out[2] = LOCAL_WORLD_SIZE (1 when unset). */
AGXR_API void agxr_host_cpus(int32_t out[3]);

AGXR_API int agxr_create(const agxr_config *cfg, agxr_runner **out);
AGXR_API int agxr_destroy(agxr_runner *r);
AGXR_API const char *agxr_last_error(const agxr_runner *r);
AGXR_API int agxr_num_actions(const agxr_runner *r);
AGXR_API void agxr_set_training(agxr_runner *r, int training);   /* AtariEnv.train()/eval(), atari_env.py:158-163 */

/* One AtariEnv._step per env (atari_env.py:119-148).
 *   motor   i32[N]  index into the minimal action set
 *   frames  u8 [N][2][210][160][3] (or [N][2][210][160] with gray_frames; n_src_rows rows instead of 210 with compact
 *           staging)  the screens after t==2 / t==3 go to slots 0 / 1
 *   cmd     u8 [N]  nvalid for agx_ingest
 *   reward  f64[N]  sign(raw) if clip_reward else raw;  raw f64[N];  done u8[N] (incl. life-loss terminals) */
AGXR_API int agxr_step(agxr_runner *r, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward,
                       double *raw, uint8_t *done);

/* The same step, asynchronous and in chunks of `chunk_envs` consecutive envs (<= 0: one chunk), so that the
 * caller can start the H2D copy of chunk c's screens while chunk c+1 is still emulating.  All buffers must stay
 * valid until agxr_step_wait(r, -1) has returned; agxr_step_wait(r, c) returns once every env of chunk c has
 * written its screens / cmd / reward / done; -1 waits for the whole step.  A second begin, or a reset, before the
 * final wait fails with AGXR_E_STATE. */
AGXR_API int agxr_step_begin(agxr_runner *r, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward,
                             double *raw, uint8_t *done, int32_t chunk_envs);
AGXR_API int agxr_step_wait(agxr_runner *r, int32_t chunk);

/* One AtariEnv._reset for the k envs in idx (atari_env.py:84-117).  noops[j] = the random.randrange(30) draw of
 * env idx[j] (ignored for a life-loss reset), supplied by the caller so that the Python side keeps drawing from
 * the global `random` like the reference.  The single reset screen of env i goes to
 * frames + i * env_stride (bytes); cmd[i] = 1 | AGX_CMD_CLEAR for a full reset; other envs get AGX_CMD_SKIP. */
AGXR_API int agxr_reset(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames,
                        int64_t env_stride, uint8_t *cmd);
/* The same reset with the screens PACKED: env idx[j]'s reset screen goes to frames + j * row_stride (the autoreset inside a
 * vector step uploads the k reset screens as one contiguous copy; cmd stays indexed by env). */
AGXR_API int agxr_reset_packed(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames,
                               int64_t row_stride, uint8_t *cmd);

/* Number of worker threads, and the CPU worker w is pinned to (-1: not pinned, or w out of range). */
AGXR_API int agxr_num_threads(const agxr_runner *r);
AGXR_API int agxr_worker_cpu(const agxr_runner *r, int32_t w);

/* lives i32[N], life_termination u8[N] (either may be NULL) */
AGXR_API int agxr_get_state(const agxr_runner *r, int32_t *lives, uint8_t *life_termination);
/* current RGB screen of env i -> out u8[210][160][3] */
AGXR_API int agxr_render(agxr_runner *r, int32_t i, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
