// examples/c_loop_demo.cpp - the whole vector step from plain C++: libagx_runner.so (emulators on host threads, compact staging)
// feeding libagx.so's native step loop (include/agx_loop.h), no Python, no torch.  Builds with
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_loop_demo.cpp -o c_loop_demo -L active-gym_amd/lib -lagx -lagx_runner \
//         -Wl,-rpath,$PWD/active-gym_amd/lib
// N scripted Atari envs, fixed fovea, autoreset: per step it prints a checksum of the observations, of the terminal observations of
// the envs that ended an episode, the reward sum and the done count; tests/test_gpu_env.py::test_c_loop_demo_matches_python_env runs
// the same envs through AtariVecEnv and compares line by line.  What gymnasium's SyncVectorEnv + the reference's AtariEnv /
// FixedFovealEnv do per step (reference atari_env.py:119-148,241, fov_env.py:187-221) is one call here: agx_loop_step.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agx.h"
#include "agx_loop.h"
#include "agx_runner.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                              \
        }                                                                          \
    } while (0)

static uint64_t fnv(const void *p, size_t n_words32) {
    const uint32_t *w = static_cast<const uint32_t *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n_words32; ++i) h = (h ^ w[i]) * 1099511628211ull;
    return h;
}

// the no-op counts of the envs about to be reset: a fixed arithmetic sequence, one draw per env in the order asked (the Python test
// hands AtariVecEnv the same sequence as its noop_fn; life-loss resets draw as well here and in the test, the runner ignores them)
struct Noops {
    uint32_t k = 0;
    agxr_runner *runner = nullptr;
    std::vector<uint8_t> lt;
};
static int draw_noops(void *user, const int32_t *idx, int32_t k, int32_t *out) {
    Noops *s = static_cast<Noops *>(user);
    agxr_get_state(s->runner, nullptr, s->lt.data());             // like NativeHostRunner.draw_noops: nothing is drawn for a life-loss reset
    for (int j = 0; j < k; ++j) out[j] = s->lt[idx[j]] ? 0 : (int32_t)((s->k++ * 7u + 3u) % 30u);
    return 0;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 48, steps = argc > 2 ? std::atoi(argv[2]) : 40, fs = 4;
    agx_config cfg = {};
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.num_envs = N;
    cfg.kind = AGX_KIND_FIXED;
    cfg.raw_h = 210, cfg.raw_w = 160, cfg.obs_h = 84, cfg.obs_w = 84, cfg.frame_stack = fs;
    cfg.fov_h = 30, cfg.fov_w = 30;
    cfg.out_mode = AGX_OUT_RESIZE, cfg.action_mode = AGX_MODE_ABSOLUTE, cfg.antialias = 1;
    agx_ctx *ctx = nullptr;
    if (agx_create(&cfg, &ctx) != AGX_OK) {
        std::fprintf(stderr, "agx_create: %s\n", agx_last_error(nullptr));
        return 3;
    }
    int32_t rows[210], n_rows = 0;
    agx_source_rows(ctx, rows, &n_rows);                          // the screen rows the resize reads: the runner stages only these

    agxr_config rc = {};
    rc.struct_size = (int32_t)sizeof(rc);
    rc.num_envs = N;
    rc.action_repeat = 4;
    rc.num_threads = 4;
    rc.seed = 21;
    rc.max_episode_frames = 108000;
    rc.scripted_actions = 4, rc.scripted_lives = 2, rc.scripted_p_life = 60, rc.scripted_p_over = 20;
    rc.backend = "scripted";
    rc.n_src_rows = n_rows;
    rc.src_rows = rows;
    agxr_runner *runner = nullptr;
    if (agxr_create(&rc, &runner) != AGXR_OK) {
        std::fprintf(stderr, "agxr_create: %s\n", agxr_last_error(nullptr));
        return 3;
    }
    Noops noops;
    noops.runner = runner;
    noops.lt.assign(N, 0);
    agx_host_source src = {};
    src.self = runner;
    src.step = reinterpret_cast<decltype(src.step)>(&agxr_step);              // same signatures: void* self = the runner handle
    src.reset_packed = reinterpret_cast<decltype(src.reset_packed)>(&agxr_reset_packed);
    src.draw_noops = draw_noops;
    src.noops_user = &noops;
    agx_loop_config lc = {(int32_t)sizeof(agx_loop_config), /*gray*/ 0, /*compact*/ 1, /*autoreset*/ 1};
    agx_loop *loop = nullptr;
    if (agx_loop_create(ctx, &src, &lc, &loop) != AGX_OK) {
        std::fprintf(stderr, "agx_loop_create: %s\n", agx_loop_last_error(nullptr));
        return 3;
    }
    const size_t row = (size_t)fs * 84 * 84, obs_n = (size_t)N * row;
    float *d_obs, *d_act;
    int32_t *d_loc;
    HIP_OK(hipMalloc(&d_obs, sizeof(float) * obs_n));
    HIP_OK(hipMalloc(&d_act, sizeof(float) * 2 * N));
    HIP_OK(hipMalloc(&d_loc, sizeof(int32_t) * 2 * N));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    std::vector<float> obs(obs_n), act(2 * N), fin;
    std::vector<int32_t> motor(N), first_noops(N);
    for (int i = 0; i < N; ++i) first_noops[i] = (int32_t)((noops.k++ * 7u + 3u) % 30u);
    if (agx_loop_reset(loop, first_noops.data(), d_obs, d_loc, nullptr, stream) != AGX_OK) {
        std::fprintf(stderr, "agx_loop_reset: %s\n", agx_loop_last_error(loop));
        return 3;
    }
    HIP_OK(hipMemcpyAsync(obs.data(), d_obs, sizeof(float) * obs_n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::printf("{\"step\": -1, \"obs\": %llu}\n", (unsigned long long)fnv(obs.data(), obs_n));
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < N; ++i) {
            motor[i] = (t + i) % 4;
            act[2 * i] = (float)((t * 5 + i * 3) % 60) - 2.5f;
            act[2 * i + 1] = (float)((t * 11 + i) % 64) - 4.0f;
        }
        HIP_OK(hipMemcpyAsync(d_act, act.data(), sizeof(float) * 2 * N, hipMemcpyHostToDevice, stream));
        agx_loop_result res = {};
        if (agx_loop_step(loop, motor.data(), d_act, AGX_DT_F32, nullptr, d_obs, d_loc, nullptr, &res, stream) != AGX_OK) {
            std::fprintf(stderr, "agx_loop_step: %s\n", agx_loop_last_error(loop));
            return 3;
        }
        uint64_t fin_hash = 0;
        if (res.n_done > 0) {
            fin.resize((size_t)res.n_done * row);
            HIP_OK(hipMemcpyAsync(fin.data(), res.d_final_obs, sizeof(float) * fin.size(), hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipMemcpyAsync(obs.data(), d_obs, sizeof(float) * obs_n, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        if (res.n_done > 0) fin_hash = fnv(fin.data(), fin.size());
        double rew = 0.0;
        for (int i = 0; i < N; ++i) rew += res.reward[i];
        std::printf("{\"step\": %d, \"obs\": %llu, \"final\": %llu, \"n_done\": %d, \"reward\": %.1f, \"h2d\": %lld}\n", t,
                    (unsigned long long)fnv(obs.data(), obs_n), (unsigned long long)fin_hash, res.n_done, rew, (long long)res.h2d_bytes);
    }
    agx_loop_destroy(loop);
    agxr_destroy(runner);
    agx_destroy(ctx);
    return 0;
}
