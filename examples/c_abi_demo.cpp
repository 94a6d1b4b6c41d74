// examples/c_abi_demo.cpp - libagx through its C ABI alone (include/agx.h): no Python, no torch.  Builds with
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp -o c_abi_demo -L active-gym_amd/lib -lagx \
//         -Wl,-rpath,$PWD/active-gym_amd/lib
// It steps N fixed-fovea envs over LCG-generated screens and prints checksums of the u8 ring, fov_loc and the float
// observations; tests/test_gpu_parity.py::test_c_abi_demo_matches_python_binding reproduces the same inputs through
// the ctypes binding and compares.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agx.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define AGX_OK_(x)                                                                 \
    do {                                                                           \
        int rc_ = (x);                                                             \
        if (rc_ != AGX_OK) {                                                       \
            std::fprintf(stderr, "%s: %d %s\n", #x, rc_, agx_last_error(ctx));     \
            return 3;                                                              \
        }                                                                          \
    } while (0)

static uint32_t lcg(uint32_t &s) {           // numerical-recipes LCG; the Python side uses the same recurrence
    s = s * 1664525u + 1013904223u;
    return s;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 64, steps = argc > 2 ? std::atoi(argv[2]) : 5, fs = 4;
    agx_config cfg = {};
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = 0;
    cfg.num_envs = N;
    cfg.kind = AGX_KIND_FIXED;
    cfg.raw_h = 210, cfg.raw_w = 160, cfg.obs_h = 84, cfg.obs_w = 84, cfg.frame_stack = fs;
    cfg.fov_h = 30, cfg.fov_w = 30;
    cfg.out_mode = AGX_OUT_RESIZE, cfg.action_mode = AGX_MODE_ABSOLUTE, cfg.antialias = 1;
    agx_ctx *ctx = nullptr;
    AGX_OK_(agx_create(&cfg, &ctx));

    const size_t fbytes = (size_t)N * 2 * 210 * 160 * 3, obs_n = (size_t)N * fs * 84 * 84;
    uint8_t *d_frames, *d_cmd, *d_ring;
    float *d_act, *d_obs;
    int32_t *d_loc;
    HIP_OK(hipMalloc(&d_frames, fbytes));
    HIP_OK(hipMalloc(&d_cmd, N));
    HIP_OK(hipMalloc(&d_ring, obs_n));
    HIP_OK(hipMalloc(&d_act, sizeof(float) * 2 * N));
    HIP_OK(hipMalloc(&d_obs, sizeof(float) * obs_n));
    HIP_OK(hipMalloc(&d_loc, sizeof(int32_t) * 2 * N));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    std::vector<uint8_t> frames(fbytes), cmd(N), ring(obs_n);
    std::vector<float> act(2 * N), obs(obs_n);
    std::vector<int32_t> loc(2 * N);
    uint32_t s = 12345u;
    for (int t = 0; t < steps; ++t) {
        for (size_t i = 0; i < fbytes; ++i) frames[i] = (uint8_t)(lcg(s) >> 24);
        for (int i = 0; i < N; ++i) cmd[i] = (uint8_t)(t == 0 ? (1 | AGX_CMD_CLEAR) : (i % 7 == 3 ? 1 : 2));
        for (int i = 0; i < 2 * N; ++i) act[i] = (float)(lcg(s) >> 16) * (65.0f / 65536.0f) - 5.0f;
        HIP_OK(hipMemcpyAsync(d_frames, frames.data(), fbytes, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_cmd, cmd.data(), N, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_act, act.data(), sizeof(float) * 2 * N, hipMemcpyHostToDevice, stream));
        AGX_OK_(agx_ingest(ctx, d_frames, d_cmd, stream));
        AGX_OK_(agx_fovea_fixed(ctx, d_act, AGX_DT_F32, nullptr, d_obs, d_loc, stream));
    }
    AGX_OK_(agx_get_stack_u8(ctx, d_ring, stream));
    HIP_OK(hipMemcpyAsync(ring.data(), d_ring, obs_n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(obs.data(), d_obs, sizeof(float) * obs_n, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(loc.data(), d_loc, sizeof(int32_t) * 2 * N, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    uint64_t ring_sum = 0, ring_hash = 1469598103934665603ull, loc_sum = 0, obs_hash = 1469598103934665603ull;
    for (size_t i = 0; i < obs_n; ++i) {
        ring_sum += ring[i];
        ring_hash = (ring_hash ^ ring[i]) * 1099511628211ull;
    }
    for (int i = 0; i < 2 * N; ++i) loc_sum += (uint64_t)loc[i] * (uint64_t)(i + 1);
    double obs_sum = 0.0;
    for (size_t i = 0; i < obs_n; ++i) {
        obs_sum += obs[i];
        uint32_t bits;
        __builtin_memcpy(&bits, &obs[i], 4);
        obs_hash = (obs_hash ^ bits) * 1099511628211ull;
    }
    std::printf("{\"N\": %d, \"steps\": %d, \"ring_sum\": %llu, \"ring_hash\": %llu, \"loc_sum\": %llu, \"obs_sum\": %.9f, "
                "\"obs_hash\": %llu, \"abi\": %d}\n",
                N, steps, (unsigned long long)ring_sum, (unsigned long long)ring_hash, (unsigned long long)loc_sum, obs_sum,
                (unsigned long long)obs_hash, agx_abi_version());
    agx_destroy(ctx);
    return 0;
}
