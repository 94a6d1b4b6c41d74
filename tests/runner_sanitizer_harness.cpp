// tests/runner_sanitizer_harness.cpp - the C++ host runner (csrc/agx_runner.cpp, compiled INTO this program) driven from plain C++ so that
// it can run under ThreadSanitizer and AddressSanitizer + UBSan (tests/test_native_runner_cpu.py::test_runner_under_sanitizers builds one
// binary per sanitizer): scripted emulators with life-loss / game-over events, whole and compact staging, RGB and gray screens, the
// blocking step, the chunked asynchronous step, full resets, packed resets of the envs that ended an episode, render, state queries, from
// 1 to 8 worker threads.  Prints one checksum per configuration (equal across thread counts: the runner is deterministic).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agx_runner.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const uint8_t *b = static_cast<const uint8_t *>(p);
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

static int run(int N, int steps, int threads, bool gray, bool compact, uint64_t *sum) {
    std::vector<int32_t> rows;
    if (compact)
        for (int y = 0; y < 210; ++y)
            if (y % 5 != 2) rows.push_back(y);                        // 168 rows, the shape of the 84-row table
    agxr_config c = {};
    c.struct_size = (int32_t)sizeof(c);
    c.num_envs = N;
    c.action_repeat = 4;
    c.num_threads = threads;
    c.seed = 7;
    c.max_episode_frames = 400;                                       // time-limit terminals as well
    c.scripted_actions = 6, c.scripted_lives = 3, c.scripted_p_life = 40, c.scripted_p_over = 10;
    c.backend = "scripted";
    c.gray_frames = gray ? 1 : 0;
    c.n_src_rows = (int32_t)rows.size();
    c.src_rows = rows.empty() ? nullptr : rows.data();
    agxr_runner *r = nullptr;
    if (agxr_create(&c, &r) != AGXR_OK) {
        std::fprintf(stderr, "agxr_create: %s\n", agxr_last_error(nullptr));
        return 1;
    }
    rows.assign(rows.size(), -1);                                     // the list was copied at create
    const size_t screen = (size_t)(compact ? 168 : 210) * 160 * (gray ? 1 : 3);
    std::vector<uint8_t> frames((size_t)N * 2 * screen), packed((size_t)N * screen), cmd(N), done(N), lt(N), shot(210 * 160 * 3);
    std::vector<double> reward(N), raw(N);
    std::vector<int32_t> motor(N), idx(N), noops(N), lives(N);
    for (int i = 0; i < N; ++i) idx[i] = i, noops[i] = (i * 7 + 3) % 30;
    uint64_t h = 1469598103934665603ull;
#define CHECK(x)                                                                    \
    do {                                                                            \
        if ((x) != AGXR_OK) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, agxr_last_error(r));               \
            return 1;                                                               \
        }                                                                           \
    } while (0)
    CHECK(agxr_reset(r, idx.data(), N, noops.data(), frames.data(), (int64_t)(2 * screen), cmd.data()));
    h = fnv(h, cmd.data(), cmd.size());
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < N; ++i) motor[i] = (t + i) % agxr_num_actions(r);
        if (t % 3 == 2) {                                             // the chunked asynchronous form
            CHECK(agxr_step_begin(r, motor.data(), frames.data(), cmd.data(), reward.data(), raw.data(), done.data(), 5));
            for (int ch = 0; ch * 5 < N; ++ch) {
                CHECK(agxr_step_wait(r, ch));
                h = fnv(h, frames.data() + (size_t)ch * 5 * 2 * screen, (size_t)(ch * 5 + 5 <= N ? 5 : N - ch * 5) * 2 * screen);
            }
            CHECK(agxr_step_wait(r, -1));
        } else {
            CHECK(agxr_step(r, motor.data(), frames.data(), cmd.data(), reward.data(), raw.data(), done.data()));
            h = fnv(h, frames.data(), frames.size());
        }
        h = fnv(h, cmd.data(), cmd.size());
        h = fnv(h, reward.data(), sizeof(double) * N);
        h = fnv(h, done.data(), done.size());
        int k = 0;
        for (int i = 0; i < N; ++i)
            if (done[i]) idx[k] = i, noops[k] = (t + i) % 30, ++k;
        if (k > 0) {
            CHECK(agxr_reset_packed(r, idx.data(), k, noops.data(), packed.data(), (int64_t)screen, cmd.data()));
            h = fnv(h, packed.data(), (size_t)k * screen);
            h = fnv(h, cmd.data(), cmd.size());
        }
        if (t % 20 == 0) {
            CHECK(agxr_get_state(r, lives.data(), lt.data()));
            CHECK(agxr_render(r, t % N, shot.data()));
            h = fnv(h, lives.data(), sizeof(int32_t) * N);
            h = fnv(h, shot.data(), shot.size());
            agxr_set_training(r, (t / 20) % 2);
        }
    }
#undef CHECK
    agxr_destroy(r);
    *sum = h;
    return 0;
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 120, N = argc > 2 ? std::atoi(argv[2]) : 23;
    int bad = 0;
    for (int gray = 0; gray < 2; ++gray)
        for (int compact = 0; compact < 2; ++compact) {
            uint64_t ref = 0;
            for (int threads : {1, 3, 8}) {
                uint64_t h = 0;
                if (run(N, steps, threads, gray != 0, compact != 0, &h)) return 2;
                if (threads == 1) ref = h;
                if (h != ref) ++bad;
                std::printf("gray=%d compact=%d threads=%d checksum=%016llx%s\n", gray, compact, threads, (unsigned long long)h, h == ref ? "" : "  DIFFERS");
            }
        }
    return bad ? 3 : 0;
}
