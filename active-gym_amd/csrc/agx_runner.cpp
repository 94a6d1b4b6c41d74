// agx_runner.cpp — native host runner (include/agx_runner.h).  Plain C++17 + pthreads, no GPU code.
#include "agx_runner.h"

#include <dlfcn.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kH = 210, kW = 160, kFrameBytes = kH * kW * 3;
constexpr uint8_t kCmdClear = 0x04, kCmdSkip = 0x08;

struct Emulator {
    virtual ~Emulator() = default;
    virtual int act(int action) = 0;          // returns the reward of one frame
    virtual bool game_over() = 0;
    virtual int lives() = 0;
    virtual void reset_game() = 0;
    virtual void screen_rgb(uint8_t *out) = 0; // [210][160][3]
    virtual void screen_gray(uint8_t *out) = 0; // [210][160]  ALE getScreenGrayscale
    virtual std::vector<int> minimal_actions() = 0;
    // compact staging (agxr_config.src_rows): only the n listed screen rows, packed -> out [n][160][3] / [n][160].  Default: the
    // whole screen into a per-thread scratch buffer (cache-resident), then the wanted rows
    virtual void screen_rgb_rows(uint8_t *out, const int32_t *rows, int n) {
        thread_local std::vector<uint8_t> full((size_t)kFrameBytes);
        screen_rgb(full.data());
        for (int k = 0; k < n; ++k) std::memcpy(out + (size_t)k * kW * 3, full.data() + (size_t)rows[k] * kW * 3, (size_t)kW * 3);
    }
    virtual void screen_gray_rows(uint8_t *out, const int32_t *rows, int n) {
        thread_local std::vector<uint8_t> full((size_t)kH * kW);
        screen_gray(full.data());
        for (int k = 0; k < n; ++k) std::memcpy(out + (size_t)k * kW, full.data() + (size_t)rows[k] * kW, (size_t)kW);
    }
};

// ---- "scripted": splitmix64 event script, arithmetic screens; mirrored by tests/lcg_ale.py ----------------
struct ScriptedEmu final : Emulator {
    uint64_t s, seed;
    int n_actions, start_lives, p_life, p_over;
    int lives_ = 0, frame = 0, episode = 0;
    bool over = false;
    ScriptedEmu(uint64_t seed_, int na, int lv, int pl, int po)
        : s(seed_ * 0x9E3779B97F4A7C15ull + 0x1234567ull), seed(seed_), n_actions(na), start_lives(lv), p_life(pl), p_over(po) {
        lives_ = lv;
    }
    uint64_t rnd() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    int act(int) override {
        ++frame;
        const uint64_t u0 = rnd(), u1 = rnd(), u2 = rnd();
        int reward = 0;
        if (u0 % 100 < 15) reward = (int)(rnd() % 10) - 2;
        if (!over) {
            if ((int)(u1 % 1000) < p_life) {
                if (--lives_ <= 0) {
                    lives_ = 0;
                    over = true;
                }
            }
            if ((int)(u2 % 1000) < p_over) over = true;
        }
        return reward;
    }
    bool game_over() override { return over; }
    int lives() override { return lives_; }
    void reset_game() override {
        lives_ = start_lives;
        over = false;
        ++episode;
        frame = 0;
    }
    uint32_t screen_key() const {
        return (uint32_t)((seed * 1000003ull + (uint64_t)episode * 7919ull + (uint64_t)frame * 31ull) & 0xFFFFu);
    }
    static void rgb_row(uint32_t K, int y, uint8_t *row) {
        if (have_vbmi()) return rgb_row_vbmi(K, y, row);
        for (int x = 0; x < kW; ++x) {
            const uint32_t base = (uint32_t)(y * 7 + x * 13) + K * 3u + (uint32_t)((y * x) >> 4);
            uint8_t *p = row + (size_t)x * 3;
            p[0] = (uint8_t)(base & 0xFF);
            p[1] = (uint8_t)((base + 29u) & 0xFF);
            p[2] = (uint8_t)((base + 58u + (K >> 3)) & 0xFF);
        }
    }
    void screen_rgb(uint8_t *out) override {
        const uint32_t K = screen_key();
        for (int y = 0; y < kH; ++y) rgb_row(K, y, out + (size_t)y * kW * 3);
        if (nt_stores()) _mm_sfence();
    }
    void screen_rgb_rows(uint8_t *out, const int32_t *rows, int n) override {      // only the wanted rows are generated at all
        const uint32_t K = screen_key();
        for (int k = 0; k < n; ++k) rgb_row(K, rows[k], out + (size_t)k * kW * 3);
        if (nt_stores()) _mm_sfence();
    }
    static void gray_lut(uint32_t K, uint8_t *lut) {
        // r, g, b are functions of (base & 0xFF) for a given K: one 256-entry table per screen, like ALE's palette
        for (uint32_t v8 = 0; v8 < 256; ++v8) {
            const uint32_t r = v8, g = (v8 + 29u) & 0xFF, b = (v8 + 58u + (K >> 3)) & 0xFF;
            // round(.2989 r + .5870 g + .1140 b): the rational value decides, except on exact .5 ties, where
            // ALE's double expression is evaluated as written
            const uint32_t t = 2989u * r + 5870u * g + 1140u * b + 5000u;
            uint32_t q = t / 10000u;
            if (t % 10000u == 0u) {
                const double v = ((double)r * 0.2989 + (double)g * 0.5870) + (double)b * 0.1140;
                const double fl = std::floor(v);
                q = (uint32_t)fl + ((v - fl) >= 0.5 ? 1u : 0u);
            }
            lut[v8] = (uint8_t)q;
        }
    }
    static void gray_row_scalar(uint32_t K, const uint8_t *lut, int y, uint8_t *row) {
        uint8_t idx[kW];                                       // the index arithmetic vectorises; the table walk follows
        for (int x = 0; x < kW; ++x)
            idx[x] = (uint8_t)((uint32_t)(y * 7 + x * 13) + K * 3u + (uint32_t)((y * x) >> 4));
        for (int x = 0; x < kW; ++x) row[x] = lut[idx[x]];
    }
    // Staging rows are written once and read next by the DMA engine: the vector paths below use streaming stores, which keep
    // 55-165 MB per step out of the cores' caches (gray e2e step 1.02 against 1.03-1.10 ms, tighter medians: profiles/
    // r04_runner_nt_stores_ab.txt).  AGXR_NO_NT_STORES=1 restores ordinary stores.
    static bool nt_stores() {
        static const bool v = std::getenv("AGXR_NO_NT_STORES") == nullptr;
        return v;
    }
    static bool have_vbmi() {
        static const bool v = !std::getenv("AGXR_NO_VBMI") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512bw") &&
                              __builtin_cpu_supports("avx512vl");
        return v;
    }
#define AGXR_VBMI __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi")))
    // The rows where the CPU has AVX-512 VBMI (every MI355X host: EPYC Zen 4 / 5).  The palette index of the 160 pixels of a row in
    // 16-bit lanes (y x < 2^16, and only the low byte of the sum is kept, so the wrap-around of the other terms does not matter),
    // truncated to bytes: idx[0..63], idx[64..127], idx[128..159] (+ 32 zero bytes).
    AGXR_VBMI static void row_index_vbmi(uint32_t K, int y, __m512i out[3]) {
        alignas(64) static const uint16_t X[kW] = {
#define R8(b) (b), (b) + 1, (b) + 2, (b) + 3, (b) + 4, (b) + 5, (b) + 6, (b) + 7
#define R32(b) R8(b), R8((b) + 8), R8((b) + 16), R8((b) + 24)
            R32(0), R32(32), R32(64), R32(96), R32(128)
#undef R32
#undef R8
        };
        const __m512i vy = _mm512_set1_epi16((short)y), vb = _mm512_set1_epi16((short)((uint32_t)(y * 7) + K * 3u));
        const __m512i v13 = _mm512_set1_epi16(13);
        __m256i b[5];
        for (int j = 0; j < 5; ++j) {
            const __m512i xs = _mm512_load_si512(X + 32 * j);
            const __m512i v = _mm512_add_epi16(_mm512_add_epi16(vb, _mm512_mullo_epi16(xs, v13)),
                                               _mm512_srli_epi16(_mm512_mullo_epi16(vy, xs), 4));
            b[j] = _mm512_cvtepi16_epi8(v);
        }
        out[0] = _mm512_inserti64x4(_mm512_castsi256_si512(b[0]), b[1], 1);
        out[1] = _mm512_inserti64x4(_mm512_castsi256_si512(b[2]), b[3], 1);
        out[2] = _mm512_zextsi256_si512(b[4]);
    }
    // n32 32-byte pieces of v[] to dst (32-byte aligned for the streaming form; rows are 160 / 480 bytes from a page-aligned base)
    AGXR_VBMI static void store_pieces(uint8_t *dst, const __m512i *v, int n32) {
        if (nt_stores() && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0) {
            __m256i *d = reinterpret_cast<__m256i *>(dst);
            for (int k = 0; k < n32; ++k)
                _mm256_stream_si256(d + k, (k & 1) ? _mm512_extracti64x4_epi64(v[k >> 1], 1) : _mm512_castsi512_si256(v[k >> 1]));
        } else {
            for (int k = 0; k < n32; ++k)
                _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst) + k,
                                    (k & 1) ? _mm512_extracti64x4_epi64(v[k >> 1], 1) : _mm512_castsi512_si256(v[k >> 1]));
        }
    }
    // gray: the 256-entry table walked 64 entries at a time - two vpermi2b over the table's halves, selected by the index's top bit.
    // The scalar table walk was 85 % of a gray screen (13 of 15 us); it is what ALE does per screen too (palette -> gray), so a
    // stand-in that spends its time there is not unfair to ALE, only slower than it has to be.
    AGXR_VBMI static void gray_row_vbmi(uint32_t K, const uint8_t *lut, int y, uint8_t *row) {
        const __m512i t0 = _mm512_loadu_si512(lut), t1 = _mm512_loadu_si512(lut + 64), t2 = _mm512_loadu_si512(lut + 128),
                      t3 = _mm512_loadu_si512(lut + 192);
        __m512i idx[3], o[3];
        row_index_vbmi(K, y, idx);
        for (int j = 0; j < 3; ++j)
            o[j] = _mm512_mask_blend_epi8(_mm512_movepi8_mask(idx[j]), _mm512_permutex2var_epi8(t0, idx[j], t1),
                                          _mm512_permutex2var_epi8(t2, idx[j], t3));
        store_pieces(row, o, 5);
    }
    // RGB: byte t of the row is idx[t / 3] + {0, 29, 58 + (K >> 3)}[t % 3] (mod 256).  192 output bytes per 64 indices, so output
    // vector j takes its indices from idx[j / 3] through one of three fixed vpermb patterns, and its offsets from one of three
    // phase patterns (64 = 1 mod 3: the phase moves by one per vector).
    AGXR_VBMI static void rgb_row_vbmi(uint32_t K, int y, uint8_t *row) {
        alignas(64) static uint8_t P[3][64], PH[3][64];
        static const bool init = [] {
            for (int m = 0; m < 3; ++m)
                for (int t = 0; t < 64; ++t) P[m][t] = (uint8_t)((64 * m + t) / 3), PH[m][t] = (uint8_t)((64 * m + t) % 3);
            return true;
        }();
        (void)init;
        alignas(64) uint8_t off[64] = {0, 29, (uint8_t)(58u + (K >> 3))};
        const __m512i voff = _mm512_load_si512(off);
        __m512i idx[3], o[8];
        row_index_vbmi(K, y, idx);
        for (int j = 0; j < 8; ++j) {
            const int m = j % 3;
            const __m512i src = _mm512_permutexvar_epi8(_mm512_load_si512(P[m]), idx[j / 3]);
            o[j] = _mm512_add_epi8(src, _mm512_permutexvar_epi8(_mm512_load_si512(PH[m]), voff));
        }
        store_pieces(row, o, 15);
    }
#undef AGXR_VBMI
    static void gray_row(uint32_t K, const uint8_t *lut, int y, uint8_t *row) {
        if (have_vbmi()) gray_row_vbmi(K, lut, y, row);
        else gray_row_scalar(K, lut, y, row);
    }
    void screen_gray(uint8_t *out) override {                 // what ALE's palette would give for these RGB values
        const uint32_t K = screen_key();
        uint8_t lut[256];
        gray_lut(K, lut);
        for (int y = 0; y < kH; ++y) gray_row(K, lut, y, out + (size_t)y * kW);
        if (nt_stores()) _mm_sfence();
    }
    void screen_gray_rows(uint8_t *out, const int32_t *rows, int n) override {
        const uint32_t K = screen_key();
        uint8_t lut[256];
        gray_lut(K, lut);
        for (int k = 0; k < n; ++k) gray_row(K, lut, rows[k], out + (size_t)k * kW);
        if (nt_stores()) _mm_sfence();
    }
    std::vector<int> minimal_actions() override {
        std::vector<int> v(n_actions);
        for (int i = 0; i < n_actions; ++i) v[i] = i;
        return v;
    }
};

// ---- "ale_c": atari_py's C wrapper, dlopen'ed ------------------------------------------------------------
struct AleApi {
    void *lib = nullptr;
    void *(*ALE_new)() = nullptr;
    void (*ALE_del)(void *) = nullptr;
    void (*setInt)(void *, const char *, int) = nullptr;
    void (*setFloat)(void *, const char *, float) = nullptr;
    void (*setBool)(void *, const char *, bool) = nullptr;
    void (*loadROM)(void *, const char *) = nullptr;
    int (*act)(void *, int) = nullptr;
    bool (*game_over)(void *) = nullptr;
    void (*reset_game)(void *) = nullptr;
    int (*lives)(void *) = nullptr;
    int (*getMinimalActionSize)(void *) = nullptr;
    void (*getMinimalActionSet)(void *, int *) = nullptr;
    void (*getScreenRGB)(void *, unsigned char *) = nullptr;
    void (*getScreenGrayscale)(void *, unsigned char *) = nullptr;
    std::string load(const char *path) {
        lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return std::string("dlopen failed: ") + dlerror();
#define SYM(name)                                                              \
    name = reinterpret_cast<decltype(name)>(dlsym(lib, #name));                \
    if (!name) return std::string("libale_c symbol missing: ") + #name;
        SYM(ALE_new) SYM(ALE_del) SYM(setInt) SYM(setFloat) SYM(setBool) SYM(loadROM) SYM(act) SYM(game_over)
        SYM(reset_game) SYM(lives) SYM(getMinimalActionSize) SYM(getMinimalActionSet) SYM(getScreenRGB) SYM(getScreenGrayscale)
#undef SYM
        return "";
    }
};

struct AleEmu final : Emulator {
    const AleApi *api;
    void *ale;
    AleEmu(const AleApi *a, const char *rom, int seed, int max_frames) : api(a), ale(a->ALE_new()) {
        api->setInt(ale, "random_seed", seed);                       // atari_env.py:45-50
        api->setInt(ale, "max_num_frames_per_episode", max_frames);
        api->setFloat(ale, "repeat_action_probability", 0.f);
        api->setInt(ale, "frame_skip", 0);
        api->setBool(ale, "color_averaging", false);
        api->loadROM(ale, rom);
    }
    ~AleEmu() override { api->ALE_del(ale); }
    int act(int a) override { return api->act(ale, a); }
    bool game_over() override { return api->game_over(ale); }
    int lives() override { return api->lives(ale); }
    void reset_game() override { api->reset_game(ale); }
    void screen_rgb(uint8_t *out) override { api->getScreenRGB(ale, out); }
    void screen_gray(uint8_t *out) override { api->getScreenGrayscale(ale, out); }
    std::vector<int> minimal_actions() override {
        std::vector<int> v(api->getMinimalActionSize(ale));
        api->getMinimalActionSet(ale, v.data());
        return v;
    }
};

// ---- a small persistent thread pool: start(fn) calls fn(worker, nworkers) on every worker, wait() joins ---
class Pool {
  public:
    // cpus: worker i is pinned to cpus[i % cpus.size()] (empty: not pinned)
    explicit Pool(int n, const std::vector<int> &cpus = {}) : n_(n) {
        for (int i = 0; i < n_; ++i) {
            threads_.emplace_back([this, i] { loop(i); });
            int cpu = -1;
            if (!cpus.empty()) {
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET(cpus[i % cpus.size()], &set);
                if (pthread_setaffinity_np(threads_.back().native_handle(), sizeof set, &set) == 0) cpu = cpus[i % cpus.size()];
            }
            cpu_.push_back(cpu);
        }
    }
    int cpu_of(int w) const { return w >= 0 && w < (int)cpu_.size() ? cpu_[w] : -1; }
    ~Pool() {
        wait();
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    int size() const { return n_; }
    bool busy() {
        std::lock_guard<std::mutex> l(m_);
        return pending_ != 0;
    }
    void start(std::function<void(int, int)> fn) {
        wait();
        {
            std::lock_guard<std::mutex> l(m_);
            fn_ = std::move(fn);
            pending_ = n_;
            ++gen_;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    void run(std::function<void(int, int)> fn) {
        start(std::move(fn));
        wait();
    }

  private:
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            fn_(id, n_);                 // fn_ is only replaced by start(), which first waits for pending_ == 0
            {
                std::lock_guard<std::mutex> l(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<int> cpu_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(int, int)> fn_;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

thread_local std::string g_create_err;

// CPUs this process may use: the scheduler affinity mask, the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us)
void host_cpus(int &affinity, double &quota, int &local_world) {
    cpu_set_t set;
    CPU_ZERO(&set);
    affinity = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    if (affinity < 1) affinity = 1;
    quota = 0.0;
    {
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string a, b;
        if (f >> a >> b) {
            if (a != "max") quota = std::atof(a.c_str()) / std::max(1.0, std::atof(b.c_str()));
        } else {
            std::ifstream q("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), per("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            double qv = 0, pv = 0;
            if ((q >> qv) && (per >> pv) && qv > 0 && pv > 0) quota = qv / pv;
        }
    }
    const char *lw = std::getenv("LOCAL_WORLD_SIZE");
    local_world = lw && std::atoi(lw) > 0 ? std::atoi(lw) : 1;
}

int default_threads() {
    int aff, lw;
    double quota;
    host_cpus(aff, quota, lw);
    int usable = aff;
    if (quota > 0.0) usable = std::max(1, std::min(aff, (int)(quota + 0.5)));
    return std::max(1, std::min(64, usable / lw));
}

}  // namespace

struct agxr_runner {
    agxr_config cfg;
    std::vector<int32_t> src_rows;           // compact staging: the staged screen rows (empty: whole screens)
    std::vector<int> cpus;                   // worker placement (empty: not pinned)
    size_t screen_bytes = 0;                 // one staged screen
    std::vector<std::unique_ptr<Emulator>> emu;
    std::vector<std::vector<int>> actions;   // per env: index -> emulator action (atari_env.py:51-52)
    std::vector<int32_t> lives;
    std::vector<uint8_t> life_termination;
    bool training = true;                    // atari_env.py:58
    std::unique_ptr<Pool> pool;
    AleApi ale;
    std::string err;
    // chunked asynchronous step (agxr_step_begin / agxr_step_wait)
    std::unique_ptr<std::atomic<int>[]> chunk_left;
    int n_chunks = 0, chunk_envs = 0;
    bool in_flight = false;               // between agxr_step_begin and agxr_step_wait(-1)
    std::mutex chunk_m;
    std::condition_variable chunk_cv;
};

static int fail(agxr_runner *r, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (r) r->err = buf; else g_create_err = buf;
    return code;
}

extern "C" {

const char *agxr_last_error(const agxr_runner *r) { return r ? r->err.c_str() : g_create_err.c_str(); }

int agxr_destroy(agxr_runner *r) {
    if (!r) return AGXR_OK;
    r->pool.reset();
    r->emu.clear();
    if (r->ale.lib) dlclose(r->ale.lib);
    delete r;
    return AGXR_OK;
}

int agxr_create(const agxr_config *cfg, agxr_runner **out) {
    if (!cfg || !out) return fail(nullptr, AGXR_E_INVALID, "agxr_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(agxr_config))
        return fail(nullptr, AGXR_E_INVALID, "agxr_create: struct_size %d != %zu", cfg->struct_size, sizeof(agxr_config));
    if (cfg->num_envs < 1 || cfg->action_repeat < 1 || !cfg->backend)
        return fail(nullptr, AGXR_E_INVALID, "agxr_create: num_envs >= 1, action_repeat >= 1 and a backend are required");
    auto r = std::make_unique<agxr_runner>();
    r->cfg = *cfg;
    const std::string be = cfg->backend;
    if (be == "ale_c") {
        if (!cfg->ale_lib || !cfg->rom_path) return fail(nullptr, AGXR_E_INVALID, "backend ale_c needs ale_lib and rom_path");
        const std::string e = r->ale.load(cfg->ale_lib);
        if (!e.empty()) return fail(nullptr, AGXR_E_BACKEND, "%s", e.c_str());
    } else if (be != "scripted") {
        return fail(nullptr, AGXR_E_INVALID, "unknown backend '%s'", cfg->backend);
    }
    for (int i = 0; i < cfg->num_envs; ++i) {
        const int64_t seed = cfg->seed + cfg->env_offset + i;
        if (be == "scripted")
            r->emu.emplace_back(new ScriptedEmu((uint64_t)seed, std::max(1, cfg->scripted_actions), std::max(1, cfg->scripted_lives),
                                                cfg->scripted_p_life, cfg->scripted_p_over));
        else
            r->emu.emplace_back(new AleEmu(&r->ale, cfg->rom_path, (int)seed, cfg->max_episode_frames));
        r->actions.push_back(r->emu.back()->minimal_actions());
    }
    r->lives.assign(cfg->num_envs, 0);
    r->life_termination.assign(cfg->num_envs, 0);
    if (cfg->n_src_rows < 0 || cfg->n_src_rows > kH || (cfg->n_src_rows > 0 && !cfg->src_rows))
        return fail(nullptr, AGXR_E_INVALID, "agxr_create: n_src_rows %d needs a list of at most %d rows", cfg->n_src_rows, kH);
    for (int k = 0; k < cfg->n_src_rows; ++k) {
        const int y = cfg->src_rows[k];
        if (y < 0 || y >= kH || (k > 0 && y <= cfg->src_rows[k - 1]))
            return fail(nullptr, AGXR_E_INVALID, "agxr_create: src_rows must be ascending rows in [0, %d)", kH);
        r->src_rows.push_back(y);
    }
    r->cfg.src_rows = nullptr;               // (the caller's list is not kept)
    r->screen_bytes = (size_t)(r->src_rows.empty() ? kH : (int)r->src_rows.size()) * kW * (cfg->gray_frames ? 1 : 3);
    if (cfg->n_cpus < 0 || (cfg->n_cpus > 0 && !cfg->cpu_list)) return fail(nullptr, AGXR_E_INVALID, "agxr_create: bad cpu_list");
    for (int k = 0; k < cfg->n_cpus; ++k) {
        if (cfg->cpu_list[k] < 0 || cfg->cpu_list[k] >= CPU_SETSIZE) return fail(nullptr, AGXR_E_INVALID, "agxr_create: cpu %d out of range", cfg->cpu_list[k]);
        r->cpus.push_back(cfg->cpu_list[k]);
    }
    r->cfg.cpu_list = nullptr;
    // default: the CPUs this process may use (affinity, cgroup quota) / LOCAL_WORLD_SIZE - one process per GPU shares the host with
    // its sibling ranks - and no more than 64 (measured on a 256-thread host: 64 workers step 1024 envs fastest; beyond that
    // wake-up and cache traffic cost more than the extra cores give)
    int nt = cfg->num_threads > 0 ? cfg->num_threads : default_threads();
    nt = std::max(1, std::min(nt, cfg->num_envs));
    r->pool = std::make_unique<Pool>(nt, r->cpus);
    *out = r.release();
    return AGXR_OK;
}

int agxr_num_actions(const agxr_runner *r) { return r ? (int)r->actions[0].size() : AGXR_E_INVALID; }

int agxr_default_threads(void) { return default_threads(); }
void agxr_host_cpus(int32_t out[3]) {
    int aff, lw;
    double quota;
    host_cpus(aff, quota, lw);
    out[0] = aff;
    out[1] = (int32_t)(quota + 0.5);
    out[2] = lw;
}
int agxr_num_threads(const agxr_runner *r) { return r ? r->pool->size() : AGXR_E_INVALID; }
int agxr_worker_cpu(const agxr_runner *r, int32_t w) { return r ? r->pool->cpu_of(w) : -1; }

// one screen of emulator e into the staging layout of this runner (whole or compact, RGB or gray)
static inline void grab(const agxr_runner *r, Emulator &e, uint8_t *dst) {
    const bool gray = r->cfg.gray_frames != 0;
    if (r->src_rows.empty()) {
        if (gray) e.screen_gray(dst); else e.screen_rgb(dst);
    } else {
        if (gray) e.screen_gray_rows(dst, r->src_rows.data(), (int)r->src_rows.size());
        else e.screen_rgb_rows(dst, r->src_rows.data(), (int)r->src_rows.size());
    }
}

void agxr_set_training(agxr_runner *r, int training) {
    if (r) r->training = training != 0;
}

static void step_env(agxr_runner *r, int i, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward, double *raw,
                     uint8_t *done) {
    Emulator &e = *r->emu[i];
    const int a = r->actions[i][motor[i]];
    int rew = 0, nvalid = 0;
    bool d = false;
    const size_t fb = r->screen_bytes;
    uint8_t *f = frames + (size_t)i * 2 * fb;
    for (int t = 0; t < r->cfg.action_repeat; ++t) {          // atari_env.py:123-131
        rew += e.act(a);
        if (t == 2) {
            grab(r, e, f);
            nvalid = 1;
        } else if (t == 3) {
            grab(r, e, f + fb);
            nvalid = 2;
        }
        d = e.game_over();
        if (d) break;
    }
    if (r->training) {                                           // atari_env.py:135-140
        const int lv = e.lives();
        if (lv < r->lives[i] && lv > 0) {
            r->life_termination[i] = d ? 0 : 1;
            d = true;
        }
        r->lives[i] = lv;
    }
    raw[i] = rew;
    reward[i] = r->cfg.clip_reward ? (double)((rew > 0) - (rew < 0)) : (double)rew;   // np.sign, atari_env.py:144
    done[i] = d ? 1 : 0;
    cmd[i] = (uint8_t)nvalid;
}

int agxr_step_begin(agxr_runner *r, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward, double *raw,
                    uint8_t *done, int32_t chunk_envs) {
    if (!r) return AGXR_E_INVALID;
    if (!motor || !frames || !cmd || !reward || !raw || !done) return fail(r, AGXR_E_INVALID, "agxr_step: null buffer");
    if (r->in_flight) return fail(r, AGXR_E_STATE, "agxr_step_begin: the previous step has not been waited for");
    const int N = r->cfg.num_envs;
    for (int i = 0; i < N; ++i)
        if (motor[i] < 0 || motor[i] >= (int)r->actions[i].size())
            return fail(r, AGXR_E_INVALID, "agxr_step: motor action %d of env %d outside [0,%zu)", motor[i], i, r->actions[i].size());
    if (chunk_envs < 1 || chunk_envs > N) chunk_envs = N;
    const int nc = (N + chunk_envs - 1) / chunk_envs;
    if (nc != r->n_chunks) r->chunk_left.reset(new std::atomic<int>[nc]);
    r->n_chunks = nc;
    r->chunk_envs = chunk_envs;
    for (int c = 0; c < nc; ++c) r->chunk_left[c].store(std::min(chunk_envs, N - c * chunk_envs));
    r->in_flight = true;
    // static, barrier-free schedule: worker w takes envs lo+w, lo+w+nw, ... of chunk 0, then of chunk 1, ... - env ->
    // thread stays fixed from step to step (emulator state stays in that core's cache), and chunks complete in
    // order, so chunk c's screens can be on their way to the GPU while chunk c+1 is still emulating
    r->pool->start([=](int w, int nw) {
        for (int c = 0; c < nc; ++c) {
            const int lo = c * chunk_envs, hi = std::min(N, lo + chunk_envs);
            int mine = 0;
            for (int i = lo + w; i < hi; i += nw, ++mine) step_env(r, i, motor, frames, cmd, reward, raw, done);
            if (mine && r->chunk_left[c].fetch_sub(mine) == mine) {
                std::lock_guard<std::mutex> l(r->chunk_m);
                r->chunk_cv.notify_all();
            }
        }
    });
    return AGXR_OK;
}

int agxr_step_wait(agxr_runner *r, int32_t chunk) {
    if (!r) return AGXR_E_INVALID;
    if (chunk < 0) {
        r->pool->wait();
        r->in_flight = false;
        return AGXR_OK;
    }
    if (!r->in_flight || chunk >= r->n_chunks) return fail(r, AGXR_E_INVALID, "agxr_step_wait: chunk %d of %d", chunk, r->n_chunks);
    std::unique_lock<std::mutex> l(r->chunk_m);
    r->chunk_cv.wait(l, [&] { return r->chunk_left[chunk].load() == 0; });
    return AGXR_OK;
}

int agxr_step(agxr_runner *r, const int32_t *motor, uint8_t *frames, uint8_t *cmd, double *reward, double *raw, uint8_t *done) {
    const int rc = agxr_step_begin(r, motor, frames, cmd, reward, raw, done, 0);
    return rc ? rc : agxr_step_wait(r, -1);
}

static int reset_impl(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames, int64_t env_stride,
                      uint8_t *cmd, bool packed) {
    if (!r) return AGXR_E_INVALID;
    if (!idx || !noops || !frames || !cmd || k < 0) return fail(r, AGXR_E_INVALID, "agxr_reset: bad argument");
    if (r->in_flight) return fail(r, AGXR_E_STATE, "agxr_reset: a step is in flight (agxr_step_wait(-1) first)");
    const int N = r->cfg.num_envs;
    for (int j = 0; j < k; ++j)
        if (idx[j] < 0 || idx[j] >= N) return fail(r, AGXR_E_INVALID, "agxr_reset: env index %d out of range", idx[j]);
    std::fill(cmd, cmd + N, kCmdSkip);
    r->pool->run([&](int w, int nw) {
        const int lo = (int)((int64_t)k * w / nw), hi = (int)((int64_t)k * (w + 1) / nw);
        for (int j = lo; j < hi; ++j) {
            const int i = idx[j];
            Emulator &e = *r->emu[i];
            uint8_t clear = 0;
            if (r->life_termination[i]) {                               // atari_env.py:86-88
                r->life_termination[i] = 0;
                e.act(0);
            } else {                                                     // atari_env.py:90-99
                clear = kCmdClear;
                e.reset_game();
                for (int t = 0; t < noops[j]; ++t) {
                    e.act(0);
                    if (e.game_over()) e.reset_game();
                }
            }
            if (r->actions[i].size() >= 3) {                            // fire reset, atari_env.py:102-108
                e.act(1);
                if (e.game_over()) {
                    e.reset_game();
                    e.act(2);
                }
                if (e.game_over()) e.reset_game();
            }
            uint8_t *dst = frames + (size_t)(packed ? j : i) * env_stride;    // packed: the j-th reset env's screen in row j
            grab(r, e, dst);
            r->lives[i] = e.lives();
            cmd[i] = (uint8_t)(1 | clear);
        }
    });
    return AGXR_OK;
}

int agxr_reset(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames, int64_t env_stride,
               uint8_t *cmd) {
    return reset_impl(r, idx, k, noops, frames, env_stride, cmd, false);
}

int agxr_reset_packed(agxr_runner *r, const int32_t *idx, int32_t k, const int32_t *noops, uint8_t *frames, int64_t row_stride,
                      uint8_t *cmd) {
    return reset_impl(r, idx, k, noops, frames, row_stride, cmd, true);
}

int agxr_get_state(const agxr_runner *r, int32_t *lives, uint8_t *life_termination) {
    if (!r) return AGXR_E_INVALID;
    if (lives) std::copy(r->lives.begin(), r->lives.end(), lives);
    if (life_termination) std::copy(r->life_termination.begin(), r->life_termination.end(), life_termination);
    return AGXR_OK;
}

int agxr_render(agxr_runner *r, int32_t i, uint8_t *out) {
    if (!r || !out || i < 0 || i >= r->cfg.num_envs) return AGXR_E_INVALID;
    r->emu[i]->screen_rgb(out);
    return AGXR_OK;
}

}  // extern "C"
