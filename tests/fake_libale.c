/* A stand-in for atari_py's ale_interface/libale_c.so with the same C entry points, whose "emulator" is the scripted
 * one of tests/lcg_ale.py (splitmix64 events + arithmetic screens).  Test infrastructure: lets the native runner's
 * dlopen backend ("ale_c") run end to end in an image without ALE.  Settings are recorded so the test can check that
 * the runner applies the reference's (atari_env.py:45-50) before loadROM. */
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t s;
    int64_t seed;
    int lives, frame, episode, over, loaded;
    int max_frames, frame_skip, color_avg;
    float sticky;
    int settings_before_rom;
} Ale;

static uint64_t rnd(Ale *a) {
    a->s += 0x9E3779B97F4A7C15ull;
    uint64_t z = a->s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void *ALE_new(void) {
    Ale *a = (Ale *)calloc(1, sizeof(Ale));
    a->max_frames = -1; a->frame_skip = -1; a->color_avg = -1; a->sticky = -1.f;
    return a;
}
void ALE_del(void *p) { free(p); }
void setInt(void *p, const char *k, int v) {
    Ale *a = (Ale *)p;
    if (!a->loaded) a->settings_before_rom++;
    if (!strcmp(k, "random_seed")) a->seed = v;
    else if (!strcmp(k, "max_num_frames_per_episode")) a->max_frames = v;
    else if (!strcmp(k, "frame_skip")) a->frame_skip = v;
}
void setFloat(void *p, const char *k, float v) {
    Ale *a = (Ale *)p;
    if (!a->loaded) a->settings_before_rom++;
    if (!strcmp(k, "repeat_action_probability")) a->sticky = v;
}
void setBool(void *p, const char *k, bool v) {
    Ale *a = (Ale *)p;
    if (!a->loaded) a->settings_before_rom++;
    if (!strcmp(k, "color_averaging")) a->color_avg = v;
}
void loadROM(void *p, const char *rom) {
    Ale *a = (Ale *)p;
    (void)rom;
    a->loaded = 1;
    a->s = (uint64_t)a->seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    a->lives = 3;
    /* the reference's five settings must all have arrived, with its values, before the ROM (atari_env.py:45-50) */
    if (a->settings_before_rom != 5 || a->max_frames != 108000 || a->frame_skip != 0 || a->color_avg != 0 || a->sticky != 0.f)
        a->lives = -99;                  /* poisons every comparison in the test */
}
int act(void *p, int action) {
    Ale *a = (Ale *)p;
    (void)action;
    a->frame++;
    uint64_t u0 = rnd(a), u1 = rnd(a), u2 = rnd(a);
    int reward = 0;
    if (u0 % 100 < 15) reward = (int)(rnd(a) % 10) - 2;
    if (!a->over) {
        if ((int)(u1 % 1000) < 60) {
            a->lives--;
            if (a->lives <= 0) { a->lives = 0; a->over = 1; }
        }
        if ((int)(u2 % 1000) < 15) a->over = 1;
    }
    return reward;
}
bool game_over(void *p) { return ((Ale *)p)->over; }
void reset_game(void *p) {
    Ale *a = (Ale *)p;
    a->lives = 3; a->over = 0; a->episode++; a->frame = 0;
}
int lives(void *p) { return ((Ale *)p)->lives; }
int getMinimalActionSize(void *p) { (void)p; return 4; }
void getMinimalActionSet(void *p, int *out) { (void)p; out[0] = 0; out[1] = 1; out[2] = 3; out[3] = 4; }   /* NOOP FIRE RIGHT LEFT */
void getScreenRGB(void *p, unsigned char *out) {
    Ale *a = (Ale *)p;
    uint32_t K = (uint32_t)(((uint64_t)a->seed * 1000003ull + (uint64_t)a->episode * 7919ull + (uint64_t)a->frame * 31ull) & 0xFFFFu);
    for (int y = 0; y < 210; ++y)
        for (int x = 0; x < 160; ++x) {
            uint32_t base = (uint32_t)(y * 7 + x * 13) + K * 3u + (uint32_t)((y * x) >> 4);
            unsigned char *q = out + ((size_t)y * 160 + x) * 3;
            q[0] = (unsigned char)(base & 0xFF);
            q[1] = (unsigned char)((base + 29u) & 0xFF);
            q[2] = (unsigned char)((base + 58u + (K >> 3)) & 0xFF);
        }
}
void getScreenGrayscale(void *p, unsigned char *out) {
    static __thread unsigned char rgb[210 * 160 * 3];
    getScreenRGB(p, rgb);
    for (int i = 0; i < 210 * 160; ++i) {
        double x = ((double)rgb[3 * i] * 0.2989 + (double)rgb[3 * i + 1] * 0.5870) + (double)rgb[3 * i + 2] * 0.1140;
        double fl = (double)(long)x;
        out[i] = (unsigned char)(fl + ((x - fl) >= 0.5 ? 1.0 : 0.0));
    }
}
