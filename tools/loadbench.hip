// tools/loadbench.hip - cache-policy bits on K1's screen loads (read-once stream, 172 MB of 206 MB per launch at N = 1024), in
// K1's own access shape (grid (7, N), 240 loader threads, 8 x 12-byte loads per thread): buffer loads with aux = sc0 / nt / sc1
// combinations (raw buffer builtins: the compiler tracks them, unlike inline-asm loads).  A pool of 8 input batches is cycled so
// that nothing is served from the Infinity Cache; R launches back to back.
// build: hipcc --offload-arch=gfx950 -O3 tools/loadbench.hip -o tools/loadbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ROWB = 480, FRAMEB = 210 * ROWB;
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <int AUX>       // gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1
__global__ __launch_bounds__(256) void k_x3(const uint8_t *frames, uint32_t *out) {
    const int n = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
    const int rg = tid / 40, g4 = tid - rg * 40;
    if (rg >= 6) return;
    const uint8_t *fb = frames + (size_t)n * 2 * FRAMEB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(fb), 0, 2 * FRAMEB, 0x00020000);
    u32x3 w0[4], w1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int f = it >> 1, dyl = rg + 6 * (it & 1), dy = band * 12 + dyl, y0 = (10 * dy + 3) >> 2;
        const int o = f * FRAMEB + g4 * 12 + y0 * ROWB;
        w0[it] = __builtin_amdgcn_raw_buffer_load_b96(rs, o, 0, AUX);
        w1[it] = __builtin_amdgcn_raw_buffer_load_b96(rs, o + ROWB, 0, AUX);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) acc ^= w0[it].x ^ w0[it].y ^ w0[it].z ^ w1[it].x ^ w1[it].y ^ w1[it].z;
    if (acc == 0x12345678u) out[0] = acc;
}

// launch-rate probes: kernels that do (almost) nothing, in K1's grid and in coarser / finer ones
__global__ void k_empty(uint32_t *out) { if (threadIdx.x == 1023 && blockIdx.x == 0x7fffffff) out[0] = 1; }
template <int VG>      // the same with VG live VGPRs per thread (register file pressure changes how many waves a SIMD admits)
__global__ void k_empty_regs(uint32_t *out, const uint32_t *in) {
    uint32_t v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = in[(threadIdx.x + i * 7) & 1023];
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) a ^= v[i];
    if (a == 0x12345678u) out[0] = a;
}

// pseudo-random bytes (the bench's screens are random; hipMemset leaves a constant pattern)
__global__ void k_fill(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}

// usage: loadbench [rand] [lds]   rand: random screen bytes instead of a constant; lds: every workgroup also owns 16 KB of LDS
int main(int argc, char **argv) {
    bool rnd = false; int lds = 0;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "rand")) rnd = true; if (!strcmp(argv[i], "lds")) lds = 16128; }
    printf("screens: %s, dynamic LDS per workgroup: %d B\n", rnd ? "random bytes" : "constant 0x5a", lds);
    const int N = 1024, POOL = 8, R = 40;
    const size_t bytes = (size_t)N * 2 * FRAMEB;
    std::vector<uint8_t *> bufs(POOL);
    uint32_t *out;
    CK(hipMalloc(&out, 4096));
    for (size_t i = 0; i < bufs.size(); ++i) {
        CK(hipMalloc(&bufs[i], bytes));
        if (rnd) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t *>(bufs[i]), bytes / 4, (uint32_t)(i * 977 + 13));
        else CK(hipMemset(bufs[i], 0x5a, bytes));
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double alg = (double)N * 2 * 168 * ROWB;
#define RUN(AUX, NAME) { for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_x3<AUX>, dim3(7, N), dim3(256), lds, 0, bufs[r % POOL], out); \
        hipDeviceSynchronize(); hipEventRecord(e0); \
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_x3<AUX>, dim3(7, N), dim3(256), lds, 0, bufs[r % POOL], out); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("K1 load shape, %-12s %7.2f us -> %.2f TB/s (algorithmic 165.2 MB)\n", NAME, ms / R * 1e3, alg / (ms / R * 1e-3) / 1e12); }
#define RUNE(GX, GY, BT, LDS, NAME) { for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_empty, dim3(GX, GY), dim3(BT), LDS, 0, out); \
        hipDeviceSynchronize(); hipEventRecord(e0); \
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_empty, dim3(GX, GY), dim3(BT), LDS, 0, out); \
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("empty kernel, %-44s %7.2f us per launch -> %.0f workgroups / us\n", NAME, ms / R * 1e3, (double)(GX) * (GY) / (ms / R * 1e3)); }
    RUNE(7, 1024, 256, 0, "grid (7, 1024) x 256, no LDS")
    RUNE(7, 1024, 256, 16128, "grid (7, 1024) x 256, 16 KB LDS")
    RUNE(7, 512, 512, 32256, "grid (7, 512) x 512, 32 KB LDS")
    RUNE(7, 256, 1024, 64512, "grid (7, 256) x 1024, 63 KB LDS")
    RUNE(7, 2048, 128, 8064, "grid (7, 2048) x 128, 8 KB LDS")
    RUNE(4, 1024, 256, 13600, "grid (4, 1024) x 256, 13 KB LDS (K2)")
    RUNE(1, 1024, 256, 16128, "grid (1, 1024) x 256, 16 KB LDS")
    RUNE(1, 256, 256, 16128, "grid (1, 256) x 256, 16 KB LDS")
    RUN(0, "plain") RUN(2, "nt") RUN(16, "sc1") RUN(17, "sc0 sc1") RUN(18, "sc1 nt") RUN(19, "sc0 sc1 nt") RUN(1, "sc0") RUN(3, "sc0 nt")
    RUN(0, "plain") RUN(2, "nt")
    return 0;
}
