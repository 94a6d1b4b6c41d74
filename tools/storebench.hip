// tools/storebench.hip - what shape of a write-once f32 stream (the observations: N x 4 x 84 x 84 floats = 115.6 MB at N = 1024)
// reaches HBM fastest on MI355X: cache policy bits, bytes per lane, contiguity per wave, grid shape.  Each variant writes the
// same buffer R times back to back (HIP events around the R launches: sustained rate, dispatch gaps included).
// build: hipcc --offload-arch=gfx950 -O3 tools/storebench.hip -o tools/storebench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
// POL: 0 plain, 1 nt (builtin), 2 "sc1", 3 "sc0 sc1", 4 "sc1 nt", 5 "sc0 sc1 nt", 6 "sc0"
template <int POL>
__device__ __forceinline__ void st16(f4 *p, f4 v) {
    if (POL == 0) *p = v;
    else if (POL == 1) __builtin_nontemporal_store(v, p);
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
// (A) K2's shape: grid (4, N) x 256 threads, each workgroup one 28,224-B frame, lane-linear float4, 7 passes of 4 KB
template <int POL>
__global__ __launch_bounds__(256) void k_frame(f4 *out, float v) {
    const size_t base = ((size_t)blockIdx.y * 4 + blockIdx.x) * 1764;
    for (int q = threadIdx.x; q < 1764; q += 256) st16<POL>(&out[base + q], f4{v, v + q, v, v});
}
// (A') the same with pseudo-random k/255 values (the real observations; (A)'s are three constants and one ramp per float4)
template <int POL>
__global__ __launch_bounds__(256) void k_frame_rand(f4 *out, unsigned seed) {
    const size_t base = ((size_t)blockIdx.y * 4 + blockIdx.x) * 1764;
    for (int q = threadIdx.x; q < 1764; q += 256) {
        unsigned x = ((unsigned)(base + q) * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
        const float k = 1.0f / 255.0f;
        st16<POL>(&out[base + q], f4{(float)(x & 255u) * k, (float)((x >> 8) & 255u) * k, (float)((x >> 16) & 255u) * k, (float)(x >> 24) * k});
    }
}
// (B) the same bytes, each LANE writing 64 contiguous bytes per pass (4 float4): a wave covers 4 KB per pass
template <int POL>
__global__ __launch_bounds__(256) void k_lane64(f4 *out, float v, size_t n4) {
    const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (size_t q = w; q + 3 < n4; q += (size_t)gridDim.x * 1024)
#pragma unroll
        for (int k = 0; k < 4; ++k) st16<POL>(&out[q + k], f4{v, v + (float)k, v, v});
}
// (C) flat grid-stride, lane-linear float4, G workgroups (persistent-style)
template <int POL>
__global__ __launch_bounds__(256) void k_flat(f4 *out, float v, size_t n4) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) st16<POL>(&out[q], f4{v, v, v, v});
}
// (D) each workgroup a contiguous 28,224-B frame like (A) but 8 B / 4 B per lane
__global__ __launch_bounds__(256) void k_frame8(float2 *out, float v) {
    const size_t base = ((size_t)blockIdx.y * 4 + blockIdx.x) * 3528;
    for (int q = threadIdx.x; q < 3528; q += 256) __builtin_nontemporal_store(v + q, &out[base + q].x), __builtin_nontemporal_store(v, &out[base + q].y);
}

template <class F>
static float timeit(F launch, int R) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) launch(r);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < R; ++r) launch(r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / R * 1e3f;
}

// `storebench sizes`: is K2's slowdown per env beyond N = 1024 (batch sweep) the 256 MB Infinity Cache (MALL) absorbing a
// 115.6 MB stream that is rewritten in place, or something that grows with the footprint (TLB reach)?  (1) one buffer of N x 112,896 B
// for N = 1024 ... 8192; (2) N = 1024 written into P buffers in rotation (P x 115.6 MB: the stream no longer fits the cache, the
// launch is the same).
static int sizes_mode() {
    const int R = 40;
    const char *pol[] = {"plain", "nt", "sc1"};
    for (int N : {256, 512, 1024, 2048, 4096, 8192}) {
        const size_t n4 = (size_t)N * 4 * 1764, bytes = n4 * 16;
        f4 *ob; CK(hipMalloc(&ob, bytes));
        float us0 = timeit([&](int r) { hipLaunchKernelGGL(k_frame_rand<0>, dim3(4, N), dim3(256), 0, 0, ob, (unsigned)(r * 7919 + 1)); }, R);
        float us2 = timeit([&](int r) { hipLaunchKernelGGL(k_frame_rand<2>, dim3(4, N), dim3(256), 0, 0, ob, (unsigned)(r * 7919 + 1)); }, R);
        printf("one buffer   N = %5d (%7.1f MB): %-5s %8.2f us -> %.2f TB/s | %-5s %8.2f us -> %.2f TB/s  (%.2f us per 1024 envs)\n", N, bytes / 1e6,
               pol[0], us0, bytes / us0 / 1e6, pol[2], us2, bytes / us2 / 1e6, us2 * 1024 / N);
        CK(hipFree(ob));
    }
    const int N = 1024;
    const size_t n4 = (size_t)N * 4 * 1764, bytes = n4 * 16;
    for (int P : {1, 2, 3, 4, 8, 16}) {
        f4 *ob; CK(hipMalloc(&ob, bytes * P));
        float us0 = timeit([&](int r) { hipLaunchKernelGGL(k_frame_rand<0>, dim3(4, N), dim3(256), 0, 0, ob + (size_t)(r % P) * n4, (unsigned)(r * 7919 + 1)); }, R);
        float us2 = timeit([&](int r) { hipLaunchKernelGGL(k_frame_rand<2>, dim3(4, N), dim3(256), 0, 0, ob + (size_t)(r % P) * n4, (unsigned)(r * 7919 + 1)); }, R);
        printf("rotating     N = 1024 x %2d buffers (%7.1f MB): %-5s %8.2f us -> %.2f TB/s | %-5s %8.2f us -> %.2f TB/s\n", P, bytes * P / 1e6,
               pol[0], us0, bytes / us0 / 1e6, pol[2], us2, bytes / us2 / 1e6);
        CK(hipFree(ob));
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 's') return sizes_mode();
    const int N = 1024, R = 40;
    const size_t n4 = (size_t)N * 4 * 1764, bytes = n4 * 16;
    f4 *ob; CK(hipMalloc(&ob, bytes));
    const char *pol[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"};
#define RUNA(P) { float us = timeit([&](int r) { hipLaunchKernelGGL(k_frame<P>, dim3(4, N), dim3(256), 0, 0, ob, (float)r); }, R); \
                  printf("A frame/WG, 16 B lanes, %-11s %7.2f us -> %.2f TB/s\n", pol[P], us, bytes / us / 1e6); }
    RUNA(0) RUNA(1) RUNA(2) RUNA(3) RUNA(4) RUNA(5) RUNA(6)
#define RUNAR(P) { float us = timeit([&](int r) { hipLaunchKernelGGL(k_frame_rand<P>, dim3(4, N), dim3(256), 0, 0, ob, (unsigned)(r * 7919 + 1)); }, R); \
                  printf("A' frame/WG, random k/255, %-11s %7.2f us -> %.2f TB/s\n", pol[P], us, bytes / us / 1e6); }
    RUNAR(0) RUNAR(1) RUNAR(2) RUNAR(0) RUNAR(2)
    { float us = timeit([&](int r) { hipLaunchKernelGGL(k_frame<2>, dim3(4, N), dim3(256), 0, 0, ob, 0.0f); }, R);
      printf("A  frame/WG, sc1, f4{0, q, 0, 0}          %7.2f us -> %.2f TB/s\n", us, bytes / us / 1e6); }
#define RUNB(P, G) { float us = timeit([&](int r) { hipLaunchKernelGGL(k_lane64<P>, dim3(G), dim3(256), 0, 0, ob, (float)r, n4); }, R); \
                  printf("B 64 B per lane, %5d WGs, %-11s %7.2f us -> %.2f TB/s\n", G, pol[P], us, bytes / us / 1e6); }
    RUNB(1, 2048) RUNB(1, 4096) RUNB(1, 7056) RUNB(0, 4096)
#define RUNC(P, G) { float us = timeit([&](int r) { hipLaunchKernelGGL(k_flat<P>, dim3(G), dim3(256), 0, 0, ob, (float)r, n4); }, R); \
                  printf("C flat grid-stride, %5d WGs, %-11s %7.2f us -> %.2f TB/s\n", G, pol[P], us, bytes / us / 1e6); }
    RUNC(1, 1024) RUNC(1, 2048) RUNC(1, 4096) RUNC(1, 8192) RUNC(0, 2048) RUNC(5, 2048)
    { float us = timeit([&](int r) { hipLaunchKernelGGL(k_frame8, dim3(4, N), dim3(256), 0, 0, (float2 *)ob, (float)r); }, R);
      printf("D frame/WG, 8 B lanes (2 x dword nt)      %7.2f us -> %.2f TB/s\n", us, bytes / us / 1e6); }
    { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipMemsetAsync(ob, 0, bytes, 0); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 10; ++r) hipMemsetAsync(ob, r, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemsetAsync                             %7.2f us -> %.2f TB/s\n", ms / 10 * 1e3, bytes / (ms / 10 * 1e3) / 1e6); }
    return 0;
}
