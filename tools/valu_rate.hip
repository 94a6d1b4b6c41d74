// tools/valu_rate.hip - issue rate of the VALU instructions K1's luminance / resize are made of, at full occupancy
// (8 waves per SIMD): cycles per wave64 instruction per SIMD.  build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, int iters) {
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u;
    unsigned b = threadIdx.x | 0x01010101u, c = 0x00ADEE74u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 5) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 8) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b));
            if (OP == 9) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 10) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a[i]));
            if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 12) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
            if (OP == 13) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 14) asm volatile("v_min3_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 15) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 13, 8" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 19) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long*)&a[i & ~1]) : "v"(b), "v"(c) : "vcc");
            if (OP == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 22) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
double run(const char *name, unsigned *out, unsigned long long *cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // per SIMD: 8 blocks per CU x 4 waves = 32 waves per CU = 8 per SIMD, each issuing iters * 16 instructions
    const double per = avg / ((double)iters * 16 * 8);
    printf("%-22s %8.0f cycles per wave -> %.2f cycles per wave64 instruction per SIMD (8 waves/SIMD)\n", name, avg, per);
    return per;
}

int main() {
    const int blocks = 256 * 8;
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    run<6>("v_fma_f32", out, cyc, blocks);
    run<0>("v_dot4_u32_u8", out, cyc, blocks);
    run<5>("v_dot2_u32_u16", out, cyc, blocks);
    run<1>("v_mul_hi_u32_u24", out, cyc, blocks);
    run<7>("v_mul_u32_u24", out, cyc, blocks);
    run<11>("v_mul_lo_u32", out, cyc, blocks);
    run<2>("v_mad_i32_i24", out, cyc, blocks);
    run<3>("v_lshl_add_u32", out, cyc, blocks);
    run<4>("v_perm_b32", out, cyc, blocks);
    run<8>("v_alignbyte_b32", out, cyc, blocks);
    run<9>("v_min3_u32", out, cyc, blocks);
    run<10>("v_lshrrev_b32", out, cyc, blocks);
    run<12>("v_cvt_f32_ubyte1", out, cyc, blocks);
    run<13>("v_mul_hi_u32", out, cyc, blocks);
    run<14>("v_min3_u16", out, cyc, blocks);
    run<15>("v_pk_min_u16", out, cyc, blocks);
    run<16>("v_bfe_u32", out, cyc, blocks);
    run<17>("v_and_or_b32", out, cyc, blocks);
    run<18>("v_mad_u32_u24", out, cyc, blocks);
    run<19>("v_lshl_or_b32", out, cyc, blocks);
    run<20>("v_mad_u64_u32", out, cyc, blocks);
    run<21>("v_and_b32", out, cyc, blocks);
    run<22>("v_max_u32", out, cyc, blocks);
    return 0;
}
