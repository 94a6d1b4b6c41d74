// tools/membench.hip — pure-load floors for K1's access shapes on MI355X (diagnostic, not product).
// build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int RAWH=210, RAWW=160, ROWB=480, FRAMEB=RAWH*ROWB;
struct __attribute__((aligned(4))) U3 { uint32_t x,y,z; };

// (a) K1's shape: grid (7, N), 240 loader threads, 8 x 12-B loads per thread, rows (10dy+3)>>2 and +1
__global__ __launch_bounds__(256) void k_x3(const uint8_t* frames, uint32_t* out) {
  const int n = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
  const int rg = tid / 40, g4 = tid - rg*40;
  const uint8_t* fb = frames + (size_t)n*2*FRAMEB;
  uint32_t acc = 0;
  U3 w0[4], w1[4];
#pragma unroll
  for (int it=0; it<4; ++it) {
    int rj = min(rg + 6*it, 23); int f = rj >= 12; int dyl = rj - f*12; int dy = band*12 + dyl;
    int y0 = (10*dy+3)>>2;
    uint32_t o = f*FRAMEB + g4*12 + y0*ROWB;
    w0[it] = *reinterpret_cast<const U3*>(fb + o);
    w1[it] = *reinterpret_cast<const U3*>(fb + o + ROWB);
  }
#pragma unroll
  for (int it=0; it<4; ++it) acc ^= w0[it].x ^ w0[it].y ^ w0[it].z ^ w1[it].x ^ w1[it].y ^ w1[it].z;
  if (acc == 0x12345678u) out[0] = acc;
}
// (b) same bytes as 16-B loads: a row pair is 960 contiguous bytes = 60 lanes; 24 pairs per band-WG
//     -> 1440 x4-loads per WG, 256 threads -> 6 per thread (5.6)
__global__ __launch_bounds__(256) void k_x4(const uint8_t* frames, uint32_t* out) {
  const int n = blockIdx.y, band = blockIdx.x, tid = threadIdx.x;
  const uint8_t* fb = frames + (size_t)n*2*FRAMEB;
  uint4 w[6]; uint32_t acc=0;
#pragma unroll
  for (int it=0; it<6; ++it) {
    int t = min(tid + 256*it, 1439); int pair = t / 60, l = t - pair*60;
    int f = pair >= 12; int dyl = pair - f*12; int dy = band*12+dyl; int y0 = (10*dy+3)>>2;
    uint32_t o = f*FRAMEB + y0*ROWB + l*16;
    w[it] = *reinterpret_cast<const uint4*>(fb + o);
  }
#pragma unroll
  for (int it=0; it<6; ++it) acc ^= w[it].x ^ w[it].y ^ w[it].z ^ w[it].w;
  if (acc == 0x12345678u) out[0] = acc;
}
// (c) whole frames linearly, 16 B per lane, grid-stride-free: one WG per 24 KiB chunk
__global__ __launch_bounds__(256) void k_lin(const uint4* src, uint32_t* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 * 6 + threadIdx.x; uint32_t acc=0; uint4 w[6];
#pragma unroll
  for (int it=0; it<6; ++it) { size_t j = i + 256*it; if (j >= n16) j = n16-1; w[it] = src[j]; }
#pragma unroll
  for (int it=0; it<6; ++it) acc ^= w[it].x ^ w[it].y ^ w[it].z ^ w[it].w;
  if (acc == 0x12345678u) out[0] = acc;
}
// (c3) whole frames linearly, 12 B per lane (K1's load width, no partial-line waste): FETCH_SIZE calibration
__global__ __launch_bounds__(256) void k_lin3(const uint8_t* src, uint32_t* out, size_t n12) {
  size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; uint32_t acc=0; U3 w[8];
#pragma unroll
  for (int it=0; it<8; ++it) { size_t j = i + 256*it; if (j >= n12) j = n12-1; w[it] = *reinterpret_cast<const U3*>(src + j*12); }
#pragma unroll
  for (int it=0; it<8; ++it) acc ^= w[it].x ^ w[it].y ^ w[it].z;
  if (acc == 0x12345678u) out[0] = acc;
}
// (d) like (a) but one WG per env walking its 7 bands with the loads of band b+1 issued before band b is reduced
__global__ __launch_bounds__(256) void k_x3_pipe(const uint8_t* frames, uint32_t* out) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const int rg = tid / 40, g4 = tid - rg*40;
  const uint8_t* fb = frames + (size_t)n*2*FRAMEB;
  uint32_t acc = 0;
  U3 a0[4], a1[4], b0[4], b1[4];
  auto issue = [&](U3 (&w0)[4], U3 (&w1)[4], int band) {
#pragma unroll
    for (int it=0; it<4; ++it) {
      int rj = min(rg + 6*it, 23); int f = rj >= 12; int dyl = rj - f*12; int dy = min(band,6)*12 + dyl;
      int y0 = (10*dy+3)>>2; uint32_t o = f*FRAMEB + g4*12 + y0*ROWB;
      w0[it] = *reinterpret_cast<const U3*>(fb + o); w1[it] = *reinterpret_cast<const U3*>(fb + o + ROWB);
    }};
  auto red = [&](U3 (&w0)[4], U3 (&w1)[4]) {
#pragma unroll
    for (int it=0; it<4; ++it) acc ^= w0[it].x ^ w0[it].y ^ w0[it].z ^ w1[it].x ^ w1[it].y ^ w1[it].z; };
  issue(a0,a1,0);
  for (int band=0; band<8; band+=2) { issue(b0,b1,band+1); red(a0,a1); issue(a0,a1,band+2); red(b0,b1); }
  if (acc == 0x12345678u) out[0] = acc;
}
// (e) K2's store shape: grid (4, N), 256 threads, 7 x float4 stores per thread, lane-linear
__global__ __launch_bounds__(256) void k_st(float4* out, float v, int nt) {
  const size_t base = ((size_t)blockIdx.y*4 + blockIdx.x) * 1764;
  for (int q = threadIdx.x; q < 1764; q += 256) {
    float4 o = make_float4(v, v+q, v, v);
    if (nt) { __builtin_nontemporal_store(o.x,&out[base+q].x); __builtin_nontemporal_store(o.y,&out[base+q].y); __builtin_nontemporal_store(o.z,&out[base+q].z); __builtin_nontemporal_store(o.w,&out[base+q].w);} else out[base+q] = o;
  }
}
// (f) the 128-thread / 14-pass store shape of k_fovea_resize_s: 126 active lanes, 2016-B steps
__global__ __launch_bounds__(128) void k_st128(float4* out, float v, int nt) {
  const size_t base = ((size_t)blockIdx.y*4 + blockIdx.x) * 1764;
  const int tid = threadIdx.x; if (tid >= 126) return;
  for (int pass = 0; pass < 14; ++pass) { int q = pass*126 + tid;
    float4 o = make_float4(v, v+q, v, v);
    if (nt) { typedef float f4 __attribute__((ext_vector_type(4))); f4 w = {o.x,o.y,o.z,o.w}; __builtin_nontemporal_store(w, (f4*)&out[base+q]); } else out[base+q] = o; }
}
// (g) 256 threads, 252 active, 7 passes of 4032 B
__global__ __launch_bounds__(256) void k_st252(float4* out, float v, int nt) {
  const size_t base = ((size_t)blockIdx.y*4 + blockIdx.x) * 1764;
  const int tid = threadIdx.x; if (tid >= 252) return;
  for (int pass = 0; pass < 7; ++pass) { int q = pass*252 + tid;
    float4 o = make_float4(v, v+q, v, v);
    if (nt) { typedef float f4 __attribute__((ext_vector_type(4))); f4 w = {o.x,o.y,o.z,o.w}; __builtin_nontemporal_store(w, (f4*)&out[base+q]); } else out[base+q] = o; }
}
int main(int argc, char** argv) {
  const int N=1024, POOL=8; const size_t bytes=(size_t)N*2*FRAMEB;
  std::vector<uint8_t*> bufs(POOL); uint32_t* out; CK(hipMalloc(&out, 4096));
  for (auto& b: bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
  if (argc > 1 && argv[1][0] == 's') {
    // sweep mode: the K1 load shape (k_x3) and the K2 store shape (k_st) at N = 256 ... 4096 envs: slope = streaming rate,
    // intercept = what a launch of that shape costs before / after it streams
    float4* ob; CK(hipMalloc(&ob, (size_t)4096*4*1764*16));
    std::vector<uint8_t*> big(2); for (auto& b: big) { CK(hipMalloc(&b, (size_t)4096*2*FRAMEB)); CK(hipMemset(b, 0x5a, (size_t)4096*2*FRAMEB)); }
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int n = 256; n <= 4096; n *= 2) {
      // back-to-back launches, one event pair around the LAST 100 of 400: sustained clocks (the shader clock steps down
      // after ~3 ms of load), no per-launch event overhead
      float tl=0, ts=0; const int W=300, R=100;
      for (int r=0;r<W;++r) hipLaunchKernelGGL(k_x3, dim3(7,n), dim3(256), 0, 0, big[r&1], out);
      hipEventRecord(e0); for (int r=0;r<R;++r) hipLaunchKernelGGL(k_x3, dim3(7,n), dim3(256), 0, 0, big[r&1], out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&tl,e0,e1);
      for (int r=0;r<W;++r) hipLaunchKernelGGL(k_st, dim3(4,n), dim3(256), 0, 0, ob, (float)r, 1);
      hipEventRecord(e0); for (int r=0;r<R;++r) hipLaunchKernelGGL(k_st, dim3(4,n), dim3(256), 0, 0, ob, (float)r, 1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ts,e0,e1);
      printf("N=%4d  x3 loads %.2f us   float4 nt stores %.2f us   (per launch incl. the ~2 us dispatch gap, sustained)\n", n, tl/R*1e3, ts/R*1e3);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'c') {
    // calibration mode for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- tools/membench cal`: a few launches of each
    // known-byte-count kernel over a 1.65 GB pool (past the 256 MiB Infinity Cache), nothing else
    float4* ob; size_t ob_bytes=(size_t)N*4*1764*16; CK(hipMalloc(&ob, ob_bytes));
    const size_t n16=bytes/16, n12=bytes/12;
    for (int r=0; r<POOL; ++r) {
      uint8_t* f = bufs[r];
      hipLaunchKernelGGL(k_lin, dim3((n16+1535)/1536), dim3(256), 0, 0, (const uint4*)f, out, n16);
      hipLaunchKernelGGL(k_lin3, dim3((n12+2047)/2048), dim3(256), 0, 0, f, out, n12);
      hipLaunchKernelGGL(k_x3, dim3(7,N), dim3(256), 0, 0, f, out);
      hipLaunchKernelGGL(k_st, dim3(4,N), dim3(256), 0, 0, ob, (float)r, 0);
      hipLaunchKernelGGL(k_st, dim3(4,N), dim3(256), 0, 0, ob, (float)r, 1);
    }
    CK(hipDeviceSynchronize());
    printf("cal: k_lin reads %zu B, k_lin3 reads %zu B, k_x3 reads %zu B (algorithmic), k_st writes %zu B per launch\n",
           bytes, n12*12, (size_t)N*2*168*ROWB, ob_bytes);
    return 0;
  }
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double alg = (double)N*2*168*ROWB;      // touched rows only
  for (int kind=0; kind<4; ++kind) {
    const char* nm[]={"x3 band-WG (K1 shape)","x4 band-WG (60 lanes/row pair)","linear x4 whole frames","x3 env-WG pipelined 2-deep"};
    float best=1e9, tot=0; const int R=40;
    for (int r=-5; r<R; ++r) {
      uint8_t* f = bufs[(r+5)%POOL];
      hipEventRecord(e0);
      if (kind==0) hipLaunchKernelGGL(k_x3, dim3(7,N), dim3(256), 0, 0, f, out);
      else if (kind==1) hipLaunchKernelGGL(k_x4, dim3(7,N), dim3(256), 0, 0, f, out);
      else if (kind==2) { size_t n16=bytes/16; hipLaunchKernelGGL(k_lin, dim3((n16+1535)/1536), dim3(256), 0, 0, (const uint4*)f, out, n16); }
      else hipLaunchKernelGGL(k_x3_pipe, dim3(1,N), dim3(256), 0, 0, f, out);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
      if (r>=0) { tot+=ms; if (ms<best) best=ms; }
    }
    double b = kind==2 ? (double)bytes : alg;
    printf("%-34s avg %.2f us  best %.2f us  -> %.2f TB/s (avg), bytes %.1f MB\n", nm[kind], tot/R*1e3, best*1e3, b/(tot/R*1e-3)/1e12, b/1e6);
  }
  { float4* ob; size_t ob_bytes=(size_t)N*4*1764*16; CK(hipMalloc(&ob, ob_bytes));
    for (int nt=0; nt<2; ++nt) { float tot=0,best=1e9; const int R=40;
      for (int r=-5;r<R;++r){ hipEventRecord(e0); hipLaunchKernelGGL(k_st, dim3(4,N), dim3(256), 0, 0, ob, (float)r, nt); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(r>=0){tot+=ms; if(ms<best)best=ms;} }
      printf("store float4 K2 shape nt=%d           avg %.2f us  best %.2f us  -> %.2f TB/s, bytes %.1f MB\n", nt, tot/R*1e3, best*1e3, ob_bytes/(tot/R*1e-3)/1e12, ob_bytes/1e6); } }
  { float4* ob; size_t ob_bytes=(size_t)N*4*1764*16; CK(hipMalloc(&ob, ob_bytes));
    for (int kind=0; kind<2; ++kind) for (int nt=0; nt<2; ++nt) { float tot=0,best=1e9; const int R=40;
      for (int r=-5;r<R;++r){ hipEventRecord(e0);
        if (kind==0) hipLaunchKernelGGL(k_st128, dim3(4,N), dim3(128), 0, 0, ob, (float)r, nt); else hipLaunchKernelGGL(k_st252, dim3(4,N), dim3(256), 0, 0, ob, (float)r, nt);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(r>=0){tot+=ms; if(ms<best)best=ms;} }
      printf("store %s nt=%d  avg %.2f us  best %.2f us  -> %.2f TB/s\n", kind==0?"128thr x14 (126 lanes)":"256thr x7 (252 lanes) ", nt, tot/R*1e3, best*1e3, ob_bytes/(tot/R*1e-3)/1e12); } }
  return 0;
}
