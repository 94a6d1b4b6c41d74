// agx_step_env.h - the whole image path of one FixedFovealEnv.step (atari_env.py:119-148 + fov_env.py:187-221) for one
// env in ONE workgroup: the seven ingest bands of the env, then the fovea of its frame stack - no launch boundary between
// ingest and fovea, no ramp / drain between them, and no cross-workgroup dependency (the env's new frame is written to
// the ring by this workgroup and read back by this workgroup through L2).
//
// grid = N, block = 512 = two TEAMS of 256 threads (waves 0-3 / 4-7).  A team runs the stand-alone kernels' bodies
// unchanged (ingest_band<256, false, 12>, fovea_fixed_body<G, RESIZE>) on its own half of the LDS; the workgroup walks a
// short schedule of ROUNDS, one job per team and round:
//     round 0..3 : bands (0,1) (2,3) (4,5) (6, fovea of an untouched slot)
//     fence      : every wave waits for its ring stores (s_waitcnt vmcnt(0)), then the round barrier
//     round 4..5 : fovea of the written slot (frame read with agent-scope loads: L2, never a stale L1 line - the last line of
//                  an older slot, possibly cached by an earlier round, also holds the first bytes of the next slot) and of
//                  the remaining untouched slots
// (full reset: every slot is rewritten, all foveas follow the fence; skipped env: no bands, four foveas in two rounds;
//  odd envs run the untouched-slot foveas BEFORE their bands, so that a CU's workgroups are not all loading, then all
//  storing, at the same moments).
// Every wave executes exactly three s_barrier per round whatever its job (1 inside a band, 2 inside a fovea, the rest
// padding), so the two teams' barriers always pair up.  LDS 2 x 19.5 KB -> 4 workgroups = 32 waves per CU.
// Bit-identical to agx_ingest + agx_fovea_fixed (same bodies, same order of arithmetic).
//
// MEASURED SLOWER, hence opt-in (AGX_STEP_ENV=1).  Same box, N = 1024, 600 steps: 74.9 us per step against 61.1-61.6 us for
// the two stand-alone launches.  With the jobs switched off selectively (AGX_STEP_ENV_DEBUG): schedule + barriers +
// parameter reloads alone 10.9 us; bands only 51.5 us (k_ingest_full12 alone: 37.5); foveas only 29.7 us (k_fovea_fixed
// alone: 23.5): the phases do not overlap and each is slower than its stand-alone kernel.  The stand-alone kernels live
// on statistical multiplexing - 8 independent 4-wave workgroups per CU, 3.5 rounds of them, each in a different phase, so
// loads, VALU and stores of different workgroups overlap - and on one or two cheap 4-wave barriers per workgroup; here
// 4 eight-wave workgroups per CU walk six rounds in step, pay ~19 eight-wave barriers each, and a round lasts as long as
// its slower team.  Alternating the job order between odd and even envs changed nothing (74.5 us); a staggered start
// made it monotonically slower (77-92 us).  DESIGN.md section 3, "One workgroup per env".
#pragma once
#include "../agx_k1_ingest.h"
#include "../agx_k2_fixed.h"

namespace agx {

// copy of a parameter block out of the kernarg segment (constant address space: scalar loads), dword by dword
template <class T>
__device__ __forceinline__ T load_kernarg(const T __attribute__((address_space(4))) *p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized parameter blocks");
    T out;
    const uint32_t __attribute__((address_space(4))) *src = (const uint32_t __attribute__((address_space(4))) *)p;
    uint32_t *dst = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return out;
}

struct StepEnvArgs {
    IngestParams pi;
    FovParams pf;
    int32_t team_lds;
    int32_t debug;        // diagnosis only (AGX_STEP_ENV_DEBUG): bit 0 drops the band jobs, bit 1 the fovea jobs
};

template <class G>
__global__ __launch_bounds__(2 * kThreads, 8) void k_step_env(StepEnvArgs args_unused) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The two bodies' parameter blocks are ~55 SGPRs together; kept alive across the whole round loop they push the kernel
    // past the SGPR file (234 v_writelane / v_readlane spills, 90 VGPRs).  So the by-value argument is never touched:
    // every round reads the block it needs from the kernarg segment (scalar loads) through a pointer laundered per
    // round, and its registers die with the round.
    typedef const StepEnvArgs __attribute__((address_space(4))) *KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = blockIdx.x;
    const int tid = threadIdx.x & (kThreads - 1);
    const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    unsigned char *tsm = smem + team * ka->team_lds;
    const int fs = ka->pf.fs;
    constexpr int kBands = 7;                                      // 84 rows / 12

    const uint32_t cmd = uniform_load_u8(ka->pi.cmd + n);
    const int h = uniform_load_i32(ka->pi.head_in + n);
    const bool skip = (cmd & AGX_CMD_SKIP) != 0, clear = (cmd & AGX_CMD_CLEAR) != 0 && !skip;
    if (skip && threadIdx.x == 0) ka->pi.head_out[n] = h;          // (band 0 writes it otherwise)
    const int nb = skip ? 0 : kBands;
    const int n_early = skip ? fs : (clear ? 0 : fs - 1);          // slots this step's ingest does not touch
    const int n_late = skip ? 0 : (clear ? fs : 1);                // slots it writes: after the fence, coherent loads
    auto early = [&](int i) { return (skip || i < h) ? i : i + 1; };   // i-th slot != h (every slot when skipped)
    auto late = [&](int i) { return clear ? i : h; };
    // Two job orders, alternating with the env index, so that the workgroups of a CU are not all in the same phase at the
    // same time (all of them loading, then all of them storing):
    //   A: bands 0..6 and one untouched-slot fovea beside the odd last band | fence | written slot, remaining foveas
    //   B: the untouched-slot foveas first, then the bands                  | fence | written slot
    const bool order_b = (n & 1) != 0;
    const int e_pre = order_b ? n_early : (((nb & 1) && n_early > 0) ? 1 : 0);   // early foveas before the fence
    const int pre_jobs = nb + e_pre, post_jobs = n_late + n_early - e_pre;
    const int pre_rounds = (pre_jobs + 1) >> 1;
    const int rounds = pre_rounds + ((post_jobs + 1) >> 1);

    // staggered start (diagnosis: AGX_STEP_ENV_DEBUG bits 8..15 = units of 1024 cycles per step of (n & 3))
    for (int d = ((ka->debug >> 8) & 0xFF) * (n & 3); d > 0; --d) __builtin_amdgcn_s_sleep(16);

    for (int r = 0; r < rounds; ++r) {
        // job of (round r, this team): 0 = none, 1 = band, 2 = fovea, 3 = fovea with coherent frame loads
        int kind = 0, arg = 0;
        if (r < pre_rounds) {
            const int k = 2 * r + team;
            if (order_b) {
                if (k < e_pre) kind = 2, arg = early(k);
                else if (k - e_pre < nb) kind = 1, arg = k - e_pre;
            } else {
                if (k < nb) kind = 1, arg = k;
                else if (k - nb < e_pre) kind = 2, arg = early(k - nb);
            }
        } else {
            const int k = 2 * (r - pre_rounds) + team;
            if (k < n_late) kind = 3, arg = late(k);
            else if (k - n_late + e_pre < n_early) kind = 2, arg = early(k - n_late + e_pre);
        }
        if (r == pre_rounds && nb > 0) {
            // the ring stores of this workgroup's bands are complete (in L2) before any wave passes the next barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // the thread index and the argument pointer are laundered once per round: everything a body derives from them
        // (row / column roles, tap loads, its parameter block) is recomputed inside the round instead of being hoisted
        // out of the loop and kept alive across the other bodies
        int t = tid;
        asm volatile("" : "+v"(t));
        KArgs a = ka;
        asm volatile("" : "+s"(a));
        const int dbg = ka->debug;
        if ((dbg & 1) && kind == 1) kind = 0;
        if ((dbg & 2) && kind >= 2) kind = 0;
        int barriers = 3;
        if (kind == 1) {
            const IngestParams pi = load_kernarg(&a->pi);
            ingest_band<kThreads, false, 12>(pi, arg, n, tsm, t);
            barriers -= 1;
        } else if (kind == 2) {
            const FovParams pf = load_kernarg(&a->pf);
            fovea_fixed_body<G, AGX_OUT_RESIZE, false>(G{}, pf, arg, n, tsm, t);
            barriers -= 2;
        } else if (kind == 3) {
            const FovParams pf = load_kernarg(&a->pf);
            fovea_fixed_body<G, AGX_OUT_RESIZE, true>(G{}, pf, arg, n, tsm, t);
            barriers -= 2;
        }
        for (int k = 0; k < barriers; ++k) __syncthreads();        // padding: three barriers per round for every wave
    }
}

}  // namespace agx
