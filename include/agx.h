/*
 * agx.h — C ABI of libagx.so: the MI355X (gfx950) observation pipeline behind
 * active-gym's Atari active-vision environments.
 *
 * The reference (elicassion/active-gym) has no FFI layer: its boundary is the
 * Python API `Atari*FovealEnv(args).reset()/step()` (reference
 * active_gym/atari_env.py:174-192, active_gym/fov_env.py:156-164,209-221,258-268,
 * 337-355).  This header is what a binding for that path would bind: every
 * entry point names the reference lines it replaces.  Plain pointers and
 * sizes only; no torch / C++ types cross the boundary.
 *
 * Conventions
 *   - every `d_*` pointer is DEVICE memory owned by the caller; nothing is
 *     allocated or freed inside ingest/observe calls (agx_create allocates the
 *     persistent per-env state, agx_destroy frees it);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     calls only enqueue work, they never synchronise;
 *   - return value: 0 = AGX_OK, negative = error; agx_last_error() returns a
 *     message for the last failing call on that context (or on NULL: for the
 *     last failing agx_create on this thread);
 *   - calls on one context are not re-entrant; one submitting thread per GPU.
 *   - there is NO CPU implementation behind this ABI: without a HIP device
 *     agx_create fails with AGX_E_HIP.
 */
#ifndef AGX_H
#define AGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_ABI_VERSION 2

#if defined(__GNUC__)
#define AGX_API __attribute__((visibility("default")))
#else
#define AGX_API
#endif

/* error codes */
#define AGX_OK            0
#define AGX_E_INVALID    -1   /* bad argument / unsupported configuration      */
#define AGX_E_HIP        -2   /* HIP runtime error (message has the HIP text)  */
#define AGX_E_NOMEM      -3
#define AGX_E_STATE      -4   /* call not valid for this context's wrapper kind */

/* which reference wrapper the context models */
#define AGX_KIND_BASE        0  /* AtariBaseEnv: RecordWrapper(AtariEnv)          atari_env.py:174-177 */
#define AGX_KIND_FIXED       1  /* FixedFovealEnv                                 fov_env.py:107-234   */
#define AGX_KIND_FLEXIBLE    2  /* FlexibleFovealEnv                              fov_env.py:240-355   */
#define AGX_KIND_PERIPHERAL  3  /* FixedFovealPeripheralEnv                       fov_env.py:358-388   */

/* observation mode, priority already resolved as in fov_env.py:176-185:
 * mask_out > resize_to_full > raw crop */
#define AGX_OUT_RAW     0
#define AGX_OUT_RESIZE  1
#define AGX_OUT_MASK    2

/* sensory_action_mode (fov_env.py:114-118,193-199) */
#define AGX_MODE_ABSOLUTE 0
#define AGX_MODE_RELATIVE 1

/* element type of the sensory-action buffer handed to agx_fovea_* */
#define AGX_DT_F32 0
#define AGX_DT_F64 1
#define AGX_DT_I32 2
#define AGX_DT_I64 3

/* FlexibleFovealEnvActionType (fov_env.py:236-238) */
#define AGX_FOV_LOC 0
#define AGX_FOV_RES 1

/* per-env ingest command byte (one per env, see agx_ingest) */
#define AGX_CMD_NVALID_MASK 0x03  /* 0,1,2 = number of sampled frames to max-pool */
#define AGX_CMD_CLEAR       0x04  /* zero the frame stack first (full reset)       */
#define AGX_CMD_SKIP        0x08  /* leave this env untouched                      */

/* kernel ids for agx_algorithmic_bytes */
#define AGX_K_INGEST      1
#define AGX_K_FOVEA       2   /* the context's own fovea kernel (fixed / flexible / peripheral) */
#define AGX_K_FULL        3
#define AGX_K_INGEST_RGB  4   /* agx_ingest_rgb */
#define AGX_K_INGEST_GRAY_RAW 5 /* agx_ingest_gray_raw */

typedef struct agx_ctx agx_ctx;

/* Mirrors the fields of `AtariEnvArgs` (atari_env.py:25-39) that reach the
 * observation path, plus the batch size.  Set struct_size = sizeof(agx_config). */
typedef struct agx_config {
    int32_t struct_size;
    int32_t device;            /* HIP device ordinal                                        */
    int32_t num_envs;          /* N                                                         */
    int32_t kind;              /* AGX_KIND_*                                                */
    int32_t raw_h, raw_w;      /* emulator screen, must be 210 x 160 (ALE)                  */
    int32_t obs_h, obs_w;      /* args.obs_size  (H, W); must be square for raw ingest, W%4==0 */
    int32_t frame_stack;       /* args.frame_stack (1..16)                                  */
    int32_t fov_h, fov_w;      /* args.fov_size          (ignored for AGX_KIND_BASE)        */
    int32_t per_h, per_w;      /* args.peripheral_res    (AGX_KIND_PERIPHERAL only)         */
    int32_t out_mode;          /* AGX_OUT_*                                                 */
    int32_t action_mode;       /* AGX_MODE_*                                                */
    int32_t antialias;         /* torchvision Resize antialias (True for >= 0.17)           */
    double  sas_lo, sas_hi;    /* args.sensory_action_space, relative mode only             */
    double  init_loc[2];       /* args.fov_init_loc (row, col); rint'ed like fov_env.py:150 */
} agx_config;

/* ---- lifetime ---------------------------------------------------------- */

AGX_API int agx_abi_version(void);

/* Identity of the built library: "libagx abi <v> src <12 hex digits>", the digits being a SHA-256 prefix over the
 * kernel and ABI sources it was compiled from (active-gym_amd/build.py).  Measurement files under profiles/ record it,
 * and bench.py refuses a PMC traffic figure collected on a different build. */
AGX_API const char *agx_build_info(void);

/* PCI address of a HIP device ("0000:72:00.0", NUL-terminated; len >= 13): what a host runner needs to find the NUMA node
 * the GPU hangs off (/sys/bus/pci/devices/<address>/numa_node, local_cpulist) and place its emulator threads and pinned
 * staging there (active_gym/hostplan.py; SURVEY.md 8e "host cores partitioned NUMA-locally"). */
AGX_API int agx_device_pci_bus_id(int device, char *buf, int len);

/* Validates the configuration (mirrors `assert fov_size < obs_size`,
 * fov_env.py:112), allocates ring / head / fov_loc / fov_res on the device,
 * builds the resize tables.  The ring starts zero-filled, fov_loc = rint(init_loc),
 * fov_res = fov_size. */
AGX_API int agx_create(const agx_config *cfg, agx_ctx **out);
AGX_API int agx_destroy(agx_ctx *ctx);
AGX_API const char *agx_last_error(const agx_ctx *ctx);

/* Shape of the observation this context's kind/out_mode produces:
 * dims = {N, frame_stack, h, w}. For AGX_KIND_FLEXIBLE + AGX_OUT_RAW (ragged
 * crops) h,w are the padded pitch obs_h,obs_w; only [0:res_h, 0:res_w] is data. */
AGX_API int agx_obs_shape(const agx_ctx *ctx, int32_t dims[4]);

/* Algorithmic (smallest lossless) HBM bytes one launch of the given kernel
 * moves for all N envs; for AGX_KIND_FLEXIBLE it uses
 * the nominal fov_size window. */
AGX_API int64_t agx_algorithmic_bytes(const agx_ctx *ctx, int kernel_id);

/* ---- K1: frame ingest --------------------------------------------------
 * One `state_buffer.append(observation)` per env (atari_env.py:121-133 for a
 * step, :80-82,:91,:111-112 for a reset):
 *   gray  = ALE luminance of the RGB screen          (getScreenGrayscale, atari_env.py:74)
 *   small = cv2.resize(gray, obs_size, INTER_LINEAR) (8-bit fixed point,  atari_env.py:74)
 *   obs   = max over the first nvalid of the two frames sampled at t==2 / t==3
 *           (zeros when nvalid == 0)                 (atari_env.py:121-132)
 *   CLEAR = zero-fill the stack before the append    (_reset_buffer, atari_env.py:80-82)
 * d_frames: u8 [N][2][raw_h][raw_w][3] RGB (HWC, as getScreenRGB returns it)
 * d_cmd   : u8 [N] command bytes (AGX_CMD_*)
 * The stack is kept as u8 numerators k of the reference's float32 k/255. */
AGX_API int agx_ingest(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, void *stream);

/* Same as agx_ingest, but from ALE's own grayscale screens: d_gray u8 [N][2][raw_h][raw_w] as
 * `ale.getScreenGrayscale()` returns them - what the reference itself reads (atari_env.py:74).  No luminance
 * arithmetic on the device (ALE's palette table has done it) and a third of the bytes over PCIe and HBM. */
AGX_API int agx_ingest_gray_raw(agx_ctx *ctx, const uint8_t *d_gray, const uint8_t *d_cmd, void *stream);

/* ---- compact source screens ---------------------------------------------------------------------------------
 * cv2.resize(.., INTER_LINEAR) from raw_h to obs_h rows reads only some source rows (168 of 210 for 84 rows: the y0 / y1 of
 * its table, atari_env.py:74); SURVEY.md 8d's algorithmic bytes count only those.  A host runner that stages only those rows
 * cuts the PCIe bytes of a step by the same 20 %.
 * agx_source_rows: rows i32 [raw_h] out (the first *n are valid, ascending; rows may be NULL to ask for n only).
 * agx_ingest_compact / agx_ingest_gray_raw_compact: agx_ingest / agx_ingest_gray_raw (same results, bit for bit) from
 *   d_rows u8 [N][2][n][raw_w][3]  (gray: [N][2][n][raw_w]) - row k of a compact screen = source row rows[k]. */
AGX_API int agx_source_rows(const agx_ctx *ctx, int32_t *rows, int32_t *n);
AGX_API int agx_ingest_compact(agx_ctx *ctx, const uint8_t *d_rows, const uint8_t *d_cmd, void *stream);
AGX_API int agx_ingest_gray_raw_compact(agx_ctx *ctx, const uint8_t *d_rows, const uint8_t *d_cmd, void *stream);

/* Same append, but from frames that are already obs-sized gray u8
 * [N][2][obs_h][obs_w] (sources that render at obs_size, e.g. the DMC path
 * dmc_env.py:175-186, and tests). */
AGX_API int agx_ingest_gray(agx_ctx *ctx, const uint8_t *d_small, const uint8_t *d_cmd, void *stream);

/* Measurement aid: the NEXT launch of the given kernel family (AGX_K_INGEST: agx_ingest / agx_ingest_gray_raw /
 * agx_ingest_rgb; AGX_K_FOVEA: agx_fovea_*) stamps `start_event` / `stop_event` (hipEvent_t created with timing) with
 * the begin / end of that kernel's execution (hipExtLaunchKernelGGL) - the interval rocprofv3's kernel trace reports,
 * with no extra packets on the stream.  One-shot; pass NULL, NULL to disarm.  The opt-in variant launches ignore it. */
AGX_API int agx_profile_next(agx_ctx *ctx, int kernel_id, void *start_event, void *stop_event);

/* DMC pixel front end (reference dmc_env.py:175-186,211-234): d_frames u8 [N][obs_h][obs_w][3] are the
 * obs-sized `physics.render()` images (RGB); the reference runs `cv2.cvtColor(obs, cv2.COLOR_BGR2GRAY)` on them,
 * i.e. OpenCV's fixed-point luma with channel 0 weighted as blue, then `/255` and one append per step (no max-pool,
 * no resize).  cmd as for agx_ingest (NVALID 0 appends zeros, CLEAR = _reset_buffer, SKIP).  gray_mode selects the
 * OpenCV generation's coefficients (AGX_GRAY_*). */
#define AGX_GRAY_CV15 0   /* OpenCV 4.x : (3735 c0 + 19235 c1 + 9798 c2 + 16384) >> 15 */
#define AGX_GRAY_CV14 1   /* OpenCV <=3 : (1868 c0 +  9617 c1 + 4899 c2 +  8192) >> 14 */
AGX_API int agx_ingest_rgb(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, int gray_mode, void *stream);

/* ---- K0: base observation ----------------------------------------------
 * `np.stack(state_buffer, 0)` (atari_env.py:114,143), oldest -> newest, as
 * float32 k/255.  d_obs: f32 [N][fs][obs_h][obs_w]. */
AGX_API int agx_observe_full(agx_ctx *ctx, float *d_obs, void *stream);

/* Test/checkpoint access to the stack in the same order, as u8 numerators. */
AGX_API int agx_get_stack_u8(agx_ctx *ctx, uint8_t *d_out, void *stream);
AGX_API int agx_set_stack_u8(agx_ctx *ctx, const uint8_t *d_in, void *stream);

/* ---- fovea state --------------------------------------------------------
 * `_init_fov_loc` / `_init_fov_res` (fov_env.py:149-150,250-251) for the envs
 * whose mask byte is non-zero (d_mask == NULL: all envs). */
AGX_API int agx_fovea_reset(agx_ctx *ctx, const uint8_t *d_mask, void *stream);
AGX_API int agx_get_fov_state(agx_ctx *ctx, int32_t *d_fov_loc /*[N][2]*/, int32_t *d_fov_res /*[N][2], may be NULL*/, void *stream);
AGX_API int agx_set_fov_state(agx_ctx *ctx, const int32_t *d_fov_loc, const int32_t *d_fov_res /*may be NULL*/, void *stream);

/* ---- K2: FixedFovealEnv._fov_step (fov_env.py:166-203) ------------------
 * d_action: [N][2] (row, col) of `action_dtype`, or NULL = keep fov_loc (the
 *           `reset()` observation, fov_env.py:156-160).
 *   absolute: fov_loc = rint(clip(a, 0, obs - fov))
 *   relative: fov_loc = rint(clip(fov_loc + rint(clip(a, sas_lo, sas_hi)), 0, obs - fov))
 * then crop / mask-out paste / bilinear resize per out_mode.
 * d_obs     : f32, shape per agx_obs_shape
 * d_fov_loc : i32 [N][2] out, may be NULL          (info["fov_loc"], fov_env.py:217)
 * d_mask    : u8 [N] or NULL; envs with 0 are left untouched (state and d_obs rows). */
AGX_API int agx_fovea_fixed(agx_ctx *ctx, const void *d_action, int action_dtype, const uint8_t *d_mask,
                    float *d_obs, int32_t *d_fov_loc, void *stream);

/* ---- one call for the whole step: agx_ingest followed by agx_fovea_fixed (d_mask = NULL), same results --------
 * One whole `FixedFovealEnv.step` image path (fov_env.py:209-221 over atari_env.py:119-148) for all envs: the two
 * stand-alone launches, in one ABI call.  (Other forms of this call - a heterogeneous fused launch, env-range parts on
 * internal streams, one workgroup per env - were built, measured equal or slower and live in the experiments build
 * libagx_exp.so only; this library has no knob for them.)
 * mid_event: optional hipEvent_t (may be NULL) recorded on `stream` between the two launches (profiling). */
AGX_API int agx_step_fixed(agx_ctx *ctx, const uint8_t *d_frames, const uint8_t *d_cmd, const void *d_action,
                           int action_dtype, float *d_obs, int32_t *d_fov_loc, void *mid_event, void *stream);

/* ---- K3: FixedFovealPeripheralEnv._get_fov_state (fov_env.py:375-388) ---
 * whole stack squeezed to peripheral_res and expanded back, full-res fovea pasted. */
AGX_API int agx_fovea_peripheral(agx_ctx *ctx, const void *d_action, int action_dtype, const uint8_t *d_mask,
                         float *d_obs, int32_t *d_fov_loc, void *stream);

/* ---- K4: FlexibleFovealEnv._fov_step (fov_env.py:270-330) ---------------
 * d_action_type: i32 [N] AGX_FOV_LOC / AGX_FOV_RES (NULL = all FOV_LOC).
 *   FOV_RES: fov_res = action (the reference stores it unclipped and only
 *   integer values inside [1, obs] are usable there; this ABI rints float
 *   input and clamps to [1, obs] — documented normalisation), then fov_loc is
 *   re-clipped to obs - fov_res.
 * iff fov_res rows > fov_size rows: crop is resized to fov_size and back
 * (fov_env.py:276-287), then mask / resize-to-obs / raw per out_mode. */
AGX_API int agx_fovea_flexible(agx_ctx *ctx, const void *d_action, int action_dtype, const int32_t *d_action_type,
                       const uint8_t *d_mask, float *d_obs, int32_t *d_fov_loc, int32_t *d_fov_res,
                       void *stream);

/* K4, raw-crop mode with the ragged crops PACKED (the reference returns [fs, res_h, res_w] per env, fov_env.py:283-298):
 * same state update as agx_fovea_flexible, then
 *   d_offsets i64 [N + 1] out : env n's crops f32 [fs][res_h][res_w] (tight) start at d_packed + d_offsets[n];
 *                               d_offsets[N] = total floats = sum_n fs * res_h[n] * res_w[n]
 *   d_packed  f32 [capacity_floats] out; an env whose crops would end past the capacity is not written
 *                               (d_offsets[N] > capacity_floats tells the caller; N * fs * obs_h * obs_w always fits)
 * Context: AGX_KIND_FLEXIBLE with AGX_OUT_RAW.  No mask: the layout of every env depends on every resolution.
 * Two launches: the state update of every env + level 1 of an exclusive scan of the crop sizes (256 envs per workgroup,
 * scratch owned by the context since agx_create - nothing is allocated here), then the crops, whose workgroups add the
 * totals of the scan blocks before theirs and write d_offsets. */
AGX_API int agx_fovea_flexible_packed(agx_ctx *ctx, const void *d_action, int action_dtype, const int32_t *d_action_type,
                                      float *d_packed, int64_t capacity_floats, int64_t *d_offsets, int32_t *d_fov_loc,
                                      int32_t *d_fov_res, void *stream);

/* One whole step of such a context - agx_ingest* followed by agx_fovea_flexible_packed (atari_env.py:119-148, then
 * fov_env.py:300-330, 283-298) - as ONE call and TWO launches: the state update + scan reads the actions and the old fov state and
 * nothing the ingest writes, so its ceil(N / 256) workgroups ride in the ingest launch (first rows of its grid) instead of costing a
 * launch of their own (4.9 us of 49 at N = 1024).  Same results as the two calls, bit for bit.
 *   d_screens / screens : the two sampled screens of every env in the layout `screens` names: 0 = whole RGB screens as agx_ingest
 *                         takes them, AGX_SCREENS_GRAY = ALE grayscale screens (agx_ingest_gray_raw), | AGX_SCREENS_COMPACT = only the
 *                         rows agx_source_rows lists (agx_ingest_compact / agx_ingest_gray_raw_compact)
 *   the rest            : as agx_ingest (d_cmd) and agx_fovea_flexible_packed
 * Geometries without the headline ingest plan (12-row bands) or crop plan take the three launches of the stand-alone entry
 * points inside this call (AGX_STEP_PACKED_UNFUSED=1 forces that, for A/B runs). */
#define AGX_SCREENS_GRAY 1
#define AGX_SCREENS_COMPACT 2
AGX_API int agx_step_flexible_packed(agx_ctx *ctx, const uint8_t *d_screens, int screens, const uint8_t *d_cmd,
                                     const void *d_action, int action_dtype, const int32_t *d_action_type, float *d_packed,
                                     int64_t capacity_floats, int64_t *d_offsets, int32_t *d_fov_loc, int32_t *d_fov_res,
                                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H */
