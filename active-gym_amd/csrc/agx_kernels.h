// agx_kernels.h — gfx950 device code of the observation pipeline (HBM-bound
// byte/float streaming work: no MFMA anywhere, there is no dense contraction).
//
//   k_ingest            K1  RGB -> ALE luminance -> OpenCV fixed-point bilinear -> 2-frame max -> u8 ring slot
//   k_ingest_gray       K1' same append from already obs-sized gray frames
//   k_ingest_rgb        K1b DMC front end: obs-sized RGB -> cv2 BGR2GRAY fixed point -> ring slot (no max, no resize)
//   k_stack_u8 / k_full K0  ring -> stack order (u8 / f32 k/255)
//   k_fovea_fixed       K2  clip/rint sensory action, crop, {raw | mask-out | bilinear upsample}
//   k_fovea_generic     K3/K4 peripheral squeeze-expand + paste, flexible (ragged) fovea
//
// Persistent per-env state (owned by the context, see agx_api.hip):
//   ring  u8 [N][fs][oh][ow]   numerators k of the reference's float32 k/255 frames
//   head  i32[2][N]            next slot to write == oldest frame; double-buffered so that the
//                              several workgroups of one env all read the pre-launch value
//   loc   i32[2][N][2], res i32[2][N][2]   fov_loc / fov_res, double-buffered for the same reason
#pragma once
#include "agx_common.h"
#include "agx_k0_stack.h"
#include "agx_k1_ingest.h"
#include "agx_fov_common.h"
#include "agx_k2_fixed.h"
#include "agx_k34_resample.h"
#include "agx_k3_per3.h"
#include "agx_k4_flex3.h"
#include "agx_k4_raw3.h"
#include "agx_k14_step_packed.h"
#ifdef AGX_EXPERIMENTS
#include "experiments/agx_experiments.h"   // measured dead ends: tools/ and the variants test only, never in libagx.so
#include "experiments/agx_packed_wave.h"
#endif
