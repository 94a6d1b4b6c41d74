// agx_taps.h — resampling tap tables shared by host table builders and device kernels.
//
// Two third-party resamplers sit on the reference's observation path:
//   * OpenCV 8-bit INTER_LINEAR (cv2.resize in AtariEnv._get_state, reference
//     atari_env.py:73-75): 11-bit fixed-point coefficients, built on the host only.
//   * torchvision Resize on float tensors (reference fov_env.py:120,182,248,277-279,
//     366-368) = ATen upsample_bilinear2d(align_corners=False), plain or antialiased.
// All index/fraction arithmetic is done in double like ATen does for float64 input;
// only the final weights are narrowed to float.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agx {

// One destination index of a separable float resample pass (16 bytes).
//  lin2 : out = a*src[lo] + b*src[aux]                (aux = second index)
//  aa   : out = b * sum_{k<aux} tri((k - a + .5)*inv) * src[lo+k]   (a = center - lo, b = 1/total)
struct Tap {
    int32_t lo;
    int32_t aux;
    float a;
    float b;
};

// ATen area_pixel_compute_source_index + compute_source_index_and_lambda (align_corners=False).
__host__ __device__ inline Tap make_tap_lin2(int i, int n_in, int n_out) {
    const double scale = (double)n_in / (double)n_out;
    double f = scale * ((double)i + 0.5) - 0.5;
    if (f < 0.0) f = 0.0;
    int i0 = (int)f;                       // f >= 0: trunc == floor
    if (i0 > n_in - 1) i0 = n_in - 1;
    const int i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
    const double l1 = f - (double)i0;
    Tap t;
    t.lo = i0;
    t.aux = i1;
    t.a = (float)(1.0 - l1);
    t.b = (float)l1;
    return t;
}

// ATen _compute_indices_min_size_weights_aa, bilinear (triangle) filter, only used for
// down-scaling (n_in > n_out); for n_in <= n_out it equals the plain taps.
__host__ __device__ inline Tap make_tap_aa(int i, int n_in, int n_out, float *inv_out) {
    const double scale = (double)n_in / (double)n_out;
    const double support = scale;                 // interp_size/2 * scale, scale >= 1
    const double invscale = 1.0 / scale;
    const double center = scale * ((double)i + 0.5);
    long long xmin = (long long)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    long long xmax = (long long)(center + support + 0.5);
    if (xmax > n_in) xmax = n_in;
    long long xsize = xmax - xmin;
    if (xsize < 0) xsize = 0;
    double total = 0.0;
    for (long long j = 0; j < xsize; ++j) {
        double x = ((double)(j + xmin) - center + 0.5) * invscale;
        if (x < 0.0) x = -x;
        const double w = x < 1.0 ? 1.0 - x : 0.0;
        total += w;
    }
    Tap t;
    t.lo = (int)xmin;
    t.aux = (int)xsize;
    t.a = (float)(center - (double)xmin);
    t.b = (float)(total != 0.0 ? 1.0 / total : 0.0);
    *inv_out = (float)invscale;
    return t;
}

}  // namespace agx
