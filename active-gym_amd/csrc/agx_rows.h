// agx_rows.h - HOST side: separable resample passes as sparse row operators, and their composition.
//
// Every torchvision Resize on the reference's path (fov_env.py:120,182,248,277-279,295,366-368,376) is, per axis,
// a linear map with a few consecutive taps per output index.  Chains of them along ONE axis
//   FlexibleFovealEnv:  Resize(fov_size) -> Resize(fov_res) -> Resize(obs_size)      fov_env.py:276-298
//   Peripheral:         Resize(peripheral_res) -> Resize(obs_size)                   fov_env.py:366-368,375-377
// are composed here in double (weights normalised exactly as ATen does for float64 input), so that the kernels
// evaluate one banded operator per axis instead of two or three passes with a barrier between each.  Passes on
// different axes commute; only float rounding differs from the reference's order of operations (~1e-7).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace agx {
namespace rows {

struct Row {
    int lo = 0;                  // first input index
    std::vector<double> w;       // weights of inputs lo, lo+1, ...
};
using Op = std::vector<Row>;     // one Row per output index

// One axis of torchvision Resize = ATen upsample_bilinear2d(align_corners=False): antialiased triangle filter when
// down-scaling with antialias on (_compute_indices_min_size_weights_aa), plain bilinear otherwise
// (area_pixel_compute_source_index).  n_in == n_out gives the identity either way.
inline Op resize_axis(int n_in, int n_out, bool antialias) {
    Op op(n_out);
    const double scale = (double)n_in / (double)n_out;
    const bool aa = antialias && n_in > n_out;
    for (int i = 0; i < n_out; ++i) {
        Row &r = op[i];
        if (aa) {
            const double support = scale, invscale = 1.0 / scale, center = scale * (i + 0.5);
            long long xmin = (long long)(center - support + 0.5);
            if (xmin < 0) xmin = 0;
            long long xmax = (long long)(center + support + 0.5);
            if (xmax > n_in) xmax = n_in;
            double total = 0.0;
            for (long long jx = xmin; jx < xmax; ++jx) {
                double x = ((double)jx - center + 0.5) * invscale;
                if (x < 0) x = -x;
                const double wv = x < 1.0 ? 1.0 - x : 0.0;
                r.w.push_back(wv);
                total += wv;
            }
            if (total != 0.0)
                for (double &v : r.w) v /= total;
            r.lo = (int)xmin;
        } else {
            double f = scale * (i + 0.5) - 0.5;
            if (f < 0.0) f = 0.0;
            int i0 = (int)f;
            if (i0 > n_in - 1) i0 = n_in - 1;
            const int i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
            const double l1 = f - i0;
            if (i1 == i0) r.w = {1.0};
            else r.w = {1.0 - l1, l1};
            r.lo = i0;
        }
    }
    return op;
}

// drop leading / trailing zero weights (keeps at least one tap)
inline void trim(Row &r) {
    size_t a = 0, b = r.w.size();
    while (b - a > 1 && r.w[b - 1] == 0.0) --b;
    while (b - a > 1 && r.w[a] == 0.0) ++a;
    r.w = std::vector<double>(r.w.begin() + a, r.w.begin() + b);
    r.lo += (int)a;
}

// (A o B)[i] = sum_k A[i][k] * B[k]: apply B first (n_in -> n_mid), then A (n_mid -> n_out)
inline Op compose(const Op &A, const Op &B) {
    Op out(A.size());
    for (size_t i = 0; i < A.size(); ++i) {
        int lo = 1 << 30, hi = -1;
        for (size_t k = 0; k < A[i].w.size(); ++k) {
            const Row &b = B[A[i].lo + (int)k];
            lo = std::min(lo, b.lo);
            hi = std::max(hi, b.lo + (int)b.w.size());
        }
        Row &r = out[i];
        r.lo = lo;
        r.w.assign(hi - lo, 0.0);
        for (size_t k = 0; k < A[i].w.size(); ++k) {
            const Row &b = B[A[i].lo + (int)k];
            for (size_t t = 0; t < b.w.size(); ++t) r.w[b.lo - lo + t] += A[i].w[k] * b.w[t];
        }
        trim(r);
    }
    return out;
}

inline int max_taps(const Op &op) {
    size_t m = 1;
    for (const Row &r : op) m = std::max(m, r.w.size());
    return (int)m;
}

// Shift rows so that lo + T - 1 <= n_in - 1 where n_in >= T (leading zeros are added), i.e. a kernel that reads
// exactly T consecutive inputs from `lo` never leaves [0, n_in).  For n_in < T: lo = 0 and the kernel is
// responsible for the T - n_in inputs it reads past the end being FINITE (their weights are 0).
inline void fit(Op &op, int n_in, int T) {
    for (Row &r : op) {
        int lo = r.lo;
        if (n_in >= T) lo = std::min(lo, n_in - T);
        else lo = 0;
        if (lo != r.lo) {
            r.w.insert(r.w.begin(), (size_t)(r.lo - lo), 0.0);
            r.lo = lo;
        }
    }
}

}  // namespace rows
}  // namespace agx
