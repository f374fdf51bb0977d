// tdeq_kernels_complex.hpp — the norm kernels for complex64 / complex128 states (gfx950).
//
// Everything LINEAR in a Runge–Kutta step has real coefficients (rk_common.py:79,89,201-205, interp.py), so a complex
// state goes through the real kernels of tdeq_kernels.hpp on its interleaved (re, im) view — per-component rounding is
// what ATen does for complex * real, tools/complex_abs_probe.py.  The two places where a complex number is more
// than two reals are the tolerance-scaled norms (misc.py:80-82 with |.| = complex abs; misc.py:50-56,68):
//
//     tol = atol + rtol * max(|y0|, |y1|)        r = err / tol        sum += |r|^2
//
// ATen forms them as (measured on the MI355X and on the CPU, profiles/r04_complex_abs_probe.json, 2^20 values each):
//     |z|  = hypot(re, im)                    — the device libm's hypot, bit for bit (NOT sqrt(re*re + im*im))
//     z / real = (re * (1/real), im * (1/real)) — multiplication by the rounded reciprocal
// and so do these kernels, in T = the real type of the state; what is accumulated (in fp64, like the real kernels' r^2) is
// re(r)^2 + im(r)^2 formed in double, not the square of a rounded modulus.
// Layout: the segment table counts COMPLEX elements; a chunk of `chunk` elements is 2*chunk reals.  16 B per lane
// (two complex64 or one complex128) when every stream is 16-byte aligned, one element per lane otherwise.
#pragma once

#include "tdeq_kernels.hpp"

namespace tdeq {

__device__ __forceinline__ float chyp(float re, float im) { return hypotf(re, im); }
__device__ __forceinline__ double chyp(double re, double im) { return hypot(re, im); }

// |r|^2 for the fp64 accumulation: re^2 + im^2 formed in double — exact for float components, one rounding for double
// ones — instead of squaring a rounded modulus (the reference squares fl_T(|r|) and accumulates in T: both lose more)
template <typename T>
__device__ __forceinline__ double cabs2(T re, T im) { return (double)re * (double)re + (double)im * (double)im; }

template <typename T>
__device__ __forceinline__ bool cfinite(T re, T im) { return __builtin_isfinite(re) && __builtin_isfinite(im); }

// One complex element of the error norm; returns r = err / tol through (r_re, r_im).
template <typename T>
__device__ __forceinline__ void cplx_tol_accumulate(T e_re, T e_im, T y0r, T y0i, T y1r, T y1i, T rtol, T atol,
                                                    double& acc, double& bad, T& r_re, T& r_im) {
    const T tol = atol + rtol * smax(chyp(y0r, y0i), chyp(y1r, y1i));
    const T inv = (T)1 / tol;
    r_re = e_re * inv;
    r_im = e_im * inv;
    acc += cabs2(r_re, r_im);
    bad += (cfinite(y0r, y0i) && cfinite(y1r, y1i)) ? 0.0 : 1.0;
}

template <typename T, int NT>
struct CplxErrArgs {
    const T* partial;     // PARTIAL: the error sum over the step's leading stages (tdeq_stage_combine_err on the real view)
    const T* y0;
    const T* y1;
    const T* k[NT > 0 ? NT : 1];
    T c[NT > 0 ? NT : 1];
    SegTable st;          // numel / chunk in complex elements
    double* part_sumsq;   // [n_chunks]
    double* part_bad;     // [n_chunks]
    T* scaled;            // WRITE: err / tol per element (padding zero-filled)
    const double* dt_dev; // hipGraph mode: c[] holds fl_T(coef), multiplied by T(*dt_dev) here
};

// err = (c0 k0 + c1 k1) + ...   or   (partial + c0 k0) + ...  — componentwise, the real kernels' order and rounding.
template <typename T, int NT, bool VEC, bool PARTIAL, bool WRITE>
__global__ __launch_bounds__(kBlock) void cplx_error_norm_kernel(const CplxErrArgs<T, NT> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = 2 * b * a.st.chunk;                  // in reals
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const int64_t nr = 2 * valid;                             // reals of this chunk that belong to the segment
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    T cc[NT > 0 ? NT : 1];
    {
        const T dtT = a.dt_dev ? (T)*a.dt_dev : (T)1;
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.dt_dev ? a.c[j] * dtT : a.c[j];
    }
    double acc[2] = {0.0, 0.0};
    int64_t t0 = 0;
    if (VEC) {
        const int64_t nv = nr / L;
        const V* y0 = reinterpret_cast<const V*>(a.y0 + base);
        const V* y1 = reinterpret_cast<const V*>(a.y1 + base);
#pragma unroll 2
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            V e;
            if (PARTIAL) {
                e = reinterpret_cast<const V*>(a.partial + base)[i];
#pragma unroll
                for (int j = 0; j < NT; ++j) e = e + reinterpret_cast<const V*>(a.k[j] + base)[i] * cc[j];
            } else {
                e = reinterpret_cast<const V*>(a.k[0] + base)[i] * cc[0];
#pragma unroll
                for (int j = 1; j < NT; ++j) e = e + reinterpret_cast<const V*>(a.k[j] + base)[i] * cc[j];
            }
            const V v0 = y0[i], v1 = y1[i];
            V r;
#pragma unroll
            for (int q = 0; q < L; q += 2) {
                T rr, ri;
                cplx_tol_accumulate<T>(e[q], e[q + 1], v0[q], v0[q + 1], v1[q], v1[q + 1], rtol, atol, acc[0], acc[1], rr, ri);
                r[q] = rr;
                r[q + 1] = ri;
            }
            if (WRITE) reinterpret_cast<V*>(a.scaled + base)[i] = r;
        }
        t0 = nv * L;
    }
    for (int64_t t = t0 + 2 * (int64_t)threadIdx.x; t < nr; t += 2 * kBlock) {
        T er, ei;
        if (PARTIAL) {
            er = a.partial[base + t];
            ei = a.partial[base + t + 1];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                er = er + a.k[j][base + t] * cc[j];
                ei = ei + a.k[j][base + t + 1] * cc[j];
            }
        } else {
            er = a.k[0][base + t] * cc[0];
            ei = a.k[0][base + t + 1] * cc[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                er = er + a.k[j][base + t] * cc[j];
                ei = ei + a.k[j][base + t + 1] * cc[j];
            }
        }
        T rr, ri;
        cplx_tol_accumulate<T>(er, ei, a.y0[base + t], a.y0[base + t + 1], a.y1[base + t], a.y1[base + t + 1], rtol, atol,
                               acc[0], acc[1], rr, ri);
        if (WRITE) {
            a.scaled[base + t] = rr;
            a.scaled[base + t + 1] = ri;
        }
    }
    if (WRITE && a.st.n_seg > 1)   // zero the padding of a segmented layout
        for (int64_t t = nr + threadIdx.x; t < 2 * a.st.chunk; t += kBlock) a.scaled[base + t] = (T)0;
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        a.part_sumsq[b] = acc[0];
        a.part_bad[b] = acc[1];
    }
}

// Initial-step norms (misc.py:50-56,68): scale = atol + |y| * rtol; quotients = value * (1 / scale).
//   MODE 0: acc0 += |a/scale|^2 ; acc1 += |b/scale|^2        MODE 1: acc0 += |(a-b)/scale|^2
// OUT: also store the quotients (user norm callables: tdeq_init_scaled) — then no sums are written.
template <typename T>
struct CplxInitArgs {
    const T* a;
    const T* b;
    const T* y;
    SegTable st;
    double* part0;
    double* part1;
    double* part_bad;
    T* out0;
    T* out1;
};

template <typename T, int MODE, bool OUT>
__global__ __launch_bounds__(kBlock) void cplx_init_norms_kernel(const CplxInitArgs<T> a) {
    __shared__ double red[3 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = 2 * b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const int64_t nr = 2 * valid;
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    double acc[3] = {0.0, 0.0, 0.0};
    // once per solve (twice with the heuristic's second norm): one complex element per lane and iteration
    for (int64_t t = 2 * (int64_t)threadIdx.x; t < nr; t += 2 * kBlock) {
        const T yr = a.y[base + t], yi = a.y[base + t + 1];
        const T scale = atol + chyp(yr, yi) * rtol;
        const T inv = (T)1 / scale;
        T q0r, q0i;
        if (MODE == 0) {
            q0r = a.a[base + t] * inv;
            q0i = a.a[base + t + 1] * inv;
            const T q1r = a.b[base + t] * inv, q1i = a.b[base + t + 1] * inv;
            acc[1] += cabs2(q1r, q1i);
            if (OUT) {
                a.out1[base + t] = q1r;
                a.out1[base + t + 1] = q1i;
            }
        } else {
            q0r = (a.a[base + t] - a.b[base + t]) * inv;
            q0i = (a.a[base + t + 1] - a.b[base + t + 1]) * inv;
        }
        acc[0] += cabs2(q0r, q0i);
        acc[2] += cfinite(yr, yi) ? 0.0 : 1.0;
        if (OUT) {
            a.out0[base + t] = q0r;
            a.out0[base + t + 1] = q0i;
        }
    }
    if (OUT) {
        if (a.st.n_seg > 1)
            for (int64_t t = nr + threadIdx.x; t < 2 * a.st.chunk; t += kBlock) {
                a.out0[base + t] = (T)0;
                if (MODE == 0) a.out1[base + t] = (T)0;
            }
        return;
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        a.part0[b] = acc[0];
        if (MODE == 0) a.part1[b] = acc[1];
        a.part_bad[b] = acc[2];
    }
}

}  // namespace tdeq
