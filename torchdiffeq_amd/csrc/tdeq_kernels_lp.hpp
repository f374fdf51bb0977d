// tdeq_kernels_lp.hpp — gfx950 device code of the Runge–Kutta hot path for bfloat16 / float16 STATES.
//
// The reference integrates a reduced-precision state in its own precision: every time-like scalar is cast to
// `y0.abs().dtype` (rk_common.py:61-65, misc.py:185-187) and every state-sized operation is an ATen op on bf16 / fp16
// tensors.  ATen evaluates such an op in float32 ("opmath") and rounds the RESULT to the storage type — once per op —
// and a `torch.sum` over a tableau row (rk_common.py:79,89,366) accumulates the already-rounded products in float32
// and rounds the sum once.  The kernels here do exactly that, fused: 16-byte loads of 8 storage elements per lane,
// arithmetic on floats that always hold storage-representable values (`S::rnd` after every reference op), one
// 16-byte store.  A trial step that costs the torch-op host path ~220 launches and dense [N, row] product tensors is
// the same handful of launches as for fp32 — at half the bytes per element.
//
// What is NOT mirrored (as for fp32 / fp64, docs/LAB_NOTEBOOK.md §8): structural zeros of a tableau row are skipped, and the norm
// accumulates the rounded squares in fp64 per chunk (ATen: float32 cascade) — both far below the 2^-8 / 2^-11 storage
// rounding that sets the result's precision.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tdeq_kernels.hpp"

namespace tdeq {
namespace lp {

__host__ __device__ __forceinline__ float bits_to_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}
__host__ __device__ __forceinline__ uint32_t float_to_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
#endif
}

// bfloat16: the upper half of a float32; round to nearest even on the dropped 16 bits (c10::BFloat16's conversion).
struct BF16 {
    static constexpr int code = TDEQ_BF16;
    __host__ __device__ static __forceinline__ float ld(uint32_t h) { return bits_to_float(h << 16); }
    // On the device the conversion is gfx950's v_cvt_pk_bf16_f32 (round to nearest even, what `(__bf16)f` compiles to):
    // one instruction instead of the five-operation integer sequence, which made the 16-bit combines ALU-bound.  The
    // host twin (coefficients, scalars) is the integer sequence; both are bit-identical on every finite input and inf
    // (tests/test_lowp_gpu.py compares whole tensors with ATen's CPU conversion).
    __host__ __device__ static __forceinline__ uint32_t st(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
        const __bf16 h = (__bf16)f;
        uint16_t b;
        __builtin_memcpy(&b, &h, 2);
        return b;
#else
        const uint32_t u = float_to_bits(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;              // NaN stays NaN (quiet)
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
#endif
    }
    __host__ __device__ static __forceinline__ float rnd(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
        return bits_to_float(st(f) << 16);
#else
        const uint32_t u = float_to_bits(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return f;
        return bits_to_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
#endif
    }
    // two packed elements of a 32-bit word
    __device__ static __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
        lo = bits_to_float(w << 16);
        hi = bits_to_float(w & 0xffff0000u);
    }
};

// IEEE half: the hardware conversions (v_cvt_f16_f32 / v_cvt_f32_f16) are round-to-nearest-even with subnormals.
struct F16 {
    static constexpr int code = TDEQ_F16;
    __host__ __device__ static __forceinline__ float ld(uint32_t h) {
        const uint16_t b = (uint16_t)h;
        _Float16 v;
        __builtin_memcpy(&v, &b, 2);
        return (float)v;
    }
    __host__ __device__ static __forceinline__ uint32_t st(float f) {
        const _Float16 v = (_Float16)f;
        uint16_t b;
        __builtin_memcpy(&b, &v, 2);
        return b;
    }
    __host__ __device__ static __forceinline__ float rnd(float f) { return (float)(_Float16)f; }
    __device__ static __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
        lo = ld(w & 0xffffu);
        hi = ld(w >> 16);
    }
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kVec = 8;      // storage elements per 16-byte lane access

template <typename S, int L>
__device__ __forceinline__ void unpack_elems(const u32x4& w, float (&v)[L]) {
    if constexpr (L == kVec) {
        S::unpack2(w.x, v[0], v[1]);
        S::unpack2(w.y, v[2], v[3]);
        S::unpack2(w.z, v[4], v[5]);
        S::unpack2(w.w, v[6], v[7]);
    } else {
        v[0] = S::ld(w.x & 0xffffu);
    }
}

template <typename S, int L>
__device__ __forceinline__ void load_elems(const uint16_t* __restrict__ p, int64_t i, float (&v)[L]) {
    if constexpr (L == kVec) {
        unpack_elems<S, L>(reinterpret_cast<const u32x4*>(p)[i], v);
    } else {
        v[0] = S::ld(p[i]);
    }
}

template <typename S, int L>
__device__ __forceinline__ void store_elems(uint16_t* __restrict__ p, int64_t i, const float (&v)[L]) {
    if constexpr (L == kVec) {
        u32x4 w;
        w.x = S::st(v[0]) | (S::st(v[1]) << 16);
        w.y = S::st(v[2]) | (S::st(v[3]) << 16);
        w.z = S::st(v[4]) | (S::st(v[5]) << 16);
        w.w = S::st(v[6]) | (S::st(v[7]) << 16);
        reinterpret_cast<u32x4*>(p)[i] = w;
    } else {
        p[i] = (uint16_t)S::st(v[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// Elementwise map: NOUT outputs from NIN input streams, one functor call per element.
// F::operator()(const float (&in)[NIN], float (&out)[NOUT]) — in[] / out[] hold storage-representable floats.
// Optional side payload (the stage times of a step, as for stage_combine_fill_kernel): workgroup 0 stores n_fill words.
// ------------------------------------------------------------------------------------------------
template <int NIN, int NOUT>
struct MapArgs {
    const uint16_t* in[NIN];
    uint16_t* out[NOUT];
    int64_t n;
    int n_live;            // outputs o >= n_live are computed and dropped (dense_eval_multi with fewer rows than M)
    uint16_t* fill_dst;
    uint16_t fill_v[16];
    int n_fill;
};

// LEAN (host: the grid covers the tensor once, no side fill, every output live) and PREP = false (the functor's
// prepare() is a no-op for this launch: coefficients already hold the step size) are compile-time properties of the common
// launches: the generic kernel reads its arguments in four dependent rounds of scalar loads before the first stream load —
// a sizeable part of a wave that handles one 16-byte access per stream.
template <typename S, int NIN, int NOUT, bool VEC, typename F, bool LEAN = false, bool PREP = true>
__global__ __launch_bounds__(kBlock) void map_kernel(const MapArgs<NIN, NOUT> a, const F f_arg) {
    constexpr int L = VEC ? kVec : 1;
    F f = f_arg;
    const int64_t ne = a.n / L;
    auto compute_store = [&](int64_t i, const float (&v)[NIN][L]) {
        float r[NOUT][L];
#pragma unroll
        for (int q = 0; q < L; ++q) {
            float x[NIN], y[NOUT];
#pragma unroll
            for (int j = 0; j < NIN; ++j) x[j] = v[j][q];
            f(x, y);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) r[o][q] = y[o];
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
            if (LEAN || o < a.n_live) store_elems<S, L>(a.out[o], i, r[o]);
    };
    if constexpr (LEAN) {
        // stream loads first; the functor's own (dependent, scalar) load of the device's step size overlaps them
        static_assert(!LEAN || VEC, "the lean instantiations are 16-byte launches");
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        u32x4 w[NIN];        // (lanes past the end never read theirs)
        if (i < ne) {
#pragma unroll
            for (int j = 0; j < NIN; ++j) w[j] = reinterpret_cast<const u32x4*>(a.in[j])[i];
        }
        if constexpr (PREP) f.prepare();
        if (i < ne) {
            float v[NIN][L];
#pragma unroll
            for (int j = 0; j < NIN; ++j) unpack_elems<S, L>(w[j], v[j]);
            compute_store(i, v);
        }
    } else {
        if constexpr (PREP) f.prepare();   // (captured steps: coefficients times the step size the device controller left in memory)
        if (blockIdx.x == 0 && (int)threadIdx.x < a.n_fill) a.fill_dst[threadIdx.x] = a.fill_v[threadIdx.x];
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
            float v[NIN][L];
#pragma unroll
            for (int j = 0; j < NIN; ++j) load_elems<S, L>(a.in[j], i, v[j]);
            compute_store(i, v);
        }
    }
    if (VEC) {   // scalar tail (n % 8 elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            float x[NIN], y[NOUT];
#pragma unroll
            for (int j = 0; j < NIN; ++j) x[j] = S::ld(a.in[j][t]);
            f(x, y);
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                if (LEAN || o < a.n_live) a.out[o][t] = (uint16_t)S::st(y[o]);
        }
    }
}

struct NoPrepare {
    __device__ __forceinline__ void prepare() {}
};

// A tableau row's sum, as `torch.sum(k[..., :n] * c, dim=-1)` evaluates it for a reduced-precision tensor: products
// rounded to the storage type, accumulated in float32, ONE rounding of the sum (rk_common.py:79,89,366).
template <typename S, int NT>
__device__ __forceinline__ float row_sum(const float* k, const float (&c)[NT]) {
    float acc = S::rnd(k[0] * c[0]);
#pragma unroll
    for (int j = 1; j < NT; ++j) acc += S::rnd(k[j] * c[j]);
    return S::rnd(acc);
}

// in = {y0, k_0 .. k_{NT-1}};  out_o = [y0 +] row_sum(c_o)          (rk_common.py:79, 83-85, 89; misc.py:65)
template <typename S, int NT, int NOUT>
struct CombineF {
    float c[NOUT][NT];     // fl_S(fl_S(coef) * fl_S(dt)) — or fl_S(coef) when the step size comes from device memory:
    const double* ctrl_dev;   // non-null (captured steps): ctrl_dev[1] = sign * fl_S(dt) of the device-resident controller
    __device__ __forceinline__ void prepare() {
        if (ctrl_dev) {
            const float dt = (float)ctrl_dev[1];
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int j = 0; j < NT; ++j) c[o][j] = S::rnd(c[o][j] * dt);
        }
    }
    __device__ __forceinline__ void operator()(const float (&in)[NT + 1], float (&out)[NOUT]) const {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float s = row_sum<S, NT>(in + 1, c[o]);
            out[o] = o == 0 ? S::rnd(in[0] + s) : s;      // output 0 = the stage input / y1, output 1 = the partial error row
        }
    }
};

// Dense output: the quartic of interp.py:17-21 on y_mid = y0 + row_sum(mid) (rk_common.py:363-369), every operation of
// those lines rounded on its own, then the Horner-free evaluation of interp.py:42-47 at M points (or the five planes).
//   in = {y0, y1, f0, f1, k_0 .. k_{NT-1}}
struct QuarticF {
    float e, d, c, b, a;
};

template <typename S, int NT>
__device__ __forceinline__ QuarticF quartic(const float* in, const float (&cm)[NT], float dt, float two_dt) {
    const float y0 = in[0], y1 = in[1], f0 = in[2], f1 = in[3];
    const float ymid = S::rnd(y0 + row_sum<S, NT>(in + 4, cm));
    QuarticF q;
    q.a = S::rnd(S::rnd(S::rnd(two_dt * S::rnd(f1 - f0)) - S::rnd(8.0f * S::rnd(y1 + y0))) + S::rnd(16.0f * ymid));
    q.b = S::rnd(S::rnd(S::rnd(S::rnd(dt * S::rnd(S::rnd(5.0f * f0) - S::rnd(3.0f * f1))) + S::rnd(18.0f * y0))
                        + S::rnd(14.0f * y1)) - S::rnd(32.0f * ymid));
    q.c = S::rnd(S::rnd(S::rnd(S::rnd(dt * S::rnd(f1 - S::rnd(4.0f * f0))) - S::rnd(11.0f * y0)) - S::rnd(5.0f * y1))
                 + S::rnd(16.0f * ymid));
    q.d = S::rnd(dt * f0);
    q.e = y0;
    return q;
}

template <typename S, int NT, int M>      // M evaluation points (x, x^2, x^3, x^4 pre-rounded by the host side)
struct DenseEvalF : NoPrepare {
    float cm[NT];
    float dt, two_dt;
    float xp[M][4];
    __device__ __forceinline__ void operator()(const float (&in)[NT + 4], float (&out)[M]) const {
        const QuarticF q = quartic<S, NT>(in, cm, dt, two_dt);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float total = S::rnd(q.e + S::rnd(xp[m][0] * q.d));
            total = S::rnd(total + S::rnd(xp[m][1] * q.c));
            total = S::rnd(total + S::rnd(xp[m][2] * q.b));
            out[m] = S::rnd(total + S::rnd(xp[m][3] * q.a));
        }
    }
};

template <typename S, int NT>
struct DenseFitF : NoPrepare {
    float cm[NT];
    float dt, two_dt;
    __device__ __forceinline__ void operator()(const float (&in)[NT + 4], float (&out)[5]) const {
        const QuarticF q = quartic<S, NT>(in, cm, dt, two_dt);
        out[0] = q.e; out[1] = q.d; out[2] = q.c; out[3] = q.b; out[4] = q.a;
    }
};

// 3/8-rule stages (rk_common.py:110-118): dt as FIRST operand is rounded to the storage type, `* _one_third` and
// `* dt` as second operands are taken at float32, 0.125 and 3 are exact.   in = {y0, k1 .. k_STAGE}
template <typename S, int STAGE>
struct Rk4F : NoPrepare {
    float dt_first, dt_second, third;
    __device__ __forceinline__ void operator()(const float (&in)[STAGE + 1], float (&out)[1]) const {
        float r;
        if constexpr (STAGE == 1) r = S::rnd(S::rnd(dt_first * in[1]) * third);
        else if constexpr (STAGE == 2) r = S::rnd(dt_first * S::rnd(in[2] - S::rnd(in[1] * third)));
        else if constexpr (STAGE == 3) r = S::rnd(dt_first * S::rnd(S::rnd(in[1] - in[2]) + in[3]));
        else r = S::rnd(S::rnd(S::rnd(S::rnd(in[1] + S::rnd(3.0f * S::rnd(in[2] + in[3]))) + in[4]) * dt_second) * 0.125f);
        out[0] = S::rnd(in[0] + r);
    }
};

// Generic fixed-grid stages (rk_common.py:121-157): MODE 1 = y0 + (k0 * dt) * w0; MODE 0 = y0 + (w0 k0 + w1 k1 ...) * dt
// with the sum a CHAIN of elementwise additions (each rounded), not a torch.sum.      in = {y0, k_0 .. k_{NT-1}}
template <typename S, int NT, int MODE>
struct FixedF : NoPrepare {
    float w[NT];      // float32 (second operands)
    float dt;         // rounded to the storage type
    __device__ __forceinline__ void operator()(const float (&in)[NT + 1], float (&out)[1]) const {
        if (MODE == 1) {
            out[0] = S::rnd(in[0] + S::rnd(S::rnd(in[1] * dt) * w[0]));
        } else {
            float acc = S::rnd(in[1] * w[0]);
#pragma unroll
            for (int j = 1; j < NT; ++j) acc = S::rnd(acc + S::rnd(in[1 + j] * w[j]));
            out[0] = S::rnd(in[0] + S::rnd(acc * dt));
        }
    }
};

// out = x_0 w_0 + x_1 w_1 + ... (a chain, each step rounded)        in = {x_0 .. x_{NT-1}}
template <typename S, int NT>
struct WeightedF : NoPrepare {
    float w[NT];
    __device__ __forceinline__ void operator()(const float (&in)[NT], float (&out)[1]) const {
        float acc = S::rnd(in[0] * w[0]);
#pragma unroll
        for (int j = 1; j < NT; ++j) acc = S::rnd(acc + S::rnd(in[j] * w[j]));
        out[0] = acc;
    }
};

// out = y0 + slope * (y1 - y0)   (solvers.py:175-181)          in = {y0, y1}
template <typename S>
struct LerpF : NoPrepare {
    float slope;
    __device__ __forceinline__ void operator()(const float (&in)[2], float (&out)[1]) const {
        out[0] = S::rnd(in[0] + S::rnd(slope * S::rnd(in[1] - in[0])));
    }
};

// Look-ahead first stage of the NEXT trial step (stage_combine_sel_kernel's 16-bit twin): the device controller's words
// {accept, sign * fl_S(dt')} select the pair the step starts from and give its size;  out = y + fl(f * fl(coef * dt')).
struct SelArgs {
    uint16_t* out;
    const uint16_t* y_acc;
    const uint16_t* f_acc;
    const uint16_t* y_rej;
    const uint16_t* f_rej;
    float coef;              // fl_S(coef)
    const double* ctrl_dev;
    int64_t n;
};

template <typename S, bool VEC>
__global__ __launch_bounds__(kBlock) void sel_kernel(const SelArgs a) {
    constexpr int L = VEC ? kVec : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    // the ACCEPTED pair is loaded before the controller's words are looked at (stage_combine_sel_kernel, tdeq_kernels.hpp);
    // a rejected step re-loads from the other pair
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float y[L], f[L], r[L];
    if (i0 < ne) {
        load_elems<S, L>(a.y_acc, i0, y);
        load_elems<S, L>(a.f_acc, i0, f);
    }
    const bool accept = a.ctrl_dev[0] != 0.0;
    const float c = S::rnd(a.coef * (float)a.ctrl_dev[1]);
    const uint16_t* ys = accept ? a.y_acc : a.y_rej;
    const uint16_t* fs = accept ? a.f_acc : a.f_rej;
    if (i0 < ne) {
        if (!accept) {
            load_elems<S, L>(ys, i0, y);
            load_elems<S, L>(fs, i0, f);
        }
#pragma unroll
        for (int q = 0; q < L; ++q) r[q] = S::rnd(y[q] + S::rnd(f[q] * c));
        store_elems<S, L>(a.out, i0, r);
    }
    for (int64_t i = i0 + stride; i < ne; i += stride) {      // (only beyond 65536 workgroups)
        load_elems<S, L>(ys, i, y);
        load_elems<S, L>(fs, i, f);
#pragma unroll
        for (int q = 0; q < L; ++q) r[q] = S::rnd(y[q] + S::rnd(f[q] * c));
        store_elems<S, L>(a.out, i, r);
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) a.out[t] = (uint16_t)S::st(S::rnd(S::ld(ys[t]) + S::rnd(S::ld(fs[t]) * c)));
    }
}

// ------------------------------------------------------------------------------------------------
// Norm passes: one workgroup per chunk -> fp64 partials per chunk (same workspace layout and finalize launch as the
// fp32 / fp64 kernels: tdeq_kernels.hpp norm_finalize_kernel).
// ------------------------------------------------------------------------------------------------
template <int NT>
struct ErrArgs {
    const uint16_t* y0;
    const uint16_t* y1;
    const uint16_t* k[NT];
    float c[NT];           // fl_S(fl_S(c_error_j) * fl_S(dt)) — or fl_S(c_error_j) with ctrl_dev (captured steps)
    const double* ctrl_dev;
    SegTable st;
    double* part_sumsq;    // [n_chunks]  sum of fl_S(|r|^2) — for a ONE-element segment |r| itself (see norm_term)
    double* part_bad;      // [n_chunks]
    uint16_t* scaled;      // WRITE: err / tol per element (padding zero-filled)
};

// One element's contribution to a segment's norm sum: fl_S(|q|^2) (misc.py:22: `x.abs().pow(2)` rounds the square).
// A segment of ONE element reports |q| instead: the adjoint's norms take their time component as `t.abs()`, not as an
// rms (adjoint.py:250, 273), and the host derives the square from it with the same rounding.
template <typename S>
__device__ __forceinline__ double norm_term(float q, bool one) {
    const float aq = __builtin_fabsf(q);
    return one ? (double)aq : (double)S::rnd(aq * aq);
}

// err = row_sum(c_error * dt); tol = atol + rtol * max(|y0|, |y1|) with rtol, atol rounded to the storage type and the
// product and the sum rounded (misc.py:81); r = err / tol rounded (misc.py:82); norm term = fl_S(|r|^2) (misc.py:22).
// SINGLE (host: one segment starting at chunk 0): the segment's fields are scalar loads from the kernel arguments;
// DEVDT = false (coefficients already multiplied by dt): no dependent load of the device's dt — see error_norm_partial_kernel.  A wave runs
// two 8-element iterations, so what precedes the first stream load is a sizeable part of its life.
template <typename S, int NT, bool VEC, bool WRITE, bool SINGLE = false, bool DEVDT = true>
__global__ __launch_bounds__(kBlock) void error_norm_kernel(const ErrArgs<NT> a) {
    constexpr int L = VEC ? kVec : 1;
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const int64_t base = b * a.st.chunk;
    int64_t valid;
    float rtol, atol;
    bool one;
    if constexpr (SINGLE) {
        valid = a.st.inl[0].numel - b * a.st.chunk;
        rtol = S::rnd((float)a.st.inl[0].rtol);
        atol = S::rnd((float)a.st.inl[0].atol);
        one = a.st.inl[0].numel == 1;
    } else {
        const tdeq_segment seg = find_segment(a.st, b);
        valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
        rtol = S::rnd((float)seg.rtol);
        atol = S::rnd((float)seg.atol);
        one = seg.numel == 1;
    }
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    double acc = 0.0;
    uint32_t n_bad = 0;          // at most chunk / kBlock per lane
    float cc[NT];
    if constexpr (!DEVDT) {
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.c[j];
    } else {
        const float dtd = a.ctrl_dev ? (float)a.ctrl_dev[1] : 1.0f;
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.ctrl_dev ? S::rnd(a.c[j] * dtd) : a.c[j];
    }
    // `single` = the segment is ONE element: only ever true in the scalar tail (such a segment has no 8-element group)
    auto elem = [&](const float* kk, float y0, float y1, bool single) -> float {
        const float e = row_sum<S, NT>(kk, cc);
        const float tol = S::rnd(S::rnd(__builtin_fmaxf(__builtin_fabsf(y0), __builtin_fabsf(y1)) * rtol) + atol);
        const float r = S::rnd(e / tol);
        acc += norm_term<S>(r, single);
        n_bad += (__builtin_isfinite(y0) && __builtin_isfinite(y1)) ? 0u : 1u;
        return r;
    };
    const int64_t nv = valid / L;
    for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
        float kk[NT][L], y0[L], y1[L], r[L];
#pragma unroll
        for (int j = 0; j < NT; ++j) load_elems<S, L>(a.k[j] + base, i, kk[j]);
        load_elems<S, L>(a.y0 + base, i, y0);
        load_elems<S, L>(a.y1 + base, i, y1);
#pragma unroll
        for (int q = 0; q < L; ++q) {
            float kq[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kq[j] = kk[j][q];
            r[q] = elem(kq, y0[q], y1[q], VEC ? false : one);
        }
        if (WRITE) store_elems<S, L>(a.scaled + base, i, r);
    }
    if (VEC) {
        const int64_t t = nv * L + threadIdx.x;
        if (t < valid) {
            float kq[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kq[j] = S::ld(a.k[j][base + t]);
            const float r = elem(kq, S::ld(a.y0[base + t]), S::ld(a.y1[base + t]), one);
            if (WRITE) a.scaled[base + t] = (uint16_t)S::st(r);
        }
    }
    if (WRITE && a.st.n_seg > 1)   // zero the padding of a segmented layout
        for (int64_t t = valid + threadIdx.x; t < a.st.chunk; t += kBlock) a.scaled[base + t] = 0;
    double sums[2] = {acc, (double)n_bad};
    block_sum<2>(sums, red);
    if (threadIdx.x == 0) {
        a.part_sumsq[b] = sums[0];
        a.part_bad[b] = sums[1];
    }
}

// Initial-step quotients (misc.py:50-66): scale = atol + |y| * rtol with rtol the SECOND operand (taken at float32),
// atol rounded to the storage type.  MODE 0: (a / scale, b / scale); MODE 1: (a - b) / scale.  WRITE: the quotients
// themselves (user norm callables) instead of their sums.
struct InitArgs {
    const uint16_t* a;
    const uint16_t* b;
    const uint16_t* y;
    SegTable st;
    double* part0;
    double* part1;
    double* part_bad;
    uint16_t* out0;
    uint16_t* out1;
};

template <typename S, int MODE, bool WRITE>
__global__ __launch_bounds__(kBlock) void init_norms_kernel(const InitArgs a) {
    __shared__ double red[3 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const float rtol = (float)seg.rtol, atol = S::rnd((float)seg.atol);
    const bool one = seg.numel == 1;
    double acc[3] = {0.0, 0.0, 0.0};
    // (three streams, once per solve: a plain scalar loop per chunk)
    for (int64_t t = threadIdx.x; t < valid; t += kBlock) {
        const float av = S::ld(a.a[base + t]), bv = S::ld(a.b[base + t]), yv = S::ld(a.y[base + t]);
        const float scale = S::rnd(S::rnd(__builtin_fabsf(yv) * rtol) + atol);
        const float q0 = MODE == 0 ? S::rnd(av / scale) : S::rnd(S::rnd(av - bv) / scale);
        if (WRITE) a.out0[base + t] = (uint16_t)S::st(q0);
        acc[0] += norm_term<S>(q0, one);
        if (MODE == 0) {
            const float q1 = S::rnd(bv / scale);
            if (WRITE) a.out1[base + t] = (uint16_t)S::st(q1);
            acc[1] += norm_term<S>(q1, one);
        }
        acc[2] += __builtin_isfinite(yv) ? 0.0 : 1.0;
    }
    if (WRITE && a.st.n_seg > 1)
        for (int64_t t = valid + threadIdx.x; t < a.st.chunk; t += kBlock) {
            a.out0[base + t] = 0;
            if (MODE == 0) a.out1[base + t] = 0;
        }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0 && !WRITE) {
        a.part0[b] = acc[0];
        if (MODE == 0) a.part1[b] = acc[1];
        a.part_bad[b] = acc[2];
    }
}

}  // namespace lp
}  // namespace tdeq
