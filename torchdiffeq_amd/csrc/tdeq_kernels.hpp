// tdeq_kernels.hpp — gfx950 (MI355X, CDNA4) device code of the explicit Runge–Kutta hot path.
//
// Everything here is bandwidth-bound streaming work (AXPY-like combines and reductions): no MFMA, no
// LDS tiling — the levers are 16 B/lane coalesced loads, many independent loads in flight per lane,
// ≫256 workgroups, wave64 shuffle reductions, and zero redundant passes over the state.
//
// Layout: the RK stages k_j are SEPARATE contiguous tensors (structure of arrays) — not the
// reference's stage-minor `k[*shape, S+1]` (rk_common.py:69), whose stride-(S+1) scatter/gather is
// uncoalesced.  Their addresses and the already-rounded coefficients c_j travel in the kernel-argument
// block, i.e. they are read with scalar loads into SGPRs once per wave (cheaper than staging the
// tableau row through LDS: no LDS round trip, no barrier, no VGPR cost).
//
// Arithmetic mirrors the reference's eager op sequence in the state dtype T with every product and
// sum rounded separately (compile with -ffp-contract=off): see the per-kernel comments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tdeq_hip.h"

namespace tdeq {

constexpr int kBlock = 256;   // 4 wave64 per workgroup
constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = f32x4; static constexpr int L = 4; };
template <> struct VecOf<double> { using type = f64x2; static constexpr int L = 2; };

template <typename V> __device__ __forceinline__ V vabs(V v);
template <> __device__ __forceinline__ f32x4 vabs(f32x4 v) {
    return f32x4{__builtin_fabsf(v.x), __builtin_fabsf(v.y), __builtin_fabsf(v.z), __builtin_fabsf(v.w)};
}
template <> __device__ __forceinline__ f64x2 vabs(f64x2 v) {
    return f64x2{__builtin_fabs(v.x), __builtin_fabs(v.y)};
}
__device__ __forceinline__ float sabs(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double sabs(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float smax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double smax(double a, double b) { return __builtin_fmax(a, b); }

// ------------------------------------------------------------------------------------------------
// Stage accumulate:  out = y0 + ((c0*k0 + c1*k1) + ... + c_{NT-1}*k_{NT-1})      (rk_common.py:79)
// ------------------------------------------------------------------------------------------------
template <typename T, int NT>
struct CombineArgs {
    T* out;
    const T* y0;
    const T* k[NT];
    T c[NT];
    int64_t n;
};

template <typename T, int NT, typename E>   // E = T (scalar path) or the 16-byte vector of T
__device__ __forceinline__ E combine_one(const CombineArgs<T, NT>& a, const E& y, const E (&kk)[NT]) {
    E acc = kk[0] * a.c[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) acc = acc + kk[j] * a.c[j];
    return y + acc;
}

// Cache-policy variants for tuning: bit 0 = non-temporal loads of the k_j / y0 streams, bit 1 =
// non-temporal store of the result.
template <int POLICY, typename E>
__device__ __forceinline__ E ld_stream(const E* p) {
    if constexpr (POLICY & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int POLICY, typename E>
__device__ __forceinline__ void st_stream(E* p, const E& v) {
    if constexpr (POLICY & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Optional side payload of the first stage combine of a step: up to 16 scalars (the stage times handed to
// func) stored by workgroup 0 — saves the separate fill launch at the latency-critical start of a step.
template <typename T>
struct SideFill {
    T* dst;
    T v[16];
    int n;
};

template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(kBlock) void stage_combine_fill_kernel(const CombineArgs<T, NT> a, const SideFill<T> f) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    if (blockIdx.x == 0 && (int)threadIdx.x < f.n) f.dst[threadIdx.x] = f.v[threadIdx.x];
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const E* __restrict__ y0 = reinterpret_cast<const E*>(a.y0);
    E* __restrict__ out = reinterpret_cast<E*>(a.out);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        E kk[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(a.k[j])[i];
        out[i] = combine_one<T, NT, E>(a, y0[i], kk);
    }
    if (VEC) {   // scalar tail (n % L elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.k[j][t];
            a.out[t] = combine_one<T, NT, T>(a, a.y0[t], kk);
        }
    }
}

// Last combine of a step fused with the partial embedded error over the SAME stage set:
//   out = y0 + sum_j c_j k_j          err_out = (e_0 k_0 + e_1 k_1) + ...        (rk_common.py:79/85 and :89)
// The k_j are in registers anyway; the extra N-word store replaces |set| N-word re-reads in the norm kernel.
template <typename T, int NT>
struct CombineErrArgs {
    CombineArgs<T, NT> c;
    T* err_out;
    T e[NT];
};

// ONEPASS (host: the grid covers the tensor once — every launch below kMaxGrid workgroups): no grid-stride loop, so no
// gridDim read and no loop-carried 64-bit index — the first stream load issues ~20 instructions into the wave.
template <typename T, int NT, bool VEC, int POLICY = 0, bool ONEPASS = false>
__global__ __launch_bounds__(kBlock) void stage_combine_err_kernel(const CombineErrArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.c.n / L;
    const int64_t stride = ONEPASS ? ne : (int64_t)gridDim.x * kBlock;
    const E* __restrict__ y0 = reinterpret_cast<const E*>(a.c.y0);
    E* __restrict__ out = reinterpret_cast<E*>(a.c.out);
    E* __restrict__ eo = reinterpret_cast<E*>(a.err_out);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        E kk[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = ld_stream<POLICY>(reinterpret_cast<const E*>(a.c.k[j]) + i);
        st_stream<POLICY>(out + i, combine_one<T, NT, E>(a.c, ld_stream<POLICY>(y0 + i), kk));
        E err = kk[0] * a.e[0];
#pragma unroll
        for (int j = 1; j < NT; ++j) err = err + kk[j] * a.e[j];
        st_stream<POLICY>(eo + i, err);
    }
    if (VEC) {   // scalar tail (n % L elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.c.n) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.c.k[j][t];
            a.c.out[t] = combine_one<T, NT, T>(a.c, a.c.y0[t], kk);
            T err = kk[0] * a.e[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) err = err + kk[j] * a.e[j];
            a.err_out[t] = err;
        }
    }
}

// U independent 16-byte elements per lane and iteration => (NT+1)*U loads in flight per lane.
template <typename T, int NT, int U, bool VEC, int POLICY = 0, bool ONEPASS = false>
__global__ __launch_bounds__(kBlock) void stage_combine_kernel(const CombineArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    if constexpr (ONEPASS && U == 1) {      // see stage_combine_err_kernel
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < ne) {
            E kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = ld_stream<POLICY>(reinterpret_cast<const E*>(a.k[j]) + i);
            st_stream<POLICY>(reinterpret_cast<E*>(a.out) + i,
                              combine_one<T, NT, E>(a, ld_stream<POLICY>(reinterpret_cast<const E*>(a.y0) + i), kk));
        }
        if (VEC) {
            const int64_t t = ne * L + threadIdx.x;
            if (blockIdx.x == 0 && t < a.n) {
                T kk[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) kk[j] = a.k[j][t];
                a.out[t] = combine_one<T, NT, T>(a, a.y0[t], kk);
            }
        }
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const E* __restrict__ y0 = reinterpret_cast<const E*>(a.y0);
    E* __restrict__ out = reinterpret_cast<E*>(a.out);
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + (U - 1) * stride < ne; i += U * stride) {
        E y[U];
        E kk[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            y[u] = ld_stream<POLICY>(y0 + i + u * stride);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                kk[u][j] = ld_stream<POLICY>(reinterpret_cast<const E*>(a.k[j]) + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st_stream<POLICY>(out + i + u * stride, combine_one<T, NT, E>(a, y[u], kk[u]));
    }
    for (; i < ne; i += stride) {
        E kk[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = ld_stream<POLICY>(reinterpret_cast<const E*>(a.k[j]) + i);
        st_stream<POLICY>(out + i, combine_one<T, NT, E>(a, ld_stream<POLICY>(y0 + i), kk));
    }
    if (VEC) {   // scalar tail (n % L elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.k[j][t];
            a.out[t] = combine_one<T, NT, T>(a, a.y0[t], kk);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Carried partial sums (tdeq_stage_combine_multi): one pass over NT stage streams, up to kMaxMultiOut outputs.
//   s_o = [acc_in +] sum_{j in mask_o, ascending} c_o[j] * k_j ;  out_o = add_y0_o ? y0 + s_o : s_o
// Output 0 is the row's own stage input; the others are left-to-right prefixes of LATER rows' sums (continued by
// those rows through acc_in) or later stage inputs that need no newer stage.  Same rounding sequence as
// stage_combine_kernel: the first product starts the sum (no 0 + x, which would turn -0 into +0), structural zeros
// are skipped by the wave-uniform mask — scalar branches on kernel arguments —, -ffp-contract=off.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxMultiOut = TDEQ_MAX_MULTI_OUT;

template <typename T, int NT>
struct MultiArgs {
    const T* y0;
    const T* acc_in;                  // nullable: prefix of output 0's sum
    const T* k[NT];
    T* out[kMaxMultiOut];
    T c[kMaxMultiOut][NT];
    uint32_t mask[kMaxMultiOut];
    uint32_t add_y0;                  // bit o: out_o = y0 + s_o
    int n_out;
    int64_t n;
    const double* dt_dev;             // non-null (hipGraph mode): c[][] holds fl_T(coef) and is multiplied by T(dt_dev[1]) here
};

// NOUTC / ACCC: the launch's output count (1 or 2; 0 = read a.n_out) and whether it continues a carried prefix (1 / 0;
// -1 = test a.acc_in) as COMPILE-TIME constants.  A wave of these kernels handles one 16-byte element per lane and lives
// ~1 us; the scalar branches of the generic form (four output slots, the acc_in tests) are a measurable share of that.
// (Also tried: the masks' bit tests compiled away for launches without structural zeros — no further gain, not kept.)
// DEVDT: the step size comes from device memory (captured steps, a.dt_dev) — a compile-time property as well: the host
// launches never pay for the dependent scalar load and the per-coefficient selects in their prologue.
template <typename T, int NT, typename E, int POLICY = 0, int NOUTC = 0, int ACCC = -1, bool DEVDT = true>
__device__ __forceinline__ void multi_elem(const MultiArgs<T, NT>& a, int64_t i) {
    const E* __restrict__ y0 = reinterpret_cast<const E*>(a.y0);
    // (r06) captured steps: ctrl_dev[1] = sign * T(dt) of the device-resident controller, fetched as a VECTOR load issued
    // ahead of the stream loads — one in-order queue, one wait (see combine_devdt_kernel)
    double dtd = 1.0;
    if (DEVDT && a.dt_dev) dtd = __hip_atomic_load(a.dt_dev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    E kk[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) kk[j] = ld_stream<POLICY>(reinterpret_cast<const E*>(a.k[j]) + i);
    const E y = ld_stream<POLICY>(y0 + i);
    E acc0 = y;                       // placeholder when there is no acc_in (never read then)
    const bool has_acc = ACCC < 0 ? (a.acc_in != nullptr) : (ACCC == 1);
    if (has_acc) acc0 = ld_stream<POLICY>(reinterpret_cast<const E*>(a.acc_in) + i);
    const T dtT = (T)dtd;
    constexpr int kOuts = NOUTC > 0 ? NOUTC : kMaxMultiOut;
#pragma unroll
    for (int o = 0; o < kOuts; ++o) {
        if (NOUTC > 0 || o < a.n_out) {
            const uint32_t m = a.mask[o];
            bool started = (o == 0) && has_acc;
            E s = acc0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if ((m >> j) & 1u) {
                    const T cj = (DEVDT && a.dt_dev) ? a.c[o][j] * dtT : a.c[o][j];     // fl_T(fl_T(coef) * T(dt)) either way
                    const E p = kk[j] * cj;
                    s = started ? s + p : p;
                    started = true;
                }
            }
            st_stream<POLICY>(reinterpret_cast<E*>(a.out[o]) + i, ((a.add_y0 >> o) & 1u) ? y + s : s);
        }
    }
}

// POLICY (VEC only): cache policy of the streams, see ld_stream / st_stream — chosen per launch by the host side
// (tdeq_abi.hip stream_policy(): non-temporal for launches whose streams exceed the 256 MiB Infinity Cache).
// ONEPASS: the grid covers the tensor exactly once (always, up to 2^24 16-byte elements): no grid-stride loop, hence no
// loop-carried pointers / masks in SGPRs (the generic form spilled them to VGPR lanes) and the loads are issued ~100
// instructions earlier in a wave's life.
template <typename T, int NT, bool VEC, int POLICY = 0, int NOUTC = 0, int ACCC = -1, bool DEVDT = true, bool ONEPASS = false>
__global__ __launch_bounds__(kBlock) void stage_combine_multi_kernel(const MultiArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if constexpr (ONEPASS) {
        const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < ne) multi_elem<T, NT, E, POLICY, NOUTC, ACCC, DEVDT>(a, i);
    } else {
        const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i0 < ne) multi_elem<T, NT, E, POLICY, NOUTC, ACCC, DEVDT>(a, i0);
        for (int64_t i = i0 + stride; i < ne; i += stride)      // (only beyond 65536 workgroups)
            multi_elem<T, NT, E, POLICY, NOUTC, ACCC, DEVDT>(a, i);
    }
    if (VEC) {   // scalar tail (n % L elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) multi_elem<T, NT, T>(a, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Reductions.  One workgroup per chunk -> one fp64 partial per chunk (deterministic order), then a
// finalize launch adds the partials of each segment in a fixed tree order.  wave64 shuffles first,
// LDS only for the 4 per-wave values.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sum over the workgroup; result valid in thread 0.  `red` = kBlock/kWave doubles of LDS per value.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        v[q] = wave_sum(v[q]);
        if (lane == 0) red[q * (kBlock / kWave) + wave] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double s = red[q * (kBlock / kWave)];
#pragma unroll
            for (int w = 1; w < kBlock / kWave; ++w) s += red[q * (kBlock / kWave) + w];
            v[q] = s;
        }
    }
}

struct SegTable {
    tdeq_segment inl[TDEQ_INLINE_SEGMENTS];
    const tdeq_segment* dev;   // used when n_seg > TDEQ_INLINE_SEGMENTS
    int n_seg;
    int64_t chunk;             // elements per chunk
    int64_t n_chunks;
};

__device__ __forceinline__ tdeq_segment get_segment(const SegTable& st, int s) {
    if (st.n_seg <= TDEQ_INLINE_SEGMENTS) return st.inl[s];
    return st.dev[s];
}

// Segment of chunk b (uniform per workgroup): last s with chunk_start[s] <= b.
__device__ __forceinline__ tdeq_segment find_segment(const SegTable& st, int64_t b) {
    if (st.n_seg == 1) return st.inl[0];
    if (st.n_seg <= TDEQ_INLINE_SEGMENTS) {
        int s = 0;
        for (int q = 1; q < st.n_seg; ++q) s = (st.inl[q].chunk_start <= b) ? q : s;
        return st.inl[s];
    }
    int lo = 0, hi = st.n_seg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (st.dev[mid].chunk_start <= b) lo = mid; else hi = mid - 1;
    }
    return st.dev[lo];
}

template <typename T, int NT>
struct ErrArgs {
    const T* y0;
    const T* y1;
    const T* k[NT];
    T c[NT];
    SegTable st;
    double* part_sumsq;   // [n_chunks]
    double* part_bad;     // [n_chunks]
    T* scaled;            // WRITE variant only: err/tol per element (padding zero-filled)
};

// err = (c0*k0 + c1*k1) + ... ; tol = atol + rtol*max(|y0|,|y1|) ; r = err/tol ; acc += r*r
// (rk_common.py:89, misc.py:80-82,22-23).  r is formed in T exactly as the reference does; the
// square and the accumulation are fp64 (the reference accumulates in T with an unspecified tree
// order — see DESIGN.md "norm accumulation").
template <typename T, int NT, typename E>
__device__ __forceinline__ E err_elem(const ErrArgs<T, NT>& a, T rtol, T atol, const E& y0,
                                      const E& y1, const E (&kk)[NT], double& acc, double& bad) {
    E e = kk[0] * a.c[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) e = e + kk[j] * a.c[j];
    if constexpr (sizeof(E) == sizeof(T)) {
        const T tol = atol + rtol * smax(sabs(y0), sabs(y1));
        const T r = e / tol;
        acc += (double)r * (double)r;
        bad += (__builtin_isfinite(y0) && __builtin_isfinite(y1)) ? 0.0 : 1.0;
        return r;
    } else {
        E rv;
#pragma unroll
        for (int q = 0; q < VecOf<T>::L; ++q) {
            const T tol = atol + rtol * smax(sabs(y0[q]), sabs(y1[q]));
            const T r = e[q] / tol;
            acc += (double)r * (double)r;
            bad += (__builtin_isfinite(y0[q]) && __builtin_isfinite(y1[q])) ? 0.0 : 1.0;
            rv[q] = r;
        }
        return rv;
    }
}

// WRITE = also store err/tol (for user-supplied norm callables, misc.py:80-82 with a custom norm).
template <typename T, int NT, bool VEC, bool WRITE>
__global__ __launch_bounds__(kBlock) void error_norm_kernel(const ErrArgs<T, NT> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    double acc[2] = {0.0, 0.0};
    if (VEC) {
        const int64_t nv = valid / L;
        const V* y0 = reinterpret_cast<const V*>(a.y0 + base);
        const V* y1 = reinterpret_cast<const V*>(a.y1 + base);
#pragma unroll 2
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            V kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const V*>(a.k[j] + base)[i];
            const V r = err_elem<T, NT, V>(a, rtol, atol, y0[i], y1[i], kk, acc[0], acc[1]);
            if (WRITE) reinterpret_cast<V*>(a.scaled + base)[i] = r;
        }
        const int64_t t = nv * L + threadIdx.x;
        if (t < valid) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.k[j][base + t];
            const T r = err_elem<T, NT, T>(a, rtol, atol, a.y0[base + t], a.y1[base + t], kk, acc[0], acc[1]);
            if (WRITE) a.scaled[base + t] = r;
        }
    } else {
        for (int64_t t = threadIdx.x; t < valid; t += kBlock) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.k[j][base + t];
            const T r = err_elem<T, NT, T>(a, rtol, atol, a.y0[base + t], a.y1[base + t], kk, acc[0], acc[1]);
            if (WRITE) a.scaled[base + t] = r;
        }
    }
    if (WRITE && a.st.n_seg > 1)   // zero the padding of a segmented layout
        for (int64_t t = valid + threadIdx.x; t < a.st.chunk; t += kBlock) a.scaled[base + t] = (T)0;
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        a.part_sumsq[b] = acc[0];
        a.part_bad[b] = acc[1];
    }
}

// Per-element tolerances (misc.py:80-82 with `rtol` / `atol` TENSORS that broadcast against the state — plain
// broadcasting in the reference; tuple tolerances with vector entries become such flat vectors, misc.py:115-123).  The
// tolerances are tensors of the time dtype W = fp64 (rk_common.py:186-187), so type promotion decides each operation:
//   rtol dimensioned:  rtol[i] * max(|y0|,|y1|)  is an fp64 product;   rtol 0-dim:  the product stays in T (rtol cast to T)
//   the sum with atol (dimensioned or 0-dim fp64 next to an fp64 tensor), the quotient err / tol and the norm are fp64
//   whenever either tolerance is dimensioned — which is the only case this kernel is launched for.
// Two extra fp64 streams at most; the raw error is never materialised (r04: error_scaled + ~10 ATen passes in fp64).
template <typename T, int NT>
struct ErrVecArgs {
    const T* y0;
    const T* y1;
    const T* partial;       // PARTIAL: the error row's leading run, summed by stage_combine_err_kernel (else unused)
    const T* k[NT > 0 ? NT : 1];
    T c[NT > 0 ? NT : 1];
    const double* rtol_v;   // per element of the flat (padded) state, or null: rtol_s
    const double* atol_v;
    double rtol_s, atol_s;
    SegTable st;
    double* part_sumsq;
    double* part_bad;
    const double* dt_dev;   // DEVDT (hipGraph mode, r06): c[] holds fl_T(coef) and is multiplied by T(*dt_dev) here
};

// PARTIAL: err = (partial + c_0 k_0) + ... over the NT >= 0 remaining stages, as error_norm_partial_kernel continues the
// sum stage_combine_err_kernel started — per-element tolerances keep the step's launch sequence (carried partial sums, the
// end-of-step error split) and pay exactly their own two fp64 streams.
template <typename T, int NT, bool PARTIAL = false, bool DEVDT = false>
__global__ __launch_bounds__(kBlock) void error_norm_vec_kernel(const ErrVecArgs<T, NT> a) {
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const T rtolT = (T)a.rtol_s;
    T cc[NT > 0 ? NT : 1];
    if constexpr (DEVDT) {
        const T dtT = (T)*a.dt_dev;         // the same product the host forms: fl_T(fl_T(coef) * T(dt))
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.c[j] * dtT;
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.c[j];
    }
    double acc[2] = {0.0, 0.0};
    // consecutive lanes read consecutive elements: 4- / 8-byte state loads and 8-byte tolerance loads, all coalesced;
    // two elements per lane and iteration keep 2 (NT + 4) loads in flight
#pragma unroll 2
    for (int64_t t = threadIdx.x; t < valid; t += kBlock) {
        const int64_t i = base + t;
        T e = PARTIAL ? a.partial[i] : a.k[0][i] * cc[0];
#pragma unroll
        for (int j = PARTIAL ? 0 : 1; j < NT; ++j) e = e + a.k[j][i] * cc[j];
        const T y0 = a.y0[i], y1 = a.y1[i];
        const T m = smax(sabs(y0), sabs(y1));
        const double prod = a.rtol_v ? a.rtol_v[i] * (double)m : (double)(rtolT * m);
        const double tol = (a.atol_v ? a.atol_v[i] : a.atol_s) + prod;
        const double r = (double)e / tol;
        acc[0] += r * r;
        acc[1] += (__builtin_isfinite(y0) && __builtin_isfinite(y1)) ? 0.0 : 1.0;
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        a.part_sumsq[b] = acc[0];
        a.part_bad[b] = acc[1];
    }
}

// Error norm continuing a partial error sum produced by stage_combine_err_kernel:
//   err = (partial + c_0 k_0) + ... over the NT >= 0 remaining stages; everything else as error_norm_kernel.
template <typename T, int NT>
struct ErrPartialArgs {
    const T* partial;
    const T* y0;
    const T* y1;
    const T* k[NT > 0 ? NT : 1];
    T c[NT > 0 ? NT : 1];
    SegTable st;
    double* part_sumsq;
    double* part_bad;
    const double* dt_dev;   // non-null (hipGraph mode): c[] holds fl_T(coef) and is multiplied by T(*dt_dev) here
    T* copy_out;            // COPY (r06, captured steps): the LAST remaining stage k[NT-1] — the step's f1 — is also written here,
                            // which saves the captured step its N-word copy node (the stream is in registers anyway)
};

template <typename T>
__device__ __forceinline__ void tol_accumulate(T e, T y0, T y1, T rtol, T atol, double& acc, double& bad) {
    const T tol = atol + rtol * smax(sabs(y0), sabs(y1));
    const T r = e / tol;
    acc += (double)r * (double)r;
    bad += (__builtin_isfinite(y0) && __builtin_isfinite(y1)) ? 0.0 : 1.0;
}

// SINGLE (host: st.n_seg == 1) reads the one segment's fields straight from the kernel arguments — scalar loads issued with
// the rest of the kernarg.  The general path picks the segment by address (inline table or device table), which the compiler
// turns into a VECTOR load from a generic pointer: one full memory latency in front of every wave's first stream load.
// DEVDT = false (host-driven steps: dt folded into c[] by the host) drops the load of *dt_dev and its dependent multiplies.
template <typename T, int NT, bool VEC, int POLICY = 0, bool SINGLE = false, bool DEVDT = true, bool COPY = false>
__global__ __launch_bounds__(kBlock) void error_norm_partial_kernel(const ErrPartialArgs<T, NT> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const int64_t base = b * a.st.chunk;
    int64_t valid;
    T rtol, atol;
    if constexpr (SINGLE) {
        valid = a.st.inl[0].numel - b * a.st.chunk;
        rtol = (T)a.st.inl[0].rtol;
        atol = (T)a.st.inl[0].atol;
    } else {
        const tdeq_segment seg = find_segment(a.st, b);
        valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
        rtol = (T)seg.rtol;
        atol = (T)seg.atol;
    }
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    T cc[NT > 0 ? NT : 1];
    if constexpr (DEVDT) {
        const T dtT = a.dt_dev ? (T)*a.dt_dev : (T)1;
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.dt_dev ? a.c[j] * dtT : a.c[j];
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) cc[j] = a.c[j];
    }
    double acc[2] = {0.0, 0.0};
    int64_t t0 = 0;
    if (VEC) {
        const int64_t nv = valid / L;
        const V* y0 = reinterpret_cast<const V*>(a.y0 + base);
        const V* y1 = reinterpret_cast<const V*>(a.y1 + base);
        const V* pe = reinterpret_cast<const V*>(a.partial + base);
#pragma unroll 2
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            V e = ld_stream<POLICY>(pe + i);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const V kv = ld_stream<POLICY>(reinterpret_cast<const V*>(a.k[j] + base) + i);
                if (COPY && j == NT - 1) reinterpret_cast<V*>(a.copy_out + base)[i] = kv;
                e = e + kv * cc[j];
            }
            const V v0 = ld_stream<POLICY>(y0 + i), v1 = ld_stream<POLICY>(y1 + i);
#pragma unroll
            for (int q = 0; q < L; ++q) tol_accumulate<T>(e[q], v0[q], v1[q], rtol, atol, acc[0], acc[1]);
        }
        t0 = nv * L;
    }
    for (int64_t t = t0 + threadIdx.x; t < valid; t += kBlock) {
        T e = a.partial[base + t];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const T kv = a.k[j][base + t];
            if (COPY && j == NT - 1) a.copy_out[base + t] = kv;
            e = e + kv * cc[j];
        }
        tol_accumulate<T>(e, a.y0[base + t], a.y1[base + t], rtol, atol, acc[0], acc[1]);
    }
    if constexpr (COPY && NT > 0) {
        // the alignment padding behind a segment of a segmented layout travels too (what the copy node moved): a chunk of
        // such a layout lies inside the buffer as a whole
        if (a.st.n_seg > 1)
            for (int64_t t = valid + threadIdx.x; t < a.st.chunk; t += kBlock) a.copy_out[base + t] = a.k[NT - 1][base + t];
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        a.part_sumsq[b] = acc[0];
        a.part_bad[b] = acc[1];
    }
}

// Initial-step norms (misc.py:53-56,68): scale = atol + |y|*rtol.
//   MODE 0: acc0 += (a/scale)^2 ; acc1 += (b/scale)^2        MODE 1: acc0 += ((a-b)/scale)^2
template <typename T>
struct InitArgs {
    const T* a;
    const T* b;
    const T* y;
    SegTable st;
    double* part0;   // [n_chunks]
    double* part1;   // [n_chunks] (MODE 0 only)
    double* part_bad;
};

template <typename T, int MODE>
__device__ __forceinline__ void init_elem(T rtol, T atol, T av, T bv, T yv, double (&acc)[3]) {
    const T scale = atol + sabs(yv) * rtol;
    if (MODE == 0) {
        const T r0 = av / scale, r1 = bv / scale;
        acc[0] += (double)r0 * (double)r0;
        acc[1] += (double)r1 * (double)r1;
    } else {
        const T r0 = (av - bv) / scale;
        acc[0] += (double)r0 * (double)r0;
    }
    acc[2] += __builtin_isfinite(yv) ? 0.0 : 1.0;
}

template <typename T, int MODE, bool VEC>
__global__ __launch_bounds__(kBlock) void init_norms_kernel(const InitArgs<T> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[3 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    double acc[3] = {0.0, 0.0, 0.0};
    int64_t t0 = 0;
    if (VEC) {
        const int64_t nv = valid / L;
#pragma unroll 2
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            const V av = reinterpret_cast<const V*>(a.a + base)[i];
            const V bv = reinterpret_cast<const V*>(a.b + base)[i];
            const V yv = reinterpret_cast<const V*>(a.y + base)[i];
#pragma unroll
            for (int q = 0; q < L; ++q) init_elem<T, MODE>(rtol, atol, av[q], bv[q], yv[q], acc);
        }
        t0 = nv * L;
        const int64_t t = t0 + threadIdx.x;
        if (t < valid) init_elem<T, MODE>(rtol, atol, a.a[base + t], a.b[base + t], a.y[base + t], acc);
    } else {
        for (int64_t t = threadIdx.x; t < valid; t += kBlock)
            init_elem<T, MODE>(rtol, atol, a.a[base + t], a.b[base + t], a.y[base + t], acc);
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        a.part0[b] = acc[0];
        if (MODE == 0) a.part1[b] = acc[1];
        a.part_bad[b] = acc[2];
    }
}

// Initial-step norms with PER-ELEMENT tolerances (r06; misc.py:50-56,68 on W = fp64 tolerance tensors that broadcast against
// the state).  Type promotion as ATen applies it, operation by operation (the same rules as error_norm_vec_kernel):
//   scale = atol + |y| * rtol :  |y| * rtol[i] is an fp64 product for a dimensioned rtol, a T product (rtol cast to T) for
//                                a 0-dim one; the sum with atol is fp64 whenever either tolerance is dimensioned
//   MODE 0: (a / scale)^2, (b / scale)^2     MODE 1: ((a - b) / scale)^2 with a - b formed in T; quotients and sums in fp64
template <typename T>
struct InitVecArgs {
    const T* a;
    const T* b;
    const T* y;
    const double* rtol_v;   // per element of the flat (padded) state, or null: rtol_s
    const double* atol_v;
    double rtol_s, atol_s;
    SegTable st;
    double* part0;
    double* part1;
    double* part_bad;
};

template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void init_norms_vec_kernel(const InitVecArgs<T> a) {
    __shared__ double red[3 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const T rtolT = (T)a.rtol_s;
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll 2
    for (int64_t t = threadIdx.x; t < valid; t += kBlock) {
        const int64_t i = base + t;
        const T yv = a.y[i];
        const T m = sabs(yv);
        const double prod = a.rtol_v ? (double)m * a.rtol_v[i] : (double)(m * rtolT);
        const double scale = (a.atol_v ? a.atol_v[i] : a.atol_s) + prod;
        if (MODE == 0) {
            const double r0 = (double)a.a[i] / scale, r1 = (double)a.b[i] / scale;
            acc[0] += r0 * r0;
            acc[1] += r1 * r1;
        } else {
            const double r0 = (double)(a.a[i] - a.b[i]) / scale;
            acc[0] += r0 * r0;
        }
        acc[2] += __builtin_isfinite(yv) ? 0.0 : 1.0;
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        a.part0[b] = acc[0];
        if (MODE == 0) a.part1[b] = acc[1];
        a.part_bad[b] = acc[2];
    }
}

// The same quotients materialised for a USER norm callable (misc.py:53-56,68 with norm != rms): out0 = a/scale,
// out1 = b/scale (MODE 0) or out0 = (a - b)/scale (MODE 1); padding of a segmented layout zero-filled.  Once per
// solve, so a plain scalar loop per chunk.
template <typename T>
struct InitScaledArgs {
    const T* a;
    const T* b;
    const T* y;
    SegTable st;
    T* out0;
    T* out1;
};

template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void init_scaled_kernel(const InitScaledArgs<T> a) {
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    for (int64_t t = threadIdx.x; t < valid; t += kBlock) {
        const T scale = atol + sabs(a.y[base + t]) * rtol;
        if (MODE == 0) {
            a.out0[base + t] = a.a[base + t] / scale;
            a.out1[base + t] = a.b[base + t] / scale;
        } else {
            a.out0[base + t] = (a.a[base + t] - a.b[base + t]) / scale;
        }
    }
    if (a.st.n_seg > 1)
        for (int64_t t = valid + threadIdx.x; t < a.st.chunk; t += kBlock) {
            a.out0[base + t] = (T)0;
            if (MODE == 0) a.out1[base + t] = (T)0;
        }
}

// Finalize: workgroup (s, q) adds the partials of segment s from array q in a fixed order.
//   q < n_sum  -> out_sumsq[q*n_seg + s]      q == n_sum -> out_bad[s]
constexpr int kFinBatch = 8;

struct FinalizeArgs {
    const double* part[3];   // part[n_sum] is the non-finite counter array
    SegTable st;
    int n_sum;
    double* out_sumsq;
    double* out_bad;
};

__global__ __launch_bounds__(kBlock) void norm_finalize_kernel(const FinalizeArgs a) {
    __shared__ double red[kBlock / kWave];
    const int s = blockIdx.x, q = blockIdx.y;
    const int64_t c0 = get_segment(a.st, s).chunk_start;
    const int64_t c1 = (s + 1 < a.st.n_seg) ? get_segment(a.st, s + 1).chunk_start : a.st.n_chunks;
    const double* __restrict__ p = a.part[q];
    double acc[1] = {0.0};
    // kFinBatch loads in flight per lane, added in index order (the same sum as one load per iteration, bit for bit,
    // without paying one memory round trip per partial — the finalize step is on the serial path of every trial step)
    int64_t i = c0 + threadIdx.x;
    for (; i + (kFinBatch - 1) * kBlock < c1; i += kFinBatch * kBlock) {
        double v[kFinBatch];
#pragma unroll
        for (int u = 0; u < kFinBatch; ++u) v[u] = p[i + u * kBlock];
#pragma unroll
        for (int u = 0; u < kFinBatch; ++u) acc[0] += v[u];
    }
    for (; i < c1; i += kBlock) acc[0] += p[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) {
        if (q < a.n_sum) a.out_sumsq[(int64_t)q * a.st.n_seg + s] = acc[0];
        else a.out_bad[s] = acc[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Device-resident step controller (rk_common.py:324-361, misc.py:85-95) + the next trial step's stage
// times (rk_common.py:72-78).  ONE workgroup: it first reduces the per-chunk partials exactly like
// norm_finalize_kernel (same per-thread strides, same block_sum => the same fp64 sums), then thread 0
// runs the scalar controller in fp64 — the arithmetic of solvers.optimal_step_size / _adaptive_step.
// The accept/reject loop itself stays on the host; this only lets the host enqueue the next trial
// step's first stage before it has read the decision back (stage_combine_sel_kernel below).
// ------------------------------------------------------------------------------------------------
struct CtrlArgs {
    const double* part_sumsq;
    const double* part_bad;
    SegTable st;              // n_seg <= TDEQ_INLINE_SEGMENTS
    tdeq_step_ctrl c;
    int is_f32;               // kind of T: 0 = fp64, 1 = fp32, 2 = bfloat16, 3 = float16 (ctl_seg_norm / ctl_round_T)
    int ratio_kind;           // the type the error ratio is formed in: = is_f32, except per-element tolerances (fp64 = 0: the
                              // reference's quotient and norm promote to the tolerances' type, misc.py:80-82)
    double* out_sumsq;        // [n_seg]   device or pinned host
    double* out_bad;          // [n_seg]
    double* out_ctrl;         // [4] = {accept, dt_next, ratio, t0_next}
    double* ctrl_dev;         // [4] = {accept, sign * T(dt'), t0', dt'}  device
    void* next_times;         // [n_times] of T                          device
    int state_in_dev;         // hipGraph mode: the trial step's (t0, dt) are ctrl_dev[2..3], not c.t0 / c.dt
    int presummed;            // n_seg > TDEQ_INLINE_SEGMENTS: norm_finalize_kernel (one workgroup per segment) has
                              // already written out_sumsq / out_bad; this kernel only runs the controller on them
    const double* in_sumsq;   // presummed only, may be null: the sums come from THESE device arrays (a lock-step
    const double* in_bad;     // sharded solve has all-reduced them over the ranks) and are mirrored to out_sumsq / out_bad
};

__device__ __forceinline__ double ctl_nan_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }
__device__ __forceinline__ double ctl_nan_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }
// torch.clamp on host doubles: NaN stays NaN, else min(max(x, lo), hi)
__device__ __forceinline__ double ctl_clamp(double x, double lo, double hi) {
    if (x != x) return x;
    const double m = x > lo ? x : lo;
    return m < hi ? m : hi;
}

// np.nextafter(x, x - 1) in T (misc.py:174-197, Perturb.PREV)
__device__ __forceinline__ float ctl_prev(float x) {
    const float y = x - 1.0f;
    if (x != x) return x;
    if (x == y) return y;
    if (x == 0.0f) return -__uint_as_float(1u);
    const uint32_t b = __float_as_uint(x);
    return __uint_as_float(x > 0.0f ? b - 1u : b + 1u);
}
__device__ __forceinline__ double ctl_prev(double x) {
    const double y = x - 1.0;
    if (x != x) return x;
    if (x == y) return y;
    if (x == 0.0) return -__longlong_as_double(1LL);
    const long long b = __double_as_longlong(x);
    return __longlong_as_double(x > 0.0 ? b - 1 : b + 1);
}

// ---- reduced-precision states (tkind 2 = bfloat16, 3 = float16; tdeq_kernels_lp.hpp holds the state-sized kernels) ----
// A value rounded to the storage type, kept in a float: what a 0-dim tensor of that type holds.
__device__ __forceinline__ float ctl_rnd16(float x, int tkind) {
    if (tkind == 2) {
        const uint32_t u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return x;
        return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
    }
    return (float)(_Float16)x;
}
__device__ __forceinline__ uint16_t ctl_bits16(float x, int tkind) {     // x already rounded
    if (tkind == 2) return (uint16_t)(__float_as_uint(x) >> 16);
    const _Float16 h = (_Float16)x;
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float ctl_from_bits16(uint16_t b, int tkind) {
    if (tkind == 2) return __uint_as_float((uint32_t)b << 16);
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
// nextafter(x, x - 1) in the storage type (misc.py:174-197, Perturb.PREV; `_scalars._LowScalar.nextafter` on the host)
__device__ __forceinline__ float ctl_prev16(float x, int tkind) {
    const float y = ctl_rnd16(x - 1.0f, tkind);
    if (x != x) return x;
    if (x == y) return y;
    if (x == 0.0f) return ctl_from_bits16(0x8001u, tkind);
    const uint16_t b = ctl_bits16(x, tkind);
    return ctl_from_bits16(x > 0.0f ? (uint16_t)(b - 1u) : (uint16_t)(b + 1u), tkind);
}
// sqrt(mean(|x|^2)) of one segment from its fp64 sum (misc.py:22-23).  fp32 / fp64: sqrt(sum / n) in fp64 (the caller
// rounds the max to T).  Reduced precision: ATen's own sequence — float32 sum / n rounded to the type, sqrt rounded — on
// the sum of ROUNDED squares the 16-bit norm kernels report; a one-element segment reports |x| itself and is squared here
// (tdeq_kernels_lp.hpp norm_term; `_lowp.LowPrecisionHipKernels.read_norms` is the host twin).
__device__ __forceinline__ double ctl_seg_norm(double sum, int64_t numel, int tkind) {
    if (tkind < 2) return __builtin_sqrt(sum / (double)numel);
    float total = (float)sum;
    if (numel == 1) {
        const float a = ctl_rnd16(total, tkind);
        total = ctl_rnd16(a * a, tkind);
    }
    const float mean = ctl_rnd16(total / (float)numel, tkind);
    return (double)ctl_rnd16(__builtin_sqrtf(mean), tkind);
}
// T(x) of a host double for the three kinds of T the controller writes back
__device__ __forceinline__ double ctl_round_T(double x, int tkind) {
    if (tkind == 0) return x;
    if (tkind == 1) return (double)(float)x;
    return (double)ctl_rnd16((float)x, tkind);
}
// stage time i in reduced precision: every operation of `t0 + alpha_i * dt` rounded to the type (rk_common.py:72-78 on
// 0-dim tensors of the state's type), alpha_i already rounded by the host
__device__ __forceinline__ uint16_t ctl_stage_time16(const tdeq_step_ctrl& c, double t0n, double dtn, int i, int tkind) {
    const float t0T = ctl_rnd16((float)t0n, tkind), dtT = ctl_rnd16((float)dtn, tkind);
    const float t1T = ctl_rnd16((float)(t0n + dtn), tkind);
    float tt;
    if ((c.alpha_is_one >> i) & 1u) tt = ctl_prev16(t1T, tkind);
    else tt = ctl_rnd16(t0T + ctl_rnd16((float)c.alpha[i] * dtT, tkind), tkind);
    return ctl_bits16((float)c.time_sign * tt, tkind);
}

// Stage time i of the next trial step (rk_common.py:72-78), one lane per stage.
template <typename T>
__device__ __forceinline__ T ctl_stage_time(const tdeq_step_ctrl& c, double t0n, double dtn, int i) {
    const T t0T = (T)t0n, dtT = (T)dtn, t1T = (T)(t0n + dtn);
    T tt;
    if ((c.alpha_is_one >> i) & 1u) tt = ctl_prev(t1T);
    else tt = t0T + (T)c.alpha[i] * dtT;
    return (T)c.time_sign * tt;
}

// NS = upper bound of the segment count this instantiation serves (1 or kCtrlInlineSegments).  r03: every lane
// accumulates ITS partials of all segments, then ONE block_sum reduces the 2·n_seg values together — one barrier
// instead of two per segment; per segment the per-lane stride, the batch order and the wave / LDS reduction order are
// those of norm_finalize_kernel, so every sum has the same bits as before.  States with more segments than
// kCtrlInlineSegments (an adjoint's augmented state: 9 at cfg3) take the parallel finalize — one workgroup per
// segment — and this kernel's `presummed` form.  Cost of finalize + controller on the 9-segment state of cfg3's backward
// solve, on top of the partial-norm launch (profiles/r03_ctrl_seg_bench.json): 16.4 us (r02, serial per-segment loop) ->
// 13.4 (one barrier) -> 5.6 (parallel finalize + presummed controller); a grouped-prefetch variant of the serial form
// measured slower (17.7 us: 194 VGPRs, guarded loads) and was dropped.
constexpr int kCtrlInlineSegments = 4;
template <int NS>
__global__ __launch_bounds__(kBlock) void norm_finalize_ctrl_kernel(const CtrlArgs a) {
    __shared__ double red[2 * NS * (kBlock / kWave)];
    __shared__ double seg_val[2][NS];
    __shared__ double next_step[2];     // {t0', dt'} broadcast to the lanes that form the stage times
    const int n_seg = a.st.n_seg;
    if (a.presummed) {
        // Many segments (an adjoint state with more than 13 parameter tensors): the per-segment sums come from the
        // parallel finalize launch; ratio = max_s sqrt(sum_s / numel_s) is order-independent (a max; NaN wins), so
        // it is formed by all lanes — each lane its segments, then one block reduction — with the value the serial
        // loop below would give.
        const double* sums = a.in_sumsq ? a.in_sumsq : a.out_sumsq;
        if (a.in_sumsq)
            for (int s = threadIdx.x; s < n_seg; s += kBlock) {
                a.out_sumsq[s] = a.in_sumsq[s];
                a.out_bad[s] = a.in_bad[s];
            }
        double part[2] = {0.0, 0.0};          // {max over this lane's segments, 1 if any of them is NaN}
        for (int s = threadIdx.x; s < a.c.n_norm_seg && s < n_seg; s += kBlock) {
            const int64_t numel = get_segment(a.st, s).numel;
            if (numel == 0) continue;
            const double v = (s == 0 && numel == 1 && a.c.leading_abs && a.ratio_kind >= 2)
                                 ? (double)ctl_rnd16((float)sums[s], a.ratio_kind)      // |x| itself (16-bit adjoint norm)
                                 : ctl_seg_norm(sums[s], numel, a.ratio_kind);
            if (v != v) part[1] = 1.0;
            else part[0] = v > part[0] ? v : part[0];
        }
        // block max of part[0] (wave shuffles + LDS) and block sum of the NaN flags
        double m = part[0];
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const double o = __shfl_down(m, off, kWave);
            m = o > m ? o : m;
        }
        double flag[1] = {part[1]};
        __shared__ double wmax[kBlock / kWave];
        if ((threadIdx.x & (kWave - 1)) == 0) wmax[threadIdx.x / kWave] = m;
        block_sum<1>(flag, red);              // (contains the __syncthreads that also publishes wmax)
        if (threadIdx.x == 0) {
            double mm = wmax[0];
            for (int w = 1; w < kBlock / kWave; ++w) mm = wmax[w] > mm ? wmax[w] : mm;
            seg_val[0][0] = flag[0] != 0.0 ? __builtin_nan("") : mm;
        }
        __syncthreads();
    } else {
        double acc[2 * NS];
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q) acc[q] = 0.0;
        const double* __restrict__ ps = a.part_sumsq;
        const double* __restrict__ pb = a.part_bad;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < n_seg) {
                const int64_t c0 = a.st.inl[s].chunk_start;
                const int64_t c1 = (s + 1 < n_seg) ? a.st.inl[s + 1].chunk_start : a.st.n_chunks;
                double a0 = 0.0, a1 = 0.0;
                int64_t i = c0 + threadIdx.x;
                for (; i + (kFinBatch - 1) * kBlock < c1; i += kFinBatch * kBlock) {     // batched loads, same sum order
                    double v[kFinBatch], w[kFinBatch];
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) {
                        v[u] = ps[i + u * kBlock];
                        w[u] = pb[i + u * kBlock];
                    }
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) {
                        a0 += v[u];
                        a1 += w[u];
                    }
                }
                for (; i < c1; i += kBlock) {
                    a0 += ps[i];
                    a1 += pb[i];
                }
                acc[2 * s] = a0;
                acc[2 * s + 1] = a1;
            }
        }
        block_sum<2 * NS>(acc, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                seg_val[0][s] = acc[2 * s];
                seg_val[1][s] = acc[2 * s + 1];
            }
        }
        __syncthreads();
    }
    const tdeq_step_ctrl& c = a.c;
    if (threadIdx.x == 0) {
        // (value selects, not address selects: choosing between &ctrl_dev[2] and &c.t0 would make the compiler spill
        // the whole argument block to scratch — 840 B of private memory and +10 µs, measured)
        const double dev_t0 = a.ctrl_dev[2], dev_dt = a.ctrl_dev[3];
        const double arg_t0 = c.t0, arg_dt = c.dt;
        const double step_t0 = a.state_in_dev ? dev_t0 : arg_t0;
        const double step_dt = a.state_in_dev ? dev_dt : arg_dt;
        // error ratio: max over segments of sqrt(mean), rounded to T (misc.py:22-33, 80-82)
        double val = 0.0;
        if (a.presummed) val = seg_val[0][0];
        for (int s = 0; s < c.n_norm_seg && s < n_seg && !a.presummed; ++s) {
            const int64_t numel = a.st.inl[s].numel;
            if (numel == 0) continue;
            const double v = (s == 0 && numel == 1 && c.leading_abs && a.ratio_kind >= 2)
                                 ? (double)ctl_rnd16((float)seg_val[0][s], a.ratio_kind)
                                 : ctl_seg_norm(seg_val[0][s], numel, a.ratio_kind);
            val = ctl_nan_max(val, v);
        }
        const double ratio = a.ratio_kind == 1 ? (double)(float)val : val;      // (16-bit kinds: rounded per segment already)
        // accept / reject (rk_common.py:324-330)
        bool accept = ratio <= 1.0;
        if (step_dt > c.max_step) accept = false;
        if (step_dt <= c.min_step) accept = true;
        // next step size (misc.py:85-95), then the clamp of rk_common.py:353
        double dt_next;
        if (ratio == 0.0) {
            dt_next = step_dt * c.ifactor;
        } else {
            const double dfactor = ratio < 1.0 ? 1.0 : c.dfactor;
            const double scaled = c.safety / pow(ratio, c.exponent);
            dt_next = step_dt * ctl_nan_min(c.ifactor, ctl_nan_max(scaled, dfactor));
        }
        dt_next = ctl_clamp(dt_next, c.min_step, c.max_step);
        // the next trial step as the host will set it up (rk_common.py:268-275)
        const double t0n = accept ? step_t0 + step_dt : step_t0;
        double dtn = dt_next;
        if (!__builtin_isfinite(dtn)) dtn = c.min_step;
        dtn = ctl_clamp(dtn, c.min_step, c.max_step);
        // device-side consumers first (the look-ahead stage is next in the stream), then the host's words
        a.ctrl_dev[0] = accept ? 1.0 : 0.0;
        a.ctrl_dev[1] = ctl_round_T(dtn, a.is_f32) * c.time_sign;
        a.ctrl_dev[2] = t0n;
        a.ctrl_dev[3] = dtn;
        next_step[0] = t0n;
        next_step[1] = dtn;
        a.out_ctrl[0] = accept ? 1.0 : 0.0;
        a.out_ctrl[1] = dt_next;
        a.out_ctrl[2] = ratio;
        a.out_ctrl[3] = t0n;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i < c.n_times) {
        if (a.is_f32 >= 2) static_cast<uint16_t*>(a.next_times)[i] = ctl_stage_time16(c, next_step[0], next_step[1], i, a.is_f32);
        else if (a.is_f32) static_cast<float*>(a.next_times)[i] = ctl_stage_time<float>(c, next_step[0], next_step[1], i);
        else static_cast<double*>(a.next_times)[i] = ctl_stage_time<double>(c, next_step[0], next_step[1], i);
    }
    if (i < n_seg && !a.presummed) {
        a.out_sumsq[i] = seg_val[0][i];
        a.out_bad[i] = seg_val[1][i];
    }
}

// ------------------------------------------------------------------------------------------------
// hipGraph mode of the adaptive solvers (small states: a trial step is launch-latency-bound).  One captured graph
// = one trial step; its kernels take the step size from device memory (ctrl_dev[1] = sign * T(dt), written by
// norm_finalize_ctrl_kernel of the previous trial) and work on static buffers:
//   combine_dev_kernel   out = y + sum_j fl_T(coef_j * dt) k_j  [, err_out = sum_j fl_T(ecoef_j * dt) k_j]
//                        — stage_combine / stage_combine_err with a run-time term count (this regime does not
//                        need the unrolled, register-resident streams of the large-state kernels)
//   step_commit_kernel   on accept: (y_prev, f_prev) <- (y_cur, f_cur) ; (y_cur, f_cur) <- (y1, f1)
//                        (rk_common.py:335-352; the previous pair stays available for the dense output)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct CombineDevArgs {
    T* out;
    T* err_out;              // may be null
    const T* y0;
    const T* k[TDEQ_MAX_TERMS];
    T c[TDEQ_MAX_TERMS];     // fl_T(coef_j)
    T e[TDEQ_MAX_TERMS];     // fl_T(err_coef_j)
    int nt;
    const double* dt_dev;    // sign * T(dt)
    int64_t n;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void combine_dev_kernel(const CombineDevArgs<T> a) {
    const T dtT = (T)*a.dt_dev;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        T k0 = a.k[0][i];
        T acc = k0 * (a.c[0] * dtT);
        T err = k0 * (a.e[0] * dtT);
        for (int j = 1; j < a.nt; ++j) {
            const T kj = a.k[j][i];
            acc = acc + kj * (a.c[j] * dtT);
            err = err + kj * (a.e[j] * dtT);
        }
        a.out[i] = a.y0[i] + acc;
        if (a.err_out) a.err_out[i] = err;
    }
}

// The bandwidth-shaped form of the same operation for states that are no longer launch-bound (r02): term count a
// template parameter (streams unrolled, register-resident), one 16-byte element per lane, coefficients formed once per
// lane as fl_T(fl_T(coef_j) * fl_T(dt)) from the device-resident step size — the rounding sequence of stage_combine /
// stage_combine_err and of combine_dev_kernel, so all three agree bit for bit.
template <typename T, int NT>
struct CombineDevTArgs {
    CombineArgs<T, NT> c;    // c.c[j] = fl_T(coef_j) (NOT yet times dt)
    T* err_out;              // ERR only
    T e[NT];                 // fl_T(err_coef_j)
    const double* dt_dev;    // sign * T(dt)
};

template <typename T, int NT, bool VEC, bool ERR>
__global__ __launch_bounds__(kBlock) void combine_devdt_kernel(const CombineDevTArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.c.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const E* __restrict__ y0 = reinterpret_cast<const E*>(a.c.y0);
    E* __restrict__ out = reinterpret_cast<E*>(a.c.out);
    E* __restrict__ eo = reinterpret_cast<E*>(a.err_out);
    // r06: the first pass's stream loads are issued BEFORE the step size is fetched.  *dt_dev was written by the previous
    // launch's controller on another CU: a scalar load that misses the scalar cache and the L2 — ~1.5 us during which a
    // wave of the r05 form (dt first, coefficients, then the loads) had nothing in flight; a wave of this kernel lives ~1 us
    // otherwise (profiles/r06_shard_l2.json: 5.1-5.7 us per captured combine against 3.4-3.8 us for the host-dt kernels).
    // The step size itself travels as a VECTOR load (an agent-scope relaxed atomic load is never selected as a scalar
    // load): issued first, it shares the in-order vmcnt queue with the stream loads behind it, so the one wait before the
    // arithmetic covers both — a scalar load would put its own full round trip (kernarg pointer, then the value) in series
    // with them, and every later kernarg wait would wait for it too (lgkmcnt counts out-of-order returns).
    const double dtd = __hip_atomic_load(a.dt_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    E kk[NT];
    E yv{};
    if (i0 < ne) {
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(a.c.k[j])[i0];
        yv = y0[i0];
    }
    const T dtT = (T)dtd;
    T c[NT], e[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        c[j] = a.c.c[j] * dtT;
        e[j] = ERR ? a.e[j] * dtT : (T)0;
    }
    if (i0 < ne) {
        E acc = kk[0] * c[0];
#pragma unroll
        for (int j = 1; j < NT; ++j) acc = acc + kk[j] * c[j];
        out[i0] = yv + acc;
        if (ERR) {
            E err = kk[0] * e[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) err = err + kk[j] * e[j];
            eo[i0] = err;
        }
    }
    for (int64_t i = i0 + stride; i < ne; i += stride) {        // (only beyond 65536 workgroups)
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(a.c.k[j])[i];
        E acc = kk[0] * c[0];
#pragma unroll
        for (int j = 1; j < NT; ++j) acc = acc + kk[j] * c[j];
        out[i] = y0[i] + acc;
        if (ERR) {
            E err = kk[0] * e[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) err = err + kk[j] * e[j];
            eo[i] = err;
        }
    }
    if (VEC) {   // scalar tail (n % L elements)
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.c.n) {
            T acc = a.c.k[0][t] * c[0];
            T err = a.c.k[0][t] * e[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                acc = acc + a.c.k[j][t] * c[j];
                err = err + a.c.k[j][t] * e[j];
            }
            a.c.out[t] = a.c.y0[t] + acc;
            if (ERR) a.err_out[t] = err;
        }
    }
}

template <typename T>
struct CommitArgs {
    T* y_prev;
    T* f_prev;
    T* y_cur;
    T* f_cur;
    const T* y1;
    const T* f1;
    const double* ctrl_dev;
    int64_t n;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void step_commit_kernel(const CommitArgs<T> a) {
    if (a.ctrl_dev[0] == 0.0) return;
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    E* __restrict__ yp = reinterpret_cast<E*>(a.y_prev);
    E* __restrict__ fp = reinterpret_cast<E*>(a.f_prev);
    E* __restrict__ yc = reinterpret_cast<E*>(a.y_cur);
    E* __restrict__ fc = reinterpret_cast<E*>(a.f_cur);
    const E* __restrict__ y1 = reinterpret_cast<const E*>(a.y1);
    const E* __restrict__ f1 = reinterpret_cast<const E*>(a.f1);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        const E a0 = yc[i], a1 = fc[i], a2 = y1[i], a3 = f1[i];
        yp[i] = a0;
        fp[i] = a1;
        yc[i] = a2;
        fc[i] = a3;
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            a.y_prev[t] = a.y_cur[t];
            a.f_prev[t] = a.f_cur[t];
            a.y_cur[t] = a.y1[t];
            a.f_cur[t] = a.f1[t];
        }
    }
}

// First stage of the NEXT trial step on the pair the controller selected (see norm_finalize_ctrl_kernel):
//   out = y + fl_T(coef * dt') * f ,  (y, f) = accept ? (y_acc, f_acc) : (y_rej, f_rej)
template <typename T>
struct SelArgs {
    T* out;
    const T* y_acc;
    const T* f_acc;
    const T* y_rej;
    const T* f_rej;
    T coef;                 // fl_T(coef)
    const double* ctrl_dev; // {accept, sign * T(dt')}
    int64_t n;
};

template <typename T, bool VEC, int POLICY = 0>
__global__ __launch_bounds__(kBlock) void stage_combine_sel_kernel(const SelArgs<T> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    E* __restrict__ out = reinterpret_cast<E*>(a.out);
    // The ACCEPTED pair is loaded before the controller's words are looked at: a wave of this kernel lives ~1 us, and
    // waiting for the (scalar) load of {accept, dt'} before the first vector load can be issued was a serial third of it.
    // Steps are accepted in the overwhelming majority; a rejected one re-loads from the other pair (one more read of 2 N
    // words, then).  Same arithmetic either way.
    const int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    E y_spec{}, f_spec{};
    if (i0 < ne) {
        y_spec = ld_stream<POLICY>(reinterpret_cast<const E*>(a.y_acc) + i0);
        f_spec = ld_stream<POLICY>(reinterpret_cast<const E*>(a.f_acc) + i0);
    }
    const bool accept = a.ctrl_dev[0] != 0.0;
    const T c = a.coef * (T)a.ctrl_dev[1];
    const T* ys = accept ? a.y_acc : a.y_rej;
    const T* fs = accept ? a.f_acc : a.f_rej;
    const E* __restrict__ y = reinterpret_cast<const E*>(ys);
    const E* __restrict__ f = reinterpret_cast<const E*>(fs);
    if (i0 < ne) {
        if (!accept) {
            y_spec = ld_stream<POLICY>(y + i0);
            f_spec = ld_stream<POLICY>(f + i0);
        }
        st_stream<POLICY>(out + i0, y_spec + f_spec * c);
    }
    for (int64_t i = i0 + stride; i < ne; i += stride)      // (only beyond 65536 workgroups)
        st_stream<POLICY>(out + i, ld_stream<POLICY>(y + i) + ld_stream<POLICY>(f + i) * c);
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) a.out[t] = ys[t] + fs[t] * c;
    }
}

// ------------------------------------------------------------------------------------------------
// Dense output (rk_common.py:363-369, interp.py:1-48).  Op order mirrors interp.py:17-21 / :42-47.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT>
struct DenseArgs {
    T* out;          // n elements (dense_eval) or 5*n (interp_fit: e,d,c,b,a)
    const T* y0;
    const T* y1;
    const T* f0;
    const T* f1;
    const T* k[NT];
    T c[NT];
    T dt;
    T x;
    int64_t n;
};

template <typename T, typename E>
struct Quartic { E e, d, c, b, a; };

template <typename T, int NT, typename E>
__device__ __forceinline__ Quartic<T, E> fit_one(const DenseArgs<T, NT>& a, const E& y0, const E& y1,
                                                 const E& f0, const E& f1, const E (&kk)[NT]) {
    E acc = kk[0] * a.c[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) acc = acc + kk[j] * a.c[j];
    const E ymid = y0 + acc;
    const T dt = a.dt;
    const T two_dt = (T)2 * dt;
    Quartic<T, E> q;
    q.a = ((f1 - f0) * two_dt - (y1 + y0) * (T)8) + ymid * (T)16;
    q.b = (((f0 * (T)5 - f1 * (T)3) * dt + y0 * (T)18) + y1 * (T)14) - ymid * (T)32;
    q.c = (((f1 - f0 * (T)4) * dt - y0 * (T)11) - y1 * (T)5) + ymid * (T)16;
    q.d = f0 * dt;
    q.e = y0;
    return q;
}

template <typename T, typename E>
__device__ __forceinline__ E eval_one(const Quartic<T, E>& q, T x) {
    E total = q.e + q.d * x;
    T xp = x * x;
    total = total + q.c * xp;
    xp = xp * x;
    total = total + q.b * xp;
    xp = xp * x;
    total = total + q.a * xp;
    return total;
}

template <typename T, int NT, bool FIT_ONLY, bool VEC>
__global__ __launch_bounds__(kBlock) void dense_kernel(const DenseArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        E kk[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(a.k[j])[i];
        const Quartic<T, E> q = fit_one<T, NT, E>(
            a, reinterpret_cast<const E*>(a.y0)[i], reinterpret_cast<const E*>(a.y1)[i],
            reinterpret_cast<const E*>(a.f0)[i], reinterpret_cast<const E*>(a.f1)[i], kk);
        if (FIT_ONLY) {
            reinterpret_cast<E*>(a.out)[i] = q.e;
            reinterpret_cast<E*>(a.out + a.n)[i] = q.d;
            reinterpret_cast<E*>(a.out + 2 * a.n)[i] = q.c;
            reinterpret_cast<E*>(a.out + 3 * a.n)[i] = q.b;
            reinterpret_cast<E*>(a.out + 4 * a.n)[i] = q.a;
        } else {
            reinterpret_cast<E*>(a.out)[i] = eval_one<T, E>(q, a.x);
        }
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = a.k[j][t];
            const Quartic<T, T> q = fit_one<T, NT, T>(a, a.y0[t], a.y1[t], a.f0[t], a.f1[t], kk);
            if (FIT_ONLY) {
                a.out[t] = q.e;
                a.out[a.n + t] = q.d;
                a.out[2 * a.n + t] = q.c;
                a.out[3 * a.n + t] = q.b;
                a.out[4 * a.n + t] = q.a;
            } else {
                a.out[t] = eval_one<T, T>(q, a.x);
            }
        }
    }
}

// Several output times inside ONE accepted step (solvers.py:28-35 calls _interp_evaluate once per output time,
// rk_common.py:243-250): the quartic is fitted once per element and evaluated at x_0..x_{m-1}; row q of the
// output goes to out + q*out_stride.  (8 + m) words per element instead of 9 m; same operations per output as
// dense_kernel, so every row is bit-identical to a single-output launch.
constexpr int kMaxDenseOutputs = 16;

template <typename T, int NT>
struct DenseMultiArgs {
    DenseArgs<T, NT> d;      // d.out = first output row, d.x unused
    T xs[kMaxDenseOutputs];
    int m;
    int64_t out_stride;      // elements between consecutive output rows
};

template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(kBlock) void dense_multi_kernel(const DenseMultiArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const DenseArgs<T, NT>& d = a.d;
    const int64_t ne = d.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        E kk[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(d.k[j])[i];
        const Quartic<T, E> q = fit_one<T, NT, E>(
            d, reinterpret_cast<const E*>(d.y0)[i], reinterpret_cast<const E*>(d.y1)[i],
            reinterpret_cast<const E*>(d.f0)[i], reinterpret_cast<const E*>(d.f1)[i], kk);
        for (int r = 0; r < a.m; ++r)
            reinterpret_cast<E*>(d.out + (int64_t)r * a.out_stride)[i] = eval_one<T, E>(q, a.xs[r]);
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < d.n) {
            T kk[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[j] = d.k[j][t];
            const Quartic<T, T> q = fit_one<T, NT, T>(d, d.y0[t], d.y1[t], d.f0[t], d.f1[t], kk);
            for (int r = 0; r < a.m; ++r) d.out[(int64_t)r * a.out_stride + t] = eval_one<T, T>(q, a.xs[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rk4 3/8 rule (rk_common.py:110-118) and the fixed-grid linear interpolation (solvers.py:175-181).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Rk4Args {
    T* out;
    const T* y0;
    const T* k1;
    const T* k2;
    const T* k3;
    const T* k4;
    T dt;
    T third;   // fl_T(1/3)
    int64_t n;
    const double* dt_dev;   // non-null (hipGraph mode): the step size is read from device memory (grid_advance_kernel)
};

template <typename T, int STAGE, typename E>
__device__ __forceinline__ E rk4_one(const Rk4Args<T>& a, T dt, int64_t i) {
    const E y0 = reinterpret_cast<const E*>(a.y0)[i];
    const E k1 = reinterpret_cast<const E*>(a.k1)[i];
    if (STAGE == 1) return y0 + (k1 * dt) * a.third;
    const E k2 = reinterpret_cast<const E*>(a.k2)[i];
    if (STAGE == 2) return y0 + (k2 - k1 * a.third) * dt;
    const E k3 = reinterpret_cast<const E*>(a.k3)[i];
    if (STAGE == 3) return y0 + ((k1 - k2) + k3) * dt;
    const E k4 = reinterpret_cast<const E*>(a.k4)[i];
    return y0 + (((k1 + (k2 + k3) * (T)3) + k4) * dt) * (T)0.125;
}

template <typename T, int STAGE, bool VEC>
__global__ __launch_bounds__(kBlock) void rk4_kernel(const Rk4Args<T> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const T dt = a.dt_dev ? (T)*a.dt_dev : a.dt;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride)
        reinterpret_cast<E*>(a.out)[i] = rk4_one<T, STAGE, E>(a, dt, i);
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) a.out[t] = rk4_one<T, STAGE, T>(a, dt, t);
    }
}

// ------------------------------------------------------------------------------------------------
// hipGraph mode of the fixed-grid rk4 solver: one captured graph = one step (4 func evaluations + 4 stage
// kernels + the two kernels below), replayed once per grid interval — the launch-latency-bound regime of small
// states (cfg1).  Everything that changes from step to step lives in device memory:
//   grid_advance_kernel   counter += 1; t0 = grid[c], t1 = grid[c+1]; dt = t1 - t0 in the grid's dtype
//                         (solvers.py:110-112); the four stage times t0, t0 + dt/3, t0 + 2dt/3, t1
//                         (rk_common.py:110-118, perturbed at the ends if `perturb`, misc.py:174-197), cast to
//                         the state dtype and multiplied by the time sign; dt_out = sign * dt (double)
//   grid_commit_kernel    solution[c + 1] = y_new ; y_cur = y_new       (solvers.py:113-127 with grid == t)
// ------------------------------------------------------------------------------------------------
struct GridAdvanceArgs {
    const void* grid;      // [n_grid] of float or double
    int grid_is_f32;
    int64_t n_grid;
    int64_t* counter;
    int perturb;           // fixed-grid `perturb` option: NEXT at t0, PREV at t1
    double sign;
    void* times_out;       // [4] of T
    int state_is_f32;
    double* dt_out;
};

__device__ __forceinline__ float ctl_next(float x) {      // np.nextafter(x, x + 1)
    const float y = x + 1.0f;
    if (x != x) return x;
    if (x == y) return y;
    if (x == 0.0f) return __uint_as_float(1u);
    const uint32_t b = __float_as_uint(x);
    return __uint_as_float(x > 0.0f ? b + 1u : b - 1u);
}
__device__ __forceinline__ double ctl_next(double x) {
    const double y = x + 1.0;
    if (x != x) return x;
    if (x == y) return y;
    if (x == 0.0) return __longlong_as_double(1LL);
    const long long b = __double_as_longlong(x);
    return __longlong_as_double(x > 0.0 ? b + 1 : b - 1);
}

template <typename G, typename T>
__device__ __forceinline__ void grid_times(const GridAdvanceArgs& a, int64_t c) {
    const G* grid = static_cast<const G*>(a.grid);
    const G t0 = grid[c], t1 = grid[c + 1];
    const G dt = t1 - t0;
    const G third = (G)(1.0 / 3.0), two_thirds = (G)(2.0 / 3.0);
    const G ts[4] = {t0, t0 + dt * third, t0 + dt * two_thirds, t1};
    T* out = static_cast<T*>(a.times_out);
    for (int i = 0; i < 4; ++i) {
        T tt = (T)ts[i];
        if (a.perturb && i == 0) tt = ctl_next(tt);
        if (a.perturb && i == 3) tt = ctl_prev(tt);
        out[i] = (T)a.sign * tt;
    }
    *a.dt_out = (double)dt * a.sign;
}

__global__ void grid_advance_kernel(const GridAdvanceArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t c = *a.counter + 1;
    *a.counter = c;
    if (c + 1 >= a.n_grid) return;       // past the last interval: nothing to prepare
    if (a.grid_is_f32) {
        if (a.state_is_f32) grid_times<float, float>(a, c); else grid_times<float, double>(a, c);
    } else {
        if (a.state_is_f32) grid_times<double, float>(a, c); else grid_times<double, double>(a, c);
    }
}

// The same for ANY explicit fixed-grid method (r03: euler, midpoint, heun2, heun3 in hipGraph mode): the stage times of
// a step are t0 + dt * fl_G(frac_i) (`dt * c` with the Python float c rounded to the grid dtype, solvers._tmul) or t1
// itself, optionally perturbed (NEXT / PREV when the solver's `perturb` option is on: fixed_grid.py, rk_common.py:110-157).
constexpr int kMaxGridStages = 4;
struct GridStagesArgs {
    GridAdvanceArgs base;
    int n_times;
    double frac[kMaxGridStages];
    int mode[kMaxGridStages];   // bit 0: the time is t1 (not t0 + dt*frac); bit 1: Perturb.NEXT; bit 2: Perturb.PREV
};

template <typename G, typename T>
__device__ __forceinline__ void grid_stage_times(const GridStagesArgs& a, int64_t c) {
    const G* grid = static_cast<const G*>(a.base.grid);
    const G t0 = grid[c], t1 = grid[c + 1];
    const G dt = t1 - t0;
    T* out = static_cast<T*>(a.base.times_out);
    for (int i = 0; i < a.n_times; ++i) {
        const G tg = (a.mode[i] & 1) ? t1 : (a.frac[i] == 0.0 ? t0 : t0 + dt * (G)a.frac[i]);
        T tt = (T)tg;
        if (a.base.perturb && (a.mode[i] & 2)) tt = ctl_next(tt);
        if (a.base.perturb && (a.mode[i] & 4)) tt = ctl_prev(tt);
        out[i] = (T)a.base.sign * tt;
    }
    *a.base.dt_out = (double)dt * a.base.sign;
}

__global__ void grid_advance_stages_kernel(const GridStagesArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t c = *a.base.counter + 1;
    *a.base.counter = c;
    if (c + 1 >= a.base.n_grid) return;       // past the last interval: nothing to prepare
    if (a.base.grid_is_f32) {
        if (a.base.state_is_f32) grid_stage_times<float, float>(a, c); else grid_stage_times<float, double>(a, c);
    } else {
        if (a.base.state_is_f32) grid_stage_times<double, float>(a, c); else grid_stage_times<double, double>(a, c);
    }
}

template <typename T>
struct GridCommitArgs {
    T* solution;           // [n_grid, row_stride]
    int64_t row_stride;
    T* y_cur;
    const T* y_new;
    const int64_t* counter;
    int64_t n;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void grid_commit_kernel(const GridCommitArgs<T> a) {
    T* row = a.solution + (*a.counter + 1) * a.row_stride;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        const T v = a.y_new[i];
        row[i] = v;
        a.y_cur[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Low-order fixed-grid steps (rk_common.py:121-157, fixed_grid.py:6-60) and generic weighted sums.
//   MODE 0:  out = y0 + dt * ((k0*w0 + k1*w1) + ...)      rk2/rk3 increments and rk3's third stage input
//   MODE 1:  out = y0 + (dt * k0) * w0                    rk2/rk3 second stage input  `y0 + dt * k1 * a21`
//   MODE 2:  out = (x0*w0 + x1*w1) + ...                  cubic Hermite output interpolation (solvers.py:166-173)
// ------------------------------------------------------------------------------------------------
template <typename T, int NT>
struct FixedArgs {
    T* out;
    const T* y0;        // unused in MODE 2
    const T* k[NT];
    T w[NT];
    T dt;
    int64_t n;
    const double* dt_dev;   // non-null (hipGraph mode of the fixed-grid methods): the step size is read on the device
};

template <typename T, int NT, int MODE, typename E>
__device__ __forceinline__ E fixed_one(const FixedArgs<T, NT>& a, T dt, int64_t i) {
    E kk[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) kk[j] = reinterpret_cast<const E*>(a.k[j])[i];
    if (MODE == 1) return reinterpret_cast<const E*>(a.y0)[i] + (kk[0] * dt) * a.w[0];
    E acc = kk[0] * a.w[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) acc = acc + kk[j] * a.w[j];
    if (MODE == 2) return acc;
    return reinterpret_cast<const E*>(a.y0)[i] + acc * dt;
}

template <typename T, int NT, int MODE, bool VEC>
__global__ __launch_bounds__(kBlock) void fixed_stage_kernel(const FixedArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const T dt = a.dt_dev ? (T)*a.dt_dev : a.dt;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride)
        reinterpret_cast<E*>(a.out)[i] = fixed_one<T, NT, MODE, E>(a, dt, i);
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) a.out[t] = fixed_one<T, NT, MODE, T>(a, dt, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the linear kernels (differentiable plain odeint).  Every elementwise kernel above is
// out = sum_m w_m(dt, x) * X_m, so its VJP is  grad X_m = w_m * g  (scale_many: g read once, NT stores)
// and  grad s = sum_m dw_m/ds * <g, X_m>  for a time-like scalar s (multi_dot: fp64 dots, one pass).
// ------------------------------------------------------------------------------------------------
template <typename T, int NT>
struct ScaleArgs {
    T* out[NT];
    const T* g;
    T w[NT];
    int64_t n;
};

template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(kBlock) void scale_many_kernel(const ScaleArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        const E g = reinterpret_cast<const E*>(a.g)[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) reinterpret_cast<E*>(a.out[j])[i] = g * a.w[j];
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) {
            const T g = a.g[t];
#pragma unroll
            for (int j = 0; j < NT; ++j) a.out[j][t] = g * a.w[j];
        }
    }
}

template <typename T, int NT>
struct DotArgs {
    const T* g;
    const T* x[NT];
    int64_t n;
    int64_t chunk;
    double* part;   // [NT][n_chunks]
    int64_t n_chunks;
};

template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(kBlock) void multi_dot_kernel(const DotArgs<T, NT> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[NT * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const int64_t base = b * a.chunk;
    int64_t valid = a.n - base;
    valid = valid > a.chunk ? a.chunk : valid;
    double acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = 0.0;
    int64_t t0 = 0;
    if (VEC) {
        const int64_t nv = valid / L;
        const V* g = reinterpret_cast<const V*>(a.g + base);
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            const V gv = g[i];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const V xv = reinterpret_cast<const V*>(a.x[j] + base)[i];
#pragma unroll
                for (int q = 0; q < L; ++q) acc[j] += (double)gv[q] * (double)xv[q];
            }
        }
        t0 = nv * L;
    }
    for (int64_t t = t0 + threadIdx.x; t < valid; t += kBlock) {
        const double gv = (double)a.g[base + t];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] += gv * (double)a.x[j][base + t];
    }
    block_sum<NT>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) a.part[(int64_t)j * a.n_chunks + b] = acc[j];
    }
}

// out[j] = sum over chunks of part[j][*], fixed order; one workgroup per j.
struct DotFinalizeArgs {
    const double* part;
    int64_t n_chunks;
    double* out;
};

__global__ __launch_bounds__(kBlock) void dot_finalize_kernel(const DotFinalizeArgs a) {
    __shared__ double red[kBlock / kWave];
    const int j = blockIdx.x;
    const double* p = a.part + (int64_t)j * a.n_chunks;
    double acc[1] = {0.0};
    for (int64_t i = threadIdx.x; i < a.n_chunks; i += kBlock) acc[0] += p[i];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) a.out[j] = acc[0];
}

template <typename T>
struct LerpArgs {
    T* out;
    const T* y0;
    const T* y1;
    T slope;
    int64_t n;
};

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void lerp_kernel(const LerpArgs<T> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride) {
        const E y0 = reinterpret_cast<const E*>(a.y0)[i];
        const E y1 = reinterpret_cast<const E*>(a.y1)[i];
        reinterpret_cast<E*>(a.out)[i] = y0 + (y1 - y0) * a.slope;
    }
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) a.out[t] = a.y0[t] + (a.y1[t] - a.y0[t]) * a.slope;
    }
}

// ------------------------------------------------------------------------------------------------
// Segment packing: the outputs of a tuple-valued func — and of the adjoint's augmented dynamics
// (vjp_t, f, vjp_y, vjp_θ...) — go into ONE flat chunk-aligned buffer with ONE launch:
//   out[chunk_start_s*chunk + i] = scale_s * src_s[i]   (i < numel_s) ;  padding and missing sources -> 0
// replacing the reference's torch.cat of the pieces (misc.py:145), the `-adj_y` negation (adjoint.py:95),
// the zeros_like for absent gradients (adjoint.py:99-103) and _ReverseFunc's multiply (misc.py:165);
// scale_s is +1 or -1, so the products are exact.  One workgroup per chunk.
// ------------------------------------------------------------------------------------------------
struct PackSeg {
    const void* src;       // may be null: the segment is zero-filled
    int64_t chunk_start;
    int64_t numel;
    double scale;
    int vec_ok;            // src is 16-byte aligned
};

template <typename T>
struct PackArgs {
    T* out;
    PackSeg seg[TDEQ_INLINE_SEGMENTS];
    int n_seg;
    int64_t chunk;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void pack_segments_kernel(const PackArgs<T> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    const int64_t b = blockIdx.x;
    int s = 0;
    for (int q = 1; q < a.n_seg; ++q) s = (a.seg[q].chunk_start <= b) ? q : s;
    const PackSeg seg = a.seg[s];
    const int64_t local0 = (b - seg.chunk_start) * a.chunk;
    int64_t valid = seg.numel - local0;
    valid = valid < 0 ? 0 : (valid > a.chunk ? a.chunk : valid);
    if (!seg.src) valid = 0;
    T* __restrict__ out = a.out + b * a.chunk;
    const T* __restrict__ src = static_cast<const T*>(seg.src) + local0;
    const T sc = (T)seg.scale;
    int64_t done = 0;
    if (seg.vec_ok) {          // local0 is a multiple of the chunk (>= 1024 elements): src + local0 stays aligned
        const int64_t nv = valid / L;
        for (int64_t i = threadIdx.x; i < nv; i += kBlock)
            reinterpret_cast<V*>(out)[i] = reinterpret_cast<const V*>(src)[i] * sc;
        done = nv * L;
    }
    for (int64_t t = done + threadIdx.x; t < valid; t += kBlock) out[t] = src[t] * sc;
    for (int64_t t = valid + threadIdx.x; t < a.chunk; t += kBlock) out[t] = (T)0;
}

struct FillArgs {
    void* dst;
    double v[16];
    int n;
    int dtype;
};

__global__ void fill_scalars_kernel(const FillArgs a) {
    const int i = threadIdx.x;
    if (i < a.n) {
        if (a.dtype == TDEQ_F32) reinterpret_cast<float*>(a.dst)[i] = (float)a.v[i];
        else reinterpret_cast<double*>(a.dst)[i] = a.v[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Adams–Bashforth(–Moulton) multistep steps on a fixed grid (fixed_adams.py:196-223).  Same streaming
// shape as the RK combines: up to 11 history tensors f_{n-j} (separate contiguous tensors, newest first)
// are read ONCE per step and feed both the predictor and the constant part of the corrector:
//   dy    = (cb_0*f_0 + cb_1*f_1) + ...     cb_j = fl_T(dt*b_j), dt*b_j formed in fp64 by the host (:205)
//   y_out = y0 + dy                         (:213 first iteration; solvers.py:115 in the explicit method)
//   delta = dt * ((cm_0*f_0 + cm_1*f_1) + ...)    cm_j = fl_T(m_{j+1}), dt rounded to T (:210)
// Python's `sum` starts from the int 0, and 0 + v == v.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT>
struct AdamsPredictArgs {
    T* y_out;
    T* dy_out;      // IMPLICIT only
    T* delta_out;   // IMPLICIT only
    const T* y0;
    const T* f[NT];
    T cb[NT];
    T cm[NT];
    T dt;
    int64_t n;
};

template <typename T, int NT, bool IMPLICIT, typename E>
__device__ __forceinline__ void adams_predict_one(const AdamsPredictArgs<T, NT>& a, int64_t i) {
    E fv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) fv[j] = reinterpret_cast<const E*>(a.f[j])[i];
    const E y0 = reinterpret_cast<const E*>(a.y0)[i];
    E dy = fv[0] * a.cb[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) dy = dy + fv[j] * a.cb[j];
    reinterpret_cast<E*>(a.y_out)[i] = y0 + dy;
    if (IMPLICIT) {
        E sm = fv[0] * a.cm[0];
#pragma unroll
        for (int j = 1; j < NT; ++j) sm = sm + fv[j] * a.cm[j];
        reinterpret_cast<E*>(a.dy_out)[i] = dy;
        reinterpret_cast<E*>(a.delta_out)[i] = sm * a.dt;
    }
}

template <typename T, int NT, bool IMPLICIT, bool VEC>
__global__ __launch_bounds__(kBlock) void adams_predict_kernel(const AdamsPredictArgs<T, NT> a) {
    using E = typename std::conditional<VEC, typename VecOf<T>::type, T>::type;
    constexpr int L = VEC ? VecOf<T>::L : 1;
    const int64_t ne = a.n / L;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ne; i += stride)
        adams_predict_one<T, NT, IMPLICIT, E>(a, i);
    if (VEC) {
        const int64_t t = ne * L + threadIdx.x;
        if (blockIdx.x == 0 && t < a.n) adams_predict_one<T, NT, IMPLICIT, T>(a, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Adams–Moulton corrector iteration + its convergence test in ONE pass (fixed_adams.py:212-216, 189-192):
//   dy    = c*f + delta                      c = fl_T(dt*m_0), dt*m_0 formed in fp64 by the host (:214)
//   y_out = y0 + dy                          input of the next evaluation, and y1 once converged
//   ratio = |dy_old - dy| / (atol + rtol*max(|dy_old|, |dy|))      (misc.py:80-82 with the l-inf norm)
// The reference reduces `ratio` with abs().max() and tests `< 1`; that boolean equals "no element has
// !(ratio < 1)" (NaN included: torch.max propagates NaN and NaN < 1 is false), so the kernel COUNTS those
// elements per chunk and the usual finalize launch adds the counts: exact, order-independent.
// One workgroup per chunk like the norm kernels; the padding of a segmented state is written (it is zero
// in every input) but not counted.  COMPUTE = false: test only, dy is read from dy_out.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct AdamsCorrectArgs {
    T* y_out;
    T* dy_out;
    const T* f;
    const T* delta;
    const T* dy_old;
    const T* y0;
    T c;
    int64_t n;
    SegTable st;
    double* part_count;
    double* part_bad;
};

template <typename T>
__device__ __forceinline__ void adams_test(T d_old, T d_new, T rtol, T atol, bool counted, double& cnt, double& bad) {
    const T e = sabs(d_old - d_new);
    const T tol = atol + rtol * smax(sabs(d_old), sabs(d_new));
    const T r = e / tol;
    cnt += (counted && !(r < (T)1)) ? 1.0 : 0.0;
    bad += (counted && !__builtin_isfinite(d_new)) ? 1.0 : 0.0;
}

template <typename T, bool COMPUTE, bool VEC>
__global__ __launch_bounds__(kBlock) void adams_correct_kernel(const AdamsCorrectArgs<T> a) {
    using V = typename VecOf<T>::type;
    constexpr int L = VecOf<T>::L;
    __shared__ double red[2 * (kBlock / kWave)];
    const int64_t b = blockIdx.x;
    const tdeq_segment seg = find_segment(a.st, b);
    const int64_t base = b * a.st.chunk;
    int64_t valid = seg.numel - (b - seg.chunk_start) * a.st.chunk;
    valid = valid < 0 ? 0 : (valid > a.st.chunk ? a.st.chunk : valid);
    int64_t lim = a.n - base;
    lim = lim < 0 ? 0 : (lim > a.st.chunk ? a.st.chunk : lim);
    if (!COMPUTE) lim = valid;
    const T rtol = (T)seg.rtol, atol = (T)seg.atol;
    double acc[2] = {0.0, 0.0};
    int64_t t0 = 0;
    if (VEC) {
        const int64_t nv = lim / L;
        for (int64_t i = threadIdx.x; i < nv; i += kBlock) {
            const V d_old = reinterpret_cast<const V*>(a.dy_old + base)[i];
            V d_new;
            if (COMPUTE) {
                const V fv = reinterpret_cast<const V*>(a.f + base)[i];
                const V dl = reinterpret_cast<const V*>(a.delta + base)[i];
                const V y0 = reinterpret_cast<const V*>(a.y0 + base)[i];
                d_new = fv * a.c + dl;
                reinterpret_cast<V*>(a.dy_out + base)[i] = d_new;
                reinterpret_cast<V*>(a.y_out + base)[i] = y0 + d_new;
            } else {
                d_new = reinterpret_cast<const V*>(a.dy_out + base)[i];
            }
#pragma unroll
            for (int q = 0; q < L; ++q) adams_test<T>(d_old[q], d_new[q], rtol, atol, i * L + q < valid, acc[0], acc[1]);
        }
        t0 = nv * L;
    }
    for (int64_t t = t0 + threadIdx.x; t < lim; t += kBlock) {
        const T d_old = a.dy_old[base + t];
        T d_new;
        if (COMPUTE) {
            d_new = a.f[base + t] * a.c + a.delta[base + t];
            a.dy_out[base + t] = d_new;
            a.y_out[base + t] = a.y0[base + t] + d_new;
        } else {
            d_new = a.dy_out[base + t];
        }
        adams_test<T>(d_old, d_new, rtol, atol, t < valid, acc[0], acc[1]);
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        a.part_count[b] = acc[0];
        a.part_bad[b] = acc[1];
    }
}

}  // namespace tdeq
