// tdeq_abi.hip — extern "C" entry points of libtdeq_hip.so (declared in include/tdeq_hip.h).
// Host-side validation + template dispatch + launch; no allocation, no synchronisation, no globals.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "tdeq_kernels.hpp"
#include "tdeq_kernels_complex.hpp"
#include "tdeq_kernels_lp.hpp"

namespace {
using namespace tdeq;

// Launch geometry for the streaming (elementwise) kernels: one 16-byte element per lane, i.e. enough
// workgroups to cover the tensor exactly once (tools/sweep_combine.hip on the MI355X: 8192 workgroups
// of one float4 per lane beat a 2048-workgroup grid-stride loop by 6-8 %, cold and warm); only beyond
// kMaxGrid workgroups does the kernel grid-stride.
constexpr int64_t kMaxGrid = 1 << 16;

inline unsigned stream_grid(int64_t n_items, int items_per_block) {
    int64_t g = (n_items + items_per_block - 1) / items_per_block;
    const int64_t cap = kMaxGrid;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return (int)e;
}

// Cache policy of a streaming launch (bit 0: non-temporal loads, bit 1: non-temporal stores), chosen from the bytes
// its streams touch.  Measured on the MI355X (tools/stream_count.hip, profiles/r05_stream_count.json): with every byte
// coming from DRAM, 5 read + 2 write streams run at 5.4 TB/s with the default policy and at 6.0 TB/s with both hints
// (10 R + 4 W fp64: 5.6 -> 5.9-6.3) — the hints keep single-use lines from churning through L2 / the Infinity Cache.
// A launch whose streams fit the 256 MiB Infinity Cache is better off WITHOUT them (its inputs were just written by
// func and are still resident; its outputs are read by func next).  TDEQ_COMBINE_POLICY = 0..3 forces one policy for
// every launch (tuning runs); TDEQ_NT_THRESHOLD_MB moves the switch point.
constexpr int64_t kNtDefaultThreshold = (int64_t)TDEQ_NT_THRESHOLD_DEFAULT_MB << 20;
constexpr int kNtPolicy = 3;

inline int stream_policy(int64_t stream_bytes) {
    static const int forced = [] {
        const char* e = getenv("TDEQ_COMBINE_POLICY");
        if (!e || !*e || *e == 'a') return -1;      // unset / "auto"
        const int v = atoi(e);
        return (v >= 0 && v <= 3) ? v : -1;
    }();
    if (forced >= 0) return forced;
    static const int64_t threshold = [] {
        const char* e = getenv("TDEQ_NT_THRESHOLD_MB");
        return (e && *e) ? ((int64_t)atoll(e) << 20) : kNtDefaultThreshold;
    }();
    return stream_bytes > threshold ? kNtPolicy : 0;
}

// ---- stage_combine --------------------------------------------------------------------------------
template <typename T, int NT>
int launch_combine(void* out, const void* y0, const void* const* k, const double* coef, double dt,
                   int64_t n, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    CombineArgs<T, NT> a;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    bool vec = aligned16(out) && aligned16(y0);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = (T)coef[j] * dtT;   // fl_T(fl_T(coef)*fl_T(dt)) — rk_common.py:79,201-205
        vec = vec && aligned16(k[j]);
    }
    a.n = n;
    constexpr int L = VecOf<T>::L;
    constexpr int U = 1;
    if (vec && ev_start && ev_stop) {
        // measurement hook (tdeq_stage_combine_timed): the events carry the dispatch's own begin / end timestamps
        const unsigned g = stream_grid(n / L, kBlock * U);
        hipExtLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 0>), dim3(g), dim3(kBlock), 0, s, ev_start, ev_stop, 0, a);
    } else if (vec) {
        const unsigned g = stream_grid(n / L, kBlock * U);
        switch (stream_policy((int64_t)(NT + 2) * n * (int64_t)sizeof(T))) {   // see stream_policy()
            case 1: hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 1>), dim3(g), dim3(kBlock), 0, s, a); break;
            case 2: hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 2>), dim3(g), dim3(kBlock), 0, s, a); break;
            case 3: hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 3>), dim3(g), dim3(kBlock), 0, s, a); break;
            default:
                if ((int64_t)g * kBlock * U >= n / L)
                    hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 0, true>), dim3(g), dim3(kBlock), 0, s, a);
                else hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, true, 0>), dim3(g), dim3(kBlock), 0, s, a);
        }
    } else {
        const unsigned g = stream_grid(n, kBlock * U);
        hipLaunchKernelGGL((stage_combine_kernel<T, NT, U, false>), dim3(g), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

template <typename T, int NT>
int launch_combine_fill(void* out, const void* y0, const void* const* k, const double* coef, double dt, int64_t n,
                        void* fill_dst, const double* fill_vals, int n_fill, hipStream_t s) {
    CombineArgs<T, NT> a;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    bool vec = aligned16(out) && aligned16(y0);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = (T)coef[j] * dtT;
        vec = vec && aligned16(k[j]);
    }
    a.n = n;
    SideFill<T> f;
    f.dst = static_cast<T*>(fill_dst);
    for (int i = 0; i < 16; ++i) f.v[i] = i < n_fill ? (T)fill_vals[i] : (T)0;
    f.n = n_fill;
    constexpr int L = VecOf<T>::L;
    if (vec) hipLaunchKernelGGL((stage_combine_fill_kernel<T, NT, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a, f);
    else hipLaunchKernelGGL((stage_combine_fill_kernel<T, NT, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a, f);
    return check_launch();
}

template <typename T, int NT>
int launch_combine_err(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                       const double* err_coef, double dt, int64_t n, hipStream_t s) {
    CombineErrArgs<T, NT> a;
    a.c.out = static_cast<T*>(out);
    a.c.y0 = static_cast<const T*>(y0);
    a.err_out = static_cast<T*>(err_out);
    bool vec = aligned16(out) && aligned16(y0) && aligned16(err_out);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.c.k[j] = static_cast<const T*>(k[j]);
        a.c.c[j] = (T)coef[j] * dtT;
        a.e[j] = (T)err_coef[j] * dtT;      // dt * c_error_j — rk_common.py:89
        vec = vec && aligned16(k[j]);
    }
    a.c.n = n;
    constexpr int L = VecOf<T>::L;
    if (vec) {
        const dim3 g(stream_grid(n / L, kBlock)), b(kBlock);
        // (r05, tried: non-temporal loads for the k streams of THIS launch only — their last use in a dopri5 step — with the
        //  default policy elsewhere: 0.3357 vs 0.3336 ms per cfg2 step, loads + stores 0.3323: noise; not kept)
        switch (stream_policy((int64_t)(NT + 3) * n * (int64_t)sizeof(T))) {
            case 1: hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, true, 1>), g, b, 0, s, a); break;
            case 2: hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, true, 2>), g, b, 0, s, a); break;
            case 3: hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, true, 3>), g, b, 0, s, a); break;
            default:
                if ((int64_t)g.x * kBlock >= n / L)
                    hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, true, 0, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, true, 0>), g, b, 0, s, a);
        }
    } else {
        hipLaunchKernelGGL((stage_combine_err_kernel<T, NT, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

template <typename T>
int dispatch_combine_err(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                         const double* err_coef, int nt, double dt, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_combine_err<T, N>(out, err_out, y0, k, coef, err_coef, dt, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T>
int dispatch_combine(void* out, const void* y0, const void* const* k, const double* coef, int nt,
                     double dt, int64_t n, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_combine<T, N>(out, y0, k, coef, dt, n, s, e0, e1);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- stage_combine_multi (carried partial sums) ---------------------------------------------------------
template <typename T, int NT>
int launch_combine_multi(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                         const void* const* k, double dt, int64_t n, hipStream_t s, hipEvent_t ev_start = nullptr,
                         hipEvent_t ev_stop = nullptr, const double* dt_dev = nullptr) {
    MultiArgs<T, NT> a;
    a.dt_dev = dt_dev;
    a.y0 = static_cast<const T*>(y0);
    a.acc_in = static_cast<const T*>(acc_in);
    bool vec = aligned16(y0) && aligned16(acc_in);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        vec = vec && aligned16(k[j]);
    }
    a.add_y0 = 0;
    for (int o = 0; o < kMaxMultiOut; ++o) {
        const bool live = o < n_out;
        a.out[o] = live ? static_cast<T*>(outs[o].out) : nullptr;
        a.mask[o] = live ? outs[o].mask : 0u;
        if (live && outs[o].add_y0) a.add_y0 |= 1u << o;
        for (int j = 0; j < NT; ++j)      // rk_common.py:79,201-205 (dt from device memory: the kernel multiplies)
            a.c[o][j] = !live ? (T)0 : (dt_dev ? (T)outs[o].coef[j] : (T)outs[o].coef[j] * dtT);
        if (live) vec = vec && aligned16(outs[o].out);
    }
    a.n_out = n_out;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    if (vec) {
        const dim3 g(stream_grid(n / L, kBlock)), b(kBlock);
        const bool timed = ev_start && ev_stop;      // measurement hook (tdeq_stage_combine_multi_timed)
        const int64_t streams = NT + 1 + (acc_in ? 1 : 0) + n_out;
#define TDEQ_MULTI(P)                                                                                                   \
    if (timed) hipExtLaunchKernelGGL((stage_combine_multi_kernel<T, NT, true, P>), g, b, 0, s, ev_start, ev_stop, 0, a); \
    else hipLaunchKernelGGL((stage_combine_multi_kernel<T, NT, true, P>), g, b, 0, s, a);
        const bool one_pass = (int64_t)g.x * kBlock >= n / L;      // exact cover (always, up to 2^24 16-byte elements)
#define TDEQ_MULTI_LAUNCH(...)                                                                                                \
    if (timed) hipExtLaunchKernelGGL((stage_combine_multi_kernel<T, NT, true, 0, __VA_ARGS__>), g, b, 0, s, ev_start, ev_stop, 0, a); \
    else hipLaunchKernelGGL((stage_combine_multi_kernel<T, NT, true, 0, __VA_ARGS__>), g, b, 0, s, a);
        // (output count, prefix continuation, dt from device memory — captured steps —, one pass)
#define TDEQ_MULTI_SHAPE(NO, AC)                                       \
    if (!one_pass) { TDEQ_MULTI_LAUNCH(NO, AC) }                       \
    else if (dt_dev) { TDEQ_MULTI_LAUNCH(NO, AC, true, true) }         \
    else { TDEQ_MULTI_LAUNCH(NO, AC, false, true) }
        switch (stream_policy(streams * n * (int64_t)sizeof(T))) {
            case 1: TDEQ_MULTI(1) break;
            case 2: TDEQ_MULTI(2) break;
            case 3: TDEQ_MULTI(3) break;
            default:
                // the in-cache regime (every dopri5 launch at cfg2): output count and prefix continuation as compile-time
                // constants for the one- and two-output launches (tdeq_kernels.hpp multi_elem)
                if (n_out == 1 && acc_in) { TDEQ_MULTI_SHAPE(1, 1) }
                else if (n_out == 1) { TDEQ_MULTI_SHAPE(1, 0) }
                else if (n_out == 2 && acc_in) { TDEQ_MULTI_SHAPE(2, 1) }
                else if (n_out == 2) { TDEQ_MULTI_SHAPE(2, 0) }
                else { TDEQ_MULTI(0) }
        }
#undef TDEQ_MULTI_SHAPE
#undef TDEQ_MULTI_LAUNCH
#undef TDEQ_MULTI
    } else {
        hipLaunchKernelGGL((stage_combine_multi_kernel<T, NT, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

template <typename T>
int dispatch_combine_multi(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                           const void* const* k, int nt, double dt, int64_t n, hipStream_t s, hipEvent_t e0 = nullptr,
                           hipEvent_t e1 = nullptr, const double* dt_dev = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_combine_multi<T, N>(outs, n_out, y0, acc_in, k, dt, n, s, e0, e1, dt_dev);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- segment table / workspace ---------------------------------------------------------------------
int fill_segtable(SegTable& st, const tdeq_segment* segs, const void* segs_dev, int n_seg,
                  int64_t chunk, int64_t n_chunks) {
    if (!segs || n_seg < 1 || n_seg > TDEQ_MAX_SEGMENTS) return TDEQ_EINVAL;
    if (chunk < TDEQ_CHUNK_QUANTUM || chunk % TDEQ_CHUNK_QUANTUM != 0 || n_chunks < 1) return TDEQ_EINVAL;
    if (n_chunks > 0x7fffffffLL) return TDEQ_EINVAL;
    if (n_seg > TDEQ_INLINE_SEGMENTS && !segs_dev) return TDEQ_EINVAL;
    for (int q = 0; q < TDEQ_INLINE_SEGMENTS; ++q) {
        if (q < n_seg && n_seg <= TDEQ_INLINE_SEGMENTS) st.inl[q] = segs[q];
        else st.inl[q] = tdeq_segment{0, 0, 0.0, 0.0};
    }
    if (n_seg > TDEQ_INLINE_SEGMENTS) st.inl[0] = segs[0];
    st.dev = static_cast<const tdeq_segment*>(segs_dev);
    st.n_seg = n_seg;
    st.chunk = chunk;
    st.n_chunks = n_chunks;
    return 0;
}

int launch_finalize(const SegTable& st, double* ws, int n_sum, double* out_sumsq, double* out_bad,
                    hipStream_t s) {
    FinalizeArgs f;
    for (int q = 0; q < 3; ++q) f.part[q] = ws + (int64_t)q * st.n_chunks;
    // the non-finite counters always live in the third array; remap so part[n_sum] is that array
    f.part[n_sum] = ws + 2 * st.n_chunks;
    f.st = st;
    f.n_sum = n_sum;
    f.out_sumsq = out_sumsq;
    f.out_bad = out_bad;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(st.n_seg, n_sum + 1), dim3(kBlock), 0, s, f);
    return check_launch();
}

// ---- error_norm ------------------------------------------------------------------------------------
template <typename T, int NT>
int launch_error(void* scaled, const void* y0, const void* y1, const void* const* k, const double* coef,
                 double dt, const SegTable& st, double* out_sumsq, double* out_bad, double* ws,
                 hipStream_t s) {
    ErrArgs<T, NT> a;
    a.scaled = static_cast<T*>(scaled);
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    bool vec = aligned16(y0) && aligned16(y1);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = (T)coef[j] * dtT;   // dt * c_error — rk_common.py:89
        vec = vec && aligned16(k[j]);
    }
    a.st = st;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (scaled) {
        vec = vec && aligned16(scaled);
        if (vec) hipLaunchKernelGGL((error_norm_kernel<T, NT, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((error_norm_kernel<T, NT, false, true>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((error_norm_kernel<T, NT, true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((error_norm_kernel<T, NT, false, false>), g, b, 0, s, a);
    }
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
}

struct CtrlBundle {
    const tdeq_step_ctrl* ctrl;
    double* out_ctrl;
    double* ctrl_dev;
    void* next_times;
    int state_in_dev;     // hipGraph mode: (t0, dt) of the trial step and the kernels' dt live in ctrl_dev
};
int launch_finalize_ctrl(const SegTable& st, double* ws, double* out_sumsq, double* out_bad, const CtrlBundle& cb,
                         int tkind, hipStream_t s, int ratio_kind);

template <typename T, int NT>
int launch_error_vec(const void* partial, const void* y0, const void* y1, const void* const* k, const double* coef, double dt,
                     const double* rtol_v, double rtol_s, const double* atol_v, double atol_s, const SegTable& st,
                     double* out_sumsq, double* out_bad, double* ws, hipStream_t s, const CtrlBundle* cb = nullptr) {
    ErrVecArgs<T, NT> a;
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    a.partial = static_cast<const T*>(partial);
    a.k[0] = nullptr;
    a.c[0] = (T)0;
    const bool dev_dt = cb && cb->state_in_dev;      // hipGraph mode: dt = ctrl_dev[1] on the device
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = dev_dt ? (T)coef[j] : (T)coef[j] * dtT;
    }
    a.dt_dev = dev_dt ? cb->ctrl_dev + 1 : nullptr;
    a.rtol_v = rtol_v;
    a.atol_v = atol_v;
    a.rtol_s = rtol_s;
    a.atol_s = atol_s;
    a.st = st;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (dev_dt) {
        if (partial) hipLaunchKernelGGL((error_norm_vec_kernel<T, NT, true, true>), g, b, 0, s, a);
        else if constexpr (NT > 0) hipLaunchKernelGGL((error_norm_vec_kernel<T, NT, false, true>), g, b, 0, s, a);
        else return TDEQ_EINVAL;
    } else if (partial) hipLaunchKernelGGL((error_norm_vec_kernel<T, NT, true>), g, b, 0, s, a);
    else if constexpr (NT > 0) hipLaunchKernelGGL((error_norm_vec_kernel<T, NT, false>), g, b, 0, s, a);
    else return TDEQ_EINVAL;
    const int e = check_launch();
    if (e) return e;
    // with a controller bundle: the ratio in fp64 (the promoted type), the step size and the stage times in T
    if (cb) return launch_finalize_ctrl(st, ws, out_sumsq, out_bad, *cb, sizeof(T) == 4 ? 1 : 0, s, 0);
    return launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
}

template <typename T>
int dispatch_error_vec(const void* partial, const void* y0, const void* y1, const void* const* k, const double* coef, int nt,
                       double dt, const double* rtol_v, double rtol_s, const double* atol_v, double atol_s, const SegTable& st,
                       double* out_sumsq, double* out_bad, double* ws, hipStream_t s, const CtrlBundle* cb = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_error_vec<T, N>(partial, y0, y1, k, coef, dt, rtol_v, rtol_s, atol_v, atol_s, st, out_sumsq, out_bad, ws, s, cb);
        TDEQ_CASE(0) TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T>
int dispatch_error(void* scaled, const void* y0, const void* y1, const void* const* k, const double* coef,
                   int nt, double dt, const SegTable& st, double* out_sumsq, double* out_bad, double* ws,
                   hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_error<T, N>(scaled, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// Optional controller bundle of the *_ctrl entry point (null ctrl => plain finalize).

// tkind: the state's real type — 0 fp64, 1 fp32 (callers pass `sizeof(T) == 4`), 2 bfloat16, 3 float16
int launch_finalize_ctrl(const SegTable& st, double* ws, double* out_sumsq, double* out_bad, const CtrlBundle& cb,
                         int tkind, hipStream_t s, int ratio_kind) {
    CtrlArgs a;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    a.st = st;
    a.c = *cb.ctrl;
    a.is_f32 = tkind;
    a.ratio_kind = ratio_kind < 0 ? tkind : ratio_kind;
    a.out_sumsq = out_sumsq;
    a.out_bad = out_bad;
    a.out_ctrl = cb.out_ctrl;
    a.ctrl_dev = cb.ctrl_dev;
    a.next_times = cb.next_times;
    a.state_in_dev = cb.state_in_dev;
    a.presummed = 0;
    a.in_sumsq = nullptr;
    a.in_bad = nullptr;
    if (st.n_seg > kCtrlInlineSegments) {
        // more segments than the one-workgroup form handles well: the per-segment sums by the parallel finalize (one
        // workgroup per segment; segment table inline, or from device memory beyond TDEQ_INLINE_SEGMENTS), then the
        // one-workgroup controller on those sums
        const int e = launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
        if (e) return e;
        a.presummed = 1;
    }
    // one instantiation per segment-count bucket (the per-lane accumulators are a static array of 2·NS doubles)
    if (a.presummed || st.n_seg == 1) hipLaunchKernelGGL(norm_finalize_ctrl_kernel<1>, dim3(1), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL(norm_finalize_ctrl_kernel<kCtrlInlineSegments>, dim3(1), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T, int NT>
int launch_error_partial(const void* partial, const void* y0, const void* y1, const void* const* k,
                         const double* coef, double dt, const SegTable& st, double* out_sumsq, double* out_bad,
                         double* ws, const CtrlBundle* cb, hipStream_t s, void* copy_last = nullptr) {
    ErrPartialArgs<T, NT> a;
    a.copy_out = static_cast<T*>(copy_last);
    if (copy_last && NT == 0) return TDEQ_EINVAL;        // nothing to copy: the error row ended with the last combine
    a.partial = static_cast<const T*>(partial);
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    bool vec = aligned16(partial) && aligned16(y0) && aligned16(y1);
    const bool dev_dt = cb && cb->state_in_dev;      // hipGraph mode: dt = ctrl_dev[1] on the device
    const T dtT = (T)dt;
    a.k[0] = nullptr;
    a.c[0] = (T)0;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = dev_dt ? (T)coef[j] : (T)coef[j] * dtT;
        vec = vec && aligned16(k[j]);
    }
    a.dt_dev = dev_dt ? cb->ctrl_dev + 1 : nullptr;
    a.st = st;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    // the common shape — one segment whose chunk_start is 0, dt folded by the host — has its own lean instantiation
    const bool single = st.n_seg == 1 && st.inl[0].chunk_start == 0, lean = single && !dev_dt;
    if (copy_last) {
        // (captured steps only: states of at most 2^22 elements, always the default cache policy)
        if constexpr (NT > 0) {
            const bool v = vec && aligned16(copy_last);
            if (v && single) hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 0, true, true, true>), g, b, 0, s, a);
            else if (v) hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 0, false, true, true>), g, b, 0, s, a);
            else hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, false, 0, false, true, true>), g, b, 0, s, a);
        }
    } else if (vec && (stream_policy((int64_t)(NT + 3) * st.n_chunks * st.chunk * (int64_t)sizeof(T)) & 1))
        hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 1>), g, b, 0, s, a);
    else if (vec && lean) hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 0, true, false>), g, b, 0, s, a);
    else if (vec && single) hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 0, true, true>), g, b, 0, s, a);
    else if (vec) hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, true, 0>), g, b, 0, s, a);
    else hipLaunchKernelGGL((error_norm_partial_kernel<T, NT, false>), g, b, 0, s, a);
    const int e = check_launch();
    if (e) return e;
    if (cb) return launch_finalize_ctrl(st, ws, out_sumsq, out_bad, *cb, sizeof(T) == 4 ? 1 : 0, s, -1);
    return launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
}

template <typename T>
int dispatch_error_partial(const void* partial, const void* y0, const void* y1, const void* const* k,
                           const double* coef, int nt, double dt, const SegTable& st, double* out_sumsq,
                           double* out_bad, double* ws, const CtrlBundle* cb, hipStream_t s, void* copy_last = nullptr) {
    switch (nt) {
        case 0: return launch_error_partial<T, 0>(partial, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s, copy_last);
        case 1: return launch_error_partial<T, 1>(partial, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s, copy_last);
        case 2: return launch_error_partial<T, 2>(partial, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s, copy_last);
    }
    return TDEQ_EINVAL;
}

template <typename T, int NT>
int launch_combine_devdt(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                         const double* err_coef, const double* dt_dev, int64_t n, hipStream_t s) {
    CombineDevTArgs<T, NT> a;
    a.c.out = static_cast<T*>(out);
    a.c.y0 = static_cast<const T*>(y0);
    a.err_out = static_cast<T*>(err_out);
    bool vec = aligned16(out) && aligned16(y0) && aligned16(err_out);
    for (int j = 0; j < NT; ++j) {
        a.c.k[j] = static_cast<const T*>(k[j]);
        a.c.c[j] = (T)coef[j];
        a.e[j] = err_out ? (T)err_coef[j] : (T)0;
        vec = vec && aligned16(k[j]);
    }
    a.c.n = n;
    a.dt_dev = dt_dev;
    constexpr int L = VecOf<T>::L;
    const dim3 g(vec ? stream_grid(n / L, kBlock) : stream_grid(n, kBlock)), b(kBlock);
    if (err_out) {
        if (vec) hipLaunchKernelGGL((combine_devdt_kernel<T, NT, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((combine_devdt_kernel<T, NT, false, true>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((combine_devdt_kernel<T, NT, true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((combine_devdt_kernel<T, NT, false, false>), g, b, 0, s, a);
    }
    return check_launch();
}

// States below this size are launch-latency-bound: the run-time-term-count kernel (one instantiation, smallest code)
// is as fast there; above it the unrolled 16-byte-per-lane form wins (profiles/r02_shard_regime.json).
constexpr int64_t kDevCombineTunedFrom = 1 << 17;

template <typename T>
int launch_combine_dev(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                       const double* err_coef, int nt, const double* dt_dev, int64_t n, hipStream_t s) {
    if (n >= kDevCombineTunedFrom) {
        switch (nt) {
#define TDEQ_CASE(N) case N: return launch_combine_devdt<T, N>(out, err_out, y0, k, coef, err_coef, dt_dev, n, s);
            TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
            TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
        }
        return TDEQ_EINVAL;
    }
    CombineDevArgs<T> a;
    a.out = static_cast<T*>(out);
    a.err_out = static_cast<T*>(err_out);
    a.y0 = static_cast<const T*>(y0);
    for (int j = 0; j < TDEQ_MAX_TERMS; ++j) {
        a.k[j] = j < nt ? static_cast<const T*>(k[j]) : nullptr;
        a.c[j] = j < nt ? (T)coef[j] : (T)0;
        a.e[j] = (j < nt && err_coef) ? (T)err_coef[j] : (T)0;
    }
    a.nt = nt;
    a.dt_dev = dt_dev;
    a.n = n;
    hipLaunchKernelGGL((combine_dev_kernel<T>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T>
int launch_commit(void* y_prev, void* f_prev, void* y_cur, void* f_cur, const void* y1, const void* f1,
                  const double* ctrl_dev, int64_t n, hipStream_t s) {
    CommitArgs<T> a;
    a.y_prev = static_cast<T*>(y_prev);
    a.f_prev = static_cast<T*>(f_prev);
    a.y_cur = static_cast<T*>(y_cur);
    a.f_cur = static_cast<T*>(f_cur);
    a.y1 = static_cast<const T*>(y1);
    a.f1 = static_cast<const T*>(f1);
    a.ctrl_dev = ctrl_dev;
    a.n = n;
    const bool vec = aligned16(y_prev) && aligned16(f_prev) && aligned16(y_cur) && aligned16(f_cur) && aligned16(y1) &&
                     aligned16(f1);
    constexpr int L = VecOf<T>::L;
    if (vec) hipLaunchKernelGGL((step_commit_kernel<T, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((step_commit_kernel<T, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T>
int launch_combine_sel(void* out, const void* y_acc, const void* f_acc, const void* y_rej, const void* f_rej,
                       double coef, const double* ctrl_dev, int64_t n, hipStream_t s) {
    SelArgs<T> a;
    a.out = static_cast<T*>(out);
    a.y_acc = static_cast<const T*>(y_acc);
    a.f_acc = static_cast<const T*>(f_acc);
    a.y_rej = static_cast<const T*>(y_rej);
    a.f_rej = static_cast<const T*>(f_rej);
    a.coef = (T)coef;
    a.ctrl_dev = ctrl_dev;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    const bool vec = aligned16(out) && aligned16(y_acc) && aligned16(f_acc) && aligned16(y_rej) && aligned16(f_rej);
    if (vec) {
        const dim3 g(stream_grid(n / L, kBlock)), b(kBlock);
        // (reads 2 of the 4 inputs; the k tensors of the step just finished are the cache's other tenants)
        switch (stream_policy((int64_t)3 * n * (int64_t)sizeof(T))) {
            case 1: hipLaunchKernelGGL((stage_combine_sel_kernel<T, true, 1>), g, b, 0, s, a); break;
            case 2: hipLaunchKernelGGL((stage_combine_sel_kernel<T, true, 2>), g, b, 0, s, a); break;
            case 3: hipLaunchKernelGGL((stage_combine_sel_kernel<T, true, 3>), g, b, 0, s, a); break;
            default: hipLaunchKernelGGL((stage_combine_sel_kernel<T, true, 0>), g, b, 0, s, a);
        }
    } else {
        hipLaunchKernelGGL((stage_combine_sel_kernel<T, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

// ---- init norms ------------------------------------------------------------------------------------
template <typename T>
int launch_init(int mode, const void* a_, const void* b_, const void* y_, const SegTable& st,
                double* out_sumsq, double* out_bad, double* ws, hipStream_t s) {
    InitArgs<T> a;
    a.a = static_cast<const T*>(a_);
    a.b = static_cast<const T*>(b_);
    a.y = static_cast<const T*>(y_);
    a.st = st;
    a.part0 = ws;
    a.part1 = ws + st.n_chunks;
    a.part_bad = ws + 2 * st.n_chunks;
    const bool vec = aligned16(a_) && aligned16(b_) && aligned16(y_);
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (mode == 0) {
        if (vec) hipLaunchKernelGGL((init_norms_kernel<T, 0, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((init_norms_kernel<T, 0, false>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((init_norms_kernel<T, 1, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((init_norms_kernel<T, 1, false>), g, b, 0, s, a);
    }
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, mode == 0 ? 2 : 1, out_sumsq, out_bad, s);
}

template <typename T>
int launch_init_vec(int mode, const void* a_, const void* b_, const void* y_, const double* rtol_v, double rtol_s,
                    const double* atol_v, double atol_s, const SegTable& st, double* out_sumsq, double* out_bad, double* ws,
                    hipStream_t s) {
    InitVecArgs<T> a;
    a.a = static_cast<const T*>(a_);
    a.b = static_cast<const T*>(b_);
    a.y = static_cast<const T*>(y_);
    a.rtol_v = rtol_v;
    a.atol_v = atol_v;
    a.rtol_s = rtol_s;
    a.atol_s = atol_s;
    a.st = st;
    a.part0 = ws;
    a.part1 = ws + st.n_chunks;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (mode == 0) hipLaunchKernelGGL((init_norms_vec_kernel<T, 0>), g, b, 0, s, a);
    else hipLaunchKernelGGL((init_norms_vec_kernel<T, 1>), g, b, 0, s, a);
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, mode == 0 ? 2 : 1, out_sumsq, out_bad, s);
}

// ---- complex states: the norm launches (tdeq_kernels_complex.hpp); T = the real type -------------------------------
template <typename T, int NT, bool PARTIAL>
int launch_cplx_error(const void* partial, void* scaled, const void* y0, const void* y1, const void* const* k,
                      const double* coef, double dt, const SegTable& st, double* out_sumsq, double* out_bad, double* ws,
                      const CtrlBundle* cb, hipStream_t s) {
    CplxErrArgs<T, NT> a;
    a.partial = static_cast<const T*>(partial);
    a.scaled = static_cast<T*>(scaled);
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    bool vec = aligned16(y0) && aligned16(y1) && (!PARTIAL || aligned16(partial)) && (!scaled || aligned16(scaled));
    const bool dev_dt = cb && cb->state_in_dev;
    const T dtT = (T)dt;
    a.k[0] = nullptr;
    a.c[0] = (T)0;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = dev_dt ? (T)coef[j] : (T)coef[j] * dtT;
        vec = vec && aligned16(k[j]);
    }
    a.dt_dev = dev_dt ? cb->ctrl_dev + 1 : nullptr;
    a.st = st;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (scaled) {
        if (vec) hipLaunchKernelGGL((cplx_error_norm_kernel<T, NT, true, PARTIAL, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((cplx_error_norm_kernel<T, NT, false, PARTIAL, true>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((cplx_error_norm_kernel<T, NT, true, PARTIAL, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((cplx_error_norm_kernel<T, NT, false, PARTIAL, false>), g, b, 0, s, a);
    }
    const int e = check_launch();
    if (e) return e;
    if (cb) return launch_finalize_ctrl(st, ws, out_sumsq, out_bad, *cb, sizeof(T) == 4 ? 1 : 0, s, -1);
    return launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
}

template <typename T>
int dispatch_cplx_error(void* scaled, const void* y0, const void* y1, const void* const* k, const double* coef, int nt,
                        double dt, const SegTable& st, double* out_sumsq, double* out_bad, double* ws, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_cplx_error<T, N, false>(nullptr, scaled, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, nullptr, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T>
int dispatch_cplx_error_partial(const void* partial, const void* y0, const void* y1, const void* const* k,
                                const double* coef, int nt, double dt, const SegTable& st, double* out_sumsq,
                                double* out_bad, double* ws, const CtrlBundle* cb, hipStream_t s) {
    switch (nt) {
        case 0: return launch_cplx_error<T, 0, true>(partial, nullptr, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s);
        case 1: return launch_cplx_error<T, 1, true>(partial, nullptr, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s);
        case 2: return launch_cplx_error<T, 2, true>(partial, nullptr, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, cb, s);
    }
    return TDEQ_EINVAL;
}

// mode 0 / 1; out0 given: the quotients are stored (tdeq_init_scaled), else their sums (tdeq_init_norms)
template <typename T>
int launch_cplx_init(int mode, const void* a_, const void* b_, const void* y_, const SegTable& st, double* out_sumsq,
                     double* out_bad, double* ws, void* out0, void* out1, hipStream_t s) {
    CplxInitArgs<T> a;
    a.a = static_cast<const T*>(a_);
    a.b = static_cast<const T*>(b_);
    a.y = static_cast<const T*>(y_);
    a.st = st;
    a.part0 = ws;
    a.part1 = ws ? ws + st.n_chunks : nullptr;
    a.part_bad = ws ? ws + 2 * st.n_chunks : nullptr;
    a.out0 = static_cast<T*>(out0);
    a.out1 = static_cast<T*>(out1);
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (out0) {
        if (mode == 0) hipLaunchKernelGGL((cplx_init_norms_kernel<T, 0, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((cplx_init_norms_kernel<T, 1, true>), g, b, 0, s, a);
        return check_launch();
    }
    if (mode == 0) hipLaunchKernelGGL((cplx_init_norms_kernel<T, 0, false>), g, b, 0, s, a);
    else hipLaunchKernelGGL((cplx_init_norms_kernel<T, 1, false>), g, b, 0, s, a);
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, mode == 0 ? 2 : 1, out_sumsq, out_bad, s);
}

// ---- dense output ----------------------------------------------------------------------------------
template <typename T, int NT, bool FIT_ONLY>
int launch_dense(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                 const void* const* k, const double* coef, double dt, double x, int64_t n, hipStream_t s) {
    DenseArgs<T, NT> a;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    a.f0 = static_cast<const T*>(f0);
    a.f1 = static_cast<const T*>(f1);
    bool vec = aligned16(out) && aligned16(y0) && aligned16(y1) && aligned16(f0) && aligned16(f1);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.c[j] = (T)coef[j] * dtT;   // dt * mid — rk_common.py:365-366
        vec = vec && aligned16(k[j]);
    }
    a.dt = dtT;
    a.x = (T)x;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    if (FIT_ONLY) vec = vec && (n % L == 0);   // the 5 coefficient planes must stay 16-byte aligned
    if (vec) {
        const unsigned g = stream_grid(n / L, kBlock);
        hipLaunchKernelGGL((dense_kernel<T, NT, FIT_ONLY, true>), dim3(g), dim3(kBlock), 0, s, a);
    } else {
        const unsigned g = stream_grid(n, kBlock);
        hipLaunchKernelGGL((dense_kernel<T, NT, FIT_ONLY, false>), dim3(g), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

template <typename T, bool FIT_ONLY>
int dispatch_dense(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                   const void* const* k, const double* coef, int nt, double dt, double x, int64_t n,
                   hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_dense<T, N, FIT_ONLY>(out, y0, y1, f0, f1, k, coef, dt, x, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T, int NT>
int launch_dense_multi(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0, const void* f1,
                       const void* const* k, const double* coef, double dt, const double* x, int n_x, int64_t n,
                       hipStream_t s) {
    DenseMultiArgs<T, NT> a;
    a.d.out = static_cast<T*>(out);
    a.d.y0 = static_cast<const T*>(y0);
    a.d.y1 = static_cast<const T*>(y1);
    a.d.f0 = static_cast<const T*>(f0);
    a.d.f1 = static_cast<const T*>(f1);
    bool vec = aligned16(out) && aligned16(y0) && aligned16(y1) && aligned16(f0) && aligned16(f1);
    const T dtT = (T)dt;
    for (int j = 0; j < NT; ++j) {
        a.d.k[j] = static_cast<const T*>(k[j]);
        a.d.c[j] = (T)coef[j] * dtT;
        vec = vec && aligned16(k[j]);
    }
    a.d.dt = dtT;
    a.d.x = (T)0;
    a.d.n = n;
    for (int r = 0; r < kMaxDenseOutputs; ++r) a.xs[r] = r < n_x ? (T)x[r] : (T)0;
    a.m = n_x;
    a.out_stride = out_stride;
    vec = vec && (n_x == 1 || (out_stride * (int64_t)sizeof(T)) % 16 == 0);   // every output row 16-byte aligned
    constexpr int L = VecOf<T>::L;
    if (vec) hipLaunchKernelGGL((dense_multi_kernel<T, NT, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((dense_multi_kernel<T, NT, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T>
int dispatch_dense_multi(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0,
                         const void* f1, const void* const* k, const double* coef, int nt, double dt,
                         const double* x, int n_x, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_dense_multi<T, N>(out, out_stride, y0, y1, f0, f1, k, coef, dt, x, n_x, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- rk4 / lerp ------------------------------------------------------------------------------------
template <typename T, int STAGE>
int launch_rk4(void* out, const void* y0, const void* k1, const void* k2, const void* k3,
               const void* k4, double dt, int64_t n, hipStream_t s, const double* dt_dev = nullptr) {
    Rk4Args<T> a;
    a.dt_dev = dt_dev;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    a.k1 = static_cast<const T*>(k1);
    a.k2 = static_cast<const T*>(k2);
    a.k3 = static_cast<const T*>(k3);
    a.k4 = static_cast<const T*>(k4);
    a.dt = (T)dt;
    a.third = (T)(1.0 / 3.0);   // `_one_third` rounded to T — rk_common.py:94,114-115
    a.n = n;
    bool vec = aligned16(out) && aligned16(y0) && aligned16(k1);
    if (STAGE >= 2) vec = vec && aligned16(k2);
    if (STAGE >= 3) vec = vec && aligned16(k3);
    if (STAGE >= 4) vec = vec && aligned16(k4);
    constexpr int L = VecOf<T>::L;
    if (vec) {
        hipLaunchKernelGGL((rk4_kernel<T, STAGE, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    } else {
        hipLaunchKernelGGL((rk4_kernel<T, STAGE, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    }
    return check_launch();
}

template <typename T>
int dispatch_rk4(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3,
                 const void* k4, double dt, int64_t n, hipStream_t s, const double* dt_dev = nullptr) {
    switch (stage) {
        case 1: return k1 ? launch_rk4<T, 1>(out, y0, k1, k2, k3, k4, dt, n, s, dt_dev) : TDEQ_EINVAL;
        case 2: return (k1 && k2) ? launch_rk4<T, 2>(out, y0, k1, k2, k3, k4, dt, n, s, dt_dev) : TDEQ_EINVAL;
        case 3: return (k1 && k2 && k3) ? launch_rk4<T, 3>(out, y0, k1, k2, k3, k4, dt, n, s, dt_dev) : TDEQ_EINVAL;
        case 4: return (k1 && k2 && k3 && k4) ? launch_rk4<T, 4>(out, y0, k1, k2, k3, k4, dt, n, s, dt_dev) : TDEQ_EINVAL;
    }
    return TDEQ_EINVAL;
}

template <typename T>
int launch_grid_commit(void* solution, int64_t row_stride, void* y_cur, const void* y_new, const int64_t* counter,
                       int64_t n, hipStream_t s) {
    GridCommitArgs<T> a;
    a.solution = static_cast<T*>(solution);
    a.row_stride = row_stride;
    a.y_cur = static_cast<T*>(y_cur);
    a.y_new = static_cast<const T*>(y_new);
    a.counter = counter;
    a.n = n;
    hipLaunchKernelGGL((grid_commit_kernel<T>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T>
int launch_lerp(void* out, const void* y0, const void* y1, double slope, int64_t n, hipStream_t s) {
    LerpArgs<T> a;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    a.y1 = static_cast<const T*>(y1);
    a.slope = (T)slope;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    if (aligned16(out) && aligned16(y0) && aligned16(y1))
        hipLaunchKernelGGL((lerp_kernel<T, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    else
        hipLaunchKernelGGL((lerp_kernel<T, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

// ---- low-order fixed-grid stages / weighted sums -----------------------------------------------------
template <typename T, int NT, int MODE>
int launch_fixed(void* out, const void* y0, const void* const* k, const double* w, double dt, int64_t n,
                 hipStream_t s, const double* dt_dev = nullptr) {
    FixedArgs<T, NT> a;
    a.dt_dev = dt_dev;
    a.out = static_cast<T*>(out);
    a.y0 = static_cast<const T*>(y0);
    bool vec = aligned16(out) && (MODE == 2 || aligned16(y0));
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const T*>(k[j]);
        a.w[j] = (T)w[j];   // Python-float weights meet a T tensor: rounded to T (rk_common.py:139-157)
        vec = vec && aligned16(k[j]);
    }
    a.dt = (T)dt;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    if (vec) hipLaunchKernelGGL((fixed_stage_kernel<T, NT, MODE, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((fixed_stage_kernel<T, NT, MODE, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T, int MODE>
int dispatch_fixed(void* out, const void* y0, const void* const* k, const double* w, int nt, double dt,
                   int64_t n, hipStream_t s, const double* dt_dev = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_fixed<T, N, MODE>(out, y0, k, w, dt, n, s, dt_dev);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4)
        case 5: if (MODE == 2) return launch_fixed<T, 5, MODE>(out, y0, k, w, dt, n, s); break;
        case 6: if (MODE == 2) return launch_fixed<T, 6, MODE>(out, y0, k, w, dt, n, s); break;
        case 7: if (MODE == 2) return launch_fixed<T, 7, MODE>(out, y0, k, w, dt, n, s); break;
        case 8: if (MODE == 2) return launch_fixed<T, 8, MODE>(out, y0, k, w, dt, n, s); break;
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- backward helpers: scale_many / multi_dot ------------------------------------------------------------
template <typename T, int NT>
int launch_scale(void* const* outs, const void* g, const double* w, int64_t n, hipStream_t s) {
    ScaleArgs<T, NT> a;
    a.g = static_cast<const T*>(g);
    bool vec = aligned16(g);
    for (int j = 0; j < NT; ++j) {
        a.out[j] = static_cast<T*>(outs[j]);
        a.w[j] = (T)w[j];
        vec = vec && aligned16(outs[j]);
    }
    a.n = n;
    constexpr int L = VecOf<T>::L;
    if (vec) hipLaunchKernelGGL((scale_many_kernel<T, NT, true>), dim3(stream_grid(n / L, kBlock)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((scale_many_kernel<T, NT, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}

template <typename T>
int dispatch_scale(void* const* outs, const void* g, const double* w, int nt, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_scale<T, N>(outs, g, w, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

constexpr int64_t kDotChunk = 4096;

template <typename T, int NT>
int launch_dots(const void* g, const void* const* x, int64_t n, double* out, double* ws, hipStream_t s) {
    DotArgs<T, NT> a;
    a.g = static_cast<const T*>(g);
    bool vec = aligned16(g);
    for (int j = 0; j < NT; ++j) {
        a.x[j] = static_cast<const T*>(x[j]);
        vec = vec && aligned16(x[j]);
    }
    a.n = n;
    a.chunk = kDotChunk;
    a.n_chunks = (n + kDotChunk - 1) / kDotChunk;
    if (a.n_chunks < 1) a.n_chunks = 1;
    a.part = ws;
    const dim3 grid((unsigned)a.n_chunks), blk(kBlock);
    if (vec) hipLaunchKernelGGL((multi_dot_kernel<T, NT, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((multi_dot_kernel<T, NT, false>), grid, blk, 0, s, a);
    const int e = check_launch();
    if (e) return e;
    DotFinalizeArgs f;
    f.part = ws;
    f.n_chunks = a.n_chunks;
    f.out = out;
    hipLaunchKernelGGL(dot_finalize_kernel, dim3(NT), dim3(kBlock), 0, s, f);
    return check_launch();
}

template <typename T>
int dispatch_dots(const void* g, const void* const* x, int nt, int64_t n, double* out, double* ws, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_dots<T, N>(g, x, n, out, ws, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T>
int launch_pack(void* out, const void* const* src, const int64_t* chunk_start, const int64_t* numel,
                const double* scale, int n_seg, int64_t chunk, int64_t n_chunks, hipStream_t s) {
    PackArgs<T> a;
    a.out = static_cast<T*>(out);
    for (int q = 0; q < TDEQ_INLINE_SEGMENTS; ++q) {
        if (q < n_seg) a.seg[q] = PackSeg{src[q], chunk_start[q], numel[q], scale[q], aligned16(src[q]) ? 1 : 0};
        else a.seg[q] = PackSeg{nullptr, 0, 0, 0.0, 0};
    }
    a.n_seg = n_seg;
    a.chunk = chunk;
    hipLaunchKernelGGL((pack_segments_kernel<T>), dim3((unsigned)n_chunks), dim3(kBlock), 0, s, a);
    return check_launch();
}

// ---- Adams–Bashforth(–Moulton) -----------------------------------------------------------------------
template <typename T, int NT>
int launch_adams_predict(void* y_out, void* dy_out, void* delta_out, const void* y0, const void* const* f,
                         const double* cb, const double* cm, double dt, int64_t n, hipStream_t s) {
    AdamsPredictArgs<T, NT> a;
    const bool implicit = dy_out != nullptr;
    a.y_out = static_cast<T*>(y_out);
    a.dy_out = static_cast<T*>(dy_out);
    a.delta_out = static_cast<T*>(delta_out);
    a.y0 = static_cast<const T*>(y0);
    bool vec = aligned16(y_out) && aligned16(y0) && aligned16(dy_out) && aligned16(delta_out);
    for (int j = 0; j < NT; ++j) {
        a.f[j] = static_cast<const T*>(f[j]);
        a.cb[j] = (T)cb[j];
        a.cm[j] = implicit ? (T)cm[j] : (T)0;
        vec = vec && aligned16(f[j]);
    }
    a.dt = (T)dt;
    a.n = n;
    constexpr int L = VecOf<T>::L;
    const dim3 g(vec ? stream_grid(n / L, kBlock) : stream_grid(n, kBlock)), b(kBlock);
    if (implicit) {
        if (vec) hipLaunchKernelGGL((adams_predict_kernel<T, NT, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((adams_predict_kernel<T, NT, true, false>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((adams_predict_kernel<T, NT, false, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((adams_predict_kernel<T, NT, false, false>), g, b, 0, s, a);
    }
    return check_launch();
}

template <typename T>
int dispatch_adams_predict(void* y_out, void* dy_out, void* delta_out, const void* y0, const void* const* f,
                           const double* cb, const double* cm, int nt, double dt, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return launch_adams_predict<T, N>(y_out, dy_out, delta_out, y0, f, cb, cm, dt, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename T>
int launch_adams_correct(void* y_out, void* dy_out, const void* f, const void* delta, const void* dy_old,
                         const void* y0, double c, bool compute, const SegTable& st, int64_t n, double* out_count,
                         double* out_bad, double* ws, hipStream_t s) {
    AdamsCorrectArgs<T> a;
    a.y_out = static_cast<T*>(y_out);
    a.dy_out = static_cast<T*>(dy_out);
    a.f = static_cast<const T*>(f);
    a.delta = static_cast<const T*>(delta);
    a.dy_old = static_cast<const T*>(dy_old);
    a.y0 = static_cast<const T*>(y0);
    a.c = (T)c;
    a.n = n;
    a.st = st;
    a.part_count = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const bool vec = aligned16(y_out) && aligned16(dy_out) && aligned16(f) && aligned16(delta) && aligned16(dy_old) &&
                     aligned16(y0);
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (compute) {
        if (vec) hipLaunchKernelGGL((adams_correct_kernel<T, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((adams_correct_kernel<T, true, false>), g, b, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((adams_correct_kernel<T, false, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((adams_correct_kernel<T, false, false>), g, b, 0, s, a);
    }
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, 1, out_count, out_bad, s);
}

inline bool bad_dtype(int dtype) { return dtype != TDEQ_F32 && dtype != TDEQ_F64; }
// bfloat16 / float16 states: the entry points of the host-driven step (tdeq_kernels_lp.hpp; include/tdeq_hip.h lists them)
inline bool lp_dtype(int dtype) { return dtype == TDEQ_BF16 || dtype == TDEQ_F16; }

#include "tdeq_abi_lp.hpp"
// the norm entry points also take interleaved complex states (TDEQ_C64 / TDEQ_C128: tdeq_kernels_complex.hpp)
inline bool bad_norm_dtype(int dtype) { return bad_dtype(dtype) && dtype != TDEQ_C64 && dtype != TDEQ_C128; }

}  // namespace

extern "C" {

int tdeq_abi_version(void) { return TDEQ_ABI_VERSION; }

size_t tdeq_workspace_bytes(int64_t n_chunks) {
    if (n_chunks < 1) n_chunks = 1;
    return (size_t)n_chunks * 3 * sizeof(double);
}

int tdeq_stage_combine(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                       double dt, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !k || !coef || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_combine<lp::BF16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, dt, n, nullptr, nullptr, 0, s);
    if (dtype == TDEQ_F16) return lp_dispatch_combine<lp::F16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, dt, n, nullptr, nullptr, 0, s);
    return dtype == TDEQ_F32 ? dispatch_combine<float>(out, y0, k, coef, n_terms, dt, n, s)
                             : dispatch_combine<double>(out, y0, k, coef, n_terms, dt, n, s);
}

int tdeq_stage_combine_timed(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                             double dt, int64_t n, int dtype, void* stream, void* start_event, void* stop_event) {
    if (!out || !y0 || !k || !coef || n < 1 || bad_dtype(dtype) || !start_event || !stop_event) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = static_cast<hipEvent_t>(start_event), e1 = static_cast<hipEvent_t>(stop_event);
    return dtype == TDEQ_F32 ? dispatch_combine<float>(out, y0, k, coef, n_terms, dt, n, s, e0, e1)
                             : dispatch_combine<double>(out, y0, k, coef, n_terms, dt, n, s, e0, e1);
}

int tdeq_stage_combine_fill(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                            double dt, int64_t n, int dtype, void* fill_dst, const double* fill_vals, int n_fill,
                            void* stream) {
    if (!out || !y0 || !k || !coef || n < 1 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > 2 || !k[0] || (n_terms == 2 && !k[1])) return TDEQ_EINVAL;
    if (!fill_dst || !fill_vals || n_fill < 1 || n_fill > 16) return TDEQ_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_combine<lp::BF16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, dt, n, fill_dst, fill_vals, n_fill, s);
    if (dtype == TDEQ_F16) return lp_dispatch_combine<lp::F16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, dt, n, fill_dst, fill_vals, n_fill, s);
    if (dtype == TDEQ_F32)
        return n_terms == 1 ? launch_combine_fill<float, 1>(out, y0, k, coef, dt, n, fill_dst, fill_vals, n_fill, s)
                            : launch_combine_fill<float, 2>(out, y0, k, coef, dt, n, fill_dst, fill_vals, n_fill, s);
    return n_terms == 1 ? launch_combine_fill<double, 1>(out, y0, k, coef, dt, n, fill_dst, fill_vals, n_fill, s)
                        : launch_combine_fill<double, 2>(out, y0, k, coef, dt, n, fill_dst, fill_vals, n_fill, s);
}

int tdeq_error_norm(void* scaled_out, const void* y0, const void* y1, const void* const* k,
                    const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                    const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                    double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                    void* stream) {
    if (!y0 || !y1 || !k || !coef || !out_sumsq || !out_nonfinite || !workspace || (bad_norm_dtype(dtype) && !lp_dtype(dtype)))
        return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    // bf16 / fp16: out_sumsq[s] = sum of fl(|r|^2) — |r| itself for a one-element segment (tdeq_kernels_lp.hpp norm_term)
    if (dtype == TDEQ_BF16)
        return lp_dispatch_error<lp::BF16>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s);
    if (dtype == TDEQ_F16)
        return lp_dispatch_error<lp::F16>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s);
    if (dtype == TDEQ_C64)
        return dispatch_cplx_error<float>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s);
    if (dtype == TDEQ_C128)
        return dispatch_cplx_error<double>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s);
    return dtype == TDEQ_F32
               ? dispatch_error<float>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s)
               : dispatch_error<double>(scaled_out, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s);
}

int tdeq_error_norm_vec(const void* err_partial, const void* y0, const void* y1, const void* const* k, const double* coef,
                        int n_terms, double dt, const double* rtol_vec, double rtol_scalar, const double* atol_vec, double atol_scalar,
                        const tdeq_segment* segs, const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks,
                        double* out_sumsq, double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                        void* stream) {
    if (!y0 || !y1 || !out_sumsq || !out_nonfinite || !workspace || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (!rtol_vec && !atol_vec) return TDEQ_EINVAL;      // two 0-dim tolerances: tdeq_error_norm (a different promotion)
    if (n_terms < (err_partial ? 0 : 1) || n_terms > TDEQ_MAX_TERMS || (n_terms > 0 && (!k || !coef))) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    return dtype == TDEQ_F32
               ? dispatch_error_vec<float>(err_partial, y0, y1, k, coef, n_terms, dt, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st,
                                           out_sumsq, out_nonfinite, ws, s)
               : dispatch_error_vec<double>(err_partial, y0, y1, k, coef, n_terms, dt, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st,
                                            out_sumsq, out_nonfinite, ws, s);
}

int tdeq_error_norm_vec_ctrl(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                             const double* coef, int n_terms, double dt, const double* rtol_vec, double rtol_scalar, const double* atol_vec, double atol_scalar,
                             const tdeq_segment* segs, const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks,
                             double* out_sumsq, double* out_nonfinite, const tdeq_step_ctrl* ctrl, double* out_ctrl,
                             double* ctrl_dev, void* next_times, int state_in_dev, void* workspace, size_t workspace_bytes,
                             int dtype, void* stream) {
    if (!y0 || !y1 || !out_sumsq || !out_nonfinite || !workspace || bad_dtype(dtype)) return TDEQ_EINVAL;
    if ((!rtol_vec && !atol_vec) || !ctrl || !out_ctrl || !ctrl_dev || !next_times) return TDEQ_EINVAL;
    if (n_terms < (err_partial ? 0 : 1) || n_terms > TDEQ_MAX_TERMS || (n_terms > 0 && (!k || !coef))) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (ctrl->n_times < 1 || ctrl->n_times > TDEQ_MAX_STAGE_TIMES || ctrl->n_norm_seg < 0 || ctrl->n_norm_seg > n_seg)
        return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    const CtrlBundle cb{ctrl, out_ctrl, ctrl_dev, next_times, state_in_dev ? 1 : 0};
    return dtype == TDEQ_F32
               ? dispatch_error_vec<float>(err_partial, y0, y1, k, coef, n_terms, dt, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st,
                                           out_sumsq, out_nonfinite, ws, s, &cb)
               : dispatch_error_vec<double>(err_partial, y0, y1, k, coef, n_terms, dt, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st,
                                            out_sumsq, out_nonfinite, ws, s, &cb);
}

int tdeq_stage_combine_err(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                           const double* err_coef, int n_terms, double dt, int64_t n, int dtype, void* stream) {
    if (!out || !err_out || !y0 || !k || !coef || !err_coef || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (bf16 / fp16: err_out is the row sum over THESE stages rounded once — a caller that continues it rounds twice)
    if (dtype == TDEQ_BF16) return lp_dispatch_combine<lp::BF16, 2>(out, err_out, y0, k, coef, err_coef, n_terms, dt, n, nullptr, nullptr, 0, s);
    if (dtype == TDEQ_F16) return lp_dispatch_combine<lp::F16, 2>(out, err_out, y0, k, coef, err_coef, n_terms, dt, n, nullptr, nullptr, 0, s);
    return dtype == TDEQ_F32 ? dispatch_combine_err<float>(out, err_out, y0, k, coef, err_coef, n_terms, dt, n, s)
                             : dispatch_combine_err<double>(out, err_out, y0, k, coef, err_coef, n_terms, dt, n, s);
}

static int combine_multi_checked(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                                 const void* const* k, int n_terms, double dt, int64_t n, int dtype, void* stream,
                                 void* e0, void* e1, const double* dt_dev = nullptr) {
    if (!outs || !y0 || !k || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS || n_out < 1 || n_out > TDEQ_MAX_MULTI_OUT) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    for (int o = 0; o < n_out; ++o) {
        if (!outs[o].out || outs[o].mask == 0u) return TDEQ_EINVAL;
        if (n_terms < 32 && (outs[o].mask >> n_terms) != 0u) return TDEQ_EINVAL;
    }
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? dispatch_combine_multi<float>(outs, n_out, y0, acc_in, k, n_terms, dt, n, s, static_cast<hipEvent_t>(e0), static_cast<hipEvent_t>(e1), dt_dev)
                             : dispatch_combine_multi<double>(outs, n_out, y0, acc_in, k, n_terms, dt, n, s, static_cast<hipEvent_t>(e0), static_cast<hipEvent_t>(e1), dt_dev);
}

int tdeq_stage_combine_multi(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                             const void* const* k, int n_terms, double dt, int64_t n, int dtype, void* stream) {
    return combine_multi_checked(outs, n_out, y0, acc_in, k, n_terms, dt, n, dtype, stream, nullptr, nullptr);
}

int tdeq_stage_combine_multi_dev(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                                 const void* const* k, int n_terms, const double* ctrl_dev, int64_t n, int dtype,
                                 void* stream) {
    if (!ctrl_dev) return TDEQ_EINVAL;
    return combine_multi_checked(outs, n_out, y0, acc_in, k, n_terms, 0.0, n, dtype, stream, nullptr, nullptr, ctrl_dev);
}

int tdeq_stage_combine_multi_timed(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                                   const void* const* k, int n_terms, double dt, int64_t n, int dtype, void* stream,
                                   void* start_event, void* stop_event) {
    if (!start_event || !stop_event) return TDEQ_EINVAL;
    return combine_multi_checked(outs, n_out, y0, acc_in, k, n_terms, dt, n, dtype, stream, start_event, stop_event);
}

int tdeq_error_norm_partial(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                            const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                            const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                            double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                            void* stream) {
    if (!err_partial || !y0 || !y1 || !out_sumsq || !out_nonfinite || !workspace || bad_norm_dtype(dtype))
        return TDEQ_EINVAL;
    if (n_terms < 0 || n_terms > 2 || (n_terms > 0 && (!k || !coef))) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (dtype == TDEQ_C64)
        return dispatch_cplx_error_partial<float>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, nullptr, s);
    if (dtype == TDEQ_C128)
        return dispatch_cplx_error_partial<double>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, nullptr, s);
    return dtype == TDEQ_F32
               ? dispatch_error_partial<float>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, nullptr, s)
               : dispatch_error_partial<double>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, nullptr, s);
}

int tdeq_error_norm_partial_ctrl(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                                 const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                                 const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                                 double* out_nonfinite, const tdeq_step_ctrl* ctrl, double* out_ctrl, double* ctrl_dev,
                                 void* next_times, int state_in_dev, void* copy_last_k, void* workspace,
                                 size_t workspace_bytes, int dtype, void* stream) {
    if (copy_last_k && (lp_dtype(dtype) || dtype == TDEQ_C64 || dtype == TDEQ_C128 || n_terms < 1)) return TDEQ_EINVAL;
    if (lp_dtype(dtype)) {
        // bf16 / fp16: the WHOLE error row in one launch (a row is rounded once: no partial sum to continue — err_partial
        // must be NULL), then finalize + controller in the state's type
        if (err_partial || !y0 || !y1 || !k || !coef || !out_sumsq || !out_nonfinite || !workspace) return TDEQ_EINVAL;
        if (!ctrl || !out_ctrl || !ctrl_dev || !next_times || n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
        for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
        if (ctrl->n_times < 1 || ctrl->n_times > TDEQ_MAX_STAGE_TIMES || ctrl->n_norm_seg < 0 || ctrl->n_norm_seg > n_seg)
            return TDEQ_EINVAL;
        SegTable st;
        const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
        if (e) return e;
        if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
        hipStream_t s = static_cast<hipStream_t>(stream);
        double* ws = static_cast<double*>(workspace);
        const CtrlBundle cb{ctrl, out_ctrl, ctrl_dev, next_times, state_in_dev ? 1 : 0};
        return dtype == TDEQ_BF16
                   ? lp_dispatch_error<lp::BF16>(nullptr, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s, &cb)
                   : lp_dispatch_error<lp::F16>(nullptr, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, s, &cb);
    }
    if (!err_partial || !y0 || !y1 || !out_sumsq || !out_nonfinite || !workspace || bad_norm_dtype(dtype))
        return TDEQ_EINVAL;
    if (!ctrl || !out_ctrl || !ctrl_dev || !next_times) return TDEQ_EINVAL;
    if (n_terms < 0 || n_terms > 2 || (n_terms > 0 && (!k || !coef))) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (ctrl->n_times < 1 || ctrl->n_times > TDEQ_MAX_STAGE_TIMES || ctrl->n_norm_seg < 0 || ctrl->n_norm_seg > n_seg)
        return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    const CtrlBundle cb{ctrl, out_ctrl, ctrl_dev, next_times, state_in_dev ? 1 : 0};
    if (dtype == TDEQ_C64)
        return dispatch_cplx_error_partial<float>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, &cb, s);
    if (dtype == TDEQ_C128)
        return dispatch_cplx_error_partial<double>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, &cb, s);
    return dtype == TDEQ_F32
               ? dispatch_error_partial<float>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, &cb, s, copy_last_k)
               : dispatch_error_partial<double>(err_partial, y0, y1, k, coef, n_terms, dt, st, out_sumsq, out_nonfinite, ws, &cb, s, copy_last_k);
}

int tdeq_step_controller(const double* sums, const double* nonfinite, const tdeq_segment* segs, const void* segs_dev,
                         int n_seg, double* out_sumsq, double* out_nonfinite, const tdeq_step_ctrl* ctrl,
                         double* out_ctrl, double* ctrl_dev, void* next_times, int state_in_dev, int dtype,
                         void* stream) {
    if (!sums || !nonfinite || !segs || !out_sumsq || !out_nonfinite || !ctrl || !out_ctrl || !ctrl_dev ||
        !next_times || bad_dtype(dtype))
        return TDEQ_EINVAL;
    if (ctrl->n_times < 1 || ctrl->n_times > TDEQ_MAX_STAGE_TIMES || ctrl->n_norm_seg < 0 || ctrl->n_norm_seg > n_seg)
        return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, TDEQ_CHUNK_QUANTUM, 1);   // only numel is read
    if (e) return e;
    CtrlArgs a;
    a.part_sumsq = nullptr;
    a.part_bad = nullptr;
    a.st = st;
    a.c = *ctrl;
    a.is_f32 = dtype == TDEQ_F32 ? 1 : 0;
    a.ratio_kind = a.is_f32;
    a.out_sumsq = out_sumsq;
    a.out_bad = out_nonfinite;
    a.out_ctrl = out_ctrl;
    a.ctrl_dev = ctrl_dev;
    a.next_times = next_times;
    a.state_in_dev = state_in_dev ? 1 : 0;
    a.presummed = 1;
    a.in_sumsq = sums;
    a.in_bad = nonfinite;
    hipLaunchKernelGGL(norm_finalize_ctrl_kernel<1>, dim3(1), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    return check_launch();
}

int tdeq_stage_combine_dev(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                           const double* err_coef, int n_terms, const double* ctrl_dev, int64_t n, int dtype,
                           void* stream) {
    if (!out || !y0 || !k || !coef || !ctrl_dev || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS || (err_out && !err_coef)) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (lp_dtype(dtype) && err_out) return TDEQ_EINVAL;      // a 16-bit row is rounded once: no partial error to continue
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16)
        return lp_dispatch_combine<lp::BF16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, 0.0, n, nullptr, nullptr, 0, s, ctrl_dev);
    if (dtype == TDEQ_F16)
        return lp_dispatch_combine<lp::F16, 1>(out, nullptr, y0, k, coef, nullptr, n_terms, 0.0, n, nullptr, nullptr, 0, s, ctrl_dev);
    return dtype == TDEQ_F32
               ? launch_combine_dev<float>(out, err_out, y0, k, coef, err_coef, n_terms, ctrl_dev + 1, n, s)
               : launch_combine_dev<double>(out, err_out, y0, k, coef, err_coef, n_terms, ctrl_dev + 1, n, s);
}

int tdeq_step_commit(void* y_prev, void* f_prev, void* y_cur, void* f_cur, const void* y1, const void* f1,
                     const double* ctrl_dev, int64_t n, int dtype, void* stream) {
    if (!y_prev || !f_prev || !y_cur || !f_cur || !y1 || !f1 || !ctrl_dev || n < 0 || bad_dtype(dtype))
        return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? launch_commit<float>(y_prev, f_prev, y_cur, f_cur, y1, f1, ctrl_dev, n, s)
                             : launch_commit<double>(y_prev, f_prev, y_cur, f_cur, y1, f1, ctrl_dev, n, s);
}

int tdeq_stage_combine_sel(void* out, const void* y_acc, const void* f_acc, const void* y_rej, const void* f_rej,
                           double coef, const double* ctrl_dev, int64_t n, int dtype, void* stream) {
    if (!out || !y_acc || !f_acc || !y_rej || !f_rej || !ctrl_dev || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_launch_sel<lp::BF16>(out, y_acc, f_acc, y_rej, f_rej, coef, ctrl_dev, n, s);
    if (dtype == TDEQ_F16) return lp_launch_sel<lp::F16>(out, y_acc, f_acc, y_rej, f_rej, coef, ctrl_dev, n, s);
    return dtype == TDEQ_F32 ? launch_combine_sel<float>(out, y_acc, f_acc, y_rej, f_rej, coef, ctrl_dev, n, s)
                             : launch_combine_sel<double>(out, y_acc, f_acc, y_rej, f_rej, coef, ctrl_dev, n, s);
}

int tdeq_init_norms(int mode, const void* a, const void* b, const void* yscale, const tdeq_segment* segs,
                    const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                    double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                    void* stream) {
    if ((mode != 0 && mode != 1) || !a || !b || !yscale || !out_sumsq || !out_nonfinite || !workspace ||
        (bad_norm_dtype(dtype) && !lp_dtype(dtype)))
        return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    if (dtype == TDEQ_BF16) return lp_launch_init<lp::BF16>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, nullptr, nullptr, s);
    if (dtype == TDEQ_F16) return lp_launch_init<lp::F16>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, nullptr, nullptr, s);
    if (dtype == TDEQ_C64) return launch_cplx_init<float>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, nullptr, nullptr, s);
    if (dtype == TDEQ_C128) return launch_cplx_init<double>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, nullptr, nullptr, s);
    return dtype == TDEQ_F32 ? launch_init<float>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, s)
                             : launch_init<double>(mode, a, b, yscale, st, out_sumsq, out_nonfinite, ws, s);
}

int tdeq_init_norms_vec(int mode, const void* a, const void* b, const void* yscale, const double* rtol_vec,
                        double rtol_scalar, const double* atol_vec, double atol_scalar, const tdeq_segment* segs,
                        const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                        double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if ((mode != 0 && mode != 1) || !a || !b || !yscale || !out_sumsq || !out_nonfinite || !workspace || bad_dtype(dtype))
        return TDEQ_EINVAL;
    if (!rtol_vec && !atol_vec) return TDEQ_EINVAL;      // two 0-dim tolerances: tdeq_init_norms (a different promotion)
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    return dtype == TDEQ_F32
               ? launch_init_vec<float>(mode, a, b, yscale, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st, out_sumsq,
                                        out_nonfinite, ws, s)
               : launch_init_vec<double>(mode, a, b, yscale, rtol_vec, rtol_scalar, atol_vec, atol_scalar, st, out_sumsq,
                                         out_nonfinite, ws, s);
}

int tdeq_init_scaled(int mode, const void* a, const void* b, const void* yscale, const tdeq_segment* segs,
                     const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, void* out0, void* out1,
                     int dtype, void* stream) {
    if ((mode != 0 && mode != 1) || !a || !b || !yscale || !out0 || (mode == 0 && !out1) ||
        (bad_norm_dtype(dtype) && !lp_dtype(dtype)))
        return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_launch_init<lp::BF16>(mode, a, b, yscale, st, nullptr, nullptr, nullptr, out0, out1, s);
    if (dtype == TDEQ_F16) return lp_launch_init<lp::F16>(mode, a, b, yscale, st, nullptr, nullptr, nullptr, out0, out1, s);
    if (dtype == TDEQ_C64) return launch_cplx_init<float>(mode, a, b, yscale, st, nullptr, nullptr, nullptr, out0, out1, s);
    if (dtype == TDEQ_C128) return launch_cplx_init<double>(mode, a, b, yscale, st, nullptr, nullptr, nullptr, out0, out1, s);
    const dim3 g((unsigned)st.n_chunks), blk(kBlock);
    if (dtype == TDEQ_F32) {
        InitScaledArgs<float> x{static_cast<const float*>(a), static_cast<const float*>(b),
                                static_cast<const float*>(yscale), st, static_cast<float*>(out0),
                                static_cast<float*>(out1)};
        if (mode == 0) hipLaunchKernelGGL((init_scaled_kernel<float, 0>), g, blk, 0, s, x);
        else hipLaunchKernelGGL((init_scaled_kernel<float, 1>), g, blk, 0, s, x);
    } else {
        InitScaledArgs<double> x{static_cast<const double*>(a), static_cast<const double*>(b),
                                 static_cast<const double*>(yscale), st, static_cast<double*>(out0),
                                 static_cast<double*>(out1)};
        if (mode == 0) hipLaunchKernelGGL((init_scaled_kernel<double, 0>), g, blk, 0, s, x);
        else hipLaunchKernelGGL((init_scaled_kernel<double, 1>), g, blk, 0, s, x);
    }
    return check_launch();
}

int tdeq_dense_eval(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                    const void* const* k, const double* coef, int n_terms, double dt, double x, int64_t n,
                    int dtype, void* stream) {
    if (!out || !y0 || !y1 || !f0 || !f1 || !k || !coef || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dense_eval<lp::BF16>(out, n, y0, y1, f0, f1, k, coef, n_terms, dt, &x, 1, n, s);
    if (dtype == TDEQ_F16) return lp_dense_eval<lp::F16>(out, n, y0, y1, f0, f1, k, coef, n_terms, dt, &x, 1, n, s);
    return dtype == TDEQ_F32
               ? dispatch_dense<float, false>(out, y0, y1, f0, f1, k, coef, n_terms, dt, x, n, s)
               : dispatch_dense<double, false>(out, y0, y1, f0, f1, k, coef, n_terms, dt, x, n, s);
}

int tdeq_dense_eval_multi(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0,
                          const void* f1, const void* const* k, const double* coef, int n_terms, double dt,
                          const double* x, int n_x, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !y1 || !f0 || !f1 || !k || !coef || !x || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS || n_x < 1 || n_x > TDEQ_MAX_DENSE_OUTPUTS) return TDEQ_EINVAL;
    if (n_x > 1 && out_stride < n) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dense_eval<lp::BF16>(out, out_stride, y0, y1, f0, f1, k, coef, n_terms, dt, x, n_x, n, s);
    if (dtype == TDEQ_F16) return lp_dense_eval<lp::F16>(out, out_stride, y0, y1, f0, f1, k, coef, n_terms, dt, x, n_x, n, s);
    return dtype == TDEQ_F32
               ? dispatch_dense_multi<float>(out, out_stride, y0, y1, f0, f1, k, coef, n_terms, dt, x, n_x, n, s)
               : dispatch_dense_multi<double>(out, out_stride, y0, y1, f0, f1, k, coef, n_terms, dt, x, n_x, n, s);
}

int tdeq_interp_fit(void* coeffs, const void* y0, const void* y1, const void* f0, const void* f1,
                    const void* const* k, const double* coef, int n_terms, double dt, int64_t n, int dtype,
                    void* stream) {
    if (!coeffs || !y0 || !y1 || !f0 || !f1 || !k || !coef || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_fit<lp::BF16>(coeffs, y0, y1, f0, f1, k, coef, n_terms, dt, n, s);
    if (dtype == TDEQ_F16) return lp_dispatch_fit<lp::F16>(coeffs, y0, y1, f0, f1, k, coef, n_terms, dt, n, s);
    return dtype == TDEQ_F32
               ? dispatch_dense<float, true>(coeffs, y0, y1, f0, f1, k, coef, n_terms, dt, 0.0, n, s)
               : dispatch_dense<double, true>(coeffs, y0, y1, f0, f1, k, coef, n_terms, dt, 0.0, n, s);
}

int tdeq_rk4_38_stage(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3,
                      const void* k4, double dt, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n == 0) return (stage >= 1 && stage <= 4) ? 0 : TDEQ_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_rk4<lp::BF16>(stage, out, y0, k1, k2, k3, k4, dt, n, s);
    if (dtype == TDEQ_F16) return lp_dispatch_rk4<lp::F16>(stage, out, y0, k1, k2, k3, k4, dt, n, s);
    return dtype == TDEQ_F32 ? dispatch_rk4<float>(stage, out, y0, k1, k2, k3, k4, dt, n, s)
                             : dispatch_rk4<double>(stage, out, y0, k1, k2, k3, k4, dt, n, s);
}

int tdeq_rk4_38_stage_dev(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3,
                          const void* k4, const double* dt_dev, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !dt_dev || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n == 0) return (stage >= 1 && stage <= 4) ? 0 : TDEQ_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? dispatch_rk4<float>(stage, out, y0, k1, k2, k3, k4, 0.0, n, s, dt_dev)
                             : dispatch_rk4<double>(stage, out, y0, k1, k2, k3, k4, 0.0, n, s, dt_dev);
}

int tdeq_grid_advance(const void* grid, int grid_dtype, int64_t n_grid, int64_t* counter, int perturb, double sign,
                      void* times_out, double* dt_out, int state_dtype, void* stream) {
    if (!grid || !counter || !times_out || !dt_out || n_grid < 2 || bad_dtype(grid_dtype) || bad_dtype(state_dtype))
        return TDEQ_EINVAL;
    GridAdvanceArgs a;
    a.grid = grid;
    a.grid_is_f32 = grid_dtype == TDEQ_F32;
    a.n_grid = n_grid;
    a.counter = counter;
    a.perturb = perturb ? 1 : 0;
    a.sign = sign;
    a.times_out = times_out;
    a.state_is_f32 = state_dtype == TDEQ_F32;
    a.dt_out = dt_out;
    hipLaunchKernelGGL(grid_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return check_launch();
}

int tdeq_grid_advance_stages(const void* grid, int grid_dtype, int64_t n_grid, int64_t* counter, int perturb, double sign,
                             const double* frac, const int* mode, int n_times, void* times_out, double* dt_out,
                             int state_dtype, void* stream) {
    if (!grid || !counter || !times_out || !dt_out || !frac || !mode || n_grid < 2 || bad_dtype(grid_dtype) ||
        bad_dtype(state_dtype) || n_times < 1 || n_times > kMaxGridStages)
        return TDEQ_EINVAL;
    GridStagesArgs a;
    a.base.grid = grid;
    a.base.grid_is_f32 = grid_dtype == TDEQ_F32;
    a.base.n_grid = n_grid;
    a.base.counter = counter;
    a.base.perturb = perturb ? 1 : 0;
    a.base.sign = sign;
    a.base.times_out = times_out;
    a.base.state_is_f32 = state_dtype == TDEQ_F32;
    a.base.dt_out = dt_out;
    a.n_times = n_times;
    for (int i = 0; i < kMaxGridStages; ++i) {
        a.frac[i] = i < n_times ? frac[i] : 0.0;
        a.mode[i] = i < n_times ? mode[i] : 0;
    }
    hipLaunchKernelGGL(grid_advance_stages_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return check_launch();
}

int tdeq_grid_commit(void* solution, int64_t row_stride, void* y_cur, const void* y_new, const int64_t* counter,
                     int64_t n, int dtype, void* stream) {
    if (!solution || !y_cur || !y_new || !counter || n < 0 || row_stride < n || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? launch_grid_commit<float>(solution, row_stride, y_cur, y_new, counter, n, s)
                             : launch_grid_commit<double>(solution, row_stride, y_cur, y_new, counter, n, s);
}

int tdeq_lerp(void* out, const void* y0, const void* y1, double slope, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !y1 || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_launch_lerp<lp::BF16>(out, y0, y1, slope, n, s);
    if (dtype == TDEQ_F16) return lp_launch_lerp<lp::F16>(out, y0, y1, slope, n, s);
    return dtype == TDEQ_F32 ? launch_lerp<float>(out, y0, y1, slope, n, s)
                             : launch_lerp<double>(out, y0, y1, slope, n, s);
}

int tdeq_fixed_stage(int mode, void* out, const void* y0, const void* const* k, const double* w, int n_terms,
                     double dt, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !k || !w || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if ((mode != 0 && mode != 1) || n_terms < 1 || n_terms > 4 || (mode == 1 && n_terms != 1)) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_fixed<lp::BF16>(mode, out, y0, k, w, n_terms, dt, n, s);
    if (dtype == TDEQ_F16) return lp_dispatch_fixed<lp::F16>(mode, out, y0, k, w, n_terms, dt, n, s);
    if (dtype == TDEQ_F32)
        return mode == 0 ? dispatch_fixed<float, 0>(out, y0, k, w, n_terms, dt, n, s)
                         : dispatch_fixed<float, 1>(out, y0, k, w, n_terms, dt, n, s);
    return mode == 0 ? dispatch_fixed<double, 0>(out, y0, k, w, n_terms, dt, n, s)
                     : dispatch_fixed<double, 1>(out, y0, k, w, n_terms, dt, n, s);
}

int tdeq_fixed_stage_dev(int mode, void* out, const void* y0, const void* const* k, const double* w, int n_terms,
                         const double* dt_dev, int64_t n, int dtype, void* stream) {
    if (!out || !y0 || !k || !w || !dt_dev || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if ((mode != 0 && mode != 1) || n_terms < 1 || n_terms > 4 || (mode == 1 && n_terms != 1)) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!k[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_F32)
        return mode == 0 ? dispatch_fixed<float, 0>(out, y0, k, w, n_terms, 0.0, n, s, dt_dev)
                         : dispatch_fixed<float, 1>(out, y0, k, w, n_terms, 0.0, n, s, dt_dev);
    return mode == 0 ? dispatch_fixed<double, 0>(out, y0, k, w, n_terms, 0.0, n, s, dt_dev)
                     : dispatch_fixed<double, 1>(out, y0, k, w, n_terms, 0.0, n, s, dt_dev);
}

int tdeq_weighted_sum(void* out, const void* const* x, const double* w, int n_terms, int64_t n, int dtype,
                      void* stream) {
    if (!out || !x || !w || n < 0 || (bad_dtype(dtype) && !lp_dtype(dtype))) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_SUM_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!x[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == TDEQ_BF16) return lp_dispatch_weighted<lp::BF16>(out, x, w, n_terms, n, s);
    if (dtype == TDEQ_F16) return lp_dispatch_weighted<lp::F16>(out, x, w, n_terms, n, s);
    return dtype == TDEQ_F32 ? dispatch_fixed<float, 2>(out, nullptr, x, w, n_terms, 0.0, n, s)
                             : dispatch_fixed<double, 2>(out, nullptr, x, w, n_terms, 0.0, n, s);
}

int tdeq_scale_many(void* const* outs, const void* g, const double* w, int n_out, int64_t n, int dtype,
                    void* stream) {
    if (!outs || !g || !w || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n_out < 1 || n_out > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_out; ++j) if (!outs[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? dispatch_scale<float>(outs, g, w, n_out, n, s)
                             : dispatch_scale<double>(outs, g, w, n_out, n, s);
}

size_t tdeq_dots_workspace_bytes(int64_t n, int n_x) {
    int64_t n_chunks = (n + kDotChunk - 1) / kDotChunk;
    if (n_chunks < 1) n_chunks = 1;
    if (n_x < 1) n_x = 1;
    return (size_t)n_chunks * (size_t)n_x * sizeof(double);
}

int tdeq_multi_dot(const void* g, const void* const* x, int n_x, int64_t n, double* out, void* workspace,
                   size_t workspace_bytes, int dtype, void* stream) {
    if (!g || !x || !out || !workspace || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n_x < 1 || n_x > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_x; ++j) if (!x[j]) return TDEQ_EINVAL;
    if (workspace_bytes < tdeq_dots_workspace_bytes(n, n_x)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    return dtype == TDEQ_F32 ? dispatch_dots<float>(g, x, n_x, n, out, ws, s)
                             : dispatch_dots<double>(g, x, n_x, n, out, ws, s);
}

int tdeq_pack_segments(void* out, const void* const* src, const int64_t* chunk_start, const int64_t* numel,
                       const double* scale, int n_seg, int64_t chunk, int64_t n_chunks, int dtype, void* stream) {
    if (!out || !src || !chunk_start || !numel || !scale || bad_dtype(dtype)) return TDEQ_EINVAL;
    if (n_seg < 1 || n_seg > TDEQ_INLINE_SEGMENTS) return TDEQ_EINVAL;
    if (chunk < TDEQ_CHUNK_QUANTUM || chunk % TDEQ_CHUNK_QUANTUM != 0 || n_chunks < 1 || n_chunks > 0x7fffffffLL)
        return TDEQ_EINVAL;
    if (chunk_start[0] != 0) return TDEQ_EINVAL;
    for (int q = 0; q < n_seg; ++q) {
        if (numel[q] < 0 || (q > 0 && chunk_start[q] <= chunk_start[q - 1]) || chunk_start[q] >= n_chunks) return TDEQ_EINVAL;
        const int64_t end = (q + 1 < n_seg) ? chunk_start[q + 1] : n_chunks;
        if (q + 1 < n_seg && chunk_start[q + 1] <= chunk_start[q]) return TDEQ_EINVAL;
        if (numel[q] > (end - chunk_start[q]) * chunk) return TDEQ_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? launch_pack<float>(out, src, chunk_start, numel, scale, n_seg, chunk, n_chunks, s)
                             : launch_pack<double>(out, src, chunk_start, numel, scale, n_seg, chunk, n_chunks, s);
}

int tdeq_fill_scalars(void* dst, const double* vals, int n_vals, int dtype, void* stream) {
    if (!dst || !vals || n_vals < 1 || n_vals > 16 || bad_dtype(dtype)) return TDEQ_EINVAL;
    FillArgs a;
    a.dst = dst;
    for (int i = 0; i < 16; ++i) a.v[i] = i < n_vals ? vals[i] : 0.0;
    a.n = n_vals;
    a.dtype = dtype;
    hipLaunchKernelGGL(fill_scalars_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return check_launch();
}

int tdeq_adams_predict(void* y_out, void* dy_out, void* delta_out, const void* y0, const void* const* f_hist,
                       const double* cb, const double* cm, int n_terms, double dt, int64_t n, int dtype,
                       void* stream) {
    if (!y_out || !y0 || !f_hist || !cb || n < 0 || bad_dtype(dtype)) return TDEQ_EINVAL;
    if ((dy_out == nullptr) != (delta_out == nullptr)) return TDEQ_EINVAL;
    if (dy_out && !cm) return TDEQ_EINVAL;
    if (n_terms < 1 || n_terms > TDEQ_MAX_TERMS) return TDEQ_EINVAL;
    for (int j = 0; j < n_terms; ++j) if (!f_hist[j]) return TDEQ_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == TDEQ_F32 ? dispatch_adams_predict<float>(y_out, dy_out, delta_out, y0, f_hist, cb, cm, n_terms, dt, n, s)
                             : dispatch_adams_predict<double>(y_out, dy_out, delta_out, y0, f_hist, cb, cm, n_terms, dt, n, s);
}

int tdeq_adams_correct(void* y_out, void* dy_out, const void* f, const void* delta, const void* dy_old,
                       const void* y0, double c, int compute, const tdeq_segment* segs, const void* segs_dev,
                       int n_seg, int64_t chunk, int64_t n_chunks, int64_t n, double* out_count,
                       double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (!dy_out || !dy_old || !out_count || !out_nonfinite || !workspace || n < 0 || bad_dtype(dtype))
        return TDEQ_EINVAL;
    if (compute && (!y_out || !f || !delta || !y0)) return TDEQ_EINVAL;
    SegTable st;
    const int e = fill_segtable(st, segs, segs_dev, n_seg, chunk, n_chunks);
    if (e) return e;
    if (workspace_bytes < tdeq_workspace_bytes(n_chunks)) return TDEQ_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* ws = static_cast<double*>(workspace);
    return dtype == TDEQ_F32
               ? launch_adams_correct<float>(y_out, dy_out, f, delta, dy_old, y0, c, compute != 0, st, n, out_count, out_nonfinite, ws, s)
               : launch_adams_correct<double>(y_out, dy_out, f, delta, dy_old, y0, c, compute != 0, st, n, out_count, out_nonfinite, ws, s);
}

}  // extern "C"
