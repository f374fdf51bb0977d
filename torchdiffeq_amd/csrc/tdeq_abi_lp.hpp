// tdeq_abi_lp.hpp — host-side validation-free dispatch of the bfloat16 / float16 kernels (tdeq_kernels_lp.hpp).
// Included by tdeq_abi.hip inside its anonymous namespace (uses stream_grid, aligned16, check_launch, launch_finalize);
// the extern "C" entry points route dtype TDEQ_BF16 / TDEQ_F16 here after their own argument checks.
//
// Host scalars follow the torch-op host path of the package (torchdiffeq_amd/_fallback.py, pinned bit for bit to the
// reference on the CPU): `rs<S>(x)` = a double rounded to float32 and then to the storage type (a 0-dim tensor of the
// state's type, or a Python number that ATen rounds first), `(float)x` = a second operand taken at float32 (opmath).
#pragma once

// (tdeq_kernels_lp.hpp is included by tdeq_abi.hip at file scope)

template <typename S>
inline float rs(double x) { return S::rnd((float)x); }

template <int NIN, int NOUT>
inline bool lp_aligned(const lp::MapArgs<NIN, NOUT>& a) {
    bool vec = true;
    for (int j = 0; j < NIN; ++j) vec = vec && aligned16(a.in[j]);
    for (int o = 0; o < NOUT; ++o) vec = vec && (o >= a.n_live || aligned16(a.out[o]));
    return vec;
}

// LEANABLE (the stage combines): the common launch shapes get map_kernel's LEAN instantiations — see there.
template <typename S, int NIN, int NOUT, typename F, bool LEANABLE = false>
int lp_launch_map(lp::MapArgs<NIN, NOUT>& a, const F& f, hipStream_t s, bool needs_prepare = true) {
    if (lp_aligned(a)) {
        const dim3 g(stream_grid(a.n / lp::kVec, kBlock)), b(kBlock);
        if constexpr (LEANABLE) {
            if ((int64_t)g.x * kBlock >= a.n / lp::kVec && a.n_fill == 0 && a.n_live == NOUT) {
                if (needs_prepare) hipLaunchKernelGGL((lp::map_kernel<S, NIN, NOUT, true, F, true, true>), g, b, 0, s, a, f);
                else hipLaunchKernelGGL((lp::map_kernel<S, NIN, NOUT, true, F, true, false>), g, b, 0, s, a, f);
                return check_launch();
            }
        }
        hipLaunchKernelGGL((lp::map_kernel<S, NIN, NOUT, true, F>), g, b, 0, s, a, f);
    } else {
        hipLaunchKernelGGL((lp::map_kernel<S, NIN, NOUT, false, F>), dim3(stream_grid(a.n, kBlock)), dim3(kBlock), 0, s, a, f);
    }
    return check_launch();
}

template <int NIN, int NOUT>
inline void lp_no_fill(lp::MapArgs<NIN, NOUT>& a, int64_t n) {
    a.n = n;
    a.n_live = NOUT;
    a.fill_dst = nullptr;
    a.n_fill = 0;
    for (int i = 0; i < 16; ++i) a.fill_v[i] = 0;
}

// ---- stage combines: out = y0 + row_sum, optionally a second row without y0 (the partial error) and the side fill ----
template <typename S, int NT, int NOUT>
int lp_launch_combine(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                      const double* err_coef, double dt, int64_t n, void* fill_dst, const double* fill_vals, int n_fill,
                      hipStream_t s, const double* ctrl_dev = nullptr) {
    lp::MapArgs<NT + 1, NOUT> a;
    lp_no_fill(a, n);
    lp::CombineF<S, NT, NOUT> f;
    const float dtS = rs<S>(dt);
    a.in[0] = static_cast<const uint16_t*>(y0);
    for (int j = 0; j < NT; ++j) {
        a.in[1 + j] = static_cast<const uint16_t*>(k[j]);
        // fl_S(fl_S(coef) * fl_S(dt)) — rk_common.py:79,201-205; with ctrl_dev the kernel multiplies by the device's dt
        f.c[0][j] = ctrl_dev ? rs<S>(coef[j]) : S::rnd(rs<S>(coef[j]) * dtS);
        if (NOUT == 2) f.c[NOUT - 1][j] = ctrl_dev ? rs<S>(err_coef[j]) : S::rnd(rs<S>(err_coef[j]) * dtS);
    }
    a.out[0] = static_cast<uint16_t*>(out);
    if (NOUT == 2) a.out[NOUT - 1] = static_cast<uint16_t*>(err_out);
    f.ctrl_dev = ctrl_dev;
    if (fill_dst) {
        a.fill_dst = static_cast<uint16_t*>(fill_dst);
        a.n_fill = n_fill;
        for (int i = 0; i < n_fill; ++i) a.fill_v[i] = (uint16_t)S::st((float)fill_vals[i]);
    }
    return lp_launch_map<S, NT + 1, NOUT, lp::CombineF<S, NT, NOUT>, true>(a, f, s, ctrl_dev != nullptr);
}

template <typename S, int NOUT>
int lp_dispatch_combine(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                        const double* err_coef, int nt, double dt, int64_t n, void* fill_dst, const double* fill_vals,
                        int n_fill, hipStream_t s, const double* ctrl_dev = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return lp_launch_combine<S, N, NOUT>(out, err_out, y0, k, coef, err_coef, dt, n, fill_dst, fill_vals, n_fill, s, ctrl_dev);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- error norm: the sum of fl_S(|r|^2) per segment (|r| itself for a one-element segment) + the non-finite census ----
template <typename S, int NT>
int lp_launch_error(void* scaled, const void* y0, const void* y1, const void* const* k, const double* coef, double dt,
                    const SegTable& st, double* out_sumsq, double* out_bad, double* ws, hipStream_t s,
                    const CtrlBundle* cb = nullptr) {
    lp::ErrArgs<NT> a;
    a.scaled = static_cast<uint16_t*>(scaled);
    a.y0 = static_cast<const uint16_t*>(y0);
    a.y1 = static_cast<const uint16_t*>(y1);
    bool vec = aligned16(y0) && aligned16(y1) && aligned16(scaled) && (st.chunk % lp::kVec == 0);
    const float dtS = rs<S>(dt);
    for (int j = 0; j < NT; ++j) {
        a.k[j] = static_cast<const uint16_t*>(k[j]);
        // dt * c_error — rk_common.py:89 (captured steps: the kernel multiplies by the device's dt)
        a.c[j] = (cb && cb->state_in_dev) ? rs<S>(coef[j]) : S::rnd(rs<S>(coef[j]) * dtS);
        vec = vec && aligned16(k[j]);
    }
    a.ctrl_dev = (cb && cb->state_in_dev) ? cb->ctrl_dev : nullptr;
    a.st = st;
    a.part_sumsq = ws;
    a.part_bad = ws + 2 * st.n_chunks;
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (scaled) {
        if (vec) hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, false, true>), g, b, 0, s, a);
    } else {
        const bool single = st.n_seg == 1 && st.inl[0].chunk_start == 0;
        if (vec && single && !a.ctrl_dev) hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, true, false, true, false>), g, b, 0, s, a);
        else if (vec && single) hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, true, false, true, true>), g, b, 0, s, a);
        else if (vec) hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, true, false>), g, b, 0, s, a);
        else hipLaunchKernelGGL((lp::error_norm_kernel<S, NT, false, false>), g, b, 0, s, a);
    }
    const int e = check_launch();
    if (e) return e;
    // with a controller bundle: finalize + the step controller in the state's type (tkind 2 / 3), as for fp32 / fp64
    if (cb) return launch_finalize_ctrl(st, ws, out_sumsq, out_bad, *cb, S::code == TDEQ_BF16 ? 2 : 3, s, -1);
    return launch_finalize(st, ws, 1, out_sumsq, out_bad, s);
}

template <typename S>
int lp_dispatch_error(void* scaled, const void* y0, const void* y1, const void* const* k, const double* coef, int nt,
                      double dt, const SegTable& st, double* out_sumsq, double* out_bad, double* ws, hipStream_t s,
                      const CtrlBundle* cb = nullptr) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return lp_launch_error<S, N>(scaled, y0, y1, k, coef, dt, st, out_sumsq, out_bad, ws, s, cb);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- initial-step quotients: their sums (out0 == nullptr) or the quotients themselves ----
template <typename S>
int lp_launch_init(int mode, const void* a_, const void* b_, const void* y, const SegTable& st, double* out_sumsq,
                   double* out_bad, double* ws, void* out0, void* out1, hipStream_t s) {
    lp::InitArgs a;
    a.a = static_cast<const uint16_t*>(a_);
    a.b = static_cast<const uint16_t*>(b_);
    a.y = static_cast<const uint16_t*>(y);
    a.st = st;
    a.part0 = ws;
    a.part1 = ws ? ws + st.n_chunks : nullptr;
    a.part_bad = ws ? ws + 2 * st.n_chunks : nullptr;
    a.out0 = static_cast<uint16_t*>(out0);
    a.out1 = static_cast<uint16_t*>(out1);
    const dim3 g((unsigned)st.n_chunks), b(kBlock);
    if (out0) {
        if (mode == 0) hipLaunchKernelGGL((lp::init_norms_kernel<S, 0, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((lp::init_norms_kernel<S, 1, true>), g, b, 0, s, a);
        return check_launch();
    }
    if (mode == 0) hipLaunchKernelGGL((lp::init_norms_kernel<S, 0, false>), g, b, 0, s, a);
    else hipLaunchKernelGGL((lp::init_norms_kernel<S, 1, false>), g, b, 0, s, a);
    const int e = check_launch();
    if (e) return e;
    return launch_finalize(st, ws, mode == 0 ? 2 : 1, out_sumsq, out_bad, s);
}

// ---- dense output ----
template <typename S, int NT, int M>
int lp_launch_dense_eval(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0, const void* f1,
                         const void* const* k, const double* coef, double dt, const double* x, int n_x, int64_t n,
                         hipStream_t s) {
    lp::MapArgs<NT + 4, M> a;
    lp_no_fill(a, n);
    a.n_live = n_x;
    lp::DenseEvalF<S, NT, M> f;
    const float dtS = rs<S>(dt);
    a.in[0] = static_cast<const uint16_t*>(y0);
    a.in[1] = static_cast<const uint16_t*>(y1);
    a.in[2] = static_cast<const uint16_t*>(f0);
    a.in[3] = static_cast<const uint16_t*>(f1);
    for (int j = 0; j < NT; ++j) {
        a.in[4 + j] = static_cast<const uint16_t*>(k[j]);
        f.cm[j] = S::rnd(rs<S>(coef[j]) * dtS);
    }
    f.dt = dtS;
    f.two_dt = S::rnd(2.0f * dtS);
    for (int m = 0; m < M; ++m) {
        a.out[m] = static_cast<uint16_t*>(out) + (int64_t)(m < n_x ? m : 0) * out_stride;
        const float xS = rs<S>(m < n_x ? x[m] : 0.0);       // interp.py:40: x cast to the state's type
        f.xp[m][0] = xS;
        f.xp[m][1] = S::rnd(xS * xS);                        // interp.py:42-47: x_power *= x
        f.xp[m][2] = S::rnd(f.xp[m][1] * xS);
        f.xp[m][3] = S::rnd(f.xp[m][2] * xS);
    }
    return lp_launch_map<S, NT + 4, M>(a, f, s);
}

template <typename S, int M>
int lp_dispatch_dense_eval(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0, const void* f1,
                           const void* const* k, const double* coef, int nt, double dt, const double* x, int n_x, int64_t n,
                           hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return lp_launch_dense_eval<S, N, M>(out, out_stride, y0, y1, f0, f1, k, coef, dt, x, n_x, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

template <typename S>
int lp_dense_eval(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0, const void* f1,
                  const void* const* k, const double* coef, int nt, double dt, const double* x, int n_x, int64_t n,
                  hipStream_t s) {
    if (n_x == 1) return lp_dispatch_dense_eval<S, 1>(out, out_stride, y0, y1, f0, f1, k, coef, nt, dt, x, n_x, n, s);
    if (n_x <= 4) return lp_dispatch_dense_eval<S, 4>(out, out_stride, y0, y1, f0, f1, k, coef, nt, dt, x, n_x, n, s);
    // more rows: groups of 4 (the quartic is re-fitted per group — output rows stay bit-identical)
    for (int lo = 0; lo < n_x; lo += 4) {
        const int m = n_x - lo < 4 ? n_x - lo : 4;
        const int e = lp_dispatch_dense_eval<S, 4>(static_cast<uint16_t*>(out) + (int64_t)lo * out_stride, out_stride, y0, y1,
                                                   f0, f1, k, coef, nt, dt, x + lo, m, n, s);
        if (e) return e;
    }
    return 0;
}

template <typename S, int NT>
int lp_launch_fit(void* coeffs, const void* y0, const void* y1, const void* f0, const void* f1, const void* const* k,
                  const double* coef, double dt, int64_t n, hipStream_t s) {
    lp::MapArgs<NT + 4, 5> a;
    lp_no_fill(a, n);
    lp::DenseFitF<S, NT> f;
    const float dtS = rs<S>(dt);
    a.in[0] = static_cast<const uint16_t*>(y0);
    a.in[1] = static_cast<const uint16_t*>(y1);
    a.in[2] = static_cast<const uint16_t*>(f0);
    a.in[3] = static_cast<const uint16_t*>(f1);
    for (int j = 0; j < NT; ++j) {
        a.in[4 + j] = static_cast<const uint16_t*>(k[j]);
        f.cm[j] = S::rnd(rs<S>(coef[j]) * dtS);
    }
    f.dt = dtS;
    f.two_dt = S::rnd(2.0f * dtS);
    for (int p = 0; p < 5; ++p) a.out[p] = static_cast<uint16_t*>(coeffs) + (int64_t)p * n;
    return lp_launch_map<S, NT + 4, 5>(a, f, s);
}

template <typename S>
int lp_dispatch_fit(void* coeffs, const void* y0, const void* y1, const void* f0, const void* f1, const void* const* k,
                    const double* coef, int nt, double dt, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return lp_launch_fit<S, N>(coeffs, y0, y1, f0, f1, k, coef, dt, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7)
        TDEQ_CASE(8) TDEQ_CASE(9) TDEQ_CASE(10) TDEQ_CASE(11) TDEQ_CASE(12) TDEQ_CASE(13) TDEQ_CASE(14)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- fixed-grid stages ----
template <typename S, int STAGE>
int lp_launch_rk4(void* out, const void* y0, const void* const (&ks)[4], double dt, int64_t n, hipStream_t s) {
    lp::MapArgs<STAGE + 1, 1> a;
    lp_no_fill(a, n);
    a.in[0] = static_cast<const uint16_t*>(y0);
    for (int j = 0; j < STAGE; ++j) {
        if (!ks[j]) return TDEQ_EINVAL;
        a.in[1 + j] = static_cast<const uint16_t*>(ks[j]);
    }
    a.out[0] = static_cast<uint16_t*>(out);
    lp::Rk4F<S, STAGE> f;
    f.dt_first = rs<S>(dt);
    f.dt_second = (float)dt;
    f.third = (float)(1.0 / 3.0);
    return lp_launch_map<S, STAGE + 1, 1>(a, f, s);
}

template <typename S>
int lp_dispatch_rk4(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3, const void* k4,
                    double dt, int64_t n, hipStream_t s) {
    const void* const ks[4] = {k1, k2, k3, k4};
    switch (stage) {
        case 1: return lp_launch_rk4<S, 1>(out, y0, ks, dt, n, s);
        case 2: return lp_launch_rk4<S, 2>(out, y0, ks, dt, n, s);
        case 3: return lp_launch_rk4<S, 3>(out, y0, ks, dt, n, s);
        case 4: return lp_launch_rk4<S, 4>(out, y0, ks, dt, n, s);
    }
    return TDEQ_EINVAL;
}

template <typename S>
int lp_launch_lerp(void* out, const void* y0, const void* y1, double slope, int64_t n, hipStream_t s) {
    lp::MapArgs<2, 1> a;
    lp_no_fill(a, n);
    a.in[0] = static_cast<const uint16_t*>(y0);
    a.in[1] = static_cast<const uint16_t*>(y1);
    a.out[0] = static_cast<uint16_t*>(out);
    lp::LerpF<S> f;
    f.slope = rs<S>(slope);
    return lp_launch_map<S, 2, 1>(a, f, s);
}

template <typename S, int NT, int MODE>
int lp_launch_fixed(void* out, const void* y0, const void* const* k, const double* w, double dt, int64_t n, hipStream_t s) {
    lp::MapArgs<NT + 1, 1> a;
    lp_no_fill(a, n);
    a.in[0] = static_cast<const uint16_t*>(y0);
    lp::FixedF<S, NT, MODE> f;
    for (int j = 0; j < NT; ++j) {
        a.in[1 + j] = static_cast<const uint16_t*>(k[j]);
        f.w[j] = (float)w[j];
    }
    f.dt = rs<S>(dt);
    a.out[0] = static_cast<uint16_t*>(out);
    return lp_launch_map<S, NT + 1, 1>(a, f, s);
}

template <typename S>
int lp_dispatch_fixed(int mode, void* out, const void* y0, const void* const* k, const double* w, int nt, double dt,
                      int64_t n, hipStream_t s) {
    if (mode == 1) return lp_launch_fixed<S, 1, 1>(out, y0, k, w, dt, n, s);
    switch (nt) {
        case 1: return lp_launch_fixed<S, 1, 0>(out, y0, k, w, dt, n, s);
        case 2: return lp_launch_fixed<S, 2, 0>(out, y0, k, w, dt, n, s);
        case 3: return lp_launch_fixed<S, 3, 0>(out, y0, k, w, dt, n, s);
        case 4: return lp_launch_fixed<S, 4, 0>(out, y0, k, w, dt, n, s);
    }
    return TDEQ_EINVAL;
}

template <typename S, int NT>
int lp_launch_weighted(void* out, const void* const* x, const double* w, int64_t n, hipStream_t s) {
    lp::MapArgs<NT, 1> a;
    lp_no_fill(a, n);
    lp::WeightedF<S, NT> f;
    for (int j = 0; j < NT; ++j) {
        a.in[j] = static_cast<const uint16_t*>(x[j]);
        f.w[j] = rs<S>(w[j]);
    }
    a.out[0] = static_cast<uint16_t*>(out);
    return lp_launch_map<S, NT, 1>(a, f, s);
}

template <typename S>
int lp_dispatch_weighted(void* out, const void* const* x, const double* w, int nt, int64_t n, hipStream_t s) {
    switch (nt) {
#define TDEQ_CASE(N) case N: return lp_launch_weighted<S, N>(out, x, w, n, s);
        TDEQ_CASE(1) TDEQ_CASE(2) TDEQ_CASE(3) TDEQ_CASE(4) TDEQ_CASE(5) TDEQ_CASE(6) TDEQ_CASE(7) TDEQ_CASE(8)
#undef TDEQ_CASE
    }
    return TDEQ_EINVAL;
}

// ---- look-ahead first stage ----
template <typename S>
int lp_launch_sel(void* out, const void* y_acc, const void* f_acc, const void* y_rej, const void* f_rej, double coef,
                  const double* ctrl_dev, int64_t n, hipStream_t s) {
    lp::SelArgs a;
    a.out = static_cast<uint16_t*>(out);
    a.y_acc = static_cast<const uint16_t*>(y_acc);
    a.f_acc = static_cast<const uint16_t*>(f_acc);
    a.y_rej = static_cast<const uint16_t*>(y_rej);
    a.f_rej = static_cast<const uint16_t*>(f_rej);
    a.coef = rs<S>(coef);
    a.ctrl_dev = ctrl_dev;
    a.n = n;
    const bool vec = aligned16(out) && aligned16(y_acc) && aligned16(f_acc) && aligned16(y_rej) && aligned16(f_rej);
    if (vec) hipLaunchKernelGGL((lp::sel_kernel<S, true>), dim3(stream_grid(n / lp::kVec, kBlock)), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((lp::sel_kernel<S, false>), dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, a);
    return check_launch();
}
