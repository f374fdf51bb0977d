/*
 * rk_oracle.c — CPU restatement of the reference's explicit Runge–Kutta step arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for libtdeq_hip.so: it restates, in plain
 * C loops on host memory, the state-sized arithmetic of rtqichen/torchdiffeq v0.2.5 that the HIP
 * kernels replace.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (torchdiffeq_amd/) never does.
 *
 * Each function cites the reference lines it follows (paths relative to torchdiffeq/_impl/).  All
 * element arithmetic is in the state dtype T with every product and sum rounded separately
 * (-ffp-contract=off), coefficients formed as fl_T(fl_T(coef) * fl_T(dt)) like the reference's
 * `beta_i * dt` on a tableau already cast to T (rk_common.py:79, 201-205).  Reductions accumulate in
 * fp64 over fixed-size chunks (deterministic for any thread count); the reference accumulates in T
 * with ATen's blocked order, which differs from the exact value by a few ulp_T — see DESIGN.md.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against vectors produced by the
 * imported reference itself (tests/golden/make_golden.py -> tests/golden/ npz files).
 *
 * Signatures mirror include/tdeq_hip.h (minus the stream) so the same test harness drives both.
 *
 * OpenMP: every loop is `parallel for ... if(large)` — the threads only start from 16384 elements (9 chunks).  The
 * parity tests' states are a handful of elements; forking 8+ spin-waiting threads for them made a 10 s test take
 * minutes whenever anything else ran on the machine.  Results do not depend on the thread count either way.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define ORACLE_F32 0
#define ORACLE_F64 1
#define ORACLE_MAX_TERMS 14

typedef struct oracle_segment {
    int64_t chunk_start;
    int64_t numel;
    double rtol;
    double atol;
} oracle_segment;

int oracle_abi_version(void) { return 1; }

/* ---------------------------------------------------------------------------------------------------
 * yi = y0 + sum_j (beta_ij * dt) * k_j            rk_common.py:79 (and :83-85, misc.py:65)
 * ------------------------------------------------------------------------------------------------- */
#define DEF_COMBINE(NAME, T)                                                                          \
    static void NAME(T* out, const T* y0, const T* const* k, const double* coef, int nt, double dt,   \
                     int64_t n) {                                                                     \
        T c[ORACLE_MAX_TERMS];                                                                        \
        const T dtT = (T)dt;                                                                          \
        for (int j = 0; j < nt; ++j) c[j] = (T)coef[j] * dtT;                                         \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            T acc = k[0][i] * c[0];                                                                   \
            for (int j = 1; j < nt; ++j) acc = acc + k[j][i] * c[j];                                  \
            out[i] = y0[i] + acc;                                                                     \
        }                                                                                             \
    }
DEF_COMBINE(combine_f32, float)
DEF_COMBINE(combine_f64, double)

int oracle_stage_combine(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                         double dt, int64_t n, int dtype) {
    if (!out || !y0 || !k || !coef || n_terms < 1 || n_terms > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32) combine_f32((float*)out, (const float*)y0, (const float* const*)k, coef, n_terms, dt, n);
    else if (dtype == ORACLE_F64) combine_f64((double*)out, (const double*)y0, (const double* const*)k, coef, n_terms, dt, n);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * y1_error = sum_j (dt * c_error_j) * k_j                       rk_common.py:89
 * error_tol = atol + rtol * max(|y0|, |y1|); err / error_tol     misc.py:81-82
 * per-segment sum of squares (-> sqrt(mean) on the caller)       misc.py:22-23, 30-33
 * non-finite census of the state                                 rk_common.py:287
 * ------------------------------------------------------------------------------------------------- */
#define DEF_ERROR(NAME, T, ABS, MAX)                                                                  \
    static int NAME(T* scaled, const T* y0, const T* y1, const T* const* k, const double* coef,       \
                    int nt, double dt, const oracle_segment* segs, int n_seg, int64_t chunk,          \
                    int64_t n_chunks, double* out_sumsq, double* out_bad) {                           \
        T c[ORACLE_MAX_TERMS];                                                                        \
        const T dtT = (T)dt;                                                                          \
        for (int j = 0; j < nt; ++j) c[j] = (T)coef[j] * dtT;                                         \
        double* part = (double*)malloc(sizeof(double) * 2 * (size_t)n_chunks);                        \
        if (!part) return -2;                                                                         \
        for (int s = 0; s < n_seg; ++s) {                                                             \
            const int64_t c0 = segs[s].chunk_start;                                                   \
            const int64_t c1 = (s + 1 < n_seg) ? segs[s + 1].chunk_start : n_chunks;                  \
            const T rtol = (T)segs[s].rtol, atol = (T)segs[s].atol;                                   \
            _Pragma("omp parallel for schedule(static) if(c1 - c0 > 8)")                              \
            for (int64_t b = c0; b < c1; ++b) {                                                       \
                int64_t valid = segs[s].numel - (b - c0) * chunk;                                     \
                if (valid > chunk) valid = chunk;                                                     \
                if (valid < 0) valid = 0;                                                             \
                const int64_t base = b * chunk;                                                       \
                double acc = 0.0, bad = 0.0;                                                          \
                for (int64_t t = 0; t < valid; ++t) {                                                 \
                    const int64_t i = base + t;                                                       \
                    T e = k[0][i] * c[0];                                                             \
                    for (int j = 1; j < nt; ++j) e = e + k[j][i] * c[j];                              \
                    const T a0 = ABS(y0[i]), a1 = ABS(y1[i]);                                         \
                    const T tol = atol + rtol * MAX(a0, a1);                                          \
                    const T r = e / tol;                                                              \
                    if (scaled) scaled[i] = r;                                                        \
                    acc += (double)r * (double)r;                                                     \
                    if (!isfinite((double)y0[i]) || !isfinite((double)y1[i])) bad += 1.0;             \
                }                                                                                     \
                if (scaled && n_seg > 1)                                                              \
                    for (int64_t t = valid; t < chunk; ++t) scaled[base + t] = (T)0;                  \
                part[2 * b] = acc;                                                                    \
                part[2 * b + 1] = bad;                                                                \
            }                                                                                         \
            double total = 0.0, bad_total = 0.0;                                                      \
            for (int64_t b = c0; b < c1; ++b) {                                                       \
                total += part[2 * b];                                                                 \
                bad_total += part[2 * b + 1];                                                         \
            }                                                                                         \
            out_sumsq[s] = total;                                                                     \
            out_bad[s] = bad_total;                                                                   \
        }                                                                                             \
        free(part);                                                                                   \
        return 0;                                                                                     \
    }
DEF_ERROR(error_f32, float, fabsf, fmaxf)
DEF_ERROR(error_f64, double, fabs, fmax)

int oracle_error_norm(void* scaled_out, const void* y0, const void* y1, const void* const* k,
                      const double* coef, int n_terms, double dt, const oracle_segment* segs, int n_seg,
                      int64_t chunk, int64_t n_chunks, double* out_sumsq, double* out_nonfinite, int dtype) {
    if (!y0 || !y1 || !k || !coef || !segs || n_terms < 1 || n_terms > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32)
        return error_f32((float*)scaled_out, (const float*)y0, (const float*)y1, (const float* const*)k, coef, n_terms, dt,
                  segs, n_seg, chunk, n_chunks, out_sumsq, out_nonfinite);
    else if (dtype == ORACLE_F64)
        return error_f64((double*)scaled_out, (const double*)y0, (const double*)y1, (const double* const*)k, coef, n_terms,
                  dt, segs, n_seg, chunk, n_chunks, out_sumsq, out_nonfinite);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Initial-step norms, scale = atol + |y0| * rtol                  misc.py:53
 *   mode 0: sum (a/scale)^2 and sum (b/scale)^2   (d0, d1)       misc.py:55-56
 *   mode 1: sum ((a-b)/scale)^2                   (d2 * h0)      misc.py:68
 * ------------------------------------------------------------------------------------------------- */
#define DEF_INIT(NAME, T, ABS)                                                                        \
    static void NAME(int mode, const T* a, const T* b, const T* y, const oracle_segment* segs,        \
                     int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq, double* out_bad) {\
        for (int s = 0; s < n_seg; ++s) {                                                             \
            const int64_t c0 = segs[s].chunk_start;                                                   \
            const int64_t c1 = (s + 1 < n_seg) ? segs[s + 1].chunk_start : n_chunks;                  \
            const T rtol = (T)segs[s].rtol, atol = (T)segs[s].atol;                                   \
            double t0 = 0.0, t1 = 0.0, bad = 0.0;                                                     \
            for (int64_t blk = c0; blk < c1; ++blk) {                                                 \
                int64_t valid = segs[s].numel - (blk - c0) * chunk;                                   \
                if (valid > chunk) valid = chunk;                                                     \
                double p0 = 0.0, p1 = 0.0;                                                            \
                for (int64_t t = 0; t < valid; ++t) {                                                 \
                    const int64_t i = blk * chunk + t;                                                \
                    const T scale = atol + ABS(y[i]) * rtol;                                          \
                    if (mode == 0) {                                                                  \
                        const T r0 = a[i] / scale, r1 = b[i] / scale;                                 \
                        p0 += (double)r0 * (double)r0;                                                \
                        p1 += (double)r1 * (double)r1;                                                \
                    } else {                                                                          \
                        const T r0 = (a[i] - b[i]) / scale;                                           \
                        p0 += (double)r0 * (double)r0;                                                \
                    }                                                                                 \
                    if (!isfinite((double)y[i])) bad += 1.0;                                          \
                }                                                                                     \
                t0 += p0;                                                                             \
                t1 += p1;                                                                             \
            }                                                                                         \
            out_sumsq[s] = t0;                                                                        \
            if (mode == 0) out_sumsq[n_seg + s] = t1;                                                 \
            out_bad[s] = bad;                                                                         \
        }                                                                                             \
    }
DEF_INIT(init_f32, float, fabsf)
DEF_INIT(init_f64, double, fabs)

int oracle_init_norms(int mode, const void* a, const void* b, const void* yscale, const oracle_segment* segs,
                      int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq, double* out_nonfinite,
                      int dtype) {
    if ((mode != 0 && mode != 1) || !a || !b || !yscale || !segs) return -1;
    if (dtype == ORACLE_F32)
        init_f32(mode, (const float*)a, (const float*)b, (const float*)yscale, segs, n_seg, chunk, n_chunks,
                 out_sumsq, out_nonfinite);
    else if (dtype == ORACLE_F64)
        init_f64(mode, (const double*)a, (const double*)b, (const double*)yscale, segs, n_seg, chunk, n_chunks,
                 out_sumsq, out_nonfinite);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Dense output.  y_mid = y0 + sum_j (dt * mid_j) k_j               rk_common.py:365-366
 *   a = 2 dt (f1 - f0) - 8 (y1 + y0) + 16 y_mid                     interp.py:17
 *   b = dt (5 f0 - 3 f1) + 18 y0 + 14 y1 - 32 y_mid                 interp.py:18
 *   c = dt (f1 - 4 f0) - 11 y0 - 5 y1 + 16 y_mid                    interp.py:19
 *   d = dt f0 ; e = y0                                              interp.py:20-21
 *   p(x) = e + x d + x^2 c + x^3 b + x^4 a  (x_power *= x)          interp.py:42-47
 * fit_only != 0: out holds the five planes [e,d,c,b,a] (5*n elements).
 * ------------------------------------------------------------------------------------------------- */
#define DEF_DENSE(NAME, T)                                                                            \
    static void NAME(T* out, const T* y0, const T* y1, const T* f0, const T* f1, const T* const* k,   \
                     const double* coef, int nt, double dt_, double x_, int64_t n, int fit_only) {    \
        T c[ORACLE_MAX_TERMS];                                                                        \
        const T dt = (T)dt_, x = (T)x_;                                                               \
        for (int j = 0; j < nt; ++j) c[j] = (T)coef[j] * dt;                                          \
        const T two_dt = (T)2 * dt;                                                                   \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            T acc = k[0][i] * c[0];                                                                   \
            for (int j = 1; j < nt; ++j) acc = acc + k[j][i] * c[j];                                  \
            const T ymid = y0[i] + acc;                                                               \
            const T qa = (two_dt * (f1[i] - f0[i]) - (T)8 * (y1[i] + y0[i])) + (T)16 * ymid;          \
            const T qb = ((dt * ((T)5 * f0[i] - (T)3 * f1[i]) + (T)18 * y0[i]) + (T)14 * y1[i]) -     \
                         (T)32 * ymid;                                                                \
            const T qc = ((dt * (f1[i] - (T)4 * f0[i]) - (T)11 * y0[i]) - (T)5 * y1[i]) +             \
                         (T)16 * ymid;                                                                \
            const T qd = dt * f0[i];                                                                  \
            const T qe = y0[i];                                                                       \
            if (fit_only) {                                                                           \
                out[i] = qe;                                                                          \
                out[n + i] = qd;                                                                      \
                out[2 * n + i] = qc;                                                                  \
                out[3 * n + i] = qb;                                                                  \
                out[4 * n + i] = qa;                                                                  \
            } else {                                                                                  \
                T total = qe + x * qd;                                                                \
                T xp = x * x;                                                                         \
                total = total + xp * qc;                                                              \
                xp = xp * x;                                                                          \
                total = total + xp * qb;                                                              \
                xp = xp * x;                                                                          \
                total = total + xp * qa;                                                              \
                out[i] = total;                                                                       \
            }                                                                                         \
        }                                                                                             \
    }
DEF_DENSE(dense_f32, float)
DEF_DENSE(dense_f64, double)

static int dense_dispatch(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                          const void* const* k, const double* coef, int n_terms, double dt, double x, int64_t n,
                          int dtype, int fit_only) {
    if (!out || !y0 || !y1 || !f0 || !f1 || !k || !coef || n_terms < 1 || n_terms > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32)
        dense_f32((float*)out, (const float*)y0, (const float*)y1, (const float*)f0, (const float*)f1,
                  (const float* const*)k, coef, n_terms, dt, x, n, fit_only);
    else if (dtype == ORACLE_F64)
        dense_f64((double*)out, (const double*)y0, (const double*)y1, (const double*)f0, (const double*)f1,
                  (const double* const*)k, coef, n_terms, dt, x, n, fit_only);
    else return -1;
    return 0;
}

int oracle_dense_eval(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                      const void* const* k, const double* coef, int n_terms, double dt, double x, int64_t n,
                      int dtype) {
    return dense_dispatch(out, y0, y1, f0, f1, k, coef, n_terms, dt, x, n, dtype, 0);
}

int oracle_interp_fit(void* coeffs, const void* y0, const void* y1, const void* f0, const void* f1,
                      const void* const* k, const double* coef, int n_terms, double dt, int64_t n, int dtype) {
    return dense_dispatch(coeffs, y0, y1, f0, f1, k, coef, n_terms, dt, 0.0, n, dtype, 1);
}

/* ---------------------------------------------------------------------------------------------------
 * rk4, 3/8 rule                                                    rk_common.py:110-118, solvers.py:115
 * ------------------------------------------------------------------------------------------------- */
#define DEF_RK4(NAME, T)                                                                              \
    static void NAME(int stage, T* out, const T* y0, const T* k1, const T* k2, const T* k3,           \
                     const T* k4, double dt_, int64_t n) {                                            \
        const T dt = (T)dt_, third = (T)(1.0 / 3.0);                                                  \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            if (stage == 1) out[i] = y0[i] + (dt * k1[i]) * third;                                    \
            else if (stage == 2) out[i] = y0[i] + dt * (k2[i] - k1[i] * third);                       \
            else if (stage == 3) out[i] = y0[i] + dt * ((k1[i] - k2[i]) + k3[i]);                     \
            else out[i] = y0[i] + (((k1[i] + (T)3 * (k2[i] + k3[i])) + k4[i]) * dt) * (T)0.125;       \
        }                                                                                             \
    }
DEF_RK4(rk4_f32, float)
DEF_RK4(rk4_f64, double)

int oracle_rk4_38_stage(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3,
                        const void* k4, double dt, int64_t n, int dtype) {
    if (stage < 1 || stage > 4 || !out || !y0 || !k1) return -1;
    if (dtype == ORACLE_F32)
        rk4_f32(stage, (float*)out, (const float*)y0, (const float*)k1, (const float*)k2, (const float*)k3,
                (const float*)k4, dt, n);
    else if (dtype == ORACLE_F64)
        rk4_f64(stage, (double*)out, (const double*)y0, (const double*)k1, (const double*)k2, (const double*)k3,
                (const double*)k4, dt, n);
    else return -1;
    return 0;
}

/* y0 + slope * (y1 - y0)                                           solvers.py:175-181 */
int oracle_lerp(void* out, const void* y0, const void* y1, double slope, int64_t n, int dtype) {
    if (!out || !y0 || !y1) return -1;
    if (dtype == ORACLE_F32) {
        const float s = (float)slope;
        for (int64_t i = 0; i < n; ++i)
            ((float*)out)[i] = ((const float*)y0)[i] + s * (((const float*)y1)[i] - ((const float*)y0)[i]);
    } else if (dtype == ORACLE_F64) {
        for (int64_t i = 0; i < n; ++i)
            ((double*)out)[i] = ((const double*)y0)[i] + slope * (((const double*)y1)[i] - ((const double*)y0)[i]);
    } else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Low-order fixed-grid steps                                      rk_common.py:121-157, fixed_grid.py:32-60
 *   mode 0: out = y0 + dt * ((k0*w0 + k1*w1) + ...)   `y0 + dt * (k1*a31 + k2*a32)`; y0 + `dt * (k1*b1 + ...)`
 *   mode 1: out = y0 + (dt * k0) * w0                 `y0 + dt * k1 * a21`
 *   mode 2: out = (x0*w0 + x1*w1) + ...               cubic Hermite interpolation, solvers.py:166-173
 * ------------------------------------------------------------------------------------------------- */
#define DEF_FIXED(NAME, T)                                                                            \
    static void NAME(int mode, T* out, const T* y0, const T* const* k, const double* w, int nt,       \
                     double dt, int64_t n) {                                                          \
        T wT[8];                                                                                      \
        const T dtT = (T)dt;                                                                          \
        for (int j = 0; j < nt; ++j) wT[j] = (T)w[j];                                                 \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            if (mode == 1) { out[i] = y0[i] + (k[0][i] * dtT) * wT[0]; continue; }                    \
            T acc = k[0][i] * wT[0];                                                                  \
            for (int j = 1; j < nt; ++j) acc = acc + k[j][i] * wT[j];                                 \
            out[i] = (mode == 2) ? acc : y0[i] + acc * dtT;                                           \
        }                                                                                             \
    }
DEF_FIXED(fixed_f32, float)
DEF_FIXED(fixed_f64, double)

int oracle_fixed_stage(int mode, void* out, const void* y0, const void* const* k, const double* w, int n_terms,
                       double dt, int64_t n, int dtype) {
    if (!out || !y0 || !k || !w || (mode != 0 && mode != 1) || n_terms < 1 || n_terms > 4) return -1;
    if (mode == 1 && n_terms != 1) return -1;
    if (dtype == ORACLE_F32) fixed_f32(mode, (float*)out, (const float*)y0, (const float* const*)k, w, n_terms, dt, n);
    else if (dtype == ORACLE_F64) fixed_f64(mode, (double*)out, (const double*)y0, (const double* const*)k, w, n_terms, dt, n);
    else return -1;
    return 0;
}

int oracle_weighted_sum(void* out, const void* const* x, const double* w, int n_terms, int64_t n, int dtype) {
    if (!out || !x || !w || n_terms < 1 || n_terms > 8) return -1;
    if (dtype == ORACLE_F32) fixed_f32(2, (float*)out, NULL, (const float* const*)x, w, n_terms, 0.0, n);
    else if (dtype == ORACLE_F64) fixed_f64(2, (double*)out, NULL, (const double* const*)x, w, n_terms, 0.0, n);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Backward helpers (differentiable plain odeint): outs[m] = w_m * g ; out[m] = <g, x_m> (fp64).
 * Restates what autograd does for the reference's eager `k * (beta * dt)` / `sum` ops (rk_common.py:79).
 * ------------------------------------------------------------------------------------------------- */
#define DEF_SCALE(NAME, T)                                                                            \
    static void NAME(T* const* outs, const T* g, const double* w, int nt, int64_t n) {                \
        for (int j = 0; j < nt; ++j) {                                                                \
            const T wj = (T)w[j];                                                                     \
            T* o = outs[j];                                                                           \
            _Pragma("omp parallel for schedule(static) if(n > 16384)")                                \
            for (int64_t i = 0; i < n; ++i) o[i] = g[i] * wj;                                         \
        }                                                                                             \
    }
DEF_SCALE(scale_f32, float)
DEF_SCALE(scale_f64, double)

int oracle_scale_many(void* const* outs, const void* g, const double* w, int n_out, int64_t n, int dtype) {
    if (!outs || !g || !w || n_out < 1 || n_out > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32) scale_f32((float* const*)outs, (const float*)g, w, n_out, n);
    else if (dtype == ORACLE_F64) scale_f64((double* const*)outs, (const double*)g, w, n_out, n);
    else return -1;
    return 0;
}

#define DEF_DOTS(NAME, T)                                                                             \
    static void NAME(const T* g, const T* const* x, int nt, int64_t n, double* out) {                 \
        for (int j = 0; j < nt; ++j) {                                                                \
            const int64_t chunk = 4096;                                                               \
            const int64_t nc = (n + chunk - 1) / chunk;                                               \
            double total = 0.0;                                                                       \
            for (int64_t c = 0; c < nc; ++c) {                                                        \
                double acc = 0.0;                                                                     \
                const int64_t hi = (c + 1) * chunk < n ? (c + 1) * chunk : n;                         \
                for (int64_t i = c * chunk; i < hi; ++i) acc += (double)g[i] * (double)x[j][i];       \
                total += acc;                                                                         \
            }                                                                                         \
            out[j] = total;                                                                           \
        }                                                                                             \
    }
DEF_DOTS(dots_f32, float)
DEF_DOTS(dots_f64, double)

int oracle_multi_dot(const void* g, const void* const* x, int n_x, int64_t n, double* out, int dtype) {
    if (!g || !x || !out || n_x < 1 || n_x > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32) dots_f32((const float*)g, (const float* const*)x, n_x, n, out);
    else if (dtype == ORACLE_F64) dots_f64((const double*)g, (const double* const*)x, n_x, n, out);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Fused end-of-step pair (same arithmetic as oracle_stage_combine + oracle_error_norm, split differently):
 *   combine_err:        out = y0 + sum_j c_j k_j ; err_out = (e_0 k_0 + e_1 k_1) + ...     rk_common.py:79/85, :89
 *   error_norm_partial: err = (partial + c_0 k_0) + ... ; tol, sums and census as above     misc.py:80-82
 * ------------------------------------------------------------------------------------------------- */
#define DEF_COMBINE_ERR(NAME, T)                                                                      \
    static void NAME(T* out, T* err_out, const T* y0, const T* const* k, const double* coef,          \
                     const double* ecoef, int nt, double dt, int64_t n) {                             \
        T c[ORACLE_MAX_TERMS], e[ORACLE_MAX_TERMS];                                                   \
        const T dtT = (T)dt;                                                                          \
        for (int j = 0; j < nt; ++j) { c[j] = (T)coef[j] * dtT; e[j] = (T)ecoef[j] * dtT; }           \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            T acc = k[0][i] * c[0];                                                                   \
            T err = k[0][i] * e[0];                                                                   \
            for (int j = 1; j < nt; ++j) { acc = acc + k[j][i] * c[j]; err = err + k[j][i] * e[j]; }  \
            out[i] = y0[i] + acc;                                                                     \
            err_out[i] = err;                                                                         \
        }                                                                                             \
    }
DEF_COMBINE_ERR(combine_err_f32, float)
DEF_COMBINE_ERR(combine_err_f64, double)

int oracle_stage_combine_err(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                             const double* err_coef, int n_terms, double dt, int64_t n, int dtype) {
    if (!out || !err_out || !y0 || !k || !coef || !err_coef || n_terms < 1 || n_terms > ORACLE_MAX_TERMS) return -1;
    if (dtype == ORACLE_F32)
        combine_err_f32((float*)out, (float*)err_out, (const float*)y0, (const float* const*)k, coef, err_coef, n_terms, dt, n);
    else if (dtype == ORACLE_F64)
        combine_err_f64((double*)out, (double*)err_out, (const double*)y0, (const double* const*)k, coef, err_coef, n_terms, dt, n);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Carried partial sums (twin of tdeq_stage_combine_multi): one pass over the stages k_0..k_{nt-1}, n_out outputs,
 *   s_o = [acc_in +] sum over the set bits j of mask_o (ascending) of fl_T(fl_T(coef_o[j])*fl_T(dt)) * k_j
 *   out_o = add_y0_o ? y0 + s_o : s_o
 * i.e. the SAME left-to-right sums as rk_common.py:79 / :89 — a prefix of a later row's sum is formed while an
 * earlier row is, and that later row continues it (acc_in, output 0 only).  tests/test_carry.py checks that rows
 * formed this way equal oracle_stage_combine / oracle_stage_combine_err bit for bit.
 * ------------------------------------------------------------------------------------------------- */
#define ORACLE_MAX_MULTI_OUT 4
typedef struct oracle_multi_out {
    void* out;
    double coef[ORACLE_MAX_TERMS];
    uint32_t mask;
    int32_t add_y0;
} oracle_multi_out;

#define DEF_COMBINE_MULTI(NAME, T)                                                                    \
    static void NAME(const oracle_multi_out* outs, int n_out, const T* y0, const T* acc_in,           \
                     const T* const* k, int nt, double dt, int64_t n) {                               \
        T c[ORACLE_MAX_MULTI_OUT][ORACLE_MAX_TERMS];                                                  \
        const T dtT = (T)dt;                                                                          \
        for (int o = 0; o < n_out; ++o)                                                               \
            for (int j = 0; j < nt; ++j) c[o][j] = (T)outs[o].coef[j] * dtT;                          \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            for (int o = 0; o < n_out; ++o) {                                                         \
                int started = (o == 0 && acc_in != NULL);                                             \
                T s = started ? acc_in[i] : (T)0;                                                     \
                for (int j = 0; j < nt; ++j) {                                                        \
                    if (!((outs[o].mask >> j) & 1u)) continue;                                        \
                    const T p = k[j][i] * c[o][j];                                                    \
                    s = started ? s + p : p;                                                          \
                    started = 1;                                                                      \
                }                                                                                     \
                ((T*)outs[o].out)[i] = outs[o].add_y0 ? y0[i] + s : s;                                \
            }                                                                                         \
        }                                                                                             \
    }
DEF_COMBINE_MULTI(combine_multi_f32, float)
DEF_COMBINE_MULTI(combine_multi_f64, double)

int oracle_stage_combine_multi(const oracle_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                               const void* const* k, int n_terms, double dt, int64_t n, int dtype) {
    if (!outs || !y0 || !k || n_terms < 1 || n_terms > ORACLE_MAX_TERMS || n_out < 1 || n_out > ORACLE_MAX_MULTI_OUT)
        return -1;
    for (int o = 0; o < n_out; ++o) if (!outs[o].out || !outs[o].mask) return -1;
    if (dtype == ORACLE_F32)
        combine_multi_f32(outs, n_out, (const float*)y0, (const float*)acc_in, (const float* const*)k, n_terms, dt, n);
    else if (dtype == ORACLE_F64)
        combine_multi_f64(outs, n_out, (const double*)y0, (const double*)acc_in, (const double* const*)k, n_terms, dt, n);
    else return -1;
    return 0;
}

#define DEF_ERROR_PARTIAL(NAME, T, ABS, MAX)                                                          \
    static int NAME(const T* partial, const T* y0, const T* y1, const T* const* k, const double* coef,\
                    int nt, double dt, const oracle_segment* segs, int n_seg, int64_t chunk,          \
                    int64_t n_chunks, double* out_sumsq, double* out_bad) {                           \
        T c[2] = {0, 0};                                                                              \
        const T dtT = (T)dt;                                                                          \
        for (int j = 0; j < nt; ++j) c[j] = (T)coef[j] * dtT;                                         \
        double* part = (double*)malloc(sizeof(double) * 2 * (size_t)n_chunks);                        \
        if (!part) return -2;                                                                         \
        for (int s = 0; s < n_seg; ++s) {                                                             \
            const int64_t c0 = segs[s].chunk_start;                                                   \
            const int64_t c1 = (s + 1 < n_seg) ? segs[s + 1].chunk_start : n_chunks;                  \
            const T rtol = (T)segs[s].rtol, atol = (T)segs[s].atol;                                   \
            _Pragma("omp parallel for schedule(static) if(c1 - c0 > 8)")                              \
            for (int64_t b = c0; b < c1; ++b) {                                                       \
                int64_t valid = segs[s].numel - (b - c0) * chunk;                                     \
                if (valid > chunk) valid = chunk;                                                     \
                if (valid < 0) valid = 0;                                                             \
                const int64_t base = b * chunk;                                                       \
                double acc = 0.0, bad = 0.0;                                                          \
                for (int64_t t = 0; t < valid; ++t) {                                                 \
                    const int64_t i = base + t;                                                       \
                    T e = partial[i];                                                                 \
                    for (int j = 0; j < nt; ++j) e = e + k[j][i] * c[j];                              \
                    const T a0 = ABS(y0[i]), a1 = ABS(y1[i]);                                         \
                    const T tol = atol + rtol * MAX(a0, a1);                                          \
                    const T r = e / tol;                                                              \
                    acc += (double)r * (double)r;                                                     \
                    if (!isfinite((double)y0[i]) || !isfinite((double)y1[i])) bad += 1.0;             \
                }                                                                                     \
                part[2 * b] = acc;                                                                    \
                part[2 * b + 1] = bad;                                                                \
            }                                                                                         \
            double total = 0.0, bad_total = 0.0;                                                      \
            for (int64_t b = c0; b < c1; ++b) {                                                       \
                total += part[2 * b];                                                                 \
                bad_total += part[2 * b + 1];                                                         \
            }                                                                                         \
            out_sumsq[s] = total;                                                                     \
            out_bad[s] = bad_total;                                                                   \
        }                                                                                             \
        free(part);                                                                                   \
        return 0;                                                                                     \
    }
DEF_ERROR_PARTIAL(error_partial_f32, float, fabsf, fmaxf)
DEF_ERROR_PARTIAL(error_partial_f64, double, fabs, fmax)

int oracle_error_norm_partial(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                              const double* coef, int n_terms, double dt, const oracle_segment* segs, int n_seg,
                              int64_t chunk, int64_t n_chunks, double* out_sumsq, double* out_nonfinite, int dtype) {
    if (!err_partial || !y0 || !y1 || !segs || n_terms < 0 || n_terms > 2) return -1;
    if (dtype == ORACLE_F32)
        return error_partial_f32((const float*)err_partial, (const float*)y0, (const float*)y1, (const float* const*)k,
                                 coef, n_terms, dt, segs, n_seg, chunk, n_chunks, out_sumsq, out_nonfinite);
    else if (dtype == ORACLE_F64)
        return error_partial_f64((const double*)err_partial, (const double*)y0, (const double*)y1,
                                 (const double* const*)k, coef, n_terms, dt, segs, n_seg, chunk, n_chunks, out_sumsq,
                                 out_nonfinite);
    return -1;
}

/* ---------------------------------------------------------------------------------------------------
 * Step controller + next trial step's stage times (CPU twin of tdeq_error_norm_partial_ctrl /
 * tdeq_stage_combine_sel in include/tdeq_hip.h).  Restates, on host doubles:
 *   error ratio  = max over segments of sqrt(mean((err/tol)^2)), in T         misc.py:22-33, 80-82
 *   accept_step  = ratio <= 1; dt > max_step -> False; dt <= min_step -> True  rk_common.py:324-330
 *   dt_next      = _optimal_step_size(...).clamp(min_step, max_step)            misc.py:85-95, rk_common.py:353-354
 *   next state   = (t1, y1, f1) if accepted else (t0, y0, f0)                   rk_common.py:335-352
 *   next trial   : dt := min_step if non-finite; clamp; ti = t1 (Perturb.PREV) if alpha_i == 1 else
 *                  t0 + alpha_i * dt, all in T = y0.abs().dtype                 rk_common.py:268-271, 60-78
 *   Perturb.PREV : nextafter(t, t - 1) in T; decreasing-time solves negate t    misc.py:158-165, 174-197
 * ------------------------------------------------------------------------------------------------- */
#define ORACLE_MAX_STAGE_TIMES 16
typedef struct oracle_step_ctrl {
    double t0, dt, safety, ifactor, dfactor, exponent, min_step, max_step, time_sign;
    double alpha[ORACLE_MAX_STAGE_TIMES];
    uint32_t alpha_is_one;
    int32_t n_times;
    int32_t n_norm_seg;
    int32_t leading_abs;
} oracle_step_ctrl;

static double o_nan_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }
static double o_nan_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }
static double o_clamp(double x, double lo, double hi) {
    if (isnan(x)) return x;
    const double m = x > lo ? x : lo;
    return m < hi ? m : hi;
}

static void controller(const oracle_step_ctrl* c, const oracle_segment* segs, int n_seg, const double* sumsq,
                       int is_f32, double* out_ctrl, double* ctrl_dev, void* next_times) {
    double val = 0.0;
    for (int s = 0; s < c->n_norm_seg && s < n_seg; ++s) {
        if (segs[s].numel == 0) continue;
        val = o_nan_max(val, sqrt(sumsq[s] / (double)segs[s].numel));
    }
    const double ratio = is_f32 ? (double)(float)val : val;
    int accept = ratio <= 1.0;
    if (c->dt > c->max_step) accept = 0;
    if (c->dt <= c->min_step) accept = 1;
    double dt_next;
    if (ratio == 0.0) {
        dt_next = c->dt * c->ifactor;
    } else {
        const double dfactor = ratio < 1.0 ? 1.0 : c->dfactor;
        const double scaled = c->safety / pow(ratio, c->exponent);
        dt_next = c->dt * o_nan_min(c->ifactor, o_nan_max(scaled, dfactor));
    }
    dt_next = o_clamp(dt_next, c->min_step, c->max_step);
    const double t0n = accept ? c->t0 + c->dt : c->t0;
    double dtn = dt_next;
    if (!isfinite(dtn)) dtn = c->min_step;
    dtn = o_clamp(dtn, c->min_step, c->max_step);
    if (is_f32) {
        float* dst = (float*)next_times;
        const float t0T = (float)t0n, dtT = (float)dtn, t1T = (float)(t0n + dtn), sg = (float)c->time_sign;
        for (int i = 0; i < c->n_times; ++i) {
            float tt;
            if ((c->alpha_is_one >> i) & 1u) tt = nextafterf(t1T, t1T - 1.0f);
            else { const float prod = (float)c->alpha[i] * dtT; tt = t0T + prod; }
            dst[i] = sg * tt;
        }
        ctrl_dev[1] = (double)(float)dtn * c->time_sign;
    } else {
        double* dst = (double*)next_times;
        const double t0T = t0n, dtT = dtn, t1T = t0n + dtn;
        for (int i = 0; i < c->n_times; ++i) {
            double tt;
            if ((c->alpha_is_one >> i) & 1u) tt = nextafter(t1T, t1T - 1.0);
            else { const double prod = c->alpha[i] * dtT; tt = t0T + prod; }
            dst[i] = c->time_sign * tt;
        }
        ctrl_dev[1] = dtn * c->time_sign;
    }
    ctrl_dev[0] = accept ? 1.0 : 0.0;
    out_ctrl[0] = accept ? 1.0 : 0.0;
    out_ctrl[1] = dt_next;
    out_ctrl[2] = ratio;
    out_ctrl[3] = t0n;
}

int oracle_error_norm_partial_ctrl(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                                   const double* coef, int n_terms, double dt, const oracle_segment* segs, int n_seg,
                                   int64_t chunk, int64_t n_chunks, double* out_sumsq, double* out_nonfinite,
                                   const oracle_step_ctrl* ctrl, double* out_ctrl, double* ctrl_dev, void* next_times,
                                   int dtype) {
    if (!ctrl || !out_ctrl || !ctrl_dev || !next_times) return -1;   /* any number of segments (the host loops) */
    if (ctrl->n_times < 1 || ctrl->n_times > ORACLE_MAX_STAGE_TIMES) return -1;
    const int e = oracle_error_norm_partial(err_partial, y0, y1, k, coef, n_terms, dt, segs, n_seg, chunk, n_chunks,
                                            out_sumsq, out_nonfinite, dtype);
    if (e) return e;
    controller(ctrl, segs, n_seg, out_sumsq, dtype == ORACLE_F32, out_ctrl, ctrl_dev, next_times);
    return 0;
}

/* The controller alone on given per-segment sums (kernel-parity tests feed it the device's sums). */
int oracle_step_controller(const oracle_segment* segs, int n_seg, const double* sumsq, const oracle_step_ctrl* ctrl,
                           double* out_ctrl, double* ctrl_dev, void* next_times, int dtype) {
    if (!segs || !sumsq || !ctrl || !out_ctrl || !ctrl_dev || !next_times) return -1;
    controller(ctrl, segs, n_seg, sumsq, dtype == ORACLE_F32, out_ctrl, ctrl_dev, next_times);
    return 0;
}

/* out = y + fl_T(fl_T(coef) * T(dt')) * f on the pair the controller selected (rk_common.py:79 with i = 0). */
int oracle_stage_combine_sel(void* out, const void* y_acc, const void* f_acc, const void* y_rej, const void* f_rej,
                             double coef, const double* ctrl_dev, int64_t n, int dtype) {
    if (!out || !y_acc || !f_acc || !y_rej || !f_rej || !ctrl_dev) return -1;
    const int accept = ctrl_dev[0] != 0.0;
    const void* y = accept ? y_acc : y_rej;
    const void* f = accept ? f_acc : f_rej;
    const void* ks[1] = {f};
    return oracle_stage_combine(out, y, ks, &coef, 1, ctrl_dev[1], n, dtype);
}

/* ---------------------------------------------------------------------------------------------------
 * Flat segmented state from the pieces of a tuple-valued func output (CPU twin of tdeq_pack_segments):
 * torch.cat of the pieces (misc.py:137-145); for the adjoint's augmented dynamics also `-adj_y`
 * (adjoint.py:95), zeros for absent gradients (adjoint.py:99-103) and _ReverseFunc's sign (misc.py:158-165).
 * ------------------------------------------------------------------------------------------------- */
int oracle_pack_segments(void* out, const void* const* src, const int64_t* chunk_start, const int64_t* numel,
                         const double* scale, int n_seg, int64_t chunk, int64_t n_chunks, int dtype) {
    if (!out || !src || !chunk_start || !numel || !scale || n_seg < 1 || n_seg > 16) return -1;
    for (int s = 0; s < n_seg; ++s) {
        const int64_t begin = chunk_start[s] * chunk;
        const int64_t end = ((s + 1 < n_seg) ? chunk_start[s + 1] : n_chunks) * chunk;
        for (int64_t i = begin; i < end; ++i) {
            const int64_t t = i - begin;
            const int live = src[s] && t < numel[s];
            if (dtype == ORACLE_F32) ((float*)out)[i] = live ? ((const float*)src[s])[t] * (float)scale[s] : 0.0f;
            else ((double*)out)[i] = live ? ((const double*)src[s])[t] * scale[s] : 0.0;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * y0/scale, f0/scale, (f1 - f0)/scale of the initial-step heuristic, materialised for a user norm callable
 * (misc.py:53-56, 68; CPU twin of tdeq_init_scaled).
 * ------------------------------------------------------------------------------------------------- */
#define DEF_INIT_SCALED(NAME, T, ABS)                                                                 \
    static void NAME(int mode, const T* a, const T* b, const T* y, const oracle_segment* segs, int n_seg, \
                     int64_t chunk, int64_t n_chunks, T* out0, T* out1) {                             \
        for (int s = 0; s < n_seg; ++s) {                                                             \
            const int64_t begin = segs[s].chunk_start * chunk;                                        \
            const int64_t end = ((s + 1 < n_seg) ? segs[s + 1].chunk_start : n_chunks) * chunk;       \
            const T rtol = (T)segs[s].rtol, atol = (T)segs[s].atol;                                   \
            for (int64_t i = begin; i < end; ++i) {                                                   \
                const int live = (i - begin) < segs[s].numel;                                         \
                if (!live) { if (n_seg > 1) { out0[i] = 0; if (mode == 0) out1[i] = 0; } continue; }  \
                const T scale = atol + ABS(y[i]) * rtol;                                              \
                if (mode == 0) { out0[i] = a[i] / scale; out1[i] = b[i] / scale; }                    \
                else out0[i] = (a[i] - b[i]) / scale;                                                 \
            }                                                                                         \
        }                                                                                             \
    }
DEF_INIT_SCALED(init_scaled_f32, float, fabsf)
DEF_INIT_SCALED(init_scaled_f64, double, fabs)

int oracle_init_scaled(int mode, const void* a, const void* b, const void* yscale, const oracle_segment* segs, int n_seg,
                       int64_t chunk, int64_t n_chunks, void* out0, void* out1, int dtype) {
    if ((mode != 0 && mode != 1) || !a || !b || !yscale || !segs || !out0 || (mode == 0 && !out1)) return -1;
    if (dtype == ORACLE_F32)
        init_scaled_f32(mode, (const float*)a, (const float*)b, (const float*)yscale, segs, n_seg, chunk, n_chunks,
                        (float*)out0, (float*)out1);
    else if (dtype == ORACLE_F64)
        init_scaled_f64(mode, (const double*)a, (const double*)b, (const double*)yscale, segs, n_seg, chunk, n_chunks,
                        (double*)out0, (double*)out1);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * Adams–Bashforth predictor and the constant part of the Adams–Moulton corrector (CPU twin of
 * tdeq_adams_predict):
 *   dy = _dot_product(dt * bashforth_coeffs, prev_f).type_as(y0)            fixed_adams.py:205
 *        (`dt * bashforth_coeffs` is an fp64 vector; each entry is rounded to T when it multiplies f_j; Python's
 *        `sum` adds the products left to right starting from the int 0)                fixed_adams.py:160-161
 *   delta = dt * _dot_product(moulton_coeffs[1:], prev_f).type_as(y0)      fixed_adams.py:210 (dt rounded to T)
 *   y0 + dy                                                                  fixed_adams.py:213, solvers.py:115
 * The caller passes cb_j = dt*b_j (fp64) and cm_j = m_{j+1}.
 * ------------------------------------------------------------------------------------------------- */
#define DEF_ADAMS_PREDICT(NAME, T)                                                                    \
    static void NAME(T* y_out, T* dy_out, T* delta_out, const T* y0, const T* const* f,               \
                     const double* cb, const double* cm, int nt, double dt, int64_t n) {              \
        T b[ORACLE_MAX_TERMS], m[ORACLE_MAX_TERMS];                                                   \
        for (int j = 0; j < nt; ++j) { b[j] = (T)cb[j]; m[j] = cm ? (T)cm[j] : (T)0; }                \
        const T dtT = (T)dt;                                                                          \
        _Pragma("omp parallel for schedule(static) if(n > 16384)")                                    \
        for (int64_t i = 0; i < n; ++i) {                                                             \
            T dy = b[0] * f[0][i];                                                                    \
            for (int j = 1; j < nt; ++j) dy = dy + b[j] * f[j][i];                                    \
            y_out[i] = y0[i] + dy;                                                                    \
            if (dy_out) {                                                                             \
                T sm = m[0] * f[0][i];                                                                \
                for (int j = 1; j < nt; ++j) sm = sm + m[j] * f[j][i];                                \
                dy_out[i] = dy;                                                                       \
                delta_out[i] = dtT * sm;                                                              \
            }                                                                                         \
        }                                                                                             \
    }
DEF_ADAMS_PREDICT(adams_predict_f32, float)
DEF_ADAMS_PREDICT(adams_predict_f64, double)

int oracle_adams_predict(void* y_out, void* dy_out, void* delta_out, const void* y0, const void* const* f_hist,
                         const double* cb, const double* cm, int n_terms, double dt, int64_t n, int dtype) {
    if (!y_out || !y0 || !f_hist || !cb || n_terms < 1 || n_terms > ORACLE_MAX_TERMS) return -1;
    if ((dy_out == NULL) != (delta_out == NULL) || (dy_out && !cm)) return -1;
    if (dtype == ORACLE_F32)
        adams_predict_f32((float*)y_out, (float*)dy_out, (float*)delta_out, (const float*)y0,
                          (const float* const*)f_hist, cb, cm, n_terms, dt, n);
    else if (dtype == ORACLE_F64)
        adams_predict_f64((double*)y_out, (double*)dy_out, (double*)delta_out, (const double*)y0,
                          (const double* const*)f_hist, cb, cm, n_terms, dt, n);
    else return -1;
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * One Adams–Moulton corrector iteration + `_has_converged` (CPU twin of tdeq_adams_correct):
 *   dy = (dt * moulton_coeffs[0] * f).type_as(y0) + delta                   fixed_adams.py:214
 *        (`dt * moulton_coeffs[0]` is a 0-dim fp64 product — the caller passes it as c — rounded to T against f)
 *   y0 + dy                                                                  fixed_adams.py:213 (next iteration) / y1
 *   error_ratio = linf(|dy_old - dy| / (atol + rtol*max(|dy_old|,|dy|))) < 1 fixed_adams.py:189-192, misc.py:80-82
 * reported as the per-segment COUNT of elements with not(ratio < 1): the boolean is "all counts are 0".
 * ------------------------------------------------------------------------------------------------- */
#define DEF_ADAMS_CORRECT(NAME, T, ABS, MAX)                                                          \
    static void NAME(T* y_out, T* dy_out, const T* f, const T* delta, const T* dy_old, const T* y0,   \
                     double c, int compute, const oracle_segment* segs, int n_seg, int64_t chunk,     \
                     int64_t n_chunks, int64_t n, double* out_count, double* out_bad) {               \
        const T cT = (T)c;                                                                            \
        if (compute) {                                                                                \
            _Pragma("omp parallel for schedule(static) if(n > 16384)")                                \
            for (int64_t i = 0; i < n; ++i) {                                                         \
                const T d = cT * f[i] + delta[i];                                                     \
                dy_out[i] = d;                                                                        \
                y_out[i] = y0[i] + d;                                                                 \
            }                                                                                         \
        }                                                                                             \
        for (int s = 0; s < n_seg; ++s) {                                                             \
            const int64_t base = segs[s].chunk_start * chunk;                                         \
            const T rtol = (T)segs[s].rtol, atol = (T)segs[s].atol;                                   \
            double cnt = 0.0, bad = 0.0;                                                              \
            _Pragma("omp parallel for schedule(static) if(segs[s].numel > 16384) reduction(+:cnt,bad)")  \
            for (int64_t t = 0; t < segs[s].numel; ++t) {                                             \
                const T d0 = dy_old[base + t], d1 = dy_out[base + t];                                 \
                const T e = ABS(d0 - d1);                                                             \
                const T tol = atol + rtol * MAX(ABS(d0), ABS(d1));                                    \
                const T r = e / tol;                                                                  \
                if (!(r < (T)1)) cnt += 1.0;                                                          \
                if (!isfinite((double)d1)) bad += 1.0;                                                \
            }                                                                                         \
            out_count[s] = cnt;                                                                       \
            out_bad[s] = bad;                                                                         \
        }                                                                                             \
        (void)n_chunks;                                                                               \
    }
DEF_ADAMS_CORRECT(adams_correct_f32, float, fabsf, fmaxf)
DEF_ADAMS_CORRECT(adams_correct_f64, double, fabs, fmax)

int oracle_adams_correct(void* y_out, void* dy_out, const void* f, const void* delta, const void* dy_old,
                         const void* y0, double c, int compute, const oracle_segment* segs, int n_seg, int64_t chunk,
                         int64_t n_chunks, int64_t n, double* out_count, double* out_nonfinite, int dtype) {
    if (!dy_out || !dy_old || !segs || !out_count || !out_nonfinite) return -1;
    if (compute && (!y_out || !f || !delta || !y0)) return -1;
    if (dtype == ORACLE_F32)
        adams_correct_f32((float*)y_out, (float*)dy_out, (const float*)f, (const float*)delta, (const float*)dy_old,
                          (const float*)y0, c, compute, segs, n_seg, chunk, n_chunks, n, out_count, out_nonfinite);
    else if (dtype == ORACLE_F64)
        adams_correct_f64((double*)y_out, (double*)dy_out, (const double*)f, (const double*)delta,
                          (const double*)dy_old, (const double*)y0, c, compute, segs, n_seg, chunk, n_chunks, n,
                          out_count, out_nonfinite);
    else return -1;
    return 0;
}
