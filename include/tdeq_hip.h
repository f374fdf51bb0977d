/*
 * tdeq_hip.h — C-ABI of libtdeq_hip.so, the MI355X (gfx950) explicit Runge–Kutta hot path.
 *
 * The reference (rtqichen/torchdiffeq v0.2.5) has no native boundary: every function below replaces a
 * group of eager ATen calls inside one reference Python function (cited per entry point as
 * torchdiffeq/_impl/<file>:<lines>).  The host solver (torchdiffeq_amd/solvers.py) is the only caller.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - every entry point returns an int: 0 on success, a hipError_t (>0) from the launch, or a
 *     TDEQ_E* (<0) for argument errors.  Nothing throws or aborts across the ABI.
 *   - all device buffers are BORROWED: owned by the caller, contiguous, kept alive until the stream
 *     reaches the launch.  No allocation, no hipDeviceSynchronize, no global mutable state.
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - `dtype`: TDEQ_F32 or TDEQ_F64 = element type T of every state-sized buffer.
 *   - coefficient arithmetic follows the reference's rounding: c_j = fl_T(fl_T(coef_j) * fl_T(dt)),
 *     products and sums in T, no FMA contraction (rk_common.py:79,89,201-205; interp.py:17-21).
 *   - `dt` may be negative: a solve in decreasing time passes sign*dt and keeps the raw func outputs
 *     in k_j, which is bit-identical to the reference's `_ReverseFunc(mul=-1)` wrapper (misc.py:158-165).
 *   - host arrays (`k`, `coef`, segment tables) are read during the call and copied into the kernel
 *     argument block; they need not outlive the call.
 *   - state buffers whose address is 16-byte aligned take the 16 B/lane vector path; otherwise a
 *     scalar path is used (same results).
 *
 * Segments.  A state vector may consist of several logical segments (tuple states, the adjoint's
 * [y | adj_y | params | vjp_t]).  Segment s starts at element chunk_start[s]*chunk and holds numel[s]
 * valid elements; the space up to the next segment start is padding that the norm kernels ignore.
 * `chunk` (elements) must be a multiple of TDEQ_CHUNK_QUANTUM.  n_seg==1 with chunk_start={0}
 * is the plain unpadded tensor case.
 */
#ifndef TDEQ_HIP_H
#define TDEQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDEQ_ABI_VERSION 21
#define TDEQ_F32 0
#define TDEQ_F64 1
/* interleaved (re, im) complex states — accepted by the NORM entry points only (tdeq_error_norm, tdeq_error_norm_partial[_ctrl],
 * tdeq_init_norms, tdeq_init_scaled): segment tables, chunk and counts are then in complex elements and T below is the
 * real type; every other entry point is linear with real coefficients and takes the state's real view (2n elements of
 * TDEQ_F32 / TDEQ_F64).  |z| = hypot(re, im), z / real = z * (1 / real): torchdiffeq/_impl/misc.py:80-82 as ATen rounds it. */
#define TDEQ_C64 2
#define TDEQ_C128 3
/* bfloat16 / float16 states (r05).  The reference integrates them in their own precision (misc.py:185-187,
 * rk_common.py:61-65): every ATen op computes in float32 and rounds its result to the storage type, a `torch.sum` over a
 * tableau row accumulates the rounded products in float32 and rounds once.  Accepted by the entry points of the
 * host-driven step — tdeq_stage_combine, tdeq_stage_combine_fill, tdeq_stage_combine_err, tdeq_error_norm,
 * tdeq_init_norms, tdeq_init_scaled, tdeq_dense_eval, tdeq_dense_eval_multi, tdeq_interp_fit, tdeq_rk4_38_stage,
 * tdeq_lerp, tdeq_fixed_stage, tdeq_weighted_sum — with exactly that rounding (tdeq_kernels_lp.hpp), and by the
 * look-ahead pair tdeq_error_norm_partial_ctrl (err_partial MUST be NULL: the whole error row in `k` / `coef`, 1..14
 * terms — a row is rounded once, there is no partial sum to continue; the controller forms the ratio, the next step and
 * its stage times in the state's type; `state_in_dev` as for fp32) + tdeq_stage_combine_sel, and by tdeq_stage_combine_dev
 * (captured steps; err_out must be NULL); every other entry point returns TDEQ_EINVAL for them.  Scalars: `dt`, tableau weights, `slope`, tolerances are rounded to the
 * storage type where the reference holds them as 0-dim tensors of the state's type or as FIRST operands, and taken at
 * float32 where ATen takes a Python number as SECOND operand of `*` (rk4's 1/3, `* dt`; tdeq_init_norms' rtol).
 * The norm entry points report per segment the sum of fl(|q|^2) — for a segment of ONE element |q| itself (the adjoint's
 * norms take their time component as `t.abs()`, adjoint.py:250, and the caller squares it with the same rounding). */
#define TDEQ_BF16 4
#define TDEQ_F16 5
#define TDEQ_MAX_TERMS 14      /* dopri8: 13 stages + FSAL slot (dopri8.py:5-70) */
#define TDEQ_MAX_SUM_TERMS 8    /* tdeq_weighted_sum */
#define TDEQ_MAX_DENSE_OUTPUTS 16 /* tdeq_dense_eval_multi */
#define TDEQ_MAX_SEGMENTS 4096
#define TDEQ_INLINE_SEGMENTS 16  /* segment tables up to this size travel in the kernel arguments */
#define TDEQ_CHUNK_QUANTUM 1024

#define TDEQ_EINVAL (-1)       /* bad argument (null pointer, n_terms out of range, bad dtype ...) */
#define TDEQ_EWORKSPACE (-2)   /* workspace too small */

/* Segment table entry (host memory, passed to the norm entry points). */
typedef struct tdeq_segment {
    int64_t chunk_start;   /* first chunk of the segment                        */
    int64_t numel;         /* valid elements in the segment                     */
    double rtol;           /* per-segment tolerances (misc.py:115-123 tuple tol) */
    double atol;
} tdeq_segment;

/*
 * Step-controller parameters of ONE adaptive trial step (host memory, copied into the kernel arguments).
 * Everything is a host double, as in the host loop (torchdiffeq_amd/solvers.py); the reference keeps the same
 * quantities as 0-dim fp64 device tensors (rk_common.py:186-194).
 */
#define TDEQ_MAX_STAGE_TIMES 16
typedef struct tdeq_step_ctrl {
    double t0;          /* start of the trial step, solver (ascending) time                                  */
    double dt;          /* its size, > 0, after the min_step / max_step clamp (rk_common.py:268-271)          */
    double safety;      /* misc.py:85-95 `_optimal_step_size` parameters (rk_common.py:172-174)               */
    double ifactor;
    double dfactor;
    double exponent;    /* 1 / order                                                                          */
    double min_step;
    double max_step;
    double time_sign;   /* +1, or -1 for a decreasing-time solve: user time = sign * solver time (misc.py:158-165) */
    double alpha[TDEQ_MAX_STAGE_TIMES];   /* stage abscissae of the tableau, already rounded to T (rk_common.py:201) */
    uint32_t alpha_is_one;                /* bit i set: stage i is evaluated at t1 with Perturb.PREV (rk_common.py:72-75) */
    int32_t n_times;    /* number of stages (func evaluations per step), 1..TDEQ_MAX_STAGE_TIMES               */
    int32_t n_norm_seg; /* leading segments that enter the max of the mixed norm (seminorm: adjoint.py:267-270) */
    int32_t leading_abs; /* 16-bit states only: segment 0, if it holds ONE element, enters the max as |x| itself instead of
                            its rms — the adjoint norms take their time component as `t.abs()` (adjoint.py:250, 273); for
                            fp32 / fp64 the two are the same number and the field is ignored */
} tdeq_step_ctrl;

/* ABI version of the loaded library (== TDEQ_ABI_VERSION). */
int tdeq_abi_version(void);

/* Bytes of device workspace the norm entry points need for a state of n_chunks chunks. */
size_t tdeq_workspace_bytes(int64_t n_chunks);

/*
 * Stage accumulate:  out = y0 + sum_j c_j * k_j ,  c_j = fl_T(fl_T(coef_j) * fl_T(dt)).
 * Replaces `yi = y0 + torch.sum(k[..., :i+1] * (beta_i * dt), dim=-1)` (rk_common.py:79), the
 * non-FSAL solution combine (rk_common.py:83-85) and `y1 = y0 + h0 * f0` of the initial-step
 * heuristic (misc.py:65).  Structural zeros of the tableau row are skipped by the caller.
 * 1 <= n_terms <= TDEQ_MAX_TERMS.  `out` may alias nothing it reads.
 */
int tdeq_stage_combine(void* out, const void* y0, const void* const* k, const double* coef,
                       int n_terms, double dt, int64_t n, int dtype, void* stream);

/*
 * Measurement hook: tdeq_stage_combine whose launch updates two caller-created hipEvent_t (enable-timing) with the
 * dispatch's own begin and end timestamps (hipExtLaunchKernelGGL), so that hipEventElapsedTime(start, stop) is the
 * kernel's duration as a profiler reports it — not the event -> launch -> event bracket, which adds ≈3 µs of
 * marker cost.  Same kernel, same results; 16-byte aligned buffers only (others take the plain launch and leave the
 * events untouched).  Used by bench.py for the live roofline figure of the dominant kernel.
 */
int tdeq_stage_combine_timed(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                             double dt, int64_t n, int dtype, void* stream, void* start_event, void* stop_event);

/*
 * tdeq_stage_combine for the FIRST stage of a step (n_terms 1 or 2, n >= 1) that also stores `n_fill` (<= 16)
 * scalars, converted to T, at fill_dst[0..n_fill): the stage times `ti = t0 + alpha_i * dt` (rk_common.py:72-78)
 * that the following func evaluations read as 0-dim tensors.  Saves the separate tdeq_fill_scalars launch at the
 * latency-critical start of a step; results are identical to the two separate calls.
 */
int tdeq_stage_combine_fill(void* out, const void* y0, const void* const* k, const double* coef, int n_terms,
                            double dt, int64_t n, int dtype, void* fill_dst, const double* fill_vals, int n_fill,
                            void* stream);

/*
 * Embedded error estimate + tolerance scaling + per-segment sum of squares, fused:
 *   err = sum_j fl_T(coef_j*dt) * k_j                       (rk_common.py:89)
 *   tol = atol + rtol * max(|y0|, |y1|)                      (misc.py:81)
 *   out_sumsq[s]  = sum over segment s of (err/tol)^2        (misc.py:22-23, 30-33, 82; fp64 accumulate)
 *   out_nonfinite[s] = number of non-finite elements of y0/y1 seen in segment s (rk_common.py:287)
 * The caller finishes sqrt(sumsq/numel) and the max over segments on the host.
 * `out_sumsq` / `out_nonfinite` (n_seg doubles each) may be device memory or pinned host memory.
 * `scaled_out` (optional, may be NULL): if given, err/tol is also stored per element (padding of a
 * segmented layout zero-filled) so that a user-supplied norm callable can reduce it.
 * `segs` is the host segment table; for n_seg > TDEQ_INLINE_SEGMENTS the caller also passes
 * `segs_dev`, a device copy of the same n_seg records (made once per solve), else it may be NULL.
 */
int tdeq_error_norm(void* scaled_out, const void* y0, const void* y1, const void* const* k,
                    const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                    const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks,
                    double* out_sumsq, double* out_nonfinite, void* workspace,
                    size_t workspace_bytes, int dtype, void* stream);

/*
 * tdeq_error_norm for PER-ELEMENT tolerances: `rtol` / `atol` tensors that broadcast against the state (misc.py:80-82 —
 * the reference needs no code for it) or tuple tolerances with vector entries (flat vectors, misc.py:115-123).  A
 * dimensioned tolerance is a device vector of fp64 (the time dtype of rk_common.py:186-187) over the flat, padded state;
 * the other one may be 0-dim (`*_vec` NULL, value in `*_scalar`); at least one must be dimensioned.  Promotion as ATen
 * does it: rtol[i] * max(|y0|,|y1|) in fp64 when rtol is dimensioned, in T (rtol cast to T) when it is 0-dim; the sum
 * with atol, err / tol and the sum of squares in fp64.  The segment table's rtol / atol are ignored.
 * `err_partial` (nullable, ABI 20): the error row's leading run as tdeq_stage_combine_err summed it —
 * err = (err_partial + c_0 k_0) + ... over the n_terms >= 0 remaining stages, as tdeq_error_norm_partial; NULL: the whole row
 * from k (n_terms >= 1).
 */
int tdeq_error_norm_vec(const void* err_partial, const void* y0, const void* y1, const void* const* k, const double* coef,
                        int n_terms, double dt, const double* rtol_vec, double rtol_scalar, const double* atol_vec, double atol_scalar,
                        const tdeq_segment* segs, const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks,
                        double* out_sumsq, double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                        void* stream);

/*
 * tdeq_error_norm_vec whose finalize step also runs the step controller on the device (as tdeq_error_norm_partial_ctrl
 * does for scalar tolerances): the error ratio in fp64 — the promoted type of the reference's quotient and norm when a
 * tolerance is dimensioned —, accept flag, next step size, and the next trial step's stage times in T; `out_ctrl`,
 * `ctrl_dev`, `next_times` as there.  Lets the look-ahead first stage (tdeq_stage_combine_sel) follow.
 * state_in_dev != 0 (ABI 21, hipGraph mode, as for tdeq_error_norm_partial_ctrl): the trial step's (t0, dt) and the step
 * size the error row's coefficients are multiplied by live in ctrl_dev; `dt` is ignored and c = fl_T(fl_T(coef) * T(dt))
 * is formed on the device — captured trial steps with per-element tolerances.
 */
int tdeq_error_norm_vec_ctrl(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                             const double* coef, int n_terms, double dt,
                             const double* rtol_vec, double rtol_scalar, const double* atol_vec, double atol_scalar,
                             const tdeq_segment* segs, const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks,
                             double* out_sumsq, double* out_nonfinite, const tdeq_step_ctrl* ctrl, double* out_ctrl,
                             double* ctrl_dev, void* next_times, int state_in_dev, void* workspace, size_t workspace_bytes,
                             int dtype, void* stream);

/*
 * Fused pair for the END of a trial step (same results as tdeq_stage_combine + tdeq_error_norm, fewer bytes):
 *   tdeq_stage_combine_err   the step's last combine (last stage row, or the c_sol combine of a non-FSAL pair)
 *                            also stores  err_out = (e_0 k_0 + e_1 k_1) + ...,  e_j = fl_T(fl_T(err_coef_j)*fl_T(dt)),
 *                            the partial embedded error over the SAME stages, left to right (rk_common.py:89);
 *   tdeq_error_norm_partial  err = (err_partial + c_0 k_0) + ... over the 0..2 remaining stages (for an FSAL
 *                            pair: the last stage slot only), then tolerance scaling, per-segment sums and the
 *                            non-finite census exactly as tdeq_error_norm.
 * The caller uses the pair only when the fused stages are a leading run of the error row's non-zero entries, so
 * the left-to-right sum order (and therefore every bit of err) is unchanged.  Traffic per dopri5 step:
 * 40 -> 37 words per element (7+1 and 4 instead of 7 and 8); dopri8: 105 -> 98.
 */
int tdeq_stage_combine_err(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                           const double* err_coef, int n_terms, double dt, int64_t n, int dtype, void* stream);
int tdeq_error_norm_partial(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                            const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                            const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                            double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype,
                            void* stream);

/*
 * Carried partial sums (r03).  Every tableau row re-reads all earlier stages: row i of `_runge_kutta_step`
 * (rk_common.py:69-81) is  y_i = y0 + ((c_i0 k_0 + c_i1 k_1) + ... + c_ii k_i), summed left to right, so the stages a
 * LATER row r needs from k_0..k_i are in registers while row i is being formed.  This entry point makes ONE pass over
 * `n_terms` stage streams and produces up to TDEQ_MAX_MULTI_OUT outputs from them:
 *
 *     s_o    = [acc_in +] sum over the set bits j of mask_o, ascending, of  fl_T(fl_T(coef_o[j]) * fl_T(dt)) * k_j
 *     out_o  = add_y0_o ? y0 + s_o : s_o
 *
 * Output 0 is the row's own stage input; a further output with add_y0 = 0 is the left-to-right PREFIX of a later
 * row's sum (a "carried" partial sum), which that row's launch continues through `acc_in` (output 0 only) over the
 * stages computed since — it then reads acc_in, the new stages and y0 instead of every earlier stage; a further
 * output with add_y0 = 1 is a later stage input that needs no newer stage at all (dopri8 row 12, whose weight on
 * k_11 is zero: dopri8.py:5-70) and is finished here.  The partial embedded error of tdeq_stage_combine_err is the
 * same thing (add_y0 = 0, continued by tdeq_error_norm_partial).  Rounding sequence = the one of tdeq_stage_combine
 * (first product not added to a zero, structural zeros skipped — not multiplied —, products and sums rounded
 * separately in T), so every output is BIT-IDENTICAL to the row-by-row kernels; only the bytes change:
 * dopri8 98 -> 75 words per element and step (13 launches instead of 14), dopri5 37 -> 35 (tableaus.carry_plan).
 * 1 <= n_terms <= TDEQ_MAX_TERMS, 1 <= n_out <= TDEQ_MAX_MULTI_OUT, every mask non-zero and within n_terms bits;
 * `acc_in` may be NULL.  Outputs may alias nothing that is read.
 */
#define TDEQ_MAX_MULTI_OUT 4
/* streaming launches whose streams add up to more than this many MiB use non-temporal loads and stores (tdeq_abi.hip
 * stream_policy; TDEQ_NT_THRESHOLD_MB overrides, TDEQ_COMBINE_POLICY=0..3 forces one policy) */
#define TDEQ_NT_THRESHOLD_DEFAULT_MB 512
typedef struct tdeq_multi_out {
    void* out;                      /* T[n]                                                                   */
    double coef[TDEQ_MAX_TERMS];    /* fp64 tableau weights of this output's row over the n_terms streams       */
    uint32_t mask;                  /* bit j set: stream j takes part in this output's sum                      */
    int32_t add_y0;                 /* 1: out = y0 + sum (a finished stage input); 0: out = sum (a partial sum) */
} tdeq_multi_out;
int tdeq_stage_combine_multi(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                             const void* const* k, int n_terms, double dt, int64_t n, int dtype, void* stream);
/* hipGraph mode: the step size is ctrl_dev[1] (sign * T(dt), maintained by tdeq_error_norm_partial_ctrl with
 * state_in_dev = 1), read on the device; c = fl_T(fl_T(coef) * T(dt)) as above, same bits as the host-dt entry point. */
int tdeq_stage_combine_multi_dev(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                                 const void* const* k, int n_terms, const double* ctrl_dev, int64_t n, int dtype,
                                 void* stream);
/* Measurement hook, as tdeq_stage_combine_timed: the dispatch stamps the two events with its own begin / end. */
int tdeq_stage_combine_multi_timed(const tdeq_multi_out* outs, int n_out, const void* y0, const void* acc_in,
                                   const void* const* k, int n_terms, double dt, int64_t n, int dtype, void* stream,
                                   void* start_event, void* stop_event);

/*
 * Device-resident step controller + look-ahead first stage.  The accept/reject LOOP stays on the host; what
 * moves to the device is the scalar decision of one trial step, so that the next trial step's first stage
 * (and the func evaluation behind it) can be enqueued BEFORE the host has read the decision back — the
 * poll -> controller -> launch latency of the host leaves the critical path.
 *
 *   tdeq_error_norm_partial_ctrl   = tdeq_error_norm_partial (same partial kernel, same per-segment sums in the
 *       same order) whose finalize step, one workgroup, also runs the reference's controller on the sums:
 *         ratio  = max_s sqrt(sumsq_s / numel_s) over the first n_norm_seg segments, rounded to T
 *                                                                                (misc.py:22-33, 80-82)
 *         accept = ratio <= 1, overridden by dt > max_step -> reject, dt <= min_step -> accept
 *                                                                                (rk_common.py:324-330)
 *         dt_next = clamp(_optimal_step_size(dt, ratio, safety, ifactor, dfactor, order), min_step, max_step)
 *                                                                                (misc.py:85-95, rk_common.py:353)
 *         next trial step: t0' = accept ? t0 + dt : t0 ; dt' = dt_next (min_step if non-finite);
 *         stage times t_i = t0' + alpha_i dt' in T, or nextafter(t1, t1 - 1) with t1 = T(t0' + dt') for alpha_i == 1
 *         (Perturb.PREV), times the time sign                                    (rk_common.py:72-78, misc.py:174-197)
 *       Outputs: out_sumsq / out_nonfinite as tdeq_error_norm_partial; out_ctrl[4] = {accept (0/1), dt_next,
 *       ratio, t0'} (device or pinned host memory, read by the host loop); ctrl_dev[4] = {accept, sign * T(dt'),
 *       t0', dt'} (device memory, read by tdeq_stage_combine_sel and by the hipGraph-mode kernels);
 *       next_times[n_times] (device memory, element type T) = the 0-dim time tensors the next trial step hands to
 *       func.  More than TDEQ_INLINE_SEGMENTS segments (segs_dev required, as for tdeq_error_norm): the sums come
 *       from the parallel per-segment finalize launch and the one-workgroup controller runs on them (one launch
 *       more; same sums, same decision).  state_in_dev != 0 (hipGraph mode, below): ctrl->t0 / ctrl->dt and
 *       the `dt` argument are ignored — the trial step's (t0, dt) are ctrl_dev[2..3] and the error coefficients are
 *       scaled by ctrl_dev[1], all left there by the previous call (or by the host before the first one).
 *       copy_last_k (ABI 21, nullable; fp32 / fp64 states, n_terms >= 1): the last remaining stage k[n_terms - 1] — the
 *       step's f1 = f(t1, y1) for an FSAL pair — is ALSO written to this buffer by the norm launch, which has the
 *       stream in registers: a captured trial step whose next step must read f1 from a buffer of its own (func's output
 *       buffer cannot be chosen) saves the N-word copy node.
 *
 *   tdeq_stage_combine_sel   first stage of the next trial step, launched before the host knows `accept`:
 *         (y, f) = accept ? (y_acc, f_acc) : (y_rej, f_rej) ;  out = y + fl_T(fl_T(coef) * T(dt')) * f
 *       i.e. tdeq_stage_combine with one term on the pair the controller selected (accepted: the new state and
 *       its FSAL derivative; rejected: the old pair, rk_common.py:335-361) and the controller's dt'.
 *       Bit-identical to the host-driven tdeq_stage_combine call it replaces.
 */
int tdeq_error_norm_partial_ctrl(const void* err_partial, const void* y0, const void* y1, const void* const* k,
                                 const double* coef, int n_terms, double dt, const tdeq_segment* segs,
                                 const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                                 double* out_nonfinite, const tdeq_step_ctrl* ctrl, double* out_ctrl, double* ctrl_dev,
                                 void* next_times, int state_in_dev, void* copy_last_k, void* workspace,
                                 size_t workspace_bytes, int dtype, void* stream);
int tdeq_stage_combine_sel(void* out, const void* y_acc, const void* f_acc, const void* y_rej, const void* f_rej,
                           double coef, const double* ctrl_dev, int64_t n, int dtype, void* stream);

/*
 * The controller alone, on per-segment sums that already sit in DEVICE memory — the lock-step mode of a batch-sharded
 * solve (no counterpart in the reference, which has no distributed code): every rank runs tdeq_error_norm_partial
 * into device buffers, the n_seg sums (+ non-finite counters) are all-reduced over the ranks ON THE DEVICE (RCCL over
 * xGMI, no host round trip), and this entry point then takes the same decision on every rank:
 *   ratio = max_s sqrt(sums[s] / segs[s].numel) over the first n_norm_seg segments — `segs[s].numel` must hold the
 *   GLOBAL element counts (summed over the ranks) — then accept / dt_next / next stage times exactly as
 *   tdeq_error_norm_partial_ctrl.  sums / nonfinite are mirrored to out_sumsq / out_nonfinite (device or pinned host
 *   memory) for the host's bookkeeping; the other outputs as tdeq_error_norm_partial_ctrl.  Only `numel` of the
 *   segment table is read (segs_dev required beyond TDEQ_INLINE_SEGMENTS segments).
 */
int tdeq_step_controller(const double* sums, const double* nonfinite, const tdeq_segment* segs, const void* segs_dev,
                         int n_seg, double* out_sumsq, double* out_nonfinite, const tdeq_step_ctrl* ctrl,
                         double* out_ctrl, double* ctrl_dev, void* next_times, int state_in_dev, int dtype,
                         void* stream);

/*
 * hipGraph mode of the adaptive solvers (small states, where a trial step is launch-latency-bound): ONE captured
 * graph = one trial step (S func evaluations, S + 4 launches) is replayed until the solve ends; the step size
 * lives in device memory (ctrl_dev, maintained by tdeq_error_norm_partial_ctrl with state_in_dev = 1) and the
 * state in static buffers.
 *   tdeq_stage_combine_dev  tdeq_stage_combine (err_out == NULL) or tdeq_stage_combine_err with
 *                           c_j = fl_T(fl_T(coef_j) * T(dt)), dt = ctrl_dev[1] read on the device; same operation
 *                           order, same results as the host-dt entry points (from 2^17 elements on with the same
 *                           unrolled, 16-byte-per-lane streams; below, one run-time-term kernel: launch-bound anyway).
 *   tdeq_step_commit        if ctrl_dev[0] (accept): (y_prev, f_prev) <- (y_cur, f_cur); (y_cur, f_cur) <- (y1, f1)
 *                           — the accepted state becomes the next trial's base (rk_common.py:335-352) and the
 *                           previous pair stays available for the dense output of the step just taken.
 *                           For callers that keep ONE captured graph and let the device select the pair.  The package's
 *                           own host (solvers._GraphStep, r03) no longer needs it: it reads the decision after every
 *                           replay anyway and alternates between two graphs over ping-pong buffers instead (no copy).
 */
int tdeq_stage_combine_dev(void* out, void* err_out, const void* y0, const void* const* k, const double* coef,
                           const double* err_coef, int n_terms, const double* ctrl_dev, int64_t n, int dtype,
                           void* stream);
int tdeq_step_commit(void* y_prev, void* f_prev, void* y_cur, void* f_cur, const void* y1, const void* f1,
                     const double* ctrl_dev, int64_t n, int dtype, void* stream);

/*
 * Initial-step norms (Hairer II.4 as in misc.py:36-77), scale = atol + |y0| * rtol:
 *   mode 0:  out_sumsq[s]          = sum (a / scale)^2           (a = y0 -> d0 ; misc.py:55)
 *            out_sumsq[n_seg + s]  = sum (b / scale)^2           (b = f0 -> d1 ; misc.py:56)
 *   mode 1:  out_sumsq[s]          = sum ((a - b) / scale)^2     (a = f1, b = f0 -> d2*h0 ; misc.py:68)
 * out_nonfinite[s] counts non-finite elements of yscale (the state) per segment.
 */
int tdeq_init_norms(int mode, const void* a, const void* b, const void* yscale,
                    const tdeq_segment* segs, const void* segs_dev, int n_seg, int64_t chunk,
                    int64_t n_chunks, double* out_sumsq, double* out_nonfinite, void* workspace, size_t workspace_bytes,
                    int dtype, void* stream);

/*
 * tdeq_init_norms with PER-ELEMENT tolerances (ABI 21): `rtol_vec` / `atol_vec` = fp64 device vectors over the flat (padded)
 * state, or NULL with the 0-dim value in `rtol_scalar` / `atol_scalar` (at least one vector) — what misc.py:50-56,68
 * computes by broadcasting when the caller's tolerances are tensors (rk_common.py:186-187 makes them W = fp64 tensors).
 * Promotion as ATen applies it: |y| * rtol[i] in fp64 for a dimensioned rtol, in T for a 0-dim one; the sum with atol,
 * the quotients and the sums of squares in fp64; (a - b) of mode 1 in T.  fp32 / fp64 states; outputs as tdeq_init_norms.
 */
int tdeq_init_norms_vec(int mode, const void* a, const void* b, const void* yscale, const double* rtol_vec,
                        double rtol_scalar, const double* atol_vec, double atol_scalar, const tdeq_segment* segs,
                        const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, double* out_sumsq,
                        double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype, void* stream);

/*
 * The quotients of tdeq_init_norms MATERIALISED, for a user-supplied norm callable (the reference hands its `norm`
 * to _select_initial_step too, rk_common.py:217 -> misc.py:55-56,68):
 *   mode 0: out0 = a / scale, out1 = b / scale          mode 1: out0 = (a - b) / scale      (out1 may be NULL)
 * element-wise over the segments; the padding of a segmented layout is zero-filled.
 */
int tdeq_init_scaled(int mode, const void* a, const void* b, const void* yscale, const tdeq_segment* segs,
                     const void* segs_dev, int n_seg, int64_t chunk, int64_t n_chunks, void* out0, void* out1,
                     int dtype, void* stream);

/*
 * Dense output, fused fit + evaluate (rk_common.py:363-369 + interp.py:1-48):
 *   y_mid = y0 + sum_j fl_T(coef_j*dt) * k_j ; quartic [e,d,c,b,a] from (y0,y1,y_mid,f0,f1,dt);
 *   out = e + x d + x^2 c + x^3 b + x^4 a, with x already rounded to T by the caller.
 * f0 / f1 are the first / last stage slots (k[...,0], k[...,-1]).
 */
int tdeq_dense_eval(void* out, const void* y0, const void* y1, const void* f0, const void* f1,
                    const void* const* k, const double* coef, int n_terms, double dt, double x,
                    int64_t n, int dtype, void* stream);

/*
 * tdeq_dense_eval for n_x (1..TDEQ_MAX_DENSE_OUTPUTS) output times that fall inside the SAME accepted step — the
 * reference calls _interp_evaluate once per output time (solvers.py:28-35 -> rk_common.py:243-250): the quartic is
 * fitted once per element and evaluated at x[0..n_x); row q is written at out + q*out_stride elements (the rows of
 * the solution tensor).  (8 + n_x) words per element instead of 9*n_x, one launch instead of n_x; each row is
 * bit-identical to the single-output call.
 */
int tdeq_dense_eval_multi(void* out, int64_t out_stride, const void* y0, const void* y1, const void* f0,
                          const void* f1, const void* const* k, const double* coef, int n_terms, double dt,
                          const double* x, int n_x, int64_t n, int dtype, void* stream);

/*
 * Dense-output fit only: writes the 5 interpolation coefficients [e,d,c,b,a] contiguously into
 * `coeffs` (5*n elements; interp.py:17-22).  Used by odeint_dense-style consumers ("next" row).
 */
int tdeq_interp_fit(void* coeffs, const void* y0, const void* y1, const void* f0, const void* f1,
                    const void* const* k, const double* coef, int n_terms, double dt, int64_t n,
                    int dtype, void* stream);

/*
 * rk4 "3/8 rule" stages (rk_common.py:110-118, solvers.py:115), dt already in T:
 *   stage 1: out = y0 + (dt*k1)*(1/3)
 *   stage 2: out = y0 + dt*(k2 - k1*(1/3))
 *   stage 3: out = y0 + dt*((k1 - k2) + k3)
 *   stage 4: out = y0 + (((k1 + 3*(k2+k3)) + k4)*dt)*0.125
 * Unused k pointers may be NULL.
 */
int tdeq_rk4_38_stage(int stage, void* out, const void* y0, const void* k1, const void* k2,
                      const void* k3, const void* k4, double dt, int64_t n, int dtype, void* stream);

/*
 * hipGraph mode of the fixed-grid rk4 solver (small states, where a step is launch-latency-bound): ONE captured
 * graph — four func evaluations, four tdeq_rk4_38_stage_dev launches, tdeq_grid_commit, tdeq_grid_advance — is
 * replayed once per grid interval; everything that changes between steps lives in device memory.
 *   tdeq_rk4_38_stage_dev  tdeq_rk4_38_stage with dt read from *dt_dev (device double holding sign*dt).
 *   tdeq_grid_advance      *counter += 1 (=: c); t0 = grid[c], t1 = grid[c+1], dt = t1 - t0 in the grid's dtype
 *                          (solvers.py:110-112); times_out[0..4) = the stage times t0, t0 + dt/3, t0 + 2dt/3, t1 of
 *                          the 3/8 rule (rk_common.py:110-118) cast to the state dtype, perturbed at both ends if
 *                          `perturb` (fixed-grid option, misc.py:174-197), times `sign`; *dt_out = sign * dt.
 *                          Nothing is written when c + 1 >= n_grid.  Start with *counter = -1.
 *   tdeq_grid_commit       solution[(*counter + 1) * row_stride + i] = y_new[i] and y_cur[i] = y_new[i], i < n: the
 *                          step's result becomes output row c + 1 and the next step's state (solvers.py:113-127
 *                          with the grid equal to the output times).
 */
int tdeq_rk4_38_stage_dev(int stage, void* out, const void* y0, const void* k1, const void* k2, const void* k3,
                          const void* k4, const double* dt_dev, int64_t n, int dtype, void* stream);
int tdeq_grid_advance(const void* grid, int grid_dtype, int64_t n_grid, int64_t* counter, int perturb, double sign,
                      void* times_out, double* dt_out, int state_dtype, void* stream);
int tdeq_grid_commit(void* solution, int64_t row_stride, void* y_cur, const void* y_new, const int64_t* counter,
                     int64_t n, int dtype, void* stream);
/*
 * hipGraph mode of the OTHER explicit fixed-grid methods (r03: euler, midpoint, heun2, heun3; fixed_grid.py:6-60,
 * rk_common.py:121-157) — the same scheme as tdeq_grid_advance / tdeq_rk4_38_stage_dev with the method described by data:
 *   tdeq_grid_advance_stages  counter += 1; t0 = grid[c], t1 = grid[c+1], dt = t1 - t0 in the grid's dtype; stage time i
 *                             = t1 if (mode_i & 1) else t0 + dt * fl_G(frac_i)  (t0 itself for frac_i = 0), cast to the
 *                             state dtype, perturbed to the NEXT (mode_i & 2) / PREVIOUS (mode_i & 4) representable value
 *                             when `perturb`, times `sign`; dt_out = sign * dt.  1 <= n_times <= 4.
 *   tdeq_fixed_stage_dev      tdeq_fixed_stage with the step size read from device memory (*dt_dev, rounded to T).
 * Euler's and midpoint's stages are tdeq_stage_combine_dev (whose ctrl_dev[1] is this dt_out).  Same operation order as the
 * host-dt entry points, so a captured step replays the eager step bit for bit.
 */
int tdeq_grid_advance_stages(const void* grid, int grid_dtype, int64_t n_grid, int64_t* counter, int perturb, double sign,
                             const double* frac, const int* mode, int n_times, void* times_out, double* dt_out,
                             int state_dtype, void* stream);
int tdeq_fixed_stage_dev(int mode, void* out, const void* y0, const void* const* k, const double* w, int n_terms,
                         const double* dt_dev, int64_t n, int dtype, void* stream);

/* Fixed-grid output interpolation  out = y0 + slope*(y1 - y0)  (solvers.py:175-181). */
int tdeq_lerp(void* out, const void* y0, const void* y1, double slope, int64_t n, int dtype,
              void* stream);

/*
 * Low-order fixed-grid Runge–Kutta stages with the reference's operation order (euler / midpoint use
 * tdeq_stage_combine, which is bit-identical for their single-term forms).  w_j and dt rounded to T:
 *   mode 0: out = y0 + dt * ((k0*w0 + k1*w1) + ...)   `y0 + dt * (k1*a31 + k2*a32)`, and y1 = y0 + dy with
 *                                                    dy = `dt * (k1*b1 + k2*b2 [+ k3*b3])`
 *                                                    (rk_common.py:139,140,156,157; solvers.py:115)
 *   mode 1: out = y0 + (dt * k0) * w0                 `y0 + dt * k1 * a21` (rk_common.py:138,156); n_terms == 1
 * 1 <= n_terms <= 4; structural zeros are skipped by the caller (heun2 / heun3, fixed_grid.py:32-60).
 */
int tdeq_fixed_stage(int mode, void* out, const void* y0, const void* const* k, const double* w, int n_terms,
                     double dt, int64_t n, int dtype, void* stream);

/*
 * out = (x0*w0 + x1*w1) + ... (left to right, w_j rounded to T), 1 <= n_terms <= TDEQ_MAX_SUM_TERMS.
 * Cubic Hermite output interpolation of the fixed-grid solvers,
 * `h00*y0 + h10*dt*f0 + h01*y1 + h11*dt*f1` (solvers.py:166-173), with the scalar products formed by the host.
 */
int tdeq_weighted_sum(void* out, const void* const* x, const double* w, int n_terms, int64_t n, int dtype,
                      void* stream);

/*
 * Backward helpers of the differentiable plain `odeint`.  Every elementwise entry
 * point above computes out = sum_m w_m(dt, x) * X_m, so the vector-Jacobian product the reference obtains
 * from autograd over its eager ops (incl. `_UncheckedAssign.backward`, rk_common.py:31-40) is:
 *   grad X_m = w_m * g                            tdeq_scale_many: g is read once, n_out tensors are written
 *   grad s   = sum_m dw_m/ds * <g, X_m>           tdeq_multi_dot: out[m] = <g, x_m> in fp64 (device memory),
 *                                                 for the time-like scalars s = dt, x when `t` requires grad
 * 1 <= n_out, n_x <= TDEQ_MAX_TERMS.  tdeq_multi_dot needs tdeq_dots_workspace_bytes(n, n_x) bytes of device
 * workspace; n == 0 yields zeros.
 */
int tdeq_scale_many(void* const* outs, const void* g, const double* w, int n_out, int64_t n, int dtype,
                    void* stream);
size_t tdeq_dots_workspace_bytes(int64_t n, int n_x);
int tdeq_multi_dot(const void* g, const void* const* x, int n_x, int64_t n, double* out, void* workspace,
                   size_t workspace_bytes, int dtype, void* stream);

/*
 * Pack the pieces of a tuple-valued func output into ONE flat segmented state with one launch:
 *   out[chunk_start[s]*chunk + i] = scale[s] * src[s][i]   for i < numel[s];   padding and src[s] == NULL -> 0
 * over n_chunks*chunk elements of `out`.  Replaces `torch.cat([f_.reshape(-1) for f_ in f])` of _TupleFunc
 * (misc.py:137-145) and, for the adjoint's augmented dynamics (adjoint.py:72-105), also the `-adj_y` negation (:95),
 * the zeros_like of absent gradients (:99-103) and _ReverseFunc's multiply (misc.py:158-165): scale[s] is +1 or -1,
 * so every product is exact.  src[s]: contiguous, element type T, numel[s] elements.  1 <= n_seg <=
 * TDEQ_INLINE_SEGMENTS, chunk_start[0] == 0 and strictly increasing, numel[s] <= room of segment s.
 */
int tdeq_pack_segments(void* out, const void* const* src, const int64_t* chunk_start, const int64_t* numel,
                       const double* scale, int n_seg, int64_t chunk, int64_t n_chunks, int dtype, void* stream);

/* Writes n_vals scalars (converted to T) to consecutive elements of dst (stage times for func). */
int tdeq_fill_scalars(void* dst, const double* vals, int n_vals, int dtype, void* stream);

/*
 * Adams–Bashforth(–Moulton) multistep steps of the fixed-grid solvers `explicit_adams`, `implicit_adams`
 * (= `fixed_adams`), fixed_adams.py:164-228.  f_hist[j] = func output at t_{n-j} (newest first, separate
 * contiguous tensors — the reference's deque `prev_f`), 1 <= n_terms <= TDEQ_MAX_TERMS (the reference uses 3..11).
 *
 * tdeq_adams_predict — one pass over the history:
 *   dy    = (cb_0*f_0 + cb_1*f_1) + ...       `_dot_product(dt * bashforth_coeffs, prev_f)` (:205); the caller passes
 *                                             cb_j = dt*b_j formed in fp64, rounded to T here (0-dim fp64 x T tensor)
 *   y_out = y0 + dy                           (:213 / solvers.py:115)
 *   delta = T(dt) * ((cm_0*f_0 + cm_1*f_1) + ...)   `dt * _dot_product(moulton_coeffs[1:], prev_f)` (:210), cm_j = m_{j+1}
 * dy_out / delta_out / cm are given together (implicit method) or all NULL (explicit method).
 *
 * tdeq_adams_correct — one corrector iteration and its convergence test in one pass (:212-216, :189-192):
 *   compute != 0:  dy = T(c)*f + delta (c = dt*m_0 formed in fp64 by the caller), dy_out = dy, y_out = y0 + dy
 *   compute == 0:  dy is read from dy_out (test only; y_out, f, delta, y0 unused)
 *   out_count[s]      = number of elements of segment s with NOT (|dy_old - dy| / (atol_s + rtol_s*max(|dy_old|,|dy|)) < 1)
 *                       — `_has_converged` (l-inf norm of the error ratio < 1, misc.py:80-82) <=> all counts are 0
 *   out_nonfinite[s]  = number of non-finite elements of dy in segment s
 * Segment table, workspace (tdeq_workspace_bytes(n_chunks)) and result placement as for tdeq_error_norm; n = number of
 * elements of every buffer (>= the end of the last segment; the padding of a segmented state is written, not counted).
 */
int tdeq_adams_predict(void* y_out, void* dy_out, void* delta_out, const void* y0, const void* const* f_hist,
                       const double* cb, const double* cm, int n_terms, double dt, int64_t n, int dtype,
                       void* stream);
int tdeq_adams_correct(void* y_out, void* dy_out, const void* f, const void* delta, const void* dy_old,
                       const void* y0, double c, int compute, const tdeq_segment* segs, const void* segs_dev,
                       int n_seg, int64_t chunk, int64_t n_chunks, int64_t n, double* out_count,
                       double* out_nonfinite, void* workspace, size_t workspace_bytes, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDEQ_HIP_H */
