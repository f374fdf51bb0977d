/*
 * A plain C client of libtdeq_hip.so: no Python, no torch — only include/tdeq_hip.h and the HIP runtime for device
 * memory.  Built and run by tests/test_abi_c_client.py.  It performs one dopri5-style stage combine, one Adams
 * predictor step, a carried-partial-sum pair (tdeq_stage_combine_multi) and — r05 — a bfloat16 stage combine on 1000003 elements and checks every element against the same arithmetic done on the host with
 * the documented rounding sequence (coefficient = fl(fl(coef) * fl(dt)), products and sums rounded separately).
 * Exit code 0 = all elements bit-identical.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tdeq_hip.h"

#define CHECK(x) do { int e_ = (int)(x); if (e_ != 0) { fprintf(stderr, "%s failed: %d\n", #x, e_); return 2; } } while (0)

int main(void) {
    const int64_t n = 1000003;
    const int nt = 3;
    if (tdeq_abi_version() != TDEQ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    float* h[5];
    for (int j = 0; j < 5; ++j) h[j] = (float*)malloc(sizeof(float) * n);
    uint32_t s = 12345u;
    for (int j = 0; j < 4; ++j)
        for (int64_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[j][i] = (float)((int32_t)(s >> 8) - (1 << 23)) / (float)(1 << 22); }
    float* d[5];
    for (int j = 0; j < 5; ++j) CHECK(hipMalloc((void**)&d[j], sizeof(float) * n));
    for (int j = 0; j < 4; ++j) CHECK(hipMemcpy(d[j], h[j], sizeof(float) * n, hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK(hipStreamCreate(&stream));
    const void* k[3] = {d[1], d[2], d[3]};
    const double coef[3] = {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0};
    const double dt = 0.0371;
    long bad = 0;

    /* y0 + sum_j fl(fl(coef_j) * fl(dt)) * k_j */
    CHECK(tdeq_stage_combine(d[4], d[0], k, coef, nt, dt, n, TDEQ_F32, stream));
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(h[4], d[4], sizeof(float) * n, hipMemcpyDeviceToHost));
    {
        volatile float c[3];
        for (int j = 0; j < nt; ++j) c[j] = (float)coef[j] * (float)dt;
        for (int64_t i = 0; i < n; ++i) {
            volatile float acc = h[1][i] * c[0];
            for (int j = 1; j < nt; ++j) { volatile float p = h[j + 1][i] * c[j]; acc = acc + p; }
            volatile float ref = h[0][i] + acc;
            if (ref != h[4][i]) ++bad;
        }
    }
    /* Adams predictor: y0 + sum_j fl(cb_j) * f_j, cb_j given in double */
    const double cb[3] = {dt * 23.0 / 12.0, dt * -16.0 / 12.0, dt * 5.0 / 12.0};
    CHECK(tdeq_adams_predict(d[4], NULL, NULL, d[0], k, cb, NULL, nt, 0.0, n, TDEQ_F32, stream));
    CHECK(hipStreamSynchronize(stream));
    CHECK(hipMemcpy(h[4], d[4], sizeof(float) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        volatile float acc = h[1][i] * (float)cb[0];
        for (int j = 1; j < nt; ++j) { volatile float p = h[j + 1][i] * (float)cb[j]; acc = acc + p; }
        volatile float ref = h[0][i] + acc;
        if (ref != h[4][i]) ++bad;
    }
    /* carried partial sums (tdeq_stage_combine_multi): one pass over k_0..k_2 forms y = y0 + (c0 k0 + c1 k1 + c2 k2) and
     * carries the prefix p = e0 k0 + e2 k2 of a later row (structural zero on k1: skipped, not multiplied); a second
     * launch continues that row over one more stage, y' = y0 + (p + e3 k3) with k3 := the first launch's y.  Host check:
     * the same left-to-right sums. */
    {
        float* d5;
        CHECK(hipMalloc((void**)&d5, sizeof(float) * n));
        float* d6;
        CHECK(hipMalloc((void**)&d6, sizeof(float) * n));
        const double e[4] = {0.37, 0.0, -1.25, 2.0 / 3.0};
        tdeq_multi_out outs[2];
        outs[0].out = d[4]; outs[0].mask = 0x7u; outs[0].add_y0 = 1;
        outs[1].out = d5;   outs[1].mask = 0x5u; outs[1].add_y0 = 0;
        for (int j = 0; j < TDEQ_MAX_TERMS; ++j) { outs[0].coef[j] = j < 3 ? coef[j] : 0.0; outs[1].coef[j] = j < 3 ? e[j] : 0.0; }
        CHECK(tdeq_stage_combine_multi(outs, 2, d[0], NULL, k, 3, dt, n, TDEQ_F32, stream));
        const void* k2[1] = {d[4]};
        tdeq_multi_out cont[1];
        cont[0].out = d6; cont[0].mask = 0x1u; cont[0].add_y0 = 1;
        for (int j = 0; j < TDEQ_MAX_TERMS; ++j) cont[0].coef[j] = j == 0 ? e[3] : 0.0;
        CHECK(tdeq_stage_combine_multi(cont, 1, d[0], d5, k2, 1, dt, n, TDEQ_F32, stream));
        CHECK(hipStreamSynchronize(stream));
        float* hy = (float*)malloc(sizeof(float) * n);
        float* hz = (float*)malloc(sizeof(float) * n);
        CHECK(hipMemcpy(hy, d[4], sizeof(float) * n, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hz, d6, sizeof(float) * n, hipMemcpyDeviceToHost));
        volatile float c[3], ee[4];
        for (int j = 0; j < 3; ++j) c[j] = (float)coef[j] * (float)dt;
        for (int j = 0; j < 4; ++j) ee[j] = (float)e[j] * (float)dt;
        for (int64_t i = 0; i < n; ++i) {
            volatile float acc = h[1][i] * c[0];
            for (int j = 1; j < 3; ++j) { volatile float p = h[j + 1][i] * c[j]; acc = acc + p; }
            volatile float y = h[0][i] + acc;
            volatile float pre = h[1][i] * ee[0];
            { volatile float p = h[3][i] * ee[2]; pre = pre + p; }
            { volatile float p = y * ee[3]; pre = pre + p; }
            volatile float z = h[0][i] + pre;
            if (y != hy[i] || z != hz[i]) ++bad;
        }
        if (tdeq_stage_combine_multi(outs, TDEQ_MAX_MULTI_OUT + 1, d[0], NULL, k, 3, dt, n, TDEQ_F32, stream) != TDEQ_EINVAL) ++bad;
        free(hy); free(hz);
    }
    /* bfloat16 state (ABI 18, dtype TDEQ_BF16): the same stage combine on 16-bit storage.  The reference integrates a bf16
     * state with ATen ops that compute in float32 and round every result to bf16 (misc.py:185-187, rk_common.py:61-65), a
     * row's torch.sum accumulates the rounded products in float32 and rounds once — restated here in plain C:
     * c_j = bf(bf(coef_j) * bf(dt)); p_j = bf(k_j * c_j); s = bf(p_0 + p_1 + p_2 in float32); y = bf(y0 + s). */
    {
        uint16_t* hb[5];
        uint16_t* db[5];
        for (int j = 0; j < 5; ++j) { hb[j] = (uint16_t*)malloc(2 * n); CHECK(hipMalloc((void**)&db[j], 2 * n)); }
#define F2BF(dst, src) do { uint32_t u_; volatile float f_ = (src); memcpy(&u_, (const void*)&f_, 4); \
                            (dst) = (uint16_t)((u_ + 0x7fffu + ((u_ >> 16) & 1u)) >> 16); } while (0)
#define BF2F(dst, src) do { uint32_t u_ = (uint32_t)(src) << 16; float f_; memcpy(&f_, &u_, 4); (dst) = f_; } while (0)
        for (int j = 0; j < 4; ++j)
            for (int64_t i = 0; i < n; ++i) F2BF(hb[j][i], h[j][i]);
        for (int j = 0; j < 4; ++j) CHECK(hipMemcpy(db[j], hb[j], 2 * n, hipMemcpyHostToDevice));
        const void* kb[3] = {db[1], db[2], db[3]};
        CHECK(tdeq_stage_combine(db[4], db[0], kb, coef, nt, dt, n, TDEQ_BF16, stream));
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipMemcpy(hb[4], db[4], 2 * n, hipMemcpyDeviceToHost));
        float cbf[3];
        {
            uint16_t t_; float dtb;
            F2BF(t_, (float)dt); BF2F(dtb, t_);
            for (int j = 0; j < nt; ++j) {
                float cj;
                F2BF(t_, (float)coef[j]); BF2F(cj, t_);
                volatile float prod = cj * dtb;
                F2BF(t_, prod); BF2F(cbf[j], t_);
            }
        }
        for (int64_t i = 0; i < n; ++i) {
            volatile float acc = 0.0f;
            for (int j = 0; j < nt; ++j) {
                float kj, pj; uint16_t t_;
                BF2F(kj, hb[j + 1][i]);
                volatile float prod = kj * cbf[j];
                F2BF(t_, prod); BF2F(pj, t_);
                acc = (j == 0) ? pj : acc + pj;
            }
            uint16_t t_, yb; float sb, y0f;
            F2BF(t_, acc); BF2F(sb, t_);
            BF2F(y0f, hb[0][i]);
            volatile float ysum = y0f + sb;
            F2BF(yb, ysum);
            if (yb != hb[4][i]) ++bad;
        }
        /* an entry point without 16-bit kernels says so instead of misreading the buffers */
        if (tdeq_adams_predict(db[4], NULL, NULL, db[0], kb, cb, NULL, nt, 0.0, n, TDEQ_BF16, stream) != TDEQ_EINVAL) ++bad;
        for (int j = 0; j < 5; ++j) { free(hb[j]); hipFree(db[j]); }
    }
    /* argument errors are reported, not crashed on */
    if (tdeq_stage_combine(NULL, d[0], k, coef, nt, dt, n, TDEQ_F32, stream) != TDEQ_EINVAL) ++bad;
    if (tdeq_stage_combine(d[4], d[0], k, coef, TDEQ_MAX_TERMS + 1, dt, n, TDEQ_F32, stream) != TDEQ_EINVAL) ++bad;
    printf("abi_client: %ld mismatching elements of %ld\n", bad, (long)(4 * n));
    return bad ? 1 : 0;
}
