/*
 * A plain C client of libtdeq_hip.so: no Python, no torch — only include/tdeq_hip.h and the HIP runtime for device
 * memory.  Built and run by tests/test_abi_c_client.py.  It performs one dopri5-style stage combine, one Adams
 * predictor step and a carried-partial-sum pair (tdeq_stage_combine_multi) on 1000003 fp32 elements and checks every element against the same arithmetic done on the host with
 * the documented rounding sequence (coefficient = fl(fl(coef) * fl(dt)), products and sums rounded separately).
 * Exit code 0 = all elements bit-identical.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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
    /* argument errors are reported, not crashed on */
    if (tdeq_stage_combine(NULL, d[0], k, coef, nt, dt, n, TDEQ_F32, stream) != TDEQ_EINVAL) ++bad;
    if (tdeq_stage_combine(d[4], d[0], k, coef, TDEQ_MAX_TERMS + 1, dt, n, TDEQ_F32, stream) != TDEQ_EINVAL) ++bad;
    printf("abi_client: %ld mismatching elements of %ld\n", bad, (long)(3 * n));
    return bad ? 1 : 0;
}
