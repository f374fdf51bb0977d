// lastblock_probe.hip — is a fused "last workgroup finalizes" error norm worth it where a dispatch costs more than a
// fence?  (r01 rejected it at 8 M elements.)
//
// Two ways to get ONE fp64 sum of (e/tol)^2 over n elements out of the GPU, embedded between two streaming kernels X
// (stand-ins for the neighbouring stage combine / func evaluation of a captured trial step), replayed as a hipGraph:
//   split : X -> P (one partial per 2048-element chunk) -> F (one workgroup adds the partials)      -> X
//   fused : X -> PF (P; every workgroup: release fence + ticket; the LAST one: acquire fence + adds) -> X
// PF uses the hand-off form MI355X_MICROARCH.md lists as valid: plain stores -> __syncthreads -> lane-0 agent release
// fence -> s_waitcnt vmcnt(0) -> relaxed agent atomic (ticket); last arriver: lane-0 agent acquire fence ->
// __syncthreads -> plain loads.  Prints microseconds per replay for both graphs at several n and checks the sums agree.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lastblock_probe.bin tools/lastblock_probe.hip && tools/lastblock_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

constexpr int kBlock = 256, kWave = 64, kChunk = 2048;

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < kBlock / kWave; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ double chunk_partial(const float* e, const float* y0, const float* y1, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    double acc = 0.0;
    for (int t = threadIdx.x * 4; t < kChunk; t += kBlock * 4) {
        if (base + t + 3 < n) {
            const float4 ev = *reinterpret_cast<const float4*>(e + base + t);
            const float4 a = *reinterpret_cast<const float4*>(y0 + base + t);
            const float4 b = *reinterpret_cast<const float4*>(y1 + base + t);
            const float ee[4] = {ev.x, ev.y, ev.z, ev.w}, aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
            for (int q = 0; q < 4; ++q) {
                const float tol = 1e-9f + 1e-7f * fmaxf(fabsf(aa[q]), fabsf(bb[q]));
                const float r = ee[q] / tol;
                acc += (double)r * (double)r;
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(kBlock) void P(const float* e, const float* y0, const float* y1, int64_t n, double* part) {
    __shared__ double red[kBlock / kWave];
    const double s = block_sum(chunk_partial(e, y0, y1, n), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__device__ __forceinline__ double add_partials(const double* part, int n_part, double* red) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kBlock) acc += part[i];
    return block_sum(acc, red);
}

__global__ __launch_bounds__(kBlock) void F(const double* part, int n_part, double* out) {
    __shared__ double red[kBlock / kWave];
    const double s = add_partials(part, n_part, red);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(kBlock) void PF(const float* e, const float* y0, const float* y1, int64_t n, double* part,
                                              unsigned* ticket, double* out) {
    __shared__ double red[kBlock / kWave];
    __shared__ int last;
    const double s = block_sum(chunk_partial(e, y0, y1, n), red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (last) {
        __syncthreads();
        const double tot = add_partials(part, (int)gridDim.x, red);
        if (threadIdx.x == 0) {
            out[0] = tot;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // re-arm for the next replay
        }
    }
}

__global__ __launch_bounds__(kBlock) void X(const float* a, const float* b, float* o, int64_t n) {     // out = a + 0.5 b
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 x = *reinterpret_cast<const float4*>(a + i), y = *reinterpret_cast<const float4*>(b + i);
        *reinterpret_cast<float4*>(o + i) = float4{x.x + 0.5f * y.x, x.y + 0.5f * y.y, x.z + 0.5f * y.z, x.w + 0.5f * y.w};
    }
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    printf("{\n \"unit\": \"us per graph replay (X -> norm -> X), median of 5 x 400 replays\"");
    for (int64_t n : {131072LL, 1048576LL, 2097152LL, 8388608LL}) {
        float *e, *y0, *y1, *o;
        double *part, *out_a, *out_b;
        unsigned* ticket;
        const int n_part = (int)(n / kChunk);
        CHECK(hipMalloc(&e, n * 4)); CHECK(hipMalloc(&y0, n * 4)); CHECK(hipMalloc(&y1, n * 4)); CHECK(hipMalloc(&o, n * 4));
        CHECK(hipMalloc(&part, n_part * 8)); CHECK(hipMalloc(&out_a, 8)); CHECK(hipMalloc(&out_b, 8)); CHECK(hipMalloc(&ticket, 4));
        CHECK(hipMemset(ticket, 0, 4));
        std::vector<float> h(n);
        for (int64_t i = 0; i < n; ++i) h[i] = 1e-7f * (float)((i * 2654435761u) % 1000) / 1000.0f;
        CHECK(hipMemcpy(e, h.data(), n * 4, hipMemcpyHostToDevice));
        for (int64_t i = 0; i < n; ++i) h[i] = 1.0f + (float)(i % 7);
        CHECK(hipMemcpy(y0, h.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(y1, h.data(), n * 4, hipMemcpyHostToDevice));
        const unsigned gx = (unsigned)((n / 4 + kBlock - 1) / kBlock);
        hipGraph_t g[2];
        hipGraphExec_t ge[2];
        for (int v = 0; v < 2; ++v) {
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            hipLaunchKernelGGL(X, dim3(gx), dim3(kBlock), 0, s, y0, y1, o, n);
            if (v == 0) {
                hipLaunchKernelGGL(P, dim3(n_part), dim3(kBlock), 0, s, e, y0, o, n, part);
                hipLaunchKernelGGL(F, dim3(1), dim3(kBlock), 0, s, part, n_part, out_a);
            } else {
                hipLaunchKernelGGL(PF, dim3(n_part), dim3(kBlock), 0, s, e, y0, o, n, part, ticket, out_b);
            }
            hipLaunchKernelGGL(X, dim3(gx), dim3(kBlock), 0, s, y0, o, y1, n);
            CHECK(hipStreamEndCapture(s, &g[v]));
            CHECK(hipGraphInstantiate(&ge[v], g[v], nullptr, nullptr, 0));
        }
        double med[2];
        for (int v = 0; v < 2; ++v) {
            std::vector<double> runs;
            for (int rep = 0; rep < 6; ++rep) {
                hipEvent_t a, b;
                CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
                CHECK(hipEventRecord(a, s));
                for (int i = 0; i < 400; ++i) CHECK(hipGraphLaunch(ge[v], s));
                CHECK(hipEventRecord(b, s));
                CHECK(hipStreamSynchronize(s));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                if (rep) runs.push_back(1e3 * ms / 400);
            }
            std::sort(runs.begin(), runs.end());
            med[v] = runs[2];
        }
        double ra, rb;
        CHECK(hipMemcpy(&ra, out_a, 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&rb, out_b, 8, hipMemcpyDeviceToHost));
        printf(",\n \"%lld\": {\"split_P_then_F\": %.3f, \"fused_last_block\": %.3f, \"gain_us\": %.3f, \"sums_equal\": %s}",
               (long long)n, med[0], med[1], med[0] - med[1], ra == rb ? "true" : "false");
        hipFree(e); hipFree(y0); hipFree(y1); hipFree(o); hipFree(part); hipFree(out_a); hipFree(out_b); hipFree(ticket);
    }
    printf("\n}\n");
    return 0;
}
