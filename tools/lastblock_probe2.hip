// lastblock_probe2.hip — r06, VERDICT r05 item 3: the "last workgroup finalizes" error norm once more, this time with
// the WRITE-THROUGH hand-off MI355X_MICROARCH.md lists as valid ("sc1 payload -> asm vmcnt(0) -> flag; sc1 loads may
// replace the acquire only when the producer stored sc1") instead of the release fence r03 measured
// (tools/lastblock_probe.hip: buffer_wbl2 per workgroup, 152 vs 56 us at 8 M elements).
//
//   split   : X -> P (one fp64 partial per 2048-element chunk, plain store) -> F (one workgroup adds them) -> X
//   fence   : X -> PF  (plain store, lane-0 agent release fence, ticket; last: acquire fence, plain loads)  -> X      [r03]
//   wt      : X -> PFw (ONE 8-byte sc1 store per workgroup, s_waitcnt vmcnt(0), relaxed agent ticket;
//                       last arriver: sc1 loads of the partials, fixed-order sum, re-arms the ticket)       -> X
//   atom    : X -> PFa (atomic exchange of the partial, ticket; last arriver reads with atomic or-0)          -> X
// Timing: hipGraph replays, median of 5 x 400.  Staleness: 2000 stream launches per fused variant in which every
// partial is a per-launch generation tag (gen * 65536 + chunk); the last arriver counts every word that is not this
// launch's, with a neighbouring streaming kernel on a second stream as uneven load.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lastblock_probe2.bin tools/lastblock_probe2.hip && tools/lastblock_probe2.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
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

enum { SPLIT = 0, FENCE = 1, WT = 2, ATOM = 3, HWT = 4 };
constexpr int kGroups = 16;     // HWT: workgroup b arrives at ticket[1 + b % kGroups]; the last of a group at ticket[0]

template <int MODE> __device__ __forceinline__ double load_partial(double* part, int i) {
    if (MODE == WT || MODE == HWT) return __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == ATOM) {
        const unsigned long long u = __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(part) + i, 0ull,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __longlong_as_double((long long)u);
    }
    return part[i];
}

template <int MODE> __device__ __forceinline__ double add_partials(double* part, int n_part, double* red) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kBlock) acc += load_partial<MODE>(part, i);
    return block_sum(acc, red);
}

__global__ __launch_bounds__(kBlock) void P(const float* e, const float* y0, const float* y1, int64_t n, double* part) {
    __shared__ double red[kBlock / kWave];
    const double s = block_sum(chunk_partial(e, y0, y1, n), red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(kBlock) void F(double* part, int n_part, double* out) {
    __shared__ double red[kBlock / kWave];
    const double s = add_partials<SPLIT>(part, n_part, red);
    if (threadIdx.x == 0) out[0] = s;
}

// gen < 0: the real norm; gen >= 0: staleness test — the partial is the tag gen * 65536 + chunk, `bad` counts wrong words
template <int MODE>
__global__ __launch_bounds__(kBlock) void PF(const float* e, const float* y0, const float* y1, int64_t n, double* part,
                                              unsigned* ticket, double* out, long long gen, unsigned long long* bad) {
    __shared__ double red[kBlock / kWave];
    __shared__ int last;
    double s = block_sum(chunk_partial(e, y0, y1, n), red);
    if (threadIdx.x == 0) {
        if (gen >= 0) s = (double)(gen * 65536 + blockIdx.x) + 0.0 * s;
        if (MODE == FENCE) {
            part[blockIdx.x] = s;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else if (MODE == WT || MODE == HWT) {
            __hip_atomic_store(part + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // global_store_dwordx2 sc1
        } else {
            __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(part) + blockIdx.x,
                                  (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == HWT) {
            // two-level arrival: 16 group tickets absorb the same-address serialisation (~9 ns per returning atomic), only
            // the last arriver of a group touches the top ticket
            const unsigned grp = blockIdx.x % kGroups;
            const unsigned in_grp = (gridDim.x - grp + kGroups - 1) / kGroups;
            const unsigned n_grp = gridDim.x < (unsigned)kGroups ? gridDim.x : (unsigned)kGroups;
            const unsigned tg = __hip_atomic_fetch_add(ticket + 1 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = 0;
            if (tg == in_grp - 1) {
                __hip_atomic_store(ticket + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tt = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (tt == n_grp - 1);
            }
        } else {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
        }
        if (last && MODE == FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (last) {
        if (gen >= 0) {
            unsigned long long wrong = 0;
            for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock)
                wrong += load_partial<MODE>(part, i) != (double)(gen * 65536 + i);
            if (wrong) atomicAdd(bad, wrong);
        }
        const double tot = add_partials<MODE>(part, (int)gridDim.x, red);
        if (threadIdx.x == 0) {
            out[0] = tot;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // re-arm for the next launch
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

template <int MODE> void launch_pf(hipStream_t s, int n_part, const float* e, const float* y0, const float* y1, int64_t n,
                                   double* part, unsigned* ticket, double* out, long long gen, unsigned long long* bad) {
    hipLaunchKernelGGL(PF<MODE>, dim3(n_part), dim3(kBlock), 0, s, e, y0, y1, n, part, ticket, out, gen, bad);
}

int main() {
    hipStream_t s, s2;
    CHECK(hipStreamCreate(&s));
    CHECK(hipStreamCreate(&s2));
    printf("{\n \"unit\": \"us per graph replay (X -> norm -> X), median of 5 x 400 replays; stale = wrong words in 2000 tagged launches under load\"");
    for (int64_t n : {131072LL, 1048576LL, 2097152LL, 8388608LL}) {
        float *e, *y0, *y1, *o, *junk;
        double *part, *out[5];
        unsigned* ticket;
        unsigned long long* bad;
        const int n_part = (int)(n / kChunk);
        CHECK(hipMalloc(&e, n * 4)); CHECK(hipMalloc(&y0, n * 4)); CHECK(hipMalloc(&y1, n * 4)); CHECK(hipMalloc(&o, n * 4));
        CHECK(hipMalloc(&junk, (8 << 20) * 4));
        CHECK(hipMalloc(&part, n_part * 8)); CHECK(hipMalloc(&ticket, 4 * (1 + kGroups))); CHECK(hipMalloc(&bad, 8));
        for (int v = 0; v < 5; ++v) CHECK(hipMalloc(&out[v], 8));
        CHECK(hipMemset(ticket, 0, 4 * (1 + kGroups))); CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(junk, 0, (8 << 20) * 4));
        std::vector<float> h(n);
        for (int64_t i = 0; i < n; ++i) h[i] = 1e-7f * (float)((i * 2654435761u) % 1000) / 1000.0f;
        CHECK(hipMemcpy(e, h.data(), n * 4, hipMemcpyHostToDevice));
        for (int64_t i = 0; i < n; ++i) h[i] = 1.0f + (float)(i % 7);
        CHECK(hipMemcpy(y0, h.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(y1, h.data(), n * 4, hipMemcpyHostToDevice));
        const unsigned gx = (unsigned)((n / 4 + kBlock - 1) / kBlock);
        hipGraph_t g[5];
        hipGraphExec_t ge[5];
        for (int v = 0; v < 5; ++v) {
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            hipLaunchKernelGGL(X, dim3(gx), dim3(kBlock), 0, s, y0, y1, o, n);
            if (v == SPLIT) {
                hipLaunchKernelGGL(P, dim3(n_part), dim3(kBlock), 0, s, e, y0, o, n, part);
                hipLaunchKernelGGL(F, dim3(1), dim3(kBlock), 0, s, part, n_part, out[v]);
            } else if (v == FENCE) {
                launch_pf<FENCE>(s, n_part, e, y0, o, n, part, ticket, out[v], -1, bad);
            } else if (v == WT) {
                launch_pf<WT>(s, n_part, e, y0, o, n, part, ticket, out[v], -1, bad);
            } else if (v == ATOM) {
                launch_pf<ATOM>(s, n_part, e, y0, o, n, part, ticket, out[v], -1, bad);
            } else {
                launch_pf<HWT>(s, n_part, e, y0, o, n, part, ticket, out[v], -1, bad);
            }
            hipLaunchKernelGGL(X, dim3(gx), dim3(kBlock), 0, s, y0, o, y1, n);
            CHECK(hipStreamEndCapture(s, &g[v]));
            CHECK(hipGraphInstantiate(&ge[v], g[v], nullptr, nullptr, 0));
        }
        double med[5], res[5];
        for (int v = 0; v < 5; ++v) {
            CHECK(hipMemcpy(y1, h.data(), n * 4, hipMemcpyHostToDevice));      // same evolution of y1 for every variant
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
            CHECK(hipMemcpy(&res[v], out[v], 8, hipMemcpyDeviceToHost));
        }
        // staleness: tagged partials, a different generation per launch, a streaming kernel on a second stream as load
        unsigned long long stale[5] = {0, 0, 0, 0, 0};
        for (int v = 1; v < 5; ++v) {
            CHECK(hipMemset(bad, 0, 8));
            for (long long gen = 0; gen < 2000; ++gen) {
                if (gen % 3 == 0) hipLaunchKernelGGL(X, dim3(8192), dim3(kBlock), 0, s2, junk, junk, junk, (int64_t)(8 << 20));
                if (v == FENCE) launch_pf<FENCE>(s, n_part, e, y0, o, n, part, ticket, out[v], gen, bad);
                if (v == WT) launch_pf<WT>(s, n_part, e, y0, o, n, part, ticket, out[v], gen, bad);
                if (v == ATOM) launch_pf<ATOM>(s, n_part, e, y0, o, n, part, ticket, out[v], gen, bad);
                if (v == HWT) launch_pf<HWT>(s, n_part, e, y0, o, n, part, ticket, out[v], gen, bad);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(&stale[v], bad, 8, hipMemcpyDeviceToHost));
        }
        printf(",\n \"%lld\": {\"split_P_then_F\": %.3f, \"fused_release_fence\": %.3f, \"fused_write_through\": %.3f, \"fused_atomics\": %.3f, "
               "\"fused_write_through_16_group_tickets\": %.3f, \"gain_write_through_us\": %.3f, \"gain_group_tickets_us\": %.3f, \"sums_equal\": %s, "
               "\"stale_words\": {\"fence\": %llu, \"write_through\": %llu, \"atomics\": %llu, \"group_tickets\": %llu}}",
               (long long)n, med[0], med[1], med[2], med[3], med[4], med[0] - med[2], med[0] - med[4],
               (res[0] == res[1] && res[0] == res[2] && res[0] == res[3] && res[0] == res[4]) ? "true" : "false", stale[1], stale[2],
               stale[3], stale[4]);
        hipFree(e); hipFree(y0); hipFree(y1); hipFree(o); hipFree(junk); hipFree(part); hipFree(ticket); hipFree(bad);
        for (int v = 0; v < 5; ++v) hipFree(out[v]);
    }
    printf("\n}\n");
    return 0;
}
