// sweep_occupancy.hip — r04: the one lever the earlier sweeps (launch geometry, unroll, cache policy, stream staggering,
// workgroup size) had not touched for the WIDE rows (dopri8 / fp64, 9-11 streams, 0.64-0.67 of the HBM peak): how many
// wavefronts are resident per CU.  Fewer resident workgroups = fewer DRAM pages open at once per stream; more = more
// bytes in flight.  Occupancy is limited here by a dynamic LDS allocation the kernel never touches (160 KB per CU).
// Second lever: persistent workgroups that each stream through ONE contiguous span of every array ("span") instead of the
// exact-cover grid ("cover").  HBM-cold (4 rotating buffer sets).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sweep_occupancy.bin tools/sweep_occupancy.hip && tools/sweep_occupancy.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int BLOCK = 256;

template <typename V, int NT>
struct Args {
    V* out;
    const V* y0;
    const V* k[NT];
    float c[NT];
    long ne;
};

template <typename V, int NT>
__device__ __forceinline__ void one(const Args<V, NT>& a, long i) {
    V kk[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) kk[j] = a.k[j][i];
    V acc = kk[0] * (decltype(kk[0].x))a.c[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) acc = acc + kk[j] * (decltype(kk[0].x))a.c[j];
    a.out[i] = a.y0[i] + acc;
}

template <typename V, int NT>
__global__ __launch_bounds__(BLOCK) void cover(const Args<V, NT> a) {
    extern __shared__ char lds_[];
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < a.ne) one<V, NT>(a, i);
}

// persistent: workgroup w owns elements [w * span, (w + 1) * span), walked in BLOCK-wide tiles
template <typename V, int NT>
__global__ __launch_bounds__(BLOCK) void span(const Args<V, NT> a, long span_len) {
    extern __shared__ char lds_[];
    const long lo = (long)blockIdx.x * span_len;
    const long hi = lo + span_len < a.ne ? lo + span_len : a.ne;
    for (long i = lo + threadIdx.x; i < hi; i += BLOCK) one<V, NT>(a, i);
}

template <typename V, int NT>
double run(std::vector<Args<V, NT>>& sets, int launches, size_t lds, int persistent_per_cu) {
    const long ne = sets[0].ne;
    unsigned grid = (unsigned)((ne + BLOCK - 1) / BLOCK);
    long span_len = 0;
    if (persistent_per_cu > 0) {
        grid = 256u * persistent_per_cu;
        span_len = ((ne + grid - 1) / grid + BLOCK - 1) / BLOCK * BLOCK;
    }
    auto launch = [&](const Args<V, NT>& a) {
        if (persistent_per_cu > 0) hipLaunchKernelGGL((span<V, NT>), dim3(grid), dim3(BLOCK), lds, 0, a, span_len);
        else hipLaunchKernelGGL((cover<V, NT>), dim3(grid), dim3(BLOCK), lds, 0, a);
    };
    for (auto& a : sets) launch(a);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> reps;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < launches; ++i) launch(sets[i % sets.size()]);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        reps.push_back(1e3 * ms / launches);
    }
    std::sort(reps.begin(), reps.end());
    return reps[2];
}

template <typename V, int NT>
void sweep(const char* name, long n_elem_scalar, int word) {
    const long ne = n_elem_scalar * word / 16;
    const size_t bytes = (size_t)ne * 16;
    std::vector<Args<V, NT>> sets(4);
    for (auto& a : sets) {
        CHECK(hipMalloc((void**)&a.out, bytes));
        void* p;
        CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); a.y0 = (const V*)p;
        for (int j = 0; j < NT; ++j) { CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); a.k[j] = (const V*)p; a.c[j] = 0.1f * (j + 1); }
        a.ne = ne;
    }
    const double gb = (double)(NT + 2) * bytes / 1e9;
    printf(" \"%s\": {\n", name);
    const size_t lds_sizes[] = {0, 20 * 1024, 32 * 1024, 40 * 1024, 53 * 1024, 64 * 1024};      // 64 KB = the per-workgroup limit
    bool first = true;
    for (size_t lds : lds_sizes) {
        const double us = run<V, NT>(sets, 24, lds, 0);
        printf("%s  \"cover lds=%zuK (<=%d workgroups per CU)\": {\"cold_us\": %.2f, \"cold_TBps\": %.3f}", first ? "" : ",\n", lds / 1024,
               lds ? (int)(160 * 1024 / lds) : 8, us, gb / us * 1e3);
        first = false;
    }
    for (int per_cu : {1, 2, 4, 8}) {
        const double us = run<V, NT>(sets, 24, 0, per_cu);
        printf(",\n  \"span %d workgroups per CU\": {\"cold_us\": %.2f, \"cold_TBps\": %.3f}", per_cu, us, gb / us * 1e3);
    }
    printf("\n }");
    for (auto& a : sets) { hipFree(a.out); hipFree((void*)a.y0); for (int j = 0; j < NT; ++j) hipFree((void*)a.k[j]); }
}

int main() {
    printf("{\n");
    sweep<f64x2, 9>("fp64 NT=9 (11 words), 8388608 elements", 8388608, 8);
    printf(",\n");
    sweep<f32x4, 5>("fp32 NT=5 (7 words), 8388608 elements", 8388608, 4);
    printf("\n}\n");
    return 0;
}
