// sweep_block.hip — workgroup size of the streaming stage-combine (never swept before r03: fixed at 256 = 4 wave64).
// out = y0 + sum_{j<NT} c_j k_j, one 16-byte element per lane, exact-cover grid; HBM-cold (rotating buffer sets whose
// total exceeds the 256 MiB Infinity Cache) and warm (one set), fp32 NT = 5 (7 words, cfg2's dominant launch) and
// fp64 NT = 9 (11 words, cfg4's wide rows).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sweep_block.bin tools/sweep_block.hip && tools/sweep_block.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

template <typename V, int NT>
struct Args {
    V* out;
    const V* y0;
    const V* k[NT];
    float c[NT];
    long ne;
};

template <typename V, int NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void combine(const Args<V, NT> a) {
    const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.ne) return;
    V kk[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) kk[j] = a.k[j][i];
    V acc = kk[0] * (decltype(kk[0].x))a.c[0];
#pragma unroll
    for (int j = 1; j < NT; ++j) acc = acc + kk[j] * (decltype(kk[0].x))a.c[j];
    a.out[i] = a.y0[i] + acc;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename V, int NT, int BLOCK>
double run(std::vector<Args<V, NT>>& sets, int launches) {
    const unsigned grid = (unsigned)((sets[0].ne + BLOCK - 1) / BLOCK);
    for (auto& a : sets) hipLaunchKernelGGL((combine<V, NT, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, a);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> reps;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < launches; ++i)
            hipLaunchKernelGGL((combine<V, NT, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, sets[i % sets.size()]);
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
    const int n_sets = 4;
    std::vector<Args<V, NT>> sets(n_sets);
    for (auto& a : sets) {
        CHECK(hipMalloc((void**)&a.out, bytes));
        void* p;
        CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); a.y0 = (const V*)p;
        for (int j = 0; j < NT; ++j) { CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0, bytes)); a.k[j] = (const V*)p; a.c[j] = 0.1f * (j + 1); }
        a.ne = ne;
    }
    std::vector<Args<V, NT>> one(sets.begin(), sets.begin() + 1);
    const double gb = (double)(NT + 2) * bytes / 1e9;
    printf(" \"%s\": {", name);
    bool first = true;
#define ONE(B)                                                                                                    \
    {                                                                                                             \
        const double cold = run<V, NT, B>(sets, 24), warm = run<V, NT, B>(one, 24);                               \
        printf("%s\"%d\": {\"cold_us\": %.2f, \"cold_TBps\": %.3f, \"warm_us\": %.2f, \"warm_TBps\": %.3f}", first ? "" : ", ", B, \
               cold, gb / cold * 1e3, warm, gb / warm * 1e3);                                                      \
        first = false;                                                                                            \
    }
    ONE(64) ONE(128) ONE(256) ONE(512) ONE(1024)
#undef ONE
    printf("}");
    for (auto& a : sets) { hipFree(a.out); hipFree((void*)a.y0); for (int j = 0; j < NT; ++j) hipFree((void*)a.k[j]); }
}

int main() {
    printf("{\n");
    sweep<f32x4, 5>("fp32 NT=5 (7 words), 8388608 elements", 8388608, 4);
    printf(",\n");
    sweep<f64x2, 9>("fp64 NT=9 (11 words), 8388608 elements", 8388608, 8);
    printf(",\n");
    sweep<f32x4, 2>("fp32 NT=2 (4 words), 8388608 elements", 8388608, 4);
    printf("\n}\n");
    return 0;
}
