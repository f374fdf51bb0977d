// Tuning sweep for the dominant kernel (stage_combine, NT = 5, fp32, cfg2 size) on the MI355X:
// launch geometry x unroll x cache policy, cold (rotating buffer sets > Infinity Cache) and warm.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Itorchdiffeq_amd/csrc tools/sweep_combine.hip -o /tmp/sweep && /tmp/sweep
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "tdeq_kernels.hpp"

using namespace tdeq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NT = 5;
constexpr int64_t N = 65536LL * 128;

template <int U, int POLICY>
float run(const std::vector<CombineArgs<float, NT>>& sets, int grid, int iters, bool rotate) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((stage_combine_kernel<float, NT, U, true, POLICY>), dim3(grid), dim3(kBlock), 0, 0, sets[i % sets.size()]);
    hipDeviceSynchronize();
    std::vector<float> ms(iters);
    for (int i = 0; i < iters; ++i) {
        const auto& a = sets[rotate ? (i % sets.size()) : 0];
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stage_combine_kernel<float, NT, U, true, POLICY>), dim3(grid), dim3(kBlock), 0, 0, a);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[i], e0, e1);
    }
    std::sort(ms.begin(), ms.end());
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms[iters / 2];
}

int main() {
    const int n_sets = 4;
    std::vector<CombineArgs<float, NT>> sets(n_sets);
    for (int s = 0; s < n_sets; ++s) {
        float* p;
        CK(hipMalloc(&p, sizeof(float) * N * (NT + 2)));
        CK(hipMemset(p, 0, sizeof(float) * N * (NT + 2)));
        sets[s].out = p;
        sets[s].y0 = p + N;
        for (int j = 0; j < NT; ++j) {
            sets[s].k[j] = p + N * (2 + j);
            sets[s].c[j] = 0.1f * (j + 1);
        }
        sets[s].n = N;
    }
    const double bytes = double(NT + 2) * N * 4;
    const int64_t nv = N / 4;
    printf("stage_combine<float,%d> N=%lld  algorithmic bytes %.1f MB\n", NT, (long long)N, bytes / 1e6);
    printf("%-8s %-6s %-8s %10s %10s %10s %10s\n", "U", "policy", "grid", "cold_us", "cold_GB/s", "warm_us", "warm_GB/s");
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384};
#define SWEEP(U_, P_)                                                                              \
    for (int g : grids) {                                                                          \
        const int64_t need = (nv + 256LL * U_ - 1) / (256LL * U_);                                 \
        const int grid = (int)std::min<int64_t>(g, need);                                          \
        const float cold = run<U_, P_>(sets, grid, 40, true);                                      \
        const float warm = run<U_, P_>(sets, grid, 40, false);                                     \
        printf("%-8d %-6d %-8d %10.1f %10.1f %10.1f %10.1f\n", U_, P_, grid, cold * 1e3,           \
               bytes / cold / 1e6, warm * 1e3, bytes / warm / 1e6);                                \
    }
    SWEEP(1, 0) SWEEP(2, 0) SWEEP(4, 0)
    SWEEP(1, 1) SWEEP(2, 1)
    SWEEP(1, 2) SWEEP(2, 2)
    SWEEP(1, 3) SWEEP(2, 3) SWEEP(4, 3)
    return 0;
}
