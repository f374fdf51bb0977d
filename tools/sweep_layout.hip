// Access-pattern sweep for the multi-stream combine (out = y0 + sum_j c_j k_j) on the MI355X, COLD
// (rotating buffer sets larger than the 256 MiB Infinity Cache):
//   A  one 16-B element per lane, exact-cover grid                      (the shipped stage_combine geometry)
//   B  V consecutive 16-B elements per lane  (lane-contiguous 32/64 B)
//   C  U wave-contiguous 16-B elements per lane (stride 64 lanes: a wave covers U KiB per stream contiguously)
//   D  U block-contiguous elements per lane (stride 256: a workgroup covers 4*U KiB per stream contiguously)
// each with the stream base addresses either exactly N*sizeof(T) apart or staggered by an odd multiple of 4 KiB.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/sweep_layout.hip -o tools/sweep_layout.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename E, typename S, int NT>
struct Args {
    E* out;
    const E* y0;
    const E* k[NT];
    S c[NT];
    long long ne;   // 16-B elements
};

// MODE 0: A (U ignored)   1: lane-contiguous   2: wave-contiguous   3: block-contiguous
template <typename E, typename S, int NT, int MODE, int U>
__global__ __launch_bounds__(256) void combine(const Args<E, S, NT> a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long base;
    long long step;
    if (MODE == 0) { base = (long long)blockIdx.x * 256 + threadIdx.x; step = 0; }
    else if (MODE == 1) { base = ((long long)blockIdx.x * 256 + threadIdx.x) * U; step = 1; }
    else if (MODE == 2) { base = ((long long)blockIdx.x * 4 + wave) * (64LL * U) + lane; step = 64; }
    else { base = (long long)blockIdx.x * (256LL * U) + threadIdx.x; step = 256; }
    constexpr int UU = MODE == 0 ? 1 : U;
    E y[UU], kk[UU][NT];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const long long i = base + u * step;
        if (i < a.ne) {
            y[u] = a.y0[i];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[u][j] = a.k[j][i];
        }
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        const long long i = base + u * step;
        if (i < a.ne) {
            E acc = kk[u][0] * a.c[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) acc = acc + kk[u][j] * a.c[j];
            a.out[i] = y[u] + acc;
        }
    }
}

template <typename E, typename S, int NT, int MODE, int U>
float run(const std::vector<Args<E, S, NT>>& sets, int iters) {
    const long long ne = sets[0].ne;
    const long long per_block = 256LL * (MODE == 0 ? 1 : U);
    const unsigned grid = (unsigned)((ne + per_block - 1) / per_block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((combine<E, S, NT, MODE, U>), dim3(grid), dim3(256), 0, 0, sets[i % sets.size()]);
    hipDeviceSynchronize();
    std::vector<float> ms(iters);
    for (int i = 0; i < iters; ++i) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((combine<E, S, NT, MODE, U>), dim3(grid), dim3(256), 0, 0, sets[i % sets.size()]);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[i], e0, e1);
    }
    std::sort(ms.begin(), ms.end());
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms[iters / 2];
}

template <typename E, typename S, int NT>
int sweep(const char* label, long long n_scalars, size_t stagger_bytes) {
    const int n_sets = 4;
    const size_t esz = sizeof(E), bytes_per = (size_t)n_scalars * sizeof(S);
    const long long ne = (long long)(bytes_per / esz);
    std::vector<Args<E, S, NT>> sets(n_sets);
    for (int s = 0; s < n_sets; ++s) {
        char* p;
        const size_t slot = bytes_per + stagger_bytes;
        CK(hipMalloc(&p, slot * (NT + 2) + (1 << 20)));
        CK(hipMemset(p, 0, slot * (NT + 2) + (1 << 20)));
        sets[s].out = reinterpret_cast<E*>(p);
        sets[s].y0 = reinterpret_cast<const E*>(p + slot);
        for (int j = 0; j < NT; ++j) {
            sets[s].k[j] = reinterpret_cast<const E*>(p + slot * (2 + j));
            sets[s].c[j] = (S)(0.1 * (j + 1));
        }
        sets[s].ne = ne;
    }
    const double bytes = double(NT + 2) * bytes_per;
    printf("%s NT=%d stagger=%zu B  (%.1f MB per launch)\n", label, NT, stagger_bytes, bytes / 1e6);
#define R(MODE, U, NAME) { const float t = run<E, S, NT, MODE, U>(sets, 30); printf("   %-34s %8.1f us %8.1f GB/s %5.1f%%\n", NAME, t * 1e3, bytes / t / 1e6, bytes / t / 1e6 / 80.0); }
    R(0, 1, "A  1 elem/lane")
    R(1, 2, "B  lane-contiguous x2")
    R(1, 4, "B  lane-contiguous x4")
    R(2, 2, "C  wave-contiguous x2")
    R(2, 4, "C  wave-contiguous x4")
    R(2, 8, "C  wave-contiguous x8")
    R(3, 2, "D  block-contiguous x2")
    R(3, 4, "D  block-contiguous x4")
#undef R
    for (int s = 0; s < n_sets; ++s) CK(hipFree(sets[s].out));
    return 0;
}

int main() {
    const long long N = 65536LL * 128;
    if (sweep<f32x4, float, 5>("fp32", N, 0)) return 1;
    if (sweep<f32x4, float, 5>("fp32", N, 4096 * 5)) return 1;
    if (sweep<f32x4, float, 5>("fp32", N, 4096 * 67)) return 1;
    if (sweep<f32x4, float, 2>("fp32", N, 0)) return 1;
    if (sweep<f64x2, double, 9>("fp64", N, 0)) return 1;
    if (sweep<f64x2, double, 9>("fp64", N, 4096 * 67)) return 1;
    return 0;
}
