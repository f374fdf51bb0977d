// sweep_small.hip — the per-kernel floor of the streaming stage-combine at SHARD size (r04b).
// An 8-GPU strong-scaling shard of cfg2 is 8192 x 128 = 2^20 fp32 elements; there every solver kernel of the captured
// step costs ~5 us whatever it moves (docs/LAB_NOTEBOOK.md §7: 8 kernels x 5.0-5.7 us, data Infinity-Cache / L2 resident), and that
// floor — not bandwidth — is what caps strong scaling at 4.2x.  This sweeps what could move the floor for
// out = y0 + sum_{j<NT} c_j k_j (NT = 1: 3 words, NT = 5: 7 words per element):
//   * elements per lane E = 1 / 2 / 4 / 8 16-byte vectors (grid = N / (BLOCK * E * 4): 1024 ... 128 workgroups),
//     consecutive vectors strided by the workgroup's span so every load stays coalesced,
//   * workgroup size 64 ... 1024,
// measured (a) as a chain of dependent launches on one stream (event-timed, 2 rotating buffer sets, warm) and (b) as
// nodes of ONE captured hipGraph replayed — the in-situ condition of the captured step.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sweep_small.bin tools/sweep_small.hip && tools/sweep_small.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
struct Args {
    f32x4* out;
    const f32x4* y0;
    const f32x4* k[NT];
    float c[NT];
    long nv;        // number of 16-byte vectors
};

template <int NT, int BLOCK, int E>
__global__ __launch_bounds__(BLOCK) void combine(const Args<NT> a) {
    const long base = (long)blockIdx.x * (BLOCK * E) + threadIdx.x;
    f32x4 kk[E][NT], y[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long i = base + (long)e * BLOCK;
        if (i < a.nv) {
            y[e] = a.y0[i];
#pragma unroll
            for (int j = 0; j < NT; ++j) kk[e][j] = a.k[j][i];
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long i = base + (long)e * BLOCK;
        if (i < a.nv) {
            f32x4 acc = kk[e][0] * a.c[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) acc = acc + kk[e][j] * a.c[j];
            a.out[i] = y[e] + acc;
        }
    }
}

template <int NT, int BLOCK, int E>
void launch(const Args<NT>& a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.nv + (long)BLOCK * E - 1) / ((long)BLOCK * E));
    hipLaunchKernelGGL((combine<NT, BLOCK, E>), dim3(grid), dim3(BLOCK), 0, s, a);
}

template <int NT, int BLOCK, int E>
void measure(std::vector<Args<NT>>& sets, hipStream_t s, const char* tag, bool first) {
    const int chain = 200;
    for (int i = 0; i < 8; ++i) launch<NT, BLOCK, E>(sets[i % 2], s);
    CHECK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> stream_us, graph_us;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < chain; ++i) launch<NT, BLOCK, E>(sets[i % 2], s);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        stream_us.push_back(1e3 * ms / chain);
    }
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i) launch<NT, BLOCK, E>(sets[i % 2], s);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipGraphLaunch(exec, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        graph_us.push_back(1e3 * ms / chain);
    }
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    std::sort(stream_us.begin(), stream_us.end());
    std::sort(graph_us.begin(), graph_us.end());
    const long nv = sets[0].nv;
    const unsigned grid = (unsigned)((nv + (long)BLOCK * E - 1) / ((long)BLOCK * E));
    const double mb = (NT + 2) * nv * 16.0 / 1e6;
    printf("%s{\"case\": \"%s\", \"words\": %d, \"block\": %d, \"vectors_per_lane\": %d, \"workgroups\": %u, \"mbytes\": %.1f, "
           "\"stream_us\": %.2f, \"graph_us\": %.2f, \"graph_gb_s\": %.0f}",
           first ? "" : ",\n  ", tag, NT + 2, BLOCK, E, grid, mb, stream_us[2], graph_us[2], mb / graph_us[2] * 1e3);
}

template <int NT>
void sweep(long n_elem, hipStream_t s, const char* tag, bool& first) {
    std::vector<Args<NT>> sets(2);
    const long nv = n_elem / 4;
    for (auto& a : sets) {
        a.nv = nv;
        CHECK(hipMalloc(&a.out, nv * 16));
        f32x4* p;
        CHECK(hipMalloc(&p, nv * 16)); CHECK(hipMemset(p, 0, nv * 16)); a.y0 = p;
        for (int j = 0; j < NT; ++j) {
            CHECK(hipMalloc(&p, nv * 16)); CHECK(hipMemset(p, 0, nv * 16)); a.k[j] = p;
            a.c[j] = 0.1f * (j + 1);
        }
    }
    measure<NT, 256, 1>(sets, s, tag, first); first = false;
    measure<NT, 256, 2>(sets, s, tag, false);
    measure<NT, 256, 4>(sets, s, tag, false);
    measure<NT, 256, 8>(sets, s, tag, false);
    measure<NT, 64, 1>(sets, s, tag, false);
    measure<NT, 128, 1>(sets, s, tag, false);
    measure<NT, 512, 1>(sets, s, tag, false);
    measure<NT, 1024, 1>(sets, s, tag, false);
    measure<NT, 1024, 2>(sets, s, tag, false);
    measure<NT, 512, 2>(sets, s, tag, false);
    measure<NT, 128, 4>(sets, s, tag, false);
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    bool first = true;
    printf("{\"tool\": \"tools/sweep_small.hip\", \"what\": \"stage-combine launch at shard size: us per launch as a dependent chain on "
           "one stream and as nodes of one replayed hipGraph (200 launches, median of 5), fp32\", \"results\": [\n  ");
    sweep<1>(1L << 20, s, "2^20 elements (1/8 shard), 3 words", first);
    sweep<5>(1L << 20, s, "2^20 elements (1/8 shard), 7 words", first);
    sweep<1>(1L << 21, s, "2^21 elements (1/4 shard), 3 words", first);
    sweep<5>(1L << 21, s, "2^21 elements (1/4 shard), 7 words", first);
    printf("\n]}\n");
    return 0;
}
