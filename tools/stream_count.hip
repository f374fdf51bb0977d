// stream_count.hip — what the MI355X's memory system sustains for R read streams + W write streams, HBM-cold.
//
// r04's question: the dominant launches run at 0.67 of the 8 TB/s peak when every byte comes from DRAM (cfg2's
// stage_combine_multi<float,4>: 5 reads + 2 writes; cfg4's stage_combine_multi<double,9>: 10 reads + 4 writes) while
// a float4 copy (1 read + 1 write) reaches 0.79.  Block size, occupancy, lane/wave/block-contiguous access and
// small shapes were swept in r03/r04 (all flat).  Not isolated before: (a) the achievable rate AS A FUNCTION OF THE
// STREAM COUNT on this box, (b) all streams carved from one slab with a per-stream base skew (j * 4 KiB + j * 256 B)
// against streams whose bases are congruent modulo every power of two up to 2 MiB (what torch.empty of equal sizes
// gives), (c) non-temporal stores on the outputs (and non-temporal loads).
//
// Every kernel: out_w[i] = y[i] + sum_r c_{w,r} * in_r[i], one 16-byte element per lane, exact-cover grid, 256
// lanes per workgroup (the shipped geometry).  R = 0 writes constants (write-only); W = 0 folds the sum into a
// value that is stored only if it is NaN (read-only).  Rotating buffer sets whose total is >= 1 GB (4x the 256 MiB
// Infinity Cache); median of 5 repetitions of 24 launches, HIP events around the batch.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/stream_count.hip -o tools/stream_count.bin
//   tools/stream_count.bin > gpurun_out/r05_stream_count.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxR = 11, kMaxW = 4;

template <typename E>
struct Args {
    const E* in[kMaxR];
    E* out[kMaxW];
    long long ne;     // 16-byte elements per stream
};

// POLICY bit 0: non-temporal loads, bit 1: non-temporal stores
template <typename E, typename S, int R, int W, int POLICY>
__global__ __launch_bounds__(256) void streams(const Args<E> a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.ne) return;
    E v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if constexpr (POLICY & 1) v[r] = __builtin_nontemporal_load(a.in[r] + i);
        else v[r] = a.in[r][i];
    }
    if constexpr (W == 0) {
        E acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) acc = acc + v[r] * (S)(0.25 + r);
        if (acc.x != acc.x) const_cast<E*>(a.in[0])[i] = acc;     // never taken (inputs are finite)
    } else {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            E acc;
            if constexpr (R == 0) {
                acc = E{(S)(w + 1)};
            } else {
                acc = v[0];
#pragma unroll
                for (int r = 1; r < R; ++r) acc = acc + v[r] * (S)(0.125 * (w + 1) + r);
            }
            if constexpr (POLICY & 2) __builtin_nontemporal_store(acc, a.out[w] + i);
            else a.out[w][i] = acc;
        }
    }
}

struct Slab {
    char* base = nullptr;
    size_t size = 0, used = 0;
};

// LAYOUT 0: every stream its own hipMalloc (what torch.empty of equal sizes gives: bases congruent mod 2 MiB)
//        1: one slab, streams back to back, each base padded up to a 2 MiB boundary (congruent by construction)
//        2: one slab, stream j (counted over the whole slab) displaced by j * 4 KiB + j * 256 B
template <typename E>
static std::vector<Args<E>> make_sets(int R, int W, long long ne, int n_sets, int layout, std::vector<void*>& owned) {
    const size_t bytes = (size_t)ne * 16;
    std::vector<Args<E>> sets(n_sets);
    Slab slab;
    const size_t two_mib = 2u << 20;
    if (layout != 0) {
        slab.size = (size_t)n_sets * (R + W) * (((bytes + two_mib - 1) / two_mib) * two_mib + two_mib) + two_mib;
        CK(hipMalloc((void**)&slab.base, slab.size));
        owned.push_back(slab.base);
        slab.used = (two_mib - ((size_t)slab.base % two_mib)) % two_mib;
    }
    int j = 0;
    auto take = [&]() -> void* {
        void* p;
        if (layout == 0) {
            CK(hipMalloc(&p, bytes));
            owned.push_back(p);
        } else {
            slab.used = ((slab.used + two_mib - 1) / two_mib) * two_mib;
            size_t skew = layout == 2 ? ((size_t)(j % 64) * 4096 + (size_t)(j % 16) * 256) : 0;
            p = slab.base + slab.used + skew;
            slab.used += bytes + skew;
            if (slab.used > slab.size) { fprintf(stderr, "slab overflow\n"); exit(1); }
        }
        ++j;
        CK(hipMemset(p, 0, bytes));
        return p;
    };
    for (auto& a : sets) {
        for (int r = 0; r < R; ++r) a.in[r] = (const E*)take();
        for (int w = 0; w < W; ++w) a.out[w] = (E*)take();
        a.ne = ne;
    }
    return sets;
}

template <typename E, typename S, int R, int W, int POLICY>
static double run(std::vector<Args<E>>& sets, int launches) {
    const unsigned grid = (unsigned)((sets[0].ne + 255) / 256);
    for (auto& a : sets) hipLaunchKernelGGL((streams<E, S, R, W, POLICY>), dim3(grid), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> reps;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < launches; ++i)
            hipLaunchKernelGGL((streams<E, S, R, W, POLICY>), dim3(grid), dim3(256), 0, 0, sets[i % sets.size()]);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        reps.push_back(1e3 * ms / launches);
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    std::sort(reps.begin(), reps.end());
    return reps[2];
}

static bool first_entry = true;

template <typename E, typename S, int R, int W>
static void one_shape(const char* dtype, long long n_scalar, int word) {
    const long long ne = n_scalar * word / 16;
    const size_t bytes = (size_t)ne * 16;
    const size_t per_set = (size_t)(R + W) * bytes;
    int n_sets = (int)std::max<size_t>(2, ((size_t)1 << 30) / per_set + 1);
    const double gb = (double)per_set / 1e9;
    printf("%s  {\"dtype\": \"%s\", \"reads\": %d, \"writes\": %d, \"elements\": %lld, \"bytes_per_launch\": %zu, "
           "\"buffer_sets\": %d", first_entry ? "" : ",\n", dtype, R, W, n_scalar, per_set, n_sets);
    first_entry = false;
    const char* layout_name[3] = {"separate_allocations", "slab_aligned_2MiB", "slab_skewed_4KiB_256B"};
    for (int layout = 0; layout < 3; ++layout) {
        std::vector<void*> owned;
        auto sets = make_sets<E>(R, W, ne, n_sets, layout, owned);
        const double t0 = run<E, S, R, W, 0>(sets, 24);
        printf(", \"%s\": {\"us\": %.2f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f", layout_name[layout], t0, gb / t0 * 1e3,
               gb / t0 * 1e3 / 8.0);
        if (layout != 1) {
            if (W > 0) {
                const double t2 = run<E, S, R, W, 2>(sets, 24);
                printf(", \"nt_store_us\": %.2f, \"nt_store_TBps\": %.3f", t2, gb / t2 * 1e3);
            }
            if (R > 0) {
                const double t1 = run<E, S, R, W, 1>(sets, 24);
                printf(", \"nt_load_us\": %.2f, \"nt_load_TBps\": %.3f", t1, gb / t1 * 1e3);
            }
            if (R > 0 && W > 0) {
                const double t3 = run<E, S, R, W, 3>(sets, 24);
                printf(", \"nt_both_us\": %.2f, \"nt_both_TBps\": %.3f", t3, gb / t3 * 1e3);
            }
        }
        printf("}");
        for (void* p : owned) CK(hipFree(p));
    }
    printf("}");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"what\": \"R read + W write streams of 16-B elements, one per lane, exact-cover "
           "grid of 256-lane workgroups, rotating buffer sets >= 1 GB in total (HBM-cold); median of 5 x 24 launches\",\n"
           " \"curves\": [\n", prop.name, prop.multiProcessorCount);
    const long long N = 8388608;
    // fp32, 33.5 MB per stream
    one_shape<f32x4, float, 1, 1>("f32", N, 4);     // the guide's float4 copy
    one_shape<f32x4, float, 1, 0>("f32", N, 4);     // read-only
    one_shape<f32x4, float, 0, 1>("f32", N, 4);     // write-only
    one_shape<f32x4, float, 2, 1>("f32", N, 4);     // stage row 0 (y0 + c k0)
    one_shape<f32x4, float, 3, 1>("f32", N, 4);
    one_shape<f32x4, float, 4, 1>("f32", N, 4);
    one_shape<f32x4, float, 5, 1>("f32", N, 4);
    one_shape<f32x4, float, 6, 1>("f32", N, 4);     // r02's dominant launch: 5 k + y0 -> y
    one_shape<f32x4, float, 5, 2>("f32", N, 4);     // cfg2's dominant launch: 4 k + y0 -> y, prefix
    one_shape<f32x4, float, 6, 0>("f32", N, 4);     // read-only, 6 streams
    one_shape<f32x4, float, 8, 1>("f32", N, 4);
    // fp64, 67 MB per stream
    one_shape<f64x2, double, 1, 1>("f64", N, 8);
    one_shape<f64x2, double, 5, 2>("f64", N, 8);
    one_shape<f64x2, double, 10, 1>("f64", N, 8);
    one_shape<f64x2, double, 10, 4>("f64", N, 8);   // cfg4's dominant launch: 9 k + y0 -> 4 outputs
    one_shape<f64x2, double, 10, 0>("f64", N, 8);
    printf("\n ]}\n");
    return 0;
}
