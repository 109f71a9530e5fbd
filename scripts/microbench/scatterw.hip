// Sparse-store microbenchmark: one store every `gap` points of a cold 1 GiB array of 32 B points —
// the pattern of the phase-C label patch (k_ground_resolve).  Store sizes: 2 B (label), 4 B, 16 B (aligned half),
// 32 B (the whole point = full sector), alone and beside a streaming copy on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int BYTES>
__global__ __launch_bounds__(256) void k_sparse(char* p, size_t n_points, unsigned gap)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // pseudo-random but deterministic target inside its own window of `gap` points
    const size_t tgt = i * gap + (i * 2654435761u) % gap;
    if (tgt >= n_points) return;
    char* q = p + tgt * 32;
    if (BYTES == 2) *reinterpret_cast<uint16_t*>(q + 28) = (uint16_t)i;
    if (BYTES == 4) *reinterpret_cast<uint32_t*>(q + 28) = (uint32_t)i;
    if (BYTES == 16) *reinterpret_cast<v4u*>(q + 16) = v4u{(unsigned)i, 1, 2, 3};
    if (BYTES == 32) { *reinterpret_cast<v4u*>(q) = v4u{(unsigned)i, 1, 2, 3}; *reinterpret_cast<v4u*>(q + 16) = v4u{(unsigned)i, 4, 5, 6}; }
}
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ s, v4u* __restrict__ d, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n_points = bytes / 32;
    char *d, *src, *dst;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 0, bytes); (void)hipMemset(src, 1, bytes);
    hipStream_t s1, s2; (void)hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, -1); (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, 0);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const unsigned gap = 16; // one patched point in 16 (the benchmark: 7.9 k reverted of 133 k slots)
    const size_t n_st = n_points / gap;
    auto flush = [&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, s1, (const v4u*)src, (v4u*)dst, bytes / 16); (void)hipStreamSynchronize(s1); };
    auto run = [&](const char* name, auto launch, bool loaded) {
        flush();
        if (loaded) for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, s2, (const v4u*)src, (v4u*)dst, bytes / 16);
        (void)hipEventRecord(a, s1); launch(); (void)hipEventRecord(b, s1); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        (void)hipDeviceSynchronize();
        printf("%-28s %s  %8.1f us  %6.1f G stores/s\n", name, loaded ? "beside a copy" : "alone        ", ms * 1e3, n_st / (ms * 1e-3) / 1e9);
    };
    const unsigned grid = (unsigned)((n_st + 255) / 256);
    for (int loaded = 0; loaded < 2; ++loaded) {
        run("2 B  (label @28)", [&] { hipLaunchKernelGGL(k_sparse<2>, dim3(grid), dim3(256), 0, s1, d, n_points, gap); }, loaded);
        run("4 B", [&] { hipLaunchKernelGGL(k_sparse<4>, dim3(grid), dim3(256), 0, s1, d, n_points, gap); }, loaded);
        run("16 B (hi half)", [&] { hipLaunchKernelGGL(k_sparse<16>, dim3(grid), dim3(256), 0, s1, d, n_points, gap); }, loaded);
        run("32 B (whole point)", [&] { hipLaunchKernelGGL(k_sparse<32>, dim3(grid), dim3(256), 0, s1, d, n_points, gap); }, loaded);
    }
    return 0;
}
