// Write-pattern microbenchmark: contiguous 16 B/lane vs. the AoS pattern (two 16 B stores per lane at a 32 B stride)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_w_contig(v4u* p, size_t n16)
{   // each thread: 2 stores, wave-contiguous 1 KiB each
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t wave = i >> 6, lane = i & 63;
    for (size_t w = wave; w * 128 + 127 < n16; w += (size_t)gridDim.x * 4) {
        v4u v = {(unsigned)w, 1, 2, 3};
        p[w * 128 + lane] = v;
        p[w * 128 + 64 + lane] = v;
    }
}
__global__ __launch_bounds__(256) void k_w_aos(v4u* p, size_t n16)
{   // each thread owns a 32 B struct: store lo half, then hi half (16 B at 32 B stride)
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t wave = i >> 6, lane = i & 63;
    for (size_t w = wave; w * 128 + 127 < n16; w += (size_t)gridDim.x * 4) {
        v4u v = {(unsigned)w, 1, 2, 3};
        p[w * 128 + 2 * lane] = v;
        p[w * 128 + 2 * lane + 1] = v;
    }
}
__global__ __launch_bounds__(256) void k_w_aos_misaligned(v4u* p, size_t n16)
{   // same, but every wave's 2 KiB region starts 96 B into a 128 B line (odd row stride of the range image)
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t wave = i >> 6, lane = i & 63;
    for (size_t w = wave; w * 128 + 127 + 6 < n16; w += (size_t)gridDim.x * 4) {
        v4u v = {(unsigned)w, 1, 2, 3};
        p[w * 128 + 6 + 2 * lane] = v;
        p[w * 128 + 6 + 2 * lane + 1] = v;
    }
}
int main()
{
    const size_t bytes = (size_t)4 << 30; v4u* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-40s %.2f TB/s\n", name, bytes * 5.0 / (ms * 1e-3) / 1e12);
    };
    const size_t n16 = bytes / 16;
    for (int grid : {2048, 16384}) {
        printf("grid %d\n", grid);
        run("contiguous 16B/lane", [&] { hipLaunchKernelGGL(k_w_contig, dim3(grid), dim3(256), 0, 0, d, n16); });
        run("AoS 2x16B at 32B stride", [&] { hipLaunchKernelGGL(k_w_aos, dim3(grid), dim3(256), 0, 0, d, n16); });
        run("AoS, region off by 96B", [&] { hipLaunchKernelGGL(k_w_aos_misaligned, dim3(grid), dim3(256), 0, 0, d, n16); });
    }
    return 0;
}
