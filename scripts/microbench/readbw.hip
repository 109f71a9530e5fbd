// Pure-read bandwidth microbenchmark: what can a read-only kernel reach on this MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read16(const uint4* __restrict__ p, size_t n16, unsigned* sink)
{
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * 256 < n16; i += (size_t)gridDim.x * 256 * U) {
        v4u v[U];
        const v4u* q = reinterpret_cast<const v4u*>(p);
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(&q[i + k * 256]) : q[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// the order-scan pattern: one dword out of every 32 bytes
template <int U>
__global__ __launch_bounds__(256) void k_read4of32(const unsigned* __restrict__ p, size_t npts, unsigned* sink)
{
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * 256 < npts; i += (size_t)gridDim.x * 256 * U) {
        unsigned v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = p[(i + k * 256) * 8 + 5];
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k];
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main()
{
    const size_t bytes = (size_t)4 << 30;
    uint4* d; unsigned* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-32s %.2f TB/s\n", name, bytes * 5.0 / (ms * 1e-3) / 1e12);
    };
    const size_t n16 = bytes / 16, npts = bytes / 32;
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d\n", grid);
        run("16B/lane U=1", [&] { hipLaunchKernelGGL((k_read16<1, false>), dim3(grid), dim3(256), 0, 0, d, n16, sink); });
        run("16B/lane U=4", [&] { hipLaunchKernelGGL((k_read16<4, false>), dim3(grid), dim3(256), 0, 0, d, n16, sink); });
        run("16B/lane U=8", [&] { hipLaunchKernelGGL((k_read16<8, false>), dim3(grid), dim3(256), 0, 0, d, n16, sink); });
        run("16B/lane U=4 nontemporal", [&] { hipLaunchKernelGGL((k_read16<4, true>), dim3(grid), dim3(256), 0, 0, d, n16, sink); });
        run("4B of 32B  U=4", [&] { hipLaunchKernelGGL((k_read4of32<4>), dim3(grid), dim3(256), 0, 0, (const unsigned*)d, npts, sink); });
        run("4B of 32B  U=8", [&] { hipLaunchKernelGGL((k_read4of32<8>), dim3(grid), dim3(256), 0, 0, (const unsigned*)d, npts, sink); });
    }
    return 0;
}
