// Mixed read/write bandwidth ceilings on this box: plain copies and a 3:2 read:write mix (the pipeline moves 11.3 MB of
// reads and 7.0 MB of writes per frame).  Bytes counted = bytes read + bytes written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kU, bool kNt>
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + (kU - 1) * stride < n16; i += kU * stride) {
        v4u v[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) v[k] = kNt ? __builtin_nontemporal_load(a + i + k * stride) : a[i + k * stride];
#pragma unroll
        for (int k = 0; k < kU; ++k) { if (kNt) __builtin_nontemporal_store(v[k], b + i + k * stride); else b[i + k * stride] = v[k]; }
    }
}
// reads 3 streams, writes 2 (3:2)
template <bool kNt>
__global__ __launch_bounds__(256) void k_mix32(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t third = n16 / 3, half = n16 / 2;
    for (; i < third && i < half; i += stride) {
        v4u x = kNt ? __builtin_nontemporal_load(a + i) : a[i];
        v4u y = kNt ? __builtin_nontemporal_load(a + third + i) : a[third + i];
        v4u z = kNt ? __builtin_nontemporal_load(a + 2 * third + i) : a[2 * third + i];
        v4u s = x + y, t = y ^ z;
        if (kNt) { __builtin_nontemporal_store(s, b + i); __builtin_nontemporal_store(t, b + half + i); }
        else { b[i] = s; b[half + i] = t; }
    }
}
__global__ __launch_bounds__(256) void k_read(const v4u* __restrict__ a, unsigned* out, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    v4u acc = {0, 0, 0, 0};
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(a + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
int main()
{
    const size_t bytes = (size_t)2 << 30; // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
    v4u *a, *b; unsigned* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double moved, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %.2f TB/s\n", name, moved * 5.0 / (ms * 1e-3) / 1e12);
    };
    for (int grid : {256 * 8, 256 * 16, 256 * 32}) {
        printf("grid %d\n", grid);
        time("read only (nt)", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n16); });
        time("copy 1 x 16 B in flight, nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<1, true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        time("copy 4 x 16 B in flight, nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        time("copy 8 x 16 B in flight, nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<8, true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        time("copy 4 x 16 B in flight, default policy", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        time("3 reads : 2 writes, nt", (double)(n16 / 3 * 3 + n16 / 3 * 2) * 16, [&] { hipLaunchKernelGGL((k_mix32<true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        time("3 reads : 2 writes, default policy", (double)(n16 / 3 * 3 + n16 / 3 * 2) * 16, [&] { hipLaunchKernelGGL((k_mix32<false>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
    }
    return 0;
}
