// LDS-DMA (global_load_lds_dwordx4) smoke test: per-lane source address, wave-uniform LDS base + lane * 16.
// build + run on the box: hipcc -O3 --offload-arch=gfx950 scripts/microbench/glds_test.hip -o /tmp/glds_test && /tmp/glds_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const u32x4 *in, u32x4 *out, int n)
{
    __shared__ u32x4 buf[2][256];
    const int tid = threadIdx.x, wv = tid >> 6;
    // two DMAs per thread: element tid (reversed source order inside the wave) and a second slot, one kept in flight
    const int src = blockIdx.x * 256 + wv * 64 + (63 - (tid & 63));
    glds16(in + src, __builtin_amdgcn_readfirstlane(lds_addr(&buf[0][wv * 64])));
    glds16(in + (src + 256) % n, __builtin_amdgcn_readfirstlane(lds_addr(&buf[1][wv * 64])));
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); // the first DMA has landed, the second may still be in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const u32x4 a = buf[0][tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const u32x4 b = buf[1][tid];
    out[blockIdx.x * 256 + tid] = u32x4{a.x, a.y, b.x, b.w};
}
int main()
{
    const int n = 256 * 1024;
    std::vector<u32x4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = u32x4{(uint32_t)i, (uint32_t)i * 3u, (uint32_t)i ^ 0x5555u, (uint32_t)i + 7u};
    u32x4 *d_in, *d_out;
    hipMalloc(&d_in, n * sizeof(u32x4));
    hipMalloc(&d_out, n * sizeof(u32x4));
    hipMemcpy(d_in, h.data(), n * sizeof(u32x4), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d_in, d_out, n);
    hipMemcpy(o.data(), d_out, n * sizeof(u32x4), hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        const int blk = i / 256, t = i % 256, wv = t / 64, l = t % 64;
        const int s0 = blk * 256 + wv * 64 + (63 - l), s1 = (s0 + 256) % n;
        if (o[i].x != h[s0].x || o[i].y != h[s0].y || o[i].z != h[s1].x || o[i].w != h[s1].w) ++bad;
    }
    printf("glds test: %ld mismatches of %d\n", bad, n);
    return bad != 0;
}
