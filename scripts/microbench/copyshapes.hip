// What a byte-moving kernel reaches on this box, by SHAPE (the guide quotes 6.29 TB/s for a float4
// copy; scripts/microbench/copybw.hip's grid-stride copies reached 4.7-5.1).  Bytes counted = bytes read + bytes written.
//   simple      one 16-B load and store per thread, no loop, grid = n / 256
//   blocked     every workgroup copies ONE contiguous chunk, kU loads in flight per thread
//   gridstride  a fixed grid strides over the buffer (copybw.hip's shape)
//   walk        the column walk's geometry without its arithmetic: a 256-thread workgroup moves 64 rows of 256 x 32 B
//               (8 KiB per row, rows 66,656 B apart = HDL_64E), kD rows in flight, 32-B points as two 16-B halves per lane
//               (as they sit in the AoS) or transposed into whole lines first
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool kNtL, bool kNtS>
__global__ __launch_bounds__(256) void k_simple(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u v = kNtL ? __builtin_nontemporal_load(a + i) : a[i];
    if (kNtS) __builtin_nontemporal_store(v, b + i); else b[i] = v;
}
template <int kT, int kU, bool kNtL, bool kNtS>
__global__ __launch_bounds__(kT) void k_blocked(const v4u* __restrict__ a, v4u* __restrict__ b, size_t chunk16)
{
    const size_t base = (size_t)blockIdx.x * chunk16;
    for (size_t i = threadIdx.x; i < chunk16; i += (size_t)kU * kT) {
        v4u v[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) v[k] = kNtL ? __builtin_nontemporal_load(a + base + i + k * kT) : a[base + i + k * kT];
#pragma unroll
        for (int k = 0; k < kU; ++k) { if (kNtS) __builtin_nontemporal_store(v[k], b + base + i + k * kT); else b[base + i + k * kT] = v[k]; }
    }
}
template <int kU, bool kNt>
__global__ __launch_bounds__(256) void k_gridstride(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n16)
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
__global__ __launch_bounds__(256) void k_read(const v4u* __restrict__ a, unsigned* out, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    v4u acc = {0, 0, 0, 0};
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(a + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_write(v4u* __restrict__ b, size_t n16)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    const v4u v = {1u, 2u, 3u, (unsigned)i};
    for (; i < n16; i += stride) __builtin_nontemporal_store(v, b + i);
}
// the walk's geometry: frame f = S points of 32 B, strip = 252 columns (+4 halo read, not written), rows H points apart
template <int kD, bool kXpose, bool kNtL>
__global__ __launch_bounds__(256) void k_walk(const v4u* __restrict__ in, v4u* __restrict__ out, int nf, int strips, int N, int H)
{
    __shared__ v4u xp[4][128];
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int fl = j / strips, strip = j - fl * strips, f = fl * 8 + x;
    if (f >= nf) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int vcol = strip * 252 - 2 + tid;
    const bool prov = vcol >= 0 && vcol < H;
    if (!prov) vcol = 0;
    const bool own = prov && tid >= 2 && tid < 254;
    const size_t fbase = (size_t)f * N * H;
    v4u lo[kD + 1], hi[kD + 1];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
        const v4u* p = in + 2 * (fbase + (size_t)d * H + vcol);
        lo[d] = kNtL ? __builtin_nontemporal_load(p) : p[0];
        hi[d] = kNtL ? __builtin_nontemporal_load(p + 1) : p[1];
    }
    for (int r0 = 0; r0 < N; r0 += kD + 1) {
#pragma unroll
        for (int u = 0; u < kD + 1; ++u) {
            const int r = r0 + u;
            if (r >= N) break;
            const int rn = r + kD < N ? r + kD : N - 1;
            constexpr int dummy = 0; (void)dummy;
            const v4u* p = in + 2 * (fbase + (size_t)rn * H + vcol);
            const int sn = (u + kD) % (kD + 1);
            lo[sn] = kNtL ? __builtin_nontemporal_load(p) : p[0];
            hi[sn] = kNtL ? __builtin_nontemporal_load(p + 1) : p[1];
            v4u a = lo[u], c = hi[u];
            c.w &= 0xffff0000u;
            v4u* dst = out + 2 * (fbase + (size_t)r * H + (strip * 252 - 2 + 64 * wv));
            if (kXpose) {
                xp[wv][2 * lane] = a;
                xp[wv][2 * lane + 1] = c;
                const v4u pa = xp[wv][lane], pb = xp[wv][64 + lane];
                const unsigned long long owners = __ballot(own);
                if ((owners >> (lane >> 1)) & 1ull) __builtin_nontemporal_store(pa, dst + lane);
                if ((owners >> (32 + (lane >> 1))) & 1ull) __builtin_nontemporal_store(pb, dst + 64 + lane);
            } else if (own) {
                __builtin_nontemporal_store(a, dst + 2 * lane);
                __builtin_nontemporal_store(c, dst + 2 * lane + 1);
            }
        }
    }
}
// the same loop with whole-line LOADS as well: lane l of a wave fetches 16-B unit l (then 64 + l) of the wave's 2 KiB of
// a row — what the walk would see if its windows arrived as they lie in memory instead of as two half-line planes
template <int kD, int kDelay = 0>
__global__ __launch_bounds__(256) void k_walk_lines(const v4u* __restrict__ in, v4u* __restrict__ out, int nf, int strips, int N, int H)
{
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int fl = j / strips, strip = j - fl * strips, f = fl * 8 + x;
    if (f >= nf) return;
    if (kDelay) { /* workgroups start out of step with each other: up to kDelay x 64 sleeps of ~27 ns (does the lock step of a launch's workgroups — all reading, then all writing — cost bandwidth?) */
        const unsigned h = (blockIdx.x * 2654435761u) >> 26; /* 0 .. 63 */
        for (unsigned k = 0; k < h * kDelay; ++k) __builtin_amdgcn_s_sleep(1);
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int c0 = strip * 252 - 2 + 64 * wv; /* first column of the wave */
    if (c0 < 0) c0 = 0;
    if (c0 + 64 > H) c0 = H - 64;
    const size_t fbase = (size_t)f * N * H;
    v4u lo[kD + 1], hi[kD + 1];
#pragma unroll
    for (int d = 0; d < kD; ++d) {
        const v4u* p = in + 2 * (fbase + (size_t)d * H + c0);
        lo[d] = p[lane];
        hi[d] = p[64 + lane];
    }
    for (int r0 = 0; r0 < N; r0 += kD + 1) {
#pragma unroll
        for (int u = 0; u < kD + 1; ++u) {
            const int r = r0 + u;
            if (r >= N) break;
            const int rn = r + kD < N ? r + kD : N - 1;
            const v4u* p = in + 2 * (fbase + (size_t)rn * H + c0);
            const int sn = (u + kD) % (kD + 1);
            lo[sn] = p[lane];
            hi[sn] = p[64 + lane];
            v4u a = lo[u], c = hi[u];
            c.w &= 0xffff0000u;
            v4u* dst = out + 2 * (fbase + (size_t)r * H + c0);
            __builtin_nontemporal_store(a, dst + lane);
            __builtin_nontemporal_store(c, dst + 64 + lane);
        }
    }
}
// the walk's loop in BURSTS of kB rows: all loads of the next kB rows requested at once, then the kB rows before them stored
// at once (does DRAM like a strip's consecutive rows — 66 KB apart: neighbouring 256-B pieces of the same pages — better
// when they arrive together than one row per step?)
template <int kB>
__global__ __launch_bounds__(256) void k_walk_burst(const v4u* __restrict__ in, v4u* __restrict__ out, int nf, int strips, int N, int H)
{
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int fl = j / strips, strip = j - fl * strips, f = fl * 8 + x;
    if (f >= nf) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int c0 = strip * 252 - 2 + 64 * wv; /* first column of the wave */
    if (c0 < 0) c0 = 0;
    if (c0 + 64 > H) c0 = H - 64;
    const size_t fbase = (size_t)f * N * H;
    v4u lo[2][kB], hi[2][kB];
#pragma unroll
    for (int d = 0; d < kB; ++d) {
        const v4u* p = in + 2 * (fbase + (size_t)d * H + c0);
        lo[0][d] = p[lane];
        hi[0][d] = p[64 + lane];
    }
    for (int r0 = 0; r0 < N; r0 += 2 * kB) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int rb = r0 + half * kB;
            if (rb >= N) break;
#pragma unroll
            for (int d = 0; d < kB; ++d) { /* the next block's loads */
                const int rn = rb + kB + d < N ? rb + kB + d : N - 1;
                const v4u* p = in + 2 * (fbase + (size_t)rn * H + c0);
                lo[half ^ 1][d] = p[lane];
                hi[half ^ 1][d] = p[64 + lane];
            }
#pragma unroll
            for (int d = 0; d < kB; ++d) { /* this block's stores */
                const int r = rb + d;
                if (r >= N) break;
                v4u a = lo[half][d], c = hi[half][d];
                c.w &= 0xffff0000u;
                v4u* dst = out + 2 * (fbase + (size_t)r * H + c0);
                __builtin_nontemporal_store(a, dst + lane);
                __builtin_nontemporal_store(c, dst + 64 + lane);
            }
        }
    }
}
int main(int argc, char** argv)
{
    const bool only_walk = argc > 1;
    const size_t bytes = (size_t)2 << 30; // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
    v4u *a, *b; unsigned* o;
    CK(hipMalloc(&a, bytes + (1 << 20))); CK(hipMalloc(&b, bytes + (1 << 20))); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz, mem clock %d MHz, bus %d bits\n", prop.name, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth);
    auto time = [&](const char* name, double moved, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %.2f TB/s\n", name, moved * 5.0 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    if (!only_walk) {
    time("read only (nt, grid-stride 4096)", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, o, n16); });
    time("write only (nt, grid-stride 4096)", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, b, n16); });
    time("hipMemcpyAsync D2D", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    const unsigned gs = (unsigned)(n16 / 256);
    time("simple, default loads + stores", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_simple<false, false>), dim3(gs), dim3(256), 0, 0, a, b, n16); });
    time("simple, nt loads + nt stores", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_simple<true, true>), dim3(gs), dim3(256), 0, 0, a, b, n16); });
    time("simple, nt loads + default stores", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_simple<true, false>), dim3(gs), dim3(256), 0, 0, a, b, n16); });
    time("simple, default loads + nt stores", 2.0 * bytes, [&] { hipLaunchKernelGGL((k_simple<false, true>), dim3(gs), dim3(256), 0, 0, a, b, n16); });
    for (size_t kb : {16, 64, 256, 1024}) {
        const size_t chunk16 = kb * 1024 / 16;
        const unsigned g = (unsigned)(n16 / chunk16);
        char nm[128];
        snprintf(nm, sizeof nm, "blocked %4zu KiB per workgroup, 256 thr, 4 in flight, nt", kb);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_blocked<256, 4, true, true>), dim3(g), dim3(256), 0, 0, a, b, chunk16); });
        snprintf(nm, sizeof nm, "blocked %4zu KiB per workgroup, 256 thr, 1 in flight, nt", kb);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_blocked<256, 1, true, true>), dim3(g), dim3(256), 0, 0, a, b, chunk16); });
        snprintf(nm, sizeof nm, "blocked %4zu KiB per workgroup, 256 thr, 4 in flight, default", kb);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_blocked<256, 4, false, false>), dim3(g), dim3(256), 0, 0, a, b, chunk16); });
        snprintf(nm, sizeof nm, "blocked %4zu KiB per workgroup, 1024 thr, 2 in flight, nt", kb);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_blocked<1024, 2, true, true>), dim3(g), dim3(1024), 0, 0, a, b, chunk16); });
    }
    for (int grid : {1024, 2048, 4096, 8192}) {
        char nm[128];
        snprintf(nm, sizeof nm, "grid-stride %d x 256 thr, 4 in flight, nt", grid);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_gridstride<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
        snprintf(nm, sizeof nm, "grid-stride %d x 256 thr, 2 in flight, nt", grid);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_gridstride<2, true>), dim3(grid), dim3(256), 0, 0, a, b, n16); });
    }
    }
    {   // the walk's geometry: HDL_64E, 9 strips, 64 rows; as many frames as fit the buffer
        const int N = 64, H = 2083, strips = 9;
        const int nf = (int)(bytes / ((size_t)N * H * 32));
        const unsigned g = 8u * ((nf + 7) / 8) * strips;
        const double moved = (double)nf * N * ((double)(H + 4 * strips) + H) * 32.0; /* halo columns read, not written */
        time("walk shape, 1 row in flight, two 16-B halves per lane", moved, [&] { hipLaunchKernelGGL((k_walk<1, false, false>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 2 rows in flight, two 16-B halves per lane", moved, [&] { hipLaunchKernelGGL((k_walk<2, false, false>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 2 rows in flight, whole-line stores (LDS transpose)", moved, [&] { hipLaunchKernelGGL((k_walk<2, true, false>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 2 rows in flight, whole-line stores, nt loads", moved, [&] { hipLaunchKernelGGL((k_walk<2, true, true>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 4 rows in flight, whole-line stores", moved, [&] { hipLaunchKernelGGL((k_walk<4, true, false>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 2 rows in flight, whole-line loads AND stores", moved, [&] { hipLaunchKernelGGL((k_walk_lines<2>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, 4 rows in flight, whole-line loads AND stores", moved, [&] { hipLaunchKernelGGL((k_walk_lines<4>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, whole lines, 4 rows in flight, workgroups start up to 10 us apart", moved, [&] { hipLaunchKernelGGL((k_walk_lines<4, 6>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, whole lines, 4 rows in flight, workgroups start up to 50 us apart", moved, [&] { hipLaunchKernelGGL((k_walk_lines<4, 30>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, whole lines, bursts of 2 rows (loads of 2 rows at once, then stores of 2)", moved, [&] { hipLaunchKernelGGL((k_walk_burst<2>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, whole lines, bursts of 4 rows", moved, [&] { hipLaunchKernelGGL((k_walk_burst<4>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        time("walk shape, whole lines, bursts of 8 rows", moved, [&] { hipLaunchKernelGGL((k_walk_burst<8>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
        for (int per_cu : {1, 2, 3, 4, 6, 8}) { /* dynamic LDS limits how many of these workgroups share a CU */
            char nm[96];
            snprintf(nm, sizeof nm, "walk shape, whole-line loads and stores, 4 rows in flight, %d workgroups per CU", per_cu);
            const size_t lds = (size_t)(160 * 1024) / per_cu - 1024;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk_lines<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            time(nm, moved, [&] { hipLaunchKernelGGL((k_walk_lines<4>), dim3(g), dim3(256), lds, 0, a, b, nf, strips, N, H); });
        }
        time("walk shape, 4 rows in flight, whole-line stores, nt loads", moved, [&] { hipLaunchKernelGGL((k_walk<4, true, true>), dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H); });
    }
    return 0;
}
