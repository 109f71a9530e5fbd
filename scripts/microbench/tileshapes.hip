// Row-block TILE shapes against the column walk's shape (scripts/microbench/copyshapes.hip measured: a no-loop copy 6.6 TB/s,
// the walk's loop over 64 rows 5.0).  A tile workgroup is short-lived: it requests ALL its input at once, waits once,
// writes ALL its output, exits.  Geometry: HDL_64E frames (64 rows x 2083 points of 32 B), strips of 252 columns + 2 halo
// columns each side, tiles of R rows + 2 halo rows above + 1 below (what phase A's stencil needs); halo rows / columns
// are read, not written.  Bytes counted = unique bytes read (every point once) + bytes written: re-reads of halo rows are
// NOT counted, so the figure is directly comparable with a copy's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// tile id -> (frame, strip, row block).  order 0: frame-major, then row block, then strip (consecutive workgroups cover
// R whole rows = contiguous memory); order 1: the same, but the 8 workgroups b, b+1, ..., b+7 (one per XCD under round-robin
// dispatch) belong to 8 different frames, so that ONE XCD sees all tiles of a frame (halo rows from its L2)
__device__ __forceinline__ bool tile_of(int b, int nf, int strips, int nrb, int order, int& f, int& strip, int& rb)
{
    const int per = strips * nrb;
    int j = b;
    if (order == 1) { const int x = b & 7; j = b >> 3; const int fl = j / per; f = fl * 8 + x; j -= fl * per; }
    else { f = j / per; j -= f * per; }
    rb = j / strips;
    strip = j - rb * strips;
    return f < nf;
}
template <int R>
__global__ __launch_bounds__(256) void k_tile_lds(const v4u* __restrict__ in, v4u* __restrict__ out, int nf, int strips, int N, int H, int order)
{
    constexpr int kRowV = 9 * 64; // v4u per row slot: 9 DMA pieces of 1 KiB >= 260 points x 32 B
    extern __shared__ v4u tile[]; // [R + 3][kRowV]
    int f, strip, rb;
    const int nrb = (N + R - 1) / R;
    if (!tile_of(blockIdx.x, nf, strips, nrb, order, f, strip, rb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t fbase = (size_t)f * N * H;
    const long long frame_v = 2ll * N * H; // v4u per frame
    const int c0 = strip * 252 - 2;
#pragma unroll
    for (int i = 0; i < R + 3; ++i) {
        int row = rb * R - 2 + i;
        row = row < 0 ? 0 : (row >= N ? N - 1 : row);
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            if (((i * 9 + c) & 3) != wv) continue; // wave-uniform
            long long v = 2ll * ((long long)row * H + c0) + c * 64 + lane;
            v = v < 0 ? 0 : (v >= frame_v ? frame_v - 1 : v);
            glds16(in + 2 * fbase + v, lds_addr(tile + i * kRowV + c * 64));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = rb * R + k;
        if (row >= N) break;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (((k * 8 + c) & 3) != wv) continue;
            const int pv = c * 64 + lane;          // v4u index inside the 252-column row piece
            const int col = c0 + 2 + (pv >> 1);
            v4u x = tile[(k + 2) * kRowV + 4 + pv];
            x.w ^= 1u;
            if (pv < 504 && col < H) __builtin_nontemporal_store(x, out + 2 * (fbase + (size_t)row * H + c0 + 2) + pv);
        }
    }
}
// the same tile through registers: thread = virtual column, all (R + 3) x 2 loads requested up front, stores as whole lines
// through an LDS transpose (as the walk does)
template <int R>
__global__ __launch_bounds__(256) void k_tile_reg(const v4u* __restrict__ in, v4u* __restrict__ out, int nf, int strips, int N, int H, int order)
{
    __shared__ v4u xp[4][128];
    int f, strip, rb;
    const int nrb = (N + R - 1) / R;
    if (!tile_of(blockIdx.x, nf, strips, nrb, order, f, strip, rb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t fbase = (size_t)f * N * H;
    int vcol = strip * 252 - 2 + tid;
    const bool prov = vcol >= 0 && vcol < H;
    if (!prov) vcol = 0;
    const bool own = prov && tid >= 2 && tid < 254;
    v4u lo[R + 3], hi[R + 3];
#pragma unroll
    for (int i = 0; i < R + 3; ++i) {
        int row = rb * R - 2 + i;
        row = row < 0 ? 0 : (row >= N ? N - 1 : row);
        const v4u* p = in + 2 * (fbase + (size_t)row * H + vcol);
        lo[i] = p[0];
        hi[i] = p[1];
    }
    const unsigned long long owners = __ballot(own);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int row = rb * R + k;
        if (row >= N) break;
        v4u a = lo[k + 2], c = hi[k + 2];
        c.w ^= lo[k].x ^ hi[k + 3].y; // (keeps the halo rows' loads alive)
        xp[wv][2 * lane] = a;
        xp[wv][2 * lane + 1] = c;
        const v4u pa = xp[wv][lane], pb = xp[wv][64 + lane];
        v4u* dst = out + 2 * (fbase + (size_t)row * H + (strip * 252 - 2 + 64 * wv));
        if ((owners >> (lane >> 1)) & 1ull) __builtin_nontemporal_store(pa, dst + lane);
        if ((owners >> (32 + (lane >> 1))) & 1ull) __builtin_nontemporal_store(pb, dst + 64 + lane);
    }
}
__global__ __launch_bounds__(256) void k_simple(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
__global__ __launch_bounds__(256) void k_simple_read(const v4u* __restrict__ a, unsigned* out, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const v4u v = __builtin_nontemporal_load(a + i);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_simple_write(v4u* __restrict__ b, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4u v = {1u, 2u, 3u, (unsigned)i};
    if (i < n16) __builtin_nontemporal_store(v, b + i);
}
int main()
{
    const size_t bytes = (size_t)2 << 30;
    v4u *a, *b; unsigned* o;
    CK(hipMalloc(&a, bytes + (1 << 20))); CK(hipMalloc(&b, bytes + (1 << 20))); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double moved, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %.2f TB/s\n", name, moved * 5.0 / (ms * 1e-3) / 1e12);
        fflush(stdout);
    };
    const unsigned gs = (unsigned)(n16 / 256);
    time("simple copy (16 B per thread, no loop, nt)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_simple, dim3(gs), dim3(256), 0, 0, a, b, n16); });
    time("simple read only", (double)bytes, [&] { hipLaunchKernelGGL(k_simple_read, dim3(gs), dim3(256), 0, 0, a, o, n16); });
    time("simple write only", (double)bytes, [&] { hipLaunchKernelGGL(k_simple_write, dim3(gs), dim3(256), 0, 0, b, n16); });
    const int N = 64, H = 2083, strips = 9;
    const int nf = (int)(bytes / ((size_t)N * H * 32)) / 8 * 8;
    const double moved = (double)nf * N * (2.0 * H) * 32.0;
#define TILE_LDS(R)                                                                                                          \
    for (int order = 0; order < 2; ++order) {                                                                                \
        const int nrb = (N + R - 1) / R;                                                                                     \
        const unsigned g = (unsigned)(nf * strips * nrb);                                                                    \
        const size_t lds = (size_t)(R + 3) * 9 * 64 * 16;                                                                    \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_lds<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        char nm[128];                                                                                                        \
        snprintf(nm, sizeof nm, "tile via LDS-DMA, %2d rows (+3 halo), %3zu KB LDS, %s", R, lds / 1024, order ? "a frame per XCD" : "frame-major");     \
        time(nm, moved, [&] { hipLaunchKernelGGL(k_tile_lds<R>, dim3(g), dim3(256), lds, 0, a, b, nf, strips, N, H, order); }); \
    }
#define TILE_REG(R)                                                                                                          \
    for (int order = 0; order < 2; ++order) {                                                                                \
        const int nrb = (N + R - 1) / R;                                                                                     \
        const unsigned g = (unsigned)(nf * strips * nrb);                                                                    \
        char nm[128];                                                                                                        \
        snprintf(nm, sizeof nm, "tile via registers, %2d rows (+3 halo), %s", R, order ? "a frame per XCD" : "frame-major"); \
        time(nm, moved, [&] { hipLaunchKernelGGL(k_tile_reg<R>, dim3(g), dim3(256), 0, 0, a, b, nf, strips, N, H, order); }); \
    }
    TILE_LDS(2) TILE_LDS(4) TILE_LDS(8) TILE_LDS(13)
    TILE_REG(2) TILE_REG(4) TILE_REG(8) TILE_REG(16)
    return 0;
}
