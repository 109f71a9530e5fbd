// Why does k_order_scan read at 3.6 TB/s when the same access pattern (one dword of every 32-byte point) reaches
// 6.3 TB/s in readbw.hip?  Replicas of its structure, one feature at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct FrameDesc { unsigned long long in_offset; unsigned n_pts; unsigned pad; };
// A: 2-D grid, per-frame descriptor from memory, 4 loads per thread, block exits
template <int U, bool DESC, bool SINK_ATOMIC>
__global__ __launch_bounds__(256) void k_scan(const unsigned* __restrict__ p, const FrameDesc* __restrict__ frames,
                                              unsigned pts_per_frame, unsigned* sink, unsigned* winner, int S)
{
    const int f = blockIdx.y;
    unsigned long long off; unsigned n;
    if (DESC) { const FrameDesc fd = frames[f]; off = fd.in_offset; n = fd.n_pts; }
    else { off = (unsigned long long)f * pts_per_frame; n = pts_per_frame; }
    const unsigned base = blockIdx.x * (256u * U) + threadIdx.x;
    if (blockIdx.x * (256u * U) >= n) return;
    const unsigned* fp = p + off * 8;
    unsigned v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const unsigned i = base + 256u * k; v[k] = i < n ? fp[(size_t)i * 8 + 5] : 0xffffffffu; }
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (SINK_ATOMIC) { if (v[k] != 0xffffffffu) atomicMax(&winner[(size_t)f * S + (base + 256u * k) % S], base + 256u * k + 1u); }
        else acc ^= v[k];
    }
    if (!SINK_ATOMIC && acc == 0x12345678u) *sink = acc;
}
int main()
{
    const unsigned P = 135646, NF = 1000, S = 133312;
    const size_t npts = (size_t)P * NF, bytes = npts * 32;
    unsigned *d, *sink, *winner; FrameDesc* fr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess || hipMalloc(&fr, NF * sizeof(FrameDesc)) != hipSuccess ||
        hipMalloc(&winner, (size_t)256 * S * 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes); (void)hipMemset(winner, 0, (size_t)256 * S * 4);
    std::vector<FrameDesc> h(NF);
    for (unsigned f = 0; f < NF; ++f) h[f] = FrameDesc{(unsigned long long)f * P, P, 0};
    (void)hipMemcpy(fr, h.data(), NF * sizeof(FrameDesc), hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        for (int s = 0; s < 4; ++s) launch(s); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a); for (int r = 0; r < 3; ++r) for (int s = 0; s < 4; ++s) launch(s); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-44s %.2f TB/s  (%.2f us per frame)\n", name, bytes * 3.0 / (ms * 1e-3) / 1e12, ms * 1e3 / 3.0 / NF);
    };
#define L(U, DESC, AT) [&](int s) { const int nf = s < 3 ? 256 : 232; hipLaunchKernelGGL((k_scan<U, DESC, AT>), dim3((P + 256 * U - 1) / (256 * U), nf), dim3(256), 0, 0, d, fr + s * 256, P, sink, winner, (int)S); }
    run("2-D grid, computed offsets, U=4", L(4, false, false));
    run("2-D grid, descriptor load, U=4", L(4, true, false));
    run("2-D grid, descriptor load, U=8", L(8, true, false));
    run("2-D grid, descriptor, U=4, atomicMax sink", L(4, true, true));
    run("2-D grid, descriptor, U=8, atomicMax sink", L(8, true, true));
    return 0;
}
