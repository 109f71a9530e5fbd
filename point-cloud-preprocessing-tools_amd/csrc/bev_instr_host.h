/*
 * bev_instr_host.h — developer builds only: host accessors of the records the instrumented kernels leave in device memory
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_INSTR_HOST_H
#define BEV_INSTR_HOST_H

namespace bevk {
#ifdef BEV_TL_ALL
} // namespace bevk
extern "C" int bev_tl_all(long long *out, int cap, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(bevk::g_tl_all_n), sizeof n) != hipSuccess) return -1;
    if (n > bevk::kTlAllCap) n = bevk::kTlAllCap;
    if ((int)n > cap) n = (unsigned)cap;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(bevk::g_tl_all), (size_t)n * 4 * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(bevk::g_tl_all_n), &z, sizeof z) != hipSuccess) return -1;
    }
    return (int)n;
}
namespace bevk {
#endif
#ifdef BEV_CS_CLOCK /* developer build: start, end, HW_ID, XCC_ID of every workgroup of the last in-place walk launch */
constexpr int kWalkTlCap = 8192;
__device__ long long g_walk_tl[kWalkTlCap][4];
} // namespace bevk
extern "C" int bev_clk_walk_timeline(long long *out, int cap)
{
    const int n = cap < bevk::kWalkTlCap ? cap : bevk::kWalkTlCap;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bevk::g_walk_tl), (size_t)n * 4 * sizeof(long long)) == hipSuccess ? n : -1;
}
namespace bevk {
#endif

} /* namespace bevk */

#endif /* BEV_INSTR_HOST_H */
