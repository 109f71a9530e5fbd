/*
 * bev_instr.h — developer instrumentation: phase clocks (make clk) and workgroup time lines (make tl / cstl); all of it compiles to nothing in the product build
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_INSTR_H
#define BEV_INSTR_H

/* developer aid (make clk): phase durations of one workgroup per kernel, printed in 10 ns ticks */
#ifdef BEV_CS_CLOCK
#define PH_DECL long long ph_clk[12]; int ph_n = 0
#define PH() ph_clk[ph_n++] = wall_clock64()
#define PH_PRINT(name, cond)                                                                      \
    do {                                                                                          \
        if (cond) {                                                                               \
            long long d_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                           \
            for (int i_ = 1; i_ < ph_n && i_ <= 8; ++i_) d_[i_ - 1] = ph_clk[i_] - ph_clk[i_ - 1]; \
            printf("%s: %lld %lld %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, d_[0], d_[1], d_[2], d_[3], d_[4], d_[5], d_[6], d_[7]); \
        }                                                                                         \
    } while (0)
#define PHA_DECL long long pha_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pha_t = wall_clock64()
#define PHA(i) do { const long long n_ = wall_clock64(); pha_[i] += n_ - pha_t; pha_t = n_; } while (0)
#define PHA_PRINT(name, cond) do { if (cond) printf("%s: %lld %lld %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, pha_[0], pha_[1], pha_[2], pha_[3], pha_[4], pha_[5], pha_[6], pha_[7]); } while (0)
#else
#define PH_DECL
#define PH()
#define PH_PRINT(name, cond)
#define PHA_DECL
#define PHA(i)
#define PHA_PRINT(name, cond)
#endif

namespace bevk {
/* developer aid (make tl): start, end and place of EVERY workgroup of the pipeline's kernels since the last reset — what
 * shares the chip with what, and when (scripts/pipeline_timeline.py) */
#ifdef BEV_TL_ALL
constexpr unsigned kTlAllCap = 1u << 19;
__device__ long long g_tl_all[kTlAllCap][4];
__device__ unsigned g_tl_all_n;
#define TL_BEGIN const long long tl_all_t0 = wall_clock64()
#define TL_END(kid)                                                                                               \
    do {                                                                                                          \
        if (threadIdx.x == 0) {                                                                                   \
            const unsigned i_ = atomicAdd(&g_tl_all_n, 1u);                                                       \
            if (i_ < kTlAllCap) {                                                                                 \
                g_tl_all[i_][0] = tl_all_t0;                                                                      \
                g_tl_all[i_][1] = wall_clock64();                                                                 \
                g_tl_all[i_][2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                \
                                  ((long long)((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 32) | \
                                  ((long long)(kid) << 40);                                                       \
                g_tl_all[i_][3] = (long long)blockIdx.x;                                                          \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#else
#define TL_BEGIN
#define TL_END(kid)
#endif

} /* namespace bevk */

#endif /* BEV_INSTR_H */
