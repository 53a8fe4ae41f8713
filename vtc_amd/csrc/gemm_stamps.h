// gemm_stamps.h -- cycle stamps of gemm_phased_kernel's tile-level phases: a DIAGNOSTIC build only (-DVTC_GEMM_STAMPS,
// tools/build_variant.sh; __graft_entry__.build() refuses it).  Results stay correct, the kernel runs ~10 % slower
// (s_memtime + lgkmcnt(0) per stamp) and the launcher synchronises to print the sums.  Without the macro everything here
// expands to nothing, so the product kernel carries no stamp code.
#pragma once
#ifdef VTC_GEMM_STAMPS
#define VTC_STAMP_INIT()                                                                        \
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tsp = 0;                                       \
  auto stamp = [&]() -> unsigned long long {                                                    \
    unsigned long long tsv;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tsv)::"memory");                 \
    return tsv;                                                                                 \
  };                                                                                            \
  tsp = stamp()
#define VTC_STAMP(i)                                    \
  {                                                     \
    const unsigned long long t_ = stamp();              \
    ph[i] += t_ - tsp;                                  \
    tsp = t_;                                           \
  }
// sums leave through p.dbg only: a buffer of their own that no other code reads
#define VTC_STAMP_TILE_END(last)                                                          \
  {                                                                                       \
    const unsigned long long t_ = stamp();                                                \
    ph[2] += t_ - tsp;                                                                    \
    tsp = t_;                                                                             \
    ph[5] += 1;                                                                           \
    if ((last) && lane == 0 && p.dbg)                                                     \
      for (int i_ = 0; i_ < 6; ++i_) p.dbg[((size_t)bid * NW + wave) * 8 + i_] = ph[i_];  \
  }
#define VTC_STAMP_HOST_BEFORE(p, stream)                                                      \
  static unsigned long long *dbg_ = nullptr;                                                  \
  if (!dbg_) (void)hipMalloc(&dbg_, (size_t)256 * 8 * 8 * sizeof(unsigned long long));        \
  (void)hipMemsetAsync(dbg_, 0, (size_t)256 * 8 * 8 * sizeof(unsigned long long), stream);    \
  p.dbg = dbg_
#define VTC_STAMP_HOST_AFTER(p, stream, grid, MODE)                                                                                    \
  {                                                                                                                                    \
    static unsigned long long host_[256 * 8 * 8];                                                                                      \
    (void)hipStreamSynchronize(stream);                                                                                                \
    (void)hipMemcpy(host_, dbg_, sizeof(host_), hipMemcpyDeviceToHost);                                                                \
    double sum_[6] = {0, 0, 0, 0, 0, 0};                                                                                               \
    const int nw_ = (grid) * 8;                                                                                                        \
    for (int w_ = 0; w_ < nw_; ++w_)                                                                                                   \
      for (int i_ = 0; i_ < 6; ++i_) sum_[i_] += (double)host_[(size_t)w_ * 8 + i_];                                                   \
    if (sum_[5] > 0)                                                                                                                   \
      fprintf(stderr,                                                                                                                  \
              "[phased stamps] M=%d N=%d K=%d mode %d: per tile cycles: k-loop %.0f | re-join %.0f | epilogue %.0f | barrier %.0f "   \
              "(tiles/wave %.1f)\n",                                                                                                   \
              p.M, p.N, p.K, MODE, sum_[0] / sum_[5], sum_[1] / sum_[5], sum_[2] / sum_[5], sum_[3] / sum_[5], sum_[5] / nw_);         \
  }
#elif defined(VTC_GEMM_PHASE_STAMPS)
// phase-level stamps of the deep K loop (a second DIAGNOSTIC build): per wave, the cycles between the barrier that opens a phase's MFMA
// cluster and the end of the cluster's issue (slot 0), and from there to the opening of the next cluster (slot 1: second barrier,
// the wave's fragment reads / LDS-DMA issue / counted waits, first barrier); slot 2 counts phases.  Two s_memtime per phase (~10 %).
#define VTC_STAMP_INIT()                                                                        \
  unsigned long long pst_[6] = {0, 0, 0, 0, 0, 0}, tsp = 0;                                       \
  auto stamp = [&]() -> unsigned long long {                                                    \
    unsigned long long tsv;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tsv)::"memory");                 \
    return tsv;                                                                                 \
  };                                                                                            \
  tsp = stamp()
#define VTC_STAMP(i) ((void)0)
#define VTC_PHASE_STAMP(i)                              \
  {                                                     \
    const unsigned long long t_ = stamp();              \
    pst_[i] += t_ - tsp;                                  \
    tsp = t_;                                           \
    if ((i) == 0) pst_[2] += 1;                           \
  }
#define VTC_STAMP_TILE_END(last)                                                          \
  {                                                                                       \
    tsp = stamp();                                                                        \
    if ((last) && lane == 0 && p.dbg)                                                     \
      for (int i_ = 0; i_ < 6; ++i_) p.dbg[((size_t)bid * NW + wave) * 8 + i_] = pst_[i_];  \
  }
#define VTC_STAMP_HOST_BEFORE(p, stream)                                                      \
  static unsigned long long *dbg_ = nullptr;                                                  \
  if (!dbg_) (void)hipMalloc(&dbg_, (size_t)256 * 8 * 8 * sizeof(unsigned long long));        \
  (void)hipMemsetAsync(dbg_, 0, (size_t)256 * 8 * 8 * sizeof(unsigned long long), stream);    \
  p.dbg = dbg_
#define VTC_STAMP_HOST_AFTER(p, stream, grid, MODE)                                                                                    \
  {                                                                                                                                    \
    static unsigned long long host_[256 * 8 * 8];                                                                                      \
    (void)hipStreamSynchronize(stream);                                                                                                \
    (void)hipMemcpy(host_, dbg_, sizeof(host_), hipMemcpyDeviceToHost);                                                                \
    double sum_[2][3] = {{0, 0, 0}, {0, 0, 0}};                                                                                        \
    const int nw_ = (grid) * 8;                                                                                                        \
    for (int w_ = 0; w_ < nw_; ++w_)                                                                                                   \
      for (int i_ = 0; i_ < 3; ++i_) sum_[(w_ & 7) >> 2][i_] += (double)host_[(size_t)w_ * 8 + i_];                                    \
    if (sum_[0][2] > 0)                                                                                                                \
      fprintf(stderr,                                                                                                                  \
              "[phase stamps] M=%d N=%d K=%d mode %d: cycles per phase: waves 0-3 MFMA cluster %.0f, rest %.0f | waves 4-7 MFMA "     \
              "cluster %.0f, rest %.0f (phases per wave %.0f)\n",                                                                      \
              p.M, p.N, p.K, MODE, sum_[0][0] / sum_[0][2], sum_[0][1] / sum_[0][2], sum_[1][0] / sum_[1][2], sum_[1][1] / sum_[1][2], \
              (sum_[0][2] + sum_[1][2]) / nw_);                                                                                        \
  }
#else
#define VTC_STAMP_INIT() ((void)0)
#define VTC_STAMP(i) ((void)0)
#define VTC_STAMP_TILE_END(last) ((void)0)
#define VTC_STAMP_HOST_BEFORE(p, stream) ((void)0)
#define VTC_STAMP_HOST_AFTER(p, stream, grid, MODE) ((void)0)
#endif
#ifndef VTC_PHASE_STAMP
#define VTC_PHASE_STAMP(i) ((void)0)
#endif
