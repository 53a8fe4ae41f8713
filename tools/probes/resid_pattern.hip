// resid_pattern.hip -- PROBE (tools only, never part of the product): what does the residual epilogue's MEMORY PATTERN reach by
// itself?  The K = N = 768 residual GEMM of the towers moves 3.09 GB in 0.73 ms = 4.2 TB/s against a chunked copy's 5.8
// (DESIGN 4.4).  Here the same persistent 256-workgroup walk over 256 x 256 tiles of a [402 432, 768] (hi, lo) bf16 pair, updated
// in place with the epilogue's access shapes, WITHOUT the K loop -- or with a timed gap per tile standing in for it:
//   P0: the epilogue's shape: 8 B per lane, 16 lanes per 128-byte row segment, 4 rows per instruction (wave = 128 rows x 64 columns)
//   P1: 16 B per lane, 8 lanes per 128-byte segment, 8 rows per instruction (same wave tile)
//   P2: 16 B per lane, 32 lanes per 512-byte segment (the tile's whole row), 2 rows per instruction (wave = 32 rows x 256 columns)
//   P3: contiguous chunk per workgroup, 16 B per lane (streaming in-place update: the bound for this traffic)
// ST: 0 plain stores, 1 nt, 2 sc1.  depth = row groups of loads in flight ahead of the stores.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/resid_pattern tools/probes/resid_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int ST, typename V> __device__ __forceinline__ void st_(V *p, V v) {
  if constexpr (ST == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (ST == 2) {
    if constexpr (sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else *p = v;
}
__device__ __forceinline__ u2 upd(u2 a, u2 b) { return (u2){a.x + 0x00010001u * (b.x & 1u), a.y ^ (b.y & 0x00010001u)}; }
__device__ __forceinline__ u4 upd(u4 a, u4 b) { return (u4){a.x + 0x00010001u * (b.x & 1u), a.y ^ (b.y & 0x00010001u), a.z + (b.z & 1u), a.w ^ (b.w & 1u)}; }

struct P {
  unsigned short *hi, *lo;
  int M, W, MT, NT;
  int gap_ticks;      // s_memrealtime ticks (100 MHz) spent per tile before its update: the K loop's stand-in
  int prefetch;       // 1: at the START of the gap every wave touches one dword per 128-byte line of the (hi, lo) rows it is about to update
                      // (4 loads per wave: lane = row), so that the lines are on their way to L2 / Infinity Cache while the "K loop" runs
};

template <int PAT, int ST, int XD>
__global__ __launch_bounds__(512, 2) void resid_kernel(P p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if constexpr (PAT == 3) {
    const size_t n16 = (size_t)p.M * p.W * 2 / 16, per = n16 / gridDim.x;
    u4 *h = reinterpret_cast<u4 *>(p.hi), *l = reinterpret_cast<u4 *>(p.lo);
    const size_t b = (size_t)blockIdx.x * per, e = b + per;
    for (size_t i = b + tid; i + 3 * 512 < e; i += 4 * 512) {
      u4 a[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = h[i + u * 512]; c[u] = l[i + u * 512]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { st_<ST>(h + i + u * 512, upd(a[u], c[u])); st_<ST>(l + i + u * 512, upd(c[u], a[u])); }
    }
    return;
  } else {
    // persistent XCD-aware walk of gemm_phased_kernel: super-rows of 4 row blocks, column by column inside
    const int ntiles = p.MT * p.NT, nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
    const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
    const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
    const int SUPER = 4;
    for (int li = slot; li < nt_x; li += nb_x) {
      const int logical = start_x + li;
      const int per_super = SUPER * p.NT;
      const int sr = logical / per_super, rem = logical - sr * per_super;
      const int gsz = min(SUPER, p.MT - sr * SUPER);
      const int nt = rem / gsz;
      const int m0 = (sr * SUPER + (rem - nt * gsz)) * 256, n0 = nt * 256;
      unsigned pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;
      if (p.prefetch) {
        const int wr = wave >> 2, wc = wave & 3;
        const size_t o = (size_t)(m0 + wr * 128 + lane) * p.W + n0 + wc * 64;
        const unsigned short *a0 = p.hi + o, *a1 = p.hi + o + (size_t)64 * p.W, *a2 = p.lo + o, *a3 = p.lo + o + (size_t)64 * p.W;
        asm volatile("global_load_dword %0, %1, off" : "=v"(pf0) : "v"(a0) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pf1) : "v"(a1) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pf2) : "v"(a2) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pf3) : "v"(a3) : "memory");
      }
      if (p.gap_ticks > 0) {
        const long long t_end = (long long)__builtin_amdgcn_s_memrealtime() + p.gap_ticks;
        while ((long long)__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(4);
      }
      if (p.prefetch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));
      }
      if constexpr (PAT == 0) {
        // wave = rows wr*128.., columns wc*64..; pass i = 16 rows; k = 4 row groups of 4 rows; 8 B per lane
        const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15;
        auto off = [&](int i, int k) { return (size_t)(m0 + wr * 128 + i * 16 + (lane >> 4) + 4 * k) * p.W + n0 + wc * 64 + l15 * 4; };
        u2 xh[XD][4], xl[XD][4];
#pragma unroll
        for (int a = 0; a < XD - 1; ++a)
#pragma unroll
          for (int k = 0; k < 4; ++k) { xh[a][k] = *reinterpret_cast<const u2 *>(p.hi + off(a, k)); xl[a][k] = *reinterpret_cast<const u2 *>(p.lo + off(a, k)); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + XD - 1 < 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              xh[(i + XD - 1) % XD][k] = *reinterpret_cast<const u2 *>(p.hi + off(i + XD - 1, k));
              xl[(i + XD - 1) % XD][k] = *reinterpret_cast<const u2 *>(p.lo + off(i + XD - 1, k));
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            st_<ST>(reinterpret_cast<u2 *>(p.hi + off(i, k)), upd(xh[i % XD][k], xl[i % XD][k]));
            st_<ST>(reinterpret_cast<u2 *>(p.lo + off(i, k)), upd(xl[i % XD][k], xh[i % XD][k]));
          }
        }
      } else if constexpr (PAT == 1) {
        // same wave tile, 16 B per lane: 8 lanes per 128-byte segment, 8 rows per instruction; pass i = 16 rows = 2 instructions
        const int wr = wave >> 2, wc = wave & 3, l7 = lane & 7;
        auto off = [&](int i, int k) { return (size_t)(m0 + wr * 128 + i * 16 + (lane >> 3) + 8 * k) * p.W + n0 + wc * 64 + l7 * 8; };
        u4 xh[XD][2], xl[XD][2];
#pragma unroll
        for (int a = 0; a < XD - 1; ++a)
#pragma unroll
          for (int k = 0; k < 2; ++k) { xh[a][k] = *reinterpret_cast<const u4 *>(p.hi + off(a, k)); xl[a][k] = *reinterpret_cast<const u4 *>(p.lo + off(a, k)); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + XD - 1 < 8) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              xh[(i + XD - 1) % XD][k] = *reinterpret_cast<const u4 *>(p.hi + off(i + XD - 1, k));
              xl[(i + XD - 1) % XD][k] = *reinterpret_cast<const u4 *>(p.lo + off(i + XD - 1, k));
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            st_<ST>(reinterpret_cast<u4 *>(p.hi + off(i, k)), upd(xh[i % XD][k], xl[i % XD][k]));
            st_<ST>(reinterpret_cast<u4 *>(p.lo + off(i, k)), upd(xl[i % XD][k], xh[i % XD][k]));
          }
        }
      } else {
        // wave = 32 rows x the tile's 256 columns: 32 lanes per 512-byte row segment, 2 rows per instruction; pass i = 4 rows
        const int l31 = lane & 31;
        auto off = [&](int i, int k) { return (size_t)(m0 + wave * 32 + i * 4 + (lane >> 5) + 2 * k) * p.W + n0 + l31 * 8; };
        u4 xh[XD][2], xl[XD][2];
#pragma unroll
        for (int a = 0; a < XD - 1; ++a)
#pragma unroll
          for (int k = 0; k < 2; ++k) { xh[a][k] = *reinterpret_cast<const u4 *>(p.hi + off(a, k)); xl[a][k] = *reinterpret_cast<const u4 *>(p.lo + off(a, k)); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + XD - 1 < 8) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              xh[(i + XD - 1) % XD][k] = *reinterpret_cast<const u4 *>(p.hi + off(i + XD - 1, k));
              xl[(i + XD - 1) % XD][k] = *reinterpret_cast<const u4 *>(p.lo + off(i + XD - 1, k));
            }
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            st_<ST>(reinterpret_cast<u4 *>(p.hi + off(i, k)), upd(xh[i % XD][k], xl[i % XD][k]));
            st_<ST>(reinterpret_cast<u4 *>(p.lo + off(i, k)), upd(xl[i % XD][k], xh[i % XD][k]));
          }
        }
      }
    }
  }
}

template <typename F>
static double run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 6;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps * 1e-3;
}

#define RUN(PAT, ST, XD, gap, label)                                                                                              \
  do {                                                                                                                            \
    p.gap_ticks = gap; p.prefetch = PFV;                                                                                          \
    const double t = run([&] { hipLaunchKernelGGL((resid_kernel<PAT, ST, XD>), dim3(cus), dim3(512), 0, 0, p); });                 \
    printf("P%d %-44s st=%d depth=%d pf=%d gap=%4.1f us : %7.1f us  %5.2f TB/s   (per tile round %.1f us)\n", PAT, label, ST, XD, PFV, gap * 0.01, t * 1e6,            \
           moved / t / 1e12, t * 1e6 / rounds);                                                                                   \
    fflush(stdout);                                                                                                               \
  } while (0)

int main() {
  int PFV = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  P p;
  p.M = 402432; p.W = 768; p.MT = p.M / 256; p.NT = p.W / 256;
  const size_t bytes = (size_t)p.M * p.W * 2;
  if (hipMalloc(&p.hi, bytes) != hipSuccess || hipMalloc(&p.lo, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(p.hi, 1, bytes); hipMemset(p.lo, 2, bytes);
  const double moved = 4.0 * bytes;     // hi and lo, read and written
  const double rounds = (double)p.MT * p.NT / cus;
  printf("in-place update of a (hi, lo) bf16 pair [%d, %d]: %.2f GB moved per launch, %d workgroups x 512 threads, %.2f tile rounds\n", p.M, p.W,
         moved / 1e9, cus, rounds);
  RUN(3, 0, 2, 0, "contiguous chunk per workgroup, 16 B/lane");
  RUN(3, 1, 2, 0, "contiguous chunk per workgroup, 16 B/lane");
  RUN(0, 0, 2, 0, "epilogue shape (8 B/lane, 4 rows x 128 B)");
  RUN(0, 0, 3, 0, "epilogue shape (8 B/lane, 4 rows x 128 B)");
  RUN(0, 0, 5, 0, "epilogue shape (8 B/lane, 4 rows x 128 B)");
  RUN(0, 1, 2, 0, "epilogue shape (8 B/lane, 4 rows x 128 B)");
  RUN(0, 2, 2, 0, "epilogue shape (8 B/lane, 4 rows x 128 B)");
  RUN(1, 0, 2, 0, "16 B/lane, 8 rows x 128 B");
  RUN(1, 0, 3, 0, "16 B/lane, 8 rows x 128 B");
  RUN(1, 1, 2, 0, "16 B/lane, 8 rows x 128 B");
  RUN(2, 0, 2, 0, "16 B/lane, 2 rows x 512 B");
  RUN(2, 0, 3, 0, "16 B/lane, 2 rows x 512 B");
  RUN(2, 0, 5, 0, "16 B/lane, 2 rows x 512 B");
  RUN(2, 1, 3, 0, "16 B/lane, 2 rows x 512 B");
  RUN(2, 2, 3, 0, "16 B/lane, 2 rows x 512 B");
  // with the K loop's stand-in: ~21 us per tile (the out-proj's 12 K-tiles), ~10 us, ~40 us
  for (int gap : {1000, 1700, 2100, 4000}) {
    PFV = 1;
    RUN(0, 0, 2, gap, "epilogue shape, lines touched at gap start");
    RUN(0, 0, 3, gap, "epilogue shape, lines touched at gap start");
    PFV = 0;
    RUN(0, 0, 2, gap, "epilogue shape (8 B/lane, 4 rows x 128 B)");
    RUN(1, 0, 2, gap, "16 B/lane, 8 rows x 128 B");
    RUN(2, 0, 3, gap, "16 B/lane, 2 rows x 512 B");
    RUN(2, 0, 5, gap, "16 B/lane, 2 rows x 512 B");
  }
  return 0;
}
