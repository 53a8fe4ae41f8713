// What does the (hi, lo) residual epilogue's ACCESS PATTERN cost, apart from the GEMM in front of it?  A streaming kernel that reads and
// rewrites two 16-bit arrays [M, 768] (hi, lo) in 256 x 256 tiles, with the epilogue's lane -> (row, 4 columns) map (8-byte accesses,
// 16 rows x 64 columns per wave and pass), against (b) the same bytes with 16-byte accesses / 8 columns per lane, (c) one workgroup
// owning whole 256-row panels (contiguous 384 KB per array), and (d) a plain contiguous copy of the same byte count.
// An optional busy phase of `spin` K-tile times between tiles stands for the K loop (no memory traffic, all waves together).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int N = 768;

__device__ __forceinline__ void spin_for(long long ticks) {      // 100 MHz counter
  if (ticks <= 0) return;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

// MODE 0: 8-byte accesses, lane = (row lane >> 4 + 4 k, cols 4 (lane & 15));  MODE 1: 16-byte, lane = (row lane >> 3 + 8 k, cols 8 (lane & 7))
template <int MODE>
__global__ __launch_bounds__(512) void tiles_kernel(unsigned short *hi, unsigned short *lo, int M, long long spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const int MT = M / 256, NT = N / 256, ntiles = MT * NT;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = (t / NT) * 256, n0 = (t % NT) * 256;
    spin_for(spin);
    for (int i = 0; i < 8; ++i) {
      const int mrow0 = m0 + wr * 128 + 16 * i, ncol0 = n0 + wc * 64;
      if (MODE == 0) {
        uint2 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const size_t e = (size_t)(mrow0 + (lane >> 4) + 4 * k) * N + ncol0 + (lane & 15) * 4;
          h[k] = *reinterpret_cast<const uint2 *>(hi + e);
          l[k] = *reinterpret_cast<const uint2 *>(lo + e);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const size_t e = (size_t)(mrow0 + (lane >> 4) + 4 * k) * N + ncol0 + (lane & 15) * 4;
          *reinterpret_cast<uint2 *>(hi + e) = make_uint2(h[k].x + l[k].y, h[k].y + 1);
          *reinterpret_cast<uint2 *>(lo + e) = make_uint2(l[k].x + 1, l[k].y + h[k].x);
        }
      } else {
        uint4 h[2], l[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const size_t e = (size_t)(mrow0 + (lane >> 3) + 8 * k) * N + ncol0 + (lane & 7) * 8;
          h[k] = *reinterpret_cast<const uint4 *>(hi + e);
          l[k] = *reinterpret_cast<const uint4 *>(lo + e);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const size_t e = (size_t)(mrow0 + (lane >> 3) + 8 * k) * N + ncol0 + (lane & 7) * 8;
          *reinterpret_cast<uint4 *>(hi + e) = make_uint4(h[k].x + l[k].y, h[k].y + 1, h[k].z, h[k].w + 1);
          *reinterpret_cast<uint4 *>(lo + e) = make_uint4(l[k].x + 1, l[k].y + h[k].x, l[k].z, l[k].w);
        }
      }
    }
  }
}

// one workgroup per 256-row panel: 384 KB contiguous per array, 16-byte accesses, wave-contiguous 1 KB pieces
__global__ __launch_bounds__(512) void panel_kernel(unsigned short *hi, unsigned short *lo, int M, long long spin) {
  const int MT = M / 256;
  for (int t = blockIdx.x; t < MT; t += gridDim.x) {
    spin_for(3 * spin);
    uint4 *h = reinterpret_cast<uint4 *>(hi + (size_t)t * 256 * N), *l = reinterpret_cast<uint4 *>(lo + (size_t)t * 256 * N);
    const int n16 = 256 * N * 2 / 16;      // 24 576 pieces of 16 bytes
    for (int i = threadIdx.x; i < n16; i += 512 * 2) {
      const uint4 a = h[i], b = l[i], c = h[i + 512], d = l[i + 512];
      h[i] = make_uint4(a.x + b.y, a.y + 1, a.z, a.w);
      l[i] = make_uint4(b.x + 1, b.y, b.z, b.w + a.x);
      h[i + 512] = make_uint4(c.x + d.y, c.y + 1, c.z, c.w);
      l[i + 512] = make_uint4(d.x + 1, d.y, d.z, d.w + c.x);
    }
  }
}

template <typename F>
static void run(const char *name, F launch, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %8.1f us  %.2f TB/s\n", name, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
  const int M = 402432;
  const size_t elems = (size_t)M * N;
  unsigned short *hi, *lo;
  hipMalloc(&hi, elems * 2); hipMalloc(&lo, elems * 2);
  hipMemset(hi, 1, elems * 2); hipMemset(lo, 2, elems * 2);
  const double bytes = 4.0 * elems * 2;      // read + write of both arrays: 2.47 GB (the GEMM adds 0.62 GB of operand reads)
  for (long long spin : {0LL, 1700LL}) {     // 1700 ticks = 17 us: the 12 K-tiles of a K = 768 tile at the measured rate
    char nm[128];
    snprintf(nm, 128, "tiles 256x256, 8-byte accesses (the epilogue's map), spin %lld", spin);
    run(nm, [&] { hipLaunchKernelGGL((tiles_kernel<0>), dim3(256), dim3(512), 0, 0, hi, lo, M, spin); }, bytes);
    snprintf(nm, 128, "tiles 256x256, 16-byte accesses, spin %lld", spin);
    run(nm, [&] { hipLaunchKernelGGL((tiles_kernel<1>), dim3(256), dim3(512), 0, 0, hi, lo, M, spin); }, bytes);
    snprintf(nm, 128, "256-row panels (contiguous), 16-byte accesses, spin 3 x %lld", spin);
    run(nm, [&] { hipLaunchKernelGGL(panel_kernel, dim3(256), dim3(512), 0, 0, hi, lo, M, spin); }, bytes);
  }
  return 0;
}
