// HBM copy bandwidth sweep: unroll depth, non-temporal loads/stores, grid size.  (What does a streaming kernel need
// to reach the ~6.3 TB/s the MI355X guide quotes for a float4 copy?)
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTL) {
        const float4 *p = src + i + u * stride;
        v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
        v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
      } else {
        v[u] = src[i + u * stride];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) {
        float4 *p = dst + i + u * stride;
        __builtin_nontemporal_store(v[u].x, &p->x); __builtin_nontemporal_store(v[u].y, &p->y);
        __builtin_nontemporal_store(v[u].z, &p->z); __builtin_nontemporal_store(v[u].w, &p->w);
      } else {
        dst[i + u * stride] = v[u];
      }
    }
  }
}

// contiguous chunk per workgroup instead of grid-stride
template <int U>
__global__ __launch_bounds__(256) void copy_chunk_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  const size_t per = n / gridDim.x;
  const float4 *s = src + (size_t)blockIdx.x * per;
  float4 *d = dst + (size_t)blockIdx.x * per;
  for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
  }
}

template <typename F>
static void run(const char *name, F launch, size_t bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.2f TB/s\n", name, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;
  float4 *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  for (int bpc : {4, 8, 16, 32, 64}) {
    const int blocks = 256 * bpc;
    char nm[96];
    snprintf(nm, 96, "grid-stride U=4, %d blocks/CU", bpc);
    run(nm, [&] { hipLaunchKernelGGL((copy_kernel<4, false, false>), dim3(blocks), dim3(256), 0, 0, src, dst, n); }, bytes);
    snprintf(nm, 96, "grid-stride U=8, %d blocks/CU", bpc);
    run(nm, [&] { hipLaunchKernelGGL((copy_kernel<8, false, false>), dim3(blocks), dim3(256), 0, 0, src, dst, n); }, bytes);
    snprintf(nm, 96, "grid-stride U=4 nt-store, %d blocks/CU", bpc);
    run(nm, [&] { hipLaunchKernelGGL((copy_kernel<4, false, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n); }, bytes);
    snprintf(nm, 96, "grid-stride U=4 nt-load+store, %d blocks/CU", bpc);
    run(nm, [&] { hipLaunchKernelGGL((copy_kernel<4, true, true>), dim3(blocks), dim3(256), 0, 0, src, dst, n); }, bytes);
    snprintf(nm, 96, "chunked U=4, %d blocks/CU", bpc);
    run(nm, [&] { hipLaunchKernelGGL((copy_chunk_kernel<4>), dim3(blocks), dim3(256), 0, 0, src, dst, n); }, bytes);
  }
  // a smaller working set (two 256 MiB buffers) and simple one-float4-per-thread copies, as most published probes are
  for (size_t mb : {256, 1024}) {
    const size_t nb = mb << 20, nn = nb / 16;
    char nm[96];
    snprintf(nm, 96, "one float4 per thread, %zu MiB buffers", mb);
    run(nm, [&] { hipLaunchKernelGGL((copy_kernel<1, false, false>), dim3((unsigned)(nn / 256)), dim3(256), 0, 0, src, dst, nn); }, nb);
    snprintf(nm, 96, "hipMemcpyDtoD, %zu MiB", mb);
    run(nm, [&] { (void)hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToDevice, 0); }, nb);
  }
  return 0;
}
