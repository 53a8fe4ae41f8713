// Measured peaks of this box, next to the nominal ones (SURVEY 8d "Evidence"):
//   * HBM: float4 copy of a buffer far larger than the 256 MiB Infinity Cache (read + write bytes / time);
//   * MFMA: register-resident loops of v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32 on random operands,
//     one wave per SIMD and two waves per SIMD, every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/peaks tools/probes/peaks.hip ; run: ./tools/probes/peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

template <int NACC>
__global__ void mfma_bf16_kernel(const uint4 *__restrict__ seed, float *__restrict__ sink, int iters) {
  const uint4 s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
  const bf16x8 a = __builtin_bit_cast(bf16x8, s0), b = __builtin_bit_cast(bf16x8, s1);
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 12345.678f) sink[0] = r;
}

template <int NACC>
__global__ void mfma_f32_kernel(const float *__restrict__ seed, float *__restrict__ sink, int iters) {
  const float a = seed[threadIdx.x & 63], b = seed[64 + (threadIdx.x & 63)];
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 12345.678f) sink[0] = r;
}

// sixteen DIFFERENT operand pairs (one per accumulator): more operand toggling than the constant-operand loop above
template <int NACC>
__global__ __launch_bounds__(512) void mfma_bf16_var_kernel(const uint4 *__restrict__ seed, float *__restrict__ sink, int iters) {
  bf16x8 a[NACC], b[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x + 7 * i) & 127]);
    b[i] = __builtin_bit_cast(bf16x8, seed[(threadIdx.x + 13 * i + 5) & 127]);
  }
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[i], acc[i], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (r == 12345.678f) sink[0] = r;
}

// calibration of s_memtime (the clock of the kernels' cycle stamps): ticks across a register-resident MFMA loop of known length
__global__ __launch_bounds__(256) void mfma_ticks_kernel(const uint4 *__restrict__ seed, float *__restrict__ sink, unsigned long long *ticks, int iters) {
  const uint4 s0 = seed[threadIdx.x & 63], s1 = seed[64 + (threadIdx.x & 63)];
  const bf16x8 a = __builtin_bit_cast(bf16x8, s0), b = __builtin_bit_cast(bf16x8, s1);
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
  if (r == 12345.678f) sink[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // ---- HBM copy: 2 GiB -> 2 GiB
  const size_t bytes = (size_t)2 << 30, n = bytes / 16;
  float4 *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  for (int blocks : {cus * 8, cus * 16, cus * 32}) {
    hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, src, dst, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    const double t = time_ms(e0, e1) / 5 * 1e-3;
    printf("HBM float4 copy, %5d blocks: %.2f TB/s (read + write of 2 GiB each in %.3f ms)\n", blocks, 2.0 * bytes / t / 1e12, t * 1e3);
  }
  // ---- MFMA loops
  std::vector<unsigned> h(512);
  srand(1);
  for (auto &v : h) {   // two random bf16 per word, magnitudes ~1
    const unsigned short lo = (unsigned short)(0x3f00 + (rand() & 0xff)) | ((rand() & 1) << 15);
    const unsigned short hi = (unsigned short)(0x3f00 + (rand() & 0xff)) | ((rand() & 1) << 15);
    v = (unsigned)lo | ((unsigned)hi << 16);
  }
  unsigned *seed; float *sink;
  hipMalloc(&seed, 2048); hipMalloc(&sink, 64);
  hipMemcpy(seed, h.data(), 2048, hipMemcpyHostToDevice);
  std::vector<float> hf(128);
  for (auto &v : hf) v = (float)rand() / RAND_MAX - 0.5f;
  float *seedf; hipMalloc(&seedf, 512); hipMemcpy(seedf, hf.data(), 512, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int wps : {1, 2}) {
    const int threads = 256 * wps;
    hipLaunchKernelGGL((mfma_bf16_kernel<16>), dim3(cus), dim3(threads), 0, 0, (const uint4 *)seed, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((mfma_bf16_kernel<16>), dim3(cus), dim3(threads), 0, 0, (const uint4 *)seed, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    double t = time_ms(e0, e1) / 3 * 1e-3;
    double flops = (double)cus * (threads / 64) * iters * 16 * (2.0 * 16 * 16 * 32);
    printf("MFMA bf16 16x16x32, %d wave(s)/SIMD, random operands: %.0f TFLOP/s\n", wps, flops / t / 1e12);
    hipLaunchKernelGGL((mfma_f32_kernel<16>), dim3(cus), dim3(threads), 0, 0, seedf, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((mfma_f32_kernel<16>), dim3(cus), dim3(threads), 0, 0, seedf, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    t = time_ms(e0, e1) / 3 * 1e-3;
    flops = (double)cus * (threads / 64) * iters * 16 * (2.0 * 16 * 16 * 4);
    printf("MFMA f32  16x16x4,  %d wave(s)/SIMD, random operands: %.1f TFLOP/s\n", wps, flops / t / 1e12);
  }
  // ---- what does s_memtime count?  ticks over a loop of 16 x iters MFMAs per wave (one wave per SIMD) against its wall time
  {
    unsigned long long *ticks; hipMalloc(&ticks, 8);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_ticks_kernel, dim3(cus), dim3(256), 0, 0, (const uint4 *)seed, sink, ticks, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
      const double t = time_ms(e0, e1) * 1e-3, n = 16.0 * iters;
      printf("s_memtime: %.1f ticks per 16x16x32 bf16 MFMA (one wave per SIMD), %.0f MHz tick rate by wall time, %.1f ns per MFMA\n", (double)h / n,
             (double)h / t / 1e6, t / n * 1e9);
    }
  }
  // ---- sustained: ~1.5 s of back-to-back bf16 MFMA launches (two waves per SIMD, sixteen operand pairs), one reading per ~100 ms
  //      window, and the clock / socket power rocm-smi reports half a second in (the 2.4 ms bursts above run before DVFS reacts)
  {
    const int threads = 512, per = 20, windows = 15;
    std::vector<hipEvent_t> ev(windows + 1);
    for (auto &e : ev) hipEventCreate(&e);
    hipLaunchKernelGGL((mfma_bf16_var_kernel<16>), dim3(cus), dim3(threads), 0, 0, (const uint4 *)seed, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(ev[0]);
    for (int w = 0; w < windows; ++w) {
      for (int r = 0; r < per; ++r)
        hipLaunchKernelGGL((mfma_bf16_var_kernel<16>), dim3(cus), dim3(threads), 0, 0, (const uint4 *)seed, sink, iters);
      hipEventRecord(ev[w + 1]);
    }
    hipEventSynchronize(ev[4]);
    if (FILE *f = popen("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' ", "r")) {
      char line[512];
      while (fgets(line, sizeof line, f)) printf("  [rocm-smi during the sustained run] %s", line);
      pclose(f);
    }
    hipEventSynchronize(ev[windows]);
    const double flops = (double)cus * (threads / 64) * iters * 16 * (2.0 * 16 * 16 * 32) * per;
    printf("MFMA bf16 16x16x32 sustained, 2 waves/SIMD, 16 operand pairs, TFLOP/s per ~100 ms window:");
    for (int w = 0; w < windows; ++w) printf(" %.0f", flops / (time_ms(ev[w], ev[w + 1]) * 1e-3) / 1e12);
    printf("\n");
  }
  return 0;
}
