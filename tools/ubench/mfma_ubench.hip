// MFMA issue-rate / clock microbenchmark for gfx950:  hipcc --offload-arch=gfx950 -O3 mfma_ubench.hip -o mfma_ubench
// Every wave runs `iters` x 16 independent v_mfma_f32_16x16x32_bf16 (no memory traffic).  Reports achieved TFLOP/s (HIP events),
// shader cycles per MFMA per SIMD (s_memtime) and the effective shader clock under MFMA load (s_memtime vs the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out; long long* clk;
  for (int wpb = 4; wpb <= 8; wpb += 4) {       // waves per block: 4 = one per SIMD, 8 = two per SIMD
    const int blocks = 256 * 4, threads = wpb * 64;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<<<blocks, threads>>>(out, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<<<blocks, threads>>>(out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * wpb * iters * 16 * 16384.0;
    printf("waves/block %d: %.1f TFLOP/s | %.2f shader cycles per MFMA per wave | shader clock under load %.0f MHz | kernel %.2f ms\n",
           wpb, flops / ms * 1e-9, (double)h[0] / (iters * 16.0), (double)h[0] / ((double)h[1] / 100.0), ms);
    hipFree(out); hipFree(clk);
  }
  return 0;
}
