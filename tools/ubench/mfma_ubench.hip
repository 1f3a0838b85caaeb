// MFMA issue-rate microbenchmark for gfx950:  hipcc --offload-arch=gfx950 -O3 mfma_ubench.hip -o mfma_ubench
// Every wave runs iters x 16 independent v_mfma_f32_16x16x32_bf16, optionally with one ds_read_b128 per `MFMA_PER_READ` MFMAs.
// Sweeps waves per SIMD (1, 2, 4) to show how much of the matrix pipe ONE wave can fill on its own.
// Reports TFLOP/s (HIP events) and s_memtime ticks per MFMA as seen by one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int READS>   // ds_read_b128 per 16 MFMAs (0, 4, 8)
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters) {
  __shared__ uint4 lds[1024];
  lds[threadIdx.x] = uint4{threadIdx.x, 1u, 2u, 3u};
  lds[threadIdx.x + 256] = uint4{threadIdx.x, 5u, 2u, 3u};
  __syncthreads();
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 a[2][4], b[2][4];     // register double buffer: iteration `it` computes from [it & 1] while [1 - (it & 1)] is being loaded
  for (int i = 0; i < 4; ++i) { a[0][i] = a[1][i] = lds[(threadIdx.x + i * 64) & 1023]; b[0][i] = b[1][i] = lds[(threadIdx.x * 3 + i) & 1023]; }
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (READS > 0 && (i % (16 / (READS ? READS : 1))) == 0) {
          const int k = i / (16 / (READS ? READS : 1));
          if (k < 4) a[1 - h][k] = lds[(threadIdx.x + it * 64 + k * 64) & 1023]; else b[1 - h][k - 4] = lds[(threadIdx.x + it * 64 + k * 64) & 1023];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[h][i >> 2]), __builtin_bit_cast(bf16x8_t, b[h][i & 3]), acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int READS>
void run(int iters, float* out, long long* clk) {
  for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD = co-resident 4-wave blocks per CU
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<READS><<<blocks, 256>>>(out, clk, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<READS><<<blocks, 256>>>(out, clk, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 16 * 16384.0;
    printf("ds_read_b128 per 16 MFMA: %d | waves/SIMD %d: %7.1f TFLOP/s | %.2f ticks per MFMA (one wave) | kernel %.2f ms\n",
           READS, wps, flops / ms * 1e-9, (double)h / (iters * 16.0), ms);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out; long long* clk;
  (void)hipMalloc(&out, sizeof(float) * 1024 * 256);
  (void)hipMalloc(&clk, 16);
  run<0>(iters, out, clk);
  run<4>(iters, out, clk);
  run<8>(iters, out, clk);
  return 0;
}
