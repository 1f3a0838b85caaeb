// Per-CU global store / load throughput vs. waves per CU (16 B per lane, 1 KiB per wave-instruction, streaming, persistent blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void store_k(uint4* out, long per_wave_vecs, int waves_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* base = out + ((long)blockIdx.x * waves_per_block + wave) * per_wave_vecs * 64;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (long i = 0; i < per_wave_vecs; ++i) base[i * 64 + lane] = v;
}
__global__ void load_k(const uint4* in, uint4* sink, long per_wave_vecs, int waves_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* base = in + ((long)blockIdx.x * waves_per_block + wave) * per_wave_vecs * 64;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 8
  for (long i = 0; i < per_wave_vecs; ++i) { uint4 v = base[i * 64 + lane]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
int main() {
  const long bytes = 1L << 30;
  uint4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cus = 256;
  for (int mode = 0; mode < 2; ++mode)
    for (int wpb : {1, 2, 4, 8, 16}) {
      const int blocks = cus;                       // one persistent block per CU
      const long total_vecs = bytes / 16 / 4;       // use a quarter of the buffer (256 MiB) per run
      const long per_wave = total_vecs / 64 / ((long)blocks * wpb);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(store_k, dim3(blocks), dim3(64 * wpb), 0, 0, b, per_wave, wpb);
        else hipLaunchKernelGGL(load_k, dim3(blocks), dim3(64 * wpb), 0, 0, a, b, per_wave, wpb);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double gb = (double)per_wave * 64 * 16 * blocks * wpb / 1e9;
      printf("%s waves/CU %2d : %7.1f us  %6.2f TB/s  (%5.1f B/clk/CU @2.1GHz, %6.0f cycles per 1KiB wave-instr)\n", mode == 0 ? "store" : "load ", wpb,
             ms * 1e3, gb / ms, gb * 1e9 / (ms * 1e-3) / 256 / 2.1e9, ms * 1e-3 * 2.1e9 / per_wave);
    }
  return 0;
}
