// How long does a wave take to ISSUE a burst of 16-byte-per-lane global loads (1 KiB per wave-instruction) on gfx950, and how long
// until the data is back?  hipcc --offload-arch=gfx950 -O3 vmem_issue.hip -o vmem_issue
// Pattern = the tile access of the conv kernels: instruction i reads 8 consecutive pixels (1 KiB contiguous) of image row i, rows `pitch`
// bytes apart.  Sweeps: loads per burst, waves per CU issuing, number of CUs active, row pitch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int N>
__global__ __launch_bounds__(256) void burst(const uint4* __restrict__ src, long pitch16, long block_stride16, int active_waves, int reps,
                                             long long* out, uint4* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= active_waves) return;
  const uint4* base = src + blockIdx.x * block_stride16 + wave * 64 + lane;
  uint4 acc = {0, 0, 0, 0};
  long long t_issue = 0, t_total = 0;
  for (int r = 0; r < reps; ++r) {
    uint4 v[N];
    const uint4* p = base + (long)r * N * pitch16;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i * pitch16];
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
    __builtin_amdgcn_sched_barrier(0);
    const long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0; t_total += t2 - t0;
  }
  if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t_issue; out[1] = t_total; }
}

template <int N>
void run(const uint4* src, long long* out, uint4* sink, int blocks, int waves, long pitch_bytes, size_t bytes) {
  const int reps = 16;
  const long pitch16 = pitch_bytes / 16;
  long block_stride16 = (long)(bytes / 16) / blocks;
  if ((long)reps * N * pitch16 + 256 > block_stride16) { printf("  (skipped: buffer too small)\n"); return; }
  hipLaunchKernelGGL(burst<N>, dim3(blocks), dim3(256), 0, 0, src, pitch16, block_stride16, waves, reps, out, sink);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(burst<N>, dim3(blocks), dim3(256), 0, 0, src, pitch16, block_stride16, waves, reps, out, sink);
  (void)hipDeviceSynchronize();
  long long h[2];
  (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  printf("  CUs %3d  waves/CU %d  pitch %6ld B  burst %2d loads: issue %6.0f cycles (%4.0f per load), data back after %6.0f cycles\n", blocks, waves, pitch_bytes, N,
         (double)h[0] / reps, (double)h[0] / reps / N, (double)h[1] / reps);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint4* src; long long* out; uint4* sink;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(src, 1, bytes);
  for (int blocks : {8, 256})
    for (int waves : {1, 4})
      for (long pitch : {1024L, 16384L}) {
        run<8>(src, out, sink, blocks, waves, pitch, bytes);
        run<16>(src, out, sink, blocks, waves, pitch, bytes);
      }
  return 0;
}
