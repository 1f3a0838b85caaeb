// How many LDS-DMA units must a CU keep in flight to stream at the copy rate?  One persistent 8-wave workgroup per CU walks its share of a 1 GiB
// buffer in units of UNIT KiB (global_load_lds_dwordx4, 1 KiB per wave instruction, exactly as the conv kernels stage their tiles), with DEPTH
// units in flight: unit u + DEPTH is requested when unit u has landed (s_waitcnt vmcnt(pieces of the DEPTH - 1 younger units) + barrier).  Nothing
// is computed: the time is the memory system's.   hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsdma_depth.hip -o tools/exp/ldsdma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void dma_1k(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory");
}
// COPY: every landed unit is also written back out (ds_read_b128 -> 16-byte stores, 1 KiB per wave instruction): reads + writes at once
template <int UNIT_KB, int DEPTH, bool COPY = false>
__global__ __launch_bounds__(512) void stream_kernel(const char* src, long units_per_wg, unsigned* sink, char* dst = nullptr) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int PPW = UNIT_KB / 8;                       // pieces per wave and unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* base = src + (long)blockIdx.x * units_per_wg * UNIT_KB * 1024;
  auto issue = [&](long u, int buf) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int piece = k * 8 + wave;
      dma_1k(base + (u * UNIT_KB + piece) * 1024 + lane * 16, lds + (buf * UNIT_KB + piece) * 1024);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  unsigned acc = 0;
  int buf = 0;
  for (long u = 0; u < units_per_wg; ++u) {
    // (COPY: the stores of the previous unit sit between the pieces in the counter: they are waited for too -- conservative)
    if (DEPTH == 1 || COPY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PPW) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (COPY) {
      char* out = dst + ((long)blockIdx.x * units_per_wg + u) * UNIT_KB * 1024;
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int piece = k * 8 + wave;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + (buf * UNIT_KB + piece) * 1024 + lane * 16);
        *reinterpret_cast<uint4*>(out + piece * 1024 + lane * 16) = v;
      }
    } else
    acc += *reinterpret_cast<const unsigned*>(smem + (buf * UNIT_KB * 1024) + (threadIdx.x & 255) * 4);      // touch the landed unit
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                            // everyone is done with it
    const long nu = u + DEPTH < units_per_wg ? u + DEPTH : units_per_wg - 1;                                    // (keeps the piece count per unit constant)
    issue(nu, buf);
    buf = buf + 1 == DEPTH ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int UNIT_KB, int DEPTH> float run(const char* src, unsigned* sink, long bytes) {
  const long units_per_wg = bytes / (256L * UNIT_KB * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<UNIT_KB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<UNIT_KB, DEPTH>), dim3(256), dim3(512), (size_t)UNIT_KB * 1024 * DEPTH, 0, src, units_per_wg, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double tb = (double)units_per_wg * 256 * UNIT_KB * 1024 / (best * 1e-3) / 1e12;
  printf("unit %3d KiB, %d in flight (%3d KiB per CU): %7.1f us  %5.2f TB/s  %5.2f us per unit\n", UNIT_KB, DEPTH, UNIT_KB * DEPTH, best * 1e3, tb,
         best * 1e3 / units_per_wg);
  return best;
}
template <int UNIT_KB, int DEPTH> float run_copy(const char* src, char* dst, unsigned* sink, long bytes) {
  const long units_per_wg = bytes / (256L * UNIT_KB * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<UNIT_KB, DEPTH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<UNIT_KB, DEPTH, true>), dim3(256), dim3(512), (size_t)UNIT_KB * 1024 * DEPTH, 0, src, units_per_wg, sink, dst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double tb = 2.0 * units_per_wg * 256 * UNIT_KB * 1024 / (best * 1e-3) / 1e12;
  printf("COPY unit %3d KiB, %d buffers: %7.1f us  %5.2f TB/s read + written  %5.2f us per unit\n", UNIT_KB, DEPTH, best * 1e3, tb, best * 1e3 / units_per_wg);
  return best;
}
int main() {
  const long bytes = 1L << 30;
  char* src; unsigned* sink; char* dst;
  hipMalloc(&src, bytes); hipMalloc(&sink, 64); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  run<16, 1>(src, sink, bytes); run<16, 2>(src, sink, bytes); run<16, 3>(src, sink, bytes);
  run<40, 1>(src, sink, bytes); run<40, 2>(src, sink, bytes); run<40, 3>(src, sink, bytes);
  run<72, 1>(src, sink, bytes); run<72, 2>(src, sink, bytes);
  run_copy<16, 2>(src, dst, sink, bytes); run_copy<40, 2>(src, dst, sink, bytes); run_copy<40, 3>(src, dst, sink, bytes); run_copy<72, 2>(src, dst, sink, bytes);
  return 0;
}
