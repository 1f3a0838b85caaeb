// LDS read-rate probe for gfx950: bytes per clock and CU of ds_read_b128, ds_read_b64 and ds_read_b64_tr_b16 under 8 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/exp/lds_probe && tools/exp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int KIND, int KEY>
__global__ __launch_bounds__(512) void probe(uint32_t* out, long long* cycles, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
  unsigned addr;
  if (KIND == 0) {            // b128 row fragments: row li, slot q ^ (li & 7)
    const int li = lane & 15, q = lane >> 4;
    addr = base + li * 128 + ((q ^ (li & 7)) << 4);
  } else {                    // transposing reads: lane (t16, gq): pixel gq*8 + (t16 >> 2), 8 bytes at slot pair (sub >> 1), half sub & 1
    // KEY 0 / 1: the layout of rounds 2-4 (a 32-lane service group holds pixels {0..3, 8..11}: p and p + 8 share p & 7 -> the same banks);
    // KEY 2: a 32-lane group holds 8 CONSECUTIVE pixels (round 5)
    const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3, pl = KEY == 2 ? (gq >> 1) * 16 + (gq & 1) * 4 + (t16 >> 2) : gq * 8 + (t16 >> 2);
    const int key = KEY == 1 ? ((pl & 3) << 1) : (pl & 7);
    addr = base + pl * 128 + (((sub >> 1) ^ key) << 4) + (sub & 1) * 8;
  }
  uint32_t accx = 0;
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (KIND == 0) {
        const u32x4 v = *reinterpret_cast<const volatile __attribute__((address_space(3))) u32x4*>(addr + k * 32);
        accx ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      } else if (KIND == 1) {
        const u32x2 v = *reinterpret_cast<const volatile __attribute__((address_space(3))) u32x2*>(addr + k * 32);
        accx ^= v[0] ^ v[1];
      } else {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(uintptr_t)(addr + k * 32));
        accx ^= (uint32_t)v[0] ^ ((uint32_t)v[1] << 8) ^ ((uint32_t)v[2] << 16) ^ ((uint32_t)v[3] << 24);
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  out[blockIdx.x * 512 + tid] = accx;
  if (tid == 0 && blockIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int KIND, int KEY>
static void run(const char* name, int bytes_per_lane) {
  uint32_t* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND, KEY>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<KIND, KEY><<<256, 512, 65536>>>(out, cyc, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<KIND, KEY><<<256, 512, 65536>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double bytes = 512.0 * bytes_per_lane * 8 * iters;      // per workgroup (= per CU)
  printf("%-34s %8.1f us  %6.1f B/clk/CU (shader clocks %lld)  %6.1f B/clk/CU at 2.4 GHz\n", name, ms * 1e3, bytes / (double)h[0], h[0], bytes / (ms * 1e-3 * 2.4e9));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("ds_read_b128 rows (key p&7)", 16);
  run<1, 0>("ds_read_b64 tr-pattern (key p&7)", 8);
  run<1, 1>("ds_read_b64 tr-pattern (key 2(p&3))", 8);
  run<2, 0>("ds_read_b64_tr_b16 (key p&7)", 8);
  run<2, 1>("ds_read_b64_tr_b16 (key 2(p&3))", 8);
  run<1, 2>("ds_read_b64 8 consecutive px / 32 lanes", 8);
  run<2, 2>("ds_read_b64_tr_b16 8 consecutive px", 8);
  return 0;
}
