// The fragment loop of the register-weight 3x3 conv (csrc/dd_conv_rw.hip, conv_rw8_kernel<T, 2, 8>) on LDS-RESIDENT data, as 16x16x32 and as
// 32x32x16 MFMAs: what would `v_mfma_f32_32x32x16_bf16` buy that kernel?  No global traffic inside the timed loop: a
// workgroup of 8 waves multiplies the same random LDS image `tiles` times (one barrier per tile, as the real kernel has), weights in registers.
//   MODE 0: 16x16x32, 16 x 16 tile, wave = 16 output channels x 8 rows (10 haloed rows x 3 shifts x 2 K chunks = 60 ds_read_b128, 144 MFMAs)
//   MODE 1: 32x32x16, 32 x 8 tile,  wave = 32 output channels x 2 rows (4 haloed rows x 3 shifts x 4 K steps = 48 reads, 72 MFMAs)
//   MODE 2: 32x32x16, 32 x 16 tile, wave = 32 output channels x 4 rows (6 haloed rows x 12 = 72 reads, 144 MFMAs; 34 x 18 x 128 B = 78 KB of LDS)
// All three do 64 -> 64 channels: 2 * 9 * 64 * 64 flop per output pixel.  Fragment layout of the 32-pixel forms: 128-byte pixel rows, 16-byte slot
// s of pixel p at s ^ ((p >> 1) & 7): the 16 lanes of a ds_read_b128 service group ({0-3, 12-15, 20-27} + c) hold 8 even and 8 odd pixels whose
// (p >> 1) & 7 are distinct -> conflict-free for every column shift.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rw_loop_ubench.hip -o tools/exp/rw_loop_ubench && tools/exp/rw_loop_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const u32x4* __restrict__ wts, uint32_t* __restrict__ out, int tiles) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int LDS_BYTES = MODE == 2 ? 34 * 18 * 128 : (MODE == 1 ? 34 * 10 * 128 : 18 * 18 * 128);
  for (int i = tid; i < LDS_BYTES / 4; i += 512) {      // small random bf16 pairs (exponent near 1.0: no denormals, no overflow)
    const uint32_t h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u | (h & 0x807f807fu);
  }
  __syncthreads();
  uint32_t chk = 0;
  if constexpr (MODE == 0) {
    constexpr int PW = 18, RH = 8, PHW = RH + 2, KC = 2, FR = 3 * KC, NF = PHW * FR, RING = 6, AHEAD = RING - 1;
    const int li = lane & 15, q = lane >> 4, half = wave >> 2;
    u32x4 wf[9][KC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) wf[t][kc] = wts[((wave & 3) * 18 + t * 2 + kc) * 64 + lane];
    unsigned d0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] = lds_base + (half * RH * PW + li) * 128 + ((q ^ ((half * RH * PW + li + c) & 7)) << 4);
    for (int tile = 0; tile < tiles; ++tile) {
      __syncthreads();
      f32x4_t acc[4];
      u32x4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / FR, j = f - FR * yy, dx = j / KC, kc = j - KC * dx, C = yy * PW + dx;
        return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((d0[C & 7] ^ ((kc & 1) << 6)) + C * 128);
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PHW; ++yy) {
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          const int f = yy * FR + j, dx = j / KC, kc = j - KC * dx;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < RH) acc[yy % 4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (j == 2 && yy >= 3) { const f32x4_t v = acc[(yy - 3) % 4]; chk ^= pack_bf16x2(v[0], v[1]) ^ pack_bf16x2(v[2], v[3]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < RH)
              acc[y % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[dy * 3 + dx][kc]), __builtin_bit_cast(bf16x8_t, ring[f % RING]), acc[y % 4], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      { const f32x4_t v = acc[(RH - 1) % 4]; chk ^= pack_bf16x2(v[0], v[1]) ^ pack_bf16x2(v[2], v[3]); }
    }
  } else {
    constexpr int PW = 34, RH = MODE == 1 ? 2 : 4, PHW = RH + 2, KS = 4, FR = 3 * KS, NF = PHW * FR, RING = MODE == 1 ? 3 : 2, AHEAD = RING - 1;
    const int p32 = lane & 31, kg = lane >> 5, rg = wave >> 1;      // pixel of the fragment, 8-channel group of the K step, row group of the wave
    u32x4 wf[9][KS];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[t][ks] = wts[((wave & 1) * 36 + t * 4 + ks) * 64 + lane];
    // fragment address of haloed pixel P = (rg*RH + yy)*PW + dx + p32, K step ks: slot 2 ks + kg at ((2 ks + kg) ^ ((P >> 1) & 7)) << 4.
    // P = base + C with C = yy*PW + dx a compile-time constant: (P >> 1) & 7 depends on (base + C): 16 lane-dependent bases cover C & 15
    unsigned d0[16];
    const int base = rg * RH * PW + p32;
#pragma unroll
    for (int c = 0; c < 16; ++c) d0[c] = lds_base + base * 128 + ((kg ^ (((base + c) >> 1) & 7)) << 4);
    for (int tile = 0; tile < tiles; ++tile) {
      __syncthreads();
      f32x16_t acc[3];      // output rows y % 3: a finished row is packed (8 registers) behind its last MFMA and leaves during the next haloed row
      uint32_t o8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      u32x4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / FR, j = f - FR * yy, dx = j / KS, ks = j - KS * dx, C = yy * PW + dx;
        return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((d0[C & 15] ^ (ks << 5)) + C * 128);
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PHW; ++yy) {
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          const int f = yy * FR + j, dx = j / KS, ks = j - KS * dx;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < RH) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[yy % 3][e] = 0.f;
          }
          if (j >= 2 && j < 6 && yy >= 3) chk ^= o8[2 * (j - 2)] ^ o8[2 * (j - 2) + 1];      // (the real kernel: four 8-byte stores of 4 channels each)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < RH)
              acc[y % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[dy * 3 + dx][ks]), __builtin_bit_cast(bf16x8_t, ring[f % RING]), acc[y % 3], 0, 0, 0);
          }
          if (j == FR - 1 && yy >= 2) {
            const f32x16_t v = acc[(yy - 2) % 3];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) chk ^= o8[e];
    }
  }
  out[blockIdx.x * 512 + tid] = chk;
}

template <int MODE>
static void run(const char* name, int px_per_tile, size_t lds) {
  std::vector<uint32_t> hw(72 * 64 * 4);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c003c00u | ((uint32_t)(i * 2246822519u) & 0x807f807fu);
  u32x4* w; uint32_t* out;
  (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int tiles = 2000 * 256 / px_per_tile;
  loop_kernel<MODE><<<256, 512, lds>>>(w, out, 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    loop_kernel<MODE><<<256, 512, lds>>>(w, out, tiles);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = 256.0 * tiles * px_per_tile * 2.0 * 9 * 64 * 64;
  printf("%-44s %8.2f ms  %7.1f TFLOP/s = %4.1f %% of 2500\n", name, best, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12 / 25.0);
  (void)hipFree(w); (void)hipFree(out);
}

int main() {
  run<0>("16x16x32, 16x16 tile, wave 16 ch x 8 rows", 256, 18 * 18 * 128);
  run<1>("32x32x16, 32x8 tile,  wave 32 ch x 2 rows", 256, 34 * 10 * 128);
  run<2>("32x32x16, 32x16 tile, wave 32 ch x 4 rows", 512, 34 * 18 * 128);
  return 0;
}
