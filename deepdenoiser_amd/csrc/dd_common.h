// Shared device/host helpers for libdd_hip (gfx950 only; wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dd_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef uint16_t bf16_t;  // storage type of a bf16 element

#define DD_LDS_ROW 128  // bytes of one pixel (or weight row) per K-slice in LDS
#define DD_TILE 16      // 16x16 output pixels per workgroup

// ------------------------------------------------------------------------------------------------ host-side errors
void dd_set_error(const char* fmt, ...);
#define DD_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      dd_set_error(__VA_ARGS__);              \
      return DD_ERR_INVALID;                  \
    }                                         \
  } while (0)
#define DD_LAUNCH_CHECK()                                                   \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dd_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return DD_ERR_LAUNCH;                                                 \
    }                                                                       \
  } while (0)

// ------------------------------------------------------------------------------------------------ per-device launch state
// (csrc/dd_pointwise.hip) One process may drive several devices: what the launchers cache is keyed by the CURRENT device.
int dd_device_cus();                          // compute units of the current device
void dd_allow_max_lds(const void* kernel, int bytes = 160 * 1024);    // hipFuncAttributeMaxDynamicSharedMemorySize, once per (device, kernel)

// ------------------------------------------------------------------------------------------------ element helpers
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// round to nearest even in hardware: the native __bf16 casts lower to v_cvt_pk_bf16_f32 on gfx950 (one VALU op per PAIR;
// a software rounding sequence costs ~7 ops per element and made the conv epilogue VALU-issue bound)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// fp16 storage (DD_F16): a distinct 2-byte type so that templates can tell it from bf16 (a plain uint16_t).  Same MFMA rate as bf16
// (v_mfma_f32_16x16x32_f16), 3 more mantissa bits, range +-65504: the inference type of BASELINE cfg-5 (inputs are log1p-standardized).
struct f16_t { uint16_t v; };
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {      // round to nearest even (v_cvt_f16_f32 / v_cvt_pk_f16_f32)
  const f16x2_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack_f16x2(uint32_t w, float& lo, float& hi) {
  const f16x2_t v = __builtin_bit_cast(f16x2_t, w);
  lo = (float)v[0]; hi = (float)v[1];
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int PER16 = 4;  // elements per 16 bytes
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int PER16 = 8;
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};
template <> struct Elem<f16_t> {
  static constexpr int PER16 = 8;
  static __device__ __forceinline__ float to_f32(f16_t v) { return f16_to_f32(v.v); }
  static __device__ __forceinline__ f16_t from_f32(float v) { return f16_t{__builtin_bit_cast(uint16_t, (_Float16)v)}; }
};
// two floats -> one packed pair of T (2-byte types)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }

// relu on a 16-byte vector of T
template <typename T> __device__ __forceinline__ uint4 relu16(uint4 v);
template <> __device__ __forceinline__ uint4 relu16<float>(uint4 v) {
  v.x = __float_as_uint(fmaxf(__uint_as_float(v.x), 0.f));
  v.y = __float_as_uint(fmaxf(__uint_as_float(v.y), 0.f));
  v.z = __float_as_uint(fmaxf(__uint_as_float(v.z), 0.f));
  v.w = __float_as_uint(fmaxf(__uint_as_float(v.w), 0.f));
  return v;
}
// relu on two packed bf16 / fp16 values: as 16-bit integers a negative float (sign bit set) is a negative integer and a non-negative one keeps
// its order, so max(., 0) per half IS relu (-0.0 -> +0.0): one v_pk_max_i16 instead of four bit operations per word
typedef short dd_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
  const dd_s16x2 v = __builtin_bit_cast(dd_s16x2, w), z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, z));
}
template <> __device__ __forceinline__ uint4 relu16<bf16_t>(uint4 v) {
  v.x = relu_bf16x2(v.x); v.y = relu_bf16x2(v.y); v.z = relu_bf16x2(v.z); v.w = relu_bf16x2(v.w);
  return v;
}
template <> __device__ __forceinline__ uint4 relu16<f16_t>(uint4 v) { return relu16<bf16_t>(v); }   // sign bit 15 in both formats

// 4 consecutive elements <-> float[4]
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  unpack_f16x2(t.x, v[0], v[1]); unpack_f16x2(t.y, v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = pack_f16x2(v[0], v[1]);
  t.y = pack_f16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = pack_bf16x2(v[0], v[1]);
  t.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

// 8 bf16 (one 16-byte vector) <-> float[8]
__device__ __forceinline__ void unpack8(uint4 u, float (&v)[8]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  u.z = pack_bf16x2(v[4], v[5]);
  u.w = pack_bf16x2(v[6], v[7]);
  return u;
}
// 8 elements of a 2-byte type T (one 16-byte vector) <-> float[8]
template <typename T> __device__ __forceinline__ void unpack8t(uint4 u, float (&v)[8]);
template <> __device__ __forceinline__ void unpack8t<bf16_t>(uint4 u, float (&v)[8]) { unpack8(u, v); }
template <> __device__ __forceinline__ void unpack8t<f16_t>(uint4 u, float (&v)[8]) {
  unpack_f16x2(u.x, v[0], v[1]); unpack_f16x2(u.y, v[2], v[3]); unpack_f16x2(u.z, v[4], v[5]); unpack_f16x2(u.w, v[6], v[7]);
}
template <typename T> __device__ __forceinline__ uint4 pack8t(const float (&v)[8]);
template <> __device__ __forceinline__ uint4 pack8t<bf16_t>(const float (&v)[8]) { return pack8(v); }
template <> __device__ __forceinline__ uint4 pack8t<f16_t>(const float (&v)[8]) {
  uint4 u;
  u.x = pack_f16x2(v[0], v[1]); u.y = pack_f16x2(v[2], v[3]); u.z = pack_f16x2(v[4], v[5]); u.w = pack_f16x2(v[6], v[7]);
  return u;
}
// keep the bf16 / fp16 lanes of `v` whose mask lane is > 0 (sign bit clear and not zero: the same test in both formats) (ReLU-backward on packed data, no unpacking)
// the same mask in compare-and-select form (the first version).  Kept for csrc/dd_conv_rw.hip, whose masks arrive by global loads: there the
// packed form above measured 12-29 % SLOWER per launch (96->96 data gradient 104 -> 116 us, 96->192 213 -> 275 us), while it is the faster one
// where the mask is read from LDS (csrc/dd_conv_bwd.hip: -18 us per launch)
__device__ __forceinline__ uint32_t mask_bf16x2_cmp(uint32_t v, uint32_t m) {
  const uint32_t lo = ((m & 0x8000u) == 0u && (m & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
  const uint32_t hi = ((m & 0x80000000u) == 0u && (m & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
  return v & (lo | hi);
}
__device__ __forceinline__ uint32_t mask_bf16x2(uint32_t v, uint32_t m) {
  // m > 0 as a float <=> m > 0 as a 16-bit integer (sign clear, not zero).  0 - m (saturating: -(-32768) must not wrap) is negative exactly
  // then; its arithmetic shift by 15 is the 0xffff / 0 keep mask of the half: three packed instructions per word
  const dd_s16x2 mm = __builtin_bit_cast(dd_s16x2, m), z = {0, 0};
  const dd_s16x2 neg = __builtin_elementwise_sub_sat(z, mm);
  const dd_s16x2 keep = neg >> 15;
  return v & __builtin_bit_cast(uint32_t, keep);
}
__device__ __forceinline__ uint4 mask_bf16x8(uint4 v, uint4 m) {
  v.x = mask_bf16x2(v.x, m.x); v.y = mask_bf16x2(v.y, m.y); v.z = mask_bf16x2(v.z, m.z); v.w = mask_bf16x2(v.w, m.w);
  return v;
}

template <typename T> __device__ __forceinline__ float ld1(const T* p) { return Elem<T>::to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { *p = Elem<T>::from_f32(v); }

// ------------------------------------------------------------------------------------------------ LDS tile image
// One "pixel row" of an LDS tile is DD_LDS_ROW = 128 bytes = 8 slots of 16 bytes (64 bf16 / 32 f32 channels).
// Slot s of row r is stored at physical slot s ^ (r & 7): MFMA fragment reads (16 rows x one 16-byte k-group per
// quarter-wave) then hit 16 distinct 16-byte bank groups per ds_read_b128 service group (conflict-free for any base row).
__device__ __forceinline__ int lds_off(int row, int slot) { return row * DD_LDS_ROW + ((slot ^ (row & 7)) << 4); }
// Pixel tiles ([py][px] with `pw` pixels per row) key the swizzle on the COLUMN only: a fragment read covers 16 consecutive px of
// one row, so this is just as conflict-free, and the lane-dependent part of a read address no longer depends on the row / tap dy
// (9 taps x 4 rows collapse to 3 x 2 address registers + immediate row offsets -- matters under the 256-VGPR cap).
__device__ __forceinline__ int lds_pix_off(int py, int px, int pw, int slot) { return (py * pw + px) * DD_LDS_ROW + ((slot ^ (px & 7)) << 4); }

// 16 bytes of zeros: invalid vectors (outside the image / channels >= cin) are read from here, so every tile load is
// unconditional and needs no select afterwards (one copy per translation unit).
static __device__ uint4 dd_zero16_v = {0u, 0u, 0u, 0u};

// MFMA on one 16-byte k-group: bf16 -> one 16x16x32 MFMA; f32 -> four exact-f32 16x16x4 MFMAs (k permuted
// consistently for both operands, which leaves the sum unchanged).
template <typename T> __device__ __forceinline__ f32x4_t mma16(uint4 a, uint4 b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mma16<bf16_t>(uint4 a, uint4 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma16<f16_t>(uint4 a, uint4 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma16<float>(uint4 a, uint4 b, f32x4_t c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  return c;
}

// ------------------------------------------------------------------------------------------------ deterministic reductions (DD_DETERMINISTIC=1)
// Every weight / bias / loss reduction across workgroups ends in fp32 atomics: the sum's ORDER depends on which workgroup finishes first, so two runs
// of the same step differ in the last bits (and after an optimizer step, everywhere).  With DD_DETERMINISTIC=1 in the environment the workgroups of
// a launch take turns: workgroup b spins on a ticket until it reads b, issues its atomics, waits until they have been performed (agent-scope
// release), and hands the ticket to b + 1 (the last one resets it for the next launch).  Every address then receives the workgroups' partial sums in
// workgroup order -- bit-reproducible run to run; a debugging mode: the flushes of a launch are serialised (a step takes several times longer).
// Per translation unit: its own ticket (kernels of one stream run one after the other), set up by dd_det_sync() from the launcher; nothing
// is read or written on the default path beyond one load of the mode word in front of a flush.  Workgroups are dispatched in index order, so
// the workgroup a ticket waits for is always resident or finished.
static __device__ unsigned dd_det_state[2];      // [0] = mode on, [1] = ticket
__device__ __forceinline__ bool dd_det_on() { return __hip_atomic_load(&dd_det_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
__device__ __forceinline__ unsigned dd_det_block() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
// in front of a wave's atomics (any number of times per launch; wave-uniform control flow)
__device__ __forceinline__ void dd_det_wait() {
  if (!dd_det_on()) return;
  const unsigned b = dd_det_block();
  while (__hip_atomic_load(&dd_det_state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != b) __builtin_amdgcn_s_sleep(8);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// at the END of the kernel, reached by every thread of the workgroup (also by waves that had nothing to add)
__device__ __forceinline__ void dd_det_end() {
  if (!dd_det_on()) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the fence's own wait: MI355X_MICROARCH.md, compiler hazard)
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const unsigned b = dd_det_block(), nb = gridDim.x * gridDim.y * gridDim.z;
    __hip_atomic_store(&dd_det_state[1], b + 1 == nb ? 0u : b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// host: before launching a kernel with atomics (reads the environment once; arms the mode word of this translation unit once per device)
static inline void dd_det_sync() {
  static int env = -1;
  if (env < 0) { const char* e = getenv("DD_DETERMINISTIC"); env = (e && e[0] && e[0] != '0') ? 1 : 0; }
  if (!env) return;
  static unsigned long long armed = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if ((armed >> (dev & 63)) & 1ull) return;
  const unsigned init[2] = {1u, 0u};
  // (synchronous copy: fails during hipGraph capture -- the mode is then NOT armed and the next launch outside a capture tries again; a run
  //  that asked for DD_DETERMINISTIC=1 must not be reported deterministic on unordered atomics, so the failure is loud.  ADVICE r5)
  if (hipMemcpyToSymbol(HIP_SYMBOL(dd_det_state), init, sizeof(init)) != hipSuccess) {
    (void)hipGetLastError();
    fprintf(stderr, "libdd_hip: DD_DETERMINISTIC=1 could not be armed on device %d (first launch of a translation unit inside a graph capture?): run one eager step first\n", dev);
    return;
  }
  armed |= 1ull << (dev & 63);
}

// Host-side dispatch on the storage dtype: `T` is float / bf16_t / f16_t inside the statement.
#define DD_DISPATCH_DTYPE(dtype, T, ...)                     \
  do {                                                       \
    if ((dtype) == DD_F32) { using T = float; __VA_ARGS__; } \
    else if ((dtype) == DD_BF16) { using T = bf16_t; __VA_ARGS__; } \
    else { using T = f16_t; __VA_ARGS__; }                   \
  } while (0)
static inline bool dd_dtype_ok(int dtype) { return dtype == DD_F32 || dtype == DD_BF16 || dtype == DD_F16; }

static inline int dd_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
