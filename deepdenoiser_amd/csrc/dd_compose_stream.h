// Shared pieces of the row-streaming compose kernels (csrc/dd_compose_stream.hip: forward; csrc/dd_compose_stream_bwd.hip: backward):
// MFMA 32x32x16 wrappers, packing helpers, the unit / cursor bookkeeping of a workgroup's stream of virtual rows.  See the forward file for
// the scheme.  Everything sits in an anonymous namespace: one copy per translation unit.
#pragma once
#include "dd_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) unsigned int cs_u32x2;

template <typename T> __device__ __forceinline__ f32x16_t mma32(uint4 a, uint4 b, f32x16_t c);
template <> __device__ __forceinline__ f32x16_t mma32<bf16_t>(uint4 a, uint4 b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16_t mma32<f16_t>(uint4 a, uint4 b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// The first MFMA of a chain, from a ZERO accumulator given as the instruction's inline constant.  (Through the builtin hipcc keeps a vector of
// 16 zero registers alive across the whole step loop as that operand -- or moves zeros into the accumulator every time.)  The chain that
// follows reads the result as its C operand, which needs no wait states; the operands come straight from LDS reads, whose waits the compiler
// inserts for inline asm as for any other use.
typedef __attribute__((ext_vector_type(4))) unsigned int cs_u32x4;
template <typename T> __device__ __forceinline__ f32x16_t mma32_zero(uint4 a, uint4 b);
template <> __device__ __forceinline__ f32x16_t mma32_zero<bf16_t>(uint4 a, uint4 b) {
  f32x16_t d;
  const cs_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(av), "v"(bv));
  return d;
}
template <> __device__ __forceinline__ f32x16_t mma32_zero<f16_t>(uint4 a, uint4 b) {
  f32x16_t d;
  const cs_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(av), "v"(bv));
  return d;
}

template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float& lo, float& hi) { unpack_f16x2(w, lo, hi); }
// One conversion instruction per pair.  (Followed directly by an integer operation on the halves -- the packed ReLU, a select -- hipcc converts
// each value on its own and merges them with a v_perm_b32: three instructions per pair.  The empty asm hides the origin of the word.)
template <typename T> __device__ __forceinline__ uint32_t packo(float lo, float hi) {
  uint32_t w = pack2<T>(lo, hi);
  asm("" : "+v"(w));
  return w;
}
template <typename T> struct One;      // 1.0 in the storage type
template <> struct One<bf16_t> { static constexpr uint32_t v = 0x3f80u; };
template <> struct One<f16_t> { static constexpr uint32_t v = 0x3c00u; };

// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2]: a vector-ALU move, no LDS crossbar)
__device__ __forceinline__ float dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
// v of lane l % 32 and of lane 32 + l % 32, in every lane (v_permlane32_swap)
__device__ __forceinline__ void both_halves(float v, float& lower, float& upper) {
  const cs_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  lower = __uint_as_float(r[0]); upper = __uint_as_float(r[1]);
}

constexpr int PIXB = 48;          // bytes of one pixel: 24 channels x 2

// Geometry of a launch (dd_compose_stream_plan): frame width (multiple of 32, <= 128), rows per step, 32-pixel tasks per row, column strips and
// their output width, band height, bands per image, virtual rows per band (BH + 8), units = N * n_strips * nb.  The parameter structs of the
// kernels carry these fields under these names.
struct Unit { int b, yb0, yb1, xs, xe, fx0; };

template <typename P>
__device__ __forceinline__ void decode_unit(const P& p, int u, Unit& U) {
  const int j = u % p.nb, t = u / p.nb;
  const int st = t % p.n_strips;
  U.b = t / p.n_strips;
  U.yb0 = j * p.BH; U.yb1 = min(p.H, U.yb0 + p.BH);
  U.xs = st * p.SO; U.xe = min(p.W, U.xs + p.SO);
  U.fx0 = p.n_strips > 1 ? U.xs - 4 : 0;
}

// Position of a stage in the workgroup's stream of virtual rows: unit index relative to the workgroup's first unit (negative while the
// pipeline fills) and the row inside the unit's VB virtual rows.
struct Cursor {
  int urel, i;
  Unit U;
  template <typename P>
  __device__ __forceinline__ void init(const P& p, int u0, int nunits, int V) {
    urel = V >= 0 ? V / p.VB : -1 - ((-V - 1) / p.VB);
    i = V - urel * p.VB;
    U = Unit{0, 0, 0, 0, 0, 0};
    if (urel >= 0 && urel < nunits) decode_unit(p, u0 + urel, U);
  }
  template <typename P>
  __device__ __forceinline__ void advance(const P& p, int u0, int nunits, int rows) {
    i += rows;
    while (i >= p.VB) {
      i -= p.VB;
      ++urel;
      if (urel >= 0 && urel < nunits) decode_unit(p, u0 + urel, U);
    }
  }
};

__device__ __forceinline__ int sfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int posmod(int v, int d) { const int m = v % d; return m < 0 ? m + d : m; }

// fp32 -> three storage-type terms whose sum is the value to 24 bits (hi + mid + lo); term e of the split
template <typename T> __device__ __forceinline__ float split3(float v, int e) {
  const float hi = Elem<T>::to_f32(Elem<T>::from_f32(v));
  const float mid = Elem<T>::to_f32(Elem<T>::from_f32(v - hi));
  const float lo = Elem<T>::to_f32(Elem<T>::from_f32(v - hi - mid));
  return e == 0 ? hi : e == 1 ? mid : lo;
}

}  // namespace
