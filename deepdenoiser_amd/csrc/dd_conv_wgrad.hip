// Weight-gradient GEMM on CDNA4 MFMA (+ fused bias gradient).
//
//   out[t][m][n] += sum_{b, pixel p}  in(P)[b, map_t(p), m] * Q[b, p, n]          (fp32 atomics into the gradient arena)
//
// conv2d            : P = layer input x (3x3: halo patch, 1x1: same pixel), Q = pre-activation output gradient,
//                     out = dKernel in TF HWIO layout [kh*kw][C_in][C_out];  bias gradient = column sums of Q.
// conv2d_transpose  : P = output gradient on the fine grid, map_t(p) = 2p+(a,b) (DD_GATHER2X2), Q = layer input,
//                     out = dKernel in TF layout [a*2+b][C_out][C_in];       bias gradient = column sums of P.
// Replaces the TF-autodiff filter / bias gradients behind tf.train.AdamOptimizer.minimize (reference
// TensorFlow/Training.py:701-702) for every conv in UNet.py / Tiramisu.py / Architecture.py:238-243 / MultiScalePrediction.py.
//
// Both operands are reduction-major in memory (NHWC: the pixel index is the GEMM K dimension), i.e. "transposed" with
// respect to what an MFMA fragment wants (8 consecutive k per lane).  bf16 path: the NHWC tiles stay as they are in
// LDS ([pixel][channel], 128-byte rows, slot-swizzled) and fragments are fetched with the gfx950 transpose read
// ds_read_b64_tr_b16 (a 16-lane group reads a [4 pixels][16 channels] block, each lane receives one channel's 4 pixels;
// pinned by tests/test_gpu_ops.py::test_tr16_probe); a tap shift is a whole-row shift in that image, so the 3x3 halo
// patch is staged once and reused by the 9 taps.  f32 path: exact-f32 MFMA 16x16x4, one element per lane (ds_read_b32).
// A workgroup owns one (m-slice, n-slice) of 128 B worth of channels each, ALL taps, and a strided subset of the
// 16x16 pixel tiles (split-K); partial results are added with one fp32 atomic per element at the end.
// The bias gradient costs no extra pass over the gradient tensor: every thread sums the vectors it stages anyway.
// All global loads are unconditional (invalid vectors read a zero page): a branch per vector serialises them.
#include <stdlib.h>

#include "dd_common.h"

#ifdef DD_PROFILE_PHASES
__device__ unsigned long long dd_wphase_cycles[16];
#define WPHASE_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WPHASE_ADD(i, a, b) if (blockIdx.x == 0 && threadIdx.x == 0) dd_wphase_cycles[i] += (b) - (a)
extern "C" int dd_debug_wphases(unsigned long long* out16, int reset) {
  if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(dd_wphase_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dd_wphase_cycles), z, sizeof(z)); }
  return 0;
}
#else
#define WPHASE_T(var)
#define WPHASE_ADD(i, a, b)
#endif

namespace {

struct WgradP {
  const void* p; const void* q; float* out; float* bias_out;
  int ldp, m, ldq, n, mv, nv;   // mv/nv: staged (16-byte rounded) channel counts
  int B, H, W, taps, flags, bias_mode;
  int tiles_x, tiles_y, ksplit, mslices, nslices;
  int hin, win;
};

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// The transpose reads (ds_read_b64_tr_b16) have their own bank-conflict classes: keying the swizzle on the linear pixel index is
// measured 1.4x faster for this kernel than the column-keyed image the forward kernel uses (64->64 @128^2: 92 vs 130 us).
__device__ __forceinline__ int wg_off(int py, int px, int pw, int slot) { return lds_off(py * pw + px, slot); }

template <int PH> struct TilePlan {
  static constexpr int NPIX = PH * PH;
  static constexpr int ITERS = (NPIX * 8 + 255) / 256;
  int yx[ITERS];    // (py << 8) | px or -1
  int lds[ITERS];
};
template <int PH>
__device__ __forceinline__ void tile_plan(TilePlan<PH>& pl, int tid) {
#pragma unroll
  for (int it = 0; it < TilePlan<PH>::ITERS; ++it) {
    const int i = tid + it * 256;
    const int pix = i >> 3, slot = i & 7;
    const int py = pix / PH, px = pix - py * PH;
    pl.yx[it] = pix < TilePlan<PH>::NPIX ? ((py << 8) | px) : -1;
    pl.lds[it] = wg_off(py, px, PH, slot);
  }
}

// global -> registers for one K-slice of a pixel tile; (oy,ox) = image coordinate of tile pixel (0,0) on the GEMM-row grid
template <typename T, int PH>
__device__ __forceinline__ void tile_load(uint4 (&reg)[TilePlan<PH>::ITERS], const TilePlan<PH>& pl, const T* __restrict__ X, long img_base, int ld, int cvalid,
                                          int ch0, int oy, int ox, int H, int W, int halo, int sy, int ay, int ax, int hin, int win, int tid) {
  constexpr int PER16 = Elem<T>::PER16;
  const int ch = ch0 + (tid & 7) * PER16;
  const bool ch_ok = ch < cvalid;
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
  const T* base = X + img_base * ld + ch;
#pragma unroll
  for (int it = 0; it < TilePlan<PH>::ITERS; ++it) {
    const int yx = pl.yx[it];
    const int ly = oy + (yx >> 8), lx = ox + (yx & 255);
    const int gy = ly * sy + ay, gx = lx * sy + ax;
    const bool ok = yx >= 0 && ch_ok && ly >= -halo && lx >= -halo && ly < H + halo && lx < W + halo && gy >= 0 && gx >= 0 && gy < hin && gx < win;
    reg[it] = *reinterpret_cast<const uint4*>(ok ? base + ((long)gy * win + gx) * ld : zero);
  }
}
template <typename T, int PH>
__device__ __forceinline__ void tile_store(char* lds, const uint4 (&reg)[TilePlan<PH>::ITERS], const TilePlan<PH>& pl, bool in_relu) {
#pragma unroll
  for (int it = 0; it < TilePlan<PH>::ITERS; ++it) {
    uint4 v = reg[it];
    if (in_relu) v = relu16<T>(v);
    if (pl.yx[it] >= 0) *reinterpret_cast<uint4*>(lds + pl.lds[it]) = v;
  }
}
// per-thread running column sums of the staged vectors (this thread always covers the same 16-byte channel group)
template <typename T, int PH>
__device__ __forceinline__ void bias_accumulate(float (&bsum)[8], const uint4 (&reg)[TilePlan<PH>::ITERS], const TilePlan<PH>& pl, bool interior_only) {
#pragma unroll
  for (int it = 0; it < TilePlan<PH>::ITERS; ++it) {
    const int yx = pl.yx[it];
    const int py = yx >> 8, px = yx & 255;
    const bool use = yx >= 0 && (!interior_only || (py >= 1 && py <= DD_TILE && px >= 1 && px <= DD_TILE));
    if (sizeof(T) == 2) {
      float v[8];
      unpack8(reg[it], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[e] += use ? v[e] : 0.f;
    } else {
      bsum[0] += use ? __uint_as_float(reg[it].x) : 0.f; bsum[1] += use ? __uint_as_float(reg[it].y) : 0.f;
      bsum[2] += use ? __uint_as_float(reg[it].z) : 0.f; bsum[3] += use ? __uint_as_float(reg[it].w) : 0.f;
    }
  }
}

// bf16 fragment (8 k-values of channel `lane&15` of 16-channel tile `ctile`) for k-step rows via two transpose reads.
__device__ __forceinline__ uint4 frag_tr_bf16(const char* base, int row, int pw, int dx, int ctile, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int y = row + (g >> 1), xb = (g & 1) * 8 + (t16 >> 2) + dx;
  const int sub = t16 & 3;
  const int slot = ctile * 2 + (sub >> 1), half = (sub & 1) * 8;
  const char* a0 = base + wg_off(y, xb, pw, slot) + half;
  const char* a1 = base + wg_off(y, xb + 4, pw, slot) + half;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
  uint4 r;
  r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}

// f32 fragment element: pixel (row, x) channel c of the 32-channel slice.
__device__ __forceinline__ float frag_f32(const char* base, int py, int px, int pw, int c) {
  return *reinterpret_cast<const float*>(base + wg_off(py, px, pw, c >> 2) + (c & 3) * 4);
}

__device__ __forceinline__ f32x4_t mfma_bf16(uint4 a, uint4 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <typename T, int TAPS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradP a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool BF = sizeof(T) == 2;
  constexpr int KC = DD_LDS_ROW / (int)sizeof(T);  // channels per slice (64 bf16 / 32 f32)
  constexpr int NPW = BF ? 4 : 1;                  // n-tiles per wave
  constexpr bool HALO = (TAPS == 9);
  constexpr int PH = HALO ? DD_TILE + 2 : DD_TILE, PW = PH;
  char* ptile = smem;
  char* qtile = smem + PH * PW * DD_LDS_ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int ns = bid % a.nslices; bid /= a.nslices;
  const int ms = bid % a.mslices; bid /= a.mslices;
  const int ks = bid;
  const int mi = BF ? wave : (wave & 1);
  const int ni0 = BF ? 0 : (wave >> 1);
  const bool gather = (a.flags & DD_GATHER2X2) != 0;
  const bool in_relu = (a.flags & DD_IN_RELU) != 0;
  const T* __restrict__ P = reinterpret_cast<const T*>(a.p);
  const T* __restrict__ Q = reinterpret_cast<const T*>(a.q);
  const bool bias_q = a.bias_mode == 1 && ms == 0, bias_p = a.bias_mode == 2 && ns == 0;

#ifdef DD_PROFILE_PHASES
  const unsigned long long k_t0 = __builtin_readcyclecounter(), k_w0 = wall_clock64();
#endif
  TilePlan<PH> pplan;
  TilePlan<DD_TILE> qplan;
  tile_plan<PH>(pplan, tid);
  tile_plan<DD_TILE>(qplan, tid);
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f32x4_t acc[TAPS][NPW];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < NPW; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int per_img = a.tiles_y * a.tiles_x;
  const int total_tiles = a.B * per_img;
  uint4 qreg[TilePlan<DD_TILE>::ITERS];
  uint4 preg[HALO ? TilePlan<PH>::ITERS : 1];
  // global loads of one pixel tile (dy tile + haloed x tile) into registers; issued one tile AHEAD of its use, so the memory round
  // trip (2-3 us on the loaded machine) overlaps the previous tile's 288 MFMAs instead of sitting in front of them
  auto load_tile = [&](int tile) {
    const int b = tile / per_img;
    const int rem = tile - b * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * DD_TILE, x0 = tx * DD_TILE;
    tile_load<T, DD_TILE>(qreg, qplan, Q, (long)b * a.H * a.W, a.ldq, a.nv, ns * KC, y0, x0, a.H, a.W, 0, 1, 0, 0, a.H, a.W, tid);
    if constexpr (HALO)
      tile_load<T, PH>(preg, pplan, P, (long)b * a.hin * a.win, a.ldp, a.mv, ms * KC, y0 - 1, x0 - 1, a.H, a.W, 1, 1, 0, 0, a.hin, a.win, tid);
  };
  if (HALO && ks < total_tiles) load_tile(ks);
  for (int tile = ks; tile < total_tiles; tile += a.ksplit) {
    const int b = tile / per_img;
    const int rem = tile - b * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * DD_TILE, x0 = tx * DD_TILE;

    if constexpr (HALO) {
      WPHASE_T(w0);
      __syncthreads();  // previous tile fully consumed
      if (bias_q) bias_accumulate<T, DD_TILE>(bsum, qreg, qplan, false);
      if (bias_p) bias_accumulate<T, PH>(bsum, preg, pplan, true);
      tile_store<T, DD_TILE>(qtile, qreg, qplan, false);
      tile_store<T, PH>(ptile, preg, pplan, in_relu);
      __syncthreads();
      WPHASE_T(w1);
      if (tile + a.ksplit < total_tiles) load_tile(tile + a.ksplit);
      WPHASE_T(w2);
      WPHASE_ADD(0, w0, w1); WPHASE_ADD(1, w1, w2); WPHASE_ADD(5, 0ull, 1ull);
      if constexpr (BF) {
        // 72 steps (8 k-steps x 9 taps) of 4 MFMAs, fully unrolled.  The x fragment of step s+2 (two transpose reads) and, during
        // taps 0..3, one dy fragment of the next k-step are issued in front of the MFMAs of step s; sched_barrier(0) pins exactly
        // that order (sched_group_barrier only pins counts: the compiler then picks reads that are needed 4 MFMAs later).
        constexpr int NSTEP = 8 * TAPS;
        uint4 bq[2][NPW], ap[3];
        auto p_frag = [&](int step) {
          const int kst = step / TAPS, t = step - kst * TAPS;
          return frag_tr_bf16(ptile, 2 * kst + t / 3, PW, t % 3, mi, lane);
        };
#pragma unroll
        for (int j = 0; j < NPW; ++j) bq[0][j] = frag_tr_bf16(qtile, 0, DD_TILE, 0, ni0 + j, lane);
        ap[0] = p_frag(0);
        ap[1] = p_frag(1);
#pragma unroll
        for (int kst = 0; kst < 8; ++kst) {
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const int step = kst * TAPS + t;
            if (step + 2 < NSTEP) ap[(step + 2) % 3] = p_frag(step + 2);
            if (t < NPW && kst + 1 < 8) bq[(kst + 1) & 1][t] = frag_tr_bf16(qtile, 2 * (kst + 1), DD_TILE, 0, ni0 + t, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPW; ++j) acc[t][j] = mfma_bf16(ap[step % 3], bq[kst & 1][j], acc[t][j]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        WPHASE_T(w3);
        WPHASE_ADD(2, w2, w3);
      } else {
        const int kq = lane >> 4, li = lane & 15;
#pragma unroll 2
        for (int u = 0; u < 64; ++u) {  // 64 groups of 4 pixels
          const int row = u >> 2, x = (u & 3) * 4 + kq;
          const float bq = frag_f32(qtile, row, x, DD_TILE, ni0 * 16 + li);
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const float ap = frag_f32(ptile, row + t / 3, x + t % 3, PW, mi * 16 + li);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bq, acc[t][0], 0, 0, 0);
          }
        }
      }
    } else {
      tile_load<T, DD_TILE>(qreg, qplan, Q, (long)b * a.H * a.W, a.ldq, a.nv, ns * KC, y0, x0, a.H, a.W, 0, 1, 0, 0, a.H, a.W, tid);
      __syncthreads();  // previous tile fully consumed
      if (bias_q) bias_accumulate<T, DD_TILE>(bsum, qreg, qplan, false);
      tile_store<T, DD_TILE>(qtile, qreg, qplan, false);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        uint4 preg[TilePlan<PH>::ITERS];
        tile_load<T, PH>(preg, pplan, P, (long)b * a.hin * a.win, a.ldp, a.mv, ms * KC, y0, x0, a.H, a.W, 0, gather ? 2 : 1, gather ? (t >> 1) : 0,
                         gather ? (t & 1) : 0, a.hin, a.win, tid);
        if (t > 0) __syncthreads();
        if (bias_p) bias_accumulate<T, PH>(bsum, preg, pplan, false);
        tile_store<T, PH>(ptile, preg, pplan, in_relu);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll 2
          for (int kst = 0; kst < 8; ++kst) {
            const uint4 ap = frag_tr_bf16(ptile, 2 * kst, PW, 0, mi, lane);
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
              const uint4 bq = frag_tr_bf16(qtile, 2 * kst, DD_TILE, 0, ni0 + j, lane);
              acc[t][j] = mfma_bf16(ap, bq, acc[t][j]);
            }
          }
        } else {
          const int kq = lane >> 4, li = lane & 15;
#pragma unroll 4
          for (int u = 0; u < 64; ++u) {
            const int row = u >> 2, x = (u & 3) * 4 + kq;
            const float ap = frag_f32(ptile, row, x, PW, mi * 16 + li);
            const float bq = frag_f32(qtile, row, x, DD_TILE, ni0 * 16 + li);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bq, acc[t][0], 0, 0, 0);
          }
        }
      }
    }
  }

  WPHASE_T(e0);
  // D[m][n]: lane holds n = lane&15, m = (lane>>4)*4 + e
  const int li = lane & 15, q4 = (lane >> 4) * 4;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int n = ns * KC + (ni0 + j) * 16 + li;
      if (n >= a.n) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = ms * KC + mi * 16 + q4 + e;
#ifdef DD_EXP_WG_ATOMIC
        if (m < a.m) __hip_atomic_fetch_add(a.out + ((long)t * a.m + m) * a.n + n, acc[t][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        if (m < a.m) atomicAdd(a.out + ((long)t * a.m + m) * a.n + n, acc[t][j][e]);
#endif
      }
    }

  WPHASE_T(e1);
  WPHASE_ADD(3, e0, e1);
#ifdef DD_PROFILE_PHASES
  if (blockIdx.x == 0 && tid == 0) { dd_wphase_cycles[14] += __builtin_readcyclecounter() - k_t0; dd_wphase_cycles[15] += wall_clock64() - k_w0; }
#endif
  if (a.bias_mode != 0) {   // (uniform) reduce the 32 threads that share a 16-byte channel group, then one atomic per channel
    constexpr int PER16 = Elem<T>::PER16;
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < PER16; ++e) red[tid * PER16 + e] = bsum[e];
    __syncthreads();
    if ((bias_q || bias_p) && tid < KC) {
      const int slot = tid / PER16, e = tid - slot * PER16;
      float s = 0.f;
      for (int j = 0; j < 32; ++j) s += red[(j * 8 + slot) * PER16 + e];
      const int c = (bias_q ? ns : ms) * KC + tid;
      if (c < (bias_q ? a.n : a.m)) atomicAdd(a.bias_out + c, s);
    }
  }
}

template <typename T, int TAPS>
int launch(const WgradP& p, hipStream_t stream) {
  constexpr int PH = (TAPS == 9) ? DD_TILE + 2 : DD_TILE;
  const size_t lds = (size_t)PH * PH * DD_LDS_ROW + (size_t)DD_TILE * DD_TILE * DD_LDS_ROW;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  const long blocks = (long)p.ksplit * p.mslices * p.nslices;
  hipLaunchKernelGGL((wgrad_kernel<T, TAPS>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
int dispatch(const WgradP& p, hipStream_t stream) {
  switch (p.taps) {
    case 9: return launch<T, 9>(p, stream);
    case 4: return launch<T, 4>(p, stream);
    default: return launch<T, 1>(p, stream);
  }
}

}  // namespace

extern "C" int dd_conv_wgrad(const dd_wgrad_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->p && a->q && a->out, "dd_conv_wgrad: null pointer");
  DD_REQUIRE(a->dtype == DD_F32 || a->dtype == DD_BF16, "dd_conv_wgrad: bad dtype %d", a->dtype);
  const int esz = a->dtype == DD_F32 ? 4 : 2, per16 = 16 / esz, kc = DD_LDS_ROW / esz;
  const bool gather = (a->flags & DD_GATHER2X2) != 0;
  DD_REQUIRE(a->taps == 9 || a->taps == 1 || (a->taps == 4 && gather), "dd_conv_wgrad: taps=%d unsupported", a->taps);
  DD_REQUIRE(!gather || a->taps == 4, "dd_conv_wgrad: DD_GATHER2X2 needs taps=4");
  const int mv = (a->m + per16 - 1) / per16 * per16, nv = (a->n + per16 - 1) / per16 * per16;
  DD_REQUIRE(a->m > 0 && a->n > 0 && a->ldp % per16 == 0 && a->ldq % per16 == 0 && mv <= a->ldp && nv <= a->ldq,
             "dd_conv_wgrad: m=%d n=%d ldp=%d ldq=%d: ld must be a multiple of %d and cover the rounded channel count", a->m, a->n, a->ldp, a->ldq, per16);
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "dd_conv_wgrad: empty grid");
  DD_REQUIRE(((uintptr_t)a->p % 16) == 0 && ((uintptr_t)a->q % 16) == 0, "dd_conv_wgrad: pointers must be 16-byte aligned");
  DD_REQUIRE(a->bias_mode >= 0 && a->bias_mode <= 2 && (a->bias_mode == 0 || a->bias_out), "dd_conv_wgrad: bias_mode=%d needs bias_out", a->bias_mode);
  WgradP p;
  p.p = a->p; p.q = a->q; p.out = a->out; p.bias_out = a->bias_out; p.bias_mode = a->bias_mode;
  p.ldp = a->ldp; p.m = a->m; p.ldq = a->ldq; p.n = a->n; p.mv = mv; p.nv = nv;
  p.B = a->B; p.H = a->H; p.W = a->W; p.taps = a->taps; p.flags = a->flags;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.mslices = dd_ceil_div(a->m, kc); p.nslices = dd_ceil_div(a->n, kc);
  p.hin = gather ? 2 * a->H : a->H; p.win = gather ? 2 * a->W : a->W;
  const long total_tiles = (long)a->B * p.tiles_x * p.tiles_y;
  int ksplit = a->ksplit;
  if (ksplit <= 0) {
    static int target = 0;
    if (!target) { const char* e = getenv("DD_WGRAD_BLOCKS"); target = e ? atoi(e) : 256; }   // measured: 256 workgroups (1/CU) beat 128 / 512 / 1024 (atomics vs overlap)
    ksplit = (int)(target / ((long)p.mslices * p.nslices));
    if (ksplit < 1) ksplit = 1;
  }
  if (ksplit > total_tiles) ksplit = (int)total_tiles;
  p.ksplit = ksplit;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return a->dtype == DD_F32 ? dispatch<float>(p, s) : dispatch<bf16_t>(p, s);
}
