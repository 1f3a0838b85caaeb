// Weight-gradient GEMM on CDNA4 MFMA.
//
//   out[t][m][n] += sum_{b, pixel p}  in(P)[b, map_t(p), m] * Q[b, p, n]          (fp32 atomics into the gradient arena)
//
// conv2d            : P = layer input x (3x3: halo patch, 1x1: same pixel), Q = pre-activation output gradient,
//                     out = dKernel in TF HWIO layout [kh*kw][C_in][C_out].
// conv2d_transpose  : P = output gradient on the fine grid, map_t(p) = 2p+(a,b) (DD_GATHER2X2), Q = layer input,
//                     out = dKernel in TF layout [a*2+b][C_out][C_in].
// Replaces the TF-autodiff filter gradients behind tf.train.AdamOptimizer.minimize (reference
// TensorFlow/Training.py:701-702) for every conv in UNet.py / Tiramisu.py / Architecture.py:238-243 / MultiScalePrediction.py.
//
// Both operands are reduction-major in memory (NHWC: the pixel index is the GEMM K dimension), i.e. "transposed" with
// respect to what an MFMA fragment wants (8 consecutive k per lane).  bf16 path: the NHWC tiles stay as they are in
// LDS ([pixel][channel], 128-byte rows, slot-swizzled) and fragments are fetched with the gfx950 transpose read
// ds_read_b64_tr_b16 (a 16-lane group reads a [4 pixels][16 channels] block, each lane receives one channel's 4 pixels);
// a tap shift is a whole-row shift in that image, so the 3x3 halo patch is staged once and reused by the 9 taps.
// f32 path: exact-f32 MFMA 16x16x4, one element per lane straight from the same image (ds_read_b32).
// A workgroup owns one (m-slice, n-slice) of 128 B worth of channels each, ALL taps, and a strided subset of the
// 16x16 pixel tiles (split-K); partial results are added with one fp32 atomic per element at the end.
#include "dd_common.h"

namespace {

struct WgradP {
  const void* p; const void* q; float* out;
  int ldp, m, ldq, n, mv, nv;   // mv/nv: staged (16-byte rounded) channel counts
  int B, H, W, taps, flags;
  int tiles_x, tiles_y, ksplit, mslices, nslices;
  int hin, win;
};

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// bf16 fragment (8 k-values of channel `lane&15` of 16-channel tile `ctile`) for k-step rows via two transpose reads.
__device__ __forceinline__ uint4 frag_tr_bf16(const char* base, int row, int pw, int dx, int ctile, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int y = row + (g >> 1), xb = (g & 1) * 8 + (t16 >> 2) + dx;
  const int sub = t16 & 3;
  const int slot = ctile * 2 + (sub >> 1), half = (sub & 1) * 8;
  const int pix0 = y * pw + xb, pix1 = pix0 + 4;
  const char* a0 = base + lds_off(pix0, slot) + half;
  const char* a1 = base + lds_off(pix1, slot) + half;
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
__device__ __forceinline__ float frag_f32(const char* base, int pix, int c) {
  return *reinterpret_cast<const float*>(base + lds_off(pix, c >> 2) + (c & 3) * 4);
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

  f32x4_t acc[TAPS][NPW];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int j = 0; j < NPW; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int per_img = a.tiles_y * a.tiles_x;
  const int total_tiles = a.B * per_img;
  for (int tile = ks; tile < total_tiles; tile += a.ksplit) {
    const int b = tile / per_img;
    const int rem = tile - b * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * DD_TILE, x0 = tx * DD_TILE;

    TileGeom gq;
    gq.ph = DD_TILE; gq.pw = DD_TILE; gq.oy = y0; gq.ox = x0; gq.sy = 1; gq.ay = 0; gq.ax = 0;
    gq.lim_y = a.H; gq.lim_x = a.W; gq.min_y = 0; gq.min_x = 0; gq.hin = a.H; gq.win = a.W;
    TileGeom gp;
    gp.ph = PH; gp.pw = PW; gp.oy = HALO ? y0 - 1 : y0; gp.ox = HALO ? x0 - 1 : x0;
    gp.sy = gather ? 2 : 1; gp.ay = 0; gp.ax = 0;
    gp.lim_y = HALO ? a.H + 1 : a.H; gp.lim_x = HALO ? a.W + 1 : a.W;
    gp.min_y = HALO ? -1 : 0; gp.min_x = HALO ? -1 : 0; gp.hin = a.hin; gp.win = a.win;

    __syncthreads();  // previous tile fully consumed
    stage_pixels<T>(qtile, Q, (long)b * a.H * a.W, a.ldq, a.nv, ns * KC, 8, gq, false, tid, 256);
    if (HALO) stage_pixels<T>(ptile, P, (long)b * a.hin * a.win, a.ldp, a.mv, ms * KC, 8, gp, in_relu, tid, 256);

    if constexpr (HALO) {
      __syncthreads();
      // k-step outer, taps inner: the Q fragments are fetched once per k-step and reused by all 9 taps
      if constexpr (BF) {
        for (int kst = 0; kst < 8; ++kst) {
          uint4 bq[NPW];
#pragma unroll
          for (int j = 0; j < NPW; ++j) bq[j] = frag_tr_bf16(qtile, 2 * kst, DD_TILE, 0, ni0 + j, lane);
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const uint4 ap = frag_tr_bf16(ptile, 2 * kst + t / 3, PW, t % 3, mi, lane);
#pragma unroll
            for (int j = 0; j < NPW; ++j)
              acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ap), __builtin_bit_cast(bf16x8_t, bq[j]), acc[t][j], 0, 0, 0);
          }
        }
      } else {
        const int kq = lane >> 4, li = lane & 15;
#pragma unroll 2
        for (int u = 0; u < 64; ++u) {  // 64 groups of 4 pixels
          const int row = u >> 2, x = (u & 3) * 4 + kq;
          const float bq = frag_f32(qtile, row * DD_TILE + x, ni0 * 16 + li);
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const float ap = frag_f32(ptile, (row + t / 3) * PW + x + t % 3, mi * 16 + li);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bq, acc[t][0], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        if (t > 0) __syncthreads();
        if (gather) { gp.ay = t >> 1; gp.ax = t & 1; }
        stage_pixels<T>(ptile, P, (long)b * a.hin * a.win, a.ldp, a.mv, ms * KC, 8, gp, in_relu, tid, 256);
        __syncthreads();
        if constexpr (BF) {
#pragma unroll 2
          for (int kst = 0; kst < 8; ++kst) {
            const uint4 ap = frag_tr_bf16(ptile, 2 * kst, PW, 0, mi, lane);
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
              const uint4 bq = frag_tr_bf16(qtile, 2 * kst, DD_TILE, 0, ni0 + j, lane);
              acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ap), __builtin_bit_cast(bf16x8_t, bq), acc[t][j], 0, 0, 0);
            }
          }
        } else {
          const int kq = lane >> 4, li = lane & 15;
#pragma unroll 4
          for (int u = 0; u < 64; ++u) {
            const int row = u >> 2, x = (u & 3) * 4 + kq;
            const float ap = frag_f32(ptile, row * PW + x, mi * 16 + li);
            const float bq = frag_f32(qtile, row * DD_TILE + x, ni0 * 16 + li);
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap, bq, acc[t][0], 0, 0, 0);
          }
        }
      }
    }
  }

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
        if (m < a.m) atomicAdd(a.out + ((long)t * a.m + m) * a.n + n, acc[t][j][e]);
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
  WgradP p;
  p.p = a->p; p.q = a->q; p.out = a->out; p.ldp = a->ldp; p.m = a->m; p.ldq = a->ldq; p.n = a->n; p.mv = mv; p.nv = nv;
  p.B = a->B; p.H = a->H; p.W = a->W; p.taps = a->taps; p.flags = a->flags;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.mslices = dd_ceil_div(a->m, kc); p.nslices = dd_ceil_div(a->n, kc);
  p.hin = gather ? 2 * a->H : a->H; p.win = gather ? 2 * a->W : a->W;
  const long total_tiles = (long)a->B * p.tiles_x * p.tiles_y;
  int ksplit = a->ksplit;
  if (ksplit <= 0) {
    ksplit = (int)(512 / ((long)p.mslices * p.nslices));
    if (ksplit < 1) ksplit = 1;
  }
  if (ksplit > total_tiles) ksplit = (int)total_tiles;
  p.ksplit = ksplit;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return a->dtype == DD_F32 ? dispatch<float>(p, s) : dispatch<bf16_t>(p, s);
}
