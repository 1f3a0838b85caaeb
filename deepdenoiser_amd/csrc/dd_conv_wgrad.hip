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
// The bias gradient costs no extra pass over the gradient tensor: it is summed from data the kernel holds anyway.
// All global loads are unconditional (invalid vectors read a zero page): a branch per vector serialises them.
// Two kernels: wgrad_dma_kernel (bf16 3x3 and 1x1 layers: LDS-DMA, double-buffered tiles, 8 waves -- see its header) and wgrad_kernel (f32,
// the 2x2 gather of the transpose convolutions, bias_mode 2, very wide layers: tiles staged through registers, next tile prefetched).
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

#ifndef WG_DMA_SPAN
#define WG_DMA_SPAN 8      // eighths of a tile's steps over which the next tile's DMA pieces are issued
#endif
namespace {

struct WgradP {
  const void* p; const void* q; float* out; float* bias_out;
  int ldp, m, ldq, n, mv, nv;   // mv/nv: staged (16-byte rounded) channel counts
  int B, H, W, taps, flags, bias_mode;
  int tiles_x, tiles_y, ksplit, mslices, nslices;
  int ncombo, cstart[65];       // LDS-DMA kernel: workgroups [cstart[c], cstart[c+1]) split the pixel tiles of (ms, ns) = (c / nslices, c % nslices)
  int hin, win;
  // stacked form (dd_wgrad_args): column block n / stack_width of the product belongs to conv j = that block, rows below stack_m0 + j * stack_width
  int stack_blocks, stack_width, stack_m0;
  float* stack_out[8]; float* stack_bias[8];
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
// one 16-byte vector of a stride-1 tile (no gather): the pieces of the NEXT tile are issued one at a time between the MFMA groups of
// the current one, so the memory pipeline never backs up behind a burst of 19 loads per thread (a burst blocks the wave at issue
// until the queue drains: ~4.5k cycles per tile, tools/phase_profile.py ... wgrad)
template <typename T, int PH>
__device__ __forceinline__ uint4 tile_load_one(int it, const TilePlan<PH>& pl, const T* __restrict__ base, bool live, int ld, int oy, int ox, int H, int W,
                                               int halo, int win) {
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
  const int yx = pl.yx[it];
  const int ly = oy + (yx >> 8), lx = ox + (yx & 255);
  const bool ok = live && yx >= 0 && ly >= -halo && lx >= -halo && ly < H + halo && lx < W + halo && ly >= 0 && lx >= 0 && ly < H && lx < W;
  return *reinterpret_cast<const uint4*>(ok ? base + ((long)ly * win + lx) * ld : zero);
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
    if constexpr (sizeof(T) == 2) {
      float v[8];
      unpack8t<T>(reg[it], v);
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
      const int ntile = tile + a.ksplit;
      if (!BF && ntile < total_tiles) load_tile(ntile);
      WPHASE_T(w2);
      WPHASE_ADD(0, w0, w1); WPHASE_ADD(1, w1, w2); WPHASE_ADD(5, 0ull, 1ull);
      if constexpr (BF) {
        // 72 steps (8 k-steps x 9 taps) of 4 MFMAs, fully unrolled.  The x fragment of step s+2 (two transpose reads) and, during
        // taps 0..3, one dy fragment of the next k-step are issued in front of the MFMAs of step s; sched_barrier(0) pins exactly
        // that order (sched_group_barrier only pins counts: the compiler then picks reads that are needed 4 MFMAs later).
        constexpr int NSTEP = 8 * TAPS;
        constexpr int QI = TilePlan<DD_TILE>::ITERS, PI = TilePlan<PH>::ITERS, EVERY = NSTEP / (QI + PI);
        // next tile's origin and per-thread base pointers (channel group tid&7 of slice ns / ms)
        const bool nlive = ntile < total_tiles;
        const int nt = nlive ? ntile : tile;
        const int nb = nt / per_img, nrem = nt - nb * per_img;
        const int nty = nrem / a.tiles_x, ny0 = nty * DD_TILE, nx0 = (nrem - nty * a.tiles_x) * DD_TILE;
        const int qch = ns * KC + (tid & 7) * Elem<T>::PER16, pch = ms * KC + (tid & 7) * Elem<T>::PER16;
        const T* qbase = Q + (long)nb * a.H * a.W * a.ldq + qch;
        const T* pbase = P + (long)nb * a.hin * a.win * a.ldp + pch;
        const bool qlive = nlive && qch < a.nv, plive = nlive && pch < a.mv;
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
            if (step % EVERY == 0 && step / EVERY < QI + PI) {      // one global load of the next tile
              constexpr int dummy = 0; (void)dummy;
              const int k = step / EVERY;
              if (k < QI) qreg[k] = tile_load_one<T, DD_TILE>(k, qplan, qbase, qlive, a.ldq, ny0, nx0, a.H, a.W, 0, a.W);
              else preg[k - QI] = tile_load_one<T, PH>(k - QI, pplan, pbase, plive, a.ldp, ny0 - 1, nx0 - 1, a.H, a.W, 1, a.win);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NPW; ++j) acc[t][j] = mma16<T>(ap[step % 3], bq[kst & 1][j], acc[t][j]);
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
              acc[t][j] = mma16<T>(ap, bq, acc[t][j]);
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
  dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
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
  dd_det_end();
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 3x3 weight gradient, LDS-DMA variant (the hot one).  The pixel tiles go global -> LDS with global_load_lds_dwordx4 (1 KiB per
// wave-instruction, no staging registers, no ds_write phase) into a DOUBLE-buffered LDS image; the 19 DMA pieces of the next tile
// are issued one at a time between the MFMA groups of the current tile.  What this replaces, per 256-pixel tile (tools/phase_profile.py
// ... wgrad, 64->64): 4.6k cycles blocked issuing 19 register loads + 3.3k cycles of ds_write staging in front of 6.6k cycles of MFMA,
// and 76 staging VGPRs that pushed the 144 accumulator registers into AGPR shuffling.
// An LDS-DMA chunk is 64 lanes x 16 B contiguous in LDS = 8 pixel rows of 128 B.  Lane l owns pixel (l >> 3) of the chunk and PHYSICAL
// slot l & 7; the image is slot-swizzled by the pixel index (wg_off), and the chunk starts at a multiple of 8 pixels, so the lane always
// fetches LOGICAL slot (l & 7) ^ (l >> 3): the swizzle is applied on the global side and the DMA stays a linear 1 KiB copy.
// The DMA is issued through inline asm: with the builtin, hipcc puts s_waitcnt vmcnt(0) in front of every later ds_read (it cannot
// tell the LDS regions apart), which would serialise exactly what this kernel overlaps.  Completion is awaited explicitly (vmcnt(0)
// + barrier) once per tile.
__device__ __forceinline__ void dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}

// LDS image of one tile: the x tile (18x18 haloed for 3x3 layers: 324 pixels -> 41 chunks of 1 KiB; 16x16 for 1x1 layers: 32 chunks)
// followed by the 16x16 dy tile (32 chunks); two such images (double buffer).
template <bool HALO> struct WgLds {
  static constexpr int PW = HALO ? DD_TILE + 2 : DD_TILE;
  static constexpr int PCH = HALO ? 41 : 32;
  static constexpr int P_BYTES = PCH * 1024, Q_BYTES = DD_TILE * DD_TILE * DD_LDS_ROW, BUF = P_BYTES + Q_BYTES;
};

// 8 waves, 2 per SIMD (with ONE wave per SIMD every s_waitcnt and every address instruction delays the next MFMA -- in-order issue:
// the same loop measured 9.8k cycles per tile against 4.6k of pure MFMA time; two waves per SIMD fill each other's gaps).
//   MODE 0 (3x3):         wave = input-channel tile (w & 3) x output-channel half (w >> 2), all 9 taps: 72 accumulator registers.
//   MODE 1 (3x3, <= 32 x 32 channels, e.g. the 24-channel compose net): only 2 x 1 channel-tile pairs exist, so the waves split the
//                         TAPS instead: wave = input tile (w & 1) x tap group (w >> 1) = taps {0,1,2} {3,4} {5,6} {7,8}.  Every gradient
//                         element still has exactly one owner (no extra atomics), each wave runs 24 steps per tile instead of 72.
//   MODE 2 (1x1):         as MODE 0 with one tap and no halo; the loop is 8 steps, fully unrolled.
template <typename T, bool IN_RELU, int MODE>
__global__ __launch_bounds__(512) void wgrad_dma_kernel(const WgradP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  static_assert(sizeof(T) == 2, "LDS-DMA weight gradient: bf16 / fp16 storage");
  constexpr bool HALO = MODE != 2;
  using LD = WgLds<HALO>;
  constexpr int KC = 64, NPW = 2, PW = LD::PW, WG_BUF = LD::BUF, WG_P_BYTES = LD::P_BYTES;
  constexpr int TW = MODE == 0 ? 9 : MODE == 1 ? 3 : 1;      // taps per k-step in this wave's loop
  constexpr int QPC = 4, PPC = HALO ? 6 : 4, NP = QPC + PPC; // DMA pieces per wave and tile
  constexpr int HSTEP = 2 * TW, NSTEP = 4 * HSTEP;           // steps per loop iteration (2 k-steps) / per tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (channel-slice pair, split index): slices with fewer valid channel tiles get fewer workgroups (they finish their tiles faster)
  int combo = 0;
  while (combo + 1 < a.ncombo && (int)blockIdx.x >= a.cstart[combo + 1]) ++combo;
  const int ks = blockIdx.x - a.cstart[combo], ksplit = a.cstart[combo + 1] - a.cstart[combo];
  const int ms = combo / a.nslices, ns = combo - ms * a.nslices;
  const int mi = MODE == 1 ? wave & 1 : wave & 3, nj = MODE == 1 ? 0 : (wave >> 2) * NPW;
  const int tbase = MODE == 1 ? ((wave >> 1) == 0 ? 0 : 1 + 2 * (wave >> 1)) : 0;      // MODE 1: first tap of this wave's group
  const int ntw = MODE == 1 ? ((wave >> 1) == 0 ? 3 : 2) : TW;                         //         and how many it owns
  const bool active = mi * 16 < a.m - ms * KC && nj * 16 < a.n - ns * KC;   // this wave's channel tiles exist
  const T* __restrict__ P = reinterpret_cast<const T*>(a.p);
  const T* __restrict__ Q = reinterpret_cast<const T*>(a.q);
  const bool bias_wave = a.bias_mode == 1 && ms == 0 && (MODE == 1 ? wave == 0 : mi == 0);   // sums the dy fragments it reads anyway
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#ifdef DD_PROFILE_PHASES
  const unsigned long long k_t0 = __builtin_readcyclecounter(), k_w0 = wall_clock64();
#endif

  // this lane's part of every chunk: pixel r of the chunk, logical channel slot ls
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const int qch = ns * KC + ls * 8, pch = ms * KC + ls * 8;
  const bool q_ok = qch < a.nv, p_ok = pch < a.mv;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const int per_img = a.tiles_y * a.tiles_x;
  const int total_tiles = a.B * per_img;
  // per-lane, tile-invariant part of the DMA source addresses (byte offsets from the tile's origin pixel)
  int q_off[2];                 // 16x16 tiles: chunk c = wave*4 + k is tile row c >> 1, pixels (c & 1)*8 + r
#pragma unroll
  for (int h = 0; h < 2; ++h) q_off[h] = ((h * 8 + r) * a.ldq + qch) * 2;
  int p_off[PPC], p_yx[PPC];    // haloed x tile: chunk c = k*8 + wave is pixels c*8 + r of the 18x18 tile; 1x1: as the dy tile
#pragma unroll
  for (int k = 0; k < PPC; ++k) {
    if (HALO) {
      const int pix = (k * 8 + wave) * 8 + r;
      const int py = (pix * 3641) >> 16, px = pix - py * PW;      // pix / 18 for pix < 400
      p_off[k] = ((py * a.win + px) * a.ldp + pch) * 2;
      p_yx[k] = pix < PW * PW ? ((py << 8) | px) : (0x7fff << 8);     // dummy pixels of the last chunk: always out of range (255 was not, for H >= 256: they then read row 17 of the tile into LDS slots nothing uses)
    } else {
      const int c = wave * PPC + k, py = c >> 1, px = (c & 1) * 8 + r;
      p_off[k] = ((py * a.win + px) * a.ldp + pch) * 2;
      p_yx[k] = (py << 8) | px;
    }
  }

  struct Origin { const char* q; const char* p; int y0, x0; bool live; };
  constexpr int HO = HALO ? 1 : 0;
  auto origin = [&](int tile) {
    Origin o;
    o.live = tile < total_tiles;
    const int t = o.live ? tile : 0;
    const int b = t / per_img, rem = t - b * per_img;
    const int ty = rem / a.tiles_x;
    o.y0 = ty * DD_TILE; o.x0 = (rem - ty * a.tiles_x) * DD_TILE;
    o.q = reinterpret_cast<const char*>(Q + ((long)b * a.H * a.W + (long)o.y0 * a.W + o.x0) * a.ldq);
    o.p = reinterpret_cast<const char*>(P + ((long)b * a.hin * a.win + (long)(o.y0 - HO) * a.win + (o.x0 - HO)) * a.ldp);
    return o;
  };
  // DMA piece k of the tile at `o` into buffer `sel`: k < QPC -> dy chunk, else x chunk
  auto piece = [&](int k, const Origin& o, int sel) {
    const unsigned buf = lds_base + sel * WG_BUF;
    if (k < QPC) {
      const int c = wave * QPC + k, row = c >> 1, h = c & 1;
      const bool ok = o.live && q_ok && o.y0 + row < a.H && o.x0 + h * 8 + r < a.W;
      dma_1k(ok ? o.q + (long)row * a.W * a.ldq * 2 + q_off[h] : zero, buf + WG_P_BYTES + c * 1024);
    } else {
      const int kk = k - QPC, c = HALO ? kk * 8 + wave : wave * PPC + kk;
      if (c < LD::PCH) {      // wave-uniform
        const int gy = o.y0 - HO + (p_yx[kk] >> 8), gx = o.x0 - HO + (p_yx[kk] & 255);
        const bool ok = o.live && p_ok && (unsigned)gy < (unsigned)a.hin && (unsigned)gx < (unsigned)a.win;
        dma_1k(ok ? o.p + p_off[kk] : zero, buf + c * 1024);
      }
    }
  };

  f32x4_t acc[TW][NPW];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int j = 0; j < NPW; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // lane-dependent part of every fragment address (see the tile loop): pixel offset of this lane within a transpose read, its slot, its half
  int pbase[8], qbase[NPW][2];
  {
    const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3;
    const int yl = gq >> 1, xl = (gq & 1) * 8 + (t16 >> 2), halfb = (sub & 1) * 8;
    const int pl = yl * PW + xl, ql = yl * DD_TILE + xl;
#pragma unroll
    for (int c = 0; c < 8; ++c) pbase[c] = pl * DD_LDS_ROW + (((mi * 2 + (sub >> 1)) ^ ((pl + c) & 7)) << 4) + halfb;
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) qbase[j][h] = ql * DD_LDS_ROW + ((((nj + j) * 2 + (sub >> 1)) ^ ((ql + 4 * h) & 7)) << 4) + halfb;
  }
  float bsum[NPW] = {0.f, 0.f};     // dy column sums of channels (nj + jj)*16 + lane&15 over this lane's pixels (bias_wave only)

  {
    const Origin o0 = origin(ks);
#pragma unroll
    for (int k = 0; k < NP; ++k) piece(k, o0, 0);
  }
  int sel = 0;
  for (int tile = ks; tile < total_tiles; tile += ksplit, sel ^= 1) {
    WPHASE_T(w0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of `tile` have landed
    __syncthreads();                      // ... and everyone's; buffer sel^1 is free (all waves finished the previous tile)
    WPHASE_T(w1);
    const Origin on = origin(tile + ksplit);
    if (!active) {      // nothing to multiply: keep feeding the pipeline
#pragma unroll
      for (int k = 0; k < NP; ++k) piece(k, on, sel ^ 1);
      continue;
    }
    const char* ptile = smem + sel * WG_BUF;
    const char* qtile = ptile + WG_P_BYTES;
    // 4 loop iterations x (2 k-steps x TW taps, unrolled).  Four tile rows further down the swizzle key repeats ((y*18 + x) & 7 and
    // (y*16 + x) & 7 are periodic in y with period 4 / 1), so the fragment addresses of iteration kp are those of iteration 0 plus kp * 4
    // rows: a few dozen address registers instead of one per step.  (1x1: 8 steps in all, fully unrolled.)
    constexpr int PADV = 4 * PW * DD_LDS_ROW, QADV = 4 * DD_TILE * DD_LDS_ROW;
    constexpr int RING = MODE == 0 ? 6 : 3, AHEAD = RING - 1;      // x fragments are read AHEAD steps before use (HSTEP % RING == 0)
    uint4 bq[2][NPW], ap[RING];
    // Fragment addresses.  A lane's pixel for a read at tile position C (= row*PW + dx, a compile-time constant) is pl + C, so its byte
    // address is  tile + [pl*128 + ((slot ^ ((pl + (C & 7)) & 7)) << 4) + half]  +  C*128 : eight lane-dependent bases (C & 7) plus an
    // immediate.  Left to itself hipcc materialises one address register per read (~100 VGPRs), which is what capped the read-ahead.
    const char* pb[8];
    const char* qb[NPW][2];
    auto set_bases = [&](const char* pt, const char* qt) {
#pragma unroll
      for (int c = 0; c < 8; ++c) pb[c] = pt + pbase[c];
#pragma unroll
      for (int j = 0; j < NPW; ++j) { qb[j][0] = qt + qbase[j][0]; qb[j][1] = qt + qbase[j][1]; }
    };
    auto tr_pair = [&](const char* a0, const char* a1) {
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
      uint4 v;
      v.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      v.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      v.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      v.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      return v;
    };
    auto p_frag = [&](const char* pt, int hs) {      // hs = step within the iteration (may run AHEAD steps into the next one)
      const int k2 = hs / TW, i = hs - k2 * TW;
      uint4 v;
      if constexpr (MODE == 1) {                     // runtime tap: generic address arithmetic
        const int t = min(tbase + i, 8);
        const int dy = (t * 11) >> 5, dx = t - 3 * dy;                       // t / 3, t % 3 for t < 9
        v = frag_tr_bf16(pt, 2 * k2 + dy, PW, dx, mi, lane);
      } else {
        const int dy = HALO ? i / 3 : 0, dx = HALO ? i % 3 : 0;
        const int c0 = (2 * k2 + dy) * PW + dx;
        v = tr_pair(pb[c0 & 7] + c0 * DD_LDS_ROW, pb[(c0 + 4) & 7] + (c0 + 4) * DD_LDS_ROW);
      }
      if (IN_RELU) v = relu16<T>(v);
      return v;
    };
    auto q_frag = [&](const char* qt, int row, int j) {
      if constexpr (MODE == 1) return frag_tr_bf16(qt, row, DD_TILE, 0, nj + j, lane);
      else return tr_pair(qb[j][0] + row * DD_TILE * DD_LDS_ROW, qb[j][1] + (row * DD_TILE + 4) * DD_LDS_ROW);
    };
    auto body = [&](int kp, int hs, const char* pt, const char* qt) {
      const int k2 = hs / TW, i = hs - k2 * TW;
      const int g = MODE == 2 ? kp * HSTEP + hs : hs;               // ring position (a compile-time constant at every call)
      ap[(g + AHEAD) % RING] = p_frag(pt, hs + AHEAD);               // the last AHEAD of the tile read past it: never used
      if (TW >= NPW) {
        if (i < NPW) bq[(k2 + 1) & 1][i] = q_frag(qt, 2 * (k2 + 1), i);
      } else {
#pragma unroll
        for (int j = 0; j < NPW; ++j) bq[(k2 + 1) & 1][j] = q_frag(qt, 2 * (k2 + 1), j);
      }
#ifndef DD_EXP_NO_PIECE
      // the NP DMA pieces of the next tile, spread evenly over the NSTEP steps (indices are compile-time constants at every call: a
      // runtime index would move p_off / p_yx to scratch)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (kp == c) {
#pragma unroll
          for (int k = 0; k < NP; ++k)
            if ((k * (NSTEP * WG_DMA_SPAN / 8 > 0 ? NSTEP * WG_DMA_SPAN / 8 : 1)) / NP == c * HSTEP + hs) piece(k, on, sel ^ 1);
        }
#endif
      if (i == (MODE == 0 ? 4 : 0) && bias_wave) {      // bias gradient from the dy fragments in registers
#pragma unroll
        for (int jj = 0; jj < NPW; ++jj) {      // bq[k2 & 1][jj]: 8 pixels of channel lane&15 of n-tile nj + jj
          float f[8];
          unpack8t<T>(bq[k2 & 1][jj], f);
          bsum[jj] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NPW; ++j) acc[i][j] = mma16<T>(ap[g % RING], bq[k2 & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    };
    set_bases(ptile, qtile);
#pragma unroll
    for (int j = 0; j < NPW; ++j) bq[0][j] = q_frag(qtile, 0, j);
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) ap[i] = p_frag(ptile, i);
    if constexpr (MODE == 2) {
#pragma unroll
      for (int kp = 0; kp < 4; ++kp) {
        set_bases(ptile + kp * PADV, qtile + kp * QADV);
#pragma unroll
        for (int hs = 0; hs < HSTEP; ++hs) body(kp, hs, ptile + kp * PADV, qtile + kp * QADV);
      }
    } else {
#pragma unroll 1
      for (int kp = 0; kp < 4; ++kp) {
        const char* pt = ptile + kp * PADV;
        const char* qt = qtile + kp * QADV;
        set_bases(pt, qt);
#pragma unroll
        for (int hs = 0; hs < HSTEP; ++hs) body(kp, hs, pt, qt);
      }
    }
    WPHASE_T(w3);
    WPHASE_ADD(0, w0, w1); WPHASE_ADD(2, w1, w3); WPHASE_ADD(5, 0ull, 1ull);
  }

  WPHASE_T(e0);
  dd_det_wait();
  const int li = lane & 15, q4 = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    if (MODE == 1 && i >= ntw) continue;       // the padding step of a 2-tap group
    const int t = tbase + i;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int n = ns * KC + (nj + j) * 16 + li;
      if (n >= a.n) continue;
      // stacked form: this column belongs to conv jb of the dense block; only the rows of ITS input prefix are its gradient
      const int jb = a.stack_blocks ? n / a.stack_width : 0;
      const int mlim = a.stack_blocks ? a.stack_m0 + jb * a.stack_width : a.m;
      const int ncols = a.stack_blocks ? a.stack_width : a.n, nn = a.stack_blocks ? n - jb * a.stack_width : n;
      float* outp = a.stack_blocks ? a.stack_out[jb] : a.out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = ms * KC + mi * 16 + q4 + e;
        if (m < mlim) atomicAdd(outp + ((long)t * mlim + m) * ncols + nn, acc[i][j][e]);
      }
    }
  }
  if (a.bias_mode == 1) {
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) {
      float b = bsum[jj];
      b += __shfl_xor(b, 16);
      b += __shfl_xor(b, 32);
      const int c = ns * KC + (nj + jj) * 16 + li;
      if (bias_wave && lane < 16 && c < a.n) {
        if (a.stack_blocks) atomicAdd(a.stack_bias[c / a.stack_width] + c % a.stack_width, b);
        else atomicAdd(a.bias_out + c, b);
      }
    }
  }
  WPHASE_T(e1);
  WPHASE_ADD(3, e0, e1);
#ifdef DD_PROFILE_PHASES
  if (blockIdx.x == 0 && tid == 0) { dd_wphase_cycles[14] += __builtin_readcyclecounter() - k_t0; dd_wphase_cycles[15] += wall_clock64() - k_w0; }
#endif
  dd_det_end();
}

template <typename T, bool IN_RELU, int MODE>
static void launch_dma_mode(const WgradP& p, long blocks, hipStream_t stream) {
  const size_t lds = 2 * (size_t)WgLds<MODE != 2>::BUF;
  dd_det_sync();
  dd_allow_max_lds(reinterpret_cast<const void*>(wgrad_dma_kernel<T, IN_RELU, MODE>));
  hipLaunchKernelGGL((wgrad_dma_kernel<T, IN_RELU, MODE>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
}

template <typename T>
static int launch_dma(WgradP& p, hipStream_t stream) {
  // Work split.  A workgroup's time per tile is set by its busiest SIMD = the number of n-halves (2 x 16 output channels) its slice has,
  // plus a DMA/barrier floor: weight 3 for a full slice, 2 for a half one.  `target` workgroups in total (1 per CU).
  const int nc = p.mslices * p.nslices;
  const long total_tiles = (long)p.B * p.tiles_x * p.tiles_y;
  const int target = p.ksplit * nc;
  if (nc > 64) { dd_set_error("dd_conv_wgrad: more than 64 channel-slice pairs"); return DD_ERR_INVALID; }
  int w[64], wsum = 0;
  for (int c = 0; c < nc; ++c) {
    const int ns = c % p.nslices;
    const int nvalid = p.n - ns * 64 < 64 ? p.n - ns * 64 : 64;
    w[c] = nvalid > 32 ? 3 : 2;
    wsum += w[c];
  }
  p.ncombo = nc;
  p.cstart[0] = 0;
  for (int c = 0; c < nc; ++c) {
    long k = (long)target * w[c] / wsum;
    if (k < 1) k = 1;
    if (k > total_tiles) k = total_tiles;
    p.cstart[c + 1] = p.cstart[c] + (int)k;
  }
  const long blocks = p.cstart[p.ncombo];
  const bool relu = (p.flags & DD_IN_RELU) != 0;
  const int mode = p.taps == 1 ? 2 : (p.m <= 32 && p.n <= 32) ? 1 : 0;
  if (mode == 2) { if (relu) launch_dma_mode<T, true, 2>(p, blocks, stream); else launch_dma_mode<T, false, 2>(p, blocks, stream); }
  else if (mode == 1) { if (relu) launch_dma_mode<T, true, 1>(p, blocks, stream); else launch_dma_mode<T, false, 1>(p, blocks, stream); }
  else { if (relu) launch_dma_mode<T, true, 0>(p, blocks, stream); else launch_dma_mode<T, false, 0>(p, blocks, stream); }
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T, int TAPS>
int launch(const WgradP& p, hipStream_t stream) {
  constexpr int PH = (TAPS == 9) ? DD_TILE + 2 : DD_TILE;
  const size_t lds = (size_t)PH * PH * DD_LDS_ROW + (size_t)DD_TILE * DD_TILE * DD_LDS_ROW;
  dd_allow_max_lds(reinterpret_cast<const void*>(wgrad_kernel<T, TAPS>), 96 * 1024);
  const long blocks = (long)p.ksplit * p.mslices * p.nslices;
  dd_det_sync();
  hipLaunchKernelGGL((wgrad_kernel<T, TAPS>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

static bool dma_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DD_WGRAD_DMA"); v = e ? atoi(e) : 1; }
  return v != 0;
}

template <typename T>
int dispatch(const WgradP& p, hipStream_t stream) {
  // (more than 64 channel-slice pairs -- > ~512 x 512 channels -- do not fit the kernel's split table: those launches take the register-staged
  //  kernel below; round 3 raised the table from 32: the 1 088 -> 96 dense-block layer of the heavy Tiramisu is 17 x 2 pairs)
  if (sizeof(T) == 2 && (p.taps == 9 || p.taps == 1) && p.bias_mode != 2 && !(p.flags & DD_GATHER2X2) && p.mslices * p.nslices <= 64 && dma_enabled()) {
    if constexpr (sizeof(T) == 2) {
      WgradP q = p;
      return launch_dma<T>(q, stream);
    }
  }
  if (p.stack_blocks) { dd_set_error("dd_conv_wgrad: the stacked form runs on the LDS-DMA kernel only (2-byte storage, <= 64 channel-slice pairs, DD_WGRAD_DMA on)"); return DD_ERR_INVALID; }
  switch (p.taps) {
    case 9: return launch<T, 9>(p, stream);
    case 4: return launch<T, 4>(p, stream);
    default: return launch<T, 1>(p, stream);
  }
}

}  // namespace

bool dd_wgrad_pw_eligible(const dd_wgrad_args* a);
int dd_wgrad_pw_launch(const dd_wgrad_args* a, hipStream_t stream);

extern "C" int dd_conv_wgrad(const dd_wgrad_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->p && a->q && (a->out || a->stack_blocks > 0), "dd_conv_wgrad: null pointer");
  if (a->stack_blocks != 0) {
    DD_REQUIRE(a->stack_blocks > 0 && a->stack_blocks <= 8 && a->stack_width > 0 && a->stack_m0 > 0 && a->taps == 9 && a->dtype != DD_F32 && !(a->flags & DD_GATHER2X2)
               && a->n == a->stack_blocks * a->stack_width && a->m == a->stack_m0 + (a->stack_blocks - 1) * a->stack_width && a->bias_mode != 2,
               "dd_conv_wgrad: stacked form: taps = 9, 2-byte storage, n = stack_blocks * stack_width, m = stack_m0 + (stack_blocks - 1) * stack_width");
    for (int j = 0; j < a->stack_blocks; ++j)
      DD_REQUIRE(a->stack_out[j] && (a->bias_mode == 0 || a->stack_bias[j]), "dd_conv_wgrad: stacked form: block %d has no output pointer", j);
  }
  DD_REQUIRE(dd_dtype_ok(a->dtype), "dd_conv_wgrad: bad dtype %d", a->dtype);
  const int esz = a->dtype == DD_F32 ? 4 : 2, per16 = 16 / esz, kc = DD_LDS_ROW / esz;
  const bool gather = (a->flags & DD_GATHER2X2) != 0;
  DD_REQUIRE(a->taps == 9 || a->taps == 1 || (a->taps == 4 && gather), "dd_conv_wgrad: taps=%d unsupported", a->taps);
  DD_REQUIRE(!gather || a->taps == 4, "dd_conv_wgrad: DD_GATHER2X2 needs taps=4");
  const int mv = (a->m + per16 - 1) / per16 * per16, nv = (a->n + per16 - 1) / per16 * per16;
  DD_REQUIRE(a->m > 0 && a->n > 0 && a->ldp % per16 == 0 && a->ldq % per16 == 0 && mv <= a->ldp && nv <= a->ldq,
             "dd_conv_wgrad: m=%d n=%d ldp=%d ldq=%d: ld must be a multiple of %d and cover the rounded channel count", a->m, a->n, a->ldp, a->ldq, per16);
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "dd_conv_wgrad: empty grid");
  DD_REQUIRE(((uintptr_t)a->p % 16) == 0 && ((uintptr_t)a->q % 16) == 0, "dd_conv_wgrad: pointers must be 16-byte aligned");
  DD_REQUIRE(a->bias_mode >= 0 && a->bias_mode <= 2 && (a->bias_mode == 0 || a->bias_out || a->stack_blocks > 0), "dd_conv_wgrad: bias_mode=%d needs bias_out", a->bias_mode);
  // wide 1x1 layers: the gradient as 256 x 256 GEMM tiles over the linear pixel index (csrc/dd_conv_pw.hip)
  if (dd_wgrad_pw_eligible(a)) return dd_wgrad_pw_launch(a, reinterpret_cast<hipStream_t>(stream));
  WgradP p;
  p.stack_blocks = a->stack_blocks; p.stack_width = a->stack_width; p.stack_m0 = a->stack_m0;
  for (int j = 0; j < 8; ++j) { p.stack_out[j] = j < a->stack_blocks ? a->stack_out[j] : nullptr; p.stack_bias[j] = j < a->stack_blocks ? a->stack_bias[j] : nullptr; }
  p.p = a->p; p.q = a->q; p.out = a->out; p.bias_out = a->bias_out; p.bias_mode = a->bias_mode;
  p.ldp = a->ldp; p.m = a->m; p.ldq = a->ldq; p.n = a->n; p.mv = mv; p.nv = nv;
  p.B = a->B; p.H = a->H; p.W = a->W; p.taps = a->taps; p.flags = a->flags;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.mslices = dd_ceil_div(a->m, kc); p.nslices = dd_ceil_div(a->n, kc);
  p.hin = gather ? 2 * a->H : a->H; p.win = gather ? 2 * a->W : a->W;
  const long total_tiles = (long)a->B * p.tiles_x * p.tiles_y;
  int ksplit = a->ksplit;
  if (ksplit <= 0) {
    const int target = 256;   // measured: 256 workgroups (1/CU) beat 128 / 512 / 1024 (atomics vs overlap)
    ksplit = (int)(target / ((long)p.mslices * p.nslices));
    if (ksplit < 1) ksplit = 1;
  }
  if (ksplit > total_tiles) ksplit = (int)total_tiles;
  p.ksplit = ksplit;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_DISPATCH_DTYPE(a->dtype, T, return dispatch<T>(p, s));
  return DD_ERR_INVALID;
}
