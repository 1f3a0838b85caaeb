// Weight gradient of a 3x3 SAME convolution with 65..96 output channels (the U-Net's 96-channel level), one pass over x and dy.
//
// Reference seam: Conv2DBackpropFilter + BiasAddGrad of tf.layers.conv2d as TensorFlow's autodiff emits them (Training.py:701-702 over
// UNet.py:38-48):  dW[t][ci][co] = sum_p x[p + off(t)][ci] * dy[p][co],  db[co] = sum_p dy[p][co].
// csrc/dd_conv_wgrad.hip splits such a layer into (64 | 32) x (64 | 32) channel-slice pairs, each pair its own set of workgroups walking every
// pixel tile: four tile passes where one carries a quarter of the arithmetic, x and dy each staged twice.  Here ONE workgroup holds the whole
// 9 x 96 x 96 gradient block: 12 waves (3 per SIMD), wave = input-channel tile i (of 6) x half h of the output channels (3 tiles), 27
// accumulator tiles = 108 registers each, updated by in-place MFMAs for the whole launch (see csrc/dd_conv_bwd.hip on why inline asm, and on
// its precondition: no spills).  Tiles are 16 x 8 pixels: the haloed x image (18 x 10 pixels, two 64-channel slices) and the dy image (two
// slices) are 78 KiB, double buffered by LDS-DMA from all 12 waves; both MFMA operands are read transposed (ds_read_b64_tr_b16: the reduction
// runs over pixels).  Input channels beyond 96 run as further workgroup columns (dy re-read per column).
#include "dd_common.h"

namespace {

struct W96P {
  const void* x; const void* dy; float* dw; float* db;
  int ldx, lddy, cin, cout, cinv, coutv;
  int B, H, W, tiles_x, tiles_y, nblk, ksplit;
};

typedef __attribute__((address_space(3))) s16x4_t* w96_tr_ptr;
typedef uint32_t w96_u32x4 __attribute__((ext_vector_type(4)));
constexpr int W96_TW = 16, W96_TH = 8, W96_PW = W96_TW + 2, W96_PH = W96_TH + 2;
constexpr int W96_PCH = (W96_PW * W96_PH + 7) / 8, W96_QCH = W96_TW * W96_TH / 8;      // 23 and 16 chunks of 8 pixels per 64-channel slice
constexpr int W96_P_SLICE = W96_PCH * 1024, W96_Q_SLICE = W96_QCH * 1024;
constexpr int W96_P_BYTES = 2 * W96_P_SLICE, W96_BUF = W96_P_BYTES + 2 * W96_Q_SLICE;   // 46 + 32 = 78 KiB
constexpr int W96_WAVES = 12;

__device__ __forceinline__ void w96_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 w96_tr_pair(unsigned a0, unsigned a1) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<w96_tr_ptr>(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<w96_tr_ptr>(a1));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return uint4{l2.x, l2.y, h2.x, h2.y};
}
template <typename T> __device__ __forceinline__ void w96_mma_inplace(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void w96_mma_inplace<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const w96_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}
template <> __device__ __forceinline__ void w96_mma_inplace<f16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const w96_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}

struct W96Tile { int b, y0, x0; bool live; };

template <typename T>
__global__ __launch_bounds__(W96_WAVES * 64) void wgrad96_kernel(const W96P a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  static_assert(sizeof(T) == 2, "bf16 / fp16 storage");
  constexpr int PW = W96_PW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int blk = blockIdx.x / a.ksplit, ks = blockIdx.x - blk * a.ksplit;
  const int per_img = a.tiles_x * a.tiles_y, total = a.B * per_img;
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
  const int tile0 = (ks % xcd_n) * (a.ksplit / xcd_n) + ks / xcd_n;
  auto tile_at = [&](int tile) {
    W96Tile t;
    t.live = tile < total;
    const int u = t.live ? tile : 0;
    t.b = u / per_img;
    const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
    t.y0 = ty * W96_TH; t.x0 = (rem - ty * a.tiles_x) * W96_TW;
    return t;
  };
  // ---- DMA: 78 chunks per tile, chunk id = k*12 + wave: ids < 46 = x (slice id / 23, pixels (id % 23)*8 + r of the haloed tile), the rest dy
  const int r0 = lane >> 3, ls0 = (lane & 7) ^ r0;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* X = reinterpret_cast<const char*>(a.x);
  const char* DY = reinterpret_cast<const char*>(a.dy);
  auto piece = [&](int k, const W96Tile& t, unsigned buf) {
    const int id = k * W96_WAVES + wave;
    int r = r0, ls = ls0;
    asm volatile("" : "+v"(r), "+v"(ls));      // (keeps the per-piece coordinates from being hoisted out of the tile loop: see csrc/dd_conv_bwd.hip)
    if (id < 2 * W96_PCH) {
      const int s = id >= W96_PCH ? 1 : 0, c = id - s * W96_PCH;
      const int pix = c * 8 + r;
      const int py = (pix * 3641) >> 16, px = pix - py * PW;      // pix / 18 for pix < 400
      const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px, chb = s * 64 + ls * 8, ch = blk * 96 + chb;
      const bool ok = t.live && pix < PW * W96_PH && chb < 96 && ch < a.cinv && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      w96_dma_1k(ok ? X + ((((long)t.b * a.H + gy) * a.W + gx) * a.ldx + ch) * 2 : zero, buf + s * W96_P_SLICE + c * 1024);
    } else if (id < 2 * W96_PCH + 2 * W96_QCH) {
      const int q2 = id - 2 * W96_PCH, s = q2 >= W96_QCH ? 1 : 0, c = q2 - s * W96_QCH;
      const int row = c >> 1, col = (c & 1) * 8 + r, ch = s * 64 + ls * 8;
      const int gy = t.y0 + row, gx = t.x0 + col;
      const bool ok = t.live && ch < a.coutv && gy < a.H && gx < a.W;
      w96_dma_1k(ok ? DY + ((((long)t.b * a.H + gy) * a.W + gx) * a.lddy + ch) * 2 : zero, buf + W96_P_BYTES + s * W96_Q_SLICE + c * 1024);
    }
  };
  constexpr int NPIECE = (2 * W96_PCH + 2 * W96_QCH + W96_WAVES - 1) / W96_WAVES;      // 7
  {
    const W96Tile t0 = tile_at(tile0);
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) piece(k, t0, lds_base);
  }

  // ---- this wave's block of the gradient: input-channel tile it (of this column's 6), output-channel tiles 3*h .. 3*h + 2
  const int it = wave % 6, h = wave / 6;
  const int li = lane & 15, q4 = (lane >> 4) * 4;
  const bool active = blk * 96 + it * 16 < a.cin && h * 48 < a.cout;
  const bool bias_wave = a.db != nullptr && blk == 0 && it == 0;
  // transposed-read addresses (32-bit LDS offsets of the CURRENT buffer): lane (t16, gq): pixel row gq >> 1 of the step's two rows, column
  // (gq & 1)*8 + (t16 >> 2) [+ 4], 4-channel piece t16 & 3.  x image: pixel pl + C (C = row*18 + dx, a compile-time constant) -> eight bases by
  // (C & 7); dy image: pixel ql + row*16 -> two bases; output-channel tile 3h + j enters as slice / slot offsets computed per j below.
  unsigned pb[8], qb[3][2];
  {
    const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3;
    const int yl = gq >> 1, xl = (gq & 1) * 8 + (t16 >> 2), halfb = (sub & 1) * 8;
    const int pl = yl * PW + xl, ql = yl * W96_TW + xl;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      pb[c] = lds_base + (it >> 2) * W96_P_SLICE + pl * DD_LDS_ROW + ((((it & 3) * 2 + (sub >> 1)) ^ ((pl + c) & 7)) << 4) + halfb;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ct = 3 * h + j;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        qb[j][hh] = lds_base + W96_P_BYTES + (ct >> 2) * W96_Q_SLICE + (ql + 4 * hh) * DD_LDS_ROW + ((((ct & 3) * 2 + (sub >> 1)) ^ ((ql + 4 * hh) & 7)) << 4) + halfb;
    }
  }
  f32x4_t acc[9][3];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[3] = {0.f, 0.f, 0.f};
  auto x_frag = [&](int step) {      // x^T shifted by tap t = step % 9, rows 2s, 2s+1 (s = step / 9)
    const int s = step / 9, t = step - 9 * s;
    const int c0 = (2 * s + t / 3) * PW + t % 3;
    return w96_tr_pair(pb[c0 & 7] + c0 * DD_LDS_ROW, pb[(c0 + 4) & 7] + (c0 + 4) * DD_LDS_ROW);
  };

  int sel = 0;
  for (int tile = tile0; tile < total; tile += a.ksplit, sel ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's chunks of `tile` have landed
    __syncthreads();                         // ... and everyone's; buffer sel^1 is free
    const W96Tile nxt = tile_at(tile + a.ksplit);
    const unsigned nbuf = lds_base + (sel ^ 1) * W96_BUF;
    if (!active) {
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) piece(k, nxt, nbuf);
      continue;
    }
    // 36 steps = 4 pixel-row pairs x 9 taps, 3 MFMAs each; the x fragment of step n+1 is requested before the MFMAs of step n; the three dy
    // fragments of the next row pair are re-read in place right behind the last tap's MFMAs (a second set would cost 12 of the 168 registers a
    // wave has at 3 per SIMD; the other two waves of the SIMD cover the wait)
    uint4 xf[2], df[3];
    xf[0] = x_frag(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) df[j] = w96_tr_pair(qb[j][0], qb[j][1]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int step = s * 9 + t;
        if (step + 1 < 36) xf[(step + 1) & 1] = x_frag(step + 1);
        {      // the DMA pieces of the next tile, spread evenly over the 36 steps
          const int k0 = (step * NPIECE + 35) / 36;
          if (k0 < NPIECE && (k0 * 36) / NPIECE == step) piece(k0, nxt, nbuf);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) w96_mma_inplace<T>(acc[t][j], xf[step & 1], df[j]);      // D[ci][co]
        if (t == 4 && bias_wave) {      // B fragment: 8 pixels of output channel (3h + j)*16 + li per lane
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float f[8];
            unpack8t<T>(df[j], f);
            bsum[j] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t == 8 && s + 1 < 4) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            df[j] = w96_tr_pair(qb[j][0] + (2 * (s + 1) * W96_TW) * DD_LDS_ROW, qb[j][1] + (2 * (s + 1) * W96_TW) * DD_LDS_ROW);
        }
      }
    const int flip = sel ? -W96_BUF : W96_BUF;
#pragma unroll
    for (int c = 0; c < 8; ++c) pb[c] += flip;
#pragma unroll
    for (int j = 0; j < 3; ++j) { qb[j][0] += flip; qb[j][1] += flip; }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (see w96_mma_inplace)
  // flush: D[t][j] rows = input channels blk*96 + it*16 + q4 + e, column = output channel (3h + j)*16 + li; TensorFlow layout [3][3][cin][cout]
  if (active) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int co = (3 * h + j) * 16 + li;
      if (co >= a.cout) continue;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ci = blk * 96 + it * 16 + q4 + e;
          if (ci < a.cin) atomicAdd(a.dw + ((long)t * a.cin + ci) * a.cout + co, acc[t][j][e]);
        }
      if (bias_wave) {
        float b = bsum[j];
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        if (lane < 16) atomicAdd(a.db + co, b);
      }
    }
  }
}

static int w96_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

}  // namespace

// Is this dd_conv_wgrad call one the single-pass kernel takes?  (3x3, bf16 / f16 storage, 65..96 output channels, bias gradient from dy)
bool dd_wgrad96_eligible(const dd_wgrad_args* a) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("DD_WGRAD96"); on = e ? atoi(e) : 1; }
  return on && a->taps == 9 && (a->dtype == DD_BF16 || a->dtype == DD_F16) && a->flags == 0 && a->n > 64 && a->n <= 96 && a->m > 64 &&
         (a->bias_mode == 0 || a->bias_mode == 1) && a->ldp % 8 == 0 && a->ldq % 8 == 0;
}

int dd_wgrad96_launch(const dd_wgrad_args* a, hipStream_t stream) {
  W96P p;
  p.x = a->p; p.dy = a->q; p.dw = a->out; p.db = a->bias_mode == 1 ? a->bias_out : nullptr;
  p.ldx = a->ldp; p.lddy = a->ldq; p.cin = a->m; p.cout = a->n; p.cinv = (a->m + 7) / 8 * 8; p.coutv = (a->n + 7) / 8 * 8;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, W96_TW); p.tiles_y = dd_ceil_div(a->H, W96_TH);
  p.nblk = dd_ceil_div(a->m, 96);
  const long total = (long)a->B * p.tiles_x * p.tiles_y;
  long ksplit = w96_cus() / p.nblk;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > total) ksplit = total;
  p.ksplit = (int)ksplit;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad96_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad96_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid((unsigned)(p.nblk * p.ksplit));
  if (a->dtype == DD_BF16) hipLaunchKernelGGL(wgrad96_kernel<bf16_t>, grid, dim3(W96_WAVES * 64), 2 * (size_t)W96_BUF, stream, p);
  else hipLaunchKernelGGL(wgrad96_kernel<f16_t>, grid, dim3(W96_WAVES * 64), 2 * (size_t)W96_BUF, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
