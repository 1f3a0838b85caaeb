// 2x2 / stride-2 transposed convolution (the U-Net up-convolution) on CDNA4: forward, and data + weight + bias gradients in one launch.
//
// Reference seam replaced: tf.layers.conv2d_transpose(filters, 2, strides=2) + ReLU of UNet.py:54-59 and the three gradient ops TensorFlow's
// autodiff emits for it (Training.py:701-702).  With x [B,H,W,Cin], y / dy [B,2H,2W,Cout] and the kernel K[a][b][co][ci] (TensorFlow's
// conv2d_transpose layout [kh,kw,out,in]):
//   y[2i+a][2j+b][co] = relu(bias[co] + sum_ci x[i][j][ci] * K[a][b][co][ci])
//   dx[i][j][ci]      = (x[i][j][ci] > 0) * sum_{a,b,co} dy[2i+a][2j+b][co] * K[a][b][co][ci]
//   dK[a][b][co][ci]  = sum_{i,j} dy[2i+a][2j+b][co] * x[i][j][ci]              db[co] = sum dy[.][.][co]
// There is almost no arithmetic here (2*4*Cin*Cout flop per input pixel): every one of these is a stream over x and y.  The layer-wise path ran
// them through the 3x3 machinery (dd_conv_igemm with DD_PIXSHUF / DD_GATHER2X2, dd_conv_wgrad): 4 channel blocks each re-reading x forward,
// dy fetched by both gradient launches, one tap = one outer iteration with its own barriers: 0.9 ms per step for 1 % of the FLOPs.
// Here a workgroup walks 8x8 input-pixel tiles (16x16 output pixels); tiles arrive by LDS-DMA one tile ahead (double buffered, 1 KiB chunks
// of 8 pixels x 128 bytes, swizzled on the pixel index); the weights of a wave's output rows live in registers as the MFMA A operand, the
// pixels are B, so results come out as [channel][pixel] and are stored straight from the accumulators, 4 channels (8 bytes) per lane.
// Backward: all 8 waves compute dx (wave = 16 input channels) and then dK (wave = tap x half of the output channels, transposed LDS reads,
// accumulated in registers for the whole launch, one set of atomics per workgroup).  C_in <= 128, C_out <= 64, bf16 / f16 storage.
#include "dd_common.h"

namespace {

struct CtP {
  const void* x; const void* y; const void* w; void* out; float* dw; float* db; const float* bias;
  int ldx, ldy, ldo;               // channel strides (elements): x, y (fwd: output; bwd: dy), bwd: dx
  int cin, cout, cinv, coutv;      // logical / staged (rounded up to 8) channel counts
  int n_pad, k_pad;                // packed weights: fwd [4*cout -> n_pad][k_pad (ci)], bwd [4][n_pad (ci)][k_pad (co)]
  int B, H, W, tiles_x, tiles_y, nwg;
  int relu;
  int co_off, cout_total;          // backward over a block of output channels [co_off, co_off + cout) of cout_total (dy / weight / dK / db offsets)
};

typedef __attribute__((address_space(3))) s16x4_t* ct_tr_ptr;
typedef uint32_t ct_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ct_u32x2 __attribute__((ext_vector_type(2)));

constexpr int CT_T = 8;                                             // input pixels per tile side
constexpr int CT_X_SLICE = CT_T * CT_T * DD_LDS_ROW;                // x image of one 64-channel slice: 64 pixels x 128 bytes
constexpr int CT_X_BYTES = 2 * CT_X_SLICE;                          // C_in <= 128
constexpr int CT_Y_BYTES = 4 * CT_T * CT_T * DD_LDS_ROW;            // dy image: 32 KiB = 16 x 16 pixels x 64 channels, or (C_out > 64) two 64-channel slices of 16 x 8
constexpr int CT_BWD_BUF = CT_Y_BYTES + CT_X_BYTES;                 // 48 KiB
constexpr int CT_FWD_BUF = CT_X_BYTES;                              // 16 KiB

__device__ __forceinline__ void ct_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 ct_lds16(unsigned off) {
  const ct_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) ct_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ uint2 ct_lds8(unsigned off) {
  const ct_u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) ct_u32x2*>(off);
  return uint2{v[0], v[1]};
}
__device__ __forceinline__ uint4 ct_tr_pair(unsigned a0, unsigned a1) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ct_tr_ptr>(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ct_tr_ptr>(a1));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return uint4{l2.x, l2.y, h2.x, h2.y};
}

struct CtTile { int b, i0, j0; bool live; };
__device__ __forceinline__ CtTile ct_tile(const CtP& a, int tile, int total, int th = CT_T) {      // tiles are 8 input pixels wide, th high
  CtTile t;
  t.live = tile < total;
  const int u = t.live ? tile : 0, per_img = a.tiles_x * a.tiles_y;
  t.b = u / per_img;
  const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
  t.i0 = ty * th; t.j0 = (rem - ty * a.tiles_x) * CT_T;
  return t;
}
// DMA of the x tile (8x8 input pixels, two 64-channel slices): 16 chunks, chunk xc = slice * 8 + tile row; a lane fetches pixel r = lane >> 3 of
// the row, logical 16-byte slot ls = (lane & 7) ^ r
template <int TH = CT_T>
__device__ __forceinline__ void ct_dma_x(const CtP& a, const CtTile& t, int xc, unsigned lds, int lane) {
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const int sx = xc / TH, row = xc % TH, ch = sx * 64 + ls * 8;
  const bool ok = t.live && ch < a.cinv && t.i0 + row < a.H && t.j0 + r < a.W;
  const char* src = reinterpret_cast<const char*>(a.x) + (((long)t.b * a.H + t.i0 + row) * a.W + t.j0 + r) * a.ldx * 2 + ch * 2;
  ct_dma_1k(ok ? src : reinterpret_cast<const char*>(&dd_zero16_v), lds + xc * 1024);
}
// DMA of the dy tile (NSD 64-channel slices of 16 x 16/NSD output pixels): 32 chunks, chunk c = slice c / (32/NSD), output row (c % (32/NSD)) >> 1,
// pixels (c & 1) * 8 + r
template <int NSD>
__device__ __forceinline__ void ct_dma_y(const CtP& a, const CtTile& t, int c, unsigned lds, int lane) {
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const int sd = c / (32 / NSD), cc = c % (32 / NSD);
  const int row = cc >> 1, col = (cc & 1) * 8 + r, ch = sd * 64 + ls * 8;
  const int gy = 2 * t.i0 + row, gx = 2 * t.j0 + col;
  const bool ok = t.live && ch < a.coutv && gy < 2 * a.H && gx < 2 * a.W;
  const char* src = reinterpret_cast<const char*>(a.y) + (((long)t.b * 2 * a.H + gy) * 2 * a.W + gx) * a.ldy * 2 + (a.co_off + ch) * 2;
  ct_dma_1k(ok ? src : reinterpret_cast<const char*>(&dd_zero16_v), lds + c * 1024);
}

// ------------------------------------------------------------------------------------------------------------------------ forward
// wave w computes output rows n = (a, b, co) of the n-tiles nt = w, w + 8, ... (< 4*cout/16), each for all 64 input pixels of the tile
template <typename T, int KC>      // KC = K chunks of 32 input channels (ceil(C_in / 32): <= 4)
__global__ __launch_bounds__(512) void convt_fwd_kernel(const CtP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int NTW = 3;           // n-tiles per wave: 4 * 64 / 16 = 16 over 8 waves -> 2 (C_out = 64); 3 covers C_out <= 96
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, q = lane >> 4;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int total = a.B * a.tiles_x * a.tiles_y;
  const int ntiles = (4 * a.cout) >> 4;
  const T* Wp = reinterpret_cast<const T*>(a.w);
  const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
  uint4 wf[NTW][KC];
  float bv[NTW][4];
  int och[NTW], opix[NTW];          // output channel of this lane's 4 results; pixel offset (a * 2W + b) of the tap
  bool nt_ok[NTW];
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int nt = wave + 8 * u;
    nt_ok[u] = nt < ntiles;
    const int nrow = nt * 16 + li;                       // A row of this lane
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int k0 = kc * 32 + q * 8;
      const bool ok = nt_ok[u] && nrow < a.n_pad && k0 < a.k_pad;
      wf[u][kc] = *reinterpret_cast<const uint4*>(ok ? Wp + (long)nrow * a.k_pad + k0 : zw);
    }
    const int n4 = nt * 16 + q * 4;                      // first of the 4 output rows this lane holds (all in one tap: cout % 16 == 0)
    const int tap = n4 / a.cout, co = n4 - tap * a.cout;
    och[u] = co;
    opix[u] = (tap >> 1) * 2 * a.W + (tap & 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[u][e] = (a.bias && nt_ok[u]) ? a.bias[co + e] : 0.f;
  }
  // x image: pixel g*16 + li (g = pair of tile rows), K chunk kc = slice kc >> 1, slot (kc & 1) * 4 + q
  unsigned xb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) xb[h] = lds_base + li * DD_LDS_ROW + (((h * 4 + q) ^ (li & 7)) << 4);
  T* __restrict__ Y = reinterpret_cast<T*>(a.out);

  CtTile cur = ct_tile(a, blockIdx.x, total);
  ct_dma_x(a, cur, wave, lds_base, lane);
  ct_dma_x(a, cur, wave + 8, lds_base, lane);
  int sel = 0;
  for (int tile = blockIdx.x; tile < total; tile += a.nwg, sel ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const CtTile nxt = ct_tile(a, tile + a.nwg, total);
    ct_dma_x(a, nxt, wave, lds_base + (sel ^ 1) * CT_FWD_BUF, lane);
    ct_dma_x(a, nxt, wave + 8, lds_base + (sel ^ 1) * CT_FWD_BUF, lane);
    const unsigned xt = sel * CT_FWD_BUF;
    f32x4_t acc[4][NTW];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int u = 0; u < NTW; ++u) acc[g][u] = f32x4_t{bv[u][0], bv[u][1], bv[u][2], bv[u][3]};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 f = ct_lds16(xt + xb[kc & 1] + (kc >> 1) * CT_X_SLICE + g * 16 * DD_LDS_ROW);
#pragma unroll
        for (int u = 0; u < NTW; ++u) acc[g][u] = mma16<T>(wf[u][kc], f, acc[g][u]);
      }
    // input pixel of this lane in group g: (i0 + 2g + (li >> 3), j0 + (li & 7)) -> output pixel (2i + a, 2j + b)
    const int jj = cur.j0 + (li & 7);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ii = cur.i0 + 2 * g + (li >> 3);
      const bool ok = ii < a.H && jj < a.W;
      const long p0 = ((long)cur.b * 2 * a.H + 2 * ii) * 2 * a.W + 2 * jj;
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        f32x4_t v = acc[g][u];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        uint2 o2;
        o2.x = pack2<T>(v[0], v[1]);
        o2.y = pack2<T>(v[2], v[3]);
        if (ok && nt_ok[u]) *reinterpret_cast<uint2*>(Y + (p0 + opix[u]) * a.ldy + och[u]) = o2;
      }
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------------------------------ backward
// acc += A * B with the accumulator tied to one register tuple (see csrc/dd_conv_bwd.hip: through the builtin hipcc renames long-lived accumulator
// tuples and spills; these are touched once per K step and read only after the tile loop, behind explicit s_nops)
template <typename T> __device__ __forceinline__ void ct_mma_inplace(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void ct_mma_inplace<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const ct_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}
template <> __device__ __forceinline__ void ct_mma_inplace<f16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const ct_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}

// NSD = 64-channel slices of dy (1: C_out <= 64, tiles of 8 x 8 input pixels; 2: C_out <= 128, tiles of 8 x 4 -- the dy image stays 32 KiB)
template <typename T, int NSD, bool MASK, bool ACCUM>
__global__ __launch_bounds__(512) void convt_bwd_kernel(const CtP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int TH = CT_T / NSD, NG = TH / 2, KC = 2 * NSD, JW = 2 * NSD, KS = TH / 4;      // tile rows, 16-pixel groups, K chunks of dx, output-channel tiles per wave of dK, K steps of dK
  constexpr int Y_SLICE = CT_Y_BYTES / NSD, X_SLICE = CT_T * TH * DD_LDS_ROW, BUF = CT_Y_BYTES + 2 * X_SLICE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, q = lane >> 4;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int total = a.B * a.tiles_x * a.tiles_y;

  // ---- data gradient: wave = input-channel tile `wave` (16 channels), all input pixels of the tile in NG groups of 16 (2 tile rows)
  const bool d_active = wave * 16 < a.cin;
  uint4 wf[4][KC];                 // [tap][K chunk of 32 output channels]
  {
    const T* Wd = reinterpret_cast<const T*>(a.w);
    const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
    const int ci_row = wave * 16 + li;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int k0 = kc * 32 + q * 8;
        const bool ok = ci_row < a.n_pad && k0 < a.coutv && a.co_off + k0 < a.k_pad;
        wf[t][kc] = *reinterpret_cast<const uint4*>(ok ? Wd + ((long)t * a.n_pad + ci_row) * a.k_pad + a.co_off + k0 : zw);
      }
  }
  // dy image addresses: input pixel (2g + (li >> 3), li & 7) of group g, tap (ta, tb) -> dy pixel pd = (2*row + ta) * 16 + 2*col + tb;
  // group g adds 64 pixels; K chunk kc = slice kc >> 1, slot (kc & 1)*4 + q; swizzle key pd & 7 = (2*col + tb) & 7
  unsigned db[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int pd = (2 * (li >> 3) + (t >> 1)) * 16 + 2 * (li & 7) + (t & 1);
    db[t] = lds_base + pd * DD_LDS_ROW + ((q ^ (pd & 7)) << 4);
  }
  // x image (behind the dy image): mask of this lane's results = pixel g*16 + li, channels wave*16 + q*4 ..
  const unsigned mb = lds_base + CT_Y_BYTES + (wave >> 2) * X_SLICE + li * DD_LDS_ROW + ((((wave & 3) * 2 + (q >> 1)) ^ (li & 7)) << 4) + (q & 1) * 8;
  const int c4 = wave * 16 + q * 4;
  T* __restrict__ DX = reinterpret_cast<T*>(a.out);

  // ---- weight gradient: wave = tap tw x half h of the output channels (output-channel tiles h*JW .. h*JW + JW-1) x all 8 input-channel tiles
  const int tw = wave & 3, h = wave >> 2;
  const bool w_active = h * JW * 16 < a.cout;
  const int nci = (a.cin + 15) >> 4;
  unsigned yb[2], xb[2];           // transposed reads: lane (t16, gq): pixel row gq (of the 4 rows of a K step), column (t16 >> 2) [+ 4], 4-channel piece t16 & 3
  {
    const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int col = (t16 >> 2) + 4 * hh;
      const int pd = (2 * gq + (tw >> 1)) * 16 + 2 * col + (tw & 1), px = gq * 8 + col;
      // output-channel tile h*JW + j: slice (h*JW + j) >> 2 (NSD = 2: slice h), slot ((h*JW + j) & 3)*2 + (sub >> 1): j enters as ^ (j << 5)
      yb[hh] = lds_base + ((h * JW) >> 2) * Y_SLICE + pd * DD_LDS_ROW + (((((h * JW) & 3) * 2 + (sub >> 1)) ^ (pd & 7)) << 4) + (sub & 1) * 8;
      xb[hh] = lds_base + CT_Y_BYTES + px * DD_LDS_ROW + (((sub >> 1) ^ (px & 7)) << 4) + (sub & 1) * 8;        // input-channel tile 0; tile i: ^ ((i & 3) << 5), + (i >> 2) * slice
    }
  }
  f32x4_t wacc[JW][8];
#pragma unroll
  for (int j = 0; j < JW; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) wacc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[JW];
#pragma unroll
  for (int j = 0; j < JW; ++j) bsum[j] = 0.f;

  // ---- DMA: 32 dy + 2*TH x chunks per tile, 5-6 per wave, all issued right behind the barrier: the arithmetic of a tile is short
  auto dma_tile = [&](const CtTile& t, unsigned buf) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ct_dma_y<NSD>(a, t, k * 8 + wave, buf, lane);
    ct_dma_x<TH>(a, t, wave, buf + CT_Y_BYTES, lane);
    if (NSD == 1) ct_dma_x<TH>(a, t, wave + 8, buf + CT_Y_BYTES, lane);
  };
  // (Round 4: three buffers with two tiles in flight and a vmcnt(pieces of one tile) wait at the top measured no faster -- 146 -> 150 us at
  //  64x64 -> 128x128, 96 -> 64 channels: the loop is not waiting for the DMA.)
  constexpr int NPW = 5 + (NSD == 1 ? 1 : 0);      // DMA pieces per wave and tile (dma_tile)
  CtTile cur = ct_tile(a, blockIdx.x, total, TH);
  dma_tile(cur, lds_base);
  int sel = 0;
  for (int tile = blockIdx.x; tile < total; tile += a.nwg, sel ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const CtTile nxt = ct_tile(a, tile + a.nwg, total, TH);
    // ACCUM: the gradient already in dx is requested BEFORE the next tile's DMA pieces and waited for with vmcnt(pieces of this wave): loads return
    // in order, so nothing of the DMA has to land first.  (Round 3 loaded it with a plain C++ load behind the pieces: hipcc's own s_waitcnt
    // before its use is vmcnt(0) -- it does not count the inline-asm DMA -- and every tile waited for the whole NEXT tile to arrive, an HBM round
    // trip per tile.)  The request is inline asm for the same reason; its registers are only touched again behind the explicit wait.  hipcc
    // believes oldr[] defined right behind the request, so nothing in the language stops it from copying or re-allocating those registers
    // between the two asm statements (a v_mov there would read them before the load has landed): deepdenoiser_amd/build.py compiles this file
    // to ISA and REFUSES the build if any instruction between a `dd_accum_request` and the next `dd_accum_wait` names a request register.
    const int jj = cur.j0 + (li & 7);
    const bool ch_ok = c4 < a.cinv && jj < a.W;
    ct_u32x2 oldr[NG];
    if (ACCUM) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ii = cur.i0 + 2 * g + (li >> 3);
        const bool ok = d_active && ch_ok && ii < a.H;
        const void* op = ok ? static_cast<const void*>(DX + (((long)cur.b * a.H + ii) * a.W + jj) * a.ldo + c4) : static_cast<const void*>(&dd_zero16_v);
        asm volatile("global_load_dwordx2 %0, %1, off ; dd_accum_request" : "=v"(oldr[g]) : "v"(op) : "memory");
      }
    }
    dma_tile(nxt, lds_base + (sel ^ 1) * BUF);
    const unsigned bt = sel * BUF;
    if (d_active) {
      f32x4_t acc[NG];
      uint2 mv[NG], oldv[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (MASK) mv[g] = ct_lds8(bt + mb + g * 16 * DD_LDS_ROW);
        oldv[g] = uint2{0u, 0u};
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const uint4 f = ct_lds16(bt + (db[t] ^ ((kc & 1) << 6)) + (kc >> 1) * Y_SLICE + g * 64 * DD_LDS_ROW);
            acc[g] = mma16<T>(wf[t][kc], f, acc[g]);
          }
      if (ACCUM) {
        static_assert(NG == 4 || NG == 2, "the wait below names every request register");
        if constexpr (NG == 4) {
          if constexpr (NPW == 6) asm volatile("s_waitcnt vmcnt(6) ; dd_accum_wait" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(oldr[3]));
          else asm volatile("s_waitcnt vmcnt(5) ; dd_accum_wait" : "+v"(oldr[0]), "+v"(oldr[1]), "+v"(oldr[2]), "+v"(oldr[3]));
        } else {
          if constexpr (NPW == 6) asm volatile("s_waitcnt vmcnt(6) ; dd_accum_wait" : "+v"(oldr[0]), "+v"(oldr[1]));
          else asm volatile("s_waitcnt vmcnt(5) ; dd_accum_wait" : "+v"(oldr[0]), "+v"(oldr[1]));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) oldv[g] = uint2{oldr[g][0], oldr[g][1]};
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ii = cur.i0 + 2 * g + (li >> 3);
        uint2 o2;
        o2.x = pack2<T>(acc[g][0], acc[g][1]);
        o2.y = pack2<T>(acc[g][2], acc[g][3]);
        if (MASK) { o2.x = mask_bf16x2(o2.x, mv[g].x); o2.y = mask_bf16x2(o2.y, mv[g].y); }
        if (ACCUM) {
          float f8[8], g8[8];
          unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
          unpack8t<T>(uint4{oldv[g].x, oldv[g].y, 0u, 0u}, g8);
          o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
          o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
        }
        if (ch_ok && ii < a.H) *reinterpret_cast<uint2*>(DX + (((long)cur.b * a.H + ii) * a.W + jj) * a.ldo + c4) = o2;
      }
    }
    if (w_active) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {      // K step = 32 input pixels = tile rows 4s .. 4s+3  (dy rows 8s .. 8s+7: + 128 pixels)
        uint4 yf[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j)
          yf[j] = ct_tr_pair(bt + (yb[0] ^ (j << 5)) + s * 128 * DD_LDS_ROW, bt + (yb[1] ^ (j << 5)) + s * 128 * DD_LDS_ROW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i >= nci) continue;
          const uint4 xf = ct_tr_pair(bt + (xb[0] ^ ((i & 3) << 5)) + (i >> 2) * X_SLICE + s * 32 * DD_LDS_ROW,
                                      bt + (xb[1] ^ ((i & 3) << 5)) + (i >> 2) * X_SLICE + s * 32 * DD_LDS_ROW);
#pragma unroll
          for (int j = 0; j < JW; ++j) ct_mma_inplace<T>(wacc[j][i], yf[j], xf);      // D[co][ci]
        }
        if (a.db) {
#pragma unroll
          for (int j = 0; j < JW; ++j) {      // A fragment: 8 pixels of output channel (h*JW + j)*16 + li per lane
            float f[8];
            unpack8t<T>(yf[j], f);
            bsum[j] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
          }
        }
      }
    }
    cur = nxt;
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (see ct_mma_inplace)
  // flush: D[j][i] rows = output channels (h*JW + j)*16 + q*4 + e, column = input channel i*16 + li; TensorFlow layout [a][b][co][ci]
  dd_det_wait();
  if (w_active) {
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ci = i * 16 + li;
        if (i >= nci || ci >= a.cin) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = (h * JW + j) * 16 + q * 4 + e;
          if (co < a.cout) atomicAdd(a.dw + ((long)tw * a.cout_total + a.co_off + co) * a.cin + ci, wacc[j][i][e]);
        }
      }
  }
  // bias gradient: the four tap waves of an output-channel half each hold the sum over their own quarter of the output pixels -- four adds to one
  // address from one workgroup; in deterministic mode they go in tap order (a barrier between them: every wave of the workgroup is here)
  const bool det = dd_det_on();
  for (int turn = 0; turn < (det ? 4 : 1); ++turn) {
    if (w_active && a.db && (!det || tw == turn)) {
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        float b = bsum[j];
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        const int co = (h * JW + j) * 16 + li;
        if (lane < 16 && co < a.cout) atomicAdd(a.db + a.co_off + co, b);
      }
    }
    if (det) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
  dd_det_end();
}

static int ct_cus() {
  return dd_device_cus();
}

template <typename K>
static void ct_launch(K kernel, const CtP& p, size_t lds, hipStream_t stream) {
  dd_det_sync();
  dd_allow_max_lds(reinterpret_cast<const void*>(kernel));
  hipLaunchKernelGGL(kernel, dim3((unsigned)p.nwg), dim3(512), lds, stream, p);
}

static int ct_fill(CtP& p, const dd_convt_args* a, bool bwd) {
  const int th = CT_T;
  DD_REQUIRE(a && a->x && a->y && a->w, "dd_convt2x2: null pointer");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_convt2x2: dtype %d (bf16 / f16 storage only; f32 takes dd_conv_igemm / dd_conv_wgrad)", a->dtype);
  DD_REQUIRE(a->cin > 0 && a->cin <= 128 && a->cout > 0 && a->cout <= (bwd ? 128 : 96) && a->cout % 16 == 0,
             "dd_convt2x2: cin=%d cout=%d (cin <= 128; cout a multiple of 16, <= 96 forward / 128 backward)", a->cin, a->cout);
  const int cinv = (a->cin + 7) / 8 * 8, coutv = (a->cout + 7) / 8 * 8;
  DD_REQUIRE(a->ld_x % 8 == 0 && a->ld_y % 8 == 0 && cinv <= a->ld_x && coutv <= a->ld_y, "dd_convt2x2: ld_x=%d ld_y=%d must be multiples of 8 covering the channels", a->ld_x, a->ld_y);
  DD_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->y % 16) == 0 && ((uintptr_t)a->w % 16) == 0, "dd_convt2x2: x / y / w must be 16-byte aligned");
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && (long)a->B * a->H * a->W * 4 < (1L << 31), "dd_convt2x2: empty or oversized grid");
  p.x = a->x; p.y = a->y; p.w = a->w; p.ldx = a->ld_x; p.ldy = a->ld_y;
  p.cin = a->cin; p.cout = a->cout; p.cinv = cinv; p.coutv = coutv; p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, CT_T); p.tiles_y = dd_ceil_div(a->H, th);
  const long total = (long)a->B * p.tiles_x * p.tiles_y;
  p.nwg = (int)(total < ct_cus() ? total : ct_cus());
  p.relu = a->relu;
  p.co_off = 0; p.cout_total = a->cout;
  return DD_OK;
}

}  // namespace

extern "C" int dd_convt2x2_fwd(const dd_convt_args* a, dd_stream stream) {
  CtP p;
  if (int rc = ct_fill(p, a, false)) return rc;
  DD_REQUIRE(a->n_pad >= 4 * a->cout && a->k_pad >= a->cin && a->k_pad % 32 == 0 && a->k_pad <= 128, "dd_convt2x2_fwd: packed weights [n_pad=%d][k_pad=%d]", a->n_pad, a->k_pad);
  p.out = const_cast<void*>(a->y); p.bias = a->bias; p.dw = nullptr; p.db = nullptr; p.ldo = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = 2 * (size_t)CT_FWD_BUF;
  const int kc = a->k_pad / 32;
#define CT_FWD(T)                                                            \
  switch (kc) {                                                              \
    case 1: ct_launch(convt_fwd_kernel<T, 1>, p, lds, s); break;             \
    case 2: ct_launch(convt_fwd_kernel<T, 2>, p, lds, s); break;             \
    case 3: ct_launch(convt_fwd_kernel<T, 3>, p, lds, s); break;             \
    default: ct_launch(convt_fwd_kernel<T, 4>, p, lds, s); break;            \
  }
  if (a->dtype == DD_BF16) { CT_FWD(bf16_t) } else { CT_FWD(f16_t) }
#undef CT_FWD
  DD_LAUNCH_CHECK();
  return DD_OK;
}

extern "C" int dd_convt2x2_bwd(const dd_convt_args* a, dd_stream stream) {
  CtP p;
  if (int rc = ct_fill(p, a, true)) return rc;
  DD_REQUIRE(a->dx && a->dw, "dd_convt2x2_bwd: null gradient pointer");
  DD_REQUIRE(a->n_pad >= a->cin && a->k_pad >= a->cout && a->k_pad % 8 == 0, "dd_convt2x2_bwd: packed weights [4][n_pad=%d][k_pad=%d]", a->n_pad, a->k_pad);
  const int cinv = (a->cin + 7) / 8 * 8;
  DD_REQUIRE(a->ld_dx % 4 == 0 && cinv <= a->ld_dx && ((uintptr_t)a->dx % 8) == 0, "dd_convt2x2_bwd: ld_dx=%d / dx alignment", a->ld_dx);
  p.out = a->dx; p.ldo = a->ld_dx; p.dw = a->dw; p.db = a->db; p.bias = nullptr;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = 2 * (size_t)CT_BWD_BUF;
#define CT_BWD_N(T, N, ACC)                                                                                \
  if (a->use_mask) { if (ACC) ct_launch(convt_bwd_kernel<T, N, true, true>, p, lds, s); else ct_launch(convt_bwd_kernel<T, N, true, false>, p, lds, s); } \
  else { if (ACC) ct_launch(convt_bwd_kernel<T, N, false, true>, p, lds, s); else ct_launch(convt_bwd_kernel<T, N, false, false>, p, lds, s); }
  // More than 64 output channels: one launch per block of 64 (dy is one 64-channel LDS slice per launch), the later ones accumulating into dx.
  // (A two-slice instantiation -- NSD = 2 -- compiles but spills 11 registers next to the in-place asm MFMAs, whose results the compiler may
  //  then store before they are written: measured wrong in f16, so it is not instantiated.)
  for (int co_off = 0; co_off < a->cout; co_off += 64) {
    const int cb = a->cout - co_off < 64 ? a->cout - co_off : 64;
    p.co_off = co_off; p.cout = cb; p.coutv = (cb + 7) / 8 * 8; p.cout_total = a->cout;
    const bool acc = a->accumulate || co_off > 0;
    if (a->dtype == DD_BF16) { CT_BWD_N(bf16_t, 1, acc) } else { CT_BWD_N(f16_t, 1, acc) }
  }
#undef CT_BWD_N
  DD_LAUNCH_CHECK();
  return DD_OK;
}
