// 3x3 SAME convolution with REGISTER-resident weights on CDNA4: the launch dd_conv_igemm uses for layers whose reduction depth is 65..96
// channels (the U-Net's 96-channel level: forward and data gradient, Training.py:701-702 over UNet.py:38-48).
//
// Why a second kernel.  csrc/dd_conv_igemm.hip keeps a layer's weights in LDS; 9 x 96 x 96 bf16 weights are 166 KB, 6 KB more than a CU has, so
// those layers run as two 48-channel blocks -- every input tile is staged twice, each pass with two K slices of their own barriers -- and sit
// at ~690 TFLOP/s where the 64-channel layers reach 900.  Here the weights of a wave's 16 output channels live in its registers as the MFMA A
// operand (9 taps x 3 K-chunks = 27 fragments = 108 VGPRs), the pixels are the B operand, and LDS holds nothing but the input tile:
//   * waves 0-5 = the six 16-channel tiles of a 96-channel output block; each walks the 10 haloed rows of a 16 x 8 pixel tile once: a row's
//     fragment feeds the three output rows it touches (rotating accumulators), an output row is finished two haloed rows later and leaves
//     straight from the accumulators (bias / ReLU / mask / accumulate fused), 4 channels = 8 bytes per lane;
//   * waves 6-7 do nothing but LDS-DMA: the next tile (18 x 10 haloed pixels, two 64-channel slices, 46 chunks of 1 KiB) streams in while this
//     one is multiplied.  The compute waves issue no DMA, so their vmcnt only ever waits for their own mask / gradient loads.
// Output channels beyond 96 run as further blocks (the input is re-read per block).
#include "dd_common.h"

namespace {

struct RwP {
  const void* x; const void* wp; const float* bias; const void* mask; void* y;
  int ldx, ldmask, ldy;
  int cin, cinv, n, n_pad, k_pad, nbias;
  int B, H, W, tiles_x, tiles_y, nblk, ksplit;
  int relu, accum, aux_residual;      // aux_residual: `mask` / ldmask hold the residual operand
};

typedef uint32_t rw_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t rw_u32x2 __attribute__((ext_vector_type(2)));
constexpr int RW_TW = 16, RW_TH = 8, RW_PW = RW_TW + 2, RW_PH = RW_TH + 2;      // tile and haloed tile
constexpr int RW_CH = (RW_PW * RW_PH + 7) / 8;                                 // 23 chunks of 8 pixels per 64-channel slice
constexpr int RW_SLICE = RW_CH * 1024, RW_BUF = 2 * RW_SLICE;                  // 46 KiB per buffer

__device__ __forceinline__ void rw_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 rw_lds16(unsigned off) {
  const rw_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) rw_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}

struct RwTile { int b, y0, x0; bool live; };

// AUX: the optional output-shaped operand read next to the results: 0 none, 1 ReLU-backward mask (y *= aux > 0), 2 residual (y = act(conv + aux),
// the second half of a conv over a channel concat: engine.Graph.conv split_at)
template <typename T, int KC, int AUX, bool ACCUM>
__global__ __launch_bounds__(512) void conv_rw_kernel(const RwP a) {
  constexpr bool MASK = AUX != 0;      // (the loads of the aux rows are the same for both kinds)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  static_assert(sizeof(T) == 2, "register-weight conv: bf16 / fp16 storage");
  constexpr int PW = RW_PW, PH = RW_PH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int blk = blockIdx.x / a.ksplit, ks = blockIdx.x - blk * a.ksplit;
  const int per_img = a.tiles_x * a.tiles_y, total = a.B * per_img;
  // tile sequence: the ksplit/8 workgroups of one XCD (blockIdx % 8) take a contiguous run of each round's tiles (shared halos meet in one L2)
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
  const int tile0 = (ks % xcd_n) * (a.ksplit / xcd_n) + ks / xcd_n;
  auto tile_at = [&](int tile) {
    RwTile t;
    t.live = tile < total;
    const int u = t.live ? tile : 0;
    t.b = u / per_img;
    const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
    t.y0 = ty * RW_TH; t.x0 = (rem - ty * a.tiles_x) * RW_TW;
    return t;
  };

  if (wave >= 6) {
    // ================================================================== I/O role: chunk c of slice s = pixels c*8 + r of the haloed tile (row-major)
    const int r = lane >> 3, ls = (lane & 7) ^ r;
    const int io = wave - 6;
    const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
    const char* X = reinterpret_cast<const char*>(a.x);
    auto load_tile = [&](const RwTile& t, unsigned buf) {
#pragma unroll 1
      for (int c2 = io; c2 < 2 * RW_CH; c2 += 2) {      // 46 chunks over the two I/O waves
        const int s = c2 >= RW_CH ? 1 : 0, c = c2 - s * RW_CH;
        const int pix = c * 8 + r;
        const int py = (pix * 3641) >> 16, px = pix - py * PW;      // pix / 18 for pix < 400
        const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px, ch = s * 64 + ls * 8;
#ifdef RW_EXP_NO_DMA
        const bool ok = false;
#else
        const bool ok = t.live && pix < PW * PH && ch < a.cinv && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#endif
        const char* src = X + ((((long)t.b * a.H + gy) * a.W + gx) * a.ldx + ch) * 2;
        rw_dma_1k(ok ? src : zero, buf + s * RW_SLICE + c * 1024);
      }
    };
    load_tile(tile_at(tile0), lds_base);
    int sel = 0;
    for (int tile = tile0; tile < total; tile += a.ksplit, sel ^= 1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's chunks of `tile` have landed
      __syncthreads();                         // ... and the other I/O wave's; the compute waves are done with buffer sel^1
      load_tile(tile_at(tile + a.ksplit), lds_base + (sel ^ 1) * RW_BUF);
    }
  } else {
    // ================================================================== compute role: wave = output-channel tile 6*blk + wave
    const int li = lane & 15, q = lane >> 4;
    const int cot = blk * 6 + wave;
    const bool active = cot * 16 < a.n;
    const int nrow = cot * 16 + li;                  // A rows: this lane's weight row
    const int c4 = cot * 16 + q * 4;                 // D rows: the 4 output channels this lane stores
    uint4 wf[9][KC];
    {
      const T* Wp = reinterpret_cast<const T*>(a.wp);
      const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int k0 = kc * 32 + q * 8;
          const bool ok = active && nrow < a.n_pad && k0 < a.k_pad;
          wf[t][kc] = *reinterpret_cast<const uint4*>(ok ? Wp + ((long)t * a.n_pad + nrow) * a.k_pad + k0 : zw);
        }
    }
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (a.bias && active && c4 + e < a.nbias) ? a.bias[c4 + e] : 0.f;
    // fragment addresses (32-bit LDS offsets of the CURRENT buffer, flipped by +-RW_BUF per tile): pixel C + li (C = row*18 + dx, a compile-time
    // constant), K chunk kc = slice kc >> 1, slot (kc & 1)*4 + q  ->  (C + li)*128 + ((slot ^ ((C + li) & 7)) << 4) = d0[C & 7] ^ ((kc & 1) << 6), + C*128
    unsigned d0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] = lds_base + li * DD_LDS_ROW + ((q ^ ((li + c) & 7)) << 4);
    T* __restrict__ Y = reinterpret_cast<T*>(a.y);
    const T* __restrict__ M = reinterpret_cast<const T*>(a.mask);
    const bool ch_ok = active && c4 < a.n;
    const long yrow = (long)a.W * a.ldy, mrow = (long)a.W * a.ldmask;

    int sel = 0;
    for (int tile = tile0; tile < total; tile += a.ksplit, sel ^= 1) {
      __syncthreads();      // the I/O waves' DMA of `tile` has landed (they wait for it before this barrier)
      if (!active) continue;
      const RwTile tc = tile_at(tile);
      const bool col_ok = ch_ok && tc.x0 + li < a.W;
      const long pix0 = ((long)tc.b * a.H + tc.y0) * a.W + tc.x0 + li;
      T* yp = Y + pix0 * a.ldy + c4;
      const T* mp = M + pix0 * a.ldmask + c4;
      f32x4_t acc[4];      // output rows y % 4: row y is complete after haloed row y + 2, written during haloed row y + 3, re-used by row y + 4
      uint2 oldv[4], mv[4];
      constexpr int FR = 3 * KC, NF = PH * FR, RING = 6, AHEAD = RING - 1;      // fragments per haloed row / per tile
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / FR, j = f - FR * yy, dx = j / KC, kc = j - KC * dx, C = yy * PW + dx;
        return rw_lds16((d0[C & 7] ^ ((kc & 1) << 6)) + (kc >> 1) * RW_SLICE + C * DD_LDS_ROW);
      };
      auto write_row = [&](int y) {      // ReLU / mask / accumulate, round, store output row y
        f32x4_t v = acc[y % 4];
        if (a.relu && AUX != 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        uint2 o2;
        o2.x = pack2<T>(v[0], v[1]);
        o2.y = pack2<T>(v[2], v[3]);
        if (AUX == 1) { o2.x = mask_bf16x2_cmp(o2.x, mv[y % 4].x); o2.y = mask_bf16x2_cmp(o2.y, mv[y % 4].y); }      // (not the packed form: dd_common.h)
        if (AUX == 2) {      // the conv result is rounded where the layer-wise path stores it, then the residual is added and the activation applied
          float f8[8], g8[8];
          unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
          unpack8t<T>(uint4{mv[y % 4].x, mv[y % 4].y, 0u, 0u}, g8);
          o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
          o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
          if (a.relu) { o2.x = relu_bf16x2(o2.x); o2.y = relu_bf16x2(o2.y); }
        }
        if (ACCUM) {
          float f8[8], g8[8];
          unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
          unpack8t<T>(uint4{oldv[y % 4].x, oldv[y % 4].y, 0u, 0u}, g8);
          o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
          o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
        }
#ifdef RW_EXP_NO_STORE
        if (col_ok && tc.y0 + y < a.H && o2.x == 0x12345678u) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;
#else
        if (col_ok && tc.y0 + y < a.H) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;
#endif
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PH; ++yy) {
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          const int f = yy * FR + j, dx = j / KC, kc = j - KC * dx;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < RW_TH) {
            acc[yy % 4] = f32x4_t{bv[0], bv[1], bv[2], bv[3]};
            const bool ok = col_ok && tc.y0 + yy < a.H;
            mv[yy % 4] = uint2{0u, 0u}; oldv[yy % 4] = uint2{0u, 0u};
            if (MASK && ok) mv[yy % 4] = *reinterpret_cast<const uint2*>(mp + yy * mrow);
            if (ACCUM && ok) oldv[yy % 4] = *reinterpret_cast<const uint2*>(yp + yy * yrow);
          }
          if (j == 2 && yy >= 3) write_row(yy - 3);      // (under this row's MFMAs)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < RW_TH) acc[y % 4] = mma16<T>(wf[dy * 3 + dx][kc], ring[f % RING], acc[y % 4]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      write_row(RW_TH - 1);
      const int flip = sel ? -RW_BUF : RW_BUF;
#pragma unroll
      for (int c = 0; c < 8; ++c) d0[c] += flip;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------ K <= 64, forward
// The plain forward (bias, ReLU; no mask, no accumulate, so the waves issue no vector loads of their own) of a layer with <= 64 input channels:
// 16 x 16 tiles with an 18 x 18 haloed single-slice image (41 KiB x 2 buffers), all 8 waves compute -- wave = one of 4 output-channel tiles x
// the upper or lower 8 output rows -- and each issues its share of the next tile's 41 DMA chunks between its fragment steps.  Output channels
// beyond 64 run as further blocks.  This is the data-gradient role of csrc/dd_conv_bwd.hip used as a forward kernel, on all 8 waves.
// KC = 2: <= 64 input channels, one 64-channel slice, 16 x 16 tiles (18 x 18 haloed: 41 chunks);  KC = 4: <= 128 input channels, two slices,
// 16 x 8 tiles (18 x 10 haloed: 2 x 23 chunks) -- either way <= 46 KiB per buffer
#ifndef RW8_DMA_SPAN
#define RW8_DMA_SPAN 2      // eighths of a tile's fragment steps over which the next tile's DMA pieces are issued (round 4: 8 -> 2, the 64-channel forward 150 -> 125 us: a piece issued late in the tile is still in flight at the tile barrier; 0 = all at once and 1 measured slightly worse than 2 - 3)
#endif
template <int KC> struct RfGeo {
  static constexpr int NS = (KC + 1) / 2, TH = NS == 1 ? DD_TILE : DD_TILE / 2, PW = DD_TILE + 2, PH = TH + 2;
  static constexpr int CH = (PW * PH + 7) / 8, SLICE = CH * 1024, BUF = NS * SLICE, NCHUNK = NS * CH;
};

// NW = waves per workgroup: 8 (blocks of 4 output-channel tiles = 64 channels) or 12 (3 per SIMD, <= 168 registers: blocks of 6 tiles = 96
// channels -- the balanced form of the 96-channel forward, whose 6 + 2 wave kernel above loads two SIMDs with two compute waves and two with one)
// AUX != 0 (round 4): an output-shaped operand next to the results, on the same waves -- 1: the ReLU-backward data gradient (y = conv(x) * (aux > 0),
// no bias), 2: a residual (y = act(round(conv(x) + bias) + aux): the second half of a conv over a channel concat).  The operand's tile (16 x TH
// pixels of the output block's channels, 64-channel slices of 1 KiB chunks like the image) is fetched by LDS-DMA next to the image, one tile
// ahead, and an output row reads its 8 bytes from LDS: the compute waves still issue no vector loads of their own, so their vmcnt waits for
// nothing but the DMA at the tile barrier (the 6 + 2 wave kernel above parks on its mask loads half of the time: profiles/r04_a_sq_table.txt).
// Measured (cfg-2 step, same box): 96 -> 96 masked data gradient at 64 x 64, B = 128: 95 - 103 -> 82 us; 96 -> 192: 206 -> 159 us.
template <typename T, int KC, int NW, int AUX = 0>
__global__ __launch_bounds__(NW * 64) void conv_rw8_kernel(const RwP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  using G = RfGeo<KC>;
  constexpr bool MASKED = AUX != 0;      // (an aux tile is staged; AUX says what an output row does with it)
  constexpr int CT = NW / 2, NPIECE = (G::NCHUNK + NW - 1) / NW;      // output-channel tiles per block; DMA pieces per wave and tile
  constexpr int MCH = DD_TILE * G::TH / 8, MSL = (CT * 16 + 63) / 64;      // mask tile: chunks of 8 pixels per slice, 64-channel slices
  constexpr int MBUF = MSL * MCH * 1024, NMP = MASKED ? (MSL * MCH + NW - 1) / NW : 0;
  constexpr int PW = G::PW, RH = G::TH / 2, PHW = RH + 2;      // a wave's RH output rows need RH + 2 haloed rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int blk = blockIdx.x / a.ksplit, ks = blockIdx.x - blk * a.ksplit;
  const int per_img = a.tiles_x * a.tiles_y, total = a.B * per_img;
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
  const int tile0 = (ks % xcd_n) * (a.ksplit / xcd_n) + ks / xcd_n;
  auto tile_at = [&](int tile) {
    RwTile t;
    t.live = tile < total;
    const int u = t.live ? tile : 0;
    t.b = u / per_img;
    const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
    t.y0 = ty * G::TH; t.x0 = (rem - ty * a.tiles_x) * DD_TILE;
    return t;
  };
  // ---- DMA: chunk id = k*8 + wave (< NS * CH): slice id / CH, pixels (id % CH)*8 + r of the haloed tile
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* X = reinterpret_cast<const char*>(a.x);
  // HOIST (8 waves: 152 / 222 registers, room for 12 more): this lane's byte offset inside the haloed tile and its (row, column) there are the same
  // for every tile -- computed once, a piece costs a bounds check and one 64-bit add on a per-tile SCALAR base.  Recomputed per piece (the
  // 12-wave build at 166 of its 168 registers; csrc/dd_conv_bwd.hip explains the opaque copy) it is ~35 vector instructions, five of them
  // quarter-rate 32-bit multiplies and two 64-bit multiply-adds: 31 + 13 of those per tile and wave, a VALU load of the order of the MFMAs'.
  constexpr bool HOIST = NW == 8;
  int p_off[NPIECE], p_yx[NPIECE];
  if constexpr (HOIST) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int id = k * NW + wave;
      const int sl = id >= G::CH ? 1 : 0, c = id - sl * G::CH;
      const int pix = c * 8 + r;
      const int py = (pix * 3641) >> 16, px = pix - py * PW;
      const int ch = sl * 64 + ls * 8;
      p_off[k] = ((py * a.W + px) * a.ldx + ch) * 2;
      p_yx[k] = (id < G::NCHUNK && ch < a.cinv && pix < PW * G::PH) ? ((py << 8) | px) : (0x7fff << 8);      // row 32 767 of the tile: never inside an image (a sentinel of 255 is in range of a 256-row image)
    }
  }
  auto piece = [&](int k, const RwTile& t, unsigned buf) {
    const int id = k * NW + wave;
    if constexpr (HOIST) {
      if (id < G::NCHUNK) {      // wave-uniform
        const int gy = t.y0 - 1 + (p_yx[k] >> 8), gx = t.x0 - 1 + (p_yx[k] & 255);
#ifdef RW_EXP_NO_DMA
        const bool ok = false;
#else
        const bool ok = t.live && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#endif
        // tile origin (haloed pixel (0, 0)): a wave-uniform 64-bit offset; rows above / left of the image are never dereferenced
        const long base = ((((long)t.b * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * a.ldx) * 2;
        const int sl = id >= G::CH ? 1 : 0, c = id - sl * G::CH;
        rw_dma_1k(ok ? X + base + p_off[k] : zero, buf + sl * G::SLICE + c * 1024);
      }
      return;
    }
    if (id < G::NCHUNK) {      // wave-uniform
      int rr = r;
      asm volatile("" : "+v"(rr));      // (keeps the per-piece coordinates from being hoisted out of the tile loop: see csrc/dd_conv_bwd.hip)
      const int sl = id >= G::CH ? 1 : 0, c = id - sl * G::CH;
      const int pix = c * 8 + rr;
      const int py = (int)(__umul24(pix, 3641) >> 16), px = pix - (int)__umul24(py, PW);
      const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px, ch = sl * 64 + ((lane & 7) ^ rr) * 8;      // (all from the opaque copy: see above)
#ifdef RW_EXP_NO_DMA
      const bool ok = false;      // (knock-out build: every chunk comes from the zero page -- what the kernel costs without its input traffic)
#else
      const bool ok = t.live && ch < a.cinv && pix < PW * G::PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#endif
      // tile origin (haloed pixel (0, 0)) as a wave-uniform 64-bit offset + this lane's 32-bit offset inside the haloed tile from full-rate 24-bit
      // multiply-adds (round 4: the per-lane 64-bit form compiled to five v_mul_lo_u32 and two v_mad_u64_u32 per piece, quarter-rate instructions
      // worth ~200 issue cycles next to 1 700 cycles of MFMAs per tile and wave -- 28 -> 16 vector instructions per piece, and NO measurable change
      // in any launch: the pieces' arithmetic is not on these kernels' critical path).  py <= 10, px <= 17: (py W + px) and 2 ldx fit 24 bits.
      const long base = ((((long)t.b * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * a.ldx) * 2;
      const unsigned off = __umul24(__umul24(py, a.W) + px, a.ldx * 2) + ch * 2;
      rw_dma_1k(ok ? X + base + off : zero, buf + sl * G::SLICE + c * 1024);
    }
  };
  const char* M = reinterpret_cast<const char*>(a.mask);
  auto mpiece = [&](int k, const RwTile& t, unsigned mbuf) {      // chunk id = k*NW + wave (< MSL * MCH): slice id / MCH, pixels (id % MCH)*8 + r of the tile
    const int id = k * NW + wave;
    if (MASKED && id < MSL * MCH) {      // wave-uniform
      int rr = r;
      asm volatile("" : "+v"(rr));
      const int sl = id / MCH, c = id - sl * MCH;
      const int pix = c * 8 + rr, gy = t.y0 + (pix >> 4), gx = t.x0 + (pix & 15);
      const int chl = sl * 64 + ((lane & 7) ^ rr) * 8, ch = blk * (CT * 16) + chl;      // (from the opaque copy: nothing here is hoisted out of the tile loop)
      const bool ok = t.live && chl < CT * 16 && ch < a.n && gy < a.H && gx < a.W;
      const long base = ((((long)t.b * a.H + t.y0) * a.W + t.x0) * a.ldmask) * 2;      // (wave-uniform; the lane's part in 24-bit multiply-adds, see piece)
      const unsigned off = __umul24(__umul24(pix >> 4, a.W) + (pix & 15), a.ldmask * 2) + ch * 2;
      rw_dma_1k(ok ? M + base + off : zero, mbuf + id * 1024);
    }
  };
  const unsigned mask_base = lds_base + 2 * G::BUF;
  RwTile cur = tile_at(tile0);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) piece(k, cur, lds_base);
#pragma unroll
  for (int k = 0; k < NMP; ++k) mpiece(k, cur, mask_base);

  // ---- compute: output-channel tile (wave & 3) of block blk, output rows RH*(wave >> 2) .. + RH - 1
  const int li = lane & 15, q = lane >> 4;
  const int cot = blk * CT + wave % CT, half = wave / CT;
  const bool active = cot * 16 < a.n;
  const int nrow = cot * 16 + li, c4 = cot * 16 + q * 4;
  uint4 wf[9][KC];
  {
    const T* Wp = reinterpret_cast<const T*>(a.wp);
    const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int k0 = kc * 32 + q * 8;
        const bool ok = active && nrow < a.n_pad && k0 < a.k_pad;
        wf[t][kc] = *reinterpret_cast<const uint4*>(ok ? Wp + ((long)t * a.n_pad + nrow) * a.k_pad + k0 : zw);
      }
  }
  float bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (AUX != 1 && a.bias && active && c4 + e < a.nbias) ? a.bias[c4 + e] : 0.f;
  // this lane's 8 bytes of the mask tile: pixel (RH*half + y)*16 + li -> chunk 2*(RH*half + y) + li/8, row li%8; channels (wave % CT)*16 + 4q of the block
  unsigned m_off = 0;
  if (MASKED) {
    const int c4l = (wave % CT) * 16 + q * 4, rr = li & 7;
    m_off = mask_base + ((c4l >> 6) * MCH + half * RH * 2 + (li >> 3)) * 1024 + rr * 128 + (((((c4l & 63) >> 3)) ^ rr) << 4) + ((c4l >> 2) & 1) * 8;
  }
  unsigned d0[8];      // haloed pixel (RH*half + yy)*18 + dx + li: the row offset of the half is folded into the bases
#pragma unroll
  for (int c = 0; c < 8; ++c) d0[c] = lds_base + (half * RH * PW + li) * DD_LDS_ROW + ((q ^ ((half * RH * PW + li + c) & 7)) << 4);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const bool ch_ok = active && c4 < a.n;
  const long yrow = (long)a.W * a.ldy;

  int sel = 0;
  for (int tile = tile0; tile < total; tile += a.ksplit, sel ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's chunks of `tile` have landed (and its stores of the previous one)
    __syncthreads();
    const RwTile nxt = tile_at(tile + a.ksplit);
    const unsigned nbuf = lds_base + (sel ^ 1) * G::BUF, nmbuf = mask_base + (sel ^ 1) * MBUF;
    if (!active) {
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) piece(k, nxt, nbuf);
#pragma unroll
      for (int k = 0; k < NMP; ++k) mpiece(k, nxt, nmbuf);
      cur = nxt;
      continue;
    }
    const bool col_ok = ch_ok && cur.x0 + li < a.W;
    T* yp = Y + (((long)cur.b * a.H + cur.y0 + half * RH) * a.W + cur.x0 + li) * a.ldy + c4;
    f32x4_t acc[4];
    constexpr int FR = 3 * KC, NF = PHW * FR, RING = NW == 8 ? 6 : 2, AHEAD = RING - 1;      // (12 waves: 168 registers, 108 of them weights)
    // (native vectors, and for RING == 2 two named registers: in the AUX builds an array of HIP uint4 structs was left in scratch memory)
    rw_u32x4 ring[RING], r0 = {0u, 0u, 0u, 0u}, r1 = r0;
    auto slot = [&](int i) -> rw_u32x4& { if constexpr (RING == 2) return (i & 1) ? r1 : r0; else return ring[i % RING]; };
    auto frag = [&](int f) {
      const int yy = f / FR, j = f - FR * yy, dx = j / KC, kc = j - KC * dx, C = yy * PW + dx;
      return *reinterpret_cast<const __attribute__((address_space(3))) rw_u32x4*>((d0[C & 7] ^ ((kc & 1) << 6)) + (kc >> 1) * G::SLICE + C * DD_LDS_ROW);
    };
    uint2 mv = uint2{0u, 0u};
    auto read_mask = [&](int y) {
      const rw_u32x2 m = *reinterpret_cast<const __attribute__((address_space(3))) rw_u32x2*>(m_off + y * 2048);
      mv = uint2{m[0], m[1]};
    };
    auto write_row = [&](int y) {
      f32x4_t v = acc[y % 4];
      uint2 o2;
      o2.x = pack2<T>(v[0], v[1]);
      o2.y = pack2<T>(v[2], v[3]);
      if (AUX == 1) { o2.x = mask_bf16x2_cmp(o2.x, mv.x); o2.y = mask_bf16x2_cmp(o2.y, mv.y); }
      if (AUX == 2) {      // the conv result is rounded where the layer-wise path stores it, then the residual is added (as the 6 + 2 wave kernel does)
        float f8[8], g8[8];
        unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
        unpack8t<T>(uint4{mv.x, mv.y, 0u, 0u}, g8);
        o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
        o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
      }
      if (AUX != 1 && a.relu) { o2.x = relu_bf16x2(o2.x); o2.y = relu_bf16x2(o2.y); }
#ifdef RW_EXP_NO_STORE
      if (col_ok && cur.y0 + half * RH + y < a.H && o2.x == 0x12345678u) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;      // (knock-out build)
#else
      if (col_ok && cur.y0 + half * RH + y < a.H) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;
#endif
    };
#pragma unroll
    for (int f = 0; f < AHEAD; ++f) slot(f) = frag(f);
#pragma unroll
    for (int yy = 0; yy < PHW; ++yy) {
#pragma unroll
      for (int j = 0; j < FR; ++j) {
        const int f = yy * FR + j, dx = j / KC, kc = j - KC * dx;
#ifndef RW_EXP_NO_MMA
        if (f + AHEAD < NF) slot(f + AHEAD) = frag(f + AHEAD);
#endif
        if (j == 0 && yy < RH) acc[yy % 4] = f32x4_t{bv[0], bv[1], bv[2], bv[3]};
        // (MASKED: the mask is read two fragment steps = 6 MFMAs ahead of its use, and both sit on steps that issue no DMA piece -- the piece's
        //  address arithmetic and the mask word together were 4 registers over the 168 of a 12-wave workgroup)
        constexpr int WJ = MASKED ? 6 : 2;
        static_assert(WJ < FR, "the row store needs its fragment step");
        if (MASKED && j == WJ - 2 && yy >= 3) read_mask(yy - 3);
        if (j == WJ && yy >= 3) write_row(yy - 3);
        {      // the DMA pieces of the next tile, spread evenly over the first RW8_DMA_SPAN / 8 of the NF steps (the rest of the tile hides their latency)
          constexpr int NP = NPIECE + NMP;
          if constexpr (RW8_DMA_SPAN == 0) {      // (experiment: all pieces back to back before the first fragment step)
            if (f == 0) {
#pragma unroll
              for (int k = 0; k < NP; ++k) { if (k < NPIECE) piece(k, nxt, nbuf); else mpiece(k - NPIECE, nxt, nmbuf); }
            }
          } else {
            constexpr int SPAN = NF * RW8_DMA_SPAN / 8 > NP ? NF * RW8_DMA_SPAN / 8 : NP;
            const int k0 = (f * NP + SPAN - 1) / SPAN;
            if (k0 < NP && (k0 * SPAN) / NP == f) { if (k0 < NPIECE) piece(k0, nxt, nbuf); else mpiece(k0 - NPIECE, nxt, nmbuf); }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef RW_EXP_NO_MMA      // (knock-out build: the kernel as a pure mover of its input tiles and output rows)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int y = yy - dy;
          if (y >= 0 && y < RH) acc[y % 4] = mma16<T>(wf[dy * 3 + dx][kc], uint4{slot(f)[0], slot(f)[1], slot(f)[2], slot(f)[3]}, acc[y % 4]);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int y = (PHW >= 3 ? PHW - 3 : 0); y < RH; ++y) {      // the rows completed by the last haloed rows
      if (MASKED) read_mask(y);
      write_row(y);
    }
    const int flip = sel ? -G::BUF : G::BUF;
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] += flip;
    if (MASKED) m_off += sel ? -MBUF : MBUF;
    cur = nxt;
  }
}

static int rw_cus() {
  return dd_device_cus();
}

template <typename T, int KC, int AUX, bool ACCUM>
static void rw_launch(const RwP& p, hipStream_t stream) {
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_rw_kernel<T, KC, AUX, ACCUM>));
  hipLaunchKernelGGL((conv_rw_kernel<T, KC, AUX, ACCUM>), dim3((unsigned)(p.nblk * p.ksplit)), dim3(512), 2 * (size_t)RW_BUF, stream, p);
}
template <typename T, int KC>
static void rw_launch_flags(const RwP& p, hipStream_t stream) {
  if (p.aux_residual) rw_launch<T, KC, 2, false>(p, stream);
  else if (p.mask) { if (p.accum) rw_launch<T, KC, 1, true>(p, stream); else rw_launch<T, KC, 1, false>(p, stream); }
  else { if (p.accum) rw_launch<T, KC, 0, true>(p, stream); else rw_launch<T, KC, 0, false>(p, stream); }
}

}  // namespace

// Is this dd_conv_igemm call one the register-weight kernels take?  3x3, bf16 / f16 storage, plain epilogue, and either 65..96 input channels
// (any epilogue of {ReLU, mask, accumulate}) or <= 64 input channels with nothing but bias / ReLU (the forward of the 64-channel level)
bool dd_conv_rw_eligible(const dd_conv_args* a) {
  static int on = -1, on8 = -1;
  if (on < 0) { const char* e = getenv("DD_CONV_RW"); on = e ? atoi(e) : 1; }
  if (on8 < 0) { const char* e = getenv("DD_CONV_RW8"); on8 = e ? atoi(e) : 1; }
  if (!on) return false;
  const int plain = DD_OUT_RELU | DD_ACCUM;
  // a residual operand: only the 65..96-channel kernel, with nothing else in the epilogue (the second half of a conv over a channel concat)
  const bool res_ok = !a->res || (!a->mask && !(a->flags & DD_ACCUM) && a->cin > 64 && a->cin <= 96 && a->ldres % 4 == 0 && ((uintptr_t)a->res % 8) == 0);
  const bool common = a->taps == 9 && (a->dtype == DD_BF16 || a->dtype == DD_F16) && (a->flags & ~plain) == 0 && res_ok && a->k_pad % 32 == 0 &&
                      a->n % 4 == 0 && a->ldx % 8 == 0 && a->ldy % 4 == 0 && (!a->mask || a->ldmask % 4 == 0) && ((uintptr_t)a->x % 16) == 0 &&
                      ((uintptr_t)a->wp % 16) == 0 && ((uintptr_t)a->y % 8) == 0 && (!a->mask || ((uintptr_t)a->mask % 8) == 0);
  if (!common) return false;
  if (a->cin > 64 && a->cin <= 96 && a->k_pad <= 96) return true;
  // forward on all 8 waves: <= 64 input channels, or 97..128 (two slices, weights 144 registers); nothing but bias / ReLU in the epilogue
  // (97..128 input channels: also the ReLU-backward data gradient -- mask tile by LDS-DMA, no bias)
  static int on8m = -1;
  if (on8m < 0) { const char* e = getenv("DD_CONV_RW8_MASK"); on8m = e ? atoi(e) : 1; }
  const bool wide = a->cin > 96 && a->cin <= 128 && a->k_pad <= 128;
  const bool mask_ok = !a->mask || (on8m && wide && !a->res && !a->bias && !(a->flags & DD_OUT_RELU) && a->n % 8 == 0 && a->ldmask % 8 == 0 && ((uintptr_t)a->mask % 16) == 0);
  return on8 && ((a->cin > 16 && a->cin <= 64 && a->k_pad <= 64) || wide) && mask_ok && !(a->flags & DD_ACCUM) && a->n >= 48;
}

template <typename T, int KC, int NW = 8, int AUX = 0>
static void rw8_launch(const RwP& p, hipStream_t stream) {
  constexpr size_t mask_lds = AUX ? 2 * (size_t)((NW / 2 * 16 + 63) / 64) * (DD_TILE * RfGeo<KC>::TH / 8) * 1024 : 0;
  static_assert(2 * (size_t)RfGeo<KC>::BUF + mask_lds <= 160 * 1024, "LDS");
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_rw8_kernel<T, KC, NW, AUX>));
  hipLaunchKernelGGL((conv_rw8_kernel<T, KC, NW, AUX>), dim3((unsigned)(p.nblk * p.ksplit)), dim3(NW * 64), 2 * (size_t)RfGeo<KC>::BUF + mask_lds, stream, p);
}

int dd_conv_rw_launch(const dd_conv_args* a, hipStream_t stream) {
  RwP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.mask = a->res ? a->res : a->mask; p.y = a->y;
  p.ldx = a->ldx; p.ldmask = a->res ? a->ldres : a->ldmask; p.ldy = a->ldy; p.aux_residual = a->res != nullptr;
  p.cin = a->cin; p.cinv = (a->cin + 7) / 8 * 8; p.n = a->n; p.n_pad = a->n_pad; p.k_pad = a->k_pad; p.nbias = a->nbias;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.relu = (a->flags & DD_OUT_RELU) != 0; p.accum = (a->flags & DD_ACCUM) != 0;
  static int on12 = -1;
  if (on12 < 0) { const char* e = getenv("DD_CONV_RW12"); on12 = e ? atoi(e) : 1; }
  const bool plain_fwd = !a->mask && !a->res && !(a->flags & DD_ACCUM);
  static int on12m = -1;
  if (on12m < 0) { const char* e = getenv("DD_CONV_RW12_MASK"); on12m = e ? atoi(e) : 1; }
  // the ReLU-backward data gradient on all 12 waves: mask tile by LDS-DMA (16-byte pieces: 8-channel groups of the mask rows must be aligned)
  const bool masked12 = on12 && on12m && a->cin > 64 && a->cin <= 96 && a->mask && !a->res && !a->bias && !(a->flags & (DD_ACCUM | DD_OUT_RELU)) &&
                        a->n >= 48 && a->n % 8 == 0 && a->ldmask % 8 == 0 && ((uintptr_t)a->mask % 16) == 0;
  // ... and the second half of a conv over a channel concat (residual operand, bias / ReLU)
  const bool res12 = on12 && on12m && a->cin > 64 && a->cin <= 96 && a->res && !a->mask && !(a->flags & DD_ACCUM) && a->n >= 48 && a->n % 8 == 0 &&
                     a->ldres % 8 == 0 && ((uintptr_t)a->res % 16) == 0;
  const bool twelve = on12 && a->cin > 64 && a->cin <= 96 && (plain_fwd || masked12 || res12) && a->n >= 48;      // all 12 waves compute
  const bool six = a->cin > 64 && a->cin <= 96 && !twelve;      // 6 compute + 2 I/O waves, any epilogue; else all waves compute, forward only
  const int th = six ? RW_TH : (a->cin <= 64 ? RfGeo<2>::TH : RfGeo<4>::TH);      // (RfGeo<3>::TH == RfGeo<4>::TH == 8)
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, th);
  p.nblk = dd_ceil_div(a->n, (six || twelve) ? 96 : 64);
  const long total = (long)a->B * p.tiles_x * p.tiles_y;
  long ksplit = rw_cus() / p.nblk;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > total) ksplit = total;
  p.ksplit = (int)ksplit;
  if (masked12) {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 3, 12, 1>(p, stream); else rw8_launch<f16_t, 3, 12, 1>(p, stream);
  } else if (res12) {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 3, 12, 2>(p, stream); else rw8_launch<f16_t, 3, 12, 2>(p, stream);
  } else if (twelve) {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 3, 12>(p, stream); else rw8_launch<f16_t, 3, 12>(p, stream);
  } else if (six) {
    if (a->dtype == DD_BF16) rw_launch_flags<bf16_t, 3>(p, stream); else rw_launch_flags<f16_t, 3>(p, stream);
  } else if (a->cin <= 64) {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 2>(p, stream); else rw8_launch<f16_t, 2>(p, stream);
  } else if (a->mask) {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 4, 8, 1>(p, stream); else rw8_launch<f16_t, 4, 8, 1>(p, stream);
  } else {
    if (a->dtype == DD_BF16) rw8_launch<bf16_t, 4>(p, stream); else rw8_launch<f16_t, 4>(p, stream);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}
