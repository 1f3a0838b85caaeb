// Fused backward of a 3x3 SAME convolution with 65 - 96 OUTPUT channels on CDNA4 (round 6): data gradient AND weight gradient of one layer
// from one pass over dy and x -- the 64 x 64 level of the U-Net (UNet.py:25-36 under Training.py:701-702; 96 -> 96, 192 -> 96, 64 -> 96).
//
//   dx[p][ci]       = (x[p][ci] > 0) * sum_{t, co} Wd[t][ci][co] * dy[p + off(t)][co]            (+= an existing gradient when the tensor has two consumers)
//   dW[t][ci][co]   = sum_q x[q][ci] * dy[q - off(t)][co]
//   db[co]          = sum_p dy[p][co]
//
// csrc/dd_conv_bwd.hip holds dW of a 64 x 64 channel block in the registers of four waves (144 accumulators each); 96 x 96 does not fit, and
// until round 5 these layers ran as two launches (weight-gradient role per 64 x 64 block pair + the register-weight data gradient), each with a
// single role per SIMD.  Here a workgroup owns a THIRD of the input channels (32) against ALL output channels:
//   * dW[9][32 ci][96 co] = 108 accumulator tiles = 27 per weight-gradient wave (108 registers);
//   * dx[32 ci] needs Wd[9][32 ci][96 co]: 27 fragments = 108 registers per data-gradient wave (input-channel tile x tile half);
//   * the haloed dy tile (18 x 18 x 96 channels) is staged once per tile as three 32-channel images of 64-byte pixel rows, the x tile
//     (16 x 16 x 32) as a fourth: 79 KiB per buffer, double buffered by LDS-DMA (global_load_lds_dwordx4) like the 64-channel kernel.
// dy is read once per input-channel third (cache-resident at this level), x once, dx written once.
//
// Roles (8 waves; a SIMD hosts one of each, 216 MFMAs per tile and wave):
//   waves 0-3  data gradient: wave = (input-channel tile, output rows 8h .. 8h+7); weights are the register-resident A operand, dy pixels B;
//              walks its 10 haloed rows once (3 column shifts x 3 K-chunks of 32 channels), a row's fragment feeds the three output rows
//              it touches; rows are masked with x (from the LDS image), rounded and stored from the accumulators.  These waves issue ALL DMA.
//   waves 4-7  weight gradient: wave = (input-channel tile, output-channel tiles of one parity: 3 of the 6).  Both operands are read
//              transposed (ds_read_b64_tr_b16; the reduction runs over pixels, 2 tile rows = 32 pixels per MFMA).  Per pair of tile rows
//              (2s, 2s+1) the taps of kernel row ty need the haloed dy rows (2s+ty, 2s+ty+1): ty = 0 and ty = 2 are ROW-PAIR ALIGNED
//              fragments -- and ty = 2 of pair s is ty = 0 of pair s+1, so each is read once and kept; ty = 1 straddles two aligned
//              fragments: its lanes 0-31 (first row of a pair) take the NEW fragment's first row (2s+2), lanes 32-63 the OLD fragment's
//              second row (2s+1) -- a lane-wise select, no lane crossing -- multiplied with the x fragment whose rows are swapped
//              (2s+1 | 2s): the pairing of x pixel and dy pixel per K slot is what matters, not the K order.  11 transposed reads per
//              27 MFMAs instead of 28.
// LDS image: pixel p of a 32-channel image at p*64, 16-byte slot s at physical slot s ^ (((p >> 2) & 1) << 1): the 16 lanes of a
// ds_read_b128 service group (pixels C+{0..3,12..15} of one slot, C+{4..11} of the next) then cover 16 distinct slots of the 256-byte bank row
// for any C (brute-forced: the only XOR keys that do are 2*(bit 2 of p) and its complements).
#include "dd_common.h"
#include <type_traits>

#ifndef B96_RING
#define B96_RING 6          // data-gradient role: ring of dy fragments (RING - 1 steps ahead)
#endif
#ifndef B96_BA
#define B96_BA 2            // weight-gradient role: new aligned dy fragments are requested this many steps ahead
#endif
#ifndef B96_DMA_SPAN
#define B96_DMA_SPAN 8      // eighths of a tile's fragment steps over which the data-gradient waves issue the next tile's DMA pieces
#endif

namespace {

struct Bw96P {
  const void* dy; const void* x; const void* wd; void* dx; float* dw; float* db;
  int lddy, ldx, lddx;
  int cout, cin, coutv, cinv;      // logical / staged (rounded up to 8) channel counts
  int n_pad, k_pad;                // packed data-gradient weights [9][n_pad (ci)][k_pad (co)]
  int B, H, W, tiles_x, tiles_y;
  int nblk, ksplit;                // 32-channel blocks of C_in; workgroups per block
};

typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr96;
typedef uint32_t b96_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t b96_u32x2 __attribute__((ext_vector_type(2)));

constexpr int B96_PW = DD_TILE + 2;                    // haloed tile width
constexpr int B96_ROW = 64;                            // bytes of a pixel row of a 32-channel image
constexpr int B96_SUBCH = (B96_PW * B96_PW + 15) / 16; // 21 chunks of 16 pixels (1 KiB) per 32-channel dy image
constexpr int B96_PSUB = B96_SUBCH * 1024;             // 21 504
constexpr int B96_NSUB = 3;                            // 96 output channels
constexpr int B96_PCH = B96_NSUB * B96_SUBCH;          // 63 dy chunks
constexpr int B96_P_BYTES = B96_PCH * 1024;            // 64 512
constexpr int B96_QCH = DD_TILE;                       // 16 x chunks (one tile row each)
constexpr int B96_Q_BYTES = B96_QCH * 1024;
constexpr int B96_BUF = B96_P_BYTES + B96_Q_BYTES;     // 80 896; two buffers = 161 792 <= 163 840

__device__ __forceinline__ void b96_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 b96_lds16(unsigned off) {
  const b96_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) b96_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ uint2 b96_lds8(unsigned off) {
  const b96_u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) b96_u32x2*>(off);
  return uint2{v[0], v[1]};
}
__device__ __forceinline__ uint4 b96_tr_pair(unsigned a0, unsigned a1) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_ptr96>(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_ptr96>(a1));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return uint4{l2.x, l2.y, h2.x, h2.y};
}
// acc += A * B on ONE register tuple (see csrc/dd_conv_bwd.hip: through the builtin hipcc renames the accumulator tuples and spills).  No
// hazard recogniser sees inside the asm: accumulators are read only after the tile loop (behind s_nops); operands written by LDS reads are
// waited for by the compiler's s_waitcnt; operands written by VALU (the ty = 1 select) are consumed five MFMAs later, behind a sched_barrier.
template <typename T> __device__ __forceinline__ void b96_mma_inplace(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void b96_mma_inplace<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const b96_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}
template <> __device__ __forceinline__ void b96_mma_inplace<f16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const b96_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}

// compile-time loop: the 90 / 72 fragment steps of a tile must be straight-line code (ring slots and accumulators are indexed by the step; hipcc
// gives up on `#pragma unroll` for bodies of this size and then keeps the rings in scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void b96_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    b96_static_for<I + 1, N>(f);
  }
}

template <typename T, bool MASK, bool ACCUM>
__global__ __launch_bounds__(512) void conv_bwd96_kernel(const Bw96P a) {
  static_assert(sizeof(T) == 2, "fused conv backward: bf16 / fp16 storage");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int PW = B96_PW, ROW = B96_ROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (input block cb, tile sequence ks).  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2.  The
  // nblk workgroups of ONE tile sequence read the same haloed dy tiles at the same time: they sit on one XCD (consecutive slots of it), so dy comes
  // out of HBM / the infinity cache once per sequence, not once per input block (192 -> 96: six blocks; measured 393 -> see DESIGN 8.1).  Sequences
  // beyond the last multiple of 8 (85 = 80 + 5 for three blocks on 256 CUs) fill the remaining workgroups in plain order.  Iteration i of a sequence
  // covers tile i*ksplit + tile0; the sequences of one XCD take a contiguous run of tiles (neighbouring tiles share halo rows and columns).
  const int block = blockIdx.x;
  const int ks_al = a.ksplit & ~7, n_al = a.nblk * ks_al;
  int cb, tile0;
  if (block < n_al) {
    const int xcd = block & 7, slot = block >> 3, m = slot / a.nblk;
    cb = slot - m * a.nblk;
    tile0 = xcd * (ks_al >> 3) + m;
  } else {
    const int r = block - n_al, m = r / a.nblk;
    cb = r - m * a.nblk;
    tile0 = ks_al + m;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int per_img = a.tiles_y * a.tiles_x;
  const int total_tiles = a.B * per_img;

#ifdef B96_EXP_STAMP      // (experiment build: cycles a wave spends in the tile-top wait / barrier, printed for two workgroups)
  long long stamp_vm = 0, stamp_bar = 0, stamp_n = 0;
  const long long stamp_t0 = __builtin_readcyclecounter();
  const long long stamp_r0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz: the ratio to the cycle counter is the shader clock
#endif
  if (wave < 4) {
    // ================================================================== data-gradient role: wave = (input-channel tile cit, tile half)
    const int wr = wave, cit = wave & 1, half = wave >> 1;
    constexpr int QPC = B96_QCH / 4, PPC = (B96_PCH + 3) / 4, NP = QPC + PPC;      // 4 + 16 DMA pieces per wave and tile
    const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    // a lane's part of a 1-KiB chunk: pixel r of the chunk's 16, physical slot lane & 3 = logical slot ls of that pixel
    const int r = lane >> 2, ls = (lane & 3) ^ (((r >> 2) & 1) << 1);
    const int xch = cb * 32 + ls * 8;
    const bool x_ok = xch < a.cinv;
    const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
    const int x_row = a.W * a.ldx * 2, x_lane = (r * a.ldx + xch) * 2;                  // bytes
    const int dy_row = a.W * a.lddy * 2, dy_pix = a.lddy * 2;
    struct Origin { const char* q; const char* p; int b, y0, x0; bool live; };
    auto origin = [&](int tile) {
      Origin o;
      o.live = tile < total_tiles;
      const int t = o.live ? tile : 0;
      o.b = t / per_img;
      const int rem = t - o.b * per_img, ty = rem / a.tiles_x;
      o.y0 = ty * DD_TILE; o.x0 = (rem - ty * a.tiles_x) * DD_TILE;
      o.q = reinterpret_cast<const char*>(X + ((long)o.b * a.H * a.W + (long)o.y0 * a.W + o.x0) * a.ldx);
      o.p = reinterpret_cast<const char*>(DY + ((long)o.b * a.H * a.W + (long)(o.y0 - 1) * a.W + (o.x0 - 1)) * a.lddy);
      return o;
    };
    auto piece = [&](int k, const Origin& o, int sel) {      // DMA piece k of the tile at `o` into buffer `sel`: k < QPC -> x chunk, else dy chunk
      const unsigned buf = lds_base + sel * B96_BUF;
      int rr = r, xl = x_lane, lsv = ls;      // (opaque copies: keeps hipcc from hoisting every piece's coordinates out of the tile loop, csrc/dd_conv_bwd.hip)
      asm volatile("" : "+v"(rr), "+v"(xl), "+v"(lsv));
      if (k < QPC) {
        const int c = wr * QPC + k;            // tile row c, its 16 pixels
#ifdef B96_EXP_NO_DMA
        const bool ok = false;      // (knock-out builds, tools/b96_knockouts.sh: every piece comes from the zero page)
#else
        const bool ok = o.live && x_ok && o.y0 + c < a.H && o.x0 + rr < a.W;
#endif
#ifdef B96_EXP_CONTIG      // (knock-out build: the same bytes per tile as 1-KiB contiguous reads -- 8 cache lines per instruction instead of 16 half lines)
        b96_dma_1k(reinterpret_cast<const char*>(X) + ((((long)(o.b * per_img) + c) * 1024 + lane * 16) % ((long)a.B * a.H * a.W * a.ldx * 2 - 1024)), buf + B96_P_BYTES + c * 1024);
#else
        b96_dma_1k(ok ? o.q + (c * x_row + xl) : zero, buf + B96_P_BYTES + c * 1024);
#endif
      } else {
        const int c = (k - QPC) * 4 + wr;      // chunk c of the 63: image kc = c / 21, pixels (c % 21)*16 + r of the 18 x 18 tile
        if (c < B96_PCH) {                     // wave-uniform
          const int kc = c / B96_SUBCH, cc = c - kc * B96_SUBCH;
          const int pix = cc * 16 + rr;
          const int py = (pix * 3641) >> 16, px = pix - py * PW;          // pix / 18 for pix < 400
          const int gy = o.y0 - 1 + py, gx = o.x0 - 1 + px;
          const int dch = kc * 32 + lsv * 8;
#ifdef B96_EXP_NO_DMA
          const bool ok = false;
#else
          const bool ok = o.live && dch < a.coutv && pix < PW * PW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#endif
#ifdef B96_EXP_CONTIG
          b96_dma_1k(reinterpret_cast<const char*>(DY) + (((long)(o.y0 * a.W + o.x0 + (long)o.b * a.H * a.W) * a.lddy * 2 + c * 1024 + lane * 16) % ((long)a.B * a.H * a.W * a.lddy * 2 - 1024)), buf + c * 1024);
#else
          b96_dma_1k(ok ? o.p + (py * dy_row + px * dy_pix + dch * 2) : zero, buf + c * 1024);
#endif
        }
      }
    };
    Origin oc = origin(tile0);
#pragma unroll
    for (int k = 0; k < NP; ++k) piece(k, oc, 0);

    const int li = lane & 15, q = lane >> 4;
    const int ci_row = cb * 32 + cit * 16 + li;                  // A rows: this lane's weight row
    const int c4 = cb * 32 + cit * 16 + q * 4;                   // D rows: the 4 input channels this lane stores
#ifdef B96_EXP_NO_DROLE
    const bool active = false;
#else
    const bool active = a.dx != nullptr && cb * 32 + cit * 16 < a.cin;
#endif
    uint4 wf[9][B96_NSUB];
    {
      const T* Wd = reinterpret_cast<const T*>(a.wd);
      const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < B96_NSUB; ++kc) {
          const int k0 = kc * 32 + q * 8;
          const bool ok = active && ci_row < a.n_pad && k0 < a.k_pad;
          wf[t][kc] = *reinterpret_cast<const uint4*>(ok ? Wd + ((long)t * a.n_pad + ci_row) * a.k_pad + k0 : zw);
        }
    }
    // Fragment addresses (32-bit LDS offsets of the CURRENT buffer; they flip by +-B96_BUF per tile).  dy image kc: pixel C + li (C = row*18 + dx
    // relative to this half's first haloed row, a compile-time constant), slot q -> (C + li)*64 + ((q ^ key(C + li)) << 4) = d0[C & 7] + C*64.
    unsigned d0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] = lds_base + (half * 8 * PW + li) * ROW + ((q ^ ((((c + li) >> 2) & 1) << 1)) << 4);
    // x image: pixel (row, li), channels cit*16 + q*4 .. +3 (8 bytes)
    unsigned mrow = lds_base + B96_P_BYTES + (half * 8 * DD_TILE + li) * ROW + (((cit * 2 + (q >> 1)) ^ (((li >> 2) & 1) << 1)) << 4) + (q & 1) * 8;
    T* __restrict__ DX = reinterpret_cast<T*>(a.dx);
    const bool ch_ok = c4 < a.cinv;
    const long row_stride = (long)a.W * a.lddx;

    int sel = 0;
    for (int tile = tile0; tile < total_tiles; tile += a.ksplit, sel ^= 1) {
#ifdef B96_EXP_STAMP
      const long long st0 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of `tile` have landed
#ifdef B96_EXP_STAMP
      const long long st1 = __builtin_readcyclecounter();
#endif
      __syncthreads();                      // ... and everyone's; buffer sel^1 is free
#ifdef B96_EXP_STAMP
      stamp_vm += st1 - st0; stamp_bar += __builtin_readcyclecounter() - st1; ++stamp_n;
#endif
      const Origin on = origin(tile + a.ksplit);
      if (!active) {
#pragma unroll
        for (int k = 0; k < NP; ++k) piece(k, on, sel ^ 1);
        oc = on;
        continue;
      }
      const bool col_ok = ch_ok && oc.x0 + li < a.W;
      const int ybase = oc.y0 + half * 8;
      T* dxp = DX + ((long)oc.b * a.H * a.W + (long)ybase * a.W + oc.x0 + li) * a.lddx + c4;
      f32x4_t acc[4];      // output rows y % 4 of this half: row y is complete after haloed row y + 2, written during haloed row y + 3
      uint2 oldv[4], mv[4];
      // 90 fragment steps = 10 haloed rows x 3 column shifts x 3 K-chunks, up to 3 MFMAs each
      constexpr int RING = B96_RING, AHEAD = RING - 1, HR = DD_TILE / 2 + 2, NF = HR * 9;
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / 9, j = f - 9 * yy, C = yy * PW + j / 3;
        return b96_lds16(d0[C & 7] + C * ROW + (j % 3) * B96_PSUB);
      };
      auto write_row = [&](int y) {      // mask, round, (accumulate,) store output row y of this half
        const f32x4_t v = acc[y % 4];
        uint2 o2;
        o2.x = pack2<T>(v[0], v[1]);
        o2.y = pack2<T>(v[2], v[3]);
        if (MASK) { o2.x = mask_bf16x2(o2.x, mv[y % 4].x); o2.y = mask_bf16x2(o2.y, mv[y % 4].y); }
        if (ACCUM) {
          float f8[8], g8[8];
          unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
          unpack8t<T>(uint4{oldv[y % 4].x, oldv[y % 4].y, 0u, 0u}, g8);
          o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
          o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
        }
#ifdef B96_EXP_NO_STORE
        if (col_ok && ybase + y < a.H && o2.x == 0x12345678u) *reinterpret_cast<uint2*>(dxp + y * row_stride) = o2;
#else
        if (col_ok && ybase + y < a.H) *reinterpret_cast<uint2*>(dxp + y * row_stride) = o2;
#endif
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
      b96_static_for<0, NF>([&](auto fc) {
        constexpr int f = decltype(fc)::value, yy = f / 9, j = f - 9 * yy, dx = j / 3, kc = j % 3;
        if constexpr (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
        if constexpr (j == 0 && yy < DD_TILE / 2) {
          acc[yy % 4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (MASK) mv[yy % 4] = b96_lds8(mrow + yy * DD_TILE * ROW);
          oldv[yy % 4] = uint2{0u, 0u};
          if (ACCUM && col_ok && ybase + yy < a.H) oldv[yy % 4] = *reinterpret_cast<const uint2*>(dxp + yy * row_stride);
        }
        if constexpr (j == 3 && yy >= 3) write_row(yy - 3);      // (under this row's MFMAs)
        {      // the NP DMA pieces of the next tile, spread evenly over the first B96_DMA_SPAN / 8 of the NF steps (piece k at step k*SPAN/NP)
          constexpr int SPAN = NF * B96_DMA_SPAN / 8 > NP ? NF * B96_DMA_SPAN / 8 : NP;
          constexpr int k0 = (f * NP + SPAN - 1) / SPAN;
          if constexpr (k0 < NP && (k0 * SPAN) / NP == f) piece(k0, on, sel ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int y = yy - dy;
          if (y >= 0 && y < DD_TILE / 2) acc[y % 4] = mma16<T>(wf[dy * 3 + dx][kc], ring[f % RING], acc[y % 4]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      write_row(DD_TILE / 2 - 1);
      const int flip = sel ? -B96_BUF : B96_BUF;
#pragma unroll
      for (int c = 0; c < 8; ++c) d0[c] += flip;
      mrow += flip;
      oc = on;
    }
  } else {
    // ================================================================== weight-gradient role: wave = (input-channel tile cit, output-channel tiles 2j + cg)
    const int w = wave - 4, cit = w & 1, cg = w >> 1;
    const int li = lane & 15, q4 = (lane >> 4) * 4;
#ifdef B96_EXP_NO_WROLE
    const bool active = false;
#else
    const bool active = cb * 32 + cit * 16 < a.cin && cg * 16 < a.cout;
#endif
    // db[co] = sum over pixels of dy: the centre-tap fragment (kernel row 1, column shift 1 = the unshifted tile) times an all-ones A operand -- one
    // more MFMA instead of 16 unpack / add instructions.  The 24 (row pair, output-channel tile) units of a tile are dealt round-robin to the
    // nblk workgroups of a tile sequence (their cit = 0 waves: ONE wave per workgroup and output channel, so that the order of the atomics on
    // db is the workgroup order under DD_DETERMINISTIC=1), so no workgroup carries the bias alone (with the first block's waves summing it on
    // the vector pipe those workgroups ran 8 % longer than the rest: cycle stamps).
    unsigned bias_mask = 0;
    if (a.db != nullptr && cit == 0)
      for (int u = 0; u < 24; ++u) bias_mask |= (unsigned)((u % a.nblk) == cb) << u;
    const unsigned one2 = sizeof(T) == 2 && std::is_same<T, bf16_t>::value ? 0x3F803F80u : 0x3C003C00u;
    uint4 ones = {one2, one2, one2, one2};
    // (opaque: a known constant is re-materialised by a v_mov right in front of the in-place MFMA that reads it -- the VALU-write -> MFMA-read
    //  hazard nothing guards inside inline asm: db came out as NaN.  Unknown to the compiler, the four registers stay live across the launch.)
    asm volatile("" : "+v"(ones.x), "+v"(ones.y), "+v"(ones.z), "+v"(ones.w));
    const bool first_row = lane < 32;      // lanes holding the FIRST tile row of a fragment's row pair
    // Fragment addresses (32-bit LDS offsets of the current buffer).  A lane's pixel of a transposed read at tile position C (a compile-time
    // constant) is pl + C: address = image + [pl*64 + ((slot ^ key(pl + C)) << 4) + half] + C*64 = pb[C & 7] + C*64.
    unsigned pb[8], qa[2], qs[2];
    {
      const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3;
      const int yl = gq >> 1, xl = (gq & 1) * 8 + (t16 >> 2), halfb = (sub & 1) * 8, shi = sub >> 1;
      const int pl = yl * PW + xl;
#pragma unroll
      for (int c = 0; c < 8; ++c) pb[c] = lds_base + pl * ROW + (((cg * 2 + shi) ^ ((((pl + c) >> 2) & 1) << 1)) << 4) + halfb;
#pragma unroll
      for (int h = 0; h < 2; ++h) {      // x fragment: rows (2s | 2s+1); its row-swapped twin: rows (2s+1 | 2s); h: pixels +0 / +4
        const int xx = xl + 4 * h, key = ((xx >> 2) & 1) << 1;
        qa[h] = lds_base + B96_P_BYTES + (yl * DD_TILE + xx) * ROW + (((cit * 2 + shi) ^ key) << 4) + halfb;
        qs[h] = lds_base + B96_P_BYTES + ((1 - yl) * DD_TILE + xx) * ROW + (((cit * 2 + shi) ^ key) << 4) + halfb;
      }
    }
    f32x4_t acc[9][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t bacc[3] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};      // every row = this wave's part of db
    auto x_frag = [&](int s) { return b96_tr_pair(qa[0] + s * 2 * DD_TILE * ROW, qa[1] + s * 2 * DD_TILE * ROW); };
    auto xs_frag = [&](int s) { return b96_tr_pair(qs[0] + s * 2 * DD_TILE * ROW, qs[1] + s * 2 * DD_TILE * ROW); };
    auto dy_frag = [&](int sp, int n) {      // haloed rows (2sp | 2sp+1) at column shift tx = n % 3, output-channel tile j = n / 3 of this wave
      const int j = n / 3, tx = n - 3 * j, C = 2 * sp * PW + tx;
      return b96_tr_pair(pb[C & 7] + C * ROW + j * B96_PSUB, pb[(C + 4) & 7] + (C + 4) * ROW + j * B96_PSUB);
    };

    int sel = 0;
    for (int tile = tile0; tile < total_tiles; tile += a.ksplit, sel ^= 1) {
#ifdef B96_EXP_STAMP
      const long long st1 = __builtin_readcyclecounter();
#endif
      __syncthreads();      // the data-gradient waves' DMA of `tile` has landed (they wait for it before this barrier); buffer sel^1 is free
#ifdef B96_EXP_STAMP
      stamp_bar += __builtin_readcyclecounter() - st1; ++stamp_n;
#endif
      if (!active) continue;
      // 72 steps g = 9 s + n: row pair s, (output-channel tile, column shift) n.  Step g: MFMAs of kernel rows 0 and 2 with the OLD / NEW aligned
      // fragment, then the kernel-row-1 MFMA of step g - 1 (its select was written a step earlier).
      constexpr int NS = 72, BA = B96_BA;      // new fragments are requested BA steps ahead
      uint4 A[9], Bq[BA + 1], M[2], xf[2], xs[2];
#pragma unroll
      for (int n = 0; n < 9; ++n) A[n] = dy_frag(0, n);
      xf[0] = x_frag(0); xs[0] = xs_frag(0);
#pragma unroll
      for (int g = 0; g < BA; ++g) Bq[g] = dy_frag(g / 9 + 1, g % 9);
      b96_static_for<0, NS>([&](auto gc) {
        constexpr int g = decltype(gc)::value, s = g / 9, n = g - 9 * s, j = n / 3, tx = n - 3 * j;
        if constexpr (g + BA < NS) Bq[(g + BA) % (BA + 1)] = dy_frag((g + BA) / 9 + 1, (g + BA) % 9);
        if constexpr (n == 4 && s + 1 < 8) xf[(s + 1) & 1] = x_frag(s + 1);
        if constexpr (n == 6 && s + 1 < 8) xs[(s + 1) & 1] = xs_frag(s + 1);
        const uint4 bn = Bq[g % (BA + 1)], an = A[n];
        uint4 m;      // kernel row 1: haloed rows (2s+2 | 2s+1) = (new fragment's first row | old fragment's second row)
#ifdef B96_EXP_NO_SEL
        m = bn;
#else
        m.x = first_row ? bn.x : an.x; m.y = first_row ? bn.y : an.y; m.z = first_row ? bn.z : an.z; m.w = first_row ? bn.w : an.w;
#endif
        asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));      // pins the select HERE: left alone hipcc sinks it to its use a step later, two MFMAs in front of the in-place MFMA that reads it
        M[g & 1] = m;
        __builtin_amdgcn_sched_barrier(0);
        b96_mma_inplace<T>(acc[tx][j], xf[s & 1], an);
        b96_mma_inplace<T>(acc[6 + tx][j], xf[s & 1], bn);
        if constexpr (g > 0) {
          constexpr int gp = g - 1, sp = gp / 9, np = gp - 9 * sp, jp = np / 3, txp = np - 3 * jp;
          b96_mma_inplace<T>(acc[3 + txp][jp], xs[sp & 1], M[gp & 1]);
          if constexpr (txp == 1)      // centre tap = the unshifted dy tile
            if ((bias_mask >> (sp * 3 + jp)) & 1u) b96_mma_inplace<T>(bacc[jp], ones, M[gp & 1]);
        }
        A[n] = bn;
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_nop 7" ::: "memory");
      b96_mma_inplace<T>(acc[3 + 2][2], xs[7 & 1], M[(NS - 1) & 1]);      // step 71: n = 8 -> j = 2, tx = 2
      const int flip = sel ? -B96_BUF : B96_BUF;
#pragma unroll
      for (int c = 0; c < 8; ++c) pb[c] += flip;
      qa[0] += flip; qa[1] += flip; qs[0] += flip; qs[1] += flip;
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (see b96_mma_inplace)
    // flush: D[t][j] rows = input channels cit*16 + q4 + e, column = output channel 32j + 16cg + li; shifted-dy tap t is TensorFlow's tap 8 - t
    dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
    if (active) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int co = 32 * j + 16 * cg + li;
        if (co < a.cout) {
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ci = cb * 32 + cit * 16 + q4 + e;
#ifdef B96_EXP_NO_FLUSH
              if (ci < a.cin && acc[t][j][e] == 123.456f) atomicAdd(a.dw + ((long)(8 - t) * a.cin + ci) * a.cout + co, acc[t][j][e]);
#else
              if (ci < a.cin) atomicAdd(a.dw + ((long)(8 - t) * a.cin + ci) * a.cout + co, acc[t][j][e]);
#endif
            }
        }
        if (bias_mask != 0u && lane < 16 && co < a.cout) atomicAdd(a.db + co, bacc[j][0]);
      }
    }
  }
#ifdef B96_EXP_STAMP
  if (lane == 0 && (blockIdx.x == 3 || blockIdx.x == 100))
    printf("block %d wave %d: tiles %lld total %lld clk (%lld ticks of 10 ns), vmcnt wait %lld, barrier %lld\n", (int)blockIdx.x, wave, stamp_n,
           (long long)__builtin_readcyclecounter() - stamp_t0, (long long)__builtin_amdgcn_s_memrealtime() - stamp_r0, stamp_vm, stamp_bar);
#endif
  dd_det_end();
}

template <typename T, bool MASK, bool ACCUM>
int launch96_flags(const Bw96P& p, hipStream_t stream) {
  const size_t lds = 2 * (size_t)B96_BUF;
  dd_det_sync();
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_bwd96_kernel<T, MASK, ACCUM>));
  const long blocks = (long)p.nblk * p.ksplit;
  hipLaunchKernelGGL((conv_bwd96_kernel<T, MASK, ACCUM>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
template <typename T>
int launch96(const Bw96P& p, bool mask, bool accum, hipStream_t stream) {
  if (mask) return accum ? launch96_flags<T, true, true>(p, stream) : launch96_flags<T, true, false>(p, stream);
  return accum ? launch96_flags<T, false, true>(p, stream) : launch96_flags<T, false, false>(p, stream);
}

}  // namespace

// Called by dd_conv3x3_bwd (csrc/dd_conv_bwd.hip, which has validated the descriptor) for 65 - 96 output channels with a data gradient.
int dd_conv_bwd96_launch(const dd_conv_bwd_args* a, hipStream_t stream) {
  Bw96P p;
  p.dy = a->dy; p.x = a->x; p.wd = a->wd; p.dx = a->dx; p.dw = a->dw; p.db = a->db;
  p.lddy = a->ld_dy; p.ldx = a->ld_x; p.lddx = a->ld_dx;
  p.cout = a->cout; p.cin = a->cin; p.coutv = (a->cout + 7) / 8 * 8; p.cinv = (a->cin + 7) / 8 * 8;
  p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.nblk = dd_ceil_div(a->cin, 32);
  const long total_tiles = (long)a->B * p.tiles_x * p.tiles_y;
  long ksplit = dd_device_cus() / p.nblk;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > total_tiles) ksplit = total_tiles;
  p.ksplit = (int)ksplit;
  return a->dtype == DD_BF16 ? launch96<bf16_t>(p, a->use_mask != 0, a->accumulate != 0, stream)
                             : launch96<f16_t>(p, a->use_mask != 0, a->accumulate != 0, stream);
}
