// Fused backward of a 3x3 SAME convolution on CDNA4: data gradient AND weight gradient of one layer from ONE pass over dy and x.
//
// Reference seam replaced: the two gradient ops TensorFlow's autodiff emits for tf.layers.conv2d (Conv2DBackpropInput and
// Conv2DBackpropFilter, reached from Training.py:701-702 through UNet.py:38-48 / Architecture.py:343-420), plus the ReluGrad that follows the
// data gradient and the BiasAddGrad of the layer.
//
//   dx[p][ci]       = (x[p][ci] > 0) * sum_{t, co} Wd[t][ci][co] * dy[p + off(t)][co]            (+= an existing gradient when the tensor has two consumers)
//   dW[t][ci][co]   = sum_p x[p + off(t)][ci] * dy[p][co]  =  sum_q x[q][ci] * dy[q - off(t)][co]
//   db[co]          = sum_p dy[p][co]
//
// The layer-by-layer path runs these as two launches (csrc/dd_conv_igemm.hip on dy, csrc/dd_conv_wgrad.hip): dy is fetched twice and x twice
// (once as the ReLU mask of dx, once as the weight-gradient operand) -- 5 activation-sized HBM transfers where 3 are needed, and both launches
// sit on the memory roof.  In the second form of dW above BOTH gradients need the same two LDS images of a 16x16 pixel tile: dy with a 1-pixel
// halo (18x18) and x without one.  So a workgroup fetches each tile once:
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4, double buffered, 8 pixels x 128 bytes per wave-instruction, swizzled on the linear pixel
//     index) -- no registers, no I/O role: the next tile streams in while this one is multiplied;
//   * waves 0-3 (one per 16 input channels) compute dx^T = Wd * dy^T with the WEIGHTS as the register-resident A operand (18 fragments) and
//     the dy pixels as B, walking the 18 haloed rows once: a row's fragment feeds the three output rows it touches, an output row is complete
//     two haloed rows later and is masked with x (read from the LDS image), rounded and stored straight from the accumulators;
//   * waves 4-7 (one per 16 output channels) accumulate dW for the whole launch in registers (9 taps x 4 input-channel tiles = 144
//     accumulators), both operands read transposed (ds_read_b64_tr_b16: the reduction runs over pixels), and db from the centre-tap fragments;
//     one set of fp32 atomics per workgroup at the end.
// A SIMD hosts one wave of each role: 576 MFMAs per tile and SIMD.  Channels: C_out <= 64 (one 128-byte LDS row per pixel), C_in in blocks of
// 64 (one workgroup column per block; dy is re-read per block).
#include "dd_common.h"

#ifndef CBW_DF_RING
#define CBW_DF_RING 2      // the weight-gradient role's ring of dy fragments (2: one step ahead)
#endif
#ifndef CBW_REUSE
#define CBW_REUSE 1        // the weight-gradient role keeps the three row-pair-aligned dy fragments of kernel row 2 for the next row pair's kernel row 0
#endif
#ifndef BW_DMA_SPAN
#define BW_DMA_SPAN 8      // eighths of a tile's fragment steps over which the data-gradient waves issue the next tile's DMA pieces
#endif
namespace {

struct BwdP {
  const void* dy; const void* x; const void* wd; void* dx; float* dw; float* db;
  int lddy, ldx, lddx;
  int cout, cin, coutv, cinv;      // logical / staged (rounded up to 8) channel counts
  int n_pad, k_pad;                // packed data-gradient weights [9][n_pad (ci)][k_pad (co)]
  int B, H, W, tiles_x, tiles_y;
  int nblk, nblk_co, ksplit;       // 64-channel blocks of C_in / of C_out (> 1 only without a data gradient); workgroups per block pair
  int co_base;                     // with a data gradient and C_out > 64: this LAUNCH covers output channels [co_base, co_base + 64) (host loop)
  int use_mask, accumulate;
};

typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;

constexpr int BW_PW = DD_TILE + 2;                                   // haloed tile width
constexpr int BW_PCH = (BW_PW * BW_PW + 7) / 8;                      // 41 chunks of 8 pixels (1 KiB)
constexpr int BW_P_BYTES = BW_PCH * 1024, BW_Q_BYTES = DD_TILE * DD_TILE * DD_LDS_ROW, BW_BUF = BW_P_BYTES + BW_Q_BYTES;

__device__ __forceinline__ void bw_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}

// LDS addresses are 32-bit byte offsets (kept in integers so that the swizzle XORs stay integer ops and every read is a ds_read with an
// immediate offset; generic pointers here compiled to flat loads and 64-bit address arithmetic)
typedef uint32_t bw_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t bw_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 bw_lds16(unsigned off) {
  const bw_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) bw_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ uint2 bw_lds8(unsigned off) {
  const bw_u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) bw_u32x2*>(off);
  return uint2{v[0], v[1]};
}
__device__ __forceinline__ uint4 bw_tr_pair(unsigned a0, unsigned a1) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_ptr>(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_ptr>(a1));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  const uint4 v = {l2.x, l2.y, h2.x, h2.y};
  return v;
}

// acc += A * B with the accumulator tied to ONE register tuple (inline asm, "+v").  Through the builtin hipcc renames the 36 accumulator tuples
// of the weight-gradient role as it goes (vDst != SrcC: 62 distinct tuples in the loop, ~100 extra registers, spills inside the MFMA stream).
// No hazard recogniser sees inside the asm: the role's accumulators are each touched once per 36 MFMAs and read only after the tile loop
// (behind explicit s_nops), its A / B operands are written by LDS reads only (waited for by the s_waitcnt the compiler still inserts).
template <typename T> __device__ __forceinline__ void bw_mma_inplace(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void bw_mma_inplace<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const bw_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}
template <> __device__ __forceinline__ void bw_mma_inplace<f16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  const bw_u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
}

// MASK / ACCUM are template flags, not runtime branches: a runtime `accumulate` left the (never executed) gradient load in the row loop, its
// destination registers shared with the row pointer, and hipcc guarded that with s_waitcnt vmcnt(0) once per output row -- a wait for every DMA
// piece in flight, i.e. an HBM round trip 16 times per tile.
template <typename T, bool MASK, bool ACCUM>
__device__ __forceinline__ void conv_bwd_body(const BwdP& a, char* smem, int block) {
  static_assert(sizeof(T) == 2, "fused conv backward: bf16 / fp16 storage");
  constexpr int PW = BW_PW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = block / a.ksplit, ks = block - pair * a.ksplit;
  const int cb = pair % a.nblk, ob = pair / a.nblk + (a.co_base >> 6);      // input- / output-channel block of this workgroup column
  const int wr = wave & 3;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int per_img = a.tiles_y * a.tiles_x;
  const int total_tiles = a.B * per_img;
  // Tile sequence of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own L2: iteration i of
  // the ksplit workgroups of a block covers ksplit consecutive tiles, and the ksplit/8 workgroups of ONE XCD take a contiguous run of them, so
  // that neighbouring tiles -- which share their dy halo rows and columns -- meet in the same L2.
#ifdef CBW_EXP_NO_XCD
  const int xcd_n = 1;
#else
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
#endif
  const int per_xcd = a.ksplit / xcd_n;
  const int tile0 = (ks % xcd_n) * per_xcd + ks / xcd_n;      // first tile; then + ksplit per iteration

  if (wave < 4) {
    // ================================================================== data-gradient role: wave = input-channel tile wr of this block
    // This role also issues ALL the LDS-DMA of the workgroup (the weight-gradient role holds 144 accumulators and has no registers for the
    // per-lane piece coordinates; with the DMA here it has no vector-memory traffic at all).  A lane's part of a 1-KiB chunk: pixel r of the
    // chunk, logical 16-byte channel slot ls.  x tile: chunk c = wr*8 + k (tile row c >> 1, pixels (c & 1)*8 + r); haloed dy tile: chunk
    // c = k*4 + wr (pixels c*8 + r of the 18x18 tile in row-major order), coordinates recomputed per piece (a few VALU ops under the MFMAs).
    constexpr int QPC = 8, PPC = (BW_PCH + 3) / 4, NP = QPC + PPC;      // 8 + 11 pieces per wave and tile
    const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
    const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
    const int r = lane >> 3, ls = (lane & 7) ^ r;
    const int xch = cb * 64 + ls * 8, dch = ob * 64 + ls * 8;
    const bool x_ok = xch < a.cinv, d_ok = dch < a.coutv;
    const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
    const int x_row = a.W * a.ldx * 2, x_lane = (r * a.ldx + xch) * 2;                  // bytes
    const int dy_row = a.W * a.lddy * 2, dy_pix = a.lddy * 2, dy_lane = dch * 2;
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
      const unsigned buf = lds_base + sel * BW_BUF;
      // (opaque copies per piece: otherwise hipcc hoists every piece's tile-invariant coordinates and offsets out of the tile loop -- ~25
      //  registers that then spill, and a scratch reload in a wave with DMA in flight means s_waitcnt vmcnt(0) = an HBM round trip mid-tile)
      int rr = r, xl = x_lane;
      asm volatile("" : "+v"(rr), "+v"(xl));
      if (k < QPC) {
        const int c = wr * QPC + k, row = c >> 1, h = c & 1;
#ifdef CBW_EXP_NO_DMA
        const bool ok = false;      // (knock-out build: every piece comes from the zero page)
#else
        const bool ok = o.live && x_ok && o.y0 + row < a.H && o.x0 + h * 8 + rr < a.W;
#endif
        bw_dma_1k(ok ? o.q + (row * x_row + h * 8 * a.ldx * 2 + xl) : zero, buf + BW_P_BYTES + c * 1024);
      } else {
        const int c = (k - QPC) * 4 + wr;
        if (c < BW_PCH) {      // wave-uniform
          const int pix = c * 8 + rr;
          const int py = (pix * 3641) >> 16, px = pix - py * PW;          // pix / 18 for pix < 400
          const int gy = o.y0 - 1 + py, gx = o.x0 - 1 + px;
#ifdef CBW_EXP_NO_DMA
          const bool ok = false;
#else
          const bool ok = o.live && d_ok && pix < PW * PW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#endif
          bw_dma_1k(ok ? o.p + (py * dy_row + px * dy_pix + dy_lane) : zero, buf + c * 1024);
        }
      }
    };
    Origin oc = origin(tile0);
#pragma unroll
    for (int k = 0; k < NP; ++k) piece(k, oc, 0);

    const int li = lane & 15, q = lane >> 4;
    const int ci_row = cb * 64 + wr * 16 + li;                  // A rows: this lane's weight row
    const int c4 = cb * 64 + wr * 16 + q * 4;                   // D rows: the 4 input channels this lane stores
#ifdef CBW_EXP_NO_DROLE
    const bool active = false;
#else
    const bool active = a.dx != nullptr && cb * 64 + wr * 16 < a.cin;      // dx == NULL (the layer's input needs no gradient): these waves only feed the DMA
#endif
    uint4 wf[9][2];
    {
      const T* Wd = reinterpret_cast<const T*>(a.wd);
      const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const int k0 = kc * 32 + q * 8;
          const bool ok = active && ci_row < a.n_pad && ob * 64 + k0 < a.k_pad;
          wf[t][kc] = *reinterpret_cast<const uint4*>(ok ? Wd + ((long)t * a.n_pad + ci_row) * a.k_pad + ob * 64 + k0 : zw);
        }
    }
    // Fragment addresses (32-bit LDS offsets of the CURRENT buffer; they flip by +-BW_BUF per tile).  dy image: pixel C + li (C = row*18 + dx,
    // a compile-time constant), 16-byte slot kc*4 + q  ->  (C + li)*128 + ((slot ^ ((C + li) & 7)) << 4) = d0[C & 7] ^ (kc << 6), + C*128.
    unsigned d0[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] = lds_base + li * DD_LDS_ROW + ((q ^ ((li + c) & 7)) << 4);
    unsigned mrow = lds_base + BW_P_BYTES + li * DD_LDS_ROW + (((wr * 2 + (q >> 1)) ^ (li & 7)) << 4) + (q & 1) * 8;     // x image: pixel (row, li), channels wr*16 + q*4 ..
    T* __restrict__ DX = reinterpret_cast<T*>(a.dx);
    const bool ch_ok = c4 < a.cinv;
    const long row_stride = (long)a.W * a.lddx;

    int sel = 0;
    for (int tile = tile0; tile < total_tiles; tile += a.ksplit, sel ^= 1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA pieces of `tile` have landed
      __syncthreads();                      // ... and everyone's; buffer sel^1 is free
      const Origin on = origin(tile + a.ksplit);
      if (!active) {
#pragma unroll
        for (int k = 0; k < NP; ++k) piece(k, on, sel ^ 1);
        oc = on;
        continue;
      }
      const bool col_ok = ch_ok && oc.x0 + li < a.W;
      T* dxp = DX + ((long)oc.b * a.H * a.W + (long)oc.y0 * a.W + oc.x0 + li) * a.lddx + c4;
      f32x4_t acc[4];      // output rows y % 4: row y is complete after haloed row y + 2, written out during haloed row y + 3, re-used by row y + 4
      uint2 oldv[4], mv[4];
      // 108 fragment steps = 18 haloed rows x 3 column shifts x 2 K-chunks, up to 3 MFMAs each (the output rows yy, yy-1, yy-2).  Fragments are
      // requested AHEAD steps before use; sched_barriers keep that order (left alone hipcc waits for each read right before its MFMAs).
#ifndef CBW_RING
#define CBW_RING 6
#endif
      constexpr int RING = CBW_RING, AHEAD = RING - 1, NF = PW * 6;
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / 6, j = f - 6 * yy, C = yy * PW + (j >> 1);
        return bw_lds16((d0[C & 7] ^ ((j & 1) << 6)) + C * DD_LDS_ROW);
      };
      auto write_row = [&](int y) {      // mask, round, (accumulate,) store output row y
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
        if (col_ok && oc.y0 + y < a.H) *reinterpret_cast<uint2*>(dxp + y * row_stride) = o2;
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PW; ++yy) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int f = yy * 6 + j, dx = j >> 1, kc = j & 1;      // (layers with <= 32 output channels multiply zeros for kc = 1: rare and small)
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < DD_TILE) {
            acc[yy % 4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (MASK) mv[yy % 4] = bw_lds8(mrow + yy * DD_TILE * DD_LDS_ROW);
            oldv[yy % 4] = uint2{0u, 0u};
            if (ACCUM && col_ok && oc.y0 + yy < a.H) oldv[yy % 4] = *reinterpret_cast<const uint2*>(dxp + yy * row_stride);
          }
          if (j == 2 && yy >= 3) write_row(yy - 3);      // (under this row's MFMAs)
          {      // the NP DMA pieces of the next tile, spread evenly over the first BW_DMA_SPAN / 8 of the NF steps (piece k at step k*SPAN/NP)
            constexpr int SPAN = NF * BW_DMA_SPAN / 8 > NP ? NF * BW_DMA_SPAN / 8 : NP;
            const int k0 = (f * NP + SPAN - 1) / SPAN;
            if (k0 < NP && (k0 * SPAN) / NP == f) piece(k0, on, sel ^ 1);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < DD_TILE) acc[y % 4] = mma16<T>(wf[dy * 3 + dx][kc], ring[f % RING], acc[y % 4]);
          }
#ifndef CBW_EXP_ONE_BARRIER
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      write_row(DD_TILE - 1);
      const int flip = sel ? -BW_BUF : BW_BUF;
#pragma unroll
      for (int c = 0; c < 8; ++c) d0[c] += flip;
      mrow += flip;
      oc = on;
    }
  } else {
    // ================================================================== weight-gradient role: wave = output-channel tile wr
    const int li = lane & 15, q4 = (lane >> 4) * 4;
#ifdef CBW_EXP_NO_WROLE
    const bool active = false;
#else
    const bool active = ob * 64 + wr * 16 < a.cout;
#endif
    const bool bias_wave = a.db != nullptr && cb == 0;
    const int nci = min(4, (a.cin - cb * 64 + 15) >> 4);        // input-channel tiles of this block that exist
    // Fragment addresses (32-bit LDS offsets).  A lane's pixel for a transposed read at tile position C (= row*PW + dx, a compile-time
    // constant) is pl + C, so its address is  tile + [pl*128 + ((slot ^ ((pl + (C & 7)) & 7)) << 4) + half] + C*128 : eight lane-dependent
    // bases plus an immediate.  The bases hold the CURRENT buffer's addresses and flip by +-BW_BUF per tile (no second copy kept).
    unsigned pb[8], qb[2];
    {
      const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3;
      const int yl = gq >> 1, xl = (gq & 1) * 8 + (t16 >> 2), halfb = (sub & 1) * 8;
      const int pl = yl * PW + xl, ql = yl * DD_TILE + xl;
#pragma unroll
      for (int c = 0; c < 8; ++c) pb[c] = lds_base + pl * DD_LDS_ROW + (((wr * 2 + (sub >> 1)) ^ ((pl + c) & 7)) << 4) + halfb;
#pragma unroll
      for (int h = 0; h < 2; ++h)      // input-channel tile 0; tile i: ^ (i << 5)
        qb[h] = lds_base + BW_P_BYTES + ql * DD_LDS_ROW + (((sub >> 1) ^ ((ql + 4 * h) & 7)) << 4) + halfb;
    }
    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    auto x_frag = [&](int s, int i) {      // x^T: rows 2s, 2s+1 of the tile, input-channel tile i
      return bw_tr_pair((qb[0] ^ (i << 5)) + (2 * s * DD_TILE) * DD_LDS_ROW, (qb[1] ^ (i << 5)) + (2 * s * DD_TILE + 4) * DD_LDS_ROW);
    };
    auto dy_frag = [&](int step) {         // dy^T shifted by tap t = step % 9, rows 2s, 2s+1 (s = step / 9)
      const int s = step / 9, t = step - 9 * s;
      const int c0 = (2 * s + t / 3) * PW + t % 3;
      return bw_tr_pair(pb[c0 & 7] + c0 * DD_LDS_ROW, pb[(c0 + 4) & 7] + (c0 + 4) * DD_LDS_ROW);
    };

    int sel = 0;
    for (int tile = tile0; tile < total_tiles; tile += a.ksplit, sel ^= 1) {
      __syncthreads();      // the data-gradient waves' DMA of `tile` has landed (they wait for it before this barrier); buffer sel^1 is free
      if (!active) continue;
      // 72 steps = 8 pixel-row pairs x 9 taps, 4 MFMAs each; the dy fragment of step n+1 and (during taps 4..7) the x fragments of the next
      // row pair are requested before the MFMAs of step n.  sched_barriers keep hipcc from hoisting a whole row pair's 26 reads (52 registers).
      // (round 6) The dy fragment of tap (kernel row 0, column shift tx) of row pair s covers the haloed rows (2s, 2s + 1) -- the SAME fragment as tap
      // (kernel row 2, tx) of row pair s - 1: it is kept in registers (3 fragments) instead of read again: 10 transposing read pairs per 36 MFMAs
      // instead of 13 on the role whose bound is that instruction's return path.
      constexpr int DFR = CBW_DF_RING, DFA = DFR - 1;      // dy fragments requested DFA steps ahead
      uint4 xf[2][4], df[DFR];
#if CBW_REUSE
      uint4 keep[3];
#define CBW_NEEDS_READ(st) (!((st) >= 9 && (st) % 9 < 3))
#else
#define CBW_NEEDS_READ(st) true
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[0][i] = x_frag(0, i);
#pragma unroll
      for (int n = 0; n < DFA; ++n) df[n] = dy_frag(n);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int step = s * 9 + t;
        if (step + DFA < 72 && CBW_NEEDS_READ(step + DFA)) df[(step + DFA) % DFR] = dy_frag(step + DFA);
#ifdef CBW_EXP_HALF_X
        if (s + 1 < 8 && t >= 4 && t < 6) xf[(s + 1) & 1][t - 4] = x_frag(s + 1, t - 4);
#else
        if (s + 1 < 8 && t >= 4 && t < 8) xf[(s + 1) & 1][t - 4] = x_frag(s + 1, t - 4);
#endif
        __builtin_amdgcn_sched_barrier(0);
#if CBW_REUSE
        const uint4 cur = (s > 0 && t < 3) ? keep[t] : df[step % DFR];
        if (t >= 6) keep[t - 6] = cur;
#else
        const uint4 cur = df[step % DFR];
#endif
#ifdef CBW_EXP_HALF_MMA
#pragma unroll
        for (int i = 0; i < 2; ++i) bw_mma_inplace<T>(acc[t][i], xf[s & 1][i], cur);
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) bw_mma_inplace<T>(acc[t][i], xf[s & 1][i], cur);
#endif
        if (t == 4 && bias_wave) {      // centre tap = the unshifted dy tile: 8 pixels of channel wr*16 + li per lane
          float f[8];
          unpack8t<T>(df[step % DFR], f);
          bsum += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
#ifndef CBW_EXP_ONE_BARRIER_W
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      // the other buffer next time
      const int flip = sel ? -BW_BUF : BW_BUF;
#pragma unroll
      for (int c = 0; c < 8; ++c) pb[c] += flip;
      qb[0] += flip; qb[1] += flip;
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (see bw_mma_inplace)
    // flush: D[t][i] rows = input channels i*16 + q4 + e, column = output channel wr*16 + li; shifted-dy tap t is TensorFlow's tap 8 - t
    dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
    if (active) {
      const int co = ob * 64 + wr * 16 + li;
      if (co < a.cout) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (i >= nci) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ci = cb * 64 + i * 16 + q4 + e;
#ifdef CBW_EXP_NO_FLUSH
              if (ci < a.cin && acc[t][i][e] == 123.456f) atomicAdd(a.dw + ((long)(8 - t) * a.cin + ci) * a.cout + co, acc[t][i][e]);
#else
              if (ci < a.cin) atomicAdd(a.dw + ((long)(8 - t) * a.cin + ci) * a.cout + co, acc[t][i][e]);
#endif
            }
          }
      }
      if (bias_wave) {
        float b = bsum;
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        if (lane < 16 && co < a.cout) atomicAdd(a.db + co, b);
      }
    }
  }
  dd_det_end();
}

template <typename T, bool MASK, bool ACCUM>
__global__ __launch_bounds__(512) void conv_bwd_kernel(const BwdP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  conv_bwd_body<T, MASK, ACCUM>(a, smem, blockIdx.x);
}

// Several weight-gradient-only problems on the same pixel grid as ONE launch (blockIdx.y picks the problem): the four 128-channel layers of the
// U-Net's 32 x 32 level each ran 256 workgroups over 4 tiles apiece and then flushed 256 x 36 864 fp32 atomics -- a third of a 57-us launch
// (7.2).  Side by side each problem gets a quarter of the workgroups with four times the tiles: the same MFMA work per workgroup, a quarter of
// the atomics per problem, three launch boundaries fewer.
constexpr int BW_MAX_MULTI = 4;
struct BwdMulti { BwdP p[BW_MAX_MULTI]; int blocks[BW_MAX_MULTI]; };
template <typename T>
__global__ __launch_bounds__(512) void conv_bwd_multi_kernel(const BwdMulti m) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  if ((int)blockIdx.x >= m.blocks[blockIdx.y]) {      // (whole workgroups: no barrier is skipped by part of one)
    dd_det_wait();                                    // (DD_DETERMINISTIC=1: an idle workgroup still takes and passes on its ticket in order)
    dd_det_end();
    return;
  }
  conv_bwd_body<T, false, false>(m.p[blockIdx.y], smem, blockIdx.x);
}

static int bwd_cus() {
  return dd_device_cus();
}

template <typename T, bool MASK, bool ACCUM>
int launch_bwd_flags(const BwdP& p, hipStream_t stream) {
  const size_t lds = 2 * (size_t)BW_BUF;
  dd_det_sync();
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_bwd_kernel<T, MASK, ACCUM>));
  const long blocks = (long)p.nblk * p.nblk_co * p.ksplit;
  hipLaunchKernelGGL((conv_bwd_kernel<T, MASK, ACCUM>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
template <typename T>
int launch_bwd(const BwdP& p, hipStream_t stream) {
  if (p.use_mask) return p.accumulate ? launch_bwd_flags<T, true, true>(p, stream) : launch_bwd_flags<T, true, false>(p, stream);
  return p.accumulate ? launch_bwd_flags<T, false, true>(p, stream) : launch_bwd_flags<T, false, false>(p, stream);
}

}  // namespace

static int bwd_fill(BwdP& p, const dd_conv_bwd_args* a);
int dd_conv_bwd96_launch(const dd_conv_bwd_args* a, hipStream_t stream);      // csrc/dd_conv_bwd96.hip

// weight / bias gradients only (dx = NULL) of up to BW_MAX_MULTI layers on the same [B, H, W] grid, one launch
extern "C" int dd_conv3x3_bwd_multi(const dd_conv_bwd_args* a, int n, dd_stream stream) {
  DD_REQUIRE(a && n >= 1 && n <= BW_MAX_MULTI, "dd_conv3x3_bwd_multi: 1 to %d problems", BW_MAX_MULTI);
  BwdMulti m;
  memset(&m, 0, sizeof(m));
  long gx = 1;
  for (int i = 0; i < n; ++i) {
    DD_REQUIRE(a[i].dx == nullptr && a[i].B == a[0].B && a[i].H == a[0].H && a[i].W == a[0].W && a[i].dtype == a[0].dtype && !a[i].use_mask && !a[i].accumulate,
               "dd_conv3x3_bwd_multi: problem %d: weight-gradient-only problems (dx = NULL) on one [B, H, W] grid and storage type", i);
    if (int rc = bwd_fill(m.p[i], &a[i])) return rc;
    BwdP& p = m.p[i];
    const long total_tiles = (long)p.B * p.tiles_x * p.tiles_y;
    p.nblk_co = dd_ceil_div(p.cout, 64);
    long ksplit = bwd_cus() / ((long)n * p.nblk * p.nblk_co);
    if (ksplit < 1) ksplit = 1;
    if (ksplit > total_tiles) ksplit = total_tiles;
    p.ksplit = (int)ksplit; p.co_base = 0; p.use_mask = 0; p.accumulate = 0;
    m.blocks[i] = p.nblk * p.nblk_co * p.ksplit;
    if (m.blocks[i] > gx) gx = m.blocks[i];
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = 2 * (size_t)BW_BUF;
  dd_det_sync();
  if (a[0].dtype == DD_BF16) {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_bwd_multi_kernel<bf16_t>));
    hipLaunchKernelGGL(conv_bwd_multi_kernel<bf16_t>, dim3((unsigned)gx, (unsigned)n), dim3(512), lds, s, m);
  } else {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_bwd_multi_kernel<f16_t>));
    hipLaunchKernelGGL(conv_bwd_multi_kernel<f16_t>, dim3((unsigned)gx, (unsigned)n), dim3(512), lds, s, m);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}

static int bwd_fill(BwdP& p, const dd_conv_bwd_args* a) {
  DD_REQUIRE(a && a->dy && a->x && a->dw && (a->wd || !a->dx), "dd_conv3x3_bwd: null pointer");      // dx (and then wd) may be NULL: weight / bias gradients only
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_conv3x3_bwd: dtype %d (bf16 / f16 storage only; f32 takes dd_conv_igemm + dd_conv_wgrad)", a->dtype);
  DD_REQUIRE(a->cout > 0 && a->cin > 0, "dd_conv3x3_bwd: cout=%d cin=%d", a->cout, a->cin);
  const int coutv = (a->cout + 7) / 8 * 8, cinv = (a->cin + 7) / 8 * 8;
  DD_REQUIRE(a->ld_dy % 8 == 0 && a->ld_x % 8 == 0 && coutv <= a->ld_dy && cinv <= a->ld_x && (!a->dx || (a->ld_dx % 4 == 0 && cinv <= a->ld_dx)),
             "dd_conv3x3_bwd: ld_dy=%d ld_x=%d ld_dx=%d must cover the channel counts rounded to 8 (ld_dy, ld_x multiples of 8)", a->ld_dy, a->ld_x, a->ld_dx);
  DD_REQUIRE(((uintptr_t)a->dy % 16) == 0 && ((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->wd % 16) == 0 && ((uintptr_t)a->dx % 8) == 0,
             "dd_conv3x3_bwd: dy / x / wd must be 16-byte aligned, dx 8-byte aligned");
  DD_REQUIRE(!a->dx || (a->n_pad >= a->cin && a->k_pad >= a->cout && a->k_pad % 8 == 0), "dd_conv3x3_bwd: packed weights [9][n_pad=%d][k_pad=%d] do not cover %d x %d", a->n_pad, a->k_pad, a->cin, a->cout);
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && (long)a->B * a->H * a->W < (1L << 31) / 256, "dd_conv3x3_bwd: empty or oversized grid");
  p.dy = a->dy; p.x = a->x; p.wd = a->wd; p.dx = a->dx; p.dw = a->dw; p.db = a->db;
  p.lddy = a->ld_dy; p.ldx = a->ld_x; p.lddx = a->ld_dx;
  p.cout = a->cout; p.cin = a->cin; p.coutv = coutv; p.cinv = cinv;
  p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.nblk = dd_ceil_div(a->cin, 64);
  p.nblk_co = 1; p.ksplit = 1; p.co_base = 0; p.use_mask = a->use_mask; p.accumulate = a->accumulate;
  return DD_OK;
}

extern "C" int dd_conv3x3_bwd(const dd_conv_bwd_args* a, dd_stream stream) {
  BwdP p;
  if (int rc = bwd_fill(p, a)) return rc;
  const long total_tiles = (long)a->B * p.tiles_x * p.tiles_y;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // 65 - 96 output channels with a data gradient (round 6): one launch of the kernel that gives a workgroup a 32-channel third of the input
  // against all output channels (csrc/dd_conv_bwd96.hip); DD_CONV_BWD96=0: one launch per 64 output channels below
  static const bool bwd96 = [] { const char* e = getenv("DD_CONV_BWD96"); return !(e && e[0] == '0'); }();
  if (bwd96 && a->dx && a->cout > 64 && a->cout <= 96) return dd_conv_bwd96_launch(a, s);
  // Without a data gradient the output-channel blocks are workgroup columns of ONE launch.  With one, a launch covers 64 output channels (its
  // data-gradient role sums over all of them): wider layers run as consecutive launches, the later ones accumulating into dx (the ReLU mask
  // distributes over the partial sums; dx is rounded once more per extra launch, as with any accumulated gradient).
  const int n_launch = a->dx ? dd_ceil_div(a->cout, 64) : 1;
  p.nblk_co = a->dx ? 1 : dd_ceil_div(a->cout, 64);
  long ksplit = bwd_cus() / (p.nblk * p.nblk_co);
  if (ksplit < 1) ksplit = 1;
  if (ksplit > total_tiles) ksplit = total_tiles;
  p.ksplit = (int)ksplit;
  p.use_mask = a->use_mask;
  for (int l = 0; l < n_launch; ++l) {
    p.co_base = l * 64;
    p.accumulate = a->accumulate || l > 0;
    const int rc = a->dtype == DD_BF16 ? launch_bwd<bf16_t>(p, s) : launch_bwd<f16_t>(p, s);
    if (rc != DD_OK) return rc;
  }
  return DD_OK;
}
