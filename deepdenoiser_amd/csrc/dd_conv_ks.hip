// K-streamed register-weight 3x3 convolution on CDNA4: the deep-K, skinny-N layers of the Tiramisu backbone (BASELINE cfg-3).
//
// Reference seam (file:line in /root/reference): TensorFlow/Tiramisu.py:26-41 (`__dense_block`: conv3x3 over the growing concat, K = 9 x up to
// 1 088 channels, N = 16 ... 128 new channels) and Tiramisu.py:60-65 (`__upsample`: tf.layers.conv2d_transpose 3x3 / stride 2 / SAME).
//
// Why a third conv kernel.  csrc/dd_conv_igemm.hip keeps a layer's weights in LDS and csrc/dd_conv_rw.hip in registers -- for the whole
// launch.  With 9 x 576 x 64 weights (663 KB) neither fits: round 2 ran those layers on the igemm kernel's streamed fallback (weights staged
// through registers into LDS tap by tap, 4 waves) at ~250 TFLOP/s = 10 % of the MFMA peak, and its N = 128 instantiation spilled 150 registers.
// Here the reduction is cut into 64-channel K-slices and BOTH operands stream per slice:
//   * the 18 x 18 haloed input tile of the slice arrives by LDS-DMA (41 chunks of 1 KiB, double buffered: slice s + 1 lands while slice s is
//     multiplied), exactly as in conv_rw8_kernel;
//   * the slice's weights (9 taps x 2 K-chunks = 18 MFMA A fragments = 72 registers per wave) are loaded straight from the packed image in L2
//     into REGISTERS, one slice ahead (a second set of 72 registers; the unit loop is unrolled by two so the sets swap by name, not by moves).
//     Every CU reads the same 72 KB per slice: an L2 stream, not an HBM one (workgroups of <= 32 output channels: once per WORKGROUP, through LDS --
//     KS_WLDS below);
//   * the 8 rows x 16 channels of a wave's output stay in 32 accumulator registers across all slices and leave once, after the last one.
// Per slice and SIMD: 2 waves x 144 MFMAs x 16 cycles = 4 608 cycles for 41 KiB of DMA + 72 KiB of weight loads = 24 B/clk/CU at full MFMA rate.
// A workgroup covers a 16 x 16 pixel tile x (CT x 16) output channels: CT = 4 (64 channels: wave = channel tile x upper / lower 8 rows), 2 or 1
// (32 / 16 channels: 4 / 8 row groups); wider layers run as several channel blocks (the input is re-read per block, from L2).
//
// MODE 1..4 is the 3x3 / stride-2 transposed convolution as its four output-PARITY sub-convolutions (o = 2 i + a): output
// pixel (2i + py, 2j + px) only receives taps a = py (mod 2), b = px (mod 2) -- 4, 2, 2 and 1 of the 9 -- so four launches on the INPUT grid
// with exactly those taps replace the 3x3 conv over a zero-stuffed 2H x 2W image: a quarter of the MACs, no stuffed tensor.
#include "dd_common.h"

namespace {

struct KsP {
  const void* x; const void* wp; const float* bias; const void* mask; void* y;
  int ldx, ldy, ldmask, cinv, n_end, n0, n_pad, k_pad, nbias, nslices;
  int B, H, W, tiles_x, tiles_y, nblk, ksplit;
  int relu, accum, out_mul, out_py, out_px, Hout, Wout;
};
// One launch = up to KS_MAX_SUB sub-problems over the same input and weight image (blockIdx.y picks one): the channel blocks of a layer (96 output
// channels = 64 + 32) and the four output parities of a transposed conv run side by side instead of as 2 ... 8 launches of a few dozen workgroups.
constexpr int KS_MAX_SUB = 8;
struct KsSub { int mode, ct, n0, n_end, nblk, ksplit; };
struct KsMulti { KsP p; KsSub sub[KS_MAX_SUB]; int in_relu, gather; };

#ifndef KS_DMA_SPAN
#define KS_DMA_SPAN 8      // eighths of a unit's fragment steps over which the next unit's DMA pieces are issued
#endif
typedef uint32_t ks_u32x4 __attribute__((ext_vector_type(4)));
constexpr int KS_PW = DD_TILE + 2, KS_PH = DD_TILE + 2, KS_CH = (KS_PW * KS_PH + 7) / 8, KS_BUF = KS_CH * 1024;      // 41 KiB per buffer
// Round 5: a workgroup of <= 32 output channels (CT = 1, 2) shares a K-slice's weights through LDS -- 9 taps x CT x 16 rows x 128 bytes arrive by
// LDS-DMA once per workgroup, double buffered behind the two input buffers, and every wave reads its 18 fragments from there at the top of a unit.
// With the register path all eight waves of a 16-channel workgroup requested the SAME 18 KiB per slice from L2 (144 KiB per unit and CU next to 41 KiB
// of input; every mode's own tap set): the knock-outs of tools/ks_knockouts.sh put that stream, not MFMA or LDS, at 25 - 35 % of the kernel's time.
#ifndef KS_WLDS
#define KS_WLDS 1
#endif
#ifndef KS_GATHER_EARLY
#define KS_GATHER_EARLY 1      // gather epilogue operands requested at the top of the tile's last unit (narrow forms)
#endif
#ifndef KS_WLDS_CT
#define KS_WLDS_CT 2      // widest workgroup (in 16-channel tiles) that takes the LDS path
#endif
constexpr int KS_WBUF_MAX = 2 * 18 * 1024;      // CT = 2: 36 KiB per weight buffer

__device__ __forceinline__ void ks_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 ks_lds16(unsigned off) {
  const ks_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) ks_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}

struct KsTile { int b, y0, x0; bool live; };

// taps of one axis: P = -1 all three (plain conv); P = 0 / 1: the taps of output parity P of the stride-2 transposed conv.  Tap t reads the input at
// offset t - 1.  In the packed image of the zero-stuffed form (flipped kernel: image tap u = K[2 - u]) parity 0 uses image taps 0 and 2 at input
// offsets -1 and 0 (kernel taps t = 0, 1 here), parity 1 image tap 1 at offset 0 (t = 1).
// P = 2 (MODE 6): taps 1 and 2 (input offsets 0 and +1) of the image as it is: the 2 x 2-tap conv over a space-to-depth tensor that is the data
// gradient of the transposed conv (see dd_conv3x3_ks mode 6).
constexpr int ks_t0(int p) { return p >= 1 ? 1 : 0; }
constexpr int ks_t1(int p) { return (p < 0 || p == 2) ? 2 : 1; }
constexpr int ks_src(int p, int t) { return (p < 0 || p == 2) ? t : (p == 0 ? (t == 0 ? 0 : 2) : 1); }

// GATHER (mode 0 only): the epilogue of a data gradient in gather form -- y = y + (mask > 0 ? sum : 0), no bias, no activation: the gradient of
// a channel range of a dense-block buffer from ALL the later convs of the block at once (their output gradients are one contiguous channel range =
// the reduction), masked by the ReLU its consumers apply on read, added to what the consumers outside the block already stored, rounded once.
template <typename T, int CT, int MODE, bool IN_RELU, bool GATHER = false>
__device__ __forceinline__ void conv_ks_body(const KsP& a, char* smem, int block) {
  static_assert(sizeof(T) == 2, "K-streamed conv: bf16 / fp16 storage");
  static_assert(CT == 1 || CT == 2 || CT == 4, "1, 2 or 4 output-channel tiles per workgroup");
  constexpr int PY = MODE == 0 ? -1 : MODE == 6 ? 2 : (MODE - 1) >> 1, PX = MODE == 0 ? -1 : MODE == 6 ? 2 : (MODE - 1) & 1;
  constexpr int TY0 = ks_t0(PY), NTY = ks_t1(PY) - TY0 + 1, TX0 = ks_t0(PX), NTX = ks_t1(PX) - TX0 + 1;
  constexpr int RG = 8 / CT, RH = DD_TILE / RG;       // row groups per tile; output rows per wave
  constexpr int NYY = RH + NTY - 1;                   // haloed rows a wave reads (TY0 .. TY0 + NYY - 1 of its band)
  constexpr int FRW = NTX * 2, NF = NYY * FRW;        // fragments per haloed row (dx x K-chunk), per unit
  constexpr int NPIECE = (KS_CH + 7) / 8;             // DMA pieces per wave and unit
  constexpr int NWL = NTY * NTX * 2;                  // weight fragments per slice
  constexpr bool WLDS = KS_WLDS && CT <= KS_WLDS_CT;      // the slice's weights through LDS (shared by the workgroup) instead of per-wave register loads
  constexpr int WCH = NTY * NTX * CT * 2, NWP = WLDS ? (WCH + 7) / 8 : 0, WBUF = WCH * 1024;      // 1-KiB weight chunks per slice (tap x 8-row group); pieces per wave
  constexpr int PW = KS_PW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int blk = block / a.ksplit, ks = block - blk * a.ksplit;
  const int per_img = a.tiles_x * a.tiles_y, total = a.B * per_img;
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
  const int tile0 = (ks % xcd_n) * (a.ksplit / xcd_n) + ks / xcd_n;
  const int NS = a.nslices;
  auto tile_at = [&](int tile) {
    KsTile t;
    t.live = tile < total;
    const int u = t.live ? tile : 0;
    t.b = u / per_img;
    const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
    t.y0 = ty * DD_TILE; t.x0 = (rem - ty * a.tiles_x) * DD_TILE;
    return t;
  };
  // ---- DMA of one unit = K-slice `sl` of tile `t`: chunk id = k*8 + wave holds pixels id*8 + r of the haloed tile, logical slot (lane & 7) ^ r
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* X = reinterpret_cast<const char*>(a.x);
  auto piece = [&](int k, const KsTile& t, int sl, unsigned buf) {
    const int id = k * 8 + wave;
    if (id < KS_CH) {      // wave-uniform
      int rr = r;
      asm volatile("" : "+v"(rr));      // (keeps the per-piece coordinates from being hoisted out of the unit loop: see csrc/dd_conv_bwd.hip)
      const int pix = id * 8 + rr;
      const int py = (pix * 3641) >> 16, px = pix - py * PW;      // pix / 18 for pix < 400
      const int gy = t.y0 - 1 + py, gx = t.x0 - 1 + px, ch = sl * 64 + ls * 8;
      const bool ok = t.live && ch < a.cinv && pix < PW * KS_PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const char* src = X + ((((long)t.b * a.H + gy) * a.W + gx) * a.ldx + ch) * 2;
#ifndef KS_EXP_NO_DMA      // (KS_EXP_*: knock-outs for tools/build_variant.sh experiment builds, never the shipped library)
      ks_dma_1k(ok ? src : zero, buf + id * 1024);
#endif
    }
  };
  // ---- this wave's output: channel tile (wave % CT) of block blk, rows RH * (wave / CT) .. + RH - 1
  const int li = lane & 15, q = lane >> 4;
  const int cot = blk * CT + wave % CT, half = wave / CT;
  const int ch0 = a.n0 + cot * 16;
  const bool active = ch0 < a.n_end;
  const int nrow = ch0 + li, c4 = ch0 + q * 4;
  const T* Wp = reinterpret_cast<const T*>(a.wp);
  const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
  // ---- WLDS: weight chunk id = k*8 + wave = tap * (CT*2) + g holds rows g*8 + r of the block's CT*16 output channels, K-slice sl (64 values = 8 slots
  //      of 16 bytes), slot (lane & 7) of row r holding logical slot (lane & 7) ^ r like the input tile: a fragment read (16 rows x one slot) is conflict-free
  const unsigned wlds_base = lds_base + 2 * KS_BUF;
  auto wpiece = [&](int k, int sl, unsigned wbuf) {
    const int id = k * 8 + wave;
    if (WLDS && id < WCH) {      // wave-uniform
      int rr = r;
      asm volatile("" : "+v"(rr));
      const int tap = id / (CT * 2), g = id - tap * (CT * 2), ty = tap / NTX, tx = tap - ty * NTX;      // tap: index into THIS mode's tap set
      const int srct = ks_src(PY, TY0 + ty) * 3 + ks_src(PX, TX0 + tx);
      const int row = a.n0 + blk * (CT * 16) + g * 8 + rr, k0 = sl * 64 + ((lane & 7) ^ rr) * 8;
      const bool ok = row < a.n_pad && k0 < a.k_pad;
#ifndef KS_EXP_NO_W
      ks_dma_1k(ok ? reinterpret_cast<const char*>(Wp + ((long)srct * a.n_pad + row) * a.k_pad + k0) : zero, wbuf + id * 1024);
#endif
    }
  };
  const unsigned wrd = wlds_base + ((wave % CT) * 16 + li) * 128 + ((q ^ (li & 7)) << 4);      // this lane's 16 bytes of (tap 0, K-chunk 0) in weight buffer 0
  // weight fragment w of slice sl: w = (ty * NTX + tx) * 2 + kc
  auto load_w = [&](int w, int sl) {
    const int kc = w & 1, tx = (w >> 1) % NTX, ty = (w >> 1) / NTX;
    const int srct = ks_src(PY, TY0 + ty) * 3 + ks_src(PX, TX0 + tx);
    const int k0 = sl * 64 + kc * 32 + q * 8;
    const bool ok = active && nrow < a.n_pad && k0 < a.k_pad;
#ifdef KS_EXP_NO_W
    return uint4{(unsigned)k0, (unsigned)srct, 0u, 0u};
#else
    return *reinterpret_cast<const uint4*>(ok ? Wp + ((long)srct * a.n_pad + nrow) * a.k_pad + k0 : zw);
#endif
  };
  uint4 wA[NWL], wB[NWL];
  if constexpr (WLDS) {
#pragma unroll
    for (int k = 0; k < NWP; ++k) wpiece(k, 0, wlds_base);
  } else {
#pragma unroll
    for (int w = 0; w < NWL; ++w) wA[w] = load_w(w, 0);
  }
  float bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bv[e] = (!GATHER && a.bias && active && c4 + e < a.nbias) ? a.bias[c4 + e] : 0.f;
  unsigned d0[8];      // haloed pixel (RH*half + yy)*18 + dx + li of the CURRENT buffer
#pragma unroll
  for (int c = 0; c < 8; ++c) d0[c] = lds_base + (half * RH * PW + li) * DD_LDS_ROW + ((q ^ ((half * RH * PW + li + c) & 7)) << 4);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const bool ch_ok = active && c4 < a.n_end;
  const long yrow = (long)a.Wout * a.ldy * a.out_mul;

  const int mine = tile0 < total ? (total - tile0 + a.ksplit - 1) / a.ksplit : 0;
  const int nunits = mine * NS;
  KsTile cur = tile_at(tile0);
  int tile = tile0, sl = 0, sel = 0;
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) piece(k, cur, 0, lds_base);
  f32x4_t acc[RH];

  // one unit: multiply slice `sl` of tile `cur` with the weights wc; meanwhile request the next unit's input tile and its weights (into wn)
  auto unit = [&](uint4 (&wc)[NWL], uint4 (&wn)[NWL]) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA chunks and weight fragments of this unit have landed (and its stores)
#ifndef KS_EXP_NO_BAR
    __syncthreads();
#endif
    const bool last = sl == NS - 1;
    const int nsl = last ? 0 : sl + 1;
    // (component-wise: a select between whole structs lives in scratch memory)
    const KsTile adv = tile_at(tile + a.ksplit);
    KsTile nxt;
    nxt.b = last ? adv.b : cur.b; nxt.y0 = last ? adv.y0 : cur.y0; nxt.x0 = last ? adv.x0 : cur.x0; nxt.live = last ? adv.live : cur.live;
    const unsigned nbuf = lds_base + (sel ^ 1) * KS_BUF, nwbuf = wlds_base + (sel ^ 1) * WBUF;
    if (!active) {
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) piece(k, nxt, nsl, nbuf);
#pragma unroll
      for (int k = 0; k < NWP; ++k) wpiece(k, nsl, nwbuf);
    } else {
      // WLDS: this unit's 18 fragments come from the workgroup's copy (landed before the barrier above): the six of tap row 0 now, tap row t + 1 during
      // the fragment steps of haloed row t, a full row ahead of its first use (all 18 up front put 23 LDS reads in front of every unit's first MFMA)
      auto wread = [&](int w) { return ks_lds16((wrd + sel * WBUF + (w >> 1) * (CT * 2048)) ^ ((w & 1) << 6)); };
      if constexpr (WLDS) {
#pragma unroll
        for (int w = 0; w < FRW; ++w) wc[w] = wread(w);
      }
      if (sl == 0) {
#pragma unroll
        for (int y = 0; y < RH; ++y) acc[y] = f32x4_t{bv[0], bv[1], bv[2], bv[3]};
      }
      // GATHER on the narrow forms (registers to spare since the weights come through LDS): the mask and the stored gradient of the tile's rows are
      // requested HERE, a whole unit of MFMAs ahead of the epilogue that needs them -- with a short reduction (one slice: the 16 -> 16 gathers of the
      // light Tiramisu) a unit was ~1 us of MFMAs followed by an exposed ~2 us round trip for these two loads.
      constexpr bool EARLY = GATHER && WLDS && KS_GATHER_EARLY;
      uint2 emv[EARLY ? RH : 1], eov[EARLY ? RH : 1];
      if constexpr (EARLY) {
        const bool has_mask = a.mask != nullptr;
        const bool col_ok = ch_ok && cur.x0 + li < a.W;
        const T* mp = reinterpret_cast<const T*>(a.mask) + (((long)cur.b * a.H + cur.y0 + half * RH) * a.W + cur.x0 + li) * a.ldmask + c4;
        const T* op = Y + (((long)cur.b * a.Hout + cur.y0 + half * RH) * a.Wout + cur.x0 + li) * a.ldy + c4;      // (GATHER: output on the input grid)
#pragma unroll
        for (int y = 0; y < RH; ++y) {
          const bool ok = last && col_ok && cur.y0 + half * RH + y < a.H;
          emv[y] = (ok && has_mask) ? *reinterpret_cast<const uint2*>(mp + y * (long)a.W * a.ldmask) : uint2{0u, 0u};
          eov[y] = (ok && a.accum) ? *reinterpret_cast<const uint2*>(op + y * yrow) : uint2{0u, 0u};
        }
      }
      constexpr int RING = 6, AHEAD = RING - 1;
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = TY0 + f / FRW, j = f % FRW, dx = TX0 + j / 2, kc = j & 1, C = yy * PW + dx;
#ifdef KS_EXP_NO_LDS
        return uint4{d0[C & 7], (unsigned)kc, 0u, 0u};
#else
        return ks_lds16((d0[C & 7] ^ (kc << 6)) + C * DD_LDS_ROW);
#endif
      };
#pragma unroll
      for (int f = 0; f < AHEAD && f < NF; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yi = 0; yi < NYY; ++yi) {
#pragma unroll
        for (int j = 0; j < FRW; ++j) {
          const int f = yi * FRW + j, txi = j / 2, kc = j & 1;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if constexpr (WLDS) {
            if (yi + 1 < NTY) wc[(yi + 1) * FRW + j] = wread((yi + 1) * FRW + j);
          }
          {      // the next unit's DMA pieces and weight fragments, spread evenly over the NF steps: piece k goes with step (k NF) / NPIECE.
                 // (A narrow wave -- 2 output rows, one tap -- has FEWER steps than pieces: several pieces per step, none may be dropped.)
            constexpr int SPAN = NF * KS_DMA_SPAN / 8 > NPIECE ? NF * KS_DMA_SPAN / 8 : NF;
#pragma unroll
            for (int k = 0; k < NPIECE; ++k)
              if ((k * SPAN) / NPIECE == f) piece(k, nxt, nsl, nbuf);
            if constexpr (WLDS) {
#pragma unroll
              for (int k = 0; k < NWP; ++k)
                if ((k * NF) / NWP == f) wpiece(k, nsl, nwbuf);
            } else {
#pragma unroll
              for (int w = 0; w < NWL; ++w)
                if ((w * NF) / NWL == f) wn[w] = load_w(w, nsl);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          const uint4 b = IN_RELU ? relu16<T>(ring[f % RING]) : ring[f % RING];
#pragma unroll
          for (int tyi = 0; tyi < NTY; ++tyi) {
            const int y = yi - tyi;
#ifdef KS_EXP_NO_MFMA
            if (y >= 0 && y < RH) acc[y][0] += __uint_as_float(wc[(tyi * NTX + txi) * 2 + kc].x ^ b.x);
#else
            if (y >= 0 && y < RH) acc[y] = mma16<T>(wc[(tyi * NTX + txi) * 2 + kc], b, acc[y]);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (last) {      // bias is in the accumulators; ReLU, round, store: 4 channels = 8 bytes per lane and row
        const bool col_ok = ch_ok && cur.x0 + li < a.W;
        const int oy0 = (cur.y0 + half * RH) * a.out_mul + a.out_py, ox = (cur.x0 + li) * a.out_mul + a.out_px;
        T* yp = Y + (((long)cur.b * a.Hout + oy0) * a.Wout + ox) * a.ldy + c4;
        if (GATHER) {      // y = (accumulate ? y : 0) + (mask ? (mask > 0 ? sum : 0) : sum), rounded once
          const bool has_mask = a.mask != nullptr;
          const T* mp = reinterpret_cast<const T*>(a.mask) + (((long)cur.b * a.H + cur.y0 + half * RH) * a.W + cur.x0 + li) * a.ldmask + c4;
          const long mrow = (long)a.W * a.ldmask;
          constexpr int GB = RH < 4 ? RH : 4;      // rows per batch: their loads first (one exposed round trip per batch), then the arithmetic
#pragma unroll
          for (int yb = 0; yb < RH; yb += GB) {
            uint2 mv[GB], ov[GB];
#pragma unroll
            for (int y = 0; y < GB; ++y) {
              const bool ok = col_ok && cur.y0 + half * RH + yb + y < a.H;
              if constexpr (EARLY) { mv[y] = emv[yb + y]; ov[y] = eov[yb + y]; continue; }
              mv[y] = (ok && has_mask) ? *reinterpret_cast<const uint2*>(mp + (yb + y) * mrow) : uint2{0u, 0u};
              ov[y] = (ok && a.accum) ? *reinterpret_cast<const uint2*>(yp + (yb + y) * yrow) : uint2{0u, 0u};
            }
#pragma unroll
            for (int y = 0; y < GB; ++y) {
              float m8[8], o8[8];
              unpack8t<T>(uint4{mv[y].x, mv[y].y, 0u, 0u}, m8);
              unpack8t<T>(uint4{ov[y].x, ov[y].y, 0u, 0u}, o8);
              const f32x4_t v = acc[yb + y];
              uint2 o2;
              o2.x = pack2<T>(o8[0] + ((!has_mask || m8[0] > 0.f) ? v[0] : 0.f), o8[1] + ((!has_mask || m8[1] > 0.f) ? v[1] : 0.f));
              o2.y = pack2<T>(o8[2] + ((!has_mask || m8[2] > 0.f) ? v[2] : 0.f), o8[3] + ((!has_mask || m8[3] > 0.f) ? v[3] : 0.f));
              if (col_ok && cur.y0 + half * RH + yb + y < a.H) *reinterpret_cast<uint2*>(yp + (yb + y) * yrow) = o2;
            }
          }
        } else {
#pragma unroll
          for (int y = 0; y < RH; ++y) {
            const f32x4_t v = acc[y];
            uint2 o2;
            o2.x = pack2<T>(v[0], v[1]);
            o2.y = pack2<T>(v[2], v[3]);
            if (a.relu) { o2.x = relu_bf16x2(o2.x); o2.y = relu_bf16x2(o2.y); }
            if (col_ok && cur.y0 + half * RH + y < a.H) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;
          }
        }
      }
    }
    const int flip = sel ? -KS_BUF : KS_BUF;
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] += flip;
    sel ^= 1;
    tile = last ? tile + a.ksplit : tile;
    cur.b = nxt.b; cur.y0 = nxt.y0; cur.x0 = nxt.x0; cur.live = nxt.live;
    sl = nsl;
  };
  for (int u = 0; u < nunits; u += 2) {
    unit(wA, wB);
    if (u + 1 < nunits) unit(wB, wA);
  }
}

template <typename T>
__global__ __launch_bounds__(512) void conv_ks_kernel(const KsMulti m) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const KsSub sb = m.sub[blockIdx.y];
  if ((int)blockIdx.x >= sb.nblk * sb.ksplit) return;      // (whole workgroup: no barrier is skipped by part of it)
  KsP a = m.p;
  a.n0 = sb.n0; a.n_end = sb.n_end; a.nblk = sb.nblk; a.ksplit = sb.ksplit;
  const bool plain = sb.mode == 0 || sb.mode == 6;      // output on the input grid
  a.out_mul = plain ? 1 : 2; a.out_py = plain ? 0 : (sb.mode - 1) >> 1; a.out_px = plain ? 0 : (sb.mode - 1) & 1;
  a.Hout = a.H * a.out_mul; a.Wout = a.W * a.out_mul;
  const int b = blockIdx.x;
#define KS_CASE(CT_, MODE_, RELU_) conv_ks_body<T, CT_, MODE_, RELU_>(a, smem, b)
#define KS_MODES(CT_)                                                      \
  switch (sb.mode) {                                                       \
    case 0: if (m.gather) conv_ks_body<T, CT_, 0, false, true>(a, smem, b);       \
            else if (m.in_relu) KS_CASE(CT_, 0, true); else KS_CASE(CT_, 0, false); break; \
    case 1: KS_CASE(CT_, 1, false); break;                                 \
    case 2: KS_CASE(CT_, 2, false); break;                                 \
    case 3: KS_CASE(CT_, 3, false); break;                                 \
    case 4: KS_CASE(CT_, 4, false); break;                                 \
    default: if (m.gather) conv_ks_body<T, CT_, 6, false, true>(a, smem, b); else KS_CASE(CT_, 6, false); break; \
  }
  if (sb.ct == 4) { KS_MODES(4) } else if (sb.ct == 2) { KS_MODES(2) } else { KS_MODES(1) }
#undef KS_MODES
#undef KS_CASE
}

static int ks_cus() {
  return dd_device_cus();
}

}  // namespace

bool dd_conv_pw_taps_eligible(const dd_conv_ks_args* a);
int dd_conv_pw_taps_launch(const dd_conv_ks_args* a, hipStream_t stream);

extern "C" int dd_conv3x3_ks(const dd_conv_ks_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->x && a->wp && a->y, "dd_conv3x3_ks: null pointer");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_conv3x3_ks: storage dtype must be DD_BF16 or DD_F16");
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->cin > 0 && a->n > 0 && a->n0 >= 0, "dd_conv3x3_ks: empty problem");
  DD_REQUIRE(a->mode >= 0 && a->mode <= 6, "dd_conv3x3_ks: mode %d (0: 3x3 SAME conv; 1..4: output parity (py, px) = ((mode-1)/2, (mode-1)%%2) of the 3x3/s2 transposed conv; 5: all four; 6: image taps (1..2, 1..2) at offsets 0 / +1)", a->mode);
  DD_REQUIRE(a->mode == 0 || !(a->flags & DD_IN_RELU), "dd_conv3x3_ks: only mode 0 takes an input ReLU");
  DD_REQUIRE((a->flags & ~(DD_IN_RELU | DD_OUT_RELU | DD_ACCUM)) == 0, "dd_conv3x3_ks: flags %d unsupported (DD_IN_RELU | DD_OUT_RELU | DD_ACCUM)", a->flags);
  // gradient epilogue (no bias, no activation): DD_ACCUM and / or a mask tensor
  const bool gather = (a->flags & DD_ACCUM) != 0 || a->mask != nullptr;
  DD_REQUIRE(!gather || ((a->mode == 0 || a->mode == 6) && !(a->flags & (DD_IN_RELU | DD_OUT_RELU)) && (!a->mask || (a->ldmask % 4 == 0 && ((uintptr_t)a->mask % 8) == 0))),
             "dd_conv3x3_ks: the gradient epilogue (DD_ACCUM and / or a mask): mode 0 or 6, mask 8-byte aligned with ldmask %% 4 == 0, no ReLU flags");
  DD_REQUIRE(a->ldx % 8 == 0 && a->ldy % 4 == 0 && a->k_pad % 32 == 0 && a->n_pad % 16 == 0 && a->n0 % 16 == 0 && a->n % 4 == 0,
             "dd_conv3x3_ks: ldx=%d (%%8) ldy=%d (%%4) k_pad=%d (%%32) n_pad=%d (%%16) n0=%d (%%16) n=%d (%%4)", a->ldx, a->ldy, a->k_pad, a->n_pad, a->n0, a->n);
  DD_REQUIRE(a->n0 + a->n <= a->n_pad && a->cin <= a->k_pad, "dd_conv3x3_ks: channel ranges exceed the packed weight image");
  DD_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->wp % 16) == 0 && ((uintptr_t)a->y % 8) == 0, "dd_conv3x3_ks: x / wp must be 16-byte, y 8-byte aligned");
  // mode 6 with many output channels: 256-wide GEMM tiles over the linear pixel index (csrc/dd_conv_pw.hip)
  if (dd_conv_pw_taps_eligible(a)) return dd_conv_pw_taps_launch(a, reinterpret_cast<hipStream_t>(stream));
  KsMulti m;
  KsP& p = m.p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.mask = a->mask; p.y = a->y;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldmask = a->ldmask; p.cinv = (a->cin + 7) / 8 * 8; p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.nbias = a->nbias; p.nslices = (a->cin + 63) / 64;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.relu = (a->flags & DD_OUT_RELU) != 0; p.accum = (a->flags & DD_ACCUM) != 0;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE);
  p.n0 = p.n_end = p.nblk = p.ksplit = p.out_mul = p.out_py = p.out_px = p.Hout = p.Wout = 0;      // per sub-problem, set by the kernel
  m.in_relu = (a->flags & DD_IN_RELU) != 0;
  m.gather = gather ? 1 : 0;
  // channel blocks: every whole 64 channels as workgroup columns of ONE sub-problem (4 tiles each), the remainder (<= 63 channels) as a
  // second one with the narrowest tile count that covers it (a 96-channel layer = 64 + 32)
  int starts[2], widths[2], cts[2], nb = 0;
  const int whole = a->n / 64 * 64, rest = a->n - whole;
  if (whole > 0) { starts[nb] = 0; widths[nb] = whole; cts[nb] = 4; ++nb; }
  if (rest > 0) { starts[nb] = whole; widths[nb] = rest; cts[nb] = rest <= 16 ? 1 : (rest <= 32 ? 2 : 4); ++nb; }
  const int first_mode = a->mode == 5 ? 1 : a->mode, last_mode = a->mode == 5 ? 4 : a->mode;
  const long total = (long)a->B * p.tiles_x * p.tiles_y;
  int ns = 0;
  long gx = 1;
  for (int mode = first_mode; mode <= last_mode; ++mode)
    for (int b = 0; b < nb; ++b) {
      DD_REQUIRE(ns < KS_MAX_SUB, "dd_conv3x3_ks: too many sub-problems");
      KsSub& sb = m.sub[ns++];
      sb.mode = mode; sb.ct = cts[b]; sb.n0 = a->n0 + starts[b]; sb.n_end = sb.n0 + widths[b];
      sb.nblk = dd_ceil_div(widths[b], sb.ct * 16);
      long ksplit = dd_ceil_div(ks_cus(), sb.nblk);
      if (ksplit < 1) ksplit = 1;
      if (ksplit > total) ksplit = total;
      sb.ksplit = (int)ksplit;
      if ((long)sb.nblk * sb.ksplit > gx) gx = (long)sb.nblk * sb.ksplit;
    }
  for (int i = ns; i < KS_MAX_SUB; ++i) m.sub[i] = KsSub{0, 1, 0, 0, 0, 0};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = 2 * (size_t)KS_BUF + (KS_WLDS ? 2 * (size_t)KS_WBUF_MAX : 0);      // two input buffers + two weight buffers (narrow workgroups)
  if (a->dtype == DD_BF16) {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_ks_kernel<bf16_t>));
    hipLaunchKernelGGL(conv_ks_kernel<bf16_t>, dim3((unsigned)gx, (unsigned)ns), dim3(512), smem, s, m);
  } else {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_ks_kernel<f16_t>));
    hipLaunchKernelGGL(conv_ks_kernel<f16_t>, dim3((unsigned)gx, (unsigned)ns), dim3(512), smem, s, m);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}
