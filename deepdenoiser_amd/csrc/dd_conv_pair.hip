// Two consecutive 3x3 SAME convolutions (bias + optional ReLU each) of at most 64 channels as ONE launch, CDNA4: the inference forward of the
// U-Net's 64-channel level (UNet.py:25-48: `number_of_convolutions_per_block` convs in a row; Prediction.py:357-369 runs them forward only).
//
// Why.  A standalone 64 -> 64 layer at 128 x 128 is paced by its bytes: csrc/dd_conv_rw.hip's forward takes 182 us of which 139 us are its loads
// and stores alone at copy speed (the kernel as a pure mover).  At inference nothing needs the tensor between two such
// layers: computed into LDS and consumed from there, a pair moves one input and one output instead of two of each.
//   * Output tile 16 rows x 14 columns.  The second conv needs the first one's output on 18 x 16 pixels (a 1-pixel halo) -- exactly ONE 16-pixel
//     MFMA column tile wide, which is why the tile is 14 wide: the first conv runs 18 row-tiles, the second 16 (14 of 16 lanes used), 2.43 MFMA
//     row-tiles per output row-tile against 2.0 for two standalone layers.  The input tile is 20 x 18 pixels (45 chunks of 1 KiB, LDS-DMA, double
//     buffered), the intermediate 18 x 16 pixels x 64 channels in LDS (37 KiB): 127 KiB in all.
//   * 8 waves = 4 output-channel tiles x 2 row halves, the weights of BOTH layers in registers as the MFMA A operand (2 x 18 fragments = 144
//     registers); a wave walks the haloed rows of its half once per layer with rotating accumulators exactly as conv_rw8_kernel does.
//   * TensorFlow pads every layer with zeros: intermediate pixels outside the image are written as zeros, whatever the first conv gives there.
#include "dd_common.h"

#ifndef PAIR_DMA_SPAN
#define PAIR_DMA_SPAN 8      // eighths of the first conv's fragment steps over which the next tile's DMA pieces are issued
#endif
namespace {

struct PairP {
  const void* x; const void* w1; const float* b1; const void* w2; const float* b2; void* y;
  int ldx, ldy, cinv, n, n_pad1, k_pad1, n_pad2, k_pad2, nb1, nb2;
  int B, H, W, tiles_x, tiles_y, ksplit;
  int relu1, relu2;
};

typedef uint32_t pr_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pr_u32x2 __attribute__((ext_vector_type(2)));
constexpr int PR_OW = 14, PR_OH = 16;                  // output tile
constexpr int PR_IW = 18, PR_IH = 20;                  // input tile: 2-pixel halo
constexpr int PR_MW = 16, PR_MH = 18;                  // intermediate tile: 1-pixel halo, one MFMA column tile wide
constexpr int PR_ICH = (PR_IW * PR_IH + 7) / 8;        // 45 chunks
constexpr int PR_IBUF = PR_ICH * 1024;
constexpr int PR_MBYTES = (PR_MW * PR_MH + 8) * DD_LDS_ROW;      // (+8 pixels: lanes 14 / 15 of the second conv read two columns past a row)

__device__ __forceinline__ void pr_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 pr_lds16(unsigned off) {
  const pr_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) pr_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}

struct PrTile { int b, y0, x0; bool live; };

template <typename T>
__global__ __launch_bounds__(512) void conv_pair_kernel(const PairP a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  static_assert(sizeof(T) == 2, "fused conv pair: bf16 / fp16 storage");
  constexpr int NPIECE = (PR_ICH + 7) / 8;      // 6
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned m_base = lds_base + 2 * PR_IBUF;
  const int ks = blockIdx.x;
  const int per_img = a.tiles_x * a.tiles_y, total = a.B * per_img;
  const int xcd_n = (a.ksplit & 7) == 0 ? 8 : 1;
  const int tile0 = (ks % xcd_n) * (a.ksplit / xcd_n) + ks / xcd_n;
  auto tile_at = [&](int tile) {
    PrTile t;
    t.live = tile < total;
    const int u = t.live ? tile : 0;
    t.b = u / per_img;
    const int rem = u - t.b * per_img, ty = rem / a.tiles_x;
    t.y0 = ty * PR_OH; t.x0 = (rem - ty * a.tiles_x) * PR_OW;
    return t;
  };
  // ---- DMA of the 20 x 18 input tile: chunk id = k*8 + wave holds pixels id*8 + r, logical slot (lane & 7) ^ r; coordinates once per launch
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* X = reinterpret_cast<const char*>(a.x);
  int p_off[NPIECE], p_yx[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int id = k * 8 + wave, pix = id * 8 + r;
    const int py = (pix * 3641) >> 16, px = pix - py * PR_IW;      // pix / 18 for pix < 400
    p_off[k] = ((py * a.W + px) * a.ldx + ls * 8) * 2;
    p_yx[k] = (id < PR_ICH && ls * 8 < a.cinv && pix < PR_IW * PR_IH) ? ((py << 8) | px) : (0x7fff << 8);      // row 32 767: never inside an image
  }
  auto piece = [&](int k, const PrTile& t, unsigned buf) {
    const int id = k * 8 + wave;
    if (id < PR_ICH) {      // wave-uniform
      const int gy = t.y0 - 2 + (p_yx[k] >> 8), gx = t.x0 - 2 + (p_yx[k] & 255);
      const bool ok = t.live && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const long base = ((((long)t.b * a.H + (t.y0 - 2)) * a.W + (t.x0 - 2)) * a.ldx) * 2;
      pr_dma_1k(ok ? X + base + p_off[k] : zero, buf + id * 1024);
    }
  };
  PrTile cur = tile_at(tile0);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) piece(k, cur, lds_base);

  // ---- roles: output-channel tile ct (of either layer), row half
  const int li = lane & 15, q = lane >> 4;
  const int ct = wave & 3, half = wave >> 2;
  const int nrow = ct * 16 + li, c4 = ct * 16 + q * 4;
  uint4 w1f[9][2], w2f[9][2];
  {
    const T* W1 = reinterpret_cast<const T*>(a.w1);
    const T* W2 = reinterpret_cast<const T*>(a.w2);
    const T* zw = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const int k0 = kc * 32 + q * 8;
        w1f[t][kc] = *reinterpret_cast<const uint4*>((nrow < a.n_pad1 && k0 < a.k_pad1) ? W1 + ((long)t * a.n_pad1 + nrow) * a.k_pad1 + k0 : zw);
        w2f[t][kc] = *reinterpret_cast<const uint4*>((nrow < a.n_pad2 && k0 < a.k_pad2) ? W2 + ((long)t * a.n_pad2 + nrow) * a.k_pad2 + k0 : zw);
      }
  }
  float bv1[4], bv2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bv1[e] = (a.b1 && c4 + e < a.nb1) ? a.b1[c4 + e] : 0.f;
    bv2[e] = (a.b2 && c4 + e < a.nb2) ? a.b2[c4 + e] : 0.f;
  }
  constexpr int RA = PR_MH / 2, RB = PR_OH / 2;      // rows per wave: 9 intermediate, 8 output
  // input fragments of the first conv: haloed pixel (RA*half + yy)*18 + dx + li of the CURRENT buffer, eight bases for the swizzle residues
  unsigned d0[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) d0[c] = lds_base + (half * RA * PR_IW + li) * DD_LDS_ROW + ((q ^ ((half * RA * PR_IW + li + c) & 7)) << 4);
  // intermediate fragments of the second conv: pixel (RB*half + yy)*16 + dx + li; the residue of (row*16 + dx + li) mod 8 is (dx + li) mod 8
  unsigned e0[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) e0[dx] = m_base + (half * RB * PR_MW + li) * DD_LDS_ROW + ((q ^ ((li + dx) & 7)) << 4);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const bool ch_ok = c4 < a.n;
  const long yrow = (long)a.W * a.ldy;

  int sel = 0;
  for (int tile = tile0; tile < total; tile += a.ksplit, sel ^= 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's chunks of `tile` have landed
    __syncthreads();                         // ... everyone's; every wave has finished reading the intermediate of the previous tile
    const PrTile nxt = tile_at(tile + a.ksplit);
    const unsigned nbuf = lds_base + (sel ^ 1) * PR_IBUF;
    constexpr int FR = 6, RING = 6, AHEAD = RING - 1;
    // ================= first conv: intermediate rows RA*half .. + 8 (global rows y0 - 1 + ...), columns x0 - 1 + li
    {
      constexpr int PHW = RA + 2, NF = PHW * FR;
      f32x4_t acc[4];
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / FR, j = f - FR * yy, dx = j >> 1, kc = j & 1, C = yy * PR_IW + dx;
        return pr_lds16((d0[C & 7] ^ (kc << 6)) + C * DD_LDS_ROW);
      };
      const int gx = cur.x0 - 1 + li;
      const bool col_in = (unsigned)gx < (unsigned)a.W;
      auto write_row = [&](int y) {
        f32x4_t v = acc[y % 4];
        pr_u32x2 o2;
        o2[0] = pack2<T>(v[0], v[1]);
        o2[1] = pack2<T>(v[2], v[3]);
        if (a.relu1) { o2[0] = relu_bf16x2(o2[0]); o2[1] = relu_bf16x2(o2[1]); }
        const int gy = cur.y0 - 1 + half * RA + y;
        if (!(col_in && (unsigned)gy < (unsigned)a.H)) { o2[0] = 0u; o2[1] = 0u; }      // the second conv's zero padding
        const int m = (half * RA + y) * PR_MW + li;
        *reinterpret_cast<__attribute__((address_space(3))) pr_u32x2*>(m_base + m * DD_LDS_ROW + (((ct * 2 + (q >> 1)) ^ (m & 7)) << 4) + (q & 1) * 8) = o2;
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PHW; ++yy) {
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          const int f = yy * FR + j, dx = j >> 1, kc = j & 1;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < RA) acc[yy % 4] = f32x4_t{bv1[0], bv1[1], bv1[2], bv1[3]};
          if (j == 2 && yy >= 3) write_row(yy - 3);
          {      // the DMA pieces of the next tile, spread over this phase
            constexpr int SPAN = NF * PAIR_DMA_SPAN / 8 > NPIECE ? NF * PAIR_DMA_SPAN / 8 : NPIECE;
            const int k0 = (f * NPIECE + SPAN - 1) / SPAN;
            if (k0 < NPIECE && (k0 * SPAN) / NPIECE == f) piece(k0, nxt, nbuf);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < RA) acc[y % 4] = mma16<T>(w1f[dy * 3 + dx][kc], ring[f % RING], acc[y % 4]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int y = PHW - 3; y < RA; ++y) write_row(y);
    }
    __syncthreads();      // the intermediate tile is complete
    // ================= second conv: output rows RB*half .. + 7, columns x0 + li (li < 14)
    {
      constexpr int PHW = RB + 2, NF = PHW * FR;
      f32x4_t acc[4];
      uint4 ring[RING];
      auto frag = [&](int f) {
        const int yy = f / FR, j = f - FR * yy, dx = j >> 1, kc = j & 1;
        return pr_lds16((e0[dx] ^ (kc << 6)) + (yy * PR_MW + dx) * DD_LDS_ROW);
      };
      const bool col_ok = ch_ok && li < PR_OW && cur.x0 + li < a.W;
      T* yp = Y + (((long)cur.b * a.H + cur.y0 + half * RB) * a.W + cur.x0 + li) * a.ldy + c4;
      auto write_row = [&](int y) {
        f32x4_t v = acc[y % 4];
        uint2 o2;
        o2.x = pack2<T>(v[0], v[1]);
        o2.y = pack2<T>(v[2], v[3]);
        if (a.relu2) { o2.x = relu_bf16x2(o2.x); o2.y = relu_bf16x2(o2.y); }
        if (col_ok && cur.live && cur.y0 + half * RB + y < a.H) *reinterpret_cast<uint2*>(yp + y * yrow) = o2;
      };
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) ring[f] = frag(f);
#pragma unroll
      for (int yy = 0; yy < PHW; ++yy) {
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          const int f = yy * FR + j, dx = j >> 1, kc = j & 1;
          if (f + AHEAD < NF) ring[(f + AHEAD) % RING] = frag(f + AHEAD);
          if (j == 0 && yy < RB) acc[yy % 4] = f32x4_t{bv2[0], bv2[1], bv2[2], bv2[3]};
          if (j == 2 && yy >= 3) write_row(yy - 3);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int y = yy - dy;
            if (y >= 0 && y < RB) acc[y % 4] = mma16<T>(w2f[dy * 3 + dx][kc], ring[f % RING], acc[y % 4]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int y = PHW - 3; y < RB; ++y) write_row(y);
    }
    const int flip = sel ? -PR_IBUF : PR_IBUF;
#pragma unroll
    for (int c = 0; c < 8; ++c) d0[c] += flip;
    cur = nxt;
  }
}

int pr_cus() {
  return dd_device_cus();
}

template <typename T>
void pair_launch(const PairP& p, hipStream_t s) {
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_pair_kernel<T>));
  hipLaunchKernelGGL(conv_pair_kernel<T>, dim3((unsigned)p.ksplit), dim3(512), 2 * (size_t)PR_IBUF + PR_MBYTES, s, p);
}

}  // namespace

extern "C" int dd_conv3x3_pair(const dd_conv_pair_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->x && a->w1 && a->w2 && a->y, "dd_conv3x3_pair: null pointer");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_conv3x3_pair: storage dtype must be DD_BF16 or DD_F16");
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "dd_conv3x3_pair: empty grid");
  DD_REQUIRE(a->cin > 0 && a->cin <= 64 && a->cin % 8 == 0 && a->cmid > 48 && a->cmid <= 64 && a->cout > 0 && a->cout <= 64 && a->cout % 4 == 0,
             "dd_conv3x3_pair: cin=%d (<= 64, %%8) cmid=%d (49..64) cout=%d (<= 64, %%4)", a->cin, a->cmid, a->cout);
  DD_REQUIRE(a->k_pad1 % 32 == 0 && a->k_pad1 >= a->cin && a->k_pad1 <= 64 && a->n_pad1 == 64 && a->k_pad2 == 64 && a->n_pad2 % 16 == 0 && a->n_pad2 >= a->cout && a->n_pad2 <= 64,
             "dd_conv3x3_pair: packed images must be [9][64][32 | 64] and [9][<= 64][64] (k_pad1=%d n_pad1=%d k_pad2=%d n_pad2=%d)", a->k_pad1, a->n_pad1, a->k_pad2, a->n_pad2);
  DD_REQUIRE(a->ldx % 8 == 0 && a->ldy % 4 == 0 && ((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->y % 8) == 0 && ((uintptr_t)a->w1 % 16) == 0 && ((uintptr_t)a->w2 % 16) == 0,
             "dd_conv3x3_pair: ldx=%d (%%8) ldy=%d (%%4), x / weights 16-byte, y 8-byte aligned", a->ldx, a->ldy);
  DD_REQUIRE((a->flags1 & ~DD_OUT_RELU) == 0 && (a->flags2 & ~DD_OUT_RELU) == 0, "dd_conv3x3_pair: only DD_OUT_RELU per layer");
  PairP p;
  p.x = a->x; p.w1 = a->w1; p.b1 = a->bias1; p.w2 = a->w2; p.b2 = a->bias2; p.y = a->y;
  p.ldx = a->ldx; p.ldy = a->ldy; p.cinv = a->cin; p.n = a->cout;
  p.n_pad1 = a->n_pad1; p.k_pad1 = a->k_pad1; p.n_pad2 = a->n_pad2; p.k_pad2 = a->k_pad2; p.nb1 = a->bias1 ? a->cmid : 0; p.nb2 = a->bias2 ? a->cout : 0;
  p.B = a->B; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, PR_OW); p.tiles_y = dd_ceil_div(a->H, PR_OH);
  const long total = (long)a->B * p.tiles_x * p.tiles_y;
  long ksplit = pr_cus();
  if (ksplit > total) ksplit = total;
  p.ksplit = (int)ksplit;
  p.relu1 = (a->flags1 & DD_OUT_RELU) != 0; p.relu2 = (a->flags2 & DD_OUT_RELU) != 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == DD_BF16) pair_launch<bf16_t>(p, s); else pair_launch<f16_t>(p, s);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
