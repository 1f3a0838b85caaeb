// 1x1 convolution as a plain GEMM over the LINEAR pixel index, CDNA4: the transition layers and the head of the Tiramisu backbone (BASELINE cfg-3).
//
// Reference seam (file:line in /root/reference): TensorFlow/Tiramisu.py:43-58 (`__transition_down`: 1x1 conv over the whole concat, C -> C with
// C = 80 ... 704), Tiramisu.py:118-124 (the 1x1 output conv) and the data gradients TensorFlow's autodiff derives for them (Training.py:701-702).
//
// Why another kernel.  A 1x1 layer has no halo and no tap reuse: per byte of input it does N MACs and nothing else, so the only lever is how
// many output channels one pass over the input covers.  csrc/dd_conv_igemm.hip covers 64 per pass (704 -> 704 = 11 passes over a 184 MB tensor:
// 485 us, all of it re-reading).  Here a workgroup owns 256 pixels x up to 256 output channels (8 waves = 4 pixel groups x 2 channel groups; a
// wave keeps 64 pixels x CTN x 16 channels = up to 128 accumulator registers), the reduction streams in 64-channel slices and BOTH operands of
// a slice arrive by LDS-DMA (256 + 256 rows of 128 B, double buffered).  Per slice and CU: 64 KiB written to and 192 KiB read from LDS against
// 2 048 MFMA cycles -- the LDS port is the ceiling (~75 %), which is above what HBM allows anyway: at 256 channels per pass the input stream
// alone needs 4096 / 256 = 16 B/clk/CU at full MFMA rate, 2.5x the HBM share of a CU.  The channel blocks of one pixel tile are handed to
// workgroups of the SAME XCD next to each other in time, so that the second and third pass over a tile hit that XCD's L2.
#include "dd_common.h"

namespace {

struct PwP {
  const void* x; const void* wp; const float* bias; const void* mask; void* y;
  int ldx, ldy, ldmask, cinv, n, n_pad, k_pad, nbias, nslices;
  long M;                       // pixels (B * H * W)
  int mtiles, nblk, nb_rows;    // pixel tiles of 256, channel blocks, channels per block (2 * CTN * 16)
  int relu, accum;
  // ntaps == 4: the 2 x 2-tap conv over a space-to-depth tensor (data gradient of the 3x3 / s2 transposed conv, dd_conv3x3_ks mode 6): K-slice
  // sl = (tap t, 64-channel chunk), tap t reads pixel (i + (t >> 1), j + (t & 1)) of the H x W grid (zero outside) against image tap
  // (1 + (t >> 1)) * 3 + 1 + (t & 1) of a [9][n_pad][k_pad] weight image.  ntaps == 1: the plain 1x1 layer.
  int ntaps, spt, H, W;         // taps, K-slices per tap, grid
  long tap_stride;              // elements between image taps
  // ntaps == 4: the K-slices that hold data.  The operand of mode 6 is block-sparse -- tap (dy, dx) only meets the parity planes with py <= 1 - dy,
  // px <= 1 - dx (9 of 16 (tap, plane) blocks) -- so the host lists the (tap, 64-channel chunk) pairs that intersect a non-zero block
  unsigned char sl_tap[40], sl_chunk[40];
};

typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));
constexpr int PW_BM = 256, PW_XBYTES = PW_BM * DD_LDS_ROW, PW_WBYTES = 256 * DD_LDS_ROW, PW_BUF = PW_XBYTES + PW_WBYTES;      // 64 KiB per buffer

__device__ __forceinline__ void pw_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint4 pw_lds16(unsigned off) {
  const pw_u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) pw_u32x4*>(off);
  return uint4{v[0], v[1], v[2], v[3]};
}
typedef unsigned pw_u32x2 __attribute__((ext_vector_type(2)));
// rows of 16 lanes r0..r3: swap32: a = [a.r0 a.r1 b.r0 b.r1], b = [a.r2 a.r3 b.r2 b.r3];  swap16: a = [a.r0 b.r0 a.r2 b.r2], b = [a.r1 b.r1 a.r3 b.r3]
__device__ __forceinline__ void pw_swap32(f32x4_t& a, f32x4_t& b) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const pw_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
    a[e] = __uint_as_float(r[0]); b[e] = __uint_as_float(r[1]);
  }
}
__device__ __forceinline__ void pw_swap16(f32x4_t& a, f32x4_t& b) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const pw_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
    a[e] = __uint_as_float(r[0]); b[e] = __uint_as_float(r[1]);
  }
}
__device__ __forceinline__ int pw_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// y[p][n] = epilogue( sum_k W[n][k] * (IN_RELU ? relu(x[p][k]) : x[p][k]) ):  + bias, ReLU, (mask > 0 ? . : 0), + old y, rounded once.
template <typename T, int CTN, bool IN_RELU>
__global__ __launch_bounds__(512) void conv_pw_kernel(const PwP a) {
  static_assert(sizeof(T) == 2, "1x1 GEMM: bf16 / fp16 storage");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // ---- work items: (pixel tile, channel block).  Workgroup g lives on XCD g % 8 (round-robin dispatch); XCD x owns the pixel tiles x, x + 8, ...
  // and its workgroups walk (tile, block) pairs in order, block fastest: the blocks of a tile run side by side on one L2.
  const int G = gridDim.x, xcd_n = (G & 7) == 0 ? 8 : 1;
  const int xcd = blockIdx.x % xcd_n, j0 = blockIdx.x / xcd_n, per = G / xcd_n;
  const int ntile_x = xcd < a.mtiles ? (a.mtiles - xcd + xcd_n - 1) / xcd_n : 0;
  const int items = ntile_x * a.nblk;
  const int mine = j0 < items ? (items - j0 + per - 1) / per : 0;
  const int NS = a.nslices;
  const int nunits = mine * NS;
  if (nunits == 0) return;

  // ---- DMA of one unit = K-slice `sl` of item `it`: chunk id holds rows id*8 + r, logical slot (lane & 7) ^ r
  const int r = lane >> 3, ls = (lane & 7) ^ r;
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* X = reinterpret_cast<const char*>(a.x);
  const char* Wp = reinterpret_cast<const char*>(a.wp);
  constexpr int WCH = 4 * CTN;      // weight chunks per block (2 * CTN * 16 rows / 8)
  // this lane's byte offsets inside a pixel tile / a weight block are the same for every unit: computed once; per unit a SCALAR base is added
  // (recomputed per chunk they are two 64-bit multiply-adds each: 16 quarter-rate vector instructions per unit next to its 64 MFMAs)
  constexpr int WPC = (WCH + 7) / 8;
  int xo[4], wo[WPC], xrow[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { xrow[c] = (c * 8 + wave) * 8 + r; xo[c] = (xrow[c] * a.ldx + ls * 8) * 2; }
#pragma unroll
  for (int c = 0; c < WPC; ++c) wo[c] = (((c * 8 + wave) * 8 + r) * a.k_pad + ls * 8) * 2;
  auto dma = [&](int it, int sl, unsigned buf) {
    const int pt = (it / a.nblk) * xcd_n + xcd, nb = it - (it / a.nblk) * a.nblk;
    if (a.ntaps == 1) {      // the plain 1x1 layer: no pixel decomposition
      const int k0 = sl * 64;
      const long pix0 = (long)pt * PW_BM;
      const char* xb = X + (pix0 * a.ldx + k0) * 2;
      const char* wb = Wp + ((long)nb * a.nb_rows * a.k_pad + k0) * 2;
      const bool k_ok = k0 + ls * 8 < a.cinv, kw_ok = k0 + ls * 8 < a.k_pad;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#ifdef PW_EXP_NO_X
        const bool ok = false;
#else
        const bool ok = k_ok && pix0 + xrow[c] < a.M;
#endif
        pw_dma_1k(ok ? xb + xo[c] : zero, buf + (c * 8 + wave) * 1024);
      }
#pragma unroll
      for (int c = 0; c < WPC; ++c) {
        const int id = c * 8 + wave;
        if (id < WCH) {      // wave-uniform
#ifdef PW_EXP_NO_W
          const bool ok = false;
#else
          const bool ok = kw_ok && nb * a.nb_rows + id * 8 + r < a.n_pad;
#endif
          pw_dma_1k(ok ? wb + wo[c] : zero, buf + PW_XBYTES + id * 1024);
        }
      }
      return;
    }
    const int rr = pw_opaque(r);
    const int tap = a.ntaps == 4 ? a.sl_tap[sl] : a.ntaps > 1 ? sl / a.spt : 0;
    const int k = (a.ntaps == 4 ? a.sl_chunk[sl] : sl - tap * a.spt) * 64 + ls * 8;
    // 4 taps: offsets 0 / +1, image taps (1..2, 1..2);  9 taps: offsets -1 .. +1, image tap = tap
    const int t3 = (tap * 11) >> 5;      // tap / 3 for tap < 9
    const int tdy = a.ntaps == 9 ? t3 - 1 : tap >> 1, tdx = a.ntaps == 9 ? tap - 3 * t3 - 1 : tap & 1;
    const int img = a.ntaps == 9 ? tap : a.ntaps == 4 ? (1 + tdy) * 3 + 1 + tdx : 0;
    const char* Wt = Wp + img * a.tap_stride * 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int id = c * 8 + wave;
      const long pix = (long)pt * PW_BM + id * 8 + rr;
#ifdef PW_EXP_NO_X
      bool ok = false;
#else
      bool ok = pix < a.M && k < a.cinv;
#endif
      long src = pix;
      if (a.ntaps > 1) {      // (M < 2^31: checked by the host)
        const unsigned up = (unsigned)pix, row = up / (unsigned)a.W;
        const int gj = (int)(up - row * (unsigned)a.W), gi = (int)(row % (unsigned)a.H);
        ok = ok && (unsigned)(gi + tdy) < (unsigned)a.H && (unsigned)(gj + tdx) < (unsigned)a.W;
        src = pix + tdy * a.W + tdx;
      }
      pw_dma_1k(ok ? X + (src * a.ldx + k) * 2 : zero, buf + id * 1024);
    }
#pragma unroll
    for (int c = 0; c < (WCH + 7) / 8; ++c) {
      const int id = c * 8 + wave;
      if (id < WCH) {      // wave-uniform
        const int row = nb * a.nb_rows + id * 8 + rr;
#ifdef PW_EXP_NO_W
        const bool ok = false;
#else
        const bool ok = row < a.n_pad && k < a.k_pad;
#endif
        pw_dma_1k(ok ? Wt + ((long)row * a.k_pad + k) * 2 : zero, buf + PW_XBYTES + id * 1024);
      }
    }
  };

  // ---- fragment addresses: pixel tile i of this wave = rows wm*64 + i*16 + li; weight tile j = rows (wn*CTN + j)*16 + li; K-chunk kc flips bit 6
  const int li = lane & 15, q = lane >> 4;
  const unsigned s0 = (unsigned)((q ^ (li & 7)) << 4);
  unsigned xoff = lds_base + (wm * 64 + li) * DD_LDS_ROW + s0;
  unsigned woff = lds_base + PW_XBYTES + (wn * CTN * 16 + li) * DD_LDS_ROW + s0;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const T* __restrict__ Mk = reinterpret_cast<const T*>(a.mask);

  f32x4_t acc[CTN][4];
  int it = j0, sl = 0, sel = 0;
  dma(it, 0, lds_base);
  for (int u = 0; u < nunits; ++u) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's chunks of the current unit have landed
    __syncthreads();
    const bool last = sl == NS - 1;
    const int nsl = last ? 0 : sl + 1, nit = last ? it + per : it;
#ifndef PW_EXP_NO_DMA
    if (u + 1 < nunits) dma(nit, nsl, lds_base + (sel ^ 1) * PW_BUF);
#endif
    const int pt = (it / a.nblk) * xcd_n + xcd, nb = it - (it / a.nblk) * a.nblk;
    const int ch_w = nb * a.nb_rows + wn * CTN * 16 + q * 4;      // first of this lane's 4 channels in weight tile 0
    if (sl == 0) {
#pragma unroll
      for (int j = 0; j < CTN; ++j) {
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (a.bias && ch_w + j * 16 + e < a.nbias) ? a.bias[ch_w + j * 16 + e] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{bv[0], bv[1], bv[2], bv[3]};
      }
    }
    // two K-chunks of 32: 4 pixel fragments + CTN weight fragments each; the weight fragments run three ahead of the MFMAs (the LDS port is
    // ~75 % busy in this loop: a read issued one step ahead is not back in time)
    uint4 xf[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xf[0][i] = pw_lds16(xoff + i * 16 * DD_LDS_ROW); if (IN_RELU) xf[0][i] = relu16<T>(xf[0][i]); }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const unsigned kx = kc ? 64u : 0u;
      constexpr int RING = 4, AHEAD = RING - 1;
      uint4 wf[RING];
#pragma unroll
      for (int j = 0; j < AHEAD && j < CTN; ++j) wf[j] = pw_lds16((woff ^ kx) + j * 16 * DD_LDS_ROW);
#pragma unroll
      for (int j = 0; j < CTN; ++j) {
        if (j + AHEAD < CTN) wf[(j + AHEAD) % RING] = pw_lds16((woff ^ kx) + (j + AHEAD) * 16 * DD_LDS_ROW);
        if (kc == 0) {      // the pixel fragments of the second chunk, spread over the first one's steps
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i * CTN / 4 == j) { xf[1][i] = pw_lds16((xoff ^ 64u) + i * 16 * DD_LDS_ROW); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mma16<T>(wf[j % RING], xf[kc][i], acc[j][i]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kc == 0 && IN_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[1][i] = relu16<T>(xf[1][i]);
      }
    }
#ifdef PW_EXP_NO_EPI
    if (last && acc[0][0][0] == 12345.f) {
#else
    if (last) {
#endif
      // Lane (li, q) of accumulator (j, i) holds channels j*16 + q*4 .. +3 of pixel li of pixel tile i: stored as it is, a wave instruction
      // would write 32-byte pieces.  Four channel tiles at a time are transposed across the four q rows of the wave (two rounds of
      // v_permlane32_swap / v_permlane16_swap, 16 instructions): afterwards row q holds all 16 channels of tile 4*jq + q, i.e. 32 contiguous
      // bytes per lane and 128 contiguous bytes per pixel across the four rows -- whole lines, 16 bytes per access, still fp32 (one rounding).
      const long pixb = (long)pt * PW_BM + wm * 64 + li;
      constexpr int NQ = (CTN + 3) / 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long pix = pixb + i * 16;
        const bool pix_ok = pix < a.M;
        T* yrow = Y + pix * a.ldy;
        const T* mrow = Mk + pix * a.ldmask;
        uint4 mv[NQ][2], ov[NQ][2];
        bool ok8[NQ][2], ok4[NQ][2];
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int tile = jq * 4 + q, ch = nb * a.nb_rows + (wn * CTN + tile) * 16 + h * 8;
            const bool tile_ok = pix_ok && tile < CTN;
            ok8[jq][h] = tile_ok && ch + 8 <= a.n;
            ok4[jq][h] = tile_ok && !ok8[jq][h] && ch + 4 <= a.n;
            mv[jq][h] = uint4{0u, 0u, 0u, 0u}; ov[jq][h] = uint4{0u, 0u, 0u, 0u};
            if (Mk) {
              if (ok8[jq][h]) mv[jq][h] = *reinterpret_cast<const uint4*>(mrow + ch);
              else if (ok4[jq][h]) { const uint2 t = *reinterpret_cast<const uint2*>(mrow + ch); mv[jq][h].x = t.x; mv[jq][h].y = t.y; }
            }
            if (a.accum) {
              if (ok8[jq][h]) ov[jq][h] = *reinterpret_cast<const uint4*>(yrow + ch);
              else if (ok4[jq][h]) { const uint2 t = *reinterpret_cast<const uint2*>(yrow + ch); ov[jq][h].x = t.x; ov[jq][h].y = t.y; }
            }
          }
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq) {
          f32x4_t rr[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) rr[t] = jq * 4 + t < CTN ? acc[jq * 4 + t < CTN ? jq * 4 + t : 0][i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
          pw_swap32(rr[0], rr[2]); pw_swap32(rr[1], rr[3]);
          pw_swap16(rr[0], rr[1]); pw_swap16(rr[2], rr[3]);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int tile = jq * 4 + q, ch = nb * a.nb_rows + (wn * CTN + tile) * 16 + h * 8;
            float v[8] = {rr[2 * h][0], rr[2 * h][1], rr[2 * h][2], rr[2 * h][3], rr[2 * h + 1][0], rr[2 * h + 1][1], rr[2 * h + 1][2], rr[2 * h + 1][3]};
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (Mk) {
              float m8[8];
              unpack8t<T>(mv[jq][h], m8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = m8[e] > 0.f ? v[e] : 0.f;
            }
            if (a.accum) {
              float o8[8];
              unpack8t<T>(ov[jq][h], o8);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += o8[e];
            }
            const uint4 o = pack8t<T>(v);
            if (ok8[jq][h]) *reinterpret_cast<uint4*>(yrow + ch) = o;
            else if (ok4[jq][h]) *reinterpret_cast<uint2*>(yrow + ch) = uint2{o.x, o.y};
          }
        }
      }
    }
    const unsigned flip = sel ? (unsigned)-PW_BUF : (unsigned)PW_BUF;
    xoff += flip; woff += flip;
    sel ^= 1;
    it = nit; sl = nsl;
  }
}

int pw_cus() {
  return dd_device_cus();
}

bool pw_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DD_CONV_PW"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}
int pw_min_k() { return 0; }
int pw_gather9() { return 1; }      // wide gather-form data gradients take the nine-tap GEMM form
int pwg_mid() {      // DD_WGRAD_PW_MID=1: also take the mid-sized gradients (129 ... 256 channels on the wider side)
  static int v = -1;
  if (v < 0) { const char* e = getenv("DD_WGRAD_PW_MID"); v = e ? atoi(e) : 0; }
  return v;
}
long pw_min_pixels() { return 32768; }

template <typename T, int CTN>
void pw_launch_ctn(const PwP& p, bool in_relu, unsigned grid, hipStream_t s) {
  if (in_relu) {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_pw_kernel<T, CTN, true>));
    hipLaunchKernelGGL((conv_pw_kernel<T, CTN, true>), dim3(grid), dim3(512), 2 * (size_t)PW_BUF, s, p);
  } else {
    dd_allow_max_lds(reinterpret_cast<const void*>(conv_pw_kernel<T, CTN, false>));
    hipLaunchKernelGGL((conv_pw_kernel<T, CTN, false>), dim3(grid), dim3(512), 2 * (size_t)PW_BUF, s, p);
  }
}

template <typename T>
void pw_launch(const PwP& p, int ctn, bool in_relu, unsigned grid, hipStream_t s) {
  switch (ctn) {
    case 3: return pw_launch_ctn<T, 3>(p, in_relu, grid, s);
    case 4: return pw_launch_ctn<T, 4>(p, in_relu, grid, s);
    case 5: return pw_launch_ctn<T, 5>(p, in_relu, grid, s);
    case 6: return pw_launch_ctn<T, 6>(p, in_relu, grid, s);
    case 7: return pw_launch_ctn<T, 7>(p, in_relu, grid, s);
    default: return pw_launch_ctn<T, 8>(p, in_relu, grid, s);
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 layer:  out[m][n] += sum_p P[p][m] * Q[p][n]  (P = layer input, Q = output gradient; fp32 atomics), optionally
// bias_out[n] += sum_p Q[p][n].  The same tile economics as the forward: csrc/dd_conv_wgrad.hip gives a workgroup a 64 x 64 corner of the
// gradient, so a 704 x 704 layer reads both operands 11 times (1 016 us for 2 x 184 MB).  Here a workgroup owns up to 320 x 256 of it
// (8 waves = 4 along m x 2 along n, CTM x CTN accumulator tiles per wave) and a contiguous range of 64-pixel units; both operands of a unit
// arrive by LDS-DMA in their stored pixel-major layout and become MFMA operands through transposing reads (ds_read_b64_tr_b16): the reduction
// index of this GEMM is the pixel.  The tile pairs that share a pixel range are workgroups of one XCD.
typedef __attribute__((address_space(3))) s16x4_t* pw_lds_s16x4_ptr;

struct PwgP {
  const void* p; const void* q; float* out; float* bias_out;
  int ldp, ldq, m, n, mv, nv, bias_mode, in_relu;
  long M;
  int nunits, mblk, nblk, spx;      // 64-pixel units; tile grid of the gradient; pixel splits per XCD
  int g9_cp, g9_cout, g9_H, g9_W;   // G9 (transposed-conv filter gradient): channel pitch of a parity plane, real output channels, input grid
};

// LDS image of the weight-gradient kernel: pixel rows of 128 B whose 16-byte slots are XOR-swizzled with key(p) = 2 * (p & 3).  A transposing
// read takes, per 16 lanes, 32 bytes (two adjacent slots) of each of 4 consecutive pixels: with this key the four pixels use the four different
// slot pairs, i.e. all 32 banks once.  (The row-fragment key p & 7 of the other kernels maps pixels 2k and 2k + 1 to the same pair: 2-way conflicts.)
__device__ __forceinline__ uint4 pw_tr_pair(unsigned lo_addr) {
  const unsigned hi_addr = lo_addr + 4 * DD_LDS_ROW;      // four pixels on: same key
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4_ptr)(uintptr_t)lo_addr);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pw_lds_s16x4_ptr)(uintptr_t)hi_addr);
  uint4 v;
  v.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  v.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  v.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  v.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return v;
}

// G9: the filter gradient of the 3x3 / stride-2 transposed conv (dd_convt3_wgrad).  P is the space-to-depth output gradient; row m of the GEMM is
// (tap (a, b) = m / cp, output channel co = m % cp) and reads s at pixel offset ((a == 2), (b == 2)), channels plane(a, b)*cp + co -- nine shifted
// channel windows of one tensor, gathered by the DMA address of each 8-channel group; the result goes to dK[a][b][co][ci].
template <typename T, int CTM, int CTN, bool G9 = false>
__global__ __launch_bounds__(512) void wgrad_pw_kernel(const PwgP a) {
  static_assert(sizeof(T) == 2, "1x1 weight gradient GEMM: bf16 / fp16 storage");
  static_assert(CTM * CTN <= 32, "at most 128 accumulator registers per wave");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int CSM = CTM, CSN = (CTN + 1) / 2;                     // 64-channel slices of the m / n block (4*CTM*16 and 2*CTN*16 channels)
  constexpr int CS_BYTES = 64 * DD_LDS_ROW;                         // one slice of one unit: 64 pixels x 128 B
  constexpr int P_BYTES = CSM * CS_BYTES, BUF = (CSM + CSN) * CS_BYTES;
  constexpr int MB = 4 * CTM * 16, NB = 2 * CTN * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // workgroup -> (tile pair, pixel split): XCD x = blockIdx % 8 holds splits x, x + 8, ...; inside an XCD the tile pairs of one split are neighbours
  const int ntp = a.mblk * a.nblk;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tp = j % ntp, sp = (j / ntp) * 8 + xcd, nsplit = a.spx * 8;
  const int mb = tp / a.nblk, nb = tp - mb * a.nblk;
  const int u0 = (int)((long)a.nunits * sp / nsplit), u1 = (int)((long)a.nunits * (sp + 1) / nsplit);

  const int r = lane >> 3, ls = (lane & 7) ^ ((r & 3) << 1);
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  const char* P = reinterpret_cast<const char*>(a.p);
  const char* Q = reinterpret_cast<const char*>(a.q);
  // G9: per 64-row slice, this lane's 8 rows belong to one tap: its pixel offset (rows, columns), and the element offset of its source window
  int g_dy[CSM], g_dx[CSM], g_off[CSM];
  if constexpr (G9) {
#pragma unroll
    for (int cs = 0; cs < CSM; ++cs) {
      const int ch = mb * MB + cs * 64 + ls * 8, tap = ch / a.g9_cp, co = ch - tap * a.g9_cp;
      const int ta = tap / 3, tb = tap - ta * 3;
      g_dy[cs] = tap < 9 ? (ta == 2) : a.g9_H;      // rows past the ninth tap: always out of range
      g_dx[cs] = tb == 2;
      g_off[cs] = ((ta == 2) * a.g9_W + (tb == 2)) * a.ldp + ((ta & 1) * 2 + (tb & 1)) * a.g9_cp + co;
    }
  }
  auto dma = [&](int u, unsigned buf) {      // unit u: pixels u*64 + wave*8 + r, every 64-channel slice of both blocks
    const long pix = (long)u * 64 + wave * 8 + pw_opaque(r);
    const bool pix_ok = pix < a.M;
    int gi = 0, gj = 0;
    if constexpr (G9) {
      const unsigned up = (unsigned)pix, row = up / (unsigned)a.g9_W;      // (M < 2^31: checked by the host)
      gj = (int)(up - row * (unsigned)a.g9_W); gi = (int)(row % (unsigned)a.g9_H);
    }
#pragma unroll
    for (int cs = 0; cs < CSM; ++cs) {
      if constexpr (G9) {
        const bool ok = pix_ok && gi + g_dy[cs] < a.g9_H && gj + g_dx[cs] < a.g9_W;
        pw_dma_1k(ok ? P + (pix * a.ldp + g_off[cs]) * 2 : zero, buf + cs * CS_BYTES + wave * 1024);
      } else {
        const int ch = mb * MB + cs * 64 + ls * 8;
        pw_dma_1k((pix_ok && ch < a.mv) ? P + (pix * a.ldp + ch) * 2 : zero, buf + cs * CS_BYTES + wave * 1024);
      }
    }
#pragma unroll
    for (int cs = 0; cs < CSN; ++cs) {
      const int ch = nb * NB + cs * 64 + ls * 8;
      pw_dma_1k((pix_ok && ch < a.nv) ? Q + (pix * a.ldq + ch) * 2 : zero, buf + P_BYTES + cs * CS_BYTES + wave * 1024);
    }
  };
  // transposing fragment reads: lane (t16, gq) supplies the address of 4 channels (sub*4 ...) of pixel gq*8 + (t16 >> 2) of the K-step; the
  // instruction hands lane (channel t16, K-group gq) its 4 pixels; a second read 4 pixels on completes the 8
  const int t16 = lane & 15, gq = lane >> 4, sub = t16 & 3, pl = gq * 8 + (t16 >> 2);
  unsigned pa[CTM], qa[CTN];
#pragma unroll
  for (int i = 0; i < CTM; ++i) {
    const int gt = wm * CTM + i, cs = gt >> 2, ct = gt & 3;
    pa[i] = lds_base + cs * CS_BYTES + pl * DD_LDS_ROW + (((ct * 2 + (sub >> 1)) ^ ((pl & 3) << 1)) << 4) + (sub & 1) * 8;
  }
#pragma unroll
  for (int jn = 0; jn < CTN; ++jn) {
    const int gt = wn * CTN + jn, cs = gt >> 2, ct = gt & 3;
    qa[jn] = lds_base + P_BYTES + cs * CS_BYTES + pl * DD_LDS_ROW + (((ct * 2 + (sub >> 1)) ^ ((pl & 3) << 1)) << 4) + (sub & 1) * 8;
  }
  const bool bias_wave = a.bias_mode == 1 && mb == 0 && wm == 0;
  float bsum[CTN];
#pragma unroll
  for (int jn = 0; jn < CTN; ++jn) bsum[jn] = 0.f;
  f32x4_t acc[CTM][CTN];
#pragma unroll
  for (int i = 0; i < CTM; ++i)
#pragma unroll
    for (int jn = 0; jn < CTN; ++jn) acc[i][jn] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (u0 < u1) dma(u0, lds_base);
  unsigned boff = 0;
  for (int u = u0; u < u1; ++u) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#ifndef PWG_EXP_NO_DMA
    if (u + 1 < u1) dma(u + 1, lds_base + (boff ^ (unsigned)BUF));
#endif
    uint4 af[2][CTM];
#pragma unroll
    for (int i = 0; i < CTM; ++i) { af[0][i] = pw_tr_pair(pa[i] + boff); if (a.in_relu) af[0][i] = relu16<T>(af[0][i]); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned koff = boff + ks * 32 * DD_LDS_ROW;
      uint4 bf[3];
      bf[0] = pw_tr_pair(qa[0] + koff);
      if (CTN > 1) bf[1] = pw_tr_pair(qa[1] + koff);
#pragma unroll
      for (int jn = 0; jn < CTN; ++jn) {
        if (jn + 2 < CTN) bf[(jn + 2) % 3] = pw_tr_pair(qa[jn + 2] + koff);
        if (ks == 0) {      // the P fragments of the second K-step, spread over the first one's channel tiles
#pragma unroll
          for (int i = 0; i < CTM; ++i)
            if (i * CTN / CTM == jn) { af[1][i] = pw_tr_pair(pa[i] + boff + 32 * DD_LDS_ROW); if (a.in_relu) af[1][i] = relu16<T>(af[1][i]); }
        }
        if (bias_wave) {
          float f[8];
          unpack8t<T>(bf[jn % 3], f);
          bsum[jn] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CTM; ++i) acc[i][jn] = mma16<T>(af[ks][i], bf[jn % 3], acc[i][jn]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    boff ^= (unsigned)BUF;
  }
  // accumulator (i, jn), lane (li = lane & 15, q = lane >> 4): out[m = mb*MB + (wm*CTM + i)*16 + q*4 + e][n = nb*NB + (wn*CTN + jn)*16 + li]
  const int li = lane & 15, q4 = (lane >> 4) * 4;
  dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
  if (u0 < u1) {
#pragma unroll
    for (int i = 0; i < CTM; ++i)
#pragma unroll
      for (int jn = 0; jn < CTN; ++jn) {
        const int n = nb * NB + (wn * CTN + jn) * 16 + li;
        if (n >= a.n) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mb * MB + (wm * CTM + i) * 16 + q4 + e;
          if constexpr (G9) {
            const int tap = m / a.g9_cp, co = m - tap * a.g9_cp;
            if (tap < 9 && co < a.g9_cout) atomicAdd(a.out + ((long)tap * a.g9_cout + co) * a.n + n, acc[i][jn][e]);
            continue;
          }
#ifdef PWG_EXP_NO_ATOMIC
          if (m < a.m && acc[i][jn][e] == 12345.f) a.out[(long)m * a.n + n] = 1.f;
#else
          if (m < a.m) atomicAdd(a.out + (long)m * a.n + n, acc[i][jn][e]);
#endif
        }
      }
    if (a.bias_mode == 1) {
#pragma unroll
      for (int jn = 0; jn < CTN; ++jn) {
        float b = bsum[jn];
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        const int c = nb * NB + (wn * CTN + jn) * 16 + li;
        if (bias_wave && lane < 16 && c < a.n) atomicAdd(a.bias_out + c, b);
      }
    }
  }
  dd_det_end();
}

struct PwgCfg { int ctm, ctn; };
constexpr PwgCfg PWG_CFGS[] = {{4, 8}, {5, 5}, {3, 6}, {2, 3}, {5, 1}, {5, 2}, {2, 8}};

template <typename T, int CTM, int CTN>
void wgrad_pw_launch_cfg(const PwgP& p, unsigned grid, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)(CTM + (CTN + 1) / 2) * 64 * DD_LDS_ROW;
  dd_det_sync();
  if (p.g9_cp) {
    dd_allow_max_lds(reinterpret_cast<const void*>(wgrad_pw_kernel<T, CTM, CTN, true>));
    hipLaunchKernelGGL((wgrad_pw_kernel<T, CTM, CTN, true>), dim3(grid), dim3(512), lds, s, p);
    return;
  }
  dd_allow_max_lds(reinterpret_cast<const void*>(wgrad_pw_kernel<T, CTM, CTN, false>));
  hipLaunchKernelGGL((wgrad_pw_kernel<T, CTM, CTN, false>), dim3(grid), dim3(512), lds, s, p);
}

template <typename T>
void wgrad_pw_launch(const PwgP& p, int cfg, unsigned grid, hipStream_t s) {
  switch (cfg) {
    case 0: return wgrad_pw_launch_cfg<T, 4, 8>(p, grid, s);
    case 1: return wgrad_pw_launch_cfg<T, 5, 5>(p, grid, s);
    case 2: return wgrad_pw_launch_cfg<T, 3, 6>(p, grid, s);
    case 3: return wgrad_pw_launch_cfg<T, 2, 3>(p, grid, s);
    case 4: return wgrad_pw_launch_cfg<T, 5, 1>(p, grid, s);
    case 5: return wgrad_pw_launch_cfg<T, 5, 2>(p, grid, s);
    default: return wgrad_pw_launch_cfg<T, 2, 8>(p, grid, s);
  }
}

}  // namespace

// 1x1 layers with more than 64 output channels on a grid large enough to fill the chip, bf16 / f16 storage, no residual operand.
bool dd_conv_pw_eligible(const dd_conv_args* a) {
  if (!pw_enabled() || a->taps != 1 || (a->dtype != DD_BF16 && a->dtype != DD_F16) || a->res) return false;
  if (a->flags & ~(DD_IN_RELU | DD_OUT_RELU | DD_ACCUM)) return false;
  if (a->n_pad <= 64 || (long)a->B * a->H * a->W < pw_min_pixels() || a->cin < pw_min_k()) return false;
  return a->cin % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && a->n % 4 == 0 && a->k_pad % 32 == 0 && (!a->mask || (a->ldmask % 8 == 0 && ((uintptr_t)a->mask % 16) == 0)) &&
         ((uintptr_t)a->y % 16) == 0;
}

static long g_pw_launches = 0;
extern "C" long dd_conv_pw_count(void) { return g_pw_launches; }

static int pw_plan_and_launch(PwP& p, int dtype, bool in_relu, hipStream_t s) {
  p.mtiles = (int)((p.M + PW_BM - 1) / PW_BM);
  // channel blocks of at most 256; all blocks equally wide (two waves x CTN tiles of 16), CTN >= 3
  const int nt = (p.n + 15) / 16;
  p.nblk = (nt + 15) / 16;
  int ctn = ((nt + p.nblk - 1) / p.nblk + 1) / 2;
  if (ctn < 3) ctn = 3;
  p.nb_rows = 2 * ctn * 16;
  const long items = (long)p.mtiles * p.nblk;
  long grid = pw_cus();
  grid = grid / 8 * 8;
  if (grid < 8) grid = 8;
  if (grid > (items + 7) / 8 * 8) grid = (items + 7) / 8 * 8;
  if (dtype == DD_BF16) pw_launch<bf16_t>(p, ctn, in_relu, (unsigned)grid, s);
  else pw_launch<f16_t>(p, ctn, in_relu, (unsigned)grid, s);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

int dd_conv_pw_launch(const dd_conv_args* a, hipStream_t s) {
  ++g_pw_launches;
  PwP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.mask = a->mask; p.y = a->y;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldmask = a->ldmask; p.cinv = a->cin; p.n = a->n; p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.nbias = a->bias ? a->nbias : 0; p.nslices = (a->cin + 63) / 64;
  p.M = (long)a->B * a->H * a->W;
  p.relu = (a->flags & DD_OUT_RELU) != 0; p.accum = (a->flags & DD_ACCUM) != 0;
  p.ntaps = 1; p.spt = p.nslices; p.H = a->H; p.W = a->W; p.tap_stride = 0;
  return pw_plan_and_launch(p, a->dtype, (a->flags & DD_IN_RELU) != 0, s);
}

// dd_conv3x3_ks mode 6 with many output channels: the K-streamed kernel covers 64 of them per pass over the (small) input; here 256.
bool dd_conv_pw_taps_eligible(const dd_conv_ks_args* a) {
  // mode 6 (any epilogue it takes), or mode 0 with the gradient epilogue (the gather-form data gradient of a wide dense-block prefix)
  const bool grad_epi = (a->flags & DD_ACCUM) != 0 || a->mask != nullptr;
  if (!pw_enabled() || !(a->mode == 6 || (a->mode == 0 && grad_epi && pw_gather9())) || a->n <= 128 || a->n0 != 0) return false;
  if (a->flags & (DD_IN_RELU | DD_OUT_RELU)) return false;
  if ((long)a->B * a->H * a->W >= (1L << 31) || (long)a->B * a->H * a->W < 2048) return false;
  if (a->mode == 6 && (a->cin % 4 != 0 || 4 * ((a->cin + 63) / 64) > 40)) return false;
  return a->cin % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && ((uintptr_t)a->y % 16) == 0 && (!a->mask || (a->ldmask % 8 == 0 && ((uintptr_t)a->mask % 16) == 0));
}

int dd_conv_pw_taps_launch(const dd_conv_ks_args* a, hipStream_t s) {
  ++g_pw_launches;
  PwP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.mask = a->mask; p.y = a->y;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldmask = a->ldmask; p.cinv = a->cin; p.n = a->n; p.n_pad = a->n_pad; p.k_pad = a->k_pad;
  p.nbias = a->bias ? a->nbias : 0;
  p.M = (long)a->B * a->H * a->W;
  p.relu = (a->flags & DD_OUT_RELU) != 0; p.accum = (a->flags & DD_ACCUM) != 0;
  p.ntaps = a->mode == 6 ? 4 : 9; p.spt = (a->cin + 63) / 64; p.nslices = p.ntaps * p.spt; p.H = a->H; p.W = a->W; p.tap_stride = (long)a->n_pad * a->k_pad;
  if (p.ntaps == 4) {      // cin = 4 cp channels, plane pl = channels [pl*cp, (pl+1)*cp); tap (dy, dx) meets plane (py, px) iff py <= 1 - dy and px <= 1 - dx
    const int cp = a->cin / 4;
    int ns = 0;
    for (int t = 0; t < 4; ++t)
      for (int c = 0; c < p.spt; ++c) {
        bool live = false;
        for (int pl = 0; pl < 4; ++pl) {
          const bool meets = (pl >> 1) <= 1 - (t >> 1) && (pl & 1) <= 1 - (t & 1);
          if (meets && pl * cp < (c + 1) * 64 && (pl + 1) * cp > c * 64) live = true;
        }
        if (live) { p.sl_tap[ns] = (unsigned char)t; p.sl_chunk[ns] = (unsigned char)c; ++ns; }
      }
    p.nslices = ns;
  }
  return pw_plan_and_launch(p, a->dtype, false, s);
}

// 1x1 weight gradients with more than 64 channels on either side (conv2d form: bias gradient from Q or none), bf16 / f16 storage.
bool dd_wgrad_pw_eligible(const dd_wgrad_args* a) {
  if (!pw_enabled() || a->taps != 1 || (a->dtype != DD_BF16 && a->dtype != DD_F16) || a->bias_mode == 2) return false;
  if (a->flags & ~DD_IN_RELU) return false;
  if ((a->m <= 64 && a->n <= 64) || (long)a->B * a->H * a->W < pw_min_pixels()) return false;
  // 129 ... 256 channels on the wider side with more than 32 on the other: one or two gradient tiles, i.e. every workgroup adds a whole
  // 192 x 192 tile atomically (9.4 M atomics for 176 x 176: 76 us against 43 us on the 64 x 64-slice kernel of csrc/dd_conv_wgrad.hip)
  const int hi = a->m > a->n ? a->m : a->n, lo = a->m > a->n ? a->n : a->m;
  if (hi > 128 && hi <= 256 && lo > 32 && !pwg_mid()) return false;
  return true;
}

static long g_pwg_launches = 0;
extern "C" long dd_wgrad_pw_count(void) { return g_pwg_launches; }

// tile shape and pixel splits for an m x n gradient over p.nunits units; launches
static int pwg_plan_and_launch(PwgP& p, int dtype, hipStream_t s) {
  // tile shape: the candidate with the least (operand re-reads + padded MFMA work), both in seconds per pixel at 4 TB/s and 1.2 PFLOP/s
  const int nmt = (p.m + 15) / 16, nnt = (p.n + 15) / 16;
  int best = 0;
  double best_cost = 1e30;
  for (int c = 0; c < (int)(sizeof(PWG_CFGS) / sizeof(PWG_CFGS[0])); ++c) {
    const int mt = 4 * PWG_CFGS[c].ctm, nt = 2 * PWG_CFGS[c].ctn;
    const int mblk = (nmt + mt - 1) / mt, nblk = (nnt + nt - 1) / nt;
    const double bytes = 2.0 * ((double)nblk * p.m + (double)mblk * p.n), flops = 2.0 * (mblk * mt * 16.0) * (nblk * nt * 16.0);
    const double cost = bytes / 4e12 + flops / 1.2e15;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  const int mt = 4 * PWG_CFGS[best].ctm, nt = 2 * PWG_CFGS[best].ctn;
  p.mblk = (nmt + mt - 1) / mt; p.nblk = (nnt + nt - 1) / nt;
  const int ntp = p.mblk * p.nblk;
  int spx = pw_cus() / 8 / ntp;
  if (spx < 1) spx = 1;
  while (spx > 1 && (long)spx * 8 > p.nunits) --spx;
  p.spx = spx;
  const unsigned grid = (unsigned)(8 * spx * ntp);
  if (dtype == DD_BF16) wgrad_pw_launch<bf16_t>(p, best, grid, s);
  else wgrad_pw_launch<f16_t>(p, best, grid, s);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

int dd_wgrad_pw_launch(const dd_wgrad_args* a, hipStream_t s) {
  ++g_pwg_launches;
  PwgP p;
  p.p = a->p; p.q = a->q; p.out = a->out; p.bias_out = a->bias_out; p.bias_mode = a->bias_mode;
  p.ldp = a->ldp; p.ldq = a->ldq; p.m = a->m; p.n = a->n; p.mv = (a->m + 7) / 8 * 8; p.nv = (a->n + 7) / 8 * 8;
  p.in_relu = (a->flags & DD_IN_RELU) != 0;
  p.M = (long)a->B * a->H * a->W;
  p.nunits = (int)((p.M + 63) / 64);
  p.g9_cp = p.g9_cout = p.g9_H = p.g9_W = 0;
  return pwg_plan_and_launch(p, a->dtype, s);
}

extern "C" int dd_convt3_wgrad(const dd_convt3_wgrad_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->s && a->x && a->dk, "dd_convt3_wgrad: null pointer");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_convt3_wgrad: storage dtype must be DD_BF16 or DD_F16");
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->cin > 0 && a->cout > 0, "dd_convt3_wgrad: empty problem");
  DD_REQUIRE(a->cp % 16 == 0 && a->cp >= a->cout && a->lds % 8 == 0 && a->lds >= 4 * a->cp && a->ldx % 8 == 0 && a->ldx >= (a->cin + 7) / 8 * 8,
             "dd_convt3_wgrad: cp=%d (%%16, >= cout=%d) lds=%d (%%8, >= 4 cp) ldx=%d (%%8, covers cin=%d rounded to 8)", a->cp, a->cout, a->lds, a->ldx, a->cin);
  DD_REQUIRE(((uintptr_t)a->s % 16) == 0 && ((uintptr_t)a->x % 16) == 0, "dd_convt3_wgrad: s / x must be 16-byte aligned");
  PwgP p;
  p.p = a->s; p.q = a->x; p.out = a->dk; p.bias_out = nullptr; p.bias_mode = 0;
  p.ldp = a->lds; p.ldq = a->ldx; p.m = 9 * a->cp; p.n = a->cin; p.mv = p.m; p.nv = (a->cin + 7) / 8 * 8;
  p.in_relu = 0;
  p.M = (long)a->B * a->H * a->W;
  p.nunits = (int)((p.M + 63) / 64);
  p.g9_cp = a->cp; p.g9_cout = a->cout; p.g9_H = a->H; p.g9_W = a->W;
  DD_REQUIRE(p.M < (1L << 31), "dd_convt3_wgrad: more than 2^31 pixels");
  return pwg_plan_and_launch(p, a->dtype, reinterpret_cast<hipStream_t>(stream));
}
