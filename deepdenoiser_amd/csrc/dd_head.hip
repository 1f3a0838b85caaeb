// Fused kernel-prediction head on CDNA4: AdjustNumberOfChannels + KernelPredictor of one scale as ONE launch each way.
//
// Reference seams replaced (file:line in /root/reference): TensorFlow/Architecture.py:237-244 (AdjustNumberOfChannels.predict: 1x1 conv
// C -> K + ReLU, 1x1 conv K -> K) and Architecture.py:260-289 -> KernelPrediction.py:11-63 (softmax over the K = k*k logits, symmetric
// pad, per-pixel k x k filter of the 3-channel source), plus their TF-autodiff gradients (Training.py:701-702).
//
//   hid    = relu(Wa^T x + ba)                 K channels (25 for the 5x5 kernel)
//   logits = Wb^T hid + bb                     K channels
//   out_c  = sum_t softmax(logits)_t * src_sympad[p + t][c]
//
// The layer-by-layer path writes and re-reads the K-channel hid / logits tensors (and their gradients) through HBM: 3 launches forward and
// 5 backward per scale for ~1 % of the flops.  Here a wave owns 16 pixels at a time; x is read ONCE, straight into MFMA B fragments
// (NHWC: a lane's 8 channels of one pixel are 16 contiguous bytes), and everything else stays in registers:
//   * a lane's hid values of the first GEMM (4 consecutive channels of each 16-channel tile of "its" pixel) ARE its k-group of the second
//     GEMM -- the reduction index of a GEMM may be permuted freely as long as the weight operand uses the same permutation (`slot_ch`),
//     so no shuffle or LDS round trip separates the two 1x1 layers, nor the two data-gradient GEMMs of the backward;
//   * the softmax and the filter apply run on the four lanes that share a pixel (two xor-shuffles per reduction).
// The backward recomputes hid / logits / softmax from x (two tiny GEMMs) instead of loading them, so the forward stores nothing.  Its weight
// gradients need pixel-major operands: each wave parks 32 pixels of [hid | d logits | d hid | x] in a private LDS strip, reads them back
// with ds_read_b64_tr_b16 and accumulates the gradient tiles in registers for the whole launch (bias gradients = an all-ones channel).
#include "dd_common.h"

#ifdef DD_PROFILE_PHASES
// cycle stamps of workgroup 0, thread 0 of the backward kernel (tools/head_phases.py): slot i accumulates the cycles between stamps i-1 and i
__device__ unsigned long long dd_hphase[16];
#define HPH_DECL() unsigned long long hph_last = __builtin_readcyclecounter(); const bool hph_on = blockIdx.x == 0 && threadIdx.x == 0
#define HPH(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_readcyclecounter(); if (hph_on) dd_hphase[i] += now_ - hph_last; hph_last = now_; } while (0)
extern "C" int dd_debug_hphases(unsigned long long* out16, int reset) {
  if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(dd_hphase), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dd_hphase), z, sizeof(z)); }
  return 0;
}
#else
#define HPH_DECL()
#define HPH(i)
#endif

namespace {

struct HeadP {
  const void* x; const float* src; const float* wa; const float* ba; const float* wb; const float* bb; float* out;
  const float* dout; void* dx; float* dwa; float* dba; float* dwb; float* dbb;
  int ldx, C, ldsrc, ldo, lddo, lddx, accumulate;
  int N, H, W;
  long npix;
};

template <int KS> struct HeadDim {
  static constexpr int K = KS * KS, NT = (K + 15) / 16, P = (KS - 1) / 2;
  static constexpr int ONES_POS = ((K / 16) << 2) + ((K % 16) >> 2) * 8 + (K & 3);   // slot position of channel K (the first unused one)
};
// hidden / logit channel held in slot e of k-group q: a lane's MFMA result registers in the order they come (tile e >> 2, element e & 3)
__device__ __forceinline__ int slot_ch(int q, int e) { return (e >> 2) * 16 + q * 4 + (e & 3); }
// ... and the inverse for a position pos = q * 8 + e of a 32-slot row
__device__ __forceinline__ int pos_ch(int pos) { return slot_ch(pos >> 3, pos & 7); }

typedef __attribute__((address_space(3))) s16x4_t* lds_tr16_ptr;

// Work unit: a SEGMENT = 16 consecutive pixels of one image row (lane li <-> x = 16 * sx + li).  A wave walks a contiguous range of segments in
// (image, row, x-segment) order, so the position advances with compares and adds on wave-uniform values -- no per-lane division of a linear
// pixel index (the first version spent most of its ~1 700 VALU instructions per 32 pixels on exactly that and on 64-bit address arithmetic).
struct SegWalk {
  int b, y, sx;          // wave-uniform
  long left;             // segments left in this wave's range
};
__device__ __forceinline__ SegWalk seg_start(long first, long count, long nsegs, int nseg_x, int H) {
  SegWalk w;
  if (first > nsegs) first = nsegs;
  w.left = first + count <= nsegs ? count : nsegs - first;
  const long row = first / nseg_x;
  w.sx = (int)(first - row * nseg_x);
  w.b = (int)(row / H);
  w.y = (int)(row - (long)w.b * H);
  return w;
}
__device__ __forceinline__ void seg_next(SegWalk& w, int nseg_x, int H) {
  --w.left;
  if (++w.sx == nseg_x) { w.sx = 0; if (++w.y == H) { w.y = 0; ++w.b; } }
}
struct PixelAddr { int b, y, x; bool ok; };
// this lane's pixel of the walker's current segment (clamped into the image when the segment is ragged or the range is exhausted)
__device__ __forceinline__ PixelAddr seg_pixel(const SegWalk& w, int li, const HeadP& a) {
  PixelAddr r;
  const bool live = w.left > 0;
  r.b = live ? w.b : 0;
  r.y = live ? w.y : 0;
  const int x = w.sx * 16 + li;
  r.ok = live && x < a.W;
  r.x = r.ok ? x : 0;
  return r;
}
__device__ __forceinline__ int sym(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }
// element offsets of the taps this lane's channels stand for (channel t <-> tap (t / KS, t % KS); channels >= K re-read tap 0, their weight is 0):
// (ty << 8) | tx per (tile j, element e), fixed for the lane
template <int KS>
__device__ __forceinline__ void lane_taps(int (&tyx)[HeadDim<KS>::NT][4], int q) {
#pragma unroll
  for (int j = 0; j < HeadDim<KS>::NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t0 = j * 16 + q * 4 + e, t = t0 < HeadDim<KS>::K ? t0 : 0;
      tyx[j][e] = ((t / KS) << 8) | (t % KS);
    }
}
// the three source channels of every tap of this lane, all requested back to back (32-bit offsets from the image's base)
template <int KS>
__device__ __forceinline__ void load_taps(float (&sv)[HeadDim<KS>::NT][4][3], const int (&tyx)[HeadDim<KS>::NT][4], const float* img, const PixelAddr& pa,
                                          const HeadP& a) {
  constexpr int P = HeadDim<KS>::P;
#pragma unroll
  for (int j = 0; j < HeadDim<KS>::NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int sy = sym(pa.y + (tyx[j][e] >> 8) - P, a.H), sx = sym(pa.x + (tyx[j][e] & 255) - P, a.W);
      const float* sp = img + (unsigned)((sy * a.W + sx) * a.ldsrc);
      sv[j][e][0] = sp[0]; sv[j][e][1] = sp[1]; sv[j][e][2] = sp[2];
    }
}

// The weight fragments of the two forward GEMMs, built once per wave from the fp32 master variables (TensorFlow layout [C][K], [K][K]).
template <typename T, int KS, int NCH> struct FwdWeights {
  uint4 a1[HeadDim<KS>::NT][NCH];   // GEMM 1: row n = j*16 + li, k = c*32 + q*8 + e          -> Wa[k][n]
  uint4 a2[HeadDim<KS>::NT];        // GEMM 2: row n = j*16 + li, k-slot (q, e) = channel slot_ch -> Wb[slot_ch][n]
  float b1[HeadDim<KS>::NT][4], b2[HeadDim<KS>::NT][4];      // biases of this lane's result channels j*16 + q*4 + e
};

// The fp32 master weights, staged ONCE per workgroup into LDS with coalesced loads: [Wa (C x K) | Wb (K x K) | ba | bb].  A lane's fragment
// elements are K floats apart (100 bytes for K = 25): gathered straight from global memory every wave-instruction touches ~50 cache lines,
// and with every wave of the launch doing it at once that alone was ~70 us per launch; from LDS the odd stride is conflict-free.
struct HeadW { const float* wa; const float* wb; const float* ba; const float* bb; int C; };
template <int KS>
__device__ __forceinline__ HeadW stage_head_weights(float* lds, const HeadP& a, int nthreads) {
  constexpr int K = HeadDim<KS>::K;
  const int nwa = a.C * K;
  for (int i = threadIdx.x; i < nwa; i += nthreads) lds[i] = a.wa[i];
  for (int i = threadIdx.x; i < K * K; i += nthreads) lds[nwa + i] = a.wb[i];
  for (int i = threadIdx.x; i < K; i += nthreads) { lds[nwa + K * K + i] = a.ba[i]; lds[nwa + K * K + K + i] = a.bb[i]; }
  __syncthreads();
  HeadW h;
  h.wa = lds; h.wb = lds + nwa; h.ba = lds + nwa + K * K; h.bb = h.ba + K; h.C = a.C;
  return h;
}

template <typename T, int KS, int NCH>
__device__ __forceinline__ void load_fwd_weights(FwdWeights<T, KS, NCH>& w, const HeadW& a, int li, int q) {
  constexpr int K = HeadDim<KS>::K, NT = HeadDim<KS>::NT;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 16 + li;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {                      // (unconditional loads from clamped indices, zeroed afterwards: see head_fwd_kernel)
        const int k = c * 32 + q * 8 + e;
        const bool ok = n < K && k < a.C;
        const float t = a.wa[ok ? k * K + n : 0];
        v[e] = ok ? t : 0.f;
      }
      w.a1[j][c] = pack8t<T>(v);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = slot_ch(q, e);
      const bool ok = n < K && (e >> 2) < NT && ch < K;
      const float t = a.wb[ok ? ch * K + n : 0];
      v[e] = ok ? t : 0.f;
    }
    w.a2[j] = pack8t<T>(v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = j * 16 + q * 4 + e;
      const float t1 = a.ba[ch < K ? ch : 0], t2 = a.bb[ch < K ? ch : 0];
      w.b1[j][e] = ch < K ? t1 : 0.f;
      w.b2[j][e] = ch < K ? t2 : 0.f;
    }
  }
}

// x fragments of one pixel (B operand of GEMM 1): 8 channels = 16 bytes per k-group; zero beyond C
template <typename T, int NCH>
__device__ __forceinline__ void load_x(uint4 (&xf)[NCH], const HeadP& a, long pix, int q) {
  const T* px = reinterpret_cast<const T*>(a.x) + pix * a.ldx;
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = c * 32 + q * 8;
    xf[c] = *reinterpret_cast<const uint4*>(ch < a.C ? px + ch : zero);
  }
}

// hid (as the k-group of GEMM 2, in the storage type) and the logits of this lane's channels, both rounded as the layer-wise path stores them
// a1_lds != nullptr: the GEMM-1 weight fragments are read from a fragment-ready LDS image ([j][c][lane] x 16 bytes) instead of registers
// (the backward kernel keeps up to 22 gradient tiles in registers and has none to spare for them)
template <typename T, int KS, int NCH>
__device__ __forceinline__ void forward_gemms(const FwdWeights<T, KS, NCH>& w, const uint4 (&xf)[NCH], uint4& hid, float (&lg)[HeadDim<KS>::NT][4]) {
  constexpr int NT = HeadDim<KS>::NT;
  float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    f32x4_t acc = {w.b1[j][0], w.b1[j][1], w.b1[j][2], w.b1[j][3]};
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc = mma16<T>(w.a1[j][c], xf[c], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) h[j * 4 + e] = fmaxf(acc[e], 0.f);
  }
  hid = pack8t<T>(h);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    f32x4_t acc = {w.b2[j][0], w.b2[j][1], w.b2[j][2], w.b2[j][3]};
    acc = mma16<T>(w.a2[j], hid, acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) lg[j][e] = Elem<T>::to_f32(Elem<T>::from_f32(acc[e]));
  }
}

// The same two GEMMs with every lane-dependent operand read from the workgroup's fragment image in LDS (backward kernel: its registers go to the
// weight-gradient tiles): image = [a1: NT * NCH][a4: 2 NCH][a3: NT][a2: NT][b1: NT][b2: NT] x 64 lanes x 16 bytes, `img` already offset by the lane.
template <typename T, int KS, int NCH>
__device__ __forceinline__ void forward_gemms_lds(const uint4* img, const uint4 (&xf)[NCH], uint4& hid, float (&lg)[HeadDim<KS>::NT][4]) {
  constexpr int NT = HeadDim<KS>::NT, A2 = NT * NCH + 2 * NCH + NT, B1 = A2 + NT, B2 = B1 + NT;
  float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    f32x4_t acc = *reinterpret_cast<const f32x4_t*>(img + (B1 + j) * 64);
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc = mma16<T>(img[(j * NCH + c) * 64], xf[c], acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) h[j * 4 + e] = fmaxf(acc[e], 0.f);
  }
  hid = pack8t<T>(h);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    f32x4_t acc = *reinterpret_cast<const f32x4_t*>(img + (B2 + j) * 64);
    acc = mma16<T>(img[(A2 + j) * 64], hid, acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) lg[j][e] = Elem<T>::to_f32(Elem<T>::from_f32(acc[e]));
  }
}

// softmax over the K logits of a pixel, spread over the four lanes (q) that share it: p[j][e] for channel j*16 + q*4 + e (0 for channels >= K)
template <int KS>
__device__ __forceinline__ void softmax4(float (&lg)[HeadDim<KS>::NT][4], int q) {
  constexpr int K = HeadDim<KS>::K, NT = HeadDim<KS>::NT;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) if (j * 16 + q * 4 + e < K) mx = fmaxf(mx, lg[j][e]);
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lg[j][e] = j * 16 + q * 4 + e < K ? __expf(lg[j][e] - mx) : 0.f;
      sum += lg[j][e];
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) lg[j][e] *= inv;
}

// ------------------------------------------------------------------------------------------------------------------------ forward
template <typename T, int KS, int NCH>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadP a) {
  constexpr int NT = HeadDim<KS>::NT;
  const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
  // (readfirstlane: the wave index is uniform, so the segment walk -- image, row, x-segment, segments left -- lives in scalar registers)
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  __shared__ float s_w[128 * HeadDim<KS>::K + HeadDim<KS>::K * HeadDim<KS>::K + 2 * HeadDim<KS>::K];
  const HeadW hw = stage_head_weights<KS>(s_w, a, 256);
  FwdWeights<T, KS, NCH> w;
  load_fwd_weights<T, KS, NCH>(w, hw, li, q);
  int tyx[NT][4];
  lane_taps<KS>(tyx, q);
  const int nseg_x = (a.W + 15) >> 4;
  const long nsegs = (long)a.N * a.H * nseg_x, per_wave = (nsegs + nwaves - 1) / nwaves;
  SegWalk sw = seg_start(wave * per_wave, per_wave, nsegs, nseg_x, a.H);
  for (; sw.left > 0; seg_next(sw, nseg_x, a.H)) {
    const PixelAddr pa = seg_pixel(sw, li, a);
    const int pix = (pa.b * a.H + pa.y) * a.W + pa.x;                    // (npix < 2^31: checked by the host)
    uint4 xf[NCH];
    load_x<T, NCH>(xf, a, pix, q);
    const float* img = a.src + (long)pa.b * a.H * a.W * a.ldsrc;
    // the taps this lane's channels stand for.  Every load is unconditional (channels >= K re-read tap 0 with weight 0): a branch per tap
    // would put a full memory round trip behind each of them instead of one behind all
    float sv[NT][4][3];
    load_taps<KS>(sv, tyx, img, pa, a);
    uint4 hid;
    float p[NT][4];
    forward_gemms<T, KS, NCH>(w, xf, hid, p);
    softmax4<KS>(p, q);
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0 += p[j][e] * sv[j][e][0]; o1 += p[j][e] * sv[j][e][1]; o2 += p[j][e] * sv[j][e][2]; }
    o0 += __shfl_xor(o0, 16); o0 += __shfl_xor(o0, 32);
    o1 += __shfl_xor(o1, 16); o1 += __shfl_xor(o1, 32);
    o2 += __shfl_xor(o2, 16); o2 += __shfl_xor(o2, 32);
    if (pa.ok && q < 3) a.out[(long)pix * a.ldo + q] = q == 0 ? o0 : (q == 1 ? o1 : o2);
  }
}

// ------------------------------------------------------------------------------------------------------------------------ backward
// transposed MFMA fragment from a wave-private [32 pixels][row bytes] strip: channel tile `ctile` (16 channels = 32 bytes), the 8 pixels of k-group g
__device__ __forceinline__ uint4 strip_frag_tr(const char* strip, int row_bytes, int ctile, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const char* a0 = strip + (g * 8 + (t16 >> 2)) * row_bytes + ctile * 32 + (t16 & 3) * 8;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr16_ptr)(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr16_ptr)(a0 + 4 * row_bytes));
  uint4 r;
  r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}

// Row pitches of a wave's strip.  The 16-byte parking stores of 8 consecutive pixels and the transposing reads of 4 consecutive pixel rows must
// spread over the banks: with the natural pitches (64 B for the three 32-slot strips, 64 / 128 / 192 / 256 B for x) pixels p and p + 4 (64 B) or
// p + 2 (128 B) share their banks -- SQ_LDS_BANK_CONFLICT was 59 % of the LDS cycles of the 64-channel instantiation.  112 B (28 dwords) and
// x + 16 B keep both patterns apart (the bank model is in DESIGN 3.4); the 128-channel instantiation keeps the natural pitches (it has no
// register left for the odd multiples: 9 spilled with them).
template <int NCH> struct HeadStrip {
#ifdef HB_EXP_NATURAL_PITCH      // (experiment builds, tools/build_variant.sh: the pitches of rounds 2-4)
  static constexpr int HROW = 64, XROW = NCH * 64;
#else
  static constexpr int HROW = NCH <= 3 ? 112 : 64;
  static constexpr int XROW = NCH == 1 ? 112 : (NCH <= 3 ? NCH * 64 + 16 : NCH * 64);
#endif
  static constexpr int BYTES = 32 * (3 * HROW + XROW);
};
constexpr int BWD_WAVES = 8;      // one persistent workgroup per CU, 2 waves per SIMD (a wave's 32-pixel step is a chain of memory round trips)
// ACCUM (d x is added to a gradient already stored) is a template flag: as a runtime branch the never-taken side still cost the 2 * NCH * 2
// registers of the old gradient, and the 128-channel instantiation spilled 55 registers (round 2).
// wg / nwgs: this workgroup's index among the workgroups of ITS problem (a launch may run several scales side by side: head_bwd_multi_kernel)
template <typename T, int KS, int NCH, bool ACCUM>
__device__ __forceinline__ void head_bwd_body(const HeadP a, char* smem, int wg_, int nwgs_) {      // (by value: through a reference the 128-channel instantiation spilled 8 registers)
  const int wg = __builtin_amdgcn_readfirstlane(wg_), nwgs = __builtin_amdgcn_readfirstlane(nwgs_);      // (workgroup-uniform: scalar registers)
  constexpr int K = HeadDim<KS>::K, NT = HeadDim<KS>::NT, ONES = HeadDim<KS>::ONES_POS;
  constexpr int CT = NCH * 2;                          // 16-channel tiles of x
  constexpr int XROW = HeadStrip<NCH>::XROW, HROW = HeadStrip<NCH>::HROW, STRIP = HeadStrip<NCH>::BYTES;
  constexpr int NA1 = HeadDim<KS>::NT * NCH;           // fragment-ready operand image behind the strips (forward_gemms_lds): a1, a4, a3, a2, b1, b2
  const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* s_hid = smem + wv * STRIP;                     // [32 px][32 slots] hid (slot order, position ONES := 1)
  char* s_dl = s_hid + 32 * HROW;                      // [32 px][32 slots] d logits
  char* s_dh = s_dl + 32 * HROW;                       // [32 px][32 slots] d hid (pre-activation)
  char* s_x = s_dh + 32 * HROW;                        // [32 px][NCH * 32 channels] x      (rows HROW / XROW bytes apart: HeadStrip)
  const long wave = (long)wg * BWD_WAVES + wv, nwaves = (long)nwgs * BWD_WAVES;

  HPH_DECL();
  const HeadW hw = stage_head_weights<KS>(reinterpret_cast<float*>(smem), a, BWD_WAVES * 64);      // (the strips are not in use yet)
  HPH(0);
  FwdWeights<T, KS, NCH> w;
  load_fwd_weights<T, KS, NCH>(w, hw, li, q);
  HPH(1);
  // data-gradient operands: GEMM 3  d hid[m] = sum_n Wb[m][n] dl[n]  (row m = j*16 + li, k-slot -> n = slot_ch)
  //                         GEMM 4  d x[c]   = sum_m Wa[c][m] dh[m]  (row c = ct*16 + li, k-slot -> m = slot_ch)
  uint4 a3[NT], a4[CT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int m = j * 16 + li;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = slot_ch(q, e);
      const bool ok = m < K && (e >> 2) < NT && n < K;
      const float t = hw.wb[ok ? m * K + n : 0];
      v[e] = ok ? t : 0.f;
    }
    a3[j] = pack8t<T>(v);
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int c = ct * 16 + li;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int mm = slot_ch(q, e);
      const bool ok = c < a.C && (e >> 2) < NT && mm < K;
      const float t = hw.wa[ok ? c * K + mm : 0];
      v[e] = ok ? t : 0.f;
    }
    a4[ct] = pack8t<T>(v);
  }
  uint4* s_a1 = reinterpret_cast<uint4*>(smem + BWD_WAVES * STRIP) + lane;
  uint4* s_a4 = s_a1 + NA1 * 64;
  uint4* s_a3 = s_a4 + CT * 64;
  if (wv == 0) {                                       // the fragments depend on the lane only: one wave publishes them for all
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) s_a1[(j * NCH + c) * 64] = w.a1[j][c];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) s_a4[ct * 64] = a4[ct];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s_a3[j * 64] = a3[j];
      s_a3[(NT + j) * 64] = w.a2[j];
      *reinterpret_cast<f32x4_t*>(s_a3 + (2 * NT + j) * 64) = f32x4_t{w.b1[j][0], w.b1[j][1], w.b1[j][2], w.b1[j][3]};
      *reinterpret_cast<f32x4_t*>(s_a3 + (3 * NT + j) * 64) = f32x4_t{w.b2[j][0], w.b2[j][1], w.b2[j][2], w.b2[j][3]};
    }
  }
  __syncthreads();                                     // every wave has its fragments: the staged weights give way to the strips
  HPH(2);
  // weight-gradient tiles, kept for the whole launch (rows / columns of the K-sized dimensions are slot POSITIONS, mapped back at the end)
  // (a 32-slot row always spans NPT = 2 position tiles: with one channel tile (3x3 kernels) the valid slots e < 4 of all four k-groups
  //  still land in both halves of the row)
  constexpr int NPT = 2;
  f32x4_t g_wb[NPT][NPT], g_ba[NPT], g_wa[CT][NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    g_ba[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NPT; ++j) g_wb[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int j = 0; j < NPT; ++j) g_wa[ct][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // A step = two consecutive segments (32 pixels: one K-step of the weight-gradient MFMAs); each wave walks a contiguous range of them.
  int tyx[NT][4];
  lane_taps<KS>(tyx, q);
  const int nseg_x = (a.W + 15) >> 4;
  const long nsegs = (long)a.N * a.H * nseg_x;
#ifdef HB_EXP_NO_LOOP
  const long per_wave = 0;
#else
  const long per_wave = ((nsegs + nwaves - 1) / nwaves + 1) & ~1L;      // even: a wave's range is whole steps
#endif
  SegWalk sw = seg_start(wave * per_wave, per_wave, nsegs, nseg_x, a.H);
  // x of the NEXT segment is requested one segment ahead (the HBM round trip); everything else a segment reads (d out, source taps, ReLU
  // masks, the gradient it accumulates into) is requested together at its top: one exposed round trip per segment, not five.
  uint4 xnext[NCH];
  {
    const PixelAddr p0 = seg_pixel(sw, li, a);
    load_x<T, NCH>(xnext, a, (p0.b * a.H + p0.y) * a.W + p0.x, q);
  }
  HPH(3);
  while (sw.left > 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const PixelAddr pa = seg_pixel(sw, li, a);
      const int pix = (pa.b * a.H + pa.y) * a.W + pa.x;
      seg_next(sw, nseg_x, a.H);                       // (past the end of the range seg_pixel clamps to pixel 0 and clears `ok`)
      uint4 xf[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) xf[c] = xnext[c];
      {
        const PixelAddr pn = seg_pixel(sw, li, a);
        load_x<T, NCH>(xnext, a, (pn.b * a.H + pn.y) * a.W + pn.x, q);
      }
      // (every load below is unconditional: ragged / exhausted segments were clamped by seg_pixel and are masked by `okf`, channels >= K
      //  re-read tap 0 and carry p = 0 -- see the forward kernel)
      const float okf = pa.ok ? 1.f : 0.f;
      const float* dop = a.dout + (long)pix * a.lddo;
      const float g0 = okf * dop[0], g1 = okf * dop[1], g2 = okf * dop[2];
      const float* img = a.src + (long)pa.b * a.H * a.W * a.ldsrc;
      float sv[NT][4][3];
#ifdef HB_EXP_NO_TAPS
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sv[j][e][0] = (float)tyx[j][e]; sv[j][e][1] = 1.f; sv[j][e][2] = (float)(size_t)img * 0.f; }
#else
      load_taps<KS>(sv, tyx, img, pa, a);
#endif
      T* dxp = reinterpret_cast<T*>(a.dx) + (long)pix * a.lddx;
      const T* xp = reinterpret_cast<const T*>(a.x) + (long)pix * a.ldx;
      uint2 xm[CT], old[ACCUM ? CT : 1];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + q * 4, cc = c < a.C ? c : 0;
#ifdef HB_EXP_NO_DX
        xm[ct] = uint2{(unsigned)cc, 0u};
        if (ACCUM) old[ACCUM ? ct : 0] = uint2{0u, 0u};
#else
        xm[ct] = *reinterpret_cast<const uint2*>(xp + cc);
        if (ACCUM) old[ACCUM ? ct : 0] = *reinterpret_cast<const uint2*>(dxp + cc);
#endif
      }

      // dp_t = d out . src[tap t] of this lane's taps, taken as soon as the taps have arrived (they were requested first: the later mask / gradient
      // loads stay in flight behind them) -- 8 live registers through the GEMMs below instead of the 24 of the source values
      float dp[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[j][e] = g0 * sv[j][e][0] + g1 * sv[j][e][1] + g2 * sv[j][e][2];
      uint4 hid;
      float p[NT][4];
      forward_gemms_lds<T, KS, NCH>(s_a1, xf, hid, p);
      softmax4<KS>(p, q);
      // d logits of this lane's channels: p_t (dp_t - sum_u p_u dp_u)
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += p[j][e] * dp[j][e];
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      float dlv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) dlv[j * 4 + e] = p[j][e] * (dp[j][e] - dot);
      const uint4 dl = pack8t<T>(dlv);                 // stored in the storage type by the layer-wise path: round here too
      // GEMM 3 + ReLU mask of hid -> d hid (pre-activation), as the k-group of GEMM 4
      // (the ReLU masks are applied to the PACKED results with 16-bit integer ops -- mask_bf16x2: 3 instructions per word -- instead of unpacking
      //  the mask, comparing and selecting per value: the loop is bound by its VALU instruction count)
      float dhv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma16<T>(s_a3[j * 64], dl, acc);
#pragma unroll
        for (int e = 0; e < 4; ++e) dhv[j * 4 + e] = acc[e];
      }
      const uint4 dh = mask_bf16x8(pack8t<T>(dhv), hid);
      // GEMM 4 -> d x, masked by x > 0 (x is a ReLU output of the backbone), optionally accumulated into an existing gradient
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + q * 4;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma16<T>(s_a4[ct * 64], dh, acc);
        uint2 o2;
        o2.x = mask_bf16x2(pack2<T>(acc[0], acc[1]), xm[ct].x);
        o2.y = mask_bf16x2(pack2<T>(acc[2], acc[3]), xm[ct].y);
        if (ACCUM) {
          float f8[8], g8[8];
          unpack8t<T>(uint4{o2.x, o2.y, 0u, 0u}, f8);
          unpack8t<T>(uint4{old[ACCUM ? ct : 0].x, old[ACCUM ? ct : 0].y, 0u, 0u}, g8);
          o2.x = pack2<T>(f8[0] + g8[0], f8[1] + g8[1]);
          o2.y = pack2<T>(f8[2] + g8[2], f8[3] + g8[3]);
        }
#ifndef HB_EXP_NO_DX
        if (pa.ok && c < a.C) *reinterpret_cast<uint2*>(dxp + c) = o2;
#else
        if (o2.x == 0x12345678u) *reinterpret_cast<uint2*>(dxp + c) = o2;
#endif
      }
      // park this pixel's operands of the weight gradients (zero rows for pixels past the end)
      const int row = s * 16 + li;
      // (component-wise selects: `ok ? a : b` on whole uint4 structs makes hipcc select between their ADDRESSES, which put x, d logits and d hid in
      //  scratch memory -- 6 KiB of scratch stores per 16 pixels against 2 KiB of d x, measured as WRITE_SIZE 4x the algorithmic bytes)
      const uint32_t okm = pa.ok ? 0xffffffffu : 0u;
      auto keep4 = [okm](const uint4& v) { return uint4{v.x & okm, v.y & okm, v.z & okm, v.w & okm}; };
      uint4 hs = keep4(hid);
      if (pa.ok && q == (ONES >> 3)) {                 // the all-ones channel: slot ONES & 7 of k-group ONES >> 3
        constexpr int OW = (ONES & 7) >> 1, OSH = ((ONES & 7) & 1) * 16;
        const uint32_t one = (pack2<T>(1.f, 1.f) & 0xffffu) << OSH, keep = ~(0xffffu << OSH);
        if (OW == 0) hs.x = (hs.x & keep) | one;
        else if (OW == 1) hs.y = (hs.y & keep) | one;
        else if (OW == 2) hs.z = (hs.z & keep) | one;
        else hs.w = (hs.w & keep) | one;
      }
      // (one opaque base per strip: left alone hipcc keeps a lane-constant address register per store alive across the whole loop)
      int park = row * HROW + q * 16;
      asm volatile("" : "+v"(park));
      *reinterpret_cast<uint4*>(s_hid + park) = hs;
      *reinterpret_cast<uint4*>(s_hid + 32 * HROW + park) = keep4(dl);
      *reinterpret_cast<uint4*>(s_hid + 2 * 32 * HROW + park) = keep4(dh);
      int parkx = row * XROW + q * 16;
      asm volatile("" : "+v"(parkx));
#pragma unroll
      for (int c = 0; c < NCH; ++c) *reinterpret_cast<uint4*>(s_x + parkx + c * 64) = keep4(xf[c]);
    }
#ifndef HB_EXP_NO_WGRAD
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // weight gradients over these 32 pixels
    uint4 fdl[NPT], fdh[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) { fdl[j] = strip_frag_tr(s_dl, HROW, j, lane); fdh[j] = strip_frag_tr(s_dh, HROW, j, lane); }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const uint4 fh = strip_frag_tr(s_hid, HROW, i, lane);
#pragma unroll
      for (int j = 0; j < NPT; ++j) g_wb[i][j] = mma16<T>(fh, fdl[j], g_wb[i][j]);         // dWb[pos m][pos n] (row ONES: d bb)
      if (i == (ONES >> 4)) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) g_ba[j] = mma16<T>(fh, fdh[j], g_ba[j]);             // row ONES: d ba
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint4 fx = strip_frag_tr(s_x, XROW, ct, lane);
#pragma unroll
      for (int j = 0; j < NPT; ++j) g_wa[ct][j] = mma16<T>(fx, fdh[j], g_wa[ct][j]);       // dWa[c][pos m]
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }

  // flush.  Thousands of waves adding into the same few thousand addresses serialise in the atomic units (measured: ~300 us per launch with
  // one flush per wave), so the waves of a workgroup first sum their tiles through LDS and ONE set of global atomics leaves per workgroup.
  // The sum is a register tree (upper half parks its tiles, lower half adds them: log2(waves) rounds of plain 16-byte stores and loads);
  // LDS float atomics for the same job (ds_add_f32 from 8 waves into one image) took 87-136 k cycles per launch, 40 % of the small launches.
  // D tile (i, j): a lane holds rows i*16 + q*4 + e, column j*16 + li.
  HPH(4);
  __syncthreads();                                     // every wave is done with its strip: the LDS is reused as [wave][tile][lane] x 16 bytes
  HPH(5);
  constexpr int NTILES = NPT * NPT + NPT + CT * NPT;
  static_assert((BWD_WAVES / 2) * NTILES * 1024 <= BWD_WAVES * STRIP, "the parked tiles must fit the strips");
  f32x4_t* red4 = reinterpret_cast<f32x4_t*>(smem);
  auto each_tile = [&](auto&& f) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int j = 0; j < NPT; ++j, ++t) f(t, g_wb[i][j]);
#pragma unroll
    for (int j = 0; j < NPT; ++j, ++t) f(t, g_ba[j]);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int j = 0; j < NPT; ++j, ++t) f(t, g_wa[ct][j]);
  };
#pragma unroll
  for (int half = BWD_WAVES >> 1; half >= 1; half >>= 1) {
    if (wv >= half && wv < 2 * half) each_tile([&](int t, f32x4_t& g) { red4[((wv - half) * NTILES + t) * 64 + lane] = g; });
    __syncthreads();
    if (wv < half) each_tile([&](int t, f32x4_t& g) { const f32x4_t o = red4[(wv * NTILES + t) * 64 + lane]; g[0] += o[0]; g[1] += o[1]; g[2] += o[2]; g[3] += o[3]; });
    __syncthreads();
  }
  if (wv == 0) each_tile([&](int t, f32x4_t& g) { red4[t * 64 + lane] = g; });
  __syncthreads();
  HPH(6);
  const float* red = reinterpret_cast<const float*>(smem);          // element e of lane l of tile t: red[t * 256 + l * 4 + e]
  // 256 threads walk the summed tiles: thread -> (element e, lane) of a tile, exactly the layout above
  const int fl = threadIdx.x & 63, fe = threadIdx.x >> 6, fli = fl & 15, fq = fl >> 4;
#ifndef HB_EXP_NO_FLUSH
  dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
  if (threadIdx.x < 256) {
    // Every workgroup starts at its own tile (round 5): 256 workgroups walking the same ~3 900 addresses in the same order at the same time serialise
    // in the atomic units -- 42 % of the 32 x 32 scale's launch was this flush (tools/head_phases.py; tools/ubench/atomic_scope.hip mode 3).
    for (int tt = 0; tt < NTILES; ++tt) {
      int t = tt + wg % NTILES;
      if (t >= NTILES) t -= NTILES;
      const float v = red[t * 256 + fl * 4 + fe];
      if (t < NPT * NPT) {                                            // dWb[pos m][pos n]; row ONES: d bb
        const int i = t / NPT, j = t - i * NPT;
        const int cpos = j * 16 + fli, n = pos_ch(cpos), pos = i * 16 + fq * 4 + fe, m = pos_ch(pos);
        const bool n_ok = ((cpos & 7) >> 2) < NT && n < K;
        if (!n_ok) continue;
        if (pos == ONES) atomicAdd(a.dbb + n, v);
        else if (((pos & 7) >> 2) < NT && m < K) atomicAdd(a.dwb + m * K + n, v);
      } else if (t < NPT * NPT + NPT) {                               // row ONES: d ba
        const int j = t - NPT * NPT;
        const int cpos = j * 16 + fli, n = pos_ch(cpos), pos = (ONES >> 4) * 16 + fq * 4 + fe;
        const bool n_ok = ((cpos & 7) >> 2) < NT && n < K;
        if (n_ok && pos == ONES) atomicAdd(a.dba + n, v);
      } else {                                                        // dWa[c][pos m]
        const int u = t - NPT * NPT - NPT, ct = u / NPT, j = u - ct * NPT;
        const int cpos = j * 16 + fli, n = pos_ch(cpos), c = ct * 16 + fq * 4 + fe;
        const bool n_ok = ((cpos & 7) >> 2) < NT && n < K;
        if (n_ok && c < a.C) atomicAdd(a.dwa + (long)c * K + n, v);
      }
    }
  }
#endif
  HPH(7);
  dd_det_end();
}

template <typename T, int KS, int NCH, bool ACCUM>
__global__ __launch_bounds__(BWD_WAVES * 64) void head_bwd_kernel(const HeadP a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  head_bwd_body<T, KS, NCH, ACCUM>(a, smem, blockIdx.x, gridDim.x);
}

// The backward of several scales as ONE launch (round 5): the scales' launches are independent, and the small ones are mostly fixed cost -- at the
// 32 x 32 scale the pixel loop is 35 % of the launch (tools/head_phases.py: weight staging, fragment images, the wait for the slowest wave, the
// flush), at 64 x 64 65 %.  Side by side every workgroup pays its fixed cost while the others multiply; workgroups [first[y], first[y + 1]) run
// problem y, split by the host in proportion to the problems' pixel-loop work.
constexpr int HEAD_MAX_MULTI = 3;
struct HeadMultiP { HeadP p[HEAD_MAX_MULTI]; int nch[HEAD_MAX_MULTI]; int first[HEAD_MAX_MULTI + 1]; int n; };
template <typename T, int KS>
__global__ __launch_bounds__(BWD_WAVES * 64) void head_bwd_multi_kernel(const HeadMultiP m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int y = 0;
  while (y + 1 < m.n && (int)blockIdx.x >= m.first[y + 1]) ++y;
  const HeadP& a = m.p[y];
  const int wg = blockIdx.x - m.first[y], nwgs = m.first[y + 1] - m.first[y];
#define HB_CASE(N_)                                                                   \
  case N_: if (a.accumulate) head_bwd_body<T, KS, N_, true>(a, smem, wg, nwgs);      \
           else head_bwd_body<T, KS, N_, false>(a, smem, wg, nwgs);                  \
           break;
  switch (m.nch[y]) { HB_CASE(1) HB_CASE(2) HB_CASE(3) default: if (a.accumulate) head_bwd_body<T, KS, 4, true>(a, smem, wg, nwgs); else head_bwd_body<T, KS, 4, false>(a, smem, wg, nwgs); }
#undef HB_CASE
}

static int head_cus() { return dd_device_cus(); }

template <typename T, int KS, int NCH>
int launch_head(const HeadP& p, bool backward, hipStream_t s) {
  const long groups = (p.npix + 15) / 16;
  if (!backward) {
    long wgs = (groups + 4 * 4 - 1) / (4 * 4);         // >= 4 groups of 16 pixels per wave: the weight gather is paid once per wave
    const long cap = (long)head_cus() * 8;
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL((head_fwd_kernel<T, KS, NCH>), dim3((unsigned)wgs), dim3(256), 0, s, p);
  } else {
    constexpr int STRIP = HeadStrip<NCH>::BYTES;
    const size_t lds = BWD_WAVES * (size_t)STRIP + (size_t)(HeadDim<KS>::NT * NCH + NCH * 2 + 4 * HeadDim<KS>::NT) * 1024;
    dd_det_sync();
    dd_allow_max_lds(reinterpret_cast<const void*>(head_bwd_kernel<T, KS, NCH, false>));
    dd_allow_max_lds(reinterpret_cast<const void*>(head_bwd_kernel<T, KS, NCH, true>));
    long wgs = (long)head_cus();                       // persistent: one flush of the gradient tiles per workgroup
    const long steps = (p.npix + 31) / 32;
    if (wgs * BWD_WAVES > steps) wgs = (steps + BWD_WAVES - 1) / BWD_WAVES;
    if (wgs < 1) wgs = 1;
    if (p.accumulate) hipLaunchKernelGGL((head_bwd_kernel<T, KS, NCH, true>), dim3((unsigned)wgs), dim3(BWD_WAVES * 64), lds, s, p);
    else hipLaunchKernelGGL((head_bwd_kernel<T, KS, NCH, false>), dim3((unsigned)wgs), dim3(BWD_WAVES * 64), lds, s, p);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T, int KS>
int dispatch_nch(const HeadP& p, bool backward, hipStream_t s) {
  switch ((p.C + 31) / 32) {
    case 1: return launch_head<T, KS, 1>(p, backward, s);
    case 2: return launch_head<T, KS, 2>(p, backward, s);
    case 3: return launch_head<T, KS, 3>(p, backward, s);
    case 4: return launch_head<T, KS, 4>(p, backward, s);
    default: dd_set_error("dd_kpcn_head: more than 128 input channels (%d) are not supported by the fused head", p.C); return DD_ERR_INVALID;
  }
}

template <typename T>
int dispatch_ks(const HeadP& p, int ks, bool backward, hipStream_t s) {
  switch (ks) {
    case 3: return dispatch_nch<T, 3>(p, backward, s);
    case 5: return dispatch_nch<T, 5>(p, backward, s);
    default: dd_set_error("dd_kpcn_head: kernel_size %d unsupported by the fused head (3, 5)", ks); return DD_ERR_INVALID;
  }
}

int head_common(const dd_head_args* a, bool backward, dd_stream stream) {
  DD_REQUIRE(a && a->x && a->src && a->wa && a->ba && a->wb && a->bb, "dd_kpcn_head: null pointer");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_kpcn_head: storage dtype must be DD_BF16 or DD_F16 (f32 runs layer by layer)");
  DD_REQUIRE(a->C > 0 && a->C % 8 == 0 && a->ldx >= a->C && a->ldx % 8 == 0 && ((uintptr_t)a->x % 16) == 0, "dd_kpcn_head: C=%d ldx=%d must be multiples of 8, x 16-byte aligned", a->C, a->ldx);
  DD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldsrc >= 3, "dd_kpcn_head: bad shape");
  DD_REQUIRE(a->H >= (a->ksize - 1) / 2 && a->W >= (a->ksize - 1) / 2, "dd_kpcn_head: image smaller than the symmetric pad");
  HeadP p;
  p.x = a->x; p.src = a->src; p.wa = a->wa; p.ba = a->ba; p.wb = a->wb; p.bb = a->bb; p.out = a->out;
  p.dout = a->dout; p.dx = a->dx; p.dwa = a->dwa; p.dba = a->dba; p.dwb = a->dwb; p.dbb = a->dbb;
  p.ldx = a->ldx; p.C = a->C; p.ldsrc = a->ldsrc; p.ldo = a->ld_out; p.lddo = a->ld_dout; p.lddx = a->ld_dx; p.accumulate = a->accumulate_dx;
  p.N = a->N; p.H = a->H; p.W = a->W; p.npix = (long)a->N * a->H * a->W;
  if (!backward) DD_REQUIRE(a->out && a->ld_out >= 3, "dd_kpcn_head_fwd: null output");
  else DD_REQUIRE(a->dout && a->ld_dout >= 3 && a->dx && a->ld_dx >= a->C && a->ld_dx % 4 == 0 && a->dwa && a->dba && a->dwb && a->dbb, "dd_kpcn_head_bwd: null gradient pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return a->dtype == DD_BF16 ? dispatch_ks<bf16_t>(p, a->ksize, backward, s) : dispatch_ks<f16_t>(p, a->ksize, backward, s);
}

template <typename T, int KS>
static void launch_head_bwd_multi(const HeadMultiP& m, size_t lds, int wgs, hipStream_t s) {
  dd_det_sync();
  dd_allow_max_lds(reinterpret_cast<const void*>(head_bwd_multi_kernel<T, KS>));
  hipLaunchKernelGGL((head_bwd_multi_kernel<T, KS>), dim3((unsigned)wgs), dim3(BWD_WAVES * 64), lds, s, m);
}
static size_t head_bwd_lds(int ks, int nch) {
  const int nt = ks == 5 ? HeadDim<5>::NT : HeadDim<3>::NT;
  const size_t strip = nch == 1 ? HeadStrip<1>::BYTES : nch == 2 ? HeadStrip<2>::BYTES : nch == 3 ? HeadStrip<3>::BYTES : HeadStrip<4>::BYTES;
  return BWD_WAVES * strip + (size_t)(nt * nch + nch * 2 + 4 * nt) * 1024;
}

}  // namespace

// The backward of up to three scales' heads as one launch (same storage type and kernel size): see head_bwd_multi_kernel.
extern "C" int dd_kpcn_head_bwd_multi(const dd_head_args* a, int n, dd_stream stream) {
  DD_REQUIRE(a && n >= 1 && n <= HEAD_MAX_MULTI, "dd_kpcn_head_bwd_multi: 1 to %d problems", HEAD_MAX_MULTI);
  if (n == 1) return dd_kpcn_head_bwd(a, stream);
  HeadMultiP m;
  memset(&m, 0, sizeof(m));
  m.n = n;
  double work[HEAD_MAX_MULTI], total = 0.0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    const dd_head_args* b = a + i;
    DD_REQUIRE(b->x && b->src && b->wa && b->ba && b->wb && b->bb, "dd_kpcn_head_bwd_multi: null pointer (problem %d)", i);
    DD_REQUIRE((b->dtype == DD_BF16 || b->dtype == DD_F16) && b->dtype == a->dtype && b->ksize == a->ksize && (b->ksize == 3 || b->ksize == 5),
               "dd_kpcn_head_bwd_multi: one storage type (bf16 / f16) and one kernel size (3, 5) per launch");
    DD_REQUIRE(b->C > 0 && b->C <= 128 && b->C % 8 == 0 && b->ldx >= b->C && b->ldx % 8 == 0 && ((uintptr_t)b->x % 16) == 0, "dd_kpcn_head_bwd_multi: C=%d ldx=%d (problem %d)", b->C, b->ldx, i);
    DD_REQUIRE(b->N > 0 && b->H > 0 && b->W > 0 && b->ldsrc >= 3 && b->H >= (b->ksize - 1) / 2 && b->W >= (b->ksize - 1) / 2, "dd_kpcn_head_bwd_multi: bad shape (problem %d)", i);
    DD_REQUIRE(b->dout && b->ld_dout >= 3 && b->dx && b->ld_dx >= b->C && b->ld_dx % 4 == 0 && b->dwa && b->dba && b->dwb && b->dbb, "dd_kpcn_head_bwd_multi: null gradient pointer (problem %d)", i);
    HeadP& p = m.p[i];
    p.x = b->x; p.src = b->src; p.wa = b->wa; p.ba = b->ba; p.wb = b->wb; p.bb = b->bb; p.out = b->out;
    p.dout = b->dout; p.dx = b->dx; p.dwa = b->dwa; p.dba = b->dba; p.dwb = b->dwb; p.dbb = b->dbb;
    p.ldx = b->ldx; p.C = b->C; p.ldsrc = b->ldsrc; p.ldo = b->ld_out; p.lddo = b->ld_dout; p.lddx = b->ld_dx; p.accumulate = b->accumulate_dx;
    p.N = b->N; p.H = b->H; p.W = b->W; p.npix = (long)b->N * b->H * b->W;
    m.nch[i] = (b->C + 31) / 32;
    // measured cycles per 32-pixel step of a wave (tools/head_phases.py, 5 x 5): 32 / 64 / 96 / 128 channels
    static const double step_cost[5] = {0.0, 12.0, 17.0, 16.0, 23.0};
    work[i] = (double)((p.npix + 31) / 32) * step_cost[m.nch[i]] + 2.0 * BWD_WAVES * step_cost[m.nch[i]];      // (+ a fixed share: nobody gets zero workgroups)
    total += work[i];
    const size_t l = head_bwd_lds(b->ksize, m.nch[i]);
    if (l > lds) lds = l;
  }
  const int cus = head_cus();
  int left = cus;
  m.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    int w = i == n - 1 ? left : (int)(cus * work[i] / total + 0.5);
    const long steps = (m.p[i].npix + 31) / 32;
    if (w < 1) w = 1;
    if ((long)w * BWD_WAVES > steps) w = (int)((steps + BWD_WAVES - 1) / BWD_WAVES);
    if (w > left - (n - 1 - i)) w = left - (n - 1 - i);
    if (w < 1) w = 1;
    m.first[i + 1] = m.first[i] + w;
    left -= w;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int wgs = m.first[n];
  if (a->dtype == DD_BF16) { if (a->ksize == 5) launch_head_bwd_multi<bf16_t, 5>(m, lds, wgs, s); else launch_head_bwd_multi<bf16_t, 3>(m, lds, wgs, s); }
  else { if (a->ksize == 5) launch_head_bwd_multi<f16_t, 5>(m, lds, wgs, s); else launch_head_bwd_multi<f16_t, 3>(m, lds, wgs, s); }
  DD_LAUNCH_CHECK();
  return DD_OK;
}

extern "C" int dd_kpcn_head_fwd(const dd_head_args* a, dd_stream stream) { return head_common(a, false, stream); }
extern "C" int dd_kpcn_head_bwd(const dd_head_args* a, dd_stream stream) { return head_common(a, true, stream); }
