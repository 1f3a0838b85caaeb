// Fused multi-scale compose net on CDNA4: MultiScalePrediction.compose_scales as ONE launch.
//
// Reference seam replaced (file:line in /root/reference): TensorFlow/MultiScalePrediction.py:36-93 (compose_scales ->
// _compose_scales_neural_network -> _residual_block x 2), called once per scale transition by MultiScalePredictor.predict
// (Architecture.py:302-325) with the SAME variables ('reused_compose_scales').
//
//   x0 = [nearest_up(small) | fine]                       6 ch
//   a1 = relu(conv1x1(x0))                               24 ch
//   r1 = conv3x3(relu(a1));  a2 = a1 + conv3x3(relu(r1))
//   r3 = conv3x3(relu(a2));  a3 = a2 + conv3x3(relu(r3))
//   w  = sigmoid(relu(conv1x1(a3)))                       1 ch
//   out = fine - w * nearest_up(avg_pool2(fine)) + w * nearest_up(small)
//
// The layer-by-layer path moves nine 24-channel tensors through HBM for 1 % of the network's flops (13 % of a training step in
// round 1).  Here one workgroup owns a 16x16 output tile and recomputes the 4-pixel halo: the 24x24 frame of a1 and the shrinking
// frames of r1 (22x22), a2 (20x20), r3 (18x18) live in LDS as [pixel][24 channels] in the storage type; only the 6 input and 3
// output channels touch HBM (plus, for training, the activations the layer-wise backward reads: each tile stores its own 16x16
// interior).  The 3x3 layers run on MFMA with the 9 x 24 = 216 reduction values packed DENSELY into 7 K-chunks of 32 (27 k-groups
// of 8 channels; a lane's k-group picks its own tap), instead of 9 chunks with 8 of every 32 lanes idle.
// Zero padding is TensorFlow's: every conv pads ITS input, so each layer's frame is forced to zero outside the image.
#include "dd_common.h"
#include <stdlib.h>

#ifdef DD_PROFILE_PHASES
// cycle stamps of workgroup 0 (tools/compose_phases.py): slot i accumulates the cycles between stamp i-1 and stamp i of the chosen thread
__device__ unsigned long long dd_cphase[64];
#define CPH_DECL(t) unsigned long long cph_last = __builtin_readcyclecounter(); const bool cph_on = blockIdx.x == 0 && threadIdx.x == (t)
#define CPH(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (cph_on) dd_cphase[i] += now_ - cph_last; cph_last = now_; } while (0)
extern "C" int dd_debug_cphases(unsigned long long* out64, int reset) {
  if (out64) (void)hipMemcpyFromSymbol(out64, HIP_SYMBOL(dd_cphase), sizeof(unsigned long long) * 64);
  if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dd_cphase), z, sizeof(z)); }
  return 0;
}
#else
#define CPH_DECL(t)
#define CPH(i)
#endif

namespace {

constexpr int FR = 24;                      // frame side: 16 + 2 * 4
constexpr int PIXB = 48;                    // bytes of one pixel: 24 channels x 2
constexpr int NCHUNK = 7;                   // K chunks of 32 per 3x3 layer
constexpr int WL_BYTES = NCHUNK * 32 * 64;  // A-operand image of one layer: [chunk][32 rows (24 real)][32 k]
constexpr int BUF_BYTES = FR * FR * PIXB;
constexpr int NFP = 6 * 24 + 24 + 4 * 24 + 24 + 4;   // fp32 parameters kept in LDS: w_in, b_in, b_res[4], w_out, b_out (+pad)
constexpr int LDS_TOTAL = 4 * WL_BYTES + 3 * BUF_BYTES + 256 * 6 * 4 + NFP * 4;

struct ComposeP {
  const float* small; const float* fine; float* out;
  const float* w_in; const float* b_in; const float* w_res[4]; const float* b_res[4]; const float* w_out; const float* b_out;
  void* save_netin; void* save_act[5]; void* save_wl;
  int ld_small, ld_fine, ld_out, ld_netin, ld_act[5], ld_wl;
  int N, H, W, tiles_x, tiles_y, total_tiles;
};

// 64-byte weight rows: slot s of row r at physical slot s ^ ((r >> 3 & 1) << 1) (conflict-free ds_read_b128 service groups)
__device__ __forceinline__ int w_off(int row, int slot) { return row * 64 + ((slot ^ (((row >> 3) & 1) << 1)) << 4); }

template <typename T>
__device__ __forceinline__ void stage_parameters(char* wts, float* fp, const ComposeP& p, int tid) {
  for (int i = tid; i < 4 * WL_BYTES / 16; i += 512) reinterpret_cast<uint4*>(wts)[i] = uint4{0u, 0u, 0u, 0u};
  for (int i = tid; i < NFP; i += 512) {
    // the 1x1 layers' WEIGHTS are rounded to the storage type, as the packed MFMA operands of the layer-wise path are; biases stay fp32
    float v = 0.f;
    if (i < 144) v = Elem<T>::to_f32(Elem<T>::from_f32(p.w_in[i]));
    else if (i < 168) v = p.b_in[i - 144];
    else if (i < 264) v = p.b_res[(i - 168) / 24][(i - 168) % 24];
    else if (i < 288) v = Elem<T>::to_f32(Elem<T>::from_f32(p.w_out[i - 264]));
    else if (i == 288) v = p.b_out[0];
    fp[i] = v;
  }
  __syncthreads();
  // HWIO fp32 master weights [tap][ci][co] -> A image: row = co, k-group g = tap * 3 + ci / 8, element ci % 8
  for (int e = tid; e < 4 * 5184; e += 512) {
    const int l = e / 5184, rem = e - l * 5184;
    const int tap = rem / 576, ci = (rem / 24) % 24, co = rem % 24;
    const int g = tap * 3 + (ci >> 3);
    T* dst = reinterpret_cast<T*>(wts + l * WL_BYTES + (g >> 2) * 2048 + w_off(co, g & 3)) + (ci & 7);
    *dst = Elem<T>::from_f32(p.w_res[l][rem]);
  }
}

// Copy the 16x16 interior of a frame buffer to global memory (training: the activations the layer-wise backward reads).
__device__ __forceinline__ void save_interior(const char* buf, void* dst, int ld, int b, int y0, int x0, int H, int W, int tid) {
  if (!dst) return;
  char* base = reinterpret_cast<char*>(dst);
  for (int v = tid; v < 256 * 3; v += 512) {
    const int px = v / 3, slot = v - px * 3;
    const int y = px >> 4, x = px & 15;
    if (y0 + y < H && x0 + x < W)
      *reinterpret_cast<uint4*>(base + (((long)b * H + y0 + y) * W + x0 + x) * ld * 2 + slot * 16) =
          *reinterpret_cast<const uint4*>(buf + ((4 + y) * FR + 4 + x) * PIXB + slot * 16);
  }
}

// One 3x3 layer on the frame: reads `in` on [L-1, 24-L+1)^2, writes `out` on [L, 24-L)^2.
//   RELU_IN : the operand is relu(in) (in holds a raw residual-stream tensor)
//   RESIDUAL: out = res + conv (raw), else out = relu(conv)
template <typename T, int L, bool RELU_IN, bool RESIDUAL>
__device__ __forceinline__ void conv_layer(const char* in, char* out, const char* res, const char* wl, const float* bias,
                                           const int (&koff)[NCHUNK], int y0, int x0, int H, int W, int wave, int li, int q) {
  constexpr int R = FR - 2 * L, NPIX = R * R, CHUNKS = (NPIX + 15) / 16;
  uint4 wf[2][NCHUNK];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) wf[j][c] = *reinterpret_cast<const uint4*>(wl + c * 2048 + w_off(j * 16 + li, q));
  f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias + q * 4);
  f32x4_t b1 = q < 2 ? *reinterpret_cast<const f32x4_t*>(bias + 16 + q * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int chunk = wave; chunk < CHUNKS; chunk += 8) {
    const int P = chunk * 16 + li;
    const int Pc = P < NPIX ? P : NPIX - 1;
    const int y = Pc / R, x = Pc - y * R;
    const char* base = in + ((L - 1 + y) * FR + (L - 1 + x)) * PIXB;
    uint4 bf[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) bf[c] = *reinterpret_cast<const uint4*>(base + koff[c]);
    f32x4_t a0 = b0, a1 = b1;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const uint4 v = RELU_IN ? relu16<T>(bf[c]) : bf[c];
      a0 = mma16<T>(wf[0][c], v, a0);
      a1 = mma16<T>(wf[1][c], v, a1);
    }
    const int fy = L + y, fx = L + x;
    const bool inside = (unsigned)(y0 - 4 + fy) < (unsigned)H && (unsigned)(x0 - 4 + fx) < (unsigned)W;
    char* o = out + (fy * FR + fx) * PIXB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && q >= 2) continue;                    // channels 24..31 do not exist
      f32x4_t a = j == 0 ? a0 : a1;
      const int ch = j * 16 + q * 4;
      if (RESIDUAL) {
        float r[4];
        load4<T>(reinterpret_cast<const T*>(res + (fy * FR + fx) * PIXB) + ch, r);
        a[0] += r[0]; a[1] += r[1]; a[2] += r[2]; a[3] += r[3];
      }
      uint2 pk;                                          // (ReLU on the packed result: one v_pk_max_i16 per word instead of two v_max_f32 per value)
      pk.x = pack2<T>(a[0], a[1]); pk.y = pack2<T>(a[2], a[3]);
      if (!RESIDUAL) { pk.x = relu_bf16x2(pk.x); pk.y = relu_bf16x2(pk.y); }
      pk.x = inside ? pk.x : 0u;                         // TensorFlow pads every conv's input with zeros: nothing exists outside the image
      pk.y = inside ? pk.y : 0u;
      if (P < NPIX) *reinterpret_cast<uint2*>(o + ch * 2) = pk;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(512) void compose_fwd_kernel(const ComposeP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wts = smem;
  char* bufA = smem + 4 * WL_BYTES;
  char* bufB = bufA + BUF_BYTES;
  char* bufC = bufB + BUF_BYTES;
  float* stash = reinterpret_cast<float*>(bufC + BUF_BYTES);      // x0 of the 16x16 interior: [pixel][6]
  float* fp = stash + 256 * 6;
  const float* w_in = fp; const float* b_in = fp + 144; const float* b_res = fp + 168; const float* w_out = fp + 264; const float* b_out = fp + 288;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  stage_parameters<T>(wts, fp, p, tid);
  // byte offset of this lane's k-group of K-chunk c from a chunk pixel's (-1,-1) neighbour: group g = 4c + q -> tap g / 3, channels 8 (g % 3)
  int koff[NCHUNK];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int g = min(4 * c + q, 26);                    // group 27 is padding: its weights are zero, read any finite data
    const int tap = g / 3, cg = g - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    koff[c] = (dy * FR + dx) * PIXB + cg * 16;
  }
  __syncthreads();

  // layer 0 on the matrix pipe: this lane's A row = output channel j*16 + li, its 6 input weights as one 8-element k-group; bias of the 4
  // channels (j*16 + q*4 ..) it receives
  static_assert(FR * FR % 64 == 0, "layer 0 works on whole waves");
  uint4 w0a, w0b;
  float bias0[8];
  {
    float w[2][6];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 6; ++k) w[j][k] = j * 16 + li < 24 ? w_in[k * 24 + j * 16 + li] : 0.f;
    w0a = uint4{pack2<T>(w[0][0], w[0][1]), pack2<T>(w[0][2], w[0][3]), pack2<T>(w[0][4], w[0][5]), 0u};
    w0b = uint4{pack2<T>(w[1][0], w[1][1]), pack2<T>(w[1][2], w[1][3]), pack2<T>(w[1][4], w[1][5]), 0u};
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias0[e] = b_in[q * 4 + e]; bias0[4 + e] = q < 2 ? b_in[16 + q * 4 + e] : 0.f; }
  }
  const int per_img = p.tiles_y * p.tiles_x;
  const int H = p.H, W = p.W, h2 = H >> 1, w2 = W >> 1;
  // The net input of the NEXT tile (6 fp32 values for each of this thread's 1-2 frame pixels) is requested while the current tile's four
  // conv layers run: with the loads in front of layer 0 every tile started with an exposed memory round trip (~2 us of ~13).
  constexpr int NPF = (FR * FR + 511) / 512;
  float xin[NPF][6];
  auto load_x0 = [&](int tile) {
    const int tt = tile < p.total_tiles ? tile : p.total_tiles - 1;
    const int b = tt / per_img, rem = tt - b * per_img, ty = rem / p.tiles_x;
    const int y0 = ty * 16, x0 = (rem - ty * p.tiles_x) * 16;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int idx = tid + i * 512;
      const int fy = idx / FR, fx = idx - fy * FR;
      const int gy = y0 - 4 + fy, gx = x0 - 4 + fx;
      const bool inside = idx < FR * FR && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const int cy = inside ? gy : 0, cx = inside ? gx : 0;              // unconditional loads from a clamped pixel, zeroed below
      const float* s = p.small + (((long)b * h2 + (cy >> 1)) * w2 + (cx >> 1)) * p.ld_small;
      const float* f = p.fine + (((long)b * H + cy) * W + cx) * p.ld_fine;
      const float m = inside ? 1.f : 0.f;
      xin[i][0] = m * s[0]; xin[i][1] = m * s[1]; xin[i][2] = m * s[2]; xin[i][3] = m * f[0]; xin[i][4] = m * f[1]; xin[i][5] = m * f[2];
    }
  };
  load_x0(blockIdx.x);
  CPH_DECL(0);
  for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
    CPH(0);
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / p.tiles_x;
    const int y0 = ty * 16, x0 = (rem - ty * p.tiles_x) * 16;

    // ---------------------------------------------------------------- layer 0: x0 -> a1 = relu(1x1) on the whole 24x24 frame
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int idx = tid + i * 512;
      if (wave * 64 + i * 512 >= FR * FR) continue;      // whole waves: 576 = 9 x 64 frame pixels, the MFMAs below need all 64 lanes
      const int fy = idx / FR, fx = idx - fy * FR;
      const int gy = y0 - 4 + fy, gx = x0 - 4 + fx;
      const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      float x[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xin[i][k];
      const bool interior = fy >= 4 && fy < 20 && fx >= 4 && fx < 20;
      if (interior) {
        float* st = stash + ((fy - 4) * 16 + (fx - 4)) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) st[k] = x[k];
        if (p.save_netin && inside) {                      // the 1x1 layer's input in the storage type (its weight gradient reads it)
          uint4 v;
          v.x = pack2<T>(x[0], x[1]); v.y = pack2<T>(x[2], x[3]); v.z = pack2<T>(x[4], x[5]); v.w = 0u;
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.save_netin) + (((long)b * H + gy) * W + gx) * p.ld_netin * 2) = v;
        }
      }
      // The 1x1 on the matrix pipe.  Every lane holds ITS pixel's 6 inputs (rounded to the storage type, where the layer-wise path stores the
      // packed net input) as the 8-element k-group of an MFMA B operand; with the weights placed in k-group g of the A operand and zeros in the
      // other three, one MFMA is the layer for the 16 pixels held by the lanes of group g: 4 groups x 2 channel tiles = 8 MFMAs per 64
      // pixels, results as [4 channels][pixel] per lane.  (As 144 FMAs per pixel on the VALU this layer was 4.7 k of a tile's 29 k cycles
      // in a kernel bound by its VALU work.)
      const uint4 bx = {pack2<T>(x[0], x[1]), pack2<T>(x[2], x[3]), pack2<T>(x[4], x[5]), 0u};
      const unsigned long long in_mask = __ballot(inside);
      const uint4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint4 wa = q == g ? w0a : z4, wb = q == g ? w0b : z4;
        const f32x4_t d0 = mma16<T>(wa, bx, f32x4_t{bias0[0], bias0[1], bias0[2], bias0[3]});
        const f32x4_t d1 = mma16<T>(wb, bx, f32x4_t{bias0[4], bias0[5], bias0[6], bias0[7]});
        const int pl = g * 16 + li;                        // the pixel these results belong to: lane pl of this wave
        const bool ins = (in_mask >> pl) & 1ull;
        char* dst = bufA + (idx - lane + pl) * PIXB;
        uint2 o0, o1;
        o0.x = ins ? relu_bf16x2(pack2<T>(d0[0], d0[1])) : 0u;
        o0.y = ins ? relu_bf16x2(pack2<T>(d0[2], d0[3])) : 0u;
        o1.x = ins ? relu_bf16x2(pack2<T>(d1[0], d1[1])) : 0u;
        o1.y = ins ? relu_bf16x2(pack2<T>(d1[2], d1[3])) : 0u;
        *reinterpret_cast<uint2*>(dst + q * 8) = o0;                          // channels q*4 .. q*4+3
        if (q < 2) *reinterpret_cast<uint2*>(dst + 32 + q * 8) = o1;          // channels 16 + q*4 ..
      }
    }
    CPH(1);
    __syncthreads();
    CPH(2);
    load_x0(tile + gridDim.x);                             // in flight during the conv layers of this tile
    save_interior(bufA, p.save_act[0], p.ld_act[0], b, y0, x0, H, W, tid);
    CPH(3);
    conv_layer<T, 1, false, false>(bufA, bufB, nullptr, wts, b_res, koff, y0, x0, H, W, wave, li, q);                 // relu(r1)
    CPH(4);
    __syncthreads();
    CPH(5);
    save_interior(bufB, p.save_act[1], p.ld_act[1], b, y0, x0, H, W, tid);
    CPH(6);
    conv_layer<T, 2, false, true>(bufB, bufC, bufA, wts + WL_BYTES, b_res + 24, koff, y0, x0, H, W, wave, li, q);     // a2 = a1 + conv
    CPH(7);
    __syncthreads();
    CPH(8);
    save_interior(bufC, p.save_act[2], p.ld_act[2], b, y0, x0, H, W, tid);
    CPH(9);
    conv_layer<T, 3, true, false>(bufC, bufB, nullptr, wts + 2 * WL_BYTES, b_res + 48, koff, y0, x0, H, W, wave, li, q);   // relu(r3)
    CPH(10);
    __syncthreads();
    CPH(11);
    save_interior(bufB, p.save_act[3], p.ld_act[3], b, y0, x0, H, W, tid);
    CPH(12);

    // ---------------------------------------------------------------- layer 4 on the 16x16 interior + 1x1 -> sigmoid -> blend
    {
      const char* wl = wts + 3 * WL_BYTES;
      uint4 wf[2][NCHUNK];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) wf[j][c] = *reinterpret_cast<const uint4*>(wl + c * 2048 + w_off(j * 16 + li, q));
      const f32x4_t bb0 = *reinterpret_cast<const f32x4_t*>(b_res + 72 + q * 4);
      const f32x4_t bb1 = q < 2 ? *reinterpret_cast<const f32x4_t*>(b_res + 72 + 16 + q * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      const f32x4_t wo0 = *reinterpret_cast<const f32x4_t*>(w_out + q * 4);
      const f32x4_t wo1 = q < 2 ? *reinterpret_cast<const f32x4_t*>(w_out + 16 + q * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int y = wave; y < 16; y += 8) {               // chunk = one row of the interior: pixel (y, li)
        const char* base = bufB + ((3 + y) * FR + (3 + li)) * PIXB;
        uint4 bf[NCHUNK];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) bf[c] = *reinterpret_cast<const uint4*>(base + koff[c]);
        f32x4_t a0 = bb0, a1 = bb1;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
          a0 = mma16<T>(wf[0][c], bf[c], a0);
          a1 = mma16<T>(wf[1][c], bf[c], a1);
        }
        const int gy = y0 + y, gx = x0 + li;
        const bool inside = gy < H && gx < W;
        const long pix = ((long)b * H + gy) * W + gx;
        float r0[4], r1[4] = {0.f, 0.f, 0.f, 0.f};
        const T* resp = reinterpret_cast<const T*>(bufC + ((4 + y) * FR + 4 + li) * PIXB);
        load4<T>(resp + q * 4, r0);
        if (q < 2) load4<T>(resp + 16 + q * 4, r1);
        // a3 in the storage type: the layer-wise path stores it before the 1x1 conv reads it
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = Elem<T>::to_f32(Elem<T>::from_f32(a0[e] + r0[e]));
          v1[e] = Elem<T>::to_f32(Elem<T>::from_f32(a1[e] + r1[e]));
        }
        if (p.save_act[4] && inside) {
          T* d = reinterpret_cast<T*>(p.save_act[4]) + pix * p.ld_act[4];
          store4<T>(d + q * 4, v0);
          if (q < 2) store4<T>(d + 16 + q * 4, v1);
        }
        float part = v0[0] * wo0[0] + v0[1] * wo0[1] + v0[2] * wo0[2] + v0[3] * wo0[3]
                   + v1[0] * wo1[0] + v1[1] * wo1[1] + v1[2] * wo1[2] + v1[3] * wo1[3];
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float wlv = Elem<T>::to_f32(Elem<T>::from_f32(fmaxf(part + b_out[0], 0.f)));   // relu, stored in the storage type
        const float w = 1.f / (1.f + __expf(-wlv));
        if (inside) {
          if (q < 3) {
            const float* st = stash + (y * 16 + li) * 6;
            const float* blk = stash + ((y & ~1) * 16 + (li & ~1)) * 6 + 3 + q;
            const float low = 0.25f * (blk[0] + blk[6] + blk[16 * 6] + blk[17 * 6]);
            p.out[pix * p.ld_out + q] = st[3 + q] - w * low + w * st[q];
          } else if (p.save_wl) {
            reinterpret_cast<T*>(p.save_wl)[pix * p.ld_wl] = Elem<T>::from_f32(wlv);
          }
        }
      }
    }
    CPH(13);
    __syncthreads();      // bufB / bufC / stash are rewritten by the next tile
    CPH(14);
#ifdef DD_PROFILE_PHASES
    if (cph_on) dd_cphase[15] += 1;
#endif
  }
}


// ============================================================================================================================
// Backward of the whole compose net as ONE launch (TF autodiff of MultiScalePrediction.py:36-93 behind Training.py:701-702).
//
// Gradients run the net in reverse on the same 24x24 frame of a 16x16 tile (4-pixel halo of OUTPUT gradients is recomputed):
//   S0  dz6 = d(out)/d(wl) on the whole frame -> dA = d a3 (frame G0);     direct blend gradients of the interior are stashed
//   S1  dK4 += r3 (x) dA;        dc3 = (K4^T * dA) . [r3 > 0]           -> G1   (22x22)
//   S2  dK3 += relu(a2) (x) dc3; d a2 = dA + (K3^T * dc3) . [a2 > 0]    -> G0   (20x20, in place)
//   S3  dK2 += r1 (x) d a2;      dc1 = (K2^T * d a2) . [r1 > 0]         -> G1   (18x18)
//   S4  dK1 += a1 (x) dc1;       dz1 = (d a2 + K1^T * dc1) . [a1 > 0]   -> G0   (16x16, in place)
//   S5  d x0 = W1^T dz1 -> d fine, d small (2x2 sums inside the tile: the tile origin is even);  dW1 += x0 (x) dz1
// The forward activations come from the tensors the fused forward stored (a1, relu(r1), a2, relu(r3), a3, wl); each stage loads the
// frame it needs (zeros outside the image, which also makes every out-of-image gradient vanish through its mask) into one LDS buffer,
// prefetched into registers during the previous stage.  The data-gradient convs reuse the forward's dense-K MFMA scheme with the
// flipped / transposed weights; the weight gradients run on MFMA too ([pixel][channel] frames read with ds_read_b64_tr_b16, as
// csrc/dd_conv_wgrad.hip does) and accumulate in REGISTERS over all tiles of the workgroup: one atomic flush per launch.  The bias
// gradient of a 3x3 layer is row 24 of its centre-tap weight-gradient tile (an all-ones input channel).
struct ComposeBwdP {
  const float* small; const float* fine; const float* gout;
  const void* act[5]; const void* wl;
  const float* w_in; const float* w_res[4]; const float* w_out;
  float* d_small; float* d_fine;
  float* dw_in; float* db_in; float* dw_res[4]; float* db_res[4]; float* dw_out; float* db_out;
  int ld_small, ld_fine, ld_gout, ld_act[5], ld_wl, ld_dsmall, ld_dfine, acc_small;
  int N, H, W, tiles_x, tiles_y, total_tiles;
};

// experiment switches (tools/build_variant.sh): knock out one part of the backward to see what it costs (results are then wrong)
#ifdef CB_EXP_NO_WGRAD
#define CB_WGRAD(...)
#else
#define CB_WGRAD(...) __VA_ARGS__
#endif
#ifdef CB_EXP_NO_DGRAD
#define CB_DGRAD(...)
#else
#define CB_DGRAD(...) __VA_ARGS__
#endif

constexpr int NFPB = 144 + 24;       // fp32 parameters of the backward kept in LDS: w_in (rounded), w_out (rounded)
constexpr int LDS_BWD = 4 * WL_BYTES + 3 * BUF_BYTES + 256 * 4 * 4 + 256 * 6 * 4 + 256 * 4 + NFPB * 4;

// 12 waves (3 per SIMD, 168 registers each): 6 data-gradient + 6 weight-gradient.  Round 2 shipped 16 waves (4 per SIMD, 128 registers: 8 + 8),
// which hipcc could only build with 37 spilled registers -- 176 bytes of scratch per lane, 305 MB of scratch writes per launch against 20 MB
// of algorithmic output (profiles/r02_j_pmc_WRITE_SIZE.txt).  CB_WAVES=16 still builds that geometry for A/B runs (tools/build_variant.sh).
#ifndef CB_WAVES
#define CB_WAVES 12
#endif
constexpr int BWD_WAVES = CB_WAVES, BWD_THREADS = BWD_WAVES * 64;
constexpr int DG_WAVES = BWD_WAVES / 2, DG_THREADS = DG_WAVES * 64;      // data-gradient role: waves 0 .. DG_WAVES - 1
constexpr int WG_NT = CB_WAVES == 16 ? 5 : 3, WG_NJ = CB_WAVES == 16 ? 1 : 2;
static_assert(CB_WAVES == 12 || CB_WAVES == 16, "compose backward: 12 or 16 waves");
// S5: threads 0..255 (data-gradient role) take one interior pixel each; 384 threads take the two 1x1 layers' weight gradients: 256..639 with
// 16 waves (half of them in either role), the whole weight-gradient role with 12
constexpr int W1_FIRST = CB_WAVES == 16 ? 256 : DG_THREADS;
static_assert(DG_THREADS >= 256 && W1_FIRST + 384 <= BWD_THREADS, "S5 thread assignment");

typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;

// Transposed MFMA fragment of a [pixel][24 channel] frame: the 8 k-values (pixels) of channel `lane & 15` of channel tile `ctile` for the
// 32-pixel k-step made of interior rows 2*kst, 2*kst+1 shifted by (sy, sx).  Lane t of a 16-lane group addresses pixel t >> 2 of its
// group's 4-pixel run and channel quad t & 3 (ds_read_b64_tr_b16 semantics: tests/test_gpu_ops.py::test_tr16_probe).
__device__ __forceinline__ uint4 frame_frag_tr(const char* frame, int kst, int sy, int sx, int ctile, int lane) {
  const int t16 = lane & 15, g = lane >> 4;
  const int y = 4 + 2 * kst + (g >> 1) + sy, x = 4 + (g & 1) * 8 + (t16 >> 2) + sx;
  const char* a0 = frame + (y * FR + x) * PIXB + ctile * 32 + (t16 & 3) * 8;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a0 + 4 * PIXB));
  uint4 r;
  r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}

// Weight gradient of one 3x3 layer for this tile's 256 interior pixels: acc[i * NJ + j] += act[p + tap_i] (x) dc[p] for the wave's NT taps
// (tap0 .. tap0 + NT - 1, those beyond 8 idle) and NJ output-channel tiles (nj0 .. nj0 + NJ - 1) of input-channel tile mi.
//   16 waves: 8 weight-gradient waves = (mi, nj, tap group of 5): NT = 5, NJ = 1, 20 accumulator tiles per wave;
//   12 waves: 6 weight-gradient waves = (mi, tap group of 3), both output-channel tiles: NT = 3, NJ = 2, 24 tiles per wave, 5 fragment
//             reads per 6 MFMAs instead of 6 per 5.
template <typename T, bool RELU_A, int NT, int NJ>
__device__ __forceinline__ void wgrad_stage(f32x4_t (&acc)[NT * NJ], const char* act, const char* dc, int mi, int nj0, int tap0, int lane) {
  const uint32_t one2 = pack2<T>(1.f, 1.f);
  const bool ones_lane = mi == 1 && (lane & 15) == 8;      // channel 24 := 1  =>  row 24 of the centre-tap tile = bias gradient
  // this lane's part of every transposed fragment read (frame_frag_tr): pixel (g >> 1, (g & 1) * 8 + (t16 >> 2)) of the k-step, channel quad
  // t16 & 3; k-step and the +4-pixel second half are immediates on top of a few address registers
  const int t16 = lane & 15, g = lane >> 4;
  const int loff = ((4 + (g >> 1)) * FR + 4 + (g & 1) * 8 + (t16 >> 2)) * PIXB + (t16 & 3) * 8;
  const char* pq = dc + loff + nj0 * 32;
  const char* pa[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int tap = min(tap0 + i, 8);
    pa[i] = act + loff + mi * 32 + ((tap / 3 - 1) * FR + (tap % 3 - 1)) * PIXB;
  }
  auto tr = [](const char* a) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(a + 4 * PIXB));
    uint4 r;
    r.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  constexpr int GRP = NT > 3 ? 3 : NT;      // fragments in flight at a time
#pragma unroll 1
  for (int kst = 0; kst < 8; ++kst) {
    const int koffs = kst * 2 * FR * PIXB;
    uint4 bq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bq[j] = tr(pq + j * 32 + koffs);
#pragma unroll
    for (int i0 = 0; i0 < NT; i0 += GRP) {
      uint4 ap[GRP];
#pragma unroll
      for (int i = i0; i < i0 + GRP && i < NT; ++i) {
        ap[i - i0] = tr(pa[i] + koffs);
        if (RELU_A) ap[i - i0] = relu16<T>(ap[i - i0]);
        if (ones_lane && tap0 + i == 4) ap[i - i0] = uint4{one2, one2, one2, one2};
      }
#pragma unroll
      for (int i = i0; i < i0 + GRP && i < NT; ++i)
        if (tap0 + i <= 8) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i * NJ + j] = mma16<T>(ap[i - i0], bq[j], acc[i * NJ + j]);
        }
    }
  }
}

// Data gradient of one 3x3 layer on the frame (the forward's geometry: reads `in` on [L-1, 25-L)^2, writes `out` on [L, 24-L)^2),
// run by the DG_WAVES data-gradient waves (w8 = 0 .. DG_WAVES - 1).
//   MODE 0: out = conv . [mask > 0]       MODE 1: out = res + conv . [mask > 0]       MODE 2: out = (res + conv) . [mask > 0]
// 16 waves (128 registers per wave): the weight fragments are re-read from LDS for every chunk, two K-chunks ahead of their MFMAs;
// 12 waves (168 registers): the layer's 14 weight fragments stay in registers for the whole stage (a chunk then reads 7 operand vectors
// from LDS instead of 21).
template <typename T, int L, int MODE>
__device__ __forceinline__ void dgrad_layer(const char* in, char* out, const char* res, const char* mask, const char* wl,
                                            const int (&koff)[NCHUNK], int w8, int li, int q) {
  constexpr int R = FR - 2 * L, NPIX = R * R, CHUNKS = (NPIX + 15) / 16;
  const char* w0 = wl + w_off(li, q);
  const char* w1 = wl + w_off(16 + li, q);
#if CB_WAVES == 12
  uint4 wfa[NCHUNK], wfb[NCHUNK];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    wfa[c] = *reinterpret_cast<const uint4*>(w0 + c * 2048);
    wfb[c] = *reinterpret_cast<const uint4*>(w1 + c * 2048);
  }
#endif
  for (int chunk = w8; chunk < CHUNKS; chunk += DG_WAVES) {
    const int P = chunk * 16 + li;
    const int Pc = P < NPIX ? P : NPIX - 1;
    const int y = Pc / R, x = Pc - y * R;
    const char* base = in + ((L - 1 + y) * FR + (L - 1 + x)) * PIXB;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#if CB_WAVES == 12
    uint4 bf[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) bf[c] = *reinterpret_cast<const uint4*>(base + koff[c]);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      a0 = mma16<T>(wfa[c], bf[c], a0);
      a1 = mma16<T>(wfb[c], bf[c], a1);
    }
#else
    uint4 bf[3], wa[3], wb[3];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf[c] = *reinterpret_cast<const uint4*>(base + koff[c]);
      wa[c] = *reinterpret_cast<const uint4*>(w0 + c * 2048);
      wb[c] = *reinterpret_cast<const uint4*>(w1 + c * 2048);
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      if (c + 2 < NCHUNK) {
        bf[(c + 2) % 3] = *reinterpret_cast<const uint4*>(base + koff[c + 2]);
        wa[(c + 2) % 3] = *reinterpret_cast<const uint4*>(w0 + (c + 2) * 2048);
        wb[(c + 2) % 3] = *reinterpret_cast<const uint4*>(w1 + (c + 2) * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
      a0 = mma16<T>(wa[c % 3], bf[c % 3], a0);
      a1 = mma16<T>(wb[c % 3], bf[c % 3], a1);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    const int po = ((L + y) * FR + (L + x)) * PIXB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && q >= 2) continue;
      const f32x4_t a = j == 0 ? a0 : a1;
      const int ch = j * 16 + q * 4;
      float m[4], r[4] = {0.f, 0.f, 0.f, 0.f}, v[4];
      load4<T>(reinterpret_cast<const T*>(mask + po) + ch, m);
      if (MODE != 0) load4<T>(reinterpret_cast<const T*>(res + po) + ch, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 2) v[e] = m[e] > 0.f ? r[e] + a[e] : 0.f;
        else v[e] = r[e] + (m[e] > 0.f ? a[e] : 0.f);
      }
      if (P < NPIX) store4<T>(reinterpret_cast<T*>(out + po) + ch, v);
    }
  }
}

// Activation frame of one stage: region [LO, 24-LO)^2 of the 24x24 frame, 3 x 16 bytes per pixel, zeros outside the image.
// The frame-copy / per-pixel index arithmetic depends on the thread index only: left alone, hipcc hoists it out of the tile loop for all four
// activation regions at once and pays for it with dozens of live registers (then spills around the MFMA stages).  An opaque copy of the thread
// index per use keeps each piece of address arithmetic next to its loads and stores.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

template <int LO> struct ActRegion { static constexpr int R = FR - 2 * LO, NV = R * R * 3, ITERS = (NV + BWD_THREADS - 1) / BWD_THREADS; };
// (pre0 / pre1 are two plain registers-quads: as a uint4[2] array indexed by the unrolled loop they were kept in scratch memory)
template <int LO>
__device__ __forceinline__ uint4 act_load_one(const char* base, const char* zero, int ld, int b, int y0, int x0, int H, int W, int v) {
  constexpr int R = ActRegion<LO>::R, NV = ActRegion<LO>::NV;
  const int px = v / 3, slot = v - px * 3;
  const int fy = LO + px / R, fx = LO + px % R;
  const int gy = y0 - 4 + fy, gx = x0 - 4 + fx;
  const bool ok = v < NV && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
  return *reinterpret_cast<const uint4*>(ok ? base + (((long)b * H + gy) * W + gx) * ld * 2 + slot * 16 : zero);
}
template <int LO>
__device__ __forceinline__ void act_load(uint4& pre0, uint4& pre1, const void* src, int ld, int b, int y0, int x0, int H, int W, int tid) {
  static_assert(ActRegion<LO>::ITERS <= 2, "two prefetch registers-quads per thread");
  const char* base = reinterpret_cast<const char*>(src);
  const char* zero = reinterpret_cast<const char*>(&dd_zero16_v);
  tid = opaque(tid);
  pre0 = act_load_one<LO>(base, zero, ld, b, y0, x0, H, W, tid);
  if (ActRegion<LO>::ITERS > 1) pre1 = act_load_one<LO>(base, zero, ld, b, y0, x0, H, W, tid + BWD_THREADS);
}
template <int LO>
__device__ __forceinline__ void act_store_one(char* buf, const uint4& pre, int v) {
  constexpr int R = ActRegion<LO>::R, NV = ActRegion<LO>::NV;
  const int px = v / 3, slot = v - px * 3;
  const int fy = LO + px / R, fx = LO + px % R;
  if (v < NV) *reinterpret_cast<uint4*>(buf + (fy * FR + fx) * PIXB + slot * 16) = pre;
}
template <int LO>
__device__ __forceinline__ void act_store(char* buf, const uint4& pre0, const uint4& pre1, int tid) {
  tid = opaque(tid);
  act_store_one<LO>(buf, pre0, tid);
  if (ActRegion<LO>::ITERS > 1) act_store_one<LO>(buf, pre1, tid + BWD_THREADS);
}

struct BwdLds { char* wts; char* bufG0; char* bufG1; char* bufAct; float* stash_gw; float* stash_x0; float* stash_dz6; const float* w_in; const float* w_out; };

// S0 for one frame pixel: blend + sigmoid + last 1x1 layer backward.
template <typename T>
__device__ __forceinline__ void s0_pixel(const ComposeBwdP& p, const BwdLds& m, int idx, int b, int y0, int x0) {
  const int H = p.H, W = p.W, h2 = H >> 1, w2 = W >> 1;
  const int fy = idx / FR, fx = idx - fy * FR;
  const int gy = y0 - 4 + fy, gx = x0 - 4 + fx;
  const bool inside = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
  float g3[3] = {0.f, 0.f, 0.f}, s3[3] = {0.f, 0.f, 0.f}, f3[3] = {0.f, 0.f, 0.f}, low[3] = {0.f, 0.f, 0.f}, wlv = 0.f;
  if (inside) {
    const long pix = ((long)b * H + gy) * W + gx;
    const float* gp = p.gout + pix * p.ld_gout;
    const float* sp = p.small + (((long)b * h2 + (gy >> 1)) * w2 + (gx >> 1)) * p.ld_small;
    const float* fb = p.fine + (((long)b * H + (gy & ~1)) * W + (gx & ~1)) * p.ld_fine;
    const float* fp0 = p.fine + pix * p.ld_fine;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g3[c] = gp[c]; s3[c] = sp[c]; f3[c] = fp0[c];
      low[c] = 0.25f * (fb[c] + fb[p.ld_fine + c] + fb[(long)W * p.ld_fine + c] + fb[((long)W + 1) * p.ld_fine + c]);
    }
    wlv = Elem<T>::to_f32(reinterpret_cast<const T*>(p.wl)[pix * p.ld_wl]);
  }
  const float w = 1.f / (1.f + __expf(-wlv));
  float dwv = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) dwv += g3[c] * (s3[c] - low[c]);
  // d wl: through the sigmoid and the ReLU of the last 1x1 layer (relu'(0) = 0); rounded where the layer-wise path stores it
  const float dz6 = Elem<T>::to_f32(Elem<T>::from_f32(wlv > 0.f ? dwv * w * (1.f - w) : 0.f));
  uint32_t ow[12];          // (plain words, no pointer into a struct array: that form lived in scratch memory, 48 bytes per lane)
#pragma unroll
  for (int n4 = 0; n4 < 6; ++n4) {
    const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(m.w_out + n4 * 4);
    ow[2 * n4] = pack2<T>(wv[0] * dz6, wv[1] * dz6);
    ow[2 * n4 + 1] = pack2<T>(wv[2] * dz6, wv[3] * dz6);
  }
  uint4* dst = reinterpret_cast<uint4*>(m.bufG0 + idx * PIXB);
  dst[0] = uint4{ow[0], ow[1], ow[2], ow[3]}; dst[1] = uint4{ow[4], ow[5], ow[6], ow[7]}; dst[2] = uint4{ow[8], ow[9], ow[10], ow[11]};
  if (fy >= 4 && fy < 20 && fx >= 4 && fx < 20) {
    const int pi = (fy - 4) * 16 + (fx - 4);
    *reinterpret_cast<float4*>(m.stash_gw + pi * 4) = make_float4(g3[0], g3[1], g3[2], w);
    float* sx = m.stash_x0 + pi * 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sx[c] = Elem<T>::to_f32(Elem<T>::from_f32(s3[c]));
      sx[3 + c] = Elem<T>::to_f32(Elem<T>::from_f32(f3[c]));
    }
    m.stash_dz6[pi] = dz6;
  }
}

// One role's whole tile loop.  WROLE = false: the data-gradient waves (+ S0 and the per-pixel tail S5); WROLE = true: the weight-gradient waves,
// whose accumulator tiles (4 layers x WG_NT taps x WG_NJ output-channel tiles) live in registers until the end of the launch.  Both roles run
// the same barrier sequence.  acc1 / acc6: the two 1x1 layers' weight gradients of the 384 threads W1_FIRST .. W1_FIRST + 383 (channel % 24,
// interior row / 24), reduced by the caller.
template <typename T, bool WROLE>
__device__ __forceinline__ void bwd_role(const ComposeBwdP& p, const BwdLds& m, const int (&koff)[NCHUNK], float (&acc1)[7], float& acc6, float& accb6) {
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
  const int w8 = __builtin_amdgcn_readfirstlane((tid >> 6) - (WROLE ? DG_WAVES : 0));      // wave index inside the role
#if CB_WAVES == 16
  const int mi = w8 & 1, nj = (w8 >> 1) & 1, tap0 = (w8 >> 2) * 5;
#else
  const int mi = w8 & 1, nj = 0, tap0 = (w8 >> 1) * 3;
#endif
  constexpr int NACC = WG_NT * WG_NJ;
  f32x4_t wacc[WROLE ? 4 : 1][NACC];
  if (WROLE) {
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int i = 0; i < NACC; ++i) wacc[l][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int per_img = p.tiles_y * p.tiles_x;
  const int H = p.H, W = p.W, h2 = H >> 1, w2 = W >> 1;
  CPH_DECL(WROLE ? DG_THREADS : 0);
  constexpr int PB = WROLE ? 40 : 16;      // slots of this role's stamps
  for (int tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
    CPH(PB + 0);
    const int b = tile / per_img, rem = tile - b * per_img;
    const int ty = rem / p.tiles_x;
    const int y0 = ty * 16, x0 = (rem - ty * p.tiles_x) * 16;
    uint4 pre0, pre1 = {0u, 0u, 0u, 0u};
    act_load<1>(pre0, pre1, p.act[3], p.ld_act[3], b, y0, x0, H, W, tid);          // relu(r3), needed from S1 on
    // ---------------------------------------------------------------- S0: blend + sigmoid + 1x1 backward on the whole frame
#ifndef CB_EXP_NO_S0
    if (!WROLE) {                                           // (the weight-gradient role's registers are taken by its accumulators)
      s0_pixel<T>(p, m, opaque(tid), b, y0, x0);
      if (tid < FR * FR - DG_THREADS) s0_pixel<T>(p, m, opaque(tid) + DG_THREADS, b, y0, x0);
    }
    CPH(PB + 1);
#endif
    act_store<1>(m.bufAct, pre0, pre1, tid);
    CPH(PB + 2);
    __syncthreads();
    CPH(PB + 3);
    // ---------------------------------------------------------------- S1: layer 4 (input relu(r3), output gradient dA)
    act_load<2>(pre0, pre1, p.act[2], p.ld_act[2], b, y0, x0, H, W, tid);          // a2 for S2
    if (WROLE) { CB_WGRAD((wgrad_stage<T, false, WG_NT, WG_NJ>(wacc[WROLE ? 3 : 0], m.bufAct, m.bufG0, mi, nj, tap0, lane))); }
    else { CB_DGRAD(dgrad_layer<T, 1, 0>(m.bufG0, m.bufG1, nullptr, m.bufAct, m.wts + 3 * WL_BYTES, koff, w8, li, q)); }
    CPH(PB + 4);
    __syncthreads();
    CPH(PB + 5);
    act_store<2>(m.bufAct, pre0, pre1, tid);
    CPH(PB + 6);
    __syncthreads();
    CPH(PB + 7);
    // ---------------------------------------------------------------- S2: layer 3 (input relu(a2), output gradient dc3)
    act_load<3>(pre0, pre1, p.act[1], p.ld_act[1], b, y0, x0, H, W, tid);          // relu(r1) for S3
    if (WROLE) { CB_WGRAD((wgrad_stage<T, true, WG_NT, WG_NJ>(wacc[WROLE ? 2 : 0], m.bufAct, m.bufG1, mi, nj, tap0, lane))); }
    else { CB_DGRAD(dgrad_layer<T, 2, 1>(m.bufG1, m.bufG0, m.bufG0, m.bufAct, m.wts + 2 * WL_BYTES, koff, w8, li, q)); }
    CPH(PB + 8);
    __syncthreads();
    CPH(PB + 9);
    act_store<3>(m.bufAct, pre0, pre1, tid);
    CPH(PB + 10);
    __syncthreads();
    CPH(PB + 11);
    // ---------------------------------------------------------------- S3: layer 2 (input relu(r1), output gradient d a2)
    act_load<3>(pre0, pre1, p.act[0], p.ld_act[0], b, y0, x0, H, W, tid);          // a1 for S4
    if (WROLE) { CB_WGRAD((wgrad_stage<T, false, WG_NT, WG_NJ>(wacc[WROLE ? 1 : 0], m.bufAct, m.bufG0, mi, nj, tap0, lane))); }
    else { CB_DGRAD(dgrad_layer<T, 3, 0>(m.bufG0, m.bufG1, nullptr, m.bufAct, m.wts + WL_BYTES, koff, w8, li, q)); }
    CPH(PB + 12);
    __syncthreads();
    CPH(PB + 13);
    act_store<3>(m.bufAct, pre0, pre1, tid);
    CPH(PB + 14);
    __syncthreads();
    CPH(PB + 15);
    // ---------------------------------------------------------------- S4: layer 1 (input a1, output gradient dc1) -> dz1 on the interior
    act_load<4>(pre0, pre1, p.act[4], p.ld_act[4], b, y0, x0, H, W, tid);          // a3 (interior only) for the last 1x1 layer's weight gradient
    if (WROLE) { CB_WGRAD((wgrad_stage<T, false, WG_NT, WG_NJ>(wacc[0], m.bufAct, m.bufG1, mi, nj, tap0, lane))); }
    else { CB_DGRAD(dgrad_layer<T, 4, 2>(m.bufG1, m.bufG0, m.bufG0, m.bufAct, m.wts, koff, w8, li, q)); }
    CPH(PB + 16);
    __syncthreads();
    CPH(PB + 17);

    // ---------------------------------------------------------------- S5: first 1x1 layer + blend: d fine, d small, dW1, dW6
    act_store<4>(m.bufAct, pre0, pre1, tid);
    __syncthreads();                                          // a3 has landed in the activation buffer; the two parts of S5 run side by side
    CPH(PB + 18);
#ifndef CB_EXP_NO_S5
    const int t5 = opaque(tid);
    const int n24 = (t5 - W1_FIRST) % 24, row0 = (t5 - W1_FIRST) / 24;      // threads W1_FIRST .. +383: channel n24, interior row row0 (0..15)
    if (!WROLE && tid < 256) {                                // threads 0..255: one interior pixel each
      const int bq = t5 >> 2, sub = t5 & 3;                   // a 2x2 block = 4 consecutive lanes
      const int y = 2 * (bq >> 3) + (sub >> 1), x = 2 * (bq & 7) + (sub & 1);
      const int pi = y * 16 + x;
      const T* zp = reinterpret_cast<const T*>(m.bufG0 + ((4 + y) * FR + 4 + x) * PIXB);
      float dx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int n4 = 0; n4 < 6; ++n4) {                        // four channels of dz1 at a time: this role has 128 registers
        float dz[4];
        load4<T>(zp + n4 * 4, dz);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(m.w_in + k * 24 + n4 * 4);
          dx[k] += wv[0] * dz[0] + wv[1] * dz[1] + wv[2] * dz[2] + wv[3] * dz[3];
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) dx[k] = Elem<T>::to_f32(Elem<T>::from_f32(dx[k]));      // the layer-wise path stores d(net input) in the storage type
      const float4 gw = *reinterpret_cast<const float4*>(m.stash_gw + pi * 4);
      const float g3[3] = {gw.x, gw.y, gw.z};
      const int gy = y0 + y, gx = x0 + x;
      const bool inside = gy < H && gx < W;
      float* dsm = p.d_small + (((long)b * h2 + (gy >> 1)) * w2 + (gx >> 1)) * p.ld_dsmall;
      float old[3] = {0.f, 0.f, 0.f};
      if (p.acc_small && inside && sub == 0) { old[0] = dsm[0]; old[1] = dsm[1]; old[2] = dsm[2]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float ts = gw.w * g3[c];                              // sum over the 2x2 block of w * d(out)
        ts += __shfl_xor(ts, 1);
        ts += __shfl_xor(ts, 2);
        float ds = dx[c];
        ds += __shfl_xor(ds, 1);
        ds += __shfl_xor(ds, 2);
        if (inside) {
          p.d_fine[(((long)b * H + gy) * W + gx) * p.ld_dfine + c] = g3[c] - 0.25f * ts + dx[3 + c];
          if (sub == 0) dsm[c] = old[c] + ts + ds;
        }
      }
    }
    if (tid >= W1_FIRST && tid < W1_FIRST + 384) {            // the two 1x1 layers' weight gradients, one (channel, interior row) per thread
      const char* zrow = m.bufG0 + ((4 + row0) * FR + 4) * PIXB;
      const char* arow = m.bufAct + ((4 + row0) * FR + 4) * PIXB;
#pragma unroll 4
      for (int x = 0; x < 16; ++x) {
        const float d = Elem<T>::to_f32(reinterpret_cast<const T*>(zrow + x * PIXB)[n24]);
        const float a3 = Elem<T>::to_f32(reinterpret_cast<const T*>(arow + x * PIXB)[n24]);
        const float* sx = m.stash_x0 + (row0 * 16 + x) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc1[k] += sx[k] * d;         // dW1[k][n] += x0[k] * dz1[n]
        acc1[6] += d;                                             // db1[n]
        const float dz6 = m.stash_dz6[row0 * 16 + x];
        acc6 += a3 * dz6;                                         // dW6[n] += a3[n] * dz6
        accb6 += dz6;                                             // db6 (every n carries the same sum; n == 0 flushes it)
      }
    }
    CPH(PB + 19);
#endif
    __syncthreads();      // frames and stashes are rewritten by the next tile
    CPH(PB + 20);
  }

  if (WROLE) {            // flush: one atomic per weight-gradient element per workgroup
    dd_det_wait();        // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int i = 0; i < WG_NT; ++i)
#pragma unroll
        for (int j = 0; j < WG_NJ; ++j) {
          const int tap = tap0 + i, co = (nj + j) * 16 + li;
          if (co >= 24 || tap > 8) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ci = mi * 16 + q * 4 + e;
            const float v = wacc[WROLE ? l : 0][i * WG_NJ + j][e];
            if (ci < 24) atomicAdd(p.dw_res[l] + (tap * 24 + ci) * 24 + co, v);
            else if (ci == 24 && tap == 4) atomicAdd(p.db_res[l] + co, v);
          }
        }
  }
}

template <typename T>
__global__ __launch_bounds__(BWD_THREADS) void compose_bwd_kernel(const ComposeBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  BwdLds m;
  m.wts = smem;
  m.bufG0 = smem + 4 * WL_BYTES;
  m.bufG1 = m.bufG0 + BUF_BYTES;
  m.bufAct = m.bufG1 + BUF_BYTES;
  m.stash_gw = reinterpret_cast<float*>(m.bufAct + BUF_BYTES);      // interior: d(out) (3) and w
  m.stash_x0 = m.stash_gw + 256 * 4;                                  // interior: the net input (6), rounded to the storage type
  m.stash_dz6 = m.stash_x0 + 256 * 6;                                 // interior: d(relu(1x1(a3)) pre-activation)
  float* fp = m.stash_dz6 + 256;
  m.w_in = fp; m.w_out = fp + 144;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4;
  // ---- parameters: flipped / transposed 3x3 weights as the A operand of the data-gradient convs, the 1x1 weights in fp32 (rounded)
  for (int i = tid; i < 4 * WL_BYTES / 16; i += BWD_THREADS) reinterpret_cast<uint4*>(m.wts)[i] = uint4{0u, 0u, 0u, 0u};
  for (int i = tid; i < NFPB; i += BWD_THREADS) fp[i] = Elem<T>::to_f32(Elem<T>::from_f32(i < 144 ? p.w_in[i] : p.w_out[i - 144]));
  __syncthreads();
  for (int e = tid; e < 4 * 5184; e += BWD_THREADS) {      // d in[ci] = sum_{tap', co} dc[p + tap' - 1][co] K[8 - tap'][ci][co]: row = ci, k-group = tap' * 3 + co / 8
    const int l = e / 5184, rem = e - l * 5184;
    const int tap = rem / 576, ci = (rem / 24) % 24, co = rem % 24;
    const int g = (8 - tap) * 3 + (co >> 3);
    T* dst = reinterpret_cast<T*>(m.wts + l * WL_BYTES + (g >> 2) * 2048 + w_off(ci, g & 3)) + (co & 7);
    *dst = Elem<T>::from_f32(p.w_res[l][rem]);
  }
  int koff[NCHUNK];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    const int g = min(4 * c + q, 26);
    const int tap = g / 3, cg = g - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    koff[c] = (dy * FR + dx) * PIXB + cg * 16;
  }
  __syncthreads();

  float acc1[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc6 = 0.f, accb6 = 0.f;
  // Nothing role-specific lives across the role branch: the weight-gradient role keeps 144 accumulator registers for the whole launch,
  // the data-gradient role its 56 weight-fragment registers per layer -- together they would not fit a wave's 256.
  if (wave < DG_WAVES) bwd_role<T, false>(p, m, koff, acc1, acc6, accb6);
  else bwd_role<T, true>(p, m, koff, acc1, acc6, accb6);

  float* red = reinterpret_cast<float*>(smem);               // [row pair][n][9]: the weight images are no longer needed
  __syncthreads();
  if (tid >= W1_FIRST && tid < W1_FIRST + 384) {
#pragma unroll
    for (int k = 0; k < 7; ++k) red[(tid - W1_FIRST) * 9 + k] = acc1[k];
    red[(tid - W1_FIRST) * 9 + 7] = acc6;
    red[(tid - W1_FIRST) * 9 + 8] = accb6;
  }
  __syncthreads();
  dd_det_wait();
  if (tid < 24 * 9) {
    const int n = tid / 9, k = tid - n * 9;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += red[(r * 24 + n) * 9 + k];
    if (k < 6) atomicAdd(p.dw_in + k * 24 + n, s);
    else if (k == 6) atomicAdd(p.db_in + n, s);
    else if (k == 7) atomicAdd(p.dw_out + n, s);
    else if (n == 0) atomicAdd(p.db_out, s);
  }
  dd_det_end();
}

}  // namespace

int dd_compose_stream_fwd_launch(const dd_compose_args* a, hipStream_t s);      // csrc/dd_compose_stream.hip
int dd_compose_stream_bwd_data_launch(const dd_compose_bwd_args* a, void* scratch, hipStream_t s);      // csrc/dd_compose_stream_bwd.hip
int dd_compose_stream_wgrad_launch(const dd_compose_bwd_args* a, void* scratch, hipStream_t s);
extern "C" long dd_compose_bwd_scratch_bytes(int N, int H, int W);

extern "C" int dd_compose_net_fwd(const dd_compose_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->small && a->fine && a->out && a->w_in && a->b_in && a->w_out && a->b_out, "dd_compose_net_fwd: null pointer");
  for (int l = 0; l < 4; ++l) DD_REQUIRE(a->w_res[l] && a->b_res[l], "dd_compose_net_fwd: null residual-block weights");
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_compose_net_fwd: storage dtype must be DD_BF16 or DD_F16 (f32 runs layer by layer)");
  DD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->H % 2 == 0 && a->W % 2 == 0, "dd_compose_net_fwd: H=%d W=%d must be positive and even", a->H, a->W);
  DD_REQUIRE(a->ld_small >= 3 && a->ld_fine >= 3 && a->ld_out >= 3, "dd_compose_net_fwd: ld < 3");
  for (int i = 0; i < 5; ++i)
    DD_REQUIRE(!a->save_act[i] || (a->ld_act[i] >= 24 && a->ld_act[i] % 8 == 0 && ((uintptr_t)a->save_act[i] % 16) == 0),
               "dd_compose_net_fwd: saved activation %d needs ld >= 24, ld %% 8 == 0 and 16-byte alignment", i);
  DD_REQUIRE(!a->save_netin || (a->ld_netin >= 8 && a->ld_netin % 8 == 0 && ((uintptr_t)a->save_netin % 16) == 0), "dd_compose_net_fwd: bad save_netin");
  DD_REQUIRE(!a->save_wl || a->ld_wl >= 1, "dd_compose_net_fwd: bad save_wl");
  {
    // round 4: the row-streaming kernel (csrc/dd_compose_stream.hip) is the forward; DD_COMPOSE_STREAM=0 keeps the 16x16-tile kernel below
    static const bool stream_fwd = !(getenv("DD_COMPOSE_STREAM") && getenv("DD_COMPOSE_STREAM")[0] == '0');
    // (the streaming kernel has 32-bit pixel offsets: launches of 2^25 pixels or more fall through to the tile kernel below, whose offsets are
    //  64-bit -- the same condition dd_compose_net_bwd applies, so the two directions agree)
    if (stream_fwd && (long)a->N * a->H * a->W < (1l << 31) / 64) return dd_compose_stream_fwd_launch(a, reinterpret_cast<hipStream_t>(stream));
  }
  ComposeP p;
  p.small = a->small; p.fine = a->fine; p.out = a->out;
  p.w_in = a->w_in; p.b_in = a->b_in; p.w_out = a->w_out; p.b_out = a->b_out;
  for (int l = 0; l < 4; ++l) { p.w_res[l] = a->w_res[l]; p.b_res[l] = a->b_res[l]; }
  p.save_netin = a->save_netin; p.save_wl = a->save_wl;
  for (int i = 0; i < 5; ++i) { p.save_act[i] = a->save_act[i]; p.ld_act[i] = a->ld_act[i]; }
  p.ld_small = a->ld_small; p.ld_fine = a->ld_fine; p.ld_out = a->ld_out; p.ld_netin = a->ld_netin; p.ld_wl = a->ld_wl;
  p.N = a->N; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, 16); p.tiles_y = dd_ceil_div(a->H, 16);
  p.total_tiles = a->N * p.tiles_x * p.tiles_y;
  const int cus = dd_device_cus();
  const int grid = p.total_tiles < cus ? p.total_tiles : cus;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == DD_BF16) {
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_fwd_kernel<bf16_t>));
    hipLaunchKernelGGL(compose_fwd_kernel<bf16_t>, dim3(grid), dim3(512), LDS_TOTAL, s, p);
  } else {
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_fwd_kernel<f16_t>));
    hipLaunchKernelGGL(compose_fwd_kernel<f16_t>, dim3(grid), dim3(512), LDS_TOTAL, s, p);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}

extern "C" int dd_compose_net_bwd(const dd_compose_bwd_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->small && a->fine && a->dout && a->wl && a->d_small && a->d_fine, "dd_compose_net_bwd: null pointer");
  DD_REQUIRE(a->w_in && a->w_out && a->dw_in && a->db_in && a->dw_out && a->db_out, "dd_compose_net_bwd: null 1x1 weights / gradients");
  for (int l = 0; l < 4; ++l) DD_REQUIRE(a->w_res[l] && a->dw_res[l] && a->db_res[l], "dd_compose_net_bwd: null residual-block weights / gradients");
  for (int i = 0; i < 5; ++i)
    DD_REQUIRE(a->act[i] && a->ld_act[i] >= 24 && a->ld_act[i] % 8 == 0 && ((uintptr_t)a->act[i] % 16) == 0,
               "dd_compose_net_bwd: activation %d needs ld >= 24, ld %% 8 == 0 and 16-byte alignment", i);
  DD_REQUIRE(a->dtype == DD_BF16 || a->dtype == DD_F16, "dd_compose_net_bwd: storage dtype must be DD_BF16 or DD_F16");
  DD_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->H % 2 == 0 && a->W % 2 == 0, "dd_compose_net_bwd: H=%d W=%d must be positive and even", a->H, a->W);
  DD_REQUIRE(a->ld_small >= 3 && a->ld_fine >= 3 && a->ld_dout >= 3 && a->ld_dsmall >= 3 && a->ld_dfine >= 3 && a->ld_wl >= 1, "dd_compose_net_bwd: bad ld");
  if (a->scratch) {
    // round 4: two row-streaming launches (csrc/dd_compose_stream_bwd.hip); DD_COMPOSE_STREAM_BWD=0 keeps the 16x16-tile kernel below
    static const bool stream_bwd = !(getenv("DD_COMPOSE_STREAM_BWD") && getenv("DD_COMPOSE_STREAM_BWD")[0] == '0');
    bool rows24 = true;
    for (int i = 0; i < 5; ++i) rows24 = rows24 && a->ld_act[i] == 24;
    if (stream_bwd && rows24 && (long)a->N * a->H * a->W < (1l << 31) / 64) {
      DD_REQUIRE(a->scratch_bytes >= dd_compose_bwd_scratch_bytes(a->N, a->H, a->W) && ((uintptr_t)a->scratch % 16) == 0,
                 "dd_compose_net_bwd: scratch of %ld bytes, %ld needed (16-byte aligned)", a->scratch_bytes, dd_compose_bwd_scratch_bytes(a->N, a->H, a->W));
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
      const int rc = dd_compose_stream_bwd_data_launch(a, a->scratch, st);
      if (rc != DD_OK) return rc;
      return dd_compose_stream_wgrad_launch(a, a->scratch, st);
    }
  }
  ComposeBwdP p;
  p.small = a->small; p.fine = a->fine; p.gout = a->dout; p.wl = a->wl;
  for (int i = 0; i < 5; ++i) { p.act[i] = a->act[i]; p.ld_act[i] = a->ld_act[i]; }
  p.w_in = a->w_in; p.w_out = a->w_out; p.dw_in = a->dw_in; p.db_in = a->db_in; p.dw_out = a->dw_out; p.db_out = a->db_out;
  for (int l = 0; l < 4; ++l) { p.w_res[l] = a->w_res[l]; p.dw_res[l] = a->dw_res[l]; p.db_res[l] = a->db_res[l]; }
  p.d_small = a->d_small; p.d_fine = a->d_fine;
  p.ld_small = a->ld_small; p.ld_fine = a->ld_fine; p.ld_gout = a->ld_dout; p.ld_wl = a->ld_wl; p.ld_dsmall = a->ld_dsmall; p.ld_dfine = a->ld_dfine;
  p.acc_small = a->accumulate_small;
  p.N = a->N; p.H = a->H; p.W = a->W;
  p.tiles_x = dd_ceil_div(a->W, 16); p.tiles_y = dd_ceil_div(a->H, 16);
  p.total_tiles = a->N * p.tiles_x * p.tiles_y;
  const int cus = dd_device_cus();
  const int grid = p.total_tiles < cus ? p.total_tiles : cus;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == DD_BF16) {
    dd_det_sync();
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_bwd_kernel<bf16_t>));
    hipLaunchKernelGGL(compose_bwd_kernel<bf16_t>, dim3(grid), dim3(BWD_THREADS), LDS_BWD, s, p);
  } else {
    dd_det_sync();
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_bwd_kernel<f16_t>));
    hipLaunchKernelGGL(compose_bwd_kernel<f16_t>, dim3(grid), dim3(BWD_THREADS), LDS_BWD, s, p);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}
