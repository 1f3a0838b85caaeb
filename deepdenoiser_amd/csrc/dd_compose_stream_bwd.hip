// Row-streaming backward of the compose net (round 4), data path: TF autodiff of MultiScalePrediction.py:36-93 behind Training.py:701-702.
//
// The gradient chain of the net has the SAME shape as its forward (csrc/dd_compose_stream.hip): a pointwise stage, four 3x3 convolutions with a
// residual link across each pair, a pointwise tail --
//     stage 0  dz6 = d(out)/d(wl) through blend, sigmoid and ReLU;  dA = d a3 = w_out dz6                         row  s
//     stage 1  dc3 = (K4^T * dA) . [r3 > 0]                                                                       row  s - 2
//     stage 2  d a2 = dA + (K3^T * dc3) . [a2 > 0]                                                                row  s - 4
//     stage 3  dc1 = (K2^T * d a2) . [r1 > 0]                                                                     row  s - 6
//     stage 4  dz1 = (d a2 + K1^T * dc1) . [a1 > 0];  d x0 = W1^T dz1;  d fine, d small (2x2 sums)                 row  s - 8
// -- so this kernel is the forward's pipeline with other epilogues: 16 waves, one gradient convolution per group of four waves, rolling windows
// of the gradient tensors in LDS, the flipped / transposed weights as register-resident MFMA A operands, one barrier per step, no halo
// recompute in y.  The ReLU masks are the activations the forward stored (read from global memory by the lane that owns the pixel: the MFMA C
// layout gives a lane 4 consecutive channels of one pixel = 8 contiguous bytes).  Every intermediate is rounded to the storage type where the
// layer-wise backward stores it, with ONE rounding per element (mask and residual are applied in fp32).
// The gradient tensors the WEIGHT gradients need (dz1, dc1, d a2, dc3: 24 channels each, and dz6) are written to caller-provided scratch; the
// weight gradients are a second launch (wgrad kernel below) that streams activations and gradients once.  The 16x16-tile kernel of
// dd_compose.hip did both in one launch on a 24x24 frame (2.25x recompute, transposing LDS reads as its bound: 668 us at 128^2 x 128).
//
// d fine / d small need the 2x2 sums of a block of two rows: stage 4 parks (w g, d small part, g + d fine part) of its row in a small LDS ring
// and the wave that wrote an ODD row combines the pair one step later (behind the barrier).
#include "dd_compose_stream.h"

namespace {

constexpr int KS = 14;
constexpr int KS_MAX = 16;
constexpr int TAILB = 40;          // bytes of a pixel's record in the tail ring: w g (3), d small part (3), g + d fine part (3), pad
constexpr int AUX_BYTES = 2 * 1024 + 128;      // behind the rings: the input layer's two A fragments (dx0 = W1^T dz1), w_out (rounded, fp32)

struct CbP {
  const float* small; const float* fine; const float* gout;
  const void* act[4];               // a1, relu(r1), a2, relu(r3): the ReLU masks of stages 4, 3, 2, 1
  const void* wl;
  const float* w_in; const float* w_res[4]; const float* w_out;
  float* d_small; float* d_fine;
  void* g[5];                       // scratch [pixel][24]: dz1, dc1, d a2, dc3, dA
  void* x0;                         // scratch [pixel][8]: the packed net input [up(small) | fine | 1 | 0]
  void* dz6;                        // scratch [pixel]
  int ld_small, ld_fine, ld_gout, ld_act[4], ld_wl, ld_dsmall, ld_dfine, acc_small;
  int N, H, W;
  int FW, R, TPR, n_strips, SO, BH, nb, VB, units;
};

// A operands of the four data-gradient convolutions (over the ring area, before it is zeroed): image [stage 1..4][K-step c][lane] of 16 bytes;
// lane = (m = lane % 32 = INPUT channel ci of the forward conv, h = lane / 32), k-group g = 2c + h < 27: tap' = g / 3, output channels
// 8 (g % 3) .. + 7:  d in[ci] = sum_{tap', co} dy[p + tap' - 1][co] K[8 - tap'][ci][co]  (HWIO [tap][ci][co], rounded to the storage type).
// Stage k differentiates conv2d_(5 - k).  Behind the rings: W1^T as two A fragments (row m < 6 = net-input channel; K order = the C layout of
// the dz1 accumulators, as the forward's output layer) and w_out rounded, in fp32.
template <typename T>
__device__ __forceinline__ void stage_weights_bwd(char* img, char* aux, const CbP& p, int tid) {
  for (int i = tid; i < 4 * KS_MAX * 64 * 8; i += 1024) {
    const int e = i & 7, ln = (i >> 3) & 63, c = (i >> 9) & (KS_MAX - 1), l = i >> 13;
    const int m = ln & 31, g = 2 * c + (ln >> 5);
    float v = 0.f;
    if (m < 24 && g < 27) v = p.w_res[3 - l][((8 - g / 3) * 24 + m) * 24 + 8 * (g % 3) + e];
    reinterpret_cast<T*>(img)[i] = Elem<T>::from_f32(v);
  }
  T* ax = reinterpret_cast<T*>(aux);
  for (int i = tid; i < 2 * 512; i += 1024) {
    const int e = i & 7, ln = (i >> 3) & 63, which = i >> 9;
    const int m = ln & 31, h = ln >> 5;
    float v = 0.f;
    if (m < 6) {
      const int ch = which == 0 ? (e < 4 ? 4 * h + e : 8 + 4 * h + e - 4) : (e < 4 ? 16 + 4 * h + e : -1);
      if (ch >= 0) v = p.w_in[m * 24 + ch];
    }
    ax[i] = Elem<T>::from_f32(v);
  }
  float* wo = reinterpret_cast<float*>(aux + 2048);
  for (int i = tid; i < 32; i += 1024) wo[i] = i < 24 ? Elem<T>::to_f32(Elem<T>::from_f32(p.w_out[i])) : 0.f;
}

// keep the lanes of a packed pair whose mask value is > 0 (dd_common.h mask_bf16x2: the same test for bf16 and fp16)
__device__ __forceinline__ uint2 mask2(uint2 v, uint2 m) { return uint2{mask_bf16x2(v.x, m.x), mask_bf16x2(v.y, m.y)}; }

// One role = one gradient convolution (LAYER 0..3 = stages 1..4); LAYER 0 also runs stage 0, LAYER 3 the tail.
template <typename T, int LAYER, int NPRE>
__device__ __forceinline__ void cb_role(const CbP& p, char* smem, char* aux, char* tail, int ring_bytes, int t, int lane, int u0, int nunits, int steps) {
  constexpr bool RES_AFTER = LAYER == 1;                      // d a2 = dA + conv . mask      (residual added behind the mask)
  constexpr bool RES_INIT = LAYER == 3;                       // dz1 = (d a2 + conv) . mask   (the accumulator starts from the residual)
  constexpr bool FINAL = LAYER == 3;
  const int R = p.R, TPR = p.TPR, FW = p.FW, VB = p.VB;
  const int n = lane & 31, h = lane >> 5;
  const int r = t / TPR, xc = t - r * TPR;
  const bool has_task = t < R * TPR;
  const int x = xc * 32 + n;                                 // frame-local column of this lane's pixel
  const int pitch = (FW + 2) * PIXB, rp = R * pitch;
  // rings: 0 dA, 1 dc3, 2 d a2, 3 dc1
  const int D0 = 3 * R + 2, D1 = 2 * R + 2;
  const int off1 = D0 * pitch, off2 = off1 + D1 * pitch, off3 = off2 + D0 * pitch;
  const int Din = (LAYER == 0 || LAYER == 2) ? D0 : D1;
  const int Dout = (LAYER == 1) ? D0 : D1;
  const int in_bytes = Din * pitch, out_bytes = Dout * pitch, res_bytes = D0 * pitch;
  char* rin = smem + (LAYER == 0 ? 0 : LAYER == 1 ? off1 : LAYER == 2 ? off2 : off3);
  char* rout = smem + (LAYER == 0 ? off1 : LAYER == 1 ? off2 : off3);       // (LAYER 3 writes no ring)
  char* rres = smem + (LAYER == 1 ? 0 : off2);
  const int DT = 2 * R + 2, tail_pitch = FW * TAILB, tail_bytes = DT * tail_pitch;

  uint4 wa[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) wa[c] = *reinterpret_cast<const uint4*>(smem + ((LAYER * KS_MAX + c) * 64 + lane) * 16);
  __syncthreads();                                           // every wave holds its fragments: the image may be overwritten
  for (int i = threadIdx.x; i < ring_bytes / 16; i += 1024) reinterpret_cast<uint4*>(smem)[i] = uint4{0u, 0u, 0u, 0u};

  const int H = p.H, W = p.W, h2 = H >> 1, w2 = W >> 1;
  const int lag = (LAYER + 1) * (R + 1);
  const int total_rows = nunits * VB;
  const int s_beg = has_task ? (lag - r + R - 1) / R : steps, s_end = has_task ? (total_rows + lag - r + R - 1) / R : steps;
  Cursor cur;
  cur.init(p, u0, nunits, r - lag);
  int o_in = posmod(r - lag, Din) * pitch, o_out = posmod(r - lag, Dout) * pitch, o_res = posmod(r - lag, D0) * pitch;
  int o_tail = posmod(r - lag, DT) * tail_pitch;
  auto wrap_add = [](int o, int add, int bytes) { o += add; return o >= bytes ? o - bytes : o; };
  const int xo = x * PIXB + h * 16;
  const int a4c = h ? x * PIXB : x * PIXB + 128;             // K-step 4: group 8 (row y - 1, k-half 0) | group 9 (row y, k-half 1)
  const int a13c = x * PIXB + 128;                           // K-step 13: group 26 (row y + 1); the k-half-1 lanes (zero weights) read the same finite data
  const int wr = (x + 1) * PIXB + h * 8;                     // where this lane's 4-channel groups go in a ring row
  int ybase = 0, pixb = 0;
  bool col_in = false, col_own = false, cols_all_in = false;
  auto unit_values = [&](const Cursor& c, int& yb, int& pb, bool& ci, bool& co, bool& call) {
    const int gx = c.U.fx0 + x;
    yb = c.U.yb0 - 4;
    ci = (unsigned)gx < (unsigned)W;
    co = ci && gx >= c.U.xs && gx < c.U.xe;
    pb = c.U.b * H * W + (ci ? gx : 0);
    call = c.U.fx0 + xc * 32 >= 0 && c.U.fx0 + xc * 32 + 32 <= W;
  };
  unit_values(cur, ybase, pixb, col_in, col_own, cols_all_in);

  // stage 0 (LAYER 0 waves): its own cursor; operands requested one step ahead
  Cursor c0;
  int o0 = 0, ybase0 = 0, pixb0 = 0, pix0 = 0;
  bool col_in0 = false, col_own0 = false, cols_all_in0 = false, in0 = false, own0 = false, odd0 = false;
  float g0[3] = {0.f, 0.f, 0.f}, sm0[3] = {0.f, 0.f, 0.f}, f0[3] = {0.f, 0.f, 0.f}, wl0 = 0.f;
  const int s0_end = has_task ? (total_rows - r + R - 1) / R : 0;
  auto load_s0 = [&](int s_next) {
    in0 = false; own0 = false;
    if (s_next < s0_end) {
      const int y = ybase0 + c0.i;
      const bool row_in = (unsigned)y < (unsigned)H;
      in0 = row_in && col_in0;
      own0 = in0 && col_own0 && y >= c0.U.yb0 && y < c0.U.yb1;
      const int cy = row_in ? y : 0;
      odd0 = (cy & 1) != 0;
      pix0 = pixb0 + cy * W;
      const int cx = pix0 - (c0.U.b * H + cy) * W;
      const float* gp = p.gout + (size_t)pix0 * p.ld_gout;
      const float* sp = p.small + (size_t)((c0.U.b * h2 + (cy >> 1)) * w2 + (cx >> 1)) * p.ld_small;
      const float* fr = p.fine + (size_t)(pix0 + (h - (cy & 1)) * W) * p.ld_fine;      // row (y & ~1) + h of the 2x2 block
#pragma unroll
      for (int c = 0; c < 3; ++c) { g0[c] = gp[c]; sm0[c] = sp[c]; f0[c] = fr[c]; }
      wl0 = Elem<T>::to_f32(reinterpret_cast<const T*>(p.wl)[(size_t)pix0 * p.ld_wl]);
    }
  };
  if (LAYER == 0) {
    c0.init(p, u0, nunits, r);
    o0 = posmod(r, D0) * pitch;
    unit_values(c0, ybase0, pixb0, col_in0, col_own0, cols_all_in0);
    load_s0(0);
  }

  // the task of a step, split around the barrier exactly as in the forward (dd_compose_stream.hip): K-steps 0 .. NPRE - 1 read rows that are at
  // least two steps old and run before the barrier, behind the previous task's epilogue
  f32x16_t acc;
  auto conv_part_n = [&](auto pre_t) {
    constexpr bool pre = decltype(pre_t)::value;
    constexpr int C0 = pre ? 0 : NPRE, C1 = pre ? NPRE : KS;
    const int o_m1 = o_in == 0 ? in_bytes - pitch : o_in - pitch;
    const int o_p1 = o_in + pitch == in_bytes ? 0 : o_in + pitch;
    const char* q0 = rin + xo + o_m1;
    const char* q1 = rin + xo + 16 + o_in;
    const char* q2 = rin + xo + o_p1;
    const char* q4 = rin + a4c + (h ? o_in : o_m1);
    const char* q13 = rin + a13c + o_p1;
    f32x16_t c_init = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (RES_INIT && pre) {
      const char* qr = rres + o_res + wr;
      uint2 rv[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) rv[b] = *reinterpret_cast<const uint2*>(qr + b * 16);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float r0, r1, r2, r3;
        unpack2<T>(rv[b].x, r0, r1); unpack2<T>(rv[b].y, r2, r3);
        c_init[4 * b] = r0; c_init[4 * b + 1] = r1; c_init[4 * b + 2] = r2; c_init[4 * b + 3] = r3;
      }
    }
    constexpr int BATCH = 6;
#pragma unroll
    for (int c0_ = C0; c0_ < C1; c0_ += BATCH) {
      uint4 bf[BATCH];
#pragma unroll
      for (int c = c0_; c < c0_ + BATCH && c < C1; ++c) {
        const char* a = c < 4 ? q0 + 32 * c : c == 4 ? q4 : c < 9 ? q1 + 32 * (c - 5) : c < 13 ? q2 + 32 * (c - 9) : q13;
        bf[c - c0_] = *reinterpret_cast<const uint4*>(a);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = c0_; c < c0_ + BATCH && c < C1; ++c) acc = mma32<T>(wa[c], bf[c - c0_], (pre && c == 0) ? c_init : acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (pre && NPRE == 0) acc = c_init;
  };

  bool act = 0 >= s_beg && 0 < s_end;
  if (act) conv_part_n(std::true_type());
  int prev_y = -1, prev_pixb = 0, prev_b = 0;                // tail: the row this wave parked in the previous step (odd rows complete a block)
  bool prev_own_row = false, prev_col_own = false;

  for (int s = 0; s < steps; ++s) {
    __syncthreads();
    // ------------------------------------------------------------------------------------------ stage 0: dz6 and dA = w_out dz6
    if (LAYER == 0) {
      if (s < s0_end) {
        float low[3], f_own[3], dwv = 0.f;
        const bool odd_row = odd0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float pair = f0[c] + dpp_xor1(f0[c]);
          float p_lo, p_hi, f_lo, f_hi;
          both_halves(pair, p_lo, p_hi);
          both_halves(f0[c], f_lo, f_hi);
          f_own[c] = odd_row ? f_hi : f_lo;                   // k-half h fetched row (y & ~1) + h of the block
          low[c] = 0.25f * (p_lo + p_hi);
          dwv += g0[c] * (sm0[c] - low[c]);
        }
        const float w = 1.f / (1.f + __expf(-wl0));
        // d wl: through the sigmoid and the ReLU of the last 1x1 layer (relu'(0) = 0); rounded where the layer-wise path stores it
        float dz6 = (in0 && wl0 > 0.f) ? dwv * w * (1.f - w) : 0.f;
        const uint32_t dzp = pack2<T>(dz6, 0.f);
        float unused;
        unpack2<T>(dzp, dz6, unused);
        if (own0 && h == 0) {
          reinterpret_cast<uint16_t*>(p.dz6)[pix0] = (uint16_t)(dzp & 0xffffu);
          // the packed net input of the pixel, as the layer-wise path stores it (+ a ones channel: the bias row of dW1), for the weight gradients
          float fo[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) fo[c] = f_own[c];
          reinterpret_cast<uint4*>(p.x0)[pix0] = uint4{pack2<T>(sm0[0], sm0[1]), pack2<T>(sm0[2], fo[0]), pack2<T>(fo[1], fo[2]), One<T>::v};
        }
        const float* wo = reinterpret_cast<const float*>(aux + 2048);
        char* o = rin + o0 + wr;
        char* gd = reinterpret_cast<char*>(p.g[4]) + ((size_t)pix0 * 24 + 4 * h) * 2;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(wo + 8 * b + 4 * h);
          uint2 pk;
          pk.x = pack2<T>(wv[0] * dz6, wv[1] * dz6); pk.y = pack2<T>(wv[2] * dz6, wv[3] * dz6);
          *reinterpret_cast<uint2*>(o + b * 16) = pk;
          if (own0) *reinterpret_cast<uint2*>(gd + b * 16) = pk;
        }
      }
      c0.i += R;
      if (c0.i >= VB) {
        c0.i -= R;
        c0.advance(p, u0, nunits, R);
        unit_values(c0, ybase0, pixb0, col_in0, col_own0, cols_all_in0);
      }
      o0 = wrap_add(o0, rp, D0 * pitch);
      load_s0(s + 1);
    }
    // ------------------------------------------------------------------------------------------ tail, second half: the 2x2 sums of a block
    if (FINAL && prev_y >= 0 && (prev_y & 1)) {
      // this wave parked row prev_y (odd) in the previous step; row prev_y - 1 was parked by then as well.  k-half h takes row prev_y - 1 + h.
      int o_odd = o_tail - R * tail_pitch; if (o_odd < 0) o_odd += tail_bytes;          // where prev_y went
      int o_even = o_odd - tail_pitch; if (o_even < 0) o_even += tail_bytes;
      const float* rec = reinterpret_cast<const float*>(tail + (h ? o_odd : o_even) + x * TAILB);
      float u[3], ds[3], q[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { u[c] = rec[c]; ds[c] = rec[3 + c]; q[c] = rec[6 + c]; }
      const int yrow = prev_y - 1 + h;
      const int pix = prev_pixb + yrow * W;
      const bool own = prev_own_row && prev_col_own;
      float* dsm = p.d_small + (size_t)((prev_b * h2 + (prev_y >> 1)) * w2 + ((pix - (prev_b * H + yrow) * W) >> 1)) * p.ld_dsmall;
      float old[3] = {0.f, 0.f, 0.f};
      const bool writer = own && h == 0 && !(n & 1);
      if (p.acc_small && writer) { old[0] = dsm[0]; old[1] = dsm[1]; old[2] = dsm[2]; }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a_lo, a_hi, b_lo, b_hi;
        both_halves(u[c] + dpp_xor1(u[c]), a_lo, a_hi);
        both_halves(ds[c] + dpp_xor1(ds[c]), b_lo, b_hi);
        const float ts = a_lo + a_hi, dsum = b_lo + b_hi;
        if (own) p.d_fine[(size_t)pix * p.ld_dfine + c] = q[c] - 0.25f * ts;
        if (writer) dsm[c] = old[c] + ts + dsum;
      }
    }
    // ------------------------------------------------------------------------------------------ this wave's gradient convolution, one 32-pixel task
    int parked_y = -1;
    if (act) {
      const int y = ybase + cur.i;
      const bool row_in = (unsigned)y < (unsigned)H;
      const bool all_in = row_in && cols_all_in;
      const bool inside = row_in && col_in;
      const bool own_row = y >= cur.U.yb0 && y < cur.U.yb1;
      const bool own = inside && col_own && own_row;
      const int cy = row_in ? y : 0;
      const int pix = pixb + cy * W;
      // the ReLU mask of this stage: the forward's stored activation of the pixel, requested before the MFMAs
      constexpr int MASK_ACT = 3 - LAYER;
      const char* mp = reinterpret_cast<const char*>(p.act[MASK_ACT]) + ((size_t)pix * p.ld_act[MASK_ACT] + 4 * h) * 2;
      uint2 mk[3];
#pragma unroll
#ifdef CB_EXP_NO_MASK_LOADS
      for (int b = 0; b < 3; ++b) mk[b] = uint2{0x3f803f80u, (unsigned)pix};
      (void)mp;
#else
      for (int b = 0; b < 3; ++b) mk[b] = *reinterpret_cast<const uint2*>(mp + b * 16);
#endif
      float gt[3] = {0.f, 0.f, 0.f}, wlt = 0.f;
      if (FINAL) {
        const float* gp = p.gout + (size_t)pix * p.ld_gout;
#pragma unroll
        for (int c = 0; c < 3; ++c) gt[c] = gp[c];
        wlt = Elem<T>::to_f32(reinterpret_cast<const T*>(p.wl)[(size_t)pix * p.ld_wl]);
      }
      uint2 rv[3] = {uint2{0u, 0u}, uint2{0u, 0u}, uint2{0u, 0u}};
      if (RES_AFTER) {
        const char* qr = rres + o_res + wr;
#pragma unroll
        for (int b = 0; b < 3; ++b) rv[b] = *reinterpret_cast<const uint2*>(qr + b * 16);
      }
      conv_part_n(std::false_type());
      uint2 pk[3];
      if (RES_AFTER) {
        // out = residual + (mask > 0 ? conv : 0), one rounding
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          float m0, m1, m2, m3, r0, r1, r2, r3;
          unpack2<T>(mk[b].x, m0, m1); unpack2<T>(mk[b].y, m2, m3);
          unpack2<T>(rv[b].x, r0, r1); unpack2<T>(rv[b].y, r2, r3);
          pk[b].x = packo<T>(r0 + (m0 > 0.f ? acc[4 * b] : 0.f), r1 + (m1 > 0.f ? acc[4 * b + 1] : 0.f));
          pk[b].y = packo<T>(r2 + (m2 > 0.f ? acc[4 * b + 2] : 0.f), r3 + (m3 > 0.f ? acc[4 * b + 3] : 0.f));
        }
      } else {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          pk[b].x = packo<T>(acc[4 * b], acc[4 * b + 1]);
          pk[b].y = packo<T>(acc[4 * b + 2], acc[4 * b + 3]);
          pk[b] = mask2(pk[b], mk[b]);
        }
      }
      if (!all_in) {                                         // nothing exists outside the image: no gradient either
#pragma unroll
        for (int b = 0; b < 3; ++b) { pk[b].x = inside ? pk[b].x : 0u; pk[b].y = inside ? pk[b].y : 0u; }
      }
#ifdef CB_EXP_NO_G_STORES
      if (own && pk[0].x == 0x12345u) {
#else
      if (own) {                                             // for the weight-gradient launch
#endif
        char* gp = reinterpret_cast<char*>(p.g[3 - LAYER]) + ((size_t)pix * 24 + 4 * h) * 2;
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<uint2*>(gp + b * 16) = pk[b];
      }
      if (!FINAL) {
        char* o = rout + o_out + wr;
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<uint2*>(o + b * 16) = pk[b];
      } else {
        // d x0 = W1^T dz1 on the matrix pipe (the packed dz1 registers are B fragments; rows 0..3 of the result sit in the k-half-0 lanes,
        // rows 4, 5 in the k-half-1 lanes), rounded where the layer-wise path stores d(net input); then this row's share of the 2x2 sums
        const uint4 wi1 = *reinterpret_cast<const uint4*>(aux + lane * 16);
        const uint4 wi2 = *reinterpret_cast<const uint4*>(aux + 1024 + lane * 16);
        const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16_t d = mma32<T>(wi1, uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y}, z);
        d = mma32<T>(wi2, uint4{pk[2].x, pk[2].y, 0u, 0u}, d);
        float dx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dx[e] = Elem<T>::to_f32(Elem<T>::from_f32(d[e]));
        const float w = 1.f / (1.f + __expf(-wlt));
        float* rec = reinterpret_cast<float*>(tail + o_tail + x * TAILB);
        const bool live = inside;                            // (pixels outside the image contribute nothing to a block)
        if (h == 0) {
          // net-input channels 0..2 = up(small), 3 = fine[0]
#pragma unroll
          for (int c = 0; c < 3; ++c) { rec[c] = live ? w * gt[c] : 0.f; rec[3 + c] = live ? dx[c] : 0.f; }
          rec[6] = gt[0] + dx[3];
        } else {
          rec[7] = gt[1] + dx[0];
          rec[8] = gt[2] + dx[1];
        }
        parked_y = y;
        prev_own_row = own_row; prev_col_own = col_own && col_in; prev_pixb = pixb; prev_b = cur.U.b;
      }
    }
    prev_y = parked_y;
    // ------------------------------------------------------------------------------------------ next step's rows
    cur.i += R;
    if (cur.i >= VB) {
      cur.i -= R;
      cur.advance(p, u0, nunits, R);
      unit_values(cur, ybase, pixb, col_in, col_own, cols_all_in);
    }
    o_in = wrap_add(o_in, rp, in_bytes);
    if (!FINAL) o_out = wrap_add(o_out, rp, out_bytes);
    if (RES_AFTER || RES_INIT) o_res = wrap_add(o_res, rp, res_bytes);
    if (FINAL) o_tail = wrap_add(o_tail, R * tail_pitch, tail_bytes);
    act = s + 1 >= s_beg && s + 1 < s_end;
    if (act) conv_part_n(std::true_type());
  }
  // the last parked row of this wave may still complete a block
  if (FINAL) {
    __syncthreads();
    if (prev_y >= 0 && (prev_y & 1)) {
      int o_odd = o_tail - R * tail_pitch; if (o_odd < 0) o_odd += tail_bytes;
      int o_even = o_odd - tail_pitch; if (o_even < 0) o_even += tail_bytes;
      const float* rec = reinterpret_cast<const float*>(tail + (h ? o_odd : o_even) + x * TAILB);
      const int yrow = prev_y - 1 + h;
      const int pix = prev_pixb + yrow * W;
      const bool own = prev_own_row && prev_col_own;
      float* dsm = p.d_small + (size_t)((prev_b * h2 + (prev_y >> 1)) * w2 + ((pix - (prev_b * H + yrow) * W) >> 1)) * p.ld_dsmall;
      const bool writer = own && h == 0 && !(n & 1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float a_lo, a_hi, b_lo, b_hi;
        both_halves(rec[c] + dpp_xor1(rec[c]), a_lo, a_hi);
        both_halves(rec[3 + c] + dpp_xor1(rec[3 + c]), b_lo, b_hi);
        const float ts = a_lo + a_hi, dsum = b_lo + b_hi;
        if (own) p.d_fine[(size_t)pix * p.ld_dfine + c] = rec[6 + c] - 0.25f * ts;
        if (writer) dsm[c] = (p.acc_small ? dsm[c] : 0.f) + ts + dsum;
      }
    }
  } else {
    __syncthreads();
  }
}

// RR = rows per step (128 / strip width): 1, 2 or 4 -- a template parameter since round 5: with a runtime row count every wave carried the three
// row roles (9 / 4 / 0 MFMAs in front of the barrier) of its layer, and the two-row instantiation spilled 25 - 33 registers (56 B of scratch per lane)
template <typename T, int RR>
__global__ __launch_bounds__(1024) void compose_stream_bwd_kernel(const CbP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = sfl(tid >> 6);
  const int rows = 10 * p.R + 8;
  const int ring_bytes = rows * (p.FW + 2) * PIXB;
  char* aux = smem + ring_bytes;
  char* tail = aux + AUX_BYTES;
  stage_weights_bwd<T>(smem, aux, p, tid);
  const int G = gridDim.x, g = blockIdx.x;
  const int u0 = (int)(((long)p.units * g) / G), u1 = (int)(((long)p.units * (g + 1)) / G);
  const int nunits = u1 - u0;
  const int steps = nunits > 0 ? (nunits * p.VB + 4 * (p.R + 1) + p.R - 1) / p.R : 0;
  const int layer = wave >> 2, t = wave & 3;
  __syncthreads();
  const int r = t / p.TPR;
#define CB_ROLE(L)                                                                                        \
  do {                                                                                                    \
    if (RR == 1 || r == 0) cb_role<T, L, 9>(p, smem, aux, tail, ring_bytes, t, lane, u0, nunits, steps);  \
    else if (RR == 2 || r == 1) cb_role<T, L, 4>(p, smem, aux, tail, ring_bytes, t, lane, u0, nunits, steps); \
    else cb_role<T, L, 0>(p, smem, aux, tail, ring_bytes, t, lane, u0, nunits, steps);                    \
  } while (0)
  if (layer == 0) CB_ROLE(0);
  else if (layer == 1) CB_ROLE(1);
  else if (layer == 2) CB_ROLE(2);
  else CB_ROLE(3);
#undef CB_ROLE
}

}  // namespace

int dd_compose_stream_plan(int N, int H, int W, int cus, int* out8);

// scratch bytes dd_compose_net_bwd needs for an [N, H, W] launch: four 24-channel gradient tensors and dz6, in the storage type
extern "C" long dd_compose_bwd_scratch_bytes(int N, int H, int W) { return (long)N * H * W * (5 * 48 + 16 + 16); }

int dd_compose_stream_bwd_data_launch(const dd_compose_bwd_args* a, void* scratch, hipStream_t s) {
  CbP p;
  p.small = a->small; p.fine = a->fine; p.gout = a->dout; p.wl = a->wl;
  for (int i = 0; i < 4; ++i) { p.act[i] = a->act[i]; p.ld_act[i] = a->ld_act[i]; }
  p.w_in = a->w_in; p.w_out = a->w_out;
  for (int l = 0; l < 4; ++l) p.w_res[l] = a->w_res[l];
  p.d_small = a->d_small; p.d_fine = a->d_fine;
  const long npix = (long)a->N * a->H * a->W;
  for (int i = 0; i < 5; ++i) p.g[i] = reinterpret_cast<char*>(scratch) + (size_t)i * npix * 48;
  p.x0 = reinterpret_cast<char*>(scratch) + (size_t)5 * npix * 48;
  p.dz6 = reinterpret_cast<char*>(scratch) + (size_t)npix * (5 * 48 + 16);
  p.ld_small = a->ld_small; p.ld_fine = a->ld_fine; p.ld_gout = a->ld_dout; p.ld_wl = a->ld_wl; p.ld_dsmall = a->ld_dsmall; p.ld_dfine = a->ld_dfine;
  p.acc_small = a->accumulate_small;
  p.N = a->N; p.H = a->H; p.W = a->W;
  const int cus = dd_device_cus();
  int g8[8];
  if (dd_compose_stream_plan(a->N, a->H, a->W, cus, g8) != DD_OK) return DD_ERR_INVALID;
  p.FW = g8[0]; p.R = g8[1]; p.TPR = g8[2]; p.n_strips = g8[3]; p.SO = g8[4]; p.BH = g8[5]; p.nb = g8[6]; p.VB = g8[7];
  p.units = a->N * p.n_strips * p.nb;
  const int grid = p.units < cus ? p.units : cus;
  const int lds = (10 * p.R + 8) * (p.FW + 2) * PIXB + AUX_BYTES + (2 * p.R + 2) * p.FW * TAILB;
#define CB_LAUNCH(T, RR)                                                                              \
  do {                                                                                                \
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_stream_bwd_kernel<T, RR>));                \
    hipLaunchKernelGGL((compose_stream_bwd_kernel<T, RR>), dim3(grid), dim3(1024), lds, s, p);        \
  } while (0)
#define CB_LAUNCH_R(T) do { if (p.R == 1) CB_LAUNCH(T, 1); else if (p.R == 2) CB_LAUNCH(T, 2); else CB_LAUNCH(T, 4); } while (0)
  DD_REQUIRE(p.R == 1 || p.R == 2 || p.R == 4, "dd_compose_net_bwd: %d rows per step", p.R);
  if (a->dtype == DD_BF16) CB_LAUNCH_R(bf16_t); else CB_LAUNCH_R(f16_t);
#undef CB_LAUNCH_R
#undef CB_LAUNCH
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ============================================================================================================================
// Weight gradients of the compose net as ONE streaming launch: every activation the forward stored and every gradient tensor the data
// launch above wrote is read once.
//     dK_l[tap][ci][co] = sum_q act_l[q][ci] g_l[q - (tap - 1)][co]   (l = 1..4: act = a1, relu(r1), relu(a2), relu(r3); g = dc1, d a2, dc3, dA)
//     db_l = sum g_l;   dW1 = x0 (x) dz1, db1 = sum dz1;   dW6 = a3 (x) dz6, db6 = sum dz6        (dA = w_out dz6 is rebuilt from dz6)
// A workgroup owns a strip of 64 image columns and walks down a band of rows; per step one activation row and the three gradient rows around
// it sit in LDS as [pixel][24 channels] (LDS-DMA straight from global memory, one row ahead; pixels outside the image come from a page of
// zeros).  14 waves: wave (l, dy) of twelve holds the three taps (dy, 0..2) of layer l -- 3 x 16 accumulator registers for the whole launch --
// and per 16 pixels reads ONE activation fragment and ONE 10-pixel window of its gradient row through the transposing LDS read
// (ds_read_b64_tr_b16: the reduction index of a weight gradient is the pixel); the fragments of the taps dx = 0 and 2 are register sub-ranges of
// the window, the one of dx = 1 four v_alignbit: 5 transposing reads per 3 MFMAs (32x32x16) instead of 8.  The bias gradient is row 24 of the
// centre tap (an all-ones activation channel).  Waves 12 / 13 take the two 1x1 layers and build the rows that exist nowhere in memory (the
// packed net input, dA).  One flush of fp32 atomics per workgroup at the end.
namespace {

constexpr int WG_SW = 64;                    // strip width
constexpr int WG_GROW = 4096, WG_AROW = 3072; // bytes of a gradient row in LDS (85 pixels: 1 halo + 64 + 1 halo + slack) / of an activation row
constexpr int WG_WAVES = 12;

struct CwP {
  const void* act[5];                       // a1, relu(r1), a2, relu(r3), a3   ([pixel][24])
  const void* g[5];                         // dz1, dc1, d a2, dc3, dA         ([pixel][24])
  const void* x0;                           // [pixel][8]: the packed net input + ones channel
  const void* dz6;                          // [pixel]
  const void* zero16;                       // 16 bytes of zeros in global memory
  float* dw_in; float* db_in; float* dw_res[4]; float* db_res[4]; float* dw_out; float* db_out;
  int N, H, W, strips, BH, nb, units;
};

typedef __attribute__((address_space(3))) s16x4_t* cw_tr_ptr;
__device__ __forceinline__ uint2 cw_tr(unsigned addr) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<cw_tr_ptr>(addr));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ void cw_dma_1k(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gptr), "s"(lds_addr) : "memory");
}

template <int N> __device__ __forceinline__ void cw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// LDS map: gradient rings G[4 tensors][5 rows] (dc1, d a2, dc3, dA), activation rows A[4][3], a3 rows [3], dz1 rows [3], the packed net
// input rows [3][64 px][16 B] and dz6 rows [3][64 px][2 B] (every block a DMA target: 1-KiB aligned)
constexpr int CW_G_OFF = 0, CW_A_OFF = CW_G_OFF + 20 * WG_GROW, CW_A3_OFF = CW_A_OFF + 12 * WG_AROW, CW_Z1_OFF = CW_A3_OFF + 3 * WG_AROW;
constexpr int CW_X0_OFF = CW_Z1_OFF + 3 * WG_AROW, CW_Z6_OFF = CW_X0_OFF + 3 * 1024, CW_LDS = CW_Z6_OFF + 3 * 1024;

template <typename T>
__global__ __launch_bounds__(WG_WAVES * 64) void compose_stream_wgrad_kernel(const CwP p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr uint32_t ONE = One<T>::v, ONE2 = ONE | (ONE << 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = sfl(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int H = p.H, W = p.W;
  const int G = gridDim.x, gb = blockIdx.x;
  const int u0 = (int)(((long)p.units * gb) / G), u1 = (int)(((long)p.units * (gb + 1)) / G);

  // ---- roles: wave = (layer 0..3 = conv2d_1..4, tap row dyi); waves 0 and 1 also take the two 1x1 layers (acc[3])
  const int layer = wave / 3, dyi = wave % 3;
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int m = lane & 31, kh = lane >> 5, t16 = lane & 15, cblk = (lane >> 4) & 1;
  // lane part of a transposing fragment read: pixel (t16 >> 2) of a 4-pixel run, channels 16 cblk + 4 (t16 & 3) ..
  const unsigned tr_lane = (unsigned)((t16 >> 2) * PIXB + cblk * 32 + (t16 & 3) * 8);
  const bool ones_row = m == 24;
  auto gslot = [](int row) { return (row + 10) % 5; };
  auto aslot = [](int row) { return (row + 9) % 3; };

  for (int u = u0; u < u1; ++u) {
    const int j = u % p.nb, tq = u / p.nb, st = tq % p.strips, b = tq / p.strips;
    const int yb0 = j * p.BH, yb1 = min(H, yb0 + p.BH), xs = st * WG_SW;
    // Rows arrive TWO steps ahead of their use (the step is shorter than an HBM round trip): `dma(y)` requests row y of the four activations,
    // of a3, dz1, the packed net input and dz6, and row y + 1 of the four gradient tensors: 36 one-KiB pieces, 3 per wave.  A piece's
    // lane-dependent part (source address at row 0, whether its pixel's column exists, destination) is worked out ONCE per unit: per step a
    // piece costs a 64-bit add, two selects and the DMA instruction.
    // (three named records, not an array: hipcc keeps arrays of such records in scratch memory)
    struct Piece { const char* src; bool colok; unsigned dst; int isg, rowb; size_t rb; };
    auto make_piece = [&](int c) {
      Piece q;
      int chunk, gx, part = 0;
      const char* base;
      size_t pxbytes = 48;
      q.rb = (size_t)W * 48;
      if (c < 16) {                                          // gradient rings 0..3 = g[1..4], 4 pieces each
        const int tensor = c >> 2; chunk = c & 3;
        const int piece = chunk * 64 + lane, px = (piece * 171) >> 9;      // (piece / 3 for piece < 256)
        part = piece - px * 3; gx = xs - 1 + px;
        base = reinterpret_cast<const char*>(p.g[1 + tensor]);
        q.isg = 1; q.rowb = WG_GROW;
        q.dst = lds0 + CW_G_OFF + tensor * 5 * WG_GROW + chunk * 1024;
      } else if (c < 34) {                                   // activations 0..3, a3, dz1: 3 pieces each
        const int tensor = (c - 16) / 3; chunk = (c - 16) % 3;
        const int piece = chunk * 64 + lane, px = (piece * 171) >> 9;
        part = piece - px * 3; gx = xs + px;
        base = tensor < 5 ? reinterpret_cast<const char*>(p.act[tensor]) : reinterpret_cast<const char*>(p.g[0]);
        q.isg = 0; q.rowb = WG_AROW;
        q.dst = lds0 + chunk * 1024 + (tensor < 4 ? CW_A_OFF + tensor * 3 * WG_AROW : tensor == 4 ? CW_A3_OFF : CW_Z1_OFF);
      } else if (c == 34) {                                  // the packed net input: 16 bytes per pixel, one piece per pixel
        gx = xs + lane; base = reinterpret_cast<const char*>(p.x0); pxbytes = 16; q.rb = (size_t)W * 16;
        q.isg = 0; q.rowb = 1024; q.dst = lds0 + CW_X0_OFF;
      } else {                                               // dz6: 2 bytes per pixel, 8 pieces carry the 64 pixels
        gx = lane < 8 ? xs + 8 * lane : W; base = reinterpret_cast<const char*>(p.dz6); pxbytes = 2; q.rb = (size_t)W * 2;
        q.isg = 0; q.rowb = 1024; q.dst = lds0 + CW_Z6_OFF;
      }
      q.colok = (unsigned)gx < (unsigned)W;
      // (a strip's last dz6 piece may straddle the image edge: it then carries the next row's first values, which only ever meet zeros --
      // the a3 row is zero there, and the ones of the bias row are switched off beyond the edge)
      q.src = base + ((size_t)(b * H) * W + (q.colok ? gx : 0)) * pxbytes + part * 16;
      return q;
    };
    const Piece q0 = make_piece(wave), q1 = make_piece(wave + 12), q2 = make_piece(wave + 24);
    auto issue = [&](const Piece& q, int y) {
      const int row = y + q.isg;
      const bool ok = (unsigned)row < (unsigned)H && q.colok;
      const char* src = ok ? q.src + (size_t)row * q.rb : reinterpret_cast<const char*>(p.zero16);
      const unsigned dst = q.dst + (q.isg ? gslot(row) : aslot(row)) * q.rowb;
      cw_dma_1k(src, (unsigned)sfl((int)dst));
    };
    auto dma = [&](int y) { issue(q0, y); issue(q1, y); issue(q2, y); };
    // the first call of a unit also needs gradient rows y - 1 and y: the sixteen gradient pieces, twice (waves 0..7 take two of each)
    auto dma_first = [&](int y) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int c = wave + 12 * e;                          // 0..31 used
        if (c < 32) {
          const int tensor = (c >> 2) & 3, chunk = c & 3, row = y - 1 + (c >> 4);
          const int piece = chunk * 64 + lane, px = (piece * 171) >> 9, part = piece - px * 3;
          const int gx = xs - 1 + px;
          const bool ok = (unsigned)row < (unsigned)H && (unsigned)gx < (unsigned)W;
          const char* src = ok ? reinterpret_cast<const char*>(p.g[1 + tensor]) + ((size_t)((b * H + row) * W + gx) * 24) * 2 + part * 16
                               : reinterpret_cast<const char*>(p.zero16);
          cw_dma_1k(src, (unsigned)sfl((int)(lds0 + CW_G_OFF + (tensor * 5 + gslot(row)) * WG_GROW + chunk * 1024)));
        }
      }
    };
    dma_first(yb0);
    dma(yb0);
    dma(yb0 + 1);
    for (int y = yb0; y < yb1; ++y) {
      cw_wait_vm<3>();                                        // the pieces of row y were requested two calls ago; the latest call's 3 may be in flight
      __syncthreads();
      dma(y + 2);
      // ---- this row: 4 K-steps of 16 pixels
      const int grow = y - (dyi - 1);                                                   // tap row dyi pairs act row y with gradient row y - (dyi - 1)
      const unsigned abase = lds0 + CW_A_OFF + (layer * 3 + aslot(y)) * WG_AROW + tr_lane + kh * 8 * PIXB;
      const unsigned gbase = lds0 + CW_G_OFF + (layer * 5 + gslot(grow)) * WG_GROW + tr_lane + kh * 8 * PIXB;      // position 0 = pixel xs - 1
      if ((unsigned)grow < (unsigned)H) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // A: 8 pixels x0 + 8 kh .. of channel m
          const uint2 a_lo = cw_tr(abase + k * 16 * PIXB), a_hi = cw_tr(abase + k * 16 * PIXB + 4 * PIXB);
          uint4 af = {a_lo.x, a_lo.y, a_hi.x, a_hi.y};
          if (layer == 2) af = relu16<T>(af);                                           // conv2d_3 read relu(a2); a2 is stored raw
          if (ones_row && dyi == 1) af = uint4{ONE2, ONE2, ONE2, ONE2};               // channel 24 := 1: row 24 of the centre tap = bias gradient
          // window of the gradient row: pixels q - 1 .. q + 10 for the activation pixels q = x0 + 8 kh .. (ring position + 1 = same pixel)
          const uint2 w01 = cw_tr(gbase + k * 16 * PIXB), w23 = cw_tr(gbase + k * 16 * PIXB + 4 * PIXB), w45 = cw_tr(gbase + k * 16 * PIXB + 8 * PIXB);
          // tap dx = 0 pairs act pixel q with gradient pixel q + 1, dx = 1 with q, dx = 2 with q - 1
          const uint4 f_m1 = {w01.x, w01.y, w23.x, w23.y};
          const uint4 f_p1 = {w01.y, w23.x, w23.y, w45.x};
          const uint4 f_0 = {__builtin_amdgcn_alignbit(w01.y, w01.x, 16), __builtin_amdgcn_alignbit(w23.x, w01.y, 16),
                             __builtin_amdgcn_alignbit(w23.y, w23.x, 16), __builtin_amdgcn_alignbit(w45.x, w23.y, 16)};
          acc[0] = mma32<T>(af, f_p1, acc[0]);
          acc[1] = mma32<T>(af, f_0, acc[1]);
          acc[2] = mma32<T>(af, f_m1, acc[2]);
        }
      }
      if (wave == 0) {
        // dW1[k][n] += x0[k] dz1[n] (row 6 of the result: db1, the ones channel of the packed net input)
        const unsigned xbase = lds0 + CW_X0_OFF + aslot(y) * 1024 + (unsigned)((t16 >> 2) * 16 + (t16 & 3) * 8) + kh * 8 * 16;
        const unsigned zbase = lds0 + CW_Z1_OFF + aslot(y) * WG_AROW + tr_lane + kh * 8 * PIXB;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint4 af = {0u, 0u, 0u, 0u};
          if (cblk == 0) {                                                                // 8 channels per pixel: only the first 16-channel block exists
            const uint2 a_lo = cw_tr(xbase + k * 16 * 16), a_hi = cw_tr(xbase + k * 16 * 16 + 4 * 16);
            af = uint4{a_lo.x, a_lo.y, a_hi.x, a_hi.y};
          }
          const uint2 b_lo = cw_tr(zbase + k * 16 * PIXB), b_hi = cw_tr(zbase + k * 16 * PIXB + 4 * PIXB);
          acc[3] = mma32<T>(af, uint4{b_lo.x, b_lo.y, b_hi.x, b_hi.y}, acc[3]);
        }
      } else if (wave == 1) {
        // dW6[n] += a3[n] dz6 (row 24: db6): the B operand has ONE column, 8 consecutive dz6 values per k-half
        const unsigned a3base = lds0 + CW_A3_OFF + aslot(y) * WG_AROW + tr_lane + kh * 8 * PIXB;
        const char* z6 = smem + CW_Z6_OFF + aslot(y) * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint2 a_lo = cw_tr(a3base + k * 16 * PIXB), a_hi = cw_tr(a3base + k * 16 * PIXB + 4 * PIXB);
          uint4 af = {a_lo.x, a_lo.y, a_hi.x, a_hi.y};
          if (ones_row) {
            const int px = xs + k * 16 + kh * 8;               // W is even: pixel pairs are inside or outside together
            af = uint4{px < W ? ONE2 : 0u, px + 2 < W ? ONE2 : 0u, px + 4 < W ? ONE2 : 0u, px + 6 < W ? ONE2 : 0u};
          }
          uint4 bfr = {0u, 0u, 0u, 0u};
          if (m == 0) bfr = *reinterpret_cast<const uint4*>(z6 + (k * 16 + kh * 8) * 2);
          acc[3] = mma32<T>(af, bfr, acc[3]);
        }
      }
    }
    cw_wait_vm<0>();                                          // the rows requested past the band
    __syncthreads();                                          // the next unit's prologue overwrites the rows
  }

  // ---- flush: one atomic per gradient element per workgroup
  dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups flush in index order, dd_common.h)
  if (u1 > u0) {
    float* dw = p.dw_res[layer];
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int tap = dyi * 3 + dxi;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i >> 2) * 8 + kh * 4 + (i & 3), col = m;
        if (col < 24) {
          if (row < 24) atomicAdd(dw + (tap * 24 + row) * 24 + col, acc[dxi][i]);
          else if (row == 24 && tap == 4) atomicAdd(p.db_res[layer] + col, acc[dxi][i]);
        }
      }
    }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i >> 2) * 8 + kh * 4 + (i & 3), col = m;
        if (col < 24) {
          if (row < 6) atomicAdd(p.dw_in + row * 24 + col, acc[3][i]);
          else if (row == 6) atomicAdd(p.db_in + col, acc[3][i]);
        }
      }
    } else if (wave == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i >> 2) * 8 + kh * 4 + (i & 3);
        if (m == 0) {
          if (row < 24) atomicAdd(p.dw_out + row, acc[3][i]);
          else if (row == 24) atomicAdd(p.db_out, acc[3][i]);
        }
      }
    }
  }
  dd_det_end();
}

}  // namespace

static __device__ uint4 dd_cw_zero_page = {0u, 0u, 0u, 0u};

int dd_compose_stream_wgrad_launch(const dd_compose_bwd_args* a, void* scratch, hipStream_t s) {
  CwP p;
  for (int i = 0; i < 5; ++i) p.act[i] = a->act[i];
  const long npix = (long)a->N * a->H * a->W;
  for (int i = 0; i < 5; ++i) p.g[i] = reinterpret_cast<char*>(scratch) + (size_t)i * npix * 48;
  p.x0 = reinterpret_cast<char*>(scratch) + (size_t)5 * npix * 48;
  p.dz6 = reinterpret_cast<char*>(scratch) + (size_t)npix * (5 * 48 + 16);
  void* zp = nullptr;
  if (hipGetSymbolAddress(&zp, HIP_SYMBOL(dd_cw_zero_page)) != hipSuccess) { dd_set_error("dd_compose_net_bwd: no zero page"); return DD_ERR_LAUNCH; }
  p.zero16 = zp;
  p.dw_in = a->dw_in; p.db_in = a->db_in; p.dw_out = a->dw_out; p.db_out = a->db_out;
  for (int l = 0; l < 4; ++l) { p.dw_res[l] = a->dw_res[l]; p.db_res[l] = a->db_res[l]; }
  p.N = a->N; p.H = a->H; p.W = a->W;
  p.strips = (a->W + WG_SW - 1) / WG_SW;
  const int cus = dd_device_cus();
  // bands: enough units to fill the device, rows a multiple of 2
  long best = -1; int bestBH = a->H;
  for (int d = 1; d <= 32; ++d) {
    int BH = (a->H + d - 1) / d; BH += BH & 1;
    if (BH < 2) BH = 2;
    const int nb = (a->H + BH - 1) / BH;
    const long units = (long)a->N * p.strips * nb;
    const long Gn = units < cus ? units : cus;
    const long cost = ((units + Gn - 1) / Gn) * (BH + 2);
    if (best < 0 || cost < best) { best = cost; bestBH = BH; }
  }
  p.BH = bestBH; p.nb = (a->H + bestBH - 1) / bestBH;
  p.units = a->N * p.strips * p.nb;
  const int grid = p.units < cus ? p.units : cus;
  constexpr int LDS = CW_LDS;
  dd_det_sync();
  if (a->dtype == DD_BF16) {
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_stream_wgrad_kernel<bf16_t>));
    hipLaunchKernelGGL(compose_stream_wgrad_kernel<bf16_t>, dim3(grid), dim3(WG_WAVES * 64), LDS, s, p);
  } else {
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_stream_wgrad_kernel<f16_t>));
    hipLaunchKernelGGL(compose_stream_wgrad_kernel<f16_t>, dim3(grid), dim3(WG_WAVES * 64), LDS, s, p);
  }
  DD_LAUNCH_CHECK();
  return DD_OK;
}
