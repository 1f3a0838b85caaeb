// Row-streaming compose net on CDNA4 (round 4): MultiScalePrediction.compose_scales forward as ONE launch with NO halo recompute in y.
//
// Reference seam replaced (file:line in /root/reference): TensorFlow/MultiScalePrediction.py:36-93 (compose_scales ->
// _compose_scales_neural_network -> _residual_block x 2), called per scale transition by MultiScalePredictor.predict
// (Architecture.py:302-325).  The net itself is stated at the top of dd_compose.hip (the 16x16-tile kernel this one supersedes: 24x24 frame per
// 16x16 outputs = 2.25x recompute, one workgroup per CU whose phases did not overlap, ~10 % of the MFMA peak).
//
// Here a workgroup (16 waves) owns a column strip of up to 128 frame columns and walks DOWN a band of image rows.  The four 3x3 layers run as
// a software pipeline over rows: in step s
//     stage 0  a1 = relu(1x1(x0))            row  s            (waves 0-3, besides their stage-1 task)
//     stage 1  r1 = relu(conv(a1))           row  s - 2        waves 0-3
//     stage 2  a2 = a1 + conv(r1)            row  s - 4        waves 4-7
//     stage 3  r3 = relu(conv(relu(a2)))     row  s - 6        waves 8-11
//     stage 4  a3 = a2 + conv(r3), 1x1, sigmoid, blend -> out   row  s - 8        waves 12-15
// (for frames narrower than 128 columns a step covers R = 128 / FW rows and stage k lags k (R + 1) rows), every stage reading rows that were
// written in EARLIER steps, so one barrier per step is the only synchronisation.  Each stage keeps a rolling window of its input in LDS
// ([row slot][column + 1][24 channels], zero border columns, 48-byte pixels): rings a1 (3R + 2 rows: window + the residual of stage 2), r1,
// relu(a2), r3 (2R + 2 each) and raw a2 (3R + 2: the residual of stage 4) -- 22 rows x 130 pixels x 48 B = 134 KiB at FW = 128.
// A wave belongs to ONE layer for the whole launch and keeps that layer's weights in registers as the MFMA A operand; per step it multiplies
// one 32-pixel task:
//   v_mfma_f32_32x32x16: M = 32 output channels (24 real), N = 32 pixels, K = 16 = two 8-channel k-groups.  The 9 x 24 = 216 reduction values
//   are packed densely into 27 k-groups; group 27 carries the BIAS (the B operand holds 1, 1, 1 there, the A operand the fp32 bias split into
//   three storage-type terms hi + mid + lo, exact to 24 bits); a residual layer's accumulator starts from the residual pixel: 14 MFMAs per task.
//   With 32 pixels per fragment a ds_read_b128 service group (MI355X_MICROARCH.md, LDS table) holds 16 lanes of the SAME k-group whose pixels
//   {0-3, 12-15, 20-27} x 48 B hit 16 distinct bank quads: conflict-free without a swizzle.
// Measured on the first version (rocprofv3 --pmc, tools/pmc_compose_stream.sh: waves 64 % parked, 19 % issuing, matrix pipe 28 % busy, ~137
// vector-ALU + 79 scalar instructions per wave and step): bias moves, residual unpack + add, a ReLU on each of stage 3's 14 B fragments, two
// conversions + a permute per packed pair and per-lane zero-padding selects.  Hence: bias and the 24 -> 1 output layer go through the matrix
// pipe, relu(a2) is stored once by its producer instead of being applied to 14 fragments by its consumer, the zero-padding selects only run
// for tasks that touch the image border, all LDS reads of a batch of K-steps are issued before its first MFMA, and the scalar state is
// advanced incrementally: 74 vector + 59 scalar instructions per wave and step.
// What bounds it now (knock-out builds, tools/build_variant.sh -DCS_EXP_*, 128 x 128 x 128 at 4 150 cycles per step): the barrier keeps the
// 16 waves in phase, so they read LDS at the same time and multiply at the same time: ~1 230 cycles of LDS-read time per step ADD to ~1 800
// of MFMA time instead of hiding behind it (no LDS reads: -47 us of 99 with the MFMAs already out; no MFMAs: -40 of 158; stage 0 + tail
// -28).  Reads of the next batch in flight behind the MFMAs of the current one need two batches of fragments beside 14 weight fragments and
// the accumulator -- more than 128 registers (built: 50 - 170 bytes of scratch per lane and slower; with part of the weights re-read from
// LDS: 180 us against 158).  The way on is 8 waves of 256 registers with two tasks each.
// A SIMD hosts one wave of every layer (wave w runs on SIMD w % 4), so the four SIMDs carry equal MFMA work.
// Bands of rows are concatenated into one stream of "virtual rows" (each band contributes its rows plus 4 halo rows either side; the rows a
// stage produces across a band junction are garbage that no valid output depends on), so the pipeline fills and drains once per launch.
// Images wider than 128 columns are cut into column strips with a 4-column halo (the only recompute left).
#include "dd_compose_stream.h"

#ifdef CS_PROFILE
// cycle stamps of workgroup 0, task-slot-0 wave of every layer (tools/compose_stream_phases.py): [layer][phase] accumulated cycles
__device__ unsigned long long cs_phase[4][8];
#define CSP_DECL() unsigned long long csp_last = __builtin_readcyclecounter(); const bool csp_on = blockIdx.x == 0 && t == 0 && lane == 0
#define CSP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (csp_on) cs_phase[LAYER][i] += now_ - csp_last; csp_last = now_; } while (0)
extern "C" int dd_debug_cs_phases(unsigned long long* out32, int reset) {
  if (out32) (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(cs_phase), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(cs_phase), z, sizeof(z)); }
  return 0;
}
#else
#define CSP_DECL()
#define CSP(i)
#endif


namespace {

constexpr int KS_MAX = 16;        // K-steps of 16 per 3x3 layer: 14 (27 k-groups + the bias group); the staged image has room for 16
constexpr int AUX_BYTES = 16 + 3 * 1024;      // behind the rings: the [1, 1, 1, 0 ...] vector, the output layer's two A fragments, the input layer's

struct CsP {
  const float* small; const float* fine; float* out;
  const float* w_in; const float* b_in; const float* w_res[4]; const float* b_res[4]; const float* w_out; const float* b_out;
  void* save_act[5]; void* save_wl; void* save_netin;
  int ld_small, ld_fine, ld_out, ld_act[5], ld_wl, ld_netin;
  int N, H, W;
  int FW, R, TPR;               // frame width (multiple of 32, <= 128), rows per step, 32-pixel tasks per row
  int n_strips, SO;             // column strips and their output width
  int BH, nb, VB;               // band height, bands per image, virtual rows per band (BH + 8)
  int units;                    // N * n_strips * nb
};

// A operands, staged once per workgroup in LDS.  3x3 layers (over the ring area, before it is zeroed): image [layer][K-step c][lane] of 16
// bytes; lane = (m = lane % 32 = output channel, h = lane / 32), k-group g = 2c + h:
//   g < 27        tap g / 3, input channels 8 (g % 3) .. + 7 of the HWIO fp32 master weights [tap][ci][co], rounded to the storage type (as
//                 dd_pack_weights does)
//   g = 27        the bias of channel m as hi, mid, lo (the B operand holds 1, 1, 1, 0 ... there)
// Behind the rings: the ones vector, the 24 -> 1 output layer (row 0 only; its K order is the C layout of the a3 accumulators: element e of
// k-half h = channel 4h + e | 8 + 4h + e - 4, then 16 + 4h + e and the bias terms) and the 6 -> 24 input layer (k-half 0: the fine image's 3
// channels + the bias terms, k-half 1: small's).
template <typename T>
__device__ __forceinline__ void stage_weights(char* img, char* aux, const CsP& p, int tid) {
  for (int i = tid; i < 4 * KS_MAX * 64 * 8; i += 1024) {
    const int e = i & 7, ln = (i >> 3) & 63, c = (i >> 9) & (KS_MAX - 1), l = i >> 13;
    const int m = ln & 31, g = 2 * c + (ln >> 5);
    float v = 0.f;
    if (m < 24) {
      if (g < 27) v = p.w_res[l][((g / 3) * 24 + 8 * (g % 3) + e) * 24 + m];
      else if (g == 27) { if (e < 3) v = split3<T>(p.b_res[l][m], e); }
    }
    reinterpret_cast<T*>(img)[i] = Elem<T>::from_f32(v);
  }
  T* ax = reinterpret_cast<T*>(aux);
  for (int i = tid; i < 8 + 3 * 512; i += 1024) {
    float v = 0.f;
    if (i < 8) v = i < 3 ? 1.f : 0.f;
    else {
      const int j = i - 8, e = j & 7, ln = (j >> 3) & 63, which = j >> 9;
      const int m = ln & 31, h = ln >> 5;
      if (which == 0) { if (m == 0) v = p.w_out[e < 4 ? 4 * h + e : 8 + 4 * h + e - 4]; }
      else if (which == 1) { if (m == 0) v = e < 4 ? p.w_out[16 + 4 * h + e] : (e < 7 && h == 0) ? split3<T>(p.b_out[0], e - 4) : 0.f; }
      else if (m < 24) v = e < 3 ? p.w_in[((h ? 0 : 3) + e) * 24 + m] : (e < 6 && h == 0) ? split3<T>(p.b_in[m], e - 3) : 0.f;
    }
    ax[i] = Elem<T>::from_f32(v);
  }
}

// One role = one 3x3 layer (LAYER 0..3 = the net's conv2d_1..4); LAYER 0 also runs the 1x1 input layer, LAYER 3 the output tail.
// The step loop is written for instruction COUNT (a SIMD issues roughly one instruction per 4 cycles whichever of its waves it comes from):
// scalar state is advanced incrementally (ring row offsets in bytes, one wrap each), per-unit values are recomputed only when the unit changes,
// and whether a stage is active follows from the step index alone.
template <typename T, int LAYER, bool SAVE, int NPRE>
__device__ __forceinline__ void cs_role(const CsP& p, char* smem, char* aux, int ring_bytes, int t, int lane, int u0, int nunits, int steps) {
  constexpr bool RESIDUAL = LAYER == 1 || LAYER == 3;        // a2 = a1 + conv, a3 = a2 + conv (raw); else relu(conv)
  constexpr bool FINAL = LAYER == 3;
  constexpr int KS = 14;
  constexpr uint32_t ONE = One<T>::v, ONE2 = ONE | (ONE << 16);
  const int R = p.R, TPR = p.TPR, FW = p.FW, VB = p.VB;
  const int n = lane & 31, h = lane >> 5;
  const int r = t / TPR, xc = t - r * TPR;
  const bool has_task = t < R * TPR;
  const int x = xc * 32 + n;                                 // frame-local column of this lane's pixel
  const int pitch = (FW + 2) * PIXB, rp = R * pitch;
  // rings: 0 a1, 1 r1, 2 relu(a2), 3 r3, 4 a2 (raw)
  const int D0 = 3 * R + 2, D1 = 2 * R + 2, D4 = 3 * R + 2;
  const int off1 = D0 * pitch, off2 = off1 + D1 * pitch, off3 = off2 + D1 * pitch, off4 = off3 + D1 * pitch;
  const int Din = LAYER == 0 ? D0 : D1;
  const int Dres = LAYER == 1 ? D0 : D4;
  const int in_bytes = Din * pitch, out_bytes = D1 * pitch, res_bytes = Dres * pitch, raw_bytes = D4 * pitch;
  char* rin = smem + (LAYER == 0 ? 0 : LAYER == 1 ? off1 : LAYER == 2 ? off2 : off3);
  char* rout = smem + (LAYER == 0 ? off1 : LAYER == 1 ? off2 : off3);       // (LAYER 3 writes no ring)
  char* rres = smem + (LAYER == 1 ? 0 : off4);

  uint4 wa[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) wa[c] = *reinterpret_cast<const uint4*>(smem + ((LAYER * KS_MAX + c) * 64 + lane) * 16);
  __syncthreads();                                           // every wave holds its fragments: the image may be overwritten
  for (int i = threadIdx.x; i < ring_bytes / 16; i += 1024) reinterpret_cast<uint4*>(smem)[i] = uint4{0u, 0u, 0u, 0u};

  const int H = p.H, W = p.W, h2 = H >> 1, w2 = W >> 1;
  const int lag = (LAYER + 1) * (R + 1);
  // stage k works on virtual row V = s R + r - lag in step s: active for s in [s_beg, s_end)
  const int total_rows = nunits * VB;
  const int s_beg = has_task ? (lag - r + R - 1) / R : steps, s_end = has_task ? (total_rows + lag - r + R - 1) / R : steps;
  Cursor cur;
  cur.init(p, u0, nunits, r - lag);
  // byte offsets of the rows this step's task touches, relative to their rings (row V -> slot V mod depth)
  int o_in = posmod(r - lag, Din) * pitch, o_out = posmod(r - lag, D1) * pitch, o_res = posmod(r - lag, Dres) * pitch, o_raw = posmod(r - lag, D4) * pitch;
  auto wrap_add = [](int o, int add, int bytes) { o += add; return o >= bytes ? o - bytes : o; };
  // lane constants of the fragment addresses
  const int xo = x * PIXB + h * 16;
  const int a4c = h ? x * PIXB : x * PIXB + 128;             // K-step 4: group 8 (row y - 1, k-half 0) | group 9 (row y, k-half 1)
  const int a13c = h ? (int)(aux - rin) : x * PIXB + 128;    // K-step 13: group 26 (row y + 1) | the ones of the bias group (a constant address)
  const int wr = (x + 1) * PIXB + h * 8;                     // where this lane's 4-channel groups go in a ring row
  // per-unit values (recomputed when the cursor enters a unit)
  int ybase = 0, pixb = 0;                                   // image row of virtual row 0; pixel index of (row 0, this lane's column)
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

  // stage 0 (LAYER 0 waves): cursor and input registers one step ahead
  Cursor c0;
  int o0 = 0, ybase0 = 0, pixb0 = 0, pix0 = 0;
  bool col_in0 = false, col_own0 = false, cols_all_in0 = false, in0 = false, own0 = false, row_in0 = false;
  float xin[3] = {0.f, 0.f, 0.f};
  const int s0_end = has_task ? (total_rows - r + R - 1) / R : 0;
  auto load_x0 = [&](int s_next) {
    in0 = false; own0 = false; row_in0 = false;
    if (s_next < s0_end) {
      const int y = ybase0 + c0.i;
      row_in0 = (unsigned)y < (unsigned)H;
      in0 = row_in0 && col_in0;
      own0 = in0 && col_own0 && y >= c0.U.yb0 && y < c0.U.yb1;
      const int cy = row_in0 ? y : 0;
      pix0 = pixb0 + cy * W;
      const int cx = pix0 - (c0.U.b * H + cy) * W;
      const float* src = h ? p.small + (size_t)((c0.U.b * h2 + (cy >> 1)) * w2 + (cx >> 1)) * p.ld_small : p.fine + (size_t)pix0 * p.ld_fine;
#ifdef CS_EXP_NO_STAGE0_LOADS
      xin[0] = (float)pix0; xin[1] = 1.f; xin[2] = 0.5f;
      (void)src;
#else
      xin[0] = src[0]; xin[1] = src[1]; xin[2] = src[2];
#endif
    }
  };
  if (LAYER == 0) {
    c0.init(p, u0, nunits, r);
    o0 = posmod(r, D0) * pitch;
    unit_values(c0, ybase0, pixb0, col_in0, col_own0, cols_all_in0);
    load_x0(0);
  }

  // The 3x3 task of a step is split around the barrier.  Of the three input rows only the last (row y + 1 for the first row of a step) was
  // written in the PREVIOUS step; the K-steps that read older rows -- 0..8 for the first row of a step (all of them when R = 1), 0..3 for the
  // second, none further down -- and the residual are read and multiplied BEFORE the barrier, right after the previous task's epilogue; the
  // accumulator crosses the barrier and only the remaining K-steps wait for it.
  f32x16_t acc;
  auto conv_part_n = [&](auto pre_t) {
    constexpr bool pre = decltype(pre_t)::value;
    constexpr int C0 = pre ? 0 : NPRE, C1 = pre ? NPRE : KS;
    const int o_m1 = o_in == 0 ? in_bytes - pitch : o_in - pitch;
    const int o_p1 = o_in + pitch == in_bytes ? 0 : o_in + pitch;
    // K-steps 0..3: k-groups 0..7 of row y - 1 at + 32 c; 5..8: groups 10..17 of row y at + 32 (c - 5); 9..12: groups 18..25 of row y + 1
    const char* q0 = rin + xo + o_m1;
    const char* q1 = rin + xo + 16 + o_in;
    const char* q2 = rin + xo + o_p1;
    const char* q4 = rin + a4c + (h ? o_in : o_m1);
    const char* q13 = rin + a13c + (h ? 0 : o_p1);
    // Every LDS read of a batch is issued before its first MFMA (sched_barrier: left alone hipcc keeps two fragments in flight and the wave
    // pays one LDS round trip per pair of MFMAs); 128 registers hold the 14 weight fragments, the accumulator and 6 operand fragments.
    f32x16_t c_init = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (RESIDUAL && pre) {
      // the accumulator starts from the residual pixel (a2 = a1 + conv, a3 = a2 + conv); the bias is K-group 27
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
#ifdef CS_EXP_NO_LDSREAD
        bf[c - c0_] = uint4{(uint32_t)x, (uint32_t)c, 0u, 0u};
#else
        bf[c - c0_] = *reinterpret_cast<const uint4*>(a);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = c0_; c < c0_ + BATCH && c < C1; ++c) {
#ifdef CS_EXP_NO_MFMA
        acc[c] = ((pre && c == 0) ? c_init[c] : acc[c]) + __uint_as_float(bf[c - c0_].x ^ wa[c].x);
#else
        acc = mma32<T>(wa[c], bf[c - c0_], (pre && c == 0) ? c_init : acc);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (pre && NPRE == 0) acc = c_init;
  };

  bool act = 0 >= s_beg && 0 < s_end;
  if (act) conv_part_n(std::true_type());

  CSP_DECL();
  for (int s = 0; s < steps; ++s) {
    CSP(0);
#ifndef CS_EXP_NO_BARRIER
    __syncthreads();
#endif
    CSP(1);
    // ------------------------------------------------------------------------------------------ stage 0: a1 = relu(1x1(x0))
    if (LAYER == 0) {
      if (s < s0_end) {
        const uint4 w0 = *reinterpret_cast<const uint4*>(aux + 16 + 2048 + lane * 16);
        const uint4 bx = {pack2<T>(xin[0], xin[1]), pack2<T>(xin[2], h ? 0.f : 1.f), h ? 0u : ONE2, 0u};
        const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x16_t a0 = mma32<T>(w0, bx, z);
        char* o = rin + o0 + wr;
        const bool in_now = in0, own_now = own0, all_in = row_in0 && cols_all_in0;
        const int pix_now = pix0;
        if (SAVE && own_now && p.save_netin) {               // layer-wise backward only: the packed net input [small | fine | 0 0] in the storage type
          T* d = reinterpret_cast<T*>(p.save_netin) + (size_t)pix_now * p.ld_netin + (h ? 0 : 3);
#pragma unroll
          for (int e = 0; e < 3; ++e) d[e] = Elem<T>::from_f32(xin[e]);
          if (!h) { d[3] = Elem<T>::from_f32(0.f); d[4] = Elem<T>::from_f32(0.f); }
        }
        uint2 pk[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          pk[b].x = relu_bf16x2(packo<T>(a0[4 * b], a0[4 * b + 1]));
          pk[b].y = relu_bf16x2(packo<T>(a0[4 * b + 2], a0[4 * b + 3]));
        }
        if (!all_in) {                                       // TensorFlow pads every conv's input with zeros: nothing exists outside the image
#pragma unroll
          for (int b = 0; b < 3; ++b) { pk[b].x = in_now ? pk[b].x : 0u; pk[b].y = in_now ? pk[b].y : 0u; }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          *reinterpret_cast<uint2*>(o + b * 16) = pk[b];
          if (SAVE && own_now && p.save_act[0])
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.save_act[0]) + ((size_t)pix_now * p.ld_act[0] + 8 * b + 4 * h) * 2) = pk[b];
        }
      }
      c0.i += R;
      if (c0.i >= VB) {
        c0.i -= R;
        c0.advance(p, u0, nunits, R);
        unit_values(c0, ybase0, pixb0, col_in0, col_own0, cols_all_in0);
      }
      o0 = wrap_add(o0, rp, D0 * pitch);
      load_x0(s + 1);                                        // next step's inputs: in flight during this step's conv task
    }
    CSP(2);
    // ------------------------------------------------------------------------------------------ this wave's 3x3 layer, one 32-pixel task
    if (act) {
      const int y = ybase + cur.i;
      const bool row_in = (unsigned)y < (unsigned)H;
      const bool all_in = row_in && cols_all_in;             // uniform: no pixel of the task needs zeroing
      const bool inside = row_in && col_in;
      const bool own = inside && col_own && y >= cur.U.yb0 && y < cur.U.yb1;
      const int cy = row_in ? y : 0;
      const int pix = pixb + cy * W;
      // (tail) blend operands, requested before the MFMAs: out = fine - w * up(avg_pool2(fine)) + w * up(small).  k-half h fetches row
      // (y & ~1) + h of the 2x2 block of `fine`; the block sum is one exchange with the x-neighbour lane and one with the other k-half.
      float f3[3] = {0.f, 0.f, 0.f}, s3[3] = {0.f, 0.f, 0.f};
      if (FINAL) {
#ifndef CS_EXP_NO_TAIL_LOADS
        const int cx = pix - (cur.U.b * H + cy) * W;
        const float* fr = p.fine + (size_t)(pix + (h - (cy & 1)) * W) * p.ld_fine;
        const float* sp = p.small + (size_t)((cur.U.b * h2 + (cy >> 1)) * w2 + (cx >> 1)) * p.ld_small;
#pragma unroll
        for (int c = 0; c < 3; ++c) { f3[c] = fr[c]; s3[c] = sp[c]; }
#endif
      }
      conv_part_n(std::false_type());                       // the K-steps that read the row written in the previous step
      CSP(4);
      uint4 wo1 = {0u, 0u, 0u, 0u}, wo2 = {0u, 0u, 0u, 0u};
      if (FINAL) {
        wo1 = *reinterpret_cast<const uint4*>(aux + 16 + lane * 16);
        wo2 = *reinterpret_cast<const uint4*>(aux + 16 + 1024 + lane * 16);
      }
      uint2 pk[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        pk[b].x = packo<T>(acc[4 * b], acc[4 * b + 1]);
        pk[b].y = packo<T>(acc[4 * b + 2], acc[4 * b + 3]);
      }
      if (!FINAL) {
        if (LAYER == 1) {                                    // a2: raw for the residual of stage 4 (and the backward), rectified for stage 3
          char* o4 = smem + off4 + o_raw + wr;
          if (!all_in) {
#pragma unroll
            for (int b = 0; b < 3; ++b) { pk[b].x = inside ? pk[b].x : 0u; pk[b].y = inside ? pk[b].y : 0u; }
          }
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            *reinterpret_cast<uint2*>(o4 + b * 16) = pk[b];
            if (SAVE && own && p.save_act[2])
              *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.save_act[2]) + ((size_t)pix * p.ld_act[2] + 8 * b + 4 * h) * 2) = pk[b];
          }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) { pk[b].x = relu_bf16x2(pk[b].x); pk[b].y = relu_bf16x2(pk[b].y); }
        // TensorFlow pads every conv's input with zeros: nothing exists outside the image
        if (LAYER != 1 && !all_in) {
#pragma unroll
          for (int b = 0; b < 3; ++b) { pk[b].x = inside ? pk[b].x : 0u; pk[b].y = inside ? pk[b].y : 0u; }
        }
        char* o = rout + o_out + wr;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          *reinterpret_cast<uint2*>(o + b * 16) = pk[b];
          if (SAVE && LAYER != 1 && own && p.save_act[LAYER + 1])
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.save_act[LAYER + 1]) + ((size_t)pix * p.ld_act[LAYER + 1] + 8 * b + 4 * h) * 2) = pk[b];
        }
      } else {
#ifdef CS_EXP_NO_TAIL
        if (own && h == 0 && pk[0].x == 0x12345u) p.out[(size_t)pix * p.ld_out] = 1.f;
#else
        // a3 (rounded to the storage type, as the layer-wise path stores it) -> 1x1 -> relu -> sigmoid -> blend.  The 24 -> 1 layer runs on
        // the matrix pipe: the packed a3 registers ARE B fragments (8 channels of this lane's pixel per k-half, in the K order the staged
        // A fragments use); row 0 of the result = the pre-activation of the pixels, in the k-half-0 lanes.
        if (SAVE && own && p.save_act[4]) {
#pragma unroll
          for (int b = 0; b < 3; ++b)
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.save_act[4]) + ((size_t)pix * p.ld_act[4] + 8 * b + 4 * h) * 2) = pk[b];
        }
        const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16_t a2 = mma32<T>(wo1, uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y}, z);
        a2 = mma32<T>(wo2, uint4{pk[2].x, pk[2].y, ONE2, ONE}, a2);
        const uint32_t wlp = pack2<T>(fmaxf(a2[0], 0.f), 0.f);          // relu, stored in the storage type
        float wlv, unused;
        unpack2<T>(wlp, wlv, unused);
        const float wgt = 1.f / (1.f + __expf(-wlv));
        float res[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float pair = f3[c] + dpp_xor1(f3[c]);
          float p_lo, p_hi, f_lo, f_hi;
          both_halves(pair, p_lo, p_hi);
          both_halves(f3[c], f_lo, f_hi);
          const float low = 0.25f * (p_lo + p_hi);
          const float fme = (cy & 1) ? f_hi : f_lo;            // (valid in the k-half-0 lanes, which store)
          res[c] = fme - wgt * low + wgt * s3[c];
        }
        if (own && h == 0) {
          float* o = p.out + (size_t)pix * p.ld_out;
#pragma unroll
          for (int c = 0; c < 3; ++c) o[c] = res[c];
          if (SAVE && p.save_wl) reinterpret_cast<uint16_t*>(p.save_wl)[(size_t)pix * p.ld_wl] = (uint16_t)(wlp & 0xffffu);
        }
#endif
      }
    }
    CSP(5);
    // ------------------------------------------------------------------------------------------ next step's rows
    cur.i += R;
    if (cur.i >= VB) {
      cur.i -= R;
      cur.advance(p, u0, nunits, R);
      unit_values(cur, ybase, pixb, col_in, col_own, cols_all_in);
    }
    o_in = wrap_add(o_in, rp, in_bytes);
    if (!FINAL) o_out = wrap_add(o_out, rp, out_bytes);
    if (RESIDUAL) o_res = wrap_add(o_res, rp, res_bytes);
    if (LAYER == 1) o_raw = wrap_add(o_raw, rp, raw_bytes);
    act = s + 1 >= s_beg && s + 1 < s_end;
    if (act) conv_part_n(std::true_type());                  // next step's task, the part that does not need the barrier
    CSP(6);
  }
}

template <typename T, bool SAVE, bool R1>
__global__ __launch_bounds__(1024) void compose_stream_fwd_kernel(const CsP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = sfl(tid >> 6);
  const int rows = 12 * p.R + 10;
  const int ring_bytes = rows * (p.FW + 2) * PIXB;
  char* aux = smem + ring_bytes;
  stage_weights<T>(smem, aux, p, tid);
  // this workgroup's contiguous range of units
  const int G = gridDim.x, g = blockIdx.x;
  const int u0 = (int)(((long)p.units * g) / G), u1 = (int)(((long)p.units * (g + 1)) / G);
  const int nunits = u1 - u0;
  const int steps = nunits > 0 ? (nunits * p.VB + 4 * (p.R + 1) + p.R - 1) / p.R : 0;
  const int layer = wave >> 2, t = wave & 3;
  __syncthreads();
  // (the first barrier of the step loop orders the zeroing of the rings before any read)
  // K-steps in front of the barrier: 9 for the first row of a step (every task when R = 1), 4 for the second, none further down
  const int r = t / p.TPR;
#define CS_ROLE(L)                                                                                   \
  do {                                                                                               \
    if (R1 || r == 0) cs_role<T, L, SAVE, 9>(p, smem, aux, ring_bytes, t, lane, u0, nunits, steps);  \
    else if (r == 1) cs_role<T, L, SAVE, 4>(p, smem, aux, ring_bytes, t, lane, u0, nunits, steps);   \
    else cs_role<T, L, SAVE, 0>(p, smem, aux, ring_bytes, t, lane, u0, nunits, steps);               \
  } while (0)
  if (layer == 0) CS_ROLE(0);
  else if (layer == 1) CS_ROLE(1);
  else if (layer == 2) CS_ROLE(2);
  else CS_ROLE(3);
#undef CS_ROLE
}

}  // namespace

// Geometry of a launch (exposed for tests / tools through dd_compose_stream_plan): frame width, rows per step, strips, bands.
extern "C" int dd_compose_stream_plan(int N, int H, int W, int cus, int* out8) {
  if (N <= 0 || H <= 0 || W <= 0 || cus <= 0 || !out8) return DD_ERR_INVALID;
  int n_strips = 1, SO = W, FW = ((W + 31) / 32) * 32;
  if (W > 128) {
    for (n_strips = 2;; ++n_strips) {
      SO = (W + n_strips - 1) / n_strips;
      SO += SO & 1;
      FW = ((SO + 8 + 31) / 32) * 32;
      if (FW <= 128) break;
    }
  }
  const int R = FW == 32 ? 4 : FW == 64 ? 2 : 1;
  long best = -1; int bestBH = 0;
  for (int d = 1; d <= 16; ++d) {
    int BH = (H + d - 1) / d;
    BH = ((BH + 3) / 4) * 4;
    if (BH < 4) BH = 4;
    const int nb = (H + BH - 1) / BH;
    const long units = (long)N * n_strips * nb;
    const long G = units < cus ? units : cus;
    const long per = (units + G - 1) / G;
    const long cost = per * (BH + 8) + 4 * (R + 1);
    if (best < 0 || cost < best) { best = cost; bestBH = BH; }
  }
  const int nb = (H + bestBH - 1) / bestBH;
  out8[0] = FW; out8[1] = R; out8[2] = FW / 32; out8[3] = n_strips; out8[4] = SO; out8[5] = bestBH; out8[6] = nb; out8[7] = bestBH + 8;
  return DD_OK;
}

int dd_compose_stream_fwd_launch(const dd_compose_args* a, hipStream_t s) {
  CsP p;
  p.small = a->small; p.fine = a->fine; p.out = a->out;
  p.w_in = a->w_in; p.b_in = a->b_in; p.w_out = a->w_out; p.b_out = a->b_out;
  for (int l = 0; l < 4; ++l) { p.w_res[l] = a->w_res[l]; p.b_res[l] = a->b_res[l]; }
  bool save = a->save_wl != nullptr;
  for (int i = 0; i < 5; ++i) { p.save_act[i] = a->save_act[i]; p.ld_act[i] = a->ld_act[i]; save = save || a->save_act[i]; }
  p.save_wl = a->save_wl; p.ld_wl = a->ld_wl;
  p.save_netin = a->save_netin; p.ld_netin = a->ld_netin; save = save || a->save_netin;
  p.ld_small = a->ld_small; p.ld_fine = a->ld_fine; p.ld_out = a->ld_out;
  p.N = a->N; p.H = a->H; p.W = a->W;
  DD_REQUIRE((long)a->N * a->H * a->W < (1l << 31) / 64, "dd_compose_net_fwd: N * H * W = %ld pixels exceed the 32-bit offsets of the kernel", (long)a->N * a->H * a->W);
  const int cus = dd_device_cus();
  int g8[8];
  if (dd_compose_stream_plan(a->N, a->H, a->W, cus, g8) != DD_OK) return DD_ERR_INVALID;
  p.FW = g8[0]; p.R = g8[1]; p.TPR = g8[2]; p.n_strips = g8[3]; p.SO = g8[4]; p.BH = g8[5]; p.nb = g8[6]; p.VB = g8[7];
  p.units = a->N * p.n_strips * p.nb;
  const int grid = p.units < cus ? p.units : cus;
  const int lds = (12 * p.R + 10) * (p.FW + 2) * PIXB + AUX_BYTES;
#define CS_LAUNCH(T, SV, R1)                                                                          \
  do {                                                                                                \
    dd_allow_max_lds(reinterpret_cast<const void*>(compose_stream_fwd_kernel<T, SV, R1>));            \
    hipLaunchKernelGGL((compose_stream_fwd_kernel<T, SV, R1>), dim3(grid), dim3(1024), lds, s, p);    \
  } while (0)
#define CS_LAUNCH_R(T, SV) do { if (p.R == 1) CS_LAUNCH(T, SV, true); else CS_LAUNCH(T, SV, false); } while (0)
  if (a->dtype == DD_BF16) { if (save) CS_LAUNCH_R(bf16_t, true); else CS_LAUNCH_R(bf16_t, false); }
  else { if (save) CS_LAUNCH_R(f16_t, true); else CS_LAUNCH_R(f16_t, false); }
#undef CS_LAUNCH_R
#undef CS_LAUNCH
  DD_LAUNCH_CHECK();
  return DD_OK;
}
