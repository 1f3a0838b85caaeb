// Implicit-GEMM convolution on CDNA4 MFMA: forward and data-gradient of every conv-like layer of the denoiser.
//
// Reference seams replaced (file:line in /root/reference): tf.layers.conv2d 3x3 / 1x1 SAME (TensorFlow/UNet.py:29-31,
// Tiramisu.py:35-37,50-52,77-79, Architecture.py:238-243, MultiScalePrediction.py:64-66,73-75,88-90),
// tf.layers.conv2d_transpose 2x2/s2 (UNet.py:56-58) and their TF-autodiff input gradients (Training.py:701-702).
//
// Mapping.  A workgroup (4 waves) owns a 16x16 pixel tile of one image and NT*16 output channels.
//   GEMM: D[cout][pixel] = sum_{tap, cin} Wp[tap][cout][cin] * X[pixel (+) tap][cin]
//   "A" operand = packed weights (rows = cout), "B" operand = pixels (cols), so after the MFMA every lane holds
//   4 CONSECUTIVE output channels of one pixel -> packed 8/16-byte NHWC stores, fused bias/ReLU/mask epilogue.
// Data movement.  Per 128-byte K-slice of input channels (64 bf16 / 32 f32) the (16+2)^2 halo patch is staged in LDS
// ONCE and reused by all 9 taps (9x less L2->LDS traffic than im2col); the weight slab of one (tap, K-slice) is
// double-buffered so there is one barrier per tap.  LDS rows are 128 B with a 16-byte-slot XOR swizzle
// (dd_common.h: lds_off) => ds_read_b128 fragment reads are bank-conflict free.
// LDS: 40.5 KiB patch + 2 * NT*2 KiB weights (<= 72.5 KiB) => 2 workgroups per CU overlap staging with MFMA.
#include "dd_common.h"

namespace {

struct ConvP {
  const void* x; const void* wp; const float* bias; const void* res; const void* mask; void* y;
  int ldx, cin, k_pad, n_pad, ldres, ldmask, ldy, n;
  int B, H, W, taps, flags, nbias;
  int tiles_x, tiles_y, nblk;   // grid decomposition
  int kchunks;                  // number of 64-byte K chunks (k_pad*sizeof(T)/64)
  int hin, win;                 // input image size
  int hout, wout;               // output image size
};

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WB = NT * 16 * DD_LDS_ROW;  // bytes of one weight slab
  const bool halo = (p.taps == 9);
  const bool gather = (p.flags & DD_GATHER2X2) != 0;
  const int ph = halo ? DD_TILE + 2 : DD_TILE, pw = ph;
  char* patch = smem;
  char* wbuf = smem + ph * pw * DD_LDS_ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int nb = bid % p.nblk; bid /= p.nblk;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y; bid /= p.tiles_y;
  const int b = bid;
  const int y0 = ty * DD_TILE, x0 = tx * DD_TILE, n0 = nb * NT * 16;

  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.wp);
  const long img_base = (long)b * p.hin * p.win;
  const bool in_relu = (p.flags & DD_IN_RELU) != 0;

  f32x4_t acc[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  TileGeom g;
  g.ph = ph; g.pw = pw;
  g.oy = halo ? y0 - 1 : y0; g.ox = halo ? x0 - 1 : x0;
  g.sy = gather ? 2 : 1; g.ay = 0; g.ax = 0;
  g.lim_y = halo ? p.H + 1 : p.H; g.lim_x = halo ? p.W + 1 : p.W;
  g.min_y = halo ? -1 : 0; g.min_x = halo ? -1 : 0;
  g.hin = p.hin; g.win = p.win;

  constexpr int KC = DD_LDS_ROW / (int)sizeof(T);  // channels per K-slice
  const int nslices = (p.kchunks + 1) >> 1;
  const int q = lane >> 4, li = lane & 15;

  for (int s = 0; s < nslices; ++s) {
    const int nch = min(2, p.kchunks - 2 * s);
    __syncthreads();  // every wave is done with the previous slice's patch
    if (halo) stage_pixels<T>(patch, X, img_base, p.ldx, p.cin, s * KC, nch * 4, g, in_relu, tid, 256);
    for (int t = 0; t < p.taps; ++t) {
      if (!halo) {
        if (t > 0) __syncthreads();
        if (gather) { g.ay = t >> 1; g.ax = t & 1; }
        stage_pixels<T>(patch, X, img_base, p.ldx, p.cin, s * KC, nch * 4, g, in_relu, tid, 256);
      }
      char* wb = wbuf + (t & 1) * WB;
      {  // weight slab of (tap t, slice s): NT*16 rows of 128 B
        const int total = NT * 16 * 8;
        for (int i = tid; i < total; i += 256) {
          const int row = i >> 3, slot = i & 7;
          if (slot >= nch * 4) continue;
          const int ng = n0 + row;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (ng < p.n_pad)
            v = *reinterpret_cast<const uint4*>(Wp + ((long)t * p.n_pad + ng) * p.k_pad + s * KC + slot * Elem<T>::PER16);
          *reinterpret_cast<uint4*>(wb + lds_off(row, slot)) = v;
        }
      }
      __syncthreads();
      const int dy = halo ? t / 3 : 0, dx = halo ? t - dy * 3 : 0;
      for (int c = 0; c < nch; ++c) {
        const int slot = c * 4 + q;
        uint4 bf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pix = (wave * 4 + r + dy) * pw + li + dx;
          bf[r] = *reinterpret_cast<const uint4*>(patch + lds_off(pix, slot));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const uint4 af = *reinterpret_cast<const uint4*>(wb + lds_off(j * 16 + li, slot));
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][r] = mma16<T>(af, bf[r], acc[j][r]);
        }
      }
    }
  }

  // ---------------------------------------------------------------- epilogue
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ M = reinterpret_cast<const T*>(p.mask);
  const bool out_relu = (p.flags & DD_OUT_RELU) != 0, accum = (p.flags & DD_ACCUM) != 0;
  const bool pixshuf = (p.flags & DD_PIXSHUF) != 0;
  const int cout = pixshuf ? p.n / 4 : p.n;
  const int ox = x0 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oy = y0 + wave * 4 + r;
    if (oy >= p.H || ox >= p.W) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + q * 4;
      if (n >= p.n) continue;
      float v[4] = {acc[j][r][0], acc[j][r][1], acc[j][r][2], acc[j][r][3]};
      if (p.bias) {
        const int bi = pixshuf ? n % cout : n;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (bi + e < p.nbias) v[e] += p.bias[bi + e];
      }
      long pix; int ch;
      if (pixshuf) {
        const int ab = n / cout; ch = n - ab * cout;
        pix = ((long)b * p.hout + 2 * oy + (ab >> 1)) * p.wout + 2 * ox + (ab & 1);
      } else {
        ch = n;
        pix = ((long)b * p.hout + oy) * p.wout + ox;
      }
      if (R) { float t[4]; load4<T>(R + pix * p.ldres + ch, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      if (out_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (M) {
        float t[4]; load4<T>(M + pix * p.ldmask + ch, t);
        v[0] = t[0] > 0.f ? v[0] : 0.f; v[1] = t[1] > 0.f ? v[1] : 0.f; v[2] = t[2] > 0.f ? v[2] : 0.f; v[3] = t[3] > 0.f ? v[3] : 0.f;
      }
      T* dst = Y + pix * p.ldy + ch;
      if (accum) { float t[4]; load4<T>(dst, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
      store4<T>(dst, v);
    }
  }
}

template <typename T, int NT>
int launch(const ConvP& p, hipStream_t stream) {
  const int ph = (p.taps == 9) ? DD_TILE + 2 : DD_TILE;
  const size_t lds = (size_t)ph * ph * DD_LDS_ROW + 2 * (size_t)NT * 16 * DD_LDS_ROW;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  const long blocks = (long)p.B * p.tiles_y * p.tiles_x * p.nblk;
  hipLaunchKernelGGL((conv_igemm_kernel<T, NT>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
int dispatch(ConvP& p, hipStream_t stream) {
  const int tiles_n = p.n_pad / 16;
  const long spatial = (long)p.B * p.tiles_y * p.tiles_x;
  static const int cand[] = {8, 6, 4, 3, 2, 1};
  int nt = 1;
  for (int c : cand) if (tiles_n % c == 0) { nt = c; break; }
  // keep >= 2 workgroups per CU in flight when the pixel grid is small
  while (nt > 2 && nt % 2 == 0 && spatial * (tiles_n / nt) < 512) nt /= 2;
  p.nblk = tiles_n / nt;
  switch (nt) {
    case 8: return launch<T, 8>(p, stream);
    case 6: return launch<T, 6>(p, stream);
    case 4: return launch<T, 4>(p, stream);
    case 3: return launch<T, 3>(p, stream);
    case 2: return launch<T, 2>(p, stream);
    default: return launch<T, 1>(p, stream);
  }
}

}  // namespace

extern "C" int dd_conv_igemm(const dd_conv_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->x && a->wp && a->y, "dd_conv_igemm: null pointer");
  DD_REQUIRE(a->dtype == DD_F32 || a->dtype == DD_BF16, "dd_conv_igemm: bad dtype %d", a->dtype);
  const int esz = a->dtype == DD_F32 ? 4 : 2;
  const int per16 = 16 / esz;
  const bool gather = (a->flags & DD_GATHER2X2) != 0, pixshuf = (a->flags & DD_PIXSHUF) != 0;
  DD_REQUIRE(a->taps == 9 || a->taps == 1 || (a->taps == 4 && gather), "dd_conv_igemm: taps=%d unsupported", a->taps);
  DD_REQUIRE(!gather || a->taps == 4, "dd_conv_igemm: DD_GATHER2X2 needs taps=4");
  DD_REQUIRE(!pixshuf || a->taps == 1, "dd_conv_igemm: DD_PIXSHUF needs taps=1");
  DD_REQUIRE(a->cin > 0 && a->cin % per16 == 0 && a->ldx % per16 == 0, "dd_conv_igemm: cin=%d ldx=%d must be multiples of %d", a->cin, a->ldx, per16);
  DD_REQUIRE(a->k_pad >= a->cin && (a->k_pad * esz) % 64 == 0, "dd_conv_igemm: k_pad=%d invalid for cin=%d", a->k_pad, a->cin);
  DD_REQUIRE(a->n_pad % 16 == 0 && a->n > 0 && a->n <= a->n_pad && a->n % 4 == 0, "dd_conv_igemm: n=%d n_pad=%d invalid", a->n, a->n_pad);
  DD_REQUIRE(!pixshuf || (a->n % 16 == 0), "dd_conv_igemm: DD_PIXSHUF needs 4*cout %% 16 == 0");
  DD_REQUIRE(a->ldy % 4 == 0 && (!a->res || a->ldres % 4 == 0) && (!a->mask || a->ldmask % 4 == 0), "dd_conv_igemm: ld must be multiple of 4");
  DD_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "dd_conv_igemm: empty grid");
  DD_REQUIRE(((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->wp % 16) == 0 && ((uintptr_t)a->y % 16) == 0, "dd_conv_igemm: pointers must be 16-byte aligned");

  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.res = a->res; p.mask = a->mask; p.y = a->y;
  p.ldx = a->ldx; p.cin = a->cin; p.k_pad = a->k_pad; p.n_pad = a->n_pad; p.ldres = a->ldres; p.ldmask = a->ldmask;
  p.ldy = a->ldy; p.n = a->n; p.B = a->B; p.H = a->H; p.W = a->W; p.taps = a->taps; p.flags = a->flags; p.nbias = a->bias ? a->nbias : 0;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE); p.nblk = 1;
  p.kchunks = a->k_pad * esz / 64;
  p.hin = gather ? 2 * a->H : a->H; p.win = gather ? 2 * a->W : a->W;
  p.hout = pixshuf ? 2 * a->H : a->H; p.wout = pixshuf ? 2 * a->W : a->W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return a->dtype == DD_F32 ? dispatch<float>(p, s) : dispatch<bf16_t>(p, s);
}
