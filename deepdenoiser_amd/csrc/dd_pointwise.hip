// HBM-bound kernels of the denoiser hot path (gfx950): weight packing, pooling, input assembly, kernel-prediction
// filter apply, multiscale compose blend, inverse standardization, fused loss head, Adam, stitch.
// Reference seams are cited per entry point in include/dd_hip.h.
#include <stdarg.h>

#include "dd_common.h"

// ------------------------------------------------------------------------------------------------ errors (dd_version: csrc/dd_version.hip)
static thread_local char g_err[512] = "";
#include <mutex>
#include <set>
#include <utility>

namespace {
std::mutex g_dev_mutex;
int g_dev_cus[64];
std::set<std::pair<int, const void*>> g_dev_lds;
int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev < 0 || dev >= 64 ? 0 : dev;
}
}  // namespace

int dd_device_cus() {
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  if (g_dev_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_dev_cus[dev] = n;
  }
  return g_dev_cus[dev];
}

void dd_allow_max_lds(const void* kernel, int bytes) {
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  if (g_dev_lds.insert(std::make_pair(dev, kernel)).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void dd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dd_last_error(void) { return g_err; }

// CRC-32C, slicing-by-8 (host only).
namespace {
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
}  // namespace
extern "C" int dd_crc32c(const void* data, size_t n, uint32_t crc, uint32_t* out) {
  DD_REQUIRE((data || !n) && out, "dd_crc32c: null pointer");
  static const Crc32cTables tab;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = tab.t[7][lo & 0xFF] ^ tab.t[6][(lo >> 8) & 0xFF] ^ tab.t[5][(lo >> 16) & 0xFF] ^ tab.t[4][lo >> 24] ^
        tab.t[3][hi & 0xFF] ^ tab.t[2][(hi >> 8) & 0xFF] ^ tab.t[1][(hi >> 16) & 0xFF] ^ tab.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = tab.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  *out = ~c;
  return 0;
}

#define S(stream) reinterpret_cast<hipStream_t>(stream)
static inline unsigned grid_for(long n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// ------------------------------------------------------------------------------------------------ pack weights
template <typename T>
__global__ void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int taps, int n, int k, int n_pad, int k_pad,
                            long s_tap, long s_n, long s_k, int flip) {
  const long total = (long)taps * n_pad * k_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % k_pad);
    const long r = i / k_pad;
    const int nn = (int)(r % n_pad);
    const int t = (int)(r / n_pad);
    float v = 0.f;
    if (nn < n && kk < k) v = src[(flip ? taps - 1 - t : t) * s_tap + nn * s_n + kk * s_k];
    dst[i] = Elem<T>::from_f32(v);
  }
}
extern "C" int dd_pack_weights(const float* src, void* dst, int dtype, int taps, int n, int k, int n_pad, int k_pad,
                               long s_tap, long s_n, long s_k, int tap_flip, dd_stream stream) {
  DD_REQUIRE(src && dst && taps > 0 && n > 0 && k > 0 && n_pad >= n && k_pad >= k, "dd_pack_weights: bad arguments");
  const long total = (long)taps * n_pad * k_pad;
  const unsigned g = min(grid_for(total), 2048u);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(pack_kernel<T>, dim3(g), dim3(256), 0, S(stream), src, (T*)dst, taps, n, k, n_pad, k_pad, s_tap, s_n, s_k, tap_flip));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void pack_batched_kernel(const dd_pack_desc* __restrict__ table) {
  const dd_pack_desc d = table[blockIdx.y];
  T* dst = reinterpret_cast<T*>(d.dst);
  const long total = (long)d.taps * d.n_pad * d.k_pad;
  const long ld = d.dst_ld ? d.dst_ld : d.k_pad, tstride = d.dst_tap_stride ? d.dst_tap_stride : (long)d.n_pad * d.k_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % d.k_pad);
    const long r = i / d.k_pad;
    const int nn = (int)(r % d.n_pad);
    const int t = (int)(r / d.n_pad);
    float v = 0.f;
    if (nn < d.n && kk < d.k) v = d.src[(d.tap_flip ? d.taps - 1 - t : t) * d.s_tap + nn * d.s_n + kk * d.s_k];
    dst[t * tstride + nn * ld + kk] = Elem<T>::from_f32(v);
  }
}
extern "C" int dd_pack_weights_batched(const dd_pack_desc* table, int n_layers, int dtype, dd_stream stream) {
  DD_REQUIRE(table && n_layers > 0, "dd_pack_weights_batched: bad arguments");
  const dim3 g(64, (unsigned)n_layers);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(pack_batched_kernel<T>, g, dim3(256), 0, S(stream), table));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ column sums
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, int ld, int c, long rows, float* __restrict__ out) {
  __shared__ float partial[256];
  const int cb = blockIdx.y * 64;            // 64 channels per block column
  const int ci = threadIdx.x & 63, rl = threadIdx.x >> 6;  // 4 row lanes
  const int ch = cb + ci;
  float s = 0.f;
  if (ch < c)
    for (long r = (long)blockIdx.x * 4 + rl; r < rows; r += (long)gridDim.x * 4) s += ld1<T>(x + r * ld + ch);
  partial[threadIdx.x] = s;
  __syncthreads();
  dd_det_wait();      // (DD_DETERMINISTIC=1: workgroups add in index order, dd_common.h)
  if (rl == 0 && ch < c) atomicAdd(out + ch, partial[ci] + partial[64 + ci] + partial[128 + ci] + partial[192 + ci]);
  dd_det_end();
}
extern "C" int dd_colsum_segments(const void* x, int ld, int c, long rows_per_segment, int n_segments, float* out, int out_ld, const int* out_row,
                                  int dtype, dd_stream stream);
extern "C" int dd_colsum(const void* x, int ld, int c, long rows, float* out, int dtype, dd_stream stream) {
  DD_REQUIRE(x && out && c > 0 && rows > 0, "dd_colsum: bad arguments");
  {      // narrow tensors (bias gradients of 16 ... 64-channel layers): the vectorised kernel of dd_colsum_segments, one segment (round 5: the
         // per-channel-lane kernel below ran 16 - 25 live lanes of 64 with 2-byte loads, 42 us per bias gradient of the light Tiramisu)
    const int per16 = dtype == DD_F32 ? 4 : 8, esz = dtype == DD_F32 ? 4 : 2;
    int nv = 1;
    while (nv * per16 < c) nv *= 2;
    static const bool vec_on = [] { const char* e = getenv("DD_COLSUM_VEC"); return !(e && e[0] == '0'); }();
    if (vec_on && dd_dtype_ok(dtype) && nv <= 8 && (c + per16 - 1) / per16 * per16 <= ld && ld % per16 == 0 && ((uintptr_t)x % 16) == 0 && (ld * esz) % 16 == 0) {
      const int row0 = 0;
      return dd_colsum_segments(x, ld, c, rows, 1, out, c, &row0, dtype, stream);
    }
  }
  dim3 g((unsigned)min((rows + 3) / 4, 1024L), (unsigned)((c + 63) / 64));
  dd_det_sync();
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_kernel<T>, g, dim3(256), 0, S(stream), (const T*)x, ld, c, rows, out));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ max pooling (TF SAME)
__host__ __device__ inline void same_pad(int size, int k, int s, int* out, int* before) {
  const int o = (size + s - 1) / s;
  int total = (o - 1) * s + k - size;
  if (total < 0) total = 0;
  *out = o; *before = total / 2;
}

// One thread = one output pixel x 16 bytes of channels (8 bf16 / 4 f32): 16-byte loads and stores, the argmax plane as 8 / 4 bytes.
// relu_mask: the pooled tensor is a ReLU output whose backward mask (x > 0) is fused here -- a window whose maximum is not positive
// records index 255 ("no gradient"), so the backward pass never has to read x again (42 % of its HBM traffic).
template <typename T> struct Vec16 { static constexpr int N = Elem<T>::PER16; };
template <typename T> __device__ __forceinline__ void vload(const T* p, float (&v)[Elem<T>::PER16]);
template <> __device__ __forceinline__ void vload<float>(const float* p, float (&v)[4]) { load4<float>(p, v); }
template <> __device__ __forceinline__ void vload<bf16_t>(const bf16_t* p, float (&v)[8]) { unpack8(*reinterpret_cast<const uint4*>(p), v); }
template <> __device__ __forceinline__ void vload<f16_t>(const f16_t* p, float (&v)[8]) { unpack8t<f16_t>(*reinterpret_cast<const uint4*>(p), v); }
template <typename T> __device__ __forceinline__ void vstore(T* p, const float (&v)[Elem<T>::PER16]);
template <> __device__ __forceinline__ void vstore<f16_t>(f16_t* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack8t<f16_t>(v); }
template <> __device__ __forceinline__ void vstore<float>(float* p, const float (&v)[4]) { store4<float>(p, v); }
template <> __device__ __forceinline__ void vstore<bf16_t>(bf16_t* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack8(v); }

// Column sums of n_seg consecutive row segments in ONE launch: out[out_row[s]][ch] += sum over the rows of segment s.  The embedding-row
// gradients of FeatureFlags.feature_flags (FeatureFlags.py:57-67: one row of the embedding matrix per tuple, tiled over the tuple's images) were
// 17 dd_colsum launches of 8 channels each -- 8 live lanes of 64, 2-byte loads: 0.4 ms of the ArchitectureExample.json step; here a thread
// loads 16-byte vectors (NV = vectors per row, a power of two <= 8), the lanes of a wave that hold the same vector are summed by xor-shuffles,
// the four waves through LDS, and one set of atomics leaves per workgroup.
struct ColsumSegP { const void* x; float* out; long rows; int ld, c, nv, out_ld; int out_row[64]; };
template <typename T>
__global__ __launch_bounds__(256) void colsum_segments_kernel(const ColsumSegP p) {
  constexpr int N = Elem<T>::PER16;
  __shared__ float red[4][8 * N];
  const int seg = blockIdx.y, nv = p.nv, v = threadIdx.x & (nv - 1);
  const T* base = reinterpret_cast<const T*>(p.x) + (long)seg * p.rows * p.ld + v * N;
  float s[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = 0.f;
  const long rstep = (long)gridDim.x * (256 / nv);
  // (nv is a power of two; vectors past the channel count -- 24 channels are 3 vectors of 4 -- are not loaded: the caller may hand over a
  //  channel VIEW whose row ends with its last real vector, and the last row's surplus vector would lie outside the allocation: ADVICE r5)
  const bool live = v * N < p.c;
  for (long r = (long)blockIdx.x * (256 / nv) + threadIdx.x / nv; live && r < p.rows; r += rstep) {
    float t[N];
    vload<T>(base + r * p.ld, t);
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] += t[e];
  }
  for (int o = 32; o >= nv; o >>= 1)
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] += __shfl_xor(s[e], o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < nv)
#pragma unroll
    for (int e = 0; e < N; ++e) red[wv][lane * N + e] = s[e];
  __syncthreads();
  dd_det_wait();
  if ((int)threadIdx.x < nv * N && (int)threadIdx.x < p.c)
    atomicAdd(p.out + (long)p.out_row[seg] * p.out_ld + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  dd_det_end();
}
extern "C" int dd_colsum_segments(const void* x, int ld, int c, long rows_per_segment, int n_segments, float* out, int out_ld, const int* out_row,
                                  int dtype, dd_stream stream) {
  DD_REQUIRE(x && out && out_row && c > 0 && rows_per_segment > 0 && n_segments > 0 && n_segments <= 64 && out_ld >= c, "dd_colsum_segments: bad arguments");
  DD_REQUIRE(dd_dtype_ok(dtype), "dd_colsum_segments: bad dtype %d", dtype);
  const int per16 = dtype == DD_F32 ? 4 : 8, esz = dtype == DD_F32 ? 4 : 2;
  int nv = 1;
  while (nv * per16 < c) nv *= 2;
  DD_REQUIRE(nv <= 8 && (c + per16 - 1) / per16 * per16 <= ld && ld % per16 == 0 && ((uintptr_t)x % 16) == 0 && (ld * esz) % 16 == 0,
             "dd_colsum_segments: c=%d ld=%d (at most %d channels, rows of whole 16-byte vectors that cover them)", c, ld, 8 * per16);
  ColsumSegP p;
  p.x = x; p.out = out; p.rows = rows_per_segment; p.ld = ld; p.c = c; p.nv = nv; p.out_ld = out_ld;
  for (int s = 0; s < n_segments; ++s) { DD_REQUIRE(out_row[s] >= 0, "dd_colsum_segments: negative output row"); p.out_row[s] = out_row[s]; }
  long bx = (rows_per_segment + (256 / nv) * 8 - 1) / ((256 / nv) * 8);      // >= 8 rows per thread
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  dd_det_sync();
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(colsum_segments_kernel<T>, dim3((unsigned)bx, (unsigned)n_segments), dim3(256), 0, S(stream), p));
  DD_LAUNCH_CHECK();
  return DD_OK;
}


template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, uint8_t* __restrict__ idx,
                                   int C, int B, int H, int W, int OH, int OW, int pool, int stride, int pby, int pbx, int relu_mask) {
  constexpr int N = Elem<T>::PER16;
  const int cg = C / N;
  const long total = (long)B * OH * OW * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * N;
  long r = i / cg;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH);
  const int b = (int)(r / OH);
  float best[N];
  int arg[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { best[e] = -INFINITY; arg[e] = 0; }
  for (int a = 0; a < pool; ++a)
    for (int bb = 0; bb < pool; ++bb) {
      const int sy = oy * stride + a - pby, sx = ox * stride + bb - pbx;
      if (sy < 0 || sx < 0 || sy >= H || sx >= W) continue;
      float v[N];
      vload<T>(x + (((long)b * H + sy) * W + sx) * ldx + c, v);
#pragma unroll
      for (int e = 0; e < N; ++e)
        if (v[e] > best[e]) { best[e] = v[e]; arg[e] = a * pool + bb; }
    }
  const long opix = ((long)b * OH + oy) * OW + ox;
  vstore<T>(y + opix * ldy + c, best);
  uint32_t packed[N / 4];
#pragma unroll
  for (int w = 0; w < N / 4; ++w) {
    packed[w] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = (relu_mask && !(best[w * 4 + e] > 0.f)) ? 255 : arg[w * 4 + e];
      packed[w] |= (uint32_t)k << (8 * e);
    }
  }
  if (idx == nullptr) return;      // (inference: nobody reads the argmax plane)
  if (N == 8) *reinterpret_cast<uint2*>(idx + opix * C + c) = make_uint2(packed[0], packed[N / 4 - 1]);
  else *reinterpret_cast<uint32_t*>(idx + opix * C + c) = packed[0];
}
// Stride-2 specialisation of the BACKWARD (POOL = 3: the U-Net's pooling, POOL = 2: Tiramisu's).  The generic kernel above loops over a RUN-TIME
// window: hipcc cannot unroll it, so a thread's loads are issued one after the other, and its index arithmetic is three 64-bit divisions per
// thread -- it ran at 2.4 - 2.9 TB/s of traffic.  Here the window is a template constant (every load of a thread is in flight at once,
// out-of-image windows read a clamped address and count as "no gradient"), the index arithmetic is 32-bit, and a thread owns a 2 x 2 block of input
// pixels: with stride 2 the four pixels share their (at most four) windows, so a block costs 4 (index, gradient) loads instead of the 9 the
// per-pixel kernel issues for the same four pixels.  Measured (tools/maxpool_bench.py, bf16, accumulate): 128 x 128 x 128 x 64 217 -> 136 us
// (4.7 TB/s), 128 x 64 x 64 x 96 61 -> 47 us.  The same treatment of the FORWARD (nine loads in flight, 32-bit indices) measured SLOWER than the
// generic kernel (84 -> 95 us: it is bound by the 2.25x re-reads through L2, not by load latency) and is not kept.
// (the 32-bit offsets of the specialisations: every tensor involved must have fewer than 2^31 elements)
static bool pool_s2_ok(int pool, int stride, int B, int H, int W, int ld_big) {
  static const bool generic = getenv("DD_MAXPOOL_GENERIC") && getenv("DD_MAXPOOL_GENERIC")[0] == '1';      // (A/B switch)
  return !generic && stride == 2 && (pool == 2 || pool == 3) && (double)B * H * W * ld_big < 2147483648.0;
}
extern "C" int dd_maxpool_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* idx, int C, int B, int H, int W,
                              int pool, int stride, int relu_mask, int dtype, dd_stream stream) {
  const int per16 = dtype == DD_F32 ? 4 : 8;
  DD_REQUIRE(dd_dtype_ok(dtype), "bad dtype %d", dtype);
  DD_REQUIRE(x && y && C % per16 == 0 && ldx % per16 == 0 && ldy % per16 == 0, "dd_maxpool_fwd: C, ld must be multiples of %d", per16);      // (idx may be NULL: forward only)
  DD_REQUIRE(pool * pool < 255, "dd_maxpool_fwd: pool=%d too large", pool);
  int OH, OW, pby, pbx;
  same_pad(H, pool, stride, &OH, &pby); same_pad(W, pool, stride, &OW, &pbx);
  const long total = (long)B * OH * OW * (C / per16);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)x, ldx, (T*)y, ldy, idx, C, B, H, W, OH, OW, pool, stride, pby, pbx, relu_mask));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int lddx,
                                   const T* __restrict__ mask, int ldmask, int C, int B, int H, int W, int OH, int OW,
                                   int pool, int stride, int pby, int pbx, int accumulate) {
  constexpr int N = Elem<T>::PER16;
  const int cg = C / N;
  const long total = (long)B * H * W * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * N;
  long r = i / cg;
  const int x = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  float g[N];
#pragma unroll
  for (int e = 0; e < N; ++e) g[e] = 0.f;
  // windows (oy, ox) that contain (y, x): oy*stride - pby <= y <= oy*stride - pby + pool - 1
  const int oy_hi = (y + pby) / stride, ox_hi = (x + pbx) / stride;
  for (int oy = oy_hi; oy >= 0 && oy * stride - pby + pool - 1 >= y; --oy) {
    if (oy >= OH) continue;
    for (int ox = ox_hi; ox >= 0 && ox * stride - pbx + pool - 1 >= x; --ox) {
      if (ox >= OW) continue;
      const uint32_t k = (uint32_t)((y - (oy * stride - pby)) * pool + (x - (ox * stride - pbx)));
      const long opix = ((long)b * OH + oy) * OW + ox;
      uint32_t a[N / 4];
      if (N == 8) { const uint2 t = *reinterpret_cast<const uint2*>(idx + opix * C + c); a[0] = t.x; a[N / 4 - 1] = t.y; }
      else a[0] = *reinterpret_cast<const uint32_t*>(idx + opix * C + c);
      float v[N];
      vload<T>(dy + opix * lddy + c, v);
#pragma unroll
      for (int e = 0; e < N; ++e)
        if (((a[e >> 2] >> (8 * (e & 3))) & 255u) == k) g[e] += v[e];
    }
  }
  const long pix = ((long)b * H + y) * W + x;
  if (mask) {
    float m[N];
    vload<T>(mask + pix * ldmask + c, m);
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
  }
  T* dst = dx + pix * lddx + c;
  if (accumulate) {
    float o[N];
    vload<T>(dst, o);
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] += o[e];
  }
  vstore<T>(dst, g);
}
// One thread = a 2 x 2 block of input pixels (rows 2k - pby + {0, 1}, columns 2j - pbx + {0, 1}) x 16 bytes of channels.  In padded coordinates
// the even row 2k lies in window k (offset 0) and, for POOL = 3, in window k - 1 (offset 2); the odd row only in window k (offset 1).
template <typename T, int POOL>
__global__ __launch_bounds__(256) void maxpool_bwd_s2_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int lddx,
                                                             const T* __restrict__ mask, int ldmask, int C, unsigned total, int H, int W, int OH, int OW,
                                                             int KB, int JB, int pby, int pbx, int accumulate) {
  constexpr int N = Elem<T>::PER16, NWIN = POOL == 3 ? 2 : 1;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned cg = (unsigned)C / N;
  unsigned r = i / cg;
  const int c = (int)(i - r * cg) * N;
  const int j = (int)(r % (unsigned)JB); r /= (unsigned)JB;
  const int k = (int)(r % (unsigned)KB);
  const int b = (int)(r / (unsigned)KB);
  float g[2][2][N];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < N; ++e) g[q >> 1][q & 1][e] = 0.f;
  const size_t ob = (size_t)b * OH * OW;
  uint32_t am[NWIN * NWIN][N / 4];
  float v[NWIN * NWIN][N];
#pragma unroll
  for (int wy = 0; wy < NWIN; ++wy)
#pragma unroll
    for (int wx = 0; wx < NWIN; ++wx) {
      const int oy = k - wy, ox = j - wx, w = wy * NWIN + wx;
      const bool ok = (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
      const size_t opix = ob + (ok ? (unsigned)(oy * OW + ox) : 0u);
      if (N == 8) { const uint2 t = *reinterpret_cast<const uint2*>(idx + opix * C + c); am[w][0] = t.x; am[w][N / 4 - 1] = t.y; }
      else am[w][0] = *reinterpret_cast<const uint32_t*>(idx + opix * C + c);
      vload<T>(dy + opix * lddy + c, v[w]);
      if (!ok) {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) am[w][q] = 0xFFFFFFFFu;      // 255: "no gradient"
      }
    }
  const int y0 = 2 * k - pby, x0 = 2 * j - pbx;
  float old[2][2][N], mk[2][2][N];
#pragma unroll
  for (int ay = 0; ay < 2; ++ay)
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
      const bool ok = (unsigned)(y0 + ay) < (unsigned)H && (unsigned)(x0 + ax) < (unsigned)W;
      const size_t pix = (size_t)b * H * W + (ok ? (unsigned)((y0 + ay) * W + x0 + ax) : 0u);
      if (accumulate) vload<T>(dx + pix * lddx + c, old[ay][ax]);
      if (mask) vload<T>(mask + pix * ldmask + c, mk[ay][ax]);
    }
#pragma unroll
  for (int wy = 0; wy < NWIN; ++wy)
#pragma unroll
    for (int wx = 0; wx < NWIN; ++wx)
#pragma unroll
      for (int ay = 0; ay < (wy ? 1 : 2); ++ay)
#pragma unroll
        for (int ax = 0; ax < (wx ? 1 : 2); ++ax) {
          const uint32_t kk = (uint32_t)((wy ? 2 : ay) * POOL + (wx ? 2 : ax));
          const int w = wy * NWIN + wx;
#pragma unroll
          for (int e = 0; e < N; ++e)
            if (((am[w][e >> 2] >> (8 * (e & 3))) & 255u) == kk) g[ay][ax][e] += v[w][e];
        }
#pragma unroll
  for (int ay = 0; ay < 2; ++ay)
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
      if (!((unsigned)(y0 + ay) < (unsigned)H && (unsigned)(x0 + ax) < (unsigned)W)) continue;
      const size_t pix = (size_t)b * H * W + (unsigned)((y0 + ay) * W + x0 + ax);
      if (mask) {
#pragma unroll
        for (int e = 0; e < N; ++e) g[ay][ax][e] = mk[ay][ax][e] > 0.f ? g[ay][ax][e] : 0.f;
      }
      if (accumulate) {
#pragma unroll
        for (int e = 0; e < N; ++e) g[ay][ax][e] += old[ay][ax][e];
      }
      vstore<T>(dx + pix * lddx + c, g[ay][ax]);
    }
}
extern "C" int dd_maxpool_bwd(const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, const void* mask, int ldmask,
                              int C, int B, int H, int W, int pool, int stride, int accumulate, int dtype, dd_stream stream) {
  const int per16 = dtype == DD_F32 ? 4 : 8;
  DD_REQUIRE(dd_dtype_ok(dtype), "bad dtype %d", dtype);
  DD_REQUIRE(dy && idx && dx && C % per16 == 0 && lddy % per16 == 0 && lddx % per16 == 0, "dd_maxpool_bwd: C, ld must be multiples of %d", per16);
  int OH, OW, pby, pbx;
  same_pad(H, pool, stride, &OH, &pby); same_pad(W, pool, stride, &OW, &pbx);
  if (pool_s2_ok(pool, stride, B, H, W, lddx > C ? lddx : C)) {
    const int KB = (H + pby + 1) / 2, JB = (W + pbx + 1) / 2;      // blocks of two padded rows / columns that hold an image pixel
    const long nt = (long)B * KB * JB * (C / per16);
    if (pool == 3) { DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_s2_kernel<T, 3>), dim3(grid_for(nt)), dim3(256), 0, S(stream), (const T*)dy, lddy, idx, (T*)dx, lddx, (const T*)mask, ldmask, C, (unsigned)nt, H, W, OH, OW, KB, JB, pby, pbx, accumulate)); }
    else { DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((maxpool_bwd_s2_kernel<T, 2>), dim3(grid_for(nt)), dim3(256), 0, S(stream), (const T*)dy, lddy, idx, (T*)dx, lddx, (const T*)mask, ldmask, C, (unsigned)nt, H, W, OH, OW, KB, JB, pby, pbx, accumulate)); }
    DD_LAUNCH_CHECK();
    return DD_OK;
  }
  const long total = (long)B * H * W * (C / per16);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)dy, lddy, idx, (T*)dx, lddx, (const T*)mask, ldmask, C, B, H, W, OH, OW, pool, stride, pby, pbx, accumulate));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ average pooling (fp32)
__global__ void avgpool_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int C, int B, int H, int W, int f) {
  const int OH = H / f, OW = W / f;
  const long total = (long)B * OH * OW * C;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long r = i / C;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH);
  const int b = (int)(r / OH);
  float s = 0.f;
  for (int a = 0; a < f; ++a)
    for (int bb = 0; bb < f; ++bb) s += x[(((long)b * H + oy * f + a) * W + ox * f + bb) * ldx + c];
  y[(((long)b * OH + oy) * OW + ox) * ldy + c] = s / (float)(f * f);
}
extern "C" int dd_avgpool(const float* x, int ldx, float* y, int ldy, int C, int B, int H, int W, int f, dd_stream stream) {
  DD_REQUIRE(x && y && f >= 1 && H % f == 0 && W % f == 0, "dd_avgpool: H=%d W=%d must be divisible by f=%d", H, W, f);
  const long total = (long)B * (H / f) * (W / f) * C;
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), x, ldx, y, ldy, C, B, H, W, f);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ input pipeline
__device__ __forceinline__ int sym_index(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }
__device__ __forceinline__ float signed_log1p(float v) { return v > 0.f ? log1pf(v) : (v < 0.f ? -log1pf(-v) : 0.f); }
__device__ __forceinline__ float standardize(float v, const dd_feature_params& fp) {
  if (fp.use_log1p) v = signed_log1p(v);
  return (v - fp.mean) * fp.inv_std;
}

// One workgroup = one 16x16 pixel tile.  The 18x18 haloed tile is staged in LDS ONCE (symmetric border indices resolved while staging),
// standardised there, and the 3x3 (or plus-shaped) local variance is taken from LDS: one signed_log1p per pixel and channel instead of
// up to nine, and no 27 scalar global loads per pixel.
__global__ __launch_bounds__(256) void prepare_feature_kernel(const float* __restrict__ src, int cs, float* __restrict__ dst, int ldd,
                                                              const dd_feature_params fp, int B, int H, int W, int tiles_x, int tiles_y) {
  constexpr int TP = 16, HP = TP + 2;
  __shared__ float s_std[3][HP][HP + 1];     // standardised value
  __shared__ float s_var[3][HP][HP + 1];     // the variance source: raw (variance_before) or standardised
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * TP, x0 = tx * TP;
  const float* img = src + (long)b * H * W * cs;
  for (int k = threadIdx.x; k < HP * HP; k += 256) {
    const int py = k / HP, px = k - py * HP;
    // mirror exactly like the per-pixel formula did: sym_index of the neighbour coordinate (coordinates past a ragged tile are clamped, unused)
    const int gy = sym_index(min(y0 - 1 + py, H), H), gx = sym_index(min(x0 - 1 + px, W), W);
    for (int c = 0; c < cs; ++c) {
      const float v = img[((long)gy * W + gx) * cs + c];
      const float sv = standardize(v, fp);
      s_std[c][py][px] = sv;
      s_var[c][py][px] = fp.variance_before ? v : sv;
    }
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  const int y = y0 + ly, x = x0 + lx;
  if (y >= H || x >= W) return;
  float s[3];
  for (int c = 0; c < cs; ++c) s[c] = s_std[c][ly + 1][lx + 1];
  const long i = ((long)b * H + y) * W + x;
  float* o = dst + i * ldd;
  const bool one_store = ldd == 4 && (!fp.use_variance || fp.compress);      // the whole pixel record is one float4
  float4 rec = make_float4(s[0], cs == 3 ? s[1] : s[0], cs == 3 ? s[2] : s[0], 0.f);
  if (!one_store) { o[0] = rec.x; o[1] = rec.y; o[2] = rec.z; }
  if (!fp.use_variance) {
    if (one_store) *reinterpret_cast<float4*>(o) = rec;
    return;
  }
  float var_acc = 0.f;
  for (int c = 0; c < cs; ++c) {
    float sum = 0.f, sumsq = 0.f;
    int cnt = 0;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
      for (int bb = -1; bb <= 1; ++bb) {
        if (fp.mode_neighbor && a != 0 && bb != 0) continue;
        const float v = s_var[c][ly + 1 + a][lx + 1 + bb];
        sum += v; sumsq += v * v; ++cnt;
      }
    const float mean = sum / cnt, meansq = sumsq / cnt;
    float var = meansq - mean * mean;
    if (fp.relative) var = var / fmaxf(mean * mean, fp.epsilon);
    if (fp.compress) var_acc += var; else o[3 + c] = var;
  }
  if (fp.compress) {
    if (one_store) { rec.w = var_acc / cs; *reinterpret_cast<float4*>(o) = rec; }
    else o[3] = var_acc / cs;
  }
}
extern "C" int dd_prepare_feature(const float* src, int cs, float* dst, int ldd, const dd_feature_params* fp, int B, int H, int W, dd_stream stream) {
  DD_REQUIRE(src && dst && fp && (cs == 1 || cs == 3), "dd_prepare_feature: cs must be 1 or 3");
  const int nv = fp->use_variance ? (fp->compress ? 1 : cs) : 0;
  DD_REQUIRE(ldd >= 3 + nv, "dd_prepare_feature: ldd=%d too small for %d channels", ldd, 3 + nv);
  const int tiles_x = dd_ceil_div(W, 16), tiles_y = dd_ceil_div(H, 16);
  hipLaunchKernelGGL(prepare_feature_kernel, dim3((unsigned)((long)B * tiles_x * tiles_y)), dim3(256), 0, S(stream), src, cs, dst, ldd, *fp, B, H, W, tiles_x, tiles_y);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// One thread = one pixel x 16 bytes of destination channels (8 bf16 / 4 f32): a wave writes 1 KiB of contiguous network input per
// store (the first version wrote every channel of a pixel with its own 2-byte store).  The tuple's entry table is tiny and read through
// the scalar cache; each destination channel takes its value from the one entry that covers it.
template <typename T>
__global__ void gather_input_kernel(const dd_gather_entry* __restrict__ table, int n_tuples, int n_entries, T* __restrict__ dst, int ld,
                                    int c_pad, int B, long hw) {
  constexpr int N = Elem<T>::PER16;
  const int groups = c_pad / N;
  const long total = (long)n_tuples * B * hw * groups;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c0 = (int)(i % groups) * N;
  const long p = i / groups;
  const long pix = p % hw;
  const long tb = p / hw;
  const int b = (int)(tb % B), t = (int)(tb / B);
  float v[N];
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = 0.f;
  for (int e = 0; e < n_entries; ++e) {
    const dd_gather_entry en = table[t * n_entries + e];
    if (en.nch <= 0 || en.dst_ch >= c0 + N || en.dst_ch + en.nch <= c0) continue;
    const float* s = en.src + ((long)b * en.batch_stride_pixels + pix) * en.pixel_stride;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int k = c0 + j - en.dst_ch;
      if (k >= 0 && k < en.nch) v[j] = s[k];
    }
  }
  vstore<T>(dst + p * ld + c0, v);
}
extern "C" int dd_gather_input(const dd_gather_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                               int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(table && dst && n_tuples > 0 && n_entries > 0 && c_pad <= ld, "dd_gather_input: bad arguments");
  const int per16 = dtype == DD_F32 ? 4 : 8;
  DD_REQUIRE(dd_dtype_ok(dtype), "bad dtype %d", dtype);
  DD_REQUIRE(c_pad % per16 == 0 && ld % per16 == 0, "dd_gather_input: c_pad=%d and ld=%d must be multiples of %d", c_pad, ld, per16);
  const long total = (long)n_tuples * B * H * W * (c_pad / per16);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(gather_input_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), table, n_tuples, n_entries, (T*)dst, ld, c_pad, B, (long)H * W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ fused input assembly
// dd_prepare_feature (one launch per render pass, fp32 staging planes) + dd_gather_input (one more pass over them) as ONE launch: a workgroup
// owns a 16x16 tile of one (tuple, tile) image and walks the tuple's entry table -- per feature entry the 18x18 haloed source tile is staged
// and standardised in LDS, the local variance taken from it, and the entry's channels written straight into the network input.
// HBM sees the raw passes once (12 B per pixel and pass) and the network input once (c_pad storage elements per pixel); the staging planes
// (16 B written + 16 B read per pixel and pass) are gone, except the standardised source of the passes the kernel-prediction head filters
// (std_out != NULL), which it needs anyway.
// Round 4 (SQ counters: the kernel was VALU-bound, 3 300 vector instructions per wave and tile of eight passes, not memory-bound):
//   * a pass's 4 channels enter the LDS image of the network-input tile as two dword stores (were four 2-byte stores), and only the padding
//     channels are zeroed.  (Writing them straight to HBM instead, one 8-byte store per pixel and pass, measured SLOWER: 224 -> 247 us, the
//     partial lines cost more than the LDS image and its coalesced copy-out);
//   * the three fp32 divisions per channel of the variance (sum / count twice, var / max(mean^2, eps)) are multiplications by 1 / count and
//     one v_rcp_f32 (<= 1 ulp; the relative variance is a network INPUT, the f32 parity gates hold: tests/test_gpu_ops.py);
//   * sign(v) * log1p(|v|) instead of two log1p calls selected by sign.
// (4-byte aligned destination: dword stores)
template <typename T> __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) { uint32_t* q = reinterpret_cast<uint32_t*>(p); q[0] = pack2<bf16_t>(a, b); q[1] = pack2<bf16_t>(c, d); }
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, float a, float b, float c, float d) { uint32_t* q = reinterpret_cast<uint32_t*>(p); q[0] = pack2<f16_t>(a, b); q[1] = pack2<f16_t>(c, d); }

template <typename T>
__global__ __launch_bounds__(256) void assemble_input_kernel(const dd_assemble_entry* __restrict__ table, int n_entries, T* __restrict__ dst, int ld, int c_pad,
                                                             int B, int H, int W, int tiles_x, int tiles_y, const int* __restrict__ origins, int frame_pitch) {
  constexpr int TP = 16, HP = TP + 2;
  __shared__ float s_std[3][HP][HP + 1];
  __shared__ float s_var[3][HP][HP + 1];
  extern __shared__ __attribute__((aligned(16))) char s_out_raw[];      // [256 pixels][c_pad elements + 4 bytes] of T
  T* s_out = reinterpret_cast<T*>(s_out_raw);
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int b = bid % B, t = bid / B;
  const int y0 = ty * TP, x0 = tx * TP;
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  const int y = y0 + ly, x = x0 + lx;
  const bool live = y < H && x < W;
  const long pix = ((long)b * H + (live ? y : 0)) * W + (live ? x : 0);
  // Where image b of a render pass starts and how far its rows are apart (pixels): a [B,H,W,cs] batch of tiles, or -- origins != NULL, the
  // inference path -- the H x W window at (origins[2b], origins[2b+1]) of a whole frame whose rows are frame_pitch pixels long (the halo tile
  // of Prediction.py:396-427 read in place: dd_extract_tiles and the tile copies of the passes are gone).  The mirrored 3x3 neighbourhood
  // stays inside the window either way: the reference takes the variance of the TILE.
  const int pitch = origins ? frame_pitch : W;
  const long img_pix = origins ? (long)origins[2 * b] * frame_pitch + origins[2 * b + 1] : (long)b * H * W;
  // a pixel's row is c_pad elements + 4 bytes: with rows of exactly 64 bytes (32 bf16 channels) the 64 lanes of a wave, each storing into its own
  // row, hit 4 bank groups; 68 bytes = 17 banks spreads them over all 64
  const int rstride = c_pad + 4 / (int)sizeof(T);
  T* mine = s_out + threadIdx.x * rstride;
  // The haloed source tile of the NEXT pass is requested (into registers: 2 pixels x 3 channels per thread) while this pass is standardised and
  // its variance taken from LDS: one exposed memory round trip per tile instead of one per pass (8 passes: 249 -> see DESIGN 3.3).
  // The two haloed pixels a thread fetches are the same for every pass: their (mirrored) pixel offsets are computed once.
  int hoff[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int k = threadIdx.x + it * 256;
    const int py = k / HP, px = k - py * HP;
    const int gy = sym_index(min(y0 - 1 + min(py, HP - 1), H), H), gx = sym_index(min(x0 - 1 + px, W), W);
    hoff[it] = gy * pitch + gx;
  }
  float raw[2][3];
  auto is_pass = [&](int e) { return e < n_entries && table[t * n_entries + e].nch > 0 && table[t * n_entries + e].kind == 0; };
  auto request = [&](int e) {
    const dd_assemble_entry en = table[t * n_entries + e];
    const int cs = en.cs;
    const float* img = en.src + img_pix * cs;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const float* sp = img + (long)hoff[it] * cs;
#ifdef AI_EXP_NO_LOAD
      raw[it][0] = raw[it][1] = raw[it][2] = (float)hoff[it]; (void)sp;
#else
      if (cs == 3) { raw[it][0] = sp[0]; raw[it][1] = sp[1]; raw[it][2] = sp[2]; }
      else { raw[it][0] = sp[0]; raw[it][1] = 0.f; raw[it][2] = 0.f; }
#endif
    }
  };
  int e_next = 0, used = 0;
  while (e_next < n_entries && !is_pass(e_next)) ++e_next;
  if (e_next < n_entries) request(e_next);
  for (int e = 0; e < n_entries; ++e) {
    const dd_assemble_entry en = table[t * n_entries + e];       // block-uniform
    if (en.nch <= 0) continue;
    used = max(used, en.dst_ch + en.nch);
    if (en.kind == 1) {                                           // a vector broadcast over the tile (the tuple's embedding row)
      for (int c = 0; c < en.nch; ++c) mine[en.dst_ch + c] = Elem<T>::from_f32(en.src[c]);
      continue;
    }
    if (en.kind == 2) {                                           // a plane copied as it is (one-hot feature flags)
      const float* sp = en.src + pix * en.nch;
      for (int c = 0; c < en.nch; ++c) mine[en.dst_ch + c] = Elem<T>::from_f32(sp[c]);
      continue;
    }
    const dd_feature_params fp = en.fp;
    const int cs = en.cs;
    __syncthreads();                                              // the previous pass's staging tiles are consumed
#pragma unroll
    for (int it = 0; it < 2; ++it) {                              // this pass's values (requested one pass ago) -> LDS, standardised
      const int k = threadIdx.x + it * 256;
      if (k < HP * HP) {
        const int py = k / HP, px = k - py * HP;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (c < cs) {
            const float v = raw[it][c];
#ifdef AI_EXP_NO_LOG
            const float sv = (v - fp.mean) * fp.inv_std;
#else
            const float sv = ((fp.use_log1p ? copysignf(log1pf(fabsf(v)), v) : v) - fp.mean) * fp.inv_std;
#endif
            s_std[c][py][px] = sv;
            s_var[c][py][px] = fp.variance_before ? v : sv;
          }
        }
      }
    }
    e_next = e + 1;
    while (e_next < n_entries && !is_pass(e_next)) ++e_next;
    if (e_next < n_entries) request(e_next);                     // in flight during the variance below
    __syncthreads();
    float rec[6];
    rec[0] = s_std[0][ly + 1][lx + 1];
    rec[1] = cs == 3 ? s_std[1][ly + 1][lx + 1] : rec[0];
    rec[2] = cs == 3 ? s_std[2][ly + 1][lx + 1] : rec[0];
    rec[3] = rec[4] = rec[5] = 0.f;
    int nv = 0;
    if (fp.use_variance) {
      const float inv_cnt = fp.mode_neighbor ? 0.2f : (1.f / 9.f);
      float var_acc = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c < cs) {
          float sum = 0.f, sumsq = 0.f;
#pragma unroll
          for (int a = -1; a <= 1; ++a)
#pragma unroll
            for (int bb = -1; bb <= 1; ++bb) {
              if (a != 0 && bb != 0 && fp.mode_neighbor) continue;
              const float v = s_var[c][ly + 1 + a][lx + 1 + bb];
              sum += v; sumsq += v * v;
            }
          const float mean = sum * inv_cnt, meansq = sumsq * inv_cnt;
          float var = meansq - mean * mean;
          if (fp.relative) var = var * __builtin_amdgcn_rcpf(fmaxf(mean * mean, fp.epsilon));
          var_acc += var;
          rec[3 + c] = var;
        }
      }
      if (fp.compress) { rec[3] = var_acc * (cs == 3 ? (1.f / 3.f) : 1.f); nv = 1; } else nv = cs;
    }
    const int n = min(3 + nv, en.nch);
    {
      T* o = mine + en.dst_ch;
      if (n == 4 && (en.dst_ch & 1) == 0) st4<T>(o, rec[0], rec[1], rec[2], rec[3]);      // (rows start 4-byte aligned)
      else {
#pragma unroll
        for (int c = 0; c < 6; ++c)
          if (c < n) o[c] = Elem<T>::from_f32(rec[c]);
      }
    }
    if (live) {
#ifdef AI_EXP_NO_STDOUT
      if (en.std_out && rec[0] == 123.456f) {
#else
      if (en.std_out) {
#endif
        float* so = en.std_out + pix * en.ld_std;
        if (en.ld_std == 4 && 3 + nv >= 4 && ((uintptr_t)en.std_out & 15) == 0) *reinterpret_cast<float4*>(so) = make_float4(rec[0], rec[1], rec[2], rec[3]);
        else {
#pragma unroll
          for (int c = 0; c < 6; ++c)
            if (c < 3 + nv && c < en.ld_std) so[c] = rec[c];
        }
      }
    }
  }
  for (int c = used; c < c_pad; ++c) mine[c] = Elem<T>::from_f32(0.f);      // padding channels of the network input
  __syncthreads();
  // the tile leaves as 16-byte vectors: pixel row = c_pad elements
  constexpr int N = Elem<T>::PER16;
  const int vpp = c_pad / N;                                      // vectors per pixel
  T* img_out = dst + ((long)(t * B + b) * H) * W * ld;
  for (int v = threadIdx.x; v < 256 * vpp; v += 256) {
    const int p = v / vpp, k = v - p * vpp;
    const int py = p >> 4, px = p & 15;
    if (y0 + py < H && x0 + px < W) {
      const uint32_t* w = reinterpret_cast<const uint32_t*>(s_out + p * rstride + k * N);      // (4-byte aligned rows: four dword reads)
      *reinterpret_cast<uint4*>(img_out + ((long)(y0 + py) * W + x0 + px) * ld + k * N) = uint4{w[0], w[1], w[2], w[3]};
    }
  }
}
static int assemble_common(const dd_assemble_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                           int B, int H, int W, int dtype, const int* origins, int frame_pitch, dd_stream stream) {
  DD_REQUIRE(table && dst && n_tuples > 0 && n_entries > 0 && c_pad > 0 && c_pad <= ld, "dd_assemble_input: bad arguments");
  DD_REQUIRE(dd_dtype_ok(dtype), "dd_assemble_input: bad dtype %d", dtype);
  const int per16 = dtype == DD_F32 ? 4 : 8, esz = dtype == DD_F32 ? 4 : 2;
  DD_REQUIRE(c_pad % per16 == 0 && ld % per16 == 0 && ((uintptr_t)dst % 16) == 0, "dd_assemble_input: c_pad=%d and ld=%d must be multiples of %d", c_pad, ld, per16);
  const size_t lds = (size_t)256 * (c_pad * esz + 4);
  DD_REQUIRE(lds <= 96 * 1024, "dd_assemble_input: %d input channels do not fit the LDS tile", c_pad);
  const int tiles_x = dd_ceil_div(W, 16), tiles_y = dd_ceil_div(H, 16);
  const unsigned grid = (unsigned)((long)n_tuples * B * tiles_x * tiles_y);
  DD_DISPATCH_DTYPE(dtype, T, {
    dd_allow_max_lds(reinterpret_cast<const void*>(assemble_input_kernel<T>), 96 * 1024);
    hipLaunchKernelGGL(assemble_input_kernel<T>, dim3(grid), dim3(256), lds, S(stream), table, n_entries, (T*)dst, ld, c_pad, B, H, W, tiles_x, tiles_y,
                       origins, frame_pitch);
  });
  DD_LAUNCH_CHECK();
  return DD_OK;
}
extern "C" int dd_assemble_input(const dd_assemble_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                                 int B, int H, int W, int dtype, dd_stream stream) {
  return assemble_common(table, n_tuples, n_entries, dst, ld, c_pad, B, H, W, dtype, nullptr, 0, stream);
}
extern "C" int dd_assemble_input_frames(const dd_assemble_entry* table, int n_tuples, int n_entries, void* dst, int ld, int c_pad,
                                        int B, int H, int W, int dtype, const int* origins_yx, int frame_h, int frame_w, dd_stream stream) {
  DD_REQUIRE(origins_yx && H > 0 && W > 0 && H <= frame_h && W <= frame_w, "dd_assemble_input_frames: the %dx%d tile does not fit the %dx%d frame (or no origins)",
             H, W, frame_h, frame_w);
  return assemble_common(table, n_tuples, n_entries, dst, ld, c_pad, B, H, W, dtype, origins_yx, frame_w, stream);
}

// ------------------------------------------------------------------------------------------------ kernel prediction
// The k*k logits of a pixel, either read from the tensor the last 1x1 layer wrote (storage type T), or -- HID -- computed here in fp32 from that
// layer's INPUT: logits[t] = bb[t] + sum_k hid[k] * wb[k * ldw + t]  (AdjustNumberOfChannels' second conv, Architecture.py:237-244, with the fp32
// master weights).  A logit of magnitude 50 stored in bf16 is off by up to 0.125, its softmax weight by 13 %: the layer-wise head of the
// half-precision programs (Tiramisu outputs, COMBINED tuples: what the fused head of csrc/dd_head.hip does not take) keeps them in registers.
// Every thread reads the same weights: scalar loads through the constant cache.
template <typename T, int K2, int K2P, bool VEC, bool HID>
__device__ __forceinline__ void kpcn_logits(const T* __restrict__ row, int kh, const float* __restrict__ wb, int ldw, const float* __restrict__ bb, float (&w)[K2P]) {
  constexpr int N = Elem<T>::PER16;
  if (HID) {
#pragma unroll
    for (int t = 0; t < K2; ++t) w[t] = bb[t];
    for (int k0 = 0; k0 < kh; k0 += N) {      // (the row is padded to whole 16-byte vectors: ld % N == 0, host-checked)
      float h8[N];
      vload<T>(row + k0, h8);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        if (k0 + e < kh) {
          const float* wr = wb + (long)(k0 + e) * ldw;
#pragma unroll
          for (int t = 0; t < K2; ++t) w[t] = fmaf(h8[e], wr[t], w[t]);
        }
      }
    }
  } else if (VEC) {
#pragma unroll
    for (int v = 0; v < K2P / N; ++v) {
      float t8[N];
      vload<T>(row + v * N, t8);
#pragma unroll
      for (int e = 0; e < N; ++e) w[v * N + e] = t8[e];
    }
  } else {     // COMBINED tuples: member j's logits start at channel j * K2, not 16-byte aligned
#pragma unroll
    for (int t = 0; t < K2; ++t) w[t] = ld1<T>(row + t);
  }
}

template <typename T, int KS, bool VEC, bool HID>
__global__ __launch_bounds__(128) void kpcn_fwd_kernel(const float* __restrict__ src, int ldsrc, const T* __restrict__ logits, int ldl,
                                float* __restrict__ out, int ldo, int B, int H, int W,
                                int kh, const float* __restrict__ wb, int ldw, const float* __restrict__ bb) {
  constexpr int K2 = KS * KS, P = (KS - 1) / 2;
  const long total = (long)B * H * W;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long r = i / W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  constexpr int N = Elem<T>::PER16, K2P = (K2 + N - 1) / N * N;      // logits are read as 16-byte vectors (ldl >= K2P, checked by the host)
  float w[K2P];
  float mx = -INFINITY;
  kpcn_logits<T, K2, K2P, VEC, HID>(logits + i * ldl, kh, wb, ldw, bb, w);
#pragma unroll
  for (int t = 0; t < K2; ++t) mx = fmaxf(mx, w[t]);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < K2; ++t) { w[t] = __expf(w[t] - mx); sum += w[t]; }
  const float inv = 1.f / sum;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  const float* img = src + (long)b * H * W * ldsrc;
#pragma unroll
  for (int a = 0; a < KS; ++a) {
    const int sy = sym_index(y + a - P, H);
#pragma unroll
    for (int bb2 = 0; bb2 < KS; ++bb2) {
      const int sx = sym_index(x + bb2 - P, W);
      const float* s = img + ((long)sy * W + sx) * ldsrc;
      const float wt = w[a * KS + bb2] * inv;
      o0 += wt * s[0]; o1 += wt * s[1]; o2 += wt * s[2];
    }
  }
  float* o = out + i * ldo;
  o[0] = o0; o[1] = o1; o[2] = o2;
}

template <typename T, int KS, bool VEC, bool HID>
__global__ __launch_bounds__(128) void kpcn_bwd_kernel(const float* __restrict__ src, int ldsrc, const T* __restrict__ logits, int ldl,
                                const float* __restrict__ dout, int lddo, T* __restrict__ dlogits, int lddl, int dl_pad,
                                int B, int H, int W, int kh, const float* __restrict__ wb, int ldw, const float* __restrict__ bb) {
  constexpr int K2 = KS * KS, P = (KS - 1) / 2;
  const long total = (long)B * H * W;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long r = i / W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  constexpr int N = Elem<T>::PER16, K2P = (K2 + N - 1) / N * N;
  float w[K2P], dw[K2];
  float mx = -INFINITY;
  kpcn_logits<T, K2, K2P, VEC, HID>(logits + i * ldl, kh, wb, ldw, bb, w);
#pragma unroll
  for (int t = 0; t < K2; ++t) mx = fmaxf(mx, w[t]);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < K2; ++t) { w[t] = __expf(w[t] - mx); sum += w[t]; }
  const float inv = 1.f / sum;
  const float g0 = dout[i * lddo], g1 = dout[i * lddo + 1], g2 = dout[i * lddo + 2];
  const float* img = src + (long)b * H * W * ldsrc;
  float dot = 0.f;
#pragma unroll
  for (int a = 0; a < KS; ++a) {
    const int sy = sym_index(y + a - P, H);
#pragma unroll
    for (int bb2 = 0; bb2 < KS; ++bb2) {
      const int sx = sym_index(x + bb2 - P, W);
      const float* s = img + ((long)sy * W + sx) * ldsrc;
      const int t = a * KS + bb2;
      w[t] *= inv;
      dw[t] = g0 * s[0] + g1 * s[1] + g2 * s[2];
      dot += w[t] * dw[t];
    }
  }
  T* o = dlogits + i * lddl;
  if (!VEC) {
#pragma unroll
    for (int t = 0; t < K2; ++t) o[t] = Elem<T>::from_f32(w[t] * (dw[t] - dot));
    for (int t = K2; t < dl_pad; ++t) o[t] = Elem<T>::from_f32(0.f);
    return;
  }
#pragma unroll
  for (int v = 0; v < K2P / N; ++v) {            // 16-byte stores; channels K2..dl_pad-1 are zero (dl_pad is a multiple of N, host-checked)
    float t8[N];
#pragma unroll
    for (int e = 0; e < N; ++e) t8[e] = v * N + e < K2 ? w[v * N + e] * (dw[v * N + e < K2 ? v * N + e : 0] - dot) : 0.f;
    if (v * N < dl_pad) vstore<T>(o + v * N, t8);
  }
  for (int t = K2P; t < dl_pad; t += N) {
    float z[N];
#pragma unroll
    for (int e = 0; e < N; ++e) z[e] = 0.f;
    vstore<T>(o + t, z);
  }
}

// hid != NULL: the logits are computed in the kernel from `hid` (kh channels, rows of ldl elements) and (wb, ldw, bb); `logits` is not read
template <typename T>
static int kpcn_dispatch(bool fwd, const float* src, int ldsrc, const void* logits, int ldl, const float* dout, int lddo,
                         void* out, int ldo, int pad, int B, int H, int W, int ks, hipStream_t s,
                         const void* hid = nullptr, int kh = 0, const float* wb = nullptr, int ldw = 0, const float* bb = nullptr) {
  const long total = (long)B * H * W;
  const dim3 g(grid_for(total, 128)), blk(128);
  // 16-byte vector path: the logits (and, backward, the logit gradients) of this call start 16-byte aligned and are padded to whole vectors
  constexpr int n = Elem<T>::PER16;
  const int k2p = (ks * ks + n - 1) / n * n;
  bool vec = hid ? true : (((uintptr_t)logits % 16) == 0 && ldl % n == 0 && ldl >= k2p);      // (the logit READS; hid rows are always vector-aligned)
  if (!fwd) vec = ((uintptr_t)out % 16) == 0 && ldo % n == 0 && pad % n == 0 && pad >= k2p && vec;
  const T* in = (const T*)(hid ? hid : logits);
#define KP_LAUNCH(K, V, HD)                                                                                                    \
    if (fwd) hipLaunchKernelGGL((kpcn_fwd_kernel<T, K, V, HD>), g, blk, 0, s, src, ldsrc, in, ldl, (float*)out, ldo, B, H, W, kh, wb, ldw, bb); \
    else hipLaunchKernelGGL((kpcn_bwd_kernel<T, K, V, HD>), g, blk, 0, s, src, ldsrc, in, ldl, dout, lddo, (T*)out, ldo, pad, B, H, W, kh, wb, ldw, bb);
#define KP_CASE(K)                                                                                                             \
  case K:                                                                                                                      \
    if (hid) { if (vec) { KP_LAUNCH(K, true, true) } else { KP_LAUNCH(K, false, true) } }                                      \
    else if (vec) { KP_LAUNCH(K, true, false) } else { KP_LAUNCH(K, false, false) }                                            \
    break;
  switch (ks) {
    KP_CASE(3) KP_CASE(5) KP_CASE(7)
    default: dd_set_error("kernel prediction: kernel_size %d unsupported (3, 5, 7)", ks); return DD_ERR_INVALID;
  }
#undef KP_CASE
#undef KP_LAUNCH
  DD_LAUNCH_CHECK();
  return DD_OK;
}
extern "C" int dd_kpcn_fwd(const float* src, int ldsrc, const void* logits, int ldl, float* out, int ldo,
                           int B, int H, int W, int ksize, int dtype, dd_stream stream) {
  DD_REQUIRE(src && logits && out && ldl >= ksize * ksize, "dd_kpcn_fwd: bad arguments");
  DD_DISPATCH_DTYPE(dtype, T, return kpcn_dispatch<T>(true, src, ldsrc, logits, ldl, nullptr, 0, out, ldo, 0, B, H, W, ksize, S(stream)));
  return DD_ERR_INVALID;
}
extern "C" int dd_kpcn_bwd(const float* src, int ldsrc, const void* logits, int ldl, const float* dout, int lddo,
                           void* dlogits, int lddl, int dl_pad, int B, int H, int W, int ksize, int dtype, dd_stream stream) {
  DD_REQUIRE(src && logits && dout && dlogits && ldl >= ksize * ksize && dl_pad <= lddl, "dd_kpcn_bwd: bad arguments");
  DD_DISPATCH_DTYPE(dtype, T, return kpcn_dispatch<T>(false, src, ldsrc, logits, ldl, dout, lddo, dlogits, lddl, dl_pad, B, H, W, ksize, S(stream)));
  return DD_ERR_INVALID;
}
static bool kpcn_hidden_ok(const void* hid, int ldh, int kh, const float* wb, int ldw, const float* bb, int ksize, int dtype) {
  const int n = dtype == DD_F32 ? 4 : 8;
  return hid && wb && bb && kh > 0 && ldh % n == 0 && ldh >= (kh + n - 1) / n * n && ((uintptr_t)hid % 16) == 0 && ldw >= ksize * ksize;
}
extern "C" int dd_kpcn_hidden_fwd(const float* src, int ldsrc, const void* hid, int ldh, int kh, const float* wb, int ldw, const float* bb,
                                  float* out, int ldo, int B, int H, int W, int ksize, int dtype, dd_stream stream) {
  DD_REQUIRE(src && out && kpcn_hidden_ok(hid, ldh, kh, wb, ldw, bb, ksize, dtype), "dd_kpcn_hidden_fwd: bad arguments (hid rows must be 16-byte vectors, ldw >= k*k)");
  DD_DISPATCH_DTYPE(dtype, T, return kpcn_dispatch<T>(true, src, ldsrc, nullptr, ldh, nullptr, 0, out, ldo, 0, B, H, W, ksize, S(stream), hid, kh, wb, ldw, bb));
  return DD_ERR_INVALID;
}
extern "C" int dd_kpcn_hidden_bwd(const float* src, int ldsrc, const void* hid, int ldh, int kh, const float* wb, int ldw, const float* bb,
                                  const float* dout, int lddo, void* dlogits, int lddl, int dl_pad, int B, int H, int W, int ksize, int dtype, dd_stream stream) {
  DD_REQUIRE(src && dout && dlogits && dl_pad <= lddl && kpcn_hidden_ok(hid, ldh, kh, wb, ldw, bb, ksize, dtype), "dd_kpcn_hidden_bwd: bad arguments");
  DD_DISPATCH_DTYPE(dtype, T, return kpcn_dispatch<T>(false, src, ldsrc, nullptr, ldh, dout, lddo, dlogits, lddl, dl_pad, B, H, W, ksize, S(stream), hid, kh, wb, ldw, bb));
  return DD_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------ data augmentation
// One thread = one OUTPUT pixel: the source pixel follows from inverting rot90(flip(.)), the channel transform depends on the pass kind.
__global__ void augment_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int B, int H, int W,
                               const dd_augment_draw* __restrict__ draws, int kind, int use_flip, int use_rotate, int use_permute,
                               int use_normal_rotation) {
  const long total = (long)B * H * W;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long r = i / W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  const dd_augment_draw d = draws[b];
  const int k = use_rotate ? (d.rotate & 3) : 0;
  const bool flip = use_flip && d.flip > 0;
  // R = rot90(F, k) counter-clockwise (tf.image.rot90 / np.rot90): R[i][j] = F[j][N-1-i], F[N-1-i][N-1-j], F[N-1-j][i] for k = 1, 2, 3
  int fy = y, fx = x;
  if (k == 1) { fy = x; fx = W - 1 - y; }
  else if (k == 2) { fy = H - 1 - y; fx = W - 1 - x; }
  else if (k == 3) { fy = H - 1 - x; fx = y; }
  if (flip) fx = W - 1 - fx;                      // F[y][x] = in[y][W-1-x]
  const float* s = src + (((long)b * H + fy) * W + fx) * C;
  float* o = dst + i * C;
  if (C != 3) {
    for (int c = 0; c < C; ++c) o[c] = s[c];
    return;
  }
  float v0 = s[0], v1 = s[1], v2 = s[2];
  if (kind == DD_AUG_SCREEN_NORMAL) {
    if (flip) v0 = -v0;                             // DataAugmentation._flip_screen_space_normals
    if (k == 1) { const float t = v0; v0 = -v1; v1 = t; }          // _rotate_90:  x -> -y, y -> x
    else if (k == 2) { v0 = -v0; v1 = -v1; }                       // _rotate_180
    else if (k == 3) { const float t = v1; v1 = -v0; v0 = t; }     // _rotate_270: x -> y, y -> -x
  } else if (kind == DD_AUG_RGB && use_permute) {
    const float a0 = v0, a1 = v1, a2 = v2;
    switch (d.permute) {                            // DataAugmentation.permute_rgb: result[c] = input[permutation[c]]
      case 1: v0 = a0; v1 = a2; v2 = a1; break;
      case 2: v0 = a1; v1 = a0; v2 = a2; break;
      case 3: v0 = a1; v1 = a2; v2 = a0; break;
      case 4: v0 = a2; v1 = a0; v2 = a1; break;
      case 5: v0 = a2; v1 = a1; v2 = a0; break;
      default: break;
    }
  } else if (kind == DD_AUG_NORMAL && use_normal_rotation) {
    const float* m = d.normal_rotation;             // tf.matmul(inputs [HW,3], rotation_matrix [3,3])
    const float a0 = v0, a1 = v1, a2 = v2;
    v0 = a0 * m[0] + a1 * m[3] + a2 * m[6];
    v1 = a0 * m[1] + a1 * m[4] + a2 * m[7];
    v2 = a0 * m[2] + a1 * m[5] + a2 * m[8];
  }
  o[0] = v0; o[1] = v1; o[2] = v2;
}
extern "C" int dd_augment(const float* src, float* dst, int C, int B, int H, int W, const dd_augment_draw* draws, int kind,
                          int use_flip, int use_rotate, int use_permute, int use_normal_rotation, dd_stream stream) {
  DD_REQUIRE(src && dst && draws && src != dst, "dd_augment: null or aliased pointers");
  DD_REQUIRE(C == 1 || C == 3, "dd_augment: C=%d (1 or 3)", C);
  DD_REQUIRE(kind >= DD_AUG_PLAIN && kind <= DD_AUG_SCREEN_NORMAL && (kind == DD_AUG_PLAIN || C == 3), "dd_augment: kind=%d needs 3 channels", kind);
  DD_REQUIRE(!use_rotate || H == W, "dd_augment: rotate_90 needs square tiles (H=%d W=%d)", H, W);
  DD_REQUIRE(!(kind == DD_AUG_NORMAL && use_flip), "dd_augment: flipping normals is not supported (DataAugmentation.py:22-23)");
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(augment_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), src, dst, C, B, H, W, draws, kind, use_flip, use_rotate,
                     use_permute, use_normal_rotation);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ multiscale compose
template <typename T>
__global__ void compose_pack_kernel(const float* __restrict__ small, int lds, const float* __restrict__ fine, int ldf,
                                    T* __restrict__ dst, int ld, int c_pad, int B, int H, int W) {
  const long total = (long)B * H * W;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long r = i / W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  const float* s = small + (((long)b * (H / 2) + y / 2) * (W / 2) + x / 2) * lds;
  const float* f = fine + i * ldf;
  T* o = dst + i * ld;
  o[0] = Elem<T>::from_f32(s[0]); o[1] = Elem<T>::from_f32(s[1]); o[2] = Elem<T>::from_f32(s[2]);
  o[3] = Elem<T>::from_f32(f[0]); o[4] = Elem<T>::from_f32(f[1]); o[5] = Elem<T>::from_f32(f[2]);
  for (int c = 6; c < c_pad; ++c) o[c] = Elem<T>::from_f32(0.f);
}
extern "C" int dd_compose_pack(const float* small, int lds, const float* fine, int ldf, void* dst, int ld, int c_pad,
                               int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(small && fine && dst && H % 2 == 0 && W % 2 == 0 && c_pad >= 6 && c_pad <= ld, "dd_compose_pack: bad arguments");
  const long total = (long)B * H * W;
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(compose_pack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), small, lds, fine, ldf, (T*)dst, ld, c_pad, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// one thread per coarse pixel (2x2 block of fine pixels)
template <typename T>
__global__ void compose_blend_fwd_kernel(const float* __restrict__ small, int lds, const float* __restrict__ fine, int ldf,
                                         const T* __restrict__ wl, int ldw, float* __restrict__ out, int ldo, int B, int H, int W) {
  const int h2 = H / 2, w2 = W / 2;
  const long total = (long)B * h2 * w2;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int qx = (int)(i % w2);
  const long r = i / w2;
  const int qy = (int)(r % h2);
  const int b = (int)(r / h2);
  const float* s = small + i * lds;
  long pix[4];
  float f[4][3], low[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pix[k] = ((long)b * H + 2 * qy + (k >> 1)) * W + 2 * qx + (k & 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) { f[k][c] = fine[pix[k] * ldf + c]; low[c] += f[k][c]; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) low[c] *= 0.25f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float w = sigmoidf(ld1<T>(wl + pix[k] * ldw));
#pragma unroll
    for (int c = 0; c < 3; ++c) out[pix[k] * ldo + c] = f[k][c] - w * low[c] + w * s[c];
  }
}
extern "C" int dd_compose_blend_fwd(const float* small, int lds, const float* fine, int ldf, const void* wl, int ldw,
                                    float* out, int ldo, int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(small && fine && wl && out && H % 2 == 0 && W % 2 == 0, "dd_compose_blend_fwd: bad arguments");
  const long total = (long)B * (H / 2) * (W / 2);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(compose_blend_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), small, lds, fine, ldf, (const T*)wl, ldw, out, ldo, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void compose_blend_bwd_kernel(const float* __restrict__ dout, int lddo, const float* __restrict__ small, int lds,
                                         const float* __restrict__ fine, int ldf, const T* __restrict__ wl, int ldw,
                                         float* __restrict__ dsmall, int ldds, int acc_small, float* __restrict__ dfine, int lddf,
                                         T* __restrict__ dwl, int lddw, int dw_pad, int B, int H, int W) {
  const int h2 = H / 2, w2 = W / 2;
  const long total = (long)B * h2 * w2;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int qx = (int)(i % w2);
  const long r = i / w2;
  const int qy = (int)(r % h2);
  const int b = (int)(r / h2);
  const float* s = small + i * lds;
  long pix[4];
  float f[4][3], g[4][3], w[4], wlv[4], low[3] = {0.f, 0.f, 0.f}, tsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pix[k] = ((long)b * H + 2 * qy + (k >> 1)) * W + 2 * qx + (k & 1);
    wlv[k] = ld1<T>(wl + pix[k] * ldw);
    w[k] = sigmoidf(wlv[k]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f[k][c] = fine[pix[k] * ldf + c];
      g[k][c] = dout[pix[k] * lddo + c];
      low[c] += f[k][c];
      tsum[c] += w[k] * g[k][c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { low[c] *= 0.25f; dsmall[i * ldds + c] = (acc_small ? dsmall[i * ldds + c] : 0.f) + tsum[c]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float dw = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dw += g[k][c] * (s[c] - low[c]);
      dfine[pix[k] * lddf + c] = g[k][c] - 0.25f * tsum[c];
    }
    const float d = wlv[k] > 0.f ? dw * w[k] * (1.f - w[k]) : 0.f;
    T* o = dwl + pix[k] * lddw;
    o[0] = Elem<T>::from_f32(d);
    for (int c = 1; c < dw_pad; ++c) o[c] = Elem<T>::from_f32(0.f);
  }
}
extern "C" int dd_compose_blend_bwd(const float* dout, int lddo, const float* small, int lds, const float* fine, int ldf,
                                    const void* wl, int ldw, float* dsmall, int ldds, int accumulate_small, float* dfine, int lddf,
                                    void* dwl, int lddw, int dw_pad, int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(dout && small && fine && wl && dsmall && dfine && dwl && H % 2 == 0 && W % 2 == 0 && dw_pad <= lddw, "dd_compose_blend_bwd: bad arguments");
  const long total = (long)B * (H / 2) * (W / 2);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(compose_blend_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), dout, lddo, small, lds, fine, ldf, (const T*)wl, ldw, dsmall, ldds, accumulate_small, dfine, lddf, (T*)dwl, lddw, dw_pad, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void compose_unpack_bwd_kernel(const T* __restrict__ dnet, int ld, float* __restrict__ dsmall, int ldds,
                                          float* __restrict__ dfine, int lddf, int B, int H, int W) {
  const int h2 = H / 2, w2 = W / 2;
  const long total = (long)B * h2 * w2;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int qx = (int)(i % w2);
  const long r = i / w2;
  const int qy = (int)(r % h2);
  const int b = (int)(r / h2);
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long pix = ((long)b * H + 2 * qy + (k >> 1)) * W + 2 * qx + (k & 1);
    const T* d = dnet + pix * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] += ld1<T>(d + c);
      dfine[pix * lddf + c] += ld1<T>(d + 3 + c);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) dsmall[i * ldds + c] += acc[c];
}
extern "C" int dd_compose_unpack_bwd(const void* dnet, int ld, float* dsmall, int ldds, float* dfine, int lddf,
                                     int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(dnet && dsmall && dfine && H % 2 == 0 && W % 2 == 0 && ld >= 6, "dd_compose_unpack_bwd: bad arguments");
  const long total = (long)B * (H / 2) * (W / 2);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(compose_unpack_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)dnet, ld, dsmall, ldds, dfine, lddf, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ inverse standardization
__global__ void invert_std_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int use_log1p, float mean, float std) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float z = x[i] * std + mean;
  if (use_log1p) z = z > 0.f ? expm1f(z) : (z < 0.f ? -expm1f(-z) : 0.f);
  y[i] = z;
}
__global__ void invert_std_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n,
                                      int use_log1p, float mean, float std) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float z = x[i] * std + mean;
  float d = std;
  if (use_log1p) d *= (z != 0.f) ? expf(fabsf(z)) : 0.f;  // d/dz sign(z)*expm1(|z|) = sign(z)^2 * exp(|z|)  (tf.sign(0) = 0)
  dx[i] = dy[i] * d;
}
extern "C" int dd_invert_std_fwd(const float* x, float* y, long n, int use_log1p, float mean, float std, dd_stream stream) {
  DD_REQUIRE(x && y && n > 0, "dd_invert_std_fwd: bad arguments");
  hipLaunchKernelGGL(invert_std_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), x, y, n, use_log1p, mean, std);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
extern "C" int dd_invert_std_bwd(const float* x, const float* dy, float* dx, long n, int use_log1p, float mean, float std, dd_stream stream) {
  DD_REQUIRE(x && dy && dx && n > 0, "dd_invert_std_bwd: bad arguments");
  hipLaunchKernelGGL(invert_std_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), x, dy, dx, n, use_log1p, mean, std);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ loss head
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
// loss term and its derivative with respect to the prediction
__device__ __forceinline__ void loss_term(int kind, float eps, float p, float t, float* l, float* dl) {
  const float d = p - t;
  switch (kind) {
    case 1: *l = d; *dl = 1.f; break;
    case 2: *l = fabsf(d); *dl = sgn(d); break;
    case 3: { const float a = fabsf(d); if (a < 1.f) { *l = 0.5f * a * a; *dl = d; } else { *l = a - 0.5f; *dl = sgn(d); } break; }
    case 4: *l = d * d; *dl = 2.f * d; break;
    default: {
      const float a = fabsf(d), den = fabsf(p) + fabsf(t) + eps;
      *l = a / den;
      *dl = sgn(d) / den - a * sgn(p) / (den * den);
    }
  }
}

// Value of one loss "source" at a pixel: prediction and target, 3 channels (1-channel passes broadcast, tf.multiply broadcasting,
// Training.py:422-426).
struct Val3 { float p[3], t[3]; };
__device__ __forceinline__ Val3 feature_value(const dd_loss_desc& d, int f, long i) {
  Val3 v;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int cf = d.nch[f] == 1 ? 0 : c;
    v.p[c] = d.pred[f][i * d.pred_ld[f] + cf];
    v.t[c] = d.target[f][i * d.target_ld[f] + cf];
  }
  return v;
}
__device__ __forceinline__ Val3 combined_value(const dd_loss_desc& d, int k, long i) {      // color * (direct + indirect)
  const Val3 c = feature_value(d, d.comb[k][0], i), dr = feature_value(d, d.comb[k][1], i), in = feature_value(d, d.comb[k][2], i);
  Val3 v;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) { v.p[ch] = c.p[ch] * (dr.p[ch] + in.p[ch]); v.t[ch] = c.t[ch] * (dr.t[ch] + in.t[ch]); }
  return v;
}
__device__ __forceinline__ Val3 image_value(const dd_loss_desc& d, long i) {                 // sum of the combined features and single passes
  Val3 v = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int j = 0; j < d.n_image_combined; ++j) {
    const Val3 a = combined_value(d, d.image_combined[j], i);
#pragma unroll
    for (int c = 0; c < 3; ++c) { v.p[c] += a.p[c]; v.t[c] += a.t[c]; }
  }
  for (int j = 0; j < d.n_image_features; ++j) {
    const Val3 a = feature_value(d, d.image_features[j], i);
#pragma unroll
    for (int c = 0; c < 3; ++c) { v.p[c] += a.p[c]; v.t[c] += a.t[c]; }
  }
  return v;
}

// Mean + variation terms of one source at pixel i: adds this pixel's share to `loss` and returns dLoss/dvalue(i) in g[0..nch).
// A pair (i, neighbour) belongs to its left / upper pixel for the loss SUM; the gradient of every pair reaches both of its pixels.
template <typename F>
__device__ __forceinline__ void source_terms(F value, long i, int x, int y, int H, int W, int nch, float w_mean, float w_var, int kind,
                                             float eps, float grad_scale, float& loss, float (&g)[3]) {
  g[0] = g[1] = g[2] = 0.f;
  if (w_mean == 0.f && w_var == 0.f) return;
  const Val3 c = value(i);
  if (w_mean != 0.f)
    for (int ch = 0; ch < nch; ++ch) {
      float l, dl;
      loss_term(kind, eps, c.p[ch], c.t[ch], &l, &dl);
      loss += w_mean * l;
      g[ch] += w_mean * dl * grad_scale;
    }
  if (w_var == 0.f) return;
  // (neighbour offset, exists, this pixel is the FIRST element of the pair)
  const long off[4] = {1, -1, (long)W, -(long)W};
  const bool ok[4] = {x + 1 < W, x > 0, y + 1 < H, y > 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!ok[k]) continue;
    const Val3 n = value(i + off[k]);
    const bool first = (k & 1) == 0;
    for (int ch = 0; ch < nch; ++ch) {
      // variation = second - first (Training.py:305-316: shift_left - shift_right = x[j+1] - x[j])
      const float vp = first ? n.p[ch] - c.p[ch] : c.p[ch] - n.p[ch];
      const float vt = first ? n.t[ch] - c.t[ch] : c.t[ch] - n.t[ch];
      float l, dl;
      loss_term(kind, eps, vp, vt, &l, &dl);
      if (first) { loss += w_var * l; g[ch] -= w_var * dl * grad_scale; }
      else g[ch] += w_var * dl * grad_scale;
    }
  }
}

// Conv2dUtilities.non_zero_mask (Conv2dUtilities.py:69-74) of feature f's target at pixel i: sign(sum_c |t_c|)
__device__ __forceinline__ float target_mask(const dd_loss_desc& d, int f, long i) {
  float s = 0.f;
  for (int c = 0; c < d.nch[f]; ++c) s += fabsf(d.target[f][i * d.target_ld[f] + c]);
  return s > 0.f ? 1.f : 0.f;
}
// per-pixel weight of the masked mean: mask / mask_sum (0 when the batch has no masked pixel, Training.py:133-137)
__device__ __forceinline__ float masked_pixel_weight(const dd_loss_desc& d, float w, int mask_f, int src, long i) {
  if (w == 0.f || mask_f < 0 || d.mask_sums == nullptr) return 0.f;
  const float msum = d.mask_sums[src];
  return msum > 0.f ? w * target_mask(d, mask_f, i) / msum : 0.f;
}

__global__ void loss_mask_sums_kernel(const dd_loss_desc d, long npix, float* __restrict__ sums) {
  __shared__ float red[256];
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (int src = 0; src < DD_MAX_FEATURES + DD_MAX_COMBINED; ++src) {       // block-uniform loop
    const bool is_f = src < DD_MAX_FEATURES;
    const int k = is_f ? src : src - DD_MAX_FEATURES;
    if (is_f ? k >= d.n_features : k >= d.n_combined) continue;
    const float w = is_f ? d.masked_weight[k] : d.comb_masked_weight[k];
    const int mf = is_f ? d.mask_feature[k] : d.comb_mask_feature[k];
    if (w == 0.f || mf < 0) continue;
    red[threadIdx.x] = i < npix ? target_mask(d, mf, i) : 0.f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0.f) { dd_det_wait(); atomicAdd(sums + src, red[0]); }
    __syncthreads();
  }
  dd_det_end();
}
extern "C" int dd_loss_mask_sums(const dd_loss_desc* desc, int B, int H, int W, float* mask_sums, dd_stream stream) {
  DD_REQUIRE(desc && mask_sums && desc->n_features > 0 && desc->n_features <= DD_MAX_FEATURES && desc->n_combined <= DD_MAX_COMBINED,
             "dd_loss_mask_sums: bad descriptor");
  const long npix = (long)B * H * W;
  if (hipMemsetAsync(mask_sums, 0, sizeof(float) * (DD_MAX_FEATURES + DD_MAX_COMBINED), S(stream)) != hipSuccess) {
    dd_set_error("dd_loss_mask_sums: hipMemsetAsync failed");
    return DD_ERR_LAUNCH;
  }
  dd_det_sync();
  hipLaunchKernelGGL(loss_mask_sums_kernel, dim3(grid_for(npix)), dim3(256), 0, S(stream), *desc, npix, mask_sums);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

__global__ void loss_head_kernel(const dd_loss_desc d, long npix, int H, int W, float inv_count, float inv_count_var, float grad_scale,
                                 float* __restrict__ loss_out) {
  __shared__ float red[256];
  float loss = 0.f;
  // grid-stride: a few thousand workgroups walk all pixels, so the descriptor (kernel argument, ~1 KB of scalar loads per wave) and the
  // block reduction are paid once per ~8 pixels of a thread instead of once per pixel
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    for (int f = 0; f < d.n_features; ++f) {
      float g[3];
      source_terms([&](long j) { return feature_value(d, f, j); }, i, x, y, H, W, d.nch[f],
                   d.weight[f] * inv_count + masked_pixel_weight(d, d.masked_weight[f], d.mask_feature[f], f, i),
                   d.var_weight[f] * inv_count_var, d.kind, d.epsilon, grad_scale, loss, g);
      float* dp = d.dpred[f] + i * 3;
      dp[0] = g[0]; dp[1] = g[1]; dp[2] = g[2];
    }
    if (d.n_combined > 0 || d.n_image_features > 0) {
      float dimg[3] = {0.f, 0.f, 0.f};
      const bool use_image = (d.image_weight != 0.f || d.image_var_weight != 0.f) && (d.n_image_combined > 0 || d.n_image_features > 0);
      if (use_image) {
        source_terms([&](long j) { return image_value(d, j); }, i, x, y, H, W, 3, d.image_weight * inv_count, d.image_var_weight * inv_count_var,
                     d.kind, d.epsilon, grad_scale, loss, dimg);
        for (int j = 0; j < d.n_image_features; ++j) {
          float* dp = d.dpred[d.image_features[j]] + i * 3;
          for (int c = 0; c < 3; ++c) dp[d.nch[d.image_features[j]] == 1 ? 0 : c] += dimg[c];
        }
      }
      for (int k = 0; k < d.n_combined; ++k) {
        const int fc = d.comb[k][0], fd = d.comb[k][1], fi = d.comb[k][2];
        bool in_image = false;
        for (int j = 0; j < d.n_image_combined; ++j) in_image |= (d.image_combined[j] == k);
        float g[3];
        source_terms([&](long j) { return combined_value(d, k, j); }, i, x, y, H, W, 3,
                     d.comb_weight[k] * inv_count + masked_pixel_weight(d, d.comb_masked_weight[k], d.comb_mask_feature[k], DD_MAX_FEATURES + k, i),
                     d.comb_var_weight[k] * inv_count_var, d.kind, d.epsilon, grad_scale, loss, g);
        const Val3 c = feature_value(d, fc, i), dr = feature_value(d, fd, i), in = feature_value(d, fi, i);
        for (int ch = 0; ch < 3; ++ch) {
          const float gt = g[ch] + ((use_image && in_image) ? dimg[ch] : 0.f);
          const int cc = d.nch[fc] == 1 ? 0 : ch, cd = d.nch[fd] == 1 ? 0 : ch, ci = d.nch[fi] == 1 ? 0 : ch;
          d.dpred[fc][i * 3 + cc] += gt * (dr.p[ch] + in.p[ch]);
          d.dpred[fd][i * 3 + cd] += gt * c.p[ch];
          d.dpred[fi][i * 3 + ci] += gt * c.p[ch];
        }
      }
    }
  }
  red[threadIdx.x] = loss;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) { dd_det_wait(); atomicAdd(loss_out, red[0]); }
  dd_det_end();
}
// Features-only fast path (no combined / image / variation / masked terms: every term is a function of one element of one feature): the
// [B,H,W,3] blocks are walked as flat float4 streams, 2 vectors in flight per thread.  Optionally fused with the inverse standardization on both
// sides of the loss (FeaturePrediction.prediction_invert_standardization, Architecture.py:134-138 -> Architecture.py:47-55, Utilities.py:3-7):
// with pred_std[f] set the kernel reads the standardized prediction x, stores p = invert(x) to pred_inv[f], and writes dL/dx instead of
// dL/dp -- the expressions of invert_std_fwd_kernel / invert_std_bwd_kernel, element for element (three launches per scale become one:
// 36 -> 16 bytes per element of HBM traffic).
struct LossSimpleF {
  const float* x; const float* t; float* p; float* d;
  float w; float mean, std; int log1p, one_channel, fused;
};
struct LossSimpleP { LossSimpleF f[DD_MAX_FEATURES]; int n; int kind; float eps, grad_scale; long nvec; float* loss_out; };

__device__ __forceinline__ void loss_simple_elem(const LossSimpleF& f, int kind, float eps, float gs, bool live, float x, float t, float& p, float& d, float& loss) {
  float dinv = 1.f;
  p = x;
  if (f.fused) {
    float z = x * f.std + f.mean;
    dinv = f.std;
    if (f.log1p) {
      dinv *= (z != 0.f) ? expf(fabsf(z)) : 0.f;
      z = z > 0.f ? expm1f(z) : (z < 0.f ? -expm1f(-z) : 0.f);
    }
    p = z;
  }
  float l, dl;
  loss_term(kind, eps, p, t, &l, &dl);
  loss += live ? f.w * l : 0.f;
  const float g = live ? f.w * dl * gs : 0.f;
  d = f.fused ? g * dinv : g;
}

__global__ __launch_bounds__(256) void loss_simple_kernel(const LossSimpleP P) {
  __shared__ float red[4];
  float loss = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (int fi = 0; fi < P.n; ++fi) {                      // block-uniform
    const LossSimpleF f = P.f[fi];
    const float4* xv = reinterpret_cast<const float4*>(f.x);
    const float4* tv = reinterpret_cast<const float4*>(f.t);
    float4* pv = reinterpret_cast<float4*>(f.p);
    float4* dv = reinterpret_cast<float4*>(f.d);
    for (long v0 = blockIdx.x * 256L + threadIdx.x; v0 < P.nvec; v0 += 2 * stride) {
      const long v1 = v0 + stride;
      const bool has1 = v1 < P.nvec;
      const float4 xa = xv[v0], ta = tv[v0];
      const float4 xb = has1 ? xv[v1] : xa, tb = has1 ? tv[v1] : ta;
      auto one = [&](long v, const float4& x4, const float4& t4) {
        // element e = 4 v + k of the block is channel e % 3 of its pixel: a 1-channel pass only counts channel 0 (Training.py:116-129)
        const int c0 = (int)((4 * v) % 3);
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ts[4] = {t4.x, t4.y, t4.z, t4.w};
        float ps[4], ds[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = c0 + k >= 3 ? (c0 + k >= 6 ? c0 + k - 6 : c0 + k - 3) : c0 + k;
          loss_simple_elem(f, P.kind, P.eps, P.grad_scale, !f.one_channel || c == 0, xs[k], ts[k], ps[k], ds[k], loss);
        }
        if (f.fused) pv[v] = float4{ps[0], ps[1], ps[2], ps[3]};
        dv[v] = float4{ds[0], ds[1], ds[2], ds[3]};
      };
      one(v0, xa, ta);
      if (has1) one(v1, xb, tb);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) { dd_det_wait(); atomicAdd(P.loss_out, red[0] + red[1] + red[2] + red[3]); }
  dd_det_end();
}

// Mean-only descriptors with combined / image terms (TrainingExample.json: feature x 1, combined x 5, image x 10; no variation term): one thread
// per pixel as in loss_head_kernel, but the pixel's predictions, targets and gradients of ALL features live in the thread's own LDS column
// ([feature][p0 p1 p2 t0 t1 t2 g0 g1 g2][lane]) -- every global value is read once (the loads of all features are in flight together) and every
// gradient written once.  loss_head_kernel re-read a feature per term and accumulated the combined / image gradients with global read-modify-
// writes: ~300 dependent memory operations per thread, 66 us even for the 8 192 pixels of the coarsest scale.  Same terms in the same order,
// so dpred is bit-identical to loss_head_kernel's.  The inverse standardization fuses exactly as in loss_simple_kernel.
__device__ __forceinline__ float invert_fwd(float x, float mean, float std, int log1p, float& dinv) {
  float z = x * std + mean;
  dinv = std;
  if (log1p) {
    dinv *= (z != 0.f) ? expf(fabsf(z)) : 0.f;
    z = z > 0.f ? expm1f(z) : (z < 0.f ? -expm1f(-z) : 0.f);
  }
  return z;
}
__global__ __launch_bounds__(64) void loss_general_kernel(const dd_loss_desc d, long npix, float inv_count, float grad_scale, float* __restrict__ loss_out) {
  extern __shared__ float lg_sm[];                       // [n_features][9][64]
  const int lane = threadIdx.x;
  float loss = 0.f;
  auto P = [&](int f, int c) -> float& { return lg_sm[(f * 9 + c) * 64 + lane]; };
  auto Tg = [&](int f, int c) -> float& { return lg_sm[(f * 9 + 3 + c) * 64 + lane]; };
  auto G = [&](int f, int c) -> float& { return lg_sm[(f * 9 + 6 + c) * 64 + lane]; };
  // mean term of one source: adds to `loss`, returns dLoss/dvalue
  auto term = [&](const float (&p)[3], const float (&t)[3], int nch, float w, float (&g)[3]) {
    g[0] = g[1] = g[2] = 0.f;
    if (w == 0.f) return;
    for (int ch = 0; ch < nch; ++ch) {
      float l, dl;
      loss_term(d.kind, d.epsilon, p[ch], t[ch], &l, &dl);
      loss += w * l;
      g[ch] += w * dl * grad_scale;
    }
  };
  auto masked_w = [&](float w, int mask_f, int src) -> float {      // masked_pixel_weight with the mask feature's target taken from LDS
    if (w == 0.f || mask_f < 0 || d.mask_sums == nullptr) return 0.f;
    const float msum = d.mask_sums[src];
    float sa = 0.f;
    for (int c = 0; c < d.nch[mask_f]; ++c) sa += fabsf(Tg(mask_f, c));
    return msum > 0.f ? w * (sa > 0.f ? 1.f : 0.f) / msum : 0.f;
  };
  auto comb_val = [&](int k, float (&p)[3], float (&t)[3]) {
    const int fc = d.comb[k][0], fd = d.comb[k][1], fi = d.comb[k][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = P(fc, c) * (P(fd, c) + P(fi, c)); t[c] = Tg(fc, c) * (Tg(fd, c) + Tg(fi, c)); }
  };
  for (long i0 = (long)blockIdx.x * 64; i0 < npix; i0 += (long)gridDim.x * 64) {
    const long i = i0 + lane;
    const bool live = i < npix;
    const long ii = live ? i : npix - 1;
    // every feature's prediction and target of this pixel -> LDS (1-channel passes broadcast channel 0, Training.py:422-426); the loads of FB
    // features are requested before the first one is used (a thread has one wave per SIMD around it: 39 KiB of LDS per 64 pixels at 17 features)
    constexpr int FB = 4;
    for (int f0 = 0; f0 < d.n_features; f0 += FB) {
      float pv[FB][3], tv[FB][3];
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int f = min(f0 + u, d.n_features - 1);
        const bool one = d.nch[f] == 1, fused = d.pred_std[f] != nullptr;
        const float* pp = fused ? d.pred_std[f] + ii * 3 : d.pred[f] + ii * d.pred_ld[f];
        const float* tp = d.target[f] + ii * d.target_ld[f];
#pragma unroll
        for (int c = 0; c < 3; ++c) { pv[u][c] = pp[(one && !fused) ? 0 : c]; tv[u][c] = tp[one ? 0 : c]; }
      }
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int f = f0 + u;
        if (f >= d.n_features) break;
        if (d.pred_std[f]) {
          float dinv;
#pragma unroll
          for (int c = 0; c < 3; ++c) pv[u][c] = invert_fwd(pv[u][c], d.inv_mean[f], d.inv_std[f], d.inv_log1p[f], dinv);
          if (live) { float* po = d.pred_inv[f] + ii * 3; po[0] = pv[u][0]; po[1] = pv[u][1]; po[2] = pv[u][2]; }
          if (d.nch[f] == 1) pv[u][1] = pv[u][2] = pv[u][0];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { P(f, c) = pv[u][c]; Tg(f, c) = tv[u][c]; }
      }
    }
    if (live) {
      for (int f = 0; f < d.n_features; ++f) {
        const float pv[3] = {P(f, 0), P(f, 1), P(f, 2)}, tv[3] = {Tg(f, 0), Tg(f, 1), Tg(f, 2)};
        float g[3];
        term(pv, tv, d.nch[f], d.weight[f] * inv_count + masked_w(d.masked_weight[f], d.mask_feature[f], f), g);
        G(f, 0) = g[0]; G(f, 1) = g[1]; G(f, 2) = g[2];
      }
      if (d.n_combined > 0 || d.n_image_features > 0) {
        float dimg[3] = {0.f, 0.f, 0.f};
        const bool use_image = (d.image_weight != 0.f || d.image_var_weight != 0.f) && (d.n_image_combined > 0 || d.n_image_features > 0);
        if (use_image) {
          float pv[3] = {0.f, 0.f, 0.f}, tv[3] = {0.f, 0.f, 0.f};
          for (int j = 0; j < d.n_image_combined; ++j) {
            float a[3], b[3];
            comb_val(d.image_combined[j], a, b);
#pragma unroll
            for (int c = 0; c < 3; ++c) { pv[c] += a[c]; tv[c] += b[c]; }
          }
          for (int j = 0; j < d.n_image_features; ++j) {
            const int f = d.image_features[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) { pv[c] += P(f, c); tv[c] += Tg(f, c); }
          }
          term(pv, tv, 3, d.image_weight * inv_count, dimg);
          for (int j = 0; j < d.n_image_features; ++j) {
            const int f = d.image_features[j];
            for (int c = 0; c < 3; ++c) G(f, d.nch[f] == 1 ? 0 : c) += dimg[c];
          }
        }
        for (int k = 0; k < d.n_combined; ++k) {
          const int fc = d.comb[k][0], fd = d.comb[k][1], fi = d.comb[k][2];
          bool in_image = false;
          for (int j = 0; j < d.n_image_combined; ++j) in_image |= (d.image_combined[j] == k);
          float pv[3], tv[3], g[3];
          comb_val(k, pv, tv);
          term(pv, tv, 3, d.comb_weight[k] * inv_count + masked_w(d.comb_masked_weight[k], d.comb_mask_feature[k], DD_MAX_FEATURES + k), g);
          for (int ch = 0; ch < 3; ++ch) {
            const float gt = g[ch] + ((use_image && in_image) ? dimg[ch] : 0.f);
            const int cc = d.nch[fc] == 1 ? 0 : ch, cd = d.nch[fd] == 1 ? 0 : ch, ci = d.nch[fi] == 1 ? 0 : ch;
            const float pc = P(fc, ch);
            G(fc, cc) += gt * (P(fd, ch) + P(fi, ch));
            G(fd, cd) += gt * pc;
            G(fi, ci) += gt * pc;
          }
        }
      }
      // dL/dp, or -- fused -- dL/dx = dL/dp . dp/dx with the chain factor of invert_std_bwd_kernel recomputed from x (L2 hits, FB features at a time)
      for (int f0 = 0; f0 < d.n_features; f0 += FB) {
        float xv[FB][3];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
          const int f = min(f0 + u, d.n_features - 1);
          const float* xp = d.pred_std[f] ? d.pred_std[f] + i * 3 : d.target[f] + i * d.target_ld[f];      // (unfused: any valid address, unused)
#pragma unroll
          for (int c = 0; c < 3; ++c) xv[u][c] = xp[d.pred_std[f] ? c : 0];
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
          const int f = f0 + u;
          if (f >= d.n_features) break;
          float* dp = d.dpred[f] + i * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float dinv = 1.f;
            if (d.pred_std[f]) (void)invert_fwd(xv[u][c], d.inv_mean[f], d.inv_std[f], d.inv_log1p[f], dinv);
            dp[c] = G(f, c) * dinv;
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  if (lane == 0) { dd_det_wait(); atomicAdd(loss_out, loss); }
  dd_det_end();
}

extern "C" int dd_loss_head(const dd_loss_desc* desc, int B, int H, int W, float* loss_out, float grad_scale, dd_stream stream) {
  DD_REQUIRE(desc && loss_out && desc->n_features > 0 && desc->n_features <= DD_MAX_FEATURES && desc->n_combined <= DD_MAX_COMBINED,
             "dd_loss_head: bad descriptor");
  DD_REQUIRE(desc->kind >= 1 && desc->kind <= 5, "dd_loss_head: unknown loss kind %d", desc->kind);
  const long npix = (long)B * H * W;
  const long npairs = (long)B * ((long)H * (W - 1) + (long)(H - 1) * W);
  // features-only descriptors take the flat float4 kernel (DD_LOSS_SIMPLE=0: the general kernel); the fused inverse standardization exists
  // only there
  bool fused_any = false, simple = desc->n_combined == 0 && (npix * 3) % 4 == 0;
  if (desc->image_weight != 0.f || desc->image_var_weight != 0.f) simple = simple && desc->n_image_combined == 0 && desc->n_image_features == 0;
  for (int f = 0; f < desc->n_features; ++f) {
    fused_any = fused_any || desc->pred_std[f] != nullptr;
    const bool active = desc->weight[f] != 0.f || desc->pred_std[f] != nullptr;
    if (desc->var_weight[f] != 0.f || (desc->masked_weight[f] != 0.f && desc->mask_feature[f] >= 0 && desc->mask_sums)) simple = false;
    if (active && (desc->pred_ld[f] != 3 || desc->target_ld[f] != 3 || (desc->nch[f] != 1 && desc->nch[f] != 3))) simple = false;
    if (active && (((uintptr_t)desc->pred[f] | (uintptr_t)desc->target[f] | (uintptr_t)desc->dpred[f] | (uintptr_t)desc->pred_std[f] | (uintptr_t)desc->pred_inv[f]) & 15)) simple = false;
    if (desc->pred_std[f]) DD_REQUIRE(desc->pred_inv[f] && desc->inv_std[f] > 0.f, "dd_loss_head: pred_std[%d] needs pred_inv and a positive inv_std", f);
  }
  static const bool simple_on = [] { const char* e = getenv("DD_LOSS_SIMPLE"); return !(e && e[0] == '0'); }();
  // no variation term anywhere: every term is a function of ONE pixel (loss_general_kernel; DD_LOSS_GENERAL=0: the older kernel)
  bool pixel_local = desc->image_var_weight == 0.f;
  for (int f = 0; f < desc->n_features; ++f) pixel_local = pixel_local && desc->var_weight[f] == 0.f;
  for (int k = 0; k < desc->n_combined; ++k) pixel_local = pixel_local && desc->comb_var_weight[k] == 0.f;
  static const bool general_on = [] { const char* e = getenv("DD_LOSS_GENERAL"); return !(e && e[0] == '0'); }();
  DD_REQUIRE(!fused_any || simple || pixel_local, "dd_loss_head: the fused inverse standardization (pred_std) needs a descriptor without variation terms");
  if (simple && (simple_on || fused_any)) {
    LossSimpleP P;
    memset(&P, 0, sizeof(P));
    for (int f = 0; f < desc->n_features; ++f) {
      const bool fused = desc->pred_std[f] != nullptr;
      if (desc->weight[f] == 0.f && !fused) continue;       // generated passes: their scratch dpred stays zero
      LossSimpleF& o = P.f[P.n++];
      o.x = fused ? desc->pred_std[f] : desc->pred[f];
      o.t = desc->target[f]; o.p = fused ? desc->pred_inv[f] : nullptr; o.d = desc->dpred[f];
      o.w = desc->weight[f] * (1.f / (float)npix);
      o.mean = desc->inv_mean[f]; o.std = desc->inv_std[f]; o.log1p = desc->inv_log1p[f]; o.one_channel = desc->nch[f] == 1; o.fused = fused;
    }
    if (P.n == 0) return DD_OK;
    P.kind = desc->kind; P.eps = desc->epsilon; P.grad_scale = grad_scale; P.nvec = npix * 3 / 4; P.loss_out = loss_out;
    dd_det_sync();
    const long want = (P.nvec + 511) / 512;
    const long cap = 1024L;      // (one atomic per workgroup into ONE address: 2 048 of them measured 8 us slower than 1 024 over the three scales)
    hipLaunchKernelGGL(loss_simple_kernel, dim3((unsigned)(want < cap ? (want < 1 ? 1 : want) : cap)), dim3(256), 0, S(stream), P);
    DD_LAUNCH_CHECK();
    return DD_OK;
  }
  if (pixel_local && (general_on || fused_any)) {
    const size_t lds = (size_t)desc->n_features * 9 * 64 * sizeof(float);
    dd_allow_max_lds(reinterpret_cast<const void*>(loss_general_kernel), 96 * 1024);
    dd_det_sync();
    const long want = (npix + 63) / 64;
    hipLaunchKernelGGL(loss_general_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(64), lds, S(stream), *desc, npix, 1.f / (float)npix, grad_scale, loss_out);
    DD_LAUNCH_CHECK();
    return DD_OK;
  }
  dd_det_sync();
  hipLaunchKernelGGL(loss_head_kernel, dim3(min(grid_for(npix), 2048u)), dim3(256), 0, S(stream), *desc, npix, H, W, 1.f / (float)npix,
                     npairs > 0 ? 1.f / (float)npairs : 0.f, grad_scale, loss_out);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ Adam (TF form)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr_t, float b1, float b2, float eps, float gs) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
extern "C" int dd_adam_step(float* params, const float* grads, float* m, float* v, long n, float lr_t, float beta1, float beta2,
                            float eps, float grad_scale, dd_stream stream) {
  DD_REQUIRE(params && grads && m && v && n > 0, "dd_adam_step: bad arguments");
  hipLaunchKernelGGL(adam_kernel, dim3(min(grid_for(n), 2048u)), dim3(256), 0, S(stream), params, grads, m, v, n, lr_t, beta1, beta2, eps, grad_scale);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ stitch
// One workgroup = one row of one entry's crop window: cw * C contiguous floats on both sides (round 5: was one workgroup per entry walking its
// ~100 x 100 x 3 elements with a 64-bit division each -- 209 workgroups, 61 us for the 66 MB of a 1080p frame).
__global__ void stitch_kernel(const float* __restrict__ tiles, int ts, int ldt, float* __restrict__ frame, int fh, int fw, int ldf, int C,
                              const dd_stitch_entry* __restrict__ table) {
  const dd_stitch_entry e = table[blockIdx.x];
  const int y = blockIdx.y, ch = e.crop_y1 - e.crop_y0, cw = e.crop_x1 - e.crop_x0;
  if (y >= ch) return;
  float* dst = frame + (((long)e.dst_img * fh + e.dst_y + y) * fw + e.dst_x) * ldf;
  const float* src = tiles + (((long)e.tile * ts + e.crop_y0 + y) * ts + e.crop_x0) * ldt;
  if (C == ldf && C == ldt) {
    for (int i = threadIdx.x; i < cw * C; i += blockDim.x) dst[i] = src[i];
    return;
  }
  for (int i = threadIdx.x; i < cw * C; i += blockDim.x) {
    const int x = i / C, c = i - x * C;
    dst[(long)x * ldf + c] = src[(long)x * ldt + c];
  }
}
extern "C" int dd_stitch(const float* tiles, int tile_size, int ldt, float* frame, int frame_h, int frame_w, int ldf, int C,
                         const dd_stitch_entry* table, int n_entries, dd_stream stream) {
  DD_REQUIRE(tiles && frame && table && n_entries > 0 && tile_size > 0 && tile_size <= 65535, "dd_stitch: bad arguments");
  hipLaunchKernelGGL(stitch_kernel, dim3(n_entries, tile_size), dim3(128), 0, S(stream), tiles, tile_size, ldt, frame, frame_h, frame_w, ldf, C, table);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// One workgroup row = one tile row: threads walk the T*C contiguous floats of the row (float4 when both sides allow it).
__global__ void extract_tiles_kernel(const float* __restrict__ frame, int fw, int ldf, int C, float* __restrict__ tiles, int ts, int ldt,
                                     const int* __restrict__ origins) {
  const int tile = blockIdx.y, row = blockIdx.x;
  const int oy = origins[2 * tile], ox = origins[2 * tile + 1];
  const float* src = frame + ((long)(oy + row) * fw + ox) * ldf;
  float* dst = tiles + ((long)tile * ts + row) * ts * ldt;
  if (C == ldf && C == ldt && ((ts * C) & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    const int n4 = ts * C / 4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    return;
  }
  for (int i = threadIdx.x; i < ts * C; i += blockDim.x) {
    const int x = i / C, c = i - x * C;
    dst[(long)x * ldt + c] = src[(long)x * ldf + c];
  }
}
extern "C" int dd_extract_tiles(const float* frame, int frame_h, int frame_w, int ldf, int C, float* tiles, int tile_size, int ldt,
                                const int* origins_yx, int n_tiles, dd_stream stream) {
  DD_REQUIRE(frame && tiles && origins_yx && n_tiles > 0 && C > 0 && C <= ldf && C <= ldt, "dd_extract_tiles: bad arguments");
  DD_REQUIRE(tile_size > 0 && tile_size <= frame_h && tile_size <= frame_w, "dd_extract_tiles: tile %d does not fit the %dx%d frame", tile_size,
             frame_h, frame_w);
  hipLaunchKernelGGL(extract_tiles_kernel, dim3(tile_size, n_tiles), dim3(128), 0, S(stream), frame, frame_w, ldf, C, tiles, tile_size, ldt, origins_yx);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// Bit-exact with the numpy chain of Prediction.py:469-481: every np.multiply / np.add is one IEEE fp32 rounding, taken in the
// reference's order (colour * (direct + indirect); then image = ((((d + g) + s) + t) + volume direct) + ...).  __fmul_rn / __fadd_rn
// keep hipcc from contracting the products into FMAs, which would round once instead of twice.
__global__ void recombine_kernel(const dd_recombine_desc d, long npix) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= npix * 3) return;
  float img = 0.f;
  bool first = true;
  for (int k = 0; k < d.n_triples; ++k) {
    const float v = __fmul_rn(d.color[k][i], __fadd_rn(d.direct[k][i], d.indirect[k][i]));
    if (d.combined[k]) d.combined[k][i] = v;
    img = first ? v : __fadd_rn(img, v);
    first = false;
  }
  for (int j = 0; j < d.n_singles; ++j) {
    img = first ? d.single[j][i] : __fadd_rn(img, d.single[j][i]);
    first = false;
  }
  d.image[i] = img;
}
extern "C" int dd_recombine(const dd_recombine_desc* desc, long npix, dd_stream stream) {
  DD_REQUIRE(desc && desc->image && desc->n_triples >= 0 && desc->n_triples <= 4 && desc->n_singles >= 0 && desc->n_singles <= 8 && npix > 0,
             "dd_recombine: bad descriptor");
  hipLaunchKernelGGL(recombine_kernel, dim3(grid_for(npix * 3)), dim3(256), 0, S(stream), *desc, npix);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ executor helpers
template <typename T>
__global__ void masked_add_kernel(T* __restrict__ dst, int lddst, const T* __restrict__ src, int ldsrc, const T* __restrict__ mask, int ldmask,
                                  int C, long npix, int accumulate) {
  const int cg = C >> 2;
  const long total = npix * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * 4;
  const long pix = i / cg;
  float v[4];
  load4<T>(src + pix * ldsrc + c, v);
  if (mask) {
    float m[4];
    load4<T>(mask + pix * ldmask + c, m);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
  }
  T* d = dst + pix * lddst + c;
  if (accumulate) {
    float o[4];
    load4<T>(d, o);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += o[e];
  }
  store4<T>(d, v);
}
extern "C" int dd_masked_add(void* dst, int lddst, const void* src, int ldsrc, const void* mask, int ldmask, int C, long npix,
                             int accumulate, int dtype, dd_stream stream) {
  DD_REQUIRE(dst && src && C % 4 == 0 && lddst % 4 == 0 && ldsrc % 4 == 0 && (!mask || ldmask % 4 == 0), "dd_masked_add: C, ld must be multiples of 4");
  const long total = npix * (C / 4);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(masked_add_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (T*)dst, lddst, (const T*)src, ldsrc, (const T*)mask, ldmask, C, npix, accumulate));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename TS, typename TD>
__global__ void convert_channels_kernel(const TS* __restrict__ src, int ldsrc, TD* __restrict__ dst, int lddst, int nch, int dst_pad, long npix) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const TS* s = src + i * ldsrc;
  TD* d = dst + i * lddst;
  for (int c = 0; c < nch; ++c) d[c] = Elem<TD>::from_f32(Elem<TS>::to_f32(s[c]));
  for (int c = nch; c < dst_pad; ++c) d[c] = Elem<TD>::from_f32(0.f);
}
extern "C" int dd_convert_channels(const void* src, int src_dtype, int ldsrc, void* dst, int dst_dtype, int lddst, int nch, int dst_pad,
                                   long npix, dd_stream stream) {
  DD_REQUIRE(src && dst && nch > 0 && nch <= ldsrc && dst_pad <= lddst && npix > 0, "dd_convert_channels: bad arguments");
  const dim3 g(grid_for(npix)), b(256);
  DD_REQUIRE(dd_dtype_ok(src_dtype) && dd_dtype_ok(dst_dtype), "dd_convert_channels: bad dtype");
  DD_DISPATCH_DTYPE(src_dtype, TS, DD_DISPATCH_DTYPE(dst_dtype, TD,
      hipLaunchKernelGGL((convert_channels_kernel<TS, TD>), g, b, 0, S(stream), (const TS*)src, ldsrc, (TD*)dst, lddst, nch, dst_pad, npix)));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void zero_stuff_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int C, int B, int H, int W) {
  const int cg = C >> 2;
  const long total = (long)B * 2 * H * 2 * W * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * 4;
  long r = i / cg;
  const int ox = (int)(r % (2 * W)); r /= 2 * W;
  const int oy = (int)(r % (2 * H));
  const int b = (int)(r / (2 * H));
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if ((oy & 1) && (ox & 1)) load4<T>(x + (((long)b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + c, v);
  store4<T>(y + (((long)b * 2 * H + oy) * 2 * W + ox) * ldy + c, v);
}
extern "C" int dd_zero_stuff(const void* x, int ldx, void* y, int ldy, int C, int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(x && y && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "dd_zero_stuff: C, ld must be multiples of 4");
  const long total = (long)B * 4 * H * W * (C / 4);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(zero_stuff_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)x, ldx, (T*)y, ldy, C, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// Space-to-depth of a fine-grid tensor (2H x 2W) by output parity: s[b][i][j][(py*2 + px)*cp + c] = y[b][2i + py][2j + px][c] for c < C, zero for
// C <= c < cp.  One thread = 4 channels of one plane of one coarse pixel.
template <typename T>
__global__ void space_to_depth2_kernel(const T* __restrict__ y, int ldy, T* __restrict__ s, int lds, int C, int cp, int B, int H, int W) {
  const int cg = cp >> 2;
  const long total = (long)B * H * W * 4 * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * 4;
  long r = i / cg;
  const int plane = (int)(r & 3); r >>= 2;
  const int x = (int)(r % W); r /= W;
  const int yy = (int)(r % H);
  const int b = (int)(r / H);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    load4<T>(y + (((long)b * 2 * H + 2 * yy + (plane >> 1)) * 2 * W + 2 * x + (plane & 1)) * ldy + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (c + e >= C) v[e] = 0.f;
  }
  store4<T>(s + (((long)b * H + yy) * W + x) * lds + plane * cp + c, v);
}
extern "C" int dd_space_to_depth2(const void* y, int ldy, void* s, int lds, int C, int cp, int B, int H, int W, int dtype, dd_stream stream) {
  DD_REQUIRE(y && s && C > 0 && cp >= C && cp % 4 == 0 && ldy % 4 == 0 && lds % 4 == 0 && lds >= 4 * cp && ldy >= (C + 3) / 4 * 4,
             "dd_space_to_depth2: C=%d cp=%d ldy=%d lds=%d (cp, ld multiples of 4; lds >= 4 cp; the 4-channel group holding channel C - 1 readable)", C, cp, ldy, lds);
  const long total = (long)B * H * W * cp;
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(space_to_depth2_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)y, ldy, (T*)s, lds, C, cp, B, H, W));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

template <typename T>
__global__ void zero_unstuff_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, const T* __restrict__ mask, int ldmask,
                                    int C, int B, int H, int W, int accumulate) {
  const int cg = C >> 2;
  const long total = (long)B * H * W * cg;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cg) * 4;
  long r = i / cg;
  const int x = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const int b = (int)(r / H);
  float v[4];
  load4<T>(dy + (((long)b * 2 * H + 2 * y + 1) * 2 * W + 2 * x + 1) * lddy + c, v);
  const long pix = ((long)b * H + y) * W + x;
  if (mask) {
    float m[4];
    load4<T>(mask + pix * ldmask + c, m);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : 0.f;
  }
  T* d = dx + pix * lddx + c;
  if (accumulate) {
    float o[4];
    load4<T>(d, o);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += o[e];
  }
  store4<T>(d, v);
}
extern "C" int dd_zero_unstuff(const void* dy, int lddy, void* dx, int lddx, const void* mask, int ldmask, int C, int B, int H, int W,
                               int accumulate, int dtype, dd_stream stream) {
  DD_REQUIRE(dy && dx && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "dd_zero_unstuff: C, ld must be multiples of 4");
  const long total = (long)B * H * W * (C / 4);
  DD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(zero_unstuff_kernel<T>, dim3(grid_for(total)), dim3(256), 0, S(stream), (const T*)dy, lddy, (T*)dx, lddx, (const T*)mask, ldmask, C, B, H, W, accumulate));
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// ------------------------------------------------------------------------------------------------ probes
__global__ void probe_tr16_kernel(const uint16_t* __restrict__ image, const int32_t* __restrict__ addr, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = image[i];
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4_t* lp;
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((char*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
extern "C" int dd_probe_tr16(const uint16_t* lds_image_4096, const int32_t* lane_byte_addr_64, uint16_t* out_64x4, dd_stream stream) {
  DD_REQUIRE(lds_image_4096 && lane_byte_addr_64 && out_64x4, "dd_probe_tr16: null pointer");
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, S(stream), lds_image_4096, lane_byte_addr_64, out_64x4);
  DD_LAUNCH_CHECK();
  return DD_OK;
}
