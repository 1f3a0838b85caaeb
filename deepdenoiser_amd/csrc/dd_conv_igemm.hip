// Implicit-GEMM convolution on CDNA4 MFMA: forward and data-gradient of every conv-like layer of the denoiser.
//
// Reference seams replaced (file:line in /root/reference): tf.layers.conv2d 3x3 / 1x1 SAME (TensorFlow/UNet.py:29-31,
// Tiramisu.py:35-37,50-52,77-79, Architecture.py:238-243, MultiScalePrediction.py:64-66,73-75,88-90),
// tf.layers.conv2d_transpose 2x2/s2 (UNet.py:56-58) and their TF-autodiff input gradients (Training.py:701-702).
//
// Mapping.  GEMM: D[cout][pixel] = sum_{tap, cin} Wp[tap][cout][cin] * X[pixel (+) tap][cin].
//   "A" operand = packed weights (rows = cout), "B" operand = pixels (cols): after the MFMA every lane holds 4 CONSECUTIVE
//   output channels of one pixel -> packed 8/16-byte NHWC stores with the bias/residual/ReLU/mask/accumulate epilogue fused.
// Schedule.  Persistent workgroups (one per CU) walk an XCD-local list of 16x16-pixel tiles for a fixed block of NT*16 output channels.
//   A "unit" = one 128-byte K-slice (64 bf16 / 32 f32 channels) of one tile: its (16+2)^2 halo patch is staged in LDS ONCE and reused
//   by all 9 taps.  Two kernels:
//   * conv_igemm_ws_kernel (bf16, NT <= 4, weights LDS-resident -- every bf16 layer of the shipped configurations): 8 waves; waves 0-3
//     only read fragments and issue MFMAs and park the finished tile in an LDS stage, waves 4-7 prefetch the next patch into registers,
//     drain the previous tile's stage to HBM (bias / residual / ReLU / mask / accumulate fused) and write the next patch to LDS.  Two
//     barriers per unit; 288 MFMAs per MFMA wave back to back for a 64->64 layer.
//   * conv_igemm_kernel (f32 parity path and anything the above does not take): 4 waves doing both jobs, the next patch's global loads
//     in flight during the MFMAs; weights resident when they fit, otherwise the next tap's slab is register-prefetched during the current
//     tap's MFMAs.
// LDS rows are 128 B with a 16-byte-slot XOR swizzle (dd_common.h: lds_off / lds_pix_off) => ds_read_b128 fragment reads are
// bank-conflict free (SQ_LDS_BANK_CONFLICT = 7 % of SQ_LDS_IDX_ACTIVE, all from the stage).
#include <stdlib.h>

#include "dd_common.h"

#ifdef DD_PROFILE_PHASES
__device__ unsigned long long dd_phase_cycles[16];
#define PHASE_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PHASE_ADD(i, a, b) if (blockIdx.x == 0 && tid == 0) dd_phase_cycles[i] += (b) - (a)
#define PHASE_ADD_T(i, a, b, t) if (blockIdx.x == 0 && tid == (t)) dd_phase_cycles[i] += (b) - (a)
extern "C" int dd_debug_phases(unsigned long long* out8, int reset) {
  if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(dd_phase_cycles), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dd_phase_cycles), z, sizeof(z)); }
  return 0;
}
#else
#define PHASE_T(var)
#define PHASE_ADD(i, a, b)
#define PHASE_ADD_T(i, a, b, t)
#endif

namespace {

struct ConvP {
  const void* x; const void* wp; const float* bias; const void* res; const void* mask; void* y;
  int ldx, cin, k_pad, n_pad, ldres, ldmask, ldy, n;
  int B, H, W, taps, flags, nbias;
  int tiles_x, tiles_y, nblk;   // tile grid and number of output-channel blocks
  int kchunks;                  // number of 64-byte K chunks (k_pad*sizeof(T)/64)
  int hin, win;                 // input image size
  int hout, wout;               // output image size
  int total_tiles;              // B * tiles_y * tiles_x
};

template <bool HALO> struct PatchDim {
  static constexpr int PH = HALO ? DD_TILE + 2 : DD_TILE;
  static constexpr int NPIX = PH * PH;
  static constexpr int ITERS = (NPIX * 8 + 255) / 256;
};

// Per-thread staging plan of a patch K-slice, computed ONCE per workgroup (the kernel is VALU-issue bound: per-vector index
// arithmetic must not be repeated per tile).  Vector `it` of this thread covers patch pixel (py,px) = yx[it]; rel[it] is its
// element offset from the patch origin pixel in the input image; lds[it] its byte offset in the LDS image.
// A vector memory instruction costs the same issue time whether 3 or 8 of a pixel's 16-byte slots are real (the rest would read the
// zero page), so narrow layers (K <= 32: one 64-byte chunk) use a COMPACT plan: 4 vectors per pixel instead of 8, half the instructions.
template <bool HALO> struct PatchPlan {
  int yx[PatchDim<HALO>::ITERS];     // (py << 8) | px, or -1 when the vector is outside the patch
  int rel[PatchDim<HALO>::ITERS];    // element offset from the patch origin pixel, INCLUDING the vector's channel offset
  int lds[PatchDim<HALO>::ITERS];
  int slot;                          // 16-byte slot of this thread's vectors within their pixel (256 % 8 == 0: the same for all)
  int n_it;                          // vectors this thread really has (wave-uniform upper bound): the rest are skipped
};

template <typename T, bool HALO>
__device__ __forceinline__ void patch_plan(PatchPlan<HALO>& pl, const ConvP& p, int tid) {
  constexpr int PH = PatchDim<HALO>::PH, NPIX = PatchDim<HALO>::NPIX, ITERS = PatchDim<HALO>::ITERS;
  const bool compact = p.kchunks == 1;
  pl.n_it = compact ? (NPIX * 4 + 255) / 256 : ITERS;
  pl.slot = compact ? tid & 3 : tid & 7;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = tid + it * 256;
    const int pix = compact ? i >> 2 : i >> 3, slot = compact ? i & 3 : i & 7;
    const int py = pix / PH, px = pix - py * PH;
    pl.yx[it] = pix < NPIX ? ((py << 8) | px) : -1;
    pl.rel[it] = (py * p.win + px) * p.ldx + slot * Elem<T>::PER16;
    pl.lds[it] = lds_pix_off(py, px, PH, slot);
  }
}

struct TileCoord { int b, y0, x0; };
__device__ __forceinline__ TileCoord tile_coord(const ConvP& p, int tile) {
  const int per_img = p.tiles_y * p.tiles_x;
  TileCoord t;
  t.b = tile / per_img;
  const int rem = tile - t.b * per_img;
  const int ty = rem / p.tiles_x;
  t.y0 = ty * DD_TILE;
  t.x0 = (rem - ty * p.tiles_x) * DD_TILE;
  return t;
}

// Issue the global loads of one patch K-slice into registers.  Interior tiles (the whole patch inside the image, no gather):
// one 64-bit add per vector.  Edge tiles / the 2x2 gather: per-vector bounds test selecting the zero page.
template <typename T, bool HALO>
__device__ __forceinline__ void patch_load(uint4 (&reg)[PatchDim<HALO>::ITERS], const PatchPlan<HALO>& pl, const T* __restrict__ X, const ConvP& p,
                                           int tile, int slice, int tap, int nslots, int tid) {
  constexpr int ITERS = PatchDim<HALO>::ITERS, PH = PatchDim<HALO>::PH;
  constexpr int PER16 = Elem<T>::PER16, KC = DD_LDS_ROW / (int)sizeof(T);
  const TileCoord tc = tile_coord(p, tile);
  const int oy = tc.y0 - (HALO ? 1 : 0), ox = tc.x0 - (HALO ? 1 : 0);
  const bool gather = (p.flags & DD_GATHER2X2) != 0;
  const int ch0 = slice * KC;
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
  const bool interior = !gather && oy >= 0 && ox >= 0 && oy + PH <= p.H && ox + PH <= p.W;
  // vector `it` is real iff it lies in the patch and its channels exist: slot < nslots (this K-slice) and channel < cin
  const bool slot_ok = pl.slot < nslots && ch0 + pl.slot * PER16 < p.cin;
  auto vec_ok = [&](int it) { return slot_ok && pl.yx[it] >= 0; };
  if (interior) {
    const T* base = X + (((long)tc.b * p.hin + oy) * p.win + ox) * p.ldx + ch0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      if (it < pl.n_it) reg[it] = *reinterpret_cast<const uint4*>(vec_ok(it) ? base + pl.rel[it] : zero);
  } else if (!gather) {
    // Edge tile: ALL addresses first (out-of-image vectors -> the zero page), then the loads back to back.  Computing each address right
    // before its load lets hipcc reuse the destination registers of loads still in flight as address temporaries, and every such reuse
    // is an s_waitcnt vmcnt() on the older loads AND stores (gfx9 counts stores in vmcnt): the edge tiles then pay a memory round trip
    // per vector, and with a static tile assignment the slowest (all-edge) workgroup sets the launch time.
    const T* base = X + (((long)tc.b * p.hin + oy) * p.win + ox) * p.ldx + ch0;      // may point outside the image; only valid vectors use it
    const T* ptr[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int yx = pl.yx[it];
      const bool ok = vec_ok(it) && (unsigned)(oy + (yx >> 8)) < (unsigned)p.hin && (unsigned)(ox + (yx & 255)) < (unsigned)p.win;
      ptr[it] = ok ? base + pl.rel[it] : zero;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      if (it < pl.n_it) reg[it] = *reinterpret_cast<const uint4*>(ptr[it]);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    const int sy = 2, ay = tap >> 1, ax = tap & 1;
    const T* base = X + (long)tc.b * p.hin * p.win * p.ldx + ch0;
    const int ylo = HALO ? -1 : 0, yhi = p.H + (HALO ? 1 : 0), xhi = p.W + (HALO ? 1 : 0);
    const T* ptr[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int yx = pl.yx[it];
      const int ly = oy + (yx >> 8), lx = ox + (yx & 255);
      const int gy = ly * sy + ay, gx = lx * sy + ax;
      const bool ok = vec_ok(it) && ly >= ylo && lx >= ylo && ly < yhi && lx < xhi && gy >= 0 && gx >= 0 && gy < p.hin && gx < p.win;
      ptr[it] = ok ? base + ((long)gy * p.win + gx) * p.ldx + pl.slot * PER16 : zero;
    }
    __builtin_amdgcn_sched_barrier(0);      // as on edge tiles: all addresses first, then the loads back to back
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      if (it < pl.n_it) reg[it] = *reinterpret_cast<const uint4*>(ptr[it]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, bool HALO>
__device__ __forceinline__ void patch_store(char* lds, const uint4 (&reg)[PatchDim<HALO>::ITERS], const PatchPlan<HALO>& pl, bool in_relu) {
#pragma unroll
  for (int it = 0; it < PatchDim<HALO>::ITERS; ++it) {
    uint4 v = reg[it];
    if (in_relu) v = relu16<T>(v);
    if (it < pl.n_it && pl.yx[it] >= 0) *reinterpret_cast<uint4*>(lds + pl.lds[it]) = v;
  }
}

// weight slab of one (tap, slice): NT*16 rows of 128 B
template <typename T, int NT>
__device__ __forceinline__ void slab_load(uint4 (&reg)[(NT + 1) / 2], const T* __restrict__ Wp, const ConvP& p, int n0, int tap, int slice, int nslots, int tid) {
  constexpr int KC = DD_LDS_ROW / (int)sizeof(T);
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
#pragma unroll
  for (int it = 0; it < (NT + 1) / 2; ++it) {
    const int i = tid + it * 256;
    const int row = i >> 3, slot = i & 7;
    const int ng = n0 + row;
    const bool ok = row < NT * 16 && slot < nslots && ng < p.n_pad;
    reg[it] = *reinterpret_cast<const uint4*>(ok ? Wp + ((long)tap * p.n_pad + ng) * p.k_pad + slice * KC + slot * Elem<T>::PER16 : zero);
  }
}
template <int NT>
__device__ __forceinline__ void slab_store(char* lds, const uint4 (&reg)[(NT + 1) / 2], int tid) {
#pragma unroll
  for (int it = 0; it < (NT + 1) / 2; ++it) {
    const int i = tid + it * 256;
    const int row = i >> 3, slot = i & 7;
    if (row < NT * 16) *reinterpret_cast<uint4*>(lds + lds_off(row, slot)) = reg[it];
  }
}

// Half slabs.  The last K-slice of a layer with an odd number of 64-byte K chunks (K = 32, 96, ...) carries only 4 slots per weight row.
// The wave-specialised kernel stores those slabs with 64-byte rows: a 96 -> 96 layer then keeps 48 output channels resident (83 KiB)
// next to the patch and the stage, i.e. 2 channel blocks instead of 3 (the patch is staged once per channel block).
// Slot s of row r sits at physical slot s ^ ((r >> 3 & 1) << 1): the 16 lanes of a ds_read_b128 service group (rows 0-3 + 12-15 of one
// slot, rows 4-11 of its neighbour) then cover 16 distinct 16-byte bank groups.
__device__ __forceinline__ int lds_off64(int row, int slot) { return row * (DD_LDS_ROW / 2) + ((slot ^ (((row >> 3) & 1) << 1)) << 4); }
// byte offset of slab (tap, slice) from the weight base; `half_last`: the last slice uses half slabs
__device__ __forceinline__ int slab_off(int tap, int slice, int taps, int nslices, int wb, bool half_last) {
  return (half_last && slice == nslices - 1) ? slice * taps * wb + tap * (wb / 2) : (slice * taps + tap) * wb;
}

// Every (tap, slice) weight slab of this workgroup's channel block -> LDS, once per launch.  BATCH 16-byte vectors per thread are in
// flight before the first is stored: staging one slab at a time costs one full memory round trip per slab, and with every
// workgroup of the launch doing it at once that was 34 us of a 63 us launch (64->64 3x3, 18 slabs; tools/phase_profile.py).
template <typename T, int NT, int NTHREADS, int BATCH, bool HALF_LAST = false>
__device__ __forceinline__ void stage_weights(char* wbase, const T* __restrict__ Wp, const ConvP& p, int n0, int nslices, int tid) {
  constexpr int KC = DD_LDS_ROW / (int)sizeof(T), ROWS = NT * 16, WB = ROWS * DD_LDS_ROW;
  const bool half_last = HALF_LAST && (p.kchunks & 1);
  const T* zero = reinterpret_cast<const T*>(&dd_zero16_v);
  const int total = p.taps * nslices * ROWS * 8;
  for (int base = 0; base < total; base += NTHREADS * BATCH) {
    uint4 reg[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int v = base + b * NTHREADS + tid;
      const int slot = v & 7, r = v >> 3;
      const int row = r % ROWS, sidx = r / ROWS;
      const int slice = sidx / p.taps, tap = sidx - slice * p.taps;
      const int nslots = min(2, p.kchunks - 2 * slice) * 4;
      const int ng = n0 + row;
      const bool ok = v < total && slot < nslots && ng < p.n_pad;
      reg[b] = *reinterpret_cast<const uint4*>(ok ? Wp + ((long)tap * p.n_pad + ng) * p.k_pad + slice * KC + slot * Elem<T>::PER16 : zero);
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int v = base + b * NTHREADS + tid;
      const int slot = v & 7, r = v >> 3;
      const int row = r % ROWS, sidx = r / ROWS;
      const int slice = sidx / p.taps, tap = sidx - slice * p.taps;
      const bool half = half_last && slice == nslices - 1;
      char* slab = wbase + slab_off(tap, slice, p.taps, nslices, WB, half_last);
      if (v < total && !(half && slot >= 4)) *reinterpret_cast<uint4*>(slab + (half ? lds_off64(row, slot) : lds_off(row, slot))) = reg[b];
    }
  }
}

// XCD-aware work assignment.  Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8) and every XCD
// has its own 4 MiB L2.  XCD x therefore owns a CONTIGUOUS range of tiles (whole images when B % 8 == 0) and all channel blocks
// of a tile run on the same XCD: halo rows shared by neighbouring tiles and the patch re-read by the other channel blocks hit
// in that XCD's L2 instead of going to the memory side.  (Fallback when the grid is not a multiple of 8 * nblk: plain striding.)
struct TileWalk { int nb, first, end, step; };
__device__ __forceinline__ TileWalk tile_walk(const ConvP& p) {
  const int b = blockIdx.x, G = gridDim.x;
  TileWalk w;
  if (G % (8 * p.nblk) == 0) {
    const int xcd = b & 7, i = b >> 3;
    w.nb = i % p.nblk;
    w.step = G / (8 * p.nblk);
    w.first = (int)((long)p.total_tiles * xcd / 8) + i / p.nblk;
    w.end = (int)((long)p.total_tiles * (xcd + 1) / 8);
  } else {
    w.nb = b % p.nblk; w.first = b / p.nblk; w.step = G / p.nblk; w.end = p.total_tiles;
  }
  return w;
}

#ifndef DD_SCHED
#define DD_SCHED 2
#endif
#ifndef DD_PRIO_MFMA
#define DD_PRIO_MFMA 1   // s_setprio of the wave-specialised kernel's matrix waves ...
#endif
#ifndef DD_PRIO_IO
#define DD_PRIO_IO 2     // ... and of its I/O waves
#endif

// MFMA work of `NTAPS` consecutive taps on one staged patch.  Fragments are double-buffered in registers: the 4+NT fragment reads
// of step s+1 are spread between the 4*NT MFMAs of step s, each read >= 11 MFMAs (~180 cycles) ahead of its first use, in the
// order the MFMAs need them (w0, p0..p3, w1..).  The interleave is pinned with sched_barrier(0): left alone -- and also with
// sched_group_barrier, which fixes only HOW MANY reads go between MFMAs, not WHICH -- hipcc issues each weight fragment 2 MFMAs
// before its use, and with one MFMA wave per SIMD every such read stalls the matrix pipe (measured 45 vs 16 cycles per MFMA).
template <typename T, int NT, int PH, int NTAPS, int NCH, bool WHALF = false>
__device__ __forceinline__ void mma_phase(f32x4_t (&acc)[NT][4], const char* patch, int shift_y, int shift_x, const char* wslab0, int wslab_stride,
                                          int tap_first, int wave, int q, int li) {
  constexpr int STEPS = NTAPS * NCH, NLOAD = 4 + NT, NMMA = 4 * NT;
  uint4 bf[2][4], af[2][NT];
  // load item i of step s: i = 0 -> weight fragment 0, 1..4 -> pixel fragments 0..3, 5.. -> weight fragments 1..
  auto load_item = [&](int s, int i) {
    const int ti = s / NCH, c = s - ti * NCH;
    const int tap = tap_first + ti;
    const int dy = NTAPS == 9 ? tap / 3 : 0, dx = NTAPS == 9 ? tap - dy * 3 : 0;
    const int slot = c * 4 + q;
    if (i >= 1 && i <= 4) {
      const int r = i - 1;
      bf[s & 1][r] = *reinterpret_cast<const uint4*>(patch + lds_pix_off(wave * 4 + r + dy + shift_y, li + dx + shift_x, PH, slot));
    } else {
      const int j = i == 0 ? 0 : i - 4;
      af[s & 1][j] = *reinterpret_cast<const uint4*>(wslab0 + ti * wslab_stride + (WHALF ? lds_off64(j * 16 + li, slot) : lds_off(j * 16 + li, slot)));
    }
  };
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) load_item(0, i);
#ifdef DD_EXP_NO_DSREAD   // experiment: MFMA loop without fragment reads (results are garbage)
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) load_item(1, i);
#endif
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
#ifndef DD_EXP_NO_DSREAD
      if (s + 1 < STEPS) load_item(s + 1, i);
#endif
#if DD_SCHED == 2
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int m = (i * NMMA) / NLOAD; m < ((i + 1) * NMMA) / NLOAD; ++m) {
        const int j = m >> 2, r = m & 3;
        acc[j][r] = mma16<T>(af[s & 1][j], bf[s & 1][r], acc[j][r]);
      }
#if DD_SCHED == 2
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
}

template <typename T, int NT, bool HALO, bool RESIDENT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PH = PatchDim<HALO>::PH, PATCH_BYTES = PatchDim<HALO>::NPIX * DD_LDS_ROW;
  constexpr int WB = NT * 16 * DD_LDS_ROW;    // bytes of one weight slab
  constexpr int INNER = HALO ? 9 : 1;         // taps that reuse one staged patch
  constexpr bool BF = sizeof(T) == 2;
  char* patch = smem;
  char* wbase = smem + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TileWalk walk = tile_walk(p);
  const int nb = walk.nb, first = walk.first, stride = walk.step, tiles_end = walk.end;
  const int n0 = nb * NT * 16;
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.wp);
  const int nslices = (p.kchunks + 1) >> 1;
  const int outer = HALO ? nslices : nslices * p.taps;   // units per tile
  const int q = lane >> 4, li = lane & 15;
  const bool in_relu = (p.flags & DD_IN_RELU) != 0;
  if (first >= tiles_end) return;

  if (RESIDENT) stage_weights<T, NT, 256, 8>(wbase, Wp, p, n0, nslices, tid);

  const bool pixshuf = (p.flags & DD_PIXSHUF) != 0;
  const int cout = pixshuf ? p.n / 4 : p.n;
  float biasr[NT][4];   // this lane's bias values (channels n0 + j*16 + q*4 + e): the accumulators START from them
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + j * 16 + q * 4 + e;
      const int bi = pixshuf ? n % cout : n;
      biasr[j][e] = (p.bias && n < p.n && bi < p.nbias) ? p.bias[bi] : 0.f;
    }
  // Coalesced-epilogue plan (bf16): the wave's 64 pixels x 64 channels go through an 8 KiB wave-private LDS region in chunks of
  // 4 channel tiles; read-back vector `it` of this lane is pixel epix[it] (tile-local index), 8 channels starting at ech.
  constexpr int ESLOTS = 8;
  int e_lds[ESLOTS], e_pix[ESLOTS];
  const int e_slot = lane & 7;
  if (BF) {
#pragma unroll
    for (int it = 0; it < ESLOTS; ++it) {
      const int pixl = (lane >> 3) + it * 8;                   // 0..63 within the wave's 4 rows
      e_lds[it] = lds_off(pixl, e_slot);
      const int tyl = wave * 4 + (pixl >> 4), txl = pixl & 15;
      e_pix[it] = pixshuf ? (2 * tyl) * p.wout + 2 * txl : tyl * p.wout + txl;     // pixel offset from the tile origin (ab added per chunk)
    }
  }
  PatchPlan<HALO> plan;
  patch_plan<T, HALO>(plan, p, tid);
  f32x4_t acc[NT][4];
  uint4 pre[PatchDim<HALO>::ITERS];
  uint4 wreg[(NT + 1) / 2];
  int tile = first, o = 0;
  {
    const int nch = min(2, p.kchunks);
    patch_load<T, HALO>(pre, plan, X, p, tile, 0, 0, nch * 4, tid);
    patch_store<T, HALO>(patch, pre, plan, in_relu);
    if (!RESIDENT) slab_load<T, NT>(wreg, Wp, p, n0, 0, 0, nch * 4, tid);
  }
  __syncthreads();

  while (tile < tiles_end) {
    // this unit: outer step o of `tile`
    const int slice = HALO ? o : o / p.taps, tap0 = HALO ? 0 : o % p.taps;
    const int nch = min(2, p.kchunks - 2 * slice);
    int no = o + 1, ntile = tile;
    if (no == outer) { no = 0; ntile = tile + stride; }
    const bool has_next = ntile < tiles_end;
    const int nslice = HALO ? no : no / p.taps, ntap0 = HALO ? 0 : no % p.taps;
    const int nnch = min(2, p.kchunks - 2 * nslice);
    PHASE_T(t0);
    if (has_next) patch_load<T, HALO>(pre, plan, X, p, ntile, nslice, ntap0, nnch * 4, tid);   // in flight during the MFMAs below
    PHASE_T(t1);
    if (o == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = f32x4_t{biasr[j][0], biasr[j][1], biasr[j][2], biasr[j][3]};
    }

    if (RESIDENT) {
      const char* w0 = wbase + (slice * p.taps + tap0) * WB;
      if (nch == 2) mma_phase<T, NT, PH, INNER, 2>(acc, patch, 0, 0, w0, WB, tap0, wave, q, li);
      else mma_phase<T, NT, PH, INNER, 1>(acc, patch, 0, 0, w0, WB, tap0, wave, q, li);
    } else {
#pragma unroll 1
      for (int ti = 0; ti < INNER; ++ti) {
        const int tap = tap0 + ti;
        char* wdst = wbase + (ti & 1) * WB;
        slab_store<NT>(wdst, wreg, tid);
        __syncthreads();
        // prefetch the next slab (next tap of this unit, or the first tap of the next unit) while this tap computes
        if (ti + 1 < INNER) slab_load<T, NT>(wreg, Wp, p, n0, tap + 1, slice, nch * 4, tid);
        else if (has_next) slab_load<T, NT>(wreg, Wp, p, n0, ntap0, nslice, nnch * 4, tid);
        const int dy = HALO ? ti / 3 : 0, dx = HALO ? ti - dy * 3 : 0;
        if (nch == 2) mma_phase<T, NT, PH, 1, 2>(acc, patch, dy, dx, wdst, 0, 0, wave, q, li);
        else mma_phase<T, NT, PH, 1, 1>(acc, patch, dy, dx, wdst, 0, 0, wave, q, li);
      }
    }
    PHASE_T(t2);
    __syncthreads();   // every wave is done reading the patch (and the streamed slabs)

    if (o == outer - 1) {
      // ---------------------------------------------------------------- epilogue of `tile`
      const TileCoord tc = tile_coord(p, tile);
      T* __restrict__ Y = reinterpret_cast<T*>(p.y);
      const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
      const T* __restrict__ M = reinterpret_cast<const T*>(p.mask);
      const bool out_relu = (p.flags & DD_OUT_RELU) != 0, accum = (p.flags & DD_ACCUM) != 0;
      if constexpr (BF) {
        // Transposed, fully coalesced epilogue: every global access is a 16-byte vector, 8 lanes = 128 contiguous bytes.
        char* stage = patch + wave * (64 * DD_LDS_ROW);   // wave-private 8 KiB region of the (consumed) patch buffer
        const bool interior = tc.y0 + DD_TILE <= p.H && tc.x0 + DD_TILE <= p.W;
        const long tile_pix = pixshuf ? ((long)tc.b * p.hout + 2 * tc.y0) * p.wout + 2 * tc.x0 : ((long)tc.b * p.hout + tc.y0) * p.wout + tc.x0;
#pragma unroll
        for (int jb = 0; jb < NT; jb += 4) {
          const int ntc = NT - jb < 4 ? NT - jb : 4;      // channel tiles in this chunk (compile-time after unrolling)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              if (jj >= ntc) continue;
              const int j = jb + jj;
              f32x4_t a = acc[j][r];
              if (out_relu && !R) { a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); a[2] = fmaxf(a[2], 0.f); a[3] = fmaxf(a[3], 0.f); }
              uint2 pk;
              pk.x = pack2<T>(a[0], a[1]);
              pk.y = pack2<T>(a[2], a[3]);
              *reinterpret_cast<uint2*>(stage + lds_off(r * 16 + li, jj * 2 + (q >> 1)) + (q & 1) * 8) = pk;
            }
          const int n = n0 + jb * 16 + e_slot * 8;        // first of this lane's 8 channels
          const bool lane_ok = e_slot < ntc * 2 && n < p.n;
          int ch = lane_ok ? n : 0; long abpix = 0;
          if (pixshuf) { const int ab = ch / cout; ch = ch - ab * cout; abpix = (long)(ab >> 1) * p.wout + (ab & 1); }
#pragma unroll
          for (int half = 0; half < ESLOTS; half += 4) {
            uint4 pv[4], mv[4], rv[4], av[4];
            long pixv[4];
            bool okv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {      // pass 1: every load of these 4 vectors in flight at once, all unconditional
              const int it = half + k;
              const int pixl = (lane >> 3) + it * 8;
              okv[k] = lane_ok && (interior || (tc.y0 + wave * 4 + (pixl >> 4) < p.H && tc.x0 + (pixl & 15) < p.W));
              pixv[k] = okv[k] ? tile_pix + abpix + e_pix[it] : tile_pix;    // the tile origin is always a valid pixel
              pv[k] = *reinterpret_cast<const uint4*>(stage + e_lds[it]);
              if (M) mv[k] = *reinterpret_cast<const uint4*>(M + pixv[k] * p.ldmask + ch);
              if (R) rv[k] = *reinterpret_cast<const uint4*>(R + pixv[k] * p.ldres + ch);
              if (accum) av[k] = *reinterpret_cast<const uint4*>(Y + pixv[k] * p.ldy + ch);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {      // pass 2: combine + predicated store
              uint4 o4 = pv[k];
              if (R) {
                float v[8], t[8];
                unpack8t<T>(o4, v); unpack8t<T>(rv[k], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] += t[e]; if (out_relu) v[e] = fmaxf(v[e], 0.f); }
                o4 = pack8t<T>(v);
              }
              if (M) o4 = mask_bf16x8(o4, mv[k]);
              if (accum) {
                float v[8], t[8];
                unpack8t<T>(o4, v); unpack8t<T>(av[k], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
                o4 = pack8t<T>(v);
              }
              if (okv[k]) *reinterpret_cast<uint4*>(Y + pixv[k] * p.ldy + ch) = o4;
            }
          }
        }
      } else {
        const int ox = tc.x0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oy = tc.y0 + wave * 4 + r;
          if (oy >= p.H || ox >= p.W) continue;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 16 + q * 4;
            if (n >= p.n) continue;
            float v[4] = {acc[j][r][0], acc[j][r][1], acc[j][r][2], acc[j][r][3]};
            long pix; int ch;
            if (pixshuf) {
              const int ab = n / cout; ch = n - ab * cout;
              pix = ((long)tc.b * p.hout + 2 * oy + (ab >> 1)) * p.wout + 2 * ox + (ab & 1);
            } else {
              ch = n;
              pix = ((long)tc.b * p.hout + oy) * p.wout + ox;
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
    }
    PHASE_T(t3);
    if (has_next) {
      if (BF && o == outer - 1) __syncthreads();   // the epilogue's staging reads are done before the patch is overwritten
      patch_store<T, HALO>(patch, pre, plan, in_relu);
    }
    PHASE_T(t4);
    __syncthreads();   // the next unit's patch is complete
    PHASE_T(t5);
    PHASE_ADD(0, t0, t1); PHASE_ADD(1, t1, t2); PHASE_ADD(2, t2, t3); PHASE_ADD(3, t3, t4); PHASE_ADD(4, t4, t5);
    if (blockIdx.x == 0 && tid == 0) { PHASE_ADD(5, 0ull, 1ull); }
    tile = ntile;
    o = no;
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant (bf16, resident weights, NT <= 4): 8 waves per workgroup, 2 per SIMD.
//   waves 0-3 (MFMA role): ds_read + MFMA only; at the end of a tile they park the bf16 result in a 32 KiB LDS stage.
//   waves 4-7 (I/O role) : issue the global loads of the NEXT unit's patch, drain the PREVIOUS tile's stage to global memory
//                          with coalesced 16-byte vectors (+ mask / residual / accumulate), then write the patch to LDS.
// A wave that stalls on memory cannot issue MFMAs, and HBM writes run at only ~3.2 TB/s (tools/ubench/mem_ubench.hip): with the
// roles split, the SIMD's scheduler overlaps the I/O waves' stalls with the MFMA waves' matrix work.  Two barriers per unit.
template <typename T, int NT, bool HALO>
__global__ __launch_bounds__(512) void conv_igemm_ws_kernel(const ConvP p) {
  static_assert(sizeof(T) == 2, "wave-specialised kernel: bf16 / fp16 storage");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PH = PatchDim<HALO>::PH, PATCH_BYTES = PatchDim<HALO>::NPIX * DD_LDS_ROW;
  constexpr int STAGE_BYTES = DD_TILE * DD_TILE * DD_LDS_ROW;
  constexpr int WB = NT * 16 * DD_LDS_ROW;
  constexpr int INNER = HALO ? 9 : 1;
  char* patch = smem;
  char* stage_all = smem + PATCH_BYTES;
  char* wbase = smem + PATCH_BYTES + STAGE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform => the role split is a scalar branch
  const bool io = wave_u >= 4;
  const int w4 = wave_u & 3, t256 = tid & 255;
  const TileWalk walk = tile_walk(p);
  const int nb = walk.nb, first = walk.first, stride = walk.step, tiles_end = walk.end;
  const int n0 = nb * NT * 16;
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.wp);
  const int nslices = (p.kchunks + 1) >> 1;
  const int outer = HALO ? nslices : nslices * p.taps;
  const int q = lane >> 4, li = lane & 15;
  const bool in_relu = (p.flags & DD_IN_RELU) != 0;
  const bool pixshuf = (p.flags & DD_PIXSHUF) != 0;
  const int cout = pixshuf ? p.n / 4 : p.n;
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ M = reinterpret_cast<const T*>(p.mask);
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  const bool out_relu = (p.flags & DD_OUT_RELU) != 0, accum = (p.flags & DD_ACCUM) != 0;
  if (first >= tiles_end) return;
  char* stage = stage_all + w4 * (64 * DD_LDS_ROW);   // MFMA wave w4 writes it, I/O wave w4 drains it
#ifdef DD_PROFILE_PHASES
  const unsigned long long k_t0 = __builtin_readcyclecounter(), k_w0 = wall_clock64();
#endif

  stage_weights<T, NT, 512, 9, true>(wbase, Wp, p, n0, nslices, tid);   // all 8 waves; made visible by the roles' first barrier
  const bool half_last = (p.kchunks & 1) != 0;      // the last K-slice is stored as half slabs
  // The bias lives in LDS too: re-reading it from global memory at every tile start put a full (loaded-machine) memory latency
  // in front of each tile's first MFMA -- half of the launch time.
  float* bias_lds = reinterpret_cast<float*>(wbase + p.taps * nslices * WB - (half_last ? p.taps * (WB / 2) : 0));
  if (tid < NT * 16) {
    const int n = n0 + tid;
    const int bi = pixshuf ? n % cout : n;
    bias_lds[tid] = (p.bias && n < p.n && bi < p.nbias) ? p.bias[bi] : 0.f;
  }

  // Nothing role-specific is computed before the role branch: the two roles then have disjoint live ranges and each fits the
  // 256-VGPR budget of an 8-wave workgroup on its own.
  int tile = first, o = 0;
  if (!io) {
    // ------------------------------------------------------------------ MFMA role
    // Issue priority: measured, the I/O waves (the critical path of the HBM-heavy layers) ABOVE the matrix waves is worth +0.7 % of a
    // training step over the opposite order; the matrix pipe is busy 16 cycles per MFMA and loses nothing by yielding an issue slot.
    __builtin_amdgcn_s_setprio(DD_PRIO_MFMA);
    __syncthreads();                 // weights + first patch staged by the I/O waves
    f32x4_t acc[NT][4];
    while (tile < tiles_end) {
      const int slice = HALO ? o : o / p.taps, tap0 = HALO ? 0 : o % p.taps;
      const int nch = min(2, p.kchunks - 2 * slice);
      int no = o + 1, ntile = tile;
      if (no == outer) { no = 0; ntile = tile + stride; }
      if (o == 0) {   // the accumulators start from the bias
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias_lds + j * 16 + q * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][r] = bv;
        }
      }
      const char* w0 = wbase + slab_off(tap0, slice, p.taps, nslices, WB, half_last);
      PHASE_T(m0);
      if (nch == 2) mma_phase<T, NT, PH, INNER, 2>(acc, patch, 0, 0, w0, WB, tap0, w4, q, li);
      else mma_phase<T, NT, PH, INNER, 1, true>(acc, patch, 0, 0, w0, WB / 2, tap0, w4, q, li);      // nch == 1 <=> the half-slab slice
      PHASE_T(m1);
      __syncthreads();   // bar1: patch consumed by every MFMA wave; stage drained by the I/O waves
      PHASE_T(m2);
      if (o == outer - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            f32x4_t a = acc[j][r];
            if (out_relu && !R) { a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); a[2] = fmaxf(a[2], 0.f); a[3] = fmaxf(a[3], 0.f); }
            uint2 pk;
            pk.x = pack2<T>(a[0], a[1]);
            pk.y = pack2<T>(a[2], a[3]);
            *reinterpret_cast<uint2*>(stage + lds_off(r * 16 + li, j * 2 + (q >> 1)) + (q & 1) * 8) = pk;
          }
      }
      PHASE_T(m3);
      __syncthreads();   // bar2: stage full; next patch complete
      PHASE_T(m4);
      PHASE_ADD_T(0, m0, m1, 0); PHASE_ADD_T(1, m1, m2, 0); PHASE_ADD_T(2, m2, m3, 0); PHASE_ADD_T(3, m3, m4, 0); PHASE_ADD_T(5, 0ull, 1ull, 0);
      tile = ntile;
      o = no;
    }
#ifdef DD_PROFILE_PHASES
    if (blockIdx.x == 0 && tid == 0) { dd_phase_cycles[14] += __builtin_readcyclecounter() - k_t0; dd_phase_cycles[15] += wall_clock64() - k_w0; }
#endif
  } else {
    // ------------------------------------------------------------------ I/O role
    __builtin_amdgcn_s_setprio(DD_PRIO_IO);
    PatchPlan<HALO> plan;
    uint4 pre[PatchDim<HALO>::ITERS];
    {
      patch_plan<T, HALO>(plan, p, t256);
      patch_load<T, HALO>(pre, plan, X, p, first, 0, 0, min(2, p.kchunks) * 4, t256);
      patch_store<T, HALO>(patch, pre, plan, in_relu);
    }
    __syncthreads();
    // Drain mapping: 2^sh 16-byte slots per pixel (8 = the full 64-channel row; narrow layers use fewer vectors per pixel, so a
    // 24-channel output costs 4 store instructions per wave instead of 8 -- store issue time does not shrink with idle lanes).
    constexpr int ESLOTS = 8;
    const int nsl_out = ((pixshuf ? NT * 16 : min(NT * 16, p.n - n0)) + 7) >> 3;
    const int sh = pixshuf ? 3 : nsl_out > 4 ? 3 : nsl_out > 2 ? 2 : nsl_out > 1 ? 1 : 0;
    const int e_n = 1 << sh;                 // store instructions per wave and tile
    int e_lds[ESLOTS], e_pix[ESLOTS];
    const int e_slot = lane & (e_n - 1);
#pragma unroll
    for (int it = 0; it < ESLOTS; ++it) {
      const int pixl = ((lane >> sh) + it * (64 >> sh)) & 63;
      e_lds[it] = lds_off(pixl, e_slot);
      const int tyl = w4 * 4 + (pixl >> 4), txl = pixl & 15;
      e_pix[it] = pixshuf ? (2 * tyl) * p.wout + 2 * txl : tyl * p.wout + txl;
    }
    const int n = n0 + e_slot * 8;
    const bool lane_ok = e_slot < NT * 2 && n < p.n;
    int ch = lane_ok ? n : 0; long abpix = 0;
    if (pixshuf) { const int ab = ch / cout; ch = ch - ab * cout; abpix = (long)(ab >> 1) * p.wout + (ab & 1); }

    auto drain = [&](int dtile) {
      const TileCoord tc = tile_coord(p, dtile);
      const bool interior = tc.y0 + DD_TILE <= p.H && tc.x0 + DD_TILE <= p.W;
      const long tile_pix = pixshuf ? ((long)tc.b * p.hout + 2 * tc.y0) * p.wout + 2 * tc.x0 : ((long)tc.b * p.hout + tc.y0) * p.wout + tc.x0;
      if (M && !R && !accum && !pixshuf && e_n == ESLOTS) {
        // The plain data-gradient drain (mask only).  The stores depend on the mask loads, and under load one memory round trip costs
        // thousands of cycles: all 8 mask vectors of the tile are requested before the first is used -- one round trip per tile
        // instead of one per 4 vectors (64->64 at 128^2: 223 -> 195 us per launch).  Only this case: with residual / accumulate
        // operands 8 more vectors in flight spill, and so does keeping the 8 masks in flight ACROSS the next patch's loads (tried:
        // 78 spilled VGPRs, every launch 30-50 % slower).
        uint4 mv[ESLOTS];
        int offv[ESLOTS];           // pixel offset from the tile origin, -1: outside the image
#pragma unroll
        for (int it = 0; it < ESLOTS; ++it) {
          const int pixl = ((lane >> 3) + it * 8) & 63;
          const bool ok = lane_ok && (interior || (tc.y0 + w4 * 4 + (pixl >> 4) < p.H && tc.x0 + (pixl & 15) < p.W));
          offv[it] = ok ? e_pix[it] : -1;
          mv[it] = *reinterpret_cast<const uint4*>(M + (tile_pix + max(offv[it], 0)) * p.ldmask + ch);
        }
#pragma unroll
        for (int it = 0; it < ESLOTS; ++it) {
          const uint4 o4 = mask_bf16x8(*reinterpret_cast<const uint4*>(stage + e_lds[it]), mv[it]);
          if (offv[it] >= 0) *reinterpret_cast<uint4*>(Y + (tile_pix + offv[it]) * p.ldy + ch) = o4;
        }
        return;
      }
#pragma unroll
      for (int half = 0; half < ESLOTS; half += 4) {
        if (half >= e_n) break;                 // wave-uniform
        uint4 pv[4], mv[4], rv[4], av[4];
        long pixv[4];
        bool okv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = half + k;
          if (it >= e_n) continue;              // wave-uniform (1 or 2 vectors per wave and tile)
          const int pixl = ((lane >> sh) + it * (64 >> sh)) & 63;
          okv[k] = lane_ok && (interior || (tc.y0 + w4 * 4 + (pixl >> 4) < p.H && tc.x0 + (pixl & 15) < p.W));
          pixv[k] = okv[k] ? tile_pix + abpix + e_pix[it] : tile_pix;
          pv[k] = *reinterpret_cast<const uint4*>(stage + e_lds[it]);
          if (M) mv[k] = *reinterpret_cast<const uint4*>(M + pixv[k] * p.ldmask + ch);
          if (R) rv[k] = *reinterpret_cast<const uint4*>(R + pixv[k] * p.ldres + ch);
          if (accum) av[k] = *reinterpret_cast<const uint4*>(Y + pixv[k] * p.ldy + ch);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (half + k >= e_n) continue;
          uint4 o4 = pv[k];
          if (R) {
            float v[8], t[8];
            unpack8t<T>(o4, v); unpack8t<T>(rv[k], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] += t[e]; if (out_relu) v[e] = fmaxf(v[e], 0.f); }
            o4 = pack8t<T>(v);
          }
          if (M) o4 = mask_bf16x8(o4, mv[k]);
          if (accum) {
            float v[8], t[8];
            unpack8t<T>(o4, v); unpack8t<T>(av[k], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
            o4 = pack8t<T>(v);
          }
          if (okv[k]) *reinterpret_cast<uint4*>(Y + pixv[k] * p.ldy + ch) = o4;
        }
      }
    };

    int pending = -1;
    while (tile < tiles_end) {
      int no = o + 1, ntile = tile;
      if (no == outer) { no = 0; ntile = tile + stride; }
      const bool has_next = ntile < tiles_end;
      const int nslice = HALO ? no : no / p.taps, ntap0 = HALO ? 0 : no % p.taps;
      const int nnch = min(2, p.kchunks - 2 * nslice);
      PHASE_T(i0);
#ifndef DD_EXP_NO_IO
      // Which goes first, the previous tile's drain or the next patch's loads?  Loads return in order.  The mask-only drain asks for
      // all its masks at once, so behind the patch loads it waits ONE round trip that the two share (64->64 dgrad: 195 -> 182 us).
      // The other masked drains make two dependent round trips of their own (4 vectors each): they go first.
      const bool mask_only = M && !R && !accum && !pixshuf && e_n == ESLOTS;
      const bool drain_first = M != nullptr && pending >= 0 && !mask_only;
      if (drain_first) { drain(pending); pending = -1; }
      if (has_next) patch_load<T, HALO>(pre, plan, X, p, ntile, nslice, ntap0, nnch * 4, t256);
#endif
      PHASE_T(i1);
#ifndef DD_EXP_NO_IO
      if (pending >= 0) { drain(pending); pending = -1; }
#endif
      PHASE_T(i2);
      __syncthreads();   // bar1
      PHASE_T(i3);
#ifndef DD_EXP_NO_IO
      if (has_next) patch_store<T, HALO>(patch, pre, plan, in_relu);
#endif
      PHASE_T(i4);
      __syncthreads();   // bar2
      PHASE_T(i5);
      PHASE_ADD_T(8, i0, i1, 256); PHASE_ADD_T(9, i1, i2, 256); PHASE_ADD_T(10, i2, i3, 256); PHASE_ADD_T(11, i3, i4, 256); PHASE_ADD_T(12, i4, i5, 256);
      if (o == outer - 1) pending = tile;
      tile = ntile;
      o = no;
    }
    if (pending >= 0) drain(pending);
  }
}


// Persistent grid: at most `slots` workgroups, a multiple of the channel-block count and -- when that idles under 7 % of the slots and
// every XCD still gets work -- of 8 * nblk, which turns on the XCD-aware tile assignment (tile_walk).
static long grid_size(long slots, const ConvP& p) {
  static int xcd_mode = -1;
  if (xcd_mode < 0) { const char* e = getenv("DD_CONV_XCD"); xcd_mode = e ? atoi(e) : 1; }
  const long need = (long)p.total_tiles * p.nblk;
  long wgs = slots / p.nblk * p.nblk;
  if (wgs < p.nblk) wgs = p.nblk;
  if (wgs > need) wgs = need;
  const long unit = 8L * p.nblk;
  const long xw = wgs / unit * unit;
  if (xcd_mode && xw > 0 && xw * 100 >= wgs * 93 && p.total_tiles >= 8 * (xw / unit)) return xw;
  if (!xcd_mode && wgs % unit == 0 && wgs > p.nblk) wgs -= p.nblk;     // DD_CONV_XCD=0: force the plain striding for A/B runs
  return wgs;
}

template <typename T, int NT, bool HALO, bool RESIDENT>
int launch(const ConvP& p, int nslabs, hipStream_t stream) {
  const size_t patch = (size_t)PatchDim<HALO>::NPIX * DD_LDS_ROW;
  const size_t wb = (size_t)NT * 16 * DD_LDS_ROW;
  const size_t lds = patch + (RESIDENT ? (size_t)nslabs * wb : 2 * wb);
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_igemm_kernel<T, NT, HALO, RESIDENT>));
  const int per_cu = (int)((160 * 1024) / lds) >= 2 ? 2 : 1;   // two resident workgroups overlap each other's memory phases
  long wgs = grid_size((long)dd_device_cus() * per_cu, p);
  hipLaunchKernelGGL((conv_igemm_kernel<T, NT, HALO, RESIDENT>), dim3((unsigned)wgs), dim3(256), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

// bytes of resident weights in the wave-specialised kernel: full 128-byte-row slabs, half slabs for an odd last K-slice
static size_t ws_weight_bytes(const ConvP& p, int nt) {
  const int nslices = (p.kchunks + 1) / 2;
  const size_t wb = (size_t)nt * 16 * DD_LDS_ROW;
  return (size_t)p.taps * nslices * wb - ((p.kchunks & 1) ? (size_t)p.taps * (wb / 2) : 0);
}

template <typename T, int NT, bool HALO>
int launch_ws(const ConvP& p, int nslabs, hipStream_t stream) {
  const size_t lds = (size_t)PatchDim<HALO>::NPIX * DD_LDS_ROW + (size_t)DD_TILE * DD_TILE * DD_LDS_ROW + ws_weight_bytes(p, NT) + NT * 16 * sizeof(float);
  dd_allow_max_lds(reinterpret_cast<const void*>(conv_igemm_ws_kernel<T, NT, HALO>));
  long wgs = grid_size((long)dd_device_cus(), p);
  hipLaunchKernelGGL((conv_igemm_ws_kernel<T, NT, HALO>), dim3((unsigned)wgs), dim3(512), lds, stream, p);
  DD_LAUNCH_CHECK();
  return DD_OK;
}

static bool ws_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DD_CONV_WS"); v = e ? atoi(e) : 1; }
  return v != 0;
}

#define LDS_BUDGET ((size_t)158 * 1024)      // LDS a workgroup may use when deciding weight residency (one workgroup per CU)

template <typename T, int NT>
int launch_nt(const ConvP& p, hipStream_t stream) {
  const bool halo = p.taps == 9;
  const int nslices = (p.kchunks + 1) / 2;
  const int nslabs = p.taps * nslices;
  const size_t patch = (size_t)(halo ? (DD_TILE + 2) * (DD_TILE + 2) : DD_TILE * DD_TILE) * DD_LDS_ROW;
  const bool resident = patch + (size_t)nslabs * NT * 16 * DD_LDS_ROW <= LDS_BUDGET;
  if constexpr (sizeof(T) == 2 && NT <= 4) {
    const size_t ws_lds = patch + (size_t)DD_TILE * DD_TILE * DD_LDS_ROW + ws_weight_bytes(p, NT) + NT * 16 * sizeof(float);
    if (ws_enabled() && ws_lds <= 160 * 1024) return halo ? launch_ws<T, NT, true>(p, nslabs, stream) : launch_ws<T, NT, false>(p, nslabs, stream);
  }
  if (halo) return resident ? launch<T, NT, true, true>(p, nslabs, stream) : launch<T, NT, true, false>(p, nslabs, stream);
  return resident ? launch<T, NT, false, true>(p, nslabs, stream) : launch<T, NT, false, false>(p, nslabs, stream);
}

// Channel-block width policy: a block narrow enough for its weights to stay resident in LDS, but never narrower than two 16-channel tiles.
static int conv_policy_min_resident_nt() { return 2; }

template <typename T>
int dispatch(ConvP& p, hipStream_t stream) {
  const int tiles_n = p.n_pad / 16;
  // Channel blocks of at most 64 (NT <= 4).  Rounds 1-2 also instantiated NT = 6 and 8; those needed more than the 512-register budget and
  // spilled 74 ... 159 registers, and measured no faster than two NT = 4 passes (f32 cfg-2 26.04 vs 26.05 ms/step, bf16 heavy Tiramisu 24.15 vs
  // 24.03 ms/step, round 3), so they are gone.
  // The last block may be ragged (weight rows past n_pad read as zero, channels past n are not stored): a width is a candidate when the
  // padding it adds stays under a quarter of the MFMA work (under 40 % for 1x1 layers, which are bound by re-reading the input once per
  // block: 80 channels = 5 tiles run as 2 blocks of 3 instead of 5 blocks of 1, 176 = 11 tiles as 3 blocks of 4 instead of 11 of 1).
  static const int allowed[] = {4, 3, 2, 1};
  int cands[4], nc = 0;
  for (int c : allowed) {
    const int nb = (tiles_n + c - 1) / c;
    if (c == 1 || (nb * c - tiles_n) * (p.taps == 1 ? 5 : 4) <= (p.taps == 1 ? 2 : 1) * tiles_n) cands[nc++] = c;
  }
  auto blocks_of = [&](int c) { return (tiles_n + c - 1) / c; };
  const bool halo = p.taps == 9;
  const int nslabs = p.taps * ((p.kchunks + 1) / 2);
  const size_t patch = (size_t)(halo ? (DD_TILE + 2) * (DD_TILE + 2) : DD_TILE * DD_TILE) * DD_LDS_ROW;
  int pick = 0;   // widest
  // bf16: the wave-specialised kernel (NT <= 4, resident weights) beats the plain one even when that means more channel blocks: no bf16
  // layer takes a wider block, and a 3x3 layer may go down to NT = 1 to keep its weights resident (192 -> 96: 6 blocks of 16 channels,
  // measured +3.8 % on the whole step against the streamed NT = 6 launch).
  const bool ws_ok = sizeof(T) == 2 && ws_enabled();
  const int min_nt = ws_ok ? 1 : conv_policy_min_resident_nt();
  for (int i = 0; i < nc; ++i)
    if (cands[i] >= min_nt && !(ws_ok && cands[i] > 4) &&
        patch + (ws_ok && cands[i] <= 4 ? (size_t)DD_TILE * DD_TILE * DD_LDS_ROW + ws_weight_bytes(p, cands[i]) + cands[i] * 64
                                         : (size_t)nslabs * cands[i] * 16 * DD_LDS_ROW) <= (ws_ok && cands[i] <= 4 ? (size_t)160 * 1024 : LDS_BUDGET + 2048)) { pick = i; break; }
  // keep every CU busy when the pixel grid is small
  while (pick + 1 < nc && cands[pick] > 2 && (long)p.total_tiles * blocks_of(cands[pick]) < 256) ++pick;
  const int nt = cands[pick];
  p.nblk = blocks_of(nt);
  switch (nt) {
    case 4: return launch_nt<T, 4>(p, stream);
    case 3: return launch_nt<T, 3>(p, stream);
    case 2: return launch_nt<T, 2>(p, stream);
    default: return launch_nt<T, 1>(p, stream);
  }
}

}  // namespace

bool dd_conv_rw_eligible(const dd_conv_args* a);
int dd_conv_rw_launch(const dd_conv_args* a, hipStream_t stream);
bool dd_conv_pw_eligible(const dd_conv_args* a);
int dd_conv_pw_launch(const dd_conv_args* a, hipStream_t stream);

extern "C" int dd_conv_igemm(const dd_conv_args* a, dd_stream stream) {
  DD_REQUIRE(a && a->x && a->wp && a->y, "dd_conv_igemm: null pointer");
  DD_REQUIRE(dd_dtype_ok(a->dtype), "dd_conv_igemm: bad dtype %d", a->dtype);
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
  // 3x3 layers with 65..96 input channels: weights in registers instead of LDS (csrc/dd_conv_rw.hip)
  if (dd_conv_rw_eligible(a)) return dd_conv_rw_launch(a, reinterpret_cast<hipStream_t>(stream));
  // wide 1x1 layers: a 256-pixel x 256-channel GEMM tile over the linear pixel index (csrc/dd_conv_pw.hip)
  if (dd_conv_pw_eligible(a)) return dd_conv_pw_launch(a, reinterpret_cast<hipStream_t>(stream));

  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.res = a->res; p.mask = a->mask; p.y = a->y;
  p.ldx = a->ldx; p.cin = a->cin; p.k_pad = a->k_pad; p.n_pad = a->n_pad; p.ldres = a->ldres; p.ldmask = a->ldmask;
  p.ldy = a->ldy; p.n = a->n; p.B = a->B; p.H = a->H; p.W = a->W; p.taps = a->taps; p.flags = a->flags; p.nbias = a->bias ? a->nbias : 0;
  p.tiles_x = dd_ceil_div(a->W, DD_TILE); p.tiles_y = dd_ceil_div(a->H, DD_TILE); p.nblk = 1;
  p.total_tiles = a->B * p.tiles_x * p.tiles_y;
  p.kchunks = a->k_pad * esz / 64;
  p.hin = gather ? 2 * a->H : a->H; p.win = gather ? 2 * a->W : a->W;
  p.hout = pixshuf ? 2 * a->H : a->H; p.wout = pixshuf ? 2 * a->W : a->W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_DISPATCH_DTYPE(a->dtype, T, return dispatch<T>(p, s));
  return DD_ERR_INVALID;
}
