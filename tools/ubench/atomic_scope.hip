// How should 256 workgroups add their partial weight gradients (37k fp32 values each) into one result?
//   mode 0: agent-scope atomics, all workgroups into ONE buffer (what the gradient kernels do)
//   mode 1: agent-scope atomics into one buffer PER XCD (8 x less contention per address)
//   mode 2: workgroup-scope atomics into one buffer per XCD (may the XCD's own L2 keep them?)
//   mode 3: as mode 0, but every workgroup starts at its own offset (blockIdx * 1 153 elements): at any moment the workgroups hit DIFFERENT addresses
// hipcc --offload-arch=gfx950 -O3 tools/ubench/atomic_scope.hip -o tools/ubench/atomic_scope && tools/ubench/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }      // HW_REG_XCC_ID[3:0]
template <int MODE>
__global__ __launch_bounds__(512) void flush_kernel(float* buf, long stride, int n, int* xcc_seen) {
  const int x = xcc_id();
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = x;
  float* dst = (MODE == 0 || MODE == 3) ? buf : buf + x * stride;
  const int rot = MODE == 3 ? (int)((blockIdx.x * 1153u) % (unsigned)n) : 0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 512) {
    int i = i0 + rot; if (i >= n) i -= n;
    const float v = 1.0f;
    if (MODE == 2) __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  const int n = 9 * 64 * 64, nwg = 256;
  const long stride = 1 << 16;
  float* buf; int* seen;
  (void)hipMalloc(&buf, 8 * stride * 4); (void)hipMalloc(&seen, nwg * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(buf, 0, 8 * stride * 4);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(flush_kernel<0>, dim3(nwg), dim3(512), 0, 0, buf, stride, n, seen);
      if (mode == 1) hipLaunchKernelGGL(flush_kernel<1>, dim3(nwg), dim3(512), 0, 0, buf, stride, n, seen);
      if (mode == 2) hipLaunchKernelGGL(flush_kernel<2>, dim3(nwg), dim3(512), 0, 0, buf, stride, n, seen);
      if (mode == 3) hipLaunchKernelGGL(flush_kernel<3>, dim3(nwg), dim3(512), 0, 0, buf, stride, n, seen);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    std::vector<float> h(8 * stride); std::vector<int> hs(nwg);
    hipMemcpy(h.data(), buf, 8 * stride * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), seen, nwg * 4, hipMemcpyDeviceToHost);
    double total = 0; for (int k = 0; k < 8; ++k) total += h[k * stride + 5];
    int rr = 0; for (int b = 0; b < nwg; ++b) rr += hs[b] == (b % 8);
    printf("mode %d: %.1f us per launch; element 5 summed over partials = %.0f (expected %d); blockIdx %% 8 == XCC_ID for %d of %d workgroups\n", mode, best * 1e3, total, 6 * nwg, rr, nwg);
  }
  return 0;
}
