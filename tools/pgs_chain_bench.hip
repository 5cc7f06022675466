// pgs_chain_bench.hip -- cycles per Gauss-Seidel row visit of ONE wave for two formulations of the visit's dependency chain:
//   A (the kernel's):  y -> v_med3 -> v_sub -> v_readlane -> v_fma -> y        (4 dependent operations)
//   B (step form):     z -> v_max -> v_readlane -> v_fma -> z                  (3 dependent operations)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/pgs_chain_bench tools/pgs_chain_bench.hip && /tmp/pgs_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// lane L of `old` takes v (two instructions off the critical chain, like the kernel's v_readlane + v_writelane pair)
template <int L> __device__ __forceinline__ float writelane_c(float v, float old) { return threadIdx.x == L ? v : old; }

template <int RR> __device__ __forceinline__ void visit_a(float& y, float& lam, float as) {
  if constexpr (RR < 48) {
    const float nl = __builtin_amdgcn_fmed3f(y, 0.0f, 1e30f);
    const float dl = readlane(nl - lam, RR);
    lam = writelane_c<RR>(nl, lam);
    y = fmaf(-as, dl, y);
    visit_a<RR + 1>(y, lam, as);
  }
}
template <int RR> __device__ __forceinline__ void visit_b(float& z, float& nlam, float as) {
  if constexpr (RR < 48) {
    const float d = readlane(fmaxf(z, nlam), RR);
    const float t = nlam - d;
    nlam = writelane_c<RR>(t, nlam);
    z = fmaf(-as, d, z);
    visit_b<RR + 1>(z, nlam, as);
  }
}
__global__ __launch_bounds__(64) void k_a(float* o, long long* cyc, int iters) {
  float y = 0.01f * threadIdx.x - 0.2f, lam = 0.0f, as = 0.001f * (threadIdx.x + 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) visit_a<0>(y, lam, as);
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = y + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_b(float* o, long long* cyc, int iters) {
  float z = 0.01f * threadIdx.x - 0.2f, nlam = 0.0f, as = 0.001f * (threadIdx.x + 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) visit_b<0>(z, nlam, as);
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = z + nlam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 1 << 16);
  long long h[1024];
  for (int waves : {1, 4}) {
    const int blocks = 256 * 4 * waves, iters = 200;
    for (int v = 0; v < 2; ++v) {
      for (int rep = 0; rep < 3; ++rep) { if (v == 0) hipLaunchKernelGGL(k_a, dim3(blocks), dim3(64), 0, 0, d, c, iters); else hipLaunchKernelGGL(k_b, dim3(blocks), dim3(64), 0, 0, d, c, iters); }
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, c, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
      printf("{\"waves_per_simd\": %d, \"form\": \"%s\", \"cycles_per_visit\": %.1f}\n", waves, v == 0 ? "A med3-sub-readlane-fma" : "B max-readlane-fma", s / 1024 / (48.0 * iters));
    }
  }
  return 0;
}
