// pk_bench.hip -- issue cost of v_fma_f32 vs v_pk_fma_f32 on gfx950, one and four waves per SIMD (tools: why the step kernel's
// dense inner products use packed fp32).  hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_bench tools/pk_bench.hip && /tmp/pk_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, UNROLL = 16;

__global__ __launch_bounds__(64) void k_fma(float* o, float s) {
  float a[UNROLL];
  for (int i = 0; i < UNROLL; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) a[i] = fmaf(a[i], s, 0.5f);
  float t = 0; for (int i = 0; i < UNROLL; ++i) t += a[i];
  o[blockIdx.x * 64 + threadIdx.x] = t;
}
__global__ __launch_bounds__(64) void k_pk(float* o, float s) {
  f2 a[UNROLL];
  for (int i = 0; i < UNROLL; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  const f2 ss = {s, s}, hh = {0.5f, 0.25f};
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) a[i] = __builtin_elementwise_fma(a[i], ss, hh);
  float t = 0; for (int i = 0; i < UNROLL; ++i) t += a[i].x + a[i].y;
  o[blockIdx.x * 64 + threadIdx.x] = t;
}
template <class K> float run(K k, int blocks, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1.0001f);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
  float* d; (void)hipMalloc(&d, 1 << 24);
  for (int waves : {1, 2, 4, 8}) {
    const int blocks = 256 * 4 * waves;
    const float tf = run(k_fma, blocks, d), tp = run(k_pk, blocks, d);
    const double n = (double)ITERS * UNROLL;
    printf("{\"waves_per_simd\": %d, \"us_fma\": %.1f, \"us_pk_fma\": %.1f, \"ns_per_wave_instr_fma\": %.3f, \"ns_per_wave_instr_pk\": %.3f}\n", waves, tf * 1e3, tp * 1e3,
           tf * 1e6 / n, tp * 1e6 / n);
  }
  return 0;
}
