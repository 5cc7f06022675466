// exec_mask_bench.hip -- what a wave64 VALU instruction costs as a function of how many lanes EXEC enables.  The step kernel runs with
// 11 of 64 lanes active per VALU instruction: does gfx950 skip the 16-lane quarters whose EXEC bits are zero (then packing work into low
// lanes pays), or is an instruction with FEW active lanes even dearer?  32 independent v_fma chains per lane; EXEC = the low K lanes (or, in
// mode 1, alternating blocks of 16 instructions with all lanes and with K lanes); W one-wave workgroups per SIMD on every SIMD.  Reports
// shader cycles (s_memtime) per instruction per wave, the shader clock (s_memtime / s_memrealtime), and wall time per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_mask_bench tools/exec_mask_bench.hip && /tmp/exec_mask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS>
__global__ __launch_bounds__(64) void k(float* o, long long* cyc, int iters, int K, int mode) {
  __shared__ float pad[LDS_FLOATS];   // sets the residency: 160 KB / CU
  const int lane = threadIdx.x;
  pad[lane] = 0.0f;
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = 1.0f + 0.001f * (float)(i + lane);
  const float m = 1.0001f, c = 0.0001f;
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (mode == 0) {
    if (lane < K) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (lane < K) {
#pragma unroll
        for (int i = 16; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = pad[lane];
#pragma unroll
  for (int i = 0; i < 32; ++i) s += a[i];
  o[blockIdx.x * 64 + lane] = s;
  if (lane == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}
static void run(float* d, long long* c, int waves, int K, int mode) {
  static long long h[16384];
  const int blocks = 256 * 4 * waves, iters = 1500;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&]() {
    switch (waves) {   // LDS per workgroup sets how many one-wave workgroups a CU holds
      case 1: hipLaunchKernelGGL(k<9000>, dim3(blocks), dim3(64), 0, 0, d, c, iters, K, mode); break;   // 36 KB: 4 per CU
      case 2: hipLaunchKernelGGL(k<4800>, dim3(blocks), dim3(64), 0, 0, d, c, iters, K, mode); break;   // 19 KB: 8 per CU
      case 3: hipLaunchKernelGGL(k<3300>, dim3(blocks), dim3(64), 0, 0, d, c, iters, K, mode); break;   // 13 KB: 12 per CU
      default: hipLaunchKernelGGL(k<2560>, dim3(blocks), dim3(64), 0, 0, d, c, iters, K, mode); break;  // 10 KB: 16 per CU
    }
  };
  launch(); launch();
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h, c, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double s = 0, r = 0; for (int i = 0; i < blocks; ++i) { s += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
  printf("{\"waves_per_simd\": %d, \"mode\": \"%s\", \"active_lanes\": %d, \"cycles_per_instruction_per_wave\": %.2f, \"shader_clock_GHz\": %.2f, "
         "\"ns_per_instruction_per_simd\": %.3f}\n", waves, mode ? "16 full + 16 masked" : "all masked", K,
         s / blocks / iters / 32.0, 0.1 * s / r, 1e6 * ms / ((double)waves * iters * 32.0));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 1 << 17);
  for (int waves : {1, 2, 3, 4})
    for (int K : {64, 32, 24, 16, 12, 8, 4, 1}) run(d, c, waves, K, 0);
  for (int K : {64, 16, 8, 1}) run(d, c, 4, K, 1);
  return 0;
}
