// lds_bench.hip -- what an LDS read costs the CU's LDS pipe as a function of width, active lanes and address pattern, with 16 one-wave
// workgroups per CU (the step kernel's residency).  Decides whether idle lanes should be masked off around LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bench tools/lds_bench.hip && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64, 4) void k(float* o, long long* cyc, int iters) {
  __shared__ float L[2560];  // 10 KB: 16 workgroups per CU
  const int lane = threadIdx.x;
  for (int i = lane; i < 2560; i += 64) L[i] = (float)i;
  __syncthreads();
  asm volatile("" ::: "v127");
  f4 acc = {0, 0, 0, 0};
  float a1 = 0;
  // per-lane distinct rows of 28 floats (the Delassus pattern), or one broadcast address
  const int row = (MODE & 1) ? 0 : lane % 48;
  const float* p = L + 28 * row;
  const bool on = (MODE & 2) ? lane < 16 : true;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (on) {
      if (MODE & 4) {  // dword reads
#pragma unroll
        for (int u = 0; u < 8; ++u) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((int)((size_t)p & 0xFFFF) + 0), "n"(4 * 0)); a1 += v; }
      } else {
#pragma unroll
        for (int u = 0; u < 7; ++u) acc += *reinterpret_cast<const f4*>(p + 4 * u);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    p += (it & 1) ? -28 : 28;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + lane] = acc.x + acc.y + acc.z + acc.w + a1;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 1 << 16);
  static long long h[4096];
  const char* names[] = {"7 x ds_read_b128, 64 lanes, per-lane rows", "7 x ds_read_b128, 64 lanes, broadcast", "7 x ds_read_b128, 16 lanes (exec), per-lane rows",
                         "7 x ds_read_b128, 16 lanes (exec), broadcast"};
  for (int waves : {1, 4}) {
    const int blocks = 256 * 4 * waves, iters = 2000;
    for (int m = 0; m < 4; ++m) {
      for (int rep = 0; rep < 3; ++rep) {
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, c, iters); break;
        }
      }
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, c, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
      printf("{\"waves_per_simd\": %d, \"pattern\": \"%s\", \"ticks_per_group_of_7\": %.1f}\n", waves, names[m], s / blocks / iters);
    }
  }
  return 0;
}
