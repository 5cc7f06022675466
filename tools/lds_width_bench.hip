// lds_width_bench.hip -- LDS pipe cost per instruction by access width, direction and address pattern, at the step kernel's residency
// (16 one-wave workgroups per CU, 10 KB of LDS each): cycles of one wave per instruction x 1/16 = pipe cycles per instruction when the
// pipe is the bottleneck.   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_width_bench tools/lds_width_bench.hip && /tmp/lds_width_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W, int WRITE, int STRIDE>   // W = dwords per lane (1, 2, 4); STRIDE = lane stride in dwords (0 = broadcast)
__global__ __launch_bounds__(64, 4) void k(float* o, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float L[2560];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2560; i += 64) L[i] = (float)i;
  __syncthreads();
  const int off = 4 * ((STRIDE * lane) % 2048);   // bytes
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (WRITE) {
        if (W == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(off), "v"(a0), "n"(16 * 0) : "memory");
        if (W == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(off), "v"((double)1.0), "n"(16 * 0) : "memory");
        if (W == 4) { typedef float f4 __attribute__((ext_vector_type(4))); f4 v = {a0, a1, a2, a3}; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(off), "v"(v), "n"(16 * 0) : "memory"); }
      } else {
        if (W == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(off), "n"(16 * 0)); a0 += v; }
        if (W == 2) { double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(off), "n"(16 * 0)); a1 += (float)v; }
        if (W == 4) { typedef float f4 __attribute__((ext_vector_type(4))); f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(off), "n"(16 * 0)); a2 += v.x; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int W, int WRITE, int STRIDE>
void run(const char* name, float* d, long long* c) {
  static long long h[4096];
  const int blocks = 4096, iters = 500;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<W, WRITE, STRIDE>), dim3(blocks), dim3(64), 0, 0, d, c, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, c, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  printf("{\"access\": \"%s\", \"wave_ticks_per_inst\": %.1f, \"pipe_ticks_per_inst_at_16_waves_per_cu\": %.2f}\n", name, s / blocks / iters / 8, s / blocks / iters / 8 / 16);
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 1 << 16);
  run<1, 0, 1>("read b32, consecutive lanes", d, c);
  run<2, 0, 2>("read b64, consecutive lanes", d, c);
  run<4, 0, 4>("read b128, consecutive lanes", d, c);
  run<1, 0, 0>("read b32, broadcast", d, c);
  run<4, 0, 0>("read b128, broadcast", d, c);
  run<2, 0, 6>("read b64, 24-byte lane stride (inertia rows)", d, c);
  run<4, 0, 8>("read b128, 32-byte lane stride", d, c);
  run<4, 0, 20>("read b128, 80-byte lane stride (body records)", d, c);
  run<4, 0, 28>("read b128, 112-byte lane stride (J rows)", d, c);
  run<1, 0, 48>("read b32, 192-byte lane stride (A row-wise)", d, c);
  run<4, 0, 48>("read b128, 192-byte lane stride (A row-wise)", d, c);
  run<1, 1, 1>("write b32, consecutive lanes", d, c);
  run<2, 1, 2>("write b64, consecutive lanes", d, c);
  run<4, 1, 4>("write b128, consecutive lanes", d, c);
  run<1, 1, 28>("write b32, 112-byte lane stride", d, c);
  run<4, 1, 28>("write b128, 112-byte lane stride", d, c);
  run<1, 1, 48>("write b32, 192-byte lane stride", d, c);
  return 0;
}
