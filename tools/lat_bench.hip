// lat_bench.hip -- dependent-issue latencies of ONE wave on a SIMD of its own (and with 4 waves per SIMD): what a latency-bound
// wave of the step kernel pays per link of its dependency chains.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_bench tools/lat_bench.hip && /tmp/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
template <int K> __device__ __forceinline__ float chain_fma(float x, float a) {
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(a));
  return x;
}
template <int K> __device__ __forceinline__ float chain_fma2(float x, float& y, float a) {  // two independent chains interleaved
#pragma unroll
  for (int i = 0; i < K; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(a)); }
  return x;
}
template <int K> __device__ __forceinline__ float chain_fma4(float x, float& y, float& z, float& w, float a) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y) : "v"(a));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(z) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w) : "v"(a));
  }
  return x;
}
template <int K> __device__ __forceinline__ float chain_readlane(float x, float a) {  // v_readlane -> SGPR -> v_fma
#pragma unroll
  for (int i = 0; i < K; ++i) {
    int s;
    asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(x));
    asm volatile("s_nop 0\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(x) : "s"(s), "v"(a));
  }
  return x;
}
template <int K> __device__ __forceinline__ float chain_lds(float x, float* L) {  // ds_write -> ds_read of another lane's value -> add
#pragma unroll
  for (int i = 0; i < K; ++i) {
    L[threadIdx.x] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    x += L[(threadIdx.x + 1) & 63];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  return x;
}
template <int K> __device__ __forceinline__ float chain_ldsread(float x, const float* L) {  // address-dependent ds_read chain (pointer chase)
  int idx = (int)x & 63;
#pragma unroll
  for (int i = 0; i < K; ++i) idx = __float_as_int(L[idx]) & 63;
  return (float)idx;
}
template <int K> __device__ __forceinline__ float chain_dpp(float x) {
#pragma unroll
  for (int i = 0; i < K; ++i) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
  return x;
}
template <int K> __device__ __forceinline__ float chain_bperm(float x) {
#pragma unroll
  for (int i = 0; i < K; ++i) x += __shfl(x, (threadIdx.x + 5) & 63, 64);
  return x;
}
template <int K> __device__ __forceinline__ float chain_rcp(float x) {
#pragma unroll
  for (int i = 0; i < K; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
  return x;
}
template <int K> __device__ __forceinline__ float chain_cnd(float x, float a) {
#pragma unroll
  for (int i = 0; i < K; ++i) { asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc"); }
  return x;
}
__global__ __launch_bounds__(64) void k(float* o, long long* cyc, int which) {
  __shared__ float L[64];
  L[threadIdx.x] = __int_as_float((threadIdx.x * 7 + 3) & 63);
  float x = 0.01f * threadIdx.x + 1.0f, a = 0.999f, y = x + 1, z = x + 2, w = x + 3;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  switch (which) {
    case 0: x = chain_fma<N>(x, a); break;
    case 1: x = chain_fma2<N>(x, y, a); break;
    case 2: x = chain_fma4<N>(x, y, z, w, a); break;
    case 3: x = chain_readlane<N>(x, a); break;
    case 4: x = chain_lds<N / 4>(x, L); break;
    case 5: x = chain_ldsread<N>(x, L); break;
    case 6: x = chain_dpp<N>(x); break;
    case 7: x = chain_bperm<N / 4>(x); break;
    case 8: x = chain_rcp<N>(x); break;
    case 9: x = chain_cnd<N>(x, a); break;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = x + y + z + w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* d; long long* c; (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 1 << 16);
  static long long h[4096];
  const char* names[] = {"v_fma dependent chain (per op)", "2 interleaved fma chains (per op)", "4 interleaved fma chains (per op)",
                         "v_readlane -> s_nop -> v_fma(sgpr) (per pair)", "ds_write + fence + ds_read other lane + add + fence (per round trip)",
                         "ds_read pointer chase (per read)", "v_mov_dpp + v_add (per pair)", "ds_bpermute + add (per pair)", "v_rcp_f32 chain (per op)",
                         "v_cmp + v_cndmask chain (per pair)"};
  const int per[] = {N, 2 * N, 4 * N, N, N / 4, N, N, N / 4, N, N};
  for (int waves : {1, 4}) {
    const int blocks = 256 * 4 * waves;
    for (int v = 0; v < 10; ++v) {
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, c, v);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, c, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
      printf("{\"waves_per_simd\": %d, \"chain\": \"%s\", \"ticks_per_link\": %.2f}\n", waves, names[v], s / blocks / per[v]);
    }
  }
  return 0;
}
