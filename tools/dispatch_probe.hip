// dispatch_probe.hip -- where does the hardware put the 4096 one-wave workgroups of a launch shaped like the step kernel
// (64 threads, 128 VGPRs, 10 KB LDS: 4 waves per SIMD, 16 per CU, every wave resident from the start)?
// Prints, per launch, how many workgroups each SIMD received and whether blockIdx -> (XCD, SE, CU, SIMD) repeats from launch
// to launch -- the facts a load-balancing block order would have to rely on.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_probe tools/dispatch_probe.hip && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
__global__ __launch_bounds__(64, 4) void probe(unsigned* out, int spin) {
  extern __shared__ float L[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("" ::: "v127");  // forces a 128-VGPR allocation: 4 waves per SIMD like the step kernel
  L[threadIdx.x] = (float)hw;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  const int n = 4096;
  unsigned* d; (void)hipMalloc(&d, n * 8);
  std::vector<unsigned> h(2 * n), first;
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 10192);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 10192, 0, d, 100000);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; XCC_ID[3:0]
    std::map<unsigned, int> per_simd;
    for (int b = 0; b < n; ++b) {
      const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15;
      const unsigned key = (x << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
      per_simd[key]++;
    }
    int mx = 0, mn = 1 << 30;
    for (auto& kv : per_simd) { mx = kv.second > mx ? kv.second : mx; mn = kv.second < mn ? kv.second : mn; }
    int same = -1;
    if (rep == 0) first = h; else { same = 0; for (int b = 0; b < n; ++b) same += ((h[2 * b] >> 4) == (first[2 * b] >> 4)) && (h[2 * b + 1] == first[2 * b + 1]); }
    printf("{\"launch\": %d, \"simds_used\": %zu, \"min_per_simd\": %d, \"max_per_simd\": %d, \"blocks_on_same_simd_as_launch0\": %d}\n", rep, per_simd.size(), mn, mx, same);
    if (rep < 3) {
      char fn[64]; snprintf(fn, sizeof fn, "gpurun_out/dispatch_map_%d.txt", rep);
      FILE* f = fopen(fn, "w");
      if (f) { for (int b = 0; b < n; ++b) { const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15; fprintf(f, "%d %u %u %u %u %u\n", b, x, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15); } fclose(f); }
    }
    if (rep == 0) {
      printf("block: xcc se sh cu simd wave\n");
      for (int b = 0; b < 40; ++b) {
        const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15;
        printf("%4d: %u %u %u %2u %u %u\n", b, x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
      }
      for (int b : {256, 257, 264, 512, 1024, 1032, 2048, 2056, 4088, 4095}) {
        const unsigned hw = h[2 * b], x = h[2 * b + 1] & 15;
        printf("%4d: %u %u %u %2u %u %u\n", b, x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
      }
    }
  }
  return 0;
}
