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
// C: the friction loop of the kernel up to r02 (dynamic lane index, exec-mask commit, A entries from LDS, bound from the normal row)
__device__ __forceinline__ float setlane(float v, int l, float old) {
  unsigned long long m, saved;
  asm volatile("s_lshl_b64 %1, 1, %4\n\ts_and_saveexec_b64 %2, %1\n\tv_mov_b32 %0, %3\n\ts_mov_b64 exec, %2"
               : "+v"(old), "=&s"(m), "=&s"(saved) : "v"(v), "s"(l) : "scc");
  return old;
}
__global__ __launch_bounds__(64) void k_c(float* o, long long* cyc, int iters, int nc) {
  __shared__ float A[48 * 48 + 64];
  for (int i = threadIdx.x; i < 48 * 48 + 64; i += 64) A[i] = 0.001f * (i % 97);
  __syncthreads();
  const int lc = threadIdx.x < 48 ? threadIdx.x : 47;
  const float* Acol = A + lc;
  float y = 0.01f * threadIdx.x - 0.2f, lam = 0.0f, invdiag = 0.5f, mu = 0.8f;
  const int nlf = 48 - 3 * nc, r_fr = nlf + nc;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float a = Acol[48 * r_fr];
#pragma unroll 1
    for (int i = 0; i < nc; ++i) {
      const float lm = mu * readlane(lam, nlf + i);
      const int rr = r_fr + 2 * i;
      const float a1 = Acol[48 * (rr + 1)], a2 = Acol[48 * (rr + 2 < 48 ? rr + 2 : 47)];
      const float as = a * invdiag, as1 = a1 * invdiag;
      float nl_ = __builtin_amdgcn_fmed3f(y, -lm, lm);
      float dl = readlane(nl_ - lam, rr);
      lam = setlane(nl_, rr, lam);
      y = fmaf(-as, dl, y);
      nl_ = __builtin_amdgcn_fmed3f(y, -lm, lm);
      dl = readlane(nl_ - lam, rr + 1);
      lam = setlane(nl_, rr + 1, lam);
      y = fmaf(-as1, dl, y);
      a = a2;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = y + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// D: friction rows on STATIC lanes (top of the wave, contact i on lanes 46 - 2i / 47 - 2i), per-lane bound refreshed once per iteration
template <int L> __device__ __forceinline__ float writelane_i(float v, float old) { asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(L)); return old; }
template <int I> __device__ __forceinline__ void visit_d(const float* Acol, int nc, float& y, float& lam, float lmv, float invdiag) {
  if constexpr (I < 12) {
    if (I >= nc) return;
    constexpr int R0 = 46 - 2 * I, R1 = 47 - 2 * I;
    const float as0 = Acol[48 * R0] * invdiag, as1 = Acol[48 * R1] * invdiag;
    float nl_ = __builtin_amdgcn_fmed3f(y, -lmv, lmv);
    float dl = readlane(nl_ - lam, R0);
    lam = writelane_i<R0>(readlane(nl_, R0), lam);
    y = fmaf(-as0, dl, y);
    nl_ = __builtin_amdgcn_fmed3f(y, -lmv, lmv);
    dl = readlane(nl_ - lam, R1);
    lam = writelane_i<R1>(readlane(nl_, R1), lam);
    y = fmaf(-as1, dl, y);
    visit_d<I + 1>(Acol, nc, y, lam, lmv, invdiag);
  }
}
__global__ __launch_bounds__(64) void k_d(float* o, long long* cyc, int iters, int nc) {
  __shared__ float A[48 * 48 + 64];
  for (int i = threadIdx.x; i < 48 * 48 + 64; i += 64) A[i] = 0.001f * (i % 97);
  __syncthreads();
  const int lc = threadIdx.x < 48 ? threadIdx.x : 47;
  const float* Acol = A + lc;
  float y = 0.01f * threadIdx.x - 0.2f, lam = 0.0f, invdiag = 0.5f, mu = 0.8f;
  const int nrow = threadIdx.x >= 24 && threadIdx.x < 48 ? (47 - threadIdx.x) >> 1 : 0;   // lane of the own contact's normal row
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const float lmv = mu * __shfl(lam, nrow, 64);
    visit_d<0>(Acol, nc, y, lam, lmv, invdiag);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = y + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// E: three-link chain z -> v_max -> v_readlane -> v_fma -> z with the bound lo - lambda formed off the chain and lambda updated by a
// constant lane mask (z = y - lambda is what every lane carries; the gain of a lane's own row is 1, so z_r drops by the step itself)
template <int RR> __device__ __forceinline__ void visit_e(float& z, float& lam, float as, float lo) {
  if constexpr (RR < 48) {
    const float bnd = lo - lam;                                    // off the chain: lam_r is known since the previous iteration
    float t;
    asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(z), "v"(bnd));      // plain v_max: no canonicalisation sequence
    const float d = readlane(t, RR);
    z = fmaf(-as, d, z);
    float inc;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(inc) : "v"(d), "s"(1ull << RR));
    lam += inc;
    visit_e<RR + 1>(z, lam, as, lo);
  }
}
__global__ __launch_bounds__(64) void k_e(float* o, long long* cyc, int iters) {
  float z = 0.01f * threadIdx.x - 0.2f, lam = 0.0f, as = 0.001f * (threadIdx.x + 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) visit_e<0>(z, lam, as, 0.0f);
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = z + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// F: TWO ENVS PER WAVE (lanes 0..31 env A, 32..63 env B), the kernel's visit form A: row RR of both envs is visited at once.  Each half needs
// ITS env's step: two v_readlane + one select on a constant half mask, and the commit takes a two-bit lane mask -- 7 VALU for two envs
// instead of 5 for one ("Config 5").  32 visits per iteration (at most 32 rows per env).
template <int RR> __device__ __forceinline__ void visit_f(float& y, float& lam, float as) {
  if constexpr (RR < 32) {
    const float nl = __builtin_amdgcn_fmed3f(y, 0.0f, 1e30f);
    const float st = nl - lam;
    const float da = readlane(st, RR), db = readlane(st, RR + 32);
    float dl, nlam;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dl) : "v"(da), "v"(db), "s"(0xFFFFFFFF00000000ull));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(nlam) : "v"(lam), "v"(nl), "s"((1ull << RR) | (1ull << (RR + 32))));
    lam = nlam;
    y = fmaf(-as, dl, y);
    visit_f<RR + 1>(y, lam, as);
  }
}
__global__ __launch_bounds__(64) void k_f(float* o, long long* cyc, int iters) {
  float y = 0.01f * threadIdx.x - 0.2f, lam = 0.0f, as = 0.001f * (threadIdx.x + 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) visit_f<0>(y, lam, as);
  const long long t1 = __builtin_amdgcn_s_memtime();
  o[blockIdx.x * 64 + threadIdx.x] = y + lam;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
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
  for (int waves : {1, 4}) {
    const int blocks = 256 * 4 * waves, iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_e, dim3(blocks), dim3(64), 0, 0, d, c, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("{\"waves_per_simd\": %d, \"form\": \"E max-readlane-fma, bound and lambda off the chain\", \"cycles_per_visit\": %.1f}\n", waves, s / 1024 / (48.0 * iters));
  }
  for (int waves : {1, 2, 4}) {
    const int blocks = 256 * 4 * waves, iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_f, dim3(blocks), dim3(64), 0, 0, d, c, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("{\"waves_per_simd\": %d, \"form\": \"F two envs per wave: 2 readlane + select + 2-lane commit\", \"cycles_per_visit_serving_two_envs\": %.1f}\n", waves, s / 1024 / (32.0 * iters));
  }
  for (int waves : {1, 4}) {
    const int blocks = 256 * 4 * waves, iters = 200, nc = 12;
    for (int v = 0; v < 2; ++v) {
      for (int rep = 0; rep < 3; ++rep) { if (v == 0) hipLaunchKernelGGL(k_c, dim3(blocks), dim3(64), 0, 0, d, c, iters, nc); else hipLaunchKernelGGL(k_d, dim3(blocks), dim3(64), 0, 0, d, c, iters, nc); }
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, c, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
      printf("{\"waves_per_simd\": %d, \"form\": \"%s\", \"cycles_per_friction_visit\": %.1f}\n", waves, v == 0 ? "C r02 friction loop (dynamic lanes)" : "D friction rows on static lanes", s / 1024 / (2.0 * nc * iters));
    }
  }
  return 0;
}
