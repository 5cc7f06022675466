// layout_bench.hip -- SURVEY.md section 7.3 "measure both": the step kernel's load/store phase with the per-env records laid
// out ENV-MAJOR (what the stepper uses: dyn[N][96], one wave reads its env's 89 floats as consecutive addresses) against
// FIELD-MAJOR structure-of-arrays (dyn[96][N]: field f of env e at f * N + e), under the stepper's mapping of ONE WAVE PER
// ENV, and -- for reference -- field-major under the mapping it is made for, one THREAD per env.
// Each variant loads the record into LDS / registers, does a token amount of arithmetic and stores it back: the HBM / L2
// side of one env.step() without the physics.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_bench tools/layout_bench.hip && /tmp/layout_bench [n_envs]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int REC = 96, USED = 89;  // floats per env in the dynamic-state buffer, floats the Walker3D kernel touches

__global__ __launch_bounds__(64) void env_major_wave_per_env(float* dyn, int n) {
  __shared__ float L[REC];
  const int env = blockIdx.x, lane = threadIdx.x;
  float* st = dyn + (size_t)env * REC;
  L[lane] = st[lane];                       // lanes 0..63: one 256-B coalesced read
  if (lane < USED - 64) L[64 + lane] = st[64 + lane];
  __syncthreads();
  const float a = L[(lane * 7) % USED] * 1.0001f;
  const float b = lane < USED - 64 ? L[(lane * 5 + 64) % USED] * 0.9999f : 0.0f;
  __syncthreads();
  st[lane] = a;
  if (lane < USED - 64) st[64 + lane] = b;
}

__global__ __launch_bounds__(64) void field_major_wave_per_env(float* dyn, int n) {
  __shared__ float L[REC];
  const int env = blockIdx.x, lane = threadIdx.x;
  L[lane] = dyn[(size_t)lane * n + env];    // every lane touches a different 64-B line
  if (lane < USED - 64) L[64 + lane] = dyn[(size_t)(64 + lane) * n + env];
  __syncthreads();
  const float a = L[(lane * 7) % USED] * 1.0001f;
  const float b = lane < USED - 64 ? L[(lane * 5 + 64) % USED] * 0.9999f : 0.0f;
  __syncthreads();
  dyn[(size_t)lane * n + env] = a;
  if (lane < USED - 64) dyn[(size_t)(64 + lane) * n + env] = b;
}

__global__ __launch_bounds__(64) void field_major_thread_per_env(float* dyn, int n) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= n) return;
  float r[USED];
#pragma unroll
  for (int f = 0; f < USED; ++f) r[f] = dyn[(size_t)f * n + env];   // coalesced across the wave's 64 envs
#pragma unroll
  for (int f = 0; f < USED; ++f) dyn[(size_t)f * n + env] = r[(f * 7) % USED] * 1.0001f;
}

template <class K>
static float time_kernel(K k, dim3 grid, float* d, int n, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, grid, dim3(64), 0, 0, d, n);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(64), 0, 0, d, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return 1000.0f * ms / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096, reps = 2000;
  float* d = nullptr;
  if (hipMalloc(&d, (size_t)n * REC * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  std::vector<float> h((size_t)n * REC, 1.0f);
  hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  const float t_env = time_kernel(env_major_wave_per_env, dim3(n), d, n, reps);
  const float t_fld = time_kernel(field_major_wave_per_env, dim3(n), d, n, reps);
  const float t_thr = time_kernel(field_major_thread_per_env, dim3((n + 63) / 64), d, n, reps);
  const double bytes = 2.0 * n * USED * 4;
  printf("{\"n_envs\": %d, \"bytes_moved\": %.0f, \"env_major_wave_per_env_us\": %.2f, \"field_major_wave_per_env_us\": %.2f, "
         "\"field_major_thread_per_env_us\": %.2f, \"note\": \"load + store of the dynamic-state record only; the step kernel itself takes ~160 us\"}\n",
         n, bytes, t_env, t_fld, t_thr);
  hipFree(d);
  return 0;
}
