// fetch_size_probe.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for THIS project's access pattern (one wave per env, each
// lane one dword of the env's record) against a known byte count, next to the guide's calibrated case (a 16-B-per-lane streaming read, which
// FETCH_SIZE reports at exactly half: MI355X_MICROARCH.md "HBM").  Settles which factor applies to profiles/traffic.json.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_size_probe tools/fetch_size_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --stats -d out -- /tmp/fetch_size_probe      (and once more with --pmc WRITE_SIZE)
// Kernels (each launched once, N = 1 Mi records or 256 MiB, far beyond L2 and the Infinity Cache between two touches of a line):
//   record_read_dword   : wave w reads words 0..54 of record w (stride 96 floats = 384 B, like the dyn record): 220 B touched = 4 lines of 64 B
//   record_read_write   : ... and writes them back (the step kernel's load_dyn / store_dyn)
//   stream_read_dwordx4 : 16 B per lane, fully coalesced: the guide's calibrated pattern (FETCH_SIZE = bytes / 2)
//   stream_read_dword   : 4 B per lane, fully coalesced
//   step_io_pattern     : the step kernel's own reads and writes per env (352 B read, 481 B written), for the calibration factor of traffic.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void record_read_dword(const float* __restrict__ src, float* __restrict__ sink, int stride, int words) {
  const int lane = threadIdx.x;
  float v = 0.0f;
  if (lane < words) v = src[(size_t)blockIdx.x * stride + lane];
  if (v == 123456.0f) sink[0] = v;   // never true: keeps the load
}
__global__ void record_read_write(float* __restrict__ buf, int stride, int words) {
  const int lane = threadIdx.x;
  if (lane < words) { float* p = buf + (size_t)blockIdx.x * stride + lane; *p = *p + 1.0f; }
}
__global__ void stream_read_dwordx4(const float4* __restrict__ src, float* __restrict__ sink, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) { const float4 v = src[i]; if (v.x == 123456.0f) sink[0] = v.y; }
}
__global__ void stream_read_dword(const float* __restrict__ src, float* __restrict__ sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = src[i]; if (v == 123456.0f) sink[0] = v; }
}

// the step kernel's own HBM pattern, nothing else: per env (= wave) read dyn[55 of stride 96] + task[40 of stride 40] + act[21 of stride 21],
// write dyn[55] + task[40] + obs[52 of stride 52] + rew[1] + done[1 byte]  (352 B read + 484 B written per env-step, counting done as 4)
__global__ void step_io_pattern(float* __restrict__ dyn, unsigned* __restrict__ task, const float* __restrict__ act, float* __restrict__ obs,
                                float* __restrict__ rew, unsigned char* __restrict__ done) {
  const int lane = threadIdx.x;
  const size_t e = blockIdx.x;
  float d = 0.0f, a = 0.0f; unsigned t = 0u;
  if (lane < 55) d = dyn[e * 96 + lane];
  if (lane < 40) t = task[e * 40 + lane];
  if (lane < 21) a = act[e * 21 + lane];
  const float x = d + a + (float)(t & 1u);
  if (lane < 55) dyn[e * 96 + lane] = x;
  if (lane < 40) task[e * 40 + lane] = t + 1u;
  if (lane < 52) obs[e * 52 + lane] = x;
  if (lane == 0) { rew[e] = x; done[e] = (unsigned char)(t & 1u); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int n_rec = 1 << 20, stride = 96, words = 55;
  const size_t rec_bytes = (size_t)n_rec * stride * 4;           // 384 MiB
  const size_t stream_bytes = (size_t)256 << 20;                 // 256 MiB
  float *rec = nullptr, *strm = nullptr, *sink = nullptr, *flush = nullptr;
  CK(hipMalloc(&rec, rec_bytes)); CK(hipMalloc(&strm, stream_bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&flush, (size_t)512 << 20));
  CK(hipMemset(rec, 0, rec_bytes)); CK(hipMemset(strm, 0, stream_bytes));
  auto cold = [&]() { (void)hipMemset(flush, 1, (size_t)512 << 20); (void)hipDeviceSynchronize(); };   // push the buffers out of L2 / Infinity Cache
  cold(); hipLaunchKernelGGL(record_read_dword, dim3(n_rec), dim3(64), 0, 0, rec, sink, stride, words); CK(hipDeviceSynchronize());
  cold(); hipLaunchKernelGGL(record_read_write, dim3(n_rec), dim3(64), 0, 0, rec, stride, words); CK(hipDeviceSynchronize());
  cold(); hipLaunchKernelGGL(stream_read_dwordx4, dim3((unsigned)(stream_bytes / 16 / 256)), dim3(256), 0, 0, (const float4*)strm, sink, stream_bytes / 16); CK(hipDeviceSynchronize());
  cold(); hipLaunchKernelGGL(stream_read_dword, dim3((unsigned)(stream_bytes / 4 / 256)), dim3(256), 0, 0, strm, sink, stream_bytes / 4); CK(hipDeviceSynchronize());
  {
    float *dyn = rec, *act = nullptr, *obs = nullptr, *rew = nullptr; unsigned* task = nullptr; unsigned char* done = nullptr;
    CK(hipMalloc(&task, (size_t)n_rec * 160)); CK(hipMalloc(&act, (size_t)n_rec * 84)); CK(hipMalloc(&obs, (size_t)n_rec * 208));
    CK(hipMalloc(&rew, (size_t)n_rec * 4)); CK(hipMalloc(&done, (size_t)n_rec));
    CK(hipMemset(task, 0, (size_t)n_rec * 160)); CK(hipMemset(act, 0, (size_t)n_rec * 84));
    cold(); hipLaunchKernelGGL(step_io_pattern, dim3(n_rec), dim3(64), 0, 0, dyn, task, act, obs, rew, done); CK(hipDeviceSynchronize());
  }
  printf("{\"records\": %d, \"record_bytes_touched\": %d, \"record_lines_64B\": %d, \"record_stride_bytes\": %d, \"stream_bytes\": %zu}\n",
         n_rec, words * 4, (words * 4 + 63) / 64, stride * 4, stream_bytes);
  return 0;
}
