/* A caller of libmocca_hip.so that is NOT Python and knows nothing about torch: plain C over include/mocca.h, device buffers from the
 * HIP runtime's C API.  tests/test_gpu_abi_client.py builds it with gcc, runs it on the GPU box and compares what it prints with the
 * same episode stepped through the Python binding, bit for bit -- the drop-in boundary is the C ABI, PyTorch is one of its users.
 *
 *   abi_client <model blob file> <task id> <n_envs> <steps> <seed>
 * prints one line per env: the reward and done flag of the last step and the FNV-1a hash of its observation row's bytes; then one line
 * "episodes <count> <sum of lengths> <truncated> | totals <episodes> <lengths> <truncated>": Monitor's statistics as the step kernel wrote them
 * (ABI 7, mocca_set_episode_stats) -- read record by record from a ring in pinned HOST memory (hipHostMalloc; one slot per step here), and the
 * device-side totals beside them. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mocca.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MOCCA(x) do { int rc_ = (x); if (rc_ != MOCCA_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, mocca_last_error(h)); return 3; } } while (0)

int main(int argc, char **argv) {
  if (argc != 6) { fprintf(stderr, "usage: abi_client blob task n_envs steps seed\n"); return 1; }
  const int task = atoi(argv[2]), n = atoi(argv[3]), steps = atoi(argv[4]);
  const uint64_t seed = strtoull(argv[5], NULL, 10);
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  const size_t want = mocca_model_sizeof();
  void *blob = malloc(want);
  if (fread(blob, 1, want, f) != want) { fprintf(stderr, "blob file is not %zu bytes\n", want); return 1; }
  fclose(f);
  if (mocca_abi_version() != MOCCA_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  mocca_handle h = NULL;
  int rc = mocca_create(blob, want, task, n, 0, &h);
  if (rc != MOCCA_OK) { fprintf(stderr, "mocca_create: %d %s\n", rc, mocca_last_error(NULL)); return 3; }
  const int od = mocca_obs_dim(h), ad = mocca_act_dim(h);
  float *d_act, *d_obs, *d_rew, *d_term;
  uint8_t *d_done;
  int32_t *d_info;
  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));
  CHECK_HIP(hipMalloc((void **)&d_act, (size_t)n * ad * 4)); CHECK_HIP(hipMalloc((void **)&d_obs, (size_t)n * od * 4));
  CHECK_HIP(hipMalloc((void **)&d_rew, (size_t)n * 4)); CHECK_HIP(hipMalloc((void **)&d_done, (size_t)n));
  CHECK_HIP(hipMalloc((void **)&d_info, (size_t)n * 4)); CHECK_HIP(hipMalloc((void **)&d_term, (size_t)n * od * 4));
  float *h_act = (float *)malloc((size_t)n * ad * 4), *h_obs = (float *)malloc((size_t)n * od * 4), *h_rew = (float *)malloc((size_t)n * 4);
  uint8_t *h_done = (uint8_t *)malloc((size_t)n);
  CHECK_MOCCA(mocca_set_param(h, MOCCA_PARAM_AUTO_RESET, 1.0));
  CHECK_MOCCA(mocca_set_terminal_obs_buffer(h, d_term));
  /* Monitor / TimeLimitMask inside the launch: masks and totals on the device, the finished envs' records straight into host memory */
  float *d_masks, *d_bad, *d_tot;
  mocca_episode_rec *ring;
  CHECK_HIP(hipMalloc((void **)&d_masks, (size_t)n * 4)); CHECK_HIP(hipMalloc((void **)&d_bad, (size_t)n * 4)); CHECK_HIP(hipMalloc((void **)&d_tot, 16));
  CHECK_HIP(hipMemset(d_tot, 0, 16));
  CHECK_HIP(hipHostMalloc((void **)&ring, (size_t)(steps + 1) * n * sizeof(mocca_episode_rec), 0));
  memset(ring, 0, (size_t)(steps + 1) * n * sizeof(mocca_episode_rec));
  CHECK_MOCCA(mocca_set_episode_stats(h, d_masks, d_bad, d_tot, ring, steps + 1, (size_t)n * sizeof(mocca_episode_rec)));
  const uint32_t first_serial = mocca_episode_serial(h);
  CHECK_MOCCA(mocca_reset(h, NULL, seed, d_obs, s));
  uint32_t lcg = 12345u;   /* the same actions the Python side generates: a 24-bit LCG mapped to [-1, 1) */
  for (int t = 0; t < steps; ++t) {
    for (int i = 0; i < n * ad; ++i) { lcg = lcg * 1664525u + 1013904223u; h_act[i] = (float)(lcg >> 8) * (2.0f / 16777216.0f) - 1.0f; }
    CHECK_HIP(hipMemcpyAsync(d_act, h_act, (size_t)n * ad * 4, hipMemcpyHostToDevice, s));
    CHECK_MOCCA(mocca_step(h, d_act, d_obs, d_rew, d_done, d_info, s));
  }
  CHECK_HIP(hipMemcpyAsync(h_obs, d_obs, (size_t)n * od * 4, hipMemcpyDeviceToHost, s));
  CHECK_HIP(hipMemcpyAsync(h_rew, d_rew, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  CHECK_HIP(hipMemcpyAsync(h_done, d_done, (size_t)n, hipMemcpyDeviceToHost, s));
  CHECK_HIP(hipStreamSynchronize(s));
  for (int e = 0; e < n; ++e) {
    uint32_t hash = 2166136261u;
    const unsigned char *b = (const unsigned char *)(h_obs + (size_t)e * od);
    for (int i = 0; i < od * 4; ++i) { hash ^= b[i]; hash *= 16777619u; }
    uint32_t rb;
    memcpy(&rb, &h_rew[e], 4);
    printf("%d %08x %d %08x\n", e, rb, (int)h_done[e], hash);
  }
  {
    long episodes = 0, lengths = 0, truncated = 0;
    for (int t = 0; t < steps; ++t) {          /* step t stamped its records with serial first_serial + t and wrote slot serial mod (steps + 1) */
      const uint32_t serial = first_serial + (uint32_t)t;
      const mocca_episode_rec *slot = ring + (size_t)(serial % (uint32_t)(steps + 1)) * n;
      for (int e = 0; e < n; ++e)
        if (slot[e].serial == serial) { ++episodes; lengths += slot[e].length; truncated += (slot[e].flags & 2u) ? 1 : 0; }
    }
    float tot[4];
    CHECK_HIP(hipMemcpy(tot, d_tot, 16, hipMemcpyDeviceToHost));
    printf("episodes %ld %ld %ld | totals %.0f %.0f %.0f\n", episodes, lengths, truncated, tot[2], tot[1], tot[3]);
  }
  CHECK_MOCCA(mocca_set_episode_stats(h, NULL, NULL, NULL, NULL, 0, 0));
  CHECK_MOCCA(mocca_destroy(h));
  return 0;
}
