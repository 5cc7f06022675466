// mocca_r64.hip -- the ACCURACY instance of the step kernel: the same device source (mocca_device.h, mocca_kernels.h) compiled with
// MAXR = 64 rows / 20 contacts per env -- every lane of the wave a constraint row, the most the lane = row solver can hold -- in its own
// namespace.  Bullet caps neither contacts nor rows; the product's 48 / 12 bind in 11 % of the envs of Stepper curriculum 9 at least once
// (up to 61 rows wanted, profiles/archive/r04_cap_pressure.jsonl).  17.4 KB of LDS and a two-waves-per-SIMD register budget per env: speed is not
// this instance's job.  mocca_create picks it for a blob with max_rows > 48 or max_contacts > 12 (VecEnv(..., max_rows=64, max_contacts=20));
// MOCCA_PARAM_KERNEL_VARIANT = 2 forces it for any blob (A/B runs: on the same blob it is bit-identical to the 48-row instance).
#define MOCCA_NS mocca_r64
#define MOCCA_MAXR 64
#define MOCCA_COMPACT 0
#define MOCCA_WAVES_PER_EU 2
#include <hip/hip_runtime.h>

#include "mocca.h"
#include "mocca_kernels.h"

namespace mocca_r64 {

template <class T, int TASK> struct LaunchStep {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_step_kernel<T, TASK>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct KernelInfo {
  static void run(hipFuncAttributes* fa, int* nb, hipError_t* e) {
    *e = hipFuncGetAttributes(fa, (const void*)mocca_step_kernel<T, TASK>);
    if (*e == hipSuccess) *e = hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, mocca_step_kernel<T, TASK>, 64, 0);
  }
};

}  // namespace mocca_r64

// `args` is a mocca::StepArgs: the struct is declared by the same header in every namespace (same layout by construction; the size is
// checked on the caller's side)
extern "C" __attribute__((visibility("hidden"))) size_t mocca_r64_args_sizeof(void) { return sizeof(mocca_r64::StepArgs); }
extern "C" __attribute__((visibility("hidden"))) int mocca_r64_max_rows(void) { return mocca_r64::MAXR; }
extern "C" __attribute__((visibility("hidden"))) int mocca_r64_max_contacts(void) { return mocca_r64::MAXC; }
extern "C" __attribute__((visibility("hidden"))) void mocca_r64_launch_step(int topo, int task_id, int n, hipStream_t s, const void* args) {
  mocca_r64::StepArgs a;
  __builtin_memcpy(&a, args, sizeof(a));
  mocca_r64::dispatch<mocca_r64::LaunchStep>(topo, task_id, n, s, a);
}
extern "C" __attribute__((visibility("hidden"))) void mocca_r64_kernel_info(int topo, int task_id, hipFuncAttributes* fa, int* nb, hipError_t* e) {
  mocca_r64::dispatch<mocca_r64::KernelInfo>(topo, task_id, fa, nb, e);
}
