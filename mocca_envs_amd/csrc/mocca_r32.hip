// mocca_r32.hip -- the COMPACT instance of the step kernel: the same device source (mocca_device.h, mocca_kernels.h) compiled with
// MAXR = 32 rows / 10 contacts per env and the compact LDS layout (7.2 KB per env instead of 10.2 KB: five or more resident waves per
// SIMD instead of four), in its own namespace so that both instances live in one library.  mocca_create picks it for a blob with
// max_rows <= 32, max_contacts <= 10 and no loop closures (mocca_api.hip: compact_ok); reset / observe / task-step kernels keep no state
// in LDS across launches and exist in the 48-row instance only.
#define MOCCA_NS mocca_r32
#define MOCCA_MAXR 32
#define MOCCA_COMPACT 1
#ifndef MOCCA_R32_WAVES   // resident waves per SIMD the register budget is set for (512 VGPRs / waves, in steps of 8): 5 -> 96 VGPRs
#define MOCCA_R32_WAVES 5
#endif
#define MOCCA_WAVES_PER_EU MOCCA_R32_WAVES
#include <hip/hip_runtime.h>

#include "mocca.h"
#include "mocca_kernels.h"

namespace mocca_r32 {

template <class T, int TASK> struct LaunchStep {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_step_kernel<T, TASK>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct KernelInfo {
  static void run(hipFuncAttributes* fa, int* nb, hipError_t* e) {
    *e = hipFuncGetAttributes(fa, (const void*)mocca_step_kernel<T, TASK>);
    if (*e == hipSuccess) *e = hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, mocca_step_kernel<T, TASK>, 64, 0);
  }
};

}  // namespace mocca_r32

// `args` is a mocca::StepArgs: the struct is declared by the same header in both namespaces (same layout by construction; the size is
// checked on the caller's side)
extern "C" __attribute__((visibility("hidden"))) size_t mocca_r32_args_sizeof(void) { return sizeof(mocca_r32::StepArgs); }
extern "C" __attribute__((visibility("hidden"))) int mocca_r32_max_rows(void) { return mocca_r32::MAXR; }
extern "C" __attribute__((visibility("hidden"))) int mocca_r32_max_contacts(void) { return mocca_r32::MAXC; }
extern "C" __attribute__((visibility("hidden"))) void mocca_r32_launch_step(int topo, int task_id, int n, hipStream_t s, const void* args) {
  mocca_r32::StepArgs a;
  __builtin_memcpy(&a, args, sizeof(a));
  mocca_r32::dispatch<mocca_r32::LaunchStep>(topo, task_id, n, s, a);
}
extern "C" __attribute__((visibility("hidden"))) void mocca_r32_kernel_info(int topo, int task_id, hipFuncAttributes* fa, int* nb, hipError_t* e) {
  mocca_r32::dispatch<mocca_r32::KernelInfo>(topo, task_id, fa, nb, e);
}
