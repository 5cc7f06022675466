// mocca_task.hip -- the INJECT = true kernel instances: env.step()'s task layer on caller-supplied contact flags
// (mocca_task_step) and resets that read recorded uniforms (mocca_set_draw_tape).  Same kernel source as the physics
// path (mocca_kernels.h), separate translation unit so the two compile in parallel.
#include <hip/hip_runtime.h>

#include "mocca.h"
#include "mocca_kernels.h"

namespace mocca {

template <class T, int TASK> struct LaunchTaskStep {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_step_kernel<T, TASK, true>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct LaunchTapedReset {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_reset_kernel<T, TASK, true>), dim3(n), dim3(64), 0, s, a); }
};

void launch_task_step(int topo, int task_id, int n, hipStream_t s, StepArgs a) { dispatch<LaunchTaskStep>(topo, task_id, n, s, a); }
void launch_taped_reset(int topo, int task_id, int n, hipStream_t s, StepArgs a) { dispatch<LaunchTapedReset>(topo, task_id, n, s, a); }

}  // namespace mocca
