// mocca_api.hip -- kernels' entry points and the C ABI of libmocca_hip.so (include/mocca.h).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see mocca_envs_amd/build.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mocca.h"
#include "mocca_kernels.h"

using namespace mocca;

// the compact instance of the step kernel (mocca_r32.hip)
extern "C" size_t mocca_r32_args_sizeof(void);
extern "C" int mocca_r32_max_rows(void);
extern "C" int mocca_r32_max_contacts(void);
extern "C" void mocca_r32_launch_step(int topo, int task_id, int n, hipStream_t s, const void* args);
extern "C" void mocca_r32_kernel_info(int topo, int task_id, hipFuncAttributes* fa, int* nb, hipError_t* e);
// the 64-row / 20-contact accuracy instance (mocca_r64.hip)
extern "C" size_t mocca_r64_args_sizeof(void);
extern "C" int mocca_r64_max_rows(void);
extern "C" int mocca_r64_max_contacts(void);
extern "C" void mocca_r64_launch_step(int topo, int task_id, int n, hipStream_t s, const void* args);
extern "C" void mocca_r64_kernel_info(int topo, int task_id, hipFuncAttributes* fa, int* nb, hipError_t* e);

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
__global__ void copy_param_kernel(float* dst, const float* src, bool broadcast, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[broadcast ? 0 : i];
}
__global__ void set_task_word_kernel(uint32_t* task, int word, const float* vals, float scalar, int use_scalar, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) task[(size_t)i * MOCCA_TASK_WORDS + word] = __float_as_uint(use_scalar ? scalar : vals[i]);
}

// Launch order of the step kernel (MOCCA_PARAM_ORDER_EVERY): a counting sort of the envs by the constraint-row count their last step ended
// with (task word 23, 0 .. 63), most rows first.  One workgroup; the order inside a bucket is whatever the atomics make it -- it only
// decides when an env's wave starts.  ~6 us for 8192 envs, every K-th step.
__global__ __launch_bounds__(1024) void order_by_rows_kernel(const uint32_t* task, int32_t* order, int n) {
  __shared__ int hist[64], start[64];
  const int t = threadIdx.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int e = t; e < n; e += 1024) {
    const uint32_t r = task[(size_t)e * MOCCA_TASK_WORDS + T_RES23];
    atomicAdd(&hist[63 - (r > 63u ? 63u : r)], 1);   // bucket 0 = heaviest
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < 64; ++k) { start[k] = acc; acc += hist[k]; }
  }
  __syncthreads();
  for (int e = t; e < n; e += 1024) {
    const uint32_t r = task[(size_t)e * MOCCA_TASK_WORDS + T_RES23];
    order[atomicAdd(&start[63 - (r > 63u ? 63u : r)], 1)] = e;
  }
}

struct mocca_ctx {
  MoccaModel model;
  int task_id = 0, n_envs = 0, device = 0, obs_dim = 0;
  MoccaModel* d_model = nullptr;
  int topo = 0;  // TOPO_*
  float* d_dyn = nullptr;
  uint32_t* d_task = nullptr;
  float* d_terrain = nullptr;
  int auto_reset = 0, eval_mode = 0, random_pose = 1, curriculum = 0, host_retarget = 0, env_offset = 0, random_reward = 0;
  float gain = 1.0f;
  bool compact = false;        // the blob fits the compact step-kernel instance (compact_ok); MOCCA_PARAM_KERNEL_VARIANT = 1 overrides
  bool wide = false;           // the blob's caps exceed 48 rows / 12 contacts: the 64-row accuracy instance (mocca_r64.hip)
  int force_full = 0;          // MOCCA_PARAM_KERNEL_VARIANT: 1 forces the 48-row instance, 2 the 64-row one
  int persist_warm = 0;        // MOCCA_PARAM_PERSIST_IMPULSES
  int pace = -18;              // MOCCA_PARAM_PACE_TICKS: self-calibrating pace priorities, 18/16 of the previous launch's mean wave time (profiles/archive/r04_pace_*.jsonl)
  unsigned long long* d_pace_acc = nullptr;  // self-calibration samples of the pace, one packed word (StepArgs.pace_acc), owned by the handle
  // Monitor / TimeLimitMask inside the launch (mocca_set_episode_stats)
  float* d_ep_ret = nullptr;       // [N] running episode returns, owned by the handle
  float *ep_masks = nullptr, *ep_bad = nullptr, *ep_totals = nullptr;   // caller-owned
  char* ep_rec = nullptr;          // caller-owned record ring: n_slots slots of [N] x 16 bytes, ep_stride bytes apart
  int ep_slots = 0;
  size_t ep_stride = 0;
  uint32_t ep_serial = 1;          // stamped into the records of the next mocca_step (0 never: a zeroed ring holds no record)
  int order_every = 0;         // MOCCA_PARAM_ORDER_EVERY: re-sort the launch order every K steps (0: envs run in index order)
  int order_age = 0;           // steps since the last sort
  int32_t* d_order = nullptr;  // [N] the permutation, owned by the handle
  bool gain_pending = false;   // a scalar MOCCA_PARAM_APPLIED_GAIN not yet written into the task records (flush_pending)
  float* final_obs = nullptr;  // caller-owned (mocca_set_terminal_obs_buffer)
  float* d_pvec[3] = {nullptr, nullptr, nullptr};  // per-env curriculum / eval_mode / applied_gain (mocca_set_param_v), lazily allocated
  bool pvec_on[3] = {false, false, false};
  const float* tape = nullptr;  // caller-owned (mocca_set_draw_tape)
  int tape_n = 0;
  int32_t* dbg = nullptr;       // caller-owned (mocca_set_debug_buffer)
  int prio = MOCCA_PRIO_T1 + 64 * MOCCA_PRIO_T2 + 4096 * MOCCA_PRIO_T3;   // MOCCA_PARAM_ISSUE_PRIORITY
  uint64_t seed = 0;
  float* d_traj = nullptr;      // Cassie mocap / phase envs: the motion table (mocca_set_trajectory), owned by the handle
  int traj_n = 0;
  double traj_tmax = 0.0, traj_cstep = 0.0;
  float* d_hf = nullptr;        // planner envs: the height field (mocca_set_heightfield), owned by the handle
  int hf_rows = 0, hf_cols = 0;
  float hf_scale = 0.0f;
  std::string err;
};

static thread_local std::string g_err;

#define HIP_TRY(h, expr)                                                         \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
      return MOCCA_E_HIP;                                                        \
    }                                                                            \
  } while (0)

// does the blob give mass or inertia to a link the compiled topology T treats as massless?
template <class T>
static bool massive_intermediates(const MoccaModel& m) {
  for (int b = 1; b < T::NB; ++b)
    if (T::massless(b)) {
      bool zero = m.mass[b] == 0.0f;
      for (int i = 0; i < 6; ++i) zero = zero && m.inertia[b][i] == 0.0f;
      if (!zero) return true;
    }
  return false;
}
template <class T>
static int check_topology_t(const MoccaModel& m, const char* name, std::string& err) {
  if (m.n_bodies != T::NB || m.n_joints != T::NJ || m.n_geoms > T::NG || m.n_slots > T::NSLOT || m.n_closures != T::NCLOS) {
    err = std::string("model blob sizes differ from the compiled topology (") + name + ")";
    return MOCCA_E_TOPOLOGY;
  }
  for (int b = 0; b < T::NB; ++b)
    if (m.parent[b] != T::parent(b) || m.anc_mask[b] != T::anc_mask(b)) {
      err = std::string("model blob tree differs from the compiled topology (") + name + ")";
      return MOCCA_E_TOPOLOGY;
    }
  if (massive_intermediates<T>(m)) {   // the ABA inward pass of T does not read the link inertia of a level that holds only such links
    err = std::string("model blob gives mass to a link the compiled topology treats as massless (") + name + ")";   // (check_topology picks the
    return MOCCA_E_TOPOLOGY;                                                                                      // ...Massive instance first)
  }
  if (m.n_pairs > 0 && 6 * T::NG > GP_FLOATS) {
    err = std::string("this topology's geom points overlap the contact records: blobs with self-collision pairs are not supported (") + name + ")";
    return MOCCA_E_ARG;
  }
  if (m.max_rows > mocca_r64_max_rows() || m.max_contacts > mocca_r64_max_contacts() || m.max_rows < 1 + 3 * T::NCLOS + (T::NCLOS > 0 && m.planar ? 3 : 0) ||
      m.n_pairs > MOCCA_MAX_PAIRS || m.n_feet != T::NFEET || m.n_ctrl > MOCCA_MAX_CTRL || m.n_ordered > MOCCA_MAX_CTRL) {
    err = "model blob caps exceed the kernel's (max_rows <= 64 and >= 1 + the closure / planar rows, max_contacts <= 20 -- beyond 48 / 12 the 64-row instance runs --, n_feet as compiled)";
    return MOCCA_E_ARG;
  }
  return MOCCA_OK;
}
static int check_topology(const MoccaModel& m, int task_id, int* topo, std::string& err) {
  // A blob that gives mass or inertia to the intermediate links of multi-hinge joints (PyBullet's importer may: pybullet_dump.py) runs on
  // the instance of the same tree that reads every link's inertia in the ABA inward pass.
  if (task_id == MOCCA_TASK_CASSIE) {
    if (m.n_bodies == TopoCassie::NB && massive_intermediates<TopoCassie>(m)) {
      *topo = TOPO_CASSIE_MASSIVE; return check_topology_t<TopoCassieMassive>(m, "TopoCassieMassive", err);
    }
    *topo = TOPO_CASSIE; return check_topology_t<TopoCassie>(m, "TopoCassie", err);
  }
  if (task_id == MOCCA_TASK_WALKER3D_CUSTOM && m.n_bodies == TopoWalker2D::NB) {
    *topo = TOPO_WALKER2D; return check_topology_t<TopoWalker2D>(m, "TopoWalker2D", err);
  }
  if (task_id == MOCCA_TASK_WALKER3D_CUSTOM && m.n_bodies == TopoCrab2D::NB) {
    *topo = TOPO_CRAB2D; return check_topology_t<TopoCrab2D>(m, "TopoCrab2D", err);
  }
  if (task_id == MOCCA_TASK_WALKER3D_PLANNER && m.n_bodies != TopoWalker3D::NB) {
    err = "the planner task runs on the Walker3D tree (Walker3DPlannerEnv, MikePlannerEnv)";
    return MOCCA_E_TOPOLOGY;
  }
  if ((task_id == MOCCA_TASK_WALKER3D_CUSTOM || task_id == MOCCA_TASK_WALKER3D_STEPPER) && m.n_bodies == TopoLaikago::NB) {
    *topo = TOPO_LAIKAGO; return check_topology_t<TopoLaikago>(m, "TopoLaikago", err);   // LaikagoCustomEnv / LaikagoStepperEnv
  }
  if (m.n_bodies == TopoWalker3D::NB && massive_intermediates<TopoWalker3D>(m)) {
    *topo = TOPO_WALKER3D_MASSIVE; return check_topology_t<TopoWalker3DMassive>(m, "TopoWalker3DMassive", err);
  }
  *topo = TOPO_WALKER3D;
  return check_topology_t<TopoWalker3D>(m, "TopoWalker3D", err);
}

// A blob whose caps fit 32 rows / 10 contacts on a tree without loop closures runs the compact instance of the step kernel (less LDS per
// env: more resident waves).  Same arithmetic, same order: which instance runs is not observable in the results.
static bool compact_ok(const MoccaModel& m, int topo) {
  return topo != TOPO_CASSIE && topo != TOPO_CASSIE_MASSIVE && m.n_closures == 0 && m.max_rows <= mocca_r32_max_rows() &&
         m.max_contacts <= mocca_r32_max_contacts() && mocca_r32_args_sizeof() == sizeof(StepArgs);
}
// ... and one whose caps exceed the 48-row instance's (48 rows / 12 contacts) runs the 64-row accuracy instance
// ... or that asks for Bullet's alternating sweep direction of the non-contact rows (compiled into that instance only: mocca_device.h ALT_SWEEPS)
static bool wide_needed(const MoccaModel& m) { return m.max_rows > MAXR || m.max_contacts > MAXC || m.sweep_alternate != 0; }
enum { INST_FULL = 0, INST_COMPACT = 1, INST_WIDE = 2 };
template <class T, int TASK> struct LaunchStep {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_step_kernel<T, TASK>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct LaunchReset {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_reset_kernel<T, TASK>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct LaunchObserve {
  static void run(int n, hipStream_t s, StepArgs a) { hipLaunchKernelGGL((mocca_observe_kernel<T, TASK>), dim3(n), dim3(64), 0, s, a); }
};
template <class T, int TASK> struct KernelInfo {
  static void run(hipFuncAttributes* fa, int* nb, hipError_t* e) {
    *e = hipFuncGetAttributes(fa, (const void*)mocca_step_kernel<T, TASK>);
    if (*e == hipSuccess) *e = hipOccupancyMaxActiveBlocksPerMultiprocessor(nb, mocca_step_kernel<T, TASK>, 64, 0);
  }
};

// The handle's buffers live on h->device: launches and copies are issued with that device current, whatever the
// caller's current device is (restored on return).  hipGetDevice / hipSetDevice are thread-local bookkeeping.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// which instance of the step kernel a handle's mocca_step launches (reset / observe / task-step kernels exist once)
static int step_instance(const mocca_ctx* h) {
  if (h->wide || h->force_full == 2) return INST_WIDE;
  if (h->compact && !h->force_full) return INST_COMPACT;
  return INST_FULL;
}

extern "C" {

int mocca_abi_version(void) { return MOCCA_ABI_VERSION; }
size_t mocca_model_sizeof(void) { return sizeof(MoccaModel); }

const char* mocca_last_error(mocca_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int mocca_create(const void* model_blob, size_t nbytes, int task_id, int n_envs, int device, mocca_handle* out) {
  if (!out) return MOCCA_E_ARG;
  *out = nullptr;
  if (!model_blob || nbytes != sizeof(MoccaModel)) { g_err = "model blob has the wrong size"; return MOCCA_E_ARG; }
  if (n_envs <= 0) { g_err = "n_envs must be positive"; return MOCCA_E_ARG; }
  if (task_id != MOCCA_TASK_WALKER3D_CUSTOM && task_id != MOCCA_TASK_WALKER3D_STEPPER && task_id != MOCCA_TASK_CASSIE &&
      task_id != MOCCA_TASK_WALKER3D_PLANNER) {
    g_err = "unknown task id"; return MOCCA_E_ARG;
  }
  mocca_ctx* h = new (std::nothrow) mocca_ctx();
  if (!h) return MOCCA_E_ARG;
  std::memcpy(&h->model, model_blob, sizeof(MoccaModel));
  if (h->model.magic != MOCCA_MODEL_MAGIC || h->model.version != MOCCA_MODEL_VERSION) {
    g_err = "bad model blob magic/version"; delete h; return MOCCA_E_ARG;
  }
  int rc = check_topology(h->model, task_id, &h->topo, g_err);
  if (rc != MOCCA_OK) { delete h; return rc; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_err = "no such HIP device"; delete h; return MOCCA_E_NODEVICE;
  }
  h->task_id = task_id; h->n_envs = n_envs; h->device = device;
  h->compact = compact_ok(h->model, h->topo);
  h->wide = wide_needed(h->model);
  if (h->wide && mocca_r64_args_sizeof() != sizeof(StepArgs)) { g_err = "the 64-row kernel instance was built from another StepArgs"; delete h; return MOCCA_E_ARG; }
  // pace priorities make the waves of a SIMD finish together -- right when all of a batch is resident (or whole generations are); the compact
  // instance exists for batches beyond one generation, whose partial last generation wants its slots refilled one by one: measured 1.5 - 2.5 %
  // slower with the pace than with the row-count priorities (8192 / 16384 envs), so its default is off
  if (h->compact) h->pace = 0;
  h->obs_dim = task_id == MOCCA_TASK_CASSIE
                   ? (h->model.cassie_mode == MOCCA_CASSIE_PLAIN ? 6 + 2 * h->model.n_ordered + 2 : 12 + 2 * h->model.n_ordered + 2)  // env_cassie.py:344-346 / :633
                   : 6 + 2 * h->model.n_joints + h->model.n_feet + (task_id == MOCCA_TASK_WALKER3D_STEPPER ? 5 * (h->model.lookbehind + 2) : 2);
  if (task_id == MOCCA_TASK_CASSIE && (h->model.cassie_mode < MOCCA_CASSIE_PLAIN || h->model.cassie_mode > MOCCA_CASSIE_PHASE_MIRROR ||
                                       (h->model.cassie_mode != MOCCA_CASSIE_PLAIN && h->model.n_ordered != 14))) {
    g_err = "Cassie blob: unknown cassie_mode (the mocap / phase envs need the 14 ordered joints)"; delete h; return MOCCA_E_ARG;
  }
  if (task_id == MOCCA_TASK_WALKER3D_STEPPER &&
      (h->model.n_planks < 1 || h->model.n_planks > MOCCA_MAX_PLANKS || h->model.lookbehind < 1 || h->model.lookbehind > 2)) {
    g_err = "Stepper blob: n_planks must be 1..4 and lookbehind 1 or 2"; delete h; return MOCCA_E_ARG;
  }
  auto fail = [&](const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    mocca_destroy(h);
    return MOCCA_E_HIP;
  };
  hipError_t e;
  DeviceGuard guard(device);  // the caller's current device is restored on return
  if ((e = guard.err) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipMalloc(&h->d_model, sizeof(MoccaModel))) != hipSuccess) return fail("hipMalloc(model)", e);
  if ((e = hipMemcpy(h->d_model, &h->model, sizeof(MoccaModel), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(model)", e);
  const size_t dyn_b = (size_t)n_envs * DYN_STRIDE * sizeof(float), task_b = (size_t)n_envs * MOCCA_TASK_WORDS * 4;
  const size_t ter_b = (size_t)n_envs * TERRAIN_STRIDE * sizeof(float);
  if ((e = hipMalloc(&h->d_dyn, dyn_b)) != hipSuccess) return fail("hipMalloc(state)", e);
  if ((e = hipMalloc(&h->d_task, task_b)) != hipSuccess) return fail("hipMalloc(task)", e);
  if ((e = hipMalloc(&h->d_terrain, ter_b)) != hipSuccess) return fail("hipMalloc(terrain)", e);
  if ((e = hipMalloc(&h->d_pace_acc, sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc(pace samples)", e);
  if ((e = hipMemset(h->d_pace_acc, 0, sizeof(unsigned long long))) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMemset(h->d_dyn, 0, dyn_b)) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMemset(h->d_terrain, 0, ter_b)) != hipSuccess) return fail("hipMemset", e);
  // task records: episode = -1 so the first reset is episode 0; applied_gain = 1
  {
    uint32_t* tmp = new uint32_t[(size_t)n_envs * MOCCA_TASK_WORDS]();
    const float one = 1.0f;
    uint32_t one_bits; std::memcpy(&one_bits, &one, 4);
    for (int i = 0; i < n_envs; ++i) {
      tmp[(size_t)i * MOCCA_TASK_WORDS + T_EPISODE] = (uint32_t)-1;
      tmp[(size_t)i * MOCCA_TASK_WORDS + T_GAIN] = one_bits;
    }
    e = hipMemcpy(h->d_task, tmp, task_b, hipMemcpyHostToDevice);
    delete[] tmp;
    if (e != hipSuccess) return fail("hipMemcpy(task)", e);
  }
  // identity quaternion so an un-reset env is still a valid state
  {
    float* tmp = new float[(size_t)n_envs * DYN_STRIDE]();
    for (int i = 0; i < n_envs; ++i) tmp[(size_t)i * DYN_STRIDE + 6] = 1.0f;
    e = hipMemcpy(h->d_dyn, tmp, dyn_b, hipMemcpyHostToDevice);
    delete[] tmp;
    if (e != hipSuccess) return fail("hipMemcpy(state)", e);
  }
  *out = h;
  return MOCCA_OK;
}

int mocca_destroy(mocca_handle h) {
  if (!h) return MOCCA_OK;
  if (h->d_model) (void)hipFree(h->d_model);
  if (h->d_dyn) (void)hipFree(h->d_dyn);
  if (h->d_task) (void)hipFree(h->d_task);
  if (h->d_terrain) (void)hipFree(h->d_terrain);
  if (h->d_traj) (void)hipFree(h->d_traj);
  if (h->d_hf) (void)hipFree(h->d_hf);
  if (h->d_order) (void)hipFree(h->d_order);
  if (h->d_pace_acc) (void)hipFree(h->d_pace_acc);
  if (h->d_ep_ret) (void)hipFree(h->d_ep_ret);
  for (float* p : h->d_pvec) if (p) (void)hipFree(p);
  delete h;
  return MOCCA_OK;
}

int mocca_n_envs(mocca_handle h) { return h ? h->n_envs : MOCCA_E_ARG; }
int mocca_obs_dim(mocca_handle h) { return h ? h->obs_dim : MOCCA_E_ARG; }
int mocca_act_dim(mocca_handle h) {
  if (!h) return MOCCA_E_ARG;
  return h->task_id == MOCCA_TASK_CASSIE ? h->model.n_ctrl - 2 : h->model.n_joints;
}
int mocca_state_dim(mocca_handle h) { return h ? MOCCA_STATE_DIM(h->model.n_joints, h->model.n_slots) : MOCCA_E_ARG; }

static StepArgs make_args(mocca_handle h) {
  StepArgs a{};
  a.model = h->d_model; a.dyn = h->d_dyn; a.task = h->d_task; a.terrain = h->d_terrain;
  a.n_envs = h->n_envs; a.obs_dim = h->obs_dim;
  a.auto_reset = h->auto_reset; a.eval_mode = h->eval_mode; a.random_pose = h->random_pose; a.curriculum = h->curriculum;
  a.host_retarget = h->host_retarget; a.env_offset = h->env_offset; a.random_reward = h->random_reward;
  a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
  a.curriculum_v = h->pvec_on[0] ? h->d_pvec[0] : nullptr;
  a.eval_mode_v = h->pvec_on[1] ? h->d_pvec[1] : nullptr;
  a.gain_v = h->pvec_on[2] ? h->d_pvec[2] : nullptr;
  a.gain = h->gain;
  a.dbg = h->dbg;
  a.prio = h->prio;
  a.traj = h->d_traj; a.traj_n = h->traj_n; a.traj_tmax = h->traj_tmax; a.traj_cstep = h->traj_cstep;
  a.final_obs = h->final_obs;
  a.persist_warm = h->persist_warm;
  a.pace = h->pace;
  a.pace_acc = h->d_pace_acc;
  a.ep_ret = h->d_ep_ret; a.ep_masks = h->ep_masks; a.ep_bad = h->ep_bad; a.ep_totals = h->ep_totals;   // (ep_rec / ep_serial: mocca_step only)
  a.hf = h->d_hf; a.hf_rows = h->hf_rows; a.hf_cols = h->hf_cols; a.hf_scale = h->hf_scale;
  return a;
}
// A scalar MOCCA_PARAM_APPLIED_GAIN is written into the task records (word T_GAIN, what apply_action reads) by the NEXT call that takes
// a stream, on that stream: ordered against the caller's in-flight steps, which also write the word (store_task).
static int flush_pending(mocca_handle h, hipStream_t s) {
  if (!h->gain_pending) return MOCCA_OK;
  hipLaunchKernelGGL(set_task_word_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, s, h->d_task, (int)T_GAIN, (const float*)nullptr, h->gain, 1, h->n_envs);
  HIP_TRY(h, hipGetLastError());
  h->gain_pending = false;
  return MOCCA_OK;
}
// the mocap / phase envs read their targets, reset poses and reward references from the motion table
static int need_trajectory(mocca_handle h) {
  if (h->task_id == MOCCA_TASK_CASSIE && h->model.cassie_mode != MOCCA_CASSIE_PLAIN && !h->d_traj) {
    h->err = "this Cassie blob (cassie_mode != 0) needs mocca_set_trajectory before reset / step / observe";
    return MOCCA_E_ARG;
  }
  if (h->task_id == MOCCA_TASK_WALKER3D_PLANNER && !h->d_hf) {   // the planner envs stand on the height field
    h->err = "the planner task needs mocca_set_heightfield before reset / step / observe";
    return MOCCA_E_ARG;
  }
  return MOCCA_OK;
}

int mocca_reset(mocca_handle h, const uint8_t* mask_dev, uint64_t seed, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return MOCCA_E_ARG;
  if (need_trajectory(h) != MOCCA_OK) return MOCCA_E_ARG;
  h->seed = seed;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.mask = mask_dev; a.obs = obs_dev;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = flush_pending(h, s)) return rc;
  if (h->tape) {  // recorded draws instead of Philox (mocca_set_draw_tape)
    a.tape = h->tape; a.tape_n = h->tape_n;
    launch_taped_reset(h->topo, h->task_id, h->n_envs, s, a);
  } else {
    dispatch<LaunchReset>(h->topo, h->task_id, h->n_envs, s, a);
  }
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_step(mocca_handle h, const float* act_dev, float* obs_dev, float* rew_dev, uint8_t* done_dev, int32_t* info_dev,
               void* stream) {
  if (!h || !act_dev || !obs_dev || !rew_dev || !done_dev) return MOCCA_E_ARG;
  if (need_trajectory(h) != MOCCA_OK) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.act = act_dev; a.obs = obs_dev; a.rew = rew_dev; a.done = done_dev; a.info = info_dev;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = flush_pending(h, s)) return rc;
  // mocca_step never allocates and never synchronises, and the pace calibrates itself on the device: the launch can be captured in a
  // hipGraph.  Two optional features keep HOST state per launch that a capture bakes into the kernel arguments: the episode-record ring
  // (slot and serial below: a replayed launch keeps writing one slot under one serial -- read ep_masks / ep_totals instead) and the
  // re-sort schedule of MOCCA_PARAM_ORDER_EVERY > 0.
  if (h->d_ep_ret && h->ep_rec) {
    a.ep_rec = h->ep_rec + (size_t)(h->ep_serial % (uint32_t)h->ep_slots) * h->ep_stride;
    a.ep_serial = h->ep_serial;
    if (++h->ep_serial == 0u) h->ep_serial = 1u;
  }
  if (h->order_every > 0 && h->d_order) {   // heaviest envs first: the permutation is rebuilt on the caller's stream, ahead of the step that reads it
    if (h->order_age >= h->order_every) {
      hipLaunchKernelGGL(order_by_rows_kernel, dim3(1), dim3(1024), 0, s, (const uint32_t*)h->d_task, h->d_order, h->n_envs);
      HIP_TRY(h, hipGetLastError());
      h->order_age = 0;
    }
    ++h->order_age;
    a.order = h->d_order;
  }
  switch (step_instance(h)) {
    case INST_COMPACT: mocca_r32_launch_step(h->topo, h->task_id, h->n_envs, s, &a); break;
    case INST_WIDE: mocca_r64_launch_step(h->topo, h->task_id, h->n_envs, s, &a); break;
    default: dispatch<LaunchStep>(h->topo, h->task_id, h->n_envs, s, a);
  }
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_task_step(mocca_handle h, const float* act_dev, const int32_t* touch_dev, const int32_t* target_dev, const int32_t* body_dev,
                    float* obs_dev, float* rew_dev, uint8_t* done_dev, int32_t* info_dev, void* stream) {
  if (!h || !act_dev || !obs_dev || !rew_dev || !done_dev) return MOCCA_E_ARG;
  if (!touch_dev && h->task_id != MOCCA_TASK_CASSIE) { h->err = "mocca_task_step needs the foot contact flags"; return MOCCA_E_ARG; }
  if (need_trajectory(h) != MOCCA_OK) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.act = act_dev; a.obs = obs_dev; a.rew = rew_dev; a.done = done_dev; a.info = info_dev;
  a.inj_touch = touch_dev; a.inj_target = target_dev; a.inj_body = body_dev;
  a.tape = h->tape; a.tape_n = h->tape_n;
  a.dbg = nullptr;
  if (int rc = flush_pending(h, (hipStream_t)stream)) return rc;
  launch_task_step(h->topo, h->task_id, h->n_envs, (hipStream_t)stream, a);
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_set_draw_tape(mocca_handle h, const float* tape_dev, int n_per_env) {
  if (!h || (tape_dev && n_per_env <= 0)) return MOCCA_E_ARG;
  h->tape = tape_dev; h->tape_n = tape_dev ? n_per_env : 0;
  return MOCCA_OK;
}

int mocca_set_debug_buffer(mocca_handle h, int32_t* dbg_dev) {
  if (!h) return MOCCA_E_ARG;
  h->dbg = dbg_dev;
  return MOCCA_OK;
}

int mocca_set_terminal_obs_buffer(mocca_handle h, float* final_obs_dev) {
  if (!h) return MOCCA_E_ARG;
  h->final_obs = final_obs_dev;
  return MOCCA_OK;
}

int mocca_set_episode_stats(mocca_handle h, float* masks_dev, float* bad_masks_dev, float* totals_dev, void* records, int n_slots,
                            size_t slot_stride_bytes) {
  if (!h) return MOCCA_E_ARG;
  const bool on = masks_dev || bad_masks_dev || totals_dev || records;
  if (records && (n_slots < 1 || slot_stride_bytes < (size_t)h->n_envs * sizeof(mocca_episode_rec) || ((uintptr_t)records & 15u) || (slot_stride_bytes & 15u))) {
    h->err = "episode records: n_slots >= 1 slots of n_envs 16-byte records, 16-byte aligned, slot_stride_bytes >= 16 n_envs"; return MOCCA_E_ARG;
  }
  DeviceGuard guard(h->device);
  if (on && !h->d_ep_ret) {
    HIP_TRY(h, hipMalloc(&h->d_ep_ret, (size_t)h->n_envs * sizeof(float)));
    HIP_TRY(h, hipMemset(h->d_ep_ret, 0, (size_t)h->n_envs * sizeof(float)));
  } else if (!on && h->d_ep_ret) {
    HIP_TRY(h, hipDeviceSynchronize());   // a launch in flight may still add to it
    HIP_TRY(h, hipFree(h->d_ep_ret));
    h->d_ep_ret = nullptr;
  }
  h->ep_masks = masks_dev; h->ep_bad = bad_masks_dev; h->ep_totals = totals_dev;
  h->ep_rec = (char*)records; h->ep_slots = records ? n_slots : 0; h->ep_stride = records ? slot_stride_bytes : 0;
  return MOCCA_OK;
}
uint32_t mocca_episode_serial(mocca_handle h) { return h ? h->ep_serial : 0u; }

int mocca_set_seed(mocca_handle h, uint64_t seed) {
  if (!h) return MOCCA_E_ARG;
  h->seed = seed;
  return MOCCA_OK;
}

int mocca_is_diagnostic_build(void) {
#if defined(MOCCA_SKIP_COLLIDE) || defined(MOCCA_SKIP_ABA) || defined(MOCCA_SKIP_SOLVE) || defined(MOCCA_DUMMY_VALU) || defined(MOCCA_STAMPS) || \
    defined(MOCCA_NO_TWO_PATHS) || defined(MOCCA_ABL_PAIRLOAD) || defined(MOCCA_ABL_NOPASS2) || defined(MOCCA_ABL_NOHITS) || MOCCA_LDS_PAD > 0
  return 1;
#else
  return 0;
#endif
}

int mocca_set_trajectory(mocca_handle h, const float* table_host, int n_frames, double max_time, double control_step) {
  if (!h) return MOCCA_E_ARG;
  if (!table_host || n_frames <= 0 || !(max_time > 0.0) || !(control_step > 0.0)) { h->err = "mocca_set_trajectory: empty table or non-positive times"; return MOCCA_E_ARG; }
  DeviceGuard guard(h->device);
  const size_t bytes = (size_t)n_frames * MOCCA_TRAJ_STRIDE * sizeof(float);
  float* d = nullptr;
  HIP_TRY(h, hipMalloc(&d, bytes));
  hipError_t e = hipMemcpy(d, table_host, bytes, hipMemcpyHostToDevice);   // synchronous: no kernel in flight still reads the old table
  if (e != hipSuccess) { (void)hipFree(d); h->err = std::string("hipMemcpy(trajectory): ") + hipGetErrorString(e); return MOCCA_E_HIP; }
  if (h->d_traj) { (void)hipDeviceSynchronize(); (void)hipFree(h->d_traj); }
  h->d_traj = d; h->traj_n = n_frames; h->traj_tmax = max_time; h->traj_cstep = control_step;
  return MOCCA_OK;
}

int mocca_set_heightfield(mocca_handle h, const float* heights_host, int rows, int cols, double scale) {
  if (!h) return MOCCA_E_ARG;
  if (!heights_host || rows < 2 || cols < 2 || !(scale > 0.0)) { h->err = "mocca_set_heightfield: needs at least 2 x 2 heights and a positive scale"; return MOCCA_E_ARG; }
  // Search window of every terrain contact slot: a sphere of reach rho = radius + margin around a centre that is at most half a cell from
  // its nearest grid point touches only cells within W = ceil(rho scale + 1/2) of that point (1e-6: a reach of exactly half a cell is W = 1).
  // W - 1 travels in bits 30..31 of the slot record; a grid so fine that a sphere spans more than 4 cells each way is refused.
  int wmax = 1;
  uint32_t wbits[MOCCA_MAX_SLOTS];
  for (int sl = 0; sl < h->model.n_slots; ++sl) {
    uint32_t ids; std::memcpy(&ids, &h->model.slot_tab[sl][2], 4);
    const double reach = (double)h->model.slot_tab[sl][0] + (double)((ids >> 17) & 0xFFu) / 8192.0;
    int w = (int)std::ceil(reach * scale + 0.5 - 1e-6);
    if (w < 1) w = 1;
    if (w > 4 && ((ids >> 25) & 1u)) { h->err = "mocca_set_heightfield: the grid is too fine for this robot (a contact sphere would span more than 4 cells each way)"; return MOCCA_E_ARG; }
    if (w > 4) w = 4;
    wbits[sl] = (ids & 0x3FFFFFFFu) | ((uint32_t)(w - 1) << 30);
    if (((ids >> 25) & 1u) && w > wmax) wmax = w;
  }
  DeviceGuard guard(h->device);
  // the heights, followed by one max-pooled copy per window 2 .. wmax (copy k: the highest point within k + 1 cells of each grid point): what a
  // wide sphere's search is pruned by with one load
  const size_t cells = (size_t)rows * cols, bytes = cells * wmax * sizeof(float);
  float* host = new (std::nothrow) float[cells * wmax];
  if (!host) { h->err = "mocca_set_heightfield: out of host memory"; return MOCCA_E_ARG; }
  std::memcpy(host, heights_host, cells * sizeof(float));
  for (int w = 2; w <= wmax; ++w) {
    float* out = host + cells * (w - 1);
    for (int j = 0; j < rows; ++j)
      for (int i = 0; i < cols; ++i) {
        float m = -1e30f;
        for (int jj = (j - w < 0 ? 0 : j - w); jj <= (j + w > rows - 1 ? rows - 1 : j + w); ++jj)
          for (int ii = (i - w < 0 ? 0 : i - w); ii <= (i + w > cols - 1 ? cols - 1 : i + w); ++ii)
            m = heights_host[(size_t)jj * cols + ii] > m ? heights_host[(size_t)jj * cols + ii] : m;
        out[(size_t)j * cols + i] = m;
      }
  }
  float* d = nullptr;
  hipError_t e = hipMalloc(&d, bytes);
  if (e == hipSuccess) e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);   // synchronous: no kernel in flight still reads the old grid
  delete[] host;
  if (e != hipSuccess) { if (d) (void)hipFree(d); h->err = std::string("mocca_set_heightfield: ") + hipGetErrorString(e); return MOCCA_E_HIP; }
  (void)hipDeviceSynchronize();   // ... nor the old slot records
  // the window bits go into a COPY of the model; the handle's host image takes them only once the device has them (a failed upload leaves
  // host and device records as they were: "refused and leaves the handle intact")
  MoccaModel* next = new (std::nothrow) MoccaModel(h->model);
  if (!next) { (void)hipFree(d); h->err = "mocca_set_heightfield: out of host memory"; return MOCCA_E_ARG; }
  for (int sl = 0; sl < next->n_slots; ++sl) std::memcpy(&next->slot_tab[sl][2], &wbits[sl], 4);
  e = hipMemcpy(h->d_model, next, sizeof(MoccaModel), hipMemcpyHostToDevice);
  if (e != hipSuccess) { delete next; (void)hipFree(d); h->err = std::string("hipMemcpy(model): ") + hipGetErrorString(e); return MOCCA_E_HIP; }
  h->model = *next;
  delete next;
  if (h->d_hf) (void)hipFree(h->d_hf);
  h->d_hf = d; h->hf_rows = rows; h->hf_cols = cols; h->hf_scale = (float)scale;
  return MOCCA_OK;
}

int mocca_observe(mocca_handle h, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return MOCCA_E_ARG;
  if (need_trajectory(h) != MOCCA_OK) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.obs = obs_dev;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = flush_pending(h, s)) return rc;
  dispatch<LaunchObserve>(h->topo, h->task_id, h->n_envs, s, a);
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_get_state(mocca_handle h, float* state_dev, void* stream) {
  if (!h || !state_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  const size_t w = (size_t)mocca_state_dim(h) * sizeof(float);
  HIP_TRY(h, hipMemcpy2DAsync(state_dev, w, h->d_dyn, DYN_STRIDE * sizeof(float), w, h->n_envs, hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_set_state(mocca_handle h, const float* state_dev, void* stream) {
  if (!h || !state_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  const size_t w = (size_t)mocca_state_dim(h) * sizeof(float);
  HIP_TRY(h, hipMemcpy2DAsync(h->d_dyn, DYN_STRIDE * sizeof(float), state_dev, w, w, h->n_envs, hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_get_task(mocca_handle h, uint32_t* task_dev, void* stream) {
  if (!h || !task_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  if (int rc = flush_pending(h, (hipStream_t)stream)) return rc;
  HIP_TRY(h, hipMemcpyAsync(task_dev, h->d_task, (size_t)h->n_envs * MOCCA_TASK_WORDS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_set_task(mocca_handle h, const uint32_t* task_dev, void* stream) {
  if (!h || !task_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  // a restored snapshot wins over an earlier scalar applied_gain: every env's T_GAIN is the snapshot's; the handle's own copy (what a
  // Custom env's reset writes, robot.applied_gain persists across resets) keeps the value of the last mocca_set_param
  h->gain_pending = false;
  HIP_TRY(h, hipMemcpyAsync(h->d_task, task_dev, (size_t)h->n_envs * MOCCA_TASK_WORDS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_get_terrain(mocca_handle h, float* terrain_dev, void* stream) {
  if (!h || !terrain_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipMemcpyAsync(terrain_dev, h->d_terrain, (size_t)h->n_envs * TERRAIN_STRIDE * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_set_terrain(mocca_handle h, const float* terrain_dev, void* stream) {
  if (!h || !terrain_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipMemcpyAsync(h->d_terrain, terrain_dev, (size_t)h->n_envs * TERRAIN_STRIDE * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MOCCA_OK;
}

int mocca_set_param(mocca_handle h, int param_id, double value) {
  if (!h) return MOCCA_E_ARG;
  switch (param_id) {
    case MOCCA_PARAM_AUTO_RESET: h->auto_reset = value != 0; break;
    case MOCCA_PARAM_EVAL_MODE: h->eval_mode = value != 0; h->pvec_on[1] = false; break;
    case MOCCA_PARAM_CURRICULUM: h->curriculum = (int)value < 0 ? 0 : ((int)value > 9 ? 9 : (int)value); h->pvec_on[0] = false; break;
    case MOCCA_PARAM_APPLIED_GAIN:  // acts on the next apply_action: the task records carry the value the kernel uses; this call has no
      h->gain = (float)value; h->pvec_on[2] = false; h->gain_pending = true;   // stream, so the write is enqueued by the next call that has
      break;                                                                    // one (flush_pending), ordered on the caller's stream
    case MOCCA_PARAM_RANDOM_POSE: h->random_pose = value != 0; break;
    case MOCCA_PARAM_HOST_RETARGET: h->host_retarget = value != 0; break;
    case MOCCA_PARAM_SEED: h->seed = (uint64_t)value; break;
    case MOCCA_PARAM_ENV_OFFSET: h->env_offset = (int)value; break;
    case MOCCA_PARAM_ISSUE_PRIORITY:
      if (value < 0 || value >= 262144) { h->err = "MOCCA_PARAM_ISSUE_PRIORITY is t1 + 64 t2 + 4096 t3 with each threshold in 0..63"; return MOCCA_E_ARG; }
      h->prio = (int)value;
      break;
    case MOCCA_PARAM_RANDOM_REWARD:
      if (value != 0 && value != 1 && value != 2) { h->err = "MOCCA_PARAM_RANDOM_REWARD is 0, 1 or 2"; return MOCCA_E_ARG; }
      h->random_reward = (int)value; break;
    case MOCCA_PARAM_PERSIST_IMPULSES: h->persist_warm = value != 0; break;
    case MOCCA_PARAM_PACE_TICKS:
      if (value < -64 || value > 1e9) { h->err = "MOCCA_PARAM_PACE_TICKS is a tick count > 0, 0 (off) or -k (self-calibrating, k / 16 of the mean wave time, k <= 64)"; return MOCCA_E_ARG; }
      h->pace = (int)value; break;
    case MOCCA_PARAM_ORDER_EVERY:
      if (value < 0 || value > 1e6) { h->err = "MOCCA_PARAM_ORDER_EVERY is a step count >= 0"; return MOCCA_E_ARG; }
      h->order_every = (int)value; h->order_age = h->order_every;
      if (h->order_every > 0 && !h->d_order) {   // (allocated here, not in mocca_step)
        DeviceGuard guard(h->device);
        HIP_TRY(h, hipMalloc(&h->d_order, (size_t)h->n_envs * sizeof(int32_t)));
      }
      break;
    case MOCCA_PARAM_KERNEL_VARIANT:
      if (value != 0 && value != 1 && value != 2) { h->err = "MOCCA_PARAM_KERNEL_VARIANT is 0 (automatic), 1 (force the 48-row instance) or 2 (force the 64-row instance)"; return MOCCA_E_ARG; }
      if (value == 1 && h->wide) { h->err = "MOCCA_PARAM_KERNEL_VARIANT = 1: this blob needs the 64-row instance (caps beyond 48 rows / 12 contacts, or sweep_alternate)"; return MOCCA_E_ARG; }
      h->force_full = (int)value; break;
    default: h->err = "unknown parameter id"; return MOCCA_E_ARG;
  }
  return MOCCA_OK;
}

int mocca_set_param_v(mocca_handle h, int param_id, const float* values_dev, int broadcast, void* stream) {
  if (!h || !values_dev) return MOCCA_E_ARG;
  const int slot = param_id == MOCCA_PARAM_CURRICULUM ? 0 : param_id == MOCCA_PARAM_EVAL_MODE ? 1 : param_id == MOCCA_PARAM_APPLIED_GAIN ? 2 : -1;
  if (slot < 0) { h->err = "this parameter has no per-env form"; return MOCCA_E_ARG; }
  DeviceGuard guard(h->device);
  if (!h->d_pvec[slot]) HIP_TRY(h, hipMalloc(&h->d_pvec[slot], (size_t)h->n_envs * sizeof(float)));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(copy_param_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, s, h->d_pvec[slot], values_dev, broadcast != 0, h->n_envs);
  HIP_TRY(h, hipGetLastError());
  if (slot == 2) {  // applied_gain acts at once (robots.py:33)
    // the per-env values supersede a scalar mocca_set_param(APPLIED_GAIN) that was not flushed yet: left pending, the next call with a
    // stream would overwrite every env's word with the stale scalar (call order must win, as it did when the scalar write was synchronous)
    h->gain_pending = false;
    hipLaunchKernelGGL(set_task_word_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, s, h->d_task, (int)T_GAIN, (const float*)h->d_pvec[slot], 0.0f, 0, h->n_envs);
    HIP_TRY(h, hipGetLastError());
  }
  h->pvec_on[slot] = true;
  return MOCCA_OK;
}

#ifdef MOCCA_STAMPS
// diagnostic builds only: the raw s_memtime marks (STAMP_SLOTS per wave) of the most recent launch
int mocca_debug_stamps(unsigned long long* out, int n_waves) {
  if (n_waves > mocca::STAMP_WAVES) n_waves = mocca::STAMP_WAVES;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mocca::g_stamps), (size_t)n_waves * mocca::STAMP_SLOTS * 8) != hipSuccess) return MOCCA_E_HIP;
  return MOCCA_OK;
}
#endif

int mocca_kernel_info(mocca_handle h, int* vgprs, int* sgprs, int* lds_bytes, int* scratch_bytes, int* max_blocks_per_cu) {
  if (!h) return MOCCA_E_ARG;
  hipFuncAttributes fa;
  int nb = 0;
  hipError_t e = hipSuccess;
  switch (step_instance(h)) {
    case INST_COMPACT: mocca_r32_kernel_info(h->topo, h->task_id, &fa, &nb, &e); break;
    case INST_WIDE: mocca_r64_kernel_info(h->topo, h->task_id, &fa, &nb, &e); break;
    default: dispatch<KernelInfo>(h->topo, h->task_id, &fa, &nb, &e);
  }
  HIP_TRY(h, e);
  if (vgprs) *vgprs = fa.numRegs;
  if (sgprs) *sgprs = -1;   // hipFuncAttributes has no scalar-register field: -1 = not reported (the count is in the code object's metadata: build.py -v prints it)
  if (lds_bytes) *lds_bytes = (int)fa.sharedSizeBytes;
  if (scratch_bytes) *scratch_bytes = (int)fa.localSizeBytes;
  if (max_blocks_per_cu) *max_blocks_per_cu = nb;
  return MOCCA_OK;
}

}  // extern "C"
