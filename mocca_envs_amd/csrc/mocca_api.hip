// mocca_api.hip -- kernels' entry points and the C ABI of libmocca_hip.so (include/mocca.h).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see mocca_envs_amd/build.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mocca.h"
#include "mocca_device.h"

using namespace mocca;

#ifndef MOCCA_WAVES_PER_EU
#define MOCCA_WAVES_PER_EU 4
#endif

// --------------------------------------------------------------------------------------------
// kernels: one 64-lane workgroup (= one wavefront) per environment
// --------------------------------------------------------------------------------------------
template <class T, int TASK>
__global__ __launch_bounds__(64, MOCCA_WAVES_PER_EU) void mocca_step_kernel(StepArgs a) {
  __shared__ float L[L_TOTAL];
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= a.n_envs) return;
  ModelP M = (ModelP)a.model;
  float* st = a.dyn + (size_t)env * DYN_STRIDE;
  uint32_t* tk = a.task + (size_t)env * MOCCA_TASK_WORDS;
  float* ter = TASK == MOCCA_TASK_WALKER3D_STEPPER ? a.terrain + (size_t)env * TERRAIN_STRIDE : nullptr;
  float* obs = a.obs + (size_t)env * a.obs_dim;

  load_dyn(st, L, lane, T::NJ, T::NSLOT);
  // the lane's root->body path, packed 5 bits per step; the only lane-derived value kept across the substeps
  const unsigned long long ppk = T::path_packed(lane < T::NB ? lane : 0);
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    // ---- CassieEnv.step (env_cassie.py:433-479): 50 x { filter joint speeds, PD torques, one physics step }
    const int no = M->n_ordered, nctl = M->n_ctrl;
    float target = 0.0f;  // env_cassie.py:434-443: base angle (residual control) + action, 0 for the springs
    if (lane < nctl) target = M->ctrl_base[lane] + (lane < nctl - 2 ? a.act[(size_t)env * (nctl - 2) + lane] : 0.0f);
    if (lane < no) {
      L[L_JVEL + lane] = __uint_as_float(tk[T_JVEL + lane]);
      L[L_Q0 + lane] = L[L_Q + M->ordered_body[lane]];
    }
    if (lane < MOCCA_MAX_BODIES) L[L_TAU + lane] = 0.0f;
    if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
    wsync();
    const int nllc = M->n_llc;
#pragma unroll 1
    for (int it = 0; it < nllc; ++it) {
      ModelP Ms = M;
      int ln = lane;
      unsigned long long pk = ppk;
      asm volatile("" : "+s"(Ms), "+v"(ln), "+v"(pk));
      if (ln < no) {  // :451-453
        const float al = Ms->jvel_alpha;
        L[L_JVEL + ln] = (1.0f - al) * L[L_JVEL + ln] + al * L[L_QD + Ms->ordered_body[ln]];
      }
      wsync();
      if (ln < nctl) {  // pd_control :380-393 + torque clip :225-230
        const int b = Ms->ctrl_body[ln];
        const float perr = target - L[L_Q + b];
        float verr = -L[L_JVEL + Ms->ctrl_oidx[ln]];
        verr = verr < -5.0f ? -5.0f : (verr > 5.0f ? 5.0f : verr);
        const float tq = Ms->ctrl_kp[ln] * perr + Ms->ctrl_kd[ln] * verr, lim = Ms->torque_limit[b];
        L[L_TAU + b] = tq < -lim ? -lim : (tq > lim ? lim : tq);
      }
      wsync();
      substep<T, TASK>(Ms, L, ln, nullptr, 0, pk);
    }
    TaskRegs t;
    load_task(tk, t);
    t.istep += nllc;
    if (lane < no) {  // :467-468 finite-difference joint velocity over the control step
      const float jv = (L[L_Q + M->ordered_body[lane]] - L[L_Q0 + lane]) / M->control_dt;
      tk[T_JVEL + lane] = __float_as_uint(jv);
    }
    stage_joints<T>(M, L, lane);
    walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
    wsync();
    t.t += 1;
    bool fin;
    const float height = cassie_obs<T>(M, L, lane, t.initz, obs, &fin);
    const float old = t.linpot;
    t.linpot = cassie_potential(M, L);
    const float alive = height > M->alive_height ? 2.0f : -1.0f;  // compute_rewards :401-414
    if (!fin || alive < 0.0f) t.done = 1;
    const int timeout = t.t >= M->max_episode_steps;
    const int dflag = (t.done ? 1 : 0) | (timeout ? 2 : 0);
    if (lane == 0) {
      a.rew[env] = alive + (t.linpot - old);
      a.done[env] = (uint8_t)dflag;
      if (a.info) a.info[env] = 0;
    }
    if (a.auto_reset && dflag) {
      wsync();
      cassie_reset_env<T>(M, L, lane, t, obs);
      if (lane < MOCCA_MAX_CTRL) tk[T_JVEL + lane] = 0u;
    }
    wsync();
    store_dyn(st, L, lane, T::NJ, T::NSLOT);
    if (lane == 0) store_task(tk, t);
    return;
  }
  // apply_action, robots.py:31-40.  Only the two task words the physics needs are read before the substeps;
  // the rest of the task record is loaded after them so it does not occupy registers across the loop.
  {
    const float applied_gain = __uint_as_float(tk[T_GAIN]);
    if (lane < T::NJ) {
      const float act_raw = a.act[(size_t)env * T::NJ + lane];
      const float c = act_raw < -1.0f ? -1.0f : (act_raw > 1.0f ? 1.0f : act_raw);
      L[L_TAU + 1 + lane] = M->gain[lane + 1] * applied_gain * c;
    }
  }
  if (lane == 0) { L[L_TAU] = 0.0f; L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
  wsync();

  STAMP(28);  // kernel prologue done
  if constexpr (TASK == MOCCA_TASK_WALKER3D_STEPPER) stage_planks(M, L, lane, ter);
  ContactFlags fl = {0, 0, 0, 0};
  const int nsub = M->n_substeps;
  const int nsi0 = TASK == MOCCA_TASK_WALKER3D_STEPPER ? (int)tk[T_NSI] : 0;
#pragma unroll 1
  for (int s = 0; s < nsub; ++s) {
    // launder the model pointer: keeps LICM from hoisting dozens of loop-invariant model loads out of the
    // substep loop, where they would sit in registers (and spill to scratch) for the whole kernel
    ModelP Ms = M;
    int ln = lane;  // same for lane-derived offsets and predicates (recomputing them costs a few instructions)
    unsigned long long pk = ppk;  // laundered too: otherwise every (ppk >> 5k) & 31 and the addresses derived from it
    asm volatile("" : "+s"(Ms), "+v"(ln), "+v"(pk));  // are hoisted out of the loop and spilled
    fl = substep<T, TASK>(Ms, L, ln, ter, nsi0, pk);
  }
  STAMP(27);  // substeps done
  TaskRegs t;
  load_task(tk, t, T::NFEET > 2);
  // the raw (unclipped) action enters the energy penalty (env_locomotion.py:185-188); re-read it rather than
  // hold a register across the substeps
  const float act_raw = lane < T::NJ ? a.act[(size_t)env * T::NJ + lane] : 0.0f;

  // ---- calc_state + task logic on the post-step state
  {
    int lo = lane;  // laundered: the walk's lane-derived body index would otherwise be kept (spilled) from kernel entry
    asm volatile("" : "+v"(lo));
    stage_joints<T>(M, L, lo);
    walk_kinematics<T, false>(M, L, lo, ppk);
  }
  wsync();
  t.t += 1;
  constexpr int NBO = 6 + 2 * T::NJ + T::NFEET;
  float rew = 0.0f;
  int info = 0;
  if (TASK == MOCCA_TASK_WALKER3D_CUSTOM) {
    if (a.eval_mode) { t.wt[0] = t.prevx + 4.0f; t.wt[1] = 0.0f; t.wt[2] = 1.0f; }  // env_locomotion.py:115-116
    t.fc0 = (float)fl.touch0; t.fc1 = (float)fl.touch1;                                // robots.py:74-86
    t.fc2 = (float)fl.touch2; t.fc3 = (float)fl.touch3;
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs, t.fc2, t.fc3);
    if (!ro.finite) t.done = 1;                                                        // :205-207
    const float old = t.linpot;
    float dist, ang;
    calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
    const float progress = t.linpot - old;
    float posture = 0.0f;
    const float pitch = ro.rpy[1], roll = ro.rpy[0];
    if (!(-0.2f < pitch && pitch < 0.4f)) posture = fabsf(pitch);                      // :178-183
    if (!(-0.4f < roll && roll < 0.4f)) posture += fabsf(roll);
    const float e1 = wave_sum(lane < T::NJ ? fabsf(act_raw * ro.spd) : 0.0f);
    const float e2 = wave_sum(lane < T::NJ ? act_raw * act_raw : 0.0f);
    const float energy = M->electricity_cost * (e1 / T::NJ) + M->stall_torque_cost * (e2 / T::NJ);
    const float joints = M->joints_at_limit_cost * (float)ro.jal;
    float tall = ro.height > M->termination_height ? 2.0f : -1.0f;
    if (tall < 0) t.done = 1;
    if (M->task_flags & MOCCA_TASKF_BODY_CONTACT) {                                    // LaikagoCustomEnv, :877-890
      tall = 0.0f;
      if (fl.body_touch) { tall = -1.0f; t.done = 1; }
    }
    float bonus = 0.0f;
    if (dist < 0.15f) { t.close += 1; bonus = 2.0f; }                                  // :198-202
    if ((float)t.close >= t.stopf && !a.host_retarget) {                               // :214-222
      t.close = 0;
      randomize_target(a, env + a.env_offset, t);
      t.wt[0] += t.dist * cosf(t.angle);
      t.wt[1] += t.dist * sinf(t.angle);
      calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
    }
    rew = progress + bonus - energy + tall - posture - joints;                         // :121-122
    if (lane == 0) softsign_tail(dist, ang, obs + NBO);
    if (M->task_flags & MOCCA_TASKF_NEVER_DONE) t.done = 0;                            // Walker2DCustomEnv.step, :302-309
  } else {
    // env_locomotion.py:515-568
    t.setstop = (t.nsi == 6 || t.nsi == 7 || t.nsi == 13 || t.nsi == 14);             // :522
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs);                         // previous step's contacts, :525
    if (!ro.finite) t.done = 1;
    const int cur_idx = t.nsi;
    // calc_feet_state :632-674
    float fd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float dx = L[L_FEET + 3 * k] - ter[6 * t.nsi], dy = L[L_FEET + 3 * k + 1] - ter[6 * t.nsi + 1];
      fd[k] = sqrtf(dx * dx + dy * dy);
    }
    t.fc0 = (float)fl.touch0; t.fc1 = (float)fl.touch1;
    const bool reached = fl.target0 || fl.target1;
    if (reached) {
      t.trc += 1;
      if (t.trc > 120) { t.stop = 0; t.setstop = 0; }
      if (t.trc >= 2) {
        if (!t.stop) {
          t.nsi += 1;
          t.trc = 0;
          if (t.nsi >= MOCCA_MAX_PLANKS) {                                              // update_steps :472-479
            const int oldest = t.nsi % MOCCA_MAX_PLANKS;
            const int nx = t.nsi < MOCCA_MAX_TERRAIN_STEPS - 1 ? t.nsi : MOCCA_MAX_TERRAIN_STEPS - 1;
            if (lane == 0) ter[120 + oldest] = (float)nx;
          }
        }
        t.stop = t.setstop;
      }
      if (t.nsi >= MOCCA_MAX_TERRAIN_STEPS) t.nsi -= 1;
    }
    // calc_base_reward :598-630
    const float old = t.linpot;
    float dist, ang;
    calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
    const float progress = t.linpot - old;
    float posture = 0.0f;
    const float pitch = ro.rpy[1], roll = ro.rpy[0];
    if (!(-0.2f < pitch && pitch < 0.4f)) posture = fabsf(pitch);
    if (!(-0.4f < roll && roll < 0.4f)) posture += fabsf(roll);
    const float e1 = wave_sum(lane < T::NJ ? fabsf(act_raw * ro.spd) : 0.0f);
    const float e2 = wave_sum(lane < T::NJ ? act_raw * act_raw : 0.0f);
    const float energy = M->electricity_cost * (e1 / T::NJ) + M->stall_torque_cost * (e2 / T::NJ);
    const float joints = M->joints_at_limit_cost * (float)ro.jal;
    const float term_h = 0.75f + (0.45f - 0.75f) * t.cur / 9;                          // :368
    const float tall = ro.height > term_h ? 2.0f : -1.0f;
    if (tall < 0) t.done = 1;
    // calc_step_reward :676-693
    const int last = MOCCA_MAX_TERRAIN_STEPS - 1;
    float step_bonus = 0.0f, bonus = 0.0f;
    if (reached && t.trc == 1 && t.nsi != last) step_bonus = 50.0f * powf(2.718f, -fminf(fd[0], fd[1]) / 0.25f);
    if ((t.nsi == last || t.stop) && dist < 0.15f) bonus = 2.0f;
    __threadfence_block();
    delta_to_k_targets(L, ter, t, ro.rpy[2], lane, obs + NBO);
    if (cur_idx != t.nsi) calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
    rew = progress - energy + step_bonus + bonus + tall - posture - joints;            // :528-531
    info = t.nsi;
  }
  t.prevx = L[L_BASE];
  const int timeout = t.t >= M->max_episode_steps;
  const int dflag = (t.done ? 1 : 0) | (timeout ? 2 : 0);
  if (lane == 0) {
    a.rew[env] = rew;
    a.done[env] = (uint8_t)dflag;
    if (a.info) a.info[env] = info;
  }
  STAMP(26);  // observation + reward done
  if (a.auto_reset && dflag) {
    wsync();
    reset_env<T, TASK>(a, M, L, ter, env + a.env_offset, lane, t, obs);
  }
  wsync();
  store_dyn(st, L, lane, T::NJ, T::NSLOT);
  if (lane == 0) store_task(tk, t, T::NFEET > 2);
  STAMP(25);  // reset (if any) + write-back done
#ifdef MOCCA_STAMPS
  if (lane == 0 && blockIdx.x < STAMP_WAVES) g_stamps[blockIdx.x * STAMP_SLOTS + 24] = (unsigned long long)(a.auto_reset && dflag);
#endif
}

template <class T, int TASK>
__global__ __launch_bounds__(64) void mocca_reset_kernel(StepArgs a) {
  __shared__ float L[L_TOTAL];
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= a.n_envs) return;
  if (a.mask && !a.mask[env]) return;
  ModelP M = (ModelP)a.model;
  float* st = a.dyn + (size_t)env * DYN_STRIDE;
  uint32_t* tk = a.task + (size_t)env * MOCCA_TASK_WORDS;
  float* ter = TASK == MOCCA_TASK_WALKER3D_STEPPER ? a.terrain + (size_t)env * TERRAIN_STRIDE : nullptr;
  TaskRegs t;
  load_task(tk, t, T::NFEET > 2);
  if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    cassie_reset_env<T>(M, L, lane, t, a.obs + (size_t)env * a.obs_dim);
    if (lane < MOCCA_MAX_CTRL) tk[T_JVEL + lane] = 0u;
  } else {
    reset_env<T, TASK>(a, M, L, ter, env + a.env_offset, lane, t, a.obs + (size_t)env * a.obs_dim);
  }
  wsync();
  store_dyn(st, L, lane, T::NJ, T::NSLOT);
  if (lane == 0) store_task(tk, t, T::NFEET > 2);
}

// calc_state + observation tail on the stored state (no physics, no randomness)
template <class T, int TASK>
__global__ __launch_bounds__(64) void mocca_observe_kernel(StepArgs a) {
  __shared__ float L[L_TOTAL];
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= a.n_envs) return;
  ModelP M = (ModelP)a.model;
  const float* st = a.dyn + (size_t)env * DYN_STRIDE;
  uint32_t* tk = a.task + (size_t)env * MOCCA_TASK_WORDS;
  const float* ter = TASK == MOCCA_TASK_WALKER3D_STEPPER ? a.terrain + (size_t)env * TERRAIN_STRIDE : nullptr;
  float* obs = a.obs + (size_t)env * a.obs_dim;
  load_dyn(st, L, lane, T::NJ, T::NSLOT);
  TaskRegs t;
  load_task(tk, t, T::NFEET > 2);
  if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
  wsync();
  stage_joints<T>(M, L, lane);
  walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
  wsync();
  constexpr int NBO = 6 + 2 * T::NJ + T::NFEET;
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    bool fin;
    cassie_obs<T>(M, L, lane, t.initz, obs, &fin);
    t.linpot = cassie_potential(M, L);
  } else {
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs, t.fc2, t.fc3);
    float dist, ang;
    if (TASK == MOCCA_TASK_WALKER3D_CUSTOM) {
      calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
      if (lane == 0) softsign_tail(dist, ang, obs + NBO);
    } else {
      delta_to_k_targets(L, ter, t, ro.rpy[2], lane, obs + NBO);
      calc_potential(M, L, t, ro.rpy[2], &dist, &ang);
    }
    t.prevx = L[L_BASE];
  }
  if (lane == 0) store_task(tk, t, T::NFEET > 2);
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
struct mocca_ctx {
  MoccaModel model;
  int task_id = 0, n_envs = 0, device = 0, obs_dim = 0;
  MoccaModel* d_model = nullptr;
  int topo = 0;  // TOPO_*
  float* d_dyn = nullptr;
  uint32_t* d_task = nullptr;
  float* d_terrain = nullptr;
  int auto_reset = 0, eval_mode = 0, random_pose = 1, curriculum = 0, host_retarget = 0, env_offset = 0;
  uint64_t seed = 0;
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
  if (m.max_rows > MAXR || m.max_contacts > MAXC || m.max_rows < 1 + 3 * T::NCLOS || m.n_pairs > MOCCA_MAX_PAIRS || m.n_feet != T::NFEET ||
      m.n_ctrl > MOCCA_MAX_CTRL || m.n_ordered > MOCCA_MAX_CTRL) {
    err = "model blob caps exceed the kernel's (max_rows <= 48, max_contacts <= 12, n_feet as compiled)";
    return MOCCA_E_ARG;
  }
  return MOCCA_OK;
}
// compiled topologies: the tree of the blob selects the kernel instance
enum { TOPO_WALKER3D = 0, TOPO_CASSIE = 1, TOPO_WALKER2D = 2, TOPO_CRAB2D = 3, TOPO_LAIKAGO = 4 };
static int check_topology(const MoccaModel& m, int task_id, int* topo, std::string& err) {
  if (task_id == MOCCA_TASK_CASSIE) { *topo = TOPO_CASSIE; return check_topology_t<TopoCassie>(m, "TopoCassie", err); }
  if (task_id == MOCCA_TASK_WALKER3D_CUSTOM && m.n_bodies == TopoWalker2D::NB) {
    *topo = TOPO_WALKER2D; return check_topology_t<TopoWalker2D>(m, "TopoWalker2D", err);
  }
  if (task_id == MOCCA_TASK_WALKER3D_CUSTOM && m.n_bodies == TopoCrab2D::NB) {
    *topo = TOPO_CRAB2D; return check_topology_t<TopoCrab2D>(m, "TopoCrab2D", err);
  }
  if (task_id == MOCCA_TASK_WALKER3D_CUSTOM && m.n_bodies == TopoLaikago::NB) {
    *topo = TOPO_LAIKAGO; return check_topology_t<TopoLaikago>(m, "TopoLaikago", err);
  }
  *topo = TOPO_WALKER3D;
  return check_topology_t<TopoWalker3D>(m, "TopoWalker3D", err);
}

// kernel selection by (topology, task id)
template <template <class, int> class Launcher, class... Args>
static void dispatch(int topo, int task_id, Args... args) {
  if (topo == TOPO_CASSIE) Launcher<TopoCassie, MOCCA_TASK_CASSIE>::run(args...);
  else if (topo == TOPO_WALKER2D) Launcher<TopoWalker2D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (topo == TOPO_CRAB2D) Launcher<TopoCrab2D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (topo == TOPO_LAIKAGO) Launcher<TopoLaikago, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (task_id == MOCCA_TASK_WALKER3D_CUSTOM) Launcher<TopoWalker3D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else Launcher<TopoWalker3D, MOCCA_TASK_WALKER3D_STEPPER>::run(args...);
}
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

extern "C" {

int mocca_abi_version(void) { return MOCCA_ABI_VERSION; }
size_t mocca_model_sizeof(void) { return sizeof(MoccaModel); }

const char* mocca_last_error(mocca_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int mocca_create(const void* model_blob, size_t nbytes, int task_id, int n_envs, int device, mocca_handle* out) {
  if (!out) return MOCCA_E_ARG;
  *out = nullptr;
  if (!model_blob || nbytes != sizeof(MoccaModel)) { g_err = "model blob has the wrong size"; return MOCCA_E_ARG; }
  if (n_envs <= 0) { g_err = "n_envs must be positive"; return MOCCA_E_ARG; }
  if (task_id != MOCCA_TASK_WALKER3D_CUSTOM && task_id != MOCCA_TASK_WALKER3D_STEPPER && task_id != MOCCA_TASK_CASSIE) {
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
  h->obs_dim = task_id == MOCCA_TASK_CASSIE
                   ? 6 + 2 * h->model.n_ordered + 2
                   : 6 + 2 * h->model.n_joints + h->model.n_feet + (task_id == MOCCA_TASK_WALKER3D_CUSTOM ? 2 : 15);
  auto fail = [&](const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    mocca_destroy(h);
    return MOCCA_E_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipMalloc(&h->d_model, sizeof(MoccaModel))) != hipSuccess) return fail("hipMalloc(model)", e);
  if ((e = hipMemcpy(h->d_model, &h->model, sizeof(MoccaModel), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(model)", e);
  const size_t dyn_b = (size_t)n_envs * DYN_STRIDE * sizeof(float), task_b = (size_t)n_envs * MOCCA_TASK_WORDS * 4;
  const size_t ter_b = (size_t)n_envs * TERRAIN_STRIDE * sizeof(float);
  if ((e = hipMalloc(&h->d_dyn, dyn_b)) != hipSuccess) return fail("hipMalloc(state)", e);
  if ((e = hipMalloc(&h->d_task, task_b)) != hipSuccess) return fail("hipMalloc(task)", e);
  if ((e = hipMalloc(&h->d_terrain, ter_b)) != hipSuccess) return fail("hipMalloc(terrain)", e);
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

// The handle's buffers live on h->device: launches and copies are issued with that device current, whatever the
// caller's current device is (restored on return).  hipGetDevice / hipSetDevice are thread-local bookkeeping.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static StepArgs make_args(mocca_handle h) {
  StepArgs a{};
  a.model = h->d_model; a.dyn = h->d_dyn; a.task = h->d_task; a.terrain = h->d_terrain;
  a.n_envs = h->n_envs; a.obs_dim = h->obs_dim;
  a.auto_reset = h->auto_reset; a.eval_mode = h->eval_mode; a.random_pose = h->random_pose; a.curriculum = h->curriculum;
  a.host_retarget = h->host_retarget; a.env_offset = h->env_offset;
  a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
  return a;
}

int mocca_reset(mocca_handle h, const uint8_t* mask_dev, uint64_t seed, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return MOCCA_E_ARG;
  h->seed = seed;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.mask = mask_dev; a.obs = obs_dev;
  hipStream_t s = (hipStream_t)stream;
  dispatch<LaunchReset>(h->topo, h->task_id, h->n_envs, s, a);
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_step(mocca_handle h, const float* act_dev, float* obs_dev, float* rew_dev, uint8_t* done_dev, int32_t* info_dev,
               void* stream) {
  if (!h || !act_dev || !obs_dev || !rew_dev || !done_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.act = act_dev; a.obs = obs_dev; a.rew = rew_dev; a.done = done_dev; a.info = info_dev;
  hipStream_t s = (hipStream_t)stream;
  dispatch<LaunchStep>(h->topo, h->task_id, h->n_envs, s, a);
  HIP_TRY(h, hipGetLastError());
  return MOCCA_OK;
}

int mocca_observe(mocca_handle h, float* obs_dev, void* stream) {
  if (!h || !obs_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
  StepArgs a = make_args(h);
  a.obs = obs_dev;
  hipStream_t s = (hipStream_t)stream;
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
  HIP_TRY(h, hipMemcpyAsync(task_dev, h->d_task, (size_t)h->n_envs * MOCCA_TASK_WORDS * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MOCCA_OK;
}
int mocca_set_task(mocca_handle h, const uint32_t* task_dev, void* stream) {
  if (!h || !task_dev) return MOCCA_E_ARG;
  DeviceGuard guard(h->device);
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
    case MOCCA_PARAM_EVAL_MODE: h->eval_mode = value != 0; break;
    case MOCCA_PARAM_CURRICULUM: h->curriculum = (int)value < 0 ? 0 : ((int)value > 9 ? 9 : (int)value); break;
    case MOCCA_PARAM_RANDOM_POSE: h->random_pose = value != 0; break;
    case MOCCA_PARAM_HOST_RETARGET: h->host_retarget = value != 0; break;
    case MOCCA_PARAM_SEED: h->seed = (uint64_t)value; break;
    case MOCCA_PARAM_ENV_OFFSET: h->env_offset = (int)value; break;
    default: h->err = "unknown parameter id"; return MOCCA_E_ARG;
  }
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
  dispatch<KernelInfo>(h->topo, h->task_id, &fa, &nb, &e);
  HIP_TRY(h, e);
  if (vgprs) *vgprs = fa.numRegs;
  if (sgprs) *sgprs = 0;
  if (lds_bytes) *lds_bytes = (int)fa.sharedSizeBytes;
  if (scratch_bytes) *scratch_bytes = (int)fa.localSizeBytes;
  if (max_blocks_per_cu) *max_blocks_per_cu = nb;
  return MOCCA_OK;
}

}  // extern "C"
