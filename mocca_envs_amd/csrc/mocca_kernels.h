// mocca_kernels.h -- the __global__ entry points (templates over topology, task and the INJECT switch).
// Instantiated by mocca_api.hip (physics kernels: INJECT = false) and mocca_task.hip (task-layer kernels behind
// mocca_task_step / taped resets: INJECT = true), two translation units so they compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "mocca.h"
#include "mocca_device.h"

namespace MOCCA_NS {

#ifndef MOCCA_WAVES_PER_EU
#define MOCCA_WAVES_PER_EU 4
#endif

// the observation assembled in LDS (L_OBS) leaves in one coalesced store
DI void flush_obs(const float* L, float* dst, int lane, int obs_dim) {
#pragma unroll 1
  for (int k = lane; k < obs_dim; k += 64) dst[k] = L[L_OBS + k];
}
// An env that ends under auto-reset: its terminal observation goes to the caller's optional buffer (mocca_set_terminal_obs_buffer) before
// the reset overwrites L_OBS with the first observation of the next episode.  The reference returns the final state together with done
// (env_locomotion.py:128-141) and gym's TimeLimit wrapper does too (__init__.py:55): a trainer bootstraps from it on truncation.
DI void keep_terminal_obs(const StepArgs& a, const float* L, int env, int lane) {
  wsync();
  if (a.final_obs) flush_obs(L, a.final_obs + (size_t)env * a.obs_dim, lane, a.obs_dim);
  wsync();
}

// Monitor + TimeLimitMask inside the launch (mocca_set_episode_stats, StepArgs.ep_*): what the reference's trainers wrap around every env --
// baselines' Monitor (info["episode"] = {r, l} in the step that ends the episode) and a TimeLimitMask (info["bad_transition"] when the episode
// was cut by max_episode_steps, /root/reference/mocca_envs/__init__.py:55) -- and the two mask columns their PPO loop builds from `done` /
// `infos`.  Lane 0; `ret0` is the env's running return loaded by the caller ahead of the observation code (the load's latency hides there).
DI void monitor_emit(const StepArgs& a, int env, float ret0, float rew, int dflag, int length, int info) {
  float r = ret0 + rew;
  if (dflag) {
    if (a.ep_rec) {   // one 16-byte store: {serial of this launch, return, length, done bits | info << 8}
      uint4 rec;
      rec.x = a.ep_serial; rec.y = __float_as_uint(r); rec.z = (uint32_t)length; rec.w = (uint32_t)dflag | ((uint32_t)info << 8);
      ((uint4*)a.ep_rec)[env] = rec;
    }
    if (a.ep_totals) {
      atomicAdd(&a.ep_totals[0], r); atomicAdd(&a.ep_totals[1], (float)length); atomicAdd(&a.ep_totals[2], 1.0f);
      if (dflag & 2) atomicAdd(&a.ep_totals[3], 1.0f);
    }
    r = 0.0f;
  }
  a.ep_ret[env] = r;
  if (a.ep_masks) a.ep_masks[env] = dflag ? 0.0f : 1.0f;
  if (a.ep_bad) a.ep_bad[env] = (dflag & 2) ? 0.0f : 1.0f;
}

// --------------------------------------------------------------------------------------------
// kernels: one 64-lane workgroup (= one wavefront) per environment
// --------------------------------------------------------------------------------------------
// INJECT = true is the task-layer entry (mocca_task_step): the same code with ZERO physics substeps, the contact query
// results taken from the caller (a.inj_*) and, when a tape is attached, its uniforms in place of the Philox draws.
template <class T, int TASK, bool INJECT = false>
__global__ __launch_bounds__(64, MOCCA_WAVES_PER_EU) void mocca_step_kernel(StepArgs a) {
  __shared__ float L[L_TOTAL];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= a.n_envs) return;
  // heaviest envs first (StepArgs.order): a launch of more envs than the chip holds at once ends with its last-started waves, which
  // should be the light ones (longest-processing-time-first); one scalar load, wave-uniform
  const int env = a.order ? uni(a.order[blockIdx.x]) : (int)blockIdx.x;
  ModelP M = (ModelP)a.model;
  float* st = a.dyn + (size_t)env * DYN_STRIDE;
  uint32_t* tk = a.task + (size_t)env * MOCCA_TASK_WORDS;
  float* ter = TASK == MOCCA_TASK_WALKER3D_STEPPER ? a.terrain + (size_t)env * TERRAIN_STRIDE : nullptr;
  float* obs_out = a.obs + (size_t)env * a.obs_dim;
  float* obs = L + L_OBS;   // assembled in LDS, stored once at the end (flush_obs)
  int32_t* dbg = a.dbg ? a.dbg + (size_t)env * MOCCA_DEBUG_WORDS : nullptr;

  // the slots' normal impulses: loaded iff the blob warm-starts, stored iff it does or the caller wants them (INJECT: no physics, the record passes through)
  const bool warm_ld = INJECT || uni(__float_as_int(M->warmstart)) != 0, warm_st = warm_ld || a.persist_warm != 0;
  if (dbg && lane == 0) { dbg[16] = 0; dbg[17] = 0; dbg[18] = 0; }   // step signature: restarted by every mocca_step
  load_dyn(st, L, lane, T::NJ, T::NSLOT, warm_ld);
  pace_start(a, L, lane, INJECT ? 0 : a.pace);
  if (lane == 0) L[L_KEEPWARM] = __int_as_float(warm_st ? 1 : 0);
  // the lane's root->body path, packed 5 bits per step; the only lane-derived value kept across the substeps
  const unsigned long long ppk = T::path_packed(lane < T::NB ? lane : 0);
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    // ---- CassieEnv.step (env_cassie.py:433-479): 50 x { filter joint speeds, PD torques, one physics step }
    const int no = M->n_ordered, nctl = M->n_ctrl, mode = M->cassie_mode;
    float target = 0.0f;  // env_cassie.py:434-443: base angle (residual control) + action, 0 for the springs
    if (lane < nctl) {
      float base = M->ctrl_base[lane];
      if (mode != MOCCA_CASSIE_PLAIN) {  // base_angles() = traj.joint_angles(mocap_time()) at the istep the step starts from (:601-602)
        const int f0 = traj_frame(a, M, (int)tk[T_ISTEP]);
        base = (lane < nctl - 2 && M->residual_control) ? a.traj[(size_t)f0 * MOCCA_TRAJ_STRIDE + M->ctrl_oidx[lane]] : 0.0f;
      }
      target = base + (lane < nctl - 2 ? a.act[(size_t)env * (nctl - 2) + lane] : 0.0f);
    }
    if (lane < no) {
      L[L_JVEL + lane] = __uint_as_float(tk[T_JVEL + lane]);
      L[L_Q0 + lane] = L[L_Q + M->ordered_body[lane]];
    }
    if (lane < MOCCA_MAX_BODIES) L[L_TAU + lane] = 0.0f;
    if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
    wsync();
    const int nllc = INJECT ? 0 : M->n_llc;
    int last_rows = uni((int)tk[T_RES23]);
    if (!INJECT && a.pace == 0) set_issue_priority(last_rows, a.prio);
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
      substep<T, TASK>(Ms, L, ln, nullptr, 0, pk, dbg, a.prio, last_rows, HeightFieldArgs{nullptr, 0, 0, 0.0f}, it, nllc);
    }
    if (!INJECT && lane == 0) tk[T_RES23] = (uint32_t)last_rows;
    TaskRegs t;
    load_task(tk, t, false, true);
    const float ep_ret0 = a.ep_ret ? a.ep_ret[env] : 0.0f;
    t.istep += M->n_llc;  // pd_control counts every low-level iteration (:381); the task-layer entry replays a whole env.step
    float jv = 0.0f;
    if (lane < no) {  // :467-468 finite-difference joint velocity over the control step
      jv = (L[L_Q + M->ordered_body[lane]] - L[L_Q0 + lane]) / M->control_dt;
      tk[T_JVEL + lane] = __float_as_uint(jv);
    }
    stage_joints<T>(M, L, lane);
    walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
    wsync();
    t.t += 1;
    const CassieState cs = cassie_state(M, L, lane, t.initz);
    const float old = t.linpot;
    t.linpot = cassie_potential(M, L);
    const float alive = cs.height > M->alive_height ? 2.0f : -1.0f;  // compute_rewards :401-414
    if (!cs.finite || alive < 0.0f) t.done = 1;
    float rew = alive + (t.linpot - old);
    if (mode == MOCCA_CASSIE_PLAIN) {
      cassie_obs(M, L, lane, cs, t.initz, obs);
    } else {  // CassieMocapRewEnv.compute_rewards replaces the reward, keeps `dead` (:495-531); get_obs of the phase envs
      float phase;
      const int f1 = traj_frame(a, M, t.istep, &phase);
      rew = cassie_mocap_reward(a, M, L, lane, cs, jv, f1);
      cassie_mocap_obs(M, L, lane, cs, jv, phase, obs);
    }
    const int timeout = t.t >= M->max_episode_steps;
    const int dflag = (t.done ? 1 : 0) | (timeout ? 2 : 0);
    if (lane == 0) {
      a.rew[env] = rew;
      a.done[env] = (uint8_t)dflag;
      if (a.info) a.info[env] = 0;
      if (a.ep_ret) monitor_emit(a, env, ep_ret0, rew, dflag, t.t, 0);
    }
    if (a.auto_reset && dflag) {
      keep_terminal_obs(a, L, env, lane);
      cassie_reset_env<T, INJECT>(a, M, L, env + a.env_offset, lane, t, obs);
      if (lane < MOCCA_MAX_CTRL) tk[T_JVEL + lane] = __float_as_uint(L[L_JVEL + lane]);
    }
    wsync();
    flush_obs(L, obs_out, lane, a.obs_dim);
    store_dyn(st, L, lane, T::NJ, T::NSLOT, uni(__float_as_int(L[L_KEEPWARM])) != 0);
    if (lane == 0) store_task(tk, t, false, true);
    if (!INJECT) pace_finish(a, L, lane, a.pace);
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
  ContactFlags fl = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int cover = 0;   // Stepper: cover mask of the last substep's contacts (mocca_device.h cover_targets)
  const int nsub = INJECT ? 0 : M->n_substeps;
  const int nsi0 = TASK == MOCCA_TASK_WALKER3D_STEPPER ? (int)tk[T_NSI] : 0;
  // the env's row count at the end of the step before (task word 23) sets the issue priority until the first substep knows better:
  // a heavy env is almost always still heavy, and more than half of a substep runs before its own count is known
  int last_rows = uni((int)tk[T_RES23]);
  if (!INJECT && a.pace == 0) set_issue_priority(last_rows, a.prio);
#pragma unroll 1
  for (int s = 0; s < nsub; ++s) {
    // launder the model pointer: keeps LICM from hoisting dozens of loop-invariant model loads out of the
    // substep loop, where they would sit in registers (and spill to scratch) for the whole kernel
    ModelP Ms = M;
    int ln = lane;  // same for lane-derived offsets and predicates (recomputing them costs a few instructions)
    unsigned long long pk = ppk;  // laundered too: otherwise every (ppk >> 5k) & 31 and the addresses derived from it
    asm volatile("" : "+s"(Ms), "+v"(ln), "+v"(pk));  // are hoisted out of the loop and spilled
    fl = substep<T, TASK>(Ms, L, ln, ter, nsi0, pk, dbg, a.prio, last_rows, HeightFieldArgs{a.hf, a.hf_rows, a.hf_cols, a.hf_scale}, s, nsub, &cover);
  }
  if constexpr (INJECT) {  // getContactPoints results handed in by the caller (robots.py:74-86, env_locomotion.py:634-650, :880-890)
    const int32_t* tc = a.inj_touch + (size_t)env * T::NFEET;
    fl.touch0 = tc[0] != 0; fl.touch1 = tc[1] != 0;
    if constexpr (T::NFEET > 2) { fl.touch2 = tc[2] != 0; fl.touch3 = tc[3] != 0; }
    if (a.inj_target) {   // per foot: 1 = on the cover of the target plank (plank next_step_index mod n_planks at the step's start), 2 = of the plank after it
      const int32_t* tg = a.inj_target + (size_t)env * T::NFEET;
      const int npl = TASK == MOCCA_TASK_WALKER3D_STEPPER ? M->n_planks : 1;
#pragma unroll
      for (int f = 0; f < T::NFEET; ++f)
        if (tg[f] == 1 || tg[f] == 2) cover |= 1 << (4 * f + (nsi0 + tg[f] - 1) % npl);
      fl.target0 = tg[0] == 1; fl.target1 = tg[1] == 1;
      if constexpr (T::NFEET > 2) { fl.target2 = tg[2] == 1; fl.target3 = tg[3] == 1; }
    }
    if (a.inj_body) fl.body_touch = a.inj_body[env] != 0;
  }
  if (!INJECT && lane == 0) tk[T_RES23] = (uint32_t)last_rows;
  STAMP(27);  // substeps done
  TaskRegs t;
  load_task(tk, t, T::NFEET > 2);
  const float ep_ret0 = a.ep_ret ? a.ep_ret[env] : 0.0f;
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
  if (TASK == MOCCA_TASK_WALKER3D_PLANNER) {
    // Walker3DPlannerEnv.step (env_locomotion.py:1075-1128).  calc_state() is called without contact ids there: feet_contact keeps the zeros
    // of robot.reset.  reward = progress; the second term of the reference, log(max(1, base_value)) / 3, is the external base controller's
    // value estimate and is added by the caller.
    t.fc0 = 0.0f; t.fc1 = 0.0f; t.fc2 = 0.0f; t.fc3 = 0.0f;
    RobotObs ro = robot_obs<T>(M, L, lane, 0.0f, 0.0f, obs);
    const float old = t.linpot;
    float dist, cd, sd;
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
    rew = t.linpot - old;
    // done = done or relative torso height < termination_height or z < -5 or the torso link touches anything (:1103-1111)
    if (ro.height < M->termination_height || L[L_BASE + 2] < M->fall_z || fl.body_touch) t.done = 1;
    if (lane == 0) softsign_tail(sd, cd, obs + NBO);
  } else if (TASK == MOCCA_TASK_WALKER3D_CUSTOM) {
    const bool evalm = live_eval_mode(a, env);
    if (evalm) { t.wt[0] = t.prevx + 4.0f; t.wt[1] = 0.0f; t.wt[2] = 1.0f; }  // env_locomotion.py:115-116
    t.fc0 = (float)fl.touch0; t.fc1 = (float)fl.touch1;                                // robots.py:74-86
    t.fc2 = (float)fl.touch2; t.fc3 = (float)fl.touch3;
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs, t.fc2, t.fc3);
    if (!ro.finite) t.done = 1;                                                        // :205-207
    const float old = t.linpot;
    float dist, cd, sd;
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
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
      randomize_target<INJECT>(a, env + a.env_offset, t, evalm);
      float sa, ca;
      fast_sincos(t.angle, &sa, &ca);
      t.wt[0] += t.dist * ca;
      t.wt[1] += t.dist * sa;
      calc_potential(M, L, t, ro, &dist, &cd, &sd);
    }
    rew = progress + bonus - energy + tall - posture - joints;                         // :121-122
    if (lane == 0) softsign_tail(sd, cd, obs + NBO);
    if (M->task_flags & MOCCA_TASKF_NEVER_DONE) t.done = 0;                            // Walker2DCustomEnv.step, :302-309
  } else {
    // env_locomotion.py:515-568
    t.setstop = (t.nsi == 6 || t.nsi == 7 || t.nsi == 13 || t.nsi == 14);             // :522
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs, t.fc2, t.fc3);           // previous step's contacts, :525
    if (!ro.finite) t.done = 1;
    const int cur_idx = t.nsi;
    const int n_planks = M->n_planks;
    // calc_feet_state :632-674
    float fdmin = 1e30f;
#pragma unroll
    for (int k = 0; k < T::NFEET; ++k) {
      const float dx = L[L_FEET + 3 * k] - ter[6 * t.nsi], dy = L[L_FEET + 3 * k + 1] - ter[6 * t.nsi + 1];
      fdmin = fminf(fdmin, sqrtf(dx * dx + dy * dy));
    }
    t.fc0 = (float)fl.touch0; t.fc1 = (float)fl.touch1;
    t.fc2 = (float)fl.touch2; t.fc3 = (float)fl.touch3;
    const bool reached = fl.target0 || fl.target1 || fl.target2 || fl.target3;
    if (reached) {
      t.trc += 1;
      if (t.trc > 120) { t.stop = 0; t.setstop = 0; }
      if (t.trc >= 2) {
        if (!t.stop) {
          t.nsi += 1;
          t.trc = 0;
          if (t.nsi >= n_planks) {                                                      // update_steps :472-479
            const int oldest = t.nsi % n_planks;
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
    float dist, cd, sd;
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
    float progress = t.linpot - old;
    float posture = 0.0f, tall;
    const float pitch = ro.rpy[1], roll = ro.rpy[0];
    const float e1 = wave_sum(lane < T::NJ ? fabsf(act_raw * ro.spd) : 0.0f);
    const float e2 = wave_sum(lane < T::NJ ? act_raw * act_raw : 0.0f);
    const float energy = M->electricity_cost * (e1 / T::NJ) + M->stall_torque_cost * (e2 / T::NJ);
    const float joints = M->joints_at_limit_cost * (float)ro.jal;
    if (!(M->task_flags & MOCCA_TASKF_QUADRUPED_STEPPER)) {
      if (!(-0.2f < pitch && pitch < 0.4f)) posture = fabsf(pitch);
      if (!(-0.4f < roll && roll < 0.4f)) posture += fabsf(roll);
      // terminal_height_curriculum[self.curriculum], :368,628: the env's CURRENT curriculum (set_env_params acts at once
      // on this line, at the next reset on terrain and gain)
      const float term_h = M->term_height_cur[0] + (M->term_height_cur[1] - M->term_height_cur[0]) * live_curriculum(a, env) / 9;
      tall = ro.height > term_h ? 2.0f : -1.0f;
      if (tall < 0) t.done = 1;
    } else {
      // LaikagoStepperEnv.calc_base_reward, :928-979: posture from the hip_x / hip_y / knee angles in degrees, progress x 2,
      // posture x 0.2, tall_bonus 2, the time-based early termination REPLACES done, a non-foot link on a plank ends it
      const float R2D = 57.29577951308232f, D2R = 0.017453292519943295f;
      float pj = 0.0f;
      if (lane < T::NJ) {
        const float adeg = L[L_Q + 1 + lane] * R2D;
        const int kind = lane % 3;
        const float lo_ = kind == 0 ? -25.0f : (kind == 1 ? -35.0f : -75.0f), hi_ = kind == 0 ? 25.0f : (kind == 1 ? 35.0f : -15.0f);
        if (!(lo_ < adeg && adeg < hi_)) pj = fabsf(adeg * D2R);
      }
      posture = wave_sum(pj);
      if (!(-25.0f < pitch * R2D && pitch * R2D < 25.0f)) posture += fabsf(pitch);
      progress *= 2.0f;
      posture *= 0.2f;
      tall = 2.0f;
      t.done = (t.t > 240 && t.nsi <= 4);
      if (fl.body_touch) { tall = -1.0f; t.done = 1; }
    }
    // calc_step_reward :676-693
    const int last = MOCCA_MAX_TERRAIN_STEPS - 1;
    float step_bonus = 0.0f, bonus = 0.0f;
    if (reached && t.trc == 1 && t.nsi != last) step_bonus = 50.0f * powf(2.718f, -powf(fdmin, M->step_bonus_smoothness) / 0.25f);
    if ((t.nsi == last || t.stop) && dist < 0.15f) bonus = 2.0f;
    __threadfence_block();
    delta_to_k_targets(M, L, ter, t, ro, lane, obs + NBO);
    if (cur_idx != t.nsi) calc_potential(M, L, t, ro, &dist, &cd, &sd);
    if (!a.random_reward) {
      rew = progress - energy + step_bonus + bonus + tall - posture - joints;          // :528-531
    } else {                                                                           // :533-547
      const float terms[8] = {progress, -energy, step_bonus, bonus, 0.0f, tall, -posture, -joints};
      rew = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float w;
        if (a.random_reward == 1) {  // np_random.uniform(0.8, 1.2, 8): eight draws of this episode's stream
          w = 0.8f + 0.4f * draw_u<INJECT>(a, env + a.env_offset, t.episode, t.draw + k);
          if (lane == 0) tk[T_RW + k] = __float_as_uint(w);
        } else {
          w = __uint_as_float(tk[T_RW + k]);   // the host drew them (MOCCA_PARAM_RANDOM_REWARD = 2)
        }
        rew += w * terms[k];
      }
      if (a.random_reward == 1) t.draw += 8;
    }
    info = t.nsi;
    t.cover = cover;   // what a reset() right after this step would still see (MOCCA_TASKF_STALE_RESET_CONTACTS)
  }
  t.prevx = L[L_BASE];
  const int timeout = t.t >= M->max_episode_steps;
  const int dflag = (t.done ? 1 : 0) | (timeout ? 2 : 0);
  if (lane == 0) {
    a.rew[env] = rew;
    a.done[env] = (uint8_t)dflag;
    if (a.info) a.info[env] = info;
    if (a.ep_ret) monitor_emit(a, env, ep_ret0, rew, dflag, t.t, info);
  }
  STAMP(26);  // observation + reward done
  if (a.auto_reset && dflag) {
    keep_terminal_obs(a, L, env, lane);
    reset_env<T, TASK, INJECT>(a, M, L, ter, env + a.env_offset, lane, t, obs);
  }
  wsync();
  flush_obs(L, obs_out, lane, a.obs_dim);
  store_dyn(st, L, lane, T::NJ, T::NSLOT, uni(__float_as_int(L[L_KEEPWARM])) != 0);
  if (lane == 0) { store_task(tk, t, T::NFEET > 2); if (TASK == MOCCA_TASK_WALKER3D_STEPPER) store_task_cover(tk, t); }
  if (!INJECT) pace_finish(a, L, lane, a.pace);
  STAMP(25);  // reset (if any) + write-back done
#ifdef MOCCA_STAMPS
  if (lane == 0 && blockIdx.x < STAMP_WAVES) g_stamps[blockIdx.x * STAMP_SLOTS + 24] = (unsigned long long)(a.auto_reset && dflag);
#endif
}

template <class T, int TASK, bool INJECT = false>
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
  load_task(tk, t, T::NFEET > 2, TASK == MOCCA_TASK_CASSIE);
  if (TASK == MOCCA_TASK_WALKER3D_STEPPER) load_task_cover(tk, t);
  if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; if (a.ep_ret) a.ep_ret[env] = 0.0f; }   // Monitor.reset: a new episode's return starts at 0
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    cassie_reset_env<T, INJECT>(a, M, L, env + a.env_offset, lane, t, L + L_OBS);
    if (lane < MOCCA_MAX_CTRL) tk[T_JVEL + lane] = __float_as_uint(L[L_JVEL + lane]);
  } else {
    reset_env<T, TASK, INJECT>(a, M, L, ter, env + a.env_offset, lane, t, L + L_OBS);
  }
  wsync();
  flush_obs(L, a.obs + (size_t)env * a.obs_dim, lane, a.obs_dim);
  store_dyn(st, L, lane, T::NJ, T::NSLOT);
  if (lane == 0) { store_task(tk, t, T::NFEET > 2, TASK == MOCCA_TASK_CASSIE); if (TASK == MOCCA_TASK_WALKER3D_STEPPER) store_task_cover(tk, t); }
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
  float* obs = L + L_OBS;
  load_dyn(st, L, lane, T::NJ, T::NSLOT);
  TaskRegs t;
  load_task(tk, t, T::NFEET > 2, TASK == MOCCA_TASK_CASSIE);
  if (lane == 0) { L[L_Q] = 0.0f; L[L_QD] = 0.0f; }
  wsync();
  stage_joints<T>(M, L, lane);
  walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
  wsync();
  constexpr int NBO = 6 + 2 * T::NJ + T::NFEET;
  if constexpr (TASK == MOCCA_TASK_CASSIE) {
    const CassieState cs = cassie_state(M, L, lane, t.initz);
    if (M->cassie_mode == MOCCA_CASSIE_PLAIN) {
      cassie_obs(M, L, lane, cs, t.initz, obs);
    } else {
      float phase;
      traj_frame(a, M, t.istep, &phase);
      cassie_mocap_obs(M, L, lane, cs, lane < MOCCA_MAX_CTRL ? __uint_as_float(tk[T_JVEL + lane]) : 0.0f, phase, obs);
    }
    t.linpot = cassie_potential(M, L);
  } else {
    RobotObs ro = robot_obs<T>(M, L, lane, t.fc0, t.fc1, obs, t.fc2, t.fc3);
    float dist, cd, sd;
    if (TASK == MOCCA_TASK_WALKER3D_CUSTOM || TASK == MOCCA_TASK_WALKER3D_PLANNER) {
      calc_potential(M, L, t, ro, &dist, &cd, &sd);
      if (lane == 0) softsign_tail(sd, cd, obs + NBO);
    } else {
      delta_to_k_targets(M, L, ter, t, ro, lane, obs + NBO);
      calc_potential(M, L, t, ro, &dist, &cd, &sd);
    }
    t.prevx = L[L_BASE];
  }
  wsync();
  flush_obs(L, a.obs + (size_t)env * a.obs_dim, lane, a.obs_dim);
  if (lane == 0) store_task(tk, t, T::NFEET > 2, TASK == MOCCA_TASK_CASSIE);
}


// compiled topologies: the tree of the blob selects the kernel instance
enum { TOPO_WALKER3D = 0, TOPO_CASSIE = 1, TOPO_WALKER2D = 2, TOPO_CRAB2D = 3, TOPO_LAIKAGO = 4,
       TOPO_WALKER3D_MASSIVE = 5, TOPO_CASSIE_MASSIVE = 6 /* the same trees, no link treated as massless (mocca_device.h) */ };
// kernel selection by (topology, task id)
template <template <class, int> class Launcher, class... Args>
static void dispatch(int topo, int task_id, Args... args) {
  if (topo == TOPO_CASSIE || topo == TOPO_CASSIE_MASSIVE) {
    if constexpr (!COMPACT) {   // (closure rows read the body frames after the ABA: Cassie runs the 48-row instance only)
      if (topo == TOPO_CASSIE) Launcher<TopoCassie, MOCCA_TASK_CASSIE>::run(args...);
      else Launcher<TopoCassieMassive, MOCCA_TASK_CASSIE>::run(args...);
    }
  }
  else if (topo == TOPO_WALKER3D_MASSIVE && task_id == MOCCA_TASK_WALKER3D_PLANNER) Launcher<TopoWalker3DMassive, MOCCA_TASK_WALKER3D_PLANNER>::run(args...);
  else if (topo == TOPO_WALKER3D && task_id == MOCCA_TASK_WALKER3D_PLANNER) Launcher<TopoWalker3D, MOCCA_TASK_WALKER3D_PLANNER>::run(args...);
  else if (topo == TOPO_WALKER3D_MASSIVE && task_id == MOCCA_TASK_WALKER3D_CUSTOM) Launcher<TopoWalker3DMassive, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (topo == TOPO_WALKER3D_MASSIVE) Launcher<TopoWalker3DMassive, MOCCA_TASK_WALKER3D_STEPPER>::run(args...);
  else if (topo == TOPO_WALKER2D) Launcher<TopoWalker2D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (topo == TOPO_CRAB2D) Launcher<TopoCrab2D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (topo == TOPO_LAIKAGO && task_id == MOCCA_TASK_WALKER3D_STEPPER) Launcher<TopoLaikago, MOCCA_TASK_WALKER3D_STEPPER>::run(args...);
  else if (topo == TOPO_LAIKAGO) Launcher<TopoLaikago, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else if (task_id == MOCCA_TASK_WALKER3D_CUSTOM) Launcher<TopoWalker3D, MOCCA_TASK_WALKER3D_CUSTOM>::run(args...);
  else Launcher<TopoWalker3D, MOCCA_TASK_WALKER3D_STEPPER>::run(args...);
}

#if !MOCCA_COMPACT
// defined in mocca_task.hip (INJECT = true instances)
void launch_task_step(int topo, int task_id, int n, hipStream_t s, StepArgs a);
void launch_taped_reset(int topo, int task_id, int n, hipStream_t s, StepArgs a);
#endif

}  // namespace MOCCA_NS
