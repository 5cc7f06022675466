/*
 * mocca.h -- C ABI of libmocca_hip.so, the MI355X-native replacement for the
 * pybullet client the reference env owns (`EnvBase._p`,
 * /root/reference/mocca_envs/env_base.py:55) and for everything the reference
 * does through it on the env.step()/reset() path.
 *
 * There is no FFI seam in the reference (it is pure Python over the pybullet
 * CPython extension); the seam is introduced here, at the `_p` object.  Each
 * entry point cites the reference calls it stands in for.
 *
 * Conventions
 *   - every pointer named *_dev is device memory of the GPU the handle was
 *     created on (e.g. torch tensor .data_ptr()); tensors are caller-owned,
 *     contiguous, row-major [n_envs][dim].
 *   - `stream` is a hipStream_t passed as void*; work is enqueued
 *     asynchronously, no call synchronises the device except the *_host
 *     helpers and mocca_create/mocca_destroy.
 *   - return 0 on success, negative MOCCA_E_* otherwise; the message is
 *     available from mocca_last_error().  Nothing throws across the ABI.
 *   - a non-finite state is NOT an error: it sets done, as the reference does
 *     (env_locomotion.py:205-207).
 *   - a handle is not thread-safe; use one handle per (process, device).
 */
#ifndef MOCCA_H
#define MOCCA_H

#include <stddef.h>
#include <stdint.h>

#include "mocca_model.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MOCCA_ABI_VERSION 7

typedef struct mocca_ctx *mocca_handle;

enum {
  MOCCA_OK = 0,
  MOCCA_E_ARG = -1,      /* bad argument / blob */
  MOCCA_E_HIP = -2,      /* HIP runtime error */
  MOCCA_E_TOPOLOGY = -3, /* blob topology differs from the compiled kernel's */
  MOCCA_E_NODEVICE = -4,
};

/* mocca_set_param ids */
enum {
  MOCCA_PARAM_AUTO_RESET = 0,  /* vec-env semantics: a done env is reset inside step() and returns the reset obs */
  MOCCA_PARAM_EVAL_MODE = 1,   /* Walker3DCustomEnv.evaluation_mode(), env_locomotion.py:76-77 */
  MOCCA_PARAM_CURRICULUM = 2,  /* set_env_params({"curriculum": k}), env_base.py:103-106 (takes effect at reset) */
  MOCCA_PARAM_RANDOM_POSE = 3, /* robot_random_start, env_locomotion.py:45 */
  MOCCA_PARAM_HOST_RETARGET = 4, /* Custom env: leave close_count >= stop_frames for the host to re-randomise the
                                    target (env_locomotion.py:214-222) with ITS RandomState, as the facade does */
  MOCCA_PARAM_SEED = 5,        /* Philox key used by step() for in-kernel draws (also set by mocca_reset) */
  MOCCA_PARAM_ENV_OFFSET = 6,  /* global index of this handle's env 0: draws are keyed by (seed, offset + env, episode),
                                  so a shard of a larger batch reproduces exactly the envs it owns */
  MOCCA_PARAM_RANDOM_REWARD = 8, /* Walker3DStepperEnv(random_reward=True), env_locomotion.py:533-547: the 8 reward terms are weighted by
                                    U(0.8, 1.2) numbers drawn every step.  1: drawn in the kernel (8 draws per step); 2: supplied by
                                    the host in task words 30..37 before each step (the single-env classes: np_random stays on the host) */
  MOCCA_PARAM_APPLIED_GAIN = 7, /* set_robot_params({"applied_gain": g}), env_base.py:108-115 / robots.py:16,33: acts on the
                                   next apply_action; the Stepper overwrites it at reset from its curriculum (:489).  The scalar form has
                                   no stream argument: the value is written into the task records by the next call that takes a stream,
                                   on that stream.  A later mocca_set_task restores the snapshot's per-env gains (word 21); the handle's
                                   own copy -- what a Custom env's reset writes -- stays at the last value set here */
  MOCCA_PARAM_ISSUE_PRIORITY = 9, /* TIMING ONLY (no reference counterpart, results do not depend on it): constraint-row counts above which a
                                     wave runs at issue priority 1 / 2 / 3 in the step kernel, packed t1 + 64 t2 + 4096 t3 (each 0..63).
                                     A launch lasts as long as its slowest wave and an env's cost grows with its rows, so the best
                                     thresholds follow the batch's row distribution; default 4 / 7 / 12 (flat-ground walker, blob v13 physics) */
  MOCCA_PARAM_PERSIST_IMPULSES = 10, /* no reference counterpart.  Words 13 + 2 n_joints .. of the state record (mocca_get_state) hold the normal
                                        impulse of every terrain contact slot in the LAST substep.  A blob that warm-starts its contact rows
                                        (MoccaModel.warmstart != 0) reads and writes them every step; the compiled blobs do not (Bullet does not
                                        warm start multibody contacts), and then mocca_step neither loads nor stores them -- 272 B per env-step of
                                        dead traffic until ABI 4 -- unless this parameter is 1 (tests and tools that want the impulses as a
                                        diagnostic).  Results do not depend on it. */
  MOCCA_PARAM_KERNEL_VARIANT = 11,   /* TIMING ONLY.  mocca_create picks the step-kernel instance from the blob: max_rows <= 32 and
                                        max_contacts <= 10 on a tree without loop closures run the COMPACT instance (32 x 32 Delassus matrix,
                                        the articulated-body view aliased under it: less LDS per env, more resident waves per CU); every other
                                        blob within 48 rows / 12 contacts the 48-row instance; a blob whose caps exceed those (max_rows <= 64,
                                        max_contacts <= 20) or that sets MoccaModel.sweep_alternate the 64-row ACCURACY instance (every lane of the wave a row; 17 KB of LDS per env and
                                        a two-waves-per-SIMD register budget: Bullet caps neither rows nor contacts, and this instance exists to
                                        measure what the product's caps change).  1 forces the 48-row instance for a blob that would run the
                                        compact one, 2 forces the 64-row instance for any blob (A/B runs), 0 = automatic.  The instances execute
                                        the same arithmetic in the same order: on the same blob results are bit-identical. */
  MOCCA_PARAM_ORDER_EVERY = 12,      /* TIMING ONLY.  K > 0: every K-th mocca_step first sorts the envs by the constraint-row count their last step
                                        ended with and the step kernel starts the heaviest first (a launch of more envs than the chip holds at
                                        once ends with the waves it started last: they should be the light ones); 0: index order.  Envs never
                                        interact, so the order changes when an env runs, not what it computes. */
  MOCCA_PARAM_PACE_TICKS = 13,       /* TIMING ONLY.  PACE priorities instead of the row-count priorities above.  The hardware serves equal-priority
                                        waves of a SIMD oldest-first, so its four resident waves finish one after the other and the launch waits
                                        for the last, which runs alone.  With a pace P a wave compares the shader-clock ticks it has used with
                                        the share of the env.step it has done (64 units per substep + 2 per constraint row) and runs at issue
                                        priority 3 / 2 / 1 / 0 as its estimated finish lies beyond 17/16 of, beyond, within 1/16 below, or
                                        further below P: the waves of a SIMD finish together.  value > 0: P in ticks; value = -k (1 <= k <= 64):
                                        self-calibrating, P = k/16 of the mean wave time of the last 64 .. 128 sampled waves (one wave in 61 adds
                                        its time to a device-side accumulator; until the first sample a handle falls back to the row-count
                                        priorities); 0: off.  Default -18; 0 for a blob that runs the compact instance (batches beyond one generation of
                                        resident waves: measured slower with it).  Since ABI 7 the calibration keeps no host state: mocca_step can be
                                        captured in a hipGraph with any pace; only ORDER_EVERY > 0 and the episode-record ring (mocca_set_episode_stats)
                                        keep per-launch HOST state that a capture would freeze. */
};

/* words of the per-env debug record (mocca_set_debug_buffer): words 0..11 the active set of the LAST physics substep, words 12..15
 * cumulative over every substep since the caller last cleared the buffer, words 16..18 a signature of EVERY substep of the last mocca_step */
#define MOCCA_DEBUG_WORDS 20
enum {
  MOCCA_DBG_ROWS = 0,        /* constraint rows solved                                  */
  MOCCA_DBG_LIMIT_ROWS = 1,  /* of which joint-limit rows                               */
  MOCCA_DBG_CONTACTS = 2,    /* contacts kept (each gives 1 normal + 2 friction rows)   */
  MOCCA_DBG_SLOTS_LO = 3,    /* bit s: terrain contact slot s is within the margin      */
  MOCCA_DBG_SLOTS_HI = 4,
  MOCCA_DBG_LIMITS_LO = 5,   /* bit 2j + side: limit candidate of joint j (0 lower, 1 upper) */
  MOCCA_DBG_LIMITS_HI = 6,
  MOCCA_DBG_SELF = 7,        /* self-collision pairs within the margin                  */
  MOCCA_DBG_CLAMP_LO = 8,    /* bit l: the row on solver lane l ended the LAST PGS iteration on a bound (unilateral rows at 0, friction  */
  MOCCA_DBG_CLAMP_HI = 9,    /* rows at +-mu lambda_n).  Lanes: limit / closure / planar / normal rows 0.. in row order, friction rows */
                             /* of contact i on lanes 46 - 2i and 47 - 2i                                                               */
  MOCCA_DBG_CLAMPSIG_LO = 10, /* the same mask folded over ALL iterations: sig = rotl64(sig, 7) ^ mask -- every discrete decision the   */
  MOCCA_DBG_CLAMPSIG_HI = 11, /* solver took in the substep                                                                             */
  /* cap pressure (Bullet has neither cap), cumulative: */
  MOCCA_DBG_CAP_CONTACTS = 12, /* substeps in which more contacts were within the margin than max_contacts                             */
  MOCCA_DBG_CAP_ROWS = 13,     /* substeps in which limit + closure + 3 x (kept contacts) rows exceeded max_rows                        */
  MOCCA_DBG_SUBSTEPS = 14,     /* substeps counted                                                                                      */
  MOCCA_DBG_ROWS_WANTED = 15,  /* largest row count an uncapped solver would have held                                                  */
  /* every discrete decision of the last mocca_step, all its substeps (4; Cassie: 50): h = 0 at the step's start, then per substep and for
   * w = 0 .. 11:  h = (h ^ (uint32) word w) * 0x9E3779B97F4A7C15 (mod 2^64).  Two implementations that took the same decisions in every
   * substep agree on it: whole-step comparisons can then be held to arithmetic tolerances (tests/test_gpu_parity.py) */
  MOCCA_DBG_STEPSIG_LO = 16,
  MOCCA_DBG_STEPSIG_HI = 17,
  MOCCA_DBG_STEPSIG_N = 18,    /* substeps folded into it */
  MOCCA_DBG_RESERVED = 19,
};

int mocca_abi_version(void);
size_t mocca_model_sizeof(void);

/* EnvBase.initialize_scene_and_robot (env_base.py:49-101): BulletClient(DIRECT), scene + physics
 * parameters, loadMJCF / loadSDF / loadURDF.  `model_blob` is a MoccaModel; `task_id` a MOCCA_TASK_*. */
int mocca_create(const void *model_blob, size_t nbytes, int task_id, int n_envs, int device, mocca_handle *out);
/* EnvBase.close (env_base.py:44-47): disconnect. */
int mocca_destroy(mocca_handle h);

int mocca_n_envs(mocca_handle h);
int mocca_obs_dim(mocca_handle h);   /* 52 (Custom, env_locomotion.py:58; Planner, :1003-1005) / 65 (Stepper, :386-393) / 36 (CassieEnv) / 42 (CassiePhase*, env_cassie.py:633) */
int mocca_act_dim(mocca_handle h);   /* 21, robots.py:21-23 (the planner envs too: the kernel takes the base controller's joint actions) */
int mocca_state_dim(mocca_handle h); /* MOCCA_STATE_DIM */

/* env.reset() (env_locomotion.py:79-109 / :481-513) for every env whose mask byte is non-zero
 * (NULL = all).  Draws come from Philox keyed by (seed, env, episode).  obs_dev: [N][obs_dim] f32;
 * rows of unmasked envs are left untouched. */
int mocca_reset(mocca_handle h, const uint8_t *mask_dev, uint64_t seed, float *obs_dev, void *stream);

/* env.step(a) (env_locomotion.py:111-141 / :515-568): apply_action (robots.py:31-40) ->
 * stepSimulation (bullet_utils.py:352-353) -> calc_state (robots.py:42-95) -> reward/termination.
 *   act_dev  [N][act_dim] f32 (clipped to [-1,1] inside, robots.py:33)
 *   obs_dev  [N][obs_dim] f32, rew_dev [N] f32
 *   done_dev [N] u8: bit0 = terminated (self.done), bit1 = TimeLimit (max_episode_steps, __init__.py:55)
 *   info_dev [N] i32 or NULL: Stepper "steps_reached" (env_locomotion.py:562-566), 0 for Custom */
int mocca_step(mocca_handle h, const float *act_dev, float *obs_dev, float *rew_dev, uint8_t *done_dev,
               int32_t *info_dev, void *stream);

/* The task layer of env.step(a) alone -- everything of env_locomotion.py:111-141 / :515-568 EXCEPT stepSimulation:
 * the stored dynamic state is taken as the post-physics state, and the contact queries the reference makes after
 * stepping (robots.py:74-86 feet_contact; calc_feet_state's target-plank test, env_locomotion.py:634-650;
 * LaikagoCustomEnv's body contacts, :880-890) are supplied by the caller.  Same kernel source as mocca_step with zero
 * substeps.  Used to replay the reference's own scripted episodes (tests/golden) through the HIP path.
 *   touch_dev  [N][n_feet] i32, target_dev [N][n_feet] i32 or NULL, body_dev [N] i32 or NULL; rest as mocca_step. */
int mocca_task_step(mocca_handle h, const float *act_dev, const int32_t *touch_dev, const int32_t *target_dev,
                    const int32_t *body_dev, float *obs_dev, float *rew_dev, uint8_t *done_dev, int32_t *info_dev,
                    void *stream);

/* Replace the Philox draws of mocca_reset / mocca_task_step by uniforms read from tape_dev [N][n_per_env] f32, indexed
 * by the episode's draw counter (task word 10) -- the numbers `np_random` gave the reference in the recorded episode.
 * NULL detaches.  mocca_step (the physics path) never reads the tape.  The buffer must outlive its use. */
int mocca_set_draw_tape(mocca_handle h, const float *tape_dev, int n_per_env);

/* robot.calc_state() + the task's observation tail on the CURRENT state, without stepping
 * (robots.py:42-95 with env_locomotion.py:102-109 / :712-759).  Used after set_state/set_task, e.g. by the
 * single-env facade whose reset draws come from a host numpy RandomState like the reference's.
 * Updates the task record's potentials (calc_potential) and, for the Stepper, walk_target. */
int mocca_observe(mocca_handle h, float *obs_dev, void *stream);

/* In-memory snapshot of the simulation (the role of saveState/restoreState, env_base.py:101): dynamic
 * state [N][state_dim] f32, task record [N][MOCCA_TASK_WORDS] 32-bit words, terrain [N][128] f32
 * (Stepper: 20 rows x 6 then the n_planks (3 or 4) live plank rows as floats). Device pointers. */
int mocca_get_state(mocca_handle h, float *state_dev, void *stream);
int mocca_set_state(mocca_handle h, const float *state_dev, void *stream);
int mocca_get_task(mocca_handle h, uint32_t *task_dev, void *stream);
int mocca_set_task(mocca_handle h, const uint32_t *task_dev, void *stream);
int mocca_get_terrain(mocca_handle h, float *terrain_dev, void *stream);
int mocca_set_terrain(mocca_handle h, const float *terrain_dev, void *stream);

/* set_env_params / evaluation_mode / auto-reset switch (see MOCCA_PARAM_*) */
int mocca_set_param(mocca_handle h, int param_id, double value);
/* Per-env form for MOCCA_PARAM_CURRICULUM / _EVAL_MODE / _APPLIED_GAIN (each env of the reference owns its own
 * attributes, env_base.py:103-115): values_dev [N] f32 is COPIED into the handle; broadcast != 0 reads values_dev[0]
 * for every env.  A later scalar mocca_set_param of the same id drops the vector. */
int mocca_set_param_v(mocca_handle h, int param_id, const float *values_dev, int broadcast, void *stream);
/* full 64-bit Philox key (MOCCA_PARAM_SEED travels through a double: exact below 2^53 only) */
int mocca_set_seed(mocca_handle h, uint64_t seed);
/* dbg_dev [N][MOCCA_DEBUG_WORDS] i32 receives the active set of each env's last substep on every mocca_step; NULL stops it */
int mocca_set_debug_buffer(mocca_handle h, int32_t *dbg_dev);
/* Terminal observations under MOCCA_PARAM_AUTO_RESET.  The reference's step() returns the observation of the FINAL state together with
 * done (env_locomotion.py:128-141), and so does gym's TimeLimit wrapper on truncation (__init__.py:55); with auto-reset mocca_step's
 * obs_dev row of a finished env already holds the first observation of its next episode.  final_obs_dev [N][obs_dim] f32 (caller-owned,
 * NULL detaches): on every mocca_step the row of each env whose done byte is non-zero receives that final observation -- bit-identical
 * to what the same step returns with auto-reset off; rows of the other envs are left untouched. */
int mocca_set_terminal_obs_buffer(mocca_handle h, float *final_obs_dev);
/* Monitor + TimeLimitMask + the PPO loop's mask columns, inside the launch.  The reference's trainers (README.md:33-39) wrap every env in
 * baselines' Monitor (info["episode"] = {"r": return, "l": length} in the step that ends an episode) and a TimeLimitMask
 * (info["bad_transition"] when gym's TimeLimit, /root/reference/mocca_envs/__init__.py:55, cut the episode), then build `masks` / `bad_masks`
 * from `done` / `infos` on the host.  With any of the four pointers non-NULL every mocca_step also, per env (the handle keeps the running
 * return itself; mocca_reset zeroes it for the envs it resets):
 *   masks_dev     [N] f32 or NULL: 0.0 where done != 0 in this step, else 1.0
 *   bad_masks_dev [N] f32 or NULL: 0.0 where the TimeLimit bit of done is set, else 1.0
 *   totals_dev    [4] f32 or NULL: += {return, length, 1, TimeLimit bit} of every episode that ends (atomics; the caller zeroes it when it likes)
 *   records       NULL, or a ring of n_slots slots, slot_stride_bytes apart, of [N] mocca_episode_rec in DEVICE-VISIBLE memory -- device memory or
 *                 pinned host memory (hipHostMalloc / torch pin_memory: the kernel then writes the few records of a step straight into the
 *                 host's memory and nothing is copied).  The k-th mocca_step after this call (k = 1, 2, ...; mocca_episode_serial() returns
 *                 the NEXT k) writes slot k mod n_slots: record i only if env i finished in that step, with serial = k -- a record whose
 *                 serial differs is left over from n_slots steps ago.  The caller reads a slot once the step's stream work has completed.
 * All NULL detaches (synchronises the device).  The slot / serial are host state of the handle: under hipGraph replay they stay frozen,
 * masks and totals do not. */
typedef struct mocca_episode_rec {
  uint32_t serial; /* which mocca_step wrote it */
  float ret;       /* Monitor's r: sum of the episode's rewards (f32) */
  int32_t length;  /* Monitor's l: steps */
  uint32_t flags;  /* bits 0..1 the step's done byte (bit1: TimeLimit -> "bad_transition"), bits 8.. the step's info word (Stepper: steps_reached) */
} mocca_episode_rec;
int mocca_set_episode_stats(mocca_handle h, float *masks_dev, float *bad_masks_dev, float *totals_dev, void *records, int n_slots,
                            size_t slot_stride_bytes);
uint32_t mocca_episode_serial(mocca_handle h);

/* 1 if the library was compiled with a profiling switch that makes results wrong or slow by construction
 * (MOCCA_SKIP_*, MOCCA_DUMMY_VALU, MOCCA_STAMPS); the Python binding refuses such a build unless told otherwise */
int mocca_is_diagnostic_build(void);

/* The reference motion of the Cassie mocap / phase envs: what `self.traj = CassieTrajectory()` (env_cassie.py:576) holds and
 * `base_angles / base_velocities / resetJoints / get_obs` (:589-605,636-642) read through joint_angles(t), joint_speeds(t),
 * rod_joint_angles(t), max_time().  table_host [n_frames][MOCCA_TRAJ_STRIDE] f32 (HOST memory, copied into the handle):
 * 14 joint angles in ordered-joint order, 14 joint speeds, 4 rod angles (right z, right y, left z, left y).  A time t maps to
 * frame int((t mod max_time) / max_time * n_frames); t = istep * control_step / n_llc (mocap_time, :359-360), evaluated in
 * double precision.  Required before reset / step when the blob's cassie_mode != MOCCA_CASSIE_PLAIN. */
int mocca_set_trajectory(mocca_handle h, const float *table_host, int n_frames, double max_time, double control_step);

/* The terrain of the planner envs: what `self.terrain = HeightField(...); self.terrain.reload(data=filename)` builds
 * (env_locomotion.py:1011-1021, bullet_objects.py:338-393: createCollisionShape(GEOM_HEIGHTFIELD, meshScale [1/scale, 1/scale, 1]), body at
 * z = (max + min) / 2, lateralFriction 1, contactStiffness 30000, contactDamping 1000 -- the last three are blob numbers).
 * heights_host [rows][cols] f32 (HOST memory, copied into the handle; x runs along the columns), `scale` grid points per metre.  The grid is
 * centred on the origin, every cell two triangles split from (ix + 1, iy) to (ix, iy + 1); a sphere / capsule end collides with the closest
 * triangle among the 2 W x 2 W cells around the grid point nearest to its centre, W = ceil((radius + contact margin) x scale + 1/2) -- every
 * cell it can reach (W = 1 for feet and limbs at 4 points per metre, 2 for the walker's 14 cm pelvis and Mike's 23 cm waist sphere, mike.xml:20,
 * whose contact ends MikePlannerEnv's episode; a grid on which a sphere would span more than 4 cells each way is refused);
 * outside the grid there is no ground.  One grid shared by all envs of the handle
 * (the reference loads the same file for every env).  Required before reset / step / observe with MOCCA_TASK_WALKER3D_PLANNER. */
int mocca_set_heightfield(mocca_handle h, const float *heights_host, int rows, int cols, double scale);

/* registers, LDS and scratch of the step kernel as built (for DESIGN.md / bench), as the HIP runtime reports them; *sgprs = -1: the
 * runtime has no scalar-register attribute (hipFuncAttributes), the count is printed by `python -m mocca_envs_amd.build -v` */
int mocca_kernel_info(mocca_handle h, int *vgprs, int *sgprs, int *lds_bytes, int *scratch_bytes, int *max_blocks_per_cu);

const char *mocca_last_error(mocca_handle h);

#ifdef __cplusplus
}
#endif
#endif /* MOCCA_H */
