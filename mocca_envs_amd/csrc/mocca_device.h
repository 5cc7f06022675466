// mocca_device.h -- device code of the MI355X (gfx950) locomotion stepper.
//
// One 64-lane wavefront advances one environment through a whole env.step():
// 4 physics substeps (articulated-body forward dynamics, collision detection,
// contact / friction / joint-limit rows, projected Gauss-Seidel, integration)
// plus observation, reward and termination -- what the reference does through
//   robots.py:31-40   apply_action            (TORQUE_CONTROL)
//   bullet_utils.py:352-353 stepSimulation()  (inside the pybullet wheel)
//   robots.py:42-95   calc_state
//   env_locomotion.py:111-222 / :515-759      reward, termination, targets
// The dynamic state is read from HBM once, lives in LDS / VGPRs for the four
// substeps and is written back once.  No MFMA: the work is a tree recursion over
// 6-vectors and 6x6 blocks plus a sequential row solver; lanes map to bodies,
// geoms, contact candidates and constraint rows.
//
// Lane roles per phase (see DESIGN.md "Kernel anatomy"):
//   kinematics walk  lane = body       root->body walk, no cross-lane traffic
//   ABA inward pass  lane = body of the current tree level (<= 4 busy lanes)
//   ABA outward pass lane = body       root->body walk
//   collision        lane = terrain contact slot / self-collision pair (strided)
//   rows             lane = constraint row: unit-impulse response (O(n) sweep),
//                    Delassus column in registers, PGS with one readlane per row
//   integration/obs  lane = joint / obs entry
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mocca_model.h"
#include "topo_walker3d.h"
#include "topo_cassie.h"
#include "topo_walker2d.h"
#include "topo_crab2d.h"
#include "topo_laikago.h"

#define DI __device__ __forceinline__

// The device code is compiled into the library THREE times: namespace mocca (48 rows / 12 contacts per env: the product), namespace mocca_r32
// (mocca_r32.hip: MOCCA_MAXR = 32, MOCCA_COMPACT = 1 -- the compact LDS layout below; blobs with max_rows <= 32, max_contacts <= 10 and no
// loop closures) and namespace mocca_r64 (mocca_r64.hip: MOCCA_MAXR = 64 rows / 20 contacts -- every lane of the wave a row; 17 KB of LDS and
// a two-waves-per-SIMD register budget: the ACCURACY instance, for blobs whose caps exceed 48 / 12: Bullet has no cap at all), picked by
// mocca_create from the blob's caps.  Same source, same arithmetic in the same order: on the same blob the instances are bit-identical.
#ifndef MOCCA_NS
#define MOCCA_NS mocca
#endif
#ifndef MOCCA_MAXR
#define MOCCA_MAXR 48
#endif
#ifndef MOCCA_COMPACT
#define MOCCA_COMPACT 0
#endif

namespace MOCCA_NS {

// The same trees with NO link treated as massless: mocca_create() selects these instances for a blob that gives mass or inertia to the
// intermediate links of the multi-hinge joints (what a PyBullet dump may report, pybullet_dump.from_pybullet_dump) -- the ABA inward pass
// then reads every level's link inertia and bias force.  Everything else (tables, paths, levels) is inherited.
struct TopoWalker3DMassive : TopoWalker3D { static constexpr bool massless(int) { return false; } };
struct TopoCassieMassive : TopoCassie { static constexpr bool massless(int) { return false; } };

#ifdef MOCCA_STAMPS  // diagnostic build only (tools/stamps.py): raw s_memtime marks of the last substep of every wave
// Plain fire-and-forget stores, no waits and no atomics (an atomic + s_waitcnt per mark cost more than the small phases).
constexpr int STAMP_SLOTS = 32, STAMP_WAVES = 8192;
__device__ unsigned long long g_stamps[STAMP_WAVES * STAMP_SLOTS];
#define STAMP(k) do { if (lane == 0 && blockIdx.x < STAMP_WAVES) g_stamps[blockIdx.x * STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_BEGIN do {} while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_BEGIN do {} while (0)
#endif

// The model blob is read-only for every kernel: it is addressed through the constant address space, so that
// wave-uniform reads become scalar loads (SGPR results, no VALU) and per-lane reads become global loads off an SGPR
// base.  (A generic pointer -- which is what a laundered kernel argument degrades to -- makes every access a flat load:
// 64-bit address arithmetic on the VALU, a wait on both memory counters, and a v_readfirstlane for uniform values.)
#define MOCCA_AS_CONST __attribute__((address_space(4)))
typedef const MOCCA_AS_CONST MoccaModel* ModelP;
typedef float f4_t __attribute__((ext_vector_type(4)));  // native vector: loadable through any address space
typedef const MOCCA_AS_CONST f4_t* CF4P;

#ifndef MOCCA_PRIO_T3  // default row-count thresholds of the issue priorities 3 / 2 / 1 (solve_constraints; MOCCA_PARAM_ISSUE_PRIORITY)
#define MOCCA_PRIO_T3 12   // re-tuned for the blob v13 physics (5.7 rows per substep, p99 17): 28 / 20 / 14 of the 12.7-row days had stopped
#define MOCCA_PRIO_T2 7    // selecting anything -- 110.0 -> 104.3 us (profiles/archive/r03_prio_sweep_v13.txt); no priorities at all: +13 % in round 2
#define MOCCA_PRIO_T1 4
#endif
constexpr int MAXR = MOCCA_MAXR;                  // rows held by one wave (MoccaModel.max_rows must be <= MAXR)
constexpr int MAXC = MAXR >= 64 ? 20 : (MAXR >= 36 ? 12 : MAXR / 3);  // contacts (MoccaModel.max_contacts <= MAXC)
constexpr bool COMPACT = MOCCA_COMPACT != 0;
constexpr bool ALT_SWEEPS = MAXR >= 64;          // MoccaModel.sweep_alternate is honoured by the 64-row accuracy instance (solve_constraints)
constexpr int DYN_STRIDE = 96;      // floats per env in the dynamic-state buffer
constexpr int TERRAIN_STRIDE = 128; // floats per env in the terrain buffer

#ifndef MOCCA_LDS_PAD
#define MOCCA_LDS_PAD 0   // diagnostic builds only (the occupancy probe of round 3, profiles/archive/r03_occupancy_probe.jsonl): extra floats of LDS per wave, to run the same code at fewer waves per SIMD
#endif
// ---- LDS layout, float offsets (one wave = one env) ----
// [0, L_V)      survives the whole step (state, torques, new velocity, warm-start impulses)
// [L_V, end)    one region with two views: the ABA view (joint vectors, articulated inertias, geom points,
//               contacts) and the solver view (Delassus matrix A[MAXR][MAXR]; Jacobian rows parked in / behind it: every J row
//               has been read into registers before the first A entry is stored)
enum : int {
  L_Q = 0,        // [24] q, index = body
  L_QD = 24,      // [24]
  L_TAU = 48,     // [24]
  L_NU = 72,      // [28] omega(3) v(3) qd(at 5+body)
  L_BASE = 100,   // [16] pos3 quat4 vel3 omg3
  L_WARM = 116,   // [40] warm-start impulses per terrain slot
  L_FEET = 156,   // [12] feet COM xyz (NFEET x 3)
  L_PLANK = 168,  // [4][12] Stepper: frames of the live planks (rotation 9, centre 3), built once per env.step;
                  //         overlays L_JVEL / L_Q0 (Cassie-only)
  L_JVEL = 168,   // [16] Cassie: filtered joint speeds of the low-level PD loop (env_cassie.py:451-453)
  L_Q0 = 184,     // [16] Cassie: joint angles at the start of the env.step (finite-difference jvel, :467-468)
  L_V = 216,
  // One 80-byte record per body with everything the passes over the tree read together (an LDS instruction costs the CU's pipe
  // ~3.5 cycles whatever its width, and that pipe is the busiest unit of the kernel):
  //   S (6) joint motion vector about the base origin, world axes | V (6) = IA S / D | c (6) velocity-product acceleration |
  //   u / D (u = tau - S.pA; before the ABA: the net joint torque) | 1 / D, D = S.(IA S) + armature (before the ABA: the armature)
  // Row sweeps read floats 0..11 (three 16-byte reads), the ABA outward walk 0..18 (five), the inward pass S and c (four).
  SVS = 20, SV_V = 6, SV_C = 12, SV_UU = 18, SV_INVD = 19,
  // per-joint walk records, 16 floats each: [jrot * Rot(axis, q)](9) jpos(3) axis(3) qd(1), rebuilt by stage_joints() before every
  // walk IN THE BODY RECORDS (L_SV + SVS j: what those hold is dead by then, and the walk writes S only after its last record read).
  // Record 0 is the identity (the blob's joint 0), which a packed path names past its end: the walk composes a fixed number of
  // records without a branch.  At the 20-float stride the joints visited at one path position by the lanes of a wave (up to MAXW of
  // them) start in different LDS banks (16 j mod 64 put joints 9, 17 and 21 of the walker on the same banks: a 3-way conflict).
  L_B0 = L_V + 0,       // [6][8] base: rows of its articulated inertia + bias component (ABA base solve)
  L_SV = L_V + 48,      // [22][SVS]
#if MOCCA_COMPACT
  // ---- ABA view, compact (mocca_r32): what is live together is what takes space.  A substep runs  stage joints -> walk phase 1 (body
  // frames) -> geom points -> collision -> walk phase 2 (S, c, link inertias, bias forces) -> ABA -> rows -> solver: the geom points, the
  // self-collision candidate list and the body frames are dead before the first link inertia is written, and live UNDER them (phase 2
  // reads its frame into registers, then a wave barrier, then writes); the limit-row candidates are compacted after the base solve, into
  // its 6 x 8 block; the Jacobian rows start right behind the body records (the contact records they overwrite were read while the rows
  // were built -- program order of one wave).  No closure topologies here: their rows read the body frames after the ABA.
  L_ROWD = L_B0,                      // [<= 48] compacted limit-row candidates (int)
  L_CT = L_V + 488,                   // [MAXC][16] contact records
  L_P = L_CT + 16 * MAXC,             // [NB][6]  bias forces
  L_M = L_P + 132,                    // [NB][36] link / articulated inertias, full 6x6 rows (lane = row in the inward pass)
  L_RT = L_M,                         // [NB][3][4] body frames (264)
  L_GP = L_M + 264,                   // [NG][2][3] geom end points rel. base origin (<= 192: Laikago's 32 geoms)
  GP_FLOATS = 192,
  L_OBS = L_GP,
  L_CAND = L_GP + GP_FLOATS,          // [MAX_PAIRS] u16 self-collision candidates
  L_ABA_END = L_M + 792,
  // ---- solver view
  L_A = L_V,                          // [MAXR][MAXR] Delassus matrix, row = updated row, column = lane
  L_J = L_V + 488,                    // [MAXR + 1][28] Jacobian rows (+ one dummy row for lanes that own no row)
  L_XL = L_V,                         // [MAXR][28] M^-1 J^T lambda, after the iterations
  L_SOLVER_END = (L_J + 28 * (MAXR + 1) > L_A + MAXR * MAXR ? L_J + 28 * (MAXR + 1) : L_A + MAXR * MAXR),
  L_TOTAL = (L_ABA_END > L_SOLVER_END ? L_ABA_END : L_SOLVER_END) + MOCCA_LDS_PAD,
#else
  // ---- ABA view
  L_A0 = L_V + 488,     // [32] free (the Cholesky factor of IA0 and the base acceleration travel in registers: aba_passes -> solve_constraints)
  L_GP = L_V + 520,     // [NG][2][3] geom end points rel. base origin (136)
  L_OBS = L_V + 520,    // [<= 136] task layer (after the substeps: the geom points are dead): the observation is assembled here and leaves
                        //          in one coalesced store -- twice for an env that ends under auto-reset (terminal observation, then the
                        //          first observation of the next episode)
  L_CT = L_V + 656,     // [MAXC][16] contact records (192; 320 in the 64-row instance, which moves everything behind them up by 128)
  GP_FLOATS = L_CT - L_GP,
  L_ROWD = L_CT + 16 * MAXC,   // (L_V + 848) [48] compacted limit-row candidates (int)
  L_RT = L_ROWD + 48,   // (L_V + 896) [NB][3][4] body frames: row i of the rotation (3) + component i of the origin rel. the base origin: one 16-byte
                        //            write per (body, row) lane of the walk, three 16-byte reads per consumer
  L_M = L_RT + 264,     // (L_V + 1160) [NB][36] link / articulated inertias, full 6x6 rows (lane = row in the inward pass)
  L_P = L_M + 792,      // (L_V + 1952) [NB][6]  bias forces
  L_ABA_END = L_P + 132,   // (L_V + 2084)
  L_CAND = L_ABA_END,   // [MAX_PAIRS] u16 self-collision candidates (collide only: the slack the joint records 9.. used during the walk)
  // ---- solver view
  L_A = L_V,            // [MAXR][MAXR] Delassus matrix, row = updated row, column = lane
  L_J = (L_V + 960 > L_RT ? L_V + 960 : L_RT),   // [MAXR][28] Jacobian rows (tail of the A region; never below the body frames, see the assert)
  L_XL = L_V,           // [MAXR][28] M^-1 J^T lambda, after the iterations
  L_TOTAL = L_V + MAXR * MAXR + 28 + MOCCA_LDS_PAD,  // + one dummy J row for lanes that own no row (A-row prefetches clamp to row MAXR - 1)
#endif
};
static_assert(L_ABA_END <= L_TOTAL, "ABA view must fit");
static_assert(L_SV % 4 == 0 && SVS % 4 == 0 && SVS >= 16, "joint records live in the body records, 16-byte aligned");
static_assert(L_J + (MAXR + 1) * 28 <= L_TOTAL, "Jacobian rows (+ dummy) must fit the tail of the A region");
static_assert(L_J % 4 == 0 && L_V % 4 == 0 && L_RT % 4 == 0 && L_CT % 4 == 0, "16-byte alignment of broadcast rows");
static_assert(MAXR % 2 == 0 && 3 * MAXC <= MAXR, "friction rows sit on the top 2 MAXC lanes of the row range, odd lane = second tangent");
static_assert(L_PLANK + 12 * MOCCA_MAX_PLANKS <= L_V && L_Q0 + 16 <= L_V, "persistent region overflows into the two-view region");
#if MOCCA_COMPACT
static_assert(L_TOTAL * 4 <= 8192 || MOCCA_LDS_PAD > 0, "more than 8 KB of LDS per wave: fewer than 20 waves per CU (5 per SIMD)");
static_assert(L_J >= L_SV + 22 * SVS, "J rows are written while S, V, 1/D are still being read");
static_assert(L_CAND + (MOCCA_MAX_PAIRS + 1) / 2 <= L_ABA_END && L_RT + 264 <= L_GP, "frames, geom points and the candidate list live under the link inertias");
#else
static_assert(L_TOTAL * 4 <= 10240 || MOCCA_LDS_PAD > 0 || MAXR > 48, "more than 10 KB of LDS per wave: fewer than 16 waves per CU, the 4096-env batch no longer fits one round");
static_assert(MAXR != 48 || (L_ROWD == L_V + 848 && L_RT == L_V + 896 && L_M == L_V + 1160 && L_P == L_V + 1952 && L_ABA_END == L_V + 2084 && L_J == L_V + 960), "the product's layout is unchanged");
static_assert(COMPACT || L_CAND + (MOCCA_MAX_PAIRS + 1) / 2 <= L_TOTAL, "candidate list behind the ABA view");
static_assert(L_J >= L_RT, "J rows may be written while S, U, 1/D, the factor of IA0 and the contacts are still being read");
#endif

// contact record fields
enum : int { C_BA = 0, C_BB = 1, C_SLOT = 2, C_P = 3, C_N = 6, C_DEPTH = 9, C_MU = 10, C_ERP = 11, C_CFM = 12, C_MA = 13, C_MB = 14 };

// task record words (include/mocca_model.h)
enum : int { T_WTX = 0, T_WTY, T_WTZ, T_LINPOT, T_ANGPOT, T_CLOSE, T_STOPF, T_DONE, T_T, T_EPISODE, T_DRAW, T_MIRROR,
             T_FC0, T_FC1, T_DIST, T_ANGLE, T_NSI, T_TRC, T_STOP, T_SETSTOP, T_CUR, T_GAIN, T_PREVX, T_RES23,
             T_JVEL = 24, T_FC2 = 24, T_FC3 = 25 /* quadrupeds; Cassie's jvel otherwise */, T_RW = 30 /* Stepper: 8 reward weights */,
             T_INITZ = 38, T_ISTEP = 39 };

struct StepArgs {
  const MoccaModel* model;
  float* dyn;        // [N][DYN_STRIDE]
  uint32_t* task;    // [N][MOCCA_TASK_WORDS]
  float* terrain;    // [N][TERRAIN_STRIDE] (stepper)
  const float* act;  // [N][NJ]
  float* obs;        // [N][obs_dim]
  float* rew;        // [N]
  uint8_t* done;     // [N]
  int32_t* info;     // [N] or null
  const uint8_t* mask;  // reset only
  int n_envs;
  int obs_dim;
  int auto_reset;
  int eval_mode;
  int random_pose;
  int curriculum;    // applied at reset (stepper)
  int host_retarget; // Custom: the host re-randomises the walk target
  int env_offset;    // global id of env 0 (RNG key)
  uint32_t seed_lo, seed_hi;
  // per-env parameter vectors (mocca_set_param_v); null = the handle-wide scalar above
  const float* curriculum_v;   // [N]
  const float* eval_mode_v;    // [N]
  int random_reward;           // Stepper, env_locomotion.py:533-547: 0 off, 1 weights drawn in the kernel, 2 weights supplied in the task record
  const float* gain_v;         // [N] robot.applied_gain of the Custom envs (set_robot_params); the Stepper derives it from the curriculum
  float gain;
  // mocca_task_step (INJECT kernels): contact query results supplied by the caller instead of the physics
  const int32_t* inj_touch;    // [N][NFEET] foot k touches the terrain
  const int32_t* inj_target;   // [N][NFEET] foot k touches the cover of the target plank (or null)
  const int32_t* inj_body;     // [N] a non-foot link touches the terrain (or null)
  // INJECT kernels only: uniforms that replace the Philox draws, indexed by the episode's draw counter (the golden
  // replays feed the very numbers the reference's numpy RandomState produced)
  const float* tape;           // [N][tape_n] or null
  int tape_n;
  // optional per-env debug record of the LAST substep (mocca_set_debug_buffer): [N][MOCCA_DEBUG_WORDS] or null
  int32_t* dbg;
  // Cassie mocap / phase envs (mocca_set_trajectory): [traj_n][MOCCA_TRAJ_STRIDE] angles 14, speeds 14, rod angles 4
  const float* traj;
  int traj_n;
  double traj_tmax, traj_cstep;   // CassieTrajectory.max_time(); control_step (mocap_time = istep * control_step / n_llc, in f64)
  int prio;   // MOCCA_PARAM_ISSUE_PRIORITY: row-count thresholds of the issue priorities 1 / 2 / 3, 6 bits each
  // planner envs (mocca_set_heightfield): heights[hf_rows][hf_cols], x along the columns, hf_scale grid points per metre; shared by all envs
  const float* hf;
  int hf_rows, hf_cols;
  float hf_scale;
  // optional (mocca_set_terminal_obs_buffer): [N][obs_dim]; the row of an env that ends under auto-reset receives the observation of
  // its final state (what the reference's step() returns with done, env_locomotion.py:128-141) before `obs` gets the next episode's first
  float* final_obs;
  // the last substep's normal impulses per terrain slot (state words 13 + 2 NJ ..) are stored although the blob does not warm-start
  // (MOCCA_PARAM_PERSIST_IMPULSES); a blob with warmstart != 0 always loads and stores them
  int persist_warm;
  // optional (MOCCA_PARAM_ORDER_EVERY): workgroup b advances env order[b] -- a permutation of 0 .. n_envs - 1, heaviest envs (most constraint
  // rows at the end of the step before) first; null = identity.  Envs never interact: the order decides WHEN an env runs, not what it computes
  const int32_t* order;
  // MOCCA_PARAM_PACE_TICKS (timing only): > 0 replaces the row-count issue priorities by PACE priorities -- a wave that is behind the pace
  // of `pace` shader-clock ticks per env.step (its own elapsed time against the fraction of the step it has done) raises its issue
  // priority, one that is ahead lowers it, so that the waves of a SIMD finish together instead of oldest-first
  int pace;
  // pace < 0: self-calibrating pace = (-pace / 16) x the mean wave time of the last 64 .. 128 sampled waves (one wave in 61 adds its elapsed
  // ticks at its end).  One 64-bit word in device memory, {sum of ticks / 64 : 40 bits | samples : 24 bits}, added to with ONE atomic; the wave
  // whose add makes the count reach 128 takes half of what it saw away again (pace_finish).  No host state per launch: a captured
  // mocca_step replays with a live pace.
  unsigned long long* pace_acc;
  // Monitor / TimeLimitMask inside the launch (mocca_set_episode_stats; all optional, ep_ret == null: off).  ep_ret [N]: the running return of
  // each env's episode (handle-owned); ep_masks / ep_bad [N]: 0.0 where the episode ended in this step / ended with the TimeLimit bit set, else
  // 1.0; ep_rec [N] x 16 bytes: {serial, return, length, done bits | info << 8} of the envs that finished, this launch's slot of the caller's
  // ring (device-visible memory, normally pinned host memory: the kernel writes the few records straight across PCIe); ep_totals [4]: sums of
  // return / length / episodes / truncated episodes (atomics) for a trainer that never leaves the device
  float* ep_ret;
  float* ep_masks;
  float* ep_bad;
  void* ep_rec;
  float* ep_totals;
  uint32_t ep_serial;
};

// ------------------------------------------------------------------ helpers
DI void wsync() {
  // one wave per env: LDS traffic of a wave is processed in issue order, so a compiler-level
  // fence is all that is needed between a phase that writes LDS and one that reads other lanes' data
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DI float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DI void cross3(const float* a, const float* b, float* o) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
DI float dot6(const float* a, const float* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
DI void matvec3(const float* R, const float* x, float* o) {
  float a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  float b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  float c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
DI void matmul3(const float* A, const float* B, float* C) {
  float T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
DI void quat_to_mat(const float* q, float* R) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
DI void crm(const float* v, const float* m, float* o) {  // spatial motion cross product
  float a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
  o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
DI void crf(const float* v, const float* f, float* o) {  // spatial force cross product
  float a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
  o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
// symmetric 6x6 stored as 21 floats, row-major upper triangle
DI constexpr int sym(int i, int j) { return i <= j ? i * 6 - i * (i - 1) / 2 + (j - i) : j * 6 - j * (j - 1) / 2 + (i - j); }
DI void symmv6(const float* A, const float* x, float* o) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += A[sym(i, j)] * x[j];
    o[i] = s;
  }
}
// the same for a spatial inertia [[Ic, m c^x], [(m c^x)^T, m 1]]: its twelve structural zeros are skipped (same order of the remaining terms;
// without fast-math the optimiser keeps every `+ 0 * x`)
DI void spatial_inertia_mv(const float* A, const float* x, float* o) {
  constexpr unsigned long long Z = (1ull << 3) | (1ull << 10) | (1ull << 17) | (1ull << 18) | (1ull << 22) | (1ull << 23) |
                                   (1ull << 25) | (1ull << 27) | (1ull << 29) | (1ull << 32) | (1ull << 33) | (1ull << 34);   // bit 6 i + j
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (!((Z >> (6 * i + j)) & 1ull)) s += A[sym(i, j)] * x[j];
    o[i] = s;
  }
}
DI float rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp
DI float rsq(float x) { return __builtin_amdgcn_rsqf(x); }      // v_rsq_f32
// Pin values at a program point.  SelectionDAG linearises un-chained ALU nodes freely inside a basic block, so
// in the fully unrolled sweeps it postpones each body's arithmetic until every body's LDS loads were issued
// (all 21 bodies' S/U live: ~270 VGPRs).  An empty volatile asm is chained, which keeps bodies in order.
DI void pin6(float* v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); }
DI void pin1(float& v) { asm volatile("" : "+v"(v)); }
template <int CTRL>
DI float dpp_mov(float v) {  // bound_ctrl: lanes without a source read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each aligned group of 8 lanes, result in all 8: quad butterfly (quad_perm) then half-row mirror
DI float group8_sum(float t) {
  t += dpp_mov<0xB1>(t);   // quad_perm [1,0,3,2]
  t += dpp_mov<0x4E>(t);   // quad_perm [2,3,0,1]
  t += dpp_mov<0x141>(t);  // row_half_mirror
  return t;
}
// the same for two values at once, as six v_add_f32_dpp: the two chains cover each other's DPP read hazard (the optimiser emitted the first
// step of each as v_mov_b32_dpp + v_fmac and part of the last ones as moves + adds)
DI void group8_sum2(float& a, float& b) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b));
}
DI int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }  // assert wave-uniformity: value moves to an SGPR
DI float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
DI float readlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
DI int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <int LANE>
DI float writelane_c(float v, float old) {  // immediate lane select
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(LANE));
  return old;
}
// old with lane LANE replaced by that lane's v: one v_cndmask on a constant lane mask (SALU moves) -- the v_readlane + v_writelane pair
// that did the same took two VALU issues
template <int LANE>
DI float commit_lane(float v, float old) {
  float r;
  unsigned long long m;  // built next to its use: as an operand the 60 masks of the unrolled solver were hoisted to kernel entry and spilled
  // (s_mov / s_bitset1 leave SCC alone: the visits sit between the s_cmp and the s_cbranch of the solver's uniform exit tests)
  if constexpr (LANE < 31)
    asm("s_mov_b64 %1, %4\n\tv_cndmask_b32 %0, %2, %3, %1" : "=v"(r), "=&s"(m) : "v"(old), "v"(v), "n"(1u << LANE));
  else
    asm("s_mov_b64 %1, 0\n\ts_bitset1_b64 %1, %4\n\tv_cndmask_b32 %0, %2, %3, %1" : "=v"(r), "=&s"(m) : "v"(old), "v"(v), "n"(LANE));
  return r;
}
DI int lane_rank(unsigned long long mask) {  // number of set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
// OR over the wave, result uniform (SGPR): DPP butterflies inside each row of 16, then one readlane per row --
// no LDS crossbar round trips (six dependent ds_bpermute cost ~900 cycles)
template <int CTRL>
DI unsigned dpp_or(unsigned v) { return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
DI unsigned wave_or(unsigned v) {
  v = dpp_or<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_or<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_or<0x141>(v);  // row_half_mirror
  v = dpp_or<0x140>(v);  // row_mirror: every lane now holds its row's OR
  return (unsigned)(readlane_i((int)v, 0) | readlane_i((int)v, 16) | readlane_i((int)v, 32) | readlane_i((int)v, 48));
}
// sum over the wave, result uniform: DPP butterflies inside each row of 16, then one readlane per row (six dependent ds_bpermute round
// trips before -- on the reward path every wave runs after its last substep)
DI float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror: every lane now holds its row's sum
  return (readlane(v, 0) + readlane(v, 16)) + (readlane(v, 32) + readlane(v, 48));
}

// max over the wave, result uniform (same DPP pattern as wave_sum: every step is a full permutation inside a row of 16)
DI float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(readlane(v, 0), readlane(v, 16)), fmaxf(readlane(v, 32), readlane(v, 48)));
}

// Philox4x32-10; identical to oracle/mocca_oracle.c so device resets are reproducible on the host
DI void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
DI float rng_uniform(uint32_t slo, uint32_t shi, uint32_t env, uint32_t episode, uint32_t d) {
  uint32_t o[4];
  philox4x32(d >> 2, episode, env, 0u, slo, shi, o);
  uint32_t w = (d & 3) == 0 ? o[0] : (d & 3) == 1 ? o[1] : (d & 3) == 2 ? o[2] : o[3];
  return (float)(w >> 8) * (1.0f / 16777216.0f);
}

// d-th uniform of (env, episode): Philox, or -- in the INJECT kernels, when a tape is attached -- tape[d]
template <bool INJECT>
DI float draw_u(const StepArgs& a, int env, int episode, int d) {  // env = global id (env_offset + index in this handle)
  if constexpr (INJECT) {
    if (a.tape) return d < a.tape_n ? a.tape[(size_t)(env - a.env_offset) * a.tape_n + d] : 0.5f;
  }
  return rng_uniform(a.seed_lo, a.seed_hi, (uint32_t)env, (uint32_t)episode, (uint32_t)d);
}

// ------------------------------------------------------------------ kinematics
// Kinematics in two phases.
//  1. lane = (body, row): three lanes per body, each composing ONE ROW of the body's rotation along the root -> body path
//     (row' = row * [jrot Rot(axis, q)], 9 FMA per step instead of 27, the offset r_i = row . jpos one component each).
//     The rows of a rotation chain never mix, so the three lanes need no exchange.  21 bodies x 3 = 63 lanes.
//     Each group also leaves its joint's world axis a = R_body * axis (component i on lane i).
//  2. lane = body: joint motion vector S = (a, r x a), then -- FULL only -- the spatial velocity as a plain sum of
//     S_j qd_j along the path (no rotation chain any more), the velocity-product acceleration, link inertia, bias force.
// One lane per body for everything cost ~90 VALU per path step; this costs ~20 + ~12.
// the body's own constants of the walk's second phase
struct WalkConsts { float cl[3], inl[6], ms, jarm, jdamp; };
template <bool FULL>
DI void walk_consts(ModelP M, int b, WalkConsts& k) {
#pragma unroll
  for (int i = 0; i < 3; ++i) k.cl[i] = M->com[b][i];
#pragma unroll
  for (int i = 0; i < 6; ++i) k.inl[i] = 0.0f;
  k.ms = 0; k.jarm = 0; k.jdamp = 0;
  if (FULL) {
#pragma unroll
    for (int i = 0; i < 6; ++i) k.inl[i] = M->inertia[b][i];
    k.ms = M->mass[b]; k.jarm = M->jarm[b]; k.jdamp = M->jdamp[b];
  }
}
// ---- phase 1: body frames (L_RT) and world joint axes (L_SV + SVS b + 0..2)
template <class T>
DI void walk_phase1(float* L, int lane, unsigned long long ppk) {
  static_assert(3 * (T::NB - 1) <= 63, "three lanes per body must fit the wave (lane 63 writes the base)");
  {
    const int g = (lane * 43) >> 7;          // lane / 3 for lane < 64
    const int ri = lane - 3 * g;             // row of the rotation this lane owns
    const int bg = g + 1 < T::NB ? g + 1 : 0;  // the group's body (0 = idle group)
    // the group's packed path: fetched from the lane that owns body bg in the lane = body layout
    const unsigned plo = (unsigned)__shfl((int)(unsigned)ppk, bg, 64), phi = (unsigned)__shfl((int)(unsigned)(ppk >> 32), bg, 64);
    const unsigned long long pk3 = ((unsigned long long)phi << 32) | plo;
    float Rb[9];
    {
      float q[4] = {L[L_BASE + 3], L[L_BASE + 4], L[L_BASE + 5], L[L_BASE + 6]};
      quat_to_mat(q, Rb);
    }
    float row[3] = {ri == 0 ? Rb[0] : ri == 1 ? Rb[3] : Rb[6], ri == 0 ? Rb[1] : ri == 1 ? Rb[4] : Rb[7], ri == 0 ? Rb[2] : ri == 1 ? Rb[5] : Rb[8]};
    float rr = 0.0f;
#pragma unroll
    for (int k = 0; k < T::MAXD; ++k) {
      const int j = (int)((pk3 >> (5 * k)) & 31ull);  // 0 past the end of the path: the identity record
      // the joint's record was staged by stage_joints(): 16-byte LDS reads, none of them on the dependent chain
      const float4* jr = reinterpret_cast<const float4*>(L + L_SV + SVS * j);
      const float4 r0 = jr[0], r1 = jr[1], r2 = jr[2];
      rr += row[0] * r2.y + row[1] * r2.z + row[2] * r2.w;  // offset of the joint in the parent frame
      const float n0 = row[0] * r0.x + row[1] * r0.w + row[2] * r1.z;
      const float n1 = row[0] * r0.y + row[1] * r1.x + row[2] * r1.w;
      const float n2 = row[0] * r0.z + row[1] * r1.y + row[2] * r2.x;
      row[0] = n0; row[1] = n1; row[2] = n2;
    }
    if (bg > 0) {
      const float4 r3 = reinterpret_cast<const float4*>(L + L_SV + SVS * bg)[3];
      *reinterpret_cast<float4*>(L + L_RT + 12 * bg + 4 * ri) = make_float4(row[0], row[1], row[2], rr);
      L[L_SV + SVS * bg + ri] = row[0] * r3.x + row[1] * r3.y + row[2] * r3.z;  // world axis: Rot(axis, q) leaves the axis in place
    }
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(L + L_RT + 4 * i) = make_float4(Rb[3 * i], Rb[3 * i + 1], Rb[3 * i + 2], 0.0f);
    }
  }
}
// ---- phase 2: lane = body (see above); in the compact layout the link inertias it writes overlay the body frames, the geom points and
// the candidate list: every lane holds its frame in registers before the first store (wave barrier inside `if (FULL)`)
template <class T, bool FULL>
DI void walk_phase2(ModelP M, float* L, int lane, int b, unsigned long long ppk, const WalkConsts& wk) {
  const float cl[3] = {wk.cl[0], wk.cl[1], wk.cl[2]};
  const float inl[6] = {wk.inl[0], wk.inl[1], wk.inl[2], wk.inl[3], wk.inl[4], wk.inl[5]};
  const float ms = wk.ms, jarm = wk.jarm, jdamp = wk.jdamp;
  float R[9], r[3], v[6], S[6] = {0, 0, 0, 0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 t = *reinterpret_cast<const float4*>(L + L_RT + 12 * b + 4 * i);
    R[3 * i] = t.x; R[3 * i + 1] = t.y; R[3 * i + 2] = t.z; r[i] = t.w;
  }
  if (FULL) {
    if (lane >= 1 && lane < T::NB) {
      float a[3] = {L[L_SV + SVS * b], L[L_SV + SVS * b + 1], L[L_SV + SVS * b + 2]}, ra[3];
      cross3(r, a, ra);
#pragma unroll
      for (int i = 0; i < 3; ++i) { S[i] = a[i]; S[3 + i] = ra[i]; L[L_SV + SVS * b + 3 + i] = ra[i]; }
    }
    wsync();
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k] = L[L_BASE + 10 + k]; v[3 + k] = L[L_BASE + 7 + k]; }
#pragma unroll
    for (int k = 0; k < T::MAXD; ++k) {
      const int j = (int)((ppk >> (5 * k)) & 31ull);  // the lane's packed path: no table access
      if (j != 0 && j != b) {  // ancestors; the body's own joint follows below
        const float qd = L[L_QD + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] += L[L_SV + SVS * j + i] * qd;
      }
    }
    if (lane >= 1 && lane < T::NB) {
      float vJ[6];
      const float qd = L[L_QD + b];
#pragma unroll
      for (int i = 0; i < 6; ++i) vJ[i] = S[i] * qd;
      crm(v, vJ, c);
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] += vJ[i];
    }
  }
  if (lane < T::NB) {
    float cw[3];
    matvec3(R, cl, cw);
#pragma unroll
    for (int i = 0; i < 3; ++i) cw[i] += r[i];
    // COM of each foot LINK (getLinkState[0], bullet_utils.py:106) for the observation: only the walk that precedes an observation
    // (FULL = false) needs it -- inside the substeps nothing reads L_FEET
#pragma unroll
    for (int f = 0; f < (FULL ? 0 : T::NFEET); ++f)
      if (b == M->foot_body[f]) {
        const float fp[3] = {M->foot_point[f][0], M->foot_point[f][1], M->foot_point[f][2]};  // uniform: scalar loads
        float fw[3];
        matvec3(R, fp, fw);
#pragma unroll
        for (int i = 0; i < 3; ++i) L[L_FEET + 3 * f + i] = fw[i] + r[i] + L[L_BASE + i];
      }
    if (FULL) {
#pragma unroll
      for (int i = 0; i < 6; ++i) L[L_SV + SVS * b + SV_C + i] = c[i];  // (S of the base is never read)
      // spatial inertia about the base origin, world axes
      const float ixx = inl[0], iyy = inl[1], izz = inl[2], ixy = inl[3], ixz = inl[4], iyz = inl[5];
      float Il[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz}, Tm[9], Iw[9];
      matmul3(R, Il, Tm);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jx = i; jx < 3; ++jx)
          Iw[3 * i + jx] = Tm[3 * i] * R[3 * jx] + Tm[3 * i + 1] * R[3 * jx + 1] + Tm[3 * i + 2] * R[3 * jx + 2];
      const float cc = dot3(cw, cw);
      float I[21];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jx = i; jx < 3; ++jx) I[sym(i, jx)] = Iw[3 * i + jx] + ms * ((i == jx ? cc : 0.0f) - cw[i] * cw[jx]);
      // upper-right block m [c]x
      I[sym(0, 3)] = 0;           I[sym(0, 4)] = -ms * cw[2]; I[sym(0, 5)] = ms * cw[1];
      I[sym(1, 3)] = ms * cw[2];  I[sym(1, 4)] = 0;           I[sym(1, 5)] = -ms * cw[0];
      I[sym(2, 3)] = -ms * cw[1]; I[sym(2, 4)] = ms * cw[0];  I[sym(2, 5)] = 0;
      I[sym(3, 3)] = ms; I[sym(3, 4)] = 0; I[sym(3, 5)] = 0; I[sym(4, 4)] = ms; I[sym(4, 5)] = 0; I[sym(5, 5)] = ms;
      float Iv[6], p[6];
      spatial_inertia_mv(I, v, Iv);
      crf(v, Iv, p);
      // gravity through the COM
      const float fz = -M->gravity * ms;
      p[0] -= cw[1] * fz; p[1] -= -cw[0] * fz; p[5] -= fz;
      {  // link damping of btMultiBody, base and every link [UNVERIFIED-BULLET], see oracle aba(): force m vc (k + k |vc|) through the
         // COM, torque Ic w (k + k |w|)
        float om[3] = {v[0], v[1], v[2]}, Iom[3], wxc[3], F[3], cxF[3];
        float Iwf[9] = {Iw[0], Iw[1], Iw[2], Iw[1], Iw[4], Iw[5], Iw[2], Iw[5], Iw[8]};
        matvec3(Iwf, om, Iom);
        cross3(om, cw, wxc);
        float vc[3] = {v[3] + wxc[0], v[4] + wxc[1], v[5] + wxc[2]};
        const float kl = unif(M->lin_damp) * ms * (1.0f + __builtin_sqrtf(dot3(vc, vc)));
        const float ka = unif(M->ang_damp) * (1.0f + __builtin_sqrtf(dot3(om, om)));
#pragma unroll
        for (int i = 0; i < 3; ++i) F[i] = kl * vc[i];
        cross3(cw, F, cxF);
#pragma unroll
        for (int i = 0; i < 3; ++i) { p[i] += ka * Iom[i] + cxF[i]; p[3 + i] += F[i]; }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jx = 0; jx < 6; ++jx) L[L_M + 36 * b + 6 * i + jx] = I[sym(i, jx)];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[L_P + 6 * b + i] = p[i];
      // staged for the inward pass (which overwrites both slots with 1/D and u): joint armature and the net joint
      // torque, so that the level loop reads LDS only -- its global loads were hoisted above all levels and spilled
      L[L_SV + SVS * b + SV_INVD] = jarm;
      L[L_SV + SVS * b + SV_UU] = L[L_TAU + b] - jdamp * L[L_QD + b];
    }
  }
}

template <class T, bool FULL>
DI void walk_kinematics(ModelP M, float* L, int lane, unsigned long long ppk) {
  asm volatile("" : "+v"(lane));  // lane-derived indices are recomputed per walk: CSE across walks kept them live from kernel entry (spilled)
  const int b = lane < T::NB ? lane : 0;
  WalkConsts wk;
  walk_consts<FULL>(M, b, wk);   // fetched before the walk so that their latency hides behind it
  walk_phase1<T>(L, lane, ppk);
  wsync();
  walk_phase2<T, FULL>(M, L, lane, b, ppk, wk);
}

// 6x6 SPD system through its Cholesky factor, symmetric storage, all indices static: F[sym(i, j)] = L_ij (i > j),
// F[sym(j, j)] = 1 / L_jj.  The factor is what is kept (not the explicit inverse): factor + two triangular solves are a
// third of the instructions of factor + inversion + product, and every consumer only ever applies the inverse to a vector.
// v_rsq (1 ulp) instead of IEEE sqrt/div.
DI void chol6_factor(const float* A, float* F) {
  float Lm[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float s = A[sym(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= Lm[j][k] * Lm[j][k];
    const float id = rsq(s);
    F[sym(j, j)] = id;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float t = A[sym(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lm[i][k] * Lm[j][k];
      Lm[i][j] = t * id;
      F[sym(i, j)] = Lm[i][j];
    }
  }
}
DI void chol6_solve(const float* F, const float* b, float* x) {
  float y[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float t = b[j];
#pragma unroll
    for (int k = 0; k < j; ++k) t -= F[sym(j, k)] * y[k];
    y[j] = t * F[sym(j, j)];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float t = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) t -= F[sym(k, i)] * x[k];
    x[i] = t * F[sym(i, i)];
  }
}

// ABA inward pass (lane = body of the current level) + base solve + outward pass (lane = body).
// Leaves S, V = U / D, 1/D, u / D in LDS and the factor of IA0 in Afac (scalar registers) for the row sweeps, and the new generalised
// velocity in L_NU.
template <class T>
DI void aba_passes(ModelP M, float* L, int lane, unsigned long long ppk, float* Afac) {
  STAMP_BEGIN;
  // ---- inward pass, one tree level at a time.  8 lanes per body (lane i < 6 owns row i of the 6x6
  // articulated inertia), up to MAXW = 4 bodies per level: ~45 VALU per level instead of ~260 with one
  // lane per body.  Children are pulled (summed in descending index order, as the oracle does).
  static_assert(T::MAXW * 8 <= 64, "level does not fit a wave");
  static_assert(T::MAXW == 4 && T::MAXCH <= 4, "level pass is written for <= 4 bodies per level, <= 4 children");
  const int s = lane >> 3, i = lane & 7;
  const int ii = i < 6 ? i : 0;
  // A body sits in the slot of its "carried" child (topo_*.h: clevel / ccarry), so along a serial chain the articulated
  // inertia row and bias of the child stay in registers (crow, cpA) and a level costs no LDS round trip: the level's
  // own link data has static addresses and nothing on the chain waits for it.  Only children in another slot (the
  // second leg at the pelvis, the limbs at the base) travel through LDS.
  float crow[6] = {0, 0, 0, 0, 0, 0}, cpA = 0.0f;
  // LDS offsets of the slot's body, carried from level to level: along a chain the body index drops by one per level, so most
  // levels update the four offsets with one subtraction each instead of rebuilding them from a per-slot select (12 integer
  // instructions per level).  Lanes of an empty slot point at body d (any existing body; they never store).
  int bb = 0, o6i = 0, o36 = 0, osv = 0, osvi = 0;
#pragma unroll
  for (int d = T::MAXD; d >= 1; --d) {
    // bodies of this level and their children are compile-time constants selected by the lane's slot
#define MOCCA_CONT(sl) (T::clevel(d, sl) < 0 || (d < T::MAXD && T::clevel(d + 1 <= T::MAXD ? d + 1 : d, sl) == T::clevel(d, sl) + 1))
    const bool sv = s == 0 ? T::clevel(d, 0) >= 0 : s == 1 ? T::clevel(d, 1) >= 0 : s == 2 ? T::clevel(d, 2) >= 0 : s == 3 ? T::clevel(d, 3) >= 0 : false;
    const bool valid = sv && i < 6;
    if (d < T::MAXD && MOCCA_CONT(0) && MOCCA_CONT(1) && MOCCA_CONT(2) && MOCCA_CONT(3)) {  // compile-time: every body of the level continues its slot's chain
      bb -= 1; o6i -= 6; o36 -= 36; osv -= SVS; osvi -= SVS;
    } else {
      const int b = s == 0 ? T::clevel(d, 0) : s == 1 ? T::clevel(d, 1) : s == 2 ? T::clevel(d, 2) : s == 3 ? T::clevel(d, 3) : -1;
      bb = b >= 0 ? b : d;
      o6i = 6 * bb + ii; o36 = 36 * bb + 6 * ii; osv = SVS * bb; osvi = osv + ii;
    }
#undef MOCCA_CONT
    float row[6], S[6], c[6], pAi;
#pragma unroll
    for (int j = 0; j < 6; ++j) { S[j] = L[L_SV + osv + j]; c[j] = L[L_SV + osv + SV_C + j]; }
    // a level that holds only massless links (the intermediate links of the multi-hinge joints: two of the walker's eight levels)
    // has nothing to read: its link inertias and bias forces are zero (mocca_create() checks the blob against T::massless)
#define MOCCA_NOMASS(sl) (T::clevel(d, sl) < 0 || T::massless(T::clevel(d, sl)))
    if (MOCCA_NOMASS(0) && MOCCA_NOMASS(1) && MOCCA_NOMASS(2) && MOCCA_NOMASS(3)) {   // compile-time
#pragma unroll
      for (int j = 0; j < 6; ++j) row[j] = crow[j];
      pAi = cpA;
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) row[j] = L[L_M + o36 + j] + crow[j];
      pAi = L[L_P + o6i] + cpA;
    }
#undef MOCCA_NOMASS
#pragma unroll
    for (int k = 0; k < T::MAXCH; ++k) {
      // the k-th child of the slot's body, unless it is the carried one
#define MOCCA_LDS_CHILD(sl) (T::cchild(T::clevel(d, sl), k) != T::ccarry(T::clevel(d, sl)) ? T::cchild(T::clevel(d, sl), k) : -1)
      const int ch = s == 0 ? MOCCA_LDS_CHILD(0) : s == 1 ? MOCCA_LDS_CHILD(1) : s == 2 ? MOCCA_LDS_CHILD(2) : s == 3 ? MOCCA_LDS_CHILD(3) : -1;
      if (MOCCA_LDS_CHILD(0) >= 0 || MOCCA_LDS_CHILD(1) >= 0 || MOCCA_LDS_CHILD(2) >= 0 || MOCCA_LDS_CHILD(3) >= 0) {  // compile-time
        if (ch >= 0) {
#pragma unroll
          for (int j = 0; j < 6; ++j) row[j] += L[L_M + 36 * ch + 6 * ii + j];
          pAi += L[L_P + 6 * ch + ii];
        }
      }
#undef MOCCA_LDS_CHILD
    }
    // Branch-free on purpose: idle lanes (rows 6, 7 of a group, empty slots) run the same arithmetic on harmless data
    // and are kept out of the sums / the stores only.  Under `valid ? ... : 0` the compiler sank the LDS reads into
    // conditional blocks, each with its own wait, and nothing of the next level could be fetched ahead.
    const float Si = i < 6 ? L[L_SV + osvi] : 0.0f;  // the lane's own component of S (0 for the two idle lanes)
    const float Ui = dot6(row, S);
    float dsum = Si * Ui, psum = Si * pAi;
    group8_sum2(dsum, psum);
    float arm = L[L_SV + osv + SV_INVD], unet = L[L_SV + osv + SV_UU];  // staged by the walk: joint armature, net joint torque
    pin1(arm); pin1(unet);                           // fetched with the level's other reads, not inside the store branch
    const float id = rcp(dsum + arm);
    const float u = unet - psum;
    // row_i -= (U_i / D) U_j for the six columns j.  The six U_j are fetched by DPP, not through LDS: within the 8-lane group the
    // first quad holds U0..U3 and the second U4..U7; a shift by four lanes inside the row of 16 brings the other quad's four
    // values in ORDER, so LO / HI hold U0..U3 / U4..U7 on quad lanes 0..3 of EVERY lane, and column j is one fused multiply-add
    // whose second factor is a quad broadcast of LO or HI (2 shifts + 2 selects + 6 broadcasts per level instead of 9 moves + 6 selects).
    const float sh_dn = dpp_mov<0x114>(Ui);   // row_shr:4  lane l <- l - 4
    const float sh_up = dpp_mov<0x104>(Ui);   // row_shl:4  lane l <- l + 4
    const bool first = i < 4;
    const float LO = first ? Ui : sh_dn, HI = first ? sh_up : Ui;
    const float uid = Ui * id;
    float Iac = 0.0f;
    // (a hand-written v_fmac_f32_dpp per column saves the six quad-broadcast moves but needs its own s_nop for the DPP read hazard
    // and measured the same: the compiler-scheduled form stays)
    const float nuid = -uid;   // negated once: with a plain multiplicand the six updates are v_fmac_f32 (VOP2), which takes the quad broadcast as a DPP source
    // the optimiser keeps the quad broadcast as its own v_mov_b32_dpp per column; one block with the DPP read hazard (two wait states after
    // the selects that wrote LO / HI) paid once saves those six moves per level
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %6, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %6, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %6, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %6, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %7, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %7, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
        : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]), "+v"(row[4]), "+v"(row[5]) : "v"(LO), "v"(HI), "v"(nuid));
#pragma unroll
    for (int j = 0; j < 6; ++j) Iac += row[j] * c[j];
    const float pOut = pAi + Iac + uid * u;
    // is any body of this level consumed through LDS (its parent carries another child, or is the base)?
#define MOCCA_VIA_LDS(sl) (T::clevel(d, sl) >= 0 && T::ccarry(T::parent(T::clevel(d, sl) >= 0 ? T::clevel(d, sl) : 0)) != T::clevel(d, sl))
    const bool store_m = MOCCA_VIA_LDS(0) || MOCCA_VIA_LDS(1) || MOCCA_VIA_LDS(2) || MOCCA_VIA_LDS(3);
    if (valid) {
      if (store_m) {
#pragma unroll
        for (int j = 0; j < 6; ++j) L[L_M + o36 + j] = row[j];
        L[L_P + o6i] = pOut;
      }
      L[L_SV + osvi + SV_V] = uid;   // V_i = U_i / D: what the outward passes and the row sweeps multiply by
      if (i == 0) { L[L_SV + osv + SV_INVD] = id; L[L_SV + osv + SV_UU] = u * id; }
    }
    // What the slot hands to the next level in registers: its result.  Only a slot whose NEXT-level body does not continue this
    // level's chain (a chain that starts there, or this level's body is consumed through LDS) must hand over zeros, and which
    // slots those are is known at compile time: most levels need no select at all.  (Lanes of a slot that stays empty, and the two
    // idle lanes of a group, carry finite garbage that is never summed or stored.)
#define MOCCA_NEEDZ(sl) (d > 1 && T::clevel(d - 1, sl) >= 0 && !(T::clevel(d, sl) >= 0 && !MOCCA_VIA_LDS(sl)))
    if (MOCCA_NEEDZ(0) || MOCCA_NEEDZ(1) || MOCCA_NEEDZ(2) || MOCCA_NEEDZ(3)) {  // compile-time
      const bool z = s == 0 ? MOCCA_NEEDZ(0) : s == 1 ? MOCCA_NEEDZ(1) : s == 2 ? MOCCA_NEEDZ(2) : MOCCA_NEEDZ(3);
#pragma unroll
      for (int j = 0; j < 6; ++j) crow[j] = z ? 0.0f : row[j];
      cpA = z ? 0.0f : pOut;
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) crow[j] = row[j];
      cpA = pOut;
    }
#undef MOCCA_NEEDZ
#undef MOCCA_VIA_LDS
    if (store_m) wsync();
  }
  STAMP(10);
  // base: the six lanes of the first group each add up ONE ROW of the base's articulated inertia (own link + the children, which all
  // arrive through LDS) and its bias component, park row + bias as eight floats, and every lane reads the 6 x 8 block back for the
  // (redundant, wave-uniform) 6x6 solve: 12 + 2 + 12 LDS instructions instead of a 51-instruction gather of four upper triangles.
  float abase[6];
  {
    float rowb[6], pb_ = L[L_P + ii];
#pragma unroll
    for (int j = 0; j < 6; ++j) rowb[j] = L[L_M + 6 * ii + j];
#pragma unroll
    for (int k = 0; k < T::MAXCH; ++k) {
      const int ch = T::cchild(0, k);
      if (ch >= 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) rowb[j] += L[L_M + 36 * ch + 6 * ii + j];
        pb_ += L[L_P + 6 * ch + ii];
      }
    }
    if (lane < 6) {
      float4* dst = reinterpret_cast<float4*>(L + L_B0 + 8 * lane);
      dst[0] = make_float4(rowb[0], rowb[1], rowb[2], rowb[3]);
      dst[1] = make_float4(rowb[4], rowb[5], pb_, 0.0f);
    }
    wsync();
    float IA[21], pA[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float4 lo = *reinterpret_cast<const float4*>(L + L_B0 + 8 * i), hi = *reinterpret_cast<const float4*>(L + L_B0 + 8 * i + 4);
      const float rr_[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
      for (int j = i; j < 6; ++j) IA[sym(i, j)] = rr_[j];
      pA[i] = hi.z;
    }
    float Ai[21], a0[6];
    chol6_factor(IA, Ai);
    chol6_solve(Ai, pA, a0);
    // every lane solved the same system: the base acceleration needs no LDS round trip, and the factor travels to the row sweeps in
    // scalar registers (uniform by construction; it used to cost six 16-byte LDS writes, a barrier and six reads per substep)
#pragma unroll
    for (int i = 0; i < 21; ++i) Afac[i] = unif(Ai[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) abase[i] = -a0[i];
  }
  STAMP(11);
  // outward pass: lane = body, walk from the root accumulating the spatial acceleration
  {
    const int b = lane < T::NB ? lane : 0;
    float a[6], qdd = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = abase[i];
#pragma unroll
    for (int k = 0; k < T::MAXD; ++k) {
      const int j = (int)((ppk >> (5 * k)) & 31ull);
      if (j != 0) {
        float U[6], S[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { a[i] += L[L_SV + SVS * j + SV_C + i]; U[i] = L[L_SV + SVS * j + SV_V + i]; S[i] = L[L_SV + SVS * j + i]; }
        qdd = L[L_SV + SVS * j + SV_UU] - dot6(U, a);   // u / D - (U / D) . a
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] += S[i] * qdd;
      }
    }
    const float dt = M->dt;
    if (lane >= 1 && lane < T::NB) L[L_NU + 5 + b] = L[L_QD + b] + dt * qdd;
    if (lane == 0) {
      float om[3] = {L[L_BASE + 10], L[L_BASE + 11], L[L_BASE + 12]}, vl[3] = {L[L_BASE + 7], L[L_BASE + 8], L[L_BASE + 9]}, wxv[3];
      cross3(om, vl, wxv);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        L[L_NU + i] = om[i] + dt * a[i];
        L[L_NU + 3 + i] = vl[i] + dt * (a[3 + i] + wxv[i]);  // spatial -> classical acceleration of the base origin
      }
    }
    wsync();
  }
  STAMP(12);
}

// ------------------------------------------------------------------ collision
DI void plane_space(const float* n, float* t1, float* t2) {  // btPlaneSpace1
  if (fabsf(n[2]) > 0.70710678f) {
    const float a = n[1] * n[1] + n[2] * n[2], k = rsq(a);  // v_rsq_f32 (1 ulp; a >= 0.5)
    t1[0] = 0; t1[1] = -n[2] * k; t1[2] = n[1] * k;
    t2[0] = a * k; t2[1] = -n[0] * t1[2]; t2[2] = n[0] * t1[1];
  } else {
    const float a = n[0] * n[0] + n[1] * n[1], k = rsq(a);
    t1[0] = -n[1] * k; t1[1] = n[0] * k; t1[2] = 0;
    t2[0] = -n[2] * t1[1]; t2[1] = n[2] * t1[0]; t2[2] = a * k;
  }
}
DI void fast_sincos(float q, float* s, float* c);
// (plank tilts and headings are a few radians at most: fast_sincos, < 1 ulp there; six libm calls per live plank and env.step cost every
// Stepper wave ~700 VALU instructions of argument reduction in its prologue)
DI void euler_to_mat(float roll, float pitch, float yaw, float* R) {
  float cr, sr, cp, sp, cy, sy;
  fast_sincos(roll, &sr, &cr); fast_sincos(pitch, &sp, &cp); fast_sincos(yaw, &sy, &cy);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}
// sphere centre in the box frame
DI void box_local(const float* C, const float* bc, const float* Rb, float* l) {
  const float d[3] = {C[0] - bc[0], C[1] - bc[1], C[2] - bc[2]};
#pragma unroll
  for (int i = 0; i < 3; ++i) l[i] = Rb[i] * d[0] + Rb[3 + i] * d[1] + Rb[6 + i] * d[2];
}
// signed gap and world normal of a sphere (centre l in the box frame) against an oriented box
DI float sphere_box(const float* l, float rad, const float* Rb, const float* h, float* n) {
  float q[3];
  bool inside = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    q[i] = l[i];
    if (q[i] > h[i]) { q[i] = h[i]; inside = false; }
    if (q[i] < -h[i]) { q[i] = -h[i]; inside = false; }
  }
  float nl[3] = {0, 0, 0}, dist;
  if (!inside) {
    float e[3] = {l[0] - q[0], l[1] - q[1], l[2] - q[2]};
    const float e2 = dot3(e, e), id = rsq(e2);  // v_rsq_f32: distance and unit normal without IEEE sqrt / division
    dist = e2 * id;
#pragma unroll
    for (int i = 0; i < 3; ++i) nl[i] = e[i] * id;
  } else {
    float d0 = h[0] - fabsf(l[0]), d1 = h[1] - fabsf(l[1]), d2 = h[2] - fabsf(l[2]);
    int best = 0;
    float bd = d0;
    if (d1 < bd) { bd = d1; best = 1; }
    if (d2 < bd) { bd = d2; best = 2; }
    const float sg = (best == 0 ? l[0] : best == 1 ? l[1] : l[2]) >= 0 ? 1.0f : -1.0f;
    nl[0] = best == 0 ? sg : 0; nl[1] = best == 1 ? sg : 0; nl[2] = best == 2 ? sg : 0;
    dist = -bd;
  }
  matvec3(Rb, nl, n);
  return dist - rad;
}
// the same against an upright cylinder (Pillar, bullet_objects.py:86-89): axis = local z, h = (radius, radius, half height)
DI float sphere_cylinder(const float* l, float rad, const float* Rb, const float* h, float* n) {
  const float rho = sqrtf(l[0] * l[0] + l[1] * l[1]), R = h[0], hz = h[2];
  const float ux = rho > 1e-12f ? l[0] / rho : 1.0f, uy = rho > 1e-12f ? l[1] / rho : 0.0f;
  float nl[3] = {0, 0, 0}, dist;
  if (rho <= R && fabsf(l[2]) <= hz) {
    const float dcap = hz - fabsf(l[2]), dside = R - rho;
    if (dcap < dside) { nl[2] = l[2] >= 0 ? 1.0f : -1.0f; dist = -dcap; }
    else { nl[0] = ux; nl[1] = uy; dist = -dside; }
  } else {
    const float qr = rho < R ? rho : R, qz = l[2] > hz ? hz : (l[2] < -hz ? -hz : l[2]);
    const float e[3] = {l[0] - qr * ux, l[1] - qr * uy, l[2] - qz};
    const float e2 = dot3(e, e), id = rsq(e2);
    dist = e2 * id;
    nl[0] = e[0] * id; nl[1] = e[1] * id; nl[2] = e[2] * id;
  }
  matvec3(Rb, nl, n);
  return dist - rad;
}
// ---- height field of the planner envs (bullet_objects.py:338-441; geometry conventions in the oracle's sphere_heightfield)
DI void closest_on_triangle(const float* p, const float* a, const float* b, const float* c, float* q) {  // Ericson 5.1.5
  float ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; bp[k] = p[k] - b[k]; cp[k] = p[k] - c[k]; }
  const float d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
  float v, w;   // q = a + v ab + w ac
  if (d1 <= 0 && d2 <= 0) { v = 0; w = 0; }
  else if (d3 >= 0 && d4 <= d3) { v = 1; w = 0; }
  else if (vc <= 0 && d1 >= 0 && d3 <= 0) { v = d1 / (d1 - d3); w = 0; }
  else if (d6 >= 0 && d5 <= d6) { v = 0; w = 1; }
  else if (vb <= 0 && d2 >= 0 && d6 <= 0) { v = 0; w = d2 / (d2 - d6); }
  else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); v = 1 - w; }
  else { const float den = 1.0f / (va + vb + vc); v = vb * den; w = vc * den; }
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = a[k] + ab[k] * v + ac[k] * w;
}
// one triangle (a, b, c) of the height field against the sphere centre C: keeps the smaller gap and its normal (tn: the triangle's normal, unnormalised; il = 1 / |tn|)
DI void hf_triangle(const float* C, float rad, const float* a, const float* b, const float* c, const float* tn, float il, float& gap, float* n) {
  float q[3], d[3];
  closest_on_triangle(C, a, b, c, q);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = C[k] - q[k];
  const float d2 = dot3(d, d), id = rsq(d2), dist = d2 > 0 ? d2 * id : 0.0f;
  if (dist - rad < gap) {
    gap = dist - rad;
    if (d2 > 1e-18f) { n[0] = d[0] * id; n[1] = d[1] * id; n[2] = d[2] * id; }
    else { n[0] = tn[0] * il; n[1] = tn[1] * il; n[2] = tn[2] * il; }
  }
}
// signed gap and world normal of a sphere (world centre C) against the height field (see the oracle's sphere_heightfield): above the surface
// the closest of the triangles of the 2 W x 2 W cells around the grid point nearest to the centre -- W = ceil((radius + margin) scale + 1/2),
// every cell a sphere of that reach can touch; W travels in the slot record (mocca_set_heightfield computes it: 1 for everything but the
// walker's 14 cm and Mike's 23 cm spheres at 4 points per metre) --; below the surface the plane of the triangle the centre is under;
// 1e30 where there is no terrain.  The central 2 x 2 cells are unrolled with their nine heights in registers (all lanes); the ring of a
// wider window is a rolled loop that only runs while such a sphere is within reach of the highest point of ITS window (one more load: the
// max-pooled copy of the grid that follows the heights, hf + (W - 1) rows cols) -- a fallen robot.
DI float sphere_heightfield(const float* __restrict__ hf, int rows, int cols, float sc, const float* C, float rad, float reach, int W, float* n) {
  float gap = 1e30f;
  n[0] = 0; n[1] = 0; n[2] = 1;
  const float cell = 1.0f / sc, hx = 0.5f * (float)(cols - 1), hy = 0.5f * (float)(rows - 1);
  const float fx = C[0] * sc + hx, fy = C[1] * sc + hy;
  if (!(fx >= -1.0f && fx <= (float)cols && fy >= -1.0f && fy <= (float)rows)) return gap;
  const int iv = (int)floorf(fx + 0.5f), jv = (int)floorf(fy + 0.5f);
  const int ic = (int)floorf(fx), jc = (int)floorf(fy);   // the cell the centre is over
  // the nine grid points around it (clamped reads; cells outside the grid are skipped below)
  float hv[3][3];
  float hmax = -1e30f;
#pragma unroll
  for (int dj = 0; dj < 3; ++dj)
#pragma unroll
    for (int di = 0; di < 3; ++di) {
      int i = iv - 1 + di, j = jv - 1 + dj;
      i = i < 0 ? 0 : (i > cols - 1 ? cols - 1 : i);
      j = j < 0 ? 0 : (j > rows - 1 ? rows - 1 : j);
      hv[dj][di] = hf[j * cols + i];
      hmax = fmaxf(hmax, hv[dj][di]);
    }
  if (W > 1) {   // the highest point within W cells of the nearest grid point (>= the nine above)
    const int i = iv < 0 ? 0 : (iv > cols - 1 ? cols - 1 : iv), j = jv < 0 ? 0 : (jv > rows - 1 ? rows - 1 : jv);
    hmax = hf[(W - 1) * rows * cols + j * cols + i];
  }
  if (C[2] - reach > hmax) return gap;   // above everything nearby: no contact possible (exact: every triangle of the window lies below hmax)
  bool below = false;
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int di = 0; di < 2; ++di) {
      const int i = iv - 1 + di, j = jv - 1 + dj;
      if (i < 0 || j < 0 || i > cols - 2 || j > rows - 2) continue;
      const float x0 = ((float)i - hx) * cell, y0 = ((float)j - hy) * cell;
      const float v00[3] = {x0, y0, hv[dj][di]}, v10[3] = {x0 + cell, y0, hv[dj][di + 1]};
      const float v01[3] = {x0, y0 + cell, hv[dj + 1][di]}, v11[3] = {x0 + cell, y0 + cell, hv[dj + 1][di + 1]};
      const float u = fx - (float)i, v = fy - (float)j;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* a = t == 0 ? v00 : v10;
        const float* b = t == 0 ? v10 : v11;
        const float* c = v01;
        float e1[3], e2[3], tn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { e1[k] = b[k] - a[k]; e2[k] = c[k] - a[k]; }
        cross3(e1, e2, tn);
        const float il = rsq(dot3(tn, tn));
        if (!below && i == ic && j == jc && (t == 0) == (u + v <= 1.0f)) {   // the triangle under the centre
          const float side = ((C[0] - a[0]) * tn[0] + (C[1] - a[1]) * tn[1] + (C[2] - a[2]) * tn[2]) * il;
          if (side < 0) { below = true; gap = side - rad; n[0] = tn[0] * il; n[1] = tn[1] * il; n[2] = tn[2] * il; }
        }
        if (below) continue;
        hf_triangle(C, rad, a, b, c, tn, il, gap, n);
      }
    }
  if (W > 1 && !below) {   // the ring around the central cells (rare: see above)
#pragma unroll 1
    for (int dj = -W; dj < W; ++dj)
#pragma unroll 1
      for (int di = -W; di < W; ++di) {
        const int i = iv + di, j = jv + dj;
        if ((dj == -1 || dj == 0) && (di == -1 || di == 0)) continue;   // done above
        if (i < 0 || j < 0 || i > cols - 2 || j > rows - 2) continue;
        const float h00 = hf[j * cols + i], h10 = hf[j * cols + i + 1], h01 = hf[(j + 1) * cols + i], h11 = hf[(j + 1) * cols + i + 1];
        if (C[2] - reach > fmaxf(fmaxf(h00, h10), fmaxf(h01, h11))) continue;   // this cell lies below the sphere's reach (exact, as above)
        const float x0 = ((float)i - hx) * cell, y0 = ((float)j - hy) * cell;
        const float v00[3] = {x0, y0, h00}, v10[3] = {x0 + cell, y0, h10}, v01[3] = {x0, y0 + cell, h01}, v11[3] = {x0 + cell, y0 + cell, h11};
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
          const float* a = t == 0 ? v00 : v10;
          const float* b = t == 0 ? v10 : v11;
          float e1[3], e2[3], tn[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) { e1[k] = b[k] - a[k]; e2[k] = v01[k] - a[k]; }
          cross3(e1, e2, tn);
          hf_triangle(C, rad, a, b, v01, tn, rsq(dot3(tn, tn)), gap, n);
        }
      }
  }
  return gap;
}
// HeightField.get_height_at (bullet_objects.py:348-353), indices clamped to the grid
DI float hf_height_at(const float* hf, int rows, int cols, float sc, float x, float y) {
  const float ox = (float)rows / sc * 0.5f, oy = (float)cols / sc * 0.5f;
  int ix = (int)((x + ox) * sc), iy = (int)((y + oy) * sc);
  ix = ix < 0 ? 0 : (ix > cols - 1 ? cols - 1 : ix);
  iy = iy < 0 ? 0 : (iy > rows - 1 ? rows - 1 : iy);
  return hf[iy * cols + ix];
}

DI float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
DI void seg_seg(const float* p1, const float* q1, const float* p2, const float* q2, float* c1, float* c2) {
  // closest points of two segments (Ericson 5.1.9); quotients through v_rcp_f32 (1 ulp) instead of IEEE division
  float d1[3], d2[3], r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d1[k] = q1[k] - p1[k]; d2[k] = q2[k] - p2[k]; r[k] = p1[k] - p2[k]; }
  const float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r), EPS = 1e-12f;
  float s, t;
  if (a <= EPS && e <= EPS) { s = t = 0; }
  else if (a <= EPS) { s = 0; t = clamp01(f * rcp(e)); }
  else {
    const float c = dot3(d1, r), ia = rcp(a);
    if (e <= EPS) { t = 0; s = clamp01(-c * ia); }
    else {
      const float b = dot3(d1, d2), den = a * e - b * b;
      s = den > EPS ? clamp01((b * f - c * e) * rcp(den)) : 0.0f;
      t = (b * s + f) * rcp(e);
      if (t < 0) { t = 0; s = clamp01(-c * ia); }
      else if (t > 1) { t = 1; s = clamp01((b - c) * ia); }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { c1[k] = p1[k] + d1[k] * s; c2[k] = p2[k] + d2[k] * t; }
}

// world (base-origin relative) end points of every geom: lane = geom end
template <class T>
DI void geom_points(ModelP M, float* L, int lane) {
  // L_GP holds L_CT - L_GP floats; a topology with more geoms spills into the contact records, which is harmless only
  // while nothing reads L_GP after the terrain contacts are written, i.e. without self-collision pairs
  // (check_topology_t rejects blobs with pairs for such topologies)
  static_assert(6 * T::NG <= GP_FLOATS || (T::NPAIR == 0 && !COMPACT), "geom points would overlap the contact records read by the self-collision pass");
  if (lane < 2 * T::NG) {
    const f4_t t = *(CF4P)(M->gp_tab[lane]);  // point (body frame) + body id, one load
    const int b = __float_as_int(t.w);
    const float pl[3] = {t.x, t.y, t.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 fr = *reinterpret_cast<const float4*>(L + L_RT + 12 * b + 4 * i);   // row i of the rotation, origin component i
      L[L_GP + 3 * lane + i] = fr.x * pl[0] + fr.y * pl[1] + fr.z * pl[2] + fr.w;
    }
  }
}

// Stepper: rotation and box centre of the three live planks (bullet_objects.py:77-83 offset included), once per env.step.
// They were rebuilt from the terrain table -- six sin/cos and seven global reads -- per (lane, plank, substep).
DI void stage_planks(ModelP M, float* L, int lane, const float* ter) {
  if (lane < M->n_planks) {
    const int row = (int)ter[120 + lane];
    const float* ti = ter + 6 * row;
    float Rb[9];
    euler_to_mat(ti[4], ti[5], ti[3], Rb);
    const float cz = M->plank_com_z, dz = -M->plank_half[2] - cz;
#pragma unroll
    for (int i = 0; i < 9; ++i) L[L_PLANK + 12 * lane + i] = Rb[i];
    L[L_PLANK + 12 * lane + 9] = ti[0] + Rb[2] * dz;
    L[L_PLANK + 12 * lane + 10] = ti[1] + Rb[5] * dz;
    L[L_PLANK + 12 * lane + 11] = ti[2] + Rb[8] * dz + cz;
  }
  wsync();
}

struct HeightFieldArgs { const float* data; int rows, cols; float scale; };   // planner envs; data == nullptr otherwise
struct ContactFlags { int touch0, touch1, target0, target1, touch2, touch3, body_touch, target2, target3; };  // feet 2, 3: quadrupeds; body_touch: a non-foot link on the terrain
// Stepper: which planks' COVERS each foot touches, 4 bits per foot (bit 4 f + k: foot f on the cover of live plank k).  target_f is bit
// (next_step_index mod n_planks) of foot f's nibble -- at the step's start for the step itself, at its END for the stale read of reset()
DI int cover_targets(int cover, int nsi, int n_planks, int f) { return (cover >> (4 * f + nsi % n_planks)) & 1; }

// lane = terrain contact slot, then self-collision pairs strided over the wave.
// Contacts are compacted in slot order, then pair order (the oracle's priority), up to max_contacts.
template <class T, int TASK>
DI ContactFlags collide(ModelP M, float* L, int lane, const float* ter, int next_step_index,
                        int* nc_out, int32_t* dbg, int* nc_wanted, const HeightFieldArgs hfa, int* cover_out = nullptr) {
  STAMP_BEGIN;
  // contacts open within the geom's margin (Bullet's relative breaking threshold of its link, a few mm; decoded from the slot record where
  // it is compared: a separate load spilled).  `mreach`: gContactBreakingThreshold itself, 20 mm -- an upper bound of every relative
  // threshold (discs below 1 m), used where a reach only prunes
  const float mreach = unif(M->contact_margin);
  ContactFlags fl = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int maxc = uni(M->max_contacts);
  // ---- terrain: lane -> (geom, end)
  bool active = false;
  float n[3] = {0, 0, 1}, P[3] = {0, 0, 0}, gap = 1e30f, mu = 0, erp = M->erp, cfm = 0;
  int body = -1, slot = lane, cover_k = -1 /* Stepper: the plank whose cover this slot rests on */, gfoot = -1, gtorso = 0;
  unsigned bmask = 0u;
  if (lane < T::NSLOT) {
    const f4_t st = *(CF4P)(M->slot_tab[lane]);  // radius, friction, ids, ancestor mask
    const int ids = __float_as_int(st.z);
    const int g = (ids >> 8) & 0xFF, e = (ids >> 16) & 1;
    bmask = __float_as_uint(st.w);
    if ((ids >> 25) & 1) {  // flags: bits 17..24 margin code, bit 25 terrain, bits 26..28 foot index + 1, bit 29 torso, bits 30..31 height-field window - 1 (mocca_set_heightfield)
      gfoot = ((ids >> 26) & 7) - 1;
      gtorso = (ids >> 29) & 1;
      float C[3], Cw[3];
      const float rad = st.x, gfric = st.y;
#pragma unroll
      for (int i = 0; i < 3; ++i) { C[i] = L[L_GP + 3 * (2 * g + e) + i]; Cw[i] = C[i] + L[L_BASE + i]; }
      body = ids & 0xFF;
      if (TASK == MOCCA_TASK_WALKER3D_PLANNER) {
        // height field (HeightField.reload: lateralFriction 1.0, contactStiffness 30000, contactDamping 1000, bullet_objects.py:386-393)
        gap = sphere_heightfield(hfa.data, hfa.rows, hfa.cols, hfa.scale, Cw, rad, rad + mreach, 1 + ((ids >> 30) & 3), n);   // bits 30..31: the slot's window - 1
        mu = M->plank_friction * gfric;
        const float kk = M->plank_stiffness, cc = M->plank_damping, dt = M->dt;
        const float ikc = rcp(dt * kk + cc);
        erp = dt * kk * ikc;
        cfm = ikc * rcp(dt);
      } else if (TASK != MOCCA_TASK_WALKER3D_STEPPER) {
        gap = Cw[2] - rad;
        mu = M->ground_friction * gfric;
      } else {
        const float h[3] = {M->plank_half[0], M->plank_half[1], M->plank_half[2]};
        const int n_planks = M->n_planks;
        const bool cyl = M->plank_shape == MOCCA_PLANK_CYLINDER;
#pragma unroll 1
        for (int k = 0; k < n_planks; ++k) {
          float Rb[9], bc[3], nn[3];  // staged by stage_planks(): the planks do not move during the substeps
#pragma unroll
          for (int i = 0; i < 9; ++i) Rb[i] = L[L_PLANK + 12 * k + i];
#pragma unroll
          for (int i = 0; i < 3; ++i) bc[i] = L[L_PLANK + 12 * k + 9 + i];
          float lb[3];
          box_local(Cw, bc, Rb, lb);
          // wave-uniform skip: no contact point of this env is within reach of plank k (exact: such a plank can neither
          // activate a slot nor win the minimum against one that does)
          const float reach = rad + mreach;
          if (__ballot(fabsf(lb[0]) < h[0] + reach && fabsf(lb[1]) < h[1] + reach && fabsf(lb[2]) < h[2] + reach) == 0ull) continue;
          const float gk = cyl ? sphere_cylinder(lb, rad, Rb, h, nn) : sphere_box(lb, rad, Rb, h, nn);  // uniform branch
          if (gk < gap) {
            gap = gk;
            n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2];
            float d[3] = {Cw[0] - rad * nn[0] - bc[0], Cw[1] - rad * nn[1] - bc[1], Cw[2] - rad * nn[2] - bc[2]};
            const float lz = Rb[2] * d[0] + Rb[5] * d[1] + Rb[8] * d[2];
            cover_k = lz >= h[2] * 0.8f ? k : -1;
          }
        }
        mu = M->plank_friction * gfric;
        const float kk = M->plank_stiffness, cc = M->plank_damping, dt = M->dt;
        const float ikc = rcp(dt * kk + cc);  // soft contact (bullet_objects.py:70-71): erp = dt k / (dt k + c), cfm = 1 / ((dt k + c) dt)
        erp = dt * kk * ikc;
        cfm = ikc * rcp(dt);
      }
      active = gap < (float)((ids >> 17) & 0xFF) * (1.0f / 8192.0f);   // the link's margin, an 8-bit multiple of 2^-13 m in the slot record (= slot_margin)
#pragma unroll
      for (int i = 0; i < 3; ++i) P[i] = C[i] - rad * n[i];
    }
  }
  // Contact manifolds (MoccaModel.manifold_max): of the terrain slots of ONE link that are within the margin at most four survive -- the
  // deepest, the one farthest from it, and the farthest on either side of the line through those two (signed area about the deepest
  // point's normal) -- what a 4-point persistent manifold keeps of a convex hull lying on the ground.  Wave-uniform loop over the links
  // that touch; only Cassie's blobs switch it on (twelve hull points per toe).
  if (uni(M->manifold_max) > 0) {
    unsigned long long todo = __ballot(active);
    while (todo) {
      const int l0 = __builtin_ctzll(todo);
      const int b0 = readlane_i(body, l0);
      const bool inG = active && body == b0;
      const unsigned long long G = __ballot(inG);
      todo &= ~G;
      if (__popcll(G) <= 4) continue;
      // (the wave reductions are statements of their own: inside `inG && ...` they would run under the group's EXEC mask -- C++ short
      // circuit -- and the DPP steps would read inactive lanes as 0)
      const float dep = inG ? -gap : -1e30f;
      const float dep_max = wave_max(dep);
      const int l1 = __builtin_ctzll(__ballot(inG && dep == dep_max));
      const float P1[3] = {readlane(P[0], l1), readlane(P[1], l1), readlane(P[2], l1)};
      const float n1[3] = {readlane(n[0], l1), readlane(n[1], l1), readlane(n[2], l1)};
      const float d[3] = {P[0] - P1[0], P[1] - P1[1], P[2] - P1[2]};
      const bool c2 = inG && lane != l1;
      const float dd = c2 ? d[0] * d[0] + d[1] * d[1] + d[2] * d[2] : -1e30f;
      const float dd_max = wave_max(dd);
      const int l2 = __builtin_ctzll(__ballot(c2 && dd == dd_max));
      const float e[3] = {readlane(P[0], l2) - P1[0], readlane(P[1], l2) - P1[1], readlane(P[2], l2) - P1[2]};
      float cx[3];
      cross3(d, e, cx);
      const float sg = n1[0] * cx[0] + n1[1] * cx[1] + n1[2] * cx[2];
      const bool c3 = c2 && lane != l2;
      const float sp = c3 ? sg : -1e30f, sm = c3 ? -sg : -1e30f;
      const float mp = wave_max(sp), mm = wave_max(sm);
      const int l3 = mp > 0.0f ? __builtin_ctzll(__ballot(c3 && sp == mp)) : -1;
      const int l4 = mm > 0.0f ? __builtin_ctzll(__ballot(c3 && sm == mm)) : -1;
      if (inG && lane != l1 && lane != l2 && lane != l3 && lane != l4) active = false;
    }
  }
  unsigned long long am = __ballot(active);
  {
    fl.touch0 = __ballot(active && gfoot == 0) != 0ull;
    fl.touch1 = __ballot(active && gfoot == 1) != 0ull;
    if constexpr (T::NFEET > 2) {
      fl.touch2 = __ballot(active && gfoot == 2) != 0ull;
      fl.touch3 = __ballot(active && gfoot == 3) != 0ull;
      // LaikagoCustomEnv ends the episode when anything but a foot link meets the ground (env_locomotion.py:880-890)
      fl.body_touch = __ballot(active && gfoot < 0) != 0ull;
    }
    if (TASK == MOCCA_TASK_WALKER3D_STEPPER) {   // calc_feet_state's target test (env_locomotion.py:634-650) from the cover mask
      const int cover = (int)wave_or((active && gfoot >= 0 && cover_k >= 0) ? 1u << (4 * gfoot + cover_k) : 0u);
      const int npl = M->n_planks;
      fl.target0 = cover_targets(cover, next_step_index, npl, 0);
      fl.target1 = cover_targets(cover, next_step_index, npl, 1);
      if constexpr (T::NFEET > 2) { fl.target2 = cover_targets(cover, next_step_index, npl, 2); fl.target3 = cover_targets(cover, next_step_index, npl, 3); }
      if (cover_out) *cover_out = cover;
    }
    // Walker3DPlannerEnv: the torso link touches anything -> done (env_locomotion.py:1104-1110); the terrain here, robot links below
    if (TASK == MOCCA_TASK_WALKER3D_PLANNER) fl.body_touch = __ballot(active && gtorso) != 0ull;
  }
  int nc = __popcll(am);
  const int n_terrain = nc;   // before the cap (debug record: cap pressure)
  int n_self = 0;
  unsigned long long kept = am;
  if (nc > maxc) {
    // More terrain contacts than the solver holds (a robot lying on the ground): keep the max_contacts DEEPEST ones (ties: lower
    // slot first), still solved in slot order.  Rare and wave-uniform: one readlane per active slot.
    int deeper = 0;
    for (unsigned long long mm = am; mm; mm &= mm - 1) {
      const int l = __builtin_ctzll(mm);
      const float gl = readlane(gap, l);
      deeper += (gl < gap || (gl == gap && l < lane)) ? 1 : 0;
    }
    kept = __ballot(active && deeper < maxc);
  }
  {
    const int idx = lane_rank(kept);
    if (active && ((kept >> lane) & 1ull)) {
      float* ct = L + L_CT + 16 * idx;
      ct[C_BA] = __int_as_float(body); ct[C_BB] = __int_as_float(-1); ct[C_SLOT] = __int_as_float(slot);
#pragma unroll
      for (int i = 0; i < 3; ++i) { ct[C_P + i] = P[i]; ct[C_N + i] = n[i]; }
      ct[C_DEPTH] = -gap; ct[C_MU] = mu; ct[C_ERP] = erp; ct[C_CFM] = cfm;
      ct[C_MA] = __uint_as_float(bmask); ct[C_MB] = __uint_as_float(0u);
    }
  }
  if (nc > maxc) nc = maxc;
  STAMP(13);
  // ---- self collisions
  // pass 1: conservative broad phase over every candidate pair (bounding spheres around the segment midpoints); the
  // survivors are compacted in pair order into a 16-bit list (the U area is not written until the ABA runs).
  // pass 2: narrow phase over the survivors only -- typically one batch of 64 instead of ceil(n_pairs / 64).
  const int npairs = uni(M->n_pairs);
  unsigned short* cand = reinterpret_cast<unsigned short*>(L + L_CAND);
  static_assert(2 * ((COMPACT ? L_ABA_END : L_TOTAL) - L_CAND) >= MOCCA_MAX_PAIRS, "candidate list must hold every pair");
  int ncand = 0;
#pragma unroll 1
  for (int base = 0; base < npairs; base += 64) {
    const int k = base + lane;
    bool near = false;
    if (k < npairs) {
#ifdef MOCCA_ABL_PAIRLOAD   // ablation (round 5's A/B, profiles/r05_l2chain_ab.txt; results WRONG by construction): the broad phase's pair record made up from the lane
      f4_t pt;              // instead of loaded -- what ANY staging of the pair table could save at most (measured: 0.2 % of the launch)
      pt.x = __int_as_float((k % T::NG) | (((k * 7 + 3) % T::NG) << 5) | (24 << 20)); pt.y = 0.05f; pt.z = 0.05f; pt.w = 0.04f;
#else
      const f4_t pt = *(CF4P)(M->pair_tab[k]);  // geoms, bodies, radii, reach: one load
#endif
      const int ids = __float_as_int(pt.x);   // geom_a | geom_b << 5 | body_a << 10 | body_b << 15 | margin code << 20
      const int ga = ids & 31, gb = (ids >> 5) & 31;
      float dm[3];  // distance of the two segment midpoints (x2); the segments' half lengths and radii are constants of the pair (pt.w)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        dm[i] = (L[L_GP + 6 * ga + i] + L[L_GP + 6 * ga + 3 + i]) - (L[L_GP + 6 * gb + i] + L[L_GP + 6 * gb + 3 + i]);
      const float reach = 2.0f * (pt.w + (float)((ids >> 20) & 0xFF) * (1.0f / 8192.0f));   // + the pair's margin (= pair_margin[k])
      near = dot3(dm, dm) < reach * reach;
    }
    const unsigned long long nm = __ballot(near);
    if (near) cand[ncand + lane_rank(nm)] = (unsigned short)k;
    ncand += __popcll(nm);
  }
  ncand = uni(ncand);
#ifdef MOCCA_ABL_NOPASS2   // ablation (results WRONG by construction): the broad phase runs, its survivors are dropped -- narrow phase + the rows its contacts add
  ncand = 0;
#endif
  wsync();
#pragma unroll 1
  for (int base = 0; base < ncand; base += 64) {
    bool hit = false;
    float nn[3] = {0, 0, 0}, PP[3] = {0, 0, 0}, g2 = 0, mu2 = 0;
    int ba = -1, bb = -1;
    if (base + lane < ncand) {
      const int k = cand[base + lane];
      const f4_t pt = *(CF4P)(M->pair_tab[k]);
      const int ids = __float_as_int(pt.x);
      const int ga = ids & 31, gb = (ids >> 5) & 31;
      float a1[3], a2[3], b1[3], b2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        a1[i] = L[L_GP + 6 * ga + i]; a2[i] = L[L_GP + 6 * ga + 3 + i];
        b1[i] = L[L_GP + 6 * gb + i]; b2[i] = L[L_GP + 6 * gb + 3 + i];
      }
      float ca[3], cb[3];
      seg_seg(a1, a2, b1, b2, ca, cb);
      float d[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
      const float d2 = dot3(d, d), id = rsq(d2), dist = d2 * id, ra = pt.y, rb = pt.z;  // v_rsq_f32 (1 ulp)
      g2 = dist - ra - rb;
      hit = g2 < (float)((ids >> 20) & 0xFF) * (1.0f / 8192.0f) && d2 > 1e-18f;   // the smaller of the two links' relative thresholds
#ifdef MOCCA_ABL_NOHITS      // ablation (results WRONG by construction): the narrow phase runs, its contacts are dropped -- separates its own cost from the rows it adds
      hit = hit && g2 < -1e30f;
#endif
      if (hit) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          nn[i] = d[i] * id;
          PP[i] = 0.5f * ((ca[i] - ra * d[i] * id) + (cb[i] + rb * d[i] * id));
        }
        ba = (ids >> 10) & 31; bb = (ids >> 15) & 31;
        mu2 = M->g_friction[ga] * M->g_friction[gb];  // rare path (a self contact): two cached loads
      }
    }
    const unsigned long long hm = __ballot(hit);
    if (TASK == MOCCA_TASK_WALKER3D_PLANNER && hm != 0ull) {   // rare: a self contact; does it involve a geom of the torso link?
      int tor = 0;
      if (hit) { const int ids2 = __float_as_int((*(CF4P)(M->pair_tab[cand[base + lane]])).x); tor = M->g_torso[ids2 & 31] | M->g_torso[(ids2 >> 5) & 31]; }
      if (__ballot(hit && tor) != 0ull) fl.body_touch = 1;
    }
    if (TASK == MOCCA_TASK_WALKER3D_STEPPER) {  // calc_feet_state counts any contact of a foot link
      const int f0 = M->foot_body[0], f1 = M->foot_body[1];
      if (__ballot(hit && (ba == f0 || bb == f0)) != 0ull) fl.touch0 = 1;
      if (__ballot(hit && (ba == f1 || bb == f1)) != 0ull) fl.touch1 = 1;
      if constexpr (T::NFEET > 2) {
        const int f2 = M->foot_body[2], f3 = M->foot_body[3];
        if (__ballot(hit && (ba == f2 || bb == f2)) != 0ull) fl.touch2 = 1;
        if (__ballot(hit && (ba == f3 || bb == f3)) != 0ull) fl.touch3 = 1;
      }
    }
    const int idx = nc + lane_rank(hm);
    if (hit && idx < maxc) {
      float* ct = L + L_CT + 16 * idx;
      ct[C_BA] = __int_as_float(ba); ct[C_BB] = __int_as_float(bb); ct[C_SLOT] = __int_as_float(-1);
#pragma unroll
      for (int i = 0; i < 3; ++i) { ct[C_P + i] = PP[i]; ct[C_N + i] = nn[i]; }
      ct[C_DEPTH] = -g2; ct[C_MU] = mu2; ct[C_ERP] = M->erp; ct[C_CFM] = 0.0f;
      ct[C_MA] = __uint_as_float(M->anc_mask[ba]); ct[C_MB] = __uint_as_float(M->anc_mask[bb]);  // rare path
    }
    nc += __popcll(hm);
    n_self += __popcll(hm);
    if (nc > maxc) nc = maxc;
  }
  STAMP(14);
  if (dbg && lane == 0) {  // active set of this substep (MOCCA_DEBUG_WORDS, include/mocca.h)
    dbg[3] = (int32_t)(unsigned)am; dbg[4] = (int32_t)(unsigned)(am >> 32); dbg[7] = n_self;
  }
  *nc_out = nc;
  *nc_wanted = n_terrain + n_self;
  return fl;
}

// Delassus matrix by symmetry: A[r][c] = J_r . X_c = A[c][r], so lane c only evaluates the rows (c + t) mod nr for
// t = 0 .. nr / 2 (a circulant schedule covers every unordered pair once) and writes each value to both places.
// Step t of every lane reads a DIFFERENT J row (stride 28 floats: the 16-byte reads of 16 consecutive lanes tile all 64
// banks, no conflicts beyond the four passes a b128 read takes anyway).  Values stay in registers until every J row has
// been read -- A overlays the J rows.  Compile-time recursion: v[TT] must be a static register.
// Rows are not contiguous in the wave (fixed-bound rows on lanes 0 .. r_fr - 1, friction rows on the top lanes, see fric_lane), but
// the J rows are stored by DENSE row number 0 .. nr - 1, so the schedule needs one add and one wrap per step (oc = 28 c, onr = 28 nr:
// unsigned min(o, o - onr) is the wrap); only the store maps dense numbers back to lanes (d below r_fr, d + gap above).
template <class T, int TT>
DI void delassus_dots(const float* L, const float* X, int oc, int onr, int tmax, float* v) {
  if constexpr (TT <= MAXR / 2) {
    if (TT > tmax) return;
    const unsigned o = (unsigned)(oc + 28 * TT);
    const unsigned ow = o - (unsigned)onr;
    const float* Jr = L + L_J + (int)(o < ow ? o : ow);
    float s;
    if constexpr (T::NCLOS > 0) {
      // Cassie runs two waves per SIMD (2048 envs) with ~45 rows per substep: the 28-term dot is a latency chain there, two chains halve it
      // (at four waves per SIMD the other waves fill the gaps and the split measured nothing)
      float s0 = 0, s1 = 0;
#pragma unroll
      for (int d = 0; d + 1 < T::ND; d += 2) { s0 += Jr[d] * X[d]; s1 += Jr[d + 1] * X[d + 1]; }
      if constexpr (T::ND & 1) s0 += Jr[T::ND - 1] * X[T::ND - 1];
      s = s0 + s1;
    } else {
      s = 0;
#pragma unroll
      for (int d = 0; d < T::ND; ++d) s += Jr[d] * X[d];
    }
    v[TT] = s;
    pin1(v[TT]);  // keeps the steps in order: hoisting the next steps' 28-register J rows above this point spills
    delassus_dots<T, TT + 1>(L, X, oc, onr, tmax, v);
  }
}
template <int TT>
DI void delassus_store(float* L, int c, int nr, int tmax, const float* v) {  // A is indexed by dense row numbers on both sides
  if constexpr (TT <= MAXR / 2) {
    if (TT > tmax) return;
    const unsigned r0 = (unsigned)(c + TT), rw = r0 - (unsigned)nr;
    const int rho = (int)(r0 < rw ? r0 : rw);
    if (TT == 0) {
      L[L_A + (MAXR + 1) * c] = 0.0f;  // the solver works on a zero diagonal (see solve_constraints); the value stays in v[0]
    } else {
      L[L_A + __mul24(MAXR, rho) + c] = v[TT];
      L[L_A + __mul24(MAXR, c) + rho] = v[TT];
    }
    delassus_store<TT + 1>(L, c, nr, tmax, v);
  }
}

// One PGS visit of fixed-bound row RR, then the next (compile-time recursion = guaranteed full unrolling with a
// wave-uniform early exit; the optimiser keeps a `#pragma unroll` loop with a break rolled).  With RR an immediate, the
// readlane / writelane index and the LDS offset are constants and the A entry of the next visit is fetched one visit
// ahead with no register shuffling.  See solve_constraints for the y formulation.
// One PGS visit of fixed-bound row RR (see solve_constraints for the y formulation): `as` = A[RR][lane] / (A_ll + cfm).
template <int RR>
DI void pgs_visit(float as, float& y, float& lam, float lo0) {
  const float nl_ = __builtin_amdgcn_fmed3f(y, lo0, 1e30f);
  const float dl = readlane(nl_ - lam, RR);
  lam = commit_lane<RR>(nl_, lam);
  y = fmaf(-as, dl, y);
}
// a friction row: the same visit with the symmetric bound |lambda| <= lm = mu * (impulse of the contact's normal row), per lane
template <int RR>
DI void pgs_visit_friction(float as, float& y, float& lam, float lm) {
  const float nl_ = __builtin_amdgcn_fmed3f(y, -lm, lm);
  const float dl = readlane(nl_ - lam, RR);
  lam = commit_lane<RR>(nl_, lam);
  y = fmaf(-as, dl, y);
}
// Implicit cone friction (MoccaModel.friction_cone; btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows): the two friction rows of
// contact I take their candidates from the SAME state (y of lanes A and B as they stand), the pair is clipped to the circle of radius
// lm = mu * lambda_n, then both deltas go out.  The clip is evaluated per LANE -- the partner's candidate arrives by one DPP quad swap (the
// pair sits on lanes 2k, 2k + 1), every friction lane holds its own contact's lm -- so only the two deltas cross to the scalar side:
// 12 VALU for the pair against 10 for the two pyramid visits it replaces (the first form, with the candidates and the bound read into
// SGPRs, took 18 and cost the launch 4 %).  min(1, lm rsq(r2)): r2 = 0 gives inf -> 1, and 0 x inf = NaN -> 1 too (v_min returns the number).
template <int A>
DI float commit_pair(float v, float old) {   // old with lanes A and A + 1 replaced by those lanes' v
  float r;
  unsigned long long m;
  asm("s_mov_b64 %1, 0\n\ts_bitset1_b64 %1, %4\n\ts_bitset1_b64 %1, %5\n\tv_cndmask_b32 %0, %2, %3, %1"
      : "=v"(r), "=&s"(m) : "v"(old), "v"(v), "n"(A), "n"(A + 1));
  return r;
}
template <int A, int B>
DI void pgs_visit_cone(float asA, float asB, float& y, float& lam, float lm) {
  static_assert(B == A + 1 && A % 2 == 0, "the pair must share a DPP quad");
  const float y2 = y * y;
  const float r2 = y2 + dpp_mov<0xB1>(y2);        // quad_perm [1,0,3,2]: + the partner row's candidate squared (one v_add_f32_dpp)
  const float sc = fminf(1.0f, lm * rsq(r2));
  const float nl_ = y * sc;                       // meaningful on the friction lanes
  const float dl = nl_ - lam;
  const float dA = readlane(dl, A), dB = readlane(dl, B);
  lam = commit_pair<A>(nl_, lam);
  y = fmaf(-asA, dA, y);
  y = fmaf(-asB, dB, y);
}
// Friction rows live on STATIC lanes at the top of the row range: contact i owns lanes MAXR - 2 - 2i (first tangent) and
// MAXR - 1 - 2i (second), whatever the number of limit / normal rows below them.  With the lane an immediate a friction visit is
// the fixed-bound visit (readlane / writelane immediates, LDS offsets immediates, A entries requested two contacts ahead): ~30
// cycles.  The former layout (friction rows right after the normals, lane = r_fr + 2i + s, a rolled loop with an exec-mask
// commit and per-contact bound look-ups) cost ~135 cycles per visit (tools/pgs_chain_bench.hip) -- most of the solver time of
// the contact-rich waves that set the launch time.
constexpr int fric_lane(int i, int s) { return MAXR - 2 - 2 * i + s; }
// The solver re-reads the same column of A in each of its iterations.  The gains of the first PGS_REG_ROWS fixed-bound rows and of
// the first PGS_REG_CONTACTS contacts' friction rows -- already multiplied by the lane's 1 / (A_cc + cfm) -- are fetched ONCE per
// substep into registers (static indices: the visits are unrolled); a typical env (12 rows) then runs its five iterations without a
// single LDS read, and a visit loses its multiply.  Rows past the window keep the per-iteration LDS reads, requested a group ahead.
// Window sizes by topology (tools/flag_sweep.sh): 16 rows + 4 contacts for the walkers (mean 12 rows; larger windows measure the
// same), 24 + 12 for Cassie, whose closures, planar rows and a dozen toe points put ~45 rows in every substep (-1.8 %).
#ifdef MOCCA_PGS_REG_ROWS   // override for sweeps
template <class T> struct PgsWin { static constexpr int ROWS = MOCCA_PGS_REG_ROWS, CONTACTS = MOCCA_PGS_REG_CONTACTS; };
#else
// (round 3 late: limit rows exist only at the stops and contacts open within millimetres -- a walker's substep holds ~11 rows, Cassie's at
// most 8 contacts, two 4-point toe manifolds: 12 + 4 and 24 + 8 cover them, and the 128-VGPR budget holds without scratch again)
template <class T> struct PgsWin { static constexpr int ROWS = T::NCLOS > 0 ? 24 : 12, CONTACTS = T::NCLOS > 0 ? 8 : 4; };
#endif
// `fpos`: bit l set iff lane l's friction bound is positive.  Bullet solves a contact's friction rows only while its normal row carries an
// impulse (`if (totalImpulse > 0)` in btMultiBodyConstraintSolver::solveSingleIteration) and leaves them as they are otherwise -- a contact
// that is only within the margin (a gap row without impulse: every other contact of a walking robot) costs a scalar bit test, no visit.
template <int PGS_REG_ROWS, int PGS_REG_CONTACTS, int I, bool CONE>
DI void pgs_friction_rows(const float* Acol, const float* af, float a0, float a1, float b0, float b1, int nc, float& y, float& lam, float invdiag, float lm,
                          unsigned long long fpos) {
  if constexpr (I < MAXC) {
    if (I >= nc) return;
    constexpr int IN = I + 2 < MAXC ? I + 2 : MAXC - 1;
    float n0 = 0.0f, n1 = 0.0f;
    if constexpr (I + 2 >= PGS_REG_CONTACTS) {
      if (I + 2 < nc) { n0 = Acol[MAXR * fric_lane(IN, 0)]; n1 = Acol[MAXR * fric_lane(IN, 1)]; }  // wave-uniform: rows of existing contacts only
    }
    const float gA = I < PGS_REG_CONTACTS ? af[2 * (I < PGS_REG_CONTACTS ? I : 0)] : a0 * invdiag;
    const float gB = I < PGS_REG_CONTACTS ? af[2 * (I < PGS_REG_CONTACTS ? I : 0) + 1] : a1 * invdiag;
    if ((fpos >> fric_lane(I, 0)) & 1ull) {   // wave-uniform
      if constexpr (CONE) {
        pgs_visit_cone<fric_lane(I, 0), fric_lane(I, 1)>(gA, gB, y, lam, lm);
      } else {
        pgs_visit_friction<fric_lane(I, 0)>(gA, y, lam, lm);
        pgs_visit_friction<fric_lane(I, 1)>(gB, y, lam, lm);
      }
    }
    if constexpr (I + 2 >= PGS_REG_CONTACTS) { pin1(n0); pin1(n1); }
    pgs_friction_rows<PGS_REG_ROWS, PGS_REG_CONTACTS, I + 1, CONE>(Acol, af, b0, b1, n0, n1, nc, y, lam, invdiag, lm, fpos);
  }
}
// fewer than four fixed-bound rows left: one uniform exit test per visit
template <int PGS_REG_ROWS, int RR, int LEFT>
DI void pgs_fixed_tail(const float* ar, float a0, float a1, float a2, int r_fr, float& y, float& lam, float invdiag, float lo0) {
  if constexpr (RR < MAXR && LEFT > 0) {
    if (RR >= r_fr) return;
    pgs_visit<RR>(RR < PGS_REG_ROWS ? ar[RR < PGS_REG_ROWS ? RR : 0] : a0 * invdiag, y, lam, lo0);
    pgs_fixed_tail<PGS_REG_ROWS, RR + 1, LEFT - 1>(ar, a1, a2, 0.0f, r_fr, y, lam, invdiag, lo0);
  }
}
// Fixed-bound rows are visited in GROUPS OF FOUR (compile-time recursion = guaranteed unrolling; readlane / writelane indices and
// LDS offsets are immediates): one uniform exit test per group instead of per visit, and the four A entries of the NEXT group are
// requested from LDS before this group's visits start -- a visit is ~30 cycles of dependent issue, an LDS round trip is more than
// two of them, so the former two-visits-ahead prefetch left every visit waiting.  The last one to three rows take pgs_fixed_tail.
template <int PGS_REG_ROWS, int RR>
DI void pgs_fixed_rows(const float* Acol, const float* ar, float a0, float a1, float a2, float a3, int r_fr, float& y, float& lam, float invdiag, float lo0) {
  if constexpr (RR + 4 <= MAXR) {
    if (RR + 4 <= r_fr) {
      constexpr int R4 = RR + 4 < MAXR ? RR + 4 : MAXR - 1, R5 = RR + 5 < MAXR ? RR + 5 : MAXR - 1;   // past the last row: any
      constexpr int R6 = RR + 6 < MAXR ? RR + 6 : MAXR - 1, R7 = RR + 7 < MAXR ? RR + 7 : MAXR - 1;   // readable row, never used
      float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
      if constexpr (RR + 4 >= PGS_REG_ROWS) { n0 = Acol[MAXR * R4]; n1 = Acol[MAXR * R5]; n2 = Acol[MAXR * R6]; n3 = Acol[MAXR * R7]; }
      constexpr bool reg = RR + 3 < PGS_REG_ROWS;   // PGS_REG_ROWS is a multiple of four: a group is all-register or all-LDS
      pgs_visit<RR>(reg ? ar[reg ? RR : 0] : a0 * invdiag, y, lam, lo0);
      pgs_visit<RR + 1>(reg ? ar[reg ? RR + 1 : 0] : a1 * invdiag, y, lam, lo0);
      pgs_visit<RR + 2>(reg ? ar[reg ? RR + 2 : 0] : a2 * invdiag, y, lam, lo0);
      pgs_visit<RR + 3>(reg ? ar[reg ? RR + 3 : 0] : a3 * invdiag, y, lam, lo0);
      if constexpr (RR + 4 >= PGS_REG_ROWS) { pin1(n0); pin1(n1); pin1(n2); pin1(n3); }   // keeps the optimiser from sinking the reads into the group that uses them
      pgs_fixed_rows<PGS_REG_ROWS, RR + 4>(Acol, ar, n0, n1, n2, n3, r_fr, y, lam, invdiag, lo0);
      return;
    }
  }
  pgs_fixed_tail<PGS_REG_ROWS, RR, 3>(ar, a0, a1, a2, r_fr, y, lam, invdiag, lo0);
}

// MoccaModel.sweep_alternate (Bullet sweeps its non-contact rows last-to-first in the even iterations, btMultiBodyConstraintSolver::
// solveSingleIteration): the even iterations of such a blob visit lanes nnc - 1 .. 0 in a ROLLED loop (dynamic lane: v_readlane with an SGPR
// select, a compare-and-select commit, the gain read from LDS -- a handful of limit rows for the walkers, 6 - 12 closure / planar / limit rows
// for Cassie), then the contact normals nnc .. r_fr - 1 forward through the unrolled visits with a uniform skip below `start`.  The odd
// iterations, and every iteration of a blob without the flag, take pgs_fixed_rows as before.
DI void pgs_reverse_rows(const float* Acol, int nnc, int lane, float& y, float& lam, float invdiag, float lo0) {
#pragma unroll 1
  for (int j = nnc - 1; j >= 0; --j) {
    const float as = Acol[MAXR * j] * invdiag;
    const float nl_ = __builtin_amdgcn_fmed3f(y, lo0, 1e30f);
    const float dl = readlane(nl_ - lam, j);
    lam = lane == j ? nl_ : lam;
    y = fmaf(-as, dl, y);
  }
}
template <int PGS_REG_ROWS, int RR>
DI void pgs_fixed_rows_from(const float* Acol, const float* ar, int start, int r_fr, float& y, float& lam, float invdiag, float lo0) {
  if constexpr (RR < MAXR) {
    if (RR >= r_fr) return;
    if (RR >= start) pgs_visit<RR>(RR < PGS_REG_ROWS ? ar[RR < PGS_REG_ROWS ? RR : 0] : Acol[MAXR * RR] * invdiag, y, lam, lo0);
    pgs_fixed_rows_from<PGS_REG_ROWS, RR + 1>(Acol, ar, start, r_fr, y, lam, invdiag, lo0);
  }
}

// ------------------------------------------------------------------ constraint rows + PGS
// lane = row.  See oracle solve_constraints() for the reference formulation.
//   1. limit-row candidates are compacted with a ballot; contacts come from collide()
//   2. every lane builds its row (force direction, bias, bounds) in registers
//   3. unit-impulse response X = M^-1 J^T by an inward sweep along the row's path(s) and an outward
//      sweep over the tree, re-using S, U, 1/D and the factor of IA0 of the ABA (unrolled over bodies so the
//      per-lane arrays stay in VGPRs)
//   4. Delassus matrix A[r][c] = J_r . X_c, by symmetry: lane c evaluates half of its column and mirrors the values
//      (delassus_dots / delassus_store); the diagonal stays in a register and is stored as zero
//   5. projected Gauss-Seidel in row order on the zero-diagonal matrix: every lane carries its own unclamped impulse,
//      a visit is clamp, subtract, readlane, one-lane commit, fma (pgs_fixed_rows + the friction loop)
//   6. nu += sum_r X_r lambda_r, summed in row order through LDS
// issue priority of a wave with nr constraint rows (thresholds: MOCCA_PARAM_ISSUE_PRIORITY, 6 bits each); nr is wave-uniform (SGPR):
// each branch is s_cmp / s_cbranch around one s_setprio (which ignores EXEC)
DI void set_issue_priority(int nr, int prio) {
  if (nr > ((prio >> 12) & 63)) __builtin_amdgcn_s_setprio(3);
  else if (nr > ((prio >> 6) & 63)) __builtin_amdgcn_s_setprio(2);
  else if (nr > (prio & 63)) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
// Pace priority (StepArgs.pace > 0): the hardware arbitrates equal-priority waves of a SIMD oldest-first, so the four resident waves finish
// one after the other and the last one runs alone -- latency-bound, three quarters of the SIMD's issue slots idle -- while the launch waits
// for it.  Each wave therefore compares the time it has used (s_memtime since its start, kept in the spare word of the base record)
// with the share of the step it has done (`done` of `total` units): estimated finish = elapsed * total / done, against the pace +- 1/8.
// All scalar: one s_memtime, one LDS word, three multiplies, three compares.
// (The pace travels in LDS next to the start time, not in scalar registers: the kernel holds all 102 of them already, and one more value
// that lives across the substeps is spilled to a VGPR lane, which in turn is spilled to scratch.)
#ifndef MOCCA_PACE_ROWUNIT
#define MOCCA_PACE_ROWUNIT 2   // pace units per constraint row (a substep without rows: 64); 0 .. 4 measured: profiles/archive/r04_pace_probe_ru*.jsonl
#endif
#ifndef MOCCA_PACE_SHIFT
#define MOCCA_PACE_SHIFT 4   // width of the priority bands around the pace: 2^-4
#endif
constexpr int PACE_TICK_SHIFT = 6;   // elapsed ticks and the pace are compared in units of 64 ticks
enum : int { L_T0 = L_BASE + 13, L_PACE = L_BASE + 14, L_KEEPWARM = L_BASE + 15 /* StepArgs.persist_warm or a warm-starting blob: same reason */ };
constexpr unsigned PACE_CNT_BITS = 24, PACE_WINDOW = 128;
DI void pace_start(const StepArgs& a, float* L, int lane, int pace) {
  if (lane == 0) {
    int p16 = pace >> PACE_TICK_SHIFT;
    if (pace < 0) {   // self-calibrating: the mean wave time of the last sampled waves (ticks / 64) x (-pace) / 16; no sample yet: row-count priorities
      const unsigned long long v = __hip_atomic_load(a.pace_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned cnt = (unsigned)v & ((1u << PACE_CNT_BITS) - 1u);
      p16 = cnt > 0u ? (int)((float)(v >> PACE_CNT_BITS) / (float)cnt * (float)(-pace) * 0.0625f) : 0;
    }
    L[L_PACE] = __int_as_float(p16);
    if (pace != 0) L[L_T0] = __uint_as_float((unsigned)__builtin_amdgcn_s_memtime());
  }
}
DI void pace_finish(const StepArgs& a, const float* L, int lane, int pace) {   // (after a barrier: lane 0's words are visible to itself anyway)
  if (pace < 0 && lane == 0 && (blockIdx.x % 61u) == 0u) {   // one wave in 61: a sample spread over the XCDs and CUs
    const unsigned el = ((unsigned)__builtin_amdgcn_s_memtime() - __float_as_uint(L[L_T0])) >> PACE_TICK_SHIFT;
    const unsigned long long old = atomicAdd(a.pace_acc, ((unsigned long long)el << PACE_CNT_BITS) | 1ull);
    // Exactly one wave sees the count arrive at the window (every add returns another value); it halves both fields.  Count and sum only
    // grow in between (the next halving is 64 samples away), so neither subtraction borrows: the mean is that of the last 64 .. 128 samples.
    if (((unsigned)old & ((1u << PACE_CNT_BITS) - 1u)) + 1u == PACE_WINDOW) {
      const unsigned long long sum = (old >> PACE_CNT_BITS) + el;
      atomicAdd(a.pace_acc, 0ull - (((sum >> 1) << PACE_CNT_BITS) | (unsigned long long)(PACE_WINDOW / 2)));
    }
  }
}
DI bool pace_on(const float* L) { return uni(__float_as_int(L[L_PACE])) > 0; }
DI void pace_checkpoint(const float* L, int done, int total) {
  const unsigned p16 = (unsigned)uni(__float_as_int(L[L_PACE]));   // pace in units of 64 ticks; 0 = off (wave-uniform branch)
  if (p16 == 0u) return;
  const unsigned el = (unsigned)__builtin_amdgcn_s_memtime() - (unsigned)uni(__float_as_int(L[L_T0]));
  // units of 64 ticks: the products stay below 2^32 with room to spare (Cassie: 2.4 M ticks per step, 50 x (64 + 2 x 48) units: 3e8)
  const unsigned lhs = (el >> PACE_TICK_SHIFT) * (unsigned)total, r = p16 * (unsigned)done;
  if (lhs > r + (r >> MOCCA_PACE_SHIFT)) __builtin_amdgcn_s_setprio(3);
  else if (lhs > r) __builtin_amdgcn_s_setprio(2);
  else if (lhs > r - (r >> MOCCA_PACE_SHIFT)) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
// debug record: fold this substep's twelve words into the step signature (MOCCA_DBG_STEPSIG_*; off the product path)
DI void dbg_fold_step(int32_t* dbg, int lane) {
  if (dbg && lane == 0) {
    unsigned long long h = ((unsigned long long)(unsigned)dbg[17] << 32) | (unsigned)dbg[16];
#pragma unroll 1
    for (int w = 0; w < 12; ++w) h = (h ^ (unsigned long long)(unsigned)dbg[w]) * 0x9E3779B97F4A7C15ull;
    dbg[16] = (int32_t)(unsigned)h; dbg[17] = (int32_t)(unsigned)(h >> 32); dbg[18] += 1;
  }
}
template <class T>
DI void solve_constraints(ModelP M, float* L, int lane, int nc_found, int nc_wanted, unsigned long long ppk, int32_t* dbg, const float* Afac, int prio, int& rows_out) {
  const bool keep_warm = uni(__float_as_int(L[L_KEEPWARM])) != 0;   // (wave-uniform) the slots' normal impulses are wanted after the substep (warm start / diagnostic)
  STAMP_BEGIN;
  // wave-uniform scalars live in SGPRs: loop control becomes s_cmp/s_cbranch instead of exec-mask bookkeeping
  const float dt = unif(M->dt), idt = rcp(dt);
  const int maxr = uni(M->max_rows);
  int nl;
  {
    bool act = false;
    if (lane < 2 * T::NJ) {
      const int b = 1 + (lane >> 1), side = lane & 1;
      const float q = L[L_Q + b];
      const float gap = side == 0 ? q - M->jlo[b] : M->jhi[b] - q;
      const float vel = (side == 0 ? 1.0f : -1.0f) * L[L_NU + 5 + b];
      act = M->limit_at_violation ? !(gap > 0.0f) : gap + dt * vel < M->limit_slack;
    }
    const unsigned long long lm = __ballot(act);
    const int rk = lane_rank(lm);
    if (act && rk < maxr) reinterpret_cast<int*>(L)[L_ROWD + rk] = lane;
    nl = uni(__popcll(lm));
    if (dbg && lane == 0) {
      dbg[5] = (int32_t)(unsigned)lm; dbg[6] = (int32_t)(unsigned)(lm >> 32);
      // cap pressure, CUMULATIVE over the substeps since the host last cleared the record (MOCCA_DBG_CAP_*): how often the 12-contact /
      // 48-row caps -- which Bullet does not have -- drop something, and the largest row count an uncapped solver would have held
      const int nfix = 3 * T::NCLOS + ((T::NCLOS > 0 && M->planar) ? 3 : 0), mc = M->max_contacts;
      const int wanted = nl + nfix + 3 * nc_wanted;
      dbg[12] += nc_wanted > mc ? 1 : 0;
      dbg[13] += nl + nfix + 3 * (nc_wanted < mc ? nc_wanted : mc) > maxr ? 1 : 0;
      dbg[14] += 1;
      if (wanted > dbg[15]) dbg[15] = wanted;
    }
    if (nl > maxr) nl = maxr;
  }
  constexpr int NCL = 3 * T::NCLOS;  // point-to-point closure rows sit between the limit and the contact rows
  // CassieEnv(planar=True): three more bilateral rows hold the base in the x-z plane (omega_x, omega_z, v_y); only the Cassie
  // topology carries them, every other kernel instance sees a compile-time zero
  const int NFIX = NCL + ((T::NCLOS > 0 && uni(M->planar)) ? 3 : 0);
  int nc = uni(nc_found);
  if (nc > (maxr - nl - NFIX) / 3) nc = (maxr - nl - NFIX) / 3;
  if (nc < 0) nc = 0;
  const int nr = nl + NFIX + 3 * nc;
  if (dbg && lane == 0) { dbg[0] = nr; dbg[1] = nl; dbg[2] = nc; }
  // Load balancing across the waves of a SIMD: the launch lasts as long as its slowest wave, and a wave's cost grows with
  // its row count (an env lying on the ground has 48 rows, a standing one ~20).  Issue priority follows the row count,
  // so heavy waves run at nearly their stand-alone speed while light ones -- which have slack -- yield.  nr is in an
  // SGPR: each branch is s_cmp / s_cbranch around one s_setprio (which ignores EXEC).
  if (!pace_on(L)) set_issue_priority(nr, prio);   // (pace priorities replace the row-count ones, substep())
  rows_out = nr;
  wsync();
  STAMP(16);
  if (nr == 0) {  // nothing touches, no limit near: nothing to solve (uniform branch)
    if (dbg && lane == 0) { dbg[8] = 0; dbg[9] = 0; dbg[10] = 0; dbg[11] = 0; }
    if (keep_warm && lane < T::NSLOT) L[L_WARM + lane] = 0.0f;
    wsync();
    return;
  }
  // ---- my row.  Lanes 0 .. r_fr - 1: limit, closure / planar and normal rows (fixed bounds, visited in lane order); friction rows of
  // contact i on the static lanes fric_lane(i, 0 / 1) at the top of the range; the lanes in between own no row.
  const int r = lane;
  const int r_fr = nl + NFIX + nc;            // first lane past the fixed-bound rows
  const int row_gap = MAXR - nr;              // dense row number d >= r_fr sits on lane d + row_gap
  const bool is_fric = r >= MAXR - 2 * nc && r < MAXR;
  const bool has_row = r < r_fr || is_fric;
  const int ci = is_fric ? (MAXR - 1 - r) >> 1 : r - nl - NFIX;   // contact of a normal / friction row
  int kind = -1, ba = 0, bb = -1, jl = -1, slot = -1;
  float F[6] = {0, 0, 0, 0, 0, 0}, sgn = 0, bias = 0, cfm = 0, lam = 0, mu = 0;
  float F2[6] = {0, 0, 0, 0, 0, 0};  // force on body bb (closures: its own pivot; self contacts: same point as F)
  if (r < nl) {
    const int c = reinterpret_cast<int*>(L)[L_ROWD + r];
    const int b = 1 + (c >> 1), side = c & 1;
    const float q = L[L_Q + b];
    const float gap = side == 0 ? q - M->jlo[b] : M->jhi[b] - q;
    kind = 0; jl = b; ba = b; sgn = side == 0 ? 1.0f : -1.0f;
    bias = gap < 0 ? M->erp_noncontact * (-gap) * idt : -gap * idt;
  } else if (T::NCLOS > 0 && r < nl + NCL) {
    // loop closure c, world axis ax: dir . (v(pivot a) - v(pivot b)) = erp (Pb - Pa)/dt, unbounded impulse
    const int c = (r - nl) / 3, ax = (r - nl) % 3;
    ba = M->cl_body_a[c]; bb = M->cl_body_b[c];
    kind = 3;
    float Pa[3], Pb[3], Ra[9], Rb[9], la[3], lb[3];
#pragma unroll
    for (int x = 0; x < 9; ++x) { Ra[x] = L[L_RT + 12 * ba + 4 * (x / 3) + x % 3]; Rb[x] = L[L_RT + 12 * bb + 4 * (x / 3) + x % 3]; }
#pragma unroll
    for (int x = 0; x < 3; ++x) { la[x] = M->cl_point_a[c][x]; lb[x] = M->cl_point_b[c][x]; }
    matvec3(Ra, la, Pa); matvec3(Rb, lb, Pb);
#pragma unroll
    for (int x = 0; x < 3; ++x) { Pa[x] += L[L_RT + 12 * ba + 4 * x + 3]; Pb[x] += L[L_RT + 12 * bb + 4 * x + 3]; }
    float dir[3] = {ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f}, pn[3];
    cross3(Pa, dir, pn);
#pragma unroll
    for (int x = 0; x < 3; ++x) { F[x] = pn[x]; F[3 + x] = dir[x]; }
    cross3(Pb, dir, pn);
#pragma unroll
    for (int x = 0; x < 3; ++x) { F2[x] = pn[x]; F2[3 + x] = dir[x]; }
    const float e = ax == 0 ? Pb[0] - Pa[0] : (ax == 1 ? Pb[1] - Pa[1] : Pb[2] - Pa[2]);
    bias = M->erp_noncontact * e * idt;
  } else if (T::NCLOS > 0 && r < nl + NFIX) {
    // planar base: the base's y axis stays the world's (R e_y = e_y <=> omega_x = omega_z = 0) and y stays where it started;
    // small-angle error of R e_y = (u_x, u_y, u_z): delta_x = u_z, delta_z = -u_x
    const int k = r - nl - NCL;
    const int comp = k == 0 ? 0 : (k == 1 ? 2 : 4);   // entry of nu = [omega; v] the row acts on
    kind = 3; ba = 0; bb = -1;
#pragma unroll
    for (int x = 0; x < 6; ++x) F[x] = x == comp ? 1.0f : 0.0f;
    const float err = k == 0 ? L[L_RT + 9] /* R[2][1] */ : (k == 1 ? -L[L_RT + 1] /* R[0][1] */ : L[L_BASE + 1] - M->init_pos[1]);
    bias = -M->erp_noncontact * err * idt;
  } else if (has_row) {
    const float* ct = L + L_CT + 16 * ci;
    float n[3] = {ct[C_N], ct[C_N + 1], ct[C_N + 2]}, P[3] = {ct[C_P], ct[C_P + 1], ct[C_P + 2]}, dir[3];
    ba = __float_as_int(ct[C_BA]); bb = __float_as_int(ct[C_BB]);
    if (!is_fric) {
      kind = 1;
      dir[0] = n[0]; dir[1] = n[1]; dir[2] = n[2];
      const float depth = ct[C_DEPTH] - unif(M->linear_slop);   // penetration = distance + m_linearSlop (0 in the compiled blobs)
      bias = depth > 0 ? ct[C_ERP] * depth * idt : depth * idt;
      cfm = ct[C_CFM];
      slot = __float_as_int(ct[C_SLOT]);
      const float wsf = unif(M->warmstart);
      lam = (wsf != 0.0f && slot >= 0) ? wsf * L[L_WARM + slot] : 0.0f;   // (compiled blobs: 0 -- Bullet does not warm start multibody contacts)
    } else {
      kind = 2;
      float t1[3], t2[3];
      plane_space(n, t1, t2);
      const bool second = (r & 1) != 0;   // fric_lane(i, 1) is the odd lane (MAXR is even)
#pragma unroll
      for (int x = 0; x < 3; ++x) dir[x] = second ? t2[x] : t1[x];
      mu = ct[C_MU];
    }
    float pn[3];
    cross3(P, dir, pn);
#pragma unroll
    for (int x = 0; x < 3; ++x) { F[x] = pn[x]; F[3 + x] = dir[x]; F2[x] = pn[x]; F2[3 + x] = dir[x]; }
  }
  STAMP(17);
  // ancestor masks travel with the contact records; limit / closure rows use the compile-time table (few distinct bodies)
  unsigned ma = 0u, mb = 0u;
  if (kind == 1 || kind == 2) {
    const float* ct = L + L_CT + 16 * ci;
    ma = __float_as_uint(ct[C_MA]); mb = __float_as_uint(ct[C_MB]);
  } else if (kind >= 0) {
    ma = M->anc_mask[ba];
    mb = (kind == 3 && bb >= 0) ? M->anc_mask[bb] : 0u;
  }
  const float sa = kind >= 1 ? 1.0f : 0.0f, sb = (kind >= 1 && bb >= 0) ? 1.0f : 0.0f;  // base part: F on a, -F2 on b

  STAMP(5);
  // ---- unit response X = M^-1 J^T
  // Jacobian entries go straight to LDS (the J rows sit in the part of the region the ABA no longer
  // needs: link frames / inertias, not S, U, 1/D, the factor of IA0, contacts); w = J nu is accumulated on the fly.
  float X[T::ND];  // the response: base part after the base solve, joint entries as the outward sweep reaches them
  float w = 0;
  const int dense = has_row ? (r < r_fr ? r : r - row_gap) : MAXR;   // dense row number; lanes without a row write the dummy row
  float* Jrow = L + L_J + 28 * dense;
  float pa[6] = {0, 0, 0, 0, 0, 0}, pb[6] = {0, 0, 0, 0, 0, 0};
  // Inward sweep ALONG THE ROW'S OWN PATH: a row only loads the bodies between its body and the base (<= MAXD of them), so
  // the sweep visits path positions, not bodies -- each lane reads the S / U / 1/D of ITS body at that depth (at most MAXW
  // distinct addresses per position: the other lanes' reads are broadcasts).  The innovation u_k = J_k - S_k . p (over D) stays in a
  // register indexed by the path POSITION (static), and the outward sweep -- which is unrolled over bodies -- picks
  // position depth(b) - 1 for the rows whose path holds b.  (A loop over all bodies with wave-uniform skips cost 21 steps for
  // the walker, and a merged two-path variant twice the arithmetic per step.)
  // Rows that act on two bodies (self contacts, loop closures) sweep the second body's path separately (the recursion is
  // linear in the applied force; both meet in the base's right-hand side); without such a row in the wave that pass is skipped.
#ifdef MOCCA_NO_TWO_PATHS  // diagnostic build (round 4's code-size probe, profiles/archive/r04_icache_counters.txt): code-size experiment, wrong with self contacts / closures
  const bool two_paths = false;
#else
  const bool two_paths = T::NCLOS > 0 || __ballot(kind >= 1 && bb >= 0) != 0ull;  // wave-uniform
#endif
  STAMP(18);
  // the packed path of the row's body, from the lane that owns that body in the lane = body layout (body 0: empty path)
  const unsigned long long pka = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(ppk >> 32), ba, 64) << 32) |
                                 (unsigned long long)(unsigned)__shfl((int)(unsigned)ppk, ba, 64);
  {
    float4* jz = reinterpret_cast<float4*>(Jrow);  // joint entries of bodies off the path are zero (entries 0..5: base part, below)
#pragma unroll
    for (int i = 1; i < 7; ++i) jz[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  float pu[T::MAXD], pub[T::MAXD];
#pragma unroll
  for (int k = T::MAXD - 1; k >= 0; --k) {
    const int j = (int)((pka >> (5 * k)) & 31ull);
    const bool valid = j != 0;   // 0: past the end of the path (body 0's record is read, the selects below discard it)
    const int jj = j;
    float S[6], U[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { S[i] = L[L_SV + SVS * jj + i]; U[i] = L[L_SV + SVS * jj + SV_V + i]; }
    float jb = dot6(S, F);
    if (jj == jl) jb = sgn;
    jb = valid ? jb : 0.0f;
    // (at the first position pa is still zero: without fast-math the six products with 0 are not folded)
    const float uu = valid ? (k == T::MAXD - 1 ? jb : jb - dot6(S, pa)) : 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) pa[i] += U[i] * uu;  // U holds V = IA S / D; uu == 0 past the end of the path
    if (valid) Jrow[5 + jj] = jb;
    w += jb * L[L_NU + 5 + jj];
    pu[k] = uu * L[L_SV + SVS * jj + SV_INVD];   // what the outward sweep starts from: u / D
    pin6(pa); pin1(pu[k]); pin1(w);
  }
  if (two_paths) {
    const int bq = bb >= 0 ? bb : 0;
    const unsigned long long pkb = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(ppk >> 32), bq, 64) << 32) |
                                   (unsigned long long)(unsigned)__shfl((int)(unsigned)ppk, bq, 64);
#pragma unroll
    for (int k = T::MAXD - 1; k >= 0; --k) {
      const int j = (int)((pkb >> (5 * k)) & 31ull);
      const bool valid = j != 0;
      const int jj = j;
      float S[6], U[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) { S[i] = L[L_SV + SVS * jj + i]; U[i] = L[L_SV + SVS * jj + SV_V + i]; }
      const float jb = valid ? -dot6(S, F2) : 0.0f;   // the force on the second body is -F2
      const float uu = valid ? (k == T::MAXD - 1 ? jb : jb - dot6(S, pb)) : 0.0f;
#pragma unroll
      for (int i = 0; i < 6; ++i) pb[i] += U[i] * uu;
      if (valid) Jrow[5 + jj] += jb;                  // common ancestors carry both paths' entries
      w += jb * L[L_NU + 5 + jj];
      pub[k] = uu * L[L_SV + SVS * jj + SV_INVD];
      pin6(pb); pin1(pub[k]); pin1(w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < T::MAXD; ++k) pub[k] = 0.0f;
  }
  // Launder the LDS pointer: otherwise the compiler keeps all 21 bodies' S/U loads of the inward sweep live
  // for the outward sweep (273 VGPRs); re-reading 13 broadcast floats per body costs far less than the occupancy.
  STAMP(19);
  int l2off = 0;  // an opaque zero offset, not an opaque pointer: a laundered pointer turns generic and its reads
  asm volatile("" : "+v"(l2off));  // become flat loads (VALU address math, both memory counters) instead of ds_read
  l2off &= ~3;  // tells the optimiser the offset keeps 16-byte alignment: without it every read below became a ds_read2_b32 off its
                // own v_add'ed base (the 8-bit offsets of the two-address form), 136 address adds per substep in the outward sweep
  const float* L2 = L + l2off;
  float a0[6];
  {
    float rhs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float jb = F[i] * sa - F2[i] * sb;
      Jrow[i] = jb;
      w += jb * L[L_NU + i];
      rhs[i] = jb - (pa[i] + pb[i]);
    }
    chol6_solve(Afac, rhs, a0);
#pragma unroll
    for (int i = 0; i < 6; ++i) X[i] = a0[i];
  }
  {
    float acc[T::NB][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[0][i] = a0[i];
#pragma unroll
    for (int b = 1; b < T::NB; ++b) {
      const int p = T::parent(b);
      float S[6], U[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) { S[i] = L2[L_SV + SVS * b + i]; U[i] = L2[L_SV + SVS * b + SV_V + i]; }   // U = V = IA S / D
      const int dpos = T::depth(b) - 1;  // (folds after unrolling) the body's position on every path that holds it
      // pu[dpos] where the row's path holds b, else 0: bit b of the ancestor mask, sign-extended, ANDs the value (v_bfe_i32 + v_and)
      float ub = __uint_as_float(__float_as_uint(pu[dpos]) & (unsigned)__builtin_amdgcn_sbfe((int)ma, b, 1));
      if (two_paths) ub += __uint_as_float(__float_as_uint(pub[dpos]) & (unsigned)__builtin_amdgcn_sbfe((int)mb, b, 1));
      const float qdd = ub - dot6(U, acc[p]);   // u / D - (IA S / D) . a_parent
      X[5 + b] = qdd;
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[b][i] = acc[p][i] + S[i] * qdd;
      pin6(acc[b]); pin1(X[5 + b]);
    }
  }
  STAMP(6);
  wsync();  // all lanes are done with the ABA view: the A matrix may overwrite it
  // ---- Delassus matrix A = J M^-1 J^T, half of it computed, mirrored on store (delassus_dots / delassus_store)
  // A[dense row][dense column]: a lane's column is its dense row number; lanes without a row read column MAXR - 1 (any readable
  // column would do) instead of branching around each access
  const int lc = has_row ? dense : MAXR - 1;
  float diag;
  {
    float av[MAXR / 2 + 1];
    const int cc = has_row ? dense : 0, tmax = nr >> 1;
    delassus_dots<T, 0>(L, X, 28 * cc, 28 * nr, tmax, av);
    diag = has_row ? av[0] : 1.0f;
    wsync();  // every J row has been read: A may overwrite them
    if (has_row) delassus_store<0>(L, cc, nr, tmax, av);
  }
  // a row with a vanishing Jacobian (out-of-plane friction of a planar mechanism's self contact) gets zero gain, as in
  // Bullet (jacDiagABInv = 0 below SIMD_EPSILON), instead of 0 * inf
  const float invdiag = diag + cfm > 1e-12f ? rcp(diag + cfm) : 0.0f;
  // The solver works on A with a ZERO diagonal: each lane keeps y = (bias - sum_{r' != c} A[r'][c] lam_r') / (A_cc + cfm),
  // the value its own impulse would take if unclamped.  A visit of row rr then is
  //     new = clamp(y_rr);  d = new - lam_rr;  lam_rr = new;  y_c -= A[rr][c] / (A_cc + cfm) * d   for every lane c
  // and lane rr needs no special case (its own column entry is the zeroed diagonal): no v_cmp / v_cndmask per visit.
  wsync();
  STAMP(7);
  // warm-start impulses act before the first iteration (normal rows only)
  // (a wave without a warm impulse skips the reads; otherwise two rows per trip, every product added -- a zero impulse adds zero)
  if (__ballot(lam != 0.0f) != 0ull) {
    int rr = nl + NFIX;
    const int rend = nl + NFIX + nc;
#pragma unroll 1
    for (; rr + 2 <= rend; rr += 2) {
      const float g0 = L[L_A + MAXR * rr + lc], g1 = L[L_A + MAXR * rr + MAXR + lc];
      w += g0 * readlane(lam, rr); w += g1 * readlane(lam, rr + 1);
    }
    if (rr < rend) w += L[L_A + MAXR * rr + lc] * readlane(lam, rr);
  }
  STAMP(20);
  // ---- projected Gauss-Seidel: limits, closures, normals (lanes 0 .. r_fr - 1 in order), then the friction rows contact by contact.
  // Limit / closure / normal rows have fixed bounds; a friction row's bound is mu * (current impulse of its normal row): every
  // friction lane fetches its contact's normal impulse once per iteration, after the normal rows were visited (they do not change
  // again before the next iteration).  A visit is ~30 cycles of dependent issue: med3, sub, readlane, fma (pgs_visit*).
  const int iters = uni(M->n_iters);
  const bool cone = uni(M->friction_cone) != 0;
  const float lo0 = kind == 3 ? -1e30f : 0.0f;
  const float* Acol = L + L_A + lc;
  const float* Acol_fr = Acol - MAXR * row_gap;   // friction visits name their rows by LANE (static): lane l holds dense row l - row_gap
  const int nrow_lane = is_fric ? nl + NFIX + ci : 0;   // lane of the normal row this friction row is bounded by
  float y = (bias - w) * invdiag;
  // this lane's gains of the first rows, scaled once (pgs_fixed_rows / pgs_friction_rows); rows that do not exist give unused values
  constexpr int PGS_REG_ROWS = PgsWin<T>::ROWS, PGS_REG_CONTACTS = PgsWin<T>::CONTACTS;
  static_assert(PGS_REG_ROWS % 4 == 0 && PGS_REG_ROWS <= MAXR && PGS_REG_CONTACTS <= MAXC, "register window of the solver");
  float ar[PGS_REG_ROWS], af[2 * PGS_REG_CONTACTS];
#pragma unroll
  for (int k = 0; k < PGS_REG_ROWS; ++k) ar[k] = Acol[MAXR * k] * invdiag;
#pragma unroll
  for (int k = 0; k < PGS_REG_CONTACTS; ++k) {
    af[2 * k] = 0.0f; af[2 * k + 1] = 0.0f;
    if (k < nc) { af[2 * k] = Acol_fr[MAXR * fric_lane(k, 0)] * invdiag; af[2 * k + 1] = Acol_fr[MAXR * fric_lane(k, 1)] * invdiag; }  // wave-uniform
  }
  unsigned long long clamp_sig = 0ull, clamp_last = 0ull;   // debug record only
  const int nnc = nl + NFIX;                                 // rows that are not contacts: lanes 0 .. nnc - 1
  // (compiled into the 64-row ACCURACY instance only -- mocca_create routes a blob with the flag there -- so that the product's instruction
  // stream stays what it was: with the second visit sequence in it the 48-row kernel measured 0.6 % (walker) / 1.5 % (Cassie) slower)
  const bool alt = ALT_SWEEPS && uni(M->sweep_alternate) != 0 && nnc > 1;  // (one such row: last-to-first is first-to-last)
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    // the row counts are laundered per iteration: as loop invariants the optimiser hoisted every `RR >= r_fr` / `I >= nc` exit test of
    // the unrolled visits out of the loop as a 64-bit lane mask each -- ~100 SGPRs, spilled to VGPR lanes and re-read per iteration
    int rf = r_fr, ncc = nc;
    asm volatile("" : "+s"(rf), "+s"(ncc));
    bool fwd = true;
    if constexpr (ALT_SWEEPS) {
      if (alt && !(it & 1)) {   // wave-uniform: the non-contact rows last-to-first, then the normals forward (pgs_reverse_rows)
        int nn = nnc;
        asm volatile("" : "+s"(nn));
        pgs_reverse_rows(Acol, nn, lane, y, lam, invdiag, lo0);
        pgs_fixed_rows_from<PGS_REG_ROWS, 0>(Acol, ar, nn, rf, y, lam, invdiag, lo0);
        fwd = false;
      }
    }
    if (fwd) pgs_fixed_rows<PGS_REG_ROWS, 0>(Acol, ar, 0.0f, 0.0f, 0.0f, 0.0f, rf, y, lam, invdiag, lo0);
    float lm = 0.0f;
    if (ncc > 0) {  // wave-uniform
      float f0 = 0.0f, f1 = 0.0f, g0 = 0.0f, g1 = 0.0f;
      if (PGS_REG_CONTACTS < 1) { f0 = Acol_fr[MAXR * fric_lane(0, 0)]; f1 = Acol_fr[MAXR * fric_lane(0, 1)]; }
      if (PGS_REG_CONTACTS < 2 && ncc > 1) { g0 = Acol_fr[MAXR * fric_lane(1, 0)]; g1 = Acol_fr[MAXR * fric_lane(1, 1)]; }
      const float ln_ = __shfl(lam, nrow_lane, 64);
      lm = mu * ln_;
      const unsigned long long fpos = __ballot(ln_ > 0.0f);
      if (cone) pgs_friction_rows<PGS_REG_ROWS, PGS_REG_CONTACTS, 0, true>(Acol_fr, af, f0, f1, g0, g1, ncc, y, lam, invdiag, lm, fpos);   // wave-uniform
      else pgs_friction_rows<PGS_REG_ROWS, PGS_REG_CONTACTS, 0, false>(Acol_fr, af, f0, f1, g0, g1, ncc, y, lam, invdiag, lm, fpos);
    }
    if (dbg) {  // wave-uniform, off the product path: which rows this iteration left ON a bound (the solver's discrete decisions)
      const float lpart = __shfl_xor(lam, 1, 64);   // a friction lane's partner: the contact's other friction row (lanes 2k, 2k + 1)
      const bool on_circle = lam * lam + lpart * lpart >= lm * lm * (1.0f - 1e-5f);   // cone: the pair sits on the circle (the oracle's expression)
      const bool clamped = has_row && (kind == 2 ? (cone ? on_circle : fabsf(lam) == lm) : (kind != 3 && lam == 0.0f));
      clamp_last = __ballot(clamped);
      if constexpr (MAXR != 48) {
        // The record names friction rows by the lanes of the 48-row instance (46 - 2i, 47 - 2i; include/mocca.h): move this instance's there.
        // A blob whose caps exceed 48 rows / 12 contacts (it can only run the 64-row instance) is recorded in that instance's own lanes,
        // 62 - 2i / 63 - 2i: the oracle follows the blob's caps.
        const bool native = MAXR > 48 && (maxr > 48 || uni(M->max_contacts) > 12);
        if (!native && nc > 0) {
          const int lo = MAXR - 2 * nc;
          clamp_last = (clamp_last & ((1ull << lo) - 1ull)) | ((clamp_last >> lo) << (48 - 2 * nc));
        }
      }
      clamp_sig = ((clamp_sig << 7) | (clamp_sig >> 57)) ^ clamp_last;
    }
  }
  if (dbg && lane == 0) {
    dbg[8] = (int32_t)(unsigned)clamp_last; dbg[9] = (int32_t)(unsigned)(clamp_last >> 32);
    dbg[10] = (int32_t)(unsigned)clamp_sig; dbg[11] = (int32_t)(unsigned)(clamp_sig >> 32);
  }
  STAMP(8);
  // ---- apply: nu += sum_r X_r lambda_r, summed in row order through LDS
  wsync();
  if (has_row) {
#pragma unroll
    for (int d = 0; d < T::ND; ++d) L[L_XL + 28 * dense + d] = X[d] * lam;
  }
  if (keep_warm && lane < T::NSLOT) L[L_WARM + lane] = 0.0f;
  wsync();
  if (keep_warm && kind == 1 && slot >= 0) L[L_WARM + slot] = lam;
  if (lane < T::ND) {
    // dense row order: fixed-bound rows, then the friction rows by lane.  Four reads in flight per round trip, summed in the same order:
    // one read per trip made this loop an LDS-latency chain as long as the row count, on exactly the waves the launch waits for
    float s = 0;
    int rr = 0;
#pragma unroll 1
    for (; rr + 4 <= nr; rr += 4) {
      const float x0 = L[L_XL + 28 * rr + lane], x1 = L[L_XL + 28 * rr + 28 + lane], x2 = L[L_XL + 28 * rr + 56 + lane], x3 = L[L_XL + 28 * rr + 84 + lane];
      s += x0; s += x1; s += x2; s += x3;
    }
#pragma unroll 1
    for (; rr < nr; ++rr) s += L[L_XL + 28 * rr + lane];
    L[L_NU + lane] += s;
  }
  wsync();
  STAMP(9);
}

// ------------------------------------------------------------------ integration
template <class T>
DI void integrate(ModelP M, float* L, int lane) {
  const float dt = M->dt;
  if (lane >= 1 && lane < T::NB) {
    float v = L[L_NU + 5 + lane];
    const float mx = M->max_qd;
    v = v > mx ? mx : (v < -mx ? -mx : v);
    L[L_QD + lane] = v;
    L[L_Q + lane] += dt * v;
  }
  if (lane == 0) {
    float om[3] = {L[L_NU], L[L_NU + 1], L[L_NU + 2]}, vl[3] = {L[L_NU + 3], L[L_NU + 4], L[L_NU + 5]};
#pragma unroll
    for (int i = 0; i < 3; ++i) { L[L_BASE + 10 + i] = om[i]; L[L_BASE + 7 + i] = vl[i]; L[L_BASE + i] += dt * vl[i]; }
    // quaternion exponential map dq = (om sin(th/2)/|om|, cos(th/2)), th = |om| dt.  x = th/2 is below 0.5 rad unless the base spins
    // faster than 240 rad/s: there sin(x)/x and cos(x) are short even series in x^2 (truncation < 1e-13, i.e. correctly rounded
    // fp32 up to the last bit), with no square root, no division and no small-angle branch; libm's sinf / cosf -- 400
    // instructions of argument reduction on one busy lane per substep -- stay as the fallback.
    const float hdt = 0.5f * dt, x2 = dot3(om, om) * hdt * hdt;
    float dq[4];
    if (x2 < 0.25f) {
      const float sinc = 1.0f + x2 * (-1.0f / 6 + x2 * (1.0f / 120 + x2 * (-1.0f / 5040 + x2 * (1.0f / 362880 + x2 * (-1.0f / 39916800)))));
      dq[3] = 1.0f + x2 * (-0.5f + x2 * (1.0f / 24 + x2 * (-1.0f / 720 + x2 * (1.0f / 40320 + x2 * (-1.0f / 3628800 + x2 * (1.0f / 479001600))))));
      const float k = hdt * sinc;
      dq[0] = om[0] * k; dq[1] = om[1] * k; dq[2] = om[2] * k;
    } else {
      const float wn = sqrtf(dot3(om, om)), sn = sinf(wn * hdt) / wn;
      dq[0] = om[0] * sn; dq[1] = om[1] * sn; dq[2] = om[2] * sn; dq[3] = cosf(wn * hdt);
    }
    const float q0 = L[L_BASE + 3], q1 = L[L_BASE + 4], q2 = L[L_BASE + 5], q3 = L[L_BASE + 6];
    float nq[4];
    nq[0] = dq[3] * q0 + dq[0] * q3 + dq[1] * q2 - dq[2] * q1;
    nq[1] = dq[3] * q1 - dq[0] * q2 + dq[1] * q3 + dq[2] * q0;
    nq[2] = dq[3] * q2 + dq[0] * q1 - dq[1] * q0 + dq[2] * q3;
    nq[3] = dq[3] * q3 - dq[0] * q0 - dq[1] * q1 - dq[2] * q2;
    const float nn = rsq(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);  // v_rsq_f32, 1 ulp (the norm is 1 +- 1e-6)
#pragma unroll
    for (int i = 0; i < 4; ++i) L[L_BASE + 3 + i] = nq[i] * nn;
  }
  wsync();
}

// sin and cos of a joint angle: quadrant reduction (Cody-Waite, two constants: exact for the |q| < 200 rad a hinge can reach) +
// odd / even series on |r| <= pi/4, < 1 ulp at these magnitudes; libm's sincosf carries a large-argument path (Payne-Hanek)
// that costs instruction-cache and issue slots on every call.
DI void fast_sincos(float q, float* s, float* c) {
  const float k = rintf(q * 0.636619772f);
  float r = fmaf(-k, 1.57079601e+00f, q);
  r = fmaf(-k, 3.13916473e-07f, r);
  r = fmaf(-k, 5.39030253e-15f, r);
  const float r2 = r * r;
  const float sr = r + r * r2 * (-1.0f / 6 + r2 * (1.0f / 120 + r2 * (-1.0f / 5040 + r2 * (1.0f / 362880))));
  const float cr = 1.0f + r2 * (-0.5f + r2 * (1.0f / 24 + r2 * (-1.0f / 720 + r2 * (1.0f / 40320 + r2 * (-1.0f / 3628800)))));
  const int qd = (int)k & 3;
  const float ss = (qd & 1) ? cr : sr, cc = (qd & 1) ? sr : cr;
  *s = (qd & 2) ? -ss : ss;
  *c = ((qd + 1) & 2) ? -cc : cc;
}

// lane = joint: sin/cos of the joint angle and everything a path walk needs about the joint, as one 16-float LDS
// record (layout at L_JR0).  Every lane of a walk visits up to MAXD joints; staging keeps the model's global loads and the
// Rodrigues formula out of that loop (they were re-done per (lane, path step), with the load latency on the chain).
template <class T>
DI void stage_joints(ModelP M, float* L, int lane) {
  if (lane < T::NB) {   // lane 0 stages the identity (blob joint 0, q[0] = 0)
    const int j = lane;
    float s, cq;
    fast_sincos(L[L_Q + j], &s, &cq);
    const float t = 1.0f - cq;
    float ax[3], jr[9], Rq[9], Tl[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) ax[i] = M->jaxis[j][i];
#pragma unroll
    for (int i = 0; i < 9; ++i) jr[i] = M->jrot[j][i];
    Rq[0] = cq + t * ax[0] * ax[0];          Rq[1] = t * ax[0] * ax[1] - s * ax[2];   Rq[2] = t * ax[0] * ax[2] + s * ax[1];
    Rq[3] = t * ax[0] * ax[1] + s * ax[2];   Rq[4] = cq + t * ax[1] * ax[1];          Rq[5] = t * ax[1] * ax[2] - s * ax[0];
    Rq[6] = t * ax[0] * ax[2] - s * ax[1];   Rq[7] = t * ax[1] * ax[2] + s * ax[0];   Rq[8] = cq + t * ax[2] * ax[2];
    matmul3(jr, Rq, Tl);
    float4* rec = reinterpret_cast<float4*>(L + L_SV + SVS * j);
    rec[0] = make_float4(Tl[0], Tl[1], Tl[2], Tl[3]);
    rec[1] = make_float4(Tl[4], Tl[5], Tl[6], Tl[7]);
    rec[2] = make_float4(Tl[8], M->jpos[j][0], M->jpos[j][1], M->jpos[j][2]);
    rec[3] = make_float4(ax[0], ax[1], ax[2], L[L_QD + j]);
  }
  wsync();
}

// one physics substep (what stepSimulation does numSubSteps times, bullet_utils.py:346-353)
template <class T, int TASK>
DI ContactFlags substep(ModelP M, float* L, int lane, const float* ter, int next_step_index,
                        unsigned long long ppk, int32_t* dbg, int prio, int& rows_out,
                        const HeightFieldArgs hfa = HeightFieldArgs{nullptr, 0, 0, 0.0f}, int sidx = 0, int nsub = 1, int* cover_out = nullptr) {
  // pace checkpoints: a substep counts 64 units + MOCCA_PACE_ROWUNIT per constraint row (the row count of the substep before stands in until
  // this one's is known) -- 20 after the collision pass, 36 after the ABA, all at its end: an env with many rows has more of its step
  // ahead of it at the same point of the program, and is given priority BEFORE it falls behind
  const int per0 = 64 + MOCCA_PACE_ROWUNIT * rows_out, done0 = sidx * per0, total = nsub * per0;
  STAMP(30);
  stage_joints<T>(M, L, lane);
  STAMP(29);
  int nc = 0, nc_wanted = 0;
  float Afac[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // Cholesky factor of the base's articulated inertia (aba_passes -> solve_constraints)
  ContactFlags fl = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (COMPACT) {
    // compact layout: collision detection sits BETWEEN the two phases of the walk -- it needs the body frames only, and the link
    // inertias of phase 2 overwrite the frames, the geom points and the candidate list (same arithmetic, another order of the phases)
    static_assert(!COMPACT || T::NCLOS == 0, "closure rows read the body frames after the ABA: not in the compact layout");
    int wl = lane;
    asm volatile("" : "+v"(wl));
    walk_phase1<T>(L, wl, ppk);
    wsync();
    STAMP(0);
    geom_points<T>(M, L, lane);
    wsync();
    STAMP(15);
#ifndef MOCCA_SKIP_COLLIDE
    fl = collide<T, TASK>(M, L, lane, ter, next_step_index, &nc, dbg, &nc_wanted, hfa, cover_out);
#endif
    wsync();   // (a compiler fence: phase 2 stores over what the collision pass read)
    const int wb = wl < T::NB ? wl : 0;
    WalkConsts wk;
    walk_consts<true>(M, wb, wk);
    walk_phase2<T, true>(M, L, wl, wb, ppk, wk);
    wsync();
  } else {
    walk_kinematics<T, true>(M, L, lane, ppk);
    wsync();
    STAMP(0);
    geom_points<T>(M, L, lane);
    wsync();
    STAMP(15);
#ifndef MOCCA_SKIP_COLLIDE  // profiling builds only (tools/ablate_time.sh): results are wrong by construction
    fl = collide<T, TASK>(M, L, lane, ter, next_step_index, &nc, dbg, &nc_wanted, hfa, cover_out);
#endif
  }
  STAMP(1);
  pace_checkpoint(L, done0 + 20, total);
#ifndef MOCCA_SKIP_ABA
  aba_passes<T>(M, L, lane, ppk, Afac);
#endif
  STAMP(2);
  pace_checkpoint(L, done0 + 36, total);
#ifndef MOCCA_SKIP_SOLVE
  solve_constraints<T>(M, L, lane, nc, nc_wanted, ppk, dbg, Afac, prio, rows_out);
#endif
  if (dbg) { __builtin_amdgcn_s_waitcnt(0); dbg_fold_step(dbg, lane); }   // (lane 0 wrote the twelve words above; it reads them back itself)
  STAMP(3);
#ifdef MOCCA_DUMMY_VALU  // experiment: is the kernel VALU-issue-bound?  (tools/ab.sh variants)
  {
    float x = (float)lane;
#ifdef MOCCA_DUMMY_ILP   // the same number of FMAs as four INDEPENDENT chains: issue slots without the dependent-issue latency
    float y = x + 1.0f, z = x + 2.0f, u = x + 3.0f;
#pragma unroll 1
    for (int i = 0; i < MOCCA_DUMMY_VALU; ++i) { x = fmaf(x, 1.0001f, 0.5f); y = fmaf(y, 0.9999f, -0.5f); z = fmaf(z, 1.0001f, 0.5f); u = fmaf(u, 0.9999f, -0.5f); }
    asm volatile("" :: "v"(y), "v"(z), "v"(u));
#else
#pragma unroll 1
    for (int i = 0; i < MOCCA_DUMMY_VALU; ++i) { x = fmaf(x, 1.0001f, 0.5f); x = fmaf(x, 0.9999f, -0.5f); x = fmaf(x, 1.0001f, 0.5f); x = fmaf(x, 0.9999f, -0.5f); }
#endif
    asm volatile("" :: "v"(x));
  }
#endif
  integrate<T>(M, L, lane);
  STAMP(4);
  pace_checkpoint(L, (sidx + 1) * (64 + MOCCA_PACE_ROWUNIT * rows_out), nsub * (64 + MOCCA_PACE_ROWUNIT * rows_out));
  return fl;
}

// ------------------------------------------------------------------ task layer
DI void quat_to_rpy(const float* q, float* rpy) {  // pybullet.getEulerFromQuaternion, bullet_utils.py:84
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float sarg = -2 * (x * z - w * y);
  if (sarg <= -0.99999f) { rpy[1] = -1.5707963267948966f; rpy[0] = 0; rpy[2] = 2 * atan2f(x, -y); }
  else if (sarg >= 0.99999f) { rpy[1] = 1.5707963267948966f; rpy[0] = 0; rpy[2] = 2 * atan2f(-x, y); }
  else {
    rpy[0] = atan2f(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
    rpy[1] = asinf(sarg);
    rpy[2] = atan2f(2 * (x * y + w * z), w * w + x * x - y * y - z * z);
  }
}

// roll, pitch and the HEADING (cos yaw, sin yaw) of the base.  Nothing downstream needs the yaw angle itself: R_z(-yaw) v, the target
// bearing sin / cos (atan2(dy, dx) - yaw) and the stepping-stone deltas are rational in (cos yaw, sin yaw), which come straight from the
// quaternion -- no atan2f / sinf / cosf chain (six libm calls per env.step for the Custom task, three more per stepping stone).
DI void quat_to_rp_heading(const float* q, float* rp, float* cy, float* sy) {  // same branches as quat_to_rpy
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float sarg = -2 * (x * z - w * y);
  if (sarg <= -0.99999f) { rp[1] = -1.5707963267948966f; rp[0] = 0; const float yaw = 2 * atan2f(x, -y); *cy = cosf(yaw); *sy = sinf(yaw); }
  else if (sarg >= 0.99999f) { rp[1] = 1.5707963267948966f; rp[0] = 0; const float yaw = 2 * atan2f(-x, y); *cy = cosf(yaw); *sy = sinf(yaw); }
  else {
    rp[0] = atan2f(2 * (y * z + w * x), w * w - x * x - y * y + z * z);
    rp[1] = asinf(sarg);
    const float A = 2 * (x * y + w * z), B = w * w + x * x - y * y - z * z, n = rsq(A * A + B * B);  // A^2 + B^2 = cos^2(pitch) > 2e-5 here
    *cy = B * n; *sy = A * n;
  }
}

struct RobotObs { float rpy[3]; /* [2] unused: see cy, sy */ float cy, sy; int jal; float spd; float height; bool finite; };

// WalkerBase.calc_state (robots.py:42-95): writes obs[0 .. 6+2NJ+NFEET) ; needs kinematics done (L_FEET).
// lane j < NJ keeps its scaled joint speed in the return value for the energy term.
template <class T>
DI RobotObs robot_obs(ModelP M, float* L, int lane, float fc0, float fc1, float* obs, float fc2 = 0.0f, float fc3 = 0.0f) {
  RobotObs ro;
  float q[4] = {L[L_BASE + 3], L[L_BASE + 4], L[L_BASE + 5], L[L_BASE + 6]};
  quat_to_rp_heading(q, ro.rpy, &ro.cy, &ro.sy);
  ro.rpy[2] = 0.0f;
  // body_vel = R_z(-yaw) v  (robots.py:63-71)
  const float vx = ro.cy * L[L_BASE + 7] + ro.sy * L[L_BASE + 8], vy = ro.cy * L[L_BASE + 8] - ro.sy * L[L_BASE + 7], vz = L[L_BASE + 9];
  float minz = fminf(L[L_FEET + 2], L[L_FEET + 5]);
  if constexpr (T::NFEET > 2) minz = fminf(minz, fminf(L[L_FEET + 8], L[L_FEET + 11]));
  const float height = L[L_BASE + 2] - minz;
  auto clip5 = [](float x) { return x > 5.f ? 5.f : (x < -5.f ? -5.f : x); };
  float nrm = 0, sp = 0;
  bool fin = true;
  if (lane < T::NJ) {
    const int b = lane + 1;
    const float ang = L[L_Q + b], lo = M->jlo[b], wt = M->jhi[b] - lo;
    nrm = 2 * (ang - lo) / wt - 1;
    sp = 0.1f * L[L_QD + b];
    obs[6 + lane] = clip5(nrm);
    obs[6 + T::NJ + lane] = clip5(sp);
    fin = isfinite(nrm) && isfinite(sp);
  }
  const float head[6] = {height, vx, vy, vz, ro.rpy[0], ro.rpy[1]};
  bool hfin = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) hfin = hfin && isfinite(head[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) obs[i] = clip5(head[i]);
    obs[6 + 2 * T::NJ] = fc0;
    obs[6 + 2 * T::NJ + 1] = fc1;
    if constexpr (T::NFEET > 2) { obs[6 + 2 * T::NJ + 2] = fc2; obs[6 + 2 * T::NJ + 3] = fc3; }
  }
  ro.jal = __popcll(__ballot(lane < T::NJ && fabsf(nrm) > 0.99f));
  ro.spd = sp;
  ro.height = clip5(height);
  ro.finite = (__ballot(!fin) == 0ull) && hfin;
  return ro;
}

struct TaskRegs {  // uniform across the wave
  float wt[3], linpot, angpot, stopf, fc0, fc1, fc2, fc3, dist, angle, gain, prevx, initz;
  int close, done, t, episode, draw, mirrored, nsi, trc, stop, setstop, cur, istep;
  int cover;   // Stepper: cover mask of the last step's contacts (word 26, as a float; ContactFlags / cover_targets) -- what reset() reads stale
};
// (`cassie`: words 38 / 39 -- initial_z, istep -- are Cassie's alone and sit in the third 64-byte line of the 160-byte record: the other tasks
// neither read nor write them, which keeps that line out of a walker step's HBM traffic where it is not shared with the next env's record)
DI void load_task(const uint32_t* tk, TaskRegs& t, bool quadruped = false, bool cassie = false) {
  auto f = [&](int i) { return __uint_as_float(tk[i]); };
  t.wt[0] = f(T_WTX); t.wt[1] = f(T_WTY); t.wt[2] = f(T_WTZ); t.linpot = f(T_LINPOT); t.angpot = f(T_ANGPOT);
  t.close = (int)tk[T_CLOSE]; t.stopf = f(T_STOPF); t.done = (int)tk[T_DONE]; t.t = (int)tk[T_T];
  t.episode = (int)tk[T_EPISODE]; t.draw = (int)tk[T_DRAW]; t.mirrored = (int)tk[T_MIRROR];
  t.fc0 = f(T_FC0); t.fc1 = f(T_FC1); t.dist = f(T_DIST); t.angle = f(T_ANGLE);
  t.fc2 = quadruped ? f(T_FC2) : 0.0f; t.fc3 = quadruped ? f(T_FC3) : 0.0f;
  t.nsi = (int)tk[T_NSI]; t.trc = (int)tk[T_TRC]; t.stop = (int)tk[T_STOP]; t.setstop = (int)tk[T_SETSTOP];
  t.cur = (int)tk[T_CUR]; t.gain = f(T_GAIN); t.prevx = f(T_PREVX);
  t.initz = cassie ? f(T_INITZ) : 0.0f; t.istep = cassie ? (int)tk[T_ISTEP] : 0;
  t.cover = 0;
}
enum : int { T_COVER = 26 };
DI void load_task_cover(const uint32_t* tk, TaskRegs& t) { t.cover = (int)__uint_as_float(tk[T_COVER]); }
DI void store_task_cover(uint32_t* tk, const TaskRegs& t) { tk[T_COVER] = __float_as_uint((float)t.cover); }
DI void store_task(uint32_t* tk, const TaskRegs& t, bool quadruped = false, bool cassie = false) {
  auto u = [](float x) { return __float_as_uint(x); };
  tk[T_WTX] = u(t.wt[0]); tk[T_WTY] = u(t.wt[1]); tk[T_WTZ] = u(t.wt[2]); tk[T_LINPOT] = u(t.linpot); tk[T_ANGPOT] = u(t.angpot);
  tk[T_CLOSE] = (uint32_t)t.close; tk[T_STOPF] = u(t.stopf); tk[T_DONE] = (uint32_t)t.done; tk[T_T] = (uint32_t)t.t;
  tk[T_EPISODE] = (uint32_t)t.episode; tk[T_DRAW] = (uint32_t)t.draw; tk[T_MIRROR] = (uint32_t)t.mirrored;
  tk[T_FC0] = u(t.fc0); tk[T_FC1] = u(t.fc1); tk[T_DIST] = u(t.dist); tk[T_ANGLE] = u(t.angle);
  if (quadruped) { tk[T_FC2] = u(t.fc2); tk[T_FC3] = u(t.fc3); }
  tk[T_NSI] = (uint32_t)t.nsi; tk[T_TRC] = (uint32_t)t.trc; tk[T_STOP] = (uint32_t)t.stop; tk[T_SETSTOP] = (uint32_t)t.setstop;
  tk[T_CUR] = (uint32_t)t.cur; tk[T_GAIN] = u(t.gain); tk[T_PREVX] = u(t.prevx);
  if (cassie) { tk[T_INITZ] = u(t.initz); tk[T_ISTEP] = (uint32_t)t.istep; }
}

// calc_potential, env_locomotion.py:143-158
// *dist = distance to the walk target; *cd, *sd = dist * cos / sin (angle_to_target), angle_to_target = atan2(dy, dx) - yaw
DI void calc_potential(ModelP M, const float* L, TaskRegs& t, const RobotObs& ro, float* dist, float* cd, float* sd) {
  const float dx = t.wt[0] - L[L_BASE], dy = t.wt[1] - L[L_BASE + 1];
  *cd = dx * ro.cy + dy * ro.sy;
  *sd = dy * ro.cy - dx * ro.sy;
  *dist = sqrtf(dx * dx + dy * dy);
  t.linpot = -(*dist) / M->control_dt;
  t.angpot = *dist > 0.0f ? *cd / *dist : ro.cy;   // cos(angle_to_target); atan2(0, 0) = 0
}
template <bool INJECT>
DI void randomize_target(const StepArgs& a, int env, TaskRegs& t, bool eval_mode) {  // env_locomotion.py:67-74
  if (eval_mode) { t.dist = 4; t.angle = 0; }
  else {
    const float u0 = draw_u<INJECT>(a, env, t.episode, t.draw);
    const float u1 = draw_u<INJECT>(a, env, t.episode, t.draw + 1);
    t.draw += 2;
    t.dist = 3 + 2 * u0;
    t.angle = -1.5707963267948966f + 3.141592653589793f * u1;
  }
  const float u2 = draw_u<INJECT>(a, env, t.episode, t.draw);
  t.draw += 1;
  t.stopf = u2 < 0.5f ? 30.0f : 60.0f;
}
DI void softsign_tail(float s, float c, float* o2) {  // s, c = dist * sin / cos (angle_to_target), env_locomotion.py:124-127
  o2[0] = s / (1 + fabsf(s));
  o2[1] = c / (1 + fabsf(c));
}

// delta_to_k_targets, env_locomotion.py:712-759: lookbehind j rows before the next step, then lookahead 2 rows from it, indices
// clamped at both ends; a stop repeats the next step.  Lane 0 writes the 5 (j + 2) floats; sets walk_target (index -1).
DI void delta_to_k_targets(ModelP M, const float* L, const float* ter, TaskRegs& t, const RobotObs& ro, int lane, float* out) {
  const int N = t.nsi, TT = MOCCA_MAX_TERRAIN_STEPS, j = M->lookbehind, nt = j + 2;
#pragma unroll 1
  for (int i = 0; i < nt; ++i) {
    int v = N - j + i;
    if (t.stop && i >= j) v = N;
    v = v < 0 ? 0 : (v > TT - 1 ? TT - 1 : v);
    const float* tt = ter + 6 * v;
    if (i == nt - 1) { t.wt[0] = tt[0]; t.wt[1] = tt[1]; t.wt[2] = tt[2]; }
    const float dx = tt[0] - L[L_BASE], dy = tt[1] - L[L_BASE + 1], dz = tt[2] - L[L_BASE + 2];
    if (lane == 0) {   // sin / cos (atan2(dy, dx) - yaw) * dist
      out[5 * i + 0] = dy * ro.cy - dx * ro.sy;
      out[5 * i + 1] = dx * ro.cy + dy * ro.sy;
      out[5 * i + 2] = dz;
      out[5 * i + 3] = tt[4];
      out[5 * i + 4] = tt[5];
    }
  }
}

// generate_step_placements, env_locomotion.py:395-441: 100 uniforms (5 x 20) -> 20 x 6 table in `ter`.
// Lanes draw in parallel (counter-based RNG) and each builds one row; the cumulative sums run over the lanes in row order.
template <bool INJECT>
DI void generate_terrain(const StepArgs& a, ModelP M, int env, TaskRegs& t, float* L, float* ter, int lane) {
  const float DEG = 3.14159265358979323846f / 180.0f, HP = 1.5707963267948966f;
  const int N = MOCCA_MAX_TERRAIN_STEPS;
  const int cur = t.cur > 9 ? 9 : t.cur;
  const float ratio = (float)cur / 9.0f;
  float* u = L + L_J;  // scratch (solver view is idle during a reset)
  for (int k = lane; k < 5 * N; k += 64) u[k] = draw_u<INJECT>(a, env, t.episode, t.draw + k);
  t.draw += 5 * N;
  wsync();
  // lane = table row for everything but the four running sums (heading, x, y, z), which walk the rows in order with one v_readlane per
  // row and quantity -- the summation order of the reference's np.cumsum.  (One lane running the whole table -- four libm calls per row
  // on a 20-step dependent chain -- took 35 k cycles, at the end of the very waves a launch waits for.)
  {
    const int i = lane < N ? lane : N - 1;   // the other lanes shadow the last row and store nothing
    const float d0 = M->dist_range[0], d1 = M->dist_range[1], sep = M->init_step_separation, dx_min = M->step_radius * 2.5f;
    const float dist_lo = d0, dist_hi = d0 + (d1 - d0) * cur / 9;
    const float yaw_lo = -M->yaw_range_deg * ratio * DEG, yaw_hi = M->yaw_range_deg * ratio * DEG;
    const float pit_lo = -M->pitch_range_deg * ratio * DEG + HP, pit_hi = M->pitch_range_deg * ratio * DEG + HP;
    const float tl_lo = -M->tilt_range_deg * ratio * DEG, tl_hi = M->tilt_range_deg * ratio * DEG;
    float dr = dist_lo + (dist_hi - dist_lo) * u[i];
    float dphi = yaw_lo + (yaw_hi - yaw_lo) * u[N + i];
    float dth = pit_lo + (pit_hi - pit_lo) * u[2 * N + i];
    float xt = tl_lo + (tl_hi - tl_lo) * u[3 * N + i];
    float yt = tl_lo + (tl_hi - tl_lo) * u[4 * N + i];
    if (i == 0) { dr = 0; dphi = 0; dth = HP; }
    if (i == 1 || i == 2) { dr = sep; dphi = 0; dth = HP; }
    if (i < 3) { xt = 0; yt = 0; }
    float phi = 0, run = 0;
#pragma unroll 1
    for (int k = 0; k < N; ++k) { run += readlane(dphi, k); if (lane == k) phi = run; }
    if (lane >= N) phi = run;
    float sth, cth, sph, cph;
    fast_sincos(dth, &sth, &cth); fast_sincos(phi, &sph, &cph);
    float dx = dr * sth * cph;
    const float dy = dr * sth * sph, dz = dr * cth;
    if (i >= 2) {
      const float ax = fabsf(dx), mx = ax > dx_min ? ax : dx_min;
      const float sg = dx > 0 ? 1.0f : (dx < 0 ? -1.0f : 0.0f);
      dx = sg * (mx < d1 ? mx : d1);
    }
    float x = 0, y = 0, z = 0, rx = 0, ry = 0, rz = 0;
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
      rx += readlane(dx, k); ry += readlane(dy, k); rz += readlane(dz, k);
      if (lane == k) { x = rx; y = ry; z = rz; }
    }
    if (lane < N) {
      ter[6 * i] = x; ter[6 * i + 1] = y; ter[6 * i + 2] = z; ter[6 * i + 3] = phi; ter[6 * i + 4] = xt; ter[6 * i + 5] = yt;
    }
    if (lane == 0) { ter[120] = 0.0f; ter[121] = 1.0f; ter[122] = 2.0f; ter[123] = 3.0f; }
  }
  wsync();
}

// env.reset() for one env (lane-parallel); leaves the new state in LDS and writes obs.
DI int live_curriculum(const StepArgs& a, int env) {  // env = index in this handle
  if (a.curriculum_v) { const int c = (int)a.curriculum_v[env]; return c < 0 ? 0 : (c > 9 ? 9 : c); }
  return a.curriculum;
}
DI bool live_eval_mode(const StepArgs& a, int env) { return a.eval_mode_v ? a.eval_mode_v[env] != 0.0f : a.eval_mode != 0; }

template <class T, int TASK, bool INJECT = false>
DI void reset_env(const StepArgs& a, ModelP M, float* L, float* ter, int env, int lane, TaskRegs& t,
                  float* obs) {
  const int ep = t.episode + 1, cur = t.cur;
  // Walker3DStepperEnv.reset reads the contact manifolds of the episode that just ended (MOCCA_TASKF_STALE_RESET_CONTACTS)
  const float sfc[4] = {t.fc0, t.fc1, t.fc2, t.fc3};
  const int snsi = t.nsi, scover = t.cover;
  t = TaskRegs{};
  t.episode = ep;
  if (TASK == MOCCA_TASK_WALKER3D_STEPPER && (M->task_flags & MOCCA_TASKF_STALE_RESET_CONTACTS)) {
    t.fc0 = sfc[0]; t.fc1 = sfc[1]; t.fc2 = sfc[2]; t.fc3 = sfc[3];   // robot.feet_contact[:] = info[:, 0] (:656): the first step's observation (:525)
    const int npl = M->n_planks;
    int reached = cover_targets(scover, snsi, npl, 0) | cover_targets(scover, snsi, npl, 1);
    if constexpr (T::NFEET > 2) reached |= cover_targets(scover, snsi, npl, 2) | cover_targets(scover, snsi, npl, 3);
    if (reached) t.trc = 1;   // target_reached_count += 1 from 0 (:484,661); below 2: nothing advances
  }
  t.cur = TASK == MOCCA_TASK_WALKER3D_STEPPER ? live_curriculum(a, env - a.env_offset) : cur;
  t.gain = a.gain_v ? a.gain_v[env - a.env_offset] : a.gain;  // robot.applied_gain persists across resets (robots.py:16,33)
  if (TASK == MOCCA_TASK_WALKER3D_CUSTOM) {
    randomize_target<INJECT>(a, env, t, live_eval_mode(a, env - a.env_offset));
    float sa, ca;
    fast_sincos(t.angle, &sa, &ca);   // |angle| <= pi / 2: no large-argument path needed
    t.wt[0] = t.dist * ca;
    t.wt[1] = t.dist * sa;
    t.wt[2] = 1.0f;
  } else if (TASK == MOCCA_TASK_WALKER3D_STEPPER) {
    t.gain = M->gain_cur[0] + (M->gain_cur[1] - M->gain_cur[0]) * t.cur / 9;           // applied_gain_curriculum[curriculum], :369,489
  }
  t.mirrored = draw_u<INJECT>(a, env, t.episode, t.draw) < 0.5f;
  t.draw += 1;
  if (lane >= 1 && lane < T::NB) {
    const int b = lane;
    int src = b;  // robots.py:182-188 mirror: swap right/left, negate abdomen z/x
    float sgn = 1.0f;
    if (t.mirrored) {
      // both tables in one batch of scalar loads (rolled, every entry was its own ~200-cycle round trip -- at the end of exactly the
      // waves the launch waits for: a fallen robot is a heavy wave AND a reset)
      const int j = b - 1, nms = M->n_mirror_side, nmn = M->n_mirror_neg;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int mr = M->mirror_right[k], ml = M->mirror_left[k];
        if (k < nms) {
          if (mr == j) src = ml + 1;
          if (ml == j) src = mr + 1;
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) if (k < nmn && M->mirror_neg[k] == j) sgn = -1.0f;
    }
    const float base = sgn * M->init_q[src];
    float qn = base;
    if (a.random_pose) {  // robots.py:190-194: deviation (drawn here only), normalise, clip to +-0.95, back to radians
      const float ds = -0.1f + 0.2f * draw_u<INJECT>(a, env, t.episode, t.draw + (b - 1));
      const float wt = M->jhi[b] - M->jlo[b], bs = M->jlo[b];
      float ps = 2 * (base + ds - bs) / wt - 1;
      ps = ps < -0.95f ? -0.95f : (ps > 0.95f ? 0.95f : ps);
      qn = wt * (ps + 1) / 2 + bs;
    }
    L[L_Q + b] = qn;
    L[L_QD + b] = 0.0f;
  }
  if (a.random_pose) t.draw += T::NJ;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { L[L_BASE + i] = M->init_pos[i]; L[L_BASE + 7 + i] = M->init_vel[i]; L[L_BASE + 10 + i] = 0; }  // robot_init_velocity, :92,493
#pragma unroll
    for (int i = 0; i < 4; ++i) L[L_BASE + 3 + i] = M->init_quat[i];
  }
  if (lane < MOCCA_MAX_SLOTS) L[L_WARM + lane] = 0.0f;
  wsync();
  stage_joints<T>(M, L, lane);
  walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
  wsync();
  const int nbo = 6 + 2 * T::NJ + T::NFEET;
  RobotObs ro = robot_obs<T>(M, L, lane, 0.0f, 0.0f, obs);
  float dist, cd, sd;
  if (TASK == MOCCA_TASK_WALKER3D_PLANNER) {
    // Walker3DPlannerEnv.reset (:1060-1073): AFTER robot.reset, the target: xy ~ U(-16, 16)^2, z = the height field under it
    const float R = M->target_range;
    const float ux = draw_u<INJECT>(a, env, t.episode, t.draw), uy = draw_u<INJECT>(a, env, t.episode, t.draw + 1);
    t.draw += 2;
    t.wt[0] = -R + 2.0f * R * ux;
    t.wt[1] = -R + 2.0f * R * uy;
    t.wt[2] = hf_height_at(a.hf, a.hf_rows, a.hf_cols, a.hf_scale, t.wt[0], t.wt[1]);
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
    if (lane == 0) softsign_tail(sd, cd, obs + nbo);
  } else if (TASK == MOCCA_TASK_WALKER3D_CUSTOM) {
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
    if (lane == 0) {
      softsign_tail(sd, cd, obs + nbo);
      if (M->task_flags & MOCCA_TASKF_RESET_TAIL_ZERO) { obs[nbo] = 0.0f; obs[nbo + 1] = 0.0f; }  // Walker2DCustomEnv.reset, :299-300
    }
  } else {
    generate_terrain<INJECT>(a, M, env, t, L, ter, lane);
    t.nsi = M->lookbehind;                                                             // :499
    delta_to_k_targets(M, L, ter, t, ro, lane, obs + nbo);
    calc_potential(M, L, t, ro, &dist, &cd, &sd);
  }
  t.prevx = L[L_BASE];
}

// ---------------- Cassie task layer (env_cassie.py:238-276, 348-479; mocap / phase variants :481-660) ----------------
// frame of the reference motion at mocap_time() = istep * control_step / llc_frame_skip (:359-360), in double precision like
// the reference: int((t mod T) / T * n) is a floor, fp32 time would pick the neighbouring frame now and then
DI int traj_frame(const StepArgs& a, ModelP M, int istep, float* phase = nullptr) {
  const double t = (double)istep * a.traj_cstep / (double)M->n_llc, T = a.traj_tmax;
  if (phase) *phase = (float)fmod(t / T, 1.0);                                          // CassiePhaseMoccaEnv.get_obs, :639
  const int i = (int)(fmod(t, T) / T * (double)a.traj_n);
  return i < a.traj_n - 1 ? i : a.traj_n - 1;
}

// Cassie.calc_state (:238-276) on the state in LDS (kinematics done): everything the observation and reward variants need.
struct CassieState {
  float rpy[3], vel[3];   // body_rpy; body_velocity = R_z(-yaw) v
  float nrm, sp;          // lane k < n_ordered: joint_angles[k] (normalised, float32) and joint_speeds[k]
  float height;           // pelvis z - lowest toe COM z
  bool finite;            // np.isfinite(robot_state).all(), :472
};
DI CassieState cassie_state(ModelP M, const float* L, int lane, float initial_z) {
  CassieState cs;
  const int no = M->n_ordered;
  float q[4] = {L[L_BASE + 3], L[L_BASE + 4], L[L_BASE + 5], L[L_BASE + 6]};
  quat_to_rpy(q, cs.rpy);
  const float yaw = cs.rpy[2], cy = cosf(-yaw), sy = sinf(-yaw);
  cs.vel[0] = cy * L[L_BASE + 7] - sy * L[L_BASE + 8];
  cs.vel[1] = sy * L[L_BASE + 7] + cy * L[L_BASE + 8];
  cs.vel[2] = L[L_BASE + 9];
  const float head[6] = {L[L_BASE + 2] - initial_z, cs.vel[0], cs.vel[1], cs.vel[2], cs.rpy[0], cs.rpy[1]};
  bool fin = true;
#pragma unroll
  for (int i = 0; i < 6; ++i) fin = fin && isfinite(head[i]);
  bool jf = true;
  cs.nrm = 0.0f; cs.sp = 0.0f;
  if (lane < no) {
    const int b = M->ordered_body[lane];
    const float lo = M->jlo[b], hi = M->jhi[b], mid = 0.5f * (lo + hi);
    cs.nrm = 2 * (L[L_Q + b] - mid) / (hi - lo);  // bullet_utils.py:212-216
    cs.sp = L[L_QD + b];
    jf = isfinite(cs.nrm) && isfinite(cs.sp);
  }
  cs.finite = fin && (__ballot(!jf) == 0ull);
  cs.height = L[L_BASE + 2] - fminf(L[L_FEET + 2], L[L_FEET + 5]);
  return cs;
}
// CassieEnv.get_obs (:416-431): robot_state (6 + 14 + 14) + the walk target in the heading frame (2)
DI void cassie_obs(ModelP M, const float* L, int lane, const CassieState& cs, float initial_z, float* obs) {
  const int no = M->n_ordered;
  if (lane < no) { obs[6 + lane] = cs.nrm; obs[6 + no + lane] = cs.sp; }
  if (lane == 0) {
    obs[0] = L[L_BASE + 2] - initial_z; obs[1] = cs.vel[0]; obs[2] = cs.vel[1]; obs[3] = cs.vel[2]; obs[4] = cs.rpy[0]; obs[5] = cs.rpy[1];
    const float tx = M->cassie_target[0], ty = M->cassie_target[1];
    const float dth = atan2f(ty - L[L_BASE + 1], tx - L[L_BASE]) - cs.rpy[2], c = cosf(dth), sn = sinf(dth);
    obs[6 + 2 * no] = c * tx + sn * ty;
    obs[6 + 2 * no + 1] = -sn * tx + c * ty;
  }
}
// rad_joint_angles = to_radians(joint_angles) (:243, 208-210): back from the float32 normalised angle
DI float cassie_rad(ModelP M, int k, float nrm) {
  const int b = M->ordered_body[k];
  const float lo = M->jlo[b], hi = M->jhi[b];
  return (hi - lo) * (nrm + 1.0f) / 2 + lo;
}
// CassieMoccaEnv.get_obs (:607-627) + CassiePhaseMoccaEnv (:636-642) + CassiePhaseMirrorEnv (:657-660), 42 floats:
//   y z | qw qx qy qz (quaternion of body_rpy) | rad_joint_angles 14 | body_velocity 3 | body_angular_speed 3 | jvel 14 | phase_l phase_r
// mirrored (mode 2, phase_l > 0.5): obs[left + right] = obs[right + left], then obs[neg + sideneg] *= -1 with the class's index
// lists (:555-571,633-634,648-655): left = 6..12, 26..32, 40; right = 13..19, 33..39, 41; negated 0 3 5 21 23 25 | 6 7 26 27
DI void cassie_mocap_obs(ModelP M, const float* L, int lane, const CassieState& cs, float jvel, float phase_l, float* obs) {
  const int no = M->n_ordered;
  const float phase_r = fmodf(phase_l + 0.5f, 1.0f);
  const bool flip = M->cassie_mode == MOCCA_CASSIE_PHASE_MIRROR && phase_l > 0.5f;
  auto put = [&](int i, float v) {
    if (flip) {
      if ((i >= 6 && i < 13) || (i >= 26 && i < 33)) i += 7;
      else if ((i >= 13 && i < 20) || (i >= 33 && i < 40)) i -= 7;
      else if (i == 40) i = 41;
      else if (i == 41) i = 40;
      if (i == 0 || i == 3 || i == 5 || i == 21 || i == 23 || i == 25 || i == 6 || i == 7 || i == 26 || i == 27) v = -v;
    }
    obs[i] = v;
  };
  if (lane < no) {
    put(6 + lane, cassie_rad(M, lane, cs.nrm));
    put(26 + lane, jvel);
  }
  if (lane == 0) {
    // pybullet.getQuaternionFromEuler(body_rpy) (:608), x y z w  [UNVERIFIED-BULLET: standard ZYX composition]
    const float hr = 0.5f * cs.rpy[0], hp = 0.5f * cs.rpy[1], hy = 0.5f * cs.rpy[2];
    const float cr = cosf(hr), sr = sinf(hr), cp = cosf(hp), sp = sinf(hp), cy = cosf(hy), sy = sinf(hy);
    const float qx = sr * cp * cy - cr * sp * sy, qy = cr * sp * cy + sr * cp * sy, qz = cr * cp * sy - sr * sp * cy,
                qw = cr * cp * cy + sr * sp * sy;
    put(0, L[L_BASE + 1]); put(1, L[L_BASE + 2]);
    put(2, qw); put(3, qx); put(4, qy); put(5, qz);
    put(20, cs.vel[0]); put(21, cs.vel[1]); put(22, cs.vel[2]);
    put(23, L[L_BASE + 10]); put(24, L[L_BASE + 11]); put(25, L[L_BASE + 12]);
    put(40, phase_l); put(41, phase_r);
  }
}
// CassieMocapRewEnv.compute_rewards (:495-531): six exp(-k * penalty) terms, weighted.  jvel = this lane's finite-difference
// joint speed (k < n_ordered); frame = the motion's frame at the NEW istep.
DI float cassie_mocap_reward(const StepArgs& a, ModelP M, const float* L, int lane, const CassieState& cs, float jvel, int frame) {
  const int npow = M->n_ctrl - 2;
  float dj = 0.0f, dv = 0.0f;
  if (lane < npow) {  // [powered_joint_inds]: the controlled joints minus the two springs
    const int oi = M->ctrl_oidx[lane];
    // the lane that owns ordered joint oi holds its nrm / jvel: fetch them
    dj = a.traj[(size_t)frame * MOCCA_TRAJ_STRIDE + oi];
    dv = a.traj[(size_t)frame * MOCCA_TRAJ_STRIDE + 14 + oi];
  }
  const int src = lane < npow ? M->ctrl_oidx[lane] : 0;
  const float my_rad = cassie_rad(M, lane < M->n_ordered ? lane : 0, cs.nrm);
  const float o_rad = __shfl(my_rad, src, 64), o_jv = __shfl(jvel, src, 64);
  if (lane < npow) { dj -= o_rad; dv -= o_jv; } else { dj = 0.0f; dv = 0.0f; }
  const float joint_penalty = sqrtf(wave_sum(dj * dj)), jvel_penalty = sqrtf(wave_sum(dv * dv));
  const float ve = cs.vel[0] - M->mocap_speed, vel_error = ve * ve;
  const float orientation = cs.rpy[0] * cs.rpy[0] + cs.rpy[1] * cs.rpy[1] + cs.rpy[2] * cs.rpy[2];
  const float w0 = L[L_BASE + 10], w1 = L[L_BASE + 11], w2 = L[L_BASE + 12], angular = w0 * w0 + w1 * w1 + w2 * w2;
  const float cy_ = L[L_BASE + 1] - M->init_pos[1], cz_ = L[L_BASE + 2] - M->init_pos[2], com = cy_ * cy_ + cz_ * cz_;  // base_position[1:], :515
  return M->mocap_w[0] * expf(-4.0f * vel_error) + M->mocap_w[1] * expf(-4.0f * joint_penalty) + M->mocap_w[2] * expf(-0.4f * jvel_penalty) +
         M->mocap_w[3] * expf(-4.0f * orientation) + M->mocap_w[4] * expf(-4.0f * angular) + M->mocap_w[5] * expf(-4.0f * com);
}
DI float cassie_potential(ModelP M, const float* L) {  // calc_potential :348-354
  const float dx = M->cassie_target[0] - L[L_BASE], dy = M->cassie_target[1] - L[L_BASE + 1];
  return -sqrtf(dx * dx + dy * dy) / M->control_dt;
}
// CassieEnv.reset (:362-378): nominal pose at rest.  CassieMoccaEnv.reset (:585-599): istep = np_random.randint(0, 10000) -- one
// draw of the episode's stream, kept only under rsi (:364) -- then joints / joint speeds / rod angles of the motion at mocap_time(),
// base moving at initial_velocity (:552).  The new filtered joint speeds (self.jvel, :357) are left in L_JVEL.
template <class T, bool INJECT = false>
DI void cassie_reset_env(const StepArgs& a, ModelP M, float* L, int env, int lane, TaskRegs& t, float* obs) {
  const int ep = t.episode + 1;
  t = TaskRegs{};
  t.episode = ep;
  t.gain = 1.0f;
  const int mode = M->cassie_mode;
  int frame = 0;
  float phase = 0.0f;
  if (mode != MOCCA_CASSIE_PLAIN) {
    const float u = draw_u<INJECT>(a, env, t.episode, t.draw);
    t.draw += 1;
    int is = (int)(u * 10000.0f);
    is = is > 9999 ? 9999 : is;
    t.istep = M->cassie_rsi ? is : 0;
    frame = traj_frame(a, M, t.istep, &phase);
  }
  if (lane >= 1 && lane < T::NB) { L[L_Q + lane] = M->init_q[lane]; L[L_QD + lane] = 0.0f; }
  if (lane < MOCCA_MAX_CTRL) L[L_JVEL + lane] = 0.0f;
  wsync();
  if (mode != MOCCA_CASSIE_PLAIN) {
    const float* fr = a.traj + (size_t)frame * MOCCA_TRAJ_STRIDE;
    if (lane < M->n_ordered) {
      const int b = M->ordered_body[lane];
      L[L_Q + b] = fr[lane]; L[L_QD + b] = fr[14 + lane]; L[L_JVEL + lane] = fr[14 + lane];
    }
    if (lane < 4) { L[L_Q + M->rod_body[lane]] = fr[28 + lane]; L[L_QD + M->rod_body[lane]] = 0.0f; }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { L[L_BASE + i] = M->init_pos[i]; L[L_BASE + 7 + i] = M->init_vel[i]; L[L_BASE + 10 + i] = 0; }
#pragma unroll
    for (int i = 0; i < 4; ++i) L[L_BASE + 3 + i] = M->init_quat[i];
  }
  if (lane < MOCCA_MAX_SLOTS) L[L_WARM + lane] = 0.0f;
  wsync();
  t.initz = L[L_BASE + 2];
  stage_joints<T>(M, L, lane);
  walk_kinematics<T, false>(M, L, lane, T::path_packed(lane < T::NB ? lane : 0));
  wsync();
  const CassieState cs = cassie_state(M, L, lane, t.initz);
  if (mode == MOCCA_CASSIE_PLAIN) cassie_obs(M, L, lane, cs, t.initz, obs);
  else cassie_mocap_obs(M, L, lane, cs, lane < MOCCA_MAX_CTRL ? L[L_JVEL + lane] : 0.0f, phase, obs);
  t.linpot = cassie_potential(M, L);
}

// `warm` (wave-uniform): the per-slot normal impulses travel with the record.  A blob that does not warm-start its contact rows never reads
// them (they start every step as zeros in LDS) and writes them only on request (StepArgs.persist_warm): 2 x 136 B per env-step of dead
// traffic otherwise.
DI void load_dyn(const float* st, float* L, int lane, int nj, int nslots, bool warm = true) {
  if (lane < 13) L[L_BASE + lane] = st[lane];
  if (lane < nj) { L[L_Q + 1 + lane] = st[13 + lane]; L[L_QD + 1 + lane] = st[13 + nj + lane]; }
  if (warm) { if (lane < nslots) L[L_WARM + lane] = st[13 + 2 * nj + lane]; }
  else if (lane < MOCCA_MAX_SLOTS) L[L_WARM + lane] = 0.0f;
}
DI void store_dyn(float* st, const float* L, int lane, int nj, int nslots, bool warm = true) {
  if (lane < 13) st[lane] = L[L_BASE + lane];
  if (lane < nj) { st[13 + lane] = L[L_Q + 1 + lane]; st[13 + nj + lane] = L[L_QD + 1 + lane]; }
  if (warm && lane < nslots) st[13 + 2 * nj + lane] = L[L_WARM + lane];
}

}  // namespace MOCCA_NS
