// rexsim.hip -- kernels + C ABI (include/rexsim.h) of the MI355X-native batched Rex simulator.
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC rexsim.hip -o librexsim_hip.so
//
// Kernel map
//   rex_step_kernel     one env.step() per lane: action -> staged gait -> Bezier/IK targets ->
//                       action_repeat x (motor model + restated stepSimulation) -> reward / done /
//                       observation (+ optional in-launch reset).  State is read once and written
//                       once per env.step (SoA, coalesced); constraint rows live in LDS.
//   rex_settle_kernel   the reference's 100 + 500 substep reset motion (rex.py:314-323), run once.
//   rex_reset_kernel    snapshot restore + per-episode draws (walk_env.py:125-154).
//   rex_ik/motor/gait   controller-only kernels for parity tests of the controller half.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

#include "rex_device.h"
#include "rex_arm_device.h"
#include "rex_controller.h"

namespace rex {

// INIT_POSES (model/rex_constants.py:10-22), motor order FL,FR,RL,RR x (shoulder, leg, foot)
__device__ __forceinline__ float pose_stand(int j) {
  const int k = j % 3;
  return k == 0 ? 0.0f : (k == 1 ? -0.88643435f : 1.30197369f);
}
__device__ __forceinline__ float pose_stand_ol(int j) {
  const int k = j % 3;
  return k == 0 ? (((j / 3) & 1) ? -0.15192765f : 0.15192765f) : (k == 1 ? -0.90412283f : 1.48156545f);
}

struct EnvState {
  PhysState ph;
  GaitState gait;
  float target, end_time, aux;
  uint32_t flags;
  int32_t steps, episode;
  uint32_t motor_en;
  uint32_t overheat[18]; // one counter per motor in registers; packed 2 x u16 per state word in HBM
  uint32_t hist;         // observation-history ring: bits 0-7 newest slot, bits 8-15 fill
  int sweeps;            // solver sweeps this env ran in this launch (transient: regrouping key)
};

// Persistent-state word layout for NM motors (include/rexsim.h spells out NM = 12 as enum RexStateWord; mark='arm'
// has NM = 18: the q / qd blocks and the overheat block grow, everything else keeps its order)
template <int NM>
struct Lay {
  static constexpr int Q = 13, QD = 13 + NM, PHI = 13 + 2 * NM, LASTT = PHI + 1, ALPHA = PHI + 2, TARGET = PHI + 3,
                       ENDTIME = PHI + 4, AUX = PHI + 5, FLAGS = PHI + 6, STEPS = PHI + 7, EPISODE = PHI + 8, MOTOR_EN = PHI + 9,
                       OVERHEAT = PHI + 10, HIST = OVERHEAT + NM / 2, WORDS = HIST + 1;
};
static_assert(Lay<12>::PHI == REX_S_PHI && Lay<12>::FLAGS == REX_S_FLAGS && Lay<12>::OVERHEAT == REX_S_OVERHEAT &&
              Lay<12>::HIST == REX_S_HIST && Lay<12>::WORDS == REX_STATE_WORDS, "layout must match include/rexsim.h");

// word w of env i at a 32-bit element offset from the block's base (rex_create checks words * n < 2^30): the loads
// and stores use the saddr + 32-bit voffset form, and no per-word 64-bit address has to stay in vector registers
// between load_env and store_env
__device__ __forceinline__ float ldw(const float* st, int n, int w, int i) { return st[(unsigned)(w * n + i)]; }
__device__ __forceinline__ uint32_t ldi(const float* st, int n, int w, int i) { return __float_as_uint(ldw(st, n, w, i)); }
__device__ __forceinline__ void stw(float* st, int n, int w, int i, float v) { st[(unsigned)(w * n + i)] = v; }
__device__ __forceinline__ void sti(float* st, int n, int w, int i, uint32_t v) { stw(st, n, w, i, __uint_as_float(v)); }

template <int NM>
__device__ __forceinline__ void load_env(const float* st, int n, int i, EnvState& e) {
  using Y = Lay<NM>;
#pragma unroll
  for (int k = 0; k < 3; ++k) { e.ph.pos[k] = ldw(st, n, REX_S_POS + k, i); e.ph.lin[k] = ldw(st, n, REX_S_LINVEL + k, i); e.ph.ang[k] = ldw(st, n, REX_S_ANGVEL + k, i); }
#pragma unroll
  for (int k = 0; k < 4; ++k) e.ph.quat[k] = ldw(st, n, REX_S_QUAT + k, i);
#pragma unroll
  for (int j = 0; j < NM; ++j) { e.ph.q[j] = ldw(st, n, Y::Q + j, i); e.ph.qd[j] = ldw(st, n, Y::QD + j, i); }
  e.gait.phi = ldw(st, n, Y::PHI, i); e.gait.last_time = ldw(st, n, Y::LASTT, i); e.gait.alpha = ldw(st, n, Y::ALPHA, i);
  e.target = ldw(st, n, Y::TARGET, i); e.end_time = ldw(st, n, Y::ENDTIME, i); e.aux = ldw(st, n, Y::AUX, i);
  e.flags = ldi(st, n, Y::FLAGS, i); e.steps = (int32_t)ldi(st, n, Y::STEPS, i); e.episode = (int32_t)ldi(st, n, Y::EPISODE, i);
  e.motor_en = ldi(st, n, Y::MOTOR_EN, i);
  e.hist = ldi(st, n, Y::HIST, i);
#pragma unroll
  for (int k = 0; k < NM / 2; ++k) {
    const uint32_t w = ldi(st, n, Y::OVERHEAT + k, i);
    e.overheat[2 * k] = w & 0xFFFFu; e.overheat[2 * k + 1] = w >> 16;
  }
}

template <int NM>
__device__ __forceinline__ void store_env(float* st, int n, int i, const EnvState& e) {
  using Y = Lay<NM>;
#pragma unroll
  for (int k = 0; k < 3; ++k) { stw(st, n, REX_S_POS + k, i, e.ph.pos[k]); stw(st, n, REX_S_LINVEL + k, i, e.ph.lin[k]); stw(st, n, REX_S_ANGVEL + k, i, e.ph.ang[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) stw(st, n, REX_S_QUAT + k, i, e.ph.quat[k]);
#pragma unroll
  for (int j = 0; j < NM; ++j) { stw(st, n, Y::Q + j, i, e.ph.q[j]); stw(st, n, Y::QD + j, i, e.ph.qd[j]); }
  stw(st, n, Y::PHI, i, e.gait.phi); stw(st, n, Y::LASTT, i, e.gait.last_time); stw(st, n, Y::ALPHA, i, e.gait.alpha);
  stw(st, n, Y::TARGET, i, e.target); stw(st, n, Y::ENDTIME, i, e.end_time); stw(st, n, Y::AUX, i, e.aux);
  sti(st, n, Y::FLAGS, i, e.flags); sti(st, n, Y::STEPS, i, (uint32_t)e.steps); sti(st, n, Y::EPISODE, i, (uint32_t)e.episode);
  sti(st, n, Y::MOTOR_EN, i, e.motor_en);
  sti(st, n, Y::HIST, i, e.hist);
#pragma unroll
  for (int k = 0; k < NM / 2; ++k) sti(st, n, Y::OVERHEAT + k, i, e.overheat[2 * k] | (e.overheat[2 * k + 1] << 16));
}

// ---- PyBullet quaternion conventions ----
__device__ __forceinline__ void quat_to_euler(const float* q, float* rpy) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  const float sarg = -2.0f * (x * z - w * y);
  if (sarg <= -0.99999f) { rpy[1] = -0.5f * kPi; rpy[0] = 0.0f; rpy[2] = 2.0f * atan2_fast(x, -y); }
  else if (sarg >= 0.99999f) { rpy[1] = 0.5f * kPi; rpy[0] = 0.0f; rpy[2] = 2.0f * atan2_fast(-x, y); }
  else {
    rpy[1] = asin_fast(sarg);
    rpy[0] = atan2_fast(2.0f * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[2] = atan2_fast(2.0f * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}
// third row (R20, R21, R22) of the matrix of the quaternion rebuilt from Euler angles
// (Rex.GetBaseOrientation, rex.py:530-537, then getMatrixFromQuaternion)
__device__ __forceinline__ void euler_to_row2(const float* rpy, float& r20, float& r21, float& r22) {
  float sr, cr, sp, cp, sy, cy;
  sincos_fast(rpy[0] * 0.5f, sr, cr); sincos_fast(rpy[1] * 0.5f, sp, cp); sincos_fast(rpy[2] * 0.5f, sy, cy);
  float x = sr * cp * cy - cr * sp * sy, y = cr * sp * cy + sr * cp * sy;
  float z = cr * cp * sy - sr * sp * cy, w = cr * cp * cy + sr * sp * sy;
  const float nn = rsqrtf(x * x + y * y + z * z + w * w);
  x *= nn; y *= nn; z *= nn; w *= nn;
  const float d = x * x + y * y + z * z + w * w, s = 2.0f / d;
  const float xs = x * s, ys = y * s, zs = z * s;
  r20 = x * zs - w * ys; r21 = y * zs + w * xs; r22 = 1.0f - (x * xs + y * ys);
}

// ---- Philox4x32-10 ----
__device__ __forceinline__ void philox4x32(uint32_t* c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = h1 ^ c[1] ^ k0, n1 = l1, n2 = h0 ^ c[3] ^ k1, n3 = l0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// Four standard normal draws (Box-Muller on one Philox block) for the sensor-noise model: Rex._AddSensorNoise
// (model/rex.py:765-769) draws np.random.normal afresh in every getter call; here a getter call site of a step is one or
// more Philox blocks keyed by (seed; episode, global env, 16 + block, step).
__device__ __forceinline__ void gauss4(uint32_t seed_lo, uint32_t seed_hi, int gidx, int episode, int step, int block, float* z) {
  uint32_t ctr[4] = {(uint32_t)episode, (uint32_t)gidx, 16u + (uint32_t)block, (uint32_t)step};
  philox4x32(ctr, seed_lo, seed_hi);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = (float)((ctr[2 * p] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(ctr[2 * p + 1]);
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    sincos_fast(6.28318530717958648f * u2, sn, cs);
    z[2 * p] = r * cs; z[2 * p + 1] = r * sn;
  }
}
// call sites of a step (blocks): orientation read by the turn env's goal test, by the reward, by is_fallen, by the
// observation; angular rates of the observation; then NM-wide reads (5 blocks each): reward torques, reward velocities,
// observed motor angles
enum { kNzGoal = 0, kNzRewardRpy = 1, kNzFallenRpy = 2, kNzObsRpy = 3, kNzObsRate = 4, kNzTorque = 8, kNzVelocity = 16, kNzAngle = 24 };

struct DevCfg {
  int32_t n, env_index_base, task, signal, action_repeat, iterations;
  float dt, kp, kd, res_thr;
  int32_t backwards;
  float target_position;
  uint32_t seed_lo, seed_hi;
  int32_t auto_reset, max_steps;
  float w_dist, w_energy, w_drift, w_shake;
  int32_t action_dim, obs_dim;
  float target_orient, init_orient;
  int32_t orient_fixed;
  int32_t pose_index;
  float pose_value;
  int32_t range_normalize;
  const float* terrain;      // [n_terrain][256*256] raw vertex heights (nullptr: plane only)
  const float* terrain_mid;  // [n_terrain]
  int32_t n_terrain;
  float* hist;               // [100][hist_words][n] observation history (nullptr: no latency model)
  int32_t hist_words;        // 3 NM + 7 words per record: q, qd, observed torque, base quaternion, base angular velocity
  float pd_latency, control_latency;
  // int(latency / time_step) and the blend weight of the older slot (rex.py:747-751), taken on the host in double on
  // the decimal values the caller wrote: the float quotient 0.02f / 0.001f is 19.999998
  int32_t pd_slots, control_slots;
  float pd_alpha, control_alpha;
  int32_t reset_substeps;    // int(0.5 / time_step), rex.py:319 (the float quotient 0.5f / 0.001f truncates to 499)
  const float* body_params;  // [3][n] word-major: base mass scale, leg mass scale, foot friction (nullptr: 1, 1, 0.5)
  float act_lo, act_hi;      // Box bounds of the env's action space (host: rex_create)
  float gait_clock;          // wall-clock seconds per simulated second seen by GaitPlanner.loop (gait_planner.py:108-110)
  // REX_TASK_MIXED: the tasks of the mix (task_mix bits, ascending), their number, and the largest action_repeat /
  // solver sweep cap among them (wave-uniform loop bounds; every env stops at its own)
  int32_t mix_task[5], n_mix, max_repeat, max_iterations;
  float mass_lo, mass_hi, mu_lo, mu_hi;   // per-reset randomisation ranges (lo == hi == 0: off)
  // large batches: envs are regrouped into waves by the solver sweeps they needed in the previous step (a wave sweeps
  // until the slowest of its envs has converged): wave slot k works on env perm[k]; sweeps[i] = this step's count of env i
  const int32_t* perm; int32_t* sweeps;
  // rex_set_timing(3): device-side launch duration -- every workgroup folds its start / end wall-clock tick (100 MHz
  // constant clock, s_memrealtime) into clock[0] (min) / clock[1] (max); nullptr otherwise
  unsigned long long* clock;
  float noise[5];            // observation_noise_stdev (rex.py:22,765-769): angles, velocities, torques, rpy, rpy rates
  int32_t noise_on;          // any of them > 0
  HfGeom geo;                // heightfield grid geometry
  int32_t hf_stride;         // floats per field of the pool
  float init_z;              // drop height of the reset (terrain.py:14-20)
  float anchor;              // on_rack: rex::kRackAnchor, else 0
  float obs_hi_ang, obs_hi_rate;
};

// per-task constants of the reference env classes (walk_env.py:34-40,104-114, gallop_env.py:45-53,
// 119-130, turn_env.py:33-39,100-110, poses_env.py:38-44,115-117, standup_env.py:32-38,99-101)
__host__ __device__ __forceinline__ int task_action_repeat(int task) { return (task == REX_TASK_GALLOP || task == REX_TASK_POSES) ? 6 : 5; }
__host__ __device__ __forceinline__ float task_action_bound(int task, int signal) {   // Box(low = -b, high = +b); gallop's is inverted
  if (task == REX_TASK_WALK) return signal == REX_SIGNAL_IK ? 0.4f : 0.01f;
  if (task == REX_TASK_GALLOP) return signal == REX_SIGNAL_IK ? -0.4f : -0.3f;
  if (task == REX_TASK_TURN) return 0.01f;
  return 0.1f;
}
__host__ __device__ __forceinline__ float task_energy_weight(int task) { return task == REX_TASK_GALLOP ? 0.005f : 0.0005f; }

// the task env `gidx` runs for its whole life in a REX_TASK_MIXED batch: a draw from its own Philox stream
__device__ __forceinline__ int mixed_task_of(const DevCfg& c, int gidx) {
  uint32_t ctr[4] = {0xFFFFFFFFu, (uint32_t)gidx, 2u, 0u};
  philox4x32(ctr, c.seed_lo, c.seed_hi);
  const int k = (int)(ctr[0] % (uint32_t)c.n_mix);
  return k == 0 ? c.mix_task[0] : (k == 1 ? c.mix_task[1] : (k == 2 ? c.mix_task[2] : (k == 3 ? c.mix_task[3] : c.mix_task[4])));
}

// this env's view of the config in a REX_TASK_MIXED batch: its task and the per-task constants that go with it
__device__ __forceinline__ void mixed_config(const DevCfg& c, int gidx, DevCfg& cm) {
  cm = c;
  cm.task = mixed_task_of(c, gidx);
  cm.action_repeat = task_action_repeat(cm.task);
  cm.iterations = 300 / cm.action_repeat;                       // rex_gym_env.py:25,184
  const float b = task_action_bound(cm.task, c.signal);
  cm.act_lo = -b; cm.act_hi = b;
  cm.w_energy = task_energy_weight(cm.task);
}
// snapshot record of (terrain, task): one settled robot per terrain and -- in a mixed batch -- per task of the mix
// (the reset motion runs under the task's own numSolverIterations)
__device__ __forceinline__ int mix_slot(const DevCfg& c, int task) {
  int sl = 0;
#pragma unroll
  for (int k = 1; k < 5; ++k) if (k < c.n_mix && c.mix_task[k] == task) sl = k;
  return sl;
}

// INIT_POSES['rest_position'] (rex_constants.py:41-46): the foot target 6 rad lies beyond the URDF bound 2.59
__device__ __forceinline__ float pose_rest(int j) {
  const int k = j % 3;
  return k == 0 ? (((j / 3) & 1) ? 0.4f : -0.4f) : (k == 1 ? -1.5f : 6.0f);
}
__device__ __forceinline__ float init_pose(const DevCfg& c, int j) { return c.signal == REX_SIGNAL_OL ? pose_stand_ol(j) : pose_stand(j); }
// the pose the reset motion drives to: reset(initial_motor_angles=...), standup_env.py:108-110 vs walk_env.py:125-131
__device__ __forceinline__ float reset_pose(const DevCfg& c, int j) { return c.task == REX_TASK_STANDUP ? pose_rest(j) : init_pose(c, j); }

// ---- latency model: Rex._observation_history / _GetDelayedObservation (model/rex.py:122,717-763) ----
__device__ __forceinline__ float& hist_at(const DevCfg& c, int i, int slot, int w) {
  return c.hist[((size_t)slot * c.hist_words + w) * c.n + i];   // hist_words = 3 NM + 7: 43 (mark 'base') or 61 ('arm')
}
// which two ring slots to blend, and with which weight, for an observation `latency` seconds old
__device__ __forceinline__ void delay_slots(uint32_t hist, float latency, int n, float blend, int& s0, int& s1, float& alpha) {
  const int head = (int)(hist & 0xFFu), len = (int)((hist >> 8) & 0xFFu);
  int k0 = 0, k1 = 0;
  alpha = 0.0f;
  if (latency > 0.0f && len != 1) {
    if (n + 1 >= len) { k0 = k1 = len - 1; }
    else { k0 = n; k1 = n + 1; alpha = blend; }
  }
  s0 = (head - k0 + 2 * REX_HISTORY_LEN) % REX_HISTORY_LEN;
  s1 = (head - k1 + 2 * REX_HISTORY_LEN) % REX_HISTORY_LEN;
}
__device__ __forceinline__ float delayed_word(const DevCfg& c, int i, int s0, int s1, float alpha, int w) {
  return (1.0f - alpha) * hist_at(c, i, s0, w) + alpha * hist_at(c, i, s1, w);
}
// the controller-facing observation (Rex._control_observation): q, qd, tau_obs, quat, angular velocity
struct CtrlObs { float q[18], qd[18], tau[18], quat[4], w[3]; };

// terrain of (global env index, episode): the reference regenerates the field on every reset
// (rex_gym_env.py:347-348); here each episode picks one of the pool entries
__device__ __forceinline__ int terrain_index(const DevCfg& c, int gidx, int episode) {
  return (int)(((uint32_t)gidx + 977u * (uint32_t)episode) % (uint32_t)c.n_terrain);
}
__device__ __forceinline__ Ground env_ground(const DevCfg& c, int i, int gidx, int episode) {
  Ground g{nullptr, 0.0f, 1.0f, 1.0f, kMu, c.geo, c.anchor};
  if (c.body_params) {
    g.base_mass_scale = c.body_params[i]; g.leg_mass_scale = c.body_params[(size_t)c.n + i]; g.mu = c.body_params[2 * (size_t)c.n + i];
  }
  if (c.mass_hi > 0.0f || c.mu_hi > 0.0f) {
    // per-reset draws of the env_randomizer hook (rex_gym_env.py:345-346): a pure function of (seed, env, episode), so
    // nothing has to be stored -- every step of the episode recomputes the same three numbers
    uint32_t ctr[4] = {(uint32_t)episode, (uint32_t)gidx, 1u, 0u};
    philox4x32(ctr, c.seed_lo, c.seed_hi);
    if (c.mass_hi > 0.0f) {
      g.base_mass_scale = fmaf(c.mass_hi - c.mass_lo, u01(ctr[0]), c.mass_lo);
      g.leg_mass_scale = fmaf(c.mass_hi - c.mass_lo, u01(ctr[1]), c.mass_lo);
    }
    if (c.mu_hi > 0.0f) g.mu = fmaf(c.mu_hi - c.mu_lo, u01(ctr[2]), c.mu_lo);
  }
  if (c.n_terrain > 0) {
    const int t = terrain_index(c, gidx, episode);
    g.h = c.terrain + (size_t)t * c.hf_stride;
    g.mid = c.terrain_mid[t];
  }
  return g;
}

// Rex.ReceiveObservation (rex.py:726-733): the true observation goes to the front of the history ring
template <int NM>
__device__ __forceinline__ void receive_observation(const DevCfg& c, EnvState& e, int i, bool live, const float* tau_obs) {
  if (!c.hist) return;
  const int head = ((int)(e.hist & 0xFFu) + 1) % REX_HISTORY_LEN;
  const int len = min((int)((e.hist >> 8) & 0xFFu) + 1, REX_HISTORY_LEN);
  e.hist = (uint32_t)head | ((uint32_t)len << 8);
  if (live) {
#pragma unroll
    for (int j = 0; j < NM; ++j) { hist_at(c, i, head, j) = e.ph.q[j]; hist_at(c, i, head, NM + j) = e.ph.qd[j]; hist_at(c, i, head, 2 * NM + j) = tau_obs[j]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) hist_at(c, i, head, 3 * NM + k) = e.ph.quat[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) hist_at(c, i, head, 3 * NM + 4 + k) = e.ph.ang[k];
  }
}

// Rex.ApplyAction + stepSimulation + ReceiveObservation (rex.py:158-163, 568-641).
template <bool LANECAP, class SM, class ARMP>
__device__ __forceinline__ void rex_substep(const DevCfg& c, EnvState& e, int i, bool live, float* cmd, float* tau_obs,
                                            const SM& sm, const Ground& ground, ARMP& armp) {
  constexpr int NM = ARMP::NM;
  float tau[18];
  const float limit = 1.0f / c.dt;  // OVERHEAT_SHUTDOWN_TIME / time_step, rex.py:607
  float qo[NM], qdo[NM];                        // what the PD loop sees: _GetPDObservation, rex.py:755-759
#pragma unroll
  for (int j = 0; j < NM; ++j) { qo[j] = e.ph.q[j]; qdo[j] = e.ph.qd[j]; }
  if (c.hist) {
    int s0, s1;
    float alpha;
    delay_slots(e.hist, c.pd_latency, c.pd_slots, c.pd_alpha, s0, s1, alpha);
#pragma unroll
    for (int j = 0; j < NM; ++j) { qo[j] = delayed_word(c, i, s0, s1, alpha, j); qdo[j] = delayed_word(c, i, s0, s1, alpha, NM + j); }
  }
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    float act, obs;
    motor_torque(cmd[j], qo[j], qdo[j], e.ph.qd[j], c.kp, c.kd, act, obs);
    uint32_t cnt = e.overheat[j];
    cnt = fabsf(act) > 2.45f ? min(cnt + 1u, 65535u) : 0u;                      // rex.py:603-606
    if ((float)cnt > limit) e.motor_en &= ~(1u << j);                           // rex.py:607-608
    e.overheat[j] = cnt;
    tau_obs[j] = obs;
    tau[j] = ((e.motor_en >> j) & 1u) ? act : 0.0f;                             // rex.py:617-623
  }
  // mark 'arm', <= 8 envs per wave: the motor-side state of the env waits in LDS while the substep runs (rex_device.h,
  // REX_MOTOR_PARK_WORDS); one lane of the group writes, all read back
  constexpr bool kPark = NM == 18 && SM::kEpw <= 8;
  if constexpr (kPark) {
    if ((threadIdx.x & 7u) == 0u) {
#pragma unroll
      for (int j = 0; j < 18; ++j) { sm.motorf(j) = cmd[j]; sm.motorf(18 + j) = tau_obs[j]; sm.motorf(36 + j) = __uint_as_float(e.overheat[j]); }
      sm.motorf(54) = e.gait.phi; sm.motorf(55) = e.gait.last_time; sm.motorf(56) = e.gait.alpha;
      sm.motorf(57) = e.target; sm.motorf(58) = e.end_time; sm.motorf(59) = e.aux;
      sm.motorf(60) = __uint_as_float(e.flags); sm.motorf(61) = __int_as_float(e.steps); sm.motorf(62) = __int_as_float(e.episode);
      sm.motorf(63) = __uint_as_float(e.motor_en); sm.motorf(64) = __uint_as_float(e.hist);
    }
    asm volatile("" ::: "memory");
  }
  physics_substep<LANECAP>(e.ph, tau, c.dt, c.max_iterations, c.iterations, c.res_thr, sm, ground, armp, e.sweeps);
  if constexpr (kPark) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 18; ++j) { cmd[j] = sm.motorf(j); tau_obs[j] = sm.motorf(18 + j); e.overheat[j] = __float_as_uint(sm.motorf(36 + j)); }
    e.gait.phi = sm.motorf(54); e.gait.last_time = sm.motorf(55); e.gait.alpha = sm.motorf(56);
    e.target = sm.motorf(57); e.end_time = sm.motorf(58); e.aux = sm.motorf(59);
    e.flags = __float_as_uint(sm.motorf(60)); e.steps = __float_as_int(sm.motorf(61)); e.episode = __float_as_int(sm.motorf(62));
    e.motor_en = __float_as_uint(sm.motorf(63)); e.hist = __float_as_uint(sm.motorf(64));
  }
  receive_observation<NM>(c, e, i, live, tau_obs);
}

// Rex._control_observation as the env-level getters see it (delayed by control_latency when the model is on)
template <int NM>
__device__ __forceinline__ void control_observation(const DevCfg& c, const EnvState& e, int i, const float* tau_obs, CtrlObs& o) {
  if (c.hist) {
    int s0, s1; float alpha;
    delay_slots(e.hist, c.control_latency, c.control_slots, c.control_alpha, s0, s1, alpha);
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      o.q[j] = delayed_word(c, i, s0, s1, alpha, j); o.qd[j] = delayed_word(c, i, s0, s1, alpha, NM + j);
      o.tau[j] = delayed_word(c, i, s0, s1, alpha, 2 * NM + j);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.quat[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) o.w[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + 4 + k);
  } else {
#pragma unroll
    for (int j = 0; j < NM; ++j) { o.q[j] = e.ph.q[j]; o.qd[j] = e.ph.qd[j]; o.tau[j] = tau_obs[j]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.quat[k] = e.ph.quat[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o.w[k] = e.ph.ang[k];
  }
}

// RangeNormalize of the observation (wrappers.py:236-240); bounds are symmetric (rex_gym_env.py:277-278)
__device__ __forceinline__ void normalize_obs(const DevCfg& c, float* obs) {
#pragma unroll
  for (int k = 0; k < 22; ++k) {
    if (k < c.obs_dim) {
      const float hi = (k == 2 || k == 3) ? c.obs_hi_rate : c.obs_hi_ang, lo = -hi;
      obs[k] = 2.0f * (obs[k] - lo) / (hi - lo) - 1.0f;
    }
  }
}

template <int NM>
__device__ __forceinline__ void env_observation(const DevCfg& c, const CtrlObs& co, float* obs, int gidx = 0, int episode = 0, int step = 0) {
  float rpy[3];
  quat_to_euler(co.quat, rpy);
  float wx = co.w[0], wy = co.w[1];
  if (c.noise_on) {                                                               // GetBaseRollPitchYaw / ...Rate: rex.py:430-442,548-558
    float z[4];
    if (c.noise[3] > 0.0f) { gauss4(c.seed_lo, c.seed_hi, gidx, episode, step, kNzObsRpy, z); rpy[0] += c.noise[3] * z[0]; rpy[1] += c.noise[3] * z[1]; }
    if (c.noise[4] > 0.0f) { gauss4(c.seed_lo, c.seed_hi, gidx, episode, step, kNzObsRate, z); wx += c.noise[4] * z[0]; wy += c.noise[4] * z[1]; }
  }
  obs[0] = rpy[0]; obs[1] = rpy[1]; obs[2] = wx; obs[3] = wy;                       // walk_env.py:356-362
  if (c.task == REX_TASK_GALLOP) {
    float nz[20];
    const bool noisy = c.noise_on && c.noise[0] > 0.0f;                           // GetMotorAngles: noise, then MapToMinusPiToPi (rex.py:457-468)
    if (noisy) {
#pragma unroll
      for (int b = 0; b < (NM + 3) / 4; ++b) gauss4(c.seed_lo, c.seed_hi, gidx, episode, step, kNzAngle + b, nz + 4 * b);
    }
#pragma unroll
    for (int j = 0; j < NM; ++j) {                                                // MapToMinusPiToPi, rex.py:26-41
      float a = fmodf(co.q[j] + (noisy ? c.noise[0] * nz[j] : 0.0f), 2.0f * kPi);
      if (a >= kPi) a -= 2.0f * kPi; else if (a < -kPi) a += 2.0f * kPi;
      obs[4 + j] = a;
    }
  }
}

// RexWalkEnv.reset / RexReactiveEnv.reset ... draws on top of the settled snapshot.  `seen` receives what the robot last
// observed of its base (quaternion, angular velocity): the settled one -- the turn env teleports the base behind the
// observation's back (turn_env.py:158-160), and reset() returns that older reading.
template <int NM>
__device__ __forceinline__ void env_reset(const DevCfg& c, const float* snap, int i, bool live, int gidx, EnvState& e, float* seen) {
  const int32_t episode = e.episode + 1;
  const float alpha = e.gait.alpha;   // the env keeps one GaitPlanner for life: its arc angle survives reset() (gait_planner.py:76-85)
  const int nrec = (c.n_terrain > 0 ? c.n_terrain : 1) * c.n_mix;
  const int rec = (c.n_terrain > 0 ? terrain_index(c, gidx, episode) : 0) * c.n_mix + (c.n_mix > 1 ? mix_slot(c, c.task) : 0);
  load_env<NM>(snap, nrec, rec, e);   // settled on this episode's terrain (under this env's task)
  if (c.hist) {   // the deque as the reset motion left it: its last 100 observations, ring position included (e.hist)
    if (live) {
      const float* ring = snap + (size_t)Lay<NM>::WORDS * nrec;
#pragma unroll 8
      for (int k = 0; k < REX_HISTORY_LEN * (3 * NM + 7); ++k) c.hist[(size_t)k * c.n + i] = ring[(size_t)k * nrec + rec];
    }
    mirror_sync();
  } else e.hist = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) seen[k] = e.ph.quat[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) seen[4 + k] = e.ph.ang[k];
  e.episode = episode;
  e.gait.phi = 0.0f; e.gait.last_time = 0.0f; e.gait.alpha = alpha;
  // key = the 64-bit seed, counter = (episode, global env index): distinct seeds give independent streams for every env
  // (a key of seed ^ index would hand (seed 0, env 1) and (seed 1, env 0) the same stream)
  uint32_t ctr[4] = {(uint32_t)episode, (uint32_t)gidx, 0u, 0u};
  philox4x32(ctr, c.seed_lo, c.seed_hi);
  e.flags = 0;
  const float u = u01(ctr[1]);
  if (c.task == REX_TASK_WALK) {
    const int backwards = c.backwards < 0 ? (int)(ctr[0] >> 31) : c.backwards;   // walk_env.py:133-136
    if (backwards) e.flags |= REX_F_BACKWARDS;
    if (c.target_position != 0.0f) e.target = c.target_position;
    else e.target = backwards ? (-2.0f - u) : (1.0f + 2.0f * u);                 // walk_env.py:143-147
  } else {
    e.target = c.target_position != 0.0f ? c.target_position : (1.0f + 2.0f * u); // gallop_env.py:150-152
  }
  e.end_time = 0.0f; e.aux = 0.0f; e.steps = 0;
  if (c.task == REX_TASK_POSES) {                                                // poses_env.py:153-192
    const int k = c.pose_index >= 0 ? c.pose_index : episode % 5;               // deque rotation: one pop per reset()
    // _ranges (rex_gym_env.py:258-265): base_y, base_z, roll, pitch, yaw
    const float lo = k == 0 ? -0.007f : (k == 1 ? -0.048f : -0.78539816339744830962f);
    const float hi = k == 0 ? 0.007f : (k == 1 ? 0.021f : 0.78539816339744830962f);
    e.aux = (float)k;
    e.target = c.pose_index >= 0 ? c.pose_value : fmaf(hi - lo, u, lo);
  }
  if (c.task == REX_TASK_TURN) {                                                 // turn_env.py:129-160
    const float tgt = (c.orient_fixed & 1) ? c.target_orient : fmaf(5.8f, u, 0.2f);
    const float ini = (c.orient_fixed & 2) ? c.init_orient : fmaf(5.8f, u01(ctr[2]), 0.2f);
    e.target = tgt; e.aux = ini;
    float sh, ch;
    sincos_fast(0.5f * ini, sh, ch);                                             // getQuaternionFromEuler([0, 0, yaw])
    const float nn = rsqrtf(sh * sh + ch * ch);
    e.ph.quat[0] = 0.0f; e.ph.quat[1] = 0.0f; e.ph.quat[2] = sh * nn; e.ph.quat[3] = ch * nn;
    e.ph.pos[0] = 0.0f; e.ph.pos[1] = 0.0f; e.ph.pos[2] = c.init_z;              // resetBasePositionAndOrientation
  }
}

__device__ __forceinline__ void order_signal(const float* ang, float* cmd) {  // FR,FL,RR,RL -> FL,FR,RL,RR
#pragma unroll
  for (int k = 0; k < 3; ++k) { cmd[k] = ang[3 + k]; cmd[3 + k] = ang[k]; cmd[6 + k] = ang[9 + k]; cmd[9 + k] = ang[6 + k]; }
}

// RexWalkEnv._transform_action_to_motor_command (walk_env.py:207-324)
__device__ __forceinline__ void walk_command(const DevCfg& c, EnvState& e, const float* action, float* cmd) {
  if (e.flags & REX_F_STAY_STILL) {
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = init_pose(c, j);
    return;
  }
  const float t = (float)(e.steps * c.action_repeat) * c.dt;                     // rex.py:155-156
  if (e.target != 0.0f && fabsf(e.ph.pos[0]) >= fabsf(e.target) - 0.15f) {       // walk_env.py:207-215
    e.flags |= REX_F_GOAL_REACHED;
    if (!(e.flags & REX_F_TERMINATING)) { e.end_time = t; e.flags |= REX_F_TERMINATING; }
  }
  const bool backwards = (e.flags & REX_F_BACKWARDS) != 0;
  if (c.signal == REX_SIGNAL_IK) {                                               // walk_env.py:252-290
    const float p = 0.8f + action[0];
    const float gait_coeff = (0.0f <= t && t <= p) ? t : 1.0f;
    const float period = backwards ? 0.5f : 0.65f;
    const float pos[3] = {backwards ? 0.0f : 0.01f, 0.0f, 0.0f}, orn[3] = {0.0f, 0.0f, 0.0f};
    float step_length = (backwards ? -0.3f : 0.6f) * gait_coeff;
    if (e.flags & REX_F_GOAL_REACHED) {
      const float pb = 0.8f + action[1];
      const float b = (e.end_time <= t && t <= pb + e.end_time) ? 1.0f - (t - e.end_time) : 0.0f;
      step_length *= b;
      if (b == 0.0f) e.flags |= REX_F_STAY_STILL;
    }
    const float direction = step_length < 0.0f ? -1.0f : 1.0f;
    float frames[12], ang[12];
    gait_loop(e.gait, 0, step_length, 0.0f, 0.0f, period, direction, t * c.gait_clock, frames);
    ik_solve(orn, pos, frames, ang);
    order_signal(ang, cmd);
  } else {                                                                       // walk_env.py:292-315
    float l_a = 0.1f, f_a = 0.2f;
    if (e.flags & REX_F_GOAL_REACHED) {
      const bool inside = e.end_time <= t && t <= 0.8f + e.end_time;
      const float b = inside ? 1.0f - (t - e.end_time) : 0.0f;
      l_a *= b; f_a *= b;
      // `if coeff is 0.0` (walk_env.py:300) is an identity test: true exactly when the brake function returns its
      // end_value argument, i.e. outside the brake window (pinned by tests/golden/env_command_golden.json)
      if (!inside) e.flags |= REX_F_STAY_STILL;
    }
    const float sc = (0.0f <= t && t <= 0.8f) ? t : 1.0f;
    l_a *= sc; f_a *= sc;
    float sph, cph;
    sincos_fast(2.0f * kPi / 0.125f * t, sph, cph);
    const float le = l_a * cph, fe = f_a * cph;
    const float pose[12] = {0.f, le + action[0], fe + action[1], 0.f, -le + action[2], -fe + action[3],
                            0.f, -le + action[4], -fe + action[5], 0.f, le + action[6], fe + action[7]};
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = pose_stand_ol(j) + pose[j];
  }
}

// RexReactiveEnv._transform_action_to_motor_command (gallop_env.py:212-313)
__device__ __forceinline__ void gallop_command(const DevCfg& c, EnvState& e, const float* action, float* cmd) {
  if (e.flags & REX_F_STAY_STILL) {
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = pose_stand(j);                         // rex.initial_pose
    return;
  }
  const float t = (float)(e.steps * c.action_repeat) * c.dt;
  if (e.target != 0.0f && fabsf(e.ph.pos[0]) >= fabsf(e.target)) {               // gallop_env.py:212-220
    e.flags |= REX_F_GOAL_REACHED;
    if (!(e.flags & REX_F_TERMINATING)) { e.end_time = t; e.flags |= REX_F_TERMINATING; }
  }
  if (c.signal == REX_SIGNAL_IK) {                                               // gallop_env.py:257-285
    const float pg = 1.0f + action[1];
    const float gait_coeff = (0.0f <= t && t <= pg) ? t : 1.0f;
    const float pos[3] = {0.01f, 0.0f, -0.007f}, orn[3] = {0.0f, 0.0f, 0.0f};
    float step_length = 1.3f * gait_coeff;
    if (e.flags & REX_F_GOAL_REACHED) {
      const float pb = 1.0f + action[0];
      step_length *= (e.end_time <= t && t <= pb + e.end_time) ? 1.0f - (t - e.end_time) : 0.0f;
    }
    float frames[12], ang[12];
    gait_loop(e.gait, 1, step_length, 0.0f, 0.0f, 0.3f, 1.0f, t * c.gait_clock, frames);
    ik_solve(orn, pos, frames, ang);
    order_signal(ang, cmd);
  } else {                                                                       // gallop_env.py:287-304
    float lp[4] = {action[0], action[1], action[2], action[3]};
    if (e.flags & REX_F_GOAL_REACHED) {
      const bool inside = e.end_time <= t && t <= 1.0f + e.end_time;
      const float b = inside ? 1.0f - (t - e.end_time) : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) lp[k] *= b;
      if (!inside) e.flags |= REX_F_STAY_STILL;   // gallop_env.py:291: `coeff is 0.0`, an identity test (see walk_command)
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      cmd[3 * l] = init_pose(c, 3 * l);
      cmd[3 * l + 1] = init_pose(c, 3 * l + 1) + (l < 2 ? lp[0] : lp[2]);
      cmd[3 * l + 2] = init_pose(c, 3 * l + 2) + (l < 2 ? lp[1] : lp[3]);
    }
  }
}

// RexPosesEnv._signal (poses_env.py:186-225)
__device__ __forceinline__ void poses_command(const DevCfg& c, EnvState& e, const float* action, float* cmd) {
  const float t = (float)(e.steps * c.action_repeat) * c.dt;
  const float p = 0.8f + action[0];
  const float coeff = (0.0f <= t && t <= p) ? t : 1.0f;
  const float staged = e.target * coeff;
  const int k = (int)e.aux;
  const float pos[3] = {0.01f, k == 0 ? staged : 0.0f, k == 1 ? staged : 0.0f};
  const float orn[3] = {k == 2 ? staged : 0.0f, k == 3 ? staged : 0.0f, k == 4 ? staged : 0.0f};
  const float frames[12] = {kIkL / 2, -kIkYDist / 2, -kIkHeight, kIkL / 2, kIkYDist / 2, -kIkHeight,
                            -kIkL / 2, -kIkYDist / 2, -kIkHeight, -kIkL / 2, kIkYDist / 2, -kIkHeight};
  float ang[12];
  ik_solve(orn, pos, frames, ang);
  order_signal(ang, cmd);
}

// RexTurnEnv._transform_action_to_motor_command (turn_env.py:239-347)
__device__ __forceinline__ void turn_command(const DevCfg& c, EnvState& e, const float* ctrl_quat, const float* action, float* cmd, int gidx) {
  const float t = (float)(e.steps * c.action_repeat) * c.dt;
  if (e.flags & REX_F_STAY_STILL) {
    if (t - e.end_time >= 1.0f) e.flags |= REX_F_ENV_GOAL;                       // _terminate_with_delay
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = init_pose(c, j);
    return;
  }
  {                                                                              // _check_target_position
    float rpy[3];
    quat_to_euler(ctrl_quat, rpy);                                               // GetBaseOrientation (delayed when latency is on)
    if (c.noise_on && c.noise[3] > 0.0f) {                                       // ... through GetBaseRollPitchYaw's sensor noise (rex.py:430-442)
      float z[4];
      gauss4(c.seed_lo, c.seed_hi, gidx, e.episode, e.steps, kNzGoal, z);
      rpy[0] += c.noise[3] * z[0]; rpy[1] += c.noise[3] * z[1]; rpy[2] += c.noise[3] * z[2];
      float q[4], r2[3];                                                         // rpy -> quaternion -> rpy, as the reference does
      {
        float sr, cr, sp, cp, sy, cy;
        sincos_fast(rpy[0] * 0.5f, sr, cr); sincos_fast(rpy[1] * 0.5f, sp, cp); sincos_fast(rpy[2] * 0.5f, sy, cy);
        q[0] = sr * cp * cy - cr * sp * sy; q[1] = cr * sp * cy + sr * cp * sy; q[2] = cr * cp * sy - sr * sp * cy; q[3] = cr * cp * cy + sr * sp * sy;
      }
      quat_to_euler(q, r2);
      rpy[2] = r2[2];
    }
    float cz = rpy[2];
    if (cz < 0.0f) cz += 6.28f;
    if (fabsf(e.target - cz) <= 0.01f) {
      e.flags |= REX_F_GOAL_REACHED;
      if (!(e.flags & REX_F_TERMINATING)) { e.end_time = t; e.flags |= REX_F_TERMINATING; }
    }
  }
  const float diff = fabsf(e.aux - e.target);                                    // _solve_direction
  const bool clockwise = e.aux < e.target ? diff > 3.14f : diff < 3.14f;
  if (e.flags & REX_F_GOAL_REACHED) e.flags |= REX_F_STAY_STILL;
  if (c.signal == REX_SIGNAL_IK) {
    const float coeff = (0.0f <= t && t <= 0.8f) ? t : 1.0f;
    float dirv = -0.5f * coeff;
    if (clockwise) dirv = -dirv;
    const float pos[3] = {0.009f, 0.0f, 0.0f}, orn[3] = {0.0f, 0.0f, 0.0f};
    float frames[12], ang[12];
    gait_loop(e.gait, 0, 0.02f, 0.0f, dirv + action[0], 0.75f + action[1], 1.0f, t * c.gait_clock, frames);
    ik_solve(orn, pos, frames, ang);
    order_signal(ang, cmd);
  } else {
    const float ext = 0.1f, swing = 0.03f + action[0], swipe = 0.05f + action[1];
    const int ith = ((int)(t / 0.1f)) % 2;
    const float ms = clockwise ? swing : -swing;     // right_* = left_* with the swing sign flipped
    const float first[12] = {swipe, ext, ms, -swipe, ext, -ms, swipe, -ext, -ms, -swipe, -ext, ms};
    const float second[12] = {-swipe, 0.f, -ms, swipe, 0.f, ms, -swipe, 0.f, ms, swipe, 0.f, -ms};
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = pose_stand_ol(j) + (ith ? second[j] : first[j]);
  }
}

// RexStandupEnv._signal (standup_env.py:113-120): the 'stand' pose, scaled by a 'brake' overshoot for the first 0.1 s
__device__ __forceinline__ void standup_command(const DevCfg& c, const EnvState& e, const float* action, float* cmd) {
  const float t = (float)(e.steps * c.action_repeat) * c.dt;               // GetTimeSinceReset, rex.py:155-156
  const float f = t > 0.1f ? 1.0f : (0.1f + action[0]) / (t + 1.0f) + 1.5f;
  const float leg = -0.88643435f * f, foot = 1.30197369f * f;
#pragma unroll
  for (int l = 0; l < 4; ++l) { cmd[3 * l] = 0.0f; cmd[3 * l + 1] = leg; cmd[3 * l + 2] = foot; }
}

// ------------------------------------------------------------------------------------------
#ifndef REX_STEP_KERNEL_ATTR
#define REX_STEP_KERNEL_ATTR          /* developer experiments: e.g. -DREX_STEP_KERNEL_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))' */
#endif
#ifndef REX_FAST_EPW
#define REX_FAST_EPW 4
#endif
template <int EPW, bool ARM, bool MIXED, bool BODY>
__global__ __launch_bounds__(REX_WAVE) REX_STEP_KERNEL_ATTR void rex_step_kernel(DevCfg c, float* __restrict__ state, const float* __restrict__ snap,
                                                            const float* __restrict__ action, float* __restrict__ obs_out,
                                                            float* __restrict__ reward_out, uint8_t* __restrict__ done_out,
                                                            float* __restrict__ cmd_out) {
  // EPW envs share this wave (host picks it, rex_step): a small batch is spread over MORE, emptier waves because
  // idle SIMDs are free and a wave leaves the PGS sweep loop only when its slowest env has converged (and skips
  // only the legs no env of the wave has in contact), so fewer envs per wave means fewer sweeps and rows per
  // wave.  With EPW <= 16 every env owns a group of LPE = 8 (EPW <= 8) or 4 (EPW = 16) adjacent lanes (for EPW = 4 the
  // upper 32 lanes repeat the lower 32): the lanes of a group run the same arithmetic on the same state, split the
  // per-leg and per-row work of a substep between them (rex_device.h) and only lane 0 of the group stores.
  // MIXED (REX_TASK_MIXED): the envs of a wave may run different tasks -- c_ below is the lane's own view of the config.
  constexpr int NM = ARM ? 18 : 12;   // mark='arm': 6 more motors held at ARM_POSES['rest'] (rex_gym_env.py:347-353)
  constexpr int kLegF4 = REX_LEG_F4_OF(EPW, ARM);
  constexpr int kRowsF4 = ARM ? REX_LDS_F4_PER_ENV_ARM_OF(EPW) : REX_ROWS_F4_OF(kLegF4);
  static_assert(!BODY || EPW <= 16, "link-box contact rows: lane-group kernels only");
  constexpr int kMotorF4 = (ARM && EPW <= 8) ? REX_MOTOR_PARK_F4 : 0;
  __shared__ float4 lds[(kRowsF4 + (EPW <= 16 ? REX_PARK_F4 : 0) + (BODY ? REX_BODY_F4 : 0) + kMotorF4) * EPW];
  REX_STAMP(t_kernel);
  if (c.clock && threadIdx.x == 0) atomicMin(&c.clock[0], (unsigned long long)wall_clock64());
  const int lane = threadIdx.x;
  constexpr int LPE = EPW < 64 ? lanes_per_env(EPW) : 1;     // EPW <= 16: lane = LPE * slot + p (rex_device.h, group layout)
  const int slot = (lane / LPE) & (EPW - 1);
  // Block b runs on XCD b % 8 (observed placement; speed only).  A 64-byte sector of a state word holds 16 envs = 16 / EPW
  // blocks' worth: hand the blocks of one sector to the same XCD, so that one L2 fetches (and writes back) the sector
  // instead of 16 / EPW of them.  A bijection on the full groups of 8 x (16 / EPW) blocks; the tail keeps its order.
  int blk = (int)blockIdx.x;
  if constexpr (EPW < 16) {
    constexpr int G = 16 / EPW;
    const int full = ((int)gridDim.x / (8 * G)) * (8 * G);
    if (blk < full) { const int xcd = blk & 7, q = blk >> 3; blk = ((q / G) * 8 + xcd) * G + (q % G); }
  }
  const int gi = blk * EPW + slot;
  const bool live = lane < LPE * EPW && (lane & (LPE - 1)) == 0 && gi < c.n;
  const int gj = gi < c.n ? gi : c.n - 1;   // tail slots shadow the last env (keeps the wave convergent)
  const int i = c.perm ? c.perm[gj] : gj;   // regrouped batches: the env this slot works on
  const Lds<EPW, kLegF4, BODY> sm{lds, slot, EPW <= 16 ? lds + kRowsF4 * EPW : nullptr, BODY ? lds + (kRowsF4 + REX_PARK_F4) * EPW : nullptr,
                                  kMotorF4 ? lds + (kRowsF4 + REX_PARK_F4 + (BODY ? REX_BODY_F4 : 0)) * EPW : nullptr};
  typename ArmHook<EPW, ARM>::type armp = ArmHook<EPW, ARM>::make(lds, slot);

  DevCfg cmix;                          // MIXED only
  if constexpr (MIXED) mixed_config(c, c.env_index_base + i, cmix);
  const DevCfg& c_ = MIXED ? cmix : c;

  EnvState e;
  load_env<NM>(state, c.n, i, e);
  e.sweeps = 0;
  float act[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float a = k < c.action_dim ? action[(size_t)i * c.action_dim + k] : 0.0f;
    if (c.range_normalize) {                       // ClipAction + RangeNormalize (wrappers.py:229-234,261-265)
      a = fminf(fmaxf(a, -1.0f), 1.0f);
      a = (a + 1.0f) / 2.0f * (c_.act_hi - c_.act_lo) + c_.act_lo;
    }
    act[k] = a;
  }

  float cmd[NM];
  if (ARM) {
#pragma unroll
    for (int j = 12; j < NM; ++j) cmd[j] = (float)REXA_REST[j - 12];
  }
  if (c_.task == REX_TASK_GALLOP) gallop_command(c_, e, act, cmd);
  else if (c_.task == REX_TASK_TURN) {
    float cq[4] = {e.ph.quat[0], e.ph.quat[1], e.ph.quat[2], e.ph.quat[3]};
    if (c.hist) {
      int s0, s1; float alpha;
      delay_slots(e.hist, c.control_latency, c.control_slots, c.control_alpha, s0, s1, alpha);
#pragma unroll
      for (int k = 0; k < 4; ++k) cq[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + k);
    }
    turn_command(c_, e, cq, act, cmd, c.env_index_base + i);
  }
  else if (c_.task == REX_TASK_POSES) poses_command(c_, e, act, cmd);
  else if (c_.task == REX_TASK_STANDUP) standup_command(c_, e, act, cmd);
  else walk_command(c_, e, act, cmd);

  float tau_obs[NM];
  const Ground ground = env_ground(c, i, c.env_index_base + i, e.episode);
  const int step0 = e.steps, episode0 = e.episode;   // keys of this step's sensor-noise draws

  // everything of env.step() after Rex.Step: reward, termination, in-launch reset, observation, stores
  auto epilogue = [&](bool commit) {
  // ---- reward (rex_gym_env.py:501-542) ----
  CtrlObs co;
  control_observation<NM>(c, e, i, tau_obs, co);
  float rpy[3], r20, r21, r22;
  quat_to_euler(co.quat, rpy);       // GetBaseOrientation: (delayed) quat -> RPY -> quat, rex.py:530-537
  if (c.noise_on) {                  // sensor noise: the reward's and is_fallen's orientation reads draw separately
    const int gx = c.env_index_base + i;
    float z[4], rp[3], d0, d1;
    if (c.noise[3] > 0.0f) {
      gauss4(c.seed_lo, c.seed_hi, gx, episode0, step0, kNzFallenRpy, z);
      rp[0] = rpy[0] + c.noise[3] * z[0]; rp[1] = rpy[1] + c.noise[3] * z[1]; rp[2] = rpy[2] + c.noise[3] * z[2];
      euler_to_row2(rp, d0, d1, r22);
      gauss4(c.seed_lo, c.seed_hi, gx, episode0, step0, kNzRewardRpy, z);
      rp[0] = rpy[0] + c.noise[3] * z[0]; rp[1] = rpy[1] + c.noise[3] * z[1]; rp[2] = rpy[2] + c.noise[3] * z[2];
      euler_to_row2(rp, r20, r21, d0);
    } else euler_to_row2(rpy, r20, r21, r22);
    float nt[20], nv[20];
#pragma unroll
    for (int b = 0; b < (NM + 3) / 4; ++b) {
      gauss4(c.seed_lo, c.seed_hi, gx, episode0, step0, kNzTorque + b, nt + 4 * b);
      gauss4(c.seed_lo, c.seed_hi, gx, episode0, step0, kNzVelocity + b, nv + 4 * b);
    }
#pragma unroll
    for (int j = 0; j < NM; ++j) { co.tau[j] += c.noise[2] * nt[j]; co.qd[j] += c.noise[1] * nv[j]; }   // GetMotorTorques / Velocities
  } else
  euler_to_row2(rpy, r20, r21, r22);
  float x = -e.ph.pos[0];
  if (c.backwards > 0) x = -x;      // `if self._backwards:` is the constructor argument, not the draw
  e.target = fabsf(e.target);       // rex_gym_env.py:510
  const float T = e.target;
  float fwd;
  if (x > T + 0.15f) fwd = T - x;
  else if (T <= x && x <= T + 0.15f) fwd = 1.0f;
  else if (x <= 0.05f) fwd = 0.0f;
  else fwd = x / T;
  const float drift = -fabsf(e.ph.pos[1]);
  const float shake = -fabsf(r20 + r21);
  float dp = 0.0f;
#pragma unroll
  for (int j = 0; j < NM; ++j) dp += co.tau[j] * co.qd[j];   // GetMotorTorques . GetMotorVelocities
  const float energy = -fabsf(dp) * c.dt;
  float reward = c.w_dist * fwd + c_.w_energy * energy + c.w_drift * drift + c.w_shake * shake;
  if (c_.task == REX_TASK_TURN) reward = 0.035f - fabsf(e.ph.pos[0]) - fabsf(e.ph.pos[1]);   // turn_env.py:362-367
  if (c_.task == REX_TASK_POSES) reward = 1.0f;                                                // poses_env.py:267-269
  if (c_.task == REX_TASK_STANDUP) {                                                           // standup_env.py:150-166
    float pr = fabsf(e.ph.pos[0]) + fabsf(e.ph.pos[1]) + fabsf(0.21f - e.ph.pos[2]);
    pr = pr < 0.1f ? 1.0f - pr : -pr;
    if (e.ph.pos[2] > 0.21f) pr = -1.0f - pr;
    reward = pr;
  }

  // ---- termination (rex_gym_env.py:490-499, walk_env.py:326-338, gallop_env.py:315-329) ----
  bool done;
  if (c_.task == REX_TASK_GALLOP || c_.task == REX_TASK_STANDUP) {   // GetTrueBaseRollPitchYaw: never delayed (gallop_env.py:319-329)
    float trpy[3];
    quat_to_euler(e.ph.quat, trpy);
    done = fabsf(trpy[0]) > 0.3f || fabsf(trpy[1]) > 0.5f || (c_.task == REX_TASK_GALLOP && e.ph.pos[1] > 0.3f);
  } else done = r22 < 0.85f;
  if ((e.flags & REX_F_ENV_GOAL) && c_.task != REX_TASK_STANDUP) done = true;     // rex_gym_env.py:495; standup overrides _termination
  if (c_.task == REX_TASK_POSES) done = false;                                    // is_fallen() returns False, poses_env.py:265
  e.steps += 1;
  if (c.max_steps > 0 && e.steps >= c.max_steps) done = true;
  if (done) e.flags |= REX_F_DONE;
  if (done && c.auto_reset) {
    float seen[7];
    env_reset<NM>(c_, snap, i, commit, c.env_index_base + i, e, seen);
#pragma unroll
    for (int j = 0; j < NM; ++j) tau_obs[j] = 0.0f;
    control_observation<NM>(c, e, i, tau_obs, co);
    if (!c.hist) {
#pragma unroll
      for (int k = 0; k < 4; ++k) co.quat[k] = seen[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) co.w[k] = seen[4 + k];
    }
  }

  float obs[22];
  if constexpr (MIXED) {
#pragma unroll
    for (int k = 4; k < 22; ++k) obs[k] = 0.0f;    // a task with a narrower observation leaves the tail of its row 0
  }
  env_observation<NM>(c_, co, obs, c.env_index_base + i, episode0, step0);
  if (c.range_normalize) normalize_obs(c, obs);
  if (commit) {
    // an opaque copy of the env index: the store addresses are rebuilt here instead of 54 address pairs being carried
    // (in AGPRs and scratch) from load_env across the whole kernel
    int is = i;
    asm volatile("" : "+v"(is));
    store_env<NM>(state, c.n, is, e);
    for (int k = 0; k < c.obs_dim; ++k) obs_out[(size_t)is * c.obs_dim + k] = obs[k];
    reward_out[is] = reward;
    done_out[is] = done ? 1 : 0;
    if (c.sweeps) c.sweeps[is] = e.sweeps;
    if (cmd_out) {
#pragma unroll
      for (int j = 0; j < NM; ++j) cmd_out[(size_t)is * NM + j] = cmd[j];
    }
  }
  };

  if constexpr (!MIXED) {
    for (int k = 0; k < c.action_repeat; ++k) rex_substep<false>(c, e, i, live, cmd, tau_obs, sm, ground, armp);   // Rex.Step
    epilogue(live);
  } else {
    // Rex.Step of tasks with different action_repeat in one wave: every lane runs max_repeat substeps (the substep is
    // full of wave-level operations), an env whose own count is reached finishes its env.step() -- epilogue, stores --
    // before the extra substeps, whose results it never stores
    for (int k = 0; k <= c.max_repeat; ++k) {
      if (k == c_.action_repeat) epilogue(live);
      if (k < c.max_repeat) rex_substep<true>(c_, e, i, live && k < c_.action_repeat, cmd, tau_obs, sm, ground, armp);
    }
  }
  if (c.clock && threadIdx.x == 0) atomicMax(&c.clock[1], (unsigned long long)wall_clock64());
#ifdef REX_PROF
  if (threadIdx.x == 0 && blockIdx.x < 1024) { g_prof[10 * blockIdx.x + 8] += clock64() - t_kernel; g_prof[10 * blockIdx.x + 9] += 1; }
#endif
}

// The reset motion of Rex.Reset (rex.py:296-324).  Plane: ONE robot, lane 0 writes the snapshot.  Terrain pool:
// lane t settles on terrain t and writes snapshot record t (word-major [53][n_terrain]).
template <bool ARM, bool BODY>
__global__ __launch_bounds__(REX_WAVE) void rex_settle_kernel(DevCfg c, float* __restrict__ snap) {
  constexpr int NM = ARM ? 18 : 12;
  constexpr int EPW = (ARM || BODY) ? 16 : REX_WAVE;   // the arm rows / link-box rows do not fit 64 envs per workgroup in LDS
  constexpr int kLegF4 = REX_LEG_F4_OF(EPW, ARM);
  constexpr int kRowsF4 = ARM ? REX_LDS_F4_PER_ENV_ARM_OF(EPW) : REX_ROWS_F4_OF(kLegF4);
  __shared__ float4 lds[(kRowsF4 + (EPW <= 16 ? REX_PARK_F4 : 0) + (BODY ? REX_BODY_F4 : 0)) * EPW];
  constexpr int LPE = EPW < 64 ? lanes_per_env(EPW) : 1;
  const int lane = (int)(threadIdx.x / LPE) & (EPW - 1);
  const Lds<EPW, kLegF4, BODY> sm{lds, lane, EPW <= 16 ? lds + kRowsF4 * EPW : nullptr, BODY ? lds + (kRowsF4 + REX_PARK_F4) * EPW : nullptr, nullptr};
  typename ArmHook<EPW, ARM>::type armp = ArmHook<EPW, ARM>::make(lds, lane);
  const int nrec = (c.n_terrain > 0 ? c.n_terrain : 1) * c.n_mix;
  const int first = (int)blockIdx.x * EPW + lane;                       // the (terrain, task) record this lane group settles
  const int rec = (threadIdx.x & (LPE - 1)) == 0 ? first : nrec;        // one lane of the group stores it
  const int t = first < nrec ? first : nrec - 1;
  const int terr = t / c.n_mix, slot = t % c.n_mix;
  Ground ground{nullptr, 0.0f, 1.0f, 1.0f, kMu, c.geo, c.anchor};
  if (c.n_terrain > 0) { ground.h = c.terrain + (size_t)terr * c.hf_stride; ground.mid = c.terrain_mid[terr]; }
  EnvState e;
  memset(&e, 0, sizeof(e));
  e.ph.pos[2] = c.init_z;
  e.ph.quat[3] = 1.0f;
#pragma unroll
  for (int j = 0; j < 12; ++j) e.ph.q[j] = pose_stand(j);       // ResetPose: INIT_POSES[pose_id = 'stand']
  if (ARM) {                                                     // ResetPose: arm motors at ARM_POSES['rest'] (rex.py:371-373)
#pragma unroll
    for (int j = 12; j < NM; ++j) e.ph.q[j] = (float)REXA_REST[j - 12];
  }
  e.motor_en = (1u << NM) - 1u;
  // the latency model runs through the reset motion as well (rex.py:309-323): the snapshot's own ring sits behind its
  // state words, [100][43][nrec]; Reset() clears the deque, observes the dropped robot once if a motion follows,
  // and observes the final state once more after it
  DevCfg cs = c;
  cs.hist = (c.pd_latency > 0.0f || c.control_latency > 0.0f) ? snap + (size_t)Lay<NM>::WORDS * nrec : nullptr;
  cs.n = nrec;
  if (c.n_mix > 1) {   // the reset motion of this record's task: its own sweep cap (rex_gym_env.py:184)
    const int task = slot == 0 ? c.mix_task[0] : (slot == 1 ? c.mix_task[1] : (slot == 2 ? c.mix_task[2] : (slot == 3 ? c.mix_task[3] : c.mix_task[4])));
    cs.iterations = 300 / task_action_repeat(task);
  }
  const bool keeps = rec < nrec;
  e.hist = (uint32_t)(REX_HISTORY_LEN - 1);
  float tau_obs[NM];
#pragma unroll
  for (int j = 0; j < NM; ++j) tau_obs[j] = 0.0f;
  if (c.task != REX_TASK_POSES) {   // RexPosesEnv: base reset() with initial_motor_angles=None skips the motion (rex.py:308)
    receive_observation<NM>(cs, e, t, keeps, tau_obs);
    float cmd[NM];
    if (ARM) {
#pragma unroll
      for (int j = 12; j < NM; ++j) cmd[j] = (float)REXA_REST[j - 12];
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = pose_stand(j);
    for (int k = 0; k < 100; ++k) rex_substep<false>(cs, e, t, keeps, cmd, tau_obs, sm, ground, armp);   // rex.py:315-318
#pragma unroll
    for (int j = 0; j < 12; ++j) cmd[j] = reset_pose(c, j);
    for (int k = 0; k < c.reset_substeps; ++k) rex_substep<false>(cs, e, t, keeps, cmd, tau_obs, sm, ground, armp);   // rex.py:319-322
  }
  receive_observation<NM>(cs, e, t, keeps, tau_obs);                                                           // rex.py:323
  if (!cs.hist) e.hist = 0u;
  if (keeps) store_env<NM>(snap, nrec, rec, e);
}

template <int NM>
__global__ void rex_reset_kernel(DevCfg c, float* __restrict__ state, const float* __restrict__ snap,
                                 const int32_t* __restrict__ indices, int count, float* __restrict__ obs_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= count) return;
  const int i = indices ? indices[r] : r;
  if (i < 0 || i >= c.n) return;
  EnvState e;
  e.episode = (int32_t)ldi(state, c.n, Lay<NM>::EPISODE, i);
  e.gait.alpha = state[(size_t)Lay<NM>::ALPHA * c.n + i];
  float seen[7];
  DevCfg cmix;
  if (c.task == REX_TASK_MIXED) mixed_config(c, c.env_index_base + i, cmix);
  const DevCfg& c_ = c.task == REX_TASK_MIXED ? cmix : c;
  env_reset<NM>(c_, snap, i, true, c.env_index_base + i, e, seen);
  store_env<NM>(state, c.n, i, e);
  float obs[22], tz[NM];
#pragma unroll
  for (int j = 0; j < NM; ++j) tz[j] = 0.0f;
  CtrlObs co;
  control_observation<NM>(c, e, i, tz, co);
  if (!c.hist) {
    for (int k = 0; k < 4; ++k) co.quat[k] = seen[k];
    for (int k = 0; k < 3; ++k) co.w[k] = seen[4 + k];
  }
  for (int k = 4; k < 22; ++k) obs[k] = 0.0f;
  env_observation<NM>(c_, co, obs, c.env_index_base + i, e.episode, -1);   // reset()'s own reading: its own noise draws
  if (c.range_normalize) normalize_obs(c, obs);
  if (obs_out) for (int k = 0; k < c.obs_dim; ++k) obs_out[(size_t)r * c.obs_dim + k] = obs[k];
}

// Regrouping of a large batch (one workgroup): counting sort of the env indices by the solver sweeps of the last step,
// most sweeps first (the long waves start first), 64 bins.  perm[k] = env of wave slot k.  The order inside a bin does
// not matter: an env's result does not depend on its wave-mates.
__global__ __launch_bounds__(1024) void rex_regroup_kernel(int n, int bin_width, const int32_t* __restrict__ sweeps, int32_t* __restrict__ perm) {
  __shared__ int hist[64], base[64];
  const int t = threadIdx.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int i = t; i < n; i += 1024) atomicAdd(&hist[63 - min(sweeps[i] / bin_width, 63)], 1);
  __syncthreads();
  if (t == 0) { int acc = 0; for (int b = 0; b < 64; ++b) { base[b] = acc; acc += hist[b]; } }
  __syncthreads();
  for (int i = t; i < n; i += 1024) perm[atomicAdd(&base[63 - min(sweeps[i] / bin_width, 63)], 1)] = i;
}
__global__ void rex_iota_kernel(int n, int32_t* __restrict__ perm, int32_t* __restrict__ sweeps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { perm[i] = i; sweeps[i] = 0; }
}

// ---- controller-only kernels ----
__global__ void rex_ik_kernel(int n, const float* __restrict__ orn, const float* __restrict__ pos,
                              const float* __restrict__ frames, float* __restrict__ angles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[3], p[3], f[12], a[12];
  for (int k = 0; k < 3; ++k) { o[k] = orn[3 * i + k]; p[k] = pos[3 * i + k]; }
  for (int k = 0; k < 12; ++k) f[k] = frames[12 * i + k];
  ik_solve(o, p, f, a);
  for (int k = 0; k < 12; ++k) angles[12 * i + k] = a[k];
}

__global__ void rex_motor_kernel(int n, const float* __restrict__ cmd, const float* __restrict__ q, const float* __restrict__ qd,
                                 const float* __restrict__ qdt, float kp, float kd, float* __restrict__ actual,
                                 float* __restrict__ observed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a, o;
  motor_torque(cmd[i], q[i], qd[i], qdt[i], kp, kd, a, o);
  actual[i] = a; observed[i] = o;
}

__global__ void rex_gait_kernel(int n, int mode, float* __restrict__ planner, const float* __restrict__ params,
                                float* __restrict__ frames) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  GaitState g{planner[3 * i], planner[3 * i + 1], planner[3 * i + 2]};
  const float* p = params + 6 * i;
  float f[12];
  gait_loop(g, mode, p[0], p[1], p[2], p[3], p[4], p[5], f);
  planner[3 * i] = g.phi; planner[3 * i + 1] = g.last_time; planner[3 * i + 2] = g.alpha;
  for (int k = 0; k < 12; ++k) frames[12 * i + k] = f[k];
}

}  // namespace rex

// =================================================================================================
//                                          host side: C ABI
// =================================================================================================
#define REX_TIMING_RING 256
struct RexSim {
  RexConfig cfg;
  rex::DevCfg dev;
  int epw;          // envs per wave of rex_step_kernel
  int device;
  float* d_state;   // caller-owned
  float* d_snap;    // state words x (n_terrain or 1) floats, word-major; then, with a latency, [100][43][records] rings
  hipEvent_t ev0, ev1;
  int timing;
  int have_timing;
  // ring of event pairs around the last REX_TIMING_RING launches: per-launch durations without a host sync in between
  unsigned long long* d_clock;   // [REX_TIMING_RING][2] device-side (min start, max end) ticks, rex_set_timing(3)
  int32_t* d_perm;   // regrouping (large batches only): wave slot -> env, and the per-env sweep counts it is sorted by
  int32_t* d_sweeps;
  hipEvent_t ring0[REX_TIMING_RING], ring1[REX_TIMING_RING];
  long long timed_steps;
  int words;   // per-env state words of the config's mark
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
  return code;
}
#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) return fail(REX_EHIP, #expr ": %s", hipGetErrorString(_e)); \
  } while (0)

extern "C" {

const char* rex_last_error(void) { return g_err; }
int rex_abi_version(void) { return REX_ABI_VERSION; }

int rex_default_config(int task, int signal, int num_envs, RexConfig* cfg) {
  if (!cfg || num_envs <= 0) return fail(REX_EINVAL, "rex_default_config: bad arguments%s", "");
  if (task != REX_TASK_WALK && task != REX_TASK_GALLOP && task != REX_TASK_TURN && task != REX_TASK_POSES && task != REX_TASK_STANDUP &&
      task != REX_TASK_MIXED) return fail(REX_EINVAL, "rex_default_config: unsupported task%s", "");
  if (signal != REX_SIGNAL_IK && signal != REX_SIGNAL_OL) return fail(REX_EINVAL, "rex_default_config: unsupported signal%s", "");
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = REX_ABI_VERSION;
  cfg->num_envs = num_envs;
  cfg->task = task;
  cfg->signal = signal;
  cfg->action_repeat = (task == REX_TASK_GALLOP || task == REX_TASK_POSES) ? 6 : 5;          /* gallop_env.py:47-48, walk_env.py:34-35 */
  cfg->solver_iterations = 300 / cfg->action_repeat;             /* rex_gym_env.py:25,184 */
  cfg->sim_time_step = 0.001f;
  cfg->motor_kp = 1.0f;
  cfg->motor_kd = 0.02f;
  cfg->backwards = -1;
  cfg->target_position = 0.0f;
  cfg->seed = 0;
  cfg->auto_reset = 0;
  cfg->max_episode_steps = 0;
  cfg->distance_weight = 1.0f;                                   /* rex_gym_env.py:56-59 */
  cfg->energy_weight = task == REX_TASK_GALLOP ? 0.005f : 0.0005f;   /* gallop_env.py:45 */
  cfg->drift_weight = 2.0f;
  cfg->shake_weight = 0.005f;
  cfg->pose_index = -1;
  /* RexPosesEnv never terminates (poses_env.py:265) and its roll poses press the rolled base's edge into the upper-leg
     boxes of the low side (profiles/r02_contact_census.md): the one env where the link boxes act */
  cfg->body_contacts = task == REX_TASK_POSES;
  cfg->solver_residual_threshold = 1e-7f;                        /* PyBullet default solverResidualThreshold */
  if (task == REX_TASK_MIXED) {   /* BASELINE.json configs[4]: walk, gallop and turn; per-task repeat / sweeps / weights apply per env */
    cfg->task_mix = (1 << REX_TASK_WALK) | (1 << REX_TASK_GALLOP) | (1 << REX_TASK_TURN);
    cfg->action_repeat = 6; cfg->solver_iterations = 60;         /* the largest of the mix (loop bounds only) */
  }
  return REX_OK;
}

static int mix_tasks(const RexConfig* c, int* out) {   // tasks of a REX_TASK_MIXED config, ascending
  int n = 0;
  for (int t = 0; t < 5; ++t) if ((c->task_mix >> t) & 1) out[n++] = t;
  return n;
}

int rex_action_dim(const RexConfig* c) {
  if (!c) return REX_EINVAL;
  if (c->task == REX_TASK_MIXED) {          /* as wide as the widest task of the mix */
    int ts[5], best = REX_EINVAL;
    RexConfig one = *c;
    for (int k = 0, n = mix_tasks(c, ts); k < n; ++k) { one.task = ts[k]; int d = rex_action_dim(&one); if (d > best) best = d; }
    return best;
  }
  if (c->task == REX_TASK_WALK) return c->signal == REX_SIGNAL_IK ? 2 : 8;      /* walk_env.py:104-112 */
  if (c->task == REX_TASK_GALLOP) return c->signal == REX_SIGNAL_IK ? 2 : 4;    /* gallop_env.py:119-130 */
  if (c->task == REX_TASK_TURN) return 2;                                       /* turn_env.py:100-110 */
  if (c->task == REX_TASK_POSES || c->task == REX_TASK_STANDUP) return 1;       /* poses_env.py:115-117, standup_env.py:99-101 */
  return REX_EINVAL;
}
int rex_num_motors(const RexConfig* c) {
  if (!c) return REX_EINVAL;
  return c->mark == REX_MARK_ARM ? REX_NUM_MOTORS_ARM : REX_NUM_MOTORS;         /* mark_constants.py MARK_DETAILS['motors_num'] */
}
int rex_state_words(const RexConfig* c) {
  if (!c) return REX_EINVAL;
  return c->mark == REX_MARK_ARM ? rex::Lay<REX_NUM_MOTORS_ARM>::WORDS : REX_STATE_WORDS;
}
int rex_obs_dim(const RexConfig* c) {
  if (!c) return REX_EINVAL;
  if (c->task == REX_TASK_MIXED) return ((c->task_mix >> REX_TASK_GALLOP) & 1) ? 4 + rex_num_motors(c) : 4;
  return c->task == REX_TASK_GALLOP ? 4 + rex_num_motors(c) : 4;                 /* gallop_env.py:349-356 */
}

// Envs per wave (measured on MI355X, walk-IK, steady state; REX_ENVS_PER_WAVE = 4, 8, 16 or 64 overrides).  A wave with
// EPW <= 16 envs spends its other lanes on the per-leg / per-row parallelism inside an env (rex_device.h: 8 lanes per
// env for EPW <= 8, 4 for EPW = 16), which cuts the instructions a lone wave has to issue per env.  Up to 4 096 envs the
// launch is one wave per SIMD and bound by its slowest wave: 4 envs per wave.  From 16 384 envs on, 16 envs per wave
// (4 workgroups per CU fit in LDS) also has the best throughput: 57 M env-steps/s at 131 072 envs against 33 M for the
// one-env-per-lane kernel (EPW = 64), whose 139 KB of LDS rows leave one wave per CU.
static int pick_envs_per_wave(int n) {
  const char* ov = getenv("REX_ENVS_PER_WAVE");
  if (ov) { int v = atoi(ov); if (v == 4 || v == 8 || v == 16 || v == 64) return v; }
  if (n <= 4096) return 4;
  if (n <= 8192) return 8;
  return 16;
}

// kernel instantiations by (envs per wave, mark); the arm rows fit 4 or 16 envs per workgroup in LDS
static void launch_step(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
static void launch_settle(RexSim* s, int nrec, hipStream_t st, float* snap);

// RexConfig carries its settings as float32; the reference computes its substep counts from the Python floats the
// caller wrote (int(0.5 / 0.001) = 500, int(0.02 / 0.001) = 20), whose float32 quotients fall just below the integer.
// The counts are therefore taken in double on the shortest decimal each float came from.
static double as_written(float x) {
  char buf[40];
  snprintf(buf, sizeof buf, "%.7g", (double)x);
  return strtod(buf, nullptr);
}
static size_t snapshot_floats(const RexSim* s, int nrec);

static int validate(const RexConfig* c) {
  if (!c) return fail(REX_EINVAL, "null config%s", "");
  if (c->abi_version != REX_ABI_VERSION) return fail(REX_EINVAL, "RexConfig.abi_version mismatch%s", "");
  if (c->num_envs <= 0) return fail(REX_EINVAL, "num_envs must be positive%s", "");
  if (rex_action_dim(c) < 0) return fail(REX_EINVAL, "unsupported task/signal%s", "");
  if (c->action_repeat <= 0 || c->solver_iterations <= 0 || !(c->sim_time_step > 0.0f))
    return fail(REX_EINVAL, "action_repeat, solver_iterations and sim_time_step must be positive%s", "");
  if (c->mark != REX_MARK_BASE && c->mark != REX_MARK_ARM) return fail(REX_EINVAL, "unknown mark%s", "");
  if ((long long)c->num_envs * 128 >= (1ll << 30)) return fail(REX_EINVAL, "num_envs too large for 32-bit state offsets%s", "");
  if (c->gait_clock_scale < 0.0f) return fail(REX_EINVAL, "gait_clock_scale must be >= 0%s", "");
  if (c->body_contacts && c->task == REX_TASK_MIXED) return fail(REX_EINVAL, "body_contacts is not offered together with REX_TASK_MIXED%s", "");
  for (int k = 0; k < 5; ++k) if (c->noise_stdev[k] < 0.0f) return fail(REX_EINVAL, "noise_stdev must be >= 0%s", "");
  if (c->task == REX_TASK_MIXED) {
    const int allowed = (1 << REX_TASK_WALK) | (1 << REX_TASK_GALLOP) | (1 << REX_TASK_TURN);   // tasks that share one reset pose per signal
    if (c->task_mix == 0 || (c->task_mix & ~allowed)) return fail(REX_EINVAL, "task_mix must be a non-empty subset of {walk, gallop, turn}%s", "");
  }
  if (c->mass_scale_lo < 0.0f || c->mass_scale_hi < c->mass_scale_lo || c->friction_lo < 0.0f || c->friction_hi < c->friction_lo)
    return fail(REX_EINVAL, "randomisation ranges must satisfy 0 <= lo <= hi%s", "");
  return REX_OK;
}

// a snapshot record: the state words, and behind all records the observation rings the reset motion leaves behind
static size_t snapshot_floats(const RexSim* s, int nrec) {
  const bool ring = s->cfg.pd_latency > 0.0f || s->cfg.control_latency > 0.0f;
  return (size_t)nrec * ((size_t)s->words + (ring ? (size_t)REX_HISTORY_LEN * (size_t)s->dev.hist_words : 0));
}

int rex_create(const RexConfig* cfg, int device, float* d_state, void* stream, RexSim** out) {
  if (!out || !d_state) return fail(REX_EINVAL, "rex_create: null pointer%s", "");
  int rc = validate(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(REX_ENODEV, "no HIP device visible%s", "");
  if (device < 0 || device >= ndev) return fail(REX_EINVAL, "device index out of range%s", "");
  HIPCHK(hipSetDevice(device));
  RexSim* s = new RexSim();
  s->cfg = *cfg;
  s->device = device;
  s->d_state = d_state;
  s->timing = 0;
  s->have_timing = 0;
  rex::DevCfg& d = s->dev;
  d.n = cfg->num_envs; d.env_index_base = cfg->env_index_base; d.task = cfg->task; d.signal = cfg->signal;
  d.action_repeat = cfg->action_repeat; d.iterations = cfg->solver_iterations; d.dt = cfg->sim_time_step;
  d.kp = cfg->motor_kp; d.kd = cfg->motor_kd; d.res_thr = sqrtf(fmaxf(cfg->solver_residual_threshold, 0.0f)); d.backwards = cfg->backwards; d.target_position = cfg->target_position;
  d.seed_lo = (uint32_t)cfg->seed; d.seed_hi = (uint32_t)(cfg->seed >> 32);
  d.auto_reset = cfg->auto_reset; d.max_steps = cfg->max_episode_steps;
  d.w_dist = cfg->distance_weight; d.w_energy = cfg->energy_weight; d.w_drift = cfg->drift_weight; d.w_shake = cfg->shake_weight;
  d.action_dim = rex_action_dim(cfg); d.obs_dim = rex_obs_dim(cfg);
  s->epw = pick_envs_per_wave(cfg->num_envs);
  if ((cfg->mark == REX_MARK_ARM || cfg->task == REX_TASK_MIXED || cfg->body_contacts) && s->epw > 16) s->epw = 16;
  if (cfg->body_contacts && s->epw > 8) s->epw = 8;   // the link-box rows of 16 envs take 76.8 KB of LDS: two workgroups per CU, half the SIMDs idle
  if (cfg->body_contacts && cfg->mark == REX_MARK_ARM) s->epw = 4;   // mark arm: 42.6 KB at 8 envs per wave would leave room for three workgroups per CU
  d.n_mix = 1; d.mix_task[0] = cfg->task; d.max_repeat = cfg->action_repeat; d.max_iterations = cfg->solver_iterations;
  for (int k = 1; k < 5; ++k) d.mix_task[k] = cfg->task;
  if (cfg->task == REX_TASK_MIXED) {
    int ts[5];
    d.n_mix = mix_tasks(cfg, ts);
    d.max_repeat = 0;
    for (int k = 0; k < d.n_mix; ++k) { d.mix_task[k] = ts[k]; const int r = rex::task_action_repeat(ts[k]); if (r > d.max_repeat) d.max_repeat = r; }
    for (int k = d.n_mix; k < 5; ++k) d.mix_task[k] = ts[0];
    d.action_repeat = d.max_repeat;          // per-env values replace these inside the mixed kernel
    d.iterations = d.max_iterations = 60;
  }
  d.mass_lo = cfg->mass_scale_lo; d.mass_hi = cfg->mass_scale_hi; d.mu_lo = cfg->friction_lo; d.mu_hi = cfg->friction_hi;
  s->words = rex_state_words(cfg);
  d.pose_index = cfg->pose_index; d.pose_value = cfg->pose_value;
  d.range_normalize = cfg->range_normalize;
  d.terrain = nullptr; d.terrain_mid = nullptr; d.n_terrain = 0; d.body_params = nullptr;
  d.perm = nullptr; d.sweeps = nullptr; s->d_perm = nullptr; s->d_sweeps = nullptr; d.clock = nullptr; s->d_clock = nullptr;
  d.geo = rex::HfGeom{256, 20.0f, 20.0f, 127.5f, 127.5f, 254.999f, 254.999f}; d.hf_stride = 65536;   /* model/terrain.py:32-54 */
  d.init_z = cfg->init_height > 0.0f ? cfg->init_height : rex::kInitZ;
  d.anchor = 0.0f;
  if (cfg->on_rack) { d.init_z = 1.0f; d.anchor = rex::kRackAnchor; }   /* INIT_RACK_POSITION, rex.py:11 */
  d.noise_on = 0;
  for (int k = 0; k < 5; ++k) { d.noise[k] = cfg->noise_stdev[k]; if (cfg->noise_stdev[k] > 0.0f) d.noise_on = 1; }
  d.hist = nullptr; d.pd_latency = cfg->pd_latency; d.control_latency = cfg->control_latency;
  d.hist_words = 3 * rex_num_motors(cfg) + 7;
  {
    const double dt = as_written(cfg->sim_time_step), pl = as_written(cfg->pd_latency), cl = as_written(cfg->control_latency);
    d.pd_slots = (int)(pl / dt); d.pd_alpha = (float)((pl - d.pd_slots * dt) / dt);
    d.control_slots = (int)(cl / dt); d.control_alpha = (float)((cl - d.control_slots * dt) / dt);
    d.reset_substeps = (int)(0.5 / dt);
  }
  {
    float b;   /* walk_env.py:104-114, gallop_env.py:119-130 (low=+b, high=-b), turn_env.py:100-110, poses_env.py:115-117 */
    if (cfg->task == REX_TASK_WALK) b = cfg->signal == REX_SIGNAL_IK ? 0.4f : 0.01f;
    else if (cfg->task == REX_TASK_GALLOP) b = cfg->signal == REX_SIGNAL_IK ? -0.4f : -0.3f;
    else if (cfg->task == REX_TASK_TURN) b = 0.01f;
    else b = 0.1f;
    d.act_lo = -b; d.act_hi = b;
    d.obs_hi_ang = (float)(2.0 * M_PI) + 0.01f;                      /* walk_env.py:364-378 + OBSERVATION_EPS */
    d.obs_hi_rate = (float)(2.0 * M_PI) / cfg->sim_time_step + 0.01f;
  }
  d.target_orient = cfg->target_orient; d.init_orient = cfg->init_orient; d.orient_fixed = cfg->orient_fixed;
  if (cfg->on_rack) { d.init_orient = 2.1f; d.orient_fixed |= 2; }   /* turn_env.py:140-143 */
  d.gait_clock = cfg->gait_clock_scale > 0.0f ? cfg->gait_clock_scale : 1.0f;
  {
    // Regrouping (opt-in: REX_REGROUP=1) can pay once a SIMD runs several waves one after the other -- below that the launch
    // ends with its slowest wave whatever the grouping.  Measured on MI355X (profiles/r02_regroup.md): +10 % on the walking
    // workload (gait clock 1.5, sweep counts 0.96 correlated from step to step), -11 % on the falling one (0.52), where the
    // scattered state access and the sort cost more than the 18 % of sweeps the grouping can save.
    const char* ov = getenv("REX_REGROUP");
    const bool want = ov ? atoi(ov) != 0 : false;
    if (want) {
      hipError_t e2 = hipMalloc(&s->d_perm, sizeof(int32_t) * (size_t)cfg->num_envs);
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_sweeps, sizeof(int32_t) * (size_t)cfg->num_envs);
      if (e2 != hipSuccess) { delete s; return fail(REX_ENOMEM, "hipMalloc(regroup): %s", hipGetErrorString(e2)); }
      hipLaunchKernelGGL(rex::rex_iota_kernel, dim3((cfg->num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, cfg->num_envs, s->d_perm, s->d_sweeps);
      d.perm = s->d_perm; d.sweeps = s->d_sweeps;
    }
  }
  hipError_t e = hipMalloc(&s->d_snap, sizeof(float) * snapshot_floats(s, d.n_mix));
  if (e != hipSuccess) { delete s; return fail(REX_ENOMEM, "hipMalloc(snapshot): %s", hipGetErrorString(e)); }
  (void)hipEventCreate(&s->ev0);
  (void)hipEventCreate(&s->ev1);
  for (int k = 0; k < REX_TIMING_RING; ++k) { s->ring0[k] = nullptr; s->ring1[k] = nullptr; }
  s->timed_steps = 0;
  hipStream_t st = (hipStream_t)stream;
  launch_settle(s, d.n_mix, st, s->d_snap);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemsetAsync(d_state, 0, sizeof(float) * (size_t)s->words * cfg->num_envs, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)hipFree(s->d_snap);
    delete s;
    return fail(REX_EHIP, "settle kernel: %s", hipGetErrorString(e));
  }
  *out = s;
  return REX_OK;
}

static int install_terrain(RexSim* s, const float* d_heights, const float* d_mids, int k, void* stream) {
  HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  float* snap = nullptr;
  const int nrec = (k > 0 ? k : 1) * s->dev.n_mix;
  HIPCHK(hipMalloc(&snap, sizeof(float) * snapshot_floats(s, nrec)));
  HIPCHK(hipStreamSynchronize(st));
  (void)hipFree(s->d_snap);
  s->d_snap = snap;
  s->dev.terrain = k > 0 ? d_heights : nullptr;
  s->dev.terrain_mid = k > 0 ? d_mids : nullptr;
  s->dev.n_terrain = k;
  launch_settle(s, nrec, st, s->d_snap);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  return REX_OK;
}

int rex_set_terrain(RexSim* s, const float* d_heights, const float* d_mids, int k, void* stream) {
  if (!s || k < 0 || (k > 0 && (!d_heights || !d_mids))) return fail(REX_EINVAL, "rex_set_terrain: bad arguments%s", "");
  s->dev.geo = rex::HfGeom{256, 20.0f, 20.0f, 127.5f, 127.5f, 254.999f, 254.999f};   /* 256 x 256 vertices, 5 cm cells, centred */
  s->dev.hf_stride = 65536;
  return install_terrain(s, d_heights, d_mids, k, stream);
}

int rex_set_heightfield(RexSim* s, const float* d_heights, const float* d_mids, int k, int nx, int ny, float cell_x, float cell_y,
                        float origin_x, float origin_y, void* stream) {
  if (!s || k <= 0 || !d_heights || !d_mids || nx < 2 || ny < 2 || !(cell_x > 0.0f) || !(cell_y > 0.0f) || (long long)nx * ny > (1ll << 26))
    return fail(REX_EINVAL, "rex_set_heightfield: bad arguments%s", "");
  rex::HfGeom g;
  g.nx = nx; g.inv_cx = 1.0f / cell_x; g.inv_cy = 1.0f / cell_y;
  g.off_x = 0.5f * (float)(nx - 1) - origin_x * g.inv_cx; g.off_y = 0.5f * (float)(ny - 1) - origin_y * g.inv_cy;
  g.max_x = (float)(nx - 1) - 0.001f; g.max_y = (float)(ny - 1) - 0.001f;
  s->dev.geo = g;
  s->dev.hf_stride = nx * ny;
  return install_terrain(s, d_heights, d_mids, k, stream);
}

int rex_set_history(RexSim* s, float* d_history) {
  if (!s) return fail(REX_EINVAL, "rex_set_history: null sim%s", "");
  // without a latency the delayed observation IS the newest one (rex.py:744-745): the ring is not needed, and the
  // snapshot holds none to restore from
  s->dev.hist = (s->cfg.pd_latency > 0.0f || s->cfg.control_latency > 0.0f) ? d_history : nullptr;
  return REX_OK;
}

int rex_set_body_params(RexSim* s, const float* d_params) {
  if (!s) return fail(REX_EINVAL, "rex_set_body_params: null sim%s", "");
  s->dev.body_params = d_params;
  return REX_OK;
}

int rex_destroy(RexSim* s) {
  if (!s) return REX_OK;
  (void)hipSetDevice(s->device);
  (void)hipFree(s->d_snap);
  if (s->d_clock) (void)hipFree(s->d_clock);
  if (s->d_perm) (void)hipFree(s->d_perm);
  if (s->d_sweeps) (void)hipFree(s->d_sweeps);
  (void)hipEventDestroy(s->ev0);
  (void)hipEventDestroy(s->ev1);
  for (int k = 0; k < REX_TIMING_RING; ++k) if (s->ring0[k]) { (void)hipEventDestroy(s->ring0[k]); (void)hipEventDestroy(s->ring1[k]); }
  delete s;
  return REX_OK;
}

int rex_reset(RexSim* s, const int32_t* d_indices, int n, float* d_obs, void* stream) {
  if (!s) return fail(REX_EINVAL, "rex_reset: null sim%s", "");
  if ((s->cfg.pd_latency > 0.0f || s->cfg.control_latency > 0.0f) && !s->dev.hist)
    return fail(REX_EINVAL, "rex_reset: pd_latency/control_latency are set but rex_set_history() was not called%s", "");
  const int count = d_indices ? n : s->cfg.num_envs;
  if (count <= 0) return d_indices ? REX_OK : fail(REX_EINVAL, "rex_reset: empty%s", "");
  HIPCHK(hipSetDevice(s->device));
  const int block = 256;
  if (s->cfg.mark == REX_MARK_ARM)
    hipLaunchKernelGGL(rex::rex_reset_kernel<18>, dim3((count + block - 1) / block), dim3(block), 0, (hipStream_t)stream, s->dev,
                       s->d_state, s->d_snap, d_indices, count, d_obs);
  else
    hipLaunchKernelGGL(rex::rex_reset_kernel<12>, dim3((count + block - 1) / block), dim3(block), 0, (hipStream_t)stream, s->dev,
                       s->d_state, s->d_snap, d_indices, count, d_obs);
  HIPCHK(hipGetLastError());
  return REX_OK;
}

int rex_step(RexSim* s, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream) {
  if (!s || !d_action || !d_obs || !d_reward || !d_done) return fail(REX_EINVAL, "rex_step: null pointer%s", "");
  HIPCHK(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (s->cfg.num_envs + s->epw - 1) / s->epw;
  hipEvent_t e0 = s->ev0, e1 = s->ev1;
  if (s->timing == 2) {
    const int k = (int)(s->timed_steps % REX_TIMING_RING);
    if (!s->ring0[k]) { HIPCHK(hipEventCreate(&s->ring0[k])); HIPCHK(hipEventCreate(&s->ring1[k])); }
    e0 = s->ring0[k]; e1 = s->ring1[k];
  }
  if (s->timing == 3 && s->timed_steps < REX_TIMING_RING) {   // device-side timestamps: this launch's (min start, max end) slot
    s->dev.clock = s->d_clock + 2 * s->timed_steps;           // (all slots were primed by rex_set_timing: nothing is copied per
    s->timed_steps++;                                         // launch, the queue stays as full as in an untimed run)
  } else s->dev.clock = nullptr;
  if (s->timing == 1 || s->timing == 2) HIPCHK(hipEventRecord(e0, st));
  launch_step(s, blocks, st, d_action, d_obs, d_reward, d_done, d_motor_cmd);
  {   // developer probe (tools/launch_gap.py): REX_STEP_REPEAT=R issues the launch R times back to back from C
    static const int repeat = getenv("REX_STEP_REPEAT") ? atoi(getenv("REX_STEP_REPEAT")) : 1;
    for (int k = 1; k < repeat; ++k) launch_step(s, blocks, st, d_action, d_obs, d_reward, d_done, d_motor_cmd);
  }
  HIPCHK(hipGetLastError());
  if (s->d_perm)   // next step's grouping from this step's sweep counts (stream-ordered behind the step)
    hipLaunchKernelGGL(rex::rex_regroup_kernel, dim3(1), dim3(1024), 0, st, s->cfg.num_envs, (s->dev.max_repeat * s->dev.max_iterations + 63) / 64,
                       s->d_sweeps, s->d_perm);
  if (s->timing == 1 || s->timing == 2) { HIPCHK(hipEventRecord(e1, st)); s->have_timing = 1; if (s->timing == 2) s->timed_steps++; }
  return REX_OK;
}

int rex_envs_per_wave(const RexSim* s) { return s ? s->epw : REX_EINVAL; }

int rex_get_sweeps(RexSim* s, int32_t* d_out, void* stream) {
  if (!s || !d_out) return fail(REX_EINVAL, "rex_get_sweeps: null pointer%s", "");
  if (!s->d_sweeps) return fail(REX_EINVAL, "rex_get_sweeps: this sim does not regroup (REX_REGROUP=1 was not set when it was created)%s", "");
  HIPCHK(hipMemcpyAsync(d_out, s->d_sweeps, sizeof(int32_t) * (size_t)s->cfg.num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return REX_OK;
}

int rex_set_timing(RexSim* s, int enable) {
  if (!s) return fail(REX_EINVAL, "rex_set_timing: null sim%s", "");
  s->timing = (enable == 2 || enable == 3) ? enable : (enable ? 1 : 0);
  s->have_timing = 0;
  s->timed_steps = 0;
  if (s->timing == 3) {   // the next REX_TIMING_RING launches are timed on the device; prime their (min, max) slots
    if (!s->d_clock) HIPCHK(hipMalloc(&s->d_clock, sizeof(unsigned long long) * 2 * REX_TIMING_RING));
    static unsigned long long init[2 * REX_TIMING_RING];
    for (int k = 0; k < REX_TIMING_RING; ++k) { init[2 * k] = ~0ull; init[2 * k + 1] = 0ull; }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(s->d_clock, init, sizeof(init), hipMemcpyHostToDevice));
  }
  return REX_OK;
}

int rex_step_times_ms(RexSim* s, float* ms, int max_count) {
  if (!s || !ms || max_count <= 0) return fail(REX_EINVAL, "rex_step_times_ms: bad arguments%s", "");
  if ((s->timing != 2 && s->timing != 3) || s->timed_steps == 0) return 0;
  long long have = s->timed_steps < REX_TIMING_RING ? s->timed_steps : REX_TIMING_RING;
  int n = (int)(have < max_count ? have : max_count);
  if (s->timing == 3) {
    static unsigned long long ticks[2 * REX_TIMING_RING];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ticks, s->d_clock, sizeof(ticks), hipMemcpyDeviceToHost));
    int khz = 100000;   // s_memrealtime: constant 100 MHz on gfx9
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device);
    for (int j = 0; j < n; ++j) {
      const int k = (int)(s->timed_steps - n + j);
      ms[j] = (float)((double)(ticks[2 * k + 1] - ticks[2 * k]) / (double)khz);
    }
    return n;
  }
  HIPCHK(hipEventSynchronize(s->ring1[(int)((s->timed_steps - 1) % REX_TIMING_RING)]));
  for (int j = 0; j < n; ++j) {   // oldest of the last n first
    const int k = (int)((s->timed_steps - n + j) % REX_TIMING_RING);
    HIPCHK(hipEventElapsedTime(&ms[j], s->ring0[k], s->ring1[k]));
  }
  return n;
}

int rex_last_step_ms(RexSim* s, float* ms) {
  if (!s || !ms) return fail(REX_EINVAL, "rex_last_step_ms: null pointer%s", "");
  if (!s->have_timing) return fail(REX_EINVAL, "rex_last_step_ms: no timed step recorded%s", "");
  HIPCHK(hipEventSynchronize(s->ev1));
  HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return REX_OK;
}

int rex_ik_solve(int n, const float* d_orn, const float* d_pos, const float* d_frames, float* d_angles, void* stream) {
  if (n <= 0 || !d_orn || !d_pos || !d_frames || !d_angles) return fail(REX_EINVAL, "rex_ik_solve: bad arguments%s", "");
  hipLaunchKernelGGL(rex::rex_ik_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_orn, d_pos, d_frames, d_angles);
  HIPCHK(hipGetLastError());
  return REX_OK;
}

int rex_motor_torque(int n, const float* d_cmd, const float* d_q, const float* d_qd, const float* d_qd_true, float kp, float kd,
                     float* d_actual, float* d_observed, void* stream) {
  if (n <= 0 || !d_cmd || !d_q || !d_qd || !d_qd_true || !d_actual || !d_observed) return fail(REX_EINVAL, "rex_motor_torque: bad arguments%s", "");
  hipLaunchKernelGGL(rex::rex_motor_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_cmd, d_q, d_qd, d_qd_true,
                     kp, kd, d_actual, d_observed);
  HIPCHK(hipGetLastError());
  return REX_OK;
}

int rex_gait_loop(int n, int mode, float* d_planner, const float* d_params, float* d_frames_out, void* stream) {
  if (n <= 0 || (mode != 0 && mode != 1) || !d_planner || !d_params || !d_frames_out) return fail(REX_EINVAL, "rex_gait_loop: bad arguments%s", "");
  hipLaunchKernelGGL(rex::rex_gait_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, mode, d_planner, d_params, d_frames_out);
  HIPCHK(hipGetLastError());
  return REX_OK;
}

#ifdef REX_PROF   /* developer build only (tools/prof_sections.py): cycle counters of the sections of a substep */
REX_API int rex_debug_prof(long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_prof), sizeof(long long) * 10 * 1024) != hipSuccess) return REX_EHIP;
  if (reset) { static long long z[10 * 1024]; if (hipMemcpyToSymbol(HIP_SYMBOL(rex::g_prof), z, sizeof(z)) != hipSuccess) return REX_EHIP; }
  return REX_OK;
}
#endif
}  // extern "C"

#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY)                                                                                  \
  hipLaunchKernelGGL((rex::rex_step_kernel<EPW, ARM, MIXED, BODY>), dim3(blocks), dim3(REX_WAVE), 0, st, s->dev, s->d_state, s->d_snap, \
                     a, o, r, d, m)
#define REX_LAUNCH_BY_EPW(ARM, MIXED, BODY)                                           \
  do {                                                                                \
    if (s->epw == 4) REX_LAUNCH_STEP(4, ARM, MIXED, BODY);                            \
    else if (s->epw == 8) REX_LAUNCH_STEP(8, ARM, MIXED, BODY);                       \
    else REX_LAUNCH_STEP(16, ARM, MIXED, BODY);                                       \
  } while (0)
static void launch_step(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
#ifdef REX_FAST_BUILD   /* developer A/B builds: the 4-envs-per-wave base kernel only (2: its link-box variant, 3: mark arm) */
  REX_LAUNCH_STEP(REX_FAST_EPW, REX_FAST_BUILD == 3 || REX_FAST_BUILD == 4, REX_FAST_BUILD == 4, REX_FAST_BUILD == 2);   /* 4: mixed tasks, mark arm */
  return;
#else
  const bool arm = s->cfg.mark == REX_MARK_ARM;
  if (s->cfg.task == REX_TASK_MIXED) {   // lane groups only (rex_create caps the envs per wave at 16)
    if (arm) REX_LAUNCH_BY_EPW(true, true, false); else REX_LAUNCH_BY_EPW(false, true, false);
  } else if (s->cfg.body_contacts) {     // link-box contact rows: 4 or 8 envs per wave (rex_create caps it)
    if (arm) REX_LAUNCH_STEP(4, true, false, true);
    else { if (s->epw == 4) REX_LAUNCH_STEP(4, false, false, true); else REX_LAUNCH_STEP(8, false, false, true); }
  } else if (arm) {
    REX_LAUNCH_BY_EPW(true, false, false);
  } else {
    if (s->epw == 64) REX_LAUNCH_STEP(64, false, false, false);
    else REX_LAUNCH_BY_EPW(false, false, false);
  }
#endif
}
static void launch_settle(RexSim* s, int nrec, hipStream_t st, float* snap) {
#ifdef REX_FAST_BUILD
#if REX_FAST_BUILD == 2
  hipLaunchKernelGGL((rex::rex_settle_kernel<false, true>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
#elif REX_FAST_BUILD == 3 || REX_FAST_BUILD == 4
  hipLaunchKernelGGL((rex::rex_settle_kernel<true, false>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
#else
  hipLaunchKernelGGL((rex::rex_settle_kernel<false, false>), dim3((nrec + REX_WAVE - 1) / REX_WAVE), dim3(REX_WAVE), 0, st, s->dev, snap);
#endif
  return;
#endif
  const bool arm = s->cfg.mark == REX_MARK_ARM, body = s->cfg.body_contacts != 0;
  if (arm && body) hipLaunchKernelGGL((rex::rex_settle_kernel<true, true>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
  else if (arm) hipLaunchKernelGGL((rex::rex_settle_kernel<true, false>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
  else if (body) hipLaunchKernelGGL((rex::rex_settle_kernel<false, true>), dim3((nrec + 15) / 16), dim3(REX_WAVE), 0, st, s->dev, snap);
  else hipLaunchKernelGGL((rex::rex_settle_kernel<false, false>), dim3((nrec + REX_WAVE - 1) / REX_WAVE), dim3(REX_WAVE), 0, st, s->dev, snap);
}
