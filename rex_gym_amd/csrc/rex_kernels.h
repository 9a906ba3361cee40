// rex_kernels.h -- the kernels of the batched Rex simulator as templates, shared by the translation units that instantiate
// them (rex_step_*.hip, rex_settle.hip, rexsim.hip): the library is compiled variant group by variant group, in parallel
// (rex_gym_amd/build.py), because one translation unit with every instantiation takes minutes.
//
// Kernel map
//   rex_step_kernel     one env.step() per lane group: action -> staged gait -> Bezier/IK targets ->
//                       action_repeat x (motor model + restated stepSimulation) -> reward / done /
//                       observation (+ optional in-launch reset).  State is read once and written
//                       once per env.step (SoA, coalesced); constraint rows live in LDS.
//   rex_settle_kernel   the reference's 100 + 500 substep reset motion (rex.py:314-323), run once.
//   rex_reset_kernel    snapshot restore + per-episode draws (walk_env.py:125-154).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

// Floating-point contraction is fixed BY THE SOURCE, for every translation unit of the library and every tool that compiles one
// (tools/kres.sh, kstat.sh, check_dpp_masks.py): `a * b + c` inside one expression is one fma, nothing else is fused.  hipcc's
// default (fast-honor-pragmas) lets the backend fuse whatever multiply-add pairs it finds after inlining, and which ones it finds
// depends on the code around them: the `_trace` instantiations of the step kernels (-DREX_TU_TRACE=1: the same source plus a few
// integer taps) came out with other fma / mul+add mixes than the product kernels and parted from them in the last bit after one
// step (round 5, tests/test_gpu_parity.py::test_trace_kernels_are_bit_identical_to_the_product_kernels).  With the contraction
// in the source the two are bit-identical, the kernels are 1-4 % faster (fewer packed-fp32 pairs, 30-90 fewer registers in the
// mark-arm kernels) -- and a kernel's arithmetic no longer depends on what else was compiled into it.
#pragma clang fp contract(on)

#include "rex_device.h"
#include "rex_arm_device.h"
#include "rex_controller.h"


#define REX_CLOCK_SLOTS 4096    /* launches timed by rex_set_timing(3) */
#define REX_CLOCK_WAYS 64       /* (start, end) tick pairs per launch: workgroup b folds into pair b % 64 -- one shared pair serialises 1 024 atomics */

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
  float phi, alpha;              // GaitPlanner._phi (float copy; its `>= 0.99` test lives in REX_F_PHASE_WRAP), _alpha
  int32_t last_step, end_step;   // env steps whose clock GaitPlanner._last_time / the env's end_time hold (rexsim.h, "Clocks")
  float target, aux;
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

// What a lane carries of the MOTOR side of its env (commands, observed torques, overheat counters).  One env per lane
// (NL = 4): all four legs.  Lane groups (NL = 1): the lane's OWN leg only -- leg `leg0` of the motor order FL, FR, RL, RR;
// the group's lanes split the controller (Bezier + IK), the motor model and the observation words by leg exactly as they
// split the leg factorisations of the physics, and meet again in the reward's dot product (a DPP sum) and in the state
// words lane 0 stores.  The 6 arm motors of mark 'arm' are carried by every lane (local indices 3 NL ...).
template <int NL, bool ARM>
struct MotorSide {
  static constexpr int NLM = 3 * NL, NA = ARM ? 6 : 0, N = NLM + NA;
  float cmd[N], tau_obs[N];
  // overheat counters (<= 65 535, rex.py:601-608), two to a register: they only sit through the solver sweeps, where every
  // register counts (mark 'arm': 5 instead of 9)
  uint32_t overheat2[(N + 1) / 2];
  __device__ __forceinline__ uint32_t heat(int jl) const { return (jl & 1) ? overheat2[jl >> 1] >> 16 : overheat2[jl >> 1] & 0xFFFFu; }
  __device__ __forceinline__ void set_heat(int jl, uint32_t v) {
    v &= 0xFFFFu;   // (callers hand in <= 65 535 -- the state packs two u16 per word and the increment saturates --; the mask folds away)
    overheat2[jl >> 1] = (jl & 1) ? (overheat2[jl >> 1] & 0xFFFFu) | (v << 16) : (overheat2[jl >> 1] & 0xFFFF0000u) | v;
  }
  __device__ __forceinline__ void clear_heat() {
#pragma unroll
    for (int k = 0; k < (N + 1) / 2; ++k) overheat2[k] = 0u;
  }
  // motor number (mark_constants.py order) of local index jl
  __device__ __forceinline__ static int motor(int leg0, int jl) { return jl < NLM ? (NL == 4 ? jl : 3 * leg0 + jl) : 12 + (jl - NLM); }
};
// entry jl of a per-motor array of all 12 leg motors as this lane sees its legs (NL = 1: a select, no dynamic indexing)
template <int NL>
__device__ __forceinline__ float legv(const float* a, int leg0, int jl) { (void)leg0; return a[jl]; }   // (lane groups keep their leg in slots 0..2)
template <int NL>
__device__ __forceinline__ uint32_t legu(const uint32_t* a, int leg0, int jl) {
  if (NL == 4) return a[jl];
  const uint32_t lo = leg0 & 1 ? a[3 + jl] : a[jl], hi = leg0 & 1 ? a[9 + jl] : a[6 + jl];
  return leg0 & 2 ? hi : lo;
}
// this lane's share of the counters loaded with the state, and back (lane groups: through the hand-over chunks in LDS)
template <int NL, bool ARM>
__device__ __forceinline__ void take_overheat(const EnvState& e, int leg0, MotorSide<NL, ARM>& ms) {
  ms.clear_heat();
#pragma unroll
  for (int jl = 0; jl < 3 * NL; ++jl) ms.set_heat(jl, legu<NL>(e.overheat, leg0, jl));
#pragma unroll
  for (int a = 0; a < (ARM ? 6 : 0); ++a) ms.set_heat(3 * NL + a, e.overheat[12 + a]);
}

// NL = 4: the whole env (one env per lane; reset kernel).  NL = 1 (lane groups): the joint state of leg `leg0` only, into
// slots 0..2 of q / qd (PhysState), next to the arm's
template <int NM, int NL = 4>
__device__ __forceinline__ void load_env(const float* st, int n, int i, EnvState& e, int leg0 = 0) {
  using Y = Lay<NM>;
#pragma unroll
  for (int k = 0; k < 3; ++k) { e.ph.pos[k] = ldw(st, n, REX_S_POS + k, i); e.ph.lin[k] = ldw(st, n, REX_S_LINVEL + k, i); e.ph.ang[k] = ldw(st, n, REX_S_ANGVEL + k, i); }
#pragma unroll
  for (int k = 0; k < 4; ++k) e.ph.quat[k] = ldw(st, n, REX_S_QUAT + k, i);
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    if (NL == 4 || j >= 12) { e.ph.q[j] = ldw(st, n, Y::Q + j, i); e.ph.qd[j] = ldw(st, n, Y::QD + j, i); }
    else if (j < 3) { e.ph.q[j] = ldw(st, n, Y::Q + 3 * leg0 + j, i); e.ph.qd[j] = ldw(st, n, Y::QD + 3 * leg0 + j, i); }
    else { e.ph.q[j] = 0.0f; e.ph.qd[j] = 0.0f; }   // lane groups never touch slots 3..11: a read of one would at least be deterministic
  }
  e.phi = ldw(st, n, Y::PHI, i); e.last_step = (int32_t)ldi(st, n, Y::LASTT, i); e.alpha = ldw(st, n, Y::ALPHA, i);
  e.target = ldw(st, n, Y::TARGET, i); e.end_step = (int32_t)ldi(st, n, Y::ENDTIME, i); e.aux = ldw(st, n, Y::AUX, i);
  e.flags = ldi(st, n, Y::FLAGS, i); e.steps = (int32_t)ldi(st, n, Y::STEPS, i); e.episode = (int32_t)ldi(st, n, Y::EPISODE, i);
  e.motor_en = ldi(st, n, Y::MOTOR_EN, i);
  e.hist = ldi(st, n, Y::HIST, i);
#pragma unroll
  for (int k = 0; k < NM / 2; ++k) {
    const uint32_t w = ldi(st, n, Y::OVERHEAT + k, i);
    e.overheat[2 * k] = w & 0xFFFFu; e.overheat[2 * k + 1] = w >> 16;
  }
}

// q12 / qd12: the 12 leg joints in motor order (lane groups gather them from their lanes first, gather_legs)
template <int NM>
__device__ __forceinline__ void store_env(float* st, int n, int i, const EnvState& e, const float* q12, const float* qd12) {
  using Y = Lay<NM>;
#pragma unroll
  for (int k = 0; k < 3; ++k) { stw(st, n, REX_S_POS + k, i, e.ph.pos[k]); stw(st, n, REX_S_LINVEL + k, i, e.ph.lin[k]); stw(st, n, REX_S_ANGVEL + k, i, e.ph.ang[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) stw(st, n, REX_S_QUAT + k, i, e.ph.quat[k]);
#pragma unroll
  for (int j = 0; j < NM; ++j) { stw(st, n, Y::Q + j, i, j < 12 ? q12[j] : e.ph.q[j]); stw(st, n, Y::QD + j, i, j < 12 ? qd12[j] : e.ph.qd[j]); }
  stw(st, n, Y::PHI, i, e.phi); sti(st, n, Y::LASTT, i, (uint32_t)e.last_step); stw(st, n, Y::ALPHA, i, e.alpha);
  stw(st, n, Y::TARGET, i, e.target); sti(st, n, Y::ENDTIME, i, (uint32_t)e.end_step); stw(st, n, Y::AUX, i, e.aux);
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
  int32_t nsteps;            // env.step() calls per launch (rex_step: 1; rex_step_segment: the segment's length)
  float dt, kp, kd, res_thr;
  // the reference's clocks are Python floats: time step and gait clock factor as the caller wrote them, in double
  // (host: as_written), for the controller's discrete decisions (rexsim.h, "Clocks")
  double dt_d, gait_clock_d;   // gait_clock_d: wall-clock seconds per simulated second seen by GaitPlanner.loop (gait_planner.py:108-110)
  int32_t backwards;
  float target_position;
  uint32_t seed_lo, seed_hi;
  int32_t auto_reset, max_steps;
  float w_dist, w_energy, w_drift, w_shake;
  float fwd_cap;             // forward_reward_cap (rex_gym_env.py:525); +inf: none
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
  // REX_TASK_MIXED: the tasks of the mix (task_mix bits, ascending), their number, and the largest action_repeat /
  // solver sweep cap among them (wave-uniform loop bounds; every env stops at its own)
  int32_t mix_task[5], n_mix, max_repeat, max_iterations;
  float mass_lo, mass_hi, mu_lo, mu_hi;   // per-reset randomisation ranges (lo == hi == 0: off)
  // large batches: envs are regrouped into waves by the solver sweeps they needed in the previous step (a wave sweeps
  // until the slowest of its envs has converged): wave slot k works on env perm[k]; sweeps[i] = this step's count of env i
  const int32_t* perm; int32_t* sweeps;
  // REX_TASK_MIXED: the task-sorted slot map (host: task_slot_map / task_region_map in rexsim.hip).  An env keeps its task for life, so the waves are
  // made of envs of ONE task: slot_env[blk * EPW + slot] = env of that wave slot (-1: padding), block_task[blk] = the
  // task of workgroup blk's envs -- the per-task constants are then wave-uniform (SGPRs), not per-lane registers
  const int32_t* slot_env; const int32_t* block_task;
  // rex_set_event_trace (debug; nullptr: off): [3][n] words per env -- [0] the chained hash of every substep's discrete events
  // (toe points in reach, their heightfield facets, joint / arm bounds reached), [1] of the solver sweep counts, [2] as [0]
  // without the arm's bounds
  unsigned* trace;
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
__device__ __forceinline__ void mixed_config_of_task(const DevCfg& c, int task, DevCfg& cm) {
  cm = c;
  cm.task = task;
  cm.action_repeat = task_action_repeat(cm.task);
  cm.iterations = 300 / cm.action_repeat;                       // rex_gym_env.py:25,184
  const float b = task_action_bound(cm.task, c.signal);
  cm.act_lo = -b; cm.act_hi = b;
  cm.w_energy = task_energy_weight(cm.task);
}
__device__ __forceinline__ void mixed_config(const DevCfg& c, int gidx, DevCfg& cm) { mixed_config_of_task(c, mixed_task_of(c, gidx), cm); }
// snapshot record of (terrain, task): one settled robot per terrain and -- in a mixed batch -- per task of the mix
// (the reset motion runs under the task's own numSolverIterations)
__device__ __forceinline__ int mix_slot(const DevCfg& c, int task) {
  int sl = 0;
#pragma unroll
  for (int k = 1; k < 5; ++k) if (k < c.n_mix && c.mix_task[k] == task) sl = k;
  return sl;
}

}  // namespace rex
#include "rex_policy.h"
namespace rex {

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
// the controller-facing observation (Rex._control_observation): q, qd, tau_obs of the lane's motors (MotorSide's local
// order), base quaternion, base angular velocity
template <int N> struct CtrlObs { float q[N], qd[N], tau[N], quat[4], w[3]; };

// terrain of (global env index, episode): the reference regenerates the field on every reset
// (rex_gym_env.py:347-348); here each episode picks one of the pool entries
__device__ __forceinline__ int terrain_index(const DevCfg& c, int gidx, int episode) {
  return (int)(((uint32_t)gidx + 977u * (uint32_t)episode) % (uint32_t)c.n_terrain);
}
__device__ __forceinline__ Ground env_ground(const DevCfg& c, int i, int gidx, int episode) {
  Ground g{nullptr, 0u, 0.0f, 1.0f, 1.0f, kMu, c.geo, c.anchor};
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
    g.h = c.terrain; g.off = (unsigned)t * (unsigned)c.hf_stride;   // (rex_set_heightfield: k fields x stride < 2^31)
    g.mid = c.terrain_mid[t];
  }
  return g;
}

// Rex.ReceiveObservation (rex.py:726-733): the true observation goes to the front of the history ring.  `owner`: this
// lane writes the words of its leg's motors; `live` (one lane per env): the arm's, the base quaternion and angular velocity.
template <int NM, int NL, bool ARM>
__device__ __forceinline__ void receive_observation(const DevCfg& c, EnvState& e, int i, bool live, bool owner, int leg0,
                                                    const MotorSide<NL, ARM>& ms) {
  if (!c.hist) return;
  const int head = ((int)(e.hist & 0xFFu) + 1) % REX_HISTORY_LEN;
  const int len = min((int)((e.hist >> 8) & 0xFFu) + 1, REX_HISTORY_LEN);
  e.hist = (uint32_t)head | ((uint32_t)len << 8);
  if (owner) {
#pragma unroll
    for (int jl = 0; jl < 3 * NL; ++jl) {
      const int j = MotorSide<NL, ARM>::motor(leg0, jl);
      hist_at(c, i, head, j) = legv<NL>(e.ph.q, leg0, jl); hist_at(c, i, head, NM + j) = legv<NL>(e.ph.qd, leg0, jl);
      hist_at(c, i, head, 2 * NM + j) = ms.tau_obs[jl];
    }
  }
  if (live) {
#pragma unroll
    for (int a = 0; a < (ARM ? 6 : 0); ++a) {
      hist_at(c, i, head, 12 + a) = e.ph.q[12 + a]; hist_at(c, i, head, NM + 12 + a) = e.ph.qd[12 + a];
      hist_at(c, i, head, 2 * NM + 12 + a) = ms.tau_obs[3 * NL + a];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) hist_at(c, i, head, 3 * NM + k) = e.ph.quat[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) hist_at(c, i, head, 3 * NM + 4 + k) = e.ph.ang[k];
  }
  if (NL == 1) mirror_sync();   // the ring entry has several writers; its readers are other lanes of the group
}

// Rex.ApplyAction + stepSimulation + ReceiveObservation (rex.py:158-163, 568-641) for the motors this lane carries.
template <bool LANECAP, bool TRACE, int NL, bool ARM, class SM, class ARMP>
__device__ __forceinline__ void rex_substep(const DevCfg& c, EnvState& e, int i, bool live, bool owner, int leg0, MotorSide<NL, ARM>& ms,
                                            const SM& sm, const Ground& ground, ARMP& armp) {
  constexpr int NM = ARMP::NM, N = MotorSide<NL, ARM>::N;
  float tau[N];
  const float limit = 1.0f / c.dt;  // OVERHEAT_SHUTDOWN_TIME / time_step, rex.py:607
  int s0 = 0, s1 = 0;
  float alpha = 0.0f;
  if (c.hist) delay_slots(e.hist, c.pd_latency, c.pd_slots, c.pd_alpha, s0, s1, alpha);   // what the PD loop sees: _GetPDObservation, rex.py:755-759
#pragma unroll
  for (int jl = 0; jl < N; ++jl) {
    const int j = MotorSide<NL, ARM>::motor(leg0, jl);
    const float qt = jl < 3 * NL ? legv<NL>(e.ph.q, leg0, jl) : e.ph.q[12 + (jl - 3 * NL)];
    const float qdt = jl < 3 * NL ? legv<NL>(e.ph.qd, leg0, jl) : e.ph.qd[12 + (jl - 3 * NL)];
    float qo = qt, qdo = qdt;
    if (c.hist) { qo = delayed_word(c, i, s0, s1, alpha, j); qdo = delayed_word(c, i, s0, s1, alpha, NM + j); }
    float act, obs;
    motor_torque(ms.cmd[jl], qo, qdo, qdt, c.kp, c.kd, act, obs);
    uint32_t cnt = ms.heat(jl);
    cnt = fabsf(act) > 2.45f ? min(cnt + 1u, 65535u) : 0u;                      // rex.py:603-606
    if ((float)cnt > limit) e.motor_en &= ~(1u << j);                           // rex.py:607-608 (lane groups: the lane's own bits;
    ms.set_heat(jl, cnt);                                                       //  the masks are merged before the state is stored)
    ms.tau_obs[jl] = obs;
    tau[jl] = ((e.motor_en >> j) & 1u) ? act : 0.0f;                            // rex.py:617-623
  }
  // (the env's words that only sit through the physics: out of the way of the sweep loop, see to_agpr)
  constexpr bool kHold = REX_HOLD_ACROSS_SWEEPS(SM::kEpw, ARM, SM::kBody, LANECAP) && NL == 1;
  if constexpr (kHold) {
    hold(e.phi); hold(e.alpha); hold(e.target); hold(e.aux); hold(e.last_step); hold(e.end_step); hold(e.flags); hold(e.steps);
    hold(e.episode); hold(e.motor_en); hold(e.hist); hold(ms.cmd); hold(ms.tau_obs); hold(ms.overheat2);
  }
  physics_substep<LANECAP, TRACE>(e.ph, tau, c.dt, c.max_iterations, c.iterations, c.res_thr, sm, ground, armp, e.sweeps, c.trace, c.n, i, live);
  if constexpr (kHold) {
    take(e.phi); take(e.alpha); take(e.target); take(e.aux); take(e.last_step); take(e.end_step); take(e.flags); take(e.steps);
    take(e.episode); take(e.motor_en); take(e.hist); take(ms.cmd); take(ms.tau_obs); take(ms.overheat2);
  }
  receive_observation<NM>(c, e, i, live, owner, leg0, ms);
}

// Rex._control_observation as the env-level getters see it (delayed by control_latency when the model is on), for the
// motors this lane carries
template <int NM, int NL, bool ARM>
__device__ __forceinline__ void control_observation(const DevCfg& c, const EnvState& e, int i, int leg0, const MotorSide<NL, ARM>& ms,
                                                    CtrlObs<MotorSide<NL, ARM>::N>& o) {
  constexpr int N = MotorSide<NL, ARM>::N;
  if (c.hist) {
    int s0, s1; float alpha;
    delay_slots(e.hist, c.control_latency, c.control_slots, c.control_alpha, s0, s1, alpha);
#pragma unroll
    for (int jl = 0; jl < N; ++jl) {
      const int j = MotorSide<NL, ARM>::motor(leg0, jl);
      o.q[jl] = delayed_word(c, i, s0, s1, alpha, j); o.qd[jl] = delayed_word(c, i, s0, s1, alpha, NM + j);
      o.tau[jl] = delayed_word(c, i, s0, s1, alpha, 2 * NM + j);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.quat[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) o.w[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + 4 + k);
  } else {
#pragma unroll
    for (int jl = 0; jl < N; ++jl) {
      o.q[jl] = jl < 3 * NL ? legv<NL>(e.ph.q, leg0, jl) : e.ph.q[12 + (jl - 3 * NL)];
      o.qd[jl] = jl < 3 * NL ? legv<NL>(e.ph.qd, leg0, jl) : e.ph.qd[12 + (jl - 3 * NL)];
      o.tau[jl] = ms.tau_obs[jl];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o.quat[k] = e.ph.quat[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o.w[k] = e.ph.ang[k];
  }
}

// RangeNormalize of the observation (wrappers.py:236-240); bounds are symmetric (rex_gym_env.py:277-278)
__device__ __forceinline__ float normalize_obs1(const DevCfg& c, int k, float v) {
  const float hi = (k == 2 || k == 3) ? c.obs_hi_rate : c.obs_hi_ang, lo = -hi;
  return 2.0f * (v - lo) / (hi - lo) - 1.0f;
}
// one standard normal draw of a getter call site's motor j (blocks of four motors, see gauss4)
__device__ __forceinline__ float gauss_motor(const DevCfg& c, int gidx, int episode, int step, int site, int j) {
  float z[4];
  gauss4(c.seed_lo, c.seed_hi, gidx, episode, step, site + (j >> 2), z);
  const int k = j & 3;
  return k == 0 ? z[0] : (k == 1 ? z[1] : (k == 2 ? z[2] : z[3]));
}

// _get_observation: obs[0..3] = roll, pitch, roll rate, pitch rate (walk_env.py:356-362); the gallop env appends the motor
// angles (gallop_env.py:349-356) -- `ang` receives those of the motors this lane carries (MotorSide's local order)
template <int NM, int N>
__device__ __forceinline__ void env_observation(const DevCfg& c, const CtrlObs<N>& co, int leg0, float* obs, float* ang,
                                                int gidx = 0, int episode = 0, int step = 0) {
  constexpr int NLM = NM == 18 ? N - 6 : N;
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
    const bool noisy = c.noise_on && c.noise[0] > 0.0f;                           // GetMotorAngles: noise, then MapToMinusPiToPi (rex.py:457-468)
#pragma unroll
    for (int jl = 0; jl < N; ++jl) {                                              // MapToMinusPiToPi, rex.py:26-41
      const int j = jl < NLM ? (NLM == 12 ? jl : 3 * leg0 + jl) : 12 + (jl - NLM);
      float a = fmodf(co.q[jl] + (noisy ? c.noise[0] * gauss_motor(c, gidx, episode, step, kNzAngle, j) : 0.0f), 2.0f * kPi);
      if (a >= kPi) a -= 2.0f * kPi; else if (a < -kPi) a += 2.0f * kPi;
      ang[jl] = a;
    }
  } else {
#pragma unroll
    for (int jl = 0; jl < N; ++jl) ang[jl] = 0.0f;    // a task with a narrower observation leaves the tail of a mixed batch's row 0
  }
}

// RexWalkEnv.reset / RexReactiveEnv.reset ... draws on top of the settled snapshot.  `seen` receives what the robot last
// observed of its base (quaternion, angular velocity): the settled one -- the turn env teleports the base behind the
// observation's back (turn_env.py:158-160), and reset() returns that older reading.
template <int NM, int NL = 4>
__device__ __forceinline__ void env_reset(const DevCfg& c, const float* snap, int i, bool live, int gidx, EnvState& e, float* seen, int leg0 = 0) {
  const int32_t episode = e.episode + 1;
  const float alpha = e.alpha;   // the env keeps one GaitPlanner for life: its arc angle survives reset() (gait_planner.py:76-85)
  const int nrec = (c.n_terrain > 0 ? c.n_terrain : 1) * c.n_mix;
  const int rec = (c.n_terrain > 0 ? terrain_index(c, gidx, episode) : 0) * c.n_mix + (c.n_mix > 1 ? mix_slot(c, c.task) : 0);
  load_env<NM, NL>(snap, nrec, rec, e, leg0);   // settled on this episode's terrain (under this env's task)
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
  e.phi = 0.0f; e.last_step = 0; e.alpha = alpha;
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
  e.end_step = 0; e.aux = 0.0f; e.steps = 0;
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

// the env clock `step_counter * time_step` (rex.py:155-156) at env step k, as the double the reference holds
__device__ __forceinline__ double step_time(const DevCfg& c, int k) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  return (double)(k * c.action_repeat) * c.dt_d;
}
// What a task's command code asks of the shared controller tail (ONE call site per kernel: the planner + IK code is the
// bulk of the controller, and every copy of it lengthens a kernel that is already longer than the reach of a short branch).
// kind 0: the joint targets are already written; 1: GaitPlanner.loop then Kinematics.solve; 2: Kinematics.solve on the
// default stance frames (RexPosesEnv).
struct LegCall {
  int kind, mode;
  float v, angle, w_rot, direction;
  double T;
  float pos[3], orn[3];
};
// GaitPlanner.loop + Kinematics.solve on the env's planner state, for the legs this lane carries: the 3 NL joint targets in
// the motor order (the planner's leg order FR, FL, RR, RL is the motor order FL, FR, RL, RR with the sides swapped:
// walk_env.py:284-289).  _last_time is the (wall) clock of the env step that latched it, and `_phi >= 0.99` was decided
// in double when _phi was computed.
template <int NL>
__device__ __forceinline__ void env_gait_ik(const DevCfg& c, EnvState& e, int leg0, const LegCall& k, float* cmd) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  if (k.kind == 0) return;
  const bool planner = k.kind == 1;
  const bool wrap = planner && (e.flags & REX_F_PHASE_WRAP) != 0;
  GaitState g{wrap ? 1.0 : 0.0, step_time(c, e.last_step) * c.gait_clock_d, e.alpha};
  if (wrap) e.last_step = e.steps;
  const double now = step_time(c, e.steps) * c.gait_clock_d;
  if constexpr (NL == 4) {
    float frames[12], ang[12];
    if (planner) gait_loop(g, k.mode, k.v, k.angle, k.w_rot, k.T, k.direction, now, frames);
    else {
#pragma unroll
      for (int l = 0; l < 4; ++l) { frames[3 * l] = gait_bx0(l); frames[3 * l + 1] = gait_by0(l); frames[3 * l + 2] = -kIkHeight; }
    }
    ik_solve(k.orn, k.pos, frames, ang);
#pragma unroll
    for (int j = 0; j < 3; ++j) { cmd[j] = ang[3 + j]; cmd[3 + j] = ang[j]; cmd[6 + j] = ang[9 + j]; cmd[9 + j] = ang[6 + j]; }
  } else {
    const int own = leg0 ^ 1;
    // the arc angle runs through the legs only while the gait turns (or a turn has left it non-zero): decided for the wave
    const bool chain = __builtin_amdgcn_ballot_w64(planner && (k.w_rot != 0.0f || g.alpha != 0.0f)) != 0;
    float frame[3] = {gait_bx0(own), gait_by0(own), -kIkHeight};
    if (planner) gait_loop_leg(g, k.mode, k.v, k.angle, k.w_rot, k.T, k.direction, now, own, chain, frame);
    ik_solve_leg(k.orn, k.pos, frame, own, cmd);
  }
  if (planner) {
    e.phi = (float)g.phi; e.alpha = g.alpha;
    e.flags = g.phi >= 0.99 ? (e.flags | REX_F_PHASE_WRAP) : (e.flags & ~REX_F_PHASE_WRAP);
  }
}
// joint j (0..11, motor order) of local index jl
template <int NL> __device__ __forceinline__ int leg_joint(int leg0, int jl) { return NL == 4 ? jl : 3 * leg0 + jl; }

// RexWalkEnv._transform_action_to_motor_command (walk_env.py:207-324)
template <int NL>
__device__ __forceinline__ void walk_command(const DevCfg& c, EnvState& e, const float* action, int leg0, float* cmd, LegCall& call) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  if (e.flags & REX_F_STAY_STILL) {
#pragma unroll
    for (int jl = 0; jl < 3 * NL; ++jl) cmd[jl] = init_pose(c, leg_joint<NL>(leg0, jl));
    return;
  }
  const double t = step_time(c, e.steps);                                        // rex.py:155-156
  if (e.target != 0.0f && fabsf(e.ph.pos[0]) >= fabsf(e.target) - 0.15f) {       // walk_env.py:207-215
    e.flags |= REX_F_GOAL_REACHED;
    if (!(e.flags & REX_F_TERMINATING)) { e.end_step = e.steps; e.flags |= REX_F_TERMINATING; }
  }
  const double end_t = step_time(c, e.end_step);
  const bool backwards = (e.flags & REX_F_BACKWARDS) != 0;
  if (c.signal == REX_SIGNAL_IK) {                                               // walk_env.py:252-290
    const double p = 0.8 + (double)action[0];
    const float gait_coeff = (0.0 <= t && t <= p) ? (float)t : 1.0f;
    const double period = backwards ? 0.5 : 0.65;
    float step_length = (backwards ? -0.3f : 0.6f) * gait_coeff;
    if (e.flags & REX_F_GOAL_REACHED) {
      const double pb = 0.8 + (double)action[1];
      const float b = (end_t <= t && t <= pb + end_t) ? (float)(1.0 - (t - end_t)) : 0.0f;
      step_length *= b;
      if (b == 0.0f) e.flags |= REX_F_STAY_STILL;
    }
    call = LegCall{1, 0, step_length, 0.0f, 0.0f, step_length < 0.0f ? -1.0f : 1.0f, period, {backwards ? 0.0f : 0.01f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  } else {                                                                       // walk_env.py:292-315
    float l_a = 0.1f, f_a = 0.2f;
    if (e.flags & REX_F_GOAL_REACHED) {
      const bool inside = end_t <= t && t <= 0.8 + end_t;
      const float b = inside ? (float)(1.0 - (t - end_t)) : 0.0f;
      l_a *= b; f_a *= b;
      // `if coeff is 0.0` (walk_env.py:300) is an identity test: true exactly when the brake function returns its
      // end_value argument, i.e. outside the brake window (pinned by tests/golden/env_command_golden.json)
      if (!inside) e.flags |= REX_F_STAY_STILL;
    }
    const float sc = (0.0 <= t && t <= 0.8) ? (float)t : 1.0f;
    l_a *= sc; f_a *= sc;
    float sph, cph;
    sincos_fast(2.0f * kPi / 0.125f * (float)t, sph, cph);
    const float le = l_a * cph, fe = f_a * cph;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int m = NL == 4 ? l : leg0;                       // legs 0 and 3 swing against legs 1 and 2; action pair (2m, 2m + 1)
      const float sg = (m == 0 || m == 3) ? 1.0f : -1.0f;
      const float a0 = m == 0 ? action[0] : (m == 1 ? action[2] : (m == 2 ? action[4] : action[6]));
      const float a1 = m == 0 ? action[1] : (m == 1 ? action[3] : (m == 2 ? action[5] : action[7]));
      cmd[3 * l] = pose_stand_ol(3 * m);
      cmd[3 * l + 1] = pose_stand_ol(3 * m + 1) + (sg * le + a0);
      cmd[3 * l + 2] = pose_stand_ol(3 * m + 2) + (sg * fe + a1);
    }
  }
}

// RexReactiveEnv._transform_action_to_motor_command (gallop_env.py:212-313)
template <int NL>
__device__ __forceinline__ void gallop_command(const DevCfg& c, EnvState& e, const float* action, int leg0, float* cmd, LegCall& call) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  if (e.flags & REX_F_STAY_STILL) {
#pragma unroll
    for (int jl = 0; jl < 3 * NL; ++jl) cmd[jl] = pose_stand(leg_joint<NL>(leg0, jl));   // rex.initial_pose
    return;
  }
  const double t = step_time(c, e.steps);
  if (e.target != 0.0f && fabsf(e.ph.pos[0]) >= fabsf(e.target)) {               // gallop_env.py:212-220
    e.flags |= REX_F_GOAL_REACHED;
    if (!(e.flags & REX_F_TERMINATING)) { e.end_step = e.steps; e.flags |= REX_F_TERMINATING; }
  }
  const double end_t = step_time(c, e.end_step);
  if (c.signal == REX_SIGNAL_IK) {                                               // gallop_env.py:257-285
    const double pg = 1.0 + (double)action[1];
    const float gait_coeff = (0.0 <= t && t <= pg) ? (float)t : 1.0f;
    float step_length = 1.3f * gait_coeff;
    if (e.flags & REX_F_GOAL_REACHED) {
      const double pb = 1.0 + (double)action[0];
      step_length *= (end_t <= t && t <= pb + end_t) ? (float)(1.0 - (t - end_t)) : 0.0f;
    }
    call = LegCall{1, 1, step_length, 0.0f, 0.0f, 1.0f, 0.3, {0.01f, 0.0f, -0.007f}, {0.0f, 0.0f, 0.0f}};
  } else {                                                                       // gallop_env.py:287-304
    float lp[4] = {action[0], action[1], action[2], action[3]};
    if (e.flags & REX_F_GOAL_REACHED) {
      const bool inside = end_t <= t && t <= 1.0 + end_t;
      const float b = inside ? (float)(1.0 - (t - end_t)) : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) lp[k] *= b;
      if (!inside) e.flags |= REX_F_STAY_STILL;   // gallop_env.py:291: `coeff is 0.0`, an identity test (see walk_command)
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int m = NL == 4 ? l : leg0;
      cmd[3 * l] = init_pose(c, 3 * m);
      cmd[3 * l + 1] = init_pose(c, 3 * m + 1) + (m < 2 ? lp[0] : lp[2]);
      cmd[3 * l + 2] = init_pose(c, 3 * m + 2) + (m < 2 ? lp[1] : lp[3]);
    }
  }
}

// RexPosesEnv._signal (poses_env.py:186-225)
__device__ __forceinline__ void poses_command(const DevCfg& c, EnvState& e, const float* action, LegCall& call) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  const double t = step_time(c, e.steps), p = 0.8 + (double)action[0];
  const float coeff = (0.0 <= t && t <= p) ? (float)t : 1.0f;
  const float staged = e.target * coeff;
  const int k = (int)e.aux;
  call = LegCall{2, 0, 0.0f, 0.0f, 0.0f, 1.0f, 1.0, {0.01f, k == 0 ? staged : 0.0f, k == 1 ? staged : 0.0f},
                 {k == 2 ? staged : 0.0f, k == 3 ? staged : 0.0f, k == 4 ? staged : 0.0f}};
}

// RexTurnEnv._transform_action_to_motor_command (turn_env.py:239-347)
template <int NL>
__device__ __forceinline__ void turn_command(const DevCfg& c, EnvState& e, const float* ctrl_quat, const float* action, int leg0, float* cmd, int gidx, LegCall& call) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  const double t = step_time(c, e.steps);
  if (e.flags & REX_F_STAY_STILL) {
    if (t - step_time(c, e.end_step) >= 1.0) e.flags |= REX_F_ENV_GOAL;          // _terminate_with_delay
#pragma unroll
    for (int jl = 0; jl < 3 * NL; ++jl) cmd[jl] = init_pose(c, leg_joint<NL>(leg0, jl));
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
      if (!(e.flags & REX_F_TERMINATING)) { e.end_step = e.steps; e.flags |= REX_F_TERMINATING; }
    }
  }
  const float diff = fabsf(e.aux - e.target);                                    // _solve_direction
  const bool clockwise = e.aux < e.target ? diff > 3.14f : diff < 3.14f;
  if (e.flags & REX_F_GOAL_REACHED) e.flags |= REX_F_STAY_STILL;
  if (c.signal == REX_SIGNAL_IK) {
    const float coeff = (0.0 <= t && t <= 0.8) ? (float)t : 1.0f;
    float dirv = -0.5f * coeff;
    if (clockwise) dirv = -dirv;
    call = LegCall{1, 0, 0.02f, 0.0f, dirv + action[0], 1.0f, 0.75 + (double)action[1], {0.009f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  } else {
    const float ext = 0.1f, swing = 0.03f + action[0], swipe = 0.05f + action[1];
    const int ith = ((int)(t / 0.1)) % 2;
    const float ms = clockwise ? swing : -swing;     // right_* = left_* with the swing sign flipped
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int m = NL == 4 ? l : leg0;
      // first pose:  ( swipe,  ext,  ms) (-swipe,  ext, -ms) ( swipe, -ext, -ms) (-swipe, -ext,  ms)
      // second pose: (-swipe,  0,   -ms) ( swipe,  0,    ms) (-swipe,  0,    ms) ( swipe,  0,   -ms)
      const float s0 = (m & 1) ? -1.0f : 1.0f, s1 = m < 2 ? 1.0f : -1.0f, s2 = (m == 0 || m == 3) ? 1.0f : -1.0f;
      cmd[3 * l] = pose_stand_ol(3 * m) + (ith ? -s0 * swipe : s0 * swipe);
      cmd[3 * l + 1] = pose_stand_ol(3 * m + 1) + (ith ? 0.0f : s1 * ext);
      cmd[3 * l + 2] = pose_stand_ol(3 * m + 2) + (ith ? -s2 * ms : s2 * ms);
    }
  }
}

// RexStandupEnv._signal (standup_env.py:113-120): the 'stand' pose, scaled by a 'brake' overshoot for the first 0.1 s
template <int NL>
__device__ __forceinline__ void standup_command(const DevCfg& c, const EnvState& e, const float* action, float* cmd) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  const double t = step_time(c, e.steps);                                  // GetTimeSinceReset, rex.py:155-156
  const float f = t > 0.1 ? 1.0f : (0.1f + action[0]) / ((float)t + 1.0f) + 1.5f;
  const float leg = -0.88643435f * f, foot = 1.30197369f * f;
#pragma unroll
  for (int l = 0; l < NL; ++l) { cmd[3 * l] = 0.0f; cmd[3 * l + 1] = leg; cmd[3 * l + 2] = foot; }
}

// The joint states and overheat counters of an env, in motor order, for the lane that stores the state.  Lane groups: every
// lane hands in its own leg's through LDS (rows 0..2 of the contact-row region, idle outside a substep; a substep rewrites
// them from scratch); one env per lane: already there.  Also fills e.overheat (all motors).
template <int NL, int LPE, bool ARM, class SM>
__device__ __forceinline__ void gather_legs(const SM& sm, int leg0, EnvState& e, const MotorSide<NL, ARM>& ms, float* q12, float* qd12) {
  if constexpr (NL == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      sm.rowf(0, 3 * leg0 + k) = e.ph.q[k]; sm.rowf(1, 3 * leg0 + k) = e.ph.qd[k];
      sm.rowf(2, 3 * leg0 + k) = __uint_as_float(ms.heat(k));
    }
    mirror_sync();
#pragma unroll
    for (int j = 0; j < 12; ++j) { q12[j] = sm.rowf(0, j); qd12[j] = sm.rowf(1, j); e.overheat[j] = __float_as_uint(sm.rowf(2, j)); }
  } else {
#pragma unroll
    for (int j = 0; j < 12; ++j) { q12[j] = e.ph.q[j]; qd12[j] = e.ph.qd[j]; e.overheat[j] = ms.heat(j); }
  }
#pragma unroll
  for (int a = 0; a < (ARM ? 6 : 0); ++a) e.overheat[12 + a] = ms.heat(3 * NL + a);
}

// ------------------------------------------------------------------------------------------
#ifndef REX_STEP_KERNEL_ATTR
#define REX_STEP_KERNEL_ATTR          /* developer experiments: e.g. -DREX_STEP_KERNEL_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))' */
#endif
#ifndef REX_FAST_EPW
#define REX_FAST_EPW 4
#endif
// waves per workgroup of the fused-actor kernels: four (one per SIMD of a CU) that share one copy of the actor's weights in LDS -- at 4 and 8
// envs per wave; at 16 the copy does not fit next to four waves' rows anyway (base 4 x 38 KB, arm 4 x 40 KB of 160), the weights are streamed,
// and one-wave workgroups keep a multi-round launch (> 16 384 envs) from waiting for the slowest of four waves before a CU takes new work
// (65 536 walk-IK envs, fused actor per step: 68.9 M env-steps/s with four-wave workgroups)
#define REX_POLICY_WAVES(EPW) ((EPW) <= 8 ? 4 : 1)
template <int EPW, bool ARM, bool MIXED, bool BODY, bool TRACE = false, bool SEG = false, bool POLICY = false>
__global__ __launch_bounds__(POLICY ? REX_WAVE * REX_POLICY_WAVES(EPW) : REX_WAVE) REX_STEP_KERNEL_ATTR void rex_step_kernel(DevCfg c, float* __restrict__ state, const float* __restrict__ snap,
                                                            const float* __restrict__ action0, float* __restrict__ obs_out0,
                                                            float* __restrict__ reward_out0, uint8_t* __restrict__ done_out0,
                                                            float* __restrict__ cmd_out0, typename PolArg<POLICY>::type pol) {
  // POLICY (the instantiations behind rex_step_policy / rex_step_segment_policy; the base and arm step units are compiled once more with
  // -DREX_TU_POL=1): a SEG kernel whose actions are not read from action0 but computed, step by step, by the reference's Gaussian MLP
  // actor on the observation the env returned last (rex_policy.h) -- a closed-loop rollout segment in one launch.  Their workgroup is
  // FOUR waves (one per SIMD of a CU, each with its own envs and its own LDS rows exactly as a one-wave workgroup has them) that share
  // one copy of the actor's weights in dynamic LDS, loaded once per launch, where it fits next to the rows (pol.in_lds; else the weights
  // are streamed from L2 every step).
  static_assert(!POLICY || (SEG && !MIXED && !BODY && !TRACE && EPW <= 16), "the fused actor: segment kernels of the single-task lane-group variants");
  // SEG (the instantiations behind rex_step_segment; every step translation unit is compiled once more with -DREX_TU_SEG=1): one launch =
  // c.nsteps consecutive env.step() calls of the shard, a rollout segment whose actions the caller already holds -- action0 /
  // obs_out0 / ... are then [nsteps][n][...] blocks.  Without SEG the loop below runs once and folds away: rex_step's kernels.  An env's state stays in its wave's
  // registers from step to step; a wave goes on to ITS envs' next step as soon as it has finished this one, so the launch ends with
  // the largest SUM of a wave's step times instead of paying the slowest wave of every step.
  // EPW envs share this wave (host picks it, rex_step): a small batch is spread over MORE, emptier waves because
  // idle SIMDs are free and a wave leaves the PGS sweep loop only when its slowest env has converged (and skips
  // only the legs no env of the wave has in contact), so fewer envs per wave means fewer sweeps and rows per
  // wave.  With EPW <= 16 every env owns a group of LPE = 8 (EPW <= 8) or 4 (EPW = 16) adjacent lanes (for EPW = 4 the
  // upper 32 lanes repeat the lower 32): the lanes of a group hold the same body state and split the work of an
  // env.step() by LEG -- the controller (Bezier + IK), the motor model, the observation words and the leg factorisation
  // of a substep are those of the lane's own leg (MotorSide; rex_device.h), the row work of a substep is split by row --
  // and lane 0 of the group stores the state.
  // MIXED (REX_TASK_MIXED): the envs of a batch run different tasks, the envs of a WAVE one (the host's slot map sorts them): c_
  // below is the wave's view of the config, its per-task constants wave-uniform.
  constexpr int NM = ARM ? 18 : 12;   // mark='arm': 6 more motors held at ARM_POSES['rest'] (rex_gym_env.py:347-353)
  constexpr int kLegF4 = REX_LEG_F4_OF(EPW, ARM, BODY);
  constexpr int kRowsF4 = ARM ? REX_LDS_F4_PER_ENV_ARM_OF(EPW) : REX_ROWS_F4_OF(kLegF4);
  static_assert(!BODY || EPW <= 16, "link-box contact rows: lane-group kernels only");
  constexpr int kWaveF4 = (kRowsF4 + (EPW <= 16 ? REX_PARK_F4_OF(EPW, ARM) : 0) + (BODY ? REX_BODY_F4 : 0)) * EPW;
  constexpr int kWaves = POLICY ? REX_POLICY_WAVES(EPW) : 1;     // waves of this workgroup (each works as a one-wave workgroup does)
  __shared__ float4 lds_wg[kWaveF4 * kWaves];
  float4* const lds = lds_wg + (kWaves > 1 ? (int)(threadIdx.x >> 6) * kWaveF4 : 0);     // this wave's rows
  REX_STAMP(t_kernel);
#ifdef REX_PROF
  const long long t_wall = (long long)wall_clock64();
#endif
  const int lane = kWaves > 1 ? (int)(threadIdx.x & (REX_WAVE - 1)) : (int)threadIdx.x;
  const int wg_block = kWaves > 1 ? (int)blockIdx.x * kWaves + (int)(threadIdx.x >> 6) : (int)blockIdx.x;   // the one-wave block this wave stands for
  if (c.clock && lane == 0) atomicMin(&c.clock[2 * (wg_block & (REX_CLOCK_WAYS - 1))], (unsigned long long)wall_clock64());
  const float* pol_wl = nullptr;          // POLICY: the actor's weights in LDS (null: streamed)
  if constexpr (POLICY) {
    extern __shared__ float4 rex_dyn_lds[];
    if (pol.in_lds) {
      policy_weights_to_lds(pol, policy_offsets(c.obs_dim, c.action_dim, pol.h1, pol.h2).total, reinterpret_cast<float*>(rex_dyn_lds), (int)threadIdx.x,
                            REX_WAVE * kWaves);
      pol_wl = reinterpret_cast<const float*>(rex_dyn_lds);
    }
    __syncthreads();
  }
  constexpr int LPE = EPW < 64 ? lanes_per_env(EPW) : 1;     // EPW <= 16: lane = LPE * slot + p (rex_device.h, group layout)
  constexpr int NL = EPW < 64 ? 1 : 4;                       // legs whose controller / motors a lane carries
  using MS = MotorSide<NL, ARM>;
  const int slot = (lane / LPE) & (EPW - 1);
  const int pl = lane & (LPE - 1);                           // lane of the group
  const int leg0 = NL == 4 ? 0 : (LPE == 8 ? pl >> 1 : pl);  // its leg (motor order FL, FR, RL, RR)
  // Block b runs on XCD b % 8 (observed placement; speed only).  A 64-byte sector of a state word holds 16 envs = 16 / EPW
  // blocks' worth: hand the blocks of one sector to the same XCD, so that one L2 fetches (and writes back) the sector
  // instead of 16 / EPW of them.  A bijection on the full groups of 8 x (16 / EPW) blocks; the tail keeps its order.
  int blk = wg_block;
  int gi, i;
  bool ingrid;
  DevCfg cmix;                          // MIXED only
  if constexpr (MIXED) {
    // the task-sorted slot map (host: task_slot_map / task_region_map in rexsim.hip) places the envs: a wave holds envs of one task out of one chunk of
    // neighbouring envs, and the chunks' workgroups are dealt to the XCDs by the map itself
    const int first = c.slot_env[blk * EPW];                 // wave-uniform (scalar load)
    if (first < 0) return;                                   // a padding workgroup of the map
    gi = c.slot_env[blk * EPW + slot];
    ingrid = lane < LPE * EPW && gi >= 0;
    i = gi >= 0 ? gi : first;                                // padding slots shadow the wave's first env (keeps the wave convergent)
    mixed_config_of_task(c, c.block_task[blk], cmix);        // wave-uniform: the task's constants stay in SGPRs
    cmix.max_repeat = cmix.action_repeat; cmix.max_iterations = cmix.iterations;
  } else {
    if constexpr (EPW < 16 && !POLICY) {   // (a four-wave workgroup covers whole sectors by itself)
      constexpr int G = 16 / EPW;
      const int full = ((int)gridDim.x / (8 * G)) * (8 * G);
      if (blk < full) { const int xcd = blk & 7, q = blk >> 3; blk = ((q / G) * 8 + xcd) * G + (q % G); }
    }
    gi = blk * EPW + slot;
    ingrid = lane < LPE * EPW && gi < c.n;
    const int gj = gi < c.n ? gi : c.n - 1;   // tail slots shadow the last env (keeps the wave convergent)
    i = c.perm ? c.perm[gj] : gj;             // regrouped batches: the env this slot works on
  }
  const bool live = ingrid && pl == 0;                       // the lane that stores the env's state
  const bool owner = ingrid && (LPE != 8 || (pl & 1) == 0);  // the lane that stores its leg's words (8 lanes per env: two carry a leg)
  const Lds<EPW, kLegF4, BODY> sm{lds, slot, EPW <= 16 ? lds + kRowsF4 * EPW : nullptr, BODY ? lds + (kRowsF4 + REX_PARK_F4_OF(EPW, ARM)) * EPW : nullptr};
  typename ArmHook<EPW, ARM>::type armp = ArmHook<EPW, ARM>::make(lds, slot);
  const DevCfg& c_ = MIXED ? cmix : c;

  EnvState e;
  load_env<NM, NL>(state, c.n, i, e, leg0);
  const int i_env = i, leg0_env = leg0;
  int seg_sweeps = 0;     // SEG: the env's solver sweeps summed over the steps of the segment so far
#pragma clang loop unroll(disable)
  for (int seg_step = 0; seg_step < (SEG ? c.nsteps : 1); ++seg_step) {
  // SEG: an opaque copy of the env index per step: otherwise every address the body forms from it (state words, history ring, output
  // rows: ~100 registers' worth) is loop-invariant, hoisted in front of the loop and carried through the solver sweeps
  int i = i_env, leg0 = leg0_env;
  if constexpr (SEG) asm volatile("" : "+v"(i), "+v"(leg0));   // (and of the lane's leg: the per-leg constants of the controller and of the leg pass)
  // this step's slices of the caller's blocks (32-bit element offsets: rex_step_segment checks nsteps * n * width < 2^31)
  const float* __restrict__ action = action0 + (unsigned)(seg_step * c.n * c.action_dim);
  float* __restrict__ obs_out = obs_out0 + (unsigned)(seg_step * c.n * c.obs_dim);
  float* __restrict__ reward_out = reward_out0 + (unsigned)(seg_step * c.n);
  uint8_t* __restrict__ done_out = done_out0 + (unsigned)(seg_step * c.n);
  float* __restrict__ cmd_out = cmd_out0 ? cmd_out0 + (unsigned)(seg_step * c.n * NM) : nullptr;
  REX_STAMP(t_step);
  e.sweeps = 0;
  MS ms;
  take_overheat(e, leg0, ms);      // (a later step of a segment: e.overheat holds all counters again -- gather_legs)
  const uint32_t motor_en0 = e.motor_en;
  float act[8];
  if constexpr (POLICY) {
    // algo.perform(prevob) (agents/ppo/algorithm.py:105-134): the observation of the previous step -- slice seg_step - 1 of the
    // segment's observation block, stored by this wave in front of the fence that ended that step; the caller's obs_in for the first
    const float* obs_prev = seg_step == 0 ? pol.obs_in : obs_out0 + (unsigned)((seg_step - 1) * c.n * c.obs_dim);
    if (pol_wl) policy_act<EPW, LPE, ARM, true>(c, pol, pol_wl, reinterpret_cast<float*>(lds), lane, slot, pl, leg0, i, ingrid, e.episode, e.steps, obs_prev,
                                                (unsigned)(seg_step * c.n * c.action_dim), act);
    else policy_act<EPW, LPE, ARM, false>(c, pol, nullptr, reinterpret_cast<float*>(lds), lane, slot, pl, leg0, i, ingrid, e.episode, e.steps, obs_prev,
                                          (unsigned)(seg_step * c.n * c.action_dim), act);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float a = POLICY ? act[k] : (k < c.action_dim ? action[(size_t)i * c.action_dim + k] : 0.0f);
    if (c.range_normalize) {                       // ClipAction + RangeNormalize (wrappers.py:229-234,261-265)
      a = fminf(fmaxf(a, -1.0f), 1.0f);
      a = (a + 1.0f) / 2.0f * (c_.act_hi - c_.act_lo) + c_.act_lo;
    }
    act[k] = a;
  }

#pragma unroll
  for (int a = 0; a < MS::NA; ++a) ms.cmd[3 * NL + a] = (float)REXA_REST[a];
  LegCall call{0, 0, 0.0f, 0.0f, 0.0f, 1.0f, 1.0, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
  if (c_.task == REX_TASK_GALLOP) gallop_command<NL>(c_, e, act, leg0, ms.cmd, call);
  else if (c_.task == REX_TASK_TURN) {
    float cq[4] = {e.ph.quat[0], e.ph.quat[1], e.ph.quat[2], e.ph.quat[3]};
    if (c.hist) {
      int s0, s1; float alpha;
      delay_slots(e.hist, c.control_latency, c.control_slots, c.control_alpha, s0, s1, alpha);
#pragma unroll
      for (int k = 0; k < 4; ++k) cq[k] = delayed_word(c, i, s0, s1, alpha, 3 * NM + k);
    }
    turn_command<NL>(c_, e, cq, act, leg0, ms.cmd, c.env_index_base + i, call);
  }
  else if (c_.task == REX_TASK_POSES) poses_command(c_, e, act, call);
  else if (c_.task == REX_TASK_STANDUP) standup_command<NL>(c_, e, act, ms.cmd);
  else walk_command<NL>(c_, e, act, leg0, ms.cmd, call);
  env_gait_ik<NL>(c_, e, leg0, call, ms.cmd);     // the planner + IK tail of the task's command, if it has one
  if constexpr (TRACE) {   // rex_set_event_trace: the controller's discrete decisions of this step (goal / brake / hold flags, gait latches)
    if (live) {
      const unsigned w = trace_mix(e.flags, (unsigned)e.last_step * 65537u + (unsigned)e.end_step);
      c.trace[i] = trace_mix(c.trace[i], w);
      c.trace[2 * c.n + i] = trace_mix(c.trace[2 * c.n + i], w);
    }
  }

  REX_STAMP(t_command);
  const Ground ground = env_ground(c, i, c.env_index_base + i, e.episode);
  const int step0 = e.steps, episode0 = e.episode;   // keys of this step's sensor-noise draws

  // everything of env.step() after Rex.Step: reward, termination, in-launch reset, observation, stores
  auto epilogue = [&](bool commit, bool own) {
  // ---- the motor bookkeeping of the group's lanes meets again: the enable mask (the counters and joint states follow at the store) ----
  if constexpr (NL == 1) {
    const uint32_t cleared = leg_or<LPE>(motor_en0 & ~e.motor_en);     // every lane switched off only its own motors
    e.motor_en = motor_en0 & ~cleared;
  }
  // ---- reward (rex_gym_env.py:501-542) ----
  CtrlObs<MS::N> co;
  control_observation<NM>(c, e, i, leg0, ms, co);
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
#pragma unroll
    for (int jl = 0; jl < MS::N; ++jl) {   // GetMotorTorques / Velocities
      const int j = MS::motor(leg0, jl);
      co.tau[jl] += c.noise[2] * gauss_motor(c, gx, episode0, step0, kNzTorque, j);
      co.qd[jl] += c.noise[1] * gauss_motor(c, gx, episode0, step0, kNzVelocity, j);
    }
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
  fwd = fminf(fwd, c.fwd_cap);      // rex_gym_env.py:525
  const float drift = -fabsf(e.ph.pos[1]);
  const float shake = -fabsf(r20 + r21);
  float dp = 0.0f;                  // GetMotorTorques . GetMotorVelocities: the lane's legs, summed over the group, plus the arm
#pragma unroll
  for (int jl = 0; jl < 3 * NL; ++jl) dp += co.tau[jl] * co.qd[jl];
  if constexpr (NL == 1) dp = leg_sum<LPE>(dp);
#pragma unroll
  for (int a = 0; a < MS::NA; ++a) dp += co.tau[3 * NL + a] * co.qd[3 * NL + a];
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
    env_reset<NM, NL>(c_, snap, i, commit, c.env_index_base + i, e, seen, leg0);
    take_overheat(e, leg0, ms);
#pragma unroll
    for (int jl = 0; jl < MS::N; ++jl) ms.tau_obs[jl] = 0.0f;
    control_observation<NM>(c, e, i, leg0, ms, co);
    if (!c.hist) {
#pragma unroll
      for (int k = 0; k < 4; ++k) co.quat[k] = seen[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) co.w[k] = seen[4 + k];
    }
  }

  float obs[4], ang[MS::N];
  env_observation<NM>(c_, co, leg0, obs, ang, c.env_index_base + i, episode0, step0);
  // an opaque copy of the env index: the store addresses are rebuilt here instead of 54 address pairs being carried
  // (in AGPRs and scratch) from load_env across the whole kernel
  int is = i;
  asm volatile("" : "+v"(is));
  float q12[12], qd12[12];
  gather_legs<NL, LPE>(sm, leg0, e, ms, q12, qd12);
  if (commit) {
    store_env<NM>(state, c.n, is, e, q12, qd12);
#pragma unroll
    for (int k = 0; k < 4; ++k) obs_out[(size_t)is * c.obs_dim + k] = c.range_normalize ? normalize_obs1(c, k, obs[k]) : obs[k];
    reward_out[is] = reward;
    done_out[is] = done ? 1 : 0;
    if (c.sweeps) {
      // (a segment: the mean over its steps so far -- what the host sorts the envs by between launches; the last step's store stands)
      if constexpr (SEG) { seg_sweeps += e.sweeps; c.sweeps[is] = (seg_sweeps + (seg_step >> 1)) / (seg_step + 1); }
      else c.sweeps[is] = e.sweeps;
    }
  }
  // the words of the lane's motors: info['action'] and -- where the row carries motor angles (gallop; a mixed batch with
  // gallop in it: the other tasks leave them 0) -- its part of the observation row
  const bool wide = c.obs_dim > 4;
#pragma unroll
  for (int jl = 0; jl < MS::N; ++jl) {
    if (jl < 3 * NL ? own : commit) {
      const int j = MS::motor(leg0, jl);
      if (cmd_out) cmd_out[(size_t)is * NM + j] = ms.cmd[jl];
      if (wide) obs_out[(size_t)is * c.obs_dim + 4 + j] = c.range_normalize && c_.task == REX_TASK_GALLOP ? normalize_obs1(c, 4 + j, ang[jl]) : ang[jl];
    }
  }
  };

  // Rex.Step (a REX_TASK_MIXED wave runs one task: c_ is its wave-uniform view of the config)
  for (int k = 0; k < c_.action_repeat; ++k) rex_substep<false, TRACE>(c_, e, i, live, owner, leg0, ms, sm, ground, armp);
  REX_STAMP(t_substeps);
  epilogue(live, owner);
#ifdef REX_PROF
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    long long* p2 = g_prof2 + 16 * blockIdx.x;
    p2[3] += t_command - t_step; p2[4] += t_substeps - t_command; p2[5] += clock64() - t_substeps;
  }
#endif
  if constexpr (SEG) mirror_sync();   // the next step's LDS writes and history reads come behind this step's LDS reads and history stores
  }
  if (c.clock && lane == 0) atomicMax(&c.clock[2 * (wg_block & (REX_CLOCK_WAYS - 1)) + 1], (unsigned long long)wall_clock64());
#ifdef REX_PROF
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_prof[10 * blockIdx.x + 8] += clock64() - t_kernel; g_prof[10 * blockIdx.x + 9] += 1;
    g_prof2[16 * blockIdx.x + 6] += (long long)wall_clock64() - t_wall;   // the 100 MHz counter over the same span: calibrates clock64()
    g_prof2[16 * blockIdx.x + 7] = t_wall;                                // when this block of the LAST launch started
  }
#endif
}

// The reset motion of Rex.Reset (rex.py:296-324).  Plane: ONE robot, lane 0 writes the snapshot.  Terrain pool:
// lane t settles on terrain t and writes snapshot record t (word-major [53][n_terrain]).
template <bool ARM, bool BODY>
__global__ __launch_bounds__(REX_WAVE) void rex_settle_kernel(DevCfg c, float* __restrict__ snap) {
  constexpr int NM = ARM ? 18 : 12;
  constexpr int EPW = (ARM || BODY) ? 16 : REX_WAVE;   // the arm rows / link-box rows do not fit 64 envs per workgroup in LDS
  constexpr int kLegF4 = REX_LEG_F4_OF(EPW, ARM, BODY);
  constexpr int kRowsF4 = ARM ? REX_LDS_F4_PER_ENV_ARM_OF(EPW) : REX_ROWS_F4_OF(kLegF4);
  __shared__ float4 lds[(kRowsF4 + (EPW <= 16 ? REX_PARK_F4_OF(EPW, ARM) : 0) + (BODY ? REX_BODY_F4 : 0)) * EPW];
  constexpr int LPE = EPW < 64 ? lanes_per_env(EPW) : 1;
  constexpr int NL = EPW < 64 ? 1 : 4;
  using MS = MotorSide<NL, ARM>;
  const int lane = (int)(threadIdx.x / LPE) & (EPW - 1);
  const int pl = (int)threadIdx.x & (LPE - 1);
  const int leg0 = NL == 4 ? 0 : (LPE == 8 ? pl >> 1 : pl);
  const Lds<EPW, kLegF4, BODY> sm{lds, lane, EPW <= 16 ? lds + kRowsF4 * EPW : nullptr, BODY ? lds + (kRowsF4 + REX_PARK_F4_OF(EPW, ARM)) * EPW : nullptr};
  typename ArmHook<EPW, ARM>::type armp = ArmHook<EPW, ARM>::make(lds, lane);
  const int nrec = (c.n_terrain > 0 ? c.n_terrain : 1) * c.n_mix;
  const int first = (int)blockIdx.x * EPW + lane;                       // the (terrain, task) record this lane group settles
  const int t = first < nrec ? first : nrec - 1;
  const bool keeps = first < nrec && pl == 0;                           // one lane of the group stores it ...
  const bool owner = first < nrec && (LPE != 8 || (pl & 1) == 0);       // ... and one lane per leg that leg's history words
  const int terr = t / c.n_mix, slot = t % c.n_mix;
  Ground ground{nullptr, 0u, 0.0f, 1.0f, 1.0f, kMu, c.geo, c.anchor};
  if (c.n_terrain > 0) { ground.h = c.terrain; ground.off = (unsigned)terr * (unsigned)c.hf_stride; ground.mid = c.terrain_mid[terr]; }
  EnvState e;
  memset(&e, 0, sizeof(e));
  e.ph.pos[2] = c.init_z;
  e.ph.quat[3] = 1.0f;
#pragma unroll
  for (int jl = 0; jl < 3 * NL; ++jl) e.ph.q[jl] = pose_stand(leg_joint<NL>(leg0, jl));       // ResetPose: INIT_POSES[pose_id = 'stand']
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
  cs.trace = nullptr;   // (the event trace belongs to the envs, not to the snapshot records)
  if (c.n_mix > 1) {   // the reset motion of this record's task: its own sweep cap (rex_gym_env.py:184) -- a per-lane cap, the
    // records of a wave belong to different tasks (rex_substep<true>: the wave sweeps to max_iterations, a lane to its own)
    const int task = slot == 0 ? c.mix_task[0] : (slot == 1 ? c.mix_task[1] : (slot == 2 ? c.mix_task[2] : (slot == 3 ? c.mix_task[3] : c.mix_task[4])));
    cs.iterations = 300 / task_action_repeat(task);
  }
  e.hist = (uint32_t)(REX_HISTORY_LEN - 1);
  MS ms;
#pragma unroll
  for (int jl = 0; jl < MS::N; ++jl) ms.tau_obs[jl] = 0.0f;
  ms.clear_heat();
  const uint32_t motor_en0 = e.motor_en;
  if (c.task != REX_TASK_POSES) {   // RexPosesEnv: base reset() with initial_motor_angles=None skips the motion (rex.py:308)
    receive_observation<NM>(cs, e, t, keeps, owner, leg0, ms);
#pragma unroll
    for (int a = 0; a < MS::NA; ++a) ms.cmd[3 * NL + a] = (float)REXA_REST[a];
#pragma unroll
    for (int jl = 0; jl < 3 * NL; ++jl) ms.cmd[jl] = pose_stand(leg_joint<NL>(leg0, jl));
    for (int k = 0; k < 100 + c.reset_substeps; ++k) {     // one call site: the substep is the bulk of the kernel's code
      if (k == 100) {                                      // rex.py:315-318 (100 substeps holding 'stand'), then :319-322
#pragma unroll
        for (int jl = 0; jl < 3 * NL; ++jl) ms.cmd[jl] = reset_pose(c, leg_joint<NL>(leg0, jl));
      }
      rex_substep<true, false>(cs, e, t, keeps, owner, leg0, ms, sm, ground, armp);
    }
  }
  receive_observation<NM>(cs, e, t, keeps, owner, leg0, ms);                                                   // rex.py:323
  if (!cs.hist) e.hist = 0u;
  if constexpr (NL == 1) {          // the group's lanes hand their motors' bookkeeping to the lane that stores the record
    const uint32_t cleared = leg_or<LPE>(motor_en0 & ~e.motor_en);
    e.motor_en = motor_en0 & ~cleared;
  }
  float q12[12], qd12[12];
  gather_legs<NL, LPE>(sm, leg0, e, ms, q12, qd12);
  if (keeps) store_env<NM>(snap, nrec, t, e, q12, qd12);
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
  e.alpha = state[(size_t)Lay<NM>::ALPHA * c.n + i];
  float seen[7];
  DevCfg cmix;
  if (c.task == REX_TASK_MIXED) mixed_config(c, c.env_index_base + i, cmix);
  const DevCfg& c_ = c.task == REX_TASK_MIXED ? cmix : c;
  env_reset<NM>(c_, snap, i, true, c.env_index_base + i, e, seen);
  store_env<NM>(state, c.n, i, e, e.ph.q, e.ph.qd);
  MotorSide<4, NM == 18> ms;
#pragma unroll
  for (int jl = 0; jl < MotorSide<4, NM == 18>::N; ++jl) ms.tau_obs[jl] = 0.0f;
  CtrlObs<MotorSide<4, NM == 18>::N> co;
  control_observation<NM>(c, e, i, 0, ms, co);
  if (!c.hist) {
    for (int k = 0; k < 4; ++k) co.quat[k] = seen[k];
    for (int k = 0; k < 3; ++k) co.w[k] = seen[4 + k];
  }
  float obs[4], ang[MotorSide<4, NM == 18>::N];
  env_observation<NM>(c_, co, 0, obs, ang, c.env_index_base + i, e.episode, -1);   // reset()'s own reading: its own noise draws
  if (!obs_out) return;
  for (int k = 0; k < 4; ++k) obs_out[(size_t)r * c.obs_dim + k] = c.range_normalize ? normalize_obs1(c, k, obs[k]) : obs[k];
  if (c.obs_dim > 4) {
    for (int j = 0; j < NM; ++j)
      obs_out[(size_t)r * c.obs_dim + 4 + j] = c.range_normalize && c_.task == REX_TASK_GALLOP ? normalize_obs1(c, 4 + j, ang[j]) : ang[j];
  }
}

}  // namespace rex

namespace rex { struct MixRegions { int32_t n_mix, bins_per_task, base[5]; }; }   // base[k]: first slot of task slot k's region of the slot map
// ---- host side shared by the translation units ----
#define REX_TIMING_RING 256     /* event pairs of rex_set_timing(2) */
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
  unsigned long long* d_clock;   // [REX_CLOCK_SLOTS][REX_CLOCK_WAYS][2] device-side (min start, max end) ticks, rex_set_timing(3)
  unsigned long long* h_clock;   // host staging of the same (per sim: no buffer is shared between sims or threads)
  int32_t* d_perm;   // regrouping (large batches only): wave slot -> env, and the per-env sweep counts it is sorted by
  int32_t* d_sweeps;
  int32_t* d_regroup;   // [chunks][64] bin counts / start offsets of the many-workgroup sort + its completion counter
  hipEvent_t ring0[REX_TIMING_RING], ring1[REX_TIMING_RING];
  long long timed_steps;
  int words;   // per-env state words of the config's mark
  int32_t* d_slot_env;   // REX_TASK_MIXED: the task-sorted slot map (DevCfg::slot_env / block_task) and its workgroup count
  int32_t* d_block_task;
  int mixed_blocks;
  int32_t* d_class;      // a regrouped mixed batch: task slot of every env (rexsim.hip, rex_regroup_mixed_*), and its regions
  rex::MixRegions mix_regions;
  rex::PolDev pol;       // rex_set_policy: the actor of rex_step_policy / rex_step_segment_policy
  float* d_polbuf;       // the packed actor (library-owned; rex_policy.h policy_offsets) and its capacity in floats
  int polbuf_floats;
  int have_policy;
  bool use_policy;        // this launch runs the fused-actor kernels (set by step_launch)
  int pol_attr_bytes;     // the dynamic-LDS limit this sim has already set on its fused-actor kernel (hipFuncSetAttribute)
  int pol_lds_bytes;      // dynamic LDS of the fused-actor kernels: the weights' copy (0: they do not fit next to four waves' rows and are streamed)
};

// launchers, one per variant group (each in its own translation unit)
void rex_launch_step_base(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_arm(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_base(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_arm(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_body(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_base_trace(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_arm_trace(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_base_trace(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_arm_trace(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_body_trace(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_base_seg(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_arm_seg(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_base_seg(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_mixed_arm_seg(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_body_seg(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_base_pol(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_step_arm_pol(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m);
void rex_launch_settle_base(RexSim* s, int nrec, hipStream_t st, float* snap);   // <false, *>
void rex_launch_settle_arm(RexSim* s, int nrec, hipStream_t st, float* snap);    // <true, *>

// Every step translation unit is compiled twice (rex_gym_amd/build.py): as it is -- the product kernels -- and with
// -DREX_TU_TRACE=1, the instantiations with the event trace compiled in (rex_set_event_trace; launcher names end in _trace)
#ifndef REX_TU_TRACE
#define REX_TU_TRACE 0
#endif
#ifndef REX_TU_SEG
#define REX_TU_SEG 0      /* -DREX_TU_SEG=1: the segment instantiations (rex_step_segment; launcher names end in _seg) */
#endif
#ifndef REX_TU_POL
#define REX_TU_POL 0      /* -DREX_TU_POL=1: the fused-actor instantiations (rex_step_policy / rex_step_segment_policy; launcher names end in _pol) */
#endif
#if REX_TU_TRACE
#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_trace
#elif REX_TU_SEG
#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_seg
#elif REX_TU_POL
#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_pol
#else
#define REX_STEP_LAUNCHER(group) rex_launch_step_##group
#endif
// the fused-actor kernels: four one-wave blocks to a workgroup; the actor's weights in dynamic LDS where they fit (RexSim::pol_lds_bytes, rex_set_policy)
template <int EPW, bool ARM>
static void rex_launch_policy_kernel(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  auto kern = rex::rex_step_kernel<EPW, ARM, false, false, false, true, true>;
  if (s->pol_lds_bytes > s->pol_attr_bytes) {      // (per sim, i.e. per device: the attribute belongs to the function on the current device)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, s->pol_lds_bytes);
    s->pol_attr_bytes = s->pol_lds_bytes;
  }
  constexpr int W = REX_POLICY_WAVES(EPW);
  hipLaunchKernelGGL(kern, dim3((blocks + W - 1) / W), dim3(REX_WAVE * W), (size_t)s->pol_lds_bytes, st,
                     s->dev, s->d_state, s->d_snap, a, o, r, d, m, s->pol);
}
#if REX_TU_POL
#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY) rex_launch_policy_kernel<EPW, ARM>(s, blocks, st, a, o, r, d, m)
#else
#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY)                                                                                  \
  hipLaunchKernelGGL((rex::rex_step_kernel<EPW, ARM, MIXED, BODY, REX_TU_TRACE != 0, REX_TU_SEG != 0, false>), dim3(blocks), dim3(REX_WAVE), 0, st, s->dev, s->d_state, s->d_snap, \
                     a, o, r, d, m, rex::NoPol{})
#endif
#define REX_LAUNCH_BY_EPW(ARM, MIXED, BODY)                                           \
  do {                                                                                \
    if (s->epw == 4) REX_LAUNCH_STEP(4, ARM, MIXED, BODY);                            \
    else if (s->epw == 8) REX_LAUNCH_STEP(8, ARM, MIXED, BODY);                       \
    else REX_LAUNCH_STEP(16, ARM, MIXED, BODY);                                       \
  } while (0)
