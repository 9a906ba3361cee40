/*
 * rexsim.h -- C ABI of the MI355X-native batched Rex simulator (librexsim_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of nicrusso7/rex-gym: what the reference does
 * per environment through `RexGymEnv.step()/reset()` (rex_gym/envs/rex_gym_env.py:296-414), i.e.
 *   envs/gym/walk_env.py:246-324   action -> staged gait parameters -> 12 joint targets
 *   model/gait_planner.py:96-134   Bezier/stance foot trajectories
 *   model/kinematics.py:104-142    closed-form leg IK
 *   model/rex.py:158-163,568-641   Step / ApplyAction (action_repeat x motor model + torque)
 *   model/motor.py:76-143          DC-motor PD actuator
 *   pybullet.stepSimulation        (third-party pybullet==2.8.3, requirements.txt:2)
 *   model/rex.py:717-733           ReceiveObservation
 *   envs/rex_gym_env.py:490-542    termination + reward, walk_env.py:356-362 observation
 * is done here for N environments at once by hand-written gfx950 kernels, one env per lane.
 *
 * The reference has no FFI: its boundary is the Python Gym protocol.  The binding a maintainer
 * adds is a ctypes stub; `rex_gym_amd/envs` is that stub plus the Gym surface.
 *
 * Conventions: every pointer named d_* is a DEVICE pointer (e.g. torch.Tensor.data_ptr()),
 * owned by the caller.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All calls return 0 on success or a negative REX_E* code; nothing throws; `rex_last_error()`
 * returns a thread-local message.  One host thread per RexSim.
 */
#ifndef REXSIM_H
#define REXSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REX_ABI_VERSION 6
#define REX_API __attribute__((visibility("default")))

/* tasks (reference env classes) */
#define REX_TASK_WALK   0   /* envs/gym/walk_env.py   RexWalkEnv      */
#define REX_TASK_GALLOP 1   /* envs/gym/gallop_env.py RexReactiveEnv  */
#define REX_TASK_TURN   2   /* envs/gym/turn_env.py   RexTurnEnv      */
#define REX_TASK_POSES  3   /* envs/gym/poses_env.py  RexPosesEnv     */
#define REX_TASK_STANDUP 4  /* envs/gym/standup_env.py RexStandupEnv  (the signal kwarg is unused by the env) */
#define REX_TASK_MIXED  5   /* every env runs ONE of the tasks in RexConfig.task_mix, drawn per env from its global
                               index (BASELINE.json configs[4]: the reference gets such a batch by handing BatchEnv a
                               list of different env objects, agents/tools/batch_env.py:21-44) */
/* signal_type kwarg of the reference envs */
#define REX_SIGNAL_IK 0
#define REX_SIGNAL_OL 1

/* `mark` kwarg of the reference envs (model/mark_constants.py:1-120): the robot variant.  'arm' adds the
 * 6-joint arm of rex_arm.urdf, its 6 motors held at ARM_POSES['rest'] (rex_gym_env.py:347-353). */
#define REX_MARK_BASE 0
#define REX_MARK_ARM  1

#define REX_NUM_MOTORS 12       /* mark 'base' */
#define REX_NUM_MOTORS_ARM 18   /* mark 'arm'  */

/* error codes */
#define REX_OK          0
#define REX_EINVAL     -1   /* bad argument / config */
#define REX_EHIP       -2   /* HIP runtime error (message in rex_last_error) */
#define REX_ENODEV     -3   /* no usable gfx950 device */
#define REX_ENOMEM     -4

/*
 * Per-env persistent state: SoA, float32 words, word-major ([REX_STATE_WORDS][num_envs]) so that
 * lane i of a wavefront reads word w of env i at d_state[w * num_envs + i] (coalesced).
 * Integer words are stored as raw int32 bits.
 *
 * Clocks (ABI 5).  The reference's clocks are Python floats: the env clock `step_counter * time_step` (rex.py:155-156),
 * GaitPlanner._last_time, the env's end_time.  With a 5 ms control step and gait periods such as 0.5 s, comparisons like
 * `phi >= 0.99`, `phi <= 0.5`, `t <= 0.8 + end_time` are exact ties in real numbers, decided by double rounding.  The
 * state therefore keeps those instants as the ENV STEP COUNT at which they were taken (int words LASTT, ENDTIME) and the
 * kernels rebuild the double the reference held from it; float32 carries values, never a decision of the controller.
 */
enum RexStateWord {
  REX_S_POS      = 0,   /* 3: base position, world [m]                   (getBasePositionAndOrientation) */
  REX_S_QUAT     = 3,   /* 4: base orientation x,y,z,w                                                  */
  REX_S_LINVEL   = 7,   /* 3: base linear velocity, world                (getBaseVelocity)               */
  REX_S_ANGVEL   = 10,  /* 3: base angular velocity, world                                               */
  REX_S_Q        = 13,  /* 12: motor angles, order of model/mark_constants.py:3-8                        */
  REX_S_QD       = 25,  /* 12: motor velocities                                                          */
  REX_S_PHI      = 37,  /* gait phase, float32 copy (GaitPlanner._phi, gait_planner.py:10); its one use, `_phi >= 0.99`
                           (gait_planner.py:106), is taken in double and kept as REX_F_PHASE_WRAP              */
  REX_S_LASTT    = 38,  /* int: env step whose clock GaitPlanner._last_time holds (gait_planner.py:12,107)       */
  REX_S_ALPHA    = 39,  /* gait rotation carry   (GaitPlanner._alpha,     gait_planner.py:13)            */
  REX_S_TARGET   = 40,  /* episode target (x position, or yaw for turn)  (walk_env.py:143-147)           */
  REX_S_ENDTIME  = 41,  /* int: env step at which the goal was reached; end_time = its clock (walk_env.py:213)   */
  REX_S_AUX      = 42,  /* task scratch (turn: start yaw)                                                */
  REX_S_FLAGS    = 43,  /* int: REX_F_* bits                                                             */
  REX_S_STEPS    = 44,  /* int: env steps this episode (RexGymEnv._env_step_counter)                     */
  REX_S_EPISODE  = 45,  /* int: episodes started (RNG counter)                                           */
  REX_S_MOTOR_EN = 46,  /* int: bit i = motor i enabled (Rex._motor_enabled_list, rex.py:302)            */
  REX_S_OVERHEAT = 47,  /* 6 ints: 12 x u16 overheat counters, motor 2k in low half (rex.py:301,601-608) */
  REX_S_HIST     = 53,  /* int: observation-history ring, bits 0-7 newest slot, bits 8-15 fill (1..100)      */
  REX_STATE_WORDS = 54
};

#define REX_F_GOAL_REACHED   1u   /* walk_env.py:211 */
#define REX_F_TERMINATING    2u   /* walk_env.py:214 */
#define REX_F_STAY_STILL     4u   /* walk_env.py:280 */
#define REX_F_BACKWARDS      8u   /* walk_env.py:133-136 (episode draw or fixed) */
#define REX_F_DONE          16u   /* last step returned done; must be reset before the next step */
#define REX_F_ENV_GOAL      32u   /* RexGymEnv.env_goal_reached (turn_env.py:338-340) */
#define REX_F_PHASE_WRAP    64u   /* GaitPlanner._phi >= 0.99 after the last loop(): the next loop() latches _last_time
                                     (gait_planner.py:106-107) */

typedef struct RexConfig {
  int32_t abi_version;        /* REX_ABI_VERSION */
  int32_t num_envs;           /* envs in THIS shard */
  int32_t env_index_base;     /* global index of local env 0 (multi-GPU shards): RNG streams are
                                 keyed by the global index so results do not depend on sharding */
  int32_t task;               /* REX_TASK_*   */
  int32_t signal;             /* REX_SIGNAL_* */
  int32_t action_repeat;      /* 5 (walk/turn) or 6 (gallop)  -- walk_env.py:35 */
  int32_t solver_iterations;  /* int(300 / action_repeat)     -- rex_gym_env.py:25,184 */
  float   sim_time_step;      /* control_time_step / action_repeat = 0.001 -- rex_gym_env.py:172 */
  float   motor_kp;           /* 1.0  -- walk_env.py:39 */
  float   motor_kd;           /* 0.02 -- walk_env.py:40 */
  int32_t backwards;          /* -1: draw per episode (reference None), 0/1: fixed -- walk_env.py:133 */
  float   target_position;    /* 0: draw per episode U(1,3) (or -U(2,3) backwards) -- walk_env.py:143 */
  uint64_t seed;              /* Philox key */
  int32_t auto_reset;         /* 1: an env that returns done is reset in the same launch and the
                                 returned observation is the first one of the new episode */
  int32_t max_episode_steps;  /* 0: off; else done at this many steps (LimitDuration, wrappers.py:268) */
  /* reward weights -- rex_gym_env.py:56-59 */
  float   distance_weight, energy_weight, drift_weight, shake_weight;
  float   solver_residual_threshold; /* Bullet's m_leastSquaresResidualThreshold: the PGS sweep loop ends
                                        early once max_row (delta_impulse / invdiag)^2 <= this; PyBullet's
                                        default is 1e-7 and the reference never changes it. 0 = always run
                                        solver_iterations sweeps. */
  /* RexTurnEnv (envs/gym/turn_env.py:129-160): yaw targets; bit 0 of orient_fixed = target_orient is
     given (else drawn U(0.2, 6) per episode), bit 1 = init_orient is given */
  float   target_orient, init_orient;
  int32_t orient_fixed;
  /* RexPosesEnv (envs/gym/poses_env.py:153-192): pose_index -1 = cycle base_y, base_z, roll, pitch, yaw
     per episode with a value drawn in the reference's ranges (rex_gym_env.py:258-265); 0..4 = that
     component held at pose_value (the base_y/base_z/base_roll/base_pitch/base_yaw kwargs) */
  int32_t pose_index;
  float   pose_value;
  /* 1: fold the reference's per-env wrapper stack into the launch (playground/trainer.py:48-52,
     agents/tools/wrappers.py:183-265): ClipAction to [-1,1], RangeNormalize of the action
     ((a+1)/2 (high-low)+low with the env's Box) and of the returned observation (2(o-low)/(high-low)-1).
     d_motor_cmd is unaffected. */
  int32_t range_normalize;
  /* seconds; Rex(pd_latency=, control_latency=) (model/rex.py:59-60,735-763). Non-zero values need
     rex_set_history() before the first reset. */
  float   pd_latency, control_latency;
  /* REX_MARK_BASE / REX_MARK_ARM.  With REX_MARK_ARM the per-env state has rex_state_words() = 69 words (the q, qd
     and overheat blocks grow to 18 motors, every other word keeps its order), d_motor_cmd rows are 18 wide and the
     gallop observation is 4 + 18; the history records of the latency model are 61 words (REX_HISTORY_WORDS_ARM). */
  int32_t mark;
  /* ---- ABI 3 ---- */
  /* GaitPlanner.loop reads WALL-CLOCK time (model/gait_planner.py:108-110: `time.time()`), so the reference's gait
     runs at (wall seconds per simulated second) x the nominal rate of the host it runs on: rendered / real-time
     playback = 1, a host that simulates faster than real time < 1, slower (GUI mode, many worker processes per core)
     > 1.  The gait phase clock here is simulated time x gait_clock_scale.  0 means 1.0. */
  float   gait_clock_scale;
  /* 1: contact rows for the link collision boxes (base, chassis, shoulder, leg, foot boxes of rex.urdf) against the
     ground, next to the toe rows, and for the leg and foot boxes against the boxes of the base body (the reference loads
     the robot with URDF_USE_SELF_COLLISION, model/rex.py:276-281); 0: toes only (the fast path; the boxes of a robot that
     satisfies its env's termination test stay clear of the ground and of each other).
     rex_default_config sets 1 for REX_TASK_POSES -- the env that never terminates and whose roll poses press the base
     into the upper-leg boxes -- and 0 for every other task. */
  int32_t body_contacts;
  /* Rex(observation_noise_stdev=...) (model/rex.py:22,765-769): Gaussian noise added by the sensor getters --
     [0] motor angles, [1] motor velocities, [2] motor torques, [3] base roll/pitch/yaw, [4] base angular rates.
     Every getter call of the reference draws afresh; here every call site of a step draws from the env's Philox
     stream (counter = episode, step, call site).  All 0 (the reference default): no draws. */
  float   noise_stdev[5];
  /* REX_TASK_MIXED: bit t set = task t is in the mix.  Env g runs task_of(g) = the (Philox(seed; g) mod count)-th set
     bit for its whole life (a reference env object never changes class).  All envs use `signal`; the action row is
     as wide as the widest task of the mix, the observation row likewise (narrower tasks leave the tail 0). */
  int32_t task_mix;
  /* Per-reset domain randomisation (the env_randomizer hook, rex_gym_env.py:345-346, turning the knobs of
     Rex.SetBaseMasses / SetLegMasses, model/rex.py:659-692, plus the foot friction): on every reset env g draws
     base-mass scale, leg-mass scale ~ U(mass_scale_lo, mass_scale_hi) and foot friction ~ U(friction_lo, friction_hi)
     from its Philox stream (counter = episode).  lo == hi == 0: off (that quantity keeps its URDF value, or the
     rex_set_body_params entry).  As with changeDynamics(mass=...), inertia tensors keep their load-time values, and
     the reset motion is the nominal robot's (the reference randomises after Rex.Reset). */
  float   mass_scale_lo, mass_scale_hi, friction_lo, friction_hi;
  /* Height the robot is dropped from at reset (ROBOT_INIT_POSITION[terrain_id][2], model/terrain.py:14-20: 0.21 on
     plane / random / maze, 0.85 on mounts, 1.98 on hills).  0 means 0.21. */
  float   init_height;
  /* ---- ABI 4 ---- */
  /* Rex(on_rack=True) (model/rex.py:269-287, rex_gym_env.py:140): the debug mode that hangs the robot on a rack --
     loadURDF(useFixedBase=True) at INIT_RACK_POSITION = [0, 0, 1].  The base neither moves nor turns, the legs swing
     under gravity and the motors; the reset pose is [0, 0, 1] (init_height is ignored) and RexTurnEnv starts at its
     fixed debug yaw 2.1 (turn_env.py:140-143). */
  int32_t on_rack;
  /* ---- ABI 5 ---- */
  /* RexGymEnv(forward_reward_cap=...) (envs/rex_gym_env.py:81,217,525): `forward_reward = min(forward_reward, cap)` in the
     base reward.  +inf (the reference default, what rex_default_config sets): no cap.  NaN is rejected. */
  float   forward_reward_cap;
  /* ---- ABI 6 ---- */
  /* RexReactiveEnv(use_angle_in_observation=False) (envs/gym/gallop_env.py:56,93,344-356,374-377): 1 = the gallop observation is the four
     base words alone (roll, pitch and their rates), without the motor angles; rex_obs_dim() is then 4 for gallop.  0: the reference
     default, 4 + num_motors. */
  int32_t gallop_no_angles;
} RexConfig;

typedef struct RexSim RexSim;

/* Fill *cfg with the reference defaults of `task`/`signal` for `num_envs` envs. */
REX_API int rex_default_config(int task, int signal, int num_envs, RexConfig* cfg);

/* Dimensions implied by a config (action/observation vector lengths of the reference env). */
REX_API int rex_action_dim(const RexConfig* cfg);
REX_API int rex_obs_dim(const RexConfig* cfg);
/* Motors (12 / 18) and per-env state words (REX_STATE_WORDS / 69) of the config's mark. */
REX_API int rex_num_motors(const RexConfig* cfg);
REX_API int rex_state_words(const RexConfig* cfg);

/* Create a simulator on HIP device `device`. `d_state` is a caller-owned device buffer of
 * rex_state_words(cfg) * num_envs float32 words; the library never allocates per-env memory.
 * Computes the settled reset snapshot (rex.py:314-323: 100 + 500 substeps holding the init
 * pose) once, on the device, with the same kernels. */
REX_API int rex_create(const RexConfig* cfg, int device, float* d_state, void* stream, RexSim** out);
REX_API int rex_destroy(RexSim* sim);

/* terrain_type='random' (model/terrain.py:32-54): install a pool of k heightfields. d_heights [k][256*256] float32
 * raw vertex heights in the reference's layout (data[i + j*256], i along x; 5 cm cells centred on the origin),
 * d_mids [k] = (min+max)/2 of each field (Bullet centres the shape there).  Caller-owned device buffers that must
 * outlive the sim.  Each episode of env g uses field (g + 977*episode) mod k; the reset motion is re-run once per
 * field (one settled snapshot each).  k = 0 returns to the flat plane.  Call before rex_reset. */
REX_API int rex_set_terrain(RexSim* sim, const float* d_heights, const float* d_mids, int k, void* stream);

/* Any other heightfield terrain (the reference's 'hills' csv and 'mounts' / 'maze' png fields, model/terrain.py:55-78,
 * whose files live in the pip package pybullet_data and are not part of rex-gym): a pool of k fields of nx x ny vertex
 * heights IN METRES (meshScale z already applied), laid out like rex_set_terrain's (data[ix + iy*nx], ix along x),
 * vertex spacing cell_x / cell_y (meshScale x / y), the grid centred on (origin_x, origin_y).  d_mids[k] = the raw
 * height that ends up at world z = 0: Bullet centres a heightfield shape on the middle of its height range and the
 * reference then places the body at some z0 (terrain.py:64,75), so d_mids[i] = (min_i + max_i) / 2 - z0.  The z = 0
 * ground box of plane.urdf stays underneath, as in the reference.  Everything else as rex_set_terrain. */
REX_API int rex_set_heightfield(RexSim* sim, const float* d_heights, const float* d_mids, int k, int nx, int ny,
                                float cell_x, float cell_y, float origin_x, float origin_y, void* stream);

/* Domain randomisation hooks (the knobs an env_randomizer turns: Rex.SetBaseMasses / SetLegMasses, model/rex.py:659-692;
 * the reference has no friction setter).  d_params: caller-owned device array [3][num_envs], word-major:
 * base-link mass scale, leg-link mass scale, foot friction coefficient; read on every step (so the caller may rewrite
 * entries when it resets envs).  As in Bullet's changeDynamics(mass=...), only masses change: the inertia tensors stay
 * the ones computed at load.  NULL restores (1, 1, 0.5). */
REX_API int rex_set_body_params(RexSim* sim, const float* d_params);

/* Debug: trace the discrete events of the restated stepSimulation.  d_trace: caller-owned device buffer uint32 [3][num_envs]
 * (NULL switches the trace off, the default; the caller zeroes it).  While set, every substep folds into word [0][i] of env i
 * which toe points are within the contact breaking distance, the heightfield facet under each of them (under the toe end's
 * centre and under the contact point, and the branch btPlaneSpace1 takes for its friction directions), and which joint / arm
 * bounds are reached -- and every control step the controller's flags and gait latches; into word [1][i] the env's solver sweep
 * count; into word [2][i] the same events as [0] without the arm's bounds (mark arm commands three arm joints beyond their
 * bounds: those rows switch with the last bit of the joint angle).  The words are chained hashes: two runs (or this library and the CPU oracle, which folds the same words the same
 * way) agree on an env's word exactly as long as they took every such decision alike -- what separates their trajectories
 * then is continuous round-off, not a contact that switched.  One read-modify-write of three words per substep; off in
 * production and in bench.py. */
REX_API int rex_set_event_trace(RexSim* sim, uint32_t* d_trace);

/* Observation-history ring for the latency model (Rex._observation_history, deque(maxlen=100) of 43-vectors:
 * q, qd, observed torque, base quaternion, base angular velocity; model/rex.py:122,717-763).  d_history: caller-owned
 * device buffer of REX_HISTORY_LEN * REX_HISTORY_WORDS (_ARM for mark 'arm': 61-vectors) * num_envs float32, laid out
 * [slot][word][env].  Required only
 * when pd_latency or control_latency is non-zero (without a latency the delayed observation is the newest one and the
 * buffer is not used).  The latency model also runs through the reset motion (rex.py:309-323): every reset, by call or
 * in-launch, restores an env's ring to the 100 observations the reset motion leaves behind, so that an episode starts
 * exactly as the reference's does. */
#define REX_HISTORY_LEN 100
#define REX_HISTORY_WORDS 43       /* mark 'base': 3 x 12 + 7 */
#define REX_HISTORY_WORDS_ARM 61   /* mark 'arm':  3 x 18 + 7 */
REX_API int rex_set_history(RexSim* sim, float* d_history);

/* Reset envs. d_indices == NULL: all envs. Else n int32 env indices (device).  Writes the first
 * observation of each reset env to d_obs[row * obs_dim] where row = position in d_indices (or the
 * env index when d_indices is NULL).  Mirrors RexWalkEnv.reset (walk_env.py:125-154). */
REX_API int rex_reset(RexSim* sim, const int32_t* d_indices, int n, float* d_obs, void* stream);

/* One env.step() for every env (rex_gym_env.py:369-414):
 *   d_action [N, action_dim] in  -- raw env action (NOT range-normalised)
 *   d_obs    [N, obs_dim]    out
 *   d_reward [N]             out
 *   d_done   [N] uint8       out
 *   d_motor_cmd [N, 12]      out, nullable -- info['action'], the 12 motor targets
 */
REX_API int rex_step(RexSim* sim, const float* d_action, float* d_obs, float* d_reward,
             uint8_t* d_done, float* d_motor_cmd, void* stream);

/* A rollout SEGMENT in one launch: num_steps consecutive env.step() calls for every env, for a caller that already holds the actions
 * of the whole segment (open-loop rollouts: random-action throughput runs, replayed action tapes, a policy that acts on a stale
 * observation by design).  Step t reads d_action[t] and writes d_obs[t], d_reward[t], d_done[t], d_motor_cmd[t]; the blocks are
 *   d_action [T, N, action_dim]   d_obs [T, N, obs_dim]   d_reward [T, N]   d_done [T, N] uint8   d_motor_cmd [T, N, num_motors] (nullable)
 * and the results are BIT-IDENTICAL to T calls of rex_step on the slices (tests/test_gpu_parity.py::
 * test_segment_launch_is_bit_identical_to_single_steps): the same step code runs T times, an env's state staying in registers in
 * between.  What it buys: a launch ends with its slowest wave; stepping one by one pays the slowest wave of EVERY step (a freshly
 * reset or fallen robot that runs into the solver's sweep cap), a segment pays the largest SUM over a wave's steps.  It replaces
 * nothing of the reference's surface (env.step() stays one rex_step): the reference has no counterpart -- its envs step one
 * action at a time (rex_gym_env.py:369).  Envs that finish an episode inside the segment are reset in the launch when the sim was
 * created with auto_reset, exactly as rex_step does; sims that regroup their envs between steps regroup between segments.
 * num_steps * N * max(action_dim, obs_dim, num_motors) must stay below 2^31. */
REX_API int rex_step_segment(RexSim* sim, int num_steps, const float* d_action, float* d_obs, float* d_reward,
             uint8_t* d_done, float* d_motor_cmd, void* stream);

/* ---- ABI 6: the actor inside the launch -- closed-loop rollouts, one launch per step or per segment ----
 * The reference's rollout is `action = algo.perform(prevob)` -> `batch_env.simulate(action)` every step (agents/tools/simulate.py:
 * 57-76, agents/ppo/algorithm.py:105-134): a policy in the loop, so the actions of step t + 1 do not exist before step t has returned
 * its observation and rex_step_segment cannot serve it.  rex_step_policy / rex_step_segment_policy evaluate the actor of the
 * reference's agents INSIDE the step kernel: the observ filter (agents/ppo/normalize.py:47-66: centre, scale, clip -- with the
 * statistics the caller hands in, frozen for the launch), the network every shipped config uses (agents/scripts/networks.py:66-110
 * ForwardGaussianPolicy: two ReLU layers, tanh mean layer, a free logstd vector; configs.py:29-34) and the Gaussian sample of
 * `network.policy.sample` (Philox keyed by seed, global env index, episode, step: a rollout is a pure function of the seeds and the
 * weights, whatever the sharding or the segment length).
 * The arrays are caller-owned DEVICE buffers; rex_set_policy SNAPSHOTS them (a packing kernel on `stream`, into a library-owned
 * buffer laid out for the kernels: rex_gym_amd/csrc/rex_policy.h), so they only have to stay valid until the stream has passed that
 * point, and a learner that has updated its weights or its filter statistics calls rex_set_policy again.  Weight matrices are
 * input-major: d_w1[k * hidden1 + j] = the weight from input k to unit j (the transpose of a torch.nn.Linear weight; the layout
 * of a TF1 `fully_connected/weights` variable).
 * Needs range_normalize = 1 in the sim's config -- the reference's agents always act through RangeNormalize + ClipAction
 * (playground/trainer.py:48-52): the sampled action is clipped to [-1, 1] and mapped to the env's Box inside the launch, a fused
 * path has no Box test to fail -- a single-task sim (not REX_TASK_MIXED), toes-only contact rows (body_contacts = 0), no event trace. */
typedef struct RexPolicy {
  int32_t obs_dim, action_dim;        /* must equal rex_obs_dim / rex_action_dim of the sim's config */
  int32_t hidden1, hidden2;           /* units of the two ReLU layers (configs.py:31: 200, 100); obs_dim + 12 + hidden1 + hidden2 floats
                                         (each term rounded up to a multiple of 4) per env must fit the kernel's contact-row region of
                                         LDS (448 for mark base up to 8 192 envs) */
  const float* d_w1; const float* d_b1;   /* [obs_dim][hidden1], [hidden1] */
  const float* d_w2; const float* d_b2;   /* [hidden1][hidden2], [hidden2] */
  const float* d_w3; const float* d_b3;   /* [hidden2][action_dim], [action_dim]: the mean layer, tanh on top */
  const float* d_logstd;                  /* [action_dim] */
  const float* d_obs_mean;                /* [obs_dim], nullable (with d_obs_scale): no observ filter */
  const float* d_obs_scale;               /* [obs_dim]: 1 / (std + 1e-8) of the filter (normalize.py:60-62) */
  float   obs_clip;                       /* 5 (algorithm.py:48-52) */
  int32_t sample;                         /* 1: action = mean + exp(logstd) * N(0, 1) (training); 0: action = mean (evaluation) */
  uint64_t seed;                          /* Philox key of the samples */
} RexPolicy;
/* Install the policy (validates the dimensions against the sim, packs the arrays on `stream`); NULL removes it. */
REX_API int rex_set_policy(RexSim* sim, const RexPolicy* policy, void* stream);

/* One closed-loop env.step() for every env: action = perform(d_obs_in), then the step.
 *   d_obs_in [N, obs_dim]     in  -- the observation the envs returned last (rex_reset, or the previous step's d_obs)
 *   d_action [N, action_dim]  out -- the action the policy took, as the agent's memory stores it (BEFORE ClipAction / RangeNormalize)
 *   d_mean   [N, action_dim]  out, nullable -- the policy's mean (algorithm.py:126-133 stores action, mean and logstd)
 *   d_obs, d_reward, d_done, d_motor_cmd  out -- as rex_step (d_obs may NOT alias d_obs_in) */
REX_API int rex_step_policy(RexSim* sim, const float* d_obs_in, float* d_action, float* d_mean, float* d_obs, float* d_reward,
             uint8_t* d_done, float* d_motor_cmd, void* stream);
/* A closed-loop rollout SEGMENT in one launch: step t acts on d_obs[t - 1] (d_obs_in for t = 0) and writes d_action[t], d_mean[t],
 * d_obs[t], d_reward[t], d_done[t], d_motor_cmd[t] -- blocks [T, N, ...] as in rex_step_segment.  BIT-IDENTICAL to T calls of
 * rex_step_policy chained through their observations (tests/test_gpu_policy.py), with what a segment launch buys (rex_step_segment).
 * A caller that keeps one block obs[T + 1, N, obs_dim] passes d_obs_in = obs[0], d_obs = obs[1]: prevob of step t is obs[t]. */
REX_API int rex_step_segment_policy(RexSim* sim, int num_steps, const float* d_obs_in, float* d_action, float* d_mean, float* d_obs,
             float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream);

/* HIP event timing of rex_step launches on their own stream (ms).  rex_set_timing(1): one event pair, read with
 * rex_last_step_ms (synchronises on the launch).  rex_set_timing(2): a ring of event pairs around the last 256 launches,
 * recorded without any host synchronisation between launches (the stream stays full, so a duration is that of the kernel
 * itself, not of an idle queue being refilled); rex_step_times_ms waits for the newest one and returns the last
 * min(max_count, recorded, 256 -- 4096 in mode 3) durations, oldest first -- the return value is their number (< 0: error).
 * rex_set_timing(3): the next 4096 launches are timed on the DEVICE: every workgroup of a launch folds its first / last wall-clock
 * tick (constant 100 MHz counter) into a (min, max) pair, so a duration is first-wave-start to last-wave-end of the kernel
 * alone -- what rocprofv3's kernel trace reports -- with no event or dispatch hand-over in it.
 * In every mode the timed span is the STEP KERNEL's: a sim that regroups its envs every step (single task: from 262 144 envs or
 * REX_REGROUP=1; mixed tasks: beyond one workgroup per SIMD) issues two small sorting launches behind each step kernel on the same
 * stream, and those are NOT inside the span -- compare with the wall clock per step (bench.py reports both: `ms_per_step` next to
 * `roofline.kernel_ms`, and `roofline.regroup_launches_ms` = their difference when the sim regroups). */
REX_API int rex_set_timing(RexSim* sim, int enable);
REX_API int rex_last_step_ms(RexSim* sim, float* ms);
REX_API int rex_step_times_ms(RexSim* sim, float* ms, int max_count);

/* ---- controller-only entry points (parity tests of the controller half) ----
 * n independent problems, row-major device arrays, all float32. */
/* model/kinematics.py:104-142 -- orn[n,3], pos[n,3], frames[n,4,3] (FR,FL,RR,RL) ->
 * angles[n,4,3] in the same leg order (theta, -alpha, -gamma) */
REX_API int rex_ik_solve(int n, const float* d_orn, const float* d_pos, const float* d_frames,
                 float* d_angles, void* stream);
/* model/motor.py:76-143 -- cmd,q,qd,qd_true [n] -> actual, observed torque [n] */
REX_API int rex_motor_torque(int n, const float* d_cmd, const float* d_q, const float* d_qd,
                     const float* d_qd_true, float kp, float kd,
                     float* d_actual, float* d_observed, void* stream);
/* model/gait_planner.py:96-134 with the phase clock on explicit time `now`.
 * mode: 0 walk, 1 gallop.  d_planner [n,3] = (phi, last_time, alpha) in/out; params[n,6] =
 * (v, angle_deg, w_rot, period, direction, now) -- both float64: the clock values and the phase decide branches and must
 * arrive unrounded (see "Clocks" above); the trajectory arithmetic is float32.  frames_out[n,4,3] float32 (FR,FL,RR,RL). */
REX_API int rex_gait_loop(int n, int mode, double* d_planner, const double* d_params,
                  float* d_frames_out, void* stream);

/* Envs per wavefront of the step kernel this sim launches (4, 8, 16: lane groups; 64: one env per lane).  Chosen from
 * the batch size; the environment variable REX_ENVS_PER_WAVE overrides (tests run every variant through it). */
REX_API int rex_envs_per_wave(const RexSim* sim);

/* How a REX_TASK_MIXED batch is laid out over the wavefronts of the step kernel (host-only, no GPU needed).  An env of a
 * mixed batch keeps the task drawn for it for life, so the library decides once which envs share a wave: chunks of
 * neighbouring envs, sorted by task inside a chunk, every task's run padded to whole waves -- a wave then runs ONE task
 * (wave-uniform action_repeat / sweep cap / action box / reward weights) and the chunks' workgroups are dealt to the XCDs.
 * slots[blk * envs_per_wave + k] = env of slot k of workgroup blk (-1: padding), tasks[blk] = that workgroup's task.
 * (A batch that needs more workgroups than one round of the machine holds is instead regrouped every step: one region of whole
 * waves per task, sorted inside by the solver sweeps of the last step; the per-wave rule -- one task -- is the same.)
 * Returns the number of workgroups (pass NULL buffers to size them), or a negative error. */
REX_API int rex_mixed_slot_map(const RexConfig* cfg, int envs_per_wave, int32_t* slots, int32_t* tasks, int max_blocks);

/* Solver sweeps every env ran in the last rex_step (summed over its substeps; after rex_step_segment: the mean per step over the
 * segment), int32 [num_envs] copied to the caller's
 * device buffer -- the key a batch created under REX_REGROUP=1 is regrouped into waves by (opt-in). */
REX_API int rex_get_sweeps(RexSim* sim, int32_t* d_out, void* stream);

REX_API const char* rex_last_error(void);
REX_API int rex_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REXSIM_H */
