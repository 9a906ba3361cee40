// rexsim.hip -- kernels + C ABI (include/rexsim.h) of the MI355X-native batched Rex simulator.
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC rexsim.hip -o librexsim_hip.so
//
// The step and settle kernels are instantiated in rex_step_*.hip / rex_settle.hip (rex_kernels.h); this file holds the
// C ABI, the reset kernel and the small kernels (regrouping, controller-only entry points).
#include "rex_kernels.h"
#include <algorithm>
#include <cstdarg>
#include <vector>

namespace rex {

// rex_set_policy: the caller's plain arrays (input-major weight matrices) -> the packed actor.  One thread per destination float.
struct PolSrc { const float *w1, *b1, *w2, *b2, *w3, *b3, *logstd, *mean, *scale; };
__global__ void rex_pack_policy_kernel(PolSrc s, int O, int A, int H1, int H2, float* __restrict__ dst) {
  const PolOff o = policy_offsets(O, A, H1, H2);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < o.total; t += gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (t < o.b1) { const int r = t & 3, j = (t >> 2) % H1, k = 4 * ((t >> 2) / H1) + r; if (k < O) v = s.w1[k * H1 + j]; }
    else if (t < o.w2) { if (t - o.b1 < H1) v = s.b1[t - o.b1]; }
    else if (t < o.b2) { const int u = t - o.w2, r = u & 3, j = (u >> 2) % H2, k = 4 * ((u >> 2) / H2) + r; if (k < H1) v = s.w2[k * H2 + j]; }
    else if (t < o.w3) { if (t - o.b2 < H2) v = s.b2[t - o.b2]; }
    else if (t < o.b3) { if (t - o.w3 < H2 * A) v = s.w3[t - o.w3]; }
    else if (t < o.logstd) { if (t - o.b3 < A) v = s.b3[t - o.b3]; }
    else if (t < o.mean) { if (t - o.logstd < A) v = s.logstd[t - o.logstd]; }
    else if (t < o.scale) { if (t - o.mean < O && s.mean) v = s.mean[t - o.mean]; }
    else if (t < o.scale + O) { v = s.scale ? s.scale[t - o.scale] : 1.0f; }
    dst[t] = v;
  }
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
// The same sort over many workgroups (the one-workgroup kernel above takes 0.3 ms for 262 144 envs: a tenth of their step):
// workgroup b counts its REX_REGROUP_CHUNK envs per bin; the last workgroup to finish turns the counts into start offsets
// (bins in order, workgroups in order inside a bin: the permutation is the one-workgroup kernel's up to the order inside a
// bin); a second launch scatters.
#define REX_REGROUP_CHUNK 1024
__device__ __forceinline__ int regroup_bin(int sweeps, int bin_width) { return 63 - min(sweeps / bin_width, 63); }
__global__ __launch_bounds__(256) void rex_regroup_count_kernel(int n, int bin_width, const int32_t* __restrict__ sweeps, int32_t* __restrict__ counts,
                                                                unsigned* __restrict__ done) {
  __shared__ int hist[64];
  __shared__ bool last;
  const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int k = t; k < REX_REGROUP_CHUNK; k += 256) {
    const int i = b * REX_REGROUP_CHUNK + k;
    if (i < n) atomicAdd(&hist[regroup_bin(sweeps[i], bin_width)], 1);
  }
  __syncthreads();
  if (t < 64) counts[b * 64 + t] = hist[t];
  __threadfence();
  __syncthreads();
  if (t == 0) last = atomicAdd(done, 1u) == (unsigned)nb - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // counts[b][bin] -> first slot of workgroup b's envs of that bin
  __shared__ int total[64], first[64];
  if (t < 64) { int acc = 0; for (int k = 0; k < nb; ++k) acc += counts[k * 64 + t]; total[t] = acc; }
  __syncthreads();
  if (t == 0) { int acc = 0; for (int k = 0; k < 64; ++k) { first[k] = acc; acc += total[k]; } *done = 0u; }
  __syncthreads();
  if (t < 64) { int acc = first[t]; for (int k = 0; k < nb; ++k) { const int c = counts[k * 64 + t]; counts[k * 64 + t] = acc; acc += c; } }
}
__global__ __launch_bounds__(256) void rex_regroup_scatter_kernel(int n, int bin_width, const int32_t* __restrict__ sweeps, const int32_t* __restrict__ counts,
                                                                  int32_t* __restrict__ perm) {
  __shared__ int cursor[64];
  const int t = threadIdx.x, b = blockIdx.x;
  if (t < 64) cursor[t] = counts[b * 64 + t];
  __syncthreads();
  for (int k = t; k < REX_REGROUP_CHUNK; k += 256) {
    const int i = b * REX_REGROUP_CHUNK + k;
    if (i < n) perm[atomicAdd(&cursor[regroup_bin(sweeps[i], bin_width)], 1)] = i;
  }
}
// Regrouping of a REX_TASK_MIXED batch: every task of the mix owns a STATIC region of the slot map (whole waves; the envs keep
// their task for life, so the region sizes and its padding slots never change); inside its region a task's envs are sorted by
// the sweeps of their last step, most first -- the waves of a large mixed batch then hold envs that need about the same number
// of sweeps, and the waves that exceed one round of the machine are the short ones.  Bins: 64 / n_mix sweep bins per task.
__device__ __forceinline__ int regroup_bin_mixed(int sweeps, int bin_width, int cls, int w) { return cls * w + (w - 1 - min(sweeps / bin_width, w - 1)); }
__global__ __launch_bounds__(256) void rex_regroup_mixed_count_kernel(int n, int bin_width, MixRegions mr, const int32_t* __restrict__ cls,
                                                                      const int32_t* __restrict__ sweeps, int32_t* __restrict__ counts,
                                                                      unsigned* __restrict__ done) {
  __shared__ int hist[64];
  __shared__ bool last;
  const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  if (t < 64) hist[t] = 0;
  __syncthreads();
  for (int k = t; k < REX_REGROUP_CHUNK; k += 256) {
    const int i = b * REX_REGROUP_CHUNK + k;
    if (i < n) atomicAdd(&hist[regroup_bin_mixed(sweeps[i], bin_width, cls[i], mr.bins_per_task)], 1);
  }
  __syncthreads();
  if (t < 64) counts[b * 64 + t] = hist[t];
  __threadfence();
  __syncthreads();
  if (t == 0) last = atomicAdd(done, 1u) == (unsigned)nb - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  __shared__ int total[64], first[64];
  if (t < 64) { int acc = 0; for (int k = 0; k < nb; ++k) acc += counts[k * 64 + t]; total[t] = acc; }
  __syncthreads();
  if (t == 0) {
    for (int c = 0; c < mr.n_mix; ++c) {   // a task's bins fill its own region from its first slot
      int acc = mr.base[c];
      for (int k = 0; k < mr.bins_per_task; ++k) { first[c * mr.bins_per_task + k] = acc; acc += total[c * mr.bins_per_task + k]; }
    }
    *done = 0u;
  }
  __syncthreads();
  if (t < mr.n_mix * mr.bins_per_task) { int acc = first[t]; for (int k = 0; k < nb; ++k) { const int c = counts[k * 64 + t]; counts[k * 64 + t] = acc; acc += c; } }
}
__global__ __launch_bounds__(256) void rex_regroup_mixed_scatter_kernel(int n, int bin_width, MixRegions mr, const int32_t* __restrict__ cls,
                                                                        const int32_t* __restrict__ sweeps, const int32_t* __restrict__ counts,
                                                                        int32_t* __restrict__ slot_env) {
  __shared__ int cursor[64];
  const int t = threadIdx.x, b = blockIdx.x;
  if (t < 64) cursor[t] = counts[b * 64 + t];
  __syncthreads();
  for (int k = t; k < REX_REGROUP_CHUNK; k += 256) {
    const int i = b * REX_REGROUP_CHUNK + k;
    if (i < n) slot_env[atomicAdd(&cursor[regroup_bin_mixed(sweeps[i], bin_width, cls[i], mr.bins_per_task)], 1)] = i;
  }
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

__global__ void rex_gait_kernel(int n, int mode, double* __restrict__ planner, const double* __restrict__ params,
                                float* __restrict__ frames) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  GaitState g{planner[3 * i], planner[3 * i + 1], (float)planner[3 * i + 2]};
  const double* p = params + 6 * i;
  float f[12];
  gait_loop(g, mode, (float)p[0], (float)p[1], (float)p[2], p[3], (float)p[4], p[5], f);
  planner[3 * i] = g.phi; planner[3 * i + 1] = g.last_time; planner[3 * i + 2] = g.alpha;
  for (int k = 0; k < 12; ++k) frames[12 * i + k] = f[k];
}

}  // namespace rex

// =================================================================================================
//                                          host side: C ABI
// =================================================================================================

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* detail) {
  snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
  return code;
}
static int failf(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int failf(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
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
  cfg->forward_reward_cap = INFINITY;                            /* rex_gym_env.py:81 */
  cfg->gallop_no_angles = 0;                                     /* gallop_env.py:56 use_angle_in_observation=True */
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
  if (c->gallop_no_angles) return 4;                                             /* use_angle_in_observation=False, gallop_env.py:344-356 */
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
  if (c->forward_reward_cap != c->forward_reward_cap) return fail(REX_EINVAL, "forward_reward_cap is NaN (use +inf for no cap)%s", "");
  if (c->task == REX_TASK_MIXED && c->energy_weight != rex::task_energy_weight(REX_TASK_WALK))
    return fail(REX_EINVAL, "energy_weight is a per-task constant in a REX_TASK_MIXED batch (gallop 0.005, the others 0.0005): leave it at the default%s", "");
  if (c->mass_scale_lo < 0.0f || c->mass_scale_hi < c->mass_scale_lo || c->friction_lo < 0.0f || c->friction_hi < c->friction_lo)
    return fail(REX_EINVAL, "randomisation ranges must satisfy 0 <= lo <= hi%s", "");
  return REX_OK;
}

// a snapshot record: the state words, and behind all records the observation rings the reset motion leaves behind
static size_t snapshot_floats(const RexSim* s, int nrec) {
  const bool ring = s->cfg.pd_latency > 0.0f || s->cfg.control_latency > 0.0f;
  return (size_t)nrec * ((size_t)s->words + (ring ? (size_t)REX_HISTORY_LEN * (size_t)s->dev.hist_words : 0));
}

// ---- REX_TASK_MIXED: the task-sorted slot map ----
// An env of a mixed batch keeps the task drawn for it (rex::mixed_task_of: one Philox block keyed by the seed and the global
// env index) for its whole life, so which envs share a wave is decided ONCE: the envs are cut into chunks of neighbouring
// indices (their state words share 64-byte sectors), a chunk's envs are sorted by task, every task's run is padded to whole
// waves, and the chunks' workgroups are dealt to the XCDs (workgroup b runs on XCD b % 8) so that one L2 fetches a chunk's
// sectors.  The step kernel then runs waves of ONE task: action_repeat, sweep cap, action box and reward weight are
// wave-uniform, a 5-substep wave does not sit through gallop's sixth substep, and the kernel is the single-task kernel.
static void host_philox4x32(uint32_t* c, uint32_t k0, uint32_t k1) {   // rex::philox4x32 on the host
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static int host_mixed_task_of(const rex::DevCfg& d, int gidx) {      // rex::mixed_task_of on the host
  uint32_t ctr[4] = {0xFFFFFFFFu, (uint32_t)gidx, 2u, 0u};
  host_philox4x32(ctr, d.seed_lo, d.seed_hi);
  return d.mix_task[ctr[0] % (uint32_t)d.n_mix];
}
// slots[blk * epw + k] = env of wave slot k of workgroup blk (-1: padding), tasks[blk] = the workgroup's task
// returns the number of workgroups that hold envs (the others leave at once)
static int task_slot_map(const rex::DevCfg& d, int epw, std::vector<int32_t>& slots, std::vector<int32_t>& tasks) {
  const int n = d.n;
  // 8 m chunks of whole sectors (m per XCD), each at most 64 waves' worth: padding <= (epw - 1) slots per task and chunk
  const int m = (n + 8 * 64 * epw - 1) / (8 * 64 * epw);
  const int chunk = ((n + 8 * m - 1) / (8 * m) + 15) / 16 * 16;
  const int nchunks = (n + chunk - 1) / chunk;
  std::vector<std::vector<int32_t>> xs(8), xt(8);    // per XCD: its workgroups' slots and tasks
  for (int c = 0; c < nchunks; ++c) {
    std::vector<int32_t>& s = xs[c & 7];
    for (int k = 0; k < d.n_mix; ++k) {
      const size_t before = s.size();
      for (int i = c * chunk; i < n && i < (c + 1) * chunk; ++i)
        if (host_mixed_task_of(d, d.env_index_base + i) == d.mix_task[k]) s.push_back(i);
      while ((s.size() - before) % (size_t)epw) s.push_back(-1);
      for (size_t b = before / epw; b < s.size() / epw; ++b) xt[c & 7].push_back(d.mix_task[k]);
    }
  }
  size_t rounds = 0, busy = 0;
  for (int x = 0; x < 8; ++x) { busy += xt[x].size(); if (xt[x].size() > rounds) rounds = xt[x].size(); }
  slots.assign(rounds * 8 * (size_t)epw, -1);
  tasks.assign(rounds * 8, d.mix_task[0]);
  for (int x = 0; x < 8; ++x)
    for (size_t j = 0; j < xt[x].size(); ++j) {
      const size_t b = 8 * j + (size_t)x;
      tasks[b] = xt[x][j];
      for (int k = 0; k < epw; ++k) slots[b * epw + k] = xs[x][j * epw + k];
    }
  while (!tasks.empty() && slots[(tasks.size() - 1) * epw] < 0) { tasks.pop_back(); slots.resize(tasks.size() * epw); }   // trailing padding workgroups
  return (int)busy;
}

// The layout a regrouped mixed batch starts from (and keeps the shape of): task slot k's envs in index order in ONE region of
// whole waves per task, regions in the order of the mix; cls[i] = task slot of env i, base[k] = first slot of region k.
static void task_region_map(const rex::DevCfg& d, int epw, std::vector<int32_t>& slots, std::vector<int32_t>& tasks, std::vector<int32_t>& cls,
                            int32_t base[5]) {
  cls.resize((size_t)d.n);
  for (int i = 0; i < d.n; ++i) {
    const int t = host_mixed_task_of(d, d.env_index_base + i);
    int k = 0;
    for (int j = 1; j < d.n_mix; ++j) if (d.mix_task[j] == t) k = j;
    cls[i] = k;
  }
  slots.clear(); tasks.clear();
  for (int k = 0; k < 5; ++k) base[k] = 0;
  for (int k = 0; k < d.n_mix; ++k) {
    base[k] = (int32_t)slots.size();
    for (int i = 0; i < d.n; ++i) if (cls[i] == k) slots.push_back(i);
    while (slots.size() % (size_t)epw) slots.push_back(-1);
    while (tasks.size() < slots.size() / epw) tasks.push_back(d.mix_task[k]);
  }
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
  s->have_policy = 0; s->use_policy = false;
  rex::DevCfg& d = s->dev;
  d.n = cfg->num_envs; d.env_index_base = cfg->env_index_base; d.task = cfg->task; d.signal = cfg->signal;
  d.nsteps = 1;
  d.action_repeat = cfg->action_repeat; d.iterations = cfg->solver_iterations; d.dt = cfg->sim_time_step;
  d.kp = cfg->motor_kp; d.kd = cfg->motor_kd; d.res_thr = sqrtf(fmaxf(cfg->solver_residual_threshold, 0.0f)); d.backwards = cfg->backwards; d.target_position = cfg->target_position;
  d.seed_lo = (uint32_t)cfg->seed; d.seed_hi = (uint32_t)(cfg->seed >> 32);
  d.auto_reset = cfg->auto_reset; d.max_steps = cfg->max_episode_steps;
  d.w_dist = cfg->distance_weight; d.w_energy = cfg->energy_weight; d.w_drift = cfg->drift_weight; d.w_shake = cfg->shake_weight;
  d.fwd_cap = cfg->forward_reward_cap;
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
    d.action_repeat = d.max_repeat;          // per-task values replace these inside the mixed kernel (wave by wave)
    d.iterations = d.max_iterations = 60;
  }
  d.trace = nullptr;
  d.slot_env = nullptr; d.block_task = nullptr; s->d_slot_env = nullptr; s->d_block_task = nullptr; s->mixed_blocks = 0; s->d_class = nullptr;
  d.mass_lo = cfg->mass_scale_lo; d.mass_hi = cfg->mass_scale_hi; d.mu_lo = cfg->friction_lo; d.mu_hi = cfg->friction_hi;
  s->words = rex_state_words(cfg);
  d.pose_index = cfg->pose_index; d.pose_value = cfg->pose_value;
  d.range_normalize = cfg->range_normalize;
  d.terrain = nullptr; d.terrain_mid = nullptr; d.n_terrain = 0; d.body_params = nullptr;
  d.perm = nullptr; d.sweeps = nullptr; s->d_perm = nullptr; s->d_sweeps = nullptr; s->d_regroup = nullptr; d.clock = nullptr; s->d_clock = nullptr; s->h_clock = nullptr;
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
  d.dt_d = as_written(cfg->sim_time_step);
  d.gait_clock_d = cfg->gait_clock_scale > 0.0f ? as_written(cfg->gait_clock_scale) : 1.0;
  {
    // Regrouping (REX_REGROUP=1 / 0 overrides) can pay once a SIMD runs several waves one after the other -- below that the
    // launch ends with its slowest wave whatever the grouping.  Measured on MI355X: round 2, one-workgroup sort
    // (profiles/r02_regroup.md): +10 % on the walking workload (gait clock 1.5, sweep counts 0.96 correlated from step to
    // step), -11 % on the falling one (0.52).  Round 3, many-workgroup sort, falling workload: the step kernel -9 % at
    // 262 144 envs and -5 % at 65 536; with the sort's two launches +5.4 % (132.9 -> 140.1 M env-steps/s) and +0.9 % on
    // the whole step.  On by default from 262 144 envs.
    const char* ov = getenv("REX_REGROUP");
    const bool want = (ov ? atoi(ov) != 0 : cfg->num_envs >= 262144) && cfg->task != REX_TASK_MIXED;   // (a mixed batch is placed by its task-sorted slot map)
    if (want) {
      hipError_t e2 = hipMalloc(&s->d_perm, sizeof(int32_t) * (size_t)cfg->num_envs);
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_sweeps, sizeof(int32_t) * (size_t)cfg->num_envs);
      const int chunks = (cfg->num_envs + REX_REGROUP_CHUNK - 1) / REX_REGROUP_CHUNK;
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_regroup, sizeof(int32_t) * ((size_t)chunks * 64 + 1));
      if (e2 == hipSuccess) e2 = hipMemsetAsync(s->d_regroup, 0, sizeof(int32_t) * ((size_t)chunks * 64 + 1), (hipStream_t)stream);
      if (e2 != hipSuccess) { (void)rex_destroy(s); return fail(REX_ENOMEM, "hipMalloc(regroup): %s", hipGetErrorString(e2)); }   // (rex_destroy frees whatever was allocated: RexSim starts zeroed)
      hipLaunchKernelGGL(rex::rex_iota_kernel, dim3((cfg->num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, cfg->num_envs, s->d_perm, s->d_sweeps);
      d.perm = s->d_perm; d.sweeps = s->d_sweeps;
    }
  }
  if (cfg->task == REX_TASK_MIXED) {
    std::vector<int32_t> slots, tasks;
    int busy = task_slot_map(d, s->epw, slots, tasks);
    // up to one workgroup per SIMD (MI355X: 256 CUs x 4 = 1 024) the launch is one wave per SIMD; the padding of the map must not
    // push it into a second round
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const int one_round = 4 * (cus > 0 ? cus : 256);
    const bool pinned = getenv("REX_ENVS_PER_WAVE") != nullptr;
    // (the envs per wave decide the arithmetic -- the variants agree to rounding, not bit for bit --, so they are a function of the CONFIG alone:
    //  one round of the machine this library is written for, MI355X's 1 024 SIMDs, whatever partition or device it runs on; the device's own
    //  CU count only enters the regrouping decision below, which leaves results unchanged)
    const int one_round_mi355x = 1024;
    while (!pinned && s->epw < 16 && busy > one_round_mi355x) { s->epw *= 2; busy = task_slot_map(d, s->epw, slots, tasks); }
    // More workgroups than the machine holds at once (one wave per SIMD): the batch is regrouped by sweep counts every
    // step (REX_REGROUP=0 / 1 overrides) -- regions of whole waves per task, sorted inside.  Measured at 16 384 mark-arm envs,
    // 16 envs per wave: the padding of the chunked map is a second round of full-length waves; sorted, the late waves are the short ones.
    const char* ov = getenv("REX_REGROUP");
    const bool regroup = ov ? atoi(ov) != 0 : busy > one_round;
    std::vector<int32_t> cls;
    if (regroup) task_region_map(d, s->epw, slots, tasks, cls, s->mix_regions.base);
    s->mixed_blocks = (int)tasks.size();
    hipError_t e2 = hipMalloc(&s->d_slot_env, sizeof(int32_t) * slots.size());
    if (regroup) {
      const int chunks = (cfg->num_envs + REX_REGROUP_CHUNK - 1) / REX_REGROUP_CHUNK;
      s->mix_regions.n_mix = d.n_mix; s->mix_regions.bins_per_task = 64 / d.n_mix;
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_class, sizeof(int32_t) * cls.size());
      if (e2 == hipSuccess) e2 = hipMemcpy(s->d_class, cls.data(), sizeof(int32_t) * cls.size(), hipMemcpyHostToDevice);
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_sweeps, sizeof(int32_t) * (size_t)cfg->num_envs);
      if (e2 == hipSuccess) e2 = hipMemsetAsync(s->d_sweeps, 0, sizeof(int32_t) * (size_t)cfg->num_envs, (hipStream_t)stream);   // (on the caller's stream, like the
      //  single-task path: ordered ahead of the first rex_regroup_mixed_count_kernel also when that stream is non-blocking; the
      //  hipStreamSynchronize at the end of rex_create covers them.  The hipMemcpy calls below block until the data is on the device.)
      if (e2 == hipSuccess) e2 = hipMalloc(&s->d_regroup, sizeof(int32_t) * ((size_t)chunks * 64 + 1));
      if (e2 == hipSuccess) e2 = hipMemsetAsync(s->d_regroup, 0, sizeof(int32_t) * ((size_t)chunks * 64 + 1), (hipStream_t)stream);
      d.sweeps = s->d_sweeps;
    }
    if (e2 == hipSuccess) e2 = hipMalloc(&s->d_block_task, sizeof(int32_t) * tasks.size());
    if (e2 == hipSuccess) e2 = hipMemcpy(s->d_slot_env, slots.data(), sizeof(int32_t) * slots.size(), hipMemcpyHostToDevice);
    if (e2 == hipSuccess) e2 = hipMemcpy(s->d_block_task, tasks.data(), sizeof(int32_t) * tasks.size(), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) { (void)rex_destroy(s); return fail(REX_ENOMEM, "task slot map: %s", hipGetErrorString(e2)); }
    d.slot_env = s->d_slot_env; d.block_task = s->d_block_task;
  }
  hipError_t e = hipMalloc(&s->d_snap, sizeof(float) * snapshot_floats(s, d.n_mix));
  if (e != hipSuccess) { (void)rex_destroy(s); return fail(REX_ENOMEM, "hipMalloc(snapshot): %s", hipGetErrorString(e)); }
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
    (void)rex_destroy(s);
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
  if (!s || k < 0 || k >= 32768 || (k > 0 && (!d_heights || !d_mids))) return fail(REX_EINVAL, "rex_set_terrain: bad arguments (0 <= k < 32768 fields)%s", "");
  s->dev.geo = rex::HfGeom{256, 20.0f, 20.0f, 127.5f, 127.5f, 254.999f, 254.999f};   /* 256 x 256 vertices, 5 cm cells, centred */
  s->dev.hf_stride = 65536;
  return install_terrain(s, d_heights, d_mids, k, stream);
}

int rex_set_heightfield(RexSim* s, const float* d_heights, const float* d_mids, int k, int nx, int ny, float cell_x, float cell_y,
                        float origin_x, float origin_y, void* stream) {
  if (!s || k <= 0 || !d_heights || !d_mids || nx < 2 || ny < 2 || !(cell_x > 0.0f) || !(cell_y > 0.0f) || (long long)nx * ny > (1ll << 26))
    return fail(REX_EINVAL, "rex_set_heightfield: bad arguments%s", "");
  if ((long long)k * nx * ny >= (1ll << 31)) return fail(REX_EINVAL, "rex_set_heightfield: the pool (k fields of nx * ny heights) must stay below 2^31 floats%s", "");
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

int rex_set_event_trace(RexSim* s, uint32_t* d_trace) {
  if (!s) return fail(REX_EINVAL, "rex_set_event_trace: null sim%s", "");
  s->dev.trace = d_trace;
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
  free(s->h_clock);
  if (s->d_perm) (void)hipFree(s->d_perm);
  if (s->d_sweeps) (void)hipFree(s->d_sweeps);
  if (s->d_regroup) (void)hipFree(s->d_regroup);
  if (s->d_slot_env) (void)hipFree(s->d_slot_env);
  if (s->d_block_task) (void)hipFree(s->d_block_task);
  if (s->d_class) (void)hipFree(s->d_class);
  if (s->d_polbuf) (void)hipFree(s->d_polbuf);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
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

static int step_launch(RexSim* s, int num_steps, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream);

int rex_step(RexSim* s, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream) {
  if (!s || !d_action || !d_obs || !d_reward || !d_done) return fail(REX_EINVAL, "rex_step: null pointer%s", "");
  return step_launch(s, 1, d_action, d_obs, d_reward, d_done, d_motor_cmd, stream);
}

int rex_step_segment(RexSim* s, int num_steps, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream) {
  if (!s || !d_action || !d_obs || !d_reward || !d_done) return fail(REX_EINVAL, "rex_step_segment: null pointer%s", "");
  if (num_steps < 1) return fail(REX_EINVAL, "rex_step_segment: num_steps must be at least 1%s", "");
  const long long width = std::max(std::max(rex_action_dim(&s->cfg), rex_obs_dim(&s->cfg)), rex_num_motors(&s->cfg));
  if ((long long)num_steps * s->cfg.num_envs * width >= (1ll << 31))
    return fail(REX_EINVAL, "rex_step_segment: a segment this long exceeds the 32-bit element offsets of its blocks (num_steps * num_envs * row width < 2^31)%s", "");
  return step_launch(s, num_steps, d_action, d_obs, d_reward, d_done, d_motor_cmd, stream);
}

// ---- the actor inside the launch (rex_policy.h) ----
static int rows_floats_per_env(const RexSim* s) {   // the contact-row region of the step kernel's LDS, per env of a wave: the actor's scratch
  const bool arm = s->cfg.mark == REX_MARK_ARM;
  const int legf4 = REX_LEG_F4_OF(s->epw, arm, false);
  return 4 * (arm ? REX_LDS_F4_PER_ENV_ARM_OF(s->epw) : REX_ROWS_F4_OF(legf4));
}

int rex_set_policy(RexSim* s, const RexPolicy* p, void* stream) {
  if (!s) return fail(REX_EINVAL, "rex_set_policy: null sim%s", "");
  if (!p) { s->have_policy = 0; return REX_OK; }
  if (s->cfg.task == REX_TASK_MIXED || s->cfg.body_contacts || s->epw > 16)
    return fail(REX_EINVAL, "rex_set_policy: the fused actor runs in the single-task lane-group kernels (not REX_TASK_MIXED, body_contacts = 0, REX_ENVS_PER_WAVE <= 16)%s", "");
  if (!s->cfg.range_normalize)
    return fail(REX_EINVAL, "rex_set_policy: the sim must fold the reference's wrapper stack (RexConfig.range_normalize = 1): the agents act through "
                            "RangeNormalize + ClipAction (playground/trainer.py:48-52)%s", "");
  if (p->obs_dim != rex_obs_dim(&s->cfg) || p->action_dim != rex_action_dim(&s->cfg))
    return failf(REX_EINVAL, "rex_set_policy: obs_dim / action_dim %d / %d do not match the sim's %d / %d", p->obs_dim, p->action_dim, rex_obs_dim(&s->cfg), rex_action_dim(&s->cfg));
  if (p->hidden1 < 1 || p->hidden2 < 1 || p->hidden1 > 4096 || p->hidden2 > 4096 || rex::policy_scratch_floats(p->obs_dim, p->hidden1, p->hidden2) > rows_floats_per_env(s))
    return failf(REX_EINVAL, "rex_set_policy: hidden layers of %d and %d units need %d floats of LDS per env, this kernel variant has %d", p->hidden1, p->hidden2,
                 rex::policy_scratch_floats(p->obs_dim, p->hidden1 > 0 ? p->hidden1 : 0, p->hidden2 > 0 ? p->hidden2 : 0), rows_floats_per_env(s));
  if (!p->d_w1 || !p->d_b1 || !p->d_w2 || !p->d_b2 || !p->d_w3 || !p->d_b3 || !p->d_logstd || (!p->d_obs_mean) != (!p->d_obs_scale))
    return fail(REX_EINVAL, "rex_set_policy: null weight pointer (d_obs_mean and d_obs_scale go together)%s", "");
  if (!(p->obs_clip > 0.0f) && p->d_obs_mean) return fail(REX_EINVAL, "rex_set_policy: obs_clip must be positive%s", "");
  HIPCHK(hipSetDevice(s->device));
  const rex::PolOff off = rex::policy_offsets(p->obs_dim, p->action_dim, p->hidden1, p->hidden2);
  if (off.total > s->polbuf_floats) {            // (a launch that is still reading the old buffer: the free below waits for the device)
    if (s->d_polbuf) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(s->d_polbuf); s->d_polbuf = nullptr; s->polbuf_floats = 0; }
    if (hipMalloc(&s->d_polbuf, sizeof(float) * (size_t)off.total) != hipSuccess) return fail(REX_ENOMEM, "rex_set_policy: hipMalloc%s", "");
    s->polbuf_floats = off.total;
  }
  // pack (stream-ordered: behind the launches that read the previous contents, ahead of those that follow): input-major matrices ->
  // [k / 4][unit][4], every block 16-byte aligned, zero padding (rex_policy.h)
  rex::PolSrc src{p->d_w1, p->d_b1, p->d_w2, p->d_b2, p->d_w3, p->d_b3, p->d_logstd, p->d_obs_mean, p->d_obs_scale};
  hipLaunchKernelGGL(rex::rex_pack_policy_kernel, dim3((off.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, p->obs_dim, p->action_dim, p->hidden1,
                     p->hidden2, s->d_polbuf);
  HIPCHK(hipGetLastError());
  rex::PolDev& d = s->pol;
  d.pk = s->d_polbuf; d.obs_in = nullptr; d.action_out = nullptr; d.mean_out = nullptr;
  d.h1 = p->hidden1; d.h2 = p->hidden2; d.obs_clip = p->d_obs_mean ? p->obs_clip : 0.0f; d.sample = p->sample ? 1 : 0;
  d.seed_lo = (uint32_t)p->seed; d.seed_hi = (uint32_t)(p->seed >> 32);
  {   // one copy of the packed actor per four-wave workgroup, in dynamic LDS, if it fits next to the waves' rows (MI355X: 160 KB per CU)
    const bool arm = s->cfg.mark == REX_MARK_ARM;
    const int wave_bytes = (rows_floats_per_env(s) / 4 + REX_PARK_F4_OF(s->epw, arm)) * s->epw * 16;
    const int want = off.total * 4;
    const char* ov = getenv("REX_POLICY_LDS");      // developer A/B runs: 0 streams the weights from L2 everywhere
    const bool fits = REX_POLICY_WAVES(s->epw) > 1 && REX_POLICY_WAVES(s->epw) * wave_bytes + want <= 160 * 1024 && !(ov && atoi(ov) == 0);
    s->pol_lds_bytes = fits ? want : 0;
    d.in_lds = fits ? 1 : 0;
  }
  s->have_policy = 1;
  return REX_OK;
}

static int policy_launch(RexSim* s, const char* who, int num_steps, const float* d_obs_in, float* d_action, float* d_mean, float* d_obs, float* d_reward,
                         uint8_t* d_done, float* d_motor_cmd, void* stream) {
  if (!s || !d_obs_in || !d_action || !d_obs || !d_reward || !d_done) return fail(REX_EINVAL, "%s: null pointer", who);
  if (!s->have_policy) return fail(REX_EINVAL, "%s: no policy set (rex_set_policy)", who);
  if (s->dev.trace) return fail(REX_EINVAL, "%s: not available while an event trace is set (rex_set_event_trace)", who);
  if (num_steps < 1) return fail(REX_EINVAL, "%s: num_steps must be at least 1", who);
  if (d_obs_in == d_obs) return fail(REX_EINVAL, "%s: d_obs must not alias d_obs_in", who);
  const long long width = std::max(std::max(rex_action_dim(&s->cfg), rex_obs_dim(&s->cfg)), rex_num_motors(&s->cfg));
  if ((long long)num_steps * s->cfg.num_envs * width >= (1ll << 31))
    return fail(REX_EINVAL, "%s: a segment this long exceeds the 32-bit element offsets of its blocks (num_steps * num_envs * row width < 2^31)", who);
  s->pol.obs_in = d_obs_in; s->pol.action_out = d_action; s->pol.mean_out = d_mean;
  return step_launch(s, -num_steps, nullptr, d_obs, d_reward, d_done, d_motor_cmd, stream);   // (negative: the fused-actor kernels)
}
int rex_step_policy(RexSim* s, const float* d_obs_in, float* d_action, float* d_mean, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream) {
  return policy_launch(s, "rex_step_policy", 1, d_obs_in, d_action, d_mean, d_obs, d_reward, d_done, d_motor_cmd, stream);
}
int rex_step_segment_policy(RexSim* s, int num_steps, const float* d_obs_in, float* d_action, float* d_mean, float* d_obs, float* d_reward, uint8_t* d_done,
                            float* d_motor_cmd, void* stream) {
  return policy_launch(s, "rex_step_segment_policy", num_steps, d_obs_in, d_action, d_mean, d_obs, d_reward, d_done, d_motor_cmd, stream);
}

static int step_launch(RexSim* s, int num_steps, const float* d_action, float* d_obs, float* d_reward, uint8_t* d_done, float* d_motor_cmd, void* stream) {
  HIPCHK(hipSetDevice(s->device));
  s->use_policy = num_steps < 0;
  if (num_steps < 0) num_steps = -num_steps;
  s->dev.nsteps = num_steps;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = s->d_slot_env ? s->mixed_blocks : (s->cfg.num_envs + s->epw - 1) / s->epw;
  hipEvent_t e0 = s->ev0, e1 = s->ev1;
  if (s->timing == 2) {
    const int k = (int)(s->timed_steps % REX_TIMING_RING);
    if (!s->ring0[k]) { HIPCHK(hipEventCreate(&s->ring0[k])); HIPCHK(hipEventCreate(&s->ring1[k])); }
    e0 = s->ring0[k]; e1 = s->ring1[k];
  }
  if (s->timing == 3 && s->timed_steps < REX_CLOCK_SLOTS) {   // device-side timestamps: this launch's (min start, max end) slot
    s->dev.clock = s->d_clock + 2 * REX_CLOCK_WAYS * s->timed_steps;   // (all slots were primed by rex_set_timing: nothing is copied per
    s->timed_steps++;                                         // launch, the queue stays as full as in an untimed run)
  } else s->dev.clock = nullptr;
  if (s->timing == 1 || s->timing == 2) HIPCHK(hipEventRecord(e0, st));
  launch_step(s, blocks, st, d_action, d_obs, d_reward, d_done, d_motor_cmd);
  {   // developer probe (tools/launch_gap.py): REX_STEP_REPEAT=R issues the launch R times back to back from C
    static const int repeat = getenv("REX_STEP_REPEAT") ? atoi(getenv("REX_STEP_REPEAT")) : 1;
    for (int k = 1; k < repeat; ++k) launch_step(s, blocks, st, d_action, d_obs, d_reward, d_done, d_motor_cmd);
  }
  HIPCHK(hipGetLastError());
  // (the timed span is the step kernel's: the regrouping launches below are not part of it)
  if (s->timing == 1 || s->timing == 2) { HIPCHK(hipEventRecord(e1, st)); s->have_timing = 1; if (s->timing == 2) s->timed_steps++; }
  if (s->d_class) {   // a regrouped mixed batch: every task's region sorted by this step's sweep counts for the next step
    const int n = s->cfg.num_envs, width = (s->dev.max_repeat * s->dev.max_iterations + s->mix_regions.bins_per_task - 1) / s->mix_regions.bins_per_task;
    const int chunks = (n + REX_REGROUP_CHUNK - 1) / REX_REGROUP_CHUNK;
    unsigned* done = reinterpret_cast<unsigned*>(s->d_regroup + (size_t)chunks * 64);
    hipLaunchKernelGGL(rex::rex_regroup_mixed_count_kernel, dim3(chunks), dim3(256), 0, st, n, width, s->mix_regions, s->d_class, s->d_sweeps, s->d_regroup, done);
    hipLaunchKernelGGL(rex::rex_regroup_mixed_scatter_kernel, dim3(chunks), dim3(256), 0, st, n, width, s->mix_regions, s->d_class, s->d_sweeps, s->d_regroup, s->d_slot_env);
  }
  if (s->d_perm) {   // next step's grouping from this step's sweep counts (stream-ordered behind the step)
    const int n = s->cfg.num_envs, width = (s->dev.max_repeat * s->dev.max_iterations + 63) / 64;
    const int chunks = (n + REX_REGROUP_CHUNK - 1) / REX_REGROUP_CHUNK;
    if (chunks <= 4) hipLaunchKernelGGL(rex::rex_regroup_kernel, dim3(1), dim3(1024), 0, st, n, width, s->d_sweeps, s->d_perm);
    else {
      unsigned* done = reinterpret_cast<unsigned*>(s->d_regroup + (size_t)chunks * 64);
      hipLaunchKernelGGL(rex::rex_regroup_count_kernel, dim3(chunks), dim3(256), 0, st, n, width, s->d_sweeps, s->d_regroup, done);
      hipLaunchKernelGGL(rex::rex_regroup_scatter_kernel, dim3(chunks), dim3(256), 0, st, n, width, s->d_sweeps, s->d_regroup, s->d_perm);
    }
  }
  return REX_OK;
}

int rex_mixed_slot_map(const RexConfig* cfg, int envs_per_wave, int32_t* slots, int32_t* tasks, int max_blocks) {
  if (validate(cfg)) return REX_EINVAL;
  if (cfg->task != REX_TASK_MIXED || (envs_per_wave != 4 && envs_per_wave != 8 && envs_per_wave != 16))
    return fail(REX_EINVAL, "rex_mixed_slot_map: a REX_TASK_MIXED config and 4, 8 or 16 envs per wave%s", "");
  rex::DevCfg d;
  memset(&d, 0, sizeof d);
  d.n = cfg->num_envs; d.env_index_base = cfg->env_index_base; d.seed_lo = (uint32_t)cfg->seed; d.seed_hi = (uint32_t)(cfg->seed >> 32);
  int ts[5];
  d.n_mix = mix_tasks(cfg, ts);
  for (int k = 0; k < 5; ++k) d.mix_task[k] = ts[k < d.n_mix ? k : 0];
  std::vector<int32_t> sl, tk;
  task_slot_map(d, envs_per_wave, sl, tk);
  if (slots && tasks) {
    if ((int)tk.size() > max_blocks) return fail(REX_EINVAL, "rex_mixed_slot_map: buffers too small%s", "");
    memcpy(slots, sl.data(), sizeof(int32_t) * sl.size());
    memcpy(tasks, tk.data(), sizeof(int32_t) * tk.size());
  }
  return (int)tk.size();
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
  if (s->timing == 3) {   // the next REX_CLOCK_SLOTS launches are timed on the device; prime their (min, max) slots
    HIPCHK(hipSetDevice(s->device));
    const size_t words = (size_t)2 * REX_CLOCK_WAYS * REX_CLOCK_SLOTS;
    if (!s->d_clock) HIPCHK(hipMalloc(&s->d_clock, sizeof(unsigned long long) * words));
    if (!s->h_clock) s->h_clock = (unsigned long long*)malloc(sizeof(unsigned long long) * words);
    if (!s->h_clock) return fail(REX_ENOMEM, "rex_set_timing: host buffer%s", "");
    for (size_t k = 0; k < words / 2; ++k) { s->h_clock[2 * k] = ~0ull; s->h_clock[2 * k + 1] = 0ull; }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(s->d_clock, s->h_clock, sizeof(unsigned long long) * words, hipMemcpyHostToDevice));
  }
  return REX_OK;
}

int rex_step_times_ms(RexSim* s, float* ms, int max_count) {
  if (!s || !ms || max_count <= 0) return fail(REX_EINVAL, "rex_step_times_ms: bad arguments%s", "");
  if ((s->timing != 2 && s->timing != 3) || s->timed_steps == 0) return 0;
  const long long cap = s->timing == 3 ? REX_CLOCK_SLOTS : REX_TIMING_RING;
  long long have = s->timed_steps < cap ? s->timed_steps : cap;
  int n = (int)(have < max_count ? have : max_count);
  HIPCHK(hipSetDevice(s->device));
  if (s->timing == 3) {
    unsigned long long* ticks = s->h_clock;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(ticks, s->d_clock, sizeof(unsigned long long) * 2 * REX_CLOCK_WAYS * REX_CLOCK_SLOTS, hipMemcpyDeviceToHost));
    int khz = 100000;   // s_memrealtime: constant 100 MHz on gfx9
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device);
    for (int j = 0; j < n; ++j) {
      const unsigned long long* t = ticks + (size_t)2 * REX_CLOCK_WAYS * (size_t)(s->timed_steps - n + j);
      unsigned long long first = ~0ull, last = 0ull;       // first wave start / last wave end over the launch's tick pairs
      for (int w = 0; w < REX_CLOCK_WAYS; ++w) { if (t[2 * w] < first) first = t[2 * w]; if (t[2 * w + 1] > last) last = t[2 * w + 1]; }
      ms[j] = last >= first ? (float)((double)(last - first) / (double)khz) : 0.0f;
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

int rex_gait_loop(int n, int mode, double* d_planner, const double* d_params, float* d_frames_out, void* stream) {
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
REX_API int rex_debug_legmask(unsigned* out, int n) {   /* per env: toe points in reach (last substep; OR since the last call) */
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_legmask), sizeof(unsigned) * (size_t)(n < 65536 ? n : 65536)) != hipSuccess) return REX_EHIP;
  static unsigned z[65536];
  if (hipMemcpyToSymbol(HIP_SYMBOL(rex::g_legmask), z, sizeof(z)) != hipSuccess) return REX_EHIP;
  return REX_OK;
}
REX_API int rex_debug_prof2(long long* out, int reset) {   /* the sections of the sweep routine (pgs_dv) */
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rex::g_prof2), sizeof(long long) * 16 * 1024) != hipSuccess) return REX_EHIP;
  if (reset) { static long long z[16 * 1024]; if (hipMemcpyToSymbol(HIP_SYMBOL(rex::g_prof2), z, sizeof(z)) != hipSuccess) return REX_EHIP; }
  return REX_OK;
}
#endif
}  // extern "C"

static void launch_step(RexSim* s, int blocks, hipStream_t st, const float* a, float* o, float* r, uint8_t* d, float* m) {
  const bool arm = s->cfg.mark == REX_MARK_ARM;
  if (s->dev.trace && s->dev.nsteps > 1) {   // a segment under the event trace (debug runs): step by step through the _trace kernels
    const int T = s->dev.nsteps, n = s->cfg.num_envs;
    const int ad = rex_action_dim(&s->cfg), od = rex_obs_dim(&s->cfg), nm = rex_num_motors(&s->cfg);
    s->dev.nsteps = 1;
    for (int t = 0; t < T; ++t)
      launch_step(s, blocks, st, a + (size_t)t * n * ad, o + (size_t)t * n * od, r + (size_t)t * n, d + (size_t)t * n, m ? m + (size_t)t * n * nm : nullptr);
    s->dev.nsteps = T;
    return;
  }
  if (s->dev.trace) {   // rex_set_event_trace: the instantiations with the event trace compiled in (debug runs)
    if (s->cfg.task == REX_TASK_MIXED) { if (arm) rex_launch_step_mixed_arm_trace(s, blocks, st, a, o, r, d, m); else rex_launch_step_mixed_base_trace(s, blocks, st, a, o, r, d, m); }
    else if (s->cfg.body_contacts) rex_launch_step_body_trace(s, blocks, st, a, o, r, d, m);
    else if (arm) rex_launch_step_arm_trace(s, blocks, st, a, o, r, d, m);
    else rex_launch_step_base_trace(s, blocks, st, a, o, r, d, m);
    return;
  }
  if (s->use_policy) {       // rex_step_policy / rex_step_segment_policy: the segment kernels with the actor in front of every step
    if (arm) rex_launch_step_arm_pol(s, blocks, st, a, o, r, d, m); else rex_launch_step_base_pol(s, blocks, st, a, o, r, d, m);
    return;
  }
  if (s->dev.nsteps > 1) {   // rex_step_segment: the instantiations with the loop over the segment's steps
    if (s->cfg.task == REX_TASK_MIXED) { if (arm) rex_launch_step_mixed_arm_seg(s, blocks, st, a, o, r, d, m); else rex_launch_step_mixed_base_seg(s, blocks, st, a, o, r, d, m); }
    else if (s->cfg.body_contacts) rex_launch_step_body_seg(s, blocks, st, a, o, r, d, m);
    else if (arm) rex_launch_step_arm_seg(s, blocks, st, a, o, r, d, m);
    else rex_launch_step_base_seg(s, blocks, st, a, o, r, d, m);
    return;
  }
  if (s->cfg.task == REX_TASK_MIXED) {   // lane groups only (rex_create caps the envs per wave at 16)
    if (arm) rex_launch_step_mixed_arm(s, blocks, st, a, o, r, d, m); else rex_launch_step_mixed_base(s, blocks, st, a, o, r, d, m);
  } else if (s->cfg.body_contacts) rex_launch_step_body(s, blocks, st, a, o, r, d, m);   // link-box contact rows: 4 or 8 envs per wave (rex_create caps it)
  else if (arm) rex_launch_step_arm(s, blocks, st, a, o, r, d, m);
  else rex_launch_step_base(s, blocks, st, a, o, r, d, m);
}
static void launch_settle(RexSim* s, int nrec, hipStream_t st, float* snap) {
  if (s->cfg.mark == REX_MARK_ARM) rex_launch_settle_arm(s, nrec, st, snap); else rex_launch_settle_base(s, nrec, st, snap);
}
