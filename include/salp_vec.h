/*
 * salp_vec.h — C ABI of the batched SALP swimmer simulator for AMD MI355X (gfx950).
 *
 * One handle (`salp_vec_t`) owns the struct-of-arrays state of `n_envs` independent
 * SalpSnakeEnv instances on ONE GPU and advances all of them with one fused HIP kernel
 * per call.  The entry points are what a binding of the reference's Gymnasium `Env`
 * surface needs — the reference has no FFI of its own, so each one cites the reference
 * method it replaces (paths relative to the reference repo; "legacy" =
 * scripts/utilities/salp_robot.py, "snake" = src/salp/environments/salp_snake_env.py):
 *
 *   salp_config_default     snake:29-33 (the 13 constructor kwargs) + legacy:32-53 (constants)
 *   salp_vec_create         SalpSnakeEnv.__init__            snake:29-90
 *   salp_vec_reset          SalpSnakeEnv.reset               snake:133-155, legacy:95-117
 *   salp_vec_step           SalpSnakeEnv.step                snake:157-202, legacy:119-156
 *   salp_vec_rollout        the caller's `for t in range(T): env.step(a[t])` loop
 *                           (train.py:110-122, eval/collect_navigation_data.py:97-114)
 *   salp_vec_get/set_state  attribute pokes `env.robot_pos = …`, `env.food_positions = …`
 *                           (eval/collect_navigation_data.py:76-89) and legacy:390-403 (_get_info)
 *   salp_vec_observe        SalpSnakeEnv._get_extended_observation   snake:366-428
 *
 * Conventions
 *   - every function returns 0 (SALP_OK) or a negative salp_status; the message for the
 *     calling thread's last failure is `salp_last_error()`.  Nothing throws or aborts.
 *   - every data buffer is caller-owned.  `flags & SALP_DEVICE_PTRS` says the data
 *     pointers are device pointers on the handle's GPU; the call is then asynchronous on
 *     `stream` (a hipStream_t passed as void*, NULL = the null stream).  Without the flag the
 *     pointers are host pointers and the call is synchronous (H2D, kernel, D2H inside).
 *   - a handle is not thread-safe; distinct handles (one per GPU) are independent.
 *   - every call runs on the handle's device and leaves the caller's current HIP device as it found it.
 *   - there is no CPU fallback: creating a handle without a usable HIP device fails with
 *     SALP_ERR_NO_DEVICE.
 *
 * Randomness (the reference uses two global un-seeded Mersenne Twisters — snake:12 `random`,
 * legacy:311 `np.random` — so its draws are not reproducible; this library defines them):
 *   block(env, n) = Philox4x32-10(counter = (env_lo, env_hi, n, 0), key = (seed_lo, seed_hi))
 *   where env is the GLOBAL env index (env_index_base + local index) and n is that env's
 *   running draw counter (state row SALP_I_RNG_COUNTER).  Each draw EVENT consumes one block,
 *   in program order of the reference:
 *     thrust jitter  (legacy:311)                u = u53(w0,w1)
 *     one food-placement attempt (snake:101-104, 127-130, 239-242, 269-272)
 *                                                x = lo+(hi-lo)*u53(w0,w1), y likewise from (w2,w3)
 *     random food count (snake:146)              1 + ((w0 * n) >> 32)
 *   u53(a,b) = ((a>>5)*2^26 + (b>>6)) / 2^53.
 *   Device-generated actions (salp_vec_rollout with act == NULL): with t = the handle's global step
 *   count and j the action component,
 *     w = word (t & 3) of Philox4x32-10(counter = (env_lo, env_hi, t >> 2, 1 + j), key)
 *     a_j = (w >> 8) * 2^-23 - 1   in [-1, 1)      (nozzle direction)
 *     a_0 = (w >> 8) * 2^-24       in [0, 1)       (inhale control, 2-action mode only)
 *   (one block serves four consecutive steps of a component).
 */
#ifndef SALP_VEC_H
#define SALP_VEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SALP_ABI_VERSION 1
#define SALP_MAX_FOOD 16        /* num_food_items upper bound (sac_gail.yaml uses 12) */
#define SALP_MAX_OBSERVED_FOOD 8

typedef enum salp_status {
  SALP_OK = 0,
  SALP_ERR_INVALID = -1,    /* bad argument / config */
  SALP_ERR_NO_DEVICE = -2,  /* no HIP device, or device_id out of range */
  SALP_ERR_HIP = -3,        /* a HIP runtime call failed */
  SALP_ERR_OOM = -4
} salp_status;

enum { SALP_DEVICE_PTRS = 1u };

/* POD of the reference's parameters.  Units are the reference's (pixels, steps, radians). */
typedef struct salp_config {
  uint32_t struct_size;            /* = sizeof(salp_config_t); checked by salp_vec_create */
  /* snake:29-33 kwargs */
  int32_t width;                   /* 800 */
  int32_t height;                  /* 600 */
  int32_t num_food_items;          /* 5   (base_num_food_items, snake:36) */
  int32_t max_observed_food;       /* 3   (K; obs_dim = 10 + 4K + 2, snake:79-80) */
  int32_t max_steps_without_food;  /* 1500 */
  int32_t forced_breathing;        /* 1   (act_dim 1; 0 -> act_dim 2, snake:69-74) */
  int32_t random_food_count;       /* 0 */
  int32_t respawn_food;            /* 1 */
  double food_reward;              /* 10.0 */
  double collision_penalty;        /* -50.0 */
  double time_penalty;             /* -0.1 */
  double efficiency_bonus;         /* 1.0 */
  double proximity_reward_weight;  /* 0.0 */
  /* legacy:32-53 constants and snake:53-54 */
  double tank_margin;              /* 50 */
  double base_radius;              /* 30 */
  double max_thrust_force;         /* 100 */
  double drag_coefficient;         /* 0.98 */
  double angular_drag;             /* 0.95 (literal at legacy:320) */
  double max_nozzle_angle;         /* pi/3 */
  double nozzle_response_rate;     /* 0.05 */
  double food_radius;              /* 15 */
  double min_food_distance;        /* 80 */
  int32_t inhale_duration;         /* 120 */
  int32_t exhale_duration;         /* 150 */
  int32_t rest_duration;           /* 60 (enters only through the 330-step modulus, legacy:161) */
  int32_t no_autoreset;            /* 0: an env that terminates or truncates starts its next episode in the same step
                                    * (VectorEnv convention).  1: it is NOT reset, exactly like the reference's single env
                                    * when its caller ignores `done` and keeps stepping (eval/collect_navigation_data.py:
                                    * 97-114 rides through wall contacts this way); flags are still reported each step. */
} salp_config_t;

/* Rows of the state snapshot exchanged by salp_vec_get_state / salp_vec_set_state.
 * f64 block: [SALP_F_COUNT(F)][n_envs] doubles, row-major (one row per quantity).
 * i32 block: [SALP_I_COUNT][n_envs] int32.
 * A collected / absent food slot (`None` in snake:215) is NaN in both coordinates. */
enum {
  SALP_F_X = 0, SALP_F_Y, SALP_F_VX, SALP_F_VY, SALP_F_THETA, SALP_F_OMEGA,
  SALP_F_NOZZLE, SALP_F_WATER,
  SALP_F_ELLIPSE_A, SALP_F_ELLIPSE_B,   /* derived on get; ignored on set */
  SALP_F_FOOD0                          /* then food_x[0..F-1], food_y[0..F-1] */
};
#define SALP_F_COUNT(F) (SALP_F_FOOD0 + 2 * (F))
enum {
  SALP_I_PHASE = 0,        /* 0 rest, 1 inhaling, 2 exhaling (legacy:374) */
  SALP_I_TIMER,            /* breathing_timer */
  SALP_I_EXHALE_DUR,       /* current_exhale_duration (legacy:223) */
  SALP_I_SHAPE_HOLD,       /* 0 = ellipse follows (phase,timer,dur,water); 7 = post-reset circle
                              (a=b=base_radius, legacy:112-113); 1..6 = inhale timer at an early
                              release that returned to rest (legacy:225-228 keeps the old a,b) */
  SALP_I_STEPS_SINCE_FOOD,
  SALP_I_FOOD_COLLECTED,
  SALP_I_RNG_COUNTER,      /* next Philox block index of this env */
  SALP_I_EPISODE_LENGTH,
  SALP_I_COUNT
};

/* Per-step info columns (int32 [n_envs][SALP_INFO_COLS]); values are those of the step's own
 * (pre-autoreset) episode, as in the info dict of snake:195-200. */
enum { SALP_INFO_FOOD_COLLECTED = 0, SALP_INFO_STEPS_SINCE_FOOD, SALP_INFO_COLLISION, SALP_INFO_COLS };

/* Running totals since create / salp_vec_clear_stats, reduced on the device
 * (wave-shuffle + one atomic per wave; fixed-point so the sum is order-independent). */
typedef struct salp_stats {
  int64_t env_steps;          /* env-steps simulated */
  int64_t episodes;           /* episodes finished (terminated | truncated) */
  int64_t terminated;
  int64_t truncated;
  int64_t collisions;
  int64_t food_collected;
  int64_t episode_length_sum; /* over finished episodes */
  double reward_sum;          /* over all env-steps (accumulated in 2^-20 fixed point) */
  double episode_return_sum;  /* over finished episodes (2^-20 fixed point) */
} salp_stats_t;

typedef struct salp_vec salp_vec_t;

const char* salp_last_error(void);
int salp_abi_version(void);
int salp_device_count(void);

/* Fills *cfg with the reference defaults (snake:29-33, legacy:32-53). */
int salp_config_default(salp_config_t* cfg);

/* device_id: HIP ordinal.  env_index_base: global index of local env 0 (multi-GPU sharding keeps
 * env i's trajectory independent of the number of shards).  The handle starts in the post-reset
 * state of every env (draw counters at 0 before that reset). */
int salp_vec_create(const salp_config_t* cfg, int64_t n_envs, int device_id, uint64_t seed,
                    int64_t env_index_base, salp_vec_t** out);
void salp_vec_destroy(salp_vec_t* h);

int64_t salp_vec_num_envs(const salp_vec_t* h);
int salp_vec_obs_dim(const salp_vec_t* h);   /* 10 + 4K + 2 */
int salp_vec_act_dim(const salp_vec_t* h);   /* 1 (forced breathing) or 2 */
int salp_vec_num_food(const salp_vec_t* h);  /* F */
int salp_vec_device(const salp_vec_t* h);

/* reset(): mask == NULL resets every env, else the envs with mask[i] != 0.
 * obs (may be NULL): float [n_envs][obs_dim]; rows of envs that were not reset are written with
 * their current observation. */
int salp_vec_reset(salp_vec_t* h, const uint8_t* mask, float* obs, uint32_t flags, void* stream);

/* step(): act float [n_envs][act_dim] (not clipped, as in the reference).
 * obs float [n_envs][obs_dim]; reward float [n_envs]; terminated, truncated uint8 [n_envs].
 * Autoreset is same-step: a finished env is reset inside the call and `obs` holds the first
 * observation of its next episode; its terminal observation goes to final_obs (float
 * [n_envs][obs_dim], rows of unfinished envs untouched) when that pointer is non-NULL.
 * info (may be NULL): int32 [n_envs][SALP_INFO_COLS]. */
int salp_vec_step(salp_vec_t* h, const float* act, float* obs, float* reward,
                  uint8_t* terminated, uint8_t* truncated, float* final_obs, int32_t* info,
                  uint32_t flags, void* stream);

/* rollout(): horizon steps in ONE kernel launch, state held in registers across steps.
 * act float [horizon][n_envs][act_dim], or NULL for device-generated U[-1,1) actions
 * (then act_out, if non-NULL, receives them).  obs float [horizon][n_envs][obs_dim];
 * reward float [horizon][n_envs]; terminated / truncated uint8 [horizon][n_envs].
 * Any of obs / reward / terminated / truncated may be NULL (not written).
 * final_obs as in step(), shaped [horizon][n_envs][obs_dim] (may be NULL). */
int salp_vec_rollout(salp_vec_t* h, const float* act, int32_t horizon, float* obs, float* reward,
                     uint8_t* terminated, uint8_t* truncated, float* final_obs, float* act_out,
                     uint32_t flags, void* stream);

/* Current observation of every env without stepping. obs float [n_envs][obs_dim]. */
int salp_vec_observe(salp_vec_t* h, float* obs, uint32_t flags, void* stream);

/* State snapshot (layout above).  f64: double [SALP_F_COUNT(F)][n_envs]; i32: int32
 * [SALP_I_COUNT][n_envs].  set_state ignores the derived ellipse rows. */
int salp_vec_get_state(salp_vec_t* h, double* f64, int32_t* i32, uint32_t flags, void* stream);
int salp_vec_set_state(salp_vec_t* h, const double* f64, const int32_t* i32, uint32_t flags,
                       void* stream);

/* Re-keys the draw streams and starts over: afterwards the handle is in exactly the state salp_vec_create(cfg, n, dev,
 * seed, base) returns — new key words, every draw counter at 0, every env freshly reset (snake:133-155 `reset(seed)`,
 * legacy:95-96), statistics and global step cleared; base_num_food_items keeps a value set by
 * salp_vec_set_base_num_food.  Nothing is freed or reallocated and no launch parameter changes (the kernels read the
 * key words from device memory), so device pointers stay valid and a hipGraph captured on this handle before the call
 * replays correctly after it.  Asynchronous on `stream` with device pointers.  obs (may be NULL): float [n_envs][obs_dim],
 * the first observations. */
int salp_vec_reseed(salp_vec_t* h, uint64_t seed, float* obs, uint32_t flags, void* stream);

/* Totals of everything issued so far: waits for the stream of the handle's most recent call (not for the device). */
int salp_vec_get_stats(salp_vec_t* h, salp_stats_t* out);
/* Stream-ordered behind the handle's most recent call; does not synchronise. */
int salp_vec_clear_stats(salp_vec_t* h);
int64_t salp_vec_global_step(const salp_vec_t* h);

/* Which kernel instantiation the handle's most recent step / rollout call ran (introspection for tests and profiles; no
 * reference counterpart): info[0] food slots of the kernel (1, 4, 8, 12, 16), [1] observed-food capacity (3, or 8 = the
 * generic instantiation), [2] 1 = the reference's constants compiled in as literals, [3] forced breathing, [4] the
 * output signature the kernel was compiled for: 1 = obs, reward, terminated, truncated and nothing else, 2 = those four plus
 * final_obs and / or info, 0 = some of the four is NULL (every store tested; always 0 for the generic instantiation), [5] 1 = actions drawn in the kernel,
 * [6] envs served by the unpredicated launch (whole wavefronts), [7] envs served by the predicated launch. */
int salp_vec_last_launch(const salp_vec_t* h, int64_t info[8]);
/* What that kernel (the unpredicated one when both were launched) holds per workgroup of 256 threads, from the runtime
 * (hipFuncGetAttributes, hipOccupancyMaxActiveBlocksPerMultiprocessor): info[0] registers per thread (VGPRs), [1] static LDS
 * bytes, [2] scratch (spill) bytes per thread, [3] workgroups resident per CU = wavefronts per SIMD.  The design's occupancy
 * claims (one food >= 4, 4 / 8 slots 4, 12 slots 3, 16 slots 2) are tested against it; no reference counterpart. */
int salp_vec_last_kernel_resources(const salp_vec_t* h, int32_t info[4]);

/* The curriculum's attribute poke `env.base_num_food_items = k` (src/salp/training/continuous_trainer.py:409-411;
 * src/salp/environments/salp_snake_env.py:36): the number of foods placed at every LATER reset of an env
 * (snake:144-148; with random_food_count the upper bound of the draw).  0 <= k <= the num_food_items the
 * handle was created with, which fixes the number of food slots (create with the curriculum's maximum and
 * lower it here).  Takes effect from the next call; envs mid-episode keep their foods. */
int salp_vec_set_base_num_food(salp_vec_t* h, int32_t k);
int32_t salp_vec_base_num_food(const salp_vec_t* h);

#ifdef __cplusplus
}
#endif
#endif /* SALP_VEC_H */
