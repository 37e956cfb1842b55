/*
 * salp_robot.h — C ABI of the batched HEAD simulator: the reference's jet-propelled
 * rigid-body `Robot` (src/salp/environments/robot.py) under `SalpRobotEnv`
 * (src/salp/environments/salp_robot_env.py), one robot per GPU lane.
 *
 *   salp_robot_config_default   Robot.__init__ / Nozzle.__init__ arguments of train_robot.py:12-18,
 *                               SalpRobotEnv.__init__ (salp_robot_env.py:32-37)
 *   salp_robot_vec_create       make_env() (train_robot.py:10-22) x n_envs
 *   salp_robot_vec_reset        SalpRobotEnv.reset            salp_robot_env.py:98-128
 *   salp_robot_vec_step         SalpRobotEnv.step             salp_robot_env.py:139-201 — ONE env step is one
 *                               whole breathing cycle: Robot.set_control + step_through_cycle
 *                               (robot.py:335-358, 422-445), up to ~1450 Euler steps of dt = 0.01 s
 *
 * Actions are float32 [n][3] in the env's Box ([0,1], [0,1], [-1,1]): contraction / 0.06 m, coast time
 * / 10 s, nozzle yaw / (pi/2).  They are widened to fp64 before the rescale of salp_robot_env.py:129-137.
 * The reference does not clip them, and neither does this library, with one exception: the length of a
 * breathing cycle (refill + jet + coast, 14.5 s at most inside the Box) is cut at 14.6 s, and a non-finite
 * length runs no Euler step — one out-of-Box or inf action must not spin a wavefront of 64 robots for ever
 * (the reference would stall that one CPU env for the corresponding number of Euler steps).
 * Observation float32 [n][6]: x - target_x, y - target_y, body-frame vx, vy, yaw, yaw rate (:400-420).
 * Same conventions as salp_vec.h (status codes, SALP_DEVICE_PTRS, streams, same-step autoreset).
 * The target point of each episode (np.random.uniform, :247-250) comes from
 *   Philox4x32-10(counter = (env_lo, env_hi, episode#, 16), key = seed): x from u53(w0,w1), y from u53(w2,w3).
 */
#ifndef SALP_ROBOT_H
#define SALP_ROBOT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct salp_robot_config {
  uint32_t struct_size;
  int32_t width, height;          /* 900, 700 (salp_robot_env.py:32) */
  double tank_margin;             /* 50 */
  /* Robot(dry_mass, init_length, init_width, max_contraction, nozzle), train_robot.py:14-15 */
  double dry_mass;                /* 1.0 kg */
  double init_length;             /* 0.3 m */
  double init_width;              /* 0.15 m */
  double max_contraction;         /* 0.06 m */
  double density;                 /* 1000 kg/m^3 (set_environment) */
  double dt;                      /* 0.01 s (robot.py:214) */
  double drag_coefficient_min;    /* 0.4 (robot.py:222) */
  double drag_coefficient_max;    /* 1.0 */
  /* Nozzle(length1, length2, length3, area, mass), train_robot.py:12 */
  double nozzle_length1, nozzle_length2, nozzle_length3;   /* 0.05 each */
  double nozzle_area;             /* 0.00016 m^2 */
  double nozzle_mass;             /* 1.0 kg */
  double nozzle_gamma;            /* pi/4 (robot.py:31) */
  int32_t max_cycles;             /* 500 (salp_robot_env.py:183) */
  int32_t reserved0;
} salp_robot_config_t;

/* rows of the fp64 state snapshot [SALP_R_COUNT][n_envs] */
enum {
  SALP_R_POS = 0, SALP_R_VEL = 3, SALP_R_EULER = 6, SALP_R_OMEGA = 9, SALP_R_VEL_WORLD = 12, SALP_R_PREV_I = 15,
  SALP_R_TARGET = 18, SALP_R_PREV_DIST = 20, SALP_R_VOLUME = 21, SALP_R_ANGLE1 = 22, SALP_R_ANGLE2 = 23,
  SALP_R_TIME = 24, SALP_R_CYCLE = 25, SALP_R_RNG = 26, SALP_R_COUNT = 27
};

typedef struct salp_robot_vec salp_robot_vec_t;

const char* salp_robot_last_error(void);   /* message of the calling thread's last failed salp_robot_* call */
int salp_robot_config_default(salp_robot_config_t* cfg);
int salp_robot_vec_create(const salp_robot_config_t* cfg, int64_t n_envs, int device_id, uint64_t seed,
                          int64_t env_index_base, salp_robot_vec_t** out);
void salp_robot_vec_destroy(salp_robot_vec_t* h);
int64_t salp_robot_vec_num_envs(const salp_robot_vec_t* h);
int salp_robot_vec_reset(salp_robot_vec_t* h, const uint8_t* mask, float* obs, uint32_t flags, void* stream);
/* obs float [n][6]; reward float [n]; terminated/truncated uint8 [n]; final_obs (nullable) float [n][6]
 * rows of finished envs; inner_steps (nullable) int32 [n] = Euler steps of this cycle. */
int salp_robot_vec_step(salp_robot_vec_t* h, const float* act, float* obs, float* reward, uint8_t* terminated,
                        uint8_t* truncated, float* final_obs, int32_t* inner_steps, uint32_t flags, void* stream);
int salp_robot_vec_get_state(salp_robot_vec_t* h, double* state, uint32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif
