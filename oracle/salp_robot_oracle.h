/*
 * salp_robot_oracle.h — CPU restatement of the reference's HEAD simulator (robot.py Robot/Nozzle under
 * salp_robot_env.py SalpRobotEnv); see salp_robot_oracle.c.  TEST INFRASTRUCTURE ONLY.
 */
#ifndef SALP_ROBOT_ORACLE_H
#define SALP_ROBOT_ORACLE_H
#include "../include/salp_robot.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct salp_robot_oracle salp_robot_oracle_t;
int salp_robot_oracle_create(const salp_robot_config_t* cfg, int64_t n_envs, uint64_t seed,
                             int64_t env_index_base, salp_robot_oracle_t** out);
void salp_robot_oracle_destroy(salp_robot_oracle_t* h);
int salp_robot_oracle_reset(salp_robot_oracle_t* h, const uint8_t* mask, float* obs);
/* one env step = one breathing cycle; reward in fp64; inner_steps (nullable) = Euler steps taken */
int salp_robot_oracle_step(salp_robot_oracle_t* h, const float* act, float* obs, double* reward,
                           uint8_t* terminated, uint8_t* truncated, float* final_obs, int32_t* inner_steps);
int salp_robot_oracle_get_state(salp_robot_oracle_t* h, double* state /* [SALP_R_COUNT][n] */);
#ifdef __cplusplus
}
#endif
#endif
