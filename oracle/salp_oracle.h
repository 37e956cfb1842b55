/*
 * salp_oracle.h — CPU restatement of the reference's SalpSnakeEnv (see salp_oracle.c).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg; never by the product.  Uses the public POD config and the state-row
 * enums of include/salp_vec.h so snapshots are interchangeable with the HIP library's.
 */
#ifndef SALP_ORACLE_H
#define SALP_ORACLE_H
#include "../include/salp_vec.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct salp_oracle salp_oracle_t;

void salp_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void salp_oracle_set_threads(int n);   /* OpenMP threads over envs; default 1 */
int salp_oracle_get_threads(void);

int salp_oracle_create(const salp_config_t* cfg, int64_t n_envs, uint64_t seed,
                       int64_t env_index_base, salp_oracle_t** out);
void salp_oracle_destroy(salp_oracle_t* h);
int salp_oracle_obs_dim(const salp_oracle_t* h);
int salp_oracle_act_dim(const salp_oracle_t* h);
int salp_oracle_reset(salp_oracle_t* h, const uint8_t* mask, float* obs);
int salp_oracle_observe(salp_oracle_t* h, float* obs);
/* reward64 (nullable) receives the un-rounded fp64 reward of the reference. */
int salp_oracle_step(salp_oracle_t* h, const float* act, float* obs, float* reward,
                     double* reward64, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                     int32_t* info);
int salp_oracle_rollout(salp_oracle_t* h, const float* act, int32_t horizon, float* obs,
                        float* reward, double* reward64, uint8_t* terminated, uint8_t* truncated,
                        float* final_obs, int32_t* info, float* act_out);
/* test-only variant with fp64 actions (the reference accepts any float; the product ABI is f32) */
int salp_oracle_rollout_f64(salp_oracle_t* h, const double* act64, int32_t horizon, float* obs,
                            double* reward64, uint8_t* terminated, uint8_t* truncated);
int salp_oracle_get_state(salp_oracle_t* h, double* f64, int32_t* i32);
int salp_oracle_set_state(salp_oracle_t* h, const double* f64, const int32_t* i32);
int64_t salp_oracle_global_step(const salp_oracle_t* h);
int salp_oracle_set_base_num_food(salp_oracle_t* h, int k);   /* snake:36 base_num_food_items, 0..num_food_items */
#ifdef __cplusplus
}
#endif
#endif
