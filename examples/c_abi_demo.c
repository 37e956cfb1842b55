/* The C ABI of include/salp_vec.h from plain C, host-pointer mode (flags = 0): what a non-Python consumer links.
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo \
 *       -Lunderwater-swimmer_rl_amd/csrc -lsalp_hip -Wl,-rpath,$PWD/underwater-swimmer_rl_amd/csrc -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
 * Needs an MI355X at run time (the library has no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "salp_vec.h"

int main(void) {
  enum { N = 4096, T = 300 };
  salp_config_t cfg;
  salp_vec_t* env = NULL;
  if (salp_config_default(&cfg) != 0) { fprintf(stderr, "%s\n", salp_last_error()); return 1; }
  cfg.num_food_items = 1;                      /* configs/single_food.yaml */
  cfg.max_steps_without_food = 1500;
  cfg.proximity_reward_weight = 5.0;
  if (salp_vec_create(&cfg, N, 0, 42u, 0, &env) != 0) { fprintf(stderr, "create: %s\n", salp_last_error()); return 1; }
  const int od = salp_vec_obs_dim(env), ad = salp_vec_act_dim(env);
  float* obs = malloc(sizeof(float) * N * od);
  float* act = calloc((size_t)N * ad, sizeof(float));
  float* rew = malloc(sizeof(float) * N);
  uint8_t* term = malloc(N);
  uint8_t* trunc = malloc(N);
  if (salp_vec_reset(env, NULL, obs, 0, NULL) != 0) { fprintf(stderr, "reset: %s\n", salp_last_error()); return 1; }
  printf("obs_dim %d act_dim %d; env 0 starts at (%.3f, %.3f) of the tank\n", od, ad, obs[0], obs[1]);
  double ret = 0.0;
  long done = 0;
  for (int t = 0; t < T; ++t) {
    for (int i = 0; i < N; ++i) act[i * ad] = (float)((i % 21) - 10) * 0.1f;   /* a fixed nozzle command per env */
    if (salp_vec_step(env, act, obs, rew, term, trunc, NULL, NULL, 0, NULL) != 0) { fprintf(stderr, "step: %s\n", salp_last_error()); return 1; }
    for (int i = 0; i < N; ++i) { ret += rew[i]; done += term[i] | trunc[i]; }
  }
  salp_stats_t st;
  salp_vec_get_stats(env, &st);
  printf("%d envs x %d steps: mean reward/step %.4f, episodes finished %ld (library counts %lld), food %lld, env-steps %lld\n",
         N, T, ret / ((double)N * T), done, (long long)st.episodes, (long long)st.food_collected, (long long)st.env_steps);
  const int ok = st.env_steps == (long long)N * T && st.episodes == done;
  salp_vec_destroy(env);
  free(obs); free(act); free(rew); free(term); free(trunc);
  return ok ? 0 : 2;
}
