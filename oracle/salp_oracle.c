/*
 * salp_oracle.c — CPU restatement of the reference's SalpSnakeEnv.step()/reset().
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library, and only as the checker / the
 * reported CPU baseline.  The shipped path (underwater-swimmer_rl_amd/csrc) never links,
 * loads or calls anything here.
 *
 * PARITY PIN: this restatement is checked BIT-FOR-BIT (fp64 state, fp64 reward, f32
 * observation, flags) against the reference's own Python implementation run in the build
 * container (tests/golden/ref_harness.py, tests/golden/gen_golden.py -> the .npz files beside it),
 * and against the reference's recorded human demonstrations (tests/golden/human_demo_*.npz).
 *
 * Every function cites the reference lines it follows ("legacy" =
 * scripts/utilities/salp_robot.py, "snake" = src/salp/environments/salp_snake_env.py).
 * Arithmetic is IEEE double in the reference's operation order, glibc libm for
 * sin/cos/atan2/sqrt/pow (CPython's math module and float.__pow__ call the same functions),
 * compiled with -ffp-contract=off so no multiply-add is fused.
 */
#define _GNU_SOURCE
#include "salp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ Philox4x32-10 */
void salp_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t hi, uint32_t lo) {
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) / 9007199254740992.0;
}

/* ------------------------------------------------------------------ one environment */
typedef struct {
  /* legacy:61-77 */
  double x, y, vx, vy, theta, omega;
  double nozzle, target_nozzle;
  int phase;  /* 0 rest, 1 inhaling, 2 exhaling */
  int timer;
  int exhale_dur; /* current_exhale_duration, legacy:223 */
  double a, b;    /* ellipse_a, ellipse_b */
  int is_inhaling;
  double water;
  /* snake:49-59 */
  int num_food;                      /* len(food_positions) of this episode */
  double food[SALP_MAX_FOOD][2];     /* NaN = None */
  double score;
  int food_collected;
  int steps_since_food;
  /* build-side bookkeeping */
  uint32_t rng_counter;
  int shape_hold;
  int episode_length;
  double episode_return;
} env_t;

struct salp_oracle {
  salp_config_t cfg;
  int64_t n;
  uint64_t seed;
  int64_t base;
  int64_t global_step;
  int base_num_food;                 /* snake:36 base_num_food_items: foods of the NEXT episodes (<= cfg.num_food_items slots) */
  env_t* env;
};

static int g_threads = 1;
void salp_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int salp_oracle_get_threads(void) { return g_threads; }

static void next_block(const struct salp_oracle* h, int64_t i, env_t* e, uint32_t w[4]) {
  uint64_t g = (uint64_t)(h->base + i);
  uint32_t ctr[4] = {(uint32_t)g, (uint32_t)(g >> 32), e->rng_counter, 0u};
  uint32_t key[2] = {(uint32_t)h->seed, (uint32_t)(h->seed >> 32)};
  salp_oracle_philox4x32_10(ctr, key, w);
  e->rng_counter += 1u;
}

static int is_none(const double f[2]) { return isnan(f[0]); }

static double pymax(double a, double b) { return (b > a) ? b : a; } /* Python max(a, b) */
static double pymin(double a, double b) { return (b < a) ? b : a; } /* Python min(a, b) */

/* random.uniform(lo, hi) pair of snake:101-104 — one Philox block per (x, y) attempt */
static void draw_xy(const struct salp_oracle* h, int64_t i, env_t* e, double* x, double* y) {
  const salp_config_t* c = &h->cfg;
  uint32_t w[4];
  next_block(h, i, e, w);
  double xlo = c->tank_margin + c->food_radius, xhi = (double)c->width - c->tank_margin - c->food_radius;
  double ylo = xlo, yhi = (double)c->height - c->tank_margin - c->food_radius;
  *x = xlo + (xhi - xlo) * u53(w[0], w[1]);
  *y = ylo + (yhi - ylo) * u53(w[2], w[3]);
}

/* snake:92-131 _generate_food_positions */
static void generate_food(const struct salp_oracle* h, int64_t i, env_t* e) {
  const salp_config_t* c = &h->cfg;
  int count = 0;
  for (int k = 0; k < SALP_MAX_FOOD; ++k) e->food[k][0] = e->food[k][1] = NAN;
  for (int f = 0; f < e->num_food; ++f) {
    int attempts = 0;
    while (attempts < 100) {
      double x, y;
      draw_xy(h, i, e, &x, &y);
      int valid = 1;
      for (int k = 0; k < count; ++k) {
        double d = sqrt(pow(x - e->food[k][0], 2.0) + pow(y - e->food[k][1], 2.0));
        if (d < c->min_food_distance) { valid = 0; break; }
      }
      double rd = sqrt(pow(x - (double)c->width / 2, 2.0) + pow(y - (double)c->height / 2, 2.0));
      if (rd < c->min_food_distance) valid = 0;
      if (valid) { e->food[count][0] = x; e->food[count][1] = y; ++count; break; }
      ++attempts;
    }
    if (attempts >= 100) {
      double x, y;
      draw_xy(h, i, e, &x, &y);
      e->food[count][0] = x; e->food[count][1] = y; ++count;
    }
  }
}

/* legacy:371-388 + snake:366-428 _get_extended_observation */
static void observe(const struct salp_oracle* h, const env_t* e, float* obs) {
  const salp_config_t* c = &h->cfg;
  const int K = c->max_observed_food;
  const double W = (double)c->width, H = (double)c->height;
  obs[0] = (float)(e->x / W);
  obs[1] = (float)(e->y / H);
  obs[2] = (float)(e->vx / 5.0);
  obs[3] = (float)(e->vy / 5.0);
  obs[4] = (float)(e->theta / M_PI);
  obs[5] = (float)(e->omega / 0.1);
  obs[6] = (float)(pymax(e->a, e->b) / c->base_radius);
  obs[7] = (float)((double)e->phase / 2.0);
  obs[8] = (float)e->water;
  obs[9] = (float)(e->nozzle / c->max_nozzle_angle);

  int idx[SALP_MAX_FOOD];
  double dist[SALP_MAX_FOOD];
  int cnt = 0;
  for (int k = 0; k < e->num_food; ++k) {
    if (is_none(e->food[k])) continue;
    double d = sqrt(pow(e->food[k][0] - e->x, 2.0) + pow(e->food[k][1] - e->y, 2.0));
    /* list.sort(key=distance) is stable: insert after every entry with dist <= d */
    int p = cnt;
    while (p > 0 && dist[p - 1] > d) { dist[p] = dist[p - 1]; idx[p] = idx[p - 1]; --p; }
    dist[p] = d; idx[p] = k; ++cnt;
  }
  const double diag = sqrt((double)((int64_t)c->width * c->width + (int64_t)c->height * c->height));
  float* fo = obs + 10;
  for (int s = 0; s < K; ++s) {
    if (s < cnt) {
      const double* fp = e->food[idx[s]];
      double rel_x = (fp[0] - e->x) / W;
      double rel_y = (fp[1] - e->y) / H;
      double nd = dist[s] / diag;
      double ang = atan2(fp[1] - e->y, fp[0] - e->x);
      double rel = ang - e->theta;
      while (rel > M_PI) rel -= 2 * M_PI;
      while (rel < -M_PI) rel += 2 * M_PI;
      fo[4 * s + 0] = (float)rel_x;
      fo[4 * s + 1] = (float)rel_y;
      fo[4 * s + 2] = (float)nd;
      fo[4 * s + 3] = (float)(rel / M_PI);
    } else {
      fo[4 * s + 0] = 0.f; fo[4 * s + 1] = 0.f; fo[4 * s + 2] = 1.f; fo[4 * s + 3] = 0.f;
    }
  }
  double nfc = pymin((double)cnt / 10.0, 1.0);
  double navg = 1.0;
  if (cnt > 0) {
    double s = 0.0; /* Python sum(): left-to-right from int 0 */
    for (int k = 0; k < cnt; ++k) s = s + dist[k];
    navg = (s / (double)cnt) / diag;
  }
  fo[4 * K + 0] = (float)nfc;
  fo[4 * K + 1] = (float)navg;
}

/* legacy:95-117 reset + snake:133-155 */
static void reset_env(const struct salp_oracle* h, int64_t i, env_t* e) {
  const salp_config_t* c = &h->cfg;
  e->x = (double)c->width / 2; e->y = (double)c->height / 2;
  e->vx = e->vy = 0.0; e->theta = 0.0; e->omega = 0.0;
  e->nozzle = 0.0; e->target_nozzle = 0.0;
  e->phase = 0; e->timer = 0;
  e->a = c->base_radius; e->b = c->base_radius;
  e->is_inhaling = 0; e->water = 0.0;
  e->score = 0.0; e->food_collected = 0; e->steps_since_food = 0;
  e->shape_hold = 7; e->episode_length = 0; e->episode_return = 0.0;
  if (c->random_food_count) { /* snake:144-146 random.randint(1, max(1, base)) */
    uint32_t w[4];
    next_block(h, i, e, w);
    int n = h->base_num_food > 1 ? h->base_num_food : 1;
    e->num_food = 1 + (int)(((uint64_t)w[0] * (uint64_t)n) >> 32);
    if (e->num_food > c->num_food_items) e->num_food = c->num_food_items;
  } else {
    e->num_food = h->base_num_food;
  }
  generate_food(h, i, e);
}

/* legacy:261-314 _apply_jet_thrust */
static void apply_jet_thrust(const struct salp_oracle* h, int64_t i, env_t* e) {
  const salp_config_t* c = &h->cfg;
  double T = c->max_thrust_force * e->water * 0.4;
  double thrust_angle = e->theta - e->nozzle;
  double tx = cos(thrust_angle) * T;
  double ty = sin(thrust_angle) * T;
  e->vx += tx * 0.012;
  e->vy += ty * 0.012;
  double primary = -e->nozzle * T * 0.0002;
  double moment_arm = pymax(e->a, e->b) * 0.7;
  double perp = T * sin(-e->nozzle);
  double moment = perp * moment_arm * 0.00005;
  double shape = -e->nozzle * T * e->water * 0.00003;
  double total = primary + moment + shape;
  e->omega += total;
  double side_angle = thrust_angle + M_PI / 2;
  double S = T * fabs(e->nozzle) * 0.3;
  double sx = cos(side_angle) * S;
  double sy = sin(side_angle) * S;
  e->vx += sx * 0.008;
  e->vy += sy * 0.008;
  uint32_t w[4];
  next_block(h, i, e, w);
  double u = u53(w[0], w[1]); /* np.random.random(), legacy:311 */
  double noise_angle = thrust_angle + (u - 0.5) * 0.05;
  double noise_force = T * 0.04;
  e->vx += cos(noise_angle) * noise_force * 0.002;
  e->vy += sin(noise_angle) * noise_force * 0.002;
}

/* snake:232-276 _respawn_food */
static void respawn_food(const struct salp_oracle* h, int64_t i, env_t* e) {
  const salp_config_t* c = &h->cfg;
  int attempts = 0;
  while (attempts < 50) {
    double x, y;
    draw_xy(h, i, e, &x, &y);
    double rd = sqrt(pow(x - e->x, 2.0) + pow(y - e->y, 2.0));
    if (rd < c->min_food_distance) { ++attempts; continue; }
    int valid = 1;
    for (int k = 0; k < e->num_food; ++k) {
      if (is_none(e->food[k])) continue;
      double d = sqrt(pow(x - e->food[k][0], 2.0) + pow(y - e->food[k][1], 2.0));
      if (d < c->min_food_distance) { valid = 0; break; }
    }
    if (valid) {
      for (int k = 0; k < e->num_food; ++k)
        if (is_none(e->food[k])) { e->food[k][0] = x; e->food[k][1] = y; return; }
    }
    ++attempts;
  }
  double x, y;
  draw_xy(h, i, e, &x, &y);
  for (int k = 0; k < e->num_food; ++k)
    if (is_none(e->food[k])) { e->food[k][0] = x; e->food[k][1] = y; return; }
}

/* snake:157-202 step over legacy:119-156 step. Returns reward; sets flags. */
static double step_env(const struct salp_oracle* h, int64_t i, env_t* e, const double* act,
                       int* terminated, int* truncated, int* collision_out) {
  const salp_config_t* c = &h->cfg;
  /* legacy:121-135 action decode */
  double nozzle_direction;
  if (c->forced_breathing) {
    nozzle_direction = act[0];
    int cycle = c->inhale_duration + c->exhale_duration + c->rest_duration; /* legacy:158-167 */
    e->is_inhaling = (e->timer % cycle) < c->inhale_duration;
  } else {
    double inhale_control = act[0];
    nozzle_direction = act[1];
    e->is_inhaling = inhale_control > 0.5;
  }
  e->target_nozzle = nozzle_direction * c->max_nozzle_angle;
  /* legacy:169-182 _update_nozzle */
  {
    double diff = e->target_nozzle - e->nozzle;
    if (fabs(diff) > c->nozzle_response_rate) {
      if (diff > 0) e->nozzle += c->nozzle_response_rate;
      else e->nozzle -= c->nozzle_response_rate;
    } else {
      e->nozzle = e->target_nozzle;
    }
    e->nozzle = pymax(-c->max_nozzle_angle, pymin(c->max_nozzle_angle, e->nozzle));
  }
  /* legacy:184-259 _update_breathing_cycle */
  const double R = c->base_radius;
  e->shape_hold = 0;
  if (e->phase == 0) {
    e->a = R * 1.3;
    e->b = R * 0.8;
    if (e->is_inhaling) { e->phase = 1; e->timer = 0; }
  } else if (e->phase == 1) {
    if (e->is_inhaling && e->timer < c->inhale_duration) {
      e->timer += 1;
      double progress = (double)e->timer / (double)c->inhale_duration;
      double start_a = R * 1.3, start_b = R * 0.8, end_a = R * 1.1, end_b = R * 1.1;
      e->a = start_a + (end_a - start_a) * progress;
      e->b = start_b + (end_b - start_b) * progress;
      e->water = progress;
    } else {
      if (e->water > 0.05) {
        e->phase = 2; e->timer = 0;
        e->exhale_dur = (int)((double)c->exhale_duration * pymax(e->water, 0.3));
      } else {
        e->shape_hold = (e->timer >= 1 && e->timer <= 6) ? e->timer : 0;
        e->phase = 0; e->timer = 0; e->water = 0.0;
      }
    }
  } else {
    e->timer += 1;
    double progress = (double)e->timer / (double)e->exhale_dur;
    if (progress <= 1.0) {
      double start_a = R * 1.1, start_b = R * 1.1, end_a = R * 1.3, end_b = R * 0.8;
      e->a = start_a + (end_a - start_a) * progress;
      e->b = start_b + (end_b - start_b) * progress;
      if (0.1 <= progress && progress <= 0.5) apply_jet_thrust(h, i, e);
      double v = e->water * (1.0 - progress);
      e->water = (v > 0) ? v : 0.0; /* max(0, v) */
    } else {
      e->phase = 0; e->timer = 0; e->water = 0.0;
    }
  }
  /* legacy:316-352 _update_physics */
  e->vx *= c->drag_coefficient;
  e->vy *= c->drag_coefficient;
  e->omega *= c->angular_drag;
  e->x += e->vx;
  e->y += e->vy;
  e->theta += e->omega;
  while (e->theta > M_PI) e->theta -= 2 * M_PI;
  while (e->theta < -M_PI) e->theta += 2 * M_PI;
  {
    double margin = c->tank_margin + pymax(e->a, e->b);
    if (e->x < margin) {
      e->x = margin; e->vx = fabs(e->vx) * 0.4; e->omega *= 0.7;
    } else if (e->x > (double)c->width - margin) {
      e->x = (double)c->width - margin; e->vx = -fabs(e->vx) * 0.4; e->omega *= 0.7;
    }
    if (e->y < margin) {
      e->y = margin; e->vy = fabs(e->vy) * 0.4; e->omega *= 0.7;
    } else if (e->y > (double)c->height - margin) {
      e->y = (double)c->height - margin; e->vy = -fabs(e->vy) * 0.4; e->omega *= 0.7;
    }
  }
  /* snake:204-217 _check_food_collection */
  int food_collected = 0;
  {
    double rr = pymax(e->a, e->b);
    for (int k = 0; k < e->num_food; ++k) {
      if (is_none(e->food[k])) continue;
      double d = sqrt(pow(e->x - e->food[k][0], 2.0) + pow(e->y - e->food[k][1], 2.0));
      if (d < rr + c->food_radius) { e->food[k][0] = e->food[k][1] = NAN; food_collected = 1; break; }
    }
  }
  /* snake:219-230 _check_wall_collision */
  int collision;
  {
    double rr = pymax(e->a, e->b);
    double m = c->tank_margin;
    collision = (e->x - rr <= m) || (e->x + rr >= (double)c->width - m) ||
                (e->y - rr <= m) || (e->y + rr >= (double)c->height - m);
  }
  /* snake:278-327 _calculate_snake_reward */
  double reward = 0.0;
  if (food_collected) {
    reward += c->food_reward;
    if (c->efficiency_bonus > 0) {
      int steps_remaining = c->max_steps_without_food - e->steps_since_food;
      reward += c->efficiency_bonus * (double)steps_remaining;
    }
  }
  if (collision) reward += c->collision_penalty;
  if (c->proximity_reward_weight > 0) {
    int nearest = -1; double best = 0.0; /* snake:350-364 strict < keeps the first minimum */
    for (int k = 0; k < e->num_food; ++k) {
      if (is_none(e->food[k])) continue;
      double d = sqrt(pow(e->x - e->food[k][0], 2.0) + pow(e->y - e->food[k][1], 2.0));
      if (nearest < 0 || d < best) { best = d; nearest = k; }
    }
    if (nearest >= 0) {
      double ang = atan2(e->food[nearest][1] - e->y, e->food[nearest][0] - e->x);
      double al = ang - e->theta;
      while (al > M_PI) al -= 2 * M_PI;
      while (al < -M_PI) al += 2 * M_PI;
      reward += c->proximity_reward_weight * cos(al);
    }
  }
  reward += c->time_penalty;
  /* snake:171-189 */
  e->steps_since_food += 1;
  if (food_collected) {
    e->food_collected += 1;
    e->score += c->food_reward;
    e->steps_since_food = 0;
    if (c->respawn_food) respawn_food(h, i, e);
  }
  *terminated = 0; *truncated = 0;
  if (collision) *terminated = 1;
  else if (e->steps_since_food > c->max_steps_without_food) *truncated = 1;
  else if (!c->respawn_food) {
    int all = 1;
    for (int k = 0; k < e->num_food; ++k) if (!is_none(e->food[k])) { all = 0; break; }
    if (all) *terminated = 1;
  }
  *collision_out = collision;
  e->episode_length += 1;
  e->episode_return += reward;
  return reward;
}

/* ------------------------------------------------------------------ C API */
static int check_cfg(const salp_config_t* c) {
  if (!c || c->struct_size != sizeof(salp_config_t)) return 0;
  if (c->num_food_items < 0 || c->num_food_items > SALP_MAX_FOOD) return 0;
  if (c->max_observed_food < 0 || c->max_observed_food > SALP_MAX_OBSERVED_FOOD) return 0;
  return 1;
}

int salp_oracle_create(const salp_config_t* cfg, int64_t n, uint64_t seed, int64_t base,
                       salp_oracle_t** out) {
  if (!check_cfg(cfg) || n <= 0 || !out) return -1;
  salp_oracle_t* h = (salp_oracle_t*)calloc(1, sizeof(*h));
  if (!h) return -4;
  h->cfg = *cfg; h->n = n; h->seed = seed; h->base = base; h->global_step = 0;
  h->base_num_food = cfg->num_food_items;
  h->env = (env_t*)calloc((size_t)n, sizeof(env_t));
  if (!h->env) { free(h); return -4; }
  for (int64_t i = 0; i < n; ++i) { h->env[i].rng_counter = 0; reset_env(h, i, &h->env[i]); }
  *out = h;
  return 0;
}

void salp_oracle_destroy(salp_oracle_t* h) {
  if (!h) return;
  free(h->env);
  free(h);
}

int salp_oracle_obs_dim(const salp_oracle_t* h) { return 10 + 4 * h->cfg.max_observed_food + 2; }
int salp_oracle_act_dim(const salp_oracle_t* h) { return h->cfg.forced_breathing ? 1 : 2; }

int salp_oracle_reset(salp_oracle_t* h, const uint8_t* mask, float* obs) {
  const int od = salp_oracle_obs_dim(h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t i = 0; i < h->n; ++i) {
    if (!mask || mask[i]) reset_env(h, i, &h->env[i]);
    if (obs) observe(h, &h->env[i], obs + i * od);
  }
  return 0;
}

int salp_oracle_observe(salp_oracle_t* h, float* obs) {
  const int od = salp_oracle_obs_dim(h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t i = 0; i < h->n; ++i) observe(h, &h->env[i], obs + i * od);
  return 0;
}

static float device_action(const salp_oracle_t* h, int64_t i, int64_t t, int j, int act_dim) {
  uint64_t g = (uint64_t)(h->base + i);
  uint32_t ts = (uint32_t)t;   /* word ts & 3 of block ts >> 2 of the action stream 1 + j */
  uint32_t ctr[4] = {(uint32_t)g, (uint32_t)(g >> 32), ts >> 2, (uint32_t)(1 + j)};
  uint32_t key[2] = {(uint32_t)h->seed, (uint32_t)(h->seed >> 32)};
  uint32_t w[4];
  salp_oracle_philox4x32_10(ctr, key, w);
  uint32_t x = w[ts & 3u];
  if (act_dim == 2 && j == 0) return (float)(x >> 8) * 5.9604644775390625e-8f; /* [0,1) */
  return (float)(x >> 8) * 1.1920928955078125e-7f - 1.0f;                      /* [-1,1) */
}

static int rollout_impl(salp_oracle_t* h, const float* act, const double* act64, int32_t horizon, float* obs,
                        float* reward, double* reward64, uint8_t* terminated, uint8_t* truncated,
                        float* final_obs, int32_t* info, float* act_out) {
  const int od = salp_oracle_obs_dim(h), ad = salp_oracle_act_dim(h);
  const int64_t n = h->n;
  const int64_t t0 = h->global_step;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    env_t* e = &h->env[i];
    for (int32_t t = 0; t < horizon; ++t) {
      const int64_t row = (int64_t)t * n + i;
      double a[2];
      if (act64) { for (int j = 0; j < ad; ++j) a[j] = act64[row * ad + j]; }
      else if (act) { for (int j = 0; j < ad; ++j) a[j] = (double)act[row * ad + j]; } /* float(action[j]) */
      else {
        for (int j = 0; j < ad; ++j) {
          float g = device_action(h, i, t0 + t, j, ad);
          a[j] = (double)g;
          if (act_out) act_out[row * ad + j] = g;
        }
      }
      int term, trunc, coll;
      double r = step_env(h, i, e, a, &term, &trunc, &coll);
      if (reward) reward[row] = (float)r;
      if (reward64) reward64[row] = r;
      if (terminated) terminated[row] = (uint8_t)term;
      if (truncated) truncated[row] = (uint8_t)trunc;
      if (info) {
        info[row * SALP_INFO_COLS + SALP_INFO_FOOD_COLLECTED] = e->food_collected;
        info[row * SALP_INFO_COLS + SALP_INFO_STEPS_SINCE_FOOD] = e->steps_since_food;
        info[row * SALP_INFO_COLS + SALP_INFO_COLLISION] = coll;
      }
      if ((term || trunc) && !h->cfg.no_autoreset) {   /* no_autoreset: the caller ignores `done`, as a hand loop may */
        if (final_obs) observe(h, e, final_obs + row * od);
        reset_env(h, i, e);
      }
      if (obs) observe(h, e, obs + row * od);
    }
  }
  h->global_step += horizon;
  return 0;
}

int salp_oracle_rollout(salp_oracle_t* h, const float* act, int32_t horizon, float* obs,
                        float* reward, double* reward64, uint8_t* terminated, uint8_t* truncated,
                        float* final_obs, int32_t* info, float* act_out) {
  return rollout_impl(h, act, NULL, horizon, obs, reward, reward64, terminated, truncated, final_obs, info, act_out);
}

/* fp64 actions, as the reference's human-demo recorder passed them (legacy:125 float(action[0])). */
int salp_oracle_rollout_f64(salp_oracle_t* h, const double* act64, int32_t horizon, float* obs,
                            double* reward64, uint8_t* terminated, uint8_t* truncated) {
  if (!act64) return -1;
  return rollout_impl(h, NULL, act64, horizon, obs, NULL, reward64, terminated, truncated, NULL, NULL, NULL);
}

int salp_oracle_step(salp_oracle_t* h, const float* act, float* obs, float* reward,
                     double* reward64, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                     int32_t* info) {
  if (!act) return -1;
  return salp_oracle_rollout(h, act, 1, obs, reward, reward64, terminated, truncated, final_obs,
                             info, NULL);
}

int salp_oracle_get_state(salp_oracle_t* h, double* f64, int32_t* i32) {
  const int64_t n = h->n;
  const int F = h->cfg.num_food_items;
  for (int64_t i = 0; i < n; ++i) {
    const env_t* e = &h->env[i];
    if (f64) {
      f64[SALP_F_X * n + i] = e->x; f64[SALP_F_Y * n + i] = e->y;
      f64[SALP_F_VX * n + i] = e->vx; f64[SALP_F_VY * n + i] = e->vy;
      f64[SALP_F_THETA * n + i] = e->theta; f64[SALP_F_OMEGA * n + i] = e->omega;
      f64[SALP_F_NOZZLE * n + i] = e->nozzle; f64[SALP_F_WATER * n + i] = e->water;
      f64[SALP_F_ELLIPSE_A * n + i] = e->a; f64[SALP_F_ELLIPSE_B * n + i] = e->b;
      for (int k = 0; k < F; ++k) {
        f64[(SALP_F_FOOD0 + k) * n + i] = (k < e->num_food) ? e->food[k][0] : NAN;
        f64[(SALP_F_FOOD0 + F + k) * n + i] = (k < e->num_food) ? e->food[k][1] : NAN;
      }
    }
    if (i32) {
      i32[SALP_I_PHASE * n + i] = e->phase; i32[SALP_I_TIMER * n + i] = e->timer;
      i32[SALP_I_EXHALE_DUR * n + i] = e->exhale_dur; i32[SALP_I_SHAPE_HOLD * n + i] = e->shape_hold;
      i32[SALP_I_STEPS_SINCE_FOOD * n + i] = e->steps_since_food;
      i32[SALP_I_FOOD_COLLECTED * n + i] = e->food_collected;
      i32[SALP_I_RNG_COUNTER * n + i] = (int32_t)e->rng_counter;
      i32[SALP_I_EPISODE_LENGTH * n + i] = e->episode_length;
    }
  }
  return 0;
}

int salp_oracle_set_state(salp_oracle_t* h, const double* f64, const int32_t* i32) {
  const int64_t n = h->n;
  const int F = h->cfg.num_food_items;
  for (int64_t i = 0; i < n; ++i) {
    env_t* e = &h->env[i];
    if (f64) {
      e->x = f64[SALP_F_X * n + i]; e->y = f64[SALP_F_Y * n + i];
      e->vx = f64[SALP_F_VX * n + i]; e->vy = f64[SALP_F_VY * n + i];
      e->theta = f64[SALP_F_THETA * n + i]; e->omega = f64[SALP_F_OMEGA * n + i];
      e->nozzle = f64[SALP_F_NOZZLE * n + i]; e->water = f64[SALP_F_WATER * n + i];
      e->a = f64[SALP_F_ELLIPSE_A * n + i]; e->b = f64[SALP_F_ELLIPSE_B * n + i];
      e->num_food = F;
      for (int k = 0; k < F; ++k) {
        e->food[k][0] = f64[(SALP_F_FOOD0 + k) * n + i];
        e->food[k][1] = f64[(SALP_F_FOOD0 + F + k) * n + i];
        if (isnan(e->food[k][0]) || isnan(e->food[k][1])) e->food[k][0] = e->food[k][1] = NAN;
      }
    }
    if (i32) {
      e->phase = i32[SALP_I_PHASE * n + i]; e->timer = i32[SALP_I_TIMER * n + i];
      e->exhale_dur = i32[SALP_I_EXHALE_DUR * n + i]; e->shape_hold = i32[SALP_I_SHAPE_HOLD * n + i];
      e->steps_since_food = i32[SALP_I_STEPS_SINCE_FOOD * n + i];
      e->food_collected = i32[SALP_I_FOOD_COLLECTED * n + i];
      e->score = (double)e->food_collected * h->cfg.food_reward;
      e->rng_counter = (uint32_t)i32[SALP_I_RNG_COUNTER * n + i];
      e->episode_length = i32[SALP_I_EPISODE_LENGTH * n + i];
    }
  }
  return 0;
}

int64_t salp_oracle_global_step(const salp_oracle_t* h) { return h->global_step; }

/* The reference's curriculum pokes `env.base_num_food_items = k` (continuous_trainer.py:409-411); it takes
   effect at each env's next reset (snake:144-148). */
int salp_oracle_set_base_num_food(salp_oracle_t* h, int k) {
  if (!h || k < 0 || k > h->cfg.num_food_items) return -1;
  h->base_num_food = k;
  return 0;
}
