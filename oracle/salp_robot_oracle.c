/*
 * salp_robot_oracle.c — CPU restatement of the reference's HEAD simulator:
 * `Nozzle` + `Robot.step_through_cycle` (src/salp/environments/robot.py) under the env wrapper of
 * src/salp/environments/salp_robot_env.py (`SalpRobotEnv.step/reset/_get_observation/
 * _calculate_reward/generate_target_point("random")`).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as salp_oracle.c).
 *
 * PARITY PIN: checked against the reference's own Python classes run in the build container
 * (tests/golden/gen_robot_golden.py -> tests/golden/robot_*.npz, tests/test_robot_oracle.py).  The
 * reference evaluates its 3x3 products with numpy/BLAS (`@`, np.linalg.inv, np.linalg.norm), whose
 * summation order and FMA use are not specified, so this pin is a TOLERANCE, not bit-for-bit:
 * |diff| <= 1e-9 (abs + rel) on every state component after every env step.
 *
 * One env step = one whole breathing cycle: Robot.set_control + step_through_cycle
 * (robot.py:335-358, 422-445), i.e. (refill + jet + coast) / dt inner Euler steps of dt = 0.01 s.
 */
#define _GNU_SOURCE
#include "salp_robot_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Philox4x32-10 lives in salp_oracle.c */
void salp_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

typedef struct {
  double angle1, angle2, yaw;
  double R_nm[9], R_mb[9], R_br[9];
} nozzle_t;

typedef struct {
  nozzle_t nz;
  double refill_time, jet_time, coast_time, contraction, contract_rate, release_rate;
  int state, cycle;
  double time, cycle_time;
  double length, width, area, volume, water_mass, prev_water_volume, prev_water_mass, drag_coefficient, mass;
  double jet_velocity[3], jet_force[3], jet_torque[3], drag_force[3], drag_torque[3];
  double position[3], velocity[3], velocity_world[3], acceleration[3];
  double euler[3], euler_rate[3], omega[3], alpha[3];
  double prev_I[3];
  /* env wrapper (salp_robot_env.py) */
  double target[2], prev_dist;
  uint32_t rng_counter;
  int64_t inner_steps; /* of the last env step */
} robot_t;

struct salp_robot_oracle {
  salp_robot_config_t cfg;
  int64_t n;
  uint64_t seed;
  int64_t base;
  robot_t* r;
};

/* ---- small linear algebra (row-major 3x3) */
static void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = s;
    }
}
static void mat3_vec(const double* A, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void mat3T_vec(const double* A, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
static double norm3(const double* v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ---- Nozzle (robot.py:5-178) */
static void nozzle_matrices(const salp_robot_config_t* c, nozzle_t* z) { /* robot.py:153-178 */
  const double g = c->nozzle_gamma;
  const double Rt[9] = {cos(g), 0, -sin(g), 0, 1, 0, sin(g), 0, cos(g)};
  const double Rn[9] = {cos(z->angle2), -sin(z->angle2), 0, sin(z->angle2), cos(z->angle2), 0, 0, 0, 1};
  const double Rm[9] = {cos(z->angle1), -sin(z->angle1), 0, sin(z->angle1), cos(z->angle1), 0, 0, 0, 1};
  const double Rb[9] = {0, 0, -1, 0, 1, 0, 1, 0, 0};
  mat3_mul(Rt, Rn, z->R_nm);
  memcpy(z->R_mb, Rm, sizeof(Rm));
  memcpy(z->R_br, Rb, sizeof(Rb));
}
static void nozzle_set_angles(const salp_robot_config_t* c, nozzle_t* z, double a1, double a2) {
  z->angle1 = a1; z->angle2 = a2;
  nozzle_matrices(c, z);
}
static void nozzle_solve_angles(nozzle_t* z) { /* robot.py:55-85 */
  double t0[3] = {-cos(z->yaw), -sin(z->yaw), -0.0};
  double t[3];
  mat3T_vec(z->R_br, t0, t);
  double val2 = clipd(2 * t[2] - 1, -1.0, 1.0);
  z->angle2 = acos(val2);
  if (z->angle2 <= -M_PI) z->angle2 += 2 * M_PI;
  else if (z->angle2 > M_PI) z->angle2 -= 2 * M_PI;
  if (z->angle2 == 0) {
    z->angle1 = 0.0;
  } else {
    double a = 0.5 * (cos(z->angle2) - 1);
    double b = sqrt(2.0) * sin(z->angle2) / 2;
    double cc = t[1];
    double val1 = clipd(cc / sqrt(a * a + b * b), -1.0, 1.0);
    z->angle1 = asin(val1) - atan2(b, a);
  }
  if (z->angle1 <= -M_PI) z->angle1 += 2 * M_PI;
  else if (z->angle1 > M_PI) z->angle1 -= 2 * M_PI;
}
static void nozzle_direction(const salp_robot_config_t* c, const nozzle_t* z, double* d) { /* robot.py:115-130 */
  const double v[3] = {cos(c->nozzle_gamma), 0, sin(c->nozzle_gamma)};
  double M1[9], M2[9];
  mat3_mul(z->R_br, z->R_mb, M1);
  mat3_mul(M1, z->R_nm, M2);
  mat3_vec(M2, v, d);
}
static void nozzle_middle_position(const salp_robot_config_t* c, const nozzle_t* z, double* p) { /* robot.py:132-151 */
  const double base[3] = {0, 0, c->nozzle_length1}, mid[3] = {0, 0, c->nozzle_length2};
  double t[3], s[3];
  mat3_vec(z->R_mb, mid, t);
  for (int i = 0; i < 3; ++i) s[i] = base[i] + t[i];
  mat3_vec(z->R_br, s, p);
}

/* ---- Robot (robot.py:181-775) */
static double water_volume(const robot_t* r) { /* robot.py:737-741 */
  return 4.0 / 3 * M_PI * (r->length / 2) * pow(r->width / 2, 2.0);
}
static double get_mass(const salp_robot_config_t* c, robot_t* r) { /* robot.py:749-759 */
  r->water_mass = c->density * water_volume(r);
  return c->dry_mass + r->water_mass + c->nozzle_mass;
}
static double drag_coefficient(const salp_robot_config_t* c, const robot_t* r) { /* robot.py:627-649 */
  double aspect = r->length / r->width;
  double init_aspect = c->init_length / c->init_width;
  double cl = c->init_length - c->max_contraction;
  double cw = c->init_length - cl + c->init_width;
  double min_aspect = cl / cw;
  double nr = (aspect - min_aspect) / (init_aspect - min_aspect);
  nr = clipd(nr, 0, 1);
  return c->drag_coefficient_max - nr * (c->drag_coefficient_max - c->drag_coefficient_min);
}
static void moment_arm(const salp_robot_config_t* c, const robot_t* r, double* arm) { /* robot.py:567-575 */
  double p[3];
  nozzle_middle_position(c, &r->nz, p);
  arm[0] = p[0] + -r->length / 2; arm[1] = p[1] + 0.0; arm[2] = p[2] + 0.0;
}
static void inertia_diag(const salp_robot_config_t* c, const robot_t* r, double* I) { /* robot.py:534-551 */
  double arm[3];
  moment_arm(c, r, arm);
  double In = c->nozzle_mass * pow(norm3(arm), 2.0);
  double hw2 = pow(r->width / 2, 2.0), hl2 = pow(r->length / 2, 2.0);
  I[0] = 0.2 * r->mass * (hw2 + hw2) + In * 0;
  I[1] = 0.2 * r->mass * (hl2 + hw2) + In * 1;
  I[2] = 0.2 * r->mass * (hw2 + hl2) + In * 1;
}

static void robot_reset(const salp_robot_config_t* c, robot_t* r) { /* robot.py:287-312 */
  r->time = 0.0; r->cycle_time = 0.0; r->cycle = 0; r->state = 3;
  memset(r->position, 0, sizeof(r->position)); memset(r->velocity, 0, sizeof(r->velocity));
  memset(r->velocity_world, 0, sizeof(r->velocity_world)); memset(r->acceleration, 0, sizeof(r->acceleration));
  memset(r->euler, 0, sizeof(r->euler)); memset(r->euler_rate, 0, sizeof(r->euler_rate));
  memset(r->omega, 0, sizeof(r->omega)); memset(r->alpha, 0, sizeof(r->alpha));
  r->length = c->init_length; r->width = c->init_width;
  r->area = M_PI * (r->length / 2) * (r->width / 2);
  r->volume = water_volume(r);
  r->mass = get_mass(c, r);
  r->prev_water_mass = r->mass; /* robot.py:309 stores the mass here; overwritten before use */
  r->prev_water_volume = r->volume;
  inertia_diag(c, r, r->prev_I);
  r->drag_coefficient = drag_coefficient(c, r);
}

static void robot_inner_step(const salp_robot_config_t* c, robot_t* r) { /* robot.py:387-396 */
  const double dt = c->dt;
  r->cycle_time += dt;
  r->time += dt;
  /* update_state, robot.py:360-373 */
  if (r->cycle_time <= r->refill_time) r->state = 0;
  else if (r->cycle_time <= r->refill_time + r->jet_time) r->state = 1;
  else if (r->cycle_time <= r->refill_time + r->jet_time + r->coast_time) r->state = 2;
  else r->state = 3;
  /* update_properties, robot.py:375-385 */
  r->prev_water_volume = r->volume;
  r->prev_water_mass = r->prev_water_volume * c->density;
  double length, width;
  if (r->state == 0) {
    length = c->init_length - r->cycle_time * r->contract_rate;
    width = c->init_width + r->cycle_time * r->contract_rate;
  } else if (r->state == 1) {
    length = c->init_length - r->contraction + (r->cycle_time - r->refill_time) * r->release_rate;
    width = c->init_width + r->contraction - (r->cycle_time - r->refill_time) * r->release_rate;
  } else {
    length = c->init_length; width = c->init_width;
  }
  r->length = length; r->width = width;
  r->area = M_PI * (r->length / 2) * (r->width / 2);
  r->volume = water_volume(r);
  r->mass = get_mass(c, r);
  r->drag_coefficient = drag_coefficient(c, r);
  /* _newton_equations, robot.py:494-505 */
  double wxv[3], Fc[3];
  cross3(r->omega, r->velocity, wxv);
  for (int i = 0; i < 3; ++i) Fc[i] = r->mass * wxv[i];
  {
    double k = -0.5 * c->density * r->area * r->drag_coefficient;
    double kq = k * norm3(r->velocity);
    for (int i = 0; i < 3; ++i) r->drag_force[i] = kq * r->velocity[i] + k * r->velocity[i];
  }
  if (r->state != 1) {
    memset(r->jet_velocity, 0, sizeof(r->jet_velocity));
    memset(r->jet_force, 0, sizeof(r->jet_force));
  } else {
    double volume_rate = -(r->volume - r->prev_water_volume) / dt;
    double jet_speed = volume_rate / c->nozzle_area;
    double dir[3];
    nozzle_direction(c, &r->nz, dir);
    for (int i = 0; i < 3; ++i) r->jet_velocity[i] = dir[i] * jet_speed;
    double mass_rate = (r->water_mass - r->prev_water_mass) / dt;
    for (int i = 0; i < 3; ++i) r->jet_force[i] = 0.1 * mass_rate * r->jet_velocity[i];
  }
  r->mass = get_mass(c, r);
  {
    double inv = 1.0 / r->mass;
    for (int i = 0; i < 3; ++i) r->acceleration[i] = inv * (r->jet_force[i] + r->drag_force[i] + Fc[i]);
  }
  /* _euler_equations, robot.py:507-522 */
  double I[3], Iw[3], wxIw[3], arm[3], Irate[3], Tdef[3];
  const double Tasym[3] = {0.0, 0.0, 0.1 * norm3(r->velocity)};
  inertia_diag(c, r, I);
  for (int i = 0; i < 3; ++i) Iw[i] = I[i] * r->omega[i];
  cross3(r->omega, Iw, wxIw);
  {
    double k = -c->density * r->drag_coefficient * (r->width / 2) * pow(r->length / 2, 4.0) * norm3(r->omega);
    for (int i = 0; i < 3; ++i) r->drag_torque[i] = k * r->omega[i];
  }
  moment_arm(c, r, arm);
  cross3(arm, r->jet_force, r->jet_torque);
  for (int i = 0; i < 3; ++i) { Irate[i] = (I[i] - r->prev_I[i]) / dt; r->prev_I[i] = I[i]; Tdef[i] = Irate[i] * r->omega[i]; }
  for (int i = 0; i < 3; ++i)
    r->alpha[i] = (1.0 / I[i]) * (r->jet_torque[i] + r->drag_torque[i] + -wxIw[i] + Tasym[i] - Tdef[i]);
  /* _update_motion_states, robot.py:524-532 */
  for (int i = 0; i < 3; ++i) r->velocity[i] += r->acceleration[i] * dt;
  for (int i = 0; i < 3; ++i) r->omega[i] += r->alpha[i] * dt;
  {
    const double phi = r->euler[0], th = r->euler[1];
    const double T[9] = {1, sin(phi) * tan(th), cos(phi) * tan(th), 0, cos(phi), -sin(phi),
                         0, sin(phi) / cos(th), cos(phi) / cos(th)};
    mat3_vec(T, r->omega, r->euler_rate);
  }
  for (int i = 0; i < 3; ++i) r->euler[i] += r->euler_rate[i] * dt;
  {
    const double phi = r->euler[0], th = r->euler[1], psi = r->euler[2];
    const double Rx[9] = {1, 0, 0, 0, cos(phi), -sin(phi), 0, sin(phi), cos(phi)};
    const double Ry[9] = {cos(th), 0, sin(th), 0, 1, 0, -sin(th), 0, cos(th)};
    const double Rz[9] = {cos(psi), -sin(psi), 0, sin(psi), cos(psi), 0, 0, 0, 1};
    double M1[9], R[9];
    mat3_mul(Rz, Ry, M1);
    mat3_mul(M1, Rx, R);
    mat3_vec(R, r->velocity, r->velocity_world);
  }
  for (int i = 0; i < 3; ++i) r->position[i] += r->velocity_world[i] * dt;
}

static double u53(uint32_t hi, uint32_t lo) {
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) / 9007199254740992.0;
}

/* salp_robot_env.py:98-128 reset(): target ~ generate_target_point("random") (:228-265), robot.reset() */
static void env_reset(const struct salp_robot_oracle* h, int64_t i, robot_t* r) {
  const salp_robot_config_t* c = &h->cfg;
  uint64_t g = (uint64_t)(h->base + i);
  uint32_t ctr[4] = {(uint32_t)g, (uint32_t)(g >> 32), r->rng_counter, 16u};
  uint32_t key[2] = {(uint32_t)h->seed, (uint32_t)(h->seed >> 32)}, w[4];
  salp_oracle_philox4x32_10(ctr, key, w);
  r->rng_counter += 1u;
  const double scale = 200.0;
  double x_min = (-(double)c->width / 2 + c->tank_margin) / scale, x_max = ((double)c->width / 2 - c->tank_margin) / scale;
  double y_min = (-(double)c->height / 2 + c->tank_margin) / scale, y_max = ((double)c->height / 2 - c->tank_margin) / scale;
  r->target[0] = x_min + (x_max - x_min) * u53(w[0], w[1]);   /* np.random.uniform(lo, hi) = lo + (hi-lo)*u */
  r->target[1] = y_min + (y_max - y_min) * u53(w[2], w[3]);
  robot_reset(c, r);
  double dx = r->position[0] - r->target[0], dy = r->position[1] - r->target[1];
  r->prev_dist = sqrt(dx * dx + dy * dy);
}

static void env_observe(const robot_t* r, float* obs) { /* salp_robot_env.py:400-420 */
  obs[0] = (float)(r->position[0] - r->target[0]);
  obs[1] = (float)(r->position[1] - r->target[1]);
  obs[2] = (float)r->velocity[0];
  obs[3] = (float)r->velocity[1];
  obs[4] = (float)r->euler[2];
  obs[5] = (float)r->omega[2];
}

/* salp_robot_env.py:139-201 step() */
static double env_step(const struct salp_robot_oracle* h, robot_t* r, const float* act, int* term, int* trunc) {
  const salp_robot_config_t* c = &h->cfg;
  /* _rescale_action :129-137, in fp64: the f32 action values are widened first (what the reference
   * computes when handed a float64 action array; with a float32 array numpy >= 2 rescales in f32) */
  double ra0 = (double)act[0] * 0.06, ra1 = (double)act[1] * 10.0, ra2 = (double)act[2] * (M_PI / 2);
  r->nz.yaw = ra2;
  nozzle_solve_angles(&r->nz);
  /* set_control :335-358 */
  r->contraction = ra0;
  r->coast_time = ra1;
  nozzle_set_angles(c, &r->nz, r->nz.angle1, r->nz.angle2);
  r->cycle += 1;
  r->cycle_time = 0.0;
  r->contract_rate = 0.06 / 3;
  r->refill_time = r->contraction / r->contract_rate;
  r->release_rate = 0.06 / 1.5;
  r->jet_time = r->contraction / r->release_rate;
  /* step_through_cycle :422-445 */
  double total = r->refill_time + r->jet_time + r->coast_time;
  r->inner_steps = 0;
  while (r->cycle_time < total) { robot_inner_step(c, r); r->inner_steps += 1; }
  /* _calculate_reward :203-243 (self.action is never updated by step(), so the smoothness term is -0.0) */
  double dx = r->position[0] - r->target[0], dy = r->position[1] - r->target[1];
  double dist = sqrt(dx * dx + dy * dy);
  double r_track = (-dist + r->prev_dist) * 100;
  r->prev_dist = dist;
  double ex = -(dx / (dist + 1e-6)), ey = -(dy / (dist + 1e-6));
  double vn = sqrt(r->velocity_world[0] * r->velocity_world[0] + r->velocity_world[1] * r->velocity_world[1]);
  double hx = r->velocity_world[0] / (vn + 1e-6), hy = r->velocity_world[1] / (vn + 1e-6);
  double r_heading = hx * ex + hy * ey;
  double reward = (1.0 * r_track) + (0.5 * r_heading) + 0.0 + -0.0;
  *term = 0; *trunc = 0;
  if (dist < 0.01) { *term = 1; reward += 10.0; }
  else if (dist > 5.0) { *trunc = 1; reward -= 5.0; }
  if (r->cycle >= 500) *trunc = 1;
  return reward;
}

/* ------------------------------------------------------------------ C API */
int salp_robot_oracle_create(const salp_robot_config_t* cfg, int64_t n, uint64_t seed, int64_t base,
                             salp_robot_oracle_t** out) {
  if (!cfg || cfg->struct_size != sizeof(*cfg) || n <= 0 || !out) return -1;
  salp_robot_oracle_t* h = (salp_robot_oracle_t*)calloc(1, sizeof(*h));
  if (!h) return -4;
  h->cfg = *cfg; h->n = n; h->seed = seed; h->base = base;
  h->r = (robot_t*)calloc((size_t)n, sizeof(robot_t));
  if (!h->r) { free(h); return -4; }
  for (int64_t i = 0; i < n; ++i) {
    nozzle_set_angles(cfg, &h->r[i].nz, 0.0, 0.0); /* train_robot.py:16 */
    env_reset(h, i, &h->r[i]);
  }
  *out = h;
  return 0;
}
void salp_robot_oracle_destroy(salp_robot_oracle_t* h) { if (h) { free(h->r); free(h); } }

int salp_robot_oracle_reset(salp_robot_oracle_t* h, const uint8_t* mask, float* obs) {
  for (int64_t i = 0; i < h->n; ++i) {
    if (!mask || mask[i]) env_reset(h, i, &h->r[i]);
    if (obs) env_observe(&h->r[i], obs + 6 * i);
  }
  return 0;
}

int salp_robot_oracle_step(salp_robot_oracle_t* h, const float* act, float* obs, double* reward,
                           uint8_t* terminated, uint8_t* truncated, float* final_obs, int32_t* inner_steps) {
  for (int64_t i = 0; i < h->n; ++i) {
    robot_t* r = &h->r[i];
    int te, tr;
    double rew = env_step(h, r, act + 3 * i, &te, &tr);
    if (reward) reward[i] = rew;
    if (terminated) terminated[i] = (uint8_t)te;
    if (truncated) truncated[i] = (uint8_t)tr;
    if (inner_steps) inner_steps[i] = (int32_t)r->inner_steps;
    if (te || tr) {
      if (final_obs) env_observe(r, final_obs + 6 * i);
      env_reset(h, i, r);
    }
    if (obs) env_observe(r, obs + 6 * i);
  }
  return 0;
}

int salp_robot_oracle_get_state(salp_robot_oracle_t* h, double* s) {
  const int64_t n = h->n;
  for (int64_t i = 0; i < n; ++i) {
    const robot_t* r = &h->r[i];
    for (int k = 0; k < 3; ++k) {
      s[(SALP_R_POS + k) * n + i] = r->position[k]; s[(SALP_R_VEL + k) * n + i] = r->velocity[k];
      s[(SALP_R_EULER + k) * n + i] = r->euler[k]; s[(SALP_R_OMEGA + k) * n + i] = r->omega[k];
      s[(SALP_R_VEL_WORLD + k) * n + i] = r->velocity_world[k]; s[(SALP_R_PREV_I + k) * n + i] = r->prev_I[k];
    }
    s[SALP_R_TARGET * n + i] = r->target[0]; s[(SALP_R_TARGET + 1) * n + i] = r->target[1];
    s[SALP_R_PREV_DIST * n + i] = r->prev_dist; s[SALP_R_VOLUME * n + i] = r->volume;
    s[SALP_R_ANGLE1 * n + i] = r->nz.angle1; s[SALP_R_ANGLE2 * n + i] = r->nz.angle2;
    s[SALP_R_TIME * n + i] = r->time; s[SALP_R_CYCLE * n + i] = (double)r->cycle;
    s[SALP_R_RNG * n + i] = (double)r->rng_counter;
  }
  return 0;
}
