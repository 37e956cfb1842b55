// salp_robot.hip — batched HEAD simulator for gfx950 and the C ABI of
// include/salp_robot.h: the reference's `Robot.step_through_cycle` (src/salp/environments/robot.py)
// under `SalpRobotEnv.step/reset` (src/salp/environments/salp_robot_env.py), one robot per lane.
//
// One env step is one whole breathing cycle: up to ~1450 explicit-Euler steps of dt = 0.01 s, each a
// rigid-body update with diagonal mass / inertia, quadratic + linear drag, a jet force during the
// release phase, Euler-angle kinematics and a body->world rotation.  Unlike the SalpSnakeEnv kernel
// this one has a real inner hot loop and almost no memory traffic (27 doubles of state in, 6 floats
// out per cycle): it is bound by the fp64 VALU rate, not by HBM.
//   * state lives in VGPRs for the whole cycle; lanes run different step counts (the cycle length is
//     action-dependent), so the loop runs until the slowest lane of the wavefront is done;
//   * the nozzle geometry (IK solve, three rotation matrices, jet direction, moment arm) is constant
//     over a cycle and is evaluated once per env step;
//   * the 3x3 matrices of the reference are diagonal (mass, inertia) or rotations about one axis, so
//     they are written out as the handful of scalar products they are;
//   * fp64 throughout, in the reference's operation order where that is defined (the reference's own
//     3x3 products go through BLAS, so parity is a tolerance: tests/test_gpu_robot.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

#include <new>
#include <string>

#include "../../include/salp_robot.h"
#include "salp_device.h"   // philox4x32_10, u53

using namespace salp;

namespace {

constexpr int kRBlock = 256;
constexpr double kPi = 3.141592653589793;

struct RobotParams {
  double dry_mass, init_length, init_width, max_contraction, density, dt, cd_min, cd_max;
  double nz_l1, nz_l2, nz_area, nz_mass, nz_gamma;
  double x_min, x_span, y_min, y_span;   // target placement, salp_robot_env.py:241-250
  int max_cycles;
  uint32_t seed_lo, seed_hi;
  uint64_t env_base;
  int64_t n, pitch;
};

struct RobotState {
  double* f;    // [SALP_R_COUNT][pitch] (the public snapshot layout is also the device layout)
};

struct Rb {   // one robot in registers
  double pos[3], vel[3], eul[3], om[3], vw[3], prevI[3];
  double target[2], prev_dist, volume, angle1, angle2, time;
  int cycle;
  uint32_t rng;
};

__device__ __forceinline__ double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// sin/cos of an Euler angle.  The angles are unbounded (yaw winds up).  sincos_small (salp_device.h)
// reduces by pi/2 with fused multiply-adds against a 33 + 53 bit split of pi/2 and keeps the quadrant in
// an int, good to |x| ~ 1e9; past 1e8 rad (never seen; wave-uniform test) the angle is first folded
// into [-pi, pi] against a double-double 2*pi, which holds to < 1e-16 rad up to |x| ~ 1e15.
__device__ __forceinline__ void sincos_euler(double x, double& s, double& c) {
  if (__any(fabs(x) > 1.0e8)) {
    const double k = __builtin_rint(x * 0.15915494309189535);
    x = fma(-k, 2.4492935982947064e-16, fma(-k, 6.283185307179586, x));
  }
  sincos_small(x, s, c);
}
__device__ __forceinline__ double sq(double x) { return x * x; }

// (s, c) <- sin / cos of (angle + d) from sin / cos of the angle, |d| <= 0.25 rad: Taylor polynomials of sin d and
// cos d (truncation < 1e-16) and the angle-addition formulas; ~16 operations instead of a full sincos.
__device__ __forceinline__ void rotate_sincos(double& s, double& c, double d) {
  const double z = d * d;
  double ps = fma(z, 2.7557319223985893e-06, -1.9841269841269841e-04);    // 1/9!, -1/7!
  ps = fma(z, ps, 8.3333333333333332e-03);
  ps = fma(z, ps, -1.6666666666666666e-01);
  const double sd = fma(d * z, ps, d);
  double pc = fma(z, -2.7557319223985888e-07, 2.4801587301587302e-05);    // -1/10!, 1/8!
  pc = fma(z, pc, -1.3888888888888889e-03);
  pc = fma(z, pc, 4.1666666666666664e-02);
  pc = fma(z, pc, -0.5);
  const double cd = fma(z, pc, 1.0);
  const double ns = fma(s, cd, c * sd), nc = fma(c, cd, -(s * sd));
  s = ns; c = nc;
}

// 1/x for a normal, finite x: v_rcp_f64 and two Newton steps (<= 1 ulp; no range scaling / fix-up pass)
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
// sqrt(x) for x = 0 or x well inside the normal range: v_rsq_f64, one coupled Newton step and a final
// residual correction (<= 1 ulp); tiny arguments take the library routine
__device__ __forceinline__ double sqrt_nr(double x) {
  if (__any(x < 1.0e-200 && x != 0.0)) return sqrt(x);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double rr = fma(-h, g, 0.5);
  g = fma(g, rr, g); h = fma(h, rr, h);
  const double rr2 = fma(-h, g, 0.5);
  g = fma(g, rr2, g); h = fma(h, rr2, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}

__device__ __forceinline__ void load_robot(Rb& r, const RobotState& S, const RobotParams& P, int64_t i) {
  const int64_t p = P.pitch;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r.pos[k] = S.f[(SALP_R_POS + k) * p + i]; r.vel[k] = S.f[(SALP_R_VEL + k) * p + i];
    r.eul[k] = S.f[(SALP_R_EULER + k) * p + i]; r.om[k] = S.f[(SALP_R_OMEGA + k) * p + i];
    r.vw[k] = S.f[(SALP_R_VEL_WORLD + k) * p + i]; r.prevI[k] = S.f[(SALP_R_PREV_I + k) * p + i];
  }
  r.target[0] = S.f[SALP_R_TARGET * p + i]; r.target[1] = S.f[(SALP_R_TARGET + 1) * p + i];
  r.prev_dist = S.f[SALP_R_PREV_DIST * p + i]; r.volume = S.f[SALP_R_VOLUME * p + i];
  r.angle1 = S.f[SALP_R_ANGLE1 * p + i]; r.angle2 = S.f[SALP_R_ANGLE2 * p + i];
  r.time = S.f[SALP_R_TIME * p + i];
  r.cycle = (int)S.f[SALP_R_CYCLE * p + i]; r.rng = (uint32_t)S.f[SALP_R_RNG * p + i];
}
__device__ __forceinline__ void store_robot(const Rb& r, const RobotState& S, const RobotParams& P, int64_t i) {
  const int64_t p = P.pitch;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    S.f[(SALP_R_POS + k) * p + i] = r.pos[k]; S.f[(SALP_R_VEL + k) * p + i] = r.vel[k];
    S.f[(SALP_R_EULER + k) * p + i] = r.eul[k]; S.f[(SALP_R_OMEGA + k) * p + i] = r.om[k];
    S.f[(SALP_R_VEL_WORLD + k) * p + i] = r.vw[k]; S.f[(SALP_R_PREV_I + k) * p + i] = r.prevI[k];
  }
  S.f[SALP_R_TARGET * p + i] = r.target[0]; S.f[(SALP_R_TARGET + 1) * p + i] = r.target[1];
  S.f[SALP_R_PREV_DIST * p + i] = r.prev_dist; S.f[SALP_R_VOLUME * p + i] = r.volume;
  S.f[SALP_R_ANGLE1 * p + i] = r.angle1; S.f[SALP_R_ANGLE2 * p + i] = r.angle2;
  S.f[SALP_R_TIME * p + i] = r.time;
  S.f[SALP_R_CYCLE * p + i] = (double)r.cycle; S.f[SALP_R_RNG * p + i] = (double)r.rng;
}

// robot.py:737-759 / 534-551 helpers on (length, width)
__device__ __forceinline__ double water_volume(double length, double width) {
  return 4.0 / 3 * kPi * (length / 2) * sq(width / 2);
}
__device__ __forceinline__ void inertia_diag(const RobotParams& P, double mass, double length, double width, double arm_norm2, double* I) {
  const double In = P.nz_mass * arm_norm2;
  const double hw2 = sq(width / 2), hl2 = sq(length / 2);
  I[0] = 0.2 * mass * (hw2 + hw2);
  I[1] = 0.2 * mass * (hl2 + hw2) + In;
  I[2] = 0.2 * mass * (hw2 + hl2) + In;
}
// |r_nozzle + r_robot|: R_br @ (base + R_mb @ middle) = (-(l1 + l2), 0, 0) for any joint angles
// (R_mb turns about z, the links lie on z; robot.py:132-151, 567-575)
__device__ __forceinline__ double arm_x(const RobotParams& P, double length) { return -(P.nz_l1 + P.nz_l2) + -length / 2; }

// robot.py:287-312 Robot.reset + salp_robot_env.py:98-128 (new target, prev_dist)
__device__ __forceinline__ void reset_robot(Rb& r, const RobotParams& P, uint64_t genv) {
  const U4 w = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), r.rng, 16u, P.seed_lo, P.seed_hi);
  r.rng += 1u;
  r.target[0] = P.x_min + P.x_span * u53(w.x, w.y);
  r.target[1] = P.y_min + P.y_span * u53(w.z, w.w);
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.pos[k] = 0.0; r.vel[k] = 0.0; r.eul[k] = 0.0; r.om[k] = 0.0; r.vw[k] = 0.0; }
  r.time = 0.0; r.cycle = 0;
  const double length = P.init_length, width = P.init_width;
  r.volume = water_volume(length, width);
  const double mass = P.dry_mass + P.density * r.volume + P.nz_mass;
  inertia_diag(P, mass, length, width, sq(arm_x(P, length)), r.prevI);
  const double dx = r.pos[0] - r.target[0], dy = r.pos[1] - r.target[1];
  r.prev_dist = sqrt(dx * dx + dy * dy);
}

__device__ __forceinline__ void observe_robot(const Rb& r, float* o) {   // salp_robot_env.py:400-420
  o[0] = (float)(r.pos[0] - r.target[0]); o[1] = (float)(r.pos[1] - r.target[1]);
  o[2] = (float)r.vel[0]; o[3] = (float)r.vel[1]; o[4] = (float)r.eul[2]; o[5] = (float)r.om[2];
}

__global__ __launch_bounds__(kRBlock) void salp_robot_reset_kernel(RobotParams P, RobotState S, const uint8_t* mask, float* obs, int do_reset) {
  const int64_t i = (int64_t)blockIdx.x * kRBlock + threadIdx.x;
  if (i >= P.n) return;
  Rb r;
  load_robot(r, S, P, i);
  if (do_reset && (!mask || mask[i])) {
    reset_robot(r, P, P.env_base + (uint64_t)i);
    store_robot(r, S, P, i);
  }
  if (obs) {
    float o[6];
    observe_robot(r, o);
#pragma unroll
    for (int k = 0; k < 6; ++k) obs[i * 6 + k] = o[k];
  }
}


// ---- cycle-length schedule ----------------------------------------------------------------------------
// A cycle lasts (50 + 25) * contraction + coast seconds, anything from 0 to 14.5 s (0..1450 Euler steps)
// depending on the action, and a wavefront runs until its slowest lane is done: with envs in index order
// about half the lane-steps are idle.  A counting sort on the step count (256 bins of ~6 steps, longest
// first) gives the order the step kernel walks the envs in; three small launches, no host round trip.
// Blocks of 1024 envs histogram in LDS first, so a block makes at most one global atomic per bin
// (one global atomic per env measured 45 us per pass at 262144 envs, 7 % of the step).
constexpr int kSchedBins = 256;
constexpr double kMaxCycleTime = 14.6;   // s; an in-Box action asks for at most 0.06 * (3 + 1.5) / 0.06 + 10 = 14.5 s
constexpr int kSchedBlock = 1024;

__device__ __forceinline__ int schedule_bin(const RobotParams& P, const float* act, int64_t i) {
  const double contraction = (double)act[i * 3 + 0] * 0.06;
  const double total = contraction * (3.0 / 0.06 + 1.5 / 0.06) + (double)act[i * 3 + 1] * 10.0;
  const double b = total * ((kSchedBins - 1) / 14.6);
  if (!(b > 0.0)) return 0;                          // also NaN
  return b >= (double)(kSchedBins - 1) ? kSchedBins - 1 : (int)b;
}
__global__ __launch_bounds__(kSchedBlock) void robot_schedule_count(RobotParams P, const float* act, uint32_t* bins) {
  __shared__ uint32_t hist[kSchedBins];
  const int t = threadIdx.x;
  if (t < kSchedBins) hist[t] = 0u;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kSchedBlock + t;
  if (i < P.n) atomicAdd(&hist[schedule_bin(P, act, i)], 1u);
  __syncthreads();
  if (t < kSchedBins && hist[t]) atomicAdd(&bins[t], hist[t]);
}
// bins[k] <- number of envs in bins above k (longest cycles get the first positions)
__global__ __launch_bounds__(kSchedBins) void robot_schedule_scan(uint32_t* bins) {
  __shared__ uint32_t part[kSchedBins];
  const int t = threadIdx.x;
  const int k = kSchedBins - 1 - t;                  // thread t owns bin k (descending order)
  const uint32_t c = bins[k];
  part[t] = c;
  __syncthreads();
  for (int d = 1; d < kSchedBins; d <<= 1) {
    const uint32_t v = t >= d ? part[t - d] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  bins[k] = part[t] - c;
}
__global__ __launch_bounds__(kSchedBlock) void robot_schedule_scatter(RobotParams P, const float* act, uint32_t* bins, int32_t* order) {
  __shared__ uint32_t hist[kSchedBins];              // count, then this block's base position, per bin
  const int t = threadIdx.x;
  if (t < kSchedBins) hist[t] = 0u;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kSchedBlock + t;
  int b = 0;
  uint32_t rank = 0;
  if (i < P.n) { b = schedule_bin(P, act, i); rank = atomicAdd(&hist[b], 1u); }
  __syncthreads();
  if (t < kSchedBins && hist[t]) hist[t] = atomicAdd(&bins[t], hist[t]);
  __syncthreads();
  if (i < P.n) order[hist[b] + rank] = (int32_t)i;
}

// SalpRobotEnv.step (salp_robot_env.py:139-201): one breathing cycle per env.
#ifndef SALP_ROBOT_WAVES
#define SALP_ROBOT_WAVES 2
#endif
__global__ __launch_bounds__(kRBlock) __attribute__((amdgpu_waves_per_eu(SALP_ROBOT_WAVES, SALP_ROBOT_WAVES)))
void salp_robot_step_kernel(RobotParams P, RobotState S, const float* act, float* obs,
                                                                  float* reward, uint8_t* terminated, uint8_t* truncated,
                                                                  float* final_obs, int32_t* inner_steps, const int32_t* order) {
  const int64_t i0 = (int64_t)blockIdx.x * kRBlock + threadIdx.x;
  const bool active = i0 < P.n;
  // `order` (robot_schedule_* below) lists the envs longest cycle first, so that the lanes of a wavefront
  // run about the same number of Euler steps; results do not depend on which lane an env runs in.
  const int64_t i = active ? (order ? (int64_t)order[i0] : i0) : (P.n - 1);
  const uint64_t genv = P.env_base + (uint64_t)i;
  Rb r;
  load_robot(r, S, P, i);

  // _rescale_action (:129-137) in fp64
  const double contraction = (double)act[i * 3 + 0] * 0.06;
  const double coast_time = (double)act[i * 3 + 1] * 10.0;
  const double yaw = (double)act[i * 3 + 2] * (kPi / 2);
  // Nozzle.solve_angles (robot.py:55-85): target = R_br^T @ -(cos yaw, sin yaw, 0) = (-0, -sin yaw, cos yaw).
  // Every angle here is within [-pi, pi]: sincos_small (salp_device.h) is exact to < 1 ulp there.
  {
    double sy, cy;
    sincos_small(yaw, sy, cy);
    const double t1 = -sy, t2 = cy;
    double a2 = acos(clipd(2 * t2 - 1, -1.0, 1.0));
    if (a2 <= -kPi) a2 += 2 * kPi; else if (a2 > kPi) a2 -= 2 * kPi;
    double a1 = 0.0;
    if (a2 != 0.0) {
      double sa2, ca2;
      sincos_small(a2, sa2, ca2);
      const double a = 0.5 * (ca2 - 1);
      const double b = sqrt(2.0) * sa2 / 2;
      a1 = asin(clipd(t1 / sqrt(a * a + b * b), -1.0, 1.0)) - atan2(b, a);
    }
    if (a1 <= -kPi) a1 += 2 * kPi; else if (a1 > kPi) a1 -= 2 * kPi;
    r.angle1 = a1; r.angle2 = a2;
  }
  // Nozzle.get_nozzle_direction (robot.py:115-130): R_br @ R_mb @ R_nm @ (cos g, 0, sin g), constant over the cycle
  double dir[3];
  {
    double cg, sg, c2, s2, c1, s1;
    sincos_small(P.nz_gamma, sg, cg);
    sincos_small(r.angle2, s2, c2);
    sincos_small(r.angle1, s1, c1);
    // R_nm = R_theta_fixed @ R_nozzle(angle2); v1 = R_nm @ (cg, 0, sg)
    const double nx = (cg * c2) * cg + (-sg) * sg;
    const double ny = s2 * cg;
    const double nzv = (sg * c2) * cg + cg * sg;
    // R_mb = rotation about z by angle1
    const double mx = c1 * nx + (-s1) * ny, my = s1 * nx + c1 * ny, mz = nzv;
    // R_br = [[0,0,-1],[0,1,0],[1,0,0]]
    dir[0] = -mz; dir[1] = my; dir[2] = mx;
  }
  // Robot.set_control (robot.py:335-358)
  r.cycle += 1;
  const double contract_rate = 0.06 / 3, release_rate = 0.06 / 1.5;
  const double refill_time = contraction / contract_rate;
  const double jet_time = contraction / release_rate;
  // The cycle length comes straight from the caller's action.  Inside the action Box [0, 1]^3 it is at most
  // 0.06 * 75 + 10 = 14.5 s; the device loop below is bounded by that maximum (kMaxCycleTime), so an unsquashed
  // or diverged policy output (1e9, +inf) cannot spin a wavefront for ever — the reference would stall ONE CPU
  // env for the corresponding 1e11 Euler steps; here a cycle longer than the Box allows is cut at the Box
  // maximum (include/salp_robot.h).  A non-finite length runs no Euler step at all.
  double total = active ? refill_time + jet_time + coast_time : 0.0;   // padding lanes do not step
  total = (total <= kMaxCycleTime) ? total : ((total > kMaxCycleTime) ? kMaxCycleTime : 0.0);
  double cycle_time = 0.0;
  int steps = 0;
  const double dt = P.dt;

  // Robot.step_through_cycle (robot.py:422-445): lanes finish at different times.
  //  * quantities that depend only on the body shape (mass, inertia, drag factors and their reciprocals)
  //    are kept in registers and recomputed only on a step where some lane's shape moves or has just
  //    stopped moving; during coast / rest (most of a cycle) the whole wavefront skips that block;
  //  * sin/cos of the Euler angles are carried from step to step (see rotate_sincos below in the loop);
  //  * divisions by dt, by cos(pitch) and by the mass / inertia diagonal are reciprocals (Newton-refined
  //    v_rcp_f64) times a product: a few ulp from the reference's quotient, far inside the parity
  //    tolerance, which is a tolerance already because the reference multiplies 3x3 blocks through BLAS.
  const double inv_dt = rcp_nr(dt);
  const double init_aspect = P.init_length / P.init_width;
  const double contracted_length = P.init_length - P.max_contraction;
  const double min_aspect = contracted_length / (P.init_length - contracted_length + P.init_width);
  const double inv_aspect_span = rcp_nr(init_aspect - min_aspect);
  const double t_jet_end = refill_time + jet_time, t_coast_end = t_jet_end + coast_time;
  double mass = 0, inv_m = 0, kd = 0, ktc = 0, ax = 0, I0 = 0, I1 = 0, I2 = 0, iI0 = 0, iI1 = 0, iI2 = 0;
  bool settled = false;    // the previous step of this lane already had the rest shape (and prevI == I)
  // sin / cos of the three Euler angles are carried through the cycle: exact at its start, then advanced by each
  // step's increment with rotate_sincos (increments are ~1e-3 rad; an increment above 0.25 rad anywhere in the
  // wavefront takes the exact path for that step).  Drift over a whole cycle stays below 1e-12.
  double sp, cp, st, ct, ss, cs;
  sincos_euler(r.eul[0], sp, cp);
  sincos_euler(r.eul[1], st, ct);
  sincos_euler(r.eul[2], ss, cs);
#pragma unroll 1
  while (__any(cycle_time < total)) {
    if (cycle_time < total) {
      // Robot.step (robot.py:387-396)
      cycle_time += dt;
      r.time += dt;
      int state;   // update_state :360-373
      if (cycle_time <= refill_time) state = 0;
      else if (cycle_time <= t_jet_end) state = 1;
      else if (cycle_time <= t_coast_end) state = 2;
      else state = 3;
      double jf0 = 0.0, jf1 = 0.0, jf2 = 0.0, td0 = 0.0, td1 = 0.0, td2 = 0.0;
      if (__any(state < 2 || !settled)) {
        // update_properties :375-385
        const double prev_volume = r.volume;
        double length, width;
        if (state == 0) { length = P.init_length - cycle_time * contract_rate; width = P.init_width + cycle_time * contract_rate; }
        else if (state == 1) {
          length = P.init_length - contraction + (cycle_time - refill_time) * release_rate;
          width = P.init_width + contraction - (cycle_time - refill_time) * release_rate;
        } else { length = P.init_length; width = P.init_width; }
        const double hl = length / 2, hw = width / 2;
        const double area = kPi * hl * hw;
        r.volume = water_volume(length, width);
        const double water_mass = P.density * r.volume;
        mass = P.dry_mass + water_mass + P.nz_mass;
        inv_m = rcp_nr(mass);
        // drag coefficient, robot.py:627-649
        const double aspect = length * rcp_nr(width);
        const double nr = clipd((aspect - min_aspect) * inv_aspect_span, 0.0, 1.0);
        const double cd = P.cd_max - nr * (P.cd_max - P.cd_min);
        kd = -0.5 * P.density * area * cd;
        ktc = -P.density * cd * hw * sq(sq(hl));
        if (state == 1) {   // jet force, :494-505
          const double volume_rate = -(r.volume - prev_volume) * inv_dt;
          const double jet_speed = volume_rate / P.nz_area;
          const double mass_rate = (water_mass - prev_volume * P.density) * inv_dt;
          jf0 = 0.1 * mass_rate * (dir[0] * jet_speed);
          jf1 = 0.1 * mass_rate * (dir[1] * jet_speed);
          jf2 = 0.1 * mass_rate * (dir[2] * jet_speed);
        }
        ax = arm_x(P, length);
        double I[3];
        inertia_diag(P, mass, length, width, ax * ax, I);
        I0 = I[0]; I1 = I[1]; I2 = I[2];
        iI0 = rcp_nr(I0); iI1 = rcp_nr(I1); iI2 = rcp_nr(I2);
        td0 = ((I0 - r.prevI[0]) * inv_dt) * r.om[0];
        td1 = ((I1 - r.prevI[1]) * inv_dt) * r.om[1];
        td2 = ((I2 - r.prevI[2]) * inv_dt) * r.om[2];
        r.prevI[0] = I0; r.prevI[1] = I1; r.prevI[2] = I2;
        settled = state >= 2;
      }
      // _newton_equations :494-505
      const double wxv0 = r.om[1] * r.vel[2] - r.om[2] * r.vel[1];
      const double wxv1 = r.om[2] * r.vel[0] - r.om[0] * r.vel[2];
      const double wxv2 = r.om[0] * r.vel[1] - r.om[1] * r.vel[0];
      const double vnorm = sqrt_nr(r.vel[0] * r.vel[0] + r.vel[1] * r.vel[1] + r.vel[2] * r.vel[2]);
      const double kq = kd * vnorm;
      const double acc0 = inv_m * (jf0 + (kq * r.vel[0] + kd * r.vel[0]) + mass * wxv0);
      const double acc1 = inv_m * (jf1 + (kq * r.vel[1] + kd * r.vel[1]) + mass * wxv1);
      const double acc2 = inv_m * (jf2 + (kq * r.vel[2] + kd * r.vel[2]) + mass * wxv2);
      // _euler_equations :507-522
      const double Iw0 = I0 * r.om[0], Iw1 = I1 * r.om[1], Iw2 = I2 * r.om[2];
      const double c0 = r.om[1] * Iw2 - r.om[2] * Iw1;
      const double c1 = r.om[2] * Iw0 - r.om[0] * Iw2;
      const double c2 = r.om[0] * Iw1 - r.om[1] * Iw0;
      const double wnorm = sqrt_nr(r.om[0] * r.om[0] + r.om[1] * r.om[1] + r.om[2] * r.om[2]);
      const double kt = ktc * wnorm;
      // jet torque = arm x jet_force, arm = (ax, 0, 0)
      const double jt1 = -ax * jf2, jt2 = ax * jf1;
      const double al0 = iI0 * (kt * r.om[0] + -c0 - td0);
      const double al1 = iI1 * (jt1 + kt * r.om[1] + -c1 - td1);
      const double al2 = iI2 * (jt2 + kt * r.om[2] + -c2 + 0.1 * vnorm - td2);
      // _update_motion_states :524-532
      r.vel[0] += acc0 * dt; r.vel[1] += acc1 * dt; r.vel[2] += acc2 * dt;
      r.om[0] += al0 * dt; r.om[1] += al1 * dt; r.om[2] += al2 * dt;
      {
        const double ict = rcp_nr(ct);
        const double tt = st * ict;
        const double e0 = r.om[0] + (sp * tt) * r.om[1] + (cp * tt) * r.om[2];
        const double e1 = cp * r.om[1] + (-sp) * r.om[2];
        const double e2 = (sp * ict) * r.om[1] + (cp * ict) * r.om[2];
        const double d0 = e0 * dt, d1 = e1 * dt, d2 = e2 * dt;
        r.eul[0] += d0; r.eul[1] += d1; r.eul[2] += d2;
        if (__any(fabs(d0) > 0.25 || fabs(d1) > 0.25 || fabs(d2) > 0.25)) {
          sincos_euler(r.eul[0], sp, cp);
          sincos_euler(r.eul[1], st, ct);
          sincos_euler(r.eul[2], ss, cs);
        } else {
          rotate_sincos(sp, cp, d0);
          rotate_sincos(st, ct, d1);
          rotate_sincos(ss, cs, d2);
        }
      }
      {
        // R = R_z @ R_y @ R_x
        const double r00 = cs * ct, r01 = cs * st * sp - ss * cp, r02 = cs * st * cp + ss * sp;
        const double r10 = ss * ct, r11 = ss * st * sp + cs * cp, r12 = ss * st * cp - cs * sp;
        const double r20 = -st, r21 = ct * sp, r22 = ct * cp;
        r.vw[0] = r00 * r.vel[0] + r01 * r.vel[1] + r02 * r.vel[2];
        r.vw[1] = r10 * r.vel[0] + r11 * r.vel[1] + r12 * r.vel[2];
        r.vw[2] = r20 * r.vel[0] + r21 * r.vel[1] + r22 * r.vel[2];
      }
      r.pos[0] += r.vw[0] * dt; r.pos[1] += r.vw[1] * dt; r.pos[2] += r.vw[2] * dt;
      ++steps;
    }
  }

  // _calculate_reward (:203-243) and termination (:171-185)
  const double dx = r.pos[0] - r.target[0], dy = r.pos[1] - r.target[1];
  const double dist = sqrt(dx * dx + dy * dy);
  const double r_track = (-dist + r.prev_dist) * 100;
  r.prev_dist = dist;
  const double ex = -(dx / (dist + 1e-6)), ey = -(dy / (dist + 1e-6));
  const double vn = sqrt(r.vw[0] * r.vw[0] + r.vw[1] * r.vw[1]);
  const double r_heading = (r.vw[0] / (vn + 1e-6)) * ex + (r.vw[1] / (vn + 1e-6)) * ey;
  double rew = r_track + 0.5 * r_heading;
  bool term = false, trunc = false;
  if (dist < 0.01) { term = true; rew += 10.0; }
  else if (dist > 5.0) { trunc = true; rew -= 5.0; }
  if (r.cycle >= P.max_cycles) trunc = true;

  float o[6];
  if (term || trunc) {
    if (final_obs && active) {
      observe_robot(r, o);
#pragma unroll
      for (int k = 0; k < 6; ++k) final_obs[i * 6 + k] = o[k];
    }
    reset_robot(r, P, genv);
  }
  if (active) {
    observe_robot(r, o);
    if (obs) {
#pragma unroll
      for (int k = 0; k < 6; ++k) obs[i * 6 + k] = o[k];
    }
    if (reward) reward[i] = (float)rew;
    if (terminated) terminated[i] = term ? 1 : 0;
    if (truncated) truncated[i] = trunc ? 1 : 0;
    if (inner_steps) inner_steps[i] = steps;
    store_robot(r, S, P, i);
  }
}

thread_local std::string g_rerr;
int rfail(int code, const std::string& m) { g_rerr = m; return code; }
#define RHIP_TRY(expr)                                                                              \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) return rfail(_e == hipErrorOutOfMemory ? -4 : -3, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// Makes `device` current for the scope of one ABI call and restores the caller's device afterwards, so that the
// library never changes the current HIP device under the caller (PyTorch keeps its own notion of it).  When the
// caller is already on the handle's device — the usual case — this is one hipGetDevice.
struct DeviceScope {
  int prev = -1, changed = 0;
  hipError_t enter(int device) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return e;
    if (prev != device) { e = hipSetDevice(device); changed = (e == hipSuccess); }
    return e;
  }
  ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
};

}  // namespace

struct salp_robot_vec {
  salp_robot_config_t cfg;
  RobotParams P;
  RobotState S;
  int device;
  int64_t n;
  void* stage;
  size_t stage_bytes;
  uint32_t* bins;     // [kSchedBins]
  int32_t* order;     // [n]
  bool schedule;      // walk the envs longest cycle first (see robot_schedule_*)
};

// Below this many envs every wavefront has an execution unit to itself and the launch lasts as long as
// its slowest env whatever the order.  SALP_ROBOT_SCHEDULE=0/1 overrides (tests run both ways).
static const int64_t kScheduleMinEnvs = 32768;

static int launch_robot_step(salp_robot_vec* h, const float* act, float* obs, float* reward, uint8_t* terminated,
                             uint8_t* truncated, float* final_obs, int32_t* inner_steps, hipStream_t st) {
  const unsigned grid = (unsigned)((h->n + kRBlock - 1) / kRBlock);
  const int32_t* order = nullptr;
  if (h->schedule) {
    RHIP_TRY(hipMemsetAsync(h->bins, 0, kSchedBins * sizeof(uint32_t), st));
    const unsigned sgrid = (unsigned)((h->n + kSchedBlock - 1) / kSchedBlock);
    hipLaunchKernelGGL(robot_schedule_count, dim3(sgrid), dim3(kSchedBlock), 0, st, h->P, act, h->bins);
    hipLaunchKernelGGL(robot_schedule_scan, dim3(1), dim3(kSchedBins), 0, st, h->bins);
    hipLaunchKernelGGL(robot_schedule_scatter, dim3(sgrid), dim3(kSchedBlock), 0, st, h->P, act, h->bins, h->order);
    order = h->order;
  }
  hipLaunchKernelGGL(salp_robot_step_kernel, dim3(grid), dim3(kRBlock), 0, st, h->P, h->S, act, obs, reward, terminated,
                     truncated, final_obs, inner_steps, order);
  RHIP_TRY(hipGetLastError());
  return 0;
}

extern "C" {

const char* salp_robot_last_error(void) { return g_rerr.c_str(); }

int salp_robot_config_default(salp_robot_config_t* c) {
  if (!c) return rfail(-1, "cfg is NULL");
  memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(*c);
  c->width = 900; c->height = 700; c->tank_margin = 50.0;
  c->dry_mass = 1.0; c->init_length = 0.3; c->init_width = 0.15; c->max_contraction = 0.06; c->density = 1000.0;
  c->dt = 0.01; c->drag_coefficient_min = 0.4; c->drag_coefficient_max = 1.0;
  c->nozzle_length1 = c->nozzle_length2 = c->nozzle_length3 = 0.05;
  c->nozzle_area = 0.00016; c->nozzle_mass = 1.0; c->nozzle_gamma = kPi / 4; c->max_cycles = 500;
  return 0;
}

int salp_robot_vec_reset(salp_robot_vec_t* h, const uint8_t* mask, float* obs, uint32_t flags, void* stream);

int salp_robot_vec_create(const salp_robot_config_t* cfg, int64_t n_envs, int device_id, uint64_t seed,
                          int64_t env_index_base, salp_robot_vec_t** out) {
  if (!out) return rfail(-1, "out is NULL");
  *out = nullptr;
  if (!cfg || cfg->struct_size != sizeof(salp_robot_config_t)) return rfail(-1, "salp_robot_config_t.struct_size mismatch");
  if (n_envs <= 0 || env_index_base < 0) return rfail(-1, "n_envs / env_index_base out of range");
  if (!(cfg->dt > 0) || !(cfg->init_width > 0) || !(cfg->nozzle_area > 0)) return rfail(-1, "dt, init_width, nozzle_area must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return rfail(-2, "no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return rfail(-2, "device_id out of range");
  DeviceScope dev_scope;
  RHIP_TRY(dev_scope.enter(device_id));
  salp_robot_vec* h = new (std::nothrow) salp_robot_vec();
  if (!h) return rfail(-4, "host allocation failed");
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg; h->device = device_id; h->n = n_envs;
  RobotParams& P = h->P;
  P.dry_mass = cfg->dry_mass; P.init_length = cfg->init_length; P.init_width = cfg->init_width;
  P.max_contraction = cfg->max_contraction; P.density = cfg->density; P.dt = cfg->dt;
  P.cd_min = cfg->drag_coefficient_min; P.cd_max = cfg->drag_coefficient_max;
  P.nz_l1 = cfg->nozzle_length1; P.nz_l2 = cfg->nozzle_length2; P.nz_area = cfg->nozzle_area;
  P.nz_mass = cfg->nozzle_mass; P.nz_gamma = cfg->nozzle_gamma;
  const double scale = 200.0;
  P.x_min = (-(double)cfg->width / 2 + cfg->tank_margin) / scale;
  P.x_span = ((double)cfg->width / 2 - cfg->tank_margin) / scale - P.x_min;
  P.y_min = (-(double)cfg->height / 2 + cfg->tank_margin) / scale;
  P.y_span = ((double)cfg->height / 2 - cfg->tank_margin) / scale - P.y_min;
  P.max_cycles = cfg->max_cycles;
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32);
  P.env_base = (uint64_t)env_index_base; P.n = n_envs; P.pitch = (n_envs + 63) / 64 * 64;
  const size_t bytes = (size_t)SALP_R_COUNT * (size_t)P.pitch * sizeof(double);
  hipError_t e = hipMalloc((void**)&h->S.f, bytes);
  if (e == hipSuccess) e = hipMemset(h->S.f, 0, bytes);
  h->schedule = n_envs >= kScheduleMinEnvs;
  if (const char* ev = getenv("SALP_ROBOT_SCHEDULE")) h->schedule = ev[0] == '1';
  if (n_envs > INT32_MAX) h->schedule = false;
  if (e == hipSuccess && h->schedule) e = hipMalloc((void**)&h->bins, kSchedBins * sizeof(uint32_t));
  if (e == hipSuccess && h->schedule) e = hipMalloc((void**)&h->order, (size_t)n_envs * sizeof(int32_t));
  if (e != hipSuccess) { std::string m = std::string("state allocation: ") + hipGetErrorString(e); salp_robot_vec_destroy(h); return rfail(-4, m); }
  int rc = salp_robot_vec_reset(h, nullptr, nullptr, 1u, nullptr);   // train_robot.py:16 angles (0, 0) are the zeroed rows
  if (rc == 0 && hipDeviceSynchronize() != hipSuccess) rc = rfail(-3, "initial reset failed");
  if (rc != 0) { std::string m = g_rerr; salp_robot_vec_destroy(h); g_rerr = m; return rc; }
  *out = h;
  return 0;
}

void salp_robot_vec_destroy(salp_robot_vec_t* h) {
  if (!h) return;
  DeviceScope dev_scope;
  (void)dev_scope.enter(h->device);
  if (h->S.f) (void)hipFree(h->S.f);
  if (h->stage) (void)hipFree(h->stage);
  if (h->bins) (void)hipFree(h->bins);
  if (h->order) (void)hipFree(h->order);
  delete h;
}

int64_t salp_robot_vec_num_envs(const salp_robot_vec_t* h) { return h ? h->n : 0; }

static int robot_stage(salp_robot_vec* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return 0;
  if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
  RHIP_TRY(hipMalloc(&h->stage, bytes));
  h->stage_bytes = bytes;
  return 0;
}

int salp_robot_vec_reset(salp_robot_vec_t* h, const uint8_t* mask, float* obs, uint32_t flags, void* stream) {
  if (!h) return rfail(-1, "handle is NULL");
  DeviceScope dev_scope;
  RHIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((h->n + kRBlock - 1) / kRBlock);
  if (flags & 1u) {
    hipLaunchKernelGGL(salp_robot_reset_kernel, dim3(grid), dim3(kRBlock), 0, st, h->P, h->S, mask, obs, 1);
    RHIP_TRY(hipGetLastError());
    return 0;
  }
  const size_t ob = (size_t)h->n * 6 * sizeof(float);
  int rc = robot_stage(h, ob + (size_t)h->n + 1024);
  if (rc) return rc;
  float* d_obs = (float*)h->stage;
  uint8_t* d_mask = (uint8_t*)h->stage + ((ob + 255) / 256) * 256;
  if (mask) RHIP_TRY(hipMemcpyAsync(d_mask, mask, (size_t)h->n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(salp_robot_reset_kernel, dim3(grid), dim3(kRBlock), 0, st, h->P, h->S, mask ? (const uint8_t*)d_mask : nullptr,
                     obs ? d_obs : nullptr, 1);
  RHIP_TRY(hipGetLastError());
  if (obs) RHIP_TRY(hipMemcpyAsync(obs, d_obs, ob, hipMemcpyDeviceToHost, st));
  RHIP_TRY(hipStreamSynchronize(st));
  return 0;
}

int salp_robot_vec_step(salp_robot_vec_t* h, const float* act, float* obs, float* reward, uint8_t* terminated,
                        uint8_t* truncated, float* final_obs, int32_t* inner_steps, uint32_t flags, void* stream) {
  if (!h || !act) return rfail(-1, "handle / act is NULL");
  DeviceScope dev_scope;
  RHIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  if (flags & 1u) return launch_robot_step(h, act, obs, reward, terminated, truncated, final_obs, inner_steps, st);
  const size_t n = (size_t)h->n;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t need = up(n * 12) + 2 * up(n * 24) + up(n * 4) + 2 * up(n) + up(n * 4) + 1024;
  int rc = robot_stage(h, need);
  if (rc) return rc;
  char* b = (char*)h->stage;
  float* d_act = (float*)b; b += up(n * 12);
  float* d_obs = (float*)b; b += up(n * 24);
  float* d_fin = (float*)b; b += up(n * 24);
  float* d_rew = (float*)b; b += up(n * 4);
  uint8_t* d_te = (uint8_t*)b; b += up(n);
  uint8_t* d_tr = (uint8_t*)b; b += up(n);
  int32_t* d_in = (int32_t*)b;
  RHIP_TRY(hipMemcpyAsync(d_act, act, n * 12, hipMemcpyHostToDevice, st));
  if (final_obs) RHIP_TRY(hipMemcpyAsync(d_fin, final_obs, n * 24, hipMemcpyHostToDevice, st));
  rc = launch_robot_step(h, d_act, obs ? d_obs : nullptr, reward ? d_rew : nullptr, terminated ? d_te : nullptr,
                         truncated ? d_tr : nullptr, final_obs ? d_fin : nullptr, inner_steps ? d_in : nullptr, st);
  if (rc) return rc;
  if (obs) RHIP_TRY(hipMemcpyAsync(obs, d_obs, n * 24, hipMemcpyDeviceToHost, st));
  if (final_obs) RHIP_TRY(hipMemcpyAsync(final_obs, d_fin, n * 24, hipMemcpyDeviceToHost, st));
  if (reward) RHIP_TRY(hipMemcpyAsync(reward, d_rew, n * 4, hipMemcpyDeviceToHost, st));
  if (terminated) RHIP_TRY(hipMemcpyAsync(terminated, d_te, n, hipMemcpyDeviceToHost, st));
  if (truncated) RHIP_TRY(hipMemcpyAsync(truncated, d_tr, n, hipMemcpyDeviceToHost, st));
  if (inner_steps) RHIP_TRY(hipMemcpyAsync(inner_steps, d_in, n * 4, hipMemcpyDeviceToHost, st));
  RHIP_TRY(hipStreamSynchronize(st));
  return 0;
}

int salp_robot_vec_get_state(salp_robot_vec_t* h, double* state, uint32_t flags, void* stream) {
  if (!h || !state) return rfail(-1, "handle / state is NULL");
  DeviceScope dev_scope;
  RHIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  // rows are [pitch] on the device and [n] in the snapshot
  RHIP_TRY(hipMemcpy2DAsync(state, (size_t)h->n * sizeof(double), h->S.f, (size_t)h->P.pitch * sizeof(double),
                            (size_t)h->n * sizeof(double), SALP_R_COUNT,
                            (flags & 1u) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
  if (!(flags & 1u)) RHIP_TRY(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
