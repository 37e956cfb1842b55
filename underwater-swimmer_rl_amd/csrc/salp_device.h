// salp_device.h — device-side arithmetic of one SALP swimmer (one env per lane), gfx950.
//
// Restates, for the GPU, the per-step arithmetic of the reference's
//   scripts/utilities/salp_robot.py  ("legacy", SalpRobotEnv.step and helpers, :119-352)
//   src/salp/environments/salp_snake_env.py ("snake", SalpSnakeEnv.step and helpers, :157-428)
//
// Numerics contract:
//   * everything that feeds back into the state (nozzle, breathing, thrust, drag/integration,
//     wall bounce, food placement) is IEEE fp64 in the reference's operation order; this file is
//     compiled with -ffp-contract=off so no multiply-add is fused unless written as fma().
//     The only departures from the reference's fp64 bit pattern are the device sin/cos (<=2 ulp
//     vs glibc) and squared-distance predicates (d^2 < t^2 instead of sqrt(d^2) < t).
//   * quantities that only leave the simulator (observation, reward shaping term) are derived
//     from that fp64 state and finished in fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace salp {

// ---- in-kernel phase stamps (experiment build -DSALP_EXP_STAMPS, profiles/stamp_profile.py) ---------------------
// Shader-clock stamps (s_memtime) at a few phase boundaries of the step, accumulated per wavefront in scalar
// registers and written to a debug buffer of their own after the loop: where a wavefront's lifetime goes, stalls
// included.  No product build executes a stamp.
#ifdef SALP_EXP_STAMPS
struct StampAcc { uint32_t last; uint32_t acc[12]; };
#define SALP_STAMP_PARAM , StampAcc* stamps_ = nullptr
#define SALP_STAMP_PASS , stamps_
#define SALP_STAMP(i) do { if (stamps_) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); \
                                          stamps_->acc[i] += now_ - stamps_->last; stamps_->last = now_; } } while (0)
#else
#define SALP_STAMP_PARAM
#define SALP_STAMP_PASS
#define SALP_STAMP(i) do {} while (0)
#endif

#ifdef SALP_EXP_COUNT     // experiment build: event counters (never in the product)
__device__ unsigned long long salp_exp_counter[8];
#define SALP_COUNT(i, cond) do { if (cond) { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&salp_exp_counter[i], 1ull); } } while (0)
#else
#define SALP_COUNT(i, cond) do {} while (0)
#endif

// ---- packed breathing word: phase[1:0] | timer[9:2] | exhale_dur[17:10] | shape_hold[20:18]
__host__ __device__ inline uint32_t pack_breath(int phase, int timer, int dur, int hold) {
  return (uint32_t)(phase & 3) | ((uint32_t)(timer & 255) << 2) | ((uint32_t)(dur & 255) << 10) |
         ((uint32_t)(hold & 7) << 18);
}
__host__ __device__ inline int bw_phase(uint32_t w) { return (int)(w & 3u); }
__host__ __device__ inline int bw_timer(uint32_t w) { return (int)((w >> 2) & 255u); }
__host__ __device__ inline int bw_dur(uint32_t w) { return (int)((w >> 10) & 255u); }
__host__ __device__ inline int bw_hold(uint32_t w) { return (int)((w >> 18) & 7u); }

// The key words of the draw streams are read through the constant address space: scalar loads (s_load_dwordx2, scalar
// cache) wherever a kernel draws, no vector-memory wait.  They only change between launches (salp_vec_reseed), and the
// scalar cache is invalidated at every kernel start.
typedef const uint32_t __attribute__((address_space(4))) seed_word_t;

// Constants derived on the host in fp64 exactly as the reference derives them.
struct DevParams;
typedef const DevParams __attribute__((address_space(4))) dev_params_c;   // the handle's copy in device memory, read with scalar loads
struct DevParams {
  // geometry / legacy constants
  double W, H, half_W, half_H;
  double margin;            // tank_margin (50)
  double wall_hi_x, wall_hi_y;  // width - margin, height - margin (snake:226,228)
  double R;                 // base_radius
  double a_rest, b_rest;    // R*1.3, R*0.8             (legacy:190-191)
  double ab_full;           // R*1.1                    (legacy:209-210)
  double da_inh, db_inh;    // (end - start) inhaling   (legacy:212-213)
  double da_exh, db_exh;    // (end - start) exhaling   (legacy:245-246)
  double max_nozzle, nozzle_rate, thrust_force, drag, ang_drag;
  double exhale_dur_d;      // (double)exhale_duration
  double food_radius, min_food_dist2;   // 15, 80^2
  double food_xlo, food_xspan, food_ylo, food_yspan;  // random.uniform(lo, hi) = lo + span*u
  double food_reward, collision_penalty, time_penalty, efficiency_bonus, prox_w;
  // fp32 reciprocals for the observation
  double inv_W, inv_H, inv_pi, inv_R, inv_max_nozzle;
  float inv_diag;           // 1/sqrt(W^2+H^2)
  float tie_c0;             // constant term of the fp32 food keys' error bound, 1.4e-7 max(W, H)^2 (salp_food_reg.h)
  int inhale_dur, exhale_dur, cycle_len, max_steps_wo_food;
  int F, K;                 // food slots (num_food_items at creation), max_observed_food
  int F_base;               // base_num_food_items: foods of the next episodes, 0..F (snake:36, :144-148)
  int forced, random_food_count, respawn;
  int autoreset;            // 0: finished envs keep running (salp_config_t.no_autoreset)
  seed_word_t* seed;        // the two key words of the draw streams, in DEVICE memory (the handle's ColdBlock): a kernel
                            // reads them where it draws, so salp_vec_reseed() changes them without changing any launch
                            // parameter — a hipGraph captured before a reseed replays with the new key
  dev_params_c* self;       // this block in DEVICE memory (the handle's ColdBlock): where the STD = false kernels read their
                            // constants, function by function (open_consts below); always valid
  int use_mem;              // 1: read the constants through `self` (set by the kernel on its own copy, a compile-time constant
                            // there; 0 from the host)
  uint64_t env_base;        // global index of local env 0
  int64_t n;                // envs in this handle
  int64_t pitch;            // row pitch (elements) of the SoA state blocks
};

// Compile-time image of the reference's hard-coded constants (legacy:32-53, snake:53-54, 800x600).
// Kernels instantiated with STD = true read these literals (rematerialisable scalar immediates: no
// SGPR pressure, no spills); any other configuration takes the values from DevParams.
struct StdConsts {
  static constexpr double W = 800.0, H = 600.0, half_W = 400.0, half_H = 300.0;
  static constexpr double margin = 50.0, wall_hi_x = 750.0, wall_hi_y = 550.0;
  static constexpr double R = 30.0, a_rest = 39.0, b_rest = 24.0, ab_full = 33.0;
  static constexpr double da_inh = -6.0, db_inh = 9.0, da_exh = 6.0, db_exh = -9.0;
  static constexpr double max_nozzle = 1.0471975511965976, nozzle_rate = 0.05, thrust_force = 100.0;
  static constexpr double drag = 0.98, ang_drag = 0.95, exhale_dur_d = 150.0;
  static constexpr double food_radius = 15.0, min_food_dist2 = 6400.0;
  static constexpr double food_xlo = 65.0, food_xspan = 670.0, food_ylo = 65.0, food_yspan = 470.0;
  static constexpr double inv_W = 1.0 / 800.0, inv_H = 1.0 / 600.0, inv_pi = 1.0 / 3.141592653589793;
  static constexpr double inv_R = 1.0 / 30.0, inv_max_nozzle = 1.0 / 1.0471975511965976;
  static constexpr float inv_diag = 1.0e-3f;
  static constexpr float tie_c0 = 0.0896f;
  static constexpr int inhale_dur = 120, exhale_dur = 150, cycle_len = 330;
};
// CV(name): the constant `name` for this instantiation.  STD = false: ~50 doubles that differ from the literals.  Taken from
// the by-value launch parameter they are all loop-invariant scalars: the compiler loads them once, needs ~100 SGPRs for
// them and spills ~90-115 of those to VGPR lanes, so that nearly every use in the step loop costs a v_readlane (the 8- and
// 12-slot kernels ran 17-25 % behind their literal-constant forms, profiles/r03/ab_notes.md session 15).  Instead every
// function that uses constants opens the handle's device copy through a pointer made opaque there (SALP_CONSTS): the scalar
// loads (s_load_dwordx2..x16, scalar cache) stay in that function, and the registers are free again after it.
// Measured (ab_pc_f*.json): 4- and 8-slot kernels -5 ... -15 % (125 VGPRs, four wavefronts per SIMD again); one-food +1.4 %,
// 12- and 16-slot kernels +13 % (two / three wavefronts per SIMD do not hide the scalar-load waits) — so a kernel chooses:
// it hands its functions a parameter block whose `use_mem` is a compile-time 0 to keep the by-value constants.
template <bool STD>
__device__ __forceinline__ dev_params_c* open_consts(const DevParams& P) {
  if constexpr (STD) return nullptr;
  else {
    // `use_mem` is a literal in the kernel's own copy of the parameters and every caller is __forceinline__: the test folds.
    // (Should it ever not fold, it is a wave-uniform run-time test and both forms stay correct: `self` is always valid.)
    if (!P.use_mem) return nullptr;
    // (the rare paths hand in the memory copy itself: its `self` word may arrive through a vector load)
    const uint64_t a = (uint64_t)(uintptr_t)P.self;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    dev_params_c* p = (dev_params_c*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    asm volatile("" : "+s"(p));
    __builtin_assume(p != nullptr);      // CV() then drops its by-value alternative
    return p;
  }
}
#define SALP_CONSTS [[maybe_unused]] dev_params_c* const PC_ = open_consts<STD>(P)
#define CV(name) (STD ? StdConsts::name : (PC_ ? PC_->name : P.name))

// SoA state rows (device layout; distinct from the public snapshot layout)
enum { SF_X = 0, SF_Y, SF_VX, SF_VY, SF_TH, SF_OM, SF_NOZ, SF_WATER, SF_EPRET, SF_FOOD0 };
enum { SI_PACKED = 0, SI_SSF, SI_FC, SI_RNG, SI_EPLEN, SI_COUNT };

struct DevState {
  double* f;     // [SF_FOOD0 + 2F][pitch]
  int32_t* i;    // [SI_COUNT][pitch]
};

struct DevStats {  // one replica = 16 x 8 B (one 128-B line)
  unsigned long long v[16];
};
enum { ST_STEPS = 0, ST_EPISODES, ST_TERM, ST_TRUNC, ST_COLL, ST_FOOD, ST_EPLEN, ST_REWARD, ST_EPRET };
#define SALP_STATS_REPLICAS 64
#define SALP_FIXED_SCALE 1048576.0  /* 2^20 */

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
// (gfx950 has no v_xor3_b32: the two xors per word stay two instructions.)
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  // ((hi>>5)*2^26 + (lo>>6)) / 2^53 : every operation is exact
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * 1.1102230246251565e-16;
}

__device__ __forceinline__ double pymax(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double pymin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ bool is_none(double fx) { return fx != fx; }

#define SALP_PI 3.141592653589793
#define SALP_2PI 6.283185307179586
#define SALP_PIO2 1.5707963267948966

// Everything of one env except the food positions.
struct EnvCore {
  double x, y, vx, vy, th, om, noz, water, epret;
  uint32_t packed;
  int ssf, fc, eplen;
  uint32_t rng;
};
// One env with its food positions in registers (all indexing static).  The multi-food rollout kernel
// keeps the foods in LDS instead (salp_food_lds.h) and only an EnvCore in registers.
template <int FMAX>
struct Env : EnvCore {
  double fx[FMAX], fy[FMAX];
};

__device__ __forceinline__ void load_core(EnvCore& e, const DevState& S, const DevParams& P, int64_t i) {
  const int64_t p = P.pitch;
  e.x = S.f[SF_X * p + i]; e.y = S.f[SF_Y * p + i];
  e.vx = S.f[SF_VX * p + i]; e.vy = S.f[SF_VY * p + i];
  e.th = S.f[SF_TH * p + i]; e.om = S.f[SF_OM * p + i];
  e.noz = S.f[SF_NOZ * p + i]; e.water = S.f[SF_WATER * p + i];
  e.epret = S.f[SF_EPRET * p + i];
  e.packed = (uint32_t)S.i[SI_PACKED * p + i];
  e.ssf = S.i[SI_SSF * p + i]; e.fc = S.i[SI_FC * p + i];
  e.rng = (uint32_t)S.i[SI_RNG * p + i]; e.eplen = S.i[SI_EPLEN * p + i];
}
__device__ __forceinline__ void store_core(const EnvCore& e, const DevState& S, const DevParams& P, int64_t i) {
  const int64_t p = P.pitch;
  S.f[SF_X * p + i] = e.x; S.f[SF_Y * p + i] = e.y;
  S.f[SF_VX * p + i] = e.vx; S.f[SF_VY * p + i] = e.vy;
  S.f[SF_TH * p + i] = e.th; S.f[SF_OM * p + i] = e.om;
  S.f[SF_NOZ * p + i] = e.noz; S.f[SF_WATER * p + i] = e.water;
  S.f[SF_EPRET * p + i] = e.epret;
  S.i[SI_PACKED * p + i] = (int32_t)e.packed;
  S.i[SI_SSF * p + i] = e.ssf; S.i[SI_FC * p + i] = e.fc;
  S.i[SI_RNG * p + i] = (int32_t)e.rng; S.i[SI_EPLEN * p + i] = e.eplen;
}

template <int FMAX>
__device__ __forceinline__ void load_env(Env<FMAX>& e, const DevState& S, const DevParams& P, int64_t i) {
  const int64_t p = P.pitch;
  load_core(e, S, P, i);
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    if (k < P.F) {
      e.fx[k] = S.f[(SF_FOOD0 + k) * p + i];
      e.fy[k] = S.f[(SF_FOOD0 + P.F + k) * p + i];
    } else {
      e.fx[k] = __builtin_nan(""); e.fy[k] = __builtin_nan("");
    }
  }
}

template <int FMAX>
__device__ __forceinline__ void store_env(const Env<FMAX>& e, const DevState& S, const DevParams& P, int64_t i) {
  const int64_t p = P.pitch;
  store_core(e, S, P, i);
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    if (k < P.F) {
      S.f[(SF_FOOD0 + k) * p + i] = e.fx[k];
      S.f[(SF_FOOD0 + P.F + k) * p + i] = e.fy[k];
    }
  }
}

// One Philox block of this env's draw stream (include/salp_vec.h "Randomness").
__device__ __forceinline__ U4 next_block(EnvCore& e, const DevParams& P, uint64_t genv) {
  // The key is made opaque here: otherwise the 18 round keys (seed + r * Weyl constant) are hoisted out of the
  // step loop as loop invariants, spilled to VGPR lanes and read back with v_readlane on every use — two
  // scalar adds per round in place are cheaper.
  uint32_t k0 = P.seed[0], k1 = P.seed[1];
  asm volatile("" : "+s"(k0), "+s"(k1));
  U4 w = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), e.rng, 0u, k0, k1);
  e.rng += 1u;
  return w;
}

template <int FMAX, bool STD>
__device__ __forceinline__ void draw_xy(EnvCore& e, const DevParams& P, uint64_t genv, double& x, double& y) {
  SALP_CONSTS;
  const U4 w = next_block(e, P, genv);
  x = CV(food_xlo) + CV(food_xspan) * u53(w.x, w.y);
  y = CV(food_ylo) + CV(food_yspan) * u53(w.z, w.w);
}

// ------------------------------------------------------------------ fp64 trigonometry
// sin and cos of |x| <= ~6 (thrust angles are bounded by pi + pi/3 + pi/2): Cody-Waite reduction by
// pi/2 in two pieces (k <= 4, so k*PIO2_1 is exact) and the fdlibm kernel polynomials on
// [-pi/4, pi/4].  <= 1 ulp-class accuracy; explicit fma() is allowed here because these values
// have no bit-exact counterpart on the CPU anyway (glibc's sin/cos are a different algorithm).
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
  const double fn = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632673412561417e+00, x);
  r = fma(-fn, 6.07710050650619224932e-11, r);
  const double z = r * r;
  // kernel sin
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  const double v = z * r;
  const double sr = fma(v, fma(z, ps, -1.66666666666666324348e-01), r);
  // kernel cos
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = (int)fn & 3;
  const double s0 = (q & 1) ? cr : sr;
  const double c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// sin(x) for |x| <= pi/3 (the nozzle angle): odd Taylor polynomial to x^21, error < 2e-22.
__device__ __forceinline__ double sin_nozzle(double x) {
  const double z = x * x;
  double p = fma(z, 1.9572941063391263e-20, -8.2206352466243295e-18);   // 1/21!, -1/19!
  p = fma(z, p, 2.8114572543455206e-15);    //  1/17!
  p = fma(z, p, -7.6471637318198164e-13);   // -1/15!
  p = fma(z, p, 1.6059043836821613e-10);    //  1/13!
  p = fma(z, p, -2.5052108385441720e-08);   // -1/11!
  p = fma(z, p, 2.7557319223985893e-06);    //  1/9!
  p = fma(z, p, -1.9841269841269841e-04);   // -1/7!
  p = fma(z, p, 8.3333333333333332e-03);    //  1/5!
  p = fma(z, p, -1.6666666666666666e-01);   // -1/3!
  return fma(x * z, p, x);
}

// ---- thrust-block constants through the scalar cache ------------------------------------------------
// gfx950 has no 64-bit literals: every fp64 constant of a kernel is an s_mov_b32 pair (or a v_mov_b32 pair
// when it is the addend of a v_fmac_f64), and the wavefront issues ONE instruction of any kind per 4 cycles
// (SQ counters, profiles/r02/ab_notes.md): in the 12-food kernel ~250 of ~1200 instructions per step were
// such moves, 99 of them in the thrust block.  The constants of the thrust block therefore sit in a table in
// constant memory and are fetched with scalar loads — one s_load_dwordx8/x16 brings 4 / 8 of them into SGPRs,
// which v_fma_f64 / v_mul_f64 take directly as an operand.  The table pointer is made opaque at the point of
// use so that the loads stay inside the (conditional) thrust block instead of being hoisted across the step
// loop, where they would only be spilled.  Values and evaluation order are those of sincos_small / sin_nozzle
// above: results are bit-identical.
typedef const double __attribute__((address_space(4))) tbl_double;
enum {
  TT_2OPI = 0, TT_PIO2_1, TT_PIO2_1T, TT_S5, TT_S4, TT_S3, TT_S2, TT_S1,          // sincos reduction, sin kernel
  TT_S0, TT_C6, TT_C5, TT_C4, TT_C3, TT_C2, TT_C1, TT_PAD0,                        // cos kernel
  TT_N10, TT_N9, TT_N8, TT_N7, TT_N6, TT_N5, TT_N4, TT_N3,                         // sin_nozzle
  TT_N2, TT_N1, TT_K012, TT_K0002, TT_K07, TT_K00005, TT_K00003, TT_PIO2,          // thrust scalings
  TT_DL, TT_K03, TT_K008, TT_K005, TT_J_S3, TT_J_S2, TT_J_S1, TT_J_S0,             // side thrust, jitter sin
  TT_J_C3, TT_J_C2, TT_J_C1, TT_K004, TT_K0002J, TT_K04, TT_COUNT
};
__constant__ double kThrustTbl[TT_COUNT + 2] = {
  6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11,
  1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
  8.33333333332248946124e-03,
  -1.66666666666666324348e-01, -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07,
  2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02, 0.0,
  1.9572941063391263e-20, -8.2206352466243295e-18, 2.8114572543455206e-15, -7.6471637318198164e-13,
  1.6059043836821613e-10, -2.5052108385441720e-08, 2.7557319223985893e-06, -1.9841269841269841e-04,
  8.3333333333333332e-03, -1.6666666666666666e-01, 0.012, 0.0002, 0.7, 0.00005, 0.00003, 1.5707963267948966,
  -6.123233995736766e-17, 0.3, 0.008, 0.05, 2.7557319223985893e-06, -1.9841269841269841e-04, 8.3333333333333332e-03,
  -1.6666666666666666e-01,
  2.4801587301587302e-05, -1.3888888888888889e-03, 4.1666666666666664e-02, 0.04, 0.002, 0.4, 0.0, 0.0
};
__device__ __forceinline__ tbl_double* thrust_table() {
  tbl_double* t = (tbl_double*)kThrustTbl;
  asm volatile("" : "+s"(t));
  return t;
}
// a * b + c and a * c with the constant c in an SGPR pair, written as instructions: left to itself the
// compiler copies a scalar addend into VGPRs (two v_mov_b32) to use the two-address v_fmac_f64.
__device__ __forceinline__ double fma_s(double a, double b, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
  return r;
}
__device__ __forceinline__ double mul_s(double a, double c_uniform) {
  double r;
  asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c_uniform));
  return r;
}
__device__ __forceinline__ void sincos_small_t(tbl_double* T, double x, double& s, double& c) {
  const double fn = __builtin_rint(mul_s(x, T[TT_2OPI]));
  const double nfn = -fn;
  double r = fma(nfn, T[TT_PIO2_1], x);
  r = fma(nfn, T[TT_PIO2_1T], r);
  const double z = r * r;
  double ps = fma_s(z, (double)T[TT_S5], T[TT_S4]);
  ps = fma_s(z, ps, T[TT_S3]);
  ps = fma_s(z, ps, T[TT_S2]);
  ps = fma_s(z, ps, T[TT_S1]);
  const double v = z * r;
  const double sr = fma(v, fma_s(z, ps, T[TT_S0]), r);
  double pc = fma_s(z, (double)T[TT_C6], T[TT_C5]);
  pc = fma_s(z, pc, T[TT_C4]);
  pc = fma_s(z, pc, T[TT_C3]);
  pc = fma_s(z, pc, T[TT_C2]);
  pc = fma_s(z, pc, T[TT_C1]);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
  const int q = (int)fn & 3;
  const double s0 = (q & 1) ? cr : sr;
  const double c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ double sin_nozzle_t(tbl_double* T, double x) {
  const double z = x * x;
  double p = fma_s(z, (double)T[TT_N10], T[TT_N9]);
  p = fma_s(z, p, T[TT_N8]);
  p = fma_s(z, p, T[TT_N7]);
  p = fma_s(z, p, T[TT_N6]);
  p = fma_s(z, p, T[TT_N5]);
  p = fma_s(z, p, T[TT_N4]);
  p = fma_s(z, p, T[TT_N3]);
  p = fma_s(z, p, T[TT_N2]);
  p = fma_s(z, p, T[TT_N1]);
  return fma(x * z, p, x);
}

// Ellipse semi-axes implied by the post-step state (see SALP_I_SHAPE_HOLD in salp_vec.h).
template <bool STD>
__device__ __forceinline__ void shape_of(const DevParams& P, uint32_t packed, double water, double& a, double& b) {
  SALP_CONSTS;
  const int phase = bw_phase(packed), timer = bw_timer(packed), dur = bw_dur(packed), hold = bw_hold(packed);
  if (hold == 7) { a = CV(R); b = CV(R); return; }
  if (hold != 0) {
    const double p = (double)hold / (double)CV(inhale_dur);
    a = CV(a_rest) + CV(da_inh) * p; b = CV(b_rest) + CV(db_inh) * p; return;
  }
  if (phase == 0) { a = CV(a_rest); b = CV(b_rest); }
  else if (phase == 1 || timer == 0) { a = CV(a_rest) + CV(da_inh) * water; b = CV(b_rest) + CV(db_inh) * water; }
  else {
    const double p = (double)timer / (double)dur;
    a = CV(ab_full) + CV(da_exh) * p; b = CV(ab_full) + CV(db_exh) * p;
  }
}

// Food placement, shared by reset (snake:92-131 _generate_food_positions) and respawn
// (snake:232-276 _respawn_food).  Both are the same rejection sampler: draw (x, y); reject if within
// min_food_distance of a live food or of the robot — at reset the robot sits at the tank centre,
// which is the point snake:115 tests against — and after `limit` rejections (100 at reset, 50 at
// respawn) accept the next draw unconditionally.  The accepted point fills the first empty slot
// (at reset slots fill in order, snake:120; at respawn snake:261-264).  `todo` = foods still to place.
template <int FMAX, bool STD>
__device__ __forceinline__ void place_food(Env<FMAX>& e, const DevParams& P, uint64_t genv, int todo, int limit) {
  SALP_CONSTS;
  int attempts = 0;
#pragma unroll 1
  while (__any(todo > 0)) {
    if (todo > 0) {
      double x, y;
      draw_xy<FMAX, STD>(e, P, genv, x, y);
      bool valid = true;
      {
        const double dx = x - e.x, dy = y - e.y;
        if (dx * dx + dy * dy < CV(min_food_dist2)) valid = false;
      }
#pragma unroll
      for (int k = 0; k < FMAX; ++k) {
        const double dx = x - e.fx[k], dy = y - e.fy[k];
        if (dx * dx + dy * dy < CV(min_food_dist2)) valid = false;   // NaN (empty) slots never reject
      }
      if (valid || attempts >= limit) {
        bool put = false;
#pragma unroll
        for (int k = 0; k < FMAX; ++k) {
          if (!put && k < P.F && is_none(e.fx[k])) { e.fx[k] = x; e.fy[k] = y; put = true; }
        }
        todo -= 1;
        attempts = 0;
      } else {
        attempts += 1;
      }
    }
  }
}

// legacy:95-117 reset + snake:133-149 (pose, breathing, counters, episode food count); the food
// itself is placed by place_food(e, ..., todo = return value, limit = 100).
// F_base: base_num_food_items of this launch (the one constant a caller may change between launches).
template <bool STD>
__device__ __forceinline__ int reset_core(EnvCore& e, const DevParams& P, uint64_t genv, int F_base) {
  SALP_CONSTS;
  e.x = CV(half_W); e.y = CV(half_H); e.vx = 0.0; e.vy = 0.0; e.th = 0.0; e.om = 0.0;
  e.noz = 0.0; e.water = 0.0; e.epret = 0.0;
  e.packed = pack_breath(0, 0, bw_dur(e.packed), 7);
  e.ssf = 0; e.fc = 0; e.eplen = 0;
  int nf = F_base;
  if (P.random_food_count) {  // snake:146 random.randint(1, max(1, base))
    const U4 w = next_block(e, P, genv);
    const uint32_t n = (uint32_t)(F_base > 1 ? F_base : 1);
    nf = 1 + (int)(((uint64_t)w.x * (uint64_t)n) >> 32);
    if (nf > P.F) nf = P.F;
  }
  return nf;
}
template <int FMAX, bool STD>
__device__ __forceinline__ int reset_pose(Env<FMAX>& e, const DevParams& P, uint64_t genv, int F_base) {
  const int nf = reset_core<STD>(e, P, genv, F_base);
#pragma unroll
  for (int k = 0; k < FMAX; ++k) { e.fx[k] = __builtin_nan(""); e.fy[k] = __builtin_nan(""); }
  return nf;
}

// legacy:261-314 _apply_jet_thrust (water is the value BEFORE this step's decay).
// The reference evaluates cos/sin at three angles: phi, fl(phi + pi/2) and fl(phi + jitter).  One
// sincos(phi) serves all three: the side angle is a quarter turn plus the (exactly recovered)
// rounding error of the addition, the jitter angle a rotation by |D| <= 0.025 (short series).
// The seven increments of one thrust step, in the order legacy:261-314 applies them:
//   vx = ((vx + t.ax) + t.bx) + t.cx,  vy likewise,  omega = omega + t.om
// (main jet, side thrust, jitter).  A function of (theta, nozzle, water before this step's decay, r = max(a, b),
// the env's draw counter and global index) only.
struct ThrustTerms { double ax, ay, bx, by, cx, cy, om; };
template <bool STD>
__device__ __forceinline__ ThrustTerms jet_thrust_terms(double th, double noz, double water, double r, uint32_t rng,
                                                        uint64_t genv, const DevParams& P) {
  SALP_CONSTS;
  ThrustTerms t;
  tbl_double* TT = thrust_table();
  const double T = mul_s(CV(thrust_force) * water, TT[TT_K04]);
  const double phi = th - noz;
  double s, c;
  sincos_small_t(TT, phi, s, c);
  t.ax = mul_s(c * T, TT[TT_K012]);
  t.ay = mul_s(s * T, TT[TT_K012]);
  const double nn = -noz;
  const double primary = mul_s(nn * T, TT[TT_K0002]);
  const double arm = r * TT[TT_K07];
  const double perp = T * sin_nozzle_t(TT, nn);
  const double moment = mul_s(perp * arm, TT[TT_K00005]);
  const double shape = mul_s((nn * T) * water, TT[TT_K00003]);
  t.om = (primary + moment) + shape;
  {  // side thrust at fl(phi + fl(pi/2)) = phi + pi/2 + dl,  dl = fl(pi/2) - pi/2 (6.1e-17) up to the rounding of the sum
     // (<= 4.4e-16: 4e-17 px per step through the 0.1-px side term, below the device sincos' own last-bit departure from
     // glibc — round 2 recovered it with a TwoSum, six fp64 instructions per thrust step)
    const double dl = TT[TT_DL];
    const double sc = -fma(dl, c, s);     // cos(side) = -sin(phi + dl)
    const double ss = fma(-dl, s, c);     // sin(side) =  cos(phi + dl)
    const double S = mul_s(T * fabs(noz), TT[TT_K03]);
    t.bx = mul_s(sc * S, TT[TT_K008]);
    t.by = mul_s(ss * S, TT[TT_K008]);
  }
  {  // jitter at fl(phi + d), d = fl((u - 0.5) * 0.05); one Philox block of the env's draw stream (counter rng)
    uint32_t k0 = P.seed[0], k1 = P.seed[1], g0 = (uint32_t)genv;
    asm volatile("" : "+s"(k0), "+s"(k1));    // see next_block
    asm volatile("" : "+v"(g0));              // likewise the first-round product 0xD2511F53 * env_lo: one v_mad_u64_u32 here
                                              // instead of a 64-bit VGPR pair held across the step loop
    const U4 w = philox4x32_10(g0, (uint32_t)(genv >> 32), rng, 0u, k0, k1);
    const double u = u53(w.x, w.y);
    const double d = mul_s(u - 0.5, TT[TT_K005]);
    const double D = d;                   // fl(phi + d) = phi + D up to the sum's rounding (<= 4.4e-16 rad on a 3e-3-px term)
    const double z = D * D;
    double sp = fma_s(z, (double)TT[TT_J_S3], TT[TT_J_S2]);
    sp = fma_s(z, sp, TT[TT_J_S1]);
    sp = fma_s(z, sp, TT[TT_J_S0]);
    const double sD = fma(D * z, sp, D);
    double cp = fma_s(z, (double)TT[TT_J_C3], TT[TT_J_C2]);
    cp = fma_s(z, cp, TT[TT_J_C1]);
    cp = fma(z, cp, -0.5);
    const double cD = fma(z, cp, 1.0);
    const double nc = c * cD - s * sD;
    const double ns = s * cD + c * sD;
    const double nf = T * TT[TT_K004];
    t.cx = mul_s(nc * nf, TT[TT_K0002J]);
    t.cy = mul_s(ns * nf, TT[TT_K0002J]);
  }
  return t;
}
__device__ __forceinline__ void apply_thrust_terms(EnvCore& e, const ThrustTerms& t) {
  e.vx = ((e.vx + t.ax) + t.bx) + t.cx;
  e.vy = ((e.vy + t.ay) + t.by) + t.cy;
  e.om = e.om + t.om;
  e.rng += 1u;
}
// legacy:261-314 _apply_jet_thrust (water is the value BEFORE this step's decay), the lane serving itself.
template <int FMAX, bool STD>
__device__ __forceinline__ void apply_jet_thrust(EnvCore& e, const DevParams& P, uint64_t genv, double r) {
  apply_thrust_terms(e, jet_thrust_terms<STD>(e.th, e.noz, e.water, r, e.rng, genv, P));
}

// (Two experiments that lived here are gone from the source, their measurements are in profiles/r02/ab_notes.md and
// their code in the history: a block-pooled thrust — one wavefront evaluating the thrust of a workgroup's four
// between two barriers, bit-identical and 35 % slower, session 6 — and the step's fp64 constants pinned in VGPRs —
// -1.8 % at equal residency, +5 % once its 217 VGPRs cost the third wavefront per SIMD, session 3.)

// Geometry of the nearest live food (first minimum, snake:350-364) in fp64 state terms.
struct Nearest {
  double dx, dy, d2;
  bool any;
};
template <int FMAX>
__device__ __forceinline__ Nearest nearest_food(const Env<FMAX>& e) {
  Nearest g;
  g.dx = 0.0; g.dy = 0.0; g.d2 = 0.0; g.any = false;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    const double dx = e.fx[k] - e.x, dy = e.fy[k] - e.y;
    const double d2 = dx * dx + dy * dy;
    if (!is_none(e.fx[k]) && (!g.any || d2 < g.d2)) { g.any = true; g.d2 = d2; g.dx = dx; g.dy = dy; }
  }
  return g;
}

// fp32 helpers for quantities that only leave the simulator (observation angle, reward shaping).  Three food bearings per
// step were 110 of the 12-food kernel's ~590 VALU instructions per wavefront-step (round 2: degree-17 polynomial, Newton
// step on the reciprocal, compare / select unfolding and wrapping); the contract is 1e-5 on angle / pi.
// atan2 for the food bearing: t = min / max through v_rcp_f32 (1 ulp), odd minimax polynomial on [0, 1] (Remez fits),
// octant / quadrant unfolding, sign by bit copy.  PRECISE = false (bearings that only go to the observation): degree 11,
// max error 1.7e-6 rad = 5.3e-7 of the observation's unit.  PRECISE = true (the nearest food's bearing, which the reward
// multiplies by proximity_reward_weight, 5 in single_food.yaml): degree 13, 2.5e-7 rad.
template <bool PRECISE>
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1.0e-30f);   // v_max3_f32; atan2(0, 0) = 0 (math.atan2): t = 0 * 1e30
  const float mn = __builtin_fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float z = t * t;
  float p;
  if (PRECISE) {
    p = fmaf(z, 0.006811792962253094f, -0.0336042195558548f);
    p = fmaf(z, p, 0.07962366938591003f);
    p = fmaf(z, p, -0.1323334276676178f);
    p = fmaf(z, p, 0.19807815551757812f);
    p = fmaf(z, p, -0.3331736922264099f);
    p = fmaf(z, p, 0.9999961256980896f);
  } else {
    p = fmaf(z, -0.01171913556754589f, 0.05264735221862793f);
    p = fmaf(z, p, -0.116426482796669f);
    p = fmaf(z, p, 0.19354037940502167f);
    p = fmaf(z, p, -0.33262282609939575f);
    p = fmaf(z, p, 0.9999772310256958f);
  }
  float a = t * p;
  if (ay > ax) a = 1.57079632679489662f - a;
  if (x < 0.0f) a = 3.14159265358979324f - a;
  return __builtin_copysignf(a, y);
}

// cos(x) for |x| <= pi (a wrapped heading): fold to [0, pi/2], even Taylor polynomial to x^12 (error 6.5e-9).
__device__ __forceinline__ float cos_wrapped(float x) {
  float ax = fabsf(x);
  const bool flip = ax > 1.57079632679489662f;
  if (flip) ax = 3.14159265358979324f - ax;
  const float z = ax * ax;
  float p = fmaf(z, 2.08767569878681e-09f, -2.755731922398589e-07f);
  p = fmaf(z, p, 2.48015873015873e-05f);
  p = fmaf(z, p, -1.3888888888888889e-03f);
  p = fmaf(z, p, 4.1666666666666664e-02f);
  p = fmaf(z, p, -0.5f);
  const float c = fmaf(z, p, 1.0f);
  return flip ? -c : c;
}

// Heading of a food relative to the body axis, wrapped to [-pi, pi] (fp32): both angles are in [-pi, pi], so one multiple
// of 2 pi at most is off — round-to-nearest-even of rel / 2 pi is 0 on the closed interval, as snake:403-407's strict loops.
template <bool PRECISE = false>
__device__ __forceinline__ float relative_heading(float dy, float dx, float th) {
  const float rel = atan2_fast<PRECISE>(dy, dx) - th;
  return fmaf(__builtin_rintf(rel * 0.15915494309189535f), -6.28318530717959f, rel);
}

struct StepOut {
  double rmax;    // max(ellipse_a, ellipse_b) of this step
  float reward;
  float rel;      // relative heading of the nearest food used by the reward (valid if rel_valid)
  bool rel_valid; // the reward evaluated `rel` for the food set that observe() will also see
  bool terminated, truncated, collision, collected;
};

// One reference step (legacy:119-156 under snake:157-189), without autoreset / observation.
// The respawn of a collected food (snake:179-180) is left to the caller — place_food(todo = 1,
// limit = 50) when o.collected && P.respawn, BEFORE any autoreset so the draw order of the
// reference is kept.  (The all-collected termination test only applies when !P.respawn.)
// legacy:119-156 up to and including the wall bounce; returns r = max(ellipse_a, ellipse_b) of this step.
template <bool FORCED, bool STD>
__device__ __forceinline__ double step_head(EnvCore& e, const DevParams& P, uint64_t genv, float a0, float a1 SALP_STAMP_PARAM) {
  SALP_CONSTS;
  int phase = bw_phase(e.packed), timer = bw_timer(e.packed), dur = bw_dur(e.packed);
  // legacy:121-135
  double nd;
  bool inhaling;
  if (FORCED) {
    nd = (double)a0;
    const int tm = (CV(cycle_len) > 255) ? timer : (timer % CV(cycle_len));
    inhaling = tm < CV(inhale_dur);
  } else {
    inhaling = a0 > 0.5f;
    nd = (double)a1;
  }
  const double max_noz = CV(max_nozzle), noz_rate = CV(nozzle_rate);
  const double target = nd * max_noz;
  // legacy:169-182 _update_nozzle
  {
    const double diff = target - e.noz;
    double nz;
    if (fabs(diff) > noz_rate) nz = (diff > 0) ? (e.noz + noz_rate) : (e.noz - noz_rate);
    else nz = target;
    e.noz = pymax(-max_noz, pymin(max_noz, nz));
  }
  // legacy:184-259 _update_breathing_cycle
  double a, b;
  int hold = 0;
  bool thrust = false;
  double water_next = e.water;
  {
    const int tnew = timer + 1;
    const double den = (phase == 2) ? (double)dur : (double)CV(inhale_dur);
    // p = tnew / den, correctly rounded, without the fp64 division sequence (~14 VALU incl. v_rcp_f64): with
    // y = RN(1 / den), q0 = RN(tnew y), r = tnew - den q0 (exact in an fma), RN(q0 + r y) is the IEEE quotient
    // (Markstein) — checked exhaustively for every den in 1..255 and tnew in 1..257.  y is a constant for the
    // inhale duration and for a full exhale (dur = exhale_duration, every exhale of forced breathing); lanes
    // whose exhale was cut short (dur from the water level, legacy:223) divide once to get theirs.
    double yden = (phase == 2) ? (1.0 / (double)CV(exhale_dur)) : (1.0 / (double)CV(inhale_dur));
    {
      const bool odd = (phase == 2) && (dur != CV(exhale_dur));
      if (__any(odd)) {
        asm volatile("" ::: "memory");   // keeps the division inside the (wave-uniform) branch: not speculated
        if (odd) yden = 1.0 / (double)dur;
      }
    }
    const double a_rest = CV(a_rest), b_rest = CV(b_rest), da_inh = CV(da_inh), db_inh = CV(db_inh);
    const double ab_full = CV(ab_full), da_exh = CV(da_exh), db_exh = CV(db_exh);
    const double tn = (double)tnew;
    const double q0 = tn * yden;
    const double p = fma(fma(-den, q0, tn), yden, q0);   // == (double)tnew / den, tests/test_source_claims.py
    if (phase == 0) {
      a = a_rest; b = b_rest;
      if (inhaling) { phase = 1; timer = 0; }
    } else if (phase == 1) {
      if (inhaling && timer < CV(inhale_dur)) {
        timer = tnew;
        a = a_rest + da_inh * p; b = b_rest + db_inh * p;
        water_next = p;
      } else {
        a = a_rest + da_inh * e.water; b = b_rest + db_inh * e.water;  // unchanged ellipse
        if (e.water > 0.05) {
          phase = 2; timer = 0;
          dur = (int)(CV(exhale_dur_d) * pymax(e.water, 0.3));
        } else {
          hold = (timer >= 1 && timer <= 6) ? timer : 0;
          phase = 0; timer = 0; water_next = 0.0;
        }
      }
    } else {
      if (p <= 1.0) {
        timer = tnew;
        a = ab_full + da_exh * p; b = ab_full + db_exh * p;
        thrust = (0.1 <= p) && (p <= 0.5);
        const double v = e.water * (1.0 - p);
        water_next = (v > 0) ? v : 0.0;
      } else {
        a = ab_full + da_exh * 1.0; b = ab_full + db_exh * 1.0;  // ellipse of the last exhale step
        phase = 0; timer = 0; water_next = 0.0;
      }
    }
  }
  // max(ellipse_a, ellipse_b) is ellipse_a whenever both come from the formulas above with the reference's radii:
  // a - b = R (0.5 - 0.5 p) inhaling, R 0.5 p exhaling, 0.5 R at rest, all >= 0 and exact in fp64 for R = 30 (a and b
  // are multiples of 2^-46 below 64: every product and sum above is exact or rounds both the same way is NOT assumed —
  // checked for every (phase, timer, duration) of the literal constants by tests/test_source_claims.py).  Other radii
  // keep the maximum.
  const double r = STD ? a : pymax(a, b);
#ifdef SALP_EXP_NO_THRUST      // experiment build (profiles/ab_bench.py): price of the thrust block
  thrust = false;
#endif
  SALP_STAMP(1);
  if (thrust) apply_jet_thrust<1, STD>(e, P, genv, r);
  SALP_STAMP(2);
  e.water = water_next;
  e.packed = pack_breath(phase, timer, dur, hold);
  // legacy:316-352 _update_physics
  e.vx = e.vx * CV(drag); e.vy = e.vy * CV(drag); e.om = e.om * CV(ang_drag);
  e.x = e.x + e.vx; e.y = e.y + e.vy; e.th = e.th + e.om;
  // legacy:329-332 `while theta > pi: theta -= 2 pi` / `while theta < -pi: ...`.  |omega| is far below
  // 2 pi, so one conditional step each is the common case; the (bounded) loops only run if a lane
  // is still outside, e.g. after an injected state.
  const double pi = SALP_PI, twopi = SALP_2PI;
  if (e.th > pi) e.th -= twopi;
  if (e.th < -pi) e.th += twopi;
  if (__any(e.th > pi || e.th < -pi)) {
#pragma unroll 1
    for (int it = 0; it < 8 && e.th > pi; ++it) e.th -= twopi;
#pragma unroll 1
    for (int it = 0; it < 8 && e.th < -pi; ++it) e.th += twopi;
  }
  {
    const double m = CV(margin) + r;
    const double hx = CV(W) - m, hy = CV(H) - m;
    const double k04 = 0.4, k07 = 0.7;
    if (e.x < m) { e.x = m; e.vx = fabs(e.vx) * k04; e.om = e.om * k07; }
    else if (e.x > hx) { e.x = hx; e.vx = -fabs(e.vx) * k04; e.om = e.om * k07; }
    if (e.y < m) { e.y = m; e.vy = fabs(e.vy) * k04; e.om = e.om * k07; }
    else if (e.y > hy) { e.y = hy; e.vy = -fabs(e.vy) * k04; e.om = e.om * k07; }
  }
  SALP_STAMP(3);
  return r;
}

// snake:171-189 counters and termination, given the step's reward so far (everything but the counters)
// and whether any food is still alive (only read when !P.respawn).
__device__ __forceinline__ void step_tail(EnvCore& e, const DevParams& P, StepOut& o, double rew, bool any_alive) {
  e.ssf += 1;
  if (o.collected) {
    e.fc += 1;
    e.ssf = 0;
    o.rel_valid = false;  // the food set changed after the reward was computed
  }
  o.terminated = false; o.truncated = false;
  if (o.collision) o.terminated = true;
  else if (e.ssf > P.max_steps_wo_food) o.truncated = true;
  else if (!P.respawn && !any_alive) o.terminated = true;
  e.eplen += 1;
  e.epret += rew;
  o.reward = (float)rew;
}

template <int FMAX, bool FORCED, bool STD>
__device__ __forceinline__ StepOut step_env(Env<FMAX>& e, const DevParams& P, uint64_t genv, float a0, float a1 SALP_STAMP_PARAM) {
  SALP_CONSTS;
  const double r = step_head<FORCED, STD>(e, P, genv, a0, a1 SALP_STAMP_PASS);
  StepOut o;
  o.rmax = r;
  // snake:204-217 _check_food_collection (first live food inside the capture radius)
  o.collected = false;
  {
    const double cr = r + CV(food_radius);
    const double cr2 = cr * cr;
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      const double dx = e.x - e.fx[k], dy = e.y - e.fy[k];
      if constexpr (FMAX == 1) {
        if (!o.collected && (dx * dx + dy * dy < cr2)) {
          o.collected = true; e.fx[k] = __builtin_nan(""); e.fy[k] = __builtin_nan("");
        }
      } else {   // multi-food: selects instead of branches
        const bool hit = !o.collected && (dx * dx + dy * dy < cr2);
        e.fx[k] = hit ? __builtin_nan("") : e.fx[k];
        e.fy[k] = hit ? __builtin_nan("") : e.fy[k];
        o.collected = o.collected || hit;
      }
    }
  }
  // snake:219-230 _check_wall_collision
  o.collision = (e.x - r <= CV(margin)) || (e.x + r >= CV(wall_hi_x)) || (e.y - r <= CV(margin)) || (e.y + r >= CV(wall_hi_y));
  // snake:278-327 _calculate_snake_reward
  double rew = 0.0;
  if (o.collected) {
    rew += P.food_reward;
    if (P.efficiency_bonus > 0) rew += P.efficiency_bonus * (double)(P.max_steps_wo_food - e.ssf);
  }
  if (o.collision) rew += P.collision_penalty;
  o.rel = 0.f; o.rel_valid = false;
  if (P.prox_w > 0) {
    const Nearest g = nearest_food(e);
    if (g.any) {
      // alignment = cos(wrap(atan2(dy,dx) - theta)); fp32 is enough for a term that only leaves
      // the simulator (snake:301-322).  The heading is handed to observe() for reuse.
      o.rel = relative_heading<true>((float)g.dy, (float)g.dx, (float)e.th);
      o.rel_valid = true;
      rew += P.prox_w * (double)cos_wrapped(o.rel);
    }
  }
  rew += P.time_penalty;
  bool any_alive = false;
  if (!P.respawn) {
#pragma unroll
    for (int k = 0; k < FMAX; ++k) any_alive = any_alive || !is_none(e.fx[k]);
  }
  step_tail(e, P, o, rew, any_alive);
  SALP_STAMP(5);
  return o;
}

// legacy:371-388 + snake:366-428: the observation row (10 + 4K + 2 floats) in registers `o`.
// rmax = max(ellipse_a, ellipse_b) of the current state.  If have_rel, `rel0` is the relative
// heading of the nearest food already evaluated by the reward for the same food set.
template <int FMAX, int KMAX, bool STD>
__device__ __forceinline__ void observe(const Env<FMAX>& e, const DevParams& P, double rmax, bool have_rel, float rel0,
                                        float (&o)[12 + 4 * KMAX]) {
  SALP_CONSTS;
  const int K = (KMAX == 3) ? 3 : P.K;
  // normalisations in fp32 on the rounded fp64 state (<= 1.5 ulp of the reference's f32 value)
  o[0] = (float)e.x * (float)CV(inv_W);
  o[1] = (float)e.y * (float)CV(inv_H);
  o[2] = (float)e.vx * 0.2f;
  o[3] = (float)e.vy * 0.2f;
  o[4] = (float)e.th * (float)CV(inv_pi);
  o[5] = (float)e.om * 10.0f;
  o[6] = (float)rmax * (float)CV(inv_R);
  o[7] = (float)bw_phase(e.packed) * 0.5f;
  o[8] = (float)e.water;
  o[9] = (float)e.noz * (float)CV(inv_max_nozzle);
  float dsum = 0.f;
  int cnt = 0;
  if constexpr (FMAX == 1) {
  // squared distances of live foods in fp64 (the sort key), distances in fp32 (the outputs)
  double d2[FMAX];
  float d[FMAX];
  uint32_t live = 0;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    const double dx = e.fx[k] - e.x, dy = e.fy[k] - e.y;
    d2[k] = dx * dx + dy * dy;
    const bool ok = !is_none(e.fx[k]);
    d[k] = __builtin_amdgcn_sqrtf((float)d2[k]);
    if (ok) { live |= (1u << k); ++cnt; }
  }
  const float th = (float)e.th;
  // K nearest, nearest first; ties keep slot order (stable sort, snake:382)
  uint32_t left = live;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    float v0 = 0.f, v1 = 0.f, v2 = 1.f, v3 = 0.f;   // padding for an empty slot (snake:412)
    if (s < K) {
      double bd2 = 0.0;
      float bd = 0.f, bx = 0.f, by = 0.f;
      int bi = -1;
#pragma unroll
      for (int k = 0; k < FMAX; ++k) {
        const bool cand = ((left >> k) & 1u) && (bi < 0 || d2[k] < bd2);
        if (cand) { bi = k; bd2 = d2[k]; bd = d[k]; bx = (float)(e.fx[k] - e.x); by = (float)(e.fy[k] - e.y); }
      }
      if (bi >= 0) {
        left &= ~(1u << bi);
        dsum += bd;
        float rel;
        if (s == 0) {
          if (__all(have_rel)) rel = rel0;                    // wave-uniform: skips the second atan2
          else rel = have_rel ? rel0 : relative_heading(by, bx, th);
        } else {
          rel = relative_heading(by, bx, th);
        }
        v0 = bx * (float)CV(inv_W);
        v1 = by * (float)CV(inv_H);
        v2 = bd * CV(inv_diag);
        v3 = rel * 0.318309886183791f;
      }
    }
    o[10 + 4 * s + 0] = v0; o[10 + 4 * s + 1] = v1; o[10 + 4 * s + 2] = v2; o[10 + 4 * s + 3] = v3;
  }
  // the mean distance runs over ALL live foods (snake:418-420), not only the K observed
#pragma unroll
  for (int k = 0; k < FMAX; ++k) if ((left >> k) & 1u) dsum += d[k];
  } else {
  // (reset / observe kernels only: the rollout kernels select through salp_food_reg.h / salp_food_lds.h.)  The sort key
  // is the reference's: the fp64 distance sqrt(dx^2 + dy^2) (snake:378-382; squared distances an ulp apart can share a
  // distance, and then slot order decides); offsets and distances of the outputs in fp32.  Written with selects.
  double d2[FMAX];
  float d[FMAX], dxf[FMAX], dyf[FMAX];
  uint32_t live = 0;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    const double dx = e.fx[k] - e.x, dy = e.fy[k] - e.y;
    const double sq = dx * dx + dy * dy;
    d2[k] = __builtin_sqrt(sq);
    const bool ok = !is_none(e.fx[k]);
    dxf[k] = (float)dx; dyf[k] = (float)dy;
    d[k] = __builtin_amdgcn_sqrtf((float)sq);
    live |= ok ? (1u << k) : 0u;
    cnt += ok ? 1 : 0;
  }
  const float th = (float)e.th;
  // K nearest, nearest first; ties keep slot order (stable sort, snake:382)
  uint32_t left = live;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    float v0 = 0.f, v1 = 0.f, v2 = 1.f, v3 = 0.f;   // padding for an empty slot (snake:412)
    if (s < K) {
      double bd2 = 0.0;
      float bd = 0.f, bx = 0.f, by = 0.f;
      int bi = -1;
#pragma unroll
      for (int k = 0; k < FMAX; ++k) {
        const bool cand = ((left >> k) & 1u) && (bi < 0 || d2[k] < bd2);
        bi = cand ? k : bi;
        bd2 = cand ? d2[k] : bd2;
        bd = cand ? d[k] : bd;
        bx = cand ? dxf[k] : bx;
        by = cand ? dyf[k] : by;
      }
      const bool found = bi >= 0;
      left &= found ? ~(1u << (bi & 31)) : 0xFFFFFFFFu;
      dsum += found ? bd : 0.f;
      float rel;
      if (s == 0 && __all(have_rel)) rel = rel0;              // wave-uniform: skips the second atan2
      else {
        rel = relative_heading(by, bx, th);
        if (s == 0) rel = have_rel ? rel0 : rel;
      }
      v0 = found ? bx * (float)CV(inv_W) : 0.f;
      v1 = found ? by * (float)CV(inv_H) : 0.f;
      v2 = found ? bd * CV(inv_diag) : 1.f;
      v3 = found ? rel * 0.318309886183791f : 0.f;
    }
    o[10 + 4 * s + 0] = v0; o[10 + 4 * s + 1] = v1; o[10 + 4 * s + 2] = v2; o[10 + 4 * s + 3] = v3;
  }
  // the mean distance runs over ALL live foods (snake:418-420), not only the K observed
#pragma unroll
  for (int k = 0; k < FMAX; ++k) dsum += ((left >> k) & 1u) ? d[k] : 0.f;
  }
  const float fcnt = (float)cnt;
  const float s0 = fminf(fcnt * 0.1f, 1.0f);
  const float s1 = (cnt > 0) ? (dsum * __builtin_amdgcn_rcpf(fcnt)) * CV(inv_diag) : 1.0f;
  // the two summary values sit right after the K-th food block
  if (KMAX == 3) {
    o[22] = s0; o[23] = s1;
  } else {
#pragma unroll
    for (int s = 0; s <= KMAX; ++s) if (s == K) { o[10 + 4 * s] = s0; o[11 + 4 * s] = s1; }
  }
}

}  // namespace salp
