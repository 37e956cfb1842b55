// salp_food_lds.h — the multi-food (num_food_items > 1) side of the rollout kernel, with the food
// positions of an env in LDS instead of registers.
//
// With 12 foods the register form (Env<12>: 48 VGPRs of positions, three unrolled 12-way distance passes
// per step, a 3 x 12 select network for the K nearest) ran at 256 VGPRs with ~30 VGPR and ~150 SGPR spills
// and ~1400 instructions per wavefront-step.  Here an env keeps only its EnvCore in registers; slot k of
// lane l lives at col[k * 64] (16 B per slot, lane-contiguous: conflict-free ds_read_b128 / ds_write_b128
// for any per-lane k), and ONE rolled pass per step over the slots serves, from one fp64 squared distance
// per food, the capture test (snake:204-217), the live count and distance sum (snake:414-420) and the
// K nearest (snake:382, stable by slot) through a sorted insertion on the exact fp64 keys.
// The arithmetic on each food is the reference's, in its order: nothing here is approximate.
#pragma once
#include "salp_device.h"

namespace salp {

constexpr int kFoodLanes = 64;

struct FoodLds {
  double2* col;   // this lane's column of the wavefront's [FMAX][64] block
  __device__ __forceinline__ void get(int k, double& x, double& y) const {
    const double2 v = col[k * kFoodLanes];
    x = v.x; y = v.y;
  }
  __device__ __forceinline__ void set(int k, double x, double y) const { col[k * kFoodLanes] = make_double2(x, y); }
  __device__ __forceinline__ void clear(int k) const { set(k, __builtin_nan(""), __builtin_nan("")); }
};

// Result of one pass over the slots.
//
// Sort keys (this file: 13..16 slots with K != 3, the <16, 8> generic instantiation; everything else: fp32 keys,
// salp_food_reg.h).  The K nearest are kept by a
// sorted insertion on PACKED keys: the fp64 squared distance with the low 4 bits of its mantissa replaced by the slot
// number (slots are 0..15).  All keys of an env are then distinct, so the compare-exchange of a chain stage is just
// v_min_f64 / v_max_f64 (no index selects) and a smaller key is a nearer food.  The reference orders by sqrt(d2) with a
// stable sort (snake:382): squared distances a few ulp apart can collapse to one distance (then slot order), while
// others inside the 16 ulp of the packing stay distinct (then distance order).  The pass therefore also keeps the
// (K + 1)-th smallest key, and when two consecutive ones of any lane are closer than 2^-44 relative the wavefront
// runs exact_order_lds() — the reference's own key — so the order is the reference's in every case
// (tests/golden/ref_tie_order_f16_k5.npz).  An empty slot's key is kDeadKey | slot (finite, above any distance).
// Everything that leaves the selection (offsets, distance) is recomputed from the slot's exact position.
template <int KMAX>
struct FoodScan {
  double key[KMAX];  // packed keys of the K nearest live foods, ascending               (LDS-food kernels)
  double next;       // the (KMAX + 1)-th smallest packed key                             (LDS-food kernels)
  uint32_t top[KMAX + 1];   // fp32 keys of the K + 1 nearest, ascending                  (register-food kernels)
  bool tie;          // this lane's key order is inside its error bound: the exact order decides
  int idx[KMAX];     // slots of the K nearest, nearest first (-1: none)
  float bx[KMAX], by[KMAX], bd[KMAX];   // offsets and distance of those foods in fp32  (filled by resolve())
  float dsum;        // sum of distances over ALL live foods
};
__device__ __forceinline__ double dead_key() { return __builtin_bit_cast(double, 0x7FEFFFFFFFFFFFF0ull); }
__device__ __forceinline__ double pack_key(double d2, int k) {
  uint2 u = __builtin_bit_cast(uint2, d2);
  u.x = (u.x & ~15u) | (uint32_t)k;                                 // one v_and_or_b32 on the low dword
  return __builtin_bit_cast(double, u);
}
// v_min_f64 / v_max_f64 written as instructions: llvm.minnum on a bit-cast value gets a canonicalising
// v_max_f64 v, v, v in front of it.  Packed keys are never NaN; for min_key(d2, dead) the kernel runs in IEEE
// mode, where the minimum of a quiet NaN (an empty slot) and a number is the number.
__device__ __forceinline__ double min_key(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double min_key_s(double a, double b_uniform) {   // b in an SGPR pair
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
  return r;
}
__device__ __forceinline__ double max_key(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int key_slot(double key) { return (int)(__builtin_bit_cast(unsigned long long, key) & 15ull); }
__device__ __forceinline__ bool key_found(double key) { return key < 1.0e300; }

// One pass over the slots around (x, y): the K smallest packed keys and the distance sum.  The pass runs over
// ceil(F / 4) * 4 slots in fully unrolled groups of four (the kernel keeps slots F..FMAX-1 empty, FMAX being a
// multiple of four): an empty slot costs what a live one does and changes nothing.
//   CAPTURE = false (the common step): nothing else.  The caller decides from key[0] whether any food can be
//     inside the capture radius at all and only then runs the CAPTURE pass.
//   CAPTURE = true: also the reference's capture test with radius^2 = cr2 — the first slot inside wins, is
//     reported in hit_k and is treated as already gone by everything else in the pass (the reward's nearest
//     food and the observation are taken after it is cleared, snake:171-189, 301).
//   COUNT: also the number of live foods in `cnt` (carried in a register between food-set changes).
template <int KMAX, bool CAPTURE, bool COUNT>
__device__ __forceinline__ void scan_foods(const FoodLds& f, int F, double x, double y, double cr2, FoodScan<KMAX>& q,
                                           bool& collected, int& hit_k, int& cnt) {
  const double dead = dead_key();
#pragma unroll
  for (int s = 0; s < KMAX; ++s) q.key[s] = dead;
  double next = dead;
  float dsum = 0.f;
  int n = 0;
  collected = false;
  hit_k = 0;
#pragma unroll 1
  for (int k0 = 0; k0 < F; k0 += 4) {
    double fx[4], fy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f.get(k0 + j, fx[j], fy[j]);   // the four LDS reads in flight together
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      const double dx = fx[j] - x, dy = fy[j] - y;
      double d2 = dx * dx + dy * dy;                 // NaN for an empty slot
      if (CAPTURE) {
        const bool hit = !collected && (d2 < cr2);   // NaN never hits
        collected = collected || hit;
        hit_k = hit ? k : hit_k;
        d2 = hit ? __builtin_nan("") : d2;
      }
      if (COUNT) n += (d2 == d2) ? 1 : 0;
      dsum += __builtin_fmaxf(__builtin_amdgcn_sqrtf((float)d2), 0.f);   // maxnum: NaN (empty) adds 0
      // sorted insertion as a chain of compare-exchanges on distinct keys; v_min_f64 turns NaN into the dead key
      double cv = pack_key(min_key_s(d2, dead), k);
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const double lo = min_key(cv, q.key[s]);
        cv = max_key(cv, q.key[s]);
        q.key[s] = lo;
      }
      next = min_key(next, cv);      // what leaves the list: the smallest of those is the (KMAX + 1)-th key
    }
  }
  q.dsum = dsum;
  q.next = next;
  if (COUNT) cnt = n;
  // near ties among the KMAX + 1 smallest (see FoodScan); a pair whose larger key is an empty slot's is no tie
  bool tie = false;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    const double hi = (s + 1 < KMAX) ? q.key[s + 1] : next;
    tie = tie || (key_found(hi) && (hi - q.key[s] < hi * 5.6843418860808015e-14));   // 2^-44
    q.idx[s] = key_found(q.key[s]) ? key_slot(q.key[s]) : -1;
  }
  q.tie = tie;
}

// The reference's order, exactly (see exact_order_reg in salp_food_reg.h): distance = sqrt(dx^2 + dy^2) in fp64, K times
// the first minimum among the foods not yet taken.  Rolled loops over the LDS slots: rare path, small code.
template <int KMAX>
__device__ __forceinline__ void exact_order_lds(const FoodLds& f, int F, int K, double x, double y, FoodScan<KMAX>& q) {
  uint32_t taken = 0u;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    int bk = -1;
    if (s < K) {
      double best = __builtin_inf();
#pragma unroll 1
      for (int k = 0; k < F; ++k) {
        double fx, fy;
        f.get(k, fx, fy);
        const double dx = fx - x, dy = fy - y;
        const double d = __builtin_sqrt(dx * dx + dy * dy);          // NaN for an empty slot
        const bool take = !((taken >> k) & 1u) && (d < best);        // NaN never; ties keep the lower slot
        best = take ? d : best;
        bk = take ? k : bk;
      }
      taken |= (bk >= 0) ? (1u << bk) : 0u;
    }
    q.idx[s] = bk;
  }
}

// fp32 geometry of the first K selected foods (what the reward and the observation consume), from the exact
// positions of the selected slots.
template <int KMAX>
__device__ __forceinline__ void resolve(const FoodLds& f, int K, double x, double y, FoodScan<KMAX>& q) {
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    q.bx[s] = 0.f; q.by[s] = 0.f; q.bd[s] = 0.f;
    if (s >= K) q.idx[s] = -1;
    if (s < K) {
      double fx, fy;
      const bool found = q.idx[s] >= 0;
      const int k = q.idx[s] & 15;
      f.get(k, fx, fy);
      const double dx = fx - x, dy = fy - y;
      q.bx[s] = found ? (float)dx : 0.f;
      q.by[s] = found ? (float)dy : 0.f;
      q.bd[s] = found ? __builtin_amdgcn_sqrtf((float)(dx * dx + dy * dy)) : 0.f;
    }
  }
}

// One reference step of a multi-food env (the LDS counterpart of step_env<FMAX>): same order of
// operations, the food loop being one scan.  Leaves the selection of the post-step food set in q and keeps
// `nlive` (live foods of this env) current.
template <int KMAX, bool FORCED, bool STD>
__device__ __forceinline__ StepOut step_env_lds(EnvCore& e, const FoodLds& f, const DevParams& P, uint64_t genv, float a0, float a1,
                                                int K, FoodScan<KMAX>& q, int& nlive) {
  SALP_CONSTS;
  const double r = step_head<FORCED, STD>(e, P, genv, a0, a1);
  StepOut o;
  o.rmax = r;
  const double cr = r + CV(food_radius);
  const double cr2 = cr * cr;
  int hit_k, cnt_;
  scan_foods<KMAX, false, false>(f, P.F, e.x, e.y, 0.0, q, o.collected, hit_k, cnt_);
  // A capture needs a live food with d2 < cr2, and then the smallest key is below cr2 too (the packing moves a
  // key by < 16 ulp: the margin).  Only then — a few percent of the wavefront-steps — run the exact test.
  if (__any(q.key[0] < cr2 * 1.00000000001)) {
    scan_foods<KMAX, true, true>(f, P.F, e.x, e.y, cr2, q, o.collected, hit_k, nlive);
    if (o.collected) f.clear(hit_k);
  }
  if (__any(q.tie)) exact_order_lds<KMAX>(f, P.F, K > 0 ? K : 1, e.x, e.y, q);
  resolve<KMAX>(f, K > 0 ? K : 1, e.x, e.y, q);   // the reward needs the nearest even when K = 0
  o.collision = (e.x - r <= CV(margin)) || (e.x + r >= CV(wall_hi_x)) || (e.y - r <= CV(margin)) || (e.y + r >= CV(wall_hi_y));
  double rew = 0.0;
  if (o.collected) {
    rew += P.food_reward;
    if (P.efficiency_bonus > 0) rew += P.efficiency_bonus * (double)(P.max_steps_wo_food - e.ssf);
  }
  if (o.collision) rew += P.collision_penalty;
  o.rel = relative_heading<true>(q.by[0], q.bx[0], (float)e.th);
  o.rel_valid = q.idx[0] >= 0;
  if (P.prox_w > 0) {
    const double al = P.prox_w * (double)cos_wrapped(o.rel);
    rew += o.rel_valid ? al : 0.0;
  }
  rew += P.time_penalty;
  step_tail(e, P, o, rew, nlive > 0);
  return o;
}

// The observation row from a scan of the CURRENT food set around the CURRENT pose.
// ALLFOUND: every lane shows K live foods (wave-uniform fact established by the caller): no padding selects.
template <int KMAX, bool STD, bool ALLFOUND = false>
__device__ __forceinline__ void observe_lds(const EnvCore& e, const DevParams& P, double rmax, int K, const FoodScan<KMAX>& q,
                                            int nlive, bool have_rel, float rel0, float (&o)[12 + 4 * KMAX]) {
  SALP_CONSTS;
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
  const float th = (float)e.th;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    float v0 = 0.f, v1 = 0.f, v2 = 1.f, v3 = 0.f;   // padding for an empty slot (snake:412)
    if (s < K) {
      const bool found = ALLFOUND ? true : (q.idx[s] >= 0);
      float rel;
      if (s == 0 && __all(have_rel)) rel = rel0;    // wave-uniform: skips the second atan2
      else {
        rel = relative_heading(q.by[s], q.bx[s], th);
        if (s == 0) rel = have_rel ? rel0 : rel;
      }
      v0 = found ? q.bx[s] * (float)CV(inv_W) : 0.f;
      v1 = found ? q.by[s] * (float)CV(inv_H) : 0.f;
      v2 = found ? q.bd[s] * CV(inv_diag) : 1.f;
      v3 = found ? rel * 0.318309886183791f : 0.f;
    }
    o[10 + 4 * s + 0] = v0; o[10 + 4 * s + 1] = v1; o[10 + 4 * s + 2] = v2; o[10 + 4 * s + 3] = v3;
  }
  const float fcnt = (float)nlive;
  const float s0 = fminf(fcnt * 0.1f, 1.0f);
  const float s1 = (ALLFOUND || nlive > 0) ? (q.dsum * __builtin_amdgcn_rcpf(fcnt)) * CV(inv_diag) : 1.0f;
  if (KMAX == 3) {
    o[22] = s0; o[23] = s1;
  } else {
#pragma unroll
    for (int s = 0; s <= KMAX; ++s) if (s == K) { o[10 + 4 * s] = s0; o[11 + 4 * s] = s1; }
  }
}

// ---- wavefront-cooperative placement ---------------------------------------------------------------------
// The serial rejection sampler (place_food in salp_device.h) run by 64 lanes for the one or two envs of a wavefront
// that need food costs the whole wavefront a serial rejection loop: an autoreset with 12 foods is ~18 draws, each a Philox block plus 12
// distance tests, ~5000 instructions with 63 lanes idle, and with ~1/800 resets per env-step some lane of
// a wavefront resets on ~8 % of the steps (measured: 37 % of the sac_gail kernel time).
// Here the 64 lanes work for one env at a time: lane l draws candidate number (consumed + l) of THAT env's
// stream, all candidates are tested against the robot and the env's live foods at once (broadcast LDS
// reads), and the sequential acceptance rule of the reference — first valid draw, or the draw after
// `limit` consecutive rejections — becomes a ballot / find-first per accepted food, with the later
// candidates re-tested against each newly accepted one.  The draws taken, their order and the number
// consumed from the env's stream are exactly those of place_food.
// value of `v` in lane `src` (src wave-uniform): two v_readlane instead of two LDS permutes
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int FMAX, bool STD>
__device__ __forceinline__ void place_food_coop(EnvCore& e, double2* wave_block, int lane, const DevParams& P, uint64_t genv,
                                                int todo, int limit) {
  SALP_CONSTS;
  unsigned long long need = __ballot(todo > 0);
  const double min2 = CV(min_food_dist2);
  while (need) {                                   // wave-uniform: one env at a time
    const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)need) - 1);
    need &= need - 1;
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)genv, L);
    const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(genv >> 32), L);
    const uint32_t rng0 = (uint32_t)__builtin_amdgcn_readlane((int)e.rng, L);
    const double rx = bcast_lane(e.x, L), ry = bcast_lane(e.y, L);
    int todo_l = __builtin_amdgcn_readlane(todo, L);
    const int limit_l = __builtin_amdgcn_readlane(limit, L);
    const double2* colL = wave_block + L;          // slot k of env L at colL[k * 64]
    // the env's foods, once: lane k holds slot k (one parallel read); empty slots as a scalar bit mask
    const double2 mine = colL[(lane < P.F ? lane : 0) * kFoodLanes];
    unsigned long long empty = __ballot(lane < P.F && is_none(mine.x));
    uint32_t consumed = 0;                         // draws of env L's stream used so far
    int attempts = 0;
    while (todo_l > 0) {
      // 64 candidates: draw number consumed + lane
      const U4 w = philox4x32_10(g_lo, g_hi, rng0 + consumed + (uint32_t)lane, 0u, P.seed[0], P.seed[1]);
      const double x = CV(food_xlo) + CV(food_xspan) * u53(w.x, w.y);
      const double y = CV(food_ylo) + CV(food_yspan) * u53(w.z, w.w);
      bool ok;
      {
        const double dx = x - rx, dy = y - ry;
        ok = !(dx * dx + dy * dy < min2);
      }
#pragma unroll
      for (int k = 0; k < FMAX; ++k) {
        if (k < P.F) {                             // wave-uniform
          const double2 fv = colL[k * kFoodLanes]; // same address in every lane: LDS broadcast
          const double dx = x - fv.x, dy = y - fv.y;
          ok = ok && !(dx * dx + dy * dy < min2);  // NaN (empty) slots never reject
        }
      }
      int j = 0;                                   // first candidate of this batch not yet judged
      while (todo_l > 0 && j < kFoodLanes) {
        const unsigned long long okm = __ballot(ok) & (~0ull << j);
        const int first_ok = okm ? (__ffsll((long long)okm) - 1) : kFoodLanes;
        const int forced = j + (limit_l - attempts);          // accepted whatever it is
        const int a = __builtin_amdgcn_readfirstlane(first_ok < forced ? first_ok : forced);
        if (a >= kFoodLanes) { attempts += kFoodLanes - j; j = kFoodLanes; break; }
        const double ax = bcast_lane(x, a), ay = bcast_lane(y, a);
        if (empty) {                               // first empty slot (snake:120, 261-264)
          const int slot = __ffsll((long long)empty) - 1;
          empty &= empty - 1;
          if (lane == 0) wave_block[slot * kFoodLanes + L] = make_double2(ax, ay);
        }
        {
          const double dx = x - ax, dy = y - ay;
          ok = ok && !(dx * dx + dy * dy < min2);
        }
        todo_l -= 1;
        attempts = 0;
        j = a + 1;
      }
      consumed += (uint32_t)j;
    }
    if (lane == L) e.rng = rng0 + consumed;
  }
}

}  // namespace salp
