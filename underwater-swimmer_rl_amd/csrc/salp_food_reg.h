// salp_food_reg.h — the multi-food side of the rollout kernel for up to 12 food slots (the presets: 5 in
// defaults.yaml, 12 in sac_gail.yaml), with the food positions of an env in VGPRs.
//
// The LDS-resident form (salp_food_lds.h, still used above 12 slots and by the generic instantiation) costs
// 16 B x slots x 64 lanes of LDS per wavefront for the whole launch: with 12 slots 12 KB next to the 6 KB
// observation tile, which caps the CU at 8 wavefronts (2 per SIMD), and the kernel is issue-bound with the VALU
// ~58 % busy at that residency.  Since the step loop no longer keeps ~45 VGPRs of hoisted constants (build.py: no
// machine LICM) the 48 VGPRs of 12 positions fit under the 168-VGPR budget of 3 wavefronts per SIMD.  What
// registers cannot do is per-lane dynamic indexing — the K selected foods are only known as slot numbers after
// the pass — so the pass leaves each slot's offset (dx, dy) in a per-wavefront LDS block that lives only from the
// pass to the selection and shares its bytes with the observation tile (salp_vec.hip), written once per slot and
// step, read K times.  The offsets are stored as the fp64 pair the pass has in registers (one ds_write_b128,
// 12 KB per wavefront = 48 KB per workgroup, three workgroups per CU) and narrowed to fp32 for the K selected
// slots only: storing fp32 pairs took two v_cvt_f32_f64 per slot and step, 24 of the kernel's ~680 VALU
// instructions per wavefront-step, for values of which 3 are read (profiles/r02/ab_notes.md session 15).
// The arithmetic on each food is the reference's, in its order, exactly as in salp_food_lds.h.
#pragma once
#include "salp_food_lds.h"

namespace salp {

struct OffsetLds {
  double2* col;   // this lane's column of the wavefront's [FMAX][64] block of (dx, dy)
};

// One pass over the slots (see scan_foods in salp_food_lds.h for CAPTURE / COUNT).  Groups of four slots
// beyond F are skipped (wave-uniform); slots F..FMAX-1 inside a processed group are empty (NaN).
// ALLLIVE: every slot 0..FMAX-1 of every lane holds a food (the common state with respawn: a captured food is
// replaced in the same step) — no NaN can occur, so the two NaN guards of a slot (max(., 0) of the distance,
// min(., dead) of the key) are dropped: 17 instead of 19 VALU per slot.
template <int FMAX, int KMAX, bool CAPTURE, bool COUNT, bool ALLLIVE = false>
__device__ __forceinline__ void scan_foods_reg(const Env<FMAX>& e, const OffsetLds& sc, int F, double cr2, FoodScan<KMAX>& q,
                                               bool& collected, int& hit_k, int& cnt) {
  const double dead = dead_key();
#pragma unroll
  for (int s = 0; s < KMAX; ++s) q.key[s] = dead;
  float dsum = 0.f;
  int n = 0;
  collected = false;
  hit_k = 0;
#pragma unroll
  for (int k0 = 0; k0 < FMAX; k0 += 4) {
    if (k0 == 0 || k0 < F) {   // the first group always runs (its slots beyond F are empty): it fills the list
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k < FMAX) {
          const double dx = e.fx[k] - e.x, dy = e.fy[k] - e.y;
          // CAPTURE (the pass that decides a capture, snake:204-217): the reference's two products and their sum.
          // Otherwise the value only orders the foods (to 16 ulp, see the packed keys) and feeds fp32 outputs:
          // one product folded into an fma (<= 1 ulp from the reference's value, one VALU instruction fewer).
          double d2 = CAPTURE ? (dx * dx + dy * dy) : fma(dy, dy, dx * dx);   // NaN for an empty slot
          if (CAPTURE) {
            const bool hit = !collected && (d2 < cr2);   // NaN never hits
            collected = collected || hit;
            hit_k = hit ? k : hit_k;
            d2 = hit ? __builtin_nan("") : d2;
          }
          if (COUNT) n += (d2 == d2) ? 1 : 0;
          if (ALLLIVE) dsum += __builtin_amdgcn_sqrtf((float)d2);
          else dsum += __builtin_fmaxf(__builtin_amdgcn_sqrtf((float)d2), 0.f);   // maxnum: NaN (empty) adds 0
          sc.col[k * kFoodLanes] = make_double2(dx, dy);
          double cv = pack_key(ALLLIVE ? d2 : min_key_s(d2, dead), k);
          if (k < KMAX) {
            // slots 0..KMAX-1 fill the list: entries k.. are still empty (= larger than any key), so slot k is
            // inserted among k entries — 0, 2, 4 min/max instead of 5 each for K = 3
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
              if (s < k) {
                const double lo = min_key(cv, q.key[s]);
                cv = max_key(cv, q.key[s]);
                q.key[s] = lo;
              }
            }
            q.key[k < KMAX ? k : 0] = cv;
          } else {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
              const double lo = min_key(cv, q.key[s]);
              if (s + 1 < KMAX) cv = max_key(cv, q.key[s]);
              q.key[s] = lo;
            }
          }
        }
      }
    }
  }
  q.dsum = dsum;
  if (COUNT) cnt = n;
}

// fp32 geometry of the first K selected foods: the offsets the pass left in LDS, narrowed here (the reference's
// float32 cast of the fp64 difference), the distance from the key (the squared distance to within 16 ulp of fp64:
// the same float except on ~3e-8 of the values, then 1 ulp).
template <int KMAX>
__device__ __forceinline__ void resolve_reg(const OffsetLds& sc, int K, FoodScan<KMAX>& q) {
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    q.bx[s] = 0.f; q.by[s] = 0.f; q.bd[s] = 0.f; q.idx[s] = -1;
    if (s < K) {
      const int k = key_slot(q.key[s]);
      const bool found = key_found(q.key[s]);
      const double2 o = sc.col[k * kFoodLanes];
      q.idx[s] = found ? k : -1;
      q.bx[s] = found ? (float)o.x : 0.f;
      q.by[s] = found ? (float)o.y : 0.f;
      q.bd[s] = found ? __builtin_amdgcn_sqrtf((float)q.key[s]) : 0.f;
    }
  }
}

// slot `k` (per-lane) := empty, for the lanes with `doit`
template <int FMAX>
__device__ __forceinline__ void clear_slot(Env<FMAX>& e, bool doit, int k) {
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    const bool m = doit && (k == j);
    e.fx[j] = m ? __builtin_nan("") : e.fx[j];
    e.fy[j] = m ? __builtin_nan("") : e.fy[j];
  }
}

// One reference step of a multi-food env (the register counterpart of step_env_lds).
template <int FMAX, int KMAX, bool FORCED, bool STD>
__device__ __forceinline__ StepOut step_env_reg(Env<FMAX>& e, const OffsetLds& sc, const DevParams& P, uint64_t genv,
                                                float a0, float a1, int K, FoodScan<KMAX>& q, int& nlive,
                                                const DevParams* cold SALP_STAMP_PARAM) {
  // `cold`: the device-memory copy of the launch constants (ColdBlock).  The capture bonus and the collision
  // penalty are read from it inside the wave-uniform branches that need them (a few percent of the steps), so
  // their four fp64 constants and two predicate masks do not sit in — and get spilled from — scalar registers.
  const double r = step_head<FORCED, STD>(e, P, genv, a0, a1 SALP_STAMP_PASS);
  StepOut o;
  o.rmax = r;
  const double cr = r + CV(food_radius);
  const double cr2 = cr * cr;
  int hit_k, cnt_;
  bool all_live = (KMAX <= FMAX) && __all(nlive == FMAX);     // wave-uniform; then every lane also has K foods to show
  if (all_live) scan_foods_reg<FMAX, KMAX, false, false, true>(e, sc, P.F, 0.0, q, o.collected, hit_k, cnt_);
  else scan_foods_reg<FMAX, KMAX, false, false>(e, sc, P.F, 0.0, q, o.collected, hit_k, cnt_);
  SALP_STAMP(4);
  double rew = 0.0;                                              // snake:278-327, terms added in the reference's order
  if (__any(q.key[0] < cr2 * 1.00000000001)) {   // see step_env_lds
    scan_foods_reg<FMAX, KMAX, true, true>(e, sc, P.F, cr2, q, o.collected, hit_k, nlive);
    clear_slot<FMAX>(e, o.collected, hit_k);
    all_live = false;
    const DevParams& C = *cold;
    double bonus = C.food_reward;
    if (C.efficiency_bonus > 0) bonus += C.efficiency_bonus * (double)(C.max_steps_wo_food - e.ssf);
    rew = o.collected ? bonus : 0.0;
  }
  // (a second, select-free copy of the selection for the all-live case costs more registers than it saves
  // instructions: 168 VGPRs + 21 spilled against 145)
  resolve_reg<KMAX>(sc, K > 0 ? K : 1, q);   // the reward needs the nearest even when K = 0
  {
    const double mg = CV(margin);
    o.collision = (e.x - r <= mg) || (e.x + r >= CV(wall_hi_x)) || (e.y - r <= mg) || (e.y + r >= CV(wall_hi_y));
  }
  if (__any(o.collision)) {
    const double pen = cold->collision_penalty;
    rew = o.collision ? rew + pen : rew;
  }
  o.rel = relative_heading(q.by[0], q.bx[0], (float)e.th);
  o.rel_valid = q.idx[0] >= 0;
  if (P.prox_w > 0) {
    const double al = P.prox_w * (double)cos_wrapped(o.rel);
    rew += o.rel_valid ? al : 0.0;
  }
  rew += P.time_penalty;
  step_tail(e, P, o, rew, nlive > 0);
  SALP_STAMP(5);
  return o;
}

// Wavefront-cooperative placement (place_food_coop of salp_food_lds.h) on register-resident foods: the foods
// of the env being served (lane L) are broadcast with v_readlane (slot index static, L wave-uniform), an
// accepted point is written into lane L's slot through a wave-uniform switch.  Draws, order and stream
// consumption are exactly those of the serial sampler (place_food in salp_device.h).
// `scratch`: FMAX double2 of LDS private to the wavefront (the kernel lends the tile bytes, idle at this point of the
// step).  An accepted point is wave-uniform and belongs in ONE lane's slot `slot` (wave-uniform as well): written as
// `for k: if (slot == k) if (lane == L) fx[k] = ax` the compiler predicates all FMAX bodies — 97 instructions per
// accepted food at 12 slots, ~1200 per reset — so the points of a batch go to the scratch (one LDS store each) and
// lane L takes them into its registers once per batch (~8 instructions per filled slot).
template <int FMAX, bool STD>
__device__ __forceinline__ void place_food_coop_reg(Env<FMAX>& e, int lane, const DevParams& P, uint64_t genv, int todo, int limit,
                                                    double2* scratch) {
  unsigned long long need = __ballot(todo > 0);
  const double min2 = CV(min_food_dist2);
  // empty slots of the own env as a bit mask (bit k: slot k < F is empty)
  uint32_t my_empty = 0u;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) my_empty |= (k < P.F && is_none(e.fx[k])) ? (1u << k) : 0u;
  while (need) {                                   // wave-uniform: one env at a time
    const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)need) - 1);
    need &= need - 1;
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)genv, L);
    const uint32_t g_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(genv >> 32), L);
    const uint32_t rng0 = (uint32_t)__builtin_amdgcn_readlane((int)e.rng, L);
    const double rx = bcast_lane(e.x, L), ry = bcast_lane(e.y, L);
    int todo_l = __builtin_amdgcn_readlane(todo, L);
    const int limit_l = __builtin_amdgcn_readlane(limit, L);
    uint32_t empty = (uint32_t)__builtin_amdgcn_readlane((int)my_empty, L);
    uint32_t consumed = 0;                         // draws of env L's stream used so far
    int attempts = 0;
    while (todo_l > 0) {
      // 64 candidates: draw number consumed + lane
      const U4 w = philox4x32_10(g_lo, g_hi, rng0 + consumed + (uint32_t)lane, 0u, P.seed_lo, P.seed_hi);
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
          const double fxk = bcast_lane(e.fx[k], L), fyk = bcast_lane(e.fy[k], L);
          const double dx = x - fxk, dy = y - fyk;
          ok = ok && !(dx * dx + dy * dy < min2);  // NaN (empty) slots never reject
        }
      }
      int j = 0;                                   // first candidate of this batch not yet judged
      uint32_t filled = 0u;                        // slots of env L that received a point in this batch
      while (todo_l > 0 && j < kFoodLanes) {
        const unsigned long long okm = __ballot(ok) & (~0ull << j);
        const int first_ok = okm ? (__ffsll((long long)okm) - 1) : kFoodLanes;
        const int forced = j + (limit_l - attempts);          // accepted whatever it is
        const int a = __builtin_amdgcn_readfirstlane(first_ok < forced ? first_ok : forced);
        if (a >= kFoodLanes) { attempts += kFoodLanes - j; j = kFoodLanes; break; }
        const double ax = bcast_lane(x, a), ay = bcast_lane(y, a);
        if (empty) {                               // first empty slot (snake:120, 261-264)
          const int slot = __builtin_amdgcn_readfirstlane(__ffs((int)empty) - 1);
          empty &= empty - 1;
          if (lane == 0) scratch[slot] = make_double2(ax, ay);
          filled |= 1u << slot;
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
      if (filled) {                                // wave-uniform: lane L takes this batch's points (a later batch tests against them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < FMAX; ++k) {
          if ((filled >> k) & 1u) {                // wave-uniform
            const double2 v = scratch[k];          // same address in every lane: LDS broadcast
            if (lane == L) { e.fx[k] = v.x; e.fy[k] = v.y; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (lane == L) e.rng = rng0 + consumed;
  }
}

}  // namespace salp
