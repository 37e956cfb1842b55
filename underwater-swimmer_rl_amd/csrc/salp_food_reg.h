// salp_food_reg.h — the multi-food side of the rollout kernel for up to 12 food slots (the presets: 5 in
// defaults.yaml, 12 in sac_gail.yaml): food positions of an env in VGPRs (fp64, authoritative) with an fp32
// mirror in LDS that everything which only LEAVES the simulator is computed from.
//
// What feeds back into the state stays exact and in the reference's operation order: the capture test
// (snake:204-217) and the placement tests (snake:92-131, 232-276) read the fp64 positions in registers.  What only
// orders foods and fills float32 outputs — the K nearest (snake:366-382), their offsets / distances / bearings
// (:386-410), the distance sum (:414-420), the reward's nearest food (:350-364) — runs in fp32 on the mirror:
//   * one pass per step over the slots: (dx, dy) = mirror - (float)(x, y), d2 = fma(dy, dy, dx dx), sqrt for the
//     distance sum, key = d2's bit pattern with the slot number in its low 4 bits.  fp32 add / mul / fma issue at 2.5
//     cycles per wavefront on gfx950 where every fp64 (and most other) VALU instruction takes 4.3
//     (profiles/micro/valu_rates2.hip): 45 against 59-63 cycles per slot for the round-2 fp64 pass, measured.
//   * K = 3 (every preset): the three smallest keys AND the fourth through a sort-by-threes / merge network on
//     v_min3 / v_med3 / v_max3_u32 — 37 instructions for 12 slots against 51 for the sorted insertion (which
//     had no fourth key).  Positive floats order as unsigned integers, a NaN (empty slot) above every number.
//   * exactness: the fp32 keys can mis-order two foods only if their squared distances are closer than the error
//     bound of the fp32 evaluation (tie_tolerance below).  When a gap between two of the four smallest keys of any
//     lane is inside that bound — ~1 % of the wavefront-steps with 12 foods — the wavefront runs exact_order_reg():
//     the reference's own key, sqrt(dx^2 + dy^2) in fp64 from the fp64 positions, first minimum first (the stable
//     sort of snake:382; strict `<` of :350-364).  The order is therefore the reference's in every case, exact ties and
//     squared distances a few ulp apart included (tests/golden/ref_tie_order_f*.npz) — round 2's packed fp64 keys
//     ordered foods within 16 ulp of each other by slot.
//   * registers cannot be indexed per lane, and the K selected foods are only known as slot numbers after the pass:
//     their positions are read back from the mirror (three ds_read_b64) — the round-2 kernel stored every slot's
//     fp64 offset pair each step (twelve ds_write_b128) for the same purpose.
// The mirror is written where the food set changes (kernel entry, capture, placement, reset): rare.
#pragma once
#include "salp_food_lds.h"

namespace salp {

struct MirrorLds {
  float2* col;   // this lane's column of the wavefront's [FMAX][64] block of (x, y)
  __device__ __forceinline__ float2 get(int k) const { return col[k * kFoodLanes]; }
  __device__ __forceinline__ void set(int k, double x, double y) const { col[k * kFoodLanes] = make_float2((float)x, (float)y); }
  __device__ __forceinline__ void clear(int k) const { col[k * kFoodLanes] = make_float2(__builtin_nanf(""), __builtin_nanf("")); }
};

// Which instantiations keep the fp32 roundings in registers as well: the K = 3 kernels with 12 and 16 slots, which LDS
// holds at three / two wavefronts per SIMD anyway (49 / 57 KB per workgroup: room for 168 / 256 VGPRs; the 12-slot kernel
// gains 2.7 % from the copies).  The 4- and 8-slot kernels read the mirror instead: they fit four wavefronts per SIMD by LDS
// (32 / 40 KB), and only without the copies by registers (<= 128 VGPRs; profiles/r03/ab_notes.md sessions 9 and 12).
constexpr bool food_in_registers(int fmax, int kmax, bool std_consts, bool full) {
  return kmax == 3 && (fmax > 12 || (fmax == 12 && std_consts && full));
}

// What the per-step pass reads (static slot index).  INREG (the K = 3 kernels): the mirror's values also in registers —
// the pass then issues no LDS read at all (with six ds_read2st64_b64 and their waits in the pass the kernel ran 8 % slower
// than round 2's, profiles/r03/ab_notes.md).  !INREG (the generic instantiation, already at its register limit): the pass
// reads the mirror.
template <int FMAX, bool INREG>
struct FoodF32 {
  float x[INREG ? FMAX : 1], y[INREG ? FMAX : 1];
  __device__ __forceinline__ void set(int k, double fx, double fy) { if (INREG) { x[k] = (float)fx; y[k] = (float)fy; } }
  __device__ __forceinline__ void clear(int k, bool doit) {
    if (INREG) { x[k] = doit ? __builtin_nanf("") : x[k]; y[k] = doit ? __builtin_nanf("") : y[k]; }
  }
  __device__ __forceinline__ float2 get(int k, const MirrorLds& m) const { return INREG ? make_float2(x[k], y[k]) : m.get(k); }
};

// ---- unsigned min / max as instructions (the three-operand forms have no builtin)
__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}

constexpr uint32_t kKeyNone = 0xFFFFFFFFu;     // above every key, NaN keys included
constexpr uint32_t kKeyInf = 0x7F800000u;      // keys below this are live foods
// key of slot k: the bit pattern of its squared distance (sign cleared: positive floats order as unsigned integers,
// +inf and NaN above every finite value) with the slot number in the low 4 bits — all keys of an env are distinct
__device__ __forceinline__ uint32_t key_of(float d2, int k) { return (__float_as_uint(d2) & 0x7FFFFFF0u) | (uint32_t)k; }

// Error bound of the fp32 keys, as a gap in squared distance below which two keys may be in the wrong order.
// Positions and pose are rounded to fp32 (half an ulp of a coordinate <= L = max(width, height): L 2^-25 each), the
// difference once more, so |delta dx| <= e = 1.8e-7 L; d2 carries 2 sqrt2 d e from that plus two fp32 roundings
// (1.2e-7 d2) plus the 4 slot bits (1.9e-6 d2): per key <= 2.83 e d + 2.1e-6 d2, per pair twice that.  With
// tol(d2) = c0 + 8e-6 d2 the pair bound 5.66 e d + 4.2e-6 d2 <= tol needs c0 >= (5.66 e)^2 / (4 * 3.8e-6) = 6.9e-8 L^2;
// c0 = 1.4e-7 L^2 (0.09 at L = 800; DevParams::tie_c0) keeps a factor two in hand.
#ifdef SALP_EXP_TIE_OFF   // experiment: the tie test and the exact order compiled in, never taken
__device__ __forceinline__ float tie_tolerance(float d2_max, float c0) { return fmaf(d2_max, 0.0f, -1.0f); }
#else
__device__ __forceinline__ float tie_tolerance(float d2_max, float c0) { return fmaf(d2_max, 8.0e-6f, c0); }
#endif

// ---- K = 3: sort-by-threes and merge, keeping the fourth smallest
struct Tri { uint32_t a, b, c; };   // ascending
__device__ __forceinline__ Tri sort3(uint32_t x, uint32_t y, uint32_t z) { return Tri{umin3(x, y, z), umed3(x, y, z), umax3(x, y, z)}; }
// the three smallest of two sorted triples, and their fourth smallest into d
__device__ __forceinline__ Tri merge33(const Tri& p, const Tri& q, uint32_t& d) {
  const uint32_t t = umax2(p.a, q.a);
  Tri r;
  r.a = umin2(p.a, q.a);
  r.b = umin3(t, p.b, q.b);
  r.c = umin3(umed3(t, p.b, q.b), p.c, q.c);
  d = umax3(t, umin2(p.b, q.c), umin2(p.c, q.b));     // third LARGEST of the six
  return r;
}
__device__ __forceinline__ Tri merge32(const Tri& p, uint32_t qa, uint32_t qb, uint32_t& d) {   // q = (qa <= qb, none)
  const uint32_t t = umax2(p.a, qa);
  Tri r;
  r.a = umin2(p.a, qa);
  r.b = umin3(t, p.b, qb);
  r.c = umin2(umed3(t, p.b, qb), p.c);
  d = umax3(t, p.b, umin2(p.c, qb));
  return r;
}
__device__ __forceinline__ Tri merge31(const Tri& p, uint32_t x, uint32_t& d) {                 // q = (x, none, none)
  const uint32_t t = umax2(p.a, x), u = umax2(p.b, t);
  Tri r;
  r.a = umin2(p.a, x);
  r.b = umin2(p.b, t);
  r.c = umin2(p.c, u);
  d = umax2(p.c, u);
  return r;
}

// Smallest KMAX + 1 keys, ascending, of key[0..FMAX-1] into top[0..KMAX].
template <int FMAX, int KMAX>
__device__ __forceinline__ void select_keys(const uint32_t (&key)[FMAX], uint32_t (&top)[KMAX + 1]) {
  if constexpr (KMAX == 3 && FMAX >= 3) {
    Tri acc = sort3(key[0], key[1], key[2]);
    uint32_t fourth = kKeyNone;
#pragma unroll
    for (int g = 3; g < FMAX; g += 3) {
      uint32_t d;
      if (g + 2 < FMAX) acc = merge33(acc, sort3(key[g], key[g + 1], key[g + 2 < FMAX ? g + 2 : 0]), d);
      else if (g + 1 < FMAX) acc = merge32(acc, umin2(key[g], key[g + 1 < FMAX ? g + 1 : 0]), umax2(key[g], key[g + 1 < FMAX ? g + 1 : 0]), d);
      else acc = merge31(acc, key[g], d);
      fourth = umin2(fourth, d);
    }
    top[0] = acc.a; top[1] = acc.b; top[2] = acc.c; top[3] = fourth;
  } else {     // generic K: sorted insertion (slot k among min(k, KMAX + 1) entries)
#pragma unroll
    for (int s = 0; s <= KMAX; ++s) top[s] = kKeyNone;
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      uint32_t cv = key[k];
#pragma unroll
      for (int s = 0; s <= KMAX; ++s) {
        if (s < k) {
          const uint32_t lo = umin2(cv, top[s]);
          if (s < KMAX) cv = umax2(cv, top[s]);
          top[s] = lo;
        }
      }
      if (k <= KMAX) top[k] = cv;
    }
  }
}

// One fp32 pass over the slots around (xf, yf): distance sum, live count (COUNT), the K nearest as slot numbers in
// q.idx (-1: none) and whether the fp32 order of this lane is inside its error bound (q.tie: then exact_order_reg decides).
// ALLLIVE: every slot of every lane holds a food (the steady state with respawn): no NaN can occur, the NaN guard of
// the distance sum and the found tests are dropped.
template <int FMAX, int KMAX, bool ALLLIVE, bool COUNT, bool INREG>
__device__ __forceinline__ void scan_foods_f32(const FoodF32<FMAX, INREG>& ff, const MirrorLds& m, int K, float xf, float yf, float tol_c0,
                                               FoodScan<KMAX>& q, int& cnt) {
  uint32_t key[FMAX];
  float dsum = 0.f;
  int n = 0;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    const float2 p = ff.get(k, m);
    const float dx = p.x - xf, dy = p.y - yf;
    const float d2 = fmaf(dy, dy, dx * dx);          // NaN for an empty slot
    if (COUNT) n += (d2 == d2) ? 1 : 0;
    const float sq = __builtin_amdgcn_sqrtf(d2);
    dsum += ALLLIVE ? sq : __builtin_fmaxf(sq, 0.f);   // maxnum: NaN (empty) adds 0
    key[k] = key_of(d2, k);
  }
  q.dsum = dsum;
  if (COUNT) cnt = n;
  select_keys<FMAX, KMAX>(key, q.top);
  // gaps between consecutive keys of the K + 1 smallest against the error bound at the largest of them
  float gap = 3.0e38f, top_d2 = 0.f;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    if (s < K) {
      const float lo = __uint_as_float(q.top[s]), hi = __uint_as_float(q.top[s + 1]);
      gap = __builtin_fminf(gap, hi - lo);                // a NaN key (none) gives a NaN gap: ignored by minnum
      top_d2 = ALLLIVE ? hi : __builtin_fmaxf(top_d2, __builtin_fmaxf(lo, hi));
    }
  }
  q.tie = gap < tie_tolerance(top_d2, tol_c0);
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    const bool found = ALLLIVE || (q.top[s] < kKeyInf);
    q.idx[s] = (s < K && found) ? (int)(q.top[s] & 15u) : -1;
  }
}

// The reference's order, exactly: the sort key is distance = sqrt(dx^2 + dy^2) in fp64 from the fp64 positions
// (snake:378-379), and the K nearest are K times the first minimum among the foods not yet taken — what a stable sort by
// distance (snake:382) and the strict `<` scan of snake:350-364 produce.  Runs for the whole wavefront when some lane's
// fp32 order is inside its error bound.
// Careful form: the distances themselves (the device's fp64 sqrt is correctly rounded), K rounds of "first minimum among
// the foods not yet taken", each round recomputing the distances from the registers (the pose is made opaque per slot so
// that the square roots are formed one after the other and do not stay live).  Only reached when two squared distances are within
// 2^-44 of each other (see below): never in a benchmark run, always in tests/golden/ref_tie_order_f*.npz.
template <int FMAX, int KMAX>
__device__ __forceinline__ void exact_order_sqrt_reg(const Env<FMAX>& e, int K, FoodScan<KMAX>& q) {
  uint32_t taken = 0u;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) q.idx[s] = -1;
#pragma unroll 1
  for (int s = 0; s < K; ++s) {
    int bk = -1;
    double best = __builtin_inf();
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      double ex = e.x, ey = e.y;
      asm volatile("" : "+v"(ex), "+v"(ey));      // one slot after the other, every round anew: short live ranges
      const double dx = e.fx[k] - ex, dy = e.fy[k] - ey;
      const double d = __builtin_sqrt(dx * dx + dy * dy);           // NaN for an empty slot
      const bool take = !((taken >> k) & 1u) && (d < best);         // NaN never; ties keep the lower slot
      best = take ? d : best;
      bk = take ? k : bk;
    }
    taken |= (bk >= 0) ? (1u << bk) : 0u;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) q.idx[j] = (j == s) ? bk : q.idx[j];
  }
}
// Usual form: ONE pass of packed fp64 keys — the reference's squared distance dx^2 + dy^2 with the slot number in its low
// 4 mantissa bits, sorted insertion on v_min_f64 / v_max_f64 (round 2's per-step pass, ~13 instructions per slot).  sqrt
// is monotonic and squared distances more than 16 + 4 ulp apart keep their order through the packing and have different
// (correctly rounded) square roots, so this IS the order of the distances unless two of the K + 1 smallest keys are closer
// than 2^-44 relative — then, and only then, the careful form decides (squared distances 1 ulp apart can share a distance:
// the tie goes to the lower slot; tests/golden/ref_tie_order_f*.npz envs 1 and 2).  It has to be cheap, not just rare:
// a swimmer rests for the first ~135 steps of an episode, so a near tie at reset recurs on every one of them, and the
// launch ends with its slowest wavefront (with three 36-slot rounds here the 12-food kernel lost 12 %, ab_notes.md).
template <int FMAX, int KMAX>
__device__ __forceinline__ void exact_order_reg(const Env<FMAX>& e, int K, FoodScan<KMAX>& q) {
  const double dead = dead_key();
  double key[KMAX + 1];
#pragma unroll
  for (int s = 0; s <= KMAX; ++s) key[s] = dead;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    double ex = e.x, ey = e.y;
    asm volatile("" : "+v"(ex), "+v"(ey));     // one slot after the other: short live ranges (this path runs at the kernel's register limit)
    const double dx = e.fx[k] - ex, dy = e.fy[k] - ey;
    double cv = pack_key(min_key_s(dx * dx + dy * dy, dead), k);     // NaN (empty slot) -> the dead key
#pragma unroll
    for (int s = 0; s <= KMAX; ++s) {
      if (s < k) {                                                     // slot k is inserted among min(k, KMAX + 1) entries
        const double lo = min_key(cv, key[s]);
        if (s < KMAX) cv = max_key(cv, key[s]);
        key[s] = lo;
      }
    }
    if (k <= KMAX) key[k] = cv;
  }
  bool close = false;
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    close = close || (s < K && key_found(key[s + 1]) && (key[s + 1] - key[s] < key[s + 1] * 5.6843418860808015e-14));   // 2^-44
    q.idx[s] = (s < K && key_found(key[s])) ? key_slot(key[s]) : -1;
  }
  if (__any(close)) exact_order_sqrt_reg<FMAX, KMAX>(e, K, q);
}

// fp32 geometry of the K selected foods from their mirror positions (the same differences the pass formed).
template <int KMAX, bool ALLFOUND>
__device__ __forceinline__ void resolve_f32(const MirrorLds& m, int K, float xf, float yf, FoodScan<KMAX>& q) {
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    q.bx[s] = 0.f; q.by[s] = 0.f; q.bd[s] = 0.f;
    if (s < K) {
      const bool found = ALLFOUND || (q.idx[s] >= 0);
      const float2 p = m.get(ALLFOUND ? q.idx[s] : (q.idx[s] & 15));
      const float bx = p.x - xf, by = p.y - yf;
      const float bd = __builtin_amdgcn_sqrtf(fmaf(by, by, bx * bx));
      q.bx[s] = found ? bx : 0.f;
      q.by[s] = found ? by : 0.f;
      q.bd[s] = found ? bd : 0.f;
    }
  }
}

// The fp32 offsets carry the roundings of both positions (up to ~6e-5 px in an 800-wide tank): a bearing error of 6e-5 / d rad.
// A live food is normally at least a capture radius away (d >= ~48 px: 1.3e-6 rad), but a fallback placement (snake:120-131,
// :270-276: the unconditional draw after the rejections are used up, crowded tanks only) can put one next to the swimmer, and
// with two foods inside the capture radius the farther slot survives a step: seen at d = 0.7 px, where the shaped reward
// 5 cos(bearing) was off by 8e-5 (tests/soak_main_kernels.py case 92).  So for observed foods nearer than 0.05 max(W, H)
// (40 px: inside every default capture radius) the offsets are taken from the exact positions; wave-uniform and rare.
template <int FMAX, int KMAX>
__device__ __forceinline__ void refine_near(const Env<FMAX>& e, int K, float tol_c0, FoodScan<KMAX>& q) {
  const float near2 = tol_c0 * 17857.f;            // tol_c0 = 1.4e-7 L^2  ->  (0.05 L)^2
  // entry 0, the nearest: the one the reward reads; a second food that close as well has not been seen
  const bool nr = (K > 0) && (q.idx[0] >= 0) && (q.bd[0] * q.bd[0] < near2);
  if (__any(nr)) {
    double dx = 0.0, dy = 0.0;
#pragma unroll
    for (int k = 0; k < FMAX; ++k) {
      const bool h = nr && (q.idx[0] == k);
      double ex = e.x, ey = e.y;
      asm volatile("" : "+v"(ex), "+v"(ey));      // one slot at a time, inside this branch (see exact_order_reg)
      dx = h ? (e.fx[k] - ex) : dx;
      dy = h ? (e.fy[k] - ey) : dy;
    }
    q.bx[0] = nr ? (float)dx : q.bx[0];
    q.by[0] = nr ? (float)dy : q.by[0];
    q.bd[0] = nr ? (float)__builtin_sqrt(dx * dx + dy * dy) : q.bd[0];
  }
}

// Selection of the current food set around the current pose: pass, exact order where needed, geometry.
template <int FMAX, int KMAX, bool ALLLIVE, bool COUNT, bool INREG>
__device__ __forceinline__ void select_foods_reg(const Env<FMAX>& e, const FoodF32<FMAX, INREG>& ff, const MirrorLds& m, int K,
                                                 float tol_c0, FoodScan<KMAX>& q, int& cnt) {
  const float xf = (float)e.x, yf = (float)e.y;
  scan_foods_f32<FMAX, KMAX, ALLLIVE, COUNT>(ff, m, K, xf, yf, tol_c0, q, cnt);
  if (__any(q.tie)) exact_order_reg<FMAX, KMAX>(e, K, q);
  resolve_f32<KMAX, ALLLIVE>(m, K, xf, yf, q);
  refine_near<FMAX, KMAX>(e, K, tol_c0, q);
}

// snake:204-217 _check_food_collection on the fp64 positions, in the reference's arithmetic: the first live food
// inside the capture radius; also the number of foods alive afterwards.
template <int FMAX>
__device__ __forceinline__ void capture_test_reg(const Env<FMAX>& e, double cr2, bool& collected, int& hit_k, int& alive) {
  collected = false;
  hit_k = 0;
  int n = 0;
#pragma unroll
  for (int k = 0; k < FMAX; ++k) {
    const double dx = e.fx[k] - e.x, dy = e.fy[k] - e.y;
    const double d2 = dx * dx + dy * dy;           // NaN for an empty slot
    const bool hit = !collected && (d2 < cr2);     // NaN never hits
    collected = collected || hit;
    hit_k = hit ? k : hit_k;
    n += (d2 == d2 && !hit) ? 1 : 0;
  }
  alive = n;
}

// slot `k` (per-lane) := empty, for the lanes with `doit`: registers (fp64 and fp32) and mirror
template <int FMAX, bool INREG>
__device__ __forceinline__ void clear_slot(Env<FMAX>& e, FoodF32<FMAX, INREG>& ff, const MirrorLds& m, bool doit, int k) {
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    const bool hit = doit && (k == j);
    e.fx[j] = hit ? __builtin_nan("") : e.fx[j];
    e.fy[j] = hit ? __builtin_nan("") : e.fy[j];
    ff.clear(j, hit);
  }
  if (doit) m.clear(k);
}

// One reference step of a multi-food env (the register counterpart of step_env_lds).
template <int FMAX, int KMAX, bool FORCED, bool STD, bool INREG>
__device__ __forceinline__ StepOut step_env_reg(Env<FMAX>& e, FoodF32<FMAX, INREG>& ff, const MirrorLds& m, const DevParams& P, uint64_t genv,
                                                float a0, float a1, int K, FoodScan<KMAX>& q, int& nlive, int& order_cache,
                                                const DevParams* cold SALP_STAMP_PARAM) {
  SALP_CONSTS;
  // `cold`: the device-memory copy of the launch constants (ColdBlock).  The capture bonus and the collision
  // penalty are read from it inside the wave-uniform branches that need them (a few percent of the steps), so
  // their four fp64 constants and two predicate masks do not sit in — and get spilled from — scalar registers.
  const double r = step_head<FORCED, STD>(e, P, genv, a0, a1 SALP_STAMP_PASS);
  StepOut o;
  o.rmax = r;
  o.collected = false;
  const double cr = r + CV(food_radius);
  const int Ksel = K > 0 ? K : 1;                // the reward needs the nearest even when K = 0
  const float xf = (float)e.x, yf = (float)e.y;
  const float tol_c0 = CV(tie_c0);
  int cnt_;
  bool all_live = (KMAX <= FMAX) && __all(nlive == FMAX);     // wave-uniform; then every lane also has K foods to show
  if (all_live) scan_foods_f32<FMAX, KMAX, true, false>(ff, m, Ksel, xf, yf, tol_c0, q, cnt_);
  else scan_foods_f32<FMAX, KMAX, false, false>(ff, m, Ksel, xf, yf, tol_c0, q, cnt_);
  SALP_STAMP(4);
  double rew = 0.0;                                              // snake:278-327, terms added in the reference's order
  // A capture needs a live food with d2 < cr^2; its fp32 key is then below cr^2 plus the key's error bound (2.83 e d +
  // 2.1e-6 d2 at d = cr <= ~60: < 0.02 + 1e-5 cr^2).  Only then — a few percent of the wavefront-steps — run the exact test.
  {
    const float crf = (float)cr;
    if (__any(__uint_as_float(q.top[0]) < fmaf(crf * crf, 1.00002f, 0.05f + tol_c0))) {
      int hit_k;
      capture_test_reg<FMAX>(e, cr * cr, o.collected, hit_k, nlive);
      if (__any(o.collected)) {
        clear_slot(e, ff, m, o.collected, hit_k);
        all_live = false;
        // The reward's nearest food is taken after the slot is cleared (snake:171-189, 301): with the captured food gone
        // the nearest is the old nearest, or the old second if the captured one WAS the nearest — no second pass over the
        // slots (240 instructions on ~7 % of the wavefront-steps).  The rest of the selection (entries 1.., the distance
        // sum, the live count) is stale for these lanes and is not read: a capture always sends the wavefront through the
        // rare region below, which selects again for the observation once the respawn has been placed.
        {
          const bool was_first = (int)(q.top[0] & 15u) == hit_k;
          const uint32_t k0 = was_first ? q.top[1] : q.top[0];
          q.idx[0] = o.collected ? ((k0 < kKeyInf) ? (int)(k0 & 15u) : -1) : q.idx[0];
        }
        const DevParams& C = *cold;
        double bonus = C.food_reward;
        if (C.efficiency_bonus > 0) bonus += C.efficiency_bonus * (double)(C.max_steps_wo_food - e.ssf);
        rew = o.collected ? bonus : 0.0;
      }
    }
  }
  SALP_COUNT(2, true);
  SALP_COUNT(3, all_live);
  if (__any(q.tie)) {
    // A swimmer that has not moved since its reset (zero velocity: legacy:95-117 until the first thrust, ~135 steps) sees the
    // same foods from the same pose on every step, so a near tie recurs on all of them with the same answer: the exact
    // order is computed once per episode and remembered (`order_cache`: K slot numbers, 4 bits each; < 0: none; the
    // kernel drops it whenever the env's food set changes or it is reset).  Without this a wavefront with one such lane
    // runs the exact pass on 135 of its steps, and a launch ends with its slowest wavefront.
    constexpr bool kOrderCache = 5 * KMAX <= 30;            // K = 3: 15 bits of one register (the generic K <= 8 recomputes)
    const bool resting = (e.vx == 0.0) && (e.vy == 0.0);
    const bool cached = kOrderCache && q.tie && resting && (order_cache >= 0);
    SALP_COUNT(0, __any(q.tie && !cached));
    if (__any(q.tie && !cached)) {
      exact_order_reg<FMAX, KMAX>(e, Ksel, q);        // every lane: the exact order is the order
      if (kOrderCache) {
        int pack = 0;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) pack |= ((q.idx[s] + 1) & 31) << (5 * s);      // 5 bits per entry: slot + 1, 0 = none
        order_cache = resting ? pack : order_cache;
      }
    }
    if (kOrderCache) {
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const int k = ((order_cache >> (5 * s)) & 31) - 1;
        q.idx[s] = cached ? k : q.idx[s];
      }
    }
  }
  if (all_live) resolve_f32<KMAX, true>(m, Ksel, xf, yf, q);
  else resolve_f32<KMAX, false>(m, Ksel, xf, yf, q);
  refine_near<FMAX, KMAX>(e, Ksel, tol_c0, q);
  {
    const double mg = CV(margin);
    o.collision = (e.x - r <= mg) || (e.x + r >= CV(wall_hi_x)) || (e.y - r <= mg) || (e.y + r >= CV(wall_hi_y));
  }
  if (__any(o.collision)) {
    const double pen = cold->collision_penalty;
    rew = o.collision ? rew + pen : rew;
  }
  o.rel = relative_heading<true>(q.by[0], q.bx[0], (float)e.th);
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
// lane L takes them into its registers (and its mirror column) once per batch (~8 instructions per filled slot).
template <int FMAX, bool STD, bool INREG>
__device__ __forceinline__ void place_food_coop_reg(Env<FMAX>& e, FoodF32<FMAX, INREG>& ff, const MirrorLds& m, int lane, const DevParams& P, uint64_t genv,
                                                    int todo, int limit, double2* scratch) {
  SALP_CONSTS;
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
    uint32_t present = ~empty & ((1u << P.F) - 1u);   // slots of env L that hold a food: only those can reject a candidate
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
        if ((present >> k) & 1u) {                 // wave-uniform; an autoreset starts with every slot empty: no test at all
          const double fxk = bcast_lane(e.fx[k], L), fyk = bcast_lane(e.fy[k], L);
          const double dx = x - fxk, dy = y - fyk;
          ok = ok && !(dx * dx + dy * dy < min2);
        }
      }
      int j = 0;                                   // first candidate of this batch not yet judged
      uint32_t filled = 0u;                        // slots of env L that received a point in this batch
      // First batch of an autoreset (no food yet, the empty slots are 0 .. nE-1): the same sequential rule on bit masks.  A
      // candidate's standing is `blocked` — the union of the accepted candidates' conflict rows (one ballot each) — instead of
      // a per-lane flag re-tested every round, and the accepted points are handed over together: the k-th accepted lane
      // writes slot k.  ~25 instead of ~52 instructions per accepted food (12 foods per reset, 5 % of the wavefront-steps).
      const int nE = __builtin_popcount(empty);
      if (present == 0u && todo_l > 1 && empty == ((nE >= 32) ? ~0u : ((1u << nE) - 1u))) {
        const unsigned long long okm0 = __ballot(ok);
        unsigned long long blocked = 0ull, acc = 0ull;
        while (todo_l > 0 && j < kFoodLanes) {
          const unsigned long long okm = okm0 & ~blocked & (~0ull << j);
          const int first_ok = okm ? (__ffsll((long long)okm) - 1) : kFoodLanes;
          const int forced = j + (limit_l - attempts);          // accepted whatever it is
          const int a = __builtin_amdgcn_readfirstlane(first_ok < forced ? first_ok : forced);
          if (a >= kFoodLanes) { attempts += kFoodLanes - j; j = kFoodLanes; break; }
          const double ax = bcast_lane(x, a), ay = bcast_lane(y, a);
          const double dx = x - ax, dy = y - ay;
          blocked |= __ballot(dx * dx + dy * dy < min2);
          acc |= 1ull << a;
          todo_l -= 1;
          attempts = 0;
          j = a + 1;
        }
        const int nacc = __builtin_popcountll(acc);
        const int nput = nacc < nE ? nacc : nE;
        const int rank = __builtin_popcountll(acc & ((1ull << lane) - 1ull));
        if (((acc >> lane) & 1ull) && rank < nput) scratch[rank] = make_double2(x, y);
        filled = (nput >= 32) ? ~0u : ((1u << nput) - 1u);
        empty &= ~filled;
      }
      else
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
      present |= filled;
      if (filled) {                                // wave-uniform: lane L takes this batch's points (a later batch tests against them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < FMAX; ++k) {
          if ((filled >> k) & 1u) {                // wave-uniform
            const double2 v = scratch[k];          // same address in every lane: LDS broadcast
            if (lane == L) { e.fx[k] = v.x; e.fy[k] = v.y; ff.set(k, v.x, v.y); m.set(k, v.x, v.y); }
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
