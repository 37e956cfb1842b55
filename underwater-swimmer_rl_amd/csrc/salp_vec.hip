// salp_vec.hip — fused SALP step/rollout kernels for gfx950 and the C ABI of include/salp_vec.h.
//
// Kernel design:
//   * one SALP per lane, 256-thread workgroups (4 wavefronts), env index = blockIdx*256 + tid;
//   * state is struct-of-arrays in HBM (row-major [quantity][env], 8-byte and 4-byte rows), read
//     once at kernel entry, held in VGPRs across the `horizon` steps, written once at exit;
//   * per step each wavefront stages its 64 observation rows (64 x obs_dim floats) in its private
//     LDS tile and streams them out as whole 16-byte-per-lane coalesced stores, so the
//     [horizon][n_envs][obs_dim] row-major output is written as contiguous 64*obs_dim*4-byte
//     runs per wavefront; actions are prefetched one step ahead;
//   * episode / reward statistics are reduced with wavefront shuffles, then one 64-bit integer
//     atomic per block and statistic into one of 64 line-sized replicas (order-independent).
// No MFMA: there is no dense contraction on this path; the bound is HBM write bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include <new>
#include <string>
#include <type_traits>

#include "../../include/salp_vec.h"
#include "salp_device.h"
#include "salp_food_lds.h"
#include "salp_food_reg.h"

using namespace salp;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;
constexpr int kWave = 64;
// Wavefronts per SIMD the launch bounds ask for (workgroups of 256 threads per CU), by food-slot count:
//   one food 4;  4 / 8 slots 4 — 128 VGPRs, their LDS (32 / 40 KB) allows four workgroups per CU; without the bound several
//   signatures landed on 129 = three per SIMD, up to 22 % slower (profiles/r03/ab_notes.md sessions 14, 20);  12 slots 3 —
//   <= 168 VGPRs, 49 KB of LDS (four: a timing build at 128 VGPRs / 36 KB was 13 % slower, session 17);  16 slots 3 for the
//   literal-constant unpredicated kernels (half-height tile, session 19), else 2;  generic K: 2 (12 slots) / 1 (16 slots).
constexpr int waves_per_simd(int fmax, int kmax, bool std_consts, bool ragged) {
  return fmax <= 1 ? 4 : (kmax != 3 ? (fmax <= 12 ? 2 : 1) : (fmax <= 8 ? 4 : (fmax <= 12 ? 3 : ((std_consts && !ragged) ? 3 : 2))));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct IOPtrs {
  const float* act;       // [H][n][act_dim] or null (device-generated)
  float* obs;             // [H][n][obs_dim]
  float* reward;          // [H][n]
  uint8_t* terminated;    // [H][n]
  uint8_t* truncated;     // [H][n]
  float* final_obs;       // [H][n][obs_dim] rows of finished envs only
  int32_t* info;          // [H][n][3]
  float* act_out;         // [H][n][act_dim]
  DevStats* stats;        // [SALP_STATS_REPLICAS] or null
  int64_t global_step;    // step index of t = 0 (device-generated actions)
};

#ifdef SALP_EXP_STAMPS
constexpr int kStampWaves = 8192;
__device__ uint32_t salp_stamp_out[kStampWaves * 16];
#endif

// Device-memory copy of the launch constants for the RARE paths of the rollout kernel (respawn / autoreset
// region, exact capture pass, state write-back).  Everything the per-step path needs arrives by value in
// `P` (scalar registers); what only the rare paths read is fetched from this block when they run, so it
// does not occupy scalar registers across the step loop (the by-value copy alone left ~125 SGPRs spilled
// to VGPR lanes, ~110 v_readlane / v_writelane per step in the 12-food kernel).  Immutable after create;
// base_num_food_items, the one field a caller may change between launches, is always taken from `P`.
struct ColdBlock {
  DevParams P;
  DevState S;
  uint32_t seed[2];   // the key of the draw streams: P.seed (in every copy of P) points here
};

// FULL = the common rollout signature (act, obs, reward, terminated, truncated all present; no
// final_obs / info): no per-step null tests.
// RAGGED = false: every wavefront of the launch is either full (64 envs) or empty, so no store is
// predicated and the compiler can count the stores issued after the action prefetch (it then waits
// for the prefetch alone instead of draining all stores with s_waitcnt vmcnt(0) every step).
// RAGGED = true: the same loop with per-lane predicates; the host launches it for the last
// n % 64 envs only (one wavefront).  `env_begin/env_end`: the env range of this launch.
// GEN = actions are generated in the kernel (salp_vec_rollout with act == NULL): no read stream at
// all — the per-step 256-B action read costs the write stream ~10 % (HBM read/write turnarounds,
// profiles/r01/ab_notes.md) — and, if act_out is given, the actions are written out instead.
template <int FMAX, int KMAX, bool FORCED, bool STD, int SIG, bool RAGGED, bool GEN>
__global__ __launch_bounds__(kBlock, waves_per_simd(FMAX, KMAX, STD, RAGGED)) void salp_rollout_kernel(DevParams P_arg, DevState S, IOPtrs io, int H, int64_t env_begin, int64_t env_end, const ColdBlock* __restrict__ cold) {
  // STD = false: where the hot path's constants come from (open_consts, salp_device.h) — the device copy, function by
  // function, for the 4- and 8-slot kernels; the by-value launch parameters for the others
  constexpr bool MEMC = !STD && KMAX == 3 && (FMAX == 4 || FMAX == 8);
  DevParams P_pol = P_arg;
  P_pol.use_mem = MEMC ? 1 : 0;
  const DevParams& P = STD ? P_arg : P_pol;
  constexpr bool FULL = SIG != 0;         // obs, reward, terminated, truncated all present: their stores are unconditional
  constexpr bool EXTRAS = SIG != 1;       // final_obs / info may be present (tested per use; SIG 0: every output is tested)
  constexpr int QMAX = 3 + KMAX;          // float4 per observation row
  // LDS tile of the wavefront's 64 observation rows.  Banking (MI355X_MICROARCH.md §LDS): ds_write_b128 goes
  // in 8 groups of 8 lanes over banks (a/4) mod 32, ds_read_b128 in 4 groups of 16 lanes ({0-3,12-15,20-27},
  // ...) over banks (a/4) mod 64.  K = 3 (Q = 6 float4 per row): unpadded 96-B rows with the float4 column
  // XOR-ed by bit 2 of the row — conflict-free for the row writes AND for the flush reads (profiles/isa_lds_model.py;
  // the 112-B padded pitch of round 1 was conflict-free for the writes only: 2-way on the reads).
  // Other K (generic instantiation, Q possibly odd): the padded pitch.
  constexpr bool SWZ = (KMAX == 3);
  constexpr int PITCH = SWZ ? 4 * QMAX : 4 * QMAX + 4;     // LDS row pitch in floats
  // Where the food positions of a multi-food env live: up to 12 slots in VGPRs with an fp32 mirror in LDS
  // (salp_food_reg.h), above that in LDS (salp_food_lds.h); one food is plain registers.
  constexpr bool REGF = FMAX > 1 && (FMAX <= 12 || KMAX == 3);   // K = 3: every slot count; generic K: up to 12 slots
  constexpr bool LDSF = FMAX > 1 && !REGF;                         // generic K with 13..16 slots
  constexpr bool MULTI = REGF || LDSF;
  double2* food_lds = nullptr;
  if constexpr (LDSF) {     // (declared only where it exists: a one-element stand-in would cost the 8-slot kernel its fourth workgroup per CU)
    __shared__ __attribute__((aligned(16))) double2 food_lds_block[(kBlock / kWave) * FMAX * kWave];
    food_lds = food_lds_block;
  }
  // Per-wavefront LDS region: the observation tile (64 rows of PITCH floats) and, for the register-food kernels, the
  // fp32 mirror of the food positions behind it (8 B per slot and lane, salp_food_reg.h; lives for the whole launch).
  // 12 slots, K = 3: 6144 + 6144 B per wavefront, 49 KB per workgroup -> 3 workgroups per CU.  The tile's bytes, idle in
  // the middle of a step, are lent to the rare paths as scratch: the exact order's distances (8 B per slot and lane,
  // <= the tile for every instantiation) and the placement's accepted points.
  // HALF: a tile of 32 rows, written and flushed twice per step (lanes 0-31, then 32-63): 3072 instead of 6144 B per wavefront.
  // The 16-slot kernel's mirror is 8192 B per wavefront: 57344 B per workgroup allowed two workgroups per CU, 45056 B allow three
  // (and without the fp32 register copies it holds 159 VGPRs <= 168).
  constexpr bool HALF = !RAGGED && REGF && KMAX == 3 && FMAX == 16 && STD;
  constexpr int TILE_ROWS = HALF ? kWave / 2 : kWave;
  constexpr int TILE_FLOATS = TILE_ROWS * PITCH;
  constexpr int WAVE_FLOATS = TILE_FLOATS + (REGF ? kWave * 2 * FMAX : 0);
  static_assert(!REGF || (4 * FMAX <= TILE_ROWS * PITCH && 96 <= TILE_ROWS * PITCH), "the placement's FMAX accepted points (16 B each) must fit in the tile");
  __shared__ __attribute__((aligned(16))) float lds[(kBlock / kWave) * WAVE_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  // the wavefront number and everything derived from it (first env, row bases, tile address) as scalars: index
  // arithmetic then is a scalar base plus the lane, not 64-bit vector registers held across the loop
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int64_t env0 = env_begin + (int64_t)blockIdx.x * kBlock + (int64_t)wave * kWave;  // first env of this wavefront
  const int64_t env = env0 + lane;
  const int rows = (int)((env_end - env0) < kWave ? ((env_end - env0) > 0 ? (env_end - env0) : 0) : kWave);
  const bool active = RAGGED ? (env < env_end) : true;     // !RAGGED: rows is 64 or 0
  // the env whose state the lane loads: its own — clamped into the range for the lanes past the end (RAGGED), or the
  // range's first wavefront for a whole wavefront past the end (!RAGGED; it runs no step and stores nothing)
  const int64_t envc = RAGGED ? ((env < env_end) ? env : (env_end - 1)) : ((rows > 0 ? env0 : env_begin) + lane);
  const uint64_t genv = P.env_base + (uint64_t)envc;
  const int K = (KMAX == 3) ? 3 : P.K;
  const int Q = 3 + K;
  const int OD = 4 * Q;
  const int AD = FORCED ? 1 : 2;
  float* tile = lds + wave * WAVE_FLOATS;
  float4* myrow4 = reinterpret_cast<float4*>(tile + (HALF ? (lane & (TILE_ROWS - 1)) : lane) * PITCH);
  // SWZ: column q of this lane's row sits at float4 (q ^ s), s = bit 2 of the row = q + s for even q, q - s for odd q
  const int swz = SWZ ? ((lane >> 2) & 1) : 0;
  float4* myrow_even = myrow4 + swz;
  float4* myrow_odd = myrow4 - swz;

  // Tile flush plan, fixed for the whole launch: float4 number f = j*64 + lane of the wavefront's
  // [rows x Q] tile lives at LDS row f / Q, column f % Q and goes to global float4 f of the run.
  int lds_off[QMAX];      // float offset inside the tile (-1: nothing to move, RAGGED only)
#pragma unroll
  for (int j = 0; j < QMAX; ++j) {
    const int f = j * kWave + lane;
    const int r = f / Q;
    const int c = f - r * Q;
    lds_off[j] = (!RAGGED || f < rows * Q) ? (r * PITCH + 4 * (SWZ ? (c ^ ((r >> 2) & 1)) : c)) : -1;
  }

  // the six source addresses of the flush as ONE register each: left to itself the compiler keeps the row term and
  // the swizzled column term of every address in separate registers and adds them on every step
  const v4f* flush_src[QMAX];
#pragma unroll
  for (int j = 0; j < QMAX; ++j) {
    int i = wave * WAVE_FLOATS + (lds_off[j] >= 0 ? lds_off[j] : 0);
    asm volatile("" : "+v"(i));
    flush_src[j] = reinterpret_cast<const v4f*>(lds + i);
  }
  // Event statistics (episodes, terminations, food, ...) change on rare steps only: they are accumulated with LDS integer
  // atomics inside the rare-event branch — in 128 bytes of the wavefront's own tile, idle there — and leave with ONE
  // global atomic instruction per wavefront and event step, into one of 64 line-sized replicas.  (Rounds 1-2 kept a
  // 128-byte block of LDS per workgroup for the whole launch: with the mirror that was exactly what pushed the 8-slot
  // kernel from four workgroups per CU to three.)
  unsigned long long* const wave_stats = reinterpret_cast<unsigned long long*>(tile) + 32;   // tile bytes 256..383 (placement scratch: 0..191)
  DevStats* const stats_replica = io.stats ? io.stats + (blockIdx.x % SALP_STATS_REPLICAS) : nullptr;

  using EnvT = std::conditional_t<LDSF, EnvCore, Env<FMAX>>;
  EnvT e;
  const FoodLds food{food_lds + (LDSF ? (wave * FMAX * kWave + lane) : 0)};
  const MirrorLds mir{reinterpret_cast<float2*>(tile + TILE_FLOATS) + lane};   // REGF only
  FoodF32<REGF ? FMAX : 1, food_in_registers(FMAX, KMAX, STD, SIG == 1) && !HALF> ff;   // REGF: fp32 roundings of the food positions (salp_food_reg.h)
  FoodScan<KMAX> fq;          // MULTI: nearest-K selection of the current food set around the current pose
  int nlive = 0;              // MULTI: live foods of this env, recounted whenever the food set changes
  int order_cache = -1;       // REGF: remembered exact order of a resting swimmer's foods (step_env_reg)
  if constexpr (LDSF) {
    load_core(e, S, P, envc);
    for (int k = 0; k < P.F; ++k) {
      const double fx = S.f[(SF_FOOD0 + k) * P.pitch + envc];
      food.set(k, fx, S.f[(SF_FOOD0 + P.F + k) * P.pitch + envc]);
      nlive += is_none(fx) ? 0 : 1;
    }
    for (int k = P.F; k < FMAX; ++k) food.clear(k);   // the scans run over whole groups of four slots
  } else {
    load_env(e, S, P, envc);
    if constexpr (REGF) {
#pragma unroll
      for (int k = 0; k < FMAX; ++k) {
        nlive += is_none(e.fx[k]) ? 0 : 1;
        ff.set(k, e.fx[k], e.fy[k]);
        mir.set(k, e.fx[k], e.fy[k]);
      }
    }
  }

  double st_reward = 0.0;   // the one per-step statistic

  float a0 = 0.f, a1 = 0.f;
  U4 aw0 = {0u, 0u, 0u, 0u}, aw1 = {0u, 0u, 0u, 0u};   // GEN: the current Philox block of each action component
  if (!GEN) {
    a0 = io.act[envc * AD];
    a1 = FORCED ? 0.f : io.act[envc * AD + 1];
  }
  // Everything loaded so far is complete before the loop is entered: otherwise the waitcnt pass keeps
  // a conservative `s_waitcnt vmcnt(1)` on the first use of the action inside the loop (for the entry
  // path), and that wait drains the previous step's stores on every iteration.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

  const int Hrun = (rows > 0) ? H : 0;   // a wavefront past the end of the range runs zero steps
#ifdef SALP_EXP_STAMPS
  StampAcc stamps;
  for (int i = 0; i < 12; ++i) stamps.acc[i] = 0u;
  stamps.last = (uint32_t)__builtin_amdgcn_s_memtime();
  StampAcc* const stamps_ = &stamps;
  const uint32_t stamp_real0 = (uint32_t)__builtin_amdgcn_s_memrealtime();   // 100 MHz: wall clock of the wavefront's loop
#endif
#pragma unroll 1
  for (int t = 0; t < Hrun; ++t) {
    const int64_t rowbase = (int64_t)t * P.n;
    float c0 = a0, c1 = a1;
    if (GEN) {
      // device action stream (include/salp_vec.h "Randomness"): word ts & 3 of block ts >> 2
      const uint32_t ts = (uint32_t)(io.global_step + t);
      if (t == 0 || (ts & 3u) == 0u) {   // wave-uniform
        aw0 = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), ts >> 2, 1u, P.seed[0], P.seed[1]);
        if (!FORCED) aw1 = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), ts >> 2, 2u, P.seed[0], P.seed[1]);
      }
      const uint32_t k = ts & 3u;
      const uint32_t w0 = (k == 0) ? aw0.x : (k == 1) ? aw0.y : (k == 2) ? aw0.z : aw0.w;
      if (FORCED) {
        c0 = (float)(w0 >> 8) * 1.1920928955078125e-7f - 1.0f;              // [-1, 1)
      } else {
        const uint32_t w1 = (k == 0) ? aw1.x : (k == 1) ? aw1.y : (k == 2) ? aw1.z : aw1.w;
        c0 = (float)(w0 >> 8) * 5.9604644775390625e-8f;                     // inhale control in [0, 1)
        c1 = (float)(w1 >> 8) * 1.1920928955078125e-7f - 1.0f;
      }
      if (io.act_out && active) {
        io.act_out[(rowbase + env) * AD] = c0;
        if (!FORCED) io.act_out[(rowbase + env) * AD + 1] = c1;
      }
    } else {  // prefetch the next step's action (the last step re-reads its own: keeps the load unconditional)
      const int64_t nb = (rowbase + ((t + 1 < H) ? P.n : 0) + envc) * AD;
      a0 = io.act[nb];
      if (!FORCED) a1 = io.act[nb + 1];
    }

#ifdef SALP_EXP_STORE_ONLY   // experiment build: no simulation, only the output stream
    StepOut o; o.rmax = 30.0; o.reward = c0; o.rel = c1; o.rel_valid = true;
    o.terminated = o.truncated = o.collision = o.collected = false;
#else
    StepOut o;
    if constexpr (LDSF) o = step_env_lds<KMAX, FORCED, STD>(e, food, P, genv, c0, c1, K, fq, nlive);
    else if constexpr (REGF) {
#ifdef SALP_EXP_STAMPS
      { StampAcc* stamps_ = &stamps; SALP_STAMP(0); }
      o = step_env_reg<FMAX, KMAX, FORCED, STD>(e, ff, mir, P, genv, c0, c1, K, fq, nlive, order_cache, &cold->P, &stamps);
#else
      o = step_env_reg<FMAX, KMAX, FORCED, STD>(e, ff, mir, P, genv, c0, c1, K, fq, nlive, order_cache, &cold->P);
#endif
    }
    else {
#ifdef SALP_EXP_STAMPS
      { StampAcc* stamps_ = &stamps; SALP_STAMP(0); }
      o = step_env<FMAX, FORCED, STD>(e, P, genv, c0, c1, &stamps);
#else
      o = step_env<FMAX, FORCED, STD>(e, P, genv, c0, c1);
#endif
    }
#endif
    const bool done = o.terminated || o.truncated;
    double rmax = o.rmax;
    bool have_rel = o.rel_valid;

    if (active) {
      // reward: one dword per lane (256 B per wavefront); flags: one byte per lane.  (Rebuilding the
      // 64 flag bytes from a ballot and storing 16 dwords was measured: no faster in the memory
      // pipeline and slower overall, profiles/r01/ab_notes.md.)
      // (cache-policy bits on these small stores, and issuing them after the rows: measured, slower — r02 sessions 11, 14, 15)
      if (FULL || io.reward) io.reward[rowbase + env] = o.reward;
      if (FULL || io.terminated) io.terminated[rowbase + env] = o.terminated ? 1 : 0;
      if (FULL || io.truncated) io.truncated[rowbase + env] = o.truncated ? 1 : 0;
      if (EXTRAS && io.info) {
        int32_t* ip = io.info + (rowbase + env) * SALP_INFO_COLS;
        ip[SALP_INFO_FOOD_COLLECTED] = e.fc;
        ip[SALP_INFO_STEPS_SINCE_FOOD] = e.ssf;
        ip[SALP_INFO_COLLISION] = o.collision ? 1 : 0;
      }
    }
    st_reward += (double)o.reward;

    SALP_STAMP(6);
    // rare events: respawn of a collected food (snake:179-180), then same-step autoreset
    int todo = (o.collected && P.respawn) ? 1 : 0;
    const int F_base = P.F_base;
#ifdef SALP_EXP_NO_RARE   // experiment build: price of the respawn / autoreset region (results are wrong)
    if (false) {
#else
    if (__any(o.collected || done)) {
#endif
      const DevParams& C = cold->P;   // rare path: constants from memory, not from scalar registers
      int limit = 50;
      if (o.collected || done) order_cache = -1;      // the food set (or the episode) changes
      if (io.stats) {
        if (lane < 16) wave_stats[lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) {
          if (o.collected) atomicAdd(&wave_stats[ST_FOOD], 1ull);
          if (o.collision) atomicAdd(&wave_stats[ST_COLL], 1ull);
          if (done && C.autoreset) {
            atomicAdd(&wave_stats[ST_EPISODES], 1ull);
            atomicAdd(&wave_stats[o.terminated ? ST_TERM : ST_TRUNC], 1ull);
            atomicAdd(&wave_stats[ST_EPLEN], (unsigned long long)e.eplen);
            atomicAdd(&wave_stats[ST_EPRET], (unsigned long long)__double2ll_rn(e.epret * SALP_FIXED_SCALE));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane <= ST_EPRET) {
          const unsigned long long v = wave_stats[lane];
          if (v != 0) atomicAdd(&stats_replica->v[lane], v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the placement's scratch and the tile rows come next
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      SALP_STAMP(10);
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && done && C.autoreset) {
          if (EXTRAS && io.final_obs && active) {
            float fo[12 + 4 * KMAX];
            if constexpr (LDSF) {   // the terminal observation sees the respawned food (pass 0)
              bool c_; int h_;
              scan_foods<KMAX, false, true>(food, C.F, e.x, e.y, 0.0, fq, c_, h_, nlive);
              if (__any(fq.tie)) exact_order_lds<KMAX>(food, C.F, K, e.x, e.y, fq);
              resolve<KMAX>(food, K, e.x, e.y, fq);
              observe_lds<KMAX, STD>(e, C, rmax, K, fq, nlive, false, 0.f, fo);
            } else if constexpr (REGF) {
              select_foods_reg<FMAX, KMAX, false, true>(e, ff, mir, K, (STD ? StdConsts::tie_c0 : C.tie_c0), fq, nlive);
              observe_lds<KMAX, STD>(e, C, rmax, K, fq, nlive, false, 0.f, fo);
            } else {
              observe<FMAX, KMAX, STD>(e, C, rmax, have_rel, o.rel, fo);
            }
            float4* dst = reinterpret_cast<float4*>(io.final_obs + (rowbase + env) * OD);
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
              if (q < Q) dst[q] = make_float4(fo[4 * q], fo[4 * q + 1], fo[4 * q + 2], fo[4 * q + 3]);
          }
          if constexpr (LDSF) {
            todo = reset_core<STD>(e, C, genv, F_base);
            for (int k = 0; k < C.F; ++k) food.clear(k);
          } else {
            todo = reset_pose<FMAX, STD>(e, C, genv, F_base);
            if constexpr (REGF) {
#pragma unroll
              for (int k = 0; k < FMAX; ++k) { ff.clear(k, true); mir.clear(k); }
            }
          }
          limit = 100;
          rmax = STD ? StdConsts::R : C.R;
          have_rel = false;
        }
        if constexpr (LDSF) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          place_food_coop<FMAX, STD>(e, food_lds + wave * FMAX * kWave, lane, C, genv, todo, limit);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        else if constexpr (REGF) place_food_coop_reg<FMAX, STD, food_in_registers(FMAX, KMAX, STD, SIG == 1) && !HALF>(e, ff, mir, lane, C, genv, todo, limit, reinterpret_cast<double2*>(tile));
        else place_food<FMAX, STD>(e, C, genv, todo, limit);
        todo = 0;
      }
      SALP_STAMP(11);
      if constexpr (LDSF) {   // the food set (or the pose) changed: select again for the observation
        bool c_; int h_;
        scan_foods<KMAX, false, true>(food, C.F, e.x, e.y, 0.0, fq, c_, h_, nlive);
        if (__any(fq.tie)) exact_order_lds<KMAX>(food, C.F, K, e.x, e.y, fq);
        resolve<KMAX>(food, K, e.x, e.y, fq);
        have_rel = false;
      }
      if constexpr (REGF) {   // likewise (the exact order's scratch and the placement's are the same idle tile bytes, used in turn)
        select_foods_reg<FMAX, KMAX, false, true>(e, ff, mir, K, (STD ? StdConsts::tie_c0 : C.tie_c0), fq, nlive);
        have_rel = false;
      }
    }
    SALP_STAMP(7);

    if (FULL || io.obs) {
      float ob[12 + 4 * KMAX];
      if constexpr (REGF) {
        // all FMAX slots of every lane alive (the steady state with respawn) => every lane shows K foods
        if ((KMAX <= FMAX) && K >= 1 && __all(nlive == FMAX)) observe_lds<KMAX, STD, true>(e, P, rmax, K, fq, nlive, have_rel, o.rel, ob);
        else observe_lds<KMAX, STD>(e, P, rmax, K, fq, nlive, have_rel, o.rel, ob);
      } else if constexpr (LDSF) observe_lds<KMAX, STD>(e, P, rmax, K, fq, nlive, have_rel, o.rel, ob);
      else observe<FMAX, KMAX, STD>(e, P, rmax, have_rel, o.rel, ob);
      SALP_STAMP(8);
      // (per-lane 96-B rows stored straight from registers, without the LDS transpose: 2.1x slower, r01 ab_notes)
      v4f* gout = reinterpret_cast<v4f*>(io.obs + (rowbase + env0) * OD) + lane;
      if constexpr (HALF) {
        // float4 f = j*64 + lane (j = 0..2) of a half lives at tile row f / 6, column f % 6 (flush_src[0..2]) and goes to global
        // float4 (3 h + j) * 64 + lane of the wavefront's 64-row block: the same 1-KB stores as the full plan, three per half
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((lane >> 5) == h) {
#pragma unroll
            for (int q = 0; q < QMAX; ++q)
              ((q & 1) ? myrow_odd : myrow_even)[q] = make_float4(ob[4 * q], ob[4 * q + 1], ob[4 * q + 2], ob[4 * q + 3]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          v4f tv[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) tv[j] = *flush_src[j];
#pragma unroll
          for (int j = 0; j < 3; ++j) __builtin_nontemporal_store(tv[j], &gout[(3 * h + j) * kWave]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the reads of this half before the writes of the next
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      } else {
#pragma unroll
      for (int q = 0; q < QMAX; ++q)   // 16-B LDS stores, conflict-free (see the tile layout above)
        if (q < Q) ((q & 1) ? myrow_odd : myrow_even)[q] = make_float4(ob[4 * q], ob[4 * q + 1], ob[4 * q + 2], ob[4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        v4f tv[QMAX];
#pragma unroll
        for (int j = 0; j < QMAX; ++j)
          if (j < Q && (!RAGGED || lds_off[j] >= 0)) tv[j] = *flush_src[j];
        // Write-once stream far larger than L2 / Infinity Cache.  The unpredicated one-food K = 3 kernels issue the row stores
        // as `global_store_dwordx4 ... sc1 nt` — system scope (written through, not retained in L2) plus the
        // streaming hint: measured -4.2 % on the one-food kernel against `nt` alone, which is what
        // __builtin_nontemporal_store emits and what was -2.4 % against plain stores (profiles/r02/ab_notes.md
        // session 14).  The compiler has no builtin for the scope bits of a plain global store, hence the asm; its
        // waitcnt pass does not count these stores, which is safe: vmcnt retires in order, so a wait computed
        // without them can only wait longer, and nothing reads the stream back.  The hazard recogniser does not see
        // them either: gfx9 wants one wait state between a store of more than 64 bits and a VALU write of its data
        // registers, so the last store carries an `s_nop 0` (the end-of-step drain below follows anyway: !MULTI).
        if constexpr (!RAGGED && QMAX == 6 && !MULTI) {   // the write-bound one-food kernel; the VALU-bound multi-food ones: +1 %, not used
          v4f* const gout4 = gout + 4 * kWave;     // the instruction's immediate offset reaches 4095 B: two bases
#pragma unroll
          for (int j = 0; j < QMAX - 1; ++j)
            asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1 nt"
                         :: "v"(j < 4 ? gout : gout4), "v"(tv[j]), "n"((j & 3) * kWave * 16) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1 nt\n\ts_nop 0"
                       :: "v"(gout4), "v"(tv[QMAX - 1]), "n"(((QMAX - 1) & 3) * kWave * 16) : "memory");
        } else
#pragma unroll
        for (int j = 0; j < QMAX; ++j)
          if (j < Q && (!RAGGED || lds_off[j] >= 0)) __builtin_nontemporal_store(tv[j], &gout[j * kWave]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    // One-food kernel (write-bound): drain this step's stores before the next step.  Measured
    // (profiles/r01/ab_notes.md): letting stores run ahead (vmcnt(9)) is 2-6 % SLOWER than draining —
    // wavefronts that stay in step keep the write stream of all CUs inside one contiguous [N x 96 B] slab
    // at a time.  The multi-food kernels are issue-bound at 2 wavefronts per SIMD: there the drain is a
    // stall nothing hides (-6.4 % without it, profiles/r02/ab_notes.md session 2).
    if constexpr (!MULTI) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    SALP_STAMP(9);
  }
#ifdef SALP_EXP_STAMPS
  if (lane == 0 && rows > 0) {
    const int gw = (int)((env0 - env_begin) / kWave) & (kStampWaves - 1);
    for (int i = 0; i < 12; ++i) salp_stamp_out[gw * 16 + i] = stamps.acc[i];
    salp_stamp_out[gw * 16 + 12] = stamp_real0;                                           // start (10-ns ticks, low word)
    salp_stamp_out[gw * 16 + 13] = (uint32_t)__builtin_amdgcn_s_memrealtime();            // end
  }
#endif

  if (rows > 0 && active) {
    const DevParams& C = cold->P;
    const DevState CS = cold->S;
    if constexpr (LDSF) {
      store_core(e, CS, C, env);
      for (int k = 0; k < C.F; ++k) {
        double fx, fy;
        food.get(k, fx, fy);
        CS.f[(SF_FOOD0 + k) * C.pitch + env] = fx;
        CS.f[(SF_FOOD0 + C.F + k) * C.pitch + env] = fy;
      }
    } else {
      store_env(e, CS, C, env);
    }
  }

  if (io.stats) {
    // reward sum and env-step count: wavefront shuffles, the four wavefronts' sums through 64 bytes of the (now idle)
    // first tile, then two 64-bit integer global atomics per WORKGROUP (per wavefront they cost the H = 1 step kernel 8 %:
    // 8192 atomics on a 14-us launch, profiles/r03/ab_notes.md session 10)
    if (!active || rows == 0) st_reward = 0.0;
    const double wr = wave_sum(st_reward);
    const int wact = wave_sum((active && rows > 0) ? 1 : 0);
    unsigned long long* const blk = reinterpret_cast<unsigned long long*>(lds);
    __syncthreads();                               // every wavefront is past its last tile flush
    if (lane == 0) {
      blk[2 * wave] = (unsigned long long)__double2ll_rn(wr * SALP_FIXED_SCALE);
      blk[2 * wave + 1] = (unsigned long long)((long long)wact * H);
    }
    __syncthreads();
    if (tid < 2) {
      unsigned long long v = 0ull;
      for (int w = 0; w < kBlock / kWave; ++w) v += blk[2 * w + tid];
      if (v != 0) atomicAdd(&stats_replica->v[tid == 0 ? ST_REWARD : ST_STEPS], v);
    }
  }
}

// Device-generated actions (salp_vec_rollout with act == NULL): a[t][env][j] from the env's
// action stream, Philox counter (env_lo, env_hi, global_step + t, 1 + j).
__global__ __launch_bounds__(kBlock) void salp_gen_actions_kernel(DevParams P, float* act, int H, int AD, int64_t global_step) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= P.n) return;
  const uint64_t genv = P.env_base + (uint64_t)i;
  U4 w = {0u, 0u, 0u, 0u}, w2 = {0u, 0u, 0u, 0u};
  for (int t = 0; t < H; ++t) {
    const uint32_t ts = (uint32_t)(global_step + t);
    if (t == 0 || (ts & 3u) == 0u) {
      w = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), ts >> 2, 1u, P.seed[0], P.seed[1]);
      if (AD == 2) w2 = philox4x32_10((uint32_t)genv, (uint32_t)(genv >> 32), ts >> 2, 2u, P.seed[0], P.seed[1]);
    }
    const uint32_t k = ts & 3u;
    const uint32_t x0 = (k == 0) ? w.x : (k == 1) ? w.y : (k == 2) ? w.z : w.w;
    const uint32_t x1 = (k == 0) ? w2.x : (k == 1) ? w2.y : (k == 2) ? w2.z : w2.w;
    const int64_t o = ((int64_t)t * P.n + i) * AD;
    if (AD == 1) {
      act[o] = (float)(x0 >> 8) * 1.1920928955078125e-7f - 1.0f;
    } else {
      act[o] = (float)(x0 >> 8) * 5.9604644775390625e-8f;
      act[o + 1] = (float)(x1 >> 8) * 1.1920928955078125e-7f - 1.0f;
    }
  }
}

// reset(mask) + observation
template <int FMAX, int KMAX, bool STD>
__global__ __launch_bounds__(kBlock) void salp_reset_kernel(DevParams P, DevState S, const uint8_t* mask, float* obs, int do_reset) {
  const int64_t env = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (env >= P.n) return;
  const uint64_t genv = P.env_base + (uint64_t)env;
  Env<FMAX> e;
  load_env(e, S, P, env);
  const bool resetting = do_reset && (!mask || mask[env]);
  int todo = 0;
  if (resetting) {
    const int nf = reset_pose<FMAX, STD>(e, P, genv, P.F_base);
    // place_food loops until no lane of the wavefront has food left to place; lanes that are not
    // being reset pass todo = 0
    todo = nf;
  }
  place_food<FMAX, STD>(e, P, genv, todo, 100);
  if (resetting) {
    store_env(e, S, P, env);
  }
  if (obs) {
    const int K = (KMAX == 3) ? 3 : P.K;
    double a, b;
    shape_of<STD>(P, e.packed, e.water, a, b);
    float ob[12 + 4 * KMAX];
    observe<FMAX, KMAX, STD>(e, P, pymax(a, b), false, 0.f, ob);
    float4* dst = reinterpret_cast<float4*>(obs + env * (12 + 4 * K));
#pragma unroll
    for (int q = 0; q < 3 + KMAX; ++q)
      if (q < 3 + K) dst[q] = make_float4(ob[4 * q], ob[4 * q + 1], ob[4 * q + 2], ob[4 * q + 3]);
  }
}

// public snapshot <-> device layout
__global__ void salp_get_state_kernel(DevParams P, DevState S, double* f64, int32_t* i32) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const int64_t p = P.pitch, n = P.n;
  const uint32_t packed = (uint32_t)S.i[SI_PACKED * p + i];
  if (f64) {
    f64[SALP_F_X * n + i] = S.f[SF_X * p + i]; f64[SALP_F_Y * n + i] = S.f[SF_Y * p + i];
    f64[SALP_F_VX * n + i] = S.f[SF_VX * p + i]; f64[SALP_F_VY * n + i] = S.f[SF_VY * p + i];
    f64[SALP_F_THETA * n + i] = S.f[SF_TH * p + i]; f64[SALP_F_OMEGA * n + i] = S.f[SF_OM * p + i];
    f64[SALP_F_NOZZLE * n + i] = S.f[SF_NOZ * p + i];
    const double water = S.f[SF_WATER * p + i];
    f64[SALP_F_WATER * n + i] = water;
    double a, b;
    shape_of<false>(P, packed, water, a, b);
    f64[SALP_F_ELLIPSE_A * n + i] = a; f64[SALP_F_ELLIPSE_B * n + i] = b;
    for (int k = 0; k < 2 * P.F; ++k) f64[(SALP_F_FOOD0 + k) * n + i] = S.f[(SF_FOOD0 + k) * p + i];
  }
  if (i32) {
    i32[SALP_I_PHASE * n + i] = bw_phase(packed); i32[SALP_I_TIMER * n + i] = bw_timer(packed);
    i32[SALP_I_EXHALE_DUR * n + i] = bw_dur(packed); i32[SALP_I_SHAPE_HOLD * n + i] = bw_hold(packed);
    i32[SALP_I_STEPS_SINCE_FOOD * n + i] = S.i[SI_SSF * p + i];
    i32[SALP_I_FOOD_COLLECTED * n + i] = S.i[SI_FC * p + i];
    i32[SALP_I_RNG_COUNTER * n + i] = S.i[SI_RNG * p + i];
    i32[SALP_I_EPISODE_LENGTH * n + i] = S.i[SI_EPLEN * p + i];
  }
}

__global__ void salp_set_state_kernel(DevParams P, DevState S, const double* f64, const int32_t* i32) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const int64_t p = P.pitch, n = P.n;
  if (f64) {
    S.f[SF_X * p + i] = f64[SALP_F_X * n + i]; S.f[SF_Y * p + i] = f64[SALP_F_Y * n + i];
    S.f[SF_VX * p + i] = f64[SALP_F_VX * n + i]; S.f[SF_VY * p + i] = f64[SALP_F_VY * n + i];
    S.f[SF_TH * p + i] = f64[SALP_F_THETA * n + i]; S.f[SF_OM * p + i] = f64[SALP_F_OMEGA * n + i];
    S.f[SF_NOZ * p + i] = f64[SALP_F_NOZZLE * n + i]; S.f[SF_WATER * p + i] = f64[SALP_F_WATER * n + i];
    for (int k = 0; k < P.F; ++k) {
      double fx = f64[(SALP_F_FOOD0 + k) * n + i], fy = f64[(SALP_F_FOOD0 + P.F + k) * n + i];
      if (fx != fx || fy != fy) { fx = __builtin_nan(""); fy = __builtin_nan(""); }
      S.f[(SF_FOOD0 + k) * p + i] = fx; S.f[(SF_FOOD0 + P.F + k) * p + i] = fy;
    }
  }
  if (i32) {
    S.i[SI_PACKED * p + i] = (int32_t)pack_breath(i32[SALP_I_PHASE * n + i], i32[SALP_I_TIMER * n + i],
                                                  i32[SALP_I_EXHALE_DUR * n + i], i32[SALP_I_SHAPE_HOLD * n + i]);
    S.i[SI_SSF * p + i] = i32[SALP_I_STEPS_SINCE_FOOD * n + i];
    S.i[SI_FC * p + i] = i32[SALP_I_FOOD_COLLECTED * n + i];
    S.i[SI_RNG * p + i] = i32[SALP_I_RNG_COUNTER * n + i];
    S.i[SI_EPLEN * p + i] = i32[SALP_I_EPISODE_LENGTH * n + i];
  }
}

// salp_vec_reseed: the state a freshly created handle has before its initial reset (every row zero, draw counters
// included), the new key words, cleared statistics.  The reset kernel that follows on the same stream reads the new key.
__global__ __launch_bounds__(kBlock) void salp_reseed_kernel(ColdBlock* cold, DevStats* stats, int64_t pitch, int nf_rows,
                                                            uint32_t seed_lo, uint32_t seed_hi) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < pitch) {
    double* f = cold->S.f;
    int32_t* w = cold->S.i;
    for (int r = 0; r < nf_rows; ++r) f[(int64_t)r * pitch + i] = 0.0;
    for (int r = 0; r < SI_COUNT; ++r) w[(int64_t)r * pitch + i] = 0;
  }
  if (i == 0) { cold->seed[0] = seed_lo; cold->seed[1] = seed_hi; }
  if (i < (int64_t)SALP_STATS_REPLICAS * 16) stats[i / 16].v[i % 16] = 0ull;
}

// ------------------------------------------------------------------ host side
thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(_e == hipErrorOutOfMemory ? SALP_ERR_OOM : SALP_ERR_HIP,                         \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                              \
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

struct salp_vec {
  salp_config_t cfg;
  DevParams P;
  DevState S;
  int device;
  int64_t n;
  uint64_t seed;
  int64_t global_step;
  int obs_dim, act_dim, F, K;
  int fmax, kmax;
  int std_consts;        // constants equal the reference defaults -> literal-constant kernels
  DevStats* stats;       // device, SALP_STATS_REPLICAS replicas
  int stats_enabled;
  ColdBlock* cold;       // device copy of P and S for the rollout kernel's rare paths
  // staging for host-pointer calls (grown on demand)
  void* stage;
  size_t stage_bytes;
  float* act_buf;        // device-generated actions when the caller gives no act_out
  size_t act_bytes;
  size_t nf_rows;
  hipStream_t last_stream;   // the stream of the handle's most recent launch: what get_stats / destroy wait for
  int64_t last_launch[8];    // salp_vec_last_launch
  const void* last_kernel;   // the main (else the predicated) kernel of the most recent launch: salp_vec_last_kernel_resources
};

namespace {

DevParams make_params(const salp_config_t& c, int64_t n, int64_t pitch, uint64_t seed, int64_t base) {
  DevParams P;
  memset(&P, 0, sizeof(P));
  P.W = (double)c.width; P.H = (double)c.height;
  P.half_W = (double)c.width / 2; P.half_H = (double)c.height / 2;
  P.margin = c.tank_margin;
  P.wall_hi_x = (double)c.width - c.tank_margin; P.wall_hi_y = (double)c.height - c.tank_margin;
  P.R = c.base_radius;
  P.a_rest = c.base_radius * 1.3; P.b_rest = c.base_radius * 0.8; P.ab_full = c.base_radius * 1.1;
  P.da_inh = P.ab_full - P.a_rest; P.db_inh = P.ab_full - P.b_rest;
  P.da_exh = P.a_rest - P.ab_full; P.db_exh = P.b_rest - P.ab_full;
  P.max_nozzle = c.max_nozzle_angle; P.nozzle_rate = c.nozzle_response_rate;
  P.thrust_force = c.max_thrust_force; P.drag = c.drag_coefficient; P.ang_drag = c.angular_drag;
  P.exhale_dur_d = (double)c.exhale_duration;
  P.food_radius = c.food_radius; P.min_food_dist2 = c.min_food_distance * c.min_food_distance;
  P.food_xlo = c.tank_margin + c.food_radius;
  P.food_xspan = ((double)c.width - c.tank_margin - c.food_radius) - P.food_xlo;
  P.food_ylo = P.food_xlo;
  P.food_yspan = ((double)c.height - c.tank_margin - c.food_radius) - P.food_ylo;
  P.food_reward = c.food_reward; P.collision_penalty = c.collision_penalty;
  P.time_penalty = c.time_penalty; P.efficiency_bonus = c.efficiency_bonus;
  P.prox_w = c.proximity_reward_weight;
  P.inv_W = 1.0 / P.W; P.inv_H = 1.0 / P.H; P.inv_pi = 1.0 / 3.141592653589793;
  P.inv_R = 1.0 / c.base_radius; P.inv_max_nozzle = 1.0 / c.max_nozzle_angle;
  P.inv_diag = (float)(1.0 / sqrt(P.W * P.W + P.H * P.H));
  { const double L = P.W > P.H ? P.W : P.H; P.tie_c0 = (float)(1.4e-7 * L * L); }
  P.inhale_dur = c.inhale_duration; P.exhale_dur = c.exhale_duration;
  P.cycle_len = c.inhale_duration + c.exhale_duration + c.rest_duration;
  P.max_steps_wo_food = c.max_steps_without_food;
  P.F = c.num_food_items; P.K = c.max_observed_food;
  P.F_base = c.num_food_items;
  P.forced = c.forced_breathing != 0; P.random_food_count = c.random_food_count != 0;
  P.respawn = c.respawn_food != 0;
  P.autoreset = c.no_autoreset == 0;
  P.seed = nullptr;   // set by salp_vec_create once the ColdBlock exists
  P.env_base = (uint64_t)base; P.n = n; P.pitch = pitch;
  return P;
}

int validate(const salp_config_t* c) {
  if (!c) return fail(SALP_ERR_INVALID, "config is NULL");
  if (c->struct_size != sizeof(salp_config_t))
    return fail(SALP_ERR_INVALID, "salp_config_t.struct_size mismatch (ABI version skew)");
  if (c->width <= 0 || c->height <= 0) return fail(SALP_ERR_INVALID, "width/height must be positive");
  if (c->num_food_items < 0 || c->num_food_items > SALP_MAX_FOOD)
    return fail(SALP_ERR_INVALID, "num_food_items must be in [0, SALP_MAX_FOOD]");
  if (c->max_observed_food < 0 || c->max_observed_food > SALP_MAX_OBSERVED_FOOD)
    return fail(SALP_ERR_INVALID, "max_observed_food must be in [0, SALP_MAX_OBSERVED_FOOD]");
  if (c->inhale_duration < 1 || c->inhale_duration > 255 || c->exhale_duration < 4 || c->exhale_duration > 254)
    return fail(SALP_ERR_INVALID, "inhale_duration in [1,255], exhale_duration in [4,254] required");
  if (c->rest_duration < 0) return fail(SALP_ERR_INVALID, "rest_duration must be >= 0");
  if (!(c->base_radius > 0) || !(c->max_nozzle_angle > 0))
    return fail(SALP_ERR_INVALID, "base_radius and max_nozzle_angle must be positive");
  return SALP_OK;
}

typedef void (*rollout_fn)(DevParams, DevState, IOPtrs, int, int64_t, int64_t, const ColdBlock*);
typedef void (*reset_fn)(DevParams, DevState, const uint8_t*, float*, int);

// Output signatures (template parameter SIG): kSigMain = obs, reward, terminated, truncated and nothing else — every
// store of the step loop is unconditional, so the compiler can count the stores issued after the action prefetch and wait
// for the prefetch alone; kSigExtras = the same four plus final_obs and / or info (salp_vec_step, rollouts that keep the
// terminal observations): the four main streams stay unconditional, only the extras are tested (the terminal rows are
// written in the rare-event region, the three info words per step); kSigPartial = some main output is NULL: every
// store is tested, the step ends in a full drain (a 12-food rollout with final_obs ran 22 % slower in that form,
// profiles/r03/ab_notes.md session 15).  The one-wavefront predicated launches exist as kSigMain and kSigPartial only.
enum { kSigPartial = 0, kSigMain = 1, kSigExtras = 2 };
template <int FMAX, int KMAX, bool FORCED, bool STD, bool RAGGED, bool GEN>
rollout_fn pick_sig(int sig) {
  if constexpr (GEN) return (rollout_fn)salp_rollout_kernel<FMAX, KMAX, FORCED, STD, kSigMain, RAGGED, true>;
  else {
    if (sig == kSigMain) return (rollout_fn)salp_rollout_kernel<FMAX, KMAX, FORCED, STD, kSigMain, RAGGED, false>;
    if constexpr (!RAGGED)
      if (sig == kSigExtras) return (rollout_fn)salp_rollout_kernel<FMAX, KMAX, FORCED, STD, kSigExtras, false, false>;
    return (rollout_fn)salp_rollout_kernel<FMAX, KMAX, FORCED, STD, kSigPartial, RAGGED, false>;
  }
}
template <int FMAX, int KMAX, bool STD, bool RAGGED>
rollout_fn pick_rollout(bool forced, int sig, bool gen) {
  if (sig == kSigMain && gen)   // in-kernel action generation exists for the main-only output signature
    return forced ? pick_sig<FMAX, KMAX, true, STD, RAGGED, true>(sig) : pick_sig<FMAX, KMAX, false, STD, RAGGED, true>(sig);
  return forced ? pick_sig<FMAX, KMAX, true, STD, RAGGED, false>(sig) : pick_sig<FMAX, KMAX, false, STD, RAGGED, false>(sig);
}

// K = 3 (every preset): kernels by food-slot count, with the reference's constants as literals (STD) or — any other
// tank size, radius, drag, thrust, timing — with the constants read from the launch parameters.  K != 3 runs the
// generic instantiation (runtime constants, F <= 16, K <= 8, foods in LDS).
// True when in-kernel action generation is available for this handle and output signature.
bool can_generate_in_kernel(const salp_vec* h, bool full) { return full && h->kmax == 3; }

template <bool STD, bool RAGGED>
rollout_fn rollout_kernel_k3(const salp_vec* h, bool forced, int sig, bool gen) {
  if (h->fmax == 1) return pick_rollout<1, 3, STD, RAGGED>(forced, sig, gen);
  if (h->fmax == 4) return pick_rollout<4, 3, STD, RAGGED>(forced, sig, gen);
  if (h->fmax == 8) return pick_rollout<8, 3, STD, RAGGED>(forced, sig, gen);
  if (h->fmax == 12) return pick_rollout<12, 3, STD, RAGGED>(forced, sig, gen);
  return pick_rollout<16, 3, STD, RAGGED>(forced, sig, gen);
}
template <bool RAGGED>
rollout_fn rollout_kernel_for(const salp_vec* h, int sig, bool gen) {
  const bool forced = h->P.forced != 0;
  if (h->kmax == 3)
    return h->std_consts ? rollout_kernel_k3<true, RAGGED>(h, forced, sig, gen) : rollout_kernel_k3<false, RAGGED>(h, forced, sig, gen);
  if (h->fmax <= 12)   // K != 3 with up to 12 foods: the register-food form of the generic instantiation (2 wavefronts per SIMD, not 1)
    return forced ? (rollout_fn)salp_rollout_kernel<12, 8, true, false, kSigPartial, RAGGED, false>
                  : (rollout_fn)salp_rollout_kernel<12, 8, false, false, kSigPartial, RAGGED, false>;
  return forced ? (rollout_fn)salp_rollout_kernel<16, 8, true, false, kSigPartial, RAGGED, false>
                : (rollout_fn)salp_rollout_kernel<16, 8, false, false, kSigPartial, RAGGED, false>;
}
template <bool STD>
reset_fn reset_kernel_k3(const salp_vec* h) {
  if (h->fmax == 1) return (reset_fn)salp_reset_kernel<1, 3, STD>;
  if (h->fmax == 4) return (reset_fn)salp_reset_kernel<4, 3, STD>;
  if (h->fmax == 8) return (reset_fn)salp_reset_kernel<8, 3, STD>;
  if (h->fmax == 12) return (reset_fn)salp_reset_kernel<12, 3, STD>;
  return (reset_fn)salp_reset_kernel<16, 3, STD>;
}
reset_fn reset_kernel_for(const salp_vec* h) {
  if (h->kmax == 3) return h->std_consts ? reset_kernel_k3<true>(h) : reset_kernel_k3<false>(h);
  return (reset_fn)salp_reset_kernel<16, 8, false>;
}

// True when every constant of DevParams equals its StdConsts literal (the reference's defaults).
bool is_std(const DevParams& P) {
  typedef StdConsts C;
  return P.W == C::W && P.H == C::H && P.half_W == C::half_W && P.half_H == C::half_H && P.margin == C::margin &&
         P.wall_hi_x == C::wall_hi_x && P.wall_hi_y == C::wall_hi_y && P.R == C::R && P.a_rest == C::a_rest &&
         P.b_rest == C::b_rest && P.ab_full == C::ab_full && P.da_inh == C::da_inh && P.db_inh == C::db_inh &&
         P.da_exh == C::da_exh && P.db_exh == C::db_exh && P.max_nozzle == C::max_nozzle &&
         P.nozzle_rate == C::nozzle_rate && P.thrust_force == C::thrust_force && P.drag == C::drag &&
         P.ang_drag == C::ang_drag && P.exhale_dur_d == C::exhale_dur_d && P.food_radius == C::food_radius &&
         P.min_food_dist2 == C::min_food_dist2 && P.food_xlo == C::food_xlo && P.food_xspan == C::food_xspan &&
         P.food_ylo == C::food_ylo && P.food_yspan == C::food_yspan && P.inv_W == C::inv_W && P.inv_H == C::inv_H &&
         P.inv_pi == C::inv_pi && P.inv_R == C::inv_R && P.inv_max_nozzle == C::inv_max_nozzle &&
         P.inv_diag == C::inv_diag && P.tie_c0 == C::tie_c0 && P.inhale_dur == C::inhale_dur && P.exhale_dur == C::exhale_dur &&
         P.cycle_len == C::cycle_len;
}

int ensure_stage(salp_vec* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return SALP_OK;
  if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
  HIP_TRY(hipMalloc(&h->stage, bytes));
  h->stage_bytes = bytes;
  return SALP_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Bump {  // carve sub-buffers out of the staging allocation
  char* base; size_t off;
  template <class T> T* take(size_t count) {
    off = align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

int launch_rollout(salp_vec* h, const IOPtrs& io, int H, hipStream_t st) {
  const bool main_outputs = io.obs && io.reward && io.terminated && io.truncated;
  const int sig = !main_outputs ? kSigPartial : ((io.final_obs || io.info) ? kSigExtras : kSigMain);
  const bool gen = io.act == nullptr;               // only reached when can_generate_in_kernel()
  // envs in full wavefronts: unpredicated kernel
  int64_t n_full = h->n / kWave * kWave;
  // A small ragged batch (step-per-launch acting loops) is launch-bound: one predicated launch over the whole
  // range instead of two; the predicates only cost when the write stream is the bound.
  if (n_full < h->n && h->n * (int64_t)H <= (int64_t)1 << 22) n_full = 0;
  h->last_launch[0] = h->fmax; h->last_launch[1] = h->kmax; h->last_launch[2] = (h->kmax == 3) ? h->std_consts : 0;
  h->last_launch[3] = h->P.forced; h->last_launch[4] = (h->kmax == 3) ? sig : kSigPartial; h->last_launch[5] = gen;
  h->last_launch[6] = n_full; h->last_launch[7] = h->n - n_full;
  if (n_full > 0) {
    const unsigned grid = (unsigned)((n_full + kBlock - 1) / kBlock);
    const rollout_fn fn = rollout_kernel_for<false>(h, sig, gen);
    h->last_kernel = (const void*)fn;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, io, H,
                       (int64_t)0, n_full, (const ColdBlock*)h->cold);
    HIP_TRY(hipGetLastError());
  }
  if (n_full < h->n) {                              // the last n % 64 envs (or the whole small batch): predicated stores
    const unsigned rgrid = (unsigned)((h->n - n_full + kBlock - 1) / kBlock);
    const rollout_fn fn = rollout_kernel_for<true>(h, sig, gen);
    if (n_full == 0) h->last_kernel = (const void*)fn;
    hipLaunchKernelGGL(fn, dim3(rgrid), dim3(kBlock), 0, st, h->P, h->S, io, H,
                       n_full, h->n, (const ColdBlock*)h->cold);
    HIP_TRY(hipGetLastError());
  }
  return SALP_OK;
}

}  // namespace

extern "C" {

const char* salp_last_error(void) { return g_err.c_str(); }
int salp_abi_version(void) { return SALP_ABI_VERSION; }

int salp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int salp_config_default(salp_config_t* c) {
  if (!c) return fail(SALP_ERR_INVALID, "cfg is NULL");
  memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(*c);
  c->width = 800; c->height = 600; c->num_food_items = 5; c->max_observed_food = 3;
  c->max_steps_without_food = 1500; c->forced_breathing = 1; c->random_food_count = 0; c->respawn_food = 1;
  c->food_reward = 10.0; c->collision_penalty = -50.0; c->time_penalty = -0.1; c->efficiency_bonus = 1.0;
  c->proximity_reward_weight = 0.0;
  c->tank_margin = 50.0; c->base_radius = 30.0; c->max_thrust_force = 100.0; c->drag_coefficient = 0.98;
  c->angular_drag = 0.95; c->max_nozzle_angle = 3.141592653589793 / 3; c->nozzle_response_rate = 0.05;
  c->food_radius = 15.0; c->min_food_distance = 80.0;
  c->inhale_duration = 120; c->exhale_duration = 150; c->rest_duration = 60;
  return SALP_OK;
}

int salp_vec_create(const salp_config_t* cfg, int64_t n_envs, int device_id, uint64_t seed,
                    int64_t env_index_base, salp_vec_t** out) {
  if (!out) return fail(SALP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = validate(cfg);
  if (rc != SALP_OK) return rc;
  if (n_envs <= 0 || n_envs > ((int64_t)1 << 31) - kBlock) return fail(SALP_ERR_INVALID, "n_envs out of range");
  if (env_index_base < 0) return fail(SALP_ERR_INVALID, "env_index_base must be >= 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SALP_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(SALP_ERR_NO_DEVICE, "device_id out of range");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(device_id));

  salp_vec* h = new (std::nothrow) salp_vec();
  if (!h) return fail(SALP_ERR_OOM, "host allocation failed");
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg; h->device = device_id; h->n = n_envs; h->seed = seed; h->global_step = 0;
  h->F = cfg->num_food_items; h->K = cfg->max_observed_food;
  h->obs_dim = 10 + 4 * h->K + 2; h->act_dim = cfg->forced_breathing ? 1 : 2;
  h->kmax = (h->K == 3) ? 3 : 8;
  // food slots of the kernel instantiation: 1 (single_food*.yaml), 4, 8 (the class default of 5 foods), 12 (sac_gail.yaml), 16
  h->fmax = (h->kmax == 3) ? (h->F <= 1 ? 1 : (h->F <= 4 ? 4 : (h->F <= 8 ? 8 : (h->F <= 12 ? 12 : 16)))) : (h->F <= 12 ? 12 : 16);
  const int64_t pitch = (int64_t)align_up((size_t)n_envs, 64);
  h->P = make_params(*cfg, n_envs, pitch, seed, env_index_base);
  h->std_consts = is_std(h->P) ? 1 : 0;
  h->nf_rows = (size_t)(SF_FOOD0 + 2 * h->F);
  h->stats_enabled = 1;

  hipError_t e1 = hipMalloc((void**)&h->S.f, h->nf_rows * (size_t)pitch * sizeof(double));
  hipError_t e2 = (e1 == hipSuccess) ? hipMalloc((void**)&h->S.i, (size_t)SI_COUNT * (size_t)pitch * sizeof(int32_t)) : e1;
  hipError_t e3 = (e2 == hipSuccess) ? hipMalloc((void**)&h->stats, SALP_STATS_REPLICAS * sizeof(DevStats)) : e2;
  if (e3 != hipSuccess) {
    std::string m = std::string("hipMalloc(state): ") + hipGetErrorString(e3);
    salp_vec_destroy(h);
    return fail(e3 == hipErrorOutOfMemory ? SALP_ERR_OOM : SALP_ERR_HIP, m);
  }
  hipError_t e4 = hipMalloc((void**)&h->cold, sizeof(ColdBlock));
  if (e4 == hipSuccess) {
    h->P.seed = (seed_word_t*)(uintptr_t)h->cold->seed;   // device address; the kernels read the key words through it
    h->P.self = (dev_params_c*)(uintptr_t)&h->cold->P;    // likewise the constants of the STD = false kernels
    ColdBlock cb;
    cb.P = h->P; cb.S = h->S;
    cb.seed[0] = (uint32_t)seed; cb.seed[1] = (uint32_t)(seed >> 32);
    e4 = hipMemcpy(h->cold, &cb, sizeof(cb), hipMemcpyHostToDevice);
  }
  if (e4 != hipSuccess) {
    std::string m = std::string("hipMalloc/hipMemcpy(cold block): ") + hipGetErrorString(e4);
    salp_vec_destroy(h);
    return fail(e4 == hipErrorOutOfMemory ? SALP_ERR_OOM : SALP_ERR_HIP, m);
  }
  hipError_t e = hipMemset(h->S.f, 0, h->nf_rows * (size_t)pitch * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->S.i, 0, (size_t)SI_COUNT * (size_t)pitch * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(h->stats, 0, SALP_STATS_REPLICAS * sizeof(DevStats));
  if (e != hipSuccess) {
    std::string m = std::string("hipMemset(state): ") + hipGetErrorString(e);
    salp_vec_destroy(h);
    return fail(SALP_ERR_HIP, m);
  }
  // initial reset of every env (consumes the first draws of each env's stream)
  rc = salp_vec_reset(h, nullptr, nullptr, SALP_DEVICE_PTRS, nullptr);
  if (rc == SALP_OK) {   // the handle's own work only (the null stream it was issued on), not a device-wide drain
    hipError_t es = hipStreamSynchronize(nullptr);
    if (es != hipSuccess) rc = fail(SALP_ERR_HIP, std::string("initial reset: ") + hipGetErrorString(es));
  }
  if (rc != SALP_OK) { std::string m = g_err; salp_vec_destroy(h); g_err = m; return rc; }
  *out = h;
  return SALP_OK;
}

void salp_vec_destroy(salp_vec_t* h) {
  if (!h) return;
  DeviceScope dev_scope;
  (void)dev_scope.enter(h->device);
  (void)hipStreamSynchronize(h->last_stream);     // hipFree does not wait for kernels on non-blocking streams
  if (h->S.f) (void)hipFree(h->S.f);
  if (h->S.i) (void)hipFree(h->S.i);
  if (h->stats) (void)hipFree(h->stats);
  if (h->cold) (void)hipFree(h->cold);
  if (h->stage) (void)hipFree(h->stage);
  if (h->act_buf) (void)hipFree(h->act_buf);
  delete h;
}

int64_t salp_vec_num_envs(const salp_vec_t* h) { return h ? h->n : 0; }
int salp_vec_obs_dim(const salp_vec_t* h) { return h ? h->obs_dim : 0; }
int salp_vec_act_dim(const salp_vec_t* h) { return h ? h->act_dim : 0; }
#ifdef SALP_EXP_STAMPS
// experiment build only: the per-wavefront phase cycle sums of the last rollout launches (16 words per wavefront)
int salp_exp_read_stamps(uint32_t* dst, int words) {
  const size_t n = sizeof(uint32_t) * (size_t)(words < kStampWaves * 16 ? words : kStampWaves * 16);
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(salp_stamp_out), n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
#ifdef SALP_EXP_COUNT
int salp_exp_read_counters(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(salp_exp_counter), 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
int salp_vec_num_food(const salp_vec_t* h) { return h ? h->F : 0; }
int salp_vec_device(const salp_vec_t* h) { return h ? h->device : -1; }
int64_t salp_vec_global_step(const salp_vec_t* h) { return h ? h->global_step : 0; }
int salp_vec_last_launch(const salp_vec_t* h, int64_t info[8]) {
  if (!h || !info) return fail(SALP_ERR_INVALID, "handle/info is NULL");
  memcpy(info, h->last_launch, sizeof(h->last_launch));
  return SALP_OK;
}

int salp_vec_last_kernel_resources(const salp_vec_t* h, int32_t info[4]) {
  if (!h || !info) return fail(SALP_ERR_INVALID, "handle/info is NULL");
  if (!h->last_kernel) return fail(SALP_ERR_INVALID, "no step / rollout call has been issued on this handle");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipFuncAttributes attr;
  HIP_TRY(hipFuncGetAttributes(&attr, h->last_kernel));
  int blocks = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, h->last_kernel, kBlock, 0));
  info[0] = attr.numRegs; info[1] = (int32_t)attr.sharedSizeBytes; info[2] = (int32_t)attr.localSizeBytes; info[3] = blocks;
  return SALP_OK;
}

int salp_vec_set_base_num_food(salp_vec_t* h, int32_t k) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  if (k < 0 || k > h->P.F) return fail(SALP_ERR_INVALID, "base_num_food_items must be within 0..num_food_items of the handle");
  h->P.F_base = k;     // kernel parameters are passed by value at every launch
  return SALP_OK;
}
int32_t salp_vec_base_num_food(const salp_vec_t* h) { return h ? h->P.F_base : 0; }

int salp_vec_reset(salp_vec_t* h, const uint8_t* mask, float* obs, uint32_t flags, void* stream) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const unsigned grid = (unsigned)((h->n + kBlock - 1) / kBlock);
  reset_fn fn = reset_kernel_for(h);
  if (flags & SALP_DEVICE_PTRS) {
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, mask, obs, 1);
    HIP_TRY(hipGetLastError());
    return SALP_OK;
  }
  const size_t obs_b = (size_t)h->n * h->obs_dim * sizeof(float);
  int rc = ensure_stage(h, align_up(obs_b, 256) + align_up((size_t)h->n, 256) + 512);
  if (rc != SALP_OK) return rc;
  Bump b{(char*)h->stage, 0};
  float* d_obs = obs ? b.take<float>((size_t)h->n * h->obs_dim) : nullptr;
  uint8_t* d_mask = mask ? b.take<uint8_t>((size_t)h->n) : nullptr;
  if (mask) HIP_TRY(hipMemcpyAsync(d_mask, mask, (size_t)h->n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, (const uint8_t*)d_mask, d_obs, 1);
  HIP_TRY(hipGetLastError());
  if (obs) HIP_TRY(hipMemcpyAsync(obs, d_obs, obs_b, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SALP_OK;
}

int salp_vec_observe(salp_vec_t* h, float* obs, uint32_t flags, void* stream) {
  if (!h || !obs) return fail(SALP_ERR_INVALID, "handle/obs is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const unsigned grid = (unsigned)((h->n + kBlock - 1) / kBlock);
  reset_fn fn = reset_kernel_for(h);
  if (flags & SALP_DEVICE_PTRS) {
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, (const uint8_t*)nullptr, obs, 0);
    HIP_TRY(hipGetLastError());
    return SALP_OK;
  }
  const size_t obs_b = (size_t)h->n * h->obs_dim * sizeof(float);
  int rc = ensure_stage(h, align_up(obs_b, 256) + 512);
  if (rc != SALP_OK) return rc;
  float* d_obs = (float*)h->stage;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, (const uint8_t*)nullptr, d_obs, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(obs, d_obs, obs_b, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SALP_OK;
}

static int rollout_impl(salp_vec_t* h, const float* act, int32_t H, float* obs, float* reward,
                        uint8_t* terminated, uint8_t* truncated, float* final_obs, int32_t* info,
                        float* act_out, uint32_t flags, void* stream) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  if (H <= 0) return fail(SALP_ERR_INVALID, "horizon must be >= 1");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  IOPtrs io;
  memset(&io, 0, sizeof(io));
  io.stats = h->stats_enabled ? h->stats : nullptr;
  io.global_step = h->global_step;
  const size_t HN = (size_t)H * (size_t)h->n;
  const unsigned agrid = (unsigned)((h->n + kBlock - 1) / kBlock);
  if (flags & SALP_DEVICE_PTRS) {
    io.act = act; io.obs = obs; io.reward = reward; io.terminated = terminated; io.truncated = truncated;
    io.final_obs = final_obs; io.info = info; io.act_out = act_out;
    const bool full_sig = obs && reward && terminated && truncated && !final_obs && !info;
    if (!act && !can_generate_in_kernel(h, full_sig)) {  // generate into act_out when given, else into the handle's buffer
      float* dst = act_out;
      if (!dst) {
        const size_t need_a = HN * h->act_dim * sizeof(float);
        if (need_a > h->act_bytes) {
          if (h->act_buf) { (void)hipFree(h->act_buf); h->act_buf = nullptr; h->act_bytes = 0; }
          HIP_TRY(hipMalloc((void**)&h->act_buf, need_a));
          h->act_bytes = need_a;
        }
        dst = h->act_buf;
      }
      hipLaunchKernelGGL(salp_gen_actions_kernel, dim3(agrid), dim3(kBlock), 0, st, h->P, dst, (int)H, h->act_dim, (int64_t)h->global_step);
      HIP_TRY(hipGetLastError());
      io.act = dst;
    }
    int rc = launch_rollout(h, io, H, st);
    if (rc == SALP_OK) h->global_step += H;
    return rc;
  }
  // host pointers: stage through device memory, synchronous
  size_t need = 4096;
  need += align_up(HN * h->act_dim * sizeof(float), 256) * 2;
  need += align_up(HN * h->obs_dim * sizeof(float), 256) * (final_obs ? 2 : 1);
  need += align_up(HN * sizeof(float), 256) + 2 * align_up(HN, 256) + align_up(HN * SALP_INFO_COLS * sizeof(int32_t), 256);
  int rc = ensure_stage(h, need);
  if (rc != SALP_OK) return rc;
  Bump b{(char*)h->stage, 0};
  float* d_act = b.take<float>(HN * h->act_dim);
  float* d_aout = (!act && act_out) ? d_act : nullptr;
  float* d_obs = obs ? b.take<float>(HN * h->obs_dim) : nullptr;
  float* d_fin = final_obs ? b.take<float>(HN * h->obs_dim) : nullptr;
  float* d_rew = reward ? b.take<float>(HN) : nullptr;
  uint8_t* d_term = terminated ? b.take<uint8_t>(HN) : nullptr;
  uint8_t* d_trunc = truncated ? b.take<uint8_t>(HN) : nullptr;
  int32_t* d_info = info ? b.take<int32_t>(HN * SALP_INFO_COLS) : nullptr;
  if (act) HIP_TRY(hipMemcpyAsync(d_act, act, HN * h->act_dim * sizeof(float), hipMemcpyHostToDevice, st));
  if (final_obs) HIP_TRY(hipMemcpyAsync(d_fin, final_obs, HN * h->obs_dim * sizeof(float), hipMemcpyHostToDevice, st));
  if (!act) {
    hipLaunchKernelGGL(salp_gen_actions_kernel, dim3(agrid), dim3(kBlock), 0, st, h->P, d_act, (int)H, h->act_dim, (int64_t)h->global_step);
    HIP_TRY(hipGetLastError());
  }
  io.act = d_act; io.obs = d_obs; io.reward = d_rew; io.terminated = d_term; io.truncated = d_trunc;
  io.final_obs = d_fin; io.info = d_info; io.act_out = d_aout;
  rc = launch_rollout(h, io, H, st);
  if (rc != SALP_OK) return rc;
  h->global_step += H;
  if (obs) HIP_TRY(hipMemcpyAsync(obs, d_obs, HN * h->obs_dim * sizeof(float), hipMemcpyDeviceToHost, st));
  if (final_obs) HIP_TRY(hipMemcpyAsync(final_obs, d_fin, HN * h->obs_dim * sizeof(float), hipMemcpyDeviceToHost, st));
  if (reward) HIP_TRY(hipMemcpyAsync(reward, d_rew, HN * sizeof(float), hipMemcpyDeviceToHost, st));
  if (terminated) HIP_TRY(hipMemcpyAsync(terminated, d_term, HN, hipMemcpyDeviceToHost, st));
  if (truncated) HIP_TRY(hipMemcpyAsync(truncated, d_trunc, HN, hipMemcpyDeviceToHost, st));
  if (info) HIP_TRY(hipMemcpyAsync(info, d_info, HN * SALP_INFO_COLS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (d_aout) HIP_TRY(hipMemcpyAsync(act_out, d_aout, HN * h->act_dim * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SALP_OK;
}

int salp_vec_step(salp_vec_t* h, const float* act, float* obs, float* reward, uint8_t* terminated,
                  uint8_t* truncated, float* final_obs, int32_t* info, uint32_t flags, void* stream) {
  if (!act) return fail(SALP_ERR_INVALID, "act is NULL");
  return rollout_impl(h, act, 1, obs, reward, terminated, truncated, final_obs, info, nullptr, flags, stream);
}

int salp_vec_rollout(salp_vec_t* h, const float* act, int32_t horizon, float* obs, float* reward,
                     uint8_t* terminated, uint8_t* truncated, float* final_obs, float* act_out,
                     uint32_t flags, void* stream) {
  return rollout_impl(h, act, horizon, obs, reward, terminated, truncated, final_obs, nullptr, act_out, flags, stream);
}

int salp_vec_get_state(salp_vec_t* h, double* f64, int32_t* i32, uint32_t flags, void* stream) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const unsigned grid = (unsigned)((h->n + kBlock - 1) / kBlock);
  const size_t fb = (size_t)SALP_F_COUNT(h->F) * h->n * sizeof(double);
  const size_t ib = (size_t)SALP_I_COUNT * h->n * sizeof(int32_t);
  if (flags & SALP_DEVICE_PTRS) {
    hipLaunchKernelGGL(salp_get_state_kernel, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, f64, i32);
    HIP_TRY(hipGetLastError());
    return SALP_OK;
  }
  int rc = ensure_stage(h, align_up(fb, 256) + align_up(ib, 256) + 512);
  if (rc != SALP_OK) return rc;
  Bump b{(char*)h->stage, 0};
  double* d_f = f64 ? b.take<double>(fb / sizeof(double)) : nullptr;
  int32_t* d_i = i32 ? b.take<int32_t>(ib / sizeof(int32_t)) : nullptr;
  hipLaunchKernelGGL(salp_get_state_kernel, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, d_f, d_i);
  HIP_TRY(hipGetLastError());
  if (f64) HIP_TRY(hipMemcpyAsync(f64, d_f, fb, hipMemcpyDeviceToHost, st));
  if (i32) HIP_TRY(hipMemcpyAsync(i32, d_i, ib, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return SALP_OK;
}

int salp_vec_set_state(salp_vec_t* h, const double* f64, const int32_t* i32, uint32_t flags, void* stream) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const unsigned grid = (unsigned)((h->n + kBlock - 1) / kBlock);
  const size_t fb = (size_t)SALP_F_COUNT(h->F) * h->n * sizeof(double);
  const size_t ib = (size_t)SALP_I_COUNT * h->n * sizeof(int32_t);
  if (flags & SALP_DEVICE_PTRS) {
    hipLaunchKernelGGL(salp_set_state_kernel, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, f64, i32);
    HIP_TRY(hipGetLastError());
    return SALP_OK;
  }
  int rc = ensure_stage(h, align_up(fb, 256) + align_up(ib, 256) + 512);
  if (rc != SALP_OK) return rc;
  Bump b{(char*)h->stage, 0};
  double* d_f = f64 ? b.take<double>(fb / sizeof(double)) : nullptr;
  int32_t* d_i = i32 ? b.take<int32_t>(ib / sizeof(int32_t)) : nullptr;
  if (f64) HIP_TRY(hipMemcpyAsync(d_f, f64, fb, hipMemcpyHostToDevice, st));
  if (i32) HIP_TRY(hipMemcpyAsync(d_i, i32, ib, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(salp_set_state_kernel, dim3(grid), dim3(kBlock), 0, st, h->P, h->S, (const double*)d_f, (const int32_t*)d_i);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return SALP_OK;
}

int salp_vec_get_stats(salp_vec_t* h, salp_stats_t* out) {
  if (!h || !out) return fail(SALP_ERR_INVALID, "handle/out is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  // The totals are complete once the handle's most recent launch is: wait for ITS stream (in-order), not for the
  // device — a device-wide drain here would also stall every other stream of the process (RCCL collectives in flight).
  DevStats host[SALP_STATS_REPLICAS];
  HIP_TRY(hipMemcpyAsync(host, h->stats, sizeof(host), hipMemcpyDeviceToHost, h->last_stream));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  long long acc[16] = {0};
  for (int r = 0; r < SALP_STATS_REPLICAS; ++r)
    for (int k = 0; k < 16; ++k) acc[k] += (long long)host[r].v[k];
  out->env_steps = acc[ST_STEPS]; out->episodes = acc[ST_EPISODES]; out->terminated = acc[ST_TERM];
  out->truncated = acc[ST_TRUNC]; out->collisions = acc[ST_COLL]; out->food_collected = acc[ST_FOOD];
  out->episode_length_sum = acc[ST_EPLEN];
  out->reward_sum = (double)acc[ST_REWARD] / SALP_FIXED_SCALE;
  out->episode_return_sum = (double)acc[ST_EPRET] / SALP_FIXED_SCALE;
  return SALP_OK;
}

int salp_vec_clear_stats(salp_vec_t* h) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  // stream-ordered behind the handle's most recent launch: no host synchronisation at all
  HIP_TRY(hipMemsetAsync(h->stats, 0, SALP_STATS_REPLICAS * sizeof(DevStats), h->last_stream));
  return SALP_OK;
}

int salp_vec_reseed(salp_vec_t* h, uint64_t seed, float* obs, uint32_t flags, void* stream) {
  if (!h) return fail(SALP_ERR_INVALID, "handle is NULL");
  DeviceScope dev_scope;
  HIP_TRY(dev_scope.enter(h->device));
  hipStream_t st = (hipStream_t)stream;
  h->last_stream = st;
  const int64_t pitch = h->P.pitch;
  const unsigned grid = (unsigned)((pitch + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(salp_reseed_kernel, dim3(grid), dim3(kBlock), 0, st, h->cold, h->stats, pitch, (int)h->nf_rows,
                     (uint32_t)seed, (uint32_t)(seed >> 32));
  HIP_TRY(hipGetLastError());
  h->seed = seed;
  h->global_step = 0;
  return salp_vec_reset(h, nullptr, obs, flags, stream);   // every env, from draw counter 0 of the new streams
}

}  // extern "C"
