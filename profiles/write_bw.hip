// write_bw.hip — pure-write microbenchmark on MI355X: how fast can 6.3 GB be written, and how much of that
// does the rollout kernel's write pattern (per wavefront and step: 6144 contiguous bytes of a [H][N][24] f32
// block, all wavefronts advancing through the steps together) give away?
// Build: hipcc -O3 --offload-arch=gfx950 profiles/write_bw.hip -o profiles/write_bw   Run: profiles/write_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (1) flat grid-stride: consecutive lanes -> consecutive 16 B, `blocks` x 256 threads sweep the buffer
template <bool NT>
__global__ __launch_bounds__(1024) void flat_kernel(v4f* out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
  }
}

// (2) the rollout pattern: wavefront w of the grid owns envs [64 w, 64 w + 64); at step t it writes its
// 6144-byte run at float4 offset (t * N + 64 w) * 6, as 6 instructions of 64 x 16 B.  DRAIN: wait for the
// step's stores before the next step.  SMALL: also the reward (4 B/lane) and two flag (1 B/lane) streams.
template <bool NT, bool DRAIN, bool SMALL>
__global__ __launch_bounds__(256) void slab_kernel(v4f* obs, float* rew, unsigned char* f0, unsigned char* f1, int H, size_t N) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t env = w * 64 + lane;
  const v4f v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll 1
  for (int t = 0; t < H; ++t) {
    v4f* p = obs + ((size_t)t * N + w * 64) * 6 + lane;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (NT) __builtin_nontemporal_store(v, &p[j * 64]); else p[j * 64] = v;
    }
    if (SMALL) {
      rew[(size_t)t * N + env] = 1.f;
      f0[(size_t)t * N + env] = 0;
      f1[(size_t)t * N + env] = 1;
    }
    if (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

// (2c) the rollout pattern with everything the real kernel moves: the per-step action read (4 B/lane, prefetched
// one step ahead, consumed so it cannot be dropped) and the small stores issued BEFORE the observation rows
template <bool READ, bool SMALL_FIRST>
__global__ __launch_bounds__(256) void full_kernel(v4f* obs, float* rew, unsigned char* f0, unsigned char* f1, const float* act, int H, size_t N) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t env = w * 64 + lane;
  float a = READ ? act[env] : 1.f;
  float acc = 0.f;
#pragma unroll 1
  for (int t = 0; t < H; ++t) {
    const float cur = a;
    if (READ) a = act[(size_t)(t + 1 < H ? t + 1 : t) * N + env];
    acc += cur;
    const v4f v = {cur, acc, 3.f, 4.f};
    if (SMALL_FIRST) { rew[(size_t)t * N + env] = acc; f0[(size_t)t * N + env] = 0; f1[(size_t)t * N + env] = 1; }
    v4f* p = obs + ((size_t)t * N + w * 64) * 6 + lane;
#pragma unroll
    for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(v, &p[j * 64]);
    if (!SMALL_FIRST) { rew[(size_t)t * N + env] = acc; f0[(size_t)t * N + env] = 0; f1[(size_t)t * N + env] = 1; }
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

// (2b) "fat wavefronts": each lane owns E envs, a wavefront 64 E consecutive envs: per step it writes E x 6144
// contiguous bytes (6 E instructions) and there are E times fewer wavefronts.  WPB wavefronts per block.
template <int E, int WPB, bool DRAIN, bool SMALL>
__global__ __launch_bounds__(64 * WPB) void fat_kernel(v4f* obs, float* rew, unsigned char* f0, unsigned char* f1, int H, size_t N) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll 1
  for (int t = 0; t < H; ++t) {
    v4f* p = obs + ((size_t)t * N + w * 64 * E) * 6 + lane;
#pragma unroll
    for (int j = 0; j < 6 * E; ++j) __builtin_nontemporal_store(v, &p[j * 64]);
    if (SMALL) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const size_t env = (w * E + e) * 64 + lane;
        rew[(size_t)t * N + env] = 1.f;
        f0[(size_t)t * N + env] = 0;
        f1[(size_t)t * N + env] = 1;
      }
    }
    if (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

// (3) the same bytes, block-major in time: a wavefront's H runs are contiguous (NOT the [H][N][24] layout;
// shows what the time-major slab order costs)
template <bool NT>
__global__ __launch_bounds__(256) void blockmajor_kernel(v4f* obs, int H) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const v4f v = {1.f, 2.f, 3.f, 4.f};
  v4f* p = obs + w * (size_t)H * 64 * 6 + lane;
#pragma unroll 1
  for (int t = 0; t < H; ++t) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (NT) __builtin_nontemporal_store(v, &p[((size_t)t * 6 + j) * 64]); else p[((size_t)t * 6 + j) * 64] = v;
    }
  }
}

template <class F>
static double time_ms(F launch, int reps = 9) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch(); launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float m; CHECK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main() {
  const size_t N = 262144; const int H = 250;
  const size_t obs_bytes = (size_t)H * N * 96, n4 = obs_bytes / 16;
  v4f* obs; float* rew; unsigned char *f0, *f1;
  CHECK(hipMalloc(&obs, obs_bytes)); CHECK(hipMalloc(&rew, (size_t)H * N * 4));
  CHECK(hipMalloc(&f0, (size_t)H * N)); CHECK(hipMalloc(&f1, (size_t)H * N));
  const double gb = obs_bytes / 1e9, gb_small = gb + (double)H * N * 6 / 1e9;
  auto rep = [&](const char* name, double ms, double g) { printf("%-58s %7.3f ms  %6.2f TB/s\n", name, ms, g / ms); fflush(stdout); };
  for (int threads : {256, 1024})
    for (int blocks : {256, 1024}) {
      char nm[96];
      snprintf(nm, sizeof nm, "flat grid-stride, %d x %d threads (window %.1f MB), plain", blocks, threads, blocks * (double)threads * 16 / 1e6);
      rep(nm, time_ms([&] { hipLaunchKernelGGL(flat_kernel<false>, dim3(blocks), dim3(threads), 0, 0, obs, n4); }, 5), gb);
    }
  const int G = (int)(N / 256);
  rep("rollout pattern [H][N][24], plain, no drain", time_ms([&] { hipLaunchKernelGGL((slab_kernel<false, false, false>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, H, N); }), gb);
  rep("rollout pattern, nt, no drain", time_ms([&] { hipLaunchKernelGGL((slab_kernel<true, false, false>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, H, N); }), gb);
  rep("rollout pattern, nt, drain per step", time_ms([&] { hipLaunchKernelGGL((slab_kernel<true, true, false>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, H, N); }), gb);
  rep("rollout pattern, nt, drain, + reward and flag streams", time_ms([&] { hipLaunchKernelGGL((slab_kernel<true, true, true>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, H, N); }), gb_small);
  rep("rollout pattern, nt, no drain, + reward and flag streams", time_ms([&] { hipLaunchKernelGGL((slab_kernel<true, false, true>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, H, N); }), gb_small);
  float* act; CHECK(hipMalloc(&act, (size_t)H * N * 4)); CHECK(hipMemset(act, 0, (size_t)H * N * 4));
  const double gb_all = gb_small + (double)H * N * 4 / 1e9;
  rep("full: obs + small stores after, no action read", time_ms([&] { hipLaunchKernelGGL((full_kernel<false, false>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, act, H, N); }), gb_small);
  rep("full: obs + small stores first, no action read", time_ms([&] { hipLaunchKernelGGL((full_kernel<false, true>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, act, H, N); }), gb_small);
  rep("full: obs + small stores after + action read", time_ms([&] { hipLaunchKernelGGL((full_kernel<true, false>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, act, H, N); }), gb_all);
  rep("full: obs + small stores first + action read (the kernel's stream)", time_ms([&] { hipLaunchKernelGGL((full_kernel<true, true>), dim3(G), dim3(256), 0, 0, obs, rew, f0, f1, act, H, N); }), gb_all);
#define FAT(E, WPB, DR, SM, label) rep(label, time_ms([&] { hipLaunchKernelGGL((fat_kernel<E, WPB, DR, SM>), dim3((unsigned)(N / (64 * E * WPB))), dim3(64 * WPB), 0, 0, obs, rew, f0, f1, H, N); }), SM ? gb_small : gb)
  FAT(1, 4, true, false, "fat E=1 (64 envs/wavefront), 4 waves/block, drain");
  FAT(2, 4, true, false, "fat E=2 (128 envs/wavefront), 4 waves/block, drain");
  FAT(4, 4, true, false, "fat E=4 (256 envs/wavefront), 4 waves/block, drain");
  FAT(4, 1, true, false, "fat E=4, 1 wave/block, drain");
  FAT(4, 4, false, false, "fat E=4, 4 waves/block, no drain");
  FAT(8, 4, true, false, "fat E=8 (512 envs/wavefront), 4 waves/block, drain");
  FAT(2, 4, true, true, "fat E=2 + reward / flag streams, drain");
  FAT(4, 4, true, true, "fat E=4 + reward / flag streams, drain");
  FAT(4, 4, false, true, "fat E=4 + reward / flag streams, no drain");
  rep("block-major in time (not the output layout), plain", time_ms([&] { hipLaunchKernelGGL(blockmajor_kernel<false>, dim3(G), dim3(256), 0, 0, obs, H); }), gb);
  rep("block-major in time (not the output layout), nt", time_ms([&] { hipLaunchKernelGGL(blockmajor_kernel<true>, dim3(G), dim3(256), 0, 0, obs, H); }), gb);
  double ms = time_ms([&] { CHECK(hipMemsetAsync(obs, 0, obs_bytes, 0)); });
  rep("hipMemsetAsync", ms, gb);
  return 0;
}
