// Issue / throughput cost of the VALU instructions the SALP kernels are made of, on gfx950, at 1..4 wavefronts per
// SIMD.  Each test runs 8 independent chains of ONE instruction (inline asm, so the compiler neither removes nor
// rewrites it) 8 x 512 times per wavefront and reports the median wavefront's shader cycles (s_memtime) per
// instruction.  With W wavefronts per SIMD, cycles-per-instruction / W approaches the pipe's throughput cost once
// the pipe (not the per-wavefront issue interval) is the limit.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;

#define CHAIN8(ASM, CONS, a) \
  asm volatile(ASM : CONS(a[0]) : "v"(b), "v"(c)); asm volatile(ASM : CONS(a[1]) : "v"(b), "v"(c)); \
  asm volatile(ASM : CONS(a[2]) : "v"(b), "v"(c)); asm volatile(ASM : CONS(a[3]) : "v"(b), "v"(c)); \
  asm volatile(ASM : CONS(a[4]) : "v"(b), "v"(c)); asm volatile(ASM : CONS(a[5]) : "v"(b), "v"(c)); \
  asm volatile(ASM : CONS(a[6]) : "v"(b), "v"(c)); asm volatile(ASM : CONS(a[7]) : "v"(b), "v"(c));

#define DEFTEST(NAME, T, ASM) \
__global__ void __launch_bounds__(1024) NAME(unsigned long long* cyc, T* sink, T b, T c) { \
  T a[8]; for (int i = 0; i < 8; ++i) a[i] = (T)(threadIdx.x + i); \
  __syncthreads(); \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < kIters; ++it) { CHAIN8(ASM, "+v", a) } \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  T s = a[0]; for (int i = 1; i < 8; ++i) s += a[i]; \
  if (s == (T)123456789) sink[0] = s; \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
}

DEFTEST(t_add_f64, double, "v_add_f64 %0, %0, %1")
DEFTEST(t_mul_f64, double, "v_mul_f64 %0, %0, %1")
DEFTEST(t_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
DEFTEST(t_min_f64, double, "v_min_f64 %0, %0, %1")
DEFTEST(t_max_f64, double, "v_max_f64 %0, %0, %1")
DEFTEST(t_rndne_f64, double, "v_rndne_f64 %0, %0")
DEFTEST(t_add_f32, float, "v_add_f32 %0, %0, %1")
DEFTEST(t_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
DEFTEST(t_sqrt_f32, float, "v_sqrt_f32 %0, %0")
DEFTEST(t_rcp_f32, float, "v_rcp_f32 %0, %0")
DEFTEST(t_max_f32, float, "v_max_f32 %0, %0, %1")
DEFTEST(t_med3_f32, float, "v_med3_f32 %0, %0, %1, %2")
DEFTEST(t_xor_b32, unsigned, "v_xor_b32 %0, %0, %1")
DEFTEST(t_and_or_b32, unsigned, "v_and_or_b32 %0, %0, %1, %2")
DEFTEST(t_mul_lo_u32, unsigned, "v_mul_lo_u32 %0, %0, %1")
DEFTEST(t_mul_hi_u32, unsigned, "v_mul_hi_u32 %0, %0, %1")
DEFTEST(t_cndmask, unsigned, "v_cndmask_b32 %0, %0, %1, vcc")
DEFTEST(t_mov_b32, unsigned, "v_mov_b32 %0, %1")
DEFTEST(t_lshl_add_u64, unsigned long long, "v_lshl_add_u64 %0, %0, 0, %1")
DEFTEST(t_mov_b64, unsigned long long, "v_mov_b64 %0, %1")

// shapes that do not fit the (T, T, T) pattern
__global__ void __launch_bounds__(1024) t_cvt_f32_f64(unsigned long long* cyc, float* sink, double b, double c) {
  double a[8]; float r[8]; for (int i = 0; i < 8; ++i) { a[i] = b + threadIdx.x + i; r[i] = 0.f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += r[i];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_cvt_f64_f32(unsigned long long* cyc, double* sink, float b, float c) {
  float a[8]; double r[8]; for (int i = 0; i < 8; ++i) { a[i] = b + threadIdx.x + i; r[i] = 0.; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < 8; ++i) s += r[i];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_mad_u64_u32(unsigned long long* cyc, unsigned long long* sink, unsigned b, unsigned c) {
  unsigned long long a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned x = b + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(c) : "vcc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123456789ull) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_cmp_f64(unsigned long long* cyc, unsigned long long* sink, double b, double c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a[0] == 12345.678) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_readlane(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned a = b + threadIdx.x; unsigned s[8];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[i]) : "v"(a));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += s[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_salu(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned s[8]; for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_readfirstlane(b + i);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(c) : "scc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += s[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// a VALU stream with an independent SALU instruction after every VALU one: does the scalar work share the
// wavefront's issue slots (cost adds) or overlap?
__global__ void __launch_bounds__(1024) t_valu_salu_mix(unsigned long long* cyc, double* sink, double b, double c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  unsigned s[8]; for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_readfirstlane((unsigned)i);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      asm volatile("s_add_u32 %0, %0, 1" : "+s"(s[i]) :: "scc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double q = 0; for (int i = 0; i < 8; ++i) q += a[i] + s[i];
  if (q == 12345.678) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// dependent chain: latency of v_add_f64 / v_min_f64 back to back on one register
__global__ void __launch_bounds__(1024) t_dep_add_f64(unsigned long long* cyc, double* sink, double b, double c) {
  double a = b + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (a == 12345.678) sink[0] = a;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}


// ---- operand-source variants: SGPR operands, VCC / SGPR-pair lane masks
__global__ void __launch_bounds__(1024) t_cndmask_sgpr(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned long long m = __builtin_amdgcn_readfirstlane(b) * 0x100000001ull;
  unsigned vb = b + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(vb), "s"(m));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_cndmask_indep(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = 0;
  unsigned vb = b + threadIdx.x, vc = c + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(vb), "v"(vc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_cmp_cnd_pair(unsigned long long* cyc, unsigned* sink, float b, float c) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  float vc = c + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // 4 (compare, select) pairs = 8 instructions
      asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(a[i + 1]), "v"(vc) : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 1234.5f) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_add_f64_sgpr(unsigned long long* cyc, double* sink, double b, double c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "s"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 12345.678) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_fma_f64_sgpr(unsigned long long* cyc, double* sink, double b, double c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  double vb = b + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(vb), "s"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 12345.678) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_add_f32_sgpr(unsigned long long* cyc, float* sink, float b, float c) {
  float a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 12345.678f) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_bfi_b32(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned vb = b + threadIdx.x, vm = c * threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(a[i]) : "v"(vb), "v"(vm));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_addc_u32(unsigned long long* cyc, unsigned* sink, unsigned b, unsigned c) {
  unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned vb = b + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(vb) : "vcc");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 123456789u) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void __launch_bounds__(1024) t_pk_fma_f32(unsigned long long* cyc, double* sink, double b, double c) {
  double a[8]; for (int i = 0; i < 8; ++i) a[i] = b + threadIdx.x + i;     // 64-bit containers of two floats
  double vb = b * threadIdx.x, vc = c;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(vb), "v"(vc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double q = 0; for (int i = 0; i < 8; ++i) q += a[i];
  if (q == 12345.678) sink[0] = q;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename K, typename... A>
int run(const char* name, K kern, unsigned long long* dcyc, A... args) {
  printf("%-16s", name);
  for (int W : {1, 2, 3, 4}) {
    // ONE block per CU (100 KB of dynamic LDS each) of 256 x W threads: W wavefronts on every SIMD, by construction
    const int blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * W), 100 * 1024, 0, dcyc, args...);   // warm-up
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * W), 100 * 1024, 0, dcyc, args...);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4 * W);
    CHK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (8.0 * kIters);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= (double)h.size() * 8.0 * kIters;
    const double mx = (double)h.back() / (8.0 * kIters);   // issue is arbitrated by age: old wavefronts run ahead, the LAST one to finish saw the whole job
    printf("  W=%d: med %5.2f max %5.2f (max/W %5.2f)", W, med, mx, mx / W);
  }
  printf("\n");
  return 0;
}

int main() {
  unsigned long long* dcyc; void* sink;
  CHK(hipMalloc(&dcyc, 8192 * 8 * 4)); CHK(hipMalloc(&sink, 64));
  printf("wavefront cycles per instruction (s_memtime), W wavefronts per SIMD (one block of 256 x W threads per CU)\n");
#define R3(NAME, T) run(#NAME, NAME, dcyc, (T*)sink, (T)1.0000001, (T)0.9999999)
  R3(t_add_f64, double); R3(t_mul_f64, double); R3(t_fma_f64, double); R3(t_min_f64, double); R3(t_max_f64, double);
  R3(t_rndne_f64, double); R3(t_add_f32, float); R3(t_fma_f32, float); R3(t_sqrt_f32, float); R3(t_rcp_f32, float);
  R3(t_max_f32, float); R3(t_med3_f32, float);
#define RU(NAME, T) run(#NAME, NAME, dcyc, (T*)sink, (T)0x9E3779B9u, (T)0x85EBCA6Bu)
  RU(t_xor_b32, unsigned); RU(t_and_or_b32, unsigned); RU(t_mul_lo_u32, unsigned); RU(t_mul_hi_u32, unsigned);
  RU(t_cndmask, unsigned); RU(t_mov_b32, unsigned); RU(t_lshl_add_u64, unsigned long long); RU(t_mov_b64, unsigned long long);
  run("t_cvt_f32_f64", t_cvt_f32_f64, dcyc, (float*)sink, 1.5, 2.5);
  run("t_cvt_f64_f32", t_cvt_f64_f32, dcyc, (double*)sink, 1.5f, 2.5f);
  run("t_mad_u64_u32", t_mad_u64_u32, dcyc, (unsigned long long*)sink, 0x9E3779B9u, 0x85EBCA6Bu);
  run("t_cmp_f64", t_cmp_f64, dcyc, (unsigned long long*)sink, 1.5, 2.5);
  run("t_readlane", t_readlane, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_salu", t_salu, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_valu_salu_mix", t_valu_salu_mix, dcyc, (double*)sink, 1.5, 2.5);
  run("t_dep_add_f64", t_dep_add_f64, dcyc, (double*)sink, 1.5, 2.5);
  run("t_cndmask_sgpr", t_cndmask_sgpr, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_cndmask_indep", t_cndmask_indep, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_cmp_cnd_pair", t_cmp_cnd_pair, dcyc, (unsigned*)sink, 1.5f, 2.5f);
  run("t_add_f64_sgpr", t_add_f64_sgpr, dcyc, (double*)sink, 1.5, 2.5);
  run("t_fma_f64_sgpr", t_fma_f64_sgpr, dcyc, (double*)sink, 1.5, 2.5);
  run("t_add_f32_sgpr", t_add_f32_sgpr, dcyc, (float*)sink, 1.5f, 2.5f);
  run("t_bfi_b32", t_bfi_b32, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_addc_u32", t_addc_u32, dcyc, (unsigned*)sink, 5u, 7u);
  run("t_pk_fma_f32", t_pk_fma_f32, dcyc, (double*)sink, 1.5, 2.5);
  return 0;
}
