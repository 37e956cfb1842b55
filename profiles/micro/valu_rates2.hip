// Round 3 additions to valu_rates.hip: the instructions a packed-fp32 / integer-key ordering pass would be made of.
// Same method (8 independent chains of ONE instruction, median / last wavefront's shader cycles per instruction at
// W = 1..4 wavefronts per SIMD).  build: hipcc --offload-arch=gfx950 -O3 -o valu_rates2 valu_rates2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kIters = 4096;
#define CHAIN8(ASM, a) \
  asm volatile(ASM : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[1]) : "v"(b), "v"(c)); \
  asm volatile(ASM : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[3]) : "v"(b), "v"(c)); \
  asm volatile(ASM : "+v"(a[4]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[5]) : "v"(b), "v"(c)); \
  asm volatile(ASM : "+v"(a[6]) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a[7]) : "v"(b), "v"(c));
#define DEFTEST(NAME, T, ASM) \
__global__ void __launch_bounds__(1024) NAME(unsigned long long* cyc, T* sink, T b0, T c0) { \
  T a[8]; for (int i = 0; i < 8; ++i) a[i] = (T)(threadIdx.x + i); \
  T b = b0 + (T)(threadIdx.x & 1), c = c0 + (T)(threadIdx.x & 3);   /* per-lane values: b and c sit in VGPRs */ \
  __syncthreads(); \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < kIters; ++it) { CHAIN8(ASM, a) } \
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  T s = a[0]; for (int i = 1; i < 8; ++i) s += a[i]; \
  if (s == (T)123456789) sink[0] = s; \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
}
DEFTEST(t_add_f32, float, "v_add_f32 %0, %0, %1")
DEFTEST(t_sub_f32, float, "v_sub_f32 %0, %0, %1")
DEFTEST(t_mul_f32, float, "v_mul_f32 %0, %0, %1")
DEFTEST(t_fmac_f32, float, "v_fmac_f32 %0, %1, %2")
DEFTEST(t_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
DEFTEST(t_min_f32, float, "v_min_f32 %0, %0, %1")
DEFTEST(t_max_f32, float, "v_max_f32 %0, %0, %1")
DEFTEST(t_min3_f32, float, "v_min3_f32 %0, %0, %1, %2")
DEFTEST(t_med3_f32, float, "v_med3_f32 %0, %0, %1, %2")
DEFTEST(t_rsq_f32, float, "v_rsq_f32 %0, %0")
DEFTEST(t_sqrt_f32, float, "v_sqrt_f32 %0, %0")
DEFTEST(t_min_u32, unsigned, "v_min_u32 %0, %0, %1")
DEFTEST(t_max_u32, unsigned, "v_max_u32 %0, %0, %1")
DEFTEST(t_min_i32, int, "v_min_i32 %0, %0, %1")
DEFTEST(t_min3_u32, unsigned, "v_min3_u32 %0, %0, %1, %2")
DEFTEST(t_med3_u32, unsigned, "v_med3_u32 %0, %0, %1, %2")
DEFTEST(t_max3_u32, unsigned, "v_max3_u32 %0, %0, %1, %2")
DEFTEST(t_and_b32, unsigned, "v_and_b32 %0, %0, %1")
DEFTEST(t_or_b32, unsigned, "v_or_b32 %0, %0, %1")
DEFTEST(t_add_u32, unsigned, "v_add_u32 %0, %0, %1")
DEFTEST(t_sub_u32, unsigned, "v_sub_u32 %0, %0, %1")
DEFTEST(t_lshlrev_b32, unsigned, "v_lshlrev_b32 %0, 3, %0")
DEFTEST(t_lshl_or_b32, unsigned, "v_lshl_or_b32 %0, %0, 4, %1")
DEFTEST(t_and_or_b32, unsigned, "v_and_or_b32 %0, %0, %1, %2")
DEFTEST(t_bfi_b32, unsigned, "v_bfi_b32 %0, %1, %2, %0")
DEFTEST(t_perm_b32, unsigned, "v_perm_b32 %0, %0, %1, %2")
DEFTEST(t_add3_u32, unsigned, "v_add3_u32 %0, %0, %1, %2")
DEFTEST(t_pk_add_f32, double, "v_pk_add_f32 %0, %0, %1")
DEFTEST(t_pk_mul_f32, double, "v_pk_mul_f32 %0, %0, %1")
DEFTEST(t_pk_fma_f32, double, "v_pk_fma_f32 %0, %0, %1, %2")
DEFTEST(t_pk_mov_b32, double, "v_pk_mov_b32 %0, %1, %2")
DEFTEST(t_min_f64, double, "v_min_f64 %0, %0, %1")
DEFTEST(t_add_f64, double, "v_add_f64 %0, %0, %1")
DEFTEST(t_cvt_f32_u32, float, "v_cvt_f32_u32 %0, %0")
DEFTEST(t_cvt_u32_f32, float, "v_cvt_u32_f32 %0, %0")
DEFTEST(t_pk_max_u16, unsigned, "v_pk_max_u16 %0, %0, %1")
DEFTEST(t_add_f32_lit, float, "v_add_f32 %0, 0x3f99999a, %0")
DEFTEST(t_mul_f32_lit, float, "v_mul_f32 %0, 0x3f99999a, %0")
DEFTEST(t_fmaak_f32, float, "v_fmaak_f32 %0, %0, %1, 0x3f99999a")
DEFTEST(t_cmp_lt_f32_cnd, float, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
DEFTEST(t_cmp_lt_u32, unsigned, "v_cmp_lt_u32 vcc, %0, %1")
DEFTEST(t_max_f32_self, float, "v_max_f32 %0, %0, %0")

// a realistic mix: the fp32 ordering pass of one slot written as instructions (sub, sub, mul, fma, sqrt, add, and_or,
// 5 x min/max u32) against its fp64 form (add, add, mul, fma, cvt, sqrt, add_f32, and_or, 5 x min/max f64)
__global__ void __launch_bounds__(1024) t_slot_f32(unsigned long long* cyc, float* sink, float b0, float c0) {
  float fx[4], fy[4]; for (int i = 0; i < 4; ++i) { fx[i] = b0 * (threadIdx.x + i); fy[i] = c0 * (threadIdx.x + 2 * i); }
  float x = b0 + threadIdx.x, y = c0 - threadIdx.x, dsum = 0.f;
  unsigned k0 = ~0u, k1 = ~0u, k2 = ~0u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float dx, dy, d2, sq; unsigned cv, lo;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(fx[i]), "v"(x));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(fy[i]), "v"(y));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(d2) : "v"(dx));
      asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(d2) : "v"(dy));
      asm volatile("v_sqrt_f32 %0, %1" : "=v"(sq) : "v"(d2));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(dsum) : "v"(sq));
      asm volatile("v_and_or_b32 %0, %1, -16, %2" : "=v"(cv) : "v"(d2), "n"(5));
      asm volatile("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k0)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(cv) : "v"(k0)); k0 = lo;
      asm volatile("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k1)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(cv) : "v"(k1)); k1 = lo;
      asm volatile("v_min_u32 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k2)); k2 = lo;
      x += 0.f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (dsum + k0 + k1 + k2 == 1234.5f) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) / 4 * 8;   // per SLOT x 8 (run() divides by 8)
}
__global__ void __launch_bounds__(1024) t_slot_f64(unsigned long long* cyc, double* sink, double b0, double c0) {
  double fx[4], fy[4]; for (int i = 0; i < 4; ++i) { fx[i] = b0 * (threadIdx.x + i); fy[i] = c0 * (threadIdx.x + 2 * i); }
  double x = b0 + threadIdx.x, y = c0 - threadIdx.x; float dsum = 0.f;
  double k0 = 1e300, k1 = 1e300, k2 = 1e300;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double dx, dy, d2, cv, lo; float f, sq;
      asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dx) : "v"(fx[i]), "v"(x));
      asm volatile("v_add_f64 %0, %1, -%2" : "=v"(dy) : "v"(fy[i]), "v"(y));
      asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d2) : "v"(dx));
      asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d2) : "v"(dy));
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d2));
      asm volatile("v_sqrt_f32 %0, %1" : "=v"(sq) : "v"(f));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(dsum) : "v"(sq));
      { uint2 u = __builtin_bit_cast(uint2, d2); asm volatile("v_and_or_b32 %0, %0, -16, %1" : "+v"(u.x) : "n"(5)); cv = __builtin_bit_cast(double, u); }
      asm volatile("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k0)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(cv) : "v"(k0)); k0 = lo;
      asm volatile("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k1)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(cv) : "v"(k1)); k1 = lo;
      asm volatile("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(cv), "v"(k2)); k2 = lo;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (dsum + k0 + k1 + k2 == 1234.5) sink[0] = 1;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) / 4 * 8;
}

template <typename K, typename... A>
int run(const char* name, K kern, unsigned long long* dcyc, A... args) {
  printf("%-18s", name);
  for (int W : {1, 2, 3, 4}) {
    const int blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * W), 100 * 1024, 0, dcyc, args...);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * W), 100 * 1024, 0, dcyc, args...);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4 * W);
    CHK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2] / (8.0 * kIters);
    const double mx = (double)h.back() / (8.0 * kIters);
    printf("  W=%d: med %6.2f max %6.2f (max/W %5.2f)", W, med, mx, mx / W);
  }
  printf("\n");
  return 0;
}
int main() {
  unsigned long long* dcyc; void* sink;
  CHK(hipMalloc(&dcyc, 8192 * 8 * 4)); CHK(hipMalloc(&sink, 64));
  printf("wavefront cycles per instruction (s_memtime), W wavefronts per SIMD; t_slot_*: cycles per food slot\n");
#define RF(NAME) run(#NAME, NAME, dcyc, (float*)sink, 1.0000001f, 0.9999999f)
#define RU(NAME) run(#NAME, NAME, dcyc, (unsigned*)sink, 0x9E3779B9u, 0x85EBCA6Bu)
#define RI(NAME) run(#NAME, NAME, dcyc, (int*)sink, (int)0x1E3779B9, (int)0x05EBCA6B)
#define RD(NAME) run(#NAME, NAME, dcyc, (double*)sink, 1.0000001, 0.9999999)
  RF(t_add_f32); RF(t_sub_f32); RF(t_mul_f32); RF(t_fmac_f32); RF(t_fma_f32); RF(t_min_f32); RF(t_max_f32); RF(t_min3_f32);
  RF(t_med3_f32); RF(t_rsq_f32); RF(t_sqrt_f32); RU(t_min_u32); RU(t_max_u32); RI(t_min_i32); RU(t_min3_u32); RU(t_med3_u32);
  RU(t_max3_u32); RU(t_and_b32); RU(t_or_b32); RU(t_add_u32); RU(t_sub_u32); RU(t_lshlrev_b32); RU(t_lshl_or_b32);
  RU(t_and_or_b32); RU(t_bfi_b32); RU(t_perm_b32); RU(t_add3_u32); RD(t_pk_add_f32); RD(t_pk_mul_f32); RD(t_pk_fma_f32);
  RD(t_pk_mov_b32); RD(t_min_f64); RD(t_add_f64); RF(t_cvt_f32_u32); RF(t_cvt_u32_f32); RU(t_pk_max_u16);
  RF(t_add_f32_lit); RF(t_mul_f32_lit); RF(t_fmaak_f32); RF(t_cmp_lt_f32_cnd); RU(t_cmp_lt_u32); RF(t_max_f32_self);
  RF(t_slot_f32); RD(t_slot_f64);
  return 0;
}
