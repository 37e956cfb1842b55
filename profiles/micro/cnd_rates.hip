// What does v_cndmask_b32 cost on gfx950, by where its lane mask comes from?  (valu_rates.hip found 19 cycles per
// instruction for back-to-back `v_cndmask_b32 ..., vcc` on a VCC nobody wrote, 5.4 for the _e64 form on an SGPR pair.)
// One wavefront per SIMD (256 blocks of 256), median wavefront cycles (s_memtime) per instruction of the group.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kIters = 2048;

#define KERNEL(NAME, PRE, BODY, NINSTR) \
__global__ void __launch_bounds__(256) NAME(unsigned long long* cyc, float* sink, float b, float c) { \
  float a0 = b + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
  float vb = b * 3 + threadIdx.x, vc = c + threadIdx.x; \
  __syncthreads(); \
  PRE \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < kIters; ++it) { \
    asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vb), "v"(vc) : "vcc", "s20", "s21", "s22", "s23"); \
  } \
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  float q = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
  if (q == 1234.5f) sink[0] = q; \
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = (t1 - t0); \
} \
constexpr int NAME##_n = NINSTR;

// 8 selects on a VCC never written in the loop
KERNEL(k_stale_vcc, ;, 
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// the same after one SALU write / one VALU compare write of VCC before the loop
KERNEL(k_stale_vcc_salu, asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
KERNEL(k_stale_vcc_valu, asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(vb), "v"(vc) : "vcc");,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// one compare, then 7 selects on it
KERNEL(k_cmp_7cnd, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// compare, 3 selects, compare, 3 selects
KERNEL(k_cmp_3cnd, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cmp_lt_f32 vcc, %4, %9\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// compare, 1 select (x4)
KERNEL(k_cmp_1cnd, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %9\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_cmp_lt_f32 vcc, %4, %9\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %9\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// compare into an SGPR pair (_e64), then 7 selects on the pair
KERNEL(k_cmp64_7cnd64, ;,
  "v_cmp_lt_f32_e64 s[20:21], %0, %9\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
  "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]", 8)
// 8 selects on an SGPR pair nobody writes in the loop
KERNEL(k_stale_sgpr, asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");,
  "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
  "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]", 8)
// _e64 encoding with VCC named as the mask
KERNEL(k_stale_vcc_e64, ;,
  "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
  "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc", 8)
// the 64-bit select as the compiler writes it: compare f64, two selects
KERNEL(k_cmp_2cnd, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %9\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)
// independent work between the compare and its select
KERNEL(k_cmp_gap_cnd, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
  "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// baseline: 8 adds
KERNEL(k_add8, ;,
  "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
  "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)


// ---- which arrangements of VOP2 selects are slow?
KERNEL(k_alt_cnd_add, ;,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_add_f32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_f32 %3, %3, %8\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_add_f32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_f32 %7, %7, %8", 8)
KERNEL(k_pairs_cnd_add, ;,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)
KERNEL(k_cnd2_add1_cnd2, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_f32 %3, %3, %8\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)
KERNEL(k_cnd3_add5, ;,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_f32 %3, %3, %8\n"
  "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)
KERNEL(k_cnd4_add4, ;,
  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
  "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8", 8)
// selects between DIFFERENT source registers (src1 differs per instruction)
KERNEL(k_cnd8_diffsrc, ;,
  "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
  "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// f64 compare + the two selects of a 64-bit value, twice, with f64 work between (the kernels' common shape)
KERNEL(k_f64_select_shape, ;,
  "v_cmp_lt_f32 vcc, %0, %9\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_f32 vcc, %3, %9\n"
  "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc", 8)
// other VOP2 instructions with an implicit VCC read
KERNEL(k_addc8, ;,
  "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
  "v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc", 8)

template <typename K>
int run(const char* name, K kern, int n, unsigned long long* dcyc, float* sink) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, dcyc, sink, 1.5f, 2.5f);
  CHK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(1024);
  CHK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-20s %6.2f cycles per instruction (group of %d: %6.1f)\n", name, (double)h[512] / ((double)n * kIters), n, (double)h[512] / kIters);
  return 0;
}
int main() {
  unsigned long long* dcyc; float* sink;
  CHK(hipMalloc(&dcyc, 1024 * 8)); CHK(hipMalloc(&sink, 64));
#define R(NAME) run(#NAME, NAME, NAME##_n, dcyc, sink)
  R(k_add8); R(k_stale_vcc); R(k_stale_vcc_salu); R(k_stale_vcc_valu); R(k_stale_vcc_e64); R(k_stale_sgpr);
  R(k_cmp_7cnd); R(k_cmp_3cnd); R(k_cmp_2cnd); R(k_cmp_1cnd); R(k_cmp_gap_cnd); R(k_cmp64_7cnd64);
  R(k_alt_cnd_add); R(k_pairs_cnd_add); R(k_cnd2_add1_cnd2); R(k_cnd3_add5); R(k_cnd4_add4); R(k_cnd8_diffsrc); R(k_f64_select_shape); R(k_addc8);
  return 0;
}
