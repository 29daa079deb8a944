// Issue cost of the VALU instruction classes the expansion kernels are made of, on gfx950: a wave executes 64 x 4 000
// independent copies of one instruction; 8 waves per SIMD, every SIMD busy.  Prints SIMD cycles per wave instruction
// at the clock the run reached (wall_clock64 is 100 MHz; the shader clock is measured with s_memtime).
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, body, nout)                                                                                   \
  __global__ __launch_bounds__(256) void name(unsigned long long *out, int iters) {                                \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = blockIdx.x | 1u;                           \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e = 1.0000001;                                                       \
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, g = 1.0001f;                                                          \
    unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                                           \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                     \
    for (int i = 0; i < iters; i++) { REP16(body) }                                                                  \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                     \
    if (a0 + a1 + a2 + a3 + d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 + q0 + q1 + q2 + q3 == 12345.678) out[1] = 1;       \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                       \
  }

// four independent chains per body: 64 instructions per loop trip
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_mul_u24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b) : "vcc");, 4)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));, 4)
KERNEL(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));, 4)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));, 4)
KERNEL(k_add_f64, asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));, 4)
KERNEL(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));, 4)
KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");, 4)
KERNEL(k_writelane, asm volatile("v_writelane_b32 %0, s2, 3\n v_writelane_b32 %1, s2, 5\n v_writelane_b32 %2, s2, 7\n v_writelane_b32 %3, s2, 9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_dpp_mov, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_cvt_i32_f64, asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %4\n v_cvt_i32_f64 %2, %4\n v_cvt_i32_f64 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e));, 4)
KERNEL(k_cmp_i32, asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cmp_gt_i32 vcc, %1, %4\n v_cmp_gt_i32 vcc, %2, %4\n v_cmp_gt_i32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)

KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s10", "s11");, 4)
KERNEL(k_cmp_cndmask, asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_gt_i32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_lshlrev_b32, asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_bfe_u32, asm volatile("v_bfe_u32 %0, %0, 3, 7\n v_bfe_u32 %1, %1, 3, 7\n v_bfe_u32 %2, %2, 3, 7\n v_bfe_u32 %3, %3, 3, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_min_i32, asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_add_f32, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));, 4)
KERNEL(k_mov_b64, asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(e));, 4)
KERNEL(k_add_u32_sgpr, asm volatile("v_add_u32 %0, s4, %0\n v_add_u32 %1, s4, %1\n v_add_u32 %2, s4, %2\n v_add_u32 %3, s4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_salu_mix, asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %4\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21", "scc");, 4)

KERNEL(k_cmp_2cnd, asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_u32 %2, %2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_cmp_3cnd, asm volatile("v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_cmp64_cnd, asm volatile("v_cmp_gt_i32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s10", "s11");, 4)
KERNEL(k_or_b32, asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_xor_b32, asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));, 4)
KERNEL(k_lshrrev_b32, asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_add_u32_imm, asm volatile("v_add_u32 %0, 7, %0\n v_add_u32 %1, 7, %1\n v_add_u32 %2, 7, %2\n v_add_u32 %3, 7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_add_u32_lit, asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_mov_sgpr, asm volatile("v_mov_b32 %0, s4\n v_mov_b32 %1, s4\n v_mov_b32 %2, s4\n v_mov_b32 %3, s4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));, 4)
KERNEL(k_add_co, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");, 4)
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));, 4)
KERNEL(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(e));, 4)

typedef void (*kern_t)(unsigned long long *, int);

int main() {
  unsigned long long *d, h[2];
  CK(hipMalloc((void **)&d, 16));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  struct { const char *name; kern_t k; } ks[] = {
      {"v_add_u32", k_add_u32}, {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_i32", k_cmp_i32}, {"v_add3_u32", k_add3},
      {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mad_u64_u32", k_mad_u64_u32},
      {"v_lshl_add_u64", k_lshl_add_u64}, {"v_fma_f32", k_fma_f32}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
      {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cndmask (sgpr mask)", k_cndmask_sgpr}, {"v_cmp + v_cndmask", k_cmp_cndmask}, {"cmp + 2 cndmask + add", k_cmp_2cnd}, {"cmp + 3 cndmask (vcc)", k_cmp_3cnd}, {"cmp + 3 cndmask (sgpr)", k_cmp64_cnd}, {"v_or_b32", k_or_b32}, {"v_xor_b32", k_xor_b32}, {"v_sub_u32", k_sub_u32}, {"v_lshrrev_b32", k_lshrrev_b32}, {"v_add_u32 (inline const)", k_add_u32_imm}, {"v_add_u32 (literal)", k_add_u32_lit}, {"v_mov_b32 (sgpr src)", k_mov_sgpr}, {"v_add_co + v_addc_co", k_add_co}, {"v_mul_f32", k_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_and_b32", k_and_b32}, {"v_lshlrev_b32", k_lshlrev_b32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_bfe_u32", k_bfe_u32}, {"v_min_i32", k_min_i32}, {"v_add_f32", k_add_f32}, {"v_mov_b64", k_mov_b64}, {"v_add_u32 (sgpr src)", k_add_u32_sgpr}, {"v_add_u32 + s_add_u32", k_salu_mix}, {"v_readlane_b32", k_readlane}, {"v_writelane_b32", k_writelane}, {"v_mov_b32_dpp", k_dpp_mov}};
  const int iters = 4000;
  for (auto &e : ks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    // wave 0 of workgroup 0 ran `iters * 64` instructions while 7 other waves shared its SIMD
    const double cyc = (double)h[0] / ((double)iters * 64.0 * 8.0);
    const double ns = (double)ms * 1e6 / ((double)iters * 64.0 * 8.0);  // per SIMD: 8 waves x iters x 64 instructions back to back
    printf("%-24s %6.3f ns per wave instruction per SIMD = %5.2f cycles at 2.4 GHz   (s_memtime ticks per instruction %5.2f)\n", e.name, ns, ns * 2.4, cyc);
  }
  return 0;
}
