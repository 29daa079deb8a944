// lds_atomic_rate.hip -- round 4: what does an LDS atomic cost on gfx950 next to a plain LDS access?  (The table kernel of
// the node-identity pass does one 64-bit compare-and-swap + one 32-bit min per successor; is it bound by them?)
// 256 threads per workgroup, 6 workgroups per CU resident (24 KB of LDS each, like id_tables_kernel), every thread ITER
// operations on pseudo-random slots of a 2048-entry table.  Prints cycles per wave-instruction per CU (all resident waves
// share the CU's LDS) = busy time of the LDS pipe per instruction when that is the bound.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kSlots = 2048, ITER = 512;
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int same) {
  __shared__ unsigned long long keys[kSlots];
  __shared__ unsigned int vals[kSlots];
  for (int i = threadIdx.x; i < kSlots; i += 256) { keys[i] = ~0ull; vals[i] = 0xffffffffu; }
  __syncthreads();
  unsigned int x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long acc = 0;
#pragma unroll 4
  for (int it = 0; it < ITER; it++) {
    x = x * 1664525u + 1013904223u;
    const unsigned int s = same ? (unsigned)(it & (kSlots - 1)) : (x >> 11) & (kSlots - 1);  // same: all lanes one slot
    if (OP == 0) acc += keys[s];
    if (OP == 1) keys[s] = x;
    if (OP == 2) acc += atomicCAS(&keys[s], ~0ull, (unsigned long long)x);
    if (OP == 3) acc += atomicCAS(&vals[s], 0xffffffffu, x);
    if (OP == 4) atomicMin(&vals[s], x);
    if (OP == 5) acc += atomicMin(&vals[s], x);
    if (OP == 6) acc += atomicAdd(&vals[s], 1u);
    if (OP == 7) acc += vals[s];
    if (OP == 8) vals[s] = x;
    if (OP == 9) atomicMin(&keys[s], (unsigned long long)x);
    if (OP == 10) acc += atomicMin(&keys[s], (unsigned long long)x);
  }
  if (acc == 0x123456789ull) out[0] = acc;
}
template <int OP>
void run(const char *name, unsigned long long *d) {
  for (int same = 0; same < 2; same++) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int wgs = 256 * 6 * 4;
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, same);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, d, same);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double wave_instr_per_cu = (double)wgs * 4 * ITER / 256.0;
    printf("%-28s %s  %.3f ms  %.1f ns per wave-instruction per CU (%.1f clk at 2.4 GHz)\n", name, same ? "one slot per instr " : "random slots       ", ms,
           ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
  }
}
int main() {
  unsigned long long *d; (void)hipMalloc(&d, 64);
  run<0>("ds_read_b64", d); run<1>("ds_write_b64", d); run<7>("ds_read_b32", d); run<8>("ds_write_b32", d);
  run<2>("ds_cmpst_rtn_b64", d); run<3>("ds_cmpst_rtn_b32", d); run<4>("ds_min_u32", d); run<5>("ds_min_rtn_u32", d);
  run<6>("ds_add_rtn_u32", d); run<9>("ds_min_u64", d); run<10>("ds_min_rtn_u64", d);
  return 0;
}
