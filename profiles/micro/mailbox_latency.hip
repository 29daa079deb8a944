// Round trip of a request to the GPU, two ways (profiles/micro/mailbox_latency.sh):
//   (a) one kernel launch + hipStreamSynchronize (what a small synchronous batch costs today, without any work)
//   (b) a resident kernel polling a doorbell word in pinned host memory and answering into another one
// The kernel in (b) leaves after `idle_us` without a request (watchdog) and on a quit word.
#include <hip/hip_runtime.h>

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Mailbox { uint64_t doorbell; uint64_t pad0[7]; uint64_t done; uint64_t pad1[7]; uint32_t quit, alive; uint32_t stage, pad2; uint64_t payload[64]; };

__global__ void empty_kernel(uint64_t *out, const uint64_t *in) { if (in) out[threadIdx.x] = in[threadIdx.x] + 1; }
// the kernel itself tells the host that it is through: the host spins on `flag` instead of hipStreamSynchronize
__global__ void flag_kernel(uint64_t *out, const uint64_t *in, uint64_t *flag, uint64_t seq) {
  if (in) out[threadIdx.x] = in[threadIdx.x] + 1;
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// All 64 lanes of wave 0 poll / publish together (same address, same value): a single lane doing it (if (threadIdx.x
// == 0) ...) next to the barriers of a loop is lane divergence around a convergent operation -- hipcc threaded lanes
// 1..63 of wave 0 into the next trip's s_barrier while lane 0 still had its store to do, and the workgroup hung.
__global__ void service_kernel(Mailbox *mb, uint64_t seq0, uint64_t idle_ticks, uint64_t *bell, uint64_t *payload) {
  uint64_t seq = seq0;
  __shared__ uint64_t s_cmd;
  const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  for (;;) {
    if (wave0) {
      const uint64_t t0 = wall_clock64();
      uint64_t v;
      for (;;) {
        v = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == seq + 1) break;
        if (__hip_atomic_load(&mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || wall_clock64() - t0 > idle_ticks) { v = ~0ull; break; }
      }
      s_cmd = v;
    }
    __syncthreads();
    const uint64_t cmd = s_cmd;
    __syncthreads();
    if (cmd == ~0ull) break;
    seq = cmd;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (threadIdx.x < 64) mb->payload[threadIdx.x] = payload[threadIdx.x] + 1;  // read the request, write the answer
    __threadfence_system();
    __syncthreads();
    if (wave0) __hip_atomic_store(&mb->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
  }
  if (wave0) __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Mailbox *mb;
  CK(hipHostMalloc((void **)&mb, sizeof(Mailbox), hipHostMallocCoherent));
  *mb = Mailbox{};
  uint64_t *dbuf;
  CK(hipMalloc((void **)&dbuf, 4096));
  const int N = 2000;
  // (a0) empty launch + sync
  for (int rep = 0; rep < 2; rep++) {
    const double t0 = now_us();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, dbuf, (const uint64_t *)nullptr); CK(hipStreamSynchronize(s)); }
    if (rep) printf("launch + sync, empty kernel:                  %.2f us\n", (now_us() - t0) / N);
  }
  // (a1) kernel reads + writes pinned host memory
  for (int rep = 0; rep < 2; rep++) {
    const double t0 = now_us();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, mb->payload, (const uint64_t *)mb->payload); CK(hipStreamSynchronize(s)); }
    if (rep) printf("launch + sync, kernel reads+writes host block: %.2f us\n", (now_us() - t0) / N);
  }
  // (a2) launch, completion seen through a flag the kernel writes into pinned memory (no hipStreamSynchronize per launch)
  for (int rep = 0; rep < 2; rep++) {
    volatile uint64_t *fl = &mb->done;
    const double t0 = now_us();
    for (int i = 0; i < N; i++) {
      const uint64_t q = (uint64_t)rep * N + i + 1;
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, mb->payload, (const uint64_t *)mb->payload, (uint64_t *)&mb->done, q);
      while (*fl != q) {}
    }
    if (rep) printf("launch, kernel-written flag instead of sync:   %.2f us\n", (now_us() - t0) / N);
    CK(hipStreamSynchronize(s));
  }
  // (a3) the flag written by a stream memory operation behind the kernel / (a4) by a second one-wave kernel
  for (int variant = 0; variant < 2; variant++) {
    for (int rep = 0; rep < 2; rep++) {
      volatile uint64_t *fl = &mb->done;
      const double t0 = now_us();
      for (int i = 0; i < N; i++) {
        const uint64_t q = 100000 + (uint64_t)variant * 10 * N + (uint64_t)rep * N + i + 1;
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, mb->payload, (const uint64_t *)mb->payload);
        if (variant == 0) CK(hipStreamWriteValue64(s, (void *)&mb->done, q, 0));
        else hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, mb->payload + 64 - 64, (const uint64_t *)nullptr, (uint64_t *)&mb->done, q);
        while (*fl != q) {}
      }
      if (rep) printf("launch + %s, host spins on the flag: %.2f us\n", variant == 0 ? "hipStreamWriteValue64" : "a second (flag) kernel ", (now_us() - t0) / N);
      CK(hipStreamSynchronize(s));
    }
  }
  mb->done = 0;
  uint64_t hz = 100000000ull;  // wall_clock64: 100 MHz
  {  // watchdog alone
    mb->quit = 0; mb->alive = 1;
    const double t0 = now_us();
    hipLaunchKernelGGL(service_kernel, dim3(1), dim3(256), 0, s, mb, 0ull, hz / 1000 * 5, &mb->doorbell, mb->payload);  // 5 ms
    CK(hipStreamSynchronize(s));
    printf("watchdog exit after %.1f ms idle, alive %u\n", (now_us() - t0) / 1e3, mb->alive);
  }
  {  // quit word alone
    mb->quit = 0; mb->alive = 1;
    const double t0 = now_us();
    hipLaunchKernelGGL(service_kernel, dim3(1), dim3(256), 0, s, mb, 0ull, hz * 3, &mb->doorbell, mb->payload);  // 3 s
    while (now_us() - t0 < 2000.0) {}
    *(volatile uint32_t *)&mb->quit = 1;
    CK(hipStreamSynchronize(s));
    printf("quit seen after %.1f ms, alive %u\n", (now_us() - t0) / 1e3, mb->alive);
    mb->quit = 0;
  }
  // (b) resident kernel
  mb->alive = 1;
  hipLaunchKernelGGL(service_kernel, dim3(1), dim3(256), 0, s, mb, 0ull, hz / 1000 * 500, &mb->doorbell, mb->payload);  // 0.5 s idle watchdog
  CK(hipGetLastError());
  volatile uint64_t *done = &mb->done;
  uint64_t seq = 0;
  for (int rep = 0; rep < 2; rep++) {
    const double t0 = now_us();
    for (int i = 0; i < N; i++) {
      seq++;
      std::atomic_thread_fence(std::memory_order_release);
      *(volatile uint64_t *)&mb->doorbell = seq;
      const double w0 = now_us();
      while (*done != seq) {
        if (now_us() - w0 > 2e6) { fprintf(stderr, "no answer at seq %llu (alive %u, stage %u, done %llu)\n", (unsigned long long)seq, mb->alive, mb->stage, (unsigned long long)mb->done); *(volatile uint32_t *)&mb->quit = 1; hipStreamSynchronize(s); return 2; }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (rep) printf("resident kernel, doorbell -> done:             %.2f us   (payload %llu)\n", (now_us() - t0) / N, (unsigned long long)mb->payload[0]);
  }
  *(volatile uint32_t *)&mb->quit = 1;
  CK(hipStreamSynchronize(s));
  printf("alive after quit: %u\n", mb->alive);
  {  // (c) doorbell and request payload in DEVICE memory that the host writes directly (fine-grained, large BAR)
    uint64_t *dreq = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&dreq, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(fine-grained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
      CK(hipMemset(dreq, 0, 4096));
      CK(hipDeviceSynchronize());
      mb->quit = 0; mb->alive = 1; mb->done = 0;
      hipLaunchKernelGGL(service_kernel, dim3(1), dim3(256), 0, s, mb, 0ull, hz / 1000 * 500, dreq, dreq + 8);
      CK(hipGetLastError());
      volatile uint64_t *bell = dreq;
      uint64_t q = 0;
      for (int rep = 0; rep < 2; rep++) {
        const double t0 = now_us();
        for (int i = 0; i < N; i++) {
          q++;
          for (int k = 0; k < 16; k++) ((volatile uint64_t *)dreq)[8 + k] = q + k;  // the request
          std::atomic_thread_fence(std::memory_order_release);
          _mm_sfence();  // device memory is mapped write-combining: the request must leave the WC buffers before the bell
          *bell = q;
          _mm_sfence();
          const double w0 = now_us();
          while (*done != q) {
            if (now_us() - w0 > 2e6) { fprintf(stderr, "(c) no answer at %llu\n", (unsigned long long)q); *(volatile uint32_t *)&mb->quit = 1; (void)hipStreamSynchronize(s); return 2; }
          }
          std::atomic_thread_fence(std::memory_order_acquire);
          if (mb->payload[3] != q + 3 + 1) { fprintf(stderr, "(c) wrong payload %llu at %llu\n", (unsigned long long)mb->payload[3], (unsigned long long)q); *(volatile uint32_t *)&mb->quit = 1; (void)hipStreamSynchronize(s); return 3; }
        }
        if (rep) printf("resident kernel, doorbell in device memory:    %.2f us\n", (now_us() - t0) / N);
      }
      *(volatile uint32_t *)&mb->quit = 1;
      CK(hipStreamSynchronize(s));
      mb->quit = 0;
    }
  }
  // watchdog: start again, post nothing
  mb->quit = 0; mb->alive = 1;
  const double t0 = now_us();
  hipLaunchKernelGGL(service_kernel, dim3(1), dim3(256), 0, s, mb, seq, hz / 1000 * 5, &mb->doorbell, mb->payload);  // 5 ms
  CK(hipStreamSynchronize(s));
  printf("watchdog exit after %.1f ms idle, alive %u\n", (now_us() - t0) / 1e3, mb->alive);
  return 0;
}
