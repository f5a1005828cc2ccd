// What a software grid barrier costs on this GPU -- the number a persistent (one cooperative
// launch) Cholesky would pay per panel / update hand-over instead of a kernel boundary
// (launch gap 1.5-2.9 us + first data 4.5 us per dependent launch).
//
// One cooperative launch (hipLaunchCooperativeKernel: every workgroup resident), `steps` rounds of
//   [each workgroup writes 64 doubles other workgroups will read] -> barrier -> [reads its
//   neighbour's 64 doubles, folds them into what it writes next]
// with the barrier a monotone counter in global memory: one agent-scope atomic add per workgroup,
// then a BOUNDED spin on the counter (a workgroup that waits longer than ~50 ms sets an abort
// flag and everybody leaves: a failed run, never a hung GPU).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o tools/grid_barrier_bench.bin
//   tools/grid_barrier_bench.bin [workgroups per CU = 1] [steps = 2000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

__global__ void __launch_bounds__(256)
k_rounds(unsigned *counter, unsigned *abort_flag, double *buf /* [2][grid][64] */, int steps, int with_data,
         unsigned long long *spins_out) {
  const unsigned G = gridDim.x;
  const int lane = threadIdx.x;
  double carry = (double)blockIdx.x;
  unsigned long long spins = 0;
  for (int s = 0; s < steps; ++s) {
    if (with_data && lane < 64) {
      __builtin_nontemporal_store(carry + lane, &buf[((size_t)(s & 1) * G + blockIdx.x) * 64 + lane]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();  // the stores above before the arrival
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(s + 1) * G;
      unsigned n = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++n > (1u << 20) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      spins += n;
    }
    __syncthreads();
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (with_data && lane < 64) {
      const unsigned nb = (blockIdx.x + 97) % G;  // a workgroup of (most likely) another XCD
      carry = 0.5 * carry + 1e-3 * __builtin_nontemporal_load(&buf[((size_t)(s & 1) * G + nb) * 64 + lane]);
    }
  }
  if (threadIdx.x == 0) spins_out[blockIdx.x] = spins;
  if (with_data && lane < 64) buf[(size_t)blockIdx.x * 64 + lane] = carry;
}

__global__ void k_tiny(double *p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0;
}

int main(int argc, char **argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
  const int steps = argc > 2 ? atoi(argv[2]) : 2000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_rounds, 256, 0));
  const int G = prop.multiProcessorCount * (per_cu < occ ? per_cu : occ);
  printf("%s: %d CUs, %d workgroups of 256 (occupancy limit %d per CU), %d rounds\n", prop.name,
         prop.multiProcessorCount, G, occ, steps);
  unsigned *counter, *abortf;
  double *buf;
  unsigned long long *spins;
  CK(hipMalloc(&counter, 256));
  abortf = counter + 32;
  CK(hipMalloc(&buf, (size_t)2 * G * 64 * sizeof(double)));
  CK(hipMalloc(&spins, G * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int with_data = 0; with_data <= 1; ++with_data) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(counter, 0, 256));
      CK(hipMemset(buf, 0, (size_t)2 * G * 64 * sizeof(double)));
      int st = steps;
      int wd = with_data;
      void *args[] = {&counter, &abortf, &buf, &st, &wd, &spins};
      CK(hipEventRecord(e0, 0));
      CK(hipLaunchCooperativeKernel((const void *)k_rounds, dim3(G), dim3(256), args, 0, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned ab = 0;
      CK(hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> hs(G);
      CK(hipMemcpy(hs.data(), spins, G * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      unsigned long long tot = 0;
      for (auto v : hs) tot += v;
      printf("  %s: %.3f us per round (%.2f ms in all), %.1f polls per workgroup and round%s\n",
             with_data ? "barrier + 512-byte hand-over between workgroups" : "barrier alone", 1e3 * ms / steps, ms,
             (double)tot / G / steps, ab ? "  ABORTED (a bounded spin ran out)" : "");
    }
  }
  // for comparison: the same number of dependent tiny launches on one stream
  double *p;
  CK(hipMalloc(&p, 8));
  CK(hipMemset(p, 0, 8));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_tiny, dim3(G), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %d dependent launches of an (almost) empty kernel with %d workgroups: %.3f us per launch\n", steps, G,
           1e3 * ms / steps);
  }
  return 0;
}
