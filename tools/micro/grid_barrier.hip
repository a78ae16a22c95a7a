// Developer microbenchmark (round 6): what one grid-wide phase boundary costs INSIDE a kernel on MI355X, placement-independent forms only
// (the bilateral solver at batch 1 is ~45 dependent phases of ~1 us of work on 20 k vertices; a kernel launch boundary costs ~5.7 us there).
//
// G workgroups (<= one per CU, all resident) run ITERS phases.  In every phase each workgroup writes one 32-byte record per thread
// (the solver's CgDyn), meets the others at a barrier, then reads ten records written by OTHER workgroups in that phase and checks them.
//   mode 0  release / acquire fences (agent scope) around an agent-scope counter — the portable form (cooperative-groups style)
//   mode 1  every shared store and load `sc1` (write-through / L2-bypassing), no fences: counter add after each storing wave's vmcnt(0) +
//           workgroup barrier; poll with an sc1 load (MI355X_MICROARCH.md "Valid forms", third table row)
//   mode 2  as 1, one poller per workgroup sleeping between polls (s_sleep 1)
// Prints us per phase and the number of stale records seen (must be 0).   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct Rec { double a, b, c, d; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(Rec* p, const Rec& r) {
  const u32x4* q = (const u32x4*)&r;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(p), "v"(q[0]), "v"(q[1]) : "memory");
}
__device__ __forceinline__ Rec ld_sc1(const Rec* p) {
  u32x4 lo, hi;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
  Rec r;
  ((u32x4*)&r)[0] = lo; ((u32x4*)&r)[1] = hi;
  return r;
}
__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(Rec* buf0, Rec* buf1, unsigned* cnt, int iters, int* stale, int* timeout) {
  const int G = gridDim.x, n = G * 256, me = blockIdx.x * 256 + threadIdx.x;
  __shared__ int ok;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    Rec* wr = (it & 1) ? buf1 : buf0;
    const Rec mine = {(double)it, (double)me, (double)(it + me), 1.0};
    if (MODE == 0) wr[me] = mine; else st_sc1(wr + me, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(it + 1) * (unsigned)G;
      int good = 1;
      if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins)
          if (spins > (1 << 22)) { good = 0; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; ld_u32_sc1(cnt) < target; ++spins) {
          if (spins > (1 << 22)) { good = 0; break; }
          if (MODE == 2) __builtin_amdgcn_s_sleep(1);
        }
      }
      ok = good;
      if (!good) *timeout = 1;
    }
    __syncthreads();
    if (!ok) return;
    // ten records written by other workgroups in THIS phase (the solver's neighbour gather)
    for (int e = 1; e <= 10; ++e) {
      const int j = (me + e * 2609) % n;              // another workgroup for almost every e
      const Rec r = MODE == 0 ? wr[j] : ld_sc1(wr + j);
      bad += (r.a != (double)it) || (r.b != (double)j);
    }
  }
  if (bad) atomicAdd(stale, bad);
}
template <int MODE> static void run(int G, int iters) {
  Rec *b0, *b1; unsigned* cnt; int *stale, *timeout;
  hipMalloc(&b0, (size_t)G * 256 * sizeof(Rec)); hipMalloc(&b1, (size_t)G * 256 * sizeof(Rec));
  hipMalloc(&cnt, 256); hipMalloc(&stale, 4); hipMalloc(&timeout, 4);
  float best = 1e30f;
  int hs = 0, ht = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(cnt, 0, 256); hipMemset(stale, 0, 4); hipMemset(timeout, 0, 4);
    hipMemset(b0, 0xff, (size_t)G * 256 * sizeof(Rec)); hipMemset(b1, 0xff, (size_t)G * 256 * sizeof(Rec));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, b0, b1, cnt, iters, stale, timeout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
    int s, t; hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost); hipMemcpy(&t, timeout, 4, hipMemcpyDeviceToHost);
    hs += s; ht += t;
  }
  printf("mode %d  G = %3d workgroups x 256: %.2f us per phase (best of 3, %d phases), stale records %d, timeouts %d\n", MODE, G, best * 1e3 / iters, iters, hs, ht);
  hipFree(b0); hipFree(b1); hipFree(cnt); hipFree(stale); hipFree(timeout);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  for (int G : {20, 40, 80, 160}) { run<0>(G, iters); run<1>(G, iters); run<2>(G, iters); }
  return 0;
}
