// Micro-benchmark / probe (r04): (1) which XCD does workgroup (blockIdx.x, blockIdx.y) run on -- is it linear_id % 8 for 1-D and 2-D grids?
// (2) do agent-scope relaxed 64-bit atomic adds from ALL XCDs to ONE address in ordinary (coarse-grained) device memory add up?
// (The LayerNorm sums of csrc/cnn.hip are 64 shards per sample chosen as (4 blockIdx.x + wave) & 63 -- with linear_id % 8 = XCD every shard
// is only ever touched from one XCD.)
// Build: hipcc --offload-arch=gfx950 -O3 xcd_atomics.hip -o xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *xcc, long long *one, long long *two) {
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0) xcc[lin] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf;   // HW_REG_XCC_ID
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(one, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(one + 1, 3LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long *d = two + (((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63) * 2);
    __hip_atomic_fetch_add(d, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 1, 3LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  int *dx; long long *d1, *d2;
  hipMalloc(&dx, 1 << 20); hipMalloc(&d1, 64); hipMalloc(&d2, 64 * 16);
  const int grids[][2] = {{1856, 1}, {480, 2}, {77, 2}, {2016, 1}, {3328, 1}, {156, 2}};
  for (auto &g : grids) {
    int bad_total = 0;
    for (int rep = 0; rep < 200; ++rep) {
      hipMemset(d1, 0, 64); hipMemset(d2, 0, 64 * 16);
      hipLaunchKernelGGL(probe, dim3(g[0], g[1]), dim3(256), 0, 0, dx, d1, d2);
      hipDeviceSynchronize();
      long long h1[2], h2[128];
      hipMemcpy(h1, d1, 16, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
      const long long n = (long long)g[0] * g[1] * 4;
      long long s0 = 0, s1 = 0;
      for (int i = 0; i < 64; ++i) { s0 += h2[2 * i]; s1 += h2[2 * i + 1]; }
      if (h1[0] != n || h1[1] != 3 * n || s0 != n || s1 != 3 * n) {
        if (bad_total < 3) printf("grid (%d, %d) rep %d: single address %lld / %lld (expected %lld / %lld); sharded %lld / %lld\n", g[0], g[1], rep, h1[0], h1[1], n, 3 * n, s0, s1);
        ++bad_total;
      }
    }
    std::vector<int> hx(g[0] * g[1]);
    hipMemcpy(hx.data(), dx, hx.size() * 4, hipMemcpyDeviceToHost);
    int mism = 0;
    for (size_t i = 0; i < hx.size(); ++i) mism += hx[i] != (int)(i % 8);
    printf("grid (%4d, %d): atomics wrong in %d of 200 launches; workgroups with XCC_ID != linear_id %% 8: %d of %zu (first ids:", g[0], g[1], bad_total, mism, hx.size());
    for (int i = 0; i < 10; ++i) printf(" %d", hx[i]);
    printf(")\n");
  }
  return 0;
}
