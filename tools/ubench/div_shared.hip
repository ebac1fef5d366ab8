// Bit-equality of the shared-reciprocal division of geometry.hip (div3_shared) with the compiler's IEEE fp32 `/` on the device:
//   (a) 2^33 pseudo-random operand pairs with the denominator in [2^-30, 2^30] and the numerator zero or in [2^-60, 2^60] (either sign),
//   (b) every denominator significand (2^23) x 16 exponents x 8 numerators.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off div_shared.hip -o div_shared && ./div_shared
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float shared_div(float n, float d, float r) {
  float q = n * r;
  q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
  return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float refined_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
__device__ __forceinline__ float make(uint32_t bits, int emin, int emax, bool allow_zero) {   // |value| in [2^emin, 2^emax)
  if (allow_zero && (bits & 0xff000000u) == 0) return 0.0f;
  const uint32_t e = 127 + emin + (bits >> 23 & 0xff) % (uint32_t)(emax - emin);
  return __builtin_bit_cast(float, (bits & 0x80000000u) | (e << 23) | (bits & 0x7fffffu));
}
__global__ void random_pairs(unsigned long long *bad, unsigned long long base) {
  const uint64_t i = base + (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long local = 0;
  for (int k = 0; k < 32; ++k) {
    const uint64_t h = mix(i * 32 + k);
    const float d = fabsf(make((uint32_t)h, -30, 30, false));
    const float n = make((uint32_t)(h >> 32), -60, 60, true);
    const float a = shared_div(n, d, refined_rcp(d)), b = n / d;
    local += __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b);
  }
  if (local) atomicAdd(bad, local);
}
__global__ void exhaustive(unsigned long long *bad) {
  const uint32_t m = blockIdx.x * 256 + threadIdx.x;               // 2^23 significands
  unsigned long long local = 0;
  const float nums[8] = {1.0f, -3.0f, 0.1f, 7.3e-9f, 5.1e11f, -2.0f / 3.0f, 1.1754944e-18f, 0.0f};
  for (int e = -30; e < 30; e += 4)
    for (int k = 0; k < 8; ++k) {
      const float d = __builtin_bit_cast(float, ((uint32_t)(127 + e) << 23) | m);
      const float a = shared_div(nums[k], d, refined_rcp(d)), b = nums[k] / d;
      local += __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b);
    }
  if (local) atomicAdd(bad, local);
}
int main() {
  unsigned long long *d, h = 0;
  hipMalloc(&d, 8); hipMemset(d, 0, 8);
  const unsigned long long per_launch = 1ull << 28;                 // threads x 32 pairs = 2^33 over 32 launches of 2^23 threads
  for (int l = 0; l < 32; ++l) hipLaunchKernelGGL(random_pairs, dim3(1u << 15), dim3(256), 0, 0, d, (unsigned long long)l * (1ull << 23));
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("random pairs: 2^33 compared, %llu mismatches\n", h);
  (void)per_launch;
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(exhaustive, dim3(1u << 15), dim3(256), 0, 0, d);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("all 2^23 denominator significands x 15 exponents x 8 numerators: %llu mismatches\n", h);
  return 0;
}
