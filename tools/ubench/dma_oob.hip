// Probe: does `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS for lanes whose offset is out
// of range (num_records), or does it skip the write?  (conv v2 relies on zero-fill for padding.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes, int soff) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 512; i += 64) smem[i] = -7.0f;  // poison
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000;             // "invalid" lanes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  float h[1024], *d, *o, ho[256];
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 1024 * 4, 1024);  // soffset = 256 floats
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("lane0 (valid, expect 257..260): %g %g %g %g\n", ho[0], ho[1], ho[2], ho[3]);
  printf("lane1 (invalid, 0 = zero-filled, -7 = write skipped): %g %g %g %g\n", ho[4], ho[5], ho[6], ho[7]);
  printf("lane62 (valid; voff 992 + soff 1024 = 2016 < 4096 expect 505..): %g  lane63 invalid: %g\n", ho[248], ho[252]);
  // partially out of range by size: num_records = 1024*4 - 8
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o, 2048, 1024);  // valid lanes beyond 2048 bytes?
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("num_records=2048, soff=1024: lane0 (voff 0): %g  lane62 (voff 992): %g lane 2 (voff 32) %g  [is soffset range-checked?]\n", ho[0], ho[248], ho[8]);
  return 0;
}
