// Micro-benchmark: does the fp32 MFMA (v_mfma_f32_32x32x2_f32; BF = 1: the bf16 v_mfma_f32_32x32x16_bf16, 32 pipe cycles each)
// overlap with VALU work on gfx950?
//   mode 0: MFMA only; mode 1: VALU only; mode 2: both interleaved in one wave;
//   mode 3: even waves MFMA-only, odd waves VALU-only (needs >= 2 waves per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV, int BF>
__global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b) {
  f32x16 acc0 = {0}, acc1 = {0};
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  const bf16x8 qa = __builtin_bit_cast(bf16x8, v4f{a, b, a, b}), qb = __builtin_bit_cast(bf16x8, v4f{b, a, b, a});
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const int wave = threadIdx.x >> 8;  // waves 0-3 (one per SIMD) vs waves 4-7 (their SIMD partners)
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (BF) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, qb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qb, qa, acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < NV / 4; ++j) v[j & 7] = v[j & 7] * a + b;
        }
      }
    }
    if (do_v && MODE != 2) {
#pragma unroll
      for (int j = 0; j < 2 * NV; ++j) v[j & 7] = v[j & 7] * a + b;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, int BF>
float run(float *d, int blocks, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV, BF>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV, BF>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int BF>
void sweep(float *d) {
  const int iters = 20000;
  // per iteration: 8 MFMAs (fp32: 512 matrix-pipe cycles, bf16: 256); VALU: 2*NV fma (mode 1/3) or 8*(NV/4) (mode 2)
  for (int wpb : {256, 512}) {  // 1 or 2 waves per SIMD (one block per CU)
    printf("%s MFMA, threads/block=%d (waves/SIMD=%d), 256 blocks, %d iters; per iter 8 MFMA = %d pipe cycles\n", BF ? "bf16" : "fp32", wpb, wpb / 256, iters, BF ? 256 : 512);
    printf("  NV=64 : mfma %.2f ms  valu(128 fma) %.2f ms  both-in-one-wave(8 MFMA + 128 fma) %.2f ms  split-waves %.2f ms\n",
           run<0, 64, BF>(d, 256, wpb, iters), run<1, 64, BF>(d, 256, wpb, iters), run<2, 64, BF>(d, 256, wpb, iters), run<3, 64, BF>(d, 256, wpb, iters));
    printf("  NV=16 : mfma %.2f ms  valu(32 fma) %.2f ms  both-in-one-wave(8 MFMA + 32 fma) %.2f ms  split-waves %.2f ms\n",
           run<0, 16, BF>(d, 256, wpb, iters), run<1, 16, BF>(d, 256, wpb, iters), run<2, 16, BF>(d, 256, wpb, iters), run<3, 16, BF>(d, 256, wpb, iters));
  }
}

int main() {
  float *d; hipMalloc(&d, 256 * 8 * 512 * sizeof(float));
  sweep<0>(d);
  sweep<1>(d);
  return 0;
}
