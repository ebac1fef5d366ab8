// Does a packed fp32 FMA (v_pk_fma_f32: two fp32 FMAs per lane) issue at the rate of a scalar one on gfx950?
// 1 024 workgroups x 256 threads, 8 independent accumulator chains per lane, 4 096 iterations.
//   hipcc --offload-arch=gfx950 -O3 pk_fma_rate.hip -o pk_fma_rate && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters) {
  f2 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, (float)i};
  const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
      else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(av.x), "v"(bv.x));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *d; hipMalloc(&d, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pk = 0; pk < 2; ++pk) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, d, 1.0001f, 0.5f, 4096);
      else hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, d, 1.0001f, 0.5f, 4096);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double insts = 1024.0 * 4 * 4096 * 8;   // wave-instructions
    printf("%s: %.3f ms, %.2f wave-instructions/ns chip-wide, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", best,
           insts / (best * 1e6), insts * 64 * 2 * (pk ? 2 : 1) / (best * 1e-3) / 1e12);
  }
  return 0;
}
