// Micro-benchmark (r04): how many ds_read_b128 fragment reads per MFMA can a CU sustain before the LDS pipe, not the matrix
// pipe, sets the pace?  The loop body of the six-product kernels (conv_halo_x3_kernel) is NR reads + 12 v_mfma_f32_32x32x16_bf16
// per wave and tap; here the same body runs with NR = 0 / 6 / 9 / 12 / 18 reads, with and without the per-tap workgroup barrier,
// at 2 / 3 / 4 workgroups of four waves per CU (dynamic LDS sets the residency).  Output: cycles of the kernel's own clock per
// iteration against the 384 matrix cycles a wave-iteration needs (x waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 lds_mfma_rate.hip -o lds_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OFF>
__device__ __forceinline__ v4f rd(unsigned addr) {
  v4f v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int NR, int BAR, int NM>
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 10240; i += 256) reinterpret_cast<float *>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  // conflict-free pattern: 64 lanes x 16 B consecutive (1 KB per instruction), 18 KB window per wave pair
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem + (wave & 1) * 20480 + lane * 16;
  f32x16 acc0 = {0}, acc1 = {0};
  v4f f[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) f[i] = v4f{0.5f, 0.25f, 1.f, 2.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define RD(i) if (NR > i) f[i] = rd<i * 1024>(base);
    RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11) RD(12) RD(13) RD(14) RD(15) RD(16) RD(17)
#undef RD
    if (NR > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]),
                                                   "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]), "+v"(f[16]), "+v"(f[17]) :: "memory");
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, f[m % 18]), b = __builtin_bit_cast(bf16x8, f[(m + 7) % 18]);
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


// PIPE: the reads of iteration it + 1 are issued BEFORE the MFMAs of iteration it (two fragment sets, 96 VGPRs): a wave's own
// reads fly under its own matrix work instead of relying on the other waves of the SIMD being in the other phase
template <int NR, int BAR, int NM>
__global__ void __launch_bounds__(256) kp(float *out, int iters, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 10240; i += 256) reinterpret_cast<float *>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem + (wave & 1) * 20480 + lane * 16;
  f32x16 acc0 = {0}, acc1 = {0};
  v4f f[12], g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) f[i] = g[i] = v4f{0.5f, 0.25f, 1.f, 2.f};
#define RDS(dst) { dst[0] = rd<0>(base); dst[1] = rd<1024>(base); dst[2] = rd<2048>(base); dst[3] = rd<3072>(base); dst[4] = rd<4096>(base); dst[5] = rd<5120>(base); \
    if (NR > 6) { dst[6] = rd<6144>(base); dst[7] = rd<7168>(base); dst[8] = rd<8192>(base); } \
    if (NR > 9) { dst[9] = rd<9216>(base); dst[10] = rd<10240>(base); dst[11] = rd<11264>(base); } }
#define WAITS(dst) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]), "+v"(dst[8]), "+v"(dst[9]), "+v"(dst[10]), "+v"(dst[11]) :: "memory");
#define MM(src) _Pragma("unroll") for (int m = 0; m < NM; ++m) { \
      const bf16x8 a = __builtin_bit_cast(bf16x8, src[m % 12]), b = __builtin_bit_cast(bf16x8, src[(m + 7) % 12]); \
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0); \
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  RDS(f)
  for (int it = 0; it < iters; it += 2) {
    WAITS(f)
    RDS(g)
    __builtin_amdgcn_sched_barrier(0);
    MM(f)
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) __builtin_amdgcn_s_barrier();
    WAITS(g)
    RDS(f)
    __builtin_amdgcn_sched_barrier(0);
    MM(g)
    __builtin_amdgcn_sched_barrier(0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  WAITS(f)
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) s += f[i].x;
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NR, int BAR, int NM, int PIPE = 0>
void run(float *d, unsigned long long *dc, int wg_per_cu, int iters) {
  const int lds = wg_per_cu == 2 ? 65536 : wg_per_cu == 3 ? 49152 : 40960;
  const int blocks = 256 * wg_per_cu;
  auto fn = PIPE ? kp<NR, BAR, NM> : k<NR, BAR, NM>;
  hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // matrix time of the launch: every SIMD runs wg_per_cu waves x iters x NM x 32 cycles (8 passes x 4)
  const double mcyc = (double)wg_per_cu * iters * NM * 32;
  static unsigned long long hc[1024];
  hipMemcpy(hc, dc, blocks * 8, hipMemcpyDeviceToHost);
  double cmin = 1e30, cmax = 0, csum = 0;
  for (int i = 0; i < blocks; ++i) { cmin = hc[i] < cmin ? hc[i] : cmin; cmax = hc[i] > cmax ? hc[i] : cmax; csum += hc[i]; }
  // s_memtime ticks at 100 MHz: the wall time is the event's; cycles from the MFMA-only run's calibration (printed first)
  printf("%s reads %2d  mfma %2d  barrier %d  wg/cu %d : %8.3f ms   (matrix cycles per SIMD %.0f -> %.2f GHz-equivalent if matrix-bound; s_memtime ticks per block min %.0f mean %.0f max %.0f = %.3f GHz at the max)\n", PIPE ? "pipelined" : "phased   ", NR, NM, BAR, wg_per_cu, ms,
         mcyc, mcyc / (ms * 1e6), cmin, csum / blocks, cmax, cmax / (ms * 1e6));
}


// RAND: matrix-only with operand registers that CHANGE between consecutive MFMAs (eight pre-loaded sets of pseudo-random fp16 bits in [-2, 2), or
// post-ReLU-like data: half of the values zero) -- how far does operand toggling alone pull the clock down?
template <int KIND>
__global__ void __launch_bounds__(256) kr(float *out, int iters, unsigned long long *cyc) {
  const int tid = threadIdx.x;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f16x8 a[8], b[8];
  unsigned s = 0x9e3779b9u * (blockIdx.x * 256 + tid + 1);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u; const float u = (float)(s >> 8) * (1.f / 16777216.f) * 4.f - 2.f;
      s = s * 1664525u + 1013904223u; const float v = (float)(s >> 8) * (1.f / 16777216.f) * 4.f - 2.f;
      a[i][j] = (_Float16)(KIND == 1 ? (u > 0.f ? u : 0.f) : u);
      b[i][j] = (_Float16)(v * 0.05f);
    }
  f32x16 acc0 = {0}, acc1 = {0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[m & 7], a[(m + 3) & 7], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[m & 7], a[(m + 5) & 7], acc0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
  out[blockIdx.x * 256 + tid] = r;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run_rand(float *d, unsigned long long *dc, int wg_per_cu, int iters) {
  const int lds = wg_per_cu == 2 ? 65536 : wg_per_cu == 3 ? 49152 : 40960;
  const int blocks = 256 * wg_per_cu;
  hipFuncSetAttribute((const void *)kr<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kr<KIND>, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kr<KIND>, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long hc[1024];
  hipMemcpy(hc, dc, blocks * 8, hipMemcpyDeviceToHost);
  double cmax = 0;
  for (int i = 0; i < blocks; ++i) cmax = hc[i] > cmax ? hc[i] : cmax;
  printf("matrix-only, %s fp16 operands changing every MFMA, wg/cu %d : %8.3f ms  ticks max %.0f = %.3f GHz  (%.0f TFLOP/s of fp16 MFMA on 256 CUs)\n",
         KIND == 1 ? "post-ReLU-like (half zeros)" : "random                     ", wg_per_cu, ms, cmax, cmax / (ms * 1e6),
         (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
}

int main() {
  float *d;
  unsigned long long *dc;
  hipMalloc(&d, 1024 * 256 * 4);
  hipMalloc(&dc, 1024 * 8);
  const int iters = 20000;
  for (int w = 2; w <= 4; ++w) { run_rand<0>(d, dc, w, iters); run_rand<1>(d, dc, w, iters); }
  for (int w = 2; w <= 4; ++w) {
    run<0, 0, 12>(d, dc, w, iters);
    run<6, 0, 12>(d, dc, w, iters);
    run<9, 0, 12>(d, dc, w, iters);
    run<12, 0, 12>(d, dc, w, iters);
    run<18, 0, 12>(d, dc, w, iters);
    run<6, 1, 12>(d, dc, w, iters);
    run<9, 1, 12>(d, dc, w, iters);
    run<12, 1, 12>(d, dc, w, iters);
    run<12, 0, 0>(d, dc, w, iters);
    run<12, 1, 0>(d, dc, w, iters);
    run<6, 1, 12, 1>(d, dc, w, iters);
    run<9, 1, 12, 1>(d, dc, w, iters);
    run<12, 1, 12, 1>(d, dc, w, iters);
    run<12, 0, 12, 1>(d, dc, w, iters);
  }
  return 0;
}
