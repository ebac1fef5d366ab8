// msi_probe_matrix_rate (round 6): what THIS part's matrix pipes sustain on changing operands -- the denominator of bench.py's
// `roofline.frac_of_sustained`.  The dense bf16 peak (2.5 PFLOP/s) is a constant-operand figure at 2.4 GHz; with operands that change between consecutive
// MFMAs the shader clock itself falls (tools/ubench/clock_probe.hip, profiles/r06_clock.txt: s_memtime -- which counts real shader cycles: cycles per dependent
// VALU instruction stay put -- advances at 2.39 GHz under constant operands and 1.89 GHz under changing ones, the matrix pipe issuing back to back in both
// cases; the driver's sclk reading does not follow).  No memory traffic, no LDS: four waves per workgroup, each issuing v_mfma_f32_32x32x16_bf16 back to
// back on twelve pseudo-random bf16 operand registers (CHANGING = 1) or on two constant ones (0).  Measurement infrastructure, not part of a frame.
#include "msi_common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHANGING>
__global__ void __launch_bounds__(256) matrix_rate_kernel(long iters, unsigned long long *ticks, float *sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  v4f f[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    unsigned u[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)   // bf16 pairs with magnitudes in [2^-31, 2): sign, exponent and significand bits all toggle
      u[c] = CHANGING ? ((((unsigned)(lane * 97 + i * 31 + c * 7 + wave * 13 + blockIdx.x * 5) * 2654435761u) >> 3 & 0x3f7f3f7fu) | 0x30003000u) : 0x3f003f00u;
    f[i] = v4f{__builtin_bit_cast(float, u[0]), __builtin_bit_cast(float, u[1]), __builtin_bit_cast(float, u[2]), __builtin_bit_cast(float, u[3])};
  }
  const unsigned long long m0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int ia = CHANGING ? m : 0, ib = CHANGING ? (m + 5) % 12 : 1;
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[ia]), __builtin_bit_cast(bf16x8, f[ib]), acc[m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  if (s == 123.456f) sink[0] = s;     // (keeps the accumulators alive; never true for these operands in practice, harmless if it is)
  if (tid == 0 && ticks) { ticks[2 * blockIdx.x] = m1 - m0; ticks[2 * blockIdx.x + 1] = q1 - q0; }
}
}  // namespace

extern "C" int32_t msi_probe_matrix_rate(int32_t changing_operands, int64_t iterations, int32_t num_workgroups, uint64_t *ticks_device, float *sink_device,
                                          msi_stream_t stream_) {
  MSI_REQUIRE(iterations > 0 && num_workgroups > 0 && sink_device, "probe_matrix_rate: bad argument");
  hipStream_t stream = msi::as_stream(stream_);
  if (changing_operands)
    matrix_rate_kernel<1><<<num_workgroups, 256, 0, stream>>>((long)iterations, reinterpret_cast<unsigned long long *>(ticks_device), sink_device);
  else
    matrix_rate_kernel<0><<<num_workgroups, 256, 0, stream>>>((long)iterations, reinterpret_cast<unsigned long long *>(ticks_device), sink_device);
  return msi::check_launch("probe_matrix_rate");
}
