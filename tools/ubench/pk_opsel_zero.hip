// Stand-alone reproducer attempt (r05) for the finding of DESIGN.md section 4 "the wobble": in csrc's convt_halo_x3_kernel<2> the low half of
//     v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]          (d.lo = a.lo * b.HI)
// came out 0 for lanes 48-63 in ~0.1 % of forwards -- in the FIRST workgroup of a CU, whose epilogue ran while its two co-resident workgroups (launched with it)
// were in the last taps of an LDS-DMA + ds_read_b128 + fp16-MFMA k-loop.  Here: three workgroups of four waves per CU (40 KB of dynamic LDS each, 160 VGPRs), every
// workgroup alternates a "k-loop" phase (per tap: two buffer_load ... lds of 16 B per lane, eight ds_read_b128, six v_mfma_f32_32x32x16_f16 on changing operands, one
// barrier) of a length that depends on its slot with an "epilogue" phase: the statistics code of emit_tile_impl as the compiler wrote it (%bb.241 / %bb.246 of that
// kernel, instruction for instruction as inline asm) over 16 values per lane, the cross-half product checked against an unpacked v_mul_f32 of the same operands.
// Output: mismatching (launch, workgroup, wave, lane) tuples and what the low product was.
// Build: hipcc --offload-arch=gfx950 -O3 pk_opsel_zero.hip -o pk_opsel_zero ; run: ./pk_opsel_zero [launches] [taps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int OFF>
__device__ __forceinline__ v4f rd(unsigned addr) {
  v4f v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

struct Rec { int launch, wg, wave, lane; float lo, want, a, b; };

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k(const float *wsrc, float *out, Rec *rec, int *nrec, int launch, int taps, int rounds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: [A region 16 KB | B ring 2 x 8 KB]
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<_Float16 *>(smem)[i] = (_Float16)(0.002f * ((i * 7 + blockIdx.x) & 255) - 0.25f);   // (the whole 32 KB as small fp16 values)
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (wave & 1) * 8192 + lane * 16, b_base = lds_base + 16384 + (wave >> 1) * 4096 + lane * 16;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wsrc, 0, 1 << 26, 0x00020000);
  f32x16 acc = {0}, acc_lo = {0};
  unsigned bad_total = 0;
  // the slot on the CU decides the phase lengths: the first workgroup of a CU leaves its loop while its neighbours are still in theirs
  const int slot = blockIdx.x / 256;              // 768 workgroups: 0, 1, 2 = first / second / third on its CU (round-robin dispatch)
  for (int r = 0; r < rounds; ++r) {
    const int my_taps = taps + slot * (taps / 8) + ((r * 5 + slot * 3) % 7);
    int st = 0;
    for (int t = 0; t < my_taps; ++t) {
      // DMA of the next tap's "weights": two planes, 16 B per lane, into the other ring stage
      char *sB = smem + 16384 + (st ^ 1) * 8192 + wave * 1024;
      const unsigned voff = (unsigned)(((blockIdx.x * 131 + t * 17 + wave * 16 + (lane >> 2)) & 0xffff) * 64 + (lane & 3) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)sB, 16, voff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(sB + 4096), 16, voff, 1 << 20, 0, 0);
      v4f ah[2], am[2], bh[2], bm[2];
      const unsigned bst = st * 8192;
      ah[0] = rd<0>(a_base); bh[0] = rd<0>(b_base + bst); am[0] = rd<1024>(a_base); bm[0] = rd<1024>(b_base + bst);
      ah[1] = rd<2048>(a_base); bh[1] = rd<2048>(b_base + bst); am[1] = rd<3072>(a_base); bm[1] = rd<3072>(b_base + bst);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[0]), "+v"(bh[0]), "+v"(am[0]), "+v"(bm[0]) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[1]), "+v"(bh[1]), "+v"(am[1]), "+v"(bm[1]) :: "memory");
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bm[s]), __builtin_bit_cast(f16x8, ah[s]), acc_lo, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[s]), __builtin_bit_cast(f16x8, ah[s]), acc, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh[s]), __builtin_bit_cast(f16x8, am[s]), acc_lo, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      st ^= 1;
    }
    // ---- "epilogue": per group of four values v = (x, y, z, w) and a wave-uniform pivot P (an SGPR pair), the compiler's sequence ----
    const float P = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0])));
    float s1 = 0.f, s2 = 0.f;
    unsigned bad = 0;
    float blo = 0.f, bwant = 0.f, ba = 0.f, bb = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float fx = acc[4 * g] + 0.37f * (lane + 1), fy = acc[4 * g + 1] - 0.11f * (lane + 3), fz = acc[4 * g + 2] + 1.7f, fw = acc[4 * g + 3] - 2.3f;
      // (the real epilogue stores the four values right before their statistics: a 16-byte global store in flight behind the sequence)
      *reinterpret_cast<v4f *>(out + 768 * 256 + ((size_t)(blockIdx.x * 256 + tid) * 4 + g) * 4) = v4f{fx, fy, fz, fw};
      v2f xy = {fx, fy}, yz = {fy, fz};
      const v2f pp = {P, P};
      v2f dxy, dyz, sq, t46, t50, acc46;
      float dw, want;
      // (dx, dy) = (x, y) - P ; (dy, dz) = (y, z) - P ; (dx^2, dy^2) packed ; then the cross-half product  lo = dy * dy  read as  a.lo * b.hi
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dxy) : "v"(xy), "v"(pp));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dyz) : "v"(yz), "v"(pp));
      asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(dxy));
      dw = fw - P;
      asm volatile("v_pk_add_f32 %0, %1, %1 op_sel_hi:[0,1]" : "=v"(t46) : "v"(dxy));                        // hi = dx + dy (lo junk), as %bb.246
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(acc46) : "v"(dyz), "v"(dxy));   // THE instruction: lo = dyz.lo * dxy.hi = dy * dy
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t50) : "v"(dyz), "v"(v2f{dxy.y, dw}));                      // (the packed add that overwrites the sources' neighbours right behind it)
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want) : "v"(dyz.x), "v"(dxy.y));
      const float lo = acc46.x;
      if (__builtin_bit_cast(unsigned, lo) != __builtin_bit_cast(unsigned, want)) { ++bad; blo = lo; bwant = want; ba = dyz.x; bb = dxy.y; }
      s1 += (dxy.x + dxy.y) + (dyz.y + dw) + t46.y * 0.f + t50.y * 0.f;
      s2 += (sq.x + lo) + (dyz.y * dyz.y + dw * dw);
    }
    if (bad) {
      const int i = atomicAdd(nrec, 1);
      if (i < 4096) rec[i] = Rec{launch, (int)blockIdx.x, wave, lane, blo, bwant, ba, bb};
    }
    bad_total += bad;
    // feed the statistics back into the accumulators so that nothing is dead and the operands keep changing
    acc[0] += 1e-20f * s1; acc[1] += 1e-20f * s2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] *= 1e-3f; acc_lo[i] *= 1e-3f; }   // (keep the accumulators finite over the rounds)
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i] + acc_lo[i];
  out[blockIdx.x * 256 + tid] = s + bad_total;
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000, taps = argc > 2 ? atoi(argv[2]) : 96, rounds = argc > 3 ? atoi(argv[3]) : 4;
  float *w, *out; Rec *rec; int *nrec;
  hipMalloc(&w, 1 << 26); hipMalloc(&out, 768 * 256 * 4 * 17); hipMalloc(&rec, 4096 * sizeof(Rec)); hipMalloc(&nrec, 4);
  std::vector<_Float16> hw((1 << 26) / 2);   // (the DMA source as small fp16 values)
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.002f * (float)((i * 2654435761u >> 20) & 255) - 0.25f);
  hipMemcpy(w, hw.data(), 1 << 26, hipMemcpyHostToDevice);
  hipMemset(nrec, 0, 4);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k, dim3(768), dim3(256), 40960, 0, w, out, rec, nrec, l, taps, rounds);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
  int n = 0; hipMemcpy(&n, nrec, 4, hipMemcpyDeviceToHost);
  std::vector<Rec> h(n < 4096 ? n : 4096);
  if (!h.empty()) hipMemcpy(h.data(), rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
  printf("%d launches x 768 workgroups x %d rounds (taps %d): %d lane-rounds with a wrong cross-half product\n", launches, rounds, taps, n);
  for (size_t i = 0; i < h.size() && i < 40; ++i)
    printf("  launch %d workgroup %d (slot %d) wave %d lane %d: lo %.9g want %.9g (a.lo %.9g b.hi %.9g)\n", h[i].launch, h[i].wg, h[i].wg / 256, h[i].wave, h[i].lane, h[i].lo, h[i].want, h[i].a, h[i].b);
  return 0;
}
