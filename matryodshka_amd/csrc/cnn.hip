// K2: the encoder-decoder CNN of nets.msi_coord_train_net / nets.msi_train_net
// (reference nets.py:471-515 / 387-450) as implicit-GEMM convolutions on the
// gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fma chain).
//
// Design rule (measured, tools/ubench/mfma_valu_overlap.hip): on gfx950 the fp32-input MFMA
// runs on the SIMD's fp32 ALUs -- MFMA and VALU work do NOT overlap, neither inside one wave
// (8 MFMA + 128 fma = 6.3 ms vs 4.4 + 2.6 alone) nor between two waves of one SIMD.  Every
// VALU instruction in the k-loop is therefore paid for in matrix throughput, so the k-loop of
// this kernel contains no VALU at all:
//   * both GEMM operands go HBM/L2 -> LDS by buffer DMA (`buffer_load_dwordx4 ... lds`):
//     no VGPR staging, no ds_write; the per-lane offsets are constant over a (tap, source)
//     segment and the channel walk is a scalar soffset;
//   * zero padding, M-tile tails and channel tails are lanes whose offset is out of range of the
//     buffer descriptor: the hardware zero-fills those LDS slots (probed: tools/ubench/dma_oob.hip);
//   * the LDS image is linear per row (128 B = 32 channels) with the 16-byte slot XOR-swizzled by
//     (row>>1)&7 -- applied to the per-lane SOURCE offset for A and baked into the packed
//     weights for B -- which makes the ds_read_b128 operand fetch bank-conflict free;
//   * the operand fetch is inline asm: eight address VGPRs, the stage offset is the ds_read
//     immediate, all eight reads of a k-step are issued before its MFMAs (counted lgkmcnt waits);
//   * a 2-stage LDS ring (32 KB => five workgroups per CU), one barrier per k-step, the next
//     k-step's DMA issued after the first MFMA quarter.
// Consequence: the producer's LayerNorm + ReLU can no longer be applied in the operand loader;
// it is applied once, in place, by ln_apply_kernel (HBM-bound, ~1 read + 1 write per activation)
// instead of 9 x Cout/BN times in VALU.
//
// One kernel template serves every layer:
//   * conv3x3 (stride 1/2, rate 1/2, SAME-zero or wrap padding), the 1x1 head,
//     and conv-transpose 4x4 s2 as four output-parity sub-convolutions of 2x2
//     taps each (a class index in the 1-D grid);
//   * GEMM view: M = pixels of one sample, N = Cout, K = taps x Cin, walked in
//     k-steps of 32 channels of one tap; two sources = skip concat by descriptor pair;
//   * CoordNet's |sin(lat)| channel (nets.py:260-265) is constant along W and independent of the image:
//     its share of the convolution is a host-built table [out row][column border class][Cout] that the
//     epilogue adds to the accumulators (no extra k-step);
//   * the accumulators are kept TRANSPOSED (weights are the MFMA's row operand): a lane owns one
//     pixel and 16 channels in four runs of four, so the epilogue stores 16-byte pieces straight
//     from registers (no LDS staging, no barrier) and takes the LayerNorm sums from the same registers;
//   * LayerNorm statistics: every wave adds its (sum x, sum x^2) -- formed about a wave-uniform pivot in fp32,
//     completed in fp64 -- to 64 sharded FIXED-POINT accumulators with integer atomics: integer addition is
//     associative, so the totals are bit-identical whatever the arrival order, and the consumer
//     (ln_apply_kernel, or the 1x1 head while loading) derives mean / variance from 64 x 4 words instead of
//     merging thousands of per-workgroup partials;
//   * bf16 operands (BF16 = 1): the same 128-byte rows hold 64 channels, v_mfma_f32_32x32x16_bf16.
//
// Tiling: 256 threads = 4 wavefronts (2x2), wave tile (BM/2)x(BN/2) of 32x32 MFMA tiles, BK=32;
// work decomposition ("tail split": the tiles of the partial last wave are cut along K inside the launch, the
// last arriving workgroup of a tile sums the partial accumulators) and the measured alternatives: DESIGN.md 4.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "msi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BK = 32;
constexpr int ROW_BYTES = BK * 4;   // one LDS row = 32 channels of one GEMM row
#ifndef MSI_NSTAGE
#define MSI_NSTAGE 2
#endif
constexpr int NSTAGE = MSI_NSTAGE;  // LDS ring depth: NSTAGE-1 k-steps of DMA in flight.  Measured (r01): 2 beats 3 and 4
                                     // (3.29 / 3.41 / 3.80 ms per frame): LDS-limited occupancy matters more than prefetch depth
constexpr int NPAD_ALIGN = 128;
constexpr int COORD_CLASSES = 5;    // column border classes of the CoordNet table: 0,1 | interior | W-2,W-1
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;  // per-lane offset that is out of range of every descriptor (device code)
constexpr int DEFAULT_CUS = 256;  // MI355X; the plan queries hipDeviceProp.multiProcessorCount (option MSI_NET_OPT_NUM_CUS overrides)
constexpr int MAX_SPLIT = 8;
constexpr int CONV_SLOTS_PER_CU = 5;   // 64x64 workgroups (32 KB of LDS each) resident per CU
constexpr double LN_EPS = 1e-12;  // slim.layer_norm variance epsilon [TF-knowledge]
// LayerNorm sums: [sample][LN_SHARDS][LN_WORDS] signed 64-bit fixed point {sum x * S1, sum x^2 * S2}: integer addition is
// associative, so the totals do not depend on the arrival order (bitwise reproducible); a wave's share is rounded to one
// unit (1 / S1, 1 / S2).  S1 = 2^(24 - e), S2 = 2^(16 - 2 e) with a PER-LAYER exponent e = round(log2(expected rms of the
// layer's raw output)) that the host derives from the weights at pack time (ln_scale_exponent: sqrt(K) * rms(w) * rms of a
// LayerNorm + ReLU'd input) and stores in the packed blob: LayerNorm removes any weight scale, so the fixed-point window
// has to follow it.  About e the window is the one measured in r02: fp32-grade statistics for an rms within
// [0.03, 3000] x 2^e on the largest layer (|sum x| < 5e11 / S1', sum x^2 < 1.4e14 / S2' per sample; a wave's scaled share
// below 2^51).  Outside it the result is NOT silently wrong: a share beyond the range sets MSI_NET_STATUS_LN_OVERFLOW, a
// total of sum x^2 below ~1e6 sqrt(waves) units (variance resolved to fewer than six digits) sets
// MSI_NET_STATUS_LN_UNDERFLOW in the status word of the forward's workspace (msi_net_plan_status).
// (Until r02 the sums were exact, hi * 2^-8 + lo * 2^-52 in two words each.  Measured, 6 interleaved repeats of the
// network: this form 2.456 ms; exact with the same cheap rounding, four atomics per wave 2.474 ms; exact with the four
// waves' shares combined through LDS, four atomics per workgroup 2.472 ms.)
constexpr int LN_SHARDS = 64, LN_WORDS = 2;
constexpr int LN_S1_BITS = 24, LN_S2_BITS = 16;   // S1 = 2^(24 - e), S2 = 2^(16 - 2 e)
constexpr int LN_SCL_DOUBLES = 4;                 // per layer in the packed blob: S1, S2, 1 / S1, 1 / S2
constexpr double LN_UNDERFLOW_UNITS_SQ = 1e12;    // (1e6 units)^2 per contributing wave, see ln_mean_inv
enum { STATUS_APPLY_AHEAD_TIMEOUT = 1, STATUS_LN_OVERFLOW = 2, STATUS_LN_UNDERFLOW = 4, STATUS_F16_SPLIT_RANGE = 8 };
constexpr int AP_FLAG_STRIDE = 16;  // ints between two row counters of the apply-ahead hand-off: one counter per 64-byte line
constexpr int HEAD_MAX_C = 256;   // the head's fused LayerNorm keeps scale | shift of its source in LDS

enum { MODE_CONV = 0, MODE_CONVT = 1, MODE_HEAD = 2 };

struct ConvParams {
  // element-typed buffers (fp32, or bf16 in the BF16 instantiation) are addressed in bytes
  const char *x0, *x1;       // NHWC sources, already normalised (x1 = second half of a skip concat)
  const char *wpk;           // packed weights [nclass][ksteps][npad][128 B], slots pre-swizzled
  const char *wpk_x3;        // conv_halo_x3_kernel: [tap][chunk][plane][npad][64 B] bf16 parts of the fp32 weights (see HaloGeomX3)
  const float *coord_bias;   // CoordNet: contribution of the |sin(lat)| channel, [Mh][COORD_CLASSES][cb_stride] fp32, or null
  int cb_stride;
  const double *ln_scl;      // fixed-point scales of THIS layer's LayerNorm sums {S1, S2, 1 / S1, 1 / S2} (packed blob)
  const double *ln_scl_src, *ln_scl_src1;   // ... of the source layers whose sums ln_sums / ln_sums1 (ap_sums) are
  int *status;               // the plan's status word (STATUS_* bits, zeroed per forward)
  const long long *ln_sums;  // head, fp32 only: the LayerNorm sums of the source layer; its affine (+ ReLU) is applied while
                             // loading (the source buffer then holds the RAW conv output); null = source already normalised
  const float *ln_gamma, *ln_beta;   // ... with the source layer's gamma / beta
  double ln_inv_n;           // ... and 1 / (elements per sample)
  const long long *ln_sums1; // convt_halo_kernel: the same for source 1 (the skip half of the concat)
  const float *ln_gamma1, *ln_beta1;
  double ln_inv_n1;
  int halo_apply;            // convt_halo_kernel: bit s = source s is RAW, apply its LayerNorm + ReLU while staging the patch
  const float *bias;         // head only
  float *y;                  // raw output NHWC [B,Hout,Wout,Cout]
  long long *sums;           // LayerNorm sums of THIS layer [B][LN_SHARDS][4] (zeroed per forward), or null
  float *partial;            // [split tiles][split][BM*BN] partial accumulators (register order, see dump_acc)
  int *tile_cnt;             // [split tiles] arrival tickets of the in-launch fix-up (zeroed per forward), or null
  int tiles_m, tiles_n, ntiles;  // output tiles per (sample, class) and in the whole launch
  int n_main, split0, split; // the first n_main tiles are cut into split0 K-ranges each (1 = whole), the rest into split
  int nb_main;               // n_main * split0: workgroups of the first group
  int C0, C1;
  int Hin, Win, Hout, Wout, Cout, npad;
  int Mh, Mw;                // GEMM row grid per sample (output grid; input grid for convT)
  unsigned mg_mw, mg_tm, mg_tn, mg_nc, mg_sp0, mg_sp;  // udiv_magic multipliers of Mw, tiles_m, tiles_n, nclass, split0, split
  int ntaps, cpt0, cpt1, ksteps;  // taps, 32-channel chunks per tap of each source, total k-steps
  int stride, rate, pad_t, pad_l;
  int mode, wrap, nclass;
  int halo_tx;               // halo-patch layers (conv_halo_kernel): spatial 4 x 16 tiles, halo_tx = W / 16 tiles per row;
  unsigned mg_htx;           // 0 = the M tiles are 64 consecutive pixels (conv_igemm_kernel)
  int halo_xor;              // 8 (conv_halo_kernel) / 0: odd rows of a halo tile map lane l to column (l & 15) ^ halo_xor (HaloGeom)
  // "apply-ahead": the first n_apply workgroups of the launch normalise source 0 (LayerNorm + ReLU of the producer layer)
  // while the tile workgroups behind them already compute; see apply_ahead() below.  n_apply = 0: source 0 is
  // normalised already (separate ln_apply launch, or the network input).
  float *ap_x;               // raw fp32 output of the producer, normalised in place (fp32 path) ...
  unsigned short *ap_yb;     // ... or written as bf16 into the operand copy (bf16 path), else null
  const long long *ap_sums;  // the producer's LayerNorm sums [B][LN_SHARDS][4]
  const float *ap_gamma, *ap_beta;
  float *ap_aff;             // published affine [B][scale | shift] (tests)
  int *ap_flags;             // [B][Hin][AP_FLAG_STRIDE] completed units per input row (zeroed per forward)
  int *ap_err;               // set to 1 if a tile workgroup gave up waiting (never in a healthy launch)
  double ap_inv_n;
  int n_apply, ap_units_per_row, ap_unit_vec, ap_row_vec;   // workgroups; units per row; float4 per unit / per row
#if defined(MSI_CONV_TIMING) || defined(MSI_DEBUG_STATS)
  unsigned long long *dbg;   // [block][6]: s_memtime at start, loop start, loop end, end; HW_ID; XCC_ID (tools/conv_timing.py)
#endif
};

// Input offset (rows or columns) of tap-row / tap-column variant v.
// wrapt (conv-transpose of msi_train_net only): the reference runs conv2d_transpose(wrap_pad(x, 2, 2), VALID) and
// LayerNorm + ReLU over its FULL (2H+10) x (2W+10) output before cropping [5:-5] (nets.py:423-435), so the border
// enters the statistics.  The GEMM rows of a parity class then cover the whole non-zero part of that output:
// row (mh, mw), mh in [0, H], mw in [0, W+4]  <->  full output (2 (mh + 2) + ph, 2 mw + pw); tap v uses kernel index
// parity + 2 v and input row mh - v (zero outside [0, H)), padded input column mw - v (valid in [0, W+4), i.e. image
// column (mw - v - 2) mod W).  Rows 0..3 and 2H+6..2H+9 of the full output are exactly zero and only enter the count.
template <int MODE>
__device__ __forceinline__ int tap_delta(int v, int parity, int rate, bool wrapt) {
  if (MODE == MODE_CONV) return v * rate;
  // conv-transpose (SAME), y[2i + k - 1] += x[i] w[k]: even outputs use k=1 (i = o/2) and k=3 (i = o/2 - 1),
  // odd outputs k=2 (i = (o-1)/2) and k=0 (i = (o+1)/2).
  if (MODE == MODE_CONVT) return wrapt ? -v : (v == 0 ? 0 : (parity ? 1 : -1));
  return 0;
}

// x / d by multiply-high with mg = floor(2^32 / d) (0xffffffff for d = 1) and one correction step:
// exact for every 32-bit x; on wave-uniform values this is two scalar multiplies instead of the
// ~35-instruction division sequence.
__device__ __forceinline__ unsigned udiv_magic(unsigned x, unsigned d, unsigned mg) {
  unsigned q = __umulhi(x, mg);
  if (x - q * d >= d) ++q;
  return q;
}

// tanh of the 1x1 head (nets.py:509-515) as (e^{2|x|} - 1) / (e^{2|x|} + 1) on the hardware exp2 / rcp (1 ulp each): absolute error
// 2.0e-7 over [-20, 20] (measured against fp64 on 2^24 points: tools/ubench/tanh_err.hip; the gate is 1e-3), 8 VALU
// instead of the ~35 of the library routine -- the fused tail runs sixteen of them per lane on two of its four waves, which,
// with the integer divisions of its index arithmetic, made that HBM-bound kernel VALU-bound.  Used by BOTH head paths (fused tail
// and stand-alone head), which therefore stay bit-identical to each other.
__device__ __forceinline__ float msi_tanh(float x) {
  const float xa = fminf(fabsf(x), 15.0f);                              // tanh(15) = 1 - 2e-13: 1.0f in fp32
  const float t = __builtin_amdgcn_exp2f(xa * 2.8853900817779268f);     // e^(2 |x|)
  const float r = (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f);
  return x != x ? x : __builtin_copysignf(r, x);
}

__device__ __forceinline__ int coord_class(int mw, int Mw) {
  return mw < 2 ? mw : (mw >= Mw - 2 ? 3 + (mw - (Mw - 2)) : 2);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS operand fetch / wait as inline asm: hipcc schedules builtin LDS loads for minimum register
// pressure (fetch a quarter, wait lgkmcnt(0), 4 MFMAs, fetch the next quarter ...) and re-adds the
// stage offset per read with VALU.  Here the order is the source order, the stage / sub-tile offset
// is the instruction's immediate, and the waits are counted (LDS reads return in order; any other
// lgkm operation in flight only makes a counted wait more conservative).
template <int OFF>
__device__ __forceinline__ v4f lds_read128(unsigned addr) {
  v4f v;
  if constexpr (OFF < 65536) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  } else {  // beyond the 16-bit immediate (only the experimental big tiles): one VALU add
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + (OFF & ~0xffff)), "n"(OFF & 0xffff) : "memory");
  }
  return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(v4f &x, v4f &y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N) : "memory");
}
// ... tying EVERY fragment register of the quarter to the wait: the ds_reads are asm, so the compiler places a consumer
// anywhere after the asm that defines its operands -- an MFMA whose operands are not operands of the wait may be (and
// was: the first MFMA of a k-step of the MT = NT = 2 tiles) scheduled above it and read registers the LDS has not
// written yet (no hardware interlock on lgkmcnt: rare, timing-dependent garbage in one accumulator tile).
template <int N, int MT, int NT>
__device__ __forceinline__ void wait_lgkm_frag(v4f (&a)[MT], v4f (&b)[NT]) {
  static_assert((MT == 1 || MT == 2 || MT == 4) && (NT == 1 || NT == 2), "fragment shapes of the conv kernels");
  if constexpr (MT == 1 && NT == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(N) : "memory");
  else if constexpr (MT == 2 && NT == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(N) : "memory");
  else if constexpr (MT == 1 && NT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
  else if constexpr (MT == 2 && NT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
  else if constexpr (MT == 4 && NT == 1) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]) : "n"(N) : "memory");
  else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}

// ---- shared device helpers of the epilogue ----------------------------------------------------
// Sum over the 64 lanes, returned to every lane (wave-uniform), fixed order.  DPP moves (quad swaps, row mirrors, the
// gfx9 row broadcasts) instead of __shfl_xor: that compiles to ds_bpermute_b32, five dependent trips through the LDS
// crossbar per sum (~600 cycles of latency in every tile's epilogue; the epilogue's length is what keeps a workgroup
// slot away from the k-loop).
__device__ __forceinline__ float wave_sum(float x) {
#define MSI_DPP_ADD(CTRL, ROWMASK)                                                                                     \
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xf, true))
  MSI_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
  MSI_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
  MSI_DPP_ADD(0x141, 0xf);   // row_half_mirror
  MSI_DPP_ADD(0x140, 0xf);   // row_mirror: every lane holds its 16-lane row's sum
  MSI_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1 and 3 add the row before them
  MSI_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2 and 3 add rows 0 + 1
#undef MSI_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// The same for a double (two 32-bit DPP moves + one v_add_f64 per step).
__device__ __forceinline__ double wave_sum_f64(double x) {
#define MSI_DPP_ADD64(CTRL, ROWMASK)                                                                                   \
  {                                                                                                                    \
    const long long b_ = __builtin_bit_cast(long long, x);                                                             \
    const int lo_ = __builtin_amdgcn_update_dpp(0, (int)b_, CTRL, ROWMASK, 0xf, true);                                 \
    const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(b_ >> 32), CTRL, ROWMASK, 0xf, true);                         \
    x += __builtin_bit_cast(double, ((long long)hi_ << 32) | (unsigned)lo_);                                           \
  }
  MSI_DPP_ADD64(0xB1, 0xf) MSI_DPP_ADD64(0x4E, 0xf) MSI_DPP_ADD64(0x141, 0xf) MSI_DPP_ADD64(0x140, 0xf)
  MSI_DPP_ADD64(0x142, 0xa) MSI_DPP_ADD64(0x143, 0xc)
#undef MSI_DPP_ADD64
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// One wave's share of a LayerNorm sum as a fixed-point integer atomic (no return value).  x_scaled = S * scale with
// |x_scaled| < 2^51: adding 1.5 * 2^52 leaves round-to-nearest-even(x_scaled) in the low mantissa bits -- two VALU
// instead of the emulated fp64 -> int64 conversion.
__device__ __forceinline__ void ln_atomic_add(long long *dst, double x_scaled, int *status) {
  const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
  if (!(fabs(x_scaled) < 2251799813685248.0 /* 2^51 */)) {   // (also NaN / inf): outside the fixed-point window
    if (status) __hip_atomic_fetch_or(status, STATUS_LN_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double t = x_scaled + MAGIC;
  const long long v = __builtin_bit_cast(long long, t) - __builtin_bit_cast(long long, MAGIC);
  __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mean and 1 / sqrt(var + eps) of one sample from its LN_SHARDS x LN_WORDS fixed-point sums -> s_stat[0..1] (LDS).
// Called by all 256 threads (ends with a barrier); wave 0 adds the shards (integers: exact, any order).
// This sits at the head of every consumer workgroup / ln_apply block, one wave per SIMD with nobody to hide a dependent
// instruction behind (tools/conv_timing.py: ~20 cycles per dependent step next to a neighbour's MFMAs), so it is written for few
// STEPS: DPP reductions (below), and (r03) 1 / sqrt as v_rsq_f64 + two Newton steps -- 8 dependent fp64 operations, within 2 ulp of
// the ~50-instruction sqrt + division sequence and far inside the float it is rounded to.  (The 64 shards through two LDS integer
// atomics instead of the DPP trees: measured, the prologue of conv3_2 went from 9.8 k to 27 k cycles.)
// (ln_shard_load + ln_mean_inv_pre: the same with the lane's shard requested earlier -- at kernel entry, under the index arithmetic)
struct LnShard { long long w0, w1; };
__device__ __forceinline__ LnShard ln_shard_load(const long long *sums, int tid) {
  LnShard r = {0, 0};
  if (tid < 64) {
    const long long *s = sums + (size_t)tid * LN_WORDS;
    r.w0 = s[0]; r.w1 = s[1];
  }
  return r;
}
template <bool PRE>
__device__ __forceinline__ void ln_mean_inv_impl(const long long *sums, LnShard pre, double inv_n, const double *scl, int *status, double *s_stat, int tid) {
  static_assert(LN_SHARDS == 64, "one shard per lane of wave 0");
  const double inv_s1 = scl[2], inv_s2 = scl[3];   // (uniform address: scalar loads, issued before the shards')
  if (tid < 64) {
    // the 64 shards as doubles (|shard| < 2^63: rounding at 2^-53 relative, far below the 2^-24 / 2^-16 units) through
    // the DPP reduction: no dependent trips through the LDS crossbar at the head of every consumer workgroup / ln_apply block
    long long w0 = pre.w0, w1 = pre.w1;
    if constexpr (!PRE) {
      const long long *s = sums + (size_t)tid * LN_WORDS;
      w0 = s[0]; w1 = s[1];
    }
    const double h1 = wave_sum_f64((double)w0), h2 = wave_sum_f64((double)w1);
    if (tid == 0) {
      const double S1 = h1 * inv_s1, S2 = h2 * inv_s2;
      // resolution: every wave's share is rounded to one unit, so the total carries ~0.5 sqrt(waves) units of rounding
      // noise; below ~1e6 sqrt(waves) units of sum x^2 the variance is resolved to less than six digits
      if (status && h2 * h2 < LN_UNDERFLOW_UNITS_SQ * (1.0 / (inv_n * 1024.0) + 1.0))
        __hip_atomic_fetch_or(status, STATUS_LN_UNDERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double mu = S1 * inv_n;
      double var = S2 * inv_n - mu * mu;
      var = var > 0.0 ? var : 0.0;
      const double x = var + LN_EPS, hx = 0.5 * x;
      double r = __builtin_amdgcn_rsq(x);
      r = r * (1.5 - hx * r * r);
      r = r * (1.5 - hx * r * r);
      s_stat[0] = mu;
      s_stat[1] = r;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void ln_mean_inv(const long long *sums, double inv_n, const double *scl, int *status, double *s_stat, int tid) {
  ln_mean_inv_impl<false>(sums, LnShard{0, 0}, inv_n, scl, status, s_stat, tid);
}
__device__ __forceinline__ void ln_mean_inv_pre(LnShard pre, double inv_n, const double *scl, int *status, double *s_stat, int tid) {
  ln_mean_inv_impl<true>(nullptr, pre, inv_n, scl, status, s_stat, tid);
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Partial accumulators of a split tile travel in REGISTER order: piece ((i*NT + j)*4 + g) of thread tid at
// 16-byte slot (piece * 256 + tid) of the slab -- 1 KB contiguous per wave instruction, no LDS staging.
// In-launch hand-off of K-range partial sums (tail split): a K-range workgroup stores its slab (sc1: written through), waits for the
// stores, takes a ticket; the last arriver reads every slab with sc1 loads.  Experiment knobs (r04, see DESIGN.md section 4 "wrapt"):
// MSI_HANDOFF_FENCE bit 0 = an agent-scope release fence (buffer_wbl2 sc1) before the ticket, bit 1 = an acquire fence (buffer_inv sc1)
// behind it -- measured 728 -> 427 frames/s at configs[1], not the default; MSI_HANDOFF_AUX = cache policy of the slab stores / loads.
#ifndef MSI_HANDOFF_FENCE
#define MSI_HANDOFF_FENCE 0
#endif
#ifndef MSI_HANDOFF_AUX   // cache policy of the slab stores / loads: 16 = sc1 (agent scope), 17 = sc0 | sc1 (system scope)
#define MSI_HANDOFF_AUX 16
#endif
__device__ __forceinline__ void handoff_release() {
  if (MSI_HANDOFF_FENCE & 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__device__ __forceinline__ void handoff_acquire() {
  if (MSI_HANDOFF_FENCE & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// aux = 16 (sc1): write-through store / L1-bypassing load, the in-launch hand-off form (cdna_hip_programming.md).
template <int MT, int NT, int AUX>
__device__ __forceinline__ void dump_acc(const f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int tid) {
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4f v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc,
                                               (unsigned)((((i * NT + j) * 4 + g) * 256 + tid) * 16), 0, AUX);
      }
}

// acc = slab 0 + slab 1 + ... + slab nsp-1, in ascending k whoever calls (deterministic)
template <int MT, int NT, int AUX>
__device__ __forceinline__ void sum_slabs(f32x16 (&acc)[MT][NT], __amdgpu_buffer_rsrc_t rsrc, int nsp, int slab_bytes, int tid) {
  for (int s = 0; s < nsp; ++s) {
    v4f t[MT][NT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          t[i][j][g] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(
              rsrc, (unsigned)((((i * NT + j) * 4 + g) * 256 + tid) * 16), s * slab_bytes, AUX));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (s == 0) {
            acc[i][j][4 * g] = t[i][j][g].x; acc[i][j][4 * g + 1] = t[i][j][g].y;
            acc[i][j][4 * g + 2] = t[i][j][g].z; acc[i][j][4 * g + 3] = t[i][j][g].w;
          } else {
            acc[i][j][4 * g] += t[i][j][g].x; acc[i][j][4 * g + 1] += t[i][j][g].y;
            acc[i][j][4 * g + 2] += t[i][j][g].z; acc[i][j][4 * g + 3] += t[i][j][g].w;
          }
        }
  }
}

// ---- epilogue of one finished tile: CoordNet table, bias + tanh (head), store, LayerNorm sums -----
// Transposed accumulators (C/D layout of v_mfma_f32_32x32x2_f32 / _32x32x16_bf16 with the weights as row
// operand): lane -> pixel (lane & 31) of the wave's 32-pixel block i; register r = 4g + e -> channel
// 32 j + 8 g + 4 (lane >> 5) + e.  A lane therefore stores four 16-byte pieces per (i, j) straight from
// registers (the two half-waves complete 32-byte runs, the four g a 128-byte line), adds the CoordNet
// table -- the |sin(lat)| channel does not depend on the input, so its part of the convolution is a
// host-built table indexed by (output row, column border class, channel) instead of a 33rd k-step -- and
// accumulates the LayerNorm sums of what it stores.
// Statistics: d = x - pivot with a wave-uniform sample pivot (no cancellation: |d| ~ sigma), s1 = sum d,
// s2 = sum d^2 in fp32 over the wave's 1024 values, then sum x = n P + s1, sum x^2 = s2 + 2 P s1 + n P^2 in
// fp64 and a fixed-point integer atomic add (ln_atomic_add).
// INTERIOR: whole tile inside the output, no row / channel masks anywhere (the common case; epilogue VALU
// is paid in matrix throughput of the co-resident workgroups).
// RAW16 (the layers of a bf16 plan): the raw output is stored as fp16 of x * 2^-e, e = the exponent of the layer's
// LayerNorm window (S1 = 2^(24 - e): the value the packer expects the output's rms to be near, so the fp16 range sits
// around it) -- half the bytes of the fp32 raw outputs that bound the bf16 layers, 11 significand bits against the 8 of
// the bf16 operand it becomes after the affine (measured on the oracle: mean |bf16 path - fp32 oracle| + 0.3 %; a bf16
// raw output would be + 19 %).  The statistics are taken from the fp32 accumulators as before.
constexpr int EPI_STAGE_BYTES = 48 * 1024;   // emit_whole_tile's staging strips (four waves x MT x 32 pixels x (row + 16 bytes)): what a caller that stages must own
#ifndef MSI_EPI_ABLATE   // timing experiments only: 1 no stores, 2 no statistics atomics, 4 no statistics arithmetic
#define MSI_EPI_ABLATE 0
#endif
#ifdef MSI_CONV_TIMING
#define MSI_STAMP(k) { if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime(); }
#else
#define MSI_STAMP(k)
#endif
// Whole tiles of the conv / conv-transpose layers (r03).  What the epilogue costs is neither its instruction count nor its bytes but
// its DEPENDENT steps and its write REQUESTS (tools/conv_timing.py --bf16 stamps every workgroup's phases; before: 13-15 k cycles per
// tile, a fifth to a third of a workgroup's life, ~50 cycles per VALU instruction in the element-wise form with a uniform branch
// between 4-value groups, and 32 requests of 16 bytes per store instruction):
//  * phases of MT x NT x 8 independent packed two-float instructions over the WHOLE tile: y = x 2^-e (RAW16; exact), fp16
//    conversion, ... , d = y - P, s1 += d, s2 += d d (four accumulator pairs each) -- one wave per SIMD and workgroup has nobody to
//    hide a dependent instruction behind, and across waves a SIMD does not overlap VALU with the neighbour's MFMAs
//    (tools/ubench/mfma_valu_overlap.hip: split-waves time >= the sum);
//  * stage != nullptr (the halo kernels: LDS is free once the k-loop's last barrier is behind): the wave's MT x 32 pixels x 32 NT
//    channels go through a wave-private LDS strip and leave as 16-byte pieces of whole pixel rows -- a store instruction covers
//    64 / NP pixels x (NP x 16 contiguous bytes) instead of 32 pixels x 16 (32) bytes, through one buffer descriptor per sample with
//    a lane offset and scalar (row, column) steps (no 64-bit address arithmetic per store).
// RAW16 statistics are taken in the scaled unit (the same numbers times a power of two: sum y 2^24 = sum x S1, sum y^2 2^16 =
// sum x^2 S2).
template <int BM, int BN, int MODE, int RAW16, bool CB, bool STAGED, int WR>
__device__ __forceinline__ void emit_whole_tile(const ConvParams &p, f32x16 (&acc)[BM / (32 * WR)][BN / 64], int tile_m, int tile_n, int cls,
                                                int b, int tid, const v4f (&cb_pre)[4], bool use_pre, float pivot, float raw_mul,
                                                double scl_s1, double scl_s2, char *stage) {
  constexpr int MT = BM / (32 * WR), NT = BN / 64, NG = NT * 4, YSZ = RAW16 ? 2 : 4;   // WR x 2 waves
  constexpr int ROWB = NT * 32 * YSZ, PITCH = ROWB + 16, NP = ROWB / 16, PPI = 64 / NP, NRD = 32 / PPI;
  constexpr bool FITS = 2 * WR * MT * 32 * PITCH <= EPI_STAGE_BYTES;
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, half = lane >> 5;
  const int ph = cls >> 1, pw = cls & 1;
  const int mtot = p.Mh * p.Mw;
  const bool want_stats = p.sums != nullptr;
  const float pv_s = pivot * raw_mul;
  const v2f rm = {raw_mul, raw_mul}, pv = {pv_s, pv_s};
  const int nbw = tile_n * BN + wn * (NT * 32), nb0 = nbw + 4 * half;
  const size_t sample_bytes = (size_t)(MODE == MODE_CONVT ? p.Hout * p.Wout : mtot) * p.Cout * YSZ;
  constexpr bool staged = STAGED;
  static_assert(!STAGED || FITS, "staging strips");
  const int tyi = p.halo_tx ? (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx) : 0;
  const int txi = tile_m - tyi * p.halo_tx;
  MSI_STAMP(16)
  // ---- the lane's own pixels (coord-bias rows; direct stores) ----
  char *yp[MT];
  const float *cbp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    yp[i] = nullptr; cbp[i] = nullptr;
    if (CB || !staged) {
      int m = tile_m * BM + wm * (MT * 32) + i * 32 + (lane & 31);
      if (p.halo_tx) {   // (BM / 16) x 16 spatial tile: local pixel = 16 * row + column
        const int local = wm * (MT * 32) + i * 32 + (lane & 31);
        m = (tyi * (BM / 16) + (local >> 4)) * p.Mw + txi * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
      }
      int mh = 0, mw = 0;
      if (MODE == MODE_CONVT || CB) {
        mh = (int)udiv_magic((unsigned)m, (unsigned)p.Mw, p.mg_mw);
        mw = m - mh * p.Mw;
      }
      const size_t opix = MODE == MODE_CONVT ? ((size_t)b * p.Hout + (2 * mh + ph)) * p.Wout + (2 * mw + pw) : (size_t)b * mtot + m;
      yp[i] = reinterpret_cast<char *>(p.y) + (opix * p.Cout + nb0) * YSZ;
      if (CB) cbp[i] = p.coord_bias + (size_t)(mh * COORD_CLASSES + coord_class(mw, p.Mw)) * p.cb_stride + nb0;
    }
  }
  // ---- per 32-pixel block: values (+ coord bias), scale, LayerNorm sums, conversion, LDS strip / direct stores ----
  char *wst = stage + wave * (MT * 32 * PITCH);
  // the lane's pixel -> its slot of the 32-pixel block (row, true column)
  const int slot = p.halo_tx ? ((lane & 16) | ((lane & 15) ^ (((lane >> 4) & 1) * p.halo_xor))) : (lane & 31);
  char *wp = wst + slot * PITCH + half * (4 * YSZ);
  v2f s1v[4] = {}, s2v[4] = {};
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    v2f ya[NG], yc[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      ya[q] = v2f{acc[i][q >> 2][4 * (q & 3)], acc[i][q >> 2][4 * (q & 3) + 1]};
      yc[q] = v2f{acc[i][q >> 2][4 * (q & 3) + 2], acc[i][q >> 2][4 * (q & 3) + 3]};
    }
    if (CB) {
      v4f cb[NG];
#pragma unroll
      for (int q = 0; q < NG; ++q)
        cb[q] = (MT == 1 && NT == 1 && use_pre) ? cb_pre[q & 3] : *reinterpret_cast<const v4f *>(cbp[i] + (q >> 2) * 32 + 8 * (q & 3));
#pragma unroll
      for (int q = 0; q < NG; ++q) { ya[q] += v2f{cb[q].x, cb[q].y}; yc[q] += v2f{cb[q].z, cb[q].w}; }
    }
    if (RAW16) {
#pragma unroll
      for (int q = 0; q < NG; ++q) { ya[q] *= rm; yc[q] *= rm; }
    }
    u2_t hw[NG];
    if (RAW16) {
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const h2_t lo = {(_Float16)ya[q].x, (_Float16)ya[q].y}, hi = {(_Float16)yc[q].x, (_Float16)yc[q].y};
        hw[q] = u2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      }
    }
    if (!(MSI_EPI_ABLATE & 1)) {
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        char *dst = staged ? wp + i * (32 * PITCH) : yp[i];
        if (RAW16) *reinterpret_cast<u2_t *>(dst + ((q >> 2) * 32 + 8 * (q & 3)) * 2) = hw[q];
        else *reinterpret_cast<v4f *>(dst + ((q >> 2) * 32 + 8 * (q & 3)) * 4) = v4f{ya[q].x, ya[q].y, yc[q].x, yc[q].y};
      }
    }
    if (want_stats && !(MSI_EPI_ABLATE & 4)) {
#pragma unroll
      for (int q = 0; q < NG; ++q) { ya[q] -= pv; yc[q] -= pv; }
#pragma unroll
      for (int q = 0; q < NG; ++q) { s1v[q & 3] += ya[q]; s1v[q & 3] += yc[q]; }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        s2v[q & 3] = __builtin_elementwise_fma(ya[q], ya[q], s2v[q & 3]);
        s2v[q & 3] = __builtin_elementwise_fma(yc[q], yc[q], s2v[q & 3]);
      }
    }
  }
  MSI_STAMP(21)
  const v2f t1 = (s1v[0] + s1v[1]) + (s1v[2] + s1v[3]), t2 = (s2v[0] + s2v[1]) + (s2v[2] + s2v[3]);
  float s1 = t1.x + t1.y, s2 = t2.x + t2.y;
  // ---- the strip's pieces -> memory ----
  if (staged && !(MSI_EPI_ABLATE & 1)) {
    const char *rp = wst + (lane / NP) * PITCH + (lane % NP) * 16;
    v4f pc[MT][NRD];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < NRD; ++k) pc[i][k] = *reinterpret_cast<const v4f *>(rp + (i * 32 + k * PPI) * PITCH);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.y) + (size_t)b * sample_bytes, 0, (int)(unsigned)sample_bytes, 0x00020000);
    const int rowb = p.Cout * YSZ;                      // bytes per output pixel
    int pix0, rowstep, colstep;                        // the lane's first pixel; what one tile row / column is in output pixels
    if (p.halo_tx) {
      const int r0 = tyi * (BM / 16) + wm * (MT * 2), c0 = txi * 16 + lane / NP;
      if (MODE == MODE_CONVT) { pix0 = (2 * r0 + ph) * p.Wout + 2 * c0 + pw; rowstep = 2 * p.Wout; colstep = 2; }
      else { pix0 = r0 * p.Mw + c0; rowstep = p.Mw; colstep = 1; }
    } else {
      pix0 = tile_m * BM + wm * (MT * 32) + lane / NP; rowstep = 16; colstep = 1;   // (linear pixels: a "row" is 16 of them)
    }
    const unsigned v0 = (unsigned)pix0 * (unsigned)rowb + (unsigned)(nbw * YSZ + (lane % NP) * 16);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int ro = 2 * i + ((k * PPI) >> 4), co = (k * PPI) & 15;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pc[i][k]), rsrc_y, v0, (ro * rowstep + co * colstep) * rowb, 0);
      }
  }
  MSI_STAMP(17)
  if (want_stats) {
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    MSI_STAMP(18)
    if (lane == 0 && !(MSI_EPI_ABLATE & 2)) {
      const double P = (double)pv_s, n = (double)(MT * NT * 16 * 64), a = (double)s1;
      const double u1 = RAW16 ? 16777216.0 : scl_s1, u2 = RAW16 ? 65536.0 : scl_s2;
      // RAW16: the tile was just stored as fp16 of y = x 2^-e, which is +-inf beyond 65504.  s2 = sum (y - pivot)^2 over the wave
      // bounds every |y - pivot|: above 32752^2 a stored value MAY have left the fp16 range (or the layer is > 1000 x the scale
      // its weights predict) -- reported like a LayerNorm sum that left its window (no silent inf -> NaN pixels)
      if (RAW16 && !(s2 <= 1.0727e9f)) __hip_atomic_fetch_or(p.status, STATUS_LN_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long *dst = p.sums + ((size_t)b * LN_SHARDS + ((blockIdx.x * 4 + wave) & (LN_SHARDS - 1))) * LN_WORDS;   // (any spread will do)
      ln_atomic_add(dst, (n * P + a) * u1, p.status);
      ln_atomic_add(dst + 1, ((double)s2 + 2.0 * P * a + n * P * P) * u2, p.status);
    }
    MSI_STAMP(19)
  }
}

template <int BM, int BN, int MODE, bool INTERIOR, int RAW16, int WR>
__device__ __forceinline__ void emit_tile_impl(const ConvParams &p, f32x16 (&acc)[BM / (32 * WR)][BN / 64], int tile_m,
                                               int tile_n, int cls, int b, int tid, const v4f (&cb_pre)[4], bool use_pre, char *stage,
                                               float raw_mul_pre) {
  constexpr int MT = BM / (32 * WR), NT = BN / 64;   // WR x 2 waves, 32 MT x 32 NT each
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, half = lane >> 5;
  const int ph = cls >> 1, pw = cls & 1;
  const int mtot = p.Mh * p.Mw;
  const bool wrapt = MODE == MODE_CONVT && p.wrap != 0;
  const bool vec_ok = (p.Cout & 3) == 0;
  const bool has_cb = MODE == MODE_CONV && p.coord_bias != nullptr;
  const bool want_stats = MODE != MODE_HEAD && p.sums != nullptr;
  const float pivot = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0][0][0])));
  // (scalar loads at the head of the epilogue: inside the lane-0 branch below they would be vector loads with a memory
  // round trip between the wave reduction and the atomics, at the end of every tile)
  double scl_s1 = 0.0, scl_s2 = 0.0;
  // (a whole RAW16 tile needs nothing but 2^-e -- its sums are taken in the scaled unit -- and the caller may have it already)
  const bool have_pre = INTERIOR && RAW16 && MODE != MODE_HEAD && raw_mul_pre > 0.f;
  if ((want_stats || RAW16) && !have_pre) { scl_s1 = p.ln_scl[0]; scl_s2 = p.ln_scl[1]; }
  const float raw_mul = have_pre ? raw_mul_pre : RAW16 ? (float)(scl_s1 * (1.0 / 16777216.0)) : 1.f;   // 2^-e
  constexpr int YSZ = RAW16 ? 2 : 4;                                          // bytes per stored element
  float s1 = 0.f, s2 = 0.f, cnt = 0.f;
  if constexpr (INTERIOR && MODE != MODE_HEAD) {
    // (staged stores: the caller owns EPI_STAGE_BYTES of free LDS, 32-bit offsets reach the sample, the pixel steps are uniform)
    constexpr bool FITS = 2 * WR * MT * 32 * (NT * 32 * (RAW16 ? 2 : 4) + 16) <= EPI_STAGE_BYTES;
    const size_t sample_bytes = (size_t)(MODE == MODE_CONVT ? p.Hout * p.Wout : p.Mh * p.Mw) * p.Cout * (RAW16 ? 2 : 4);
    const bool staged = FITS && stage != nullptr && sample_bytes < 0xfffffff0ull && (p.halo_tx != 0 || MODE == MODE_CONV);
    if constexpr (FITS) {
      if (staged) {
        if (has_cb) emit_whole_tile<BM, BN, MODE, RAW16, true, true, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, pivot, raw_mul, scl_s1, scl_s2, stage);
        else emit_whole_tile<BM, BN, MODE, RAW16, false, true, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, pivot, raw_mul, scl_s1, scl_s2, stage);
        return;
      }
    }
    if (has_cb) emit_whole_tile<BM, BN, MODE, RAW16, true, false, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, pivot, raw_mul, scl_s1, scl_s2, stage);
    else emit_whole_tile<BM, BN, MODE, RAW16, false, false, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, pivot, raw_mul, scl_s1, scl_s2, stage);
    return;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = tile_m * BM + wm * (MT * 32) + i * 32 + (lane & 31);
    if ((MODE == MODE_CONV || MODE == MODE_CONVT) && p.halo_tx) {   // (BM / 16) x 16 spatial tile: local pixel = 16 * row + column
      const int local = wm * (MT * 32) + i * 32 + (lane & 31);
      const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
      m = (tyi * (BM / 16) + (local >> 4)) * p.Mw + (tile_m - tyi * p.halo_tx) * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
      // (ragged grids -- msi_train_net's conv-transposes, (H + 1) x (W + 5) GEMM rows: a column beyond the row's end is no pixel)
      if (!INTERIOR && (tile_m - tyi * p.halo_tx) * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor)) >= p.Mw) m = mtot;
    }
    const bool mok = INTERIOR || m < mtot;
    int mh = 0, mw = 0;
    if (MODE == MODE_CONVT || has_cb) {
      mh = (int)udiv_magic((unsigned)m, (unsigned)p.Mw, p.mg_mw);
      mw = m - mh * p.Mw;
    }
    size_t opix;
    bool sok = mok;   // stored (wrapt: computed for the statistics, stored only inside the [5:-5] crop)
    if (MODE == MODE_CONVT) {
      int orow = 2 * mh + ph, ocol = 2 * mw + pw;
      if (wrapt) {
        orow -= 1; ocol -= 5;
        sok = mok && orow >= 0 && orow < p.Hout && ocol >= 0 && ocol < p.Wout;
      }
      opix = ((size_t)b * p.Hout + orow) * p.Wout + ocol;
    } else {
      opix = (size_t)b * mtot + m;
    }
    const float *cbrow = has_cb ? p.coord_bias + (size_t)(mh * COORD_CLASSES + coord_class(mw, p.Mw)) * p.cb_stride : nullptr;
    char *yrow = reinterpret_cast<char *>(p.y) + opix * p.Cout * YSZ;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = tile_n * BN + wn * (NT * 32) + j * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        const bool ok = mok && (INTERIOR || n < p.Cout);
        v4f v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (ok) {
          if (has_cb) {   // cb_stride is Cout rounded up to 4: the whole float4 is in range
            // (64x64 fp32 tiles: requested before the k-loop by load_coord_bias)
            const v4f cb = (MT == 1 && NT == 1 && use_pre) ? cb_pre[g] : *reinterpret_cast<const v4f *>(cbrow + n);
            v.x += cb.x; v.y += cb.y; v.z += cb.z; v.w += cb.w;
          }
          if (MODE == MODE_HEAD) {   // (the packed bias is padded to a multiple of 4 as well)
            const v4f bs = *reinterpret_cast<const v4f *>(p.bias + n);
            v.x = msi_tanh(v.x + bs.x); v.y = msi_tanh(v.y + bs.y); v.z = msi_tanh(v.z + bs.z); v.w = msi_tanh(v.w + bs.w);
          }
          if (MSI_EPI_ABLATE & 1) {
          } else if (sok && RAW16) {   // (Cout % 4 == 0 in a bf16 plan: whole 8-byte pieces)
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u2_t __attribute__((ext_vector_type(2)));
            const h2_t lo = {(_Float16)(v.x * raw_mul), (_Float16)(v.y * raw_mul)}, hi = {(_Float16)(v.z * raw_mul), (_Float16)(v.w * raw_mul)};
            *reinterpret_cast<u2_t *>(yrow + n * 2) = u2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
          } else if (sok) {
            float *dst = reinterpret_cast<float *>(yrow) + n;
            if (vec_ok) {
              *reinterpret_cast<v4f *>(dst) = v;
            } else {
              dst[0] = v.x;
              if (n + 1 < p.Cout) dst[1] = v.y;
              if (n + 2 < p.Cout) dst[2] = v.z;
              if (n + 3 < p.Cout) dst[3] = v.w;
            }
          }
          if (want_stats && !(MSI_EPI_ABLATE & 4)) {
            const float dx = v.x - pivot, dy = v.y - pivot, dz = v.z - pivot, dw = v.w - pivot;
            if (INTERIOR || n + 3 < p.Cout) {
              s1 += (dx + dy) + (dz + dw);
              s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
              cnt += 4.f;
            } else {   // channel tail inside the last float4
              s1 += dx; s2 += dx * dx; cnt += 1.f;
              if (n + 1 < p.Cout) { s1 += dy; s2 += dy * dy; cnt += 1.f; }
              if (n + 2 < p.Cout) { s1 += dz; s2 += dz * dz; cnt += 1.f; }
            }
          }
        }
      }
    }
  }
  if (want_stats) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float wcnt = INTERIOR ? (float)(MT * NT * 16 * 64) : wave_sum(cnt);
    if (lane == 0 && wcnt > 0.f && !(MSI_EPI_ABLATE & 2)) {
      const double P = (double)pivot, n = (double)wcnt, a = (double)s1;
      long long *dst = p.sums + ((size_t)b * LN_SHARDS + ((blockIdx.x * 4 + wave) & (LN_SHARDS - 1))) * LN_WORDS;   // (any spread will do)
      // (RAW16, see emit_whole_tile: here the sums are in x, the stored value is x raw_mul)
      if (RAW16 && !(s2 * raw_mul * raw_mul <= 1.0727e9f)) __hip_atomic_fetch_or(p.status, STATUS_LN_OVERFLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ln_atomic_add(dst, (n * P + a) * scl_s1, p.status);
      ln_atomic_add(dst + 1, ((double)s2 + 2.0 * P * a + n * P * P) * scl_s2, p.status);
#if defined(MSI_DEBUG_STATS)   // (debug: every wave's share, for run-to-run comparison -- tools/conv_timing.py area + 16384 * 24)
      if (p.dbg) {
        unsigned long long *o = p.dbg + 16384 * 24 + ((size_t)((blockIdx.x + gridDim.x * blockIdx.y) * 2 + (cls & 1)) * 4 + wave) * 4;
        o[0] = ((unsigned long long)__builtin_bit_cast(unsigned, s2) << 32) | __builtin_bit_cast(unsigned, s1);
        o[1] = ((unsigned long long)__builtin_bit_cast(unsigned, pivot) << 32) | __builtin_bit_cast(unsigned, wcnt);
        o[2] = __builtin_bit_cast(unsigned long long, (n * P + a) * scl_s1);
        o[3] = __builtin_bit_cast(unsigned long long, ((double)s2 + 2.0 * P * a + n * P * P) * scl_s2);
      }
#endif
    }
  }
}

template <int BM, int BN, int MODE, int RAW16 = 0, int WR = 2>
__device__ __forceinline__ void emit_tile(const ConvParams &p, f32x16 (&acc)[BM / (32 * WR)][BN / 64], int tile_m, int tile_n,
                                          int cls, int b, int tid, const v4f (&cb_pre)[4], bool use_pre, char *stage = nullptr,
                                          float raw_mul_pre = 0.f) {
  const bool interior = !(MODE == MODE_CONVT && p.wrap != 0) && (tile_m + 1) * BM <= p.Mh * p.Mw &&
                        (tile_n + 1) * BN <= p.Cout && (p.Cout & 3) == 0;
  if (interior) emit_tile_impl<BM, BN, MODE, true, RAW16, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, stage, raw_mul_pre);
  else emit_tile_impl<BM, BN, MODE, false, RAW16, WR>(p, acc, tile_m, tile_n, cls, b, tid, cb_pre, use_pre, nullptr, 0.f);
}
template <int BM, int BN, int MODE, int RAW16 = 0, int WR = 2>
__device__ __forceinline__ void emit_tile(const ConvParams &p, f32x16 (&acc)[BM / (32 * WR)][BN / 64], int tile_m, int tile_n,
                                          int cls, int b, int tid, char *stage = nullptr, float raw_mul_pre = 0.f) {
  const v4f none[4] = {};
  emit_tile<BM, BN, MODE, RAW16, WR>(p, acc, tile_m, tile_n, cls, b, tid, none, false, stage, raw_mul_pre);
}

// The CoordNet table values of this lane's pixel and 16 channels (64x64 tile, transposed accumulator layout), requested
// BEFORE the k-loop and parked in 16 VGPRs: four loads whose round trip would otherwise open every tile's epilogue (the
// epilogue's latency keeps a workgroup slot away from the k-loop).  Out-of-range pixels / channels are clamped (their
// values are never used).  Zeros without CoordNet.
__device__ __forceinline__ void load_coord_bias(const ConvParams &p, int tile_m, int tile_n, int tid, v4f (&cbv)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) cbv[g] = v4f{0.f, 0.f, 0.f, 0.f};
  if (p.coord_bias == nullptr) return;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, half = lane >> 5;
  int m = tile_m * 64 + wm * 32 + (lane & 31);
  if (p.halo_tx) {
    const int local = wm * 32 + (lane & 31);
    const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
    m = (tyi * 4 + (local >> 4)) * p.Mw + (tile_m - tyi * p.halo_tx) * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
  }
  m = min(m, p.Mh * p.Mw - 1);
  const int mh = (int)udiv_magic((unsigned)m, (unsigned)p.Mw, p.mg_mw), mw = m - mh * p.Mw;
  const float *cbrow = p.coord_bias + (size_t)(mh * COORD_CLASSES + coord_class(mw, p.Mw)) * p.cb_stride;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = min(tile_n * 64 + wn * 32 + 4 * half + 8 * g, p.cb_stride - 4);
    cbv[g] = *reinterpret_cast<const v4f *>(cbrow + n);
  }
}


#ifdef MSI_EXPERIMENTS   // measured-slower variants (apply-ahead, fp32 128x64 / 64x128 tiles) are compiled only on request:
// MSI_CNN_DEFINES=-DMSI_EXPERIMENTS python -m matryodshka_amd.build --force; the default library has no kernel with spills
// ---- apply-ahead: LayerNorm + ReLU of the producer inside the consumer's launch ------------------------------
// The LayerNorm of layer N needs all of layer N (global statistics), so it cannot be folded into N's epilogue, and the
// k-loop of layer N+1 has no VALU slot for it; as a launch of its own it is an HBM-bound pass (read + write every
// activation: 0.15 ms of a 2.7 ms frame) during which the matrix pipes idle, plus a kernel boundary per layer.
// Here the first n_apply workgroups of layer N+1's launch do that pass -- row by row, in place, publishing a counter per
// input row -- and every tile workgroup waits only for the input rows its halo touches: the HBM-bound pass overlaps
// the MFMA-bound one.  The unit sequence is dealt out like the tiles (XCD x sweeps the x-th eighth of the rows, in
// order), so the rows a tile workgroup needs first are normalised first, by workgroups of its own XCD.
// Hand-off (cdna_hip_programming.md, write-through form): the apply workgroups read the raw values with sc1 loads (the
// raw lines never enter an L1) and write the normalised ones with sc1 stores (write-through), drain vmcnt, barrier,
// one relaxed agent-scope atomic per unit; a tile workgroup polls the counters of its rows with relaxed agent-scope
// loads and only then issues its first DMA -- no line of the activation is fetched by anyone before it is final, so
// no cache holds a stale copy.  Dead-lock freedom: the apply workgroups have the lowest block indices, never wait,
// and are all resident before any tile workgroup can occupy their slots; the wait is bounded anyway (ap_err).
__device__ __forceinline__ void apply_ahead(const ConvParams &p, char *smem, int tid) {
  float *s_aff = reinterpret_cast<float *>(smem);                 // scale[C0] | shift[C0]
  double *s_stat = reinterpret_cast<double *>(smem + 2 * 512 * 4 + 64);
  const int C = p.C0;
  const int upr = p.ap_units_per_row;
  const long units_per_sample = (long)p.Hin * upr;
  // batch is not a kernel parameter: the grid covers ntiles = tiles per sample * batch
  const int batch = p.ntiles / (p.tiles_m * p.tiles_n * p.nclass);
  const long total = units_per_sample * batch;
  const long per = (total + 7) / 8;                                // units of one XCD's range
  const int x = blockIdx.x & 7;
  const __amdgpu_buffer_rsrc_t rs_aff = __builtin_amdgcn_make_buffer_rsrc((void *)p.ap_aff, 0, 0x7fffffff, 0x00020000);
  (void)rs_aff;
  int cur_b = -1;
  for (long l = blockIdx.x >> 3; l < per; l += p.n_apply >> 3) {
    const long u = (long)x * per + l;
    if (u >= total) break;
    const int b = (int)(u / units_per_sample);
    const long ur = u - (long)b * units_per_sample;
    const int row = (int)(ur / upr), part = (int)(ur - (long)row * upr);
    if (b != cur_b) {   // (the sweep is in order: the sample changes at most a few times per workgroup)
      __syncthreads();
      ln_mean_inv(p.ap_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ap_inv_n, p.ln_scl_src, p.status, s_stat, tid);
      const double mu = s_stat[0], inv = s_stat[1];
      for (int c = tid; c < C; c += 256) {
        const double sc = inv * (double)p.ap_gamma[c];
        const float fs = (float)sc, ft = (float)((double)p.ap_beta[c] - mu * sc);
        s_aff[c] = fs;
        s_aff[C + c] = ft;
        if (row == 0 && part == 0) {   // exactly one workgroup per sample starts at its first unit
          p.ap_aff[(size_t)b * 2 * C + c] = fs;
          p.ap_aff[(size_t)b * 2 * C + C + c] = ft;
        }
      }
      __syncthreads();
      cur_b = b;
    }
    const size_t row_elems = (size_t)p.ap_row_vec * 4;
    const size_t base = ((size_t)b * p.Hin + row) * row_elems;    // element offset of the row
    const int v0 = part * p.ap_unit_vec;
    const int v1 = min(v0 + p.ap_unit_vec, p.ap_row_vec);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.ap_x + base), 0, (int)(row_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = p.ap_yb ? __builtin_amdgcn_make_buffer_rsrc((void *)(p.ap_yb + base), 0, (int)(row_elems * 2), 0x00020000) : rs;
    auto bf16_bits = [](float f) __attribute__((always_inline)) -> unsigned {
      const unsigned uu = __builtin_bit_cast(unsigned, f);
      return (uu + 0x7fffu + ((uu >> 16) & 1u)) >> 16;
    };
    for (int v = v0 + tid; v < v1; v += 4 * 256) {
      v4f xv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)   // out-of-range offsets read zeros and are not stored
        xv[k] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((v + 256 * k) * 16), 0, 16));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int vv = v + 256 * k;
        if (vv >= v1) break;
        const int c = (vv * 4) % C;                               // C % 4 == 0: a float4 never straddles channels' wrap
        const v4f s4 = *reinterpret_cast<const v4f *>(s_aff + c), t4 = *reinterpret_cast<const v4f *>(s_aff + C + c);
        v4f y;
        y.x = fmaxf(xv[k].x * s4.x + t4.x, 0.f); y.y = fmaxf(xv[k].y * s4.y + t4.y, 0.f);
        y.z = fmaxf(xv[k].z * s4.z + t4.z, 0.f); y.w = fmaxf(xv[k].w * s4.w + t4.w, 0.f);
        if (p.ap_yb) {
          typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
          const v2u_t o = {bf16_bits(y.x) | (bf16_bits(y.y) << 16), bf16_bits(y.z) | (bf16_bits(y.w) << 16)};
          __builtin_amdgcn_raw_buffer_store_b64(o, rd, (unsigned)(vv * 8), 0, 16);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, y), rs, (unsigned)(vv * 16), 0, 16);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(p.ap_flags + ((size_t)b * p.Hin + row) * AP_FLAG_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A tile workgroup's side, in two steps so that the round trip of the counter loads hides behind the prologue:
// rows_probe (right after the tile decode) -- lane l loads the counter of input row r0 + l once;
// rows_wait (before the first DMA) -- all done: nothing more; else ONE lane polls the missing rows, last row first
// (the sweep is in row order), one counter per 64-byte line: thousands of lanes polling a few shared lines starve the
// apply workgroups' own counter updates (measured: +30 % on every layer).
__device__ __forceinline__ int rows_probe(const ConvParams &p, int b, int r0, int r1, int tid) {
  const int r = r0 + tid;
  if (r > r1) return 0x7fffffff;
  return __hip_atomic_load(p.ap_flags + ((size_t)b * p.Hin + r) * AP_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void rows_wait(const ConvParams &p, int b, int r0, int r1, int probed, int tid, int *s_flag) {
  const bool ready = probed >= p.ap_units_per_row;
  if (tid < 64) {   // rows of one tile fit one wave's lanes (host-checked: <= 64 input rows per tile)
    const bool all = __builtin_amdgcn_ballot_w64(!ready) == 0;
    if (tid == 0) *s_flag = all ? 1 : 0;
  }
  __syncthreads();
  if (*s_flag) return;
  if (tid == 0) {
    int spins = 0;
    for (int r = r1; r >= r0; --r) {
      const int *f = p.ap_flags + ((size_t)b * p.Hin + r) * AP_FLAG_STRIDE;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.ap_units_per_row) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1 << 20)) { __hip_atomic_fetch_or(p.ap_err, STATUS_APPLY_AHEAD_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r = r0; break; }   // ~1 s: something is badly wrong; do not hang the GPU
      }
    }
  }
  __syncthreads();
}

#endif  // MSI_EXPERIMENTS

// amdgpu_waves_per_eu: with a dynamic LDS size hipcc cannot see that five 32 KB workgroups share a CU
// and spends registers freely (116 for the 64x64 tile => four waves per SIMD); five need <= 96.
// BF16 = 0: fp32 operands, 32 channels per k-step, v_mfma_f32_32x32x2_f32 (16 per k-step and wave);
// BF16 = 1: bf16 operands (fp32 accumulate, fp32 raw output), 64 channels per k-step -- the same 128-byte
// rows, swizzle and DMA -- and v_mfma_f32_32x32x16_bf16 (4 per k-step and wave).
template <int BM, int BN, int MODE, int BF16>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM * BN >= 128 * 128 ? 2 : (BM * BN > 64 * 64 || NSTAGE > 2 ? 3 : 5))))
conv_igemm_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (the body uses device-only types)
  constexpr int ESZ = BF16 ? 2 : 4;          // bytes per operand element
  constexpr int BKE = ROW_BYTES / ESZ;       // channels per k-step: 32 (fp32) / 64 (bf16)
  constexpr int MT = BM / 64, NT = BN / 64;  // 32x32 MFMA tiles per wave (2x2 waves)
  constexpr int AI = BM / 32, BI = BN / 32;  // DMA wave-instructions (8 rows x 128 B each) per wave per k-step
  constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work decomposition: "tail split" ----------------------------------------------------------
  // All output tiles of a layer are co-resident (five workgroups fit a CU), so the launch takes as
  // long as the busiest CU: with T tiles on 256 CUs that is ceil(T/256) tile-times although the
  // average is T/256 (800 tiles: 4 vs 3.125 -> 78 %).  The first n_main = 256*floor(T/256) tiles are
  // therefore computed whole (every CU gets the same number), and each of the remaining tiles is cut
  // into `split` K-ranges computed by separate, short workgroups of the SAME launch whose partial
  // accumulators conv_fixup_kernel sums in k order.
  // XCD-aware order for the whole tiles: workgroup b runs on XCD b % 8 (observed, speed only) and
  // each XCD has a private L2; consecutive tiles share halo rows and weights, so every XCD gets a
  // CONTIGUOUS range of tiles instead of every eighth one (bijective remap).
#ifdef MSI_EXPERIMENTS
  if (p.n_apply > 0 && (int)blockIdx.x < p.n_apply) {   // apply-ahead workgroup (see apply_ahead)
    apply_ahead(p, smem, tid);
    return;
  }
#endif
  const int S = p.ksteps;
  int t, k0 = 0, k1 = S, ks = 0, slot = 0;   // slot: index of this K-range's partial accumulator
  {
#ifdef MSI_EXPERIMENTS
    const int bid = (int)blockIdx.x - p.n_apply;   // (n_apply is a multiple of 8: the XCD of a tile workgroup is still bid % 8)
#else
    const int bid = (int)blockIdx.x;
#endif
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      k0 = (int)udiv_magic((unsigned)(ks * S), (unsigned)sp, mg);
      k1 = (int)udiv_magic((unsigned)((ks + 1) * S), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (k0 == 0) & (k1 == S);
  int tile_m, tile_n, cls, b;
  {
    // tile order: M tiles fastest (measured on the same box: 3.03 ms per frame vs 3.11 ms with N tiles
    // fastest and 3.09 ms for the previous 3-D grid without the tail split)
    // conv-transpose: the parity class is the FASTEST index -- the four classes of an M tile read the same input pixels,
    // and as neighbours in the order they run on the same XCD at about the same time (one fetch into its L2 instead of
    // four through HBM: the bf16 conv-transposes were bound by exactly that traffic)
    int r = t;
    const int q0 = (int)udiv_magic((unsigned)r, (unsigned)p.nclass, p.mg_nc);
    cls = r - q0 * p.nclass; r = q0;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  constexpr bool CB_PRE = BM == 64 && BN == 64 && MODE == MODE_CONV && !BF16;
  v4f cbv[4];
  if constexpr (CB_PRE) load_coord_bias(p, tile_m, tile_n, tid, cbv);   // in flight during the prologue and the k-loop
  const int ph = cls >> 1, pw = cls & 1;
  const int mtot = p.Mh * p.Mw;
  const int wrap_w = p.wrap ? p.Win : 0;
  const bool wrapt = MODE == MODE_CONVT && p.wrap != 0;
  // apply-ahead: source 0 is being normalised by the first workgroups of this launch; the tile needs input rows
  // [ap_r0, ap_r1] of it (source 1, the skip, was normalised by an earlier launch).  Probe their counters now.
#ifdef MSI_EXPERIMENTS
  int ap_r0 = 0, ap_r1 = -1, ap_probe = 0;
  if (p.n_apply > 0) {
    const int m_lo = tile_m * BM, m_hi = min(m_lo + BM, mtot) - 1;
    const int mh_lo = (int)udiv_magic((unsigned)m_lo, (unsigned)p.Mw, p.mg_mw), mh_hi = (int)udiv_magic((unsigned)m_hi, (unsigned)p.Mw, p.mg_mw);
    if (MODE == MODE_CONV) { ap_r0 = mh_lo * p.stride - p.pad_t; ap_r1 = mh_hi * p.stride - p.pad_t + 2 * p.rate; }
    else if (MODE == MODE_CONVT) { ap_r0 = mh_lo - 1; ap_r1 = wrapt ? mh_hi : mh_hi + 1; }
    else { ap_r0 = mh_lo; ap_r1 = mh_hi; }
    ap_r0 = max(ap_r0, 0);
    ap_r1 = min(ap_r1, p.Hin - 1);
    ap_probe = rows_probe(p, b, ap_r0, ap_r1, tid);
  }
#endif

  // ---- DMA lane mapping: instruction i of this wave fills LDS rows [wave*BM/4 + 8i, +8);
  // lane -> (row = lane>>3, 16-byte slot = lane&7); the slot holds data chunk slot ^ ((row>>1)&7).
  const int drow = lane >> 3, dslot = lane & 7;
  // Per A row: everything a (tap, source) segment switch needs, so that the switch itself -- VALU work
  // inside the MFMA loop, paid in matrix throughput -- is ~7 instructions per row: the input row
  // base, the (wrapped) input column for each of the NV tap columns, and one validity bit per tap.
  constexpr int NV = MODE == MODE_CONV ? 3 : (MODE == MODE_CONVT ? 2 : 1);  // tap rows = tap columns
  int rowbase[AI], colw0[AI], colw1[AI], colw2[AI];  // (three arrays: a [AI][NV] array selected by
                                                                           // the tap column ends up indexed in scratch)
  unsigned vmask[AI], a_chunk16[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = wave * (BM / 4) + i * 8 + drow;
    const int m = tile_m * BM + r;
    const int mh = (int)udiv_magic((unsigned)m, (unsigned)p.Mw, p.mg_mw);
    const int mw = m - mh * p.Mw;
    const int ih0 = mh * p.stride - p.pad_t, iw0 = mw * p.stride - p.pad_l;
    rowbase[i] = ih0 * p.Win;
    const bool mok = m < mtot;
    unsigned rowok = 0, colok = 0;
    colw1[i] = colw2[i] = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int ih = ih0 + tap_delta<MODE>(v, ph, p.rate, wrapt);
      const int iwu = iw0 + tap_delta<MODE>(v, pw, p.rate, wrapt);
      const int iw = iwu < 0 ? iwu + wrap_w : (iwu >= p.Win ? iwu - wrap_w : iwu);  // wrap_w = 0: plain zero padding
      if (v == 0) colw0[i] = iw;
      if (v == 1) colw1[i] = iw;
      if (v == 2) colw2[i] = iw;
      if (ih >= 0 && ih < p.Hin) rowok |= 1u << v;
      if (iw >= 0 && iw < p.Win && !(wrapt && (iwu < -2 || iwu >= p.Win + 2))) colok |= 1u << v;
    }
    // bit (vr*NV + vc) = rowok[vr] & colok[vc]: replicate colok into every NV-bit group, keep the groups of valid rows
    unsigned colrep = 0, rowrep = 0;
#pragma unroll
    for (int vr = 0; vr < NV; ++vr) {
      colrep |= colok << (vr * NV);
      rowrep |= ((rowok >> vr) & 1u) * (((1u << NV) - 1u) << (vr * NV));
    }
    const unsigned vm = mok ? (colrep & rowrep) : 0u;
    vmask[i] = vm;
    a_chunk16[i] = (unsigned)((dslot ^ ((r >> 1) & 7)) * 16);  // byte offset of the data chunk this lane fetches
  }
  // B: rows [wave*BN/4 + 8i, +8) of the weight tile; the packed blob is already swizzled
  const unsigned b_voff = (unsigned)((tile_n * BN + wave * (BN / 4) + drow) * ROW_BYTES + dslot * 16);

  const size_t in_pix = (size_t)p.Hin * p.Win;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(p.wpk + (size_t)cls * S * p.npad * ROW_BYTES), 0, (int)((size_t)S * p.npad * ROW_BYTES), 0x00020000);
  const char *src0 = p.x0 + (size_t)b * in_pix * p.C0 * ESZ;
  const long d_src = (p.x1 + (size_t)b * in_pix * p.C1 * ESZ) - src0;  // integer select, see gen below
  const int bytes0 = (int)(in_pix * p.C0 * ESZ), bytes1 = (int)(in_pix * p.C1 * ESZ);

  // ---- k-step generator: (tap, source, chunk) segments ------------------------------------------
  // Per segment the per-lane A offsets are fixed; the channel walk is the scalar soffset.
  unsigned a_voff[AI];       // byte offset of (pixel, data chunk) inside the source, or OOB
  __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)src0, 0, bytes0, 0x00020000);
  const int nreg = p.ntaps * (p.cpt0 + p.cpt1);  // = p.ksteps
  int g_step = k0, g_tap = 0, g_src = 0, g_chunk = 0, g_C = p.C0;
  if (k0 > 0 && k0 < nreg) {   // a K-range of a split tile starts in the middle of the k-step list
    const int cpt = p.cpt0 + p.cpt1;
    g_tap = k0 / cpt;
    const int within = k0 - g_tap * cpt;
    g_src = within >= p.cpt0 ? 1 : 0;
    g_chunk = g_src ? within - p.cpt0 : within;
  }

  // The k-step issue is a macro, not a lambda: a by-reference closure keeps pointers to colw0/1/2 in
  // adjacent fields, hipcc turns the tap-column select into an INDEXED load from the closure, and
  // everything the closure references (the kernel arguments included) then lives in scratch.
  bool g_new = true;  // the per-lane offsets of the current (tap, source) segment are not computed yet
  // A and B tiles of k-step g_step -> LDS stage; then advance the generator by one k-step.
#define MSI_ISSUE(stage)                                                                                  \
  {                                                                                                                              \
    /* A and B tiles of k-step g_step -> LDS stage; then advance the generator by one k-step. */                                 \
    char *sA = smem + (stage) * STAGE_BYTES + wave * (BM / 4) * ROW_BYTES;                                                         \
    char *sB = smem + (stage) * STAGE_BYTES + BM * ROW_BYTES + wave * (BN / 4) * ROW_BYTES;                                        \
    const int soff_b = g_step * p.npad * ROW_BYTES;                                                                              \
    /* the weights first: their addresses need no per-row work, so on a segment switch they are on their way while the */       \
    /* A offsets are recomputed                                                                                           */       \
    /* B rows [wave*BN/4 + 8i, +8): the instruction's immediate offset advances BOTH the source and the LDS address */                                        \
    static_assert(BI <= 4, "B rows per wave: written out for immediate offsets");                                                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB, 16, b_voff, soff_b, 0, 0);                                  \
    if (BI > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB, 16, b_voff, soff_b, 8 * ROW_BYTES, 0);          \
    if (BI > 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB, 16, b_voff, soff_b, 16 * ROW_BYTES, 0);         \
    if (BI > 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB, 16, b_voff, soff_b, 24 * ROW_BYTES, 0);         \
    if (g_step < nreg) {                                                                                                         \
      if (g_new) {                                                                                                               \
        /* segment switch: tap -> (tap row, tap column) variant, all scalar; ~7 VALU per row */                                  \
        g_new = false;                                                                                                           \
        int vr, vc;                                                                                                              \
        if (MODE == MODE_CONV) { vr = g_tap / 3; vc = g_tap - vr * 3; }                                                          \
        else if (MODE == MODE_CONVT) { vr = g_tap >> 1; vc = g_tap & 1; }                                                        \
        else { vr = 0; vc = 0; }                                                                                                 \
        const int srow = tap_delta<MODE>(vr, ph, p.rate, wrapt) * p.Win;                                                         \
        const unsigned bit = 1u << (vr * NV + vc);                                                                               \
        g_C = g_src ? p.C1 : p.C0;                                                                                               \
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(src0 + (g_src ? d_src : 0L)), 0, g_src ? bytes1 : bytes0,            \
                                                   0x00020000);                                                                  \
        const unsigned pix_bytes = (unsigned)g_C * (unsigned)ESZ;                                                                           \
_Pragma("unroll")                                                                                                                \
        for (int i = 0; i < AI; ++i) {                                                                                           \
          int iw = colw0[i];                                                                                                     \
          if (NV > 1) iw = vc == 1 ? colw1[i] : iw;                                                                              \
          if (NV > 2) iw = vc == 2 ? colw2[i] : iw;                                                                              \
          /* pixel index < 2^24 and bytes per pixel < 2^24 (checked on the host): the 24-bit multiply is */                      \
          /* full rate, a 32-bit multiply a quarter */                                                                           \
          const unsigned off = __umul24((unsigned)(rowbase[i] + srow + iw), pix_bytes) + a_chunk16[i];                           \
          a_voff[i] = (vmask[i] & bit) != 0 ? off : OOB;                                                                         \
        }                                                                                                                        \
      }                                                                                                                          \
      const int soff_a = g_chunk * ROW_BYTES;                                                                                    \
      const int cleft = g_C - g_chunk * BKE; /* channels from this chunk on; < BKE only when C % BKE != 0 (wave-uniform) */        \
      if (MODE == MODE_HEAD && !BF16 && p.ln_sums != nullptr) {                                                                  \
        /* fused LayerNorm apply of the producer (head only: two k-steps, HBM-bound -- VALU is free here): the A rows go */      \
        /* through registers, x -> max(x * scale[c] + shift[c], 0), and land in the LDS slots the DMA would have filled  */      \
        const float *aff_ = s_haff;                                                                                              \
_Pragma("unroll")                                                                                                                \
        for (int i = 0; i < AI; ++i) {                                                                                           \
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));                                                            \
          const v4f x = __builtin_bit_cast(v4f, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], soff_a, 0)); \
          const int c0 = g_chunk * BKE + (int)(a_chunk16[i] >> 2);                                                               \
          const v4f s4 = *reinterpret_cast<const v4f *>(aff_ + c0), t4 = *reinterpret_cast<const v4f *>(aff_ + p.C0 + c0);       \
          v4f y;                                                                                                                 \
          y.x = fmaxf(x.x * s4.x + t4.x, 0.f); y.y = fmaxf(x.y * s4.y + t4.y, 0.f);                                              \
          y.z = fmaxf(x.z * s4.z + t4.z, 0.f); y.w = fmaxf(x.w * s4.w + t4.w, 0.f);                                              \
          *reinterpret_cast<v4f *>(sA + i * 8 * ROW_BYTES + lane * 16) = y;                                                      \
        }                                                                                                                        \
      } else if (cleft >= BKE) {                                                                                                 \
_Pragma("unroll")                                                                                                                \
        for (int i = 0; i < AI; ++i)                                                                                             \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void *)(sA + i * 8 * ROW_BYTES), 16, a_voff[i], soff_a, 0, 0);   \
      } else {                                                                                                                   \
        /* channel tail: lanes whose 16-byte chunk starts beyond the source's channels fetch zeros */                            \
_Pragma("unroll")                                                                                                                \
        for (int i = 0; i < AI; ++i)                                                                                             \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void *)(sA + i * 8 * ROW_BYTES), 16,                             \
                                                   a_chunk16[i] < (unsigned)(cleft * ESZ) ? a_voff[i] : OOB, soff_a, 0, 0);         \
      }                                                                                                                          \
    }                                                                                                                            \
    /* advance (wave-uniform scalar state) */                                                                                    \
    ++g_step;                                                                                                                    \
    if (g_step < nreg) {                                                                                                         \
      ++g_chunk;                                                                                                                 \
      const int cpt = g_src ? p.cpt1 : p.cpt0;                                                                                   \
      if (g_chunk == cpt) {                                                                                                      \
        g_chunk = 0;                                                                                                             \
        if (g_src == 0 && p.cpt1 > 0) {                                                                                          \
          g_src = 1;                                                                                                             \
        } else {                                                                                                                 \
          g_src = 0;                                                                                                             \
          ++g_tap;                                                                                                               \
        }                                                                                                                        \
        g_new = true;                                                                                                            \
      }                                                                                                                          \
    }                                                                                                                            \
  }

#ifdef MSI_EXPERIMENTS
  if (p.n_apply > 0) rows_wait(p, b, ap_r0, ap_r1, ap_probe, tid, reinterpret_cast<int *>(smem));   // (LDS is still unused)
#endif

  // fp32 head: the affine of its source's LayerNorm (scale | shift per channel) from the source's sums -> LDS; the
  // k-step issue applies it (+ ReLU) while loading, so the source is read RAW and never normalised in memory
  float *s_haff = nullptr;
  if constexpr (MODE == MODE_HEAD && !BF16) {
    __shared__ __attribute__((aligned(16))) float s_haff_store[2 * HEAD_MAX_C];
    __shared__ double s_hstat[2];
    s_haff = s_haff_store;
    if (p.ln_sums != nullptr) {
      ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_hstat, tid);
      const double mu = s_hstat[0], inv = s_hstat[1];
      for (int c = tid; c < p.C0; c += 256) {
        const double sc = inv * (double)p.ln_gamma[c];
        s_haff_store[c] = (float)sc;
        s_haff_store[p.C0 + c] = (float)((double)p.ln_beta[c] - mu * sc);
      }
      __syncthreads();
    }
  }

  // the first k-step's DMA goes out before the rest of the set-up: its latency (HBM under load) is the
  // longest single wait of the prologue
  const int nsteps = k1 - k0;
  MSI_ISSUE(0)

  // ---- MFMA side: precomputed ds_read addresses (no VALU in the loop) --------------------------
  // lane reads row (lane&31) of its wave tile, k-quarter q of half h = lane>>5: data chunk h*4+q
  // lives in slot (h*4+q) ^ ((row>>1)&7).
  const int frow = lane & 31, fh = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  // Only the eight stage-0 addresses live in VGPRs; stage and sub-tile offsets are ds_read immediates.
  unsigned a_q[4], b_q[4];
  {
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // quarter q of the k-step, lane half fh: fp32 -> channels 16 fh + 4q .. +4 (chunk 4 fh + q, one per
      // four 32x32x2 MFMAs); bf16 -> channels 16q + 8 fh .. +8 (chunk 2q + fh, the A/B fragment of one 32x32x16)
      const int slot = ((BF16 ? 2 * q + fh : fh * 4 + q) ^ fswz) * 16;
      a_q[q] = lds_base + (wm * (MT * 32) + frow) * ROW_BYTES + slot;
      b_q[q] = lds_base + BM * ROW_BYTES + (wn * (NT * 32) + frow) * ROW_BYTES + slot;
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One k-step = fetch (all eight operand quarters, 32 VGPRs per MFMA tile row/column) + mma (16
  // MFMAs per MFMA tile, each quarter waiting only for its own two fetches).  A wave keeps its MFMA
  // stream fed on its own instead of relying on the other waves of the SIMD to cover every ds_read
  // round trip; whatever is placed between fetch and mma (the DMA issue of the next k-step) runs
  // in the shadow of the LDS latency.
  struct Frag { v4f a[4][MT], b[4][NT]; };
#define MSI_FETCH(F, ST)                                                                              \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                 \
      F.a[q_][i_] = i_ == 0 ? lds_read128<(ST) * STAGE_BYTES>(a_q[q_])                                \
                            : lds_read128<(ST) * STAGE_BYTES + (MT - 1) * 32 * ROW_BYTES>(a_q[q_]);    \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                 \
      F.b[q_][j_] = j_ == 0 ? lds_read128<(ST) * STAGE_BYTES>(b_q[q_])                                \
                            : lds_read128<(ST) * STAGE_BYTES + (NT - 1) * 32 * ROW_BYTES>(b_q[q_]);    \
  }
  static_assert(MT <= 2 && NT <= 2, "MSI_FETCH addresses at most two MFMA tiles per direction");
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  auto mma_quarter = [&](Frag &f, const int q) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (BF16) {
          // weights first: D = W^T-tile x pixels, i.e. D row = channel, D column = pixel (transposed accumulators)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[q][j]),
                                                              __builtin_bit_cast(bf16x8, f.a[q][i]), acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[q][j].x, f.a[q][i].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[q][j].y, f.a[q][i].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[q][j].z, f.a[q][i].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[q][j].w, f.a[q][i].w, acc[i][j], 0, 0, 0);
        }
      }
  };
  // Quarter Q of a fetched k-step.  Counted wait: LDS reads return in order, quarter Q needs the
  // first (Q+1)*(MT+NT) of the 4*(MT+NT) fetches.  sched_barrier: the MFMAs are not volatile --
  // without it hipcc hoists all waits above them.
#define MSI_MMA_Q(F, Q)                                                                           \
  wait_lgkm_frag<(3 - (Q)) * (MT + NT), MT, NT>(F.a[Q], F.b[Q]);                                  \
  mma_quarter(F, Q);                                                                              \
  __builtin_amdgcn_sched_barrier(0);

  // ---- main loop: double-buffered LDS, one barrier per k-step ------------------------------------
  //   k-step s:  fetch the operands of s | first MFMA quarter | issue the DMA of s+1 into the other
  //   buffer | remaining quarters | s_waitcnt vmcnt(0) | barrier.
  // Occupancy (five workgroups per CU) covers the barrier.  Measured alternatives, all slower on the
  // BASELINE network: 3- and 4-stage rings, 128x64 / 128x128 tiles, two k-steps per barrier (with and
  // without prefetching the second k-step's operands); the DMA issue before the first MFMA quarter
  // (2.91 ms per frame vs 2.88) or after the second (2.89).
  // Unrolled by two with literal buffer indices (stage offsets are ds_read immediates).
  static_assert(NSTAGE == 2 || NSTAGE == 3, "the main loop is unrolled for a 2- or 3-stage ring");
  constexpr int PD = NSTAGE - 1;          // prefetch distance in k-steps
  constexpr int DMA_PER_STEP = AI + BI;   // buffer_load ... lds instructions per wave per k-step
  if (PD > 1 && nsteps > 1) {
    MSI_ISSUE(1)
    wait_vmcnt<DMA_PER_STEP>();           // k-step 0 landed, k-step 1 in flight
  } else {
    wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the head's fused-LayerNorm path fills its A rows with ds_write)
  __builtin_amdgcn_s_barrier();

  // k-step S in stage U: issue k-step S+PD into the stage freed by the previous barrier; before the
  // closing barrier k-step S+1 must have landed (the youngest PD-1 k-steps may stay in flight).
#define MSI_KSTEP(U, S)                                                                   \
  {                                                                                       \
    Frag f_;                                                                              \
    MSI_FETCH(f_, U)                                                                      \
    MSI_MMA_Q(f_, 0)                                                                      \
    const bool more_ = (S) + PD < nsteps;                                                 \
    if (more_) MSI_ISSUE(((U) + PD) % NSTAGE)                                             \
    MSI_MMA_Q(f_, 1)                                                                      \
    MSI_MMA_Q(f_, 2)                                                                      \
    MSI_MMA_Q(f_, 3)                                                                      \
    if (PD > 1 && more_) wait_vmcnt<(PD - 1) * DMA_PER_STEP>(); else wait_vmcnt<0>();     \
    __builtin_amdgcn_s_barrier();                                                         \
  }
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (int S = 0; S < nsteps; S += NSTAGE) {
    MSI_KSTEP(0, S);
    if (S + 1 >= nsteps) break;
    MSI_KSTEP(1, S + 1);
    if (NSTAGE > 2) {
      if (S + 2 >= nsteps) break;
      MSI_KSTEP(NSTAGE - 1, S + 2);
    }
  }
#undef MSI_KSTEP
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
  auto stamp = [&]() __attribute__((always_inline)) {
    if (p.dbg && tid == 0) {
      unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID: wave, simd, cu, sh, se ...
      o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
    }
  };
#endif
#undef MSI_ISSUE
#undef MSI_MMA_Q
#undef MSI_FETCH

  // ---- epilogue ------------------------------------------------------------------------------
  if (!full) {
    // K-range of a split tile: the raw accumulators go to this range's slab; the LAST of the tile's workgroups to
    // arrive sums the slabs in k order (deterministic whoever is last) and emits the tile.  Hand-off per
    // cdna_hip_programming.md (in-launch split-K, write-through form): sc1 slab stores -> vmcnt(0) -> workgroup
    // barrier -> one lane takes a relaxed agent-scope ticket; the last arriver reads the slabs with sc1 loads.
    // (Plan option MSI_NET_OPT_FIXUP_KERNEL: plain stores here, conv_fixup_kernel as a separate launch.)
    constexpr int SLAB = BM * BN * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.partial + (size_t)slot * (BM * BN)), 0, SLAB, 0x00020000);
    if (p.tile_cnt == nullptr) {
      dump_acc<MT, NT, 0>(acc, rsrc_p, tid);
#ifdef MSI_CONV_TIMING
      stamp();
#endif
      return;
    }
    dump_acc<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_p, tid);
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab stores have left (sc1: written through)
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);        // (all LDS reads of the main loop are behind its last barrier)
    if (tid == 0) {
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*s_old != nsp - 1) {
#ifdef MSI_CONV_TIMING
      stamp();
#endif
      return;
    }
    handoff_acquire();   // (the last arriver reads every slab with sc1 loads)
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.partial + (size_t)(slot - ks) * (BM * BN)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (every thread has read s_old before the epilogue's strips reuse LDS)
  }
  // (LDS is free: the k-loop's last barrier is behind; NSTAGE >= 2 stages hold the strips of every instantiation)
  static_assert((size_t)2 * (BM + BN) * ROW_BYTES >= (size_t)4 * (BM / 64) * 32 * ((BN / 64) * 32 * (BF16 ? 2 : 4) + 16), "staging strips");
  emit_tile<BM, BN, MODE, (BF16 && MODE != MODE_HEAD) ? 1 : 0>(p, acc, tile_m, tile_n, cls, b, tid, cbv, CB_PRE && p.coord_bias != nullptr, smem);
#ifdef MSI_CONV_TIMING
  stamp();
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- halo-patch convolution (stride-1 3x3 layers, fp32) -------------------------------------------------------------
// The tap-DMA kernel above fetches every input pixel nine times (once per tap) from L2 and needs its input normalised in
// memory (the k-loop has no VALU slot for the producer's LayerNorm).  Here a workgroup owns a 4 x 16-pixel SPATIAL tile
// and, per 32-channel chunk of the input, stages the (4 + 2r) x (16 + 2r) halo patch in LDS ONCE: through registers, so
// that the producer's LayerNorm + ReLU is applied on the way -- 2 VALU per element against its 9 taps x 64 output
// channels = 576 MACs (0.4 % of the MFMA time) -- and the nine taps are nine `ds_read` IMMEDIATE offsets into that patch
// (one base address VGPR; pixel stride 144 bytes = 128 + 16 of padding: 16 consecutive pixels of a row cover all 64
// banks exactly once, no swizzle).  The weights stream per tap through a 3-stage DMA ring as before.  A layer whose
// every consumer is a halo layer is never normalised in memory: its ln_apply launch (an HBM round trip of the whole
// activation) disappears.  k order: chunk-major, tap-minor (the packed blob stays tap-major: only the DMA's scalar
// offset changes).  K-ranges of split tiles are cut at chunk boundaries.
template <int RATE>
struct HaloGeom {
  static constexpr int PW = 16 + 2 * RATE, PH = 4 + 2 * RATE, NPX = PW * PH;
  static constexpr int PIX_BYTES = 144;
  // Bank groups of a ds_read_b128 (served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; MI355X_MICROARCH.md):
  // the 144-byte pixel stride spreads 16 consecutive pixels of a row over the 16 groups; lanes 16-31 of an MFMA block read
  // the block's SECOND row.  With a row pitch of 256 n bytes they would be conflict-free as they are (conv_halo_bf16_kernel),
  // but that pitch does not fit four workgroups per CU here; with a pitch of 256 n + 128 bytes they are conflict-free
  // when the second row's lanes take its columns rotated by 8 (lane l <-> column (l & 15) ^ 8: emit_tile's halo_xor).
  // Measured before (pitch PW x 144): SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE.
  static constexpr int ROW_PITCH = ((PW * PIX_BYTES + 127) / 256) * 256 + 128;
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_STAGE = 64 * ROW_BYTES;          // one k-step of weights: 64 output rows x 128 B
  static constexpr int NSTG = 3;                          // 9 taps per chunk = 3 x 3 stages: the stage of a tap is a literal
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;     // float4 patch elements per thread and chunk
};

template <int RATE, int APPLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RATE == 1 ? 4 : 3)))
conv_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeom<RATE> G;
  constexpr int R = RATE, PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = 1, NT = 1;
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work decomposition: as conv_igemm_kernel (tail split), K-ranges in whole chunks ----
  const int CH = p.cpt0;                                  // 32-channel chunks of the input
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;                                               // (nclass = 1)
  }
  LnShard shard = {0, 0};   // (the lane's shard of the source's LayerNorm sums: requested here, reduced after the patch requests)
  if (APPLY) shard = ln_shard_load(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, tid);
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * 4, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win, C = p.C0;
  v4f cbv[4];
  load_coord_bias(p, tile_m, tile_n, tid, cbv);           // in flight during the prologue and the k-loop
  // the first two weight k-steps go out before the patch addresses are worked out (they depend on tile_n and the wave only)
  const int S = p.ksteps;                                 // 9 CH
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)S * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + drow) * ROW_BYTES + dslot * 16);
  {
    char *sB0 = smem + G::A_BYTES + wave * 16 * ROW_BYTES;
    const int so0 = c0 * p.npad * ROW_BYTES, so1 = (CH + c0) * p.npad * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB0, 16, b_voff, so0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB0, 16, b_voff, so0, 8 * ROW_BYTES, 0);
    {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB0 + G::B_STAGE), 16, b_voff, so1, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB0 + G::B_STAGE), 16, b_voff, so1, 8 * ROW_BYTES, 0);
    }
  }

  // ---- per-lane patch elements: e = tid + 256 k -> patch pixel e / 8, 16-byte channel slot e % 8 (= tid % 8) ----
  unsigned voff[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - R + py;
    int iw = ow0 - R + px;
    if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);   // msi_train_net: wrap along W, zeros along H
    pok[k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
    voff[k] = pok[k] ? __umul24((unsigned)(ih * W + iw), (unsigned)(C * 4)) + (unsigned)(cslot * 16) : OOB;
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
  }
  const size_t in_bytes = (size_t)H * W * C * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)in_bytes, 0x00020000);

  // producer's LayerNorm: mean / inv once per workgroup; the per-channel affine per chunk (the lane's four channels)
  float inv_f = 1.f, mu_hi = 0.f, mu_lo = 0.f;
  bool has_pad = false;                                   // wave-uniform: any of the wave's patch pixels is padding
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NLOAD], g4, be4;
  // patch of chunk c -> registers (+ gamma / beta of the lane's channels)
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[k_], (c) * ROW_BYTES, 0)); \
    if (APPLY) {                                                                                                       \
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + (c) * 32 + cslot * 4);                                          \
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + (c) * 32 + cslot * 4);                                          \
    }                                                                                                                  \
  }
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};                                                          \
    if (APPLY) {   /* scale = inv * gamma; shift = beta - mean * scale with the mean as hi + lo floats: fp32 ops only */ \
      s4 = inv_f * g4;                                                                                                 \
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (APPLY) {                                                                                                     \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */          \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;                                    \
    }                                                                                                                  \
  }
  // weights of k-step (chunk c, tap) -> ring stage st; the packed blob is tap-major: row block tap * CH + c
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * ROW_BYTES;                                         \
    const int soff_ = ((tap) * CH + (c)) * p.npad * ROW_BYTES;                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);            \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((2 * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 64);
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * 32 + frow) * ROW_BYTES + (((fh * 4 + q) ^ fswz) << 4);
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  // one k-step = tap TAP of the current chunk, weights in ring stage TAP % 3; the DMA of the k-step two ahead is issued
  // after the first MFMA quarter; before the closing barrier the NEXT k-step's weights must have landed: every VMEM
  // operation issued before them (the next chunk's patch loads, issued in tap 0) completes first (in-order return)
#define MSI_HTAP(TAP)                                                                                                  \
  {                                                                                                                    \
    constexpr int KH_ = (TAP) / 3, KW_ = (TAP) % 3, ST_ = (TAP) % 3;                                                   \
    constexpr int AOFF_ = KH_ * R * G::ROW_PITCH + KW_ * R * G::PIX_BYTES;                                             \
    v4f a_[4], b_[4];                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      a_[q_] = q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 16>(a_base)                        \
             : q_ == 2 ? lds_read128<AOFF_ + 32>(a_base) : lds_read128<AOFF_ + 48>(a_base);                            \
      b_[q_] = lds_read128<ST_ * G::B_STAGE>(b_q[q_]);                                                                 \
    }                                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);                                                                         \
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);                                                                         \
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);                                                                         \
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);                                                                         \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[0][0], 0, 0, 0);                        \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (q_ == 0) {                                                                                                   \
        if ((TAP) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                            \
        /* k-step two ahead: (c, TAP + 2) or (c + 1, TAP - 7) */                                                       \
        if ((TAP) + 2 < 9) { MSI_B_ISSUE(c, (TAP) + 2, ((TAP) + 2) % 3) }                                              \
        else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, (TAP) - 7, ((TAP) + 2) % 3) }                                        \
      }                                                                                                                \
    }                                                                                                                  \
    {                                                                                                                  \
      const bool issued_ = ((TAP) + 2 < 9) || (c + 1 < c1);                                                            \
      if ((TAP) == 0 && c + 1 < c1) wait_vmcnt<2 + NLOAD + (APPLY ? 2 : 0)>();   /* patch loads + this tap's DMA in flight */ \
      else if (issued_) wait_vmcnt<2>();                                                                               \
      else wait_vmcnt<0>();                                                                                            \
    }                                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                                      \
  }

  // ---- prologue: first patch, first two weight k-steps ----
  int c = c0;
  MSI_PATCH_LOAD(c0)                                      // (the weights of k-steps 0 and 1 are on their way already)
  if (APPLY) {   // the sums' round trip rides on the patch's (s_stat sits in the A region: read back before the patch lands)
    double *s_stat = reinterpret_cast<double *>(smem);
    ln_mean_inv_pre(shard, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    const double mu = s_stat[0];
    inv_f = (float)s_stat[1];
    mu_hi = (float)mu;
    mu_lo = (float)(mu - (double)mu_hi);
    __syncthreads();
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    MSI_HTAP(0) MSI_HTAP(1) MSI_HTAP(2) MSI_HTAP(3) MSI_HTAP(4) MSI_HTAP(5) MSI_HTAP(6) MSI_HTAP(7) MSI_HTAP(8)
    if (c + 1 < c1) {   // every wave has read the last tap of this chunk (closing barrier of tap 8): swap the patch
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_HTAP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD

  // ---- epilogue: as conv_igemm_kernel ----
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
  auto stamp = [&]() __attribute__((always_inline)) {
    if (p.dbg && tid == 0) {
      unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
      o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
  };
#endif
  if (!full) {
    constexpr int SLAB = 64 * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)slot * (64 * 64)), 0, SLAB, 0x00020000);
    if (p.tile_cnt == nullptr) {
      dump_acc<MT, NT, 0>(acc, rsrc_p, tid);
#ifdef MSI_CONV_TIMING
      stamp();
#endif
      return;
    }
    dump_acc<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_p, tid);
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)(slot - ks) * (64 * 64)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (every thread has read s_old before the epilogue's strips reuse LDS)
  }
  emit_tile<64, 64, MODE_CONV>(p, acc, tile_m, tile_n, 0, b, tid, cbv, p.coord_bias != nullptr, smem);
#ifdef MSI_CONV_TIMING
  stamp();
#endif
#endif
}

// ---- halo-patch convolution, fp32 through a 3-way bf16 split with six products (round 4; plan option F32_SPLIT3) ----------------
// the native fp32 MFMA (v_mfma_f32_32x32x2_f32, 256 flop per cycle and SIMD) is at 0.82 of its peak and the
// rest is per-visit overhead.  The bf16 MFMA is 16 x faster; an fp32 operand x is EXACTLY h + m + l + (|rest| <= 2^-25 |x|) with bf16
// parts h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even; both differences are exact in fp32), so
//     x w  =  h.h + h.m + m.h + h.l + l.h + m.m  +  (m.l + l.m + l.l + rest terms: <= 2^-26 |x w|, below fp32's own product rounding)
// -- six v_mfma_f32_32x32x16_bf16 (exact products, fp32 accumulation of 16 terms each) per 16 channels: 192 instead of 512 matrix
// cycles.  NOT the 2-way / 3-product split (that is TF32-grade and narrower than the reference's fp32).  The activations stay
// fp32 in memory: the patch is staged through registers as in conv_halo_kernel (the producer's LayerNorm applied on the
// way) and split there -- three v_cvt_pk_bf16_f32 pairs and two exact subtractions per float4 -- into three 64-byte planes per
// pixel (pixel stride 208 B: 13 x 16, odd, so the fragment reads stay conflict-free with the row pitch / column rotation of
// HaloGeom).  The weights are split on the host at pack time (x3 block of the packed blob).  Everything around the k-loop --
// work decomposition, tail split, in-launch hand-off, epilogue, LayerNorm sums -- is conv_halo_kernel's.
// Numerics: oracle emulation of this arithmetic against the fp32 oracle at the configs[1] frame: pred 3.0e-6, rgba 1.9e-6,
// rgb 1.0e-6 max-abs (profiles/r04_split3_numerics.txt: the native fp32 path's own summation-order error is 4.5e-6 on pred).
#ifndef MSI_X3_EARLY_DMA
#define MSI_X3_EARLY_DMA 0
#endif
#ifndef MSI_X3_NSTG   // weight ring of conv_halo_x3_kernel: 0 = by rate -- two stages at rate 1 (47.8 KB of LDS: three workgroups per CU;
                      // measured 84.7 -> 81.2 us per layer against three stages / two workgroups) and three at rate 2 (two workgroups either
                      // way: 92.5 vs 99.7 us); 2 / 3 force it (tuning)
#define MSI_X3_NSTG 0
#endif
#ifndef MSI_X3_ABLATE   // timing experiments only (wrong results): 1 no weight DMA, 4 no per-tap barrier, 8 no fragment reads, 16 no MFMAs, 32 no patch swap
#define MSI_X3_ABLATE 0
#endif
// NPL = 3: x = h + m + l in bf16, six products (F32_SPLIT3).  NPL = 2: x = h + m' 2^-11 in fp16, three products h.h + (h.m' + m'.h) 2^-11
// (F32_SPLIT_F16: 22 significand bits per operand, operands limited to the fp16 RANGE -- the patch store flags |x| > 65504 in the status word)
template <int RATE, int NS = (MSI_X3_NSTG ? MSI_X3_NSTG : (RATE == 1 ? 2 : 3)), int NPL = 3, int TH = 4>
struct HaloGeomX3 {
  static constexpr int PW = 16 + 2 * RATE, PH = TH + 2 * RATE, NPX = PW * PH;   // TH x 16 output pixels per workgroup (TH = 4, or 8: conv_halo8_x3_kernel)
  static constexpr int PIX_BYTES = NPL * 64 + 16;         // NPL planes x 32 two-byte parts + 16 (13 or 9 sixteen-byte slots: odd)
  static constexpr int ROW_PITCH = ((PW * PIX_BYTES + 127) / 256) * 256 + 128;
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_ROW = 64;                        // 32 bf16 channels of one output row and plane
  static constexpr int B_PLANE = 64 * B_ROW, B_STAGE = NPL * B_PLANE;
  static constexpr int NSTG = NS;
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;
};
// fp16-split range tracking (NPL / NP == 2): the largest operand magnitude a lane stored, as the BIT PATTERN of |x| in an unsigned max -- for sign-cleared floats integer
// order is float order, and every NaN pattern lies above +inf, so a NaN operand trips the check as |x| > 65504 does (fmaxf drops NaNs: ADVICE r04)
__device__ __forceinline__ void f16_range_track(unsigned &amax, v4f y) {
  // (through float temporaries: clang 22 evaluates __builtin_bit_cast(unsigned, y.y) on an ext-vector ELEMENT as element 0 -- found when the range test stopped firing)
  const float fx = y.x, fy = y.y, fz = y.z, fw = y.w;
  const unsigned a = __builtin_bit_cast(unsigned, fx) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, fy) & 0x7fffffffu;
  const unsigned c = __builtin_bit_cast(unsigned, fz) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, fw) & 0x7fffffffu;
  amax = max(max(amax, a), max(b, max(c, d)));
}
constexpr unsigned F16_MAX_BITS = 0x477fe000u;   // 65504.0f

template <int N>
__device__ __forceinline__ void wait_lgkm4(v4f &a, v4f &b, v4f &c, v4f &d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm6(v4f &a, v4f &b, v4f &c, v4f &d, v4f &e, v4f &f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}

#ifndef MSI_X2_NSTG   // weight ring of the fp16 form (half the matrix work per tap: the DMA latency budget of a two-stage ring is one SHORT tap)
#define MSI_X2_NSTG 3
#endif
#ifndef MSI_X2_WAVES    // fp16 form at rate 1: four waves per SIMD = four workgroups per CU (40.7 KB of LDS each).  With the coordinate-bias registers requested
#define MSI_X2_WAVES 4  // AFTER the k-loop (MSI_X2_LATE_CB: 147 -> 131 VGPRs) the allocator reaches 128 without a spill: measured 57.5 -> 54.4 us per layer
#endif                  // (r04; forcing 128 with the bias registers held through the loop spilled 68 bytes and gained nothing)
#ifndef MSI_X2_LATE_CB
#define MSI_X2_LATE_CB 1
#endif
template <int NP>
__device__ __forceinline__ void split_mfma(f32x16 &acc, f32x16 &lo, const v4f &ah, const v4f &am, const v4f &al, const v4f &bh, const v4f &bm, const v4f &bl);
// TH = 4: the 4 x 16-pixel x 64-channel tile (one 32 x 32 accumulator per wave).  TH = 8 (r05, conv_halo8_x3_kernel, six-product form at rate 1): an 8 x 16-pixel
// tile -- a wave owns four tile rows = TWO 32 x 32 accumulators that share the weight fragments (18 instead of 24 fragment reads per 24 MFMAs), the 10 x 18 patch
// serves twice the outputs of the 6 x 18 one (halo 1.41 instead of 1.69), and per output pixel the workgroup moves HALF the weight bytes from L2 into LDS and runs
// half the prologues / patch swaps / barriers; 64.3 KB of LDS: two workgroups per CU.
template <int RATE, int APPLY, int NPL, int TH>
__device__ __forceinline__ void conv_halo_x3_body(const ConvParams &p, char *smem) {
  typedef HaloGeomX3<RATE, (NPL == 2 ? MSI_X2_NSTG : (MSI_X3_NSTG ? MSI_X3_NSTG : (RATE == 1 ? 2 : 3))), NPL, TH> G;
  constexpr int R = RATE, PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = TH / 4, NT = 1, BM = 16 * TH;
  static_assert(TH == 4 || (TH == 8 && NPL == 3 && RATE == 1), "the 8-row tile is built for the six-product form at rate 1");
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work decomposition: as conv_igemm_kernel (tail split), K-ranges in whole chunks ----
  const int CH = p.cpt0;                                  // 32-channel chunks of the input
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;                                               // (nclass = 1)
  }
  LnShard shard = {0, 0};   // (the lane's shard of the source's LayerNorm sums: requested here, reduced after the patch requests)
  if (APPLY) shard = ln_shard_load(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, tid);
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win, C = p.C0;
  v4f cbv[4] = {};
  if (MT == 1 && (!MSI_X2_LATE_CB || NPL != 2)) load_coord_bias(p, tile_m, tile_n, tid, cbv);   // in flight during the prologue and the k-loop (MT = 2: read by the epilogue)
  // the first two weight k-steps go out before the patch addresses are worked out (they depend on tile_n and the wave only)
  const int S = p.ksteps;                                 // 9 CH
  // x3 block of the packed blob: [tap][chunk][plane h | m | l][npad rows][64 B = 32 bf16 channels], 16-byte slots swizzled by
  // (row >> 2) & 3.  A wave's DMA instruction moves 16 rows x 64 B = 1 KB of one plane: three instructions per k-step
  const int plane_bytes = p.npad * G::B_ROW;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk_x3, 0, (int)((size_t)S * NPL * plane_bytes), 0x00020000);
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + (lane >> 2)) * G::B_ROW + (lane & 3) * 16);
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  if (!(MSI_X3_ABLATE & 1)) {                                                                                          \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * G::B_ROW;                                          \
    const int soff_ = ((tap) * CH + (c)) * NPL * plane_bytes;                                                          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0); \
    if (NPL == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0); \
  }
  MSI_B_ISSUE(c0, 0, 0)
  if (G::NSTG == 3) MSI_B_ISSUE(c0, 1, 1)

  // ---- per-lane patch elements: e = tid + 256 k -> patch pixel e / 8, 16-byte channel slot e % 8 (= tid % 8) ----
  unsigned voff[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - R + py;
    int iw = ow0 - R + px;
    if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);   // msi_train_net: wrap along W, zeros along H
    pok[k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
    voff[k] = pok[k] ? __umul24((unsigned)(ih * W + iw), (unsigned)(C * 4)) + (unsigned)(cslot * 16) : OOB;
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 8) : 0xffffffffu;
  }
  const size_t in_bytes = (size_t)H * W * C * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)in_bytes, 0x00020000);

  // producer's LayerNorm: mean / inv once per workgroup; the per-channel affine per chunk (the lane's four channels)
  float inv_f = 1.f, mu_hi = 0.f, mu_lo = 0.f;
  bool has_pad = false;                                   // wave-uniform: any of the wave's patch pixels is padding
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NLOAD], g4, be4;
  unsigned amax_ = 0u;   // (NPL == 2: the largest operand magnitude this lane stored -- the fp16 range check, f16_range_track)
  // patch of chunk c -> registers (+ gamma / beta of the lane's channels)
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[k_], (c) * ROW_BYTES, 0)); \
    if (APPLY) {                                                                                                       \
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + (c) * 32 + cslot * 4);                                          \
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + (c) * 32 + cslot * 4);                                          \
    }                                                                                                                  \
  }
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};                                                          \
    if (APPLY) {   /* scale = inv * gamma; shift = beta - mean * scale with the mean as hi + lo floats: fp32 ops only */ \
      s4 = inv_f * g4;                                                                                                 \
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (APPLY) {                                                                                                     \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */          \
      }                                                                                                                \
      if (NPL == 2) {   /* y = h + m' 2^-11, fp16 parts (round to nearest even; y - h is exact in fp32) */                 \
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));                                                     \
        typedef unsigned u2x_t __attribute__((ext_vector_type(2)));                                                    \
        f16_range_track(amax_, y);                                                                                       \
        const h2_t ha = {(_Float16)y.x, (_Float16)y.y}, hb = {(_Float16)y.z, (_Float16)y.w};                          \
        const h2_t ma = {(_Float16)((y.x - (float)ha.x) * 2048.f), (_Float16)((y.y - (float)ha.y) * 2048.f)};          \
        const h2_t mb = {(_Float16)((y.z - (float)hb.x) * 2048.f), (_Float16)((y.w - (float)hb.y) * 2048.f)};          \
        if (lds_a[k_] != 0xffffffffu) {                                                                                \
          *reinterpret_cast<u2x_t *>(smem + lds_a[k_]) = u2x_t{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)}; \
          *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 64) = u2x_t{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)}; \
        }                                                                                                              \
      } else {                                                                                                         \
      /* y = h + m + l, bf16 parts (round to nearest even; y - h and (y - h) - m are exact in fp32) */                  \
      unsigned h0, h1, m0, m1, l0, l1;                                                                                 \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h0) : "v"(y.x), "v"(y.y));                                             \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h1) : "v"(y.z), "v"(y.w));                                             \
      v4f r = y - v4f{__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h0 & 0xffff0000u),               \
                      __builtin_bit_cast(float, h1 << 16), __builtin_bit_cast(float, h1 & 0xffff0000u)};              \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m0) : "v"(r.x), "v"(r.y));                                             \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m1) : "v"(r.z), "v"(r.w));                                             \
      r = r - v4f{__builtin_bit_cast(float, m0 << 16), __builtin_bit_cast(float, m0 & 0xffff0000u),                    \
                  __builtin_bit_cast(float, m1 << 16), __builtin_bit_cast(float, m1 & 0xffff0000u)};                  \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l0) : "v"(r.x), "v"(r.y));                                             \
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l1) : "v"(r.z), "v"(r.w));                                             \
      if (lds_a[k_] != 0xffffffffu) {                                                                                  \
        typedef unsigned u2x_t __attribute__((ext_vector_type(2)));                                                    \
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_]) = u2x_t{h0, h1};                                                  \
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 64) = u2x_t{m0, m1};                                             \
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 128) = u2x_t{l0, l1};                                            \
      }                                                                                                                \
    }                                                                                                                  \
    }                                                                                                                  \
  }
  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  // A: plane P, K16-step s of the lane's pixel at + P * 64 + s * 32 (fh * 16 in the base); B: row wn * 32 + frow of plane P
  // at + P * B_PLANE, slot (2 s + fh) ^ ((row >> 2) & 3)
  // (MT = 2: the wave's second 32-pixel block is the two tile rows below: + 2 ROW_PITCH, an immediate)
  const unsigned a_base = lds_base + (unsigned)((2 * MT * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 16);
  unsigned b_s[2];
  (void)fswz;
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
    b_s[s_] = lds_base + G::A_BYTES + (wn * 32 + frow) * G::B_ROW + (((2 * s_ + fh) ^ ((frow >> 2) & 3)) << 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f32x16 acc[MT][1], acc_lo;   // (NPL == 2: acc = h.h, acc_lo = (h.m' + m'.h), folded as acc + acc_lo 2^-11 after the loop)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = acc_lo[r] = 0.f;

  // one k-step = tap TAP of the current chunk, weights in ring stage TAP % 3; the DMA of the k-step two ahead is issued
  // after the first MFMA quarter; before the closing barrier the NEXT k-step's weights must have landed: every VMEM
  // operation issued before them (the next chunk's patch loads, issued in tap 0) completes first (in-order return)
#define MSI_HTAP(TAP)                                                                                                  \
  {                                                                                                                    \
    constexpr int KH_ = (TAP) / 3, KW_ = (TAP) % 3;                                                                    \
    /* ring stage of this k-step: three stages -> TAP % 3 (a literal); two stages -> (TAP + chunk parity) & 1 (run-time scalar) */ \
    const unsigned bst_ = (unsigned)(G::NSTG == 3 ? (TAP) % 3 : (((TAP) ^ cpar) & 1)) * G::B_STAGE;                     \
    constexpr int AOFF_ = KH_ * R * G::ROW_PITCH + KW_ * R * G::PIX_BYTES;                                             \
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];                                                                \
    if (MSI_X3_EARLY_DMA || G::NSTG == 2) {   /* the k-step NSTG - 1 ahead: its ring stage was last read in the previous k-step (closing barrier passed) */ \
      constexpr int PD_ = G::NSTG - 1;                                                                                 \
      const int stn_ = G::NSTG == 3 ? ((TAP) + 2) % 3 : ((((TAP) ^ cpar) & 1) ^ 1);   /* (two stages: the other one) */  \
      if ((TAP) + PD_ < 9) { MSI_B_ISSUE(c, (TAP) + PD_, stn_) }                                                       \
      else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, (TAP) + PD_ - 9, stn_) }                                               \
    }                                                                                                                  \
    if (NPL == 2) {                                                                                                    \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                               \
        ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);                              \
        bh_[s_] = lds_read128<0>(b_s[s_] + bst_);                                                                      \
        am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);                         \
        bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);                                                             \
      }                                                                                                                \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                               \
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);                                                    \
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);                                                            \
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bm_[s_]), __builtin_bit_cast(f16x8, ah_[s_]), acc_lo, 0, 0, 0); \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh_[s_]), __builtin_bit_cast(f16x8, ah_[s_]), acc[0][0], 0, 0, 0); \
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh_[s_]), __builtin_bit_cast(f16x8, am_[s_]), acc_lo, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (s_ == 0) {                                                                                                 \
          if ((TAP) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                          \
          if (!MSI_X3_EARLY_DMA && G::NSTG == 3) {                                                                     \
            if ((TAP) + 2 < 9) { MSI_B_ISSUE(c, (TAP) + 2, ((TAP) + 2) % 3) }                                          \
            else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, (TAP) - 7, ((TAP) + 2) % 3) }                                    \
          }                                                                                                            \
        }                                                                                                              \
      }                                                                                                                \
    } else if (MT == 2) {                                                                                              \
      /* two pixel blocks i = 0, 1 against ONE set of weight fragments per K16 step s: 18 reads (at most 12 in flight: lgkmcnt is four bits), 24 MFMAs */ \
      constexpr int A1_ = AOFF_ + 2 * G::ROW_PITCH;                                                                    \
      v4f xh_[2][2], xm_[2][2], xl_[2][2];   /* [s][i] */                                                              \
      bh_[0] = lds_read128<0>(b_s[0] + bst_); bm_[0] = lds_read128<G::B_PLANE>(b_s[0] + bst_); bl_[0] = lds_read128<2 * G::B_PLANE>(b_s[0] + bst_); \
      xh_[0][0] = lds_read128<AOFF_>(a_base); xm_[0][0] = lds_read128<AOFF_ + 64>(a_base); xl_[0][0] = lds_read128<AOFF_ + 128>(a_base); \
      xh_[0][1] = lds_read128<A1_>(a_base); xm_[0][1] = lds_read128<A1_ + 64>(a_base); xl_[0][1] = lds_read128<A1_ + 128>(a_base); \
      bh_[1] = lds_read128<0>(b_s[1] + bst_); bm_[1] = lds_read128<G::B_PLANE>(b_s[1] + bst_); bl_[1] = lds_read128<2 * G::B_PLANE>(b_s[1] + bst_); \
      wait_lgkm6<6>(bh_[0], bm_[0], bl_[0], xh_[0][0], xm_[0][0], xl_[0][0]);                                          \
      split_mfma<3>(acc[0][0], acc_lo, xh_[0][0], xm_[0][0], xl_[0][0], bh_[0], bm_[0], bl_[0]);                       \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      xh_[1][0] = lds_read128<AOFF_ + 32>(a_base); xm_[1][0] = lds_read128<AOFF_ + 96>(a_base); xl_[1][0] = lds_read128<AOFF_ + 160>(a_base); \
      xh_[1][1] = lds_read128<A1_ + 32>(a_base); xm_[1][1] = lds_read128<A1_ + 96>(a_base); xl_[1][1] = lds_read128<A1_ + 160>(a_base); \
      if ((TAP) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                              \
      wait_lgkm6<9>(xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);                                          \
      split_mfma<3>(acc[1][0], acc_lo, xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);                       \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      wait_lgkm6<3>(bh_[1], bm_[1], bl_[1], xh_[1][0], xm_[1][0], xl_[1][0]);                                          \
      split_mfma<3>(acc[0][0], acc_lo, xh_[1][0], xm_[1][0], xl_[1][0], bh_[1], bm_[1], bl_[1]);                       \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      wait_lgkm6<0>(xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);                                          \
      split_mfma<3>(acc[1][0], acc_lo, xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);                       \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    } else {                                                                                                           \
    if (!(MSI_X3_ABLATE & 8))                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);                                \
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);                                                                \
      am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);                           \
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);                                                   \
      al_[s_] = s_ == 0 ? lds_read128<AOFF_ + 128>(a_base) : lds_read128<AOFF_ + 160>(a_base);                         \
      bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);                                               \
    }                                                                                                                  \
    /* six products per K16 step, small terms first: m.m, l.h, h.l, m.h, h.m, h.h (weights = the MFMA's row operand) */  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);                                      \
      else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);                                              \
      if (!(MSI_X3_ABLATE & 16)) {                                                                                     \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm_[s_]), __builtin_bit_cast(bf16x8, am_[s_]), acc[0][0], 0, 0, 0); \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, al_[s_]), acc[0][0], 0, 0, 0); \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bl_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0); \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, am_[s_]), acc[0][0], 0, 0, 0); \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0); \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0); \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (s_ == 0) {                                                                                                   \
        if ((TAP) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                            \
        /* k-step two ahead: (c, TAP + 2) or (c + 1, TAP - 7) */                                                       \
        if (!MSI_X3_EARLY_DMA && G::NSTG == 3) {                                                                       \
        if ((TAP) + 2 < 9) { MSI_B_ISSUE(c, (TAP) + 2, ((TAP) + 2) % 3) }                                              \
        else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, (TAP) - 7, ((TAP) + 2) % 3) }                                        \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    }                                                                                                                  \
    {                                                                                                                  \
      const bool issued_ = ((TAP) + 2 < 9) || (c + 1 < c1);                                                            \
      if (G::NSTG == 2) {                                                                                              \
        if ((TAP) == 0 && c + 1 < c1) wait_vmcnt<NLOAD + (APPLY ? 2 : 0)>();   /* (the patch loads were issued after the DMA) */ \
        else wait_vmcnt<0>();                                                                                          \
      } else if ((TAP) == 0 && c + 1 < c1) wait_vmcnt<NPL + NLOAD + (APPLY ? 2 : 0)>();   /* patch loads + this tap's DMA in flight (either order) */ \
      else if (issued_) wait_vmcnt<NPL>();                                                                             \
      else wait_vmcnt<0>();                                                                                            \
    }                                                                                                                  \
    if (!(MSI_X3_ABLATE & 4)) __builtin_amdgcn_s_barrier();                                                            \
  }

  // ---- prologue: first patch, first two weight k-steps ----
  int c = c0;
  MSI_PATCH_LOAD(c0)                                      // (the weights of k-steps 0 and 1 are on their way already)
  if (APPLY) {   // the sums' round trip rides on the patch's (s_stat sits in the A region: read back before the patch lands)
    double *s_stat = reinterpret_cast<double *>(smem);
    ln_mean_inv_pre(shard, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    const double mu = s_stat[0];
    inv_f = (float)s_stat[1];
    mu_hi = (float)mu;
    mu_lo = (float)(mu - (double)mu_hi);
    __syncthreads();
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    const int cpar = (c - c0) & 1;   // (two-stage ring: nine k-steps per chunk flip the stage parity)
    MSI_HTAP(0) MSI_HTAP(1) MSI_HTAP(2) MSI_HTAP(3) MSI_HTAP(4) MSI_HTAP(5) MSI_HTAP(6) MSI_HTAP(7) MSI_HTAP(8)
    if (c + 1 < c1 && !(MSI_X3_ABLATE & 32)) {   // every wave has read the last tap of this chunk (closing barrier of tap 8): swap the patch
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_HTAP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
  if (MSI_X2_LATE_CB && NPL == 2) load_coord_bias(p, tile_m, tile_n, tid, cbv);   // (fp16 form: 16 registers less through the loop -- a fourth workgroup per CU)
  if (NPL == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = __builtin_fmaf(acc_lo[r], 1.f / 2048.f, acc[0][0][r]);
    // an operand beyond the fp16 range became inf (h) and NaN (m'): the layer's output is garbage -- say so
    if (__builtin_amdgcn_ballot_w64(amax_ > F16_MAX_BITS) != 0 && lane == 0) atomicOr(p.status, STATUS_F16_SPLIT_RANGE);
  }

  // ---- epilogue: as conv_igemm_kernel ----
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
  auto stamp = [&]() __attribute__((always_inline)) {
    if (p.dbg && tid == 0) {
      unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
      o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
  };
#endif
  if (!full) {
    constexpr int SLAB = BM * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)slot * (BM * 64)), 0, SLAB, 0x00020000);
    if (p.tile_cnt == nullptr) {
      dump_acc<MT, NT, 0>(acc, rsrc_p, tid);
#ifdef MSI_CONV_TIMING
      stamp();
#endif
      return;
    }
    dump_acc<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_p, tid);
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)(slot - ks) * (BM * 64)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (every thread has read s_old before the epilogue's strips reuse LDS)
  }
  emit_tile<BM, 64, MODE_CONV>(p, acc, tile_m, tile_n, 0, b, tid, cbv, MT == 1 && p.coord_bias != nullptr, smem);
#ifdef MSI_CONV_TIMING
  stamp();
#endif
}

template <int RATE, int APPLY, int NPL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NPL == 2 && RATE == 1) ? MSI_X2_WAVES : 2)))
conv_halo_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_halo_x3_body<RATE, APPLY, NPL, 4>(p, smem);
#endif
}
// the 8 x 16-pixel tile of the six-product form at rate 1 (conv_halo_x3_body, TH = 8): two workgroups per CU
template <int APPLY, int NPL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
conv_halo8_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_halo_x3_body<1, APPLY, NPL, 8>(p, smem);
#endif
}

// ---- halo-patch convolution, stride 2 (fp32; r03) --------------------------------------------------------------------
// The three stride-2 3x3 layers (conv1_2, conv2_2, conv3_3) were the weakest fp32 layers on the tap kernel (68-77 % of the MFMA
// peak) and, reading their input through DMA, kept their producers' ln_apply launches alive (conv1_1's is the biggest of the
// network).  A stride-2 tap (kh, kw) of output pixel (oh, ow) reads input (2 oh + kh - pad, 2 ow + kw - pad): taps of equal
// (kh, kw) parity read ONE of the four parity planes of the input at unit stride, so per 32-channel group the kernel stages four
// small patches in turn -- UNIT u = 2 (kh_min) + (kw_min), (4 + 1) x (16 + 1) pixels of plane (kh_min - pad, kw_min - pad) mod 2 --
// and runs that unit's taps on it exactly like conv_halo_kernel runs its nine (immediate LDS offsets dy, dx in {0, 1}):
//   unit 0: taps (0,0) (0,2) (2,0) (2,2)   unit 1: (0,1) (2,1)   unit 2: (1,0) (1,2)   unit 3: (1,1)      -- 9 k-steps per group,
// so the weight ring's stage of a k-step is a literal as before (blob tap-major, row block tap * CH + group).  A unit's patch is
// requested during the previous unit's first k-step (16-byte slots through registers: the producer's LayerNorm + ReLU applied on
// the way when APPLY) and stored after its last; four patch swaps per group instead of one, each a fifth of the stride-1 patch.
// pad = 0 (TF SAME with an even input: CoordNet) or 1 (wrap_pad(1, 1) + VALID: msi_train_net; rows -1 / H are zeros, columns wrap).
struct HaloGeomS2 {
  static constexpr int PW = 17, PH = 5, NPX = PW * PH;
  static constexpr int PIX_BYTES = 144;
  static constexpr int ROW_PITCH = ((PW * PIX_BYTES + 127) / 256) * 256 + 128;   // (256 n + 128: see HaloGeom)
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_STAGE = 64 * ROW_BYTES;
  static constexpr int NSTG = 3;
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;
  static_assert(LDS_BYTES >= EPI_STAGE_BYTES / 2, "the epilogue's staging strips of a 64 x 64 fp32 tile (18 KB)");
};

template <int APPLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
conv_halo_s2_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeomS2 G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = 1, NT = 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work decomposition: as conv_halo_kernel (tail split; K-ranges in whole 32-channel groups) ----
  const int CH = p.cpt0;
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * 4, ow0 = (tile_m - tyi * p.halo_tx) * 16;   // the tile of the OUTPUT grid
  const int H = p.Hin, W = p.Win, C = p.C0;
  // the first two weight k-steps (taps (0,0) and (0,2) of group c0) before anything else
  const int S = p.ksteps;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)S * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + drow) * ROW_BYTES + dslot * 16);
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * ROW_BYTES;                                         \
    const int soff_ = ((tap) * CH + (c)) * p.npad * ROW_BYTES;                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);            \
  }
  MSI_B_ISSUE(c0, 0, 0)
  if (G::NSTG == 3) MSI_B_ISSUE(c0, 2, 1)

  // ---- per-lane patch slots of the four units: e = tid + 256 k -> patch pixel e / 8, 16-byte channel slot e % 8 ----
  unsigned voff[4][NLOAD], lds_a[NLOAD];
  bool pok[4][NLOAD];
  const int cslot = tid & 7;
  const size_t in_bytes = (size_t)H * W * C * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)in_bytes, 0x00020000);
  v4f araw[NLOAD], g4, be4;
  // (unit 0 first and its patch of group c0 requested at once: the other units' offsets are worked out under that round trip)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int pp = (tid + 256 * k) >> 3;
      const int py = pp / PW, px = pp - py * PW;
      if (u == 0) lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
      const int ih = 2 * (oh0 + py) + (u >> 1) - p.pad_t;
      int iw = 2 * (ow0 + px) + (u & 1) - p.pad_l;
      if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);
      pok[u][k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
      voff[u][k] = pok[u][k] ? __umul24((unsigned)(ih * W + iw), (unsigned)(C * 4)) + (unsigned)(cslot * 16) : OOB;
    }
    if (u == 0) {
#pragma unroll
      for (int k = 0; k < NLOAD; ++k)
        araw[k] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[0][k], c0 * ROW_BYTES, 0));
      if (APPLY) {
        g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + c0 * 32 + cslot * 4);
        be4 = *reinterpret_cast<const v4f *>(p.ln_beta + c0 * 32 + cslot * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float inv_f = 1.f, mu_hi = 0.f, mu_lo = 0.f;
  bool has_pad = false;                                   // wave-uniform: any of the wave's patch pixels (any unit) is padding
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) bad |= (lds_a[k] != 0xffffffffu) && !pok[u][k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};   // the group's affine (the lane's four channels): set with unit 0
  // patch of (group c, unit U) -> registers (+ gamma / beta of the lane's channels with unit 0)
#define MSI_PATCH_LOAD(c, U)                                                                                           \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[U][k_], (c) * ROW_BYTES, 0)); \
    if (APPLY && (U) == 0) {                                                                                           \
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + (c) * 32 + cslot * 4);                                          \
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + (c) * 32 + cslot * 4);                                          \
    }                                                                                                                  \
  }
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
#define MSI_PATCH_STORE(U)                                                                                             \
  {                                                                                                                    \
    if (APPLY && (U) == 0) {                                                                                           \
      s4 = inv_f * g4;                                                                                                 \
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (APPLY) {                                                                                                     \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[U][k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */       \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;                                    \
    }                                                                                                                  \
  }

  // ---- MFMA side (as conv_halo_kernel: a wave owns two tile rows x 16 columns x 32 channels) ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((2 * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 64);
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * 32 + frow) * ROW_BYTES + (((fh * 4 + q) ^ fswz) << 4);
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  // k-step J = 0..8 of the current group: unit, tap and patch offsets are literals
#define MSI_S2_TAP(J) ((J) == 0 ? 0 : (J) == 1 ? 2 : (J) == 2 ? 6 : (J) == 3 ? 8 : (J) == 4 ? 1 : (J) == 5 ? 7 : (J) == 6 ? 3 : (J) == 7 ? 5 : 4)
#define MSI_S2_UNIT(J) ((J) < 4 ? 0 : (J) < 6 ? 1 : (J) < 8 ? 2 : 3)
#define MSI_S2STEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int TAP_ = MSI_S2_TAP(J), U_ = MSI_S2_UNIT(J), ST_ = (J) % 3;                                            \
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;                                                        \
    constexpr bool FIRST_ = (J) == 0 || (J) == 4 || (J) == 6 || (J) == 8, LAST_ = (J) == 3 || (J) == 5 || (J) == 7 || (J) == 8; \
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;                                                     \
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */                               \
    v4f a_[4], b_[4];                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      a_[q_] = q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 16>(a_base)                        \
             : q_ == 2 ? lds_read128<AOFF_ + 32>(a_base) : lds_read128<AOFF_ + 48>(a_base);                            \
      b_[q_] = lds_read128<ST_ * G::B_STAGE>(b_q[q_]);                                                                 \
    }                                                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);                                                                         \
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);                                                                         \
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);                                                                         \
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);                                                                         \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[0][0], 0, 0, 0);                        \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[0][0], 0, 0, 0);                        \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (q_ == 0) {                                                                                                   \
        if (FIRST_ && more_) {                                                                                         \
          if (U_ < 3) MSI_PATCH_LOAD(c, (U_ + 1) & 3) else MSI_PATCH_LOAD(c + 1, 0)                                    \
        }                                                                                                              \
        /* k-step two ahead: (c, J + 2) or (c + 1, J - 7) */                                                           \
        if ((J) + 2 < 9) { MSI_B_ISSUE(c, MSI_S2_TAP(((J) + 2) % 9), ((J) + 2) % 3) }                                  \
        else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, MSI_S2_TAP(((J) + 2) % 9), ((J) + 2) % 3) }                          \
      }                                                                                                                \
    }                                                                                                                  \
    {                                                                                                                  \
      const bool issued_ = ((J) + 2 < 9) || (c + 1 < c1);                                                              \
      /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */ \
      if (FIRST_ && !LAST_ && more_) wait_vmcnt<2 + NLOAD>();                                                          \
      else if (issued_) wait_vmcnt<2>();                                                                               \
      else wait_vmcnt<0>();                                                                                            \
    }                                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                                      \
    if (LAST_ && more_) {   /* every wave has read this unit's last tap: swap the patch */                             \
      MSI_PATCH_STORE((U_ + 1) & 3)                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
      __builtin_amdgcn_s_barrier();                                                                                    \
    }                                                                                                                  \
  }

  // ---- prologue: unit 0 of the first group ----
  int c = c0;   // (unit 0's patch of group c0 is on its way)
  if (APPLY) {
    double *s_stat = reinterpret_cast<double *>(smem);
    ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    const double mu = s_stat[0];
    inv_f = (float)s_stat[1];
    mu_hi = (float)mu;
    mu_lo = (float)(mu - (double)mu_hi);
    __syncthreads();
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    MSI_S2STEP(0) MSI_S2STEP(1) MSI_S2STEP(2) MSI_S2STEP(3) MSI_S2STEP(4) MSI_S2STEP(5) MSI_S2STEP(6) MSI_S2STEP(7) MSI_S2STEP(8)
  }
#undef MSI_S2STEP
#undef MSI_S2_UNIT
#undef MSI_S2_TAP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD

  // ---- epilogue: as conv_halo_kernel ----
  if (!full) {
    constexpr int SLAB = 64 * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)slot * (64 * 64)), 0, SLAB, 0x00020000);
    if (p.tile_cnt == nullptr) {
      dump_acc<MT, NT, 0>(acc, rsrc_p, tid);
      return;
    }
    dump_acc<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_p, tid);
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)(slot - ks) * (64 * 64)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (s_old has been read by every thread before the strip below reuses LDS)
  }
  emit_tile<64, 64, MODE_CONV>(p, acc, tile_m, tile_n, 0, b, tid, smem);
#endif
}


// ---- shared pieces of the split kernels' stride-2 / conv-transpose forms (NP = 3: bf16 h | m | l, six products; NP = 2: fp16 h | m', three) ----
template <int NP>
__device__ __forceinline__ void split_store(char *smem, unsigned off, v4f y, unsigned &amax) {
  typedef unsigned u2x_t __attribute__((ext_vector_type(2)));
  if (NP == 2) {   // y = h + m' 2^-11, fp16 parts (round to nearest even; y - h is exact in fp32)
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    f16_range_track(amax, y);
    const h2_t ha = {(_Float16)y.x, (_Float16)y.y}, hb = {(_Float16)y.z, (_Float16)y.w};
    const h2_t ma = {(_Float16)((y.x - (float)ha.x) * 2048.f), (_Float16)((y.y - (float)ha.y) * 2048.f)};
    const h2_t mb = {(_Float16)((y.z - (float)hb.x) * 2048.f), (_Float16)((y.w - (float)hb.y) * 2048.f)};
    if (off != 0xffffffffu) {
      *reinterpret_cast<u2x_t *>(smem + off) = u2x_t{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
      *reinterpret_cast<u2x_t *>(smem + off + 64) = u2x_t{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
    }
  } else {         // y = h + m + l, bf16 parts (see conv_halo_x3_kernel)
    unsigned h0, h1, m0, m1, l0, l1;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h0) : "v"(y.x), "v"(y.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h1) : "v"(y.z), "v"(y.w));
    v4f r = y - v4f{__builtin_bit_cast(float, h0 << 16), __builtin_bit_cast(float, h0 & 0xffff0000u),
                    __builtin_bit_cast(float, h1 << 16), __builtin_bit_cast(float, h1 & 0xffff0000u)};
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m0) : "v"(r.x), "v"(r.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m1) : "v"(r.z), "v"(r.w));
    r = r - v4f{__builtin_bit_cast(float, m0 << 16), __builtin_bit_cast(float, m0 & 0xffff0000u),
                __builtin_bit_cast(float, m1 << 16), __builtin_bit_cast(float, m1 & 0xffff0000u)};
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l0) : "v"(r.x), "v"(r.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l1) : "v"(r.z), "v"(r.w));
    if (off != 0xffffffffu) {
      *reinterpret_cast<u2x_t *>(smem + off) = u2x_t{h0, h1};
      *reinterpret_cast<u2x_t *>(smem + off + 64) = u2x_t{m0, m1};
      *reinterpret_cast<u2x_t *>(smem + off + 128) = u2x_t{l0, l1};
    }
  }
}
// the products of one K16 step (weights = the MFMA's row operand), small terms first; NP = 2: lo collects h.m' + m'.h
template <int NP>
__device__ __forceinline__ void split_mfma(f32x16 &acc, f32x16 &lo, const v4f &ah, const v4f &am, const v4f &al, const v4f &bh, const v4f &bm, const v4f &bl) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  if (NP == 2) {
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bm), __builtin_bit_cast(f16x8, ah), lo, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh), __builtin_bit_cast(f16x8, ah), acc, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh), __builtin_bit_cast(f16x8, am), lo, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm), __builtin_bit_cast(bf16x8, am), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh), __builtin_bit_cast(bf16x8, al), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bl), __builtin_bit_cast(bf16x8, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh), __builtin_bit_cast(bf16x8, am), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm), __builtin_bit_cast(bf16x8, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh), __builtin_bit_cast(bf16x8, ah), acc, 0, 0, 0);
  }
}
// NP = 2, after the k-loop: acc += lo 2^-11; an operand beyond the fp16 range (h = inf, m' = NaN) is reported
template <int NP>
__device__ __forceinline__ void split_finish(f32x16 &acc, const f32x16 &lo, unsigned amax, int lane, int *status) {
  if (NP == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(lo[r], 1.f / 2048.f, acc[r]);
    if (__builtin_amdgcn_ballot_w64(amax > F16_MAX_BITS) != 0 && lane == 0) atomicOr(status, STATUS_F16_SPLIT_RANGE);
  }
}

// ---- the stride-2 halo-patch kernel through the six-product bf16 split (conv_halo_s2_kernel x conv_halo_x3_kernel; r04) ----------
#ifndef MSI_S2X3_NSTG   // weight ring of the six-product stride-2 kernel: 2 (r05: 43.1 KB of LDS, three workgroups per CU; the DMA of a k-step is issued at the head of
#define MSI_S2X3_NSTG 2 // the one before it, as in conv_halo_x3_kernel at rate 1) or 3 (r04: 55.4 KB, two workgroups per CU)
#endif
template <int NP>
struct HaloGeomS2X3 {
  static constexpr int PW = 17, PH = 5, NPX = PW * PH;
  static constexpr int PIX_BYTES = NP * 64 + 16;
  static constexpr int ROW_PITCH = ((PW * PIX_BYTES + 127) / 256) * 256 + 128;
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_ROW = 64, B_PLANE = 64 * B_ROW, B_STAGE = NP * B_PLANE;
  static constexpr int NSTG = NP == 3 ? MSI_S2X3_NSTG : 3;
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;
};

#ifndef MSI_S2X_WAVES
#define MSI_S2X_WAVES 3
#endif
template <int APPLY, int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 2 ? MSI_S2X_WAVES : (MSI_S2X3_NSTG == 2 ? 3 : 2))))
conv_halo_s2_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeomS2X3<NP> G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = 1, NT = 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // ---- work decomposition: as conv_halo_kernel (tail split; K-ranges in whole 32-channel groups) ----
  const int CH = p.cpt0;
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * 4, ow0 = (tile_m - tyi * p.halo_tx) * 16;   // the tile of the OUTPUT grid
  const int H = p.Hin, W = p.Win, C = p.C0;
  // the first two weight k-steps (taps (0,0) and (0,2) of group c0) before anything else
  const int S = p.ksteps;
  // (weights: the x3 block of the packed blob, three 64-byte-row planes per k-step -- see conv_halo_x3_kernel)
  const int plane_bytes = p.npad * G::B_ROW;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk_x3, 0, (int)((size_t)S * NP * plane_bytes), 0x00020000);
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + (lane >> 2)) * G::B_ROW + (lane & 3) * 16);
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * G::B_ROW;                                          \
    const int soff_ = ((tap) * CH + (c)) * NP * plane_bytes;                                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0); \
    if (NP == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0); \
  }
  MSI_B_ISSUE(c0, 0, 0)
  if (G::NSTG == 3) MSI_B_ISSUE(c0, 2, 1)

  // ---- per-lane patch slots of the four units: e = tid + 256 k -> patch pixel e / 8, 16-byte channel slot e % 8 ----
  unsigned voff[4][NLOAD], lds_a[NLOAD];
  bool pok[4][NLOAD];
  const int cslot = tid & 7;
  const size_t in_bytes = (size_t)H * W * C * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)in_bytes, 0x00020000);
  v4f araw[NLOAD], g4, be4;
  unsigned amax_ = 0u;
  // (unit 0 first and its patch of group c0 requested at once: the other units' offsets are worked out under that round trip)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int pp = (tid + 256 * k) >> 3;
      const int py = pp / PW, px = pp - py * PW;
      if (u == 0) lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 8) : 0xffffffffu;
      const int ih = 2 * (oh0 + py) + (u >> 1) - p.pad_t;
      int iw = 2 * (ow0 + px) + (u & 1) - p.pad_l;
      if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);
      pok[u][k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
      voff[u][k] = pok[u][k] ? __umul24((unsigned)(ih * W + iw), (unsigned)(C * 4)) + (unsigned)(cslot * 16) : OOB;
    }
    if (u == 0) {
#pragma unroll
      for (int k = 0; k < NLOAD; ++k)
        araw[k] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[0][k], c0 * ROW_BYTES, 0));
      if (APPLY) {
        g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + c0 * 32 + cslot * 4);
        be4 = *reinterpret_cast<const v4f *>(p.ln_beta + c0 * 32 + cslot * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float inv_f = 1.f, mu_hi = 0.f, mu_lo = 0.f;
  bool has_pad = false;                                   // wave-uniform: any of the wave's patch pixels (any unit) is padding
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) bad |= (lds_a[k] != 0xffffffffu) && !pok[u][k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};   // the group's affine (the lane's four channels): set with unit 0
  // patch of (group c, unit U) -> registers (+ gamma / beta of the lane's channels with unit 0)
#define MSI_PATCH_LOAD(c, U)                                                                                           \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[U][k_], (c) * ROW_BYTES, 0)); \
    if (APPLY && (U) == 0) {                                                                                           \
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + (c) * 32 + cslot * 4);                                          \
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + (c) * 32 + cslot * 4);                                          \
    }                                                                                                                  \
  }
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
#define MSI_PATCH_STORE(U)                                                                                             \
  {                                                                                                                    \
    if (APPLY && (U) == 0) {                                                                                           \
      s4 = inv_f * g4;                                                                                                 \
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (APPLY) {                                                                                                     \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[U][k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */       \
      }                                                                                                                \
      split_store<NP>(smem, lds_a[k_], y, amax_);                                                                      \
    }                                                                                                                  \
  }

  // ---- MFMA side (as conv_halo_kernel: a wave owns two tile rows x 16 columns x 32 channels) ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((2 * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 16);
  unsigned b_s[2];
  (void)fswz;
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
    b_s[s_] = lds_base + G::A_BYTES + (wn * 32 + frow) * G::B_ROW + (((2 * s_ + fh) ^ ((frow >> 2) & 3)) << 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  f32x16 acc[1][1], acc_lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = acc_lo[r] = 0.f;

  // k-step J = 0..8 of the current group: unit, tap and patch offsets are literals
#define MSI_S2_TAP(J) ((J) == 0 ? 0 : (J) == 1 ? 2 : (J) == 2 ? 6 : (J) == 3 ? 8 : (J) == 4 ? 1 : (J) == 5 ? 7 : (J) == 6 ? 3 : (J) == 7 ? 5 : 4)
#define MSI_S2_UNIT(J) ((J) < 4 ? 0 : (J) < 6 ? 1 : (J) < 8 ? 2 : 3)
#define MSI_S2STEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int TAP_ = MSI_S2_TAP(J), U_ = MSI_S2_UNIT(J);                                                           \
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;                                                        \
    constexpr bool FIRST_ = (J) == 0 || (J) == 4 || (J) == 6 || (J) == 8, LAST_ = (J) == 3 || (J) == 5 || (J) == 7 || (J) == 8; \
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;                                                     \
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */                               \
    /* ring stage of this k-step: three stages -> J % 3 (a literal); two -> (J + group parity) & 1 (nine k-steps per group flip it) */ \
    const unsigned bst_ = (unsigned)(G::NSTG == 3 ? (J) % 3 : (((J) ^ cpar) & 1)) * G::B_STAGE;                         \
    if (G::NSTG == 2) {   /* the NEXT k-step's weights into the other stage: it was last read in the previous k-step (closing barrier passed) */ \
      const int stn_ = (((J) ^ cpar) & 1) ^ 1;                                                                         \
      if ((J) + 1 < 9) { MSI_B_ISSUE(c, MSI_S2_TAP(((J) + 1) % 9), stn_) }                                             \
      else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, MSI_S2_TAP(0), stn_) }                                                 \
    }                                                                                                                  \
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);                                \
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);                                                                        \
      am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);                           \
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);                                                               \
      if (NP == 3) {                                                                                                   \
        al_[s_] = s_ == 0 ? lds_read128<AOFF_ + 128>(a_base) : lds_read128<AOFF_ + 160>(a_base);                       \
        bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);                                                         \
      } else { al_[s_] = ah_[s_]; bl_[s_] = bh_[s_]; }                                                                 \
    }                                                                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      if (NP == 3) {                                                                                                   \
        if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);                                    \
        else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);                                            \
      } else {                                                                                                         \
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);                                                    \
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);                                                            \
      }                                                                                                                \
      split_mfma<NP>(acc[0][0], acc_lo, ah_[s_], am_[s_], al_[s_], bh_[s_], bm_[s_], bl_[s_]);                         \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (s_ == 0) {                                                                                                   \
        if (FIRST_ && more_) {                                                                                         \
          if (U_ < 3) MSI_PATCH_LOAD(c, (U_ + 1) & 3) else MSI_PATCH_LOAD(c + 1, 0)                                    \
        }                                                                                                              \
        /* (three stages) k-step two ahead: (c, J + 2) or (c + 1, J - 7) */                                            \
        if (G::NSTG == 3) {                                                                                            \
        if ((J) + 2 < 9) { MSI_B_ISSUE(c, MSI_S2_TAP(((J) + 2) % 9), ((J) + 2) % 3) }                                  \
        else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, MSI_S2_TAP(((J) + 2) % 9), ((J) + 2) % 3) }                          \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    {                                                                                                                  \
      const bool issued_ = ((J) + 2 < 9) || (c + 1 < c1);                                                              \
      /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */ \
      if (G::NSTG == 2) {   /* (the DMA went out BEFORE the patch loads of this k-step: in-order return) */                \
        if (FIRST_ && !LAST_ && more_) wait_vmcnt<NLOAD>();                                                            \
        else wait_vmcnt<0>();                                                                                          \
      } else if (FIRST_ && !LAST_ && more_) wait_vmcnt<NP + NLOAD>();                                                   \
      else if (issued_) wait_vmcnt<NP>();                                                                              \
      else wait_vmcnt<0>();                                                                                            \
    }                                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                                      \
    if (LAST_ && more_) {   /* every wave has read this unit's last tap: swap the patch */                             \
      MSI_PATCH_STORE((U_ + 1) & 3)                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
      __builtin_amdgcn_s_barrier();                                                                                    \
    }                                                                                                                  \
  }

  // ---- prologue: unit 0 of the first group ----
  int c = c0;   // (unit 0's patch of group c0 is on its way)
  if (APPLY) {
    double *s_stat = reinterpret_cast<double *>(smem);
    ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    const double mu = s_stat[0];
    inv_f = (float)s_stat[1];
    mu_hi = (float)mu;
    mu_lo = (float)(mu - (double)mu_hi);
    __syncthreads();
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    const int cpar = (c - c0) & 1;   // (two-stage ring: nine k-steps per group flip the stage parity)
    (void)cpar;
    MSI_S2STEP(0) MSI_S2STEP(1) MSI_S2STEP(2) MSI_S2STEP(3) MSI_S2STEP(4) MSI_S2STEP(5) MSI_S2STEP(6) MSI_S2STEP(7) MSI_S2STEP(8)
  }
#undef MSI_S2STEP
#undef MSI_S2_UNIT
#undef MSI_S2_TAP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
  split_finish<NP>(acc[0][0], acc_lo, amax_, lane, p.status);

  // ---- epilogue: as conv_halo_kernel ----
  if (!full) {
    constexpr int SLAB = 64 * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)slot * (64 * 64)), 0, SLAB, 0x00020000);
    if (p.tile_cnt == nullptr) {
      dump_acc<MT, NT, 0>(acc, rsrc_p, tid);
      return;
    }
    dump_acc<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_p, tid);
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)(slot - ks) * (64 * 64)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (s_old has been read by every thread before the strip below reuses LDS)
  }
  emit_tile<64, 64, MODE_CONV>(p, acc, tile_m, tile_n, 0, b, tid, smem);
#endif
}

// ---- halo-patch kernel for the conv-transpose layers (4x4, stride 2, SAME; fp32) ----------------------------------
// Output (2 mh + ph, 2 mw + pw) of parity class (ph, pw) reads input rows mh + {0, ph ? +1 : -1} and columns
// mw + {0, pw ? +1 : -1} (tap_delta).  A workgroup owns a 4 x 16 tile of the INPUT grid x 64 channels for the TWO classes
// of one output-row parity ph (pw = 0, 1): per 32-channel chunk of either source of the skip concat it stages the 6 x 18
// halo patch ONCE -- through registers, so that a RAW source gets its producer's LayerNorm + ReLU on the way
// (p.halo_apply bit per source: neither the decoder input nor the skip tensor needs an ln_apply launch for this consumer)
// -- and runs 2 classes x 4 taps = 8 k-steps on it, each class into its own accumulator tile (2 x 16 registers).
// Class pw, tap (th, tw) reads patch row 1 + (ph ? th : -th) (the only run-time part of a fragment address: two base
// registers) and column 1 + (pw ? tw : -tw) (immediate).  The two workgroups of a tile (ph = 0, 1) are grid neighbours
// (same XCD: the patch comes from HBM once).  Weights: 3-stage DMA ring with a run-time stage index (8 k-steps per chunk
// do not divide by 3; four stages would leave three workgroups per CU instead of four), k order per class: chunk-major,
// tap-minor over the tap-major packed blob.  K-ranges of split tiles: whole chunks; a partial tile dumps two slabs
// (class-minor) and the last arriver sums each class in ascending k.
// MEASURED, twice.  r02: all FOUR classes per workgroup (64 accumulator registers -> three workgroups per CU, four slabs per
// K-range) lost to the tap kernel (conv8_1 218 vs 200 us, profiles/r02_E_convt_halo_kernel_stats.txt).  r03: this two-class
// form (four workgroups per CU, 125 VGPRs, no scratch) is correct -- every parity / determinism / fix-up-equality test passes
// with it on -- and still loses: conv6_1 223 vs 197 us, conv7_1 210 vs 195, conv8_1 214 vs 202; it drops six ln_apply launches
// (77 -> 35 us per frame) but makes conv2_1 / conv3_1 / conv4_1 APPLY layers (+10 us): network 2.525-2.532 ms against
// 2.476-2.485 ms with the tap kernel (two interleaved repeats, profiles/r03_b_convt_halo2_kernel_stats.txt).  The tap kernel's
// k-loop has no VALU at all and five workgroups per CU; here every chunk costs ~120 VALU (patch affine + ReLU through
// registers, run-time ring stage, per-source address selects) per 8 192 matrix cycles.  Plan option HALO bit 1, default off.
struct ConvtHaloGeom : HaloGeom<1> {};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
convt_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef ConvtHaloGeom G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD, NSTG = G::NSTG, PD = NSTG - 1;
  static_assert(NSTG == 3 && PD == 2, "prefetch distance two");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int CH = p.cpt0 + p.cpt1;                         // 32-channel chunks of both sources
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int ph, tile_m, tile_n, b;
  {   // row parity fastest (p.nclass = 2 here), then M tiles, N tiles, samples
    int r = t;
    ph = r & 1; r >>= 1;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * 4, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win;

  // the first two weight k-steps (class pw = 0, taps 0 and 1 of chunk c0) go out before the patch addresses are worked out
  const int S = p.ksteps;                                 // k-steps per class: 4 CH
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)4 * S * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + drow) * ROW_BYTES + dslot * 16);
  // weights of k-step (class, tap, chunk c) -> ring stage st; packed blob: [class][tap * CH + c][npad][128 B]
#define MSI_B_ISSUE(cls, tap, c, st)                                                                                   \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * ROW_BYTES;                                         \
    const int soff_ = (((cls) * S + (tap) * CH + (c)) * p.npad) * ROW_BYTES;                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);            \
  }
  MSI_B_ISSUE(2 * ph, 0, c0, 0)
  MSI_B_ISSUE(2 * ph, 1, c0, 1)

  // ---- per-lane patch elements (as conv_halo_kernel; the byte offset depends on the source's channel count) ----
  unsigned pixi[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    pok[k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;   // SAME: zeros outside
    pixi[k] = (unsigned)(ih * W + iw);
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
  }
  const size_t in0 = (size_t)H * W * p.C0 * 4, in1 = (size_t)H * W * p.C1 * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in0), 0, (int)in0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x1 + (size_t)b * in1), 0, (int)(in1 ? in1 : 16), 0x00020000);

  // LayerNorm of the raw sources: mean as hi + lo floats and 1 / sigma per source
  float inv_f[2] = {1.f, 1.f}, mu_hi[2] = {0.f, 0.f}, mu_lo[2] = {0.f, 0.f};
  bool has_pad = false;
  {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NLOAD], g4, be4;
  int src_ld = 0;                                         // source of the patch held in araw
  // patch of chunk c -> registers (+ gamma / beta of the lane's channels when that source is raw)
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    const int s_ = (c) >= p.cpt0 ? 1 : 0, cc_ = s_ ? (c) - p.cpt0 : (c);                                               \
    const unsigned cb_ = (unsigned)((s_ ? p.C1 : p.C0) * 4);                                                           \
    src_ld = s_;                                                                                                       \
    if (s_ == 0) {                                                                                                     \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a0, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));           \
    } else {                                                                                                           \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a1, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));           \
    }                                                                                                                  \
    const float *gp_ = (s_ ? p.ln_gamma1 : p.ln_gamma), *bp_ = (s_ ? p.ln_beta1 : p.ln_beta);                          \
    if ((p.halo_apply >> s_) & 1) {                                                                                    \
      g4 = *reinterpret_cast<const v4f *>(gp_ + cc_ * 32 + cslot * 4);                                                 \
      be4 = *reinterpret_cast<const v4f *>(bp_ + cc_ * 32 + cslot * 4);                                                \
    } else {   /* (same number of VMEM operations on both paths: the vmcnt arithmetic of the k-steps counts them) */   \
      g4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16);                                                         \
      be4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16 + 128);                                                  \
    }                                                                                                                  \
  }
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    const bool ap_ = (p.halo_apply >> src_ld) & 1;                                                                     \
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};                                                          \
    if (ap_) {                                                                                                         \
      const float ih_ = src_ld ? inv_f[1] : inv_f[0], mh_ = src_ld ? mu_hi[1] : mu_hi[0], ml_ = src_ld ? mu_lo[1] : mu_lo[0]; \
      s4 = ih_ * g4;                                                                                                   \
      const v4f nh = {-mh_, -mh_, -mh_, -mh_}, nl = {-ml_, -ml_, -ml_, -ml_};                                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (ap_) {                                                                                                       \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};                                                          \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;                                    \
    }                                                                                                                  \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  // fragment base of tap row th = 0 (patch row 1 + local row) and of th = 1 (one row up for ph = 0, one down for ph = 1)
  const unsigned a_base0 = lds_base + (unsigned)((1 + 2 * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 64);
  const unsigned a_base1 = ph ? a_base0 + G::ROW_PITCH : a_base0 - G::ROW_PITCH;
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * 32 + frow) * ROW_BYTES + (((fh * 4 + q) ^ fswz) << 4);
  f32x16 acc[2][1][1];
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cl][0][0][r] = 0.f;

  // k-step J of the chunk = class pw = J / 4, tap (th, tw) = ((J / 2) & 1, J & 1), weights in ring stage st (run-time).
  // The DMA of the k-step PD = 2 ahead and (J == 0) the next chunk's patch loads are issued after the first MFMA quarter;
  // before the closing barrier the NEXT k-step's weights must have landed: they were issued one k-step ago, so only what
  // THIS k-step issued (2 DMA, + the patch loads of J == 0) may still be in flight (in-order return).
  constexpr int NPL = NLOAD + 2;                          // VMEM operations of a patch load
#define MSI_CTSTEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int PWC_ = (J) >> 2, TH_ = ((J) >> 1) & 1, TW_ = (J) & 1;                                                \
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */                \
    const unsigned ab_ = TH_ ? a_base1 : a_base0;                                                                      \
    v4f a_[4], b_[4];                                                                                                  \
    const unsigned bst_ = (unsigned)st * G::B_STAGE;                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      a_[q_] = q_ == 0 ? lds_read128<COFF_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 16>(ab_)                              \
             : q_ == 2 ? lds_read128<COFF_ + 32>(ab_) : lds_read128<COFF_ + 48>(ab_);                                  \
      b_[q_] = lds_read128<0>(b_q[q_] + bst_);                                                                         \
    }                                                                                                                  \
    bool issued_ = false;                                                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);                                                                         \
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);                                                                         \
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);                                                                         \
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);                                                                         \
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[PWC_][0][0], 0, 0, 0);            \
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[PWC_][0][0], 0, 0, 0);            \
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[PWC_][0][0], 0, 0, 0);            \
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[PWC_][0][0], 0, 0, 0);            \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (q_ == 0) {                                                                                                   \
        if ((J) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                              \
        int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;                                                       \
        constexpr int JN_ = ((J) + PD) & 7;                                                                            \
        if ((J) + PD < 8) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_) }                        \
        else if (c + 1 < c1) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_) }                 \
      }                                                                                                                \
    }                                                                                                                  \
    if ((J) == 0 && c + 1 < c1) wait_vmcnt<2 + NPL>();                                                                 \
    else if (issued_) wait_vmcnt<2>();                                                                                 \
    else wait_vmcnt<0>();                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                      \
    st = st + 1 == NSTG ? 0 : st + 1;                                                                                  \
  }

  // ---- prologue: first patch (the first two weight k-steps are on their way), the sources' LayerNorm statistics ----
  int c = c0, st = 0;
  MSI_PATCH_LOAD(c0)
  if (p.halo_apply) {
    double *s_stat = reinterpret_cast<double *>(smem);
    if (p.halo_apply & 1) {
      ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv_f[0] = (float)s_stat[1]; mu_hi[0] = (float)mu; mu_lo[0] = (float)(mu - (double)mu_hi[0]);
      __syncthreads();
    }
    if (p.halo_apply & 2) {
      ln_mean_inv(p.ln_sums1 + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n1, p.ln_scl_src1, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv_f[1] = (float)s_stat[1]; mu_hi[1] = (float)mu; mu_lo[1] = (float)(mu - (double)mu_hi[1]);
      __syncthreads();
    }
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    MSI_CTSTEP(0) MSI_CTSTEP(1) MSI_CTSTEP(2) MSI_CTSTEP(3) MSI_CTSTEP(4) MSI_CTSTEP(5) MSI_CTSTEP(6) MSI_CTSTEP(7)
    if (c + 1 < c1) {
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_CTSTEP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD

  // ---- epilogue: two class tiles ----
  if (!full) {
    constexpr int SLAB = 64 * 64 * 4;
    if (p.tile_cnt == nullptr) {                          // separate fix-up launch (conv_fixup_kernel, class = blockIdx.y)
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (64 * 64)), 0, SLAB, 0x00020000);
        dump_acc<1, 1, 0>(acc[cl], rsrc_p, tid);
      }
      return;
    }
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (64 * 64)), 0, SLAB, 0x00020000);
      dump_acc<1, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_p, tid);
    }
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)(slot - ks) * 2 + cl) * (64 * 64)), 0, nsp * 2 * SLAB, 0x00020000);
      sum_slabs<1, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_t, nsp, 2 * SLAB, tid);
    }
  }
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) emit_tile<64, 64, MODE_CONVT>(p, acc[cl], tile_m, tile_n, 2 * ph + cl, b, tid);
#endif
}

// ---- the conv-transpose halo kernel through the six-product bf16 split (convt_halo_kernel x conv_halo_x3_kernel; r04) ----------
// At native fp32 the two-class halo form lost to the tap kernel (above): the tap kernel's k-loop has no VALU and five workgroups
// per CU.  The tap kernel cannot split its operands (both arrive by DMA), this one stages the patch through registers anyway:
// with 192 instead of 512 matrix cycles per 16 channels it wins (conv8_1 197 -> ... us, see profiles/r04_*), and its sources
// need no ln_apply launch.
#ifndef MSI_CT_MAXW
#define MSI_CT_MAXW 8
#endif
#ifndef MSI_CT3_NSTG   // weight ring of the six-product conv-transpose kernel: 2 (r05: 48.4 KB of LDS, three workgroups per CU; eight k-steps per chunk, so the
#define MSI_CT3_NSTG 2 // stage of k-step J is the literal J & 1 and the DMA of k-step J + 1 goes out at the head of k-step J) or 3 (r04: 60.7 KB, two per CU)
#endif
template <int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, MSI_CT_MAXW)))
convt_halo_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeomX3<1, (NP == 3 ? MSI_CT3_NSTG : 3), NP> G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD, NSTG = G::NSTG, PD = NSTG - 1;
  static_assert(NSTG == 3 || NSTG == 2, "prefetch distance two or one");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int CH = p.cpt0 + p.cpt1;                         // 32-channel chunks of both sources
  int t, c0 = 0, c1 = CH, ks = 0, slot = 0;
  {
    const int bid = blockIdx.x;
    if (bid < p.nb_main && p.split0 == 1) {
      const int q = p.n_main >> 3, r = p.n_main & 7, xcd = bid & 7, local = bid >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    } else {
      int sp, r, tbase;
      unsigned mg;
      if (bid < p.nb_main) { sp = p.split0; mg = p.mg_sp0; r = bid; tbase = 0; }
      else { sp = p.split; mg = p.mg_sp; r = bid - p.nb_main; tbase = p.n_main; }
      const int tl = (int)udiv_magic((unsigned)r, (unsigned)sp, mg);
      ks = r - tl * sp;
      t = tbase + tl;
      c0 = (int)udiv_magic((unsigned)(ks * CH), (unsigned)sp, mg);
      c1 = (int)udiv_magic((unsigned)((ks + 1) * CH), (unsigned)sp, mg);
      slot = bid - (p.split0 == 1 ? p.nb_main : 0);
    }
  }
  const bool full = (c0 == 0) & (c1 == CH);
  int ph, tile_m, tile_n, b;
  {   // row parity fastest (p.nclass = 2 here), then M tiles, N tiles, samples
    int r = t;
    ph = r & 1; r >>= 1;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * 4, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win;

  // the first two weight k-steps (class pw = 0, taps 0 and 1 of chunk c0) go out before the patch addresses are worked out
  const int S = p.ksteps;                                 // k-steps per class: 4 CH
  // (weights: the x3 block, [class][tap * CH + c][plane h | m | l][npad][64 B] -- see conv_halo_x3_kernel)
  const int plane_bytes = p.npad * G::B_ROW;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk_x3, 0, (int)((size_t)4 * S * NP * plane_bytes), 0x00020000);
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + (lane >> 2)) * G::B_ROW + (lane & 3) * 16);
#define MSI_B_ISSUE(cls, tap, c, st)                                                                                   \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * 16 * G::B_ROW;                                          \
    const int soff_ = ((cls) * S + (tap) * CH + (c)) * NP * plane_bytes;                                               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0); \
    if (NP == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0); \
  }
  MSI_B_ISSUE(2 * ph, 0, c0, 0)
  if (NSTG == 3) MSI_B_ISSUE(2 * ph, 1, c0, 1)

  // ---- per-lane patch elements (as conv_halo_kernel; the byte offset depends on the source's channel count) ----
  unsigned pixi[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - 1 + py;
    int iw = ow0 - 1 + px;
    bool cok = iw >= 0 && iw < W;                                  // SAME: zeros outside
    if (p.wrap) {   // msi_train_net: GEMM column mw reads PADDED column mw - v of wrap_pad(x, 2, 2), valid in [0, W + 4): image column (. - 2) mod W
      cok = iw >= 0 && iw < W + 4;
      iw -= 2;
      iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);
    }
    pok[k] = pp < NPX && ih >= 0 && ih < H && cok;
    pixi[k] = (unsigned)(ih * W + iw);
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 8) : 0xffffffffu;
  }
  const size_t in0 = (size_t)H * W * p.C0 * 4, in1 = (size_t)H * W * p.C1 * 4;
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in0), 0, (int)in0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x1 + (size_t)b * in1), 0, (int)(in1 ? in1 : 16), 0x00020000);

  // LayerNorm of the raw sources: mean as hi + lo floats and 1 / sigma per source
  float inv_f[2] = {1.f, 1.f}, mu_hi[2] = {0.f, 0.f}, mu_lo[2] = {0.f, 0.f};
  bool has_pad = false;
  {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NLOAD], g4, be4;
  unsigned amax_ = 0u;
  int src_ld = 0;                                         // source of the patch held in araw
  // patch of chunk c -> registers (+ gamma / beta of the lane's channels when that source is raw)
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    const int s_ = (c) >= p.cpt0 ? 1 : 0, cc_ = s_ ? (c) - p.cpt0 : (c);                                               \
    const unsigned cb_ = (unsigned)((s_ ? p.C1 : p.C0) * 4);                                                           \
    src_ld = s_;                                                                                                       \
    if (s_ == 0) {                                                                                                     \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a0, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));           \
    } else {                                                                                                           \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a1, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));           \
    }                                                                                                                  \
    const float *gp_ = (s_ ? p.ln_gamma1 : p.ln_gamma), *bp_ = (s_ ? p.ln_beta1 : p.ln_beta);                          \
    if ((p.halo_apply >> s_) & 1) {                                                                                    \
      g4 = *reinterpret_cast<const v4f *>(gp_ + cc_ * 32 + cslot * 4);                                                 \
      be4 = *reinterpret_cast<const v4f *>(bp_ + cc_ * 32 + cslot * 4);                                                \
    } else {   /* (same number of VMEM operations on both paths: the vmcnt arithmetic of the k-steps counts them) */   \
      g4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16);                                                         \
      be4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16 + 128);                                                  \
    }                                                                                                                  \
  }
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    const bool ap_ = (p.halo_apply >> src_ld) & 1;                                                                     \
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};                                                          \
    if (ap_) {                                                                                                         \
      const float ih_ = src_ld ? inv_f[1] : inv_f[0], mh_ = src_ld ? mu_hi[1] : mu_hi[0], ml_ = src_ld ? mu_lo[1] : mu_lo[0]; \
      s4 = ih_ * g4;                                                                                                   \
      const v4f nh = {-mh_, -mh_, -mh_, -mh_}, nl = {-ml_, -ml_, -ml_, -ml_};                                          \
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f y = araw[k_];                                                                                                \
      if (ap_) {                                                                                                       \
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});                  \
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};                                                          \
      }                                                                                                                \
      split_store<NP>(smem, lds_a[k_], y, amax_);                                                                      \
    }                                                                                                                  \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  // fragment base of tap row th = 0 (patch row 1 + local row) and of th = 1 (one row up for ph = 0, one down for ph = 1)
  const unsigned a_base0 = lds_base + (unsigned)((1 + 2 * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 16);
  // (msi_train_net's VALID form: tap 1 is the row ABOVE / the column to the LEFT in both parities -- tap_delta)
  const unsigned a_base1 = (ph && !p.wrap) ? a_base0 + G::ROW_PITCH : a_base0 - G::ROW_PITCH;
  const unsigned wadj = p.wrap ? 2u * G::PIX_BYTES : 0u;
  unsigned b_s[2];
  (void)fswz;
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
    b_s[s_] = lds_base + G::A_BYTES + (wn * 32 + frow) * G::B_ROW + (((2 * s_ + fh) ^ ((frow >> 2) & 3)) << 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  f32x16 acc[2][1][1], acc_lo[2];
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cl][0][0][r] = acc_lo[cl][r] = 0.f;

  // k-step J of the chunk = class pw = J / 4, tap (th, tw) = ((J / 2) & 1, J & 1), weights in ring stage st (run-time).
  // The DMA of the k-step PD = 2 ahead and (J == 0) the next chunk's patch loads are issued after the first MFMA quarter;
  // before the closing barrier the NEXT k-step's weights must have landed: they were issued one k-step ago, so only what
  // THIS k-step issued (2 DMA, + the patch loads of J == 0) may still be in flight (in-order return).
  constexpr int NPLD = NLOAD + 2;                         // VMEM operations of a patch load
#define MSI_CTSTEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int PWC_ = (J) >> 2, TH_ = ((J) >> 1) & 1, TW_ = (J) & 1;                                                \
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */                \
    const unsigned ab_ = (TH_ ? a_base1 : a_base0) - ((PWC_ && TW_) ? wadj : 0u);                                      \
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];                                                                \
    const unsigned bst_ = (unsigned)st * G::B_STAGE;                                                                   \
    bool issued_ = false;                                                                                              \
    if (NSTG == 2) {   /* the NEXT k-step's weights into the other stage (last read in the previous k-step: closing barrier passed) */ \
      constexpr int JN_ = ((J) + 1) & 7;                                                                               \
      if ((J) + 1 < 8) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c, st ^ 1) }                        \
      else if (c + 1 < c1) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, st ^ 1) }                \
    }                                                                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      ah_[s_] = s_ == 0 ? lds_read128<COFF_>(ab_) : lds_read128<COFF_ + 32>(ab_);                                      \
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);                                                                        \
      am_[s_] = s_ == 0 ? lds_read128<COFF_ + 64>(ab_) : lds_read128<COFF_ + 96>(ab_);                                 \
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);                                                               \
      if (NP == 3) {                                                                                                   \
        al_[s_] = s_ == 0 ? lds_read128<COFF_ + 128>(ab_) : lds_read128<COFF_ + 160>(ab_);                             \
        bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);                                                         \
      } else { al_[s_] = ah_[s_]; bl_[s_] = bh_[s_]; }                                                                 \
    }                                                                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                 \
      if (NP == 3) {                                                                                                   \
        if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);                                    \
        else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);                                            \
      } else {                                                                                                         \
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);                                                    \
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);                                                            \
      }                                                                                                                \
      split_mfma<NP>(acc[PWC_][0][0], acc_lo[PWC_], ah_[s_], am_[s_], al_[s_], bh_[s_], bm_[s_], bl_[s_]);             \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
      if (s_ == 0) {                                                                                                   \
        if ((J) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                              \
        if (NSTG == 3) {                                                                                               \
        int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;                                                       \
        constexpr int JN_ = ((J) + PD) & 7;                                                                            \
        if ((J) + PD < 8) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_) }                        \
        else if (c + 1 < c1) { issued_ = true; MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_) }                 \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
    if (NSTG == 2) {   /* (the DMA went out before this k-step's patch loads: in-order return) */                      \
      if ((J) == 0 && c + 1 < c1) wait_vmcnt<NPLD>();                                                                  \
      else wait_vmcnt<0>();                                                                                            \
    } else if ((J) == 0 && c + 1 < c1) wait_vmcnt<NP + NPLD>();                                                          \
    else if (issued_) wait_vmcnt<NP>();                                                                                \
    else wait_vmcnt<0>();                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                      \
    st = st + 1 == NSTG ? 0 : st + 1;                                                                                  \
  }

  // ---- prologue: first patch (the first two weight k-steps are on their way), the sources' LayerNorm statistics ----
  int c = c0, st = 0;
  MSI_PATCH_LOAD(c0)
  if (p.halo_apply) {
    double *s_stat = reinterpret_cast<double *>(smem);
    if (p.halo_apply & 1) {
      ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv_f[0] = (float)s_stat[1]; mu_hi[0] = (float)mu; mu_lo[0] = (float)(mu - (double)mu_hi[0]);
      __syncthreads();
    }
    if (p.halo_apply & 2) {
      ln_mean_inv(p.ln_sums1 + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n1, p.ln_scl_src1, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv_f[1] = (float)s_stat[1]; mu_hi[1] = (float)mu; mu_lo[1] = (float)(mu - (double)mu_hi[1]);
      __syncthreads();
    }
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    MSI_CTSTEP(0) MSI_CTSTEP(1) MSI_CTSTEP(2) MSI_CTSTEP(3) MSI_CTSTEP(4) MSI_CTSTEP(5) MSI_CTSTEP(6) MSI_CTSTEP(7)
    if (c + 1 < c1) {
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_CTSTEP
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
  split_finish<NP>(acc[0][0][0], acc_lo[0], amax_, lane, p.status);
  split_finish<NP>(acc[1][0][0], acc_lo[1], amax_, lane, p.status);

  // ---- epilogue: two class tiles ----
  if (!full) {
    constexpr int SLAB = 64 * 64 * 4;
    if (p.tile_cnt == nullptr) {                          // separate fix-up launch (conv_fixup_kernel, class = blockIdx.y)
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (64 * 64)), 0, SLAB, 0x00020000);
        dump_acc<1, 1, 0>(acc[cl], rsrc_p, tid);
      }
      return;
    }
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (64 * 64)), 0, SLAB, 0x00020000);
      dump_acc<1, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_p, tid);
    }
    const int nsp = t < p.n_main ? p.split0 : p.split;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    handoff_release();
    __syncthreads();
    int *s_old = reinterpret_cast<int *>(smem);
    if (tid == 0)
      *s_old = __hip_atomic_fetch_add(p.tile_cnt + (t - (p.split0 == 1 ? p.n_main : 0)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_old != nsp - 1) return;
    handoff_acquire();
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)(slot - ks) * 2 + cl) * (64 * 64)), 0, nsp * 2 * SLAB, 0x00020000);
      sum_slabs<1, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_t, nsp, 2 * SLAB, tid);
    }
  }
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) emit_tile<64, 64, MODE_CONVT>(p, acc[cl], tile_m, tile_n, 2 * ph + cl, b, tid);
#endif
}

// ---- halo-patch kernel, bf16 operands ----------------------------------------------------------------------------
// Same idea as conv_halo_kernel at the shapes the 16x faster bf16 MFMA needs: at 4 MFMAs per wave and k-step the
// 64x64 tile cannot be fed (the tap kernel's bf16 instantiations are bound by their L2 -> LDS traffic: 32 KB per k-step
// of a 128x128 tile, half of it the pixels' nine tap fetches), so a workgroup owns (BM / 16) x 16 output pixels x BN
// channels with BM x BN = 128 x 128 (Cout in multiples of 128) or 256 x 64 (the full-resolution Cout = 64 layers), a
// wave 32 MT x 32 NT of it (16 MFMAs = 512 matrix cycles per k-step), the chunk is 64 channels (the 128-byte rows of
// the packed weights, one k-step per tap), and the patch is staged through registers once per chunk:
//   APPLY = 0: from the bf16 operand copy (the network input, or what ln_apply wrote),
//   APPLY = 1: from the producer's RAW output (fp16 of x * 2^-e, see emit_tile_impl RAW16), its LayerNorm + ReLU applied and rounded to bf16 (round to nearest
//              even, v_cvt_pk_bf16_f32) on the way -- the producer then has no ln_apply launch and no bf16 copy.
// Weights: NSTG-stage DMA ring of BN rows (three stages where two workgroups per CU still fit, else two), the stage
// index is a run-time scalar (4 VALU adds per 512-cycle k-step).  Whole tiles only (big grids: no K split).
template <int BM, int BN, int RATE>
struct HaloGeomB {
  static constexpr int TH = BM / 16;
  static constexpr int PW = 16 + 2 * RATE, PH = TH + 2 * RATE, NPX = PW * PH;
  static constexpr int PIX_BYTES = 144;                   // 64 bf16 channels + 16 bytes: 16 consecutive pixels -> 16 distinct 16-byte bank groups
  // A ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): lanes 20-27 are pixels
  // 4-11 of the block's SECOND row, which take exactly the bank groups pixels 0-3, 12-15 of the first row leave free
  // iff the row pitch is a multiple of 256 bytes (measured with PW * 144: SQ_LDS_BANK_CONFLICT = 31 % of the LDS cycles)
  static constexpr int ROW_PITCH = (PW * PIX_BYTES + 255) / 256 * 256;
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_STAGE = BN * ROW_BYTES;
  static constexpr int NSTG = (A_BYTES + 3 * B_STAGE <= 80 * 1024) ? 3 : 2;
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;     // 8-channel patch slots per thread and chunk
};

#ifndef MSI_HALO_ABLATE   // timing experiments only (tools/_variants): 1 no weight DMA, 2 no patch traffic, 4 no k-step barrier, 8 no fragment reads
#define MSI_HALO_ABLATE 0
#endif
template <int BM, int BN, int RATE, int APPLY, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2)))
conv_halo_bf16_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int ABL = MSI_HALO_ABLATE;
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  typedef HaloGeomB<BM, BN, RATE> G;
  constexpr int NTHR = 64 * NW, WR = NW / 2;                   // NW = 4 or 8 waves in WR x 2: a wave owns 32 MT x 32 NT of the tile
  constexpr int R = RATE, PW = G::PW, NPX = G::NPX, NLOAD = (NPX * 8 + NTHR - 1) / NTHR, MT = BM / (32 * WR), NT = BN / 64;
  constexpr int NSTG = G::NSTG, PD = NSTG - 1, BI = BN / (8 * NW);   // BI: weight DMA instructions per wave and k-step (8 rows each)
  static_assert(BI == 1 || BI == 2 || BI == 4, "weight rows per wave");
  constexpr int NRAW = NLOAD;                                   // 16-byte patch loads per thread and chunk (bf16 copy, or fp16 raw)
  constexpr int NPL = NRAW;                                     // VMEM operations of a patch load
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int CH = p.cpt0;                                  // 64-channel chunks of the input
  int t;
  {   // XCD x works through the x-th eighth of the tiles (M tiles fastest: neighbours share halo rows and weights in its L2)
    const int bid = blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, local = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  // (requested here, used after the index arithmetic and the patch requests: the lane's shard of the source's LayerNorm sums, and
  // the layer's own window exponent for the epilogue -- neither round trip is then waited for where it is needed)
  LnShard shard = {0, 0};
  if (APPLY) shard = ln_shard_load(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, tid);
  const float raw_mul_pre = (float)(p.ln_scl[0] * (1.0 / 16777216.0));   // 2^-e (scalar load)
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * G::TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win, C = p.C0;
  constexpr int ESZ = 2;                                  // bytes per source element: the bf16 operand copy, or (APPLY) the producer's fp16 raw output

  // ---- per-lane patch slots: e = tid + 256 k -> patch pixel e / 8, 8-channel slot e % 8 (= tid % 8) ----
  unsigned voff[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
  constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + NTHR * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - R + py;
    int iw = ow0 - R + px;
    if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);
    pok[k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
    voff[k] = pok[k] ? (unsigned)(ih * W + iw) * (unsigned)(C * ESZ) + (unsigned)(cslot * 8 * ESZ) : OOB;
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
  }
  const size_t in_bytes = (size_t)H * W * C * ESZ;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)(in_bytes < 0xfffffff0u ? in_bytes : 0xfffffff0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)p.ksteps * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * BN + wave * (BN / NW) + drow) * ROW_BYTES + dslot * 16);

  int c_ld = 0;                                           // chunk of the patch held in araw
  bool has_pad = false;
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NRAW];   // (eight bf16 operands, or eight fp16 raw values, per 16-byte slot)
  float *s_tab = reinterpret_cast<float *>(smem + G::LDS_BYTES);   // APPLY: scale[C] | shift[C] of the source's LayerNorm
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[k_], (c) * 128, 0)); \
    if (APPLY) c_ld = (c);                                                                                             \
  }
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    v4f s_[2], t_[2];                                                                                                  \
    if (APPLY) {   /* the thread's eight channels of chunk c_ld: four ds_read_b128 from the table built in the prologue */ \
      const float *sp_ = s_tab + c_ld * 64 + cslot * 8;                                                                \
      s_[0] = *reinterpret_cast<const v4f *>(sp_); s_[1] = *reinterpret_cast<const v4f *>(sp_ + 4);                    \
      t_[0] = *reinterpret_cast<const v4f *>(sp_ + C); t_[1] = *reinterpret_cast<const v4f *>(sp_ + C + 4);            \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f o_;                                                                                                          \
      if (APPLY) {                                                                                                     \
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};                                                                           \
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));                                                     \
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);   /* eight fp16 raw values x * 2^-e (s_ carries 2^e) */    \
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};                                  \
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};                                  \
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);                          \
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);                          \
        if (has_pad && !pok[k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */                 \
        unsigned w0, w1, w2, w3;                                                                                       \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));                                         \
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});                                                         \
      } else {                                                                                                         \
        o_ = araw[k_];                                                                                                 \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;                                   \
    }                                                                                                                  \
  }
  // weights of k-step (chunk c, tap) -> ring stage st (run-time); the packed blob is tap-major: row block tap * CH + c
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * (BN / NW) * ROW_BYTES;                                  \
    const int soff_ = ((tap) * CH + (c)) * p.npad * ROW_BYTES;                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    if (BI >= 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0); \
    if (BI == 4) {                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);         \
    }                                                                                                                  \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((wm * (MT * 2) + (frow >> 4)) * G::ROW_PITCH + (frow & 15) * G::PIX_BYTES + fh * 16);
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * (NT * 32) + frow) * ROW_BYTES + (((2 * q + fh) ^ fswz) << 4);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

  // one k-step = tap TAP of the current chunk with the weights in ring stage st.  All 4 (MT + NT) fragments are fetched
  // first; quarter q waits for its own; the next chunk's patch loads (tap 0) and the DMA of the k-step PD ahead are issued
  // after the first quarter.  Before the closing barrier the NEXT k-step's weights must have landed: with PD = 2 they
  // were issued one k-step ago, and only what this k-step issued may still be in flight (in-order return).
#define MSI_HQ(Q)                                                                                                      \
  wait_lgkm_frag<(3 - (Q)) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                  \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),                    \
                                                            __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[i_][j_], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
#define MSI_HTAP(TAP)                                                                                                  \
  {                                                                                                                    \
    constexpr int KH_ = (TAP) / 3, KW_ = (TAP) % 3;                                                                    \
    constexpr int AOFF_ = KH_ * R * G::ROW_PITCH + KW_ * R * G::PIX_BYTES;                                             \
    constexpr int AROW_ = 2 * G::ROW_PITCH;   /* next 32-pixel block: two patch rows down */                           \
    v4f fa_[4][MT], fb_[4][NT];                                                                                        \
    const unsigned bst_ = (unsigned)st * G::B_STAGE;                                                                   \
    if (ABL & 8) {                                                                                                     \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) asm volatile("" : "=v"(fa_[q_][i_]));                        \
        _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) asm volatile("" : "=v"(fb_[q_][j_]));                        \
      }                                                                                                                \
    } else                                                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      const unsigned ba_ = b_q[q_] + bst_;                                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                \
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 32>(a_base)      \
                                : q_ == 2 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base))         \
                    : i_ == 1 ? (q_ == 0 ? lds_read128<AOFF_ + AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + AROW_ + 32>(a_base) \
                                : q_ == 2 ? lds_read128<AOFF_ + AROW_ + 64>(a_base) : lds_read128<AOFF_ + AROW_ + 96>(a_base)) \
                    : i_ == 2 ? (q_ == 0 ? lds_read128<AOFF_ + 2 * AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 2 * AROW_ + 32>(a_base) \
                                : q_ == 2 ? lds_read128<AOFF_ + 2 * AROW_ + 64>(a_base) : lds_read128<AOFF_ + 2 * AROW_ + 96>(a_base)) \
                              : (q_ == 0 ? lds_read128<AOFF_ + 3 * AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 3 * AROW_ + 32>(a_base) \
                                : q_ == 2 ? lds_read128<AOFF_ + 3 * AROW_ + 64>(a_base) : lds_read128<AOFF_ + 3 * AROW_ + 96>(a_base)); \
      _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                \
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);                                \
    }                                                                                                                  \
    MSI_HQ(0)                                                                                                          \
    bool issued_;                                                                                                      \
    {                                                                                                                  \
      if (!(ABL & 2) && (TAP) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                \
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;                                                         \
      issued_ = ((TAP) + PD < 9) || (c + 1 < c1);                                                                      \
      if (ABL & 1) { }                                                                                                 \
      else if ((TAP) + PD < 9) { MSI_B_ISSUE(c, (TAP) + PD, sn_) }                                                     \
      else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, (TAP) + PD - 9, sn_) }                                                 \
    }                                                                                                                  \
    MSI_HQ(1) MSI_HQ(2) MSI_HQ(3)                                                                                      \
    if (ABL & 3) {                                                                                                     \
      wait_vmcnt<0>();                                                                                                 \
    } else if (PD == 2) {                                                                                              \
      if ((TAP) == 0 && c + 1 < c1) wait_vmcnt<BI + NPL>();                                                            \
      else if (issued_) wait_vmcnt<BI>();                                                                              \
      else wait_vmcnt<0>();                                                                                            \
    } else {                                                                                                           \
      wait_vmcnt<0>();                                                                                                 \
    }                                                                                                                  \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                                                      \
    st = st + 1 == NSTG ? 0 : st + 1;                                                                                  \
  }

  // ---- prologue: first patch, first PD weight k-steps ----
  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  MSI_STAMP(6)
  MSI_PATCH_LOAD(c0)
  MSI_B_ISSUE(c0, 0, 0)
  if (PD == 2) MSI_B_ISSUE(c0, 1, 1)
  MSI_STAMP(7)
  if (APPLY) {
    // the affine of the source's LayerNorm for every input channel, once per workgroup: scale = 2^e inv gamma (the stored
    // raw value is fp16 of x * 2^-e), shift = beta - mean inv gamma with the mean as hi + lo floats, fp32 operations only
    // (the expressions ln_apply's fp32 table would give up to the last bit are not needed: the result is rounded to bf16)
    double *s_stat = reinterpret_cast<double *>(smem);
    // (gamma / beta of the thread's <= 2 channels are requested BEFORE the statistics' round trip, not after it)
    const int ch0 = tid < C ? tid : 0, ch1 = tid + NTHR < C ? tid + NTHR : 0;
    const float g0 = p.ln_gamma[ch0], b0 = p.ln_beta[ch0], g1 = p.ln_gamma[ch1], b1 = p.ln_beta[ch1];
    ln_mean_inv_pre(shard, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    MSI_STAMP(8)
    const double mu = s_stat[0];
    const float inv_f = (float)s_stat[1], mu_hi = (float)mu, mu_lo = (float)(mu - (double)mu_hi);
    const float up_f = (float)(p.ln_scl_src[2] * 16777216.0);   // 2^e of the source layer's window
    __syncthreads();
    if (tid < C) {
      const float su = inv_f * g0;
      s_tab[C + tid] = __builtin_fmaf(-mu_lo, su, __builtin_fmaf(-mu_hi, su, b0));
      s_tab[tid] = up_f * su;
    }
    if (tid + NTHR < C) {
      const float su = inv_f * g1;
      s_tab[C + tid + NTHR] = __builtin_fmaf(-mu_lo, su, __builtin_fmaf(-mu_hi, su, b1));
      s_tab[tid + NTHR] = up_f * su;
    }
    __syncthreads();
  }
  MSI_STAMP(9)
  wait_vmcnt<0>();
  MSI_STAMP(10)
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    MSI_HTAP(0) MSI_HTAP(1) MSI_HTAP(2) MSI_HTAP(3) MSI_HTAP(4) MSI_HTAP(5) MSI_HTAP(6) MSI_HTAP(7) MSI_HTAP(8)
    if (c + 1 < c1 && !(ABL & 2)) {
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_HTAP
#undef MSI_HQ
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  emit_tile<BM, BN, MODE_CONV, 1, WR>(p, acc, tile_m, tile_n, 0, b, tid, smem, raw_mul_pre);   // (the k-loop ended with a barrier: LDS is free)
#ifdef MSI_CONV_TIMING
  if (p.dbg && tid == 0) {
    unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
    o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
#endif
#endif
}

// ---- halo-patch kernel, stride 2, bf16 operands (r03) ----------------------------------------------------------------
// conv_halo_s2_kernel's parity-plane units (four (8 + 1) x (16 + 1)-pixel patches per 64-channel group, 4 + 2 + 2 + 1 taps) with
// conv_halo_bf16_kernel's machinery: 8 x 16 output pixels x 128 channels per workgroup, NW waves of 32 MT x 64 channels, the patch
// staged through registers from the bf16 operand copy or (APPLY) from the producer's raw fp16 output with its LayerNorm + ReLU +
// bf16 rounding on the way (per-channel affine table in LDS), weights through the three-stage DMA ring, whole tiles only.  The
// bf16 tap kernel ran these three layers at 19-29 % of the peak AND kept the ln_apply launches of conv1_1 / conv2_1 / conv3_2 alive
// (1.5 GB of HBM round trips per 16 frames).
struct HaloGeomBS2 {
  static constexpr int TH = 8;
  static constexpr int PW = 17, PH = TH + 1, NPX = PW * PH;
  static constexpr int PIX_BYTES = 144;
  static constexpr int ROW_PITCH = (PW * PIX_BYTES + 255) / 256 * 256;   // (a multiple of 256: see HaloGeomB)
  static constexpr int A_BYTES = PH * ROW_PITCH;
  static constexpr int B_STAGE = 128 * ROW_BYTES;
  static constexpr int NSTG = 3;
  static constexpr int LDS_BYTES = A_BYTES + NSTG * B_STAGE;
};

template <int APPLY, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2)))
conv_halo_bf16_s2_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeomBS2 G;
  constexpr int BM = 128, BN = 128;
  constexpr int NTHR = 64 * NW, WR = NW / 2;                   // NW = 4 or 8 waves in WR x 2: a wave owns 32 MT x 32 NT of the tile
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = (NPX * 8 + NTHR - 1) / NTHR, MT = BM / (32 * WR), NT = BN / 64;
  constexpr int NSTG = G::NSTG, PD = NSTG - 1, BI = BN / (8 * NW);   // BI: weight DMA instructions per wave and k-step (8 rows each)
  static_assert(BI == 1 || BI == 2 || BI == 4, "weight rows per wave");
  constexpr int NRAW = NLOAD;                                   // 16-byte patch loads per thread and chunk (bf16 copy, or fp16 raw)
  constexpr int NPL = NRAW;                                     // VMEM operations of a patch load
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int CH = p.cpt0;                                  // 64-channel chunks of the input
  int t;
  {   // XCD x works through the x-th eighth of the tiles (M tiles fastest: neighbours share halo rows and weights in its L2)
    const int bid = blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, local = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int tile_m, tile_n, b;
  {
    int r = t;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const float raw_mul_pre = (float)(p.ln_scl[0] * (1.0 / 16777216.0));   // 2^-e (scalar load)
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * G::TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win, C = p.C0;
  constexpr int ESZ = 2;                                  // bytes per source element: the bf16 operand copy, or (APPLY) the producer's fp16 raw output

  // ---- per-lane patch slots: e = tid + 256 k -> patch pixel e / 8, 8-channel slot e % 8 (= tid % 8) ----
  unsigned voff[4][NLOAD], lds_a[NLOAD];
  bool pok[4][NLOAD];
  const int cslot = tid & 7;
  constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + NTHR * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // unit u: the parity plane of taps kh = (u >> 1) (+ 2), kw = (u & 1) (+ 2)
      const int ih = 2 * (oh0 + py) + (u >> 1) - p.pad_t;
      int iw = 2 * (ow0 + px) + (u & 1) - p.pad_l;
      if (p.wrap) iw = iw < 0 ? iw + W : (iw >= W ? iw - W : iw);
      pok[u][k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;
      voff[u][k] = pok[u][k] ? (unsigned)(ih * W + iw) * (unsigned)(C * ESZ) + (unsigned)(cslot * 8 * ESZ) : OOB;
    }
  }
  const size_t in_bytes = (size_t)H * W * C * ESZ;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in_bytes), 0, (int)(in_bytes < 0xfffffff0u ? in_bytes : 0xfffffff0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)p.ksteps * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * BN + wave * (BN / NW) + drow) * ROW_BYTES + dslot * 16);

  int c_ld = 0;                                           // chunk of the patch held in araw
  bool has_pad = false;
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) bad |= (lds_a[k] != 0xffffffffu) && !pok[u][k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  v4f araw[NRAW];   // (eight bf16 operands, or eight fp16 raw values, per 16-byte slot)
  float *s_tab = reinterpret_cast<float *>(smem + G::LDS_BYTES);   // APPLY: scale[C] | shift[C] of the source's LayerNorm
#define MSI_PATCH_LOAD(c, U)                                                                                           \
  {                                                                                                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                               \
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[U][k_], (c) * 128, 0)); \
    if (APPLY) c_ld = (c);                                                                                             \
  }
#define MSI_PATCH_STORE(U)                                                                                             \
  {                                                                                                                    \
    v4f s_[2], t_[2];                                                                                                  \
    if (APPLY) {   /* the thread's eight channels of chunk c_ld: four ds_read_b128 from the table built in the prologue */ \
      const float *sp_ = s_tab + c_ld * 64 + cslot * 8;                                                                \
      s_[0] = *reinterpret_cast<const v4f *>(sp_); s_[1] = *reinterpret_cast<const v4f *>(sp_ + 4);                    \
      t_[0] = *reinterpret_cast<const v4f *>(sp_ + C); t_[1] = *reinterpret_cast<const v4f *>(sp_ + C + 4);            \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f o_;                                                                                                          \
      if (APPLY) {                                                                                                     \
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};                                                                           \
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));                                                     \
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);   /* eight fp16 raw values x * 2^-e (s_ carries 2^e) */    \
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};                                  \
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};                                  \
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);                          \
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);                          \
        if (has_pad && !pok[U][k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */                 \
        unsigned w0, w1, w2, w3;                                                                                       \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));                                         \
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});                                                         \
      } else {                                                                                                         \
        o_ = araw[k_];                                                                                                 \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;                                   \
    }                                                                                                                  \
  }
  // weights of k-step (chunk c, tap) -> ring stage st (run-time); the packed blob is tap-major: row block tap * CH + c
#define MSI_B_ISSUE(c, tap, st)                                                                                        \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * (BN / NW) * ROW_BYTES;                                  \
    const int soff_ = ((tap) * CH + (c)) * p.npad * ROW_BYTES;                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    if (BI >= 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0); \
    if (BI == 4) {                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);         \
    }                                                                                                                  \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((wm * (MT * 2) + (frow >> 4)) * G::ROW_PITCH + (frow & 15) * G::PIX_BYTES + fh * 16);
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * (NT * 32) + frow) * ROW_BYTES + (((2 * q + fh) ^ fswz) << 4);
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

  // one k-step = tap TAP of the current chunk with the weights in ring stage st.  All 4 (MT + NT) fragments are fetched
  // first; quarter q waits for its own; the next chunk's patch loads (tap 0) and the DMA of the k-step PD ahead are issued
  // after the first quarter.  Before the closing barrier the NEXT k-step's weights must have landed: with PD = 2 they
  // were issued one k-step ago, and only what this k-step issued may still be in flight (in-order return).
#define MSI_HQ(Q)                                                                                                      \
  wait_lgkm_frag<(3 - (Q)) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                  \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),                    \
                                                            __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[i_][j_], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
#define MSI_S2_TAP(J) ((J) == 0 ? 0 : (J) == 1 ? 2 : (J) == 2 ? 6 : (J) == 3 ? 8 : (J) == 4 ? 1 : (J) == 5 ? 7 : (J) == 6 ? 3 : (J) == 7 ? 5 : 4)
#define MSI_S2_UNIT(J) ((J) < 4 ? 0 : (J) < 6 ? 1 : (J) < 8 ? 2 : 3)
  // k-step J = 0..8 of the current 64-channel group: unit, tap and patch offsets are literals (conv_halo_s2_kernel's order)
#define MSI_S2STEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int TAP_ = MSI_S2_TAP(J), U_ = MSI_S2_UNIT(J);                                                           \
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;                                                        \
    constexpr bool FIRST_ = (J) == 0 || (J) == 4 || (J) == 6 || (J) == 8, LAST_ = (J) == 3 || (J) == 5 || (J) == 7 || (J) == 8; \
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;                                                     \
    constexpr int AROW_ = 2 * G::ROW_PITCH;   /* next 32-pixel block: two patch rows down */                           \
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */                               \
    v4f fa_[4][MT], fb_[4][NT];                                                                                        \
    const unsigned bst_ = (unsigned)st * G::B_STAGE;                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      const unsigned ba_ = b_q[q_] + bst_;                                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                \
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 32>(a_base)      \
                                : q_ == 2 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base))         \
                              : (q_ == 0 ? lds_read128<AOFF_ + AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + AROW_ + 32>(a_base) \
                                : q_ == 2 ? lds_read128<AOFF_ + AROW_ + 64>(a_base) : lds_read128<AOFF_ + AROW_ + 96>(a_base)); \
      _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                \
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);                                \
    }                                                                                                                  \
    MSI_HQ(0)                                                                                                          \
    bool issued_;                                                                                                      \
    {                                                                                                                  \
      if (FIRST_ && more_) {                                                                                           \
        if (U_ < 3) MSI_PATCH_LOAD(c, (U_ + 1) & 3) else MSI_PATCH_LOAD(c + 1, 0)                                      \
      }                                                                                                                \
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;                                                         \
      issued_ = ((J) + PD < 9) || (c + 1 < c1);                                                                        \
      if ((J) + PD < 9) { MSI_B_ISSUE(c, MSI_S2_TAP(((J) + PD) % 9), sn_) }                                            \
      else if (c + 1 < c1) { MSI_B_ISSUE(c + 1, MSI_S2_TAP(((J) + PD) % 9), sn_) }                                     \
    }                                                                                                                  \
    MSI_HQ(1) MSI_HQ(2) MSI_HQ(3)                                                                                      \
    /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */ \
    if (FIRST_ && !LAST_ && more_) wait_vmcnt<BI + NPL>();                                                             \
    else if (issued_) wait_vmcnt<BI>();                                                                                \
    else wait_vmcnt<0>();                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                      \
    st = st + 1 == NSTG ? 0 : st + 1;                                                                                  \
    if (LAST_ && more_) {   /* every wave has read this unit's last tap: swap the patch */                             \
      MSI_PATCH_STORE((U_ + 1) & 3)                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
      __builtin_amdgcn_s_barrier();                                                                                    \
    }                                                                                                                  \
  }

  // ---- prologue: unit 0 of the first group, the first two weight k-steps (taps (0,0), (0,2)) ----
  static_assert(PD == 2 && MT <= 2, "three-stage weight ring; one or two 32-pixel blocks per wave");
  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  MSI_PATCH_LOAD(c0, 0)
  MSI_B_ISSUE(c0, MSI_S2_TAP(0), 0)
  MSI_B_ISSUE(c0, MSI_S2_TAP(1), 1)
  if (APPLY) {
    // the affine of the source's LayerNorm for every input channel, once per workgroup (as conv_halo_bf16_kernel)
    double *s_stat = reinterpret_cast<double *>(smem);
    const int ch0 = tid < C ? tid : 0, ch1 = tid + NTHR < C ? tid + NTHR : 0;
    const float g0 = p.ln_gamma[ch0], b0 = p.ln_beta[ch0], g1 = p.ln_gamma[ch1], b1 = p.ln_beta[ch1];
    ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
    const double mu = s_stat[0];
    const float inv_f = (float)s_stat[1], mu_hi = (float)mu, mu_lo = (float)(mu - (double)mu_hi);
    const float up_f = (float)(p.ln_scl_src[2] * 16777216.0);   // 2^e of the source layer's window
    __syncthreads();
    if (tid < C) {
      const float su = inv_f * g0;
      s_tab[C + tid] = __builtin_fmaf(-mu_lo, su, __builtin_fmaf(-mu_hi, su, b0));
      s_tab[tid] = up_f * su;
    }
    if (tid + NTHR < C) {
      const float su = inv_f * g1;
      s_tab[C + tid + NTHR] = __builtin_fmaf(-mu_lo, su, __builtin_fmaf(-mu_hi, su, b1));
      s_tab[tid + NTHR] = up_f * su;
    }
    __syncthreads();
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    MSI_S2STEP(0) MSI_S2STEP(1) MSI_S2STEP(2) MSI_S2STEP(3) MSI_S2STEP(4) MSI_S2STEP(5) MSI_S2STEP(6) MSI_S2STEP(7) MSI_S2STEP(8)
  }
#undef MSI_S2STEP
#undef MSI_S2_UNIT
#undef MSI_S2_TAP
#undef MSI_HQ
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
  emit_tile<BM, BN, MODE_CONV, 1, WR>(p, acc, tile_m, tile_n, 0, b, tid, smem, raw_mul_pre);   // (the k-loop ended with a barrier: LDS is free)
#endif
}

// ---- halo-patch kernel for the conv-transpose layers, bf16 operands ------------------------------------------------
// The bf16 tap kernel is bound by its L2 -> LDS traffic, and a conv-transpose fetches every input element four times per
// parity class.  Here a workgroup owns a (BM / 16) x 16 tile of the INPUT grid x BN channels for the TWO classes of one
// output-row parity ph (pw = 0, 1): per 64-channel chunk of either source of the skip concat it stages the halo patch
// once (from the bf16 operand copies) and runs 2 classes x 4 taps = 8 k-steps on it, each class into its own
// accumulators (2 x MT x NT tiles = 128 registers).  Class (ph, pw), tap (th, tw) reads input row mh + (ph ? th : -th)
// and column mw + (pw ? tw : -tw) (tap_delta): the row offset of th = 1 is the only run-time part of a fragment
// address (two base registers), columns are immediates.  The two workgroups of a tile (ph = 0, 1) are neighbours in
// the grid order (same XCD: the patch comes from HBM once).  Weights: the packed blob's [class][tap * CH + c] row
// blocks through the conv kernel's DMA ring.  Whole tiles only.  Unlike the fp32 attempt (convt_halo_kernel, slower than
// its tap kernel) this one replaces a kernel that is traffic-bound: configs[2] conv8_1 963 -> 537 us, conv7_1 558 -> 416,
// conv6_1 475 -> 390 per 16 frames (profiles/r02_T_bf16_convt_halo.txt).
// APPLY = 1 (r03): a source whose bit is set in p.halo_apply is read from its producer's RAW output (fp16 of x * 2^-e) with
// the producer's LayerNorm + ReLU + bf16 rounding applied while staging, as conv_halo_bf16_kernel<.., 1> does; the other
// source (if any) still comes from its bf16 operand copy.  Both encodings are 16 bytes per 8-channel slot.
template <int BM, int BN, int APPLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
convt_halo_bf16_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  typedef HaloGeomB<BM, BN, 1> G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD, MT = BM / 64, NT = BN / 64;
  constexpr int NSTG = G::NSTG, PD = NSTG - 1, BI = BN / 32;
  static_assert(NSTG == 3, "the k-step bookkeeping below assumes a prefetch distance of two");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int CH = p.cpt0 + p.cpt1;                         // 64-channel chunks of both sources
  int t;
  {
    const int bid = blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, local = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  int ph, tile_m, tile_n, b;
  {   // row parity fastest (p.nclass = 2 here), then M tiles, N tiles, samples
    int r = t;
    ph = r & 1; r >>= 1;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  const int tyi = (int)udiv_magic((unsigned)tile_m, (unsigned)p.halo_tx, p.mg_htx);
  const int oh0 = tyi * G::TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win;

  unsigned pixi[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
  constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - 1 + py, iw = ow0 - 1 + px;
    pok[k] = pp < NPX && ih >= 0 && ih < H && iw >= 0 && iw < W;   // SAME: zeros outside
    pixi[k] = (unsigned)(ih * W + iw);
    lds_a[k] = pp < NPX ? (unsigned)(py * G::ROW_PITCH + px * G::PIX_BYTES + cslot * 16) : 0xffffffffu;
  }
  const size_t in0 = (size_t)H * W * p.C0 * 2, in1 = (size_t)H * W * p.C1 * 2;
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x0 + (size_t)b * in0), 0, (int)in0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.x1 + (size_t)b * in1), 0, (int)(in1 ? in1 : 16), 0x00020000);
  const int S = p.ksteps;                                 // k-steps per class: 4 CH
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk, 0, (int)((size_t)4 * S * p.npad * ROW_BYTES), 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;
  const unsigned b_voff = (unsigned)((tile_n * BN + wave * (BN / 4) + drow) * ROW_BYTES + dslot * 16);

  v4f araw[NLOAD], g8[2], be8[2];
  int src_ld = 0;                                         // source of the patch held in araw
  float inv0 = 1.f, inv1 = 1.f, mh0 = 0.f, mh1 = 0.f, ml0 = 0.f, ml1 = 0.f, up0 = 1.f, up1 = 1.f;   // (scalars, not arrays: no scratch)
  bool has_pad = false;
  if (APPLY) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) bad |= (lds_a[k] != 0xffffffffu) && !pok[k];
    has_pad = __builtin_amdgcn_ballot_w64(bad) != 0;
  }
  constexpr int NPL = NLOAD + (APPLY ? 4 : 0);            // VMEM operations of a patch load
#define MSI_PATCH_LOAD(c)                                                                                              \
  {                                                                                                                    \
    const int s_ = (c) >= p.cpt0 ? 1 : 0, cc_ = s_ ? (c) - p.cpt0 : (c);                                               \
    const unsigned cb_ = (unsigned)((s_ ? p.C1 : p.C0) * 2);                                                           \
    src_ld = s_;                                                                                                       \
    if (APPLY) {   /* gamma / beta of the thread's 8 channels (dummy rows when this source is not raw: the vmcnt  */   \
      /* arithmetic of the k-steps counts the same number of VMEM operations on both paths)                      */   \
      const bool raw_ = (p.halo_apply >> s_) & 1;                                                                      \
      const float *gb_ = raw_ ? (s_ ? p.ln_gamma1 : p.ln_gamma) : reinterpret_cast<const float *>(p.wpk);              \
      const float *bb_ = raw_ ? (s_ ? p.ln_beta1 : p.ln_beta) : reinterpret_cast<const float *>(p.wpk);                \
      const float *gp_ = gb_ + (raw_ ? cc_ * 64 : 0) + cslot * 8, *bp_ = bb_ + (raw_ ? cc_ * 64 : 64) + cslot * 8;     \
      g8[0] = *reinterpret_cast<const v4f *>(gp_); g8[1] = *reinterpret_cast<const v4f *>(gp_ + 4);                     \
      be8[0] = *reinterpret_cast<const v4f *>(bp_); be8[1] = *reinterpret_cast<const v4f *>(bp_ + 4);                   \
    }                                                                                                                  \
    if (s_ == 0) {                                                                                                     \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a0, pok[k_] ? pixi[k_] * cb_ + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));                    \
    } else {                                                                                                           \
      _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_)                                                             \
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(                             \
            rsrc_a1, pok[k_] ? pixi[k_] * cb_ + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));                    \
    }                                                                                                                  \
  }
#define MSI_PATCH_STORE()                                                                                              \
  {                                                                                                                    \
    const bool ap_ = APPLY && ((p.halo_apply >> src_ld) & 1);                                                          \
    v4f s_[2], t_[2];                                                                                                  \
    if (ap_) {                                                                                                         \
      const float ih_ = src_ld ? inv1 : inv0, mh_ = src_ld ? mh1 : mh0, ml_ = src_ld ? ml1 : ml0;                       \
      const float uf_ = src_ld ? up1 : up0;                                                                            \
      const v4f nh = {-mh_, -mh_, -mh_, -mh_}, nl = {-ml_, -ml_, -ml_, -ml_};                                          \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                               \
        const v4f su_ = ih_ * g8[h_];                                                                                  \
        t_[h_] = __builtin_elementwise_fma(nl, su_, __builtin_elementwise_fma(nh, su_, be8[h_]));                      \
        s_[h_] = uf_ * su_;                                                                                            \
      }                                                                                                                \
    }                                                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < NLOAD; ++k_) {                                                             \
      v4f o_ = araw[k_];                                                                                               \
      if (ap_) {                                                                                                       \
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};                                                                           \
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));                                                     \
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);                                                           \
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};                                  \
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};                                  \
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);                          \
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);                          \
        if (has_pad && !pok[k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */                 \
        unsigned w0, w1, w2, w3;                                                                                       \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));                                         \
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));                                         \
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});                                                         \
      }                                                                                                                \
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;                                   \
    }                                                                                                                  \
  }
  // weights of k-step (class, tap, chunk c) -> ring stage st (run-time)
#define MSI_B_ISSUE(cls, tap, c, st)                                                                                   \
  {                                                                                                                    \
    char *sB_ = smem + G::A_BYTES + (st) * G::B_STAGE + wave * (BN / 4) * ROW_BYTES;                                   \
    const int soff_ = (((cls) * S + (tap) * CH + (c)) * p.npad) * ROW_BYTES;                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);            \
    if (BI == 4) {                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);         \
    }                                                                                                                  \
  }

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  // fragment base of tap row th = 0 (patch row 1 + local row) and of th = 1 (one row up for ph = 0, one down for ph = 1)
  const unsigned a_base0 = lds_base + (unsigned)((1 + wm * (MT * 2) + (frow >> 4)) * G::ROW_PITCH + (frow & 15) * G::PIX_BYTES + fh * 16);
  const unsigned a_base1 = ph ? a_base0 + G::ROW_PITCH : a_base0 - G::ROW_PITCH;
  unsigned b_q[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    b_q[q] = lds_base + G::A_BYTES + (wn * (NT * 32) + frow) * ROW_BYTES + (((2 * q + fh) ^ fswz) << 4);
  f32x16 acc[2][MT][NT];
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cl][i][j][r] = 0.f;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

  // k-step J of the chunk: class pw = J / 4, tap (th, tw) = ((J / 2) & 1, J & 1); weights in ring stage st
#define MSI_CQ(Q, PWC)                                                                                                 \
  wait_lgkm_frag<(3 - (Q)) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                  \
      acc[PWC][i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),               \
                                                                 __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[PWC][i_][j_], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
#define MSI_CTSTEP(J)                                                                                                  \
  {                                                                                                                    \
    constexpr int PWC_ = (J) >> 2, TH_ = ((J) >> 1) & 1, TW_ = (J) & 1;                                                \
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */                \
    constexpr int AROW_ = 2 * G::ROW_PITCH;                                                                            \
    const unsigned ab_ = TH_ ? a_base1 : a_base0;                                                                      \
    v4f fa_[4][MT], fb_[4][NT];                                                                                        \
    const unsigned bst_ = (unsigned)st * G::B_STAGE;                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                 \
      const unsigned ba_ = b_q[q_] + bst_;                                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                                                \
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<COFF_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 32>(ab_)            \
                                : q_ == 2 ? lds_read128<COFF_ + 64>(ab_) : lds_read128<COFF_ + 96>(ab_))               \
                    : i_ == 1 ? (q_ == 0 ? lds_read128<COFF_ + AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + AROW_ + 32>(ab_) \
                                : q_ == 2 ? lds_read128<COFF_ + AROW_ + 64>(ab_) : lds_read128<COFF_ + AROW_ + 96>(ab_)) \
                    : i_ == 2 ? (q_ == 0 ? lds_read128<COFF_ + 2 * AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 2 * AROW_ + 32>(ab_) \
                                : q_ == 2 ? lds_read128<COFF_ + 2 * AROW_ + 64>(ab_) : lds_read128<COFF_ + 2 * AROW_ + 96>(ab_)) \
                              : (q_ == 0 ? lds_read128<COFF_ + 3 * AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 3 * AROW_ + 32>(ab_) \
                                : q_ == 2 ? lds_read128<COFF_ + 3 * AROW_ + 64>(ab_) : lds_read128<COFF_ + 3 * AROW_ + 96>(ab_)); \
      _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)                                                                \
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);                                \
    }                                                                                                                  \
    MSI_CQ(0, PWC_)                                                                                                    \
    bool issued_;                                                                                                      \
    {                                                                                                                  \
      if ((J) == 0 && c + 1 < c1) MSI_PATCH_LOAD(c + 1)                                                                \
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;                                                         \
      constexpr int JN_ = ((J) + PD) & 7;                                                                              \
      issued_ = ((J) + PD < 8) || (c + 1 < c1);                                                                        \
      if ((J) + PD < 8) { MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_) }                                          \
      else if (c + 1 < c1) { MSI_B_ISSUE(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_) }                                   \
    }                                                                                                                  \
    MSI_CQ(1, PWC_) MSI_CQ(2, PWC_) MSI_CQ(3, PWC_)                                                                    \
    if ((J) == 0 && c + 1 < c1) wait_vmcnt<BI + NPL>();                                                                \
    else if (issued_) wait_vmcnt<BI>();                                                                                \
    else wait_vmcnt<0>();                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                      \
    st = st + 1 == NSTG ? 0 : st + 1;                                                                                  \
  }

  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  MSI_PATCH_LOAD(c0)
  MSI_B_ISSUE(2 * ph, 0, c0, 0)
  MSI_B_ISSUE(2 * ph, 1, c0, 1)
  if (APPLY && p.halo_apply) {
    double *s_stat = reinterpret_cast<double *>(smem);
    if (p.halo_apply & 1) {
      ln_mean_inv(p.ln_sums + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n, p.ln_scl_src, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv0 = (float)s_stat[1]; mh0 = (float)mu; ml0 = (float)(mu - (double)mh0);
      up0 = (float)(p.ln_scl_src[2] * 16777216.0);
      __syncthreads();
    }
    if (p.halo_apply & 2) {
      ln_mean_inv(p.ln_sums1 + (size_t)b * LN_SHARDS * LN_WORDS, p.ln_inv_n1, p.ln_scl_src1, p.status, s_stat, tid);
      const double mu = s_stat[0];
      inv1 = (float)s_stat[1]; mh1 = (float)mu; ml1 = (float)(mu - (double)mh1);
      up1 = (float)(p.ln_scl_src1[2] * 16777216.0);
      __syncthreads();
    }
  }
  wait_vmcnt<0>();
  MSI_PATCH_STORE()
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    MSI_CTSTEP(0) MSI_CTSTEP(1) MSI_CTSTEP(2) MSI_CTSTEP(3) MSI_CTSTEP(4) MSI_CTSTEP(5) MSI_CTSTEP(6) MSI_CTSTEP(7)
    if (c + 1 < c1) {
      MSI_PATCH_STORE()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#undef MSI_CTSTEP
#undef MSI_CQ
#undef MSI_B_ISSUE
#undef MSI_PATCH_STORE
#undef MSI_PATCH_LOAD
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int pwc = 0; pwc < 2; ++pwc) {   // (LDS is free: the k-loop ended with a barrier)
    emit_tile<BM, BN, MODE_CONVT, 1>(p, acc[pwc], tile_m, tile_n, 2 * ph + pwc, b, tid, smem);
    __builtin_amdgcn_sched_barrier(0);   // one class after the other: interleaved, the two epilogues do not fit the register file
  }
#ifdef MSI_CONV_TIMING
  if (p.dbg && tid == 0) {
    unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
    o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
#endif
#endif
}

// Fix-up of the split tiles as a separate launch (plan option MSI_NET_OPT_FIXUP_KERNEL; the default is the in-launch
// hand-off above): sums the K-range slabs of a tile in k order and runs the same epilogue.  One workgroup per split tile.
template <int BM, int BN, int MODE, int RAW16 = 0>
__global__ void __launch_bounds__(256)
conv_fixup_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MT = BM / 64, NT = BN / 64;
  const int tid = threadIdx.x;
  // split tiles: [0, ntiles) when the first group is split too, else [n_main, ntiles); their partial
  // slots are consecutive in workgroup order of conv_igemm_kernel
  const int t = (p.split0 == 1 ? p.n_main : 0) + blockIdx.x;
  const int nsp = t < p.n_main ? p.split0 : p.split;
  const int slot0 = p.split0 == 1 ? (t - p.n_main) * p.split
                                  : (t < p.n_main ? t * p.split0 : p.nb_main + (t - p.n_main) * p.split);
  int tile_m, tile_n, cls, b;
  {
    int r = t;   // same order as conv_igemm_kernel: class fastest, then M tiles
    const int q0 = (int)udiv_magic((unsigned)r, (unsigned)p.nclass, p.mg_nc);
    cls = r - q0 * p.nclass; r = q0;
    const int q1 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_m, p.mg_tm);
    tile_m = r - q1 * p.tiles_m; r = q1;
    const int q2 = (int)udiv_magic((unsigned)r, (unsigned)p.tiles_n, p.mg_tn);
    tile_n = r - q2 * p.tiles_n;
    b = q2;
  }
  constexpr int SLAB = BM * BN * 4;
  f32x16 acc[MT][NT];
  if (MODE == MODE_CONVT && p.halo_tx) {   // convt_halo_kernel: the tile index carries ph, two class slabs (pw = blockIdx.y) per K-range
    const int pwc = blockIdx.y;
    cls = 2 * cls + pwc;                   // (nclass = 2 in this enumeration: `cls` decoded above is ph)
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.partial + ((size_t)slot0 * 2 + pwc) * (BM * BN)), 0, nsp * 2 * SLAB, 0x00020000);
    sum_slabs<MT, NT, 0>(acc, rsrc_h, nsp, 2 * SLAB, tid);
  } else {
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.partial + (size_t)slot0 * (BM * BN)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, 0>(acc, rsrc_t, nsp, SLAB, tid);
  }
  emit_tile<BM, BN, MODE, RAW16>(p, acc, tile_m, tile_n, cls, b, tid);
#endif
}

// ---- fused tail: 1x1 head (+ the producer's LayerNorm) + RGBA layer assembly ------------------------------------
// color_pred (nets.py:509-515) followed by infer_msi's layer_prediction for blend_psv (msi.py:130-147) in ONE kernel:
// `pred` (52 MB at the BASELINE size) is never written to or re-read from HBM and one launch disappears.  A workgroup
// owns 32 consecutive pixels: the sweep-volume tile (32 x 6D floats, contiguous) is requested first and stays in
// flight while the 32 x C0 activations (conv8_2 raw, normalised + ReLU'd on the way like the stand-alone head does)
// and the 2D x C0 weights go to LDS and 2 D / 32 waves run the k-steps on the fp32 MFMA -- the SAME instruction
// sequence as the stand-alone head (k ascending, transposed accumulators), so the prediction is bit-identical --
// then bias + tanh + (x+1)/2 land in an LDS tile and the assembly of K3 (geometry.hip, same expressions, no
// contraction) writes float4 texels of the D-major stack.  HBM-bound: reads C0 + 6D floats, writes 4D per pixel.
constexpr int HA_TP = 32;   // pixels per workgroup
constexpr int HA_LG = 32;   // at most this many layers per workgroup: D = 64 runs as two layer groups (grid.y)

struct HeadAsmParams {
  const float *x;            // conv8_2 raw [B,H,W,C0]
  const float *wpk;          // packed head weights [ksteps][npad][32 floats] (slots swizzled by output row)
  const float *bias;
  const float *aff;          // affine of the source layer's LayerNorm [B][scale[C0] | shift[C0]] (ln_finish_kernel)
  const void *psv;           // [B,H,W,6D] fp32, or bf16 (BF16IN)
  float4 *rgba;              // [B,D,H,W] float4
  float *bw_out, *al_out;    // optional [B,H,W,D]
  float *pred_out;           // optional [B,H,W,2D] (tanh output)
  int C0, ksteps, npad, nd, hw;
  int lg, ng;                // layers per workgroup (a multiple of 4, <= HA_LG) and layer groups: D = lg * ng
  unsigned mg_vpp, mg_nchunk, mg_hw;   // udiv_magic multipliers of the 16-byte vectors per pixel of the sweep-volume tile, of the
                             // 16-byte chunks per pixel of the activation tile, of H * W (run-time integer divisions are ~25 VALU each)
  long npix_total;
};

// A workgroup owns 32 pixels x lg layers (layer group g = blockIdx.y: blend weights g lg .. + lg, the alphas behind
// them, and the foreground / background colours of those layers: two runs of 3 lg channels of the sweep volume).
// Locally everything is a D = lg problem: output column n < lg is blend weight g lg + n, column lg + n its alpha.
// D <= 32: one group (the whole row is one run).  D = 64: two groups; the 32 x C0 activations are read by both
// (8 KB of 65 KB per workgroup), each keeps the 34 KB LDS footprint = four workgroups per CU (one 64-layer workgroup
// needed 66 KB: two per CU, 1.6 TB/s).
// BF16IN (bf16 plans): the sweep volume is bf16 (widened exactly on the way into LDS) and the normalised activation is
// rounded to bf16 (round to nearest even, where ln_apply_kernel<1> rounds it) before it enters the fp32 MFMA with the
// bf16-rounded weights: the operands of the bf16 head, exact products, fp32 accumulation.
template <int BF16IN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BF16IN ? 6 : 5)))
head_assemble_kernel(const HeadAsmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BN = 64;                                        // >= 2 lg local output columns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2): id -> XCD id & 7, slot id >> 3.  The ng
  // layer groups of a pixel tile take CONSECUTIVE slots of ONE XCD, so what they share -- the 32 x C0 activations and the
  // 128-byte lines their 192-byte colour runs straddle (1.43x the algorithmic read bytes when group 1 ran a whole grid
  // later, r03_g_hbm_traffic_config2) -- is fetched from HBM once and hit in that XCD's L2 the second time.
  const int nd = p.nd, lg = p.lg, ng = p.ng;
  const unsigned slot = blockIdx.x >> 3;
  const unsigned tsl = ng == 1 ? slot : (ng == 2 ? slot >> 1 : slot / (unsigned)ng);   // (D <= 64: one or two groups)
  const int g = (int)(slot - tsl * (unsigned)ng);
  const long tile = (long)tsl * 8 + (blockIdx.x & 7u);
  if (tile * HA_TP >= p.npix_total) return;                    // (grid rounded up to 8 ng workgroups)
  const int c_psv = 6 * nd, c_pred = 2 * nd;                    // global row lengths
  const int l_cpsv = 6 * lg, l_cpred = 2 * lg;                  // local ones
  const int s_psv = l_cpsv + 1, s_pred = l_cpred + 1;           // odd row strides (see assemble_kernel)
  // BF16IN (r03): the kernel was sensitive to its occupancy (three workgroups per CU instead of four: +14 %), and its LDS footprint
  // was the head's WEIGHTS (16 KB) next to the activations, and the bf16 sweep tile widened to fp32 (24.7 KB).  A bf16 plan therefore
  // (a) fetches the two active waves' weight fragments straight from the packed blob into registers (L2-resident, 8 x 16 bytes per
  // lane) -- no B tile in LDS -- and (b) keeps the sweep tile as packed bf16 with a row stride of 3 lg + 1 dwords (odd: conflict-free
  // for the 32 pixels of a half-wave), widened on the way out: 21 KB per workgroup, six to seven workgroups per CU.
  const int s_psv16 = 3 * lg + 1;                               // dwords per pixel row of the packed-bf16 sweep tile
  // LDS: [affine 2 C0 | stat | R | pred tile]; R holds A (ksteps x 32 rows) | B (ksteps x BN rows) during the GEMM and
  // the sweep-volume tile afterwards (33.6 KB per workgroup at lg = 32: four workgroups per CU, like assemble_kernel)
  float *s_aff = reinterpret_cast<float *>(smem);
  char *sR = smem + 2 * 64 * 4 + 64;
  char *sA = sR;
  char *sB = sA + p.ksteps * HA_TP * ROW_BYTES;
  const size_t r_bytes = BF16IN ? max((size_t)p.ksteps * HA_TP * ROW_BYTES, (size_t)HA_TP * s_psv16 * sizeof(unsigned))
                                : max((size_t)p.ksteps * (HA_TP + BN) * ROW_BYTES, (size_t)HA_TP * s_psv * sizeof(float));
  float *l_psv = reinterpret_cast<float *>(sR);
  float *l_pred = reinterpret_cast<float *>(sR + ((r_bytes + 15) & ~(size_t)15));
  // local output column -> global one (in float4 groups: lg % 4 == 0)
  auto gcol = [&](int n) __attribute__((always_inline)) -> int { return n < lg ? g * lg + n : nd + g * lg + (n - lg); };

  const long p0 = tile * HA_TP;
  const int b = (int)udiv_magic((unsigned)p0, (unsigned)p.hw, p.mg_hw);   // (H * W is a multiple of 32: a tile never straddles samples; B * H * W < 2^32, host-checked)
  // 1. every global load of the workgroup goes out first and is parked in registers: the sweep-volume tile, the raw
  //    activations (C0 <= 64: at most two float4 per thread), the weight rows -- ONE memory round trip per workgroup
  constexpr int PSV_PER_THREAD = BF16IN ? 3 : 6;                // 32 x 6 lg elements in 16-byte vectors / 256 threads, lg <= 32
  constexpr int B_PER_THREAD = 4;                               // ksteps (<= 2) x BN rows x 8 float4 / 256
  constexpr int PSV_VEC = BF16IN ? 8 : 4;                       // elements per 16-byte vector
  constexpr int ESZ = BF16IN ? 2 : 4;
  // runs of the global row this group needs: the whole row (one group), or its foreground and background colours
  const int nrun = ng == 1 ? 1 : 2;
  const int run_len = ng == 1 ? c_psv : 3 * lg;          // elements; a multiple of PSV_VEC (host-checked)
  const int vpr = run_len / PSV_VEC, vpp = nrun * vpr;          // vectors per run / per pixel
  const int nv_psv = HA_TP * vpp;
  float4 q[PSV_PER_THREAD];
  {
    const char *gp = static_cast<const char *>(p.psv) + (size_t)p0 * c_psv * ESZ;
#pragma unroll
    for (int k = 0; k < PSV_PER_THREAD; ++k) {
      const int v = tid + 256 * k;
      if (v < nv_psv) {
        if (ng == 1) {   // the whole tile is contiguous: no index arithmetic in front of the loads
          q[k] = reinterpret_cast<const float4 *>(gp)[v];
        } else {
          const int px = (int)udiv_magic((unsigned)v, (unsigned)vpp, p.mg_vpp), w = v - px * vpp;
          const int r = w >= vpr ? 1 : 0, idx = w - r * vpr;
          const int start = r == 0 ? 3 * g * lg : 3 * (nd + g * lg);
          q[k] = *reinterpret_cast<const float4 *>(gp + ((size_t)px * c_psv + start + idx * PSV_VEC) * ESZ);
        }
      }
    }
  }
  const int nchunk = p.ksteps * 8;                              // 16-byte chunks per pixel (zero beyond C0)
  v4f araw[2];
  if (BF16IN) {
    // a bf16 plan keeps conv8_2's raw output as fp16 of x * 2^-e (the affine of ln_finish_kernel carries 2^e): thread t loads
    // the 16 bytes that hold its two chunks e = 2 t, 2 t + 1 (eight channels) -- one 16-byte load per thread, as in the fp32 form
    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
    const int r = tid >> 3, c = (tid & 7) * 8;   // 32 pixels x 8 slots of 8 channels (C0 <= 64)
    araw[0] = araw[1] = v4f{0.f, 0.f, 0.f, 0.f};
    if (c < p.C0) {   // (C0 % 8 == 0 in a bf16 plan)
      const h8_t h = *reinterpret_cast<const h8_t *>(reinterpret_cast<const _Float16 *>(p.x) + (size_t)(p0 + r) * p.C0 + c);
      araw[0] = v4f{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      araw[1] = v4f{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k;
      const int r = (int)udiv_magic((unsigned)e, (unsigned)nchunk, p.mg_nchunk), c = (e - r * nchunk) * 4;
      araw[k] = v4f{0.f, 0.f, 0.f, 0.f};
      if (e < HA_TP * nchunk && c < p.C0) araw[k] = *reinterpret_cast<const v4f *>(p.x + (size_t)(p0 + r) * p.C0 + c);
    }
  }
  const int nb = p.ksteps * BN * 8;
  v4f braw[B_PER_THREAD];
  v4f wfrag[4];                                                 // BF16IN, waves 0 / 1: the weight fragments of the lane's output column
  if (BF16IN) {
    // round 4: the head of a bf16 plan runs on v_mfma_f32_32x32x16_bf16 over the packed bf16 rows of color_pred themselves
    // (64 channels per 128-byte row: one k-step for C0 <= 64; MFMA q takes chunk 2 q + half, as in the conv kernels) --
    // the same operands as before (the fp32 MFMA ran on fp32-format copies of these bf16 values), exact products, fp32
    // accumulation, another summation order; 4 MFMAs of 8 passes instead of 32 of 16 per wave and tile
    if (wave < 2) {
      const int frow = lane & 31, fh = lane >> 5;
      const int nloc = wave * 32 + frow;
      const int gn = gcol(nloc < l_cpred ? nloc : 0);
      const int fswz = (gn >> 1) & 7;                           // (a packed row's slots are swizzled by its GLOBAL row)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        wfrag[qq] = v4f{0.f, 0.f, 0.f, 0.f};
        if (nloc < l_cpred)
          wfrag[qq] = *reinterpret_cast<const v4f *>(p.wpk + (size_t)gn * (ROW_BYTES / 4) + (((2 * qq + fh) ^ fswz) << 2));
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < B_PER_THREAD; ++k) {
      const int e = tid + 256 * k;
      const int row = e >> 3, j = e & 7;
      const int ks = row / BN, n = row - ks * BN;
      braw[k] = v4f{0.f, 0.f, 0.f, 0.f};                          // local rows >= 2 lg: zero
      if (e < nb && n < l_cpred)   // (a packed row keeps the slot swizzle of its GLOBAL row (gcol(n) >> 1) & 7: see fswz_b below)
        braw[k] = *reinterpret_cast<const v4f *>(p.wpk + ((size_t)ks * p.npad + gcol(n)) * (ROW_BYTES / 4) + j * 4);
    }
  }
  // 2. affine of the source's LayerNorm (precomputed once per forward by ln_finish_kernel: 6 400 workgroups deriving it
  //    from the sums themselves put two more dependent round trips on every workgroup's critical path)
  if (tid < 2 * p.C0) s_aff[tid] = p.aff[(size_t)b * 2 * p.C0 + tid];
  __syncthreads();
  // 3. A: 32 pixels x (ksteps * 32) channels, LayerNorm + ReLU applied (the stand-alone head's expression), zero beyond
  //    C0; the 16-byte slot s of row r holds data chunk s ^ ((r >> 1) & 7) (the conv kernel's LDS image).  B: the group's
  //    rows of every k-step of the packed blob as they are (pre-swizzled by their GLOBAL row)
  if (BF16IN) {
    // thread t holds channels 8 (t & 7) .. + 7 of pixel t >> 3: LayerNorm + ReLU, two values per v_cvt_pk_bf16_f32 (round to
    // nearest even, where ln_apply_kernel<1> rounds), one 16-byte slot of the pixel's 128-byte row (64 channels)
    const int r = tid >> 3, ch = tid & 7, c = ch * 8;
    unsigned pk[4] = {0u, 0u, 0u, 0u};
    if (c < p.C0) {                                             // C0 % 8 == 0 in a bf16 plan
      float y[8];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const v4f s4 = *reinterpret_cast<const v4f *>(s_aff + c + 4 * k), t4 = *reinterpret_cast<const v4f *>(s_aff + p.C0 + c + 4 * k);
        y[4 * k] = fmaxf(araw[k].x * s4.x + t4.x, 0.f); y[4 * k + 1] = fmaxf(araw[k].y * s4.y + t4.y, 0.f);
        y[4 * k + 2] = fmaxf(araw[k].z * s4.z + t4.z, 0.f); y[4 * k + 3] = fmaxf(araw[k].w * s4.w + t4.w, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[k]) : "v"(y[2 * k]), "v"(y[2 * k + 1]));
    }
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    *reinterpret_cast<v4u_t *>(sA + r * ROW_BYTES + ((ch ^ ((r >> 1) & 7)) << 4)) = v4u_t{pk[0], pk[1], pk[2], pk[3]};
  } else
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = tid + 256 * k;                          // (the chunk araw[k] holds: see the loads above)
    if (e < HA_TP * nchunk) {
      const int r = (int)udiv_magic((unsigned)e, (unsigned)nchunk, p.mg_nchunk), ch = e - r * nchunk;
      const int c = ch * 4;
      v4f y = {0.f, 0.f, 0.f, 0.f};
      if (c < p.C0) {                                           // C0 % 4 == 0
        const v4f s4 = *reinterpret_cast<const v4f *>(s_aff + c), t4 = *reinterpret_cast<const v4f *>(s_aff + p.C0 + c);
        y.x = fmaxf(araw[k].x * s4.x + t4.x, 0.f); y.y = fmaxf(araw[k].y * s4.y + t4.y, 0.f);
        y.z = fmaxf(araw[k].z * s4.z + t4.z, 0.f); y.w = fmaxf(araw[k].w * s4.w + t4.w, 0.f);
      }
      const int ks = ch >> 3, chunk = ch & 7;
      *reinterpret_cast<v4f *>(sA + (ks * HA_TP + r) * ROW_BYTES + ((chunk ^ ((r >> 1) & 7)) << 4)) = y;
    }
  }
  if (!BF16IN) {
#pragma unroll
    for (int k = 0; k < B_PER_THREAD; ++k) {
      const int e = tid + 256 * k;
      if (e < nb) *reinterpret_cast<v4f *>(sB + (e >> 3) * ROW_BYTES + ((e & 7) << 4)) = braw[k];
    }
  }
  __syncthreads();
  // 4. the GEMM: wave w owns local output columns [32 w, 32 w + 32) of the 32 pixels; 5. bias + tanh (-> optional pred),
  //    (x + 1) / 2 (msi.py:132-133) -> LDS tile (+ the optional [B,H,W,D] outputs)
  if (wave < 2) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int frow = lane & 31, fh = lane >> 5;
    const int fswz_a = (frow >> 1) & 7;
    const int fswz_b = (gcol(wave * 32 + frow < l_cpred ? wave * 32 + frow : 0) >> 1) & 7;   // B rows keep the swizzle of their global row
    if (BF16IN) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const v4f a = *reinterpret_cast<const v4f *>(sA + frow * ROW_BYTES + (((2 * qq + fh) ^ fswz_a) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag[qq]), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
      }
    } else
    for (int ks = 0; ks < p.ksteps; ++ks) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const v4f a = *reinterpret_cast<const v4f *>(sA + (ks * HA_TP + frow) * ROW_BYTES + (((fh * 4 + qq) ^ fswz_a) << 4));
        const v4f w = *reinterpret_cast<const v4f *>(sB + (ks * BN + wave * 32 + frow) * ROW_BYTES + (((fh * 4 + qq) ^ fswz_b) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a.w, acc, 0, 0, 0);
      }
    }
    const int px = lane & 31, half = lane >> 5;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int n = wave * 32 + 8 * gg + 4 * half;             // local column
      if (n < l_cpred) {                                        // 2 lg is a multiple of 8: whole float4 in range
        const int gn = gcol(n);
        const v4f bs = *reinterpret_cast<const v4f *>(p.bias + gn);
        v4f t = {msi_tanh(acc[4 * gg] + bs.x), msi_tanh(acc[4 * gg + 1] + bs.y), msi_tanh(acc[4 * gg + 2] + bs.z), msi_tanh(acc[4 * gg + 3] + bs.w)};
        if (p.pred_out) *reinterpret_cast<v4f *>(p.pred_out + (p0 + px) * c_pred + gn) = t;
        t.x = (t.x + 1.0f) / 2.0f; t.y = (t.y + 1.0f) / 2.0f; t.z = (t.z + 1.0f) / 2.0f; t.w = (t.w + 1.0f) / 2.0f;
        float *dst = l_pred + px * s_pred + n;
        dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
        if (n < lg) {
          if (p.bw_out) *reinterpret_cast<v4f *>(p.bw_out + (p0 + px) * nd + gn) = t;
        } else {
          if (p.al_out) *reinterpret_cast<v4f *>(p.al_out + (p0 + px) * nd + (gn - nd)) = t;
        }
      }
    }
  }
  __syncthreads();                                              // A | B have been read: the sweep-volume tile replaces them
  // 6. the parked sweep-volume tile -> LDS: local row = [foreground run | background run], padded to an odd stride
#pragma unroll
  for (int k = 0; k < PSV_PER_THREAD; ++k) {
    const int v = tid + 256 * k;
    if (v < nv_psv) {
      const int px = (int)udiv_magic((unsigned)v, (unsigned)vpp, p.mg_vpp), w = v - px * vpp;   // (run r of the pixel starts at local column r * run_len)
      if (BF16IN) {   // eight bf16 = four dwords, as they are (element e of the local row = half e & 1 of dword e >> 1)
        unsigned *dst = reinterpret_cast<unsigned *>(l_psv) + px * s_psv16 + w * 4;
        dst[0] = __builtin_bit_cast(unsigned, q[k].x); dst[1] = __builtin_bit_cast(unsigned, q[k].y);
        dst[2] = __builtin_bit_cast(unsigned, q[k].z); dst[3] = __builtin_bit_cast(unsigned, q[k].w);
      } else {
        float *dst = l_psv + px * s_psv + w * PSV_VEC;
        dst[0] = q[k].x; dst[1] = q[k].y; dst[2] = q[k].z; dst[3] = q[k].w;
      }
    }
  }
  __syncthreads();
  // 7. assembly (assemble_kernel, COLOR_BLEND_PSV; no contraction, like geometry.hip): thread -> (pixel, every 8th layer)
  {
#pragma clang fp contract(off)
    const int px = tid & (HA_TP - 1);
    const long pp = p0 + px;
    const long off = pp - (long)b * p.hw;
    const float *rp = l_psv + px * s_psv;
    const float *rq = l_pred + px * s_pred;
    float4 *dst = p.rgba + ((long)b * nd + g * lg + tid / HA_TP) * p.hw + off;   // (one 64-bit multiply per thread, not per layer)
    const long dstep = (long)(256 / HA_TP) * p.hw;
    const unsigned *rp16 = reinterpret_cast<const unsigned *>(l_psv) + px * s_psv16;
    for (int d = tid / HA_TP; d < lg; d += 256 / HA_TP, dst += dstep) {
      float fgv[3], bgv[3];
      if (BF16IN) {   // elements 3 d .. 3 d + 2 and 3 (lg + d) .. + 2 of the packed-bf16 row: two dwords each, widened exactly
        const int ef = 3 * d, eb = 3 * (lg + d);
        const unsigned f0 = rp16[ef >> 1], f1 = rp16[(ef >> 1) + 1], b0 = rp16[eb >> 1], b1 = rp16[(eb >> 1) + 1];
        const unsigned long long fw = ((unsigned long long)f1 << 32 | f0) >> ((ef & 1) * 16), bw = ((unsigned long long)b1 << 32 | b0) >> ((eb & 1) * 16);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          fgv[c] = __builtin_bit_cast(float, (unsigned)(fw >> (16 * c)) << 16);
          bgv[c] = __builtin_bit_cast(float, (unsigned)(bw >> (16 * c)) << 16);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { fgv[c] = rp[d * 3 + c]; bgv[c] = rp[(lg + d) * 3 + c]; }
      }
      const float *fg = fgv;
      const float *bg = bgv;
      const float w = rq[d];
      const float omw = 1.0f - w;
      float4 o;
      o.x = w * fg[0] + omw * bg[0];
      o.y = w * fg[1] + omw * bg[1];
      o.z = w * fg[2] + omw * bg[2];
      o.w = rq[lg + d];
      *dst = o;
    }
  }
#endif
}

// The affine of one layer's LayerNorm, scale | shift per channel, for consumers that apply it themselves while loading
// (head_assemble_kernel): one workgroup per sample.
__global__ void __launch_bounds__(256)
ln_finish_kernel(const long long *__restrict__ sums, double inv_n, const double *__restrict__ scl, int *status,
                 const float *__restrict__ gamma, const float *__restrict__ beta, int C, float *__restrict__ aff, int raw16) {
  __shared__ double s_stat[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  ln_mean_inv(sums + (size_t)b * LN_SHARDS * LN_WORDS, inv_n, scl, status, s_stat, tid);
  const double mu = s_stat[0], inv = s_stat[1];
  const double up = raw16 ? scl[2] * 16777216.0 : 1.0;   // 2^e: the consumer reads the raw output as fp16 of x * 2^-e
  for (int c = tid; c < C; c += 256) {
    const double sc = inv * (double)gamma[c];
    aff[(size_t)b * 2 * C + c] = (float)(sc * up);
    aff[(size_t)b * 2 * C + C + c] = (float)((double)beta[c] - mu * sc);
  }
}

// LayerNorm apply (+ ReLU), one launch per layer.  Every workgroup derives the affine of slim.layer_norm,
//   scale = gamma * rsqrt(var + eps), shift = beta - mean * scale,
// from the sample's 64 x 2 fixed-point sums (ln_mean_inv), keeps it in LDS, and applies
// x = max(x*scale[c] + shift[c], 0) to its grid-stride slice (nets.py:401,485 arg_scope: normalizer, then the
// default ReLU).  Workgroup 0 also publishes the affine (tests).
// BF16OUT = 1: the normalised activation is written as bf16 to `yb` (the operand buffer of the bf16 path)
// and the fp32 raw output is left alone; 0: x is normalised in place.
// tickets and LayerNorm sums of one forward start from zero
__global__ void __launch_bounds__(256) zero_kernel(float4 *__restrict__ p, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) p[i] = float4{0.f, 0.f, 0.f, 0.f};
}

template <int BF16OUT>
__global__ void __launch_bounds__(256)
ln_apply_kernel(float *__restrict__ x, const long long *__restrict__ sums, double inv_n, const double *__restrict__ scl,
                int *status, const float *__restrict__ gamma, const float *__restrict__ beta, size_t per_sample, int C,
                float *__restrict__ aff, unsigned short *__restrict__ yb) {
  extern __shared__ __attribute__((aligned(16))) float s_aff[];  // scale[C] shift[C]
  __shared__ double s_stat[2];
  const int b = blockIdx.y, tid = threadIdx.x;
  ln_mean_inv(sums + (size_t)b * LN_SHARDS * LN_WORDS, inv_n, scl, (blockIdx.x == 0 ? status : nullptr), s_stat, tid);
  const double mu = s_stat[0], inv = s_stat[1];
  // BF16OUT: the raw output is fp16 of x * 2^-e (emit_tile_impl RAW16): the scale applied to it carries 2^e (exact)
  const double up = BF16OUT ? scl[2] * 16777216.0 : 1.0;
  for (int c = tid; c < C; c += 256) {
    const double sc = inv * (double)gamma[c];
    const float fs = (float)sc, ft = (float)((double)beta[c] - mu * sc);
    s_aff[c] = BF16OUT ? (float)(sc * up) : fs;
    s_aff[C + c] = ft;
    if (blockIdx.x == 0) {      // (published for the tests: the affine of the UNSCALED raw output)
      aff[(size_t)b * 2 * C + c] = fs;
      aff[(size_t)b * 2 * C + C + c] = ft;
    }
  }
  __syncthreads();

  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  const h4_t *xh = reinterpret_cast<const h4_t *>(reinterpret_cast<const _Float16 *>(x) + (size_t)b * per_sample);
  v4f *xv = reinterpret_cast<v4f *>(x + (size_t)b * per_sample);
  auto get = [&](size_t i) __attribute__((always_inline)) -> v4f {
    if (BF16OUT == 1) { const h4_t h = xh[i]; return v4f{(float)h.x, (float)h.y, (float)h.z, (float)h.w}; }
    return xv[i];
  };
  const size_t nvec = per_sample / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto apply = [](v4f v, const v4f s4, const v4f t4) __attribute__((always_inline)) -> v4f {
    v.x = fmaxf(v.x * s4.x + t4.x, 0.f);
    v.y = fmaxf(v.y * s4.y + t4.y, 0.f);
    v.z = fmaxf(v.z * s4.z + t4.z, 0.f);
    v.w = fmaxf(v.w * s4.w + t4.w, 0.f);
    return v;
  };
  // fp32 -> bf16, round to nearest even (finite inputs: the LayerNorm output)
  auto bf16_bits = [](float f) __attribute__((always_inline)) -> unsigned {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  v2u *yv = reinterpret_cast<v2u *>(yb + (size_t)b * per_sample);
  auto put = [&](size_t i, const v4f v) __attribute__((always_inline)) {
    if (BF16OUT == 1) yv[i] = v2u{bf16_bits(v.x) | (bf16_bits(v.y) << 16), bf16_bits(v.z) | (bf16_bits(v.w) << 16)};
    else xv[i] = v;
  };
  if ((256 * 4) % C == 0) {
    // every grid-stride step advances a thread by a multiple of C floats: its four channels are fixed
    const int c = (tid * 4) % C;
    const v4f s4 = *reinterpret_cast<const v4f *>(s_aff + c);
    const v4f t4 = *reinterpret_cast<const v4f *>(s_aff + C + c);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < nvec; i += stride) put(i, apply(get(i), s4, t4));
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < nvec; i += stride) {
      const int c = (int)((i * 4) % C);
      put(i, apply(get(i), *reinterpret_cast<const v4f *>(s_aff + c), *reinterpret_cast<const v4f *>(s_aff + C + c)));
    }
  }
}

// ============================================================================================
// host: layer table, parameter packing, plan, forward
// ============================================================================================
struct Layer {
  char name[16];
  int kind;  // MODE_*
  int cin, cout, has_coord, stride, rate;
  int in_h, in_w, out_h, out_w;
  int src0, src1;  // producer layer indices (-1 = net_input; src1 = -1: none)
  int c0, c1;
  int ntaps, cpt0, cpt1, ksteps, nclass, npad;
  int wrapt;       // conv-transpose of msi_train_net: GEMM rows cover the uncropped VALID output (see tap_delta)
  int mh, mw;      // GEMM row grid per sample and class
  double ln_count; // elements per sample the LayerNorm statistics run over
  size_t param_off, param_floats;  // floats
  size_t packed_off;               // floats: weights, then gamma, beta (or bias), then the CoordNet bias table
  size_t packed_w_floats;
  size_t gamma_off, beta_off, coord_off;  // floats inside the packed blob
  size_t lnscl_off;                       // floats inside the packed blob: LN_SCL_DOUBLES doubles (8-byte aligned)
  size_t x3_off;                          // floats inside the packed blob: the 3-way bf16 split of the weights (conv_halo_x3_kernel), 0 = none
  size_t x2_off;                          // ... the 2-way fp16 split (h, m' = (w - h) 2^11: plan option F32_SPLIT_F16), 0 = none
  size_t raw_off, aff_off;                // bytes inside the workspace
  size_t act_off;                         // bf16 path: normalised bf16 activation (the next layer's operand)
  size_t sums_off;                        // LayerNorm sums [B][LN_SHARDS][LN_WORDS] int64
  size_t flags_off;                       // apply-ahead row counters of THIS layer's output [B][out_h] ints
};

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Net {
  std::vector<Layer> layers;
  size_t param_floats = 0, packed_floats = 0, ws_bytes = 0, partial_off = 0, partial_bytes = 0;
  size_t zero_off = 0, zero_bytes = 0;   // [tickets of the in-launch fix-up | LayerNorm sums]: one memset per forward
  size_t cnt_off = 0;   // arrival tickets: [layer][5 * num_cus] ints
  size_t err_off = 0;   // one int: a tile workgroup gave up waiting for apply-ahead rows (stays 0)
  // bf16 plans: the head's weights (rounded to bf16) once more as fp32 rows, for the fused tail (head_assemble_kernel
  // runs the 1x1 head on the fp32 MFMA: exact products of bf16 values, fp32 accumulate -- the bf16 head's arithmetic)
  size_t head_f32_off = 0;
  int head_f32_ksteps = 0, head_f32_npad = 0;
};

int build_net(const msi_net_desc *d, int num_cus, Net &net) {
  if (!d) return msi::fail(MSI_E_BADARG, "net: null descriptor");
  if (d->batch < 0 || d->height <= 0 || d->width <= 0 || d->in_channels <= 0 || d->num_outputs <= 0 ||
      d->ngf <= 0)
    return msi::fail(MSI_E_BADARG, "net: bad descriptor");
  if (d->height % 8 || d->width % 8)
    return msi::fail(MSI_E_UNSUPPORTED, "net: height and width must be multiples of 8 (got %dx%d)",
                     d->height, d->width);
  if (d->in_channels % 4 || d->ngf % 4)
    return msi::fail(MSI_E_UNSUPPORTED, "net: in_channels and ngf must be multiples of 4");
  if (d->dtype != MSI_DTYPE_F32 && d->dtype != MSI_DTYPE_BF16)
    return msi::fail(MSI_E_BADARG, "net: dtype %d (MSI_DTYPE_F32 or MSI_DTYPE_BF16)", d->dtype);
  const int bf16 = d->dtype == MSI_DTYPE_BF16;
  const int esz = bf16 ? 2 : 4, bke = ROW_BYTES / esz;   // operand bytes, channels per k-step
  if (bf16 && (d->in_channels % 8 || d->ngf % 8))
    return msi::fail(MSI_E_UNSUPPORTED, "net: bf16 needs in_channels and ngf in multiples of 8 (16-byte channel chunks)");
  if ((long)(d->height + 16) * (d->width + 16) >= (1L << 24))
    return msi::fail(MSI_E_UNSUPPORTED, "net: more than 2^24 pixels per sample (24-bit pixel index in the conv kernel)");
  const int ngf = d->ngf, ex = d->coord_net ? 1 : 0;
  struct Spec { const char *name; int kind, src0, src1, cout, stride, rate; };
  const Spec specs[MSI_NET_NUM_LAYERS] = {
      {"conv1_1", MODE_CONV, -1, -1, ngf, 1, 1},      {"conv1_2", MODE_CONV, 0, -1, ngf * 2, 2, 1},
      {"conv2_1", MODE_CONV, 1, -1, ngf * 2, 1, 1},   {"conv2_2", MODE_CONV, 2, -1, ngf * 4, 2, 1},
      {"conv3_1", MODE_CONV, 3, -1, ngf * 4, 1, 1},   {"conv3_2", MODE_CONV, 4, -1, ngf * 4, 1, 1},
      {"conv3_3", MODE_CONV, 5, -1, ngf * 8, 2, 1},   {"conv4_1", MODE_CONV, 6, -1, ngf * 8, 1, 2},
      {"conv4_2", MODE_CONV, 7, -1, ngf * 8, 1, 2},   {"conv4_3", MODE_CONV, 8, -1, ngf * 8, 1, 2},
      {"conv6_1", MODE_CONVT, 9, 6, ngf * 4, 2, 1},   {"conv6_2", MODE_CONV, 10, -1, ngf * 4, 1, 1},
      {"conv6_3", MODE_CONV, 11, -1, ngf * 4, 1, 1},  {"conv7_1", MODE_CONVT, 12, 3, ngf * 2, 2, 1},
      {"conv7_2", MODE_CONV, 13, -1, ngf * 2, 1, 1},  {"conv8_1", MODE_CONVT, 14, 1, ngf, 2, 1},
      {"conv8_2", MODE_CONV, 15, -1, ngf, 1, 1},      {"color_pred", MODE_HEAD, 16, -1, d->num_outputs, 1, 1},
  };
  net.layers.resize(MSI_NET_NUM_LAYERS);
  size_t poff = 0, koff = 0, woff = 0;
  for (int i = 0; i < MSI_NET_NUM_LAYERS; ++i) {
    Layer &L = net.layers[i];
    const Spec &s = specs[i];
    memset(&L, 0, sizeof(L));
    strncpy(L.name, s.name, sizeof(L.name) - 1);
    L.kind = s.kind;
    L.src0 = s.src0;
    L.src1 = s.src1;
    L.cout = s.cout;
    L.stride = s.stride;
    L.rate = s.rate;
    const int sh = s.src0 < 0 ? d->height : net.layers[s.src0].out_h;
    const int sw = s.src0 < 0 ? d->width : net.layers[s.src0].out_w;
    L.in_h = sh;
    L.in_w = sw;
    L.c0 = s.src0 < 0 ? d->in_channels : net.layers[s.src0].cout;
    L.c1 = s.src1 < 0 ? 0 : net.layers[s.src1].cout;
    if (s.src1 >= 0 && (net.layers[s.src1].out_h != sh || net.layers[s.src1].out_w != sw))
      return msi::fail(MSI_E_BADARG, "net: skip shapes disagree at %s", s.name);
    L.cin = L.c0 + L.c1;
    L.has_coord = (s.kind == MODE_CONV) ? ex : 0;
    if (s.kind == MODE_CONV) {
      L.out_h = (sh + s.stride - 1) / s.stride;
      L.out_w = (sw + s.stride - 1) / s.stride;
      L.ntaps = 9;
      L.nclass = 1;
      L.mh = L.out_h; L.mw = L.out_w;
      L.ln_count = (double)L.out_h * L.out_w * L.cout;
    } else if (s.kind == MODE_CONVT) {
      L.out_h = sh * 2;
      L.out_w = sw * 2;
      L.ntaps = 4;
      L.nclass = 4;
      L.wrapt = d->coord_net ? 0 : 1;
      if (L.wrapt) {   // nets.py:423-435: LayerNorm over the uncropped (2H+10) x (2W+10) VALID output
        L.mh = sh + 1; L.mw = sw + 5;
        L.ln_count = (double)(2 * sh + 10) * (2 * sw + 10) * L.cout;
      } else {
        L.mh = sh; L.mw = sw;
        L.ln_count = (double)L.out_h * L.out_w * L.cout;
      }
    } else {
      L.out_h = sh;
      L.out_w = sw;
      L.ntaps = 1;
      L.nclass = 1;
      L.mh = sh; L.mw = sw;
    }
    if ((size_t)sh * sw * (size_t)(L.c0 > L.c1 ? L.c0 : L.c1) * esz >= ((size_t)1 << 31))
      return msi::fail(MSI_E_UNSUPPORTED, "net: %s input exceeds 2 GiB per sample", s.name);
    L.cpt0 = (L.c0 + bke - 1) / bke;
    L.cpt1 = (L.c1 + bke - 1) / bke;
    L.ksteps = L.ntaps * (L.cpt0 + L.cpt1);   // (the CoordNet channel is not a k-step: see the bias table below)
    L.npad = (int)round_up(L.cout, NPAD_ALIGN);
    // parameter blob (reference layout)
    const size_t wf = (s.kind == MODE_CONV)    ? (size_t)9 * (L.cin + L.has_coord) * L.cout
                      : (s.kind == MODE_CONVT) ? (size_t)16 * L.cout * L.cin
                                               : (size_t)L.cin * L.cout;
    L.param_off = poff;
    L.param_floats = wf + (s.kind == MODE_HEAD ? (size_t)L.cout : (size_t)2 * L.cout);
    poff += L.param_floats;
    // packed blob
    L.packed_off = koff;
    L.packed_w_floats = (size_t)L.nclass * L.ksteps * L.npad * (ROW_BYTES / 4);   // 128-byte rows in both types
    L.gamma_off = L.packed_off + L.packed_w_floats;
    L.beta_off = L.gamma_off + round_up(L.cout, 4);
    L.lnscl_off = L.beta_off + round_up(L.cout, 4);
    L.coord_off = L.lnscl_off + 2 * LN_SCL_DOUBLES;
    koff = L.coord_off + (L.has_coord ? (size_t)L.out_h * COORD_CLASSES * round_up(L.cout, 4) : 0);
    koff = round_up(koff, 64);
    // fp32 plans: the stride-1 one-source 3x3 layers also carry their weights as three bf16 planes (plan option F32_SPLIT3):
    // [tap][chunk of 32 channels][plane][npad rows][64 B]
    if (!bf16 && ((s.kind == MODE_CONV && s.src1 < 0 && L.c0 % 32 == 0) ||
                  (s.kind == MODE_CONVT && L.c0 % 32 == 0 && L.c1 % 32 == 0))) {
      L.x3_off = koff;
      koff = round_up(koff + (size_t)L.nclass * L.ksteps * 3 * L.npad * 16, 64);
      L.x2_off = koff;   // the same rows as two fp16 planes
      koff = round_up(koff + (size_t)L.nclass * L.ksteps * 2 * L.npad * 16, 64);
    }
    // workspace
    if (s.kind != MODE_HEAD) {
      L.raw_off = woff;
      woff += round_up((size_t)d->batch * L.out_h * L.out_w * L.cout * sizeof(float), 256);
      L.aff_off = woff;
      woff += round_up((size_t)d->batch * 2 * L.cout * sizeof(float), 256);
      if (bf16) {
        L.act_off = woff;
        woff += round_up((size_t)d->batch * L.out_h * L.out_w * L.cout * 2, 256);
      }
    } else {
      L.raw_off = (size_t)-1;
      L.aff_off = (size_t)-1;
    }
  }
  if (bf16) {
    const Layer &H = net.layers.back();
    net.head_f32_ksteps = (H.c0 + 31) / 32;
    net.head_f32_npad = (int)round_up(H.cout, 64);
    net.head_f32_off = koff;
    koff = round_up(koff + (size_t)net.head_f32_ksteps * net.head_f32_npad * (ROW_BYTES / 4), 64);
  }
  net.param_floats = poff;
  net.packed_floats = koff;
  net.partial_off = woff;
  // split tiles per launch: < num_cus remainder tiles, or < 2 num_cus when the first group is split too;
  // at most MAX_SPLIT K-ranges each, 64x64 fp32 accumulators per range
  net.partial_bytes = (size_t)2 * num_cus * MAX_SPLIT * 64 * 64 * sizeof(float);
  net.zero_off = net.partial_off + net.partial_bytes;
  net.cnt_off = net.zero_off;
  size_t zoff = net.cnt_off + round_up((size_t)MSI_NET_NUM_LAYERS * CONV_SLOTS_PER_CU * num_cus * sizeof(int), 256);
  for (int i = 0; i < MSI_NET_NUM_LAYERS; ++i) {
    net.layers[i].sums_off = zoff;
    if (net.layers[i].kind != MODE_HEAD) zoff += (size_t)d->batch * LN_SHARDS * LN_WORDS * sizeof(long long);
  }
  for (int i = 0; i < MSI_NET_NUM_LAYERS; ++i) {
    net.layers[i].flags_off = zoff;
    if (net.layers[i].kind != MODE_HEAD) zoff += (size_t)d->batch * net.layers[i].out_h * AP_FLAG_STRIDE * sizeof(int);
  }
  net.err_off = zoff;
  zoff += 64;
  net.zero_bytes = zoff - net.zero_off;
  net.ws_bytes = round_up(zoff, 256);
  return MSI_OK;
}

int device_cu_count() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
    (void)hipGetLastError();   // no device (build container): sizes for the default part
    return DEFAULT_CUS;
  }
  return prop.multiProcessorCount;
}

enum { TILE_64x64 = 0, TILE_128x128 = 1, TILE_128x64 = 2, TILE_64x128 = 3 };

// One layer's launch, everything but the pointers resolved at plan time.
struct LayerLaunch {
  ConvParams p;     // pointer members are filled per forward
  int tile;         // TILE_*
  int nblocks, nfix;
  int inlaunch;     // the split tiles are summed inside the conv launch (tickets) rather than by conv_fixup_kernel
  int fuse_ln;      // head: applies its source's LayerNorm while loading (the source is not normalised in memory)
  int skip_apply;   // this layer's output is consumed raw by the head, or normalised by its consumer's launch: no ln_apply launch
  int halo;         // conv_halo_kernel (fp32) / conv_halo_bf16_kernel instead of conv_igemm_kernel
  int halo_s2;      // ... conv_halo_s2_kernel: the stride-2 3x3 layers through parity-plane patches (fp32)
  int halo_x3;      // ... conv_halo_x3_kernel: fp32 through the 3-way bf16 split with six products (plan option F32_SPLIT3)
  int halo_x2;      // ... its fp16 form: 2-way split, three products (plan option F32_SPLIT_F16; needs halo_x3)
  int x3_th8;       // ... its 8 x 16-pixel tile (conv_halo8_x3_kernel: six-product form, rate 1; plan option X3_TILE8)
  int hbm, hbn;     // bf16 halo tile: 128 x 128 or 256 x 64
  int halo_t;       // convt_halo_kernel (conv-transpose, fp32): the two classes of one output-row parity per workgroup
  int halo_tb;      // convt_halo_bf16_kernel (conv-transpose, bf16): the two classes of one output-row parity per workgroup
  int halo_apply;   // ... applying the producer's LayerNorm while staging the patch (the producer's buffer stays raw)
  unsigned ln_blocks;
};

}  // namespace

struct msi_net_plan {
  msi_net_desc desc;
  int num_cus;
  int opt[MSI_NET_OPT_COUNT];
  Net net;
  LayerLaunch launch[MSI_NET_NUM_LAYERS];
};

namespace {

// Work decomposition of one layer ("tail split", see the kernel) for a BM x BN tile.
void plan_tiles(ConvParams &p, int BM, int BN, int batch, int num_cus, int tailsplit, int max_split, int *nblocks, int *nfix,
                int uniform_split = 0, int split_overhead = 0, bool split_any_tile = false) {
  const int mtot = p.Mh * p.Mw;
  p.tiles_m = (mtot + BM - 1) / BM;
  if (p.halo_tx) p.tiles_m = ((p.Mh + BM / 16 - 1) / (BM / 16)) * p.halo_tx;   // (BM / 16) x 16 spatial tiles (ragged at the right / bottom edge when Mh, Mw are no multiples)
  p.tiles_n = (p.Cout + BN - 1) / BN;
  p.ntiles = p.tiles_m * p.tiles_n * p.nclass * batch;
  // whole tiles in multiples of the CU count, the remainder cut into `split` K-ranges so that
  // (remainder x split) is again close to a multiple of the CU count
  p.n_main = p.ntiles;
  p.split0 = 1;
  p.split = 1;
  // Residency-aware form (tailsplit = 2; NOT the default -- measured slower, see below), for grids of at least one full
  // residency Q = 5 workgroups x CUs:
  // per-workgroup phase stamps (tools/conv_timing.py, r02_l) show the matrix pipes saturated while five workgroups
  // share a CU and starved when the last whole tile of a CU runs beside one short K-range -- e.g. 1 600 tiles = 6 whole
  // per CU + a quarter: the sixth tile ran with 2 waves per SIMD for a whole tile time (14 % of the launch).  So whole
  // tiles are issued in multiples of Q only, and the remaining < Q tiles are cut into K-ranges that fill one more
  // residency (1 600 -> 1 280 whole + 320 x 4 ranges; 3 200 -> 2 560 + 640 x 2): long blocks first, short ones last.
  // Measured (profiles/r02_m): conv2_1 115 -> 124 us, conv7_1 189 -> 205, conv1_1 325 -> 335: the extra K-range blocks
  // (prologue + epilogue + slab traffic each) cost more than the straggler they remove.
  const int Q = CONV_SLOTS_PER_CU * num_cus;
  if (BM * BN == 64 * 64 && tailsplit == 2 && p.ntiles >= Q && p.ntiles % Q != 0 && p.ksteps >= 2 * MAX_SPLIT) {
    const int remq = p.ntiles % Q;
    int best = 1;
    double best_cost = 1.0;   // time of the tail in tile-times: ceil(rem*s/Q)/s
    for (int sp = 2; sp <= max_split; ++sp) {
      if ((long)remq * sp > 2L * num_cus * MAX_SPLIT) break;     // slab capacity of the workspace
      const double cost = (double)((remq * sp + Q - 1) / Q) / sp;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    if (best > 1) { p.split = best; p.n_main = p.ntiles - remq; }
  }
  // Uniform split (plan option UNIFORM_SPLIT = s >= 2): layers with between one and two tiles per CU (the 40x80 ones: 400
  // tiles on 256 CUs) cut EVERY tile into s equal K-ranges instead of halves of the first group + sixths of the rest
  const bool uniform = uniform_split >= 2 && BM * BN == 64 * 64 && tailsplit && p.ntiles >= num_cus && p.ntiles < 2 * num_cus &&
                       uniform_split <= max_split && p.ksteps >= 2 * MAX_SPLIT;
  if (uniform) { p.n_main = 0; p.split0 = 1; p.split = uniform_split; }
  const int rem = p.ntiles % num_cus;
  if (!uniform && (BM * BN == 64 * 64 || split_any_tile) && p.split == 1 &&   // (the bf16 big tiles are only chosen for big grids)
      rem != 0 && p.ntiles > num_cus / 2 && p.ksteps >= 2 * MAX_SPLIT && tailsplit && !(tailsplit == 2 && p.ntiles >= Q)) {
    int best = 1;
    // time of the tail in k-steps: ceil(rem * s / CUs) rounds of K / s k-steps, each visit paying `split_overhead` k-steps of
    // prologue + epilogue (plan option SPLIT_OVERHEAD; 0 = the r01 rule, which minimises ceil(rem * s / CUs) / s alone)
    const double K = (double)p.ksteps, ovh = (double)split_overhead;
    double best_cost = K + ovh;
    for (int sp = 2; sp <= max_split; ++sp) {
      const double cost = (double)((rem * sp + num_cus - 1) / num_cus) * (K / sp + ovh);
      if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    if (best > 1) { p.split = best; p.n_main = p.ntiles - rem; }
    // One whole tile per CU next to four short K-ranges ends with that tile running alone (one wave per
    // SIMD, nothing to hide its barriers behind): such layers (CUs <= tiles < 2 CUs: the 40x80 ones) also cut
    // the first group in two.  Measured (r01): 2.750 -> 2.72 ms per frame, flat over 2..4 x 5..8.
    if (p.n_main == num_cus && p.split > 1 && p.ksteps >= 4 * MAX_SPLIT && max_split >= 2) {
      p.split0 = 2;
      if (split_overhead == 0) p.split = p.split > 6 ? 6 : p.split;   // remainder ranges not much shorter than the halves
    }
  }
  p.nb_main = p.n_main * p.split0;
  auto magic = [](int d) { return d == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)d); };
  p.mg_mw = magic(p.Mw); p.mg_tm = magic(p.tiles_m); p.mg_tn = magic(p.tiles_n); p.mg_nc = magic(p.nclass);
  p.mg_sp0 = magic(p.split0);
  p.mg_sp = magic(p.split);
  *nblocks = p.nb_main + (p.ntiles - p.n_main) * p.split;
  *nfix = (p.split0 > 1 ? p.n_main : 0) + (p.split > 1 ? p.ntiles - p.n_main : 0);
}

int plan_layers(msi_net_plan *pl) {
  const msi_net_desc *desc = &pl->desc;
  int rc = build_net(desc, pl->num_cus, pl->net);
  if (rc) return rc;
  const Net &net = pl->net;
  const int bf16 = desc->dtype == MSI_DTYPE_BF16;
  const int head_src = net.layers[MSI_NET_NUM_LAYERS - 1].src0;
  // The head (1x1, two k-steps, HBM-bound) applies its producer's LayerNorm + ReLU itself: one HBM round trip of
  // that activation less (fp32 only; option MSI_NET_OPT_HEAD_FUSE_LN = 0 restores the separate pass)
  const bool fuse_head_ln = !bf16 && pl->opt[MSI_NET_OPT_HEAD_FUSE_LN] && net.layers[head_src].cout <= HEAD_MAX_C;
  for (int li = 0; li < MSI_NET_NUM_LAYERS; ++li) {
    const Layer &L = net.layers[li];
    LayerLaunch &Q = pl->launch[li];
    memset(&Q, 0, sizeof(Q));
    ConvParams &p = Q.p;
    p.C0 = L.c0;
    p.C1 = L.src1 >= 0 ? L.c1 : 0;
    p.cb_stride = (int)round_up(L.cout, 4);
    p.Hin = L.in_h; p.Win = L.in_w; p.Hout = L.out_h; p.Wout = L.out_w;
    p.Cout = L.cout; p.npad = L.npad;
    p.ntaps = L.ntaps; p.cpt0 = L.cpt0; p.cpt1 = L.cpt1; p.ksteps = L.ksteps;
    p.mode = L.kind; p.nclass = L.nclass;
    p.wrap = desc->coord_net ? 0 : 1;
    p.rate = L.rate;
    p.Mh = L.mh; p.Mw = L.mw;
    p.stride = 1;
    if (L.kind == MODE_CONV) {
      p.stride = L.stride;
      if (desc->coord_net) {
        // TF SAME: total = max((out-1)*s + k_eff - in, 0), floor(total/2) before
        const int keff = 2 * L.rate + 1;
        const int th = (L.out_h - 1) * L.stride + keff - L.in_h, tw = (L.out_w - 1) * L.stride + keff - L.in_w;
        p.pad_t = (th > 0 ? th : 0) / 2;
        p.pad_l = (tw > 0 ? tw : 0) / 2;
      } else {
        p.pad_t = L.rate;  // wrap_pad(x, rate, rate) + VALID (nets.py:403-421)
        p.pad_l = L.rate;
      }
    } else if (L.kind == MODE_CONVT && L.wrapt) {
      p.pad_t = 0;   // tap v reads input row mh - v and padded column mw - v = image column mw - v - 2 (tap_delta)
      p.pad_l = 2;
    }
    if (L.kind == MODE_HEAD && fuse_head_ln) {
      Q.fuse_ln = 1;
      p.ln_inv_n = 1.0 / net.layers[L.src0].ln_count;
    }
    Q.skip_apply = fuse_head_ln && li == head_src;
    // bf16: at 4 MFMAs per k-step the 64x64 tile is bound by its LDS traffic; where the grid stays large
    // (>= 4 tiles per CU) and Cout allows, the 128x128 tile (64x64 per wave) halves that traffic per flop
    const long tiles_big = (long)((p.Mh * p.Mw + 127) / 128) * ((L.cout + 127) / 128) * L.nclass * desc->batch;
    const int bigmode = pl->opt[MSI_NET_OPT_BIGTILE];   // 0 = never, 1 = auto, 2 = whenever Cout allows (tests)
    int BM = 64, BN = 64;
    Q.tile = TILE_64x64;
    if (bf16 && L.cout % 128 == 0 && bigmode != 0 && (tiles_big >= 4L * pl->num_cus || bigmode == 2)) {
      Q.tile = TILE_128x128; BM = 128; BN = 128;
    } else if (bf16 && L.cout % 64 == 0 && bigmode != 0 && ((tiles_big >= 4L * pl->num_cus && L.cin <= 128) || bigmode == 2)) {
      Q.tile = TILE_128x64; BM = 128; BN = 64;   // Cout = 64, short K (conv8_2: 495 vs 599 us; conv1_1 / conv8_1 are faster at 64x64)
    }
    // fp32 tile experiments (per-layer mask in MSI_NET_OPT_F32_TILE_MASK): 128x64 (MT = 2) or 64x128 (NT = 2) instead of
    // 64x64 on the layers whose bit is set -- more MFMA work per prologue / epilogue and per DMA byte
    if (!bf16 && ((pl->opt[MSI_NET_OPT_F32_TILE_MASK] >> li) & 1) && L.kind != MODE_HEAD) {
      if (pl->opt[MSI_NET_OPT_F32_TILE] == 1) { Q.tile = TILE_128x64; BM = 128; BN = 64; }
      else if (pl->opt[MSI_NET_OPT_F32_TILE] == 2 && L.cout % 128 == 0) { Q.tile = TILE_64x128; BM = 64; BN = 128; }
    }
    // halo-patch kernel (conv_halo_kernel): stride-1 3x3 layers with one source, fp32, whole 4 x 16 tiles and 32-channel chunks
    const bool halo_ok = !((pl->opt[MSI_NET_OPT_HALO_SKIP] >> li) & 1);
    Q.halo = halo_ok && (pl->opt[MSI_NET_OPT_HALO] & 1) && !bf16 && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] && Q.tile == TILE_64x64 &&
             L.kind == MODE_CONV && L.stride == 1 && L.src1 < 0 && L.in_h % 4 == 0 && L.in_w % 16 == 0 && L.c0 % 32 == 0 &&
             (L.rate == 1 || L.rate == 2);
    // stride-2 halo kernel (conv_halo_s2_kernel; HALO bit 2): the stride-2 3x3 layers, fp32, one source, whole 4 x 16 tiles of the
    // OUTPUT grid, an even input (TF SAME then pads one row / column at the far side only) or wrap_pad(1, 1) + VALID
    Q.halo_s2 = halo_ok && (pl->opt[MSI_NET_OPT_HALO] & 4) && !bf16 && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] && Q.tile == TILE_64x64 &&
                L.kind == MODE_CONV && L.stride == 2 && L.rate == 1 && L.src1 < 0 && L.in_h % 2 == 0 && L.in_w % 2 == 0 &&
                L.out_h % 4 == 0 && L.out_w % 16 == 0 && L.c0 % 32 == 0 && p.pad_t == p.pad_l && (p.pad_t == 0 || p.pad_t == 1) &&
                // (measured at 640 x 320: conv1_2 / conv2_2 gain their producers' ln_apply launches, -22 / -11 us for +4 / +3 us of
                // kernel time; conv3_3, 400 tiles cut into K-ranges of two groups, loses 11 us to save 6: tap kernel)
                ((long)(L.out_h / 4) * (L.out_w / 16) * (L.cout / 64) * desc->batch >= 3L * pl->num_cus ||
                 // (r04: through the six-product split the halo form wins on conv3_3's 400 tiles as well: 76 -> 5x us)
                 (L.x3_off != 0 && ((pl->opt[MSI_NET_OPT_F32_SPLIT3] >> li) & 1) && !(pl->opt[MSI_NET_OPT_HALO_SKIP] >> 20 & 1)));
    if (Q.halo_s2) Q.halo = 1;
    Q.halo_x3 = Q.halo && !bf16 && L.x3_off != 0 && ((pl->opt[MSI_NET_OPT_F32_SPLIT3] >> li) & 1);
    Q.halo_x2 = Q.halo_x3 && ((pl->opt[MSI_NET_OPT_F32_SPLIT_F16] >> li) & 1);
    // the 8 x 16-pixel tile of the six-product form (conv_halo8_x3_kernel): stride 1, rate 1, whole 8-row tiles
    // where the grid stays >= 3 tiles per CU (measured at 640 x 320, profiles/r05_tile8.txt: conv1_1 215 -> 201, conv2_1 80 -> 74, conv7_2 82 -> 75, conv8_2 87 -> 81 us;
    // the 400-tile layers conv3_x / conv6_x, cut into K-ranges either way, LOSE 12 %); bit 30 of the option forces it on every eligible layer (tests)
    Q.x3_th8 = Q.halo_x3 && !Q.halo_x2 && !Q.halo_s2 && L.rate == 1 && L.in_h % 8 == 0 && ((pl->opt[MSI_NET_OPT_X3_TILE8] >> li) & 1) &&
               ((long)(L.in_h / 8) * (L.in_w / 16) * ((L.cout + 63) / 64) * desc->batch >= 3L * pl->num_cus || ((pl->opt[MSI_NET_OPT_X3_TILE8] >> 30) & 1));
    if (Q.x3_th8) BM = 128;
    int max_split = MAX_SPLIT;
    // bf16 halo-patch kernel (conv_halo_bf16_kernel): the same layers with 64-channel chunks and whole
    // 8 x 16 pixel x 128 channel or 16 x 16 x 64 tiles
    if (halo_ok && (pl->opt[MSI_NET_OPT_HALO] & 1) && bf16 && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] && L.kind == MODE_CONV && L.stride == 1 && L.src1 < 0 && L.in_w % 16 == 0 &&
        L.c0 % 64 == 0 && bigmode != 0) {
      if (L.cout % 128 == 0 && L.in_h % 8 == 0 && (L.rate == 1 || L.rate == 2)) { Q.halo = 1; Q.hbm = 128; Q.hbn = 128; }
      else if (L.cout == 64 && L.in_h % 16 == 0 && L.rate == 1) { Q.halo = 1; Q.hbm = 256; Q.hbn = 64; }
      if (Q.halo) { BM = Q.hbm; BN = Q.hbn; max_split = 1; }
    }
    // ... and its stride-2 form (conv_halo_bf16_s2_kernel; HALO bit 2): whole 8 x 16 x 128 tiles of the OUTPUT grid, an even input
    if (halo_ok && (pl->opt[MSI_NET_OPT_HALO] & 4) && bf16 && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] && L.kind == MODE_CONV && L.stride == 2 && L.rate == 1 &&
        L.src1 < 0 && L.in_h % 2 == 0 && L.in_w % 2 == 0 && L.out_h % 8 == 0 && L.out_w % 16 == 0 && L.c0 % 64 == 0 && L.cout % 128 == 0 &&
        p.pad_t == p.pad_l && (p.pad_t == 0 || p.pad_t == 1) && bigmode != 0) {
      Q.halo = 1; Q.halo_s2 = 1; Q.hbm = 128; Q.hbn = 128; BM = 128; BN = 128; max_split = 1;
    }
    // conv-transpose halo kernel (convt_halo_kernel; HALO bit 1, NOT the default -- measured slower, see the kernel): SAME conv-transposes (CoordNet), fp32, whole
    // 4 x 16 input tiles and 32-channel chunks of both sources; one workgroup per output-row parity (enumerated as two "classes")
    const bool x3_on = !bf16 && L.x3_off != 0 && ((pl->opt[MSI_NET_OPT_F32_SPLIT3] >> li) & 1);
    Q.halo_t = halo_ok && ((pl->opt[MSI_NET_OPT_HALO] & 2) || (x3_on && pl->opt[MSI_NET_OPT_HALO] != 0)) && !bf16   // (HALO = 0: no halo-patch kernel at all)
               && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] &&
               Q.tile == TILE_64x64 && L.kind == MODE_CONVT && ((!L.wrapt && L.in_h % 4 == 0 && L.in_w % 16 == 0) || (L.wrapt && x3_on)) &&
               L.c0 % 32 == 0 && L.c1 % 32 == 0;   // (wrapt: (H + 1) x (W + 5) GEMM rows per class in ragged 4 x 16 tiles -- the split form only)
    if (Q.halo_t) {
      Q.halo = 1;
      Q.halo_x3 = x3_on;
      // (r04 kept msi_train_net's VALID transposes off the fp16 form: one wave's share of the layer's sum of squares came out low in ~0.1 % of
      // back-to-back forwards.  r05 found the instruction: a compiler-made `v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` of the generic
      // epilogue's statistics -- low lane = a.lo * b.HI -- evaluated to 0 for lanes 48-63; this file is now built with -fno-slp-vectorize, which
      // is what forms that operand routing, and matryodshka_amd/build.py refuses a library that contains it.  DESIGN.md section 4, "the wobble".)
      Q.halo_x2 = x3_on && ((pl->opt[MSI_NET_OPT_F32_SPLIT_F16] >> li) & 1);
      p.nclass = 2;                                      // tiles are enumerated per (ph, tile_m, tile_n, sample): a workgroup owns pw = 0, 1
      if (L.cpt0 + L.cpt1 < max_split) max_split = L.cpt0 + L.cpt1;
    }
    // bf16 conv-transpose halo kernel (convt_halo_bf16_kernel): SAME conv-transposes, 64-channel chunks of both sources,
    // whole 8 x 16 x 128 or 16 x 16 x 64 tiles, one workgroup per output-row parity (enumerated as two "classes")
    if (halo_ok && (pl->opt[MSI_NET_OPT_HALO] & 1) && bf16 && !pl->opt[MSI_NET_OPT_APPLY_AHEAD] && L.kind == MODE_CONVT && !L.wrapt &&
        L.in_w % 16 == 0 && L.c0 % 64 == 0 && L.c1 % 64 == 0 && bigmode != 0) {
      if (L.cout % 128 == 0 && L.in_h % 8 == 0) { Q.halo_tb = 1; Q.hbm = 128; Q.hbn = 128; }
      else if (L.cout == 64 && L.in_h % 8 == 0) { Q.halo_tb = 1; Q.hbm = 128; Q.hbn = 64; }   // (256 x 64 with two classes spills: 128 accumulator + 80 fragment registers)
      if (Q.halo_tb) { Q.halo = 1; BM = Q.hbm; BN = Q.hbn; max_split = 1; p.nclass = 2; }
    }
    if (Q.halo) {
      p.halo_tx = (Q.halo_s2 ? L.out_w : L.in_w) / 16;
      if (Q.halo_t && L.wrapt) p.halo_tx = (L.mw + 15) / 16;
      p.halo_xor = bf16 ? 0 : 8;
      p.mg_htx = p.halo_tx == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)p.halo_tx);
      if (!Q.halo_t && L.cpt0 < max_split) max_split = L.cpt0;      // K-ranges are whole chunks (bf16: whole tiles only)
    }
    plan_tiles(p, BM, BN, desc->batch, pl->num_cus, pl->opt[MSI_NET_OPT_TAILSPLIT], max_split, &Q.nblocks, &Q.nfix,
               pl->opt[MSI_NET_OPT_UNIFORM_SPLIT], pl->opt[MSI_NET_OPT_SPLIT_OVERHEAD], Q.x3_th8 != 0);
    // apply-ahead (see apply_ahead): this launch also normalises its source 0
    if (pl->opt[MSI_NET_OPT_APPLY_AHEAD] && !bf16 && L.src0 >= 0 && L.kind != MODE_HEAD && L.c0 <= 512 && L.c0 % 4 == 0 &&   // (bf16: fp16 raw outputs, r03)
        ((long)L.in_w * L.c0) % 4 == 0) {
      constexpr int UNIT_VEC = 2048;   // float4 per unit: 32 KB of fp32
      p.ap_row_vec = L.in_w * L.c0 / 4;
      p.ap_unit_vec = UNIT_VEC;
      p.ap_units_per_row = (p.ap_row_vec + UNIT_VEC - 1) / UNIT_VEC;
      p.ap_inv_n = 1.0 / net.layers[L.src0].ln_count;
      const long units = (long)desc->batch * L.in_h * p.ap_units_per_row;
      long n = 2L * pl->num_cus;                      // two apply workgroups per CU keep ~8 MB of loads in flight
      if (n > units) n = units;
      p.n_apply = (int)((n + 7) / 8 * 8);
      pl->launch[L.src0].skip_apply = 1;              // (the producer precedes its consumer in graph order)
    }
    Q.inlaunch = !pl->opt[MSI_NET_OPT_FIXUP_KERNEL] && Q.nfix <= CONV_SLOTS_PER_CU * pl->num_cus;
    if (Q.halo_t && (size_t)(Q.nblocks - (p.split0 == 1 ? p.nb_main : 0)) * 2 * BM * BN * sizeof(float) > net.partial_bytes) {
      // two slabs per K-range do not fit the partial-accumulator workspace -> the tap kernel
      Q.halo_t = 0; Q.halo = 0; Q.halo_x3 = 0; Q.halo_x2 = 0;
      p.halo_tx = 0; p.halo_xor = 0; p.nclass = L.nclass;
      plan_tiles(p, BM, BN, desc->batch, pl->num_cus, pl->opt[MSI_NET_OPT_TAILSPLIT], MAX_SPLIT, &Q.nblocks, &Q.nfix);
      Q.inlaunch = !pl->opt[MSI_NET_OPT_FIXUP_KERNEL] && Q.nfix <= CONV_SLOTS_PER_CU * pl->num_cus;
    }
    if ((size_t)(Q.nblocks - (p.split0 == 1 ? p.nb_main : 0)) * (Q.halo_t ? 2 : 1) * BM * BN * sizeof(float) > net.partial_bytes)
      return msi::fail(MSI_E_WORKSPACE, "conv %s: %d partial accumulators exceed the workspace", L.name, Q.nblocks);
    if (L.kind != MODE_HEAD) {
      const size_t per_sample = (size_t)L.out_h * L.out_w * L.cout;
      size_t blocks = (per_sample / 4 + 255) / 256;
      if (blocks > 1024) blocks = 1024;  // grid-stride
      Q.ln_blocks = (unsigned)blocks;
    }
  }
  // A layer whose EVERY consumer can apply its LayerNorm while staging a patch is never normalised in memory: halo conv
  // layers (their one source) and conv-transpose halo layers (either source).  (Until r03 the bf16 256x64 tile and the
  // bf16 conv-transpose halo kernel read bf16 copies only: with fp32 raw outputs they had no registers for the staging;
  // the fp16 raw output is 16 bytes per 8-channel slot like the copy.)
  for (int s = 0; s < MSI_NET_NUM_LAYERS - 1; ++s) {
    int consumers = 0, capable = 0;
    for (int li = s + 1; li < MSI_NET_NUM_LAYERS; ++li) {
      const Layer &L = net.layers[li];
      if (L.src0 == s || L.src1 == s) {
        ++consumers;
        const LayerLaunch &C = pl->launch[li];
        // (bf16 conv-transposes: only the 128 x 64 tile has registers for the staging -- 128 x 128 with APPLY spills)
        const int stage_raw = pl->opt[MSI_NET_OPT_BF16_STAGE_RAW];   // bit 0: the 256 x 64 conv tile, bit 1: the 128 x 64 conv-transpose tile
        if (C.halo_t || (C.halo_tb && C.hbn == 64 && (stage_raw & 2)) ||
            (C.halo && !C.halo_tb && L.src0 == s && (!bf16 || (L.c0 <= 512 && (C.hbm != 256 || (stage_raw & 1)))))) ++capable;
      }
    }
    if (consumers > 0 && consumers == capable) {
      pl->launch[s].skip_apply = 1;
      for (int li = s + 1; li < MSI_NET_NUM_LAYERS; ++li) {
        const Layer &L = net.layers[li];
        LayerLaunch &C = pl->launch[li];
        if (C.halo_t || C.halo_tb) {
          if (L.src0 == s) { C.p.halo_apply |= 1; C.p.ln_inv_n = 1.0 / net.layers[s].ln_count; }
          if (L.src1 == s) { C.p.halo_apply |= 2; C.p.ln_inv_n1 = 1.0 / net.layers[s].ln_count; }
        } else if (L.src0 == s) {
          C.halo_apply = 1;
          C.p.ln_inv_n = 1.0 / net.layers[s].ln_count;
        }
      }
    }
  }
  return MSI_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A DEVICE: set once per (instantiation,
// device) -- `done` is a per-instantiation bit mask over the device ordinal, so a thread that drives a second GPU sets
// the attribute there as well (ordinals >= 64: set on every launch).
inline int set_max_lds(const void *fn, int lds, unsigned long long &done, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && ((done >> dev) & 1ull)) return MSI_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return msi::fail(MSI_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  if (tracked) done |= 1ull << dev;
  return MSI_OK;
}

template <int BM, int BN, int MODE, int BF16>
int launch_conv_mode(const LayerLaunch &Q, const ConvParams &p, hipStream_t stream) {
  const size_t lds = (size_t)NSTAGE * (BM + BN) * ROW_BYTES;
  if (lds > 64 * 1024) {
    static thread_local unsigned long long done = 0;
    int rc0 = set_max_lds(reinterpret_cast<const void *>(conv_igemm_kernel<BM, BN, MODE, BF16>), (int)lds, done, "conv");
    if (rc0) return rc0;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, MODE, BF16>), dim3(Q.nblocks + p.n_apply), dim3(256), lds, stream, p);
  int rc = msi::check_launch("conv_igemm");
  if (rc || Q.nfix == 0 || p.tile_cnt != nullptr) return rc;
  if constexpr (BM * BN == 64 * 64) {   // (big tiles are never split)
    hipLaunchKernelGGL((conv_fixup_kernel<BM, BN, MODE, (BF16 && MODE != MODE_HEAD) ? 1 : 0>), dim3(Q.nfix), dim3(256), 0, stream, p);
    return msi::check_launch("conv_fixup");
  } else {
    return msi::fail(MSI_E_UNSUPPORTED, "conv: split big tile");
  }
}

template <int BM, int BN, int RATE, int APPLY, int NW>
int launch_halo_bf16(const LayerLaunch &Q, const ConvParams &p, hipStream_t stream) {
#ifdef MSI_ONE_PER_CU   // timing experiment: one workgroup per CU (no co-resident workgroup's MFMAs)
  constexpr int lds = 100 * 1024;
#else
  constexpr int lds = HaloGeomB<BM, BN, RATE>::LDS_BYTES + (APPLY ? 8 * 512 : 0);   // + scale | shift of <= 512 input channels (4 KB: two workgroups per CU still fit)
#endif
  static_assert(lds >= EPI_STAGE_BYTES, "the epilogue's staging strips");
  if (APPLY && p.C0 > 512) return msi::fail(MSI_E_UNSUPPORTED, "conv_halo_bf16: APPLY with more than 512 input channels");
  static thread_local unsigned long long done = 0;
  int rc0 = set_max_lds(reinterpret_cast<const void *>(conv_halo_bf16_kernel<BM, BN, RATE, APPLY, NW>), lds, done, "conv_halo_bf16");
  if (rc0) return rc0;
  hipLaunchKernelGGL((conv_halo_bf16_kernel<BM, BN, RATE, APPLY, NW>), dim3(Q.nblocks), dim3(64 * NW), lds, stream, p);
  return msi::check_launch("conv_halo_bf16");
}

template <int APPLY, int NW>
int launch_halo_bf16_s2(const LayerLaunch &Q, const ConvParams &p, hipStream_t stream) {
  constexpr int lds = HaloGeomBS2::LDS_BYTES + (APPLY ? 8 * 512 : 0);
  static_assert(lds >= EPI_STAGE_BYTES && 2 * lds <= 160 * 1024, "staging strips; two workgroups per CU");
  if (APPLY && p.C0 > 512) return msi::fail(MSI_E_UNSUPPORTED, "conv_halo_bf16_s2: APPLY with more than 512 input channels");
  static thread_local unsigned long long done = 0;
  int rc0 = set_max_lds(reinterpret_cast<const void *>(conv_halo_bf16_s2_kernel<APPLY, NW>), lds, done, "conv_halo_bf16_s2");
  if (rc0) return rc0;
  hipLaunchKernelGGL((conv_halo_bf16_s2_kernel<APPLY, NW>), dim3(Q.nblocks), dim3(64 * NW), lds, stream, p);
  return msi::check_launch("conv_halo_bf16_s2");
}

template <int BM, int BN, int APPLY>
int launch_convt_halo_bf16(const LayerLaunch &Q, const ConvParams &p, hipStream_t stream) {
  constexpr int lds = HaloGeomB<BM, BN, 1>::LDS_BYTES;
  static thread_local unsigned long long done = 0;
  int rc0 = set_max_lds(reinterpret_cast<const void *>(convt_halo_bf16_kernel<BM, BN, APPLY>), lds, done, "convt_halo_bf16");
  if (rc0) return rc0;
  hipLaunchKernelGGL((convt_halo_bf16_kernel<BM, BN, APPLY>), dim3(Q.nblocks), dim3(256), lds, stream, p);
  return msi::check_launch("convt_halo_bf16");
}

template <int BM, int BN>
int launch_conv(const LayerLaunch &Q, const ConvParams &p, int bf16, hipStream_t stream) {
  if constexpr (BM == 64 && BN == 128) {
    if (bf16) return msi::fail(MSI_E_UNSUPPORTED, "conv: 64x128 is an fp32 tile");
  } else if (bf16) {
    switch (p.mode) {
      case MODE_CONV: return launch_conv_mode<BM, BN, MODE_CONV, 1>(Q, p, stream);
      case MODE_CONVT: return launch_conv_mode<BM, BN, MODE_CONVT, 1>(Q, p, stream);
      default: return launch_conv_mode<BM, BN, MODE_HEAD, 1>(Q, p, stream);
    }
  }
  if constexpr (BM * BN == 64 * 64) {
    switch (p.mode) {
      case MODE_CONV: return launch_conv_mode<BM, BN, MODE_CONV, 0>(Q, p, stream);
      case MODE_CONVT: return launch_conv_mode<BM, BN, MODE_CONVT, 0>(Q, p, stream);
      default: return launch_conv_mode<BM, BN, MODE_HEAD, 0>(Q, p, stream);
    }
#ifdef MSI_EXPERIMENTS
  } else if constexpr (BM * BN == 128 * 64) {
    switch (p.mode) {
      case MODE_CONV: return launch_conv_mode<BM, BN, MODE_CONV, 0>(Q, p, stream);
      case MODE_CONVT: return launch_conv_mode<BM, BN, MODE_CONVT, 0>(Q, p, stream);
      default: return msi::fail(MSI_E_UNSUPPORTED, "conv: fp32 head uses the 64x64 tile");
    }
#endif
  } else {
    return msi::fail(MSI_E_UNSUPPORTED, "conv: the fp32 path is built for the 64x64, 128x64 and 64x128 tiles");
  }
}

}  // namespace

#if defined(MSI_CONV_TIMING) || defined(MSI_DEBUG_STATS)
// tools/conv_timing.py: per-workgroup phase stamps of one layer's conv launch (debug builds only)
static unsigned long long *g_timing_buf = nullptr;
static int g_timing_layer = -1;
extern "C" int msi_debug_conv_occupancy(int lds_bytes) {
  int n = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<64, 64, MODE_CONV, 0>, 256, (size_t)lds_bytes);
  return n;
}
extern "C" void msi_debug_conv_timing(void *device_buffer, int layer) {
  g_timing_buf = static_cast<unsigned long long *>(device_buffer);
  g_timing_layer = layer;
}
#endif

extern "C" {

int msi_net_layer_info(const msi_net_desc *desc, int32_t layer, msi_layer_info *out) {
  Net net;
  int rc = build_net(desc, DEFAULT_CUS, net);
  if (rc) return rc;
  MSI_REQUIRE(out && layer >= 0 && layer < MSI_NET_NUM_LAYERS, "net_layer_info: bad layer %d", layer);
  const Layer &L = net.layers[layer];
  memset(out, 0, sizeof(*out));
  strncpy(out->name, L.name, sizeof(out->name) - 1);
  out->kind = L.kind; out->cin = L.cin; out->cout = L.cout; out->has_coord = L.has_coord;
  out->stride = L.stride; out->rate = L.rate;
  out->in_h = L.in_h; out->in_w = L.in_w; out->out_h = L.out_h; out->out_w = L.out_w;
  out->param_offset = L.param_off; out->param_floats = L.param_floats;
  out->raw_offset = (uint64_t)L.raw_off; out->affine_offset = (uint64_t)L.aff_off;
  out->ln_scale_offset = (uint64_t)L.lnscl_off;
  return MSI_OK;
}

size_t msi_net_param_floats(const msi_net_desc *desc) {
  Net net;
  return build_net(desc, DEFAULT_CUS, net) ? 0 : net.param_floats;
}

size_t msi_net_packed_floats(const msi_net_desc *desc) {
  Net net;
  return build_net(desc, DEFAULT_CUS, net) ? 0 : net.packed_floats;
}

int msi_net_pack_weights_host(const msi_net_desc *desc, const float *params, float *packed) {
  Net net;
  int rc = build_net(desc, DEFAULT_CUS, net);
  if (rc) return rc;
  MSI_REQUIRE(params && packed, "net_pack_weights: null pointer");
  memset(packed, 0, net.packed_floats * sizeof(float));
  const int bf16 = desc->dtype == MSI_DTYPE_BF16;
  const int bke = bf16 ? 64 : 32;
  // element kk of a 128-byte row: 16-byte chunk (kk * esz / 16) goes to slot chunk ^ swz
  auto put_elem = [bf16](char *row, int kk, int swz, float v) {
    if (bf16) {
      uint32_t u;
      memcpy(&u, &v, 4);
      const uint16_t h = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);   // round to nearest even
      memcpy(row + (((kk >> 3) ^ swz) << 4) + (kk & 7) * 2, &h, 2);
    } else {
      memcpy(row + (((kk >> 2) ^ swz) << 4) + (kk & 3) * 4, &v, 4);
    }
  };
  for (const Layer &L : net.layers) {
    const float *w = params + L.param_off;
    float *o = packed + L.packed_off;
    const int cin_w = L.cin + L.has_coord;  // channel extent of the TF weight tensor
    const int cpt = L.cpt0 + L.cpt1;
    for (int cls = 0; cls < L.nclass; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      for (int s = 0; s < L.ksteps; ++s) {
        // k-step order of the kernel's generator: tap-major, then source 0 chunks, then source 1 chunks
        const int tap_s = s / cpt;
        const int within = s % cpt;
        const int src = within < L.cpt0 ? 0 : 1;
        const int chunk = src ? within - L.cpt0 : within;
        const int csrc = src ? L.c1 : L.c0, cbase = src ? L.c0 : 0;
        for (int n = 0; n < L.cout; ++n) {
          char *row = reinterpret_cast<char *>(o) + (((size_t)cls * L.ksteps + s) * L.npad + n) * ROW_BYTES;
          const int swz = (n >> 1) & 7;  // LDS slot j of row n holds data chunk j ^ swz (see the kernel)
          for (int kk = 0; kk < bke; ++kk) {
            if (chunk * bke + kk >= csrc) continue;
            const int tap = tap_s, c = cbase + chunk * bke + kk;
            float v;
            if (L.kind == MODE_CONV) {            // [3,3,cin_w,cout]
              v = w[((size_t)tap * cin_w + c) * L.cout + n];
            } else if (L.kind == MODE_CONVT) {    // [4,4,cout,cin]
              const int th = tap >> 1, tw = tap & 1;
              // SAME: y[2i+k-1] += x[i] w[k] (see tap_delta); VALID over the wrap-padded input: k = parity + 2 v
              const int kh = L.wrapt ? ph + 2 * th : (ph == 0 ? 1 + 2 * th : 2 - 2 * th);
              const int kw = L.wrapt ? pw + 2 * tw : (pw == 0 ? 1 + 2 * tw : 2 - 2 * tw);
              v = w[(((size_t)kh * 4 + kw) * L.cout + n) * L.cin + c];
            } else {                              // [1,1,cin,cout]
              v = w[(size_t)c * L.cout + n];
            }
            put_elem(row, kk, swz, v);
          }
        }
      }
    }
    const size_t wf = L.param_floats - (L.kind == MODE_HEAD ? (size_t)L.cout : (size_t)2 * L.cout);
    if (L.kind != MODE_HEAD) {
      // fixed-point window of this layer's LayerNorm sums (see LN_S1_BITS): e = round(log2(expected rms of the raw output)),
      // expected rms = sqrt(K) * rms(w) * rms(input), K = products per output, rms(input) = 0.5 for the sweep volume
      // (images in [-1, 1]) and sqrt(mean(gamma^2) / 2 + mean(beta^2)) for a LayerNorm + ReLU'd producer
      double sw = 0.0;
      for (size_t i = 0; i < wf; ++i) sw += (double)w[i] * (double)w[i];
      const double rms_w = sqrt(sw / (double)(wf ? wf : 1));
      auto in_ms = [&](int src) -> double {
        if (src < 0) return 0.25;
        const Layer &S = net.layers[src];
        const float *g = params + S.param_off + (S.param_floats - 2 * (size_t)S.cout), *be = g + S.cout;
        double sg = 0.0, sb = 0.0;
        for (int c = 0; c < S.cout; ++c) { sg += (double)g[c] * g[c]; sb += (double)be[c] * be[c]; }
        return 0.5 * sg / S.cout + sb / S.cout;
      };
      double ms_in = in_ms(L.src0);
      if (L.src1 >= 0) ms_in = (ms_in * L.c0 + in_ms(L.src1) * L.c1) / (double)(L.c0 + L.c1);
      const double K = (L.kind == MODE_CONV ? 9.0 : 4.0) * (double)(L.cin + L.has_coord);
      const double est = sqrt(K * ms_in) * rms_w;
      int e = (est > 0.0 && std::isfinite(est)) ? (int)lrint(log2(est)) : 0;
      e = e < -60 ? -60 : (e > 60 ? 60 : e);
      const double scl[LN_SCL_DOUBLES] = {ldexp(1.0, LN_S1_BITS - e), ldexp(1.0, LN_S2_BITS - 2 * e),
                                          ldexp(1.0, -(LN_S1_BITS - e)), ldexp(1.0, -(LN_S2_BITS - 2 * e))};
      memcpy(packed + L.lnscl_off, scl, sizeof(scl));
    }
    if (L.x3_off) {
      // x3 block (conv_halo_x3_kernel and its stride-2 / conv-transpose forms): the k-steps of the loop above, each as three planes
      // of 64-byte rows -- [class][k-step][plane h | m | l][npad][32 bf16] -- with w = h + m + l, bf16 parts by round-to-nearest-even
      // of the successive (exact) remainders; 16-byte slot j (channels 8 j .. 8 j + 7 of the chunk) of row n is stored at slot
      // j ^ ((n >> 2) & 3) (HaloGeomX3: conflict-free fragment reads)
      auto bf16_rne = [](float v) -> uint16_t {
        uint32_t u;
        memcpy(&u, &v, 4);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      };
      auto widen = [](uint16_t h) -> float {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
      };
      char *base = reinterpret_cast<char *>(packed + L.x3_off);
      for (int cls = 0; cls < L.nclass; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        for (int s = 0; s < L.ksteps; ++s) {
          const int tap = s / cpt, within = s % cpt;
          const int src = within < L.cpt0 ? 0 : 1;
          const int chunk = src ? within - L.cpt0 : within;
          const int cbase = src ? L.c0 : 0;
          for (int n = 0; n < L.cout; ++n)
            for (int kk = 0; kk < 32; ++kk) {
              const int c = cbase + chunk * 32 + kk;
              float v;
              if (L.kind == MODE_CONV) {
                v = w[((size_t)tap * cin_w + c) * L.cout + n];
              } else {   // MODE_CONVT: SAME, or VALID over the wrap-padded input (kernel index = parity + 2 tap: see tap_delta)
                const int th = tap >> 1, tw = tap & 1;
                const int kh = L.wrapt ? ph + 2 * th : (ph == 0 ? 1 + 2 * th : 2 - 2 * th);
                const int kw = L.wrapt ? pw + 2 * tw : (pw == 0 ? 1 + 2 * tw : 2 - 2 * tw);
                v = w[(((size_t)kh * 4 + kw) * L.cout + n) * L.cin + c];
              }
              uint16_t part[3];
              part[0] = bf16_rne(v);
              const float r1 = v - widen(part[0]);
              part[1] = bf16_rne(r1);
              part[2] = bf16_rne(r1 - widen(part[1]));
              const int slot = (kk >> 3) ^ ((n >> 2) & 3);
              for (int pl = 0; pl < 3; ++pl)
                memcpy(base + ((((size_t)cls * L.ksteps + s) * 3 + pl) * L.npad + n) * 64 + slot * 16 + (kk & 7) * 2, &part[pl], 2);
              // x2 block: w = h + m' 2^-11 with fp16 parts (round to nearest even; w - h is exact, m' keeps 11 of its bits:
              // 22 significand bits in all).  |w| > 65504 packs as inf and poisons the layer (LN_OVERFLOW in the status word)
              const _Float16 hh = (_Float16)v;
              const _Float16 hm = (_Float16)((v - (float)hh) * 2048.f);
              char *base2 = reinterpret_cast<char *>(packed + L.x2_off);
              memcpy(base2 + ((((size_t)cls * L.ksteps + s) * 2 + 0) * L.npad + n) * 64 + slot * 16 + (kk & 7) * 2, &hh, 2);
              memcpy(base2 + ((((size_t)cls * L.ksteps + s) * 2 + 1) * L.npad + n) * 64 + slot * 16 + (kk & 7) * 2, &hm, 2);
            }
        }
      }
    }
    if (L.kind == MODE_HEAD) {
      memcpy(packed + L.gamma_off, w + wf, L.cout * sizeof(float));  // biases
    } else {
      memcpy(packed + L.gamma_off, w + wf, L.cout * sizeof(float));
      memcpy(packed + L.beta_off, w + wf + L.cout, L.cout * sizeof(float));
    }
    if (L.has_coord) {
      // nets.add_sph_coords (nets.py:260-265): the extra input channel abs(sin(np.linspace(-pi/2, pi/2, H)))
      // (fp64 -> fp32) is constant along W and independent of the image, so its share of the 3x3
      // convolution is tabulated here instead of being computed per frame:
      //   bias[out_row][column class][n] = sum over the taps (kh,kw) that land inside the image of
      //   coord[ih] * w[kh][kw][cin][n]     (zero padding elsewhere; column classes = the two border
      //   columns on each side | interior), accumulated in fp64, stored fp32 and added to the fp32
      //   accumulators in the conv epilogue.  In the bf16 path both factors are rounded to bf16 first
      //   (they are convolution operands there).
      const double PI = 3.14159265358979323846;
      const double start = -PI / 2.0, stop = PI / 2.0;
      const int h = L.in_h;
      const double step = h > 1 ? (stop - start) / (h - 1) : 0.0;
      auto operand = [bf16](float v) -> double {
        if (!bf16) return (double)v;
        uint32_t u;
        memcpy(&u, &v, 4);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        float r;
        memcpy(&r, &u, 4);
        return (double)r;
      };
      std::vector<double> coord(h);
      for (int i = 0; i < h; ++i) {
        double a = (double)i * step + start;
        if (i == h - 1 && h > 1) a = stop;
        coord[i] = operand((float)fabs(sin(a)));
      }
      const int keff = 2 * L.rate + 1;
      const int th = (L.out_h - 1) * L.stride + keff - L.in_h, tw = (L.out_w - 1) * L.stride + keff - L.in_w;
      const int pad_t = (th > 0 ? th : 0) / 2, pad_l = (tw > 0 ? tw : 0) / 2;  // TF SAME (CoordNet only)
      const int reps[COORD_CLASSES] = {0, 1, 2, L.out_w - 2, L.out_w - 1};
      const size_t cbs = round_up(L.cout, 4);
      float *tab = packed + L.coord_off;
      for (int mh = 0; mh < L.out_h; ++mh)
        for (int cc = 0; cc < COORD_CLASSES; ++cc) {
          const int mw = reps[cc];
          if (mw < 0 || mw >= L.out_w) continue;
          for (int n = 0; n < L.cout; ++n) {
            double acc = 0.0;
            for (int tap = 0; tap < 9; ++tap) {
              const int kh = tap / 3, kw = tap % 3;
              const int ih = mh * L.stride - pad_t + kh * L.rate, iw = mw * L.stride - pad_l + kw * L.rate;
              if (ih < 0 || ih >= L.in_h || iw < 0 || iw >= L.in_w) continue;
              acc += coord[ih] * operand(w[((size_t)tap * cin_w + L.cin) * L.cout + n]);
            }
            tab[((size_t)mh * COORD_CLASSES + cc) * cbs + n] = (float)acc;
          }
        }
    }
  }
  if (bf16) {   // fp32 rows of the bf16-rounded head weights (the fp32 kernels' LDS image: 32 channels per 128-byte row)
    const Layer &H = net.layers.back();
    const float *w = params + H.param_off;
    for (int ks = 0; ks < net.head_f32_ksteps; ++ks)
      for (int n = 0; n < H.cout; ++n) {
        char *row = reinterpret_cast<char *>(packed + net.head_f32_off) + ((size_t)ks * net.head_f32_npad + n) * ROW_BYTES;
        const int swz = (n >> 1) & 7;
        for (int kk = 0; kk < 32; ++kk) {
          const int c = ks * 32 + kk;
          if (c >= H.c0) continue;
          float v = w[(size_t)c * H.cout + n];
          uint32_t u;
          memcpy(&u, &v, 4);
          u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
          memcpy(row + (((kk >> 2) ^ swz) << 4) + (kk & 3) * 4, &u, 4);
        }
      }
  }
  return MSI_OK;
}

// ---- plan ---------------------------------------------------------------------------------------
int msi_net_plan_create(const msi_net_desc *desc, msi_net_plan **out) {
  MSI_REQUIRE(desc && out, "net_plan_create: null pointer");
  *out = nullptr;
  msi_net_plan *pl = new (std::nothrow) msi_net_plan();
  if (!pl) return msi::fail(MSI_E_WORKSPACE, "net_plan_create: out of host memory");
  pl->desc = *desc;
  pl->num_cus = device_cu_count();
  pl->opt[MSI_NET_OPT_FIXUP_KERNEL] = 0;
  pl->opt[MSI_NET_OPT_TAILSPLIT] = 1;   // (2, the residency-aware form, measured 6-10 % slower on every layer it changes: r02_m)
  pl->opt[MSI_NET_OPT_BIGTILE] = 1;
  pl->opt[MSI_NET_OPT_HEAD_FUSE_LN] = 1;
  pl->opt[MSI_NET_OPT_NUM_CUS] = pl->num_cus;
  pl->opt[MSI_NET_OPT_APPLY_AHEAD] = 0;   // measured r02_h: correct and bit-identical, but 2.69 vs 2.56 ms per network
  pl->opt[MSI_NET_OPT_HALO] = 5;   // bits 0 and 2 (bit 1, the fp32 conv-transpose halo kernel: measured slower than the tap kernel + ln_apply, see the kernel)
  pl->opt[MSI_NET_OPT_F32_TILE] = 0;
  pl->opt[MSI_NET_OPT_F32_TILE_MASK] = 0;
  pl->opt[MSI_NET_OPT_UNIFORM_SPLIT] = 0;
  pl->opt[MSI_NET_OPT_SPLIT_OVERHEAD] = 0;
  pl->opt[MSI_NET_OPT_BF16_WAVES] = 8;
  // F32_SPLIT_F16 (the three-product fp16 form) is OPT-IN: measured against fp64 it has the error of a plain fp32 convolution at half the matrix work of the
  // six-product bf16 form (profiles/r04_split_numerics.txt) -- but its operands carry 22 significand bits, not 24, and the round-3 review ruled that a
  // two-way / three-product split must not be the arithmetic a `dtype f32` number is quoted on.  The default stays the six-product form (dropped terms < 2^-26).
  pl->opt[MSI_NET_OPT_F32_SPLIT_F16] = 0;
  pl->opt[MSI_NET_OPT_X3_TILE8] = 0x3ffff;   // (r05: every eligible layer whose grid is >= 3 tiles per CU)
  pl->opt[MSI_NET_OPT_F32_SPLIT3] = 0x3ffff;   // every layer that has the kernel (r04: same error against the oracle as the native path, 1.35-1.45 x faster per layer)
  pl->opt[MSI_NET_OPT_BF16_STAGE_RAW] = 1;   // (bit 1, conv8_1 staging its raw sources: measured 50 us per 16 frames SLOWER -- ~180 VALU per chunk
                                               // against 2 048 matrix cycles of the 128 x 64 tile; bit 0, conv8_2: 130 us faster.  Three interleaved repeats)
  int rc = plan_layers(pl);
  if (rc) { delete pl; return rc; }
  *out = pl;
  return MSI_OK;
}

void msi_net_plan_destroy(msi_net_plan *plan) { delete plan; }

int msi_net_plan_set_option(msi_net_plan *plan, int32_t option, int32_t value) {
  MSI_REQUIRE(plan, "net_plan_set_option: null plan");
  MSI_REQUIRE(option >= 0 && option < MSI_NET_OPT_COUNT, "net_plan_set_option: unknown option %d", option);
  if (option == MSI_NET_OPT_NUM_CUS) {
    MSI_REQUIRE(value >= 8 && value <= 4096, "net_plan_set_option: num_cus %d out of range", value);
    plan->num_cus = value;
  }
  if (option == MSI_NET_OPT_BIGTILE) MSI_REQUIRE(value >= 0 && value <= 2, "net_plan_set_option: bigtile %d", value);
  if (option == MSI_NET_OPT_HALO) MSI_REQUIRE(value >= 0 && value <= 7, "net_plan_set_option: halo %d (bit 0 conv, bit 1 conv-transpose, bit 2 stride-2 conv)", value);
  if (option == MSI_NET_OPT_TAILSPLIT) MSI_REQUIRE(value >= 0 && value <= 2, "net_plan_set_option: tailsplit %d", value);
  if (option == MSI_NET_OPT_F32_TILE) MSI_REQUIRE(value >= 0 && value <= 2, "net_plan_set_option: f32 tile %d", value);
  if (option == MSI_NET_OPT_BF16_WAVES) MSI_REQUIRE(value == 4 || value == 8, "net_plan_set_option: bf16 waves %d (4 or 8)", value);
#ifndef MSI_EXPERIMENTS
  if ((option == MSI_NET_OPT_F32_TILE || option == MSI_NET_OPT_F32_TILE_MASK || option == MSI_NET_OPT_APPLY_AHEAD) && value != 0)
    return msi::fail(MSI_E_UNSUPPORTED, "net_plan_set_option: option %d is an experiment this library was not built with "
                     "(MSI_CNN_DEFINES=-DMSI_EXPERIMENTS python -m matryodshka_amd.build --force)", option);
#endif
  const int old = plan->opt[option];
  plan->opt[option] = value;
  int rc = plan_layers(plan);
  if (rc) {   // keep the plan usable
    plan->opt[option] = old;
    if (option == MSI_NET_OPT_NUM_CUS) plan->num_cus = old;
    plan_layers(plan);
  }
  return rc;
}

size_t msi_net_plan_workspace_bytes(const msi_net_plan *plan) { return plan ? plan->net.ws_bytes : 0; }

int32_t msi_net_plan_layer_is_normalized(const msi_net_plan *plan, int32_t layer) {
  if (!plan || layer < 0 || layer >= MSI_NET_NUM_LAYERS - 1) return -1;
  return plan->launch[layer].skip_apply ? 0 : 1;
}

static int run_layers(const msi_net_plan *plan, const float *packed, const void *net_input, float *pred,
                      void *workspace, size_t workspace_bytes, msi_stream_t stream_, int nlayers);

// The kernel instantiation run_layers launches for `layer` (the if-chain below, restated: keep the two in step), spelled
// as rocprofv3 prints it without the namespace -- so that a parity test can assert WHICH variants a plan at a given batch
// took (the choice depends on batch x tiles vs CUs) and a profile's kernel table can be matched against tested plans.
int32_t msi_net_plan_layer_kernel(const msi_net_plan *plan, int32_t layer, char *name, size_t name_bytes, int32_t *nblocks,
                                  int32_t *nsplit_tiles) {
  MSI_REQUIRE(plan && name && name_bytes > 0, "net_plan_layer_kernel: null pointer");
  MSI_REQUIRE(layer >= 0 && layer < MSI_NET_NUM_LAYERS, "net_plan_layer_kernel: bad layer %d", layer);
  const Layer &L = plan->net.layers[layer];
  const LayerLaunch &Q = plan->launch[layer];
  const int bf16 = plan->desc.dtype == MSI_DTYPE_BF16;
  const char *mode = L.kind == MODE_CONV ? "0" : (L.kind == MODE_CONVT ? "1" : "2");
  if (Q.halo_tb) {
    snprintf(name, name_bytes, "convt_halo_bf16_kernel<128, %d, %d>", Q.hbn, Q.p.halo_apply ? 1 : 0);
  } else if (Q.halo && bf16) {
    if (Q.halo_s2) snprintf(name, name_bytes, "conv_halo_bf16_s2_kernel<%d, 4>", Q.halo_apply ? 1 : 0);
    else if (Q.hbm == 128) snprintf(name, name_bytes, "conv_halo_bf16_kernel<128, 128, %d, %d, %d>", L.rate, Q.halo_apply ? 1 : 0,
                                    plan->opt[MSI_NET_OPT_BF16_WAVES] == 8 ? 8 : 4);
    else snprintf(name, name_bytes, "conv_halo_bf16_kernel<256, 64, 1, %d, 4>", Q.halo_apply ? 1 : 0);
  } else if (Q.halo_t) {
    if (Q.halo_x3) snprintf(name, name_bytes, "convt_halo_x3_kernel<%d>", Q.halo_x2 ? 2 : 3);
    else snprintf(name, name_bytes, "convt_halo_kernel");
  } else if (Q.halo) {
    if (Q.halo_s2 && Q.halo_x3) snprintf(name, name_bytes, "conv_halo_s2_x3_kernel<%d, %d>", Q.halo_apply ? 1 : 0, Q.halo_x2 ? 2 : 3);
    else if (Q.halo_s2) snprintf(name, name_bytes, "conv_halo_s2_kernel<%d>", Q.halo_apply ? 1 : 0);
    else if (Q.x3_th8) snprintf(name, name_bytes, "conv_halo8_x3_kernel<%d, 3>", Q.halo_apply ? 1 : 0);
    else if (Q.halo_x3) snprintf(name, name_bytes, "conv_halo_x3_kernel<%d, %d, %d>", L.rate, Q.halo_apply ? 1 : 0, Q.halo_x2 ? 2 : 3);
    else snprintf(name, name_bytes, "conv_halo_kernel<%d, %d>", L.rate, Q.halo_apply ? 1 : 0);
  } else {
    const int bm = Q.tile == TILE_128x128 || Q.tile == TILE_128x64 ? 128 : 64;
    const int bn = Q.tile == TILE_128x128 || Q.tile == TILE_64x128 ? 128 : 64;
    snprintf(name, name_bytes, "conv_igemm_kernel<%d, %d, %s, %d>", bm, bn, mode, bf16);
  }
  if (nblocks) *nblocks = Q.nblocks + Q.p.n_apply;
  if (nsplit_tiles) *nsplit_tiles = Q.nfix;
  return MSI_OK;
}

#ifdef MSI_DEBUG_SUMS   // (debug builds only: byte offset of a layer's LayerNorm sums in the workspace)
extern "C" long long msi_debug_sums_offset(const msi_net_plan *plan, int layer) { return (long long)plan->net.layers[layer].sums_off; }
extern "C" long long msi_debug_partial_offset(const msi_net_plan *plan) { return (long long)plan->net.partial_off; }
#endif
int32_t msi_net_plan_status(const msi_net_plan *plan, const void *workspace, msi_stream_t stream_, int32_t *status_bits) {
  MSI_REQUIRE(plan && workspace, "net_plan_status: null pointer");
  hipStream_t stream = msi::as_stream(stream_);
  int word = 0;
  hipError_t e = hipMemcpyAsync(&word, static_cast<const char *>(workspace) + plan->net.err_off, sizeof(int), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_plan_status: %s", hipGetErrorString(e));
  if (status_bits) *status_bits = word;
  if (word == 0) return MSI_OK;
  return msi::fail(MSI_E_RANGE, "net_plan_status: 0x%x:%s%s%s%s", word,
                   (word & STATUS_F16_SPLIT_RANGE) ? " an operand of a layer on the fp16 split exceeded the fp16 range (|x| > 65504): set MSI_NET_OPT_F32_SPLIT_F16 = 0" : "",
                   (word & STATUS_LN_OVERFLOW) ? " a LayerNorm sum left its fixed-point window (raw convolution output far above the scale the weights predict: non-finite or mis-scaled input?)" : "",
                   (word & STATUS_LN_UNDERFLOW) ? " a LayerNorm variance is below the resolution of its fixed-point window (raw convolution output far below the scale the weights predict, or constant)" : "",
                   (word & STATUS_APPLY_AHEAD_TIMEOUT) ? " an apply-ahead wait timed out" : "");
}

// LayerNorm window calibration.  The fixed-point window of a layer's sums follows an exponent the packer ESTIMATES from the weights
// (sqrt(K) rms(w) rms(input)); trained gamma / beta / weights may put a layer's real raw output 2^+-12 away from that estimate, and then every forward ends in
// MSI_E_RANGE.  This entry point MEASURES instead: layer by layer (a layer's input is only right once its producers' windows are), run layers 0 .. L, read L's sums
// and the status word, move the exponent by 12 while the sums overflow / underflow, then centre it on the measured rms (log-mid of the smallest and largest sample),
// write {S1, S2, 1 / S1, 1 / S2} into the packed blob ON THE DEVICE and run 0 .. L once more so that L's consumers see sums in the new unit.  Synchronous (it reads
// the sums back: a one-off, ~40 forwards' worth of kernels), allocates nothing, leaves the workspace as a forward of layers 0 .. 16 would.
int32_t msi_net_plan_calibrate(const msi_net_plan *plan, float *packed, const void *net_input, void *workspace, size_t workspace_bytes,
                               msi_stream_t stream_, int32_t *layers_changed) {
  MSI_REQUIRE(plan && packed && net_input && workspace, "net_plan_calibrate: null pointer");
  const Net &net = plan->net;
  const msi_net_desc *desc = &plan->desc;
  hipStream_t stream = msi::as_stream(stream_);
  if (layers_changed) *layers_changed = 0;
  if (desc->batch == 0) return MSI_OK;
  const size_t nwords = (size_t)desc->batch * LN_SHARDS * LN_WORDS;
  std::vector<long long> sums(nwords);
  char *ws = static_cast<char *>(workspace);
  for (int li = 0; li < MSI_NET_NUM_LAYERS - 1; ++li) {
    const Layer &L = net.layers[li];
    double scl[LN_SCL_DOUBLES];
    hipError_t he = hipMemcpyAsync(scl, packed + L.lnscl_off, sizeof(scl), hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (he != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he));
    int e0 = LN_S1_BITS - (int)lrint(log2(scl[0])), e = e0;
    bool settled = false;
    for (int it = 0; it < 24 && !settled; ++it) {
      const double ns[LN_SCL_DOUBLES] = {ldexp(1.0, LN_S1_BITS - e), ldexp(1.0, LN_S2_BITS - 2 * e), ldexp(1.0, -(LN_S1_BITS - e)), ldexp(1.0, -(LN_S2_BITS - 2 * e))};
      he = hipMemcpyAsync(packed + L.lnscl_off, ns, sizeof(ns), hipMemcpyHostToDevice, stream);
      if (he == hipSuccess) he = hipStreamSynchronize(stream);   // (ns is on this stack frame)
      if (he != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he));
      int rc = run_layers(plan, packed, net_input, nullptr, workspace, workspace_bytes, stream_, li + 1);
      if (rc) return rc;
      int word = 0;
      he = hipMemcpyAsync(sums.data(), ws + L.sums_off, nwords * sizeof(long long), hipMemcpyDeviceToHost, stream);
      if (he == hipSuccess) he = hipMemcpyAsync(&word, ws + net.err_off, sizeof(int), hipMemcpyDeviceToHost, stream);
      if (he == hipSuccess) he = hipStreamSynchronize(stream);
      if (he != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he));
      if (word & STATUS_LN_OVERFLOW) { e += 12; if (e > 120) break; continue; }   // a share left the window (or the data is not finite: the loop gives up at 2^120)
      // per sample: sum x^2 in units of 1 / S2; below ~1e6 sqrt(waves) units the variance is resolved to < 6 digits (ln_mean_inv's rule, restated on the host)
      double rmin = 1e300, rmax = 0.0;
      bool under = false;
      for (int b = 0; b < desc->batch; ++b) {
        double h1 = 0.0, h2 = 0.0;
        for (int sh = 0; sh < LN_SHARDS; ++sh) {
          h1 += (double)sums[((size_t)b * LN_SHARDS + sh) * LN_WORDS];
          h2 += (double)sums[((size_t)b * LN_SHARDS + sh) * LN_WORDS + 1];
        }
        if (h2 * h2 < LN_UNDERFLOW_UNITS_SQ * (L.ln_count / 1024.0 + 1.0)) under = true;
        const double mu = h1 * ns[2] / L.ln_count;
        double ms = h2 * ns[3] / L.ln_count;               // E[x^2]: the window has to hold the raw values themselves
        (void)mu;
        const double r = sqrt(ms > 0.0 ? ms : 0.0);
        rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
      }
      if (under && rmax == 0.0) { e -= 12; if (e < -120) break; continue; }        // nothing resolved at all: look lower
      if (rmax > 0.0) {
        const int ec = (int)lrint(0.5 * (log2(rmax) + log2(rmin > 0.0 ? rmin : rmax)));
        if (ec > e + 1 || ec < e - 1) { e = ec; continue; }                         // centre the window (to within an octave) and measure once more in the new unit
      }
      if (under) { e -= 12; if (e < -120) break; continue; }
      settled = true;
    }
    if (!settled)
      return msi::fail(MSI_E_RANGE, "net_plan_calibrate: layer %s has no finite, non-constant raw output to centre a LayerNorm window on (non-finite input or weights?)", L.name);
    if (e != e0 && layers_changed) ++*layers_changed;
  }
  return MSI_OK;
}

int msi_net_plan_forward(const msi_net_plan *plan, const float *packed, const void *net_input, float *pred,
                         void *workspace, size_t workspace_bytes, msi_stream_t stream_) {
  MSI_REQUIRE(pred, "net_forward: null pointer");
  return run_layers(plan, packed, net_input, pred, workspace, workspace_bytes, stream_, MSI_NET_NUM_LAYERS);
}

int msi_net_plan_forward_rgba(const msi_net_plan *plan, const float *packed, const void *net_input, float *rgba_native,
                              float *blend_weights, float *alphas, float *pred, void *workspace, size_t workspace_bytes,
                              msi_stream_t stream_, void *event_after_convs) {
  MSI_REQUIRE(plan, "net_forward_rgba: null plan");
  const msi_net_desc *desc = &plan->desc;
  const Net &net = plan->net;
  const Layer &H = net.layers[MSI_NET_NUM_LAYERS - 1];
  const int nd = desc->num_outputs / 2;
  const int bf16 = desc->dtype == MSI_DTYPE_BF16;
  if ((!bf16 && !plan->launch[MSI_NET_NUM_LAYERS - 1].fuse_ln) || (bf16 && !plan->opt[MSI_NET_OPT_HEAD_FUSE_LN]) ||
      H.c0 > 64 || H.c0 % 4 != 0 || desc->num_outputs != 2 * nd || nd % 4 != 0 || nd > 64 || desc->in_channels != 6 * nd ||
      ((long)desc->height * desc->width) % HA_TP != 0)
    return msi::fail(MSI_E_UNSUPPORTED, "net_forward_rgba: fused tail needs a blend_psv network (in = 6 D, out = 2 D, "
                     "D %% 4 == 0, D <= 64, ngf <= 64, HEAD_FUSE_LN on)");
  MSI_REQUIRE(rgba_native, "net_forward_rgba: null pointer");
  const int ng = (nd + HA_LG - 1) / HA_LG;   // layer groups (grid.y of the fused tail): D = 64 -> 2 x 32 layers
  if (nd % ng != 0 || (nd / ng) % 4 != 0 || (ng > 1 && bf16 && (nd / ng) % 8 != 0))
    return msi::fail(MSI_E_UNSUPPORTED, "net_forward_rgba: D = %d does not split into layer groups of a multiple of %d", nd, bf16 ? 8 : 4);
  // (bf16: the head's source stays raw fp32 -- no ln_apply launch, no bf16 copy: this kernel normalises and rounds it)
  int rc = run_layers(plan, packed, net_input, nullptr, workspace, workspace_bytes, stream_, MSI_NET_NUM_LAYERS - 1);
  if (rc || desc->batch == 0) return rc;
  hipStream_t stream = msi::as_stream(stream_);
  if (event_after_convs) {
    hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event_after_convs), stream);
    if (e != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_forward_rgba: %s", hipGetErrorString(e));
  }
  char *ws = static_cast<char *>(workspace);
  const Layer &S = net.layers[H.src0];
  HeadAsmParams q;
  q.x = reinterpret_cast<const float *>(ws + S.raw_off);
  q.wpk = packed + H.packed_off;   // (bf16 plans: the packed bf16 rows themselves -- the fp32-format copy at head_f32_off is unused since r04)
  q.bias = packed + H.gamma_off;
  float *aff = reinterpret_cast<float *>(ws + S.aff_off);
  hipLaunchKernelGGL(ln_finish_kernel, dim3(desc->batch), dim3(256), 0, stream,
                     reinterpret_cast<const long long *>(ws + S.sums_off), 1.0 / S.ln_count,
                     reinterpret_cast<const double *>(packed + S.lnscl_off), reinterpret_cast<int *>(ws + net.err_off),
                     packed + S.gamma_off, packed + S.beta_off, S.cout, aff, bf16);
  rc = msi::check_launch("ln_finish");
  if (rc) return rc;
  q.aff = aff;
  q.psv = net_input;
  q.rgba = reinterpret_cast<float4 *>(rgba_native);
  q.bw_out = blend_weights;
  q.al_out = alphas;
  q.pred_out = pred;
  q.C0 = H.c0; q.ksteps = H.ksteps; q.npad = H.npad;   // (bf16: one k-step of 64 channels)
  q.nd = nd; q.hw = desc->height * desc->width;
  q.npix_total = (long)desc->batch * q.hw;
  q.lg = nd / ng;
  q.ng = ng;
  {
    auto magic = [](unsigned d) { return d == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / d); };
    const int vec = bf16 ? 8 : 4;                                  // elements per 16-byte vector of the sweep volume
    const unsigned vpp = (unsigned)((ng == 1 ? 6 * nd : 2 * 3 * q.lg) / vec);
    q.mg_vpp = magic(vpp);
    q.mg_nchunk = magic((unsigned)(q.ksteps * 8));
    q.mg_hw = magic((unsigned)q.hw);
    if (q.npix_total >= (1L << 32)) return msi::fail(MSI_E_UNSUPPORTED, "net_forward_rgba: more than 2^32 pixels per batch");
  }
  constexpr int BN = 64;
  size_t r_bytes = (size_t)q.ksteps * (HA_TP + BN) * ROW_BYTES;
  if (r_bytes < (size_t)HA_TP * (6 * q.lg + 1) * sizeof(float)) r_bytes = (size_t)HA_TP * (6 * q.lg + 1) * sizeof(float);
  if (bf16) {   // no weight tile in LDS, packed-bf16 sweep tile (see the kernel)
    r_bytes = (size_t)q.ksteps * HA_TP * ROW_BYTES;
    if (r_bytes < (size_t)HA_TP * (3 * q.lg + 1) * sizeof(unsigned)) r_bytes = (size_t)HA_TP * (3 * q.lg + 1) * sizeof(unsigned);
  }
  const size_t lds = 2 * 64 * 4 + 64 + ((r_bytes + 15) & ~(size_t)15) + (size_t)HA_TP * (2 * q.lg + 1) * sizeof(float);
  const long ntile = q.npix_total / HA_TP;
  if (((ntile + 7) / 8) * 8 * ng >= (1L << 31)) return msi::fail(MSI_E_UNSUPPORTED, "net_forward_rgba: too many pixel tiles for one launch");
  const dim3 grid((unsigned)(((ntile + 7) / 8) * 8 * ng));     // (see the kernel: XCD x takes tiles x, x + 8, ...; the layer groups of a tile are neighbours there)
  if (bf16) hipLaunchKernelGGL(head_assemble_kernel<1>, grid, dim3(256), lds, stream, q);
  else hipLaunchKernelGGL(head_assemble_kernel<0>, grid, dim3(256), lds, stream, q);
  return msi::check_launch("head_assemble");
}

static int run_layers(const msi_net_plan *plan, const float *packed, const void *net_input, float *pred,
                      void *workspace, size_t workspace_bytes, msi_stream_t stream_, int nlayers) {
  MSI_REQUIRE(plan, "net_forward: null plan");
  const msi_net_desc *desc = &plan->desc;
  const Net &net = plan->net;
  const int bf16 = desc->dtype == MSI_DTYPE_BF16;
  MSI_REQUIRE(packed && net_input && workspace, "net_forward: null pointer");
  if (workspace_bytes < net.ws_bytes)
    return msi::fail(MSI_E_WORKSPACE, "net_forward: workspace %zu B < required %zu B", workspace_bytes,
                     net.ws_bytes);
  if (desc->batch == 0) return MSI_OK;
  hipStream_t stream = msi::as_stream(stream_);
  char *ws = static_cast<char *>(workspace);
  // tickets of the in-launch fix-ups and the LayerNorm sums start from zero
  // (a kernel of the library's own instead of hipMemsetAsync: the runtime's fill is a blit with its own barrier packets)
  {
    const size_t n16 = net.zero_bytes / 16;               // zero_off and zero_bytes are multiples of 256
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<float4 *>(ws + net.zero_off), n16);
    int rc0 = msi::check_launch("zero");
    if (rc0) return rc0;
  }
  int *cnt = reinterpret_cast<int *>(ws + net.cnt_off);
  for (int li = 0; li < nlayers; ++li) {
    const Layer &L = net.layers[li];
    const LayerLaunch &Q = plan->launch[li];
    ConvParams p = Q.p;
    // sources: the network input, or the normalised output of the producer -- in place in its raw
    // buffer (fp32) or the bf16 copy ln_apply wrote next to it
    auto src_ptr = [&](int s) -> const char * {
      if (s < 0) return static_cast<const char *>(net_input);
      return ws + (bf16 ? net.layers[s].act_off : net.layers[s].raw_off);
    };
    p.x0 = src_ptr(L.src0);
    p.x1 = L.src1 >= 0 ? src_ptr(L.src1) : p.x0;  // unused second source: alias the first (cpt1 = 0 keeps it unselected)
    p.wpk = reinterpret_cast<const char *>(packed + L.packed_off);
    p.coord_bias = L.has_coord ? packed + L.coord_off : nullptr;
    p.bias = L.kind == MODE_HEAD ? packed + L.gamma_off : nullptr;
    if (Q.fuse_ln) {   // fp32 head: its producer's LayerNorm + ReLU is applied while loading (the producer's buffer holds the raw output)
      const Layer &S = net.layers[L.src0];
      p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
      p.ln_gamma = packed + S.gamma_off;
      p.ln_beta = packed + S.beta_off;
    }
    p.y = L.kind == MODE_HEAD ? pred : reinterpret_cast<float *>(ws + L.raw_off);
    p.sums = L.kind == MODE_HEAD ? nullptr : reinterpret_cast<long long *>(ws + L.sums_off);
    auto scl_of = [&](int li2) { return reinterpret_cast<const double *>(packed + net.layers[li2].lnscl_off); };
    p.ln_scl = L.kind == MODE_HEAD ? nullptr : scl_of(li);
    p.ln_scl_src = L.src0 >= 0 ? scl_of(L.src0) : nullptr;
    p.ln_scl_src1 = L.src1 >= 0 ? scl_of(L.src1) : nullptr;
    p.status = reinterpret_cast<int *>(ws + net.err_off);
    p.partial = reinterpret_cast<float *>(ws + net.partial_off);
    p.tile_cnt = Q.inlaunch ? cnt + (size_t)li * CONV_SLOTS_PER_CU * plan->num_cus : nullptr;
    if (p.n_apply > 0) {
      const Layer &S = net.layers[L.src0];
      p.ap_x = reinterpret_cast<float *>(ws + S.raw_off);
      p.ap_yb = bf16 ? reinterpret_cast<unsigned short *>(ws + S.act_off) : nullptr;
      p.ap_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
      p.ap_gamma = packed + S.gamma_off;
      p.ap_beta = packed + S.beta_off;
      p.ap_aff = reinterpret_cast<float *>(ws + S.aff_off);
      p.ap_flags = reinterpret_cast<int *>(ws + S.flags_off);
      p.ap_err = reinterpret_cast<int *>(ws + net.err_off);
    }
#if defined(MSI_CONV_TIMING) || defined(MSI_DEBUG_STATS)
    p.dbg = (li == g_timing_layer) ? g_timing_buf : nullptr;
#endif
    int rc;
    if (Q.halo_tb) {
      if (p.halo_apply & 1) {   // source 0 / 1 are read RAW (fp16), their LayerNorm applied while staging
        const Layer &S = net.layers[L.src0];
        p.x0 = ws + S.raw_off;
        p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma = packed + S.gamma_off;
        p.ln_beta = packed + S.beta_off;
      }
      if (p.halo_apply & 2) {
        const Layer &S = net.layers[L.src1];
        p.x1 = ws + S.raw_off;
        p.ln_sums1 = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma1 = packed + S.gamma_off;
        p.ln_beta1 = packed + S.beta_off;
      }
      // (the 128 x 128 tile with APPLY needs more than the 256 registers of two waves per SIMD -- 120 bytes of scratch inside
      // the chunk loop -- and is not built: the plan only marks sources of the 128 x 64 tile as raw)
      if (p.halo_apply) rc = Q.hbn == 128 ? msi::fail(MSI_E_UNSUPPORTED, "convt_halo_bf16: APPLY is built for the 128x64 tile")
                                          : launch_convt_halo_bf16<128, 64, 1>(Q, p, stream);
      else rc = Q.hbn == 128 ? launch_convt_halo_bf16<128, 128, 0>(Q, p, stream) : launch_convt_halo_bf16<128, 64, 0>(Q, p, stream);
    } else if (Q.halo && bf16) {
      if (Q.halo_apply) {   // the patch comes from the producer's RAW fp32 output
        const Layer &S = net.layers[L.src0];
        p.x0 = ws + S.raw_off;
        p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma = packed + S.gamma_off;
        p.ln_beta = packed + S.beta_off;
      }
      const bool w8 = plan->opt[MSI_NET_OPT_BF16_WAVES] == 8;
      if (Q.halo_s2) {   // (four waves: with eight the staging path does not fit 128 registers -- 48 bytes of scratch -- and was measured
                         // 0.8 % of the network slower, three interleaved repeats)
        rc = Q.halo_apply ? launch_halo_bf16_s2<1, 4>(Q, p, stream) : launch_halo_bf16_s2<0, 4>(Q, p, stream);
      } else
#define MSI_HB(BM_, BN_, R_, A_) (w8 ? launch_halo_bf16<BM_, BN_, R_, A_, 8>(Q, p, stream) : launch_halo_bf16<BM_, BN_, R_, A_, 4>(Q, p, stream))
      if (Q.hbm == 128) {
        if (L.rate == 1) rc = Q.halo_apply ? MSI_HB(128, 128, 1, 1) : MSI_HB(128, 128, 1, 0);
        else rc = Q.halo_apply ? MSI_HB(128, 128, 2, 1) : MSI_HB(128, 128, 2, 0);
      } else {   // (the 256 x 64 tile stays at four waves: eight do not fit their 128 registers -- 44-64 bytes of scratch -- and were
                 // measured slower, conv8_2 27.8 k -> 32.9 k cycles per workgroup)
        rc = Q.halo_apply ? launch_halo_bf16<256, 64, 1, 1, 4>(Q, p, stream) : launch_halo_bf16<256, 64, 1, 0, 4>(Q, p, stream);
      }
#undef MSI_HB
    } else if (Q.halo_t) {
      if (p.halo_apply & 1) {
        const Layer &S = net.layers[L.src0];
        p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma = packed + S.gamma_off;
        p.ln_beta = packed + S.beta_off;
      }
      if (p.halo_apply & 2) {
        const Layer &S = net.layers[L.src1];
        p.ln_sums1 = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma1 = packed + S.gamma_off;
        p.ln_beta1 = packed + S.beta_off;
      }
      if (Q.halo_x3) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + (Q.halo_x2 ? L.x2_off : L.x3_off));
        constexpr int lds_ct3 = HaloGeomX3<1, MSI_CT3_NSTG, 3>::LDS_BYTES, lds_ct2 = HaloGeomX3<1, 3, 2>::LDS_BYTES;
        if (Q.halo_x2) hipLaunchKernelGGL(convt_halo_x3_kernel<2>, dim3(Q.nblocks), dim3(256), lds_ct2, stream, p);
        else hipLaunchKernelGGL(convt_halo_x3_kernel<3>, dim3(Q.nblocks), dim3(256), lds_ct3, stream, p);
      } else {
        hipLaunchKernelGGL(convt_halo_kernel, dim3(Q.nblocks), dim3(256), ConvtHaloGeom::LDS_BYTES, stream, p);
      }
      rc = msi::check_launch("convt_halo");
      if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) {
        hipLaunchKernelGGL((conv_fixup_kernel<64, 64, MODE_CONVT>), dim3(Q.nfix, 2), dim3(256), 0, stream, p);
        rc = msi::check_launch("conv_fixup");
      }
    } else if (Q.halo) {
      if (Q.halo_apply) {
        const Layer &S = net.layers[L.src0];
        p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off);
        p.ln_gamma = packed + S.gamma_off;
        p.ln_beta = packed + S.beta_off;
      }
      const dim3 grid(Q.nblocks), block(256);
      if (Q.halo_s2 && Q.halo_x3) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + (Q.halo_x2 ? L.x2_off : L.x3_off));
        if (Q.halo_x2) {
          if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_x3_kernel<1, 2>), grid, block, HaloGeomS2X3<2>::LDS_BYTES, stream, p);
          else hipLaunchKernelGGL((conv_halo_s2_x3_kernel<0, 2>), grid, block, HaloGeomS2X3<2>::LDS_BYTES, stream, p);
        } else {
          if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_x3_kernel<1, 3>), grid, block, HaloGeomS2X3<3>::LDS_BYTES, stream, p);
          else hipLaunchKernelGGL((conv_halo_s2_x3_kernel<0, 3>), grid, block, HaloGeomS2X3<3>::LDS_BYTES, stream, p);
        }
      } else if (Q.halo_s2) {
        if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_kernel<1>), grid, block, HaloGeomS2::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv_halo_s2_kernel<0>), grid, block, HaloGeomS2::LDS_BYTES, stream, p);
      } else if (Q.x3_th8) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + L.x3_off);
        typedef HaloGeomX3<1, (MSI_X3_NSTG ? MSI_X3_NSTG : 2), 3, 8> G8_;
        static_assert(G8_::LDS_BYTES <= 65536, "conv_halo8_x3_kernel: LDS without the launch attribute");
        if (Q.halo_apply) hipLaunchKernelGGL((conv_halo8_x3_kernel<1, 3>), grid, block, G8_::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv_halo8_x3_kernel<0, 3>), grid, block, G8_::LDS_BYTES, stream, p);
      } else if (Q.halo_x3) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + (Q.halo_x2 ? L.x2_off : L.x3_off));
        static thread_local unsigned long long done2[8] = {0};       // (above 64 KB of LDS the launch needs the attribute)
#define MSI_X3_LAUNCH(R, A, N)                                                                                         \
  {                                                                                                                    \
    typedef HaloGeomX3<R, (N == 2 ? MSI_X2_NSTG : (MSI_X3_NSTG ? MSI_X3_NSTG : (R == 1 ? 2 : 3))), N> G_;              \
    if (G_::LDS_BYTES > 65536) {                                                                                       \
      int rc0 = set_max_lds(reinterpret_cast<const void *>(conv_halo_x3_kernel<R, A, N>), G_::LDS_BYTES, done2[(R - 1) * 4 + A * 2 + (N - 2)], "conv_halo_x3"); \
      if (rc0) return rc0;                                                                                             \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_halo_x3_kernel<R, A, N>), grid, block, G_::LDS_BYTES, stream, p);                          \
  }
        const int sel = (L.rate == 1 ? 0 : 4) + (Q.halo_apply ? 2 : 0) + (Q.halo_x2 ? 0 : 1);
        switch (sel) {
          case 0: MSI_X3_LAUNCH(1, 0, 2) break;
          case 1: MSI_X3_LAUNCH(1, 0, 3) break;
          case 2: MSI_X3_LAUNCH(1, 1, 2) break;
          case 3: MSI_X3_LAUNCH(1, 1, 3) break;
          case 4: MSI_X3_LAUNCH(2, 0, 2) break;
          case 5: MSI_X3_LAUNCH(2, 0, 3) break;
          case 6: MSI_X3_LAUNCH(2, 1, 2) break;
          default: MSI_X3_LAUNCH(2, 1, 3) break;
        }
#undef MSI_X3_LAUNCH
      } else if (L.rate == 1) {
        if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_kernel<1, 1>), grid, block, HaloGeom<1>::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv_halo_kernel<1, 0>), grid, block, HaloGeom<1>::LDS_BYTES, stream, p);
      } else {
        if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_kernel<2, 1>), grid, block, HaloGeom<2>::LDS_BYTES, stream, p);
        else hipLaunchKernelGGL((conv_halo_kernel<2, 0>), grid, block, HaloGeom<2>::LDS_BYTES, stream, p);
      }
      rc = msi::check_launch("conv_halo");
      if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) {
        if (Q.x3_th8) hipLaunchKernelGGL((conv_fixup_kernel<128, 64, MODE_CONV>), dim3(Q.nfix), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_fixup_kernel<64, 64, MODE_CONV>), dim3(Q.nfix), dim3(256), 0, stream, p);
        rc = msi::check_launch("conv_fixup");
      }
    } else
    switch (Q.tile) {
      case TILE_128x128: rc = launch_conv<128, 128>(Q, p, bf16, stream); break;
      case TILE_128x64: rc = launch_conv<128, 64>(Q, p, bf16, stream); break;
#ifdef MSI_EXPERIMENTS
      case TILE_64x128: rc = bf16 ? msi::fail(MSI_E_UNSUPPORTED, "conv: 64x128 is an fp32 tile") : launch_conv<64, 128>(Q, p, 0, stream); break;
#else
      case TILE_64x128: rc = msi::fail(MSI_E_UNSUPPORTED, "conv: the 64x128 fp32 tile is an experiment (MSI_EXPERIMENTS)"); break;
#endif
      default: rc = launch_conv<64, 64>(Q, p, bf16, stream); break;
    }
    if (rc) return rc;
    // (bf16 fused tail: head_assemble_kernel normalises + rounds the head's source itself)
    const bool tail_src = bf16 && nlayers == MSI_NET_NUM_LAYERS - 1 && li == net.layers[MSI_NET_NUM_LAYERS - 1].src0;
    if (L.kind != MODE_HEAD && !Q.skip_apply && !tail_src) {
      const size_t per_sample = (size_t)L.out_h * L.out_w * L.cout;
      float *raw = reinterpret_cast<float *>(ws + L.raw_off), *aff = reinterpret_cast<float *>(ws + L.aff_off);
      const dim3 grid(Q.ln_blocks, desc->batch);
      const size_t lds = (size_t)2 * L.cout * sizeof(float);
      const long long *sums = reinterpret_cast<const long long *>(ws + L.sums_off);
      if (bf16)
        hipLaunchKernelGGL(ln_apply_kernel<1>, grid, dim3(256), lds, stream, raw, sums, 1.0 / L.ln_count, scl_of(li), p.status,
                           packed + L.gamma_off, packed + L.beta_off, per_sample, L.cout, aff,
                           reinterpret_cast<unsigned short *>(ws + L.act_off));
      else
        hipLaunchKernelGGL(ln_apply_kernel<0>, grid, dim3(256), lds, stream, raw, sums, 1.0 / L.ln_count, scl_of(li), p.status,
                           packed + L.gamma_off, packed + L.beta_off, per_sample, L.cout, aff,
                           static_cast<unsigned short *>(nullptr));
      rc = msi::check_launch("ln_apply");
      if (rc) return rc;
    }
  }
  return MSI_OK;
}

// ---- descriptor-level convenience (a transient plan per call; the frame loop uses a plan) -----------
size_t msi_net_workspace_bytes(const msi_net_desc *desc) {
  msi_net_plan *pl = nullptr;
  if (msi_net_plan_create(desc, &pl)) return 0;
  const size_t n = pl->net.ws_bytes;
  msi_net_plan_destroy(pl);
  return n;
}

static int forward_once(const msi_net_desc *desc, const float *packed, const void *net_input, float *pred,
                        void *workspace, size_t workspace_bytes, msi_stream_t stream) {
  msi_net_plan *pl = nullptr;
  int rc = msi_net_plan_create(desc, &pl);
  if (rc) return rc;
  rc = msi_net_plan_forward(pl, packed, net_input, pred, workspace, workspace_bytes, stream);
  msi_net_plan_destroy(pl);
  return rc;
}

int msi_net_forward_f32(const msi_net_desc *desc, const float *packed, const float *net_input,
                        float *pred, void *workspace, size_t workspace_bytes, msi_stream_t stream) {
  MSI_REQUIRE(desc && desc->dtype == MSI_DTYPE_F32, "net_forward_f32: desc->dtype must be MSI_DTYPE_F32");
  return forward_once(desc, packed, net_input, pred, workspace, workspace_bytes, stream);
}

int msi_net_forward_bf16(const msi_net_desc *desc, const float *packed, const void *net_input_bf16,
                         float *pred, void *workspace, size_t workspace_bytes, msi_stream_t stream) {
  MSI_REQUIRE(desc && desc->dtype == MSI_DTYPE_BF16, "net_forward_bf16: desc->dtype must be MSI_DTYPE_BF16");
  return forward_once(desc, packed, net_input_bf16, pred, workspace, workspace_bytes, stream);
}

}  // extern "C"
