// Shared by the translation units of the K2 convolution kernels (r05: cnn.hip used to be ONE 400 KB translation unit -- VERDICT r04 item 8):
//   cnn.hip        host side: layer table, packing, plan, forward, C ABI
//   cnn_igemm.hip  conv_igemm_kernel (tap-DMA implicit GEMM, fp32 / bf16), conv_fixup_kernel
//   cnn_halo.hip   the native fp32 halo-patch kernels (stride 1 / 2, conv-transpose)
//   cnn_x3.hip     fp32 through the six-product bf16 / three-product fp16 split (stride 1 incl. the 8-row tile, stride 2, conv-transpose)
//   cnn_bf16.hip   the bf16 halo-patch kernels (stride 1 / 2, conv-transpose)
//   cnn_tail.hip   fused tail (head + RGBA assembly), LayerNorm finish / apply, zero
// Here: constants, ConvParams, the per-layer launch record (namespace msi_cnn: shared TYPES need one identity across translation units), the device helpers
// of the k-loops and the epilogue (anonymous namespace: every unit inlines its own), and the launch entry points each family exports.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "msi_common.h"

namespace msi_cnn {


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
  int row_par;               // 1 (rate-2 layers of the split kernels, r05): a halo tile's rows are every OTHER image row -- tile row index t = 2 t' + parity covers rows
                             // 2 TH t' + parity + 2 r: a dilation-2 convolution restricted to one row parity is a dilation-1 convolution along H (halo_row)
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


constexpr int HA_TP = 32;   // pixels per workgroup
#ifndef MSI_HA_LG
#define MSI_HA_LG 32
#endif
constexpr int HA_LG = MSI_HA_LG;   // at most this many layers per workgroup: D = 64 runs as two layer groups (grid.y)

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

// ---- launch entry points of the kernel families (each returns an MSI_* code; p carries every pointer) ----
int launch_igemm(const LayerLaunch &Q, const ConvParams &p, int bf16, hipStream_t stream);                        // cnn_igemm.hip: Q.tile, p.mode; its split tiles' fix-up launch included
int launch_fixup(int bm, int bn, int mode, unsigned grid_x, unsigned grid_y, const ConvParams &p, hipStream_t stream);   // cnn_igemm.hip: conv_fixup_kernel<bm, 64, mode> for the halo families
int launch_halo_f32(const LayerLaunch &Q, const ConvParams &p, int rate, hipStream_t stream);                     // cnn_halo.hip: convt_halo_kernel / conv_halo_s2_kernel / conv_halo_kernel
int launch_x3(const LayerLaunch &Q, const ConvParams &p, int rate, hipStream_t stream);                           // cnn_x3.hip: every split kernel (Q.halo_x3)
int launch_bf16_halo(const LayerLaunch &Q, const ConvParams &p, int rate, bool eight_waves, hipStream_t stream);  // cnn_bf16.hip: Q.halo_tb, or Q.halo of a bf16 plan
int launch_zero(void *p, size_t n16, hipStream_t stream);                                                         // cnn_tail.hip
int launch_ln_finish(int batch, hipStream_t stream, const long long *sums, double inv_n, const double *scl, int *status, const float *gamma, const float *beta, int C,
                     float *aff, int raw16);
int launch_ln_apply(int bf16out, unsigned blocks, int batch, size_t lds, hipStream_t stream, float *x, const long long *sums, double inv_n, const double *scl, int *status,
                    const float *gamma, const float *beta, size_t per_sample, int C, float *aff, unsigned short *yb);
int launch_head_assemble(int bf16in, unsigned grid_x, size_t lds, hipStream_t stream, const HeadAsmParams &q);
int debug_conv_occupancy(int lds_bytes);                                                                          // cnn_igemm.hip (tools/conv_timing.py)

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

// a compile-time integer as a lambda argument: the k-steps of the kernels are generic lambdas called with their tap / step index (r05: they used to be macros)
template <int N> using IC = std::integral_constant<int, N>;
}  // namespace msi_cnn
using namespace msi_cnn;

namespace {

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

// image row of row r of halo tile row tyi (rows_per_tile rows per tile): consecutive rows, or (ConvParams::row_par) every other row of one parity
__device__ __forceinline__ int halo_row(const ConvParams &p, int tyi, int r, int rows_per_tile) {
  return p.row_par ? (tyi >> 1) * (2 * rows_per_tile) + (tyi & 1) + 2 * r : tyi * rows_per_tile + r;
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
#ifndef MSI_EPI_NT   // experiment (r06): 1 = non-temporal stores of the staged epilogue's raw outputs
#define MSI_EPI_NT 0
#endif
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
        m = halo_row(p, tyi, local >> 4, BM / 16) * p.Mw + txi * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
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
      const int r0 = halo_row(p, tyi, wm * (MT * 2), BM / 16), c0 = txi * 16 + lane / NP;
      if (MODE == MODE_CONVT) { pix0 = (2 * r0 + ph) * p.Wout + 2 * c0 + pw; rowstep = 2 * p.Wout; colstep = 2; }
      else { pix0 = r0 * p.Mw + c0; rowstep = p.row_par ? 2 * p.Mw : p.Mw; colstep = 1; }
    } else {
      pix0 = tile_m * BM + wm * (MT * 32) + lane / NP; rowstep = 16; colstep = 1;   // (linear pixels: a "row" is 16 of them)
    }
    const unsigned v0 = (unsigned)pix0 * (unsigned)rowb + (unsigned)(nbw * YSZ + (lane % NP) * 16);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const int ro = 2 * i + ((k * PPI) >> 4), co = (k * PPI) & 15;
        // The row / column step goes into the VECTOR offset, the scalar offset stays the literal 0.  r05: `buffer_store_dwordx4 vdata, voff, rsrc, sN offen` followed IMMEDIATELY by a
        // VALU write of vdata lost a few stores per launch on gfx950 (the next instruction was the address arithmetic of the following store, allocated onto the freed data
        // register).  The compiler's hazard recognizer inserts the wait state for wide stores only when soffset is NOT a register; with an SGPR there it emits none.  Proven by
        // inserting `s_nop 0` after the stores of ONE kernel in the assembly (profiles/r05_store_hazard.txt); matryodshka_amd/isa_lint.py now refuses a library with that sequence.
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pc[i][k]), rsrc_y, v0 + (unsigned)((ro * rowstep + co * colstep) * rowb), 0, MSI_EPI_NT ? 2 : 0);
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
      m = halo_row(p, tyi, local >> 4, BM / 16) * p.Mw + (tile_m - tyi * p.halo_tx) * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
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
    m = halo_row(p, tyi, local >> 4, 4) * p.Mw + (tile_m - tyi * p.halo_tx) * 16 + ((local & 15) ^ (((local >> 4) & 1) * p.halo_xor));
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

// k-step J of a 32-channel group of the stride-2 form -> tap (unit by unit: the four units' taps in the order the patches are swapped) and unit
__device__ constexpr int s2_tap(int J) { return J == 0 ? 0 : J == 1 ? 2 : J == 2 ? 6 : J == 3 ? 8 : J == 4 ? 1 : J == 5 ? 7 : J == 6 ? 3 : J == 7 ? 5 : 4; }
__device__ constexpr int s2_unit(int J) { return J < 4 ? 0 : J < 6 ? 1 : J < 8 ? 2 : 3; }

}  // namespace
