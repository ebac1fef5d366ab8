// fp32 convolutions through the 3-way bf16 split (six products) / 2-way fp16 split (three): conv_halo_x3_kernel, conv_halo8_x3_kernel, conv_halo_s2_x3_kernel, convt_halo_x3_kernel -- part of the K2 convolution path (see cnn.hip for the design notes, cnn_device.h for the shared pieces).
#include "cnn_device.h"

namespace {

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
#ifndef MSI_X2_NSTG   // weight ring of the fp16 form (half the matrix work per tap: the DMA latency budget of a two-stage ring is one SHORT tap)
#define MSI_X2_NSTG 3
#endif
// RATE: 1, 2 = the dilation; 3 (r05) = dilation 2 on a ROW-PARITY tile (ConvParams::row_par): the tile's four rows are every other image row, so along H the taps are one
// tile row apart (halo 1) and only W keeps the dilation: a 6 x 20-pixel patch (25.3 KB) instead of 8 x 20 (33.8 KB), i.e. the two-stage ring and THREE workgroups per CU.
constexpr int x3_rate_y(int RATE) { return RATE == 3 ? 1 : RATE; }
constexpr int x3_rate_x(int RATE) { return RATE == 3 ? 2 : RATE; }
constexpr int x3_nstg(int RATE, int NPL) { return NPL == 2 ? MSI_X2_NSTG : (MSI_X3_NSTG ? MSI_X3_NSTG : (RATE == 2 ? 3 : 2)); }
template <int RATE, int NS = x3_nstg(RATE, 3), int NPL = 3, int TH = 4>
struct HaloGeomX3 {
  static constexpr int PW = 16 + 2 * x3_rate_x(RATE), PH = TH + 2 * x3_rate_y(RATE), NPX = PW * PH;   // TH x 16 output pixels per workgroup (TH = 4, or 8: conv_halo8_x3_kernel)
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
  typedef HaloGeomX3<RATE, x3_nstg(RATE, NPL), NPL, TH> G;
  constexpr int RY = x3_rate_y(RATE), RX = x3_rate_x(RATE), PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  static_assert(RATE != 3 || TH == 4, "row-parity tiles: four rows");
  constexpr int MT = TH / 4, NT = 1, BM = 16 * TH;
  static_assert(TH == 4 || (TH == 8 && NPL == 3 && RATE == 1), "the 8-row tile is built for the six-product form at rate 1");
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), ts0r = __builtin_amdgcn_s_memrealtime();   // (ts0r: the constant 100 MHz counter, tools/clock_probe.sh)
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
  const int ow0 = (tile_m - tyi * p.halo_tx) * 16;
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
  int c = c0, cpar = 0;   // chunk of the k-loop and (two-stage ring) its stage parity: captured by the k-step lambdas below
  (void)cpar;
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    if (MSI_X3_ABLATE & 1) return;
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * G::B_ROW;
    const int soff_ = (tap * CH + c) * NPL * plane_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0);
    if (NPL == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0);
  };
  b_issue(c0, 0, 0);
  if (G::NSTG == 3) b_issue(c0, 1, 1);

  // ---- per-lane patch elements: e = tid + 256 k -> patch pixel e / 8, 16-byte channel slot e % 8 (= tid % 8) ----
  unsigned voff[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = RATE == 3 ? (tyi >> 1) * (2 * TH) + (tyi & 1) + 2 * (py - 1) : tyi * TH - RY + py;   // (row-parity tile: patch row py is image row base + 2 (py - 1))
    int iw = ow0 - RX + px;
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
  auto patch_load = [&](const int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_)
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[k_], c * ROW_BYTES, 0));
    if (APPLY) {
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + c * 32 + cslot * 4);
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + c * 32 + cslot * 4);
    }
  };
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
  auto patch_store = [&]() __attribute__((always_inline)) {
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    if (APPLY) {   /* scale = inv * gamma; shift = beta - mean * scale with the mean as hi + lo floats: fp32 ops only */
      s4 = inv_f * g4;
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f y = araw[k_];
      if (APPLY) {
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */
      }
      if (NPL == 2) {   /* y = h + m' 2^-11, fp16 parts (round to nearest even; y - h is exact in fp32) */
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u2x_t __attribute__((ext_vector_type(2)));
        f16_range_track(amax_, y);
        const h2_t ha = {(_Float16)y.x, (_Float16)y.y}, hb = {(_Float16)y.z, (_Float16)y.w};
        const h2_t ma = {(_Float16)((y.x - (float)ha.x) * 2048.f), (_Float16)((y.y - (float)ha.y) * 2048.f)};
        const h2_t mb = {(_Float16)((y.z - (float)hb.x) * 2048.f), (_Float16)((y.w - (float)hb.y) * 2048.f)};
        if (lds_a[k_] != 0xffffffffu) {
          *reinterpret_cast<u2x_t *>(smem + lds_a[k_]) = u2x_t{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
          *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 64) = u2x_t{__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
        }
      } else {
      /* y = h + m + l, bf16 parts (round to nearest even; y - h and (y - h) - m are exact in fp32) */
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
      if (lds_a[k_] != 0xffffffffu) {
        typedef unsigned u2x_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_]) = u2x_t{h0, h1};
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 64) = u2x_t{m0, m1};
        *reinterpret_cast<u2x_t *>(smem + lds_a[k_] + 128) = u2x_t{l0, l1};
      }
    }
    }
  };
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
  auto htap = [&](auto TAP_c) __attribute__((always_inline)) {
    constexpr int TAP = decltype(TAP_c)::value;
    constexpr int KH_ = TAP / 3, KW_ = TAP % 3;
    /* ring stage of this k-step: three stages -> TAP % 3 (a literal); two stages -> (TAP + chunk parity) & 1 (run-time scalar) */
    const unsigned bst_ = (unsigned)(G::NSTG == 3 ? TAP % 3 : ((TAP ^ cpar) & 1)) * G::B_STAGE;
    constexpr int AOFF_ = KH_ * RY * G::ROW_PITCH + KW_ * RX * G::PIX_BYTES;
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];
    if (MSI_X3_EARLY_DMA || G::NSTG == 2) {   /* the k-step NSTG - 1 ahead: its ring stage was last read in the previous k-step (closing barrier passed) */
      constexpr int PD_ = G::NSTG - 1;
      const int stn_ = G::NSTG == 3 ? (TAP + 2) % 3 : (((TAP ^ cpar) & 1) ^ 1);   /* (two stages: the other one) */
      if (TAP + PD_ < 9) { b_issue(c, TAP + PD_, stn_); }
      else if (c + 1 < c1) { b_issue(c + 1, TAP + PD_ - 9, stn_); }
    }
    if (NPL == 2) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);
        bh_[s_] = lds_read128<0>(b_s[s_] + bst_);
        am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);
        bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);
      }
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bm_[s_]), __builtin_bit_cast(f16x8, ah_[s_]), acc_lo, 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh_[s_]), __builtin_bit_cast(f16x8, ah_[s_]), acc[0][0], 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bh_[s_]), __builtin_bit_cast(f16x8, am_[s_]), acc_lo, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s_ == 0) {
          if (TAP == 0 && c + 1 < c1) patch_load(c + 1);
          if (!MSI_X3_EARLY_DMA && G::NSTG == 3) {
            if (TAP + 2 < 9) { b_issue(c, TAP + 2, (TAP + 2) % 3); }
            else if (c + 1 < c1) { b_issue(c + 1, TAP - 7, (TAP + 2) % 3); }
          }
        }
      }
    } else if constexpr (MT == 2) {
      /* two pixel blocks i = 0, 1 against ONE set of weight fragments per K16 step s: 18 reads (at most 12 in flight: lgkmcnt is four bits), 24 MFMAs */
      constexpr int A1_ = AOFF_ + 2 * G::ROW_PITCH;
      v4f xh_[2][2], xm_[2][2], xl_[2][2];   /* [s][i] */
      bh_[0] = lds_read128<0>(b_s[0] + bst_); bm_[0] = lds_read128<G::B_PLANE>(b_s[0] + bst_); bl_[0] = lds_read128<2 * G::B_PLANE>(b_s[0] + bst_);
      xh_[0][0] = lds_read128<AOFF_>(a_base); xm_[0][0] = lds_read128<AOFF_ + 64>(a_base); xl_[0][0] = lds_read128<AOFF_ + 128>(a_base);
      xh_[0][1] = lds_read128<A1_>(a_base); xm_[0][1] = lds_read128<A1_ + 64>(a_base); xl_[0][1] = lds_read128<A1_ + 128>(a_base);
      bh_[1] = lds_read128<0>(b_s[1] + bst_); bm_[1] = lds_read128<G::B_PLANE>(b_s[1] + bst_); bl_[1] = lds_read128<2 * G::B_PLANE>(b_s[1] + bst_);
      wait_lgkm6<6>(bh_[0], bm_[0], bl_[0], xh_[0][0], xm_[0][0], xl_[0][0]);
      split_mfma<3>(acc[0][0], acc_lo, xh_[0][0], xm_[0][0], xl_[0][0], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      xh_[1][0] = lds_read128<AOFF_ + 32>(a_base); xm_[1][0] = lds_read128<AOFF_ + 96>(a_base); xl_[1][0] = lds_read128<AOFF_ + 160>(a_base);
      xh_[1][1] = lds_read128<A1_ + 32>(a_base); xm_[1][1] = lds_read128<A1_ + 96>(a_base); xl_[1][1] = lds_read128<A1_ + 160>(a_base);
      if (TAP == 0 && c + 1 < c1) patch_load(c + 1);
      wait_lgkm6<9>(xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      split_mfma<3>(acc[1][0], acc_lo, xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<3>(bh_[1], bm_[1], bl_[1], xh_[1][0], xm_[1][0], xl_[1][0]);
      split_mfma<3>(acc[0][0], acc_lo, xh_[1][0], xm_[1][0], xl_[1][0], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<0>(xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      split_mfma<3>(acc[1][0], acc_lo, xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    if (!(MSI_X3_ABLATE & 8))
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
      ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);
      am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);
      al_[s_] = s_ == 0 ? lds_read128<AOFF_ + 128>(a_base) : lds_read128<AOFF_ + 160>(a_base);
      bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);
    }
    /* six products per K16 step, small terms first: m.m, l.h, h.l, m.h, h.m, h.h (weights = the MFMA's row operand) */
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
      if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);
      else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);
      if (!(MSI_X3_ABLATE & 16)) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm_[s_]), __builtin_bit_cast(bf16x8, am_[s_]), acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, al_[s_]), acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bl_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, am_[s_]), acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bm_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bh_[s_]), __builtin_bit_cast(bf16x8, ah_[s_]), acc[0][0], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == 0) {
        if (TAP == 0 && c + 1 < c1) patch_load(c + 1);
        /* k-step two ahead: (c, TAP + 2) or (c + 1, TAP - 7) */
        if (!MSI_X3_EARLY_DMA && G::NSTG == 3) {
        if (TAP + 2 < 9) { b_issue(c, TAP + 2, (TAP + 2) % 3); }
        else if (c + 1 < c1) { b_issue(c + 1, TAP - 7, (TAP + 2) % 3); }
        }
      }
    }
    }
    {
      const bool issued_ = (TAP + 2 < 9) || (c + 1 < c1);
      if (G::NSTG == 2) {
        if (TAP == 0 && c + 1 < c1) wait_vmcnt<NLOAD + (APPLY ? 2 : 0)>();   /* (the patch loads were issued after the DMA) */
        else wait_vmcnt<0>();
      } else if (TAP == 0 && c + 1 < c1) wait_vmcnt<NPL + NLOAD + (APPLY ? 2 : 0)>();   /* patch loads + this tap's DMA in flight (either order) */
      else if (issued_) wait_vmcnt<NPL>();
      else wait_vmcnt<0>();
    }
    if (!(MSI_X3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: first patch, first two weight k-steps ----
  patch_load(c0);                                      // (the weights of k-steps 0 and 1 are on their way already)
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
  patch_store();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    cpar = (c - c0) & 1;   // (two-stage ring: nine k-steps per chunk flip the stage parity)
    htap(IC<0>{}); htap(IC<1>{}); htap(IC<2>{}); htap(IC<3>{}); htap(IC<4>{}); htap(IC<5>{}); htap(IC<6>{}); htap(IC<7>{}); htap(IC<8>{});
    if (c + 1 < c1 && !(MSI_X3_ABLATE & 32)) {   // every wave has read the last tap of this chunk (closing barrier of tap 8): swap the patch
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
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
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[22] = ts0r; o[23] = __builtin_amdgcn_s_memrealtime();
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
template <int NP, int TH = 4>
struct HaloGeomS2X3 {
  static constexpr int PW = 17, PH = TH + 1, NPX = PW * PH;
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
#ifndef MSI_S2X3_ABLATE   // timing experiments only (wrong results): 1 no weight DMA, 4 no per-k-step barrier, 8 no fragment reads, 16 no MFMAs, 32 no patch swap, 64 no patch loads
#define MSI_S2X3_ABLATE 0
#endif
#endif
// TH = 8 (r05, conv_halo8_s2_x3_kernel, six-product form): 8 x 16 output pixels per workgroup -- a wave owns four output rows = TWO 32-pixel blocks that share the weight fragments;
// the four unit patches are 9 x 17 pixels and serve twice the outputs per swap; 58.0 KB of LDS: two workgroups per CU.  Grid rule as for the other 8-row tiles.
template <int APPLY, int NP, int TH>
__device__ __forceinline__ void conv_halo_s2_x3_body(const ConvParams &p, char *smem) {
  typedef HaloGeomS2X3<NP, TH> G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = TH / 4, NT = 1, BM = 16 * TH;
  static_assert(TH == 4 || (TH == 8 && NP == 3 && G::NSTG == 2), "the 8-row tile: six-product form, two-stage ring");
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
  const int oh0 = tyi * TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;   // the tile of the OUTPUT grid
  const int H = p.Hin, W = p.Win, C = p.C0;
  // the first two weight k-steps (taps (0,0) and (0,2) of group c0) before anything else
  const int S = p.ksteps;
  // (weights: the x3 block of the packed blob, three 64-byte-row planes per k-step -- see conv_halo_x3_kernel)
  const int plane_bytes = p.npad * G::B_ROW;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk_x3, 0, (int)((size_t)S * NP * plane_bytes), 0x00020000);
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + (lane >> 2)) * G::B_ROW + (lane & 3) * 16);
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    if (MSI_S2X3_ABLATE & 1) return;
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * G::B_ROW;
    const int soff_ = (tap * CH + c) * NP * plane_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0);
    if (NP == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0);
  };
  b_issue(c0, 0, 0);
  if (G::NSTG == 3) b_issue(c0, 2, 1);

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
  auto patch_load = [&](const int c, auto U_c) __attribute__((always_inline)) {
    constexpr int U = decltype(U_c)::value;
    if (MSI_S2X3_ABLATE & 64) return;
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_)
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[U][k_], c * ROW_BYTES, 0));
    if (APPLY && U == 0) {
      g4 = *reinterpret_cast<const v4f *>(p.ln_gamma + c * 32 + cslot * 4);
      be4 = *reinterpret_cast<const v4f *>(p.ln_beta + c * 32 + cslot * 4);
    }
  };
  // registers -> LDS patch, the producer's affine + ReLU applied (ln_apply_kernel's expressions: same bits)
  auto patch_store = [&](auto U_c) __attribute__((always_inline)) {
    constexpr int U = decltype(U_c)::value;
    if (APPLY && U == 0) {
      s4 = inv_f * g4;
      const v4f nh = {-mu_hi, -mu_hi, -mu_hi, -mu_hi}, nl = {-mu_lo, -mu_lo, -mu_lo, -mu_lo};
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f y = araw[k_];
      if (APPLY) {
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});
        if (has_pad && !pok[U][k_]) y = v4f{0.f, 0.f, 0.f, 0.f};   /* padding is zero AFTER the normalisation */
      }
      split_store<NP>(smem, lds_a[k_], y, amax_);
    }
  };

  // ---- MFMA side (as conv_halo_kernel: a wave owns two tile rows x 16 columns x 32 channels) ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  const unsigned a_base = lds_base + (unsigned)((2 * MT * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 16);
  unsigned b_s[2];
  (void)fswz;
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
    b_s[s_] = lds_base + G::A_BYTES + (wn * 32 + frow) * G::B_ROW + (((2 * s_ + fh) ^ ((frow >> 2) & 3)) << 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8), unused));
  f32x16 acc[MT][1], acc_lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc_lo[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][0][r] = 0.f;
  }

  // k-step J = 0..8 of the current group: unit, tap and patch offsets are literals
  int c = c0, cpar = 0;   // (the group being multiplied -- unit 0's patch of group c0 is on its way -- and its ring parity)
  auto s2step = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int TAP_ = s2_tap(J), U_ = s2_unit(J);
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;
    constexpr bool FIRST_ = J == 0 || J == 4 || J == 6 || J == 8, LAST_ = J == 3 || J == 5 || J == 7 || J == 8;
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */
    /* ring stage of this k-step: three stages -> J % 3 (a literal); two -> (J + group parity) & 1 (nine k-steps per group flip it) */
    const unsigned bst_ = (unsigned)(G::NSTG == 3 ? J % 3 : ((J ^ cpar) & 1)) * G::B_STAGE;
    if (G::NSTG == 2) {   /* the NEXT k-step's weights into the other stage: it was last read in the previous k-step (closing barrier passed) */
      const int stn_ = ((J ^ cpar) & 1) ^ 1;
      if (J + 1 < 9) b_issue(c, s2_tap((J + 1) % 9), stn_);
      else if (c + 1 < c1) b_issue(c + 1, s2_tap(0), stn_);
    }
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];
    if constexpr (MT == 2) {
      /* two pixel blocks i = 0, 1 (the wave's output rows 0-1 and 2-3) against ONE set of weight fragments per K16 step s: 18 reads, 24 MFMAs (conv_halo_x3_body's MT = 2 sequence) */
      constexpr int A1_ = AOFF_ + 2 * G::ROW_PITCH;
      v4f xh_[2][2], xm_[2][2], xl_[2][2];   /* [s][i] */
      bh_[0] = lds_read128<0>(b_s[0] + bst_); bm_[0] = lds_read128<G::B_PLANE>(b_s[0] + bst_); bl_[0] = lds_read128<2 * G::B_PLANE>(b_s[0] + bst_);
      xh_[0][0] = lds_read128<AOFF_>(a_base); xm_[0][0] = lds_read128<AOFF_ + 64>(a_base); xl_[0][0] = lds_read128<AOFF_ + 128>(a_base);
      xh_[0][1] = lds_read128<A1_>(a_base); xm_[0][1] = lds_read128<A1_ + 64>(a_base); xl_[0][1] = lds_read128<A1_ + 128>(a_base);
      bh_[1] = lds_read128<0>(b_s[1] + bst_); bm_[1] = lds_read128<G::B_PLANE>(b_s[1] + bst_); bl_[1] = lds_read128<2 * G::B_PLANE>(b_s[1] + bst_);
      wait_lgkm6<6>(bh_[0], bm_[0], bl_[0], xh_[0][0], xm_[0][0], xl_[0][0]);
      split_mfma<3>(acc[0][0], acc_lo, xh_[0][0], xm_[0][0], xl_[0][0], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      xh_[1][0] = lds_read128<AOFF_ + 32>(a_base); xm_[1][0] = lds_read128<AOFF_ + 96>(a_base); xl_[1][0] = lds_read128<AOFF_ + 160>(a_base);
      xh_[1][1] = lds_read128<A1_ + 32>(a_base); xm_[1][1] = lds_read128<A1_ + 96>(a_base); xl_[1][1] = lds_read128<A1_ + 160>(a_base);
      if (FIRST_ && more_) {
        if (U_ < 3) patch_load(c, IC<(U_ + 1) & 3>{});
        else patch_load(c + 1, IC<0>{});
      }
      wait_lgkm6<9>(xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      split_mfma<3>(acc[1][0], acc_lo, xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<3>(bh_[1], bm_[1], bl_[1], xh_[1][0], xm_[1][0], xl_[1][0]);
      split_mfma<3>(acc[0][0], acc_lo, xh_[1][0], xm_[1][0], xl_[1][0], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<0>(xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      split_mfma<3>(acc[1][0], acc_lo, xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    if (!(MSI_S2X3_ABLATE & 8))
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      ah_[s_] = s_ == 0 ? lds_read128<AOFF_>(a_base) : lds_read128<AOFF_ + 32>(a_base);
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);
      am_[s_] = s_ == 0 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base);
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);
      if (NP == 3) {
        al_[s_] = s_ == 0 ? lds_read128<AOFF_ + 128>(a_base) : lds_read128<AOFF_ + 160>(a_base);
        bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);
      } else { al_[s_] = ah_[s_]; bl_[s_] = bh_[s_]; }
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      if (NP == 3) {
        if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);
        else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);
      } else {
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);
      }
      if (!(MSI_S2X3_ABLATE & 16)) split_mfma<NP>(acc[0][0], acc_lo, ah_[s_], am_[s_], al_[s_], bh_[s_], bm_[s_], bl_[s_]);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == 0) {
        if (FIRST_ && more_) {
          if (U_ < 3) patch_load(c, IC<(U_ + 1) & 3>{});
          else patch_load(c + 1, IC<0>{});
        }
        /* (three stages) k-step two ahead: (c, J + 2) or (c + 1, J - 7) */
        if (G::NSTG == 3) {
        if (J + 2 < 9) b_issue(c, s2_tap((J + 2) % 9), (J + 2) % 3);
        else if (c + 1 < c1) b_issue(c + 1, s2_tap((J + 2) % 9), (J + 2) % 3);
        }
      }
    }
    }
    {
      const bool issued_ = (J + 2 < 9) || (c + 1 < c1);
      /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */
      if (G::NSTG == 2) {   /* (the DMA went out BEFORE the patch loads of this k-step: in-order return) */
        if (FIRST_ && !LAST_ && more_) wait_vmcnt<NLOAD>();
        else wait_vmcnt<0>();
      } else if (FIRST_ && !LAST_ && more_) wait_vmcnt<NP + NLOAD>();
      else if (issued_) wait_vmcnt<NP>();
      else wait_vmcnt<0>();
    }
    if (!(MSI_S2X3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    if (LAST_ && more_ && !(MSI_S2X3_ABLATE & 32)) {   /* every wave has read this unit's last tap: swap the patch */
      patch_store(IC<(U_ + 1) & 3>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };

  // ---- prologue: unit 0 of the first group ----
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
  patch_store(IC<0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    cpar = (c - c0) & 1;   // (two-stage ring: nine k-steps per group flip the stage parity)
    s2step(IC<0>{}); s2step(IC<1>{}); s2step(IC<2>{}); s2step(IC<3>{}); s2step(IC<4>{});
    s2step(IC<5>{}); s2step(IC<6>{}); s2step(IC<7>{}); s2step(IC<8>{});
  }
  split_finish<NP>(acc[0][0], acc_lo, amax_, lane, p.status);

  // ---- epilogue: as conv_halo_kernel ----
  if (!full) {
    constexpr int SLAB = BM * 64 * 4;
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)slot * (BM * 64)), 0, SLAB, 0x00020000);
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
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + (size_t)(slot - ks) * (BM * 64)), 0, nsp * SLAB, 0x00020000);
    sum_slabs<MT, NT, MSI_HANDOFF_AUX>(acc, rsrc_t, nsp, SLAB, tid);
    __syncthreads();   // (s_old has been read by every thread before the strip below reuses LDS)
  }
  emit_tile<BM, 64, MODE_CONV>(p, acc, tile_m, tile_n, 0, b, tid, smem);
}

template <int APPLY, int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 2 ? MSI_S2X_WAVES : (MSI_S2X3_NSTG == 2 ? 3 : 2))))
conv_halo_s2_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_halo_s2_x3_body<APPLY, NP, 4>(p, smem);
#endif
}

template <int APPLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
conv_halo8_s2_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_halo_s2_x3_body<APPLY, 3, 8>(p, smem);
#endif
}

// ---- the conv-transpose halo kernel through the six-product bf16 split (convt_halo_kernel x conv_halo_x3_kernel; r04) ----------
// At native fp32 the two-class halo form lost to the tap kernel (above): the tap kernel's k-loop has no VALU and five workgroups
// per CU.  The tap kernel cannot split its operands (both arrive by DMA), this one stages the patch through registers anyway:
// with 192 instead of 512 matrix cycles per 16 channels it wins (conv8_1 197 -> ... us, see profiles/r04_*), and its sources
// need no ln_apply launch.
// (r05) FIVE patch rows: a class row of parity ph reads the input row itself and ONE neighbour -- the row above (ph = 0, and both parities of msi_train_net's VALID form) or the
// row below (ph = 1) -- so the sixth row of the stride-1 geometry was staged and never read: 90 instead of 108 patch pixels (three instead of four loads per lane).
template <int NS, int NPL, int TH = 4>
struct HaloGeomCT3 : HaloGeomX3<1, NS, NPL, TH> {
  typedef HaloGeomX3<1, NS, NPL, TH> B_;
  static constexpr int PH = TH + 1, NPX = B_::PW * PH;
  static constexpr int A_BYTES = PH * B_::ROW_PITCH;
  static constexpr int LDS_BYTES = A_BYTES + B_::NSTG * B_::B_STAGE;
  static constexpr int NLOAD = (NPX * 8 + 255) / 256;
};
#ifndef MSI_CT_MAXW
#define MSI_CT_MAXW 8
#endif
#ifndef MSI_CT3_ABLATE   // timing experiments only (wrong results): bits as MSI_S2X3_ABLATE
#define MSI_CT3_ABLATE 0
#endif
#ifndef MSI_CT3_NSTG   // weight ring of the six-product conv-transpose kernel: 2 (r05: 48.4 KB of LDS, three workgroups per CU; eight k-steps per chunk, so the
#define MSI_CT3_NSTG 2 // stage of k-step J is the literal J & 1 and the DMA of k-step J + 1 goes out at the head of k-step J) or 3 (r04: 60.7 KB, two per CU)
#endif
// TH = 8 (r05, convt_halo8_x3_kernel, six-product form): 8 x 16 input pixels per workgroup and row parity -- a wave owns four input rows = TWO 32-pixel blocks per class that
// share the weight fragments (as conv_halo8_x3_kernel: 18 instead of 24 fragment reads per 24 MFMAs, half the weight bytes / prologues / patch swaps per output); four
// accumulators per wave, 9 x 18-pixel patch, 60.3 KB of LDS: two workgroups per CU.  Whole-grid rule as for the stride-1 tile (plan: >= 3 tiles per CU).
template <int NP, int TH>
__device__ __forceinline__ void convt_halo_x3_body(const ConvParams &p, char *smem) {
  typedef HaloGeomCT3<(NP == 3 ? MSI_CT3_NSTG : 3), NP, TH> G;
  constexpr int PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD, NSTG = G::NSTG, PD = NSTG - 1;
  constexpr int MT = TH / 4, BM = 16 * TH;
  static_assert(NSTG == 3 || NSTG == 2, "prefetch distance two or one");
  static_assert(TH == 4 || (TH == 8 && NP == 3 && NSTG == 2), "the 8-row tile: six-product form, two-stage ring");
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
  const int oh0 = tyi * TH, ow0 = (tile_m - tyi * p.halo_tx) * 16;
  const int H = p.Hin, W = p.Win;
  const int rowup = (ph && !p.wrap) ? 0 : 1;              // patch rows oh0 - rowup .. oh0 - rowup + 4: the neighbour row is above (1) or below (0) -- HaloGeomCT3

  // the first two weight k-steps (class pw = 0, taps 0 and 1 of chunk c0) go out before the patch addresses are worked out
  const int S = p.ksteps;                                 // k-steps per class: 4 CH
  // (weights: the x3 block, [class][tap * CH + c][plane h | m | l][npad][64 B] -- see conv_halo_x3_kernel)
  const int plane_bytes = p.npad * G::B_ROW;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpk_x3, 0, (int)((size_t)4 * S * NP * plane_bytes), 0x00020000);
  const unsigned b_voff = (unsigned)((tile_n * 64 + wave * 16 + (lane >> 2)) * G::B_ROW + (lane & 3) * 16);
  auto b_issue = [&](const int cls, const int tap, const int c, const int st) __attribute__((always_inline)) {
    if (MSI_CT3_ABLATE & 1) return;
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * G::B_ROW;
    const int soff_ = (cls * S + tap * CH + c) * NP * plane_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + G::B_PLANE), 16, b_voff, soff_ + plane_bytes, 0, 0);
    if (NP == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)(sB_ + 2 * G::B_PLANE), 16, b_voff, soff_ + 2 * plane_bytes, 0, 0);
  };
  b_issue(2 * ph, 0, c0, 0);
  if (NSTG == 3) b_issue(2 * ph, 1, c0, 1);

  // ---- per-lane patch elements (as conv_halo_kernel; the byte offset depends on the source's channel count) ----
  unsigned pixi[NLOAD], lds_a[NLOAD];
  bool pok[NLOAD];
  const int cslot = tid & 7;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int pp = (tid + 256 * k) >> 3;
    const int py = pp / PW, px = pp - py * PW;
    const int ih = oh0 - rowup + py;
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
  auto patch_load = [&](const int c) __attribute__((always_inline)) {
    if (MSI_CT3_ABLATE & 64) return;
    const int s_ = c >= p.cpt0 ? 1 : 0, cc_ = s_ ? c - p.cpt0 : c;
    const unsigned cb_ = (unsigned)((s_ ? p.C1 : p.C0) * 4);
    src_ld = s_;
    if (s_ == 0) {
#pragma unroll
      for (int k_ = 0; k_ < NLOAD; ++k_)
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(
            rsrc_a0, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));
    } else {
#pragma unroll
      for (int k_ = 0; k_ < NLOAD; ++k_)
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(
            rsrc_a1, pok[k_] ? __umul24(pixi[k_], cb_) + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));
    }
    const float *gp_ = (s_ ? p.ln_gamma1 : p.ln_gamma), *bp_ = (s_ ? p.ln_beta1 : p.ln_beta);
    if ((p.halo_apply >> s_) & 1) {
      g4 = *reinterpret_cast<const v4f *>(gp_ + cc_ * 32 + cslot * 4);
      be4 = *reinterpret_cast<const v4f *>(bp_ + cc_ * 32 + cslot * 4);
    } else {   /* (same number of VMEM operations on both paths: the vmcnt arithmetic of the k-steps counts them) */
      g4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16);
      be4 = *reinterpret_cast<const v4f *>(p.wpk + cslot * 16 + 128);
    }
  };
  auto patch_store = [&]() __attribute__((always_inline)) {
    const bool ap_ = (p.halo_apply >> src_ld) & 1;
    v4f s4 = {1.f, 1.f, 1.f, 1.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    if (ap_) {
      const float ih_ = src_ld ? inv_f[1] : inv_f[0], mh_ = src_ld ? mu_hi[1] : mu_hi[0], ml_ = src_ld ? mu_lo[1] : mu_lo[0];
      s4 = ih_ * g4;
      const v4f nh = {-mh_, -mh_, -mh_, -mh_}, nl = {-ml_, -ml_, -ml_, -ml_};
      t4 = __builtin_elementwise_fma(nl, s4, __builtin_elementwise_fma(nh, s4, be4));
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f y = araw[k_];
      if (ap_) {
        y = __builtin_elementwise_max(__builtin_elementwise_fma(y, s4, t4), v4f{0.f, 0.f, 0.f, 0.f});
        if (has_pad && !pok[k_]) y = v4f{0.f, 0.f, 0.f, 0.f};
      }
      split_store<NP>(smem, lds_a[k_], y, amax_);
    }
  };

  // ---- MFMA side ----
  const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;
  // fragment base of tap row th = 0 (patch row rowup + local row) and of th = 1 (one row up for ph = 0, one down for ph = 1)
  const unsigned a_base0 = lds_base + (unsigned)((rowup + 2 * MT * wm + (frow >> 4)) * G::ROW_PITCH + ((frow & 15) ^ ((frow >> 4) << 3)) * G::PIX_BYTES + fh * 16);
  // (msi_train_net's VALID form: tap 1 is the row ABOVE / the column to the LEFT in both parities -- tap_delta)
  const unsigned a_base1 = rowup ? a_base0 - G::ROW_PITCH : a_base0 + G::ROW_PITCH;
  const unsigned wadj = p.wrap ? 2u * G::PIX_BYTES : 0u;
  unsigned b_s[2];
  (void)fswz;
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
    b_s[s_] = lds_base + G::A_BYTES + (wn * 32 + frow) * G::B_ROW + (((2 * s_ + fh) ^ ((frow >> 2) & 3)) << 4);
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8), unused));
  f32x16 acc[2][MT][1], acc_lo[2];   // [class][32-pixel block]
#pragma unroll
  for (int cl = 0; cl < 2; ++cl)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc_lo[cl][r] = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[cl][i][0][r] = 0.f;
    }

  // k-step J of the chunk = class pw = J / 4, tap (th, tw) = ((J / 2) & 1, J & 1), weights in ring stage st (run-time).
  // The DMA of the k-step PD = 2 ahead and (J == 0) the next chunk's patch loads are issued after the first MFMA quarter;
  // before the closing barrier the NEXT k-step's weights must have landed: they were issued one k-step ago, so only what
  // THIS k-step issued (2 DMA, + the patch loads of J == 0) may still be in flight (in-order return).
  int c = c0, st = 0;                                     // the chunk being multiplied and the ring stage of the current k-step
  constexpr int NPLD = NLOAD + 2;                         // VMEM operations of a patch load
  auto ctstep = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int PWC_ = J >> 2, TH_ = (J >> 1) & 1, TW_ = J & 1;
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */
    const unsigned ab_ = (TH_ ? a_base1 : a_base0) - ((PWC_ && TW_) ? wadj : 0u);
    v4f ah_[2], am_[2], al_[2], bh_[2], bm_[2], bl_[2];
    const unsigned bst_ = (unsigned)st * G::B_STAGE;
    bool issued_ = false;
    if (NSTG == 2) {   /* the NEXT k-step's weights into the other stage (last read in the previous k-step: closing barrier passed) */
      constexpr int JN_ = (J + 1) & 7;
      if (J + 1 < 8) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c, st ^ 1); }
      else if (c + 1 < c1) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, st ^ 1); }
    }
    if constexpr (MT == 2) {
      /* two pixel blocks i = 0, 1 (the wave's rows 0-1 and 2-3) against ONE set of weight fragments per K16 step s: 18 reads, 24 MFMAs (conv_halo_x3_body's MT = 2 sequence) */
      constexpr int C1_ = COFF_ + 2 * G::ROW_PITCH;
      v4f xh_[2][2], xm_[2][2], xl_[2][2];   /* [s][i] */
      bh_[0] = lds_read128<0>(b_s[0] + bst_); bm_[0] = lds_read128<G::B_PLANE>(b_s[0] + bst_); bl_[0] = lds_read128<2 * G::B_PLANE>(b_s[0] + bst_);
      xh_[0][0] = lds_read128<COFF_>(ab_); xm_[0][0] = lds_read128<COFF_ + 64>(ab_); xl_[0][0] = lds_read128<COFF_ + 128>(ab_);
      xh_[0][1] = lds_read128<C1_>(ab_); xm_[0][1] = lds_read128<C1_ + 64>(ab_); xl_[0][1] = lds_read128<C1_ + 128>(ab_);
      bh_[1] = lds_read128<0>(b_s[1] + bst_); bm_[1] = lds_read128<G::B_PLANE>(b_s[1] + bst_); bl_[1] = lds_read128<2 * G::B_PLANE>(b_s[1] + bst_);
      wait_lgkm6<6>(bh_[0], bm_[0], bl_[0], xh_[0][0], xm_[0][0], xl_[0][0]);
      split_mfma<3>(acc[PWC_][0][0], acc_lo[PWC_], xh_[0][0], xm_[0][0], xl_[0][0], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      xh_[1][0] = lds_read128<COFF_ + 32>(ab_); xm_[1][0] = lds_read128<COFF_ + 96>(ab_); xl_[1][0] = lds_read128<COFF_ + 160>(ab_);
      xh_[1][1] = lds_read128<C1_ + 32>(ab_); xm_[1][1] = lds_read128<C1_ + 96>(ab_); xl_[1][1] = lds_read128<C1_ + 160>(ab_);
      if (J == 0 && c + 1 < c1) patch_load(c + 1);
      wait_lgkm6<9>(xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      split_mfma<3>(acc[PWC_][1][0], acc_lo[PWC_], xh_[0][1], xm_[0][1], xl_[0][1], bh_[0], bm_[0], bl_[0]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<3>(bh_[1], bm_[1], bl_[1], xh_[1][0], xm_[1][0], xl_[1][0]);
      split_mfma<3>(acc[PWC_][0][0], acc_lo[PWC_], xh_[1][0], xm_[1][0], xl_[1][0], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm6<0>(xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      split_mfma<3>(acc[PWC_][1][0], acc_lo[PWC_], xh_[1][1], xm_[1][1], xl_[1][1], bh_[1], bm_[1], bl_[1]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
    if (!(MSI_CT3_ABLATE & 8))
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      ah_[s_] = s_ == 0 ? lds_read128<COFF_>(ab_) : lds_read128<COFF_ + 32>(ab_);
      bh_[s_] = lds_read128<0>(b_s[s_] + bst_);
      am_[s_] = s_ == 0 ? lds_read128<COFF_ + 64>(ab_) : lds_read128<COFF_ + 96>(ab_);
      bm_[s_] = lds_read128<G::B_PLANE>(b_s[s_] + bst_);
      if (NP == 3) {
        al_[s_] = s_ == 0 ? lds_read128<COFF_ + 128>(ab_) : lds_read128<COFF_ + 160>(ab_);
        bl_[s_] = lds_read128<2 * G::B_PLANE>(b_s[s_] + bst_);
      } else { al_[s_] = ah_[s_]; bl_[s_] = bh_[s_]; }
    }
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      if (NP == 3) {
        if (s_ == 0) wait_lgkm6<6>(ah_[0], bh_[0], am_[0], bm_[0], al_[0], bl_[0]);
        else wait_lgkm6<0>(ah_[1], bh_[1], am_[1], bm_[1], al_[1], bl_[1]);
      } else {
        if (s_ == 0) wait_lgkm4<4>(ah_[0], bh_[0], am_[0], bm_[0]);
        else wait_lgkm4<0>(ah_[1], bh_[1], am_[1], bm_[1]);
      }
      if (!(MSI_CT3_ABLATE & 16)) split_mfma<NP>(acc[PWC_][0][0], acc_lo[PWC_], ah_[s_], am_[s_], al_[s_], bh_[s_], bm_[s_], bl_[s_]);
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == 0) {
        if (J == 0 && c + 1 < c1) patch_load(c + 1);
        if (NSTG == 3) {
        int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;
        constexpr int JN_ = (J + PD) & 7;
        if (J + PD < 8) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_); }
        else if (c + 1 < c1) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_); }
        }
      }
    }
    }
    if (NSTG == 2) {   /* (the DMA went out before this k-step's patch loads: in-order return) */
      if (J == 0 && c + 1 < c1) wait_vmcnt<NPLD>();
      else wait_vmcnt<0>();
    } else if (J == 0 && c + 1 < c1) wait_vmcnt<NP + NPLD>();
    else if (issued_) wait_vmcnt<NP>();
    else wait_vmcnt<0>();
    if (!(MSI_CT3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    st = st + 1 == NSTG ? 0 : st + 1;
  };

  // ---- prologue: first patch (the first two weight k-steps are on their way), the sources' LayerNorm statistics ----
  patch_load(c0);
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
  patch_store();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    ctstep(IC<0>{}); ctstep(IC<1>{}); ctstep(IC<2>{}); ctstep(IC<3>{}); ctstep(IC<4>{}); ctstep(IC<5>{}); ctstep(IC<6>{}); ctstep(IC<7>{});
    if (c + 1 < c1 && !(MSI_CT3_ABLATE & 32)) {
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  split_finish<NP>(acc[0][0][0], acc_lo[0], amax_, lane, p.status);
  split_finish<NP>(acc[1][0][0], acc_lo[1], amax_, lane, p.status);

  // ---- epilogue: two class tiles ----
  if (!full) {
    constexpr int SLAB = BM * 64 * 4;
    if (p.tile_cnt == nullptr) {                          // separate fix-up launch (conv_fixup_kernel, class = blockIdx.y)
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (BM * 64)), 0, SLAB, 0x00020000);
        dump_acc<MT, 1, 0>(acc[cl], rsrc_p, tid);
      }
      return;
    }
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)slot * 2 + cl) * (BM * 64)), 0, SLAB, 0x00020000);
      dump_acc<MT, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_p, tid);
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
      const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)(p.partial + ((size_t)(slot - ks) * 2 + cl) * (BM * 64)), 0, nsp * 2 * SLAB, 0x00020000);
      sum_slabs<MT, 1, MSI_HANDOFF_AUX>(acc[cl], rsrc_t, nsp, 2 * SLAB, tid);
    }
  }
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) emit_tile<BM, 64, MODE_CONVT>(p, acc[cl], tile_m, tile_n, 2 * ph + cl, b, tid);
}

template <int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, MSI_CT_MAXW)))
convt_halo_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  convt_halo_x3_body<NP, 4>(p, smem);
#endif
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
convt_halo8_x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  convt_halo_x3_body<3, 8>(p, smem);
#endif
}

}  // namespace

namespace msi_cnn {
int launch_x3(const LayerLaunch &Q, const ConvParams &p, int rate, hipStream_t stream) {
  const dim3 grid(Q.nblocks), block(256);
  if (Q.halo_t) {
    constexpr int lds_ct3 = HaloGeomCT3<MSI_CT3_NSTG, 3>::LDS_BYTES, lds_ct2 = HaloGeomCT3<3, 2>::LDS_BYTES;
    constexpr int lds_ct8 = HaloGeomCT3<MSI_CT3_NSTG, 3, 8>::LDS_BYTES;
    static_assert(lds_ct8 <= 65536, "convt_halo8_x3_kernel: LDS without the launch attribute");
    if (Q.x3_th8) hipLaunchKernelGGL(convt_halo8_x3_kernel, grid, block, lds_ct8, stream, p);
    else if (Q.halo_x2) hipLaunchKernelGGL(convt_halo_x3_kernel<2>, grid, block, lds_ct2, stream, p);
    else hipLaunchKernelGGL(convt_halo_x3_kernel<3>, grid, block, lds_ct3, stream, p);
    int rc = msi::check_launch("convt_halo_x3");
    if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) rc = launch_fixup(Q.x3_th8 ? 128 : 64, 64, MODE_CONVT, Q.nfix, 2, p, stream);
    return rc;
  }
  if (Q.halo_s2 && Q.x3_th8) {
    typedef HaloGeomS2X3<3, 8> G8s_;
    static_assert(G8s_::LDS_BYTES <= 65536, "conv_halo8_s2_x3_kernel: LDS without the launch attribute");
    if (Q.halo_apply) hipLaunchKernelGGL((conv_halo8_s2_x3_kernel<1>), grid, block, G8s_::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_halo8_s2_x3_kernel<0>), grid, block, G8s_::LDS_BYTES, stream, p);
  } else if (Q.halo_s2) {
    if (Q.halo_x2) {
      if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_x3_kernel<1, 2>), grid, block, HaloGeomS2X3<2>::LDS_BYTES, stream, p);
      else hipLaunchKernelGGL((conv_halo_s2_x3_kernel<0, 2>), grid, block, HaloGeomS2X3<2>::LDS_BYTES, stream, p);
    } else {
      if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_x3_kernel<1, 3>), grid, block, HaloGeomS2X3<3>::LDS_BYTES, stream, p);
      else hipLaunchKernelGGL((conv_halo_s2_x3_kernel<0, 3>), grid, block, HaloGeomS2X3<3>::LDS_BYTES, stream, p);
    }
  } else if (Q.x3_th8) {
    typedef HaloGeomX3<1, (MSI_X3_NSTG ? MSI_X3_NSTG : 2), 3, 8> G8_;
    static_assert(G8_::LDS_BYTES <= 65536, "conv_halo8_x3_kernel: LDS without the launch attribute");
    if (Q.halo_apply) hipLaunchKernelGGL((conv_halo8_x3_kernel<1, 3>), grid, block, G8_::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_halo8_x3_kernel<0, 3>), grid, block, G8_::LDS_BYTES, stream, p);
  } else {
    static thread_local unsigned long long done2[12] = {0};       // (above 64 KB of LDS the launch needs the attribute)
#define MSI_X3_LAUNCH(R, A, N)                                                                                         \
  {                                                                                                                    \
    typedef HaloGeomX3<R, x3_nstg(R, N), N> G_;                                                                        \
    if (G_::LDS_BYTES > 65536) {                                                                                       \
      int rc0 = set_max_lds(reinterpret_cast<const void *>(conv_halo_x3_kernel<R, A, N>), G_::LDS_BYTES, done2[(R - 1) * 4 + A * 2 + (N - 2)], "conv_halo_x3"); \
      if (rc0) return rc0;                                                                                             \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_halo_x3_kernel<R, A, N>), grid, block, G_::LDS_BYTES, stream, p);                          \
  }
    const int sel = (rate == 1 ? 0 : (p.row_par ? 8 : 4)) + (Q.halo_apply ? 2 : 0) + (Q.halo_x2 ? 0 : 1);
    switch (sel) {
      case 0: MSI_X3_LAUNCH(1, 0, 2) break;
      case 1: MSI_X3_LAUNCH(1, 0, 3) break;
      case 2: MSI_X3_LAUNCH(1, 1, 2) break;
      case 3: MSI_X3_LAUNCH(1, 1, 3) break;
      case 4: MSI_X3_LAUNCH(2, 0, 2) break;
      case 5: MSI_X3_LAUNCH(2, 0, 3) break;
      case 6: MSI_X3_LAUNCH(2, 1, 2) break;
      case 7: MSI_X3_LAUNCH(2, 1, 3) break;
      case 8: MSI_X3_LAUNCH(3, 0, 2) break;
      case 9: MSI_X3_LAUNCH(3, 0, 3) break;
      case 10: MSI_X3_LAUNCH(3, 1, 2) break;
      default: MSI_X3_LAUNCH(3, 1, 3) break;
    }
#undef MSI_X3_LAUNCH
  }
  int rc = msi::check_launch("conv_halo_x3");
  if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) rc = launch_fixup(Q.x3_th8 ? 128 : 64, 64, MODE_CONV, Q.nfix, 1, p, stream);
  return rc;
}
}  // namespace msi_cnn
