// the bf16 halo-patch kernels: conv_halo_bf16_kernel (stride 1), conv_halo_bf16_s2_kernel (stride 2), convt_halo_bf16_kernel (conv-transpose) -- part of the K2 convolution path (see cnn.hip for the design notes, cnn_device.h for the shared pieces).
#include "cnn_device.h"

namespace {

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
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), ts0r = __builtin_amdgcn_s_memrealtime();   // (ts0r: the constant 100 MHz counter, tools/clock_probe.sh)
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
  auto patch_load = [&](const int c) __attribute__((always_inline)) {
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_)
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[k_], c * 128, 0));
    if (APPLY) c_ld = c;
  };
  auto patch_store = [&]() __attribute__((always_inline)) {
    v4f s_[2], t_[2];
    if (APPLY) {   /* the thread's eight channels of chunk c_ld: four ds_read_b128 from the table built in the prologue */
      const float *sp_ = s_tab + c_ld * 64 + cslot * 8;
      s_[0] = *reinterpret_cast<const v4f *>(sp_); s_[1] = *reinterpret_cast<const v4f *>(sp_ + 4);
      t_[0] = *reinterpret_cast<const v4f *>(sp_ + C); t_[1] = *reinterpret_cast<const v4f *>(sp_ + C + 4);
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f o_;
      if (APPLY) {
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);   /* eight fp16 raw values x * 2^-e (s_ carries 2^e) */
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);
        if (has_pad && !pok[k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */
        unsigned w0, w1, w2, w3;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});
      } else {
        o_ = araw[k_];
      }
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;
    }
  };
  // weights of k-step (chunk c, tap) -> ring stage st (run-time); the packed blob is tap-major: row block tap * CH + c
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * (BN / NW) * ROW_BYTES;
    const int soff_ = (tap * CH + c) * p.npad * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    if (BI >= 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
    if (BI == 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);
    }
  };

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
  auto hq = [&](auto Q_c, v4f (&fa_)[4][MT], v4f (&fb_)[4][NT]) __attribute__((always_inline)) {
    constexpr int Q = decltype(Q_c)::value;
    wait_lgkm_frag<(3 - Q) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);
#pragma unroll
    for (int i_ = 0; i_ < MT; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),
                                                              __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[i_][j_], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  auto htap = [&](auto TAP_c) __attribute__((always_inline)) {
    constexpr int TAP = decltype(TAP_c)::value;
    constexpr int KH_ = TAP / 3, KW_ = TAP % 3;
    constexpr int AOFF_ = KH_ * R * G::ROW_PITCH + KW_ * R * G::PIX_BYTES;
    constexpr int AROW_ = 2 * G::ROW_PITCH;   /* next 32-pixel block: two patch rows down */
    v4f fa_[4][MT], fb_[4][NT];
    const unsigned bst_ = (unsigned)st * G::B_STAGE;
    if (ABL & 8) {
#pragma unroll
      for (int q_ = 0; q_ < 4; ++q_) {
#pragma unroll
        for (int i_ = 0; i_ < MT; ++i_) asm volatile("" : "=v"(fa_[q_][i_]));
#pragma unroll
        for (int j_ = 0; j_ < NT; ++j_) asm volatile("" : "=v"(fb_[q_][j_]));
      }
    } else
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      const unsigned ba_ = b_q[q_] + bst_;
#pragma unroll
      for (int i_ = 0; i_ < MT; ++i_)
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base))
                    : i_ == 1 ? (q_ == 0 ? lds_read128<AOFF_ + AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + AROW_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + AROW_ + 64>(a_base) : lds_read128<AOFF_ + AROW_ + 96>(a_base))
                    : i_ == 2 ? (q_ == 0 ? lds_read128<AOFF_ + 2 * AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 2 * AROW_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + 2 * AROW_ + 64>(a_base) : lds_read128<AOFF_ + 2 * AROW_ + 96>(a_base))
                              : (q_ == 0 ? lds_read128<AOFF_ + 3 * AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 3 * AROW_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + 3 * AROW_ + 64>(a_base) : lds_read128<AOFF_ + 3 * AROW_ + 96>(a_base));
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);
    }
    hq(IC<0>{}, fa_, fb_);
    bool issued_;
    {
      if (!(ABL & 2) && TAP == 0 && c + 1 < c1) patch_load(c + 1);
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;
      issued_ = (TAP + PD < 9) || (c + 1 < c1);
      if (ABL & 1) { }
      else if (TAP + PD < 9) { b_issue(c, TAP + PD, sn_); }
      else if (c + 1 < c1) { b_issue(c + 1, TAP + PD - 9, sn_); }
    }
    hq(IC<1>{}, fa_, fb_); hq(IC<2>{}, fa_, fb_); hq(IC<3>{}, fa_, fb_);
    if (ABL & 3) {
      wait_vmcnt<0>();
    } else if (PD == 2) {
      if (TAP == 0 && c + 1 < c1) wait_vmcnt<BI + NPL>();
      else if (issued_) wait_vmcnt<BI>();
      else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();
    }
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    st = st + 1 == NSTG ? 0 : st + 1;
  };

  // ---- prologue: first patch, first PD weight k-steps ----
  MSI_STAMP(6)
  patch_load(c0);
  b_issue(c0, 0, 0);
  if (PD == 2) b_issue(c0, 1, 1);
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
  patch_store();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    htap(IC<0>{}); htap(IC<1>{}); htap(IC<2>{}); htap(IC<3>{}); htap(IC<4>{}); htap(IC<5>{}); htap(IC<6>{}); htap(IC<7>{}); htap(IC<8>{});
    if (c + 1 < c1 && !(ABL & 2)) {
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#ifdef MSI_CONV_TIMING
  const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  emit_tile<BM, BN, MODE_CONV, 1, WR>(p, acc, tile_m, tile_n, 0, b, tid, smem, raw_mul_pre);   // (the k-loop ended with a barrier: LDS is free)
#ifdef MSI_CONV_TIMING
  if (p.dbg && tid == 0) {
    unsigned long long *o = p.dbg + (size_t)blockIdx.x * 24;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[22] = ts0r; o[23] = __builtin_amdgcn_s_memrealtime();
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
  [[maybe_unused]] static constexpr int PW = 17, PH = TH + 1, NPX = PW * PH;
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
  auto patch_load = [&](const int c, auto U_c) __attribute__((always_inline)) {
    constexpr int U = decltype(U_c)::value;
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_)
      araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[U][k_], c * 128, 0));
    if (APPLY) c_ld = c;
  };
  auto patch_store = [&](auto U_c) __attribute__((always_inline)) {
    constexpr int U = decltype(U_c)::value;
    v4f s_[2], t_[2];
    if (APPLY) {   /* the thread's eight channels of chunk c_ld: four ds_read_b128 from the table built in the prologue */
      const float *sp_ = s_tab + c_ld * 64 + cslot * 8;
      s_[0] = *reinterpret_cast<const v4f *>(sp_); s_[1] = *reinterpret_cast<const v4f *>(sp_ + 4);
      t_[0] = *reinterpret_cast<const v4f *>(sp_ + C); t_[1] = *reinterpret_cast<const v4f *>(sp_ + C + 4);
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f o_;
      if (APPLY) {
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);   /* eight fp16 raw values x * 2^-e (s_ carries 2^e) */
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);
        if (has_pad && !pok[U][k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */
        unsigned w0, w1, w2, w3;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});
      } else {
        o_ = araw[k_];
      }
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;
    }
  };
  // weights of k-step (chunk c, tap) -> ring stage st (run-time); the packed blob is tap-major: row block tap * CH + c
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * (BN / NW) * ROW_BYTES;
    const int soff_ = (tap * CH + c) * p.npad * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    if (BI >= 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
    if (BI == 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);
    }
  };

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
  auto hq = [&](auto Q_c, v4f (&fa_)[4][MT], v4f (&fb_)[4][NT]) __attribute__((always_inline)) {
    constexpr int Q = decltype(Q_c)::value;
    wait_lgkm_frag<(3 - Q) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);
#pragma unroll
    for (int i_ = 0; i_ < MT; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),
                                                              __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[i_][j_], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // k-step J = 0..8 of the current 64-channel group: unit, tap and patch offsets are literals (conv_halo_s2_kernel's order)
  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  auto s2step = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int TAP_ = s2_tap(J), U_ = s2_unit(J);
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;
    constexpr bool FIRST_ = J == 0 || J == 4 || J == 6 || J == 8, LAST_ = J == 3 || J == 5 || J == 7 || J == 8;
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;
    constexpr int AROW_ = 2 * G::ROW_PITCH;   /* next 32-pixel block: two patch rows down */
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */
    v4f fa_[4][MT], fb_[4][NT];
    const unsigned bst_ = (unsigned)st * G::B_STAGE;
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      const unsigned ba_ = b_q[q_] + bst_;
#pragma unroll
      for (int i_ = 0; i_ < MT; ++i_)
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + 64>(a_base) : lds_read128<AOFF_ + 96>(a_base))
                              : (q_ == 0 ? lds_read128<AOFF_ + AROW_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + AROW_ + 32>(a_base)
                                : q_ == 2 ? lds_read128<AOFF_ + AROW_ + 64>(a_base) : lds_read128<AOFF_ + AROW_ + 96>(a_base));
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);
    }
    hq(IC<0>{}, fa_, fb_);
    bool issued_;
    {
      if (FIRST_ && more_) {
        if (U_ < 3) patch_load(c, IC<(U_ + 1) & 3>{}); else patch_load(c + 1, IC<0>{});
      }
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;
      issued_ = (J + PD < 9) || (c + 1 < c1);
      if (J + PD < 9) { b_issue(c, s2_tap((J + PD) % 9), sn_); }
      else if (c + 1 < c1) { b_issue(c + 1, s2_tap((J + PD) % 9), sn_); }
    }
    hq(IC<1>{}, fa_, fb_); hq(IC<2>{}, fa_, fb_); hq(IC<3>{}, fa_, fb_);
    /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */
    if (FIRST_ && !LAST_ && more_) wait_vmcnt<BI + NPL>();
    else if (issued_) wait_vmcnt<BI>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    st = st + 1 == NSTG ? 0 : st + 1;
    if (LAST_ && more_) {   /* every wave has read this unit's last tap: swap the patch */
      patch_store(IC<(U_ + 1) & 3>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };

  // ---- prologue: unit 0 of the first group, the first two weight k-steps (taps (0,0), (0,2)) ----
  static_assert(PD == 2 && MT <= 2, "three-stage weight ring; one or two 32-pixel blocks per wave");
  patch_load(c0, IC<0>{});
  b_issue(c0, s2_tap(0), 0);
  b_issue(c0, s2_tap(1), 1);
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
  patch_store(IC<0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; c < c1; ++c) {
    s2step(IC<0>{}); s2step(IC<1>{}); s2step(IC<2>{}); s2step(IC<3>{}); s2step(IC<4>{}); s2step(IC<5>{}); s2step(IC<6>{}); s2step(IC<7>{}); s2step(IC<8>{});
  }
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
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), ts0r = __builtin_amdgcn_s_memrealtime();   // (ts0r: the constant 100 MHz counter, tools/clock_probe.sh)
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
  auto patch_load = [&](const int c) __attribute__((always_inline)) {
    const int s_ = c >= p.cpt0 ? 1 : 0, cc_ = s_ ? c - p.cpt0 : c;
    const unsigned cb_ = (unsigned)((s_ ? p.C1 : p.C0) * 2);
    src_ld = s_;
    if (APPLY) {   /* gamma / beta of the thread's 8 channels (dummy rows when this source is not raw: the vmcnt  */
      /* arithmetic of the k-steps counts the same number of VMEM operations on both paths)                      */
      const bool raw_ = (p.halo_apply >> s_) & 1;
      const float *gb_ = raw_ ? (s_ ? p.ln_gamma1 : p.ln_gamma) : reinterpret_cast<const float *>(p.wpk);
      const float *bb_ = raw_ ? (s_ ? p.ln_beta1 : p.ln_beta) : reinterpret_cast<const float *>(p.wpk);
      const float *gp_ = gb_ + (raw_ ? cc_ * 64 : 0) + cslot * 8, *bp_ = bb_ + (raw_ ? cc_ * 64 : 64) + cslot * 8;
      g8[0] = *reinterpret_cast<const v4f *>(gp_); g8[1] = *reinterpret_cast<const v4f *>(gp_ + 4);
      be8[0] = *reinterpret_cast<const v4f *>(bp_); be8[1] = *reinterpret_cast<const v4f *>(bp_ + 4);
    }
    if (s_ == 0) {
#pragma unroll
      for (int k_ = 0; k_ < NLOAD; ++k_)
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(
            rsrc_a0, pok[k_] ? pixi[k_] * cb_ + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));
    } else {
#pragma unroll
      for (int k_ = 0; k_ < NLOAD; ++k_)
        araw[k_] = __builtin_bit_cast(v4f, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(
            rsrc_a1, pok[k_] ? pixi[k_] * cb_ + (unsigned)(cslot * 16) : OOB, cc_ * ROW_BYTES, 0));
    }
  };
  auto patch_store = [&]() __attribute__((always_inline)) {
    const bool ap_ = APPLY && ((p.halo_apply >> src_ld) & 1);
    v4f s_[2], t_[2];
    if (ap_) {
      const float ih_ = src_ld ? inv1 : inv0, mh_ = src_ld ? mh1 : mh0, ml_ = src_ld ? ml1 : ml0;
      const float uf_ = src_ld ? up1 : up0;
      const v4f nh = {-mh_, -mh_, -mh_, -mh_}, nl = {-ml_, -ml_, -ml_, -ml_};
#pragma unroll
      for (int h_ = 0; h_ < 2; ++h_) {
        const v4f su_ = ih_ * g8[h_];
        t_[h_] = __builtin_elementwise_fma(nl, su_, __builtin_elementwise_fma(nh, su_, be8[h_]));
        s_[h_] = uf_ * su_;
      }
    }
#pragma unroll
    for (int k_ = 0; k_ < NLOAD; ++k_) {
      v4f o_ = araw[k_];
      if (ap_) {
        const v4f z_ = {0.f, 0.f, 0.f, 0.f};
        typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
        const h8_t hx_ = __builtin_bit_cast(h8_t, araw[k_]);
        const v4f x0_ = {(float)hx_[0], (float)hx_[1], (float)hx_[2], (float)hx_[3]};
        const v4f x1_ = {(float)hx_[4], (float)hx_[5], (float)hx_[6], (float)hx_[7]};
        v4f y0 = __builtin_elementwise_max(__builtin_elementwise_fma(x0_, s_[0], t_[0]), z_);
        v4f y1 = __builtin_elementwise_max(__builtin_elementwise_fma(x1_, s_[1], t_[1]), z_);
        if (has_pad && !pok[k_]) { y0 = z_; y1 = z_; }   /* padding is zero AFTER the normalisation */
        unsigned w0, w1, w2, w3;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(y0.x), "v"(y0.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(y0.z), "v"(y0.w));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w2) : "v"(y1.x), "v"(y1.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w3) : "v"(y1.z), "v"(y1.w));
        o_ = __builtin_bit_cast(v4f, u32x4_t{w0, w1, w2, w3});
      }
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = o_;
    }
  };
  // weights of k-step (class, tap, chunk c) -> ring stage st (run-time)
  auto b_issue = [&](const int cls, const int tap, const int c, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * (BN / 4) * ROW_BYTES;
    const int soff_ = ((cls * S + tap * CH + c) * p.npad) * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
    if (BI == 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 16 * ROW_BYTES, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 24 * ROW_BYTES, 0);
    }
  };

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
  auto cq = [&](auto Q_c, auto PWC_c, v4f (&fa_)[4][MT], v4f (&fb_)[4][NT]) __attribute__((always_inline)) {
    constexpr int Q = decltype(Q_c)::value;
    constexpr int PWC = decltype(PWC_c)::value;
    wait_lgkm_frag<(3 - Q) * (MT + NT), MT, NT>(fa_[Q], fb_[Q]);
#pragma unroll
    for (int i_ = 0; i_ < MT; ++i_)
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        acc[PWC][i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb_[Q][j_]),
                                                                   __builtin_bit_cast(bf16x8, fa_[Q][i_]), acc[PWC][i_][j_], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  const int c0 = 0, c1 = CH;
  int c = c0, st = 0;
  auto ctstep = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int PWC_ = J >> 2, TH_ = (J >> 1) & 1, TW_ = J & 1;
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */
    constexpr int AROW_ = 2 * G::ROW_PITCH;
    const unsigned ab_ = TH_ ? a_base1 : a_base0;
    v4f fa_[4][MT], fb_[4][NT];
    const unsigned bst_ = (unsigned)st * G::B_STAGE;
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      const unsigned ba_ = b_q[q_] + bst_;
#pragma unroll
      for (int i_ = 0; i_ < MT; ++i_)
        fa_[q_][i_] = i_ == 0 ? (q_ == 0 ? lds_read128<COFF_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 32>(ab_)
                                : q_ == 2 ? lds_read128<COFF_ + 64>(ab_) : lds_read128<COFF_ + 96>(ab_))
                    : i_ == 1 ? (q_ == 0 ? lds_read128<COFF_ + AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + AROW_ + 32>(ab_)
                                : q_ == 2 ? lds_read128<COFF_ + AROW_ + 64>(ab_) : lds_read128<COFF_ + AROW_ + 96>(ab_))
                    : i_ == 2 ? (q_ == 0 ? lds_read128<COFF_ + 2 * AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 2 * AROW_ + 32>(ab_)
                                : q_ == 2 ? lds_read128<COFF_ + 2 * AROW_ + 64>(ab_) : lds_read128<COFF_ + 2 * AROW_ + 96>(ab_))
                              : (q_ == 0 ? lds_read128<COFF_ + 3 * AROW_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 3 * AROW_ + 32>(ab_)
                                : q_ == 2 ? lds_read128<COFF_ + 3 * AROW_ + 64>(ab_) : lds_read128<COFF_ + 3 * AROW_ + 96>(ab_));
#pragma unroll
      for (int j_ = 0; j_ < NT; ++j_)
        fb_[q_][j_] = j_ == 0 ? lds_read128<0>(ba_) : lds_read128<32 * ROW_BYTES>(ba_);
    }
    cq(IC<0>{}, IC<PWC_>{}, fa_, fb_);
    bool issued_;
    {
      if (J == 0 && c + 1 < c1) patch_load(c + 1);
      int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;
      constexpr int JN_ = (J + PD) & 7;
      issued_ = (J + PD < 8) || (c + 1 < c1);
      if (J + PD < 8) { b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_); }
      else if (c + 1 < c1) { b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_); }
    }
    cq(IC<1>{}, IC<PWC_>{}, fa_, fb_); cq(IC<2>{}, IC<PWC_>{}, fa_, fb_); cq(IC<3>{}, IC<PWC_>{}, fa_, fb_);
    if (J == 0 && c + 1 < c1) wait_vmcnt<BI + NPL>();
    else if (issued_) wait_vmcnt<BI>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    st = st + 1 == NSTG ? 0 : st + 1;
  };

  patch_load(c0);
  b_issue(2 * ph, 0, c0, 0);
  b_issue(2 * ph, 1, c0, 1);
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
  patch_store();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef MSI_CONV_TIMING
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  for (; c < c1; ++c) {
    ctstep(IC<0>{}); ctstep(IC<1>{}); ctstep(IC<2>{}); ctstep(IC<3>{}); ctstep(IC<4>{}); ctstep(IC<5>{}); ctstep(IC<6>{}); ctstep(IC<7>{});
    if (c + 1 < c1) {
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
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
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[22] = ts0r; o[23] = __builtin_amdgcn_s_memrealtime();
    o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
#endif
#endif
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

}  // namespace

namespace msi_cnn {
int launch_bf16_halo(const LayerLaunch &Q, const ConvParams &p, int rate, bool w8, hipStream_t stream) {
  if (Q.halo_tb) {
    // (the 128 x 128 tile with APPLY needs more than the 256 registers of two waves per SIMD -- 120 bytes of scratch inside
    // the chunk loop -- and is not built: the plan only marks sources of the 128 x 64 tile as raw)
    if (p.halo_apply) return Q.hbn == 128 ? msi::fail(MSI_E_UNSUPPORTED, "convt_halo_bf16: APPLY is built for the 128x64 tile")
                                          : launch_convt_halo_bf16<128, 64, 1>(Q, p, stream);
    return Q.hbn == 128 ? launch_convt_halo_bf16<128, 128, 0>(Q, p, stream) : launch_convt_halo_bf16<128, 64, 0>(Q, p, stream);
  }
  if (Q.halo_s2)   // (four waves: with eight the staging path does not fit 128 registers -- 48 bytes of scratch -- and was measured
                   // 0.8 % of the network slower, three interleaved repeats)
    return Q.halo_apply ? launch_halo_bf16_s2<1, 4>(Q, p, stream) : launch_halo_bf16_s2<0, 4>(Q, p, stream);
#define MSI_HB(BM_, BN_, R_, A_) (w8 ? launch_halo_bf16<BM_, BN_, R_, A_, 8>(Q, p, stream) : launch_halo_bf16<BM_, BN_, R_, A_, 4>(Q, p, stream))
  if (Q.hbm == 128) {
    if (rate == 1) return Q.halo_apply ? MSI_HB(128, 128, 1, 1) : MSI_HB(128, 128, 1, 0);
    return Q.halo_apply ? MSI_HB(128, 128, 2, 1) : MSI_HB(128, 128, 2, 0);
  }
#undef MSI_HB
  // (the 256 x 64 tile stays at four waves: eight do not fit their 128 registers -- 44-64 bytes of scratch -- and were
  // measured slower, conv8_2 27.8 k -> 32.9 k cycles per workgroup)
  return Q.halo_apply ? launch_halo_bf16<256, 64, 1, 1, 4>(Q, p, stream) : launch_halo_bf16<256, 64, 1, 0, 4>(Q, p, stream);
}
}  // namespace msi_cnn
