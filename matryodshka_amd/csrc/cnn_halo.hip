// the native fp32 halo-patch kernels: conv_halo_kernel (stride 1, rate 1 / 2), conv_halo_s2_kernel (stride 2), convt_halo_kernel (conv-transpose) -- part of the K2 convolution path (see cnn.hip for the design notes, cnn_device.h for the shared pieces).
#include "cnn_device.h"

namespace {

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
  [[maybe_unused]] static constexpr int NLOAD = (NPX * 8 + 255) / 256;     // float4 patch elements per thread and chunk
};

template <int RATE, int APPLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RATE == 1 ? 4 : 3)))
conv_halo_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef HaloGeom<RATE> G;
  constexpr int R = RATE, PW = G::PW, NPX = G::NPX, NLOAD = G::NLOAD;
  constexpr int MT = 1, NT = 1;
#ifdef MSI_CONV_TIMING
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), ts0r = __builtin_amdgcn_s_memrealtime();   // (ts0r: the constant 100 MHz counter, tools/clock_probe.sh)
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
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;
    }
  };
  // weights of k-step (chunk c, tap) -> ring stage st; the packed blob is tap-major: row block tap * CH + c
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * ROW_BYTES;
    const int soff_ = (tap * CH + c) * p.npad * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
  };

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
  int c = c0;
  auto htap = [&](auto TAP_c) __attribute__((always_inline)) {
    constexpr int TAP = decltype(TAP_c)::value;
    constexpr int KH_ = TAP / 3, KW_ = TAP % 3, ST_ = TAP % 3;
    constexpr int AOFF_ = KH_ * R * G::ROW_PITCH + KW_ * R * G::PIX_BYTES;
    v4f a_[4], b_[4];
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      a_[q_] = q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 16>(a_base)
             : q_ == 2 ? lds_read128<AOFF_ + 32>(a_base) : lds_read128<AOFF_ + 48>(a_base);
      b_[q_] = lds_read128<ST_ * G::B_STAGE>(b_q[q_]);
    }
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q_ == 0) {
        if (TAP == 0 && c + 1 < c1) patch_load(c + 1);
        /* k-step two ahead: (c, TAP + 2) or (c + 1, TAP - 7) */
        if (TAP + 2 < 9) { b_issue(c, TAP + 2, (TAP + 2) % 3); }
        else if (c + 1 < c1) { b_issue(c + 1, TAP - 7, (TAP + 2) % 3); }
      }
    }
    {
      const bool issued_ = (TAP + 2 < 9) || (c + 1 < c1);
      if (TAP == 0 && c + 1 < c1) wait_vmcnt<2 + NLOAD + (APPLY ? 2 : 0)>();   /* patch loads + this tap's DMA in flight */
      else if (issued_) wait_vmcnt<2>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
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
    htap(IC<0>{}); htap(IC<1>{}); htap(IC<2>{}); htap(IC<3>{}); htap(IC<4>{}); htap(IC<5>{}); htap(IC<6>{}); htap(IC<7>{}); htap(IC<8>{});
    if (c + 1 < c1) {   // every wave has read the last tap of this chunk (closing barrier of tap 8): swap the patch
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
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
  [[maybe_unused]] static constexpr int NLOAD = (NPX * 8 + 255) / 256;
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
  auto b_issue = [&](const int c, const int tap, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * ROW_BYTES;
    const int soff_ = (tap * CH + c) * p.npad * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
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
  auto patch_load = [&](const int c, auto U_c) __attribute__((always_inline)) {
    constexpr int U = decltype(U_c)::value;
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
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;
    }
  };

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
  int c = c0;   // (unit 0's patch of group c0 is on its way)
  auto s2step = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int TAP_ = s2_tap(J), U_ = s2_unit(J), ST_ = J % 3;
    constexpr int DY_ = (TAP_ / 3) >> 1, DX_ = (TAP_ % 3) >> 1;
    constexpr bool FIRST_ = J == 0 || J == 4 || J == 6 || J == 8, LAST_ = J == 3 || J == 5 || J == 7 || J == 8;
    constexpr int AOFF_ = DY_ * G::ROW_PITCH + DX_ * G::PIX_BYTES;
    const bool more_ = U_ < 3 || c + 1 < c1;               /* a unit follows this one */
    v4f a_[4], b_[4];
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      a_[q_] = q_ == 0 ? lds_read128<AOFF_>(a_base) : q_ == 1 ? lds_read128<AOFF_ + 16>(a_base)
             : q_ == 2 ? lds_read128<AOFF_ + 32>(a_base) : lds_read128<AOFF_ + 48>(a_base);
      b_[q_] = lds_read128<ST_ * G::B_STAGE>(b_q[q_]);
    }
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q_ == 0) {
        if (FIRST_ && more_) {
          if (U_ < 3) patch_load(c, IC<(U_ + 1) & 3>{}); else patch_load(c + 1, IC<0>{});
        }
        /* k-step two ahead: (c, J + 2) or (c + 1, J - 7) */
        if (J + 2 < 9) { b_issue(c, s2_tap((J + 2) % 9), (J + 2) % 3); }
        else if (c + 1 < c1) { b_issue(c + 1, s2_tap((J + 2) % 9), (J + 2) % 3); }
      }
    }
    {
      const bool issued_ = (J + 2 < 9) || (c + 1 < c1);
      /* the NEXT k-step's weights must have landed; the patch requested in this k-step may stay in flight unless it is stored now */
      if (FIRST_ && !LAST_ && more_) wait_vmcnt<2 + NLOAD>();
      else if (issued_) wait_vmcnt<2>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (LAST_ && more_) {   /* every wave has read this unit's last tap: swap the patch */
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
    s2step(IC<0>{}); s2step(IC<1>{}); s2step(IC<2>{}); s2step(IC<3>{}); s2step(IC<4>{}); s2step(IC<5>{}); s2step(IC<6>{}); s2step(IC<7>{}); s2step(IC<8>{});
  }

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
  auto b_issue = [&](const int cls, const int tap, const int c, const int st) __attribute__((always_inline)) {
    char *sB_ = smem + G::A_BYTES + st * G::B_STAGE + wave * 16 * ROW_BYTES;
    const int soff_ = ((cls * S + tap * CH + c) * p.npad) * ROW_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void *)sB_, 16, b_voff, soff_, 8 * ROW_BYTES, 0);
  };
  b_issue(2 * ph, 0, c0, 0);
  b_issue(2 * ph, 1, c0, 1);

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
  auto patch_load = [&](const int c) __attribute__((always_inline)) {
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
      if (lds_a[k_] != 0xffffffffu) *reinterpret_cast<v4f *>(smem + lds_a[k_]) = y;
    }
  };

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
  int c = c0, st = 0;
  auto ctstep = [&](auto J_c) __attribute__((always_inline)) {
    constexpr int J = decltype(J_c)::value;
    constexpr int PWC_ = J >> 2, TH_ = (J >> 1) & 1, TW_ = J & 1;
    constexpr int COFF_ = (1 + (PWC_ ? TW_ : -TW_)) * G::PIX_BYTES;   /* column of the tap: immediate */
    const unsigned ab_ = TH_ ? a_base1 : a_base0;
    v4f a_[4], b_[4];
    const unsigned bst_ = (unsigned)st * G::B_STAGE;
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      a_[q_] = q_ == 0 ? lds_read128<COFF_>(ab_) : q_ == 1 ? lds_read128<COFF_ + 16>(ab_)
             : q_ == 2 ? lds_read128<COFF_ + 32>(ab_) : lds_read128<COFF_ + 48>(ab_);
      b_[q_] = lds_read128<0>(b_q[q_] + bst_);
    }
    bool issued_ = false;
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      if (q_ == 0) wait_lgkm<6>(a_[0], b_[0]);
      if (q_ == 1) wait_lgkm<4>(a_[1], b_[1]);
      if (q_ == 2) wait_lgkm<2>(a_[2], b_[2]);
      if (q_ == 3) wait_lgkm<0>(a_[3], b_[3]);
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].x, a_[q_].x, acc[PWC_][0][0], 0, 0, 0);
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].y, a_[q_].y, acc[PWC_][0][0], 0, 0, 0);
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].z, a_[q_].z, acc[PWC_][0][0], 0, 0, 0);
      acc[PWC_][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_[q_].w, a_[q_].w, acc[PWC_][0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q_ == 0) {
        if (J == 0 && c + 1 < c1) patch_load(c + 1);
        int sn_ = st + PD; sn_ = sn_ >= NSTG ? sn_ - NSTG : sn_;
        constexpr int JN_ = (J + PD) & 7;
        if (J + PD < 8) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c, sn_); }
        else if (c + 1 < c1) { issued_ = true; b_issue(2 * ph + (JN_ >> 2), JN_ & 3, c + 1, sn_); }
      }
    }
    if (J == 0 && c + 1 < c1) wait_vmcnt<2 + NPL>();
    else if (issued_) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
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
    if (c + 1 < c1) {
      patch_store();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

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

}  // namespace

namespace msi_cnn {
int launch_halo_f32(const LayerLaunch &Q, const ConvParams &p, int rate, hipStream_t stream) {
  const dim3 grid(Q.nblocks), block(256);
  if (Q.halo_t) {
    hipLaunchKernelGGL(convt_halo_kernel, grid, block, ConvtHaloGeom::LDS_BYTES, stream, p);
    int rc = msi::check_launch("convt_halo");
    if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) rc = launch_fixup(64, 64, MODE_CONVT, Q.nfix, 2, p, stream);
    return rc;
  }
  if (Q.halo_s2) {
    if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_s2_kernel<1>), grid, block, HaloGeomS2::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_halo_s2_kernel<0>), grid, block, HaloGeomS2::LDS_BYTES, stream, p);
  } else if (rate == 1) {
    if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_kernel<1, 1>), grid, block, HaloGeom<1>::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_halo_kernel<1, 0>), grid, block, HaloGeom<1>::LDS_BYTES, stream, p);
  } else {
    if (Q.halo_apply) hipLaunchKernelGGL((conv_halo_kernel<2, 1>), grid, block, HaloGeom<2>::LDS_BYTES, stream, p);
    else hipLaunchKernelGGL((conv_halo_kernel<2, 0>), grid, block, HaloGeom<2>::LDS_BYTES, stream, p);
  }
  int rc = msi::check_launch("conv_halo");
  if (!rc && Q.nfix > 0 && p.tile_cnt == nullptr) rc = launch_fixup(64, 64, MODE_CONV, Q.nfix, 1, p, stream);
  return rc;
}
}  // namespace msi_cnn
