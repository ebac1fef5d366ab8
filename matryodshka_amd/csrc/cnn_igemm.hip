// conv_igemm_kernel (tap-DMA implicit-GEMM convolution, fp32 / bf16) and conv_fixup_kernel -- part of the K2 convolution path (see cnn.hip for the design notes, cnn_device.h for the shared pieces).
#include "cnn_device.h"

namespace {

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
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), ts0r = __builtin_amdgcn_s_memrealtime();   // (ts0r: the constant 100 MHz counter, tools/clock_probe.sh)
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
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[22] = ts0r; o[23] = __builtin_amdgcn_s_memrealtime();
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

namespace msi_cnn {
int launch_igemm(const LayerLaunch &Q, const ConvParams &p, int bf16, hipStream_t stream) {
  switch (Q.tile) {
    case TILE_128x128: return launch_conv<128, 128>(Q, p, bf16, stream);
    case TILE_128x64: return launch_conv<128, 64>(Q, p, bf16, stream);
#ifdef MSI_EXPERIMENTS
    case TILE_64x128: return bf16 ? msi::fail(MSI_E_UNSUPPORTED, "conv: 64x128 is an fp32 tile") : launch_conv<64, 128>(Q, p, 0, stream);
#else
    case TILE_64x128: return msi::fail(MSI_E_UNSUPPORTED, "conv: the 64x128 fp32 tile is an experiment (MSI_EXPERIMENTS)");
#endif
    default: return launch_conv<64, 64>(Q, p, bf16, stream);
  }
}
int launch_fixup(int bm, int bn, int mode, unsigned gx, unsigned gy, const ConvParams &p, hipStream_t stream) {
  if (bn != 64) return msi::fail(MSI_E_UNSUPPORTED, "conv_fixup: tile %d x %d", bm, bn);
  if (bm == 64 && mode == MODE_CONVT) hipLaunchKernelGGL((conv_fixup_kernel<64, 64, MODE_CONVT>), dim3(gx, gy), dim3(256), 0, stream, p);
  else if (bm == 64 && mode == MODE_CONV) hipLaunchKernelGGL((conv_fixup_kernel<64, 64, MODE_CONV>), dim3(gx, gy), dim3(256), 0, stream, p);
  else if (bm == 128 && mode == MODE_CONV) hipLaunchKernelGGL((conv_fixup_kernel<128, 64, MODE_CONV>), dim3(gx, gy), dim3(256), 0, stream, p);
  else if (bm == 128 && mode == MODE_CONVT) hipLaunchKernelGGL((conv_fixup_kernel<128, 64, MODE_CONVT>), dim3(gx, gy), dim3(256), 0, stream, p);
  else return msi::fail(MSI_E_UNSUPPORTED, "conv_fixup: tile %d x %d, mode %d", bm, bn, mode);
  return msi::check_launch("conv_fixup");
}
int debug_conv_occupancy(int lds_bytes) {
  int n = -1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_igemm_kernel<64, 64, MODE_CONV, 0>, 256, (size_t)lds_bytes);
  return n;
}
}  // namespace msi_cnn
