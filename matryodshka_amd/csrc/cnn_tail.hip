// the fused tail (1x1 head + RGBA assembly), LayerNorm finish / apply, zero -- part of the K2 convolution path (see cnn.hip for the design notes, cnn_device.h for the shared pieces).
#include "cnn_device.h"

namespace {

// ---- fused tail: 1x1 head (+ the producer's LayerNorm) + RGBA layer assembly ------------------------------------
// color_pred (nets.py:509-515) followed by infer_msi's layer_prediction for blend_psv (msi.py:130-147) in ONE kernel:
// `pred` (52 MB at the BASELINE size) is never written to or re-read from HBM and one launch disappears.  A workgroup
// owns 32 consecutive pixels: the sweep-volume tile (32 x 6D floats, contiguous) is requested first and stays in
// flight while the 32 x C0 activations (conv8_2 raw, normalised + ReLU'd on the way like the stand-alone head does)
// and the 2D x C0 weights go to LDS and 2 D / 32 waves run the k-steps on the fp32 MFMA -- the SAME instruction
// sequence as the stand-alone head (k ascending, transposed accumulators), so the prediction is bit-identical --
// then bias + tanh + (x+1)/2 land in an LDS tile and the assembly of K3 (geometry.hip, same expressions, no
// contraction) writes float4 texels of the D-major stack.  HBM-bound: reads C0 + 6D floats, writes 4D per pixel.

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
#if defined(MSI_HA_NT) && MSI_HA_NT
      __builtin_nontemporal_store(o.x, &dst->x); __builtin_nontemporal_store(o.y, &dst->y); __builtin_nontemporal_store(o.z, &dst->z); __builtin_nontemporal_store(o.w, &dst->w);
#else
      *dst = o;
#endif
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

}  // namespace

namespace msi_cnn {
int launch_zero(void *ptr, size_t n16, hipStream_t stream) {
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<float4 *>(ptr), n16);
  return msi::check_launch("zero");
}
int launch_ln_finish(int batch, hipStream_t stream, const long long *sums, double inv_n, const double *scl, int *status, const float *gamma, const float *beta, int C,
                     float *aff, int raw16) {
  hipLaunchKernelGGL(ln_finish_kernel, dim3(batch), dim3(256), 0, stream, sums, inv_n, scl, status, gamma, beta, C, aff, raw16);
  return msi::check_launch("ln_finish");
}
int launch_ln_apply(int bf16out, unsigned blocks, int batch, size_t lds, hipStream_t stream, float *x, const long long *sums, double inv_n, const double *scl, int *status,
                    const float *gamma, const float *beta, size_t per_sample, int C, float *aff, unsigned short *yb) {
  const dim3 grid(blocks, batch);
  if (bf16out) hipLaunchKernelGGL(ln_apply_kernel<1>, grid, dim3(256), lds, stream, x, sums, inv_n, scl, status, gamma, beta, per_sample, C, aff, yb);
  else hipLaunchKernelGGL(ln_apply_kernel<0>, grid, dim3(256), lds, stream, x, sums, inv_n, scl, status, gamma, beta, per_sample, C, aff, yb);
  return msi::check_launch("ln_apply");
}
int launch_head_assemble(int bf16in, unsigned grid_x, size_t lds, hipStream_t stream, const HeadAsmParams &q) {
  if (bf16in) hipLaunchKernelGGL(head_assemble_kernel<1>, dim3(grid_x), dim3(256), lds, stream, q);
  else hipLaunchKernelGGL(head_assemble_kernel<0>, dim3(grid_x), dim3(256), lds, stream, q);
  return msi::check_launch("head_assemble");
}
}  // namespace msi_cnn
