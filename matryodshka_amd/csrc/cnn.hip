// K2: the encoder-decoder CNN of nets.msi_coord_train_net / nets.msi_train_net
// (reference nets.py:471-515 / 387-450) as implicit-GEMM convolutions on the
// gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fma chain).
//
// One kernel template serves every layer:
//   * conv3x3 (stride 1/2, rate 1/2, SAME-zero or wrap padding), the 1x1 head,
//     and conv-transpose 4x4 s2 as four output-parity sub-convolutions of 2x2
//     taps each (blockIdx.z selects the parity class);
//   * GEMM view: M = pixels of one sample, N = Cout, K = taps x Cin, walked in
//     k-steps of 32 channels of one tap; A is gathered on the fly from the NHWC
//     producer(s) (two sources = skip concat by pointer pair, no concat copy);
//   * the producer's LayerNorm (+ReLU) is applied in the A-operand loader as a
//     per-channel scale/shift (zero padding stays zero: padding is applied to
//     the normalised activation in the reference), so normalised activations
//     are never written to HBM;
//   * CoordNet's |sin(lat)| channel (nets.py:260-265) is constant along W: it is
//     one extra k-step whose 32 "channels" are the <=9 taps of that channel,
//     keeping K a multiple of 32;
//   * the epilogue writes the raw conv output and one (count, mean, M2) partial
//     per workgroup; a small finish kernel merges the partials in fp64 in a fixed
//     order (Chan) into the per-channel scale/shift the consumer loads.
//
// Tiling: 256 threads = 4 wavefronts (2x2), wave tile (BM/2)x(BN/2) of 32x32
// MFMA tiles, BK=32, double-buffered LDS (row stride 36 floats: ds_read_b128 of
// 16 distinct rows is bank-conflict free), register-staged global prefetch of
// k-step s+1 issued before the MFMAs of step s, one barrier per k-step.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "msi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 struct copies lower to memcpy through scratch)

constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;
constexpr int NPAD_ALIGN = 128;
constexpr double LN_EPS = 1e-12;  // slim.layer_norm variance epsilon [TF-knowledge]

enum { MODE_CONV = 0, MODE_CONVT = 1, MODE_HEAD = 2 };

struct ConvParams {
  const float *x0, *x1;      // NHWC sources (x1 = second half of a skip concat or null)
  const float *aff0, *aff1;  // [2][C]: scale | shift of the producer's LayerNorm (identity table for raw inputs)
  int aff_bs0, aff_bs1;      // per-sample stride of aff0/aff1 in floats (2*C, or 0 for the identity table)
  float floor0, floor1;      // 0 (ReLU after LayerNorm) or -inf (raw input)
  const float *wpk;          // packed weights [nclass][ksteps][npad][32]
  const float *coord;        // |sin(lat)| per input row [Hin], or null
  const float *bias;         // head only
  float *y;                  // raw output NHWC [B,Hout,Wout,Cout]
  float *stats;              // [B][nparts][4] (count, mean, M2, -) or null
  int C0, C1;
  int Hin, Win, Hout, Wout, Cout, npad;
  int Mh, Mw;                // GEMM row grid per sample (output grid; input grid for convT)
  int ntaps, cpt, ksteps;    // taps, 32-channel chunks per tap, total k-steps (incl. coord step)
  int stride, rate, pad_t, pad_l;
  int mode, wrap, nclass;
  int ablate;                // debug only (MSI_CONV_ABLATE): 1 = skip global loads, 2 = skip MFMAs
};

template <int MODE>
__device__ __forceinline__ void tap_offset(int rate, int tap, int ph, int pw, int &dh, int &dw) {
  if (MODE == MODE_CONV) {
    const int kh = tap / 3, kw = tap - kh * 3;
    dh = kh * rate;
    dw = kw * rate;
  } else if (MODE == MODE_CONVT) {
    // y[2i + k - 1] += x[i] w[k]: even outputs use k=1 (i = o/2) and k=3 (i = o/2 - 1),
    // odd outputs k=2 (i = (o-1)/2) and k=0 (i = (o+1)/2).
    const int th = tap >> 1, tw = tap & 1;
    dh = th == 0 ? 0 : (ph ? 1 : -1);
    dw = tw == 0 ? 0 : (pw ? 1 : -1);
  } else {
    dh = 0;
    dw = 0;
  }
}

template <int BM, int BN, int MODE>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(const ConvParams p) {
  constexpr int MT = BM / 64, NT = BN / 64;  // 32x32 tiles per wave (2x2 waves)
  constexpr int AR = BM / 32, BR = BN / 32;  // v4f rows per thread per k-step
  constexpr int STAGE = (BM + BN) * LDS_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_m = blockIdx.x, tile_n = blockIdx.y;
  const int cls = blockIdx.z % p.nclass, b = blockIdx.z / p.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  const int mtot = p.Mh * p.Mw;

  // ---- loader-side row bookkeeping ---------------------------------------------------------
  const int cq = tid & 7, lr = tid >> 3;
  int r_ih[AR], r_iw[AR];
  unsigned r_ok = 0;
#pragma unroll
  for (int rr = 0; rr < AR; ++rr) {
    const int m = tile_m * BM + lr + 32 * rr;
    const int mh = m / p.Mw, mw = m - mh * p.Mw;
    r_ih[rr] = mh * p.stride - p.pad_t;
    r_iw[rr] = mw * p.stride - p.pad_l;
    if (m < mtot) r_ok |= 1u << rr;
  }
  const size_t in_pix = (size_t)p.Hin * p.Win;
  const int wrap_w = p.wrap ? p.Win : 0;
  const float *wbase = p.wpk + ((size_t)cls * p.ksteps * p.npad + (size_t)tile_n * BN) * BK + tid * 4;

  // register staging of one k-step (A rows, B rows, the producer's LayerNorm affine, masks);
  // two instances: the loads of steps s+1 and s+2 are both in flight while step s computes
  struct Stage {
    v4f ra[AR], rb[BR];
    v4f sc4, sh4;
    float floor_v;
    unsigned okm;
  };
  Stage stA, stB;

  // Input coordinates of GEMM row rr for tap offset (dh, dw); loads are UNCONDITIONAL from a
  // clamped address (no control flow around the global loads, so the whole k-step is one basic
  // block and the loads stay in flight across the MFMAs); the validity bit masks the value when
  // it is written to LDS.
  auto tap_coords = [&](int rr, int dh, int dw, int &ihc, int &iwc) __attribute__((always_inline)) -> bool {
    const int ih = r_ih[rr] + dh;
    int iw = r_iw[rr] + dw;
    iw = iw < 0 ? iw + wrap_w : (iw >= p.Win ? iw - wrap_w : iw);  // wrap_w = 0: plain zero padding
    const bool ok = ((r_ok >> rr) & 1u) & (ih >= 0) & (ih < p.Hin) & (iw >= 0) & (iw < p.Win);
    ihc = min(max(ih, 0), p.Hin - 1);
    iwc = min(max(iw, 0), p.Win - 1);
    return ok;
  };

  // Regular k-step = 32 channels [c0, c0+32) of tap `tap`.  The expensive part of the address
  // arithmetic (tap offset, wrap, bounds, clamping) depends only on the tap, so it is done once
  // per tap (new_tap) and a k-step itself costs one multiply-add per row (gen_addr).
  int t_pix[AR];             // clamped input pixel index of each row for the current tap
  unsigned t_okm = 0;        // validity of each row for the current tap
  int a_off[AR];             // element offsets of the A rows inside the selected source
  unsigned a_okm = 0;
  const float *a_src = p.x0, *a_aff = p.aff0, *a_wb = wbase;
  int a_C = p.C0;
  int a_cc = 0;
  float a_floor = -INFINITY;
  // per-source bases hoisted into SGPRs (selecting p.x0/p.x1 directly makes the compiler index
  // the kernarg segment with a dependent s_load every k-step)
  const int kC0 = p.C0, kC1 = p.C1;
  const float *src0 = p.x0 + (size_t)b * in_pix * kC0;
  const float *src1 = p.x1 + (size_t)b * in_pix * kC1;
  const float *affp0 = p.aff0 + (size_t)b * p.aff_bs0;
  const float *affp1 = p.aff1 + (size_t)b * p.aff_bs1;
  const float kfloor0 = p.floor0, kfloor1 = p.floor1;
  // source selection as a byte delta from source 0 (a select between two POINTERS is lowered to a
  // lookup in a private array + flat loads; an integer select stays in SGPRs and keeps the
  // global address space)
  const long d_src = (const char *)src1 - (const char *)src0;
  const long d_aff = (const char *)affp1 - (const char *)affp0;

  auto new_tap = [&](int tap) __attribute__((always_inline)) {
    int dh, dw;
    tap_offset<MODE>(p.rate, tap, ph, pw, dh, dw);
    t_okm = 0;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) {
      int ihc, iwc;
      const bool ok = tap_coords(rr, dh, dw, ihc, iwc);
      t_pix[rr] = ihc * p.Win + iwc;
      t_okm |= (ok ? 1u : 0u) << rr;
    }
  };

  auto gen_addr = [&](int s, int c0) __attribute__((always_inline)) {
    const bool first = c0 < kC0;  // wave-uniform: scalar selects, no branch
    const int C = first ? kC0 : kC1;
    const int c = (first ? c0 : c0 - kC0) + cq * 4;
    a_src = reinterpret_cast<const float *>((const char *)src0 + (first ? 0L : d_src));
    a_aff = reinterpret_cast<const float *>((const char *)affp0 + (first ? 0L : d_aff));
    a_floor = first ? kfloor0 : kfloor1;
    a_C = C;
    const bool cvalid = c < C;
    a_cc = cvalid ? c : 0;
    a_okm = cvalid ? t_okm : 0u;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) a_off[rr] = t_pix[rr] * C + a_cc;
    a_wb = wbase + (size_t)s * p.npad * BK;
  };

  auto issue_load = [&](Stage &st) __attribute__((always_inline)) {
    st.sc4 = *reinterpret_cast<const v4f *>(a_aff + a_cc);
    st.sh4 = *reinterpret_cast<const v4f *>(a_aff + a_C + a_cc);
    st.floor_v = a_floor;
    st.okm = a_okm;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) st.ra[rr] = *reinterpret_cast<const v4f *>(a_src + a_off[rr]);
#pragma unroll
    for (int rr = 0; rr < BR; ++rr) st.rb[rr] = *reinterpret_cast<const v4f *>(a_wb + rr * 1024);
  };

  // CoordNet k-step (the last one): "channel" kk is tap kk of the |sin(lat)| plane
  auto load_coord_step = [&](int s, Stage &st) __attribute__((always_inline)) {
    st.sc4 = v4f{1.f, 1.f, 1.f, 1.f};
    st.sh4 = v4f{0.f, 0.f, 0.f, 0.f};
    st.floor_v = -INFINITY;
    st.okm = r_ok;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) {
      float vv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int tap = cq * 4 + e;
        int dh, dw, ihc, iwc;
        tap_offset<MODE>(p.rate, tap < p.ntaps ? tap : 0, ph, pw, dh, dw);
        const bool ok = tap_coords(rr, dh, dw, ihc, iwc) & (tap < p.ntaps);
        const float cv = p.coord[ihc];
        vv[e] = ok ? cv : 0.f;
      }
      st.ra[rr] = v4f{vv[0], vv[1], vv[2], vv[3]};
    }
    const float *wb = wbase + (size_t)s * p.npad * BK;
#pragma unroll
    for (int rr = 0; rr < BR; ++rr) st.rb[rr] = *reinterpret_cast<const v4f *>(wb + rr * 1024);
  };

  auto store_step = [&](int buf, const Stage &st) __attribute__((always_inline)) {
    float *As = smem + buf * STAGE;
    float *Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) {
      v4f v = st.ra[rr];
      // slim.layer_norm + ReLU of the producer: y = max(x*inv*gamma + (beta - mean*inv*gamma), 0);
      // raw sources (the network input) use scale 1, shift 0, floor -inf: the identity.
      v.x = fmaxf(v.x * st.sc4.x + st.sh4.x, st.floor_v);
      v.y = fmaxf(v.y * st.sc4.y + st.sh4.y, st.floor_v);
      v.z = fmaxf(v.z * st.sc4.z + st.sh4.z, st.floor_v);
      v.w = fmaxf(v.w * st.sc4.w + st.sh4.w, st.floor_v);
      const bool ok = (st.okm >> rr) & 1u;  // zero padding / tile tails (applied AFTER the LayerNorm)
      v.x = ok ? v.x : 0.f;
      v.y = ok ? v.y : 0.f;
      v.z = ok ? v.z : 0.f;
      v.w = ok ? v.w : 0.f;
      *reinterpret_cast<v4f *>(As + (lr + 32 * rr) * LDS_STRIDE + cq * 4) = v;
    }
#pragma unroll
    for (int rr = 0; rr < BR; ++rr)
      *reinterpret_cast<v4f *>(Bs + (lr + 32 * rr) * LDS_STRIDE + cq * 4) = st.rb[rr];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow = wm * (MT * 32) + (lane & 31);
  const int brow = wn * (NT * 32) + (lane & 31);
  const int kh0 = (lane >> 5) * 16;

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float *As = smem + buf * STAGE;
    const float *Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v4f a[MT], bb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const v4f *>(As + (arow + i * 32) * LDS_STRIDE + kh0 + q * 4);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bb[j] = *reinterpret_cast<const v4f *>(Bs + (brow + j * 32) * LDS_STRIDE + kh0 + q * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, bb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, bb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, bb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, bb[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  // ---- main loop ---------------------------------------------------------------------------
  // Two k-steps of global loads are in flight (register stages stA/stB) while a third computes:
  //   top    : addresses (one mad per row) + global loads of step s+2
  //   middle : 16*MT*NT MFMAs of step s from LDS buffer s&1
  //   bottom : wait for step s+1's loads (issued one iteration ago), LayerNorm/ReLU/mask,
  //            ds_write to the other LDS buffer, barrier
  // The loop is unrolled by two so the stages are addressed statically; past the last step the
  // address generator re-issues the last step instead of branching around the loads.
  const int nreg = p.ntaps * p.cpt;  // regular k-steps; + 1 coord step when p.coord
  int tap = 0, chunk = 0, gstep = 0;
  auto next_addr = [&]() __attribute__((always_inline)) {
    if (gstep + 1 < nreg) {              // wave-uniform
      ++gstep;
      if (++chunk == p.cpt) {            // once per tap
        chunk = 0;
        new_tap(++tap);
      }
    }
    gen_addr(gstep, chunk * BK);
  };
  new_tap(0);
  gen_addr(0, 0);
  issue_load(stA);
  store_step(0, stA);
  next_addr();
  issue_load(stA);                       // step 1 in flight
  __syncthreads();
  for (int s = 0; s < nreg; s += 2) {
    next_addr();
    if (!(p.ablate & 1)) issue_load(stB);  // step s+2
    __builtin_amdgcn_sched_barrier(0);     // keep the loads above the MFMAs
    if (!(p.ablate & 2)) compute(0);       // step s
    __builtin_amdgcn_sched_barrier(0);     // the tail must not pull its vmcnt wait up
    store_step(1, stA);                    // step s+1
    __syncthreads();
    if (s + 1 >= nreg) break;
    next_addr();
    if (!(p.ablate & 1)) issue_load(stA);  // step s+3
    __builtin_amdgcn_sched_barrier(0);
    if (!(p.ablate & 2)) compute(1);       // step s+1
    __builtin_amdgcn_sched_barrier(0);
    store_step(0, stB);                    // step s+2
    __syncthreads();
  }
  if (p.coord != nullptr) {
    const int buf = nreg & 1;
    load_coord_step(nreg, stA);
    store_step(buf, stA);
    __syncthreads();
    compute(buf);
    __syncthreads();
  }

  // ---- epilogue: store + LayerNorm partial ------------------------------------------------
  // C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int col = lane & 31, rowq = 4 * (lane >> 5);
  float lsum = 0.f;
  float cnt = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = tile_m * BM + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + rowq;
      if (m >= mtot) continue;
      size_t opix;
      if (MODE == MODE_CONVT) {
        const int mh = m / p.Mw, mw = m - mh * p.Mw;
        opix = ((size_t)b * p.Hout + (2 * mh + ph)) * p.Wout + (2 * mw + pw);
      } else {
        opix = (size_t)b * mtot + m;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = tile_n * BN + wn * (NT * 32) + j * 32 + col;
        if (n >= p.Cout) continue;
        float v = acc[i][j][r];
        if (MODE == MODE_HEAD) v = tanhf(v + p.bias[n]);
        p.y[opix * p.Cout + n] = v;
        lsum += v;
        cnt += 1.f;
      }
    }
  }
  if (MODE == MODE_HEAD || p.stats == nullptr) return;

  // block mean, then M2 about the block mean (two-pass inside the block: the values
  // are still in registers), reduced in a fixed order.
  float *red = smem;  // all LDS reads of the main loop are behind the last barrier
  auto block_sum = [&](float v) __attribute__((always_inline)) -> float {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const float bsum = block_sum(lsum);
  const float bcnt = block_sum(cnt);
  const float bmean = bcnt > 0.f ? bsum / bcnt : 0.f;
  float lm2 = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = tile_m * BM + wm * (MT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + rowq;
      if (m >= mtot) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = tile_n * BN + wn * (NT * 32) + j * 32 + col;
        if (n >= p.Cout) continue;
        const float dlt = acc[i][j][r] - bmean;
        lm2 += dlt * dlt;
      }
    }
  }
  const float bm2 = block_sum(lm2);
  if (tid == 0) {
    const int nparts = gridDim.x * gridDim.y * p.nclass;
    const int part = (cls * gridDim.y + tile_n) * gridDim.x + tile_m;
    float *o = p.stats + ((size_t)b * nparts + part) * 4;
    o[0] = bcnt;
    o[1] = bmean;
    o[2] = bm2;
    o[3] = 0.f;
  }
}

// Merge the per-workgroup (count, mean, M2) partials of one sample in fp64 (Chan's
// pairwise update, fixed order => deterministic) and emit the LayerNorm affine of
// slim.layer_norm: scale = gamma * rsqrt(var + eps), shift = beta - mean * scale.
__global__ void __launch_bounds__(256)
ln_finish_kernel(const float *__restrict__ stats, int nparts, const float *__restrict__ gamma,
                 const float *__restrict__ beta, int C, float *__restrict__ aff) {
  __shared__ double s_n[256], s_mean[256], s_m2[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *st = stats + (size_t)b * nparts * 4;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int i = tid; i < nparts; i += 256) {
    const double nb = st[i * 4 + 0], mb = st[i * 4 + 1], m2b = st[i * 4 + 2];
    if (nb > 0.0) {
      const double nt = n + nb, dl = mb - mean;
      mean += dl * (nb / nt);
      m2 += m2b + dl * dl * (n * nb / nt);
      n = nt;
    }
  }
  s_n[tid] = n; s_mean[tid] = mean; s_m2[tid] = m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      const double na = s_n[tid], nb = s_n[tid + off];
      if (nb > 0.0) {
        const double nt = na + nb, dl = s_mean[tid + off] - s_mean[tid];
        s_mean[tid] += dl * (nb / nt);
        s_m2[tid] += s_m2[tid + off] + dl * dl * (na * nb / nt);
        s_n[tid] = nt;
      }
    }
    __syncthreads();
  }
  const double var = s_m2[0] / s_n[0];
  const double inv = 1.0 / sqrt(var + LN_EPS);
  const double mu = s_mean[0];
  float *o = aff + (size_t)b * 2 * C;
  for (int c = tid; c < C; c += 256) {
    const double sc = inv * (double)gamma[c];
    o[c] = (float)sc;
    o[C + c] = (float)((double)beta[c] - mu * sc);
  }
}

// ============================================================================================
// host: layer table, parameter packing, forward
// ============================================================================================
struct Layer {
  char name[16];
  int kind;  // MODE_*
  int cin, cout, has_coord, stride, rate;
  int in_h, in_w, out_h, out_w;
  int src0, src1;  // producer layer indices (-1 = net_input; src1 = -1: none)
  int c0, c1;
  int ntaps, cpt, ksteps, nclass, npad;
  size_t param_off, param_floats;  // floats
  size_t packed_off;               // floats: weights, then gamma, beta (or bias), then coord table
  size_t packed_w_floats;
  size_t gamma_off, beta_off, coord_off;  // floats inside the packed blob
  size_t raw_off, aff_off;                // bytes inside the workspace
};

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Net {
  std::vector<Layer> layers;
  size_t param_floats = 0, packed_floats = 0, ws_bytes = 0, stats_off = 0, stats_bytes = 0;
  size_t ident_off = 0;  // floats inside the packed blob: ones[in_channels] zeros[in_channels]
};

int build_net(const msi_net_desc *d, Net &net) {
  if (!d) return msi::fail(MSI_E_BADARG, "net: null descriptor");
  if (d->batch < 0 || d->height <= 0 || d->width <= 0 || d->in_channels <= 0 || d->num_outputs <= 0 ||
      d->ngf <= 0)
    return msi::fail(MSI_E_BADARG, "net: bad descriptor");
  if (d->height % 8 || d->width % 8)
    return msi::fail(MSI_E_UNSUPPORTED, "net: height and width must be multiples of 8 (got %dx%d)",
                     d->height, d->width);
  if (d->in_channels % 4 || d->ngf % 4)
    return msi::fail(MSI_E_UNSUPPORTED, "net: in_channels and ngf must be multiples of 4");
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
  size_t max_parts = 1;
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
    } else if (s.kind == MODE_CONVT) {
      L.out_h = sh * 2;
      L.out_w = sw * 2;
      L.ntaps = 4;
      L.nclass = 4;
    } else {
      L.out_h = sh;
      L.out_w = sw;
      L.ntaps = 1;
      L.nclass = 1;
    }
    if (L.c1 > 0 && L.c0 % BK)
      return msi::fail(MSI_E_UNSUPPORTED, "net: first skip operand of %s has %d channels (need a multiple of %d)",
                       s.name, L.c0, BK);
    if ((size_t)sh * sw * (size_t)(L.c0 > L.c1 ? L.c0 : L.c1) >= ((size_t)1 << 31))
      return msi::fail(MSI_E_UNSUPPORTED, "net: %s input exceeds 2^31 elements per sample", s.name);
    L.cpt = (int)((L.cin + BK - 1) / BK);
    L.ksteps = L.ntaps * L.cpt + (L.has_coord ? 1 : 0);
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
    L.packed_w_floats = (size_t)L.nclass * L.ksteps * L.npad * BK;
    L.gamma_off = L.packed_off + L.packed_w_floats;
    L.beta_off = L.gamma_off + round_up(L.cout, 4);
    L.coord_off = L.beta_off + round_up(L.cout, 4);
    koff = L.coord_off + (L.has_coord ? round_up(L.in_h, 4) : 0);
    koff = round_up(koff, 64);
    // workspace
    if (s.kind != MODE_HEAD) {
      L.raw_off = woff;
      woff += round_up((size_t)d->batch * L.out_h * L.out_w * L.cout * sizeof(float), 256);
      L.aff_off = woff;
      woff += round_up((size_t)d->batch * 2 * L.cout * sizeof(float), 256);
    } else {
      L.raw_off = (size_t)-1;
      L.aff_off = (size_t)-1;
    }
    // the smallest tile (64x64) bounds the number of LayerNorm partials
    const size_t mgrid = (s.kind == MODE_CONVT) ? (size_t)sh * sw : (size_t)L.out_h * L.out_w;
    const size_t parts = ((mgrid + 63) / 64) * ((L.cout + 63) / 64) * L.nclass;
    if (parts > max_parts) max_parts = parts;
  }
  net.param_floats = poff;
  net.ident_off = koff;
  koff = round_up(koff + (size_t)2 * d->in_channels, 64);
  net.packed_floats = koff;
  net.stats_off = woff;
  net.stats_bytes = round_up((size_t)d->batch * max_parts * 4 * sizeof(float), 256);
  net.ws_bytes = woff + net.stats_bytes;
  return MSI_OK;
}

template <int BM, int BN, int MODE>
int launch_conv_mode(const ConvParams &p, int batch, hipStream_t stream, int *nparts) {
  const int mtot = p.Mh * p.Mw;
  const dim3 grid((mtot + BM - 1) / BM, (p.Cout + BN - 1) / BN, batch * p.nclass);
  const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
  if (lds > 64 * 1024) {
    static thread_local bool done = false;
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_igemm_kernel<BM, BN, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return msi::fail(MSI_E_LAUNCH, "conv: %s", hipGetErrorString(e));
      done = true;
    }
  }
  *nparts = grid.x * grid.y * p.nclass;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, MODE>), grid, dim3(256), lds, stream, p);
  return msi::check_launch("conv_igemm");
}

template <int BM, int BN>
int launch_conv(const ConvParams &p, int batch, hipStream_t stream, int *nparts) {
  switch (p.mode) {
    case MODE_CONV: return launch_conv_mode<BM, BN, MODE_CONV>(p, batch, stream, nparts);
    case MODE_CONVT: return launch_conv_mode<BM, BN, MODE_CONVT>(p, batch, stream, nparts);
    default: return launch_conv_mode<BM, BN, MODE_HEAD>(p, batch, stream, nparts);
  }
}

// Tile choice: the largest tile that still gives every CU about two workgroups.
void choose_tile(int mtot, int cout, int zdim, int &bm, int &bn) {
  const int cands[3][2] = {{128, 128}, {128, 64}, {64, 64}};
  const long target = 2 * 256;
  long best_blocks = -1;
  bm = 64; bn = 64;
  for (auto &c : cands) {
    if (c[1] > 64 && cout <= 64) continue;  // do not pad N to 128 for 64-channel layers
    const long blocks = (long)((mtot + c[0] - 1) / c[0]) * ((cout + c[1] - 1) / c[1]) * zdim;
    if (blocks >= target) { bm = c[0]; bn = c[1]; return; }
    if (blocks > best_blocks) { best_blocks = blocks; bm = c[0]; bn = c[1]; }
  }
}

}  // namespace

extern "C" {

int msi_net_layer_info(const msi_net_desc *desc, int32_t layer, msi_layer_info *out) {
  Net net;
  int rc = build_net(desc, net);
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
  return MSI_OK;
}

size_t msi_net_param_floats(const msi_net_desc *desc) {
  Net net;
  return build_net(desc, net) ? 0 : net.param_floats;
}

size_t msi_net_packed_floats(const msi_net_desc *desc) {
  Net net;
  return build_net(desc, net) ? 0 : net.packed_floats;
}

size_t msi_net_workspace_bytes(const msi_net_desc *desc) {
  Net net;
  return build_net(desc, net) ? 0 : net.ws_bytes;
}

int msi_net_pack_weights_host(const msi_net_desc *desc, const float *params, float *packed) {
  Net net;
  int rc = build_net(desc, net);
  if (rc) return rc;
  MSI_REQUIRE(params && packed, "net_pack_weights: null pointer");
  memset(packed, 0, net.packed_floats * sizeof(float));
  for (int c = 0; c < desc->in_channels; ++c) packed[net.ident_off + c] = 1.0f;  // identity affine of the raw input
  for (const Layer &L : net.layers) {
    const float *w = params + L.param_off;
    float *o = packed + L.packed_off;
    const int cin_w = L.cin + L.has_coord;  // channel extent of the TF weight tensor
    for (int cls = 0; cls < L.nclass; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      for (int s = 0; s < L.ksteps; ++s) {
        const bool coord_step = L.has_coord && s == L.ksteps - 1;
        for (int n = 0; n < L.cout; ++n) {
          float *row = o + (((size_t)cls * L.ksteps + s) * L.npad + n) * BK;
          for (int kk = 0; kk < BK; ++kk) {
            int tap, c;
            if (coord_step) { tap = kk; c = L.cin; if (tap >= L.ntaps) continue; }
            else { tap = s / L.cpt; c = (s % L.cpt) * BK + kk; if (c >= L.cin) continue; }
            float v;
            if (L.kind == MODE_CONV) {            // [3,3,cin_w,cout]
              v = w[((size_t)tap * cin_w + c) * L.cout + n];
            } else if (L.kind == MODE_CONVT) {    // [4,4,cout,cin]
              const int th = tap >> 1, tw = tap & 1;
              const int kh = ph == 0 ? 1 + 2 * th : 2 - 2 * th;
              const int kw = pw == 0 ? 1 + 2 * tw : 2 - 2 * tw;
              v = w[(((size_t)kh * 4 + kw) * L.cout + n) * L.cin + c];
            } else {                              // [1,1,cin,cout]
              v = w[(size_t)c * L.cout + n];
            }
            row[kk] = v;
          }
        }
      }
    }
    const size_t wf = L.param_floats - (L.kind == MODE_HEAD ? (size_t)L.cout : (size_t)2 * L.cout);
    if (L.kind == MODE_HEAD) {
      memcpy(packed + L.gamma_off, w + wf, L.cout * sizeof(float));  // biases
    } else {
      memcpy(packed + L.gamma_off, w + wf, L.cout * sizeof(float));
      memcpy(packed + L.beta_off, w + wf + L.cout, L.cout * sizeof(float));
    }
    if (L.has_coord) {
      // nets.add_sph_coords (nets.py:260-265): abs(sin(np.linspace(-pi/2, pi/2, H))) in fp64 -> fp32
      const double PI = 3.14159265358979323846;
      const double start = -PI / 2.0, stop = PI / 2.0;
      const int h = L.in_h;
      const double step = h > 1 ? (stop - start) / (h - 1) : 0.0;
      for (int i = 0; i < h; ++i) {
        double a = (double)i * step + start;
        if (i == h - 1 && h > 1) a = stop;
        packed[L.coord_off + i] = (float)fabs(sin(a));
      }
    }
  }
  return MSI_OK;
}

int msi_net_forward_f32(const msi_net_desc *desc, const float *packed, const float *net_input,
                        float *pred, void *workspace, size_t workspace_bytes, msi_stream_t stream_) {
  Net net;
  int rc = build_net(desc, net);
  if (rc) return rc;
  MSI_REQUIRE(packed && net_input && pred && workspace, "net_forward: null pointer");
  if (workspace_bytes < net.ws_bytes)
    return msi::fail(MSI_E_WORKSPACE, "net_forward: workspace %zu B < required %zu B", workspace_bytes,
                     net.ws_bytes);
  if (desc->batch == 0) return MSI_OK;
  hipStream_t stream = msi::as_stream(stream_);
  char *ws = static_cast<char *>(workspace);
  float *stats = reinterpret_cast<float *>(ws + net.stats_off);

  for (int li = 0; li < MSI_NET_NUM_LAYERS; ++li) {
    const Layer &L = net.layers[li];
    ConvParams p;
    memset(&p, 0, sizeof(p));
    auto src_ptr = [&](int s) -> const float * {
      return s < 0 ? net_input : reinterpret_cast<const float *>(ws + net.layers[s].raw_off);
    };
    auto set_src = [&](int s, const float *&x, const float *&aff, int &bs, float &floor_v, int C) {
      if (s < 0) {  // the network input: no LayerNorm/ReLU in front of it
        x = net_input;
        aff = packed + net.ident_off;
        bs = 0;
        floor_v = -INFINITY;
      } else {
        x = reinterpret_cast<const float *>(ws + net.layers[s].raw_off);
        aff = reinterpret_cast<const float *>(ws + net.layers[s].aff_off);
        bs = 2 * C;
        floor_v = 0.0f;
      }
    };
    set_src(L.src0, p.x0, p.aff0, p.aff_bs0, p.floor0, L.c0);
    p.C0 = L.c0;
    // unused second source: alias the first (never selected since c0 < C0 always holds)
    p.x1 = p.x0; p.aff1 = p.aff0; p.aff_bs1 = p.aff_bs0; p.floor1 = p.floor0; p.C1 = 0;
    if (L.src1 >= 0) {
      set_src(L.src1, p.x1, p.aff1, p.aff_bs1, p.floor1, L.c1);
      p.C1 = L.c1;
    }
    p.wpk = packed + L.packed_off;
    p.coord = L.has_coord ? packed + L.coord_off : nullptr;
    p.bias = L.kind == MODE_HEAD ? packed + L.gamma_off : nullptr;
    p.y = L.kind == MODE_HEAD ? pred : reinterpret_cast<float *>(ws + L.raw_off);
    p.stats = L.kind == MODE_HEAD ? nullptr : stats;
    p.Hin = L.in_h; p.Win = L.in_w; p.Hout = L.out_h; p.Wout = L.out_w;
    p.Cout = L.cout; p.npad = L.npad;
    p.ntaps = L.ntaps; p.cpt = L.cpt; p.ksteps = L.ksteps;
    p.mode = L.kind; p.nclass = L.nclass;
    p.wrap = desc->coord_net ? 0 : 1;
    p.rate = L.rate;
    if (L.kind == MODE_CONV) {
      p.Mh = L.out_h; p.Mw = L.out_w; p.stride = L.stride;
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
    } else if (L.kind == MODE_CONVT) {
      p.Mh = L.in_h; p.Mw = L.in_w; p.stride = 1;
    } else {
      p.Mh = L.out_h; p.Mw = L.out_w; p.stride = 1;
    }
    int bm, bn, nparts = 0;
    choose_tile(p.Mh * p.Mw, L.cout, desc->batch * L.nclass, bm, bn);
    {  // debug knobs, read once
      static const char *abl = getenv("MSI_CONV_ABLATE");
      static const char *til = getenv("MSI_CONV_TILE");
      p.ablate = abl ? atoi(abl) : 0;
      if (til) { int a = 0, c = 0; if (sscanf(til, "%dx%d", &a, &c) == 2 && (c <= 64 || L.cout > 64)) { bm = a; bn = c; } }
    }
    if (bm == 128 && bn == 128) rc = launch_conv<128, 128>(p, desc->batch, stream, &nparts);
    else if (bm == 128 && bn == 64) rc = launch_conv<128, 64>(p, desc->batch, stream, &nparts);
    else rc = launch_conv<64, 64>(p, desc->batch, stream, &nparts);
    if (rc) return rc;
    if (L.kind != MODE_HEAD) {
      hipLaunchKernelGGL(ln_finish_kernel, dim3(desc->batch), dim3(256), 0, stream, stats, nparts,
                         packed + L.gamma_off, packed + L.beta_off, L.cout,
                         reinterpret_cast<float *>(ws + L.aff_off));
      rc = msi::check_launch("ln_finish");
      if (rc) return rc;
    }
  }
  return MSI_OK;
}

}  // extern "C"
