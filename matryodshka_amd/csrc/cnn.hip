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
#include "cnn_device.h"

namespace {

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
    // ... and the stride-2 layers of the six-product form (conv_halo8_s2_x3_kernel, r05): whole 8 x 16 tiles of the OUTPUT grid, same bit and grid rule
    if (Q.halo_x3 && !Q.halo_x2 && Q.halo_s2 && L.out_h % 8 == 0 && ((pl->opt[MSI_NET_OPT_X3_TILE8] >> li) & 1) &&
        ((long)(L.out_h / 8) * (L.out_w / 16) * ((L.cout + 63) / 64) * desc->batch >= 3L * pl->num_cus || ((pl->opt[MSI_NET_OPT_X3_TILE8] >> 30) & 1)))
      Q.x3_th8 = 1;
    if (Q.x3_th8) BM = 128;
    // rate-2 layers of the split kernels on row-parity tiles (conv_halo_x3_kernel<3, ...>, halo_row): the dilation along H becomes the tile's row stride --
    // a 6 x 20-pixel patch, the two-stage weight ring, three workgroups per CU (the plain rate-2 tile: 8 x 20, three stages, two)
    p.row_par = (Q.halo_x3 && !Q.halo_s2 && L.kind == MODE_CONV && L.rate == 2 && L.in_h % 8 == 0 && ((pl->opt[MSI_NET_OPT_X3_ROWPAR] >> li) & 1)) ? 1 : 0;
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
      // the 8 x 16-pixel tile of the six-product conv-transpose (convt_halo8_x3_kernel, r05): same rule as the stride-1 tile (X3_TILE8: bit li, >= 3 tiles per CU or bit 30)
      if (Q.halo_x3 && !Q.halo_x2 && !L.wrapt && L.in_h % 8 == 0 && ((pl->opt[MSI_NET_OPT_X3_TILE8] >> li) & 1) &&
          (2L * (L.in_h / 8) * (L.in_w / 16) * ((L.cout + 63) / 64) * desc->batch >= 3L * pl->num_cus || ((pl->opt[MSI_NET_OPT_X3_TILE8] >> 30) & 1))) {
        Q.x3_th8 = 1; BM = 128;
      }
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
    if (Q.halo_t && Q.x3_th8 && (size_t)(Q.nblocks - (p.split0 == 1 ? p.nb_main : 0)) * 2 * BM * BN * sizeof(float) > net.partial_bytes) {   // (slabs of the 8-row tile do not fit: 4-row tile)
      Q.x3_th8 = 0; BM = 64;
      plan_tiles(p, BM, BN, desc->batch, pl->num_cus, pl->opt[MSI_NET_OPT_TAILSPLIT], max_split, &Q.nblocks, &Q.nfix,
                 pl->opt[MSI_NET_OPT_UNIFORM_SPLIT], pl->opt[MSI_NET_OPT_SPLIT_OVERHEAD], false);
      Q.inlaunch = !pl->opt[MSI_NET_OPT_FIXUP_KERNEL] && Q.nfix <= CONV_SLOTS_PER_CU * pl->num_cus;
    }
    if (Q.halo_t && (size_t)(Q.nblocks - (p.split0 == 1 ? p.nb_main : 0)) * 2 * BM * BN * sizeof(float) > net.partial_bytes) {
      // two slabs per K-range do not fit the partial-accumulator workspace -> the tap kernel
      Q.halo_t = 0; Q.halo = 0; Q.halo_x3 = 0; Q.halo_x2 = 0; Q.x3_th8 = 0;
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

}  // namespace

#if defined(MSI_CONV_TIMING) || defined(MSI_DEBUG_STATS)
// tools/conv_timing.py: per-workgroup phase stamps of one layer's conv launch (debug builds only)
static unsigned long long *g_timing_buf = nullptr;
static int g_timing_layer = -1;
extern "C" int msi_debug_conv_occupancy(int lds_bytes) {
  return msi_cnn::debug_conv_occupancy(lds_bytes);
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
  pl->opt[MSI_NET_OPT_X3_ROWPAR] = 0x3ffff;  // (r05: every rate-2 layer of the split kernels)
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
    if (Q.halo_x3 && Q.x3_th8) snprintf(name, name_bytes, "convt_halo8_x3_kernel");
    else if (Q.halo_x3) snprintf(name, name_bytes, "convt_halo_x3_kernel<%d>", Q.halo_x2 ? 2 : 3);
    else snprintf(name, name_bytes, "convt_halo_kernel");
  } else if (Q.halo) {
    if (Q.halo_s2 && Q.halo_x3 && Q.x3_th8) snprintf(name, name_bytes, "conv_halo8_s2_x3_kernel<%d>", Q.halo_apply ? 1 : 0);
    else if (Q.halo_s2 && Q.halo_x3) snprintf(name, name_bytes, "conv_halo_s2_x3_kernel<%d, %d>", Q.halo_apply ? 1 : 0, Q.halo_x2 ? 2 : 3);
    else if (Q.halo_s2) snprintf(name, name_bytes, "conv_halo_s2_kernel<%d>", Q.halo_apply ? 1 : 0);
    else if (Q.x3_th8) snprintf(name, name_bytes, "conv_halo8_x3_kernel<%d, 3>", Q.halo_apply ? 1 : 0);
    else if (Q.halo_x3) snprintf(name, name_bytes, "conv_halo_x3_kernel<%d, %d, %d>", Q.p.row_par ? 3 : L.rate, Q.halo_apply ? 1 : 0, Q.halo_x2 ? 2 : 3);
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
  // ALL OR NOTHING: the windows of every layer as they were on entry; any error return writes back those of the layers touched so far, so a frame that
  // cannot be calibrated on (constant, non-finite) leaves the packed blob exactly as it found it.
  constexpr int NL = MSI_NET_NUM_LAYERS - 1;
  double orig[NL][LN_SCL_DOUBLES];
  hipError_t he = hipSuccess;
  for (int li = 0; li < NL && he == hipSuccess; ++li)
    he = hipMemcpyAsync(orig[li], packed + net.layers[li].lnscl_off, sizeof(orig[li]), hipMemcpyDeviceToHost, stream);
  if (he == hipSuccess) he = hipStreamSynchronize(stream);
  if (he != hipSuccess) return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he));
  int touched = 0;                         // layers 0 .. touched - 1 may hold new windows
  auto restore = [&]() {
    for (int l = 0; l < touched; ++l)
      (void)hipMemcpyAsync(packed + net.layers[l].lnscl_off, orig[l], sizeof(orig[l]), hipMemcpyHostToDevice, stream);
    (void)hipStreamSynchronize(stream);    // (orig is on this stack frame)
  };
  for (int li = 0; li < NL; ++li) {
    const Layer &L = net.layers[li];
    const double *scl = orig[li];
    int e0 = LN_S1_BITS - (int)lrint(log2(scl[0])), e = e0;
    bool settled = false;
    touched = li + 1;
    for (int it = 0; it < 24 && !settled; ++it) {
      const double ns[LN_SCL_DOUBLES] = {ldexp(1.0, LN_S1_BITS - e), ldexp(1.0, LN_S2_BITS - 2 * e), ldexp(1.0, -(LN_S1_BITS - e)), ldexp(1.0, -(LN_S2_BITS - 2 * e))};
      he = hipMemcpyAsync(packed + L.lnscl_off, ns, sizeof(ns), hipMemcpyHostToDevice, stream);
      if (he == hipSuccess) he = hipStreamSynchronize(stream);   // (ns is on this stack frame)
      if (he != hipSuccess) { restore(); return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he)); }
      int rc = run_layers(plan, packed, net_input, nullptr, workspace, workspace_bytes, stream_, li + 1);
      if (rc) { restore(); return rc; }
      int word = 0;
      he = hipMemcpyAsync(sums.data(), ws + L.sums_off, nwords * sizeof(long long), hipMemcpyDeviceToHost, stream);
      if (he == hipSuccess) he = hipMemcpyAsync(&word, ws + net.err_off, sizeof(int), hipMemcpyDeviceToHost, stream);
      if (he == hipSuccess) he = hipStreamSynchronize(stream);
      if (he != hipSuccess) { restore(); return msi::fail(MSI_E_LAUNCH, "net_plan_calibrate: %s", hipGetErrorString(he)); }
      if (word & STATUS_LN_OVERFLOW) { e += 12; if (e > 120) break; continue; }   // a share left the window (or the data is not finite: the loop gives up at 2^120)
      // per sample: sum x^2 in units of 1 / S2; below ~1e6 sqrt(waves) units the variance is resolved to < 6 digits (ln_mean_inv's rule, restated on the host).
      // The window is centred on the samples it RESOLVES; a constant sample (a black frame in a batch) or one far below the others is left out instead of
      // dragging the window down 12 bits at a time until nothing fits -- a forward flags such a sample itself.
      double rmin = 1e300, rmax = 0.0, amin = 1e300, amax = 0.0;   // (resolved samples | every sample with a non-zero sum of squares)
      for (int b = 0; b < desc->batch; ++b) {
        double h2 = 0.0;
        for (int sh = 0; sh < LN_SHARDS; ++sh)
          h2 += (double)sums[((size_t)b * LN_SHARDS + sh) * LN_WORDS + 1];
        const bool under = h2 * h2 < LN_UNDERFLOW_UNITS_SQ * (L.ln_count / 1024.0 + 1.0);
        const double ms = h2 * ns[3] / L.ln_count;           // E[x^2]: the window has to hold the raw values themselves
        const double r = sqrt(ms > 0.0 ? ms : 0.0);
        if (r > 0.0) { amin = r < amin ? r : amin; amax = r > amax ? r : amax; }
        if (r > 0.0 && !under) { rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax; }
      }
      if (rmax > 0.0) {                                                              // something is resolved: centre on it (to within an octave), measure once more in the new unit
        const int ec = (int)lrint(0.5 * (log2(rmax) + log2(rmin)));
        if (ec > e + 1 || ec < e - 1) { e = ec; continue; }
        settled = true;
      } else if (amax > 0.0) {                                                       // seen but not resolved: go to where the coarse reading points, else lower
        const int ec = (int)lrint(0.5 * (log2(amax) + log2(amin)));
        if (ec > e + 1 || ec < e - 1) e = ec; else e -= 12;
        if (e < -120) break;
      } else { e -= 12; if (e < -120) break; }                                       // nothing at all in this window: look lower
    }
    if (!settled) {
      restore();
      if (layers_changed) *layers_changed = 0;
      return msi::fail(MSI_E_RANGE, "net_plan_calibrate: layer %s has no finite, non-constant raw output to centre a LayerNorm window on (non-finite input or weights, "
                       "or a constant frame?); the windows are unchanged", L.name);
    }
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
  rc = launch_ln_finish(desc->batch, stream, reinterpret_cast<const long long *>(ws + S.sums_off), 1.0 / S.ln_count,
                        reinterpret_cast<const double *>(packed + S.lnscl_off), reinterpret_cast<int *>(ws + net.err_off),
                        packed + S.gamma_off, packed + S.beta_off, S.cout, aff, bf16);
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
  // (see the kernel: XCD x takes tiles x, x + 8, ...; the layer groups of a tile are neighbours there)
  return launch_head_assemble(bf16, (unsigned)(((ntile + 7) / 8) * 8 * ng), lds, stream, q);
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
    int rc0 = launch_zero(ws + net.zero_off, n16, stream);
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
    // pointers of the sources whose LayerNorm the consumer applies while staging its patch, then the family's launch (cnn_device.h)
    auto raw_source = [&](int s, bool second, bool repoint) {   // source s is read RAW: its sums / gamma / beta (bf16 plans: also the raw fp16 buffer)
      const Layer &S = net.layers[s];
      if (!second) {
        if (repoint) p.x0 = ws + S.raw_off;
        p.ln_sums = reinterpret_cast<const long long *>(ws + S.sums_off); p.ln_gamma = packed + S.gamma_off; p.ln_beta = packed + S.beta_off;
      } else {
        if (repoint) p.x1 = ws + S.raw_off;
        p.ln_sums1 = reinterpret_cast<const long long *>(ws + S.sums_off); p.ln_gamma1 = packed + S.gamma_off; p.ln_beta1 = packed + S.beta_off;
      }
    };
    int rc;
    if (Q.halo_tb) {            // bf16 conv-transpose halo kernel: either source may be raw (fp16)
      if (p.halo_apply & 1) raw_source(L.src0, false, true);
      if (p.halo_apply & 2) raw_source(L.src1, true, true);
      rc = launch_bf16_halo(Q, p, L.rate, false, stream);
    } else if (Q.halo && bf16) {
      if (Q.halo_apply) raw_source(L.src0, false, true);   // the patch comes from the producer's RAW output
      rc = launch_bf16_halo(Q, p, L.rate, plan->opt[MSI_NET_OPT_BF16_WAVES] == 8, stream);
    } else if (Q.halo_t) {      // fp32 conv-transpose halo kernels (the fp32 raw buffer IS the source buffer)
      if (p.halo_apply & 1) raw_source(L.src0, false, false);
      if (p.halo_apply & 2) raw_source(L.src1, true, false);
      if (Q.halo_x3) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + (Q.halo_x2 ? L.x2_off : L.x3_off));
        rc = launch_x3(Q, p, L.rate, stream);
      } else {
        rc = launch_halo_f32(Q, p, L.rate, stream);
      }
    } else if (Q.halo) {
      if (Q.halo_apply) raw_source(L.src0, false, false);
      if (Q.halo_x3) {
        p.wpk_x3 = reinterpret_cast<const char *>(packed + (Q.halo_x2 ? L.x2_off : L.x3_off));
        rc = launch_x3(Q, p, L.rate, stream);
      } else {
        rc = launch_halo_f32(Q, p, L.rate, stream);
      }
    } else {
      rc = launch_igemm(Q, p, bf16, stream);
    }
    if (rc) return rc;
    // (bf16 fused tail: head_assemble_kernel normalises + rounds the head's source itself)
    const bool tail_src = bf16 && nlayers == MSI_NET_NUM_LAYERS - 1 && li == net.layers[MSI_NET_NUM_LAYERS - 1].src0;
    if (L.kind != MODE_HEAD && !Q.skip_apply && !tail_src) {
      const size_t per_sample = (size_t)L.out_h * L.out_w * L.cout;
      float *raw = reinterpret_cast<float *>(ws + L.raw_off), *aff = reinterpret_cast<float *>(ws + L.aff_off);
      const size_t lds = (size_t)2 * L.cout * sizeof(float);
      const long long *sums = reinterpret_cast<const long long *>(ws + L.sums_off);
      rc = launch_ln_apply(bf16, Q.ln_blocks, desc->batch, lds, stream, raw, sums, 1.0 / L.ln_count, scl_of(li), p.status, packed + L.gamma_off, packed + L.beta_off,
                           per_sample, L.cout, aff, bf16 ? reinterpret_cast<unsigned short *>(ws + L.act_off) : static_cast<unsigned short *>(nullptr));
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

