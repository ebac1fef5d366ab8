// Geometry-side kernels of the MSI infer->render path for gfx950 (HBM-bound):
//   K5 pre/deprocess, K1 ODS sphere sweep, K3 RGBA assembly,
//   K4 fused reprojection + wrap-around bilinear gather + over-composite.
//
// This file is compiled with -ffp-contract=off: the reference evaluates every
// elementwise TF op separately in fp32 (no FMA contraction), and project_ods'
// discriminant is ill-conditioned enough that a contracted
// multiply-add flips the `disc >= 0` branch on ~1% of far-plane pixels.  With the
// reference's operation order, IEEE sqrt/div (hipcc default) and the host-built
// trig tables, every branch below is bit-reproducible against the CPU oracle.
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "msi_common.h"

namespace {

// fp32 constants the reference folds from Python doubles (spherical.py:54-68,
// :222-223), rounded once on the host.
struct PixConsts {
  float pi, pi_over_w, u_den, wm1;          // u = ((theta + pi) - pi/W) / (2pi - 2pi/W) * (W-1)
  float half_pi, half_pi_over_h, v_den, hm1;  // v = ((phi + pi/2) - (pi/2)/H) / (pi - pi/H) * (H-1)
  float u_scale, v_scale;                   // (W-1) / u_den, (H-1) / v_den, rounded once from fp64 (fast tail)
};

PixConsts make_consts(int height, int width) {
  const double PI = 3.14159265358979323846;
  PixConsts k;
  k.pi = (float)PI;
  k.pi_over_w = (float)(PI / width);
  k.u_den = (float)(2 * PI - 2 * PI / width);
  k.wm1 = (float)(width - 1);
  k.half_pi = (float)(0.5 * PI);
  k.half_pi_over_h = (float)(0.5 * PI / height);
  k.v_den = (float)(PI - PI / height);
  k.hm1 = (float)(height - 1);
  k.u_scale = (float)((double)(width - 1) / (2 * PI - 2 * PI / width));
  k.v_scale = (float)((double)(height - 1) / (PI - PI / height));
  return k;
}

// tf.mod on int32 = floor-mod.  Pixel coordinates that come out of the angle formulas lie in
// [-1, n], so the index (a = x + n) is in [n-1, 2n]: one conditional subtract; anything else (only
// reachable through NaN/inf garbage) takes the generic integer-division path (~20 instructions,
// which used to be paid 4x per bilinear lookup).
__device__ __forceinline__ int floor_mod(int a, int n) {
  if (a >= 0 && a < 2 * n) return a >= n ? a - n : a;
  int m = a % n;
  return m < 0 ? m + n : m;
}

// Corner indices and area weights of sampling.resample (sampling.py:150-165,
// 187-190): weights from the UNWRAPPED corners, indices wrapped in both axes.
struct Taps {
  int x0, x1, y0, y1;
  float wa, wb, wc, wd;
};

__device__ __forceinline__ Taps make_taps(float x, float y, int width, int height) {
  Taps t;
  const float fx0 = floorf(x), fy0 = floorf(y);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const int x1 = x0 + 1, y1 = y0 + 1;
  const float dx0 = x - (float)x0, dy0 = y - (float)y0;
  const float dx1 = (float)x1 - x, dy1 = (float)y1 - y;
  t.x0 = floor_mod(x0 + width, width);
  t.y0 = floor_mod(y0 + height, height);
  t.x1 = floor_mod(x1 + width, width);
  t.y1 = floor_mod(y1 + height, height);
  t.wa = dy1 * dx1;
  t.wb = dy1 * dx0;
  t.wc = dy0 * dx1;
  t.wd = dy0 * dx0;
  return t;
}

// make_taps for coordinates that lie in [-1, n] by construction -- the sweep and the render derive them from angles
// (theta in [-pi, pi], phi clamped / in [-pi/2, pi/2]  =>  u in [-0.5, W-0.5], v in [-0.5, H-0.5]) -- so the floor-mod
// is one unsigned min per corner, nothing branches, and the corner positions are 24-bit pixel offsets (row * W + col,
// full-rate v_mul_u32_u24; the host checks H * W < 2^24) for 32-bit buffer addressing: the generic form above spent
// a quarter of these VALU-bound kernels in 64-bit address multiplies and exec-masked division fall-backs.
// Weights as above (unwrapped corners; (float)(int)floor(x) == floor(x) in this range); garbage inputs (NaN -> 0, inf)
// are clamped into the image instead of taking the reference's undefined int cast.
struct TapsR {
  unsigned oa, ob, oc, od;   // pixel offsets of (y0,x0) (y0,x1) (y1,x0) (y1,x1)
  float wa, wb, wc, wd;
};

__device__ __forceinline__ TapsR make_taps_ranged(float x, float y, int width, int height) {
  TapsR t;
  const float fx0 = floorf(x), fy0 = floorf(y);
  const float dx0 = x - fx0, dy0 = y - fy0;
  const float dx1 = (fx0 + 1.0f) - x, dy1 = (fy0 + 1.0f) - y;
  t.wa = dy1 * dx1;
  t.wb = dy1 * dx0;
  t.wc = dy0 * dx1;
  t.wd = dy0 * dx0;
  const int x0 = max(-1, min((int)fx0, width - 1)), y0 = max(-1, min((int)fy0, height - 1));
  const unsigned ax = (unsigned)(x0 + width), ay = (unsigned)(y0 + height);
  const unsigned x0w = min(ax, ax - (unsigned)width), y0w = min(ay, ay - (unsigned)height);   // -1 -> n-1
  const unsigned bx = (unsigned)(x0 + 1), by = (unsigned)(y0 + 1);
  const unsigned x1w = min(bx, bx - (unsigned)width), y1w = min(by, by - (unsigned)height);   // n -> 0
  const unsigned r0 = __umul24(y0w, (unsigned)width), r1 = __umul24(y1w, (unsigned)width);
  t.oa = r0 + x0w; t.ob = r0 + x1w; t.oc = r1 + x0w; t.od = r1 + x1w;
  return t;
}

__device__ __forceinline__ float blend4(const TapsR &t, float a, float b, float c, float d) {
  return ((t.wa * a + t.wb * b) + t.wc * c) + t.wd * d;
}

__device__ __forceinline__ float blend4(const Taps &t, float a, float b, float c, float d) {
  // tf.add_n([area_a*A, area_b*B, area_c*C, area_d*D]) summed in list order.
  return ((t.wa * a + t.wb * b) + t.wc * c) + t.wd * d;
}

// ---- the CONTINUOUS tail of the angle math ----------------------------------------------------------
// Everything that feeds a branch of the reference (|z| > |x|, sign(pz), disc >= 0: the quadratic of project_ods up
// to `disc`) is evaluated op for op in IEEE fp32 above / below.  What follows the branches -- the root, the direction,
// the two angles and the pixel coordinates -- is a continuous function of its inputs, so 1-ulp primitives
// (v_rcp_f32 + one Newton step, v_sqrt_f32, a degree-7 odd minimax atan, 1.3e-7 rad) move a sample by <= 2e-5 px,
// i.e. the bilinear result by ~1e-5 of the image range (tolerance 1e-3), and cost a third of the IEEE sequences
// (the sweep and the render are VALU-bound: ~470 / ~260 instructions per sample with libm atan2f and IEEE
// divide / sqrt, profiles/r01_*).  -DMSI_FAST_TAIL=0 restores the IEEE / libm tail.
#ifndef MSI_FAST_TAIL
#define MSI_FAST_TAIL 1
#endif

__device__ __forceinline__ float t_sqrt(float x) {
#if MSI_FAST_TAIL
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}

__device__ __forceinline__ float t_div(float a, float b) {
#if MSI_FAST_TAIL
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);   // one correction step: <= 1 ulp for normal operands
#else
  return a / b;
#endif
}

__device__ __forceinline__ float t_atan2(float y, float x) {
#if MSI_FAST_TAIL
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = mn * __builtin_amdgcn_rcpf(mx);
  a = (mx == 0.0f) ? 0.0f : a;                                // atan2(+-0, +-0) = +-0 / +-pi like libm
  const float s = a * a;
  float p = -0.0040545277297496796f;
  p = __builtin_fmaf(p, s, 0.021862812340259552f);
  p = __builtin_fmaf(p, s, -0.05591211095452309f);
  p = __builtin_fmaf(p, s, 0.09642180800437927f);
  p = __builtin_fmaf(p, s, -0.13908623158931732f);
  p = __builtin_fmaf(p, s, 0.19946564733982086f);
  p = __builtin_fmaf(p, s, -0.33329859375953674f);
  p = __builtin_fmaf(p, s, 0.9999993443489075f);
  float r = p * a;
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < 0.0f || (x == 0.0f && __builtin_signbitf(x))) ? 3.14159265358979324f - r : r;
  r = (mx != mx || mn != mn) ? __builtin_nanf("") : r;        // NaN in -> NaN out (the callers test for it)
  return __builtin_copysignf(r, y);
#else
  return atan2f(y, x);
#endif
}

// atan on [-1, 1] (the polynomial of t_atan2 without its range reduction) and the two angles of a point (x, y, z) with
// horizontal distance h = sqrt(x^2 + z^2) and norm R through half-angle forms, tan(a / 2) = sin a / (1 + cos a):
//   atan2(y, h) = 2 atan(y / (h + R))                       (|phi| <= pi/2: the argument is in [-1, 1] as it is)
//   atan2(z, x) = 2 atan(z / (h + x))            for x >= 0,
//               = copysign(pi, z) - 2 atan(z / (h - x))  for x < 0   (both arguments in [-1, 1])
// -- no min / max / swap range reduction, no quadrant fix-ups, one shared square root: ~14 instead of ~30 instructions
// per angle in the render kernel, which is bound by its VALU stream.  Same 1.3e-7 rad polynomial (doubled: 2.6e-7 rad =
// 3e-5 px at W = 640); atan2(+-0, +-0) = +-0 like libm's for (+0, +0).  Used by the render kernel (no data-dependent reference
// branch in its chain, finite inputs) and, since r04, by the sweep's continuous tail (ods_tail): there the angles feed floor() of
// the pixel coordinates, so a sample within 3e-5 px of an integer may take the neighbouring tap pair -- with weights (1 - eps, eps)
// against (eps', 1 - eps') on the same two texels, i.e. the bilinear value moves by <= 3e-5 of a texel difference (the resample is
// continuous across tap boundaries); the reference's branch-deciding values (disc, |z| > |x|, sign(pz)) are computed before this,
// IEEE op for op.  NaN inputs (disc < 0 pixels) produce NaN / garbage angles that ods_tail overrides as the reference does.
__device__ __forceinline__ float t_atan_unit(float a) {
  const float s = a * a;
  float p = -0.0040545277297496796f;
  p = __builtin_fmaf(p, s, 0.021862812340259552f);
  p = __builtin_fmaf(p, s, -0.05591211095452309f);
  p = __builtin_fmaf(p, s, 0.09642180800437927f);
  p = __builtin_fmaf(p, s, -0.13908623158931732f);
  p = __builtin_fmaf(p, s, 0.19946564733982086f);
  p = __builtin_fmaf(p, s, -0.33329859375953674f);
  p = __builtin_fmaf(p, s, 0.9999993443489075f);
  return p * a;
}

__device__ __forceinline__ void t_angles(float x, float y, float z, float R, float &theta_neg, float &phi) {
#if defined(MSI_RENDER_OLD_ANGLES)   // (A/B: the r03 form, two range-reduced atan2)
  theta_neg = -t_atan2(z, x);
  phi = t_atan2(y, t_sqrt(x * x + z * z));
#elif MSI_FAST_TAIL
  const float h = __builtin_amdgcn_sqrtf(x * x + z * z);
  const float den = fmaxf(h + fabsf(x), 1.17549435e-38f);               // (x = z = 0: 0 / tiny = 0)
  const float a2 = 2.0f * t_atan_unit(z * __builtin_amdgcn_rcpf(den));
  const float th = __builtin_signbitf(x) ? __builtin_copysignf(3.14159265358979324f, z) - a2 : a2;
  theta_neg = -th;
  phi = 2.0f * t_atan_unit(y * __builtin_amdgcn_rcpf(h + R));
#else
  theta_neg = -atan2f(z, x);
  phi = atan2f(y, sqrtf(x * x + z * z));
#endif
}

// ------------------------------------------------------------------------ K5
__global__ void preprocess_u8_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
                                     size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float x = (float)in[i] * (1.0f / 255.0f);
    out[i] = x * 2.0f - 1.0f;
  }
}

__global__ void preprocess_f32_kernel(const float *__restrict__ in, float *__restrict__ out,
                                      size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i] * 2.0f - 1.0f;
}

__global__ void deprocess_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, size_t n,
                                 int is_depth) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float x = in[i];
    if (!is_depth) x = (x + 1.0f) / 2.0f;
    float y = truncf(x * 255.5f);
    y = fminf(fmaxf(y, 0.0f), 255.0f);
    out[i] = (uint8_t)y;
  }
}

// The two images of a frame (ref + src in, rgb + depth out) in ONE launch each: at 2.7 ms per frame every ~5 us
// launch of these 2.5 MB kernels is 0.2 % of the frame.
__global__ void preprocess_u8_pair_kernel(const uint8_t *__restrict__ in0, const uint8_t *__restrict__ in1,
                                          float *__restrict__ out0, float *__restrict__ out1, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < 2 * n; i += stride) {
    const bool second = i >= n;
    const size_t k = second ? i - n : i;
    const float x = (float)(second ? in1 : in0)[k] * (1.0f / 255.0f);
    (second ? out1 : out0)[k] = x * 2.0f - 1.0f;
  }
}

__global__ void deprocess_pair_kernel(const float *__restrict__ rgb, const float *__restrict__ depth,
                                      uint8_t *__restrict__ out_rgb, uint8_t *__restrict__ out_depth, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < 2 * n; i += stride) {
    const bool second = i >= n;
    const size_t k = second ? i - n : i;
    float x = (second ? depth : rgb)[k];
    if (!second) x = (x + 1.0f) / 2.0f;
    float y = truncf(x * 255.5f);
    y = fminf(fmaxf(y, 0.0f), 255.0f);
    (second ? out_depth : out_rgb)[k] = (uint8_t)y;
  }
}

// [B,4,4] @ [B,4,4], one thread per output element, products summed k = 0..3 (no fma: this file is
// compiled with -ffp-contract=off), like a plain fp32 matmul loop.
__global__ void __launch_bounds__(256)
compose_poses_kernel(const float *__restrict__ lhs, const float *__restrict__ rhs, float *__restrict__ out, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * 16) return;
  const int b = i >> 4, r = (i >> 2) & 3, c = i & 3;
  const float *A = lhs + b * 16 + r * 4, *Bm = rhs + b * 16 + c;
  float acc = A[0] * Bm[0];
  acc = acc + A[1] * Bm[4];
  acc = acc + A[2] * Bm[8];
  acc = acc + A[3] * Bm[12];
  out[i] = acc;
}

// out0 = lhs0 @ rhs, out1 = lhs1 @ rhs in one launch: the two curr_pose of format_network_input (msi.py:1124-1125)
__global__ void __launch_bounds__(256)
compose_pose_pair_kernel(const float *__restrict__ lhs0, const float *__restrict__ lhs1, const float *__restrict__ rhs,
                         float *__restrict__ out0, float *__restrict__ out1, int batch) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * 32) return;
  const int which = t / (batch * 16), i = t - which * batch * 16;
  const int b = i >> 4, r = (i >> 2) & 3, c = i & 3;
  const float *A = (which ? lhs1 : lhs0) + b * 16 + r * 4, *Bm = rhs + b * 16 + c;
  float acc = A[0] * Bm[0];
  acc = acc + A[1] * Bm[4];
  acc = acc + A[2] * Bm[8];
  acc = acc + A[3] * Bm[12];
  (which ? out1 : out0)[i] = acc;
}

// fp32 <-> bf16 (round to nearest even; the values stored here are finite)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __builtin_bit_cast(float, (unsigned)h << 16);
}
__device__ __forceinline__ void store_elem(float *p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void store_elem(unsigned short *p, size_t i, float v) { p[i] = f32_to_bf16(v); }

// ------------------------------------------------------------------------ K1
// project_ods (spherical.py:181-229) in two parts.
// (1) ods_quad: everything up to the discriminant, op for op in IEEE fp32 (no contraction: this file is compiled with
//     -ffp-contract=off) -- it decides the reference's three branches (|z| > |x|, sign(pz), disc >= 0), and `disc` is
//     ill conditioned, so not one rounding may differ from the oracle here.
// (2) ods_tail: root, ray direction, angles, pixel coordinates -- continuous in (a, bq, f, px, pz, disc), evaluated
//     with the 1-ulp primitives above.  It is the only part that depends on `order`, so when both sources of the sweep
//     volume have the same pose (the test path: identity poses, msi.py:1125) they share (1).
struct OdsQuad {
  float f, px, pz, a, bq, disc, y;
  bool zlx;
};

__device__ __forceinline__ OdsQuad ods_quad(const float *__restrict__ P, float r, float depth, float csct, float st, float ssct) {
  OdsQuad q;
  // backproject_spherical (spherical.py:125-128)
  float x = depth * csct;
  float y = depth * st;
  float z = depth * ssct;
  // apply_pose (projector.py:275-291): pose @ [x,y,z,1], terms summed left to right
  const float px_ = ((P[0] * x + P[1] * y) + P[2] * z) + P[3] * 1.0f;
  const float py_ = ((P[4] * x + P[5] * y) + P[6] * z) + P[7] * 1.0f;
  const float pz_ = ((P[8] * x + P[9] * y) + P[10] * z) + P[11] * 1.0f;
  x = px_;
  y = py_;
  z = pz_;
  // project_ods (spherical.py:181-192)
  q.f = r * r - (x * x + z * z);
  q.zlx = fabsf(z) > fabsf(x);
  q.px = q.zlx ? x : z;
  q.pz = q.zlx ? z : x;
  const float pz2 = q.pz * q.pz;
  q.a = 1.0f + (q.px * q.px) / pz2;
  q.bq = ((-2.0f * q.f) * q.px) / pz2;
  const float c = q.f + (q.f * q.f) / pz2;
  q.disc = q.bq * q.bq - (4.0f * q.a) * c;
  q.y = y;
  return q;
}

// (spherical.py:195-229) -> pixel coordinates (u, v); (1, 1) where disc < 0 or NaN
__device__ __forceinline__ void ods_tail(const OdsQuad &q, float order, const PixConsts &K, float &u, float &v) {
  const float sgn = (q.pz > 0.0f) ? 1.0f : ((q.pz < 0.0f) ? -1.0f : q.pz);
  float s = ((-order) * sgn) * t_sqrt(q.disc);
  s = q.zlx ? s : -s;
  float dx = t_div(-q.bq + s, 2.0f * q.a);
  float dz = t_div(q.f - q.px * dx, q.pz);
  const float dxf = q.zlx ? -dx : -dz;
  const float dzf = q.zlx ? -dz : -dx;
  dx = dxf;
  dz = dzf;
#if MSI_FAST_TAIL && !defined(MSI_SWEEP_OLD_ANGLES)
  // (r04: the render kernel's half-angle forms -- one square root more, two range reductions less; where disc < 0 everything
  // here is NaN or garbage and the override below applies, as before)
  float theta, phi;
  t_angles(dx, q.y, dz, t_sqrt((dx * dx + dz * dz) + q.y * q.y), theta, phi);
#else
  const float theta = -t_atan2(dz, dx);
  float phi = t_atan2(q.y, t_sqrt(dx * dx + dz * dz));
#endif
  if (phi != phi) phi = 1.0f;
  phi = (phi <= K.half_pi) ? phi : K.half_pi;
  phi = (phi >= -K.half_pi) ? phi : -K.half_pi;
#if MSI_FAST_TAIL
  u = ((theta + K.pi) - K.pi_over_w) * K.u_scale;
  v = ((phi + K.half_pi) - K.half_pi_over_h) * K.v_scale;
#else
  u = (((theta + K.pi) - K.pi_over_w) / K.u_den) * K.wm1;
  v = (((phi + K.half_pi) - K.half_pi_over_h) / K.v_den) * K.hm1;
#endif
  if (!(q.disc >= 0.0f)) {
    u = 1.0f;
    v = 1.0f;
  }
}

// resample (sampling.py:135-197) of one RGB texel; `img` = buffer descriptor of one sample's [H,W,3] image.
// The four corners as BYTE offsets (pixel offset x 12, 24-bit multiply) + area weights: everything of a sample that does
// not depend on the image -- kept in registers while the frames of a batch that share (pose, baseline) are gathered.
typedef unsigned u32x3_g __attribute__((ext_vector_type(3)));
typedef float f32x3_g __attribute__((ext_vector_type(3)));
struct TapsB {
  unsigned oa, ob, oc, od;
  float wa, wb, wc, wd;
};
__device__ __forceinline__ TapsB make_taps_bytes(float u, float v, int width, int height) {
  const TapsR t = make_taps_ranged(u, v, width, height);
  TapsB r;
  r.oa = __umul24(t.oa, 12u); r.ob = __umul24(t.ob, 12u); r.oc = __umul24(t.oc, 12u); r.od = __umul24(t.od, 12u);
  r.wa = t.wa; r.wb = t.wb; r.wc = t.wc; r.wd = t.wd;
  return r;
}
__device__ __forceinline__ float blend4(const TapsB &t, float a, float b, float c, float d) {
  return ((t.wa * a + t.wb * b) + t.wc * c) + t.wd * d;   // tf.add_n order (sampling.py:187-190)
}
__device__ __forceinline__ void gather3(__amdgpu_buffer_rsrc_t img, const TapsB &t, float *out) {
  const f32x3_g a = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, t.oa, 0, 0));
#ifdef MSI_SWEEP_ABLATE_TAPS   // timing experiment only (wrong volume): two of the four corner loads
  const f32x3_g b = a;
  const f32x3_g c = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, t.oc, 0, 0));
  const f32x3_g d = c;
#else
  const f32x3_g b = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, t.ob, 0, 0));
  const f32x3_g c = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, t.oc, 0, 0));
  const f32x3_g d = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, t.od, 0, 0));
#endif
  out[0] = blend4(t, a.x, b.x, c.x, d.x);
  out[1] = blend4(t, a.y, b.y, c.y, d.y);
  out[2] = blend4(t, a.z, b.z, c.z, d.z);
}

// The 16-byte pieces of a wave's whole-pixel strip.  nt != 0 (block-uniform; the host sets it when the volume is larger than the 256-MB Infinity Cache): non-temporal
// stores -- a volume that cannot stay cached until conv1_1 reads it should not evict what can (r06, same-box A/B at configs[2]: sweep 1.07 -> 0.98 ms per 16 frames;
// configs[3] -1 %; at batch 1 the 157-MB volume stays cached and keeps plain stores).  -DMSI_SWEEP_ABLATE_STORE: timing experiment only (no stores at all).
__device__ __forceinline__ void sweep_store16(uint4 *dst, const uint4 &v, int nt) {
#ifdef MSI_SWEEP_ABLATE_STORE
  if (nt < 0) *dst = v;
#else
  if (nt) {
    __builtin_nontemporal_store(v.x, &dst->x); __builtin_nontemporal_store(v.y, &dst->y); __builtin_nontemporal_store(v.z, &dst->z); __builtin_nontemporal_store(v.w, &dst->w);
  } else {
    *dst = v;
  }
#endif
}

// One work item = (pixel, NS consecutive depths) for up to `bchunk` frames of the batch; depth is the fastest index so a
// wavefront's 64 lanes write 64 x NS consecutive 12-byte texels of the NHWC volume.  The source images (2.4 MB) stay L2-resident.
// OutT = float, or unsigned short = bf16 bits (the bf16 network input of BASELINE configs[2]).
// NS = samples (consecutive depths of one pixel) per thread: with two, the per-pixel work (trigonometry,
// index arithmetic) is shared and hipcc pairs the independent multiplies / adds of the two samples into
// v_pk_mul_f32 / v_pk_add_f32 (IEEE, same roundings as the scalar forms).
// NSRC = 1: one source (image0, pose0, order) -> channels [coff, coff + 3D).
// NSRC = 2: the whole double volume of format_network_input (msi.py:1124-1129): source 0 with order +1 into
// channels [0, 3D), source 1 with order -1 into [3D, 6D); the quadratic is shared when the two poses are equal.
// BATCH LOOP (round 4): the sample position (u, v), its four corners and weights depend on (pixel, depth, pose, baseline)
// only -- not on the image.  The reference recomputes them per batch element (projector.py:129-170 runs per sample); on
// the test path every frame has the identity pose and the camera file's baseline (data_loader.py:146-160), so a thread
// keeps the corners of its samples in registers (TapsB: ~230 of the ~276 VALU per sample) and walks the frames
// [blockIdx.z * bchunk, + bchunk): per frame a wave-uniform comparison of the 24 pose entries + baseline with those the
// previous frame's (bit patterns; scalar loads, scalar branch) decides between "gather with the held corners" and
// "recompute" -- the same arithmetic either way, so the volume is bit-identical to the frame-at-a-time form.
// LOOP = 0: one frame per thread (grid.z = batch, nothing to reuse: the compiler interleaves the corner arithmetic with
// the gathers and needs 88 instead of 119 registers -- five waves per SIMD; 61 vs 70 us at batch 1).
#ifndef MSI_SWEEP_WAVES
#define MSI_SWEEP_WAVES 4
#endif
template <typename OutT, int NS, int NSRC, int LOOP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LOOP ? MSI_SWEEP_WAVES : 1, 8)))
ods_sweep_kernel(const float *__restrict__ image0, const float *__restrict__ image1, const float *__restrict__ pose0,
                 const float *__restrict__ pose1, const float *__restrict__ intrinsics, const float *__restrict__ depths,
                 const float *__restrict__ trig, int batch, int height, int width, int nd,
                 float order, OutT *__restrict__ psv, int channels, int coff, PixConsts K, unsigned ng_magic, int coalesce,
                 int bchunk) {
  // grid = (ceil(W*(D/NS) / 256), H, ceil(B / bchunk)): 32-bit index math only (64-bit div/mod are emulated in ~100
  // VALU instructions each and used to dominate this kernel)
  const int ng = nd / NS;                       // depth groups per pixel (nd % NS == 0, checked on the host)
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= width * ng) return;
  unsigned jq = __umulhi((unsigned)idx, ng_magic);   // idx / ng by multiply-high (+ one correction), see cnn.hip udiv_magic
  if ((unsigned)idx - jq * (unsigned)ng >= (unsigned)ng) ++jq;
  const int j = (int)jq, d0 = (idx - j * ng) * NS;
  const int i = blockIdx.y;
  const int b_lo = LOOP ? blockIdx.z * bchunk : blockIdx.z, b_hi = LOOP ? min(batch, b_lo + bchunk) : b_lo + 1;

  const float cs = trig[j], ss = trig[width + j];
  const float ct = trig[2 * width + i], st = trig[2 * width + height + i];
  const int img_bytes = height * width * 12;
  const float csct = cs * ct, ssct = ss * ct;
  float depth[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) depth[q] = depths[d0 + q];

  // lane -> (pixel of the wave, depth group) for the whole-pixel store path
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned plq = __umulhi((unsigned)lane, ng_magic);      // lane / ng by multiply-high (+ one correction), like idx / ng above
  if ((unsigned)lane - plq * (unsigned)ng >= (unsigned)ng) ++plq;
  const int pl = (int)plq, dg = lane - pl * ng;
  constexpr int WAVE_ELEMS = 384 * NS;
  __shared__ __attribute__((aligned(16))) OutT s_out[4][WAVE_ELEMS];

  // bit k of eqmask: frame b_lo + k has bit for bit the poses and the baseline of frame b_lo + k - 1 (wave-uniform: scalar
  // loads and integer compares, all issued before the frame loop so that no frame waits on them)
  unsigned eqmask = 0;
  if (LOOP) {
    const unsigned *U0 = reinterpret_cast<const unsigned *>(pose0), *U1 = reinterpret_cast<const unsigned *>(NSRC == 2 ? pose1 : pose0);
    const unsigned *UI = reinterpret_cast<const unsigned *>(intrinsics);
    for (int b = b_lo + 1; b < b_hi; ++b) {
      bool eq = UI[(size_t)b * 9] == UI[(size_t)(b - 1) * 9];
#pragma unroll
      for (int k = 0; k < 12; ++k)
        eq = eq && (U0[(size_t)b * 16 + k] == U0[(size_t)(b - 1) * 16 + k]) && (NSRC == 1 || U1[(size_t)b * 16 + k] == U1[(size_t)(b - 1) * 16 + k]);
      eqmask |= (eq ? 1u : 0u) << (b - b_lo);
    }
    eqmask = __builtin_amdgcn_readfirstlane(eqmask);
  }
  TapsB taps[NSRC][NS];
  for (int b = b_lo; b < b_hi; ++b) {
    const float *P0 = pose0 + (size_t)b * 16;
    const float *P1 = NSRC == 2 ? pose1 + (size_t)b * 16 : P0;
    const bool reuse = (eqmask >> (b - b_lo)) & 1u;   // the corners in `taps` were computed for an identical frame
    const __amdgpu_buffer_rsrc_t img0 = __builtin_amdgcn_make_buffer_rsrc((void *)(image0 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t img1 = NSRC == 2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(image1 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000) : img0;
    float out[NSRC][NS][3];
    if (!reuse) {
      const float r = intrinsics[(size_t)b * 9];
      bool same = NSRC == 2;                      // both sources share the quadratic when their poses are equal
      if (NSRC == 2) {
#pragma unroll
        for (int k = 0; k < 12; ++k) same = same && (P0[k] == P1[k]);
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        float u, v;
        const OdsQuad q0 = ods_quad(P0, r, depth[q], csct, st, ssct);
        ods_tail(q0, NSRC == 2 ? 1.0f : order, K, u, v);
        taps[0][q] = make_taps_bytes(u, v, width, height);
        if (!LOOP) gather3(img0, taps[0][q], out[0][q]);      // (one frame per thread: request each sample as soon as its corners exist)
        if (NSRC == 2) {
          if (same) {
            ods_tail(q0, -1.0f, K, u, v);
          } else {
            const OdsQuad q1 = ods_quad(P1, r, depth[q], csct, st, ssct);
            ods_tail(q1, -1.0f, K, u, v);
          }
          taps[NSRC - 1][q] = make_taps_bytes(u, v, width, height);
          if (!LOOP) gather3(img1, taps[NSRC - 1][q], out[NSRC - 1][q]);
        }
      }
    }
    if (LOOP) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        gather3(img0, taps[0][q], out[0][q]);
        if (NSRC == 2) gather3(img1, taps[NSRC - 1][q], out[NSRC - 1][q]);
      }
    }
    const long p = ((long)b * height + i) * width + j;
    if (NSRC == 2 && coalesce) {
      // Whole-pixel stores: with both sources in one thread a wavefront owns 64 / ng complete pixels = 384 NS
      // CONTIGUOUS elements of the NHWC volume.  They are exchanged through a wave-private LDS strip and leave as
      // 16-byte-per-lane stores (3 fully coalesced instructions per wave instead of 12 strided dword stores per lane,
      // which kept the store path -- not the VALU -- the limiter of this kernel).  No block barrier: LDS operations of
      // one wave are performed in order (the next frame's strip writes follow this frame's strip reads).
      OutT *w = s_out[wave];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int e0 = pl * 6 * nd + sidx * 3 * nd + dg * NS * 3;     // a lane's 3 NS values per source are contiguous
        if constexpr (sizeof(OutT) == 2 && NS % 2 == 0) {
          // bf16, two values per instruction (v_cvt_pk_bf16_f32: round to nearest even, what f32_to_bf16 computes for the
          // finite values stored here) and dword LDS stores (e0 is even): 12 integer-rounding sequences + 12 ds_write_b16 per
          // lane and frame were a third of the per-frame instruction stream once the corners are reused
          unsigned *wd = reinterpret_cast<unsigned *>(w + e0);
          const float *v = &out[sidx][0][0];
#pragma unroll
          for (int k = 0; k < 3 * NS / 2; ++k) {
            unsigned pk;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[2 * k]), "v"(v[2 * k + 1]));
            wd[k] = pk;
          }
        } else {
#pragma unroll
          for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) store_elem(w, (size_t)(e0 + q * 3 + c), out[sidx][q][c]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const size_t first = (size_t)(p - pl) * channels;      // element offset of the wave's first pixel (lane 0's pixel)
      const uint4 *src = reinterpret_cast<const uint4 *>(w);
      uint4 *dst = reinterpret_cast<uint4 *>(psv + first);
      constexpr int NV = WAVE_ELEMS * (int)sizeof(OutT) / 16;
#pragma unroll
      for (int k = lane; k < NV; k += 64) sweep_store16(dst + k, src[k], coalesce >> 1);
      continue;
    }
#pragma unroll
    for (int sidx = 0; sidx < NSRC; ++sidx) {
      const size_t o = (size_t)p * channels + (NSRC == 2 ? sidx * 3 * nd : coff) + d0 * 3;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        store_elem(psv, o + q * 3 + 0, out[sidx][q][0]);
        store_elem(psv, o + q * 3 + 1, out[sidx][q][1]);
        store_elem(psv, o + q * 3 + 2, out[sidx][q][2]);
      }
    }
  }
}

// ---- K1 with an LDS-staged source patch (round 6) --------------------------------
// What bounds ods_sweep_kernel once the corners are reused across frames is neither bytes (1.07 x the algorithmic traffic) nor VALU but the
// SIXTEEN dependent 12-byte gathers per thread and frame through the vector L1's address path (~21 cycles per instruction and CU).  The
// footprint of those gathers is tiny: a block's 256 / ng pixels of one row sample, over all depths and both sources, a patch of a few rows
// x (pixels + 2 disparities) columns of each image.  This kernel stages that patch once per frame and block in LDS -- coalesced row segments,
// texels padded to 16 bytes -- and gathers with ds_read_b128 (two address registers per sample: the corners are base, base + 16, base + pitch,
// base + pitch + 16) instead of buffer_load_dwordx3.  The patch is found, not assumed: when a thread computes its corners it also takes the
// UNWRAPPED corner (x0, y0) relative to the block's own position (W - 1 - j0, i), wrapped into [-n/2, n/2) so that the seam is nothing special;
// a block-wide min / max gives the box, and a block whose box does not fit (polar rows, where the disparity grows like 1 / cos(lat); rows whose
// near planes are invalid and sample pixel (1, 1); rotated poses) keeps gathering from memory exactly as ods_sweep_kernel does -- a block-uniform
// choice made once per (pose, baseline), not per frame.  Same corners, same weights, same blend: the volume is bit-identical.
// Frames: patch buffers alternate (frame b in buffer b & 1), so ONE block barrier per frame orders both "patch b is visible" and "everybody is
// done with patch b - 1"; the next frame's texels are requested into registers before this frame's gathers.
// Occupancy is what this kernel is most sensitive to (r06, same-box A/Bs at configs[2] / [3], gpurun_out/r06_occ*.log, r06_w6.log): 384 texels and four waves per SIMD
// 0.98 / 5.97 ms; 256 texels -- ONE staged texel per thread and source instead of two -- 0.90 / 5.27; five waves per SIMD (91 VGPRs) 0.81 / 5.05-5.3; and, once the
// per-sample state was cut to a weight quadruple + one packed corner word (TapsC) and the depths are re-read where the corners are computed, SIX waves without a spill
// (80 VGPRs; 192 texels so that six blocks' LDS fit with fp32 strips: 24.6 KB) 0.74 / 4.85 ms.  Six waves WITH spills (56 bytes) had lost (0.93 / 6.4); 128 texels: more
// fallback blocks at configs[3] (5.2 ms).
#ifndef MSI_SWEEP_PREFETCH   // 1: the next frame's patch texels are requested into registers before this frame's gathers (tuning: 0 frees six registers)
#define MSI_SWEEP_PREFETCH 1
#endif
#ifndef MSI_SWEEP_PMAX
#define MSI_SWEEP_PMAX 192
#endif
#ifndef MSI_SWEEP_LDS_WAVES   // amdgpu_waves_per_eu of ods_sweep_lds_kernel (tuning)
#define MSI_SWEEP_LDS_WAVES 6
#endif
constexpr int SW_PMAX = MSI_SWEEP_PMAX;                 // texels per source and buffer (16 B each): 2 buffers x 2 sources x 3 KB
constexpr int SW_NST = (SW_PMAX + 255) / 256;   // texels a thread stages per source

struct TapsL {
  TapsB t;
  int x0, y0;                                // unwrapped, clamped corner (make_taps_ranged's x0 / y0)
};
// What ods_sweep_lds_kernel keeps per sample across frames: the weights and ONE word for the corners.  Blocks that read their patch from LDS need no memory offsets at
// all; fallback blocks rebuild the four byte offsets from corner a's (bits 0-27: H * W * 12 < 2^28) and two flags -- x1 wrapped to column 0 (bit 30), y1 wrapped to row 0
// (bit 31) -- with three adds per frame and sample.  Twelve registers less than four offsets per sample: the kernel fits five waves per SIMD with room to spare.
struct TapsC {
  unsigned oaf;
  float wa, wb, wc, wd;
};
__device__ __forceinline__ TapsC compact_taps(const TapsB &t) {
  TapsC c;
  c.oaf = t.oa | (t.ob < t.oa ? 0x40000000u : 0u) | (t.oc < t.oa ? 0x80000000u : 0u);   // (x1 = x0 + 1 unless it wrapped to 0: then ob < oa; likewise y1)
  c.wa = t.wa; c.wb = t.wb; c.wc = t.wc; c.wd = t.wd;
  return c;
}
__device__ __forceinline__ TapsB expand_taps(const TapsC &c, int width, int height) {
  TapsB t;
  t.oa = c.oaf & 0x0fffffffu;
  const unsigned dx = (c.oaf & 0x40000000u) ? (unsigned)(-(width - 1) * 12) : 12u;
  const unsigned dy = (c.oaf & 0x80000000u) ? (unsigned)(-(height - 1) * width * 12) : (unsigned)(width * 12);
  t.ob = t.oa + dx; t.oc = t.oa + dy; t.od = t.oa + dy + dx;
  t.wa = c.wa; t.wb = c.wb; t.wc = c.wc; t.wd = c.wd;
  return t;
}
__device__ __forceinline__ float blend4(const TapsC &t, float a, float b, float c, float d) {
  return ((t.wa * a + t.wb * b) + t.wc * c) + t.wd * d;
}
__device__ __forceinline__ TapsL make_taps_lds(float u, float v, int width, int height) {
  TapsL r;
  r.t = make_taps_bytes(u, v, width, height);
  r.x0 = max(-1, min((int)floorf(u), width - 1));
  r.y0 = max(-1, min((int)floorf(v), height - 1));
  return r;
}
__device__ __forceinline__ int centred(int d, int n) {          // d in (-2n, 2n) -> the representative of d mod n in [-n/2, n/2)
  const int half = n >> 1;
  d = d < -half ? d + n : d;
  d = d >= n - half ? d - n : d;
  d = d < -half ? d + n : d;
  d = d >= n - half ? d - n : d;
  return d;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

template <typename OutT, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MSI_SWEEP_LDS_WAVES, 8)))
ods_sweep_lds_kernel(const float *__restrict__ image0, const float *__restrict__ image1, const float *__restrict__ pose0,
                     const float *__restrict__ pose1, const float *__restrict__ intrinsics, const float *__restrict__ depths,
                     const float *__restrict__ trig, int batch, int height, int width, int nd,
                     OutT *__restrict__ psv, int channels, PixConsts K, unsigned ng_magic, int bchunk, int nt) {
  // grid as ods_sweep_kernel<OutT, NS, 2, 1> with the whole-pixel store path; the host guarantees 256 % ng == 0 and (W * ng) % 256 == 0,
  // so every block is full and owns 256 / ng complete pixels of row blockIdx.y
  const int ng = nd / NS;
  const int tid = threadIdx.x;
  const int idx = blockIdx.x * 256 + tid;
  unsigned jq = __umulhi((unsigned)idx, ng_magic);
  if ((unsigned)idx - jq * (unsigned)ng >= (unsigned)ng) ++jq;
  const int j = (int)jq, d0 = (idx - j * ng) * NS;
  const int i = blockIdx.y;
  const int b_lo = blockIdx.z * bchunk, b_hi = min(batch, b_lo + bchunk);
  unsigned j0q = __umulhi((unsigned)(blockIdx.x * 256), ng_magic);
  if ((unsigned)(blockIdx.x * 256) - j0q * (unsigned)ng >= (unsigned)ng) ++j0q;
  const int xc = width - 1 - (int)j0q;        // where the block's first pixel samples at infinite depth (identity pose): the centre the box is measured from

  const float cs = trig[j], ss = trig[width + j];
  const float ct = trig[2 * width + i], st = trig[2 * width + height + i];
  const int img_bytes = height * width * 12;
  const float csct = cs * ct, ssct = ss * ct;

  const int lane = tid & 63, wave = tid >> 6;
  unsigned plq = __umulhi((unsigned)lane, ng_magic);
  if ((unsigned)lane - plq * (unsigned)ng >= (unsigned)ng) ++plq;
  const int pl = (int)plq, dg = lane - pl * ng;
  constexpr int WAVE_ELEMS = 384 * NS;
  __shared__ __attribute__((aligned(16))) OutT s_out[4][WAVE_ELEMS];
  __shared__ __attribute__((aligned(16))) float4 s_patch[2][2][SW_PMAX];     // [buffer][source][texel]
  __shared__ int s_box[4];

  // (r06, measured and dropped: the frame-equality mask worked out by wave 0 alone and handed over through LDS -- 3 % slower; wave-PRIVATE patches without any block
  //  barrier in the frame loop -- 8 % slower at configs[2], 4 % at configs[3]: 2.3 x the staging loads, more fallback waves; gpurun_out/r06_eq.log, r06_wp.log)
  unsigned eqmask = 0;
  {
    const unsigned *U0 = reinterpret_cast<const unsigned *>(pose0), *U1 = reinterpret_cast<const unsigned *>(pose1);
    const unsigned *UI = reinterpret_cast<const unsigned *>(intrinsics);
    for (int b = b_lo + 1; b < b_hi; ++b) {
      bool eq = UI[(size_t)b * 9] == UI[(size_t)(b - 1) * 9];
#pragma unroll
      for (int k = 0; k < 12; ++k)
        eq = eq && (U0[(size_t)b * 16 + k] == U0[(size_t)(b - 1) * 16 + k]) && (U1[(size_t)b * 16 + k] == U1[(size_t)(b - 1) * 16 + k]);
      eqmask |= (eq ? 1u : 0u) << (b - b_lo);
    }
    eqmask = __builtin_amdgcn_readfirstlane(eqmask);
  }
  // (the wave's strip goes to its first pixel's row of the volume: one 64-bit product here, an add per frame)
  const size_t frame_elems = (size_t)height * width * channels;
  OutT *wdst = psv + (size_t)((((long)b_lo * height + i) * width + j) - pl) * channels;
  TapsC taps[2][NS];
  int lbase[2][NS];                            // float4 index of a sample's corner (y0, x0) in its source's patch
  unsigned goff[SW_NST];                       // byte offset in the image of the texels this thread stages (same for both sources), ~0u = none
  f32x3_g stg[2][SW_NST];
  int pitch = 0;
  bool lds_mode = false, pending = false;
  auto stage_load = [&](int b) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t im0 = __builtin_amdgcn_make_buffer_rsrc((void *)(image0 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t im1 = __builtin_amdgcn_make_buffer_rsrc((void *)(image1 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < SW_NST; ++k) {           // (offsets beyond the descriptor -- ~0u -- return zeros: texels nobody reads)
      stg[0][k] = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(im0, goff[k], 0, 0));
      stg[1][k] = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(im1, goff[k], 0, 0));
    }
  };
  for (int b = b_lo; b < b_hi; ++b) {
    const float *P0 = pose0 + (size_t)b * 16;
    const float *P1 = pose1 + (size_t)b * 16;
    const bool reuse = (eqmask >> (b - b_lo)) & 1u;
    float out[2][NS][3];
    if (!reuse) {
      const float r = intrinsics[(size_t)b * 9];
      bool same = true;
#pragma unroll
      for (int k = 0; k < 12; ++k) same = same && (P0[k] == P1[k]);
      int xr[2][NS], yr[2][NS];
      float depth[NS];                          // (read where the corners are computed, not held across the frame loop)
#pragma unroll
      for (int q = 0; q < NS; ++q) depth[q] = depths[d0 + q];
      int xmin = 0x7fffffff, xmax = -0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        float u, v;
        const OdsQuad q0 = ods_quad(P0, r, depth[q], csct, st, ssct);
        ods_tail(q0, 1.0f, K, u, v);
        TapsL t0 = make_taps_lds(u, v, width, height);
        if (same) {
          ods_tail(q0, -1.0f, K, u, v);
        } else {
          const OdsQuad q1 = ods_quad(P1, r, depth[q], csct, st, ssct);
          ods_tail(q1, -1.0f, K, u, v);
        }
        TapsL t1 = make_taps_lds(u, v, width, height);
        taps[0][q] = compact_taps(t0.t); taps[1][q] = compact_taps(t1.t);
        xr[0][q] = centred(t0.x0 - xc, width); yr[0][q] = centred(t0.y0 - i, height);
        xr[1][q] = centred(t1.x0 - xc, width); yr[1][q] = centred(t1.y0 - i, height);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          xmin = min(xmin, xr[s_][q]); xmax = max(xmax, xr[s_][q]);
          ymin = min(ymin, yr[s_][q]); ymax = max(ymax, yr[s_][q]);
        }
      }
      // the block's box (both sources share it: the two disparities mirror each other around the same centre)
      if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = -0x7fffffff; s_box[2] = 0x7fffffff; s_box[3] = -0x7fffffff; }
      __syncthreads();
      xmin = wave_min_i(xmin); xmax = wave_max_i(xmax); ymin = wave_min_i(ymin); ymax = wave_max_i(ymax);
      if (lane == 0) { atomicMin(&s_box[0], xmin); atomicMax(&s_box[1], xmax); atomicMin(&s_box[2], ymin); atomicMax(&s_box[3], ymax); }
      __syncthreads();
      xmin = s_box[0]; xmax = s_box[1]; ymin = s_box[2]; ymax = s_box[3];
      __syncthreads();                                              // (every wave has read the box before a later recompute -- possibly with no frame barrier in between: fallback blocks -- re-initialises it)
      const int pw = xmax - xmin + 2, ph = ymax - ymin + 2;          // (+ the x1 / y1 corners)
      pitch = ((pw + 7) & ~15) + 8;                                   // >= pw, = 8 mod 16: consecutive patch rows start half the banks apart
      lds_mode = ph * pitch <= SW_PMAX;
      pending = false;
      if (lds_mode) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          lbase[0][q] = (yr[0][q] - ymin) * pitch + (xr[0][q] - xmin);
          lbase[1][q] = (yr[1][q] - ymin) * pitch + (xr[1][q] - xmin);
        }
        const float rp = 1.0f / (float)pitch;
#pragma unroll
        for (int k = 0; k < SW_NST; ++k) {
          const int t = tid + k * 256;
          int yy = (int)(((float)t + 0.5f) * rp);                     // t / pitch (t < 512, pitch <= 384: exact up to the correction below)
          yy += (t - yy * pitch >= pitch) ? 1 : 0;
          yy -= (t - yy * pitch < 0) ? 1 : 0;
          const int xx = t - yy * pitch;
          const int row = floor_mod(i + ymin + yy + height, height), col = floor_mod(xc + xmin + xx + width, width);
          goff[k] = (t < SW_PMAX && yy < ph && xx < pw) ? __umul24((unsigned)(row * width + col), 12u) : 0xffffffffu;
        }
      }
    }
    if (lds_mode) {
      if (!pending) stage_load(b);
      float4 *pb0 = s_patch[b & 1][0], *pb1 = s_patch[b & 1][1];
#pragma unroll
      for (int k = 0; k < SW_NST; ++k) {
        const int t = tid + k * 256;
        if (t < SW_PMAX) {
          pb0[t] = float4{stg[0][k].x, stg[0][k].y, stg[0][k].z, 0.f};
          pb1[t] = float4{stg[1][k].x, stg[1][k].y, stg[1][k].z, 0.f};
        }
      }
      __syncthreads();                                                 // patch b visible; everybody is done with patch b - 1 (its buffer is free for b + 1)
      pending = MSI_SWEEP_PREFETCH && (b + 1 < b_hi) && ((eqmask >> (b + 1 - b_lo)) & 1u);
      if (pending) stage_load(b + 1);
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const float4 *pp = (s_ ? pb1 : pb0) + lbase[s_][q];
          const float4 a = pp[0], bb = pp[1], c = pp[pitch], d = pp[pitch + 1];
          const TapsC &t = taps[s_][q];
          out[s_][q][0] = blend4(t, a.x, bb.x, c.x, d.x);
          out[s_][q][1] = blend4(t, a.y, bb.y, c.y, d.y);
          out[s_][q][2] = blend4(t, a.z, bb.z, c.z, d.z);
        }
    } else {
      const __amdgpu_buffer_rsrc_t img0 = __builtin_amdgcn_make_buffer_rsrc((void *)(image0 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t img1 = __builtin_amdgcn_make_buffer_rsrc((void *)(image1 + (size_t)b * height * width * 3), 0, img_bytes, 0x00020000);
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        gather3(img0, expand_taps(taps[0][q], width, height), out[0][q]);
        gather3(img1, expand_taps(taps[1][q], width, height), out[1][q]);
      }
    }
    // whole-pixel stores through the wave's strip: exactly ods_sweep_kernel's
    OutT *w = s_out[wave];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int e0 = pl * 6 * nd + sidx * 3 * nd + dg * NS * 3;
      if constexpr (sizeof(OutT) == 2 && NS % 2 == 0) {
        unsigned *wd = reinterpret_cast<unsigned *>(w + e0);
        const float *v = &out[sidx][0][0];
#pragma unroll
        for (int k = 0; k < 3 * NS / 2; ++k) {
          unsigned pk;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(v[2 * k]), "v"(v[2 * k + 1]));
          wd[k] = pk;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int c = 0; c < 3; ++c) store_elem(w, (size_t)(e0 + q * 3 + c), out[sidx][q][c]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint4 *src = reinterpret_cast<const uint4 *>(w);
    uint4 *dst = reinterpret_cast<uint4 *>(wdst);          // the wave's first pixel of frame b
    wdst += frame_elems;
    constexpr int NV = WAVE_ELEMS * (int)sizeof(OutT) / 16;
#pragma unroll
    for (int k = lane; k < NV; k += 64) sweep_store16(dst + k, src[k], nt);
  }
}

// ------------------------------------------------------------------------ K3
// A block owns TP=32 consecutive pixels.  Phase 1 streams the contiguous PSV
// (32 x 6D floats) and pred (32 x 2D floats) tiles into LDS with 16-byte loads;
// phase 2 re-reads them transposed (row stride padded to an odd dword count, so
// the 32 lanes of a half-wave hit 32 different banks) and writes float4 texels
// of the D-major stack: 512 contiguous bytes per (half-wave, layer).
constexpr int K3_TP = 32;

// PSV_BF16: the PSV is the bf16 network input (converted to fp32 on its way into LDS).
// COLOR: which_color_pred of infer_msi (msi.py:119-275):
//   0 blend_psv    pred = [w | alpha]              rgb = w fg + (1-w) bg_psv                       (msi.py:130-147)
//   1 blend_bg     pred = [w | alpha | bg(3)]      rgb = w fg + (1-w) bg      (bg: raw tanh output, msi.py:177-188)
//   2 blend_bg_psv pred = [w | alpha | bw | bg(3)] rgb = bw (w fg + (1-w) bg_psv) + (1-bw) bg      (msi.py:223-242)
//   3 alpha_only   pred = [alpha]                  rgb = fg                                        (msi.py:258-268)
// with w, alpha, bw = (x + 1) / 2.  The channel counts of 1 and 2 are odd, so those modes place pred element
// by element (the 32-pixel tile is still one contiguous, 16-byte aligned run) and write the optional
// [B,H,W,D] outputs from phase 2.
enum { COLOR_BLEND_PSV = 0, COLOR_BLEND_BG = 1, COLOR_BLEND_BG_PSV = 2, COLOR_ALPHA_ONLY = 3 };

template <int PSV_BF16, int COLOR>
__global__ void __launch_bounds__(256)
assemble_kernel(const void *__restrict__ psv_, const float *__restrict__ pred,
                float4 *__restrict__ rgba, float *__restrict__ bw_out,
                float *__restrict__ al_out, float *__restrict__ bgw_out, long npix_total, int hw, int nd, int pred_scaled) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int c_psv = 6 * nd;
  const int c_pred = COLOR == COLOR_BLEND_PSV ? 2 * nd : (COLOR == COLOR_BLEND_BG ? 2 * nd + 3 : (COLOR == COLOR_BLEND_BG_PSV ? 3 * nd + 3 : nd));
  const int s_psv = c_psv + 1, s_pred = c_pred | 1;  // odd row strides
  float *l_psv = smem;
  float *l_pred = smem + K3_TP * s_psv;

  const long p0 = (long)blockIdx.x * K3_TP;
  const int npx = (int)((npix_total - p0) < K3_TP ? (npix_total - p0) : K3_TP);
  const int tid = threadIdx.x;

  if (PSV_BF16) {
    const uint4 *g = reinterpret_cast<const uint4 *>(static_cast<const unsigned short *>(psv_) + p0 * c_psv);
    const int nv = npx * c_psv / 8;
    for (int v = tid; v < nv; v += 256) {
      const uint4 q = g[v];
      const int e = v * 8;
      const int row = e / c_psv, col = e - row * c_psv;  // c_psv % 8 == 0 (D % 4 == 0): no row straddle
      float *dst = l_psv + row * s_psv + col;
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dst[2 * k] = bf16_to_f32((unsigned short)(w[k] & 0xffffu));
        dst[2 * k + 1] = bf16_to_f32((unsigned short)(w[k] >> 16));
      }
    }
  } else {
    const float4 *g = reinterpret_cast<const float4 *>(static_cast<const float *>(psv_) + p0 * c_psv);
    const int nv = npx * c_psv / 4;
    for (int v = tid; v < nv; v += 256) {
      const float4 q = g[v];
      const int e = v * 4;
      const int row = e / c_psv, col = e - row * c_psv;  // c_psv % 4 == 0: no row straddle
      float *dst = l_psv + row * s_psv + col;
      dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
    }
  }
  if (COLOR == COLOR_BLEND_PSV) {
    const float4 *g = reinterpret_cast<const float4 *>(pred + p0 * c_pred);
    const int nv = npx * c_pred / 4;
    for (int v = tid; v < nv; v += 256) {
      float4 q = g[v];
      if (!pred_scaled) {  // (x + 1) / 2 (msi.py:132-133); already applied on the high-res path
        q.x = (q.x + 1.0f) / 2.0f; q.y = (q.y + 1.0f) / 2.0f;
        q.z = (q.z + 1.0f) / 2.0f; q.w = (q.w + 1.0f) / 2.0f;
      }
      const int e = v * 4;
      const int row = e / c_pred, col = e - row * c_pred;
      float *dst = l_pred + row * s_pred + col;
      dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
      // optional extra outputs, [B,H,W,D] each (msi.py:281-287)
      if (col < nd) {
        if (bw_out) *reinterpret_cast<float4 *>(bw_out + (p0 + row) * nd + col) = q;
      } else {
        if (al_out) *reinterpret_cast<float4 *>(al_out + (p0 + row) * nd + (col - nd)) = q;
      }
    }
  } else {
    const float *g = pred + p0 * c_pred;
    const int n = npx * c_pred;
    const int nscaled = c_pred - ((COLOR == COLOR_BLEND_BG || COLOR == COLOR_BLEND_BG_PSV) ? 3 : 0);  // bg stays in [-1, 1]
    for (int e = tid; e < n; e += 256) {
      const int row = e / c_pred, col = e - row * c_pred;
      float x = g[e];
      if (col < nscaled) x = (x + 1.0f) / 2.0f;
      l_pred[row * s_pred + col] = x;
    }
  }
  __syncthreads();

  const int px = tid & (K3_TP - 1);
  if (px >= npx) return;
  const long p = p0 + px;
  const long b = p / hw;
  const long off = p - b * hw;
  const float *rp = l_psv + px * s_psv;
  const float *rq = l_pred + px * s_pred;
  for (int d = tid / K3_TP; d < nd; d += 256 / K3_TP) {
    const float *fg = rp + d * 3;
    const float *bg = rp + (nd + d) * 3;
    float4 o;
    if (COLOR == COLOR_BLEND_PSV) {
      const float w = rq[d];
      const float omw = 1.0f - w;
      o.x = w * fg[0] + omw * bg[0];
      o.y = w * fg[1] + omw * bg[1];
      o.z = w * fg[2] + omw * bg[2];
      o.w = rq[nd + d];
    } else if (COLOR == COLOR_BLEND_BG) {
      const float w = rq[d];
      const float omw = 1.0f - w;
      const float *pb = rq + 2 * nd;
      o.x = w * fg[0] + omw * pb[0];
      o.y = w * fg[1] + omw * pb[1];
      o.z = w * fg[2] + omw * pb[2];
      o.w = rq[nd + d];
      if (bw_out) bw_out[p * nd + d] = w;
      if (al_out) al_out[p * nd + d] = o.w;
    } else if (COLOR == COLOR_BLEND_BG_PSV) {
      const float w = rq[d], bw = rq[2 * nd + d];
      const float omw = 1.0f - w, ombw = 1.0f - bw;
      const float *pb = rq + 3 * nd;
      o.x = bw * (w * fg[0] + omw * bg[0]) + ombw * pb[0];
      o.y = bw * (w * fg[1] + omw * bg[1]) + ombw * pb[1];
      o.z = bw * (w * fg[2] + omw * bg[2]) + ombw * pb[2];
      o.w = rq[nd + d];
      if (bw_out) bw_out[p * nd + d] = w;
      if (al_out) al_out[p * nd + d] = o.w;
      if (bgw_out) bgw_out[p * nd + d] = bw;
    } else {
      o.x = fg[0]; o.y = fg[1]; o.z = fg[2];
      o.w = rq[d];
      if (al_out) al_out[p * nd + d] = o.w;
    }
    rgba[(b * nd + d) * hw + off] = o;
  }
}

// ------------------------------------------------------------------------ K4
// One thread per target pixel, 64x4-pixel tiles (a wavefront = 64 consecutive
// columns, so each tap row is a ~1 KiB coalesced float4 segment and vertically
// adjacent waves share tap rows in L1).  The ray/sphere quadratic's a and b do
// not depend on the layer, so per layer only sqrt, one divide and the two atan2
// remain; the running composite lives in registers and every texel of the
// D x H x W x 4 stack is fetched from HBM once.
enum RenderMode { RENDER_RGB = 1, RENDER_DEPTH = 2, RENDER_LAYERS = 4 };
constexpr int RENDER_SEGS = 4;   // layer segments per pixel (threads in y)
// ray model of the TARGET view: equirect (spherical.intersect_sphere, spherical.py:268-326),
// ODS eye (intersect_ods, :328-365) or the hard-coded perspective crop (intersect_perspective, :367-401)
enum RayModel { RAY_EQUIRECT = 0, RAY_ODS = 1, RAY_PERSPECTIVE = 2 };

// (i / len) of projector.py:242 per layer: a Python double division converted to an fp32 tensor constant.  Tabulated on the
// host (kernel argument, read with a scalar load: the layer index is wave-uniform) -- as an expression in the kernel it was an
// emulated fp64 division, ~25 half-rate instructions per (thread, layer), a fifth of the render kernel's VALU time.
constexpr int DEPTH_FRAC_MAX = 128;
struct DepthFrac { float f[DEPTH_FRAC_MAX]; };

struct RayParams {
  int out_h, out_w;        // target image size (== layer size except for RAY_PERSPECTIVE)
  float order;             // RAY_ODS: +1 left eye / -1 right eye
  float s0, sstep, t0, tstep;  // RAY_PERSPECTIVE: uv_grid = tf.linspace(-1+1/n, 1-1/n, n) (spherical.py:46-48)
};

template <int MODE, int RAY>
__global__ void __launch_bounds__(256)
render_kernel(const float4 *__restrict__ rgba, const float *__restrict__ pose_rt,
              const float *__restrict__ tgt_pos, const float *__restrict__ intrinsics,
              const float *__restrict__ depths, const float *__restrict__ trig, int batch, int height,
              int width, int nd, float *__restrict__ out_rgb, float *__restrict__ out_depth,
              float4 *__restrict__ out_layers, PixConsts K, RayParams R, DepthFrac F, int *__restrict__ status) {
  // block = 64 pixels of one target row x RENDER_SEGS layer segments: a pixel's D layers are split over RENDER_SEGS
  // threads (same lane, different waves), each compositing its contiguous range of layers from transparent black,
  //   C <- rgb a + C (1-a),   T <- T (1-a),
  // and the segments are combined back to front, out = C_3 + T_3 (C_2 + T_2 (C_1 + T_1 C_0)) (the over operator is
  // associative; rounding differs from the strictly sequential form by ~1e-7).  One thread per pixel left the chip with
  // 3 200 wavefronts for 8 192 wave slots and a 32-deep chain of dependent-latency gathers per thread.
  // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs and each XCD has its own L2, so XCD x takes the
  // x-th eighth of the (sample, row, 64-pixel block) sequence, in order: vertically adjacent target rows -- which share
  // their bilinear tap rows -- are gathered through the SAME L2 (with rows dealt round-robin every tap row was fetched
  // by two XCDs: 2.3x the stack's bytes at D = 64, batch 16; 1.43x at the BASELINE size)
  const unsigned gx = (unsigned)(R.out_w + 63) >> 6;
  const unsigned nblk = gx * (unsigned)R.out_h * (unsigned)batch, per = gridDim.x >> 3;
  const unsigned lin = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if (lin >= nblk) return;                        // (grid rounded up to a multiple of 8; whole workgroups leave)
  const unsigned rowb = lin / gx;
  const int j_raw = (int)(lin - rowb * gx) * 64 + threadIdx.x;
  const int b = (int)(rowb / (unsigned)R.out_h);
  const int i = (int)(rowb - (unsigned)b * (unsigned)R.out_h);
  // (a wavefront is one threadIdx.y: telling the compiler keeps the layer index -- and with it the per-layer buffer
  // descriptor -- in SGPRs; without it every gather sat in a readfirstlane waterfall loop, ~12 instructions each)
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const bool valid = j_raw < R.out_w;
  const int j = valid ? j_raw : R.out_w - 1;      // (lanes past the row end compute a duplicate and store nothing: no early return before the barrier)
  __shared__ float s_part[RENDER_SEGS][64][5];

  float rx, ry, rz, cx, cy, cz;
  if (RAY == RAY_PERSPECTIVE) {
    // spherical.py:381-392: rx = S*0.1, ry = T*0.05, rz = -0.05; centre (c0, c1, -c2)
    const float S = R.s0 + R.sstep * (float)j, T = R.t0 + R.tstep * (float)i;
    rx = S * 0.1f; ry = T * 0.05f; rz = -1.0f * 0.05f;
    const float *tp = tgt_pos + (size_t)b * 3;
    cx = tp[0]; cy = tp[1]; cz = -tp[2];
  } else {
    const float cs = trig[j], ss = trig[width + j];
    const float ct = trig[2 * width + i], st = trig[2 * width + height + i];
    if (RAY == RAY_ODS) {
      // spherical.py:349-358: ray (cosS cosT, sinT, -sinS cosT) from the viewing circle
      const float bl = intrinsics[(size_t)b * 9];
      rx = cs * ct; ry = st; rz = (-ss) * ct;
      cx = ((-ss) * bl) * R.order; cy = 0.0f; cz = ((-cs) * bl) * R.order;
    } else {
      // spherical.py:280-288: ray (cosS cosT, sinT, sinS cosT); origin = tgt_pos with x<->z swapped
      rx = cs * ct; ry = st; rz = ss * ct;
      const float *tp = tgt_pos + (size_t)b * 3;
      cx = tp[2]; cy = tp[1]; cz = tp[0];
    }
  }
  const float *P = pose_rt + (size_t)b * 16;
  {
    const float x = (P[0] * rx + P[1] * ry) + P[2] * rz;
    const float y = (P[4] * rx + P[5] * ry) + P[6] * rz;
    const float z = (P[8] * rx + P[9] * ry) + P[10] * rz;
    rx = x; ry = y; rz = z;
  }
  // ray origin through the full 4x4 (spherical.py:303-310 / transform_ray :70-94)
  {
    const float x = ((P[0] * cx + P[1] * cy) + P[2] * cz) + P[3] * 1.0f;
    const float y = ((P[4] * cx + P[5] * cy) + P[6] * cz) + P[7] * 1.0f;
    const float z = ((P[8] * cx + P[9] * cy) + P[10] * cz) + P[11] * 1.0f;
    cx = x; cy = y; cz = z;
  }
  const float qa = (rx * rx + ry * ry) + rz * rz;
  const float qb = 2.0f * ((rx * cx + ry * cy) + rz * cz);
  const float cc = (cx * cx + cy * cy) + cz * cz;
  const float qb2 = qb * qb;
  const float fa = 4.0f * qa, ta = 2.0f * qa;
  const float inv_ta = __builtin_amdgcn_rcpf(ta);

  const size_t hw = (size_t)height * width;             // source layer size
  const int layer_bytes = (int)(hw * 16);                // (H * W < 2^24, checked on the host)
  const size_t ohw = (size_t)R.out_h * R.out_w;          // target size
  const size_t pix = (size_t)i * R.out_w + j;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, od = 0.f, tr = 1.f;
  float qc_max = -1.0f;                                  // max over this thread's layers of |origin|^2 - radius^2 (>= 0: origin not inside that sphere)
  const int d_lo = (seg * nd) / RENDER_SEGS, d_hi = ((seg + 1) * nd) / RENDER_SEGS;

#pragma unroll 4
  for (int d = d_lo; d < d_hi; ++d) {
    const float radius = depths[d];
    const float qc = cc - radius * radius;
    qc_max = fmaxf(qc_max, qc);
    const float disc = qb2 - fa * qc;
    // disc >= 0 whenever the ray origin is inside the sphere (the documented domain; the host-side guard of the MSI class
    // enforces it for host inputs).  Outside it the reference takes sqrt of a negative number and casts NaN to int
    // (undefined); the clamp keeps device-side inputs finite and changes nothing inside the domain.
    // (no branch of the reference depends on these values: the whole chain is continuous -> 1-ulp primitives)
#if MSI_FAST_TAIL
    const float num = t_sqrt(fmaxf(disc, 0.0f)) - qb;           // t = num / ta with the pixel's 1 / ta (one correction step: <= 1 ulp)
    const float tq = num * inv_ta;
    const float t = __builtin_fmaf(__builtin_fmaf(-tq, ta, num), inv_ta, tq);
#else
    const float t = t_div(-qb + t_sqrt(fmaxf(disc, 0.0f)), ta);
#endif
    const float x = cx + t * rx;
    const float y = cy + t * ry;
    const float z = cz + t * rz;
    // project_spherical (spherical.py:243-246) + theta_phi_to_pixels (:54-68); the point lies on the layer's sphere: |P| = radius
    float theta, phi;
    t_angles(x, y, z, radius, theta, phi);
#if MSI_FAST_TAIL
    const float u = ((theta + K.pi) - K.pi_over_w) * K.u_scale;
    const float v = ((phi + K.half_pi) - K.half_pi_over_h) * K.v_scale;
#else
    const float u = (((theta + K.pi) - K.pi_over_w) / K.u_den) * K.wm1;
    const float v = (((phi + K.half_pi) - K.half_pi_over_h) / K.v_den) * K.hm1;
#endif

#ifdef MSI_RENDER_ABLATE_MATH   // timing experiment only (wrong pixels): identity warp, no per-layer angle math
    const TapsR tp4 = make_taps_ranged((float)j + 0.3f + 0.001f * radius, (float)i + 0.3f, width, height);
    (void)u; (void)v;
#else
    const TapsR tp4 = make_taps_ranged(u, v, width, height);
#endif
    // one descriptor per layer (scalar work): 32-bit texel offsets also for stacks beyond 2 GiB
    const __amdgpu_buffer_rsrc_t L = __builtin_amdgcn_make_buffer_rsrc((void *)(rgba + ((size_t)b * nd + d) * hw), 0, layer_bytes, 0x00020000);
    typedef unsigned u32x4_g __attribute__((ext_vector_type(4)));
    const float4 A = __builtin_bit_cast(float4, (u32x4_g)__builtin_amdgcn_raw_buffer_load_b128(L, tp4.oa << 4, 0, 0));
#if defined(MSI_RENDER_ABLATE_TAPS) && MSI_RENDER_ABLATE_TAPS == 3   // timing experiments only (wrong pixels): one / two of the four corner loads
    const float4 Bv = A, C = A, Dv = A;
#elif defined(MSI_RENDER_ABLATE_TAPS)
    const float4 Bv = A;
    const float4 C = __builtin_bit_cast(float4, (u32x4_g)__builtin_amdgcn_raw_buffer_load_b128(L, tp4.oc << 4, 0, 0));
    const float4 Dv = C;
#else
    const float4 Bv = __builtin_bit_cast(float4, (u32x4_g)__builtin_amdgcn_raw_buffer_load_b128(L, tp4.ob << 4, 0, 0));
    const float4 C = __builtin_bit_cast(float4, (u32x4_g)__builtin_amdgcn_raw_buffer_load_b128(L, tp4.oc << 4, 0, 0));
    const float4 Dv = __builtin_bit_cast(float4, (u32x4_g)__builtin_amdgcn_raw_buffer_load_b128(L, tp4.od << 4, 0, 0));
#endif
    const float al = blend4(tp4, A.w, Bv.w, C.w, Dv.w);
    if (MODE & RENDER_LAYERS) {
      float4 o;
      o.x = blend4(tp4, A.x, Bv.x, C.x, Dv.x);
      o.y = blend4(tp4, A.y, Bv.y, C.y, Dv.y);
      o.z = blend4(tp4, A.z, Bv.z, C.z, Dv.z);
      o.w = al;
      if (valid) out_layers[((size_t)d * batch + b) * ohw + pix] = o;
    }
    if (MODE & RENDER_RGB) {
      const float r = blend4(tp4, A.x, Bv.x, C.x, Dv.x);
      const float g = blend4(tp4, A.y, Bv.y, C.y, Dv.y);
      const float bl = blend4(tp4, A.z, Bv.z, C.z, Dv.z);
      if (d == 0) {  // projector.py:259-260: the farthest layer's alpha is ignored
        o0 = r; o1 = g; o2 = bl;
      } else {       // projector.py:262-263
        const float om = 1.0f - al;
        o0 = r * al + o0 * om;
        o1 = g * al + o1 * om;
        o2 = bl * al + o2 * om;
      }
    }
    if (MODE & RENDER_DEPTH) {
      if (d == 0) {
        od = 0.0f;   // projector.py:239-240
      } else {       // projector.py:242: (i / len) * alpha + output * (1 - alpha)
        const float frac = nd <= DEPTH_FRAC_MAX ? F.f[d] : (float)((double)d / (double)nd);
        od = frac * al + od * (1.0f - al);
      }
    }
    tr = tr * (1.0f - al);
  }
  // the ray origin of this sample is not inside every sphere it was intersected with: outside the reference's domain
  // (spherical.py:316-318 takes the square root of a negative number there); the clamp above kept the pixels finite, the
  // status word says so (one lane per wave; the origin is a property of the sample, so all lanes agree)
  // (fmaxf drops NaNs: a NaN pose / target position makes cc NaN and leaves qc_max at -1 -- tested separately)
  if (status != nullptr && threadIdx.x == 0 && (!(qc_max < 0.0f) || cc != cc)) atomicOr(status, MSI_RENDER_STATUS_ORIGIN_OUTSIDE);
  if (MODE & RENDER_LAYERS) return;            // (nothing to combine)
  s_part[seg][threadIdx.x][0] = o0; s_part[seg][threadIdx.x][1] = o1; s_part[seg][threadIdx.x][2] = o2;
  s_part[seg][threadIdx.x][3] = od; s_part[seg][threadIdx.x][4] = tr;
  __syncthreads();
  if (seg != 0 || !valid) return;
#pragma unroll
  for (int sg = 1; sg < RENDER_SEGS; ++sg) {    // back to front: segment 0 holds the farthest layers
    const float *q = s_part[sg][threadIdx.x];
    o0 = q[0] + q[4] * o0; o1 = q[1] + q[4] * o1; o2 = q[2] + q[4] * o2;
    od = q[3] + q[4] * od;
  }
  if (MODE & RENDER_RGB) {
    float *o = out_rgb + ((size_t)b * ohw + pix) * 3;
    o[0] = o0; o[1] = o1; o[2] = o2;
  }
  if (MODE & RENDER_DEPTH) {
    float *o = out_depth + ((size_t)b * ohw + pix) * 3;
    o[0] = od; o[1] = od; o[2] = od;
  }
}

// tf.image.resize(..., BILINEAR, align_corners=True) [TF-knowledge: resize_bilinear_op]:
// src = dst * (in-1)/(out-1); lower = floor(src), upper = min(ceil(src), in-1), lerp = src - lower;
// top = tl + (tr - tl)*xl; bottom = bl + (br - bl)*xl; out = top + (bottom - top)*yl.
// Used by the high-res re-render to upsample blend weights / alphas (test.py:319-325).
__global__ void resize_bilinear_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n,
                                       int in_h, int in_w, int c4, int out_h, int out_w, float sy, float sx) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; idx < n; idx += stride) {
    const int c = (int)(idx % c4);
    size_t r = idx / c4;
    const int x = (int)(r % out_w);
    r /= out_w;
    const int y = (int)(r % out_h);
    const size_t b = r / out_h;
    const float fy = (float)y * sy, fx = (float)x * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min((int)ceilf(fy), in_h - 1), x1 = min((int)ceilf(fx), in_w - 1);
    const float yl = fy - (float)y0, xl = fx - (float)x0;
    const float4 *base = in + b * (size_t)in_h * in_w * c4;
    const float4 tl = base[((size_t)y0 * in_w + x0) * c4 + c], tr = base[((size_t)y0 * in_w + x1) * c4 + c];
    const float4 bl = base[((size_t)y1 * in_w + x0) * c4 + c], br = base[((size_t)y1 * in_w + x1) * c4 + c];
    float4 o;
#define MSI_LERP2(f)                                                        \
    {                                                                       \
      const float top = tl.f + (tr.f - tl.f) * xl;                          \
      const float bot = bl.f + (br.f - bl.f) * xl;                          \
      o.f = top + (bot - top) * yl;                                         \
    }
    MSI_LERP2(x) MSI_LERP2(y) MSI_LERP2(z) MSI_LERP2(w)
#undef MSI_LERP2
    out[idx] = o;
  }
}

// ------------------------------------------------------------------------ PP path (config 5)
// pj.perspective_plane_sweep (projector.py:221-223) = sweep_one with spherical.uv_grid (:46-48),
// backproject_planar (:131-149), apply_pose, project_perspective (:248-266) and the SAME
// wrap-around sampler as the ODS sweep.  Faithful to the reference, the pose is applied twice:
// once by apply_pose (projector.py:155) and once inside project_perspective through
// intrinsics @ pose (spherical.py:258-259); the 3x3 intrinsics are zero-padded to 4x4
// (projector.py:145-148), so only rows 0..2 of the product are used.
// FAST (round 6): the form for full waves of complete pixels (64 % D == 0, (W * D) % 256 == 0, 16-byte-aligned runs; the host decides).  Same arithmetic, same bits;
// what changes is the plumbing the ODS sweep went through in rounds 1-2: (a) M = K4 @ pose -- 60 multiply-adds that depend on the FACE only -- is computed by twelve
// threads and broadcast through LDS instead of by every thread; (b) a corner is ONE 12-byte buffer load, not three dword loads behind 64-bit address arithmetic;
// (c) a wave's 64 / D complete pixels leave through a wave-private LDS strip as 16-byte-per-lane stores of whole 3 D-float runs instead of 3 dword stores at a
// 12-byte stride per lane; non-temporal when the volume exceeds the Infinity Cache (sweep_store16).  Measured at configs[4] (64 faces, two sweeps each): DESIGN.md section 4.
template <int FAST>
__global__ void __launch_bounds__(256)
pp_sweep_kernel(const float *__restrict__ image, const float *__restrict__ pose,
                const float *__restrict__ intrinsics, const float *__restrict__ depths, int batch,
                int height, int width, int nd, float s0, float sstep, float t0, float tstep,
                float *__restrict__ psv, int channels, int coff, unsigned nd_magic, int nt) {
  // grid = (ceil(W*D / 256), H, B): 32-bit index math only (64-bit div/mod are emulated in ~100
  // VALU instructions each and used to dominate this kernel)
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (!FAST && idx >= width * nd) return;
  int j, d;
  if (FAST) {
    unsigned jq = __umulhi((unsigned)idx, nd_magic);   // idx / nd by multiply-high (+ one correction)
    if ((unsigned)idx - jq * (unsigned)nd >= (unsigned)nd) ++jq;
    j = (int)jq; d = idx - j * nd;
  } else {
    j = idx / nd; d = idx - j * nd;
  }
  const int i = blockIdx.y, b = blockIdx.z;
  const long p = ((long)b * height + i) * width + j;
  const float S = s0 + sstep * (float)j, T = t0 + tstep * (float)i;
  const float depth = depths[d];
  const float *Kb = intrinsics + (size_t)b * 9;
  const float fx = Kb[0], fy = Kb[4], cx = Kb[2], cy = Kb[5];
  // backproject_planar (spherical.py:146-148): x = depth*S*cx/fx, y = depth*T*cy/fy, z = depth*1
  float x = ((depth * S) * cx) / fx;
  float y = ((depth * T) * cy) / fy;
  float z = depth * 1.0f;
  const float *P = pose + (size_t)b * 16;
  {  // apply_pose (projector.py:275-291)
    const float ax = ((P[0] * x + P[1] * y) + P[2] * z) + P[3] * 1.0f;
    const float ay = ((P[4] * x + P[5] * y) + P[6] * z) + P[7] * 1.0f;
    const float az = ((P[8] * x + P[9] * y) + P[10] * z) + P[11] * 1.0f;
    x = ax; y = ay; z = az;
  }
  // project_perspective: M = K4 @ pose, rows 0..2; the padded column contributes 0 * pose[3][c]
  float pr[3];
  __shared__ float s_m[12];
  if (FAST) {
    if (threadIdx.x < 12) {
      const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
      s_m[threadIdx.x] = ((Kb[r * 3 + 0] * P[c] + Kb[r * 3 + 1] * P[4 + c]) + Kb[r * 3 + 2] * P[8 + c]) + 0.0f * P[12 + c];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) pr[r] = ((s_m[r * 4 + 0] * x + s_m[r * 4 + 1] * y) + s_m[r * 4 + 2] * z) + s_m[r * 4 + 3] * 1.0f;
  } else {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      m[c] = ((Kb[r * 3 + 0] * P[c] + Kb[r * 3 + 1] * P[4 + c]) + Kb[r * 3 + 2] * P[8 + c]) + 0.0f * P[12 + c];
    pr[r] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3] * 1.0f;
  }
  }
  const float u = pr[0] / pr[2], v = pr[1] / pr[2];
  const Taps t = make_taps(u, v, width, height);
  if (FAST) {
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc((void *)(image + (size_t)b * height * width * 3), 0, height * width * 12, 0x00020000);
    const f32x3_g a = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, (unsigned)(t.y0 * width + t.x0) * 12u, 0, 0));
    const f32x3_g bq = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, (unsigned)(t.y0 * width + t.x1) * 12u, 0, 0));
    const f32x3_g c = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, (unsigned)(t.y1 * width + t.x0) * 12u, 0, 0));
    const f32x3_g dq = __builtin_bit_cast(f32x3_g, (u32x3_g)__builtin_amdgcn_raw_buffer_load_b96(img, (unsigned)(t.y1 * width + t.x1) * 12u, 0, 0));
    const float o0 = blend4(t, a.x, bq.x, c.x, dq.x), o1 = blend4(t, a.y, bq.y, c.y, dq.y), o2 = blend4(t, a.z, bq.z, c.z, dq.z);
    // whole-pixel runs through the wave's strip: lane = (pixel of the wave, depth); the wave's 64 / D pixels are consecutive, each owns 3 D contiguous floats at + coff
    __shared__ __attribute__((aligned(16))) float s_out[4][192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *w = s_out[wave];
    __builtin_amdgcn_wave_barrier();
    w[lane * 3 + 0] = o0; w[lane * 3 + 1] = o1; w[lane * 3 + 2] = o2;      // (lane = pl * D + d: the strip IS the pixels' runs back to back)
    __builtin_amdgcn_wave_barrier();
    if (lane < 48) {
      const int vpp = (3 * nd) >> 2;                                         // 16-byte vectors per pixel run
      unsigned plq = __umulhi((unsigned)lane, 0xffffffffu / (unsigned)vpp + 1u);
      if ((unsigned)lane - plq * (unsigned)vpp >= (unsigned)vpp) --plq;      // (lane < 48, vpp >= 3: the estimate is exact or one too large)
      const int pl = (int)plq, k = lane - pl * vpp;
      const long pw0 = ((long)b * height + i) * width + (long)((blockIdx.x * 256 + wave * 64) / nd);
      uint4 *dst = reinterpret_cast<uint4 *>(psv + (size_t)(pw0 + pl) * channels + coff) + k;
      sweep_store16(dst, reinterpret_cast<const uint4 *>(w)[lane], nt);
    }
    return;
  }
  const float *img = image + (size_t)b * height * width * 3;
  const float *pa = img + ((size_t)t.y0 * width + t.x0) * 3;
  const float *pb = img + ((size_t)t.y0 * width + t.x1) * 3;
  const float *pc = img + ((size_t)t.y1 * width + t.x0) * 3;
  const float *pd = img + ((size_t)t.y1 * width + t.x1) * 3;
  float *o = psv + (size_t)p * channels + coff + d * 3;
  o[0] = blend4(t, pa[0], pb[0], pc[0], pd[0]);
  o[1] = blend4(t, pa[1], pb[1], pc[1], pd[1]);
  o[2] = blend4(t, pa[2], pb[2], pc[2], pd[2]);
}

// MSI.mpi_render_view (msi.py:527-548): pj.projective_forward_homography (projector.py:343-373) ->
// homography.planar_transform (homography.py:120-157: inv_homography :35-58, transform_points
// :60-80, normalize_homogeneous :82-94, divide_safe :30-33) -> sampling.bilinear_wrapper =
// tf.contrib.resampler (zero padding) -> pj.over_composite, fused.  The per-(layer, sample)
// inverse homographies (a few dozen flops each) are computed once per workgroup into LDS.
constexpr int MPI_MAX_PLANES = 128;

__device__ __forceinline__ float4 fetch_or_zero(const float4 *L, int x, int y, int width, int height) {
  if (x >= 0 && y >= 0 && x < width && y < height) return L[(size_t)y * width + x];
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256)
mpi_render_kernel(const float4 *__restrict__ rgba, const float *__restrict__ tgt_pose,
                  const float *__restrict__ intrinsics, const float *__restrict__ intrinsics_inv,
                  const float *__restrict__ depths, int batch, int height, int width, int nd,
                  float *__restrict__ out_rgb) {
  __shared__ float hom[MPI_MAX_PLANES][9];
  const int b = blockIdx.z;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if (tid < nd) {
    const float *P = tgt_pose + (size_t)b * 16;
    const float *Ks = intrinsics + (size_t)b * 9, *Ki = intrinsics_inv + (size_t)b * 9;
    float rt[3][3], t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rt[r][c] = P[c * 4 + r];  // rot_t = transpose(pose[:3,:3])
      t[r] = P[r * 4 + 3];
    }
    const float a = -depths[tid];
    // n_hat = [0,0,1]: n_hat @ rot_t = row 2 of rot_t
    const float nrt_t = (rt[2][0] * t[0] + rt[2][1] * t[1]) + rt[2][2] * t[2];
    float den = a - nrt_t;
    den += 1e-8f * (den == 0.0f ? 1.0f : 0.0f);  // divide_safe
    float m1[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float q = (rt[r][0] * t[0] + rt[r][1] * t[1]) + rt[r][2] * t[2];  // (rot_t @ t)[r]
#pragma unroll
      for (int c = 0; c < 3; ++c) m1[r][c] = rt[r][c] + (q * rt[2][c]) / den;
    }
    float m2[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        m2[r][c] = (Ks[r * 3 + 0] * m1[0][c] + Ks[r * 3 + 1] * m1[1][c]) + Ks[r * 3 + 2] * m1[2][c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        hom[tid][r * 3 + c] = (m2[r][0] * Ki[0 * 3 + c] + m2[r][1] * Ki[1 * 3 + c]) + m2[r][2] * Ki[2 * 3 + c];
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int i = blockIdx.y * 4 + threadIdx.y;
  if (j >= width || i >= height) return;
  const float uu = (float)j, vv = (float)i;  // meshgrid_abs (projector.py:478-499)
  const size_t hw = (size_t)height * width;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  for (int d = 0; d < nd; ++d) {
    const float *h = hom[d];
    const float xs = (uu * h[0] + vv * h[1]) + 1.0f * h[2];
    const float ys = (uu * h[3] + vv * h[4]) + 1.0f * h[5];
    float ws = (uu * h[6] + vv * h[7]) + 1.0f * h[8];
    ws += 1e-8f * (ws == 0.0f ? 1.0f : 0.0f);
    const float x = xs / ws, y = ys / ws;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    // tf.contrib.resampler [TF-knowledge]: zero outside (-1, W) x (-1, H); missing corners are 0
    if (x > -1.0f && y > -1.0f && x < (float)width && y < (float)height) {
      const float fxf = floorf(x), fyf = floorf(y);
      const int fx = (int)fxf, fy = (int)fyf, cx = fx + 1, cy = fy + 1;
      const float dx = (float)cx - x, dy = (float)cy - y;
      const float4 *L = rgba + ((size_t)b * nd + d) * hw;
      const float4 a00 = fetch_or_zero(L, fx, fy, width, height), a11 = fetch_or_zero(L, cx, cy, width, height);
      const float4 a01 = fetch_or_zero(L, fx, cy, width, height), a10 = fetch_or_zero(L, cx, fy, width, height);
      const float w00 = dx * dy, w11 = (1.0f - dx) * (1.0f - dy), w01 = dx * (1.0f - dy), w10 = (1.0f - dx) * dy;
      v.x = ((w00 * a00.x + w11 * a11.x) + w01 * a01.x) + w10 * a10.x;
      v.y = ((w00 * a00.y + w11 * a11.y) + w01 * a01.y) + w10 * a10.y;
      v.z = ((w00 * a00.z + w11 * a11.z) + w01 * a01.z) + w10 * a10.z;
      v.w = ((w00 * a00.w + w11 * a11.w) + w01 * a01.w) + w10 * a10.w;
    }
    if (d == 0) {
      o0 = v.x; o1 = v.y; o2 = v.z;
    } else {
      const float om = 1.0f - v.w;
      o0 = v.x * v.w + o0 * om;
      o1 = v.y * v.w + o1 * om;
      o2 = v.z * v.w + o2 * om;
    }
  }
  float *o = out_rgb + ((size_t)b * hw + (size_t)i * width + j) * 3;
  o[0] = o0; o[1] = o1; o[2] = o2;
}

int grid_1d(size_t n) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest
  if (blocks == 0) blocks = 1;
  return (int)blocks;
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

size_t msi_trig_table_floats(int32_t height, int32_t width) {
  if (height <= 0 || width <= 0) return 0;
  return (size_t)2 * width + (size_t)2 * height;
}

static void linspace_f32(double start_d, double stop_d, int n, float *out) {
  // tf.linspace, TF 1.14: step = (stop - start) / (num - 1); v[i] = start + step * i (fp32)
  const float start = (float)start_d, stop = (float)stop_d;
  if (n == 1) { out[0] = start; return; }
  volatile float step = (stop - start) / (float)(n - 1);
  for (int i = 0; i < n; ++i) {
    volatile float prod = step * (float)i;  // volatile: one rounding per op on the host too
    out[i] = start + prod;
  }
}

int msi_build_trig_tables_host(int32_t height, int32_t width, float *out_host) {
  MSI_REQUIRE(height > 0 && width > 0 && out_host, "build_trig_tables: bad arguments");
  const double PI = 3.14159265358979323846;
  float *cs = out_host, *ss = out_host + width;
  float *ct = out_host + 2 * width, *st = ct + height;
  linspace_f32(-PI + PI / width, PI - PI / width, width, cs);
  linspace_f32(-PI / 2.0 + PI / (2 * height), PI / 2.0 - PI / (2 * height), height, ct);
  for (int j = 0; j < width; ++j) {
    const double a = (double)cs[j];
    ss[j] = (float)sin(a);
    cs[j] = (float)cos(a);
  }
  for (int i = 0; i < height; ++i) {
    const double a = (double)ct[i];
    st[i] = (float)sin(a);
    ct[i] = (float)cos(a);
  }
  return MSI_OK;
}

int msi_preprocess_u8_f32(const uint8_t *in, float *out, size_t n, msi_stream_t stream) {
  MSI_REQUIRE(in && out, "preprocess_u8: null pointer");
  if (n == 0) return MSI_OK;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_1d(n)), dim3(256), 0, msi::as_stream(stream),
                     in, out, n);
  return msi::check_launch("preprocess_u8");
}

int msi_preprocess_pair_u8_f32(const uint8_t *in0, const uint8_t *in1, float *out0, float *out1, size_t n,
                               msi_stream_t stream) {
  MSI_REQUIRE(in0 && in1 && out0 && out1, "preprocess_pair: null pointer");
  if (n == 0) return MSI_OK;
  hipLaunchKernelGGL(preprocess_u8_pair_kernel, dim3(grid_1d(2 * n)), dim3(256), 0, msi::as_stream(stream), in0, in1,
                     out0, out1, n);
  return msi::check_launch("preprocess_pair");
}

int msi_deprocess_pair_f32_u8(const float *rgb, const float *depth, uint8_t *out_rgb, uint8_t *out_depth, size_t n,
                              msi_stream_t stream) {
  MSI_REQUIRE(rgb && depth && out_rgb && out_depth, "deprocess_pair: null pointer");
  if (n == 0) return MSI_OK;
  hipLaunchKernelGGL(deprocess_pair_kernel, dim3(grid_1d(2 * n)), dim3(256), 0, msi::as_stream(stream), rgb, depth,
                     out_rgb, out_depth, n);
  return msi::check_launch("deprocess_pair");
}

int msi_preprocess_f32(const float *in, float *out, size_t n, msi_stream_t stream) {
  MSI_REQUIRE(in && out, "preprocess_f32: null pointer");
  if (n == 0) return MSI_OK;
  hipLaunchKernelGGL(preprocess_f32_kernel, dim3(grid_1d(n)), dim3(256), 0, msi::as_stream(stream),
                     in, out, n);
  return msi::check_launch("preprocess_f32");
}

int msi_deprocess_f32_u8(const float *in, uint8_t *out, size_t n, int32_t is_depth,
                         msi_stream_t stream) {
  MSI_REQUIRE(in && out, "deprocess: null pointer");
  if (n == 0) return MSI_OK;
  hipLaunchKernelGGL(deprocess_kernel, dim3(grid_1d(n)), dim3(256), 0, msi::as_stream(stream), in,
                     out, n, (int)is_depth);
  return msi::check_launch("deprocess");
}

int msi_compose_pose_pair_f32(const float *lhs0, const float *lhs1, const float *rhs, float *out0, float *out1,
                              int32_t batch, msi_stream_t stream) {
  MSI_REQUIRE(lhs0 && lhs1 && rhs && out0 && out1, "compose_pose_pair: null pointer");
  MSI_REQUIRE(batch >= 0, "compose_pose_pair: bad batch");
  if (batch == 0) return MSI_OK;
  hipLaunchKernelGGL(compose_pose_pair_kernel, dim3(grid_1d((size_t)batch * 32)), dim3(256), 0, msi::as_stream(stream),
                     lhs0, lhs1, rhs, out0, out1, (int)batch);
  return msi::check_launch("compose_pose_pair");
}

int msi_compose_poses_f32(const float *lhs, const float *rhs, float *out, int32_t batch,
                          msi_stream_t stream) {
  MSI_REQUIRE(lhs && rhs && out, "compose_poses: null pointer");
  MSI_REQUIRE(batch >= 0, "compose_poses: bad batch");
  if (batch == 0) return MSI_OK;
  hipLaunchKernelGGL(compose_poses_kernel, dim3(grid_1d((size_t)batch * 16)), dim3(256), 0, msi::as_stream(stream),
                     lhs, rhs, out, (int)batch);
  return msi::check_launch("compose_poses");
}

#ifndef MSI_SWEEP_NS_DEFAULT
#define MSI_SWEEP_NS_DEFAULT 2
#endif
#ifndef MSI_SWEEP_BCHUNK
#define MSI_SWEEP_BCHUNK 16
#endif
#ifndef MSI_SWEEP_NT   // -1 never / 0 by volume size / 1 always (A/B builds)
#define MSI_SWEEP_NT 0
#endif
#ifndef MSI_SWEEP_LDS_MIN_BATCH   // smallest batch that takes ods_sweep_lds_kernel (tuning: -DMSI_SWEEP_LDS_MIN_BATCH=1 / a huge value)
#define MSI_SWEEP_LDS_MIN_BATCH 2
#endif
static int sweep_common(const float *image, const float *image1, const float *pose, const float *pose1,
                        const float *intrinsics,
                        const float *depths, const float *trig, int32_t batch,
                        int32_t height, int32_t width, int32_t num_depths, int32_t order,
                        void *psv, int psv_bf16, int32_t psv_channels, int32_t channel_offset,
                        msi_stream_t stream) {
  const bool pair = image1 != nullptr;
  MSI_REQUIRE(image && pose && intrinsics && depths && trig && psv && (!pair || pose1), "ods_sphere_sweep: null pointer");
  MSI_REQUIRE(batch >= 0 && height > 0 && width > 0 && num_depths > 0, "ods_sphere_sweep: bad dims");
  MSI_REQUIRE(order == 1 || order == -1, "ods_sphere_sweep: order must be +1 or -1");
  MSI_REQUIRE(channel_offset >= 0 && channel_offset + (pair ? 6 : 3) * num_depths <= psv_channels,
              "ods_sphere_sweep: channel window [%d,%d) outside %d channels", channel_offset,
              channel_offset + (pair ? 6 : 3) * num_depths, psv_channels);
  if (batch == 0) return MSI_OK;
  MSI_REQUIRE((long)width * num_depths < 2147483647L && height <= 65535 && batch <= 65535 && (long)height * width < (1L << 24),
              "ods_sphere_sweep: problem too large (24-bit pixel offsets: H * W < 2^24)");
  // NS depths per thread (bit-identical results for every NS; -DMSI_SWEEP_NS_DEFAULT=1/2/4 at build time)
  int ns = MSI_SWEEP_NS_DEFAULT;
  while (num_depths % ns != 0) ns >>= 1;
  // frames per thread (see the kernel's BATCH LOOP): up to MSI_SWEEP_BCHUNK consecutive frames share a thread's sample
  // corners when their poses / baselines agree; chunks keep grid.z >= 1 and every chunk but the last full
  const int bchunk = batch < MSI_SWEEP_BCHUNK ? batch : MSI_SWEEP_BCHUNK;
  const dim3 grid((unsigned)(((long)width * (num_depths / ns) + 255) / 256), height, (batch + bchunk - 1) / bchunk);
  // non-temporal whole-pixel stores when the volume cannot stay in the 256-MB Infinity Cache anyway (sweep_store16)
  const int nt = MSI_SWEEP_NT < 0 ? 0 : (MSI_SWEEP_NT > 0 || (size_t)batch * height * width * psv_channels * (psv_bf16 ? 2 : 4) > ((size_t)256 << 20)) ? 1 : 0;
#define MSI_LAUNCH_SWEEP(T, NS_, NSRC_, LOOP_)                                                                          \
  hipLaunchKernelGGL((ods_sweep_kernel<T, NS_, NSRC_, LOOP_>), grid, dim3(256), 0, msi::as_stream(stream), image, image1, \
                     pose, pose1, intrinsics, depths, trig, batch, height, width, num_depths, (float)order,      \
                     static_cast<T *>(psv), psv_channels, channel_offset, make_consts(height, width),           \
                     (num_depths / NS_) == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)(num_depths / NS_)),  \
                     (pair && 64 % (num_depths / NS_) == 0 && ((long)width * (num_depths / NS_)) % 64 == 0) ? 1 + 2 * nt : 0, bchunk)
#define MSI_LAUNCH_SWEEP_N(T, NSRC_, LOOP_)                                                              \
  { if (ns == 4) MSI_LAUNCH_SWEEP(T, 4, NSRC_, LOOP_); else if (ns == 2) MSI_LAUNCH_SWEEP(T, 2, NSRC_, LOOP_); else MSI_LAUNCH_SWEEP(T, 1, NSRC_, LOOP_); }
#define MSI_LAUNCH_SWEEP_L(T, NSRC_) { if (bchunk > 1) MSI_LAUNCH_SWEEP_N(T, NSRC_, 1) else MSI_LAUNCH_SWEEP_N(T, NSRC_, 0) }
#define MSI_LAUNCH_SWEEP_T(T) { if (pair) MSI_LAUNCH_SWEEP_L(T, 2) else MSI_LAUNCH_SWEEP_L(T, 1) }
  // the LDS-staged form (ods_sweep_lds_kernel): the double volume with whole-pixel stores, full blocks of complete pixels, NS = 2, and a batch worth a frame loop
  const int ng2 = num_depths / 2;
  if (pair && ns == 2 && batch >= MSI_SWEEP_LDS_MIN_BATCH && ng2 > 0 && 64 % ng2 == 0 && ((long)width * ng2) % 256 == 0) {
    const unsigned magic = ng2 == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)ng2);
    if (psv_bf16)
      hipLaunchKernelGGL((ods_sweep_lds_kernel<unsigned short, 2>), grid, dim3(256), 0, msi::as_stream(stream), image, image1, pose, pose1, intrinsics, depths, trig, batch,
                         height, width, num_depths, static_cast<unsigned short *>(psv), psv_channels, make_consts(height, width), magic, bchunk, nt);
    else
      hipLaunchKernelGGL((ods_sweep_lds_kernel<float, 2>), grid, dim3(256), 0, msi::as_stream(stream), image, image1, pose, pose1, intrinsics, depths, trig, batch,
                         height, width, num_depths, static_cast<float *>(psv), psv_channels, make_consts(height, width), magic, bchunk, nt);
    return msi::check_launch("ods_sphere_sweep (lds)");
  }
  if (psv_bf16) MSI_LAUNCH_SWEEP_T(unsigned short) else MSI_LAUNCH_SWEEP_T(float)
#undef MSI_LAUNCH_SWEEP_T
#undef MSI_LAUNCH_SWEEP_L
#undef MSI_LAUNCH_SWEEP_N
#undef MSI_LAUNCH_SWEEP
  return msi::check_launch("ods_sphere_sweep");
}

int msi_ods_sphere_sweep_f32(const float *image, const float *pose, const float *intrinsics,
                             const float *depths, const float *trig, int32_t batch,
                             int32_t height, int32_t width, int32_t num_depths, int32_t order,
                             float *psv, int32_t psv_channels, int32_t channel_offset,
                             msi_stream_t stream) {
  return sweep_common(image, nullptr, pose, nullptr, intrinsics, depths, trig, batch, height, width, num_depths, order, psv, 0,
                      psv_channels, channel_offset, stream);
}

int msi_ods_sphere_sweep_bf16(const float *image, const float *pose, const float *intrinsics,
                              const float *depths, const float *trig, int32_t batch,
                              int32_t height, int32_t width, int32_t num_depths, int32_t order,
                              void *psv_bf16, int32_t psv_channels, int32_t channel_offset,
                              msi_stream_t stream) {
  return sweep_common(image, nullptr, pose, nullptr, intrinsics, depths, trig, batch, height, width, num_depths, order, psv_bf16, 1,
                      psv_channels, channel_offset, stream);
}

int msi_ods_sweep_volume(const float *ref_image, const float *src_image, const float *ref_curr_pose,
                         const float *src_curr_pose, const float *intrinsics, const float *depths, const float *trig,
                         int32_t batch, int32_t height, int32_t width, int32_t num_depths, void *psv, int32_t psv_is_bf16,
                         msi_stream_t stream) {
  MSI_REQUIRE(src_image, "ods_sweep_volume: null pointer");
  return sweep_common(ref_image, src_image, ref_curr_pose, src_curr_pose, intrinsics, depths, trig, batch, height, width,
                      num_depths, 1, psv, psv_is_bf16 != 0, 6 * num_depths, 0, stream);
}

static int assemble_common(const void *psv, int psv_bf16, const float *pred, int color, float *rgba_native,
                           float *blend_weights, float *alphas, float *bg_blend_weights, int32_t batch, int32_t height,
                           int32_t width, int32_t num_planes, int pred_scaled, msi_stream_t stream) {
  MSI_REQUIRE(psv && pred && rgba_native, "assemble_rgba: null pointer");
  MSI_REQUIRE(batch >= 0 && height > 0 && width > 0 && num_planes > 0, "assemble_rgba: bad dims");
  MSI_REQUIRE(color >= COLOR_BLEND_PSV && color <= COLOR_ALPHA_ONLY, "assemble_rgba: which_color_pred %d", color);
  if (num_planes % 4 != 0)
    return msi::fail(MSI_E_UNSUPPORTED, "assemble_rgba: num_planes=%d must be a multiple of 4",
                     num_planes);
  const int c_pred = color == COLOR_BLEND_PSV ? 2 * num_planes : (color == COLOR_BLEND_BG ? 2 * num_planes + 3
                     : (color == COLOR_BLEND_BG_PSV ? 3 * num_planes + 3 : num_planes));
  const size_t lds = (size_t)K3_TP * ((6 * num_planes + 1) + (c_pred | 1)) * sizeof(float);
  if (lds > 160 * 1024)
    return msi::fail(MSI_E_UNSUPPORTED, "assemble_rgba: num_planes=%d needs %zu B of LDS", num_planes,
                     lds);
  const long npix = (long)batch * height * width;
  if (npix == 0) return MSI_OK;
  const long blocks = (npix + K3_TP - 1) / K3_TP;
  typedef void (*kern_t)(const void *, const float *, float4 *, float *, float *, float *, long, int, int, int);
  static const kern_t table[2][4] = {
      {assemble_kernel<0, 0>, assemble_kernel<0, 1>, assemble_kernel<0, 2>, assemble_kernel<0, 3>},
      {assemble_kernel<1, 0>, assemble_kernel<1, 1>, assemble_kernel<1, 2>, assemble_kernel<1, 3>}};
  const kern_t kern = table[psv_bf16 ? 1 : 0][color];
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return msi::fail(MSI_E_LAUNCH, "assemble_rgba: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, msi::as_stream(stream), psv, pred,
                     reinterpret_cast<float4 *>(rgba_native), blend_weights, alphas, bg_blend_weights, npix,
                     height * width, num_planes, pred_scaled);
  return msi::check_launch("assemble_rgba");
}

int msi_assemble_rgba_f32(const float *psv, const float *pred, float *rgba_native,
                          float *blend_weights, float *alphas, int32_t batch, int32_t height,
                          int32_t width, int32_t num_planes, msi_stream_t stream) {
  return assemble_common(psv, 0, pred, COLOR_BLEND_PSV, rgba_native, blend_weights, alphas, nullptr, batch, height, width,
                         num_planes, 0, stream);
}

int msi_assemble_rgba_bf16psv_f32(const void *psv_bf16, const float *pred, float *rgba_native,
                                  float *blend_weights, float *alphas, int32_t batch, int32_t height,
                                  int32_t width, int32_t num_planes, msi_stream_t stream) {
  return assemble_common(psv_bf16, 1, pred, COLOR_BLEND_PSV, rgba_native, blend_weights, alphas, nullptr, batch, height,
                         width, num_planes, 0, stream);
}

int msi_assemble_rgba_color_f32(const void *psv, int32_t psv_is_bf16, const float *pred, int32_t which_color_pred,
                                float *rgba_native, float *blend_weights, float *alphas, float *bg_blend_weights,
                                int32_t batch, int32_t height, int32_t width, int32_t num_planes, msi_stream_t stream) {
  return assemble_common(psv, psv_is_bf16 != 0, pred, which_color_pred, rgba_native, blend_weights, alphas,
                         bg_blend_weights, batch, height, width, num_planes, 0, stream);
}

int msi_assemble_rgba_scaled_f32(const float *psv, const float *weights_alphas, float *rgba_native,
                                 int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                                 msi_stream_t stream) {
  return assemble_common(psv, 0, weights_alphas, COLOR_BLEND_PSV, rgba_native, nullptr, nullptr, nullptr, batch, height,
                         width, num_planes, 1, stream);
}

int msi_resize_bilinear_f32(const float *in, float *out, int32_t batch, int32_t in_h, int32_t in_w,
                            int32_t channels, int32_t out_h, int32_t out_w, msi_stream_t stream) {
  MSI_REQUIRE(in && out, "resize_bilinear: null pointer");
  MSI_REQUIRE(batch >= 0 && in_h > 0 && in_w > 0 && channels > 0 && out_h > 0 && out_w > 0, "resize_bilinear: bad dims");
  if (channels % 4 != 0)
    return msi::fail(MSI_E_UNSUPPORTED, "resize_bilinear: channels=%d must be a multiple of 4", channels);
  const size_t n = (size_t)batch * out_h * out_w * (channels / 4);
  if (n == 0) return MSI_OK;
  const float sy = out_h > 1 ? (float)(in_h - 1) / (float)(out_h - 1) : 0.0f;
  const float sx = out_w > 1 ? (float)(in_w - 1) / (float)(out_w - 1) : 0.0f;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_1d(n)), dim3(256), 0, msi::as_stream(stream),
                     reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), n, in_h, in_w,
                     channels / 4, out_h, out_w, sy, sx);
  return msi::check_launch("resize_bilinear");
}

static int render_common(int mode, int ray, const float *rgba_native, const float *pose, const float *tgt_pos,
                         const float *intrinsics, const float *depths, const float *trig, int32_t batch,
                         int32_t height, int32_t width, int32_t num_planes, RayParams R, float *out_rgb,
                         float *out_depth, float *out_layers, int32_t *status, msi_stream_t stream) {
  MSI_REQUIRE(rgba_native && pose && depths, "render: null pointer");
  MSI_REQUIRE(batch >= 0 && height > 0 && width > 0 && num_planes > 0 && R.out_h > 0 && R.out_w > 0,
              "render: bad dims");
  MSI_REQUIRE((long)height * width < (1L << 24), "render: layers of more than 2^24 texels (24-bit texel offsets)");
  if (batch == 0) return MSI_OK;
  const long nblk = (long)((R.out_w + 63) / 64) * R.out_h * batch;
  MSI_REQUIRE(nblk < (1L << 31) - 8, "render: too many target pixels for one launch");
  const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(64, RENDER_SEGS);
  const PixConsts K = make_consts(height, width);
  const float4 *src = reinterpret_cast<const float4 *>(rgba_native);
  float4 *lay = reinterpret_cast<float4 *>(out_layers);
  hipStream_t s = msi::as_stream(stream);
  DepthFrac F;
  for (int d = 0; d < DEPTH_FRAC_MAX; ++d) F.f[d] = d < num_planes ? (float)((double)d / (double)num_planes) : 0.0f;
#define MSI_LAUNCH_RENDER(M, RY)                                                                       \
  hipLaunchKernelGGL((render_kernel<M, RY>), grid, block, 0, s, src, pose, tgt_pos, intrinsics, depths, \
                     trig, batch, height, width, num_planes, out_rgb, out_depth, lay, K, R, F, status)
  if (ray == RAY_EQUIRECT) {
    switch (mode) {
      case RENDER_RGB: MSI_LAUNCH_RENDER(RENDER_RGB, RAY_EQUIRECT); break;
      case RENDER_DEPTH: MSI_LAUNCH_RENDER(RENDER_DEPTH, RAY_EQUIRECT); break;
      case RENDER_RGB | RENDER_DEPTH: MSI_LAUNCH_RENDER(RENDER_RGB | RENDER_DEPTH, RAY_EQUIRECT); break;
      case RENDER_LAYERS: MSI_LAUNCH_RENDER(RENDER_LAYERS, RAY_EQUIRECT); break;
      default: return msi::fail(MSI_E_BADARG, "render: bad mode %d", mode);
    }
  } else if (ray == RAY_ODS) {
    MSI_LAUNCH_RENDER(RENDER_RGB, RAY_ODS);
  } else {
    MSI_LAUNCH_RENDER(RENDER_RGB, RAY_PERSPECTIVE);
  }
#undef MSI_LAUNCH_RENDER
  return msi::check_launch("render");
}

static RayParams same_size(int32_t height, int32_t width) {
  RayParams R;
  R.out_h = height; R.out_w = width; R.order = 1.0f;
  R.s0 = R.sstep = R.t0 = R.tstep = 0.0f;
  return R;
}

int msi_render_equirect_f32(const float *rgba_native, const float *tgt_pose_rt,
                            const float *tgt_pos, const float *depths, const float *trig,
                            int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                            float *out_rgb, float *out_depth, int32_t *status_device, msi_stream_t stream) {
  MSI_REQUIRE(out_rgb || out_depth, "render_equirect: both outputs are NULL");
  MSI_REQUIRE(tgt_pos && trig, "render_equirect: null pointer");
  const int mode = (out_rgb ? RENDER_RGB : 0) | (out_depth ? RENDER_DEPTH : 0);
  return render_common(mode, RAY_EQUIRECT, rgba_native, tgt_pose_rt, tgt_pos, nullptr, depths, trig, batch, height,
                       width, num_planes, same_size(height, width), out_rgb, out_depth, nullptr, status_device, stream);
}

int msi_project_layers_f32(const float *rgba_native, const float *tgt_pose_rt,
                           const float *tgt_pos, const float *depths, const float *trig,
                           int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                           float *out_layers, int32_t *status_device, msi_stream_t stream) {
  MSI_REQUIRE(out_layers && tgt_pos && trig, "project_layers: null pointer");
  return render_common(RENDER_LAYERS, RAY_EQUIRECT, rgba_native, tgt_pose_rt, tgt_pos, nullptr, depths, trig, batch,
                       height, width, num_planes, same_size(height, width), nullptr, nullptr, out_layers, status_device, stream);
}

int msi_perspective_plane_sweep_f32(const float *image, const float *pose, const float *intrinsics,
                                    const float *depths, int32_t batch, int32_t height, int32_t width,
                                    int32_t num_depths, float *psv, int32_t psv_channels,
                                    int32_t channel_offset, msi_stream_t stream) {
  MSI_REQUIRE(image && pose && intrinsics && depths && psv, "perspective_plane_sweep: null pointer");
  MSI_REQUIRE(batch >= 0 && height > 1 && width > 1 && num_depths > 0, "perspective_plane_sweep: bad dims");
  MSI_REQUIRE(channel_offset >= 0 && channel_offset + 3 * num_depths <= psv_channels,
              "perspective_plane_sweep: channel window outside %d channels", psv_channels);
  if (batch == 0) return MSI_OK;
  MSI_REQUIRE((long)width * num_depths < 2147483647L && height <= 65535 && batch <= 65535,
              "perspective_plane_sweep: problem too large");
  const dim3 grid((unsigned)(((long)width * num_depths + 255) / 256), height, batch);
  // spherical.uv_grid (spherical.py:46-48), tf.linspace fp32 semantics
  const float s0 = (float)(-1.0 + 1.0 / width), s1 = (float)(1.0 - 1.0 / width);
  const float t0 = (float)(-1.0 + 1.0 / height), t1 = (float)(1.0 - 1.0 / height);
  const unsigned magic = num_depths == 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)num_depths);
  const int nt = (size_t)batch * height * width * psv_channels * 4 > ((size_t)256 << 20) ? 1 : 0;
  // full waves of complete pixels whose 3 D-float runs are 16-byte aligned, 32-bit byte offsets into one face
  const bool fast = 64 % num_depths == 0 && ((long)width * num_depths) % 256 == 0 && (3 * num_depths) % 4 == 0 && psv_channels % 4 == 0 && channel_offset % 4 == 0 &&
                    (long)height * width * 12 < 2147483647L && num_depths >= 4;
  if (fast)
    hipLaunchKernelGGL(pp_sweep_kernel<1>, grid, dim3(256), 0, msi::as_stream(stream), image, pose,
                       intrinsics, depths, batch, height, width, num_depths, s0, (s1 - s0) / (float)(width - 1), t0,
                       (t1 - t0) / (float)(height - 1), psv, psv_channels, channel_offset, magic, nt);
  else
    hipLaunchKernelGGL(pp_sweep_kernel<0>, grid, dim3(256), 0, msi::as_stream(stream), image, pose,
                       intrinsics, depths, batch, height, width, num_depths, s0, (s1 - s0) / (float)(width - 1), t0,
                       (t1 - t0) / (float)(height - 1), psv, psv_channels, channel_offset, magic, nt);
  return msi::check_launch("perspective_plane_sweep");
}

int msi_mpi_render_f32(const float *rgba_native, const float *tgt_pose, const float *intrinsics,
                       const float *intrinsics_inv, const float *depths, int32_t batch, int32_t height,
                       int32_t width, int32_t num_planes, float *out_rgb, msi_stream_t stream) {
  MSI_REQUIRE(rgba_native && tgt_pose && intrinsics && intrinsics_inv && depths && out_rgb, "mpi_render: null pointer");
  MSI_REQUIRE(batch >= 0 && height > 0 && width > 0 && num_planes > 0, "mpi_render: bad dims");
  if (num_planes > MPI_MAX_PLANES)
    return msi::fail(MSI_E_UNSUPPORTED, "mpi_render: at most %d planes", MPI_MAX_PLANES);
  if (batch == 0) return MSI_OK;
  const dim3 grid((width + 63) / 64, (height + 3) / 4, batch), block(64, 4);
  hipLaunchKernelGGL(mpi_render_kernel, grid, block, 0, msi::as_stream(stream),
                     reinterpret_cast<const float4 *>(rgba_native), tgt_pose, intrinsics, intrinsics_inv, depths, batch,
                     height, width, num_planes, out_rgb);
  return msi::check_launch("mpi_render");
}

int msi_render_ods_f32(const float *rgba_native, const float *pose, const float *intrinsics,
                       const float *depths, const float *trig, int32_t batch, int32_t height,
                       int32_t width, int32_t num_planes, int32_t order, float *out_rgb, int32_t *status_device,
                       msi_stream_t stream) {
  MSI_REQUIRE(out_rgb && intrinsics && trig, "render_ods: null pointer");
  MSI_REQUIRE(order == 1 || order == -1, "render_ods: order must be +1 or -1");
  RayParams R = same_size(height, width);
  R.order = (float)order;
  return render_common(RENDER_RGB, RAY_ODS, rgba_native, pose, nullptr, intrinsics, depths, trig, batch, height,
                       width, num_planes, R, out_rgb, nullptr, nullptr, status_device, stream);
}

int msi_render_perspective_f32(const float *rgba_native, const float *pose, const float *tgt_pos,
                               const float *depths, int32_t batch, int32_t height, int32_t width,
                               int32_t num_planes, int32_t tgt_height, int32_t tgt_width, float *out_rgb,
                               int32_t *status_device, msi_stream_t stream) {
  MSI_REQUIRE(out_rgb && tgt_pos, "render_perspective: null pointer");
  MSI_REQUIRE(tgt_height > 1 && tgt_width > 1, "render_perspective: bad target size");
  RayParams R = same_size(tgt_height, tgt_width);
  // spherical.uv_grid (spherical.py:46-48) with tf.linspace fp32 semantics
  const float s0 = (float)(-1.0 + 1.0 / tgt_width), s1 = (float)(1.0 - 1.0 / tgt_width);
  const float t0 = (float)(-1.0 + 1.0 / tgt_height), t1 = (float)(1.0 - 1.0 / tgt_height);
  R.s0 = s0; R.sstep = (s1 - s0) / (float)(tgt_width - 1);
  R.t0 = t0; R.tstep = (t1 - t0) / (float)(tgt_height - 1);
  return render_common(RENDER_RGB, RAY_PERSPECTIVE, rgba_native, pose, tgt_pos, nullptr, depths, nullptr, batch,
                       height, width, num_planes, R, out_rgb, nullptr, nullptr, status_device, stream);
}

}  // extern "C"
