// Micro-benchmark (r06): WHICH clock do the conv kernels run at?  DESIGN.md had two stories -- s_memtime advancing at 1.4-1.6 GHz under the
// six-product kernels (r04) and rocm-smi sampling sclk 1.92-1.99 GHz during the same load (r05) -- and nobody had put both instruments on one run.
//
// One launch, one workgroup per CU (dynamic LDS 96 KB forces that): the first NPROBE workgroups are PROBES, the rest are LOAD.
//   probe: wave 0 runs chains of 4096 DEPENDENT v_fma_f32 (one instruction per issue slot of its SIMD, nothing else on the CU) and stamps every chain with
//          s_memtime (the "shader clock" counter) and s_memrealtime (constant 100 MHz).  Cycles per fma are a property of the pipeline, not of the frequency:
//          if s_memtime ticks per chain stay put while ticks per MICROSECOND fall, s_memtime counts real shader cycles and the fall IS the clock.
//   load : four waves per CU issue v_mfma_f32_32x32x16_bf16 back to back, mode 0: nothing (idle chip), 1: constant operands, 2: operands that change between
//          consecutive MFMAs (eight register sets of pseudo-random bf16), 3: mode 2 + 12 ds_read_b128 per 12 MFMAs (the six-product k-loop's mix).
// The host samples the driver's view at ~20 Hz meanwhile (hwmon freq1_input = sclk, power1_average / power1_input; pp_dpm_sclk's starred level).
// Output per mode: s_memtime GHz on probe and load CUs, ticks per fma, MFMA rate (PFLOP/s dense bf16), driver sclk / power min-median-max.
// Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe -lpthread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NPROBE = 8, CHAIN = 4096, MAXREC = 8192;
struct Rec { unsigned long long mt, rt; };

template <int OFF>
__device__ __forceinline__ v4f rd(unsigned addr) {
  v4f v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int MODE>
__global__ void __launch_bounds__(256) k(Rec *rec, int *nrec, unsigned long long *load, float *sink, long iters, unsigned long long run_ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x < NPROBE) {
    if (wave != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float x = 1.0f + lane * 1e-9f;
    int n = 0;
    Rec *o = rec + (size_t)blockIdx.x * MAXREC;
    while (n < MAXREC) {
      const unsigned long long m0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < CHAIN / 64; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.999999f), "v"(1e-7f));
      }
      const unsigned long long m1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
      if (lane == 0) { o[n].mt = m1 - m0; o[n].rt = q1 - q0; }
      ++n;
      if (q1 - r0 > run_ticks) break;
      // spread the chains over the run: idle ~20 us between them (s_sleep does not count as load)
      for (int i = 0; i < 24; ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) nrec[blockIdx.x] = n;
    sink[blockIdx.x * 64 + lane] = x;
    return;
  }
  if (MODE == 0) return;
  for (int i = tid; i < 12288; i += 256) reinterpret_cast<unsigned *>(smem)[i] = (i * 2654435761u) >> 3 & 0x3f7f3f7fu;   // bf16 pairs in (-2, 2)
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem + (wave & 1) * 20480 + lane * 16;
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  v4f f[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    unsigned u[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) u[c] = MODE == 1 ? 0x3f003f00u : (((lane * 97 + i * 31 + c * 7 + wave * 13) * 2654435761u) >> 3 & 0x3f7f3f7fu) | 0x30003000u;
    f[i] = __builtin_bit_cast(v4f, *reinterpret_cast<uint4 *>(u));
  }
  const unsigned long long m0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  for (long it = 0; it < iters; ++it) {
    if (MODE == 3) {
#define RD(i) f[i] = rd<i * 1024>(base);
      RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11)
#undef RD
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]) :: "memory");
    }
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int ia = MODE == 1 ? 0 : m, ib = MODE == 1 ? 1 : (m + 5) % 12;
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[ia]), __builtin_bit_cast(bf16x8, f[ib]), acc[m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  sink[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { load[blockIdx.x * 2] = m1 - m0; load[blockIdx.x * 2 + 1] = q1 - q0; }
}

static std::vector<std::string> globv(const char *pat) {
  glob_t g; std::vector<std::string> out;
  if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; ++i) out.push_back(g.gl_pathv[i]);
  globfree(&g);
  return out;
}
static bool slurp(const std::string &p, std::string &s) {
  FILE *f = fopen(p.c_str(), "r"); if (!f) return false;
  char buf[4096]; size_t n = fread(buf, 1, sizeof(buf) - 1, f); fclose(f); buf[n] = 0; s = buf; return true;
}
struct Samples { std::vector<double> sclk_mhz, power_w, dpm_mhz; };
static void stat3(const char *name, std::vector<double> v, const char *unit) {
  if (v.empty()) { printf("    %-34s (no such sysfs node on this box)\n", name); return; }
  std::sort(v.begin(), v.end());
  printf("    %-34s min %.0f  median %.0f  max %.0f %s (%zu samples)\n", name, v.front(), v[v.size() / 2], v.back(), unit, v.size());
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 1.5;
  int dev = 0; hipSetDevice(dev);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  const int ncu = prop.multiProcessorCount;
  printf("clock_probe: %s, %d CUs, %d probe CUs + %d load CUs, one workgroup per CU, %.1f s per mode\n", prop.name, ncu, NPROBE, ncu - NPROBE, secs);
  const auto freq = globv("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input");
  auto pow_ = globv("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average");
  if (pow_.empty()) pow_ = globv("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input");
  const auto dpm = globv("/sys/class/drm/card*/device/pp_dpm_sclk");
  printf("  sysfs: freq1_input x%zu, power x%zu, pp_dpm_sclk x%zu (first of each is sampled)\n", freq.size(), pow_.size(), dpm.size());
  Rec *rec; int *nrec; unsigned long long *load; float *sink;
  hipMalloc(&rec, sizeof(Rec) * NPROBE * MAXREC); hipMalloc(&nrec, sizeof(int) * NPROBE); hipMalloc(&load, 16 * ncu); hipMalloc(&sink, 4 * 256 * ncu);
  std::vector<Rec> hrec((size_t)NPROBE * MAXREC); std::vector<int> hn(NPROBE); std::vector<unsigned long long> hload(2 * ncu);
  const char *names[4] = {"0: probes only (idle chip)", "1: MFMA, constant operands", "2: MFMA, changing operands", "3: MFMA changing + 12 ds_read_b128 / 12 MFMA"};
  // 12 MFMAs of 8 passes x 4 cycles = 384 cycles per iteration at best; aim at `secs` at 2.4 GHz (a slower clock only makes the load outlast the probes' run_ticks)
  const long iters = (long)(secs * 2.4e9 / 384.0);
  const unsigned long long run_ticks = (unsigned long long)(secs * 0.8 * 100e6);
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(rec, 0, sizeof(Rec) * NPROBE * MAXREC); hipMemset(nrec, 0, sizeof(int) * NPROBE); hipMemset(load, 0, 16 * ncu);
    std::atomic<bool> stop{false};
    Samples S;
    std::thread sampler([&]() {
      std::string s;
      while (!stop.load()) {
        if (!freq.empty() && slurp(freq[0], s)) S.sclk_mhz.push_back(atof(s.c_str()) / 1e6);
        if (!pow_.empty() && slurp(pow_[0], s)) S.power_w.push_back(atof(s.c_str()) / 1e6);
        if (!dpm.empty() && slurp(dpm[0], s)) { size_t st = s.find('*'); if (st != std::string::npos) { size_t b = s.rfind(':', st); if (b != std::string::npos) S.dpm_mhz.push_back(atof(s.c_str() + b + 1)); } }
        std::this_thread::sleep_for(std::chrono::milliseconds(40));
      }
    });
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::this_thread::sleep_for(std::chrono::milliseconds(150));
    S = Samples();
    hipEventRecord(e0);
    const size_t lds = 96 * 1024;
    switch (mode) {
      case 0: hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<0><<<ncu, 256, lds>>>(rec, nrec, load, sink, iters, run_ticks); break;
      case 1: hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<1><<<ncu, 256, lds>>>(rec, nrec, load, sink, iters, run_ticks); break;
      case 2: hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<2><<<ncu, 256, lds>>>(rec, nrec, load, sink, iters, run_ticks); break;
      default: hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<3><<<ncu, 256, lds>>>(rec, nrec, load, sink, iters, run_ticks); break;
    }
    hipEventRecord(e1);
    hipError_t err = hipEventSynchronize(e1);
    stop = true; sampler.join();
    if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hrec.data(), rec, sizeof(Rec) * NPROBE * MAXREC, hipMemcpyDeviceToHost); hipMemcpy(hn.data(), nrec, sizeof(int) * NPROBE, hipMemcpyDeviceToHost);
    hipMemcpy(hload.data(), load, 16 * ncu, hipMemcpyDeviceToHost);
    // probes: records of the middle half of the run
    std::vector<double> tpf, ghz;
    for (int p = 0; p < NPROBE; ++p)
      for (int i = hn[p] / 4; i < hn[p] * 3 / 4; ++i) {
        const Rec &r = hrec[(size_t)p * MAXREC + i];
        if (r.rt == 0) continue;
        tpf.push_back((double)r.mt / CHAIN); ghz.push_back((double)r.mt / ((double)r.rt / 100e6) / 1e9);
      }
    std::sort(tpf.begin(), tpf.end()); std::sort(ghz.begin(), ghz.end());
    printf("\n[pass %d] mode %s   (launch %.0f ms by HIP events)\n", rep, names[mode], ms);
    if (!tpf.empty())
      printf("    probe CUs : s_memtime ticks per dependent v_fma_f32 %.3f (min %.3f max %.3f) | s_memtime advances at %.3f GHz (min %.3f max %.3f) over %zu chains\n",
             tpf[tpf.size() / 2], tpf.front(), tpf.back(), ghz[ghz.size() / 2], ghz.front(), ghz.back(), tpf.size());
    if (mode) {
      std::vector<double> lg, rate;
      for (int b = NPROBE; b < ncu; ++b) {
        const double mt = (double)hload[2 * b], rt = (double)hload[2 * b + 1] / 100e6;
        if (rt <= 0) continue;
        lg.push_back(mt / rt / 1e9);
        rate.push_back(mt / ((double)iters * 12.0));            // s_memtime ticks per MFMA per wave (one wave per SIMD): 32 = the matrix pipe never idles
      }
      std::sort(lg.begin(), lg.end()); std::sort(rate.begin(), rate.end());
      const double wall = (double)hload[2 * NPROBE + 1] / 100e6;
      const double pflops = (double)(ncu - NPROBE) * 4 * iters * 12.0 * 2.0 * 32 * 32 * 16 / wall / 1e15;
      printf("    load CUs  : s_memtime advances at %.3f GHz (min %.3f max %.3f) | %.2f ticks per MFMA and wave (32 = back to back) | %.3f PFLOP/s dense bf16 on %d CUs = %.3f on %d\n",
             lg[lg.size() / 2], lg.front(), lg.back(), rate[rate.size() / 2], pflops, ncu - NPROBE, pflops * ncu / (ncu - NPROBE), ncu);
    }
    stat3("driver sclk (hwmon freq1_input)", S.sclk_mhz, "MHz");
    stat3("driver sclk (pp_dpm_sclk *)", S.dpm_mhz, "MHz");
    stat3("socket power (hwmon power1)", S.power_w, "W");
  }
  return 0;
}
