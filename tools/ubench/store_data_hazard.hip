// Stand-alone reproducer (r05) for DESIGN.md section 4 "the lost stores": on gfx950 a 16-byte buffer store whose soffset is an SGPR, followed IMMEDIATELY by a VALU
// write of its first data register, can carry the NEW value -- hipcc's hazard recognizer inserts the wait state for wide MUBUF stores only when soffset is not a
// register.  Every lane stores four dwords {tag, tag, tag, tag} (tag = a per-lane, per-iteration value < 0x80000000) and overwrites the first data register with
// 0xdeadbeef after 0, 1, 2 or 3 wait states (nothing, s_nop 0, s_nop 1, s_nop 2), with the store's soffset an SGPR (the sequence the compiler emitted, no wait state) or
// the literal 0 (the form in which the compiler inserts the wait states itself).
// The host counts stored 16-byte records whose first dword is 0xdeadbeef.  Many workgroups store at once so that the memory pipeline backs up (the data is read late only then).
// Build: hipcc --offload-arch=gfx950 -O3 store_data_hazard.hip -o store_data_hazard ; run: ./store_data_hazard [launches] [iterations per lane]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int WAIT, int LITERAL>   // WAIT wait states between the store and the overwrite (0 = none, n = s_nop n-1); LITERAL: soffset is the literal 0 (the step is added to voffset)
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned slot_bytes) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0x7fffffff, 0x00020000);
  unsigned voff = tid * 16u;
  unsigned soff = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned tag = (tid * 2654435761u + (unsigned)it * 40503u) & 0x7fffffffu;
    unsigned o0;
    // data in v[4:7] (v4 tied to the output so that the overwrite is visible to the compiler), voffset in %5, descriptor in %6, soffset in %7
#define MSI_SDH(WS)                                                                                                              \
    if (LITERAL) asm volatile("v_add_u32 %5, %5, %7\n\tbuffer_store_dwordx4 v[4:7], %5, %6, 0 offen\n\t" WS "v_mov_b32 v4, 0xdeadbeef\n\tv_sub_u32 %5, %5, %7" \
                              : "={v4}"(o0) : "0"(tag), "{v5}"(tag), "{v6}"(tag), "{v7}"(tag), "v"(voff), "s"(rsrc), "s"(soff) : "memory");        \
    else asm volatile("buffer_store_dwordx4 v[4:7], %5, %6, %7 offen\n\t" WS "v_mov_b32 v4, 0xdeadbeef"                                           \
                      : "={v4}"(o0) : "0"(tag), "{v5}"(tag), "{v6}"(tag), "{v7}"(tag), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (WAIT == 0) { MSI_SDH("") } else if (WAIT == 1) { MSI_SDH("s_nop 0\n\t") } else if (WAIT == 2) { MSI_SDH("s_nop 1\n\t") } else { MSI_SDH("s_nop 2\n\t") }
#undef MSI_SDH
    if (o0 != 0xdeadbeefu) out[0] = 1;                     // (keeps the overwrite alive; never true)
    soff += slot_bytes;                                    // next iteration: the next slab of the buffer (wave-uniform -> an SGPR)
  }
}

template <int WAIT, int SADDR>   // the same experiment with global_store_dwordx4: SADDR = 1 -> `v, v[4:7], s[base:base+1]`, 0 -> `v[addr:addr+1], v[4:7], off`
__global__ void __launch_bounds__(256) kg(unsigned *out, int iters, unsigned slot_bytes) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const unsigned tag = (tid * 2654435761u + (unsigned)it * 40503u) & 0x7fffffffu;
    unsigned o0;
    const unsigned voff = tid * 16u;
    unsigned char *base = reinterpret_cast<unsigned char *>(out) + (size_t)it * slot_bytes;   // wave-uniform
    unsigned long long vaddr = (unsigned long long)(base + voff);
#define MSI_SDG(WS)                                                                                                              \
    if (SADDR) asm volatile("global_store_dwordx4 %5, v[4:7], %6\n\t" WS "v_mov_b32 v4, 0xdeadbeef"                               \
                            : "={v4}"(o0) : "0"(tag), "{v5}"(tag), "{v6}"(tag), "{v7}"(tag), "v"(voff), "s"(base) : "memory");       \
    else asm volatile("global_store_dwordx4 %5, v[4:7], off\n\t" WS "v_mov_b32 v4, 0xdeadbeef"                                     \
                      : "={v4}"(o0) : "0"(tag), "{v5}"(tag), "{v6}"(tag), "{v7}"(tag), "v"(vaddr) : "memory");
    if (WAIT == 0) { MSI_SDG("") } else if (WAIT == 1) { MSI_SDG("s_nop 0\n\t") } else { MSI_SDG("s_nop 1\n\t") }
#undef MSI_SDG
    if (o0 != 0xdeadbeefu) out[0] = 1;
  }
}

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 6, iters = argc > 2 ? atoi(argv[2]) : 32;
  const int blocks = 256 * 8;
  const size_t slot = (size_t)blocks * 256 * 16;            // bytes one iteration of the whole grid writes
  unsigned *d;
  (void)hipMalloc(&d, slot * iters);
  std::vector<unsigned> h(slot * iters / 4);
  for (int form = 0; form < 8; ++form) {
    const int wait = form >> 1, literal = form & 1;
    long bad = 0, total = 0, other = 0;
    for (int l = 0; l < launches; ++l) {
      (void)hipMemset(d, 0, slot * iters);
      const dim3 g(blocks), b(256);
      switch (form) {
        case 0: hipLaunchKernelGGL((k<0, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 1: hipLaunchKernelGGL((k<0, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 2: hipLaunchKernelGGL((k<1, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 3: hipLaunchKernelGGL((k<1, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 4: hipLaunchKernelGGL((k<2, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 5: hipLaunchKernelGGL((k<2, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 6: hipLaunchKernelGGL((k<3, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        default: hipLaunchKernelGGL((k<3, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
      }
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d, slot * iters, hipMemcpyDeviceToHost);
      for (int it = 0; it < iters; ++it)
        for (unsigned t = 0; t < (unsigned)blocks * 256; ++t) {
          const unsigned *r = &h[((size_t)it * blocks * 256 + t) * 4];
          const unsigned tag = (t * 2654435761u + (unsigned)it * 40503u) & 0x7fffffffu;
          ++total;
          if (r[0] == 0xdeadbeefu && r[1] == tag && r[2] == tag && r[3] == tag) ++bad;
          else if (!(r[0] == tag && r[1] == tag && r[2] == tag && r[3] == tag) && !(it == 0 && t == 0)) ++other;
        }
    }
    printf("buffer_store %-8s soffset, %d wait state(s) before the VALU write: %9ld of %ld stored records carry the overwritten first dword (%ld otherwise wrong)\n",
           literal ? "literal" : "SGPR", wait, bad, total, other);
  }
  for (int form = 0; form < 6; ++form) {
    const int wait = form >> 1, saddr = form & 1;
    long bad = 0, total = 0, other = 0;
    for (int l = 0; l < launches; ++l) {
      (void)hipMemset(d, 0, slot * iters);
      const dim3 g(blocks), b(256);
      switch (form) {
        case 0: hipLaunchKernelGGL((kg<0, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 1: hipLaunchKernelGGL((kg<0, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 2: hipLaunchKernelGGL((kg<1, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 3: hipLaunchKernelGGL((kg<1, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        case 4: hipLaunchKernelGGL((kg<2, 0>), g, b, 0, 0, d, iters, (unsigned)slot); break;
        default: hipLaunchKernelGGL((kg<2, 1>), g, b, 0, 0, d, iters, (unsigned)slot); break;
      }
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h.data(), d, slot * iters, hipMemcpyDeviceToHost);
      for (int it = 0; it < iters; ++it)
        for (unsigned t = 0; t < (unsigned)blocks * 256; ++t) {
          const unsigned *r = &h[((size_t)it * blocks * 256 + t) * 4];
          const unsigned tag = (t * 2654435761u + (unsigned)it * 40503u) & 0x7fffffffu;
          ++total;
          if (r[0] == 0xdeadbeefu && r[1] == tag && r[2] == tag && r[3] == tag) ++bad;
          else if (!(r[0] == tag && r[1] == tag && r[2] == tag && r[3] == tag) && !(it == 0 && t == 0)) ++other;
        }
    }
    printf("global_store %-5s, %d wait state(s) before the VALU write: %9ld of %ld stored records carry the overwritten first dword (%ld otherwise wrong)\n",
           saddr ? "saddr" : "off", wait, bad, total, other);
  }
  (void)hipFree(d);
  return 0;
}
