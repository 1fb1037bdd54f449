#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256, 1) void syn(const char* A, float* out, int n) {
  extern __shared__ char smem[];
  const unsigned voff = threadIdx.x * 16;
  unsigned lds = (unsigned)(size_t)smem;
  int cnt = n;
  asm volatile(
    "s_mov_b64 s[40:41], %[a]\n\t"
    "s_mov_b32 s42, 0x20000\n\t"
    "L_top%=:\n\t"
    "s_add_u32 m0, s42, 0x400\n\t"
    "s_nop 0\n\t"
    "global_load_lds_dwordx4 %[vo], s[40:41] offset:64\n\t"
    "s_add_u32 s40, s40, 64\n\t"
    "s_addc_u32 s41, s41, 0\n\t"
    "ds_read_b128 v[100:103], %[la] offset:1024\n\t"
    "ds_read_b128 v[104:107], %[la] offset:65024\n\t"
    "s_waitcnt lgkmcnt(0)\n\t"
    "v_mfma_f32_16x16x32_bf16 a[0:3], v[100:103], v[104:107], a[0:3]\n\t"
    "v_mfma_f32_16x16x32_bf16 a[252:255], v[100:103], v[104:107], a[252:255]\n\t"
    "s_waitcnt vmcnt(0)\n\t"
    "s_barrier\n\t"
    "s_sub_u32 %[c], %[c], 1\n\t"
    "s_cmp_lg_u32 %[c], 0\n\t"
    "s_cbranch_scc1 L_top%=\n\t"
    "s_nop 15\n\t"
    : [c] "+s"(cnt)
    : [a] "s"(A), [vo] "v"(voff), [la] "v"(lds)
    : "memory", "s40", "s41", "s42", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "a0", "a1", "a2", "a3", "a252", "a253", "a254", "a255", "scc");
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "i"(252));
  out[threadIdx.x] = r;
}
