// Micro-benchmark: issue rate of the integer VALU ops the DP kernels are made of,
// per SIMD, at a given number of resident waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 16
template <int OP>
__global__ void k(int* out, int iters, int seed) {
  int v[REP];
#pragma unroll
  for (int i = 0; i < REP; ++i) v[i] = threadIdx.x * (i + 3) + seed;
  int b = seed + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 2) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 3) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 4) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
      if (OP == 7) asm volatile("v_cmp_eq_u32 vcc, %0, %1" ::"v"(v[i]), "v"(b) : "vcc");
      if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(long long*)&v[i & ~1]) : "v"(*(long long*)&v[i & ~1]));
      if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 11) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(v[i]));
      if (OP == 12) asm volatile("v_pk_mad_i16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 14) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 15) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 16) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 17) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 18) asm volatile("v_pk_lshlrev_b16 %0, 2, %0" : "+v"(v[i]));
      if (OP == 19) asm volatile("v_readlane_b32 s20, %0, 3" ::"v"(v[i]) : "s20");
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < REP; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int wavesPerSimd) {
  int ncu = 256;
  int blocks = ncu * 4 * wavesPerSimd;  // 64-thread blocks: one wave each
  int iters = 4000;
  int* out;
  hipMalloc(&out, blocks * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<OP><<<blocks, 64>>>(out, 10, 1);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 64>>>(out, iters, 1);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double instr = (double)blocks * iters * REP;     // wave-instructions
  double per_simd_per_s = instr / (ncu * 4) / (ms * 1e-3);
  printf("%-16s waves/SIMD %d: %.3f ms, %.1f M wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz\n", name, wavesPerSimd,
         ms, per_simd_per_s / 1e6, 2.4e9 / per_simd_per_s);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_u32", w);
    run<1>("v_max_i32", w);
    run<2>("v_max3_i32", w);
    run<13>("v_add3_u32", w);
    run<10>("v_lshl_add_u32", w);
    run<11>("v_bfe_u32", w);
    run<15>("v_xor_b32", w);
    run<16>("v_bfi_b32", w);
    run<6>("v_cndmask_b32", w);
    run<7>("v_cmp_eq_u32", w);
    run<5>("v_mov_dpp_wshr", w);
    run<19>("v_readlane", w);
    run<3>("v_pk_add_i16", w);
    run<17>("v_pk_sub_i16", w);
    run<4>("v_pk_max_i16", w);
    run<14>("v_pk_min_u16", w);
    run<12>("v_pk_mad_i16", w);
    run<18>("v_pk_lshlrev_b16", w);
    run<8>("v_fma_f32", w);
    run<9>("v_pk_fma_f32", w);
  }
  return 0;
}
