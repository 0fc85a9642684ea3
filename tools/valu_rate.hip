// Micro-benchmark (round 2): issue rate of the integer VALU ops the DP kernels are made of, per SIMD, at 1 / 2 / 4 / 8
// resident wavefronts per SIMD, with INDEPENDENT chains (16 accumulators per lane: the throughput the issue port
// sustains) and ONE DEPENDENT chain (every instruction reads the previous result: the latency a dependent recurrence
// sees).  Kernels run >= 50 ms so that clock ramp and launch overhead do not matter.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin && tools/valu_rate.bin
// Reference point (/opt/skills/guides/MI355X_MICROARCH.md): a SIMD issues a wave64 VALU instruction over 2 cycles
// (32 lanes per cycle), i.e. 1.2 G wave-instructions/s per SIMD at 2.4 GHz (157.3 TFLOP/s fp32 = 1024 x 32 x 2 x 2.4e9).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP 16
#define OPS(X)                                                                                         \
  X(0, "v_add_u32", "v_add_u32 %0, %0, %1")                                                            \
  X(1, "v_max_i32", "v_max_i32 %0, %0, %1")                                                            \
  X(2, "v_max3_i32", "v_max3_i32 %0, %0, %1, %1")                                                      \
  X(3, "v_pk_add_i16", "v_pk_add_i16 %0, %0, %1")                                                      \
  X(4, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1")                                                      \
  X(5, "v_mov_dpp_wave_shr", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")             \
  X(6, "v_and_b32", "v_and_b32 %0, %0, %1")                                                            \
  X(7, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1")                                                        \
  X(8, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1")                                               \
  X(9, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 15, %0")                                              \
  X(10, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1")                                                     \
  X(11, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1")

template <int OP, bool DEP>
__global__ void k(int* out, int iters, int seed) {
  int v[REP];
#pragma unroll
  for (int i = 0; i < REP; ++i) v[i] = threadIdx.x * (i + 3) + seed;
  int b = seed + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      int& x = v[DEP ? 0 : i];
#define X(ID, NAME, ASM) \
      if (OP == ID) asm volatile(ASM : "+v"(x) : "v"(b));
      OPS(X)
#undef X
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < REP; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, bool DEP>
void run(const char* name, int wavesPerSimd, int ncu, FILE* f) {
  const int blocks = ncu * 4 * wavesPerSimd;  // 64-thread blocks: one wave each
  int* out;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<OP, DEP><<<blocks, 64>>>(out, 1000, 1);
  hipDeviceSynchronize();
  // calibrate to ~60 ms
  int iters = 20000;
  float ms = 0;
  for (int round = 0; round < 2; ++round) {
    hipEventRecord(a);
    k<OP, DEP><<<blocks, 64>>>(out, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    if (round == 0) iters = (int)(iters * (60.0 / (ms > 0.01 ? ms : 0.01)));
  }
  const double instr = (double)blocks * iters * REP;     // wave-instructions
  const double per_simd_per_s = instr / (ncu * 4) / (ms * 1e-3);
  fprintf(f, "%-20s %-11s waves/SIMD %d: %7.2f ms, %7.1f M wave-instr/s/SIMD = %5.2f cycles/instr @2.4GHz\n", name,
          DEP ? "dependent" : "independent", wavesPerSimd, ms, per_simd_per_s / 1e6, 2.4e9 / per_simd_per_s);
  fflush(f);
  hipFree(out);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  FILE* f = stdout;
  fprintf(f, "# %s, %d CUs, clock %d MHz (prop.clockRate)\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
  for (int w : {1, 2, 4, 8}) {
#define X(ID, NAME, ASM)               \
    run<ID, false>(NAME, w, ncu, f);   \
    if (w <= 2) run<ID, true>(NAME, w, ncu, f);
    OPS(X)
#undef X
  }
  return 0;
}
