// Which clock does the VALU issue rate of tools/valu_rate.hip refer to?  Every wavefront runs a long chain of
// independent v_add_u32 / v_fma_f32 / v_pk_fma_f32 and reads BOTH counters before and after: s_memtime (clock64(): the
// shader clock domain on gfx9-family parts) and s_memrealtime (wall_clock64(): constant 100 MHz).  Their ratio is the
// shader clock DURING the measurement, and instructions / s_memtime ticks is cycles per instruction without assuming a
// frequency.  Run beside `rocm-smi --showclocks` polling (tools/valu_clock.sh).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_clock.hip -o tools/valu_clock.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 16
template <int OP>
__global__ void k(unsigned long long* out, int iters) {
  int v[REP];
  float f[REP];
#pragma unroll
  for (int i = 0; i < REP; ++i) { v[i] = threadIdx.x * (i + 3); f[i] = (float)v[i]; }
  int b = threadIdx.x + 1;
  float fb = 1.0001f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fb));
      if (OP == 2) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (OP == 3) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[i]) : "v"(b));
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  int s = 0;
#pragma unroll
  for (int i = 0; i < REP; ++i) s += v[i] + (int)f[i];
  if (threadIdx.x == 0) {
    out[blockIdx.x * 3 + 0] = c1 - c0;
    out[blockIdx.x * 3 + 1] = w1 - w0;
    out[blockIdx.x * 3 + 2] = (unsigned long long)s;
  }
}

template <int OP>
void run(const char* name, int waves, int ncu) {
  const int blocks = ncu * 4 * waves;
  unsigned long long* d;
  hipMalloc(&d, (size_t)blocks * 3 * 8);
  const int iters = 400000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 64>>>(d, 2000);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 64>>>(d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h((size_t)blocks * 3);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sc = 0, sw = 0;
  for (int i = 0; i < blocks; ++i) { sc += (double)h[i * 3]; sw += (double)h[i * 3 + 1]; }
  sc /= blocks; sw /= blocks;
  const double instr = (double)iters * REP;                   // per wavefront
  const double wall_s = sw / 100e6;
  printf("%-12s waves/SIMD %d: kernel %.1f ms | per wavefront: s_memtime ticks %.3g, s_memrealtime %.3g (= %.2f ms) -> s_memtime runs at %.1f MHz | "
         "%.2f s_memtime ticks per instruction per wavefront, %.2f per instruction per SIMD | %.1f M wave-instr/s/SIMD\n",
         name, waves, ms, sc, sw, wall_s * 1e3, sc / wall_s / 1e6, sc / instr, sc / instr / waves, instr * waves / wall_s / 1e6);
  fflush(stdout);
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("# %s, %d CUs, prop.clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_u32", w, p.multiProcessorCount);
    run<1>("v_fma_f32", w, p.multiProcessorCount);
    run<2>("v_max_i32", w, p.multiProcessorCount);
    run<3>("v_pk_add_i16", w, p.multiProcessorCount);
  }
  return 0;
}
