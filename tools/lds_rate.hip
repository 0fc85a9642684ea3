// What do the LDS access patterns of the short-read kernels cost?  Every wavefront issues a long run of INDEPENDENT LDS
// instructions of one pattern (8 in flight); cycles per instruction per CU = the LDS pipeline's occupancy for that pattern
// (16 wavefronts per CU, so latency is hidden and the pipeline is the limit).  Clock: s_memtime (shader clock).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_rate.hip -o tools/lds_rate.bin && tools/lds_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned long long ld8(const unsigned char* p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ unsigned ld4(const unsigned char* p) { unsigned v; __builtin_memcpy(&v, p, 4); return v; }

template <int P>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char s[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) s[i] = (unsigned char)(i * 7);
  __syncthreads();
  const int jit = (lane * 7) % 11;   // a small per-lane offset (the row a diagonal has reached)
  unsigned long long acc = 0;
  int base = 0;
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = (base + u * 16) & 1023;
      if (P == 0) acc += ld8(s + b * 4 + lane * 8);                       // aligned quadwords, stride 8
      if (P == 1) acc += ld8(s + b + lane + jit);                          // quadwords at byte stride 1 + jitter (round 3's letter loads)
      if (P == 2) acc += ld8(s + b + 4 * lane + (u & 3) + jit);            // quadwords at byte stride 4 + jitter (four diagonals per lane)
      if (P == 3) acc += s[b + lane];                                      // bytes, stride 1
      if (P == 4) acc += ld4(s + b * 4 + 4 * lane + 3);                    // unaligned dwords, stride 4
      if (P == 5) acc += ld4(s + b * 4 + 4 * lane);                        // aligned dwords, stride 4
      if (P == 6) acc += (unsigned)__builtin_amdgcn_ds_bpermute(((lane ^ (u + 1)) & 63) << 2, (int)(acc & 0xffff) + u);
      if (P == 7) acc += ld8(s + b + jit);                                 // quadwords, every lane inside the same 18 bytes (consensus letters)
      if (P == 8) acc += ld8(s + b * 4 + lane * 8 + 1);                    // unaligned quadwords, stride 8
      if (P == 9) acc += ld8(s + b + 2 * lane + jit);                      // quadwords at byte stride 2 + jitter
    }
    base += 128;
  }
  const unsigned long long c1 = clock64();
  if (lane == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = acc; }
  else if (acc == 0x123456789abcdefull) out[0] = acc;
}

template <int P>
void run(const char* name, int waves_per_cu) {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount, blocks = ncu * waves_per_cu, iters = 20000;
  unsigned long long* d;
  hipMalloc(&d, (size_t)blocks * 16);
  k<P><<<blocks, 64>>>(d, 200);
  hipDeviceSynchronize();
  k<P><<<blocks, 64>>>(d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * 2);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sc = 0;
  for (int i = 0; i < blocks; ++i) sc += (double)h[i * 2];
  const double per_wave = sc / blocks;                       // ticks one wavefront needed for iters * 8 instructions
  // s_memtime ticks at 100 MHz on this part?  report both the raw ticks per instruction per wavefront and per CU
  printf("%-64s %2d waves/CU: %8.2f ticks per instruction per wavefront, %7.2f per CU\n", name, waves_per_cu, per_wave / (iters * 8.0),
         per_wave / (iters * 8.0) / waves_per_cu);
  hipFree(d);
}

int main() {
  for (int w : {1, 16}) {
    run<0>("ds_read_b64 aligned, stride 8", w);
    run<8>("ds_read_b64 unaligned (+1), stride 8", w);
    run<1>("ds_read_b64 byte stride 1 + jitter (lane = diagonal)", w);
    run<9>("ds_read_b64 byte stride 2 + jitter", w);
    run<2>("ds_read_b64 byte stride 4 + jitter (4 diagonals per lane)", w);
    run<7>("ds_read_b64 all lanes within 18 bytes", w);
    run<3>("ds_read_u8 stride 1", w);
    run<5>("ds_read_b32 aligned, stride 4", w);
    run<4>("ds_read_b32 unaligned (+3), stride 4", w);
    run<6>("ds_bpermute_b32", w);
  }
  return 0;
}
