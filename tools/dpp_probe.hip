// probe: direction of row_ror, and ds_read_i8_d16 / _d16_hi semantics on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int* out) {
  __shared__ int8_t tab[64];
  const int lane = threadIdx.x;
  tab[lane] = (int8_t)(lane - 32);
  __syncthreads();
  out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x121, 0xf, 0xf, true);        // row_ror:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x127, 0xf, 0xf, true);   // row_ror:7
  uint32_t x = 0xdeadbeefu;
  const uint32_t base = (uint32_t)(uintptr_t)&tab[0];
  const uint32_t a = base + lane, b = base + (63 - lane);
  asm volatile("ds_read_i8_d16 %0, %1\n\tds_read_i8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(a), "v"(b) : "memory");
  out[128 + lane] = (int)x;
}
int main() {
  int* d;
  hipMalloc(&d, 192 * 4);
  k<<<1, 64>>>(d);
  int h[192];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("row_ror:1 lane0..17:");
  for (int i = 0; i < 18; ++i) printf(" %d", h[i]);
  printf("\nrow_ror:7 lane0..17:");
  for (int i = 0; i < 18; ++i) printf(" %d", h[64 + i]);
  printf("\nd16: lane 0 %08x (want lo=%04x hi=%04x)  lane 40 %08x (want lo=%04x hi=%04x)\n", h[128], (uint16_t)(int16_t)-32, (uint16_t)(int16_t)31, h[128 + 40],
         (uint16_t)(int16_t)8, (uint16_t)(int16_t)-9);
  return 0;
}
