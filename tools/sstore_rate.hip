// Micro-benchmark (round 5): what do the four trace predicates of a profile-Gotoh cell (src/gotoh.h:135-138) cost when the
// compare masks leave the wavefront as SCALAR stores (v_cmp -> SGPR pair -> s_store_dwordx2: 1 VALU + 1 SMEM instruction per
// predicate) instead of being shifted into per-lane accumulators (v_cmp + v_addc_co: 2 VALU per predicate, one coalesced
// dword store per lane, row and 16 steps)?  The loop below is shaped like the DP step of msa_body.inc (K = 3 rows per lane,
// OTHER VALU instructions of filler per row, four predicates per row), 4 096 wavefronts, 16 per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/sstore_rate.hip -o tools/sstore_rate.bin && tools/sstore_rate.bin
// Also checks that the masks written through the scalar cache are what a vector load sees after s_dcache_wb.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

constexpr int K = 3;
constexpr int OTHER = 10;    // recurrence (8) + table address + slack, per row and step

// MODE 0: v_cmp + v_addc (today), MODE 1: v_cmp + s_store_dwordx2, MODE 2: filler only (no predicates)
template <int MODE>
__global__ __launch_bounds__(64, 4) void k(uint32_t* out, unsigned long long* mout, int steps, int seed) {
  const int lane = threadIdx.x;
  int a[K], b[K];
  uint32_t acc[K][4];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    a[i] = lane * 3 + i + seed;
    b[i] = lane * 5 - i;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = 0;
  }
  uint32_t* wout = out + (size_t)blockIdx.x * ((steps + 15) / 16) * K * 4 * 64;
  unsigned long long* mbase = mout + (size_t)blockIdx.x * steps * K * 4;
  for (int t = 0; t < steps; ++t) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
#pragma unroll
      for (int o = 0; o < OTHER; ++o) {
        if (o & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else asm volatile("v_max_i32 %0, %0, %1" : "+v"(b[i]) : "v"(a[(i + 1) % K]));
      }
      if (MODE == 0) {
        asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[i][0]) : "v"(a[i]), "v"(b[i]) : "vcc");
        asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[i][1]) : "v"(a[i]), "v"(b[i]) : "vcc");
        asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[i][2]) : "v"(a[i]), "v"(b[i]) : "vcc");
        asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc[i][3]) : "v"(a[i]), "v"(b[i]) : "vcc");
      } else if (MODE == 1) {
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_gt_i32 %0, %1, %2" : "=s"(m0) : "v"(a[i]), "v"(b[i]));
        asm volatile("v_cmp_lt_i32 %0, %1, %2" : "=s"(m1) : "v"(a[i]), "v"(b[i]));
        asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(m2) : "v"(a[i]), "v"(b[i]));
        asm volatile("v_cmp_ge_i32 %0, %1, %2" : "=s"(m3) : "v"(a[i]), "v"(b[i]));
        unsigned long long* p = mbase + ((size_t)t * K + i) * 4;
        // (the SGPR address is uniform: the compiler keeps it in SGPRs)
        asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(m0), "s"(p) : "memory");
        asm volatile("s_store_dwordx2 %0, %1, 0x8" ::"s"(m1), "s"(p) : "memory");
        asm volatile("s_store_dwordx2 %0, %1, 0x10" ::"s"(m2), "s"(p) : "memory");
        asm volatile("s_store_dwordx2 %0, %1, 0x18" ::"s"(m3), "s"(p) : "memory");
      }
    }
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the data SGPRs are reused by the next step)
    if (MODE == 0 && (t & 15) == 15) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        wout[((size_t)((t >> 4) * 2 + 0) * K + i) * 64 + lane] = (acc[i][0] & 0xffffu) | (acc[i][1] << 16);
        wout[((size_t)((t >> 4) * 2 + 1) * K + i) * 64 + lane] = (acc[i][2] & 0xffffu) | (acc[i][3] << 16);
      }
    }
  }
  if (MODE == 1) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) s += (uint32_t)a[i] + (uint32_t)b[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (MODE == 1) {
    // what a vector load sees of this wavefront's own masks (first step, last step)
    const unsigned long long first = __builtin_nontemporal_load(mbase + (lane & 7));
    const unsigned long long last = __builtin_nontemporal_load(mbase + ((size_t)(steps - 1) * K * 4) + (lane & 7));
    s += (uint32_t)(first >> lane) + (uint32_t)(last >> lane);
  }
  if (s == 0x12345678u) out[0] = s;
}

// correctness: a kernel whose predicates are known -- lane l, step t, row i, predicate q is ((l + t + i + q) % 3 == 0)
template <int WAITMODE>   // 0: s_waitcnt after every store, 1: once per step (12 stores in flight, SGPRs reused by the allocator)
__global__ __launch_bounds__(64) void kcheck(unsigned long long* mout, int steps) {
  const int lane = threadIdx.x;
  unsigned long long* mbase = mout + (size_t)blockIdx.x * steps * K * 4;
  for (int t = 0; t < steps; ++t)
    for (int i = 0; i < K; ++i) {
      unsigned long long* p = mbase + ((size_t)t * K + i) * 4;
      for (int q = 0; q < 4; ++q) {
        const int x = (lane + t + i + q + (int)blockIdx.x) % 3, z = 0;
        unsigned long long m;
        asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(m) : "v"(x), "v"(z));
        asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(m), "s"(p + q) : "memory");
        if (WAITMODE == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (WAITMODE == 1 && i == K - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int MODE>
float run(const char* name, int blocks, int steps, uint32_t* out, unsigned long long* mout) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<blocks, 64>>>(out, mout, steps, 1);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a);
    k<MODE><<<blocks, 64>>>(out, mout, steps, r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double rowsteps = (double)blocks * steps * K;
  printf("%-28s blocks %5d steps %5d: %8.3f ms  = %6.2f ns per wave row-step, %5.1f G row-steps/s\n", name, blocks, steps, best,
         best * 1e6 / ((double)steps * K) / ((blocks + 4095) / 4096), rowsteps / (best * 1e-3) / 1e9);
  fflush(stdout);
  return best;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  const int steps = 213 * 19;   // one junction of 19 merges
  const int maxblocks = 8192;
  uint32_t* out;
  unsigned long long* mout;
  hipMalloc(&out, (size_t)maxblocks * ((steps + 15) / 16) * K * 4 * 64 * 4);
  hipMalloc(&mout, (size_t)maxblocks * steps * K * 4 * 8);
  hipMemset(mout, 0, (size_t)maxblocks * steps * K * 4 * 8);
  // correctness of the scalar-store path
  for (int wm = 0; wm < 2; ++wm) {
    const int cb = 2048, cs = 100;
    hipMemset(mout, 0xee, (size_t)cb * cs * K * 4 * 8);
    if (wm == 0) kcheck<0><<<cb, 64>>>(mout, cs);
    else kcheck<1><<<cb, 64>>>(mout, cs);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)cb * cs * K * 4);
    hipMemcpy(h.data(), mout, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int bl = 0; bl < cb; ++bl)
      for (int t = 0; t < cs; ++t)
        for (int i = 0; i < K; ++i)
          for (int q = 0; q < 4; ++q) {
            unsigned long long e = 0;
            for (int l = 0; l < 64; ++l)
              if ((l + t + i + q + bl) % 3 == 0) e |= 1ull << l;
            if (h[(((size_t)bl * cs + t) * K + i) * 4 + q] != e) ++bad;
          }
    printf("scalar-store correctness (%s): %zu wrong masks of %zu\n", wm ? "one wait per step" : "wait after every store", bad, h.size());
  }
  for (int blocks : {1024, 4096, 8192}) {
    run<2>("filler only", blocks, steps, out, mout);
    run<0>("v_cmp + v_addc (today)", blocks, steps, out, mout);
    run<1>("v_cmp + s_store_dwordx2", blocks, steps, out, mout);
  }
  hipError_t e = hipDeviceSynchronize();
  printf("final status: %s\n", hipGetErrorString(e));
  return 0;
}
