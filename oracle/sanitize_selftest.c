/* TEST INFRASTRUCTURE ONLY -- sanitizer self-test of the C restatement ("build oracle + host shim with
 * -fsanitize=address,undefined").  Drives every batch-level function of oracle/delly_oracle.c over seeded synthetic junctions
 * (deletions with noisy / low-complexity consensus, read sets for msa, an insertion, long reads for msaEdlib / msaWfa) under
 * AddressSanitizer + UndefinedBehaviorSanitizer; any out-of-bounds access, use of uninitialised stack as an index, signed
 * overflow or misaligned access aborts with a report.  Built and run by `make -C oracle sanitize` (tests/test_oracle_golden.py). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "delly_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static char base(void) { return "ACGT"[rnd() & 3]; }

int main(void) {
  enum { G = 40000, NJ = 24, NR = 6 };
  char* chr = (char*)malloc(G);
  for (int i = 0; i < G; ++i) chr[i] = base();
  for (int i = 5000; i < 5080; ++i) chr[i] = 'A';                  /* a homopolymer and a (CA)n run on breakpoints */
  for (int i = 9000; i < 9100; ++i) chr[i] = (i & 1) ? 'A' : 'C';
  chr[12000] = 'N';
  chr[15000] = 'a';
  dellyhip_params P = {5, -4, -10, -1, 2, 13, 1000, 100, 0.95f, 0};
  dellyhip_junction J[NJ];
  uint64_t off[NJ * NR + 1];
  size_t cap = (size_t)NJ * NR * 400, used = 0;
  char* blob = (char*)malloc(cap);
  int nseq = 0;
  off[0] = 0;
  for (int j = 0; j < NJ; ++j) {
    const int s = 2000 + 1500 * j, e = s + 300 + (int)(rnd() % 500), svt = (j % 6 == 5) ? 4 : (j % 6 == 4 ? 3 : 2);
    memset(&J[j], 0, sizeof J[j]);
    J[j].svid = j; J[j].svt = svt; J[j].chr = 0; J[j].chr2 = 0; J[j].sv_start = s; J[j].sv_end = (svt == 4) ? s + 1 : e;
    J[j].ins_len = (svt == 4) ? 40 : 0;
    J[j].seq_first = (uint64_t)nseq;
    const int nr = 2 + (int)(rnd() % (NR - 1));
    J[j].n_seq = nr;
    for (int r = 0; r < nr; ++r) {
      const int o = (int)(rnd() % 40), len = 120 + (int)(rnd() % 40);
      char* dst = blob + used;
      int k = 0;
      for (int i = 0; i < len; ++i) {
        const int p = o + i;                                        /* ALT haplotype: left flank | (insertion) | right flank */
        char c = (p < 75) ? chr[s - 75 + p] : (svt == 4 ? (p < 115 ? base() : chr[s + p - 115]) : (svt == 3 ? chr[s + (p - 75)] : chr[e + (p - 75)]));
        if (rnd() % 100 == 0) c = base();
        dst[k++] = c;
      }
      used += (size_t)k;
      off[++nseq] = used;
    }
  }
  const char* chrs[1] = {chr};
  const int64_t lens[1] = {G};
  dellyhip_result* R = (dellyhip_result*)calloc(NJ, sizeof *R);
  size_t ocap = 1 << 22;
  char* out = (char*)malloc(ocap);
  uint64_t oused = 0;
  int rc = dor_refine_batch(&P, 1, chrs, lens, NJ, J, blob, off, R, out, ocap, &oused, /*with_msa*/1, /*want_alignment*/1, /*threads*/2);
  int ok = 0;
  for (int j = 0; j < NJ; ++j) ok += R[j].ok;
  printf("sanitize_selftest: msa + alignConsensus over %d junctions rc=%d refined=%d blob=%llu\n", NJ, rc, ok, (unsigned long long)oused);
  /* given-consensus path on the first read of every junction */
  dellyhip_junction J1[NJ];
  uint64_t off1[NJ + 1];
  for (int j = 0; j < NJ; ++j) { J1[j] = J[j]; J1[j].n_seq = 1; J1[j].seq_first = (uint64_t)j; }
  /* (re-pack: one sequence per junction) */
  char* blob1 = (char*)malloc(cap);
  size_t u1 = 0;
  off1[0] = 0;
  for (int j = 0; j < NJ; ++j) {
    const uint64_t a = off[J[j].seq_first], b = off[J[j].seq_first + 1];
    memcpy(blob1 + u1, blob + a, (size_t)(b - a));
    u1 += (size_t)(b - a);
    off1[j + 1] = u1;
  }
  rc |= dor_refine_batch(&P, 1, chrs, lens, NJ, J1, blob1, off1, R, out, ocap, &oused, 0, 1, 1);
  /* primitives at their edges */
  char rows[4096];
  int len = 0, diag[5];
  (void)dor_long_needle("ACGT", 4, "A", 1, rows, 2048, &len, diag);
  (void)dor_long_needle("A", 1, "ACGTACGTAC", 10, rows, 2048, &len, diag);
  (void)dor_lcs("", 0, "ACGT", 4);
  (void)dor_longest_homology("AAAA", 4, "AAAT", 4, -1);
  char rc4[5] = "ACGN";
  dor_reverse_complement(rc4, 4);
  printf("sanitize_selftest: done rc=%d\n", rc);
  free(chr); free(blob); free(blob1); free(R); free(out);
  return rc ? 1 : 0;
}
