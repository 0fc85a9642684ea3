/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle.  Never imported, linked or
 * executed by the product path (delly_amd/, libdellyhip.so); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Plain-C restatement of the reference's split-read refinement algorithm
 * (dellytools/delly v2.5.1).  Pinned against the reference itself:
 * oracle/_ref/libdelly_ref.so compiles the reference's own headers, and
 * tests/test_oracle_golden.py + tests/golden/ hold every function below to
 * bit-identical outputs on seeded inputs (the reference ships no tests or
 * golden vectors of its own -- SURVEY.md F8).
 *
 * edlib (vendored by the reference, src/edlib.cpp) is restated as the exact
 * unit-cost DP it evaluates, including additional equalities and the
 * Hirschberg split of obtainAlignment; msaEdlib is restated on top of it.
 * msaWfa (long-read insertions) is restated too (dor_msa_wfa).
 */
#ifndef DELLY_ORACLE_H
#define DELLY_ORACLE_H

#include <stdint.h>
#include "../include/dellyhip.h"

#ifdef __cplusplus
extern "C" {
#endif

int dor_lcs(const char* s1, int m, const char* s2, int n);
void dor_reverse_complement(char* s, int n);
int dor_longest_homology(const char* a, int la, const char* b, int lb, int thr);
/* returns 1 found / 0 not found / -1 cap too small; diag[5] (may be NULL) =
 * {mat[m][n], bestScore, consLeft, refLeft, refRight} */
int dor_long_needle(const char* s1, int m, const char* s2, int n, char* rows, int cap, int* len,
                    int* diag);
/* edlibAlign(q, t, {-1, mode, task}); mode 0 NW/1 SHW/2 HW, task 0 DISTANCE/1 LOC/2 PATH;
 * out[4] = {editDistance, numLocations, endLocations[0], startLocations[0]}; returns alignmentLength or <0 */
int dor_edlib_align(const char* q, int qn, const char* t, int tn, int mode, int task, int* out,
                    unsigned char* aln, int cap);
/* splitAlign + row swap (split.h:480-552); rows[0..len) consensus row, rows[cap..) reference row;
 * internals[5] = {csStart, csEnd, bestJoin, leftEnd, rightStart} */
int dor_split_align(const char* cons, int m, const char* ref, int n, char* rows, int cap, int* len,
                    int* internals);
int dor_gotoh(const dellyhip_params* p, const char* a1, int r1, int m, const char* a2, int r2, int n,
              char* out, int cap, int* len);
int dor_consensus(const dellyhip_params* p, const char* a, int r, int m, char* cs, int cap);
int dor_guide_tree(int n_reads, const char* blob, const uint64_t* off, int* dflat, int* pflat);
int dor_msa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs,
            int cap, int* cs_len);
/* msaEdlib(c, sps, cs)  src/assemble.h:383-473 */
int dor_msa_edlib(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, char* cs,
                  int cap, int* cs_len);
/* msaWfa(c, sps, cs, prefix, suffix)  src/assemble.h:547-726 */
int dor_msa_wfa(const dellyhip_params* p, int n_reads, const char* blob, const uint64_t* off, const char* prefix, int pn,
                const char* suffix, int sn, char* cs, int cap, int* cs_len);
/* worker body of process_batch, src/coverage.h:418-434 (split-read genotyping classifier) */
int dor_classify_reads(const dellyhip_params* p, uint64_t n_jobs, const dellyhip_align_job* jobs, const char* blob,
                       dellyhip_align_result* out, int n_threads, int with_dist, double* worker_seconds);
/* _editDistanceNW src/genotype.h:21-30 for every pair (long-read genotyping) */
int dor_edit_distance_nw_batch(uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob, int32_t* out, int n_threads,
                               double* worker_seconds);
/* per-SV body of _generateProbes, src/coverage.h:196-258 */
int dor_generate_probes(const dellyhip_params* p, int n_chr, const char* const* chr_seq, const int64_t* chr_len, int n_junc,
                        const dellyhip_junction* junc, const char* blob, const uint64_t* off, dellyhip_probes* probes,
                        char* out_blob, uint64_t out_cap, uint64_t* out_used);
int dor_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                     const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                     const char* blob, const uint64_t* off, dellyhip_result* results,
                     char* out_blob, uint64_t out_cap, uint64_t* out_used, int with_msa,
                     int want_alignment, int n_threads);

/* bench.py cpu_baseline timer: `reps` passes of the worker loop, results discarded, clock inside (seconds) */
int dor_time_refine_batch(const dellyhip_params* p, int n_chr, const char* const* chr_seq,
                          const int64_t* chr_len, int n_junc, const dellyhip_junction* junc,
                          const char* blob, const uint64_t* off, int with_msa, int n_threads, int reps,
                          double* seconds, int64_t* n_ok);

#ifdef __cplusplus
}
#endif
#endif
