/*
 * dellyhip.h -- C-ABI of the MI355X-native split-read refinement path.
 *
 * Drop-in boundary for the one data-parallel hot path of dellytools/delly
 * (v2.5.1): per-junction consensus building + consensus-vs-reference split
 * alignment.  The reference has no FFI layer for this path (C++ header
 * templates in namespace torali); the functions below are what a maintainer
 * binds at the loop bodies of
 *     src/shortpe.h:175-201 / :243-268      (sr: msa() + alignConsensus())
 *     src/assemble.h:833-872                 (lr: msaEdlib()/alignConsensus())
 * Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, caller owns every buffer, no callee allocation
 *     crosses the ABI (matches std::string& out-parameters of the reference).
 *   - return value: 0 = ok, <0 = infrastructure error (DELLYHIP_E_*).  The
 *     per-junction boolean of alignConsensus()/longNeedle() is in result.ok.
 *     There is NO CPU fallback inside the library: if no gfx950 device or the
 *     code object is unusable, calls fail with DELLYHIP_E_NODEVICE.
 *   - coordinates are 0-based genomic offsets exactly as in
 *     torali::StructuralVariantRecord (src/tags.h:93-130).
 */
#ifndef DELLYHIP_H
#define DELLYHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DELLYHIP_VERSION 1

#define DELLYHIP_OK 0
#define DELLYHIP_E_NODEVICE (-1)  /* no usable gfx950 device / kernel image      */
#define DELLYHIP_E_ARG (-2)       /* malformed arguments                          */
#define DELLYHIP_E_RUNTIME (-3)   /* HIP runtime error (message: dellyhip_last_error) */
#define DELLYHIP_E_LIMIT (-4)     /* a junction exceeds a compiled kernel limit   */
#define DELLYHIP_E_NOMEM (-5)

/* Duck-typed TConfig fields read by the path:
 * aliscore (src/delly.h:393), minCliqueSize, flankQuality (:394),
 * minimumFlankSize (:397), indelsize (:398), minConsWindow. */
typedef struct dellyhip_params {
  int32_t match;              /* c.aliscore.match     5  */
  int32_t mismatch;           /* c.aliscore.mismatch -4  */
  int32_t gap_open;           /* c.aliscore.go      -10  */
  int32_t gap_extend;         /* c.aliscore.ge       -1  */
  int32_t min_clique_size;    /* c.minCliqueSize (sr default 2, -z) */
  int32_t minimum_flank_size; /* c.minimumFlankSize  13 sr / 100 lr */
  int32_t indelsize;          /* c.indelsize       1000 sr / 10000 lr */
  int32_t min_cons_window;    /* c.minConsWindow    100 sr / 1000 lr */
  float flank_quality;        /* c.flankQuality    0.95 sr / 0.9 lr */
  int32_t reserved;          /* bit 0: the `realign` argument of alignConsensus (src/split.h:644-646,
                              * orientation test :564-572; long-read call sites pass true).
                              * bit 2 (DELLYHIP_COMPACT_ALLELES): compact result payload -- where the exact "REF,ALT" alleles of
                              * src/split.h:606-624 are plain substrings of the reference window and the consensus (every junction
                              * the sparse longNeedle kernels finish: letters A, C, G, T, N only), their bytes are NOT produced;
                              * the record carries allele_len = -(length) and dellyhip_recut_alleles() rebuilds them on the host
                              * from c_start / c_end / r_start / r_end.  ~70 % of a deletion's result bytes (bit 1 is internal) */
} dellyhip_params;

/* One SV candidate ("junction").  Mirrors the fields of
 * torali::StructuralVariantRecord that the path reads (src/tags.h:93-130):
 * chr, svStart, chr2, svEnd, svt, insLen (+ consensus when given). */
typedef struct dellyhip_junction {
  int32_t svid;      /* caller tag, echoed in the result                      */
  int32_t svt;       /* 0 INV3to3, 1 INV5to5, 2 DEL, 3 DUP, 4 INS, 5-8 BND    */
  int32_t chr;       /* index into the uploaded chromosome table              */
  int32_t chr2;
  int32_t sv_start;  /* sv.svStart */
  int32_t sv_end;    /* sv.svEnd   */
  int32_t ins_len;   /* sv.insLen (svt 4 only)                                 */
  int32_t n_seq;     /* #reads (refine_batch) or 1 (align_consensus_batch)    */
  uint64_t seq_first;/* index of this junction's first entry in seq_off[]     */
} dellyhip_junction;

/* What alignConsensus() writes into the StructuralVariantRecord
 * (src/split.h:626-637) plus msa()'s return value and consensus. */
#define DELLYHIP_SCORE_UNKNOWN (-(1 << 30))
typedef struct dellyhip_result {
  int32_t svid;
  int32_t ok;          /* 1: alignConsensus() returned true; 0: false          */
  int32_t sv_start;    /* sv.svStart = finalGapStart                            */
  int32_t sv_end;      /* sv.svEnd   = finalGapEnd                              */
  int32_t ci_wiggle;   /* ciposhigh = ciendhigh = -ciposlow = -ciendlow         */
  int32_t ins_len;     /* sv.insLen = cEnd - cStart - 1                         */
  int32_t cons_bp;     /* sv.consBp = cStart                                    */
  int32_t hom_len;     /* sv.homLen                                             */
  int32_t sr_support;  /* msa() return value = rows of the MSA                  */
  float sr_align_quality; /* ad.percId (float32 division, src/split.h:315)      */
  int32_t matches;     /* ma, mm of _percentIdentity (so the host can recheck)  */
  int32_t mismatches;
  int32_t c_start, c_end, r_start, r_end; /* AlignDescriptor (src/split.h:15)   */
  int32_t hom_left, hom_right;
  /* longNeedle internals (parity diagnostics; src/needle.h:104-123).  For svt 4
   * (splitAlign, src/split.h:480-538) the five slots carry csStart, csEnd,
   * bestJoin, leftEnd, rightStart instead; -1 = not reached. */
  int32_t score_unsplit;  /* mat[m][n]; DELLYHIP_SCORE_UNKNOWN when the sparse longNeedle proved the split score larger
                           * without computing it (the reference only compares the two, src/needle.h:152)          */
  int32_t score_best;     /* bestScore            */
  int32_t cons_left, ref_left, ref_right;
  int32_t cons_len;       /* |consensus| used for the split alignment          */
  int32_t ref_len;        /* |svRefStr|                                         */
  /* byte ranges in out_blob */
  uint64_t cons_off;      /* consensus (sv.consensus), cons_len bytes            */
  uint64_t allele_off;    /* sv.alleles = "REF,ALT" (src/split.h:606-624)        */
  uint64_t aln_off;       /* 2-row gapped alignment, row-major 2 x aln_len       */
  int32_t allele_len;     /* 0 when that block is not executed                   */
  int32_t aln_len;        /* columns of the alignment (0 if not requested)       */
  int32_t status;         /* 0, or DELLYHIP_E_LIMIT for this junction            */
  int32_t reserved;
} dellyhip_result;

#define DELLYHIP_REALIGN 1
#define DELLYHIP_COMPACT_ALLELES 4

typedef struct dellyhip_ctx dellyhip_ctx;

/* ---- context ----------------------------------------------------------- */

/* Creates a context bound to HIP device `device` (one ctx per GPU / process
 * rank).  Replaces the ThreadPool(c.maxThreads) of src/shortpe.h:80. */
int dellyhip_create(const dellyhip_params* params, int device, dellyhip_ctx** out);
void dellyhip_destroy(dellyhip_ctx* ctx);
const char* dellyhip_last_error(void);
/* out[0..3] = DELLYHIP_VERSION, sizeof(dellyhip_params), sizeof(dellyhip_junction),
 * sizeof(dellyhip_result): lets a binding verify its struct layout without a GPU. */
void dellyhip_abi_info(int32_t out[4]);
/* The alleles of a record produced with DELLYHIP_COMPACT_ALLELES (allele_len < 0), exactly as src/split.h:606-624 builds them:
 * "REF,ALT" with REF = toupper(chr_seq[window start + r_start - 1 .. + r_end - 1)) and ALT = consensus[c_start - 1 .. c_end - 1),
 * the window start being svStartBeg of _initBreakpoint (src/tags.h:151-172) for the junction AS SUBMITTED (`junction`: sv_start,
 * svt, ins_len before refinement).  consensus / cons_len: the record's consensus bytes (out_blob + cons_off, result->cons_len).
 * chr_seq: the caller's copy of chromosome junction->chr (the `seq` of src/shortpe.h:88).  Pure host code, no device, re-entrant.
 * Returns the number of bytes written to out (= -result->allele_len), 0 when the record has no compact alleles, or
 * DELLYHIP_E_ARG (cap too small, inconsistent record). */
int64_t dellyhip_recut_alleles(const dellyhip_params* params, const dellyhip_junction* junction, const dellyhip_result* result,
                               const char* consensus, const char* chr_seq, int64_t chr_len, char* out, uint64_t cap);
/* The same over a batch of n records (junctions[i] as submitted, results[i] / blob as returned): the alleles go to out back to
 * back, out_off[i] .. out_off[i + 1] (n + 1 entries) being record i's (empty where it has none to re-cut); chr_seq / chr_len: the
 * caller's chromosome table.  Returns the bytes written or a negative error. */
int64_t dellyhip_recut_alleles_batch(const dellyhip_params* params, int32_t n, const dellyhip_junction* junctions, const dellyhip_result* results,
                                     const char* blob, const char* const* chr_seq, const int64_t* chr_len, int32_t n_chr, char* out,
                                     uint64_t cap, uint64_t* out_off);
/* Default parameters of `delly sr` (src/delly.h:393-398) and `delly lr`
 * (src/tegua.h:237-241). */
void dellyhip_default_params_sr(dellyhip_params* p);
void dellyhip_default_params_lr(dellyhip_params* p);

/* A second context on the same device that SHARES share_with's resident chromosomes (one ref-counted table: a
 * dellyhip_set_chromosome through either context is seen by both; the chromosomes are freed with the last context).
 * Own HIP stream, scratch area and work counters, so the two can run batches concurrently.  params NULL = the same
 * parameters.  This is what the worker threads of the reference's ThreadPool (src/shortpe.h:80,175-201) use: one
 * context per thread, ONE copy of the genome per GPU. */
int dellyhip_create_shared(dellyhip_ctx* share_with, const dellyhip_params* params, dellyhip_ctx** out);

/* Two HIP streams (hipStream_t) of ctx's device that were VERIFIED to run side by side.  The runtime maps HIP streams onto
 * a few hardware queues and two streams that share one execute in order; with more streams alive than hardware queues
 * (a torch.cuda.Stream() alone creates 32) two freshly created streams may well share.  The library probes candidates
 * once per device and process (a kernel on one waits <= 0.5 ms for a flag only a kernel on the other sets) and keeps the
 * pair for the slots of every dellyhip_stream.  A caller that keeps two resident batches in flight through
 * dellyhip_batch_run (two contexts from dellyhip_create_shared, one per stream) should pass these two streams:
 * 34 instead of 22.5 M alignments/s at 10 000 junctions per batch (the tail of one launch under the head of the next).
 * The streams belong to the library; do not destroy them. */
int dellyhip_compute_streams(dellyhip_ctx* ctx, void* out[2]);

/* Pins [p, p + bytes) of the CALLER's host memory for the device (hipHostRegister), so that dellyhip_batch_fetch /
 * dellyhip_gather_results write into it at the PCIe rate -- e.g. a POSIX shared-memory segment that the process which
 * writes the VCF has mapped too: on one node every rank then returns its results over its OWN PCIe link and no
 * collective carries them.  For bindings that do not link the HIP runtime themselves. */
int dellyhip_host_register(dellyhip_ctx* ctx, void* p, uint64_t bytes);
int dellyhip_host_unregister(dellyhip_ctx* ctx, void* p);

/* Memory policy: device and pinned host blocks the library releases (freed batches, destroyed contexts and streams)
 * are parked in a process-wide free list per device and handed out again -- they are not returned to the HIP runtime
 * while the process lives, so the footprint is the high-water mark of what was alive at once (parked blocks are capped
 * at half of the device's memory -- DELLYHIP_POOL_LIMIT_MB overrides -- beyond which the oldest go back to the runtime).  (hipMalloc / hipFree
 * synchronise the device, and allocations made after earlier ones were freed were measured to download at a fraction
 * of the PCIe rate: CHANGELOG.md 1b.)  dellyhip_trim_memory waits for the device, returns every parked block of ctx's
 * device to the runtime and reports the bytes released. */
uint64_t dellyhip_trim_memory(dellyhip_ctx* ctx);

/* Keeps chromosome `chr` resident in HBM.  Replaces the per-chromosome
 * faidx_fetch_seq() buffer `seq` that src/shortpe.h:88 hands to
 * alignConsensus(c, hdr, seq, sndSeq, sv), and hdr->target_len[chr]
 * (src/tags.h:156-169).  A whole genome (3.1 GB) fits many times in 288 GB. */
int dellyhip_set_chromosome(dellyhip_ctx* ctx, int32_t chr, const char* seq, int64_t len);

/* ---- batched hot path -------------------------------------------------- */

/* msa(c, seqStore[svid], sv.consensus) + alignConsensus(c, hdr, seq, sndSeq, sv)
 * for every junction: the loop body of src/shortpe.h:183-197 / :248-266.
 *   seq_blob / seq_off : reads of all junctions, concatenated, in the HOST's
 *                        std::unordered_set iteration order;
 *                        read i spans seq_blob[seq_off[i], seq_off[i+1]).
 *   out_blob           : receives consensus (+ alignment rows if want_alignment)
 *   out_blob_cap/len   : capacity in, bytes used out.
 */
int dellyhip_refine_batch(dellyhip_ctx* ctx, int32_t n_junctions,
                          const dellyhip_junction* junctions, const char* seq_blob,
                          const uint64_t* seq_off, uint64_t n_seq, dellyhip_result* results,
                          char* out_blob, uint64_t out_blob_cap, uint64_t* out_blob_len,
                          int want_alignment);

/* alignConsensus(c, hdr, seq, sndSeq, sv, realign=false) only
 * (src/split.h:644-672), consensus supplied by the caller: each junction has
 * n_seq == 1 and its sequence is sv.consensus.  This is BASELINE.json's unit U. */
int dellyhip_align_consensus_batch(dellyhip_ctx* ctx, int32_t n_junctions,
                                   const dellyhip_junction* junctions, const char* seq_blob,
                                   const uint64_t* seq_off, uint64_t n_seq,
                                   dellyhip_result* results, char* out_blob,
                                   uint64_t out_blob_cap, uint64_t* out_blob_len,
                                   int want_alignment);

/* ---- pipelined host-buffer path ------------------------------------------
 * The loop of src/shortpe.h:175-201 over MANY batches (one per chromosome, or chunks of a chromosome's junctions) with
 * the host work, the PCIe copies and the kernels of consecutive batches overlapped.  A stream owns `depth` slots (each:
 * a context of its own sharing the resident chromosomes, pinned staging for inputs and outputs, device buffers that
 * are grown geometrically and reused -- no allocation per batch once warmed up).
 *   submit : validates + routes the junctions, copies the inputs into pinned staging, enqueues ONE H2D copy, the
 *            kernels and the device-side compaction (the D2H copies follow as soon as a later call finds the
 *            compaction finished); returns without waiting.
 *            with_msa as dellyhip_batch_upload (0 = consensus given, 1 = msa() first, 2 = the long-read loop body; the
 *            modes whose routing needs the consensus lengths on the host -- 2, and 1 with insertions or long-read
 *            parameters -- wait for the MSA stage inside submit).  DELLYHIP_E_ARG when every slot is in flight or held.
 *   collect: waits for the OLDEST submitted batch and hands out pointers into its pinned output block: n records
 *            (blob offsets relative to *blob) and the consensus / "REF,ALT" (/ alignment) bytes.  The pointers stay
 *            valid until the NEXT collect (or dellyhip_stream_release) on this stream: the slot is not reused before that.
 * Typical loop, depth d:  submit(0 .. d - 2); for k: collect(k) -> consume; submit(k + d - 1).  depth 1 .. 8; a batch of
 * 10 000 short-read junctions is in flight for about 1 ms and a new one can start every 0.3 ms, so depth 6 is what
 * saturates the GPU (32 M junctions/s; depth 4: 28 M/s, depth 3: 18 M/s).
 * One stream per calling thread (several streams, one per thread, may share a genome through dellyhip_create_shared). */
typedef struct dellyhip_stream dellyhip_stream;
int dellyhip_stream_create(dellyhip_ctx* ctx, int32_t depth, int32_t with_msa, int32_t want_alignment, dellyhip_stream** out);
void dellyhip_stream_destroy(dellyhip_stream* stream);
int dellyhip_stream_submit(dellyhip_stream* stream, int32_t n_junctions, const dellyhip_junction* junctions,
                           const char* seq_blob, const uint64_t* seq_off, uint64_t n_seq, uint64_t tag);
int dellyhip_stream_collect(dellyhip_stream* stream, const dellyhip_result** results, const char** blob,
                            uint64_t* blob_len, int32_t* n_junctions, uint64_t* tag);
/* on != 0: a seq_blob passed to dellyhip_stream_submit that lies in PINNED host memory (dellyhip_host_register, hipHostMalloc) is
 * read by the copy engine in place instead of being copied into the stream's staging arena first -- the sequence bytes are three
 * quarters of a batch's upload and the staging copy was the largest part of a submit's host time.  The caller must then leave
 * those bytes unchanged until the batch has been collected.  Pageable memory is staged as before (no error).  Default: off. */
int dellyhip_stream_zero_copy(dellyhip_stream* stream, int32_t on);
/* Errors: a submit that fails leaves the stream as it was before the call (anything it had enqueued has been waited for,
 * the slot is free).  A collect that fails DROPS the batch it was waiting for -- the stream's pending count goes down by
 * one, the slot is free again and the following collects return the following batches; the synchronous entry points
 * (dellyhip_refine_batch, ...) therefore stay usable after a failed call. */
/* Gives the output block of the last collect() back before the next collect() does (a caller that has copied what it
 * needs; with depth 1 this is what allows the next submit). */
void dellyhip_stream_release(dellyhip_stream* stream);
/* Where the host time of this stream went since its creation (or the last reset), in seconds: out[0] validation +
 * routing + staging copies, out[1] kernel launches, out[2] enqueueing compaction + downloads, out[3] waiting in collect;
 * out[4] batches that needed the dense kernels for junctions the sparse kernel left (routed at collect time),
 * out[5] batches whose blob download had to be topped up (the size estimate was short). */
void dellyhip_stream_stats(dellyhip_stream* stream, double out[6], int32_t reset);
/* batches submitted and not yet collected */
int dellyhip_stream_pending(dellyhip_stream* stream);

/* ---- device-resident batches (bench / pipelined callers) ---------------- */

/* Uploads a batch once; run_resident() then executes the same work as
 * dellyhip_align_consensus_batch / dellyhip_refine_batch with inputs already in
 * HBM and results left in HBM; fetch_results() copies them back. `stream` is a
 * hipStream_t passed as void* (0 = the context's own stream).  with_msa: 0 = the
 * consensus is given (unit U), 1 = msa() first (short reads), 2 = the long-read
 * loop body (msaEdlib / msaWfa, see dellyhip_refine_batch_lr).  Runs of ONE context
 * share its scratch area and are ordered after each other even across streams;
 * to overlap batches use one context per stream. */
typedef struct dellyhip_batch dellyhip_batch;
int dellyhip_batch_upload(dellyhip_ctx* ctx, int32_t n_junctions,
                          const dellyhip_junction* junctions, const char* seq_blob,
                          const uint64_t* seq_off, uint64_t n_seq, int with_msa,
                          dellyhip_batch** out);
int dellyhip_batch_run(dellyhip_ctx* ctx, dellyhip_batch* b, void* stream);
int dellyhip_batch_sync(dellyhip_ctx* ctx, dellyhip_batch* b);
int dellyhip_batch_fetch(dellyhip_ctx* ctx, dellyhip_batch* b, dellyhip_result* results,
                         char* out_blob, uint64_t out_blob_cap, uint64_t* out_blob_len);
/* dellyhip_batch_fetch in two halves, for a caller that keeps the device busy while results travel (the N > 1 step of
 * bench.py: every rank returns step k-1's records into its own pinned shared-memory segment while step k runs; the place
 * of the reference's per-thread push into the shared svs vector, src/shortpe.h:175-201).
 * _begin queues, on the stream of the batch's last run and therefore behind its kernels, the device-side compaction of
 * dellyhip_batch_fetch followed by a kernel that WRITES the records (blob offsets rebased exactly as dellyhip_batch_fetch
 * returns them) and the used blob bytes straight into `results` / `out_blob`; it returns without waiting.  Both areas must be
 * pinned, device-visible host memory (dellyhip_host_register, hipHostMalloc): DELLYHIP_E_ARG otherwise.  They belong to the
 * library until _end.  The batch may be run again before _end (the run is ordered behind the queued fetch); one fetch per
 * batch may be in flight.
 * _end waits for that fetch only (not for later runs), *out_blob_len = blob bytes; DELLYHIP_E_ARG "out_blob too small" (with
 * the size needed in *out_blob_len; the records are complete, the blob area untouched) when out_blob_cap was not enough. */
int dellyhip_batch_fetch_begin(dellyhip_ctx* ctx, dellyhip_batch* b, dellyhip_result* results,
                               char* out_blob, uint64_t out_blob_cap);
int dellyhip_batch_fetch_end(dellyhip_ctx* ctx, dellyhip_batch* b, uint64_t* out_blob_len);
void dellyhip_batch_free(dellyhip_ctx* ctx, dellyhip_batch* b);
/* Device pointer / byte size of the result records (n x dellyhip_result) left in
 * HBM by dellyhip_batch_run: lets a multi-GPU caller hand them to an RCCL
 * gather without a host round trip. */
int dellyhip_batch_device_results(dellyhip_ctx* ctx, dellyhip_batch* b, void** dptr, uint64_t* bytes);
/* Average duration in milliseconds of the dominant kernel (split alignment)
 * over the launches since the last call, measured with hipEvents on the
 * launch stream; also returns the launch count. */
int dellyhip_batch_kernel_ms(dellyhip_ctx* ctx, dellyhip_batch* b, double* ms_split,
                             double* ms_msa, int32_t* launches);
/* Average duration (ms) of the dominant kernel alone over the launches covered by the last
 * dellyhip_batch_kernel_ms() call (HIP events on the launch stream): split_sparse_kernel (sparse longNeedle, one
 * junction per wavefront) when every short-read junction of the batch is offered to it -- consensus <= 254 bp, window <=
 * 1280 bp, window + consensus <= 1535 bp, DELLYHIP_SR_SPARSE != 0 --, else the packed dense DP kernels (split_quad_kernel<KQ, KP>, four
 * junctions per wavefront; split_pair_kernel<K> for consensus sequences of 160 .. 319 bp). */
int dellyhip_batch_dp_kernel_ms(dellyhip_ctx* ctx, dellyhip_batch* b, double* ms_dp);

/* Junctions that split_sparse_kernel did not finish -- shapes beyond its tile, letters outside A, C, G, T, N, deficits
 * beyond its 32 levels -- and handed to the dense kernels, in the LAST RUN ON THIS CONTEXT (the counter belongs to the
 * context, not to the batch: with several resident batches per context call this after the run of `b` and before the next
 * dellyhip_batch_run on the context; the pipelined path reports the number per batch with its results).  0 when the sparse
 * kernel is off or the run had no short-read junctions.  The sparse kernel's cost grows with a junction's deficit, the dense kernels' does
 * not (like the reference, src/needle.h:64-115): this is the number that says which regime a batch ran in. */
int dellyhip_batch_sparse_left(dellyhip_ctx* ctx, dellyhip_batch* b, int32_t* left);
/* Long-read batches: the dense strip fallback of the strip kernel runs on teams of wavefronts beside it (CHANGELOG.md 3.7).
 * out[0] = teams launched for this batch (0: none -- DELLYHIP_LR_TEAMS=0, or no long-read junction), out[1] = junctions of
 * the last run the teams swept, out[2] = claims the teams made on the list (>= out[0]: every team ends with one that finds nothing), out[3] = 1 if a team gave up waiting
 * (the junction it was on carries status DELLYHIP_E_RUNTIME; since round 5 dellyhip_batch_sync -- and with it fetch, the stream
 * and the gather -- FAILS such a batch with DELLYHIP_E_RUNTIME: junctions still on the teams' list would otherwise come back
 * as "not refined", indistinguishable from a consensus the reference rejects; the failure is sticky until the batch's next
 * run).  Synchronises the batch; for such a batch out[] is filled (out[3] = 1) AND the sync's error code is returned. */
int dellyhip_batch_lr_team_stats(dellyhip_ctx* ctx, dellyhip_batch* b, int32_t out[4]);
/* msa() batches: how the LAST RUN ON THIS CONTEXT went through the MSA kernels (counters of the context, like
 * dellyhip_batch_sparse_left).  out[0] = junctions the score-table kernel deferred to the direct-float kernel (a node with
 * more column types than the table holds, a NaN score; src/align.h:104-110 evaluated per cell there), out[1] = junctions
 * beyond the standard instance's shapes that went to the second instance (more / longer reads, wider nodes), out[2] =
 * wavefronts per junction of the launch (1, or a team of 2 / 4 for launches that do not fill the chip), out[3] = its blocks.
 * Synchronises the batch. */
int dellyhip_batch_msa_stats(dellyhip_ctx* ctx, dellyhip_batch* b, int32_t out[4]);

/* ---- multi-GPU: junction sharding + gather of the results to one rank ------------------------ */

/* owner[i] = rank (0 .. world-1) of junction i, balanced by predicted cost -- N^2 L^2 for the pairwise stage of
 * msa()/msaEdlib(), (N-1) L^2 for the progressive alignments, |consensus| x |svRefStr| for the split alignment
 * (N reads of mean length L; consensus ~ L for short reads, given for n_seq == 1) -- longest first onto the lightest
 * rank.  Pure host arithmetic (no GPU needed); every rank computes the same assignment from the same junction list.
 * Junctions are independent (src/shortpe.h:183-197 touches only svs[svid]), so any assignment gives the same results. */
int dellyhip_shard_by_cost(const dellyhip_params* params, int32_t n_junctions, const dellyhip_junction* junctions,
                           const uint64_t* seq_off, uint64_t n_seq, int32_t world, int32_t* owner);

/* One communicator per process (one process per GPU), RCCL over xGMI.  Rank 0 fills id128 (128 bytes,
 * ncclUniqueId) with dellyhip_comm_unique_id and hands it to the other ranks by any means (file, MPI, torch.distributed
 * store); every rank then calls dellyhip_comm_create.  world == 1 needs no id and never loads RCCL. */
typedef struct dellyhip_comm dellyhip_comm;
int dellyhip_comm_unique_id(void* id128);
int dellyhip_comm_create(dellyhip_ctx* ctx, const void* id128, int32_t rank, int32_t world, dellyhip_comm** out);
void dellyhip_comm_destroy(dellyhip_comm* comm);

/* The same communicator over POSIX shared memory instead of RCCL, for the ranks of ONE node: RCCL refuses a communicator
 * whose ranks share a device, so this is the transport when several processes drive one GPU (oversubscribed runs, the
 * two-process tests on a one-GPU box); device buffers are staged through the sender's pinned outbox segment.  `name`
 * (<= 96 bytes, no '/') must be unique to the job -- rank 0 creates "/dellyhip_<name>_ctl", the others wait for it
 * (DELLYHIP_LINK_TIMEOUT_S, default 120 s, bounds every wait: a dead peer is an error, not a hang).  ctx == NULL gives a
 * device-less communicator on which only the two exchanges below can run (verification of the protocol without GPUs).
 * dellyhip_gather_results(_device) run the same protocol code on either transport. */
int dellyhip_comm_create_hostlink(dellyhip_ctx* ctx, const char* name, int32_t rank, int32_t world, dellyhip_comm** out);

/* rank / world as created, transport_ranks = what the transport itself reports (ncclCommCount; attached processes of a
 * hostlink; 1 without a transport), kind16 = "rccl" | "hostlink" | "none".  Any out pointer may be NULL. */
int dellyhip_comm_info(dellyhip_comm* comm, int32_t* rank, int32_t* world, int32_t* transport_ranks, char* kind16);

/* The two collective steps in front of the payload of dellyhip_gather_results, exposed so that the abort protocol can be
 * verified on its own: (1) every rank contributes (count, bytes) -- or `failed` != 0 -- and receives all ranks' pairs in
 * all[2 * world]; if ANY rank failed EVERY rank returns an error (the failing rank DELLYHIP_E_RUNTIME with its own message,
 * the others naming the rank) and nobody goes on to the payload; (2) the root says whether it could size its receive
 * areas; if not, every rank returns DELLYHIP_E_NOMEM.  Collective: all ranks call them in the same order.  ctx may be
 * NULL for a device-less hostlink. */
int dellyhip_comm_exchange_sizes(dellyhip_ctx* ctx, dellyhip_comm* comm, uint64_t count, uint64_t bytes, int32_t failed, uint64_t* all);
int dellyhip_comm_exchange_ready(dellyhip_ctx* ctx, dellyhip_comm* comm, int32_t root, int32_t root_failed);

/* gatherv of one opaque payload per rank on the same protocol (size exchange, root readiness, one send / receive group):
 * the root receives the ranks' payloads back to back in rank order in out[0 .. sum sizes), sizes[r] = bytes of rank r (every
 * rank gets the sizes).  A root whose out_cap is short makes every rank return DELLYHIP_E_NOMEM before any payload moves.
 * With a context the pointers are device (or pinned / registered host) memory; on a device-less hostlink plain host memory.
 * The gather of per-rank classifier results or probe blobs to the rank that merges them. */
int dellyhip_comm_gather_bytes(dellyhip_ctx* ctx, dellyhip_comm* comm, int32_t root, const void* mine, uint64_t bytes,
                               void* out, uint64_t out_cap, uint64_t* sizes);

/* Gathers what dellyhip_batch_fetch returns -- result records AND consensus / "REF,ALT" (/ alignment) bytes -- of every
 * rank's batch to `root`: the all-gatherv of SURVEY.md 8e (RCCL has none: one ncclAllGather of the counts, then grouped
 * ncclSend / ncclRecv of the two variable-length pieces).  Collective: every rank calls it with its own batch (run
 * before; n = 0 is fine).  On root: results[0 .. sum counts) in rank order (rank r's junctions in its batch order; sort
 * by svid for the CPU order), blob offsets rebased into out_blob, counts[r] = junctions of rank r.  Other ranks pass
 * NULL buffers.  results_cap in records, out_blob_cap in bytes (DELLYHIP_E_ARG with the needed sizes in *n_results /
 * *out_blob_len when too small). */
int dellyhip_gather_results(dellyhip_ctx* ctx, dellyhip_comm* comm, dellyhip_batch* b, int32_t root,
                            dellyhip_result* results, uint64_t results_cap, uint64_t* n_results, char* out_blob,
                            uint64_t out_blob_cap, uint64_t* out_blob_len, int32_t* counts);

/* What the root does to the gathered records after the exchange, exposed so that it can be verified without GPUs: the
 * records of rank r (counts[r] of them, in rank order) point into a compact blob of bytes[r] bytes whose pieces lie back
 * to back in record order; their cons_off / allele_off / aln_off become offsets into the concatenation of all ranks' blobs.
 * Fails (DELLYHIP_E_RUNTIME) when a rank's records do not add up to its blob.  Pure host arithmetic. */
int dellyhip_rebase_gathered(dellyhip_result* results, uint64_t n_results, int32_t world, const uint64_t* counts, const uint64_t* bytes);

/* The same exchange, results left in the root's HBM (pipelined callers: the gather of batch k overlaps the refinement
 * of batch k+1; a later dellyhip_gather_results / hipMemcpy moves them to the host): *d_records = n_results records in
 * rank order, *d_blob = the ranks' compact blobs back to back (offsets inside the records are still per rank, as
 * dellyhip_batch_run left them).  The pointers stay valid until the next gather on this communicator. */
int dellyhip_gather_results_device(dellyhip_ctx* ctx, dellyhip_comm* comm, dellyhip_batch* b, int32_t root,
                                   const void** d_records, uint64_t* n_results, const void** d_blob, uint64_t* blob_bytes);

/* Long-read flavour of dellyhip_refine_batch: the loop body of src/assemble.h:833-872 for
 * non-insertion junctions -- msaEdlib(c, seqStore[svid], consensus) (src/assemble.h:383-473)
 * followed by alignConsensus(c, hdr, seq, NULL, sv, realign) with the `delly lr` parameters
 * (dellyhip_default_params_lr; realign = bit 0 of params.reserved).  Reads: at most 32 per
 * junction (maxReadPerSV, default 15), each <= 32000 bytes; consensus <= 12799.  svt 4 junctions take the
 * insertion branch of the loop (:855-860): msaWfa with the reference anchors around svStart (<= 4096 bytes
 * each: minConsWindow), then alignConsensus with realign = false (reads >= 8 bytes; superstring / alignment
 * <= min(32766, 2 x longest read + 2048) columns). */
int dellyhip_refine_batch_lr(dellyhip_ctx* ctx, int32_t n_junctions, const dellyhip_junction* junctions,
                             const char* seq_blob, const uint64_t* seq_off, uint64_t n_seq,
                             dellyhip_result* results, char* out_blob, uint64_t out_blob_cap,
                             uint64_t* out_blob_len, int want_alignment);

/* ---- split-read genotyping: the read classifier -------- */

/* One AlignJob of src/coverage.h:87-96: a breakpoint-spanning read and the two probes of the
 * SV it spans.  The three strings live in one byte blob (several jobs may share a probe or a
 * read); file_index / sv_id / qual are carried through to the result as in :418-434. */
typedef struct dellyhip_align_job {
  uint64_t cons_off;   /* consProbe  (ALT probe, src/coverage.h:255) */
  uint64_t ref_off;    /* refProbe   (:256)                          */
  uint64_t seq_off;    /* read sequence, already orientation-adjusted (:518-522) */
  uint32_t cons_len;
  uint32_t ref_len;
  uint32_t seq_len;
  uint32_t file_index;
  uint32_t sv_id;
  uint8_t qual;        /* rec->core.qual */
  uint8_t pad_[3];
} dellyhip_align_job;

/* AlignResult of src/coverage.h:98-105 plus the two edlib distances behind it. */
typedef struct dellyhip_align_result {
  uint32_t file_index;
  uint32_t sv_id;
  int32_t dist_alt;    /* edlibAlign(consProbe, read, k, HW, DISTANCE).editDistance, -1 beyond k */
  int32_t dist_ref;    /* same for refProbe */
  uint8_t type;        /* 'R', 'A' or 'N' */
  uint8_t qual;
  int16_t status;      /* 0, or DELLYHIP_E_LIMIT (probe > 6144 bytes) */
} dellyhip_align_result;

/* The worker body of process_batch, src/coverage.h:418-434, for every job:
 *   scoreAlt = _editDistanceHW(c, consProbe, sequence); scoreRef = _editDistanceHW(c, refProbe, sequence)
 *   (src/coverage.h:107-115: edlib HW distance with k = (int)(2 * flankQuality * |probe|), score =
 *   (1 - flankQuality) * |probe| / (distance + 1) in double, 0 beyond k), then type / qual as in :424-432.
 * Uses params.flank_quality of the context.  Limits: probes <= 6144 bytes (<= 64: one job per lane; <= 256: four words per
 * probe; beyond: one job per wavefront); reads: any length.
 * The merge into countMap (:438-449, order dependent) stays with the caller. */
int dellyhip_classify_reads(dellyhip_ctx* ctx, uint64_t n_jobs, const dellyhip_align_job* jobs,
                            const char* blob, uint64_t blob_len, dellyhip_align_result* results);

/* Device-resident flavour (jobs of one BAM batch, src/coverage.h:271: 131072 x threads):
 * upload once, run (stream = hipStream_t as void*, 0 = the context's stream), fetch. */
typedef struct dellyhip_jobs dellyhip_jobs;
int dellyhip_jobs_upload(dellyhip_ctx* ctx, uint64_t n_jobs, const dellyhip_align_job* jobs,
                         const char* blob, uint64_t blob_len, dellyhip_jobs** out);
int dellyhip_jobs_run(dellyhip_ctx* ctx, dellyhip_jobs* b, void* stream);
int dellyhip_jobs_sync(dellyhip_ctx* ctx, dellyhip_jobs* b);
int dellyhip_jobs_fetch(dellyhip_ctx* ctx, dellyhip_jobs* b, dellyhip_align_result* results);
void dellyhip_jobs_free(dellyhip_ctx* ctx, dellyhip_jobs* b);
/* Average duration (ms) of classify_kernel over the runs since the last call (HIP events on the launch stream). */
int dellyhip_jobs_kernel_ms(dellyhip_ctx* ctx, dellyhip_jobs* b, double* ms, int32_t* launches);

/* ---- split-read genotyping: probe generation ----------- */

/* What _generateProbes (src/coverage.h:164-263) derives per precise SV from _consRefAlignment + _findSplit
 * (:214-217): consProbeArr[bpPoint][id], refProbeArr[bpPoint][id] (:255-256) and the BpRegion (:257), for
 * bpPoint 0 (svStart side) and 1 (svEnd side).  The probe bytes are ranges of out_blob. */
typedef struct dellyhip_probes {
  int32_t svid;
  int32_t ok;              /* _consRefAlignment && _findSplit (no probes otherwise, src/coverage.h:214-217) */
  int32_t hom_left;        /* ad.homLeft / ad.homRight */
  int32_t hom_right;
  int32_t region_start[2]; /* BpRegion(regionStart, regionEnd, bppos, ...) src/coverage.h:236-253 */
  int32_t region_end[2];
  int32_t bppos[2];
  int32_t cons_len[2];
  int32_t ref_len[2];
  uint64_t cons_off[2];
  uint64_t ref_off[2];
  int32_t status;          /* 0, or DELLYHIP_E_LIMIT (long-read shapes, probe > 640 bytes) */
  int32_t reserved;
} dellyhip_probes;

/* _generateProbes' per-SV work for every junction (consensus given: n_seq == 1, as
 * dellyhip_align_consensus_batch): the same window / alignment / split detection as alignConsensus, WITHOUT
 * its early length test (src/split.h:647, which _generateProbes does not have), then the four substrings.
 * Short-read shapes only (|consensus| <= 319, |svRefStr| <= 2048). */
int dellyhip_generate_probes_batch(dellyhip_ctx* ctx, int32_t n_junctions, const dellyhip_junction* junctions,
                                   const char* seq_blob, const uint64_t* seq_off, uint64_t n_seq,
                                   dellyhip_probes* probes, char* out_blob, uint64_t out_blob_cap,
                                   uint64_t* out_blob_len);
/* The same for a resident batch that has been run (dellyhip_batch_run): probes are cut from the descriptors
 * the run left in HBM, nothing is re-aligned.  Upload the batch with (with_msa | 16) to drop the early
 * length test of alignConsensus as _generateProbes does. */
int dellyhip_batch_probes(dellyhip_ctx* ctx, dellyhip_batch* b, dellyhip_probes* probes, char* out_blob,
                          uint64_t out_blob_cap, uint64_t* out_blob_len);

/* ---- long-read genotyping: batched _editDistanceNW ------ */

/* One _editDistanceNW(query, target) call of src/genotype.h:21-30 (edlibAlign(..., k = -1, EDLIB_MODE_NW,
 * EDLIB_TASK_DISTANCE)) as made per read and breakpoint at src/genotype.h:276,284 (ref / alt allele slice vs
 * read slice, each 2 * offset bytes).  Both strings are ranges of one byte blob. */
typedef struct dellyhip_nw_job {
  uint64_t query_off;
  uint64_t target_off;
  uint32_t query_len;
  uint32_t target_len;
} dellyhip_nw_job;

/* distances[i] = editDistance of job i (max(len) if one string is empty, src/edlib.cpp:157-163).  One job per
 * 64-lane wavefront, Myers bit-vectors; pairs with both strings beyond 6144 bytes run in strips of 6144 rows. */
int dellyhip_edit_distance_nw_batch(dellyhip_ctx* ctx, uint64_t n_jobs, const dellyhip_nw_job* jobs,
                                    const char* blob, uint64_t blob_len, int32_t* distances);
/* Device-resident flavour (bench / pipelined callers), as dellyhip_jobs_*. */
typedef struct dellyhip_nwjobs dellyhip_nwjobs;
int dellyhip_nwjobs_upload(dellyhip_ctx* ctx, uint64_t n_jobs, const dellyhip_nw_job* jobs, const char* blob,
                           uint64_t blob_len, dellyhip_nwjobs** out);
int dellyhip_nwjobs_run(dellyhip_ctx* ctx, dellyhip_nwjobs* b, void* stream);
int dellyhip_nwjobs_fetch(dellyhip_ctx* ctx, dellyhip_nwjobs* b, int32_t* distances);
void dellyhip_nwjobs_free(dellyhip_ctx* ctx, dellyhip_nwjobs* b);
int dellyhip_nwjobs_kernel_ms(dellyhip_ctx* ctx, dellyhip_nwjobs* b, double* ms, int32_t* launches);

/* ---- single-item wrappers (parity tests, assemble.h / asmode.h call sites) */

/* bool longNeedle(s1, s2, align, AlignConfig<true,false>, DnaScore(1,-1,-1,-1))
 * src/needle.h:45-222 as called from src/split.h:555.  align_rows receives the
 * 2 x *aln_len gapped alignment (row-major, row stride aln_cap). */
int dellyhip_long_needle(dellyhip_ctx* ctx, const char* s1, int32_t m, const char* s2, int32_t n,
                         char* align_rows, int32_t aln_cap, int32_t* aln_len, int32_t* found);

/* bool splitAlign(cons, svRefStr, align)  src/split.h:480-538 (the svt 4 branch of _consRefAlignment, :546-552).
 * align_rows as for dellyhip_long_needle, in _consRefAlignment's orientation: row 0 = consensus, row 1 = reference
 * (splitAlign itself returns them the other way round; the caller swaps).  Short-read insertion shapes:
 * |cons| <= 319, 3 <= |svRefStr| <= 2048. */
int dellyhip_split_align(dellyhip_ctx* ctx, const char* cons, int32_t m, const char* ref, int32_t n,
                         char* align_rows, int32_t aln_cap, int32_t* aln_len, int32_t* found);

/* EdlibAlignResult edlibAlign(query, queryLength, target, targetLength,
 *                             edlibNewAlignConfig(-1, mode, task, NULL, 0))
 * src/edlib.h:242-246, src/edlib.cpp:139-300 -- the calls made by splitAlign
 * (src/split.h:485-527, HW/SHW + PATH) and _alignConsensus (src/split.h:568-569,
 * NW + DISTANCE).  mode = EdlibAlignMode (0 NW, 1 SHW, 2 HW), task =
 * EdlibAlignTask (0 DISTANCE, 1 LOC, 2 PATH).  out[4] = {editDistance,
 * numLocations, endLocations[0], startLocations[0]} (start = -2 for DISTANCE);
 * ops receives the EDLIB_EDOP_* alignment of the first location (PATH).
 * Limits of this wrapper: targetLength <= 319, queryLength <= 2048 (the shapes
 * of the short-read insertion path; edlib's Hirschberg regime is not reached);
 * NW + DISTANCE (bit-vector kernel): min(queryLength, targetLength) <= 6144.
 * (The batched entry points have no such limits: long strings go through dellyhip_refine_batch_lr /
 * dellyhip_msa_edlib / dellyhip_msa_wfa / dellyhip_edit_distance_nw_batch.) */
int dellyhip_edlib_align(dellyhip_ctx* ctx, const char* query, int32_t query_len, const char* target,
                         int32_t target_len, int32_t mode, int32_t task, int32_t out[4],
                         unsigned char* ops, int32_t ops_cap, int32_t* ops_len);

/* The whole of EdlibAlignResult (src/edlib.h:160-217) for one edlibAlign(query, target, edlibNewAlignConfig(k, mode, task,
 * equalities ? the 20 extended-IUPAC pairs of src/assemble.h:425 : NULL, ...)) call: *edit_distance (-1 when k >= 0 and the
 * distance exceeds k: then no locations and no alignment, src/edlib.h:166), ALL optimal end locations in ascending order
 * (end_locs[0 .. *num_locations)), the start location of each (task LOC / PATH; start_locs may be NULL for DISTANCE) and
 * the EDLIB_EDOP_* alignment of the first pair (task PATH), including edlib's Hirschberg regime (src/edlib.cpp:1188-1389).
 * Shapes: target <= 32766, query <= 32000 letters (DELLYHIP_E_LIMIT beyond).  loc_cap: target_len + 1 always suffices.
 * This is what include/delly_dropin/edlib.h builds the reference's C API on; any other additionalEqualities set has no
 * caller in the reference and is rejected there. */
int dellyhip_edlib_align_full(dellyhip_ctx* ctx, const char* query, int32_t query_len, const char* target, int32_t target_len,
                              int32_t k, int32_t mode, int32_t task, int32_t equalities, int32_t* edit_distance,
                              int32_t* num_locations, int32_t* end_locs, int32_t* start_locs, int32_t loc_cap,
                              unsigned char* ops, int32_t ops_cap, int32_t* ops_len);

/* int lcs(s1, s2)  src/msa.h:10-30 */
int dellyhip_lcs(dellyhip_ctx* ctx, const char* s1, int32_t m, const char* s2, int32_t n,
                 int32_t* out);

/* int gotoh(a1, a2, align, AlignConfig<true,true>, c.aliscore) src/gotoh.h:71-174
 * as called from palign, src/msa.h:106-107.  a1 is r1 x m, a2 is r2 x n
 * (row-major); align_out receives (r1+r2) x *len (row stride cap). */
int dellyhip_gotoh(dellyhip_ctx* ctx, const char* a1, int32_t r1, int32_t m, const char* a2,
                   int32_t r2, int32_t n, char* align_out, int32_t cap, int32_t* len,
                   int32_t* score);

/* int msa(c, sps, cs)  src/msa.h:185-239: returns rows in *rows, consensus in cs. */
int dellyhip_msa(dellyhip_ctx* ctx, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off,
                 char* cs, int32_t cs_cap, int32_t* cs_len, int32_t* rows);

/* int msaWfa(c, sps, cs, prefix, suffix)  src/assemble.h:547-726 (prefix / suffix may be empty). */
int dellyhip_msa_wfa(dellyhip_ctx* ctx, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off,
                     const char* prefix, int32_t prefix_len, const char* suffix, int32_t suffix_len,
                     char* cs, int32_t cs_cap, int32_t* cs_len, int32_t* rows);

/* int msaEdlib(c, sps, cs)  src/assemble.h:383-473: returns rows in *rows, consensus in cs. */
int dellyhip_msa_edlib(dellyhip_ctx* ctx, int32_t n_reads, const char* seq_blob, const uint64_t* seq_off,
                       char* cs, int32_t cs_cap, int32_t* cs_len, int32_t* rows);

#ifdef __cplusplus
}
#endif
#endif /* DELLYHIP_H */
