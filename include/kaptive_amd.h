/* kaptive_amd.h -- C ABI of libkaptive_amd.so: the MI355X (gfx950) implementation of Kaptive's typing hot path.
 *
 * The reference (klebgenomics/Kaptive) has no C ABI for this path: its native seam is the Python binding of the
 * third-party rammappy wheel plus numba-compiled kernels.  Each entry point below names the reference interface it
 * stands in for (paths relative to the reference checkout); INTEGRATION.md shows the ctypes binding a Kaptive
 * maintainer would add.  Conventions: every call returns 0 on success or a negative KP_E* code and records a message
 * retrievable with kp_last_error(); no exceptions cross the boundary; the caller owns every host buffer it passes and
 * may free it as soon as the call returns unless stated otherwise; the library owns all device memory.  A context is
 * bound to one GPU; calls on one context must be serialised by the caller, distinct contexts are independent (one
 * context per device / per host thread).
 *
 * Ownership of device memory.  Work buffers (anchors, band tasks, hit tables, reduction records) belong to the context,
 * not to a batch: a context owns KP_WORK_SLOTS work sets, kp_batch_align takes the next one round-robin, and the sizes
 * the library learns for them (overflow -> automatic rerun with more room) are kept for every later batch.  So a stream
 * of batches allocates nothing after its first few, and the context holds the alignment results of its KP_WORK_SLOTS
 * most recently aligned batches; calls that read results of a batch displaced since return KP_ESTATE.  The device
 * copies of batch inputs are recycled through the context in the same way.
 */
#ifndef KAPTIVE_AMD_H
#define KAPTIVE_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "kp_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KP_API __attribute__((visibility("default")))

#define KP_OK 0
#define KP_EINVAL (-1)    /* bad argument (null pointer, limits of kp_spec.h exceeded, inconsistent offsets) */
#define KP_EHIP (-2)      /* a HIP runtime call failed; message carries hipGetErrorString */
#define KP_ENOMEM (-3)    /* host or device allocation failed */
#define KP_ESTATE (-4)    /* call out of order (no database loaded, no batch aligned, ...) */
#define KP_EOVERFLOW (-5) /* an internal device buffer overflowed and the automatic retry also failed */
#define KP_ENOTSUP (-6)   /* the host lacks what the call needs (kp_fasta_ingest: no libbz2 / liblzma to load) */
#define KP_EIO (-7)       /* a file could not be opened or mapped (kp_fasta_ingest_file) */

typedef struct kp_ctx kp_ctx;     /* one GPU + stream + resident database */
typedef struct kp_batch kp_batch; /* a set of packed assemblies resident in HBM, plus its results */

/* ---- context -------------------------------------------------------------------------------------------------- */
/* GPUs the process sees (>= 0), or a negative KP_E* code: `kaptive assembly --devices all` shards over all of them, one
 * process each (the reference has no device notion; its unit of parallelism is the worker thread, cli.py:183-210). */
KP_API int kp_device_count(void);
/* NUMA node of a GPU (its PCI function's /sys/bus/pci/devices/<bus id>/numa_node), -1 when the host does not say: the
 * per-device processes pin themselves and their page-locked shards to it (kaptive_amd/affinity.py).  No context needed. */
KP_API int kp_device_numa_node(int device_id);
KP_API int kp_ctx_create(int device_id, kp_ctx **out);
KP_API void kp_ctx_destroy(kp_ctx *ctx);
/* Message of the last failed call on ctx (ctx may be NULL for a failed kp_ctx_create). Never NULL. */
KP_API const char *kp_last_error(const kp_ctx *ctx);
/* The context's own hipStream_t: database uploads and stand-alone protein alignments.  (Alignment passes run on the
 * stream of the work set they take, so that the passes of consecutive batches overlap; their stage times: kp_batch_profile.) */
KP_API void *kp_ctx_stream(kp_ctx *ctx);
/* Tuning knobs.  Defaults are read from the environment once, in kp_ctx_create (KAPTIVE_AMD_<NAME in upper case>);
 * names: anchor_cap, tasks_per_asm, hit_cap, trace_kb_per_asm, kept_cap, piece_cap, prot_cap (initial sizes of the work
 * buffers; trace_kb_per_asm: direction bits of the banded Smith-Waterman, KiB per assembly of a batch -- setting
 * one also forgets what the context has learnt for it), library_sort (anchors are sorted by the library's segmented radix sort instead of
 * the bucket sort of kp_bsort.hip; same result), scan_mode (ablation modes of the scan kernel, tools/scan_ablate.py). */
KP_API int kp_ctx_set_option(kp_ctx *ctx, const char *name, int64_t value);
#define KP_WORK_SLOTS 3

/* ---- pinned host memory ---------------------------------------------------------------------------------------------------
 * For callers that stream shards: kp_batch_create_async only overlaps with device work when `words` is page-locked. */
KP_API int kp_host_alloc(size_t bytes, void **out);
KP_API void kp_host_free(void *p);
/* The same in two steps, for a caller whose own threads fill the block: kp_host_reserve hands out plain anonymous
 * memory (2 MB-aligned, advised to use huge pages; no call into the device runtime, so it works -- and costs nothing --
 * before a context exists), kp_host_lock page-locks it once it has been written (2 ms per GB of touched huge pages;
 * untouched pages are faulted in by the lock itself, on the calling thread).  kp_host_free takes either state. */
KP_API int kp_host_reserve(size_t bytes, void **out);
KP_API int kp_host_lock(void *p);
/* Page-locked bytes this process currently holds through the library: kp_host_alloc blocks plus the table staging of
 * the contexts' input buffers.  (What a rank pins matters when eight of them share one host.) */
KP_API int64_t kp_host_pinned_bytes(void);
/* Device buffers the library has had to RE-allocate (free + allocate a larger one) since the process started.  Each of
 * those waits for the whole device, i.e. for every pass in flight: a stream of batches should settle at a constant
 * (work buffers are sized from what the context has learnt, with head-room) and a caller can check that it does. */
KP_API int64_t kp_device_allocations(void);

/* ---- database ---------------------------------------------------------------------------------------------------
 * Replaces what Serotyper.__init__ prepares for the aligner -- the list of (name, gene bytes) handed to
 * rammappy's map_batch (src/kaptive/serotyping/core.py:111-121,154) -- with a device-resident seed index over all
 * genes (both strands) and 4-bit copies of the gene sequences.
 *   gene_codes : one byte per base, 0..3 = ACGT, 4 = anything else; genes back to back
 *   gene_off   : n_genes + 1 offsets into gene_codes
 * Loading a database replaces any previous one. */
KP_API int kp_db_load(kp_ctx *ctx, const uint8_t *gene_codes, const int32_t *gene_off, int32_t n_genes);
/* Number of (k-mer, gene, strand, position) postings in the resident seed index. */
KP_API int64_t kp_db_n_postings(const kp_ctx *ctx);

/* ---- FASTA ingest (host only) --------------------------------------------------------------------------------------------
 * Replaces rammappy.fasta.parse_fasta_bytes + Sequences.from_records + the per-contig copies handed to Index.build
 * (src/kaptive/core/genome.py:35-46,188; src/kaptive/core/seq.py:281-325): one pass from (decompressed) FASTA text to
 * the packed layout of kp_spec.h.  The result is owned by the library until kp_fasta_free. */
typedef struct kp_packed_fasta {
    int64_t padded_len;   /* bases, multiple of KP_ASM_ALIGN */
    int32_t n_contigs, n_runs;
    uint32_t *words;      /* padded_len / 16 */
    int32_t *ctg_start, *ctg_len;
    int32_t *n_run_pairs; /* 2 * n_runs */
    char *names;          /* contig names back to back (first word of each header), not NUL-terminated */
    int32_t *name_off;    /* n_contigs + 1 */
    uint8_t *seqs;        /* KP_FASTA_KEEP_TEXT: the contigs' symbols as written, whitespace removed, back to back */
    int64_t n_seq_bytes;  /*   (contig c = seqs[sum of ctg_len[0..c) .. + ctg_len[c]]); NULL / 0 otherwise */
} kp_packed_fasta;
KP_API int kp_fasta_pack(const uint8_t *data, int64_t n, kp_packed_fasta **out);
/* The same from a file's bytes as they are on disk: KP_FASTA_GZIP inflates first (zlib; gzip or zlib framing, several
 * members), KP_FASTA_KEEP_TEXT also returns the sequence text, which is what GenomeAssembly.from_file needs next to the
 * packed form (src/kaptive/core/genome.py:194-214: open by suffix, read everything, parse). */
#define KP_FASTA_GZIP 1
#define KP_FASTA_KEEP_TEXT 2
#define KP_FASTA_BZ2 4 /* bzip2 stream(s): libbz2 of the host, looked up at first use; KP_ENOTSUP when the host has none */
#define KP_FASTA_XZ 8  /* .xz stream(s): liblzma of the host, likewise (the reference opens .gz / .bz2 / .xz by suffix) */
KP_API int kp_fasta_ingest(const uint8_t *data, int64_t n, int32_t flags, kp_packed_fasta **out);
/* The same from the file itself (GenomeAssembly.from_file, src/kaptive/core/genome.py:194-214, reads the whole file into a
 * bytes object first): the library reads it into a recycled buffer of its own and parses it there.
 * `flags` as above (the caller picks the compression from the suffix, as the reference does); KP_EIO when the path
 * cannot be opened or is not a regular file. */
KP_API int kp_fasta_ingest_file(const char *path, int32_t flags, kp_packed_fasta **out);
/* Many files in one call, on `threads` host threads of the library's own (0 = one per core, at most n_files): file i is
 * data[i][0 .. n[i]) with flags[i]; out[i] and rc[i] are what kp_fasta_ingest would have returned for it.  A reader that
 * feeds a GPU has to turn tens of GB/s of text into packed words; a Python thread per file spends too much of each call
 * holding the interpreter lock.  Returns KP_OK when the arguments were usable (look at rc[] for the files). */
KP_API int kp_fasta_ingest_many(const uint8_t *const *data, const int64_t *n, const int32_t *flags, int32_t n_files,
                                int32_t threads, kp_packed_fasta **out, int32_t *rc);
/* A chunk of files -> the tables of one batch, for a reader that feeds a GPU from FASTA files (the reference reads, parses
 * and indexes one file at a time on its worker threads, src/kaptive/serotyping/cli.py:183-210, core/genome.py:177-214).
 * kp_fasta_ingest_shard parses the files (kp_fasta_ingest_file, sequence text not kept) on `threads` threads of the library
 * (0 = one per core) and lays the tables out exactly as kp_batch_create* takes them; a file that failed is an empty
 * assembly here, rc[i] holds its code (n_failed / first_failed say whether to look).  kp_shard_words_into then copies the
 * packed words of all assemblies back to back into `dst` (total_words of them; page-locked memory the caller sized after
 * the first call) and releases the per-file buffers; the tables stay valid until kp_shard_free. */
typedef struct kp_packed_shard {
    int32_t n_asm, n_failed, first_failed;
    int64_t total_words;
    int64_t *asm_word_off;                /* n_asm + 1 */
    int32_t *ctg_start, *ctg_len;         /* asm_first_ctg[n_asm] each */
    int32_t *asm_first_ctg;               /* n_asm + 1 */
    int32_t *n_runs;                      /* 2 * asm_first_nrun[n_asm] */
    int32_t *asm_first_nrun;              /* n_asm + 1 */
    int32_t *rc;                          /* n_asm */
} kp_packed_shard;
KP_API int kp_fasta_ingest_shard(const char *const *paths, const int32_t *flags, int32_t n_files, int32_t threads,
                                 kp_packed_shard **out);
KP_API int kp_shard_words_into(kp_packed_shard *shard, uint32_t *dst, int64_t dst_words, int32_t threads);
KP_API void kp_shard_free(kp_packed_shard *shard);
/* Which text path the ingest runs on this host: 2 = 64 bytes at a time with AVX-512 (BW, VBMI2) + BMI2, 1 = AVX2 + BMI2,
 * 0 = table look-ups; settled from cpuid at first use (environment KAPTIVE_AMD_FASTA_SIMD caps it).  cap >= 0 lowers it
 * for the calls that follow (tests compare the paths), cap < 0 only asks.  All paths give the same bytes. */
KP_API int kp_fasta_simd(int32_t cap);
/* The same layout from contigs already in memory (Sequences.seqs / offsets / lengths of the reference's containers,
 * src/kaptive/core/seq.py:307-325): contig c is seqs[offsets[c] .. offsets[c] + lengths[c]).  names / name_off of the
 * result are empty. */
KP_API int kp_pack_contigs(const uint8_t *seqs, const int64_t *offsets, const int32_t *lengths, int32_t n_contigs,
                           kp_packed_fasta **out);
KP_API void kp_fasta_free(kp_packed_fasta *packed);

/* ---- batches of packed assemblies -------------------------------------------------------------------------------
 * Replaces GenomeAssembly.get_rammappy_index / rammappy.Index.build (src/kaptive/core/genome.py:177-191): instead of
 * a minimizer index per assembly, the 2-bit packed contigs themselves are made resident (layout: kp_spec.h).
 *   words          : packed bases of all assemblies back to back (assembly a = words[asm_word_off[a] .. asm_word_off[a+1]))
 *   asm_word_off   : n_asm + 1 offsets in uint32 words; each assembly's length is a multiple of KP_ASM_ALIGN bases
 *   ctg_start/len  : per contig, start (multiple of KP_CONTIG_ALIGN) and length in the assembly's own padded space
 *   asm_first_ctg  : n_asm + 1 offsets into ctg_start/ctg_len
 *   n_runs         : [start,end) pairs of N runs, sorted per assembly, in the assembly's padded space
 *   asm_first_nrun : n_asm + 1 offsets into n_runs (in runs, not ints)
 * kp_batch_create copies from host memory; kp_batch_create_device adopts `words` already in device memory (it must
 * stay valid until kp_batch_destroy) and copies only the small tables.  kp_batch_create_async enqueues the copies on
 * the context's copy stream and returns: `words` must stay valid and unchanged until kp_batch_upload_wait (or the
 * batch's first kp_batch_wait) has returned -- the tables may be freed at once; kp_batch_align of the batch waits for
 * the copies on the device, so the upload of batch i+1 overlaps the alignment pass of batch i. */
KP_API int kp_batch_create(kp_ctx *ctx, int32_t n_asm, const uint32_t *words, const int64_t *asm_word_off,
                    const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                    const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out);
KP_API int kp_batch_create_async(kp_ctx *ctx, int32_t n_asm, const uint32_t *words, const int64_t *asm_word_off,
                          const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                          const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out);
KP_API int kp_batch_upload_wait(kp_ctx *ctx, kp_batch *batch);
/* `batch` adopted (kp_batch_create_device) the device words of `other`, a batch of another context on the same GPU whose
 * upload may still be in flight: alignment passes of `batch` then wait for that upload on the device.  `other` must
 * outlive `batch`. */
KP_API int kp_batch_depends_on(kp_ctx *ctx, kp_batch *batch, kp_batch *other);
KP_API int kp_batch_create_device(kp_ctx *ctx, int32_t n_asm, const uint32_t *d_words, const int64_t *asm_word_off,
                           const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                           const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out);
/* Device address of the batch's packed words (valid until kp_batch_destroy): lets a second context on the same GPU
 * type the same resident assemblies against another database via kp_batch_create_device instead of a second copy. */
KP_API const void *kp_batch_device_words(const kp_batch *batch);
KP_API void kp_batch_destroy(kp_batch *batch);

/* ---- alignment --------------------------------------------------------------------------------------------------
 * Replaces Aligner(...).map_batch(gene_seqs) with best_n=50000, pri_ratio=0.0 (src/kaptive/serotyping/core.py:
 * 148-154): every database gene against every contig of every assembly of the batch, all hits reported.
 * kp_batch_align enqueues the kernels (seed scan -> anchor sort -> band tasks -> banded Smith-Waterman) and returns;
 * kp_batch_wait blocks until they finish, then finalises the hit table (emission order of kp_spec.h).  */
KP_API int kp_batch_align(kp_ctx *ctx, kp_batch *batch);
KP_API int kp_batch_wait(kp_ctx *ctx, kp_batch *batch);
/* hit_off[n_asm+1]: hits of assembly a are rows hit_off[a] .. hit_off[a+1] of the table returned by kp_batch_hits.
 * These are the columns the reference reads off rammappy hit objects (src/kaptive/core/alignment.py:415-446). */
KP_API int kp_batch_hit_offsets(kp_ctx *ctx, kp_batch *batch, int64_t *hit_off);
KP_API int kp_batch_hits(kp_ctx *ctx, kp_batch *batch, kp_hit *out, int64_t cap);
/* Replaces the batch's finished hit table (valid after kp_batch_wait) with the caller's: rows hit_off[a] .. hit_off[a+1] of
 * `hits` become assembly a's hits, taken as they are -- emission order, mapq and all.  What follows (kp_batch_score,
 * kp_batch_reduce, kp_batch_typing) then reduces THAT table.  The reference's reduction takes any rammappy hit table
 * (src/kaptive/core/alignment.py:392-474, src/kaptive/serotyping/core.py:157-396); this is how hit tables no aligner would
 * produce (equal scores, mapq 0 / 255 / 1 ties: src/kaptive/core/alignment.py:669-675) reach the device reduction in the
 * parity tests. */
KP_API int kp_batch_set_hits(kp_ctx *ctx, kp_batch *batch, const kp_hit *hits, const int64_t *hit_off);
/* counters of the last kp_batch_align: [0] anchors, [1] band tasks, [2] DP cells, [3] hits, [4] overflow retries */
KP_API int kp_batch_stats(kp_ctx *ctx, kp_batch *batch, int64_t *stats5);

/* Stage durations of the most recent alignment pass of this batch (valid after kp_batch_wait), from HIP events the
 * library records on the context's stream around every pass: ms7 = seed scan (kp_scan_kernel), candidate expansion +
 * anchor compaction + sort, chaining + task ordering, banded SW fill (all band widths run in one launch: ms7[3]), its
 * traceback (ms7[4]); ms7[5..6] are 0 and kept for layout stability.  bytes_scanned receives the algorithmic bytes the scan kernel
 * streams (4 * total words). */
KP_API int kp_batch_profile(kp_ctx *ctx, kp_batch *batch, float *ms7, int64_t *bytes_scanned);

/* Stage outputs for stage-by-stage parity tests (valid after kp_batch_wait): sorted anchor keys of one assembly, and
 * the band tasks of one assembly as 8 x int32 rows (gs, contig, lo, width, n_anchors, qmin, qmax, chain_score) in device
 * order. */
KP_API int64_t kp_batch_anchors(kp_ctx *ctx, kp_batch *batch, int32_t asm_index, uint64_t *out, int64_t cap);
KP_API int64_t kp_batch_tasks(kp_ctx *ctx, kp_batch *batch, int32_t asm_index, int32_t *out8, int64_t cap);
/* ... and what the banded Smith-Waterman made of each of those tasks, row for row in the order of kp_batch_tasks: 7 x
 * int32 (score, q_start, q_end, t_start, t_end, matches, block_len).  Tasks whose best cell stays below the score
 * cut-off (KP_MIN_DP_SCORE) are not traced back: their row holds the score and zeros. */
KP_API int64_t kp_batch_task_results(kp_ctx *ctx, kp_batch *batch, int32_t asm_index, int32_t *out7, int64_t cap);
/* ... and the JOINS of one assembly (kp_spec.h, kp-align v5: chains of a group's anchors across diagonal jumps of up to KP_JOIN_BW, the
 * stand-in for minimap2's bw = 500 chaining inside rammappy's map_batch, src/kaptive/serotyping/core.py:147-155), in device
 * order, KP_JOIN_ROW_INTS x int32 each: gs, contig, n_pieces, n_anchors, chain_score, width, lo[KP_JOIN_MAX_PIECES], then per
 * piece 11 values -- state (0 nothing, 1 joined hit, 2 rejected by the drop test), mask of the pieces its path visits, cell
 * score, q_start, q_end, t_start, t_end (query as aligned, target in assembly coordinates), matches, block_len, reported
 * score, order-score bonus. */
#define KP_JOIN_ROW_INTS (6 + KP_JOIN_MAX_PIECES + 11 * KP_JOIN_MAX_PIECES)
KP_API int64_t kp_batch_joins(kp_ctx *ctx, kp_batch *batch, int32_t asm_index, int32_t *out, int64_t cap);

/* ---- batched typing: hit table -> per-assembly records -----------------------------------------------------------------
 * Replaces, for a whole batch, the reduction Serotyper.__call__ runs per genome after the aligner returns
 * (src/kaptive/serotyping/core.py:157-396): per-gene best coverage and locus scores, overlap cull, clustering, locus
 * pieces, inside/expected flags, missing genes, gene extraction + translation, protein identity, gene states.  Three
 * float steps stay with the caller in numpy because only numpy reproduces their bits (kaptive_amd/serotyping/batch.py):
 * the completeness**3 penalty and argmax between kp_batch_score and kp_batch_reduce, the argsort of piece positions
 * and the float32 mean identity afterwards.
 *
 * kp_db_load_typing: the Database columns the reduction reads (src/kaptive/db/core.py:82-98), host pointers; call it
 *   after kp_db_load (the gene count and gene lengths come from there). */
typedef struct kp_typing_tables {
    const uint16_t *gene_locus;    /* Database.gene_locus_indices */
    const uint8_t *gene_extra;     /* Database.extra_genes */
    const uint16_t *gene_pos;      /* Database.gene_positions */
    const int8_t *gene_strand;     /* Database.gene_intervals.strands */
    const int32_t *locus_gene_off; /* Database.locus_gene_offsets */
    const int32_t *locus_gene_len; /* Database.locus_gene_lengths */
    int32_t n_loci;
    const uint8_t *prot;           /* Database.translations.seqs (stop codons kept) */
    const int32_t *prot_off, *prot_len;
} kp_typing_tables;
KP_API int kp_db_load_typing(kp_ctx *ctx, const kp_typing_tables *tables);
/* Several databases in one context (typing groups).  The reference types an assembly against the K database and then
 * against the O database, aligning each database's genes in its own map_batch call (src/kaptive/serotyping/cli.py:183-210
 * runs one Serotyper per database).  Here the genes of all databases can be loaded into one context back to back
 * (kp_db_load over the concatenation) so that an assembly's bases are scanned, sorted, chained and aligned once; the
 * hit table is sorted by gene, so every database's reduction gets its own contiguous run of it.
 *   kp_db_load_typing_group: typing tables of the database whose genes are [gene_lo, gene_hi) of the context's genes
 *     (tables index genes relative to gene_lo); group is a small caller-chosen index < KP_MAX_TYPING_GROUPS.
 *     kp_db_load_typing(ctx, t) is kp_db_load_typing_group(ctx, 0, 0, n_genes, t).
 *   kp_batch_use_group: the group that kp_batch_score / kp_batch_reduce / kp_batch_typing_caps / kp_batch_typing /
 *     kp_batch_proteins address from now on (0 after kp_batch_create); every group keeps its own results. */
#define KP_MAX_TYPING_GROUPS 16
KP_API int kp_db_load_typing_group(kp_ctx *ctx, int32_t group, int32_t gene_lo, int32_t gene_hi,
                                   const kp_typing_tables *tables);
KP_API int kp_batch_use_group(kp_ctx *ctx, kp_batch *batch, int32_t group);
/* After kp_batch_align: finalises the hit tables on the device and returns locus_scores (float64, sum of best query
 * coverages per locus, core.py:188-193) and locus_counts (genes counted, core.py:196-198), both [n_asm][n_loci]. */
KP_API int kp_batch_score(kp_ctx *ctx, kp_batch *batch, double min_gene_coverage, double *locus_scores,
                          int32_t *locus_counts);
/* Enqueues the rest of the reduction for the caller's choice of best locus per assembly (core.py:206). */
KP_API int kp_batch_reduce(kp_ctx *ctx, kp_batch *batch, const int32_t *best_locus, const kp_typing_params *params);
/* kp_batch_typing_caps waits for the reduction and reports the largest number of kept hits / pieces any assembly of the
 * batch has (at least 1): the smallest strides kp_batch_typing accepts.  kp_batch_typing copies out one summary per
 * assembly and its kept hits / pieces at kept[a * kept_stride], pieces[a * piece_stride]. */
KP_API int kp_batch_typing_caps(kp_ctx *ctx, kp_batch *batch, int32_t *kept_cap, int32_t *piece_cap);
KP_API int kp_batch_typing(kp_ctx *ctx, kp_batch *batch, kp_asm_summary *summaries, kp_kept *kept, int32_t kept_stride,
                           kp_piece *pieces, int32_t piece_stride);
/* Translated proteins of one assembly (kp_kept.prot_off/prot_len index into it); returns bytes copied. */
KP_API int kp_batch_proteins(kp_ctx *ctx, kp_batch *batch, int32_t asm_index, uint8_t *out, int64_t cap);

/* ---- report rows (host only) -----------------------------------------------------------------------------------------------
 * Replaces KaptiveRow.from_result + bytes(row) per genome (src/kaptive/serotyping/io.py:191-296, 37-43): the TSV lines
 * of a whole batch from the records kp_batch_typing returned and the per-assembly decisions the caller finished
 * (phenotype rules, typeability, problems: src/kaptive/serotyping/core.py:398-459, models.py:538-558).  Strings are
 * byte blobs with n + 1 offsets.  Returns the number of bytes the rows need (written to `out` while they fit `cap`), or
 * a negative error code.  No GPU involved. */
typedef struct kp_row_tables {   /* per database */
    const char *prefix;          /* "<Kaptive version>\t<database name>\t<database version>\t" */
    int32_t prefix_len;
    const char *gene_ids;        /* Database.genes.ids */
    const int32_t *gene_id_off;
    const char *locus_names;     /* Database.loci.ids */
    const int32_t *locus_name_off;
    const int32_t *locus_gene_off, *locus_gene_len;
} kp_row_tables;
typedef struct kp_row_columns {  /* per assembly of the batch */
    const char *asm_ids;
    const int32_t *asm_id_off;
    const char *phenotypes;      /* Best match type */
    const int32_t *phenotype_off;
    const int32_t *best_locus;
    const uint8_t *typeable;
    const int32_t *problems;     /* bit 0..4 = ? + - * ! */
    const double *identity, *coverage, *length_discrepancy; /* NaN discrepancy -> n/a */
} kp_row_columns;
KP_API int64_t kp_format_rows(const kp_row_tables *tables, int32_t n_asm, const kp_asm_summary *summaries,
                              const kp_kept *kept, int32_t kept_stride, const kp_row_columns *columns, char *out,
                              int64_t cap);

/* ---- JSON lines of a whole batch (host only) ----------------------------------------------------------------------------------
 * Replaces orjson.dumps(SerotypingResult.to_dict(), OPT_SERIALIZE_NUMPY | OPT_APPEND_NEWLINE) per genome
 * (src/kaptive/serotyping/cli.py:67-76, src/kaptive/serotyping/models.py:629-654) together with the construction of the
 * result object it serialises (GeneHits columns: src/kaptive/serotyping/core.py:303-329; locus / gene / protein sequences:
 * src/kaptive/core/seq.py:327-408): one line per assembly from the records of kp_batch_typing, the columns the host
 * finished and the contigs' text.  Strings that come from the database or the batch arrive as ready JSON literals (quotes
 * and escapes included; blobs with n + 1 offsets); contig names of the locus pieces arrive escaped but unquoted.  Returns the
 * number of bytes the lines need (written to `out` while they fit `cap`), or a negative error code. */
typedef struct kp_json_tables {  /* per database */
    const char *head;            /* {"kaptive_version":"..","database_name":"..",...,"database_taxon":N,"genome": */
    int32_t head_len;
    const char *gene_names;      /* Database.genes.ids (missing genes, ids of gene and protein sequences) */
    const int64_t *gene_name_off;
    const char *gene_ids, *cluster_names, *products;  /* GeneHits text columns: S32 / S10 / S64 truncations, per gene */
    const int64_t *gene_id_off, *cluster_name_off, *product_off;
    const char *locus_names;
    const int64_t *locus_name_off;
    const int32_t *locus_gene_off, *locus_gene_len;
    const int32_t *gene_position; /* Database.gene_positions */
    const int8_t *gene_strand;    /* Database.gene_intervals.strands */
    const uint8_t *comp_map;      /* [256] complement of a sequence byte (core/seq.py) */
    const uint8_t *char_map;      /* [256] byte -> 0..4 */
    const uint8_t *codon_map;     /* [125] codon -> amino acid, 42 = stop */
    int32_t n_loci, n_genes;      /* sizes of the tables above: every locus / gene index of the records is checked against them */
} kp_json_tables;
typedef struct kp_json_columns { /* per assembly of the batch */
    const char *asm_ids;
    const int64_t *asm_id_off;
    const char *phenotypes;
    const int64_t *phenotype_off;
    const int32_t *best_locus;
    const uint8_t *typeable;
    const int32_t *problems;
    const double *best_score, *completeness, *identity, *coverage, *length_discrepancy;
    const int32_t *piece_order;  /* [n_asm * piece_stride] numpy's argsort of the pieces' mean positions */
    const char *piece_ctg_names; /* escaped name of the contig of piece [a * piece_stride + p] */
    const int64_t *piece_ctg_name_off;
    const uint8_t *const *ctg_seqs;  /* per assembly: the contigs' text back to back ... */
    const int32_t *const *ctg_off;   /* ... and where each contig starts in it */
    const int32_t *n_ctg;            /* per assembly: how many contigs ctg_off[a] lists ... */
    const int64_t *ctg_text_len;     /* ... and how many bytes ctg_seqs[a] holds: records that point outside are KP_EINVAL */
} kp_json_columns;
KP_API int64_t kp_format_json(const kp_json_tables *tables, int32_t n_asm, const kp_asm_summary *summaries, const kp_kept *kept,
                              int32_t kept_stride, const kp_piece *pieces, int32_t piece_stride, const kp_json_columns *columns,
                              char *out, int64_t cap);
/* The per-assembly FASTA outputs (-l / -g / -p; src/kaptive/serotyping/cli.py:78-114: locus_seqs / gene_seqs /
 * translations .to_fasta() per result) from the same tables: kind 0 = locus pieces, 1 = gene sequences, 2 = translations;
 * `names` are RAW bytes -- for kind 0 the contig name of piece slot [a * piece_stride + p], for 1 and 2 the gene names by
 * gene index --; the records of all assemblies back to back, asm_end[a] = end of assembly a's.  Returns the size needed. */
KP_API int64_t kp_format_fasta(const kp_json_tables *tables, int32_t n_asm, const kp_asm_summary *summaries, const kp_kept *kept,
                               int32_t kept_stride, const kp_piece *pieces, int32_t piece_stride, const kp_json_columns *columns,
                               int32_t kind, const char *names, const int64_t *name_off, char *out, int64_t cap, int64_t *asm_end);

/* ---- protein alignment ------------------------------------------------------------------------------------------
 * Replaces PairwiseAligner.__call__ / _batched_banded_gotoh (src/kaptive/core/pairwise.py:255-325, 395-584) in its
 * unseeded mode with the defaults gap_open 11, gap_extend 1, k 20.  Sequences are raw bytes (amino-acid letters).
 *   out8 : n rows of score, matches, mismatches, gaps, q_start, q_end, t_start, t_end  */
KP_API int kp_protein_align(kp_ctx *ctx, const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                     const int32_t *t_off, const int32_t *t_len, int32_t n, int32_t *out8);
/* The seeded mode of the same kernel (pairwise.py:449-451, PairwiseAligner.align_seeds; caller compare.LocusComparator,
 * src/kaptive/compare.py:360-366): the band is k diagonals either side of the seed diagonal of every pair,
 * |j - (i - diagonal_offsets[p])| <= k, and no longer widens with the length difference. */
KP_API int kp_protein_align_seeded(kp_ctx *ctx, const uint8_t *q, const int32_t *q_off, const int32_t *q_len,
                                   const uint8_t *t, const int32_t *t_off, const int32_t *t_len, int32_t n,
                                   const int32_t *diagonal_offsets, int32_t k, int32_t *out8);

/* ---- locus comparison seeds (host only) -------------------------------------------------------------------------------------
 * Replace the numba kernels of kaptive.core.kmers.RandstrobeIndex that compare.LocusComparator uses
 * (src/kaptive/core/kmers.py:997-1155, 779-819, 1158-1282; src/kaptive/compare.py:343-358).
 * kp_randstrobes: records {hash u64, seq_idx u32, pos1 u32, pos2 u32} (20 bytes, packed) of every sequence, in sequence
 *   and position order, or stably sorted by hash; returns how many there are (written when they fit `cap`).
 * kp_randstrobe_top_hits: per query sequence the target sequence sharing most record hashes (first maximum), that count
 *   and the diagonal offset pos1(query) - pos1(target) of the first shared record met; queries without records get 0. */
KP_API int64_t kp_randstrobes(const uint8_t *seqs, const int32_t *offsets, const int32_t *lengths, int32_t n_seqs,
                              const uint8_t *lut256, int32_t k, int32_t s, int32_t w_min, int32_t w_max,
                              int32_t sort_by_hash, void *out_records, int64_t cap);
KP_API int kp_randstrobe_top_hits(const void *query_records, int64_t n_query_records, int32_t n_queries,
                                  const void *target_records_sorted, int64_t n_target_records, int32_t n_targets,
                                  uint32_t *best_target, uint32_t *best_score, int32_t *diagonal_offset);

#ifdef __cplusplus
}
#endif
#endif /* KAPTIVE_AMD_H */
