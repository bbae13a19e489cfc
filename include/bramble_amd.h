/* bramble_amd -- MI355X-native genome -> transcriptome projection engine.
 *
 * C ABI of libbramble_amd.so: the drop-in boundary for bramble's projection hot
 * path.  Every entry point names the reference interface it replaces
 * (paths relative to the bramble tree).  No torch / HIP types appear here:
 * plain pointers, sizes and opaque handles only.  Device pointers are passed as
 * `void *` / typed pointers that the caller obtained from its own allocator
 * (hipMalloc, a torch tensor's data_ptr, ...).
 *
 * Conventions (same as the reference):
 *   - exon intervals are 1-based half-open [start, end) (GTF end + 1;
 *     src/bramble.cpp:164-165, bramble-rs/src/annotation.rs:52-64);
 *   - ref_start / mate_start are 1-based SAM POS; projected `pos` is the 0-based
 *     BAM core.pos the reference writes (src/core.cpp:153-158);
 *   - CIGAR words are BAM-packed (len << 4 | op), ops 0..9 as in the SAM spec;
 *   - input batches are name-collated: all alignments of one query name are
 *     contiguous (README.md:39-52).
 *
 * All functions return BR_OK (0) or a negative BR_ERR_* code; none aborts.
 * The library never falls back to a CPU path: without a usable HIP device the
 * device entry points return BR_ERR_NO_DEVICE.
 */
#ifndef BRAMBLE_AMD_H
#define BRAMBLE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BR_OK 0
#define BR_ERR_INVALID_ARG (-1)
#define BR_ERR_NO_DEVICE (-2)
#define BR_ERR_HIP (-3)
#define BR_ERR_ANNOTATION (-4) /* unknown seqname, overlapping exons inside a transcript, ... */
#define BR_ERR_CAPACITY (-5)   /* batch exceeds 32-bit device offsets; split it */
#define BR_ERR_UNSUPPORTED (-6)

/* ---- annotation ------------------------------------------------------------ */

/* bramble-rs/src/annotation.rs:13-17 (Exon) */
typedef struct br_exon { uint32_t start, end; } br_exon;

/* bramble-rs/src/annotation.rs:19-25 (Transcript); strand '+' or '-' */
typedef struct br_transcript {
  const char *id;
  const char *seqname;
  char strand;
  const br_exon *exons;
  uint32_t n_exons;
} br_transcript;

/* bramble-rs/src/fasta.rs:21-108 (FastaDb::from_seqs): one named sequence */
typedef struct br_fasta_seq { const char *name; const char *seq; uint64_t len; } br_fasta_seq;

typedef struct br_index br_index; /* immutable after build; shareable across threads (g2tTree) */
typedef struct br_ctx br_ctx;     /* per host thread / per stream scratch (ProjectionContext) */

/* Replaces build_g2t_tree (src/bramble.cpp:132-211) / build_g2t_from_refnames
 * (bramble-rs/src/g2t.rs:515-532).  tid = position in `transcripts`.  `device`
 * is the HIP device ordinal the flattened exon tables are uploaded to; pass -1
 * for a host-only index (accessors work, projection does not).  `fasta` may be
 * NULL (no -S). */
int br_index_build(const br_transcript *transcripts, size_t n_transcripts, const char *const *refnames,
                   size_t n_refnames, const br_fasta_seq *fasta, size_t n_fasta, int device,
                   br_index **out);
/* Same index from flat arrays (the form a GTF loader or a generator holds):
 * transcript t lies on reference tx_ref_id[t] (0-based, < n_refs) with strand
 * tx_strand[t] and exons [tx_exon_off[t], tx_exon_off[t+1]) of (ex_start, ex_end).
 * tx_names may be NULL ("tx<t>"); fasta_by_ref is NULL or n_refs entries
 * (seq == NULL: no sequence for that reference). */
int br_index_build_flat(size_t n_tx, const int32_t *tx_ref_id, const int8_t *tx_strand,
                        const uint64_t *tx_exon_off, const uint32_t *ex_start, const uint32_t *ex_end,
                        const char *const *tx_names, size_t n_refs, const br_fasta_seq *fasta_by_ref,
                        int device, br_index **out);
void br_index_free(br_index *);
/* bramble-rs/src/g2t.rs:320-342 accessors */
size_t br_index_num_transcripts(const br_index *);
const char *br_index_transcript_name(const br_index *, uint32_t tid); /* NULL if out of range */
int64_t br_index_transcript_len(const br_index *, uint32_t tid);      /* -1 if out of range */
size_t br_index_num_refs(const br_index *);
size_t br_index_num_intervals(const br_index *);                      /* transcript-exon rows */
size_t br_index_device_bytes(const br_index *);

/* ---- configuration --------------------------------------------------------- */

/* ProjectionConfig (bramble-rs/src/api.rs:184-206) widened with the C++ CLI
 * switches that select the evaluator presets (src/bramble.cpp:457-485,
 * src/evaluate.cpp:1136-1221).  has_* = the matching --max-... / --similarity-
 * threshold override was given. */
typedef struct br_config {
  int32_t lr, lr_hq, strict; /* --lr, --lr-hq, --strict */
  int32_t use_fasta;         /* -S given (index must have been built with sequences) */
  int32_t fr, rf;            /* --fr, --rf */
  int32_t has_max_clip, has_max_junc_ins, has_max_junc_gap, has_sim_thr, has_max_error_exon;
  uint32_t max_clip, max_junc_ins, max_junc_gap, max_error_exon;
  float sim_thr;
  double junc_miss_discount; /* Rust-only knob (api.rs:197-205); must be 1.0 (C++ has none) */
} br_config;

void br_config_short_read(br_config *); /* ProjectionConfig::short_read, api.rs:210-216 */
void br_config_long_read(br_config *);  /* ProjectionConfig::long_read (= --lr), api.rs:219-225 */

/* Resolved evaluator thresholds (ReadEvaluationConfig, include/evaluate.h:275-285) */
typedef struct br_thresholds {
  uint32_t max_clip, max_junc_ins, max_junc_gap, max_error_exon;
  int32_t ignore_small_exons, filter_by_similarity;
  float similarity_threshold;
} br_thresholds;
int br_config_resolve(const br_config *, br_thresholds *out);

/* ---- struct-of-arrays batches (host memory) --------------------------------- */

/* Flat form of GenomicAlignment[] (bramble-rs/src/api.rs:73-126). */
typedef struct br_batch {
  int64_t n_aln;
  const int32_t *ref_id;     /* 0-based; < 0: alignment skipped (api.rs:316-318) */
  const int32_t *ref_start;  /* 1-based */
  const uint16_t *flags;     /* SAM flag bits: 0x1 paired, 0x10 reverse, 0x40 read1, 0x80 read2 */
  const int8_t *xs;          /* first char of XS tag or 0 */
  const int8_t *ts;          /* first char of ts tag or 0 */
  const uint64_t *cigar_off; /* n_aln + 1 */
  const uint32_t *cigar;
  const int32_t *mate_ref_id; /* -1 if none */
  const int32_t *mate_start;  /* 1-based, 0 if none */
  const uint64_t *name_off;   /* n_aln + 1 */
  const char *names;
  const uint64_t *seq_off;    /* n_aln + 1 or NULL */
  const char *seqs;           /* ASCII bases */
  const int32_t *l_qseq;
} br_batch;

/* Input contract of convert_reads computed on the host: read-name groups
 * (src/core.cpp:347-380) and the mate index of process_pairs
 * (src/bramble.cpp:272-311; at most one mate per alignment).  group_off must
 * hold n_aln + 1 entries; *n_groups receives the group count. */
int br_batch_prepare(const br_batch *, int32_t *mate_idx, uint32_t *group_off, int64_t *n_groups);

/* With -S the clip rescue uses ONE sequence per read-name group: the first
 * record of the group that has one (src/core.cpp:353-378).  seq_src[i] = index of
 * that record for alignment i, or -1. */
int br_batch_seq_source(const br_batch *, const uint32_t *group_off, int64_t n_groups, int32_t *seq_src);

/* One emitted BAM record (ProjectedAlignment, bramble-rs/src/api.rs:135-176, plus
 * the fields the C++ writer sets: rewritten CIGAR, MAPQ, mate fields;
 * src/core.cpp:96-212, src/bam.cpp:531-588).  Struct-of-arrays; arrays are owned
 * by the context and stay valid until the next projection call on it. */
typedef struct br_rows {
  int64_t n_rows;
  const int32_t *input_index;
  const uint32_t *transcript_id;
  const uint32_t *pos;            /* 0-based transcript position (fwpos / rcpos by strand) */
  const int8_t *strand;           /* transcript strand '+' / '-' */
  const uint64_t *cigar_off;      /* n_rows + 1, into cigar */
  const uint32_t *cigar;          /* rewritten CIGAR (update_cigar, src/bam.cpp:502-528) */
  const double *similarity_score;
  const int32_t *clip_score;
  const int32_t *junc_hits;
  const int32_t *aligned_len;     /* ref_consumed: transcript bases spanned */
  const uint32_t *nh, *hi, *mapq;
  const uint8_t *is_primary;      /* best similarity score per read name, libstdc++-exact tie-break (src/core.cpp:243-307) */
  const uint8_t *is_paired;       /* emitted together with its mate */
  const uint8_t *same_transcript_as_mate;
  const uint8_t *is_first;        /* read1 side of the emitted pair */
  const int32_t *mate_transcript_id, *mate_pos, *insert_size;
  const uint32_t *group;          /* read-name group index */
  /* counters of src/bramble.cpp:729-736 */
  uint64_t total_complete, total_unique, dropped_reads, total_processed;
} br_rows;

/* ---- contexts and projection ------------------------------------------------ */

int br_ctx_new(const br_index *, br_ctx **out);
void br_ctx_free(br_ctx *);

/* Replaces convert_reads minus BAM writing (include/evaluate.h:377-381; sole
 * caller process_bundle, src/threads.cpp:100-104): host batch in, rows out.
 * Uploads, runs the HIP pipeline on the context's stream, downloads, and
 * finalises primary flags on the host. */
int br_project_batch(br_ctx *, const br_config *, const br_batch *, br_rows *out);

/* Device-resident form: every pointer is a device pointer on the index's
 * device; offsets are 32-bit.  mate_idx / group_off come from
 * br_batch_prepare.  `stream` is a hipStream_t (NULL = default stream). */
typedef struct br_device_batch {
  int64_t n_aln, n_groups;
  const int32_t *ref_id;
  const int32_t *ref_start;
  const uint16_t *flags;
  const int8_t *xs, *ts;
  const uint32_t *cigar_off; /* n_aln + 1 */
  const uint32_t *cigar;
  const int32_t *mate_idx;   /* n_aln */
  const uint32_t *group_off; /* n_groups + 1 */
  const int32_t *l_qseq;
  const uint32_t *seq_off;   /* n_aln + 1 or NULL */
  const uint8_t *seqs;
  int64_t n_cigar_words;     /* total words in cigar */
  int32_t max_n_cigar;       /* longest single CIGAR in the batch */
  /* -S clip rescue only (use_fasta with --lr / --lr-hq): */
  const int32_t *seq_src;    /* n_aln, from br_batch_seq_source; NULL without sequences */
  int32_t max_soft_clip;     /* longest leading / trailing S op in the batch */
  /* optional read names (primary / secondary choice needs them; NULL: is_primary stays 0) */
  const uint32_t *name_off;  /* n_aln + 1 */
  const uint8_t *names;
} br_device_batch;

/* ABI version 2: the rows of the device-resident entry points are PACKED (24 bytes + 16 bytes of detail per emitted
 * record instead of 22 separate arrays): the row stage of the pipeline is bound by the bytes it writes, and a host
 * that downloads rows pays for them a second time over PCIe.  The wide one-array-per-field view (ABI version 1's
 * br_device_rows, now br_device_wide_rows) is derived from the packed table on request.
 * ABI version 3: the projection writes the 24 bytes only; the 16 bytes of detail (br_row_x) are derived on request as
 * well (br_device_rows.x is NULL: br_device_rows_detail(); host rows carry it with "host_detail" = 1 as before). */
#define BR_ABI_VERSION 3

/* One emitted BAM record (ProjectedAlignment, bramble-rs/src/api.rs:135-176; the fields write_to_bam sets,
 * src/core.cpp:96-212).  Rows of one read name are contiguous and a pair's two records are adjacent (the leader's own
 * record first), so
 *   HI   = 1-based rank of the row among its read name's rows (also stored in br_row_x),
 *   MAPQ = br_row_mapq(nh, long_reads)                                    (get_mapq, src/core.cpp:46-58),
 *   mate transcript / position = the adjacent row's, insert size from the two positions and the read length
 *                                                                         (set_mate_info, src/bam.cpp:531-588). */
typedef struct br_row_a {
  uint32_t transcript_id;
  uint32_t pos;   /* 0-based transcript position (fwpos / rcpos by strand) */
  uint32_t meta;  /* BR_ROW_* bits */
  uint32_t nh;    /* records emitted for the read name (NH tag) */
} br_row_a;
#define BR_ROW_NCIGAR(meta) ((meta) & 0xffffffu) /* ops of the rewritten CIGAR */
#define BR_ROW_MINUS (1u << 24)                  /* transcript strand '-' */
#define BR_ROW_PAIRED (1u << 25)                 /* emitted together with its mate (the adjacent row) */
#define BR_ROW_SAME_TX (1u << 26)                /* ... on the same transcript */
#define BR_ROW_FIRST (1u << 27)                  /* the leading record of the pair / an unpaired record; its input
                                                  * alignment is the leader of row_off[], the other row's is the mate */
#define BR_ROW_PRIMARY (1u << 28)                /* primary record of the read name (src/core.cpp:243-307) */
typedef struct br_row_x {
  uint32_t input_index; /* alignment of the batch this record rewrites */
  uint32_t junc_hits;
  uint32_t aligned_len; /* ref_consumed: transcript bases spanned */
  uint32_t hi;          /* HI tag */
} br_row_x;
uint32_t br_row_mapq(uint32_t nh, int long_reads); /* get_mapq, src/core.cpp:46-58 */

/* Packed rows as device pointers; valid until the next projection call on the context.
 * cigar[r]: the rewritten CIGAR (update_cigar, src/bam.cpp:502-528) itself when it has <= 2 ops (op 0 in the low
 * word, op 1 in the high word), else the offset of its ops in pool[] (on the device: the sparse CIGAR arena the emit
 * pass wrote; the host entry points compact it).  row_off[i] .. row_off[i + 1] are the rows
 * emitted by alignment i as the leader (for itself and, alternating, its mate). */
typedef struct br_device_rows {
  int64_t n_rows, n_matches, n_pool_words;
  const br_row_a *a;
  const uint64_t *cigar;
  const br_row_x *x;              /* NULL: the detail column is derived on request, br_device_rows_detail() */
  const double *similarity_score; /* NULL unless the preset filters by similarity (then every score is 0.0) */
  const int32_t *clip_score;      /* NULL likewise (0) */
  const uint32_t *pool;
  const uint64_t *row_off;        /* n_aln + 1 */
  uint64_t total_complete, total_unique, dropped_reads, total_processed;
} br_device_rows;

/* br_row_x of the rows of the context's last projection call (a device array of n_rows entries, valid until the next
 * projection call): derived on the first request from what the call left in HBM -- the projection itself writes 24 bytes
 * per record (br_row_a + the CIGAR reference). */
int br_device_rows_detail(br_ctx *, void *stream, const br_row_x **x);

/* The wide view: same field meaning as br_rows; is_primary is filled when the batch carries read names. */
typedef struct br_device_wide_rows {
  int64_t n_rows, n_cigar_words;
  const int32_t *input_index;
  const uint32_t *transcript_id, *pos;
  const int8_t *strand;
  const uint64_t *cigar_off;
  const uint32_t *cigar;
  const double *similarity_score;
  const int32_t *clip_score, *junc_hits, *aligned_len;
  const uint32_t *nh, *hi, *mapq;
  const uint8_t *is_paired, *same_transcript_as_mate, *is_first;
  const int32_t *mate_transcript_id, *mate_pos, *insert_size;
  const uint32_t *group;
  const uint8_t *is_primary;
} br_device_wide_rows;

int br_project_batch_device(br_ctx *, const br_config *, const br_device_batch *, void *stream,
                            br_device_rows *out);
/* Derives the wide view of the LAST projection call's rows on the device (arrays owned by the context, valid until
 * the next projection call). */
int br_device_rows_expand(br_ctx *, void *stream, br_device_wide_rows *out);

/* ---- flat batches over PCIe: packed rows home, uploads / downloads overlapped with the projection ------------- */

/* Packed rows in pinned host memory owned by the context (one set per staging slot).  The per-record fields are
 * those of br_device_rows; x (detail) is NULL unless br_ctx_set_param("host_detail", 1).  mate_idx is the mate index
 * the device computed for the batch (process_pairs, src/bramble.cpp:272-311): the input alignment of a row of leader
 * i is i when BR_ROW_FIRST is set, else mate_idx[i]. */
typedef struct br_host_rows {
  int64_t n_rows, n_aln, n_groups, n_pool_words;
  const br_row_a *a;
  const uint64_t *cigar;
  const uint32_t *pool;
  const uint64_t *row_off;  /* n_aln + 1 */
  const int32_t *mate_idx;  /* n_aln */
  const br_row_x *x;
  const double *similarity_score;
  const int32_t *clip_score;
  uint64_t total_complete, total_unique, dropped_reads, total_processed;
} br_host_rows;

/* convert_reads minus BAM writing for a flat host batch in three steps, so that batch k + 1 uploads and batch
 * k - 1 downloads while batch k is projected (one host thread suffices):
 *   br_batch_stage(ctx, batch, slot)      queues the uploads of `batch` on the context's copy stream (slot 0 / 1);
 *                                         the arrays must stay valid, and should be pinned (br_pin_host), until
 *                                         br_project_staged(slot) has returned;
 *   br_project_staged(ctx, cfg, slot, out) computes the input contract on the device (read-name groups, mate index,
 *                                         the group's shared sequence: what br_batch_prepare / br_batch_seq_source
 *                                         do on the host), projects, and queues the download of the packed rows;
 *   br_host_rows_wait(ctx, slot)          returns when that slot's rows are in host memory; they stay valid until
 *                                         the slot's next br_project_staged.
 * br_project_batch_packed = the three in sequence on slot 0. */
int br_batch_stage(br_ctx *, const br_batch *, int slot);
int br_project_staged(br_ctx *, const br_config *, int slot, br_host_rows *out);
int br_host_rows_wait(br_ctx *, int slot);
int br_project_batch_packed(br_ctx *, const br_config *, const br_batch *, br_host_rows *out);
/* Page-locks caller memory for asynchronous transfers (hipHostRegister / hipHostUnregister behind plain C). */
int br_pin_host(void *p, size_t bytes);
int br_unpin_host(void *p);

/* AoS convenience mirroring project_group_with (bramble-rs/src/api.rs:285-464):
 * all alignments of ONE query name in, one br_projected per emitted record out
 * (array owned by the context; alignments with ref_id < 0 are skipped, api.rs:316-318).
 * The shape is the Rust library's, the values are the C++ path's.
 * NOTE FOR CALLERS COMING FROM bramble-rs -- PAIRING DIFFERS: mates pair up by the C++ rule
 * (process_pairs, src/bramble.cpp:272-311: the LEFT mate, mate_start > start, is entered
 * under name + position and only a later RIGHT mate looks it up), not by the order-independent
 * mutual match of find_mate_pairs (groups.rs:125-193).  A group that lists the right-hand mate
 * BEFORE the left-hand one is a pair in bramble-rs and two unpaired alignments here (and in the
 * C++ binary: groups.rs:131-140 describes exactly this).  There is no switch: the C++ path is
 * the results authority of this library.  hit_index is carried for layout parity only (neither
 * rule reads it).  br_project_groups takes any number of name-collated groups in one call
 * (the Rust CLI hands over 64 at a time, bramble-cli/src/pipeline.rs:29): one trip through
 * the device pipeline instead of one per group. */
typedef struct br_alignment { /* GenomicAlignment, api.rs:73-126 */
  const char *query_name;
  int32_t ref_id;
  int64_t ref_start;
  uint8_t is_reverse, is_paired, is_first_in_pair, mate_is_unmapped;
  char xs_strand, ts_strand; /* 0 = absent */
  int32_t hit_index;
  int32_t mate_ref_id;       /* -1 = none */
  int64_t mate_ref_start;    /* 0 = none */
  const uint32_t *cigar;     /* BAM-packed */
  uint32_t n_cigar;
  const char *sequence;      /* ASCII or NULL */
  uint32_t sequence_len;
  uint32_t read_len;
} br_alignment;

typedef struct br_projected { /* ProjectedAlignment, api.rs:135-176 */
  uint32_t transcript_id;
  uint32_t transcript_start; /* align_pos (groups.rs:362-368): fwpos / rcpos, 0-BASED like the Rust code and the BAM
                              * POS field (the doc comment at api.rs:141-142 says 1-based; the code does not add 1) */
  uint32_t transcript_end;   /* transcript_start + aligned_len - 1, saturating */
  uint32_t aligned_len, query_aligned_len;
  uint8_t is_reverse;        /* api.rs:453 <- evaluate.rs:1062: transcript strand != the read's inferred strand
                              * (infer_strand, api.rs:470-489: XS, else ts flipped on a reverse read, else '.': then 1) */
  char transcript_strand;    /* '+' / '-': AlignInfo::strand of the C++ path (what src/bam.cpp:549-553 acts on) */
  double similarity_score;
  uint32_t nh, hi;
  uint8_t is_primary, same_transcript_as_mate, is_paired_out;
  int32_t insert_size;
  uint64_t input_index;
  uint32_t mapq;
  const uint32_t *cigar;     /* rewritten CIGAR, owned by the context */
  uint32_t n_cigar;
} br_projected;

int br_project_group(br_ctx *, const br_config *, const br_alignment *alns, size_t n,
                     const br_projected **out, size_t *n_out);
int br_project_groups(br_ctx *, const br_config *, const br_alignment *alns, size_t n,
                      const br_projected **out, size_t *n_out);

/* ---- BAM record re-encoding (next row of the scope table: the data format after the path) -- */

/* The batch's original alignment records in HBM: BAM file layout starting at refID
 * (without the 4-byte block_size), record i = blob[rec_off[i] .. rec_off[i+1]). */
typedef struct br_device_records {
  const uint8_t *blob;
  const uint64_t *rec_off; /* n_aln + 1 (n_aln suffices when rec_len is given) */
  int64_t n_aln;
  const uint32_t *rec_len; /* n_aln record lengths (the BAM block_size), or NULL: rec_off[i+1]-rec_off[i] */
} br_device_records;

typedef struct br_device_bam {
  const uint8_t *data;     /* [block_size][record] per emitted row, concatenated (uncompressed BAM stream) */
  uint64_t n_bytes;
  const uint64_t *row_off; /* n_rows + 1 */
  int64_t n_rows;
} br_device_bam;

/* Replaces the per-record byte work of write_to_bam (src/core.cpp:96-212: update_cigar,
 * NH/HI/AS tags, XS/ts deletion, reverse_complement_bam, set_mate_info; src/bam.cpp:474-702)
 * for every row of the LAST br_project_batch_device call on this context.  The rows must have
 * been produced with read names (is_primary decides the secondary flag).  A rewritten CIGAR of more
 * than 65535 ops is written the way htslib's bam_write1 writes it: <l_seq>S<ref_len>N in the CIGAR
 * field, the ops in a CG:B,I tag behind the other tags (SAM spec 4.2.2); an input record in that form
 * is read from its tag and loses it, as after bam_read1 (the reference sees records only through
 * htslib, include/bramble.h:29-85).  BR_ERR_UNSUPPORTED only when such a CIGAR spans 2^28 reference
 * bases or more (bam_write1 fails there too). */
int br_bam_encode_device(br_ctx *, const br_config *, const br_device_records *, void *stream,
                         br_device_bam *out);

/* ---- BAM bundle entry: raw alignment records in, raw projected records out ---------------- */

/* Replaces the reader side of process_reads / process_read_in (src/bramble.cpp:313-441: start,
 * refid, XS/ts strand inputs via GSamRecord::tag_char1, gclib/GSam.cpp:310-318; process_pairs,
 * src/bramble.cpp:272-311; the name group's shared sequence, src/core.cpp:353-378), then
 * convert_reads and write_to_bam, for one bundle of raw BAM records that is resident in HBM:
 * every field is extracted from the records on the device.  Records must be mapped
 * (process_reads skips unmapped ones, bramble.cpp:376-379), name-collated, and the bundle must
 * end at a read-name boundary.  ref_map (HOST memory) maps the input header's refID to the
 * annotation's reference index (position in br_index_build's `refnames`), -1 or any id >= br_index_num_refs for names
 * the annotation lacks.  rows_out may be NULL. */
int br_project_bam_device(br_ctx *, const br_config *, const br_device_records *, const int32_t *ref_map,
                          int32_t n_ref_map, void *stream, br_device_rows *rows_out, br_device_bam *out);

/* Host-memory form of the same call (uploads the records, downloads the stream). */
typedef struct br_bam_bundle {
  const uint8_t *blob;      /* uncompressed BAM alignment section */
  uint64_t n_bytes;
  const uint64_t *rec_off;  /* n_records: offset of each record's refID word (just after its block_size) */
  const uint32_t *rec_len;  /* n_records: its block_size */
  int64_t n_records;
  const int32_t *ref_map;
  int32_t n_ref_map;
  int32_t bgzf_on_device;   /* 1: deflate the projected stream on the device; br_host_bam.data then holds complete
                             * BGZF blocks (append them to the output file as they are) */
} br_bam_bundle;

typedef struct br_host_bam {
  const uint8_t *data;      /* [block_size][record]... ready for BGZF framing; owned by the context, valid until
                             * the SECOND next br_project_bam_bundle call (two pinned buffers alternate, so a
                             * writer thread can compress bundle k while bundle k+1 is projected) */
  uint64_t n_bytes;
  int64_t n_rows;
  uint64_t total_complete, total_unique, dropped_reads, total_processed;
} br_host_bam;

int br_project_bam_bundle(br_ctx *, const br_config *, const br_bam_bundle *, br_host_bam *out);

/* The same call in two steps, so that the upload of the next bundles (own copy stream; may be issued from another
 * host thread) overlaps the projection of the current one: stage bundle k into slot k % 3, later project it from that
 * slot.  A slot may be staged again once its br_project_bam_staged call has returned; the bundle's host memory must
 * stay valid until then. */
int br_bam_bundle_stage(br_ctx *, const br_bam_bundle *, int slot /* 0..2 */);
int br_project_bam_staged(br_ctx *, const br_config *, const br_bam_bundle *, int slot, br_host_bam *out);
/* br_project_bam_staged that does not wait for its result to arrive: with bgzf_on_device the counters, n_bytes and the
 * address in out->data are final when the call returns, the bytes behind it may still be crossing PCIe (on a stream of
 * their own, beside the next bundle's kernels).  br_host_bam_wait(ctx, out) -- from any one thread -- returns when they are
 * there; the buffer rule of the br_host_bam structure -- valid until the second next call -- is unchanged.  Without bgzf_on_device the call
 * is br_project_bam_staged. */
int br_project_bam_staged_nowait(br_ctx *, const br_config *, const br_bam_bundle *, int slot, br_host_bam *out);
int br_host_bam_wait(br_ctx *, const br_host_bam *);

/* br_bam_split on the device: `data` is an inflated BAM alignment section in HBM that starts at a record (n_bytes of it);
 * n_ref = the header's reference count (the boundary search tests reference ids against it).  recs->blob = data, rec_off /
 * rec_len = device arrays of the MAPPED records in stream order (owned by the context, valid until its next call),
 * *n_unmapped the records skipped, *consumed the first byte that belongs to no complete record.  Segments of the stream
 * guess their first record and a verification pass against the real chain makes the result exact (split_kernels.hip).
 * BR_ERR_INVALID_ARG on a record whose fixed fields overrun its block_size. */
int br_bam_split_device(br_ctx *, const uint8_t *data, uint64_t n_bytes, int32_t n_ref, void *stream, br_device_records *recs,
                        int64_t *n_unmapped, uint64_t *consumed);

/* A BAM reader on the device: BGZF bytes of the file in (host memory, e.g. the mapped file), bundles of device-resident records
 * of whole read-name groups out -- inflate, record split and the cut at a read-name change (src/bramble.cpp:313-441 reads
 * record by record through htslib and flushes a bundle at a name change) without the host touching an inflated byte.  Needs
 * no index: it can run beside the guide loading.
 *   br_bam_reader_new      header_bytes = inflated size of the BAM header (magic, text, references): the first record follows it
 *   br_bam_reader_next     takes the complete BGZF blocks at the front of data[0 .. n_bytes) (at most ~200 MB inflated per call;
 *                          *consumed says how far it got -- call again from there; `last` != 0: data ends with the file) and
 *                          returns the records of every read-name group that is complete: bundle->blob / rec_off / rec_len in
 *                          HBM (mapped records only, like br_bam_split; *n_unmapped = the ones skipped), valid until
 *                          br_bam_reader_release(id).  The last, possibly unfinished group stays behind for the next call;
 *                          the call that ends the file returns everything.  A bundle may be empty.
 * BR_ERR_INVALID_ARG for malformed BGZF / BAM, a block whose CRC32 is wrong, a file that ends inside a block or a record; a
 * reader that has returned an error is to be freed.  br_bam_reader_next is called from one thread; br_bam_reader_release may
 * come from another (the thread that projects the bundles). */
typedef struct br_bam_reader br_bam_reader;
int br_bam_reader_new(int device, int32_t n_ref, uint64_t header_bytes, br_bam_reader **out);
int br_bam_reader_next(br_bam_reader *, const uint8_t *data, uint64_t n_bytes, int last, uint64_t *consumed,
                       br_device_records *bundle, int64_t *id, int64_t *n_unmapped);
int br_bam_reader_set_piece_blocks(br_bam_reader *, int64_t blocks);   /* BGZF blocks per br_bam_reader_next call (default 3072) */
int br_bam_reader_release(br_bam_reader *, int64_t id);
/* Piece-wise form (round 4: several readers on several devices, uploads beside the inflating of the piece before).  The caller
 * holds the whole file's block table (br_bgzf_scan over the mapping) and hands out pieces [b0, b1) of it, to one reader in
 * order or to several readers: a piece needs nothing from its neighbours.  A piece owns the records between two cuts that
 * are defined the same way at either end (the read-name group that straddles a piece boundary goes to the piece in front,
 * which inflates blocks up to b1x > b1 to find its end); a piece that does not know where a record starts in its first
 * block (start_rel = -1) guesses, and reports where it started: when that differs from the end_rel its neighbour in front
 * reports, the caller processes the piece again with start_rel = that end_rel (the result never depends on a guess).
 *   br_bam_piece_upload    compressed bytes of blocks [b0, b1x) into upload slot 0 / 1 (its own stream; may be called from a
 *                          second thread, one piece ahead of br_bam_piece_process)
 *   br_bam_piece_process   inflate + record split + cuts of the slot's piece [b0, b1) -> bundle (valid until
 *                          br_bam_reader_release(id)) and info.  start_rel >= 0: inflated bytes from the start of block b0 to
 *                          the piece's first record (the BAM header's size for the first piece).  Returns 1 (BR_PIECE_MORE)
 *                          when the end cut lies beyond block b1x: upload up to a larger b1x and call again. */
#define BR_PIECE_MORE 1
/* one BGZF block: its deflate payload in the file, where its bytes go in the inflated stream (br_bgzf_scan) */
typedef struct br_bgzf_block { uint64_t src_off, dst_off; uint32_t clen, ulen, crc, pad; } br_bgzf_block;
typedef struct br_piece_info {
  uint64_t start_rel;   /* where the bundle starts: inflated bytes behind the start of block b0 */
  uint64_t end_rel;     /* where it ends: inflated bytes behind the start of block b1 (= the next piece's start_rel) */
  int64_t n_unmapped;   /* unmapped records between the two (not in the bundle) */
  int32_t guessed, at_end;
} br_piece_info;
int br_bam_piece_upload(br_bam_reader *, int slot, const uint8_t *file, uint64_t file_bytes, const br_bgzf_block *blocks,
                        int64_t n_blocks, int64_t b0, int64_t b1x);
int br_bam_piece_process(br_bam_reader *, int slot, const br_bgzf_block *blocks, int64_t n_blocks, int64_t b1, int64_t start_rel,
                         br_device_records *bundle, int64_t *id, br_piece_info *info);
double br_bam_reader_seconds(const br_bam_reader *);   /* time spent inside br_bam_piece_process so far */
double br_bam_reader_upload_seconds(const br_bam_reader *);   /* ... inside br_bam_piece_upload (host copies into pinned buffers + queueing) */
void br_bam_reader_free(br_bam_reader *);
/* br_project_bam_staged / _nowait for records that are in HBM already (a br_bam_reader bundle, or br_bam_split_device's) */
int br_project_bam_resident(br_ctx *, const br_config *, const br_device_records *recs, const int32_t *ref_map, int32_t n_ref_map,
                            int bgzf_on_device, int nowait, br_host_bam *out);

/* BGZF inflate on the device (one wave per block: the reader side of br_bgzf_deflate_device).  br_bgzf_scan (host) walks
 * the block headers of a piece of a BGZF file -- up to `cap` complete blocks; empty ones (the EOF marker) are stepped over --
 * and lists for every block where its DEFLATE payload lies in `data`, where its bytes go in the inflated stream (dst_off: a
 * running sum from 0), and the CRC32 / ISIZE of its trailer; *consumed is where the next call continues.
 * br_bgzf_inflate_device inflates the listed blocks of the same bytes in HBM (`src`, `n_src`); *out is a device pointer to
 * the inflated stream, valid until the next call on the context.  BR_ERR_INVALID_ARG for a malformed header, or for a block
 * that does not inflate to ISIZE bytes with its CRC32 (what htslib's bgzf_read reports as a read error). */
int br_bgzf_scan(const uint8_t *data, uint64_t n_bytes, int64_t cap, br_bgzf_block *blocks, int64_t *n_blocks,
                 uint64_t *consumed, uint64_t *out_bytes);
int br_bgzf_inflate_device(br_ctx *, const uint8_t *src, uint64_t n_src, const br_bgzf_block *blocks, int64_t n_blocks,
                           void *stream, const uint8_t **out, uint64_t *out_bytes);

/* Walks the block_size chain of an uncompressed BAM alignment section (host): fills rec_off /
 * rec_len for up to `cap` MAPPED records (unmapped ones are counted and skipped like
 * bramble.cpp:376-379) and reports how many bytes were consumed (a trailing partial record is
 * left for the next call).  BR_ERR_INVALID_ARG on a record whose fixed fields overrun its
 * block_size. */
int br_bam_split(const uint8_t *data, uint64_t n_bytes, int64_t cap, uint64_t *rec_off, uint32_t *rec_len,
                 int64_t *n_records, int64_t *n_unmapped, uint64_t *consumed);

/* BGZF-compress n bytes that sit in HBM (one wave per 56 KiB block: hash-table LZ77, per-block dynamic Huffman codes --
 * or the fixed code with br_ctx_set_param("deflate_dynamic", 0) -- and CRC32);
 * *out is a device pointer to the concatenated blocks (no EOF marker), valid until the next call on the context. */
int br_bgzf_deflate_device(br_ctx *, const uint8_t *src, uint64_t n, void *stream, const uint8_t **out, uint64_t *out_bytes);

/* ---- annotation loading and the command line (scope table rows f-1 / f-3) ----------------- */

/* GTF / GFF3 (plain or gzip) -> transcripts in the reference's guide order: gclib GffReader with
 * transcripts only, sorted by location, reference names compared lexicographically
 * (src/bramble.cpp:497-512; gfo_cmpByLoc, gclib/gff.cpp:75-90), so that position = the tid the
 * reference assigns through its output header (src/bramble.cpp:559-596).  Exons are 1-based
 * half-open [start, end+1) like src/bramble.cpp:164-165.  The arrays stay owned by the handle and
 * feed br_index_build directly. */
typedef struct br_annotation br_annotation;
int br_annotation_load(const char *path, br_annotation **out);
/* the same with `threads` workers taking the lines apart (br_annotation_load: up to 16); what the lines mean depends on
 * the lines before them and is applied in file order whatever the thread count */
int br_annotation_load_mt(const char *path, int threads, br_annotation **out);
void br_annotation_free(br_annotation *);
size_t br_annotation_num_transcripts(const br_annotation *);
const br_transcript *br_annotation_transcripts(const br_annotation *);
size_t br_annotation_num_refs(const br_annotation *);            /* reference names in order of first appearance */
const char *const *br_annotation_refnames(const br_annotation *);

/* The reference's command line (src/bramble.cpp:443-485): in.bam -G -o [-S] [-p] [--fr|--rf]
 * [--lr|--lr-hq] [--strict] [--max-*] [--similarity-threshold] [--quiet], plus --device-deflate (default: BGZF blocks made on the
 * GPU) / --host-deflate / --compression-level N (host codec), --device-reader (default for a regular file on one device: the
 * input is inflated and split into records on the GPU, br_bam_reader) / --host-reader, --bundle-size and --device / --devices.
 * Returns the process exit code. */
int br_cli_main(int argc, char **argv);
/* For a process whose only job is that one call (the `bramble` binary): with `on` != 0 br_cli_main does not return after a
 * run that got as far as the final report -- the output file is closed and renamed, the streams are flushed, and the process
 * leaves through _exit(code) without unwinding gigabytes of record buffers, pinned memory and device allocations first
 * (bramble-cli/src/main.rs:56-60 keeps its index in a ManuallyDrop for the same reason).  Off by default: a host that calls
 * br_cli_main as a function gets every device and pinned allocation of the run released before the call returns. */
void br_cli_exit_at_end(int on);
/* First touch of a device (HIP runtime start-up, context creation): callable from a thread of its own so that it overlaps
 * other start-up work.  BR_ERR_NO_DEVICE when the device does not exist. */
int br_device_warmup(int device);

/* BGZF container utilities (host only; what the reference gets from htslib's bgzf layer behind
 * GSamReader / GSamWriter): whole-file inflate / deflate on `threads` host threads.  The reader
 * verifies each block's CRC32; the writer emits 0xff00-byte payload blocks and the EOF marker. */
int br_bgzf_write_file(const char *path, const uint8_t *data, uint64_t n, int threads, int level);
int br_bgzf_read_file(const char *path, int threads, uint8_t **out, uint64_t *n); /* free with br_free_buffer */
void br_free_buffer(uint8_t *);
const char *br_bgzf_codec(void); /* "libdeflate" (bound at run time when present) or "zlib" */

/* ---- measurement hooks ------------------------------------------------------ */

/* Kernel names reported by br_ctx_kernel_ms / rocprof. */
#define BR_K_SEGMENT 0    /* k_segment */
#define BR_K_COUNT 1      /* k_project<G,false> (count pass; with "count_split" the main kernel, without the exon walk) */
#define BR_K_EMIT 2       /* k_emit_dense (general class; the whole list for long-read presets) */
#define BR_K_PAIR_COUNT 3 /* k_pair */
#define BR_K_PAIR_EMIT 4  /* k_pair<true> (per-record {match, input, NH, HI | flags}) */
#define BR_K_ROWS 5       /* k_rows (the packed row table) */
#define BR_K_SCAN 6       /* k_scan_* */
#define BR_K_EMIT_AUX 7   /* k_project<64,true> (alignments with > 64 candidate rows) */
#define BR_K_KSW 8        /* k_ksw (-S clip rescue DP) */
#define BR_K_BAM 9        /* k_bam_scan + k_bam_size + k_bam_tasks (or k_bam_encode<G>) */
#define BR_K_PARSE 10     /* k_rec_fields + k_group_off + k_rec_copy + k_mates* + k_seq_* */
#define BR_K_CODEC 11     /* k_deflate_dynamic | k_deflate_fixed, k_bgzf_compact; k_inflate */
#define BR_K_EMIT_SIMPLE 12 /* k_emit_dense, simple class (one read exon from a single M op) */
#define BR_K_PRIMARY 13   /* k_primary (+ the per-read-name counters) */
#define BR_K_CIGAR_POOL 14 /* unused since ABI version 2 */
#define BR_K_COUNT_WALK 15 /* k_project<G,false,false,2>: the deferred alignments of the split count pass, with the exon walk */
#define BR_K_EXPAND 16    /* k_expand (emit work list) */
#define BR_K_GROUP_IDS 17 /* k_group_ids */
#define BR_K_P1 18        /* k_project1<G,1>: single-pass count + emit, main kernel (simple class written, general class listed) */
#define BR_K_P1_WALK 19   /* k_project1<G,2>: the alignments that need the exon walk */
#define BR_K_EMIT_WL 20   /* k_emit_wl: the general class from the single pass's work list */
#define BR_K_NAME_SEED 21  /* k_name_seed: first mt19937_64 output per read name (direct rows) */
#define BR_K_PAIR_MASK 22  /* k_pair_mask: pairing on the survivor sets, before the emit pass */
#define BR_K_PAIR_BIG 23   /* k_big<0> + k_pair_big: the same for alignments with > 64 candidate rows */
#define BR_K_GROUP_DESC 24 /* k_group_desc: NH / HI / primary per read name (+ the per-read-name counters) */
#define BR_K_EXPAND_ROWS 25 /* k_expand_rows: emit work list + emit descriptors */
#define BR_K_EMIT_ROWS_SIMPLE 26 /* k_emit_rows<1>: packed rows of the simple class */
#define BR_K_EMIT_ROWS 27  /* k_emit_rows<2|0>: packed rows of the general class (or of everything) */
#define BR_K_BIG_EMIT 28   /* k_big<1>: packed rows of the alignments with > 64 candidate rows */
#define BR_K_NUM 29
/* When enabled, every launch is bracketed by hipEvents on the launch stream. */
int br_ctx_set_profiling(br_ctx *, int enabled);
/* Launch tuning: "group_lanes" (8|16|32|64 lanes cooperating on one alignment),
 * "blocks_per_cu" (grid size of the grid-stride projection kernels), "bam_lanes" (0, the default: a wave per 32
 * re-encoded records, their byte regions as 16-byte copy tasks; 4..64: that many lanes per record), "deflate_dynamic" (1: per-block Huffman codes, 0: the fixed code), "emit_split" (1: the emit
 * work list is launched per class, 0: one launch), "count_split" (1: short-read presets run the count pass as a main
 * kernel without the exon walk plus a second one for the alignments that need it, 0: one kernel). */
int br_ctx_set_param(br_ctx *, const char *key, int64_t value);
/* Device time (ms) of kernel `which` during the last projection call, summed
 * over its launches; *launches receives the launch count. */
int br_ctx_kernel_ms(br_ctx *, int which, double *ms, int32_t *launches);
/* ... summed over every call since br_ctx_set_profiling(ctx, 1) (a timed loop reads it once, behind its last step) */
int br_ctx_kernel_ms_sum(br_ctx *, int which, double *ms, int64_t *launches);
/* Diagnostic pass (never part of a timed region): computes the exact counters below
 * for the device batch the context projected last. */
int br_ctx_collect_counters(br_ctx *, const br_device_batch *, void *stream);
/* Exact algorithmic byte counters (SURVEY.md 8d formula), after br_ctx_collect_counters:
 * out[0]=B_in, out[1]=B_idx, out[2]=B_out, out[3]=sum n_cigar, out[4]=read exons,
 * out[5]=overlap hits, out[6]=matches, out[7]=rewritten-CIGAR words over all matches. */
int br_ctx_last_counters(br_ctx *, uint64_t out[8]);

/* -S runs: out[0] = rescue problems of the last call, out[1] = ksw2 DP cells (sum of qlen x tlen),
 * out[2] = accepted rescues, out[3] = coded sequence bytes. */
int br_ctx_rescue_stats(br_ctx *, uint64_t out[4]);
/* Diagnostic: routing of the last call's rescue DP problems: pieces, problems per register-array shape (64 / 128 / 256 /
 * 384 target columns), problems left to the general kernel before the DP, direction-tape bytes of the largest piece, 0,
 * leftovers of the last piece after the DP, tape rows (array steps) per shape [4], 0 x 4.  Keys of br_ctx_set_param: "ksw_fast" (0 = general kernel only),
 * "ksw_tape_mb" (tape budget; larger batches are processed in pieces), "ksw_tape_pct" (test hook: share of the tape the
 * array kernels may use). */
int br_ctx_ksw_diag(br_ctx *, uint64_t out[16]);

/* Diagnostic: the -S rescue DP alone (k_ksw = ksw_extz2_sse as src/evaluate.cpp:296-313 calls it, on the device).
 * n (target, query) ASCII pairs in; per pair: ok[p] = the rescue would be accepted (max >= 10 and the walk reached the
 * last cell, src/evaluate.cpp:484,643), max[p], and -- when ok -- the traceback CIGAR in forward order (BAM-packed
 * M / I / D) at cigar[p * cigar_cap ...], n_cigar[p] ops. */
int br_ctx_ksw_pairs(br_ctx *, int64_t n, const char *const *tseq, const char *const *qseq, int32_t *ok, int32_t *max,
                     uint32_t *n_cigar, uint32_t *cigar, uint32_t cigar_cap);

/* The reference's primary tie-break (src/core.cpp:214-218,298-299):
 * uniform_int_distribution<uint32_t>(0, n_tied-1)(mt19937_64(std::hash<std::string>(name))),
 * restated bit-exactly for libstdc++ (GCC 11, x86-64). */
uint32_t br_primary_pick(const char *name, size_t len, uint32_t n_tied);

const char *br_version(void);
const char *br_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* BRAMBLE_AMD_H */
