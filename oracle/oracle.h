/* TEST INFRASTRUCTURE ONLY -- C interface of the CPU oracle (liboracle.so).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (bramble_amd/) never does.  See oracle_core.hpp for
 * what the oracle restates and how far it is pinned.
 *
 * All arrays are plain C arrays owned by the caller (inputs) or by the result
 * handle (outputs, valid until orc_result_free).
 */
#ifndef BRAMBLE_ORACLE_H
#define BRAMBLE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_flags {
  int32_t lr, lr_hq, strict, use_fasta, fr, rf;
  int32_t has_max_clip, has_max_junc_ins, has_max_junc_gap, has_sim_thr, has_max_error_exon;
  uint32_t max_clip, max_junc_ins, max_junc_gap, max_error_exon;
  float sim_thr;
} orc_flags;

/* Name-collated alignments, struct-of-arrays.  ref_start / mate_start are
 * 1-based (mate_start 0 = none), cigar words are BAM-packed (len<<4|op). */
typedef struct orc_batch {
  int64_t n_aln;
  const int32_t *ref_id;
  const int32_t *ref_start;
  const uint16_t *flags;      /* SAM flag bits */
  const int8_t *xs;           /* first char of XS tag or 0 */
  const int8_t *ts;           /* first char of ts tag or 0 */
  const uint64_t *cigar_off;  /* n_aln + 1 */
  const uint32_t *cigar;
  const int32_t *mate_ref_id;
  const int32_t *mate_start;
  const uint64_t *name_off;   /* n_aln + 1 */
  const char *names;
  const uint64_t *seq_off;    /* n_aln + 1, or NULL (no sequences) */
  const char *seqs;           /* ASCII bases */
  const int32_t *l_qseq;
} orc_batch;

/* Evaluate-level output: matches of every alignment after the similarity
 * filter, ascending tid within an alignment. */
typedef struct orc_matches {
  int64_t n_aln, n_matches;
  const uint64_t *aln_off;    /* n_aln + 1 */
  const uint32_t *tid, *fwpos, *rcpos;
  const int8_t *strand;
  const double *similarity_score, *total_coverage, *total_operations;
  const int32_t *junc_hits, *ref_consumed, *clip_score;
  const uint64_t *ideal_off;  /* n_matches + 1 */
  const uint32_t *ideal;
  const uint64_t *out_off;    /* n_matches + 1 */
  const uint32_t *out;        /* merged (rewritten) CIGAR */
  const int32_t *n_exons;     /* per alignment: read exon count */
  const int32_t *mate_idx;    /* per alignment: paired mate index or -1 */
} orc_matches;

/* Row-level output: one entry per emitted BAM record, in emission order. */
typedef struct orc_rows {
  int64_t n_rows;
  const int32_t *input_index;
  const uint32_t *tid, *pos;
  const int8_t *strand;
  const uint64_t *cigar_off;  /* n_rows + 1 */
  const uint32_t *cigar;
  const double *similarity_score;
  const int32_t *clip_score, *junc_hits, *ref_consumed;
  const uint32_t *nh, *hi, *mapq;
  const uint8_t *primary, *is_paired, *same_transcript, *is_first;
  const int32_t *mate_tid, *mate_pos, *isize;
  const uint32_t *group;      /* index of the read-name group */
  /* counters of src/bramble.cpp:729-736 */
  uint64_t total_complete, total_unique, dropped_reads, total_processed;
} orc_rows;

typedef struct orc_index orc_index;
typedef struct orc_result orc_result;

orc_index *orc_index_new(void);
/* exons: n_exons pairs (start, end) 1-based half-open, any order (sorted by start
 * inside).  ref_seq may be NULL.  Returns the tid. */
int64_t orc_index_add_transcript(orc_index *, int32_t ref_id, char strand, const char *name,
                                 const uint32_t *exons, int32_t n_exons, const char *ref_seq,
                                 int64_t ref_seq_len);
void orc_index_finish(orc_index *);
int64_t orc_index_num_transcripts(const orc_index *);
uint32_t orc_index_transcript_len(const orc_index *, int64_t tid);
void orc_index_free(orc_index *);

/* Whole-path run (convert_reads restatement).  n_threads <= 1: one thread.
 * want_matches: also keep the evaluate-level table (costly; tests only). */
orc_result *orc_run(const orc_index *, const orc_flags *, const orc_batch *, int32_t n_threads,
                    int32_t want_matches);
const orc_rows *orc_result_rows(const orc_result *);
const orc_matches *orc_result_matches(const orc_result *);
double orc_result_seconds(const orc_result *);  /* wall time of the projection itself */
void orc_result_free(orc_result *);

/* Unit-level entry points used by the golden-vector tests. */
/* real (+) ideal -> out; returns n_out (out must hold n_real + n_ideal + 1 words) */
int32_t orc_merge_cigar(const uint32_t *real, int32_t n_real, const uint32_t *ideal, int32_t n_ideal,
                        uint32_t *out);
/* CIGAR -> read exons; returns count or -1 where the reference aborts */
int32_t orc_segments(int32_t ref_start, const uint32_t *cigar, int32_t n_cigar, uint32_t *out_pairs,
                     int32_t cap);
/* resolved presets: out[0..4] = max_clip, max_junc_ins, max_junc_gap, max_error_exon,
 * ignore_small_exons; thr_out = threshold; returns filter_by_similarity */
int32_t orc_resolve_config(const orc_flags *, uint32_t *out5, float *thr_out);
/* write_to_bam over the rows of a finished run (records: BAM layout from refID on); returns the
 * byte count of the uncompressed BAM stream placed in *out (free with orc_free_buffer) */
int64_t orc_bam_encode(const orc_result *, const uint8_t *blob, const uint64_t *rec_off, const uint32_t *rec_len,
                       int64_t n_aln, int32_t long_reads, uint8_t **out);
/* reader side (process_reads / process_read_in, src/bramble.cpp:313-441): raw mapped BAM records ->
 * the batch orc_run takes.  rec_len may be NULL (records contiguous: rec_off has n + 1 entries). */
typedef struct orc_parsed orc_parsed;
orc_parsed *orc_bam_parse(const uint8_t *blob, const uint64_t *rec_off, const uint32_t *rec_len, int64_t n,
                          const int32_t *ref_map, int32_t n_ref_map);
const orc_batch *orc_parsed_batch(const orc_parsed *);
void orc_parsed_free(orc_parsed *);
void orc_free_buffer(uint8_t *);
/* primary tie-break: get_rand(n_tied, std::hash<std::string>(name)), src/core.cpp:214-218,298-299 */
uint32_t orc_primary_pick(const char *name, int64_t len, uint32_t n_tied);
/* ksw2 extension as bramble calls it (ASCII in): returns n_cigar, fills score/max */
int32_t orc_ksw_align(const char *tseq, const char *qseq, int32_t *score, int32_t *max, uint32_t *cigar,
                      int32_t cap);

#ifdef __cplusplus
}
#endif
#endif
