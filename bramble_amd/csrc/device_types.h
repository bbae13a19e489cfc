// Shared host/device plain structs of the projection pipeline (gfx950 only).
#pragma once
#include <stdint.h>

namespace br {

// Flattened exon index in HBM (replaces the per-(refid,strand) cgranges trees of
// include/g2t.h:47-77).  A "slab" is one (refid, strand) pair: slab = 2*refid +
// (strand == '-').  Within a slab, transcript-exon rows are sorted by start.
struct DevIndex {
  uint32_t n_refs;
  uint32_t n_tx;
  uint32_t n_rows;             // transcript-exon rows over all slabs
  const uint32_t *slab_off;    // [2*n_refs + 1] row ranges
  const uint32_t *s_start;     // [n_rows] exon start (1-based, inclusive)
  const uint32_t *s_pmax;      // [n_rows] running max of s_end inside the slab
  const uint32_t *s_tid;       // [n_rows] tid alone (rank lookups)
  const uint4 *s_row;          // [2*n_rows] 32-byte row: {start, end (exclusive), start of the transcript's next exon
                               // (genomic order; ~0u if none), pos_start}, {tid, genomic exon idx (bit 31: the transcript has > 256 exons), first row of tid in tx_ex, end of that next exon}
  const uint4 *tx_ex;          // per transcript: exons in genomic order {start, end, pos_start, seq_off},
                               // closed by a sentinel {~0u, ~0u, 0, 0}
  const uint32_t *tx_first;    // [n_tx + 1] first tx_ex row of each transcript (incl. sentinels)
  const uint8_t *seq_pool;     // exon sequences (only with -S)
  // bucket tables over genomic coordinate (bin = coord >> bin_shift), per reference, both strands' slabs in one
  // 16-byte entry {lo+, hi+, lo-, hi-}: hi = first row of the slab with start >= b<<shift, lo = first row whose running
  // max end exceeds b<<shift (a read's two table words per strand come with two loads, usually from one line).
  // Reference r owns entries [bin_off[r], bin_off[r+1]).
  uint32_t bin_shift;
  const uint32_t *bin_off;     // [n_refs + 1]
  const uint4 *t_bin;
};

// Resolved evaluator thresholds (ReadEvaluationConfig, include/evaluate.h:275-285)
struct DevCfg {
  uint32_t max_clip, max_junc_ins, max_junc_gap, max_error_exon;
  int32_t ignore_small_exons, filter_by_similarity, long_reads, use_fasta;
  int32_t fr, rf;
  double thr;  // (double)(float)similarity_threshold
};

// Per-alignment metadata written by k_segment
struct AlnMeta {
  uint32_t n_seg;         // read exon count (0: nothing to project)
  uint32_t smode;         // bit0: try '+', bit1: try '-'
  uint32_t n_left_clip;   // leading soft clip (get_clips), long reads only
  uint32_t n_right_clip;
};

enum { ST_FIRST = 0, ST_MIDDLE = 1, ST_LAST = 2, ST_ONLY = 3 };

enum : uint32_t {
  OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8, OP_B = 9,
  OP_MATCH_OVR = 10, OP_DEL_OVR = 11, OP_INS_OVR = 12, OP_CLIP_OVR = 13
};

// row flag bits
enum : uint8_t { RF_PAIRED = 1, RF_SAME_TX = 2, RF_FIRST = 4 };

}  // namespace br
