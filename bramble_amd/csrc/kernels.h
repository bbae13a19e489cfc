// Launch interface between the host pipeline (abi.cpp) and project_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "ksw_types.h"

namespace br {

struct ProjectArgs {
  DevIndex ix;
  DevCfg cfg;
  int64_t n_aln;
  const int32_t *ref_id;
  const uint32_t *cigar_off;
  const uint32_t *cigar;
  const uint2 *seg;
  const AlnMeta *meta;
  const uint4 *head;     // [n_aln] {exon0.start, exon0.end, n_seg, refid<<2|smode}
  const uint4 *head2;    // [n_aln] read exons 1 and 2
  const uint32_t *fast_flag;  // [n_aln] 1: one read exon from a single M op (short-read presets)
  const uint32_t *fast_pre;   // [n_aln + 1] exclusive prefix of the simple alignments' matches
  // count pass outputs / emit pass inputs
  uint32_t *n_matches;   // [n_aln]
  uint4 *ranges;         // [n_aln] candidate row ranges (lo+,hi+,lo-,hi-)
  uint64_t *mask;        // [n_aln] survivor bit per candidate row (<= 64 rows)
  uint32_t *big_list;    // alignments with > 64 candidate rows and >= 1 match
  uint32_t *n_big;       // their count (device counter, zeroed per batch)
  uint32_t *walk_list;   // count pass, MODE 1 -> 2: alignments left to the kernel that carries the exon walk (null: one kernel)
  uint32_t *n_walk;      // their count (device counter, zeroed per batch)
  uint32_t *m_aln;       // [n_matches] alignment of each match slot (k_expand)
  const uint32_t *match_off;  // [n_aln + 1]
  const uint64_t *cig_base;   // [n_aln + 1]
  // match table (emit)
  uint32_t *m_tid, *m_aux;
  uint2 *m_p;            // {pos, n_cigar | minus<<31}: what the packed row needs
  uint2 *m_x;            // {junc_hits, ref_consumed}: the detail column only
  uint4 *m_b;            // {clip_score, 0, similarity lo, similarity hi}
  uint64_t *m_cigoff;
  uint32_t *cig_arena;
  // single-pass count + emit (k_project1 / k_emit_wl; null p1: the two-pass path): the kernel that finds an alignment's
  // survivors takes its slice of the match table, of the general-class work list and of the CIGAR arena from pages its
  // wave owns, and writes match_off / cig_base itself
  uint32_t *match_off_w;
  uint64_t *cig_base_w;
  uint4 *wl;             // work list of the general class: {alignment, slab row | strand << 31, match slot, rank}; ~0u in .x: hole
  uint64_t *p1;          // device counters, see P1_* below
  uint64_t cap_m, cap_w, cap_c;   // capacities of the match table (slots), the work list (entries) and the arena (words)
  // small batches without host round trips: the scan totals stay on the device (tot[0] matches, tot[1] arena words,
  // tot[2] matches of the simple class), the tables were sized from upper bounds lim_m / lim_c, and a kernel that finds the
  // totals beyond them does nothing (the host sees the same totals at the end and takes the ordinary path).  null: the
  // host read the totals and passes them by value
  const uint64_t *tot;
  uint64_t lim_m, lim_c;
  uint64_t cover;        // work-list entries the launch's grid covers (large batches launched from predicted counts): more -> nothing is done
};
__device__ __forceinline__ bool tot_over(const uint64_t *tot, uint64_t lim_m, uint64_t lim_c) { return tot && (tot[0] > lim_m || tot[1] > lim_c); }
// k_project1's counters (u64 each): the three allocators' high-water marks, "a capacity was exceeded", matches found,
// work-list entries written
// (the three allocators sit in different 128-byte lines: same-line atomics serialise in one L2 channel)
enum { P1_M = 0, P1_W = 16, P1_C = 32, P1_OVF = 48, P1_NM = 49, P1_NW = 50, P1_WORDS = 64 };
#define P1_PAGE_M 16384u   // match slots a wave takes from the global counter at a time
#define P1_PAGE_W 4096u    // work-list entries per page (unused entries are filled with hole markers)
#define P1_PAGE_C 262144u  // arena words per page


// -S clip rescue (rescue_kernels.inc)
struct FaArgs {
  const int32_t *seq_src;      // [n_aln] alignment whose sequence the read-name group shares (-1: none)
  const uint32_t *seq_off;     // [n_aln + 1]
  const uint8_t *seqs;         // ASCII read sequences
  uint32_t *n_prob;            // [n_aln] rescue problems per alignment
  uint32_t *seq_bytes;         // [n_aln] coded sequence bytes per alignment
  const uint32_t *prob_off;    // [n_aln + 1]
  const uint64_t *seqarena_off;  // [n_aln + 1]
  KswProb *probs;
  FaSrc *srcs;                 // [n_prob] where the sequences of a problem come from
  KswRes *results;
  uint8_t *seq_arena;          // query / target codes of every problem
  uint32_t *clip_ops;          // clip-segment CIGARs: problem p at seq_off(p) + p
  uint32_t *ideal_cap;         // [n_aln] ideal-CIGAR capacity incl. clip ops
  uint64_t *want_l, *want_r;   // [n_aln] k_project_fa<2>: candidates (by index, alignments with <= 64 candidate rows) that asked for a left / right rescue
};

struct KswArgs {
  int64_t n_prob;
  const KswProb *probs;
  KswRes *results;
  const uint8_t *seq_arena;
  uint32_t *clip_ops;
  uint8_t *scratch;
  size_t scratch_per_wave, pmat_bytes, raw_words;
  int64_t n_waves;   // waves that own a scratch slice
  uint32_t tmax;
  uint64_t *stats;  // [2] DP cells, accepted rescues (may be null)
  // diagnostic (br_ctx_ksw_pairs): the raw traceback CIGAR (forward order) and the maximum of every problem; null otherwise
  uint32_t *raw_out, *raw_n; int32_t *max_out; uint32_t raw_cap;
  // when set: only the problems list[0 .. *n_list) (the ones the streamed DP leaves to this kernel)
  const uint32_t *list; const uint32_t *n_list;
};

// The streamed DP over one range of problems [p0, p0 + n): k_ksw_bin -> k_ksw_dp<G,K> per bin -> k_ksw (the listed
// leftovers) -> k_ksw_trace.  counters (u32[32]): [0..3] problems per bin, [4] leftovers, [5] / [6] longest query / target among them, u64 view [4..7] = [8..15] tape
// rows per bin, [16..19] the bins' problem queues, u64 at [24] tape bytes handed out.
struct KswFastArgs {
  int64_t p0, n;
  const KswProb *probs;
  KswRes *results;
  const uint8_t *seq_arena;
  uint32_t *clip_ops, *raw_ops;
  KswDesc *desc[KSW_N_BINS];       // [n] each
  uint32_t *counters;
  uint32_t *leftover;              // [n] problem indices for k_ksw
  KswDp *dp;                       // [n], indexed by p - p0
  uint8_t *tape; uint64_t tape_cap;
  uint32_t n_bin[KSW_N_BINS], n_groups[KSW_N_BINS];
  uint64_t *stats;
  uint32_t *raw_out, *raw_n; int32_t *max_out; uint32_t raw_cap;
};

struct ScanArgs {
  int64_t n;
  const uint32_t *src32;
  const uint32_t *cigar_off;  // mode 1 only
  const uint4 *head;          // mode 1 only
  const uint32_t *ideal_cap;  // mode 3 only
  const uint32_t *fast_flag;  // scan3 only
  uint64_t *tile_sums;
  int64_t n_tiles;            // scan3 only (set by the launcher)
};

// Packed row table: the product of the row stage (one row per emitted BAM record, rows of one read-name group
// contiguous, a pair's two records adjacent).  Everything else a record carries follows from these words:
//   NH = rows of the group, HI = rank inside the group (also stored), MAPQ = get_mapq(NH) (src/core.cpp:46-58),
//   mate transcript / position / insert size = the adjacent row of the pair (set_mate_info, src/bam.cpp:531-588).
//   r_a[r] = {transcript id, 0-based transcript position, meta, NH}
//   r_c[r] = the rewritten CIGAR itself when it has <= 2 ops, else the 64-bit word offset of its ops in the CIGAR arena
//   r_x[r] = {input alignment, junc_hits, aligned_len (ref_consumed), HI}
#define RM_NCIG 0x00ffffffu      // meta bits 0..23: ops of the rewritten CIGAR
#define RM_MINUS (1u << 24)      // transcript strand '-'
#define RM_PAIRED (1u << 25)     // emitted together with its mate
#define RM_SAME (1u << 26)       // ... on the same transcript
#define RM_FIRST (1u << 27)      // the leader's own record (its mate's record follows)
#define RM_PRIMARY (1u << 28)    // primary record of its read name
struct PairArgs {
  int64_t n_groups, n_aln, n_rows_total;
  int32_t long_reads;
  const uint32_t *group_off;
  const uint32_t *aln_group;  // [n_aln] group of each alignment (k_group_ids)
  const int32_t *mate_idx;
  const uint32_t *match_off;
  const uint32_t *n_matches;
  const uint32_t *m_tid;
  const uint2 *m_p, *m_x;
  const uint4 *m_b;
  const uint64_t *m_cigoff;
  uint32_t *n_rows;         // count pass: records per leader alignment
  uint64_t *pick;           // [n_groups] or null: k_primary<false> leaves the primary record's row here (~0: none) and k_rows sets the bit
  uint8_t *pbit;            // [n_aln] count pass: this leader emits pairs (k_primary counts its units without touching the records)
  uint64_t *pmask;          // [n_aln] count pass -> emit pass: list positions of a pair's common transcripts (lists <= 64)
  const uint64_t *row_off;  // [n_aln + 1] emit pass
  // per record {match, input alignment, NH, HI | RR_* bits}: written by k_pair_emit (one lane per alignment), flagged by
  // k_primary, turned into the packed row by k_rows (one lane per record)
  uint4 *r_rec;
  uint4 *r_a;
  uint2 *r_c;
  uint4 *r_x;
  double *r_sim;            // aux presets only (similarity filter on)
  int32_t *r_clip;
  uint64_t *counters;       // [4] total_complete, total_unique, dropped_reads, a field overflowed its packed width
  const uint64_t *tot;      // small batches (see ProjectArgs): tot[0..1] against lim_m / lim_c, tot[3] = records (n_rows_total is then the tables' capacity)
  uint64_t lim_m, lim_c;
  uint64_t lim_r;           // records the row tables (and the row kernel's grid) hold; tot[3] beyond it -> the kernels behind the row scan do nothing
};
__device__ __forceinline__ bool rows_over(const uint64_t *tot, uint64_t lim_r) { return tot && tot[3] > lim_r; }
#define RR_HI 0x0fffffffu        // r_rec.w bits 0..27: HI
#define RR_PRIMARY (1u << 28)
#define RR_PAIRED (1u << 29)
#define RR_SAME (1u << 30)
#define RR_FIRST (1u << 31)

// ---------------------------------------------------------------------------
// Direct rows (direct_kernels.inc): presets without the similarity filter and without -S.  Pairing is decided on the
// count pass's survivor sets BEFORE anything is emitted (src/mates.cpp:150-261 needs only the two mates' transcript-id
// sets), one fused scan places every kept match, and the emit kernels write the packed rows themselves: no match table,
// no per-record r_rec, no k_pair<true> / k_rows.
//   k_name_seed     per read name: the first mt19937_64 output of its seed (the primary tie-break's only heavy part; it
//                   needs nothing but the name, so it runs beside the count pass)
//   k_pair_mask     per alignment: filtered survivor mask, records per leader
//   k_big_collect / k_pair_big   the same for alignments with > 64 candidate rows (tid lists in a side arena)
//   k_scan5_*       kept matches -> class-list position, CIGAR arena base, row offsets (one pass, one host wait)
//   k_group_desc    per read name: NH, first HI of every alignment, the primary record (all scores tie: core.cpp:283-303)
//   k_expand_rows   emit work list + the 16-byte emit descriptor of every alignment
//   k_emit_rows<CLS>, k_big_emit   one lane per kept match -> packed row at row_off[leader] + stride * rank (+ is_mate)
// ---------------------------------------------------------------------------
enum : uint32_t { PF_PAIRED = 1, PF_SAME = 2, PF_MATE = 4, PF_BIG = 8 };   // pflag / low byte of the descriptor's .y
struct DirectArgs {
  int64_t n_aln, n_groups;
  const uint32_t *group_off;     // [n_groups + 1]
  const uint32_t *aln_group;     // [n_aln]
  const int32_t *mate_idx;       // [n_aln] mutual mate pointers (br_batch_prepare / k_mates) or -1
  const uint32_t *n_matches;     // [n_aln] count pass: survivors per alignment
  const uint64_t *mask;          // [n_aln] count pass: survivor bit per candidate row (<= 64 rows)
  const uint4 *ranges;           // [n_aln] count pass: candidate row ranges
  const uint32_t *fast_flag;     // [n_aln] k_segment: class bit + CIGAR slot capacity
  const uint32_t *s_tid;         // index: tid of every slab row
  const uint32_t *big_list; const uint32_t *n_big;   // alignments with > 64 candidate rows and >= 1 survivor
  // k_pair_mask / k_pair_big
  uint2 *fm;                     // [n_aln] the filtered survivor mask (the kept ones); big alignments: offset of their list in the side arena
  uint32_t *n_kept;              // [n_aln] kept survivors = emitted records of the alignment
  uint32_t *n_rows;              // [n_aln] records of a leader (its own + its mate's), else 0
  uint8_t *pflag;                // [n_aln] PF_*
  uint2 *side;                   // big alignments: {tid, rank among the kept | ~0u: dropped} per survivor, candidate order
  uint64_t side_cap;
  uint32_t *pm_list, *pm_n;      // k_pair_mask: windows (index / 64) whose tid lists exceed its staging area, and their count
  unsigned long long *side_used; // [0] entries handed out, [1] set when the arena ran out (the host grows it and repeats)
  // fused scan
  uint32_t *cls_pos;             // [n_aln] first entry of the alignment in the class-partitioned emit work list
  uint64_t *cig_base;            // [n_aln + 1]
  uint64_t *row_off;             // [n_aln + 1] first record of leader a (br_device_rows.row_off)
  // primary choice
  const uint32_t *name_off; const uint8_t *names;   // null: no primary flags
  uint64_t *rnd0;                // [n_groups] k_name_seed
  // the emit descriptor of an alignment, as two dense 8-byte arrays (their writers run on different streams):
  uint2 *gd;                     // [n_aln] k_group_desc: {PF_* | (primary rank + 1) << 8, NH}
  uint2 *dpos;                   // [n_aln] k_expand_rows: {first record, position of its entries in the emit work list}
  uint32_t *hi0;                 // [n_aln] HI of the alignment's first record
  uint64_t *counters;            // [4] total_complete, total_unique, dropped_reads, a field overflowed its packed width; + k_group_desc's slots (GD_*)
  const uint64_t *tot;           // scan totals on the device: [0] kept, [1] arena words, [2] kept of the simple class, [3] records, [4] survivors
  // emit
  uint32_t *m_aln;               // emit work list
  uint64_t m_aln_cap;            // its room in entries when k_expand_rows is launched ahead of the host's look at the totals (0: the list was sized for them)
  uint4 *r_a; uint2 *r_c;        // packed rows
  uint4 *r_x;                    // detail column {input, junc_hits, aligned_len, HI} or null
};
// k_group_desc's partial sums of total_unique / dropped_reads: GD_SLOTS pairs of u64, a cache line apart, behind the four counters
#define GD_SLOTS 16
#define GD_SLOT0 16
#define GD_SLOT_STRIDE 16
#define GD_COUNTER_WORDS (GD_SLOT0 + GD_SLOTS * GD_SLOT_STRIDE)
void launch_name_seed(hipStream_t st, const DirectArgs &D);
void launch_pair_mask(hipStream_t st, const DirectArgs &D, int wide_blocks);
void launch_big_collect(hipStream_t st, const ProjectArgs &A, const DirectArgs &D, int n_blocks);
void launch_pair_big(hipStream_t st, const DirectArgs &D, int n_blocks);
void launch_scan5(hipStream_t st, const DirectArgs &D, uint64_t *tile_sums, uint64_t *total_out5);
void launch_group_desc(hipStream_t st, const DirectArgs &D);
void launch_expand_rows(hipStream_t st, const DirectArgs &D);
void launch_emit_rows(hipStream_t st, const ProjectArgs &A, const DirectArgs &D, int64_t n_kept, int64_t n_simple, int part);
void launch_big_emit(hipStream_t st, const ProjectArgs &A, const DirectArgs &D, int n_blocks);

// dense copy of the long (> 2 op) rewritten CIGARs for the host (the device rows point into the sparse arena)
struct PoolArgs {
  int64_t n_rows;
  const uint4 *r_a; const uint2 *r_c;
  const uint32_t *arena;
  uint32_t *sizes;          // [n_rows] ops of a long CIGAR, else 0
  const uint64_t *off;      // [n_rows + 1] scan of sizes
  uint2 *c_out;             // [n_rows] r_c with the arena offsets replaced by pool offsets
  uint32_t *pool;
};

// wide (one array per field) view of the packed rows: br_device_rows
struct WideArgs {
  int64_t n_rows, n_aln;
  int32_t long_reads;
  const uint4 *r_a; const uint2 *r_c; const uint4 *r_x;
  const double *r_sim; const int32_t *r_clip;  // null: all zero
  const uint32_t *pool;
  const uint32_t *aln_group;
  const int32_t *l_qseq;
  int32_t *w_input;
  uint32_t *w_nh, *w_hi, *w_mapq, *w_group;
  int32_t *w_mate_tid, *w_mate_pos, *w_isize;
  uint32_t *w_tid, *w_pos, *w_ncig;
  int8_t *w_strand;
  double *w_sim;
  int32_t *w_clip, *w_junc, *w_refc;
  uint8_t *w_paired, *w_same, *w_first, *w_primary;
  const uint64_t *w_cigoff;  // [n_rows + 1] scan of w_ncig
  uint32_t *w_cigar;         // dense rewritten CIGARs
};

struct StatsArgs {
  DevIndex ix;
  int64_t n_aln;
  const int32_t *ref_id;
  const uint32_t *cigar_off;
  const uint2 *seg;
  const uint4 *head, *head2;
  uint64_t *out;  // [8]
};
void launch_stats(hipStream_t st, const StatsArgs &T, const uint2 *m_p, const uint32_t *match_off, const uint32_t *n_matches);

// BAM re-encoding (bam_kernels.hip)
struct BamAux {
  uint32_t off[4], len[4];  // removal intervals inside the aux area, sorted by offset (~0u: none)
  int32_t as_val;           // input AS value (long reads)
  uint32_t aux_start, aux_len;
  // what the encoder needs of the fixed fields, so that it starts from this one record instead of a dependent read of
  // the input: l_read_name | mapq << 8 | bin << 16, n_cigar_op | flag << 16, l_seq (record bytes 8..19); and whether
  // qualities are present (first QUAL byte != 0xff, src/bam.cpp:680)
  uint32_t c_a, c_b, c_c, qual_present;
  uint32_t cg_len;          // bytes of the CG:B,I tag that holds the record's real CIGAR (0: none), see bam_cg.h
};
#define BLOB_END_SLOTS 64
#define BLOB_END_STRIDE 16
struct BamArgs {
  int64_t n_aln, n_rows;
  int32_t long_reads;
  const uint8_t *blob;       // original records, BAM layout from refID on
  const uint64_t *rec_off;   // [n_aln + 1] (or [n_aln] with rec_len)
  const uint32_t *rec_len;   // [n_aln] or null: rec_off[i + 1] - rec_off[i]
  int8_t *xs_out, *ts_out;   // [n_aln] or null: tag_char1("XS") / tag_char1("ts") of every record
  BamAux *aux;               // [n_aln]
  uint32_t *base_len;        // [n_aln] bytes of an output row of this record without its CIGAR (k_bam_scan -> k_bam_size)
  const uint4 *r_a; const uint2 *r_c; const uint4 *r_rec;  // packed rows + the records behind them (PairArgs)
  int32_t rec_x;             // r_rec holds the direct path's detail rows (br_row_x) instead of k_pair_emit's records
  const double *r_sim; const int32_t *r_clip;              // null: all zero
  const uint32_t *pool;
  const int32_t *l_qseq;
  uint32_t *out_len;         // [n_rows]
  const uint64_t *out_off;   // [n_rows + 1]
  uint8_t *out;
  uint64_t *too_long;        // set when a spilled CIGAR's reference length does not fit the placeholder's 28 bits (bam_write1 fails there)
  uint64_t *blob_end;        // k_bam_scan: end (byte offset in blob) of the record that ends last, as the maximum over BLOB_END_SLOTS words a
                             // cache line apart (one atomicMax per block of k_bam_scan: 80 000 of them on one address queue up in one L2 channel);
                             // k_bam_tasks never loads past it
};
void launch_bam_scan(hipStream_t st, const BamArgs &B);

// BGZF on the device (codec_kernels.hip)
#define DEFLATE_PAYLOAD 57344u   // uncompressed bytes per BGZF block: worst case (all 9-bit literals) still fits 64 KiB
#define DEFLATE_SLOT 65536u
#define DEFLATE_CRC_CHUNK 896u   // 64 lane-chunks per full block
struct DeflateArgs {
  const uint8_t *src;
  uint64_t n_bytes, n_blocks;
  uint8_t *slots;              // n_blocks * DEFLATE_SLOT
  uint32_t *sizes;             // [n_blocks] BGZF block sizes
  const uint32_t *crc_tab;     // [256] reflected CRC-32 table
  const uint32_t *crc_shift;   // [4][256] "append DEFLATE_CRC_CHUNK zero bytes" operator
  uint32_t *tokens;            // dynamic Huffman: DEFLATE_PAYLOAD words per resident wave
  uint32_t *queue;             // dynamic Huffman: next block to take (zeroed before the launch)
};
// dynamic_waves: 0 = fixed Huffman (one wave per block); > 0 = dynamic Huffman with that many persistent waves
void launch_deflate(hipStream_t st, const DeflateArgs &A, int dynamic_waves);
void launch_bgzf_compact(hipStream_t st, const DeflateArgs &A, const uint64_t *off, uint8_t *dense);

// BGZF inflate on the device (inflate_kernels.hip): one wave per BGZF block
#define INFLATE_CRC_CHUNK 1024u   // 64 lane-chunks cover a 64 KiB block
struct InflateBlock { uint64_t src_off, dst_off; uint32_t clen, ulen, crc, pad; };   // deflate payload in src; where its bytes go
struct InflateArgs {
  const uint8_t *src; uint64_t n_src;
  uint8_t *dst;
  const InflateBlock *blocks; uint64_t n_blocks;
  uint32_t *queue;             // next block to take (zeroed before the launch)
  uint32_t *n_bad;             // blocks that did not inflate to their ISIZE bytes with their CRC32
  const uint32_t *crc_tab4;    // [4][256] slice-by-4 tables of the reflected CRC-32
  const uint32_t *crc_shift;   // [4][256] "append INFLATE_CRC_CHUNK zero bytes" operator
};
void launch_inflate(hipStream_t st, const InflateArgs &A, int n_waves);

// record boundaries of an inflated BAM stream (split_kernels.hip)
#define SPLIT_SEG_BYTES 32768u
struct SplitArgs {
  const uint8_t *data; uint64_t n_bytes;   // starts at a record
  int32_t n_ref; uint32_t seg_bytes; int64_t n_seg;
  uint64_t *entry, *entry_next, *exit_;    // [n_seg] first record start in the segment (~0: none), what the check makes of it; where its walk ends
  uint32_t *n_map, *n_unm, *ended;         // [n_seg] mapped / unmapped records that start in the segment; 1: the walk met the end of the data, 2: a malformed record
  uint32_t *flags;                         // [0] a malformed record, [1] entries the check replaced
  const uint64_t *map_pre;                 // [n_seg] exclusive scan of n_map
  uint64_t *rec_off; uint32_t *rec_len;    // mapped records: offset of the refID field, block_size
  uint64_t *totals;                        // [0] unmapped records, [1] bytes consumed
};
void launch_split_guess(hipStream_t st, const SplitArgs &S);
void launch_split_spoil(hipStream_t st, const SplitArgs &S, int k);
void launch_split_walk(hipStream_t st, const SplitArgs &S, const uint32_t *redo);
void launch_split_check(hipStream_t st, const SplitArgs &S, uint32_t *redo);
void launch_split_emit(hipStream_t st, const SplitArgs &S);
void launch_split_totals(hipStream_t st, const SplitArgs &S);
void launch_last_group(hipStream_t st, const uint8_t *data, const uint64_t *rec_off, int64_t n, unsigned long long *out);
void launch_unmapped_before(hipStream_t st, const SplitArgs &S, uint64_t limit, unsigned long long *out);
// piece-wise reading: first plausible record start; the piece's two cuts, their byte offsets and the unmapped records between (see split_kernels.hip)
void launch_first_record(hipStream_t st, const SplitArgs &S, uint64_t limit, unsigned long long *out);
void launch_piece_cut(hipStream_t st, const SplitArgs &S, const uint64_t *rec_off, int64_t n, uint64_t bound, uint64_t used, int guess, unsigned long long *cut);

// BAM records -> input tables (parse_kernels.hip)
struct ParseArgs {
  int64_t n, n_groups;
  const uint8_t *blob;
  const uint64_t *rec_off;
  const uint32_t *rec_len;     // or null
  const int32_t *ref_map;      // BAM refID -> annotation reference index
  int32_t n_ref_map;
  int32_t *ref_id, *ref_start, *l_qseq;
  uint16_t *flags;
  uint32_t *ncig, *name_len, *isnew;
  uint32_t *maxima;            // [2] longest CIGAR, longest soft clip
  const uint32_t *group_pre;   // [n + 1] exclusive scan of isnew
  uint32_t *group_off;         // [n_groups + 1]
  const uint32_t *cigar_off, *name_off;  // [n + 1]
  uint32_t *cigar;
  uint8_t *names;
  int32_t *mate_idx;
  uint32_t *n_big_groups, *big_groups;
  int32_t *seq_src;
  uint32_t *seq_len;
  const uint32_t *seq_off;
  uint8_t *seqs;
  // flat (br_batch) input instead of records (blob == null): the pairing inputs come from these arrays
  const int32_t *mate_ref_id, *mate_start;
};
// br_batch arrays as uploaded -> the input contract on the device (the same work br_batch_prepare /
// br_batch_seq_source do on the host): 32-bit offsets, "starts a new read name" flags, batch maxima
struct SoaArgs {
  int64_t n;
  const uint64_t *cigar_off64, *name_off64, *seq_off64;  // seq_off64 may be null
  uint32_t *cigar_off, *name_off, *seq_off;
  const uint8_t *names;
  const uint32_t *cigar;
  uint32_t *isnew;
  uint32_t *maxima;  // [2] longest CIGAR, longest leading / trailing soft clip
};
void launch_soa_fields(hipStream_t st, const SoaArgs &S);
void launch_rec_fields(hipStream_t st, const ParseArgs &P);
void launch_group_off(hipStream_t st, const ParseArgs &P);
void launch_rec_copy(hipStream_t st, const ParseArgs &P);
void launch_mates(hipStream_t st, const ParseArgs &P);
void launch_seq_src(hipStream_t st, const ParseArgs &P);
void launch_seq_ascii(hipStream_t st, const ParseArgs &P);
void launch_bam_size(hipStream_t st, const BamArgs &B);
void launch_bam_encode(hipStream_t st, const BamArgs &B, int lanes);

// chores k_segment can take along for a small batch (null / 0: none): read-name group labels, zeroing of device counters
struct SegExtra {
  const uint32_t *group_off; uint32_t *aln_group; int64_t n_groups;
  uint64_t *zero_a; int n_zero_a; uint64_t *zero_b; int n_zero_b;
};
void launch_segment(hipStream_t st, int64_t n_aln, const int32_t *ref_id, const int32_t *ref_start,
                    const uint16_t *flags, const int8_t *xs, const int8_t *ts, const uint32_t *cigar_off,
                    const uint32_t *cigar, const DevCfg &cfg, uint32_t n_refs, uint2 *seg, AlnMeta *meta,
                    uint4 *head, uint4 *head2, uint32_t *fast_flag, const SegExtra *extra = nullptr);
// part: see launch_project_g (0 = everything)
void launch_project(hipStream_t st, const ProjectArgs &A, bool emit, int group_lanes, int n_blocks, int part = 0);
void launch_project_fa(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int mode, int n_blocks);
void launch_fa_fill(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int64_t n_prob);   // coded sequences of every problem
void launch_ksw(hipStream_t st, const KswArgs &K, int n_blocks);
void launch_ksw_bin(hipStream_t st, const KswFastArgs &A);
void launch_ksw_dp(hipStream_t st, const KswFastArgs &A, int bin);
void launch_ksw_trace(hipStream_t st, const KswFastArgs &A, int bin);   // bin < 0: every problem of the piece
uint32_t ksw_dp_resident_groups(int bin, int n_cu);
size_t ksw_prob_bytes();
size_t ksw_res_bytes();
// single pass: part 1 = the main kernel (alignments that need the exon walk go to walk_list), 2 = the listed ones
void launch_project1(hipStream_t st, const ProjectArgs &A, int group_lanes, int n_blocks, int part);
void launch_emit_wl(hipStream_t st, const ProjectArgs &A, int64_t n_entries);
void launch_expand(hipStream_t st, const ProjectArgs &A);
// n_simple: length of the work list's simple-class prefix (k_scan3's third total); part 0: one launch over
// everything, 1: the simple prefix, 2: the rest
void launch_emit_dense(hipStream_t st, const ProjectArgs &A, int64_t n_matches, int64_t n_simple, int part);
void launch_emit_dense_fa(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int64_t n_matches);
int64_t scan_tiles_for(int64_t n);
// mode 0: src32 as is; 1: n_matches * CIGAR slot capacity; 2: src32 as is (alias of 0);
// 3: n_matches * CIGAR slot capacity with the per-alignment ideal_cap[] of the -S path
void launch_scan(hipStream_t st, const ScanArgs &S, int mode, void *out, bool out64, uint64_t *total_out);
void launch_group_ids(hipStream_t st, int64_t n_groups, const uint32_t *group_off, uint32_t *aln_group);
bool launch_scan3(hipStream_t st, ScanArgs S, uint32_t *match_off, uint64_t *cig_base, uint32_t *fast_pre,
                  uint64_t *total_out3, const ProjectArgs *expand = nullptr);
void launch_pair(hipStream_t st, const PairArgs &P, bool emit);  // count pass: records per leader alignment; emit pass: r_rec
// primary record per read name (RR_PRIMARY) + the per-group counters; names may be null (no primary flags)
void launch_primary(hipStream_t st, const PairArgs &P, const uint32_t *name_off, const uint8_t *names, bool has_scores);
void launch_rows(hipStream_t st, const PairArgs &P, bool aux);  // r_rec + match table -> packed rows (r_a, r_c, aux columns)
void launch_rows_detail(hipStream_t st, const PairArgs &P);    // on request: r_x = {input, junc_hits, aligned_len, HI}
void launch_pool_sizes(hipStream_t st, const PoolArgs &Q);
void launch_pool_copy(hipStream_t st, const PoolArgs &Q, bool long_cigars);
void launch_wide_fields(hipStream_t st, const WideArgs &W);
void launch_wide_cigars(hipStream_t st, const WideArgs &W, int64_t n_words);

}  // namespace br
