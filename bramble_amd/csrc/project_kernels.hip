// Hand-written HIP kernels of the projection hot path for gfx950 (CDNA4, wave64).
//
// Integer interval work: no MFMA.  The design is candidate-centric instead of
// query-centric: the reference re-queries its interval tree for every read exon
// and intersects per-tid hash maps (src/evaluate.cpp:184-282); here one lane owns
// one candidate (transcript, first exon) found by a coalesced search of the
// start-sorted exon slab, and walks THAT transcript's own exon table against the
// read's exon list.  Equivalence with the reference's "query + intersect"
// formulation is argued in DESIGN.md and enforced by tests against oracle/.
//
// Kernels:
//   k_segment     a1,a2,a6 of SURVEY.md 8a: packed CIGAR -> read exons, clips, strands
//   k_project<G,EMIT>  a4,a5,a7,a8,a11-a15: candidates, exon-chain walk, ideal
//                 CIGAR, similarity, CIGAR merge.  EMIT=false counts matches,
//                 EMIT=true writes them at scanned offsets (tid-sorted per read).
//   k_pair<EMIT>  a16,a17: mate-pair transcript-set intersection, NH/HI/MAPQ
//   k_gather      packs the rewritten CIGARs of the emitted rows densely
//   k_scan_*      exclusive scans (u32 / u64)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_types.h"
#include "kernels.h"
#include "primary_pick.h"

namespace br {

#define CIG_OP(c) ((c) & 0xfu)
#define CIG_LEN(c) ((c) >> 4)
#define CIG_GEN(l, o) (((l) << 4) | (o))

// ---------------------------------------------------------------------------
// k_segment: one lane per alignment.
//   gclib/GSam.cpp:197-291 (setupCoordinates) + src/bramble.cpp:246-255 (end++)
//   src/bramble.cpp:213-244 (get_strand) + gclib/GSam.cpp:338-349 (spliceStrand)
//   src/evaluate.cpp:58-67 (strands to check), :69-109 (get_clips)
// Exons of alignment a go to seg[cigar_off[a] + a ...] (at most n_cigar, or 1
// when the CIGAR is empty).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_segment(int64_t n_aln, const int32_t *__restrict__ ref_id,
                                                 const int32_t *__restrict__ ref_start,
                                                 const uint16_t *__restrict__ flags,
                                                 const int8_t *__restrict__ xs, const int8_t *__restrict__ ts,
                                                 const uint32_t *__restrict__ cigar_off,
                                                 const uint32_t *__restrict__ cigar, DevCfg cfg,
                                                 uint32_t n_refs, uint2 *__restrict__ seg,
                                                 AlnMeta *__restrict__ meta, uint4 *__restrict__ head,
                                                 uint4 *__restrict__ head2, uint32_t *__restrict__ fast_flag, SegExtra X) {
  int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // small batches: two chores that would otherwise be launches of their own ride along -- the alignments' read-name group
  // labels (k_group_ids) and the zeroing of the batch's few device counters
  if (X.aln_group && a < X.n_groups) for (uint32_t i = X.group_off[a]; i < X.group_off[a + 1]; i++) X.aln_group[i] = (uint32_t)a;
  if (a == 0) { for (int k = 0; k < X.n_zero_a; k++) X.zero_a[k] = 0; for (int k = 0; k < X.n_zero_b; k++) X.zero_b[k] = 0; }
  if (a >= n_aln) return;
  uint32_t c0 = cigar_off[a], c1 = cigar_off[a + 1];
  uint32_t n_cigar = c1 - c0;
  const uint32_t *cg = cigar + c0;
  uint2 *out = seg + (size_t)c0 + (size_t)a;
  int32_t pos0 = ref_start[a] - 1;
  int l = 0, exstart = pos0;
  bool exon_started = false, intron = false, ins = false;
  uint32_t n = 0;
  uint32_t ex_start = 0, ex_end = 0;  // GSeg exon (zero-initialised)
  uint2 e0 = make_uint2(0, 0), e1 = e0, e2 = e0;   // the first three read exons stay in registers for head / head2 (reading them
                                                   // back from seg[] would wait for the stores above and then for the loads)
  // the first four CIGAR words with one load (nearly every short read has no more): the loop below then waits once, not
  // once per word.  Not for the last words of the array: the load would read past it.
  uint32_t w4[4] = {0, 0, 0, 0};
  const bool pre4 = (uint64_t)c0 + 4u <= (uint64_t)cigar_off[n_aln];
  if (pre4) { struct __attribute__((packed, aligned(4))) Q4 { uint32_t a, b, c, d; }; const Q4 q = *(const Q4 *)cg; w4[0] = q.a; w4[1] = q.b; w4[2] = q.c; w4[3] = q.d; }
  for (uint32_t i = 0; i < n_cigar; ++i) {
    uint32_t w = (pre4 && i < 4u) ? (i == 0 ? w4[0] : i == 1 ? w4[1] : i == 2 ? w4[2] : w4[3]) : cg[i], op = CIG_OP(w), len = CIG_LEN(w);
    switch (op) {
      case OP_EQ: case OP_X: case OP_M:
        exon_started = true; l += (int)len; intron = false; ins = false; break;
      case OP_D: l += (int)len; ins = false; break;
      case OP_I: ins = true; break;
      case OP_N:
        if (!exon_started) break;
        if (!ins || !intron) {
          ex_end = (uint32_t)(pos0 + l); ex_start = (uint32_t)(exstart + 1);
          const uint2 v = make_uint2(ex_start, ex_end + 1);
          if (n == 0) e0 = v; else if (n == 1) e1 = v; else if (n == 2) e2 = v;
          else { if (n == 3) { out[0] = e0; out[1] = e1; out[2] = e2; } out[n] = v; }   // seg[] is read from exon 3 on (ReadCtx::exon): reads of up to three exons never write it
          n++;
        }
        l += (int)len; exstart = pos0 + l; intron = true;
        break;
      case OP_S: case OP_H: ins = false; break;
      default: break;
    }
  }
  if (!intron) {
    ex_start = (uint32_t)(exstart + 1); ex_end = (uint32_t)(pos0 + l);
    const uint2 v = make_uint2(ex_start, ex_end + 1);
    if (n == 0) e0 = v; else if (n == 1) e1 = v; else if (n == 2) e2 = v;
    else { if (n == 3) { out[0] = e0; out[1] = e1; out[2] = e2; } out[n] = v; }
    n++;
  }
  if (ex_end == 0) n = 0;  // the reference aborts here (GSam.cpp:290); we project nothing
  int32_t rid = ref_id[a];
  if (rid < 0 || (uint32_t)rid >= n_refs) n = 0;  // no tree for this refid (src/g2t.cpp:278-287)

  // strand of the read, then the strands to try
  uint32_t smode = 3;
  uint16_t f = flags[a];
  if (!cfg.long_reads) {
    char c = (char)xs[a];
    if (c == 0) {
      char m = (char)ts[a];
      if (m == '+' || m == '-') c = (f & 0x10) ? ((m == '+') ? '-' : '+') : m;
    }
    char strand = (c == '+' || c == '-') ? c : '.';
    if (strand == '.' && (cfg.fr || cfg.rf)) {
      bool is_rev = f & 0x10;
      bool cond = (cfg.rf && is_rev) || (cfg.fr && !is_rev);
      if (f & 0x1) {
        int order = (f & 0x40) ? 1 : ((f & 0x80) ? 2 : 0);
        strand = (order == 1) ? (cond ? '-' : '+') : (cond ? '+' : '-');
      } else {
        strand = cond ? '-' : '+';
      }
    }
    smode = (strand == '+') ? 1u : (strand == '-') ? 2u : 3u;
  }
  uint32_t lclip = 0, rclip = 0;
  if (cfg.long_reads) {
    if (n_cigar == 0) { n = 0; }  // "alignment is missing CIGAR" -> failure (evaluate.cpp:76-81)
    else {
      uint32_t w0 = cg[0];
      if (CIG_OP(w0) == OP_H) { if (n_cigar > 1 && CIG_OP(cg[1]) == OP_S) lclip = CIG_LEN(cg[1]); }
      else if (CIG_OP(w0) == OP_S) lclip = CIG_LEN(w0);
      uint32_t wl = cg[n_cigar - 1];
      if (CIG_OP(wl) == OP_H) { if (n_cigar >= 2 && CIG_OP(cg[n_cigar - 2]) == OP_S) rclip = CIG_LEN(cg[n_cigar - 2]); }
      else if (CIG_OP(wl) == OP_S) rclip = CIG_LEN(wl);
    }
  }
  // clip lengths, for the -S rescue kernels only (long reads): short-read presets do not write the 16 bytes
  if (cfg.long_reads) { AlnMeta m; m.n_seg = n; m.smode = smode; m.n_left_clip = lclip; m.n_right_clip = rclip; meta[a] = m; }
  // everything k_project needs for read exon 0 in one 16-byte record
  uint2 q0 = n ? e0 : make_uint2(0, 0);
  head[a] = make_uint4(q0.x, q0.y, n, n ? (((uint32_t)rid << 2) | smode) : 0u);
  // read exons 1 and 2: only a spliced read has them (the readers load the record unconditionally and use it from two
  // exons on: for the others the 16 bytes are not written)
  if (n > 1) { const uint2 q2 = n > 2 ? e2 : make_uint2(0, 0); head2[a] = make_uint4(e1.x, e1.y, q2.x, q2.y); }
  // "simple" alignments: one read exon from a single M op, short-read presets.  They
  // are processed first (k_perm) so that whole waves take the short code paths.
  uint32_t fast = (n == 1 && n_cigar == 1 && CIG_OP(cg[0]) == OP_M && !cfg.filter_by_similarity && !cfg.long_reads) ? 1u : 0u;
  // bit 31: simple; bits 0..30: CIGAR slot capacity of one match = n_real + 2 * (4 * n_seg + 2)
  fast_flag[a] = (fast << 31) | ((n_cigar + 2u * (4u * n + 2u)) & 0x7fffffffu);
}

// ---------------------------------------------------------------------------
// per-lane building blocks
// ---------------------------------------------------------------------------
struct Hit { uint32_t pos, left_ins, right_ins, left_gap, right_gap; };

// Tolerance table of IntervalTree::findOverlapping (src/g2t.cpp:144-227),
// including the '-' strand right-overhang quirk at :204.
__device__ __forceinline__ bool classify(bool minus, int status, uint32_t qs, uint32_t qe, uint32_t s,
                                         uint32_t e, uint32_t pos_start, const DevCfg &c, Hit &h) {
  // Branch-free restatement: the kernels that call this are bound by instruction issue, the scalar unit above all
  // (divergent control flow), and the table's eight-way branching cost more than its arithmetic.  Both strands share
  // the left side; on the right '-' tests an overhang against max_junc_ins whatever the status (:204).  pos: '+'
  // counts from the exon start, '-' from its end.
  const bool junc_left = (status == ST_MIDDLE || status == ST_LAST);
  const bool junc_right = (status == ST_FIRST || status == ST_MIDDLE);
  const uint32_t lg = qs > s ? qs - s : 0u, li = s > qs ? s - qs : 0u;
  const uint32_t rg = e > qe ? e - qe : 0u, ri = qe > e ? qe - e : 0u;
  const uint32_t lim_li = junc_left ? c.max_junc_ins : c.max_clip;
  const uint32_t lim_ri = (junc_right || minus) ? c.max_junc_ins : c.max_clip;
  const uint32_t lim_lg = junc_left ? c.max_junc_gap : 0xffffffffu;
  const uint32_t lim_rg = junc_right ? c.max_junc_gap : 0xffffffffu;
  h.pos = pos_start + (minus ? rg : lg); h.left_ins = li; h.right_ins = ri; h.left_gap = lg; h.right_gap = rg;
  return (li <= lim_li) & (ri <= lim_ri) & (lg <= lim_lg) & (rg <= lim_rg);
}

// Ideal-CIGAR accumulator: Cigar::add_operation (include/evaluate.h:108-126).
// One pending op lives in registers; completed ops go to `buf` (may be null
// when only the accumulators are wanted).
struct IdealSink {
  uint32_t *buf; uint32_t n; uint32_t cur; bool has;
  __device__ __forceinline__ void init(uint32_t *b) { buf = b; n = 0; cur = 0; has = false; }
  __device__ __forceinline__ void add(uint32_t len, uint32_t op) {
    // one predicated store instead of three exits (the emit kernels are bound by instruction issue)
    const bool same = has && CIG_OP(cur) == op;
    const bool flush = has && !same;
    if (flush && buf) buf[n] = cur;
    n += flush ? 1u : 0u;
    cur = same ? cur + (len << 4) : CIG_GEN(len, op);
    has = true;
  }
  __device__ __forceinline__ uint32_t finish() { if (has) { if (buf) buf[n] = cur; n++; has = false; } return n; }
};

// ExonChainMatch accumulators (include/evaluate.h:168-181, create_match :658-673)
struct Acc {
  double cov, ops;
  int32_t ref_consumed, junc_hits, clip_score;
  uint32_t prev_op;
  uint32_t last_pos;  // transcript position of the last guide segment's hit (rcpos on '-')
  __device__ __forceinline__ void init() { cov = 0; ops = 0; ref_consumed = 0; junc_hits = 0; clip_score = 0; prev_op = OP_M; last_pos = 0; }
};

// build_cigar_match (src/evaluate.cpp:675-786)
__device__ __forceinline__ void build_match(Acc &m, IdealSink &sk, const Hit &h, int status, uint32_t qs,
                                            uint32_t qe, uint32_t gs, uint32_t ge, bool first_match,
                                            bool last_match, bool has_lc, bool has_rc) {
  if (h.left_ins > 0) {
    if (status == ST_FIRST || status == ST_ONLY) {
      if (!has_lc) { sk.add(h.left_ins, OP_S); m.ops += h.left_ins; m.prev_op = OP_S; }
    } else {  // MIDDLE / LAST (|| has_left_clip is then irrelevant)
      sk.add(h.left_ins, OP_I); m.ops += h.left_ins;
      if (m.prev_op == OP_D) m.cov += h.left_ins;
      else if (m.prev_op == OP_I) m.ops += (m.ops * 0.2);
      m.prev_op = OP_I;
    }
  } else if (h.left_gap > 0) {
    if (!first_match && (status == ST_MIDDLE || status == ST_LAST || has_lc)) {
      sk.add(h.left_gap, OP_D); m.ops += h.left_gap; m.ref_consumed += (int32_t)h.left_gap;
      if (m.prev_op == OP_I) m.cov += h.left_gap;
      else if (m.prev_op == OP_D) m.ops += (m.ops * 0.2);
      m.prev_op = OP_D;
    }
  } else {
    m.junc_hits++;
  }
  uint32_t os = qs > gs ? qs : gs, oe = qe < ge ? qe : ge;
  if (oe >= os) {
    uint32_t ml = oe - os;
    sk.add(ml, OP_M); m.ops += ml; m.cov += ml; m.ref_consumed += (int32_t)ml; m.prev_op = OP_M;
  }
  if (h.right_ins > 0) {
    if (status == ST_LAST || status == ST_ONLY) {
      if (!has_rc) { sk.add(h.right_ins, OP_S); m.ops += h.right_ins; m.prev_op = OP_S; }
    } else {
      sk.add(h.right_ins, OP_I); m.ops += h.right_ins;
      if (m.prev_op == OP_D) m.cov += h.right_ins;
      m.prev_op = OP_I;
    }
  } else if (h.right_gap > 0) {
    if (!last_match && (status == ST_FIRST || status == ST_MIDDLE || has_rc)) {
      sk.add(h.right_gap, OP_D); m.ops += h.right_gap; m.ref_consumed += (int32_t)h.right_gap;
      if (m.prev_op == OP_I) m.cov += h.right_gap;
      m.prev_op = OP_D;
    }
  } else {
    m.junc_hits++;
  }
}

// merge_ops (src/bam.cpp:22-111); 95 ('_') = drop.
__device__ __forceinline__ uint32_t merge_ops(uint32_t r, uint32_t i) {
  bool r_ms = (r == OP_M || r == OP_S);
  if (r_ms && i == OP_CLIP_OVR) return OP_S;
  if (r_ms && i == OP_MATCH_OVR) return OP_M;
  if (r_ms && i == OP_INS_OVR) return OP_I;
  if (r_ms && i == OP_DEL_OVR) return OP_D;
  if (r == OP_D && (i == OP_S || i == OP_CLIP_OVR)) return 95u;
  if (r == OP_D && i == OP_MATCH_OVR) return OP_D;
  if (r == OP_I && i == OP_CLIP_OVR) return OP_S;
  if (r == OP_I && i == OP_MATCH_OVR) return OP_I;
  if (i == OP_CLIP_OVR) return OP_S;
  if (i == OP_MATCH_OVR) return OP_M;
  if (i == OP_INS_OVR) return OP_I;
  if (i == OP_DEL_OVR) return OP_D;
  if (r == OP_P) return i;
  if (r == OP_H) return OP_H;
  if (r == OP_I && i == OP_S) return OP_S;
  if (i == OP_S || i == OP_D || i == OP_I) return i;
  if (r == OP_S || r == OP_D || r == OP_I) return r;
  if (i == OP_M || i == OP_EQ) return OP_M;
  if (i == OP_X) return OP_X;
  if (r == OP_M || r == OP_EQ) return OP_M;
  if (r == OP_X) return OP_X;
  return i;
}

// get_new_cigar + merge_cigars (src/bam.cpp:443-472, :113-315).  `real` is the
// alignment's packed CIGAR, `ideal` the lane's ideal CIGAR; the rewritten CIGAR
// goes to `out` (room for n_real + n_ideal words).  Returns its length.
// The alignment's packed CIGAR with its first four words in registers: merge_cigars' loop depends on the word it
// just read, so every word left in memory is a (cache-hit) round trip on the critical path; fetch() is issued as
// soon as the CIGAR's offset is known, together with the other loads of that stage.
struct RealCig {
  const uint32_t *p; uint32_t w0, w1, w2, w3;
  __device__ __forceinline__ void fetch(const uint32_t *q, uint32_t n) {  // n >= 1
    const uint32_t l = n - 1;
    p = q; w0 = q[0]; w1 = q[l < 1u ? l : 1u]; w2 = q[l < 2u ? l : 2u]; w3 = q[l < 3u ? l : 3u];
  }
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const {
    if (i >= 4u) return p[i];
    return i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : w3;
  }
};

// What one step of the main merge loop does for (real op, ideal op), src/bam.cpp:236-288: 0 skip the real N op,
// 1 real D under an ideal S / I (or override): both advance, nothing is written, 2 real I: written whole,
// 3 ideal D (or override): written whole, 4 the ops are merged over their common length.
__device__ __forceinline__ uint32_t merge_action(uint32_t r, uint32_t i) {
  if (r == OP_N) return 0;
  if (r == OP_D && (i == OP_S || i == OP_CLIP_OVR || i == OP_I || i == OP_INS_OVR)) return 1;
  if (r == OP_I) return 2;
  if (i == OP_D || i == OP_DEL_OVR) return 3;
  return 4;
}
// mops: (merge_action << 8 | merge_ops) as a 16 x 16 table in LDS (k_emit_dense fills it once per block: the kernel is
// bound by the scalar unit, and the 22-way chain of merge_ops and the case ladder of the loop are most of the merge's
// control flow), or null = evaluate the chains
__device__ uint32_t merge_cigars(const RealCig &real, uint32_t n_real,
                                 const uint32_t *ideal, uint32_t n_ideal, uint32_t *out, const uint16_t *mops = nullptr) {
#define MERGE_OPS(R, I) ((mops && (I) < 16u) ? (uint32_t)(mops[((R) << 4) | (I)] & 0xffu) : merge_ops((R), (I)))
  uint32_t front_h = 0, front_s = 0, ci = 0;
  if (n_real > 0 && CIG_OP(real[0]) == OP_H) { front_h = CIG_LEN(real[0]); ci++; }
  if (ci < n_real && CIG_OP(real[ci]) == OP_S) front_s = CIG_LEN(real[ci]);

  uint32_t ridx = 0, ri = 0, ii = 0, real_pos = 0, ideal_pos = 0;
  uint32_t last = 0;  // copy of out[ridx-1]
  bool saw_ins = false;
#define ADD_OP(OPV, LENV)                                                          \
  do {                                                                             \
    uint32_t _op = (OPV), _len = (LENV);                                           \
    if (_len != 0 && _op != 95u) {                                                 \
      if (ridx > 0 && CIG_OP(last) == (_op & 0xffu)) { last += (_len << 4); out[ridx - 1] = last; } \
      else { last = CIG_GEN(_len, _op); out[ridx++] = last; }                      \
      saw_ins |= ((_op & 0xffu) == OP_I);                                          \
    }                                                                              \
  } while (0)

  uint32_t clips = front_h;
  while (clips > 0 && ri < n_real) {
    uint32_t rw = real[ri];
    uint32_t avail = CIG_LEN(rw) - real_pos;
    uint32_t chunk = clips < avail ? clips : avail;
    ADD_OP(CIG_OP(rw), chunk);
    clips -= chunk; real_pos += chunk;
    if (real_pos >= CIG_LEN(rw)) { ri++; real_pos = 0; }
  }
  clips = front_s;
  while (clips > 0 && ri < n_real) {
    uint32_t rw = real[ri];
    uint32_t real_op = CIG_OP(rw);
    bool have_i = ii < n_ideal;
    uint32_t iw = have_i ? ideal[ii] : 0;
    uint32_t ideal_op = have_i ? CIG_OP(iw) : 0xffu;
    uint32_t real_rem = CIG_LEN(rw) - real_pos;
    uint32_t ideal_rem = have_i ? CIG_LEN(iw) - ideal_pos : 0xffffffffu;
    bool is_ovr = have_i && (ideal_op >= OP_MATCH_OVR && ideal_op <= OP_CLIP_OVR);
    if (is_ovr) {
      if (ideal_op == OP_DEL_OVR) {
        uint32_t chunk = ideal_rem;
        ADD_OP(MERGE_OPS(real_op, ideal_op), chunk);
        ideal_pos += chunk;
        if (ideal_pos >= CIG_LEN(iw)) { ii++; ideal_pos = 0; }
      } else {
        uint32_t chunk = clips;
        if (chunk > real_rem) chunk = real_rem;
        if (chunk > ideal_rem) chunk = ideal_rem;
        ADD_OP(MERGE_OPS(real_op, ideal_op), chunk);
        clips -= chunk; real_pos += chunk; ideal_pos += chunk;
        if (real_pos >= CIG_LEN(rw)) { ri++; real_pos = 0; }
        if (ideal_pos >= CIG_LEN(iw)) { ii++; ideal_pos = 0; }
      }
    } else {
      uint32_t chunk = clips;
      if (chunk > real_rem) chunk = real_rem;
      ADD_OP(MERGE_OPS(real_op, ideal_op), chunk);
      clips -= chunk; real_pos += chunk;
      if (real_pos >= CIG_LEN(rw)) { ri++; real_pos = 0; }
    }
  }
  while (ri < n_real || ii < n_ideal) {
    if (ri >= n_real) {
      uint32_t iw = ideal[ii];
      ADD_OP(CIG_OP(iw), CIG_LEN(iw) - ideal_pos);
      ii++; ideal_pos = 0;
      continue;
    }
    uint32_t rw = real[ri];
    if (ii >= n_ideal) {
      ADD_OP(CIG_OP(rw), CIG_LEN(rw) - real_pos);
      ri++; real_pos = 0;
      continue;
    }
    uint32_t iw = ideal[ii];
    uint32_t real_op = CIG_OP(rw), ideal_op = CIG_OP(iw);
    uint32_t real_rem = CIG_LEN(rw) - real_pos, ideal_rem = CIG_LEN(iw) - ideal_pos;
    // one body for the five cases: what is written and how far either side advances follow from the action code
    const uint32_t t = mops ? (uint32_t)mops[(real_op << 4) | ideal_op] : ((merge_action(real_op, ideal_op) << 8) | merge_ops(real_op, ideal_op));
    const uint32_t act = t >> 8;
    const uint32_t chunk = real_rem < ideal_rem ? real_rem : ideal_rem;
    const bool both = act == 1u || act == 4u;
    const uint32_t adv_r = (act == 0u || act == 2u) ? real_rem : both ? chunk : 0u;
    const uint32_t adv_i = act == 3u ? ideal_rem : both ? chunk : 0u;
    const uint32_t w_len = act == 2u ? real_rem : act == 3u ? ideal_rem : act == 4u ? chunk : 0u;
    const uint32_t w_op = act == 2u ? (uint32_t)OP_I : act == 3u ? (uint32_t)OP_D : (t & 0xffu);
    ADD_OP(w_op, w_len);
    real_pos += adv_r; ideal_pos += adv_i;
    if (real_pos >= CIG_LEN(rw)) { ri++; real_pos = 0; }
    if (ideal_pos >= CIG_LEN(iw)) { ii++; ideal_pos = 0; }
  }
#undef ADD_OP
#undef MERGE_OPS
  // "I between clips -> clip" fix-up (bam.cpp:292-300), then re-coalesce (:302-311)
  // Without an I op the fix-up is a no-op, and ADD_OP never leaves equal
  // neighbours, so re-coalescing only matters after a fix-up rewrote an op.
  bool changed = false;
  if (saw_ins) {
    for (uint32_t i = 1; i + 1 < ridx; i++) {
      uint32_t w = out[i];
      if (CIG_OP(w) != OP_I) continue;
      uint32_t prev = CIG_OP(out[i - 1]), next = CIG_OP(out[i + 1]);
      if ((prev == OP_S || prev == OP_H) && (next == OP_S || next == OP_H)) { out[i] = CIG_GEN(CIG_LEN(w), prev); changed = true; }
    }
  }
  if (!changed) return ridx;
  uint32_t nidx = 0, lastw = 0;
  for (uint32_t i = 0; i < ridx; i++) {
    uint32_t w = out[i];
    if (nidx > 0 && CIG_OP(lastw) == CIG_OP(w)) { lastw += (CIG_LEN(w) << 4); out[nidx - 1] = lastw; }
    else { lastw = w; out[nidx++] = w; }
  }
  return nidx;
}

// Lane-local "does ANY row of the slab pass the tolerance test for this read
// exon" = the return value of g2tTree::getGuideExons (src/g2t.cpp:334-344).
// Needed only for the INS_EXON rule (src/evaluate.cpp:250-273).
__device__ bool any_pass_global(const DevIndex &ix, uint32_t sb, uint32_t se, bool minus, int status,
                                uint32_t qs, uint32_t qe, const DevCfg &cfg) {
  uint32_t a = sb, b = se;  // hi: first row with start >= qe
  while (a < b) { uint32_t m = (a + b) >> 1; if (ix.s_start[m] < qe) a = m + 1; else b = m; }
  uint32_t hi = a;
  a = sb; b = hi;           // lo: first row whose running max end exceeds qs
  while (a < b) { uint32_t m = (a + b) >> 1; if (ix.s_pmax[m] <= qs) a = m + 1; else b = m; }
  for (uint32_t r = a; r < hi; r++) {
    uint32_t e = ix.s_row[2 * (size_t)r].y;
    if (e <= qs) continue;
    Hit h;
    if (classify(minus, status, qs, qe, ix.s_start[r], e, 0, cfg, h)) return true;
  }
  return false;
}

// Result of advancing one candidate over read exon j >= 1.
enum { STEP_DEAD = 0, STEP_HIT = 1, STEP_INS = 2 };

struct Walker {
  const uint4 *E;       // transcript's exon table (genomic order, sentinel-terminated)
  uint32_t g0;          // genomic exon index of E[0] within the transcript (always 0)
  bool minus;
};

// The exon after a candidate's slab row, from the row itself: e0 = {start, end, pos_start, -}, nx = {next start,
// next end} (nx.x == 0: not known, ~0u: none).  pos_start runs in transcript order (src/bramble.cpp:161-175): on '+'
// the next exon starts where this one ends, on '-' it lies before this one by its own length.
__device__ __forceinline__ uint4 next_row(uint4 e0, uint2 nx, bool minus) {
  return make_uint4(nx.x, nx.y, minus ? e0.z - (nx.y - nx.x) : e0.z + (e0.y - e0.x), 0u);
}

// One read exon against the candidate's transcript: the passing hits of this
// tid (src/evaluate.cpp:205-237), correct_for_gaps (:111-182) and the
// injectivity rules (:1023-1047) folded together.  i_last = table index of the
// last guide segment; on STEP_HIT i_hit/h/gap2 are set (gap2: a GAP_EXON segment
// for exon i_hit-1 precedes the match segment).
__device__ __forceinline__ int step_exon(const DevIndex &ix, const DevCfg &cfg, const uint4 *E, bool minus,
                                         uint32_t sb, uint32_t se, uint32_t i_last, uint4 e_last, bool have_e1, uint4 e1k,
                                         int status, uint32_t qs, uint32_t qe, uint32_t &i_hit, Hit &h, uint4 &ge, bool &gap2) {
  uint32_t cnt = 0;
  Hit hh; uint4 ee;
  // The row of the last guide segment is in registers, and so is the one after it when the walk still stands on
  // the candidate's slab row (have_e1: the 32-byte row carries the next exon's start and end).  A later row is
  // fetched only if the current one does not end the scan: exons of a transcript are disjoint and sorted, so
  // once one reaches qend every later one starts at or beyond it (sentinel start = ~0u).
  uint4 e = e_last, e1 = e1k;
  if (!have_e1) e1 = E[i_last + 1];
  for (uint32_t i = i_last;; i++) {
    if (e.x >= qe) break;             // start >= qend
    if (e.y > qs && classify(minus, status, qs, qe, e.x, e.y, e.z, cfg, hh)) {
      if (cnt == 0) { i_hit = i; h = hh; ee = e; }
      cnt++;
    }
    if (e.y >= qe) break;
    e = e1;
    if (e.x < qe && e.y < qe) e1 = E[i + 2];
  }
  gap2 = false;
  if (cnt == 0) {
    if (status == ST_MIDDLE && cfg.ignore_small_exons && (qe - qs <= cfg.max_error_exon)) {
      if (!any_pass_global(ix, sb, se, minus, status, qs, qe, cfg)) return STEP_INS;
    }
    return STEP_DEAD;
  }
  if (cnt >= 2) return STEP_DEAD;      // two guide exons for one query exon
  ge = ee;
  uint32_t gap = (i_hit - i_last) & 0xffu;  // uint8_t exon_id arithmetic
  if (!cfg.long_reads) { if (gap != 1) return STEP_DEAD; }
  else {
    if (gap > 2) return STEP_DEAD;
    if (gap == 2) {
      uint4 pe = E[i_hit - 1];         // genomic predecessor (i_hit >= 2 here)
      if (pe.y - pe.x > cfg.max_error_exon) return STEP_DEAD;
      gap2 = true;
    }
  }
  if (i_hit == i_last) return STEP_DEAD;  // same guide exon for two query exons
  return STEP_HIT;
}

// Everything one lane needs about its alignment.
struct ReadCtx {
  const uint2 *seg;      // read exons (global)
  uint32_t n_seg;
  const uint32_t *real;  // packed real CIGAR
  uint32_t n_real;
  uint4 q12;             // read exons 1 and 2 (from head2), so 2-3 exon reads never touch seg[]
  __device__ __forceinline__ uint2 exon(uint32_t j) const {
    if (j == 1) return make_uint2(q12.x, q12.y);
    if (j == 2) return make_uint2(q12.z, q12.w);
    return seg[j];
  }
};

struct CandOut {
  bool alive;
  uint32_t fwpos, rcpos, n_seg, n_gex;
  uint32_t i_lastm;                 // table index of the last matched guide exon
  uint32_t last_right_ins, last_right_gap;
};

// One rescued soft clip (LEFTC_EXON / RIGHTC_EXON segment, src/evaluate.cpp:397-448,548-598)
struct ClipSide {
  bool ok;
  const uint32_t *ops;   // override-op CIGAR of the segment
  uint32_t n_ops;
  int32_t score;         // ksw max
  uint32_t refc;         // reference bases the rescue alignment consumed
};
__device__ __forceinline__ ClipSide no_clip() { ClipSide c; c.ok = false; c.ops = nullptr; c.n_ops = 0; c.score = 0; c.refc = 0; return c; }

// Pass 1 (src/evaluate.cpp:1004-1065): survival, segment counts, fwpos/rcpos.
__device__ __forceinline__ CandOut walk_pass1(const DevIndex &ix, const DevCfg &cfg, const ReadCtx &rd,
                                              const uint4 *E, bool minus, uint32_t sb, uint32_t se,
                                              uint32_t i0, uint4 e0, uint2 nx, uint2 q0, const Hit &h0) {
  CandOut o; o.alive = true; o.fwpos = h0.pos; o.rcpos = h0.pos; o.n_seg = 1; o.n_gex = 1;
  o.i_lastm = i0; o.last_right_ins = h0.right_ins; o.last_right_gap = h0.right_gap;
  uint32_t i_last = i0;
  uint4 e_last = e0;
  uint2 pq = q0;
  for (uint32_t j = 1; j < rd.n_seg; j++) {
    uint2 q = rd.exon(j);
    int status = (j < rd.n_seg - 1) ? ST_MIDDLE : ST_LAST;
    uint32_t i_hit = 0; Hit h; uint4 ge; bool gap2;
    int r = step_exon(ix, cfg, E, minus, sb, se, i_last, e_last, nx.x != 0 && i_last == i0, next_row(e0, nx, minus), status, q.x, q.y, i_hit, h, ge, gap2);
    if (r == STEP_DEAD || (q.x == pq.x && q.y == pq.y)) { o.alive = false; break; }
    pq = q;
    if (r == STEP_INS) { o.n_seg++; continue; }
    if (gap2) { o.n_seg++; o.n_gex++; }
    o.n_seg++; o.n_gex++;
    i_last = i_hit; e_last = ge;
    if (minus) o.rcpos = h.pos;
    o.i_lastm = i_hit; o.last_right_ins = h.right_ins; o.last_right_gap = h.right_gap;
  }
  return o;
}

// Pass-1 bookkeeping of rescued clips (src/evaluate.cpp:1050-1064 with a LEFTC /
// RIGHTC segment at either end): segment counts, fwpos from the left dummy exon,
// rcpos from the last guide segment on '-'.
__device__ __forceinline__ void apply_clips(CandOut &o, bool minus, uint32_t pos_start0,
                                            uint32_t pos_start_last, const ClipSide &L, const ClipSide &R) {
  if (L.ok) {
    o.n_seg++; o.n_gex++;
    uint32_t dpos = pos_start0 - L.refc;
    o.fwpos = dpos;
    if (!minus) o.rcpos = dpos;  // on '-' every match segment is a later guide segment and overwrites rcpos
  }
  if (R.ok) {
    o.n_seg++; o.n_gex++;
    if (minus) o.rcpos = pos_start_last - R.refc;
  }
}

// build_cigar_clip (src/evaluate.cpp:824-841)
__device__ __forceinline__ void build_clip(Acc &m, IdealSink &sk, const ClipSide &c) {
  for (uint32_t i = 0; i < c.n_ops; i++) {
    uint32_t w = c.ops[i], op = CIG_OP(w), len = CIG_LEN(w);
    sk.add(len, op);
    if (op == OP_MATCH_OVR || op == OP_DEL_OVR) m.ref_consumed += (int32_t)len;
  }
  m.clip_score += c.score;
}

// Pass 2 (src/evaluate.cpp:1070-1106): ideal CIGAR + accumulators.  L / R: rescued
// clips (both !ok without -S): a clip segment opens / closes the chain, zeroes the
// neighbouring match segment's left_ins / right_ins (:489-490,648) and turns on
// td.has_left_clip / has_right_clip for build_cigar_match.
__device__ __forceinline__ void walk_pass2(const DevIndex &ix, const DevCfg &cfg, const ReadCtx &rd,
                                           const uint4 *E, bool minus, uint32_t sb, uint32_t se, uint32_t i0,
                                           uint2 q0, uint4 e0, uint2 nx, const Hit &h0_in,
                                           const CandOut &p1, Acc &acc, IdealSink &sk, const ClipSide &L,
                                           const ClipSide &R) {
  acc.init();
  uint32_t k = 0;
  if (L.ok) { build_clip(acc, sk, L); k++; }
  Hit h0 = h0_in;
  if (L.ok) h0.left_ins = 0;
  if (R.ok && rd.n_seg == 1) h0.right_ins = 0;
  int st0 = (rd.n_seg == 1) ? ST_ONLY : ST_FIRST;
  build_match(acc, sk, h0, st0, q0.x, q0.y, e0.x, e0.y, k == 0, k == p1.n_gex - 1, L.ok, R.ok);
  k++;
  uint32_t i_last = i0;
  uint4 e_last = e0;
  for (uint32_t j = 1; j < rd.n_seg; j++) {
    uint2 q = rd.exon(j);
    int status = (j < rd.n_seg - 1) ? ST_MIDDLE : ST_LAST;
    uint32_t i_hit = 0; Hit h; uint4 ge; bool gap2;
    int r = step_exon(ix, cfg, E, minus, sb, se, i_last, e_last, nx.x != 0 && i_last == i0, next_row(e0, nx, minus), status, q.x, q.y, i_hit, h, ge, gap2);
    if (r == STEP_INS) {  // build_cigar_ins (:788-806) + junc_hits (:1089-1091)
      uint32_t len = q.y - q.x;
      bool edge = (k == 0 || k == p1.n_seg - 1);
      if (edge) { sk.add(len, OP_S); acc.prev_op = OP_S; } else { sk.add(len, OP_I); acc.prev_op = OP_I; }
      acc.ops += len; acc.cov += len;
      acc.junc_hits -= edge ? 1 : 2;
      k++;
      continue;
    }
    if (gap2) {          // build_cigar_gap (:808-822) + junc_hits (:1093-1095)
      uint4 pe = E[i_hit - 1];
      uint32_t len = pe.y - pe.x;
      sk.add(len, OP_D); acc.prev_op = OP_D; acc.ops += len; acc.cov += len; acc.ref_consumed += (int32_t)len;
      acc.junc_hits -= 2;
      k++;
    }
    if (R.ok && j == rd.n_seg - 1) h.right_ins = 0;
    build_match(acc, sk, h, status, q.x, q.y, ge.x, ge.y, k == 0, k == p1.n_gex - 1, L.ok, R.ok);
    k++;
    i_last = i_hit; e_last = ge; acc.last_pos = h.pos;
  }
  if (R.ok) { build_clip(acc, sk, R); k++; }
  if (acc.junc_hits < 0) acc.junc_hits = 0;
}

// similarity filter (src/evaluate.cpp:843-865); returns keep, sets score
__device__ __forceinline__ bool similarity(const DevCfg &cfg, const Acc &acc, double &score) {
  score = 0.0;
  if (!cfg.filter_by_similarity) return true;
  double sim = (acc.ops > 0) ? (acc.cov / acc.ops) : 0.0;
  if (sim > cfg.thr) {
    double x = ((sim - cfg.thr) / (1.0 - cfg.thr));
    score = (x * x * (double)(acc.junc_hits + 1));
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------
// k_project<G, EMIT>: one G-lane group per alignment, grid-stride.
//
// Per alignment the group (1) takes the candidate row range of read exon 0 on each
// strand to try from the bucket tables (a superset of the overlapping rows; an exact
// G-wide count only when it exceeds the 64-bit survivor mask), (2) gives every
// candidate row to one lane, which tests it for overlap, classifies it and walks its
// own transcript, (3) counts (EMIT=false) or writes (EMIT=true) the survivors.  Small
// ideal / rewritten CIGARs are staged in LDS.  The count pass is a chain of three
// dependent stages (head -> table words -> rows); each stage's loads are unconditional
// and issued together.
// ---------------------------------------------------------------------------
#define SLAB_LDS 1025   // slab_off entries cached in LDS (<= 512 references)
#define WALK_LDS 2048   // deferred alignments a block of the main count pass collects before it appends them
#define BIG_LDS 256     // ... and alignments with more than 64 candidate rows
#define LDS_SLOT 25     // words of CIGAR scratch per lane (odd: conflict-free)
#define LDS_IDEAL 10    // ideal CIGAR words kept in LDS (n_seg <= 2)

// MODE (count pass of the presets without the similarity filter): 1 = main pass: an alignment with a candidate that
// needs the full exon walk (three or more read exons, or a two-exon read outside the shortcut) is put on walk_list
// instead, so that the walk's code and registers stay out of this kernel (it sits at the 64-VGPR ceiling and is
// bound by instruction issue); 2 = the listed alignments, with the walk.  0 = everything in one kernel.
template <int G, bool EMIT, bool SIMF, int MODE = 0>
__global__ void __launch_bounds__(256, EMIT ? 4 : 8) k_project(ProjectArgs A) {
  __shared__ uint32_t sh_slab[SLAB_LDS];
  __shared__ uint32_t sh_bin[EMIT ? 1 : SLAB_LDS];
  __shared__ uint32_t sh_cig[EMIT ? 256 * LDS_SLOT : 1];
  // MODE 1: the block's deferred alignments are collected in LDS and appended to walk_list with ONE global atomic at the
  // end (a same-address atomic per deferred alignment serialises: 13 ms for ~1.5 M of them)
  __shared__ uint32_t sh_wl[MODE == 1 ? WALK_LDS : 1];
  __shared__ uint32_t sh_wn, sh_wbase;
  // the count pass's alignments with more than 64 candidate rows (big_list) go the same way
  __shared__ uint32_t sh_bl[EMIT ? 1 : BIG_LDS];
  __shared__ uint32_t sh_bn, sh_bbase;
  if (MODE == 1 || !EMIT) { if (threadIdx.x == 0) { sh_wn = 0; sh_bn = 0; } __syncthreads(); }
  const int gl = threadIdx.x & (G - 1);
  const int gbase = (threadIdx.x & 63) & ~(G - 1);
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1);
  const int64_t groups_total = (int64_t)gridDim.x * (blockDim.x / G);
  const int64_t gid = (int64_t)blockIdx.x * (blockDim.x / G) + threadIdx.x / G;
  const DevIndex &ix = A.ix;
  const DevCfg &cfg = A.cfg;
  const uint32_t n_slab_off = 2 * ix.n_refs + 1;
  const bool slab_in_lds = n_slab_off <= SLAB_LDS;
  if (slab_in_lds) {
    for (uint32_t i = threadIdx.x; i < n_slab_off; i += blockDim.x) {
      sh_slab[i] = ix.slab_off[i];
      if (!EMIT && i <= ix.n_refs) sh_bin[i] = ix.bin_off[i];
    }
    __syncthreads();
  }

  // EMIT: only the alignments with more than 64 candidate rows come here (the
  // rest is written by k_emit_dense); the count pass listed them in big_list.
  if (EMIT && tot_over(A.tot, A.lim_m, A.lim_c)) return;
  const int64_t n_work = EMIT ? (int64_t)*A.n_big : MODE == 2 ? (int64_t)*A.n_walk : A.n_aln;
  for (int64_t w = gid; w < n_work; w += groups_total) {
    const int64_t a = EMIT ? (int64_t)A.big_list[w] : MODE == 2 ? (int64_t)A.walk_list[w] : w;
    uint4 hd = A.head[a];
    uint4 hd2 = A.head2[a];
    uint32_t n_seg = hd.z;
    if (EMIT && n_seg == 0) continue;
    uint2 q0 = make_uint2(hd.x, hd.y);
    // count pass: an alignment without read exons (head = {0, 0, 0, 0}) runs through with no strand to try instead
    // of leaving early, so that the head is ONE 16-byte load (an early exit on hd.z makes the compiler fetch that
    // word first and the rest a round trip later)
    uint32_t smode = hd.w & 3u, rid = hd.w >> 2;
    int st0 = (n_seg == 1) ? ST_ONLY : ST_FIRST;
    uint32_t sb[2], se[2];
    if (slab_in_lds) { sb[0] = sh_slab[2 * rid]; se[0] = sh_slab[2 * rid + 1]; se[1] = sh_slab[2 * rid + 2]; }
    else { sb[0] = ix.slab_off[2 * rid]; se[0] = ix.slab_off[2 * rid + 1]; se[1] = ix.slab_off[2 * rid + 2]; }
    sb[1] = se[0];

    uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
    if (EMIT) {
      if (A.n_matches[a] == 0) continue;
      uint4 rg = A.ranges[a];
      lo[0] = rg.x; hi[0] = rg.y; lo[1] = rg.z; hi[1] = rg.w;
    } else {
      // Candidate rows of read exon 0 on strand s, bounded by the bucket tables alone: from the first row whose
      // running max end exceeds the bin edge at or below qstart to the first row that starts at or beyond the bin
      // edge above qend.  A superset (x1.5 rows on GENCODE-shaped data) of the exact [lo, hi) a search inside the bins
      // would give -- every row is tested for overlap anyway -- that costs one dependent round trip less.
      uint32_t tw[4];
      // every reference owns >= 3 table entries: unconditional loads, both strands' words in each
      const uint32_t bo = slab_in_lds ? sh_bin[rid] : ix.bin_off[rid];
      const uint32_t nb = (slab_in_lds ? sh_bin[rid + 1] : ix.bin_off[rid + 1]) - bo - 1;
      uint32_t bh = q0.y >> ix.bin_shift, bl = q0.x >> ix.bin_shift;
      bh = bh < nb - 1 ? bh : nb - 1; bl = bl < nb - 1 ? bl : nb - 1;
      {
        const uint4 tl = ix.t_bin[bo + bl], th = ix.t_bin[bo + bh + 1];
        tw[0] = tl.x; tw[1] = th.y; tw[2] = tl.z; tw[3] = th.w;
      }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const bool use = ((smode >> s) & 1u) && sb[s] != se[s];
        hi[s] = use ? tw[2 * s + 1] : sb[s];
        lo[s] = use ? (tw[2 * s] < hi[s] ? tw[2 * s] : hi[s]) : sb[s];
      }
      if ((hi[0] - lo[0]) + (hi[1] - lo[1]) > 64u) {
        // Dense locus: more rows than the survivor mask has bits.  Tighten to the exact range (hi = first row with
        // start >= qend, lo = first row whose running max end exceeds qstart) with a G-wide count inside the two bins,
        // so that only alignments with > 64 overlapping-range rows go to the wave-per-alignment emit kernel.
        uint32_t ra[4], rb[4];  // 0/1: hi/lo on '+', 2/3: hi/lo on '-'
        const uint4 th0 = ix.t_bin[bo + bh], tl1 = ix.t_bin[bo + bl + 1];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          ra[2 * s] = rb[2 * s] = ra[2 * s + 1] = rb[2 * s + 1] = sb[s];
          if (!((smode >> s) & 1u) || sb[s] == se[s]) continue;
          ra[2 * s] = s ? th0.w : th0.y; rb[2 * s] = hi[s];
          ra[2 * s + 1] = tw[2 * s]; rb[2 * s + 1] = s ? tl1.z : tl1.x;
        }
        uint32_t res[4] = {ra[0], ra[1], ra[2], ra[3]};
        for (uint32_t it = 0;; it += G) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < 4; k++) any |= (ra[k] + it) < rb[k];
          if (!any) break;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            uint32_t r = ra[k] + it + (uint32_t)gl;
            bool in = r < rb[k];
            uint32_t v = in ? ((k & 1) ? ix.s_pmax[r] : ix.s_start[r]) : 0u;
            bool t = in && ((k & 1) ? (v <= q0.x) : (v < q0.y));
            res[k] += (uint32_t)__popcll((__ballot(t) >> gbase) & gmask);
          }
        }
        hi[0] = res[0]; lo[0] = res[1] < res[0] ? res[1] : res[0];
        hi[1] = res[2]; lo[1] = res[3] < res[2] ? res[3] : res[2];
      }
      if (gl == 0) A.ranges[a] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
    }
    uint32_t n0 = hi[0] - lo[0], n1 = hi[1] - lo[1];
    uint32_t n_items = n0 + n1;

    ReadCtx rd;
    rd.n_seg = n_seg; rd.seg = nullptr; rd.real = nullptr; rd.n_real = 0; rd.q12 = hd2;
    uint32_t moff = 0; uint64_t cbase = 0; uint32_t cap = 0, ideal_cap = 0;
    bool have_mask = false; uint64_t mask_in = 0;
    if (EMIT || n_seg > 3) {
      uint32_t c0 = A.cigar_off[a];
      rd.seg = A.seg + (size_t)c0 + (size_t)a;
      rd.real = A.cigar + c0;
      if (EMIT) rd.n_real = A.cigar_off[a + 1] - c0;
    }
    if (EMIT) {
      moff = A.match_off[a]; cbase = A.cig_base[a];
      ideal_cap = 4u * n_seg + 2u;
      cap = rd.n_real + 2u * ideal_cap;
      have_mask = n_items <= 64 && !A.p1;   // the single-pass count kernel stores no masks (and sends alignments with > P1_STASH survivors here too)
      mask_in = have_mask ? A.mask[a] : 0;
    }
    uint32_t total = 0;
    uint64_t mask_all = 0;
    bool defer = false;  // MODE 1: some candidate needs the walk

    // EMIT without a stored mask (> 64 candidate rows): sweep 0 records every
    // survivor's tid in m_aux[], sweep 1 ranks against that list.
    const int n_sweeps = (EMIT && !have_mask) ? 2 : 1;
    uint32_t aux_n = 0;
    for (int sweep = 0; sweep < n_sweeps; sweep++) {
      for (uint32_t base = 0; base < n_items; base += G) {
        uint32_t item = base + (uint32_t)gl;
        bool valid = item < n_items;
        bool alive = false;
        int s = 0; uint32_t row = 0, gs = 0, gend = 0, nxt = 0, nxe = 0;
        uint4 pay = make_uint4(0xffffffffu, 0, 0, 0);
        Hit h0; CandOut p1; const uint4 *E = nullptr; uint32_t i0 = 0;
        p1.alive = false; p1.fwpos = 0; p1.rcpos = 0; p1.n_seg = 0; p1.n_gex = 0;
        h0.pos = 0; h0.left_ins = h0.right_ins = h0.left_gap = h0.right_gap = 0;
        if (valid) {
          s = item < n0 ? 0 : 1;
          row = s == 0 ? lo[0] + item : lo[1] + (item - n0);
          // one round trip, one sector: the 32-byte row
          const uint4 r_a = ix.s_row[2 * (size_t)row], r_b = ix.s_row[2 * (size_t)row + 1];
          gs = r_a.x; gend = r_a.y; nxt = r_a.z; nxe = r_b.w; pay = make_uint4(r_b.x, r_b.y, r_a.w, r_b.z);
          bool want = true;
          if (EMIT && have_mask) want = (mask_in >> item) & 1ull;
          if (want && gend > q0.x && gs < q0.y && classify(s == 1, st0, q0.x, q0.y, gs, gend, pay.z, cfg, h0)) {
            E = ix.tx_ex + pay.w; i0 = pay.y & 0x7fffffffu;  // bit 31: the transcript has more than 256 exons
            // first-exon duplicate tid: the LAST passing row of the tid wins
            // (src/evaluate.cpp:218-224); later rows of the same tid are the
            // following exons of its table.  r_a.z = start of the next one.
            bool superseded = false;
            if (nxt < q0.y) {
              for (uint32_t i = i0 + 1;; i++) {
                uint4 e = E[i];
                if (e.x >= q0.y) break;
                Hit hx;
                if (e.y > q0.x && classify(s == 1, st0, q0.x, q0.y, e.x, e.y, e.z, cfg, hx)) { superseded = true; break; }
              }
            }
            if (!superseded) {
              if (n_seg == 1) { p1.alive = true; p1.fwpos = h0.pos; p1.rcpos = h0.pos; p1.n_seg = 1; p1.n_gex = 1; p1.i_lastm = i0; p1.last_right_ins = h0.right_ins; p1.last_right_gap = h0.right_gap; }
              else if (!EMIT && !SIMF && n_seg == 2 && (nxt >= hd2.y || nxe >= hd2.y)) {
                // Count pass, two read exons, and the scan of the second one ends at the exon after the candidate's
                // (the usual spliced short read): survival without the walk -- step_exon + walk_pass1 for j = 1 with
                // status LAST, written out.  The second read exon must pass against exactly one guide exon, and that
                // one must be the next exon (a hit on the candidate's own exon is "same guide exon twice").
                const uint32_t qs1 = hd2.x, qe1 = hd2.y;
                Hit hh;
                bool alive2 = false;
                if (gend < qe1 && !(qs1 == q0.x && qe1 == q0.y)) {
                  const bool c0 = gend > qs1 && classify(s == 1, ST_LAST, qs1, qe1, gs, gend, pay.z, cfg, hh);
                  const uint4 e1 = next_row(make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), s == 1);
                  const bool c1 = nxt < qe1 && nxe > qs1 && classify(s == 1, ST_LAST, qs1, qe1, e1.x, e1.y, e1.z, cfg, hh);
                  alive2 = c1 && !c0;
                }
                p1.alive = alive2;
              }
              else if (MODE == 1 && !cfg.long_reads && !cfg.ignore_small_exons && !(pay.y >> 31)) {
                // Defer only what can still survive.  In these presets every read exon must hit exactly the next guide
                // exon (gap 1, no skipped or inserted exon), so a candidate whose second read exon does not pass against
                // the exon after its own -- or passes against its own -- is dead whatever the rest of the walk finds.
                // Not for transcripts with more than 256 exons: the reference's uint8 exon-id arithmetic also accepts a hit
                // 257 exons on (evaluate.cpp:131), which only the walk sees.
                const uint32_t qs1 = hd2.x, qe1 = hd2.y;
                const int st1 = n_seg == 2 ? ST_LAST : ST_MIDDLE;
                Hit hh;
                if (gend < qe1 && !(qs1 == q0.x && qe1 == q0.y)) {
                  const bool c0 = gend > qs1 && classify(s == 1, st1, qs1, qe1, gs, gend, pay.z, cfg, hh);
                  const uint4 e1 = next_row(make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), s == 1);
                  const bool c1 = nxt < qe1 && nxe > qs1 && classify(s == 1, st1, qs1, qe1, e1.x, e1.y, e1.z, cfg, hh);
                  if (c1 && !c0) defer = true;
                }
                p1.alive = false;
              }
              else if (MODE == 1) { defer = true; p1.alive = false; }
              else p1 = walk_pass1(ix, cfg, rd, E, s == 1, sb[s], se[s], i0, make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), q0, h0);
              alive = p1.alive;
            }
          }
        }
        Acc acc; IdealSink sk; double score = 0.0;
        if (!EMIT || !have_mask) {
          // similarity filter needs the pass-2 accumulators (long reads only; SIMF is
          // cfg.filter_by_similarity lifted to compile time so the short-read kernel drops this code)
          if (SIMF && alive) {
            sk.init(nullptr);
            walk_pass2(ix, cfg, rd, E, s == 1, sb[s], se[s], i0, q0, make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), h0, p1, acc, sk, no_clip(), no_clip());
            alive = similarity(cfg, acc, score);
          }
        }
        uint32_t rank = 0;
        bool do_emit = false;
        if (!EMIT) {
          uint64_t m = (__ballot(alive) >> gbase) & gmask;
          total += (uint32_t)__popcll(m);
          if (base < 64) mask_all |= (G == 64) ? m : (m << base);
        } else if (have_mask) {
          // rank by tid among the read's survivors (group-uniform loop over the mask):
          // rows of this chunk come from a lane, the others from the slab.
          uint64_t mm = mask_in;
          while (mm) {
            int b = __ffsll((long long)mm) - 1; mm &= mm - 1;
            uint32_t t2;
            if ((uint32_t)b >= base && (uint32_t)b < base + G) t2 = __shfl(pay.x, b - (int)base, G);
            else {
              uint32_t r2 = (uint32_t)b < n0 ? lo[0] + (uint32_t)b : lo[1] + ((uint32_t)b - n0);
              t2 = ix.s_tid[r2];
            }
            rank += (t2 < pay.x) ? 1u : 0u;
          }
          do_emit = alive;
        } else {
          uint64_t m = (__ballot(alive) >> gbase) & gmask;
          if (sweep == 0) {
            if (alive) {
              uint32_t k = aux_n + (uint32_t)__popcll(m & ((1ull << gl) - 1ull));
              A.m_aux[moff + k] = pay.x;
            }
            aux_n += (uint32_t)__popcll(m);
          } else if (alive) {
            for (uint32_t k = 0; k < aux_n; k++) rank += (A.m_aux[moff + k] < pay.x) ? 1u : 0u;
            do_emit = true;
          }
        }
        if (EMIT && do_emit) {
          uint32_t *slot = A.cig_arena + cbase + (uint64_t)rank * cap;
          uint32_t *lds = &sh_cig[(EMIT ? threadIdx.x : 0) * LDS_SLOT];
          bool ideal_lds = ideal_cap <= LDS_IDEAL;
          uint32_t *ideal = ideal_lds ? lds : slot + rd.n_real + ideal_cap;
          sk.init(ideal);
          walk_pass2(ix, cfg, rd, E, s == 1, sb[s], se[s], i0, q0, make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), h0, p1, acc, sk, no_clip(), no_clip());
          uint32_t n_ideal = sk.finish();
          similarity(cfg, acc, score);
          bool out_lds = ideal_lds && (rd.n_real + n_ideal <= LDS_SLOT - LDS_IDEAL);
          uint32_t *outp = out_lds ? lds + LDS_IDEAL : slot;
          RealCig rc; rc.fetch(rd.real, rd.n_real);
          uint32_t n_out = merge_cigars(rc, rd.n_real, ideal, n_ideal, outp);
          uint64_t cig_ref = cbase + (uint64_t)rank * cap;
          if (n_out <= 2) cig_ref = (uint64_t)(n_out > 0 ? outp[0] : 0u) | ((uint64_t)(n_out > 1 ? outp[1] : 0u) << 32);
          else if (out_lds) for (uint32_t k = 0; k < n_out; k++) slot[k] = outp[k];
          uint32_t mi = moff + rank;
          A.m_tid[mi] = pay.x;
          A.m_p[mi] = make_uint2((s == 0) ? p1.fwpos : p1.rcpos, n_out | ((uint32_t)s << 31));
          A.m_x[mi] = make_uint2((uint32_t)acc.junc_hits, (uint32_t)acc.ref_consumed);
          if (SIMF) {
            unsigned long long sb64 = (unsigned long long)__double_as_longlong(score);
            A.m_b[mi] = make_uint4((uint32_t)acc.clip_score, 0u, (uint32_t)sb64, (uint32_t)(sb64 >> 32));
          }
          A.m_cigoff[mi] = cig_ref;
        }
      }
      if (EMIT && !have_mask && sweep == 0) __threadfence_block();  // m_aux[] written above is read below
    }
    if (MODE == 1 && ((__ballot(defer) >> gbase) & gmask) != 0) {  // group-uniform: the second pass redoes this alignment
      if (gl == 0) {
        uint32_t k = atomicAdd(&sh_wn, 1u);
        if (k < WALK_LDS) sh_wl[MODE == 1 ? k : 0] = (uint32_t)a;
        else { uint32_t k2 = atomicAdd(A.n_walk, 1u); A.walk_list[k2] = (uint32_t)a; }   // LDS list full: rare, slow, correct
      }
      continue;
    }
    if (!EMIT && gl == 0) {
      A.n_matches[a] = total; A.mask[a] = mask_all;
      if (n_items > 64 && total) {
        const uint32_t k = atomicAdd(&sh_bn, 1u);
        if (k < BIG_LDS) sh_bl[EMIT ? 0 : k] = (uint32_t)a;
        else { const uint32_t k2 = atomicAdd(A.n_big, 1u); A.big_list[k2] = (uint32_t)a; }   // (the block's list is full)
      }
    }
  }
  if (!EMIT) {
    __syncthreads();
    const uint32_t n_loc = sh_bn < BIG_LDS ? sh_bn : BIG_LDS;
    if (threadIdx.x == 0) sh_bbase = n_loc ? atomicAdd(A.n_big, n_loc) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_loc; i += blockDim.x) A.big_list[sh_bbase + i] = sh_bl[EMIT ? 0 : i];
  }
  if (MODE == 1) {
    __syncthreads();
    const uint32_t n_loc = sh_wn < WALK_LDS ? sh_wn : WALK_LDS;
    if (threadIdx.x == 0) sh_wbase = n_loc ? atomicAdd(A.n_walk, n_loc) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_loc; i += blockDim.x) A.walk_list[sh_wbase + i] = sh_wl[MODE == 1 ? i : 0];
  }
}

// k_expand: the emit work list, one entry per match:
// (alignment, k-th survivor) pairs, simple alignments first so that whole waves of
// k_emit_dense take the short path.  Alignments with > 64 candidate rows are left
// to the group kernel (~0u entries).
// (a = the thread's alignment; every wave of the block calls it, whole waves at a time)
__device__ __forceinline__ void expand_chunk(const ProjectArgs &A, int64_t a, uint32_t (*sh_pre)[64], uint32_t (*sh_v)[64]) {
  // A wave takes 64 alignments.  Their entries of one class are one contiguous run of the list (the class offsets are
  // prefix sums over the alignments), so the wave writes the run with all its lanes -- entry e belongs to the alignment
  // whose exclusive count prefix is the last one <= e -- instead of every lane writing its own n_matches entries one by one.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t nm = 0, pos = 0, v = 0;
  bool fast = false;
  if (a < A.n_aln) nm = A.n_matches[a];
  if (nm) {
    const uint32_t fp = A.fast_pre[a];
    fast = A.fast_flag[a] >> 31;
    pos = fast ? fp : A.fast_pre[A.n_aln] + (A.match_off[a] - fp);
    const uint4 rg = A.ranges[a];
    const uint32_t n_items = (rg.y - rg.x) + (rg.w - rg.z);
    v = n_items <= 64 ? (uint32_t)a : 0xffffffffu;
  }
  sh_v[wv][lane] = v;
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const uint32_t mine = (nm && fast == (c == 0)) ? nm : 0u;
    const uint64_t have = __ballot(mine != 0u);
    if (!have) continue;                       // the same for the whole wave
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d); if (lane >= d) inc += up; }
    const uint32_t total = __shfl(inc, 63), first = __shfl(pos, (int)__builtin_ctzll(have));
    __builtin_amdgcn_wave_barrier();           // (the previous class's reads of sh_pre are done)
    sh_pre[wv][lane] = inc - mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (uint32_t e = lane; e < total; e += 64) {
      int l = 0;
#pragma unroll
      for (int st = 32; st; st >>= 1) if (sh_pre[wv][l + st] <= e) l += st;
      A.m_aln[first + e] = sh_v[wv][l];
    }
  }
  __builtin_amdgcn_wave_barrier();   // (the next chunk of a looping caller rewrites sh_v)
}
__global__ void __launch_bounds__(256) k_expand(ProjectArgs A) {
  __shared__ uint32_t sh_pre[4][64], sh_v[4][64];
  if (tot_over(A.tot, A.lim_m, A.lim_c)) return;
  expand_chunk(A, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, sh_pre, sh_v);
}

// k_emit_dense: one lane per match.  Match mi of alignment a is the k-th set bit
// (k = mi - match_off[a]) of the survivor mask the count pass stored; the lane
// re-derives the candidate from its slab row, ranks it by tid among the read's
// survivors, builds the ideal CIGAR, merges it with the real CIGAR and writes the
// match record at match_off[a] + rank.
// CLS: 0 = the whole work list, 1 = its simple prefix only (no walk, no merge loop, no LDS: a much lighter kernel),
// 2 = the rest; the list is partitioned by class (k_expand), `first` is where this launch starts
// FA: the -S rescue's emit pass (long reads): the survivor mask and the candidates' rescue wishes come from
// k_project_fa<2>, the clip segments from the DP's results; no fast class.
template <bool SIMF, int CLS, bool FA = false>
__global__ void __launch_bounds__(256, CLS == 1 ? 8 : 6) k_emit_dense(ProjectArgs A, int64_t first, int64_t n_matches, FaArgs F) {
  __shared__ uint32_t sh_cig[CLS == 1 ? 1 : 256 * LDS_SLOT];
  __shared__ uint16_t sh_mops[CLS == 1 ? 1 : 256];
  if (CLS != 1) {
    sh_mops[threadIdx.x] = (uint16_t)((merge_action(threadIdx.x >> 4, threadIdx.x & 15u) << 8) | merge_ops(threadIdx.x >> 4, threadIdx.x & 15u));
    __syncthreads();
  }
  if (A.tot) {   // the totals never left the device: the whole list (CLS 0), its simple prefix (1) or the rest (2) from tot[]
    if (tot_over(A.tot, A.lim_m, A.lim_c)) return;
    n_matches = CLS == 1 ? (int64_t)A.tot[2] : (int64_t)A.tot[0];
    first = CLS == 2 ? (int64_t)A.tot[2] : 0;
    if ((uint64_t)(n_matches - first) > A.cover) return;   // the grid was sized from a prediction that fell short: the host sees the same and redoes the batch
  }
  int64_t mi64 = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mi64 >= n_matches) return;
  const DevIndex &ix = A.ix;
  const DevCfg &cfg = A.cfg;
  uint32_t a = A.m_aln[mi64];
  if (a == 0xffffffffu) return;
  uint4 hd = A.head[a];
  uint4 hd2 = A.head2[a];
  uint4 rg = A.ranges[a];
  uint64_t mask = A.mask[a];
  const uint32_t part_fast = CLS == 1 ? 1u : CLS == 2 ? 0u : (A.fast_flag[a] >> 31);   // which part of the work list the entry is in
  const uint32_t is_fast = FA ? 0u : part_fast;
  uint32_t moff = A.match_off[a];
  uint64_t cbase = A.cig_base[a];
  uint32_t c0 = A.cigar_off[a], c1 = A.cigar_off[a + 1];
  uint32_t n_seg = hd.z, rid = hd.w >> 2;
  uint2 q0 = make_uint2(hd.x, hd.y);
  int st0 = (n_seg == 1) ? ST_ONLY : ST_FIRST;
  uint32_t n0 = rg.y - rg.x;
  RealCig rc;
  if (CLS != 1) rc.fetch(A.cigar + c0, c1 - c0);  // in flight with the row and rank loads below
  // k-th survivor in candidate-row order
  // which survivor: the entry's distance from the alignment's first entry in the class-partitioned list (k_expand)
  const uint32_t fpre = A.fast_pre[a];
  uint32_t k = (uint32_t)mi64 - (part_fast ? fpre : A.fast_pre[A.n_aln] + (moff - fpre));
  uint64_t mm = mask;
  for (uint32_t j = 0; j < k; j++) mm &= mm - 1;
  uint32_t item = (uint32_t)(__ffsll((long long)mm) - 1);
  int s = item < n0 ? 0 : 1;
  uint32_t row = s == 0 ? rg.x + item : rg.z + (item - n0);
  const uint4 r_a = ix.s_row[2 * (size_t)row], r_b = ix.s_row[2 * (size_t)row + 1];
  uint32_t gs = r_a.x, gend = r_a.y;
  uint4 pay = make_uint4(r_b.x, r_b.y, r_a.w, r_b.z);
  // rank by tid among the survivors
  uint32_t rank = 0;
  for (uint64_t m2 = mask; m2;) {  // eight loads in flight per step: unconditional (spent slots re-read the lane's own row)
    uint32_t t8[8]; bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      ok[u] = m2 != 0;
      uint32_t b = ok[u] ? (uint32_t)(__ffsll((long long)m2) - 1) : item;
      m2 &= m2 - 1;
      t8[u] = ix.s_tid[b < n0 ? rg.x + b : rg.z + (b - n0)];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) rank += (ok[u] && t8[u] < pay.x) ? 1u : 0u;
  }
  Hit h0;
  classify(s == 1, st0, q0.x, q0.y, gs, gend, pay.z, cfg, h0);
  if (is_fast) {
    // one read exon from a single M op (short-read presets): the ideal CIGAR is
    // [S left_ins] M [S right_ins] (src/evaluate.cpp:699-706,739-749,752-759) and merging it
    // with "<len>M" returns it unchanged (src/bam.cpp:85-88,96-98) -- no walk, no merge loop.
    uint32_t os = q0.x > gs ? q0.x : gs, oe = q0.y < gend ? q0.y : gend;
    uint32_t ml = oe - os;
    uint32_t w0, w1 = 0, w2 = 0, n_out;
    if (h0.left_ins) { w0 = CIG_GEN(h0.left_ins, OP_S); w1 = CIG_GEN(ml, OP_M); n_out = 2; if (h0.right_ins) { w2 = CIG_GEN(h0.right_ins, OP_S); n_out = 3; } }
    else { w0 = CIG_GEN(ml, OP_M); n_out = 1; if (h0.right_ins) { w1 = CIG_GEN(h0.right_ins, OP_S); n_out = 2; } }
    uint32_t junc = ((h0.left_ins == 0 && h0.left_gap == 0) ? 1u : 0u) + ((h0.right_ins == 0 && h0.right_gap == 0) ? 1u : 0u);
    uint64_t cref = (uint64_t)w0 | ((uint64_t)w1 << 32);
    if (n_out == 3) {
      cref = cbase + (uint64_t)rank * (1u + 2u * 6u);
      uint32_t *slot = A.cig_arena + cref;
      slot[0] = w0; slot[1] = w1; slot[2] = w2;
    }
    uint32_t mo = moff + rank;
    A.m_tid[mo] = pay.x;
    A.m_p[mo] = make_uint2(h0.pos, n_out | ((uint32_t)s << 31));
    A.m_x[mo] = make_uint2(junc, ml);
    if (SIMF) A.m_b[mo] = make_uint4(0u, 0u, 0u, 0u);
    A.m_cigoff[mo] = cref;
    return;
  }
  if (CLS == 1) return;
  ReadCtx rd;
  rd.n_seg = n_seg; rd.seg = A.seg + (size_t)c0 + (size_t)a; rd.real = A.cigar + c0; rd.n_real = c1 - c0; rd.q12 = hd2;
  const uint4 *E = ix.tx_ex + pay.w;
  uint32_t i0 = pay.y & 0x7fffffffu;
  CandOut p1;
  bool plain = false;
  uint32_t sb = ix.slab_off[2 * rid + s], se = ix.slab_off[2 * rid + s + 1];
  if (n_seg == 1) { p1.alive = true; p1.fwpos = h0.pos; p1.rcpos = h0.pos; p1.n_seg = 1; p1.n_gex = 1; p1.i_lastm = i0; p1.last_right_ins = h0.right_ins; p1.last_right_gap = h0.right_gap; }
  else if (!FA && !cfg.long_reads && !cfg.ignore_small_exons) {
    // short-read presets without --max-error-exon: a survivor's every read exon is one plain hit on the next guide exon
    // (no skipped-exon or inserted-exon segments), so pass 1's counts are the read's exon count and its walk is not
    // needed; rcpos comes out of pass 2 (the last hit's position)
    plain = true;
    p1.alive = true; p1.fwpos = h0.pos; p1.rcpos = h0.pos; p1.n_seg = n_seg; p1.n_gex = n_seg; p1.i_lastm = i0;
    p1.last_right_ins = 0; p1.last_right_gap = 0;
  }
  else p1 = walk_pass1(ix, cfg, rd, E, s == 1, sb, se, i0, make_uint4(gs, gend, pay.z, 0), make_uint2(r_a.z, r_b.w), q0, h0);
  // -S: the candidate's rescue problems are numbered in candidate order, left before right (k_project_fa<0 / 1>)
  ClipSide L = no_clip(), R = no_clip();
  if (FA) {
    const uint64_t wl = F.want_l[a], wr = F.want_r[a], lower = (1ull << item) - 1ull;
    const uint32_t pL = F.prob_off[a] + (uint32_t)__popcll(wl & lower) + (uint32_t)__popcll(wr & lower);
    const uint32_t pR = pL + (uint32_t)((wl >> item) & 1ull);
    if ((wl >> item) & 1ull) {
      const KswRes rs = F.results[pL];
      if (rs.ok) { L.ok = true; L.ops = F.clip_ops + (F.probs[pL].seq_off + pL); L.n_ops = rs.n_ops; L.score = rs.score; L.refc = (uint32_t)rs.refc; }
    }
    if ((wr >> item) & 1ull) {
      const KswRes rs = F.results[pR];
      if (rs.ok) { R.ok = true; R.ops = F.clip_ops + (F.probs[pR].seq_off + pR); R.n_ops = rs.n_ops; R.score = rs.score; R.refc = (uint32_t)rs.refc; }
    }
    apply_clips(p1, s == 1, pay.z, E[p1.i_lastm].z, L, R);
  }
  uint32_t ideal_cap = FA ? F.ideal_cap[a] : 4u * n_seg + 2u;
  uint32_t cap = rd.n_real + 2u * ideal_cap;
  uint32_t *slot = A.cig_arena + cbase + (uint64_t)rank * cap;
  uint32_t *lds = &sh_cig[threadIdx.x * LDS_SLOT];
  bool ideal_lds = ideal_cap <= LDS_IDEAL;
  uint32_t *ideal = ideal_lds ? lds : slot + rd.n_real + ideal_cap;
  Acc acc; IdealSink sk; double score = 0.0;
  sk.init(ideal);
  walk_pass2(ix, cfg, rd, E, s == 1, sb, se, i0, q0, make_uint4(gs, gend, pay.z, 0), make_uint2(r_a.z, r_b.w), h0, p1, acc, sk, L, R);
  uint32_t n_ideal = sk.finish();
  if (SIMF) similarity(cfg, acc, score);
  bool out_lds = ideal_lds && (rd.n_real + n_ideal <= LDS_SLOT - LDS_IDEAL);
  uint32_t *outp = out_lds ? lds + LDS_IDEAL : slot;
  uint32_t n_out = merge_cigars(rc, rd.n_real, ideal, n_ideal, outp, sh_mops);
  // rewritten CIGARs of <= 2 ops travel inside the match record (m_cigoff holds the
  // words themselves); longer ones stay in the arena slot
  uint64_t cig_ref = cbase + (uint64_t)rank * cap;
  if (n_out <= 2) cig_ref = (uint64_t)(n_out > 0 ? outp[0] : 0u) | ((uint64_t)(n_out > 1 ? outp[1] : 0u) << 32);
  else if (out_lds) for (uint32_t q = 0; q < n_out; q++) slot[q] = outp[q];
  uint32_t mo = moff + rank;
  A.m_tid[mo] = pay.x;
  A.m_p[mo] = make_uint2((s == 0) ? p1.fwpos : (plain ? acc.last_pos : p1.rcpos), n_out | ((uint32_t)s << 31));
  A.m_x[mo] = make_uint2((uint32_t)acc.junc_hits, (uint32_t)acc.ref_consumed);
  // without the similarity filter the score is 0.0 and (no -S rescue in this kernel) the clip score is 0: the row
  // kernel then neither reads m_b nor rewrites the two all-zero row columns
  if (SIMF) {
    unsigned long long sb64 = (unsigned long long)__double_as_longlong(score);
    A.m_b[mo] = make_uint4((uint32_t)acc.clip_score, 0u, (uint32_t)sb64, (uint32_t)(sb64 >> 32));
  }
  A.m_cigoff[mo] = cig_ref;
}

// ---------------------------------------------------------------------------
// Single pass (short-read presets, no similarity filter, no -S): count and emit in one sweep.
//
// The two-pass path re-fetches, re-classifies and re-ranks in the emit pass what the count pass had in registers, and
// needs a scan, a work-list expansion and per-alignment ranges / masks to connect the two.  Here the G-lane group that
// finds an alignment's survivors also places them: survivors (slab row, tid) go to a small LDS stash while the rows are
// tested; afterwards the wave takes the slices its eight alignments need -- match slots, work-list entries, CIGAR arena
// words -- from PAGES THE WAVE OWNS (one global atomic per page, not per alignment: same-address atomics serialise at
// ~9 ns each), each survivor ranks itself by tid against the stash, and
//   * the simple class (one read exon from a single M op) writes its match record right away,
//   * the general class writes a 16-byte work item {alignment, row, match slot, rank} for k_emit_wl, which then starts
//     from the item instead of from masks, ranges, scanned offsets and eight rank keys.
// A wave walks CHUNKS of consecutive alignments (P1_CHUNK at a time), so the matches of neighbouring alignments stay
// neighbours in the match table (k_pair / k_rows read it in alignment order).  Rows are still addressed by alignment
// (match_off[a], n_matches[a]): the result does not depend on which wave got which page.
// Capacities are the host's guess; a wave that would exceed one sets the overflow flag and stops writing, the host
// grows the buffers to what the counters say and runs the pass again.
// ---------------------------------------------------------------------------
#define P1_STASH 32       // survivors of one alignment kept in LDS; more (or > 64 candidate rows): the dense-locus kernel
#define P1_CHUNK 128      // consecutive alignments a wave takes at a time
#define WALK1_LDS 1024
#define P1_SLAB 513

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }
// Wave-uniform (all arguments are the same in every lane; the state lives in scalar registers): `need` units from the
// wave's page [cur, end); a new page (or, for a request beyond a page, exactly the request) comes from the global counter
// with one atomic.  Returns the first unit; ovf is set past the capacity.
__device__ __forceinline__ uint32_t page_take32(uint32_t &cur, uint32_t &end, uint32_t need, uint64_t *ctr, uint32_t page,
                                                uint64_t cap, bool &ovf, int lane) {
  if (need != 0 && (uint64_t)cur + need > (uint64_t)end) {
    const uint32_t take = need > page ? need : page;
    unsigned long long b = 0;
    if (lane == 0) b = atomicAdd((unsigned long long *)ctr, (unsigned long long)take);
    b = rfl64(b);
    if (b + take > cap) ovf = true;
    cur = (uint32_t)b; end = cur + take;
  }
  const uint32_t r = cur;
  cur = rfl(cur + need);
  return r;
}
__device__ __forceinline__ uint64_t page_take64(uint64_t &cur, uint64_t &end, uint64_t need, uint64_t *ctr, uint64_t page,
                                                uint64_t cap, bool &ovf, int lane) {
  if (need != 0 && cur + need > end) {
    const uint64_t take = need > page ? need : page;
    unsigned long long b = 0;
    if (lane == 0) b = atomicAdd((unsigned long long *)ctr, (unsigned long long)take);
    b = rfl64(b);
    cur = b; end = b + take;
    if (end > cap) ovf = true;
  }
  const uint64_t r = cur;
  cur = rfl64(cur + need);
  return r;
}
// inclusive sum over the groups of a wave (every lane of a group holds the group's value)
template <int G, typename T>
__device__ __forceinline__ T wave_group_scan(T v, int lane) {
  T x = v;
#pragma unroll
  for (int d = G; d < 64; d <<= 1) { const T y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
  return x;
}

template <int G, int MODE>
__global__ void __launch_bounds__(256, 8) k_project1(ProjectArgs A) {
  __shared__ uint32_t sh_slab[P1_SLAB];   // (<= 256 references; more: the offsets come from memory)
  __shared__ uint32_t sh_bin[P1_SLAB];
  __shared__ uint32_t sh_wl[MODE == 1 ? WALK1_LDS : 1];
  __shared__ uint32_t sh_wn, sh_wbase;
  __shared__ uint64_t sh_st[256 / G][P1_STASH];   // survivor stash: tid << 32 | strand << 31 | slab row
  if (MODE == 1 && threadIdx.x == 0) sh_wn = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = threadIdx.x & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int grp = threadIdx.x / G;
  const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << G) - 1);
  constexpr int GPW = 64 / G;
  const DevIndex &ix = A.ix;
  const DevCfg &cfg = A.cfg;
  const uint32_t n_slab_off = 2 * ix.n_refs + 1;
  const bool slab_in_lds = n_slab_off <= P1_SLAB;
  if (slab_in_lds) {
    for (uint32_t i = threadIdx.x; i < n_slab_off; i += blockDim.x) {
      sh_slab[i] = ix.slab_off[i];
      if (i <= ix.n_refs) sh_bin[i] = ix.bin_off[i];
    }
  }
  __syncthreads();
  const uint32_t n_work = MODE == 2 ? *A.n_walk : (uint32_t)A.n_aln;   // (n_aln < 2^31, checked by the host)
  const uint32_t n_waves = gridDim.x * 4;
  // the wave's pages (wave-uniform, scalar registers) and what it found
  uint32_t m_cur = 0, m_end = 0, w_cur = 0, w_end = 0;
  uint64_t c_cur = 0, c_end = 0;
  bool ovf = false;
  uint32_t found_m = 0, found_w = 0;
  for (uint32_t w_first = rfl((blockIdx.x * 4 + wv) * P1_CHUNK); w_first < n_work; w_first += n_waves * P1_CHUNK) {
   for (uint32_t it = 0; it < P1_CHUNK / GPW && w_first + it * GPW < n_work; it++) {
    const uint32_t w = w_first + it * GPW + (uint32_t)(lane / G);
    const bool valid = w < n_work;
    const uint32_t a = valid ? (MODE == 2 ? A.walk_list[w] : w) : 0u;
    uint4 hd = make_uint4(0, 0, 0, 0), hd2 = hd;
    uint32_t ff = 0;
    if (valid) { hd = A.head[a]; hd2 = A.head2[a]; ff = A.fast_flag[a]; }
    const uint32_t n_seg = hd.z;
    const uint2 q0 = make_uint2(hd.x, hd.y);
    const uint32_t smode = hd.w & 3u, rid = hd.w >> 2;
    const int st0 = (n_seg == 1) ? ST_ONLY : ST_FIRST;
    uint32_t sb[2], se[2];
    if (slab_in_lds) { sb[0] = sh_slab[2 * rid]; se[0] = sh_slab[2 * rid + 1]; se[1] = sh_slab[2 * rid + 2]; }
    else { sb[0] = ix.slab_off[2 * rid]; se[0] = ix.slab_off[2 * rid + 1]; se[1] = ix.slab_off[2 * rid + 2]; }
    sb[1] = se[0];

    // candidate rows of read exon 0 from the bucket tables (see k_project)
    uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
    {
      uint32_t tw[4];
      const uint32_t bo = slab_in_lds ? sh_bin[rid] : ix.bin_off[rid];
      const uint32_t nb = (slab_in_lds ? sh_bin[rid + 1] : ix.bin_off[rid + 1]) - bo - 1;
      uint32_t bh = q0.y >> ix.bin_shift, bl = q0.x >> ix.bin_shift;
      bh = bh < nb - 1 ? bh : nb - 1; bl = bl < nb - 1 ? bl : nb - 1;
      {
        const uint4 tl = ix.t_bin[bo + bl], th = ix.t_bin[bo + bh + 1];
        tw[0] = tl.x; tw[1] = th.y; tw[2] = tl.z; tw[3] = th.w;
      }
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const bool use = ((smode >> s) & 1u) && sb[s] != se[s];
        hi[s] = use ? tw[2 * s + 1] : sb[s];
        lo[s] = use ? (tw[2 * s] < hi[s] ? tw[2 * s] : hi[s]) : sb[s];
      }
      if ((hi[0] - lo[0]) + (hi[1] - lo[1]) > 64u) {
        uint32_t ra[4], rb[4];  // 0/1: hi/lo on '+', 2/3: hi/lo on '-'
        const uint4 th0 = ix.t_bin[bo + bh], tl1 = ix.t_bin[bo + bl + 1];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          ra[2 * s] = rb[2 * s] = ra[2 * s + 1] = rb[2 * s + 1] = sb[s];
          if (!((smode >> s) & 1u) || sb[s] == se[s]) continue;
          ra[2 * s] = s ? th0.w : th0.y; rb[2 * s] = hi[s];
          ra[2 * s + 1] = tw[2 * s]; rb[2 * s + 1] = s ? tl1.z : tl1.x;
        }
        uint32_t res[4] = {ra[0], ra[1], ra[2], ra[3]};
        for (uint32_t itx = 0;; itx += G) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < 4; k++) any |= (ra[k] + itx) < rb[k];
          if (!any) break;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            uint32_t r = ra[k] + itx + (uint32_t)gl;
            bool in = r < rb[k];
            uint32_t v = in ? ((k & 1) ? ix.s_pmax[r] : ix.s_start[r]) : 0u;
            bool t = in && ((k & 1) ? (v <= q0.x) : (v < q0.y));
            res[k] += (uint32_t)__popcll((__ballot(t) >> gbase) & gmask);
          }
        }
        hi[0] = res[0]; lo[0] = res[1] < res[0] ? res[1] : res[0];
        hi[1] = res[2]; lo[1] = res[3] < res[2] ? res[3] : res[2];
      }
    }
    const uint32_t n0 = hi[0] - lo[0], n1 = hi[1] - lo[1];
    const uint32_t n_items = n0 + n1;

    ReadCtx rd;
    rd.n_seg = n_seg; rd.seg = nullptr; rd.real = nullptr; rd.n_real = 0; rd.q12 = hd2;
    if (MODE == 2 && n_seg > 3) {
      const uint32_t c0 = A.cigar_off[a];
      rd.seg = A.seg + (size_t)c0 + (size_t)a;
      rd.real = A.cigar + c0;
    }
    uint32_t total = 0;
    bool defer = false;
    for (uint32_t base = 0; base < n_items; base += G) {
      const uint32_t item = base + (uint32_t)gl;
      bool alive = false;
      int s = 0; uint32_t row = 0, tid = 0;
      if (item < n_items) {
        s = item < n0 ? 0 : 1;
        row = s == 0 ? lo[0] + item : lo[1] + (item - n0);
        const uint4 r_a = ix.s_row[2 * (size_t)row], r_b = ix.s_row[2 * (size_t)row + 1];
        const uint32_t gs = r_a.x, gend = r_a.y, nxt = r_a.z, nxe = r_b.w;
        const uint4 pay = make_uint4(r_b.x, r_b.y, r_a.w, r_b.z);
        tid = pay.x;
        Hit h0;
        if (gend > q0.x && gs < q0.y && classify(s == 1, st0, q0.x, q0.y, gs, gend, pay.z, cfg, h0)) {
          const uint4 *E = ix.tx_ex + pay.w;
          const uint32_t i0 = pay.y & 0x7fffffffu;
          bool superseded = false;   // first-exon duplicate tid: the LAST passing row of the tid wins (src/evaluate.cpp:218-224)
          if (nxt < q0.y) {
            for (uint32_t i = i0 + 1;; i++) {
              const uint4 e = E[i];
              if (e.x >= q0.y) break;
              Hit hx;
              if (e.y > q0.x && classify(s == 1, st0, q0.x, q0.y, e.x, e.y, e.z, cfg, hx)) { superseded = true; break; }
            }
          }
          if (!superseded) {
            if (n_seg == 1) alive = true;
            else if (MODE == 1 && n_seg == 2 && (nxt >= hd2.y || nxe >= hd2.y)) {
              // two read exons and the scan of the second ends at the exon after the candidate's: survival without the walk
              const uint32_t qs1 = hd2.x, qe1 = hd2.y;
              Hit hh;
              if (gend < qe1 && !(qs1 == q0.x && qe1 == q0.y)) {
                const bool c0 = gend > qs1 && classify(s == 1, ST_LAST, qs1, qe1, gs, gend, pay.z, cfg, hh);
                const uint4 e1 = next_row(make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), s == 1);
                const bool c1 = nxt < qe1 && nxe > qs1 && classify(s == 1, ST_LAST, qs1, qe1, e1.x, e1.y, e1.z, cfg, hh);
                alive = c1 && !c0;
              }
            }
            else if (MODE == 1 && !cfg.long_reads && !cfg.ignore_small_exons && !(pay.y >> 31)) {
              // defer only what can still survive (see k_project MODE 1)
              const uint32_t qs1 = hd2.x, qe1 = hd2.y;
              const int st1 = n_seg == 2 ? ST_LAST : ST_MIDDLE;
              Hit hh;
              if (gend < qe1 && !(qs1 == q0.x && qe1 == q0.y)) {
                const bool c0 = gend > qs1 && classify(s == 1, st1, qs1, qe1, gs, gend, pay.z, cfg, hh);
                const uint4 e1 = next_row(make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), s == 1);
                const bool c1 = nxt < qe1 && nxe > qs1 && classify(s == 1, st1, qs1, qe1, e1.x, e1.y, e1.z, cfg, hh);
                if (c1 && !c0) defer = true;
              }
            }
            else if (MODE == 1) defer = true;
            else alive = walk_pass1(ix, cfg, rd, E, s == 1, sb[s], se[s], i0, make_uint4(gs, gend, pay.z, 0), make_uint2(nxt, nxe), q0, h0).alive;
          }
        }
      }
      const uint64_t m = (__ballot(alive) >> gbase) & gmask;
      if (alive) {
        const uint32_t k = total + (uint32_t)__popcll(m & ((1ull << gl) - 1ull));
        if (k < P1_STASH) sh_st[grp][k] = ((uint64_t)tid << 32) | ((uint64_t)s << 31) | row;
      }
      total += (uint32_t)__popcll(m);
    }
    bool deferred = false;
    if (MODE == 1) deferred = ((__ballot(defer) >> gbase) & gmask) != 0;
    if (deferred) {
      if (gl == 0) {
        uint32_t k = atomicAdd(&sh_wn, 1u);
        if (k < WALK1_LDS) sh_wl[MODE == 1 ? k : 0] = a;
        else { uint32_t k2 = atomicAdd(A.n_walk, 1u); A.walk_list[k2] = a; }   // LDS list full: rare, slow, correct
      }
      total = 0;
    }
    // ---- placement: what the wave's alignments need, summed over its groups ----
    const bool big = total != 0 && (n_items > 64u || total > P1_STASH);   // left to k_project<64,true> (needs ranges / arena slots)
    const bool fast = (ff >> 31) != 0;
    const uint32_t cap = ff & 0x7fffffffu;
    const uint32_t need_m = total, need_w = (fast || big) ? 0u : total;
    const uint64_t need_c = (!fast || big) ? (uint64_t)total * cap : 0;
    uint32_t inc_m, inc_w; uint64_t inc_c;
    const bool wide = __any(big || cap > 0xffffu);   // wave-uniform
    if (!wide) {   // two 32-bit scans: <= 32 matches per group, cap <= 65535
      const uint32_t x = wave_group_scan<G, uint32_t>(need_m | (need_w << 16), lane);
      inc_m = x & 0xffffu; inc_w = x >> 16;
      inc_c = wave_group_scan<G, uint32_t>((uint32_t)need_c, lane);
    } else {
      inc_m = wave_group_scan<G, uint32_t>(need_m, lane);
      inc_w = wave_group_scan<G, uint32_t>(need_w, lane);
      inc_c = wave_group_scan<G, uint64_t>(need_c, lane);
    }
    const uint32_t tot_m = rfl(__shfl(inc_m, 63, 64)), tot_w = rfl(__shfl(inc_w, 63, 64));
    const uint64_t tot_c = rfl64(__shfl(inc_c, 63, 64));
    if (gl == 0 && valid && !deferred) A.n_matches[a] = total;
    if (tot_m == 0) continue;   // wave-uniform: nothing found by any group
    // hole markers for the work-list entries a page switch leaves behind (k_emit_wl walks the list densely)
    if (tot_w != 0 && (uint64_t)w_cur + tot_w > (uint64_t)w_end && !ovf) for (uint32_t i = w_cur + lane; i < w_end; i += 64) A.wl[i] = make_uint4(0xffffffffu, 0, 0, 0);
    const uint32_t m0 = page_take32(m_cur, m_end, tot_m, A.p1 + P1_M, P1_PAGE_M, A.cap_m, ovf, lane) + (inc_m - need_m);
    const uint32_t w0 = page_take32(w_cur, w_end, tot_w, A.p1 + P1_W, P1_PAGE_W, A.cap_w, ovf, lane) + (inc_w - need_w);
    const uint64_t c0 = page_take64(c_cur, c_end, tot_c, A.p1 + P1_C, P1_PAGE_C, A.cap_c, ovf, lane) + (inc_c - need_c);
    found_m += tot_m; found_w += tot_w;
    if (gl == 0 && valid && !deferred) {
      if (total && !ovf) {
        A.match_off_w[a] = m0;
        if (!fast || big) A.cig_base_w[a] = c0;
        if (big) {
          A.ranges[a] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
          uint32_t k = atomicAdd(A.n_big, 1u); A.big_list[k] = a;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- every survivor ranks itself by tid and is written ----
    const uint32_t n_emit = (big || ovf) ? 0u : total;
    for (uint32_t kb = 0; __any(kb < n_emit); kb += G) {   // wave-uniform trip count (the arena page below is the wave's)
      const uint32_t k = kb + (uint32_t)gl;
      const bool act = k < n_emit;
      uint32_t w_3 = 0;
      uint32_t rowS = 0, tid = 0, rank = 0;
      Hit h0; uint32_t gs = 0, gend = 0;
      h0.pos = 0; h0.left_ins = h0.right_ins = h0.left_gap = h0.right_gap = 0;
      if (act) {
        const uint64_t e = sh_st[grp][k];
        tid = (uint32_t)(e >> 32); rowS = (uint32_t)e;
        for (uint32_t j = 0; j < n_emit; j++) rank += ((uint32_t)(sh_st[grp][j] >> 32) < tid) ? 1u : 0u;
        if (fast) {
          const uint4 r_a = ix.s_row[2 * (size_t)(rowS & 0x7fffffffu)];   // the group loaded this line a moment ago
          gs = r_a.x; gend = r_a.y;
          classify((rowS >> 31) != 0, ST_ONLY, q0.x, q0.y, gs, gend, r_a.w, cfg, h0);
          w_3 = (h0.left_ins && h0.right_ins) ? 3u : 0u;
        } else {
          A.wl[w0 + k] = make_uint4(a, rowS, m0 + rank, rank);
        }
      }
      if (!__any(act && fast)) continue;
      // three-op CIGARs ([S] M [S] with both clips) live in the arena: three words each from the wave's page
      const uint64_t m3 = __ballot(w_3 != 0);
      uint64_t c3 = 0;
      if (m3) {
        bool ovf3 = false;
        c3 = page_take64(c_cur, c_end, 3ull * (uint64_t)__popcll(m3), A.p1 + P1_C, P1_PAGE_C, A.cap_c, ovf3, lane) + 3ull * (uint64_t)__popcll(m3 & ((1ull << lane) - 1ull));
        if (ovf3) { ovf = true; continue; }
      }
      if (act && fast) {
        // one read exon from a single M op: the ideal CIGAR is [S left_ins] M [S right_ins] and merging it with "<len>M"
        // returns it unchanged (see k_emit_dense)
        const uint32_t os = q0.x > gs ? q0.x : gs, oe = q0.y < gend ? q0.y : gend;
        const uint32_t ml = oe - os;
        uint32_t c_w0, c_w1 = 0, n_out;
        if (h0.left_ins) { c_w0 = CIG_GEN(h0.left_ins, OP_S); c_w1 = CIG_GEN(ml, OP_M); n_out = h0.right_ins ? 3u : 2u; }
        else { c_w0 = CIG_GEN(ml, OP_M); n_out = 1; if (h0.right_ins) { c_w1 = CIG_GEN(h0.right_ins, OP_S); n_out = 2; } }
        const uint32_t junc = ((h0.left_ins == 0 && h0.left_gap == 0) ? 1u : 0u) + ((h0.right_ins == 0 && h0.right_gap == 0) ? 1u : 0u);
        uint64_t cref = (uint64_t)c_w0 | ((uint64_t)c_w1 << 32);
        if (n_out == 3) {
          cref = c3;
          uint32_t *slot = A.cig_arena + c3;
          slot[0] = c_w0; slot[1] = c_w1; slot[2] = CIG_GEN(h0.right_ins, OP_S);
        }
        const uint32_t mo = m0 + rank;
        A.m_tid[mo] = tid;
        A.m_p[mo] = make_uint2(h0.pos, n_out | (rowS & 0x80000000u));
        A.m_x[mo] = make_uint2(junc, ml);
        A.m_cigoff[mo] = cref;
      }
    }
    __builtin_amdgcn_wave_barrier();   // the stash is rewritten by the next alignment
   }
  }
  // the rest of the wave's work-list page: hole markers
  if (!ovf) for (uint32_t i = w_cur + lane; i < w_end; i += 64) A.wl[i] = make_uint4(0xffffffffu, 0, 0, 0);
  if (lane == 0) {
    if (found_m) atomicAdd((unsigned long long *)(A.p1 + P1_NM), (unsigned long long)found_m);
    if (found_w) atomicAdd((unsigned long long *)(A.p1 + P1_NW), (unsigned long long)found_w);
    if (ovf) A.p1[P1_OVF] = 1;
  }
  if (MODE == 1) {
    __syncthreads();
    const uint32_t n_loc = sh_wn < WALK1_LDS ? sh_wn : WALK1_LDS;
    if (threadIdx.x == 0) sh_wbase = n_loc ? atomicAdd(A.n_walk, n_loc) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_loc; i += blockDim.x) A.walk_list[sh_wbase + i] = sh_wl[MODE == 1 ? i : 0];
  }
}

// k_emit_wl: one lane per work item of the general class (written by k_project1): the item names the alignment, the
// candidate's slab row and strand, its match slot and its rank; what is left is the walk (pass 2), the ideal CIGAR, the
// merge with the real CIGAR and the match record -- k_emit_dense<false, 2> without its mask / range / rank stage.
__global__ void __launch_bounds__(256, 6) k_emit_wl(ProjectArgs A, int64_t n_entries) {
  __shared__ uint32_t sh_cig[256 * LDS_SLOT];
  __shared__ uint16_t sh_mops[256];
  sh_mops[threadIdx.x] = (uint16_t)((merge_action(threadIdx.x >> 4, threadIdx.x & 15u) << 8) | merge_ops(threadIdx.x >> 4, threadIdx.x & 15u));
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_entries) return;
  const uint4 item = A.wl[i];
  if (item.x == 0xffffffffu) return;
  const DevIndex &ix = A.ix;
  const DevCfg &cfg = A.cfg;
  const uint32_t a = item.x, row = item.y & 0x7fffffffu, rank = item.w;
  const int s = (int)(item.y >> 31);
  const uint4 hd = A.head[a];
  const uint4 hd2 = A.head2[a];
  const uint32_t c0 = A.cigar_off[a], c1 = A.cigar_off[a + 1];
  const uint4 r_a = ix.s_row[2 * (size_t)row], r_b = ix.s_row[2 * (size_t)row + 1];
  const uint32_t n_seg = hd.z, rid = hd.w >> 2;
  const uint2 q0 = make_uint2(hd.x, hd.y);
  const int st0 = (n_seg == 1) ? ST_ONLY : ST_FIRST;
  RealCig rc;
  rc.fetch(A.cigar + c0, c1 - c0);
  const uint32_t gs = r_a.x, gend = r_a.y;
  const uint4 pay = make_uint4(r_b.x, r_b.y, r_a.w, r_b.z);
  Hit h0;
  classify(s == 1, st0, q0.x, q0.y, gs, gend, pay.z, cfg, h0);
  ReadCtx rd;
  rd.n_seg = n_seg; rd.seg = A.seg + (size_t)c0 + (size_t)a; rd.real = A.cigar + c0; rd.n_real = c1 - c0; rd.q12 = hd2;
  const uint4 *E = ix.tx_ex + pay.w;
  const uint32_t i0 = pay.y & 0x7fffffffu;
  CandOut p1;
  bool plain = false;
  const uint32_t sb = ix.slab_off[2 * rid + s], se = ix.slab_off[2 * rid + s + 1];
  if (n_seg == 1) { p1.alive = true; p1.fwpos = h0.pos; p1.rcpos = h0.pos; p1.n_seg = 1; p1.n_gex = 1; p1.i_lastm = i0; p1.last_right_ins = h0.right_ins; p1.last_right_gap = h0.right_gap; }
  else if (!cfg.long_reads && !cfg.ignore_small_exons) {
    plain = true;   // see k_emit_dense: pass 1's counts are the read's exon count
    p1.alive = true; p1.fwpos = h0.pos; p1.rcpos = h0.pos; p1.n_seg = n_seg; p1.n_gex = n_seg; p1.i_lastm = i0;
    p1.last_right_ins = 0; p1.last_right_gap = 0;
  }
  else p1 = walk_pass1(ix, cfg, rd, E, s == 1, sb, se, i0, make_uint4(gs, gend, pay.z, 0), make_uint2(r_a.z, r_b.w), q0, h0);
  const uint32_t ideal_cap = 4u * n_seg + 2u;
  const uint32_t cap = rd.n_real + 2u * ideal_cap;
  uint32_t *lds = &sh_cig[threadIdx.x * LDS_SLOT];
  const bool ideal_lds = ideal_cap <= LDS_IDEAL;
  // the arena slot (cig_base[a] + rank * cap) is looked up only by the lanes that need it
  uint32_t *slot = nullptr; uint64_t slot_off = 0;
  if (!ideal_lds) { slot_off = A.cig_base[a] + (uint64_t)rank * cap; slot = A.cig_arena + slot_off; }
  uint32_t *ideal = ideal_lds ? lds : slot + rd.n_real + ideal_cap;
  Acc acc; IdealSink sk;
  sk.init(ideal);
  walk_pass2(ix, cfg, rd, E, s == 1, sb, se, i0, q0, make_uint4(gs, gend, pay.z, 0), make_uint2(r_a.z, r_b.w), h0, p1, acc, sk, no_clip(), no_clip());
  const uint32_t n_ideal = sk.finish();
  const bool out_lds = ideal_lds && (rd.n_real + n_ideal <= LDS_SLOT - LDS_IDEAL);
  if (!out_lds && !slot) { slot_off = A.cig_base[a] + (uint64_t)rank * cap; slot = A.cig_arena + slot_off; }
  uint32_t *outp = out_lds ? lds + LDS_IDEAL : slot;
  const uint32_t n_out = merge_cigars(rc, rd.n_real, ideal, n_ideal, outp, sh_mops);
  uint64_t cig_ref;
  if (n_out <= 2) cig_ref = (uint64_t)(n_out > 0 ? outp[0] : 0u) | ((uint64_t)(n_out > 1 ? outp[1] : 0u) << 32);
  else {
    if (!slot) { slot_off = A.cig_base[a] + (uint64_t)rank * cap; slot = A.cig_arena + slot_off; }
    if (out_lds) for (uint32_t q = 0; q < n_out; q++) slot[q] = outp[q];
    cig_ref = slot_off;
  }
  const uint32_t mo = item.z;
  A.m_tid[mo] = pay.x;
  A.m_p[mo] = make_uint2((s == 0) ? p1.fwpos : (plain ? acc.last_pos : p1.rcpos), n_out | ((uint32_t)s << 31));
  A.m_x[mo] = make_uint2((uint32_t)acc.junc_hits, (uint32_t)acc.ref_consumed);
  A.m_cigoff[mo] = cig_ref;
}

// ---------------------------------------------------------------------------
// exclusive scans.  Tile = 256 threads x 8 items.
// ---------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (256 * SCAN_ITEMS)

__device__ __forceinline__ uint64_t block_excl_scan_256(uint64_t v, uint64_t *sh, uint64_t &block_total) {
  // wave-level inclusive scan by shuffles, then across the 4 waves through LDS
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  uint64_t wbase = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { if (i < w) wbase += sh[i]; tot += sh[i]; }
  __syncthreads();
  block_total = tot;
  return wbase + x - v;
}

// value of element i: MODE 0: n_matches[i]; MODE 1: n_matches[i] * cap(i); MODE 2: src32[i]
template <int MODE>
__device__ __forceinline__ uint64_t scan_value(const ScanArgs &S, int64_t i) {
  if (MODE == 0 || MODE == 2) return S.src32[i];
  uint32_t nm = S.src32[i];
  if (nm == 0) return 0;
  uint32_t n_real = S.cigar_off[i + 1] - S.cigar_off[i];
  if (MODE == 3) return (uint64_t)nm * (uint64_t)(n_real + 2u * S.ideal_cap[i]);
  uint32_t ideal_cap = 4u * S.head[i].z + 2u;
  return (uint64_t)nm * (uint64_t)(n_real + 2u * ideal_cap);
}

// A thread's SCAN_ITEMS consecutive 32-bit inputs as two 16-byte loads (base is a multiple of 8 items = 32 bytes; the
// arrays are allocations or 16-byte aligned offsets into one).  One 4-byte load per item made a wave touch every
// eighth word of a 2 KB span eight times over.
__device__ __forceinline__ void load8(const uint32_t *p, int64_t base, int64_t n, uint32_t v[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && ((uintptr_t)(p + base) & 15u) == 0) {
    const uint4 a = *(const uint4 *)(p + base), b = *(const uint4 *)(p + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) v[k] = (base + k < n) ? p[base + k] : 0u;
  }
}
__device__ __forceinline__ void store8(uint32_t *p, int64_t base, int64_t n, const uint32_t v[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && ((uintptr_t)(p + base) & 15u) == 0) {
    *(uint4 *)(p + base) = make_uint4(v[0], v[1], v[2], v[3]); *(uint4 *)(p + base + 4) = make_uint4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) p[base + k] = v[k];
  }
}
__device__ __forceinline__ void store8(uint64_t *p, int64_t base, int64_t n, const uint64_t v[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && ((uintptr_t)(p + base) & 15u) == 0) {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k += 2) *(ulonglong2 *)(p + base + k) = make_ulonglong2(v[k], v[k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) p[base + k] = v[k];
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_scan_tiles(ScanArgs S) {
  __shared__ uint64_t sh[4];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t sum = 0;
  if (MODE == 0 || MODE == 2) {
    uint32_t v[SCAN_ITEMS];
    load8(S.src32, base, S.n, v);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) sum += v[k];
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { int64_t i = base + k; if (i < S.n) sum += scan_value<MODE>(S, i); }
  }
  uint64_t tot;
  block_excl_scan_256(sum, sh, tot);
  if (threadIdx.x == 0) S.tile_sums[blockIdx.x] = tot;
}

// The tile sums of a scan, scanned in place by ONE block: eight consecutive sums per thread and round (2048 per round: a
// handful of rounds for 10^4 tiles, the loads of a round in flight together) -- one sum per thread and round was a chain
// of n_tiles / 256 load-scan-store rounds, 30 us for the rows' scan and 80 us for the fused three-value one.
template <int C>
__device__ __forceinline__ void scan_top_rounds(uint64_t *tile_sums, int64_t n_tiles, uint64_t *total_out, uint64_t *sh) {
  uint64_t carry[C];
#pragma unroll
  for (int c = 0; c < C; c++) carry[c] = 0;
  for (int64_t base = 0; base < n_tiles; base += 256 * 8) {
    const int64_t i0 = base + (int64_t)threadIdx.x * 8;
    uint64_t v[C][8];
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
      for (int k = 0; k < 8; k++) v[c][k] = i0 + k < n_tiles ? tile_sums[(int64_t)c * n_tiles + i0 + k] : 0;
#pragma unroll
    for (int c = 0; c < C; c++) {
      uint64_t sum = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) sum += v[c][k];
      uint64_t tot;
      uint64_t ex = carry[c] + block_excl_scan_256(sum, sh, tot);
#pragma unroll
      for (int k = 0; k < 8; k++) { if (i0 + k < n_tiles) tile_sums[(int64_t)c * n_tiles + i0 + k] = ex; ex += v[c][k]; }
      carry[c] += tot;
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < C; c++) total_out[c] = carry[c];
  }
}
__global__ void __launch_bounds__(256) k_scan_top(uint64_t *tile_sums, int64_t n_tiles, uint64_t *total_out) {
  __shared__ uint64_t sh[4];
  scan_top_rounds<1>(tile_sums, n_tiles, total_out, sh);
}

template <int MODE, typename OutT>
__global__ void __launch_bounds__(256) k_scan_apply(ScanArgs S, OutT *out) {
  __shared__ uint64_t sh[4];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS];
  uint64_t sum = 0;
  if (MODE == 0 || MODE == 2) {
    uint32_t w[SCAN_ITEMS];
    load8(S.src32, base, S.n, w);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = w[k]; sum += v[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { int64_t i = base + k; v[k] = i < S.n ? scan_value<MODE>(S, i) : 0; sum += v[k]; }
  }
  uint64_t tot;
  uint64_t ex = block_excl_scan_256(sum, sh, tot) + S.tile_sums[blockIdx.x];
  OutT o[SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = (OutT)ex; ex += v[k]; }
  store8(out, base, S.n, o);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[S.n] = (OutT)ex;
}

// Fused scan of the count pass: per alignment (n_matches, n_matches * CIGAR slot
// capacity, n_matches of simple alignments) -> match_off (u32), cig_base (u64),
// fast_pre (u32).  One read of the inputs instead of three.
struct Scan3 { uint64_t v[3]; };
__device__ __forceinline__ Scan3 scan3_value(const ScanArgs &S, int64_t i) {
  Scan3 r; r.v[0] = r.v[1] = r.v[2] = 0;
  uint32_t nm = S.src32[i];
  if (nm) {
    uint32_t cf = S.fast_flag[i];
    uint32_t cap = cf & 0x7fffffffu;
    if (S.ideal_cap) cap = (S.cigar_off[i + 1] - S.cigar_off[i]) + 2u * S.ideal_cap[i];  // -S path: clip ops included
    r.v[0] = nm; r.v[1] = (uint64_t)nm * (uint64_t)cap; r.v[2] = (cf >> 31) ? nm : 0;
  }
  return r;
}
// a thread's SCAN_ITEMS values: vector loads of n_matches and the class word unless the -S capacities are in play
__device__ __forceinline__ void scan3_load(const ScanArgs &S, int64_t base, Scan3 v[SCAN_ITEMS]) {
  if (!S.ideal_cap) {
    uint32_t nm[SCAN_ITEMS], cf[SCAN_ITEMS];
    load8(S.src32, base, S.n, nm); load8(S.fast_flag, base, S.n, cf);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      v[k].v[0] = nm[k]; v[k].v[1] = (uint64_t)nm[k] * (uint64_t)(cf[k] & 0x7fffffffu); v[k].v[2] = (cf[k] >> 31) ? nm[k] : 0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      int64_t i = base + k;
      if (i < S.n) v[k] = scan3_value(S, i); else { v[k].v[0] = v[k].v[1] = v[k].v[2] = 0; }
    }
  }
}
__global__ void __launch_bounds__(256) k_scan3_tiles(ScanArgs S) {
  __shared__ uint64_t sh[4];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t sum[3] = {0, 0, 0};
  Scan3 v[SCAN_ITEMS];
  scan3_load(S, base, v);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { sum[0] += v[k].v[0]; sum[1] += v[k].v[1]; sum[2] += v[k].v[2]; }
#pragma unroll
  for (int c = 0; c < 3; c++) { uint64_t tot; block_excl_scan_256(sum[c], sh, tot); if (threadIdx.x == 0) S.tile_sums[(int64_t)c * S.n_tiles + blockIdx.x] = tot; }
}
__global__ void __launch_bounds__(256) k_scan3_top(uint64_t *tile_sums, int64_t n_tiles, uint64_t *total_out) {
  __shared__ uint64_t sh[4];
  scan_top_rounds<3>(tile_sums, n_tiles, total_out, sh);
}
__global__ void __launch_bounds__(256) k_scan3_apply(ScanArgs S, uint32_t *match_off, uint64_t *cig_base, uint32_t *fast_pre) {
  __shared__ uint64_t sh[4];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  Scan3 v[SCAN_ITEMS];
  uint64_t sum[3] = {0, 0, 0};
  scan3_load(S, base, v);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { sum[0] += v[k].v[0]; sum[1] += v[k].v[1]; sum[2] += v[k].v[2]; }
  uint64_t ex[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { uint64_t tot; ex[c] = block_excl_scan_256(sum[c], sh, tot) + S.tile_sums[(int64_t)c * S.n_tiles + blockIdx.x]; }
  uint32_t o0[SCAN_ITEMS], o2[SCAN_ITEMS]; uint64_t o1[SCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    o0[k] = (uint32_t)ex[0]; o1[k] = ex[1]; o2[k] = (uint32_t)ex[2];
    ex[0] += v[k].v[0]; ex[1] += v[k].v[1]; ex[2] += v[k].v[2];
  }
  store8(match_off, base, S.n, o0); store8(cig_base, base, S.n, o1); store8(fast_pre, base, S.n, o2);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) { match_off[S.n] = (uint32_t)ex[0]; cig_base[S.n] = ex[1]; fast_pre[S.n] = (uint32_t)ex[2]; }
}

// Small inputs (at most SCAN_SMALL_TILES tiles): the whole scan by ONE block, tile after tile with a running carry -- one
// launch instead of three (tile sums, their scan, apply): what a batch of a few thousand alignments spends its time on is
// launches, not bytes.
#define SCAN_SMALL_TILES 4
#define EXPAND_SMALL_N 1024   // the scanning block writes the work list too up to this many alignments (four chunks of its 256 threads)
template <int MODE, typename OutT>
__global__ void __launch_bounds__(256) k_scan_small(ScanArgs S, OutT *out, uint64_t *total_out) {
  __shared__ uint64_t sh[4];
  uint64_t carry = 0;
  for (int64_t t0 = 0; t0 < S.n || t0 == 0; t0 += SCAN_TILE) {
    const int64_t base = t0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t sum = 0;
    if (MODE == 0 || MODE == 2) {
      uint32_t w[SCAN_ITEMS];
      load8(S.src32, base, S.n, w);
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = w[k]; sum += v[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < SCAN_ITEMS; k++) { int64_t i = base + k; v[k] = i < S.n ? scan_value<MODE>(S, i) : 0; sum += v[k]; }
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_256(sum, sh, tot) + carry;
    OutT o[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { o[k] = (OutT)ex; ex += v[k]; }
    store8(out, base, S.n, o);
    carry += tot;
  }
  if (threadIdx.x == 0) { out[S.n] = (OutT)carry; total_out[0] = carry; }
}
// EXPAND: the emit work list too (k_expand's job), by the same block once its offsets are in place
template <bool EXPAND>
__global__ void __launch_bounds__(256) k_scan3_small(ScanArgs S, uint32_t *match_off, uint64_t *cig_base, uint32_t *fast_pre, uint64_t *total_out,
                                                     ProjectArgs A) {
  __shared__ uint64_t sh[4];
  __shared__ uint32_t sh_pre[EXPAND ? 4 : 1][64], sh_v[EXPAND ? 4 : 1][64];
  uint64_t carry[3] = {0, 0, 0};
  for (int64_t t0 = 0; t0 < S.n || t0 == 0; t0 += SCAN_TILE) {
    const int64_t base = t0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    Scan3 v[SCAN_ITEMS];
    uint64_t sum[3] = {0, 0, 0};
    scan3_load(S, base, v);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { sum[0] += v[k].v[0]; sum[1] += v[k].v[1]; sum[2] += v[k].v[2]; }
    uint64_t ex[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { uint64_t tot; ex[c] = block_excl_scan_256(sum[c], sh, tot) + carry[c]; carry[c] += tot; }
    uint32_t o0[SCAN_ITEMS], o2[SCAN_ITEMS]; uint64_t o1[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      o0[k] = (uint32_t)ex[0]; o1[k] = ex[1]; o2[k] = (uint32_t)ex[2];
      ex[0] += v[k].v[0]; ex[1] += v[k].v[1]; ex[2] += v[k].v[2];
    }
    store8(match_off, base, S.n, o0); store8(cig_base, base, S.n, o1); store8(fast_pre, base, S.n, o2);
  }
  if (threadIdx.x == 0) {
    match_off[S.n] = (uint32_t)carry[0]; cig_base[S.n] = carry[1]; fast_pre[S.n] = (uint32_t)carry[2];
    total_out[0] = carry[0]; total_out[1] = carry[1]; total_out[2] = carry[2];
  }
  if (EXPAND) {
    if (carry[0] > A.lim_m || carry[1] > A.lim_c) return;   // (every thread holds the totals)
    __threadfence_block();
    __syncthreads();                                        // the offsets above are read below, by other threads
    for (int64_t base = 0; base < S.n; base += 256) expand_chunk(A, base + threadIdx.x, sh_pre, sh_v);
  }
}

// ---------------------------------------------------------------------------
// k_pair<EMIT>: one lane per read-name group.
//   src/core.cpp:343-426 (which alignments pair up), src/mates.cpp:150-261
//   (transcript-set cases), src/core.cpp:237-258,309-325 (NH, HI, MAPQ),
//   src/bam.cpp:531-588 (mate fields).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mapq_of(uint32_t nh, bool long_reads) {  // src/core.cpp:46-58
  if (!long_reads) { return nh == 1 ? 255u : nh == 2 ? 3u : (nh == 3 || nh == 4) ? 1u : 0u; }
  return nh > 1 ? 0u : 3u;
}

// k_group_ids: one lane per read-name group labels its alignments
__global__ void __launch_bounds__(256) k_group_ids(int64_t n_groups, const uint32_t *__restrict__ group_off,
                                                   uint32_t *__restrict__ aln_group) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  for (uint32_t i = group_off[g]; i < group_off[g + 1]; i++) aln_group[i] = (uint32_t)g;
}

// k_pair<EMIT>: one lane per alignment.  A "leader" (no mate, or its mate comes
// later in the group) emits for itself and its mate, exactly the calls
// convert_reads makes (src/core.cpp:384-415).  EMIT=false counts the records,
// EMIT=true writes {match, input, NH, HI | flags} per record at the scanned
// offset: NH = records of the read name (flush, src/core.cpp:250-258), HI = 1-based rank among them (:309-325).
// k_rows turns those into the packed rows, one lane per record.
#define PAIR_TIDS 1024   // transcript ids of a wave's 64 alignments staged in LDS (count pass)
template <bool EMIT>
__global__ void __launch_bounds__(256) k_pair(PairArgs P) {
  int64_t i64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tot_over(P.tot, P.lim_m, P.lim_c)) { if (!EMIT && i64 < P.n_aln) { P.n_rows[i64] = 0; P.pbit[i64] = 0; } return; }
  // count pass: the merge of two mates' tid lists is a chain of dependent loads, two per step.  The lists of a wave's 64
  // alignments are one contiguous piece of m_tid (match_off is a prefix sum) and mates sit next to each other, so the wave
  // copies that piece to LDS first (coalesced) and the chains run on LDS latency; lists outside the piece, or a piece that
  // does not fit, are read from memory as before.
  __shared__ uint32_t sh_tid[EMIT ? 1 : 4][EMIT ? 1 : PAIR_TIDS];
  uint32_t t0 = 0, t_n = 0;
  const int wv_ = EMIT ? 0 : (int)(threadIdx.x >> 6);
  if (!EMIT) {
    const int lane = threadIdx.x & 63;
    const int64_t w0 = i64 - lane;
    if (w0 < P.n_aln) {
      // the window [t0, t0 + t_n): from the first non-empty list of the wave to the end of the last list that still ends
      // inside PAIR_TIDS entries.  (Scanned offsets ascend over the wave; with the single-pass placement they ascend until
      // the producing wave changed its page -- lists beyond that point fail the test and are read from memory.)
      const uint32_t ni_ = i64 < P.n_aln ? P.n_matches[i64] : 0u;
      const uint32_t mo_ = ni_ ? P.match_off[i64] : 0u;
      const uint64_t have = __ballot(ni_ != 0u);
      if (have) {
        t0 = __shfl(mo_, (int)__builtin_ctzll(have));
        const uint32_t e = mo_ - t0 + ni_;
        const uint64_t okm = __ballot(ni_ != 0u && mo_ >= t0 && e <= PAIR_TIDS);
        t_n = __shfl(e, 63 - (int)__builtin_clzll(okm));   // (the first non-empty lane always passes when its list fits)
        if (!okm) t_n = 0;
        for (uint32_t k = lane; k < t_n; k += 64) sh_tid[wv_][k] = P.m_tid[t0 + k];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  auto tid_at = [&](uint32_t idx) -> uint32_t { return (!EMIT && idx - t0 < t_n) ? sh_tid[wv_][idx - t0] : P.m_tid[idx]; };
  if (i64 >= P.n_aln) return;
  uint32_t i = (uint32_t)i64;
  uint32_t rows = 0;
  uint64_t r0 = 0;
  if (EMIT) { r0 = P.row_off[i]; if (P.row_off[i + 1] == r0) return; }   // not a leader, or nothing to emit
  int32_t m = P.mate_idx[i];
  uint32_t g = P.aln_group[i];
  uint32_t a0 = P.group_off[g], a1 = P.group_off[g + 1];
  uint32_t mi0 = P.match_off[i], ni = P.n_matches[i];
  bool leader = !(m >= 0 && (uint32_t)m < i && (uint32_t)m >= a0);  // else: handled as the mate of an earlier leader
  uint32_t nh = 0, hi0 = 0;
  bool unpaired = true;
  uint4 *__restrict__ rec = P.r_rec;
  if (EMIT) {
    const uint64_t gs = P.row_off[a0];
    const uint64_t gn = P.row_off[a1] - gs;
    nh = (uint32_t)gn; hi0 = (uint32_t)(r0 - gs) + 1u;
    if (gn > (uint64_t)RR_HI) P.counters[3] = 1;   // HI does not fit its 28 bits
  }
  if (leader && ni) {                          // a leader without matches drops the pair (mates.cpp:153)
    uint32_t nm = 0, mm0 = 0;
    if (m >= 0 && (uint32_t)m > i && (uint32_t)m < a1) { mm0 = P.match_off[m]; nm = P.n_matches[m]; }
    unpaired = nm == 0;
    if (nm == 0) {
      // unpaired emission: one record per transcript, ascending tid (mates.cpp:157-176)
      if (EMIT)
        for (uint32_t k = 0; k < ni; k++) rec[r0 + k] = make_uint4(mi0 + k, i, nh, (hi0 + k) | RR_FIRST);
      rows = ni;
    } else {
      // both mates matched: sorted-set intersection (mates.cpp:204-231).  The count pass does the merge and leaves, per
      // mate, the bit set of its list positions that are common (both lists ascend, so the k-th common position of one
      // pairs with the k-th of the other); the emit pass only walks those bits -- no second merge over m_tid[].
      uint32_t x = 0, y = 0, common = 0;
      const bool masked = ni <= 64u && nm <= 64u;
      if (EMIT && masked) {
        uint64_t ma = P.pmask[i], mb = P.pmask[m];
        while (ma) {
          uint32_t xx = (uint32_t)__builtin_ctzll(ma), yy = (uint32_t)__builtin_ctzll(mb);
          ma &= ma - 1; mb &= mb - 1;
          uint64_t r = r0 + 2ull * common;
          rec[r] = make_uint4(mi0 + xx, i, nh, (hi0 + 2u * common) | RR_FIRST | RR_PAIRED | RR_SAME);
          rec[r + 1] = make_uint4(mm0 + yy, (uint32_t)m, nh, (hi0 + 2u * common + 1u) | RR_PAIRED | RR_SAME);
          common++;
        }
        x = ni;  // skip the merge
      }
      uint64_t bits_a = 0, bits_b = 0;
      while (x < ni && y < nm) {
        uint32_t tx = tid_at(mi0 + x), ty = tid_at(mm0 + y);
        if (tx < ty) x++;
        else if (ty < tx) y++;
        else {
          if (EMIT) {
            uint64_t r = r0 + 2ull * common;
            rec[r] = make_uint4(mi0 + x, i, nh, (hi0 + 2u * common) | RR_FIRST | RR_PAIRED | RR_SAME);
            rec[r + 1] = make_uint4(mm0 + y, (uint32_t)m, nh, (hi0 + 2u * common + 1u) | RR_PAIRED | RR_SAME);
          } else if (masked) { bits_a |= 1ull << x; bits_b |= 1ull << y; }
          common++; x++; y++;
        }
      }
      if (!EMIT && masked) { P.pmask[i] = bits_a; P.pmask[m] = bits_b; }
      if (common) rows = 2 * common;
      else if (ni == 1 && nm == 1) {  // one transcript each, different ones
        if (EMIT) {
          rec[r0] = make_uint4(mi0, i, nh, hi0 | RR_FIRST | RR_PAIRED);
          rec[r0 + 1] = make_uint4(mm0, (uint32_t)m, nh, (hi0 + 1u) | RR_PAIRED);
        }
        rows = 2;
      }
    }
  }
  if (!EMIT) { P.n_rows[i] = rows; P.pbit[i] = (rows && !unpaired) ? 1 : 0; }
}

// k_pair_emit: the emit pass of k_pair with the records written by the whole wave.  A wave takes 64 alignments; the
// records of its leaders are one contiguous run of the row order (row_off is a prefix sum over the alignments), so
//   1. lane i describes leader i (case, first match of either mate, NH, first HI, the two sets of common list positions
//      the count pass left) in LDS,
//   2. every lane takes one record of the run per round: record e -> leader (search over the wave's count prefixes) ->
//      the k-th record of that leader, computed from the description (the c-th common pair = the c-th set bits of the two
//      position sets), one 16-byte store, coalesced,
// instead of every leader lane writing its 1..2 x 64 records one after the other.  Leaders whose lists exceed the 64-bit
// position sets (rare) write their records themselves with the merge loop, as k_pair<true> does.
struct PairLead { uint32_t pre, kind, mi0, mm0, m, nh, hi0, pad; uint64_t pa, pb; };   // kind: 0 none, 1 unpaired, 2 common pairs, 3 one pair of different transcripts, 4 merge loop
__global__ void __launch_bounds__(256) k_pair_emit(PairArgs P) {
  __shared__ PairLead sh_l[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = (uint32_t)i64;
  if (tot_over(P.tot, P.lim_m, P.lim_c) || rows_over(P.tot, P.lim_r)) return;
  uint4 *__restrict__ rec = P.r_rec;
  uint32_t rows = 0, kind = 0, mi0 = 0, mm0 = 0, ni = 0, nm = 0, nh = 0, hi0 = 0;
  int32_t m = -1;
  uint64_t r0 = 0, pa = 0, pb = 0;
  if (i64 < P.n_aln) {
    r0 = P.row_off[i];
    rows = (uint32_t)(P.row_off[i + 1] - r0);
  }
  if (rows) {   // a leader with records (the count pass: leader && ni, src/mates.cpp:153)
    m = P.mate_idx[i];
    const uint32_t g = P.aln_group[i];
    const uint32_t a0 = P.group_off[g], a1 = P.group_off[g + 1];
    mi0 = P.match_off[i]; ni = P.n_matches[i];
    const uint64_t gs = P.row_off[a0], gn = P.row_off[a1] - gs;
    nh = (uint32_t)gn; hi0 = (uint32_t)(r0 - gs) + 1u;
    if (gn > (uint64_t)RR_HI) P.counters[3] = 1;   // HI does not fit its 28 bits
    if (m >= 0 && (uint32_t)m > i && (uint32_t)m < a1) { mm0 = P.match_off[m]; nm = P.n_matches[m]; }
    if (nm == 0) kind = 1;
    else if (ni <= 64u && nm <= 64u) { pa = P.pmask[i]; pb = P.pmask[m]; kind = pa ? 2u : 3u; }
    else kind = 4;
  }
  // the wave's record run
  uint32_t inc = rows;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(inc, d); if (lane >= d) inc += up; }
  const uint32_t total = __shfl(inc, 63);
  const uint64_t have = __ballot(rows != 0u);
  if (!have) return;   // the same for the whole wave
  const int first_lane = (int)__builtin_ctzll(have);
  const uint64_t run0 = ((uint64_t)__shfl((uint32_t)(r0 >> 32), first_lane) << 32) | __shfl((uint32_t)r0, first_lane);
  PairLead &L = sh_l[wv][lane];
  L.pre = inc - rows; L.kind = kind; L.mi0 = mi0; L.mm0 = mm0; L.m = (uint32_t)m; L.nh = nh; L.hi0 = hi0; L.pa = pa; L.pb = pb;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const uint32_t i_base = i - (uint32_t)lane;
  for (uint32_t e = lane; e < total; e += 64) {
    int l = 0;
#pragma unroll
    for (int st = 32; st; st >>= 1) if (sh_l[wv][l + st].pre <= e) l += st;
    const PairLead &D = sh_l[wv][l];
    const uint32_t k = e - D.pre, kd = D.kind;
    if (kd == 4u) continue;   // written by its leader below
    uint4 v;
    if (kd == 1u) {
      // unpaired emission: one record per transcript, ascending tid (mates.cpp:157-176)
      v = make_uint4(D.mi0 + k, i_base + (uint32_t)l, D.nh, (D.hi0 + k) | RR_FIRST);
    } else if (kd == 2u) {
      // both mates matched, the c-th common transcript (mates.cpp:204-231): the c-th set bit of either position set
      const uint32_t c = k >> 1, second = k & 1u;
      uint64_t bits = second ? D.pb : D.pa;
      for (uint32_t j = 0; j < c; j++) bits &= bits - 1;
      const uint32_t p = (uint32_t)__builtin_ctzll(bits);
      v = second ? make_uint4(D.mm0 + p, D.m, D.nh, (D.hi0 + k) | RR_PAIRED | RR_SAME)
                 : make_uint4(D.mi0 + p, i_base + (uint32_t)l, D.nh, (D.hi0 + k) | RR_FIRST | RR_PAIRED | RR_SAME);
    } else {
      // one transcript each, different ones
      v = k ? make_uint4(D.mm0, D.m, D.nh, (D.hi0 + 1u) | RR_PAIRED)
            : make_uint4(D.mi0, i_base + (uint32_t)l, D.nh, D.hi0 | RR_FIRST | RR_PAIRED);
    }
    rec[run0 + e] = v;
  }
  if (kind == 4u) {   // lists beyond the 64-bit position sets: the merge itself
    uint32_t x = 0, y = 0, common = 0;
    while (x < ni && y < nm) {
      const uint32_t tx = P.m_tid[mi0 + x], ty = P.m_tid[mm0 + y];
      if (tx < ty) x++;
      else if (ty < tx) y++;
      else {
        const uint64_t r = r0 + 2ull * common;
        rec[r] = make_uint4(mi0 + x, i, nh, (hi0 + 2u * common) | RR_FIRST | RR_PAIRED | RR_SAME);
        rec[r + 1] = make_uint4(mm0 + y, (uint32_t)m, nh, (hi0 + 2u * common + 1u) | RR_PAIRED | RR_SAME);
        common++; x++; y++;
      }
    }
  }
}

// k_primary: one lane per read name (grid-stride).  Primary = the emitted record (pair) with the
// best similarity score; ties are broken by get_rand(n_tied, std::hash(name))
// (src/core.cpp:243-307), restated in primary_pick.h.  A leader's records are all paired or all
// single, so its units follow from its record count and one flag word.
// SCORES = false: presets without the similarity filter leave every score at 0.0 (src/evaluate.cpp:843-865), so
// every emitted unit ties and the scores need not be read at all.
// Also the counters of src/bramble.cpp:729-736 (one atomic per wave at the end).
template <bool SCORES>
__global__ void __launch_bounds__(256) k_primary(PairArgs P, const uint32_t *__restrict__ name_off,
                                                 const uint8_t *__restrict__ names) {
  unsigned long long uniq = 0, dropped = 0;
  if (tot_over(P.tot, P.lim_m, P.lim_c) || rows_over(P.tot, P.lim_r)) return;
  uint32_t *flagw = (uint32_t *)P.r_rec + 3;   // flag word of record r: flagw[4 * r]
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < P.n_groups; g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t a0 = P.group_off[g], a1 = P.group_off[g + 1];
    const uint64_t rs = P.row_off[a0], re = P.row_off[a1];
    uniq += (re - rs) == 1 ? 1 : 0;
    uint32_t any = 0;
    for (uint32_t i = a0; i < a1; i++) any |= P.n_matches[i];
    dropped += any ? 0 : 1;
    if (!SCORES && P.pick) P.pick[g] = ~0ull;     // no primary record (no names / no records): rewritten below otherwise
    if (!names || re == rs) continue;
    uint64_t pick = rs;
    if (!SCORES) {
      if (re - rs > 1) {
        // units per leader = its record count, halved when they are pairs (the flag of its first record): a chain of
        // three dependent stages (group -> row offsets -> flags) instead of one load per unit.  Up to four alignments
        // per name (nearly every group) with every stage's loads issued together.
        const uint32_t na = a1 - a0;
        if (na <= 4u) {
          uint64_t ro[5]; uint32_t fl[4], un[4];
#pragma unroll
          for (int k = 0; k < 5; k++) ro[k] = P.row_off[a0 + ((uint32_t)k < na ? (uint32_t)k : na)];
#pragma unroll
          for (int k = 0; k < 4; k++) fl[k] = P.pbit[a0 + ((uint32_t)k < na ? (uint32_t)k : 0u)] ? RR_PAIRED : 0u;   // the count pass's note, no record gather
          uint32_t units = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) { const uint32_t n = (uint32_t)(ro[k + 1] - ro[k]); un[k] = (fl[k] & RR_PAIRED) ? n / 2 : n; units += un[k]; }
          if (units > 1) {
            uint32_t idx = primary_pick(names + name_off[a0], name_off[a0 + 1] - name_off[a0], units);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const bool here = idx < un[k];
              if (here) pick = ro[k] + ((fl[k] & RR_PAIRED) ? 2ull * idx : idx);
              idx = here ? 0xffffffffu : idx - un[k];   // 0xffffffff: found (no later leader can match)
              un[k] = here ? 0u : un[k];
            }
          }
        } else {
          uint32_t units = 0;
          for (uint32_t i = a0; i < a1; i++) {
            const uint64_t b = P.row_off[i]; const uint32_t n = (uint32_t)(P.row_off[i + 1] - b);
            if (n) units += P.pbit[i] ? n / 2 : n;
          }
          if (units > 1) {
            uint32_t idx = primary_pick(names + name_off[a0], name_off[a0 + 1] - name_off[a0], units);
            for (uint32_t i = a0; i < a1; i++) {
              const uint64_t b = P.row_off[i]; const uint32_t n = (uint32_t)(P.row_off[i + 1] - b);
              if (!n) continue;
              const bool paired = P.pbit[i] != 0;
              const uint32_t u = paired ? n / 2 : n;
              if (idx < u) { pick = b + (paired ? 2ull * idx : idx); break; }
              idx -= u;
            }
          }
        }
      }
    } else {
      auto score_of = [&](uint64_t r) -> double {  // similarity of the record's match
        const uint4 mb = P.m_b[P.r_rec[r].x];
        return __longlong_as_double((long long)(((unsigned long long)mb.w << 32) | mb.z));
      };
      double best = -__builtin_inf(); uint32_t at_best = 0;
      for (uint64_t r = rs; r < re;) {
        const bool paired = flagw[4 * r] & RR_PAIRED;
        double sc = score_of(r);
        if (paired) { double s2 = score_of(r + 1); sc = sc > s2 ? sc : s2; }   // std::max(pair_score, m_align score)
        if (sc > best) { best = sc; pick = r; at_best = 1; } else if (sc == best) at_best++;
        r += paired ? 2 : 1;
      }
      if (at_best > 1) {
        uint32_t idx = primary_pick(names + name_off[a0], name_off[a0 + 1] - name_off[a0], at_best);
        uint32_t seen = 0;
        for (uint64_t r = rs; r < re;) {
          const bool paired = flagw[4 * r] & RR_PAIRED;
          double sc = score_of(r);
          if (paired) { double s2 = score_of(r + 1); sc = sc > s2 ? sc : s2; }
          if (sc == best) { if (seen == idx) { pick = r; break; } seen++; }
          r += paired ? 2 : 1;
        }
      }
    }
    if (!SCORES && P.pick) { P.pick[g] = pick; continue; }   // k_rows sets the bit: no record is touched here (the kernel
                                                              // runs beside the emit pass of k_pair)
    const uint32_t fw = flagw[4 * pick];
    flagw[4 * pick] = fw | RR_PRIMARY;
    if (fw & RR_PAIRED) flagw[4 * (pick + 1)] |= RR_PRIMARY;
  }
  // (one pair of atomics per block: same-address atomics queue up in one L2 channel, see k_group_desc)
  __shared__ unsigned long long sh_pc[4][2];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { uniq += __shfl_down(uniq, d, 64); dropped += __shfl_down(dropped, d, 64); }
  if ((threadIdx.x & 63) == 0) { sh_pc[(threadIdx.x >> 6) & 3][0] = uniq; sh_pc[(threadIdx.x >> 6) & 3][1] = dropped; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long u = 0, dr = 0;
    for (unsigned w = 0; w < (blockDim.x + 63) / 64 && w < 4; w++) { u += sh_pc[w][0]; dr += sh_pc[w][1]; }
    if (u) atomicAdd((unsigned long long *)&P.counters[1], u);
    if (dr) atomicAdd((unsigned long long *)&P.counters[2], dr);
  }
}

// k_rows: one lane per emitted record: r_rec + its match's record -> the packed row (see PairArgs; the detail column
// r_x is k_rows_detail's).  Rewritten
// CIGARs of more than two ops stay in their arena slot; the row carries the slot's offset.
template <bool AUX>
__global__ void __launch_bounds__(256) k_rows(PairArgs P) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (P.tot) { if (tot_over(P.tot, P.lim_m, P.lim_c) || rows_over(P.tot, P.lim_r) || r >= (int64_t)P.tot[3]) return; }   // the record count is on the device
  else if (r >= P.n_rows_total) return;
  const uint4 rec = P.r_rec[r];
  const uint32_t x = rec.x;
  const uint32_t tid = P.m_tid[x];
  const uint2 ma = P.m_p[x];   // {pos, n_cigar | minus << 31}; junc_hits / aligned_len sit in m_x, which only the detail column reads
  const uint64_t cg = P.m_cigoff[x];
  const uint32_t n = ma.y & 0x7fffffffu;
  if (n > RM_NCIG) P.counters[3] = 1;
  const uint32_t f = rec.w;
  bool primary = (f & RR_PRIMARY) != 0;
  if (P.pick) {   // the read name's primary record (pair: the leader's record and the one after it) from k_primary's choice
    const uint64_t pk = P.pick[P.aln_group[rec.y]];
    primary = (uint64_t)r == pk || ((f & RR_PAIRED) && !(f & RR_FIRST) && (uint64_t)r == pk + 1ull);
  }
  const uint32_t meta = (n & RM_NCIG) | ((ma.y >> 31) ? RM_MINUS : 0u) | ((f & RR_PAIRED) ? RM_PAIRED : 0u) |
                        ((f & RR_SAME) ? RM_SAME : 0u) | ((f & RR_FIRST) ? RM_FIRST : 0u) | (primary ? RM_PRIMARY : 0u);
  P.r_a[r] = make_uint4(tid, ma.x, meta, rec.z);
  P.r_c[r] = make_uint2((uint32_t)cg, (uint32_t)(cg >> 32));
  if (P.r_x) { const uint2 mx = P.m_x[x]; P.r_x[r] = make_uint4(rec.y, mx.x, mx.y, rec.w & RR_HI); }   // the detail column at once (small calls that want it)
  if (AUX) {
    const uint4 mb = P.m_b[x];
    P.r_clip[r] = (int32_t)mb.x;
    P.r_sim[r] = __longlong_as_double((long long)(((unsigned long long)mb.w << 32) | mb.z));
  }
}

// k_rows_detail: the detail column br_row_x of the packed rows, on request only (the wide view, host downloads with
// host_detail): everything in it is in r_rec and the match table, which stay valid until the next projection call
__global__ void __launch_bounds__(256) k_rows_detail(PairArgs P) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P.n_rows_total) return;
  const uint4 rec = P.r_rec[r];
  const uint2 mx = P.m_x[rec.x];
  P.r_x[r] = make_uint4(rec.y, mx.x, mx.y, rec.w & RR_HI);
}

// dense pool of the long rewritten CIGARs (host downloads only): sizes -> scan -> copy
__global__ void __launch_bounds__(256) k_pool_sizes(PoolArgs Q) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Q.n_rows) return;
  const uint32_t n = ((const uint32_t *)(Q.r_a + r))[2] & RM_NCIG;
  Q.sizes[r] = n > 2u ? n : 0u;
}
template <int G>
__global__ void __launch_bounds__(256) k_pool_copy(PoolArgs Q) {
  const int lane = threadIdx.x & (G - 1);
  int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (r >= Q.n_rows) return;
  const uint32_t n = ((const uint32_t *)(Q.r_a + r))[2] & RM_NCIG;
  const uint2 c = Q.r_c[r];
  if (n <= 2u) { if (lane == 0) Q.c_out[r] = c; return; }
  const uint64_t d0 = Q.off[r];
  if (lane == 0) Q.c_out[r] = make_uint2((uint32_t)d0, (uint32_t)(d0 >> 32));
  const uint32_t *src = Q.arena + (((uint64_t)c.y << 32) | c.x);
  for (uint32_t k = lane; k < n; k += G) Q.pool[d0 + k] = src[k];
}

// ---------------------------------------------------------------------------
// wide view of the packed rows (br_device_rows / br_rows): one array per field.  Not part of the
// projection itself: run on request (br_device_rows_expand).
// k_wide_fields: one lane per row.  MAPQ of get_mapq, the mate fields of set_mate_info
// (src/bam.cpp:531-588) from the pair's adjacent row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wide_fields(WideArgs W) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= W.n_rows) return;
  const uint4 a = W.r_a[r], x = W.r_x[r];
  const uint32_t meta = a.z;
  const int32_t input = (int32_t)x.x;
  int32_t mate_tid = -1, mate_pos = -1, isize = 0;
  if (meta & RM_PAIRED) {
    const uint4 b = W.r_a[(meta & RM_FIRST) ? r + 1 : r - 1];
    const int32_t my_pos = (int32_t)a.y;
    mate_pos = (int32_t)b.y;
    if (meta & RM_SAME) {
      mate_tid = (int32_t)a.x;
      const int32_t lq = W.l_qseq[input];
      isize = (my_pos <= mate_pos) ? (mate_pos + lq) - my_pos : -((my_pos + lq) - mate_pos);
    } else mate_tid = (int32_t)b.x;
  }
  W.w_input[r] = input; W.w_nh[r] = a.w; W.w_hi[r] = x.w; W.w_mapq[r] = mapq_of(a.w, W.long_reads);
  W.w_group[r] = W.aln_group[input];
  W.w_mate_tid[r] = mate_tid; W.w_mate_pos[r] = mate_pos; W.w_isize[r] = isize;
  W.w_tid[r] = a.x; W.w_pos[r] = a.y; W.w_ncig[r] = meta & RM_NCIG;
  W.w_strand[r] = (meta & RM_MINUS) ? (int8_t)'-' : (int8_t)'+';
  W.w_sim[r] = W.r_sim ? W.r_sim[r] : 0.0; W.w_clip[r] = W.r_clip ? W.r_clip[r] : 0;
  W.w_junc[r] = (int32_t)x.y; W.w_refc[r] = (int32_t)x.z;
  W.w_paired[r] = (meta & RM_PAIRED) ? 1 : 0; W.w_same[r] = (meta & RM_SAME) ? 1 : 0;
  W.w_first[r] = (meta & RM_FIRST) ? 1 : 0; W.w_primary[r] = (meta & RM_PRIMARY) ? 1 : 0;
}

// k_wide_cigars: G lanes per row copy its rewritten CIGAR to the dense per-row pool of the wide view
template <int G>
__global__ void __launch_bounds__(256) k_wide_cigars(WideArgs W) {
  const int lane = threadIdx.x & (G - 1);
  int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  if (r >= W.n_rows) return;
  const uint32_t n = ((const uint32_t *)(W.r_a + r))[2] & RM_NCIG;
  const uint2 c = W.r_c[r];
  const uint64_t d0 = W.w_cigoff[r];
  if (n <= 2) {
    if (lane == 0) {
      if (n > 0) W.w_cigar[d0] = c.x;
      if (n > 1) W.w_cigar[d0 + 1] = c.y;
    }
    return;
  }
  const uint32_t *src = W.pool + (((uint64_t)c.y << 32) | c.x);
  for (uint32_t k = lane; k < n; k += G) W.w_cigar[d0 + k] = src[k];
}

// ---------------------------------------------------------------------------
// k_stats (diagnostic, never timed): exact counters of the algorithmic-bytes
// formula of SURVEY.md 8(d).  One lane per alignment; for every strand tried and
// every read exon: the two key searches (8 * ceil(log2 N_slab) bytes) and 40 bytes
// per overlapping row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stats(StatsArgs T) {
  int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b_idx = 0, hits = 0, exons = 0, ncig = 0;
  if (a < T.n_aln) {
    const uint4 hd = T.head[a];   // {exon 0, n_seg, refid << 2 | strands to try}
    AlnMeta mt; mt.n_seg = hd.z; mt.smode = hd.w & 3u; mt.n_left_clip = mt.n_right_clip = 0;
    uint32_t c0 = T.cigar_off[a];
    ncig = T.cigar_off[a + 1] - c0;
    if (mt.n_seg) {
      const uint2 *seg = T.seg + (size_t)c0 + (size_t)a;
      const uint4 h2 = mt.n_seg > 1 ? T.head2[a] : make_uint4(0, 0, 0, 0);
      uint32_t rid = (uint32_t)T.ref_id[a];
      exons = mt.n_seg;
      for (int s = 0; s < 2; s++) {
        if (!((mt.smode >> s) & 1u)) continue;
        uint32_t sb = T.ix.slab_off[2 * rid + s], se = T.ix.slab_off[2 * rid + s + 1];
        uint32_t N = se - sb, lg = 0;
        while ((1u << lg) < N) lg++;
        for (uint32_t j = 0; j < mt.n_seg; j++) {
          uint2 q = j == 0 ? make_uint2(hd.x, hd.y) : j == 1 ? make_uint2(h2.x, h2.y) : j == 2 ? make_uint2(h2.z, h2.w) : seg[j];   // (k_segment writes seg[] from the fourth exon on)
          uint32_t x = sb, y = se;
          while (x < y) { uint32_t m = (x + y) >> 1; if (T.ix.s_start[m] < q.y) x = m + 1; else y = m; }
          uint32_t hi = x; x = sb; y = hi;
          while (x < y) { uint32_t m = (x + y) >> 1; if (T.ix.s_pmax[m] <= q.x) x = m + 1; else y = m; }
          uint32_t h = 0;
          for (uint32_t r = x; r < hi; r++) h += T.ix.s_row[2 * (size_t)r].y > q.x ? 1u : 0u;
          hits += h; b_idx += 8ull * lg + 40ull * h;
        }
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    b_idx += __shfl_down(b_idx, d, 64); hits += __shfl_down(hits, d, 64);
    exons += __shfl_down(exons, d, 64); ncig += __shfl_down(ncig, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd((unsigned long long *)&T.out[1], b_idx); atomicAdd((unsigned long long *)&T.out[3], ncig);
    atomicAdd((unsigned long long *)&T.out[4], exons); atomicAdd((unsigned long long *)&T.out[5], hits);
  }
}

// (per alignment: with the single-pass placement the match table has unused slots between the waves' pages)
__global__ void __launch_bounds__(256) k_sum_ncig(const uint2 *m_p, const uint32_t *match_off, const uint32_t *n_matches, int64_t n, uint64_t *out) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t nm = n_matches[i], m0 = nm ? match_off[i] : 0u;
    for (uint32_t k = 0; k < nm; k++) acc += m_p[m0 + k].y & 0x7fffffffu;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)out, acc);
}

#ifndef FA_GROUP_LANES
#define FA_GROUP_LANES 8   // lanes per alignment in the -S kernels
#endif
#include "rescue_kernels.inc"
#include "direct_kernels.inc"

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_project_fa(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int mode, int n_blocks) {
  if (A.n_aln <= 0) return;
  constexpr int FG = FA_GROUP_LANES;
  int64_t need = (A.n_aln + (256 / FG) - 1) / (256 / FG);
  if (need < n_blocks) n_blocks = (int)need;
  if (n_blocks < 1) n_blocks = 1;
  dim3 g(n_blocks), b(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL((k_project_fa<0, FG>), g, b, 0, st, A, F); break;
    case 1: hipLaunchKernelGGL((k_project_fa<1, FG>), g, b, 0, st, A, F); break;
    case 2: hipLaunchKernelGGL((k_project_fa<2, FG>), g, b, 0, st, A, F); break;
    default: hipLaunchKernelGGL((k_project_fa<3, FG>), g, b, 0, st, A, F); break;
  }
}

void launch_fa_fill(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int64_t n_prob) {
  if (n_prob <= 0) return;
  int64_t blocks = std::min<int64_t>((n_prob + 15) / 16, 256 * 32);
  hipLaunchKernelGGL(k_fa_fill, dim3((unsigned)blocks), dim3(256), 0, st, A.ix.tx_ex, A.ix.seq_pool, F, n_prob);
}

static inline int grid_for(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

void launch_segment(hipStream_t st, int64_t n_aln, const int32_t *ref_id, const int32_t *ref_start,
                    const uint16_t *flags, const int8_t *xs, const int8_t *ts, const uint32_t *cigar_off,
                    const uint32_t *cigar, const DevCfg &cfg, uint32_t n_refs, uint2 *seg, AlnMeta *meta,
                    uint4 *head, uint4 *head2, uint32_t *fast_flag, const SegExtra *extra) {
  if (n_aln <= 0) return;
  SegExtra X{};
  if (extra) X = *extra;
  hipLaunchKernelGGL(k_segment, dim3(grid_for(n_aln, 256)), dim3(256), 0, st, n_aln, ref_id, ref_start, flags,
                     xs, ts, cigar_off, cigar, cfg, n_refs, seg, meta, head, head2, fast_flag, X);
}

// part (count pass of the presets without the similarity filter, walk_list set): 0 = both kernels, 1 = the main one, 2 = the
// one with the exon walk over the deferred alignments
template <int G>
static void launch_project_g(hipStream_t st, const ProjectArgs &A, bool emit, int n_blocks, int part) {
  bool simf = A.cfg.filter_by_similarity != 0;
  if (emit) {
    if (simf) hipLaunchKernelGGL((k_project<G, true, true>), dim3(n_blocks), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((k_project<G, true, false>), dim3(n_blocks), dim3(256), 0, st, A);
  } else {
    if (simf) hipLaunchKernelGGL((k_project<G, false, true>), dim3(n_blocks), dim3(256), 0, st, A);
    else if (A.walk_list) {
      if (part != 2) hipLaunchKernelGGL((k_project<G, false, false, 1>), dim3(n_blocks), dim3(256), 0, st, A);
      if (part != 1) hipLaunchKernelGGL((k_project<G, false, false, 2>), dim3(n_blocks), dim3(256), 0, st, A);
    } else hipLaunchKernelGGL((k_project<G, false, false>), dim3(n_blocks), dim3(256), 0, st, A);
  }
}

void launch_project(hipStream_t st, const ProjectArgs &A, bool emit, int group_lanes, int n_blocks, int part) {
  if (!emit && A.ix.n_rows == 0) {
    if (part == 2) return;  // empty annotation: nothing can match (and the kernel's clamped loads need one row)
    (void)hipMemsetAsync(A.n_matches, 0, (size_t)A.n_aln * 4, st);
    (void)hipMemsetAsync(A.mask, 0, (size_t)A.n_aln * 8, st);
    (void)hipMemsetAsync(A.ranges, 0, (size_t)A.n_aln * sizeof(uint4), st);
    return;
  }
  if (A.n_aln <= 0) return;
  int64_t groups = A.n_aln;
  int per_block = 256 / group_lanes;
  int64_t need = (groups + per_block - 1) / per_block;
  if (need < n_blocks) n_blocks = (int)need;
  if (n_blocks < 1) n_blocks = 1;
  switch (group_lanes) {
    case 8: launch_project_g<8>(st, A, emit, n_blocks, part); break;
    case 16: launch_project_g<16>(st, A, emit, n_blocks, part); break;
    case 32: launch_project_g<32>(st, A, emit, n_blocks, part); break;
    default: launch_project_g<64>(st, A, emit, n_blocks, part); break;
  }
}

template <int G>
static void launch_project1_g(hipStream_t st, const ProjectArgs &A, int n_blocks, int part) {
  if (part == 1) hipLaunchKernelGGL((k_project1<G, 1>), dim3(n_blocks), dim3(256), 0, st, A);
  else hipLaunchKernelGGL((k_project1<G, 2>), dim3(n_blocks), dim3(256), 0, st, A);
}
void launch_project1(hipStream_t st, const ProjectArgs &A, int group_lanes, int n_blocks, int part) {
  if (A.n_aln <= 0) return;
  if (A.ix.n_rows == 0) {   // empty annotation: nothing can match (and the kernel's clamped loads need one row)
    if (part == 1) (void)hipMemsetAsync(A.n_matches, 0, (size_t)A.n_aln * 4, st);
    return;
  }
  const int64_t need = (A.n_aln + P1_CHUNK - 1) / P1_CHUNK / 4 + 1;   // a wave per chunk
  if (need < n_blocks) n_blocks = (int)need;
  if (n_blocks < 1) n_blocks = 1;
  switch (group_lanes) {
    case 16: launch_project1_g<16>(st, A, n_blocks, part); break;
    default: launch_project1_g<8>(st, A, n_blocks, part); break;
  }
}
void launch_emit_wl(hipStream_t st, const ProjectArgs &A, int64_t n_entries) {
  if (n_entries > 0) hipLaunchKernelGGL(k_emit_wl, dim3(grid_for(n_entries, 256)), dim3(256), 0, st, A, n_entries);
}

void launch_expand(hipStream_t st, const ProjectArgs &A) {
  if (A.n_aln <= 0) return;
  hipLaunchKernelGGL(k_expand, dim3(grid_for(A.n_aln, 256)), dim3(256), 0, st, A);
}

// With A.tot set, n_matches / n_simple are what the GRIDS cover (predictions or capacities); the kernels take the real counts
// from the device and give up when a grid falls short (A.cover).
void launch_emit_dense(hipStream_t st, const ProjectArgs &A0, int64_t n_matches, int64_t n_simple, int part) {
  if (A0.n_aln <= 0 || n_matches <= 0) return;
  ProjectArgs A = A0;
  if (A.tot) A.cover = (uint64_t)(part == 1 ? n_simple : part == 2 ? n_matches - n_simple : n_matches);
  const FaArgs F{};
  if (A.cfg.filter_by_similarity) { hipLaunchKernelGGL((k_emit_dense<true, 0>), dim3(grid_for(n_matches, 256)), dim3(256), 0, st, A, (int64_t)0, n_matches, F); return; }
  if (part == 0 || n_simple < 0) { hipLaunchKernelGGL((k_emit_dense<false, 0>), dim3(grid_for(n_matches, 256)), dim3(256), 0, st, A, (int64_t)0, n_matches, F); return; }
  // the simple prefix of the work list and the rest as two launches: the first needs a third of the registers and no LDS
  if (part == 1) { if (n_simple > 0) hipLaunchKernelGGL((k_emit_dense<false, 1>), dim3(grid_for(n_simple, 256)), dim3(256), 0, st, A, (int64_t)0, n_simple, F); }
  else if (n_matches > n_simple) hipLaunchKernelGGL((k_emit_dense<false, 2>), dim3(grid_for(n_matches - n_simple, 256)), dim3(256), 0, st, A, n_simple, n_matches, F);
}

// the -S rescue's emit pass over the work list (alignments with at most 64 candidate rows; the others: k_project_fa<3>)
void launch_emit_dense_fa(hipStream_t st, const ProjectArgs &A, const FaArgs &F, int64_t n_matches) {
  if (A.n_aln <= 0 || n_matches <= 0) return;
  hipLaunchKernelGGL((k_emit_dense<true, 0, true>), dim3(grid_for(n_matches, 256)), dim3(256), 0, st, A, (int64_t)0, n_matches, F);
}

int64_t scan_tiles_for(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

void launch_scan(hipStream_t st, const ScanArgs &S, int mode, void *out, bool out64, uint64_t *total_out) {
  int64_t tiles = scan_tiles_for(S.n);
  if (tiles < 1) tiles = 1;
  dim3 g((unsigned)tiles), b(256);
  if (tiles <= SCAN_SMALL_TILES) {   // one launch
    const dim3 one(1);
    if (mode == 1) hipLaunchKernelGGL((k_scan_small<1, uint64_t>), one, b, 0, st, S, (uint64_t *)out, total_out);
    else if (mode == 3) hipLaunchKernelGGL((k_scan_small<3, uint64_t>), one, b, 0, st, S, (uint64_t *)out, total_out);
    else if (out64) hipLaunchKernelGGL((k_scan_small<2, uint64_t>), one, b, 0, st, S, (uint64_t *)out, total_out);
    else hipLaunchKernelGGL((k_scan_small<2, uint32_t>), one, b, 0, st, S, (uint32_t *)out, total_out);
    return;
  }
  if (mode == 0) hipLaunchKernelGGL((k_scan_tiles<0>), g, b, 0, st, S);
  else if (mode == 1) hipLaunchKernelGGL((k_scan_tiles<1>), g, b, 0, st, S);
  else if (mode == 3) hipLaunchKernelGGL((k_scan_tiles<3>), g, b, 0, st, S);
  else hipLaunchKernelGGL((k_scan_tiles<2>), g, b, 0, st, S);
  hipLaunchKernelGGL(k_scan_top, dim3(1), b, 0, st, S.tile_sums, tiles, total_out);
  if (mode == 0) {
    if (out64) hipLaunchKernelGGL((k_scan_apply<0, uint64_t>), g, b, 0, st, S, (uint64_t *)out);
    else hipLaunchKernelGGL((k_scan_apply<0, uint32_t>), g, b, 0, st, S, (uint32_t *)out);
  } else if (mode == 1) {
    hipLaunchKernelGGL((k_scan_apply<1, uint64_t>), g, b, 0, st, S, (uint64_t *)out);
  } else if (mode == 3) {
    hipLaunchKernelGGL((k_scan_apply<3, uint64_t>), g, b, 0, st, S, (uint64_t *)out);
  } else {
    if (out64) hipLaunchKernelGGL((k_scan_apply<2, uint64_t>), g, b, 0, st, S, (uint64_t *)out);
    else hipLaunchKernelGGL((k_scan_apply<2, uint32_t>), g, b, 0, st, S, (uint32_t *)out);
  }
}

// expand (small batches): the work list as well; returns whether it was written (one launch did both), else the caller
// launches k_expand
bool launch_scan3(hipStream_t st, ScanArgs S, uint32_t *match_off, uint64_t *cig_base, uint32_t *fast_pre,
                  uint64_t *total_out3, const ProjectArgs *expand) {
  int64_t tiles = scan_tiles_for(S.n);
  if (tiles < 1) tiles = 1;
  S.n_tiles = tiles;
  dim3 g((unsigned)tiles), b(256);
  if (tiles <= SCAN_SMALL_TILES) {
    if (expand && S.n <= EXPAND_SMALL_N) { hipLaunchKernelGGL((k_scan3_small<true>), dim3(1), b, 0, st, S, match_off, cig_base, fast_pre, total_out3, *expand); return true; }
    const ProjectArgs none{};
    hipLaunchKernelGGL((k_scan3_small<false>), dim3(1), b, 0, st, S, match_off, cig_base, fast_pre, total_out3, none);
    return false;
  }
  hipLaunchKernelGGL(k_scan3_tiles, g, b, 0, st, S);
  hipLaunchKernelGGL(k_scan3_top, dim3(1), b, 0, st, S.tile_sums, tiles, total_out3);
  hipLaunchKernelGGL(k_scan3_apply, g, b, 0, st, S, match_off, cig_base, fast_pre);
  return false;
}

void launch_stats(hipStream_t st, const StatsArgs &T, const uint2 *m_p, const uint32_t *match_off, const uint32_t *n_matches) {
  if (T.n_aln > 0) hipLaunchKernelGGL(k_stats, dim3(grid_for(T.n_aln, 256)), dim3(256), 0, st, T);
  if (T.n_aln > 0 && m_p) hipLaunchKernelGGL(k_sum_ncig, dim3(1024), dim3(256), 0, st, m_p, match_off, n_matches, T.n_aln, T.out + 7);
}

void launch_group_ids(hipStream_t st, int64_t n_groups, const uint32_t *group_off, uint32_t *aln_group) {
  if (n_groups <= 0) return;
  hipLaunchKernelGGL(k_group_ids, dim3(grid_for(n_groups, 256)), dim3(256), 0, st, n_groups, group_off, aln_group);
}

void launch_pair(hipStream_t st, const PairArgs &P, bool emit) {
  if (P.n_aln <= 0) return;
  if (emit) hipLaunchKernelGGL(k_pair_emit, dim3(grid_for(P.n_aln, 256)), dim3(256), 0, st, P);
  else hipLaunchKernelGGL((k_pair<false>), dim3(grid_for(P.n_aln, 256)), dim3(256), 0, st, P);
}

void launch_primary(hipStream_t st, const PairArgs &P, const uint32_t *name_off, const uint8_t *names, bool has_scores) {
  if (P.n_groups <= 0) return;
  dim3 g(std::min(grid_for(P.n_groups, 256), 4096)), b(256);
  if (has_scores) hipLaunchKernelGGL((k_primary<true>), g, b, 0, st, P, name_off, names);
  else hipLaunchKernelGGL((k_primary<false>), g, b, 0, st, P, name_off, names);
}

void launch_rows(hipStream_t st, const PairArgs &P, bool aux) {
  if (P.n_rows_total <= 0) return;
  dim3 g(grid_for(P.n_rows_total, 256)), b(256);
  if (aux) hipLaunchKernelGGL((k_rows<true>), g, b, 0, st, P);
  else hipLaunchKernelGGL((k_rows<false>), g, b, 0, st, P);
}

void launch_rows_detail(hipStream_t st, const PairArgs &P) {
  if (P.n_rows_total <= 0) return;
  hipLaunchKernelGGL(k_rows_detail, dim3(grid_for(P.n_rows_total, 256)), dim3(256), 0, st, P);
}

void launch_pool_sizes(hipStream_t st, const PoolArgs &Q) {
  if (Q.n_rows > 0) hipLaunchKernelGGL(k_pool_sizes, dim3(grid_for(Q.n_rows, 256)), dim3(256), 0, st, Q);
}

void launch_pool_copy(hipStream_t st, const PoolArgs &Q, bool long_cigars) {
  if (Q.n_rows <= 0) return;
  if (long_cigars) hipLaunchKernelGGL((k_pool_copy<16>), dim3(grid_for(Q.n_rows * 16, 256)), dim3(256), 0, st, Q);
  else hipLaunchKernelGGL((k_pool_copy<1>), dim3(grid_for(Q.n_rows, 256)), dim3(256), 0, st, Q);
}

void launch_wide_fields(hipStream_t st, const WideArgs &W) {
  if (W.n_rows > 0) hipLaunchKernelGGL(k_wide_fields, dim3(grid_for(W.n_rows, 256)), dim3(256), 0, st, W);
}

void launch_wide_cigars(hipStream_t st, const WideArgs &W, int64_t n_words) {
  if (W.n_rows <= 0) return;
  if (n_words > 8 * W.n_rows) hipLaunchKernelGGL((k_wide_cigars<16>), dim3(grid_for(W.n_rows * 16, 256)), dim3(256), 0, st, W);
  else hipLaunchKernelGGL((k_wide_cigars<1>), dim3(grid_for(W.n_rows, 256)), dim3(256), 0, st, W);
}

}  // namespace br
