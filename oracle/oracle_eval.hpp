// ============================================================================
// TEST INFRASTRUCTURE ONLY -- per-alignment evaluator, CIGAR merge, mate
// pairing and NH finalisation of the CPU oracle (see oracle_core.hpp for the
// oracle's role, pinning status and usage rules).
// ============================================================================
#pragma once
#include "oracle_core.hpp"
#include "oracle_ksw2.hpp"

namespace orc {

// Insertion-ordered map tid -> TidData: the iteration-order contract of
// ankerl::unordered_dense::map (include/types.h:11-13; values live in a dense
// vector in insertion order; assigning to an existing key keeps its slot).
struct TidMap {
  std::vector<std::pair<uint32_t, TidData>> items;
  std::unordered_map<uint32_t, size_t> pos;
  void clear() { items.clear(); pos.clear(); }
  bool empty() const { return items.empty(); }
  TidData *find(uint32_t tid) {
    auto it = pos.find(tid);
    return it == pos.end() ? nullptr : &items[it->second].second;
  }
  void assign(uint32_t tid, const TidData &td) {
    auto it = pos.find(tid);
    if (it == pos.end()) { pos[tid] = items.size(); items.emplace_back(tid, td); }
    else items[it->second].second = td;
  }
};

struct Evaluator {
  const G2T *g2t;
  Flags flags;
  EvalConfig config;
  Evaluator(const G2T *g, const Flags &f) : g2t(g), flags(f), config(resolve_config(f)) {}

  // src/evaluate.cpp:44-56
  static ExonStatus get_exon_status(uint32_t exon_count, uint32_t j) {
    if (exon_count == 1) return ONLY_EXON;
    if (j == 0) return FIRST_EXON;
    if (j < exon_count - 1) return MIDDLE_EXON;
    return LAST_EXON;
  }

  // src/evaluate.cpp:58-67
  std::vector<char> get_strands_to_check(const Read &read) const {
    if (flags.long_reads()) return {'+', '-'};
    if (read.strand == '+') return {'+'};
    if (read.strand == '-') return {'-'};
    return {'+', '-'};
  }

  // src/evaluate.cpp:69-109
  void get_clips(const Read &read, bool &failure, bool &has_left_clip, bool &has_right_clip,
                 uint32_t &n_left_clip, uint32_t &n_right_clip) const {
    const std::vector<uint32_t> &cigar = read.cigar;
    uint32_t n_cigar = (uint32_t)cigar.size();
    if (n_cigar == 0) { failure = true; return; }
    if (cig_op(cigar[0]) == C_HARD_CLIP) {
      if (1 < n_cigar && cig_op(cigar[1]) == C_SOFT_CLIP) {
        has_left_clip = flags.use_fasta; n_left_clip = cig_len(cigar[1]);
      }
    } else if (cig_op(cigar[0]) == C_SOFT_CLIP) {
      has_left_clip = flags.use_fasta; n_left_clip = cig_len(cigar[0]);
    }
    if (cig_op(cigar[n_cigar - 1]) == C_HARD_CLIP) {
      // `n_cigar - 2 >= 0` is always true upstream (unsigned); a 1-op CIGAR "nH"
      // would read cigar[-1] there -- the oracle treats that read as no clip.
      if (n_cigar >= 2 && cig_op(cigar[n_cigar - 2]) == C_SOFT_CLIP) {
        has_right_clip = flags.use_fasta; n_right_clip = cig_len(cigar[n_cigar - 2]);
      }
    } else if (cig_op(cigar[n_cigar - 1]) == C_SOFT_CLIP) {
      has_right_clip = flags.use_fasta; n_right_clip = cig_len(cigar[n_cigar - 1]);
    }
  }

  // src/evaluate.cpp:111-182
  bool correct_for_gaps(TidData &td, uint32_t tid, Segment &seg2, char strand, int refid) const {
    const Segment *prev_guide = nullptr;
    for (int k = (int)td.segments.size() - 1; k >= 0; k--)
      if (td.segments[k].has_gexon) { prev_guide = &td.segments[k]; break; }
    if (!prev_guide) return true;
    const Segment &seg1 = *prev_guide;
    uint8_t gap = (uint8_t)(seg2.gexon.exon_id - seg1.gexon.exon_id);
    if (!flags.long_reads()) {
      if (gap != 1) { td.elim = true; return false; }
      return true;
    }
    if (gap > 2) { td.elim = true; return false; }
    if (gap == 2) {
      uint32_t gap_start = (strand == '+') ? seg2.gexon.prev_start : seg2.gexon.next_start;
      uint32_t gap_end = (strand == '+') ? seg2.gexon.prev_end : seg2.gexon.next_end;
      if ((gap_start == 0 && gap_end == 0) || (gap_end - gap_start > config.max_error_exon)) {
        td.elim = true; return false;
      }
      GuideExon gap_exon;
      if (!g2t->guide_exon_for_tid(refid, strand, tid, gap_start, gap_end, gap_exon)) {
        td.elim = true; return false;
      }
      Segment gap_seg;
      gap_seg.gexon = gap_exon; gap_seg.has_gexon = true; gap_seg.status = GAP_EXON;
      gap_seg.is_small_exon = (gap_exon.end - gap_exon.start <= config.max_error_exon);
      td.segments.push_back(gap_seg);
    }
    return true;
  }

  // src/evaluate.cpp:184-282
  void get_intervals(TidMap &data, const Read &read, uint32_t j, uint32_t exon_count, int refid,
                     char strand, bool has_left_clip, bool has_right_clip, bool &failure) const {
    const GSeg &qexon = read.segs[j];
    ExonStatus status = get_exon_status(exon_count, j);
    bool is_small_exon = (qexon.end - qexon.start <= config.max_error_exon);
    bool data_empty = data.empty();
    std::vector<GuideExon> guide_exons;
    std::vector<uint32_t> candidate_tids;
    if (g2t->guide_exons(refid, strand, qexon, config, status, guide_exons)) {
      for (auto &gexon : guide_exons) {
        uint32_t tid = gexon.tid;
        candidate_tids.push_back(tid);
        Segment segment;
        segment.has_gexon = true; segment.has_qexon = true; segment.gexon = gexon;
        segment.qexon = qexon; segment.status = status; segment.is_small_exon = is_small_exon;
        if (data_empty) {
          TidData td;
          td.has_left_clip = has_left_clip; td.has_right_clip = has_right_clip;
          td.segments.push_back(segment);
          data.assign(tid, td);
        } else {
          TidData *tdp = data.find(tid);
          if (!tdp || tdp->elim) continue;
          correct_for_gaps(*tdp, tid, segment, strand, refid);
          tdp->segments.push_back(segment);
        }
      }
      for (auto &pair : data.items)
        if (std::find(candidate_tids.begin(), candidate_tids.end(), pair.first) == candidate_tids.end())
          pair.second.elim = true;
      return;
    }
    if (status != ONLY_EXON && config.ignore_small_exons && is_small_exon) {
      if (status == MIDDLE_EXON) {
        if (data.empty()) { failure = true; return; }
        for (auto &pair : data.items) {
          Segment ignore;
          ignore.qexon = qexon; ignore.has_qexon = true; ignore.has_gexon = false;
          ignore.status = INS_EXON; ignore.is_small_exon = true;
          pair.second.segments.push_back(ignore);
        }
        return;
      }
      failure = true; return;
    }
    failure = true;
  }

  // src/evaluate.cpp:284-317 (align): ksw_extz2_sse(0, ql, qs, tl, ts, 5, mat, gapo, gape, -1,
  // zdrop, 0, EXTZ_ONLY|APPROX_MAX|APPROX_DROP, &ez)
  struct KswResult { std::vector<uint32_t> cigar; int score = 0, max = 0; };
  static KswResult align(const std::string &tseq, const std::string &qseq, int sc_mch, int sc_mis,
                         int gapo, int gape, int zdrop) {
    KswResult result;
    int a = sc_mch, b = sc_mis < 0 ? sc_mis : -sc_mis;
    int8_t mat[25] = {(int8_t)a, (int8_t)b, (int8_t)b, (int8_t)b, 0, (int8_t)b, (int8_t)a, (int8_t)b,
                      (int8_t)b, 0, (int8_t)b, (int8_t)b, (int8_t)a, (int8_t)b, 0, (int8_t)b,
                      (int8_t)b, (int8_t)b, (int8_t)a, 0, 0, 0, 0, 0, 0};
    // strlen() semantics: sequences stop at the first NUL
    size_t tl = strlen(tseq.c_str()), ql = strlen(qseq.c_str());
    uint8_t c[256];
    memset(c, 4, 256);
    c['A'] = c['a'] = 0; c['C'] = c['c'] = 1; c['G'] = c['g'] = 2; c['T'] = c['t'] = 3;
    std::vector<uint8_t> ts(tl), qs(ql);
    for (size_t i = 0; i < tl; ++i) ts[i] = c[(uint8_t)tseq[i]];
    for (size_t i = 0; i < ql; ++i) qs[i] = c[(uint8_t)qseq[i]];
    ksw_extz ez;
    ksw_extz2_scalar((int)ql, qs.data(), (int)tl, ts.data(), 5, mat, (int8_t)gapo, (int8_t)gape,
                     zdrop, &ez);
    result.score = ez.score; result.max = (int)ez.max; result.cigar = ez.cigar;
    return result;
  }

  // src/evaluate.cpp:319-330 / :500-511.  `seq` is the ASCII read sequence (the
  // reference decodes the same characters from BAM 4-bit codes through
  // seq_nt16_str).
  static std::string left_rescue_query(const std::string &seq, uint32_t n_left_clip, uint32_t left_ins) {
    int seq_len = (int)seq.size();
    uint32_t total = n_left_clip + left_ins;
    if ((int)total > seq_len) total = (uint32_t)seq_len;
    return seq.substr(0, total);
  }
  static std::string right_rescue_query(const std::string &seq, uint32_t n_right_clip, uint32_t right_ins) {
    int seq_len = (int)seq.size();
    uint32_t total = right_ins + n_right_clip;
    if ((int)total > seq_len) total = (uint32_t)seq_len;
    int start = seq_len - (int)total;
    return seq.substr(start, total);
  }

  // src/evaluate.cpp:333-365
  bool collect_left_exons(const std::string &qseq, int refid, char strand, uint32_t tid,
                          const GuideExon &start, std::string &out_gseq, int &n_out) const {
    GuideExon curr = start;
    int i = 0;
    n_out = 0;
    while (qseq.length() > out_gseq.length()) {
      i++;
      bool has_neighbor = (strand == '+') ? curr.has_prev : curr.has_next;
      if (!has_neighbor) { if (i == 1) return false; break; }
      GuideExon next;
      bool ok;
      if (strand == '+') ok = g2t->guide_exon_for_tid(refid, strand, tid, curr.prev_start, curr.prev_end, next);
      else ok = g2t->guide_exon_for_tid(refid, strand, tid, curr.next_start, curr.next_end, next);
      if (!ok || !next.seq) break;  // upstream would read an uninitialised exon; cannot happen for a consistent index
      out_gseq = std::string(next.seq->c_str()) + out_gseq;
      n_out++;
      curr = next;
    }
    return n_out != 0;
  }
  // src/evaluate.cpp:513-546
  bool collect_right_exons(const std::string &qseq, int refid, char strand, uint32_t tid,
                           const GuideExon &start, std::string &out_gseq, int &n_out) const {
    GuideExon curr = start;
    int i = 0;
    n_out = 0;
    while (qseq.length() > out_gseq.length()) {
      i++;
      bool has_neighbor = (strand == '+') ? curr.has_next : curr.has_prev;
      if (!has_neighbor) { if (i == 1) return false; break; }
      GuideExon next;
      bool ok;
      if (strand == '+') ok = g2t->guide_exon_for_tid(refid, strand, tid, curr.next_start, curr.next_end, next);
      else ok = g2t->guide_exon_for_tid(refid, strand, tid, curr.prev_start, curr.prev_end, next);
      if (!ok || !next.seq) break;
      out_gseq += next.seq->c_str();
      n_out++;
      curr = next;
    }
    return n_out != 0;
  }

  // src/evaluate.cpp:368-395
  static KswResult align_reversed(const std::string &qseq, const std::string &gseq) {
    int len1 = (int)qseq.length(), len2 = (int)gseq.length();
    int start_pos = std::max(0, len2 - (len1 + 40));
    std::string gseq_short = gseq.substr(start_pos);
    std::string qrev = qseq, grev = gseq_short;
    std::reverse(qrev.begin(), qrev.end());
    std::reverse(grev.begin(), grev.end());
    return align(grev, qrev, 1, -4, 4, 1, 40);
  }

  // src/evaluate.cpp:397-448
  static Segment build_left_clip_segment(const KswResult &result, int q_len, int small_exon,
                                         const GuideExon &gexon) {
    int query_consumed = 0, ref_consumed = 0;
    int n = (int)result.cigar.size();
    for (int i = 0; i < n; ++i) {
      int op = cig_op(result.cigar[i]), len = cig_len(result.cigar[i]);
      if (op == C_MATCH || op == C_INS || op == C_SOFT_CLIP) query_consumed += len;
      if (op == C_MATCH || op == C_DEL) ref_consumed += len;
    }
    int left_clip = q_len - query_consumed;
    Segment seg;
    seg.has_qexon = false; seg.has_gexon = true; seg.status = LEFTC_EXON;
    seg.is_small_exon = (q_len <= small_exon);
    seg.score = result.max;
    GuideExon dummy;
    dummy.start = gexon.start - ref_consumed; dummy.end = gexon.start;
    dummy.pos = gexon.pos_start - ref_consumed;
    seg.gexon = dummy;
    if (left_clip > 0) seg.cigar.add_operation(left_clip, C_CLIP_OVERRIDE);
    for (int i = n - 1; i >= 0; --i) {
      int len = cig_len(result.cigar[i]), op = cig_op(result.cigar[i]);
      char op_char = "MID"[op];
      if (i == n - 1 && op_char == 'D') {}
      else if (i == n - 1 && op_char == 'I') seg.cigar.add_operation(len, C_CLIP_OVERRIDE);
      else if (op_char == 'D') seg.cigar.add_operation(len, C_DEL_OVERRIDE);
      else if (op_char == 'I') seg.cigar.add_operation(len, C_INS_OVERRIDE);
      else seg.cigar.add_operation(len, C_MATCH_OVERRIDE);
    }
    return seg;
  }

  // src/evaluate.cpp:451-498
  void left_clip_rescue(TidData &td, char strand, int refid, uint32_t tid, uint32_t n_left_clip,
                        const std::string &seq) const {
    td.has_left_clip = false;
    Segment &seg = td.segments[0];
    if (!seg.has_gexon || seg.gexon.left_gap > 0) return;
    GuideExon &gexon = seg.gexon;
    std::string qseq = left_rescue_query(seq, n_left_clip, (uint32_t)gexon.left_ins);
    std::string gseq; int n_exons = 0;
    if (!collect_left_exons(qseq, refid, strand, tid, gexon, gseq, n_exons)) return;
    KswResult result = align_reversed(qseq, gseq);
    if (result.max < 10 || result.score == KSW_NEG_INF) return;
    if (gexon.left_ins > 0) gexon.left_ins = 0;
    Segment left_clip = build_left_clip_segment(result, (int)qseq.length(), (int)config.max_error_exon, gexon);
    td.segments.insert(td.segments.begin(), left_clip);
    td.has_left_clip = true;
  }

  // src/evaluate.cpp:548-598
  static Segment build_right_clip_segment(const KswResult &result, int q_len, int small_exon,
                                          const GuideExon &gexon) {
    int query_consumed = 0, ref_consumed = 0;
    int n = (int)result.cigar.size();
    for (int i = 0; i < n; ++i) {
      int op = cig_op(result.cigar[i]), len = cig_len(result.cigar[i]);
      if (op == C_MATCH || op == C_INS || op == C_SOFT_CLIP) query_consumed += len;
      if (op == C_MATCH || op == C_DEL) ref_consumed += len;
    }
    int right_clip = q_len - query_consumed;
    Segment seg;
    seg.has_qexon = false; seg.has_gexon = true; seg.status = RIGHTC_EXON;
    seg.is_small_exon = (q_len <= small_exon);
    seg.score = result.max;
    GuideExon dummy;
    dummy.start = gexon.end; dummy.end = gexon.end + ref_consumed;
    dummy.pos = gexon.pos_start - ref_consumed;
    seg.gexon = dummy;
    for (int i = 0; i < n; ++i) {
      int len = cig_len(result.cigar[i]), op = cig_op(result.cigar[i]);
      char op_char = "MID"[op];
      if (i == n - 1 && op_char == 'D') {}
      else if (i == n - 1 && op_char == 'I') seg.cigar.add_operation(len, C_CLIP_OVERRIDE);
      else if (op_char == 'D') seg.cigar.add_operation(len, C_DEL_OVERRIDE);
      else if (op_char == 'I') seg.cigar.add_operation(len, C_INS_OVERRIDE);
      else seg.cigar.add_operation(len, C_MATCH_OVERRIDE);
    }
    if (right_clip > 0) seg.cigar.add_operation(right_clip, C_CLIP_OVERRIDE);
    return seg;
  }

  // src/evaluate.cpp:600-656
  void right_clip_rescue(TidData &td, char strand, int refid, uint32_t tid, uint32_t n_right_clip,
                         const std::string &seq) const {
    td.has_right_clip = false;
    Segment &seg = td.segments.back();
    if (!seg.has_gexon || seg.gexon.right_gap > 0) return;
    GuideExon &gexon = seg.gexon;
    std::string qseq = right_rescue_query(seq, n_right_clip, (uint32_t)gexon.right_ins);
    std::string gseq; int n_exons = 0;
    if (!collect_right_exons(qseq, refid, strand, tid, gexon, gseq, n_exons)) return;
    std::string gseq_short = gseq.substr(0, qseq.length() + 40);
    KswResult result = align(gseq_short, qseq, 1, -4, 4, 1, 40);
    if (result.max < 10 || result.score == KSW_NEG_INF) return;
    if (gexon.right_ins > 0) gexon.right_ins = 0;
    Segment right_clip = build_right_clip_segment(result, (int)qseq.length(), (int)config.max_error_exon, gexon);
    td.segments.push_back(right_clip);
    td.has_right_clip = true;
  }

  // src/evaluate.cpp:658-673
  static void create_match(TidData &td, const GuideExon &gexon, uint32_t tid, char strand) {
    ExonChainMatch &match = td.match;
    match.tid = tid;
    match.align.fwpos = gexon.pos; match.align.rcpos = gexon.pos;
    match.transcript_len = (int32_t)gexon.transcript_len;
    match.align.strand = strand;
    match.align.cigar = Cigar();
    match.align.similarity_score = 0;
    match.total_coverage = 0; match.total_operations = 0;
    match.ref_consumed = 0; match.prev_op = C_MATCH; match.junc_hits = 0;
  }

  // src/evaluate.cpp:675-786
  static void build_cigar_match(const Segment &seg, const TidData &td, ExonChainMatch &match,
                                bool first_match, bool last_match) {
    uint32_t qstart = seg.qexon.start, qend = seg.qexon.end;
    uint32_t gstart = seg.gexon.start, gend = seg.gexon.end;
    uint32_t left_ins = (uint32_t)seg.gexon.left_ins, left_gap = (uint32_t)seg.gexon.left_gap;
    uint32_t right_ins = (uint32_t)seg.gexon.right_ins, right_gap = (uint32_t)seg.gexon.right_gap;
    Cigar &cigar = match.align.cigar;
    if (left_ins > 0) {
      if (seg.status == FIRST_EXON || seg.status == ONLY_EXON) {
        if (!td.has_left_clip) {
          cigar.add_operation(left_ins, C_SOFT_CLIP);
          match.total_operations += left_ins;
          match.prev_op = C_SOFT_CLIP;
        }
      } else if (seg.status == MIDDLE_EXON || seg.status == LAST_EXON || td.has_left_clip) {
        cigar.add_operation(left_ins, C_INS);
        match.total_operations += left_ins;
        if (match.prev_op == C_DEL) match.total_coverage += left_ins;
        else if (match.prev_op == C_INS) match.total_operations += (match.total_operations * 0.2);
        match.prev_op = C_INS;
      }
    } else if (left_gap > 0) {
      if (!first_match && (seg.status == MIDDLE_EXON || seg.status == LAST_EXON || td.has_left_clip)) {
        cigar.add_operation(left_gap, C_DEL);
        match.total_operations += left_gap;
        match.ref_consumed += left_gap;
        if (match.prev_op == C_INS) match.total_coverage += left_gap;
        else if (match.prev_op == C_DEL) match.total_operations += (match.total_operations * 0.2);
        match.prev_op = C_DEL;
      }
    } else {
      match.junc_hits++;
    }
    uint32_t overlap_start = std::max(qstart, gstart);
    uint32_t overlap_end = std::min(qend, gend);
    if (overlap_end >= overlap_start) {
      uint32_t match_length = overlap_end - overlap_start;
      cigar.add_operation(match_length, C_MATCH);
      match.total_operations += match_length;
      match.total_coverage += match_length;
      match.ref_consumed += match_length;
      match.prev_op = C_MATCH;
    }
    if (right_ins > 0) {
      if (seg.status == LAST_EXON || seg.status == ONLY_EXON) {
        if (!td.has_right_clip) {
          cigar.add_operation(right_ins, C_SOFT_CLIP);
          match.total_operations += right_ins;
          match.prev_op = C_SOFT_CLIP;
        }
      } else if (seg.status == FIRST_EXON || seg.status == MIDDLE_EXON || td.has_right_clip) {
        cigar.add_operation(right_ins, C_INS);
        match.total_operations += right_ins;
        if (match.prev_op == C_DEL) match.total_coverage += right_ins;
        match.prev_op = C_INS;
      }
    } else if (right_gap > 0) {
      if (!last_match && (seg.status == FIRST_EXON || seg.status == MIDDLE_EXON || td.has_right_clip)) {
        cigar.add_operation(right_gap, C_DEL);
        match.total_operations += right_gap;
        match.ref_consumed += right_gap;
        if (match.prev_op == C_INS) match.total_coverage += right_gap;
        match.prev_op = C_DEL;
      }
    } else {
      match.junc_hits++;
    }
  }

  // src/evaluate.cpp:788-806
  static void build_cigar_ins(const Segment &seg, uint32_t k, uint32_t n, ExonChainMatch &match) {
    uint32_t len = seg.qexon.end - seg.qexon.start;
    Cigar &cigar = match.align.cigar;
    if (k == 0 || k == (n - 1)) { cigar.add_operation(len, C_SOFT_CLIP); match.prev_op = C_SOFT_CLIP; }
    else { cigar.add_operation(len, C_INS); match.prev_op = C_INS; }
    match.total_operations += len;
    match.total_coverage += len;
  }
  // src/evaluate.cpp:808-822
  static void build_cigar_gap(const Segment &seg, ExonChainMatch &match) {
    uint32_t len = seg.gexon.end - seg.gexon.start;
    match.align.cigar.add_operation(len, C_DEL);
    match.prev_op = C_DEL;
    match.total_operations += len;
    match.total_coverage += len;
    match.ref_consumed += len;
  }
  // src/evaluate.cpp:824-841
  static void build_cigar_clip(const Segment &seg, ExonChainMatch &match) {
    for (uint32_t cig : seg.cigar.ops) {
      uint8_t op = cig_op(cig); uint32_t len = cig_len(cig);
      match.align.cigar.add_operation(len, op);
      if (op == C_MATCH_OVERRIDE || op == C_DEL_OVERRIDE) match.ref_consumed += len;
    }
    match.align.clip_score += seg.score;
  }

  // src/evaluate.cpp:843-886 (the BRAMBLE_DEBUG block never runs: no flag sets it)
  void filter_by_similarity(std::vector<ExonChainMatch> &matches) const {
    if (!config.filter_by_similarity) return;
    for (auto it = matches.begin(); it != matches.end();) {
      ExonChainMatch &match = *it;
      double similarity = (match.total_operations > 0) ? (match.total_coverage / match.total_operations) : 0.0;
      if (similarity > config.similarity_threshold) {
        double x = ((similarity - config.similarity_threshold) / (1.0 - config.similarity_threshold));
        match.align.similarity_score = (x * x * static_cast<double>(match.junc_hits + 1));
        ++it;
      } else {
        it = matches.erase(it);
      }
    }
  }

  // src/evaluate.cpp:888-1134 (evaluate_exon_chains) reached through :1136-1221
  std::vector<ExonChainMatch> evaluate(const Read &read, const std::string &seq) const {
    uint32_t exon_count = (uint32_t)read.segs.size();
    int refid = read.refid;
    std::vector<ExonChainMatch> matches_by_strand;
    bool has_left_clip = false, has_right_clip = false;
    uint32_t n_left_clip = 0, n_right_clip = 0;
    bool failure = false;
    TidMap data;
    if (flags.long_reads())
      get_clips(read, failure, has_left_clip, has_right_clip, n_left_clip, n_right_clip);

    for (char strand : get_strands_to_check(read)) {
      data.clear();
      failure = false;
      for (uint32_t j = 0; j < exon_count; j++) {
        get_intervals(data, read, j, exon_count, refid, strand, has_left_clip, has_right_clip, failure);
        if (failure) break;
      }
      if (failure) continue;

      if (flags.long_reads() && flags.use_fasta) {
        for (auto &pair : data.items) {
          uint32_t tid = pair.first; TidData &td = pair.second;
          if (td.elim) continue;
          if (td.has_left_clip) {
            if (n_left_clip >= 5) left_clip_rescue(td, strand, refid, tid, n_left_clip, seq);
            else td.has_left_clip = false;
          }
          if (td.has_right_clip) {
            if (n_right_clip >= 5) right_clip_rescue(td, strand, refid, tid, n_right_clip, seq);
            else td.has_right_clip = false;
          }
        }
      }

      for (auto &pair : data.items) {
        uint32_t tid = pair.first; TidData &td = pair.second;
        if (td.elim) continue;
        uint32_t n_segments = (uint32_t)td.segments.size();
        bool match_created = false;
        uint32_t first_match_idx = (uint32_t)-1, last_match_idx = (uint32_t)-1;
        uint32_t prev_gs = UINT32_MAX, prev_ge = UINT32_MAX, prev_qs = UINT32_MAX, prev_qe = UINT32_MAX;
        bool qset = false, gset = false;
        for (uint32_t k = 0; k < n_segments; k++) {
          Segment &seg = td.segments[k];
          if (seg.has_gexon) {
            if (gset && seg.gexon.start == prev_gs && seg.gexon.end == prev_ge) { td.elim = true; break; }
            prev_gs = seg.gexon.start; prev_ge = seg.gexon.end; gset = true;
          }
          if (seg.has_qexon) {
            if (qset && seg.qexon.start == prev_qs && seg.qexon.end == prev_qe) { td.elim = true; break; }
            prev_qs = seg.qexon.start; prev_qe = seg.qexon.end; qset = true;
          }
          if (!match_created && seg.has_gexon) {
            create_match(td, seg.gexon, tid, strand);
            match_created = true;
            first_match_idx++; last_match_idx++;
          } else if (match_created && seg.has_gexon && seg.status != INS_EXON) {
            last_match_idx++;
            if (strand == '-') td.match.align.rcpos = seg.gexon.pos;
          }
        }
        for (uint32_t k = 0; k < n_segments; k++) {
          Segment &seg = td.segments[k];
          if (td.elim) break;
          bool first_match = (k == first_match_idx), last_match = (k == last_match_idx);
          bool is_match = (seg.status == FIRST_EXON || seg.status == MIDDLE_EXON ||
                           seg.status == LAST_EXON || seg.status == ONLY_EXON);
          if (is_match) build_cigar_match(seg, td, td.match, first_match, last_match);
          else if (seg.status == INS_EXON) {
            build_cigar_ins(seg, k, n_segments, td.match);
            td.match.junc_hits -= (k == 0 || k == n_segments - 1) ? 1 : 2;
          } else if (seg.status == GAP_EXON) {
            build_cigar_gap(seg, td.match);
            td.match.junc_hits -= 2;
          } else if (seg.status == LEFTC_EXON || seg.status == RIGHTC_EXON) {
            build_cigar_clip(seg, td.match);
          }
        }
        if (td.match.junc_hits < 0) td.match.junc_hits = 0;
        if (!td.elim) matches_by_strand.push_back(td.match);
      }
    }
    if (!matches_by_strand.empty()) filter_by_similarity(matches_by_strand);
    return matches_by_strand;
  }
};

// ---- CIGAR merge ------------------------------------------------------------
// src/bam.cpp:22-111.  `char` is signed in the reference build
// (meson.build: -fsigned-char); op codes 0..13 and '_' (95) are unaffected.
static inline char merge_ops(char real_op, char ideal_op) {
  if ((real_op == C_MATCH || real_op == C_SOFT_CLIP) && ideal_op == C_CLIP_OVERRIDE) return C_SOFT_CLIP;
  if ((real_op == C_MATCH || real_op == C_SOFT_CLIP) && ideal_op == C_MATCH_OVERRIDE) return C_MATCH;
  if ((real_op == C_MATCH || real_op == C_SOFT_CLIP) && ideal_op == C_INS_OVERRIDE) return C_INS;
  if ((real_op == C_MATCH || real_op == C_SOFT_CLIP) && ideal_op == C_DEL_OVERRIDE) return C_DEL;
  if (real_op == C_DEL && (ideal_op == C_SOFT_CLIP || ideal_op == C_CLIP_OVERRIDE)) return '_';
  if (real_op == C_DEL && ideal_op == C_MATCH_OVERRIDE) return C_DEL;
  if (real_op == C_INS && ideal_op == C_CLIP_OVERRIDE) return C_SOFT_CLIP;
  if (real_op == C_INS && ideal_op == C_MATCH_OVERRIDE) return C_INS;
  if (ideal_op == C_CLIP_OVERRIDE) return C_SOFT_CLIP;
  if (ideal_op == C_MATCH_OVERRIDE) return C_MATCH;
  if (ideal_op == C_INS_OVERRIDE) return C_INS;
  if (ideal_op == C_DEL_OVERRIDE) return C_DEL;
  if (real_op == C_PAD) return ideal_op;
  if (real_op == C_HARD_CLIP) return C_HARD_CLIP;
  if (real_op == C_INS && ideal_op == C_SOFT_CLIP) return C_SOFT_CLIP;
  if (ideal_op == C_SOFT_CLIP || ideal_op == C_DEL || ideal_op == C_INS) return ideal_op;
  if (real_op == C_SOFT_CLIP || real_op == C_DEL || real_op == C_INS) return real_op;
  if (ideal_op == C_MATCH || ideal_op == C_EQUAL) return C_MATCH;
  if (ideal_op == C_DIFF) return C_DIFF;
  if (real_op == C_MATCH || real_op == C_EQUAL) return C_MATCH;
  if (real_op == C_DIFF) return C_DIFF;
  return ideal_op;
}

// src/bam.cpp:113-315
static inline std::vector<uint32_t> merge_cigars(const uint32_t *real_cigar, uint32_t n_real_cigar,
                                                 const std::vector<uint32_t> &ideal,
                                                 uint32_t real_front_hard_clip,
                                                 uint32_t real_front_soft_clip) {
  uint32_t n_ideal_cigar = (uint32_t)ideal.size();
  std::vector<uint32_t> result(n_real_cigar + n_ideal_cigar + 1, 0);
  uint32_t result_idx = 0, ri = 0, ii = 0, real_pos = 0, ideal_pos = 0;
  auto add_op = [&](uint8_t op, uint32_t len) {
    if (len == 0) return;
    if (op == '_') return;
    if (result_idx > 0 && cig_op(result[result_idx - 1]) == op) result[result_idx - 1] += (len << 4);
    else result[result_idx++] = cig_gen(len, op);
  };
  auto get_remaining = [](uint32_t cigar_val, uint32_t pos) { return cig_len(cigar_val) - pos; };

  uint32_t clips_remaining = real_front_hard_clip;
  while (clips_remaining > 0 && ri < n_real_cigar) {
    uint32_t available = get_remaining(real_cigar[ri], real_pos);
    uint32_t chunk = (clips_remaining < available) ? clips_remaining : available;
    add_op((uint8_t)cig_op(real_cigar[ri]), chunk);
    clips_remaining -= chunk;
    real_pos += chunk;
    if (real_pos >= cig_len(real_cigar[ri])) { ri++; real_pos = 0; }
  }

  clips_remaining = real_front_soft_clip;
  while (clips_remaining > 0 && ri < n_real_cigar) {
    uint8_t real_op = (uint8_t)cig_op(real_cigar[ri]);
    uint8_t ideal_op = (ii < n_ideal_cigar) ? (uint8_t)cig_op(ideal[ii]) : 0xff;
    uint32_t real_remaining = get_remaining(real_cigar[ri], real_pos);
    uint32_t ideal_remaining = (ii < n_ideal_cigar) ? get_remaining(ideal[ii], ideal_pos) : UINT32_MAX;
    bool is_override = (ii < n_ideal_cigar &&
        (ideal_op == C_MATCH_OVERRIDE || ideal_op == C_DEL_OVERRIDE ||
         ideal_op == C_INS_OVERRIDE || ideal_op == C_CLIP_OVERRIDE));
    if (is_override) {
      if (ideal_op == C_DEL_OVERRIDE) {
        uint32_t chunk = ideal_remaining;
        add_op((uint8_t)merge_ops((char)real_op, (char)ideal_op), chunk);
        ideal_pos += chunk;
        if (ideal_pos >= cig_len(ideal[ii])) { ii++; ideal_pos = 0; }
      } else {
        uint32_t chunk = clips_remaining;
        if (chunk > real_remaining) chunk = real_remaining;
        if (chunk > ideal_remaining) chunk = ideal_remaining;
        add_op((uint8_t)merge_ops((char)real_op, (char)ideal_op), chunk);
        clips_remaining -= chunk;
        real_pos += chunk; ideal_pos += chunk;
        if (real_pos >= cig_len(real_cigar[ri])) { ri++; real_pos = 0; }
        if (ideal_pos >= cig_len(ideal[ii])) { ii++; ideal_pos = 0; }
      }
    } else {
      uint32_t chunk = clips_remaining;
      if (chunk > real_remaining) chunk = real_remaining;
      add_op((uint8_t)merge_ops((char)real_op, (char)ideal_op), chunk);
      clips_remaining -= chunk;
      real_pos += chunk;
      if (real_pos >= cig_len(real_cigar[ri])) { ri++; real_pos = 0; }
    }
  }

  while (ri < n_real_cigar || ii < n_ideal_cigar) {
    if (ri >= n_real_cigar) {
      uint32_t remaining = get_remaining(ideal[ii], ideal_pos);
      add_op((uint8_t)cig_op(ideal[ii]), remaining);
      ii++; ideal_pos = 0;
      continue;
    }
    if (ii >= n_ideal_cigar) {
      uint32_t remaining = get_remaining(real_cigar[ri], real_pos);
      add_op((uint8_t)cig_op(real_cigar[ri]), remaining);
      ri++; real_pos = 0;
      continue;
    }
    uint8_t real_op = (uint8_t)cig_op(real_cigar[ri]);
    uint8_t ideal_op = (uint8_t)cig_op(ideal[ii]);
    uint32_t real_remaining = get_remaining(real_cigar[ri], real_pos);
    uint32_t ideal_remaining = get_remaining(ideal[ii], ideal_pos);
    if (real_op == C_REF_SKIP) {
      ri++; real_pos = 0;
    } else if (real_op == C_DEL && (ideal_op == C_SOFT_CLIP || ideal_op == C_CLIP_OVERRIDE ||
                                    ideal_op == C_INS || ideal_op == C_INS_OVERRIDE)) {
      uint32_t chunk = (real_remaining < ideal_remaining) ? real_remaining : ideal_remaining;
      real_pos += chunk; ideal_pos += chunk;
      if (real_pos >= cig_len(real_cigar[ri])) { ri++; real_pos = 0; }
      if (ideal_pos >= cig_len(ideal[ii])) { ii++; ideal_pos = 0; }
    } else if (real_op == C_INS) {
      add_op(C_INS, real_remaining);
      ri++; real_pos = 0;
    } else if (ideal_op == C_DEL || ideal_op == C_DEL_OVERRIDE) {
      add_op(C_DEL, ideal_remaining);
      ii++; ideal_pos = 0;
    } else {
      uint32_t chunk = (real_remaining < ideal_remaining) ? real_remaining : ideal_remaining;
      uint8_t merged_op = (uint8_t)merge_ops((char)real_op, (char)ideal_op);
      add_op(merged_op, chunk);
      real_pos += chunk; ideal_pos += chunk;
      if (real_pos >= cig_len(real_cigar[ri])) { ri++; real_pos = 0; }
      if (ideal_pos >= cig_len(ideal[ii])) { ii++; ideal_pos = 0; }
    }
  }

  for (uint32_t i = 1; i + 1 < result_idx; i++) {
    if (cig_op(result[i]) != C_INS) continue;
    uint8_t prev = (uint8_t)cig_op(result[i - 1]), next = (uint8_t)cig_op(result[i + 1]);
    if ((prev == C_SOFT_CLIP || prev == C_HARD_CLIP) && (next == C_SOFT_CLIP || next == C_HARD_CLIP))
      result[i] = cig_gen(cig_len(result[i]), prev);
  }
  uint32_t new_idx = 0;
  for (uint32_t i = 0; i < result_idx; i++) {
    uint8_t op = (uint8_t)cig_op(result[i]); uint32_t len = cig_len(result[i]);
    if (new_idx > 0 && cig_op(result[new_idx - 1]) == op) result[new_idx - 1] += (len << 4);
    else result[new_idx++] = cig_gen(len, op);
  }
  result.resize(new_idx);
  return result;
}

// src/bam.cpp:443-472 (get_new_cigar): leading H then S lengths feed merge_cigars
static inline std::vector<uint32_t> get_new_cigar(const uint32_t *real_cigar, uint32_t n_real_cigar,
                                                  const std::vector<uint32_t> &ideal) {
  uint32_t real_front_hard_clip = 0, real_front_soft_clip = 0, cigar_idx = 0;
  if (n_real_cigar > 0 && cig_op(real_cigar[0]) == C_HARD_CLIP) {
    real_front_hard_clip = cig_len(real_cigar[0]); cigar_idx++;
  }
  if (cigar_idx < n_real_cigar && cig_op(real_cigar[cigar_idx]) == C_SOFT_CLIP)
    real_front_soft_clip = cig_len(real_cigar[cigar_idx]);
  return merge_cigars(real_cigar, n_real_cigar, ideal, real_front_hard_clip, real_front_soft_clip);
}

// ---- pairing + finalisation ---------------------------------------------------
// One emitted BAM record (what write_to_bam, src/core.cpp:96-212, produces for
// one side of a BamInfo).
struct OutRow {
  int32_t input_index = -1;
  uint32_t tid = 0;
  uint32_t pos = 0;             // b->core.pos: fwpos for '+', rcpos for '-'
  char strand = '+';
  std::vector<uint32_t> cigar;  // update_cigar result (before reverse_complement_bam)
  double similarity_score = 0;
  int32_t clip_score = 0;
  int32_t junc_hits = 0, ref_consumed = 0;
  uint32_t nh = 0, hi = 0, mapq = 0;
  bool primary = false;
  bool is_paired = false;       // BamInfo::is_paired (pair emitted together)
  bool same_transcript = false;
  bool is_first = true;         // read1 of the BamInfo (prepare_read is_first)
  int32_t mate_tid = -1, mate_pos = -1, isize = 0;  // set_mate_info, src/bam.cpp:531-588
  uint32_t group = 0;
};

struct BamInfo {  // include/evaluate.h:250-273
  bool same_transcript = false, is_paired = false;
  int read1 = -1, read2 = -1;
  uint32_t r_tid = 0, m_tid = 0;
  AlignInfo r_align, m_align;
  int32_t r_junc = 0, r_refc = 0, m_junc = 0, m_refc = 0;
};

struct ReadInfo {  // include/evaluate.h:234-248
  std::vector<ExonChainMatch> matches;
  int index = -1;
};

// src/core.cpp:46-58
static inline uint32_t get_mapq(uint32_t nh, bool long_reads) {
  if (!long_reads) {
    if (nh == 1) return 255;
    if (nh == 2) return 3;
    if (nh == 3 || nh == 4) return 1;
    return 0;
  }
  return nh > 1 ? 0 : 3;
}

// src/core.cpp:214-218
static inline int32_t get_rand(uint32_t x, uint64_t seed_key) {
  std::mt19937_64 gen(seed_key);
  std::uniform_int_distribution<uint32_t> dis(0, x - 1);
  return (int32_t)dis(gen);
}

// src/mates.cpp:127-141
static inline void update_read_matches(ReadInfo *r, const std::vector<uint32_t> &final_transcripts) {
  std::vector<ExonChainMatch> nm;
  for (const auto &m : r->matches)
    if (std::find(final_transcripts.begin(), final_transcripts.end(), m.tid) != final_transcripts.end())
      nm.push_back(m);
  r->matches = std::move(nm);
}

// src/mates.cpp:150-261 (process_mate_pair) + :28-119 (add_mate_info)
static inline void process_mate_pair(ReadInfo *this_read, ReadInfo *mate_read, std::vector<BamInfo> &emit) {
  if (!this_read) return;
  auto find_match = [](ReadInfo *r, uint32_t tid) -> const ExonChainMatch * {
    // read_alignments[tid] = align: the last match with this tid wins (tids are unique anyway)
    const ExonChainMatch *res = nullptr;
    for (const auto &m : r->matches) if (m.tid == tid) res = &m;
    return res;
  };
  if (mate_read == nullptr) {
    std::vector<uint32_t> read_transcripts;
    std::vector<ExonChainMatch> snapshot = this_read->matches;
    ReadInfo snap; snap.matches = snapshot;
    for (auto &m : this_read->matches) read_transcripts.push_back(m.tid);
    std::sort(read_transcripts.begin(), read_transcripts.end());
    for (uint32_t tid : read_transcripts) {
      const ExonChainMatch *m = find_match(&snap, tid);
      BamInfo bi; bi.is_paired = false; bi.same_transcript = false;
      bi.read1 = this_read->index; bi.r_tid = tid; bi.r_align = m->align;
      bi.r_junc = m->junc_hits; bi.r_refc = m->ref_consumed;
      emit.push_back(bi);
    }
    return;
  }
  ReadInfo rsnap, msnap;
  rsnap.matches = this_read->matches; msnap.matches = mate_read->matches;
  std::vector<uint32_t> read_transcripts, mate_transcripts;
  for (auto &m : this_read->matches) read_transcripts.push_back(m.tid);
  for (auto &m : mate_read->matches) mate_transcripts.push_back(m.tid);
  std::sort(read_transcripts.begin(), read_transcripts.end());
  std::sort(mate_transcripts.begin(), mate_transcripts.end());
  std::vector<uint32_t> common;
  std::set_intersection(read_transcripts.begin(), read_transcripts.end(), mate_transcripts.begin(),
                        mate_transcripts.end(), std::back_inserter(common));
  std::vector<uint32_t> final_transcripts;
  int mate_case;
  if (!common.empty()) { final_transcripts = common; mate_case = 1; }
  else if (read_transcripts.size() == 1 && mate_transcripts.size() == 1) {
    final_transcripts.push_back(read_transcripts[0]);
    final_transcripts.push_back(mate_transcripts[0]);
    mate_case = 2;
  } else return;
  update_read_matches(this_read, final_transcripts);
  update_read_matches(mate_read, final_transcripts);
  if (mate_case == 1) {
    for (uint32_t tid : final_transcripts) {
      const ExonChainMatch *r = find_match(&rsnap, tid), *m = find_match(&msnap, tid);
      BamInfo bi; bi.is_paired = true; bi.same_transcript = true;
      bi.read1 = this_read->index; bi.read2 = mate_read->index;
      bi.r_tid = tid; bi.m_tid = tid; bi.r_align = r->align; bi.m_align = m->align;
      bi.r_junc = r->junc_hits; bi.r_refc = r->ref_consumed;
      bi.m_junc = m->junc_hits; bi.m_refc = m->ref_consumed;
      emit.push_back(bi);
    }
  } else {
    uint32_t r_tid = read_transcripts[0], m_tid = mate_transcripts[0];
    const ExonChainMatch *r = find_match(&rsnap, r_tid), *m = find_match(&msnap, m_tid);
    BamInfo bi; bi.is_paired = true; bi.same_transcript = false;
    bi.read1 = this_read->index; bi.read2 = mate_read->index;
    bi.r_tid = r_tid; bi.m_tid = m_tid; bi.r_align = r->align; bi.m_align = m->align;
    bi.r_junc = r->junc_hits; bi.r_refc = r->ref_consumed;
    bi.m_junc = m->junc_hits; bi.m_refc = m->ref_consumed;
    emit.push_back(bi);
  }
}

}  // namespace orc
