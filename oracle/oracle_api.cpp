// ============================================================================
// TEST INFRASTRUCTURE ONLY -- bundle driver + C interface of the CPU oracle
// (see oracle_core.hpp for the oracle's role, pinning status and usage rules).
//
// Follows src/bramble.cpp:272-435 (mate index, bundling), src/threads.cpp:100-162
// (one bundle per worker) and src/core.cpp:60-94,220-427 (convert_reads, flush).
// ============================================================================
#include "oracle.h"
#include "oracle_eval.hpp"
#include "oracle_bam.hpp"

#include <atomic>
#include <chrono>
#include <thread>

using namespace orc;

// ref_cache: test infrastructure convenience, not reference behaviour: a reference sequence handed over with every
// transcript is copied once per reference, not once per transcript
struct orc_index { G2T g2t; std::unordered_map<int32_t, std::string> ref_cache; };

namespace {

struct RowStore {
  std::vector<int32_t> input_index; std::vector<uint32_t> tid, pos; std::vector<int8_t> strand;
  std::vector<uint64_t> cigar_off{0}; std::vector<uint32_t> cigar;
  std::vector<double> sim; std::vector<int32_t> clip_score, junc_hits, ref_consumed;
  std::vector<uint32_t> nh, hi, mapq;
  std::vector<uint8_t> primary, is_paired, same_transcript, is_first;
  std::vector<int32_t> mate_tid, mate_pos, isize; std::vector<uint32_t> group;
  uint64_t total_complete = 0, total_unique = 0, dropped_reads = 0, total_processed = 0;
  void push(const OutRow &r) {
    input_index.push_back(r.input_index); tid.push_back(r.tid); pos.push_back(r.pos);
    strand.push_back((int8_t)r.strand);
    cigar.insert(cigar.end(), r.cigar.begin(), r.cigar.end()); cigar_off.push_back(cigar.size());
    sim.push_back(r.similarity_score); clip_score.push_back(r.clip_score);
    junc_hits.push_back(r.junc_hits); ref_consumed.push_back(r.ref_consumed);
    nh.push_back(r.nh); hi.push_back(r.hi); mapq.push_back(r.mapq);
    primary.push_back(r.primary); is_paired.push_back(r.is_paired);
    same_transcript.push_back(r.same_transcript); is_first.push_back(r.is_first);
    mate_tid.push_back(r.mate_tid); mate_pos.push_back(r.mate_pos); isize.push_back(r.isize);
    group.push_back(r.group);
  }
  void append(const RowStore &o) {
    auto cat = [](auto &a, const auto &b) { a.insert(a.end(), b.begin(), b.end()); };
    uint64_t base = cigar.size();
    cat(input_index, o.input_index); cat(tid, o.tid); cat(pos, o.pos); cat(strand, o.strand);
    cat(cigar, o.cigar);
    for (size_t i = 1; i < o.cigar_off.size(); i++) cigar_off.push_back(base + o.cigar_off[i]);
    cat(sim, o.sim); cat(clip_score, o.clip_score); cat(junc_hits, o.junc_hits);
    cat(ref_consumed, o.ref_consumed); cat(nh, o.nh); cat(hi, o.hi); cat(mapq, o.mapq);
    cat(primary, o.primary); cat(is_paired, o.is_paired); cat(same_transcript, o.same_transcript);
    cat(is_first, o.is_first); cat(mate_tid, o.mate_tid); cat(mate_pos, o.mate_pos);
    cat(isize, o.isize); cat(group, o.group);
    total_complete += o.total_complete; total_unique += o.total_unique;
    dropped_reads += o.dropped_reads; total_processed += o.total_processed;
  }
};

struct MatchStore {
  std::vector<uint64_t> aln_off; std::vector<uint32_t> tid, fwpos, rcpos; std::vector<int8_t> strand;
  std::vector<double> sim, cov, ops; std::vector<int32_t> junc, refc, clip;
  std::vector<uint64_t> ideal_off{0}; std::vector<uint32_t> ideal;
  std::vector<uint64_t> out_off{0}; std::vector<uint32_t> out;
  std::vector<int32_t> n_exons, mate_idx;
};

struct Bundle { int64_t begin, end; };  // alignment index range

std::string name_of(const orc_batch *b, int64_t i) {
  return std::string(b->names + b->name_off[i], b->names + b->name_off[i + 1]);
}

}  // namespace

struct orc_result {
  RowStore rows; MatchStore matches; orc_rows rows_view{}; orc_matches matches_view{};
  double seconds = 0;
};

namespace {

// One bundle: src/bramble.cpp:313-327 (process_read_in) for every alignment,
// then src/core.cpp:220-427 (convert_reads).
void run_bundle(const G2T &g2t, const Flags &flags, const orc_batch *b, Bundle bd, uint32_t group_base,
                RowStore &rows, std::vector<std::vector<ExonChainMatch>> *keep_matches,
                std::vector<int32_t> *keep_nexons, std::vector<int32_t> *keep_mate) {
  Evaluator evaluator(&g2t, flags);
  int64_t n = bd.end - bd.begin;
  std::vector<Read> reads((size_t)n);
  std::vector<std::string> names((size_t)n);
  std::vector<bool> valid((size_t)n, true);
  std::unordered_map<std::string, int> hashread;
  for (int64_t k = 0; k < n; k++) {
    int64_t i = bd.begin + k;
    Read &r = reads[k];
    names[k] = name_of(b, i);
    r.flags = b->flags[i];
    r.strand = read_strand(flags, r.flags, (char)b->xs[i], (char)b->ts[i]);
    r.refid = b->ref_id[i];
    r.start = (uint32_t)b->ref_start[i];
    r.cigar.assign(b->cigar + b->cigar_off[i], b->cigar + b->cigar_off[i + 1]);
    r.l_qseq = b->l_qseq ? b->l_qseq[i] : 0;
    valid[k] = segments_from_cigar(b->ref_start[i] - 1, r.cigar.data(), (uint32_t)r.cigar.size(), r.segs);
    // src/bramble.cpp:272-311 (process_pairs)
    if (r.flags & F_PAIRED) {
      if (b->ref_id[i] == b->mate_ref_id[i]) {
        int32_t read_start = (int32_t)r.start, mate_start = b->mate_start[i];
        std::string mate_key = names[k] + '-' + std::to_string(mate_start);
        auto it = hashread.find(mate_key);
        if (it != hashread.end()) {
          if (reads[k].mate_idx != it->second) reads[k].mate_idx = it->second;
          if (reads[it->second].mate_idx != (int)k) reads[it->second].mate_idx = (int)k;
          hashread.erase(it);
        } else {
          hashread[names[k] + '-' + std::to_string(read_start)] = (int)k;
        }
      }
    }
  }
  if (keep_mate)
    for (int64_t k = 0; k < n; k++)
      (*keep_mate)[bd.begin + k] = reads[k].mate_idx < 0 ? -1 : (int32_t)(bd.begin + reads[k].mate_idx);
  if (keep_nexons)
    for (int64_t k = 0; k < n; k++) (*keep_nexons)[bd.begin + k] = (int32_t)reads[k].segs.size();

  // ---- convert_reads ----
  struct Pending { std::string name; std::vector<BamInfo> pairs; uint32_t group; };
  std::vector<Pending> pending;  // insertion-ordered stand-in for pairs_by_name
  std::unordered_map<std::string, size_t> pending_pos;
  uint32_t n_pairs = 0;
  const uint32_t CHUNK_SIZE = 5000;  // src/core.cpp:27

  auto flush = [&]() {  // src/core.cpp:237-332 + write_to_bam :96-212 (record fields only)
    for (auto &pn : pending) {
      auto &pairs = pn.pairs;
      int best = -1; double best_score = -std::numeric_limits<double>::infinity();
      int count_at_best = 0; int32_t hit_index = 1; int total_matches = 0;
      for (size_t it = 0; it < pairs.size(); ++it) {
        BamInfo &info = pairs[it];
        info.r_align.hit_index = hit_index++; total_matches++;
        if (info.is_paired) { info.m_align.hit_index = hit_index++; total_matches++; }
        double pair_score = info.r_align.similarity_score;
        if (info.is_paired) pair_score = std::max(pair_score, info.m_align.similarity_score);
        if (pair_score > best_score) { best_score = pair_score; best = (int)it; count_at_best = 1; }
        else if (pair_score == best_score) count_at_best++;
      }
      if (best >= 0) {
        if (count_at_best == 1) {
          pairs[best].r_align.primary_alignment = true;
          if (pairs[best].is_paired) pairs[best].m_align.primary_alignment = true;
        } else {
          std::vector<size_t> tied;
          for (size_t it = 0; it < pairs.size(); ++it) {
            double pair_score = pairs[it].r_align.similarity_score;
            if (pairs[it].is_paired) pair_score = std::max(pair_score, pairs[it].m_align.similarity_score);
            if (pair_score == best_score) tied.push_back(it);
          }
          uint64_t seed_key = std::hash<std::string>{}(pn.name);
          int32_t rand_idx = get_rand((uint32_t)tied.size(), seed_key);
          BamInfo &sec = pairs[tied[rand_idx]];
          sec.r_align.primary_alignment = true;
          if (sec.is_paired) sec.m_align.primary_alignment = true;
        }
      }
      uint32_t new_nh = (uint32_t)total_matches;
      uint32_t new_mapq = get_mapq(new_nh, flags.long_reads());
      rows.total_complete += total_matches;
      if (total_matches == 1) rows.total_unique++;
      for (auto &pair : pairs) {
        auto emit_side = [&](bool is_first) {
          const AlignInfo &al = is_first ? pair.r_align : pair.m_align;
          int ridx = is_first ? pair.read1 : pair.read2;
          const Read &rd = reads[ridx];
          OutRow row;
          row.input_index = (int32_t)(bd.begin + ridx);
          row.tid = is_first ? pair.r_tid : pair.m_tid;
          row.strand = al.strand;
          row.pos = (al.strand == '+') ? al.fwpos : al.rcpos;  // src/core.cpp:153-158
          row.cigar = get_new_cigar(rd.cigar.data(), (uint32_t)rd.cigar.size(), al.cigar.ops);
          row.similarity_score = al.similarity_score; row.clip_score = al.clip_score;
          row.junc_hits = is_first ? pair.r_junc : pair.m_junc;
          row.ref_consumed = is_first ? pair.r_refc : pair.m_refc;
          row.nh = new_nh; row.mapq = new_mapq; row.hi = (uint32_t)al.hit_index;
          row.primary = al.primary_alignment; row.is_paired = pair.is_paired;
          row.same_transcript = pair.same_transcript; row.is_first = is_first;
          row.group = pn.group;
          // src/bam.cpp:531-588 (set_mate_info)
          if (!pair.is_paired) { row.mate_tid = -1; row.mate_pos = -1; row.isize = 0; }
          else {
            int32_t r_pos = (pair.r_align.strand == '+') ? (int32_t)pair.r_align.fwpos : (int32_t)pair.r_align.rcpos;
            int32_t m_pos = (pair.m_align.strand == '+') ? (int32_t)pair.m_align.fwpos : (int32_t)pair.m_align.rcpos;
            if (pair.same_transcript) {
              int32_t my_pos = is_first ? r_pos : m_pos, mate_pos = is_first ? m_pos : r_pos;
              row.mate_tid = (int32_t)row.tid; row.mate_pos = mate_pos;
              if (my_pos <= mate_pos) row.isize = (mate_pos + rd.l_qseq) - my_pos;
              else row.isize = -((my_pos + rd.l_qseq) - mate_pos);
            } else {
              row.mate_tid = is_first ? (int32_t)pair.m_tid : (int32_t)pair.r_tid;
              row.mate_pos = is_first ? m_pos : r_pos;
              row.isize = 0;
            }
          }
          rows.push(row);
        };
        emit_side(true);
        if (pair.is_paired) emit_side(false);
      }
    }
    pending.clear(); pending_pos.clear(); n_pairs = 0;
  };

  std::vector<bool> seen((size_t)n, false);
  uint32_t group = group_base;
  for (int64_t id = 0; id < n;) {
    int64_t start = id;
    const std::string &name = names[id];
    // src/core.cpp:353-378: sequence of the first record in the group that has one
    std::string seq;
    bool found = false;
    auto try_seq = [&](int64_t k) {
      if (!flags.use_fasta || found || !b->seq_off) return;
      int64_t i = bd.begin + k;
      if (b->seq_off[i + 1] > b->seq_off[i]) {
        seq.assign(b->seqs + b->seq_off[i], b->seqs + b->seq_off[i + 1]); found = true;
      }
    };
    try_seq(id);
    id++;
    while (id < n && names[id] == name) { try_seq(id); id++; }
    int64_t end = id;

    bool dropped = true;
    std::vector<BamInfo> emitted;
    auto process_read_out = [&](int64_t k) -> ReadInfo * {  // src/core.cpp:60-94
      std::vector<ExonChainMatch> matches;
      if (valid[k]) matches = evaluator.evaluate(reads[k], seq);
      rows.total_processed++;
      if (keep_matches) (*keep_matches)[bd.begin + k] = matches;
      if (matches.empty()) return nullptr;
      ReadInfo *ri = new ReadInfo; ri->matches = std::move(matches); ri->index = (int)k;
      return ri;
    };
    for (int64_t i = start; i < end; i++) {
      if (seen[i]) continue;
      ReadInfo *this_read = process_read_out(i);
      if (this_read) dropped = false;
      if (reads[i].mate_idx < 0) {
        process_mate_pair(this_read, nullptr, emitted);
        delete this_read; seen[i] = true; continue;
      }
      int mate_id = reads[i].mate_idx;
      if (!(mate_id < 0 || mate_id >= n) && !seen[mate_id]) {
        ReadInfo *mate_read = process_read_out(mate_id);
        if (mate_read) dropped = false;
        process_mate_pair(this_read, mate_read, emitted);
        delete mate_read; seen[mate_id] = true;
      }
      delete this_read; seen[i] = true;
    }
    if (dropped) rows.dropped_reads++;
    for (auto &bi : emitted) {  // emit_pair, src/core.cpp:334-341
      const std::string &key = names[bi.read1];
      auto it = pending_pos.find(key);
      if (it == pending_pos.end()) {
        pending_pos[key] = pending.size();
        pending.push_back({key, {}, group});
        it = pending_pos.find(key);
      }
      pending[it->second].pairs.push_back(bi);
      n_pairs++;
    }
    group++;
    if (n_pairs >= CHUNK_SIZE) flush();
  }
  if (!pending.empty()) flush();
}

}  // namespace

extern "C" {

orc_index *orc_index_new(void) { return new orc_index(); }

int64_t orc_index_add_transcript(orc_index *ix, int32_t ref_id, char strand, const char *name,
                                 const uint32_t *exons, int32_t n_exons, const char *ref_seq,
                                 int64_t ref_seq_len) {
  std::vector<GSeg> ex((size_t)n_exons);
  for (int i = 0; i < n_exons; i++) { ex[i].start = exons[2 * i]; ex[i].end = exons[2 * i + 1]; }
  std::stable_sort(ex.begin(), ex.end(), [](const GSeg &a, const GSeg &b) { return a.start < b.start; });
  const std::string *seq = nullptr;
  if (ref_seq) {
    auto it = ix->ref_cache.find(ref_id);
    if (it == ix->ref_cache.end() || (int64_t)it->second.size() != ref_seq_len) it = ix->ref_cache.insert_or_assign(ref_id, std::string(ref_seq, ref_seq + ref_seq_len)).first;
    seq = &it->second;
  }
  return (int64_t)ix->g2t.add_transcript(ref_id, strand, name ? name : "", ex, seq);
}
void orc_index_finish(orc_index *ix) { ix->g2t.finish(); }
int64_t orc_index_num_transcripts(const orc_index *ix) { return (int64_t)ix->g2t.tid_names.size(); }
uint32_t orc_index_transcript_len(const orc_index *ix, int64_t tid) { return ix->g2t.tid_lengths[(size_t)tid]; }
void orc_index_free(orc_index *ix) { delete ix; }

static Flags to_flags(const orc_flags *f) {
  Flags o;
  o.lr = f->lr; o.lr_hq = f->lr_hq; o.strict = f->strict; o.use_fasta = f->use_fasta;
  o.fr = f->fr; o.rf = f->rf;
  o.has_max_clip = f->has_max_clip; o.has_max_junc_ins = f->has_max_junc_ins;
  o.has_max_junc_gap = f->has_max_junc_gap; o.has_sim_thr = f->has_sim_thr;
  o.has_max_error_exon = f->has_max_error_exon;
  o.max_clip = f->max_clip; o.max_junc_ins = f->max_junc_ins; o.max_junc_gap = f->max_junc_gap;
  o.max_error_exon = f->max_error_exon; o.sim_thr = f->sim_thr;
  return o;
}

orc_result *orc_run(const orc_index *ix, const orc_flags *cf, const orc_batch *b, int32_t n_threads,
                    int32_t want_matches) {
  Flags flags = to_flags(cf);
  orc_result *res = new orc_result();
  int64_t n = b->n_aln;
  // src/bramble.cpp:362,396-398: a bundle closes at the first read-name change
  // once it holds >= 100000 alignments.
  std::vector<Bundle> bundles; std::vector<uint32_t> group_base;
  {
    int64_t begin = 0; uint32_t groups = 0, gb = 0;
    for (int64_t i = 0; i < n; i++) {
      bool new_name = (i == begin) ||
          (b->name_off[i + 1] - b->name_off[i] != b->name_off[i] - b->name_off[i - 1]) ||
          memcmp(b->names + b->name_off[i], b->names + b->name_off[i - 1], b->name_off[i + 1] - b->name_off[i]) != 0;
      if (i - begin >= 100000 && new_name) {
        bundles.push_back({begin, i}); group_base.push_back(gb); begin = i; gb = groups;
      }
      if (new_name) groups++;
    }
    if (n > begin) { bundles.push_back({begin, n}); group_base.push_back(gb); }
  }
  std::vector<std::vector<ExonChainMatch>> keep;
  std::vector<int32_t> keep_nexons, keep_mate;
  if (want_matches) { keep.resize((size_t)n); keep_nexons.assign((size_t)n, 0); keep_mate.assign((size_t)n, -1); }
  std::vector<RowStore> per_bundle(bundles.size());
  auto t0 = std::chrono::steady_clock::now();
  if (n_threads <= 1) {
    for (size_t k = 0; k < bundles.size(); k++)
      run_bundle(ix->g2t, flags, b, bundles[k], group_base[k], per_bundle[k],
                 want_matches ? &keep : nullptr, want_matches ? &keep_nexons : nullptr,
                 want_matches ? &keep_mate : nullptr);
  } else {  // src/threads.cpp:114-162: workers pull whole bundles
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
      th.emplace_back([&]() {
        for (;;) {
          size_t k = next.fetch_add(1);
          if (k >= bundles.size()) break;
          run_bundle(ix->g2t, flags, b, bundles[k], group_base[k], per_bundle[k],
                     want_matches ? &keep : nullptr, want_matches ? &keep_nexons : nullptr,
                     want_matches ? &keep_mate : nullptr);
        }
      });
    for (auto &t : th) t.join();
  }
  res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto &rs : per_bundle) res->rows.append(rs);

  RowStore &R = res->rows; orc_rows &V = res->rows_view;
  V.n_rows = (int64_t)R.tid.size();
  V.input_index = R.input_index.data(); V.tid = R.tid.data(); V.pos = R.pos.data();
  V.strand = R.strand.data(); V.cigar_off = R.cigar_off.data(); V.cigar = R.cigar.data();
  V.similarity_score = R.sim.data(); V.clip_score = R.clip_score.data();
  V.junc_hits = R.junc_hits.data(); V.ref_consumed = R.ref_consumed.data();
  V.nh = R.nh.data(); V.hi = R.hi.data(); V.mapq = R.mapq.data();
  V.primary = R.primary.data(); V.is_paired = R.is_paired.data();
  V.same_transcript = R.same_transcript.data(); V.is_first = R.is_first.data();
  V.mate_tid = R.mate_tid.data(); V.mate_pos = R.mate_pos.data(); V.isize = R.isize.data();
  V.group = R.group.data();
  V.total_complete = R.total_complete; V.total_unique = R.total_unique;
  V.dropped_reads = R.dropped_reads; V.total_processed = R.total_processed;

  if (want_matches) {
    MatchStore &M = res->matches;
    M.aln_off.push_back(0);
    for (int64_t i = 0; i < n; i++) {
      std::vector<ExonChainMatch> ms = keep[(size_t)i];
      std::stable_sort(ms.begin(), ms.end(), [](const ExonChainMatch &a, const ExonChainMatch &c) { return a.tid < c.tid; });
      for (auto &m : ms) {
        M.tid.push_back(m.tid); M.fwpos.push_back(m.align.fwpos); M.rcpos.push_back(m.align.rcpos);
        M.strand.push_back((int8_t)m.align.strand); M.sim.push_back(m.align.similarity_score);
        M.cov.push_back(m.total_coverage); M.ops.push_back(m.total_operations);
        M.junc.push_back(m.junc_hits); M.refc.push_back(m.ref_consumed); M.clip.push_back(m.align.clip_score);
        M.ideal.insert(M.ideal.end(), m.align.cigar.ops.begin(), m.align.cigar.ops.end());
        M.ideal_off.push_back(M.ideal.size());
        std::vector<uint32_t> out = get_new_cigar(b->cigar + b->cigar_off[i],
            (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]), m.align.cigar.ops);
        M.out.insert(M.out.end(), out.begin(), out.end());
        M.out_off.push_back(M.out.size());
      }
      M.aln_off.push_back(M.tid.size());
    }
    M.n_exons = keep_nexons; M.mate_idx = keep_mate;
    orc_matches &W = res->matches_view;
    W.n_aln = n; W.n_matches = (int64_t)M.tid.size(); W.aln_off = M.aln_off.data();
    W.tid = M.tid.data(); W.fwpos = M.fwpos.data(); W.rcpos = M.rcpos.data(); W.strand = M.strand.data();
    W.similarity_score = M.sim.data(); W.total_coverage = M.cov.data(); W.total_operations = M.ops.data();
    W.junc_hits = M.junc.data(); W.ref_consumed = M.refc.data(); W.clip_score = M.clip.data();
    W.ideal_off = M.ideal_off.data(); W.ideal = M.ideal.data();
    W.out_off = M.out_off.data(); W.out = M.out.data();
    W.n_exons = M.n_exons.data(); W.mate_idx = M.mate_idx.data();
  }
  return res;
}

const orc_rows *orc_result_rows(const orc_result *r) { return &r->rows_view; }
const orc_matches *orc_result_matches(const orc_result *r) { return &r->matches_view; }
double orc_result_seconds(const orc_result *r) { return r->seconds; }
void orc_result_free(orc_result *r) { delete r; }

int32_t orc_merge_cigar(const uint32_t *real, int32_t n_real, const uint32_t *ideal, int32_t n_ideal,
                        uint32_t *out) {
  std::vector<uint32_t> id(ideal, ideal + n_ideal);
  std::vector<uint32_t> r = get_new_cigar(real, (uint32_t)n_real, id);
  for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
  return (int32_t)r.size();
}

int32_t orc_segments(int32_t ref_start, const uint32_t *cigar, int32_t n_cigar, uint32_t *out_pairs,
                     int32_t cap) {
  std::vector<GSeg> segs;
  if (!segments_from_cigar(ref_start - 1, cigar, (uint32_t)n_cigar, segs)) return -1;
  for (size_t i = 0; i < segs.size() && (int32_t)i < cap; i++) {
    out_pairs[2 * i] = segs[i].start; out_pairs[2 * i + 1] = segs[i].end;
  }
  return (int32_t)segs.size();
}

int32_t orc_resolve_config(const orc_flags *cf, uint32_t *out5, float *thr_out) {
  EvalConfig c = resolve_config(to_flags(cf));
  out5[0] = c.max_clip; out5[1] = c.max_junc_ins; out5[2] = c.max_junc_gap;
  out5[3] = c.max_error_exon; out5[4] = c.ignore_small_exons;
  *thr_out = c.similarity_threshold;
  return c.filter_by_similarity;
}

// write_to_bam (src/core.cpp:96-212) over the rows of a finished run: returns the
// uncompressed BAM stream ([block_size][record] per row) in a malloc'd buffer.
struct orc_parsed { ParsedBatch pb; orc_batch view; };

orc_parsed *orc_bam_parse(const uint8_t *blob, const uint64_t *rec_off, const uint32_t *rec_len, int64_t n,
                          const int32_t *ref_map, int32_t n_ref_map) {
  orc_parsed *p = new orc_parsed();
  parse_records(blob, rec_off, rec_len, n, ref_map, n_ref_map, p->pb);
  ParsedBatch &b = p->pb; orc_batch &v = p->view;
  v.n_aln = n; v.ref_id = b.ref_id.data(); v.ref_start = b.ref_start.data(); v.flags = b.flags.data();
  v.xs = b.xs.data(); v.ts = b.ts.data(); v.cigar_off = b.cigar_off.data(); v.cigar = b.cigar.data();
  v.mate_ref_id = b.mate_ref_id.data(); v.mate_start = b.mate_start.data(); v.name_off = b.name_off.data();
  v.names = b.names.data(); v.seq_off = b.seq_off.data(); v.seqs = b.seqs.data(); v.l_qseq = b.l_qseq.data();
  return p;
}
const orc_batch *orc_parsed_batch(const orc_parsed *p) { return &p->view; }
void orc_parsed_free(orc_parsed *p) { delete p; }

int64_t orc_bam_encode(const orc_result *res, const uint8_t *blob, const uint64_t *rec_off, const uint32_t *rec_len,
                       int64_t n_aln, int32_t long_reads, uint8_t **out) {
  const RowStore &R = res->rows;
  std::vector<Bam1> old((size_t)n_aln);
  std::vector<bool> seen((size_t)n_aln, false);
  std::vector<uint8_t> stream;
  for (size_t r = 0; r < R.tid.size(); r++) {
    size_t i = (size_t)R.input_index[r];
    if (!seen[i]) {  // first use of this read: tags go onto the ORIGINAL record (core.cpp:115-124)
      seen[i] = true;
      old[i] = bam_parse(blob + rec_off[i], rec_len ? (size_t)rec_len[i] : (size_t)(rec_off[i + 1] - rec_off[i]));
      set_int_tag(old[i], "NH", (int32_t)R.nh[r]);
      del_tag(old[i], long_reads ? "ts" : "XS");
    }
    Bam1 b = old[i];  // bam_dup1
    update_cigar(b, R.cigar.data() + R.cigar_off[r], (uint32_t)(R.cigar_off[r + 1] - R.cigar_off[r]));
    b.qual = (uint8_t)R.mapq[r];
    b.tid = (int32_t)R.tid[r];
    if (R.primary[r]) b.flag &= ~0x100; else b.flag |= 0x100;
    char strand = (char)R.strand[r];
    if (strand == '-') reverse_complement_bam(b);
    b.pos = (int32_t)R.pos[r];
    if (long_reads) set_as_tag(b, R.sim[r], R.clip_score[r]);
    set_int_tag(b, "HI", (int32_t)R.hi[r]);
    set_mate_info(b, R.is_paired[r], R.same_transcript[r], strand, R.mate_tid[r], R.mate_pos[r], R.isize[r]);
    bam_serialize(b, stream);
  }
  *out = (uint8_t *)malloc(stream.size() ? stream.size() : 1);
  memcpy(*out, stream.data(), stream.size());
  return (int64_t)stream.size();
}
void orc_free_buffer(uint8_t *p) { free(p); }

uint32_t orc_primary_pick(const char *name, int64_t len, uint32_t n_tied) {
  // src/core.cpp:298-299 with the real libstdc++ facilities
  uint64_t seed_key = std::hash<std::string>{}(std::string(name, name + len));
  return (uint32_t)get_rand(n_tied, seed_key);
}

int32_t orc_ksw_align(const char *tseq, const char *qseq, int32_t *score, int32_t *max, uint32_t *cigar,
                      int32_t cap) {
  Evaluator::KswResult r = Evaluator::align(tseq, qseq, 1, -4, 4, 1, 40);
  *score = r.score; *max = r.max;
  for (size_t i = 0; i < r.cigar.size() && (int32_t)i < cap; i++) cigar[i] = r.cigar[i];
  return (int32_t)r.cigar.size();
}

}  // extern "C"
