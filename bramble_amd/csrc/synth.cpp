// Synthetic annotation + alignment generator (libbramble_synth.so, g++ only).
// Bench / test input tooling, not part of the projection path: it produces the
// GENCODE-shaped annotation and the name-collated alignment batches that
// SURVEY.md 8(d) specifies (there is no network for real GTF/BAM files).
// Everything is derived from a 64-bit seed; the same seed gives the same bytes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {  // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0; }
  bool chance(double p) { return uni() < p; }
  double normal() { double u1 = uni(), u2 = uni(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
  double lognormal(double median, double sigma) { return median * std::exp(sigma * normal()); }
  uint32_t geometric(double mean) { double p = 1.0 / (mean + 1.0); double u = uni(); if (u < 1e-300) u = 1e-300; return (uint32_t)(std::log(u) / std::log(1.0 - p)); }
};

struct Exon { uint32_t s, e; };  // 1-based half-open

struct Annotation {
  int n_refs = 0;
  std::vector<int32_t> tx_ref; std::vector<int8_t> tx_strand; std::vector<uint32_t> tx_gene;
  std::vector<uint64_t> tx_exon_off{0};
  std::vector<uint32_t> ex_start, ex_end;
  std::vector<uint32_t> ref_len;
  std::vector<std::vector<char>> ref_seq;  // optional genome
  size_t n_tx() const { return tx_ref.size(); }
};

struct AnnParams {
  uint64_t seed; int32_t n_refs; int32_t n_genes; double mean_isoforms; double mean_exons; int32_t max_exons;
  int32_t with_genome;
};

Annotation *gen_annotation(const AnnParams &P) {
  Annotation *A = new Annotation();
  A->n_refs = P.n_refs;
  Rng rng(P.seed);
  int genes_per_ref = std::max(1, P.n_genes / std::max(1, P.n_refs));
  uint32_t gene_id = 0;
  for (int r = 0; r < P.n_refs; r++) {
    uint32_t cursor = 1000 + rng.below(5000);
    uint32_t prev_gene_start = 0, prev_gene_end = 0;
    for (int g = 0; g < genes_per_ref; g++, gene_id++) {
      // master exon chain
      uint32_t m = 1 + rng.geometric(P.mean_exons * 1.05);
      if (gene_id % 4001 == 2000) m = 260 + rng.below(140);  // exercise the uint8 exon_id wrap
      if (m > (uint32_t)P.max_exons) m = P.max_exons;
      uint32_t start;
      if (g > 0 && rng.chance(0.10) && prev_gene_end > prev_gene_start + 200)
        start = prev_gene_start + rng.below(prev_gene_end - prev_gene_start);  // overlapping gene
      else
        start = cursor + (uint32_t)std::min(2.0e6, rng.lognormal(15000.0, 1.0));
      char strand = rng.chance(0.5) ? '+' : '-';
      std::vector<Exon> master;
      uint32_t p = start;
      for (uint32_t k = 0; k < m; k++) {
        uint32_t len = rng.chance(0.05) ? 8 + rng.below(28) : (uint32_t)std::max(20.0, std::min(8000.0, rng.lognormal(130.0, 0.65)));
        if ((k == 0 || k == m - 1) && m > 1) len += (uint32_t)std::min(4000.0, rng.lognormal(300.0, 0.8));  // UTRs
        master.push_back({p, p + len});
        uint32_t intron = (uint32_t)std::max(70.0, std::min(400000.0, rng.lognormal(1500.0, 1.2)));
        p += len + intron;
      }
      uint32_t gene_end = master.back().e;
      uint32_t n_iso = 1 + rng.geometric(P.mean_isoforms - 1.0);
      if (n_iso > 40) n_iso = 40;
      for (uint32_t t = 0; t < n_iso; t++) {
        std::vector<Exon> ex;
        uint32_t a = 0, b = m - 1;
        if (t > 0 && m > 2) {
          if (rng.chance(0.3)) a = rng.below(m / 2 + 1);
          if (rng.chance(0.3)) b = m - 1 - rng.below(m / 2 + 1);
          if (b < a) b = a;
        }
        for (uint32_t k = a; k <= b; k++) {
          if (t > 0 && k > a && k < b && rng.chance(0.10)) continue;  // exon skipping
          ex.push_back(master[k]);
        }
        if (t > 0 && ex.size() > 1 && rng.chance(0.05)) {  // retained intron
          size_t k = rng.below((uint32_t)ex.size() - 1);
          ex[k].e = ex[k + 1].e; ex.erase(ex.begin() + k + 1);
        }
        if (t > 0) {
          if (rng.chance(0.25)) { uint32_t d = 1 + rng.below(60); if (ex[0].e - ex[0].s > d + 10) ex[0].s += d; }
          if (rng.chance(0.25)) { uint32_t d = 1 + rng.below(60); if (ex.back().e - ex.back().s > d + 10) ex.back().e -= d; }
          if (ex.size() > 2 && rng.chance(0.08)) {  // alternative splice site
            size_t k = 1 + rng.below((uint32_t)ex.size() - 2);
            uint32_t d = 3 + rng.below(27);
            if (rng.chance(0.5)) { if (ex[k].e - ex[k].s > d + 10) ex[k].s += d; }
            else { if (ex[k].e - ex[k].s > d + 10) ex[k].e -= d; }
          }
        }
        A->tx_ref.push_back(r); A->tx_strand.push_back((int8_t)strand); A->tx_gene.push_back(gene_id);
        for (auto &e : ex) { A->ex_start.push_back(e.s); A->ex_end.push_back(e.e); }
        A->tx_exon_off.push_back(A->ex_start.size());
      }
      prev_gene_start = start; prev_gene_end = gene_end;
      if (gene_end > cursor) cursor = gene_end;
    }
    A->ref_len.push_back(cursor + 10000);
  }
  if (P.with_genome) {
    A->ref_seq.resize(P.n_refs);
    Rng g(P.seed ^ 0x5eedba5e5ull);
    for (int r = 0; r < P.n_refs; r++) {
      A->ref_seq[r].resize(A->ref_len[r]);
      for (auto &c : A->ref_seq[r]) c = "ACGT"[g.next() >> 62];
    }
  }
  return A;
}

struct ReadParams {
  uint64_t seed; int64_t n_templates; int32_t mode;  // 0 short SE, 1 short PE, 2 long
  int32_t read_len; double frag_mean, frag_sd;
  double p_softclip, p_indel, p_junc_shift, p_intergenic, p_multimap;
  double long_median, long_sigma; int32_t wobble; double p_wobble, p_skip_small, p_novel_small, p_clip;
  int32_t max_clip; int32_t with_seq; int32_t xs_tag;  // xs_tag: emit XS strand tag (stranded)
  int32_t with_records;
};

struct Reads {
  std::vector<int32_t> ref_id, ref_start, mate_ref_id, mate_start, l_qseq;
  std::vector<uint16_t> flags; std::vector<int8_t> xs, ts;
  std::vector<uint64_t> cigar_off{0}, name_off{0}, seq_off{0};
  std::vector<uint32_t> cigar; std::vector<char> names, seqs;
  std::vector<uint32_t> src_tx;  // transcript the template was drawn from (0xffffffff: none)
  // optional: full BAM alignment records (file layout from refID on), for the re-encoding stage
  bool with_records = false;
  std::vector<uint8_t> rec_blob; std::vector<uint64_t> rec_off{0};
  uint64_t rec_rng = 0x243f6a8885a308d3ull;
};

inline uint32_t cg(uint32_t len, uint32_t op) { return (len << 4) | op; }

// spliced interval [f, f+len) of a transcript's genomic-order exon chain -> pos + M/N blocks
void splice_map(const Annotation &A, uint32_t tx, uint32_t f, uint32_t len, uint32_t &pos,
                std::vector<std::pair<uint32_t, uint32_t>> &blocks /* (start,end) genomic half-open */) {
  blocks.clear();
  uint64_t e0 = A.tx_exon_off[tx], e1 = A.tx_exon_off[tx + 1];
  uint32_t acc = 0, need = len;
  for (uint64_t k = e0; k < e1 && need; k++) {
    uint32_t s = A.ex_start[k], e = A.ex_end[k], l = e - s;
    if (f >= acc + l) { acc += l; continue; }
    uint32_t off = f > acc ? f - acc : 0;
    uint32_t take = std::min(need, l - off);
    blocks.push_back({s + off, s + off + take});
    need -= take; acc += l; f = acc;
  }
  pos = blocks.empty() ? 0 : blocks[0].first;
}

uint32_t tx_len(const Annotation &A, uint32_t tx) {
  uint32_t L = 0;
  for (uint64_t k = A.tx_exon_off[tx]; k < A.tx_exon_off[tx + 1]; k++) L += A.ex_end[k] - A.ex_start[k];
  return L;
}

// genome bases of the spliced interval [a, b) of a transcript's exon chain (genomic order)
std::string spliced_seq(const Annotation &A, uint32_t tx, int64_t a, int64_t b) {
  std::string out;
  if (A.ref_seq.empty() || b <= a) return out;
  const std::vector<char> &g = A.ref_seq[A.tx_ref[tx]];
  int64_t acc = 0;
  for (uint64_t k = A.tx_exon_off[tx]; k < A.tx_exon_off[tx + 1]; k++) {
    int64_t s = A.ex_start[k], l = (int64_t)A.ex_end[k] - s;
    int64_t lo = std::max<int64_t>(a, acc), hi = std::min<int64_t>(b, acc + l);
    for (int64_t p = lo; p < hi; p++) out.push_back(g[(size_t)(s + (p - acc) - 1)]);
    acc += l;
  }
  return out;
}

// read sequence in reference orientation for a CIGAR at `pos`: aligned bases from the
// genome (2 % substitutions), insertions random; clips are the transcript's own
// neighbouring bases with probability p_true (rescuable), else random
std::string read_seq(Rng &rng, const Annotation &A, uint32_t tx, uint32_t pos, const std::vector<uint32_t> &cig,
                     int64_t f, int64_t flen, bool lead_true, bool trail_true) {
  std::string out;
  const std::vector<char> &g = A.ref_seq[A.tx_ref[tx]];
  uint64_t rp = pos;
  auto rnd = [&](uint32_t n) { std::string s; for (uint32_t i = 0; i < n; i++) s.push_back("ACGT"[rng.next() >> 62]); return s; };
  for (size_t i = 0; i < cig.size(); i++) {
    uint32_t op = cig[i] & 15, len = cig[i] >> 4;
    if (op == 0 || op == 7 || op == 8) {
      for (uint32_t k = 0; k < len; k++) {
        char c = (rp - 1 + k) < g.size() ? g[(size_t)(rp - 1 + k)] : 'N';
        if (rng.chance(0.02)) c = "ACGT"[rng.next() >> 62];
        out.push_back(c);
      }
      rp += len;
    } else if (op == 1) out += rnd(len);
    else if (op == 2 || op == 3) rp += len;
    else if (op == 4) {
      bool lead = (i == 0);
      std::string t;
      if (lead ? lead_true : trail_true) t = lead ? spliced_seq(A, tx, f - (int64_t)len, f) : spliced_seq(A, tx, f + flen, f + flen + (int64_t)len);
      if (t.size() < len) t = lead ? rnd(len - (uint32_t)t.size()) + t : t + rnd(len - (uint32_t)t.size());
      out += t;
    }
  }
  return out;
}

void push_read(Reads &R, const std::string &name, int32_t ref, uint32_t pos, const std::vector<uint32_t> &cig,
               uint16_t flags, int32_t mref, int32_t mstart, int8_t xs, int8_t ts, uint32_t src,
               const std::string *seq) {
  R.ref_id.push_back(ref); R.ref_start.push_back((int32_t)pos); R.flags.push_back(flags);
  R.mate_ref_id.push_back(mref); R.mate_start.push_back(mstart); R.xs.push_back(xs); R.ts.push_back(ts);
  R.cigar.insert(R.cigar.end(), cig.begin(), cig.end()); R.cigar_off.push_back(R.cigar.size());
  R.names.insert(R.names.end(), name.begin(), name.end()); R.name_off.push_back(R.names.size());
  uint32_t ql = 0;
  for (uint32_t w : cig) { uint32_t op = w & 15; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += w >> 4; }
  R.l_qseq.push_back((int32_t)ql);
  if (seq) R.seqs.insert(R.seqs.end(), seq->begin(), seq->end());
  R.seq_off.push_back(R.seqs.size());
  R.src_tx.push_back(src);
  if (R.with_records) {
    // a plausible aligner record: name, CIGAR, 4-bit sequence, qualities and a mix of aux tags in
    // varying order / integer widths (NH, HI, AS, XS|ts, NM, MD:Z, a B array)
    auto rnd = [&]() { uint64_t z = (R.rec_rng += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    std::vector<uint8_t> &o = R.rec_blob;
    auto p32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) o.push_back((uint8_t)(v >> (8 * k))); };
    auto p16 = [&](uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); };
    uint32_t lq = ql;
    if (flags & 0x100) { if (rnd() % 3 == 0) lq = 0; }  // secondary records often carry no SEQ
    R.l_qseq.back() = (int32_t)lq;                       // the flat table says what the record says
    p32((uint32_t)ref); p32(pos - 1);
    o.push_back((uint8_t)(name.size() + 1)); o.push_back((uint8_t)(rnd() % 61));
    p16((uint32_t)(4681 + (pos >> 14))); p16((uint32_t)cig.size()); p16(flags);
    p32(lq); p32((uint32_t)mref); p32((uint32_t)(mstart - 1)); p32(0);
    o.insert(o.end(), name.begin(), name.end()); o.push_back(0);
    for (uint32_t w : cig) p32(w);
    for (uint32_t i = 0; i < lq; i += 2) {
      auto code = [&](uint32_t k) -> uint8_t { if (k >= lq) return 0; char c = seq ? (*seq)[k] : "ACGT"[rnd() >> 62]; if (rnd() % 97 == 0) c = 'N'; return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; };
      o.push_back((uint8_t)(code(i) << 4 | code(i + 1)));
    }
    bool noqual = rnd() % 50 == 0;
    for (uint32_t i = 0; i < lq; i++) o.push_back(noqual ? 0xff : (uint8_t)(2 + rnd() % 39));
    // aux tags
    int order = (int)(rnd() % 4);
    auto tagC = [&](const char *t, uint32_t v) { o.push_back(t[0]); o.push_back(t[1]); o.push_back('C'); o.push_back((uint8_t)v); };
    auto tagInt = [&](const char *t, int32_t v) {
      o.push_back(t[0]); o.push_back(t[1]);
      switch (rnd() % 4) { case 0: if (v >= 0 && v < 256) { o.push_back('C'); o.push_back((uint8_t)v); break; }
                           case 1: if (v >= -32768 && v < 32768) { o.push_back('s'); p16((uint32_t)(uint16_t)(int16_t)v); break; }
                           case 2: if (v >= 0 && v < 65536) { o.push_back('S'); p16((uint32_t)v); break; }
                           default: o.push_back('i'); p32((uint32_t)v); }
    };
    auto tagA = [&](const char *t, char v) { o.push_back(t[0]); o.push_back(t[1]); o.push_back('A'); o.push_back((uint8_t)v); };
    auto md = [&]() { o.push_back('M'); o.push_back('D'); o.push_back('Z'); uint32_t n = 1 + rnd() % 12; for (uint32_t i = 0; i < n; i++) o.push_back((uint8_t)("0123456789ACGT^"[rnd() % 15])); o.push_back(0); };
    auto barr = [&]() { o.push_back('Z'); o.push_back('B'); o.push_back('B'); o.push_back('S'); uint32_t n = rnd() % 4; p32(n); for (uint32_t i = 0; i < n; i++) p16((uint32_t)(rnd() & 0xffff)); };
    bool has_nh = rnd() % 8 != 0, has_hi = rnd() % 3 == 0, has_as = rnd() % 8 != 0, has_md = rnd() % 2 == 0, has_b = rnd() % 16 == 0;
    if (order == 0) { if (has_as) tagInt("AS", (int32_t)(rnd() % 200)); if (xs) tagA("XS", (char)xs); tagC("NM", rnd() % 5); if (has_nh) tagC("NH", 1 + rnd() % 3); if (ts) tagA("ts", (char)ts); if (has_hi) tagC("HI", 1); if (has_md) md(); }
    else if (order == 1) { if (has_nh) tagInt("NH", 1); if (has_hi) tagInt("HI", 1 + (int32_t)(rnd() % 3)); if (has_md) md(); if (xs) tagA("XS", (char)xs); if (ts) tagA("ts", (char)ts); if (has_as) tagInt("AS", (int32_t)(rnd() % 30000) - 100); }
    else if (order == 2) { if (has_b) barr(); if (ts) tagA("ts", (char)ts); if (xs) tagA("XS", (char)xs); if (has_as) tagInt("AS", (int32_t)(rnd() % 4000)); if (has_nh) tagInt("NH", 2); tagC("NM", rnd() % 9); }
    else { if (has_md) md(); if (has_hi) tagInt("HI", 0); if (ts) tagA("ts", (char)ts); if (xs) tagA("XS", (char)xs); if (has_b) barr(); if (has_nh) tagC("NH", 1); if (has_as) tagInt("AS", 77); if (rnd() % 9 == 0) tagC("NH", 5); }
    R.rec_off.push_back(o.size());
  }
}

// blocks -> CIGAR with optional perturbations (short reads)
void short_cigar(Rng &rng, const ReadParams &P, std::vector<std::pair<uint32_t, uint32_t>> blocks, uint32_t &pos,
                 std::vector<uint32_t> &cig) {
  cig.clear();
  if (blocks.empty()) return;
  if (blocks.size() > 1 && rng.chance(P.p_junc_shift)) {  // junction off by <= 3 bp
    size_t k = rng.below((uint32_t)blocks.size() - 1);
    uint32_t d = 1 + rng.below(3);
    if (rng.chance(0.5)) { if (blocks[k + 1].second - blocks[k + 1].first > d + 1) { blocks[k].second += d; blocks[k + 1].first += d; } }
    else { if (blocks[k].second - blocks[k].first > d + 1) { blocks[k].second -= d; blocks[k + 1].first -= d; } }
  }
  uint32_t lead = 0, trail = 0;
  if (rng.chance(P.p_softclip)) {
    uint32_t k = 1 + rng.below(5);
    if (rng.chance(0.5)) { if (blocks[0].second - blocks[0].first > k + 1) { lead = k; blocks[0].first += k; } }
    else { if (blocks.back().second - blocks.back().first > k + 1) { trail = k; blocks.back().second -= k; } }
  }
  pos = blocks[0].first;
  if (lead) cig.push_back(cg(lead, 4));
  int indel_block = rng.chance(P.p_indel) ? (int)rng.below((uint32_t)blocks.size()) : -1;
  for (size_t k = 0; k < blocks.size(); k++) {
    uint32_t len = blocks[k].second - blocks[k].first;
    if (k) cig.push_back(cg(blocks[k].first - blocks[k - 1].second, 3));
    if ((int)k == indel_block && len > 12) {
      uint32_t at = 4 + rng.below(len - 8), d = 1 + rng.below(3);
      if (rng.chance(0.5)) { cig.push_back(cg(at, 0)); cig.push_back(cg(d, 1)); cig.push_back(cg(len - at, 0)); }
      else if (len - at > d) { cig.push_back(cg(at, 0)); cig.push_back(cg(d, 2)); cig.push_back(cg(len - at - d, 0)); }
      else cig.push_back(cg(len, 0));
    } else {
      cig.push_back(cg(len, 0));
    }
  }
  if (trail) cig.push_back(cg(trail, 4));
}

// long-read CIGAR: junction wobble, skipped small exons, novel small exons, in-block indels, clips
void long_cigar(Rng &rng, const ReadParams &P, std::vector<std::pair<uint32_t, uint32_t>> blocks, uint32_t &pos,
                std::vector<uint32_t> &cig, uint32_t lead_clip, uint32_t trail_clip) {
  cig.clear();
  if (blocks.empty()) return;
  // skipped small middle exons (the read splices over an annotated exon <= 35 bp)
  for (size_t k = 1; k + 1 < blocks.size();) {
    if (blocks[k].second - blocks[k].first <= 35 && rng.chance(P.p_skip_small * 4)) blocks.erase(blocks.begin() + k);
    else k++;
  }
  // novel small exon inside an intron (INS_EXON rule)
  if (blocks.size() > 1 && rng.chance(P.p_novel_small)) {
    size_t k = rng.below((uint32_t)blocks.size() - 1);
    uint32_t gap = blocks[k + 1].first - blocks[k].second;
    if (gap > 200) { uint32_t s = blocks[k].second + 60 + rng.below(gap - 150), l = 8 + rng.below(25); blocks.insert(blocks.begin() + k + 1, {s, s + l}); }
  }
  for (size_t k = 0; k + 1 < blocks.size(); k++) {
    if (!rng.chance(P.p_wobble)) continue;
    int32_t d = (int32_t)rng.below(2 * P.wobble + 1) - P.wobble;
    if (rng.chance(0.5)) { int64_t ne = (int64_t)blocks[k].second + d; if (ne > (int64_t)blocks[k].first + 5 && ne < (int64_t)blocks[k + 1].first - 5) blocks[k].second = (uint32_t)ne; }
    else { int64_t ns = (int64_t)blocks[k + 1].first + d; if (ns > (int64_t)blocks[k].second + 5 && ns + 5 < (int64_t)blocks[k + 1].second) blocks[k + 1].first = (uint32_t)ns; }
  }
  pos = blocks[0].first;
  if (lead_clip) cig.push_back(cg(lead_clip, 4));
  for (size_t k = 0; k < blocks.size(); k++) {
    if (k) cig.push_back(cg(blocks[k].first - blocks[k - 1].second, 3));
    uint32_t len = blocks[k].second - blocks[k].first, done = 0;
    while (done < len) {  // sprinkle sequencing indels: one every ~60 bases
      uint32_t run = 20 + rng.below(90);
      if (run >= len - done) { cig.push_back(cg(len - done, rng.chance(0.1) ? 7u : 0u)); done = len; break; }
      cig.push_back(cg(run, 0)); done += run;
      uint32_t d = 1 + rng.below(3);
      if (rng.chance(0.5)) cig.push_back(cg(d, 1));
      else if (len - done > d + 1) { cig.push_back(cg(d, 2)); done += d; }
    }
  }
  if (trail_clip) cig.push_back(cg(trail_clip, 4));
  // coalesce equal neighbours (a valid BAM CIGAR has none)
  std::vector<uint32_t> out;
  for (uint32_t w : cig) { if (!out.empty() && (out.back() & 15) == (w & 15)) out.back() += (w >> 4) << 4; else out.push_back(w); }
  cig.swap(out);
}

Reads *gen_reads(const Annotation &A, const ReadParams &P) {
  Reads *R = new Reads();
  R->with_records = P.with_records != 0;
  Rng rng(P.seed);
  size_t ntx = A.n_tx();
  std::vector<uint32_t> tlen(ntx);
  for (size_t t = 0; t < ntx; t++) tlen[t] = tx_len(A, (uint32_t)t);
  std::vector<std::pair<uint32_t, uint32_t>> b1, b2;
  std::vector<uint32_t> c1, c2;
  char buf[32];
  for (int64_t i = 0; i < P.n_templates; i++) {
    snprintf(buf, sizeof buf, "r%lld", (long long)i);
    std::string name(buf);
    int n_loc = 1 + ((P.mode != 2 && rng.chance(P.p_multimap)) ? 1 + (int)rng.below(2) : 0);
    for (int loc = 0; loc < n_loc; loc++) {
      uint16_t sec = loc ? 0x100 : 0;
      if (P.mode == 2) {
        uint32_t tx = rng.below((uint32_t)ntx);
        uint32_t L = tlen[tx];
        uint32_t len = (uint32_t)std::min((double)L, std::max(80.0, rng.lognormal(P.long_median, P.long_sigma)));  // never longer than the transcript
        uint32_t f = rng.below(L - len + 1), pos;
        splice_map(A, tx, f, len, pos, b1);
        // soft clips: an aligner that could not place a short terminal exon piece clips it,
        // so the aligned part starts / ends exactly on an exon boundary (the rescuable case);
        // otherwise a clip of arbitrary bases
        uint32_t lead_clip = 0, trail_clip = 0; bool lead_true = false, trail_true = false;
        int64_t fa = f, la = len;
        if (rng.chance(P.p_clip)) {
          uint32_t l0 = b1[0].second - b1[0].first;
          if (b1.size() > 1 && l0 <= (uint32_t)P.max_clip && rng.chance(0.75)) { lead_clip = l0; lead_true = rng.chance(0.8); fa += l0; la -= l0; b1.erase(b1.begin()); }
          else lead_clip = 1 + rng.below(P.max_clip);
        }
        if (rng.chance(P.p_clip)) {
          uint32_t l1 = b1.back().second - b1.back().first;
          if (b1.size() > 1 && l1 <= (uint32_t)P.max_clip && rng.chance(0.75)) { trail_clip = l1; trail_true = rng.chance(0.8); la -= l1; b1.pop_back(); }
          else trail_clip = 1 + rng.below(P.max_clip);
        }
        long_cigar(rng, P, b1, pos, c1, lead_clip, trail_clip);
        uint16_t fl = (rng.chance(0.5) ? 0x10 : 0) | sec;
        int8_t ts = rng.chance(0.7) ? (int8_t)(A.tx_strand[tx]) : 0;
        if (P.with_seq && !A.ref_seq.empty()) {
          std::string sq = read_seq(rng, A, tx, pos, c1, fa, la, lead_true, trail_true);
          push_read(*R, name, A.tx_ref[tx], pos, c1, fl, -1, 0, 0, ts, tx, &sq);
        } else
        push_read(*R, name, A.tx_ref[tx], pos, c1, fl, -1, 0, 0, ts, tx, nullptr);
        continue;
      }
      bool intergenic = rng.chance(P.p_intergenic);
      uint32_t tx = rng.below((uint32_t)ntx);
      int guard = 0;
      while (!intergenic && tlen[tx] < (uint32_t)P.read_len && guard++ < 100) tx = rng.below((uint32_t)ntx);
      if (tlen[tx] < (uint32_t)P.read_len) intergenic = true;
      int32_t ref = A.tx_ref[tx];
      uint32_t p1 = 0, p2 = 0;
      if (intergenic) {
        ref = (int32_t)rng.below((uint32_t)A.n_refs);
        p1 = 1 + rng.below(A.ref_len[ref] - 2000);
        p2 = p1 + 150 + rng.below(200);
        c1.assign(1, cg(P.read_len, 0)); c2 = c1;
        tx = 0xffffffffu;
      } else {
        uint32_t L = tlen[tx];
        uint32_t flen = (uint32_t)std::max((double)P.read_len, std::min((double)L, P.frag_mean + P.frag_sd * rng.normal()));
        uint32_t f = rng.below(L - flen + 1);
        splice_map(A, tx, f, P.read_len, p1, b1);
        splice_map(A, tx, f + flen - P.read_len, P.read_len, p2, b2);
        short_cigar(rng, P, b1, p1, c1);
        short_cigar(rng, P, b2, p2, c2);
      }
      int8_t xs = 0;
      if (P.xs_tag && tx != 0xffffffffu) xs = A.tx_strand[tx];
      if (P.mode == 0) {
        uint16_t fl = (rng.chance(0.5) ? 0x10 : 0) | sec;
        push_read(*R, name, ref, p1, c1, fl, -1, 0, xs, 0, tx, nullptr);
      } else {
        bool swap = rng.chance(0.5);  // which mate is listed first / is read1
        uint16_t fl_left = 0x1 | 0x2 | 0x20 | sec, fl_right = 0x1 | 0x2 | 0x10 | sec;
        fl_left |= swap ? 0x80 : 0x40; fl_right |= swap ? 0x40 : 0x80;
        if (!swap) {
          push_read(*R, name, ref, p1, c1, fl_left, ref, (int32_t)p2, xs, 0, tx, nullptr);
          push_read(*R, name, ref, p2, c2, fl_right, ref, (int32_t)p1, xs, 0, tx, nullptr);
        } else {
          push_read(*R, name, ref, p2, c2, fl_right, ref, (int32_t)p1, xs, 0, tx, nullptr);
          push_read(*R, name, ref, p1, c1, fl_left, ref, (int32_t)p2, xs, 0, tx, nullptr);
        }
      }
    }
  }
  return R;
}

}  // namespace

extern "C" {

void *synth_annotation_new(uint64_t seed, int32_t n_refs, int32_t n_genes, double mean_isoforms,
                           double mean_exons, int32_t max_exons, int32_t with_genome) {
  AnnParams P{seed, n_refs, n_genes, mean_isoforms, mean_exons, max_exons, with_genome};
  return gen_annotation(P);
}
void synth_annotation_free(void *h) { delete (Annotation *)h; }
int64_t synth_annotation_n_tx(void *h) { return (int64_t)((Annotation *)h)->n_tx(); }
int64_t synth_annotation_n_exons(void *h) { return (int64_t)((Annotation *)h)->ex_start.size(); }
const int32_t *synth_annotation_tx_ref(void *h) { return ((Annotation *)h)->tx_ref.data(); }
const int8_t *synth_annotation_tx_strand(void *h) { return ((Annotation *)h)->tx_strand.data(); }
const uint32_t *synth_annotation_tx_gene(void *h) { return ((Annotation *)h)->tx_gene.data(); }
const uint64_t *synth_annotation_tx_exon_off(void *h) { return ((Annotation *)h)->tx_exon_off.data(); }
const uint32_t *synth_annotation_ex_start(void *h) { return ((Annotation *)h)->ex_start.data(); }
const uint32_t *synth_annotation_ex_end(void *h) { return ((Annotation *)h)->ex_end.data(); }
const uint32_t *synth_annotation_ref_len(void *h) { return ((Annotation *)h)->ref_len.data(); }
const char *synth_annotation_ref_seq(void *h, int32_t r) {
  Annotation *A = (Annotation *)h;
  return (size_t)r < A->ref_seq.size() ? A->ref_seq[r].data() : nullptr;
}

struct synth_read_params {
  uint64_t seed; int64_t n_templates; int32_t mode; int32_t read_len; double frag_mean, frag_sd;
  double p_softclip, p_indel, p_junc_shift, p_intergenic, p_multimap;
  double long_median, long_sigma; int32_t wobble; double p_wobble, p_skip_small, p_novel_small, p_clip;
  int32_t max_clip; int32_t with_seq; int32_t xs_tag; int32_t with_records;
};

void *synth_reads_new(void *ann, const synth_read_params *p) {
  ReadParams P{p->seed, p->n_templates, p->mode, p->read_len, p->frag_mean, p->frag_sd, p->p_softclip, p->p_indel,
               p->p_junc_shift, p->p_intergenic, p->p_multimap, p->long_median, p->long_sigma, p->wobble,
               p->p_wobble, p->p_skip_small, p->p_novel_small, p->p_clip, p->max_clip, p->with_seq, p->xs_tag, p->with_records};
  return gen_reads(*(Annotation *)ann, P);
}
void synth_reads_free(void *h) { delete (Reads *)h; }
int64_t synth_reads_n(void *h) { return (int64_t)((Reads *)h)->ref_id.size(); }
const int32_t *synth_reads_ref_id(void *h) { return ((Reads *)h)->ref_id.data(); }
const int32_t *synth_reads_ref_start(void *h) { return ((Reads *)h)->ref_start.data(); }
const int32_t *synth_reads_mate_ref_id(void *h) { return ((Reads *)h)->mate_ref_id.data(); }
const int32_t *synth_reads_mate_start(void *h) { return ((Reads *)h)->mate_start.data(); }
const int32_t *synth_reads_l_qseq(void *h) { return ((Reads *)h)->l_qseq.data(); }
const uint16_t *synth_reads_flags(void *h) { return ((Reads *)h)->flags.data(); }
const int8_t *synth_reads_xs(void *h) { return ((Reads *)h)->xs.data(); }
const int8_t *synth_reads_ts(void *h) { return ((Reads *)h)->ts.data(); }
const uint64_t *synth_reads_cigar_off(void *h) { return ((Reads *)h)->cigar_off.data(); }
const uint32_t *synth_reads_cigar(void *h) { return ((Reads *)h)->cigar.data(); }
const uint64_t *synth_reads_name_off(void *h) { return ((Reads *)h)->name_off.data(); }
const char *synth_reads_names(void *h) { return ((Reads *)h)->names.data(); }
const uint32_t *synth_reads_src_tx(void *h) { return ((Reads *)h)->src_tx.data(); }
const uint64_t *synth_reads_rec_off(void *h) { return ((Reads *)h)->rec_off.data(); }
const uint8_t *synth_reads_rec_blob(void *h) { return ((Reads *)h)->rec_blob.data(); }
// contiguous records -> an uncompressed BAM alignment section: [block_size][record]...; out holds total + 4 n bytes
void synth_frame_records(const uint8_t *blob, const uint64_t *off, int64_t n, uint8_t *out) {
  uint64_t o = 0;
  for (int64_t i = 0; i < n; i++) {
    uint32_t len = (uint32_t)(off[i + 1] - off[i]);
    memcpy(out + o, &len, 4);
    memcpy(out + o + 4, blob + off[i], len);
    o += 4 + (uint64_t)len;
  }
}
const uint64_t *synth_reads_seq_off(void *h) { return ((Reads *)h)->seq_off.data(); }
const char *synth_reads_seqs(void *h) { return ((Reads *)h)->seqs.data(); }

}  // extern "C"
