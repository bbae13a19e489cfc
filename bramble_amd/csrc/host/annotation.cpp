// GTF / GFF3 guide loader with the reference's transcript order (scope table row f-3).
//
// The reference loads guides with gclib's GffReader(f, transcripts_only=true, sort=true) +
// setRefAlphaSorted() + readAll(keep_attrs, merge_close_exons=false, no_exon_attrs=false)
// (src/bramble.cpp:497-512) and numbers transcripts by their @SQ position in the output header,
// i.e. by gflst order (src/bramble.cpp:559-596; get_tid, include/bramble.h:74-76).  gclib is a
// general GFF toolkit; what is restated here is the part that decides bramble's inputs:
//   * line classification by feature name            gclib/gff.cpp:478-533 (GffLine::GffLine)
//   * GTF ids: transcript_id of transcript / exon-like lines (:733-772); GFF3: ID= / Parent= (:556-565,665-693)
//   * exon-like segments (exon, UTR, CDS, start/stop codon) are merged into the exon list when they
//     overlap OR ABUT (GffObj::addExon / exonOverlapIdx / expandSegment, :963-1079; CDS segments are
//     added to the exons in finalize, :2123-2130)
//   * an exonless transcript gets one exon spanning it (finalize :2079-2086; bramble.cpp:568-576)
//   * transcript bounds become the exon span (:2192-2201)
//   * order: reference NAME (strcmp), start, level, end, strcmp(ID) (gfo_cmpByLoc, :75-90).  level = depth below
//     the features that were on file BEFORE it and that it names as its parent (updateParent, :1436-1441): a GFF3
//     transcript with Parent=<gene or transcript seen earlier> is one below that parent; a GTF `transcript` line whose
//     gene_id names an earlier `gene` line is at level 1 (GffLine sets Parent = gene_id for it, :733-741); a transcript
//     known only from its exon lines, or whose parent comes later or not at all, stays at level 0
// Out of the restated subset (documented in DESIGN.md): BED / TLF input, "*_gene_segment"
// redistribution, discontinuous features that reuse one ID, Ensembl id/version merging
// (procEnsemblID is off in bramble), non-transcript parents promoted by their exon children.
#include <ctype.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/bramble_amd.h"

struct br_annotation {
  std::vector<std::string> refnames;             // order of first appearance (gclib's gseq ids)
  std::vector<std::string> ids, seqnames;
  std::vector<char> strands;
  std::vector<std::vector<br_exon>> exons;       // 1-based half-open [start, end+1), like bramble.cpp:164-165
  std::vector<br_transcript> view;
  std::vector<const char *> refname_view;
};

namespace {

struct Tx {
  std::string id, seqname;
  char strand = '.';
  uint32_t start = 0, end = 0;                   // feature line coordinates (used when exonless)
  bool has_line = false;
  int ref_rank = 0;                              // place of the reference name in strcmp order (for the final sort)
  bool by_exon = false;                          // GffObj::createdByExon: the record began with an exon-like line (never cleared, gff.cpp:1486)
  uint32_t cur_start = 0;                        // GffObj::start while the file is read: what gfoFind measures the locus distance from
  int level = 0;                                 // GffObj::gff_level
  std::vector<std::pair<uint32_t, uint32_t>> segs;  // sorted, merged, 1-based inclusive
};

// GffObj::addExon restated for a sorted segment list: merge with every overlapping or adjacent segment
void add_segment(std::vector<std::pair<uint32_t, uint32_t>> &v, uint32_t s, uint32_t e) {
  if (s > e) std::swap(s, e);
  size_t i = 0;
  while (i < v.size() && (uint64_t)v[i].second + 1 < s) i++;
  if (i == v.size() || v[i].first > (uint64_t)e + 1) { v.insert(v.begin() + (ptrdiff_t)i, {s, e}); return; }
  v[i].first = std::min(v[i].first, s); v[i].second = std::max(v[i].second, e);
  while (i + 1 < v.size() && v[i + 1].first <= (uint64_t)v[i].second + 1) {
    v[i].second = std::max(v[i].second, v[i + 1].second);
    v.erase(v.begin() + (ptrdiff_t)i + 1);
  }
}


bool ends_with(const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
bool starts_with(const std::string &s, const char *pre) { return s.compare(0, strlen(pre), pre) == 0; }

enum Kind { K_SKIP, K_EXONLIKE, K_TRANSCRIPT, K_GENE, K_OTHER };

Kind classify(const std::string &f) {  // gff.cpp:478-533
  if (ends_with(f, "match")) return K_SKIP;
  if (f.find("utr") != std::string::npos) return K_EXONLIKE;
  if (ends_with(f, "exon")) return K_EXONLIKE;
  bool codon_or_cds = f.find("codon") != std::string::npos || f.find("cds") != std::string::npos;
  if (f.find("stop") != std::string::npos && codon_or_cds && f.find("redefined") == std::string::npos &&
      f.find("selenocysteine") == std::string::npos) return K_EXONLIKE;
  if (f.find("start") != std::string::npos && codon_or_cds) return K_EXONLIKE;
  if (f == "cds") return K_EXONLIKE;
  if (starts_with(f, "intron") || ends_with(f, "intron")) return K_OTHER;
  if (ends_with(f, "rna") || ends_with(f, "transcript")) return K_TRANSCRIPT;
  if (ends_with(f, "_gene_segment")) return K_TRANSCRIPT;
  if (ends_with(f, "gene") || starts_with(f, "gene")) return K_GENE;
  return K_OTHER;
}

typedef std::string_view sv;

// GTF attribute: name at a token start, value quoted or bare up to ';' (info = [p, end))
bool gtf_attr(const char *p, const char *end, const char *name, sv &out) {
  size_t nl = strlen(name);
  while (p < end) {
    while (p < end && (*p == ' ' || *p == ';' || *p == '\t')) p++;
    if (p >= end) break;
    const char *tok = p;
    while (p < end && *p != ' ' && *p != ';' && *p != '=') p++;
    bool match = (size_t)(p - tok) == nl && strncmp(tok, name, nl) == 0;
    while (p < end && (*p == ' ' || *p == '=')) p++;
    const char *vs = p, *ve;
    if (p < end && *p == '"') { vs = ++p; while (p < end && *p != '"') p++; ve = p; if (p < end) p++; }
    else { while (p < end && *p != ';') p++; ve = p; while (ve > vs && ve[-1] == ' ') ve--; }
    if (match) { out = sv(vs, (size_t)(ve - vs)); return true; }
    while (p < end && *p != ';') p++;
  }
  return false;
}

// GFF3 attribute "Name=" (case-sensitive, at a field start)
bool gff_attr(const char *p, const char *end, const char *name_eq, sv &out) {
  size_t nl = strlen(name_eq);
  while (p < end) {
    while (p < end && (*p == ' ' || *p == ';')) p++;
    if ((size_t)(end - p) >= nl && strncmp(p, name_eq, nl) == 0) {
      const char *vs = p + nl, *ve = vs;
      while (ve < end && *ve != ';') ve++;
      while (ve > vs && (ve[-1] == ' ' || ve[-1] == '\r')) ve--;
      if (ve - vs >= 2 && *vs == '"' && ve[-1] == '"') { vs++; ve--; }
      out = sv(vs, (size_t)(ve - vs));
      return true;
    }
    while (p < end && *p != ';') p++;
  }
  return false;
}

// One line of the file, taken apart (views into the block it came from).  Taking lines apart is the loader's time (1.9 M
// lines, 170 MB for a GENCODE-sized GTF) and independent line by line: worker threads do it for a block of the file at a
// time; what the lines MEAN depends on the lines before them (first appearance of a reference, features "read before",
// the format the first attribute decides), so the pass that applies them stays sequential, in file order.
struct Rec {
  uint8_t state;      // 0 skipped before any decision (comment, too few fields, no number, "match" features), 1 a bad strand character, 2 usable
  uint8_t kind; char strand;
  bool has_id, has_par, has_tid, has_gid;
  uint32_t fs, fe;
  sv seq, id, parent, tid, gid;
};

void parse_line(const char *b, const char *e, Rec &r) {
  r.state = 0;
  while (e > b && (e[-1] == '\n' || e[-1] == '\r')) e--;
  if (b == e || *b == '#') return;
  const char *t[9], *te[9]; int nt = 0;
  t[nt] = b;
  for (const char *p = b; p < e && nt < 8; p++) if (*p == '\t') { te[nt] = p; nt++; t[nt] = p + 1; }
  te[nt] = e; nt++;                       // (the ninth field runs to the end of the line, tabs included)
  if (nt < 8) return;
  const char *ib = nt >= 9 ? t[8] : e, *ie = e;
  auto number = [](const char *p, const char *pe, unsigned long &v) {   // strtoul(.., 10): leading blanks, optional sign, digits
    while (p < pe && (*p == ' ' || *p == '\t')) p++;
    bool neg = false;
    if (p < pe && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    if (p >= pe || *p < '0' || *p > '9') return false;
    unsigned long x = 0;
    while (p < pe && *p >= '0' && *p <= '9') { x = x * 10 + (unsigned long)(*p - '0'); p++; }
    v = neg ? (unsigned long)(-(long)x) : x;
    return true;
  };
  unsigned long fs, fe;
  if (!number(t[3], te[3], fs) || !number(t[4], te[4], fe)) return;
  if (fe < fs) std::swap(fs, fe);
  const char strand = t[6] < te[6] ? t[6][0] : 0;
  if (strand != '+' && strand != '-' && strand != '.') { r.state = 1; return; }
  std::string f(t[2], (size_t)(te[2] - t[2]));
  for (auto &c : f) c = (char)tolower((unsigned char)c);
  const Kind kind = classify(f);
  if (kind == K_SKIP) return;
  r.state = 2; r.kind = (uint8_t)kind; r.strand = strand; r.fs = (uint32_t)fs; r.fe = (uint32_t)fe;
  r.seq = sv(t[0], (size_t)(te[0] - t[0]));
  r.has_id = gff_attr(ib, ie, "ID=", r.id); r.has_par = gff_attr(ib, ie, "Parent=", r.parent);
  r.has_tid = gtf_attr(ib, ie, "transcript_id", r.tid); r.has_gid = gtf_attr(ib, ie, "gene_id", r.gid);
}

}  // namespace

extern "C" int br_annotation_load_mt(const char *path, int threads, br_annotation **out) {
  if (!path || !out) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  gzFile f = gzopen(path, "rb");  // plain text or gzip
  if (!f) { fprintf(stderr, "[bramble_amd] cannot open annotation %s\n", path); return BR_ERR_ANNOTATION; }
  gzbuffer(f, 1 << 20);
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  std::vector<Tx> txs;
  std::unordered_map<std::string, int> feat_level;      // ids of the gene / transcript features read so far -> gff_level
  std::vector<std::string> refnames;
  std::unordered_map<std::string, int> ref_of;
  int fmt = 0;  // 0 unknown, 1 GFF3, 2 GTF
  // Records are found by ID within a LOCUS (gfoFind, gclib/gff.cpp:1405-1434): same reference, same strand (a record
  // without a strand takes any), start within GFF_MAX_LOCUS = 7 000 000 bases of the record's current start; the first
  // record of the ID, in order of creation, that qualifies.  The same ID further away, or on the other strand, is another
  // transcript (RefSeq-style annotations place one accession at several loci).
  const int64_t MAX_LOCUS = 7000000;
  std::unordered_map<std::string, std::vector<size_t>> by_key;   // id + '\t' + seqname -> its records, oldest first
  std::string key;
  std::vector<size_t> *last_list = nullptr; std::string last_id_s, last_seq_s;   // consecutive lines mostly name the same transcript
  auto list_of = [&](sv id, sv seq) -> std::vector<size_t> & {
    if (last_list && last_id_s == id && last_seq_s == seq) return *last_list;
    key.assign(id.data(), id.size()); key.push_back('\t'); key.append(seq.data(), seq.size());
    last_list = &by_key[key];
    last_id_s.assign(id.data(), id.size()); last_seq_s.assign(seq.data(), seq.size());
    return *last_list;
  };
  auto find_tx = [&](std::vector<size_t> &lst, char strand, uint32_t fstart) -> Tx * {
    for (size_t i : lst) {
      Tx &t = txs[i];
      if (t.strand != '.' && strand != t.strand) continue;
      if (fstart > 0 && std::llabs((int64_t)(int32_t)fstart - (int64_t)(int32_t)t.cur_start) > MAX_LOCUS) continue;
      return &t;
    }
    return nullptr;
  };
  auto new_tx = [&](std::vector<size_t> &lst, sv id, sv seq, char strand, uint32_t fstart) -> Tx & {
    lst.push_back(txs.size());
    Tx t; t.id.assign(id.data(), id.size()); t.seqname.assign(seq.data(), seq.size()); t.strand = strand; t.cur_start = fstart;
    txs.push_back(std::move(t));
    if (!ref_of.count(txs.back().seqname)) { ref_of.emplace(txs.back().seqname, (int)refnames.size()); refnames.push_back(txs.back().seqname); }
    return txs.back();
  };
  // an exon-like line for the transcript `id` (readAll, gff.cpp:1763-1836; readExonFeature, :1503-1533): it joins the record
  // of that ID in its locus; without one, it starts a record of its own ("new GTF-like record starting directly here")
  auto exon_line = [&](sv id, const Rec &r) {
    std::vector<size_t> &lst = list_of(id, r.seq);
    Tx *tx = find_tx(lst, r.strand, r.fs);
    if (!tx) { Tx &n = new_tx(lst, id, r.seq, r.strand, r.fs); n.by_exon = true; tx = &n; }
    else if (tx->strand == '.') tx->strand = r.strand;            // (find_tx lets a line through only on the record's strand, or on none)
    add_segment(tx->segs, r.fs, r.fe);
    tx->cur_start = std::min(tx->cur_start, std::min(r.fs, r.fe));
  };
  // a transcript line with that ID (gff.cpp:1684-1733): it completes a record that exon lines began; a record that already
  // came from a transcript line stays as it is and the line becomes a separate record under the same ID (a discontinuous
  // feature in the GFF3 sense) -- exon lines that follow still go to the first record of the locus
  auto transcript_line = [&](sv id, const Rec &r) -> Tx & {
    std::vector<size_t> &lst = list_of(id, r.seq);
    Tx *tx = find_tx(lst, r.strand, r.fs);
    if (tx && tx->by_exon) { tx->has_line = true; tx->start = r.fs; tx->end = r.fe; tx->cur_start = r.fs; return *tx; }   // updateGffRec (:1484-1501)
    Tx &n = new_tx(lst, id, r.seq, r.strand, r.fs);
    n.has_line = true; n.start = r.fs; n.end = r.fe;
    return n;
  };
  // level of a gene / transcript feature: one below the last of its parents that was read before it
  auto level_under = [&](sv parents) -> int {
    int lv = 0;
    size_t a = 0;
    while (a <= parents.size() && !parents.empty()) {
      size_t b = parents.find(',', a);
      if (b == sv::npos) b = parents.size();
      sv pid = parents.substr(a, b - a);
      while (!pid.empty() && pid.back() == ' ') pid.remove_suffix(1);
      auto it = feat_level.find(std::string(pid));
      if (it != feat_level.end()) lv = it->second + 1;
      a = b + 1;
    }
    return lv;
  };
  // what one usable line means, given everything before it (the body of the old line loop); false: stop with an error
  auto apply = [&](const Rec &r) {
    const Kind kind = (Kind)r.kind;
    if (fmt != 2 && fmt == 0) {
      if (r.has_id || r.has_par) fmt = 1;
      else if (r.has_tid || r.has_gid) fmt = 2;
      else return;
    }
    if (fmt == 1) {
      const sv id = r.has_id ? r.id : sv(), parent = r.has_par ? r.parent : sv();
      if (kind == K_GENE) { if (!id.empty()) feat_level.emplace(std::string(id), level_under(parent)); return; }
      if (kind == K_TRANSCRIPT) {
        if (id.empty()) return;
        Tx &tx = transcript_line(id, r);
        tx.level = level_under(parent);
        feat_level[std::string(id)] = tx.level;
      } else if (kind == K_EXONLIKE) {
        if (parent.empty()) return;
        size_t a = 0;
        while (a <= parent.size()) {  // Parent=id1,id2
          size_t b = parent.find(',', a);
          if (b == sv::npos) b = parent.size();
          sv pid = parent.substr(a, b - a);
          while (!pid.empty() && pid.back() == ' ') pid.remove_suffix(1);
          if (!pid.empty()) exon_line(pid, r);
          a = b + 1;
        }
      }
    } else {  // GTF: unrecognised features are dropped when only transcripts are loaded (gff.cpp:696-698)
      if (kind == K_OTHER) return;
      if (kind == K_GENE) {   // its ID is the transcript_id when it has one, else the gene_id (gff.cpp:704-721)
        if (r.has_tid && !r.tid.empty()) feat_level.emplace(std::string(r.tid), 0);
        else if (r.has_gid && !r.gid.empty()) feat_level.emplace(std::string(r.gid), 0);
        return;
      }
      if (!r.has_tid || r.tid.empty()) return;
      if (kind == K_TRANSCRIPT) {
        Tx &tx = transcript_line(r.tid, r);
        // a `transcript` line names its gene as parent (gff.cpp:733-741)
        if (r.has_gid) { auto it = feat_level.find(std::string(r.gid)); if (it != feat_level.end()) tx.level = it->second + 1; }
      }
      else exon_line(r.tid, r);
    }
  };

  // The file in blocks of whole lines.  A block's lines are taken apart by the worker threads (produce, on a thread of its own:
  // read, cut at "##FASTA", parse) while the lines of the block before it are applied, in order, by this one: what a line
  // means depends on everything in front of it, so the applying stays serial -- it is the longer of the two.
  const size_t BLOCK = 16u << 20;
  double T_read = 0, T_parse = 0, T_apply = 0;
  auto secs_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
  struct Block { std::string data; std::vector<std::vector<Rec>> recs; bool any = false; };
  Block blk[2];
  for (auto &bk : blk) bk.recs.resize((size_t)threads);
  std::string carry;
  bool eof = false, fasta = false;
  // fills B with the next block; B.any = false when the file has ended
  auto produce = [&](Block &B) {
    B.any = false;
    std::string &data = B.data;
    while (!eof && !fasta) {
      auto t0 = std::chrono::steady_clock::now();
      data.assign(carry); carry.clear();
      const size_t had = data.size();
      data.resize(had + BLOCK);
      size_t got = 0;
      while (got < BLOCK) {
        const int n = gzread(f, &data[had + got], (unsigned)std::min<size_t>(BLOCK - got, 1u << 30));
        if (n <= 0) { eof = true; break; }
        got += (size_t)n;
      }
      data.resize(had + got);
      if (!eof) {   // keep the unfinished last line for the next block
        const size_t nl = data.rfind('\n');
        if (nl == std::string::npos) { carry.swap(data); continue; }   // (a line longer than a block: read on)
        carry.assign(data, nl + 1, std::string::npos);
        data.resize(nl + 1);
      }
      if (data.empty()) return;
      // "##FASTA" ends the annotation (the sequence section of a GFF3 file)
      {
        size_t p = 0;
        while ((p = data.find("##FASTA", p)) != std::string::npos) {
          const bool at_start = p == 0 || data[p - 1] == '\n';
          size_t q = p + 7;
          while (q < data.size() && data[q] == '\r') q++;
          if (at_start && (q == data.size() || data[q] == '\n')) { data.resize(p); fasta = true; break; }
          p += 7;
        }
      }
      T_read += secs_since(t0); t0 = std::chrono::steady_clock::now();
      // thread t takes the lines that START in its slice of the block
      const char *base = data.data(), *end = base + data.size();
      std::vector<const char *> cut((size_t)threads + 1);
      cut[0] = base; cut[(size_t)threads] = end;
      for (int t = 1; t < threads; t++) {
        const char *p = base + data.size() * (size_t)t / (size_t)threads;
        if (p < cut[(size_t)t - 1]) p = cut[(size_t)t - 1];
        while (p < end && p > base && p[-1] != '\n') p++;
        cut[(size_t)t] = p;
      }
      auto work = [&](int t) {
        std::vector<Rec> &v = B.recs[(size_t)t];
        v.clear();
        const char *p = cut[(size_t)t], *pe = cut[(size_t)t + 1];
        while (p < pe) {
          const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
          const char *le = nl ? nl + 1 : end;
          Rec r;
          parse_line(p, le, r);
          if (r.state) v.push_back(r);
          p = le;
        }
      };
      if (threads == 1) work(0);
      else {
        std::vector<std::thread> th;
        for (int t = 1; t < threads; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
      }
      T_parse += secs_since(t0);
      B.any = true;
      return;
    }
  };
  {
    int cur = 0;
    produce(blk[0]);
    while (blk[cur].any) {
      std::thread next(produce, std::ref(blk[cur ^ 1]));
      auto t0 = std::chrono::steady_clock::now();
      bool bad_strand = false;
      for (int t = 0; t < threads && !bad_strand; t++)
        for (const Rec &r : blk[cur].recs[(size_t)t]) {
          if (r.state == 1) { bad_strand = true; break; }
          apply(r);
        }
      T_apply += secs_since(t0);
      next.join();
      if (bad_strand) { gzclose(f); fprintf(stderr, "[bramble_amd] bad strand in annotation line\n"); return BR_ERR_ANNOTATION; }
      cur ^= 1;
    }
  }
  gzclose(f);
  if (getenv("BRAMBLE_AMD_TIMING")) fprintf(stderr, "[annotation] read %.3f parse %.3f apply %.3f\n", T_read, T_parse, T_apply);
  // exonless transcripts get one exon over the feature; ids that only ever appeared as gene features are not transcripts
  const auto t_sort0 = std::chrono::steady_clock::now();
  std::vector<Tx *> order;
  for (auto &tx : txs) {
    if (tx.segs.empty()) { if (!tx.has_line) continue; tx.segs.push_back({tx.start, tx.end}); }
    tx.start = tx.segs.front().first; tx.end = tx.segs.back().second;
    order.push_back(&tx);
  }
  if (order.empty()) { fprintf(stderr, "[bramble_amd] could not find valid reference transcripts in %s\n", path); return BR_ERR_ANNOTATION; }
  // (the references' strcmp order as a rank per transcript: 18 comparisons per transcript otherwise each start with a strcmp)
  {
    std::vector<std::string> sorted_refs(refnames);
    std::sort(sorted_refs.begin(), sorted_refs.end(), [](const std::string &a, const std::string &b) { return strcmp(a.c_str(), b.c_str()) < 0; });
    std::unordered_map<std::string, int> rank;
    for (size_t k = 0; k < sorted_refs.size(); k++) rank.emplace(sorted_refs[k], (int)k);
    const std::string *last = nullptr; int last_rank = 0;
    for (Tx *tx : order) {
      if (!last || *last != tx->seqname) { last = &tx->seqname; last_rank = rank[tx->seqname]; }
      tx->ref_rank = last_rank;
    }
  }
  std::stable_sort(order.begin(), order.end(), [](const Tx *a, const Tx *b) {  // gfo_cmpByLoc
    if (a->ref_rank != b->ref_rank) return a->ref_rank < b->ref_rank;
    if (a->start != b->start) return a->start < b->start;
    if (a->level != b->level) return a->level < b->level;
    if (a->end != b->end) return a->end < b->end;
    return strcmp(a->id.c_str(), b->id.c_str()) < 0;
  });
  if (getenv("BRAMBLE_AMD_TIMING")) fprintf(stderr, "[annotation] finish + sort %.3f\n", secs_since(t_sort0));
  br_annotation *A = new br_annotation();
  A->refnames = refnames;
  A->ids.reserve(order.size()); A->seqnames.reserve(order.size()); A->strands.reserve(order.size()); A->exons.reserve(order.size()); A->view.reserve(order.size());
  for (Tx *tx : order) {
    A->ids.push_back(std::move(tx->id)); A->seqnames.push_back(std::move(tx->seqname)); A->strands.push_back(tx->strand);
    std::vector<br_exon> ex;
    for (auto &sg : tx->segs) ex.push_back({sg.first, sg.second + 1});
    A->exons.push_back(std::move(ex));
  }
  for (size_t i = 0; i < A->ids.size(); i++)
    A->view.push_back({A->ids[i].c_str(), A->seqnames[i].c_str(), A->strands[i], A->exons[i].data(), (uint32_t)A->exons[i].size()});
  for (auto &r : A->refnames) A->refname_view.push_back(r.c_str());
  *out = A;
  return BR_OK;
}

extern "C" int br_annotation_load(const char *path, br_annotation **out) {
  unsigned hw = std::thread::hardware_concurrency();
  return br_annotation_load_mt(path, (int)std::min<unsigned>(hw ? hw : 1u, 16u), out);
}

extern "C" void br_annotation_free(br_annotation *a) { delete a; }
extern "C" size_t br_annotation_num_transcripts(const br_annotation *a) { return a ? a->view.size() : 0; }
extern "C" const br_transcript *br_annotation_transcripts(const br_annotation *a) { return a ? a->view.data() : nullptr; }
extern "C" size_t br_annotation_num_refs(const br_annotation *a) { return a ? a->refnames.size() : 0; }
extern "C" const char *const *br_annotation_refnames(const br_annotation *a) { return a ? a->refname_view.data() : nullptr; }
