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
#include <map>
#include <string>
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

std::string lower(const char *s) { std::string o(s); for (auto &c : o) c = (char)tolower((unsigned char)c); return o; }
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

// GTF attribute: name at a token start, value quoted or bare up to ';'
bool gtf_attr(const char *info, const char *name, std::string &out) {
  size_t nl = strlen(name);
  const char *p = info;
  while (*p) {
    while (*p == ' ' || *p == ';' || *p == '\t') p++;
    if (!*p) break;
    const char *tok = p;
    while (*p && *p != ' ' && *p != ';' && *p != '=') p++;
    bool match = (size_t)(p - tok) == nl && strncmp(tok, name, nl) == 0;
    while (*p == ' ' || *p == '=') p++;
    const char *vs = p, *ve;
    if (*p == '"') { vs = ++p; while (*p && *p != '"') p++; ve = p; if (*p) p++; }
    else { while (*p && *p != ';') p++; ve = p; while (ve > vs && ve[-1] == ' ') ve--; }
    if (match) { out.assign(vs, ve); return true; }
    while (*p && *p != ';') p++;
  }
  return false;
}

// GFF3 attribute "Name=" (case-sensitive, at a field start)
bool gff_attr(const char *info, const char *name_eq, std::string &out) {
  size_t nl = strlen(name_eq);
  const char *p = info;
  while (*p) {
    while (*p == ' ' || *p == ';') p++;
    if (strncmp(p, name_eq, nl) == 0) {
      const char *vs = p + nl, *ve = vs;
      while (*ve && *ve != ';') ve++;
      while (ve > vs && (ve[-1] == ' ' || ve[-1] == '\r')) ve--;
      if (ve - vs >= 2 && *vs == '"' && ve[-1] == '"') { vs++; ve--; }
      out.assign(vs, ve);
      return true;
    }
    while (*p && *p != ';') p++;
  }
  return false;
}

}  // namespace

extern "C" int br_annotation_load(const char *path, br_annotation **out) {
  if (!path || !out) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  gzFile f = gzopen(path, "rb");  // plain text or gzip
  if (!f) { fprintf(stderr, "[bramble_amd] cannot open annotation %s\n", path); return BR_ERR_ANNOTATION; }
  gzbuffer(f, 1 << 20);
  std::vector<Tx> txs;
  std::unordered_map<std::string, size_t> by_key;       // id + '\t' + seqname + strand
  std::unordered_map<std::string, int> feat_level;      // ids of the gene / transcript features read so far -> gff_level
  std::vector<std::string> refnames;
  std::unordered_map<std::string, int> ref_of;
  int fmt = 0;  // 0 unknown, 1 GFF3, 2 GTF
  std::string line;
  std::vector<char> buf(1 << 16);
  auto get_tx = [&](const std::string &id, const char *seq, char strand) -> Tx & {
    std::string key = id; key.push_back('\t'); key += seq;
    auto it = by_key.find(key);
    if (it != by_key.end()) return txs[it->second];
    by_key.emplace(key, txs.size());
    Tx t; t.id = id; t.seqname = seq; t.strand = strand;
    txs.push_back(std::move(t));
    if (!ref_of.count(seq)) { ref_of.emplace(seq, (int)refnames.size()); refnames.push_back(seq); }
    return txs.back();
  };
  for (;;) {
    line.clear();
    bool got = false;
    for (;;) {  // lines of any length
      if (!gzgets(f, buf.data(), (int)buf.size())) break;
      got = true;
      line += buf.data();
      if (!line.empty() && line.back() == '\n') break;
    }
    if (!got) break;
    while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
    if (line.empty() || line[0] == '#') { if (line == "##FASTA") break; continue; }
    char *t[9]; int nt = 0;
    char *s = &line[0];
    t[nt++] = s;
    for (char *p = s; *p && nt < 9; p++) if (*p == '\t') { *p = 0; t[nt++] = p + 1; }
    if (nt < 9) { if (nt < 8) continue; }
    const char *info = nt >= 9 ? t[8] : "";
    char *endp = nullptr;
    unsigned long fs = strtoul(t[3], &endp, 10); if (endp == t[3]) continue;
    unsigned long fe = strtoul(t[4], &endp, 10); if (endp == t[4]) continue;
    if (fe < fs) std::swap(fs, fe);
    char strand = t[6][0];
    if (strand != '+' && strand != '-' && strand != '.') { gzclose(f); fprintf(stderr, "[bramble_amd] bad strand in annotation line\n"); return BR_ERR_ANNOTATION; }
    Kind kind = classify(lower(t[2]));
    if (kind == K_SKIP) continue;
    std::string id, parent;
    if (fmt != 2) {
      bool has_id = gff_attr(info, "ID=", id), has_par = gff_attr(info, "Parent=", parent);
      if (fmt == 0) {
        if (has_id || has_par) fmt = 1;
        else {
          std::string tmp;
          if (gtf_attr(info, "transcript_id", tmp) || gtf_attr(info, "gene_id", tmp)) fmt = 2; else continue;
        }
      }
    }
    if (fmt == 1) {
      // level of a gene / transcript feature: one below the last of its parents that was read before it
      auto level_under = [&](const std::string &parents) -> int {
        int lv = 0;
        size_t a = 0;
        while (a <= parents.size() && !parents.empty()) {
          size_t b = parents.find(',', a);
          if (b == std::string::npos) b = parents.size();
          std::string pid = parents.substr(a, b - a);
          while (!pid.empty() && pid.back() == ' ') pid.pop_back();
          auto it = feat_level.find(pid);
          if (it != feat_level.end()) lv = it->second + 1;
          a = b + 1;
        }
        return lv;
      };
      if (kind == K_GENE) { if (!id.empty()) feat_level.emplace(id, level_under(parent)); continue; }
      if (kind == K_TRANSCRIPT) {
        if (id.empty()) continue;
        Tx &tx = get_tx(id, t[0], strand);
        tx.has_line = true; tx.strand = strand; tx.start = (uint32_t)fs; tx.end = (uint32_t)fe;
        tx.level = level_under(parent);
        feat_level[id] = tx.level;
      } else if (kind == K_EXONLIKE) {
        if (parent.empty()) continue;
        size_t a = 0;
        while (a <= parent.size()) {  // Parent=id1,id2
          size_t b = parent.find(',', a);
          if (b == std::string::npos) b = parent.size();
          std::string pid = parent.substr(a, b - a);
          while (!pid.empty() && pid.back() == ' ') pid.pop_back();
          if (!pid.empty()) {
            Tx &tx = get_tx(pid, t[0], strand);
            if (!tx.has_line && tx.segs.empty()) tx.strand = strand;
            add_segment(tx.segs, (uint32_t)fs, (uint32_t)fe);
          }
          a = b + 1;
        }
      }
    } else {  // GTF: unrecognised features are dropped when only transcripts are loaded (gff.cpp:696-698)
      if (kind == K_OTHER) continue;
      if (kind == K_GENE) {   // its ID is the transcript_id when it has one, else the gene_id (gff.cpp:704-721)
        std::string gid;
        if ((gtf_attr(info, "transcript_id", gid) && !gid.empty()) || (gtf_attr(info, "gene_id", gid) && !gid.empty())) feat_level.emplace(gid, 0);
        continue;
      }
      if (!gtf_attr(info, "transcript_id", id) || id.empty()) continue;
      Tx &tx = get_tx(id, t[0], strand);
      if (kind == K_TRANSCRIPT) {
        tx.has_line = true; tx.strand = strand; tx.start = (uint32_t)fs; tx.end = (uint32_t)fe;
        std::string gid;   // a `transcript` line names its gene as parent (gff.cpp:733-741)
        if (gtf_attr(info, "gene_id", gid)) { auto it = feat_level.find(gid); if (it != feat_level.end()) tx.level = it->second + 1; }
      }
      else { if (!tx.has_line && tx.segs.empty()) tx.strand = strand; add_segment(tx.segs, (uint32_t)fs, (uint32_t)fe); }
    }
  }
  gzclose(f);
  // exonless transcripts get one exon over the feature; ids that only ever appeared as gene features are not transcripts
  std::vector<Tx *> order;
  for (auto &tx : txs) {
    if (tx.segs.empty()) { if (!tx.has_line) continue; tx.segs.push_back({tx.start, tx.end}); }
    tx.start = tx.segs.front().first; tx.end = tx.segs.back().second;
    order.push_back(&tx);
  }
  if (order.empty()) { fprintf(stderr, "[bramble_amd] could not find valid reference transcripts in %s\n", path); return BR_ERR_ANNOTATION; }
  std::stable_sort(order.begin(), order.end(), [](const Tx *a, const Tx *b) {  // gfo_cmpByLoc
    int c = strcmp(a->seqname.c_str(), b->seqname.c_str());
    if (c) return c < 0;
    if (a->start != b->start) return a->start < b->start;
    if (a->level != b->level) return a->level < b->level;
    if (a->end != b->end) return a->end < b->end;
    return strcmp(a->id.c_str(), b->id.c_str()) < 0;
  });
  br_annotation *A = new br_annotation();
  A->refnames = refnames;
  for (Tx *tx : order) {
    A->ids.push_back(tx->id); A->seqnames.push_back(tx->seqname); A->strands.push_back(tx->strand);
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

extern "C" void br_annotation_free(br_annotation *a) { delete a; }
extern "C" size_t br_annotation_num_transcripts(const br_annotation *a) { return a ? a->view.size() : 0; }
extern "C" const br_transcript *br_annotation_transcripts(const br_annotation *a) { return a ? a->view.data() : nullptr; }
extern "C" size_t br_annotation_num_refs(const br_annotation *a) { return a ? a->refnames.size() : 0; }
extern "C" const char *const *br_annotation_refnames(const br_annotation *a) { return a ? a->refname_view.data() : nullptr; }
