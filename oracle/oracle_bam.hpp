// ============================================================================
// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's BAM record
// mutation (write_to_bam and helpers), see oracle_core.hpp for the oracle's role.
//
// Follows src/core.cpp:96-212 (write_to_bam), src/bam.cpp:474-528 (update_cigar /
// copy_cigar_memory), :531-588 (set_mate_info), :590-634 (tag setters), :636-702
// (reverse_complement_bam) as in-place edits of one record, the way the reference
// drives htslib.  htslib itself (1.22, subprojects/htslib.wrap) is absent from the
// reference tree: bam_aux_get / bam_aux_del / bam_aux_append / bam_aux2i / bam_dup1 and the
// record layout are restated from the SAM/BAM specification and htslib's published
// behaviour -> "parity unpinned" for those primitives.
// ============================================================================
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace orc {

struct Bam1 {  // bam1_t: core + data (qname | cigar | seq | qual | aux)
  int32_t tid = -1, pos = -1; uint16_t bin = 0; uint8_t qual = 0, l_qname = 0; uint16_t flag = 0;
  uint32_t n_cigar = 0; int32_t l_qseq = 0, mtid = -1, mpos = -1, isize = 0;
  std::vector<uint8_t> data;
  size_t cigar_at() const { return l_qname; }
  size_t seq_at() const { return l_qname + 4u * n_cigar; }
  size_t qual_at() const { return seq_at() + (size_t)((l_qseq > 0 ? l_qseq : 0) + 1) / 2; }
  size_t aux_at() const { return qual_at() + (size_t)(l_qseq > 0 ? l_qseq : 0); }
};

static inline uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static inline void wr32(std::vector<uint8_t> &o, uint32_t v) { for (int k = 0; k < 4; k++) o.push_back((uint8_t)(v >> (8 * k))); }

static inline void bam_tag2cigar(Bam1 &b);
// bam_read1: the record as it sits in the file, then htslib's restore of a CIGAR that was spilled into a CG:B,I tag
// (more than 65535 ops, SAM spec 4.2.2) -- the reference reads every record through it (include/bramble.h:29-85 over
// gclib/GSam.cpp), so bam_get_cigar (gclib/GSam.cpp:197-201) already sees the real ops
static inline Bam1 bam_parse(const uint8_t *r, size_t len) {  // BAM record from refID on
  Bam1 b;
  b.tid = (int32_t)rd32(r); b.pos = (int32_t)rd32(r + 4); b.l_qname = r[8]; b.qual = r[9];
  b.bin = (uint16_t)(r[10] | r[11] << 8); b.n_cigar = (uint32_t)(r[12] | r[13] << 8); b.flag = (uint16_t)(r[14] | r[15] << 8);
  b.l_qseq = (int32_t)rd32(r + 16); b.mtid = (int32_t)rd32(r + 20); b.mpos = (int32_t)rd32(r + 24); b.isize = (int32_t)rd32(r + 28);
  b.data.assign(r + 32, r + len);
  bam_tag2cigar(b);
  return b;
}
// bam_write1: [block_size][record]; a CIGAR of more than 65535 ops is written as the placeholder <l_qseq>S<ref_len>N with
// the real ops in a CG:B,I tag behind everything else (htslib 1.22 sam.c, restated from its published behaviour: the
// source is not in the reference tree -> parity unpinned)
static inline void bam_serialize(const Bam1 &b, std::vector<uint8_t> &o) {
  const bool spill = b.n_cigar > 0xffffu;
  wr32(o, (uint32_t)(32 + b.data.size() + (spill ? 16 : 0)));
  wr32(o, (uint32_t)b.tid); wr32(o, (uint32_t)b.pos);
  o.push_back(b.l_qname); o.push_back(b.qual); o.push_back((uint8_t)b.bin); o.push_back((uint8_t)(b.bin >> 8));
  const uint32_t nc = spill ? 2u : b.n_cigar;
  o.push_back((uint8_t)nc); o.push_back((uint8_t)(nc >> 8)); o.push_back((uint8_t)b.flag); o.push_back((uint8_t)(b.flag >> 8));
  wr32(o, (uint32_t)b.l_qseq); wr32(o, (uint32_t)b.mtid); wr32(o, (uint32_t)b.mpos); wr32(o, (uint32_t)b.isize);
  if (!spill) { o.insert(o.end(), b.data.begin(), b.data.end()); return; }
  uint64_t reflen = 0;   // bam_cigar2rlen: M, D, N, =, X consume the reference
  for (uint32_t k = 0; k < b.n_cigar; k++) {
    const uint32_t w = rd32(&b.data[b.cigar_at() + 4 * (size_t)k]), op = w & 0xf;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += w >> 4;
  }
  o.insert(o.end(), b.data.begin(), b.data.begin() + b.cigar_at());
  wr32(o, ((uint32_t)b.l_qseq << 4) | 4u);             // <l_qseq>S
  wr32(o, ((uint32_t)reflen << 4) | 3u);               // <ref_len>N
  o.insert(o.end(), b.data.begin() + b.seq_at(), b.data.end());
  o.push_back('C'); o.push_back('G'); o.push_back('B'); o.push_back('I');
  wr32(o, b.n_cigar);
  o.insert(o.end(), b.data.begin() + b.cigar_at(), b.data.begin() + b.seq_at());
}

// htslib skip_aux: bytes of the value after the type byte, -1 when malformed
static inline long aux_skip(const std::vector<uint8_t> &d, size_t p) {
  if (p >= d.size()) return -1;
  uint8_t ty = d[p]; size_t v = p + 1, end = d.size();
  switch (ty) {
    case 'A': case 'c': case 'C': return v + 1 <= end ? 1 : -1;
    case 's': case 'S': return v + 2 <= end ? 2 : -1;
    case 'i': case 'I': case 'f': return v + 4 <= end ? 4 : -1;
    case 'd': return v + 8 <= end ? 8 : -1;
    case 'Z': case 'H': { size_t q = v; while (q < end && d[q]) q++; return q < end ? (long)(q - v + 1) : -1; }
    case 'B': {
      if (v + 5 > end) return -1;
      uint8_t st = d[v]; uint32_t n = rd32(&d[v + 1]);
      int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
      if (!sz) return -1;
      uint64_t tot = 5 + (uint64_t)n * sz;
      return v + tot <= end ? (long)tot : -1;
    }
    default: return -1;
  }
}
// bam_aux_get: index of the TYPE byte of the first occurrence of tag, or -1
static inline long aux_get(const Bam1 &b, const char tag[2]) {
  size_t s = b.aux_at();
  while (s + 3 <= b.data.size()) {
    long vl = aux_skip(b.data, s + 2);
    if (vl < 0) return -1;
    if (b.data[s] == (uint8_t)tag[0] && b.data[s + 1] == (uint8_t)tag[1]) return (long)(s + 2);
    s += 3 + (size_t)vl;
  }
  return -1;
}
// htslib bam_tag2cigar (called by bam_read1): when the first CIGAR op is a soft clip of the whole read and a CG tag of
// type B,I (or B,i) with at least n_cigar entries exists, the tag's array IS the CIGAR; it moves into place, the tag goes
static inline void bam_tag2cigar(Bam1 &b) {
  if (b.n_cigar == 0 || b.tid < 0 || b.pos < 0) return;
  if (b.cigar_at() + 4u * (size_t)b.n_cigar > b.data.size()) return;
  const uint32_t w0 = rd32(&b.data[b.cigar_at()]);
  if ((w0 & 0xfu) != 4u || (w0 >> 4) != (uint32_t)b.l_qseq) return;
  if (b.aux_at() > b.data.size()) return;
  const long at = aux_get(b, "CG");
  if (at < 0) return;
  if (b.data[(size_t)at] != 'B' || !(b.data[(size_t)at + 1] == 'I' || b.data[(size_t)at + 1] == 'i')) return;
  const uint32_t cg_len = rd32(&b.data[(size_t)at + 2]);
  if (cg_len < b.n_cigar || cg_len >= (1u << 29)) return;
  std::vector<uint8_t> real(b.data.begin() + (at + 6), b.data.begin() + (at + 6 + 4 * (long)cg_len));
  std::vector<uint8_t> nd(b.data.begin(), b.data.begin() + b.cigar_at());
  nd.insert(nd.end(), real.begin(), real.end());
  nd.insert(nd.end(), b.data.begin() + b.seq_at(), b.data.begin() + (at - 2));                     // SEQ, QUAL, aux in front of the tag
  nd.insert(nd.end(), b.data.begin() + (at + 6 + 4 * (long)cg_len), b.data.end());                 // aux behind it
  b.data.swap(nd);
  b.n_cigar = cg_len;
}
static inline void aux_del(Bam1 &b, long type_at) {
  long vl = aux_skip(b.data, (size_t)type_at);
  b.data.erase(b.data.begin() + (type_at - 2), b.data.begin() + (type_at + 1 + vl));
}
static inline void aux_append(Bam1 &b, const char tag[2], char type, int len, const uint8_t *data) {
  b.data.push_back((uint8_t)tag[0]); b.data.push_back((uint8_t)tag[1]); b.data.push_back((uint8_t)type);
  b.data.insert(b.data.end(), data, data + len);
}
static inline int64_t aux2i(const Bam1 &b, long type_at) {
  const uint8_t *v = &b.data[(size_t)type_at + 1];
  switch (b.data[(size_t)type_at]) {
    case 'c': return (int8_t)v[0]; case 'C': return v[0];
    case 's': return (int16_t)(v[0] | v[1] << 8); case 'S': return (uint16_t)(v[0] | v[1] << 8);
    case 'i': return (int32_t)rd32(v); case 'I': return rd32(v);
    default: return 0;
  }
}

// src/bam.cpp:590-634
static inline void set_int_tag(Bam1 &b, const char tag[2], int32_t v) {
  long at = aux_get(b, tag);
  if (at >= 0) aux_del(b, at);
  uint8_t raw[4]; memcpy(raw, &v, 4);
  aux_append(b, tag, 'i', 4, raw);
}
static inline void del_tag(Bam1 &b, const char tag[2]) { long at = aux_get(b, tag); if (at >= 0) aux_del(b, at); }
static inline void set_as_tag(Bam1 &b, double similarity_score, int clip_score) {
  long at = aux_get(b, "AS");
  int32_t gn_as = 0;
  if (at >= 0) gn_as = (int32_t)aux2i(b, at);
  if (at >= 0) aux_del(b, at);
  double score = (static_cast<double>(gn_as) + static_cast<double>(clip_score)) * similarity_score;
  int32_t new_as = static_cast<int32_t>(score);
  uint8_t raw[4]; memcpy(raw, &new_as, 4);
  aux_append(b, "AS", 'i', 4, raw);
}

// src/bam.cpp:474-528 (update_cigar + copy_cigar_memory): swap the CIGAR bytes
static inline void update_cigar(Bam1 &b, const uint32_t *cig, uint32_t n) {
  std::vector<uint8_t> nd(b.data.begin(), b.data.begin() + b.l_qname);
  for (uint32_t k = 0; k < n; k++) wr32(nd, cig[k]);
  nd.insert(nd.end(), b.data.begin() + b.seq_at(), b.data.end());
  b.data.swap(nd);
  b.n_cigar = n;
}

// src/bam.cpp:636-702
static inline void reverse_complement_bam(Bam1 &b) {
  auto rev_cigar = [&]() {
    uint8_t *c = &b.data[b.cigar_at()];
    for (uint32_t i = 0; i < b.n_cigar / 2; i++)
      for (int k = 0; k < 4; k++) std::swap(c[4 * i + k], c[4 * (b.n_cigar - 1 - i) + k]);
  };
  if (b.l_qseq <= 0) { rev_cigar(); b.flag ^= 0x10; return; }
  int len = b.l_qseq;
  uint8_t *seq = &b.data[b.seq_at()], *qual = &b.data[b.qual_at()];
  static const uint8_t comp_table[16] = {15, 8, 4, 15, 2, 15, 15, 15, 1, 15, 15, 15, 15, 15, 15, 15};
  std::vector<uint8_t> tmp((len + 1) / 2);
  for (int i = 0; i < len; ++i) {
    int j = len - 1 - i;
    uint8_t nt = (seq[j >> 1] >> ((~j & 1) << 2)) & 0xf;                       // bam_seqi
    uint8_t c = comp_table[nt];
    tmp[i >> 1] = (uint8_t)((tmp[i >> 1] & (0xf0 >> ((~i & 1) << 2))) | (c << ((~i & 1) << 2)));  // bam_set_seqi
  }
  memcpy(seq, tmp.data(), (len + 1) / 2);
  if (qual[0] != 0xff) for (int i = 0; i < len / 2; ++i) std::swap(qual[i], qual[len - 1 - i]);
  rev_cigar();
  b.flag ^= 0x10;
}

// src/bam.cpp:531-588; own_strand: the strand this record's transcript lies on (both
// branches at :551-555 end up testing it)
static inline void set_mate_info(Bam1 &b, bool is_paired, bool same_transcript, char own_strand, int32_t mate_tid,
                                 int32_t mate_pos, int32_t isize) {
  if (!is_paired) { b.flag &= ~(0x1 | 0x2 | 0x20); b.mtid = -1; b.mpos = -1; b.isize = 0; return; }
  b.flag |= 0x1;
  if (own_strand == '-') b.flag |= 0x20;
  if (same_transcript) { b.mtid = b.tid; b.mpos = mate_pos; b.flag |= 0x2; b.isize = isize; }
  else { b.mtid = mate_tid; b.mpos = mate_pos; b.isize = 0; b.flag &= ~0x2; }
}


// ---- reader side: what process_reads / process_read_in take from each record ------------
// (src/bramble.cpp:313-441; gclib/GSam.h:310-344; tag_char1 gclib/GSam.cpp:310-318).
// Unmapped records are expected to have been dropped already (bramble.cpp:376-379).
struct ParsedBatch {
  std::vector<int32_t> ref_id, ref_start, mate_ref_id, mate_start, l_qseq;
  std::vector<uint16_t> flags;
  std::vector<int8_t> xs, ts;
  std::vector<uint64_t> cigar_off, name_off, seq_off;
  std::vector<uint32_t> cigar;
  std::string names, seqs;
};

static inline char tag_char1(const Bam1 &b, const char tag[2]) {
  long at = aux_get(b, tag);          // bam_aux_get: the type byte of the first such tag
  if (at < 0) return 0;
  char type = (char)b.data[(size_t)at];
  if (type == 'A' || type == 'Z') return (char)b.data[(size_t)at + 1];
  return 0;
}

static inline void parse_records(const uint8_t *blob, const uint64_t *rec_off, const uint32_t *rec_len, int64_t n,
                                 const int32_t *ref_map, int32_t n_ref_map, ParsedBatch &out) {
  static const char nt16[] = "=ACMGRSVTWYHKDBN";  // htslib seq_nt16_str
  out.cigar_off.push_back(0); out.name_off.push_back(0); out.seq_off.push_back(0);
  for (int64_t i = 0; i < n; i++) {
    size_t len = rec_len ? rec_len[i] : (size_t)(rec_off[i + 1] - rec_off[i]);
    Bam1 b = bam_parse(blob + rec_off[i], len);
    // refid: index of the reference NAME in the annotation's name table (gseqs.addName), handed in as ref_map
    int32_t ref = (b.tid >= 0 && b.tid < n_ref_map) ? ref_map[b.tid] : -1;
    int32_t mref = (b.mtid >= 0 && b.mtid < n_ref_map) ? ref_map[b.mtid] : -1;
    // process_pairs compares the raw header ids (refId() != mate_refId()): keep distinct raw ids distinct
    if (b.tid != b.mtid && ref == mref) mref = -2 - (b.mtid < 0 ? 0 : b.mtid);
    out.ref_id.push_back(ref);
    out.ref_start.push_back(b.pos + 1);                 // GSamRecord::start (1-based)
    out.mate_ref_id.push_back(mref);
    out.mate_start.push_back(b.mpos < 0 ? 0 : b.mpos + 1);  // GSam.h:344
    out.flags.push_back(b.flag);
    out.l_qseq.push_back(b.l_qseq);
    out.xs.push_back((int8_t)tag_char1(b, "XS"));
    out.ts.push_back((int8_t)tag_char1(b, "ts"));
    for (uint32_t k = 0; k < b.n_cigar; k++) out.cigar.push_back(rd32(b.data.data() + b.cigar_at() + 4 * k));
    out.cigar_off.push_back(out.cigar.size());
    out.names.append((const char *)b.data.data(), strnlen((const char *)b.data.data(), b.l_qname));  // bam_get_qname: C string
    out.name_off.push_back(out.names.size());
    int32_t ls = b.l_qseq > 0 ? b.l_qseq : 0;
    for (int32_t k = 0; k < ls; k++) {
      uint8_t byte = b.data[b.seq_at() + (size_t)(k >> 1)];
      out.seqs.push_back(nt16[(k & 1) ? (byte & 0xf) : (byte >> 4)]);  // bam_seqi
    }
    out.seq_off.push_back(out.seqs.size());
  }
}

}  // namespace orc
