// C ABI of libbramble_amd.so (include/bramble_amd.h): index build, contexts and
// the HIP projection pipeline.  Host side of the drop-in boundary; the reference
// counterparts are cited per function in the header.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <sys/mman.h>
#include <thread>
#include <functional>
#include <memory>
#include <mutex>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/bramble_amd.h"
#include "device_types.h"
#include "kernels.h"
#include "primary_pick.h"

using namespace br;

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      fprintf(stderr, "[bramble_amd] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e), __FILE__, \
              __LINE__, #expr);                                                               \
      return BR_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

// ---------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------
struct br_index {
  int device = -1;
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
  uint32_t n_refs = 0;
  bool has_seq = false;
  // host copies of the flattened tables
  std::vector<uint32_t> slab_off, s_start, s_pmax, s_tid, tx_first, bin_off;
  std::vector<uint4> t_bin;
  std::vector<uint4> s_row, tx_ex;
  std::vector<uint8_t> seq_pool;
  // device copies
  void *d_slab_off = nullptr, *d_s_start = nullptr, *d_s_pmax = nullptr, *d_bin_off = nullptr, *d_t_bin = nullptr, *d_s_tid = nullptr,
       *d_s_row = nullptr, *d_tx_ex = nullptr, *d_tx_first = nullptr, *d_seq_pool = nullptr;
  size_t device_bytes = 0;
  DevIndex dev{};
};

static int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device >= n) return BR_ERR_NO_DEVICE;
  return BR_OK;
}

template <typename T>
static int upload(void **dst, const std::vector<T> &src, size_t &acc) {
  size_t bytes = std::max<size_t>(src.size() * sizeof(T), 16);
  HIPCHK(hipMalloc(dst, bytes));
  if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  acc += bytes;
  return BR_OK;
}

// Common builder over flat arrays: transcript t has reference tx_ref[t], strand
// tx_strand[t] and exons [tx_exon_off[t], tx_exon_off[t+1]) in (ex_start, ex_end).
static int build_index_flat(size_t n_tx, const int32_t *tx_ref, const int8_t *tx_strand, const uint64_t *tx_exon_off,
                            const uint32_t *ex_start, const uint32_t *ex_end, std::vector<std::string> &&names,
                            size_t n_refs, const std::vector<const br_fasta_seq *> &fasta_by_ref, bool has_seq,
                            int device, br_index **out) {
  if (n_tx >= 0xffffffffull) return BR_ERR_CAPACITY;
  br_index *ix = new br_index();
  ix->n_refs = (uint32_t)n_refs;
  ix->has_seq = has_seq;
  ix->names = std::move(names);
  struct Row { uint32_t start, end, tid, gidx, pos_start; };
  std::vector<std::vector<Row>> slabs(2 * n_refs);
  ix->tx_first.reserve(n_tx + 1);
  std::vector<br_exon> ex;
  std::vector<uint32_t> pos_start;
  for (size_t t = 0; t < n_tx; t++) {
    int32_t refid = tx_ref[t];
    if (refid < 0 || (size_t)refid >= n_refs) { delete ix; return BR_ERR_ANNOTATION; }
    ex.clear();
    for (uint64_t k = tx_exon_off[t]; k < tx_exon_off[t + 1]; k++) ex.push_back({ex_start[k], ex_end[k]});
    std::stable_sort(ex.begin(), ex.end(), [](const br_exon &a, const br_exon &b) { return a.start < b.start; });
    for (size_t i = 0; i < ex.size(); i++) {
      if (ex[i].end < ex[i].start || (i + 1 < ex.size() && ex[i].end > ex[i + 1].start)) {
        fprintf(stderr, "[bramble_amd] transcript '%s': exons overlap or are inverted\n", ix->names[t].c_str());
        delete ix; return BR_ERR_ANNOTATION;
      }
    }
    uint32_t tlen = 0;
    for (auto &e : ex) tlen += e.end - e.start;
    if (tlen == 0) {
      // the reference writes no @SQ line for a transcript of length 0 (src/bramble.cpp:588-596) while tids keep counting
      // every guide: its header and its records would disagree.  GTF coordinates are inclusive, so a loader never makes
      // one; refuse it here instead of numbering around it.
      fprintf(stderr, "[bramble_amd] transcript '%s' has no exonic bases\n", ix->names[t].c_str());
      delete ix; return BR_ERR_ANNOTATION;
    }
    ix->lengths.push_back(tlen);
    ix->tx_first.push_back((uint32_t)ix->tx_ex.size());
    char strand = (char)tx_strand[t];
    bool minus = strand == '-';
    bool stranded = strand == '+' || strand == '-';
    const br_fasta_seq *fa = has_seq ? fasta_by_ref[(size_t)refid] : nullptr;
    // pos_start: cumulative spliced offset in TRANSCRIPT order (src/bramble.cpp:161-175)
    uint32_t acc = 0;
    pos_start.assign(ex.size(), 0);
    if (!minus) { for (size_t i = 0; i < ex.size(); i++) { pos_start[i] = acc; acc += ex[i].end - ex[i].start; } }
    else { for (size_t i = ex.size(); i-- > 0;) { pos_start[i] = acc; acc += ex[i].end - ex[i].start; } }
    for (size_t i = 0; i < ex.size(); i++) {
      uint32_t seq_off = 0;
      if (has_seq) {  // src/g2t.cpp:50-55: upper-cased exon sequence
        seq_off = (uint32_t)ix->seq_pool.size();
        for (uint32_t p = ex[i].start; p < ex[i].end; p++) {
          char ch = (fa && p >= 1 && (uint64_t)(p - 1) < fa->len) ? fa->seq[p - 1] : 'N';
          if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 'a' + 'A');
          ix->seq_pool.push_back((uint8_t)ch);
        }
      }
      ix->tx_ex.push_back(make_uint4(ex[i].start, ex[i].end, pos_start[i], seq_off));
      if (stranded) slabs[2 * (size_t)refid + (minus ? 1 : 0)].push_back({ex[i].start, ex[i].end, (uint32_t)t, (uint32_t)i, pos_start[i]});
    }
    ix->tx_ex.push_back(make_uint4(0xffffffffu, 0xffffffffu, 0, 0));  // sentinel
    if (ix->tx_ex.size() >= 0xfffffff0ull || ix->seq_pool.size() >= 0xfffffff0ull) { delete ix; return BR_ERR_CAPACITY; }
  }
  ix->tx_first.push_back((uint32_t)ix->tx_ex.size());
  for (int pad = 0; pad < 4; pad++) ix->tx_ex.push_back(make_uint4(0xffffffffu, 0xffffffffu, 0, 0));  // prefetch slack
  // The slabs (one per reference and strand) and the bucket tables (one per reference) are independent of each other: their
  // places in the flat arrays are fixed first, then a few host threads sort and fill them side by side (the command line
  // waits for this between the guide loader and the first bundle).
  auto for_each_par = [](size_t n, const std::function<void(size_t)> &fn) {
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::min<size_t>(n, std::max(1u, std::min(hw ? hw : 1u, 16u)));
    if (nt <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    auto body = [&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back(body);
    body();
    for (auto &x : th) x.join();
  };
  ix->slab_off.assign(slabs.size() + 1, 0);
  {
    uint64_t tot = 0;
    for (size_t k = 0; k < slabs.size(); k++) { tot += slabs[k].size(); if (tot >= 0xfffffff0ull) { delete ix; return BR_ERR_CAPACITY; } ix->slab_off[k + 1] = (uint32_t)tot; }
    ix->s_start.resize((size_t)tot); ix->s_pmax.resize((size_t)tot); ix->s_tid.resize((size_t)tot); ix->s_row.resize((size_t)tot * 2);
  }
  for_each_par(slabs.size(), [&](size_t k) {
    auto &rows = slabs[k];
    std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.start < b.start; });
    uint32_t m = 0;
    size_t at = ix->slab_off[k];
    for (auto &r : rows) {
      m = std::max(m, r.end);
      ix->s_start[at] = r.start; ix->s_pmax[at] = m;
      // one 32-byte row = everything a candidate lane needs, in one 64-byte sector
      ix->s_row[2 * at] = make_uint4(r.start, r.end, ix->tx_ex[ix->tx_first[r.tid] + r.gidx + 1].x /* sentinel start = ~0u */, r.pos_start);
      const uint32_t n_ex = ix->tx_first[r.tid + 1] - ix->tx_first[r.tid] - 1;  // rows of the transcript minus its sentinel
      ix->s_row[2 * at + 1] = make_uint4(r.tid, r.gidx | (n_ex > 256u ? 0x80000000u : 0u), ix->tx_first[r.tid], ix->tx_ex[ix->tx_first[r.tid] + r.gidx + 1].y /* next exon's end */);
      ix->s_tid[at] = r.tid;
      at++;
    }
  });
  // bucket tables (replace the per-read binary search of the slab)
  const uint32_t SHIFT = 10;
  std::vector<uint64_t> n_bins(ix->n_refs, 2);
  ix->bin_off.assign((size_t)ix->n_refs + 1, 0);
  {
    uint64_t tot = 0;
    for (uint32_t r = 0; r < ix->n_refs; r++) {
      uint64_t nb = 2;
      for (int sd = 0; sd < 2; sd++) {
        uint32_t sb = ix->slab_off[2 * r + sd], se = ix->slab_off[2 * r + sd + 1];
        if (se > sb) nb = std::max<uint64_t>(nb, ((uint64_t)std::max(ix->s_start[se - 1], ix->s_pmax[se - 1]) >> SHIFT) + 2);
      }
      n_bins[r] = nb;
      tot += nb + 1;
      if (tot >= 0xfffffff0ull) { delete ix; return BR_ERR_CAPACITY; }
      ix->bin_off[r + 1] = (uint32_t)tot;
    }
    ix->t_bin.resize((size_t)tot);
  }
  for_each_par(ix->n_refs, [&](size_t r) {
    const uint64_t nb = n_bins[r];
    const size_t first = ix->bin_off[r];
    for (int sd = 0; sd < 2; sd++) {
      uint32_t sb = ix->slab_off[2 * r + sd], se = ix->slab_off[2 * r + sd + 1];
      uint32_t rh = sb, rl = sb;
      for (uint64_t bk = 0; bk <= nb; bk++) {
        if (bk < nb) {
          uint64_t edge = bk << SHIFT;
          while (rh < se && (uint64_t)ix->s_start[rh] < edge) rh++;
          while (rl < se && (uint64_t)ix->s_pmax[rl] <= edge) rl++;
        } else rh = rl = se;
        uint4 &e = ix->t_bin[first + bk];
        if (sd == 0) { e.x = rl; e.y = rh; } else { e.z = rl; e.w = rh; }
      }
    }
  });
  ix->device = device;
  if (device >= 0) {
    int rc = check_device(device);
    if (rc != BR_OK) { delete ix; return rc; }
    HIPCHK(hipSetDevice(device));
    size_t acc = 0;
    if ((rc = upload(&ix->d_slab_off, ix->slab_off, acc)) || (rc = upload(&ix->d_s_start, ix->s_start, acc)) ||
        (rc = upload(&ix->d_s_pmax, ix->s_pmax, acc)) || (rc = upload(&ix->d_bin_off, ix->bin_off, acc)) ||
        (rc = upload(&ix->d_t_bin, ix->t_bin, acc)) ||
        (rc = upload(&ix->d_s_tid, ix->s_tid, acc)) ||
        (rc = upload(&ix->d_s_row, ix->s_row, acc)) || (rc = upload(&ix->d_tx_ex, ix->tx_ex, acc)) ||
        (rc = upload(&ix->d_tx_first, ix->tx_first, acc)) || (rc = upload(&ix->d_seq_pool, ix->seq_pool, acc))) {
      br_index_free(ix); return rc;
    }
    ix->device_bytes = acc;
    DevIndex &d = ix->dev;
    d.n_refs = ix->n_refs; d.n_tx = (uint32_t)n_tx; d.n_rows = (uint32_t)ix->s_start.size();
    d.slab_off = (const uint32_t *)ix->d_slab_off; d.s_start = (const uint32_t *)ix->d_s_start;
    d.s_pmax = (const uint32_t *)ix->d_s_pmax;
    d.bin_shift = SHIFT; d.bin_off = (const uint32_t *)ix->d_bin_off;
    d.t_bin = (const uint4 *)ix->d_t_bin;
    d.s_tid = (const uint32_t *)ix->d_s_tid;
    d.s_row = (const uint4 *)ix->d_s_row; d.tx_ex = (const uint4 *)ix->d_tx_ex;
    d.tx_first = (const uint32_t *)ix->d_tx_first; d.seq_pool = (const uint8_t *)ix->d_seq_pool;
  }
  *out = ix;
  return BR_OK;
}

extern "C" int br_index_build(const br_transcript *transcripts, size_t n_transcripts,
                              const char *const *refnames, size_t n_refnames, const br_fasta_seq *fasta,
                              size_t n_fasta, int device, br_index **out) {
  if (!out || (!transcripts && n_transcripts) || (!refnames && n_refnames)) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  std::unordered_map<std::string, uint32_t> ref_of;
  for (size_t i = 0; i < n_refnames; i++) ref_of.emplace(refnames[i], (uint32_t)i);
  std::vector<const br_fasta_seq *> fa_by_ref(n_refnames, nullptr);
  for (size_t i = 0; i < n_fasta; i++) { auto it = ref_of.find(fasta[i].name ? fasta[i].name : ""); if (it != ref_of.end()) fa_by_ref[it->second] = &fasta[i]; }
  std::vector<int32_t> tx_ref(n_transcripts); std::vector<int8_t> tx_strand(n_transcripts);
  std::vector<uint64_t> off(n_transcripts + 1, 0); std::vector<uint32_t> es, ee; std::vector<std::string> names;
  for (size_t t = 0; t < n_transcripts; t++) {
    const br_transcript &tx = transcripts[t];
    names.emplace_back(tx.id ? tx.id : "");
    auto it = ref_of.find(tx.seqname ? tx.seqname : "");
    if (it == ref_of.end()) {  // bramble-rs/src/g2t.rs:442-444
      fprintf(stderr, "[bramble_amd] reference '%s' of transcript '%s' not in refnames\n",
              tx.seqname ? tx.seqname : "", tx.id ? tx.id : "");
      return BR_ERR_ANNOTATION;
    }
    tx_ref[t] = (int32_t)it->second; tx_strand[t] = (int8_t)tx.strand;
    for (uint32_t k = 0; k < tx.n_exons; k++) { es.push_back(tx.exons[k].start); ee.push_back(tx.exons[k].end); }
    off[t + 1] = es.size();
  }
  return build_index_flat(n_transcripts, tx_ref.data(), tx_strand.data(), off.data(), es.data(), ee.data(),
                          std::move(names), n_refnames, fa_by_ref, n_fasta > 0, device, out);
}

extern "C" int br_index_build_flat(size_t n_tx, const int32_t *tx_ref_id, const int8_t *tx_strand,
                                   const uint64_t *tx_exon_off, const uint32_t *ex_start, const uint32_t *ex_end,
                                   const char *const *tx_names, size_t n_refs, const br_fasta_seq *fasta_by_ref,
                                   int device, br_index **out) {
  if (!out || (n_tx && (!tx_ref_id || !tx_strand || !tx_exon_off))) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  std::vector<std::string> names(n_tx);
  for (size_t t = 0; t < n_tx; t++) names[t] = tx_names ? tx_names[t] : ("tx" + std::to_string(t));
  std::vector<const br_fasta_seq *> fa(n_refs, nullptr);
  if (fasta_by_ref) for (size_t r = 0; r < n_refs; r++) fa[r] = fasta_by_ref[r].seq ? &fasta_by_ref[r] : nullptr;
  return build_index_flat(n_tx, tx_ref_id, tx_strand, tx_exon_off, ex_start, ex_end, std::move(names), n_refs, fa,
                          fasta_by_ref != nullptr, device, out);
}

extern "C" void br_index_free(br_index *ix) {
  if (!ix) return;
  if (ix->device >= 0) {
    (void)hipSetDevice(ix->device);
    void *ptrs[] = {ix->d_slab_off, ix->d_s_start, ix->d_s_pmax, ix->d_bin_off, ix->d_t_bin, ix->d_s_tid, ix->d_s_row, ix->d_tx_ex,
                    ix->d_tx_first, ix->d_seq_pool};
    for (void *p : ptrs) if (p) (void)hipFree(p);
  }
  delete ix;
}
extern "C" size_t br_index_num_transcripts(const br_index *ix) { return ix ? ix->names.size() : 0; }
extern "C" const char *br_index_transcript_name(const br_index *ix, uint32_t tid) {
  return (ix && tid < ix->names.size()) ? ix->names[tid].c_str() : nullptr;
}
extern "C" int64_t br_index_transcript_len(const br_index *ix, uint32_t tid) {
  return (ix && tid < ix->lengths.size()) ? (int64_t)ix->lengths[tid] : -1;
}
extern "C" size_t br_index_num_refs(const br_index *ix) { return ix ? ix->n_refs : 0; }
extern "C" size_t br_index_num_intervals(const br_index *ix) { return ix ? ix->s_start.size() : 0; }
extern "C" size_t br_index_device_bytes(const br_index *ix) { return ix ? ix->device_bytes : 0; }

// ---------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------
extern "C" void br_config_short_read(br_config *c) { memset(c, 0, sizeof(*c)); c->junc_miss_discount = 1.0; }
extern "C" void br_config_long_read(br_config *c) { memset(c, 0, sizeof(*c)); c->lr = 1; c->junc_miss_discount = 1.0; }

// src/evaluate.cpp:1136-1221: presets (+ overrides).  The LR branch comes before
// LR_HQ, and --strict only reaches short-read runs.
extern "C" int br_config_resolve(const br_config *c, br_thresholds *t) {
  if (!c || !t) return BR_ERR_INVALID_ARG;
  if (c->junc_miss_discount != 1.0 && c->junc_miss_discount != 0.0) return BR_ERR_UNSUPPORTED;
  bool longr = c->lr || c->lr_hq;
  uint32_t mc, mji, mjg, mee; float thr;
  if (!longr) { mc = c->strict ? 0 : 5; mji = 0; mjg = 0; thr = 1.0f; mee = 0; }
  else if (c->lr) { mc = 40; mji = 40; mjg = 40; thr = (float)0.60; mee = 35; }
  else { mc = 5; mji = 10; mjg = 10; thr = (float)0.90; mee = 35; }
  if (c->has_max_clip) mc = c->max_clip;
  if (c->has_max_junc_ins) mji = c->max_junc_ins;
  if (c->has_max_junc_gap) mjg = c->max_junc_gap;
  if (c->has_sim_thr) thr = c->sim_thr;
  if (c->has_max_error_exon) mee = c->max_error_exon;
  t->max_clip = mc; t->max_junc_ins = mji; t->max_junc_gap = mjg; t->max_error_exon = mee;
  t->ignore_small_exons = mee > 0; t->similarity_threshold = thr;
  t->filter_by_similarity = thr < 1.0;
  return BR_OK;
}

static int make_devcfg(const br_config *c, DevCfg &d) {
  br_thresholds t;
  int rc = br_config_resolve(c, &t);
  if (rc) return rc;
  d.max_clip = t.max_clip; d.max_junc_ins = t.max_junc_ins; d.max_junc_gap = t.max_junc_gap;
  d.max_error_exon = t.max_error_exon; d.ignore_small_exons = t.ignore_small_exons;
  d.filter_by_similarity = t.filter_by_similarity; d.long_reads = (c->lr || c->lr_hq) ? 1 : 0;
  d.use_fasta = c->use_fasta ? 1 : 0; d.fr = c->fr ? 1 : 0; d.rf = c->rf ? 1 : 0;
  d.thr = (double)t.similarity_threshold;  // float widened to double (include/evaluate.h:281)
  return BR_OK;
}

// ---------------------------------------------------------------------------
// host-side input contract: name groups + mate index
// ---------------------------------------------------------------------------
extern "C" int br_batch_prepare(const br_batch *b, int32_t *mate_idx, uint32_t *group_off, int64_t *n_groups) {
  if (!b || !mate_idx || !group_off || !n_groups) return BR_ERR_INVALID_ARG;
  int64_t n = b->n_aln;
  if (n >= 0x7fffffffll) return BR_ERR_CAPACITY;
  int64_t ng = 0;
  for (int64_t i = 0; i < n; i++) mate_idx[i] = -1;
  std::vector<std::pair<int32_t, int32_t>> open_small;  // (ref_start, index) of still-unpaired records
  std::unordered_map<int32_t, int32_t> open_big;
  int64_t i = 0;
  while (i < n) {
    int64_t j = i + 1;
    uint64_t len = b->name_off[i + 1] - b->name_off[i];
    const char *nm = b->names + b->name_off[i];
    while (j < n && b->name_off[j + 1] - b->name_off[j] == len && memcmp(b->names + b->name_off[j], nm, len) == 0) j++;
    group_off[ng++] = (uint32_t)i;
    // src/bramble.cpp:272-311 (process_pairs): key = name + '-' + start; within a
    // name group the name part is constant, so the key is the start.
    bool big = (j - i) > 32;
    open_small.clear();
    if (big) open_big.clear();
    for (int64_t k = i; k < j; k++) {
      if (!(b->flags[k] & 0x1)) continue;
      if (b->ref_id[k] != b->mate_ref_id[k]) continue;
      int32_t ms = b->mate_start[k], rs = b->ref_start[k];
      int32_t found = -1;
      if (big) {
        auto it = open_big.find(ms);
        if (it != open_big.end()) { found = it->second; open_big.erase(it); }
      } else {
        for (size_t q = 0; q < open_small.size(); q++)
          if (open_small[q].first == ms) { found = open_small[q].second; open_small.erase(open_small.begin() + q); break; }
      }
      if (found >= 0) {
        mate_idx[k] = found; mate_idx[found] = (int32_t)k;
      } else if (big) {
        open_big[rs] = (int32_t)k;
      } else {
        bool repl = false;
        for (auto &p : open_small) if (p.first == rs) { p.second = (int32_t)k; repl = true; break; }
        if (!repl) open_small.emplace_back(rs, (int32_t)k);
      }
    }
    i = j;
  }
  group_off[ng] = (uint32_t)n;
  *n_groups = ng;
  return BR_OK;
}

extern "C" int br_batch_seq_source(const br_batch *b, const uint32_t *group_off, int64_t n_groups, int32_t *seq_src) {
  if (!b || !group_off || !seq_src) return BR_ERR_INVALID_ARG;
  for (int64_t g = 0; g < n_groups; g++) {
    int32_t src = -1;
    if (b->seq_off)
      for (uint32_t i = group_off[g]; i < group_off[g + 1]; i++)
        if (b->seq_off[i + 1] > b->seq_off[i]) { src = (int32_t)i; break; }
    for (uint32_t i = group_off[g]; i < group_off[g + 1]; i++) seq_src[i] = src;
  }
  return BR_OK;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Large pinned host buffers: an anonymous mapping on transparent huge pages, touched, then registered -- 10-12 ms for 250 MB
// and the same for two threads at once, where hipHostMalloc takes 33-41 ms and 83-101 ms for the second of two concurrent
// calls (profiles/pin_probe.cpp: the command line's workers all pin their download buffers when their first bundles finish)
struct BigPinned {
  uint8_t *p = nullptr; size_t cap = 0; void *map = nullptr; size_t map_bytes = 0; bool registered = false;
  int alloc(size_t bytes) {
    release();
    const size_t huge = (size_t)2 << 20;
    map_bytes = ((bytes + huge - 1) & ~(huge - 1)) + huge;
    map = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (map == MAP_FAILED) { map = nullptr; map_bytes = 0; return BR_ERR_CAPACITY; }
    p = (uint8_t *)(((uintptr_t)map + huge - 1) & ~(uintptr_t)(huge - 1));
    const size_t span = (bytes + huge - 1) & ~(huge - 1);
    (void)madvise(p, span, MADV_HUGEPAGE);
    for (size_t i = 0; i < span; i += 4096) p[i] = 0;
    if (hipHostRegister(p, span, hipHostRegisterDefault) != hipSuccess) {   // (no registration: the plain way)
      (void)hipGetLastError();
      munmap(map, map_bytes); map = nullptr; map_bytes = 0; p = nullptr;
      HIPCHK(hipHostMalloc((void **)&p, bytes, hipHostMallocDefault));
      registered = false; cap = bytes;
      return BR_OK;
    }
    registered = true; cap = span;
    return BR_OK;
  }
  void release() {
    if (p && registered) { (void)hipHostUnregister(p); munmap(map, map_bytes); }
    else if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0; map = nullptr; map_bytes = 0; registered = false;
  }
};

struct DevBuf {
  void *p = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return BR_OK;
    if (p) { HIPCHK(hipFree(p)); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    return BR_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T *as() { return (T *)p; }
};

// growable pinned host array (contents are not preserved across growth: every call rewrites it)
// (large ones on huge pages, registered: BigPinned; small ones from hipHostMalloc)
template <typename T>
struct PinnedVec {
  T *p = nullptr; size_t n = 0, cap = 0; BigPinned big;
  int resize(size_t m) {
    if (m > cap) {
      release();
      size_t want = m + m / 4 + 64;
      if (want * sizeof(T) >= ((size_t)4 << 20)) { const int brc = big.alloc(want * sizeof(T)); if (brc) return brc; p = (T *)big.p; }
      else HIPCHK(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
      cap = want;
    }
    n = m;
    return BR_OK;
  }
  T *data() { return p; }
  void release() { if (big.p) big.release(); else if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
};

struct KEvent { int which; hipEvent_t a, b; };

struct br_ctx {
  const br_index *ix = nullptr;
  int group_lanes = 8;
  int bam_lanes = 0;   // 0: k_bam_tasks (a wave per 32 rows); 4..64: k_bam_encode<G>, G lanes per row
  int blocks_per_cu = 8;
  int split_spoil = 0;   // test hook: k_split_spoil plants wrong segment guesses (tests/test_gpu_split.py)
  int n_cu = 256;
  bool profiling = false;
  std::vector<KEvent> events; size_t events_used = 0;
  double k_ms[BR_K_NUM] = {0}; int32_t k_launches[BR_K_NUM] = {0};
  double k_ms_sum[BR_K_NUM] = {0}; int64_t k_launches_sum[BR_K_NUM] = {0};   // over the calls since profiling was switched on (br_ctx_kernel_ms_sum)
  uint64_t counters[8] = {0};
  uint64_t rescue_stats[4] = {0};  // problems, DP cells, accepted rescues, coded sequence bytes
  // device scratch
  DevBuf seg, meta, head, head2, fast_flag, fast_pre, n_matches, ranges, mask, match_off, cig_base, tile_sums, totals, counters_d;
  DevBuf m_tid, m_aux, m_p, m_x, m_b, m_cigoff, cig_arena, big_list, n_big, m_aln;
  DevBuf bam_aux, bam_base, bam_len, bam_off, bam_out, bam_end;
  struct StageSlot { DevBuf blob, off, len; hipEvent_t ready = nullptr; std::vector<uint64_t> h_off; int64_t n = 0; };
  StageSlot stage[3];              // br_bam_bundle_stage: uploads of the next bundles overlap the current projection
  hipStream_t copy_stream = nullptr;
  // flat (br_batch) staging: two input slots, uploads on copy_stream; packed rows go back on d2h_stream into the
  // slot's pinned arrays while the next batch is being projected on run_stream
  struct InSlot {
    DevBuf ref_id, ref_start, flags, xs, ts, cigar_off64, cigar, mate_ref, mate_start, name_off64, names, lqseq, seq_off64, seqs;
    DevBuf cigar_off, name_off, seq_off, mate_idx, group_off, seq_src, isnew, group_pre;
    hipEvent_t ready = nullptr, rows_home = nullptr;
    int64_t n = -1; uint64_t n_words = 0, n_name = 0, n_seq = 0; bool has_seq = false, staged = false, rows_pending = false;
    PinnedVec<uint4> h_a; PinnedVec<uint64_t> h_c, h_row_off; PinnedVec<uint32_t> h_pool; PinnedVec<uint4> h_x;
    PinnedVec<int32_t> h_mate, h_clip; PinnedVec<double> h_sim;
  };
  InSlot in_slot[2];
  hipStream_t run_stream = nullptr, d2h_stream = nullptr;
  hipEvent_t rows_busy = nullptr;   // recorded after the last packed download of the CURRENT row-table set was queued: the kernels that write the set wait for it
  bool rows_busy_set = false;
  // br_project_staged alternates between two sets of the tables a packed download reads (rows, detail, scores, dense CIGAR
  // references + pool, row_off): batch k's kernels write one set while batch k - 1's rows are still crossing PCIe out of the
  // other -- with one set the projection of batch k stood still behind the count pass until the wire was idle
  struct RowSetAlt { DevBuf pk_a, pk_x, pk_sim, pk_clip, pk_ch, pool, row_off; hipEvent_t busy = nullptr; bool busy_set = false; } alt;
  int host_detail = 0;              // br_host_rows carries the x (detail) array
  DevBuf z_slots, z_sizes, z_off, z_dense, z_dense_alt, z_tabs, z_tokens;
  DevBuf inf_out, inf_blocks, inf_tabs, inf_cnt; bool inf_tabs_ready = false;   // br_bgzf_inflate_device
  DevBuf sp_entry, sp_entry2, sp_exit, sp_nmap, sp_nunm, sp_ended, sp_redo, sp_pre, sp_small, sp_off, sp_len;   // br_bam_split_device
  int z_dense_which = 0;           // br_project_bam_staged_nowait: the packed blocks of call j are still on their way home while call j + 1 packs its own
  hipStream_t down_stream = nullptr; hipEvent_t ev_home[2] = {nullptr, nullptr}; std::atomic<bool> home_pending[2] = {{false}, {false}};
  int deflate_dynamic = 1;
  int emit_split = 1;
  int count_split = 1;   // count pass as two kernels: the main one without the exon walk, a second one for the alignments that need it
  int64_t speculate_n = 4194304;
  int speculate = 1;         // large batches are launched from the last call's counts, checked once at the end (run_device_small, big)
  int64_t hist_n = 0; uint64_t hist[4] = {0, 0, 0, 0}; bool hist_simf = false;   // the last call: alignments; matches, arena words, simple-class matches, records
  int small_batch = 1;       // batches of at most small_n alignments run without a host round trip before the final one (run_device_small)
  int64_t small_n = 65536;
  int single_pass = 0;   // short-read presets: count and emit in one sweep (k_project1 + k_emit_wl) instead of count / scan / expand / emit.
                         // Built for VERDICT r02 item 2, bit-exact, and SLOWER (11.3-12.3 ms against 9.2-9.8 ms per 10 M pairs,
                         // DESIGN section 10): off unless asked for ("single_pass" / BRAMBLE_AMD_SINGLE_PASS=1, the A/B switch)
  DevBuf wl, p1;         // the single pass's work list and counters
  uint64_t p1_hw[3] = {0, 0, 0};   // high-water marks of its three allocators (match slots, work-list entries, arena words): next call's capacities
  DevBuf walk_list, pmask, pbit, pick;
  // direct rows (run_device_direct): presets without the similarity filter and without -S pair on the count pass's survivor
  // sets before anything is emitted, and the emit kernels write the packed rows themselves
  int direct_rows = 1;       // "direct_rows" / BRAMBLE_AMD_DIRECT_ROWS=0: the match-table path (k_emit_dense -> k_pair -> k_rows), the A/B switch
  DevBuf d_fm, d_nkept, d_desc, d_hi0, d_clspos, d_rnd, d_side, d_sidectr;
  uint64_t d_side_cap = 0;
  bool last_direct = false;  // the last call's rows came from the direct path: the detail column is re-emitted on request, not gathered
  bool want_x = false;       // the caller of run_device needs the detail column (input alignment, HI: the BAM encoder) with the rows
  ProjectArgs dA{}; DirectArgs dD{}; int64_t d_kept = 0, d_simple = 0; bool d_split = false;
  // packed row table (the product of the row stage) and what its kernels need
  DevBuf r_rec, pk_a, pk_c, pk_x, pk_sim, pk_clip;
  DevBuf pool, pool_sizes, pool_off, pk_ch;   // dense long-CIGAR pool + rewritten references for host downloads
  bool last_aux_cols = false;           // the last call's rows carry similarity / clip scores
  bool wide_valid = false;              // the wide view below matches the last call's rows
  bool detail_valid = false;            // pk_x (br_row_x) has been derived for the last call's rows
  const int32_t *last_l_qseq = nullptr; // the last batch's l_qseq (device; insert sizes of the wide view / the encoder)
  int32_t last_long_reads = 0;
  int64_t last_n_pool = 0;
  bool z_tabs_ready = false;
  DevBuf p_ncig, p_name_len, p_isnew, p_group_pre, p_small, p_big, p_seq_len, p_ref_map;
  uint8_t *h_bam[2] = {nullptr, nullptr}; size_t h_bam_cap[2] = {0, 0}; int h_bam_next = 0;  // pinned download buffers of br_project_bam_bundle (alternating)
  BigPinned h_bam_mem[2];
  int64_t last_n_rows = 0, last_n_aln = 0;
  DevBuf fa_stats, fa_n_prob, fa_seq_bytes, fa_prob_off, fa_seqarena_off, fa_probs, fa_results, fa_seq_arena, fa_clip_ops,
      fa_ideal_cap, fa_scratch, fa_srcs, fa_want, b_seq_off, b_seqs, b_seq_src;
  // the streamed -S DP (ksw_kernels.hip): per-bin descriptors, per-problem DP results, leftovers, counters, group
  // rows / offsets, the direction tape, raw traceback ops
  DevBuf ksw_desc, ksw_dp, ksw_left, ksw_cnt, ksw_group, ksw_tape, ksw_raw;
  int ksw_fast = 1;            // 0: every problem through the general kernel k_ksw
  hipStream_t ksw_stream = nullptr; hipEvent_t ksw_ev[KSW_N_BINS + 1] = {}; hipEvent_t aux_ev[8] = {};   // the second stream
  hipStream_t aux2_stream = nullptr; hipEvent_t aux2_ev = nullptr;   // a third one: the name seeds of the direct path beside k_pair_mask
  uint32_t ksw_groups[KSW_N_BINS] = {0};
  int64_t ksw_tape_mb = 49152; // HBM set aside for the direction tape; larger batches go through in pieces
  int ksw_tape_pct = 100;      // test hook: the share of the computed tape the DP kernels may use (the rest of the problems goes to k_ksw)
  uint64_t ksw_diag[16] = {0};  // last call: pieces, problems per bin [4], leftovers before the DP, tape bytes (largest piece), spare, tape rows per bin [4]
  DevBuf n_rows, row_off, aln_group;
  // wide view of the rows (br_device_rows_expand): one array per field
  DevBuf r_input, r_nh, r_hi, r_mapq, r_group, r_mate_tid, r_mate_pos,
      r_isize, r_tid, r_pos, r_ncig, r_strand, r_sim, r_clip, r_junc, r_refc, r_cigoff, cigar_out;
  DevBuf r_paired, r_same, r_first, r_primary;
  DevBuf b_name_off, b_names;
  // device staging of host batches (br_project_batch)
  DevBuf b_ref_id, b_ref_start, b_flags, b_xs, b_ts, b_cigar_off, b_cigar, b_mate_idx, b_group_off, b_lqseq;
  uint64_t *h_totals = nullptr;  // pinned, 512 words ([192..] the direct path's counters with k_group_desc's slots)
  // host result storage (br_project_batch / br_project_group)
  // pinned: the row download runs at PCIe speed instead of through the pageable bounce path
  PinnedVec<int32_t> h_input, h_clip, h_junc, h_refc, h_mate_tid, h_mate_pos, h_isize;
  PinnedVec<uint32_t> h_tid, h_pos, h_nh, h_hi, h_mapq, h_group, h_cigar;
  PinnedVec<int8_t> h_strand;
  PinnedVec<uint64_t> h_cigoff;
  PinnedVec<double> h_sim;
  PinnedVec<uint8_t> h_primary, h_paired, h_same, h_first;
  std::vector<br_projected> h_proj;
  // br_project_group(s): one packed upload of the call's alignments, and the packed rows / their CIGAR words back
  PinnedVec<uint8_t> g_host; DevBuf g_dev;
  PinnedVec<uint4> g_a, g_x; PinnedVec<uint2> g_c; PinnedVec<uint32_t> g_pool, g_cig; PinnedVec<double> g_sim;
  bool rows_to_host = false, rows_at_host = false;   // br_project_group(s): the small path's row kernel writes g_a / g_c / g_x / g_sim (pinned host memory) itself
  DevBuf *all() { return &seg; }
};

extern "C" int br_ctx_new(const br_index *ix, br_ctx **out) {
  if (!ix || !out) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  if (ix->device < 0) return BR_ERR_NO_DEVICE;
  int rc = check_device(ix->device);
  if (rc) return rc;
  HIPCHK(hipSetDevice(ix->device));
  br_ctx *c = new br_ctx();
  c->ix = ix;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ix->device));
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCHK(hipHostMalloc((void **)&c->h_totals, 512 * sizeof(uint64_t), hipHostMallocDefault));   // [0..31] scan totals and counters, [96..] the single pass's counters
  const char *bl = getenv("BRAMBLE_AMD_BAM_LANES");
  if (bl) { int v = atoi(bl); if (v == 0 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) c->bam_lanes = v; }
  const char *spec = getenv("BRAMBLE_AMD_SPECULATE");      // A/B: 0 = large batches always through the ordinary pipeline (three host round trips)
  if (spec) c->speculate = atoi(spec) != 0;
  const char *sp = getenv("BRAMBLE_AMD_SINGLE_PASS");   // A/B: 0 = the two-pass count / scan / expand / emit path
  if (sp) c->single_pass = atoi(sp) != 0;
  const char *dr = getenv("BRAMBLE_AMD_DIRECT_ROWS");   // A/B: 0 = the match-table path
  if (dr) c->direct_rows = atoi(dr) != 0;
  const char *g = getenv("BRAMBLE_AMD_GROUP_LANES");
  if (g) { int v = atoi(g); if (v == 8 || v == 16 || v == 32 || v == 64) c->group_lanes = v; }
  *out = c;
  return BR_OK;
}

extern "C" void br_ctx_free(br_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->ix->device);
  DevBuf *bufs[] = {&c->seg, &c->meta, &c->head, &c->head2, &c->fast_flag, &c->fast_pre, &c->n_matches, &c->ranges, &c->mask, &c->match_off, &c->cig_base,
                    &c->tile_sums, &c->totals, &c->counters_d, &c->m_tid, &c->m_aux, &c->m_p, &c->m_x, &c->m_b,
                    &c->m_cigoff, &c->cig_arena, &c->big_list, &c->n_big, &c->m_aln,
                    &c->bam_aux, &c->bam_base, &c->bam_len, &c->bam_off, &c->bam_out, &c->bam_end, &c->z_slots, &c->z_sizes, &c->z_off, &c->z_dense, &c->z_dense_alt, &c->inf_out, &c->inf_blocks, &c->inf_tabs, &c->inf_cnt, &c->sp_entry, &c->sp_entry2, &c->sp_exit, &c->sp_nmap, &c->sp_nunm, &c->sp_ended, &c->sp_redo, &c->sp_pre, &c->sp_small, &c->sp_off, &c->sp_len, &c->z_tabs, &c->z_tokens, &c->p_ncig, &c->p_name_len, &c->p_isnew, &c->p_group_pre, &c->p_small, &c->p_big, &c->p_seq_len, &c->p_ref_map, &c->fa_stats, &c->fa_n_prob, &c->fa_seq_bytes, &c->fa_prob_off, &c->fa_seqarena_off, &c->fa_probs, &c->fa_results,
                    &c->fa_seq_arena, &c->fa_clip_ops, &c->fa_ideal_cap, &c->fa_scratch, &c->b_seq_off, &c->b_seqs, &c->b_seq_src,
                    &c->r_rec, &c->pk_a, &c->pk_c, &c->pk_x, &c->pk_sim, &c->pk_clip, &c->pool, &c->pool_sizes, &c->pool_off, &c->pk_ch,
                    &c->n_rows, &c->row_off, &c->aln_group, &c->r_input, &c->r_nh, &c->r_hi, &c->r_mapq,
                    &c->r_group, &c->r_mate_tid, &c->r_mate_pos, &c->r_isize, &c->r_tid, &c->r_pos,
                    &c->r_ncig, &c->r_strand, &c->r_sim, &c->r_clip, &c->r_junc, &c->r_refc, &c->r_cigoff,
                    &c->cigar_out, &c->r_paired, &c->r_same, &c->r_first, &c->r_primary, &c->b_name_off, &c->b_names, &c->b_ref_id, &c->b_ref_start,
                    &c->b_flags, &c->b_xs, &c->b_ts, &c->b_cigar_off, &c->b_cigar, &c->b_mate_idx,
                    &c->b_group_off, &c->b_lqseq, &c->walk_list, &c->pmask, &c->pbit, &c->pick, &c->wl, &c->p1, &c->g_dev,
                    &c->d_fm, &c->d_nkept, &c->d_desc, &c->d_hi0, &c->d_clspos, &c->d_rnd, &c->d_side, &c->d_sidectr,
                    &c->fa_srcs, &c->fa_want, &c->ksw_desc, &c->ksw_dp, &c->ksw_left, &c->ksw_cnt, &c->ksw_group, &c->ksw_tape, &c->ksw_raw,
                    &c->alt.pk_a, &c->alt.pk_x, &c->alt.pk_sim, &c->alt.pk_clip, &c->alt.pk_ch, &c->alt.pool, &c->alt.row_off};
  for (DevBuf *b : bufs) b->release();
  for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->h_totals) (void)hipHostFree(c->h_totals);
  for (int k = 0; k < 2; k++) c->h_bam_mem[k].release();
  for (auto &S : c->stage) { S.blob.release(); S.off.release(); S.len.release(); if (S.ready) (void)hipEventDestroy(S.ready); }
  for (auto &S : c->in_slot) {
    DevBuf *ib[] = {&S.ref_id, &S.ref_start, &S.flags, &S.xs, &S.ts, &S.cigar_off64, &S.cigar, &S.mate_ref, &S.mate_start, &S.name_off64,
                    &S.names, &S.lqseq, &S.seq_off64, &S.seqs, &S.cigar_off, &S.name_off, &S.seq_off, &S.mate_idx, &S.group_off,
                    &S.seq_src, &S.isnew, &S.group_pre};
    for (DevBuf *b : ib) b->release();
    if (S.ready) (void)hipEventDestroy(S.ready);
    if (S.rows_home) (void)hipEventDestroy(S.rows_home);
    S.h_a.release(); S.h_c.release(); S.h_row_off.release(); S.h_pool.release(); S.h_x.release(); S.h_mate.release();
    S.h_clip.release(); S.h_sim.release();
  }
  if (c->rows_busy) (void)hipEventDestroy(c->rows_busy);
  if (c->alt.busy) (void)hipEventDestroy(c->alt.busy);
  if (c->aux2_stream) { (void)hipStreamDestroy(c->aux2_stream); (void)hipEventDestroy(c->aux2_ev); }
  if (c->ksw_stream) { (void)hipStreamDestroy(c->ksw_stream); for (auto &e : c->ksw_ev) if (e) (void)hipEventDestroy(e); for (auto &e : c->aux_ev) if (e) (void)hipEventDestroy(e); }
  if (c->run_stream) (void)hipStreamDestroy(c->run_stream);
  if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
  for (int k = 0; k < 2; k++) if (c->ev_home[k]) (void)hipEventDestroy(c->ev_home[k]);
  c->h_input.release(); c->h_clip.release(); c->h_junc.release(); c->h_refc.release(); c->h_mate_tid.release(); c->h_mate_pos.release();
  c->h_isize.release(); c->h_tid.release(); c->h_pos.release(); c->h_nh.release(); c->h_hi.release(); c->h_mapq.release(); c->h_group.release();
  c->h_cigar.release(); c->h_strand.release(); c->h_cigoff.release(); c->h_sim.release(); c->h_primary.release(); c->h_paired.release();
  c->h_same.release(); c->h_first.release();
  c->g_host.release(); c->g_a.release(); c->g_x.release(); c->g_c.release(); c->g_pool.release(); c->g_cig.release(); c->g_sim.release();
  delete c;
}

extern "C" int br_ctx_set_profiling(br_ctx *c, int enabled) {
  if (!c) return BR_ERR_INVALID_ARG;
  if (enabled && !c->profiling) for (int k = 0; k < BR_K_NUM; k++) { c->k_ms_sum[k] = 0; c->k_launches_sum[k] = 0; }
  c->profiling = enabled != 0;
  return BR_OK;
}
extern "C" int br_ctx_kernel_ms_sum(br_ctx *c, int which, double *ms, int64_t *launches) {
  if (!c || which < 0 || which >= BR_K_NUM) return BR_ERR_INVALID_ARG;
  if (ms) *ms = c->k_ms_sum[which];
  if (launches) *launches = c->k_launches_sum[which];
  return BR_OK;
}
extern "C" int br_ctx_set_param(br_ctx *c, const char *key, int64_t v) {
  if (!c || !key) return BR_ERR_INVALID_ARG;
  if (!strcmp(key, "group_lanes")) { if (v != 8 && v != 16 && v != 32 && v != 64) return BR_ERR_INVALID_ARG; c->group_lanes = (int)v; return BR_OK; }
  if (!strcmp(key, "emit_split")) { c->emit_split = v != 0; return BR_OK; }
  if (!strcmp(key, "count_split")) { c->count_split = v != 0; return BR_OK; }
  if (!strcmp(key, "single_pass")) { c->single_pass = v != 0; return BR_OK; }
  if (!strcmp(key, "direct_rows")) { c->direct_rows = v != 0; return BR_OK; }
  if (!strcmp(key, "small_batch")) { c->small_batch = v != 0; return BR_OK; }
  if (!strcmp(key, "speculate")) { c->speculate = v != 0; return BR_OK; }
  if (!strcmp(key, "speculate_n")) { if (v < 0) return BR_ERR_INVALID_ARG; c->speculate_n = v; return BR_OK; }
  if (!strcmp(key, "small_n")) { if (v < 0) return BR_ERR_INVALID_ARG; c->small_n = v; return BR_OK; }
  if (!strcmp(key, "deflate_dynamic")) { c->deflate_dynamic = v != 0; return BR_OK; }
  if (!strcmp(key, "bam_lanes")) { if (v != 0 && v != 4 && v != 8 && v != 16 && v != 32 && v != 64) return BR_ERR_INVALID_ARG; c->bam_lanes = (int)v; return BR_OK; }
  if (!strcmp(key, "ksw_fast")) { c->ksw_fast = v != 0; return BR_OK; }
  if (!strcmp(key, "ksw_tape_pct")) { if (v < 0 || v > 100) return BR_ERR_INVALID_ARG; c->ksw_tape_pct = (int)v; return BR_OK; }
  if (!strcmp(key, "ksw_tape_mb")) { if (v < 1) return BR_ERR_INVALID_ARG; c->ksw_tape_mb = v; return BR_OK; }
  if (!strcmp(key, "host_detail")) { c->host_detail = v != 0; return BR_OK; }
  if (!strcmp(key, "split_spoil")) { if (v < 0 || v > 1000000) return BR_ERR_INVALID_ARG; c->split_spoil = (int)v; return BR_OK; }   // test hook, see split_impl
  if (!strcmp(key, "blocks_per_cu")) { if (v < 1 || v > 64) return BR_ERR_INVALID_ARG; c->blocks_per_cu = (int)v; return BR_OK; }
  return BR_ERR_INVALID_ARG;
}
extern "C" int br_ctx_kernel_ms(br_ctx *c, int which, double *ms, int32_t *launches) {
  if (!c || which < 0 || which >= BR_K_NUM) return BR_ERR_INVALID_ARG;
  if (ms) *ms = c->k_ms[which];
  if (launches) *launches = c->k_launches[which];
  return BR_OK;
}
// Diagnostic: how the last -S call's DP problems were routed (pieces, problems per array shape, leftovers handed to the
// general kernel before the DP, tape bytes of the largest piece, leftovers of the last piece after the DP).
extern "C" int br_ctx_ksw_diag(br_ctx *c, uint64_t out[16]) {
  if (!c || !out) return BR_ERR_INVALID_ARG;
  memcpy(out, c->ksw_diag, sizeof(c->ksw_diag));
  out[7] = (uint32_t)c->h_totals[24];
  return BR_OK;
}

extern "C" int br_ctx_rescue_stats(br_ctx *c, uint64_t out[4]) {
  if (!c || !out) return BR_ERR_INVALID_ARG;
  memcpy(out, c->rescue_stats, sizeof(c->rescue_stats));
  return BR_OK;
}
extern "C" int br_ctx_last_counters(br_ctx *c, uint64_t out[8]) {
  if (!c || !out) return BR_ERR_INVALID_ARG;
  memcpy(out, c->counters, sizeof(c->counters));
  return BR_OK;
}

namespace {

struct Prof {
  br_ctx *c; hipStream_t st;
  hipStream_t cur = nullptr;   // stream of the open begin / end pair
  int begin(int which, hipStream_t on = nullptr) {
    cur = on ? on : st;
    if (!c->profiling) return BR_OK;
    if (c->events_used == c->events.size()) {
      KEvent e; e.which = which;
      HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b));
      c->events.push_back(e);
    }
    c->events[c->events_used].which = which;
    HIPCHK(hipEventRecord(c->events[c->events_used].a, cur));
    return BR_OK;
  }
  int end() {
    if (!c->profiling) return BR_OK;
    HIPCHK(hipEventRecord(c->events[c->events_used].b, cur));
    c->events_used++;
    return BR_OK;
  }
  int collect() {
    for (int k = 0; k < BR_K_NUM; k++) { c->k_ms[k] = 0; c->k_launches[k] = 0; }
    if (!c->profiling) return BR_OK;
    for (size_t i = 0; i < c->events_used; i++) {
      float ms = 0;
      // (the call has waited for its streams already: as a rule the events are complete and one query each is all it takes)
      if (hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipEventSynchronize(c->events[i].b));
        HIPCHK(hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b));
      }
      c->k_ms[c->events[i].which] += ms; c->k_launches[c->events[i].which]++;
      c->k_ms_sum[c->events[i].which] += ms; c->k_launches_sum[c->events[i].which]++;
    }
    c->events_used = 0;
    return BR_OK;
  }
};

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// the next run_device call should leave the detail column (input alignment, junc_hits, aligned_len, HI) next to the rows:
// the direct path then writes it in the emit pass instead of emitting a second time on request
struct WantDetail { br_ctx *c; bool old; WantDetail(br_ctx *c_, bool v) : c(c_), old(c_->want_x) { c->want_x = v; } ~WantDetail() { c->want_x = old; } };

// a second stream for kernels that can run beside the main one (a shape's tracebacks beside the next shape's DP; the
// few-block emit kernel of the > 64-candidate alignments beside the work-list kernels)
// priority of the context's side streams (A/B: BRAMBLE_AMD_AUX_PRIO=low|high; default: normal)
static int aux_stream_priority() {
  static const int prio = []() {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return 0;
    const char *e = getenv("BRAMBLE_AMD_AUX_PRIO");
    if (e && !strcmp(e, "low")) return lo;
    if (e && !strcmp(e, "high")) return hi;
    return 0;
  }();
  return prio;
}
static int ensure_aux_stream(br_ctx *c) {
  if (c->ksw_stream) return BR_OK;
  HIPCHK(hipStreamCreateWithPriority(&c->ksw_stream, hipStreamNonBlocking, aux_stream_priority()));
  for (auto &e : c->ksw_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : c->aux_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return BR_OK;
}

// The -S rescue DP over n_prob problems.  Problems whose target fits a register array go
// through the streamed kernels piece by piece (a piece = a range of problems whose direction tape fits the budget):
// k_ksw_bin -> [host reads the bin sizes] -> k_ksw_plan -> k_ksw_dp per bin -> k_ksw over the leftovers -> k_ksw_trace.
struct KswRun {
  int64_t n_prob; const KswProb *probs; KswRes *results; const uint8_t *seq_arena; uint32_t *clip_ops;
  uint64_t seq_total, qmax, tmax; uint64_t *stats;
  uint32_t *raw_out, *raw_n; int32_t *max_out; uint32_t raw_cap;
};

static int run_ksw(br_ctx *c, hipStream_t st, const KswRun &R) {
  if (R.n_prob <= 0) return BR_OK;
  const uint64_t n_all = (uint64_t)R.n_prob;
  const uint64_t qmax = std::max<uint64_t>(R.qmax, 1), tmax = std::max<uint64_t>(R.tmax, 1);
  KswArgs K{};
  K.n_prob = R.n_prob; K.probs = R.probs; K.results = R.results; K.seq_arena = R.seq_arena; K.clip_ops = R.clip_ops;
  K.stats = R.stats;
  K.raw_out = R.raw_out; K.raw_n = R.raw_n; K.max_out = R.max_out; K.raw_cap = R.raw_cap;
  // the general kernel: per-wave scratch = direction matrix + raw traceback ops + (large targets) u / v / x / y, sized by the
  // longest query / target it will see.  16 GB are set aside; one outlier (a 100 kb soft clip) may take more: then a single
  // wave runs, as long as its matrix fits in 60 % of the free HBM
  auto general = [&](uint64_t n_work, uint64_t q_hi, uint64_t t_hi, const uint32_t *list, const uint32_t *n_list) -> int {
    q_hi = std::max<uint64_t>(q_hi, 1); t_hi = std::max<uint64_t>(t_hi, 1);
    K.tmax = (uint32_t)t_hi;
    K.pmat_bytes = (size_t)(((q_hi + t_hi) * t_hi + 15) & ~15ull);
    K.raw_words = (size_t)((q_hi + t_hi + 4 + 3) & ~3ull);
    K.scratch_per_wave = K.pmat_bytes + K.raw_words * 4 + ((4 * t_hi + 15) & ~15ull);
    const uint64_t budget = 16ull << 30;
    uint64_t waves = std::min<uint64_t>({(uint64_t)c->n_cu * 16, budget / K.scratch_per_wave, n_work});
    if (waves == 0) {
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      if ((double)K.scratch_per_wave > 0.6 * (double)(free_b + c->fa_scratch.cap)) return BR_ERR_CAPACITY;
      waves = 1;
    }
    RC(c->fa_scratch.ensure((size_t)waves * K.scratch_per_wave));
    K.scratch = c->fa_scratch.as<uint8_t>(); K.list = list; K.n_list = n_list; K.n_waves = (int64_t)waves;
    launch_ksw(st, K, (int)((waves + 3) / 4));
    return BR_OK;
  };
  memset(c->ksw_diag, 0, sizeof(c->ksw_diag));
  c->h_totals[24] = 0;
  if (!c->ksw_fast) return general(n_all, qmax, tmax, nullptr, nullptr);

  RC(c->ksw_raw.ensure((size_t)(R.seq_total + n_all + 1) * 4));
  RC(c->ksw_cnt.ensure(128));
  uint32_t *h_cnt = (uint32_t *)(c->h_totals + 16);   // 16 words of the pinned totals
  const uint64_t tape_budget = (uint64_t)c->ksw_tape_mb << 20;
  // groups of a bin = what is resident at once (one wave of blocks: every group runs from the first cycle)
  uint32_t max_groups[KSW_N_BINS];
  for (int b = 0; b < KSW_N_BINS; b++) {
    if (!c->ksw_groups[b]) c->ksw_groups[b] = ksw_dp_resident_groups(b, c->n_cu);
    max_groups[b] = c->ksw_groups[b];
  }
  std::vector<std::pair<uint64_t, uint64_t>> todo;   // [p0, p1)
  todo.emplace_back(0, n_all);
  while (!todo.empty()) {
    const uint64_t p0 = todo.back().first, p1 = todo.back().second, n = p1 - p0;
    todo.pop_back();
    RC(c->ksw_desc.ensure((size_t)n * sizeof(KswDesc) * KSW_N_BINS));
    RC(c->ksw_dp.ensure((size_t)n * sizeof(KswDp)));
    RC(c->ksw_left.ensure((size_t)n * 4));
    KswFastArgs A{};
    A.p0 = (int64_t)p0; A.n = (int64_t)n; A.probs = R.probs; A.results = R.results; A.seq_arena = R.seq_arena;
    A.clip_ops = R.clip_ops; A.raw_ops = c->ksw_raw.as<uint32_t>();
    for (int b = 0; b < KSW_N_BINS; b++) A.desc[b] = c->ksw_desc.as<KswDesc>() + (size_t)b * n;
    A.counters = c->ksw_cnt.as<uint32_t>(); A.leftover = c->ksw_left.as<uint32_t>(); A.dp = c->ksw_dp.as<KswDp>();
    A.stats = R.stats; A.raw_out = R.raw_out; A.raw_n = R.raw_n; A.max_out = R.max_out; A.raw_cap = R.raw_cap;
    HIPCHK(hipMemsetAsync(c->ksw_cnt.p, 0, 128, st));
    launch_ksw_bin(st, A);
    HIPCHK(hipMemcpyAsync(h_cnt, c->ksw_cnt.p, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t *h_rows = (const uint64_t *)(h_cnt + 8);
    // tape: a row = one step of a wave (512 or 1024 B); a bin's waves run (tape rows of its problems) / (groups per wave) steps when
    // its groups stay equally busy (they share one queue), 15 % on top, and every wave rounds up to chunks and drains
    uint64_t tape_bytes = 0, tape_fixed = 0; uint32_t n_groups_total = 0;   // tape_fixed: what every wave rounds up and drains, whatever the piece holds
    for (int b = 0; b < KSW_N_BINS; b++) {
      A.n_bin[b] = h_cnt[b];
      A.n_groups[b] = std::min<uint32_t>(max_groups[b], (h_cnt[b] + 7u) / 8u);   // a group takes its problems eight at a time
      n_groups_total += A.n_groups[b];
      const uint64_t gpw = 64u / (uint64_t)KSW_BIN_G(b), waves = (A.n_groups[b] + gpw - 1) / gpw;
      const uint64_t fixed = waves * (1ull * KSW_CHUNK_ROWS + KSW_BIN_W(b) + 64);
      const uint64_t rows = h_rows[b] / gpw + h_rows[b] / gpw / 7 + fixed;
      if (A.n_bin[b]) { tape_bytes += rows * (uint64_t)KSW_BIN_ROWBYTES(b); tape_fixed += fixed * (uint64_t)KSW_BIN_ROWBYTES(b); }
    }
    if (tape_bytes > tape_budget && n >= 2048) {
      // as few pieces as fit: the part of the tape that scales with the problems over what a piece has left for it (halves
      // when the fixed part alone nearly fills the budget); a piece that still does not fit is cut again
      uint64_t k = 2;
      if (tape_budget > tape_fixed + tape_fixed / 4) k = ((tape_bytes - tape_fixed) + (tape_budget - tape_fixed) - 1) / (tape_budget - tape_fixed);
      k = std::min<uint64_t>(std::max<uint64_t>(k, 2), std::min<uint64_t>(64, n / 1024));
      for (uint64_t i = k; i-- > 0;) todo.emplace_back(p0 + n * i / k, p0 + n * (i + 1) / k);
      continue;
    }
    const uint32_t n_left = h_cnt[KSW_N_BINS];
    c->ksw_diag[0]++;
    for (int b = 0; b < KSW_N_BINS; b++) c->ksw_diag[1 + b] += h_cnt[b];
    c->ksw_diag[5] += n_left; c->ksw_diag[6] = std::max<uint64_t>(c->ksw_diag[6], tape_bytes);
    for (int b = 0; b < KSW_N_BINS; b++) c->ksw_diag[8 + b] += h_rows[b];
    if (n_groups_total) {
      RC(c->ksw_tape.ensure((size_t)tape_bytes + 256));
      A.tape = c->ksw_tape.as<uint8_t>(); A.tape_cap = tape_bytes / 100 * (uint64_t)c->ksw_tape_pct;
      // a shape's tracebacks (one lane per problem, waiting on tape lines) run on a second stream beside the next shape's
      // DP kernel (issue-bound, one wave of registers to spare per SIMD)
      RC(ensure_aux_stream(c));
      for (int b = KSW_N_BINS - 1; b >= 0; b--) {     // widest shape first: the exposed last traceback is the smallest shape's
        if (!A.n_bin[b]) continue;
        launch_ksw_dp(st, A, b);
        HIPCHK(hipEventRecord(c->ksw_ev[b], st));
        HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->ksw_ev[b], 0));
        launch_ksw_trace(c->ksw_stream, A, b);
      }
      HIPCHK(hipEventRecord(c->ksw_ev[KSW_N_BINS], c->ksw_stream));
    }
    // what the arrays do not take: targets beyond the widest array, and (never seen outside tests) groups whose tape ran out
    // (problems a wave hands back when the tape runs out come from the arrays: at most KSW_MAX_SPAN bases)
    RC(general(n_left ? n_left : 64, std::max<uint64_t>(h_cnt[5], KSW_MAX_SPAN), std::max<uint64_t>(h_cnt[6], KSW_BIN_W(KSW_N_BINS - 1)), A.leftover, A.counters + KSW_N_BINS));
    if (n_groups_total) HIPCHK(hipStreamWaitEvent(st, c->ksw_ev[KSW_N_BINS], 0));   // the tape and the result arrays are free again
    // leftovers after the DP (those of the last piece; read by br_ctx_ksw_diag once the stream has been synchronised)
    HIPCHK(hipMemcpyAsync((uint32_t *)(c->h_totals + 24), A.counters + KSW_N_BINS, 4, hipMemcpyDeviceToHost, st));
  }
  return BR_OK;
}

// Small batches (a read-name group, the 64 groups a bramble-cli worker holds, the 100 k alignments of a reference bundle):
// the ordinary pipeline stops three times for the host to read a total and size the next tables, and launches about
// twenty kernels -- for 10 k alignments that is 0.3 ms of which the kernels are a fraction.  Here the tables are sized from
// upper bounds (32 matches per alignment and the CIGAR room that goes with them), the scan totals stay on the device
// (ProjectArgs::tot: the kernels that need a count read it there, and do nothing when a total is beyond its table), the
// split kernels run in their single-launch forms, everything goes down ONE stream, and the host waits once, at the end.
// A batch that does not fit the bounds (a dense locus) comes back as BR_RETRY_ORDINARY and takes the ordinary path.
#define BR_RETRY_ORDINARY 1000
// LARGE batches take the same route when the context has projected a batch before (`big`): the tables are what earlier calls
// left behind (grown with a quarter of headroom), the launch grids of the two emit classes and of the row kernel come from
// the LAST call's counts scaled to this batch's size (+15 %), the kernels keep their split, two-stream forms -- and the
// host, instead of stopping three times, checks once at the end that nothing outgrew its table or its grid.  A batch that
// did is redone the ordinary way, which also grows the tables.
static int run_device_small(br_ctx *c, const DevCfg &dc, const br_device_batch *b, hipStream_t st, br_device_rows *out, Prof &pf,
                            bool keep_events, bool big) {
  const br_index *ix = c->ix;
  const int64_t n = b->n_aln, ng = b->n_groups;
  const bool aux_cols = dc.filter_by_similarity != 0;
  uint64_t cap_m = 32ull * (uint64_t)n + 8192;
  const uint64_t per = 9ull * (uint64_t)std::min<int32_t>(std::max<int32_t>(b->max_n_cigar, 4), 64) + 12ull;   // n_real + 2 (4 n_seg + 2) <= 9 n_cigar + 12
  uint64_t cap_c = std::min<uint64_t>(cap_m * per, 1ull << 28);
  uint64_t cap_r = cap_m, cover_m = cap_m, cover_s = cap_m, cover_g = cap_m, cover_r = cap_m;   // tables' capacities; what the emit / row grids cover (all matches, simple class, general class, records)
  if (big) {
    const double f = 1.15 * (double)n / (double)std::max<int64_t>(c->hist_n, 1);
    cover_m = (uint64_t)(f * (double)c->hist[0]) + 4096; cover_s = (uint64_t)(f * (double)c->hist[2]) + 4096; cover_r = (uint64_t)(f * (double)c->hist[3]) + 4096;
    cover_g = (uint64_t)(f * (double)(c->hist[0] - std::min(c->hist[0], c->hist[2]))) + 4096;
    const uint64_t need_c = (uint64_t)(f * (double)c->hist[1]) + 4096;
    cap_m = std::min<uint64_t>({c->m_tid.cap / 4, c->m_aux.cap / 4, c->m_p.cap / 8, c->m_x.cap / 8, c->m_b.cap / 16, c->m_cigoff.cap / 8, c->m_aln.cap / 4});
    cap_c = c->cig_arena.cap / 4;
    cap_r = std::min<uint64_t>({c->r_rec.cap / 16, c->pk_a.cap / 16, c->pk_c.cap / 8});
    if (aux_cols) cap_r = std::min<uint64_t>({cap_r, c->pk_sim.cap / 8, c->pk_clip.cap / 4});
    if (cover_m > cap_m || need_c > cap_c || cover_r > cap_r) return BR_RETRY_ORDINARY;   // the tables have to grow: the ordinary path does that
  }
  const int64_t tiles = std::max<int64_t>(scan_tiles_for(std::max<int64_t>(n, ng) + 1), 1);
  RC(c->seg.ensure((size_t)(b->n_cigar_words + n) * sizeof(uint2)));
  RC(c->meta.ensure((size_t)n * sizeof(AlnMeta))); RC(c->head.ensure((size_t)n * sizeof(uint4))); RC(c->head2.ensure((size_t)n * sizeof(uint4)));
  RC(c->fast_flag.ensure((size_t)n * 4)); RC(c->fast_pre.ensure((size_t)(n + 1) * 4));
  RC(c->n_matches.ensure((size_t)n * 4)); RC(c->ranges.ensure((size_t)n * sizeof(uint4)));
  RC(c->mask.ensure((size_t)n * 8)); RC(c->match_off.ensure((size_t)(n + 1) * 4));
  RC(c->cig_base.ensure((size_t)(n + 1) * 8)); RC(c->tile_sums.ensure((size_t)tiles * 8 * 3));
  RC(c->totals.ensure(16 * 8)); RC(c->counters_d.ensure(GD_COUNTER_WORDS * 8));
  RC(c->big_list.ensure((size_t)n * 4)); RC(c->n_big.ensure(16));
  if (!big) {
    RC(c->m_tid.ensure(cap_m * 4)); RC(c->m_aux.ensure(cap_m * 4)); RC(c->m_p.ensure(cap_m * sizeof(uint2))); RC(c->m_x.ensure(cap_m * sizeof(uint2)));
    RC(c->m_b.ensure(cap_m * sizeof(uint4))); RC(c->m_cigoff.ensure(cap_m * 8)); RC(c->m_aln.ensure(cap_m * 4));
    RC(c->cig_arena.ensure(cap_c * 4));
  }
  // a queued packed download of the last call (br_project_staged) may still read the row tables: small batches wait for it here;
  // large ones keep the overlap -- their tables are not reallocated (checked above) -- and make the row scan wait instead
  const bool rows_busy_wait = c->rows_busy_set && big && c->row_off.cap >= (size_t)(n + 1) * 8;
  if (c->rows_busy_set && !rows_busy_wait) HIPCHK(hipEventSynchronize(c->rows_busy));
  RC(c->n_rows.ensure((size_t)n * 4)); RC(c->row_off.ensure((size_t)(n + 1) * 8)); RC(c->aln_group.ensure((size_t)n * 4));
  RC(c->pmask.ensure((size_t)n * 8)); RC(c->pbit.ensure((size_t)n));
  if (!big) {
    RC(c->r_rec.ensure(cap_m * sizeof(uint4))); RC(c->pk_a.ensure(cap_m * sizeof(uint4))); RC(c->pk_c.ensure(cap_m * sizeof(uint2)));
    if (aux_cols) { RC(c->pk_sim.ensure(cap_m * 8)); RC(c->pk_clip.ensure(cap_m * 4)); }
  }
  if (!aux_cols) RC(c->pick.ensure((size_t)std::max<int64_t>(ng, 1) * 8));
  if (big) { RC(c->walk_list.ensure((size_t)n * 4)); RC(ensure_aux_stream(c)); }
  uint64_t *d_tot = c->totals.as<uint64_t>();

  // (the per-batch counters sit behind the totals, d_tot[8..11]: one download brings both home; k_segment zeroes them and
  // the two work-list counters, and labels the alignments with their read-name groups)
  uint64_t *d_cnt = d_tot + 8;
  SegExtra X{};
  X.group_off = b->group_off; X.aln_group = c->aln_group.as<uint32_t>(); X.n_groups = ng;
  X.zero_a = (uint64_t *)c->n_big.p; X.n_zero_a = 1; X.zero_b = d_cnt; X.n_zero_b = 4;
  RC(pf.begin(BR_K_SEGMENT));
  launch_segment(st, n, b->ref_id, b->ref_start, b->flags, b->xs, b->ts, b->cigar_off, b->cigar, dc, ix->n_refs,
                 c->seg.as<uint2>(), c->meta.as<AlnMeta>(), c->head.as<uint4>(), c->head2.as<uint4>(), c->fast_flag.as<uint32_t>(), &X);
  RC(pf.end());
  ProjectArgs A{};
  A.ix = ix->dev; A.cfg = dc; A.n_aln = n; A.ref_id = b->ref_id; A.cigar_off = b->cigar_off; A.cigar = b->cigar;
  A.seg = c->seg.as<uint2>(); A.meta = c->meta.as<AlnMeta>(); A.head = c->head.as<uint4>(); A.head2 = c->head2.as<uint4>();
  A.fast_flag = c->fast_flag.as<uint32_t>(); A.fast_pre = c->fast_pre.as<uint32_t>(); A.n_matches = c->n_matches.as<uint32_t>();
  A.ranges = c->ranges.as<uint4>(); A.mask = c->mask.as<uint64_t>();
  A.match_off = c->match_off.as<uint32_t>(); A.cig_base = c->cig_base.as<uint64_t>();
  A.big_list = c->big_list.as<uint32_t>(); A.n_big = c->n_big.as<uint32_t>();
  A.m_aln = c->m_aln.as<uint32_t>(); A.m_tid = c->m_tid.as<uint32_t>(); A.m_aux = c->m_aux.as<uint32_t>(); A.m_p = c->m_p.as<uint2>();
  A.m_x = c->m_x.as<uint2>(); A.m_b = c->m_b.as<uint4>(); A.m_cigoff = c->m_cigoff.as<uint64_t>(); A.cig_arena = c->cig_arena.as<uint32_t>();
  A.tot = d_tot; A.lim_m = cap_m; A.lim_c = cap_c;
  const int n_blocks = c->n_cu * c->blocks_per_cu;
  const bool split = big && c->count_split && !dc.filter_by_similarity;
  if (split) { A.walk_list = c->walk_list.as<uint32_t>(); A.n_walk = c->n_big.as<uint32_t>() + 1; }
  RC(pf.begin(BR_K_COUNT));
  launch_project(st, A, false, c->group_lanes, n_blocks, split ? 1 : 0);   // small: one kernel, the exon walk inline
  RC(pf.end());
  if (split) {
    RC(pf.begin(BR_K_COUNT_WALK));
    launch_project(st, A, false, c->group_lanes, n_blocks, 2);
    RC(pf.end());
  }
  ScanArgs S{};
  S.n = n; S.src32 = c->n_matches.as<uint32_t>(); S.cigar_off = b->cigar_off; S.head = c->head.as<uint4>();
  S.tile_sums = c->tile_sums.as<uint64_t>(); S.fast_flag = c->fast_flag.as<uint32_t>();
  RC(pf.begin(BR_K_SCAN));
  const bool expanded = launch_scan3(st, S, c->match_off.as<uint32_t>(), c->cig_base.as<uint64_t>(), c->fast_pre.as<uint32_t>(), d_tot + 0, &A);
  RC(pf.end());
  if (!expanded) {
    RC(pf.begin(BR_K_EXPAND));
    launch_expand(st, A);
    RC(pf.end());
  }
  if (!big) {
    RC(pf.begin(BR_K_EMIT));
    launch_emit_dense(st, A, (int64_t)cover_m, -1, 0);          // one launch over the whole list; the kernel stops at tot[0]
    launch_project(st, A, true, 64, std::min(c->n_cu, 64));     // alignments with > 64 candidate rows (reads *n_big)
    RC(pf.end());
  } else {
    // as the ordinary path: the dense-locus kernel on the second stream beside the two classes of the work list
    HIPCHK(hipEventRecord(c->aux_ev[0], st));
    HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->aux_ev[0], 0));
    RC(pf.begin(BR_K_EMIT_AUX, c->ksw_stream));
    launch_project(c->ksw_stream, A, true, 64, c->n_cu);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[1], c->ksw_stream));
    if (c->emit_split && !dc.filter_by_similarity) {
      RC(pf.begin(BR_K_EMIT_SIMPLE));
      launch_emit_dense(st, A, (int64_t)cover_m, (int64_t)cover_s, 1);
      RC(pf.end());
      RC(pf.begin(BR_K_EMIT));
      launch_emit_dense(st, A, (int64_t)(cover_g + cover_s), (int64_t)cover_s, 2);   // (its grid: cover_g entries of the general class)
      RC(pf.end());
    } else {
      RC(pf.begin(BR_K_EMIT));
      launch_emit_dense(st, A, (int64_t)cover_m, -1, 0);
      RC(pf.end());
    }
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));
  }

  PairArgs P{};
  P.n_groups = ng; P.n_aln = n; P.long_reads = dc.long_reads; P.group_off = b->group_off; P.mate_idx = b->mate_idx;
  P.aln_group = c->aln_group.as<uint32_t>();
  P.match_off = c->match_off.as<uint32_t>(); P.n_matches = c->n_matches.as<uint32_t>(); P.m_tid = A.m_tid; P.m_p = A.m_p; P.m_x = A.m_x; P.m_b = A.m_b;
  P.m_cigoff = A.m_cigoff;
  P.n_rows = c->n_rows.as<uint32_t>(); P.row_off = c->row_off.as<uint64_t>(); P.counters = d_cnt;
  P.pmask = c->pmask.as<uint64_t>(); P.pbit = c->pbit.as<uint8_t>();
  P.tot = d_tot; P.lim_m = cap_m; P.lim_c = cap_c; P.lim_r = std::min(cap_r, cover_r);
  RC(pf.begin(BR_K_PAIR_COUNT));
  launch_pair(st, P, false);
  RC(pf.end());
  ScanArgs S2{};
  S2.n = n; S2.src32 = c->n_rows.as<uint32_t>(); S2.tile_sums = c->tile_sums.as<uint64_t>();
  if (rows_busy_wait) HIPCHK(hipStreamWaitEvent(st, c->rows_busy, 0));
  RC(pf.begin(BR_K_SCAN));
  launch_scan(st, S2, 2, c->row_off.p, true, d_tot + 3);
  RC(pf.end());
  P.n_rows_total = (int64_t)P.lim_r; P.r_rec = c->r_rec.as<uint4>();
  P.r_a = c->pk_a.as<uint4>(); P.r_c = c->pk_c.as<uint2>(); P.r_x = nullptr;
  P.r_sim = aux_cols ? c->pk_sim.as<double>() : nullptr; P.r_clip = aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
  // a caller that wants the few rows of a small call on the host (br_project_group): the row kernel writes the packed rows
  // and their detail column straight into pinned host memory -- no download, no second wait
  c->rows_at_host = false;
  if (c->rows_to_host) {
    RC(c->g_a.resize(cap_m)); RC(c->g_c.resize(cap_m)); RC(c->g_x.resize(cap_m));
    if (aux_cols) RC(c->g_sim.resize(cap_m));
    P.r_a = c->g_a.p; P.r_c = c->g_c.p; P.r_x = c->g_x.p;
    if (aux_cols) P.r_sim = c->g_sim.p;
    c->rows_at_host = true;
  }
  const uint8_t *names = (b->names && b->name_off) ? b->names : nullptr;
  if (!aux_cols) {   // the primary choice needs row_off and the pair bits only: before the records exist, its pick applied by k_rows
    P.pick = c->pick.as<uint64_t>();
    hipStream_t ps = st;
    if (big) {   // ... and beside the emit pass, on the second stream
      ps = c->ksw_stream;
      HIPCHK(hipEventRecord(c->aux_ev[0], st));
      HIPCHK(hipStreamWaitEvent(ps, c->aux_ev[0], 0));
    }
    RC(pf.begin(BR_K_PRIMARY, ps));
    launch_primary(ps, P, b->name_off, names, false);
    RC(pf.end());
    if (big) HIPCHK(hipEventRecord(c->aux_ev[1], ps));
  }
  RC(pf.begin(BR_K_PAIR_EMIT));
  launch_pair(st, P, true);
  RC(pf.end());
  if (big && !aux_cols) HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));
  if (aux_cols) {
    RC(pf.begin(BR_K_PRIMARY));
    launch_primary(st, P, b->name_off, names, true);
    RC(pf.end());
  }
  RC(pf.begin(BR_K_ROWS));
  launch_rows(st, P, aux_cols);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 32, d_tot, 12 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < 4; k++) { c->h_totals[k] = c->h_totals[32 + k]; c->h_totals[4 + k] = c->h_totals[40 + k]; }
  const uint64_t n_matches = c->h_totals[0], n_cig_arena = c->h_totals[1], n_simple = c->h_totals[2], n_rows = c->h_totals[3];
  // nothing was written past a table or left out by a grid: the kernels checked the same totals and did nothing then
  if (n_matches > cap_m || n_cig_arena > cap_c || n_rows > P.lim_r) return BR_RETRY_ORDINARY;
  if (big && (c->emit_split && !dc.filter_by_similarity ? (n_simple > cover_s || n_matches - n_simple > cover_g) : n_matches > cover_m)) return BR_RETRY_ORDINARY;
  c->hist_simf = dc.filter_by_similarity != 0;
  c->hist_n = n; c->hist[0] = n_matches; c->hist[1] = n_cig_arena; c->hist[2] = n_simple; c->hist[3] = n_rows;
  if (!keep_events) RC(pf.collect());
  if (c->h_totals[7]) return BR_ERR_UNSUPPORTED;  // a rewritten CIGAR with more than 2^24 - 1 ops
  out->n_matches = (int64_t)n_matches; out->n_rows = (int64_t)n_rows; out->n_pool_words = (int64_t)n_cig_arena;
  out->total_complete = n_rows; out->total_unique = c->h_totals[5]; out->dropped_reads = c->h_totals[6];
  out->a = (const br_row_a *)P.r_a; out->cigar = (const uint64_t *)P.r_c; out->x = (const br_row_x *)P.r_x;
  out->similarity_score = aux_cols ? P.r_sim : nullptr;
  out->clip_score = aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
  out->pool = c->cig_arena.as<uint32_t>(); out->row_off = c->row_off.as<uint64_t>();
  c->counters[6] = n_matches;
  c->last_n_rows = (int64_t)n_rows; c->last_n_aln = n; c->last_n_pool = (int64_t)n_cig_arena;
  c->last_aux_cols = aux_cols; c->wide_valid = false; c->detail_valid = false; c->last_l_qseq = b->l_qseq; c->last_long_reads = dc.long_reads;
  return BR_OK;
}

// The HIP pipeline over a device-resident batch.
// Direct rows: presets without the similarity filter and without -S (every short-read preset; long reads with the filter
// switched off).  segment -> count -> [k_pair_mask || k_big<0> + k_pair_big] -> k_scan5 (one host wait: sizes) ->
// k_expand_rows -> k_emit_rows (|| k_big<1>): the packed rows are written once, by the lane that computes the match; the
// match table, the per-record r_rec, k_pair<true> and k_rows do not exist on this path.  On the second stream: the name
// seeds of the primary tie-break (beside segment + count), the big alignments' pairing (beside k_pair_mask), k_group_desc
// (beside the scan), k_big<1> (beside the emit kernels).
static int run_device_direct(br_ctx *c, const DevCfg &dc, const br_device_batch *b, hipStream_t st, br_device_rows *out, Prof &pf,
                             bool keep_events) {
  const br_index *ix = c->ix;
  const int64_t n = b->n_aln, ng = b->n_groups;
  const int64_t tiles = std::max<int64_t>(scan_tiles_for(std::max<int64_t>(n, ng) + 1), 1);
  RC(c->seg.ensure((size_t)(b->n_cigar_words + n) * sizeof(uint2)));
  RC(c->meta.ensure((size_t)n * sizeof(AlnMeta))); RC(c->head.ensure((size_t)n * sizeof(uint4))); RC(c->head2.ensure((size_t)n * sizeof(uint4)));
  RC(c->fast_flag.ensure((size_t)n * 4)); RC(c->n_matches.ensure((size_t)n * 4 + 16)); RC(c->ranges.ensure((size_t)n * sizeof(uint4)));   // (+ 16: k_group_desc reads four elements at a time)
  RC(c->mask.ensure((size_t)n * 8)); RC(c->cig_base.ensure((size_t)(n + 1) * 8)); RC(c->tile_sums.ensure((size_t)tiles * 8 * 5));
  // every small counter of the step in one block, zeroed by one fill at the start: [0, GD_COUNTER_WORDS) the four counters + k_group_desc's
  // slots, then the side arena's two words, then n_big | n_walk | pm_n | -
  RC(c->totals.ensure(16 * 8)); RC(c->counters_d.ensure((GD_COUNTER_WORDS + 4) * 8));
  uint64_t *const dz = c->counters_d.as<uint64_t>();
  uint32_t *const dz_nbig = (uint32_t *)(dz + GD_COUNTER_WORDS + 2);
  RC(c->big_list.ensure((size_t)n * 4)); RC(c->walk_list.ensure((size_t)n * 4));
  RC(c->aln_group.ensure((size_t)n * 4)); RC(c->n_rows.ensure((size_t)n * 4 + 16)); RC(c->pbit.ensure((size_t)n + 16));
  RC(c->d_fm.ensure((size_t)n * sizeof(uint2) + (size_t)(n / 62 + 2) * 4));   // + the window list of k_pair_mask
  RC(c->d_nkept.ensure((size_t)n * 4)); RC(c->d_desc.ensure((size_t)n * sizeof(uint4)));
  RC(c->d_hi0.ensure((size_t)n * 4)); RC(c->d_clspos.ensure((size_t)n * 4)); RC(c->d_rnd.ensure((size_t)std::max<int64_t>(ng, 1) * 8));
  if (c->d_side_cap == 0) c->d_side_cap = std::max<uint64_t>((uint64_t)n / 4, 1u << 20);
  RC(c->d_side.ensure((size_t)c->d_side_cap * sizeof(uint2)));
  // (a buffer that a queued packed download still reads must not be reallocated under it)
  if (c->rows_busy_set && c->row_off.cap < (size_t)(n + 1) * 8) HIPCHK(hipEventSynchronize(c->rows_busy));
  RC(c->row_off.ensure((size_t)(n + 1) * 8));
  RC(ensure_aux_stream(c));
  hipStream_t ax = c->ksw_stream;
  uint64_t *d_tot = c->totals.as<uint64_t>();

  ProjectArgs A{};
  A.ix = ix->dev; A.cfg = dc; A.n_aln = n; A.ref_id = b->ref_id; A.cigar_off = b->cigar_off; A.cigar = b->cigar;
  A.seg = c->seg.as<uint2>(); A.meta = c->meta.as<AlnMeta>(); A.head = c->head.as<uint4>(); A.head2 = c->head2.as<uint4>();
  A.fast_flag = c->fast_flag.as<uint32_t>(); A.n_matches = c->n_matches.as<uint32_t>();
  A.ranges = c->ranges.as<uint4>(); A.mask = c->mask.as<uint64_t>();
  A.big_list = c->big_list.as<uint32_t>(); A.n_big = dz_nbig;
  if (c->count_split) { A.walk_list = c->walk_list.as<uint32_t>(); A.n_walk = dz_nbig + 1; }
  const bool have_names = b->names && b->name_off;
  DirectArgs D{};
  D.n_aln = n; D.n_groups = ng; D.group_off = b->group_off; D.aln_group = c->aln_group.as<uint32_t>(); D.mate_idx = b->mate_idx;
  D.n_matches = c->n_matches.as<uint32_t>(); D.mask = c->mask.as<uint64_t>(); D.ranges = c->ranges.as<uint4>();
  D.fast_flag = c->fast_flag.as<uint32_t>(); D.s_tid = ix->dev.s_tid; D.big_list = A.big_list; D.n_big = A.n_big;
  D.fm = c->d_fm.as<uint2>(); D.pm_list = (uint32_t *)(c->d_fm.as<uint2>() + n); D.pm_n = dz_nbig + 2; D.n_kept = c->d_nkept.as<uint32_t>(); D.n_rows = c->n_rows.as<uint32_t>(); D.pflag = c->pbit.as<uint8_t>();
  D.side = c->d_side.as<uint2>(); D.side_cap = c->d_side_cap; D.side_used = (unsigned long long *)(dz + GD_COUNTER_WORDS);
  D.cls_pos = c->d_clspos.as<uint32_t>(); D.cig_base = c->cig_base.as<uint64_t>(); D.row_off = c->row_off.as<uint64_t>();
  D.name_off = have_names ? b->name_off : nullptr; D.names = have_names ? b->names : nullptr; D.rnd0 = c->d_rnd.as<uint64_t>();
  D.gd = c->d_desc.as<uint2>(); D.dpos = c->d_desc.as<uint2>() + n; D.hi0 = c->d_hi0.as<uint32_t>(); D.counters = c->counters_d.as<uint64_t>(); D.tot = d_tot;

  if (!c->aux2_stream) { HIPCHK(hipStreamCreateWithPriority(&c->aux2_stream, hipStreamNonBlocking, aux_stream_priority())); HIPCHK(hipEventCreateWithFlags(&c->aux2_ev, hipEventDisableTiming)); }
  hipStream_t ax2 = c->aux2_stream;
  HIPCHK(hipMemsetAsync(dz, 0, (GD_COUNTER_WORDS + 4) * 8, st));
  HIPCHK(hipEventRecord(c->aux_ev[0], st));
  HIPCHK(hipStreamWaitEvent(ax, c->aux_ev[0], 0));
  // a1/a2/a6: CIGAR -> read exons; a3-a8, a11-a14 (survival only): the count pass
  RC(pf.begin(BR_K_SEGMENT));
  launch_segment(st, n, b->ref_id, b->ref_start, b->flags, b->xs, b->ts, b->cigar_off, b->cigar, dc, ix->n_refs,
                 c->seg.as<uint2>(), c->meta.as<AlnMeta>(), c->head.as<uint4>(), c->head2.as<uint4>(), c->fast_flag.as<uint32_t>());
  RC(pf.end());
  RC(pf.begin(BR_K_GROUP_IDS));
  launch_group_ids(st, ng, b->group_off, c->aln_group.as<uint32_t>());
  RC(pf.end());
  const int n_blocks = c->n_cu * c->blocks_per_cu;
  const bool split = A.walk_list != nullptr;
  RC(pf.begin(BR_K_COUNT));
  launch_project(st, A, false, c->group_lanes, n_blocks, split ? 1 : 0);
  RC(pf.end());
  if (split) {
    RC(pf.begin(BR_K_COUNT_WALK));
    launch_project(st, A, false, c->group_lanes, n_blocks, 2);
    RC(pf.end());
  }
  // a packed download of the previous call may still be reading row_off / the row tables (br_project_staged)
  if (c->rows_busy_set) { HIPCHK(hipStreamWaitEvent(st, c->rows_busy, 0)); }
  // a16 (src/mates.cpp:150-261) on the survivor sets, then placement
  const int big_blocks = c->n_cu * 4;
  uint64_t kept = 0, arena = 0, n_simple = 0, n_rows = 0, n_raw = 0;
  bool expanded_ahead = false;
  // third stream: the name seeds need nothing but the names, and their 156 dependent multiplies per read name are pure ALU work:
  // beside k_pair_mask, which waits on LDS and memory most of the time
  if (have_names) {
    HIPCHK(hipEventRecord(c->aux_ev[7], st));
    HIPCHK(hipStreamWaitEvent(ax2, c->aux_ev[7], 0));
    RC(pf.begin(BR_K_NAME_SEED, ax2));
    launch_name_seed(ax2, D);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux2_ev, ax2));
  }
  for (int attempt = 0;; attempt++) {
    if (attempt) {   // (the first attempt's counters were zeroed with everything else at the start)
      HIPCHK(hipMemsetAsync(dz, 0, (GD_COUNTER_WORDS + 2) * 8, st));
      HIPCHK(hipMemsetAsync(D.pm_n, 0, 4, st));
    }
    HIPCHK(hipEventRecord(c->aux_ev[1], st));
    HIPCHK(hipStreamWaitEvent(ax, c->aux_ev[1], 0));
    RC(pf.begin(BR_K_PAIR_BIG, ax));
    launch_big_collect(ax, A, D, big_blocks);
    launch_pair_big(ax, D, big_blocks);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[2], ax));
    RC(pf.begin(BR_K_PAIR_MASK));
    launch_pair_mask(st, D, c->n_cu * 2);
    RC(pf.end());
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[2], 0));
    // NH / HI / primary per read name on the second stream beside the scan
    HIPCHK(hipEventRecord(c->aux_ev[3], st));
    HIPCHK(hipStreamWaitEvent(ax, c->aux_ev[3], 0));
    if (have_names) HIPCHK(hipStreamWaitEvent(ax, c->aux2_ev, 0));
    RC(pf.begin(BR_K_GROUP_DESC, ax));
    launch_group_desc(ax, D);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[4], ax));
    RC(pf.begin(BR_K_SCAN));
    launch_scan5(st, D, c->tile_sums.as<uint64_t>(), d_tot);
    RC(pf.end());
    // the work list goes out at once, into the list the last call left (it checks the total against that room on the
    // device): it runs while the host waits for the totals, wakes up and sizes the row tables
    expanded_ahead = false;
    if (c->m_aln.cap >= 4) {
      D.m_aln = c->m_aln.as<uint32_t>(); D.m_aln_cap = c->m_aln.cap / 4;
      RC(pf.begin(BR_K_EXPAND_ROWS));
      launch_expand_rows(st, D);
      RC(pf.end());
      expanded_ahead = true;
    }
    HIPCHK(hipMemcpyAsync(c->h_totals, d_tot, 5 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(c->h_totals + 8, dz + GD_COUNTER_WORDS, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c->h_totals[9]) {   // the side arena of the > 64-candidate alignments ran out: grow it to what was asked for, repeat
      if (attempt >= 3) return BR_ERR_CAPACITY;
      HIPCHK(hipEventSynchronize(c->aux_ev[4]));
      c->d_side_cap = c->h_totals[8] + c->h_totals[8] / 4 + 4096;
      RC(c->d_side.ensure((size_t)c->d_side_cap * sizeof(uint2)));
      D.side = c->d_side.as<uint2>(); D.side_cap = c->d_side_cap;
      continue;
    }
    kept = c->h_totals[0]; arena = c->h_totals[1]; n_simple = c->h_totals[2]; n_rows = c->h_totals[3]; n_raw = c->h_totals[4];
    break;
  }
  if (n_raw >= 0xffffffffull || kept >= 0xffffffffull) return BR_ERR_CAPACITY;
  if (kept != n_rows) return BR_ERR_HIP;   // (every kept match is one record: the pairing kernels disagree with themselves)
  out->n_matches = (int64_t)n_raw; out->n_rows = (int64_t)n_rows; out->n_pool_words = (int64_t)arena;

  const size_t nr = (size_t)std::max<uint64_t>(n_rows, 1);
  if (c->rows_busy_set && (c->pk_a.cap < nr * sizeof(uint4) || c->pk_c.cap < nr * sizeof(uint2))) HIPCHK(hipEventSynchronize(c->rows_busy));
  RC(c->pk_a.ensure(nr * sizeof(uint4))); RC(c->pk_c.ensure(nr * sizeof(uint2)));
  if (expanded_ahead && kept > D.m_aln_cap) expanded_ahead = false;   // (the kernel saw the same and did nothing)
  RC(c->m_aln.ensure(nr * 4)); RC(c->cig_arena.ensure((size_t)std::max<uint64_t>(arena, 1) * 4));
  const bool with_x = c->want_x;
  if (with_x) {
    if (c->rows_busy_set && c->pk_x.cap < nr * sizeof(uint4)) HIPCHK(hipEventSynchronize(c->rows_busy));
    RC(c->pk_x.ensure(nr * sizeof(uint4)));
  }
  A.cig_arena = c->cig_arena.as<uint32_t>();
  D.m_aln = c->m_aln.as<uint32_t>(); D.r_a = c->pk_a.as<uint4>(); D.r_c = c->pk_c.as<uint2>(); D.r_x = with_x ? c->pk_x.as<uint4>() : nullptr;
  const bool emit_split = c->emit_split != 0;
  D.m_aln_cap = 0;
  if (kept) {
    if (!expanded_ahead) {
      RC(pf.begin(BR_K_EXPAND_ROWS));
      launch_expand_rows(st, D);
      RC(pf.end());
    }
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[4], 0));   // k_group_desc: the descriptors' first halves
    HIPCHK(hipEventRecord(c->aux_ev[5], st));
    HIPCHK(hipStreamWaitEvent(ax, c->aux_ev[5], 0));
    RC(pf.begin(BR_K_BIG_EMIT, ax));
    launch_big_emit(ax, A, D, big_blocks);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[6], ax));
    if (emit_split) {
      RC(pf.begin(BR_K_EMIT_ROWS_SIMPLE));
      launch_emit_rows(st, A, D, (int64_t)kept, (int64_t)n_simple, 1);
      RC(pf.end());
      RC(pf.begin(BR_K_EMIT_ROWS));
      launch_emit_rows(st, A, D, (int64_t)kept, (int64_t)n_simple, 2);
      RC(pf.end());
    } else {
      RC(pf.begin(BR_K_EMIT_ROWS));
      launch_emit_rows(st, A, D, (int64_t)kept, (int64_t)n_simple, 0);
      RC(pf.end());
    }
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[6], 0));
  } else {
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[4], 0));
  }
  HIPCHK(hipMemcpyAsync(c->h_totals + 192, c->counters_d.p, GD_COUNTER_WORDS * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!keep_events) RC(pf.collect());
  for (int k = 0; k < 4; k++) c->h_totals[4 + k] = c->h_totals[192 + k];
  for (int k = 0; k < GD_SLOTS; k++) { c->h_totals[5] += c->h_totals[192 + GD_SLOT0 + k * GD_SLOT_STRIDE]; c->h_totals[6] += c->h_totals[192 + GD_SLOT0 + k * GD_SLOT_STRIDE + 1]; }
  if (c->h_totals[7]) return BR_ERR_UNSUPPORTED;  // a rewritten CIGAR with more than 2^24 - 1 ops, or NH beyond 28 bits
  out->total_complete = n_rows; out->total_unique = c->h_totals[5]; out->dropped_reads = c->h_totals[6];
  c->hist_n = 0;   // (nothing a later speculative launch of the match-table path could be sized from)
  out->a = (const br_row_a *)c->pk_a.p; out->cigar = (const uint64_t *)c->pk_c.p; out->x = with_x ? (const br_row_x *)c->pk_x.p : nullptr;
  out->similarity_score = nullptr; out->clip_score = nullptr;
  out->pool = c->cig_arena.as<uint32_t>(); out->row_off = c->row_off.as<uint64_t>();
  c->counters[6] = n_raw;
  c->last_n_rows = (int64_t)n_rows; c->last_n_aln = n; c->last_n_pool = (int64_t)arena;
  c->last_aux_cols = false; c->wide_valid = false; c->detail_valid = with_x; c->last_l_qseq = b->l_qseq; c->last_long_reads = dc.long_reads;
  c->last_direct = true; c->dA = A; c->dD = D; c->d_kept = (int64_t)kept; c->d_simple = (int64_t)n_simple; c->d_split = emit_split;
  return BR_OK;
}

static int run_device_impl(br_ctx *c, const br_config *cfg, const br_device_batch *b, hipStream_t st, br_device_rows *out,
                           bool keep_events);
int run_device(br_ctx *c, const br_config *cfg, const br_device_batch *b, hipStream_t st, br_device_rows *out) {
  return run_device_impl(c, cfg, b, st, out, false);
}
static int run_device_impl(br_ctx *c, const br_config *cfg, const br_device_batch *b, hipStream_t st, br_device_rows *out,
                           bool keep_events) {
  const br_index *ix = c->ix;
  memset(out, 0, sizeof(*out));
  DevCfg dc;
  RC(make_devcfg(cfg, dc));
  // -S only changes long-read runs (src/evaluate.cpp:916-919,939)
  const bool fa_mode = dc.use_fasta && dc.long_reads;
  if (fa_mode && (!ix->has_seq || !b->seq_src || !b->seq_off || !b->seqs)) return BR_ERR_INVALID_ARG;
  int64_t n = b->n_aln, ng = b->n_groups;
  if (n < 0 || ng < 0 || n >= 0x7fffffffll || b->n_cigar_words >= 0xffffffffll - n) return BR_ERR_CAPACITY;
  HIPCHK(hipSetDevice(ix->device));
  Prof pf{c, st};
  if (!keep_events) c->events_used = 0;
  out->total_processed = (uint64_t)n;
  c->last_n_rows = 0; c->last_n_aln = n; c->last_n_pool = 0; c->wide_valid = false; c->last_aux_cols = false; c->last_direct = false;
  c->last_l_qseq = b->l_qseq; c->last_long_reads = dc.long_reads;
  if (n == 0) { pf.collect(); return BR_OK; }
  if (!fa_mode && c->small_batch && ix->dev.n_rows != 0) {
    // small batches always; large ones when an earlier call left tables and counts to predict from (same preset class)
    const bool small = n <= c->small_n;
    // (up to speculate_n alignments: at 20 M alignments the three waits are 2 % of the step and the 15 % of empty blocks in
    // the predicted grids cost as much, profiles/r03/ab_speculate.log; at 0.1-1 M alignments the step gets 6-13 % shorter)
    const bool big = !small && n <= c->speculate_n && c->speculate && c->hist_n > 0 && c->hist_simf == (dc.filter_by_similarity != 0) && !c->single_pass;
    if (small || big) {
      const int rc = run_device_small(c, dc, b, st, out, pf, keep_events, big);
      if (rc != BR_RETRY_ORDINARY) return rc;
      if (!keep_events) c->events_used = 0;
    }
  }

  if (!fa_mode && !dc.filter_by_similarity && c->direct_rows && !c->single_pass) return run_device_direct(c, dc, b, st, out, pf, keep_events);

  int64_t tiles = std::max<int64_t>(scan_tiles_for(std::max<int64_t>(n, ng) + 1), 1);
  RC(c->seg.ensure((size_t)(b->n_cigar_words + n) * sizeof(uint2)));
  RC(c->meta.ensure((size_t)n * sizeof(AlnMeta))); RC(c->head.ensure((size_t)n * sizeof(uint4))); RC(c->head2.ensure((size_t)n * sizeof(uint4)));
  RC(c->fast_flag.ensure((size_t)n * 4)); RC(c->fast_pre.ensure((size_t)(n + 1) * 4));
  RC(c->n_matches.ensure((size_t)n * 4)); RC(c->ranges.ensure((size_t)n * sizeof(uint4)));
  RC(c->mask.ensure((size_t)n * 8)); RC(c->match_off.ensure((size_t)(n + 1) * 4));
  RC(c->cig_base.ensure((size_t)(n + 1) * 8)); RC(c->tile_sums.ensure((size_t)tiles * 8 * 3));
  RC(c->totals.ensure(16 * 8)); RC(c->counters_d.ensure(4 * 8));
  uint64_t *d_tot = c->totals.as<uint64_t>();

  // a1/a2/a6: CIGAR -> read exons
  RC(pf.begin(BR_K_SEGMENT));
  launch_segment(st, n, b->ref_id, b->ref_start, b->flags, b->xs, b->ts, b->cigar_off, b->cigar, dc, ix->n_refs,
                 c->seg.as<uint2>(), c->meta.as<AlnMeta>(), c->head.as<uint4>(), c->head2.as<uint4>(),
                 c->fast_flag.as<uint32_t>());
  RC(pf.end());
  ProjectArgs A{};
  A.ix = ix->dev; A.cfg = dc; A.n_aln = n; A.ref_id = b->ref_id; A.cigar_off = b->cigar_off; A.cigar = b->cigar;
  A.seg = c->seg.as<uint2>(); A.meta = c->meta.as<AlnMeta>(); A.head = c->head.as<uint4>(); A.head2 = c->head2.as<uint4>();
  A.fast_flag = c->fast_flag.as<uint32_t>(); A.fast_pre = c->fast_pre.as<uint32_t>();
  A.n_matches = c->n_matches.as<uint32_t>();
  A.ranges = c->ranges.as<uint4>(); A.mask = c->mask.as<uint64_t>();
  A.match_off = c->match_off.as<uint32_t>(); A.cig_base = c->cig_base.as<uint64_t>();
  RC(c->big_list.ensure((size_t)n * 4)); RC(c->n_big.ensure(16));
  HIPCHK(hipMemsetAsync(c->n_big.p, 0, 8, st));

  A.big_list = c->big_list.as<uint32_t>(); A.n_big = c->n_big.as<uint32_t>();
  if (c->count_split) { RC(c->walk_list.ensure((size_t)n * 4)); A.walk_list = c->walk_list.as<uint32_t>(); A.n_walk = c->n_big.as<uint32_t>() + 1; }
  int n_blocks = c->n_cu * c->blocks_per_cu;
  ScanArgs S{};
  S.n = n; S.src32 = c->n_matches.as<uint32_t>(); S.cigar_off = b->cigar_off; S.head = c->head.as<uint4>();
  S.tile_sums = c->tile_sums.as<uint64_t>(); S.fast_flag = c->fast_flag.as<uint32_t>();
  FaArgs F{};
  // Short-read presets (no similarity filter, no -S): count and emit in one sweep.  The match table, the general class's
  // work list and the CIGAR arena are sized from the last call's high-water marks (first call: a guess); a pass that runs
  // out of room says so, and is run again with what its counters ask for.
  bool one_pass = !fa_mode && !dc.filter_by_similarity && c->single_pass && c->count_split && ix->dev.n_rows < 0x7fffffffu &&
                  (c->group_lanes == 8 || c->group_lanes == 16);
  uint64_t p1_matches = 0, p1_entries = 0, p1_words = 0;
  if (one_pass) {
    RC(c->p1.ensure(P1_WORDS * 8));
    RC(c->walk_list.ensure((size_t)n * 4)); A.walk_list = c->walk_list.as<uint32_t>(); A.n_walk = c->n_big.as<uint32_t>() + 1;
    int nb1 = n_blocks;
    { const int64_t need = (n + 127) / 128 / 4 + 1; if (need < nb1) nb1 = (int)need; }
    const uint64_t n_waves = (uint64_t)nb1 * 4;
    uint64_t cap_m = std::max<uint64_t>(c->p1_hw[0] + c->p1_hw[0] / 4, 8ull * (uint64_t)n) + n_waves * P1_PAGE_M + 4096;
    uint64_t cap_w = std::max<uint64_t>(c->p1_hw[1] + c->p1_hw[1] / 4, 4ull * (uint64_t)n) + n_waves * P1_PAGE_W + 4096;
    uint64_t cap_c = std::max<uint64_t>(c->p1_hw[2] + c->p1_hw[2] / 4, 48ull * (uint64_t)n) + n_waves * P1_PAGE_C + 4096;
    for (int attempt = 0;; attempt++) {
      if (cap_m >= 0xffffffffull || cap_w >= 0xffffffffull) { one_pass = false; break; }   // slots are 32-bit: the two-pass path reports the capacity error
      RC(c->m_tid.ensure(cap_m * 4)); RC(c->m_aux.ensure(cap_m * 4)); RC(c->m_p.ensure(cap_m * sizeof(uint2))); RC(c->m_x.ensure(cap_m * sizeof(uint2)));
      RC(c->m_cigoff.ensure(cap_m * 8)); RC(c->wl.ensure(cap_w * sizeof(uint4))); RC(c->cig_arena.ensure(cap_c * 4));
      A.m_tid = c->m_tid.as<uint32_t>(); A.m_aux = c->m_aux.as<uint32_t>(); A.m_p = c->m_p.as<uint2>(); A.m_x = c->m_x.as<uint2>();
      A.m_b = nullptr; A.m_cigoff = c->m_cigoff.as<uint64_t>(); A.cig_arena = c->cig_arena.as<uint32_t>();
      A.match_off_w = c->match_off.as<uint32_t>(); A.cig_base_w = c->cig_base.as<uint64_t>(); A.wl = c->wl.as<uint4>();
      A.p1 = c->p1.as<uint64_t>(); A.cap_m = cap_m; A.cap_w = cap_w; A.cap_c = cap_c;
      HIPCHK(hipMemsetAsync(c->p1.p, 0, P1_WORDS * 8, st));
      HIPCHK(hipMemsetAsync(c->n_big.p, 0, 8, st));
      RC(pf.begin(BR_K_P1));
      launch_project1(st, A, c->group_lanes, n_blocks, 1);
      RC(pf.end());
      RC(pf.begin(BR_K_P1_WALK));
      launch_project1(st, A, c->group_lanes, n_blocks, 2);
      RC(pf.end());
      HIPCHK(hipMemcpyAsync(c->h_totals + 96, c->p1.p, P1_WORDS * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      const uint64_t *h = c->h_totals + 96;
      if (!h[P1_OVF]) { p1_matches = h[P1_NM]; p1_entries = h[P1_W]; p1_words = h[P1_C]; c->p1_hw[0] = h[P1_M]; c->p1_hw[1] = h[P1_W]; c->p1_hw[2] = h[P1_C]; break; }
      if (attempt >= 4) return BR_ERR_CAPACITY;
      // what the counters reached (waves that ran out stopped placing, but kept counting their requests) + a margin
      cap_m = std::max(cap_m, h[P1_M]) + std::max(cap_m, h[P1_M]) / 4 + n_waves * P1_PAGE_M;
      cap_w = std::max(cap_w, h[P1_W]) + std::max(cap_w, h[P1_W]) / 4 + n_waves * P1_PAGE_W;
      cap_c = std::max(cap_c, h[P1_C]) + std::max(cap_c, h[P1_C]) / 4 + n_waves * P1_PAGE_C;
    }
    if (!one_pass) { A.p1 = nullptr; A.wl = nullptr; HIPCHK(hipMemsetAsync(c->n_big.p, 0, 8, st)); }
  }
  if (one_pass) {
    // nothing: counted and placed above
  } else if (!fa_mode) {
    const bool split = A.walk_list && !dc.filter_by_similarity;
    RC(pf.begin(BR_K_COUNT));
    launch_project(st, A, false, c->group_lanes, n_blocks, split ? 1 : 0);
    RC(pf.end());
    if (split) {
      RC(pf.begin(BR_K_COUNT_WALK));
      launch_project(st, A, false, c->group_lanes, n_blocks, 2);
      RC(pf.end());
    }
    RC(pf.begin(BR_K_SCAN));
    launch_scan3(st, S, c->match_off.as<uint32_t>(), c->cig_base.as<uint64_t>(), c->fast_pre.as<uint32_t>(), d_tot + 0);
    RC(pf.end());
  } else {
    // rescue planning -> ksw2 DP -> count with the rescue results
    RC(c->fa_n_prob.ensure((size_t)n * 4)); RC(c->fa_seq_bytes.ensure((size_t)n * 4));
    RC(c->fa_prob_off.ensure((size_t)(n + 1) * 4)); RC(c->fa_seqarena_off.ensure((size_t)(n + 1) * 8));
    RC(c->fa_ideal_cap.ensure((size_t)n * 4)); RC(c->fa_want.ensure((size_t)n * 16));
    F.seq_src = b->seq_src; F.seq_off = b->seq_off; F.seqs = b->seqs;
    F.n_prob = c->fa_n_prob.as<uint32_t>(); F.seq_bytes = c->fa_seq_bytes.as<uint32_t>();
    F.prob_off = c->fa_prob_off.as<uint32_t>(); F.seqarena_off = c->fa_seqarena_off.as<uint64_t>();
    F.ideal_cap = c->fa_ideal_cap.as<uint32_t>(); F.want_l = c->fa_want.as<uint64_t>(); F.want_r = F.want_l + n;
    RC(pf.begin(BR_K_COUNT));
    launch_project_fa(st, A, F, 0, n_blocks);
    RC(pf.end());
    ScanArgs SP{}; SP.n = n; SP.src32 = F.n_prob; SP.tile_sums = c->tile_sums.as<uint64_t>();
    ScanArgs SB{}; SB.n = n; SB.src32 = F.seq_bytes; SB.tile_sums = c->tile_sums.as<uint64_t>();
    RC(pf.begin(BR_K_SCAN));
    launch_scan(st, SP, 2, c->fa_prob_off.p, false, d_tot + 4);
    launch_scan(st, SB, 2, c->fa_seqarena_off.p, true, d_tot + 5);
    RC(pf.end());
    HIPCHK(hipMemcpyAsync(c->h_totals + 4, d_tot + 4, 2 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t n_prob = c->h_totals[4], seq_total = c->h_totals[5];
    RC(c->fa_stats.ensure(16));
    HIPCHK(hipMemsetAsync(c->fa_stats.p, 0, 16, st));
    c->rescue_stats[0] = n_prob; c->rescue_stats[1] = 0; c->rescue_stats[2] = 0; c->rescue_stats[3] = seq_total;
    if (n_prob >= 0xffffffffull) return BR_ERR_CAPACITY;
    RC(c->fa_probs.ensure(std::max<size_t>(n_prob, 1) * ksw_prob_bytes()));
    RC(c->fa_results.ensure(std::max<size_t>(n_prob, 1) * ksw_res_bytes()));
    RC(c->fa_seq_arena.ensure((size_t)seq_total + 1024));   // the streamed DP reads whole dwords past a problem's last base
    RC(c->fa_clip_ops.ensure((std::max<size_t>(seq_total + n_prob, 1)) * 4));
    RC(c->fa_srcs.ensure(std::max<size_t>(n_prob, 1) * sizeof(FaSrc)));
    F.probs = (KswProb *)c->fa_probs.p; F.results = (KswRes *)c->fa_results.p; F.srcs = c->fa_srcs.as<FaSrc>();
    F.seq_arena = c->fa_seq_arena.as<uint8_t>(); F.clip_ops = c->fa_clip_ops.as<uint32_t>();
    if (n_prob) {
      RC(pf.begin(BR_K_COUNT));
      launch_project_fa(st, A, F, 1, n_blocks);
      launch_fa_fill(st, A, F, (int64_t)n_prob);
      RC(pf.end());
      uint64_t qmax = (uint64_t)std::max(b->max_soft_clip, 0) + std::max(dc.max_clip, dc.max_junc_ins);
      KswRun R{};
      R.n_prob = (int64_t)n_prob; R.probs = F.probs; R.results = F.results; R.seq_arena = F.seq_arena; R.clip_ops = F.clip_ops;
      R.seq_total = seq_total; R.qmax = qmax; R.tmax = qmax + 40; R.stats = c->fa_stats.as<uint64_t>();
      RC(pf.begin(BR_K_KSW));
      RC(run_ksw(c, st, R));
      RC(pf.end());
      HIPCHK(hipMemcpyAsync(c->h_totals + 8, c->fa_stats.p, 16, hipMemcpyDeviceToHost, st));
    }
    RC(pf.begin(BR_K_COUNT));
    launch_project_fa(st, A, F, 2, n_blocks);
    RC(pf.end());
    S.ideal_cap = F.ideal_cap;
    RC(pf.begin(BR_K_SCAN));
    launch_scan3(st, S, c->match_off.as<uint32_t>(), c->cig_base.as<uint64_t>(), c->fast_pre.as<uint32_t>(), d_tot + 0);
    RC(pf.end());
  }
  uint64_t n_matches = 0, n_cig_arena = 0;
  int64_t n_simple = -1;
  if (one_pass) { n_matches = p1_matches; n_cig_arena = p1_words; }
  else {
    HIPCHK(hipMemcpyAsync(c->h_totals, d_tot, 3 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    n_matches = c->h_totals[0]; n_cig_arena = c->h_totals[1];
    n_simple = fa_mode ? -1 : (int64_t)c->h_totals[2];  // matches of the single-M class (first in the emit list)
  }
  if (n_matches >= 0xffffffffull) return BR_ERR_CAPACITY;
  out->n_matches = (int64_t)n_matches;

  if (!one_pass) {
    size_t nm = (size_t)std::max<uint64_t>(n_matches, 1);
    RC(c->m_tid.ensure(nm * 4)); RC(c->m_aux.ensure(nm * 4)); RC(c->m_p.ensure(nm * sizeof(uint2))); RC(c->m_x.ensure(nm * sizeof(uint2)));
    RC(c->m_b.ensure(nm * sizeof(uint4))); RC(c->m_cigoff.ensure(nm * 8)); RC(c->m_aln.ensure(nm * 4));
    A.m_aln = c->m_aln.as<uint32_t>();
    RC(c->cig_arena.ensure((size_t)std::max<uint64_t>(n_cig_arena, 1) * 4));
    A.m_tid = c->m_tid.as<uint32_t>(); A.m_aux = c->m_aux.as<uint32_t>(); A.m_p = c->m_p.as<uint2>(); A.m_x = c->m_x.as<uint2>();
    A.m_b = c->m_b.as<uint4>(); A.m_cigoff = c->m_cigoff.as<uint64_t>(); A.cig_arena = c->cig_arena.as<uint32_t>();
  }
  if (n_matches && one_pass) {
    // alignments with > 64 candidate rows (or more survivors than the stash holds): a few long-running blocks on the second
    // stream beside the work-list kernel
    RC(ensure_aux_stream(c));
    HIPCHK(hipEventRecord(c->aux_ev[0], st));
    HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->aux_ev[0], 0));
    RC(pf.begin(BR_K_EMIT_AUX, c->ksw_stream));
    launch_project(c->ksw_stream, A, true, 64, c->n_cu);
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[1], c->ksw_stream));
    RC(pf.begin(BR_K_EMIT_WL));
    launch_emit_wl(st, A, (int64_t)p1_entries);
    RC(pf.end());
    HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));
  } else if (n_matches) {
    if (fa_mode) {
      // the work list + one lane per match for alignments with at most 64 candidate rows, k_project_fa<3> for the others
      RC(pf.begin(BR_K_EXPAND));
      launch_expand(st, A);
      RC(pf.end());
      RC(pf.begin(BR_K_EMIT));
      launch_project_fa(st, A, F, 3, n_blocks);
      launch_emit_dense_fa(st, A, F, (int64_t)n_matches);
      RC(pf.end());
    } else {
      // alignments with > 64 candidate rows only: a few long-running blocks, on the second stream beside the work list
      RC(ensure_aux_stream(c));
      HIPCHK(hipEventRecord(c->aux_ev[0], st));
      HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->aux_ev[0], 0));
      RC(pf.begin(BR_K_EMIT_AUX, c->ksw_stream));
      launch_project(c->ksw_stream, A, true, 64, c->n_cu);
      RC(pf.end());
      HIPCHK(hipEventRecord(c->aux_ev[1], c->ksw_stream));
      RC(pf.begin(BR_K_EXPAND));
      launch_expand(st, A);
      RC(pf.end());
      if (c->emit_split && n_simple >= 0 && !dc.filter_by_similarity) {
        RC(pf.begin(BR_K_EMIT_SIMPLE));
        launch_emit_dense(st, A, (int64_t)n_matches, n_simple, 1);
        RC(pf.end());
        RC(pf.begin(BR_K_EMIT));
        launch_emit_dense(st, A, (int64_t)n_matches, n_simple, 2);
        RC(pf.end());
      } else {
        RC(pf.begin(BR_K_EMIT));
        launch_emit_dense(st, A, (int64_t)n_matches, -1, 0);
        RC(pf.end());
      }
      HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));
    }
  }

  // a16/a17: pairing + NH -> the packed row table
  // (a buffer that a queued packed download still reads must not be reallocated under it)
  if (c->rows_busy_set && c->row_off.cap < (size_t)(n + 1) * 8) HIPCHK(hipEventSynchronize(c->rows_busy));
  RC(c->n_rows.ensure((size_t)n * 4)); RC(c->row_off.ensure((size_t)(n + 1) * 8)); RC(c->aln_group.ensure((size_t)n * 4));
  RC(pf.begin(BR_K_GROUP_IDS));
  launch_group_ids(st, ng, b->group_off, c->aln_group.as<uint32_t>());
  RC(pf.end());
  HIPCHK(hipMemsetAsync(c->counters_d.p, 0, 4 * 8, st));
  PairArgs P{};
  P.n_groups = ng; P.n_aln = n; P.long_reads = dc.long_reads; P.group_off = b->group_off; P.mate_idx = b->mate_idx;
  P.aln_group = c->aln_group.as<uint32_t>();
  P.match_off = c->match_off.as<uint32_t>(); P.n_matches = c->n_matches.as<uint32_t>(); P.m_tid = A.m_tid; P.m_p = A.m_p; P.m_x = A.m_x; P.m_b = A.m_b;
  P.m_cigoff = A.m_cigoff;
  P.n_rows = c->n_rows.as<uint32_t>();
  P.row_off = c->row_off.as<uint64_t>(); P.counters = c->counters_d.as<uint64_t>();
  RC(c->pmask.ensure((size_t)n * 8)); P.pmask = c->pmask.as<uint64_t>();
  RC(c->pbit.ensure((size_t)n)); P.pbit = c->pbit.as<uint8_t>();
  RC(pf.begin(BR_K_PAIR_COUNT));
  launch_pair(st, P, false);
  RC(pf.end());
  ScanArgs S2{};
  S2.n = n; S2.src32 = c->n_rows.as<uint32_t>(); S2.tile_sums = c->tile_sums.as<uint64_t>();
  // a packed download of the previous call may still be reading row_off / the row tables (br_project_staged)
  if (c->rows_busy_set) { HIPCHK(hipStreamWaitEvent(st, c->rows_busy, 0)); }
  RC(pf.begin(BR_K_SCAN));
  launch_scan(st, S2, 2, c->row_off.p, true, d_tot + 2);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 2, d_tot + 2, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t n_rows = c->h_totals[2];
  out->n_rows = (int64_t)n_rows; out->n_pool_words = (int64_t)n_cig_arena;

  const size_t nr = (size_t)std::max<uint64_t>(n_rows, 1);
  // clip score / similarity score columns exist only when the preset filters by similarity (long reads): else all zero
  const bool aux_cols = dc.filter_by_similarity != 0;
  if (c->rows_busy_set && (c->pk_a.cap < nr * sizeof(uint4) || (aux_cols && (c->pk_sim.cap < nr * 8 || c->pk_clip.cap < nr * 4))))
    HIPCHK(hipEventSynchronize(c->rows_busy));
  RC(c->r_rec.ensure(nr * sizeof(uint4)));
  RC(c->pk_a.ensure(nr * sizeof(uint4))); RC(c->pk_c.ensure(nr * sizeof(uint2)));
  if (aux_cols) { RC(c->pk_sim.ensure(nr * 8)); RC(c->pk_clip.ensure(nr * 4)); }
  P.n_rows_total = (int64_t)n_rows; P.r_rec = c->r_rec.as<uint4>();
  P.r_a = c->pk_a.as<uint4>(); P.r_c = c->pk_c.as<uint2>(); P.r_x = nullptr;
  P.r_sim = aux_cols ? c->pk_sim.as<double>() : nullptr; P.r_clip = aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
  // presets without scores: the primary choice (ALU: the mt19937_64 seeding chain) needs row_off and the pair bits only,
  // so it runs on the second stream beside the emit pass of k_pair and leaves its choice for k_rows
  const bool split_primary = !aux_cols;
  if (split_primary) {
    RC(ensure_aux_stream(c));
    RC(c->pick.ensure((size_t)std::max<int64_t>(ng, 1) * 8)); P.pick = c->pick.as<uint64_t>();
    HIPCHK(hipEventRecord(c->aux_ev[0], st));
    HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->aux_ev[0], 0));
    RC(pf.begin(BR_K_PRIMARY, c->ksw_stream));
    launch_primary(c->ksw_stream, P, b->name_off, (b->names && b->name_off) ? b->names : nullptr, false);  // + per-group counters
    RC(pf.end());
    HIPCHK(hipEventRecord(c->aux_ev[1], c->ksw_stream));
  }
  if (n_rows) {
    RC(pf.begin(BR_K_PAIR_EMIT));
    launch_pair(st, P, true);
    RC(pf.end());
  }
  if (split_primary) HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));
  else {
    RC(pf.begin(BR_K_PRIMARY));
    launch_primary(st, P, b->name_off, (b->names && b->name_off) ? b->names : nullptr, aux_cols);  // + per-group counters
    RC(pf.end());
  }
  if (n_rows) {
    RC(pf.begin(BR_K_ROWS));
    launch_rows(st, P, aux_cols);
    RC(pf.end());
  }
  HIPCHK(hipMemcpyAsync(c->h_totals + 4, c->counters_d.p, 4 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!keep_events) RC(pf.collect());
  if (c->h_totals[7]) return BR_ERR_UNSUPPORTED;  // a rewritten CIGAR with more than 2^24 - 1 ops
  out->total_complete = n_rows; out->total_unique = c->h_totals[5]; out->dropped_reads = c->h_totals[6];
  if (fa_mode && c->rescue_stats[0]) { c->rescue_stats[1] = c->h_totals[8]; c->rescue_stats[2] = c->h_totals[9]; }

  // what a later large batch is predicted from (run_device_small, big)
  c->hist_n = n; c->hist[0] = n_matches; c->hist[1] = n_cig_arena; c->hist[2] = n_simple >= 0 ? (uint64_t)n_simple : 0; c->hist[3] = n_rows;
  c->hist_simf = dc.filter_by_similarity != 0;
  if (fa_mode) c->hist_n = 0;
  out->a = (const br_row_a *)c->pk_a.p; out->cigar = (const uint64_t *)c->pk_c.p; out->x = nullptr;   // br_device_rows_detail
  out->similarity_score = aux_cols ? c->pk_sim.as<double>() : nullptr;
  out->clip_score = aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
  out->pool = c->cig_arena.as<uint32_t>(); out->row_off = c->row_off.as<uint64_t>();
  c->counters[6] = n_matches;
  c->last_n_rows = (int64_t)n_rows; c->last_n_aln = n; c->last_n_pool = (int64_t)n_cig_arena;
  c->last_aux_cols = aux_cols; c->wide_valid = false; c->detail_valid = false; c->last_l_qseq = b->l_qseq; c->last_long_reads = dc.long_reads;
  return BR_OK;
}

// br_row_x of the last call's rows, derived on first request (k_rows_detail)
static int ensure_detail(br_ctx *c, hipStream_t st) {
  if (c->detail_valid) return BR_OK;
  const size_t nr = (size_t)std::max<int64_t>(c->last_n_rows, 1);
  if (c->rows_busy_set && c->pk_x.cap < nr * sizeof(uint4)) HIPCHK(hipEventSynchronize(c->rows_busy));   // a download may still read it
  RC(c->pk_x.ensure(nr * sizeof(uint4)));
  if (c->last_direct) {
    // direct rows keep no match table to gather from: the emit kernels run once more and write the detail column next to
    // the rows (the last call's batch and the context's tables are still in place: nothing has run since)
    if (c->last_n_rows > 0) {
      DirectArgs D = c->dD;
      D.r_x = c->pk_x.as<uint4>();
      if (c->d_split) { launch_emit_rows(st, c->dA, D, c->d_kept, c->d_simple, 1); launch_emit_rows(st, c->dA, D, c->d_kept, c->d_simple, 2); }
      else launch_emit_rows(st, c->dA, D, c->d_kept, c->d_simple, 0);
      launch_big_emit(st, c->dA, D, c->n_cu * 4);
    }
    c->detail_valid = true;
    return BR_OK;
  }
  if (c->last_n_rows > 0) {
    PairArgs P{};
    P.n_rows_total = c->last_n_rows; P.r_rec = c->r_rec.as<uint4>(); P.m_x = c->m_x.as<uint2>(); P.r_x = c->pk_x.as<uint4>();
    launch_rows_detail(st, P);
  }
  c->detail_valid = true;
  return BR_OK;
}

extern "C" int br_device_rows_detail(br_ctx *c, void *stream, const br_row_x **x) {
  if (!c || !x) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  RC(ensure_detail(c, (hipStream_t)stream));
  *x = (const br_row_x *)c->pk_x.p;
  return BR_OK;
}

// The wide view of the last call's rows: one device array per field (what tests, debuggers and the host-row entry
// points read).  Everything is derived from the packed table; nothing here is on the projection's own path.
static int expand_rows(br_ctx *c, hipStream_t st, br_device_wide_rows *out) {
  memset(out, 0, sizeof(*out));
  HIPCHK(hipSetDevice(c->ix->device));
  const uint64_t n_rows = (uint64_t)c->last_n_rows;
  const size_t nr = (size_t)std::max<uint64_t>(n_rows, 1);
  RC(c->r_input.ensure(nr * 4)); RC(c->r_nh.ensure(nr * 4)); RC(c->r_hi.ensure(nr * 4));
  RC(c->r_mapq.ensure(nr * 4)); RC(c->r_group.ensure(nr * 4));
  RC(c->r_mate_tid.ensure(nr * 4)); RC(c->r_mate_pos.ensure(nr * 4)); RC(c->r_isize.ensure(nr * 4));
  RC(c->r_tid.ensure(nr * 4)); RC(c->r_pos.ensure(nr * 4)); RC(c->r_ncig.ensure(nr * 4)); RC(c->r_strand.ensure(nr));
  RC(c->r_sim.ensure(nr * 8)); RC(c->r_clip.ensure(nr * 4)); RC(c->r_junc.ensure(nr * 4)); RC(c->r_refc.ensure(nr * 4));
  RC(c->r_cigoff.ensure((nr + 1) * 8));
  RC(c->r_paired.ensure(nr)); RC(c->r_same.ensure(nr)); RC(c->r_first.ensure(nr)); RC(c->r_primary.ensure(nr));
  uint64_t n_out_words = 0;
  if (n_rows) {
    WideArgs W{};
    W.n_rows = (int64_t)n_rows; W.n_aln = c->last_n_aln; W.long_reads = c->last_long_reads;
    RC(ensure_detail(c, st));
    W.r_a = c->pk_a.as<uint4>(); W.r_c = c->pk_c.as<uint2>(); W.r_x = c->pk_x.as<uint4>();
    W.r_sim = c->last_aux_cols ? c->pk_sim.as<double>() : nullptr; W.r_clip = c->last_aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
    W.pool = c->cig_arena.as<uint32_t>(); W.aln_group = c->aln_group.as<uint32_t>(); W.l_qseq = c->last_l_qseq;
    W.w_input = c->r_input.as<int32_t>(); W.w_nh = c->r_nh.as<uint32_t>(); W.w_hi = c->r_hi.as<uint32_t>();
    W.w_mapq = c->r_mapq.as<uint32_t>(); W.w_group = c->r_group.as<uint32_t>(); W.w_mate_tid = c->r_mate_tid.as<int32_t>();
    W.w_mate_pos = c->r_mate_pos.as<int32_t>(); W.w_isize = c->r_isize.as<int32_t>(); W.w_tid = c->r_tid.as<uint32_t>();
    W.w_pos = c->r_pos.as<uint32_t>(); W.w_ncig = c->r_ncig.as<uint32_t>(); W.w_strand = c->r_strand.as<int8_t>();
    W.w_sim = c->r_sim.as<double>(); W.w_clip = c->r_clip.as<int32_t>(); W.w_junc = c->r_junc.as<int32_t>();
    W.w_refc = c->r_refc.as<int32_t>(); W.w_paired = c->r_paired.as<uint8_t>(); W.w_same = c->r_same.as<uint8_t>();
    W.w_first = c->r_first.as<uint8_t>(); W.w_primary = c->r_primary.as<uint8_t>();
    launch_wide_fields(st, W);
    ScanArgs S3{};
    RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for((int64_t)n_rows + 1), 1) * 8 * 3));
    RC(c->totals.ensure(16 * 8));
    S3.n = (int64_t)n_rows; S3.src32 = c->r_ncig.as<uint32_t>(); S3.tile_sums = c->tile_sums.as<uint64_t>();
    uint64_t *d_tot = c->totals.as<uint64_t>();
    launch_scan(st, S3, 2, c->r_cigoff.p, true, d_tot + 8);
    HIPCHK(hipMemcpyAsync(c->h_totals + 12, d_tot + 8, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    n_out_words = c->h_totals[12];
    RC(c->cigar_out.ensure((size_t)std::max<uint64_t>(n_out_words, 1) * 4));
    W.w_cigoff = c->r_cigoff.as<uint64_t>(); W.w_cigar = c->cigar_out.as<uint32_t>();
    launch_wide_cigars(st, W, (int64_t)n_out_words);
  } else {
    HIPCHK(hipMemsetAsync(c->r_cigoff.p, 0, 8, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  c->wide_valid = true;
  out->n_rows = (int64_t)n_rows; out->n_cigar_words = (int64_t)n_out_words;
  out->input_index = c->r_input.as<int32_t>(); out->transcript_id = c->r_tid.as<uint32_t>();
  out->pos = c->r_pos.as<uint32_t>(); out->strand = c->r_strand.as<int8_t>();
  out->cigar_off = c->r_cigoff.as<uint64_t>(); out->cigar = c->cigar_out.as<uint32_t>();
  out->similarity_score = c->r_sim.as<double>(); out->clip_score = c->r_clip.as<int32_t>();
  out->junc_hits = c->r_junc.as<int32_t>(); out->aligned_len = c->r_refc.as<int32_t>();
  out->nh = c->r_nh.as<uint32_t>(); out->hi = c->r_hi.as<uint32_t>(); out->mapq = c->r_mapq.as<uint32_t>();
  out->is_paired = c->r_paired.as<uint8_t>(); out->same_transcript_as_mate = c->r_same.as<uint8_t>();
  out->is_first = c->r_first.as<uint8_t>();
  out->mate_transcript_id = c->r_mate_tid.as<int32_t>(); out->mate_pos = c->r_mate_pos.as<int32_t>();
  out->insert_size = c->r_isize.as<int32_t>(); out->group = c->r_group.as<uint32_t>();
  out->is_primary = c->r_primary.as<uint8_t>();
  return BR_OK;
}

}  // namespace

// Exact counters of the algorithmic-bytes formula for the batch the context
// projected last (its exon / match tables are still resident).
extern "C" int br_ctx_collect_counters(br_ctx *c, const br_device_batch *b, void *stream) {
  if (!c || !b) return BR_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->ix->device));
  if (c->last_direct) {
    // The formula's B_out counts the rewritten CIGAR words of every MATCH (the evaluator's output, before pairing), and
    // only the match table holds those: this diagnostic projects the batch once more through the match-table path (never timed).
    br_config cfgc; memset(&cfgc, 0, sizeof(cfgc)); cfgc.junc_miss_discount = 1.0;
    const DevCfg &d = c->dA.cfg;
    cfgc.lr = d.long_reads; cfgc.fr = d.fr; cfgc.rf = d.rf;
    cfgc.has_max_clip = 1; cfgc.max_clip = d.max_clip; cfgc.has_max_junc_ins = 1; cfgc.max_junc_ins = d.max_junc_ins;
    cfgc.has_max_junc_gap = 1; cfgc.max_junc_gap = d.max_junc_gap; cfgc.has_max_error_exon = 1; cfgc.max_error_exon = d.max_error_exon;
    cfgc.has_sim_thr = 1; cfgc.sim_thr = 1.0f;
    const int keep_direct = c->direct_rows, keep_small = c->small_batch;
    c->direct_rows = 0; c->small_batch = 0;
    br_device_rows tmp;
    const int rc = run_device(c, &cfgc, b, st, &tmp);
    c->direct_rows = keep_direct; c->small_batch = keep_small;
    if (rc) return rc;
  }
  RC(c->totals.ensure(16 * 8));
  DevBuf stats; RC(stats.ensure(8 * 8));
  HIPCHK(hipMemsetAsync(stats.p, 0, 8 * 8, st));
  StatsArgs T{};
  T.ix = c->ix->dev; T.n_aln = b->n_aln; T.ref_id = b->ref_id; T.cigar_off = b->cigar_off;
  T.seg = c->seg.as<uint2>(); T.head = c->head.as<uint4>(); T.head2 = c->head2.as<uint4>(); T.out = stats.as<uint64_t>();
  int64_t nm = (int64_t)c->counters[6];
  launch_stats(st, T, nm ? c->m_p.as<uint2>() : nullptr, c->match_off.as<uint32_t>(), c->n_matches.as<uint32_t>());
  uint64_t h[8];
  HIPCHK(hipMemcpyAsync(h, stats.p, 8 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  stats.release();
  uint64_t n = (uint64_t)b->n_aln;
  c->counters[3] = h[3]; c->counters[4] = h[4]; c->counters[5] = h[5]; c->counters[7] = h[7];
  c->counters[0] = 24ull * n + 4ull * h[3];
  c->counters[1] = h[1];
  c->counters[2] = 4ull * n + 24ull * (uint64_t)nm + 4ull * h[7];
  return BR_OK;
}

// aux_done: k_bam_scan already ran over these records with the same configuration
// (br_project_bam_device); keep_events: append to the running event list instead of restarting it
static int bam_encode_impl(br_ctx *c, const br_config *cfg, const br_device_records *recs, hipStream_t st,
                           br_device_bam *out, bool aux_done, bool keep_events) {
  memset(out, 0, sizeof(*out));
  if (recs->n_aln != c->last_n_aln) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  int64_t nr = c->last_n_rows, n = recs->n_aln;
  out->n_rows = nr;
  Prof pf{c, st};
  if (!keep_events) c->events_used = 0;
  BamArgs B{};
  B.n_aln = n; B.n_rows = nr; B.long_reads = (cfg->lr || cfg->lr_hq) ? 1 : 0;
  B.blob = recs->blob; B.rec_off = recs->rec_off; B.rec_len = recs->rec_len;
  RC(c->bam_aux.ensure(std::max<size_t>((size_t)n, 1) * sizeof(BamAux)));
  RC(c->bam_base.ensure(std::max<size_t>((size_t)n, 1) * 4));
  RC(c->bam_len.ensure(std::max<size_t>((size_t)nr, 1) * 4)); RC(c->bam_off.ensure(((size_t)nr + 1) * 8));
  B.aux = (BamAux *)c->bam_aux.p; B.base_len = c->bam_base.as<uint32_t>();
  RC(c->bam_end.ensure(BLOB_END_SLOTS * BLOB_END_STRIDE * 8)); B.blob_end = c->bam_end.as<uint64_t>();
  B.r_a = c->pk_a.as<uint4>(); B.r_c = c->pk_c.as<uint2>(); B.r_rec = c->r_rec.as<uint4>();
  if (c->last_direct) {   // no r_rec on the direct path: the detail column carries the input alignment and HI
    RC(ensure_detail(c, st));
    B.r_rec = c->pk_x.as<uint4>(); B.rec_x = 1;
  }
  B.r_sim = c->last_aux_cols ? c->pk_sim.as<double>() : nullptr; B.r_clip = c->last_aux_cols ? c->pk_clip.as<int32_t>() : nullptr;
  B.pool = c->cig_arena.as<uint32_t>(); B.l_qseq = c->last_l_qseq;
  B.out_len = c->bam_len.as<uint32_t>(); B.out_off = c->bam_off.as<uint64_t>();
  RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for(nr + 1), 1) * 8 * 3));
  RC(c->totals.ensure(16 * 8));
  B.too_long = c->totals.as<uint64_t>() + 6;
  HIPCHK(hipMemsetAsync(B.too_long, 0, 8, st));
  RC(pf.begin(BR_K_BAM));
  if (!aux_done) { HIPCHK(hipMemsetAsync(B.blob_end, 0, BLOB_END_SLOTS * BLOB_END_STRIDE * 8, st)); launch_bam_scan(st, B); }
  launch_bam_size(st, B);
  RC(pf.end());
  ScanArgs S{}; S.n = nr; S.src32 = B.out_len; S.tile_sums = c->tile_sums.as<uint64_t>();
  RC(pf.begin(BR_K_SCAN));
  launch_scan(st, S, 2, c->bam_off.p, true, c->totals.as<uint64_t>() + 7);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 10, c->totals.as<uint64_t>() + 6, 16, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // (h_totals[10], "a spilled CIGAR spans 2^28 reference bases or more", is set by the encoder: checked after it)
  uint64_t total = nr ? c->h_totals[11] : 0;
  RC(c->bam_out.ensure(std::max<size_t>(total, 16)));
  B.out = c->bam_out.as<uint8_t>();
  RC(pf.begin(BR_K_BAM));
  launch_bam_encode(st, B, c->bam_lanes);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 10, c->totals.as<uint64_t>() + 6, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (nr && c->h_totals[10]) { pf.collect(); return BR_ERR_UNSUPPORTED; }  // bam_write1 refuses such a record too
  RC(pf.collect());
  out->data = c->bam_out.as<uint8_t>(); out->n_bytes = total; out->row_off = c->bam_off.as<uint64_t>();
  return BR_OK;
}

extern "C" int br_bam_encode_device(br_ctx *c, const br_config *cfg, const br_device_records *recs, void *stream,
                                    br_device_bam *out) {
  if (!c || !cfg || !recs || !out) return BR_ERR_INVALID_ARG;
  return bam_encode_impl(c, cfg, recs, (hipStream_t)stream, out, false, false);
}

// ---------------------------------------------------------------------------
// BAM bundle entry: records -> input tables -> projection -> records
// ---------------------------------------------------------------------------
extern "C" int br_project_bam_device(br_ctx *c, const br_config *cfg, const br_device_records *recs,
                                     const int32_t *ref_map, int32_t n_ref_map, void *stream,
                                     br_device_rows *rows_out, br_device_bam *out) {
  if (!c || !cfg || !recs || !out || n_ref_map < 0 || (n_ref_map && !ref_map)) return BR_ERR_INVALID_ARG;
  if (recs->n_aln && (!recs->blob || !recs->rec_off)) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  br_device_rows local_rows;
  br_device_rows *rows = rows_out ? rows_out : &local_rows;
  memset(rows, 0, sizeof(*rows));
  hipStream_t st = (hipStream_t)stream;
  const br_index *ix = c->ix;
  DevCfg dc;
  RC(make_devcfg(cfg, dc));
  const bool fa_mode = dc.use_fasta && dc.long_reads;
  int64_t n = recs->n_aln;
  if (n < 0 || n >= 0x7fffffffll) return BR_ERR_CAPACITY;
  HIPCHK(hipSetDevice(ix->device));
  c->last_n_aln = n; c->last_n_rows = 0;
  c->events_used = 0;
  Prof pf{c, st};
  if (n == 0) { pf.collect(); return BR_OK; }

  size_t nn = (size_t)n;
  RC(c->b_ref_id.ensure(nn * 4)); RC(c->b_ref_start.ensure(nn * 4)); RC(c->b_flags.ensure(nn * 2));
  RC(c->b_xs.ensure(nn)); RC(c->b_ts.ensure(nn)); RC(c->b_lqseq.ensure(nn * 4));
  RC(c->b_cigar_off.ensure((nn + 1) * 4)); RC(c->b_name_off.ensure((nn + 1) * 4)); RC(c->b_mate_idx.ensure(nn * 4));
  RC(c->p_ncig.ensure(nn * 4)); RC(c->p_name_len.ensure(nn * 4)); RC(c->p_isnew.ensure(nn * 4));
  RC(c->p_group_pre.ensure((nn + 1) * 4)); RC(c->p_small.ensure(64)); RC(c->p_big.ensure((nn / 96 + 2) * 4));
  RC(c->p_ref_map.ensure(std::max<size_t>((size_t)n_ref_map, 1) * 4));
  RC(c->bam_aux.ensure(nn * sizeof(BamAux)));
  RC(c->bam_base.ensure(nn * 4));
  RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for(n + 1), 1) * 8 * 3));
  RC(c->totals.ensure(16 * 8));
  if (n_ref_map) HIPCHK(hipMemcpyAsync(c->p_ref_map.p, ref_map, (size_t)n_ref_map * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(c->p_small.p, 0, 64, st));  // [0] max n_cigar, [1] max soft clip, [2] big-group count

  ParseArgs P{};
  P.n = n; P.blob = recs->blob; P.rec_off = recs->rec_off; P.rec_len = recs->rec_len;
  P.ref_map = c->p_ref_map.as<int32_t>(); P.n_ref_map = n_ref_map;
  P.ref_id = c->b_ref_id.as<int32_t>(); P.ref_start = c->b_ref_start.as<int32_t>(); P.l_qseq = c->b_lqseq.as<int32_t>();
  P.flags = c->b_flags.as<uint16_t>(); P.ncig = c->p_ncig.as<uint32_t>(); P.name_len = c->p_name_len.as<uint32_t>();
  P.isnew = c->p_isnew.as<uint32_t>(); P.maxima = c->p_small.as<uint32_t>(); P.n_big_groups = c->p_small.as<uint32_t>() + 2;
  P.big_groups = c->p_big.as<uint32_t>(); P.group_pre = c->p_group_pre.as<uint32_t>();
  P.cigar_off = c->b_cigar_off.as<uint32_t>(); P.name_off = c->b_name_off.as<uint32_t>(); P.mate_idx = c->b_mate_idx.as<int32_t>();

  BamArgs B{};
  B.n_aln = n; B.long_reads = dc.long_reads ? 1 : 0; B.blob = recs->blob; B.rec_off = recs->rec_off; B.rec_len = recs->rec_len;
  B.aux = (BamAux *)c->bam_aux.p; B.base_len = c->bam_base.as<uint32_t>(); B.xs_out = c->b_xs.as<int8_t>(); B.ts_out = c->b_ts.as<int8_t>();
  RC(c->bam_end.ensure(BLOB_END_SLOTS * BLOB_END_STRIDE * 8)); B.blob_end = c->bam_end.as<uint64_t>();
  HIPCHK(hipMemsetAsync(B.blob_end, 0, BLOB_END_SLOTS * BLOB_END_STRIDE * 8, st));

  // the aux walk of the records (one lane per record, latency-bound) on the second stream beside the reader side
  // (k_rec_fields .. k_mates, the same kind of kernel over the same records): joined below, in front of the projection
  RC(ensure_aux_stream(c));
  HIPCHK(hipEventRecord(c->aux_ev[0], st));
  HIPCHK(hipStreamWaitEvent(c->ksw_stream, c->aux_ev[0], 0));
  RC(pf.begin(BR_K_BAM, c->ksw_stream));
  launch_bam_scan(c->ksw_stream, B);
  RC(pf.end());
  HIPCHK(hipEventRecord(c->aux_ev[1], c->ksw_stream));
  RC(pf.begin(BR_K_PARSE));
  launch_rec_fields(st, P);
  RC(pf.end());
  uint64_t *d_tot = c->totals.as<uint64_t>();
  ScanArgs S{}; S.n = n; S.tile_sums = c->tile_sums.as<uint64_t>();
  RC(pf.begin(BR_K_SCAN));
  S.src32 = P.ncig;     launch_scan(st, S, 2, c->b_cigar_off.p, false, d_tot + 0);
  S.src32 = P.name_len; launch_scan(st, S, 2, c->b_name_off.p, false, d_tot + 1);
  S.src32 = P.isnew;    launch_scan(st, S, 2, c->p_group_pre.p, false, d_tot + 2);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 16, d_tot, 3 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(c->h_totals + 20, c->p_small.p, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  uint64_t n_words = c->h_totals[16], name_bytes = c->h_totals[17], ng = c->h_totals[18];
  uint32_t max_nc = (uint32_t)(c->h_totals[20] & 0xffffffffu), max_clip = (uint32_t)(c->h_totals[20] >> 32);
  if (n_words >= 0xffffffffull - (uint64_t)n || name_bytes >= 0xfffffff0ull) { (void)hipStreamSynchronize(c->ksw_stream); pf.collect(); return BR_ERR_CAPACITY; }
  RC(c->b_cigar.ensure(std::max<size_t>((size_t)n_words, 1) * 4)); RC(c->b_names.ensure(std::max<size_t>((size_t)name_bytes, 1)));
  RC(c->b_group_off.ensure(((size_t)ng + 1) * 4));
  P.n_groups = (int64_t)ng; P.group_off = c->b_group_off.as<uint32_t>();
  P.cigar = c->b_cigar.as<uint32_t>(); P.names = c->b_names.as<uint8_t>();
  RC(pf.begin(BR_K_PARSE));
  launch_group_off(st, P);
  launch_rec_copy(st, P);
  launch_mates(st, P);
  RC(pf.end());
  HIPCHK(hipStreamWaitEvent(st, c->aux_ev[1], 0));   // k_bam_scan: XS / ts characters, the aux table

  br_device_batch db{};
  if (fa_mode) {
    RC(c->b_seq_src.ensure(nn * 4)); RC(c->p_seq_len.ensure(nn * 4)); RC(c->b_seq_off.ensure((nn + 1) * 4));
    P.seq_src = c->b_seq_src.as<int32_t>(); P.seq_len = c->p_seq_len.as<uint32_t>(); P.seq_off = c->b_seq_off.as<uint32_t>();
    RC(pf.begin(BR_K_PARSE));
    launch_seq_src(st, P);
    RC(pf.end());
    RC(pf.begin(BR_K_SCAN));
    S.src32 = P.seq_len; launch_scan(st, S, 2, c->b_seq_off.p, false, d_tot + 3);
    RC(pf.end());
    HIPCHK(hipMemcpyAsync(c->h_totals + 19, d_tot + 3, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint64_t sbytes = c->h_totals[19];
    if (sbytes >= 0xfffffff0ull) { pf.collect(); return BR_ERR_CAPACITY; }
    RC(c->b_seqs.ensure(std::max<size_t>((size_t)sbytes, 1)));
    P.seqs = c->b_seqs.as<uint8_t>();
    RC(pf.begin(BR_K_PARSE));
    launch_seq_ascii(st, P);
    RC(pf.end());
    db.seq_off = P.seq_off; db.seqs = P.seqs; db.seq_src = P.seq_src; db.max_soft_clip = (int32_t)max_clip;
  }
  db.n_aln = n; db.n_groups = (int64_t)ng; db.ref_id = P.ref_id; db.ref_start = P.ref_start; db.flags = P.flags;
  db.xs = c->b_xs.as<int8_t>(); db.ts = c->b_ts.as<int8_t>(); db.cigar_off = P.cigar_off; db.cigar = P.cigar;
  db.mate_idx = P.mate_idx; db.group_off = P.group_off; db.l_qseq = P.l_qseq;
  db.n_cigar_words = (int64_t)n_words; db.max_n_cigar = (int32_t)max_nc;
  db.name_off = P.name_off; db.names = P.names;
  { WantDetail wd(c, true); RC(run_device_impl(c, cfg, &db, st, rows, true)); }   // the encoder reads input alignment and HI of every row
  RC(bam_encode_impl(c, cfg, recs, st, out, true, true));
  return BR_OK;
}

// ---------------------------------------------------------------------------
// BGZF deflate on the device
// ---------------------------------------------------------------------------
static int deflate_device_impl(br_ctx *c, const uint8_t *src, uint64_t n, hipStream_t st, const uint8_t **out, uint64_t *out_bytes,
                               bool keep_events) {
  *out = nullptr; *out_bytes = 0;
  Prof pf{c, st};
  if (!keep_events) c->events_used = 0;
  if (n == 0) { if (!keep_events) pf.collect(); return BR_OK; }
  if (!c->z_tabs_ready) {
    // CRC-32 (reflected 0xEDB88320) byte table and the operator that appends DEFLATE_CRC_CHUNK zero bytes to a
    // register (zlib's crc32_combine does the same with squared matrices; here the length is fixed)
    std::vector<uint32_t> t(256 + 1024);
    for (uint32_t i = 0; i < 256; i++) { uint32_t v = i; for (int k = 0; k < 8; k++) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1; t[i] = v; }
    uint32_t col[32];
    for (int b = 0; b < 32; b++) { uint32_t v = 1u << b; for (uint32_t k = 0; k < DEFLATE_CRC_CHUNK; k++) v = (v >> 8) ^ t[v & 0xffu]; col[b] = v; }
    for (int byte = 0; byte < 4; byte++)
      for (uint32_t x = 0; x < 256; x++) { uint32_t v = 0; for (int b = 0; b < 8; b++) if (x & (1u << b)) v ^= col[8 * byte + b]; t[256 + 256 * byte + x] = v; }
    RC(c->z_tabs.ensure(t.size() * 4));
    HIPCHK(hipMemcpyAsync(c->z_tabs.p, t.data(), t.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    c->z_tabs_ready = true;
  }
  uint64_t nb = (n + DEFLATE_PAYLOAD - 1) / DEFLATE_PAYLOAD;
  RC(c->z_slots.ensure((size_t)nb * DEFLATE_SLOT)); RC(c->z_sizes.ensure((size_t)nb * 4)); RC(c->z_off.ensure(((size_t)nb + 1) * 8));
  RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for((int64_t)nb + 1), 1) * 8 * 3));
  RC(c->totals.ensure(16 * 8));
  DeflateArgs A{};
  A.src = src; A.n_bytes = n; A.n_blocks = nb; A.slots = c->z_slots.as<uint8_t>(); A.sizes = c->z_sizes.as<uint32_t>();
  A.crc_tab = c->z_tabs.as<uint32_t>(); A.crc_shift = c->z_tabs.as<uint32_t>() + 256;
  int dyn_waves = 0;
  if (c->deflate_dynamic) {  // persistent waves: as many as the chip holds (6 workgroups of 4 waves per CU by their LDS), a token list each
    uint64_t want = (uint64_t)c->n_cu * 24;
    dyn_waves = (int)std::min<uint64_t>((nb + 3) / 4 * 4, want / 4 * 4);
    if (dyn_waves < 4) dyn_waves = 4;
    RC(c->z_tokens.ensure((size_t)dyn_waves * DEFLATE_PAYLOAD * 4 + 64));
#ifdef DEFLATE_PROFILE
    HIPCHK(hipMemsetAsync(c->z_tokens.as<uint8_t>() + (size_t)dyn_waves * DEFLATE_PAYLOAD * 4, 0, 64, st));
#endif
    A.tokens = c->z_tokens.as<uint32_t>();
    A.queue = (uint32_t *)(c->totals.as<uint64_t>() + 15);
    HIPCHK(hipMemsetAsync(A.queue, 0, 8, st));
  }
  RC(pf.begin(BR_K_CODEC));
  launch_deflate(st, A, dyn_waves);
  RC(pf.end());
#ifdef DEFLATE_PROFILE
  if (dyn_waves) {
    uint64_t pt[8];
    HIPCHK(hipMemcpyAsync(pt, c->z_tokens.as<uint8_t>() + (size_t)dyn_waves * DEFLATE_PAYLOAD * 4, 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double tot = 0; for (int k = 0; k < 8; k++) tot += (double)pt[k];
    static const char *nm[8] = {"clear", "parse_step", "tokens+hist", "code build", "header", "replay", "crc+frame", "claim"};
    for (int k = 0; k < 8; k++) fprintf(stderr, "[deflate profile] %-12s %5.1f %%\n", nm[k], 100.0 * (double)pt[k] / tot);
  }
#endif
  ScanArgs S{}; S.n = (int64_t)nb; S.src32 = A.sizes; S.tile_sums = c->tile_sums.as<uint64_t>();
  RC(pf.begin(BR_K_SCAN));
  launch_scan(st, S, 2, c->z_off.p, true, c->totals.as<uint64_t>() + 5);
  RC(pf.end());
  HIPCHK(hipMemcpyAsync(c->h_totals + 24, c->totals.as<uint64_t>() + 5, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  uint64_t total = c->h_totals[24];
  DevBuf &dense = c->z_dense_which ? c->z_dense_alt : c->z_dense;
  RC(dense.ensure((size_t)total + 16));
  RC(pf.begin(BR_K_CODEC));
  launch_bgzf_compact(st, A, c->z_off.as<uint64_t>(), dense.as<uint8_t>());
  RC(pf.end());
  if (!keep_events) { HIPCHK(hipStreamSynchronize(st)); RC(pf.collect()); }
  *out = dense.as<uint8_t>(); *out_bytes = total;
  return BR_OK;
}

extern "C" int br_bgzf_deflate_device(br_ctx *c, const uint8_t *src, uint64_t n, void *stream, const uint8_t **out,
                                      uint64_t *out_bytes) {
  if (!c || (!src && n) || !out || !out_bytes) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  return deflate_device_impl(c, src, n, (hipStream_t)stream, out, out_bytes, false);
}

// ---------------------------------------------------------------------------
// BGZF inflate on the device
// ---------------------------------------------------------------------------
static_assert(sizeof(br_bgzf_block) == sizeof(InflateBlock), "br_bgzf_block is the kernel's block descriptor");

extern "C" int br_bgzf_scan(const uint8_t *data, uint64_t n_bytes, int64_t cap, br_bgzf_block *blocks, int64_t *n_blocks,
                            uint64_t *consumed, uint64_t *out_bytes) {
  if ((!data && n_bytes) || !blocks || !n_blocks || !consumed || !out_bytes || cap < 0) return BR_ERR_INVALID_ARG;
  uint64_t p = 0, total = 0; int64_t n = 0;
  while (n < cap && p + 18 <= n_bytes) {
    const uint8_t *h = data + p;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return BR_ERR_INVALID_ARG;   // not a BGZF block
    const uint32_t xlen = h[10] | ((uint32_t)h[11] << 8);
    if (p + 12 + xlen > n_bytes) break;
    int64_t bsize = -1;
    for (uint32_t q = 0; q + 4 <= xlen;) {
      const uint8_t *x = h + 12 + q;
      const uint32_t slen = x[2] | ((uint32_t)x[3] << 8);
      if (x[0] == 'B' && x[1] == 'C' && slen == 2 && q + 6 <= xlen) bsize = (x[4] | (x[5] << 8)) + 1;
      q += 4 + slen;
    }
    if (bsize < (int64_t)(12 + xlen + 8)) return BR_ERR_INVALID_ARG;            // no BC subfield
    if (p + (uint64_t)bsize > n_bytes) break;                                     // partial block: next call
    const uint8_t *t = h + bsize - 8;
    const uint32_t crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24), ulen = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
    if (ulen > 65536) return BR_ERR_INVALID_ARG;
    if (ulen) {   // (empty blocks -- the EOF marker -- are stepped over)
      br_bgzf_block &b = blocks[n++];
      b.src_off = p + 12 + xlen; b.dst_off = total; b.clen = (uint32_t)(bsize - 12 - xlen - 8); b.ulen = ulen; b.crc = crc; b.pad = 0;
      total += ulen;
    }
    p += (uint64_t)bsize;
  }
  *n_blocks = n; *consumed = p; *out_bytes = total;
  return BR_OK;
}

// dst_ext: where the inflated bytes go (room for every block's dst_off + ulen), or null: the context's own buffer
static int inflate_impl(br_ctx *c, const uint8_t *src, uint64_t n_src, const br_bgzf_block *blocks, int64_t n_blocks, hipStream_t st,
                        uint8_t *dst_ext, const uint8_t **out, uint64_t *out_bytes) {
  *out = nullptr; *out_bytes = 0;
  if (n_blocks == 0) return BR_OK;
  uint64_t total = 0;
  for (int64_t i = 0; i < n_blocks; i++) {
    const br_bgzf_block &b = blocks[i];
    if (b.ulen > 65536 || b.src_off + b.clen + 8 > n_src) return BR_ERR_INVALID_ARG;   // (+ 8: the block's CRC32 / ISIZE trailer lies inside the buffer)
    total = std::max<uint64_t>(total, b.dst_off + b.ulen);
  }
  if (!c->inf_tabs_ready) {
    // slice-by-4 tables of the reflected CRC-32 and the operator that appends INFLATE_CRC_CHUNK zero bytes (see deflate_device_impl)
    std::vector<uint32_t> t(1024 + 1024);
    for (uint32_t i = 0; i < 256; i++) { uint32_t v = i; for (int k = 0; k < 8; k++) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1; t[i] = v; }
    for (int k = 1; k < 4; k++) for (uint32_t i = 0; i < 256; i++) { const uint32_t p = t[256 * (k - 1) + i]; t[256 * k + i] = (p >> 8) ^ t[p & 0xffu]; }
    uint32_t col[32];
    for (int b = 0; b < 32; b++) { uint32_t v = 1u << b; for (uint32_t k = 0; k < INFLATE_CRC_CHUNK; k++) v = (v >> 8) ^ t[v & 0xffu]; col[b] = v; }
    for (int byte = 0; byte < 4; byte++)
      for (uint32_t x = 0; x < 256; x++) { uint32_t v = 0; for (int b = 0; b < 8; b++) if (x & (1u << b)) v ^= col[8 * byte + b]; t[1024 + 256 * byte + x] = v; }
    RC(c->inf_tabs.ensure(t.size() * 4));
    HIPCHK(hipMemcpyAsync(c->inf_tabs.p, t.data(), t.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    c->inf_tabs_ready = true;
  }
  if (!dst_ext) RC(c->inf_out.ensure((size_t)total + 16));
  RC(c->inf_blocks.ensure((size_t)n_blocks * sizeof(InflateBlock))); RC(c->inf_cnt.ensure(16));
  HIPCHK(hipMemcpyAsync(c->inf_blocks.p, blocks, (size_t)n_blocks * sizeof(InflateBlock), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(c->inf_cnt.p, 0, 16, st));
  InflateArgs A{};
  A.src = src; A.n_src = n_src; A.dst = dst_ext ? dst_ext : c->inf_out.as<uint8_t>(); A.blocks = (const InflateBlock *)c->inf_blocks.p; A.n_blocks = (uint64_t)n_blocks;
  A.queue = c->inf_cnt.as<uint32_t>(); A.n_bad = c->inf_cnt.as<uint32_t>() + 1;
  A.crc_tab4 = c->inf_tabs.as<uint32_t>(); A.crc_shift = c->inf_tabs.as<uint32_t>() + 1024;
  const int waves = (int)std::min<uint64_t>(((uint64_t)n_blocks + 3) / 4 * 4, (uint64_t)c->n_cu * 20);   // five workgroups of four waves per CU (their LDS and registers)
  Prof pf{c, st};
  c->events_used = 0;
  RC(pf.begin(BR_K_CODEC));
  launch_inflate(st, A, waves);
  RC(pf.end());
  uint32_t bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, A.n_bad, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  RC(pf.collect());
  if (bad) return BR_ERR_INVALID_ARG;   // a block that does not inflate to its ISIZE bytes with its CRC32
  *out = A.dst; *out_bytes = total;
  return BR_OK;
}

extern "C" int br_bgzf_inflate_device(br_ctx *c, const uint8_t *src, uint64_t n_src, const br_bgzf_block *blocks, int64_t n_blocks,
                                      void *stream, const uint8_t **out, uint64_t *out_bytes) {
  if (!c || (!src && n_src) || (!blocks && n_blocks) || n_blocks < 0 || !out || !out_bytes) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  return inflate_impl(c, src, n_src, blocks, n_blocks, (hipStream_t)stream, nullptr, out, out_bytes);
}

// br_bam_split on the device (split_kernels.hip): data = an inflated BAM alignment section in HBM that starts at a record
static int split_impl(br_ctx *c, const uint8_t *data, uint64_t n_bytes, int32_t n_ref, hipStream_t st, br_device_records *recs,
                      int64_t *n_unmapped, uint64_t *consumed, SplitArgs *S_out) {
  memset(recs, 0, sizeof(*recs));
  recs->blob = data; *consumed = 0;
  if (n_unmapped) *n_unmapped = 0;
  if (S_out) *S_out = SplitArgs{};
  if (n_bytes == 0) return BR_OK;
  const int64_t n_seg = (int64_t)((n_bytes + SPLIT_SEG_BYTES - 1) / SPLIT_SEG_BYTES);
  const size_t ns = (size_t)n_seg;
  RC(c->sp_entry.ensure(ns * 8)); RC(c->sp_entry2.ensure(ns * 8)); RC(c->sp_exit.ensure(ns * 8)); RC(c->sp_nmap.ensure(ns * 4));
  RC(c->sp_nunm.ensure(ns * 4)); RC(c->sp_ended.ensure(ns * 4)); RC(c->sp_redo.ensure(ns * 4)); RC(c->sp_pre.ensure((ns + 1) * 8));
  RC(c->sp_small.ensure(64));   // flags[2] (u32) | totals[2] (u64) at +16 | mapped total (u64) at +32
  RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for(n_seg + 1), 1) * 8 * 3));
  HIPCHK(hipMemsetAsync(c->sp_small.p, 0, 64, st));
  SplitArgs S{};
  S.data = data; S.n_bytes = n_bytes; S.n_ref = n_ref; S.seg_bytes = SPLIT_SEG_BYTES; S.n_seg = n_seg;
  S.entry = c->sp_entry.as<uint64_t>(); S.entry_next = c->sp_entry2.as<uint64_t>(); S.exit_ = c->sp_exit.as<uint64_t>();
  S.n_map = c->sp_nmap.as<uint32_t>(); S.n_unm = c->sp_nunm.as<uint32_t>(); S.ended = c->sp_ended.as<uint32_t>();
  S.flags = c->sp_small.as<uint32_t>(); S.totals = (uint64_t *)(c->sp_small.as<uint8_t>() + 16);
  uint32_t *redo = c->sp_redo.as<uint32_t>();
  launch_split_guess(st, S);
  if (c->split_spoil > 0) launch_split_spoil(st, S, c->split_spoil);   // test hook (br_ctx_set_param "split_spoil"): wrong guesses on purpose
  launch_split_walk(st, S, nullptr);
  for (int pass = 0;; pass++) {
    // every guess against where the chain of the segments in front arrives; the segments that were wrong walk again
    HIPCHK(hipMemsetAsync(S.flags + 1, 0, 4, st));
    launch_split_check(st, S, redo);
    uint32_t changed = 0;
    HIPCHK(hipMemcpyAsync(&changed, S.flags + 1, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::swap(S.entry, S.entry_next);
    static const bool split_debug = getenv("BRAMBLE_AMD_SPLIT_DEBUG") != nullptr;
    if (split_debug && (pass < 12 || !changed)) fprintf(stderr, "[split] pass %d: %u of %lld segments took another entry\n", pass, changed, (long long)n_seg);
    if (!changed) break;
    if (pass > n_seg + 2) return BR_ERR_INVALID_ARG;   // (cannot happen: every pass settles at least the first wrong segment)
    launch_split_walk(st, S, redo);
  }
  ScanArgs SC{}; SC.n = n_seg; SC.src32 = S.n_map; SC.tile_sums = c->tile_sums.as<uint64_t>();
  launch_scan(st, SC, 2, c->sp_pre.p, true, (uint64_t *)(c->sp_small.as<uint8_t>() + 32));
  uint64_t n_mapped = 0;
  HIPCHK(hipMemcpyAsync(&n_mapped, c->sp_small.as<uint8_t>() + 32, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  RC(c->sp_off.ensure(std::max<size_t>((size_t)n_mapped, 1) * 8)); RC(c->sp_len.ensure(std::max<size_t>((size_t)n_mapped, 1) * 4));
  S.map_pre = c->sp_pre.as<uint64_t>(); S.rec_off = c->sp_off.as<uint64_t>(); S.rec_len = c->sp_len.as<uint32_t>();
  launch_split_emit(st, S);
  launch_split_totals(st, S);
  struct { uint32_t flags[4]; uint64_t totals[2]; } h;
  HIPCHK(hipMemcpyAsync(&h, c->sp_small.p, 32, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (h.flags[0] & 1u) return BR_ERR_INVALID_ARG;      // a record whose fixed fields overrun its block_size (as br_bam_split)
  recs->rec_off = S.rec_off; recs->rec_len = S.rec_len; recs->n_aln = (int64_t)n_mapped;
  if (n_unmapped) *n_unmapped = (int64_t)h.totals[0];
  *consumed = h.totals[1];
  if (S_out) *S_out = S;
  return BR_OK;
}

extern "C" int br_bam_split_device(br_ctx *c, const uint8_t *data, uint64_t n_bytes, int32_t n_ref, void *stream, br_device_records *recs,
                                   int64_t *n_unmapped, uint64_t *consumed) {
  if (!c || (!data && n_bytes) || !recs || !consumed || n_ref < 0) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  return split_impl(c, data, n_bytes, n_ref, (hipStream_t)stream, recs, n_unmapped, consumed, nullptr);
}

// ---------------------------------------------------------------------------
// br_bam_reader: BGZF bytes of a BAM file in, bundles of device-resident records of whole read-name groups out.  What the host
// reader of the command line does with sixteen inflate threads and a serial chain walk (inflate -> br_bam_split -> cut at a
// read-name change -> upload), done where the records are needed: the compressed bytes go up as they are, k_inflate and
// k_split_* make records of them, and the bytes behind the last complete name group wait in HBM for the next piece.
// It needs no index (the command line runs it beside the guide parsing and the index build).
// ---------------------------------------------------------------------------
struct br_bam_reader {
  br_index shell;                  // carries the device for the private context below; never used for projection
  br_ctx *c = nullptr;
  int32_t n_ref = 0;
  uint64_t skip = 0;               // inflated bytes still to skip (the BAM header in front of the first record)
  int64_t max_blocks = 3072;       // BGZF blocks per piece (about 200 MB inflated)
  hipStream_t st = nullptr;
  struct Chunk { DevBuf data, off, len; int64_t id = -1; bool out = false; };   // out: handed to the caller, not yet released
  std::vector<std::unique_ptr<Chunk>> chunks;
  std::mutex m;
  Chunk *carry_from = nullptr; uint64_t carry_off = 0, carry_len = 0;
  DevBuf comp, small;
  // piece-wise reading (br_bam_piece_*): two upload slots, filled on a copy stream of their own beside the processing of the
  // piece before
  struct PieceSlot { DevBuf comp; hipEvent_t up = nullptr; int64_t b0 = -1, b1x = -1; uint64_t src0 = 0, n_src = 0; };
  PieceSlot pslot[2];
  hipStream_t copy_st = nullptr;
  // the way up: PIN_THREADS host threads copy the mapped file's bytes into pinned buffers of their own (two each) and
  // queue the transfers from there -- a transfer straight from the pageable mapping goes through the driver's one staging
  // thread at a fifth of the wire's rate
  static constexpr int PIN_THREADS = 4, PIN_SLOTS = 2;
  static constexpr size_t PIN_BYTES = 4u << 20;
  struct PinBuf { uint8_t *p = nullptr; hipEvent_t done = nullptr; bool used = false; };
  PinBuf pin[PIN_THREADS][PIN_SLOTS];
  std::mutex up_m;                 // one upload at a time (the pinned buffers; a piece that asks for more blocks uploads from the processing thread)
  double t_upload = 0;
  std::vector<br_bgzf_block> pblocks;
  double t_proc = 0;
  std::vector<br_bgzf_block> blocks;
  int64_t next_id = 0;
  bool finished = false;
  double t_scan = 0, t_up = 0, t_inflate = 0, t_split = 0, t_cut = 0;   // BRAMBLE_AMD_TIMING
};

extern "C" int br_bam_reader_new(int device, int32_t n_ref, uint64_t header_bytes, br_bam_reader **out) {
  if (!out || n_ref < 0) return BR_ERR_INVALID_ARG;
  *out = nullptr;
  int rc = check_device(device);
  if (rc) return rc;
  auto r = std::make_unique<br_bam_reader>();
  r->shell.device = device; r->n_ref = n_ref; r->skip = header_bytes;
  RC(br_ctx_new(&r->shell, &r->c));
  // the lowest priority the device offers: once the projection has started, its kernels go first (what the reader makes is
  // needed a few bundles later; what the runner makes is what the writer waits for)
  int prio_low = 0, prio_high = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  HIPCHK(hipStreamCreateWithPriority(&r->st, hipStreamNonBlocking, prio_low));
  *out = r.release();
  return BR_OK;
}

extern "C" int br_bam_reader_set_piece_blocks(br_bam_reader *r, int64_t blocks) {
  if (!r || blocks < 1 || blocks > (1 << 20)) return BR_ERR_INVALID_ARG;
  r->max_blocks = blocks;
  return BR_OK;
}

extern "C" void br_bam_reader_free(br_bam_reader *r) {
  if (!r) return;
  (void)hipSetDevice(r->shell.device);
  for (auto &ch : r->chunks) { ch->data.release(); ch->off.release(); ch->len.release(); }
  r->comp.release(); r->small.release();
  for (auto &ps : r->pslot) { ps.comp.release(); if (ps.up) (void)hipEventDestroy(ps.up); }
  for (auto &row : r->pin) for (auto &pb : row) { if (pb.used && pb.done) (void)hipEventSynchronize(pb.done); if (pb.p) (void)hipHostFree(pb.p); if (pb.done) (void)hipEventDestroy(pb.done); }
  if (r->copy_st) (void)hipStreamDestroy(r->copy_st);
  if (r->st) (void)hipStreamDestroy(r->st);
  if (r->c) br_ctx_free(r->c);
  delete r;
}

extern "C" int br_bam_reader_release(br_bam_reader *r, int64_t id) {
  if (!r) return BR_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> l(r->m);
  for (auto &ch : r->chunks) if (ch->id == id) { ch->out = false; return BR_OK; }
  return BR_ERR_INVALID_ARG;
}

extern "C" int br_bam_reader_next(br_bam_reader *r, const uint8_t *data, uint64_t n_bytes, int last, uint64_t *consumed,
                                  br_device_records *bundle, int64_t *id, int64_t *n_unmapped) {
  if (!r || (!data && n_bytes) || !consumed || !bundle || !id || !n_unmapped) return BR_ERR_INVALID_ARG;
  memset(bundle, 0, sizeof(*bundle));
  *consumed = 0; *id = -1; *n_unmapped = 0;
  if (r->finished) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(r->shell.device));
  hipStream_t st = r->st;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tsec = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  auto tp = tnow();
  // the complete blocks of this piece
  r->blocks.resize((size_t)r->max_blocks);
  int64_t nb = 0; uint64_t used = 0, total = 0;
  RC(br_bgzf_scan(data, n_bytes, r->max_blocks, r->blocks.data(), &nb, &used, &total));
  r->t_scan += tsec(tp); tp = tnow();
  *consumed = used;
  const bool at_end = last && used == n_bytes;       // nothing of the file is left behind this piece
  if (last && nb < r->max_blocks && used != n_bytes) return BR_ERR_INVALID_ARG;   // a truncated block at the end of the file
  if (nb == 0 && !at_end) return BR_OK;              // (only empty blocks so far)
  // a chunk to hold: what the last piece left over + this piece's bytes
  br_bam_reader::Chunk *ch = nullptr;
  {
    std::lock_guard<std::mutex> l(r->m);
    for (auto &x : r->chunks) if (!x->out && x.get() != r->carry_from) { ch = x.get(); break; }
    if (!ch) { r->chunks.push_back(std::make_unique<br_bam_reader::Chunk>()); ch = r->chunks.back().get(); }
  }
  RC(ch->data.ensure((size_t)(r->carry_len + total) + 64));
  if (r->carry_len) HIPCHK(hipMemcpyAsync(ch->data.p, r->carry_from->data.as<uint8_t>() + r->carry_off, (size_t)r->carry_len, hipMemcpyDeviceToDevice, st));
  if (nb) {
    RC(r->comp.ensure((size_t)used + 64));
    HIPCHK(hipMemcpyAsync(r->comp.p, data, (size_t)used, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    r->t_up += tsec(tp); tp = tnow();
    const uint8_t *o = nullptr; uint64_t ob = 0;
    RC(inflate_impl(r->c, r->comp.as<uint8_t>(), used, r->blocks.data(), nb, st, ch->data.as<uint8_t>() + r->carry_len, &o, &ob));
    r->t_inflate += tsec(tp); tp = tnow();
  }
  uint64_t have = r->carry_len + total, start = 0;
  if (r->skip) { start = std::min<uint64_t>(r->skip, have); r->skip -= start; }   // (the header never leaves a carry: nothing is split before it ends)
  const uint8_t *base = ch->data.as<uint8_t>() + start;
  const uint64_t nbytes = have - start;
  br_device_records recs; int64_t unm_all = 0; uint64_t used_bytes = 0; SplitArgs S{};
  RC(split_impl(r->c, base, nbytes, r->n_ref, st, &recs, &unm_all, &used_bytes, &S));
  r->t_split += tsec(tp); tp = tnow();
  const int64_t n = recs.n_aln;
  // the cut: everything in front of the last read-name group (it may go on in the next piece); at the end of the file, all
  int64_t n_take = n; uint64_t cut = used_bytes;
  RC(r->small.ensure(64));
  if (!at_end && n > 0) {
    HIPCHK(hipMemsetAsync(r->small.p, 0, 16, st));
    launch_last_group(st, base, recs.rec_off, n, (unsigned long long *)r->small.p);
    uint64_t g = 0;
    HIPCHK(hipMemcpyAsync(&g, r->small.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    n_take = (int64_t)g;
    uint64_t off_g = 0;
    HIPCHK(hipMemcpyAsync(&off_g, recs.rec_off + n_take, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    cut = off_g - 4;
  }
  if (at_end && used_bytes != nbytes) return BR_ERR_INVALID_ARG;   // a truncated record at the end of the file
  // unmapped records in front of the cut (the ones behind it are met again with the next piece)
  int64_t unm = unm_all;
  if (cut != used_bytes && S.n_seg) {
    HIPCHK(hipMemsetAsync(r->small.as<uint8_t>() + 16, 0, 8, st));
    launch_unmapped_before(st, S, cut, (unsigned long long *)(r->small.as<uint8_t>() + 16));
    uint64_t u = 0;
    HIPCHK(hipMemcpyAsync(&u, r->small.as<uint8_t>() + 16, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    unm = (int64_t)u;
  }
  // the bundle's tables live with the chunk (the context's are overwritten by the next piece)
  if (n_take) {
    RC(ch->off.ensure((size_t)n_take * 8)); RC(ch->len.ensure((size_t)n_take * 4));
    HIPCHK(hipMemcpyAsync(ch->off.p, recs.rec_off, (size_t)n_take * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(ch->len.p, recs.rec_len, (size_t)n_take * 4, hipMemcpyDeviceToDevice, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  {
    std::lock_guard<std::mutex> l(r->m);
    ch->id = r->next_id++; ch->out = true;
    r->carry_from = ch; r->carry_off = start + cut; r->carry_len = nbytes - cut;
  }
  r->t_cut += tsec(tp);
  if (at_end) {
    r->finished = true;
    if (getenv("BRAMBLE_AMD_TIMING")) fprintf(stderr, "[reader] block scan %.2fs, upload of the compressed bytes %.2fs, inflate %.2fs, record split %.2fs, cuts + tables %.2fs\n", r->t_scan, r->t_up, r->t_inflate, r->t_split, r->t_cut);
  }
  bundle->blob = base; bundle->rec_off = ch->off.as<uint64_t>(); bundle->rec_len = ch->len.as<uint32_t>(); bundle->n_aln = n_take;
  *id = ch->id; *n_unmapped = unm;
  return BR_OK;
}

// ---------------------------------------------------------------------------
// Piece-wise device reader.  The caller holds the whole file's block table (br_bgzf_scan over the mapping) and hands out
// pieces [b0, b1) of it -- to one reader in order, or to several readers on several devices: a piece needs nothing from
// its neighbours (see split_kernels.hip: the cut rule).  br_bam_piece_upload may run on another thread than
// br_bam_piece_process, one piece ahead (two slots).
// ---------------------------------------------------------------------------
extern "C" int br_bam_piece_upload(br_bam_reader *r, int slot, const uint8_t *file, uint64_t file_bytes, const br_bgzf_block *blocks,
                                   int64_t n_blocks, int64_t b0, int64_t b1x) {
  if (!r || slot < 0 || slot > 1 || !file || !blocks || b0 < 0 || b1x <= b0 || b1x > n_blocks) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(r->shell.device));
  if (!r->copy_st) HIPCHK(hipStreamCreateWithFlags(&r->copy_st, hipStreamNonBlocking));
  br_bam_reader::PieceSlot &P = r->pslot[slot];
  if (!P.up) HIPCHK(hipEventCreateWithFlags(&P.up, hipEventDisableTiming));
  const uint64_t src0 = blocks[b0].src_off, src1 = blocks[b1x - 1].src_off + blocks[b1x - 1].clen + 8;
  if (src1 > file_bytes || src1 <= src0) return BR_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> up_lock(r->up_m);
  RC(P.comp.ensure((size_t)(src1 - src0) + 64));
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t n = src1 - src0;
  const uint64_t n_chunks = (n + br_bam_reader::PIN_BYTES - 1) / br_bam_reader::PIN_BYTES;
  for (auto &row : r->pin) for (auto &pb : row) {
    if (!pb.p) { HIPCHK(hipHostMalloc((void **)&pb.p, br_bam_reader::PIN_BYTES, hipHostMallocDefault)); HIPCHK(hipEventCreateWithFlags(&pb.done, hipEventDisableTiming)); }
  }
  std::atomic<int> failed{0};
  auto work = [&](int w) {
    if (hipSetDevice(r->shell.device) != hipSuccess) { failed = 1; return; }
    int j = 0;
    for (uint64_t k = (uint64_t)w; k < n_chunks && !failed; k += br_bam_reader::PIN_THREADS, j ^= 1) {
      br_bam_reader::PinBuf &pb = r->pin[w][j];
      if (pb.used && hipEventSynchronize(pb.done) != hipSuccess) { failed = 1; return; }   // its last transfer (this call's or an earlier one's)
      const uint64_t off = k * br_bam_reader::PIN_BYTES, len = std::min<uint64_t>(br_bam_reader::PIN_BYTES, n - off);
      memcpy(pb.p, file + src0 + off, (size_t)len);
      if (hipMemcpyAsync(P.comp.as<uint8_t>() + off, pb.p, (size_t)len, hipMemcpyHostToDevice, r->copy_st) != hipSuccess ||
          hipEventRecord(pb.done, r->copy_st) != hipSuccess) { failed = 1; return; }
      pb.used = true;
    }
  };
  {
    std::vector<std::thread> th;
    const int nt = (int)std::min<uint64_t>(br_bam_reader::PIN_THREADS, n_chunks);
    for (int w = 1; w < nt; w++) th.emplace_back(work, w);
    work(0);
    for (auto &t : th) t.join();
  }
  if (failed) return BR_ERR_HIP;
  HIPCHK(hipEventRecord(P.up, r->copy_st));   // (everything queued above; br_bam_piece_process waits for it on its own stream)
  r->t_upload += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  P.b0 = b0; P.b1x = b1x; P.src0 = src0; P.n_src = src1 - src0;
  return BR_OK;
}

// Inflates the slot's blocks [b0, b1x) (b1x >= b1: the piece's own blocks and a few of the next piece's, for the END cut),
// splits them into records and returns the piece's bundle.
//   start_rel >= 0: the piece's first record starts that many inflated bytes behind the start of block b0 (the BAM header's
//                   size for the first piece; the END of the piece in front otherwise);  -1: guess it
//   info->start_rel / end_rel: where the bundle starts (behind block b0) and ends (behind block b1); end_rel of piece k is
//                   the start_rel of piece k + 1 -- a guessing reader's start_rel must equal its neighbour's end_rel, or the
//                   piece is to be processed again with that value
// Returns BR_PIECE_MORE (1) when the END cut lies beyond block b1x: upload more blocks and call again.
extern "C" int br_bam_piece_process(br_bam_reader *r, int slot, const br_bgzf_block *blocks, int64_t n_blocks, int64_t b1,
                                    int64_t start_rel, br_device_records *bundle, int64_t *id, br_piece_info *info) {
  if (!r || slot < 0 || slot > 1 || !blocks || !bundle || !id || !info) return BR_ERR_INVALID_ARG;
  br_bam_reader::PieceSlot &P = r->pslot[slot];
  const int64_t b0 = P.b0, b1x = P.b1x;
  if (b0 < 0 || b1 <= b0 || b1 > b1x || b1x > n_blocks) return BR_ERR_INVALID_ARG;
  memset(bundle, 0, sizeof(*bundle)); memset(info, 0, sizeof(*info));
  *id = -1;
  HIPCHK(hipSetDevice(r->shell.device));
  hipStream_t st = r->st;
  auto tp = std::chrono::steady_clock::now();
  const bool file_ends = b1x == n_blocks, last_piece = b1 == n_blocks;
  const uint64_t dst0 = blocks[b0].dst_off;
  const uint64_t total = blocks[b1x - 1].dst_off + blocks[b1x - 1].ulen - dst0;
  const uint64_t bound = last_piece ? total : blocks[b1].dst_off - dst0;   // where the next piece's first block starts
  r->pblocks.assign(blocks + b0, blocks + b1x);
  for (auto &b : r->pblocks) { b.src_off -= P.src0; b.dst_off -= dst0; }
  br_bam_reader::Chunk *ch = nullptr;
  {
    std::lock_guard<std::mutex> l(r->m);
    for (auto &x : r->chunks) if (!x->out) { ch = x.get(); break; }
    if (!ch) { r->chunks.push_back(std::make_unique<br_bam_reader::Chunk>()); ch = r->chunks.back().get(); }
  }
  RC(ch->data.ensure((size_t)total + 64));
  HIPCHK(hipStreamWaitEvent(st, P.up, 0));
  const uint8_t *o = nullptr; uint64_t ob = 0;
  RC(inflate_impl(r->c, P.comp.as<uint8_t>(), P.n_src, r->pblocks.data(), b1x - b0, st, ch->data.as<uint8_t>(), &o, &ob));
  RC(r->small.ensure(64));
  unsigned long long *cut = (unsigned long long *)r->small.p;
  uint64_t start = 0;
  const int guess = start_rel < 0 ? 1 : 0;
  if (guess) {   // the first offset that starts a run of records
    SplitArgs G{}; G.data = ch->data.as<uint8_t>(); G.n_bytes = total; G.n_ref = r->n_ref;
    launch_first_record(st, G, total, cut);
    unsigned long long e = 0;
    HIPCHK(hipMemcpyAsync(&e, cut, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (e == ~0ull) { if (file_ends) e = total; else return 1; }   // no record starts in here (one long record): more blocks
    start = e;
  } else {
    if ((uint64_t)start_rel > total) return file_ends ? BR_ERR_INVALID_ARG : 1;
    start = (uint64_t)start_rel;
  }
  const uint8_t *base = ch->data.as<uint8_t>() + start;
  const uint64_t nbytes = total - start;
  br_device_records recs; int64_t unm_all = 0; uint64_t used_bytes = 0; SplitArgs S{};
  RC(split_impl(r->c, base, nbytes, r->n_ref, st, &recs, &unm_all, &used_bytes, &S));
  if (file_ends && used_bytes != nbytes) return BR_ERR_INVALID_ARG;   // a truncated record at the end of the file
  const int64_t n = recs.n_aln;
  // the two cuts (see split_kernels.hip), their offsets, the unmapped records between: one read-back
  const unsigned long long init[5] = {guess ? (unsigned long long)n : 0ull, ~0ull, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(cut, init, sizeof(init), hipMemcpyHostToDevice, st));
  const uint64_t bound_rel = bound > start ? bound - start : 0;   // (relative to base)
  launch_piece_cut(st, S, recs.rec_off, n, last_piece ? ~0ull : bound_rel, used_bytes, guess, cut);
  unsigned long long h[5];
  HIPCHK(hipMemcpyAsync(h, cut, sizeof(h), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int64_t iS = (int64_t)std::min<unsigned long long>(h[0], (unsigned long long)n), iE = n;
  if (!last_piece) {
    if (h[1] == ~0ull) { if (!file_ends) return 1; }   // the group at the boundary goes on past the data: more blocks (or the file ends: all of it)
    else iE = (int64_t)h[1];
  }
  if (iS > iE) iS = iE;   // (a read-name group that covers the whole piece and more: the piece in front takes it all)
  const uint64_t off_S = h[2], off_E = std::max<uint64_t>(h[3], h[2]);
  const int64_t n_take = iE - iS;
  if (n_take) {
    RC(ch->off.ensure((size_t)n_take * 8)); RC(ch->len.ensure((size_t)n_take * 4));
    HIPCHK(hipMemcpyAsync(ch->off.p, recs.rec_off + iS, (size_t)n_take * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(ch->len.p, recs.rec_len + iS, (size_t)n_take * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  { std::lock_guard<std::mutex> l(r->m); ch->id = r->next_id++; ch->out = true; }
  bundle->blob = base; bundle->rec_off = ch->off.as<uint64_t>(); bundle->rec_len = ch->len.as<uint32_t>(); bundle->n_aln = n_take;
  *id = ch->id;
  info->start_rel = start + off_S;
  info->end_rel = start + off_E >= bound ? start + off_E - bound : 0;
  info->n_unmapped = (int64_t)h[4];
  info->guessed = guess; info->at_end = last_piece ? 1 : 0;
  r->t_proc += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
  return BR_OK;
}
extern "C" double br_bam_reader_seconds(const br_bam_reader *r) { return r ? r->t_proc : 0.0; }
extern "C" double br_bam_reader_upload_seconds(const br_bam_reader *r) { return r ? r->t_upload : 0.0; }

extern "C" int br_bam_split(const uint8_t *data, uint64_t n_bytes, int64_t cap, uint64_t *rec_off, uint32_t *rec_len,
                            int64_t *n_records, int64_t *n_unmapped, uint64_t *consumed) {
  if ((!data && n_bytes) || !rec_off || !rec_len || !n_records || !consumed || cap < 0) return BR_ERR_INVALID_ARG;
  uint64_t p = 0; int64_t n = 0, un = 0;
  while (n < cap && p + 4 <= n_bytes) {
    uint32_t bs; memcpy(&bs, data + p, 4);
    if (bs < 32) return BR_ERR_INVALID_ARG;
    if (p + 4 + (uint64_t)bs > n_bytes) break;  // partial record: next call
    const uint8_t *r = data + p + 4;
    uint32_t l_qname = r[8]; uint16_t ncig, flag; int32_t l_seq;
    memcpy(&ncig, r + 12, 2); memcpy(&flag, r + 14, 2); memcpy(&l_seq, r + 16, 4);
    uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
    if (32ull + l_qname + 4ull * ncig + (ls + 1) / 2 + ls > bs || l_qname == 0) return BR_ERR_INVALID_ARG;
    if (flag & 0x4) un++;
    else { rec_off[n] = p + 4; rec_len[n] = bs; n++; }
    p += 4 + (uint64_t)bs;
  }
  *n_records = n; if (n_unmapped) *n_unmapped = un; *consumed = p;
  return BR_OK;
}

extern "C" int br_bam_bundle_stage(br_ctx *c, const br_bam_bundle *bb, int slot) {
  if (!c || !bb || slot < 0 || slot > 2) return BR_ERR_INVALID_ARG;
  int64_t n = bb->n_records;
  if (n < 0 || (n && (!bb->blob || !bb->rec_off || !bb->rec_len))) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  br_ctx::StageSlot &S = c->stage[slot];
  if (!c->copy_stream) { int pl = 0, ph = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&pl, &ph)); HIPCHK(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, ph)); }   // (see ensure_streams)
  if (!S.ready) HIPCHK(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
  S.n = n;
  if (n) {
    // upload only the span the records cover
    uint64_t lo = bb->rec_off[0], hi = bb->rec_off[n - 1] + bb->rec_len[n - 1];
    if (hi > bb->n_bytes || lo > hi) return BR_ERR_INVALID_ARG;
    RC(S.blob.ensure((size_t)(hi - lo) + 16)); RC(S.off.ensure((size_t)n * 8)); RC(S.len.ensure((size_t)n * 4));
    S.h_off.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) S.h_off[(size_t)i] = bb->rec_off[i] - lo;
    HIPCHK(hipMemcpyAsync(S.blob.p, bb->blob + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(hipMemcpyAsync(S.off.p, S.h_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(hipMemcpyAsync(S.len.p, bb->rec_len, (size_t)n * 4, hipMemcpyHostToDevice, c->copy_stream));
  }
  HIPCHK(hipEventRecord(S.ready, c->copy_stream));
  return BR_OK;
}

// records in HBM -> projected records (or their BGZF blocks) in pinned host memory: the part the staged and the resident
// entry points share
static int project_bam_tail(br_ctx *c, const br_config *cfg, const br_device_records *dr, const int32_t *ref_map, int32_t n_ref_map,
                            bool bgzf_on_device, bool nowait, double wait_ms, br_host_bam *out) {
  static const bool timing = getenv("BRAMBLE_AMD_TIMING") != nullptr;
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  hipStream_t st = nullptr;
  const int64_t n = dr->n_aln;
  auto t1 = tnow();
  br_device_rows rows; br_device_bam db;
  RC(br_project_bam_device(c, cfg, dr, ref_map, n_ref_map, st, &rows, &db));
  auto t2 = tnow();
  int hs = c->h_bam_next; c->h_bam_next ^= 1;
  if (c->home_pending[hs]) { HIPCHK(hipEventSynchronize(c->ev_home[hs])); c->home_pending[hs] = false; }   // (a caller that never asked)
  const bool later = nowait && bgzf_on_device && db.n_bytes;
  if (bgzf_on_device && db.n_bytes) {
    c->z_dense_which = hs;
    const uint8_t *z = nullptr; uint64_t zn = 0;
    RC(deflate_device_impl(c, db.data, db.n_bytes, st, &z, &zn, false));
    db.data = z; db.n_bytes = zn;
  }
  auto t3 = tnow();
  if (db.n_bytes > c->h_bam_cap[hs]) {
    c->h_bam[hs] = nullptr; c->h_bam_cap[hs] = 0;
    size_t want = (size_t)db.n_bytes + (size_t)db.n_bytes / 4 + 4096;
    RC(c->h_bam_mem[hs].alloc(want));
    c->h_bam[hs] = c->h_bam_mem[hs].p; c->h_bam_cap[hs] = c->h_bam_mem[hs].cap;
  }
  if (later) {
    // everything on `st` is complete (the deflate step ends with the block sizes on the host): the copy goes to a stream of its
    // own and the caller asks for it with br_host_bam_wait, so the next bundle's kernels start without the 4 ms of PCIe in front
    if (!c->down_stream) HIPCHK(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    if (!c->ev_home[hs]) HIPCHK(hipEventCreateWithFlags(&c->ev_home[hs], hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(c->h_bam[hs], db.data, (size_t)db.n_bytes, hipMemcpyDeviceToHost, c->down_stream));
    HIPCHK(hipEventRecord(c->ev_home[hs], c->down_stream));
    c->home_pending[hs] = true;
  } else {
    if (db.n_bytes) HIPCHK(hipMemcpyAsync(c->h_bam[hs], db.data, (size_t)db.n_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  if (timing) fprintf(stderr, "[bundle] %lld records: upload wait %.1f ms, records -> records %.1f ms, deflate %.1f ms, download of %.0f MB %.1f ms\n", (long long)n, wait_ms, tms(t1, t2), tms(t2, t3), (double)db.n_bytes / 1e6, tms(t3, tnow()));
  out->data = c->h_bam[hs]; out->n_bytes = db.n_bytes; out->n_rows = db.n_rows;
  out->total_complete = rows.total_complete; out->total_unique = rows.total_unique;
  out->dropped_reads = rows.dropped_reads; out->total_processed = rows.total_processed;
  return BR_OK;
}

static int project_bam_staged_impl(br_ctx *c, const br_config *cfg, const br_bam_bundle *bb, int slot, br_host_bam *out, bool nowait) {
  if (!c || !cfg || !bb || !out || slot < 0 || slot > 2) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  br_ctx::StageSlot &S = c->stage[slot];
  if (!S.ready || S.n != bb->n_records) return BR_ERR_INVALID_ARG;   // not staged (or another bundle was)
  HIPCHK(hipSetDevice(c->ix->device));
  int64_t n = S.n;
  out->total_processed = (uint64_t)n;
  auto t0 = std::chrono::steady_clock::now();
  HIPCHK(hipEventSynchronize(S.ready));
  const double wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (n == 0) return BR_OK;
  br_device_records dr{S.blob.as<uint8_t>(), S.off.as<uint64_t>(), n, S.len.as<uint32_t>()};
  return project_bam_tail(c, cfg, &dr, bb->ref_map, bb->n_ref_map, bb->bgzf_on_device != 0, nowait, wait_ms, out);
}

extern "C" int br_project_bam_resident(br_ctx *c, const br_config *cfg, const br_device_records *recs, const int32_t *ref_map, int32_t n_ref_map,
                                       int bgzf_on_device, int nowait, br_host_bam *out) {
  if (!c || !cfg || !recs || !out || recs->n_aln < 0 || (recs->n_aln && (!recs->blob || !recs->rec_off))) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  HIPCHK(hipSetDevice(c->ix->device));
  out->total_processed = (uint64_t)recs->n_aln;
  if (recs->n_aln == 0) return BR_OK;
  return project_bam_tail(c, cfg, recs, ref_map, n_ref_map, bgzf_on_device != 0, nowait != 0, 0.0, out);
}

extern "C" int br_project_bam_staged(br_ctx *c, const br_config *cfg, const br_bam_bundle *bb, int slot, br_host_bam *out) {
  return project_bam_staged_impl(c, cfg, bb, slot, out, false);
}
extern "C" int br_project_bam_staged_nowait(br_ctx *c, const br_config *cfg, const br_bam_bundle *bb, int slot, br_host_bam *out) {
  return project_bam_staged_impl(c, cfg, bb, slot, out, true);
}
extern "C" int br_host_bam_wait(br_ctx *c, const br_host_bam *hb) {
  if (!c || !hb) return BR_ERR_INVALID_ARG;
  for (int k = 0; k < 2; k++)
    if (hb->data && hb->data == c->h_bam[k] && c->home_pending[k]) {
      HIPCHK(hipSetDevice(c->ix->device));
      HIPCHK(hipEventSynchronize(c->ev_home[k]));
      c->home_pending[k] = false;
    }
  return BR_OK;
}

extern "C" int br_project_bam_bundle(br_ctx *c, const br_config *cfg, const br_bam_bundle *bb, br_host_bam *out) {
  if (!c || !cfg || !bb || !out) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  RC(br_bam_bundle_stage(c, bb, 0));
  return br_project_bam_staged(c, cfg, bb, 0, out);
}

extern "C" int br_project_batch_device(br_ctx *c, const br_config *cfg, const br_device_batch *b, void *stream,
                                       br_device_rows *out) {
  if (!c || !cfg || !b || !out) return BR_ERR_INVALID_ARG;
  return run_device(c, cfg, b, (hipStream_t)stream, out);
}

extern "C" int br_device_rows_expand(br_ctx *c, void *stream, br_device_wide_rows *out) {
  if (!c || !out) return BR_ERR_INVALID_ARG;
  return expand_rows(c, (hipStream_t)stream, out);
}

// ---------------------------------------------------------------------------
// host-batch entry: upload, run, download, finalise primary flags
// ---------------------------------------------------------------------------
template <typename T>
static int h2d(DevBuf &buf, const T *src, size_t n, hipStream_t st) {
  RC(buf.ensure(std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIPCHK(hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
  return BR_OK;
}
template <typename T>
static int d2h(PinnedVec<T> &dst, const void *src, size_t n, hipStream_t st) {
  RC(dst.resize(n));
  if (n) HIPCHK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, st));
  return BR_OK;
}

// ---- flat batches: staging, the input contract on the device, packed rows home ----
static int ensure_streams(br_ctx *c) {
  // The runtime keeps a small pool of hardware queues per stream priority and lets streams of one priority share them once
  // there are more streams than queues: two streams on one queue run one after the other.  The context's kernel streams
  // (run, aux, aux2, the caller's) are of normal priority; the upload stream takes the high pool and the download stream the
  // low one, so that neither transfer ever queues behind the other or behind a kernel stream (a context that had already
  // created its aux streams -- a device-resident call first -- found its uploads and downloads serialised: 80 ms per
  // PCIe-inclusive step where 60 is the wire, profiles/pcie_phases.py)
  int prio_low = 0, prio_high = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  if (!c->copy_stream) HIPCHK(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_high));
  if (!c->run_stream) HIPCHK(hipStreamCreateWithFlags(&c->run_stream, hipStreamNonBlocking));
  if (!c->d2h_stream) HIPCHK(hipStreamCreateWithPriority(&c->d2h_stream, hipStreamNonBlocking, prio_low));
  if (!c->rows_busy) HIPCHK(hipEventCreateWithFlags(&c->rows_busy, hipEventDisableTiming));
  if (!c->alt.busy) HIPCHK(hipEventCreateWithFlags(&c->alt.busy, hipEventDisableTiming));
  return BR_OK;
}

extern "C" int br_batch_stage(br_ctx *c, const br_batch *b, int slot) {
  if (!c || !b || slot < 0 || slot > 1) return BR_ERR_INVALID_ARG;
  const int64_t n = b->n_aln;
  if (n < 0 || n >= 0x7fffffffll) return BR_ERR_CAPACITY;
  if (n && (!b->ref_id || !b->ref_start || !b->flags || !b->xs || !b->ts || !b->cigar_off || !b->name_off || !b->mate_ref_id ||
            !b->mate_start)) return BR_ERR_INVALID_ARG;
  const uint64_t n_words = n ? b->cigar_off[n] : 0, n_name = n ? b->name_off[n] : 0;
  const bool has_seq = b->seq_off && b->seqs;
  const uint64_t n_seq = (n && has_seq) ? b->seq_off[n] : 0;
  if (n_words >= 0xfffffff0ull - (uint64_t)n || n_name >= 0xfffffff0ull || n_seq >= 0xfffffff0ull) return BR_ERR_CAPACITY;
  if ((n_words && !b->cigar) || (n_name && !b->names)) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  RC(ensure_streams(c));
  br_ctx::InSlot &S = c->in_slot[slot];
  if (!S.ready) HIPCHK(hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
  if (!S.rows_home) HIPCHK(hipEventCreateWithFlags(&S.rows_home, hipEventDisableTiming));
  hipStream_t cs = c->copy_stream;
  S.n = n; S.n_words = n_words; S.n_name = n_name; S.n_seq = n_seq; S.has_seq = has_seq; S.staged = true;
  const size_t nn = (size_t)n;
  RC(h2d(S.ref_id, b->ref_id, nn, cs)); RC(h2d(S.ref_start, b->ref_start, nn, cs)); RC(h2d(S.flags, b->flags, nn, cs));
  RC(h2d(S.xs, b->xs, nn, cs)); RC(h2d(S.ts, b->ts, nn, cs));
  RC(h2d(S.mate_ref, b->mate_ref_id, nn, cs)); RC(h2d(S.mate_start, b->mate_start, nn, cs));
  RC(S.cigar_off64.ensure((nn + 1) * 8)); RC(S.name_off64.ensure((nn + 1) * 8));
  if (n) {
    HIPCHK(hipMemcpyAsync(S.cigar_off64.p, b->cigar_off, (nn + 1) * 8, hipMemcpyHostToDevice, cs));
    HIPCHK(hipMemcpyAsync(S.name_off64.p, b->name_off, (nn + 1) * 8, hipMemcpyHostToDevice, cs));
  } else {
    HIPCHK(hipMemsetAsync(S.cigar_off64.p, 0, 8, cs)); HIPCHK(hipMemsetAsync(S.name_off64.p, 0, 8, cs));
  }
  RC(h2d(S.cigar, b->cigar, (size_t)n_words, cs)); RC(h2d(S.names, (const uint8_t *)b->names, (size_t)n_name, cs));
  RC(S.lqseq.ensure(std::max<size_t>(nn, 1) * 4));
  if (b->l_qseq) { if (n) HIPCHK(hipMemcpyAsync(S.lqseq.p, b->l_qseq, nn * 4, hipMemcpyHostToDevice, cs)); }
  else HIPCHK(hipMemsetAsync(S.lqseq.p, 0, std::max<size_t>(nn, 1) * 4, cs));
  if (has_seq) {
    RC(S.seq_off64.ensure((nn + 1) * 8));
    if (n) HIPCHK(hipMemcpyAsync(S.seq_off64.p, b->seq_off, (nn + 1) * 8, hipMemcpyHostToDevice, cs));
    else HIPCHK(hipMemsetAsync(S.seq_off64.p, 0, 8, cs));
    RC(h2d(S.seqs, (const uint8_t *)b->seqs, (size_t)n_seq, cs));
  }
  HIPCHK(hipEventRecord(S.ready, cs));
  return BR_OK;
}

// The staged slot's input contract on the device (what br_batch_prepare / br_batch_seq_source compute on the host:
// read-name groups src/core.cpp:347-380, mate index src/bramble.cpp:272-311, the group's shared sequence
// src/core.cpp:353-378) and the device batch over it.
static int prep_staged(br_ctx *c, const br_config *cfg, br_ctx::InSlot &S, hipStream_t st, br_device_batch *db) {
  memset(db, 0, sizeof(*db));
  const int64_t n = S.n;
  const size_t nn = (size_t)n;
  HIPCHK(hipStreamWaitEvent(st, S.ready, 0));
  db->n_aln = n;
  if (n == 0) return BR_OK;
  RC(S.cigar_off.ensure((nn + 1) * 4)); RC(S.name_off.ensure((nn + 1) * 4)); RC(S.isnew.ensure(nn * 4));
  RC(S.group_pre.ensure((nn + 1) * 4)); RC(S.mate_idx.ensure(nn * 4));
  if (S.has_seq) { RC(S.seq_off.ensure((nn + 1) * 4)); RC(S.seq_src.ensure(nn * 4)); }
  RC(c->p_small.ensure(64)); RC(c->p_big.ensure((nn / 96 + 2) * 4));
  RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for(n + 1), 1) * 8 * 3));
  RC(c->totals.ensure(16 * 8));
  HIPCHK(hipMemsetAsync(c->p_small.p, 0, 64, st));
  SoaArgs A{};
  A.n = n; A.cigar_off64 = S.cigar_off64.as<uint64_t>(); A.name_off64 = S.name_off64.as<uint64_t>();
  A.seq_off64 = S.has_seq ? S.seq_off64.as<uint64_t>() : nullptr;
  A.cigar_off = S.cigar_off.as<uint32_t>(); A.name_off = S.name_off.as<uint32_t>(); A.seq_off = S.has_seq ? S.seq_off.as<uint32_t>() : nullptr;
  A.names = S.names.as<uint8_t>(); A.cigar = S.cigar.as<uint32_t>(); A.isnew = S.isnew.as<uint32_t>(); A.maxima = c->p_small.as<uint32_t>();
  launch_soa_fields(st, A);
  uint64_t *d_tot = c->totals.as<uint64_t>();
  ScanArgs SC{}; SC.n = n; SC.tile_sums = c->tile_sums.as<uint64_t>(); SC.src32 = A.isnew;
  launch_scan(st, SC, 2, S.group_pre.p, false, d_tot + 9);
  HIPCHK(hipMemcpyAsync(c->h_totals + 26, d_tot + 9, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(c->h_totals + 27, c->p_small.p, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint64_t ng = c->h_totals[26];
  const uint32_t max_nc = (uint32_t)(c->h_totals[27] & 0xffffffffu), max_clip = (uint32_t)(c->h_totals[27] >> 32);
  RC(S.group_off.ensure(((size_t)ng + 1) * 4));
  ParseArgs P{};
  P.n = n; P.n_groups = (int64_t)ng; P.isnew = A.isnew; P.group_pre = S.group_pre.as<uint32_t>(); P.group_off = S.group_off.as<uint32_t>();
  P.flags = S.flags.as<uint16_t>(); P.ref_id = S.ref_id.as<int32_t>(); P.ref_start = S.ref_start.as<int32_t>();
  P.mate_ref_id = S.mate_ref.as<int32_t>(); P.mate_start = S.mate_start.as<int32_t>(); P.mate_idx = S.mate_idx.as<int32_t>();
  P.n_big_groups = c->p_small.as<uint32_t>() + 2; P.big_groups = c->p_big.as<uint32_t>();
  launch_group_off(st, P);
  launch_mates(st, P);
  db->n_groups = (int64_t)ng; db->ref_id = P.ref_id; db->ref_start = P.ref_start; db->flags = P.flags;
  db->xs = S.xs.as<int8_t>(); db->ts = S.ts.as<int8_t>(); db->cigar_off = A.cigar_off; db->cigar = A.cigar;
  db->mate_idx = P.mate_idx; db->group_off = P.group_off; db->l_qseq = S.lqseq.as<int32_t>();
  db->n_cigar_words = (int64_t)S.n_words; db->max_n_cigar = (int32_t)max_nc;
  db->name_off = A.name_off; db->names = A.names;
  if (cfg->use_fasta && (cfg->lr || cfg->lr_hq)) {
    if (!S.has_seq) return BR_ERR_INVALID_ARG;
    P.seq_off = A.seq_off; P.seq_src = S.seq_src.as<int32_t>();
    launch_seq_src(st, P);
    db->seq_off = A.seq_off; db->seqs = S.seqs.as<uint8_t>(); db->seq_src = P.seq_src; db->max_soft_clip = (int32_t)max_clip;
  }
  return BR_OK;
}

extern "C" int br_project_staged(br_ctx *c, const br_config *cfg, int slot, br_host_rows *out) {
  if (!c || !cfg || !out || slot < 0 || slot > 1) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  br_ctx::InSlot &S = c->in_slot[slot];
  if (!S.staged) return BR_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->ix->device));
  RC(ensure_streams(c));
  if (S.rows_pending) { HIPCHK(hipEventSynchronize(S.rows_home)); S.rows_pending = false; }  // the slot's pinned arrays are rewritten below
  hipStream_t st = c->run_stream;
  // the other set of row tables: what the last call's download reads stays untouched (rows_busy follows its set)
  std::swap(c->pk_a, c->alt.pk_a); std::swap(c->pk_x, c->alt.pk_x); std::swap(c->pk_sim, c->alt.pk_sim); std::swap(c->pk_clip, c->alt.pk_clip);
  std::swap(c->pk_ch, c->alt.pk_ch); std::swap(c->pool, c->alt.pool); std::swap(c->row_off, c->alt.row_off);
  std::swap(c->rows_busy, c->alt.busy); std::swap(c->rows_busy_set, c->alt.busy_set);
  c->detail_valid = false; c->wide_valid = false; c->last_direct = false;   // (they describe the other set)
  br_device_batch db;
  RC(prep_staged(c, cfg, S, st, &db));
  S.staged = false;
  br_device_rows pr;
  { WantDetail wd(c, c->host_detail != 0); RC(run_device(c, cfg, &db, st, &pr)); }   // returns with the stream drained
  const size_t nr = (size_t)pr.n_rows, nn = (size_t)S.n;
  // the long (> 2 op) rewritten CIGARs sit in the sparse arena: a dense copy for the host (sizes -> scan -> copy)
  size_t np = 0;
  if (nr) {
    RC(c->pool_sizes.ensure(nr * 4)); RC(c->pool_off.ensure((nr + 1) * 8)); RC(c->pk_ch.ensure(nr * sizeof(uint2)));
    RC(c->tile_sums.ensure((size_t)std::max<int64_t>(scan_tiles_for((int64_t)nr + 1), 1) * 8 * 3));
    PoolArgs Q{};
    Q.n_rows = (int64_t)nr; Q.r_a = c->pk_a.as<uint4>(); Q.r_c = c->pk_c.as<uint2>(); Q.arena = c->cig_arena.as<uint32_t>();
    Q.sizes = c->pool_sizes.as<uint32_t>(); Q.off = c->pool_off.as<uint64_t>(); Q.c_out = c->pk_ch.as<uint2>();
    if (c->host_detail) RC(ensure_detail(c, st));
    launch_pool_sizes(st, Q);
    ScanArgs SP{}; SP.n = (int64_t)nr; SP.src32 = Q.sizes; SP.tile_sums = c->tile_sums.as<uint64_t>();
    launch_scan(st, SP, 2, c->pool_off.p, true, c->totals.as<uint64_t>() + 10);
    HIPCHK(hipMemcpyAsync(c->h_totals + 28, c->totals.as<uint64_t>() + 10, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    np = (size_t)c->h_totals[28];
    RC(c->pool.ensure(std::max<size_t>(np, 1) * 4));
    Q.pool = c->pool.as<uint32_t>();
    launch_pool_copy(st, Q, np > 8 * nr);
    HIPCHK(hipStreamSynchronize(st));
  }
  hipStream_t ds = c->d2h_stream;
  RC(d2h(S.h_a, pr.a, nr, ds)); RC(d2h(S.h_c, c->pk_ch.p, nr, ds)); RC(d2h(S.h_pool, c->pool.p, np, ds));
  RC(S.h_row_off.resize(nn + 1));
  if (nn) HIPCHK(hipMemcpyAsync(S.h_row_off.data(), pr.row_off, (nn + 1) * 8, hipMemcpyDeviceToHost, ds));
  else S.h_row_off.p[0] = 0;
  RC(d2h(S.h_mate, db.mate_idx, nn, ds));
  if (c->host_detail) RC(d2h(S.h_x, c->pk_x.p, nr, ds));
  if (pr.similarity_score) { RC(d2h(S.h_sim, pr.similarity_score, nr, ds)); RC(d2h(S.h_clip, pr.clip_score, nr, ds)); }
  HIPCHK(hipEventRecord(S.rows_home, ds));
  HIPCHK(hipEventRecord(c->rows_busy, ds));
  c->rows_busy_set = true; S.rows_pending = true;
  out->n_rows = pr.n_rows; out->n_aln = S.n; out->n_groups = db.n_groups; out->n_pool_words = (int64_t)np;
  out->a = (const br_row_a *)S.h_a.data(); out->cigar = S.h_c.data(); out->pool = S.h_pool.data();
  out->row_off = S.h_row_off.data(); out->mate_idx = S.h_mate.data();
  out->x = c->host_detail ? (const br_row_x *)S.h_x.data() : nullptr;
  out->similarity_score = pr.similarity_score ? S.h_sim.data() : nullptr;
  out->clip_score = pr.similarity_score ? S.h_clip.data() : nullptr;
  out->total_complete = pr.total_complete; out->total_unique = pr.total_unique;
  out->dropped_reads = pr.dropped_reads; out->total_processed = pr.total_processed;
  return BR_OK;
}

extern "C" int br_host_rows_wait(br_ctx *c, int slot) {
  if (!c || slot < 0 || slot > 1) return BR_ERR_INVALID_ARG;
  br_ctx::InSlot &S = c->in_slot[slot];
  if (S.rows_pending) { HIPCHK(hipEventSynchronize(S.rows_home)); S.rows_pending = false; }
  return BR_OK;
}

extern "C" int br_project_batch_packed(br_ctx *c, const br_config *cfg, const br_batch *b, br_host_rows *out) {
  if (!c || !cfg || !b || !out) return BR_ERR_INVALID_ARG;
  RC(br_batch_stage(c, b, 0));
  RC(br_project_staged(c, cfg, 0, out));
  return br_host_rows_wait(c, 0);
}

extern "C" int br_pin_host(void *p, size_t bytes) {
  if (!p || !bytes) return BR_ERR_INVALID_ARG;
  HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return BR_OK;
}
extern "C" int br_unpin_host(void *p) {
  if (!p) return BR_ERR_INVALID_ARG;
  HIPCHK(hipHostUnregister(p));
  return BR_OK;
}

// The wide host rows (ABI version 1 layout): the same staging and device-side input contract, then the wide view
// derived on the device and downloaded array by array.
extern "C" int br_project_batch(br_ctx *c, const br_config *cfg, const br_batch *b, br_rows *out) {
  if (!c || !cfg || !b || !out) return BR_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  RC(br_batch_stage(c, b, 0));
  br_ctx::InSlot &S = c->in_slot[0];
  hipStream_t st = c->run_stream;
  br_device_batch db;
  RC(prep_staged(c, cfg, S, st, &db));
  S.staged = false;
  br_device_rows pr;
  { WantDetail wd(c, true); RC(run_device(c, cfg, &db, st, &pr)); }
  br_device_wide_rows dr;
  RC(expand_rows(c, st, &dr));

  size_t nr = (size_t)dr.n_rows;
  RC(d2h(c->h_input, dr.input_index, nr, st)); RC(d2h(c->h_tid, dr.transcript_id, nr, st));
  RC(d2h(c->h_pos, dr.pos, nr, st)); RC(d2h(c->h_strand, dr.strand, nr, st));
  RC(d2h(c->h_cigoff, dr.cigar_off, nr ? nr + 1 : 0, st)); RC(d2h(c->h_cigar, dr.cigar, (size_t)dr.n_cigar_words, st));
  RC(d2h(c->h_sim, dr.similarity_score, nr, st)); RC(d2h(c->h_clip, dr.clip_score, nr, st));
  RC(d2h(c->h_junc, dr.junc_hits, nr, st)); RC(d2h(c->h_refc, dr.aligned_len, nr, st));
  RC(d2h(c->h_nh, dr.nh, nr, st)); RC(d2h(c->h_hi, dr.hi, nr, st)); RC(d2h(c->h_mapq, dr.mapq, nr, st));
  RC(d2h(c->h_paired, dr.is_paired, nr, st)); RC(d2h(c->h_same, dr.same_transcript_as_mate, nr, st));
  RC(d2h(c->h_first, dr.is_first, nr, st)); RC(d2h(c->h_mate_tid, dr.mate_transcript_id, nr, st));
  RC(d2h(c->h_mate_pos, dr.mate_pos, nr, st)); RC(d2h(c->h_isize, dr.insert_size, nr, st));
  RC(d2h(c->h_group, dr.group, nr, st)); RC(d2h(c->h_primary, dr.is_primary, nr, st));
  HIPCHK(hipStreamSynchronize(st));
  if (nr == 0) { RC(c->h_cigoff.resize(1)); c->h_cigoff.p[0] = 0; }

  out->n_rows = (int64_t)nr;
  out->input_index = c->h_input.data(); out->transcript_id = c->h_tid.data(); out->pos = c->h_pos.data();
  out->strand = c->h_strand.data(); out->cigar_off = c->h_cigoff.data(); out->cigar = c->h_cigar.data();
  out->similarity_score = c->h_sim.data(); out->clip_score = c->h_clip.data(); out->junc_hits = c->h_junc.data();
  out->aligned_len = c->h_refc.data(); out->nh = c->h_nh.data(); out->hi = c->h_hi.data(); out->mapq = c->h_mapq.data();
  out->is_primary = c->h_primary.data(); out->is_paired = c->h_paired.data();
  out->same_transcript_as_mate = c->h_same.data(); out->is_first = c->h_first.data();
  out->mate_transcript_id = c->h_mate_tid.data(); out->mate_pos = c->h_mate_pos.data();
  out->insert_size = c->h_isize.data(); out->group = c->h_group.data();
  out->total_complete = pr.total_complete; out->total_unique = pr.total_unique;
  out->dropped_reads = pr.dropped_reads; out->total_processed = pr.total_processed;
  return BR_OK;
}

extern "C" uint32_t br_row_mapq(uint32_t nh, int long_reads);
static int project_groups_lean(br_ctx *c, const br_config *cfg, const br_batch &b, const std::vector<uint64_t> &kept,
                               const std::vector<char> &read_strand, const br_projected **out, size_t *n_out) {
  const size_t n = (size_t)b.n_aln;
  const uint64_t n_words = b.cigar_off[n], n_name = b.name_off[n], n_seq = b.seq_off ? b.seq_off[n] : 0;
  if (n_words >= 0x7fffffffull || n_name >= 0x7fffffffull || n_seq >= 0x7fffffffull) return BR_RETRY_ORDINARY;
  HIPCHK(hipSetDevice(c->ix->device));
  RC(ensure_streams(c));
  hipStream_t st = c->run_stream;
  // the contract on the host (src/core.cpp:347-380, src/bramble.cpp:272-311, src/core.cpp:353-378)
  std::vector<int32_t> mate_idx(n), seq_src;
  std::vector<uint32_t> group_off(n + 1);
  int64_t ng = 0;
  RC(br_batch_prepare(&b, mate_idx.data(), group_off.data(), &ng));
  if (b.seq_off) { seq_src.resize(n); RC(br_batch_seq_source(&b, group_off.data(), ng, seq_src.data())); }
  // one packed upload: every array at a 16-byte aligned offset
  size_t at = 0;
  auto place = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 15) & ~(size_t)15; return o; };
  const size_t o_ref = place(4 * n), o_start = place(4 * n), o_flags = place(2 * n), o_xs = place(n), o_ts = place(n),
               o_coff = place(4 * (n + 1)), o_cig = place(4 * (size_t)n_words), o_mate = place(4 * n), o_goff = place(4 * ((size_t)ng + 1)),
               o_lq = place(4 * n), o_noff = place(4 * (n + 1)), o_names = place((size_t)n_name),
               o_soff = place(b.seq_off ? 4 * (n + 1) : 0), o_seqs = place((size_t)n_seq), o_ssrc = place(b.seq_off ? 4 * n : 0);
  RC(c->g_host.resize(at + 16));
  RC(c->g_dev.ensure(at + 16));
  uint8_t *h = c->g_host.data();
  memcpy(h + o_ref, b.ref_id, 4 * n); memcpy(h + o_start, b.ref_start, 4 * n); memcpy(h + o_flags, b.flags, 2 * n);
  memcpy(h + o_xs, b.xs, n); memcpy(h + o_ts, b.ts, n);
  int32_t max_nc = 0, max_clip = 0;
  for (size_t i = 0; i <= n; i++) { ((uint32_t *)(h + o_coff))[i] = (uint32_t)b.cigar_off[i]; ((uint32_t *)(h + o_noff))[i] = (uint32_t)b.name_off[i]; }
  for (size_t i = 0; i < n; i++) {
    const uint64_t c0 = b.cigar_off[i], c1 = b.cigar_off[i + 1];
    max_nc = std::max<int32_t>(max_nc, (int32_t)(c1 - c0));
    if (c1 > c0) {   // leading / trailing soft clips (sizing of the rescue buffers), as k_soa_fields
      uint32_t w = b.cigar[c0];
      if ((w & 0xfu) == 5u && c1 - c0 > 1) w = b.cigar[c0 + 1];
      if ((w & 0xfu) == 4u) max_clip = std::max<int32_t>(max_clip, (int32_t)(w >> 4));
      w = b.cigar[c1 - 1];
      if ((w & 0xfu) == 5u && c1 - c0 > 1) w = b.cigar[c1 - 2];
      if ((w & 0xfu) == 4u) max_clip = std::max<int32_t>(max_clip, (int32_t)(w >> 4));
    }
  }
  memcpy(h + o_cig, b.cigar, 4 * (size_t)n_words); memcpy(h + o_mate, mate_idx.data(), 4 * n);
  memcpy(h + o_goff, group_off.data(), 4 * ((size_t)ng + 1));
  if (b.l_qseq) memcpy(h + o_lq, b.l_qseq, 4 * n); else memset(h + o_lq, 0, 4 * n);
  memcpy(h + o_names, b.names, (size_t)n_name);
  if (b.seq_off) {
    for (size_t i = 0; i <= n; i++) ((uint32_t *)(h + o_soff))[i] = (uint32_t)b.seq_off[i];
    memcpy(h + o_seqs, b.seqs, (size_t)n_seq); memcpy(h + o_ssrc, seq_src.data(), 4 * n);
  }
  HIPCHK(hipMemcpyAsync(c->g_dev.p, h, at, hipMemcpyHostToDevice, st));
  const uint8_t *d = c->g_dev.as<uint8_t>();
  br_device_batch db{};
  db.n_aln = (int64_t)n; db.n_groups = ng;
  db.ref_id = (const int32_t *)(d + o_ref); db.ref_start = (const int32_t *)(d + o_start); db.flags = (const uint16_t *)(d + o_flags);
  db.xs = (const int8_t *)(d + o_xs); db.ts = (const int8_t *)(d + o_ts); db.cigar_off = (const uint32_t *)(d + o_coff);
  db.cigar = (const uint32_t *)(d + o_cig); db.mate_idx = (const int32_t *)(d + o_mate); db.group_off = (const uint32_t *)(d + o_goff);
  db.l_qseq = (const int32_t *)(d + o_lq); db.name_off = (const uint32_t *)(d + o_noff); db.names = d + o_names;
  db.n_cigar_words = (int64_t)n_words; db.max_n_cigar = max_nc;
  if (b.seq_off) { db.seq_off = (const uint32_t *)(d + o_soff); db.seqs = d + o_seqs; db.seq_src = (const int32_t *)(d + o_ssrc); db.max_soft_clip = max_clip; }
  br_device_rows pr;
  c->rows_to_host = true;
  int rrc;
  { WantDetail wd(c, true); rrc = run_device(c, cfg, &db, st, &pr); }   // returns with the stream drained
  c->rows_to_host = false;
  RC(rrc);
  const size_t nr = (size_t)pr.n_rows, np = (size_t)pr.n_pool_words;
  if (np > (1u << 20)) return BR_RETRY_ORDINARY;   // a CIGAR arena of more than 4 MB: the dense pool of the batch path
  bool pool_home = false;
  if (!c->rows_at_host) {   // the call went down the ordinary pipeline (-S, or a dense locus): fetch the rows
    RC(ensure_detail(c, st));
    RC(d2h(c->g_a, pr.a, nr, st)); RC(d2h(c->g_c, pr.cigar, nr, st)); RC(d2h(c->g_x, c->pk_x.p, nr, st));
    if (pr.similarity_score) RC(d2h(c->g_sim, pr.similarity_score, nr, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  else c->last_n_rows = 0;   // the context's device row tables were not written: nothing for br_device_rows_expand / br_bam_encode_device to find
  c->rows_at_host = false;
  size_t n_cig_words = 0;
  for (size_t r = 0; r < nr; r++) n_cig_words += c->g_a.p[r].z & RM_NCIG;
  RC(c->g_cig.resize(n_cig_words + 1));
  c->h_proj.resize(nr);
  size_t cw = 0;
  const int long_reads = (cfg->lr || cfg->lr_hq) ? 1 : 0;
  for (size_t r = 0; r < nr; r++) {
    const uint4 a = c->g_a.p[r], x = c->g_x.p[r];
    const uint2 cr = c->g_c.p[r];
    const uint32_t meta = a.z, nc = meta & RM_NCIG;
    br_projected &p = c->h_proj[r];
    p.transcript_id = a.x; p.transcript_start = a.y;
    p.aligned_len = (uint32_t)std::max<int32_t>((int32_t)x.z, 0);
    uint64_t e = (uint64_t)p.transcript_start + p.aligned_len;  // saturating add, then saturating sub 1
    if (e > 0xffffffffull) e = 0xffffffffull;
    p.transcript_end = e ? (uint32_t)(e - 1) : 0;
    uint32_t *cg = c->g_cig.p + cw;
    if (nc <= 2) { if (nc > 0) cg[0] = cr.x; if (nc > 1) cg[1] = cr.y; }
    else {
      if (!pool_home) {   // the arena's used part, once, when a record has more than two ops
        RC(d2h(c->g_pool, pr.pool, np, st));
        HIPCHK(hipStreamSynchronize(st));
        pool_home = true;
      }
      const uint64_t off = ((uint64_t)cr.y << 32) | cr.x;
      if (off + nc > np) return BR_ERR_HIP;
      memcpy(cg, c->g_pool.p + off, 4 * (size_t)nc);
    }
    cw += nc;
    uint32_t qa = 0;
    for (uint32_t k = 0; k < nc; k++) {
      const uint32_t op = cg[k] & 0xf;
      if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_MATCH_OVR || op == OP_INS_OVR) qa += cg[k] >> 4;
    }
    p.query_aligned_len = qa;
    const size_t bi = (size_t)x.x;
    p.transcript_strand = (meta & RM_MINUS) ? '-' : '+';
    p.is_reverse = p.transcript_strand != read_strand[bi];   // api.rs:453 <- evaluate.rs:1062 (see project_groups_impl)
    p.similarity_score = pr.similarity_score ? c->g_sim.p[r] : 0.0;
    p.nh = a.w; p.hi = x.w; p.is_primary = (meta & RM_PRIMARY) ? 1 : 0;
    p.same_transcript_as_mate = (meta & RM_SAME) ? 1 : 0; p.is_paired_out = (meta & RM_PAIRED) ? 1 : 0;
    int32_t isize = 0;   // set_mate_info (src/bam.cpp:531-588): the pair's other record is the adjacent row
    if ((meta & RM_PAIRED) && (meta & RM_SAME)) {
      const uint4 o = c->g_a.p[(meta & RM_FIRST) ? r + 1 : r - 1];
      const int32_t my_pos = (int32_t)a.y, mate_pos = (int32_t)o.y, lq = b.l_qseq ? b.l_qseq[bi] : 0;
      isize = (my_pos <= mate_pos) ? (mate_pos + lq) - my_pos : -((my_pos + lq) - mate_pos);
    }
    p.insert_size = isize; p.input_index = kept[bi];
    p.mapq = br_row_mapq(a.w, long_reads); p.cigar = cg; p.n_cigar = nc;
  }
  *out = c->h_proj.data(); *n_out = nr;
  return BR_OK;
}

// project_group_with (bramble-rs/src/api.rs:285-464), AoS in/out.  Shape and field meanings are the Rust library's;
// the values are the C++ path's: mates pair up by the C++ rule (name + position hash, src/bramble.cpp:272-311
// = k_mates), not by find_mate_pairs' mutual pointers (groups.rs:126-190), and hit_index is carried for layout parity
// only -- neither the C++ reader nor find_mate_pairs reads it.
// single_name: one call = one query name (br_project_group); else any number of name-collated groups (br_project_groups).
static int project_groups_impl(br_ctx *c, const br_config *cfg, const br_alignment *alns, size_t n, bool single_name,
                               const br_projected **out, size_t *n_out) {
  if (!c || !cfg || (!alns && n) || !out || !n_out) return BR_ERR_INVALID_ARG;
  *out = nullptr; *n_out = 0;
  std::vector<int32_t> ref_id, ref_start, mate_ref, mate_start, lq;
  std::vector<uint16_t> flags; std::vector<int8_t> xs, ts;
  std::vector<uint64_t> coff(1, 0), noff(1, 0), soff(1, 0);
  std::vector<uint32_t> cig; std::string names, seqs;
  std::vector<uint64_t> kept;          // batch position -> caller's index (alignments with ref_id < 0 are skipped, api.rs:316-318)
  std::vector<char> read_strand;       // infer_strand (api.rs:470-489) per kept alignment
  const char *name0 = n ? (alns[0].query_name ? alns[0].query_name : "") : "";
  bool any_seq = false;
  for (size_t i = 0; i < n; i++) {
    const br_alignment &a = alns[i];
    const char *nm = a.query_name ? a.query_name : "";
    // one call = one query name (GenomicAlignment::query_name: "shared by all alignments in the group", api.rs:74-75)
    if (single_name && strcmp(nm, name0) != 0) return BR_ERR_INVALID_ARG;
    if (a.ref_start < 0 || a.ref_start > 0x7fffffffll || a.mate_ref_start < 0 || a.mate_ref_start > 0x7fffffffll) return BR_ERR_INVALID_ARG;
    if ((a.n_cigar && !a.cigar) || (a.sequence_len && !a.sequence)) return BR_ERR_INVALID_ARG;
    if (a.ref_id < 0) continue;        // api.rs:316-318
    kept.push_back(i);
    ref_id.push_back(a.ref_id); ref_start.push_back((int32_t)a.ref_start);
    uint16_t f = 0;
    if (a.is_paired) { f |= 0x1; if (a.mate_is_unmapped) f |= 0x8; f |= a.is_first_in_pair ? 0x40 : 0x80; }
    if (a.is_reverse) f |= 0x10;
    flags.push_back(f); xs.push_back((int8_t)a.xs_strand); ts.push_back((int8_t)a.ts_strand);
    char rs = '.';
    if (a.xs_strand == '+' || a.xs_strand == '-') rs = a.xs_strand;
    else if (a.ts_strand == '+' || a.ts_strand == '-') rs = a.is_reverse ? (a.ts_strand == '+' ? '-' : '+') : a.ts_strand;
    read_strand.push_back(rs);
    mate_ref.push_back(a.mate_ref_id); mate_start.push_back((int32_t)a.mate_ref_start);
    cig.insert(cig.end(), a.cigar, a.cigar + a.n_cigar); coff.push_back(cig.size());
    names += nm; noff.push_back(names.size());
    // sequence: Option<Vec<u8>> (api.rs:91-95); the clip rescue shares the first one of the group (api.rs:308-312, src/core.cpp:353-378)
    if (a.sequence && a.sequence_len) { seqs.append(a.sequence, a.sequence_len); any_seq = true; }
    soff.push_back(seqs.size());
    lq.push_back((int32_t)(a.read_len ? a.read_len : a.sequence_len));   // api.rs:345-349
  }
  const size_t nk = kept.size();
  if (nk == 0) { c->h_proj.clear(); *out = c->h_proj.data(); return BR_OK; }   // api.rs:392-394
  br_batch b{};
  b.n_aln = (int64_t)nk; b.ref_id = ref_id.data(); b.ref_start = ref_start.data(); b.flags = flags.data();
  b.xs = xs.data(); b.ts = ts.data(); b.cigar_off = coff.data(); b.cigar = cig.data();
  b.mate_ref_id = mate_ref.data(); b.mate_start = mate_start.data(); b.name_off = noff.data();
  b.names = names.data(); b.l_qseq = lq.data();
  if (any_seq) { b.seq_off = soff.data(); b.seqs = seqs.data(); }
  else if (cfg->use_fasta && (cfg->lr || cfg->lr_hq)) { seqs.assign(1, 'N'); b.seq_off = soff.data(); b.seqs = seqs.data(); }  // no sequence: nothing to rescue
  // The lean way (a call that carries a name group or a few dozen of them): the input contract -- read-name groups, mate
  // index, the group's shared sequence -- on the host (a few alignments), ONE upload of everything, the device path
  // (without host round trips at this size), the packed rows and the CIGAR words they point at back in a handful of small
  // copies, and the record fields put together here.  The staged batch path (a dozen uploads, the contract on the device,
  // the wide row view, twenty-one downloads) is built for bundles; it stays the route for large calls.
  if (nk <= 8192) {
    int rc = project_groups_lean(c, cfg, b, kept, read_strand, out, n_out);
    if (rc != BR_RETRY_ORDINARY) return rc;
  }
  br_rows rows;
  RC(br_project_batch(c, cfg, &b, &rows));
  c->h_proj.resize((size_t)rows.n_rows);
  for (int64_t r = 0; r < rows.n_rows; r++) {
    br_projected &p = c->h_proj[(size_t)r];
    p.transcript_id = rows.transcript_id[r];
    p.transcript_start = rows.pos[r];
    p.aligned_len = (uint32_t)std::max(rows.aligned_len[r], 0);
    uint64_t e = (uint64_t)p.transcript_start + p.aligned_len;  // saturating add, then saturating sub 1
    if (e > 0xffffffffull) e = 0xffffffffull;
    p.transcript_end = e ? (uint32_t)(e - 1) : 0;
    const uint32_t *cg = rows.cigar + rows.cigar_off[r];
    uint32_t nc = (uint32_t)(rows.cigar_off[r + 1] - rows.cigar_off[r]);
    uint32_t qa = 0;
    for (uint32_t k = 0; k < nc; k++) {
      uint32_t op = cg[k] & 0xf;
      if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_MATCH_OVR || op == OP_INS_OVR) qa += cg[k] >> 4;
    }
    p.query_aligned_len = qa;
    const size_t bi = (size_t)rows.input_index[r];
    // api.rs:453 <- evaluate.rs:1062: the transcript's strand differs from the read's INFERRED strand ('.' for a read
    // without XS / ts: then true on either strand).  The C++ AlignInfo::is_reverse is never assigned (include/evaluate.h:157);
    // what the C++ path acts on is the transcript strand (src/bam.cpp:549-553): transcript_strand below.
    p.transcript_strand = (char)rows.strand[r];
    p.is_reverse = p.transcript_strand != read_strand[bi];
    p.similarity_score = rows.similarity_score[r];
    p.nh = rows.nh[r]; p.hi = rows.hi[r]; p.is_primary = rows.is_primary[r];
    p.same_transcript_as_mate = rows.same_transcript_as_mate[r]; p.is_paired_out = rows.is_paired[r];
    p.insert_size = rows.insert_size[r]; p.input_index = kept[bi];
    p.mapq = rows.mapq[r]; p.cigar = cg; p.n_cigar = nc;
  }
  *out = c->h_proj.data(); *n_out = c->h_proj.size();
  return BR_OK;
}

extern "C" int br_project_group(br_ctx *c, const br_config *cfg, const br_alignment *alns, size_t n,
                                const br_projected **out, size_t *n_out) {
  return project_groups_impl(c, cfg, alns, n, true, out, n_out);
}

// Many read-name groups per call (name-collated: each query name one contiguous run of `alns`): what a caller that holds
// batches of groups (bramble-cli batches 64, bramble-cli/src/pipeline.rs:29) should use -- one trip through the device
// pipeline instead of one per group.  NH / HI / primary are per query name, as in the per-group call.
extern "C" int br_project_groups(br_ctx *c, const br_config *cfg, const br_alignment *alns, size_t n,
                                 const br_projected **out, size_t *n_out) {
  return project_groups_impl(c, cfg, alns, n, false, out, n_out);
}

extern "C" uint32_t br_primary_pick(const char *name, size_t len, uint32_t n_tied) {
  return n_tied ? br::primary_pick((const uint8_t *)name, len, n_tied) : 0;
}

// Diagnostic: the -S rescue DP alone.  Runs k_ksw on n (target, query) pairs as right-side problems and returns, per
// pair, whether the rescue is accepted, the maximum, and the raw traceback CIGAR (forward order, BAM-packed M / I / D).
extern "C" int br_ctx_ksw_pairs(br_ctx *c, int64_t n, const char *const *tseq, const char *const *qseq, int32_t *ok,
                                int32_t *max, uint32_t *n_cigar, uint32_t *cigar, uint32_t cigar_cap) {
  if (!c || n < 0 || (n && (!tseq || !qseq || !ok || !max || !n_cigar || !cigar)) || !cigar_cap) return BR_ERR_INVALID_ARG;
  if (n == 0) return BR_OK;
  HIPCHK(hipSetDevice(c->ix->device));
  hipStream_t st = nullptr;
  struct HProb { uint32_t qlen, tlen, side, pad; uint64_t seq_off; };
  if (ksw_prob_bytes() != sizeof(HProb)) return BR_ERR_UNSUPPORTED;
  std::vector<HProb> probs((size_t)n);
  std::vector<uint8_t> arena;
  auto code = [](char ch) -> uint8_t { switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } };
  uint64_t qmax = 0, tmaxv = 0;
  for (int64_t p = 0; p < n; p++) {
    size_t ql = strlen(qseq[p]), tl = strlen(tseq[p]);
    probs[(size_t)p] = HProb{(uint32_t)ql, (uint32_t)tl, 1u, 0u, (uint64_t)arena.size()};
    for (size_t k = 0; k < ql; k++) arena.push_back(code(qseq[p][k]));
    for (size_t k = 0; k < tl; k++) { uint8_t cd = code(tseq[p][k]); probs[(size_t)p].pad |= cd >> 2; arena.push_back(cd); }
    qmax = std::max<uint64_t>(qmax, ql); tmaxv = std::max<uint64_t>(tmaxv, tl);
  }
  DevBuf d_probs, d_res, d_arena, d_ops, d_raw, d_rawn, d_max;
  auto cleanup = [&]() { d_probs.release(); d_res.release(); d_arena.release(); d_ops.release(); d_raw.release(); d_rawn.release(); d_max.release(); };
  int rc = BR_OK;
  struct HRes { int32_t ok, score, refc; uint32_t n_ops; };
  std::vector<HRes> res((size_t)n);
  do {
    if ((rc = d_probs.ensure((size_t)n * sizeof(HProb))) || (rc = d_res.ensure((size_t)n * ksw_res_bytes())) ||
        (rc = d_arena.ensure(arena.size() + 1024)) || (rc = d_ops.ensure((arena.size() + (size_t)n + 1) * 4)) ||
        (rc = d_raw.ensure((size_t)n * cigar_cap * 4)) || (rc = d_rawn.ensure((size_t)n * 4)) || (rc = d_max.ensure((size_t)n * 4))) break;
    KswRun R{};
    R.n_prob = n; R.probs = (const KswProb *)d_probs.p; R.results = (KswRes *)d_res.p; R.seq_arena = d_arena.as<uint8_t>();
    R.clip_ops = d_ops.as<uint32_t>(); R.seq_total = arena.size(); R.qmax = qmax; R.tmax = tmaxv; R.stats = nullptr;
    R.raw_out = d_raw.as<uint32_t>(); R.raw_n = d_rawn.as<uint32_t>(); R.max_out = d_max.as<int32_t>(); R.raw_cap = cigar_cap;
    if (hipMemsetAsync(d_max.p, 0, (size_t)n * 4, st) != hipSuccess || hipMemsetAsync(d_rawn.p, 0, (size_t)n * 4, st) != hipSuccess ||
        hipMemcpyAsync(d_probs.p, probs.data(), (size_t)n * sizeof(HProb), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_arena.p, arena.data(), arena.size(), hipMemcpyHostToDevice, st) != hipSuccess) { rc = BR_ERR_HIP; break; }
    if ((rc = run_ksw(c, st, R))) break;
    if (hipMemcpyAsync(res.data(), d_res.p, (size_t)n * sizeof(HRes), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(max, d_max.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(n_cigar, d_rawn.p, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(cigar, d_raw.p, (size_t)n * cigar_cap * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { rc = BR_ERR_HIP; break; }
    for (int64_t p = 0; p < n; p++) ok[p] = res[(size_t)p].ok;
  } while (0);
  cleanup();
  return rc;
}

extern "C" uint32_t br_row_mapq(uint32_t nh, int long_reads) {  // src/core.cpp:46-58
  if (!long_reads) return nh == 1 ? 255u : nh == 2 ? 3u : (nh == 3 || nh == 4) ? 1u : 0u;
  return nh > 1 ? 0u : 3u;
}

// First touch of a device (runtime start-up, context creation: a few tenths of a second) -- something a caller can do on a
// thread of its own while it is busy elsewhere (the command line does, while the guides are parsed).
extern "C" int br_device_warmup(int device) {
  int rc = check_device(device);
  if (rc) return rc;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipFree(nullptr));
  return BR_OK;
}

extern "C" const char *br_version(void) { return "bramble_amd 0.3.0 (gfx950, ABI 3)"; }
extern "C" const char *br_strerror(int code) {
  switch (code) {
    case BR_OK: return "ok";
    case BR_ERR_INVALID_ARG: return "invalid argument";
    case BR_ERR_NO_DEVICE: return "no usable HIP device (the projection path has no CPU fallback)";
    case BR_ERR_HIP: return "HIP runtime error";
    case BR_ERR_ANNOTATION: return "invalid annotation";
    case BR_ERR_CAPACITY: return "batch exceeds 32-bit device offsets; split it";
    case BR_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
