// The bramble command line on top of the C ABI (scope table rows f-1 / f-3 / f-4): same flags as the
// reference's CLI11 front-end (src/bramble.cpp:443-485), same output header layout
// (src/bramble.cpp:513-623), same final report (src/bramble.cpp:727-736).
//
//   reader thread : BGZF inflate (threaded) -> record boundaries -> bundles cut at a read-name change
//                   (process_reads, src/bramble.cpp:330-441; a bundle here is millions of records, the
//                   result does not depend on where a name-collated stream is cut)
//   uploader      : br_bam_bundle_stage (records to one of three device slots, own copy stream)
//   main thread   : br_project_bam_staged (everything between the raw records on the device)
//   writer thread : BGZF deflate (threaded) -> output file
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/bramble_amd.h"
#include "bgzf.h"

#define BRAMBLE_REF_VERSION "0.1.6"  // src/bramble.cpp:35

namespace {

using brio::BgzfReader;
using brio::BgzfWriter;

struct Options {
  std::string in_bam, out_bam, gff, fasta;
  br_config cfg;
  int threads = 1, level = 6;
  std::vector<int> devices{0};  // --device N / --devices a,b,...: one worker (index replica + context + host threads) per entry
  int64_t bundle_records = 1000000;   // (1 M: 1.20 s inside the program for 20.9 M alignments, 2 M: 1.38 s, 0.5 M: 1.47 s; the pinned result buffers scale with it)
  bool quiet = false;
  bool device_deflate = true;   // BGZF blocks made on the GPU unless a host level is asked for
  int device_reader = -1;       // inflate + record split on the GPU (br_bam_reader): -1 = when the input is a regular file and one device is used
};

void usage(FILE *f) {
  fprintf(f,
          "bramble (MI355X) usage:\n\n"
          "bramble <in.bam> -G <annotation.gtf> -o <out.bam> [-p <cpus>] [-S <genome.fa>]\n"
          " [--help] [--version] [--quiet] [--fr] [--rf] [--lr] [--lr-hq] [--strict]\n"
          " [--max-soft-clip N] [--max-junction-insertion N] [--max-junction-deletion N]\n"
          " [--max-error-exon N] [--similarity-threshold X]\n"
          " [--device-deflate | --host-deflate | --compression-level 0-9] [--device-reader | --host-reader] [--bundle-size N]\n"
          "               [--device N | --devices a,b,...]\n\n"
          "Project spliced genomic alignments into transcriptomic space.\n"
          "The output BGZF blocks are deflated on the GPU by default (per-block Huffman codes); --host-deflate or\n"
          "--compression-level N use the host codec (libdeflate / zlib, level 6 like the reference unless N is given).\n"
          "--devices 0,1,...: bundles are dealt to one worker per listed GPU (an index replica each, no exchange between them);\n"
          "the output keeps the input order.\n");
}

bool parse_u32(const char *s, uint32_t &v) { char *e; unsigned long x = strtoul(s, &e, 10); if (e == s || *e) return false; v = (uint32_t)x; return true; }

// returns 0 to continue, 1 to exit(0), <0 on error
int parse_args(int argc, char **argv, Options &o) {
  memset(&o.cfg, 0, sizeof(o.cfg));
  o.cfg.junc_miss_discount = 1.0;
  auto need = [&](int &i) -> const char * { if (i + 1 >= argc) { fprintf(stderr, "%s: missing value\n", argv[i]); return nullptr; } return argv[++i]; };
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    std::string val; bool has_eq = false;
    if (a.rfind("--", 0) == 0) { size_t eq = a.find('='); if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); has_eq = true; } }
    auto value = [&]() -> const char * { if (has_eq) return val.c_str(); return need(i); };
    if (a == "--help" || a == "-h") { usage(stdout); return 1; }
    else if (a == "--version" || a == "-V") { printf("version: %s (bramble_amd %s)\n", BRAMBLE_REF_VERSION, br_version()); return 1; }
    else if (a == "--quiet" || a == "-q") o.quiet = true;          // -q: bramble-cli/src/cli.rs:60
    else if (a == "--fr") o.cfg.fr = 1;
    else if (a == "--rf") o.cfg.rf = 1;
    else if (a == "--lr") o.cfg.lr = 1;
    else if (a == "--lr-hq" || a == "--lr:hq") o.cfg.lr_hq = 1;    // Rust spelling: bramble-cli/src/cli.rs:32
    else if (a == "--unordered") {}                                // bramble-cli/src/cli.rs:64: the order here is always the input order
    else if (a == "--unordered-flush-records") { if (!value()) return -1; }
    else if (a == "--strict") o.cfg.strict = 1;
    else if (a == "--max-soft-clip") { const char *v = value(); if (!v || !parse_u32(v, o.cfg.max_clip)) return -1; o.cfg.has_max_clip = 1; }
    else if (a == "--max-junction-insertion") { const char *v = value(); if (!v || !parse_u32(v, o.cfg.max_junc_ins)) return -1; o.cfg.has_max_junc_ins = 1; }
    else if (a == "--max-junction-deletion") { const char *v = value(); if (!v || !parse_u32(v, o.cfg.max_junc_gap)) return -1; o.cfg.has_max_junc_gap = 1; }
    else if (a == "--max-error-exon") { const char *v = value(); if (!v || !parse_u32(v, o.cfg.max_error_exon)) return -1; o.cfg.has_max_error_exon = 1; }
    else if (a == "--similarity-threshold") { const char *v = value(); if (!v) return -1; o.cfg.sim_thr = strtof(v, nullptr); o.cfg.has_sim_thr = 1; }
    else if (a == "-G" || a == "--guide") { const char *v = value(); if (!v) return -1; o.gff = v; }
    else if (a == "-S" || a == "--genome") { const char *v = value(); if (!v) return -1; o.fasta = v; }
    else if (a == "-o" || a == "--out") { const char *v = value(); if (!v) return -1; o.out_bam = v; }
    else if (a == "-p" || a == "--threads") { const char *v = value(); if (!v) return -1; o.threads = atoi(v); if (o.threads < 1) o.threads = 1; }
    else if (a == "--compression-level") { const char *v = value(); if (!v) return -1; o.level = atoi(v); if (o.level < 0 || o.level > 9) return -1; o.device_deflate = false; }
    else if (a == "--host-deflate") o.device_deflate = false;
    else if (a == "--bundle-size") { const char *v = value(); if (!v) return -1; o.bundle_records = atoll(v); if (o.bundle_records < 1) return -1; }
    else if (a == "--device-deflate") o.device_deflate = true;
    else if (a == "--device-reader") o.device_reader = 1;
    else if (a == "--host-reader") o.device_reader = 0;
    else if (a == "--device") { const char *v = value(); if (!v) return -1; o.devices.assign(1, atoi(v)); }
    else if (a == "--devices") {
      const char *v = value(); if (!v) return -1;
      o.devices.clear();
      for (const char *q = v; *q;) { char *e; long d = strtol(q, &e, 10); if (e == q || d < 0) return -1; o.devices.push_back((int)d); q = *e == ',' ? e + 1 : e; if (*e && *e != ',') return -1; }
      if (o.devices.empty() || o.devices.size() > 64) return -1;
    }
    else if (!a.empty() && a[0] == '-' && a != "-") { fprintf(stderr, "unknown option %s\n", a.c_str()); return -1; }
    else if (o.in_bam.empty()) o.in_bam = a;
    else { fprintf(stderr, "unexpected argument %s\n", a.c_str()); return -1; }
  }
  if (o.in_bam.empty()) { fprintf(stderr, "in.bam is required\n"); return -1; }
  if (o.out_bam.empty()) { fprintf(stderr, "--out is required\n"); return -1; }
  if (o.gff.empty()) { fprintf(stderr, "--guide is required\n"); return -1; }
  if (!o.fasta.empty()) o.cfg.use_fasta = 1;
  return 0;
}

// ---- FASTA (plain or gzip): name = first word of the '>' line ---------------------------------
struct Fasta { std::vector<std::string> names, seqs; };
bool load_fasta(const char *path, Fasta &fa) {
  gzFile f = gzopen(path, "rb");
  if (!f) return false;
  gzbuffer(f, 1 << 20);
  std::vector<char> buf(1 << 16);
  while (gzgets(f, buf.data(), (int)buf.size())) {
    char *s = buf.data();
    size_t n = strlen(s);
    bool full = n && s[n - 1] == '\n';
    while (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) n--;
    if (s[0] == '>') {
      size_t e = 1; while (e < n && s[e] != ' ' && s[e] != '\t') e++;
      fa.names.emplace_back(s + 1, e - 1); fa.seqs.emplace_back();
      while (!full && gzgets(f, buf.data(), (int)buf.size())) { size_t m = strlen(buf.data()); full = m && buf[m - 1] == '\n'; }
    } else if (!fa.seqs.empty()) fa.seqs.back().append(s, n);
  }
  gzclose(f);
  return true;
}

// ---- BAM header ---------------------------------------------------------------------------------
struct BamHeader { std::string text; std::vector<std::string> ref_names; std::vector<uint32_t> ref_lens; };

// consumes the header from the front of `buf` (reading more as needed); false on a malformed file
bool read_header(BgzfReader &rd, brio::ByteBuf &buf, size_t &pos, BamHeader &h, std::string &err) {
  auto need = [&](size_t n) -> bool {
    while (buf.size() - pos < n) { int64_t got = rd.read(buf, 1 << 20); if (got < 0) { err = rd.error(); return false; } if (got == 0) { err = "truncated BAM header"; return false; } }
    return true;
  };
  auto u32 = [&](size_t at) { uint32_t v; memcpy(&v, buf.data() + at, 4); return v; };
  if (!need(12)) return false;
  if (memcmp(buf.data() + pos, "BAM\1", 4) != 0) { err = "not a BAM file (bad magic)"; return false; }
  uint32_t l_text = u32(pos + 4);
  if (!need(12 + (size_t)l_text)) return false;
  h.text.assign((const char *)buf.data() + pos + 8, l_text);
  while (!h.text.empty() && h.text.back() == '\0') h.text.pop_back();
  size_t p = pos + 8 + l_text;
  uint32_t n_ref = u32(p); p += 4;
  for (uint32_t r = 0; r < n_ref; r++) {
    if (!need(p - pos + 4)) return false;
    uint32_t l_name = u32(p); p += 4;
    if (!need(p - pos + l_name + 4)) return false;
    h.ref_names.emplace_back((const char *)buf.data() + p, l_name ? l_name - 1 : 0); p += l_name;
    h.ref_lens.push_back(u32(p)); p += 4;
  }
  pos = p;
  return true;
}

// src/bramble.cpp:513-623: @HD first, one @SQ per transcript in guide order, then every other input
// line except @SQ / @HD (with the new @PG appended the way sam_hdr_add_pg chains it), then the @CO line.
std::string make_header_text(const std::string &in_text, const br_index *ix, const std::string &cl, const std::string &gff) {
  std::vector<std::string> lines;
  for (size_t a = 0; a < in_text.size();) { size_t b = in_text.find('\n', a); if (b == std::string::npos) b = in_text.size(); if (b > a) lines.emplace_back(in_text, a, b - a); a = b + 1; }
  std::string out;
  for (auto &l : lines) if (l.compare(0, 3, "@HD") == 0) { out += l; out += '\n'; }
  size_t nt = br_index_num_transcripts(ix);
  for (size_t t = 0; t < nt; t++) {
    int64_t len = br_index_transcript_len(ix, (uint32_t)t);
    if (len > 0) { out += "@SQ\tSN:"; out += br_index_transcript_name(ix, (uint32_t)t); out += "\tLN:"; out += std::to_string(len); out += '\n'; }
  }
  // @PG chain ends: ids no other @PG names as its PP (htslib sam_hdr_add_pg links the new record to each)
  std::vector<std::string> pg_ids, pg_pp;
  auto field = [](const std::string &l, const char *key) -> std::string {
    size_t p = 0;
    while ((p = l.find('\t', p)) != std::string::npos) { p++; if (l.compare(p, 3, key) == 0) { size_t e = l.find('\t', p); return l.substr(p + 3, e == std::string::npos ? std::string::npos : e - p - 3); } }
    return "";
  };
  for (auto &l : lines) if (l.compare(0, 3, "@PG") == 0) { pg_ids.push_back(field(l, "ID:")); pg_pp.push_back(field(l, "PP:")); }
  std::vector<std::string> ends;
  for (auto &id : pg_ids) { bool used = false; for (auto &pp : pg_pp) if (pp == id) used = true; if (!used && !id.empty()) ends.push_back(id); }
  for (auto &l : lines) if (l.compare(0, 3, "@SQ") != 0 && l.compare(0, 3, "@HD") != 0) { out += l; out += '\n'; }
  auto unique_id = [&](int &serial) { for (;;) { std::string id = serial ? "bramble." + std::to_string(serial) : "bramble"; serial++; bool clash = false; for (auto &x : pg_ids) if (x == id) clash = true; if (!clash) { pg_ids.push_back(id); return id; } } };
  int serial = 0;
  auto pg_line = [&](const std::string &pp) {
    std::string l = "@PG\tID:" + unique_id(serial) + "\tPN:bramble";
    if (!pp.empty()) l += "\tPP:" + pp;
    l += "\tVN:" BRAMBLE_REF_VERSION "+amd." + std::string(br_version()) + "\tCL:" + cl + "\n";
    return l;
  };
  if (ends.empty()) out += pg_line("");
  else for (auto &e : ends) out += pg_line(e);
  out += "@CO\tGenerated from GTF: " + gff + "\n";
  return out;
}

std::vector<uint8_t> make_bam_header(const std::string &text, const br_index *ix) {
  std::vector<uint8_t> o;
  auto p32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) o.push_back((uint8_t)(v >> (8 * k))); };
  o.insert(o.end(), {'B', 'A', 'M', 1});
  p32((uint32_t)text.size()); o.insert(o.end(), text.begin(), text.end());
  size_t nt = br_index_num_transcripts(ix);
  uint32_t n_sq = 0;
  for (size_t t = 0; t < nt; t++) if (br_index_transcript_len(ix, (uint32_t)t) > 0) n_sq++;
  p32(n_sq);
  for (size_t t = 0; t < nt; t++) {
    int64_t len = br_index_transcript_len(ix, (uint32_t)t);
    if (len <= 0) continue;
    const char *nm = br_index_transcript_name(ix, (uint32_t)t);
    uint32_t l = (uint32_t)strlen(nm) + 1;
    p32(l); o.insert(o.end(), nm, nm + l); p32((uint32_t)len);
  }
  return o;
}

// ---- bounded single-slot hand-off between pipeline stages ---------------------------------------
template <typename T>
struct Slot {  // bounded FIFO between two pipeline stages
  explicit Slot(size_t depth = 1) : depth_(depth) {}
  std::mutex m; std::condition_variable cv; std::deque<std::unique_ptr<T>> q; bool done = false; size_t depth_;
  void put(std::unique_ptr<T> v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < depth_; }); q.push_back(std::move(v)); cv.notify_all(); }
  void finish() { std::unique_lock<std::mutex> l(m); done = true; cv.notify_all(); }
  std::unique_ptr<T> take() {
    std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || done; });
    if (q.empty()) return nullptr;
    auto v = std::move(q.front()); q.pop_front(); cv.notify_all(); return v;
  }
  // consumer that keeps the (depth-1) slot occupied while it works on the item: put() of the next one waits for release()
  T *hold() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || done; }); return q.empty() ? nullptr : q.front().get(); }
  void release() { std::unique_lock<std::mutex> l(m); q.pop_front(); cv.notify_all(); }
};

struct Bundle { brio::ByteBuf blob; std::vector<uint64_t> off; std::vector<uint32_t> len; uint64_t seq = 0; };
struct OutChunk { const uint8_t *data; uint64_t n; int worker; };

const uint8_t *rec_name(const brio::ByteBuf &b, uint64_t off, uint32_t &l) { l = b[off + 8]; return b.data() + off + 32; }

}  // namespace

static std::atomic<int> g_exit_at_end{0};
extern "C" void br_cli_exit_at_end(int on) { g_exit_at_end.store(on ? 1 : 0); }

extern "C" int br_cli_main(int argc, char **argv) {
  Options o;
  int prc = parse_args(argc, argv, o);
  if (prc > 0) return 0;
  if (prc < 0) { usage(stderr); return 2; }
  std::string cl;
  for (int i = 0; i < argc; i++) { if (i) cl += ' '; cl += argv[i]; }
  auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  if (o.out_bam == "-") o.quiet = true;   // the BAM stream owns standard output
  if (!o.quiet) { printf("\n[bramble] starting version: %s (bramble_amd %s)\n", BRAMBLE_REF_VERSION, br_version()); printf("[bramble] loading reference annotation...\n"); }

  // the guide loader starts first, on a thread of its own: opening the input, reading its header and starting the reader
  // (below) happen beside it instead of in front of it
  br_annotation *ann = nullptr;
  std::future<int> ann_job = std::async(std::launch::async, [&]() { return br_annotation_load_mt(o.gff.c_str(), std::max(1, std::min(o.threads, 32)), &ann); });   // the lines are taken apart by -p threads
  struct AnnJoin { std::future<int> &f; br_annotation *&a; ~AnnJoin() { if (f.valid()) { (void)f.get(); if (a) br_annotation_free(a); a = nullptr; } } } ann_join{ann_job, ann};   // (an early return: wait for it, drop its result)
  // the devices' first touch (runtime start-up, contexts) happens beside the guide parsing, not in front of the index build
  std::vector<std::thread> warm;
  for (int d : o.devices) warm.emplace_back([d]() { (void)br_device_warmup(d); });
  struct WarmJoin { std::vector<std::thread> &t; ~WarmJoin() { for (auto &x : t) if (x.joinable()) x.join(); } } warm_join{warm};
  BgzfReader rd;
  if (!rd.open(o.in_bam.c_str(), o.threads)) { fprintf(stderr, "error: %s\n", rd.error().c_str()); return 1; }
  brio::ByteBuf buf; size_t pos = 0;
  BamHeader hdr; std::string err;
  if (!read_header(rd, buf, pos, hdr, err)) { fprintf(stderr, "error: %s: %s\n", o.in_bam.c_str(), err.c_str()); return 1; }
  Slot<Bundle> to_gpu(16);     // the reader runs ahead while the guides are parsed and the indexes are built (sixteen bundles: about 3 GB of records)
  // consumed bundle buffers go back to the reader: their pages are already faulted in
  std::mutex pool_m; std::vector<std::unique_ptr<brio::ByteBuf>> pool;
  uint64_t total_reads = 0, unmapped_reads = 0, next_seq = 0;
  std::string reader_err, writer_err;
  std::atomic<bool> cancel{false};
  double t_inflate = 0, t_split = 0, t_copy = 0, t_deflate = 0, t_reserve = 0, t_put = 0, t_reader = 0;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

  // Device readers (br_bam_reader, piece-wise): the mapped file's bytes go to the GPUs as they are; inflate, the record split
  // and the cuts at read-name changes happen there, beside the guide parsing and the index build (they need neither), and the
  // bundles stay in the HBM of the device that made them until its runner has projected them.  The file's BGZF blocks are cut
  // into pieces of --bundle-size x 3 / 1000 blocks; piece k goes to device k mod N (reader k mod N inflates it, worker k mod N
  // projects it, the writer puts the results back in piece order), so N devices read, project and deflate N pieces at a
  // time -- nothing here is one host thread wide (the reference's one reader thread, src/bramble.cpp:329-435, feeds all its
  // workers).  Every piece cuts itself off at read-name changes by a rule both neighbours can evaluate (include/bramble_amd.h,
  // br_bam_piece_process); with one device the pieces follow each other and every start is known, with several a piece guesses
  // where its first record starts and the guess is checked against what the piece in front found: a piece that guessed wrong is
  // processed again with the true start before anything of it is used.
  struct DevBundle { br_device_records recs; int64_t id; uint64_t seq; };
  std::mutex out_m; std::condition_variable out_cv; std::map<uint64_t, OutChunk> out_map; uint64_t out_next = 0; bool out_done = false;   // the ordered writer's inbox
  if (o.device_reader < 0) if (const char *e = getenv("BRAMBLE_AMD_DEVICE_READER")) o.device_reader = atoi(e) != 0;   // (A/B with one command line: the @PG line quotes it)
  const bool use_dev_reader = o.device_reader != 0 && rd.mapped() && (o.device_reader > 0 || rd.mapped_size() >= (1u << 20));
  const size_t n_dev = o.devices.size();
  std::vector<std::unique_ptr<Slot<DevBundle>>> to_dev;
  std::vector<br_bam_reader *> dev_readers(n_dev, nullptr);
  for (size_t d = 0; d < n_dev; d++) to_dev.emplace_back(new Slot<DevBundle>(64));
  // the whole file's block table: it grows while the readers are already at work on its first pieces (a lazily committed
  // mapping of the worst-case size, so that the table never moves: a block is at least 28 bytes)
  struct BlockTable { br_bgzf_block *p = nullptr; size_t bytes = 0; ~BlockTable() { if (p) munmap(p, bytes); } } blk;
  int64_t n_blk = 0, n_pieces = 0;          // blocks known so far (under piece_m); pieces: known once the table is done
  const int64_t piece_blocks = std::max<int64_t>(1, std::min<int64_t>(o.bundle_records * 3 / 1000, 8192));
  std::mutex piece_m; std::condition_variable piece_cv;
  std::vector<uint64_t> piece_end; std::vector<char> piece_known;   // end_rel of every finished piece (the next one's true start)
  bool table_ready = false, table_failed = false;   // ready: the whole file has been walked
  std::atomic<uint64_t> total_reads_a{0}, unmapped_reads_a{0}, reprocessed{0};
  const int64_t piece_spoil = getenv("BRAMBLE_AMD_PIECE_SPOIL") ? atoll(getenv("BRAMBLE_AMD_PIECE_SPOIL")) : 0;   // test hook (tests/test_gpu_cli.py): every k-th guessed start counts as wrong
  std::mutex err_m;
  double t_dev_reader = 0, t_block_scan = 0;
  std::vector<std::thread> dev_threads;
  struct UpState { std::mutex m; std::condition_variable cv; int free_slots = 2; std::deque<int64_t> ready; bool done = false; };
  std::vector<std::unique_ptr<UpState>> ups;
  for (size_t d = 0; d < n_dev; d++) ups.emplace_back(new UpState());
  auto set_reader_err = [&](const std::string &m) { std::lock_guard<std::mutex> l(err_m); if (reader_err.empty()) reader_err = m; cancel = true; piece_cv.notify_all(); for (auto &u : ups) u->cv.notify_all(); };
  // blocks [b0, b1) of piece k and the `extra` blocks behind them; waits until the table has grown past them (or is whole).
  // false: no such piece (the table ended in front of it), or the run is being cancelled
  auto piece_range = [&](int64_t k, int64_t extra, int64_t &b0, int64_t &b1, int64_t &b1x, int64_t &nb_now) -> bool {
    std::unique_lock<std::mutex> l(piece_m);
    const int64_t want = (k + 1) * piece_blocks + extra;
    piece_cv.wait(l, [&] { return n_blk > want || table_ready || cancel; });
    if (cancel || table_failed) return false;
    nb_now = n_blk;
    b0 = k * piece_blocks;
    if (b0 >= n_blk) return false;
    b1 = std::min(n_blk, b0 + piece_blocks); b1x = std::min(n_blk, b1 + extra);
    return true;
  };
  if (use_dev_reader) {
    const uint8_t *file = rd.mapped(); const uint64_t fsize = rd.mapped_size();
    // the block table: one walk over the block headers of the mapping (a cache line per block), published as it grows
    blk.bytes = (size_t)(fsize / 28 + 16) * sizeof(br_bgzf_block);
    void *tab_mem = mmap(nullptr, blk.bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (tab_mem == MAP_FAILED) { fprintf(stderr, "error: out of memory (block table)\n"); return 1; }
    blk.p = (br_bgzf_block *)tab_mem;
    dev_threads.emplace_back([&, file, fsize]() {
      auto t0 = now();
      int64_t nb = 0; uint64_t src_base = 0, dst_base = 0;
      int rc2 = 0;
      const int64_t step = std::max<int64_t>(256, std::min<int64_t>(piece_blocks, 4096));
      while (!rc2 && src_base < fsize && !cancel) {
        int64_t got = 0; uint64_t used = 0, total = 0;
        rc2 = br_bgzf_scan(file + src_base, fsize - src_base, step, blk.p + nb, &got, &used, &total);
        if (rc2) break;
        for (int64_t i = 0; i < got; i++) { blk.p[nb + i].src_off += src_base; blk.p[nb + i].dst_off += dst_base; }
        if (used == 0) { rc2 = BR_ERR_INVALID_ARG; break; }   // a truncated block at the end of the file
        src_base += used; dst_base += total; nb += got;
        { std::lock_guard<std::mutex> l(piece_m); n_blk = nb; const size_t np = (size_t)((nb + piece_blocks - 1) / piece_blocks); piece_end.resize(np, 0); piece_known.resize(np, 0); }
        piece_cv.notify_all();
      }
      t_block_scan = secs(t0, now());
      {
        std::lock_guard<std::mutex> l(piece_m);
        n_blk = nb; n_pieces = nb ? (nb + piece_blocks - 1) / piece_blocks : 0;
        piece_end.resize((size_t)n_pieces, 0); piece_known.resize((size_t)n_pieces, 0);
        table_ready = true; table_failed = rc2 != 0;
      }
      if (rc2) set_reader_err(std::string("malformed or truncated BAM file (") + br_strerror(rc2) + ")");
      piece_cv.notify_all();
    });
    for (size_t d = 0; d < n_dev; d++) {
      // uploader of device d: the compressed bytes of its pieces, one piece ahead of the processing
      dev_threads.emplace_back([&, d, file, fsize]() {
        UpState &U = *ups[d];
        int rrc = cancel ? 0 : br_bam_reader_new(o.devices[d], (int32_t)hdr.ref_names.size(), (uint64_t)pos, &dev_readers[d]);
        if (rrc) set_reader_err(std::string("device reader: ") + br_strerror(rrc));
        int64_t j = 0;
        for (int64_t k = (int64_t)d; !rrc && !cancel; k += (int64_t)n_dev, j++) {
          int64_t b0, b1, b1x, nb_now;
          if (!piece_range(k, 2, b0, b1, b1x, nb_now)) break;
          { std::unique_lock<std::mutex> l(U.m); U.cv.wait(l, [&] { return U.free_slots > 0 || cancel; }); if (cancel) break; U.free_slots--; }
          rrc = br_bam_piece_upload(dev_readers[d], (int)(j & 1), file, fsize, blk.p, nb_now, b0, b1x);
          if (rrc) { set_reader_err(std::string("device reader: ") + br_strerror(rrc)); break; }
          { std::lock_guard<std::mutex> l(U.m); U.ready.push_back(k); }
          U.cv.notify_all();
        }
        { std::lock_guard<std::mutex> l(U.m); U.done = true; }
        U.cv.notify_all();
      });
      // processor of device d: inflate, split and cut its pieces; check a guessed start against the piece in front
      dev_threads.emplace_back([&, d, file, fsize]() {
        auto tr0 = now();
        UpState &U = *ups[d];
        int64_t j = 0;
        for (;; j++) {
          int64_t k = -1;
          { std::unique_lock<std::mutex> l(U.m); U.cv.wait(l, [&] { return !U.ready.empty() || U.done || cancel; }); if (!U.ready.empty()) { k = U.ready.front(); U.ready.pop_front(); } }
          if (k < 0) break;
          const int slot = (int)(j & 1);
          br_bam_reader *R = dev_readers[d];
          int64_t b0, b1, b1x, nb_now;
          if (!piece_range(k, 2, b0, b1, b1x, nb_now)) break;   // (the uploader has seen this range already: no waiting here)
          auto b = std::make_unique<DevBundle>();
          br_piece_info info; memset(&info, 0, sizeof(info));
          // the start: the header's end (first piece), the end of the piece in front when this reader made it itself, else a guess
          int64_t start_rel = -1;
          if (k == 0) start_rel = (int64_t)pos;
          else if (n_dev == 1) { std::lock_guard<std::mutex> l(piece_m); start_rel = (int64_t)piece_end[(size_t)k - 1]; }
          int rrc = 0;
          int64_t extra = 2;
          for (int tries = 0;; tries++) {
            rrc = cancel ? BR_ERR_INVALID_ARG : br_bam_piece_process(R, slot, blk.p, nb_now, b1, start_rel, &b->recs, &b->id, &info);
            if (rrc == BR_PIECE_MORE && tries < 12) {   // the group at the piece's end goes on: more of the next piece's blocks
              extra *= 8;
              if (!piece_range(k, extra, b0, b1, b1x, nb_now)) { rrc = BR_ERR_INVALID_ARG; break; }
              rrc = br_bam_piece_upload(R, slot, file, fsize, blk.p, nb_now, b0, b1x);
              if (!rrc) continue;
            }
            if (rrc) break;
            if (start_rel >= 0) break;
            // a guessed start: what did the piece in front find?
            uint64_t want = 0;
            { std::unique_lock<std::mutex> l(piece_m); piece_cv.wait(l, [&] { return piece_known[(size_t)k - 1] || cancel; }); want = piece_end[(size_t)k - 1]; }
            if (cancel) { rrc = BR_ERR_INVALID_ARG; break; }
            if (piece_spoil > 0 && k % piece_spoil == 0) info.start_rel ^= 1u;   // test hook: treat the guess as wrong
            if (info.start_rel == want) break;
            (void)br_bam_reader_release(R, b->id);    // the guess was wrong: once more, from the true start
            reprocessed++;
            start_rel = (int64_t)want;
          }
          if (rrc) { if (!cancel) set_reader_err(rrc == BR_PIECE_MORE ? std::string("a read-name group spans more than the reader can hold") : rrc == BR_ERR_INVALID_ARG ? std::string("malformed or truncated BAM file (") + br_strerror(rrc) + ")" : std::string("device reader: ") + br_strerror(rrc)); break; }
          { std::lock_guard<std::mutex> l(piece_m); piece_end[(size_t)k] = info.end_rel; piece_known[(size_t)k] = 1; }
          piece_cv.notify_all();
          { std::lock_guard<std::mutex> l(U.m); U.free_slots++; }
          U.cv.notify_all();
          total_reads_a += (uint64_t)(b->recs.n_aln + info.n_unmapped); unmapped_reads_a += (uint64_t)info.n_unmapped;
          b->seq = (uint64_t)k;
          if (b->recs.n_aln == 0) {   // nothing to project: the writer steps over this piece
            (void)br_bam_reader_release(R, b->id);
            { std::lock_guard<std::mutex> l(out_m); out_map[(uint64_t)k] = OutChunk{nullptr, 0, -1}; }
            out_cv.notify_all();
            continue;
          }
          to_dev[d]->put(std::move(b));
        }
        to_dev[d]->finish();
        const double t = secs(tr0, now());
        { std::lock_guard<std::mutex> l(err_m); t_dev_reader = std::max(t_dev_reader, t); }
      });
    }
  }
  std::thread reader = use_dev_reader ? std::thread([&]() {
    // (the device readers run on dev_threads; this thread only waits for the block table, for next_seq)
    std::unique_lock<std::mutex> l(piece_m); piece_cv.wait(l, [&] { return table_ready || cancel; });
    next_seq = (uint64_t)n_pieces;
  }) : std::thread([&]() {
    buf.erase_front(pos); pos = 0;
    std::vector<uint64_t> off; std::vector<uint32_t> len;
    bool eof = false;
    size_t scanned = 0;  // bytes of buf already split into off/len
    // the next chunk inflates (threaded, into the reserved tail of buf) while this thread walks the records of the
    // previous one; `valid` is how far the walker may look
    size_t valid = buf.size();
    const size_t CHUNK = 64u << 20;
    std::future<int64_t> fut; bool inflight = false;
    size_t bundle_bytes = 0;   // size of the largest bundle cut so far: the next buffer is reserved whole instead of growing chunk by chunk
    uint64_t split_bytes = 0, split_recs = 0;   // running mean record length: tells whether the bytes at hand already hold the next cut
    auto tr0 = now();
    struct ReaderClock { double &t; std::chrono::steady_clock::time_point t0; ~ReaderClock() { t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } reader_clock{t_reader, tr0};
    auto launch = [&]() {
      auto tv0 = now();
      const size_t need = buf.size() + CHUNK + (1u << 20);
      if (need > buf.capacity()) {   // only the first bundles get here (later buffers are reserved whole): one move, from the mean record length
        size_t est = split_recs ? (size_t)std::min<uint64_t>((split_bytes / split_recs + 1) * ((uint64_t)o.bundle_records + 64), (uint64_t)1 << 30) : 0;
        buf.reserve(std::max(std::max(need, est + 2 * CHUNK), std::max(bundle_bytes + 2 * CHUNK, buf.capacity() + buf.capacity() / 2)));
      }
      t_reserve += secs(tv0, now());
      fut = std::async(std::launch::async, [&]() { return rd.read(buf, CHUNK); }); inflight = true;
    };
    auto land = [&]() -> bool {
      auto ti0 = now();
      int64_t got = fut.get(); inflight = false;
      t_inflate += secs(ti0, now());
      if (got < 0) { reader_err = rd.error(); return false; }
      if (got == 0) eof = true;
      valid = buf.size();
      return true;
    };
    // do the bytes already inflated reach past the next cut?  Then they are split first and the next read starts in the NEXT
    // bundle's buffer (beside the copy of this one's tail) instead of landing behind the cut and being copied over with it.
    auto cut_expected = [&]() -> bool {
      if (!split_recs) return false;
      const uint64_t mean = split_bytes / split_recs + 1;
      return off.size() + (valid - scanned) / mean > (size_t)o.bundle_records + 64;
    };
    for (;;) {
      if (cancel) break;
      // split what is there; read more until a cut point exists
      int64_t cut = -1;
      size_t searched = std::max<size_t>((size_t)o.bundle_records, 1);  // records below this index cannot be a cut
      for (;;) {
        if (!eof && !inflight && !cut_expected()) launch();
        auto ts0 = now();
        for (;;) {
          // room for the records to come: from the mean record length (the worst case, 36 bytes a record, is a table six
          // times too large, value-initialised on every pass); a piece that fills its room is followed by another
          const size_t left = valid - scanned, worst = left / 36 + 1;
          size_t cap = split_recs ? std::min<size_t>(worst, (size_t)(left / (split_bytes / split_recs + 1)) * 5 / 4 + 4096) : worst;
          const size_t base = off.size();
          off.resize(base + cap); len.resize(base + cap);
          int64_t n = 0, un = 0; uint64_t used = 0;
          int r = br_bam_split(buf.data() + scanned, left, (int64_t)cap, off.data() + base, len.data() + base, &n, &un, &used);
          if (r) { if (inflight) (void)fut.get(); reader_err = "malformed BAM record"; to_gpu.finish(); return; }
          for (int64_t i = 0; i < n; i++) off[base + (size_t)i] += scanned;
          off.resize(base + (size_t)n); len.resize(base + (size_t)n);
          total_reads += (uint64_t)(n + un); unmapped_reads += (uint64_t)un;
          split_bytes += used; split_recs += (uint64_t)(n + un);
          scanned += used;
          if ((size_t)n < cap || used == 0) break;   // the bytes ran out (or end in a partial record), not the room
        }
        // cut: first record >= bundle_records whose name differs from its predecessor's
        for (size_t i = searched; i < off.size(); i++) {
          uint32_t la, lb; const uint8_t *a = rec_name(buf, off[i - 1], la), *b = rec_name(buf, off[i], lb);
          if (la != lb || memcmp(a, b, la) != 0) { cut = (int64_t)i; break; }
        }
        searched = std::max(searched, off.size());
        t_split += secs(ts0, now());
        if (cut >= 0) break;
        if (inflight) { if (!land()) { to_gpu.finish(); return; } continue; }
        if (!eof) { launch(); if (!land()) { to_gpu.finish(); return; } continue; }   // the estimate was short of the cut
        // end of stream, nothing in flight
        if (scanned != valid) { reader_err = "truncated BAM record at end of file"; to_gpu.finish(); return; }
        break;
      }
      if (inflight && !land()) { to_gpu.finish(); return; }   // the buffer must be still before its tail moves
      size_t n_take = cut >= 0 ? (size_t)cut : off.size();
      if (n_take) {
        auto tc0 = now();
        auto b = std::make_unique<Bundle>();
        size_t byte_end = (n_take < off.size()) ? (size_t)off[n_take] - 4 : scanned;
        b->off.assign(off.begin(), off.begin() + (ptrdiff_t)n_take); b->len.assign(len.begin(), len.begin() + (ptrdiff_t)n_take);
        // the bundle takes the buffer; only the tail (records past the cut, < one read chunk) is copied over
        brio::ByteBuf tail;
        { std::lock_guard<std::mutex> l(pool_m); if (!pool.empty()) { tail.swap(*pool.back()); pool.pop_back(); } }
        bundle_bytes = std::max(bundle_bytes, byte_end);
        tail.clear(); tail.reserve(bundle_bytes + 2 * CHUNK);   // whole, while it is empty: growing it later moves the mapping
        const size_t tail_bytes = buf.size() - byte_end;
        tail.resize(tail_bytes);
        buf.resize(byte_end);
        b->blob.swap(buf);
        buf.swap(tail);
        valid = tail_bytes;
        scanned -= byte_end;
        off.erase(off.begin(), off.begin() + (ptrdiff_t)n_take); len.erase(len.begin(), len.begin() + (ptrdiff_t)n_take);   // (the tables keep their capacity)
        for (auto &x : off) x -= byte_end;
        // the next read lands behind the tail's place in the new buffer while the tail itself is still on its way there
        if (!eof && !cut_expected()) launch();
        if (tail_bytes) memcpy(buf.data(), b->blob.data() + byte_end, tail_bytes);   // (launch() may have moved the buffer; the read itself never does)
        t_copy += secs(tc0, now());
        b->seq = next_seq++;   // the writer restores this order whatever worker projects the bundle
        auto tp0 = now();
        to_gpu.put(std::move(b));
        t_put += secs(tp0, now());
      }
      if (cut < 0 && eof && !inflight) break;
    }
    if (inflight) (void)fut.get();
    to_gpu.finish();
  });

  // the reader is already inflating while the guides are parsed and the indexes are built
  // (only the queues the running reader feeds are ever finished: taking from the others would wait for ever)
  auto join_dev_threads = [&]() { for (auto &t : dev_threads) if (t.joinable()) t.join(); };
  auto free_dev_readers = [&]() { for (auto &r : dev_readers) { if (r) br_bam_reader_free(r); r = nullptr; } };
  auto stop_reader = [&]() -> int {
    cancel = true;
    piece_cv.notify_all(); for (auto &u : ups) u->cv.notify_all();
    if (use_dev_reader) { for (auto &q : to_dev) while (q->take()) {} } else { while (to_gpu.take()) {} }
    reader.join(); join_dev_threads(); free_dev_readers();
    return 1;
  };
  int rc = ann_job.get();
  const double t_guides = since();
  if (rc) { fprintf(stderr, "error: could not load reference annotation %s: %s\n", o.gff.c_str(), br_strerror(rc)); return stop_reader(); }
  size_t n_tx = br_annotation_num_transcripts(ann), n_refs = br_annotation_num_refs(ann);
  const char *const *refnames = br_annotation_refnames(ann);
  Fasta fa;
  std::vector<br_fasta_seq> fseqs;
  if (o.cfg.use_fasta) {
    if (!load_fasta(o.fasta.c_str(), fa)) { fprintf(stderr, "error: could not open genome %s\n", o.fasta.c_str()); br_annotation_free(ann); return stop_reader(); }
    for (size_t i = 0; i < fa.names.size(); i++) fseqs.push_back({fa.names[i].c_str(), fa.seqs[i].data(), fa.seqs[i].size()});
  }
  if (!o.quiet) {
    printf("[bramble] reference annotation loaded! %zu unique transcripts were found (%.1fs)\n", n_tx, since());
    if (o.cfg.lr) printf("[bramble] using long-read mode (--lr)\n");
    else if (o.cfg.lr_hq) printf("[bramble] using long-read mode (--lr-hq)\n");
    else printf("[bramble] using short-read mode (have long reads? try running with --lr or --lr-hq)\n");
    printf("[bramble] building g2t index%s\n", o.devices.size() > 1 ? " (one replica per device)" : "");
  }

  // One worker per listed device: its own index replica and context (the reference's workers share one read-only tree,
  // src/threads.cpp:114-162; here every GPU holds a copy), an uploader thread and a projecting thread.  Workers take
  // bundles from the reader's queue as they become free -- no exchange between them; the writer restores bundle order
  // (bramble-cli/src/pipeline.rs:226-240 keeps a BTreeMap for the same purpose).
  struct Staged { std::unique_ptr<Bundle> b; int slot; int rc; };
  struct Worker {
    int id = 0, device = 0;
    br_index *ix = nullptr; br_ctx *ctx = nullptr;
    int build_rc = 0;
    std::unique_ptr<Slot<Staged>> to_main;
    std::mutex permit_m; std::condition_variable permit_cv; int permits = 3;   // three device staging slots
    std::mutex done_m; std::condition_variable done_cv; uint64_t produced = 0, written = 0;  // chunks handed to / finished by the writer
    std::thread uploader, runner;
    uint64_t total_complete = 0, total_unique = 0, dropped = 0, n_bundles = 0;
    double gpu_seconds = 0, t_upload = 0, t_wait_in = 0;
  };
  const size_t n_workers = o.devices.size();
  std::vector<std::unique_ptr<Worker>> workers;
  for (size_t w = 0; w < n_workers; w++) { workers.emplace_back(new Worker()); workers[w]->id = (int)w; workers[w]->device = o.devices[w]; workers[w]->to_main.reset(new Slot<Staged>(2)); }
  auto free_all = [&]() {
    for (auto &w : workers) { if (w->ctx) br_ctx_free(w->ctx); if (w->ix) br_index_free(w->ix); w->ctx = nullptr; w->ix = nullptr; }
    if (ann) br_annotation_free(ann);
    ann = nullptr;
  };
  {
    std::vector<std::thread> builders;
    for (auto &wp : workers) builders.emplace_back([&, w = wp.get()]() {
      w->build_rc = br_index_build(br_annotation_transcripts(ann), n_tx, refnames, n_refs, fseqs.empty() ? nullptr : fseqs.data(), fseqs.size(), w->device, &w->ix);
      if (!w->build_rc) w->build_rc = br_ctx_new(w->ix, &w->ctx);
    });
    for (auto &t : builders) t.join();
  }
  for (auto &w : workers)
    if (w->build_rc) { fprintf(stderr, "error: index build failed on device %d: %s\n", w->device, br_strerror(w->build_rc)); free_all(); return stop_reader(); }
  const double t_index = since() - t_guides;
  fa = Fasta();  // the indexes hold the exon sequences now
  const br_index *ix0 = workers[0]->ix;

  // input refID -> annotation reference index; names the annotation lacks get ids past its table
  // (gseqs.addName, src/bramble.cpp:384: a new id with no interval tree behind it)
  std::unordered_map<std::string, int32_t> ref_of;
  for (size_t r = 0; r < n_refs; r++) ref_of.emplace(refnames[r], (int32_t)r);
  std::vector<int32_t> ref_map(hdr.ref_names.size());
  int32_t extra = (int32_t)n_refs;
  for (size_t r = 0; r < hdr.ref_names.size(); r++) { auto it = ref_of.find(hdr.ref_names[r]); ref_map[r] = it != ref_of.end() ? it->second : extra++; }

  // the output goes to a temporary name next to the target and is renamed on success: a failed run leaves no file that
  // looks complete (a truncated stream with a valid EOF block)
  const bool to_stdout = o.out_bam == "-";
  const std::string out_tmp = to_stdout ? o.out_bam : o.out_bam + ".tmp-bramble";
  BgzfWriter wr;
  if (!wr.open(out_tmp.c_str(), o.threads, o.level)) { fprintf(stderr, "error: %s\n", wr.error().c_str()); free_all(); return stop_reader(); }
  {
    std::vector<uint8_t> h = make_bam_header(make_header_text(hdr.text, ix0, cl, o.gff), ix0);
    if (!wr.write(h.data(), h.size())) { fprintf(stderr, "error: %s\n", wr.error().c_str()); wr.abandon(); if (!to_stdout) remove(out_tmp.c_str()); free_all(); return stop_reader(); }
  }
  if (!o.quiet) printf("[bramble] processing alignments :-)\n");
  double t_setup = since();

  // One failure anywhere stops every runner.  The flag flips under each worker's done_m before its condition variable is
  // notified: a runner that has just evaluated the wait predicate as false holds that mutex until it blocks, so the
  // notification cannot fall between its check and its wait (a lost wakeup would leave it -- and the join below -- hanging).
  std::atomic<int> fail{0};
  auto raise_fail = [&]() {
    for (auto &x : workers) { std::lock_guard<std::mutex> l(x->done_m); fail = 1; }
    cancel = true;   // the reader stops making bundles nobody will project (the runners keep draining what is queued)
    for (auto &x : workers) x->done_cv.notify_all();
  };

  // ordered writer: chunks arrive tagged with their bundle's sequence number
  std::thread writer([&]() {
    for (;;) {
      OutChunk c;
      {
        std::unique_lock<std::mutex> l(out_m);
        out_cv.wait(l, [&] { return out_map.count(out_next) || out_done; });
        auto it = out_map.find(out_next);
        if (it == out_map.end()) break;   // done, and the next chunk never came (a worker failed): stop here
        c = it->second; out_map.erase(it); out_next++;
      }
      auto td0 = now();
      if (c.n) {   // the chunk's bytes may still be on their way from the device
        br_host_bam hb; memset(&hb, 0, sizeof(hb)); hb.data = c.data; hb.n_bytes = c.n;
        int wrc = br_host_bam_wait(workers[(size_t)c.worker]->ctx, &hb);
        if (wrc && writer_err.empty()) { writer_err = std::string("download failed: ") + br_strerror(wrc); raise_fail(); }
      }
      if (writer_err.empty() && c.n && !(o.device_deflate ? wr.write_raw(c.data, (size_t)c.n) : wr.write(c.data, (size_t)c.n))) {
        writer_err = wr.error();
        raise_fail();                      // nothing projected from here on could be written: the runners drain
      }
      t_deflate += secs(td0, now());
      if (c.worker < 0) continue;   // (a piece without records: nothing was produced for it)
      Worker *w = workers[(size_t)c.worker].get();
      { std::lock_guard<std::mutex> l(w->done_m); w->written++; }
      w->done_cv.notify_all();
    }
  });

  if (use_dev_reader) {
    // no uploaders: every worker's bundles are in its device's HBM already (its own reader made them)
    for (auto &wp : workers) {
      Worker *w = wp.get();
      w->runner = std::thread([&, w]() {
        Slot<DevBundle> &in = *to_dev[(size_t)w->id];
        for (;;) {
          auto tw0 = now();
          auto b = in.take();
          w->t_wait_in += secs(tw0, now());
          if (!b) break;
          br_host_bam hb;
          memset(&hb, 0, sizeof(hb));
          if (!fail) {
            { std::unique_lock<std::mutex> l(w->done_m); w->done_cv.wait(l, [&] { return w->written + 2 > w->produced || fail; }); }
            auto t0 = now();
            int prc2 = fail ? 0 : br_project_bam_resident(w->ctx, &o.cfg, &b->recs, ref_map.data(), (int32_t)ref_map.size(), o.device_deflate ? 1 : 0, 1, &hb);
            w->gpu_seconds += secs(t0, now());
            if (prc2) { fprintf(stderr, "error: projection failed on device %d: %s\n", w->device, br_strerror(prc2)); raise_fail(); }
          }
          (void)br_bam_reader_release(dev_readers[(size_t)w->id], b->id);
          if (fail) continue;  // drain
          w->total_complete += hb.total_complete; w->total_unique += hb.total_unique; w->dropped += hb.dropped_reads; w->n_bundles++;
          { std::lock_guard<std::mutex> l(w->done_m); w->produced++; }
          { std::lock_guard<std::mutex> l(out_m); out_map[b->seq] = OutChunk{hb.data, hb.n_bytes, w->id}; }
          out_cv.notify_all();
        }
      });
    }
  } else
  for (auto &wp : workers) {
    Worker *w = wp.get();
    // uploader: stages bundle k of this worker into device slot k % 3 on the context's copy stream while the runner
    // projects an earlier one; three permits = three slots, a permit returns when a slot's projection is done
    w->uploader = std::thread([&, w]() {
      int64_t k = 0;
      for (;;) {
        auto b = to_gpu.take();
        if (!b) break;
        { std::unique_lock<std::mutex> l(w->permit_m); w->permit_cv.wait(l, [&] { return w->permits > 0; }); w->permits--; }
        auto st = std::make_unique<Staged>();
        st->slot = (int)(k++ % 3);
        br_bam_bundle bb{b->blob.data(), b->blob.size(), b->off.data(), b->len.data(), (int64_t)b->off.size(), ref_map.data(), (int32_t)ref_map.size(), o.device_deflate ? 1 : 0};
        auto t0 = now();
        st->rc = fail ? 0 : br_bam_bundle_stage(w->ctx, &bb, st->slot);
        w->t_upload += secs(t0, now());
        st->b = std::move(b);
        w->to_main->put(std::move(st));
      }
      w->to_main->finish();
    });
    w->runner = std::thread([&, w]() {
      for (;;) {
        auto tw0 = now();
        auto st = w->to_main->take();
        w->t_wait_in += secs(tw0, now());
        if (!st) break;
        auto &b = st->b;
        if (!fail && st->rc) { fprintf(stderr, "error: upload failed on device %d: %s\n", w->device, br_strerror(st->rc)); raise_fail(); }
        br_host_bam hb;
        memset(&hb, 0, sizeof(hb));
        if (!fail) {
          // the context's two pinned result buffers alternate: chunk j - 2 of this worker must be on disk before call j
          { std::unique_lock<std::mutex> l(w->done_m); w->done_cv.wait(l, [&] { return w->written + 2 > w->produced || fail; }); }
          br_bam_bundle bb{b->blob.data(), b->blob.size(), b->off.data(), b->len.data(), (int64_t)b->off.size(), ref_map.data(), (int32_t)ref_map.size(), o.device_deflate ? 1 : 0};
          auto t0 = now();
          int prc2 = fail ? 0 : br_project_bam_staged_nowait(w->ctx, &o.cfg, &bb, st->slot, &hb);   // the writer waits for the bytes
          w->gpu_seconds += secs(t0, now());
          if (prc2) { fprintf(stderr, "error: projection failed on device %d: %s\n", w->device, br_strerror(prc2)); raise_fail(); }
        }
        const uint64_t seq = b->seq;
        { auto spare = std::make_unique<brio::ByteBuf>(); spare->swap(b->blob); std::lock_guard<std::mutex> l(pool_m); pool.push_back(std::move(spare)); }
        { std::lock_guard<std::mutex> l(w->permit_m); w->permits++; }
        w->permit_cv.notify_all();
        if (fail) continue;  // drain
        w->total_complete += hb.total_complete; w->total_unique += hb.total_unique; w->dropped += hb.dropped_reads; w->n_bundles++;
        { std::lock_guard<std::mutex> l(w->done_m); w->produced++; }
        { std::lock_guard<std::mutex> l(out_m); out_map[seq] = OutChunk{hb.data, hb.n_bytes, w->id}; }
        out_cv.notify_all();
      }
    });
  }
  for (auto &w : workers) { if (w->uploader.joinable()) w->uploader.join(); if (w->runner.joinable()) w->runner.join(); }
  { std::lock_guard<std::mutex> l(out_m); out_done = true; }
  out_cv.notify_all();
  reader.join(); join_dev_threads(); writer.join();
  if (use_dev_reader) { total_reads = total_reads_a.load(); unmapped_reads = unmapped_reads_a.load(); }
  int failed = fail.load();
  if (!reader_err.empty()) { fprintf(stderr, "error: %s: %s\n", o.in_bam.c_str(), reader_err.c_str()); failed = 1; }
  if (!writer_err.empty()) { fprintf(stderr, "error: %s: %s\n", o.out_bam.c_str(), writer_err.c_str()); failed = 1; }
  if (!failed && out_next != next_seq) { fprintf(stderr, "error: %s: output incomplete\n", o.out_bam.c_str()); failed = 1; }
  if (failed) {
    wr.abandon();                          // no EOF block: the stream must not look complete
    if (!to_stdout) remove(out_tmp.c_str());
  } else {
    if (!wr.close()) { fprintf(stderr, "error: %s: %s\n", o.out_bam.c_str(), wr.error().c_str()); failed = 1; if (!to_stdout) remove(out_tmp.c_str()); }
    else if (!to_stdout && rename(out_tmp.c_str(), o.out_bam.c_str()) != 0) { fprintf(stderr, "error: could not rename %s to %s\n", out_tmp.c_str(), o.out_bam.c_str()); failed = 1; }
  }
  double t_done = since();
  uint64_t total_complete = 0, total_unique = 0, dropped = 0, n_bundles = 0;
  double gpu_seconds = 0, t_upload = 0, t_wait_gpu_in = 0;
  for (auto &w : workers) {
    total_complete += w->total_complete; total_unique += w->total_unique; dropped += w->dropped; n_bundles += w->n_bundles;
    gpu_seconds += w->gpu_seconds; t_upload += w->t_upload; t_wait_gpu_in += w->t_wait_in;
  }
  // the command line's process ends here: handing tens of gigabytes of device and pinned memory back piece by piece is
  // 0.12-0.16 s that the process exit does for nothing (bramble-cli keeps its index in a ManuallyDrop for the same reason,
  // bramble-cli/src/main.rs:56-60).  That is the `bramble` binary (br_cli_exit_at_end); a host that calls br_cli_main as a
  // function gets everything released, and so does a run under BRAMBLE_AMD_CLI_CLEANUP=1
  if (!g_exit_at_end.load() || getenv("BRAMBLE_AMD_CLI_CLEANUP")) { free_all(); free_dev_readers(); }
  double t_freed = since();
  if (!o.quiet) {  // src/bramble.cpp:727-736
    printf("\n[bramble] final report:\n");
    printf("# input alignments:   %llu\n", (unsigned long long)total_reads);
    printf("# unmapped reads:     %llu\n", (unsigned long long)unmapped_reads);
    printf("# dropped alignments: %llu\n", (unsigned long long)dropped);
    printf("# total alignments:   %llu\n", (unsigned long long)total_complete);
    printf("# unique alignments:  %llu\n\n", (unsigned long long)total_unique);
    printf("[bramble] %llu bundles on %zu device worker(s), %.2fs on the device path (upload + kernels + download, summed), %.2fs wall (setup %.2fs: guides %.2fs + index %.2fs, codec %s)\n",
           (unsigned long long)n_bundles, n_workers, gpu_seconds, since(), t_setup, t_guides, t_index, brio::codec_name());
    printf("[bramble] release of device / pinned memory: %.2fs\n", t_freed - t_done);
    printf("[bramble] stage busy time: inflate %.2fs, split %.2fs, bundle copy %.2fs, upload %.2fs, device %.2fs (waited for input %.2fs), deflate+write %.2fs\n",
           t_inflate, t_split, t_copy, t_upload, gpu_seconds, t_wait_gpu_in, t_deflate);
  }
  if (getenv("BRAMBLE_AMD_TIMING")) {   // where the resident memory is: anonymous (record buffers), file, shared (pinned / device-visible)
    if (FILE *f = fopen("/proc/self/status", "r")) {
      char line[256];
      while (fgets(line, sizeof line, f)) if (!strncmp(line, "VmHWM", 5) || !strncmp(line, "VmRSS", 5) || !strncmp(line, "Rss", 3) || !strncmp(line, "AnonHuge", 8)) fprintf(stderr, "[bramble] %s", line);
      fclose(f);
    }
  }
  if (getenv("BRAMBLE_AMD_TIMING") && use_dev_reader) {
    double t_in = 0, t_up = 0;
    for (auto r : dev_readers) { t_in = std::max(t_in, br_bam_reader_seconds(r)); t_up = std::max(t_up, br_bam_reader_upload_seconds(r)); }
    fprintf(stderr, "[bramble] device readers: the compressed bytes went up in %.2fs of the longest uploader (%.1f GB/s of the file's %.2f GB; pinned buffers filled by four threads)\n",
            t_up, t_up > 0 ? 1e-9 * (double)rd.mapped_size() / (double)n_dev / t_up : 0.0, 1e-9 * (double)rd.mapped_size());
    fprintf(stderr, "[bramble] device readers: block table %.2fs; the longest processing thread %.2fs in all (inflate + record split + cuts: %.2fs; the rest: waiting for its uploads, its neighbour's cut, the runner's queue); %llu pieces, %llu processed again from the true start\n",
            t_block_scan, t_dev_reader, t_in, (unsigned long long)n_pieces, (unsigned long long)reprocessed.load());
  }
  if (getenv("BRAMBLE_AMD_TIMING") && !use_dev_reader) fprintf(stderr, "[bramble] reader thread: %.2fs in all, %.2fs reserving buffers, %.2fs waiting for a free queue slot\n", t_reader, t_reserve, t_put);
  // the unwinding below this line (record buffers, worker contexts, reader and writer pools) was 0.5 s of a 1.9 s run
  if (g_exit_at_end.load() && !getenv("BRAMBLE_AMD_CLI_CLEANUP")) {   // (tools that write their results from exit handlers -- a profiler -- ask for the clean return)
    if (getenv("BRAMBLE_AMD_TIMING")) { struct timespec t; clock_gettime(CLOCK_REALTIME, &t); fprintf(stderr, "[bramble] leaving at %.3f\n", (double)t.tv_sec + 1e-9 * (double)t.tv_nsec); }
    fflush(stdout); fflush(stderr); _exit(failed);
  }
  return failed;
}

// ---- BGZF utilities (host only) -------------------------------------------------------------------
extern "C" int br_bgzf_write_file(const char *path, const uint8_t *data, uint64_t n, int threads, int level) {
  if (!path || (!data && n)) return BR_ERR_INVALID_ARG;
  BgzfWriter wr;
  if (!wr.open(path, threads, level)) return BR_ERR_INVALID_ARG;
  if (n && !wr.write(data, (size_t)n)) return BR_ERR_INVALID_ARG;
  return wr.close() ? BR_OK : BR_ERR_INVALID_ARG;
}

extern "C" int br_bgzf_read_file(const char *path, int threads, uint8_t **out, uint64_t *n) {
  if (!path || !out || !n) return BR_ERR_INVALID_ARG;
  *out = nullptr; *n = 0;
  BgzfReader rd;
  if (!rd.open(path, threads)) return BR_ERR_INVALID_ARG;
  brio::ByteBuf buf;
  for (;;) { int64_t got = rd.read(buf, 64u << 20); if (got < 0) return BR_ERR_INVALID_ARG; if (got == 0) break; }
  uint8_t *p = (uint8_t *)malloc(buf.size() ? buf.size() : 1);
  if (!p) return BR_ERR_CAPACITY;
  memcpy(p, buf.data(), buf.size());
  *out = p; *n = buf.size();
  return BR_OK;
}

extern "C" void br_free_buffer(uint8_t *p) { free(p); }
extern "C" const char *br_bgzf_codec(void) { return brio::codec_name(); }
