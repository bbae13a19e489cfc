#include "bgzf.h"

#include <string.h>
#include <zlib.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>

#include <atomic>
#include <thread>

namespace brio {

// libdeflate (2-3x zlib's speed on 64 KiB blocks) when the shared object is on the box; the image ships
// libdeflate.so.0 without headers, so the five entry points are bound by name.  zlib otherwise.
namespace {
struct LibDeflate {
  void *(*alloc_c)(int) = nullptr;
  size_t (*compress)(void *, const void *, size_t, void *, size_t) = nullptr;
  void (*free_c)(void *) = nullptr;
  void *(*alloc_d)(void) = nullptr;
  int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
  void (*free_d)(void *) = nullptr;
  uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
  bool ok = false;
  LibDeflate() {
    if (getenv("BRAMBLE_AMD_NO_LIBDEFLATE")) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
    compress = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
    free_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
    alloc_d = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
    decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
    free_d = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
    crc = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
    ok = alloc_c && compress && free_c && alloc_d && decompress && free_d && crc;
  }
};
const LibDeflate &ld() { static LibDeflate l; return l; }
// one codec object per worker thread and level (libdeflate objects are not shareable)
struct TlsCodec {
  void *c = nullptr, *d = nullptr; int level = -100;
  ~TlsCodec() { if (c) ld().free_c(c); if (d) ld().free_d(d); }
  void *comp(int lv) { if (!c || lv != level) { if (c) ld().free_c(c); c = ld().alloc_c(lv); level = lv; } return c; }
  void *decomp() { if (!d) d = ld().alloc_d(); return d; }
};
thread_local TlsCodec tls_codec;
}  // namespace

const char *codec_name() { return ld().ok ? "libdeflate" : "zlib"; }

void parallel_for(size_t n, int threads, void (*fn)(size_t, void *), void *arg) {
  if (n == 0) return;
  if (threads <= 1 || n == 1) { for (size_t i = 0; i < n; i++) fn(i, arg); return; }
  std::atomic<size_t> next{0};
  auto body = [&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= n) break; fn(i, arg); } };
  size_t nt = std::min<size_t>((size_t)threads, n);
  std::vector<std::thread> th;
  for (size_t t = 1; t < nt; t++) th.emplace_back(body);
  body();
  for (auto &t : th) t.join();
}

static const size_t BLOCK_DATA = 0xff00;  // htslib BGZF_BLOCK_SIZE: uncompressed payload per block
static const uint8_t EOF_BLOCK[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0,
                                      0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

BgzfReader::~BgzfReader() { if (map_) munmap((void *)map_, map_size_); if (f_ && f_ != stdin) fclose(f_); }

bool BgzfReader::open(const char *path, int threads) {
  f_ = strcmp(path, "-") == 0 ? stdin : fopen(path, "rb");   // "-": standard input, like htslib
  threads_ = threads < 1 ? 1 : threads;
  if (!f_) { err_ = std::string("cannot open ") + path; return false; }
  // regular files are mapped: the inflate workers read the page cache directly (no fread copy on the reader thread)
  struct stat sb;
  if (fstat(fileno(f_), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fileno(f_), 0);
    if (m != MAP_FAILED) { map_ = (const uint8_t *)m; map_size_ = (size_t)sb.st_size; madvise(m, map_size_, MADV_SEQUENTIAL); }
  }
  if (!fill(18) || have() < 18 || cur()[0] != 0x1f || cur()[1] != 0x8b || !(cur()[3] & 4)) {
    err_ = std::string(path) + " is not a BGZF (BAM) file";
    return false;
  }
  return true;
}

// make at least `need` compressed bytes available at cpos_ (fewer at end of file)
bool BgzfReader::fill(size_t need) {
  if (map_) return have() >= need;
  if (cbuf_.size() - cpos_ >= need) return true;
  if (cpos_ > 0) { cbuf_.erase_front(cpos_); cpos_ = 0; }
  size_t chunk = std::max<size_t>(need, 16u << 20);
  size_t old = cbuf_.size();
  cbuf_.resize(old + chunk);
  size_t got = fread(cbuf_.data() + old, 1, chunk, f_);
  cbuf_.resize(old + got);
  return cbuf_.size() >= need;
}

struct InflateJob {
  const uint8_t *src; uint32_t clen;  // deflate payload
  uint8_t *dst; uint32_t ulen; uint32_t crc;
};
struct InflateCtx { std::vector<InflateJob> *jobs; std::atomic<int> bad{0}; };

static void inflate_one(size_t i, void *arg) {
  InflateCtx *c = (InflateCtx *)arg;
  InflateJob &j = (*c->jobs)[i];
  if (j.ulen == 0) return;
  if (ld().ok) {
    size_t got = 0;
    int rc = ld().decompress(tls_codec.decomp(), j.src, j.clen, j.dst, j.ulen, &got);
    if (rc != 0 || got != j.ulen || ld().crc(0, j.dst, j.ulen) != j.crc) c->bad = 1;
    return;
  }
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) { c->bad = 1; return; }
  zs.next_in = (Bytef *)j.src; zs.avail_in = j.clen; zs.next_out = j.dst; zs.avail_out = j.ulen;
  int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || zs.total_out != j.ulen) { c->bad = 1; return; }
  if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), j.dst, j.ulen) != j.crc) c->bad = 1;
}

int64_t BgzfReader::read(ByteBuf &out, size_t want) {
  if (eof_) return 0;
  std::vector<InflateJob> jobs;
  std::vector<size_t> src_off;  // offsets into cbuf_ (pointers are fixed up after the last fill)
  size_t total = 0, rel = 0;  // rel: offset of the next block relative to cpos_ (fill() may compact cbuf_)
  while (total < want) {
    fill(rel + 18);
    size_t avail = have();
    if (avail == rel) { eof_ = true; break; }
    if (avail - rel < 18) { err_ = "truncated BGZF block header"; return -1; }
    const uint8_t *h = cur() + rel;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err_ = "bad BGZF block magic"; return -1; }
    uint32_t xlen = h[10] | (h[11] << 8);
    fill(rel + 12 + xlen);
    if (have() - rel < 12 + (size_t)xlen) { err_ = "truncated BGZF extra field"; return -1; }
    h = cur() + rel;
    int64_t bsize = -1;
    for (uint32_t p = 0; p + 4 <= xlen;) {
      const uint8_t *x = h + 12 + p;
      uint32_t slen = x[2] | (x[3] << 8);
      if (x[0] == 'B' && x[1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = (x[4] | (x[5] << 8)) + 1;
      p += 4 + slen;
    }
    if (bsize < (int64_t)(12 + xlen + 8)) { err_ = "BGZF block without BC subfield"; return -1; }
    fill(rel + (size_t)bsize);
    if (have() - rel < (size_t)bsize) { err_ = "truncated BGZF block"; return -1; }
    h = cur() + rel;
    InflateJob j;
    j.clen = (uint32_t)(bsize - 12 - xlen - 8);
    const uint8_t *tail = h + bsize - 8;
    j.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    j.ulen = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    if (j.ulen > 65536) { err_ = "BGZF block larger than 64 KiB"; return -1; }
    j.src = nullptr; j.dst = nullptr;
    src_off.push_back(rel + 12 + xlen);
    jobs.push_back(j);
    total += j.ulen;
    rel += (size_t)bsize;
  }
  size_t base = out.size();
  out.resize(base + total);
  size_t o = base;
  for (size_t i = 0; i < jobs.size(); i++) {
    jobs[i].src = cur() + src_off[i];
    jobs[i].dst = out.data() + o; o += jobs[i].ulen;
  }
  InflateCtx ctx; ctx.jobs = &jobs;
  parallel_for(jobs.size(), threads_, inflate_one, &ctx);
  if (ctx.bad) { err_ = "corrupt BGZF block (inflate or CRC failed)"; return -1; }
  cpos_ += rel;
  return (int64_t)total;
}

BgzfWriter::~BgzfWriter() { if (f_ && f_ != stdout) fclose(f_); }

bool BgzfWriter::open(const char *path, int threads, int level) {
  f_ = strcmp(path, "-") == 0 ? stdout : fopen(path, "wb");
  threads_ = threads < 1 ? 1 : threads; level_ = level;
  if (!f_) { err_ = std::string("cannot create ") + path; return false; }
  return true;
}

struct DeflateCtx {
  const uint8_t *src; size_t n_blocks, last_len; int level;
  uint8_t *dst; std::vector<uint32_t> *clen; std::atomic<int> bad{0};
};
static const size_t SLOT = 0x10000 + 64;  // worst-case compressed block incl. framing

static void deflate_one(size_t i, void *arg) {
  DeflateCtx *c = (DeflateCtx *)arg;
  size_t len = (i + 1 == c->n_blocks) ? c->last_len : BLOCK_DATA;
  const uint8_t *src = c->src + i * BLOCK_DATA;
  uint8_t *o = c->dst + i * SLOT;
  static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  memcpy(o, head, 16);
  z_stream zs; memset(&zs, 0, sizeof(zs));
  uint32_t clen = 0;
  int level = c->level;
  bool done = false;
  if (ld().ok && level > 0) {
    size_t got = ld().compress(tls_codec.comp(level), src, len, o + 18, 0x10000 - 18 - 8);
    if (got) { clen = (uint32_t)got; done = true; } else level = 0;  // did not fit: stored block below
  }
  while (!done) {
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { c->bad = 1; return; }
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)len; zs.next_out = o + 18; zs.avail_out = 0x10000 - 18 - 8;
    int rc = deflate(&zs, Z_FINISH);
    clen = (uint32_t)zs.total_out;
    deflateEnd(&zs);
    if (rc == Z_STREAM_END) break;
    if (level == 0) { c->bad = 1; return; }
    level = 0;  // incompressible payload: stored block always fits (0xff00 + 5 bytes)
  }
  uint32_t bsize = 18 + clen + 8 - 1;
  o[16] = (uint8_t)bsize; o[17] = (uint8_t)(bsize >> 8);
  uint32_t crc = ld().ok ? ld().crc(0, src, len) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len);
  uint8_t *t = o + 18 + clen;
  for (int k = 0; k < 4; k++) t[k] = (uint8_t)(crc >> (8 * k));
  for (int k = 0; k < 4; k++) t[4 + k] = (uint8_t)((uint32_t)len >> (8 * k));
  (*c->clen)[i] = bsize + 1;
}

bool BgzfWriter::flush_blocks(const uint8_t *p, size_t n_blocks, size_t last_len) {
  // bounded batches so that the staging buffer stays small
  const size_t BATCH = 4096;
  for (size_t b0 = 0; b0 < n_blocks; b0 += BATCH) {
    size_t nb = std::min(BATCH, n_blocks - b0);
    cout_.resize(nb * SLOT);
    std::vector<uint32_t> clen(nb);
    DeflateCtx c; c.src = p + b0 * BLOCK_DATA; c.n_blocks = nb; c.last_len = (b0 + nb == n_blocks) ? last_len : BLOCK_DATA;
    c.level = level_; c.dst = cout_.data(); c.clen = &clen;
    parallel_for(nb, threads_, deflate_one, &c);
    if (c.bad) { err_ = "deflate failed"; return false; }
    for (size_t i = 0; i < nb; i++) {
      if (fwrite(cout_.data() + i * SLOT, 1, clen[i], f_) != clen[i]) { err_ = "short write"; return false; }
      bytes_out_ += clen[i];
    }
  }
  return true;
}

bool BgzfWriter::write(const uint8_t *p, size_t n) {
  if (!pending_.empty()) {  // top the kept partial block up first
    size_t take = std::min(n, BLOCK_DATA - pending_.size());
    pending_.insert(pending_.end(), p, p + take);
    p += take; n -= take;
    if (pending_.size() < BLOCK_DATA) return true;
    if (!flush_blocks(pending_.data(), 1, BLOCK_DATA)) return false;
    pending_.clear();
  }
  size_t full = n / BLOCK_DATA;
  if (full && !flush_blocks(p, full, BLOCK_DATA)) return false;
  pending_.insert(pending_.end(), p + full * BLOCK_DATA, p + n);
  return true;
}

bool BgzfWriter::write_raw(const uint8_t *p, size_t n) {
  if (!pending_.empty()) { if (!flush_blocks(pending_.data(), 1, pending_.size())) return false; pending_.clear(); }
  if (n && fwrite(p, 1, n, f_) != n) { err_ = "short write"; return false; }
  bytes_out_ += n;
  return true;
}

bool BgzfWriter::close() {
  if (!f_) return true;
  bool ok = true;
  if (!pending_.empty()) { ok = flush_blocks(pending_.data(), 1, pending_.size()); pending_.clear(); }
  if (ok && fwrite(EOF_BLOCK, 1, 28, f_) != 28) { err_ = "short write"; ok = false; }
  bytes_out_ += 28;
  if ((f_ == stdout ? fflush(f_) : fclose(f_)) != 0) { err_ = "close failed"; ok = false; }
  f_ = nullptr;
  return ok;
}

void BgzfWriter::abandon() {
  if (!f_) return;
  if (f_ == stdout) fflush(f_); else fclose(f_);
  f_ = nullptr; pending_.clear();
}

}  // namespace brio
