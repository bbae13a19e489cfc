// BGZF (blocked gzip) streams for the BAM container: threaded block inflate / deflate on the host.
// Replaces what the reference gets from htslib's bgzf layer behind GSamReader / GSamWriter
// (gclib/GSam.h; include/bramble.h:29-85 BamIO).  The compressed bytes are not part of the parity
// contract (any valid BGZF framing of the same BAM stream is the same file content).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <string>
#include <utility>
#include <vector>

namespace brio {

// Large host buffers ask for transparent huge pages (the boxes run THP in "madvise" mode): a 20.9 M-record file is 3.3 GB
// of record bytes first touched by the reader threads, and 4 KB faults were 1.3 s of system time per run.
inline void advise_huge(void *p, size_t n) {
  static const bool on = [] { const char *e = getenv("BRAMBLE_AMD_HUGE_PAGES"); return !e || atoi(e) != 0; }();
  if (!on) return;
  const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + n) & ~(uintptr_t)4095;
  if (e > a && e - a >= ((size_t)4 << 20)) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
}

// growable byte buffer without value-initialisation (std::vector<uint8_t>::resize zero-fills gigabytes here)
class ByteBuf {
 public:
  ByteBuf() {}
  ~ByteBuf() { free(p_); }
  ByteBuf(const ByteBuf &) = delete;
  ByteBuf &operator=(const ByteBuf &) = delete;
  uint8_t *data() { return p_; }
  const uint8_t *data() const { return p_; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  uint8_t &operator[](size_t i) { return p_[i]; }
  const uint8_t &operator[](size_t i) const { return p_[i]; }
  void resize(size_t n) {
    if (n > cap_) { size_t c = cap_ + cap_ / 2; if (c < n) c = n; if (c < 4096) c = 4096; p_ = (uint8_t *)realloc(p_, c); if (!p_) abort(); cap_ = c; advise_huge(p_, c); }
    n_ = n;
  }
  void reserve(size_t c) {
    if (c <= cap_) return;
    if (n_ == 0) { free(p_); p_ = (uint8_t *)malloc(c); }   // nothing to carry over: a fresh mapping, not a moved one
    else p_ = (uint8_t *)realloc(p_, c);
    if (!p_) abort();
    cap_ = c; advise_huge(p_, c);
  }
  void erase_front(size_t k) { if (k >= n_) { n_ = 0; return; } memmove(p_, p_ + k, n_ - k); n_ -= k; }
  void swap(ByteBuf &o) { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }
  void clear() { n_ = 0; }

 private:
  uint8_t *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};

class BgzfReader {
 public:
  ~BgzfReader();
  // returns false (and sets error()) when the file cannot be opened or is not BGZF
  bool open(const char *path, int threads);
  // appends at least `want` uncompressed bytes to `out` unless the stream ends first;
  // returns the number of bytes appended, 0 at end of stream, -1 on a corrupt block
  int64_t read(ByteBuf &out, size_t want);
  // the mapped file (regular files only; null for a pipe): the device reader takes the compressed bytes from here
  const uint8_t *mapped() const { return map_; }
  size_t mapped_size() const { return map_ ? map_size_ : 0; }
  bool eof() const { return eof_; }
  const std::string &error() const { return err_; }

 private:
  bool fill(size_t need);
  const uint8_t *cur() const { return map_ ? map_ + cpos_ : cbuf_.data() + cpos_; }
  size_t have() const { return (map_ ? map_size_ : cbuf_.size()) - cpos_; }
  const uint8_t *map_ = nullptr; size_t map_size_ = 0;
  FILE *f_ = nullptr;
  int threads_ = 1;
  bool eof_ = false;
  ByteBuf cbuf_;  // compressed bytes not yet consumed
  size_t cpos_ = 0;
  std::string err_;
};

class BgzfWriter {
 public:
  ~BgzfWriter();
  bool open(const char *path, int threads, int level);
  // compresses [p, p+n) into 0xff00-byte blocks (a trailing partial block is kept for the next call)
  bool write(const uint8_t *p, size_t n);
  // closes the pending partial block, then appends bytes that already ARE complete BGZF blocks (device deflate)
  bool write_raw(const uint8_t *p, size_t n);
  bool close();  // flushes and appends the 28-byte EOF block
  void abandon();  // closes the file without the EOF block (a failed run must not leave a stream that looks complete)
  const std::string &error() const { return err_; }
  uint64_t bytes_out() const { return bytes_out_; }

 private:
  bool flush_blocks(const uint8_t *p, size_t n_blocks, size_t last_len);
  FILE *f_ = nullptr;
  int threads_ = 1, level_ = 6;
  std::vector<uint8_t> pending_;
  std::vector<uint8_t> cout_;
  std::string err_;
  uint64_t bytes_out_ = 0;
};

const char *codec_name();  // "libdeflate" or "zlib"

// runs fn(i) for i in [0, n) on up to `threads` threads
void parallel_for(size_t n, int threads, void (*fn)(size_t, void *), void *arg);

}  // namespace brio
