// BGZF (blocked gzip) streams for the BAM container: threaded block inflate / deflate on the host.
// Replaces what the reference gets from htslib's bgzf layer behind GSamReader / GSamWriter
// (gclib/GSam.h; include/bramble.h:29-85 BamIO).  The compressed bytes are not part of the parity
// contract (any valid BGZF framing of the same BAM stream is the same file content).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

namespace brio {

class BgzfReader {
 public:
  ~BgzfReader();
  // returns false (and sets error()) when the file cannot be opened or is not BGZF
  bool open(const char *path, int threads);
  // appends at least `want` uncompressed bytes to `out` unless the stream ends first;
  // returns the number of bytes appended, 0 at end of stream, -1 on a corrupt block
  int64_t read(std::vector<uint8_t> &out, size_t want);
  bool eof() const { return eof_; }
  const std::string &error() const { return err_; }

 private:
  bool fill(size_t need);
  FILE *f_ = nullptr;
  int threads_ = 1;
  bool eof_ = false;
  std::vector<uint8_t> cbuf_;  // compressed bytes not yet consumed
  size_t cpos_ = 0;
  std::string err_;
};

class BgzfWriter {
 public:
  ~BgzfWriter();
  bool open(const char *path, int threads, int level);
  // compresses [p, p+n) into 0xff00-byte blocks (a trailing partial block is kept for the next call)
  bool write(const uint8_t *p, size_t n);
  bool close();  // flushes and appends the 28-byte EOF block
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
