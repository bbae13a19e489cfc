// Aux-area helpers shared by the reader-side kernels (parse_kernels.hip) and the re-encoder (bam_kernels.hip), and the
// restore of a CIGAR that was spilled into a CG:B,I tag.
//
// A BAM record holds at most 65535 CIGAR ops in its n_cigar_op field; a longer CIGAR (ultra-long ONT reads) is stored
// as the placeholder <l_seq>S<ref_len>N with the real ops in a CG:B,I tag (SAM spec 4.2.2).  The reference reads every
// record through htslib (include/bramble.h:29-85 over gclib/GSam.cpp), whose bam_read1 puts the real CIGAR back before
// gclib/GSam.cpp:197-201 walks it, and whose bam_write1 spills a long CIGAR again on the way out.  Here the record
// stays as it is in the blob: the kernels that need the CIGAR take it from the tag, the re-encoder drops the tag.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace br {

typedef uint32_t cg_u32u __attribute__((aligned(1)));
typedef uint16_t cg_u16u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { return *(const cg_u16u *)p; }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) { return *(const cg_u32u *)p; }

// htslib skip_aux: size of the value of a tag of `type` at p (p = first value byte), or -1
__device__ inline int64_t aux_value_len(uint8_t type, const uint8_t *p, const uint8_t *end) {
  switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': {   // up to and including the NUL: four bytes per load while four are left
      const uint8_t *q = p;
      while (end - q >= 4) {
        const uint32_t w = ld_u32(q), z = (w - 0x01010101u) & ~w & 0x80808080u;
        if (z) return (q - p) + (__builtin_ctz(z) >> 3) + 1;
        q += 4;
      }
      while (q < end && *q) q++;
      return q < end ? (q - p) + 1 : -1;
    }
    case 'B': {
      if (end - p < 5) return -1;
      uint8_t st = p[0]; uint32_t n = ld_u32(p + 1);
      int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
      if (!sz) return -1;
      return 5 + (int64_t)n * sz;
    }
    default: return -1;
  }
}

struct CgTag { uint32_t tag_at, tag_len, n; };   // the tag's first byte (record-relative), its bytes ("CG" 'B' subtype count words), ops

// htslib's bam_tag2cigar rule, cheap test first (nearly every record stops there): a mapped record whose first op is a
// soft clip of the whole read ...
__device__ __forceinline__ bool cg_candidate(const uint8_t *rec, uint64_t rlen, uint32_t l_qname, uint32_t n_cig, int32_t l_seq) {
  if (n_cig == 0 || rlen < 32 || 32ull + l_qname + 4ull * n_cig > rlen) return false;
  if ((int32_t)ld_u32(rec) < 0 || (int32_t)ld_u32(rec + 4) < 0) return false;   // tid, pos
  const uint32_t w0 = ld_u32(rec + 32 + l_qname);
  return (w0 & 0xfu) == 4u && (w0 >> 4) == (uint32_t)l_seq;
}
// ... and whose FIRST CG tag (bam_aux_get) is of type B,I or B,i with n_cigar <= count < 2^29
__device__ inline bool cg_find(const uint8_t *rec, uint64_t rlen, uint32_t l_qname, uint32_t n_cig, int32_t l_seq, CgTag &t) {
  if (!cg_candidate(rec, rlen, l_qname, n_cig, l_seq)) return false;
  const uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
  const uint64_t start = 32ull + l_qname + 4ull * n_cig + (ls + 1) / 2 + ls;
  if (start > rlen) return false;
  const uint8_t *s = rec + start, *end = rec + rlen;
  while (end - s >= 3) {
    const int64_t vl = aux_value_len(s[2], s + 3, end);
    if (vl < 0 || s + 3 + vl > end) return false;   // malformed: the walk stops, the tag was not found
    if (s[0] == 'C' && s[1] == 'G') {
      if (s[2] != 'B' || !(s[3] == 'I' || s[3] == 'i')) return false;
      const uint32_t n = ld_u32(s + 4);
      if (n < n_cig || n >= (1u << 29)) return false;
      t.tag_at = (uint32_t)(s - rec); t.tag_len = (uint32_t)(3 + vl); t.n = n;
      return true;
    }
    s += 3 + vl;
  }
  return false;
}

}  // namespace br
