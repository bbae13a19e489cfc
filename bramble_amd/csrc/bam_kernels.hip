// BAM record re-encoding on the device: the byte work of
// write_to_bam (src/core.cpp:96-212) -- update_cigar (src/bam.cpp:474-528), NH / HI /
// AS tags and XS / ts deletion (:590-634), reverse_complement_bam (:636-702),
// set_mate_info (:531-588) -- for every emitted row at once.
//
// Input: the batch's original BAM alignment records (BAM file layout, starting at
// refID, i.e. without the 4-byte block_size) as one blob in HBM + the row table of
// the projection.  Output: one [block_size][record] per row, ready for BGZF.
// HBM-bound byte streaming: ~record bytes read + ~record bytes written per row.
//
//   k_bam_scan    one lane per alignment: walks the aux area once (htslib
//                 bam_aux_get semantics: first occurrence wins) and notes where NH, HI,
//                 XS|ts and (long reads) AS live, plus the AS value (bam_aux2i).
//   k_bam_size    one lane per row: output length -> scanned into offsets.
//   k_bam_encode  one wave per row: lanes stream the bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace br {

__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// htslib skip_aux: size of the value of a tag of `type` at p (p = first value byte), or -1
__device__ int64_t aux_value_len(uint8_t type, const uint8_t *p, const uint8_t *end) {
  switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': { const uint8_t *q = p; while (q < end && *q) q++; return q < end ? (q - p) + 1 : -1; }
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

__global__ void __launch_bounds__(256) k_bam_scan(BamArgs B) {
  int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= B.n_aln) return;
  const uint8_t *rec = B.blob + B.rec_off[a];
  uint64_t rlen = B.rec_off[a + 1] - B.rec_off[a];
  BamAux x;
  for (int k = 0; k < 4; k++) { x.off[k] = 0xffffffffu; x.len[k] = 0; }
  x.as_val = 0; x.aux_start = 0; x.aux_len = 0;
  if (rlen >= 32) {
    uint32_t l_qname = rec[8], n_cig = ld_u16(rec + 12);
    int32_t l_seq = (int32_t)ld_u32(rec + 16);
    uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
    uint64_t start = 32 + (uint64_t)l_qname + 4ull * n_cig + (ls + 1) / 2 + ls;
    if (start <= rlen) {
      x.aux_start = (uint32_t)start; x.aux_len = (uint32_t)(rlen - start);
      const uint8_t *s = rec + start, *end = rec + rlen;
      // slots: 0 NH, 1 XS (short) / ts (long), 2 HI, 3 AS (long reads only)
      bool have[4] = {false, false, false, false};
      while (end - s >= 3) {
        uint8_t t0 = s[0], t1 = s[1], ty = s[2];
        int64_t vl = aux_value_len(ty, s + 3, end);
        if (vl < 0 || s + 3 + vl > end) break;  // malformed: htslib stops here too
        int slot = -1;
        if (t0 == 'N' && t1 == 'H') slot = 0;
        else if (!B.long_reads && t0 == 'X' && t1 == 'S') slot = 1;
        else if (B.long_reads && t0 == 't' && t1 == 's') slot = 1;
        else if (t0 == 'H' && t1 == 'I') slot = 2;
        else if (B.long_reads && t0 == 'A' && t1 == 'S') slot = 3;
        if (slot >= 0 && !have[slot]) {
          have[slot] = true;
          x.off[slot] = (uint32_t)(s - (rec + start)); x.len[slot] = (uint32_t)(3 + vl);
          if (slot == 3) {  // bam_aux2i
            const uint8_t *v = s + 3;
            switch (ty) {
              case 'c': x.as_val = (int8_t)v[0]; break;
              case 'C': x.as_val = v[0]; break;
              case 's': x.as_val = (int16_t)ld_u16(v); break;
              case 'S': x.as_val = (int32_t)ld_u16(v); break;
              case 'i': x.as_val = (int32_t)ld_u32(v); break;
              case 'I': x.as_val = (int32_t)ld_u32(v); break;
              default: x.as_val = 0; break;
            }
          }
        }
        s += 3 + vl;
      }
      // sort the (at most four) removal intervals by offset: a tiny insertion sort
      for (int i = 1; i < 4; i++)
        for (int j = i; j > 0 && x.off[j] < x.off[j - 1]; j--) {
          uint32_t t = x.off[j]; x.off[j] = x.off[j - 1]; x.off[j - 1] = t;
          t = x.len[j]; x.len[j] = x.len[j - 1]; x.len[j - 1] = t;
        }
    }
  }
  B.aux[a] = x;
}

__device__ __forceinline__ uint32_t row_out_len(const BamArgs &B, int64_t r, const uint8_t *rec, const BamAux &x) {
  uint32_t l_qname = rec[8];
  int32_t l_seq = (int32_t)ld_u32(rec + 16);
  uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0;
  uint32_t removed = x.len[0] + x.len[1] + x.len[2] + x.len[3];
  uint32_t added = 7u + 7u + (B.long_reads ? 7u : 0u);  // NH:i, HI:i, AS:i
  return 4u + 32u + l_qname + 4u * B.r_ncig[r] + (ls + 1) / 2 + ls + (x.aux_len - removed) + added;
}

__global__ void __launch_bounds__(256) k_bam_size(BamArgs B) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B.n_rows) return;
  int32_t a = B.r_input[r];
  const uint8_t *rec = B.blob + B.rec_off[a];
  B.out_len[r] = row_out_len(B, r, rec, B.aux[a]);
}

// 4-bit base complement of reverse_complement_bam (src/bam.cpp:658-667)
__device__ __forceinline__ uint8_t comp4(uint8_t nt) { return nt == 1 ? 8 : nt == 2 ? 4 : nt == 4 ? 2 : nt == 8 ? 1 : 15; }

__global__ void __launch_bounds__(256) k_bam_encode(BamArgs B) {
  const int lane = threadIdx.x & 63;
  int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= B.n_rows) return;
  int32_t a = B.r_input[r];
  const uint8_t *rec = B.blob + B.rec_off[a];
  BamAux x = B.aux[a];
  uint8_t *out = B.out + B.out_off[r];
  uint32_t total = (uint32_t)(B.out_off[r + 1] - B.out_off[r]);

  uint32_t l_qname = rec[8];
  uint32_t flag = ld_u16(rec + 14);
  int32_t l_seq = (int32_t)ld_u32(rec + 16);
  uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0;
  uint32_t n_cig_in = ld_u16(rec + 12);
  uint32_t n_cig = B.r_ncig[r];
  bool minus = B.r_strand[r] == '-';
  bool paired = B.r_paired[r], same = B.r_same[r];
  // flags: secondary (src/core.cpp:142-143), reverse (bam.cpp:698), mate bits (bam.cpp:531-588)
  if (B.r_primary[r]) flag &= ~0x100u; else flag |= 0x100u;
  if (minus) flag ^= 0x10u;
  int32_t mtid = -1, mpos = -1, tlen = 0;
  if (!paired) flag &= ~(0x1u | 0x2u | 0x20u);
  else {
    flag |= 0x1u;
    if (minus) flag |= 0x20u;  // both branches of bam.cpp:551-555 test the record's own transcript strand
    mtid = B.r_mate_tid[r]; mpos = B.r_mate_pos[r];
    if (same) { flag |= 0x2u; tlen = B.r_isize[r]; } else flag &= ~0x2u;
  }
  // header: block_size + the 32 fixed bytes, one word per lane 0..8
  if (lane < 9) {
    uint32_t w;
    switch (lane) {
      case 0: w = total - 4u; break;
      case 1: w = B.r_tid[r]; break;
      case 2: w = B.r_pos[r]; break;
      case 3: w = l_qname | ((B.r_mapq[r] & 0xffu) << 8) | (ld_u16(rec + 10) << 16); break;   // l_read_name, mapq, bin (kept)
      case 4: w = (n_cig & 0xffffu) | (flag << 16); break;
      case 5: w = (uint32_t)l_seq; break;
      case 6: w = (uint32_t)mtid; break;
      case 7: w = (uint32_t)mpos; break;
      default: w = (uint32_t)tlen; break;
    }
    uint8_t *o = out + 4 * lane;
    o[0] = (uint8_t)w; o[1] = (uint8_t)(w >> 8); o[2] = (uint8_t)(w >> 16); o[3] = (uint8_t)(w >> 24);
  }
  uint32_t o = 36;
  // read name
  for (uint32_t i = lane; i < l_qname; i += 64) out[o + i] = rec[32 + i];
  o += l_qname;
  // rewritten CIGAR (op order reversed on '-', bam.cpp:688-695)
  const uint32_t *cg = B.cigar + B.r_cigoff[r];
  for (uint32_t i = lane; i < 4 * n_cig; i += 64) {
    uint32_t k = i >> 2, w = cg[minus ? n_cig - 1 - k : k];
    out[o + i] = (uint8_t)(w >> (8 * (i & 3)));
  }
  o += 4 * n_cig;
  // sequence: reverse-complemented nibbles on '-' (bam.cpp:671-678; the pad nibble of an odd length stays 0)
  const uint8_t *seq = rec + 32 + l_qname + 4 * n_cig_in;
  uint32_t sb = (ls + 1) / 2;
  for (uint32_t i = lane; i < sb; i += 64) {
    uint8_t v;
    if (!minus) v = seq[i];
    else {
      uint32_t p0 = 2 * i, p1 = 2 * i + 1;
      uint32_t s0 = ls - 1 - p0;
      uint8_t n0 = (seq[s0 >> 1] >> ((~s0 & 1) << 2)) & 0xf, n1 = 0;
      v = (uint8_t)(comp4(n0) << 4);
      if (p1 < ls) { uint32_t s1 = ls - 1 - p1; n1 = (seq[s1 >> 1] >> ((~s1 & 1) << 2)) & 0xf; v |= comp4(n1); }
    }
    out[o + i] = v;
  }
  o += sb;
  // qualities: reversed on '-' unless absent (0xff) (bam.cpp:680-686)
  const uint8_t *qual = seq + sb;
  bool rev_q = minus && ls > 0 && qual[0] != 0xff;
  for (uint32_t i = lane; i < ls; i += 64) out[o + i] = qual[rev_q ? ls - 1 - i : i];
  o += ls;
  // aux: original minus the first NH, XS|ts, HI (and AS for long reads) ...
  const uint8_t *aux = rec + x.aux_start;
  uint32_t removed = x.len[0] + x.len[1] + x.len[2] + x.len[3];
  uint32_t keep = x.aux_len - removed;
  for (uint32_t i = lane; i < keep; i += 64) {
    uint32_t src = i;
#pragma unroll
    for (int k = 0; k < 4; k++) if (x.off[k] != 0xffffffffu && x.off[k] <= src) src += x.len[k];
    out[o + i] = aux[src];
  }
  o += keep;
  // ... plus NH:i, (AS:i,) HI:i appended in that order (bam.cpp:590-634, core.cpp:118-161)
  if (lane < 21) {
    int which = lane / 7, k = lane % 7;
    bool lr = B.long_reads != 0;
    if (which < (lr ? 3 : 2)) {
      int kind = which == 0 ? 0 : (lr ? (which == 1 ? 1 : 2) : 2);  // 0 NH, 1 AS, 2 HI
      uint32_t val;
      if (kind == 0) val = B.r_nh[r];
      else if (kind == 2) val = B.r_hi[r];
      else val = (uint32_t)(int32_t)(((double)x.as_val + (double)B.r_clip[r]) * B.r_sim[r]);  // set_as_tag
      uint8_t byte;
      const char *tags = "NHASHI";
      if (k < 2) byte = (uint8_t)tags[2 * kind + k];
      else if (k == 2) byte = 'i';
      else byte = (uint8_t)(val >> (8 * (k - 3)));
      out[o + 7 * which + k] = byte;
    }
  }
}

void launch_bam_scan(hipStream_t st, const BamArgs &B) {
  if (B.n_aln > 0) hipLaunchKernelGGL(k_bam_scan, dim3((unsigned)((B.n_aln + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_size(hipStream_t st, const BamArgs &B) {
  if (B.n_rows > 0) hipLaunchKernelGGL(k_bam_size, dim3((unsigned)((B.n_rows + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_encode(hipStream_t st, const BamArgs &B) {
  if (B.n_rows > 0) hipLaunchKernelGGL(k_bam_encode, dim3((unsigned)((B.n_rows + 3) / 4)), dim3(256), 0, st, B);
}
size_t bam_aux_bytes() { return sizeof(BamAux); }

}  // namespace br
