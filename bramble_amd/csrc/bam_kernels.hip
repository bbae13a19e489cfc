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
//                 XS|ts and (long reads) AS live, plus the AS value (bam_aux2i) and the
//                 tag_char1 value of XS / ts for the reader side (parse_kernels.hip).
//   k_bam_size    one lane per row: output length -> scanned into offsets.
//   k_bam_encode  G lanes (default 8) per row: 16-byte unaligned copies of name / SEQ / QUAL / kept aux pieces,
//                 bit-reverse reverse complement, the fixed fields and the appended tags.
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
  uint64_t rlen = B.rec_len ? (uint64_t)B.rec_len[a] : B.rec_off[a + 1] - B.rec_off[a];
  BamAux x;
  int8_t xs_c = 0, ts_c = 0;
  bool have_xs = false, have_ts = false;
  for (int k = 0; k < 4; k++) { x.off[k] = 0xffffffffu; x.len[k] = 0; }
  x.as_val = 0; x.aux_start = 0; x.aux_len = 0; x.c_a = x.c_b = x.c_c = 0; x.qual_present = 0;
  if (rlen >= 32) {
    uint32_t l_qname = rec[8], n_cig = ld_u16(rec + 12);
    int32_t l_seq = (int32_t)ld_u32(rec + 16);
    uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
    uint64_t start = 32 + (uint64_t)l_qname + 4ull * n_cig + (ls + 1) / 2 + ls;
    x.c_a = ld_u32(rec + 8); x.c_b = ld_u32(rec + 12); x.c_c = (uint32_t)l_seq;
    if (start <= rlen) {
      if (ls > 0) x.qual_present = rec[32 + (uint64_t)l_qname + 4ull * n_cig + (ls + 1) / 2] != 0xff;
      x.aux_start = (uint32_t)start; x.aux_len = (uint32_t)(rlen - start);
      const uint8_t *s = rec + start, *end = rec + rlen;
      // slots: 0 NH, 1 XS (short) / ts (long), 2 HI, 3 AS (long reads only)
      bool have[4] = {false, false, false, false};
      bool have_cg = false;
      while (end - s >= 3) {
        uint8_t t0 = s[0], t1 = s[1], ty = s[2];
        int64_t vl = aux_value_len(ty, s + 3, end);
        if (vl < 0 || s + 3 + vl > end) break;  // malformed: htslib stops here too
        if (t0 == 'C' && t1 == 'G' && ty == 'B' && s[3] == 'I') have_cg = true;
        // tag_char1 (gclib/GSam.cpp:310-318): first value byte of the first XS / ts tag when A or Z
        if (t0 == 'X' && t1 == 'S' && !have_xs) { have_xs = true; if (ty == 'A' || ty == 'Z') xs_c = (int8_t)s[3]; }
        if (t0 == 't' && t1 == 's' && !have_ts) { have_ts = true; if (ty == 'A' || ty == 'Z') ts_c = (int8_t)s[3]; }
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
      // A CIGAR of more than 65535 ops lives in a CG:B,I tag behind the placeholder <l_seq>S<ref_len>N (SAM spec 4.2.2);
      // htslib's bam_read1 restores it for the reference, this reader does not: the bundle is refused.
      if (have_cg && n_cig == 2 && B.cg_flag) {
        const uint8_t *cg = rec + 32 + l_qname;
        const uint32_t w0 = ld_u32(cg), w1 = ld_u32(cg + 4);
        if ((w0 & 0xfu) == 4u && (w0 >> 4) == (uint32_t)l_seq && (w1 & 0xfu) == 3u) *B.cg_flag = 1u;
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
  if (B.xs_out) { B.xs_out[a] = xs_c; B.ts_out[a] = ts_c; }
}

__device__ __forceinline__ uint32_t row_out_len(const BamArgs &B, int64_t r, const uint8_t *rec, const BamAux &x) {
  uint32_t l_qname = x.c_a & 0xffu;
  int32_t l_seq = (int32_t)x.c_c;
  uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0;
  uint32_t removed = x.len[0] + x.len[1] + x.len[2] + x.len[3];
  uint32_t added = 7u + 7u + (B.long_reads ? 7u : 0u);  // NH:i, HI:i, AS:i
  return 4u + 32u + l_qname + 4u * (((const uint32_t *)(B.r_a + r))[2] & RM_NCIG) + (ls + 1) / 2 + ls + (x.aux_len - removed) + added;
}

__global__ void __launch_bounds__(256) k_bam_size(BamArgs B) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B.n_rows) return;
  int32_t a = (int32_t)((const uint32_t *)(B.r_rec + r))[1];
  const uint8_t *rec = B.blob + B.rec_off[a];
  B.out_len[r] = row_out_len(B, r, rec, B.aux[a]);
  if ((((const uint32_t *)(B.r_a + r))[2] & RM_NCIG) > 65535u) *B.too_long = 1;  // would need htslib's CG:B,I spill-over; refused by the host
}

// 4-bit base complement of reverse_complement_bam (src/bam.cpp:658-667)
__device__ __forceinline__ uint8_t comp4(uint8_t nt) { return nt == 1 ? 8 : nt == 2 ? 4 : nt == 4 ? 2 : nt == 8 ? 1 : 15; }

// Unaligned wide accesses: gfx950 global loads / stores need no alignment, so byte
// regions are moved 16 bytes per lane-instruction.
struct __attribute__((packed, aligned(1))) W4 { uint32_t a, b, c, d; };
typedef uint32_t u32u __attribute__((aligned(1)));

template <int G>
__device__ __forceinline__ void copy_fwd(uint8_t *dst, const uint8_t *src, uint32_t n, int lane) {
  if (n >= 16u) {  // ceil(n / 16) chunks, the last one ending at n: it overlaps its predecessor (same bytes) and saves the byte tail
    for (uint32_t i = 16u * lane; i < n; i += 16u * G) { uint32_t o = i + 16u <= n ? i : n - 16u; *(W4 *)(dst + o) = *(const W4 *)(src + o); }
  } else {
    for (uint32_t i = lane; i < n; i += G) dst[i] = src[i];
  }
}
// dst[i] = src[n-1-i]
template <int G>
__device__ __forceinline__ void copy_rev(uint8_t *dst, const uint8_t *src, uint32_t n, int lane) {
  if (n >= 16u) {
    for (uint32_t i0 = 16u * lane; i0 < n; i0 += 16u * G) {
      uint32_t i = i0 + 16u <= n ? i0 : n - 16u;
      W4 w = *(const W4 *)(src + (n - 16u - i));
      W4 o; o.a = __builtin_bswap32(w.d); o.b = __builtin_bswap32(w.c); o.c = __builtin_bswap32(w.b); o.d = __builtin_bswap32(w.a);
      *(W4 *)(dst + i) = o;
    }
  } else {
    for (uint32_t i = lane; i < n; i += G) dst[i] = src[n - 1 - i];
  }
}
// reverse-complement of 8 packed 4-bit bases (one dword of BAM SEQ): reversing all 32 bits
// reverses the base order AND maps A(1)<->T(8), C(2)<->G(4); every other code becomes N(15)
// (comp_table of src/bam.cpp:658-667)
__device__ __forceinline__ uint32_t revcomp8(uint32_t v) {
  uint32_t x = __builtin_bitreverse32(v);
  uint32_t pop = (x & 0x11111111u) + ((x >> 1) & 0x11111111u) + ((x >> 2) & 0x11111111u) + ((x >> 3) & 0x11111111u);
  uint32_t y = pop ^ 0x11111111u;                       // zero nibble <=> exactly one bit set
  uint32_t bad = (y | (y >> 1) | (y >> 2) | (y >> 3)) & 0x11111111u;
  return x | (bad * 15u);
}

// G lanes cooperate on one row (records are ~200-300 bytes: a full wave per row would
// leave most lanes idle in every region loop)
// one output record: `rec` = the input record (global memory, or its staged copy in LDS), `out` = where the
// record goes (global memory, or the block's staging span in LDS)
template <int G>
__device__ __forceinline__ void encode_row(const BamArgs &B, int64_t r, int lane, const uint8_t *rec, const BamAux &x,
                                           uint8_t *out, uint32_t total) {

  // l_qname|mapq|bin, n_cigar|flag, l_seq as k_bam_scan read them: no dependent read of the record before its regions
  uint32_t l_qname = x.c_a & 0xffu;
  uint32_t bin = x.c_a >> 16;
  uint32_t n_cig_in = x.c_b & 0xffffu, flag = x.c_b >> 16;
  int32_t l_seq = (int32_t)x.c_c;
  uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0;
  // the packed row {tid, pos, meta, NH} + the record {match, input, NH, HI | flags}; the pair's other record is the adjacent row
  const uint4 ra = B.r_a[r];
  const uint32_t meta = ra.z, nh = ra.w, hi = ((const uint32_t *)(B.r_rec + r))[3] & RR_HI;
  uint32_t n_cig = meta & RM_NCIG;
  bool minus = meta & RM_MINUS;
  bool paired = meta & RM_PAIRED, same = meta & RM_SAME;
  // flags: secondary (src/core.cpp:142-143), reverse (bam.cpp:698), mate bits (bam.cpp:531-588)
  if (meta & RM_PRIMARY) flag &= ~0x100u; else flag |= 0x100u;
  if (minus) flag ^= 0x10u;
  int32_t mtid = -1, mpos = -1, tlen = 0;
  if (!paired) flag &= ~(0x1u | 0x2u | 0x20u);
  else {
    flag |= 0x1u;
    if (minus) flag |= 0x20u;  // both branches of bam.cpp:551-555 test the record's own transcript strand
    const uint4 rb = B.r_a[(meta & RM_FIRST) ? r + 1 : r - 1];
    const int32_t my_pos = (int32_t)ra.y;
    mpos = (int32_t)rb.y;
    if (same) {
      flag |= 0x2u; mtid = (int32_t)ra.x;
      const int32_t lq = B.l_qseq[((const uint32_t *)(B.r_rec + r))[1]];
      tlen = (my_pos <= mpos) ? (mpos + lq) - my_pos : -((my_pos + lq) - mpos);
    } else { flag &= ~0x2u; mtid = (int32_t)rb.x; }
  }
  // get_mapq (src/core.cpp:46-58)
  const uint32_t mapq = B.long_reads ? (nh > 1 ? 0u : 3u) : (nh == 1 ? 255u : nh == 2 ? 3u : (nh == 3 || nh == 4) ? 1u : 0u);
  // block_size + the 32 fixed bytes: nine dwords
  if (lane == 0) {
    W4 h0; h0.a = total - 4u; h0.b = ra.x; h0.c = ra.y;
    h0.d = l_qname | (mapq << 8) | (bin << 16);                              // l_read_name, mapq, bin (kept)
    *(W4 *)out = h0;
  } else if (lane == 1) {
    W4 h1; h1.a = (n_cig & 0xffffu) | (flag << 16); h1.b = (uint32_t)l_seq; h1.c = (uint32_t)mtid; h1.d = (uint32_t)mpos;
    *(W4 *)(out + 16) = h1;
  } else if (lane == 2) {
    *(u32u *)(out + 32) = (uint32_t)tlen;
  }
  uint32_t o = 36;
  copy_fwd<G>(out + o, rec + 32, l_qname, lane);                            // read name
  o += l_qname;
  // rewritten CIGAR (op order reversed on '-', bam.cpp:688-695)
  {
    const uint2 c = B.r_c[r];   // the ops themselves (<= 2), or their offset in the pool
    if (n_cig <= 2u) {
      if (lane < (int)n_cig) *(u32u *)(out + o + 4 * lane) = ((minus ? n_cig - 1 - lane : lane) == 0) ? c.x : c.y;
    } else {
      const uint32_t *cg = B.pool + (((uint64_t)c.y << 32) | c.x);
      for (uint32_t k = lane; k < n_cig; k += G) *(u32u *)(out + o + 4 * k) = cg[minus ? n_cig - 1 - k : k];
    }
  }
  o += 4 * n_cig;
  // sequence: reverse-complemented nibbles on '-' (bam.cpp:671-678; the pad nibble of an odd length stays 0)
  const uint8_t *seq = rec + 32 + l_qname + 4 * n_cig_in;
  uint32_t sb = (ls + 1) / 2;
  if (!minus) copy_fwd<G>(out + o, seq, sb, lane);
  else if ((ls & 1u) == 0) {
    uint32_t n4 = sb & ~3u;
    for (uint32_t i = 4u * lane; i < n4; i += 4u * G) *(u32u *)(out + o + i) = revcomp8(*(const u32u *)(seq + (sb - 4u - i)));
    for (uint32_t i = n4 + lane; i < sb; i += G) {  // tail: one byte = two bases
      uint8_t v = seq[sb - 1 - i];
      out[o + i] = (uint8_t)((comp4(v & 0xf) << 4) | comp4(v >> 4));
    }
  } else {
    for (uint32_t i = lane; i < sb; i += G) {
      uint32_t p0 = 2 * i, p1 = 2 * i + 1;
      uint32_t s0 = ls - 1 - p0;
      uint8_t n0 = (seq[s0 >> 1] >> ((~s0 & 1) << 2)) & 0xf;
      uint8_t v = (uint8_t)(comp4(n0) << 4);
      if (p1 < ls) { uint32_t s1 = ls - 1 - p1; uint8_t n1 = (seq[s1 >> 1] >> ((~s1 & 1) << 2)) & 0xf; v |= comp4(n1); }
      out[o + i] = v;
    }
  }
  o += sb;
  // qualities: reversed on '-' unless absent (0xff) (bam.cpp:680-686)
  const uint8_t *qual = seq + sb;
  bool rev_q = minus && ls > 0 && x.qual_present;
  if (rev_q) copy_rev<G>(out + o, qual, ls, lane); else copy_fwd<G>(out + o, qual, ls, lane);
  o += ls;
  // aux: original minus the first NH, XS|ts, HI (and AS for long reads): up to five kept pieces ...
  const uint8_t *aux = rec + x.aux_start;
  uint32_t src = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (x.off[k] == 0xffffffffu) break;
    uint32_t piece = x.off[k] - src;
    copy_fwd<G>(out + o, aux + src, piece, lane);
    o += piece; src = x.off[k] + x.len[k];
  }
  copy_fwd<G>(out + o, aux + src, x.aux_len - src, lane);
  o += x.aux_len - src;
  // ... plus NH:i, (AS:i,) HI:i appended in that order (bam.cpp:590-634, core.cpp:118-161)
  // one lane per tag: seven bytes [t0 t1 'i' v0 v1 v2 v3] as two overlapping unaligned dwords
  {
    const bool lr = B.long_reads != 0;
    const int n_tags = lr ? 3 : 2;
    if (lane < n_tags) {
      const int kind = lane == 0 ? 0 : (lr ? (lane == 1 ? 1 : 2) : 2);  // 0 NH, 1 AS, 2 HI
      uint32_t val, t01;
      if (kind == 0) { val = nh; t01 = (uint32_t)'N' | ((uint32_t)'H' << 8); }
      else if (kind == 2) { val = hi; t01 = (uint32_t)'H' | ((uint32_t)'I' << 8); }
      else { val = (uint32_t)(int32_t)(((double)x.as_val + (double)(B.r_clip ? B.r_clip[r] : 0)) * (B.r_sim ? B.r_sim[r] : 0.0)); t01 = (uint32_t)'A' | ((uint32_t)'S' << 8); }  // set_as_tag
      uint8_t *t = out + o + 7 * lane;
      *(u32u *)t = t01 | ((uint32_t)'i' << 16) | (val << 24);
      *(u32u *)(t + 3) = val;
    }
  }
}

template <int G>
__global__ void __launch_bounds__(256) k_bam_encode(BamArgs B) {
  const int lane = threadIdx.x & (G - 1);
  int64_t r = (int64_t)blockIdx.x * (256 / G) + (threadIdx.x / G);
  if (r >= B.n_rows) return;
  int32_t a = (int32_t)((const uint32_t *)(B.r_rec + r))[1];
  encode_row<G>(B, r, lane, B.blob + B.rec_off[a], B.aux[a], B.out + B.out_off[r], (uint32_t)(B.out_off[r + 1] - B.out_off[r]));
}

void launch_bam_scan(hipStream_t st, const BamArgs &B) {
  if (B.n_aln > 0) hipLaunchKernelGGL(k_bam_scan, dim3((unsigned)((B.n_aln + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_size(hipStream_t st, const BamArgs &B) {
  if (B.n_rows > 0) hipLaunchKernelGGL(k_bam_size, dim3((unsigned)((B.n_rows + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_encode(hipStream_t st, const BamArgs &B, int lanes) {
  if (B.n_rows <= 0) return;
  if (lanes == 4) hipLaunchKernelGGL((k_bam_encode<4>), dim3((unsigned)((B.n_rows + 63) / 64)), dim3(256), 0, st, B);
  else if (lanes == 8) hipLaunchKernelGGL((k_bam_encode<8>), dim3((unsigned)((B.n_rows + 31) / 32)), dim3(256), 0, st, B);
  else if (lanes == 32) hipLaunchKernelGGL((k_bam_encode<32>), dim3((unsigned)((B.n_rows + 7) / 8)), dim3(256), 0, st, B);
  else if (lanes == 64) hipLaunchKernelGGL((k_bam_encode<64>), dim3((unsigned)((B.n_rows + 3) / 4)), dim3(256), 0, st, B);
  else hipLaunchKernelGGL((k_bam_encode<16>), dim3((unsigned)((B.n_rows + 15) / 16)), dim3(256), 0, st, B);
}
size_t bam_aux_bytes() { return sizeof(BamAux); }

}  // namespace br
