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
//   k_bam_tasks   (default) a wave per 32 rows: the rows' byte regions as 16-byte copy tasks over all 64 lanes.
//   k_bam_encode  G lanes per row (bam_lanes = 4..64): 16-byte unaligned copies of name / SEQ / QUAL / kept aux pieces,
//                 bit-reverse reverse complement, the fixed fields and the appended tags.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bam_cg.h"
#include "kernels.h"
#include "records_inl.h"

namespace br {

// Unaligned wide accesses: gfx950 global loads / stores need no alignment.
struct __attribute__((packed, aligned(1))) W4 { uint32_t a, b, c, d; };
typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint16_t u16u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t fix_nib(uint32_t x) { return rec_fix_nib(x); }   // records_inl.h
// The input alignment behind output row r: word 1 of the match-table path's r_rec {match, input, NH, HI | flags}, word 0 of
// the direct path's detail row {input, junc_hits, aligned_len, HI} (B.rec_x); HI is word 3 of either.
__device__ __forceinline__ uint32_t rec_input(const BamArgs &B, int64_t r) { return ((const uint32_t *)(B.r_rec + r))[B.rec_x ? 0 : 1]; }

__global__ void __launch_bounds__(256) k_bam_scan(BamArgs B) {
  __shared__ unsigned long long sh_end[4];
  int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // end of the last record byte of the blob: a k_bam_tasks copy of fewer than 16 bytes loads 16 and must not read past it
  unsigned long long e = 0;
  if (a < B.n_aln) e = B.rec_off[a] + (B.rec_len ? (uint64_t)B.rec_len[a] : B.rec_off[a + 1] - B.rec_off[a]);
  for (int o = 32; o; o >>= 1) { unsigned long long t = __shfl_xor(e, o); e = t > e ? t : e; }
  if ((threadIdx.x & 63) == 0) sh_end[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) e = sh_end[w] > e ? sh_end[w] : e;
    atomicMax((unsigned long long *)B.blob_end + (size_t)(blockIdx.x & (BLOB_END_SLOTS - 1)) * BLOB_END_STRIDE, e);   // (one of 64 lines: see kernels.h)
  }
  if (a < B.n_aln) bam_scan_one(B, a, GlobalRec{B.blob + B.rec_off[a]}, B.rec_len ? (uint64_t)B.rec_len[a] : B.rec_off[a + 1] - B.rec_off[a]);
}

__global__ void __launch_bounds__(256) k_bam_size(BamArgs B) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B.n_rows) return;
  int32_t a = (int32_t)rec_input(B, r);
  const uint32_t n_cig = ((const uint32_t *)(B.r_a + r))[2] & RM_NCIG;
  // more than 65535 ops: bam_write1's placeholder (8 bytes) in the CIGAR field, "CGBI" + count + the ops behind the aux area
  B.out_len[r] = B.base_len[a] + 4u * n_cig + (n_cig > 65535u ? 16u : 0u);
}

// 4-bit base complement of reverse_complement_bam (src/bam.cpp:658-667)
__device__ __forceinline__ uint8_t comp4(uint8_t nt) { return nt == 1 ? 8 : nt == 2 ? 4 : nt == 4 ? 2 : nt == 8 ? 1 : 15; }

// Unaligned wide accesses: gfx950 global loads / stores need no alignment, so byte
// regions are moved 16 bytes per lane-instruction.

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
  const uint32_t meta = ra.z, nh = ra.w, hi = ((const uint32_t *)(B.r_rec + r))[3] & RR_HI;   // (a detail row's HI has no flag bits above it)
  uint32_t n_cig = meta & RM_NCIG;
  // more than 65535 ops: what htslib's bam_write1 does -- the CIGAR field holds <l_seq>S<ref_len>N, the ops follow the
  // aux area as a CG:B,I tag
  const bool spill = n_cig > 65535u;
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
      const int32_t lq = B.l_qseq[rec_input(B, r)];
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
    W4 h1; h1.a = (spill ? 2u : n_cig) | (flag << 16); h1.b = (uint32_t)l_seq; h1.c = (uint32_t)mtid; h1.d = (uint32_t)mpos;
    *(W4 *)(out + 16) = h1;
  } else if (lane == 2) {
    *(u32u *)(out + 32) = (uint32_t)tlen;
  }
  uint32_t o = 36;
  copy_fwd<G>(out + o, rec + 32, l_qname, lane);                            // read name
  o += l_qname;
  // rewritten CIGAR (op order reversed on '-', bam.cpp:688-695)
  const uint2 c = B.r_c[r];   // the ops themselves (<= 2), or their offset in the pool
  const uint32_t *cgp = B.pool + (((uint64_t)c.y << 32) | c.x);
  if (spill) {
    uint32_t part = 0;        // bam_cigar2rlen: M, D, N, =, X consume the reference
    for (uint32_t k = lane; k < n_cig; k += G) { const uint32_t w = cgp[k], op = w & 0xfu; if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) part += w >> 4; }
#pragma unroll
    for (int d = G / 2; d; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, G);
    if (part >= (1u << 28) && lane == 0) *B.too_long = 1;   // bam_write1 fails on such a record
    if (lane == 0) { *(u32u *)(out + o) = ((uint32_t)l_seq << 4) | 4u; *(u32u *)(out + o + 4) = (part << 4) | 3u; }
    o += 8;
  } else {
    if (n_cig <= 2u) {
      if (lane < (int)n_cig) *(u32u *)(out + o + 4 * lane) = ((minus ? n_cig - 1 - lane : lane) == 0) ? c.x : c.y;
    } else {
      for (uint32_t k = lane; k < n_cig; k += G) *(u32u *)(out + o + 4 * k) = cgp[minus ? n_cig - 1 - k : k];
    }
    o += 4 * n_cig;
  }
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
  bool rev_q = minus && ls > 0 && (x.qual_present & 1u);
  if (rev_q) copy_rev<G>(out + o, qual, ls, lane); else copy_fwd<G>(out + o, qual, ls, lane);
  o += ls;
  // aux: original minus the first NH, XS|ts, HI (and AS for long reads): up to five kept pieces ...
  const uint8_t *aux = rec + x.aux_start;
  uint32_t src = 0;
  // ... and minus the CG:B,I tag of a record whose CIGAR was restored from it (bam_read1 drops it): a fifth interval
  uint32_t cg_off = 0xffffffffu;
  if (x.cg_len) { CgTag t; if (cg_find(rec, (uint64_t)x.aux_start + x.aux_len, l_qname, n_cig_in, l_seq, t)) cg_off = t.tag_at - x.aux_start; }
  uint32_t k4 = 0;
  for (int k = 0; k < 5; k++) {   // the four sorted intervals with the tag's merged in where it belongs
    uint32_t ko = k4 < 4u ? x.off[k4] : 0xffffffffu, kl = k4 < 4u ? x.len[k4] : 0u;
    if (cg_off < ko) { ko = cg_off; kl = x.cg_len; cg_off = 0xffffffffu; } else k4++;
    if (ko == 0xffffffffu) break;
    const uint32_t piece = ko - src;
    copy_fwd<G>(out + o, aux + src, piece, lane);
    o += piece; src = ko + kl;
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
    o += 7u * (uint32_t)n_tags;
  }
  if (spill) {   // "CG" 'B' 'I' count ops (op order reversed on '-', like the CIGAR field)
    if (lane == 0) { *(u32u *)(out + o) = (uint32_t)'C' | ((uint32_t)'G' << 8) | ((uint32_t)'B' << 16) | ((uint32_t)'I' << 24); *(u32u *)(out + o + 4) = n_cig; }
    o += 8;
    for (uint32_t k = lane; k < n_cig; k += G) *(u32u *)(out + o + 4 * k) = cgp[minus ? n_cig - 1 - k : k];
  }
}

template <int G>
__global__ void __launch_bounds__(256) k_bam_encode(BamArgs B) {
  const int lane = threadIdx.x & (G - 1);
  int64_t r = (int64_t)blockIdx.x * (256 / G) + (threadIdx.x / G);
  if (r >= B.n_rows) return;
  int32_t a = (int32_t)rec_input(B, r);
  encode_row<G>(B, r, lane, B.blob + B.rec_off[a], B.aux[a], B.out + B.out_off[r], (uint32_t)(B.out_off[r + 1] - B.out_off[r]));
}

enum { BM_SKIP = 0, BM_COPY = 1, BM_REV = 2, BM_REVC_CLEAN = 3, BM_REVC = 4, BM_REVC_ODD = 5 };
#define BM_MODE_SHIFT 60             // src: mode in bits 60..62
#define BM_OFF ((1ull << 48) - 1)    // src: offset in the blob of the segment's first source byte

// 16 source bytes at p; bytes at or after `end` read as 0 (only a task at the very end of the blob takes the byte path)
__device__ __forceinline__ uint4 load16_end(const uint8_t *p, const uint8_t *end) {
  if (p + 16 <= end) { W4 w = *(const W4 *)p; return make_uint4(w.a, w.b, w.c, w.d); }
  uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 16; j++) if (p + j < end) v[j >> 2] |= (uint32_t)p[j] << (8 * (j & 3));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------------------
// k_bam_tasks<R>: one wave encodes R consecutive rows; the byte regions of all of them are cut into 16-byte copy
// tasks that fill the wave's lanes densely.  (k_bam_encode<G> gives a row G lanes and walks its regions one after the
// other: ~43 vector-memory instructions per wave of 8 rows, most with 2-7 of a row's 8 lanes at work, every region's
// loads waiting for its predecessor's stores.)
//   1. lane i < R describes row i (what encode_row computes per row, once instead of G times) and lists the row's copy segments -- read name, SEQ, QUAL, kept aux
//      pieces; a copy that continues its predecessor's source and destination joins it -- with where the bytes go, where
//      they come from, how (copy, byte-reversed, reverse-complemented) and how many tasks that is: ceil(len / 16), the
//      last one ending at the segment's end (it overlaps its predecessor) instead of a byte tail; one task for a segment
//      under 16 bytes.  A wave-wide scan of the rows' task counts numbers the tasks.
//   2. task t -> row (binary search over the rows' first tasks) -> segment (over the segments' first tasks) -> chunk:
//      one 16-byte load, the turn-around if any, one 16-byte store; no masks, no chunk is shared between segments.
//   3. lane i writes row i's fixed fields, <= 2-op CIGAR and tags; longer CIGARs are copied from the arena by the wave.
#define BT_SEGS 8   // name, seq, qual, 5 aux pieces
struct __attribute__((aligned(16))) BamTaskRow {
  uint32_t tp[12];                   // first task of segment s (row-relative); tp[n_seg] = the row's tasks, then ~0
  uint32_t dst[BT_SEGS], len[BT_SEGS];
  uint64_t src[BT_SEGS];             // BM_OFF | mode << BM_MODE_SHIFT
  uint32_t cig_at, cig_n;            // > 2-op CIGAR: span-relative output offset, n_cigar | minus << 31
  uint64_t cig_src;                  // ... and its word offset in the arena
};
template <int R> struct __attribute__((aligned(16))) BamTaskLds { BamTaskRow d[R]; uint32_t row_tp[R + 4], row_at[R + 4]; };

template <int R>
__global__ void __launch_bounds__(256) k_bam_tasks(BamArgs B) {
  __shared__ BamTaskLds<R> sh_w[4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  BamTaskLds<R> &L = sh_w[wv];
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * R;
  if (r0 >= B.n_rows) return;
  const int nr = (int)(B.n_rows - r0 < R ? B.n_rows - r0 : R);
  const uint64_t span0 = B.out_off[r0];
  uint64_t be = B.blob_end[(size_t)(lane & (BLOB_END_SLOTS - 1)) * BLOB_END_STRIDE];   // the maximum over k_bam_scan's slots
#pragma unroll
  for (int o = 32; o; o >>= 1) { const uint64_t t = __shfl_xor(be, o); be = t > be ? t : be; }
  const uint8_t *blob_end = B.blob + be;

  // ---- 1. one lane per row ----
  uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0, h8 = 0;   // block_size + the 32 fixed bytes
  uint32_t row_at = 0, o_cig = 0, o_tags = 0, n_cig = 0, c_x = 0, c_y = 0;
  uint32_t tag_nh = 0, tag_hi = 0, tag_as = 0, row_tasks = 0;
  bool minus = false;
  bool slow = false;   // a record whose CIGAR came out of a CG tag, or a row whose CIGAR must go into one: encode_row, below
  if (lane < nr) {
    const int64_t r = r0 + lane;
    const uint4 ra = B.r_a[r];
    const uint2 c = B.r_c[r];
    const uint64_t oo = B.out_off[r];
    const uint32_t total = (uint32_t)(B.out_off[r + 1] - oo);
    const int32_t a = (int32_t)rec_input(B, r);
    const BamAux x = B.aux[a];
    const uint64_t ro = B.rec_off[a];
    row_at = (uint32_t)(oo - span0);
    const uint32_t l_qname = x.c_a & 0xffu, bin = x.c_a >> 16, n_cig_in = x.c_b & 0xffffu;
    uint32_t flag = x.c_b >> 16;
    const int32_t l_seq = (int32_t)x.c_c;
    const uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0, sb = (ls + 1) / 2;
    const uint32_t meta = ra.z, nh = ra.w;
    n_cig = meta & RM_NCIG; minus = meta & RM_MINUS; c_x = c.x; c_y = c.y;
    const bool paired = meta & RM_PAIRED, same = meta & RM_SAME;
    // flags: secondary (src/core.cpp:142-143), reverse (bam.cpp:698), mate bits (bam.cpp:531-588)
    if (meta & RM_PRIMARY) flag &= ~0x100u; else flag |= 0x100u;
    if (minus) flag ^= 0x10u;
    int32_t mtid = -1, mpos = -1, tlen = 0;
    if (!paired) flag &= ~(0x1u | 0x2u | 0x20u);
    else {
      flag |= 0x1u;
      if (minus) flag |= 0x20u;
      const uint4 rb = B.r_a[(meta & RM_FIRST) ? r + 1 : r - 1];
      const int32_t my_pos = (int32_t)ra.y;
      mpos = (int32_t)rb.y;
      if (same) {
        flag |= 0x2u; mtid = (int32_t)ra.x;
        const int32_t lq = B.l_qseq[a];
        tlen = (my_pos <= mpos) ? (mpos + lq) - my_pos : -((my_pos + lq) - mpos);
      } else { flag &= ~0x2u; mtid = (int32_t)rb.x; }
    }
    const uint32_t mapq = B.long_reads ? (nh > 1 ? 0u : 3u) : (nh == 1 ? 255u : nh == 2 ? 3u : (nh == 3 || nh == 4) ? 1u : 0u);  // get_mapq (src/core.cpp:46-58)
    h0 = total - 4u; h1 = ra.x; h2 = ra.y; h3 = l_qname | (mapq << 8) | (bin << 16);
    h4 = (n_cig & 0xffffu) | (flag << 16); h5 = (uint32_t)l_seq; h6 = (uint32_t)mtid; h7 = (uint32_t)mpos; h8 = (uint32_t)tlen;
    tag_nh = nh; tag_hi = ((const uint32_t *)(B.r_rec + r))[3] & RR_HI;
    if (B.long_reads) tag_as = (uint32_t)(int32_t)(((double)x.as_val + (double)(B.r_clip ? B.r_clip[r] : 0)) * (B.r_sim ? B.r_sim[r] : 0.0));  // set_as_tag
    slow = x.cg_len != 0u || n_cig > 65535u;
    BamTaskRow &D = L.d[lane];
    int ns = 0;
    uint32_t o = 36u, prev_mode = 0xffu, prev_end = 0, prev_len = 0, prev_o = 0;
#define BT_TASKS(n) ((n) >= 16u ? ((n) + 15u) >> 4 : 1u)
#define BT_SEG(len_, mode_, src_) do {                                                                                  \
      const uint32_t sl = (len_), sm = (mode_), ss = (src_);                                                            \
      if (sl != 0u) {                                                                                                   \
        if (sm == BM_COPY && prev_mode == BM_COPY && prev_end == ss && prev_o == o) {                                   \
          prev_len += sl; prev_end += sl; prev_o += sl; o += sl;                                                        \
          D.len[ns - 1] = prev_len;                                                                                     \
          row_tasks = prev_tp + BT_TASKS(prev_len);                                                                     \
        } else {                                                                                                        \
          D.tp[ns] = row_tasks; D.dst[ns] = o; D.len[ns] = sl; D.src[ns] = (ro + ss) | ((uint64_t)sm << BM_MODE_SHIFT); \
          prev_tp = row_tasks; row_tasks += BT_TASKS(sl); ns++; o += sl;                                                \
          prev_mode = sm; prev_end = ss + sl; prev_len = sl; prev_o = o;                                                \
        }                                                                                                               \
      }                                                                                                                 \
    } while (0)
    uint32_t prev_tp = 0;
    if (!slow) {
    BT_SEG(l_qname, BM_COPY, 32u);
    o_cig = o;
    o += 4u * n_cig;
    const uint32_t seq_at = 32u + l_qname + 4u * n_cig_in;
    BT_SEG(sb, !minus ? BM_COPY : (ls & 1u) ? BM_REVC_ODD : (x.qual_present & 2u) ? BM_REVC_CLEAN : BM_REVC, seq_at);
    BT_SEG(ls, (minus && (x.qual_present & 1u)) ? BM_REV : BM_COPY, seq_at + sb);
    uint32_t src = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (x.off[k] == 0xffffffffu) break;
      BT_SEG(x.off[k] - src, BM_COPY, x.aux_start + src);
      src = x.off[k] + x.len[k];
    }
    BT_SEG(x.aux_len - src, BM_COPY, x.aux_start + src);
    }
    o_tags = o;
    D.tp[ns] = row_tasks;
    for (int k = ns + 1; k < 12; k++) D.tp[k] = 0xffffffffu;
    D.cig_at = row_at + o_cig; D.cig_n = n_cig | (minus ? 0x80000000u : 0u);
    D.cig_src = ((uint64_t)c.y << 32) | c.x;
  }
  static_assert(R + 4 <= 64 && (R & (R - 1)) == 0, "rows per wave");
  // first task of every row: a scan over the wave
  uint32_t incl = row_tasks;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
  const uint32_t n_tasks = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(incl, 63));
  if (lane < R + 4) { L.row_tp[lane] = lane < nr ? incl - row_tasks : 0xffffffffu; L.row_at[lane] = row_at; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  // ---- 2. one lane per copy task, two tasks in flight: the next task is looked up and its load issued before the
  //         current one is finished, so that neither the look-up (two chains of LDS reads) nor the store sit between a
  //         load and the wait for it ----
  uint8_t *const out = B.out + span0;
  struct Task { const uint8_t *p; uint8_t *d; uint32_t len, at, mode; bool on; };
  auto find = [&](uint32_t t) __attribute__((always_inline)) {
    Task S;
    S.on = t < n_tasks;
    const uint32_t tt = S.on ? t : 0u;
    int i = 0;
#pragma unroll
    for (int st = R / 2; st; st >>= 1) if (L.row_tp[i + st] <= tt) i += st;
    const BamTaskRow &D = L.d[i];
    const uint32_t t1 = tt - L.row_tp[i];
    int s = 0;
#pragma unroll
    for (int st = 4; st; st >>= 1) if (D.tp[s + st] <= t1) s += st;
    const uint32_t j = t1 - D.tp[s];
    const uint64_t sd = D.src[s];
    S.len = D.len[s];
    S.mode = (uint32_t)(sd >> BM_MODE_SHIFT) & 7u;
    // a segment of 16 bytes or more: chunk j, the last one ending at the segment's end.  A shorter one: the 16 bytes that
    // start (copy) or end (the modes that read backwards) with it, of which the first len are stored.
    S.at = S.len < 16u ? 0u : 16u * j + 16u <= S.len ? 16u * j : S.len - 16u;
    // copy: bytes [at, at + 16).  The others read backwards: output byte u <- source byte len - 1 - u
    S.p = B.blob + (sd & BM_OFF) + (int32_t)(S.mode == BM_COPY ? S.at : S.len - 16u - S.at);
    S.d = out + L.row_at[i] + D.dst[s];
    return S;
  };
  // (a copy of fewer than 16 bytes may reach past the blob's last record: that one is loaded again, byte by byte, in finish)
  auto past_end = [&](const Task &S) __attribute__((always_inline)) { return S.len < 16u && S.mode == BM_COPY && S.p + 16 > blob_end; };
  auto fetch = [&](const Task &S) __attribute__((always_inline)) {
    const W4 ld = *(const W4 *)(past_end(S) ? B.blob : S.p);
    return make_uint4(ld.a, ld.b, ld.c, ld.d);
  };
  // "the data of the current task is needed here": placed right behind the next task's load, in straight-line code, where
  // the compiler counts exactly one younger operation (s_waitcnt vmcnt(1)); left to the first real use, behind the branches
  // of finish, its count ends in "wait for everything", the next task's load included
  auto landed = [&](uint4 v) __attribute__((always_inline)) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
  };
  auto finish = [&](const Task &S, uint4 v) __attribute__((always_inline)) {
    if (!S.on) return;
    const uint32_t mode = S.mode, len = S.len;
    if (past_end(S)) v = load16_end(S.p, blob_end);
    if (len < 16u && mode == BM_REVC_ODD) {   // fewer than 31 bases, odd: byte by byte
      const uint8_t *src = S.p + 16 - len;
      const uint32_t ls = 2u * len - 1u;
      for (uint32_t b = 0; b < len; b++) {
        const uint32_t s0 = ls - 1u - 2u * b;
        uint8_t q = (uint8_t)(comp4((src[s0 >> 1] >> ((~s0 & 1u) << 2)) & 0xf) << 4);
        if (2u * b + 1u < ls) { const uint32_t s1 = s0 - 1u; q |= comp4((src[s1 >> 1] >> ((~s1 & 1u) << 2)) & 0xf); }
        S.d[b] = q;
      }
      return;
    }
    if (mode != BM_COPY) {
      // qualities: bytes reversed; bases: all 128 bits reversed = base order, and A<->T, C<->G inside every nibble (bam.cpp:671-686)
      const bool bits = mode != BM_REV;
      const uint4 u = v;
      v.x = bits ? __builtin_bitreverse32(u.w) : __builtin_bswap32(u.w); v.y = bits ? __builtin_bitreverse32(u.z) : __builtin_bswap32(u.z);
      v.z = bits ? __builtin_bitreverse32(u.y) : __builtin_bswap32(u.y); v.w = bits ? __builtin_bitreverse32(u.x) : __builtin_bswap32(u.x);
      if (mode >= BM_REVC) {   // a base code other than A C G T N in the record, or an odd length
        if (mode == BM_REVC_ODD) {
          // odd length: the stream moves up by one nibble; the low nibble of output byte u is the high nibble of the next
          const W4 l2 = *(const W4 *)(S.p - 1);
          const uint32_t y0 = __builtin_bitreverse32(l2.d), y1 = __builtin_bitreverse32(l2.c), y2 = __builtin_bitreverse32(l2.b), y3 = __builtin_bitreverse32(l2.a);
          v.x = ((v.x & 0x0f0f0f0fu) << 4) | ((y0 >> 4) & 0x0f0f0f0fu); v.y = ((v.y & 0x0f0f0f0fu) << 4) | ((y1 >> 4) & 0x0f0f0f0fu);
          v.z = ((v.z & 0x0f0f0f0fu) << 4) | ((y2 >> 4) & 0x0f0f0f0fu); v.w = ((v.w & 0x0f0f0f0fu) << 4) | ((y3 >> 4) & 0x0f0f0f0fu);
        }
        v.x = fix_nib(v.x); v.y = fix_nib(v.y); v.z = fix_nib(v.z); v.w = fix_nib(v.w);
        if (mode == BM_REVC_ODD && S.at + 16u == len) v.w &= 0xf0ffffffu;   // the pad nibble of the last byte stays 0
      }
    }
    if (len >= 16u) {
      // (the output records are written once: streaming stores, the 24 GB do not displace the input records, which
      // about five rows each come back to)
      typedef uint32_t ntw4 __attribute__((ext_vector_type(4), aligned(1)));
      const ntw4 o4 = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(o4, (ntw4 *)(S.d + S.at));
    } else {
      typedef uint16_t u16u __attribute__((aligned(1)));
      struct __attribute__((packed, aligned(1))) W2 { uint32_t a, b; };
      uint8_t *d = S.d;
      if (len & 8u) { W2 o2; o2.a = v.x; o2.b = v.y; *(W2 *)d = o2; v.x = v.z; v.y = v.w; d += 8; }
      if (len & 4u) { *(u32u *)d = v.x; v.x = v.y; d += 4; }
      if (len & 2u) { *(u16u *)d = (uint16_t)v.x; v.x >>= 16; d += 2; }
      if (len & 1u) *d = (uint8_t)v.x;
    }
  };
  if (n_tasks) {
    // (two copies of the loop body, the tasks alternating between two sets of registers: a register copy of the next
    // task's data at the end of a round would wait for its load)
    // (two copies of the loop body, the tasks alternating between two sets of registers; no branch between a task's load
    // and the "landed" of the task before it)
    Task S0 = find(lane), S1;
    uint4 v0 = fetch(S0), v1;
    for (uint32_t base = 0;; base += 128) {
      if (base + 64u >= n_tasks) { finish(S0, v0); break; }   // the same for every lane
      S1 = find(base + 64u + lane); v1 = fetch(S1);
      finish(S0, landed(v0));
      if (base + 128u >= n_tasks) { finish(S1, v1); break; }
      S0 = find(base + 128u + lane); v0 = fetch(S0);
      finish(S1, landed(v1));
    }
  }

  // ---- 3. the synthesized bytes ----
  if (lane < nr && !slow) {
    uint8_t *w = out + row_at;
    W4 x0; x0.a = h0; x0.b = h1; x0.c = h2; x0.d = h3; *(W4 *)w = x0;
    W4 x1; x1.a = h4; x1.b = h5; x1.c = h6; x1.d = h7; *(W4 *)(w + 16) = x1;
    *(u32u *)(w + 32) = h8;
    // rewritten CIGAR (op order reversed on '-', bam.cpp:688-695)
    if (n_cig == 1u) *(u32u *)(w + o_cig) = c_x;
    else if (n_cig == 2u) { *(u32u *)(w + o_cig) = minus ? c_y : c_x; *(u32u *)(w + o_cig + 4) = minus ? c_x : c_y; }
    // NH:i, (AS:i,) HI:i appended in that order (bam.cpp:590-634, core.cpp:118-161): seven bytes [t0 t1 'i' v0 v1 v2 v3] each
    uint8_t *t = w + o_tags;
    *(u32u *)t = (uint32_t)'N' | ((uint32_t)'H' << 8) | ((uint32_t)'i' << 16) | (tag_nh << 24);
    *(u32u *)(t + 3) = tag_nh;
    if (B.long_reads) {
      *(u32u *)(t + 7) = (uint32_t)'A' | ((uint32_t)'S' << 8) | ((uint32_t)'i' << 16) | (tag_as << 24);
      *(u32u *)(t + 10) = tag_as;
      t += 7;
    }
    *(u32u *)(t + 7) = (uint32_t)'H' | ((uint32_t)'I' << 8) | ((uint32_t)'i' << 16) | (tag_hi << 24);
    *(u32u *)(t + 10) = tag_hi;
  }
  // CIGARs of more than two ops: the whole wave, row by row
  uint64_t big = __ballot(lane < nr && n_cig > 2u && !slow);
  while (big) {
    const int i = __builtin_ctzll(big);
    big &= big - 1;
    const uint32_t at = L.d[i].cig_at, cn = L.d[i].cig_n, n = cn & 0x7fffffffu;
    const uint32_t *cg = B.pool + L.d[i].cig_src;
    for (uint32_t k2 = lane; k2 < n; k2 += 64) *(u32u *)(out + at + 4u * k2) = cg[(cn >> 31) ? n - 1u - k2 : k2];
  }
  // rows around a CG tag (ultra-long reads: rare), row by row with the whole wave: the generic encoder
  uint64_t slowm = __ballot(lane < nr && slow);
  while (slowm) {
    const int i = __builtin_ctzll(slowm);
    slowm &= slowm - 1;
    const int64_t r = r0 + i;
    const int32_t a = (int32_t)rec_input(B, r);
    encode_row<64>(B, r, lane, B.blob + B.rec_off[a], B.aux[a], B.out + B.out_off[r], (uint32_t)(B.out_off[r + 1] - B.out_off[r]));
  }
}

void launch_bam_scan(hipStream_t st, const BamArgs &B) {
  if (B.n_aln > 0) hipLaunchKernelGGL(k_bam_scan, dim3((unsigned)((B.n_aln + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_size(hipStream_t st, const BamArgs &B) {
  if (B.n_rows > 0) hipLaunchKernelGGL(k_bam_size, dim3((unsigned)((B.n_rows + 255) / 256)), dim3(256), 0, st, B);
}
void launch_bam_encode(hipStream_t st, const BamArgs &B, int lanes) {
  if (B.n_rows <= 0) return;
  if (lanes == 0) { hipLaunchKernelGGL((k_bam_tasks<32>), dim3((unsigned)((B.n_rows + 127) / 128)), dim3(256), 0, st, B); return; }
  if (lanes == 4) hipLaunchKernelGGL((k_bam_encode<4>), dim3((unsigned)((B.n_rows + 63) / 64)), dim3(256), 0, st, B);
  else if (lanes == 8) hipLaunchKernelGGL((k_bam_encode<8>), dim3((unsigned)((B.n_rows + 31) / 32)), dim3(256), 0, st, B);
  else if (lanes == 32) hipLaunchKernelGGL((k_bam_encode<32>), dim3((unsigned)((B.n_rows + 7) / 8)), dim3(256), 0, st, B);
  else if (lanes == 64) hipLaunchKernelGGL((k_bam_encode<64>), dim3((unsigned)((B.n_rows + 3) / 4)), dim3(256), 0, st, B);
  else hipLaunchKernelGGL((k_bam_encode<16>), dim3((unsigned)((B.n_rows + 15) / 16)), dim3(256), 0, st, B);
}
size_t bam_aux_bytes() { return sizeof(BamAux); }

}  // namespace br
