// Per-record work of the reader side, written over a record accessor (u8 / u16 / u32 / w4 at a record-relative offset)
// instead of raw pointers.  GlobalRec reads the record where it lies; a second accessor over records staged in LDS by the
// wave (one fused kernel for both functions) was built and measured in round 3 and did not pay.
//
//   rec_fields_one   fixed fields -> the projection's input columns, "starts a new read name"    (see parse_kernels.hip)
//   bam_scan_one     aux walk -> BamAux (removal intervals, tag characters, sequence class)      (see bam_kernels.hip)
//
// An accessor R has u8(off) / u16(off) / u32(off) / w4(off) with `off` relative to the record's refID field.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace br {

struct RecW4 { uint32_t a, b, c, d; };

struct GlobalRec {
  typedef uint64_t off_t;   // offsets inside a record (a record may be megabytes: a CG tag)
  const uint8_t *p;
  typedef uint32_t u32u __attribute__((aligned(1)));
  typedef uint16_t u16u __attribute__((aligned(1)));
  struct __attribute__((packed, aligned(1))) W4u { uint32_t a, b, c, d; };
  __device__ __forceinline__ uint32_t u8(uint64_t o) const { return p[o]; }
  __device__ __forceinline__ uint32_t u16(uint64_t o) const { return *(const u16u *)(p + o); }
  __device__ __forceinline__ uint32_t u32(uint64_t o) const { return *(const u32u *)(p + o); }
  __device__ __forceinline__ RecW4 w4(uint64_t o) const { const W4u v = *(const W4u *)(p + o); return RecW4{v.a, v.b, v.c, v.d}; }
};

// htslib skip_aux: size of the value of a tag of `type` whose first value byte is at `p` (record-relative), or -1
template <class R>
__device__ inline int64_t rec_aux_value_len(const R &rec, uint32_t type, typename R::off_t p, typename R::off_t end) {
  typedef typename R::off_t O;
  switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    case 'Z': case 'H': {   // up to and including the NUL: four bytes per load while four are left
      O q = p;
      while (end - q >= 4) {
        const uint32_t w = rec.u32(q), z = (w - 0x01010101u) & ~w & 0x80808080u;
        if (z) return (int64_t)(q - p) + (__builtin_ctz(z) >> 3) + 1;
        q += 4;
      }
      while (q < end && rec.u8(q)) q++;
      return q < end ? (int64_t)(q - p) + 1 : -1;
    }
    case 'B': {
      if (end - p < 5) return -1;
      const uint32_t st = rec.u8(p), n = rec.u32(p + 1);
      const int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : (st == 'i' || st == 'I' || st == 'f') ? 4 : 0;
      if (!sz) return -1;
      return 5 + (int64_t)n * sz;
    }
    default: return -1;
  }
}

// bam_cg.h's cg_candidate / cg_find over an accessor
template <class R>
__device__ __forceinline__ bool rec_cg_candidate(const R &rec, uint64_t rlen, uint32_t l_qname, uint32_t n_cig, int32_t l_seq) {
  if (n_cig == 0 || rlen < 32 || 32ull + l_qname + 4ull * n_cig > rlen) return false;   // (n_cig <= 65535: 32 bits hold the sum)
  if ((int32_t)rec.u32(0) < 0 || (int32_t)rec.u32(4) < 0) return false;   // tid, pos
  const uint32_t w0 = rec.u32(32 + l_qname);
  return (w0 & 0xfu) == 4u && (w0 >> 4) == (uint32_t)l_seq;
}
template <class R>
__device__ inline bool rec_cg_find(const R &rec, uint64_t rlen, uint32_t l_qname, uint32_t n_cig, int32_t l_seq, uint32_t &tag_at, uint32_t &n_ops) {
  if (!rec_cg_candidate(rec, rlen, l_qname, n_cig, l_seq)) return false;
  typedef typename R::off_t O;
  const uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
  const uint64_t start = 32ull + l_qname + 4ull * n_cig + (ls + 1) / 2 + ls;
  if (start > rlen) return false;
  O s = (O)start; const O end = (O)rlen;
  while (end - s >= 3) {
    const int64_t vl = rec_aux_value_len(rec, rec.u8(s + 2), (O)(s + 3), end);
    if (vl < 0 || (uint64_t)s + 3 + (uint64_t)vl > (uint64_t)end) return false;   // malformed: the walk stops, the tag was not found
    if (rec.u8(s) == 'C' && rec.u8(s + 1) == 'G') {
      if (end - s < 8 || rec.u8(s + 2) != 'B' || !(rec.u8(s + 3) == 'I' || rec.u8(s + 3) == 'i')) return false;
      const uint32_t n = rec.u32(s + 4);
      if (n < n_cig || n >= (1u << 29)) return false;
      tag_at = (uint32_t)s; n_ops = n;
      return true;
    }
    s += (O)(3 + vl);
  }
  return false;
}

// ---- k_rec_fields, one record ---------------------------------------------------------------------------------------
// `prev` / `plen`: the record in front (ignored when i == 0).  ncig / max_s: this record's contribution to the batch maxima.
template <class R>
__device__ __forceinline__ void rec_fields_one(const ParseArgs &P, int64_t i, const R &rec, uint32_t rlen, const R &prev, uint32_t plen,
                                               uint32_t &ncig, uint32_t &max_s) {
  int32_t ref = -1, start = 0, lq = 0;
  uint32_t flag = 0, nlen = 0, isnew = 1;
  ncig = 0; max_s = 0;
  if (rlen >= 32) {
    const RecW4 w0 = rec.w4(0);          // refID, pos, l_read_name|mapq|bin, n_cigar_op|flag
    const int32_t l_seq = (int32_t)rec.u32(16);
    const int32_t raw_ref = (int32_t)w0.a;
    ref = (raw_ref >= 0 && raw_ref < P.n_ref_map) ? P.ref_map[raw_ref] : -1;
    start = (int32_t)w0.b + 1;           // GSamRecord::start is 1-based
    uint32_t l_qname = w0.c & 0xffu;
    ncig = w0.d & 0xffffu; flag = w0.d >> 16;
    if (32ull + l_qname + 4ull * ncig > rlen) { ncig = 0; l_qname = 0; }  // br_bam_split rejects these; stay in bounds anyway
    nlen = l_qname ? l_qname - 1 : 0;    // without the NUL
    lq = l_seq;
    // leading / trailing soft clips (sizing of the rescue buffers): S is only legal next to the ends (after H)
    typename R::off_t cg = 32 + l_qname;
    // a CIGAR spilled into a CG:B,I tag (more than 65535 ops): the real ops are the tag's array (bam_cg.h)
    if (rec_cg_candidate(rec, rlen, l_qname, ncig, l_seq)) { uint32_t at, n; if (rec_cg_find(rec, rlen, l_qname, ncig, l_seq, at, n)) { ncig = n; cg = (typename R::off_t)at + 8; } }
    if (ncig) {
      uint32_t w = rec.u32(cg);
      if ((w & 0xfu) == 5u && ncig > 1) w = rec.u32(cg + 4);
      if ((w & 0xfu) == 4u) max_s = w >> 4;
      w = rec.u32(cg + (typename R::off_t)4 * (ncig - 1));
      if ((w & 0xfu) == 5u && ncig > 1) w = rec.u32(cg + (typename R::off_t)4 * (ncig - 2));
      if ((w & 0xfu) == 4u) max_s = max(max_s, w >> 4);
    }
    if (i > 0 && plen >= 32) {
      const uint32_t pw = prev.u32(8), pc = prev.u16(12);
      uint32_t pl = pw & 0xffu;
      if (32ull + pl + 4ull * pc > plen) pl = 0;
      const uint32_t pn = pl ? pl - 1 : 0;
      if (pn == nlen) {   // names compared four bytes at a time, tail bytewise
        bool same = true;
        uint32_t k = 0;
        for (; same && k + 4 <= nlen; k += 4) same = prev.u32(32 + k) == rec.u32(32 + k);
        for (; same && k < nlen; k++) same = prev.u8(32 + k) == rec.u8(32 + k);
        if (same) isnew = 0;
      }
    }
  }
  P.ref_id[i] = ref; P.ref_start[i] = start; P.flags[i] = (uint16_t)flag; P.l_qseq[i] = lq;
  P.ncig[i] = ncig; P.name_len[i] = nlen; P.isnew[i] = isnew;
}

// ---- k_bam_scan, one record -----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rec_fix_nib(uint32_t x) {   // nibbles that are not one of 1, 2, 4, 8 become 15 (comp_table, src/bam.cpp:658-667)
  uint32_t pop = (x & 0x11111111u) + ((x >> 1) & 0x11111111u) + ((x >> 2) & 0x11111111u) + ((x >> 3) & 0x11111111u);
  uint32_t y = pop ^ 0x11111111u;
  uint32_t bad = (y | (y >> 1) | (y >> 2) | (y >> 3)) & 0x11111111u;
  return x | (bad * 15u);
}

// what a record contributes to the length of each of its output rows, whatever their CIGAR (k_bam_scan leaves it per
// record, so that k_bam_size reads four bytes per row instead of the aux table's 64)
__device__ __forceinline__ uint32_t row_base_len(const BamArgs &B, const BamAux &x) {
  uint32_t l_qname = x.c_a & 0xffu;
  int32_t l_seq = (int32_t)x.c_c;
  uint32_t ls = l_seq > 0 ? (uint32_t)l_seq : 0;
  uint32_t removed = x.len[0] + x.len[1] + x.len[2] + x.len[3] + x.cg_len;
  uint32_t added = 7u + 7u + (B.long_reads ? 7u : 0u);  // NH:i, HI:i, AS:i
  return 4u + 32u + l_qname + (ls + 1) / 2 + ls + (x.aux_len - removed) + added;
}

template <class R>
__device__ __forceinline__ void bam_scan_one(const BamArgs &B, int64_t a, const R &rec, uint64_t rlen) {
  typedef typename R::off_t O;
  BamAux x;
  int8_t xs_c = 0, ts_c = 0;
  bool have_xs = false, have_ts = false;
  for (int k = 0; k < 4; k++) { x.off[k] = 0xffffffffu; x.len[k] = 0; }
  x.as_val = 0; x.aux_start = 0; x.aux_len = 0; x.c_a = x.c_b = x.c_c = 0; x.qual_present = 0; x.cg_len = 0;
  if (rlen >= 32) {
    x.c_a = rec.u32(8); x.c_b = rec.u32(12);
    const uint32_t l_qname = x.c_a & 0xffu, n_cig = x.c_b & 0xffffu;
    const int32_t l_seq = (int32_t)rec.u32(16);
    const uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
    const uint64_t start = 32 + (uint64_t)l_qname + 4ull * n_cig + (ls + 1) / 2 + ls;
    x.c_c = (uint32_t)l_seq;
    if (start <= rlen) {
      if (ls > 0) x.qual_present = rec.u8((O)(32 + (uint64_t)l_qname + 4ull * n_cig + (ls + 1) / 2)) != 0xff;
      {
        // bit 1: every base code is one of 1, 2, 4, 8, 15 (A C G T N) -- then the reverse complement of the record is a
        // plain bit reversal and k_bam_tasks skips the per-nibble repair of the other codes
        const O sq = (O)(32 + (uint64_t)l_qname + 4ull * n_cig);
        const O sb = (O)((ls + 1) / 2), full = (O)(ls / 2);   // bytes with two bases
        uint32_t dirty = 0;
        O i = 0;
        for (; i + 16 <= full; i += 16) {
          const RecW4 v = rec.w4(sq + i);
          dirty |= (rec_fix_nib(v.a) ^ v.a) | (rec_fix_nib(v.b) ^ v.b) | (rec_fix_nib(v.c) ^ v.c) | (rec_fix_nib(v.d) ^ v.d);
        }
        for (; i + 4 <= full; i += 4) { uint32_t v = rec.u32(sq + i); dirty |= rec_fix_nib(v) ^ v; }
        for (; i < full; i++) { uint32_t v = rec.u8(sq + i) | 0x11111100u; dirty |= rec_fix_nib(v) ^ v; }
        if (sb > full) { uint32_t v = (rec.u8(sq + full) >> 4) | 0x11111110u; dirty |= rec_fix_nib(v) ^ v; }
        if (!dirty) x.qual_present |= 2u;
      }
      x.aux_start = (uint32_t)start; x.aux_len = (uint32_t)(rlen - start);
      O s = (O)start; const O end = (O)rlen;
      // slots: 0 NH, 1 XS (short) / ts (long), 2 HI, 3 AS (long reads only)
      // (the four slots live in named registers: indexing x.off[slot] with a run-time slot puts the structure into scratch
      // memory, and every access there is a trip to global memory in the middle of a dependent walk)
      uint32_t have = 0, o0 = 0xffffffffu, o1 = 0xffffffffu, o2 = 0xffffffffu, o3 = 0xffffffffu, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
      bool have_cg = false, cg_ok = false; uint32_t cg_len = 0;
      while (end - s >= 3) {
        // tag, type and the first value byte in one load (a tag without room for a value ends the walk below)
        const uint32_t tw = end - s >= 4 ? rec.u32(s) : rec.u8(s) | (rec.u8(s + 1) << 8) | (rec.u8(s + 2) << 16);
        const uint32_t t0 = tw & 0xffu, t1 = (tw >> 8) & 0xffu, ty = (tw >> 16) & 0xffu, b3 = tw >> 24;
        const int64_t vl = rec_aux_value_len(rec, ty, (O)(s + 3), end);
        if (vl < 0 || (uint64_t)s + 3 + (uint64_t)vl > (uint64_t)end) break;  // malformed: htslib stops here too
        if (t0 == 'C' && t1 == 'G' && !have_cg) {   // the first CG tag (bam_aux_get): a spilled CIGAR when of type B,I / B,i with enough entries
          have_cg = true;
          if (ty == 'B' && (b3 == 'I' || b3 == 'i') && end - s >= 8) { const uint32_t n = rec.u32(s + 4); cg_ok = n >= n_cig && n < (1u << 29); cg_len = (uint32_t)(3 + vl); }
        }
        // tag_char1 (gclib/GSam.cpp:310-318): first value byte of the first XS / ts tag when A or Z
        if (t0 == 'X' && t1 == 'S' && !have_xs) { have_xs = true; if (ty == 'A' || ty == 'Z') xs_c = (int8_t)b3; }
        if (t0 == 't' && t1 == 's' && !have_ts) { have_ts = true; if (ty == 'A' || ty == 'Z') ts_c = (int8_t)b3; }
        int slot = -1;
        if (t0 == 'N' && t1 == 'H') slot = 0;
        else if (!B.long_reads && t0 == 'X' && t1 == 'S') slot = 1;
        else if (B.long_reads && t0 == 't' && t1 == 's') slot = 1;
        else if (t0 == 'H' && t1 == 'I') slot = 2;
        else if (B.long_reads && t0 == 'A' && t1 == 'S') slot = 3;
        if (slot >= 0 && !((have >> slot) & 1u)) {
          have |= 1u << slot;
          const uint32_t so = (uint32_t)(s - (O)start), sl = (uint32_t)(3 + vl);
          if (slot == 0) { o0 = so; l0 = sl; } else if (slot == 1) { o1 = so; l1 = sl; } else if (slot == 2) { o2 = so; l2 = sl; } else { o3 = so; l3 = sl; }
          if (slot == 3) {  // bam_aux2i
            const O v = s + 3;
            switch (ty) {
              case 'c': x.as_val = (int8_t)rec.u8(v); break;
              case 'C': x.as_val = (int32_t)rec.u8(v); break;
              case 's': x.as_val = (int16_t)rec.u16(v); break;
              case 'S': x.as_val = (int32_t)rec.u16(v); break;
              case 'i': x.as_val = (int32_t)rec.u32(v); break;
              case 'I': x.as_val = (int32_t)rec.u32(v); break;
              default: x.as_val = 0; break;
            }
          }
        }
        s += (O)(3 + vl);
      }
      // A CIGAR of more than 65535 ops lives in a CG:B,I tag behind the placeholder <l_seq>S<ref_len>N (bam_cg.h): htslib's
      // bam_read1 moves it into place and drops the tag, so the tag's bytes leave every output row of this record
      // (encode_row finds the tag again; the fast task kernel hands such rows to it)
      x.cg_len = (cg_ok && rec_cg_candidate(rec, rlen, l_qname, n_cig, l_seq)) ? cg_len : 0u;
      // sort the (at most four) removal intervals by offset (the offsets in use are distinct, the unused ones all ~0 with
      // length 0): five compare-exchanges
#define BR_CEX(oa, la, ob, lb) do { if (ob < oa) { uint32_t t_ = oa; oa = ob; ob = t_; t_ = la; la = lb; lb = t_; } } while (0)
      BR_CEX(o0, l0, o1, l1); BR_CEX(o2, l2, o3, l3); BR_CEX(o0, l0, o2, l2); BR_CEX(o1, l1, o3, l3); BR_CEX(o1, l1, o2, l2);
#undef BR_CEX
      x.off[0] = o0; x.off[1] = o1; x.off[2] = o2; x.off[3] = o3; x.len[0] = l0; x.len[1] = l1; x.len[2] = l2; x.len[3] = l3;
    }
  }
  B.aux[a] = x;
  B.base_len[a] = row_base_len(B, x);
  if (B.xs_out) { B.xs_out[a] = xs_c; B.ts_out[a] = ts_c; }
}

}  // namespace br
