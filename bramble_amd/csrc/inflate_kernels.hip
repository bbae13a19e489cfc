// BGZF inflate on the device (scope table row f-1, the reader side of "a GPU inflate/deflate is a plausible follow-on"):
// what the reference gets from htslib's bgzf_read behind GSamReader (gclib/GSam.h; include/bramble.h:29-85).
//
// k_inflate: one wave per BGZF block (persistent waves taking blocks off one counter).  A BGZF block is a complete DEFLATE
// stream of at most 64 KiB (RFC 1951: stored, fixed and dynamic blocks, several per stream).  Symbol decoding is a serial
// chain: every lane of the wave runs the same chain on wave-uniform values (broadcast LDS reads) -- except behind a literal,
// where the lanes decode the codes at every bit offset of the buffer and the run of literals is taken in one go -- and the
// parallelism is across blocks: a 3 GB file is 50 000 of them.  The lanes work together where there is width: the compressed bytes come
// in through a 1 KiB LDS ring (one coalesced 512-byte load ahead of the decoder), the decode tables (10-bit primary table for
// literal/length codes, 8-bit for distances, canonical count arrays for the longer codes) are built by all lanes, a match
// is copied by as many lanes as it has bytes, the output leaves through a 2 KiB LDS window in coalesced 512-byte pieces
// (matches that reach further back read the wave's own earlier output from HBM), and the CRC32 of the result is computed
// over 64 lane-chunks and folded with a precomputed zero-append operator (as in codec_kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace br {

#define IN_RING_DW 256u          // 1 KiB of compressed input per wave
#define IN_CHUNK 512u            // refilled half a KiB at a time
#define OUT_WIN 2048u            // output window in LDS
#define OUT_PIECE 512u           // flushed to HBM in pieces of this size
#define LL_BITS 10
#define D_BITS 8

struct __attribute__((packed, aligned(1))) IW4 { uint32_t a, b, c, d; };

__constant__ uint8_t INFLATE_ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // RFC 1951 3.2.7

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct __attribute__((aligned(16))) WaveLds {
  uint32_t in[IN_RING_DW + 4];             // (+ the first four dwords once more)
  uint32_t win[OUT_WIN / 4];
  uint16_t lut_ll[1u << LL_BITS];          // symbol | length << 9 (0: the code is longer than LL_BITS, or unused)
  uint16_t lut_d[1u << D_BITS];            // symbol | length << 5 (also the 7-bit table of the code-length alphabet)
  uint16_t sorted_ll[288], sorted_d[32];   // symbols in canonical order (by length, then by value)
  uint16_t cnt_ll[16], cnt_d[16];
  uint16_t fcode[16], findex[16];          // table construction: first code / first index of a length
  uint32_t long_lim[6];                    // literal/length codes longer than LL_BITS: the first code past length LL_BITS + k, left-aligned to 15 bits
  int32_t long_base[6];                    // ... and what turns a code of length LL_BITS + k into its index in sorted_ll (k = 1 .. 5)
  uint8_t lens[352];                       // code lengths: [0, 320) literal/length + distance, [320, 339) the code-length alphabet
};

// canonical Huffman tables from code lengths lens[0 .. n): counts per length, symbols in canonical order, the primary table.
// false: over-subscribed.  (Incomplete codes are legal for distances; an unused table entry decodes as an error.)
__device__ __forceinline__ bool build_tables(WaveLds &W, const uint8_t *lens, int n, uint16_t *cnt, uint16_t *sorted, uint16_t *lut, int bits,
                                             int sym_shift, int lane) {
  uint32_t c = 0;                                   // lane L counts the symbols of length L
  if (lane >= 1 && lane <= 15) for (int i = 0; i < n; i++) c += lens[i] == lane;
  uint32_t code = 0, index = 0, over = 0, my_fc = 0, my_fi = 0;
  for (int l = 1; l <= 15; l++) {
    const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, l);
    code <<= 1;
    if (lane == l) { my_fc = code; my_fi = index; }
    code += cl; index += cl;
    if (code > (1u << l)) over = 1;
  }
  if (lane < 16) { cnt[lane] = (uint16_t)c; W.fcode[lane] = (uint16_t)my_fc; W.findex[lane] = (uint16_t)my_fi; }
  for (int i = lane; i < (1 << bits); i += 64) lut[i] = 0;
  __builtin_amdgcn_wave_barrier();
  if (over) return false;
  for (int i = lane; i < n; i += 64) {
    const uint32_t l = lens[i];
    if (!l) continue;
    uint32_t rank = 0;
    for (int j = 0; j < i; j++) rank += lens[j] == l;
    const uint32_t cd = (uint32_t)W.fcode[l] + rank;
    sorted[(uint32_t)W.findex[l] + rank] = (uint16_t)i;
    if (l <= (uint32_t)bits) {
      const uint32_t r = __builtin_bitreverse32(cd) >> (32u - l);
      for (uint32_t k = r; k < (1u << bits); k += 1u << l) lut[k] = (uint16_t)((uint32_t)i | (l << sym_shift));
    }
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_inflate(InflateArgs A) {
  __shared__ WaveLds sh_w[4];
  __shared__ uint32_t sh_crc[4][256];
  for (int i = threadIdx.x; i < 1024; i += 256) sh_crc[i >> 8][i & 255] = A.crc_tab4[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveLds &W = sh_w[wave];
  uint8_t *const win8 = (uint8_t *)W.win;

  for (;;) {
    uint32_t take = 0;
    if (lane == 0) take = atomicAdd(A.queue, 1u);
    const uint64_t blk = (uint64_t)uni(take);
    if (blk >= A.n_blocks) break;
    const InflateBlock B = A.blocks[blk];
    const uint8_t *src = A.src + B.src_off;
    const uint64_t src_room = A.n_src - B.src_off;           // bytes that may be read from src on
    uint8_t *out = A.dst + B.dst_off;
    const uint32_t ulen = B.ulen;
    const uint32_t bit_end = (B.clen + 64u) * 8u;             // a decoder that is still going this far behind the payload is lost
    bool bad = false;

    // ---- input ring: chunk k of the block's bytes (512 of them) sits in ring half k & 1.  While the decoder is in chunk k,
    // chunks k and k + 1 are in the ring and chunk k + 2 is on its way (in `ahead`); the ring's first four dwords are kept
    // once more behind its end, so that three consecutive dwords can be read from any index without wrapping.
    // (Eight bytes at a time: a piece that holds payload bytes ends inside the block's eight-byte trailer at the latest, so
    // nothing is read past the buffer and nothing needs a byte loop; what lies behind the buffer reads as zero.)
    auto load_chunk = [&](uint32_t k) {
      const uint64_t o = (uint64_t)k * IN_CHUNK + 16u * (uint32_t)lane;
      struct __attribute__((packed, aligned(1))) IW2 { uint32_t a, b; };
      uint4 v = make_uint4(0, 0, 0, 0);
      if (16u * (uint32_t)lane >= IN_CHUNK) return v;          // (a chunk is IN_CHUNK / 16 lanes wide)
      if (o + 8 <= src_room) { const IW2 t = *(const IW2 *)(src + o); v.x = t.a; v.y = t.b; }
      if (o + 16 <= src_room) { const IW2 t = *(const IW2 *)(src + o + 8); v.z = t.a; v.w = t.b; }
      return v;
    };
    auto put_chunk = [&](uint32_t k, uint4 v) {
      if (16u * (uint32_t)lane < IN_CHUNK) *(uint4 *)(W.in + (k & 1u) * (IN_CHUNK / 4) + 4u * (uint32_t)lane) = v;
      if (lane == 0 && !(k & 1u)) *(uint4 *)(W.in + IN_RING_DW) = v;
    };
    uint32_t bitpos = 0, chunk = 0;                            // the next bit of the stream; the chunk it lies in
    uint4 ahead;
    auto prime = [&]() {                                       // the ring around bitpos, from the source
      chunk = bitpos / (8u * IN_CHUNK);
      __builtin_amdgcn_wave_barrier();
      put_chunk(chunk, load_chunk(chunk)); put_chunk(chunk + 1u, load_chunk(chunk + 1u)); ahead = load_chunk(chunk + 2u);
      __builtin_amdgcn_wave_barrier();
    };
    prime();
    auto skip = [&](uint32_t n) {                              // n <= 8 * IN_CHUNK bits
      bitpos += n;
      if (bitpos / (8u * IN_CHUNK) != chunk) {                 // entering chunk k + 1: chunk k's half takes chunk k + 2, k + 3 is asked for
        chunk++;
        __builtin_amdgcn_wave_barrier();
        put_chunk(chunk + 1u, ahead);
        ahead = load_chunk(chunk + 2u);
        __builtin_amdgcn_wave_barrier();
      }
    };
    auto peek = [&]() -> uint32_t {                            // the next 32 bits, wave-uniform
      const uint32_t k = (bitpos >> 5) & (IN_RING_DW - 1u);
      const uint32_t d0 = uni(W.in[k]), d1 = uni(W.in[k + 1u]);
      return (uint32_t)(((((uint64_t)d1) << 32) | (uint64_t)d0) >> (bitpos & 31u));
    };
    auto bits = [&](uint32_t n) { const uint32_t v = peek() & ((1u << n) - 1u); skip(n); return v; };   // n <= 16
    // a code longer than the primary table (or an unused entry): bit by bit against the canonical counts
    auto slow = [&](const uint16_t *cnt, const uint16_t *sorted) -> int {
      uint32_t w = peek();
      uint32_t code = 0, first = 0, index = 0;
      for (uint32_t l = 1; l <= 15u; l++) {
        code |= w & 1u; w >>= 1;
        const uint32_t count = uni(cnt[l]);
        if (code < first + count) { skip(l); return (int)uni(sorted[index + (code - first)]); }
        index += count; first += count; first <<= 1; code <<= 1;
      }
      return -1;
    };

    uint32_t pos = 0, flushed = 0;                            // output bytes produced; of them in HBM
    auto flush_to = [&](uint32_t upto) {                      // whole pieces of the window [flushed, upto) go home; upto a multiple of OUT_PIECE or the end
      while (flushed < upto) {
        const uint32_t n = upto - flushed < OUT_PIECE ? upto - flushed : OUT_PIECE;
        const uint32_t o = 16u * (uint32_t)lane;
        __builtin_amdgcn_wave_barrier();
        if (o + 16u <= n) {
          const uint4 v = *(const uint4 *)(W.win + (((flushed + o) & (OUT_WIN - 1u)) >> 2));
          IW4 t; t.a = v.x; t.b = v.y; t.c = v.z; t.d = v.w;
          *(IW4 *)(out + flushed + o) = t;
        } else if (o < n) {
          for (uint32_t j = o; j < n; j++) out[flushed + j] = win8[(flushed + j) & (OUT_WIN - 1u)];
        }
        flushed += n;
      }
    };
    // a match: len bytes at `at` from dist back, by as many lanes as it has bytes (1 <= dist <= at; hi: the end of what this
    // step has written to the window so far -- literals behind the match included)
    auto copy_bytes = [&](uint32_t at, uint32_t len, uint32_t dist, uint32_t hi) {
      __builtin_amdgcn_wave_barrier();
#ifdef INFLATE_FAKE_FAR
      if (true) {   // (profiles/inflate_far_probe.py: what would the kernel cost if no match ever left the window?)
#else
      if (dist + (hi - at) <= OUT_WIN) {                        // the source is still in the window
#endif
        const float rcp = 1.0f / (float)dist;
        for (uint32_t i = (uint32_t)lane; i < len; i += 64u) {
          uint32_t so = i;
          if (dist < len) {                                     // the match overlaps itself: byte i repeats byte i mod dist
            const uint32_t q = (uint32_t)((float)i * rcp);      // (i < 258: the quotient is exact or one short)
            so = i - q * dist;
            if (so >= dist) so -= dist;
          }
          win8[(at + i) & (OUT_WIN - 1u)] = win8[(at - dist + so) & (OUT_WIN - 1u)];
        }
      } else {
        // far back: the bytes left the window, they are in HBM (written by this wave, flushed long ago: a step writes at most
        // 1024 + 258 bytes; dist > len here: no overlap with [at, ..))
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t i = (uint32_t)lane; i < len; i += 64u) win8[(at + i) & (OUT_WIN - 1u)] = out[at - dist + i];
      }
      __builtin_amdgcn_wave_barrier();
    };
    auto copy_match = [&](uint32_t len, uint32_t dist) {        // dist <= pos, pos + len <= ulen
      copy_bytes(pos, len, dist, pos + len);
      pos += len;
      if ((pos & ~(OUT_PIECE - 1u)) > flushed) flush_to(pos & ~(OUT_PIECE - 1u));
    };

    // ---- the DEFLATE blocks of the stream
    bool last = false;
    while (!last && !bad) {
      if (bitpos > bit_end) { bad = true; break; }
      last = bits(1) != 0u;
      const uint32_t btype = bits(2);
      if (btype == 0u) {                                      // stored: to the byte boundary, LEN, NLEN, bytes
        skip((8u - (bitpos & 7u)) & 7u);
        const uint32_t len = bits(16);
        const uint32_t nlen = bits(16);
        if ((len ^ nlen) != 0xffffu || pos + len > ulen) { bad = true; break; }
        const uint32_t sp = bitpos >> 3;                        // the bytes come straight from the source
        for (uint32_t base = 0; base < len; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < len) {
            const uint64_t so = (uint64_t)sp + i;
            win8[(pos + i) & (OUT_WIN - 1u)] = so < src_room ? src[so] : (uint8_t)0;
          }
          // finished pieces leave at once (the window holds OUT_WIN bytes)
          const uint32_t reach = pos + (base + 64u < len ? base + 64u : len);
          if ((reach & ~(OUT_PIECE - 1u)) > flushed) flush_to(reach & ~(OUT_PIECE - 1u));
        }
        pos += len;
        flush_to(pos & ~(OUT_PIECE - 1u));
        if (len) { bitpos += 8u * len; prime(); }               // the ring continues behind the copied bytes
        continue;
      }
      if (btype == 3u) { bad = true; break; }
      int n_ll = 288, n_d = 30;
      if (btype == 1u) {                                      // fixed code (RFC 1951 3.2.6)
        for (int i = lane; i < 288; i += 64) W.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        if (lane < 32) W.lens[288 + lane] = 5;
        __builtin_amdgcn_wave_barrier();
        n_ll = 288; n_d = 30;
      } else {                                                // dynamic code (3.2.7)
        const uint32_t hlit = bits(5) + 257u, hdist = bits(5) + 1u, hclen = bits(4) + 4u;
        if (hlit > 286u || hdist > 30u) { bad = true; break; }
        if (lane < 19) W.lens[320 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = 0; i < hclen; i++) {
          const uint32_t v = bits(3);
          if (lane == 0) W.lens[320 + INFLATE_ORD[i]] = (uint8_t)v;
        }
        __builtin_amdgcn_wave_barrier();
        if (!build_tables(W, W.lens + 320, 19, W.cnt_d, W.sorted_d, W.lut_d, 7, 5, lane)) { bad = true; break; }
        const uint32_t total = hlit + hdist;
        uint32_t i = 0;
        while (i < total && !bad) {
          int sym;
          const uint32_t w = peek();
          const uint32_t e = uni(W.lut_d[w & 127u]);
          uint32_t used = e >> 5;
          if (used) sym = (int)(e & 31u); else sym = slow(W.cnt_d, W.sorted_d);   // (slow() skips its bits itself)
          if (sym < 0 || sym > 18) { bad = true; break; }
          const uint32_t x = w >> used;                          // the extra bits of 16 / 17 / 18 (used + 7 <= 14 bits of w)
          if (sym < 16) { if (used) skip(used); if (lane == 0) W.lens[i] = (uint8_t)sym; i++; __builtin_amdgcn_wave_barrier(); continue; }
          if (!used) { bad = true; break; }                      // (a code-length code is at most seven bits long: the table holds all of them)
          uint32_t rep, val = 0;
          if (sym == 16) { if (i == 0) { bad = true; break; } __builtin_amdgcn_wave_barrier(); val = uni(W.lens[i - 1]); rep = 3u + (x & 3u); used += 2u; }
          else if (sym == 17) { rep = 3u + (x & 7u); used += 3u; }
          else { rep = 11u + (x & 127u); used += 7u; }
          skip(used);
          if (i + rep > total) { bad = true; break; }
          for (uint32_t k = (uint32_t)lane; k < rep; k += 64u) W.lens[i + k] = (uint8_t)val;
          i += rep;
          __builtin_amdgcn_wave_barrier();
        }
        if (bad) break;
        if (uni(W.lens[256]) == 0u) { bad = true; break; }   // no end-of-block code
        n_ll = (int)hlit; n_d = (int)hdist;
        // the distance lengths follow the literal/length ones: move them to a place of their own
        uint8_t dl = 0;
        if (lane < n_d) dl = W.lens[hlit + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) W.lens[288 + lane] = lane < n_d ? dl : 0;
        __builtin_amdgcn_wave_barrier();
      }
      if (!build_tables(W, W.lens, n_ll, W.cnt_ll, W.sorted_ll, W.lut_ll, LL_BITS, 9, lane)) { bad = true; break; }
      if (lane < 6) {                                           // (fcode / findex are the next build's scratch)
        const uint32_t l = (uint32_t)(LL_BITS + lane);
        W.long_lim[lane] = ((uint32_t)W.fcode[l] + (uint32_t)W.cnt_ll[l]) << (15u - l);
        W.long_base[lane] = (int32_t)W.findex[l] - (int32_t)W.fcode[l];
      }
      __builtin_amdgcn_wave_barrier();
      if (!build_tables(W, W.lens + 288, n_d, W.cnt_d, W.sorted_d, W.lut_d, D_BITS, 5, lane)) { bad = true; break; }

      // ---- symbols.  Lane i decodes the whole token that would start at bit offset i behind bitpos (two LDS gathers: the
      // literal/length table, the distance table); the chain of real token starts is then followed from offset 0 through the
      // lanes' "next" values by the scalar unit (a v_readlane and four scalar instructions per literal), up to the first
      // token that is not a literal; the literals on the chain are stored by their lanes, the token behind them (a match,
      // the end of the block) is taken from its lane's registers.  Only codes longer than the tables go bit by bit.
      for (;;) {
        if (bitpos > bit_end) { bad = true; break; }
        const uint32_t p = bitpos + (uint32_t)lane;
        const uint32_t k = (p >> 5) & (IN_RING_DW - 1u);
        const uint32_t d0 = W.in[k], d1 = W.in[k + 1u], d2 = W.in[k + 2u];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, p & 31u), hi = __builtin_amdgcn_alignbit(d2, d1, p & 31u);
        uint32_t el = W.lut_ll[lo & ((1u << LL_BITS) - 1u)];
        if (el < 512u) {
          // no code of at most LL_BITS bits starts here: a longer one, by its place among the canonical codes (the codes of
          // length l are the 15-bit left-aligned values in [lim[l - 1], lim[l]))
          const uint32_t rev = __builtin_bitreverse32(lo) >> 17;
          const uint32_t k = (uint32_t)(rev >= W.long_lim[1]) + (uint32_t)(rev >= W.long_lim[2]) + (uint32_t)(rev >= W.long_lim[3]) + (uint32_t)(rev >= W.long_lim[4]) + 1u;
          if (rev >= W.long_lim[0] && rev < W.long_lim[5]) el = (uint32_t)W.sorted_ll[(uint32_t)(W.long_base[k] + (int32_t)(rev >> (5u - k)))] | ((LL_BITS + k) << 9);
        }
        const uint32_t l1 = el >> 9, sy = el & 511u;
        const bool coded = el >= 512u;                           // a literal/length code starts here
        // next: where the token behind this lane's literal starts; 128 + lane: no literal here
        // the token as a match, whatever it is (no branch: a lane at a wrong offset decodes noise and nobody asks for it)
        const uint32_t ls = sy - 257u;                            // 0 .. 28 for a length code
        const bool len_short = ls < 8u, len_top = ls == 28u;
        const uint32_t eb = len_short || len_top ? 0u : ((ls - 4u) >> 2) & 7u;
        const uint32_t base = len_short ? ls + 3u : len_top ? 258u : ((4u + ((ls - 4u) & 3u)) << eb) + 3u;
        const uint64_t v = ((((uint64_t)hi) << 32) | (uint64_t)lo) >> l1;
        const uint32_t mlen = base + ((uint32_t)v & ((1u << eb) - 1u));
        const uint32_t x = (uint32_t)(v >> eb);                  // 32 bits from the distance code on (l1 + eb <= 15 of the 64 are gone)
        const uint32_t ed = W.lut_d[x & ((1u << D_BITS) - 1u)];
        const uint32_t mdl = ed >> 5, mds = ed & 31u;
        const uint32_t deb = mds < 4u ? 0u : (mds >> 1) - 1u;
        const uint32_t dbase = mds < 4u ? mds + 1u : ((2u + (mds & 1u)) << deb) + 1u;
        const uint32_t mdist = dbase + ((x >> mdl) & ((1u << deb) - 1u));
        // len | (dist - 1) << 9 | bits << 24 | kind << 30 (1: match, 2: end of block; 0: not decoded here)
        const bool matched = coded && ls < 29u && ed >= 32u && mds < 30u;
        const uint32_t mbits = l1 + eb + mdl + deb;
        uint32_t info = matched ? (1u << 30) | (mbits << 24) | ((mdist - 1u) << 9) | mlen : 0u;
        if (coded && sy == 256u) info = (2u << 30) | (l1 << 24);
        // next: where the token behind this lane's starts, for a literal and for a short match (at most 32 bytes: a step then
        // writes at most 1024 of them); 128 + lane: a token that ends the step (a long match, the end of the block, a distance
        // code longer than its table)
        const bool through = matched && mlen <= 32u;
        const uint32_t nxt = coded && sy < 256u ? (uint32_t)lane + l1 : through ? (uint32_t)lane + mbits : 128u + (uint32_t)lane;
        // the chain of token starts from offset 0: the scalar unit follows every fourth token (pointers doubled twice by the
        // lanes), the tokens between are marked by their predecessors (a push to the lane of the offset; nobody pushes to
        // lane 0 -- an offset behind a token is at least 1 -- so the lanes with nothing to say push there)
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nxt & 63u) << 2), (int)nxt);
        const uint32_t j1 = nxt < 64u ? t1 : nxt;                // two tokens on (or where the first one left the buffer / was no literal)
        const uint32_t t2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((j1 & 63u) << 2), (int)j1);
        const uint32_t j2 = j1 < 64u ? t2 : j1;                  // four tokens on
        uint32_t at = 0, n;
        uint64_t sel = 0;
        do {
          n = (uint32_t)__builtin_amdgcn_readlane((int)j2, (int)at);
          sel |= 1ull << at;
          at = n;
        } while (n < 64u);
        {
          const bool in = (sel >> lane) & 1ull;
          const bool push = in && j1 < 64u;
          const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(push ? j1 << 2 : 0u), push ? 1 : 0);
          sel |= __builtin_amdgcn_ballot_w64(got != 0u);
        }
        {
          const bool in = (sel >> lane) & 1ull;
          const bool push = in && nxt < 64u;
          const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(push ? nxt << 2 : 0u), push ? 1 : 0);
          sel |= __builtin_amdgcn_ballot_w64(got != 0u);
        }
        uint32_t term = 64u;                                     // the lane of the token that ended the run, if it starts inside the buffer
        if (at >= 128u) { term = at - 128u; sel &= ~(1ull << term); at = term; }
        const uint64_t msel = sel & __builtin_amdgcn_ballot_w64(through);   // the short matches on the chain
        const uint64_t lsel = sel & ~msel;                       // its literals
        const uint32_t n_lit = (uint32_t)__builtin_popcountll(lsel);
        if (msel == 0ull) {
          if (n_lit > ulen - pos) { bad = true; break; }         // more bytes than the block holds
          if (n_lit) {
            if ((lsel >> lane) & 1ull) win8[(pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(lsel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lsel, 0u))) & (OUT_WIN - 1u)] = (uint8_t)sy;
            pos += n_lit;
            if ((pos & ~(OUT_PIECE - 1u)) > flushed) flush_to(pos & ~(OUT_PIECE - 1u));
          }
        } else {
          // literals and matches: every literal lands behind the bytes of the matches in front of it (all literals first: they
          // need nothing), then the matches in stream order (a match may repeat what an earlier token of the step wrote)
          uint32_t before = 0, tot_m = 0;                        // bytes of the matches in front of this lane's token; of all
          for (uint64_t mm = msel; mm; mm &= mm - 1ull) {
            const uint32_t m = (uint32_t)__builtin_ctzll(mm);
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)m) & 511u;
            before += (uint32_t)lane > m ? L : 0u;
            tot_m += L;
          }
          const uint32_t total = n_lit + tot_m;
          if (total > ulen - pos) { bad = true; break; }
          if ((lsel >> lane) & 1ull) win8[(pos + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(lsel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lsel, 0u))) & (OUT_WIN - 1u)] = (uint8_t)sy;
          uint32_t run = 0;
          for (uint64_t mm = msel; mm; mm &= mm - 1ull) {
            const uint32_t m = (uint32_t)__builtin_ctzll(mm);
            const uint32_t tm = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)m);
            const uint32_t L = tm & 511u, D = ((tm >> 9) & 0x7fffu) + 1u;
            const uint32_t mpos = pos + (uint32_t)__builtin_popcountll(lsel & ((1ull << m) - 1ull)) + run;
            if (D > mpos) { bad = true; break; }
            copy_bytes(mpos, L, D, pos + total);
            run += L;
          }
          if (bad) break;
          pos += total;
          if ((pos & ~(OUT_PIECE - 1u)) > flushed) flush_to(pos & ~(OUT_PIECE - 1u));
        }
        if (term == 64u) { skip(at); continue; }                 // the run goes on behind the buffer
        const uint32_t ti = (uint32_t)__builtin_amdgcn_readlane((int)info, (int)term);
        const uint32_t kind = ti >> 30;
        if (kind == 1u) {
          const uint32_t len = ti & 511u, dist = ((ti >> 9) & 0x7fffu) + 1u;
          skip(at + ((ti >> 24) & 63u));
          if (dist > pos || pos + len > ulen) { bad = true; break; }
          copy_match(len, dist);
          continue;
        }
        if (kind == 2u) { skip(at + ((ti >> 24) & 63u)); break; }
        // not decoded by the lane (a code longer than a table): this one token bit by bit
        skip(at);
        int sym;
        {
          const uint32_t e = uni(W.lut_ll[peek() & ((1u << LL_BITS) - 1u)]);
          if (e >> 9) { sym = (int)(e & 511u); skip(e >> 9); } else sym = slow(W.cnt_ll, W.sorted_ll);
        }
        if (sym < 0 || sym > 285) { bad = true; break; }
        if (sym < 256) {
          if (pos >= ulen) { bad = true; break; }
          if (lane == 0) win8[pos & (OUT_WIN - 1u)] = (uint8_t)sym;
          pos++;
          if ((pos & (OUT_PIECE - 1u)) == 0u) flush_to(pos);
          continue;
        }
        if (sym == 256) break;
        uint32_t len;
        const uint32_t s = (uint32_t)sym;
        if (s < 265u) len = s - 254u;
        else if (s == 285u) len = 258u;
        else { const uint32_t eb = (s - 261u) >> 2; len = ((4u + ((s - 261u) & 3u)) << eb) + 3u + bits(eb); }
        int ds;
        {
          const uint32_t ed = uni(W.lut_d[peek() & ((1u << D_BITS) - 1u)]);
          if (ed >> 5) { ds = (int)(ed & 31u); skip(ed >> 5); } else ds = slow(W.cnt_d, W.sorted_d);
        }
        if (ds < 0 || ds > 29) { bad = true; break; }
        uint32_t dist;
        if (ds < 4) dist = (uint32_t)ds + 1u;
        else { const uint32_t eb = ((uint32_t)ds >> 1) - 1u; dist = ((2u + ((uint32_t)ds & 1u)) << eb) + 1u + bits(eb); }
        if (dist > pos || pos + len > ulen) { bad = true; break; }
        copy_match(len, dist);
      }
    }
    if (!bad && pos != ulen) bad = true;
    if (!bad) flush_to(pos);
    // ---- CRC32 of the block's bytes (read back from HBM: K lane-chunks, the first takes the remainder)
    if (!bad) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const uint32_t K = (ulen + INFLATE_CRC_CHUNK - 1u) / INFLATE_CRC_CHUNK;
      uint32_t c = 0;
      if ((uint32_t)lane < K) {
        const uint32_t first = ulen - (K - 1u) * INFLATE_CRC_CHUNK;
        const uint32_t s0 = lane == 0 ? 0u : first + (uint32_t)(lane - 1) * INFLATE_CRC_CHUNK;
        const uint32_t e0 = lane == 0 ? first : s0 + INFLATE_CRC_CHUNK;
        c = 0xffffffffu;
        uint32_t i = s0;
        typedef uint32_t u32u __attribute__((aligned(1)));
        for (; i + 4u <= e0; i += 4u) {
          const uint32_t d = *(const u32u *)(out + i) ^ c;
          c = sh_crc[3][d & 0xffu] ^ sh_crc[2][(d >> 8) & 0xffu] ^ sh_crc[1][(d >> 16) & 0xffu] ^ sh_crc[0][d >> 24];
        }
        for (; i < e0; i++) c = sh_crc[0][(c ^ out[i]) & 0xffu] ^ (c >> 8);
        c ^= 0xffffffffu;
      }
      const uint32_t *shift = A.crc_shift;
      uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
      for (uint32_t i = 1; i < K; i++) {
        const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)i);
        acc = shift[acc & 0xffu] ^ shift[256 + ((acc >> 8) & 0xffu)] ^ shift[512 + ((acc >> 16) & 0xffu)] ^ shift[768 + (acc >> 24)];
        acc ^= ci;
      }
      if (ulen == 0u) acc = 0u;
      if (acc != B.crc) bad = true;
    }
    if (bad && lane == 0) atomicAdd(A.n_bad, 1u);
    __builtin_amdgcn_wave_barrier();
  }
}

void launch_inflate(hipStream_t st, const InflateArgs &A, int n_waves) {
  if (!A.n_blocks || n_waves < 4) return;
  hipLaunchKernelGGL(k_inflate, dim3((unsigned)(n_waves / 4)), dim3(256), 0, st, A);
}

}  // namespace br
