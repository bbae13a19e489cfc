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
#define OUT_NEAR (OUT_WIN - OUT_PIECE - 258u)   // a match at most this far back is served from the window
#define LL_BITS 10
#define D_BITS 8

struct __attribute__((packed, aligned(1))) IW4 { uint32_t a, b, c, d; };

__constant__ uint8_t INFLATE_ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // RFC 1951 3.2.7

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct __attribute__((aligned(16))) WaveLds {
  uint32_t in[IN_RING_DW];
  uint32_t win[OUT_WIN / 4];
  uint16_t lut_ll[1u << LL_BITS];          // symbol | length << 9 (0: the code is longer than LL_BITS, or unused)
  uint16_t lut_d[1u << D_BITS];            // symbol | length << 5 (also the 7-bit table of the code-length alphabet)
  uint16_t sorted_ll[288], sorted_d[32];   // symbols in canonical order (by length, then by value)
  uint16_t cnt_ll[16], cnt_d[16];
  uint16_t fcode[16], findex[16];          // table construction: first code / first index of a length
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
    bool bad = false;

    // ---- input ring: chunk k of the block's bytes sits in ring half k & 1; chunk k + 1 is there too, chunk k + 2 on its way
    // (eight bytes at a time: a piece that holds payload bytes ends inside the block's eight-byte trailer at the latest, so
    // nothing is read past the buffer and nothing needs a byte loop)
    auto load_chunk = [&](uint32_t k) {
      const uint64_t o = (uint64_t)k * IN_CHUNK + 16u * (uint32_t)lane;
      struct __attribute__((packed, aligned(1))) IW2 { uint32_t a, b; };
      uint4 v = make_uint4(0, 0, 0, 0);
      if (16u * (uint32_t)lane >= IN_CHUNK) return v;          // (a chunk is IN_CHUNK / 16 lanes wide)
      if (o + 8 <= src_room) { const IW2 t = *(const IW2 *)(src + o); v.x = t.a; v.y = t.b; }
      if (o + 16 <= src_room) { const IW2 t = *(const IW2 *)(src + o + 8); v.z = t.a; v.w = t.b; }
      return v;
    };
    auto put_chunk = [&](uint32_t k, uint4 v) { if (16u * (uint32_t)lane < IN_CHUNK) *(uint4 *)(W.in + (k & 1u) * (IN_CHUNK / 4) + 4u * (uint32_t)lane) = v; };
    put_chunk(0, load_chunk(0));
    put_chunk(1, load_chunk(1));
    uint4 ahead = load_chunk(2);
    __builtin_amdgcn_wave_barrier();

    // bit buffer: the next bc (<= 128) bits of the stream in bh:bb, ip = bytes taken from the ring.  After refill() more than
    // 96 are there: a whole token (<= 48 bits) at each of the first 48 bit offsets, for the lanes that decode ahead
    uint64_t bb = 0, bh = 0; uint32_t bc = 0, ip = 0;
    auto refill = [&]() {
      while (bc <= 96u) {
        const uint64_t w = (uint64_t)uni(W.in[(ip >> 2) & (IN_RING_DW - 1u)]);
        if (bc < 64u) { bb |= w << bc; if (bc > 32u) bh |= w >> (64u - bc); }
        else bh |= w << (bc - 64u);
        bc += 32u; ip += 4u;
        if ((ip & (IN_CHUNK - 1u)) == 0u) {                   // entering chunk k: chunk k - 1's half takes chunk k + 1, k + 2 is asked for
          const uint32_t k = ip / IN_CHUNK;
          __builtin_amdgcn_wave_barrier();
          put_chunk(k + 1u, ahead);
          ahead = load_chunk(k + 2u);
          __builtin_amdgcn_wave_barrier();
        }
      }
    };
    auto consume = [&](uint32_t n) {                          // n <= bc
      if (n >= 64u) { bb = bh >> (n - 64u); bh = 0; }
      else if (n) { bb = (bb >> n) | (bh << (64u - n)); bh >>= n; }
      bc -= n;
    };
    auto bits = [&](uint32_t n) { const uint32_t v = (uint32_t)bb & ((1u << n) - 1u); consume(n); return v; };
    // a code longer than the primary table (or an unused entry): bit by bit against the canonical counts
    auto slow = [&](const uint16_t *cnt, const uint16_t *sorted) -> int {
      uint32_t code = 0, first = 0, index = 0;
      for (int l = 1; l <= 15; l++) {
        code |= (uint32_t)bb & 1u; consume(1u);
        const uint32_t count = uni(cnt[l]);
        if (code < first + count) return (int)uni(sorted[index + (code - first)]);
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

    // ---- the DEFLATE blocks of the stream
    bool last = false;
    while (!last && !bad) {
      refill();
      last = bits(1) != 0u;
      const uint32_t btype = bits(2);
      if (btype == 0u) {                                      // stored: to the byte boundary, LEN, NLEN, bytes
        bits(bc & 7u);
        refill();
        const uint32_t len = bits(16); refill();
        const uint32_t nlen = bits(16);
        if ((len ^ nlen) != 0xffffu || pos + len > ulen) { bad = true; break; }
        // bytes still in the bit buffer first, then straight from the source
        uint32_t done = 0;
        while (done < len && bc >= 8u) { if (lane == 0) win8[(pos + done) & (OUT_WIN - 1u)] = (uint8_t)bb; consume(8u); done++; }
        // (bc is 0 here unless len ran out: the ring position ip is the next source byte)
        for (uint32_t base = done; base < len; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < len) {
            const uint64_t so = (uint64_t)ip + (i - done);
            win8[(pos + i) & (OUT_WIN - 1u)] = so < src_room ? src[so] : (uint8_t)0;
          }
          // finished pieces leave at once (the window holds OUT_WIN bytes)
          const uint32_t reach = pos + (base + 64u < len ? base + 64u : len);
          if ((reach & ~(OUT_PIECE - 1u)) > flushed) flush_to(reach & ~(OUT_PIECE - 1u));
        }
        if (done < len) {
          // the ring continues behind the copied bytes: reload it from there
          const uint32_t np = ip + (len - done);
          ip = np & ~3u;
          const uint32_t k = ip / IN_CHUNK;
          __builtin_amdgcn_wave_barrier();
          put_chunk(k, load_chunk(k)); put_chunk(k + 1u, load_chunk(k + 1u)); ahead = load_chunk(k + 2u);
          __builtin_amdgcn_wave_barrier();
          bb = 0; bh = 0; bc = 0;
          refill();
          bits(8u * (np & 3u));
        }
        pos += len;
        flush_to(pos & ~(OUT_PIECE - 1u));
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
        refill();
        const uint32_t hlit = bits(5) + 257u, hdist = bits(5) + 1u, hclen = bits(4) + 4u;
        if (hlit > 286u || hdist > 30u) { bad = true; break; }
        if (lane < 19) W.lens[320 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = 0; i < hclen; i++) {
          refill();
          const uint32_t v = bits(3);
          if (lane == 0) W.lens[320 + INFLATE_ORD[i]] = (uint8_t)v;
        }
        __builtin_amdgcn_wave_barrier();
        if (!build_tables(W, W.lens + 320, 19, W.cnt_d, W.sorted_d, W.lut_d, 7, 5, lane)) { bad = true; break; }
        const uint32_t total = hlit + hdist;
        uint32_t i = 0;
        while (i < total && !bad) {
          refill();
          int sym;
          const uint32_t e = uni(W.lut_d[(uint32_t)bb & 127u]);
          if (e >> 5) { sym = (int)(e & 31u); consume(e >> 5); } else sym = slow(W.cnt_d, W.sorted_d);
          if (sym < 0 || sym > 18) { bad = true; break; }
          if (sym < 16) { if (lane == 0) W.lens[i] = (uint8_t)sym; i++; __builtin_amdgcn_wave_barrier(); continue; }
          uint32_t rep, val = 0;
          if (sym == 16) { if (i == 0) { bad = true; break; } __builtin_amdgcn_wave_barrier(); val = uni(W.lens[i - 1]); rep = 3u + bits(2); }
          else if (sym == 17) rep = 3u + bits(3);
          else rep = 11u + bits(7);
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
      if (!build_tables(W, W.lens + 288, n_d, W.cnt_d, W.sorted_d, W.lut_d, D_BITS, 5, lane)) { bad = true; break; }

      // ---- symbols
      bool in_run = false;                                      // the last token was a literal: the next ones probably are
      for (;;) {
        refill();
        if (in_run) {
          // A run of literals at once: lane i decodes the code that would start at bit offset i of the buffer (one LDS gather),
          // the chain of real starts is followed from offset 0 through the lanes (a v_readlane and an add per literal where
          // the serial path needs a broadcast LDS read and ~25 scalar instructions), and the lanes on the chain store their bytes.
          uint64_t v = bb >> (uint32_t)lane;
          if (lane) v |= bh << (64u - (uint32_t)lane);
          const uint32_t el = W.lut_ll[(uint32_t)v & ((1u << LL_BITS) - 1u)];
          const uint32_t l1 = el >> 9, sy = el & 511u;
          // where the token behind this lane's literal starts; 255: no (short-coded) literal here, or its code would leave the buffer
          const uint32_t lim = bc - 16u < 64u ? bc - 16u : 64u;
          const uint32_t nxt = (l1 != 0u && sy < 256u && (uint32_t)lane < lim) ? (uint32_t)lane + l1 : 255u;
          uint64_t sel = 0; uint32_t at = 0, n_lit = 0;
          for (;;) {                                            // (kept this plain: every test in here is scalar instructions per literal)
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)at);
            if (n >= 128u) break;
            sel |= 1ull << at; n_lit++;
            at = n;
            if (at >= 64u) break;
          }
          if (n_lit > ulen - pos) { bad = true; break; }        // more bytes than the block holds
          if (n_lit) {
            if ((sel >> lane) & 1ull) win8[(pos + (uint32_t)__builtin_popcountll(sel & ((1ull << lane) - 1ull))) & (OUT_WIN - 1u)] = (uint8_t)sy;
            pos += n_lit;
            consume(at);
            if ((pos & ~(OUT_PIECE - 1u)) > flushed) flush_to(pos & ~(OUT_PIECE - 1u));
            continue;
          }
          in_run = false;
        }
        int sym;
        const uint32_t e = uni(W.lut_ll[(uint32_t)bb & ((1u << LL_BITS) - 1u)]);
        if (e >> 9) { sym = (int)(e & 511u); consume(e >> 9); } else sym = slow(W.cnt_ll, W.sorted_ll);
        if (sym < 0 || sym > 285) { bad = true; break; }
        if (sym < 256) {
          if (pos >= ulen) { bad = true; break; }
          if (lane == 0) win8[pos & (OUT_WIN - 1u)] = (uint8_t)sym;
          pos++;
          if ((pos & (OUT_PIECE - 1u)) == 0u) flush_to(pos);   // every finished piece leaves at once: the window never holds more than a piece + a match of unsent bytes
          in_run = true;
          continue;
        }
        if (sym == 256) break;
        // length
        uint32_t len;
        const uint32_t s = (uint32_t)sym;
        if (s < 265u) len = s - 254u;
        else if (s == 285u) len = 258u;
        else { const uint32_t eb = (s - 261u) >> 2; len = ((4u + ((s - 261u) & 3u)) << eb) + 3u + bits(eb); }
        refill();
        int ds;
        const uint32_t ed = uni(W.lut_d[(uint32_t)bb & ((1u << D_BITS) - 1u)]);
        if (ed >> 5) { ds = (int)(ed & 31u); consume(ed >> 5); } else ds = slow(W.cnt_d, W.sorted_d);
        if (ds < 0 || ds > 29) { bad = true; break; }
        uint32_t dist;
        if (ds < 4) dist = (uint32_t)ds + 1u;
        else { const uint32_t eb = ((uint32_t)ds >> 1) - 1u; dist = ((2u + ((uint32_t)ds & 1u)) << eb) + 1u + bits(eb); }
        if (dist > pos || pos + len > ulen) { bad = true; break; }
        __builtin_amdgcn_wave_barrier();
        if (dist <= OUT_NEAR) {
          for (uint32_t i = (uint32_t)lane; i < len; i += 64u) {
            const uint32_t so = dist >= len ? i : i % dist;
            win8[(pos + i) & (OUT_WIN - 1u)] = win8[(pos - dist + so) & (OUT_WIN - 1u)];
          }
        } else {
          // far back: the bytes left the window, they are in HBM (written by this wave; at most 258 bytes, no overlap with [pos, ..))
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          for (uint32_t i = (uint32_t)lane; i < len; i += 64u) win8[(pos + i) & (OUT_WIN - 1u)] = out[pos - dist + i];
        }
        __builtin_amdgcn_wave_barrier();
        pos += len;
        if ((pos & ~(OUT_PIECE - 1u)) > flushed) flush_to(pos & ~(OUT_PIECE - 1u));
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
