// BGZF on the device: DEFLATE encoding of the projected record stream (scope table row f-1, "a GPU
// inflate/deflate is a plausible follow-on": after the device path the host-side deflate is the command
// line's bottleneck).
//
//   k_deflate_dynamic one wave per BGZF block (56 KiB payload, persistent waves): greedy LZ77 with a 2048-entry
//                     hash table in LDS, 64 positions per round (one per lane); per-block Huffman codes
//                     (RFC 1951 3.2.7) from the token histograms; CRC32 of the payload by 64 lane-chunks folded
//                     with a precomputed zero-append operator.  The default.
//   k_deflate_fixed   the same parse with the fixed code (RFC 1951 3.2.6): faster, ~16 % larger output.
//   k_bgzf_compact    slots -> one dense byte stream (after a scan of the block sizes).
//
// The compressed bytes are not part of the parity contract (the reference writes through htslib/zlib);
// what is checked is that any inflater reproduces the stream (tests/test_gpu_codec.py: Python zlib / gzip).
// The projected stream repeats every read's name/SEQ/QUAL/aux once per compatible transcript a few hundred
// bytes apart, which is what the hash-table matcher picks up.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace br {

typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint64_t u64u __attribute__((aligned(1)));
typedef uint16_t u16u __attribute__((aligned(1)));

#ifndef HASH_BITS
#define HASH_BITS 11   // 4 KiB of LDS per wave: occupancy (7 workgroups per CU) matters more than the last 0.3 % of ratio
#endif
#define HASH_SIZE (1 << HASH_BITS)
#define EMPTY16 0xffffu
#define OBUF_WORDS 132  // one round emits <= 31 carried bits + 64 * (31 + 31) bits

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int nbits) { return __builtin_bitreverse32(v) >> (32 - nbits); }

// length 3..258 -> fixed-Huffman bits (code reversed, then the extra bits): returns bit count, value in v
__device__ __forceinline__ int len_bits(uint32_t len, uint32_t &v) {
  uint32_t sym, eb = 0, ev = 0;
  uint32_t m = len - 3u;
  if (len == 258u) sym = 285u;
  else if (m < 8u) sym = 257u + m;
  else { eb = (31u - (uint32_t)__builtin_clz(m)) - 2u; sym = 261u + 4u * eb + ((m >> eb) - 4u); ev = m & ((1u << eb) - 1u); }
  int nb;
  uint32_t code;
  if (sym < 280u) { code = bitrev(sym - 256u, 7); nb = 7; } else { code = bitrev(0xC0u + (sym - 280u), 8); nb = 8; }
  v = code | (ev << nb);
  return nb + (int)eb;
}
// distance 1..32768 -> 5-bit code (reversed) + extra bits
__device__ __forceinline__ int dist_bits(uint32_t dist, uint32_t &v) {
  uint32_t m = dist - 1u, dc, eb = 0, ev = 0;
  if (m < 4u) dc = m;
  else { eb = (31u - (uint32_t)__builtin_clz(m)) - 1u; dc = 2u * (eb + 1u) + ((m >> eb) & 1u); ev = m & ((1u << eb) - 1u); }
  v = bitrev(dc, 5) | (ev << 5);
  return 5 + (int)eb;
}

// ---- shared pieces ----------------------------------------------------------------------------------
// LSB-first bit stream of one BGZF payload, assembled 64 tokens at a time in an LDS buffer with atomic ORs
struct BitWriter {
  uint32_t *obuf;   // LDS, all zero between rounds
  uint8_t *pay;     // payload start in the slot
  uint32_t carry, cbits, wbase;

  // every lane contributes up to two pieces (<= 31 bits each), A before B, lane order
  __device__ __forceinline__ void round(int lane, uint32_t va, uint32_t na, uint32_t vb, uint32_t nb) {
    uint32_t nbits = na + nb, inc = nbits;
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63) + cbits;
    uint32_t pos = cbits + inc - nbits;
    if (lane == 0) obuf[0] = carry;
    __builtin_amdgcn_wave_barrier();
    if (na) {
      uint64_t lo = (uint64_t)va << (pos & 31u);
      atomicOr(&obuf[pos >> 5], (uint32_t)lo);
      if ((uint32_t)(lo >> 32)) atomicOr(&obuf[(pos >> 5) + 1], (uint32_t)(lo >> 32));
    }
    if (nb) {
      uint32_t p2 = pos + na;
      uint64_t lo = (uint64_t)vb << (p2 & 31u);
      atomicOr(&obuf[p2 >> 5], (uint32_t)lo);
      if ((uint32_t)(lo >> 32)) atomicOr(&obuf[(p2 >> 5) + 1], (uint32_t)(lo >> 32));
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t full = total >> 5;                       // <= (31 + 64 * 62) / 32 = 124
    for (uint32_t i = lane; i < full; i += 64) *(u32u *)(pay + 4u * (wbase + i)) = obuf[i];
    uint32_t nxt = obuf[full];
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i <= full + 1u && i < OBUF_WORDS; i += 64) obuf[i] = 0;
    __builtin_amdgcn_wave_barrier();
    cbits = total & 31u;
    carry = cbits ? (nxt & ((1u << cbits) - 1u)) : 0u;
    wbase += full;
  }
  // flush the remaining bits (plus `extra_zero_bits` zero bits) and return the payload byte count
  __device__ __forceinline__ uint32_t finish(int lane) {
    uint32_t nbytes = 4u * wbase + (cbits + 7u) / 8u;
    if (lane < 4) { if (4u * wbase + (uint32_t)lane < nbytes) pay[4u * wbase + lane] = (uint8_t)(carry >> (8 * lane)); }
    return nbytes;
  }
};

// Greedy parse, 128 positions per step (two half-rounds of 64, one position per lane each).  The table look-ups and the
// four-byte candidate checks of both halves are issued together -- the step is bound by dependent memory latency, and
// this halves the number of latency chains per byte.  The second half does not see the first half's table entries
// (distances below 128 are rare in a record stream and cost little).  Token kinds: 0 none, 1 literal, 2 match.
struct Token { uint32_t kind, byte, len, dist; };

// the greedy walk over one half-round: which of the 64 positions start a token, given each lane's checked candidate
__device__ __forceinline__ Token parse_half(const uint8_t *in, uint32_t n, uint32_t p, int lane, uint32_t w, uint32_t cand,
                                            uint32_t dist, bool v4, uint32_t &skip_until) {
  Token tk{0, 0, 0, 0};
  const uint32_t q = p + lane;
  const bool act = q < n;
  const bool can = q + 4u <= n;
  if (p + 64u <= skip_until) return tk;              // the whole half lies inside a match
  const uint64_t active = __ballot(act);
  const uint64_t covered = __ballot(act && q < skip_until);
  const uint64_t hasm = __ballot(v4 && q >= skip_until);
  // only the matches that are TAKEN get their length computed, by the whole wave: lane L compares bytes
  // [4 + 4L, 8 + 4L), one step covers all 258
  uint32_t mlen = 0;
  uint64_t lit = 0, mat = 0, undec = active & ~covered;
  while (undec) {
    uint64_t mm = hasm & undec;
    if (!mm) { lit |= undec; break; }
    int f = __builtin_ctzll(mm);
    uint64_t below = (1ull << f) - 1ull;
    lit |= undec & below;
    mat |= 1ull << f;
    const uint32_t qf = p + (uint32_t)f;
    const uint32_t cf = (uint32_t)__builtin_amdgcn_readlane((int)cand, f);
    uint32_t lim = n - qf; if (lim > 258u) lim = 258u;
    const uint32_t off = 4u + 4u * (uint32_t)lane;
    uint32_t x = 0;
    if (off < lim) {
      if (qf + off + 4u <= n) {
        x = *(const u32u *)(in + cf + off) ^ *(const u32u *)(in + qf + off);
        if (off + 4u > lim) x &= (1u << (8u * (lim - off))) - 1u;
      } else {
        for (uint32_t b2 = 0; b2 < 4u && off + b2 < lim; b2++) x |= (uint32_t)(in[cf + off + b2] ^ in[qf + off + b2]) << (8u * b2);
      }
    }
    uint64_t mis = __ballot(x != 0);
    uint32_t L = lim;
    if (mis) {
      int fl = __builtin_ctzll(mis);
      uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, fl);
      L = 4u + 4u * (uint32_t)fl + ((uint32_t)__builtin_ctz(xf) >> 3);
    }
    if (lane == f) mlen = L;
    uint32_t end = (uint32_t)f + L;
    uint64_t cov = end >= 64u ? (~0ull << f) : (((1ull << end) - 1ull) & ~below);
    undec &= ~(below | cov);
    if (p + end > skip_until) skip_until = p + end;
  }
  if ((lit >> lane) & 1ull) { tk.kind = 1; tk.byte = can ? (w & 0xffu) : (uint32_t)in[q]; }
  else if ((mat >> lane) & 1ull) { tk.kind = 2; tk.len = mlen; tk.dist = dist; }
  return tk;
}

__device__ __forceinline__ void parse_step(const uint8_t *in, uint32_t n, uint32_t p, int lane, uint16_t *tab, uint32_t w0, uint32_t w1,
                                           uint32_t &skip_until, Token &ta, Token &tb) {
  const uint32_t q0 = p + lane, q1 = p + 64u + lane;
  const bool can0 = q0 + 4u <= n, can1 = q1 + 4u <= n;
  const uint32_t h0 = (w0 * 2654435761u) >> (32 - HASH_BITS), h1 = (w1 * 2654435761u) >> (32 - HASH_BITS);
  ta = Token{0, 0, 0, 0}; tb = Token{0, 0, 0, 0};
  if (p + 128u <= skip_until) {                      // everything lies inside a match: only feed the table
    if (can0) tab[h0] = (uint16_t)q0;
    __builtin_amdgcn_wave_barrier();
    if (can1) tab[h1] = (uint16_t)q1;
    return;
  }
  uint32_t c0 = can0 ? tab[h0] : EMPTY16, c1 = can1 ? tab[h1] : EMPTY16;
  __builtin_amdgcn_wave_barrier();
  if (can0) tab[h0] = (uint16_t)q0;                  // any writer of a clashing slot is fine: all are < the next step's p
  __builtin_amdgcn_wave_barrier();
  if (can1) tab[h1] = (uint16_t)q1;                  // the later half wins a clash within the lane
  uint32_t d0 = q0 - c0, d1 = q1 - c1;
  bool try0 = can0 && q0 >= skip_until && c0 != EMPTY16 && d0 <= 32768u;
  bool try1 = can1 && q1 >= skip_until && c1 != EMPTY16 && d1 <= 32768u;
  uint32_t f0 = try0 ? *(const u32u *)(in + c0) : 0u, f1 = try1 ? *(const u32u *)(in + c1) : 0u;   // both in flight together
  bool v0 = try0 && f0 == w0, v1 = try1 && f1 == w1;
  ta = parse_half(in, n, p, lane, w0, c0, d0, v0, skip_until);
  if (p + 64u < n) tb = parse_half(in, n, p + 64u, lane, w1, c1, d1, v1, skip_until);
}

// CRC32 of the payload (K lane-chunks, the first takes the remainder; acc = shift(acc) ^ crc_i), then the
// BGZF header, trailer and the block size
__device__ __forceinline__ void finish_block(const DeflateArgs &A, uint64_t blk, const uint8_t *in, uint32_t n, uint8_t *out,
                                             uint32_t nbytes, int lane, const uint32_t *sh_crc) {
  const uint32_t *shift = A.crc_shift;   // [4][256], 256 look-ups per block: read from global memory, not worth 4 KiB of LDS
  const uint32_t K = (n + DEFLATE_CRC_CHUNK - 1u) / DEFLATE_CRC_CHUNK;
  uint32_t c = 0;
  if ((uint32_t)lane < K) {
    uint32_t first = n - (K - 1u) * DEFLATE_CRC_CHUNK;
    uint32_t s = lane == 0 ? 0u : first + (uint32_t)(lane - 1) * DEFLATE_CRC_CHUNK;
    uint32_t e = lane == 0 ? first : s + DEFLATE_CRC_CHUNK;
    c = 0xffffffffu;
    uint32_t i = s;
    for (; i + 4u <= e; i += 4u) {
      uint32_t d = *(const u32u *)(in + i);
      c = sh_crc[(c ^ d) & 0xffu] ^ (c >> 8);
      c = sh_crc[(c ^ (d >> 8)) & 0xffu] ^ (c >> 8);
      c = sh_crc[(c ^ (d >> 16)) & 0xffu] ^ (c >> 8);
      c = sh_crc[(c ^ (d >> 24)) & 0xffu] ^ (c >> 8);
    }
    for (; i < e; i++) c = sh_crc[(c ^ in[i]) & 0xffu] ^ (c >> 8);
    c ^= 0xffffffffu;
  }
  uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
  for (uint32_t i = 1; i < K; i++) {
    uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)c, i);
    acc = shift[acc & 0xffu] ^ shift[256 + ((acc >> 8) & 0xffu)] ^ shift[512 + ((acc >> 16) & 0xffu)] ^ shift[768 + (acc >> 24)];
    acc ^= ci;
  }
  if (lane == 0) {
    uint32_t bsize = 18u + nbytes + 8u - 1u;
    const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    for (int k = 0; k < 16; k++) out[k] = head[k];
    out[16] = (uint8_t)bsize; out[17] = (uint8_t)(bsize >> 8);
    uint8_t *t = out + 18 + nbytes;
    for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(acc >> (8 * k)); t[4 + k] = (uint8_t)(n >> (8 * k)); }
    A.sizes[blk] = bsize + 1u;
  }
}

// length 3..258 -> (symbol, extra bit count, extra value); distance 1..32768 likewise
__device__ __forceinline__ void len_symbol(uint32_t len, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
  uint32_t m = len - 3u; eb = 0; ev = 0;
  if (len == 258u) sym = 285u;
  else if (m < 8u) sym = 257u + m;
  else { eb = (31u - (uint32_t)__builtin_clz(m)) - 2u; sym = 261u + 4u * eb + ((m >> eb) - 4u); ev = m & ((1u << eb) - 1u); }
}
__device__ __forceinline__ void dist_symbol(uint32_t dist, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
  uint32_t m = dist - 1u; eb = 0; ev = 0;
  if (m < 4u) sym = m;
  else { eb = (31u - (uint32_t)__builtin_clz(m)) - 1u; sym = 2u * (eb + 1u) + ((m >> eb) & 1u); ev = m & ((1u << eb) - 1u); }
}

__global__ void __launch_bounds__(256) k_deflate_fixed(DeflateArgs A) {
  __shared__ uint16_t sh_tab[4][HASH_SIZE];
  __shared__ uint32_t sh_obuf[4][OBUF_WORDS];
  __shared__ uint32_t sh_crc[256];
  for (int i = threadIdx.x; i < 256; i += 256) sh_crc[i] = A.crc_tab[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t blk = (uint64_t)blockIdx.x * 4 + wave;
  if (blk >= A.n_blocks) return;                     // wave-uniform; no block barrier below
  uint16_t *tab = sh_tab[wave];
  const uint8_t *in = A.src + blk * DEFLATE_PAYLOAD;
  uint64_t left = A.n_bytes - blk * DEFLATE_PAYLOAD;
  const uint32_t n = left < DEFLATE_PAYLOAD ? (uint32_t)left : DEFLATE_PAYLOAD;
  uint8_t *out = A.slots + blk * DEFLATE_SLOT;
  for (int i = lane; i < HASH_SIZE; i += 64) tab[i] = EMPTY16;
  for (int i = lane; i < OBUF_WORDS; i += 64) sh_obuf[wave][i] = 0;
  __builtin_amdgcn_wave_barrier();
  BitWriter bw{sh_obuf[wave], out + 18, 3u, 3u, 0u};   // BFINAL = 1, BTYPE = 01 (fixed Huffman)
  uint32_t skip_until = 0;
  // the dwords at each lane's two positions are loaded one step ahead (their latency hides behind the current step)
  uint32_t wn0 = (lane + 4u <= n) ? *(const u32u *)(in + lane) : 0u, wn1 = (lane + 68u <= n) ? *(const u32u *)(in + lane + 64u) : 0u;
  for (uint32_t p = 0; p < n; p += 128) {
    const uint32_t w0 = wn0, w1 = wn1;
    if (p + lane + 128u + 4u <= n) wn0 = *(const u32u *)(in + p + lane + 128u);
    if (p + lane + 192u + 4u <= n) wn1 = *(const u32u *)(in + p + lane + 192u);
    Token tk[2];
    parse_step(in, n, p, lane, tab, w0, w1, skip_until, tk[0], tk[1]);
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      if (!__ballot(tk[hh].kind != 0)) continue;     // nothing starts in this half
      uint32_t va = 0, na = 0;
      if (tk[hh].kind == 1) {
        if (tk[hh].byte < 144u) { va = bitrev(0x30u + tk[hh].byte, 8); na = 8; } else { va = bitrev(0x190u + (tk[hh].byte - 144u), 9); na = 9; }
      } else if (tk[hh].kind == 2) {
        uint32_t lv, dv;
        int ln = len_bits(tk[hh].len, lv), dn = dist_bits(tk[hh].dist, dv);
        va = lv | (dv << ln); na = (uint32_t)(ln + dn);   // <= 13 + 18 = 31 bits: one piece
      }
      bw.round(lane, va, na, 0, 0);
    }
  }
  bw.round(lane, 0, lane == 0 ? 7u : 0u, 0, 0);        // end-of-block symbol 256: seven zero bits
  uint32_t nbytes = bw.finish(lane);
  finish_block(A, blk, in, n, out, nbytes, lane, sh_crc);
}

// ---- dynamic Huffman ---------------------------------------------------------------------------------------
// Same parse; the tokens of a block go to a per-wave scratch list while the literal/length and distance
// histograms build up in LDS.  Code lengths: Shannon lengths ceil(log2(total / f)) limited to 15 bits, then the
// Kraft sum is brought to exactly one -- lengthening the rarest symbols while it is above, shortening the most
// frequent ones that still fit while it is below (a complete prefix code, which inflaters require).  The code
// lengths travel uncompressed-run-length (every length as a 4-bit symbol of a flat code-length code): 167 bytes
// of header per 56 KiB block.  Then the token list is replayed through the canonical codes.
#define DYN_LL 286
#define DYN_D 30

__device__ __forceinline__ void build_lengths(uint32_t *freq, uint8_t *lens, int nsym, int lane) {
  // symbols of this lane: lane, lane + 64, ...
  uint32_t total = 0, used = 0;
  for (int i = lane; i < nsym; i += 64) { total += freq[i]; used += freq[i] ? 1u : 0u; }
  for (int o = 32; o > 0; o >>= 1) { total += __shfl_xor(total, o); used += __shfl_xor(used, o); }
  if (used == 0) { for (int i = lane; i < nsym; i += 64) lens[i] = 0; return; }
  if (used == 1) {   // one symbol: one bit (an inflater accepts the incomplete code only for distances; callers make
                     // sure the literal/length alphabet has at least two symbols: a literal or match plus end-of-block)
    for (int i = lane; i < nsym; i += 64) lens[i] = freq[i] ? 1 : 0;
    return;
  }
  int32_t k15 = 0;   // Kraft sum in units of 2^-15
  for (int i = lane; i < nsym; i += 64) {
    uint32_t f = freq[i]; uint8_t l = 0;
    if (f) {
      // ceil(log2(total / f)) without floating point: smallest l with f << l >= total
      uint32_t ll = 1; while (ll < 15u && ((uint64_t)f << ll) < total) ll++;
      l = (uint8_t)ll; k15 += 1 << (15 - ll);
    }
    lens[i] = l;
  }
  for (int o = 32; o > 0; o >>= 1) k15 += __shfl_xor(k15, o);
  __builtin_amdgcn_wave_barrier();
  // above one (only through the 15-bit limit): lengthen the rarest symbol that is not at the limit yet
  while (k15 > 32768) {
    uint32_t best = 0xffffffffu; int bi = -1;
    for (int i = lane; i < nsym; i += 64) if (lens[i] && lens[i] < 15 && freq[i] < best) { best = freq[i]; bi = i; }
    uint64_t key = ((uint64_t)best << 32) | (uint32_t)bi;
    for (int o = 32; o > 0; o >>= 1) { uint64_t t = __shfl_xor(key, o); key = t < key ? t : key; }
    int pick = (int)(uint32_t)key;
    if (pick < 0) break;                              // cannot happen: 286 symbols at 15 bits sum to < 1
    uint8_t l = lens[pick];
    k15 -= 1 << (15 - l - 1);
    if (lane == 0) lens[pick] = l + 1;
    __builtin_amdgcn_wave_barrier();
  }
  // below one: hand the slack out from the big steps to the small ones -- at length l a symbol may move to l - 1 for
  // 2^(15-l) units; floor(slack / step) of the length-l symbols (in index order: symbols of one Shannon length have
  // frequencies within a factor two of each other) take it.  A sweep always makes progress (the slack is a multiple
  // of the longest code's step), a few sweeps reach exactly one.
  int32_t slack = 32768 - k15;
  const uint64_t lt = (1ull << lane) - 1ull;
  while (slack > 0) {
    for (int l = 2; l <= 15 && slack > 0; l++) {
      const int32_t step = 1 << (15 - l);
      if (step > slack) continue;
      const int32_t can = slack / step;
      int32_t taken = 0;
      for (int base = 0; base < nsym && taken < can; base += 64) {
        int i = base + lane;
        bool has = i < nsym && lens[i] == l;
        uint64_t m = __ballot(has);
        int32_t before = (int32_t)__builtin_popcountll(m & lt);
        if (has && taken + before < can) lens[i] = (uint8_t)(l - 1);
        int32_t c = (int32_t)__builtin_popcountll(m);
        taken += c < can - taken ? c : can - taken;
      }
      slack -= taken * step;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// canonical codes (RFC 1951 3.2.2) -> enc[i] = reversed code | length << 16; nc: 16 words of LDS scratch
__device__ __forceinline__ void build_codes(const uint8_t *lens, uint32_t *enc, int nsym, int lane, uint32_t *nc) {
  // bl_count: lane l counts the symbols of length l
  uint32_t cnt = 0;
  if (lane >= 1 && lane <= 15) for (int i = 0; i < nsym; i++) cnt += lens[i] == lane;
  // next_code[l] = (next_code[l-1] + bl_count[l-1]) << 1: a serial chain over the 15 lengths
  uint32_t code = 0, prev_cnt = 0;
  for (int l = 1; l <= 15; l++) {
    code = (code + prev_cnt) << 1;
    if (lane == l) nc[l] = code;
    prev_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l);
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < nsym; i += 64) {
    uint32_t l = lens[i], e = 0;
    if (l) {
      uint32_t rank = 0;
      for (int j = 0; j < i; j++) rank += lens[j] == l;
      e = bitrev(nc[l] + rank, (int)l) | (l << 16);
    }
    enc[i] = e;
  }
  __builtin_amdgcn_wave_barrier();
}

// make EXTRA=-DDEFLATE_PROFILE: wave time per section, summed over the waves and printed by the host after every launch
#ifdef DEFLATE_PROFILE
#define TICK(k) do { uint64_t t_ = __builtin_readcyclecounter(); pt[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define TICK(k) do { } while (0)
#endif
__device__ __forceinline__ void deflate_dynamic_body(const DeflateArgs &A) {
#ifdef DEFLATE_PROFILE
  uint64_t pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t t_last = __builtin_readcyclecounter();
#endif
  __shared__ uint16_t sh_tab[4][HASH_SIZE];
  __shared__ uint32_t sh_obuf[4][OBUF_WORDS];
  __shared__ uint32_t sh_crc[256];
  __shared__ uint32_t sh_fll[4][DYN_LL + 2], sh_fd[4][DYN_D + 2];   // histograms, then the encode tables
  __shared__ uint8_t sh_lll[4][DYN_LL + 2], sh_ld[4][DYN_D + 2];
  __shared__ uint32_t sh_nc[4][16];
  for (int i = threadIdx.x; i < 256; i += 256) sh_crc[i] = A.crc_tab[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + wave;
  uint16_t *tab = sh_tab[wave];
  uint32_t *fll = sh_fll[wave], *fd = sh_fd[wave];
  uint8_t *lll = sh_lll[wave], *ld = sh_ld[wave];
  uint32_t *tokens = A.tokens + wave_id * DEFLATE_PAYLOAD;
  for (;;) {   // persistent waves (the token scratch is per wave) taking blocks off one counter: a command-line bundle is
               // only 3.4 blocks per resident wave, and a fixed stride left the chip to the slowest quarter of the grid
    uint32_t take = 0;
    if (lane == 0) take = atomicAdd(A.queue, 1u);
    const uint64_t blk = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)take);
    if (blk >= A.n_blocks) break;
    TICK(7);
    const uint8_t *in = A.src + blk * DEFLATE_PAYLOAD;
    uint64_t left = A.n_bytes - blk * DEFLATE_PAYLOAD;
    const uint32_t n = left < DEFLATE_PAYLOAD ? (uint32_t)left : DEFLATE_PAYLOAD;
    uint8_t *out = A.slots + blk * DEFLATE_SLOT;
    for (int i = lane; i < HASH_SIZE; i += 64) tab[i] = EMPTY16;
    for (int i = lane; i < OBUF_WORDS; i += 64) sh_obuf[wave][i] = 0;
    for (int i = lane; i < DYN_LL + 2; i += 64) fll[i] = 0;
    for (int i = lane; i < DYN_D + 2; i += 64) fd[i] = 0;
    __builtin_amdgcn_wave_barrier();
    TICK(0);
    // ---- parse: tokens + histograms
    uint32_t skip_until = 0, n_tok = 0;
    uint32_t wn0 = (lane + 4u <= n) ? *(const u32u *)(in + lane) : 0u, wn1 = (lane + 68u <= n) ? *(const u32u *)(in + lane + 64u) : 0u;
    for (uint32_t p = 0; p < n; p += 128) {
      const uint32_t w0 = wn0, w1 = wn1;
      if (p + lane + 128u + 4u <= n) wn0 = *(const u32u *)(in + p + lane + 128u);
      if (p + lane + 192u + 4u <= n) wn1 = *(const u32u *)(in + p + lane + 192u);
      Token tk2[2];
      parse_step(in, n, p, lane, tab, w0, w1, skip_until, tk2[0], tk2[1]);
      TICK(1);
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const Token tk = tk2[hh];
        uint64_t sel = __ballot(tk.kind != 0);
        if (!sel) continue;
        uint32_t rank = (uint32_t)__builtin_popcountll(sel & ((1ull << lane) - 1ull));
        if (tk.kind == 1) { tokens[n_tok + rank] = tk.byte; atomicAdd(&fll[tk.byte], 1u); }
        else if (tk.kind == 2) {
          tokens[n_tok + rank] = 0x80000000u | ((tk.len - 3u) << 16) | (tk.dist - 1u);
          uint32_t s1, e1, v1, s2, e2, v2;
          len_symbol(tk.len, s1, e1, v1); dist_symbol(tk.dist, s2, e2, v2);
          atomicAdd(&fll[s1], 1u); atomicAdd(&fd[s2], 1u);
        }
        n_tok += (uint32_t)__builtin_popcountll(sel);
      }
      TICK(2);
    }
    if (lane == 0) fll[256] = 1;                      // end of block
    __builtin_amdgcn_wave_barrier();
    // ---- codes
    build_lengths(fll, lll, DYN_LL, lane);
    build_lengths(fd, ld, DYN_D, lane);
    __builtin_amdgcn_wave_barrier();
    build_codes(lll, fll, DYN_LL, lane, sh_nc[wave]);  // the histograms become the encode tables
    build_codes(ld, fd, DYN_D, lane, sh_nc[wave]);
    __builtin_amdgcn_wave_barrier();
    TICK(3);
    // ---- header: BFINAL 1, BTYPE 10, HLIT 29 (286), HDIST 29 (30), HCLEN 15 (19); code-length code: symbols 0..15 get
    // four bits each (flat, complete), 16/17/18 unused; order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    BitWriter bw{sh_obuf[wave], out + 18, 0u, 0u, 0u};
    {
      uint32_t va = 0, na = 0;
      if (lane == 0) { va = 1u | (2u << 1) | (29u << 3) | (29u << 8) | (15u << 13); na = 17; }
      else if (lane <= 19) { va = (lane <= 3) ? 0u : 4u; na = 3; }   // lanes 1..3: symbols 16, 17, 18
      bw.round(lane, va, na, 0, 0);
    }
    for (int base = 0; base < DYN_LL + DYN_D; base += 64) {          // 316 code lengths, 4 bits each (code = reversed value)
      int i = base + lane;
      uint32_t va = 0, na = 0;
      if (i < DYN_LL + DYN_D) { uint32_t l = i < DYN_LL ? lll[i] : ld[i - DYN_LL]; va = bitrev(l, 4); na = 4; }
      bw.round(lane, va, na, 0, 0);
    }
    TICK(4);
    // ---- tokens through the codes
    for (uint32_t t0 = 0; t0 < n_tok; t0 += 64) {
      uint32_t t = t0 + lane;
      uint32_t va = 0, na = 0, vb = 0, nb = 0;
      if (t < n_tok) {
        uint32_t tk = tokens[t];
        if (!(tk & 0x80000000u)) { uint32_t e = fll[tk]; va = e & 0xffffu; na = e >> 16; }
        else {
          uint32_t len = ((tk >> 16) & 0xffu) + 3u, dist = (tk & 0x7fffu) + 1u;
          uint32_t s1, e1, v1, s2, e2, v2;
          len_symbol(len, s1, e1, v1); dist_symbol(dist, s2, e2, v2);
          uint32_t c1 = fll[s1], c2 = fd[s2];
          va = (c1 & 0xffffu) | (v1 << (c1 >> 16)); na = (c1 >> 16) + e1;      // <= 15 + 5
          vb = (c2 & 0xffffu) | (v2 << (c2 >> 16)); nb = (c2 >> 16) + e2;      // <= 15 + 13
        }
      }
      bw.round(lane, va, na, vb, nb);
    }
    { uint32_t e = fll[256]; bw.round(lane, lane == 0 ? (e & 0xffffu) : 0u, lane == 0 ? (e >> 16) : 0u, 0, 0); }
    uint32_t nbytes = bw.finish(lane);
    TICK(5);
    finish_block(A, blk, in, n, out, nbytes, lane, sh_crc);
    __builtin_amdgcn_wave_barrier();
    TICK(6);
  }
#ifdef DEFLATE_PROFILE
  if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd((unsigned long long *)(A.tokens + (uint64_t)gridDim.x * 4 * DEFLATE_PAYLOAD) + k, (unsigned long long)pt[k]);
#endif
}

// Six waves per SIMD (80 VGPRs, ten of them spilled) against the four the compiler settles on by itself (105 VGPRs): the parse
// is a chain of dependent scalar and vector instructions (2.3 SALU + 1.8 VALU per input byte, profiles/r03/pmc_deflate.txt),
// and the fifth wave is what fills the issue slots: 19.9 -> 17.0 ms per 2.4 GB (five waves: 17.1; LDS allows six workgroups).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) k_deflate_dynamic(DeflateArgs A) { deflate_dynamic_body(A); }

__global__ void __launch_bounds__(256) k_bgzf_compact(DeflateArgs A, const uint64_t *off, uint8_t *dense) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t blk = (uint64_t)blockIdx.x * 4 + wave;
  if (blk >= A.n_blocks) return;
  const uint8_t *s = A.slots + blk * DEFLATE_SLOT;
  uint8_t *d = dense + off[blk];
  uint32_t n = A.sizes[blk];
  uint32_t n16 = n & ~15u;
  struct __attribute__((packed, aligned(1))) W4 { uint32_t a, b, c, d; };
  for (uint32_t i = 16u * lane; i < n16; i += 16u * 64u) *(W4 *)(d + i) = *(const W4 *)(s + i);
  for (uint32_t i = n16 + lane; i < n; i += 64) d[i] = s[i];
}

void launch_deflate(hipStream_t st, const DeflateArgs &A, int dynamic_waves) {
  if (!A.n_blocks) return;
  if (dynamic_waves > 0) hipLaunchKernelGGL(k_deflate_dynamic, dim3((unsigned)(dynamic_waves / 4)), dim3(256), 0, st, A);
  else hipLaunchKernelGGL(k_deflate_fixed, dim3((unsigned)((A.n_blocks + 3) / 4)), dim3(256), 0, st, A);
}
void launch_bgzf_compact(hipStream_t st, const DeflateArgs &A, const uint64_t *off, uint8_t *dense) {
  if (A.n_blocks) hipLaunchKernelGGL(k_bgzf_compact, dim3((unsigned)((A.n_blocks + 3) / 4)), dim3(256), 0, st, A, off, dense);
}

}  // namespace br
