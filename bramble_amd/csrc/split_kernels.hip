// Record boundaries of an inflated BAM alignment section in HBM: br_bam_split on the device (scope table row f-1; the
// reference walks the same chain one bam_read1 at a time, gclib/GSam.cpp over htslib).
//
// A BAM stream is a chain -- every record starts with its own length -- so the boundaries behind a byte depend on all the
// records in front of it.  The stream is cut into 32 KiB segments, and every segment first GUESSES where its first record
// starts: the first offset whose fixed fields are those of a record (lengths that add up, reference ids inside the header's
// range, a NUL where the read name ends) and whose next two hops pass the same test.  Every segment then walks its own
// records from its guess, and a verification pass compares every guess with where the chain of the segments in front
// actually arrives; a segment that guessed wrong (or in which no record starts: one long read can cover many segments)
// takes the chain's value and walks again.  The result is exact whatever the guesses were -- they only decide how many
// passes it takes (one, in practice).
//
//   k_split_guess   one lane per segment: the first plausible record start
//   k_split_walk    one lane per segment: count mapped / unmapped records, where the chain leaves the segment
//   k_split_check   one lane per segment: the entry the chain gives it; mismatches are counted and repaired
//   k_split_emit    one lane per segment: rec_off / rec_len of the mapped records (after a scan of the counts)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace br {

typedef uint32_t su32u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t sl32(const uint8_t *p) { return *(const su32u *)p; }

// the fixed fields at p look like a record's (p = the block_size field); needs p + 36 <= n
__device__ __forceinline__ bool plausible(const SplitArgs &S, uint64_t p) {
  const uint8_t *d = S.data + p;
  const uint32_t bs = sl32(d);
  if (bs < 32u || bs > (1u << 28)) return false;
  const int32_t ref = (int32_t)sl32(d + 4), pos = (int32_t)sl32(d + 8);
  if (ref < -1 || ref >= S.n_ref || pos < -1) return false;
  const uint32_t l_rn = d[12], n_cig = sl32(d + 16) & 0xffffu;
  const int32_t l_seq = (int32_t)sl32(d + 20), nref = (int32_t)sl32(d + 24), npos = (int32_t)sl32(d + 28);
  if (l_rn == 0u || l_seq < 0 || nref < -1 || nref >= S.n_ref || npos < -1) return false;
  const uint64_t ls = (uint64_t)l_seq;
  if (32ull + l_rn + 4ull * n_cig + (ls + 1) / 2 + ls > bs) return false;
  const uint64_t nul = p + 36 + l_rn - 1;
  if (nul < S.n_bytes && S.data[nul] != 0) return false;
  return true;
}

__global__ void __launch_bounds__(64) k_split_guess(SplitArgs S) {
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= S.n_seg) return;
  if (s == 0) { S.entry[0] = 0; return; }                    // the caller's data starts at a record
  const uint64_t lo = (uint64_t)s * S.seg_bytes, hi = lo + S.seg_bytes < S.n_bytes ? lo + S.seg_bytes : S.n_bytes;
  uint64_t found = ~0ull;
  for (uint64_t p = lo; p < hi && p + 36 <= S.n_bytes; p++) {
    if (!plausible(S, p)) continue;
    const uint64_t p1 = p + 4 + sl32(S.data + p);
    if (p1 > S.n_bytes) continue;   // a record that does not fit: only the chain's very last record may look like that, and the check finds that one
    bool ok = true;
    if (p1 + 36 <= S.n_bytes) {
      ok = plausible(S, p1);
      if (ok) { const uint64_t p2 = p1 + 4 + sl32(S.data + p1); if (p2 + 36 <= S.n_bytes) ok = plausible(S, p2); }
    }
    if (ok) { found = p; break; }
  }
  S.entry[s] = found;
}

// walks the records that start in segment s from entry[s]; EMIT: writes the mapped records' offsets and lengths
template <bool EMIT>
__device__ __forceinline__ void walk_segment(const SplitArgs &S, int64_t s) {
  const uint64_t lo = (uint64_t)s * S.seg_bytes, hi = lo + S.seg_bytes < S.n_bytes ? lo + S.seg_bytes : S.n_bytes;
  uint64_t p = S.entry[s];
  uint32_t nm = 0, nu = 0, ended = 0;
  uint64_t out = EMIT ? S.map_pre[s] : 0;
  if (p != ~0ull) {
    while (p < hi) {
      if (p + 4 > S.n_bytes) { ended = 1; break; }            // the length field itself is cut off
      const uint32_t bs = sl32(S.data + p);
      if (bs < 32u) { ended = 2; break; }                     // malformed (br_bam_split: BR_ERR_INVALID_ARG) -- if this walk is on the chain
      if (p + 4 + (uint64_t)bs > S.n_bytes) { ended = 1; break; } // a partial record: the next call's
      const uint8_t *r = S.data + p + 4;
      const uint32_t l_qname = r[8], w = sl32(r + 12), n_cig = w & 0xffffu, flag = w >> 16;
      const int32_t l_seq = (int32_t)sl32(r + 16);
      const uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
      if (32ull + l_qname + 4ull * n_cig + (ls + 1) / 2 + ls > bs || l_qname == 0u) { ended = 2; break; }
      if (flag & 0x4u) nu++;
      else { if (EMIT) { S.rec_off[out] = p + 4; S.rec_len[out] = bs; out++; } nm++; }
      p += 4 + (uint64_t)bs;
    }
  }
  if (!EMIT) { S.exit_[s] = p; S.n_map[s] = nm; S.n_unm[s] = nu; S.ended[s] = ended; }
}

// all segments (redo == null), or the segments whose entry the check just changed
__global__ void __launch_bounds__(64) k_split_walk(SplitArgs S, const uint32_t *redo) {
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= S.n_seg) return;
  if (redo && !redo[s]) return;
  walk_segment<false>(S, s);
}

__global__ void __launch_bounds__(64) k_split_emit(SplitArgs S) {
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= S.n_seg) return;
  walk_segment<true>(S, s);
}

// the entry the chain gives segment s: where the walk of the nearest segment in front that has records leaves it, when that
// lies inside s; nothing when the chain jumps over s or has ended.  A differing entry is replaced and the segment marked.
__global__ void __launch_bounds__(64) k_split_check(SplitArgs S, uint32_t *redo) {
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s >= S.n_seg) return;
  uint32_t changed = 0;
  if (s > 0) {
    const uint64_t lo = (uint64_t)s * S.seg_bytes, hi = lo + S.seg_bytes < S.n_bytes ? lo + S.seg_bytes : S.n_bytes;
    int64_t t = s - 1;
    while (t > 0 && S.entry[t] == ~0ull) t--;                 // (segment 0 always has an entry)
    // (the chain arrives in a segment between t and s that has no entry yet -- one that is being repaired: s keeps what it has
    // until that one has walked)
    uint64_t want = ~0ull;
    if (!S.ended[t]) { const uint64_t e = S.exit_[t]; if (e >= lo && e < hi) want = e; else if (e < lo) want = S.entry[s]; }
    S.entry_next[s] = want;
    if (S.entry[s] != want) { changed = 1; atomicAdd(S.flags + 1, 1u); }
  } else S.entry_next[0] = 0;
  redo[s] = changed;
}

// totals: the chain's end (where the next call continues), unmapped records, whether the chain met a malformed record
__global__ void __launch_bounds__(256) k_split_totals(SplitArgs S) {
  __shared__ unsigned long long sh_unm[4];
  unsigned long long unm = 0;
  for (int64_t s = threadIdx.x; s < S.n_seg; s += 256) { unm += S.n_unm[s]; if (S.entry[s] != ~0ull && S.ended[s] == 2u) atomicOr(S.flags, 1u); }
  for (int o = 32; o; o >>= 1) unm += __shfl_xor(unm, o);
  if ((threadIdx.x & 63) == 0) sh_unm[threadIdx.x >> 6] = unm;
  __syncthreads();
  if (threadIdx.x == 0) {
    S.totals[0] = sh_unm[0] + sh_unm[1] + sh_unm[2] + sh_unm[3];
    int64_t t = S.n_seg - 1;
    while (t > 0 && S.entry[t] == ~0ull) t--;
    S.totals[1] = S.exit_[t];                                 // first byte that belongs to no complete record
  }
}

// start of the last read-name group among records [0, n): the largest i whose name differs from record i - 1's (0 when all
// share one name).  out[0] must be zero before the launch.
__global__ void __launch_bounds__(256) k_last_group(const uint8_t *data, const uint64_t *rec_off, int64_t n, unsigned long long *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long best = 0;
  if (i > 0 && i < n) {
    const uint8_t *a = data + rec_off[i - 1], *b = data + rec_off[i];
    const uint32_t la = a[8], lb = b[8];
    bool same = la == lb;
    for (uint32_t k = 0; same && k < la; k++) same = a[32 + k] == b[32 + k];
    if (!same) best = (unsigned long long)i;
  }
  for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = t > best ? t : best; }
  if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

// unmapped records that start before byte `limit` (after k_split_*: whole segments from their counts, the one that holds the
// limit by walking it again)
__global__ void __launch_bounds__(256) k_unmapped_before(SplitArgs S, uint64_t limit, unsigned long long *out) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  if (s < S.n_seg) {
    const uint64_t lo = (uint64_t)s * S.seg_bytes, hi = lo + S.seg_bytes;
    if (hi <= limit) n = S.n_unm[s];
    else if (lo < limit && S.entry[s] != ~0ull) {
      uint64_t p = S.entry[s];
      while (p < limit && p + 4 <= S.n_bytes) {
        const uint32_t bs = sl32(S.data + p);
        if (bs < 32u || p + 4 + (uint64_t)bs > S.n_bytes) break;
        if ((sl32(S.data + p + 4 + 12) >> 16) & 0x4u) n++;
        p += 4 + (uint64_t)bs;
      }
    }
  }
  for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

// ---------------------------------------------------------------------------
// Piece-wise reading (br_bam_piece_*): a piece = a run of BGZF blocks [b0, b1) handled on its own, by whatever device.  It owns
// the records from its START to its END, both defined by the same rule on either side of a block boundary B:
//     q = the first mapped record that starts at or after B;  the cut = the first record after q whose read name differs
//     from its predecessor's (the read-name group that holds q goes to the piece in front).
// The piece in front walks the true record chain across B (it inflates a few blocks past its end); the piece behind does not
// know where a record starts in its first block and GUESSES q's predecessor-free start (k_first_record: the first offset
// whose fixed fields and next hops look like records) -- the caller compares the two cuts and repeats the piece with the
// true start when they differ, so the result never depends on the guess.
// ---------------------------------------------------------------------------
// first plausible record start in data[0, limit): one wave, 64 offsets per step
__global__ void __launch_bounds__(64) k_first_record(SplitArgs S, uint64_t limit, unsigned long long *out) {
  const int lane = threadIdx.x;
  if (limit > S.n_bytes) limit = S.n_bytes;
  for (uint64_t p0 = 0; p0 < limit; p0 += 64) {
    const uint64_t p = p0 + lane;
    bool ok = false;
    if (p < limit && p + 36 <= S.n_bytes && plausible(S, p)) {
      uint64_t q = p; ok = true;
      for (int hop = 0; hop < 4 && ok; hop++) {   // four more records in a row (or the end of the data)
        q += 4 + (uint64_t)sl32(S.data + q);
        if (q > S.n_bytes) { ok = false; break; }
        if (q + 36 > S.n_bytes) break;
        ok = plausible(S, q);
      }
    }
    const unsigned long long m = __ballot(ok);
    if (m) { if (lane == 0) out[0] = p0 + (unsigned long long)__builtin_ctzll(m); return; }
  }
  if (lane == 0) out[0] = ~0ull;
}
// cut[0] (preset n, or 0 for a known start) = first i >= 1 whose name differs from record i - 1's; cut[1] (preset ~0) = the
// first such i whose predecessor starts at or after `bound`
__global__ void __launch_bounds__(256) k_piece_bounds(const uint8_t *data, const uint64_t *rec_off, int64_t n, uint64_t bound, int guess,
                                                      unsigned long long *cut) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < 1 || i >= n) return;
  const uint64_t oa = rec_off[i - 1];
  const uint8_t *a = data + oa, *b = data + rec_off[i];
  const uint32_t la = a[8], lb = b[8];
  bool same = la == lb;
  for (uint32_t k = 0; same && k < la; k++) same = a[32 + k] == b[32 + k];
  if (same) return;
  if (guess) atomicMin(&cut[0], (unsigned long long)i);
  if (oa - 4 >= bound) atomicMin(&cut[1], (unsigned long long)i);
}
// cut[2] / cut[3] = byte offsets of records cut[0] / cut[1] (their block_size fields; `used` = the end of the last complete
// record when the index is n or beyond); cut[0] is left as it is for a known start (offset 0)
__global__ void k_piece_offsets(const uint64_t *rec_off, int64_t n, uint64_t used, int guess, unsigned long long *cut) {
  const unsigned long long iS = cut[0], iE = cut[1] < (unsigned long long)n ? cut[1] : (unsigned long long)n;
  cut[2] = guess ? (iS < (unsigned long long)n ? rec_off[iS] - 4 : used) : 0ull;
  cut[3] = iE < (unsigned long long)n ? rec_off[iE] - 4 : used;
  cut[4] = 0;
}
// cut[4] += unmapped records that start in [cut[2], cut[3])
__global__ void __launch_bounds__(256) k_unmapped_in(SplitArgs S, unsigned long long *cut) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t from = cut[2], limit = cut[3];
  unsigned long long n = 0;
  if (s < S.n_seg && S.entry[s] != ~0ull) {
    const uint64_t lo = (uint64_t)s * S.seg_bytes, hi = lo + S.seg_bytes;
    if (lo >= from && hi <= limit) n = S.n_unm[s];
    else if (hi > from && lo < limit) {
      uint64_t p = S.entry[s];
      while (p < limit && p < hi && p + 4 <= S.n_bytes) {
        const uint32_t bs = sl32(S.data + p);
        if (bs < 32u || p + 4 + (uint64_t)bs > S.n_bytes) break;
        if (p >= from && ((sl32(S.data + p + 4 + 12) >> 16) & 0x4u)) n++;
        p += 4 + (uint64_t)bs;
      }
    }
  }
  for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&cut[4], n);
}
void launch_first_record(hipStream_t st, const SplitArgs &S, uint64_t limit, unsigned long long *out) { hipLaunchKernelGGL(k_first_record, dim3(1), dim3(64), 0, st, S, limit, out); }
void launch_piece_cut(hipStream_t st, const SplitArgs &S, const uint64_t *rec_off, int64_t n, uint64_t bound, uint64_t used, int guess, unsigned long long *cut) {
  if (n > 1) hipLaunchKernelGGL(k_piece_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S.data, rec_off, n, bound, guess, cut);
  hipLaunchKernelGGL(k_piece_offsets, dim3(1), dim3(1), 0, st, rec_off, n, used, guess, cut);
  if (S.n_seg > 0) hipLaunchKernelGGL(k_unmapped_in, dim3((unsigned)((S.n_seg + 255) / 256)), dim3(256), 0, st, S, cut);
}

void launch_last_group(hipStream_t st, const uint8_t *data, const uint64_t *rec_off, int64_t n, unsigned long long *out) {
  if (n > 1) hipLaunchKernelGGL(k_last_group, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, data, rec_off, n, out);
}
void launch_unmapped_before(hipStream_t st, const SplitArgs &S, uint64_t limit, unsigned long long *out) {
  hipLaunchKernelGGL(k_unmapped_before, dim3((unsigned)((S.n_seg + 255) / 256)), dim3(256), 0, st, S, limit, out);
}

// test hook (br_ctx_set_param "split_spoil" = k): wrong guesses on purpose -- every k-th segment forgets its entry, the segments in
// between every 2k-th take a byte offset that starts no record, runs of them included -- so that the check / repair passes
// are exercised on data whose honest guesses are all right
__global__ void __launch_bounds__(64) k_split_spoil(SplitArgs S, int k) {
  const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (s < 1 || s >= S.n_seg) return;
  if (s % k == 0) S.entry[s] = ~0ull;
  else if (s % (2 * k) < 4 && S.entry[s] != ~0ull && S.entry[s] + 40 < S.n_bytes) S.entry[s] += 1 + (uint64_t)(s % 3);
}
void launch_split_spoil(hipStream_t st, const SplitArgs &S, int k) { if (k > 0) hipLaunchKernelGGL(k_split_spoil, dim3((unsigned)((S.n_seg + 63) / 64)), dim3(64), 0, st, S, k); }

void launch_split_guess(hipStream_t st, const SplitArgs &S) { hipLaunchKernelGGL(k_split_guess, dim3((unsigned)((S.n_seg + 63) / 64)), dim3(64), 0, st, S); }
void launch_split_walk(hipStream_t st, const SplitArgs &S, const uint32_t *redo) { hipLaunchKernelGGL(k_split_walk, dim3((unsigned)((S.n_seg + 63) / 64)), dim3(64), 0, st, S, redo); }
void launch_split_check(hipStream_t st, const SplitArgs &S, uint32_t *redo) { hipLaunchKernelGGL(k_split_check, dim3((unsigned)((S.n_seg + 63) / 64)), dim3(64), 0, st, S, redo); }
void launch_split_emit(hipStream_t st, const SplitArgs &S) { hipLaunchKernelGGL(k_split_emit, dim3((unsigned)((S.n_seg + 63) / 64)), dim3(64), 0, st, S); }
void launch_split_totals(hipStream_t st, const SplitArgs &S) { hipLaunchKernelGGL(k_split_totals, dim3(1), dim3(256), 0, st, S); }

}  // namespace br
