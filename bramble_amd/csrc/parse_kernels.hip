// BAM records -> the projection kernels' input tables, on the device (scope table row f-1:
// "Host BAM/BGZF decode -> flat batches").  The host only inflates BGZF blocks and walks the
// block_size chain; every field the path needs is pulled out of the raw records here.
//
// Restates, per record, what the reference's reader side does before convert_reads:
//   process_reads / process_read_in   src/bramble.cpp:313-441  (start = pos + 1, refid, strand inputs)
//   GSamRecord accessors              gclib/GSam.h:310-344     (name, isPaired, mate_start = mpos<0 ? 0 : mpos+1)
//   tag_char1("XS") / ("ts")          gclib/GSam.cpp:310-318   (first value byte of an A or Z tag) -- in k_bam_scan
//   process_pairs                     src/bramble.cpp:272-311  (mate index by name + mate position) -> k_mates*
//   shared sequence of a name group   src/core.cpp:353-378     (first record of the group with a sequence)
//
//   k_rec_fields  one lane per record: fixed-field extraction, "starts a new read name" flag
//   k_group_off   scatter group starts after the scan of those flags
//   k_rec_copy    G lanes per record: CIGAR words and read name into the aligned arenas
//   k_mates       one lane per read-name group (sequential hash-map semantics, groups <= MATES_BIG)
//   k_mates_big   one wave per large group (the open-set search runs across the lanes)
//   k_seq_src     one lane per group: which record's sequence the group shares (-S only)
//   k_seq_ascii   G lanes per source record: 4-bit SEQ -> ASCII (-S only)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bam_cg.h"
#include "kernels.h"
#include "records_inl.h"

namespace br {

typedef uint32_t u32u __attribute__((aligned(1)));
typedef uint16_t u16u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) P4 { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t rec_length(const ParseArgs &P, int64_t i) {
  return P.rec_len ? P.rec_len[i] : (uint32_t)(P.rec_off[i + 1] - P.rec_off[i]);
}

// names compared four bytes at a time (unaligned dword loads), tail bytewise
__device__ __forceinline__ bool same_bytes(const uint8_t *a, const uint8_t *b, uint32_t n) {
  uint32_t k = 0;
  for (; k + 4 <= n; k += 4) if (*(const u32u *)(a + k) != *(const u32u *)(b + k)) return false;
  for (; k < n; k++) if (a[k] != b[k]) return false;
  return true;
}

__global__ void __launch_bounds__(256) k_rec_fields(ParseArgs P) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t ncig = 0, max_s = 0;
  if (i < P.n) {
    const GlobalRec rec{P.blob + P.rec_off[i]};
    const GlobalRec prev{P.blob + (i > 0 ? P.rec_off[i - 1] : 0)};
    rec_fields_one(P, i, rec, rec_length(P, i), prev, i > 0 ? rec_length(P, i - 1) : 0u, ncig, max_s);   // records_inl.h
  }
  // batch maxima: one atomic per wave
  for (int o = 32; o > 0; o >>= 1) {
    ncig = max(ncig, (uint32_t)__shfl_xor((int)ncig, o));
    max_s = max(max_s, (uint32_t)__shfl_xor((int)max_s, o));
  }
  // the maxima only grow: a (possibly stale) plain read filters out nearly every same-address atomic
  if ((threadIdx.x & 63) == 0) {
    if (ncig > __builtin_nontemporal_load(P.maxima)) atomicMax(P.maxima, ncig);
    if (max_s > __builtin_nontemporal_load(P.maxima + 1)) atomicMax(P.maxima + 1, max_s);
  }
}

// after the exclusive scan of isnew: group g starts at record i when isnew[i]
__global__ void __launch_bounds__(256) k_group_off(ParseArgs P) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > P.n) return;
  if (i == P.n) { P.group_off[P.group_pre[P.n]] = (uint32_t)P.n; return; }
  if (P.isnew[i]) P.group_off[P.group_pre[i]] = (uint32_t)i;
}

template <int G>
__global__ void __launch_bounds__(256) k_rec_copy(ParseArgs P) {
  const int lane = threadIdx.x & (G - 1);
  int64_t i = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  if (i >= P.n) return;
  const uint8_t *rec = P.blob + P.rec_off[i];
  uint32_t nc = P.cigar_off[i + 1] - P.cigar_off[i];
  uint32_t nl = P.name_off[i + 1] - P.name_off[i];
  if (nc == 0 && nl == 0) return;
  uint32_t l_qname = rec[8];
  const uint8_t *cg = rec + 32 + l_qname;
  {   // the real CIGAR of a record whose CIGAR field holds the CG placeholder (k_rec_fields counted the tag's ops)
    const uint32_t nf = ld_u16(rec + 12);
    const int32_t l_seq = (int32_t)ld_u32(rec + 16);
    const uint64_t rlen = rec_length(P, i);
    if (nc != nf && cg_candidate(rec, rlen, l_qname, nf, l_seq)) { CgTag t; if (cg_find(rec, rlen, l_qname, nf, l_seq, t)) cg = rec + t.tag_at + 8; }
  }
  uint32_t *dst = P.cigar + P.cigar_off[i];
  for (uint32_t k = lane; k < nc; k += G) dst[k] = *(const u32u *)(cg + 4 * k);
  uint8_t *nd = P.names + P.name_off[i];
  const uint8_t *nm = rec + 32;
  for (uint32_t k = lane; k < nl; k += G) nd[k] = nm[k];
}

// process_pairs (src/bramble.cpp:272-311) inside one read-name group [a, b): the map key is
// name + '-' + position and the name is constant here, so the key is the position.  A record
// first looks up its mate's position; when found the two are paired and the entry erased;
// otherwise it stores its own position (overwriting an entry with the same key).
__device__ __forceinline__ bool pair_eligible(const ParseArgs &P, int64_t k, int32_t &mate_start) {
  if (!P.blob) {  // flat batch: br_batch_prepare's test on the caller's arrays
    if (!(P.flags[k] & 1u) || P.ref_id[k] != P.mate_ref_id[k]) return false;
    mate_start = P.mate_start[k];
    return true;
  }
  const uint8_t *rec = P.blob + P.rec_off[k];
  if (rec_length(P, k) < 32) return false;
  uint32_t flag = *(const u16u *)(rec + 14);
  if (!(flag & 1u)) return false;
  int32_t tid = (int32_t)*(const u32u *)rec, mtid = (int32_t)*(const u32u *)(rec + 20);
  if (tid != mtid) return false;        // raw BAM reference ids, as brec->refId() != brec->mate_refId()
  int32_t mpos = (int32_t)*(const u32u *)(rec + 24);
  mate_start = mpos < 0 ? 0 : mpos + 1; // GSam.h:344
  return true;
}

__device__ __forceinline__ bool seq_fits(const ParseArgs &P, int64_t k) {
  const uint8_t *rec = P.blob + P.rec_off[k];
  uint32_t rlen = rec_length(P, k);
  if (rlen < 32) return false;
  uint64_t l_qname = rec[8], ncig = *(const u16u *)(rec + 12);
  int32_t l_seq = (int32_t)*(const u32u *)(rec + 16);
  uint64_t ls = l_seq > 0 ? (uint64_t)l_seq : 0;
  return 32ull + l_qname + 4ull * ncig + (ls + 1) / 2 + ls <= rlen;
}

#define MATES_BIG 96
#define M_NONE (-1)
#define M_OPEN (-2)   // this record's position is in the map, waiting for its mate

__global__ void __launch_bounds__(256) k_mates(ParseArgs P) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.n_groups) return;
  uint32_t a = P.group_off[g], b = P.group_off[g + 1];
  if (b - a > MATES_BIG) { uint32_t s = atomicAdd(P.n_big_groups, 1u); P.big_groups[s] = (uint32_t)g; return; }
  for (uint32_t k = a; k < b; k++) P.mate_idx[k] = M_NONE;
  if (b - a < 2) return;
  for (uint32_t k = a; k < b; k++) {
    int32_t ms;
    if (!pair_eligible(P, k, ms)) continue;
    int32_t rs = P.ref_start[k];
    int32_t found = -1;
    for (uint32_t j = a; j < k; j++) if (P.mate_idx[j] == M_OPEN && P.ref_start[j] == ms) { found = (int32_t)j; break; }
    if (found >= 0) { P.mate_idx[k] = found; P.mate_idx[found] = (int32_t)k; }
    else {
      for (uint32_t j = a; j < k; j++) if (P.mate_idx[j] == M_OPEN && P.ref_start[j] == rs) P.mate_idx[j] = M_NONE;  // same key: overwritten
      P.mate_idx[k] = M_OPEN;
    }
  }
  for (uint32_t k = a; k < b; k++) if (P.mate_idx[k] == M_OPEN) P.mate_idx[k] = M_NONE;
}

// the same, one wave per large group: the sequential order over k stays, the searches over the
// open set run 64 wide.  mate_idx is read and written with agent-scope atomics so that lanes
// see each other's updates.
__device__ __forceinline__ int32_t ld_m(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_m(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(64) k_mates_big(ParseArgs P) {
  uint32_t nb = *P.n_big_groups;
  const int lane = threadIdx.x;
  for (uint32_t q = blockIdx.x; q < nb; q += gridDim.x) {
    uint32_t g = P.big_groups[q];
    uint32_t a = P.group_off[g], b = P.group_off[g + 1];
    for (uint32_t k = a + lane; k < b; k += 64) st_m(P.mate_idx + k, M_NONE);
    __syncthreads();
    for (uint32_t k = a; k < b; k++) {
      int32_t ms = 0;
      if (!pair_eligible(P, k, ms)) continue;   // uniform across the wave
      int32_t rs = P.ref_start[k];
      int32_t found = -1;
      for (uint32_t j0 = a; j0 < k && found < 0; j0 += 64) {
        uint32_t j = j0 + lane;
        bool hit = j < k && ld_m(P.mate_idx + j) == M_OPEN && P.ref_start[j] == ms;
        uint64_t m = __ballot(hit);
        if (m) found = (int32_t)(j0 + (uint32_t)__builtin_ctzll(m));
      }
      if (found >= 0) {
        if (lane == 0) { st_m(P.mate_idx + k, found); st_m(P.mate_idx + found, (int32_t)k); }
      } else {
        for (uint32_t j = a + lane; j < k; j += 64) if (ld_m(P.mate_idx + j) == M_OPEN && P.ref_start[j] == rs) st_m(P.mate_idx + j, M_NONE);
        if (lane == 0) st_m(P.mate_idx + k, M_OPEN);
      }
      __syncthreads();
    }
    for (uint32_t k = a + lane; k < b; k += 64) if (ld_m(P.mate_idx + k) == M_OPEN) st_m(P.mate_idx + k, M_NONE);
    __syncthreads();
  }
}

// -S: the group's shared read sequence is the first record of the group that carries one
// (src/core.cpp:353-378); seq_len[i] = l_seq for that record only
__global__ void __launch_bounds__(256) k_seq_src(ParseArgs P) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.n_groups) return;
  uint32_t a = P.group_off[g], b = P.group_off[g + 1];
  int32_t src = -1;
  if (!P.blob) {  // flat batch (br_batch_seq_source): the sequences are already ASCII at seq_off
    for (uint32_t k = a; k < b; k++) if (P.seq_off[k + 1] > P.seq_off[k]) { src = (int32_t)k; break; }
    for (uint32_t k = a; k < b; k++) P.seq_src[k] = src;
    return;
  }
  for (uint32_t k = a; k < b; k++) if (P.l_qseq[k] > 0 && seq_fits(P, k)) { src = (int32_t)k; break; }
  for (uint32_t k = a; k < b; k++) { P.seq_src[k] = src; P.seq_len[k] = ((int32_t)k == src) ? (uint32_t)P.l_qseq[k] : 0u; }
}

template <int G>
__global__ void __launch_bounds__(256) k_seq_ascii(ParseArgs P) {
  const int lane = threadIdx.x & (G - 1);
  int64_t i = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
  if (i >= P.n) return;
  uint32_t n = P.seq_off[i + 1] - P.seq_off[i];
  if (!n) return;
  const uint8_t *rec = P.blob + P.rec_off[i];
  uint32_t l_qname = rec[8], ncig = *(const u16u *)(rec + 12);
  const uint8_t *seq = rec + 32 + l_qname + 4 * ncig;
  uint8_t *dst = P.seqs + P.seq_off[i];
  const char *tbl = "=ACMGRSVTWYHKDBN";  // seq_nt16_str
  for (uint32_t k = lane; k < n; k += G) dst[k] = (uint8_t)tbl[(seq[k >> 1] >> ((~k & 1) << 2)) & 0xf];
}

// k_soa_fields: one lane per alignment of a flat batch (+ one for the closing offsets)
__global__ void __launch_bounds__(256) k_soa_fields(SoaArgs S) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t ncig = 0, max_s = 0;
  if (i <= S.n) {
    const uint64_t c0 = S.cigar_off64[i], n0 = S.name_off64[i];
    S.cigar_off[i] = (uint32_t)c0; S.name_off[i] = (uint32_t)n0;
    if (S.seq_off64) S.seq_off[i] = (uint32_t)S.seq_off64[i];
    if (i < S.n) {
      const uint64_t c1 = S.cigar_off64[i + 1], n1 = S.name_off64[i + 1];
      ncig = (uint32_t)(c1 - c0);
      if (ncig) {   // leading / trailing soft clips (sizing of the rescue buffers), as k_rec_fields
        const uint32_t *cg = S.cigar + c0;
        uint32_t w = cg[0];
        if ((w & 0xfu) == 5u && ncig > 1) w = cg[1];
        if ((w & 0xfu) == 4u) max_s = w >> 4;
        w = cg[ncig - 1];
        if ((w & 0xfu) == 5u && ncig > 1) w = cg[ncig - 2];
        if ((w & 0xfu) == 4u) max_s = max(max_s, w >> 4);
      }
      uint32_t isnew = 1;
      if (i > 0) {
        const uint64_t p0 = S.name_off64[i - 1];
        const uint32_t nl = (uint32_t)(n1 - n0);
        if ((uint32_t)(n0 - p0) == nl && (nl == 0 || same_bytes(S.names + p0, S.names + n0, nl))) isnew = 0;
      }
      S.isnew[i] = isnew;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    ncig = max(ncig, (uint32_t)__shfl_xor((int)ncig, o));
    max_s = max(max_s, (uint32_t)__shfl_xor((int)max_s, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (ncig > __builtin_nontemporal_load(S.maxima)) atomicMax(S.maxima, ncig);
    if (max_s > __builtin_nontemporal_load(S.maxima + 1)) atomicMax(S.maxima + 1, max_s);
  }
}

void launch_soa_fields(hipStream_t st, const SoaArgs &S) {
  hipLaunchKernelGGL(k_soa_fields, dim3((unsigned)((S.n + 1 + 255) / 256)), dim3(256), 0, st, S);
}
void launch_rec_fields(hipStream_t st, const ParseArgs &P) {
  if (P.n > 0) hipLaunchKernelGGL(k_rec_fields, dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, P);
}
void launch_group_off(hipStream_t st, const ParseArgs &P) {
  hipLaunchKernelGGL(k_group_off, dim3((unsigned)((P.n + 1 + 255) / 256)), dim3(256), 0, st, P);
}
void launch_rec_copy(hipStream_t st, const ParseArgs &P) {
  if (P.n > 0) hipLaunchKernelGGL((k_rec_copy<8>), dim3((unsigned)((P.n + 31) / 32)), dim3(256), 0, st, P);
}
void launch_mates(hipStream_t st, const ParseArgs &P) {
  if (P.n_groups <= 0) return;
  hipLaunchKernelGGL(k_mates, dim3((unsigned)((P.n_groups + 255) / 256)), dim3(256), 0, st, P);
  hipLaunchKernelGGL(k_mates_big, dim3(1024), dim3(64), 0, st, P);
}
void launch_seq_src(hipStream_t st, const ParseArgs &P) {
  if (P.n_groups > 0) hipLaunchKernelGGL(k_seq_src, dim3((unsigned)((P.n_groups + 255) / 256)), dim3(256), 0, st, P);
}
void launch_seq_ascii(hipStream_t st, const ParseArgs &P) {
  if (P.n > 0) hipLaunchKernelGGL((k_seq_ascii<16>), dim3((unsigned)((P.n + 15) / 16)), dim3(256), 0, st, P);
}

}  // namespace br
