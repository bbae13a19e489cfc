// ksw_extz2 for the -S soft-clip rescue (rows a9 / a10 of SURVEY.md 8a) on gfx950.
//
//   subprojects/packagefiles/ksw2/ksw2_extz2_sse.cpp:37-318 (anti-diagonal u/v/x/y difference recurrence, approximate
//   maximum, z-drop), ksw2.h ksw_backtrack, src/evaluate.cpp:284-317,397-448,548-598 (acceptance, clip segment).
//   Scores as src/evaluate.cpp:296-313: match 1, mismatch -4, N -1, gap open 4, extend 1, z-drop 40, full band,
//   EXTZ_ONLY | APPROX_MAX | APPROX_DROP.
//
// k_ksw_dp<G,K>: a systolic array in registers.  A group of G lanes owns G*K target columns (K columns per lane, two
// 16-bit cells per VGPR, packed-16 arithmetic; pair j of a lane = columns c0 + j and c0 + j + K/2, so that only pair 0
// needs a shifted operand); the query streams through the columns one column per step, so that column c holds cell
// (r, c) of anti-diagonal r with query base r - c.  A cell needs x, v of column c-1 from the step before (one wave_shr
// DPP move per stream and lane) and its own u, y: no LDS, no loads in the recurrence.  Problems follow each other
// through the array back to back: while problem A's band leaves the low columns, problem B's first anti-diagonals
// already use them, which removes the two idle triangles of an anti-diagonal sweep.  The first base of a problem
// carries a flag; a column that sees it resets its u / y and takes the next problem's target base.  Values are kept
// x4 with a 2-bit tag in the low bits (match 2 > deletion 1 > insertion 0), so that one packed max gives the winner and
// ksw2's tie order at once.  Per step every lane stores the 4-bit directions of its K columns (8 or 16 bytes) in the
// wave's tape row (row = step; chunks of 1024 rows from a pool); k_ksw_trace walks the tape with one lane per problem.
// The groups of a launch share one queue of problems per array shape (descriptors three problems ahead, through an LDS
// ring).  The approximate-maximum / z-drop bookkeeping (one scalar chain per problem in ksw2) runs in the lanes of the
// group, one problem per lane, on the u / v values the columns publish to LDS.
// k_ksw: the general kernel (any size; one wave per problem, state in LDS / HBM) for what does not fit the arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels.h"

namespace br {

#define KSW_NEG_INF (-0x40000000)

// ---------------------------------------------------------------------------
// traceback + clip segment, shared by k_ksw and k_ksw_trace
// ---------------------------------------------------------------------------
struct RawSink {
  uint32_t *raw; uint32_t n;
  __device__ __forceinline__ void push(uint32_t op, int len) {
    if (n == 0 || op != (raw[n - 1] & 0xf)) raw[n++] = ((uint32_t)len << 4) | op;
    else raw[n - 1] += (uint32_t)len << 4;
  }
};

// raw[] = the forward CIGAR reversed (forward op k = raw[n-1-k]); build_left/right_clip_segment (src/evaluate.cpp:397-448,548-598)
__device__ __forceinline__ KswRes clip_segment(const uint32_t *raw, uint32_t n, const KswProb &pr, int ez_max, uint32_t *dst) {
  int query_consumed = 0, ref_consumed = 0;
  for (uint32_t k = 0; k < n; k++) {
    uint32_t op = raw[k] & 0xf, len = raw[k] >> 4;
    if (op == 0 || op == 1) query_consumed += (int)len;
    if (op == 0 || op == 2) ref_consumed += (int)len;
  }
  int rest = (int)pr.qlen - query_consumed;  // unaligned remainder -> CLIP_OVERRIDE
  uint32_t m = 0;
  auto add = [&](uint32_t len, uint32_t op) {
    if (m && (dst[m - 1] & 0xf) == op) dst[m - 1] = (((dst[m - 1] >> 4) + len) << 4) | op;
    else dst[m++] = (len << 4) | op;
  };
  auto emit = [&](uint32_t wv, bool outermost) {
    uint32_t op = wv & 0xf, len = wv >> 4;  // 0 M, 1 I, 2 D
    if (outermost && op == 2) return;
    if (outermost && op == 1) { add(len, OP_CLIP_OVR); return; }
    add(len, op == 2 ? OP_DEL_OVR : op == 1 ? OP_INS_OVR : OP_MATCH_OVR);
  };
  if (pr.side == 0) {
    // left: sequences were reversed, so walk the forward CIGAR backwards (= raw order)
    if (rest > 0) add((uint32_t)rest, OP_CLIP_OVR);
    for (uint32_t k = 0; k < n; k++) emit(raw[k], k == 0);
  } else {
    for (uint32_t k = 0; k < n; k++) emit(raw[n - 1 - k], k == n - 1);
    if (rest > 0) add((uint32_t)rest, OP_CLIP_OVR);
  }
  KswRes rs; rs.ok = 1; rs.score = ez_max; rs.refc = ref_consumed; rs.n_ops = m;
  return rs;
}

// ---------------------------------------------------------------------------
// k_ksw: one wave per rescue problem, grid-stride.  Per-wave HBM scratch: direction matrix
// p[(qlen+tlen-1) x tlen], raw traceback ops.
// ---------------------------------------------------------------------------
#define KSW_LDS_T 2048  // u/v/x/y live in LDS up to this target length, else in HBM scratch

__global__ void __launch_bounds__(256) k_ksw(KswArgs K) {
  __shared__ uint32_t sh_uvxy[4][KSW_LDS_T];   // u | v << 8 | x << 16 | y << 24 per target position: one LDS word per cell
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t waves_total = K.n_waves;   // waves that own scratch (the grid's last block may hold idle ones)
  const int64_t wid = (int64_t)blockIdx.x * 4 + w;
  if (wid >= waves_total) return;
  uint8_t *pmat = K.scratch + (size_t)wid * K.scratch_per_wave;
  uint32_t *raw = (uint32_t *)(pmat + K.pmat_bytes);
  uint32_t *gl_uvxy = (uint32_t *)(raw + K.raw_words);
  const int8_t q = 4, e = 1; const int qe = 5; const int8_t qe2 = 10;
  const int8_t sc_mch = 1, sc_mis = -4, sc_N = -1, max_sc_v = 11;
  const int zdrop = 40;
  unsigned long long cells = 0, rescued = 0;
  const int64_t n_work = K.list ? (int64_t)*K.n_list : K.n_prob;

  for (int64_t pi = wid; pi < n_work; pi += waves_total) {
    const int64_t p = K.list ? (int64_t)K.list[pi] : pi;
    KswProb pr = K.probs[p];
    int qlen = (int)pr.qlen, tlen = (int)pr.tlen;
    const uint8_t *query = K.seq_arena + pr.seq_off, *target = query + qlen;
    KswRes rs; rs.ok = 0; rs.score = 0; rs.refc = 0; rs.n_ops = 0;
    if (qlen <= 0 || tlen <= 0) { if (lane == 0) K.results[p] = rs; continue; }
    cells += (unsigned long long)qlen * (unsigned long long)tlen;
    uint32_t *cell = tlen <= KSW_LDS_T ? &sh_uvxy[w][0] : gl_uvxy;
    auto U = [](uint32_t c) { return (int8_t)c; };
    auto V = [](uint32_t c) { return (int8_t)(c >> 8); };
    auto X = [](uint32_t c) { return (int8_t)(c >> 16); };
    auto Y = [](uint32_t c) { return (int8_t)(c >> 24); };
    auto PACK = [](int8_t u8, int8_t v8, int8_t x8, int8_t y8) {
      return (uint32_t)(uint8_t)u8 | ((uint32_t)(uint8_t)v8 << 8) | ((uint32_t)(uint8_t)x8 << 16) | ((uint32_t)(uint8_t)y8 << 24);
    };
    for (int t = lane; t < tlen; t += 64) cell[t] = 0;
    __builtin_amdgcn_wave_barrier();
    int32_t ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_score = KSW_NEG_INF; bool zdropped = false;
    int32_t H0 = 0, last_H0_t = 0;
    int last_st = -1, last_en = -1;
    int n_rows = qlen + tlen - 1;
    for (int r = 0; r < n_rows; ++r) {
      int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0;
      int en0 = r < tlen - 1 ? r : tlen - 1;
      if (tlen > KSW_LDS_T) __threadfence_block();  // cells of the previous row were written by other lanes
      if (en0 >= r && lane == 0) { uint32_t c0 = cell[r]; cell[r] = PACK(r ? q : 0, V(c0), X(c0), 0); }
      if (tlen > KSW_LDS_T) __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      int8_t cx, cv;  // x[r-1][t-1], v[r-1][t-1] for the first cell of the next chunk
      if (st0 > 0) { if (st0 - 1 >= last_st && st0 - 1 <= last_en) { uint32_t c1 = cell[st0 - 1]; cx = X(c1); cv = V(c1); } else { cx = 0; cv = 0; } }
      else { cx = 0; cv = r ? q : 0; }
      uint8_t *prow = pmat + (size_t)r * (size_t)tlen;
      for (int c = st0; c <= en0; c += 64) {
        int t = c + lane;
        bool act = t <= en0;
        int8_t ox = 0, ov = 0, ut = 0, yt = 0;
        if (act) { uint32_t cw = cell[t]; ox = X(cw); ov = V(cw); ut = U(cw); yt = Y(cw); }
        int8_t xt1 = (int8_t)__shfl_up((int)ox, 1, 64), vt1 = (int8_t)__shfl_up((int)ov, 1, 64);
        if (lane == 0) { xt1 = cx; vt1 = cv; }
        cx = (int8_t)__shfl((int)ox, 63, 64); cv = (int8_t)__shfl((int)ov, 63, 64);
        if (act) {
          uint8_t sq = target[t], sqr = query[r - t];
          int8_t sc = (sq == 4 || sqr == 4) ? sc_N : (sq == sqr ? sc_mch : sc_mis);
          int8_t z = (int8_t)(sc + qe2);
          int8_t a = (int8_t)(xt1 + vt1);
          int8_t b = (int8_t)(yt + ut);
          uint8_t d = (a > z) ? 1 : 0;
          z = z > a ? z : a;
          if (b > z) d = 2;
          z = (int8_t)((uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b);
          z = (int8_t)((uint8_t)z < (uint8_t)max_sc_v ? (uint8_t)z : (uint8_t)max_sc_v);
          int8_t nu = (int8_t)(z - vt1), nv = (int8_t)(z - ut), nx = 0, ny = 0;
          z = (int8_t)(z - q);
          a = (int8_t)(a - z);
          b = (int8_t)(b - z);
          if (a > 0) { nx = a; d |= 0x08; }
          if (b > 0) { ny = b; d |= 0x10; }
          cell[t] = PACK(nu, nv, nx, ny);
          prow[t] = d;
        }
        __builtin_amdgcn_wave_barrier();
      }
      // approximate max + z-drop (wave-uniform: every lane reads the same cells)
      if (tlen > KSW_LDS_T) __threadfence_block();
      bool stop = false;
      if (r > 0) {
        if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
          int32_t d0 = (int32_t)(uint8_t)V(cell[last_H0_t]) - qe, d1 = (int32_t)(uint8_t)U(cell[last_H0_t + 1]) - qe;
          if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; }
        } else if (last_H0_t >= st0 && last_H0_t <= en0) {
          H0 += (int32_t)(uint8_t)V(cell[last_H0_t]) - qe;
        } else {
          ++last_H0_t; H0 += (int32_t)(uint8_t)U(cell[last_H0_t]) - qe;
        }
        // ksw_apply_zdrop (rotated form)
        int tt = last_H0_t;
        if (H0 > ez_max) { ez_max = H0; ez_max_t = tt; ez_max_q = r - tt; }
        else if (tt >= ez_max_t && r - tt >= ez_max_q) {
          int tl = tt - ez_max_t, ql = (r - tt) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
          if (zdrop >= 0 && ez_max - H0 > zdrop + l * e) { zdropped = true; stop = true; }
        }
      } else { H0 = (int32_t)(uint8_t)V(cell[0]) - qe - qe; last_H0_t = 0; }
      if (stop) break;
      if (r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H0;
      last_st = st0; last_en = en0;
    }
    (void)zdropped;
    // acceptance: max >= 10 and the DP reached the last cell (src/evaluate.cpp:484,643)
    if (K.max_out && lane == 0) { K.max_out[p] = ez_max; K.raw_n[p] = 0; }
    if (ez_max < 10 || ez_score == KSW_NEG_INF || ez_max_t < 0 || ez_max_q < 0) { if (lane == 0) K.results[p] = rs; continue; }
    __threadfence_block();
    if (lane == 0) {
      // ksw_backtrack (is_rot): ops pushed from the end to the start into raw[]
      int i = ez_max_t, j = ez_max_q, state = 0;
      RawSink sk{raw, 0};
      while (i >= 0 && j >= 0) {
        int r = i + j;
        uint32_t tmp = pmat[(size_t)r * (size_t)tlen + (size_t)i];
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (state == 0) { sk.push(0, 1); --i; --j; }
        else if (state == 1 || state == 3) { sk.push(2, 1); --i; }
        else { sk.push(1, 1); --j; }
      }
      if (i >= 0) sk.push(2, i + 1);
      if (j >= 0) sk.push(1, j + 1);
      uint32_t n = sk.n;
      if (K.raw_out) { K.raw_n[p] = n; for (uint32_t k = 0; k < n && k < K.raw_cap; k++) K.raw_out[(size_t)p * K.raw_cap + k] = raw[n - 1 - k]; }
      K.results[p] = clip_segment(raw, n, pr, ez_max, K.clip_ops + (pr.seq_off + (uint64_t)p));
      rescued++;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0 && K.stats) {
    atomicAdd((unsigned long long *)&K.stats[0], cells);     // DP cells (qlen x tlen summed)
    atomicAdd((unsigned long long *)&K.stats[1], rescued);   // accepted rescues
  }
}

// ---------------------------------------------------------------------------
// k_ksw_bin: one lane per problem: its array shape (by target length), a slot in that bin's descriptor array,
// the bin's tape rows.  Degenerate problems get their (empty) result here.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int ksw_bin_of(uint32_t qlen, uint32_t tlen) {
  if (qlen > 0xffffu || qlen + tlen > (uint32_t)KSW_MAX_SPAN) return KSW_N_BINS;   // a problem spans at most two tape chunks
  if (tlen <= (uint32_t)KSW_BIN_W(0)) return 0;
  if (tlen <= (uint32_t)KSW_BIN_W(1)) return 1;
  if (tlen <= (uint32_t)KSW_BIN_W(2)) return 2;
  if (tlen <= (uint32_t)KSW_BIN_W(3)) return 3;
  return KSW_N_BINS;
}

// rows a problem occupies on its group's tape: its query bases, at least K steps between two first bases
__device__ __forceinline__ uint32_t ksw_rows_of(uint32_t qlen, int K) { return (qlen > (uint32_t)K ? qlen : (uint32_t)K) + 1u; }

#define KSW_BIN_PER_THREAD 16
__global__ void __launch_bounds__(256) k_ksw_bin(KswFastArgs A) {
  // a block bins 256 x 16 consecutive problems: counts in LDS, one global atomic per bin and block for the bases (a
  // wave-level version spent 2.3 ms per 1.6 M problems queueing on nine addresses)
  __shared__ uint32_t sh_cnt[KSW_N_BINS + 1], sh_base[KSW_N_BINS + 1], sh_left[2];   // sh_left: longest query / target left to k_ksw
  __shared__ unsigned long long sh_rows[KSW_N_BINS];
  if (threadIdx.x <= KSW_N_BINS) sh_cnt[threadIdx.x] = 0;
  if (threadIdx.x < KSW_N_BINS) sh_rows[threadIdx.x] = 0;
  if (threadIdx.x < 2) sh_left[threadIdx.x] = 0;
  __syncthreads();
  const int64_t first = (int64_t)blockIdx.x * (256 * KSW_BIN_PER_THREAD);
  uint64_t bins = 0;            // 4 bits per problem: bin + 1 (0 = none)
  uint32_t local[KSW_BIN_PER_THREAD];
#pragma unroll
  for (int k = 0; k < KSW_BIN_PER_THREAD; k++) {
    const int64_t i = first + (int64_t)k * 256 + threadIdx.x;
    local[k] = 0;
    if (i >= A.n) continue;
    const KswProb pr = A.probs[A.p0 + i];
    KswDp d; d.max = 0; d.max_t = d.max_q = -1; d.flags = 0; d.tape = 0;
    A.dp[i] = d;
    if (pr.qlen == 0 || pr.tlen == 0) {
      KswRes rs; rs.ok = 0; rs.score = 0; rs.refc = 0; rs.n_ops = 0;
      A.results[A.p0 + i] = rs;
      if (A.max_out) { A.max_out[A.p0 + i] = 0; A.raw_n[A.p0 + i] = 0; }
      continue;
    }
    const int b = pr.t_has_n ? KSW_N_BINS : ksw_bin_of(pr.qlen, pr.tlen);   // the arrays score a target base by t ^ q: no N
    bins |= (uint64_t)(b + 1) << (4 * k);
    local[k] = atomicAdd(&sh_cnt[b], 1u);
    if (b < KSW_N_BINS) atomicAdd(&sh_rows[b], (unsigned long long)ksw_rows_of(pr.qlen, b == 0 ? KSW_BIN_K(0) : b == 1 ? KSW_BIN_K(1) : b == 2 ? KSW_BIN_K(2) : KSW_BIN_K(3)));
    else { atomicMax(&sh_left[0], pr.qlen); atomicMax(&sh_left[1], pr.tlen); }
  }
  __syncthreads();
  if (threadIdx.x <= KSW_N_BINS) sh_base[threadIdx.x] = sh_cnt[threadIdx.x] ? atomicAdd(&A.counters[threadIdx.x], sh_cnt[threadIdx.x]) : 0u;
  if (threadIdx.x < KSW_N_BINS && sh_rows[threadIdx.x]) atomicAdd((unsigned long long *)(A.counters + 8) + threadIdx.x, sh_rows[threadIdx.x]);
  if (threadIdx.x < 2 && sh_left[threadIdx.x]) atomicMax(&A.counters[5 + threadIdx.x], sh_left[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KSW_BIN_PER_THREAD; k++) {
    const int bb = (int)((bins >> (4 * k)) & 15u) - 1;
    if (bb < 0) continue;
    const int64_t i = first + (int64_t)k * 256 + threadIdx.x;
    const uint32_t pos = sh_base[bb] + local[k];
    if (bb < KSW_N_BINS) {
      const KswProb pr = A.probs[A.p0 + i];
      KswDesc d; d.qt = pr.qlen | (pr.tlen << 16); d.prob = (uint32_t)(A.p0 + i); d.seq_off = pr.seq_off;
      A.desc[bb][pos] = d;
    } else A.leftover[pos] = (uint32_t)(A.p0 + i);
  }
}

// ---------------------------------------------------------------------------
// k_ksw_dp
// ---------------------------------------------------------------------------
typedef short ksw_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short ksw_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ksw_s2, a) + __builtin_bit_cast(ksw_s2, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ksw_s2, a) - __builtin_bit_cast(ksw_s2, b)); }
__device__ __forceinline__ uint32_t pk_max_i(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ksw_s2, a), __builtin_bit_cast(ksw_s2, b))); }
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ksw_u2, a), __builtin_bit_cast(ksw_u2, b))); }
__device__ __forceinline__ uint32_t pk_sign(uint32_t a) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ksw_s2, a) >> (ksw_s2)(15)); }
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }
// single instructions the compiler takes apart when it sees literal masks / shifts (and + and + or, shift + or)
__device__ __forceinline__ uint32_t v_bfi(uint32_t m, uint32_t a, uint32_t b) { uint32_t d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b)); return d; }
template <int SH>
__device__ __forceinline__ uint32_t v_lshl_or(uint32_t a, uint32_t b) { uint32_t d; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(SH), "v"(b)); return d; }
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false); }

// lanes of one wave exchange data through LDS: program order is enough for the hardware, the fence keeps the compiler from
// moving the accesses
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// scaled constants (values x4, tag in the low two bits)
#define KSW4_Q 0x00100010u       // gap open 4
#define KSW4_ZMAX 0x002c002cu    // max_sc_v 11
#define KSW_LUT_LO 0x1a1a1a2eu   // t ^ q = 0: match (1 + 10) * 4 + 2 = 46; 1..3: mismatch (-4 + 10) * 4 + 2 = 26
#define KSW_LUT_HI 0x26262626u   // 4..7: the query base is N

struct KswDpArgs {
  const KswDesc *desc; uint32_t n, n_groups;
  uint32_t *queue;                 // next problem of the bin's descriptor array
  unsigned long long *tape_used;   // bytes of the tape handed out
  uint64_t tape_cap;
  uint8_t *tape; const uint8_t *seq_arena;
  KswDp *dp; int64_t p0;
  uint32_t *leftover, *n_leftover;
  uint32_t bin;
};

#define KSW_NO_PROB 0xffffffffu

template <int G, int K>
__global__ void __launch_bounds__(256) k_ksw_dp(KswDpArgs A) {
  constexpr int P = K / 2, W = G * K, GPW = 64 / G, R = 2 * G, LB = K > 16 ? 16 : 8, RB = 64 * LB, TW = (K + 3) / 4;   // TW: dwords of target codes
  // column layout of a lane: pair j = columns c0 + j (low half) and c0 + j + P (high half), so that the left neighbours
  // of both halves of pair j are the two halves of pair j - 1: only pair 0 needs a shifted operand
  __shared__ uint32_t sh_pub[4][2][64 * P];     // [wave][v | u][lane * P + pair]: what the columns hold after the step
  __shared__ uint4 sh_mb[4][64];                // per bookkeeping lane: {first step + 1 (0 = free), qlen | tlen << 16, problem, 0}
  __shared__ uint32_t sh_cx[4][GPW];            // per group: problem + 1 of a problem that z-dropped
  __shared__ uint4 sh_ring[4][GPW][R];          // per group: descriptors of its problems n, n + 1, n + 2 (slot = sequence % R)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int gl = lane & (G - 1), grp = lane / G;
  const uint32_t g = ((uint32_t)blockIdx.x * 4u + (uint32_t)w) * GPW + (uint32_t)grp;
  const bool live = g < A.n_groups;
  const uint32_t c0 = (uint32_t)gl * K;
  auto load8 = [&](uint64_t off) { uint2 v; __builtin_memcpy(&v, A.seq_arena + off, 8); return v; };
  struct Raw { uint32_t w[TW]; };
  auto load_t = [&](uint64_t off) {   // K target codes (unaligned): 8, 16 or 16 + 8 bytes
    Raw v;
    const uint8_t *src = A.seq_arena + off;
    if (TW == 2) { uint2 a; __builtin_memcpy(&a, src, 8); v.w[0] = a.x; v.w[1] = a.y; }
    else {
      uint4 a; __builtin_memcpy(&a, src, 16); v.w[0] = a.x; v.w[1] = a.y; v.w[2 < TW ? 2 : 0] = a.z; v.w[3 < TW ? 3 : 0] = a.w;
      if (TW > 4) { uint2 b; __builtin_memcpy(&b, src + 16, 8); v.w[4 < TW ? 4 : 0] = b.x; v.w[5 < TW ? 5 : 0] = b.y; }
    }
    return v;
  };
  auto ring_get = [&](uint32_t n) { return sh_ring[w][grp][n & (R - 1)]; };     // {qt, prob, seq_off lo, hi}
  auto seq_off_of = [](uint4 d) { return (uint64_t)d.z | ((uint64_t)d.w << 32); };
  // feeder only: the next problem of the bin (the queue is shared by every group of the launch)
  // (claimed eight at a time: one atomic per problem on a single address is ~15 ns of queueing each)
  // ... while the queue is long; over its last stretch (fewer than sixteen problems per group left, judged by what the last
  // claim returned) one at a time: a group that sits on eight claimed problems while its neighbours have run dry is the
  // kernel's tail, paid once per shape and piece
  uint32_t cl_next = 0, cl_end = 0;
  auto claim = [&]() {
    if (cl_next == cl_end) {
      const uint32_t want = (cl_end < A.n && A.n - cl_end > 16u * A.n_groups) ? 8u : 1u;
      cl_next = atomicAdd(A.queue, want); cl_end = cl_next + want;
    }
    const uint32_t k = cl_next++;
    return k < A.n ? k : KSW_NO_PROB;
  };
  auto fetch = [&](uint32_t k) { uint4 d = make_uint4(0u, KSW_NO_PROB, 0u, 0u); if (k != KSW_NO_PROB) { KswDesc e = A.desc[k]; d = make_uint4(e.qt, e.prob, (uint32_t)e.seq_off, (uint32_t)(e.seq_off >> 32)); } return d; };

  sh_mb[w][lane] = make_uint4(0, 0, 0, 0);
  if (gl == 0) sh_cx[w][grp] = 0;
  for (int k = gl; k < R; k += G) sh_ring[w][grp][k] = make_uint4(0u, KSW_NO_PROB, 0u, 0u);
  wave_sync();

  // ---- feeder (lane 0 of the group): descriptors run three problems ahead of the one being fed
  int32_t fk = -1; uint32_t fi = 0, fq = 0, fprob = KSW_NO_PROB, fgap = K;
  uint64_t fbuf = 0, fnext = 0, fqn = 0, fqoff = 0;
  uint4 fd_next = make_uint4(0u, KSW_NO_PROB, 0u, 0u), fd_loaded = make_uint4(0u, KSW_NO_PROB, 0u, 0u);   // descriptor of problem fk + 1; of fk + 3 (on its way to the ring)
  uint32_t fi_pending = KSW_NO_PROB;              // queue index claimed for problem fk + 4
  if (gl == 0 && live) {
    const uint4 d0 = fetch(claim()), d1 = fetch(claim());
    sh_ring[w][grp][0] = d0; sh_ring[w][grp][1] = d1;
    fd_loaded = fetch(claim());      // problem 2: goes to the ring when problem 0 starts
    fi_pending = claim();            // problem 3
    fd_next = d0;
    if (d0.y != KSW_NO_PROB) { uint2 v = load8(seq_off_of(d0)); fqn = (uint64_t)v.x | ((uint64_t)v.y << 32); }
  }
  wave_sync();

  uint32_t U[P], Y[P], V[P], S[P], Q[P], T[P], TX[P];
  const uint32_t initu0 = (gl == 0) ? 0x00100000u : KSW4_Q;     // u of a problem's diagonal cell: q, 0 in column 0
#pragma unroll
  for (int p = 0; p < P; p++) { U[p] = Y[p] = V[p] = Q[p] = 0; S[p] = 0x00010001u; T[p] = TX[p] = 0x0c000c00u; }
  // target codes of the lane's columns c0 .. c0 + K - 1 (16 raw bytes) -> per pair the perm selector halves 0x0c00 | code
  auto make_t = [&](const Raw &raw, uint32_t *t) {
    const uint32_t *wd = raw.w;
#pragma unroll
    for (int p = 0; p < P; p++) {
      const int j0 = p, j1 = p + P;   // bytes j0 -> bits 0..7, j1 -> bits 16..23
      const uint32_t sel = (uint32_t)((j0 & 3) + ((j0 >> 2) == (j1 >> 2) ? 0 : 4)) | 0x0c00u | ((uint32_t)(j1 & 3) << 16) | 0x0c000000u;
      // v_perm_b32 {hi dword, lo dword}: bytes 0-3 = the second operand; j1's dword goes there, j0's (when different) above it
      const uint32_t lo = wd[j1 >> 2], hi = wd[j0 >> 2];
      t[p] = __builtin_amdgcn_perm(hi, lo, sel) | 0x0c000c00u;
    }
  };
  // the target bases of the problems to come: TX = next first base to arrive (problem kl), txx = the one after; the
  // descriptor after that comes from the ring when txx moves up
  uint32_t kl = 0;
  Raw txx;
#pragma unroll
  for (int k = 0; k < TW; k++) txx.w[k] = 0;
  {
    const uint4 d0 = ring_get(0), d1 = ring_get(1);
    if (d0.y != KSW_NO_PROB) make_t(load_t(seq_off_of(d0) + (d0.x & 0xffffu) + c0), TX);
    if (d1.y != KSW_NO_PROB) txx = load_t(seq_off_of(d1) + (d1.x & 0xffffu) + c0);
  }
  // ---- bookkeeping lane: one problem in flight
  bool tr_on = false; int32_t tS = 0, tq = 0, tt = 0, H0 = 0, lastT = 0, emax = 0, emax_t = -1, emax_q = -1; uint32_t tprob = 0;
  const uint16_t *pubV = (const uint16_t *)&sh_pub[w][0][0] + grp * W, *pubU = (const uint16_t *)&sh_pub[w][1][0] + grp * W;
  // u16 index of group column t in a published array
  auto pub_idx = [&](int t) { const uint32_t ut = (uint32_t)t, ln = ut / K, jj = ut - ln * K, hf = jj >= (uint32_t)P ? 1u : 0u; return ln * K + 2u * (jj - hf * P) + hf; };
  // ---- tape: the wave takes chunks of KSW_CHUNK_ROWS rows (one row = one step of the whole wave)
  uint64_t chunk_cur = 0, chunk_prev = 0;   // byte offsets in the tape (wave-uniform)
  bool dead = false;                         // the tape ran out: the wave gives its claimed problems back (k_ksw takes them)

  for (uint32_t s = 0;; s++) {
    bool busy = tr_on || (gl == 0 && fd_next.y != KSW_NO_PROB);
    if (!__any(busy)) break;
    if ((s & (KSW_CHUNK_ROWS - 1)) == 0) {
      unsigned long long off = 0;
      if (lane == 0) off = atomicAdd(A.tape_used, (unsigned long long)KSW_CHUNK_ROWS * RB);
      off = (unsigned long long)__shfl((long long)off, 0, 64);
      chunk_prev = chunk_cur; chunk_cur = off;
      if (off + (unsigned long long)KSW_CHUNK_ROWS * RB > A.tape_cap) dead = true;
    }
    if (dead) {
      if (tr_on) { A.leftover[atomicAdd(A.n_leftover, 1u)] = tprob; tr_on = false; }
      if (gl == 0) {
        // claimed but not started: fk + 1, fk + 2 (ring), fk + 3 (fd_loaded), fk + 4 (fi_pending), the claimed batch's rest
        if (fd_next.y != KSW_NO_PROB) A.leftover[atomicAdd(A.n_leftover, 1u)] = fd_next.y;
        const uint4 d2 = ring_get((uint32_t)(fk + 2));
        if (d2.y != KSW_NO_PROB) A.leftover[atomicAdd(A.n_leftover, 1u)] = d2.y;
        if (fd_loaded.y != KSW_NO_PROB) A.leftover[atomicAdd(A.n_leftover, 1u)] = fd_loaded.y;
        if (fi_pending != KSW_NO_PROB) A.leftover[atomicAdd(A.n_leftover, 1u)] = A.desc[fi_pending].prob;
        for (; cl_next < cl_end && cl_next < A.n; cl_next++) A.leftover[atomicAdd(A.n_leftover, 1u)] = A.desc[cl_next].prob;   // the rest of the claimed batch
        fd_next = make_uint4(0u, KSW_NO_PROB, 0u, 0u); fd_loaded = make_uint4(0u, KSW_NO_PROB, 0u, 0u); fi_pending = KSW_NO_PROB; fi = fq = 0;
        sh_ring[w][grp][(uint32_t)(fk + 2) & (R - 1)] = make_uint4(0u, KSW_NO_PROB, 0u, 0u);
      }
      continue;   // the next pass over the loop head leaves (nothing is busy any more)
    }

    // ---- feeder: the base that enters column 0 now
    uint32_t qin = 0, vin = 0;
    if (gl == 0) {
      bool feeding = fk >= 0 && fi < fq && sh_cx[w][grp] != fprob + 1u;
      if (!feeding && fd_next.y != KSW_NO_PROB && fgap >= (uint32_t)K && sh_mb[w][grp * G + ((uint32_t)(fk + 1) & (G - 1))].x == 0) {
        fk++;
        fq = fd_next.x & 0xffffu; fi = 0; fqoff = seq_off_of(fd_next); fprob = fd_next.y; fbuf = fqn;
        { uint2 v = load8(fqoff + 8); fnext = (uint64_t)v.x | ((uint64_t)v.y << 32); }
        sh_mb[w][grp * G + ((uint32_t)fk & (G - 1))] = make_uint4(s + 1u, fd_next.x, fd_next.y, 0u);
        // the descriptor pipeline moves one problem: fk + 2 reaches the ring, fk + 3 is fetched, fk + 4 claimed
        sh_ring[w][grp][(uint32_t)(fk + 2) & (R - 1)] = fd_loaded;
        fd_loaded = fetch(fi_pending);
        fi_pending = claim();
        fd_next = ring_get((uint32_t)(fk + 1));
        if (fd_next.y != KSW_NO_PROB) { uint2 v = load8(seq_off_of(fd_next)); fqn = (uint64_t)v.x | ((uint64_t)v.y << 32); }
        feeding = true; fgap = 0;
      }
      if (feeding) {
        uint32_t ch = (uint32_t)(fbuf >> (8u * (fi & 7u))) & 7u;
        qin = ch | (fi == 0 ? 0x8000u : 0u); vin = fi ? 16u : 0u;
        fi++;
        if ((fi & 7u) == 0) { fbuf = fnext; uint2 v = load8(fqoff + fi + 8); fnext = (uint64_t)v.x | ((uint64_t)v.y << 32); }
      }
      fgap++;
    }
    wave_sync();

    // ---- what enters pair 0: low half from the lane before (or the feeder), high half from this lane's pair P - 1
    uint32_t qp = wave_shr1(Q[P - 1]), vp = wave_shr1(V[P - 1]), sp = wave_shr1(S[P - 1]);
    if (gl == 0) { qp = qin << 16; vp = vin << 16; sp = (vin + 1u) << 16; }
    const uint32_t qb = __builtin_amdgcn_alignbit(Q[P - 1], qp, 16), vb = __builtin_amdgcn_alignbit(V[P - 1], vp, 16),
                   sb = __builtin_amdgcn_alignbit(S[P - 1], sp, 16);
    const bool leaving = (Q[P - 2 >= 0 ? P - 2 : 0] >> 31) != 0;   // a first base reaches the lane's last column in this step
    // ---- cells, pairs in descending order: pair j reads what pair j - 1 held after the step before
    uint32_t D[P];
    const uint32_t m3 = 0x00030003u;
#pragma unroll
    for (int p = P - 1; p >= 0; p--) {
      const uint32_t qn = p ? Q[p - 1] : qb, v1 = p ? V[p - 1] : vb, a = p ? S[p - 1] : sb;
      uint32_t fm = pk_sign(qn);                                 // first base of a problem: new target base, u = q, y = 0
      asm("" : "+v"(fm));                                        // keeps the selects below as v_bfi_b32 (else: per-half compares + cndmasks)
      T[p] = bfi32(fm, TX[p], T[p]);
      const uint32_t ue = bfi32(fm, p == 0 ? initu0 : KSW4_Q, U[p]), ye = Y[p] & ~fm;
      const uint32_t z = __builtin_amdgcn_perm(KSW_LUT_HI, KSW_LUT_LO, T[p] ^ qn) & 0x00ff00ffu;
      const uint32_t b = pk_add(ye, ue);
      const uint32_t wv = pk_max_i(pk_max_i(z, a), b);          // tags: ties go to match, then deletion
      const uint32_t zc = pk_min_u(wv & 0xfffcfffcu, KSW4_ZMAX);
      const uint32_t nv = pk_sub(zc, ue);
      U[p] = pk_sub(zc, v1);
      const uint32_t zq = pk_sub(zc, KSW4_Q);
      const uint32_t xt = pk_max_i(pk_sub(a, zq), 0x00010001u);  // x * 4 + 1: the deletion tag of the next column's a
      const uint32_t yn = pk_max_i(pk_sub(b, zq), 0u);
      V[p] = nv; Y[p] = yn; S[p] = pk_add(xt, nv);
      // direction nibble: winner tag | x continues << 2 | y continues << 3
      D[p] = v_lshl_or<1>(pk_min_u(yn, 0x00040004u), v_bfi(m3, wv, pk_min_u(xt, 0x00050005u)));
      Q[p] = qn;
    }
    {
      // pairs 4k .. 4k + 3 share a dword: low halves in bits 0..15, high halves in bits 16..31
      uint32_t word[LB / 4];
#pragma unroll
      for (int k = 0; k < LB / 4; k++) {
        word[k] = 4 * k < P ? D[4 * k < P ? 4 * k : 0] : 0u;
        if (4 * k + 1 < P) word[k] = v_lshl_or<4>(D[4 * k + 1 < P ? 4 * k + 1 : 0], word[k]);
        if (4 * k + 2 < P) word[k] = v_lshl_or<8>(D[4 * k + 2 < P ? 4 * k + 2 : 0], word[k]);
        if (4 * k + 3 < P) word[k] = v_lshl_or<12>(D[4 * k + 3 < P ? 4 * k + 3 : 0], word[k]);
      }
      uint8_t *dst = A.tape + chunk_cur + (size_t)(s & (KSW_CHUNK_ROWS - 1)) * RB + (size_t)lane * LB;
      if (LB == 8) *(uint2 *)dst = make_uint2(word[0], word[1]);
      else *(uint4 *)dst = make_uint4(word[0], word[1], word[LB / 4 > 2 ? 2 : 0], word[LB / 4 > 3 ? 3 : 0]);
    }

    // ---- a first base leaves the lane: the next problem's target moves up
    if (leaving) {
      make_t(txx, TX);
      kl++;
      const uint4 dn = ring_get(kl + 1);
      if (dn.y != KSW_NO_PROB) txx = load_t(seq_off_of(dn) + (dn.x & 0xffffu) + c0);
    }

    // ---- publish v, u; approximate maximum and z-drop of the problems in flight
#pragma unroll
    for (int p = 0; p < P; p++) { sh_pub[w][0][lane * P + p] = V[p]; sh_pub[w][1][lane * P + p] = U[p]; }
    wave_sync();
    if (!tr_on) {
      const uint4 mb = sh_mb[w][lane];
      if (mb.x) { tr_on = true; tS = (int32_t)(mb.x - 1u); tq = (int32_t)(mb.y & 0xffffu); tt = (int32_t)(mb.y >> 16); tprob = mb.z; H0 = 0; lastT = 0; emax = 0; emax_t = emax_q = -1; }
    }
    if (tr_on) {
      const int r = (int32_t)s - tS;
      const int st0 = r - tq + 1 > 0 ? r - tq + 1 : 0, en0 = r < tt - 1 ? r : tt - 1;
      bool stop = false;
      if (r == 0) { H0 = (int32_t)(pubV[0] >> 2) - 10; lastT = 0; }
      else {
        const bool in0 = lastT >= st0 && lastT <= en0, in1 = lastT + 1 >= st0 && lastT + 1 <= en0;
        const int t1 = lastT + 1 < W ? lastT + 1 : W - 1;
        const int32_t d0 = (int32_t)(pubV[pub_idx(lastT)] >> 2) - 5, d1 = (int32_t)(pubU[pub_idx(t1)] >> 2) - 5;
        if (in0 && in1) { if (d0 > d1) H0 += d0; else { H0 += d1; ++lastT; } }
        else if (in0) H0 += d0;
        else { ++lastT; H0 += d1; }
        const int tc = lastT;
        if (H0 > emax) { emax = H0; emax_t = tc; emax_q = r - tc; }
        else if (tc >= emax_t && r - tc >= emax_q) {
          int tl = tc - emax_t, ql = (r - tc) - emax_q, l = tl > ql ? tl - ql : ql - tl;
          if (emax - H0 > 40 + l) stop = true;
        }
      }
      const bool last = r == tq + tt - 2;
      if (stop || last) {
        KswDp d; d.max = emax; d.max_t = emax_t; d.max_q = emax_q;
        d.flags = 1u | ((last && !stop) ? 2u : 0u) | (A.bin << 8);
        // rows of the problem: tS .. s, in the current chunk or starting in the one before
        const uint32_t srow = (uint32_t)tS & (KSW_CHUNK_ROWS - 1);
        const bool same = ((uint32_t)tS / KSW_CHUNK_ROWS) == (s / KSW_CHUNK_ROWS);
        d.tape = (same ? chunk_cur : chunk_prev) + (uint64_t)srow * RB + (uint64_t)(grp * G) * LB;
        d.tape2 = chunk_cur + (uint64_t)(grp * G) * LB;
        d.split = same ? 0xffffffffu : (uint32_t)KSW_CHUNK_ROWS - srow; d.pad = 0;
        A.dp[(int64_t)tprob - A.p0] = d;
        if (stop) sh_cx[w][grp] = tprob + 1u;
        sh_mb[w][lane] = make_uint4(0, 0, 0, 0);
        tr_on = false;
      }
    }
    wave_sync();
  }
}

// k_ksw_unclaimed: after a bin's k_ksw_dp: problems nobody took from the queue (every wave ran out of tape) go to the
// general kernel's list.  Normally the queue is empty and this is a no-op.
__global__ void __launch_bounds__(256) k_ksw_unclaimed(KswDpArgs A) {
  const uint32_t q0 = *A.queue;
  for (uint64_t k = (uint64_t)q0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < A.n; k += (uint64_t)gridDim.x * blockDim.x)
    A.leftover[atomicAdd(A.n_leftover, 1u)] = A.desc[k].prob;
}

// ---------------------------------------------------------------------------
// k_ksw_trace: one lane per problem: acceptance (src/evaluate.cpp:484,643), ksw_backtrack over the tape, clip segment
// ---------------------------------------------------------------------------
// list / n_list: the problems of one array shape (its descriptor array), so that a shape's tracebacks can run beside the
// next shape's DP kernel; null: every problem of the piece
__global__ void __launch_bounds__(256) k_ksw_trace(KswFastArgs A, const KswDesc *list, uint32_t n_list) {
  const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long rescued = 0, cells = 0;
  const bool have = list ? li < (int64_t)n_list : li < A.n;
  const int64_t i = !have ? 0 : list ? (int64_t)list[li].prob - A.p0 : li;
  if (have) {
    const KswDp d = A.dp[i];
    if (d.flags & 1u) {
      const int64_t p = A.p0 + i;
      const KswProb pr = A.probs[p];
      cells = (unsigned long long)pr.qlen * pr.tlen;
      KswRes rs; rs.ok = 0; rs.score = 0; rs.refc = 0; rs.n_ops = 0;
      if (A.max_out) { A.max_out[p] = d.max; A.raw_n[p] = 0; }
      if (d.max < 10 || !(d.flags & 2u) || d.max_t < 0 || d.max_q < 0) A.results[p] = rs;
      else {
        const int b = (int)(d.flags >> 8);
        const uint32_t K = (uint32_t)KSW_BIN_K(b), P = K / 2, lb = (uint32_t)KSW_BIN_LANEBYTES(b), rb = 64u * lb;
        const uint8_t *tp = A.tape + d.tape, *tp2 = A.tape + d.tape2;
        uint32_t *raw = A.raw_ops + (pr.seq_off + (uint64_t)p);
        int ci = d.max_t, cj = d.max_q, state = 0;
        RawSink sk{raw, 0};
        while (ci >= 0 && cj >= 0) {
          // column ci = lane ci / K, pair (ci % K) % P, half (ci % K) / P; pairs 0-3 sit in the lane's first dword
          const uint32_t r = (uint32_t)(ci + cj), ln = (uint32_t)ci / K, jj = (uint32_t)ci - ln * K, hf = jj >= P ? 1u : 0u, pj = jj - hf * P;
          const uint8_t *rowp = r < d.split ? tp + (size_t)r * rb : tp2 + (size_t)(r - d.split) * rb;
          const uint32_t word = *(const uint32_t *)(rowp + (size_t)ln * lb + (pj >> 2) * 4u);
          const uint32_t nib = (word >> (hf * 16u + 4u * (pj & 3u))) & 0xfu;
          const uint32_t tmp = (2u - (nib & 3u)) | ((nib & 0xcu) << 1);
          if (state == 0) state = tmp & 7;
          else if (!(tmp >> (state + 2) & 1)) state = 0;
          if (state == 0) state = tmp & 7;
          if (state == 0) { sk.push(0, 1); --ci; --cj; }
          else if (state == 1 || state == 3) { sk.push(2, 1); --ci; }
          else { sk.push(1, 1); --cj; }
        }
        if (ci >= 0) sk.push(2, ci + 1);
        if (cj >= 0) sk.push(1, cj + 1);
        const uint32_t n = sk.n;
        if (A.raw_out) { A.raw_n[p] = n; for (uint32_t k = 0; k < n && k < A.raw_cap; k++) A.raw_out[(size_t)p * A.raw_cap + k] = raw[n - 1 - k]; }
        A.results[p] = clip_segment(raw, n, pr, d.max, A.clip_ops + (pr.seq_off + (uint64_t)p));
        rescued = 1;
      }
    }
  }
  uint64_t m = __ballot(rescued != 0);
  if (lane == 0 && m && A.stats) atomicAdd((unsigned long long *)&A.stats[1], (unsigned long long)__popcll(m));
  for (int d = 32; d >= 1; d >>= 1) cells += (unsigned long long)__shfl_xor((long long)cells, d, 64);   // qlen x tlen of the stats line
  if (lane == 0 && cells && A.stats) atomicAdd((unsigned long long *)&A.stats[0], cells);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
size_t ksw_prob_bytes() { return sizeof(KswProb); }
size_t ksw_res_bytes() { return sizeof(KswRes); }

void launch_ksw(hipStream_t st, const KswArgs &K, int n_blocks) {
  if (K.n_prob <= 0 || n_blocks <= 0 || K.n_waves <= 0) return;
  hipLaunchKernelGGL(k_ksw, dim3(n_blocks), dim3(256), 0, st, K);
}

void launch_ksw_bin(hipStream_t st, const KswFastArgs &A) {
  if (A.n <= 0) return;
  const int64_t per_block = 256 * KSW_BIN_PER_THREAD;
  hipLaunchKernelGGL(k_ksw_bin, dim3((unsigned)((A.n + per_block - 1) / per_block)), dim3(256), 0, st, A);
}

void launch_ksw_dp(hipStream_t st, const KswFastArgs &A, int bin) {
  if (!A.n_bin[bin] || !A.n_groups[bin]) return;
  KswDpArgs D{};
  D.desc = A.desc[bin]; D.n = A.n_bin[bin]; D.n_groups = A.n_groups[bin];
  D.queue = A.counters + 16 + bin; D.tape_used = (unsigned long long *)(A.counters + 24); D.tape_cap = A.tape_cap;
  D.tape = A.tape; D.seq_arena = A.seq_arena; D.dp = A.dp; D.p0 = A.p0;
  D.leftover = A.leftover; D.n_leftover = A.counters + KSW_N_BINS; D.bin = (uint32_t)bin;
  const uint32_t gpw = 64u / (uint32_t)KSW_BIN_G(bin);
  const uint32_t blocks = (A.n_groups[bin] + 4u * gpw - 1u) / (4u * gpw);
  switch (bin) {
    case 0: hipLaunchKernelGGL((k_ksw_dp<KSW_BIN_G(0), KSW_BIN_K(0)>), dim3(blocks), dim3(256), 0, st, D); break;
    case 1: hipLaunchKernelGGL((k_ksw_dp<KSW_BIN_G(1), KSW_BIN_K(1)>), dim3(blocks), dim3(256), 0, st, D); break;
    case 2: hipLaunchKernelGGL((k_ksw_dp<KSW_BIN_G(2), KSW_BIN_K(2)>), dim3(blocks), dim3(256), 0, st, D); break;
    default: hipLaunchKernelGGL((k_ksw_dp<KSW_BIN_G(3), KSW_BIN_K(3)>), dim3(blocks), dim3(256), 0, st, D); break;
  }
  hipLaunchKernelGGL(k_ksw_unclaimed, dim3(64), dim3(256), 0, st, D);
}

// groups of a bin that are resident at once (blocks per CU from the occupancy query x 4 waves x groups per wave)
uint32_t ksw_dp_resident_groups(int bin, int n_cu) {
  int blocks = 0;
  hipError_t e;
  switch (bin) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_ksw_dp<KSW_BIN_G(0), KSW_BIN_K(0)>, 256, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_ksw_dp<KSW_BIN_G(1), KSW_BIN_K(1)>, 256, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_ksw_dp<KSW_BIN_G(2), KSW_BIN_K(2)>, 256, 0); break;
    default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_ksw_dp<KSW_BIN_G(3), KSW_BIN_K(3)>, 256, 0); break;
  }
  if (e != hipSuccess || blocks < 1) blocks = 3;
  return (uint32_t)blocks * (uint32_t)n_cu * 4u * (64u / (uint32_t)KSW_BIN_G(bin));
}

void launch_ksw_trace(hipStream_t st, const KswFastArgs &A, int bin) {
  const int64_t n = bin < 0 ? A.n : (int64_t)A.n_bin[bin];
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ksw_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, bin < 0 ? (const KswDesc *)nullptr : A.desc[bin], (uint32_t)n);
}

}  // namespace br
