// ============================================================================
// TEST INFRASTRUCTURE ONLY -- scalar restatement of ksw_extz2_sse as bramble
// calls it (see oracle_core.hpp header for the oracle's role and pinning).
//
// Follows subprojects/packagefiles/ksw2/ksw2_extz2_sse.cpp (in /root/reference)
// for the DP; ksw2.h helpers (absent from the reference tree; lh3/ksw2@289609b per
// subprojects/ksw2.wrap) are restated from the published upstream header.  No
// reference test exercises this path, so there is no reference-held vector for it.
// PINNED INDEPENDENTLY instead: tests/ksw2_gotoh.c is a differently structured
// implementation (full-matrix int32 Gotoh, traceback by comparing matrix values,
// derived from the in-tree kernel source alone) and tests/test_ksw2_pinned.py
// demands equal score / max / end cell / CIGAR on 10^5 random rescue-shaped pairs
// and on the committed vectors tests/golden/ksw2_cases.json.
// ============================================================================
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

enum { KSW_EZ_SCORE_ONLY = 0x01, KSW_EZ_RIGHT = 0x02, KSW_EZ_GENERIC_SC = 0x04,
       KSW_EZ_APPROX_MAX = 0x08, KSW_EZ_APPROX_DROP = 0x10, KSW_EZ_EXTZ_ONLY = 0x40,
       KSW_EZ_REV_CIGAR = 0x80 };

struct ksw_extz {
  uint32_t max = 0; bool zdropped = false;  // `uint32_t max:31, zdropped:1` upstream
  int max_q = -1, max_t = -1, mqe = 0, mqe_t = -1, mte = 0, mte_q = -1, score = 0;
  int reach_end = 0;
  std::vector<uint32_t> cigar;  // n_cigar == cigar.size()
};

// ksw2.h: ksw_reset_extz
static inline void ksw_reset_extz(ksw_extz *ez) {
  ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
  ez->max = 0; ez->score = ez->mqe = ez->mte = -0x40000000;
  ez->cigar.clear(); ez->zdropped = false; ez->reach_end = 0;
}

// ksw2.h: ksw_apply_zdrop (is_rot == 1 form: a = r, b = t)
static inline int ksw_apply_zdrop_rot(ksw_extz *ez, int32_t H, int r, int t, int zdrop, int8_t e) {
  if (H > (int32_t)ez->max) {
    ez->max = (uint32_t)H; ez->max_t = t; ez->max_q = r - t;
  } else if (t >= ez->max_t && r - t >= ez->max_q) {
    int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
    l = tl > ql ? tl - ql : ql - tl;
    if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) { ez->zdropped = true; return 1; }
  }
  return 0;
}

// ksw2.h: ksw_push_cigar
static inline void ksw_push_cigar(std::vector<uint32_t> &cigar, uint32_t op, int len) {
  if (cigar.empty() || op != (cigar.back() & 0xf)) cigar.push_back((uint32_t)len << 4 | op);
  else cigar.back() += (uint32_t)len << 4;
}

// ksw2.h: ksw_backtrack (is_rot = 1, is_rev = 0, min_intron_len = 0).
// p is addressed as p[r][t - off[r]]; rows are stored in a flat vector with
// per-row base pbase[r].
static inline void ksw_backtrack_rot(const std::vector<uint8_t> &p, const std::vector<size_t> &pbase,
                                     const std::vector<int> &off, const std::vector<int> &off_end,
                                     int i0, int j0, std::vector<uint32_t> &cigar) {
  int i = i0, j = j0, r, state = 0;
  cigar.clear();
  while (i >= 0 && j >= 0) {
    int force_state = -1;
    uint32_t tmp;
    r = i + j;
    if (i < off[r]) force_state = 2;
    if (i > off_end[r]) force_state = 1;
    tmp = force_state < 0 ? p[pbase[r] + (size_t)(i - off[r])] : 0;
    if (state == 0) state = tmp & 7;
    else if (!(tmp >> (state + 2) & 1)) state = 0;
    if (state == 0) state = tmp & 7;
    if (force_state >= 0) state = force_state;
    if (state == 0) { ksw_push_cigar(cigar, 0, 1); --i; --j; }
    else if (state == 1 || state == 3) { ksw_push_cigar(cigar, 2, 1); --i; }
    else { ksw_push_cigar(cigar, 1, 1); --j; }
  }
  if (i >= 0) ksw_push_cigar(cigar, 2, i + 1);
  if (j >= 0) ksw_push_cigar(cigar, 1, j + 1);
  for (size_t k = 0; k < cigar.size() >> 1; ++k) std::swap(cigar[k], cigar[cigar.size() - 1 - k]);
}

// ksw2_extz2_sse.cpp:37-318 for flag = EXTZ_ONLY|APPROX_MAX|APPROX_DROP, w = -1
// (full band), end_bonus = 0, left-aligned gaps, with CIGAR.  One int8 lane per
// cell; cells outside [st0,en0] (which the SSE code computes as padding) are
// never read by a valid cell or by the traceback, so they are not computed.
static inline void ksw_extz2_scalar(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                                    int8_t m, const int8_t *mat, int8_t q, int8_t e, int zdrop,
                                    ksw_extz *ez) {
  ksw_reset_extz(ez);
  if (m <= 0 || qlen <= 0 || tlen <= 0) return;
  int qe = q + e;
  int8_t qe2 = (int8_t)((q + e) * 2);
  int8_t sc_mch = mat[0], sc_mis = mat[1];
  int8_t sc_N = mat[m * m - 1] == 0 ? (int8_t)(-e) : mat[m * m - 1];
  int8_t max_sc_v = (int8_t)(mat[0] + (q + e) * 2);
  int w = tlen > qlen ? tlen : qlen;
  int wl = w, wr = w;
  int max_sc = mat[0], min_sc = mat[1];
  for (int t = 1; t < m * m; ++t) {
    max_sc = max_sc > mat[t] ? max_sc : mat[t];
    min_sc = min_sc < mat[t] ? min_sc : mat[t];
  }
  (void)max_sc;
  if (-min_sc > 2 * (q + e)) return;

  std::vector<int8_t> u(tlen + 1, 0), v(tlen + 1, 0), x(tlen + 1, 0), y(tlen + 1, 0);
  std::vector<uint8_t> qr(qlen);
  for (int t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
  int n_rows = qlen + tlen - 1;
  std::vector<int> off(n_rows, 0), off_end(n_rows, -1);
  std::vector<size_t> pbase(n_rows, 0);
  std::vector<uint8_t> p;
  p.reserve((size_t)qlen * tlen);

  int32_t H0 = 0, last_H0_t = 0;
  int last_st = -1, last_en = -1;
  for (int r = 0; r < n_rows; ++r) {
    int st = 0, en = tlen - 1;
    if (st < r - qlen + 1) st = r - qlen + 1;
    if (en > r) en = r;
    if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
    if (en > (r + wl) >> 1) en = (r + wl) >> 1;
    if (st > en) { ez->zdropped = true; break; }
    int st0 = st, en0 = en;
    // boundary conditions (ksw2_extz2_sse.cpp:131-137), expressed per cell:
    // the cell at t needs x[r-1][t-1], v[r-1][t-1]; at t == 0 they are 0 and
    // (r ? q : 0); otherwise they were computed in the previous row when
    // t-1 lies in [last_st,last_en], else 0.
    if (en0 >= r) { y[r] = 0; u[r] = r ? q : 0; }
    const uint8_t *qrr = qr.data() + (qlen - 1 - r);  // qrr[t] valid for t in [st0,en0]
    off[r] = st0; off_end[r] = en0; pbase[r] = p.size();
    p.resize(p.size() + (size_t)(en0 - st0 + 1));
    uint8_t *pr = p.data() + pbase[r];
    int8_t x1, v1;
    if (st0 > 0) {
      if (st0 - 1 >= last_st && st0 - 1 <= last_en) { x1 = x[st0 - 1]; v1 = v[st0 - 1]; }
      else { x1 = 0; v1 = 0; }
    } else { x1 = 0; v1 = r ? q : 0; }
    for (int t = st0; t <= en0; ++t) {
      uint8_t sq = target[t], sqr = qrr[t];
      int8_t s = (sq == (uint8_t)(m - 1) || sqr == (uint8_t)(m - 1)) ? sc_N : (sq == sqr ? sc_mch : sc_mis);
      int8_t z = (int8_t)(s + qe2);
      int8_t xt1 = x1, vt1 = v1;
      x1 = x[t]; v1 = v[t];  // become x[r-1][t], v[r-1][t] for the next cell
      int8_t a = (int8_t)(xt1 + vt1);
      int8_t ut = u[t];
      int8_t b = (int8_t)(y[t] + ut);
      uint8_t d = (a > z) ? 1 : 0;
      z = z > a ? z : a;                       // signed max (SSE4.1 form; equal to the SSE2 form for in-range cells)
      if (b > z) d = 2;
      z = (int8_t)((uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b);         // _mm_max_epu8
      z = (int8_t)((uint8_t)z < (uint8_t)max_sc_v ? (uint8_t)z : (uint8_t)max_sc_v);  // _mm_min_epu8
      u[t] = (int8_t)(z - vt1);
      v[t] = (int8_t)(z - ut);
      z = (int8_t)(z - q);
      a = (int8_t)(a - z);
      b = (int8_t)(b - z);
      if (a > 0) { x[t] = a; d |= 0x08; } else x[t] = 0;
      if (b > 0) { y[t] = b; d |= 0x10; } else y[t] = 0;
      pr[t - st0] = d;
    }
    // approximate max (ksw2_extz2_sse.cpp:284-300)
    if (r > 0) {
      if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
        int32_t d0 = (int32_t)(uint8_t)v[last_H0_t] - qe;
        int32_t d1 = (int32_t)(uint8_t)u[last_H0_t + 1] - qe;
        if (d0 > d1) H0 += d0;
        else { H0 += d1; ++last_H0_t; }
      } else if (last_H0_t >= st0 && last_H0_t <= en0) {
        H0 += (int32_t)(uint8_t)v[last_H0_t] - qe;
      } else {
        ++last_H0_t; H0 += (int32_t)(uint8_t)u[last_H0_t] - qe;
      }
      if (ksw_apply_zdrop_rot(ez, H0, r, last_H0_t, zdrop, e)) break;
    } else { H0 = (int32_t)(uint8_t)v[0] - qe - qe; last_H0_t = 0; }
    if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
    last_st = st0; last_en = en0;
  }
  // backtrack (ksw2_extz2_sse.cpp:306-317): with APPROX_MAX mqe stays NEG_INF,
  // so only the third branch can fire.
  if (!ez->zdropped && ez->mqe + 0 > (int)ez->max) {
    // unreachable with APPROX_MAX (mqe is never updated)
  } else if (ez->max_t >= 0 && ez->max_q >= 0) {
    ksw_backtrack_rot(p, pbase, off, off_end, ez->max_t, ez->max_q, ez->cigar);
  }
}

}  // namespace orc
