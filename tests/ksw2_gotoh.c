/* TEST INFRASTRUCTURE ONLY -- an independent checker for the oracle's ksw2 restatement (oracle/oracle_ksw2.hpp) and,
 * through it, for the k_ksw HIP kernel.
 *
 * What it is: the textbook full-matrix Gotoh extension alignment -- three int32 matrices H / E / F filled row by row,
 * no anti-diagonals, no difference encoding, no int8 arithmetic, no direction bytes -- followed by the three steps
 * ksw_extz2_sse performs as bramble calls it (flag = EXTZ_ONLY | APPROX_MAX | APPROX_DROP, w = -1, end_bonus = 0;
 * src/evaluate.cpp:296-313), each restated on the MATRICES instead of on the kernel's u/v/x/y vectors:
 *
 *   1. the approximate maximum: a greedy walk over H from cell (0,0), one anti-diagonal at a time, to the better of the
 *      cell below (target + 1) and the cell to the right (query + 1), ties to the cell below
 *      (subprojects/packagefiles/ksw2/ksw2_extz2_sse.cpp:284-297: d0 = H(t, q+1) - H(t, q), d1 = H(t+1, q) - H(t, q),
 *      "if (d0 > d1) stay else ++t"); cell (0,0) itself is never a maximum candidate (:298-299 skip the update at r = 0);
 *   2. the z-drop rule of ksw_apply_zdrop along that walk (max - H > zdrop + e * |dt - dq|);
 *   3. the traceback from the maximum cell, by comparing matrix values where the kernel consults its direction byte:
 *      H(i,j) comes from the diagonal unless E is strictly greater, from E unless F is strictly greater than both
 *      (:186-193 "d = a > z ? 1 : 0 ... d = b > z ? 2 : d"); a gap state continues while extending the gap was strictly
 *      better than opening it from H (:196-201 "d |= a > 0 ? 0x08", "d |= b > 0 ? 0x10", where a, b are E - H + q and
 *      F - H + q of the cell); leftovers at the matrix edge become one D / one I.
 *
 * The DP recurrences and the meaning of every direction bit are read off the in-tree kernel source cited above; the
 * order of the traceback's tests is the one any consumer of those bits must follow.  ksw2.h itself (ksw_backtrack,
 * ksw_apply_zdrop) is not in the reference tree -- SURVEY.md section 8c / Appendix A.
 *
 * Scores: match +1, mismatch -4, anything against N (code 4) -e = -1 (mat[24] == 0 selects sc_N = -e, :67), gap open 4,
 * gap extend 1 (a gap of length l costs 4 + l), z-drop 40.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-0x40000000)

static int code_of(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

static void push(uint32_t *cig, int *n, int cap, uint32_t op, uint32_t len) {
  if (*n > 0 && (cig[*n - 1] & 0xfu) == op) { cig[*n - 1] += len << 4; return; }
  if (*n < cap) cig[*n] = (len << 4) | op;
  (*n)++;
}

/* Returns the number of CIGAR ops (forward order, BAM-packed: 0 = M, 1 = I, 2 = D) or -1 on allocation failure.
 * out[0] = score (H of the last cell when the walk got there without a z-drop, else NEG_INF), out[1] = max,
 * out[2] = max_t, out[3] = max_q, out[4] = zdropped. */
int gotoh_extz(const char *tseq, const char *qseq, int32_t out[5], uint32_t *cigar, int cap) {
  const int sc_mch = 1, sc_mis = -4, go = 4, ge = 1, zdrop = 40, sc_n = -ge;
  const int tl = (int)strlen(tseq), ql = (int)strlen(qseq);
  out[0] = NEG_INF; out[1] = 0; out[2] = -1; out[3] = -1; out[4] = 0;
  if (tl <= 0 || ql <= 0) return 0;
  const size_t W = (size_t)ql;
  int32_t *H = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)tl * W);
  if (!H) return -1;
  int32_t *E = H + (size_t)tl * W, *F = E + (size_t)tl * W;
#define AT(M, i, j) M[(size_t)(i) * W + (size_t)(j)]
  /* boundary: the alignment starts before (0,0); a leading gap of length l costs go + l * ge */
  for (int i = 0; i < tl; i++) {
    const int ti = code_of(tseq[i]);
    for (int j = 0; j < ql; j++) {
      const int qj = code_of(qseq[j]);
      const int s = (ti == 4 || qj == 4) ? sc_n : (ti == qj ? sc_mch : sc_mis);
      const int32_t h_diag = (i && j) ? AT(H, i - 1, j - 1) : (i ? -(go + i * ge) : (j ? -(go + j * ge) : 0));
      const int32_t h_up = i ? AT(H, i - 1, j) : -(go + (j + 1) * ge);      /* H(i-1, j): one target base less */
      const int32_t h_left = j ? AT(H, i, j - 1) : -(go + (i + 1) * ge);    /* H(i, j-1): one query base less */
      const int32_t e_up = i ? AT(E, i - 1, j) : NEG_INF;                   /* no deletion is open on the boundary */
      const int32_t f_left = j ? AT(F, i, j - 1) : NEG_INF;
      const int32_t e = (h_up - go > e_up ? h_up - go : e_up) - ge;         /* deletion: consumes target i */
      const int32_t f = (h_left - go > f_left ? h_left - go : f_left) - ge; /* insertion: consumes query j */
      int32_t h = h_diag + s;
      if (e > h) h = e;
      if (f > h) h = f;
      AT(E, i, j) = e; AT(F, i, j) = f; AT(H, i, j) = h;
    }
  }
  /* 1 + 2: greedy walk, maximum and z-drop */
  int32_t max = 0; int max_t = -1, max_q = -1, zdropped = 0;
  int i = 0, j = 0;
  for (int r = 1; r <= tl + ql - 2; r++) {
    const int can_right = j + 1 < ql, can_down = i + 1 < tl;
    if (can_right && can_down) {
      const int32_t d0 = AT(H, i, j + 1) - AT(H, i, j), d1 = AT(H, i + 1, j) - AT(H, i, j);
      if (d0 > d1) j++; else i++;
    } else if (can_right) j++;
    else i++;
    const int32_t h = AT(H, i, j);
    if (h > max) { max = h; max_t = i; max_q = j; }
    else if (i >= max_t && j >= max_q) {
      const int dt = i - max_t, dq = j - max_q, l = dt > dq ? dt - dq : dq - dt;
      if (max - h > zdrop + l * ge) { zdropped = 1; break; }
    }
  }
  if (!zdropped) out[0] = (tl + ql - 2 >= 1) ? AT(H, tl - 1, ql - 1) : AT(H, 0, 0);
  out[1] = max; out[2] = max_t; out[3] = max_q; out[4] = zdropped;
  /* 3: traceback from the maximum cell (collected backwards, reversed at the end) */
  int n = 0;
  if (max_t >= 0 && max_q >= 0) {
    uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(tl + ql + 2));
    if (!rev) { free(H); return -1; }
    int m = 0, state = 0;  /* 0: in H, 1: inside a deletion, 2: inside an insertion */
    i = max_t; j = max_q;
    while (i >= 0 && j >= 0) {
      const int32_t h = AT(H, i, j), e = AT(E, i, j), f = AT(F, i, j);
      const int ti = code_of(tseq[i]), qj = code_of(qseq[j]);
      const int s = (ti == 4 || qj == 4) ? sc_n : (ti == qj ? sc_mch : sc_mis);
      const int32_t h_diag = (i && j) ? AT(H, i - 1, j - 1) : (i ? -(go + i * ge) : (j ? -(go + j * ge) : 0));
      /* which of the three made H(i,j): diagonal unless E strictly greater; E unless F strictly greater than both */
      int best = 0;
      int32_t z = h_diag + s;
      if (e > z) { best = 1; z = e; }
      if (f > z) best = 2;
      /* a gap entered at the cell after this one goes on through this cell iff extending beat opening here */
      const int cont_e = e > h - go, cont_f = f > h - go;
      if (state == 1 && !cont_e) state = 0;
      else if (state == 2 && !cont_f) state = 0;
      if (state == 0) state = best;
      if (state == 0) { push(rev, &m, tl + ql + 2, 0, 1); i--; j--; }
      else if (state == 1) { push(rev, &m, tl + ql + 2, 2, 1); i--; }
      else { push(rev, &m, tl + ql + 2, 1, 1); j--; }
    }
    if (i >= 0) push(rev, &m, tl + ql + 2, 2, (uint32_t)(i + 1));
    if (j >= 0) push(rev, &m, tl + ql + 2, 1, (uint32_t)(j + 1));
    for (int k = m - 1; k >= 0; k--) { if (n < cap) cigar[n] = rev[k]; n++; }
    free(rev);
  }
  free(H);
  return n;
#undef AT
}
