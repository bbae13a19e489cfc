// Types shared by the -S rescue kernels (rescue_kernels.inc: planning / applying; ksw_kernels.hip: the DP itself).
#pragma once
#include <stdint.h>

namespace br {

struct KswProb { uint32_t qlen, tlen; uint32_t side; uint32_t t_has_n; uint64_t seq_off; };   // q codes, then t codes; t_has_n: an N among the target codes
// where a problem's sequences come from (written by k_project_fa<1>, read by k_fa_fill): the alignment, its transcript's
// first tx_ex row, the first neighbour exon of the target window
struct FaSrc { uint32_t aln, e_base, i_from, pad; };
struct KswRes { int32_t ok, score, refc; uint32_t n_ops; };                              // ops at clip_ops[seq_off + p]

// Streamed DP (k_ksw_dp): problems are binned by target length into four array shapes; a bin's problems sit in a
// compact descriptor array that the lanes of a group read with plain strided indices (no pointer chase).
struct KswDesc { uint32_t qt; uint32_t prob; uint64_t seq_off; };   // qlen | tlen << 16, problem index, first query code
// What the DP leaves per problem for the traceback kernel.
//   flags: bit 0 = handled by k_ksw_dp, bit 1 = the DP reached the last cell (score valid, no z-drop), bits 8.. = bin
//   tape : byte offset (in the direction tape) of the group's first lane in the row of the problem's anti-diagonal 0;
//   anti-diagonals >= split continue at tape2 (the wave's next tape chunk)
struct KswDp { int32_t max, max_t, max_q; uint32_t flags; uint64_t tape, tape2; uint32_t split, pad; };

#define KSW_N_BINS 4
// lanes per group / target columns per lane of every bin (columns = product); a lane stores 8 or 16 tape bytes per step
#define KSW_BIN_G(b) ((b) == 0 ? 8 : (b) == 1 ? 8 : (b) == 2 ? 16 : 16)
#define KSW_BIN_K(b) ((b) == 0 ? 8 : (b) == 3 ? 24 : 16)
#define KSW_BIN_W(b) (KSW_BIN_G(b) * KSW_BIN_K(b))      // 64, 128, 256, 384 target columns
#define KSW_BIN_LANEBYTES(b) (KSW_BIN_K(b) > 16 ? 16 : 8)   // tape bytes of a lane and step (a nibble per column, rounded up)
#define KSW_BIN_ROWBYTES(b) (64 * KSW_BIN_LANEBYTES(b))     // a tape row = one step of a wave
#define KSW_CHUNK_ROWS 1024                              // waves take tape in chunks; a problem spans at most two
#define KSW_MAX_SPAN KSW_CHUNK_ROWS                      // qlen + tlen of a problem the arrays take

}  // namespace br
