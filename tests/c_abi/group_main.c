/* Plain-C consumer of include/bramble_amd.h (no ctypes, no C++): builds the index of the reference's
 * short_read_projection fixture (bramble-rs/tests/short_read_projection.rs:35-91; tests/golden K9 / K10), projects
 * each read through br_project_group -- the C form of project_group_with (bramble-rs/src/api.rs:285-290) -- and prints
 * one line per projected record for the Python test to compare with the known answers. */
#include <stdio.h>
#include <string.h>

#include "bramble_amd.h"

static int die(const char *what, int rc) {
  fprintf(stderr, "%s: %s (%d)\n", what, br_strerror(rc), rc);
  return 1;
}

int main(void) {
  /* exons are 1-based half-open: GTF end + 1 */
  const br_exon e1[] = {{101, 301}};
  const br_exon e2[] = {{500, 601}, {800, 901}};
  const br_transcript tx[] = {{"tx1", "chr1", '+', e1, 1}, {"tx2", "chr1", '+', e2, 2}};
  const char *refs[] = {"chr1"};
  br_index *ix = NULL;
  br_ctx *ctx = NULL;
  int rc = br_index_build(tx, 2, refs, 1, NULL, 0, 0, &ix);
  if (rc) return die("br_index_build", rc);
  if ((rc = br_ctx_new(ix, &ctx))) return die("br_ctx_new", rc);
  br_config cfg;
  br_config_short_read(&cfg);

  const uint32_t c9[] = {(100u << 4) | 0u};
  const uint32_t c10[] = {(50u << 4) | 0u, (199u << 4) | 3u, (50u << 4) | 0u};
  br_alignment reads[2];
  memset(reads, 0, sizeof(reads));
  reads[0].query_name = "unspliced"; reads[0].ref_id = 0; reads[0].ref_start = 151; reads[0].cigar = c9; reads[0].n_cigar = 1;
  reads[0].mate_ref_id = -1; reads[0].read_len = 100;
  reads[1].query_name = "spliced"; reads[1].ref_id = 0; reads[1].ref_start = 551; reads[1].cigar = c10; reads[1].n_cigar = 3;
  reads[1].mate_ref_id = -1; reads[1].read_len = 100;
  for (int k = 0; k < 2; k++) {
    const br_projected *out = NULL;
    size_t n = 0;
    if ((rc = br_project_group(ctx, &cfg, &reads[k], 1, &out, &n))) return die("br_project_group", rc);
    for (size_t r = 0; r < n; r++) {
      printf("%s %s %u %u %u %u %c nh=%u hi=%u mapq=%u primary=%u cigar=", reads[k].query_name,
             br_index_transcript_name(ix, out[r].transcript_id), out[r].transcript_start, out[r].transcript_end,
             out[r].aligned_len, out[r].query_aligned_len, out[r].transcript_strand, out[r].nh, out[r].hi, out[r].mapq,
             (unsigned)out[r].is_primary);
      for (uint32_t j = 0; j < out[r].n_cigar; j++) printf("%u%c", out[r].cigar[j] >> 4, "MIDNSHP=XB"[out[r].cigar[j] & 15u]);
      printf("\n");
    }
  }
  /* two different query names in one call are refused */
  {
    const br_projected *out = NULL;
    size_t n = 0;
    rc = br_project_group(ctx, &cfg, reads, 2, &out, &n);
    printf("mixed-names rc=%d\n", rc);
  }
  br_ctx_free(ctx);
  br_index_free(ix);
  return 0;
}
