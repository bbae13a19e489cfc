/* br_project_group host to host, from plain C (no Python marshalling): one read pair per call against a small
 * annotation, and the same pairs 64 to a call through br_project_groups.
 *   gcc -O2 -std=c99 -I include profiles/group_latency.c -o /tmp/group_latency -L bramble_amd -lbramble_amd -Wl,-rpath,$PWD/bramble_amd && /tmp/group_latency */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bramble_amd.h"

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

int main(void) {
  /* 200 genes of three exons on either strand of one reference */
  enum { NG = 200 };
  static br_exon ex[NG][3];
  static br_transcript tx[NG];
  static char names[NG][16];
  for (int g = 0; g < NG; g++) {
    const uint32_t s = 1000u + 5000u * (uint32_t)g;
    ex[g][0].start = s; ex[g][0].end = s + 300; ex[g][1].start = s + 1000; ex[g][1].end = s + 1200; ex[g][2].start = s + 2000; ex[g][2].end = s + 2400;
    snprintf(names[g], sizeof names[g], "tx%d", g);
    tx[g].id = names[g]; tx[g].seqname = "chr1"; tx[g].strand = (g & 1) ? '-' : '+'; tx[g].exons = ex[g]; tx[g].n_exons = 3;
  }
  const char *refs[1] = {"chr1"};
  br_index *ix = NULL; br_ctx *ctx = NULL;
  int rc = br_index_build(tx, NG, refs, 1, NULL, 0, 0, &ix);
  if (rc) { fprintf(stderr, "br_index_build: %s\n", br_strerror(rc)); return 1; }
  if ((rc = br_ctx_new(ix, &ctx))) { fprintf(stderr, "br_ctx_new: %s\n", br_strerror(rc)); return 1; }
  br_config cfg; br_config_short_read(&cfg);
  enum { NP = 64 };
  static br_alignment al[2 * NP];
  static uint32_t cig[2 * NP][3];
  static char qn[NP][16];
  memset(al, 0, sizeof al);
  for (int p = 0; p < NP; p++) {
    const uint32_t s = 1000u + 5000u * (uint32_t)(p * 3 % NG);
    snprintf(qn[p], sizeof qn[p], "r%d", p);
    /* mate 1 inside exon 0, mate 2 spliced from exon 0 into exon 1 */
    br_alignment *a = &al[2 * p], *b = &al[2 * p + 1];
    cig[2 * p][0] = 100u << 4;
    cig[2 * p + 1][0] = 50u << 4; cig[2 * p + 1][1] = (700u << 4) | 3u; cig[2 * p + 1][2] = 50u << 4;
    a->query_name = qn[p]; a->ref_id = 0; a->ref_start = s + 20; a->cigar = cig[2 * p]; a->n_cigar = 1; a->read_len = 100;
    a->is_paired = 1; a->is_first_in_pair = 1; a->mate_ref_id = 0; a->mate_ref_start = s + 250;
    b->query_name = qn[p]; b->ref_id = 0; b->ref_start = s + 250; b->cigar = cig[2 * p + 1]; b->n_cigar = 3; b->read_len = 100;
    b->is_paired = 1; b->is_reverse = 1; b->mate_ref_id = 0; b->mate_ref_start = s + 20;
  }
  const br_projected *out = NULL; size_t n = 0, rows = 0;
  for (int k = 0; k < 50; k++) if ((rc = br_project_group(ctx, &cfg, &al[2 * (k % NP)], 2, &out, &n))) { fprintf(stderr, "br_project_group: %s\n", br_strerror(rc)); return 1; }
  const int K = 2000;
  double t0 = now_us();
  for (int k = 0; k < K; k++) { br_project_group(ctx, &cfg, &al[2 * (k % NP)], 2, &out, &n); rows += n; }
  double per_group = (now_us() - t0) / K;
  for (int k = 0; k < 20; k++) br_project_groups(ctx, &cfg, al, 2 * NP, &out, &n);
  t0 = now_us();
  for (int k = 0; k < 500; k++) br_project_groups(ctx, &cfg, al, 2 * NP, &out, &n);
  double per_64 = (now_us() - t0) / 500;
  printf("{\"br_project_group_us_per_call_one_pair\": %.1f, \"records_per_call\": %.2f, \"br_project_groups_us_per_call_64_pairs\": %.1f, \"records_per_64\": %zu}\n",
         per_group, (double)rows / K, per_64, n);
  br_ctx_free(ctx); br_index_free(ix);
  return 0;
}
