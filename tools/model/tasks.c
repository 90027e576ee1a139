/* prints the fast decoder's task structure of a code as JSON (tools/valu_issue_model.py): check-node items per (degree,
 * ext), bit-node items per loop bound, per pass */
#include <stdio.h>
#include <stdlib.h>
#include "ldpc_graph.h"
int main(int argc, char **argv)
{
  static ldpc_code_desc_t d;
  const int BG = argc > 1 ? atoi(argv[1]) : 1, Z = argc > 2 ? atoi(argv[2]) : 384, R = argc > 3 ? atoi(argv[3]) : 13;
  if (ldpc_build_code_desc(BG, Z, R, &d) != 0 || !d.f_ok) return 1;
  printf("{\"BG\": %d, \"Z\": %d, \"R\": %d, \"nedges\": %d, \"threads\": %d, \"cn_tasks\": [", BG, Z, R, d.nedges, d.f_n_threads);
  for (int t = 0; t < d.f_n_cn_tasks; t++) {
    const int dbl = (d.f_cn_task[t][0] & 0x100) != 0, deg = d.f_cn_task[t][0] & 0xff, per = dbl ? 128 : 64; /* double task: 128 items, two per thread */
    int first = d.f_cn_task[t][2], gend = d.f_cn_task[t][4], items = gend - first < per ? gend - first : per;
    printf("%s{\"deg\": %d, \"ext\": %d, \"items\": %d, \"pair\": %d, \"double\": %d}", t ? ", " : "", deg, d.f_cn_task[t][1], items,
           deg == 19 && d.f_pair19, dbl);
  }
  printf("], \"bn_tasks\": [");
  for (int t = 0; t < d.f_n_bn_tasks; t++) {
    int first = d.f_bn_task[t][0], end = d.f_bn_task[t][1], items = end - first < 64 ? end - first : 64;
    printf("%s{\"maxdeg\": %d, \"items\": %d}", t ? ", " : "", d.f_bn_task[t][2], items);
  }
  printf("]}\n");
  return 0;
}
