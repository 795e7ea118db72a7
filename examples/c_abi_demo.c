/* Plain-C use of the drop-in boundary: proves the header is C-clean and that
 * every entry point links.  Run on a GPU box it performs one real match; with
 * no device it reports CMX_DEVICE_ERROR (there is no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>

#include "cartographer_mi355x.h"

int main(void) {
  enum { NX = 64, NY = 48, N = 5 };
  uint16_t* cells = (uint16_t*)calloc(NX * NY, sizeof(uint16_t));
  const cmx_grid2d_limits limits = {0.05, 1.2, 1.6, NX, NY, 0.1f, 0.9f};
  const cmx_fast2d_options options = {1.0, 0.3, 4};
  const float cloud[3 * N] = {0.5f, 0.f, 0.f, 0.f, 0.5f, 0.f, -0.5f, 0.f, 0.f, 0.f, -0.5f, 0.f,
                              0.25f, 0.25f, 0.f};
  int i;
  for (i = 0; i < NX * NY; i += 7) cells[i] = 20000;
  printf("%s, %d device(s)\n", cmx_version(), cmx_device_count());
  cmx_fast2d* matcher = NULL;
  cmx_status st = cmx_fast2d_create(&options, &limits, cells, 0, &matcher);
  if (st != CMX_OK) {
    printf("create: %s (%s)\n", cmx_status_string(st), cmx_last_error());
    free(cells);
    return st == CMX_DEVICE_ERROR ? 0 : 1;   /* expected without a GPU */
  }
  const cmx_pose2d init = {0.3, 0.4, 0.1};
  int32_t found = 0;
  float score = 0.f;
  cmx_pose2d pose;
  cmx_match_stats stats;
  st = cmx_fast2d_match(matcher, &init, cloud, N, 0.05f, &found, &score, &pose, &stats);
  printf("match: %s found=%d score=%f pose=(%f, %f, %f) candidates=%lld\n", cmx_status_string(st),
         found, score, pose.x, pose.y, pose.theta, (long long)stats.candidates_scored);
  cmx_fast2d_destroy(matcher);
  free(cells);
  return st == CMX_OK ? 0 : 1;
}
