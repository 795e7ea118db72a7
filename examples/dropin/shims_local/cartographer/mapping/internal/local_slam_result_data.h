// Stand-in for mapping/internal/local_slam_result_data.h: range_data_collator.cc includes it and
// uses nothing of it.
#ifndef DROPIN_SHIMS_LOCAL_LOCAL_SLAM_RESULT_DATA_H_
#define DROPIN_SHIMS_LOCAL_LOCAL_SLAM_RESULT_DATA_H_
#include <algorithm>
#include <map>
#include <set>
#include <string>
#endif  // DROPIN_SHIMS_LOCAL_LOCAL_SLAM_RESULT_DATA_H_
