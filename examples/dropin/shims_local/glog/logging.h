// Extends oracle/ref_shims' glog stand-in by the one macro motion_filter.cc adds.
#ifndef DROPIN_SHIMS_LOCAL_GLOG_LOGGING_H_
#define DROPIN_SHIMS_LOCAL_GLOG_LOGGING_H_
#include_next "glog/logging.h"
#define LOG_IF_EVERY_N(severity, condition, n) ::ref_shims::NullStream()
#endif  // DROPIN_SHIMS_LOCAL_GLOG_LOGGING_H_
