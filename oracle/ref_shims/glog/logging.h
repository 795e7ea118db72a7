// CHECK* / DCHECK* that abort with a message, LOG(...) that swallows its stream.
#ifndef ORACLE_REF_SHIMS_GLOG_LOGGING_H_
#define ORACLE_REF_SHIMS_GLOG_LOGGING_H_

#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

namespace ref_shims {
struct NullStream {
  template <typename T>
  NullStream& operator<<(const T&) { return *this; }
  NullStream& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
[[noreturn]] inline void CheckFailed(const char* file, int line, const char* expr) {
  std::fprintf(stderr, "%s:%d: Check failed: %s\n", file, line, expr);
  std::abort();
}
inline NullStream& Null() {
  static NullStream stream;
  return stream;
}
struct Voidify { void operator&(NullStream&) {} };
}  // namespace ref_shims

#define REF_SHIMS_CHECK(cond)                                                          \
  (cond) ? (void)0                                                                     \
         : ::ref_shims::Voidify() &                                                    \
               (::ref_shims::CheckFailed(__FILE__, __LINE__, #cond), ::ref_shims::Null())
#define CHECK(cond) REF_SHIMS_CHECK(cond)
#define CHECK_NOTNULL(p) REF_SHIMS_CHECK((p) != nullptr)
#define CHECK_EQ(a, b) REF_SHIMS_CHECK((a) == (b))
#define CHECK_NE(a, b) REF_SHIMS_CHECK((a) != (b))
#define CHECK_LE(a, b) REF_SHIMS_CHECK((a) <= (b))
#define CHECK_LT(a, b) REF_SHIMS_CHECK((a) < (b))
#define CHECK_GE(a, b) REF_SHIMS_CHECK((a) >= (b))
#define CHECK_GT(a, b) REF_SHIMS_CHECK((a) > (b))
#ifdef NDEBUG   // glog compiles DCHECKs out of release builds
#define REF_SHIMS_DCHECK(cond) \
  true ? (void)0 : ::ref_shims::Voidify() & (::ref_shims::Null())
#else
#define REF_SHIMS_DCHECK(cond) REF_SHIMS_CHECK(cond)
#endif
#define DCHECK(cond) REF_SHIMS_DCHECK(cond)
#define DCHECK_EQ(a, b) REF_SHIMS_DCHECK((a) == (b))
#define DCHECK_NE(a, b) REF_SHIMS_DCHECK((a) != (b))
#define DCHECK_LE(a, b) REF_SHIMS_DCHECK((a) <= (b))
#define DCHECK_LT(a, b) REF_SHIMS_DCHECK((a) < (b))
#define DCHECK_GE(a, b) REF_SHIMS_DCHECK((a) >= (b))
#define DCHECK_GT(a, b) REF_SHIMS_DCHECK((a) > (b))
#define LOG(severity) ::ref_shims::NullStream()
#define LOG_IF(severity, condition) ::ref_shims::NullStream()

#endif  // ORACLE_REF_SHIMS_GLOG_LOGGING_H_
