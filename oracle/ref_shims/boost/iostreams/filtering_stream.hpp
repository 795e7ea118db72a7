// Placeholders for the Boost types common/port.h names in two inline gzip helpers that the
// files built by `make ref` never call.
#ifndef ORACLE_REF_SHIMS_BOOST_IOSTREAMS_HPP_
#define ORACLE_REF_SHIMS_BOOST_IOSTREAMS_HPP_
#include <cstdlib>
#include <ios>
#include <string>
namespace boost {
namespace iostreams {
namespace zlib { const int best_speed = 1; }
struct gzip_compressor { explicit gzip_compressor(int) {} };
struct gzip_decompressor {};
struct back_inserter_device {};
inline back_inserter_device back_inserter(std::string&) { return back_inserter_device(); }
struct filtering_ostream {
  template <typename T>
  void push(const T&) {}
};
inline void write(filtering_ostream&, const char*, std::streamsize) { std::abort(); }
}  // namespace iostreams
}  // namespace boost
#endif  // ORACLE_REF_SHIMS_BOOST_IOSTREAMS_HPP_
