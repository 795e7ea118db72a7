// Stand-in: the matcher sources only mention the type in their Create*Options functions,
// which `make ref` compiles but nothing calls.
#ifndef ORACLE_REF_SHIMS_LUA_PARAMETER_DICTIONARY_H_
#define ORACLE_REF_SHIMS_LUA_PARAMETER_DICTIONARY_H_
#include <cstdlib>
#include <memory>
#include <string>
namespace cartographer {
namespace common {
class LuaParameterDictionary {
 public:
  double GetDouble(const std::string&) { std::abort(); }
  int GetInt(const std::string&) { std::abort(); }
  int GetNonNegativeInt(const std::string&) { std::abort(); }
  bool GetBool(const std::string&) { std::abort(); }
  bool HasKey(const std::string&) { std::abort(); }
  std::string GetString(const std::string&) { std::abort(); }
  std::unique_ptr<LuaParameterDictionary> GetDictionary(const std::string&) { std::abort(); }
};
}  // namespace common
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_LUA_PARAMETER_DICTIONARY_H_
