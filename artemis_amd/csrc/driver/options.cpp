// The option table of libartemis_hip.so (options.hpp) and its two C-ABI entry points.  Plain C++: linked into the product
// library and into the CPU test double alike.
#include <cctype>
#include <cstdlib>
#include <mutex>
#include <string>

#include "../../../include/artemis_hip.h"
#include "../options.hpp"

namespace artemis {
long g_options[OPT_COUNT];
bool g_options_ready = false;
namespace {
const char *const option_names[OPT_COUNT] = {
#define X(name) #name,
    ARTEMIS_OPTION_LIST(X)
#undef X
};
std::mutex g_options_mu;
int find_option(const char *name) {
  std::string n = name ? name : "";
  for (char &c : n) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  if (n.rfind("ARTEMIS_", 0) == 0) n = n.substr(8);
  for (int o = 0; o < OPT_COUNT; ++o)
    if (n == option_names[o]) return o;
  return -1;
}
} // namespace
void options_load() {
  std::lock_guard<std::mutex> lk(g_options_mu);
  if (g_options_ready) return;
  for (int o = 0; o < OPT_COUNT; ++o) {
    const std::string var = std::string("ARTEMIS_") + option_names[o];
    const char *e = std::getenv(var.c_str());
    long v = 0;
    if (e) {
      char *end = nullptr;
      v = std::strtol(e, &end, 10);
      if (end == e) v = 1; // (set, but not a number: a plain switch)
    }
    g_options[o] = v;
  }
  g_options_ready = true;
}
} // namespace artemis

extern "C" {
int artemis_hip_set_option(const char *name, long value) {
  if (!artemis::g_options_ready) artemis::options_load();
  const int o = artemis::find_option(name);
  if (o < 0) return ARTEMIS_HIP_EINVAL; // (unknown name)
  artemis::g_options[o] = value;
  return ARTEMIS_HIP_OK;
}
long artemis_hip_get_option(const char *name) {
  if (!artemis::g_options_ready) artemis::options_load();
  const int o = artemis::find_option(name);
  return o < 0 ? -1 : artemis::g_options[o];
}
}
