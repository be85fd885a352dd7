// capi.cpp -- error plumbing for the C ABI (no exceptions cross the boundary).
#include <string>

#include "../../include/hugectr_amd.h"

namespace hctr {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace hctr

extern "C" {
const char* hctr_last_error(void) { return hctr::g_last_error.c_str(); }
int hctr_version(void) { return 100; }
}
