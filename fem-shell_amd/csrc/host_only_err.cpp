// host_only_err.cpp -- error plumbing of the sanitizer builds of the host-side code (`make san`): plan.cpp, reorder.cpp,
// amg_setup.cpp and plan_api.cpp compiled without the HIP half of the library, which owns these three functions in the
// product (api.cpp).  Not part of libfemshell.so.
#include <string>

#include "femshell.h"

namespace femshell {

namespace {
thread_local std::string g_err;
}

int set_err(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

const std::string &last_err() { return g_err; }

} // namespace femshell

extern "C" const char *femshell_last_error(void) { return femshell::last_err().c_str(); }
