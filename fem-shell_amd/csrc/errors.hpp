// errors.hpp -- error reporting shared by every translation unit of libfemshell (and by the host-only sanitizer builds).
#pragma once

#include <string>

#include "femshell.h"

namespace femshell {

// records the message femshell_last_error() returns on the calling thread and passes `code` through
int set_err(int code, const std::string &msg);
const std::string &last_err();

} // namespace femshell
