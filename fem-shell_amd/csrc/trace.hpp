// trace.hpp -- roctx ranges around the phases of the library (tracing): they show up as named spans in
// `rocprofv3 --marker-trace` / rocprof timelines and cost two library calls otherwise.
#pragma once

#include <roctracer/roctx.h>

namespace femshell {

struct TraceRange {
    explicit TraceRange(const char *name) { roctxRangePushA(name); }
    ~TraceRange() { roctxRangePop(); }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

} // namespace femshell
