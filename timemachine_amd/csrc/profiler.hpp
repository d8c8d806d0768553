// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: zero overhead on the MD path.
#pragma once
#include "common.hpp"

#include <map>
#include <utility>

namespace tmamd {

class Profiler {
public:
    static Profiler &get();
    void set_enabled(bool e) { enabled_ = e; }
    bool enabled() const { return enabled_; }
    // returns an index to pass to end(); -1 when disabled
    int begin(const char *name, hipStream_t stream);
    void end(const char *name, int idx, hipStream_t stream);
    void read(const char *name, double *total_ms, long long *launches); // synchronises the device
    void reset();

private:
    bool enabled_ = false;
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> events_;
};

} // namespace tmamd
