// Development switches.  The PRODUCT build of libdq_hip.so reads no environment variable: what a caller may tune goes through
// dq_set_option (dq_options.h).  The A-B switches that select a code path the current one replaced exist only in a library built with
// -DDQ_DEV_SWITCHES (`make dev` -> build/dev/libdq_hip_dev.so; tests/test_tiny_levels.py and tools/ab_env.sh load it through DQ_HIP_LIB);
// in the product build DQ_DEV_FLAG(...) is the constant false and the code it guards -- together with the kernels only it launches --
// is compiled out where it stands inside `#ifdef DQ_DEV_SWITCHES`.
#pragma once
#ifdef DQ_DEV_SWITCHES
#include <cstdlib>
namespace dq {
inline bool dev_env_is(const char* name, char c) { const char* e = std::getenv(name); return e && e[0] == c; }
}  // namespace dq
// true when the environment variable NAME starts with the character C (read once per call site)
#define DQ_DEV_FLAG(NAME, C) ([] { static const bool v = ::dq::dev_env_is(NAME, C); return v; }())
#else
#define DQ_DEV_FLAG(NAME, C) false
#endif
