// capi.hip — library-level entry points of libparl_hip.so (version, errors).
#include "common.hpp"
#include "srchash.gen.hpp"

namespace parlhip {
thread_local int g_last_hip_error = 0;
}  // namespace parlhip

PARLHIP_EXPORT int parlhip_version(void) { return 100; /* 0.1.0 */ }

// sha256[:16] over the sources this library was built from (csrc/srchash.py)
PARLHIP_EXPORT const char* parlhip_source_hash(void) { return PARLHIP_SOURCE_HASH; }

PARLHIP_EXPORT const char* parlhip_strerror(int code) {
  switch (code) {
    case PARLHIP_OK: return "ok";
    case PARLHIP_EINVAL: return "invalid argument";
    case PARLHIP_ELAUNCH: return "HIP runtime / kernel launch error";
    case PARLHIP_ENOSUP: return "not supported";
    case PARLHIP_ENOMEM: return "workspace too small";
    default: return "unknown parlhip error";
  }
}

PARLHIP_EXPORT int parlhip_last_hip_error(void) { return parlhip::g_last_hip_error; }
