// Stand-in for ppl.common's RetCode (the header is not part of the reference tree; semantics from its in-tree uses:
// src/backends/cuda/resource_manager.cc, post_processor.cc, src/engine/llm_engine.cc).  Values map 1:1 onto
// pplhip_status with the sign flipped (include/pplhip.h).
#pragma once
#include <stdint.h>

namespace ppl { namespace common {

typedef uint32_t RetCode;
enum {
    RC_SUCCESS = 0,
    RC_OTHER_ERROR = 1,
    RC_INVALID_VALUE = 2,
    RC_OUT_OF_MEMORY = 3,
    RC_DEVICE_RUNTIME_ERROR = 4,
    RC_DEVICE_MEMORY_ERROR = 5,
    RC_NOT_FOUND = 6,
    RC_UNSUPPORTED = 7,
};

inline const char* GetRetCodeStr(RetCode rc) {
    static const char* names[] = {"success", "other error", "invalid value", "out of memory", "device runtime error",
                                  "device memory error", "not found", "unsupported"};
    return rc < 8 ? names[rc] : "unknown";
}

inline RetCode FromPplHipStatus(int st) { return st >= 0 ? RC_SUCCESS : (RetCode)(-st); }

typedef uint16_t float16_t;  // storage only (sizeof == 2), as used by resource_manager.cc:387

}}  // namespace ppl::common
