// Stand-in for ppl.common's types.h: the data type / data format enums and helpers the reference names
// (src/utils/utils.cc:96-163).
#pragma once
#include <stdint.h>

#include <string>

namespace ppl { namespace common {

typedef uint32_t datatype_t;
enum {
    DATATYPE_UNKNOWN = 0, DATATYPE_UINT8, DATATYPE_UINT16, DATATYPE_UINT32, DATATYPE_UINT64, DATATYPE_FLOAT16, DATATYPE_FLOAT32,
    DATATYPE_FLOAT64, DATATYPE_BFLOAT16, DATATYPE_INT4B, DATATYPE_INT8, DATATYPE_INT16, DATATYPE_INT32, DATATYPE_INT64, DATATYPE_BOOL,
};
typedef uint32_t dataformat_t;
enum { DATAFORMAT_UNKNOWN = 0, DATAFORMAT_NDARRAY = 1 };

inline const char* GetDataTypeStr(datatype_t dt) {
    static const char* names[] = {"UNKNOWN", "UINT8", "UINT16", "UINT32", "UINT64", "FLOAT16", "FLOAT32", "FLOAT64",
                                  "BFLOAT16", "INT4B", "INT8", "INT16", "INT32", "INT64", "BOOL"};
    return dt <= DATATYPE_BOOL ? names[dt] : "UNKNOWN";
}
inline uint32_t GetSizeOfDataType(datatype_t dt) {
    switch (dt) {
        case DATATYPE_UINT8: case DATATYPE_INT8: case DATATYPE_BOOL: return 1;
        case DATATYPE_UINT16: case DATATYPE_INT16: case DATATYPE_FLOAT16: case DATATYPE_BFLOAT16: return 2;
        case DATATYPE_UINT32: case DATATYPE_INT32: case DATATYPE_FLOAT32: return 4;
        case DATATYPE_UINT64: case DATATYPE_INT64: case DATATYPE_FLOAT64: return 8;
        default: return 0;
    }
}
template <typename T>
inline std::string ToString(const T& v) { return std::to_string(v); }

}}  // namespace ppl::common
