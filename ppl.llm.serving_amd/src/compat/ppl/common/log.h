// Stand-in for ppl.common's stream logger: LOG(DEBUG|INFO|WARNING|ERROR) << ... ; level from PPL_LOG_LEVEL
// (0 debug .. 3 error, default 2 = warnings and errors).
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>

namespace ppl { namespace common {

enum { LOG_LEVEL_DEBUG = 0, LOG_LEVEL_INFO = 1, LOG_LEVEL_WARNING = 2, LOG_LEVEL_ERROR = 3 };

inline int GetLogLevel() {
    static int lvl = [] { const char* e = getenv("PPL_LOG_LEVEL"); return e ? atoi(e) : (int)LOG_LEVEL_WARNING; }();
    return lvl;
}

class LogMessage final {
public:
    LogMessage(int level, const char* file, int line) : on_(level >= GetLogLevel()) {
        static const char* tag[] = {"DEBUG", "INFO", "WARNING", "ERROR"};
        if (on_) ss_ << "[" << tag[level] << "][" << file << ":" << line << "] ";
    }
    ~LogMessage() {
        if (on_) {
            ss_ << "\n";
            std::cerr << ss_.str();
        }
    }
    template <typename T>
    LogMessage& operator<<(const T& v) {
        if (on_) ss_ << v;
        return *this;
    }

private:
    bool on_;
    std::ostringstream ss_;
};

}}  // namespace ppl::common

#define LOG(level) ::ppl::common::LogMessage(::ppl::common::LOG_LEVEL_##level, __FILE__, __LINE__)
