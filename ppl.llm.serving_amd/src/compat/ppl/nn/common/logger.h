// ppl.nn header name used by the reference (src/tokenizer/tokenizer_impl.h:22): the logger is ppl.common's.
#pragma once
#include "ppl/common/log.h"
#include "ppl/common/retcode.h"
