// ppl.common header name used by the reference (src/engine/llm_engine.h:29): PageManager lives in allocators.h here.
#pragma once
#include "allocators.h"
