// Header name included by the reference (src/engine/llm_engine.h:24); nothing of it is used on the hot path.
#pragma once
#include "ppl/nn/runtime/runtime.h"
#include "ppl/nn/engines/engine.h"
