// ppl.common header name used by the reference (src/engine/llm_engine.h:28): EventCount lives in mpsc_queue.h here.
#pragma once
#include "mpsc_queue.h"
