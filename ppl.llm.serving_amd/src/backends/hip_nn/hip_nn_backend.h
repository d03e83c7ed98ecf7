// src/backends/hip in the form the REFERENCE's engine consumes it: ppl::nn::Runtime / Tensor / Engine /
// DeviceContext objects over libpplhip's C ABI, and the reference's own PostProcessor interface.
//
// The reference's LLMEngine binds runtime inputs 0..10 and output 0 BY INDEX, reshapes them and calls CopyFromHostAsync per
// tensor and step, then Runtime::Run() (src/engine/llm_engine.h:124-147, src/engine/llm_engine.cc:29-116).  Here a tensor's
// CopyFromHostAsync only records the caller's host pointer (scalars are copied at once); Run() hands the collected step to
// pplhip_set_inputs (ONE pinned staging copy + one H2D) and pplhip_run.  The page table counts as changed exactly when the
// engine copied `cache_indices` since the last Run (llm_engine.cc:67-71 copies it only when the batch changed).
//
// This file is compiled against the reference's own src/common/{resource,config,post_processor}.h -- by `make ref` in the
// build container, together with the reference's unmodified llm_engine.cc / llm_generator.cc (tests/host/ref_backend_driver.cc).
// The repo's own engine uses the leaner one-record form of the same step (src/backends/hip, src/common/resource.h).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "common/config.h"
#include "common/resource.h"
#include "ppl/common/threadpool.h"

struct pplhip_ctx;

namespace ppl { namespace llm { namespace hip_nn {

// what params.json / the command line carry beyond the reference's ModelConfig (DESIGN.md "weight container")
struct ExtraConfig {
    int32_t weight_quant_bit = 0, weight_quant_group = 128, max_position = 4096;
    float norm_eps = 1e-5f, rope_theta = 10000.0f;
    bool synthetic_weights = false;
    uint64_t synthetic_seed = 1234;
    uint64_t kv_cache_max_tokens = 0;  // 0: max_tokens_scale x free memory (resource_manager.cc:330-341)
    int32_t max_tokens_per_step = 8192;
};

class Backend final {
public:
    Backend();
    ~Backend();
    ppl::common::RetCode Init(const ModelConfig&, const ResourceConfig&, const ExtraConfig&);
    void FillResource(Resource*);   // raw, non-owning pointers: destroy generator / engine first (offline_inference.cc:414)

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}}}  // namespace ppl::llm::hip_nn
