#include "resource_manager.h"

#include <string.h>

#include "../../utils/utils.h"
#include "ppl/common/log.h"

using namespace ppl::common;

namespace ppl { namespace llm { namespace hip {

/* ---------------------------------------------------------------------------------------------- HipRuntime */

RetCode HipRuntime::SetInputs(const StepInputs& in) {
    pplhip_step st;
    memset(&st, 0, sizeof(st));
    st.batch = in.batch;
    st.num_tokens = in.num_tokens;
    st.decoding_batches = in.decoding_batches;
    st.max_seq_len = in.max_seq_len;
    st.max_kv_len = in.max_kv_len;
    st.max_pages = in.max_pages;
    st.token_inputs = in.token_inputs;
    st.seq_starts = in.seq_starts;
    st.kv_starts = in.kv_starts;
    st.start_pos = in.start_pos;
    st.cache_indices = in.cache_indices;
    st.req_list_changed = in.req_list_changed ? 1 : 0;
    return FromPplHipStatus(pplhip_set_inputs(ctx_, rank_, &st));
}

RetCode HipRuntime::Run(bool is_prefix_cache_hit) {
    return FromPplHipStatus(pplhip_run(ctx_, rank_, is_prefix_cache_hit ? 1 : 0));
}

float* HipRuntime::GetLogits(int64_t* batch_stride) {
    float* p = nullptr;
    pplhip_logits(ctx_, rank_, &p, batch_stride);
    return p;
}

/* ---------------------------------------------------------------------------------------------- HipPostProcessor */

RetCode HipPostProcessor::InitPostProcessorMem(int, int, bool) {
    return RC_SUCCESS;  // the sampler's device memory is part of the context (sized by pplhip_opts)
}

RetCode HipPostProcessor::SampleTopKTopP(const float* logits_device, const float* temperatures_host, const int32_t* top_k_host,
                                         const float* top_p_host, int32_t batch, int32_t vocab_size, int32_t batch_stride,
                                         int32_t default_top_k, float default_top_p, bool req_list_changed,
                                         int32_t* output_host, float* logprob_host, bool enable_penalty) {
    pplhip_sample_args a;
    memset(&a, 0, sizeof(a));
    a.temperatures = temperatures_host;
    a.top_k = top_k_host;
    a.top_p = top_p_host;
    a.batch = batch;
    a.vocab_size = vocab_size;
    a.batch_stride = batch_stride;
    a.default_top_k = default_top_k;
    a.default_top_p = default_top_p;
    a.req_list_changed = req_list_changed ? 1 : 0;
    a.enable_penalty = enable_penalty ? 1 : 0;
    const int st = pplhip_sample(ctx_, logits_device, &a, output_host, logprob_host);
    if (st) LOG(ERROR) << "sampling failed: " << pplhip_last_error(ctx_, 0);
    return FromPplHipStatus(st);
}

RetCode HipPostProcessor::ApplyPenalty(const float* temperatures_host, const float* repetition_penalties_host,
                                       const float* presence_penalties_host, const float* frequency_penalties_host,
                                       const int64_t* batch_slots_host, const int64_t*, const int64_t*, const int64_t*,
                                       int32_t batch, int32_t vocab_size, bool req_list_changed, float* logits) {
    // token_inputs / seqstarts / start_pos: the library uses the step's device-resident copies itself
    pplhip_penalty_args a;
    memset(&a, 0, sizeof(a));
    a.temperatures = temperatures_host;
    a.repetition_penalties = repetition_penalties_host;
    a.presence_penalties = presence_penalties_host;
    a.frequency_penalties = frequency_penalties_host;
    a.batch_slots = batch_slots_host;
    a.batch = batch;
    a.vocab_size = vocab_size;
    a.req_list_changed = req_list_changed ? 1 : 0;
    const int st = pplhip_penalty(ctx_, logits, &a);
    if (st) LOG(ERROR) << "apply_penalty failed: " << pplhip_last_error(ctx_, 0);
    return FromPplHipStatus(st);
}

/* ---------------------------------------------------------------------------------------------- HipResourceManager */

HipResourceManager::~HipResourceManager() {
    runtimes.clear();
    post_processor.reset();
    if (ctx) pplhip_destroy(ctx);
}

// per-rank initialisation, run on the rank's worker thread (reference InitTask, resource_manager.cc:213-371)
static RetCode InitRank(uint32_t id, pplhip_ctx* ctx, const ResourceConfig& rc, float max_tokens_scale, Barrier* barrier,
                        HipResourceManager* mgr) {
    int st;
    if (rc.synthetic_weights) {
        st = pplhip_rank_init_synthetic(ctx, (int)id, rc.synthetic_seed);
        if (!st && rc.synthetic_decisive_head) st = pplhip_rank_tie_output(ctx, (int)id, rc.synthetic_decisive_head, rc.synthetic_seed, 4.0f);   // margins 0.35-0.68 of the logit scale at 32 layers (profiles/r05_tie_probe.log)
    } else {
        const std::string slice = rc.model_dir + "/model_slice_" + std::to_string(id);
        LOG(INFO) << "model_slice_" << id << ": " << slice;
        st = pplhip_rank_load(ctx, (int)id, slice.c_str());
    }
    if (st) {
        LOG(ERROR) << "load weights of rank [" << id << "] failed: " << pplhip_last_error(ctx, (int)id);
        if (!rc.synthetic_weights) {
            // a ppl.pmx export (model_slice_<r>/model.onnx, the reference's --model-format onnx / pmx) is converted once, offline
            const std::string onnx = rc.model_dir + "/model_slice_" + std::to_string(id) + "/model.onnx";
            if (FILE* f = fopen(onnx.c_str(), "rb")) {
                fclose(f);
                LOG(ERROR) << rc.model_dir << " holds a ppl.pmx export (model.onnx): convert it with "
                           << "`python ppl.llm.serving_amd/tools/import_pmx_onnx.py --model-dir " << rc.model_dir
                           << " --out <dir> --quant {none,w8a16,w4a16}` and pass the result as --model-dir";
            }
        }
        barrier->Wait();
        return FromPplHipStatus(st);
    }
    if (id == 0) {  // rank 0 sizes the slab for everybody (resource_manager.cc:329-342)
        uint64_t tokens = rc.kv_cache_max_tokens_override;
        if (tokens == 0 && pplhip_kv_capacity(ctx, max_tokens_scale, &tokens) != 0) tokens = 0;
        mgr->kv_cache_max_tokens = tokens;
        LOG(INFO) << "max_tokens: " << tokens;
    }
    barrier->Wait();
    if (mgr->kv_cache_max_tokens == 0) return RC_OUT_OF_MEMORY;
    st = pplhip_kv_alloc(ctx, (int)id, mgr->kv_cache_max_tokens);
    if (st) {
        LOG(ERROR) << "alloc kv cache on rank [" << id << "] failed: " << pplhip_last_error(ctx, (int)id);
        return FromPplHipStatus(st);
    }
    ResourceItem item;
    pplhip_kv_ptrs(ctx, (int)id, &item.kv_cache_mem, &item.kv_scale_mem);
    mgr->runtimes[id].reset(new HipRuntime(ctx, (int)id));
    item.runtime = mgr->runtimes[id].get();
    mgr->items[id] = item;
    return RC_SUCCESS;
}

RetCode HipResourceManager::Init(const ModelConfig& mc, const ResourceConfig& rc) {
    const int tp = rc.tensor_parallel_size;
    if (tp < 1 || (tp & (tp - 1))) {
        LOG(ERROR) << "tensor_parallel_size must be a power of two";
        return RC_INVALID_VALUE;
    }
    // the reference's two modes (src/backends/cuda/resource_manager.cc:49-56): "none" and "online_i8i8" (W8A8: int8 weights
    // quantised per output row at load time unless the slices already hold int8, int8 activations per token at run time)
    const std::string& qm = rc.engine_config.quant_method;
    const bool online_i8i8 = qm == "online_i8i8";
    if (!online_i8i8 && qm != "none" && !qm.empty()) {
        LOG(ERROR) << "unknown/unsupported --quant-method option: " << qm
                   << " (weight-only quantisation is a property of the exported slices: params.json weight_quant_bit)";
        return RC_UNSUPPORTED;
    }
    if (online_i8i8 && mc.weight_quant_bit == 4) {
        LOG(ERROR) << "--quant-method online_i8i8 needs fp16 or int8 (per-channel) slices; these are W4A16";
        return RC_UNSUPPORTED;
    }
    pplhip_model_desc d;
    memset(&d, 0, sizeof(d));
    d.hidden_dim = mc.hidden_dim;
    d.intermediate_dim = mc.intermediate_dim;
    d.num_layers = mc.num_layers;
    d.num_heads = mc.num_heads;
    d.num_kv_heads = mc.num_kv_heads;
    d.vocab_size = mc.vocab_size;
    d.norm_eps = mc.norm_eps;
    d.rope_theta = mc.rope_theta;
    d.max_position = mc.max_position;
    d.cache_quant_bit = mc.cache_quant_bit;
    d.cache_quant_group = mc.cache_quant_group;
    d.cache_layout = mc.cache_layout;
    d.cache_mode = mc.cache_mode;
    d.page_size = mc.page_size;
    d.weight_quant_bit = online_i8i8 ? 8 : mc.weight_quant_bit;
    d.weight_quant_group = mc.weight_quant_group;
    d.act_quant_bit = online_i8i8 ? 8 : 0;

    pplhip_opts o;
    memset(&o, 0, sizeof(o));
    o.n_local_ranks = tp;  // single process, one worker thread per GPU, ncclCommInitAll: the reference's mode
    o.world_size = tp;
    o.max_running_batch = rc.max_running_batch;
    o.max_tokens_per_step = rc.max_tokens_per_step;
    o.enable_penalty = rc.enable_penalty ? 1 : 0;
    o.decoding_attn_split_k = rc.engine_config.configure_decoding_attn_split_k;
    o.decoding_attn_tpb = rc.engine_config.specify_decoding_attn_tpb;
    const int st = pplhip_init(&d, &o, &ctx);
    if (st) {
        LOG(ERROR) << "pplhip_init failed: " << GetRetCodeStr(FromPplHipStatus(st));
        return FromPplHipStatus(st);
    }
    tensor_parallel_size = (uint32_t)tp;
    items.resize(tp);
    runtimes.resize(tp);
    RetCode r = device_worker_pool_.Init(tp);
    if (r != RC_SUCCESS) {
        LOG(ERROR) << "init device worker failed.";
        return r;
    }
    Barrier barrier;
    barrier.Reset(tp);
    r = utils::ParallelExecute(InitRank, &device_worker_pool_, ctx, rc, rc.max_tokens_scale, &barrier, this);
    if (r != RC_SUCCESS) {
        LOG(ERROR) << "ParallelExecute(InitTask) failed.";
        return r;
    }
    post_processor.reset(new HipPostProcessor(ctx));
    return post_processor->InitPostProcessorMem(rc.max_running_batch, mc.vocab_size, rc.enable_penalty);
}

void HipResourceManager::FillResource(Resource* resource) {
    resource->tensor_parallel_size = tensor_parallel_size;
    resource->kv_cache_max_tokens = kv_cache_max_tokens;
    resource->items = items;
    resource->post_processor = post_processor.get();
    resource->device_worker_pool_ = &device_worker_pool_;
}

}}}  // namespace ppl::llm::hip
