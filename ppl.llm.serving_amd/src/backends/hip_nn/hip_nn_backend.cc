#include "hip_nn_backend.h"

#include <cstdarg>
#include <cstring>

#include "../../../../include/pplhip.h"
#include "ppl/common/log.h"
#include "ppl/nn/engines/llm_cuda/options.h"

using namespace ppl::common;

namespace ppl { namespace llm { namespace hip_nn {

namespace {

class HostContext final : public ppl::nn::DeviceContext {
public:
    const char* GetType() const override { return "cpu"; }
    RetCode Configure(uint32_t, ...) override { return RC_SUCCESS; }
};
class HipContext final : public ppl::nn::DeviceContext {
public:
    const char* GetType() const override { return "hip"; }
    RetCode Configure(uint32_t, ...) override { return RC_UNSUPPORTED; }  // the stream stays inside libpplhip
};

struct RankState;

// one of the eleven inputs / the logits output.  Nothing is copied to the device here: Run() does that in one piece.
class StepTensor final : public ppl::nn::Tensor {
public:
    StepTensor(const char* name, datatype_t dt, bool scalar, RankState* rs) : name_(name), scalar_(scalar), rs_(rs) {
        shape_.SetDataType(dt);
        if (scalar) shape_.ReshapeAsScalar();
    }
    const char* GetName() const override { return name_; }
    ppl::nn::TensorShape* GetShape() const override { return const_cast<ppl::nn::TensorShape*>(&shape_); }
    ppl::nn::DeviceContext* GetDeviceContext() const override { return dev_; }
    void SetDeviceContext(ppl::nn::DeviceContext* d) override { dev_ = d; }
    void SetBufferPtr(void* p) override { buf_ = p; }          // kv_cache / kv_scale: must be the slab libpplhip allocated
    void* GetBufferPtr() const override { return buf_; }
    RetCode ReallocBuffer() override { return RC_SUCCESS; }
    void FreeBuffer() override {}                              // per-step buffers live inside libpplhip (llm_engine.cc:31-35)
    RetCode CopyFromHostAsync(const void* src) override {
        if (scalar_) scalar_value_ = *(const int64_t*)src;     // &model_input.decoding_batches etc.: read now
        else host_ = src;                                       // vectors of ModelInput: alive until Run() (same Execute)
        ++copies_;
        return RC_SUCCESS;
    }
    RetCode CopyFromHost(const void* src) override { return CopyFromHostAsync(src); }
    RetCode CopyToHost(void*) const override { return RC_UNSUPPORTED; }
    RetCode ConvertToHost(void*, const ppl::nn::TensorShape&) const override { return RC_UNSUPPORTED; }

    const void* host_ = nullptr;
    int64_t scalar_value_ = 0;
    int copies_ = 0, copies_seen_ = 0;

private:
    const char* name_;
    bool scalar_;
    RankState* rs_;
    ppl::nn::TensorShape shape_;
    ppl::nn::DeviceContext* dev_ = nullptr;
    void* buf_ = nullptr;
};

class NNEngine final : public ppl::nn::Engine {
public:
    const char* GetName() const override { return "llm_hip"; }
    RetCode Configure(uint32_t option, ...) override {
        va_list ap;
        va_start(ap, option);
        if (option == ppl::nn::llm::cuda::ENGINE_CONF_CACHE_PREFILL) cache_prefill = va_arg(ap, int);
        va_end(ap);
        return RC_SUCCESS;
    }
    int cache_prefill = 0;
};

class NNRuntime;

struct RankState {
    pplhip_ctx* ctx = nullptr;
    int rank = 0;
    int cache_mode = 0;
    int vocab = 0;
    HipContext hip_dev;
    HostContext host_dev;
    NNEngine engine;
    std::unique_ptr<NNRuntime> runtime;
};

class NNRuntime final : public ppl::nn::Runtime {
public:
    NNRuntime(RankState* rs, bool quant) : rs_(rs) {
        static const char* names[11] = {"token_ids", "attn_mask", "seq_starts", "kv_starts", "cache_indices", "decoding_batches",
                                        "start_pos", "max_seq_len", "max_kv_len", "kv_cache", "kv_scale"};
        for (int i = 0; i < (quant ? 11 : 10); ++i) {
            const bool scalar = i == 5 || i == 7 || i == 8;
            in_.emplace_back(new StepTensor(names[i], i == 9 ? DATATYPE_INT8 : (i == 10 ? DATATYPE_FLOAT16 : DATATYPE_INT64), scalar, rs));
            in_.back()->SetDeviceContext(&rs->hip_dev);
        }
        logits_.reset(new StepTensor("logits", DATATYPE_FLOAT32, false, rs));
        logits_->SetDeviceContext(&rs->hip_dev);
    }
    uint32_t GetInputCount() const override { return (uint32_t)in_.size(); }
    ppl::nn::Tensor* GetInputTensor(uint32_t i) const override { return i < in_.size() ? in_[i].get() : nullptr; }
    uint32_t GetOutputCount() const override { return 1; }
    ppl::nn::Tensor* GetOutputTensor(uint32_t i) const override { return i == 0 ? logits_.get() : nullptr; }
    uint32_t GetDeviceContextCount() const override { return 1; }
    ppl::nn::DeviceContext* GetDeviceContext(uint32_t) const override { return &rs_->hip_dev; }

    RetCode Run() override {
        StepTensor *tok = in_[0].get(), *seq = in_[2].get(), *kvs = in_[3].get(), *ci = in_[4].get(), *sp = in_[6].get();
        pplhip_step st;
        memset(&st, 0, sizeof(st));
        st.num_tokens = tok->GetShape()->GetDimCount() ? tok->GetShape()->GetDim(0) : 0;
        st.batch = sp->GetShape()->GetDimCount() ? sp->GetShape()->GetDim(0) : 0;
        st.decoding_batches = in_[5]->scalar_value_;
        st.max_seq_len = in_[7]->scalar_value_;
        st.max_kv_len = in_[8]->scalar_value_;
        st.token_inputs = (const int64_t*)tok->host_;
        st.seq_starts = (const int64_t*)seq->host_;
        st.kv_starts = (const int64_t*)kvs->host_;
        st.start_pos = (const int64_t*)sp->host_;
        st.cache_indices = (const int64_t*)ci->host_;
        if (rs_->cache_mode == 1) {
            st.req_list_changed = ci->copies_ != ci->copies_seen_;   // the engine copied the page table this step
            ci->copies_seen_ = ci->copies_;
            st.max_pages = ci->GetShape()->GetDimCount() == 2 ? ci->GetShape()->GetDim(1) : 0;
        } else {
            st.req_list_changed = 1;
        }
        int rc = pplhip_set_inputs(rs_->ctx, rs_->rank, &st);
        if (rc == 0) rc = pplhip_run(rs_->ctx, rs_->rank, rs_->engine.cache_prefill);
        if (rc) {
            LOG(ERROR) << "hip runtime of rank " << rs_->rank << ": " << pplhip_last_error(rs_->ctx, rs_->rank);
            return FromPplHipStatus(rc);
        }
        float* lg = nullptr;
        int64_t stride = 0;
        pplhip_logits(rs_->ctx, rs_->rank, &lg, &stride);
        logits_->SetBufferPtr(lg);
        logits_->GetShape()->Reshape({st.batch, stride});
        return RC_SUCCESS;
    }

private:
    RankState* rs_;
    std::vector<std::unique_ptr<StepTensor>> in_;
    std::unique_ptr<StepTensor> logits_;
};

// the reference's PostProcessor interface (src/common/post_processor.h:25-43) over pplhip_sample / pplhip_penalty
class NNPostProcessor final : public PostProcessor {
public:
    explicit NNPostProcessor(pplhip_ctx* c) : ctx_(c) {}
    RetCode InitPostProcessorMem(int, int, bool) override { return RC_SUCCESS; }  // sized by pplhip_init
    RetCode SampleTopKTopP(const float* logits_device, const float* temperatures_host, const int32_t* top_k_host, const float* top_p_host,
                           int32_t batch, int32_t vocab_size, int32_t batch_stride, int32_t default_top_k, float default_top_p,
                           bool req_list_changed, int32_t* output_host, float* logprob_host, bool enable_penalty) override {
        pplhip_sample_args a;
        memset(&a, 0, sizeof(a));
        a.temperatures = temperatures_host; a.top_k = top_k_host; a.top_p = top_p_host;
        a.batch = batch; a.vocab_size = vocab_size; a.batch_stride = batch_stride;
        a.default_top_k = default_top_k; a.default_top_p = default_top_p;
        a.req_list_changed = req_list_changed; a.enable_penalty = enable_penalty;
        return FromPplHipStatus(pplhip_sample(ctx_, logits_device, &a, output_host, logprob_host));
    }
    RetCode ApplyPenalty(const float* temperatures_host, const float* repetition_penalties_host, const float* presence_penalties_host,
                         const float* frequency_penalties_host, const int64_t* batch_slots_host, const int64_t*, const int64_t*,
                         const int64_t*, int32_t batch, int32_t vocab_size, bool req_list_changed, float* logits) override {
        pplhip_penalty_args a;
        memset(&a, 0, sizeof(a));
        a.temperatures = temperatures_host; a.repetition_penalties = repetition_penalties_host;
        a.presence_penalties = presence_penalties_host; a.frequency_penalties = frequency_penalties_host;
        a.batch_slots = batch_slots_host; a.batch = batch; a.vocab_size = vocab_size; a.req_list_changed = req_list_changed;
        return FromPplHipStatus(pplhip_penalty(ctx_, logits, &a));  // (the step's device-resident arrays are libpplhip's own)
    }

private:
    pplhip_ctx* ctx_;
};

}  // namespace

struct Backend::Impl {
    pplhip_ctx* ctx = nullptr;
    int tp = 0;
    uint64_t kv_tokens = 0;
    std::vector<std::unique_ptr<RankState>> ranks;
    std::vector<void*> kv_cache, kv_scale;
    std::unique_ptr<NNPostProcessor> post;
    StaticThreadPool pool;
};

Backend::Backend() : impl_(new Impl()) {}
Backend::~Backend() {
    impl_->ranks.clear();
    impl_->post.reset();
    if (impl_->ctx) pplhip_destroy(impl_->ctx);
}

RetCode Backend::Init(const ModelConfig& mc, const ResourceConfig& rc, const ExtraConfig& ex) {
    Impl& I = *impl_;
    I.tp = rc.tensor_parallel_size;
    pplhip_model_desc d;
    memset(&d, 0, sizeof(d));
    d.hidden_dim = mc.hidden_dim; d.intermediate_dim = mc.intermediate_dim; d.num_layers = mc.num_layers;
    d.num_heads = mc.num_heads; d.num_kv_heads = mc.num_kv_heads; d.vocab_size = mc.vocab_size;
    d.norm_eps = ex.norm_eps; d.rope_theta = ex.rope_theta; d.max_position = ex.max_position;
    d.cache_quant_bit = mc.cache_quant_bit; d.cache_quant_group = mc.cache_quant_group; d.cache_layout = mc.cache_layout;
    d.cache_mode = mc.cache_mode; d.page_size = mc.page_size;
    d.weight_quant_bit = ex.weight_quant_bit; d.weight_quant_group = ex.weight_quant_group;
    if (rc.engine_config.quant_method == "online_i8i8") {  // the reference's W8A8 mode (src/backends/cuda/resource_manager.cc:51-52)
        if (d.weight_quant_bit == 4) return ppl::common::RC_UNSUPPORTED;
        d.weight_quant_bit = 8; d.act_quant_bit = 8;
    } else if (rc.engine_config.quant_method != "none" && !rc.engine_config.quant_method.empty()) {
        return ppl::common::RC_UNSUPPORTED;
    }
    pplhip_opts o;
    memset(&o, 0, sizeof(o));
    o.n_local_ranks = I.tp; o.world_size = I.tp;
    o.max_running_batch = rc.max_running_batch; o.max_tokens_per_step = ex.max_tokens_per_step;
    o.enable_penalty = rc.enable_penalty;
    o.decoding_attn_split_k = rc.engine_config.configure_decoding_attn_split_k;
    o.decoding_attn_tpb = rc.engine_config.specify_decoding_attn_tpb;
    int st = pplhip_init(&d, &o, &I.ctx);
    if (st) { LOG(ERROR) << "pplhip_init failed: " << st; return FromPplHipStatus(st); }
    if (I.pool.Init(I.tp) != RC_SUCCESS) return RC_OTHER_ERROR;
    for (int r = 0; r < I.tp; ++r) {
        st = ex.synthetic_weights ? pplhip_rank_init_synthetic(I.ctx, r, ex.synthetic_seed)
                                  : pplhip_rank_load(I.ctx, r, (rc.model_dir + "/model_slice_" + std::to_string(r)).c_str());
        if (st) { LOG(ERROR) << "weights of rank " << r << ": " << pplhip_last_error(I.ctx, r); return FromPplHipStatus(st); }
    }
    I.kv_tokens = ex.kv_cache_max_tokens;
    if (!I.kv_tokens) {
        st = pplhip_kv_capacity(I.ctx, rc.max_tokens_scale, &I.kv_tokens);
        if (st) return FromPplHipStatus(st);
    }
    I.kv_cache.resize(I.tp); I.kv_scale.resize(I.tp);
    for (int r = 0; r < I.tp; ++r) {
        st = pplhip_kv_alloc(I.ctx, r, I.kv_tokens);
        if (st) { LOG(ERROR) << "kv slab of rank " << r << ": " << pplhip_last_error(I.ctx, r); return FromPplHipStatus(st); }
        pplhip_kv_ptrs(I.ctx, r, &I.kv_cache[r], &I.kv_scale[r]);
        std::unique_ptr<RankState> rs(new RankState());
        rs->ctx = I.ctx; rs->rank = r; rs->cache_mode = mc.cache_mode; rs->vocab = mc.vocab_size;
        rs->runtime.reset(new NNRuntime(rs.get(), mc.cache_quant_bit > 0));
        I.ranks.push_back(std::move(rs));
    }
    I.post.reset(new NNPostProcessor(I.ctx));
    return RC_SUCCESS;
}

void Backend::FillResource(Resource* res) {
    Impl& I = *impl_;
    res->tensor_parallel_size = (uint32_t)I.tp;
    res->kv_cache_max_tokens = I.kv_tokens;
    res->items.resize(I.tp);
    for (int r = 0; r < I.tp; ++r) {
        res->items[r].kv_cache_mem = I.kv_cache[r];
        res->items[r].kv_scale_mem = I.kv_scale[r];
        res->items[r].runtime = I.ranks[r]->runtime.get();
        res->items[r].host_device = &I.ranks[r]->host_dev;
        res->items[r].engine = &I.ranks[r]->engine;
    }
    res->post_processor = I.post.get();
    res->device_worker_pool_ = &I.pool;
    res->tokenizer = nullptr;
}

}}}  // namespace ppl::llm::hip_nn
