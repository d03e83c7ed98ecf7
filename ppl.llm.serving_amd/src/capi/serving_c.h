/* C ABI of the serving core (LLMGenerator + engine + hip backend) for front ends that are not C++: the gRPC server of
 * ppl.llm.serving_amd/serving/grpc_server.py binds it with ctypes.  It plays the role of the reference's
 * GRPCConnection/GRPCServer pair towards the generator (src/serving/grpc/grpc_server.cc:88-341): requests go in through
 * LLMGenerator::Process, responses come back through a Connection whose Send() fills a queue that pplsrv_poll drains. */
#ifndef PPLSRV_SERVING_C_H_
#define PPLSRV_SERVING_C_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define PPLSRV_API __attribute__((visibility("default")))

typedef struct pplsrv pplsrv;

/* the tool flags of tools/offline_inference.cc:40-90 that matter to a server (0 / NULL = the tool's default) */
typedef struct pplsrv_config {
    const char* model_param_path;  /* params.json */
    const char* model_dir;         /* model_slice_<rank>/weights.pplhip; ignored with synthetic_weights */
    int32_t tensor_parallel_size;
    int32_t synthetic_weights;
    uint64_t synthetic_seed;
    uint64_t kv_cache_max_tokens;  /* 0: max_tokens_scale x free memory */
    float max_tokens_scale;
    int32_t max_running_batch, max_tokens_per_step;
    int32_t max_input_tokens_per_request, max_output_tokens_per_request, max_total_tokens_per_request;
    int32_t max_prefill_batch, max_cooldown_request;
    int32_t enable_prefix_cache, enable_penalty;
    const int32_t* stop_tokens;    /* EOS-like tokens (GeneratorConfig::stop_tokens) */
    int32_t n_stop_tokens;
    const char* tokenizer_path;    /* --tokenizer-path: text requests are tokenised / detokenised inside the generator (src/tokenizer);
                                      NULL or "": token-in/token-out only, a text request fails */
    const char* tokenizer_type;    /* --tokenizer-type, NULL = "sentencepiece" */
    const char* model_type;        /* --model-type, NULL = "llama" (LlamaTokenizer: BOS first) */
    const char* quant_method;      /* --quant-method: NULL / "none" / "online_i8i8" */
    float top_p;                   /* --top-p, --top-k: the generator's defaults (GeneratorConfig; 0 / 0 = the tools' 0.0 and 1) */
    int32_t top_k;
    int32_t decoding_attn_split_k; /* --configure-decoding-attn-split-k + 1 (0 = the default, heuristic) */
    int32_t decoding_attn_tpb;     /* --specify-decoding-attn-tpb */
} pplsrv_config;

/* ParseRequest of grpc_server.cc:218-252 already applied by the caller */
typedef struct pplsrv_request {
    uint64_t id;
    const int32_t* tokens;
    int32_t n_tokens;
    float temperature, top_p;
    int32_t top_k;
    float repetition_penalty, presence_penalty, frequency_penalty;
    int32_t generation_length;
    int32_t early_stopping;
    const char* prompt;            /* text request (proto Request.prompt) when tokens == NULL: n_prompt UTF-8 bytes */
    int32_t n_prompt;
} pplsrv_request;

enum { PPLSRV_PROCESSING = 0, PPLSRV_FINISHED = 1, PPLSRV_FAILED = 2 };          /* proto Status */
enum { PPLSRV_REASON_LENGTH = 0, PPLSRV_REASON_EOS = 1, PPLSRV_REASON_STOP = 2 }; /* proto FinishReason */

typedef struct pplsrv_response {
    uint64_t id;
    int32_t token;
    float logprob;
    int32_t status;
    int32_t finish_reason;
    int32_t is_special;
    int32_t text_len;              /* text requests: bytes of this response's `generated` text ... */
    int64_t text_off;              /* ... at this offset of the text buffer handed to pplsrv_poll_text (-1: none / did not fit) */
} pplsrv_response;

PPLSRV_API int pplsrv_create(const pplsrv_config* cfg, pplsrv** out);      /* 0 on success, a negated RetCode otherwise */
PPLSRV_API int pplsrv_submit(pplsrv* s, const pplsrv_request* reqs, int32_t n);
/* waits up to timeout_ms for at least one response, then returns up to `max` of them (0 on timeout) */
PPLSRV_API int pplsrv_poll(pplsrv* s, pplsrv_response* out, int32_t max, int32_t timeout_ms);
/* the same, plus the generated text of text requests (what DecodeAndSendTask, llm_generator.cc:58-112, put into Response::generated:
 * possibly empty while a multi-byte character is still incomplete): copied back to back into text_buf */
PPLSRV_API int pplsrv_poll_text(pplsrv* s, pplsrv_response* out, int32_t max, int32_t timeout_ms, char* text_buf, int64_t text_buf_bytes);
PPLSRV_API int pplsrv_cancel(pplsrv* s, uint64_t id);                      /* client went away: LLMGenerator::ClearTask */
PPLSRV_API uint64_t pplsrv_kv_cache_max_tokens(pplsrv* s);
PPLSRV_API void pplsrv_destroy(pplsrv* s);

#ifdef __cplusplus
}
#endif
#endif
