// [PERF] report, same quantities as the reference's PrintProfiler (src/common/profiler.cc:6-68): QPS, TPS =
// output_token_cnt / total_cost, per-phase costs (cur / avg / total, ms), KV usage, prefix-cache hit rate and the
// "schedule cost" share.
#include <stdio.h>

#include "request.h"

namespace ppl { namespace llm {

static void Phase(const char* name, uint64_t cur, uint64_t total, uint64_t steps) {
    fprintf(stderr, "[PERF]  |-- %-16s | cur: %.2f ms, | avg: %.2f ms, | total: %.2f ms\n", name, cur / 1e3,
            steps ? total / 1e3 / steps : 0.0, total / 1e3);
}

void PrintProfiler(const WorkerProfiler& p) {
    const auto& g = p.step_counter.global;
    const auto& c = p.step_counter.current;
    const double secs = g.total_cost / 1e6;
    const double qps = secs > 0 ? p.finished_task_cnt / secs : 0.0;
    const double tps = secs > 0 ? g.output_token_cnt / secs : 0.0;
    const double hit = g.input_token_cnt ? 100.0 * g.cache_hit_count / g.input_token_cnt : 0.0;
    fprintf(stderr, "[PERF] --- step %lu -------------------------------------------------\n", (unsigned long)g.step_cnt);
    fprintf(stderr, "[PERF]  |- memory usage: (%.2f - %.2f) -> %.2f GiB\n", p.dev_mem_total / 1e9, p.dev_mem_free / 1e9,
            (p.dev_mem_total - p.dev_mem_free) / 1e9);
    fprintf(stderr, "[PERF]  |- kv cache usage: %.2f %%\n",
            p.kv_max_blk ? (1.0 - (double)p.kv_rest_blk / p.kv_max_blk) * 100.0 : 0.0);
    fprintf(stderr, "[PERF]  |- pending task number: %lu\n", (unsigned long)p.pending_task_size);
    fprintf(stderr, "[PERF]  |- running batch: %lu, max running batch: %lu\n", (unsigned long)p.running_task,
            (unsigned long)p.max_running_task);
    fprintf(stderr, "[PERF]  |- prefill batch: %lu , prefill tokens: %lu\n", (unsigned long)p.prefill_batch,
            (unsigned long)p.prefill_tokens);
    fprintf(stderr, "[PERF]  |- prefix cache hit rate: %.2f %%\n", hit);
    fprintf(stderr, "[PERF]  |- finished query count: %lu, QPS: %.2f\n", (unsigned long)p.finished_task_cnt, qps);
    fprintf(stderr, "[PERF]  |- gen token count: %lu, avg gen len: %.2f, TPS: %.2f\n", (unsigned long)g.output_token_cnt,
            p.finished_task_cnt ? (double)g.output_token_cnt / p.finished_task_cnt : 0.0, tps);
    fprintf(stderr, "[PERF]  |- pipeline          | cur: %.2f ms, | avg: %.2f ms, | total: %.2f ms\n", c.total_cost / 1e3,
            g.step_cnt ? g.total_cost / 1e3 / g.step_cnt : 0.0, g.total_cost / 1e3);
    Phase("batching", c.prepare_cost, g.prepare_cost, g.step_cnt);
    Phase("set inputs", c.set_input_cost, g.set_input_cost, g.step_cnt);
    Phase("model inference", c.model_forward_cost, g.model_forward_cost, g.step_cnt);
    Phase("choose token", c.choose_token_cost, g.choose_token_cost, g.step_cnt);
    Phase("post process", c.post_process_cost, g.post_process_cost, g.step_cnt);
    fprintf(stderr, "[PERF]  |- schedule cost: %.2f %%\n",
            g.total_cost ? 100.0 * (double)(g.total_cost - g.model_forward_cost - g.choose_token_cost) / g.total_cost : 0.0);
}

}}  // namespace ppl::llm
