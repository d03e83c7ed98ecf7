// libpplhip.so -- the C ABI of include/pplhip.h: per-rank runtime (streams, weights, KV slab, step inputs,
// activations), the decoder forward as a sequence of hand-written gfx950 kernels, RCCL collectives for tensor
// parallelism and the sampler.  This file replaces what the reference reaches through ppl.nn
// (Engine/Runtime/Tensor, src/backends/cuda/resource_manager.cc:43-211) and ppl.llm.kernel.cuda.
#include "../../include/pplhip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

using namespace pplhip;

namespace {

struct Linear {
    int N = 0, K = 0, qbit = 0, group = 0;
    int Kp = 0;  // row stride on the device: K rounded up to the GEMM k-tile (64) with zero columns, so that a tensor-
                 // parallel slice like 11008 / 8 = 1376 still runs on the DMA tile kernels (the activation operand is
                 // padded with zeros to the same stride)
    void* w = nullptr;
    uint16_t* scale = nullptr;
    uint64_t elt_bytes(uint64_t n) const { return qbit == 0 ? n * 2 : qbit == 8 ? n : n / 2; }
    uint64_t w_bytes() const { return elt_bytes((uint64_t)N * K); }       // container (unpadded) size
    uint64_t alloc_bytes() const { return elt_bytes((uint64_t)N * Kp); }  // device size
    uint64_t s_bytes() const { return qbit == 0 ? 0 : qbit == 8 ? (uint64_t)N * 2 : (uint64_t)N * (K / group) * 2; }
};

struct GraphEntry { hipGraphExec_t exec; uint64_t tick; };

struct Layer {
    uint16_t* attn_norm = nullptr;
    uint16_t* ffn_norm = nullptr;
    Linear wqkv, wo, w13, w2;
};

struct ProfEvent {
    int cls;
    hipEvent_t a, b;
};

struct Rank {
    int device = 0;
    int global_rank = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    ncclComm_t comm2 = nullptr;         // the second half-batch's collectives of a two-stream decode step (a communicator serialises its streams)
    hipStream_t comm_stream = nullptr;  // collectives of an overlapped step (pplhip_run)
    // two-stream decode (run_launches, "dual"): the second half of a mid-size pure-decode step runs its layers on stream2, beside the first
    // half on `stream`, with split-K / attention workspaces of its own; ev_fork / ev_join order the two around the step
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* gemm_ws2 = nullptr;
    bool dual_seen = false;
    bool keep_part2 = false;            // this launch of w2 must write part2 (its consumer runs on the other stream): no deferred slabs
    hipEvent_t ev_compute[2] = {nullptr, nullptr}, ev_comm[2] = {nullptr, nullptr};
    const int64_t* h_seq = nullptr;     // host copy of this step's seq_starts (lives in the staging buffer)
    std::string err;

    // weights
    uint16_t* embed = nullptr;
    uint16_t* norm = nullptr;
    Linear output;
    std::vector<Layer> layers;
    float* rope = nullptr;
    std::vector<void*> allocs;

    // KV slab
    uint64_t kv_tokens = 0;
    void* kv_cache = nullptr;
    uint16_t* kv_scale = nullptr;

    // step inputs
    int64_t* stage_host[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_cur = 0;
    int64_t* step_dev = nullptr;  // packed: token_ids | seq_starts | kv_starts | start_pos | cache_indices(mode 0)
    int64_t cap_T = 0, cap_B = 0;
    int64_t *d_tok = nullptr, *d_seq = nullptr, *d_kvs = nullptr, *d_sp = nullptr, *d_ci = nullptr;
    int64_t* pages_dev = nullptr;   // mode 1: [B, max_pages]
    int64_t* pages_host = nullptr;  // pinned
    uint64_t pages_cap = 0;
    int64_t B = 0, T = 0, decoding_batches = 0, max_seq_len = 0, max_kv_len = 0, max_pages = 0;
    // split-K results left unreduced for the kernel that consumes them (kernels.h SplitSlabs): wqkv -> RoPE + KV write, wo -> FFN norm,
    // w2 -> the next layer's attention norm (or the final norm)
    SplitSlabs sl_qkv, sl_part, sl_part2;
    bool defer_reduce = false;   // this step: tensor-parallel size 1, weight-only quantisation, no residual dump
    bool defer_qkv = false;      // ... wqkv's slabs alone also under tensor parallelism (RoPE + KV write consumes them on the rank itself)

    // activations
    uint16_t *h = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *part = nullptr, *part2 = nullptr, *gu = nullptr,
             *act = nullptr, *hn = nullptr;
    std::unordered_map<uint64_t, GraphEntry> graphs;  // captured pure-decode steps by shape (run_decode_graph)
    uint64_t graph_tick = 0;
    int8_t* xq = nullptr;  // online_i8i8: int8 activations [cap_T, max row] and per-token scales
    float* sx = nullptr;
    float* logits_local = nullptr;  // [B, V/tp] (tp > 1)
    float* logits_gather = nullptr; // [tp, B, V/tp]
    float* logits = nullptr;        // [B, V]
    float* attn_ws = nullptr;
    size_t attn_ws_bytes = 0;
    float* gemm_ws = nullptr;  // split-K partial slabs
    size_t gemm_ws_bytes = 0;

    // sampler (local rank 0)
    float *d_temp = nullptr, *d_topp = nullptr, *d_rand = nullptr, *d_lp = nullptr;
    int32_t* d_tokout = nullptr;
    uint16_t* count_map = nullptr;
    int64_t* d_slots = nullptr;
    float *d_rep = nullptr, *d_pres = nullptr, *d_freq = nullptr, *d_ptemp = nullptr;
    float* h_rand = nullptr;  // pinned

    // exchange region of the direct collectives (k_comm.hip): fine-grained, peer-mapped; holds the flag words and the buffers
    // the collectives work in place on (part, part2) or push into (logits_gather)
    char* xbase = nullptr;
    size_t xbytes = 0, x_part = 0, x_part2 = 0, x_scratch[2] = {0, 0}, x_local = 0;  // byte offsets inside the region
    uint32_t ar_count = 0;              // all-reduces issued (scratch double buffer)
    // channel 1 of the direct collectives (second half-batch of a two-stream decode step): flag set, scratch pair, counters of its own
    size_t x_scratch2[2] = {0, 0};
    uint32_t ar_count2 = 0, p2p_epoch2 = 0;
    int channel = 0;                    // which of the two the launches being issued belong to
    P2pPeers peers{};
    bool peer_ipc[P2P_MAX_RANKS] = {};  // peers.base[g] came from hipIpcOpenMemHandle (closed on destroy)
    uint32_t p2p_epoch = 0;
    uint32_t* p2p_status = nullptr;     // pinned host word a kernel raises when one of its bounded spins timed out
    int32_t* d_flag = nullptr;          // one int for the cross-process agreement on the self-test
    // stream hand-offs of the two-chunk schedule through device-memory epochs (k_comm.hip handoff_*): words 0..1 = "chunk i's
    // partial sums are complete" (compute -> communication stream), 2..3 = "chunk i is reduced" (communication -> compute)
    uint32_t* hflags = nullptr;
    uint32_t h_ready_ep[2] = {0, 0}, h_done_ep[2] = {0, 0};

    // diagnosis (pplhip_debug_run_dump): residual stream h and the pending row-parallel FFN output after every layer
    uint16_t* dump_dev = nullptr;  // [L+1][2][T, hidden] fp16 (slot 0 = h, slot 1 = pending), allocated for one run

    // profiling
    std::vector<ProfEvent> prof;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
};

}  // namespace

struct pplhip_ctx {
    pplhip_model_desc d;
    pplhip_opts o;
    int tp = 1;
    bool tp_overlap = true;              // PPLHIP_TP_OVERLAP=0 keeps the collectives on the compute stream
    // PPLHIP_TP_OVERLAP_MIN_TOKENS.  Measured on one MI355X with identity collectives (bench.py --emulate-tp 2/4/8,
    // profiles/r01_tp_emulation.txt): the two-chunk schedule costs a 1024-row decode step +6..7 ms (27 us per
    // cross-stream hand-off x 128, plus two half-size GEMM launches instead of one) -- about what the 64 all-reduces
    // of 8 MB it hides are worth -- so it is used for steps that carry prefill tokens (T >= 2048: 16+ MB per
    // all-reduce, chunks of >= 1024 rows keep the GEMM tiles full), not for pure decode steps of <= 1024 rows.
    int64_t tp_overlap_min_tokens = 2048;
    bool handoff_flags = true;  // PPLHIP_TP_HANDOFF=events: HIP events instead of device-memory epochs between the two streams
    bool tp_on = false;     // the tensor-parallel step schedule (collectives after wo / w2, logits gather) is active
    // collectives: 2 = direct kernels over peer-mapped memory (k_comm.hip), 1 = RCCL, 0 = none.  PPLHIP_COMM=auto (default:
    // direct when the self-test passes on every rank, else RCCL) | p2p (direct or fail) | rccl
    int comm_mode = 0;
    int comm_want = 0;      // 0 auto, 1 rccl only, 2 p2p only
    bool p2p_connected = false;
    uint64_t p2p_timeout_ticks = 0;  // s_memrealtime ticks (100 MHz)
    // PPLHIP_DUAL_STREAM=1: pure-decode steps of dual_min_rows..dual_max_rows rows as two half-batches on two streams (run_launches)
    // -1 (default): automatic -- under real tensor parallelism (a communicator over >= 2 ranks) pure-decode steps of 512..1024 rows run as
    // two half-batches on two streams, so that each half's all-reduces overlap the other half's matmuls (the north star's schedule;
    // +0.25 ms of compute per 1024-row 7B / TP8 step against ~2.2 ms of link time, profiles/r04_late_experiments.md 1); config 4's 256
    // rows stay on one stream (halves of 128 rows measured -8 %).  Automatic only on the direct collectives (two channels, validated on
    // one device at tp 2 / 4): two RCCL communicators running concurrently on one device have never run on >= 2 devices, so a
    // group that fell back to RCCL keeps one stream unless PPLHIP_DUAL_STREAM=1 asks for it.  1: rows dual_min_rows..dual_max_rows at any tp; 0: never
    int dual_mode = -1;
    bool dual_auto = false;          // the automatic rule is in force (row window 512..1024)
    int64_t dual_min_rows = 96, dual_max_rows = 512;
    // Sequence-parallel residual stream (round 6): on the direct collectives the all-reduce behind wo / w2 also does the residual add and the
    // RMSNorm that consumes it, on the rows the rank owns after the first shot, and the second shot gathers NORMED rows (k_comm.hip
    // p2p_allreduce_norm_kernel): 1 / tp of the norm work per rank and one launch less per half-layer.  Each rank then keeps only its own
    // rows of the residual stream h.  Not with int8 activations (the norm writes the quantised operand there), residual dumps, RCCL
    // (plain all-reduce + replicated norm as before) or hidden > 8192.  PPLHIP_TP_FUSE_NORM=0: off (A/B runs)
    bool fuse_norm_want = true;
    bool emulate_tp = false;         // PPLHIP_EMULATE_TP (bench.py --emulate-tp): one rank's slice, collectives are local identities
    int selftest = 0;                // direct collectives' start-up self-test: 0 not run, 1 passed on every rank, -1 failed (RCCL in charge)
    std::string comm_notes;          // every fallback taken at start-up, in order (never silent: also on stderr)
    bool graph_on = false;           // PPLHIP_DECODE_GRAPH=1: replay pure-decode steps as HIP graphs (opt-in, see run_decode_graph)
    int64_t graph_max_batch = 64;    // above this a step is seconds of GPU work per thousand launches: nothing to gain
    int H = 0, Hkv = 0, D = 0, inter = 0, vocab_local = 0;
    std::vector<Rank> ranks;
    std::string err;
};

namespace {

bool verbose() {   // PPLHIP_VERBOSE: messages on stderr (errors, the collectives' choice, graph capture)
    static const bool on = getenv("PPLHIP_VERBOSE") != nullptr;
    return on;
}

// a fallback was taken (direct collectives -> RCCL, two streams -> one): always on stderr, and kept for pplhip_comm_info
void comm_note(pplhip_ctx* c, const std::string& msg) {
    fprintf(stderr, "[pplhip] %s\n", msg.c_str());
    if (c) { if (!c->comm_notes.empty()) c->comm_notes += "; "; c->comm_notes += msg; }
}

int fail(pplhip_ctx* c, int rank, int code, const std::string& msg) {
    if (c) {
        if (rank >= 0 && rank < (int)c->ranks.size()) c->ranks[rank].err = msg; else c->err = msg;
    }
    if (verbose()) fprintf(stderr, "[pplhip] error %d (rank %d): %s\n", code, rank, msg.c_str());
    return code;
}

#define HIPCK(c, r, expr)                                                                                            \
    do {                                                                                                             \
        hipError_t e__ = (expr);                                                                                     \
        if (e__ != hipSuccess)                                                                                       \
            return fail(c, r, e__ == hipErrorOutOfMemory ? PPLHIP_OUT_OF_MEMORY : PPLHIP_DEVICE_RUNTIME_ERROR,       \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                                         \
    } while (0)

#define NCCLCK(c, r, expr)                                                                                           \
    do {                                                                                                             \
        ncclResult_t e__ = (expr);                                                                                   \
        if (e__ != ncclSuccess)                                                                                      \
            return fail(c, r, PPLHIP_DEVICE_RUNTIME_ERROR, std::string(#expr) + ": " + ncclGetErrorString(e__));     \
    } while (0)

static inline float h2f_host(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
static inline uint16_t f2h_host(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// k-tile the padded row stride of w2 / the SwiGLU output is rounded to: 64 (fp16-activation tile GEMM), 128 (int8 x int8)
static int k_tile(const pplhip_model_desc& d) { return d.act_quant_bit == 8 ? 128 : 64; }

int dev_alloc(pplhip_ctx* c, int r, void** p, uint64_t bytes) {
    *p = nullptr;
    if (bytes == 0) return 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(c, r, PPLHIP_OUT_OF_MEMORY, "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e));
    c->ranks[r].allocs.push_back(*p);
    return 0;
}

int linear_alloc(pplhip_ctx* c, int r, Linear* l, int N, int K, int qbit, int group, bool pad_k = false) {
    l->N = N; l->K = K; l->qbit = qbit; l->group = group;
    const int kt = k_tile(c->d);
    l->Kp = (pad_k && qbit != 4) ? (K + kt - 1) / kt * kt : K;
    int rc = dev_alloc(c, r, &l->w, l->alloc_bytes());
    if (rc) return rc;
    if (l->Kp != K) {
        hipError_t e = hipMemset(l->w, 0, l->alloc_bytes());
        if (e != hipSuccess) return fail(c, r, PPLHIP_DEVICE_RUNTIME_ERROR, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    return dev_alloc(c, r, (void**)&l->scale, l->s_bytes());
}

// name -> device buffer of a rank's slice; names: DESIGN.md "weight container"
bool find_tensor(pplhip_ctx* c, Rank& R, const char* name, void** ptr, uint64_t* bytes, Linear** lin_out = nullptr) {
    if (lin_out) *lin_out = nullptr;
    const int hd = c->d.hidden_dim;
    if (!strcmp(name, "tok_embeddings.weight")) { *ptr = R.embed; *bytes = (uint64_t)c->d.vocab_size * hd * 2; return true; }
    if (!strcmp(name, "norm.weight")) { *ptr = R.norm; *bytes = (uint64_t)hd * 2; return true; }
    if (!strcmp(name, "output.weight")) { *ptr = R.output.w; *bytes = R.output.w_bytes(); if (lin_out) *lin_out = &R.output; return true; }
    int l = -1;
    char rest[128];
    if (sscanf(name, "layers.%d.%127s", &l, rest) != 2 || l < 0 || l >= c->d.num_layers) return false;
    Layer& L = R.layers[l];
    if (!strcmp(rest, "attention_norm.weight")) { *ptr = L.attn_norm; *bytes = (uint64_t)hd * 2; return true; }
    if (!strcmp(rest, "ffn_norm.weight")) { *ptr = L.ffn_norm; *bytes = (uint64_t)hd * 2; return true; }
    struct { const char* n; Linear* lin; } tab[] = {{"attention.wqkv", &L.wqkv}, {"attention.wo", &L.wo},
                                                      {"feed_forward.w13", &L.w13}, {"feed_forward.w2", &L.w2}};
    for (auto& t : tab) {
        const size_t nl = strlen(t.n);
        if (strncmp(rest, t.n, nl)) continue;
        if (!strcmp(rest + nl, ".weight")) { *ptr = t.lin->w; *bytes = t.lin->w_bytes(); if (lin_out) *lin_out = t.lin; return true; }
        if (!strcmp(rest + nl, ".scale") && t.lin->qbit) { *ptr = t.lin->scale; *bytes = t.lin->s_bytes(); return true; }
    }
    return false;
}

KvAddr make_kv_addr(const pplhip_model_desc& d, int Hkv, int D, uint64_t tokens, void* cache, uint16_t* scale, int layer) {
    const int elt = d.cache_quant_bit == 8 ? 1 : 2;
    const int g = d.cache_quant_group > 0 ? d.cache_quant_group : 1;
    const KvStrides cs = kv_strides(d.cache_layout, (int64_t)tokens, d.num_layers, Hkv, D);
    const KvStrides ss = kv_strides(d.cache_layout, (int64_t)tokens, d.num_layers, Hkv, D / g);
    KvAddr a;
    a.cache = (char*)cache + (int64_t)layer * cs.sL * elt;
    a.scale = scale ? scale + (int64_t)layer * ss.sL : nullptr;
    a.sKV = cs.sKV; a.sH = cs.sH; a.sN = cs.sN;
    a.ssKV = ss.sKV; a.ssH = ss.sH; a.ssN = ss.sN;
    a.mode = d.cache_mode;
    a.page_size = d.page_size > 0 ? d.page_size : 1;
    a.page_shift = (a.page_size & (a.page_size - 1)) == 0 ? __builtin_ctz((unsigned)a.page_size) : -1;
    return a;
}

void prof_begin(pplhip_ctx* c, Rank& R, int cls, ProfEvent* ev) {
    ev->cls = -1;
    if (!c->o.enable_profiling) return;
    // 2 = light: only the decode-attention launches (through hipExtLaunchKernel's start / stop events) and the whole run.  An event record is a barrier packet on the
    // stream: the full set (10 per layer) costs a batch-1024 decode step 1.3 ms of 43.9 (profiles/probes/prof_overhead.py)
    if (c->o.enable_profiling == 2 && cls != PPLHIP_PROF_RUN) return;   // (decode attention: timed through its own launch)
    std::pair<hipEvent_t, hipEvent_t> p;
    if (!R.prof_free.empty()) { p = R.prof_free.back(); R.prof_free.pop_back(); }
    else { hipEventCreate(&p.first); hipEventCreate(&p.second); }
    ev->cls = cls; ev->a = p.first; ev->b = p.second;
    hipEventRecord(ev->a, R.stream);
}
void prof_end(Rank& R, ProfEvent* ev) {
    if (ev->cls < 0) return;
    hipEventRecord(ev->b, R.stream);
    R.prof.push_back(*ev);
}

int decode_split(const pplhip_ctx* c, int64_t nb, int64_t max_kv_len) {
    const int mode = c->o.decoding_attn_split_k;
    if (mode == 0) return 1;
    const bool gqa = attn_decode_gqa_supported(c->d.cache_quant_bit, c->H, c->Hkv, c->D);
    const int64_t blocks = nb * (gqa ? c->Hkv : c->H);  // GQA kernel: one block per KV head
    int split = 1;
    // measured (profiles/attn_microbench.py): 512 workgroups already stream at 5.3 TB/s; split only below ~256.  (The grouped-query
    // kernel at 256 blocks -- 70B / TP8, 256 requests x 1 KV head -- is faster unsplit: 37 us against 46 us with split 2, the reduce
    // launch costs more than the second block per CU brings, profiles/r03_roofline_sweep.json)
    if (mode == 2 || (blocks < 256 && max_kv_len >= 512)) {
        int64_t want = (512 + blocks - 1) / blocks;            // aim for >= 512 workgroups
        // >= 256 tokens per split; >= 128 when there are very few blocks (batch 1-4 of a multi-head model: kv 512 split 4 8.6 us against 10.3 us
        // with split 2, kv 2048 split 8 11.0 against 13.8 with 4 -- one memory round trip per wave instead of two, round 4)
        // (never below the >= 256-tokens rule: at kv 8192 a single stream keeps split 16 / 32 -- ADVICE r4: the 8-way limit of the
        // 128-token rule had lowered it to 8 from kv 2048 on)
        int64_t cap = std::max<int64_t>(1, std::max<int64_t>(max_kv_len / 256, blocks <= 128 ? std::min<int64_t>(max_kv_len / 128, 8) : 0));
        split = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, cap), 32));
        if (mode == 2 && split < 2 && max_kv_len >= 64) split = 2;
    }
    return split;
}

// ---- direct collectives: connection + self-test -------------------------------------------------------------------------

// the direct path cannot be used: an error when it was demanded (or no RCCL communicator exists), else RCCL stays in charge
int p2p_unavailable(pplhip_ctx* c, int rank, const std::string& why) {
    if (c->comm_want == 2 || c->comm_mode != 1) return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, "direct collectives unavailable: " + why);
    c->selftest = -1;
    comm_note(c, "direct collectives unavailable (" + why + "): RCCL takes over");
    return 0;
}

// Runs both collectives four times on changing patterns on every local rank (all ranks of the group do this at the same time,
// in this process or in others) and compares with the exact answer.  Every rank then learns whether ALL ranks passed
// (one RCCL all-reduce when a communicator exists); only then comm_mode becomes 2.
// `local_failure`: this process could not even map its peers -- it skips the kernels but still takes part in the agreement, so that
// the other processes (whose self-test kernels give up waiting for it after the bounded spin) do not wait in the all-reduce forever.
int p2p_selftest(pplhip_ctx* c, const std::string* local_failure = nullptr) {
    const int n = (int)c->ranks.size(), tp = c->tp, hd = c->d.hidden_dim;
    // granules of the direct kernels (k_comm.hip): 16-byte pieces of the fp16 partial sums, 8- or 16-byte pieces of a shard's fp32
    // logits rows; a model that does not fit them keeps RCCL instead of failing at its first real step
    std::string shape_failure;
    if (!local_failure && (c->vocab_local % 2 != 0 || hd % 8 != 0)) {
        shape_failure = "vocab/tp must be even and hidden a multiple of 8 (vocab_local " + std::to_string(c->vocab_local) + ", hidden " + std::to_string(hd) + ")";
        local_failure = &shape_failure;
    }
    const int64_t cnt = std::min<int64_t>((int64_t)1 << 20, c->ranks[0].cap_T * (int64_t)hd) / 8 * 8;
    // the gather test: every rank's [grows, gcols] fp32 block -> [grows, gcols * tp]
    const int64_t gcols = c->vocab_local / 2 * 2, grows = std::min<int64_t>(c->ranks[0].cap_B, std::max<int64_t>(1, ((int64_t)1 << 18) / gcols));
    const int64_t gcnt = grows * gcols;
    const char* e = getenv("PPLHIP_P2P_SELFTEST_MS");
    const uint64_t ticks = (uint64_t)(e ? std::max(1, atoi(e)) : 10000) * 100000ull;
    auto pat = [](int64_t i, int g, int round) { return p2p_pattern_value(i, g, round); };
    bool ok = local_failure == nullptr;
    std::string why = local_failure ? *local_failure : std::string();
    std::vector<uint16_t> hbuf(cnt);
    std::vector<float> gbuf(gcnt), gall;
    for (int round = 0; round < 4 && ok; ++round) {
        for (int r = 0; r < n; ++r) {
            Rank& R = c->ranks[r];
            HIPCK(c, r, hipSetDevice(R.device));
            // inputs are written by a kernel on the rank's stream, like the partial sums and logits shards of a real step
            HIPCK(c, r, launch_p2p_pattern(R.stream, R.part, cnt, R.logits_local, gcnt, R.global_rank, round));
        }
        for (int r = 0; r < n; ++r) {
            Rank& R = c->ranks[r];
            HIPCK(c, r, hipSetDevice(R.device));
            HIPCK(c, r, launch_p2p_allreduce(R.stream, R.peers, R.global_rank, tp, R.x_part, R.x_scratch[R.ar_count++ & 1], cnt, ++R.p2p_epoch,
                                             ticks, R.p2p_status));
            HIPCK(c, r, launch_p2p_allgather(R.stream, R.peers, R.global_rank, tp, R.x_local, R.logits, grows, gcols * 4, (int64_t)gcols * 4 * tp,
                                             ++R.p2p_epoch, ticks, R.p2p_status));
        }
        for (int r = 0; r < n && ok; ++r) {
            Rank& R = c->ranks[r];
            HIPCK(c, r, hipSetDevice(R.device));
            HIPCK(c, r, hipStreamSynchronize(R.stream));
            if (*R.p2p_status) { ok = false; why = "spin timed out (status " + std::to_string(*R.p2p_status) + ")"; *R.p2p_status = 0; break; }
            HIPCK(c, r, hipMemcpy(hbuf.data(), R.part, cnt * 2, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < cnt && ok; ++i) {
                float want = 0.f;
                for (int g = 0; g < tp; ++g) want += pat(i, g, round);
                if ((float)__builtin_bit_cast(_Float16, hbuf[i]) != want) {
                    ok = false;
                    int64_t nbad = 0, last = i;
                    for (int64_t j = i; j < cnt; ++j) {
                        float w2 = 0.f;
                        for (int g = 0; g < tp; ++g) w2 += pat(j, g, round);
                        if ((float)__builtin_bit_cast(_Float16, hbuf[j]) != w2) { ++nbad; last = j; }
                    }
                    why = "all-reduce mismatch: round " + std::to_string(round) + " rank " + std::to_string(R.global_rank) + " first " + std::to_string(i) +
                          " last " + std::to_string(last) + " bad " + std::to_string(nbad) + "/" + std::to_string(cnt) + " got " +
                          std::to_string((float)__builtin_bit_cast(_Float16, hbuf[i])) + " want " + std::to_string(want);
                }
            }
            gall.resize((size_t)gcnt * tp);
            HIPCK(c, r, hipMemcpy(gall.data(), R.logits, gall.size() * 4, hipMemcpyDeviceToHost));
            for (int g = 0; g < tp && ok; ++g)
                for (int64_t i = 0; i < gcnt && ok; ++i)
                    if (gall[(size_t)(i / gcols) * gcols * tp + (size_t)g * gcols + i % gcols] != pat(i, g, round + 2)) {
                        ok = false;
                        int64_t nbad = 0, last = i;
                        for (int64_t j = i; j < gcnt; ++j)
                            if (gall[(size_t)(j / gcols) * gcols * tp + (size_t)g * gcols + j % gcols] != pat(j, g, round + 2)) { ++nbad; last = j; }
                        why = "all-gather mismatch: round " + std::to_string(round) + " rank " + std::to_string(R.global_rank) + " block " + std::to_string(g) +
                              " first " + std::to_string(i) + " last " + std::to_string(last) + " bad " + std::to_string(nbad) + "/" + std::to_string(gcnt) +
                              " got " + std::to_string(gall[(size_t)(i / gcols) * gcols * tp + (size_t)g * gcols + i % gcols]) + " want " + std::to_string(pat(i, g, round + 2)) +
                              " prev-round " + std::to_string(pat(i, g, round + 1));
                    }
        }
    }
    // the fused all-reduce + residual add + RMSNorm (p2p_allreduce_norm_kernel, what a step issues when fuse_norm_active): against a LOCAL
    // reference built from parts that are already verified -- the plain all-reduce above leaves the exact sums on every rank, the kernel's
    // own one-rank form ("solo": same arithmetic, no peers) turns them into the expected residual and normed rows; the real kernel, run on
    // the unreduced patterns, must then give the same bits in every row it gathered and in the residual rows it owns
    const int64_t frows = hd > 0 ? cnt / hd : 0;
    if (ok && c->fuse_norm_want && c->d.act_quant_bit != 8 && hd % 8 == 0 && hd <= P2P_NORM_MAX_HIDDEN && frows >= 1 &&
        c->ranks[0].gemm_ws_bytes >= (size_t)frows * hd * 4 + (size_t)hd * 2) {
        const int64_t fcnt = frows * hd;
        std::vector<uint16_t> want_x(fcnt), want_h(fcnt);
        for (int round = 0; round < 2 && ok; ++round) {
            for (int r = 0; r < n; ++r) {
                Rank& R = c->ranks[r];
                HIPCK(c, r, hipSetDevice(R.device));
                uint16_t* xref = (uint16_t*)R.gemm_ws, *href = xref + fcnt, *w = href + fcnt;
                HIPCK(c, r, launch_p2p_pattern(R.stream, R.part, fcnt, nullptr, 0, R.global_rank, round + 4));
                HIPCK(c, r, launch_p2p_allreduce(R.stream, R.peers, R.global_rank, tp, R.x_part, R.x_scratch[R.ar_count++ & 1], fcnt, ++R.p2p_epoch, ticks, R.p2p_status));
                HIPCK(c, r, launch_p2p_pattern(R.stream, R.h, fcnt, nullptr, 0, 17, round + 4));    // the residual stream: the same on every rank
                HIPCK(c, r, launch_p2p_pattern(R.stream, w, hd, nullptr, 0, 23, round + 4));
                HIPCK(c, r, hipMemcpyAsync(href, R.h, (size_t)fcnt * 2, hipMemcpyDeviceToDevice, R.stream));
                HIPCK(c, r, launch_p2p_allreduce_norm(R.stream, R.peers, R.global_rank, 1, R.x_part, 0, frows, hd, href, w, c->d.norm_eps, xref, 0, ticks, R.p2p_status));
                HIPCK(c, r, launch_p2p_pattern(R.stream, R.part, fcnt, nullptr, 0, R.global_rank, round + 4));
                HIPCK(c, r, launch_p2p_allreduce_norm(R.stream, R.peers, R.global_rank, tp, R.x_part, R.x_scratch[R.ar_count++ & 1], frows, hd, R.h, w, c->d.norm_eps,
                                                      R.xn, ++R.p2p_epoch, ticks, R.p2p_status));
            }
            for (int r = 0; r < n && ok; ++r) {
                Rank& R = c->ranks[r];
                HIPCK(c, r, hipSetDevice(R.device));
                HIPCK(c, r, hipStreamSynchronize(R.stream));
                if (*R.p2p_status) { ok = false; why = "fused all-reduce + norm: spin timed out (status " + std::to_string(*R.p2p_status) + ")"; *R.p2p_status = 0; break; }
                HIPCK(c, r, hipMemcpy(want_x.data(), R.gemm_ws, (size_t)fcnt * 2, hipMemcpyDeviceToHost));
                HIPCK(c, r, hipMemcpy(want_h.data(), (uint16_t*)R.gemm_ws + fcnt, (size_t)fcnt * 2, hipMemcpyDeviceToHost));
                HIPCK(c, r, hipMemcpy(hbuf.data(), R.xn, (size_t)fcnt * 2, hipMemcpyDeviceToHost));
                int64_t bad = 0, first = -1;
                for (int64_t i = 0; i < fcnt; ++i) if (hbuf[i] != want_x[i]) { if (!bad) first = i; ++bad; }
                if (bad) { ok = false; why = "fused all-reduce + norm: " + std::to_string(bad) + "/" + std::to_string(fcnt) + " normed values differ on rank " + std::to_string(R.global_rank) + " (first: row " + std::to_string(first / hd) + ")"; break; }
                const int64_t per = (frows + tp - 1) / tp, lo = std::min<int64_t>(per * R.global_rank, frows), hi = std::min<int64_t>(lo + per, frows);
                HIPCK(c, r, hipMemcpy(hbuf.data(), R.h, (size_t)fcnt * 2, hipMemcpyDeviceToHost));
                for (int64_t i = lo * hd; i < hi * hd; ++i) if (hbuf[i] != want_h[i]) { ++bad; }
                if (bad) { ok = false; why = "fused all-reduce + norm: " + std::to_string(bad) + " residual values of the owned rows differ on rank " + std::to_string(R.global_rank); break; }
            }
        }
    }
    // agreement: min over all ranks of the group
    if (c->comm_mode == 1) {
        for (int r = 0; r < n; ++r) {
            Rank& R = c->ranks[r];
            HIPCK(c, r, hipSetDevice(R.device));
            const int32_t v = ok ? 1 : 0;
            HIPCK(c, r, hipMemcpy(R.d_flag, &v, 4, hipMemcpyHostToDevice));
        }
        NCCLCK(c, -1, ncclGroupStart());
        for (int r = 0; r < n; ++r) NCCLCK(c, r, ncclAllReduce(c->ranks[r].d_flag, c->ranks[r].d_flag, 1, ncclInt32, ncclMin, c->ranks[r].comm, c->ranks[r].stream));
        NCCLCK(c, -1, ncclGroupEnd());
        for (int r = 0; r < n; ++r) {
            Rank& R = c->ranks[r];
            HIPCK(c, r, hipSetDevice(R.device));
            HIPCK(c, r, hipStreamSynchronize(R.stream));
            int32_t v = 0;
            HIPCK(c, r, hipMemcpy(&v, R.d_flag, 4, hipMemcpyDeviceToHost));
            if (!v && ok) { ok = false; why = "another rank failed its self-test"; }
        }
    }
    if (!ok) {
        // the check loop stopped at the first rank that showed a failure: drain EVERY local rank (the others normally time out on
        // the same dead peer) and clear every status word, or the RCCL steps that follow would report a stale "timed out"
        for (int r = 0; r < n; ++r) {
            Rank& R = c->ranks[r];
            (void)hipSetDevice(R.device);
            (void)hipStreamSynchronize(R.stream);
            if (R.p2p_status) *R.p2p_status = 0;
        }
        return p2p_unavailable(c, 0, "self-test: " + why);
    }
    c->comm_mode = 2;
    c->selftest = 1;
    if (verbose()) fprintf(stderr, "[pplhip] direct collectives over peer-mapped memory: self-test passed on %d local rank(s) of %d\n", n, tp);
    return 0;
}

// all ranks of the group are in this process (the reference's mode, resource_manager.cc:392-422)
int p2p_connect_local(pplhip_ctx* c) {
    const int n = (int)c->ranks.size();
    for (int r = 0; r < n; ++r) {
        Rank& R = c->ranks[r];
        HIPCK(c, r, hipSetDevice(R.device));
        for (int g = 0; g < n; ++g) {
            Rank& Q = c->ranks[g];
            if (Q.device != R.device) {
                hipError_t e = hipDeviceEnablePeerAccess(Q.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return p2p_unavailable(c, r, std::string("peer access: ") + hipGetErrorString(e)); }
                (void)hipGetLastError();
            }
            R.peers.base[g] = Q.xbase;
        }
    }
    c->p2p_connected = true;
    return p2p_selftest(c);
}

// a kernel of the direct path gave up waiting for a peer: surfaced at the step's synchronisation points
int p2p_check(pplhip_ctx* c, int rank) {
    Rank& R = c->ranks[rank];
    if (R.p2p_status && *R.p2p_status == 64)
        return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, "stream hand-off timed out (compute / communication stream of a tensor-parallel step)");
    if (c->comm_mode == 2 && R.p2p_status && *R.p2p_status) {
        const uint32_t v = *R.p2p_status;
        return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, "direct collective timed out waiting for rank " + std::to_string((v - 1) & 15) +
                    ((v - 1) & 16 ? " (end barrier)" : " (start barrier)"));
    }
    return 0;
}

}  // namespace

static bool fuse_norm_active(const pplhip_ctx* c, const Rank& R);

extern "C" {

int pplhip_version(void) { return (1 << 16) | 1; }  // 1.1: pplhip_model_desc.act_quant_bit, comm_* and W8A8 operator entry points

int pplhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return PPLHIP_DEVICE_RUNTIME_ERROR;
    return n;
}

int pplhip_get_unique_id(void* out) {
    static_assert(sizeof(ncclUniqueId) <= PPLHIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return PPLHIP_DEVICE_RUNTIME_ERROR;
    memset(out, 0, PPLHIP_UNIQUE_ID_BYTES);
    memcpy(out, &id, sizeof(id));
    return 0;
}

int pplhip_build_rope_table(float* out, int32_t max_position, int32_t head_dim, float theta) {
    const int half = head_dim / 2;
    for (int p = 0; p < max_position; ++p)
        for (int i = 0; i < half; ++i) {
            const double freq = pow((double)theta, -2.0 * (double)i / (double)head_dim);
            const double a = (double)p * freq;
            out[(size_t)p * head_dim + i] = (float)cos(a);
            out[(size_t)p * head_dim + half + i] = (float)sin(a);
        }
    return 0;
}

void pplhip_destroy(pplhip_ctx* c) {
    if (!c) return;
    for (auto& R : c->ranks) {
        hipSetDevice(R.device);
        if (R.stream) hipStreamSynchronize(R.stream);
        if (R.comm_stream) hipStreamSynchronize(R.comm_stream);
        if (R.stream2) { hipStreamSynchronize(R.stream2); hipStreamDestroy(R.stream2); }
        if (R.ev_fork) hipEventDestroy(R.ev_fork);
        if (R.ev_join) hipEventDestroy(R.ev_join);
        if (R.comm2) ncclCommDestroy(R.comm2);
        if (R.comm) ncclCommDestroy(R.comm);
        if (R.comm_stream) hipStreamDestroy(R.comm_stream);
        for (int i = 0; i < 2; ++i) {
            if (R.ev_compute[i]) hipEventDestroy(R.ev_compute[i]);
            if (R.ev_comm[i]) hipEventDestroy(R.ev_comm[i]);
        }
        for (auto& g : R.graphs) if (g.second.exec) hipGraphExecDestroy(g.second.exec);
        for (auto& e : R.prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        for (auto& p : R.prof_free) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
        for (void* p : R.allocs) hipFree(p);
        for (int g = 0; g < P2P_MAX_RANKS; ++g)
            if (R.peer_ipc[g] && R.peers.base[g]) hipIpcCloseMemHandle(R.peers.base[g]);
        if (R.xbase) hipFree(R.xbase);
        if (R.p2p_status) hipHostFree(R.p2p_status);
        if (R.kv_cache) hipFree(R.kv_cache);
        if (R.kv_scale) hipFree(R.kv_scale);
        for (int i = 0; i < 2; ++i) {
            if (R.stage_host[i]) hipHostFree(R.stage_host[i]);
            if (R.stage_ev[i]) hipEventDestroy(R.stage_ev[i]);
        }
        if (R.pages_host) hipHostFree(R.pages_host);
        if (R.pages_dev) hipFree(R.pages_dev);
        if (R.h_rand) hipHostFree(R.h_rand);
        if (R.stream) hipStreamDestroy(R.stream);
    }
    delete c;
}

const char* pplhip_last_error(pplhip_ctx* c, int rank) {
    if (!c) return "null context";
    if (rank >= 0 && rank < (int)c->ranks.size() && !c->ranks[rank].err.empty()) return c->ranks[rank].err.c_str();
    return c->err.c_str();
}

int pplhip_init(const pplhip_model_desc* desc, const pplhip_opts* opts, pplhip_ctx** out) {
    if (!desc || !opts || !out) return PPLHIP_INVALID_VALUE;
    *out = nullptr;
    std::unique_ptr<pplhip_ctx> c(new pplhip_ctx());
    c->d = *desc;
    c->o = *opts;
    c->o.device_ids = nullptr;
    c->o.nccl_unique_id = nullptr;
    const int n = opts->n_local_ranks;
    const int tp = opts->world_size > 0 ? opts->world_size : n;
    c->tp = tp;
    const pplhip_model_desc& d = c->d;
    auto bad = [&](const char* m) { if (verbose()) fprintf(stderr, "[pplhip] init: %s\n", m); return PPLHIP_INVALID_VALUE; };
    if (n < 1 || tp < n || opts->rank_base < 0 || opts->rank_base + n > tp) return bad("rank layout");
    if (d.num_heads <= 0 || d.hidden_dim % d.num_heads) return bad("heads");
    if (d.num_heads % tp || d.num_kv_heads % tp || d.intermediate_dim % tp || d.vocab_size % tp) return bad("tp divisibility");
    if (d.num_heads % d.num_kv_heads) return bad("gqa");
    // src/generator/llm_generator.cc:114-144 (CheckParameters)
    if (!((d.cache_quant_bit == 8 && d.cache_quant_group == 8) || (d.cache_quant_bit == 0 && d.cache_quant_group == 1))) return bad("cache quant");
    if (d.cache_layout < 0 || d.cache_layout > 3 || d.cache_mode < 0 || d.cache_mode > 1) return bad("cache layout/mode");
    if (d.cache_mode == 1 && d.page_size <= 0) return bad("page_size");
    if (d.weight_quant_bit != 0 && d.weight_quant_bit != 8 && d.weight_quant_bit != 4) return bad("weight quant");
    if (d.act_quant_bit != 0 && !(d.act_quant_bit == 8 && d.weight_quant_bit == 8)) return bad("act quant (8 needs weight_quant_bit 8)");
    if (opts->max_running_batch <= 0 || opts->max_tokens_per_step <= 0 || d.max_position <= 0) return bad("limits");
    c->D = d.hidden_dim / d.num_heads;
    c->H = d.num_heads / tp;
    c->Hkv = d.num_kv_heads / tp;
    c->inter = d.intermediate_dim / tp;
    c->vocab_local = d.vocab_size / tp;
    if (c->D != 32 && c->D != 64 && c->D != 128) return bad("head_dim must be 32, 64 or 128");
    if (d.hidden_dim % 16 || c->inter % 16 || c->vocab_local % 4) return bad("dims alignment");
    if (d.weight_quant_bit == 4 && (d.weight_quant_group % 32 || d.hidden_dim % d.weight_quant_group ||
                                    (c->H * c->D) % d.weight_quant_group || c->inter % d.weight_quant_group)) return bad("w4 group");

    c->ranks.resize(n);
    pplhip_ctx* cp = c.get();
    std::vector<int> devs(n);
    for (int r = 0; r < n; ++r) devs[r] = opts->device_ids ? opts->device_ids[r] : r;
    if (!opts->device_ids) {  // PPLHIP_DEVICE_IDS=0,0,...: device of each local rank (tests: a whole group on one GPU)
        if (const char* e = getenv("PPLHIP_DEVICE_IDS")) {
            int r = 0;
            for (const char* q = e; *q && r < n; ++r) {
                devs[r] = atoi(q);
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
    }

    // communicators (replaces ppl::common::InitNccl, resource_manager.cc:393)
    // PPLHIP_FORCE_COMM=1: create the communicator and run every collective even at world size 1 (an identity) --
    // lets a single-GPU box exercise the RCCL call sequence, the communication stream and its event wiring
    const bool want_comm = tp > 1 || getenv("PPLHIP_FORCE_COMM") != nullptr;
    c->tp_on = want_comm;
    if (const char* e = getenv("PPLHIP_DECODE_GRAPH")) c->graph_on = atoi(e) != 0;
    if (const char* e = getenv("PPLHIP_TP_FUSE_NORM")) c->fuse_norm_want = atoi(e) != 0;
    c->emulate_tp = getenv("PPLHIP_EMULATE_TP") != nullptr;
    if (const char* e = getenv("PPLHIP_DUAL_STREAM")) c->dual_mode = atoi(e);
    int dual_min_env = 0;
    if (const char* e = getenv("PPLHIP_DUAL_MIN_ROWS")) dual_min_env = std::max(2, atoi(e));
    if (c->dual_mode < 0) {   // automatic: only where there is link time to hide
        c->dual_auto = tp > 1 && !getenv("PPLHIP_EMULATE_TP");
        c->dual_mode = c->dual_auto ? 1 : 0;
        if (c->dual_auto) { c->dual_min_rows = dual_min_env ? dual_min_env : 512; c->dual_max_rows = 1024; }
        // (no second stream or workspace for a context whose steps can never reach the window: memory, not safety -- the safety check is
        // the stream count against the hardware queues below)
        if (c->dual_auto && opts->max_running_batch < c->dual_min_rows) { c->dual_auto = false; c->dual_mode = 0; }
    }
    if (dual_min_env) c->dual_min_rows = dual_min_env;
    if (const char* e = getenv("PPLHIP_DUAL_MAX_ROWS")) c->dual_max_rows = atoi(e);
    if (const char* e = getenv("PPLHIP_DECODE_GRAPH_MAX_BATCH")) c->graph_max_batch = std::max(1, atoi(e));
    if (const char* e = getenv("PPLHIP_COMM")) c->comm_want = !strcmp(e, "rccl") ? 1 : (!strcmp(e, "p2p") ? 2 : 0);
    {   // bounded spins of the direct collectives: s_memrealtime runs at 100 MHz
        const char* e = getenv("PPLHIP_P2P_TIMEOUT_MS");
        c->p2p_timeout_ticks = (uint64_t)(e ? std::max(1, atoi(e)) : 20000) * 100000ull;
    }
    // several ranks on ONE device (tests: a whole tensor-parallel group emulated on a single GPU): RCCL refuses duplicate
    // devices, the direct collectives do not care where a peer's region lives
    bool dup_dev = false;
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) dup_dev |= devs[a] == devs[b];
    if (dup_dev && c->comm_want == 1) return fail(cp, -1, PPLHIP_INVALID_VALUE, "PPLHIP_COMM=rccl needs distinct devices");
    if (dup_dev) c->comm_want = 2;
    if (tp > P2P_MAX_RANKS || getenv("PPLHIP_EMULATE_TP") || tp < 2) {
        if (c->comm_want == 2 && want_comm) return fail(cp, -1, PPLHIP_INVALID_VALUE, "direct collectives need 2..8 connected ranks");
        c->comm_want = 1;
    }
    if (const char* e = getenv("PPLHIP_TP_OVERLAP")) c->tp_overlap = atoi(e) != 0;
    if (const char* e = getenv("PPLHIP_TP_HANDOFF")) c->handoff_flags = strcmp(e, "events") != 0;
    {   // Streams against hardware queues.  Kernels of this library WAIT for kernels on other streams of the same process (direct collectives
        // of ranks that share a device, the hand-off kernels of the two-chunk schedule, the halves of a two-stream step meeting in a
        // collective): the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4), and a
        // spinning kernel queued IN FRONT of the kernel it waits for on the same hardware queue never sees it run.  That is what round 5's
        // "stream hand-off timed out" was (a tp-8 group on ONE device with automatic two-stream decode: 8 x 3 streams on 24 queues).
        // Checked here as a resource, per device: streams this context will create there against the queues the runtime will give it.
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const int hwq = q && atoi(q) > 0 ? atoi(q) : 4;
        auto streams_per_rank = [&]() { return 1 + (c->dual_mode ? 1 : 0) + (c->tp_on ? 1 : 0); };
        int worst = 0;
        for (int a = 0; a < n; ++a) {
            int same = 0;
            for (int b = 0; b < n; ++b) same += devs[a] == devs[b];
            worst = std::max(worst, same);
        }
        if (c->dual_auto && worst * streams_per_rank() > hwq) {
            comm_note(c.get(), "two-stream decode off: " + std::to_string(worst) + " local rank(s) per device x " + std::to_string(streams_per_rank()) +
                                   " streams exceed " + std::to_string(hwq) + " hardware queues (GPU_MAX_HW_QUEUES)");
            c->dual_auto = false;
            c->dual_mode = 0;
        }
        if (c->tp_on && worst * streams_per_rank() > hwq) {
            if (c->handoff_flags && !getenv("PPLHIP_TP_HANDOFF")) {
                c->handoff_flags = false;   // events never spin
                comm_note(c.get(), "stream hand-offs by events, not device flags: " + std::to_string(worst * streams_per_rank()) + " streams on one device exceed " +
                                       std::to_string(hwq) + " hardware queues");
            }
            if (worst > 1)
                comm_note(c.get(), "warning: " + std::to_string(worst) + " ranks share a device with " + std::to_string(worst * streams_per_rank()) + " streams on " +
                                       std::to_string(hwq) + " hardware queues -- direct collectives between them can time out (set GPU_MAX_HW_QUEUES)");
        }
    }
    if (const char* e = getenv("PPLHIP_TP_OVERLAP_MIN_TOKENS")) c->tp_overlap_min_tokens = std::max(2, atoi(e));
    if (want_comm && c->comm_want != 2) {
        std::vector<ncclComm_t> comms(n);
        // PPLHIP_EMULATE_TP=1 (measurement only): this process holds n of the tp slices but the communicator spans only
        // those n, so one GPU can time the per-rank work of a tp-way step (collectives become local identities and the
        // gathered logits are incomplete)
        if ((tp == n && !opts->nccl_unique_id) || getenv("PPLHIP_EMULATE_TP")) {
            NCCLCK(cp, -1, ncclCommInitAll(comms.data(), n, devs.data()));
        } else {
            if (!opts->nccl_unique_id) return fail(cp, -1, PPLHIP_INVALID_VALUE, "nccl_unique_id required when world_size > n_local_ranks");
            ncclUniqueId id;
            memcpy(&id, opts->nccl_unique_id, sizeof(id));
            NCCLCK(cp, -1, ncclGroupStart());
            for (int r = 0; r < n; ++r) {
                HIPCK(cp, -1, hipSetDevice(devs[r]));
                NCCLCK(cp, -1, ncclCommInitRank(&comms[r], tp, id, opts->rank_base + r));
            }
            NCCLCK(cp, -1, ncclGroupEnd());
        }
        for (int r = 0; r < n; ++r) c->ranks[r].comm = comms[r];
        // a second communicator over the same ranks for the second stream of a two-stream decode step -- only when two-stream decode was
        // asked for explicitly (PPLHIP_DUAL_STREAM=1): the automatic mode runs on the direct collectives' second channel and never uses it
        // (run_launches), and two RCCL communicators on one device have never run on two devices
        if (c->dual_mode && !c->dual_auto) {
            std::vector<ncclComm_t> comms2(n, nullptr);
            ncclResult_t rc2 = ncclGroupStart();
            for (int r = 0; r < n && rc2 == ncclSuccess; ++r) rc2 = ncclCommSplit(comms[r], 0, opts->rank_base + r, &comms2[r], nullptr);
            const ncclResult_t rc3 = ncclGroupEnd();
            if (rc2 == ncclSuccess && rc3 == ncclSuccess) {
                for (int r = 0; r < n; ++r) c->ranks[r].comm2 = comms2[r];
            } else {
                // ladder: without a second communicator a two-stream step can still run on the direct collectives (their second channel);
                // on RCCL it falls back to one stream (run_launches checks R.comm2)
                for (int r = 0; r < n; ++r) if (comms2[r]) (void)ncclCommAbort(comms2[r]);   // partially created group
                comm_note(c.get(), std::string("second RCCL communicator unavailable (") + ncclGetErrorString(rc2 != ncclSuccess ? rc2 : rc3) +
                                       "): two-stream decode only on the direct collectives, else one stream");
            }
        }
        c->comm_mode = 1;
    }

    const int64_t cap_B = opts->max_running_batch;
    const int64_t cap_T = std::max<int64_t>(opts->max_tokens_per_step, opts->max_running_batch);
    const int hd = d.hidden_dim, q = d.weight_quant_bit, g = d.weight_quant_group;
    std::vector<float> rope((size_t)d.max_position * c->D);
    pplhip_build_rope_table(rope.data(), d.max_position, c->D, d.rope_theta);

    for (int r = 0; r < n; ++r) {
        Rank& R = c->ranks[r];
        R.device = devs[r];
        R.global_rank = opts->rank_base + r;
        HIPCK(cp, r, hipSetDevice(R.device));
        HIPCK(cp, r, hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
        if (c->dual_mode) {
            HIPCK(cp, r, hipStreamCreateWithFlags(&R.stream2, hipStreamNonBlocking));
            HIPCK(cp, r, hipEventCreateWithFlags(&R.ev_fork, hipEventDisableTiming));
            HIPCK(cp, r, hipEventCreateWithFlags(&R.ev_join, hipEventDisableTiming));
        }
        if (c->tp_on) {
            HIPCK(cp, r, hipStreamCreateWithFlags(&R.comm_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIPCK(cp, r, hipEventCreateWithFlags(&R.ev_compute[i], hipEventDisableTiming));
                HIPCK(cp, r, hipEventCreateWithFlags(&R.ev_comm[i], hipEventDisableTiming));
            }
        }
        int rc;
#define ALLOC(ptr, bytes) if ((rc = dev_alloc(cp, r, (void**)&(ptr), (uint64_t)(bytes)))) return rc
        ALLOC(R.embed, (uint64_t)d.vocab_size * hd * 2);
        ALLOC(R.norm, hd * 2);
        if ((rc = linear_alloc(cp, r, &R.output, c->vocab_local, hd, 0, 0))) return rc;
        R.layers.resize(d.num_layers);
        for (int l = 0; l < d.num_layers; ++l) {
            Layer& L = R.layers[l];
            ALLOC(L.attn_norm, hd * 2);
            ALLOC(L.ffn_norm, hd * 2);
            if ((rc = linear_alloc(cp, r, &L.wqkv, (c->H + 2 * c->Hkv) * c->D, hd, q, g))) return rc;
            if ((rc = linear_alloc(cp, r, &L.wo, hd, c->H * c->D, q, g))) return rc;
            if ((rc = linear_alloc(cp, r, &L.w13, 2 * c->inter, hd, q, g))) return rc;
            if ((rc = linear_alloc(cp, r, &L.w2, hd, c->inter, q, g, /*pad_k=*/true))) return rc;
        }
        ALLOC(R.rope, rope.size() * sizeof(float));
        HIPCK(cp, r, hipMemcpy(R.rope, rope.data(), rope.size() * sizeof(float), hipMemcpyHostToDevice));

        // step inputs
        R.cap_B = cap_B; R.cap_T = cap_T;
        const uint64_t step_elems = (uint64_t)cap_T + 4 * (cap_B + 1);
        for (int i = 0; i < 2; ++i) {
            HIPCK(cp, r, hipHostMalloc((void**)&R.stage_host[i], step_elems * 8, hipHostMallocDefault));
            HIPCK(cp, r, hipEventCreateWithFlags(&R.stage_ev[i], hipEventDisableTiming));
        }
        ALLOC(R.step_dev, step_elems * 8);

        // activations
        ALLOC(R.h, (uint64_t)cap_T * hd * 2);
        ALLOC(R.xn, (uint64_t)cap_T * hd * 2);
        ALLOC(R.qkv, (uint64_t)cap_T * (c->H + 2 * c->Hkv) * c->D * 2);
        ALLOC(R.att, (uint64_t)cap_T * c->H * c->D * 2);
        if (!c->tp_on) {
            ALLOC(R.part, (uint64_t)cap_T * hd * 2);
            ALLOC(R.part2, (uint64_t)cap_T * hd * 2);
        } else {
            // the buffers collectives touch live in ONE fine-grained allocation that peers map (k_comm.hip); RCCL works on them too
            auto up = [](size_t v) { return (v + 4095) / 4096 * 4096; };
            R.x_part = P2P_DATA_START;
            R.x_part2 = R.x_part + up((size_t)cap_T * hd * 2);
            // a rank's reduced slice: <= half the buffer (n >= 2) -- as elements (plain all-reduce) or as whole rows (fused norm: ceil(T / 2) rows)
            const size_t slice_bytes = up(((size_t)cap_T / 2 + 1) * hd * 2 + 64);
            R.x_scratch[0] = R.x_part2 + up((size_t)cap_T * hd * 2);
            R.x_scratch[1] = R.x_scratch[0] + slice_bytes;
            R.x_local = R.x_scratch[1] + slice_bytes;
            if (c->dual_mode) {
                R.x_scratch2[0] = R.x_local;
                R.x_scratch2[1] = R.x_scratch2[0] + slice_bytes;
                R.x_local = R.x_scratch2[1] + slice_bytes;
            }
            R.xbytes = R.x_local + up((size_t)cap_B * c->vocab_local * 4);
            // FINE-GRAINED device memory: the kind HIP defines as coherent between devices at system scope (what RCCL puts its
            // peer buffers in).  hipDeviceMallocUncached is NOT usable here: with several ranks on one MI355X a kernel reading
            // a buffer that the previous kernel of another stream had just written through an uncached mapping saw stale data
            // (profiles/probes/tp_matrix.sh: 8 bad steps of 36 and aborted runs, against 0 of 96 for fine-grained and plain
            // memory).  PPLHIP_P2P_XALLOC=0 (plain) / 1 (uncached) exist for that probe only.
            static const int xalloc = getenv("PPLHIP_P2P_XALLOC") ? atoi(getenv("PPLHIP_P2P_XALLOC")) : 2;
            hipError_t e = xalloc == 0 ? hipMalloc((void**)&R.xbase, R.xbytes)
                                       : hipExtMallocWithFlags((void**)&R.xbase, R.xbytes, xalloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            if (e != hipSuccess) return fail(cp, r, PPLHIP_OUT_OF_MEMORY, std::string("exchange region: ") + hipGetErrorString(e));
            HIPCK(cp, r, hipMemset(R.xbase, 0, P2P_DATA_START));
            R.part = (uint16_t*)(R.xbase + R.x_part);
            R.part2 = (uint16_t*)(R.xbase + R.x_part2);
            R.logits_local = (float*)(R.xbase + R.x_local);
            if (R.comm) ALLOC(R.logits_gather, (uint64_t)cap_B * d.vocab_size * 4);  // RCCL all-gather target
            HIPCK(cp, r, hipHostMalloc((void**)&R.p2p_status, 64, hipHostMallocMapped));
            *R.p2p_status = 0;
            ALLOC(R.d_flag, 64);
            ALLOC(R.hflags, 64);
            HIPCK(cp, r, hipMemset(R.hflags, 0, 64));
        }
        const int inter_p = (d.weight_quant_bit != 4) ? (c->inter + k_tile(d) - 1) / k_tile(d) * k_tile(d) : c->inter;  // = layers[*].w2.Kp
        if (d.act_quant_bit == 8) {  // online_i8i8: the int8 copy of whatever feeds the next linear + its per-token scales
            ALLOC(R.xq, (uint64_t)cap_T * std::max(std::max(hd, c->H * c->D), inter_p));
            ALLOC(R.sx, (uint64_t)cap_T * 4);
        }
        ALLOC(R.act, (uint64_t)cap_T * inter_p * 2);
        if (inter_p != c->inter) HIPCK(cp, r, hipMemset(R.act, 0, (uint64_t)cap_T * inter_p * 2));  // pad columns stay zero
        ALLOC(R.hn, (uint64_t)cap_B * hd * 2);
        ALLOC(R.logits, (uint64_t)cap_B * d.vocab_size * 4);
        R.attn_ws_bytes = attn_decode_workspace_bytes(cap_B, c->H, c->D, 32);
        ALLOC(R.attn_ws, R.attn_ws_bytes);
        R.gemm_ws_bytes = (size_t)8 * 256 * (size_t)std::max(std::max(2 * c->inter, (c->H + 2 * c->Hkv) * c->D), std::max(hd, c->vocab_local)) * sizeof(float);
        R.gemm_ws_bytes = std::max<size_t>(R.gemm_ws_bytes, (size_t)96 << 20);
        ALLOC(R.gemm_ws, R.gemm_ws_bytes);
        if (c->dual_mode) ALLOC(R.gemm_ws2, R.gemm_ws_bytes);

        if (r == 0) {  // sampler lives on local rank 0 (src/backends/cuda/resource_manager.cc:315-327)
            ALLOC(R.d_temp, cap_B * 4); ALLOC(R.d_topp, cap_B * 4); ALLOC(R.d_rand, cap_B * 4);
            ALLOC(R.d_lp, cap_B * 4); ALLOC(R.d_tokout, cap_B * 4);
            HIPCK(cp, r, hipHostMalloc((void**)&R.h_rand, cap_B * 4, hipHostMallocDefault));
            if (opts->enable_penalty) {
                ALLOC(R.count_map, (uint64_t)cap_B * d.vocab_size * 2);
                ALLOC(R.d_slots, cap_B * 8);
                ALLOC(R.d_rep, cap_B * 4); ALLOC(R.d_pres, cap_B * 4); ALLOC(R.d_freq, cap_B * 4); ALLOC(R.d_ptemp, cap_B * 4);
                HIPCK(cp, r, hipMemsetAsync(R.count_map, 0, (uint64_t)cap_B * d.vocab_size * 2, R.stream));
            }
        }
#undef ALLOC
        HIPCK(cp, r, hipDeviceSynchronize());
    }
    // every rank of the group lives in this process: map the peers' regions now and try the direct collectives
    if (c->tp_on && c->comm_want != 1 && tp == n) {
        int rc = p2p_connect_local(cp);
        if (rc) return rc;
    }
    *out = c.release();
    return 0;
}

int pplhip_comm_export(pplhip_ctx* c, int rank, void* handle_out) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !handle_out) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    memset(handle_out, 0, PPLHIP_IPC_HANDLE_BYTES);
    if (!R.xbase) return fail(c, rank, PPLHIP_INVALID_VALUE, "no exchange region (tensor parallelism is off)");
    static_assert(sizeof(hipIpcMemHandle_t) <= PPLHIP_IPC_HANDLE_BYTES, "ipc handle size");
    HIPCK(c, rank, hipSetDevice(R.device));
    hipIpcMemHandle_t h;
    HIPCK(c, rank, hipIpcGetMemHandle(&h, R.xbase));
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int pplhip_comm_connect(pplhip_ctx* c, const void* all_handles) {
    if (!c || !all_handles) return PPLHIP_INVALID_VALUE;
    if (!c->tp_on || c->tp > P2P_MAX_RANKS) return fail(c, -1, PPLHIP_INVALID_VALUE, "nothing to connect");
    if (c->comm_want == 1) return 0;  // PPLHIP_COMM=rccl
    if (c->p2p_connected) return fail(c, -1, PPLHIP_INVALID_VALUE, "already connected");
    const int n = (int)c->ranks.size(), base = c->o.rank_base;
    for (int r = 0; r < n; ++r) {
        Rank& R = c->ranks[r];
        HIPCK(c, r, hipSetDevice(R.device));
        for (int g = 0; g < c->tp; ++g) {
            if (g >= base && g < base + n) {  // a rank of this process
                Rank& Q = c->ranks[g - base];
                if (Q.device != R.device) {
                    hipError_t e = hipDeviceEnablePeerAccess(Q.device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                        (void)hipGetLastError();
                        const std::string why = std::string("peer access: ") + hipGetErrorString(e);
                        return p2p_selftest(c, &why);
                    }
                    (void)hipGetLastError();
                }
                R.peers.base[g] = Q.xbase;
                continue;
            }
            hipIpcMemHandle_t h;
            memcpy(&h, (const char*)all_handles + (size_t)g * PPLHIP_IPC_HANDLE_BYTES, sizeof(h));
            void* ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                const std::string why = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e);
                return p2p_selftest(c, &why);
            }
            R.peers.base[g] = (char*)ptr;
            R.peer_ipc[g] = true;
        }
    }
    c->p2p_connected = true;
    return p2p_selftest(c);
}

int pplhip_comm_mode(pplhip_ctx* c) { return c ? c->comm_mode : PPLHIP_INVALID_VALUE; }
int pplhip_comm_fused_norm(pplhip_ctx* c) { return c && !c->ranks.empty() ? (fuse_norm_active(c, c->ranks[0]) ? 1 : 0) : PPLHIP_INVALID_VALUE; }

/* ------------------------------------------------------------------------------------------------ weights */

int pplhip_rank_set_tensor(pplhip_ctx* c, int rank, const char* name, const void* data, uint64_t bytes) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !name || !data) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    void* p; uint64_t b;
    Linear* lin = nullptr;
    if (!find_tensor(c, R, name, &p, &b, &lin)) return fail(c, rank, PPLHIP_NOT_FOUND, std::string("unknown tensor ") + name);
    const char* w13 = strstr(name, "feed_forward.w13.");
    if (lin && lin->qbit == 8 && c->d.act_quant_bit == 8 && bytes == (uint64_t)lin->N * lin->K * 2) {
        // online_i8i8 "online" half: an fp16 [N, K] matrix for an int8 linear is quantised per output row on the device
        HIPCK(c, rank, hipSetDevice(R.device));
        void *tmp = nullptr, *tmp2 = nullptr;
        HIPCK(c, rank, hipMalloc(&tmp, bytes));
        hipError_t e = hipMemcpy(tmp, data, bytes, hipMemcpyHostToDevice);
        const uint16_t* src = (const uint16_t*)tmp;
        if (e == hipSuccess && w13) {  // rows interleaved (gate_i, up_i) first: the per-row quantisation moves with the row
            e = hipMalloc(&tmp2, bytes);
            if (e == hipSuccess) e = launch_interleave_rows(R.stream, tmp, tmp2, lin->N, (int64_t)lin->K * 2);
            src = (const uint16_t*)tmp2;
        }
        if (e == hipSuccess) e = launch_quant_weight(R.stream, src, lin->N, lin->K, (int8_t*)lin->w, lin->Kp, lin->scale);
        if (e == hipSuccess) e = hipStreamSynchronize(R.stream);
        hipFree(tmp);
        if (tmp2) hipFree(tmp2);
        if (e != hipSuccess) return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, std::string("online weight quantisation: ") + hipGetErrorString(e));
        return 0;
    }
    if (b != bytes) return fail(c, rank, PPLHIP_INVALID_VALUE, std::string("tensor ") + name + ": got " + std::to_string(bytes) + " bytes, want " + std::to_string(b));
    HIPCK(c, rank, hipSetDevice(R.device));
    if (w13) {
        // the container stores w13 as [gate rows | up rows]; on the device the rows are interleaved (gate_i, up_i) so
        // that the GEMM epilogue can apply SwiGLU (kernels.h: launch_linear swiglu)
        void* tmp = nullptr;
        HIPCK(c, rank, hipMalloc(&tmp, bytes));
        hipError_t e = hipMemcpy(tmp, data, bytes, hipMemcpyHostToDevice);
        const int rows = 2 * c->inter;
        if (e == hipSuccess) e = launch_interleave_rows(R.stream, tmp, p, rows, (int64_t)(bytes / rows));
        if (e == hipSuccess) e = hipStreamSynchronize(R.stream);
        hipFree(tmp);
        if (e != hipSuccess) return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, std::string("interleave w13: ") + hipGetErrorString(e));
        return 0;
    }
    if (lin && lin->Kp != lin->K) {  // zero-padded rows (linear_alloc cleared the buffer)
        const size_t rb = (size_t)lin->elt_bytes(lin->K), rbp = (size_t)lin->elt_bytes(lin->Kp);
        HIPCK(c, rank, hipMemcpy2D(p, rbp, data, rb, rb, lin->N, hipMemcpyHostToDevice));
        return 0;
    }
    HIPCK(c, rank, hipMemcpy(p, data, bytes, hipMemcpyHostToDevice));
    return 0;
}

// container: "PPLHIPW1" | u32 count | count x { u32 name_len | name | u64 nbytes | pad to 64 | data }
int pplhip_rank_load(pplhip_ctx* c, int rank, const char* slice_dir) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !slice_dir) return PPLHIP_INVALID_VALUE;
    const std::string path = std::string(slice_dir) + "/weights.pplhip";
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return fail(c, rank, PPLHIP_NOT_FOUND, "cannot open " + path);
    char magic[8];
    uint32_t count = 0;
    int rc = 0;
    std::vector<char> buf;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "PPLHIPW1", 8) || fread(&count, 4, 1, f) != 1) {
        fclose(f);
        return fail(c, rank, PPLHIP_INVALID_VALUE, "bad header in " + path);
    }
    for (uint32_t i = 0; i < count && !rc; ++i) {
        uint32_t nl = 0; uint64_t nb = 0;
        char name[256];
        if (fread(&nl, 4, 1, f) != 1 || nl >= sizeof(name) || fread(name, 1, nl, f) != nl || fread(&nb, 8, 1, f) != 1) { rc = PPLHIP_INVALID_VALUE; break; }
        name[nl] = 0;
        long pos = ftell(f);
        long pad = (64 - pos % 64) % 64;
        fseek(f, pad, SEEK_CUR);
        buf.resize(nb);
        if (fread(buf.data(), 1, nb, f) != nb) { rc = PPLHIP_INVALID_VALUE; break; }
        rc = pplhip_rank_set_tensor(c, rank, name, buf.data(), nb);
    }
    fclose(f);
    if (rc == PPLHIP_INVALID_VALUE && c->ranks[rank].err.empty()) fail(c, rank, rc, "truncated container " + path);
    return rc;
}

int pplhip_rank_init_synthetic(pplhip_ctx* c, int rank, uint64_t seed) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    HIPCK(c, rank, hipSetDevice(R.device));
    const int hd = c->d.hidden_dim;
    const uint32_t st = 1u + (uint32_t)R.global_rank;
    const float AMP = 0.034641016f;
    auto tid = [](int layer, int slot) { return (uint32_t)((layer + 1) * 32 + slot); };
    hipStream_t s = R.stream;
    auto lin = [&](Linear& l, int layer, int wslot) -> hipError_t {
        hipError_t e;
        if (l.qbit == 0) return launch_synth_fill(s, 0, seed, tid(layer, wslot), st, AMP, (uint64_t)l.N * l.K, l.w);
        if (l.qbit == 8) {
            if ((e = launch_synth_fill(s, 1, seed, tid(layer, wslot), st, 0.f, (uint64_t)l.N * l.K, l.w)) != hipSuccess) return e;
            return launch_synth_fill(s, 3, seed, tid(layer, wslot + 1), st, AMP / 127.0f, (uint64_t)l.N, l.scale);
        }
        if ((e = launch_synth_fill(s, 2, seed, tid(layer, wslot), st, 0.f, (uint64_t)l.N * l.K / 2, l.w)) != hipSuccess) return e;
        return launch_synth_fill(s, 3, seed, tid(layer, wslot + 1), st, AMP / 7.0f, (uint64_t)l.N * (l.K / l.group), l.scale);
    };
    HIPCK(c, rank, launch_synth_fill(s, 0, seed, tid(-1, 0), 0, 1.0f, (uint64_t)c->d.vocab_size * hd, R.embed));
    HIPCK(c, rank, launch_synth_fill(s, 4, seed, tid(-1, 11), 0, 0.f, hd, R.norm));
    HIPCK(c, rank, launch_synth_fill(s, 0, seed, tid(-1, 12), st, AMP, (uint64_t)c->vocab_local * hd, R.output.w));
    for (int l = 0; l < c->d.num_layers; ++l) {
        Layer& L = R.layers[l];
        HIPCK(c, rank, launch_synth_fill(s, 4, seed, tid(l, 1), 0, 0.f, hd, L.attn_norm));
        HIPCK(c, rank, launch_synth_fill(s, 4, seed, tid(l, 6), 0, 0.f, hd, L.ffn_norm));
        HIPCK(c, rank, lin(L.wqkv, l, 2));
        HIPCK(c, rank, lin(L.wo, l, 4));
        {   // generated in container order, then row-interleaved like an uploaded tensor
            Linear tmp = L.w13;
            HIPCK(c, rank, hipMalloc(&tmp.w, L.w13.w_bytes()));
            if (L.w13.s_bytes()) HIPCK(c, rank, hipMalloc((void**)&tmp.scale, L.w13.s_bytes()));
            HIPCK(c, rank, lin(tmp, l, 7));
            HIPCK(c, rank, launch_interleave_rows(s, tmp.w, L.w13.w, L.w13.N, (int64_t)(L.w13.w_bytes() / L.w13.N)));
            if (L.w13.s_bytes()) HIPCK(c, rank, launch_interleave_rows(s, tmp.scale, L.w13.scale, L.w13.N, (int64_t)(L.w13.s_bytes() / L.w13.N)));
            HIPCK(c, rank, hipStreamSynchronize(s));
            hipFree(tmp.w);
            if (tmp.scale) hipFree(tmp.scale);
        }
        if (L.w2.Kp == L.w2.K) {
            HIPCK(c, rank, lin(L.w2, l, 9));
        } else {  // generated contiguous, then copied into the zero-padded rows
            Linear tmp = L.w2;
            HIPCK(c, rank, hipMalloc(&tmp.w, L.w2.w_bytes()));
            HIPCK(c, rank, lin(tmp, l, 9));  // scale goes straight to L.w2.scale (same pointer)
            const size_t rb = (size_t)L.w2.elt_bytes(L.w2.K), rbp = (size_t)L.w2.elt_bytes(L.w2.Kp);
            HIPCK(c, rank, hipMemcpy2DAsync(L.w2.w, rbp, tmp.w, rb, rb, L.w2.N, hipMemcpyDeviceToDevice, s));
            HIPCK(c, rank, hipStreamSynchronize(s));
            hipFree(tmp.w);
        }
    }
    HIPCK(c, rank, hipStreamSynchronize(s));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ KV slab */

// A DECISIVE synthetic model: the embedding table is regenerated at amplitude `embed_amp` (the synthetic default is 1; behind 32 synthetic
// layers the residual stream has an rms of ~10, which buries a unit-amplitude embedding: profiles/r05_tie_probe.log) and
// output.weight[v] := tok_embeddings.weight[(v - shift) mod vocab] on this rank's vocabulary shard.  The final hidden state of a position
// then keeps a strong component along its own token's embedding, so token t is answered by t + shift with a top-2 margin of a third or more of the
// logit scale -- far above what two evaluation orders of the same arithmetic differ by.  For harness checks that compare ANSWERS of two
// runs (benchmark_prefix_cache_offline --synthetic-decisive-head: cold run vs prefix-cache hits); what such a check pins is the
// bookkeeping of the path (pages, start positions, hand-over of tokens), not its arithmetic.
int pplhip_rank_tie_output(pplhip_ctx* c, int rank, int64_t shift, uint64_t seed, float embed_amp) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    if (R.output.qbit != 0) return fail(c, rank, PPLHIP_INVALID_VALUE, "tie_output: the lm_head is fp16");
    HIPCK(c, rank, hipSetDevice(R.device));
    const int64_t V = c->d.vocab_size, hd = c->d.hidden_dim, v0 = (int64_t)R.global_rank * c->vocab_local;
    if (embed_amp > 0.f) HIPCK(c, rank, launch_synth_fill(R.stream, 0, seed, (uint32_t)(0 * 32 + 0), 0, embed_amp, (uint64_t)V * hd, R.embed));   // tensor id of pplhip_rank_init_synthetic's embedding
    shift = ((shift % V) + V) % V;
    for (int64_t v = 0; v < c->vocab_local;) {          // at most two contiguous runs
        const int64_t src = ((v0 + v - shift) % V + V) % V;
        const int64_t n = std::min<int64_t>(c->vocab_local - v, V - src);
        HIPCK(c, rank, hipMemcpyAsync((char*)R.output.w + v * hd * 2, (const char*)R.embed + src * hd * 2, (size_t)n * hd * 2, hipMemcpyDeviceToDevice, R.stream));
        v += n;
    }
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    return 0;
}

int pplhip_kv_block_bytes(pplhip_ctx* c, uint64_t* cache_bytes, uint64_t* scale_bytes) {
    if (!c) return PPLHIP_INVALID_VALUE;
    // src/backends/cuda/resource_manager.cc:381-387
    const uint64_t elt = c->d.cache_quant_bit == 8 ? 1 : 2;
    const uint64_t kb = (uint64_t)c->d.num_layers * 2 * c->Hkv * c->D * elt;
    const uint64_t sb = c->d.cache_quant_bit > 0 ? (uint64_t)c->d.num_layers * 2 * c->Hkv * (c->D / c->d.cache_quant_group) * 2 : 0;
    if (cache_bytes) *cache_bytes = kb;
    if (scale_bytes) *scale_bytes = sb;
    return 0;
}

int pplhip_kv_capacity(pplhip_ctx* c, float max_tokens_scale, uint64_t* tokens) {
    if (!c || !tokens) return PPLHIP_INVALID_VALUE;
    uint64_t kb, sb;
    pplhip_kv_block_bytes(c, &kb, &sb);
    HIPCK(c, 0, hipSetDevice(c->ranks[0].device));
    size_t free_b = 0, total = 0;
    HIPCK(c, 0, hipMemGetInfo(&free_b, &total));
    // resource_manager.cc:330-341
    const uint64_t kv_cache_max_bytes = (uint64_t)((double)max_tokens_scale * (double)free_b * (double)kb / (double)(kb + sb));
    *tokens = kv_cache_max_bytes / kb;
    return 0;
}

int pplhip_kv_alloc(pplhip_ctx* c, int rank, uint64_t tokens) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || tokens == 0) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    HIPCK(c, rank, hipSetDevice(R.device));
    uint64_t kb, sb;
    pplhip_kv_block_bytes(c, &kb, &sb);
    if (R.kv_cache) { hipFree(R.kv_cache); R.kv_cache = nullptr; }
    if (R.kv_scale) { hipFree(R.kv_scale); R.kv_scale = nullptr; }
    hipError_t e = hipMalloc(&R.kv_cache, tokens * kb);
    if (e != hipSuccess) return fail(c, rank, PPLHIP_OUT_OF_MEMORY, "alloc kv cache [" + std::to_string(tokens * kb) + "] failed: " + hipGetErrorString(e));
    if (sb) {
        e = hipMalloc((void**)&R.kv_scale, tokens * sb);
        if (e != hipSuccess) {
            hipFree(R.kv_cache); R.kv_cache = nullptr;
            return fail(c, rank, PPLHIP_OUT_OF_MEMORY, "alloc kv scale [" + std::to_string(tokens * sb) + "] failed: " + hipGetErrorString(e));
        }
    }
    // deterministic contents for never-written slots (the reference's cudaMalloc leaves them undefined)
    // stream-ordered on the rank's own (non-blocking) stream: a null-stream memset is NOT ordered against it and could
    // land after the first steps have written the slab
    HIPCK(c, rank, hipMemsetAsync(R.kv_cache, 0, tokens * kb, R.stream));
    if (sb) HIPCK(c, rank, hipMemsetAsync(R.kv_scale, 0, tokens * sb, R.stream));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    R.kv_tokens = tokens;
    return 0;
}

int pplhip_kv_ptrs(pplhip_ctx* c, int rank, void** kv_cache_mem, void** kv_scale_mem) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    if (kv_cache_mem) *kv_cache_mem = c->ranks[rank].kv_cache;
    if (kv_scale_mem) *kv_scale_mem = c->ranks[rank].kv_scale;
    return 0;
}

static int kv_rw(pplhip_ctx* c, int rank, int which, uint64_t offset, void* host, uint64_t bytes, bool read) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !host) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    uint64_t kb, sb;
    pplhip_kv_block_bytes(c, &kb, &sb);
    char* base = which ? (char*)R.kv_scale : (char*)R.kv_cache;
    const uint64_t total = R.kv_tokens * (which ? sb : kb);
    if (!base || offset + bytes > total) return fail(c, rank, PPLHIP_INVALID_VALUE, "kv read/write out of range");
    HIPCK(c, rank, hipSetDevice(R.device));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    if (read) HIPCK(c, rank, hipMemcpy(host, base + offset, bytes, hipMemcpyDeviceToHost));
    else HIPCK(c, rank, hipMemcpy(base + offset, host, bytes, hipMemcpyHostToDevice));
    return 0;
}
int pplhip_kv_read(pplhip_ctx* c, int rank, int which, uint64_t offset, void* dst, uint64_t bytes) {
    return kv_rw(c, rank, which, offset, dst, bytes, true);
}
int pplhip_kv_write(pplhip_ctx* c, int rank, int which, uint64_t offset, const void* src, uint64_t bytes) {
    return kv_rw(c, rank, which, offset, const_cast<void*>(src), bytes, false);
}

int pplhip_kv_fill_synthetic(pplhip_ctx* c, int rank, uint64_t seed) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    if (!R.kv_cache) return fail(c, rank, PPLHIP_INVALID_VALUE, "kv slab not allocated");
    HIPCK(c, rank, hipSetDevice(R.device));
    uint64_t kb, sb;
    pplhip_kv_block_bytes(c, &kb, &sb);
    const uint32_t st = 100u + (uint32_t)R.global_rank;
    if (c->d.cache_quant_bit == 8) {
        HIPCK(c, rank, launch_synth_fill(R.stream, 1, seed, 1, st, 0.f, R.kv_tokens * kb, R.kv_cache));
        HIPCK(c, rank, launch_synth_fill(R.stream, 3, seed, 2, st, 0.02f, R.kv_tokens * sb / 2, R.kv_scale));
    } else {
        HIPCK(c, rank, launch_synth_fill(R.stream, 0, seed, 1, st, 1.5f, R.kv_tokens * kb / 2, R.kv_cache));
    }
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ the step */

int pplhip_set_inputs(pplhip_ctx* c, int rank, const pplhip_step* st) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !st) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    const int64_t B = st->batch, T = st->num_tokens;
    if (B < 0 || T < 0 || B > R.cap_B || T > R.cap_T)
        return fail(c, rank, PPLHIP_INVALID_VALUE, "step exceeds max_running_batch / max_tokens_per_step");
    if (B == 0) {  // an empty step is legal (LLMGenerator never sends one, llm_generator.cc:658-660) and does nothing
        if (T != 0) return fail(c, rank, PPLHIP_INVALID_VALUE, "tokens without requests");
        R.B = 0; R.T = 0; R.decoding_batches = 0; R.h_seq = nullptr;
        return 0;
    }
    if (!st->token_inputs || !st->seq_starts || !st->kv_starts || !st->start_pos) return PPLHIP_INVALID_VALUE;
    if (st->max_kv_len > c->d.max_position) return fail(c, rank, PPLHIP_INVALID_VALUE, "max_kv_len exceeds max_position");
    {   // the kernels index the rope table, the embedding table and the KV slab with these values: reject anything that
        // would leave them (O(B + T) on the host; the reference trusts its generator, this boundary does not)
        const int64_t* ss = st->seq_starts;
        if (ss[0] != 0 || ss[B] != T) return fail(c, rank, PPLHIP_INVALID_VALUE, "seq_starts must run from 0 to num_tokens");
        if (st->decoding_batches < 0 || st->decoding_batches > B) return fail(c, rank, PPLHIP_INVALID_VALUE, "decoding_batches out of range");
        const int64_t P = c->d.page_size > 0 ? c->d.page_size : 1;
        for (int64_t b = 0; b < B; ++b) {
            const int64_t len = ss[b + 1] - ss[b], sp = st->start_pos[b];
            if (len <= 0) return fail(c, rank, PPLHIP_INVALID_VALUE, "request " + std::to_string(b) + " has no tokens");
            if (b < st->decoding_batches && len != 1) return fail(c, rank, PPLHIP_INVALID_VALUE, "decode request " + std::to_string(b) + " must have one token");
            if (sp < 0 || sp + len > c->d.max_position) return fail(c, rank, PPLHIP_INVALID_VALUE, "request " + std::to_string(b) + " exceeds max_position");
            if (c->d.cache_mode == 0) {
                const int64_t ci = st->cache_indices ? st->cache_indices[b] : -1;
                if (ci < 0 || (uint64_t)(ci + sp + len) > R.kv_tokens)
                    return fail(c, rank, PPLHIP_INVALID_VALUE, "request " + std::to_string(b) + ": KV range outside the slab");
            } else if (st->req_list_changed && st->cache_indices && st->max_pages > 0) {
                const int64_t need = (sp + len + P - 1) / P;
                if (need > st->max_pages) return fail(c, rank, PPLHIP_INVALID_VALUE, "request " + std::to_string(b) + ": page list too short");
                for (int64_t pg = 0; pg < need; ++pg) {
                    const int64_t id = st->cache_indices[b * st->max_pages + pg];
                    if (id < 0 || (uint64_t)id >= R.kv_tokens / (uint64_t)P)  // (also catches the INT64_MAX padding)
                        return fail(c, rank, PPLHIP_INVALID_VALUE, "request " + std::to_string(b) + ": page outside the slab");
                }
            }
        }
        for (int64_t t = 0; t < T; ++t)
            if (st->token_inputs[t] < 0 || st->token_inputs[t] >= c->d.vocab_size)
                return fail(c, rank, PPLHIP_INVALID_VALUE, "token id outside the vocabulary at row " + std::to_string(t));
    }
    HIPCK(c, rank, hipSetDevice(R.device));
    const int cur = R.stage_cur;
    R.stage_cur ^= 1;
    HIPCK(c, rank, hipEventSynchronize(R.stage_ev[cur]));  // the copy that last used this staging buffer is done
    int64_t* hbuf = R.stage_host[cur];
    // packed layout: token_ids[T] | seq_starts[B+1] | kv_starts[B+1] | start_pos[B] | cache_indices[B] (mode 0)
    int64_t off = 0;
    memcpy(hbuf + off, st->token_inputs, T * 8); R.d_tok = R.step_dev + off; off += T;
    memcpy(hbuf + off, st->seq_starts, (B + 1) * 8); R.d_seq = R.step_dev + off; R.h_seq = hbuf + off; off += B + 1;
    memcpy(hbuf + off, st->kv_starts, (B + 1) * 8); R.d_kvs = R.step_dev + off; off += B + 1;
    memcpy(hbuf + off, st->start_pos, B * 8); R.d_sp = R.step_dev + off; off += B;
    if (c->d.cache_mode == 0) {
        if (B > 0 && !st->cache_indices) return PPLHIP_INVALID_VALUE;
        memcpy(hbuf + off, st->cache_indices, B * 8); R.d_ci = R.step_dev + off; off += B;
    }
    HIPCK(c, rank, hipMemcpyAsync(R.step_dev, hbuf, off * 8, hipMemcpyHostToDevice, R.stream));
    HIPCK(c, rank, hipEventRecord(R.stage_ev[cur], R.stream));
    if (c->d.cache_mode == 1 && st->req_list_changed) {  // src/engine/llm_engine.cc:67-71
        if (!st->cache_indices || st->max_pages <= 0) return fail(c, rank, PPLHIP_INVALID_VALUE, "page list missing");
        const uint64_t n = (uint64_t)B * st->max_pages;
        if (n > R.pages_cap) {
            HIPCK(c, rank, hipStreamSynchronize(R.stream));
            if (R.pages_dev) hipFree(R.pages_dev);
            if (R.pages_host) hipHostFree(R.pages_host);
            R.pages_cap = std::max<uint64_t>(n * 2, 4096);
            HIPCK(c, rank, hipMalloc((void**)&R.pages_dev, R.pages_cap * 8));
            HIPCK(c, rank, hipHostMalloc((void**)&R.pages_host, R.pages_cap * 8, hipHostMallocDefault));
        } else {
            HIPCK(c, rank, hipStreamSynchronize(R.stream));  // previous async copy from pages_host
        }
        memcpy(R.pages_host, st->cache_indices, n * 8);
        HIPCK(c, rank, hipMemcpyAsync(R.pages_dev, R.pages_host, n * 8, hipMemcpyHostToDevice, R.stream));
        R.max_pages = st->max_pages;
    }
    if (c->d.cache_mode == 1) R.d_ci = R.pages_dev;
    R.B = B; R.T = T;
    R.decoding_batches = st->decoding_batches;
    R.max_seq_len = st->max_seq_len;
    R.max_kv_len = st->max_kv_len;
    return 0;
}

// A chunk of a step: requests [b0, b0 + bn) = token rows [t0, t0 + tn); the first nd requests of the chunk are
// decode rows.  A step runs as ONE chunk, or -- tensor parallel, see pplhip_run -- as two that alternate between the
// compute stream and the communication stream.
struct Chunk {
    int64_t b0, bn, t0, tn, nd;
};

// one layer linear over a chunk: fp16 activations straight into the (weight-only quantised) GEMM, or -- online_i8i8 -- quantised
// per token first and multiplied in int8.  x rows have stride l.Kp.
// pre_quantised: R.xq / R.sx already hold the rows (written by the RMSNorm in front of wqkv / w13).
static int layer_linear(pplhip_ctx* c, int rank, const Linear& l, const uint16_t* x, int64_t M, void* y, int64_t ldy, bool swiglu,
                        bool pre_quantised = false, SplitSlabs* defer = nullptr) {
    Rank& R = c->ranks[rank];
    if (c->d.act_quant_bit == 8) {
        if (!pre_quantised) HIPCK(c, rank, launch_quant_act(R.stream, x, M, l.Kp, l.Kp, R.xq, l.Kp, R.sx));
        HIPCK(c, rank, launch_linear_i8(R.stream, R.xq, R.sx, (const int8_t*)l.w, l.scale, M, l.N, l.Kp, y, ldy, false, swiglu));
        return 0;
    }
    HIPCK(c, rank, launch_linear(R.stream, x, l.w, l.scale, l.qbit, l.group, M, l.N, l.Kp, y, ldy, false, R.gemm_ws, R.gemm_ws_bytes, swiglu,
                                 (R.defer_reduce || (R.defer_qkv && defer == &R.sl_qkv)) ? defer : nullptr));
    return 0;
}

// attention block of layer l for one chunk: (Skip)RMSNorm -> wqkv -> RoPE + KV write -> attention -> wo (partial sums)
// xn_ready: the fused collective of the previous half-layer already left this block's normalised input in R.xn (and the residual in R.h)
static int layer_attention_part(pplhip_ctx* c, int rank, int l, const Chunk& k, const uint16_t* pending, int split, int threads, bool xn_ready = false) {
    Rank& R = c->ranks[rank];
    const pplhip_model_desc& d = c->d;
    hipStream_t s = R.stream;
    const int hd = d.hidden_dim, H = c->H, Hkv = c->Hkv, D = c->D;
    const int nqkv = (H + 2 * Hkv) * D;
    Layer& L = R.layers[l];
    ProfEvent ev;
    uint16_t* h = R.h + k.t0 * hd;
    uint16_t* xn = R.xn + k.t0 * hd;
    const bool a8 = d.act_quant_bit == 8;  // the norm writes the int8 operand of the next linear directly (no fp16 xn, no separate pass)
    {
        if (!xn_ready)
            HIPCK(c, rank, launch_rmsnorm(s, h, pending ? pending + k.t0 * hd : nullptr, L.attn_norm, d.norm_eps, k.tn, hd, nullptr, xn,
                                          pending ? h : nullptr, a8 ? R.xq : nullptr, a8 ? R.sx : nullptr, pending ? &R.sl_part2 : nullptr));
        R.sl_part2 = SplitSlabs{};
        prof_begin(c, R, PPLHIP_PROF_GEMM, &ev);
        { int rc = layer_linear(c, rank, L.wqkv, xn, k.tn, R.qkv + k.t0 * nqkv, L.wqkv.N, false, a8, &R.sl_qkv); if (rc) return rc; }
        prof_end(R, &ev);
    }
    const KvAddr kv = make_kv_addr(d, Hkv, D, R.kv_tokens, R.kv_cache, R.kv_scale, l);
    HIPCK(c, rank, launch_rope_kv_write(s, R.qkv, R.rope, kv, d.cache_quant_bit, d.cache_quant_group, R.d_seq, R.d_sp, R.d_ci,
                                        R.max_pages, R.B, k.t0, k.tn, H, Hkv, D, &R.sl_qkv));
    R.sl_qkv = SplitSlabs{};
    // decode rows of the chunk: per-request arrays shifted to the chunk (q rows stay absolute through seq_starts);
    // prefill requests: absolute request range
    const int64_t ci_stride = d.cache_mode == 1 ? R.max_pages : 1;
    if (k.nd > 0) {
        if (c->o.enable_profiling == 2) {
            // light profiling: the launch carries its own start / stop events (timestamps of the dispatch packet, no barrier
            // packets on the stream; with split-K the reduce kernel is not included)
            std::pair<hipEvent_t, hipEvent_t> p;
            if (!R.prof_free.empty()) { p = R.prof_free.back(); R.prof_free.pop_back(); }
            else { hipEventCreate(&p.first); hipEventCreate(&p.second); }
            ev.cls = PPLHIP_PROF_ATTN_DECODE; ev.a = p.first; ev.b = p.second;
            HIPCK(c, rank, launch_attn_decode(s, R.qkv, kv, d.cache_quant_bit, R.d_seq + k.b0, R.d_sp + k.b0, R.d_ci + k.b0 * ci_stride,
                                              R.max_pages, k.nd, H, Hkv, D, R.max_kv_len, split, threads, R.attn_ws,
                                              R.att + k.b0 * (int64_t)H * D, ev.a, ev.b));
            R.prof.push_back(ev);
        } else {
            prof_begin(c, R, PPLHIP_PROF_ATTN_DECODE, &ev);
            HIPCK(c, rank, launch_attn_decode(s, R.qkv, kv, d.cache_quant_bit, R.d_seq + k.b0, R.d_sp + k.b0, R.d_ci + k.b0 * ci_stride,
                                              R.max_pages, k.nd, H, Hkv, D, R.max_kv_len, split, threads, R.attn_ws,
                                              R.att + k.b0 * (int64_t)H * D));
            prof_end(R, &ev);
        }
    }
    if (k.bn > k.nd) {
        prof_begin(c, R, PPLHIP_PROF_ATTN_PREFILL, &ev);
        // (decode requests own one token row each, so the prefill requests' rows are [t0 + nd, t0 + tn): the split-KV form for short suffixes)
        HIPCK(c, rank, launch_attn_prefill(s, R.qkv, kv, d.cache_quant_bit, R.d_seq, R.d_sp, R.d_ci, R.max_pages, k.b0 + k.nd,
                                           k.b0 + k.bn, H, Hkv, D, R.max_seq_len, R.att, R.max_kv_len, R.attn_ws, R.attn_ws_bytes,
                                           k.t0 + k.nd, k.tn - k.nd));
        prof_end(R, &ev);
    }
    prof_begin(c, R, PPLHIP_PROF_GEMM, &ev);
    {
        int rc = layer_linear(c, rank, L.wo, R.att + k.t0 * (int64_t)H * D, k.tn, R.part + k.t0 * hd, hd, false, false, &R.sl_part);
        if (rc) return rc;
    }
    prof_end(R, &ev);
    return 0;
}

// feed-forward block of layer l for one chunk: SkipRMSNorm -> w13 with fused SwiGLU (K3 + K10) -> w2 (partial sums)
static int layer_ffn_part(pplhip_ctx* c, int rank, int l, const Chunk& k, bool xn_ready = false) {
    Rank& R = c->ranks[rank];
    const pplhip_model_desc& d = c->d;
    hipStream_t s = R.stream;
    const int hd = d.hidden_dim;
    Layer& L = R.layers[l];
    ProfEvent ev;
    uint16_t* h = R.h + k.t0 * hd;
    uint16_t* xn = R.xn + k.t0 * hd;
    uint16_t* act = R.act + k.t0 * (int64_t)L.w2.Kp;
    const bool a8 = d.act_quant_bit == 8;
    if (!xn_ready)
        HIPCK(c, rank, launch_rmsnorm(s, h, R.part + k.t0 * hd, L.ffn_norm, d.norm_eps, k.tn, hd, nullptr, xn, h, a8 ? R.xq : nullptr,
                                      a8 ? R.sx : nullptr, &R.sl_part));
    R.sl_part = SplitSlabs{};
    prof_begin(c, R, PPLHIP_PROF_GEMM, &ev);
    { int rc = layer_linear(c, rank, L.w13, xn, k.tn, act, L.w2.Kp, /*swiglu=*/true, a8); if (rc) return rc; }
    prof_end(R, &ev);
    prof_begin(c, R, PPLHIP_PROF_GEMM, &ev);
    { int rc = layer_linear(c, rank, L.w2, act, k.tn, R.part2 + k.t0 * hd, hd, false, false, R.keep_part2 ? nullptr : &R.sl_part2); if (rc) return rc; }
    prof_end(R, &ev);
    return 0;
}

// the sequence-parallel residual stream is in force for this rank's steps (pplhip_ctx::fuse_norm_want)
static bool fuse_norm_active(const pplhip_ctx* c, const Rank& R) {
    const int hd = c->d.hidden_dim;
    return c->fuse_norm_want && c->tp_on && c->tp > 1 && c->d.act_quant_bit != 8 && !R.dump_dev && hd % 8 == 0 && hd <= P2P_NORM_MAX_HIDDEN &&
           (c->comm_mode == 2 || c->emulate_tp);
}

// all-reduce(sum) of the chunk's rows of `buf` ([T, hidden] fp16).  Overlapped mode: on the communication stream,
// after the compute stream's work so far; the compute stream picks the result up through R.ev_comm[ci] later.
// norm_w != NULL (fuse_norm_active): the collective also folds the reduced rows into the residual stream and normalises them with norm_w --
// afterwards R.xn holds rmsnorm(h + sum) * norm_w for every row of the chunk and R.h the new residual on the rows this rank owns.
static int chunk_allreduce(pplhip_ctx* c, int rank, uint16_t* buf, const Chunk& k, int ci, bool overlapped, const uint16_t* norm_w = nullptr) {
    Rank& R = c->ranks[rank];
    const int hd = c->d.hidden_dim;
    uint16_t* p = buf + k.t0 * hd;
    auto reduce_on = [&](hipStream_t st) -> int {
        if (norm_w && c->comm_mode == 2) {
            const bool ch1 = R.channel == 1;
            HIPCK(c, rank, launch_p2p_allreduce_norm(st, R.peers, R.global_rank, c->tp, (size_t)((char*)p - R.xbase),
                                                     ch1 ? R.x_scratch2[R.ar_count2++ & 1] : R.x_scratch[R.ar_count++ & 1], k.tn, hd, R.h + k.t0 * hd, norm_w,
                                                     c->d.norm_eps, R.xn + k.t0 * hd, ch1 ? ++R.p2p_epoch2 : ++R.p2p_epoch, c->p2p_timeout_ticks, R.p2p_status,
                                                     ch1 ? 1 : 0));
            return 0;
        }
        if (norm_w) {
            // one rank's slice without its peers (bench.py --emulate-tp): the link part of the collective is skipped as before; what stays is
            // the rank's OWN share of the fused kernel's arithmetic -- residual add + norm of the rows it owns (the other rows of xn keep
            // whatever they held: the emulated step's values are meaningless anyway, its timing is the point)
            const int64_t per = (k.tn + c->tp - 1) / c->tp, lo = std::min<int64_t>(per * R.global_rank, k.tn), nown = std::min<int64_t>(per, k.tn - lo);
            uint16_t* h = R.h + (k.t0 + lo) * hd;
            if (nown > 0) HIPCK(c, rank, launch_rmsnorm(st, h, p + lo * hd, norm_w, c->d.norm_eps, nown, hd, nullptr, R.xn + (k.t0 + lo) * hd, h));
            return 0;
        }
        if (c->comm_mode == 2) {
            if (R.channel == 1)
                HIPCK(c, rank, launch_p2p_allreduce(st, R.peers, R.global_rank, c->tp, (size_t)((char*)p - R.xbase), R.x_scratch2[R.ar_count2++ & 1],
                                                    k.tn * hd, ++R.p2p_epoch2, c->p2p_timeout_ticks, R.p2p_status, 1));
            else
                HIPCK(c, rank, launch_p2p_allreduce(st, R.peers, R.global_rank, c->tp, (size_t)((char*)p - R.xbase), R.x_scratch[R.ar_count++ & 1],
                                                    k.tn * hd, ++R.p2p_epoch, c->p2p_timeout_ticks, R.p2p_status));
        } else if (R.comm) {
            NCCLCK(c, rank, ncclAllReduce(p, p, (size_t)k.tn * hd, ncclFloat16, ncclSum, R.channel == 1 ? R.comm2 : R.comm, st));
        }
        return 0;
    };
    static const int dbg = getenv("PPLHIP_TP_DEBUG") ? atoi(getenv("PPLHIP_TP_DEBUG")) : 0;
    if (!overlapped || (dbg & 2)) return reduce_on(R.stream);
    if (c->handoff_flags) {
        // compute stream: "chunk ci's partial sums are complete"; communication stream: wait for that, reduce, "chunk ci is reduced"
        HIPCK(c, rank, launch_handoff_signal(R.stream, R.hflags + ci, ++R.h_ready_ep[ci]));
        const bool identity = c->comm_mode != 2 && !R.comm;  // ranks emulated on one device (bench.py --emulate-tp): nothing to reduce
        HIPCK(c, rank, launch_handoff_wait(R.comm_stream, R.hflags + ci, R.h_ready_ep[ci], identity ? R.hflags + 2 + ci : nullptr,
                                           identity ? R.h_done_ep[ci] + 1 : 0, c->p2p_timeout_ticks, R.p2p_status));
        ++R.h_done_ep[ci];
        if (!identity) {
            if (int rc = reduce_on(R.comm_stream)) return rc;
            HIPCK(c, rank, launch_handoff_signal(R.comm_stream, R.hflags + 2 + ci, R.h_done_ep[ci]));
        }
        return 0;
    }
    if (dbg & 1) {  // diagnosis: never re-record an event inside a step
        hipEventCreateWithFlags(&R.ev_compute[ci], hipEventDisableTiming);
        hipEventCreateWithFlags(&R.ev_comm[ci], hipEventDisableTiming);
    }
    HIPCK(c, rank, hipEventRecord(R.ev_compute[ci], R.stream));
    HIPCK(c, rank, hipStreamWaitEvent(R.comm_stream, R.ev_compute[ci], 0));
    if (int rc = reduce_on(R.comm_stream)) return rc;
    HIPCK(c, rank, hipEventRecord(R.ev_comm[ci], R.comm_stream));
    return 0;
}

// the compute stream picks up chunk ci's reduced rows
static int wait_reduced(pplhip_ctx* c, int rank, int ci) {
    Rank& R = c->ranks[rank];
    if (c->handoff_flags)
        HIPCK(c, rank, launch_handoff_wait(R.stream, R.hflags + 2 + ci, R.h_done_ep[ci], nullptr, 0, c->p2p_timeout_ticks, R.p2p_status));
    else
        HIPCK(c, rank, hipStreamWaitEvent(R.stream, R.ev_comm[ci], 0));
    return 0;
}

// the launches of one step on the rank's stream(s)
static int run_launches(pplhip_ctx* c, int rank) {
    Rank& R = c->ranks[rank];
    const pplhip_model_desc& d = c->d;
    hipStream_t s = R.stream;
    const int64_t T = R.T, B = R.B;
    const int hd = d.hidden_dim;
    const int64_t nb_decode = std::min<int64_t>(std::max<int64_t>(R.decoding_batches, 0), B);
    // (round 4: 128-thread blocks for many short requests looked 3-6 % faster in the operator micro-benchmark and measured 2 % SLOWER inside
    // the model step, interleaved A/B: 0.885 vs 0.868 ms per launch at batch 1024 / kv 520 -- not adopted)
    const int threads = c->o.decoding_attn_tpb == 512 ? 512 : 256;
    const bool comm = c->tp_on;

    // Tensor parallel: the step is cut at a request boundary into two chunks of about T/2 rows.  While RCCL reduces
    // the partial sums of one chunk on the communication stream, the compute stream runs the other chunk's block
    // (attention part or FFN part), so the xGMI time hides under the matmuls instead of adding to them.
    Chunk ck[2];
    int nck = 1;
    ck[0] = Chunk{0, B, 0, T, nb_decode};
    if (comm && c->tp_overlap && R.h_seq && T >= c->tp_overlap_min_tokens && B >= 2) {
        int64_t bm = 1;
        while (bm < B - 1 && R.h_seq[bm] < T / 2) ++bm;  // first request boundary at or after row T/2
        if (nb_decode == B) {                             // pure decode: whole 128-row GEMM tiles in the first chunk
            const int64_t r = (T / 2 + 127) / 128 * 128;
            bm = r < B ? r : B / 2;
        }
        const int64_t tm = R.h_seq[bm];
        ck[0] = Chunk{0, bm, 0, tm, std::min(nb_decode, bm)};
        ck[1] = Chunk{bm, B - bm, tm, T - tm, std::max<int64_t>(0, nb_decode - bm)};
        nck = 2;
    }
    const bool ov = nck == 2;
    // Two-stream decode (PPLHIP_DUAL_STREAM=1): a pure-decode step of dual_min_rows..dual_max_rows rows is cut into two half-batches that run
    // their layers on two streams, side by side.  At these sizes no kernel of the step fills the chip -- the linears are 20-256 tiles with
    // short K loops, the attention launch a few hundred blocks -- so the halves' kernels interleave on the CUs instead of queueing behind each
    // other's ramps and tails, and one half's HBM-bound attention runs beside the other's matrix work.  The halves share nothing but the
    // weights and the KV slab: rows [t0, t0 + tn) of every activation buffer, a split-K workspace and an attention workspace each.  Results
    // are those of the same rows run as a step of their own.  Under tensor parallelism every half issues its all-reduces on its OWN stream
    // and channel (direct collectives: a second flag set / scratch pair / epoch counter, k_comm.hip; RCCL: a second communicator over the
    // same ranks) -- the all-reduce of one half runs beside the other half's matmuls with no hand-off kernels and no extra launches, which
    // is what the chunked schedule above pays +3.5 ms per 1024-row step for.  Not with int8 activations (shared operand buffer), residual
    // dumps or graph capture.
    const bool identity_comm = comm && c->comm_mode != 2 && !R.comm;   // ranks emulated on one device (bench.py --emulate-tp)
    bool dual = false;
    if (c->dual_mode && R.stream2 && !ov && nb_decode == B && T == B && B >= c->dual_min_rows && B <= c->dual_max_rows && B >= 2 &&
        (!comm || identity_comm || c->comm_mode == 2 || (R.comm && R.comm2 && !c->dual_auto)) && d.act_quant_bit != 8 && !R.dump_dev) {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(R.stream, &cst);
        if (cst == hipStreamCaptureStatusNone) {
            const int64_t bm = B >= 64 ? (B / 2 + 15) / 16 * 16 : B / 2;   // whole 16-row activation sub-tiles in the first half
            ck[0] = Chunk{0, bm, 0, bm, bm};
            ck[1] = Chunk{bm, B - bm, bm, B - bm, B - bm};
            nck = 2;
            dual = true;
            static const bool verbose_ = verbose();
            if (verbose_ && !R.dual_seen) {
                R.dual_seen = true;
                fprintf(stderr, "[pplhip] rank %d: two-stream decode (rows %lld + %lld)\n", rank, (long long)bm, (long long)(B - bm));
            }
        }
    }
    // the second half's launches go out with its own stream, workspaces and deferred-slab state in the rank's working fields
    struct HalfState { SplitSlabs sl_qkv, sl_part, sl_part2; } half2;
    const int64_t aws_off = dual ? ck[1].b0 * (int64_t)c->H * 32 * (c->D + 2) : 0;   // floats: past every row the first half can use
    auto enter2 = [&]() {
        std::swap(R.stream, R.stream2); std::swap(R.gemm_ws, R.gemm_ws2);
        R.attn_ws += aws_off; R.attn_ws_bytes -= (size_t)aws_off * 4;
        R.channel = 1;
        std::swap(R.sl_qkv, half2.sl_qkv); std::swap(R.sl_part, half2.sl_part); std::swap(R.sl_part2, half2.sl_part2);
    };
    auto leave2 = [&]() {
        std::swap(R.stream, R.stream2); std::swap(R.gemm_ws, R.gemm_ws2);
        R.attn_ws -= aws_off; R.attn_ws_bytes += (size_t)aws_off * 4;
        R.channel = 0;
        std::swap(R.sl_qkv, half2.sl_qkv); std::swap(R.sl_part, half2.sl_part); std::swap(R.sl_part2, half2.sl_part2);
    };
    struct Half2Guard {   // (an error return between enter2 and leave2 must not leave the rank on the second stream)
        decltype(leave2)& l; bool in = false;
        ~Half2Guard() { if (in) l(); }
    } g2{leave2};
    static const bool tpdbg2 = getenv("PPLHIP_TP_DEBUG") && (atoi(getenv("PPLHIP_TP_DEBUG")) & 2);  // diagnosis: chunks, but collectives in stream
    int split[2] = {1, 1};
    for (int i = 0; i < nck; ++i) split[i] = ck[i].nd > 0 ? decode_split(c, ck[i].nd, R.max_kv_len) : 1;

    const bool fuse = fuse_norm_active(c, R);
    static const int defer_on = getenv("PPLHIP_DEFER_REDUCE") ? atoi(getenv("PPLHIP_DEFER_REDUCE")) : 1;
    R.defer_reduce = defer_on && !comm && d.act_quant_bit != 8 && !R.dump_dev && (nck == 1 || dual);
    R.defer_qkv = defer_on && d.act_quant_bit != 8 && !R.dump_dev && (nck == 1 || dual);   // (wo / w2 feed the all-reduce: their slabs are summed first)
    R.keep_part2 = false;
    R.sl_qkv = R.sl_part = R.sl_part2 = SplitSlabs{};
    ProfEvent ev_run, ev;
    prof_begin(c, R, PPLHIP_PROF_RUN, &ev_run);
    HIPCK(c, rank, launch_embedding(s, R.d_tok, R.embed, T, hd, R.h));
    const uint16_t* pending = nullptr;
    int rc;
    const size_t dump_n = (size_t)T * hd;  // elements of one dumped matrix
    if (R.dump_dev) HIPCK(c, rank, hipMemcpyAsync(R.dump_dev, R.h, dump_n * 2, hipMemcpyDeviceToDevice, s));
    if (dual) {   // the second stream starts behind the embedding (and with it behind the step's inputs and the previous step)
        HIPCK(c, rank, hipEventRecord(R.ev_fork, s));
        HIPCK(c, rank, hipStreamWaitEvent(R.stream2, R.ev_fork, 0));
    }
    for (int l = 0; l < d.num_layers; ++l) {
        // (the last layer's w2 writes part2 in both halves: the final norm gathers rows of both on the first stream)
        R.keep_part2 = dual && l == d.num_layers - 1;
        for (int i = 0; i < nck; ++i) {
            if (ov && l > 0 && !tpdbg2 && (rc = wait_reduced(c, rank, i))) return rc;  // part2 rows of chunk i are reduced
            if (dual && i == 1) { enter2(); g2.in = true; }
            if ((rc = layer_attention_part(c, rank, l, ck[i], pending, split[i], threads, fuse && l > 0))) return rc;
            if (comm && (rc = chunk_allreduce(c, rank, R.part, ck[i], i, ov, fuse ? R.layers[l].ffn_norm : nullptr))) return rc;
            if (g2.in) { leave2(); g2.in = false; }
        }
        for (int i = 0; i < nck; ++i) {
            if (ov && !tpdbg2 && (rc = wait_reduced(c, rank, i))) return rc;            // part rows of chunk i are reduced
            if (dual && i == 1) { enter2(); g2.in = true; }
            if ((rc = layer_ffn_part(c, rank, l, ck[i], fuse))) return rc;
            // (fused: the collective behind w2 normalises for the NEXT consumer -- the next layer's attention norm, or the final norm)
            if (comm && (rc = chunk_allreduce(c, rank, R.part2, ck[i], i, ov, !fuse ? nullptr : (l + 1 < d.num_layers ? R.layers[l + 1].attn_norm : R.norm)))) return rc;
            if (g2.in) { leave2(); g2.in = false; }
        }
        pending = fuse ? nullptr : R.part2;
        if (R.dump_dev) {
            if (ov && !tpdbg2) for (int i = 0; i < nck; ++i) if ((rc = wait_reduced(c, rank, i))) return rc;
            HIPCK(c, rank, hipMemcpyAsync(R.dump_dev + (size_t)(l + 1) * 2 * dump_n, R.h, dump_n * 2, hipMemcpyDeviceToDevice, s));
            HIPCK(c, rank, hipMemcpyAsync(R.dump_dev + ((size_t)(l + 1) * 2 + 1) * dump_n, R.part2, dump_n * 2, hipMemcpyDeviceToDevice, s));
        }
    }
    if (ov && !tpdbg2) for (int i = 0; i < nck; ++i) if ((rc = wait_reduced(c, rank, i))) return rc;
    R.keep_part2 = false;
    if (dual) {   // the first stream goes on behind the second half's last layer
        HIPCK(c, rank, hipEventRecord(R.ev_join, R.stream2));
        HIPCK(c, rank, hipStreamWaitEvent(s, R.ev_join, 0));
    }
    // K11: last-token gather + final (Skip)RMSNorm (the last FFN output is folded into the residual of the gathered
    // rows only) + lm_head (+ all-gather of the vocab shards)
    const uint16_t* hn = R.hn;
    if (fuse && d.num_layers > 0) {
        // the last collective already applied the final norm to every row: only the last-token gather is left (a pure-decode step's rows
        // ARE the last tokens, in order)
        if (T == B) hn = R.xn;
        else HIPCK(c, rank, launch_gather_last_rows(s, R.xn, R.d_seq, B, hd, R.hn));
    } else {
        HIPCK(c, rank, launch_rmsnorm(s, R.h, pending, R.norm, d.norm_eps, B, hd, R.d_seq, R.hn, nullptr, nullptr, nullptr, pending ? &R.sl_part2 : nullptr));
    }
    R.sl_part2 = SplitSlabs{};
    prof_begin(c, R, PPLHIP_PROF_GEMM, &ev);
    if (!comm) {
        HIPCK(c, rank, launch_linear(s, hn, R.output.w, nullptr, 0, 0, B, R.output.N, hd, R.logits, d.vocab_size, true, R.gemm_ws, R.gemm_ws_bytes));
        prof_end(R, &ev);
    } else {
        const int vl = c->vocab_local;
        HIPCK(c, rank, launch_linear(s, hn, R.output.w, nullptr, 0, 0, B, vl, hd, R.logits_local, vl, true, R.gemm_ws, R.gemm_ws_bytes));
        prof_end(R, &ev);
        // every collective of this communicator is issued on ONE stream (the communication stream when overlapping)
        hipStream_t cs = (ov && !tpdbg2) ? R.comm_stream : s;
        if (ov && !tpdbg2) {
            if (c->handoff_flags) {
                HIPCK(c, rank, launch_handoff_signal(s, R.hflags + 0, ++R.h_ready_ep[0]));
                HIPCK(c, rank, launch_handoff_wait(cs, R.hflags + 0, R.h_ready_ep[0], nullptr, 0, c->p2p_timeout_ticks, R.p2p_status));
            } else {
                HIPCK(c, rank, hipEventRecord(R.ev_compute[0], s));
                HIPCK(c, rank, hipStreamWaitEvent(cs, R.ev_compute[0], 0));
            }
        }
        if (c->comm_mode == 2) {  // every rank pulls every shard straight into its [B, V] logits
            HIPCK(c, rank, launch_p2p_allgather(cs, R.peers, R.global_rank, c->tp, R.x_local, R.logits, B, (int64_t)vl * 4,
                                                (int64_t)d.vocab_size * 4, ++R.p2p_epoch, c->p2p_timeout_ticks, R.p2p_status));
        } else if (R.comm) {
            NCCLCK(c, rank, ncclAllGather(R.logits_local, R.logits_gather, (size_t)B * vl, ncclFloat32, R.comm, cs));
        } else {  // world size 1 without a communicator cannot happen (tp_on implies one of the two)
            return fail(c, rank, PPLHIP_OTHER_ERROR, "tensor-parallel step without collectives");
        }
        if (ov && !tpdbg2) {
            if (c->handoff_flags) {
                HIPCK(c, rank, launch_handoff_signal(cs, R.hflags + 2, ++R.h_done_ep[0]));
                if ((rc = wait_reduced(c, rank, 0))) return rc;
            } else {
                HIPCK(c, rank, hipEventRecord(R.ev_comm[0], cs));
                HIPCK(c, rank, hipStreamWaitEvent(s, R.ev_comm[0], 0));
            }
        }
        for (int r = 0; r < c->tp && c->comm_mode != 2; ++r)
            HIPCK(c, rank, hipMemcpy2DAsync(R.logits + (size_t)r * vl, (size_t)d.vocab_size * 4, R.logits_gather + (size_t)r * B * vl,
                                            (size_t)vl * 4, (size_t)vl * 4, B, hipMemcpyDeviceToDevice, s));
    }
    prof_end(R, &ev_run);
    return 0;
}

// HIP-graph replay of pure-decode steps.  Everything a decode step launches depends only on (batch, attention split, page-table
// width): token ids, positions, cache slots and page lists are read from the step buffer, whose layout is fixed by the batch size.
// So the ~290 launches of a step whose shape was seen before can be replayed as one graph launch (the second occurrence of a
// shape captures it).  Measured (profiles/small_batch_latency.py, 7B W8A16, kv 512): batch 1 3.30 ms replayed vs 3.26 ms eager,
// batch 64 6.08 vs 6.07 -- the step is GPU-bound (the kernels already run back to back; what a small batch loses is the ramp-up
// of 290 short kernels, which a graph does not remove), so the replay only saves host time and stays opt-in
// (PPLHIP_DECODE_GRAPH=1).  Round 4, with the step at 2.3 ms: batch 1 / 2 2.33 / 2.53 -> 2.28 / 2.48 ms replayed, nothing from batch 4 up;
// still opt-in (a paged cache changes the page-table width, hence the captured shape, every page_size steps).
// Never under tensor parallelism (collectives on a second stream) or profiling.
static int run_decode_graph(pplhip_ctx* c, int rank, bool* done) {
    Rank& R = c->ranks[rank];
    *done = false;
    const int64_t B = R.B;
    const int64_t nb_decode = std::min<int64_t>(std::max<int64_t>(R.decoding_batches, 0), B);
    if (!c->graph_on || c->tp_on || c->o.enable_profiling || R.T != B || nb_decode != B || B > c->graph_max_batch) return 0;
    const uint64_t key = (uint64_t)B | ((uint64_t)decode_split(c, B, R.max_kv_len) << 24) | ((uint64_t)R.max_pages << 32);
    auto it = R.graphs.find(key);
    if (it == R.graphs.end()) {  // first sight: run eagerly (one-time function attributes, lazily loaded code objects)
        if (R.graphs.size() >= 32) {  // drop the least recently used shape
            auto old = R.graphs.begin();
            for (auto j = R.graphs.begin(); j != R.graphs.end(); ++j) if (j->second.tick < old->second.tick) old = j;
            if (old->second.exec) hipGraphExecDestroy(old->second.exec);
            R.graphs.erase(old);
        }
        R.graphs[key] = GraphEntry{nullptr, ++R.graph_tick};
        return 0;
    }
    it->second.tick = ++R.graph_tick;
    if (!it->second.exec) {
        hipGraph_t g = nullptr;
        HIPCK(c, rank, hipStreamBeginCapture(R.stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_launches(c, rank);
        const hipError_t e = hipStreamEndCapture(R.stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) { (void)hipGetLastError(); c->graph_on = false; return 0; }  // capture unsupported here: eager from now on
        hipGraphExec_t ex = nullptr;
        const hipError_t e2 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e2 != hipSuccess || !ex) { (void)hipGetLastError(); c->graph_on = false; return 0; }
        it->second.exec = ex;
        if (verbose()) fprintf(stderr, "[pplhip] rank %d: decode step captured as a HIP graph (batch %lld, split %d, page-table width %lld)\n",
                                              rank, (long long)B, (int)((key >> 24) & 0xff), (long long)R.max_pages);
    }
    HIPCK(c, rank, hipGraphLaunch(it->second.exec, R.stream));
    *done = true;
    return 0;
}

int pplhip_comm_info(pplhip_ctx* c, int64_t rows, pplhip_comm_info_t* out) {
    if (!c || !out) return PPLHIP_INVALID_VALUE;
    memset(out, 0, sizeof(*out));
    out->mode = c->comm_mode;
    out->selftest = c->selftest;
    out->has_rccl = !c->ranks.empty() && c->ranks[0].comm != nullptr;
    out->dual_min_rows = c->dual_mode ? c->dual_min_rows : 0;
    out->dual_max_rows = c->dual_mode ? c->dual_max_rows : 0;
    // the schedule run_launches picks for a pure-decode step of `rows` rows (int8 activations, residual dumps and graph capture aside)
    const Rank& R = c->ranks[0];
    const bool two_chunks = c->tp_on && c->tp_overlap && rows >= c->tp_overlap_min_tokens && rows >= 2;
    const bool two_streams = !two_chunks && c->dual_mode && R.stream2 && rows >= c->dual_min_rows && rows <= c->dual_max_rows && rows >= 2 &&
                             (!c->tp_on || c->comm_mode == 2 || !R.comm || (R.comm && R.comm2 && !c->dual_auto)) && c->d.act_quant_bit != 8;
    out->schedule = two_chunks ? 2 : (two_streams ? 1 : 0);
    snprintf(out->notes, sizeof(out->notes), "%s", c->comm_notes.c_str());
    return 0;
}

// one all-reduce of fp16 [rows, hidden], `iters` times back to back on the rank's stream, between two events.  Collective: every rank
// of the group calls it with the same arguments.  path 0: the collectives in use; 1: RCCL (when a communicator exists).
// The call enqueues AND synchronises: a context that holds several local ranks must call it from one thread per rank at the same time (like
// pplhip_run under ParallelExecute) -- rank by rank from one thread would wait for peers that were never enqueued.
int pplhip_comm_allreduce_us(pplhip_ctx* c, int rank, int64_t rows, int32_t iters, int32_t path, float* us) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !us || iters < 1 || rows < 1) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    *us = -1.f;
    if (!c->tp_on || rows > R.cap_T) return PPLHIP_INVALID_VALUE;
    if (path == 1 && !R.comm) return 0;              // no RCCL communicator: nothing to compare with
    if (path == 0 && c->comm_mode != 2 && !R.comm) return 0;
    HIPCK(c, rank, hipSetDevice(R.device));
    struct Ev {   // destroyed on every return path
        hipEvent_t e = nullptr;
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    } ev0, ev1;
    HIPCK(c, rank, hipEventCreate(&ev0.e));
    HIPCK(c, rank, hipEventCreate(&ev1.e));
    hipEvent_t e0 = ev0.e, e1 = ev1.e;
    const int hd = c->d.hidden_dim;
    const Chunk k{0, rows, 0, rows, rows};
    for (int it = -2; it < iters; ++it) {            // two warm-up rounds
        if (it == 0) HIPCK(c, rank, hipEventRecord(e0, R.stream));
        if (path == 1) NCCLCK(c, rank, ncclAllReduce(R.part, R.part, (size_t)rows * hd, ncclFloat16, ncclSum, R.comm, R.stream));
        else if (int rc = chunk_allreduce(c, rank, R.part, k, 0, false, fuse_norm_active(c, R) ? R.norm : nullptr)) return rc;   // what a step issues
    }
    HIPCK(c, rank, hipEventRecord(e1, R.stream));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    float ms = 0.f;
    HIPCK(c, rank, hipEventElapsedTime(&ms, e0, e1));
    *us = ms * 1e3f / (float)iters;
    if (R.p2p_status && *R.p2p_status) return fail(c, rank, PPLHIP_DEVICE_RUNTIME_ERROR, "a direct collective timed out during pplhip_comm_allreduce_us");
    return 0;
}

int pplhip_run(pplhip_ctx* c, int rank, int cache_prefill) {
    (void)cache_prefill;  // K6 and K7 are one kernel here: attention always reads K/V back from the slab
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    if (!R.kv_cache) return fail(c, rank, PPLHIP_INVALID_VALUE, "kv slab not allocated");
    HIPCK(c, rank, hipSetDevice(R.device));
    if (R.B == 0) return 0;
    bool done = false;
    if (int rc = run_decode_graph(c, rank, &done)) return rc;
    return done ? 0 : run_launches(c, rank);
}

// Diagnosis entry point (tests bisect a logits difference per layer with it): pplhip_run with the residual stream captured after
// every layer, in the oracle's hidden_dump convention (oracle/llama_ref.c ref_forward): out[0] = embeddings, out[l + 1] =
// fp16(h + row-parallel FFN output of layer l) as fp32, [L + 1, T, hidden].  Eager launches, synchronises.
int pplhip_debug_run_dump(pplhip_ctx* c, int rank, float* hidden_dump_host) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !hidden_dump_host) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    if (!R.kv_cache) return fail(c, rank, PPLHIP_INVALID_VALUE, "kv slab not allocated");
    HIPCK(c, rank, hipSetDevice(R.device));
    if (R.B == 0) return 0;
    const int L = c->d.num_layers;
    const size_t n = (size_t)R.T * c->d.hidden_dim;
    HIPCK(c, rank, hipMalloc((void**)&R.dump_dev, (size_t)(L + 1) * 2 * n * 2));
    int rc = run_launches(c, rank);
    std::vector<uint16_t> host((size_t)(L + 1) * 2 * n);
    hipError_t e = hipStreamSynchronize(R.stream);
    if (e == hipSuccess) e = hipMemcpy(host.data(), R.dump_dev, host.size() * 2, hipMemcpyDeviceToHost);
    hipFree(R.dump_dev);
    R.dump_dev = nullptr;
    if (rc) return rc;
    HIPCK(c, rank, e);
    for (size_t i = 0; i < n; ++i) hidden_dump_host[i] = h2f_host(host[i]);
    for (int l = 1; l <= L; ++l) {
        const uint16_t* a = host.data() + (size_t)l * 2 * n;
        const uint16_t* b = a + n;
        float* o = hidden_dump_host + (size_t)l * n;
        for (size_t i = 0; i < n; ++i) o[i] = h2f_host(f2h_host(h2f_host(a[i]) + h2f_host(b[i])));
    }
    return p2p_check(c, rank);
}

int pplhip_logits(pplhip_ctx* c, int rank, float** logits_device, int64_t* stride) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    if (logits_device) *logits_device = c->ranks[rank].logits;
    if (stride) *stride = c->d.vocab_size;
    return 0;
}

int pplhip_sync(pplhip_ctx* c, int rank) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    HIPCK(c, rank, hipSetDevice(c->ranks[rank].device));
    HIPCK(c, rank, hipStreamSynchronize(c->ranks[rank].stream));
    return p2p_check(c, rank);
}

int pplhip_copy_logits(pplhip_ctx* c, int rank, float* dst, int64_t batch) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size() || !dst) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    HIPCK(c, rank, hipSetDevice(R.device));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    if (int rc = p2p_check(c, rank)) return rc;
    HIPCK(c, rank, hipMemcpy(dst, R.logits, (size_t)batch * c->d.vocab_size * 4, hipMemcpyDeviceToHost));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ sampler */

int pplhip_sample(pplhip_ctx* c, const float* logits_device, const pplhip_sample_args* a, int32_t* output_host,
                  float* logprob_host) {
    if (!c || !a || !logits_device || !output_host || !logprob_host) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[0];
    const int B = a->batch;
    if (B < 0 || B > R.cap_B) return fail(c, 0, PPLHIP_INVALID_VALUE, "batch exceeds max_running_batch");
    HIPCK(c, 0, hipSetDevice(R.device));
    hipStream_t s = R.stream;
    // src/backends/cuda/post_processor.cc:126,154-177: temperatures are skipped when the penalty kernel already
    // applied them; device copies are refreshed only when the batch changed and (quirk Q3 of SURVEY.md, kept)
    // passed to the kernel only on those steps.
    const float* temps_host = a->enable_penalty ? nullptr : a->temperatures;
    const float* temp_opt = nullptr;
    const float* topp_opt = nullptr;
    if (a->req_list_changed) {
        if (temps_host) { HIPCK(c, 0, hipMemcpyAsync(R.d_temp, temps_host, B * 4, hipMemcpyHostToDevice, s)); temp_opt = R.d_temp; }
        if (a->top_p) { HIPCK(c, 0, hipMemcpyAsync(R.d_topp, a->top_p, B * 4, hipMemcpyHostToDevice, s)); topp_opt = R.d_topp; }
    }
    static const bool trace = getenv("PPLHIP_SAMPLE_TRACE") != nullptr;  // diagnosis: what every sampling call was handed
    if (trace) {
        fprintf(stderr, "[sample] B %d top_k0 %d top_p0 %g changed %d penalty %d temps", B, a->default_top_k, a->default_top_p, a->req_list_changed, a->enable_penalty);
        for (int i = 0; i < B && a->temperatures; ++i) fprintf(stderr, " %g", a->temperatures[i]);
        fprintf(stderr, " top_p");
        for (int i = 0; i < B && a->top_p; ++i) fprintf(stderr, " %g", a->top_p[i]);
        fprintf(stderr, " top_k");
        for (int i = 0; i < B && a->top_k; ++i) fprintf(stderr, " %d", a->top_k[i]);
        fprintf(stderr, "\n");
    }
    // post_processor.cc:179-183: unseeded rand() sequence (one default value, then one per row)
    const float default_rand = (float)rand() / (float)RAND_MAX;
    (void)default_rand;
    for (int i = 0; i < B; ++i) R.h_rand[i] = (float)rand() / (float)RAND_MAX;
    if (a->default_top_k == 1) {
        HIPCK(c, 0, launch_sample_greedy(s, logits_device, temp_opt, B, a->vocab_size, a->batch_stride, R.d_tokout, R.d_lp));
    } else {
        // top_k <= 0: top-p over the whole vocabulary; top_k > 1024 is clamped (k_sample.hip).  default_top_k is the FIRST
        // row's client-supplied value: it must never turn into a batch-wide Execute failure.
        HIPCK(c, 0, hipMemcpyAsync(R.d_rand, R.h_rand, B * 4, hipMemcpyHostToDevice, s));
        HIPCK(c, 0, launch_sample_topk_topp(s, logits_device, temp_opt, topp_opt, R.d_rand, B, a->vocab_size, a->batch_stride,
                                            a->default_top_k, a->default_top_p, nullptr, R.d_tokout, R.d_lp));
    }
    HIPCK(c, 0, hipMemcpyAsync(output_host, R.d_tokout, B * 4, hipMemcpyDeviceToHost, s));
    HIPCK(c, 0, hipMemcpyAsync(logprob_host, R.d_lp, B * 4, hipMemcpyDeviceToHost, s));
    HIPCK(c, 0, hipStreamSynchronize(s));  // the step's only host<->device synchronisation (post_processor.cc:212)
    if (trace) {
        std::vector<float> row(a->vocab_size);
        for (int i = 0; i < B; ++i) {
            (void)hipMemcpy(row.data(), logits_device + (size_t)i * a->batch_stride, (size_t)a->vocab_size * 4, hipMemcpyDeviceToHost);
            double sum = 0;
            int am = 0;
            for (int v = 0; v < a->vocab_size; ++v) { sum += row[v]; if (row[v] > row[am]) am = v; }
            fprintf(stderr, "[sample]   row %d -> token %d logprob %.6f | logits sum %.6f argmax %d (%.6f) rand %.8f\n", i, output_host[i], logprob_host[i], sum, am,
                    row[am], R.h_rand[i]);
        }
    }
    return p2p_check(c, 0);
}

int pplhip_penalty(pplhip_ctx* c, float* logits_device, const pplhip_penalty_args* a) {
    if (!c || !a || !logits_device) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[0];
    if (!R.count_map) return fail(c, 0, PPLHIP_INVALID_VALUE, "context was created without enable_penalty");
    const int B = a->batch;
    if (B < 0 || B > R.cap_B || B != R.B) return fail(c, 0, PPLHIP_INVALID_VALUE, "penalty batch does not match the step");
    HIPCK(c, 0, hipSetDevice(R.device));
    hipStream_t s = R.stream;
    if (a->req_list_changed) {  // post_processor.cc:231-263
        HIPCK(c, 0, hipMemcpyAsync(R.d_ptemp, a->temperatures, B * 4, hipMemcpyHostToDevice, s));
        HIPCK(c, 0, hipMemcpyAsync(R.d_slots, a->batch_slots, B * 8, hipMemcpyHostToDevice, s));
        HIPCK(c, 0, hipMemcpyAsync(R.d_rep, a->repetition_penalties, B * 4, hipMemcpyHostToDevice, s));
        if (a->presence_penalties) HIPCK(c, 0, hipMemcpyAsync(R.d_pres, a->presence_penalties, B * 4, hipMemcpyHostToDevice, s));
        if (a->frequency_penalties) HIPCK(c, 0, hipMemcpyAsync(R.d_freq, a->frequency_penalties, B * 4, hipMemcpyHostToDevice, s));
    }
    HIPCK(c, 0, launch_penalty(s, logits_device, R.d_ptemp, R.d_rep, a->presence_penalties ? R.d_pres : nullptr,
                               a->frequency_penalties ? R.d_freq : nullptr, R.d_slots, R.d_tok, R.d_seq, R.d_sp, B,
                               a->vocab_size, c->d.vocab_size, (int)R.decoding_batches, R.count_map));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ measurement */

int pplhip_profile_reset(pplhip_ctx* c, int rank) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    HIPCK(c, rank, hipSetDevice(R.device));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    for (auto& e : R.prof) R.prof_free.push_back({e.a, e.b});
    R.prof.clear();
    return 0;
}

int pplhip_profile_get(pplhip_ctx* c, int rank, int cls, int64_t* launches, double* total_ms) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    Rank& R = c->ranks[rank];
    HIPCK(c, rank, hipSetDevice(R.device));
    HIPCK(c, rank, hipStreamSynchronize(R.stream));
    int64_t n = 0;
    double ms = 0;
    for (auto& e : R.prof)
        if (e.cls == cls) {
            float t = 0;
            if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) { ms += t; ++n; }
        }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 0;
}

int pplhip_profile_mode(pplhip_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return PPLHIP_INVALID_VALUE;
    c->o.enable_profiling = mode;
    return 0;
}

int pplhip_mem_info(pplhip_ctx* c, int rank, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!c || rank < 0 || rank >= (int)c->ranks.size()) return PPLHIP_INVALID_VALUE;
    HIPCK(c, rank, hipSetDevice(c->ranks[rank].device));
    size_t f = 0, t = 0;
    HIPCK(c, rank, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

/* ------------------------------------------------------------------------------------------------ operators */

static int op_rc(hipError_t e) { return e == hipSuccess ? 0 : (e == hipErrorInvalidValue ? PPLHIP_INVALID_VALUE : PPLHIP_DEVICE_RUNTIME_ERROR); }

int pplhip_op_embedding(void* stream, const int64_t* token_ids, const void* table, int64_t T, int32_t hidden, void* out) {
    return op_rc(launch_embedding((hipStream_t)stream, token_ids, (const uint16_t*)table, T, hidden, (uint16_t*)out));
}

int pplhip_op_rmsnorm(void* stream, const void* x, const void* skip, const void* w, float eps, int64_t T, int32_t hidden,
                      void* out, void* residual_out) {
    return op_rc(launch_rmsnorm((hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)skip, (const uint16_t*)w, eps, T,
                                hidden, nullptr, (uint16_t*)out, (uint16_t*)residual_out));
}

// split-K scratch of the stand-alone operators (the runtime owns its own per rank); one per device, never freed
static float* op_linear_ws(size_t* bytes) {
    static float* ws[16] = {nullptr};
    static const size_t ws_bytes = (size_t)64 << 20;
    int dev = 0;
    *bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!ws[dev] && hipMalloc((void**)&ws[dev], ws_bytes) != hipSuccess) ws[dev] = nullptr;
    if (ws[dev]) *bytes = ws_bytes;
    return ws[dev];
}

int pplhip_op_linear(void* stream, const void* x, const void* w, const void* scale, int32_t wq_bit, int32_t group, int64_t M,
                     int32_t N, int32_t K, void* y, int32_t out_fp32) {
    size_t ws_bytes = 0;
    float* ws = op_linear_ws(&ws_bytes);
    return op_rc(launch_linear((hipStream_t)stream, (const uint16_t*)x, w, (const uint16_t*)scale, wq_bit, group, M, N, K, y, N,
                               out_fp32 != 0, ws, ws_bytes));
}

// (the same scratch: without it 4 <= M <= 256 ran unsplit -- N / 128 blocks instead of a full chip)
int pplhip_op_linear_swiglu(void* stream, const void* x, const void* w, const void* scale, int32_t wq_bit, int32_t group, int64_t M,
                            int32_t N, int32_t K, void* y) {
    size_t ws_bytes = 0;
    float* ws = op_linear_ws(&ws_bytes);
    return op_rc(launch_linear((hipStream_t)stream, (const uint16_t*)x, w, (const uint16_t*)scale, wq_bit, group, M, N, K, y, N / 2,
                               false, ws, ws_bytes, true));
}

int pplhip_op_rmsnorm_quant(void* stream, const void* x, const void* skip, const void* w, float eps, int64_t T, int32_t hidden,
                            void* residual_out, void* q, float* sx) {
    return op_rc(launch_rmsnorm((hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)skip, (const uint16_t*)w, eps, T, hidden, nullptr,
                                nullptr, (uint16_t*)residual_out, (int8_t*)q, sx));
}

int pplhip_op_quant_act(void* stream, const void* x, int64_t M, int32_t K, void* q, float* sx) {
    return op_rc(launch_quant_act((hipStream_t)stream, (const uint16_t*)x, M, K, K, (int8_t*)q, K, sx));
}

int pplhip_op_quant_weight(void* stream, const void* w, int32_t N, int32_t K, void* q, void* scale) {
    return op_rc(launch_quant_weight((hipStream_t)stream, (const uint16_t*)w, N, K, (int8_t*)q, K, (uint16_t*)scale));
}

int pplhip_op_linear_i8(void* stream, const void* xq, const float* sx, const void* w, const void* scale, int64_t M, int32_t N, int32_t K,
                        void* y, int32_t out_fp32, int32_t swiglu) {
    return op_rc(launch_linear_i8((hipStream_t)stream, (const int8_t*)xq, sx, (const int8_t*)w, (const uint16_t*)scale, M, N, K, y,
                                  swiglu ? N / 2 : N, out_fp32 != 0, swiglu != 0));
}

int pplhip_op_silu_mul(void* stream, const void* gate_up, int64_t T, int32_t inter, void* out) {
    return op_rc(launch_silu_mul((hipStream_t)stream, (const uint16_t*)gate_up, T, inter, (uint16_t*)out));
}

static KvAddr view_addr(const pplhip_kv_view* v) {
    pplhip_model_desc d;
    memset(&d, 0, sizeof(d));
    d.num_layers = v->num_layers;
    d.cache_quant_bit = v->quant_bit; d.cache_quant_group = v->quant_group;
    d.cache_layout = v->layout; d.cache_mode = v->mode; d.page_size = v->page_size;
    return make_kv_addr(d, v->kv_heads, v->head_dim, (uint64_t)v->max_tokens, v->cache, (uint16_t*)v->scale, v->layer);
}

int pplhip_op_rope_kv_write(void* stream, void* qkv, const float* cos_sin, const pplhip_kv_view* kv, const int64_t* seq_starts,
                            const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t T,
                            int32_t num_heads) {
    if (!kv) return PPLHIP_INVALID_VALUE;
    return op_rc(launch_rope_kv_write((hipStream_t)stream, (uint16_t*)qkv, cos_sin, view_addr(kv), kv->quant_bit, kv->quant_group,
                                      seq_starts, start_pos, cache_indices, max_pages, B, 0, T, num_heads, kv->kv_heads, kv->head_dim));
}

int pplhip_op_attention(void* stream, const void* qkv, const pplhip_kv_view* kv, const int64_t* seq_starts,
                        const int64_t* start_pos, const int64_t* cache_indices, int64_t max_pages, int64_t B, int64_t T,
                        int64_t decoding_batches, int64_t max_seq_len, int64_t max_kv_len, int32_t num_heads, int32_t split_k,
                        void* workspace, uint64_t workspace_bytes, void* out) {
    (void)T;
    if (!kv) return PPLHIP_INVALID_VALUE;
    const int64_t nb = std::min<int64_t>(std::max<int64_t>(decoding_batches, 0), B);
    int split = split_k < 1 ? 1 : split_k;
    if (split > 1 && attn_decode_workspace_bytes(nb, num_heads, kv->head_dim, split) > workspace_bytes) return PPLHIP_INVALID_VALUE;
    hipStream_t s = (hipStream_t)stream;
    const KvAddr a = view_addr(kv);
    hipError_t e = hipSuccess;
    if (nb > 0)
        e = launch_attn_decode(s, (const uint16_t*)qkv, a, kv->quant_bit, seq_starts, start_pos, cache_indices, max_pages, nb,
                               num_heads, kv->kv_heads, kv->head_dim, max_kv_len, split, 256, (float*)workspace, (uint16_t*)out);
    if (e == hipSuccess && B > nb)
        e = launch_attn_prefill(s, (const uint16_t*)qkv, a, kv->quant_bit, seq_starts, start_pos, cache_indices, max_pages, nb, B,
                                num_heads, kv->kv_heads, kv->head_dim, max_seq_len, (uint16_t*)out, max_kv_len, (float*)workspace,
                                (size_t)workspace_bytes, nb, T - nb);
    return op_rc(e);
}

}  // extern "C"
