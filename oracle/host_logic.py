"""Oracle for the HOST logic of the hot path (TEST INFRASTRUCTURE ONLY): a plain-Python restatement of the reference's
scheduler pieces, each function citing the reference file:line it follows (paths relative to /root/reference).

Pinned: `HashCombine` and `PrefixCacheManager` against values captured from the reference's own code
(tests/golden/host_logic.json).  The external allocators (ppl.common PageManager /
CompactAddrManager) are not in the reference tree: their contracts are the ones its call sites rely on
(SURVEY.md section 10); allocation ORDER (lowest page id / first fit) is this build's choice and is restated here.
"""
import math

INT64_MAX = 2**63 - 1
M64 = 2**64 - 1


def hash_combine(prev, vec):
    """src/utils/utils.cc:87-94.  `vec[i] + 0x9e3779b9` is 32-bit unsigned arithmetic (int + unsigned literal),
    zero-extended; `prev + 0x9e3779b9` is 64-bit."""
    seed = len(vec) & M64
    seed ^= (prev + 0x9e3779b9 + ((seed << 6) & M64) + (seed >> 2)) & M64
    for v in vec:
        mixed = (v + 0x9e3779b9) & 0xFFFFFFFF
        seed ^= (mixed + ((seed << 6) & M64) + (seed >> 2)) & M64
    return seed


class PrefixCacheManager:
    """src/utils/prefix_cache_manager.h:118-186 (+ LRUCache :14-116: AddToHead on insert, eviction from the tail)."""

    def __init__(self):
        self.map = {}      # hash -> [page_id, ref_count]
        self.lru = []      # index 0 = head (most recently released)

    def find(self, h):
        return self.map[h][0] if h in self.map else -1

    def insert(self, h, page_id):
        if h not in self.map:          # unordered_map::insert does not overwrite
            self.map[h] = [page_id, 1]

    def inc_ref(self, hashes):
        for h in hashes:
            if h not in self.map:
                break
            self.map[h][1] += 1
            if h in self.lru:
                self.lru.remove(h)

    def dec_ref(self, hashes):
        for h in hashes:
            if h not in self.map:
                break
            self.map[h][1] -= 1
            if self.map[h][1] == 0 and h not in self.lru:
                self.lru.insert(0, h)

    def evict(self, n):
        pages = []
        for _ in range(min(n, len(self.lru))):
            h = self.lru.pop()          # tail = least recently released
            pages.append(self.map.pop(h)[0])
        return pages

    def size(self):
        return len(self.map)

    def reset(self):
        self.map.clear()
        self.lru.clear()


class IndexManager:
    """src/utils/index_manager.h:25-78 over a first-fit, lowest-start, coalescing range allocator."""

    def __init__(self, n):
        self.free = [(0, n)] if n else []
        self.avail = n

    def alloc(self, nr):
        if nr == 0:
            return INT64_MAX
        for i, (s, l) in enumerate(self.free):
            if l >= nr:
                if l == nr:
                    self.free.pop(i)
                else:
                    self.free[i] = (s + nr, l - nr)
                self.avail -= nr
                return s
        return INT64_MAX

    def release(self, start, nr):
        self.free.append((start, nr))
        self.free.sort()
        merged = []
        for s, l in self.free:
            if merged and merged[-1][0] + merged[-1][1] == s:
                merged[-1] = (merged[-1][0], merged[-1][1] + l)
            else:
                merged.append((s, l))
        self.free = merged
        self.avail += nr


class PageManager:
    """ppl.common PageManager contract: Alloc appends n ids or fails whole; LIFO free list
    seeded so that a fresh manager yields 0, 1, 2, ..."""

    def __init__(self, max_tokens, page_size):
        n = max_tokens // page_size if page_size > 0 else 0
        self.free = list(range(n - 1, -1, -1))

    def alloc(self, n, out):
        if n < 0 or len(self.free) < n:
            return False
        for _ in range(n):
            out.append(self.free.pop())
        return True

    def release(self, ids):
        self.free.extend(ids)

    def avail(self):
        return len(self.free)


def fake_next_token(last_token, kv_len, vocab):
    """the scripted model of tests/host/sched_trace.cc"""
    return (31 * last_token + 7 * kv_len + 3) % vocab


def simulate(sc):
    """LLMGenerator::Generate (src/generator/llm_generator.cc:574-786) for a scenario whose requests are all queued
    before the generator thread starts.  Returns (steps, responses, failed)."""
    m, g = sc["model"], sc["generator"]
    mode, P, vocab = m.get("cache_mode", 0), m.get("page_size", 0), m.get("vocab_size", 1000)
    max_batch = g.get("max_running_batch", 1024)
    max_in = g.get("max_input_tokens_per_request", 4096)
    max_out = g.get("max_output_tokens_per_request", 4096)
    max_total = g.get("max_total_tokens_per_request", 8192)
    max_step = g.get("max_tokens_per_step", 8192)
    max_cool = g.get("max_cooldown_request", 2)
    prefix = g.get("enable_prefix_cache", False)
    max_prefill = 1 if prefix else g.get("max_prefill_batch", 64)   # tools/offline_inference.cc:97-99
    penalty = g.get("enable_penalty", False)
    gstop = set(g.get("stop_tokens", []))
    N = sc.get("kv_cache_max_tokens", 4096)
    fail_at_run = sc.get("fail_at_run", -1)
    cancel = {}
    for c in sc.get("cancel", []):
        cancel.setdefault(c["at_step"], []).append(c["id"])

    idx, slots, pages, pc = IndexManager(N), IndexManager(max_batch), PageManager(N, P), PrefixCacheManager()
    queue = [dict(r) for r in sc["requests"]]
    stash = None
    responses, failed, steps = {}, {}, []
    runs = 0
    # GeneratorThreadFunc (:342-366): Generate() returns when nothing is running (:657-660) and is entered again while the
    # scheduler still holds requests -- with fresh locals (running_batch, cool-down count, loop_step, ModelInput, :574-590).
    # NB the stale running_batch of :628 (the previous step's batch, finished rows included) makes a request that fits an idle
    # generator wait for exactly that re-entry.
    while queue or stash is not None:
        rows = []            # running requests (dicts), row order
        mi = dict(decoding_batches=0, max_pages=0, start_pos=[], cache_indices=[], page_list=[], batch_slots=[])
        finished = []        # finished_tasks queue (ids)
        running_batch, cool, changed, loop_step = 0, 0, True, 0
        stash, runs = _generate(locals())
    return steps, responses, failed


def _generate(E):
    """one call of LLMGenerator::Generate on the state `E` built by simulate(); returns (stash, runs)"""
    (mode, P, vocab, max_batch, max_in, max_out, max_total, max_step, max_cool, prefix, max_prefill, penalty, gstop, fail_at_run,
     cancel, idx, slots, pages, pc, queue, stash, responses, failed, steps, runs, rows, mi, finished, running_batch, cool, changed,
     loop_step) = [E[k] for k in ("mode", "P", "vocab", "max_batch", "max_in", "max_out", "max_total", "max_step", "max_cool", "prefix",
                                  "max_prefill", "penalty", "gstop", "fail_at_run", "cancel", "idx", "slots", "pages", "pc", "queue",
                                  "stash", "responses", "failed", "steps", "runs", "rows", "mi", "finished", "running_batch", "cool",
                                  "changed", "loop_step")]
    while True:
        hit_flag = False
        tot = running_batch                                  # :628-631
        adm_running, adm_prefill = running_batch, 0
        while adm_running < max_batch and adm_prefill < max_prefill and cool <= 0:      # :634-637
            req = stash if stash is not None else (queue[0] if queue else None)
            if req is None:
                break
            # ---- check_func :590-617
            toks = req["tokens"]
            first, rest, err = len(toks), -1, False
            cache_index, plist, hlist, hit, slot = INT64_MAX, [], [], 0, INT64_MAX
            tot += first
            if tot > max_step:
                accept = False
            else:
                accept = True
                if first == 0:                                # deviation of this build: empty prompt rejected
                    first, err = -1, True
                else:
                    # CheckTotalLen :441-478
                    if first > max_in:
                        first, err = -1, True
                    else:
                        rest = req["generation_length"]
                        if req["generation_length"] > max_out:
                            rest = max_out
                            if rest <= 0:
                                err = True
                        if not err and first + req["generation_length"] > max_total:
                            rest = max_total - first
                            if rest <= 0:
                                err = True
                    if not err and rest <= 0:                 # deviation: rejected before any KV is reserved
                        err = True
                if not err:
                    # CheckAndAllocGPUMemory :480-572
                    total_len = first + rest - 1
                    ok = True
                    if mode == 0:
                        cache_index = idx.alloc(total_len)
                        if cache_index == INT64_MAX:
                            cool = min(max(1, int(math.floor(len(rows) * 0.1))), max_cool)
                            ok = False
                    elif prefix:
                        prev, start = 0, 0
                        while start + P <= len(toks):
                            h = hash_combine(prev, toks[start:start + P])
                            pid = pc.find(h)
                            if pid == -1:
                                break
                            prev = h
                            plist.append(pid)
                            hlist.append(h)
                            start += P
                        pc.inc_ref(hlist)
                        avail = pages.avail()
                        need = (total_len - start + P - 1) // P
                        if avail < need:
                            ev = pc.evict(need - avail)
                            pages.release(ev)
                            if len(ev) < need - avail:
                                pc.dec_ref(hlist)
                                ok = False
                        if ok:
                            hit = len(hlist) * P
                            if hit:
                                hit_flag = True
                            if not pages.alloc(need, plist):
                                pc.dec_ref(hlist)
                                ok = False
                        if ok:
                            pos = start
                            while pos + P <= len(toks):
                                h = hash_combine(prev, toks[pos:pos + P])
                                pc.insert(h, plist[pos // P])
                                prev = h
                                hlist.append(h)
                                pos += P
                    else:
                        if not pages.alloc((total_len + P - 1) // P, plist):
                            ok = False
                    if ok and penalty:
                        slot = slots.alloc(1)
                        if slot == INT64_MAX:
                            ok = False
                    if not ok:
                        accept = False
                    else:
                        adm_running += 1
                        adm_prefill += 1
            # ---- MPSCRequestScheduler::TryPopRequest :58-88
            if not accept:
                if stash is None:
                    stash = queue.pop(0)
                break
            if stash is not None:
                stash = None
            else:
                queue.pop(0)
            # ---- ParseRequest :193-261
            if rest <= 0 or first == -1:
                failed[req["id"]] = 2                          # RC_INVALID_VALUE
                changed = True
                continue
            t = dict(tid=req["id"], rest=rest, total_len=first + rest, early=req.get("early_stopping", True),
                     stop=set(req["stop_tokens"]) if "stop_tokens" in req else None, steps=0, slot=slot,
                     cache_index=cache_index, pages=plist, hashes=hlist)
            if hit == 0:
                t["next"], t["start_pos"] = list(toks), 0
            elif hit == len(toks):
                t["next"], t["start_pos"] = [toks[-1]], hit - 1
            else:
                t["next"], t["start_pos"] = list(toks[hit:]), hit
            rows.append(t)
            mi["start_pos"].append(t["start_pos"])
            mi["batch_slots"].append(slot)
            if mode == 0:
                mi["cache_indices"].append(cache_index)
            else:
                mi["max_pages"] = max(len(plist), mi["max_pages"])
            changed = True
        running_batch = len(rows)
        if running_batch == 0:
            break
        # ---- UpdateInput :263-298
        token_inputs, seq_starts, kv_starts, msl, mkl = [], [0], [0], 0, 0
        if changed and mode == 1:
            mi["page_list"] = [INT64_MAX] * (running_batch * mi["max_pages"])
        for i, t in enumerate(rows):
            sl = len(t["next"])
            token_inputs += t["next"]
            seq_starts.append(seq_starts[i] + sl)
            kv_starts.append(kv_starts[i] + t["start_pos"] + sl)
            msl, mkl = max(msl, sl), max(mkl, t["start_pos"] + sl)
            if changed and mode == 1:
                mi["page_list"][i * mi["max_pages"]: i * mi["max_pages"] + len(t["pages"])] = t["pages"]
        steps.append(dict(step=loop_step, decoding_batches=mi["decoding_batches"], max_seq_len=msl, max_kv_len=mkl,
                          max_pages=mi["max_pages"], req_list_changed=int(changed), prefix_hit=int(hit_flag),
                          token_inputs=token_inputs, seq_starts=seq_starts, kv_starts=kv_starts,
                          start_pos=list(mi["start_pos"]), cache_indices=list(mi["cache_indices"]),
                          page_list=list(mi["page_list"]), batch_slots=list(mi["batch_slots"])))
        for cid in cancel.get(loop_step, []):                  # Connection-side ClearTask (llm_generator.h:143-145)
            finished.append(cid)
        # ---- Execute; failure path :681-688
        if fail_at_run >= 0 and runs == fail_at_run:
            runs += 1
            for t in rows:
                failed[t["tid"]] = 1                           # RC_OTHER_ERROR from LLMEngine::Execute
            # ReleaseResource :368-385 -- every page of a running request goes back (cached prefix pages included) and the
            # prefix cache is dropped whole
            for i, t in enumerate(rows):
                if mode == 0:
                    idx.release(t["cache_index"], t["total_len"] - 1)
                else:
                    pages.release(t["pages"])
                if penalty:
                    slots.release(mi["batch_slots"][i], 1)
            pc.reset()
            break
        runs += 1
        changed = False
        # ---- post-processing :699-735
        for i, t in enumerate(rows):
            tok = fake_next_token(t["next"][-1], mi["start_pos"][i] + len(t["next"]), vocab)
            fed = len(t["next"])
            t["next"] = [tok]
            if t["steps"] == 0:
                mi["start_pos"][i] += fed
                mi["decoding_batches"] += 1
            else:
                mi["start_pos"][i] += 1
            t["start_pos"] += fed
            t["steps"] += 1
            t["rest"] -= 1
            flag = 0
            stop_hit = t["early"] and (tok in gstop or (t["stop"] is not None and tok in t["stop"]))
            if t["rest"] <= 0 or stop_hit:
                flag = 1 if t["rest"] <= 0 else 2              # LENGTH / EOS_TOKEN
                if cool > 0:
                    cool -= 1
                finished.append(t["tid"])
                changed = True
            r = responses.setdefault(t["tid"], dict(tokens=[], finish=0))
            r["tokens"].append(tok)
            if flag:
                r["finish"] = flag
        # ---- DeleteTasks :387-439 + RemoveFinishedTask :300-340
        if finished:
            live = {t["tid"]: t for t in rows if t is not None}
            erased = set()
            for fid in finished:
                if fid not in live or fid in erased:
                    continue
                mi["decoding_batches"] -= 1
                row = next(i for i, t in enumerate(rows) if t is not None and t["tid"] == fid)
                t = rows[row]
                rows[row] = None
                erased.add(fid)
                # deviation from the reference (it leaves the flag alone): rows shift, so the next Execute must re-upload
                # the page table / sampler and penalty rows even when the removal came from a cancel on a quiet step
                changed = True
                if mode == 0:
                    idx.release(t["cache_index"], t["total_len"] - 1)
                elif prefix:
                    nh = len(t["hashes"])
                    pc.dec_ref(t["hashes"])
                    pages.release(t["pages"][nh:])
                else:
                    pages.release(t["pages"])
                if penalty:
                    slots.release(mi["batch_slots"][row], 1)
            finished = []
            keep = [i for i, t in enumerate(rows) if t is not None]
            rows = [rows[i] for i in keep]
            mi["start_pos"] = [mi["start_pos"][i] for i in keep]
            mi["batch_slots"] = [mi["batch_slots"][i] for i in keep]
            if mode == 0:
                mi["cache_indices"] = [mi["cache_indices"][i] for i in keep]
            else:
                mi["max_pages"] = max([len(t["pages"]) for t in rows], default=0)
        loop_step += 1
    return stash, runs
