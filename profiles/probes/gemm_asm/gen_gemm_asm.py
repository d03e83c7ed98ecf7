#!/usr/bin/env python3
"""Generates k_gemm_asm_body.inc: the hand-allocated, hand-scheduled K loop of the W8A16 consumer waves of k_gemm_asm.hip (round 4).

Why a generator: the loop is ONE inline-asm statement per kernel variant (hipcc neither schedules nor allocates anything inside it), every
register below is a physical register chosen here, and every s_waitcnt lgkmcnt(N) is COUNTED from the order in which the LDS reads are
issued -- the script keeps that queue and derives N, so that a change of the schedule cannot leave a stale count behind.

The wave (one consumer wave per SIMD, beside one producer wave that only refills the LDS-DMA ring):
  * owns 128 activation rows (m) x 32 NI weight rows (n) of the block tile: 4 x NI accumulators of v_mfma_f32_32x32x16_f16 (A = weights,
    B = activations: a lane then holds 4 consecutive output channels of one activation row per accumulator quad, like the 16x16 kernels);
  * K tile = 64 = 4 k-steps of 16.  k-step s: 4 NI MFMAs, ordered activation block j outer / weight block i inner, so that the slot of
    activation fragment j is free after 3j + NI MFMAs and is refilled with k-step s + 1's fragment at once (ONE register set of activation
    fragments + one spare slot), while the next k-step's int8 weight fragments are read (ds_read_b64), converted to fp16 in registers
    (v_xor / v_perm / v_pk_add: exact) and written to the OTHER of two converted-weight sets -- all of it in the issue slots between MFMAs;
  * meets the other waves ONCE per K tile (s_barrier after the first MFMA of the tile's last k-step, when every LDS read of the tile has
    landed) and reads the next tile's first fragments behind that barrier, under the remaining MFMAs of the tile.

Register file of a consumer wave (2 waves per SIMD -> 256 registers): accumulators = asm output operands (compiler-allocated, 64 NI
registers), everything else physical (V_* below).  Hazards handled here because hipcc pads nothing inside an asm statement (measured on
hipcc's own code, round 4): VALU write -> MFMA A/B read needs 2 wait states (guaranteed by distance: a fragment is converted a whole k-step
before it is used); MFMA A/B read -> overwrite: >= 1 instruction (LDS returns arrive > 60 cycles later); MFMA result -> compiler code
after the statement: the statement ends with 24 wait states.
"""
import sys

NI = int(sys.argv[1]) if len(sys.argv) > 1 else 3           # weight row blocks of 32 per wave
ABL = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # diagnosis builds (WRONG results): 1 no fragment reads in the loop, 2 no conversion, 4 no barrier
ST = int(sys.argv[3]) if len(sys.argv) > 3 else 3            # ring stages
PRIO = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # s_setprio of the consumer waves
XB = 128 * 64 * 2                                           # activation bytes per ring stage
WB = 4 * 32 * NI * 64                                       # weight bytes per ring stage (4 consumer waves)
NJ = 4

# ---- physical registers -----------------------------------------------------------------------------------------------------------
TOP = 255
def alloc(n, align=1):
    global TOP
    TOP -= n
    while (TOP + 1) % align: TOP -= 1
    return TOP + 1
V_X = [alloc(4, 2) for _ in range(5)]                       # activation fragment slots X0..X3 + the spare X4
V_WC = [[alloc(4, 2) for _ in range(NI)] for _ in range(2)]  # converted weight fragments, sets [even k-step][odd k-step]
V_RAW = [alloc(2, 2) for _ in range(NI)]                    # int8 weight fragments as read (8 bytes per lane)
V_C64 = alloc(1)
V_XA = [alloc(1) for _ in range(4)]                         # LDS byte address of this lane's activation chunk, per k-step
V_WA = [alloc(1) for _ in range(4)]                         # ... weight chunk
LOW = TOP + 1                                               # lowest physical register used here
S_CNT, S_STAGE, S_DX, S_DW, S_SEL0, S_SEL1, S_BIAS, S_XBP, S_XBN, S_WBP, S_WBN = range(84, 95)

def vr(base, n): return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"
def acc(i, j): return f"%{j * NI + i}"                      # operand numbers 0 .. 4 NI - 1: accumulators, j-major

class Gen:
    def __init__(self):
        self.lines = []
        self.q = []          # outstanding LDS reads, issue order
        self.in_loop = False

    def emit(self, s): self.lines.append(s)

    def ds_x(self, slot, s, j, tag):
        if (ABL & 1) and self.in_loop: return
        self.emit(f"ds_read_b128 {vr(V_X[slot], 4)}, v{V_XA[s]} offset:{j * 4096}")
        self.q.append(tag)

    def ds_w(self, s, i, tag):
        if (ABL & 1) and self.in_loop: return
        self.emit(f"ds_read_b64 {vr(V_RAW[i], 2)}, v{V_WA[s]} offset:{i * 2048}")
        self.q.append(tag)

    def wait(self, tag):
        """block until the read `tag` (and, in-order return, everything issued before it) has landed"""
        if tag not in self.q: return
        k = self.q.index(tag)
        n = len(self.q) - 1 - k
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.q = self.q[k + 1:]

    def wait_all(self):
        if self.q:
            self.emit("s_waitcnt lgkmcnt(0)")
            self.q = []

    def cvt_ops(self, wset, i):
        """the 10 VALU ops that turn raw fragment i (8 int8 per lane) into 8 fp16 in converted set `wset`"""
        r, d = V_RAW[i], V_WC[wset][i]
        if (ABL & 2) and self.in_loop: return []
        ops = [f"v_xor_b32_e32 v{r}, 0x80808080, v{r}", f"v_xor_b32_e32 v{r + 1}, 0x80808080, v{r + 1}"]
        ops += [f"v_perm_b32 v{d}, v{V_C64}, v{r}, s{S_SEL0}", f"v_perm_b32 v{d + 1}, v{V_C64}, v{r}, s{S_SEL1}",
                f"v_perm_b32 v{d + 2}, v{V_C64}, v{r + 1}, s{S_SEL0}", f"v_perm_b32 v{d + 3}, v{V_C64}, v{r + 1}, s{S_SEL1}"]
        ops += [f"v_pk_add_f16 v{d + k}, v{d + k}, s{S_BIAS}" for k in range(4)]
        return ops

    def mfma(self, i, j, wset, slot):
        self.emit(f"v_mfma_f32_32x32x16_f16 {acc(i, j)}, {vr(V_WC[wset][i], 4)}, {vr(V_X[slot], 4)}, {acc(i, j)}")

    # one k-step: 4 NI MFMAs with the fillers of `plan` hung behind them.  plan: dict m (1-based MFMA index) -> list of callables
    def kstep(self, s, plan, xslots):
        wset = s & 1
        m = 0
        for j in range(NJ):
            self.wait(("X", xslots[j]))          # activation fragment of this block (no-op when already retired)
            for i in range(NI):
                m += 1
                self.mfma(i, j, wset, xslots[j])
                for f in plan.get(m, []): f()


def body(g, last_tile):
    """k-steps 0..3 of one K tile.  State on entry: Wc[0] holds k-step 0's weights; X0..X3 hold / are receiving k-step 0's fragments."""
    nm = NJ * NI
    for s in range(4):
        plan = {}
        def add(m, f): plan.setdefault(m, []).append(f)
        nxt = s + 1
        xs = [0, 1, 2, 3] if s < 3 else [0, 1, 2, 4]
        cvt_from = 1 + NI                               # first MFMA behind which conversion ops are hung
        if s < 3:
            # next k-step of the same tile
            for i in range(NI): add(1, (lambda i=i: g.ds_w(nxt, i, ("W", i))))
            if s == 2: add(1, lambda: g.ds_x(4, 3, 3, ("X", 4)))                    # the spare slot takes k-step 3's last block early
            for j in range(NJ):
                if s == 2 and j == 3: continue
                add(NI * (j + 1), (lambda j=j: g.ds_x(j, nxt, j, ("X", j))))         # slot j is free behind its last MFMA
        elif not last_tile:
            # behind the first MFMA: every read of this tile has landed -> barrier -> the ring moves on -> first reads of the next tile
            def sync():
                g.wait_all()
                if not (ABL & 4): g.emit("s_barrier")
                g.emit(f"s_add_u32 s{S_STAGE}, s{S_STAGE}, 1")
                g.emit(f"s_cmp_eq_u32 s{S_STAGE}, {ST}")
                g.emit(f"s_cselect_b32 s{S_DX}, s{S_XBN}, s{S_XBP}")
                g.emit(f"s_cselect_b32 s{S_DW}, s{S_WBN}, s{S_WBP}")
                g.emit(f"s_cselect_b32 s{S_STAGE}, 0, s{S_STAGE}")
                for k in range(4): g.emit(f"v_add_u32_e32 v{V_WA[k]}, s{S_DW}, v{V_WA[k]}")
                for k in range(4): g.emit(f"v_add_u32_e32 v{V_XA[k]}, s{S_DX}, v{V_XA[k]}")
            add(1, sync)
            for i in range(NI): add(2, (lambda i=i: g.ds_w(0, i, ("W", i))))
            add(2, lambda: g.ds_x(3, 0, 3, ("X", 3)))                                # X3 is idle during this k-step (block 3 sits in X4)
            for j in range(3): add(NI * (j + 1) + (1 if j == 0 else 0), (lambda j=j: g.ds_x(j, 0, j, ("X", j))))
            cvt_from = 2 + NI
        if s < 3 or not last_tile:
            # conversion of the fragments just requested: NI x 10 VALU ops spread over the MFMAs cvt_from .. nm - 1
            ops = []
            for i in range(NI): ops += g.cvt_ops(nxt & 1, i)
            slots = list(range(cvt_from, nm))
            per = (len(ops) + len(slots) - 1) // len(slots)
            first = [True]
            def chunk(lo, hi):
                def f():
                    if first[0]:
                        g.wait(("W", NI - 1))
                        first[0] = False
                    for o in ops[lo:hi]: g.emit(o)
                return f
            k = 0
            if not ops:
                add(cvt_from, lambda: g.wait(("W", NI - 1)))
                slots = []
            for m in slots:
                if k >= len(ops): break
                add(m, chunk(k, min(k + per, len(ops))))
                k += per
            assert k >= len(ops)
        g.kstep(s, plan, xs)


def generate():
    g = Gen()
    e = g.emit
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if PRIO: e(f"s_setprio {PRIO}")
    # constants, addresses
    e(f"s_mov_b32 s{S_SEL0}, 0x04010400"); e(f"s_mov_b32 s{S_SEL1}, 0x04030402"); e(f"s_mov_b32 s{S_BIAS}, 0xe480e480")
    e(f"s_mov_b32 s{S_XBP}, {XB}"); e(f"s_mov_b32 s{S_XBN}, {-(ST - 1) * XB & 0xffffffff:#x}")
    e(f"s_mov_b32 s{S_WBP}, {WB}"); e(f"s_mov_b32 s{S_WBN}, {-(ST - 1) * WB & 0xffffffff:#x}")
    e(f"s_mov_b32 s{S_STAGE}, 0")
    e(f"s_sub_u32 s{S_CNT}, %{4 * NI + 2}, 1")
    e(f"v_mov_b32_e32 v{V_C64}, 0x64646464")
    for k in range(4):
        if k == 0:
            e(f"v_mov_b32_e32 v{V_XA[0]}, %{4 * NI}"); e(f"v_mov_b32_e32 v{V_WA[0]}, %{4 * NI + 1}")
        else:
            e(f"v_xor_b32_e32 v{V_XA[k]}, {32 * k}, v{V_XA[0]}"); e(f"v_xor_b32_e32 v{V_WA[k]}, {16 * k}, v{V_WA[0]}")
    for j in range(NJ):
        for i in range(NI):
            for k in range(16): e(f"v_mov_b32_e32 {acc(i, j)}[{k}], 0") if False else None
    # (accumulators are zeroed by the caller: "+v" operands)
    e("s_barrier")                                   # tile 0 is published
    # prologue = the fillers of a tile's last k-step without its MFMAs: same issue order, same queue state on loop entry
    for i in range(NI): g.ds_w(0, i, ("W", i))
    g.ds_x(3, 0, 3, ("X", 3))
    g.ds_x(0, 0, 0, ("X", 0))
    g.wait(("W", NI - 1))
    for i in range(NI):
        for o in g.cvt_ops(0, i): e(o)
    g.ds_x(1, 0, 1, ("X", 1))
    g.ds_x(2, 0, 2, ("X", 2))
    entry_q = list(g.q)
    e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 2f")
    e(".p2align 6")
    e("1:")
    g.in_loop = True
    body(g, last_tile=False)
    g.in_loop = False
    if ABL & 1: g.q = list(entry_q)
    assert g.q == entry_q, (g.q, entry_q)            # the loop's back edge sees the queue it was entered with
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_lg_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 1b")
    e("2:")
    g.q = list(entry_q)
    body(g, last_tile=True)
    e("s_nop 15"); e("s_nop 7")
    return g.lines


def main():
    lines = generate()
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    clob = [f"v{r}" for r in range(LOW, 256)] + [f"s{r}" for r in range(S_CNT, S_WBN + 1)] + ["scc", "memory"]
    out = [f"// GENERATED by gen_gemm_asm.py {NI} {ABL} {ST} {PRIO} -- do not edit.  {len(lines)} lines, {n_mfma} MFMAs; physical registers v{LOW}..v255.",
           f"// operands: %0..%{4 * NI - 1} accumulators (j-major), %{4 * NI} xa0, %{4 * NI + 1} wa0 (VGPR), %{4 * NI + 2} ktiles (SGPR)",
           f"#define GEMM_ASM_NI{NI}_LOW_VGPR {LOW}",
           f"#define GEMM_ASM_NI{NI}_TEXT \\"]
    for l in lines: out.append(f'    "{l}\\n\\t" \\')
    out.append('    ""')
    out.append(f"#define GEMM_ASM_NI{NI}_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
