#!/usr/bin/env python3
"""Generator of the hand-scheduled tile loop of dino_amd/csrc/attention_za.hip (writes dino_amd/csrc/attention_za_gen.inc).

    python tools/gen_attn_asm.py            # rewrite the .inc
    python tools/gen_attn_asm.py --check    # exit 1 if the committed .inc is not what this script generates
    python tools/gen_attn_asm.py --dump [--mq 2]   # one body as plain text

What is generated: ONE inline-asm body per (queries per wave, operand format) -- the whole K/V tile loop of a wave of
attn_fwd_za_kernel (MQ blocks of 32 queries, head dimension 64, 64-key tiles in a four-slot LDS ring) as a software pipeline over
32-key blocks.  Stage b of a wave runs, interleaved instruction by instruction,

    matrix pipe :  S(b+1) = K(b+1) . Q^T   (4 MFMAs per query block, fresh accumulator)   and   O^T += V(b-1)^T . P(b-1)^T   (4 per block)
    vector port :  P(b) = 2^S(b) in place (16 v_exp_f32 per query block), the row sum (16 v_add_f32), the pack to bf16 (8 v_cvt_pk_bf16_f32)
    LDS         :  the K fragments of block b+1 / b+2 and the V^T fragments of block b-1 / b, each four fragment uses ahead

so every MFMA is followed by two exponentials, two adds, one pack and at most two fragment reads ("gap"), and no instruction of a gap
depends on the MFMA in front of it.  Arithmetic, operand orientation, LDS image and summation order are those of attention_z.hip's
attn_fwd_z_kernel<1, 4, 8>: the outputs are bit-identical (tests/test_ops_gpu.py).  With MQ = 2 every K / V^T fragment read from LDS
feeds TWO MFMAs (one per query block): half the LDS read bytes per score at two waves per SIMD instead of four.

Why a generator: the waits.  LDS reads return in order, so the wait in front of an MFMA is `s_waitcnt lgkmcnt(N)` with N = the number of
reads issued after the one the MFMA needs; the emitter keeps the queue of outstanding reads and computes every N, and checks at every
label that all paths arrive with the same queue.  Registers are assigned by hand (class Regs) and pinned in the asm statement's
constraints; hipcc only moves the inputs in and the accumulators out.
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "dino_amd", "csrc", "attention_za_gen.inc")

SLOT = 16384                   # bytes per ring slot: K slab (8 KiB) + V slab (8 KiB)
NSLOT = 4
TILE_BYTES = 8192              # one tile of K (or V) rows in HBM: 64 rows x 128 B


class Regs:
    """VGPR map of a wave with MQ blocks of 32 queries (the kernel's constraint list in attention_za.hip mirrors it)."""

    def __init__(self, mq):
        self.mq = mq
        n = 0

        def take(k):
            nonlocal n
            r = n
            n += k
            return r
        self.O = [[take(16) for _ in range(2)] for _ in range(mq)]     # O^T accumulators [query block][d-block]          in/out
        self.Q = [[take(4) for _ in range(4)] for _ in range(mq)]      # Q fragments [query block][k-step]                  in
        self.S = [[take(16) for _ in range(mq)] for _ in range(2)]     # score blocks [parity][query block]                 scratch
        self.P = [[take(8) for _ in range(mq)] for _ in range(2)]      # packed probabilities [parity][query block]; parity 1 enters as zeros
        self.KF = [take(4), take(4)]                                   # K fragment buffers                                 scratch
        self.VF = [take(4), take(4)]                                   # V^T fragment buffers                               scratch
        self.KA = take(4)                                              # LDS addresses of the K fragment reads, k-step 0..3 in
        self.VA = take(4)                                              # LDS addresses of the V^T reads [d-block][half]     in
        self.SOFF = take(1)                                            # lane offset of the LDS-DMA source, next tile       in/out
        self.PS = [take(1) for _ in range(mq)]                         # row-sum partial of the current tile                scratch
        self.L = [take(1) for _ in range(mq)]                          # running row sums                                   in/out
        self.THR = take(1)                                             # valid keys of the last tile minus 8 * (lane >> 5)  in
        self.NINF = take(1)                                            # -inf (a literal and vcc do not fit one VOP2)       scratch
        self.NV = n                                                    # first VGPR the asm does not touch


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


class Emitter:
    def __init__(self, qk_op, mq=1, ablate=0, qbs=None):
        self.qk_op = qk_op
        self.pv_op = "v_mfma_f32_32x32x16_bf16"
        self.R = Regs(mq)
        self.mq = mq
        self.qbs = list(qbs) if qbs is not None else list(range(mq))       # the query blocks this body computes: (0,) on the two-block map
                                       # = the body of a wave whose second block lies past the last row -- same registers, same LDS-DMA
                                       # share, same operand list as the full body (a second pinning would cost the kernel spills)
        self.ablate = ablate           # timing-only builds (wrong results): bit 0 = no V^T fragment reads, bit 1 = no K fragment reads,
                                       # 4 = V fragments by ONE ds_read_b128 (K's address pattern on the V slab) instead of two transposing
                                       # reads, 8 = V reads as they are but the P.V MFMAs take a constant A operand (the Q fragments)
        self.lines = []
        self.q = []                    # outstanding LDS reads, oldest first (tags)
        self.label_state = {}
        self.prev_trans_dst = None     # destination of the instruction just emitted if it was a transcendental
        self.n_inst = 0

    # -- raw emission --
    def raw(self, s, trans_dst=None, reads=()):
        if self.prev_trans_dst is not None and self.prev_trans_dst in reads:
            self._line("s_nop 0")      # gfx940+: a non-transcendental VALU may not read a v_exp result in the next issue slot
        self._line(s)
        self.prev_trans_dst = trans_dst

    def _line(self, s):
        self.lines.append(s)
        self.n_inst += 1

    def comment(self, s):
        self.lines.append(f"; {s}")

    def label(self, name):
        st = tuple(self.q)
        if name in self.label_state:
            assert self.label_state[name] == st, f"label {name}: LDS queue {st} != {self.label_state[name]}"
        self.label_state[name] = st
        self.lines.append(f"{name}_%=:")
        self.prev_trans_dst = None

    def branch(self, op, name):
        """A jump to `name`: the target must see this queue."""
        st = tuple(self.q)
        if name in self.label_state:
            assert self.label_state[name] == st, f"branch to {name}: LDS queue {st} != {self.label_state[name]}"
        else:
            self.label_state[name] = st
        self.raw(f"{op} {name}_%=")

    def set_queue(self, st):
        self.q = list(st)

    # -- LDS --
    def read_k(self, buf, s, slot, half):
        off = (slot % NSLOT) * SLOT + half * 4096
        if self.ablate & 2:
            return
        self.raw(f"ds_read_b128 {vr(buf, 4)}, {vr(self.R.KA + s)} offset:{off}")
        self.q.append(buf)

    def read_v(self, buf, db, ks, slot):
        off = (slot % NSLOT) * SLOT + ks * 2048          # (the V slab's 8 KiB are part of the address registers)
        if self.ablate & 1:
            return
        if self.ablate & 4:
            self.raw(f"ds_read_b128 {vr(buf, 4)}, {vr(self.R.KA + 2 * db)} offset:{off + 8192}")
            self.q += [buf]
            return
        self.raw(f"ds_read_b64_tr_b16 {vr(buf, 2)}, {vr(self.R.VA + 2 * db)} offset:{off}")
        self.raw(f"ds_read_b64_tr_b16 {vr(buf + 2, 2)}, {vr(self.R.VA + 2 * db + 1)} offset:{off}")
        self.q += [buf, buf]

    def need(self, buf):
        if buf not in self.q:
            return                      # (a second consumer of a fragment already waited for, or an ablated read)
        idx = len(self.q) - 1 - self.q[::-1].index(buf)
        n = len(self.q) - 1 - idx
        self.raw(f"s_waitcnt lgkmcnt({n})")
        self.q = self.q[idx + 1:]

    # -- matrix --
    def mfma_qk(self, acc, kbuf, qb, s, first):
        self.need(kbuf)
        c = "0" if first else vr(acc, 16)
        self.raw(f"{self.qk_op} {vr(acc, 16)}, {vr(kbuf, 4)}, {vr(self.R.Q[qb][s], 4)}, {c}")

    def mfma_pv(self, o, vbuf, p, ksl):
        self.need(vbuf)
        a = vr(self.R.Q[0][ksl], 4) if self.ablate & 8 else vr(vbuf, 4)
        self.raw(f"{self.pv_op} {vr(o, 16)}, {a}, {vr(p + 4 * ksl, 4)}, {vr(o, 16)}")

    # -- vector --
    def mask(self, reg, const):
        """element of the last tile: -inf unless its key exists (key-in-tile `const` + 8 * (lane >> 5) < valid keys)"""
        self.raw(f"v_cmp_lt_i32_e32 vcc, {const}, {vr(self.R.THR)}")
        self.raw("s_nop 1")                                # VALU write of vcc -> VALU read as a mask: two wait states
        self.raw(f"v_cndmask_b32_e32 {vr(reg)}, {vr(self.R.NINF)}, {vr(reg)}, vcc")

    def exp(self, reg):
        self.raw(f"v_exp_f32_e32 {vr(reg)}, {vr(reg)}", trans_dst=reg)

    def add(self, dst, a, b):
        self.raw(f"v_add_f32_e32 {vr(dst)}, {vr(a)}, {vr(b)}", reads=(a, b))

    def pack(self, dst, a, b):
        self.raw(f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(a)}, {vr(b)}", reads=(a, b))

    # -- LDS-DMA of one tile into a ring slot: this wave's K pieces and V pieces (MQ = 1: eight waves, one 8-row piece of each slab;
    #    MQ = 2: four waves, two pieces of each slab 32 rows apart = 4 KiB in the source and in the image) --
    def dma_tile(self, slot):
        base = (slot % NSLOT) * SLOT
        for slab, src in ((0, "kb"), (8192, "vb")):
            for j in range(self.mq):
                self.raw(f"s_add_u32 m0, %[lds], {base + slab + 4096 * j}")
                self.raw("s_nop 0")
                self.raw(f"global_load_lds_dwordx4 {vr(self.R.SOFF)}, %[{src}{'2' if j else ''}]")

    def advance_soff(self):
        self.raw(f"v_add_u32_e32 {vr(self.R.SOFF)}, 0x{TILE_BYTES:x}, {vr(self.R.SOFF)}")

    @property
    def dma_per_tile(self):
        return 2 * self.mq


def key_in_tile(r, odd):
    """accumulator register r of a score block holds key (r >> 3) * 16 + 8 * (lane >> 5) + (r & 7) of the block"""
    return (r >> 3) * 16 + (r & 7) + 32 * odd


def stage(e, slot, odd, *, qk=True, pv=True, masked=False, barrier=False, prefetch_k=True, prefetch_v=True, add_l=False):
    """One pipeline stage of the tile in ring slot `slot`; odd = 0: its first 32 keys are exponentiated, 1: its last 32.

    Entry: reads in flight, oldest first: KF[0] (k-step 0), VF[0] (d-block 0), KF[1] (k-step 1), VF[1] (d-block 1) of this stage (those
    that exist).  Eight fragment uses per stage (k-steps 0..3 of the score product alternating with (d-block, key half) of the P.V
    product); each is MQ MFMAs (one per query block), and the gap behind every MFMA carries two elements of ITS query block.
    """
    R, mq = e.R, e.mq
    s_cur, s_nxt = R.S[odd], R.S[odd ^ 1]
    p_cur, p_prv = R.P[odd], R.P[odd ^ 1]
    # operands of this stage: K of the NEXT block (same tile, second half / next tile, first half), V of the PREVIOUS block
    k_slot, k_half = (slot + 1, 0) if odd else (slot, 1)
    v_slot, v_ks0 = (slot, 0) if odd else (slot - 1, 2)
    # operands of the stage after this one
    nk_slot, nk_half = (slot + 1, 1) if odd else (slot + 1, 0)
    nv_slot, nv_ks0 = (slot, 2) if odd else (slot, 0)
    e.comment(f"---- stage: slot {slot % NSLOT}, {'odd' if odd else 'even'}{' masked' if masked else ''}"
              f"{'' if qk else ' no-QK'}{'' if pv else ' no-PV'}")
    for g in range(8):
        for qb in e.qbs:
            last = qb == e.qbs[-1]
            # ---- the MFMA in front of this gap ----
            if g % 2 == 0:
                if qk:
                    e.mfma_qk(s_nxt[qb], R.KF[(g // 2) & 1], qb, g // 2, first=(g == 0))
            else:
                if pv:
                    db, ksl = (g // 2) & 1, g // 4
                    e.mfma_pv(R.O[qb][db], R.VF[db], p_prv[qb], ksl)
            # ---- the gap: elements 2g, 2g + 1 of the current block of query block qb ----
            a, b = s_cur[qb] + 2 * g, s_cur[qb] + 2 * g + 1
            if masked:
                e.mask(a, key_in_tile(2 * g, odd))
                e.mask(b, key_in_tile(2 * g + 1, odd))
            e.exp(a)
            e.exp(b)
            if barrier and g == 4 and last:
                # tile t + 1 is about to be read: this wave's pieces of it have landed, then everyone's; tile t + 2 goes into the slot
                # whose tile was last read two barriers ago
                e.raw("s_waitcnt vmcnt(0)")
                e.raw("s_barrier")
                e.raw("s_cmp_lt_u32 %[cnt], 2")
                e.branch("s_cbranch_scc1", f"NODMA{slot % NSLOT}")
                e.dma_tile(slot + 2)
                e.label(f"NODMA{slot % NSLOT}")
            # fragment reads behind the LAST use of the buffer: fragment uses 0-3 reload for the second half of this stage, 4-7 for
            # the first half of the next one
            if last:
                if g == 0 and qk:
                    e.read_k(R.KF[0], 2, k_slot, k_half)
                elif g == 1 and pv:
                    e.read_v(R.VF[0], 0, v_ks0 + 1, v_slot)
                elif g == 2 and qk:
                    e.read_k(R.KF[1], 3, k_slot, k_half)
                elif g == 3 and pv:
                    e.read_v(R.VF[1], 1, v_ks0 + 1, v_slot)
                elif g == 4 and prefetch_k:
                    e.read_k(R.KF[0], 0, nk_slot, nk_half)
                elif g == 5 and prefetch_v:
                    e.read_v(R.VF[0], 0, nv_ks0, nv_slot)
                elif g == 6 and prefetch_k:
                    e.read_k(R.KF[1], 1, nk_slot, nk_half)
                elif g == 7 and prefetch_v:
                    e.read_v(R.VF[1], 1, nv_ks0, nv_slot)
                if barrier and g == 5:
                    e.advance_soff()                        # (harmless when the DMA was skipped: no tile follows)
            if g == 0 and not odd:
                e.add(R.PS[qb], a, b)                       # a fresh partial sum per tile: (0 + a) + b
            else:
                e.add(R.PS[qb], R.PS[qb], a)
                e.add(R.PS[qb], R.PS[qb], b)
            e.pack(p_cur[qb] + g, a, b)
    if add_l:
        for qb in e.qbs:
            e.add(R.L[qb], R.L[qb], R.PS[qb])


def generate(qk_op, mq=1, ablate=0, qbs=None):
    e = Emitter(qk_op, mq, ablate, qbs)
    R = e.R
    e.comment("GENERATED by tools/gen_attn_asm.py -- do not edit")
    e.raw("s_mov_b32 %[m0s], m0")
    # ---- prologue: tiles 0 and 1 were issued by the kernel AHEAD of its Q loads (their latencies overlap; the Q wait covers them: one
    #      in-order counter), SOFF enters two tiles on ----
    e.raw("s_waitcnt vmcnt(0)")
    e.raw("s_waitcnt lgkmcnt(0)")          # (the zero fill of slot 3's last V rows, written by the kernel in front of this statement)
    e.raw("s_barrier")
    # ---- head: S(0) = K(0) Q^T, nothing to overlap it with ----
    e.read_k(R.KF[0], 0, 0, 0)
    e.read_k(R.KF[1], 1, 0, 0)
    for s in range(4):
        for qb in e.qbs:
            e.mfma_qk(R.S[0][qb], R.KF[s & 1], qb, s, first=(s == 0))
        if s < 2:
            e.read_k(R.KF[s & 1], s + 2, 0, 0)
    # fragments of the first stage (tile 0, even): K of block 1, V of "block -1" = slot 3's zeroed rows against P = 0
    e.read_k(R.KF[0], 0, 0, 1)
    e.read_v(R.VF[0], 0, 2, -1)
    e.read_k(R.KF[1], 1, 0, 1)
    e.read_v(R.VF[1], 1, 2, -1)
    e.raw("s_nop 7")                       # S(0) is read by the first stage's v_exp: one MFMA issue + this cover the 8-pass result latency
    entry = tuple(e.q)
    e.raw("s_cmp_eq_u32 %[cnt], 0")
    e.branch("s_cbranch_scc1", "FINAL0")
    # ---- plain tiles: every tile but the last, four per loop iteration (the ring slot is an immediate) ----
    for s in range(NSLOT):
        e.label(f"PLAIN{s}")
        stage(e, s, 0, barrier=True)
        e.raw("s_sub_u32 %[cnt], %[cnt], 1")
        stage(e, s, 1, add_l=True)
        assert tuple(e.q) == entry
        e.raw("s_cmp_eq_u32 %[cnt], 0")
        e.branch("s_cbranch_scc1", f"FINAL{(s + 1) % NSLOT}")
        if s == NSLOT - 1:
            e.branch("s_branch", "PLAIN0")
    # ---- the last tile (keys beyond ntok masked; no block follows it) and the tail O^T += V(last)^T P(last)^T ----
    for s in range(NSLOT):
        e.set_queue(entry)
        e.label(f"FINAL{s}")
        e.raw(f"v_mov_b32_e32 {vr(R.NINF)}, 0xff800000")
        stage(e, s, 0, masked=True, prefetch_k=False)
        e.raw("s_nop 7")                   # (no score MFMA leads the next stage: cover the latency of the last S block)
        stage(e, s, 1, masked=True, qk=False, prefetch_k=False, add_l=True)
        e.comment("---- tail")
        for ksl in range(2):
            for db in range(2):
                for qb in e.qbs:
                    e.mfma_pv(R.O[qb][db], R.VF[db], R.P[1][qb], ksl)
                if ksl == 0:
                    e.read_v(R.VF[db], db, 3, s)
        assert not e.q
        if s != NSLOT - 1:
            e.branch("s_branch", "END")
    e.set_queue(())
    e.label("END")
    e.raw("s_mov_b32 m0, %[m0s]")
    e.raw("s_nop 15")                      # the accumulators are read by compiler code next
    e.raw("s_nop 7")
    return e


BODIES = (("AZA_BODY_BF16", "v_mfma_f32_32x32x16_bf16", 1, 0), ("AZA_BODY_FP16", "v_mfma_f32_32x32x16_f16", 1, 0),
          ("AZA2_BODY_BF16", "v_mfma_f32_32x32x16_bf16", 2, 0), ("AZA2_BODY_FP16", "v_mfma_f32_32x32x16_f16", 2, 0),
          ("AZA2L_BODY_BF16", "v_mfma_f32_32x32x16_bf16", -1, 0), ("AZA2L_BODY_FP16", "v_mfma_f32_32x32x16_f16", -1, 0),
          ("AZA_BODY_ABL1", "v_mfma_f32_32x32x16_bf16", 1, 1), ("AZA_BODY_ABL2", "v_mfma_f32_32x32x16_bf16", 1, 2),
          ("AZA_BODY_ABL3", "v_mfma_f32_32x32x16_bf16", 1, 4), ("AZA_BODY_ABL4", "v_mfma_f32_32x32x16_bf16", 1, 8))


# -- MFMA write -> vector read hazard (hipcc pads nothing inside an asm statement) --
_RE_RANGE = re.compile(r"v\[(\d+):(\d+)\]")
_RE_ONE = re.compile(r"\bv(\d+)\b")
MFMA_TO_VALU_WAIT_STATES = 11      # an 8-pass XDL result read by a VALU / LDS / memory instruction (CDNA3 ISA guide, data hazards)
MFMA_TO_END_WAIT_STATES = 18       # ... and by whatever the compiler places behind the statement


def _regs(tok):
    out = set()
    for a, b in _RE_RANGE.findall(tok):
        out |= set(range(int(a), int(b) + 1))
    for a in _RE_ONE.findall(_RE_RANGE.sub("", tok)):
        out.add(int(a))
    return out


def check_mfma_hazards(name, lines):
    """Every register an MFMA writes must not be read by a non-MFMA instruction fewer than MFMA_TO_VALU_WAIT_STATES wait states later, in
    text order (an instruction = one wait state, `s_nop k` = k + 1: a lower bound of the real distance, an MFMA issue takes longer), nor sit
    closer than MFMA_TO_END_WAIT_STATES to the end of the body.  Returns the smallest distance found."""
    last, clock, worst = {}, 0, (10 ** 9, None)
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        if op == "s_nop":
            clock += int(ops[0]) + 1
            continue
        if op.startswith("v_mfma"):
            clock += 1
            for r in _regs(ops[0]):
                last[r] = clock
            continue
        if op.startswith(("v_", "ds_", "global_")):
            first_src = 0 if op.startswith(("global_store", "ds_write")) else 1
            for tok in ops[first_src:]:
                for r in _regs(tok):
                    if r in last and clock - last[r] < worst[0]:
                        worst = (clock - last[r], t)
            if op.startswith("v_") and ops:
                for r in _regs(ops[0]):
                    last.pop(r, None)
        clock += 1
    assert worst[0] >= MFMA_TO_VALU_WAIT_STATES, f"{name}: `{worst[1]}` reads an MFMA result {worst[0]} wait states after its write"
    tail = min((clock - c for c in last.values()), default=10 ** 9)
    assert tail >= MFMA_TO_END_WAIT_STATES, f"{name}: an MFMA result is {tail} wait states old at the end of the body"
    return worst[0]


def render():
    out = ["// GENERATED by tools/gen_attn_asm.py -- do not edit (python tools/gen_attn_asm.py rewrites it; --check compares).",
           "// The tile loop of attn_fwd_za_kernel (attention_za.hip) as one inline-asm body per (queries per wave, operand format); register",
           "// map, pipeline and the lgkmcnt bookkeeping are described in the generator.  AZA_BODY_ABL*: timing-only ablations (-DAZA_ABLATIONS).", ""]
    for name, op, mq, abl in BODIES:
        e = generate(op, mq, abl) if mq > 0 else generate(op, 2, abl, qbs=(0,))      # (-1: the one-block body on the two-block map)
        check_mfma_hazards(name, e.lines)
        if abl:
            out.append("#ifdef AZA_ABLATIONS")
        out.append(f"// {e.n_inst} instructions")
        out.append(f"#define {name} \\")
        body = [ln for ln in e.lines]
        for i, ln in enumerate(body):
            esc = ln.replace("\\", "\\\\").replace('"', '\\"')
            out.append(f'    "{esc}\\n\\t"{"" if i == len(body) - 1 else " "}\\')
        out[-1] = out[-1].rstrip("\\").rstrip()
        if abl:
            out.append("#endif")
        out.append("")
    for mq in (1, 2):
        out.append(f"#define AZA{'' if mq == 1 else '2'}_FIRST_FREE_VGPR {Regs(mq).NV}")
    out.append("")
    for mq in (1, 2):
        out += operands(mq)
    for name, op, abl in (("AZA3_BODY_BF16", "v_mfma_f32_32x32x16_bf16", 0), ("AZA3_BODY_FP16", "v_mfma_f32_32x32x16_f16", 0),
                          ("AZA3_BODY_ABL1", "v_mfma_f32_32x32x16_bf16", 16), ("AZA3_BODY_ABL2", "v_mfma_f32_32x32x16_bf16", 32),
                          ("AZA3_BODY_ABL3", "v_mfma_f32_32x32x16_bf16", 64), ("AZA3_BODY_ABL4", "v_mfma_f32_32x32x16_bf16", 128)):
        e = generate3(op, abl)
        check_mfma_hazards(name, e.lines)
        if abl:
            out.append("#ifdef AZA_ABLATIONS")
        out.append(f"// {e.n_inst} instructions")
        out.append(f"#define {name} \\")
        for i, ln in enumerate(e.lines):
            esc = ln.replace("\\", "\\\\").replace('"', '\\"')
            out.append(f'    "{esc}\\n\\t"{"" if i == len(e.lines) - 1 else " "}\\')
        out[-1] = out[-1].rstrip("\\").rstrip()
        if abl:
            out.append("#endif")
        out.append("")
    out.append(f"#define AZA3_FIRST_FREE_VGPR {Regs3().NV}")
    out.append("")
    out += operands3()
    return "\n".join(out)


def operands(mq):
    """The asm statement's operand lists for the kernel's variable names (attention_za.hip): o[qb][db], pz[qb] (zeros), soff, l_run[qb],
    cnt, m0s | qf[qb][s], ka_abs[s], va_abs[i], thr, kb (kb2), vb (vb2), lds_piece | clobbers = every scratch register of the map."""
    R = Regs(mq)
    pin = lambda base, n=1: "{" + vr(base, n) + "}"
    outs = [f'"+{pin(R.O[qb][db], 16)}"(o[qb][db])'.replace("qb", str(qb)).replace("db", str(db)) for qb in range(mq) for db in range(2)]
    outs += [f'"+{pin(R.P[1][qb] + 4 * h, 4)}"(pz[{2 * qb + h}])' for qb in range(mq) for h in range(2)]
    outs += [f'"+{pin(R.SOFF)}"(soff)'] + [f'"+{pin(R.L[qb])}"(l_run[{qb}])' for qb in range(mq)]
    outs += ['[cnt] "+s"(cnt)', '[m0s] "=&s"(m0s)']
    ins = [f'"{pin(R.Q[qb][s], 4)}"(qf[{qb}][{s}])' for qb in range(mq) for s in range(4)]
    ins += [f'"{pin(R.KA + s)}"(ka_abs[{s}])' for s in range(4)] + [f'"{pin(R.VA + i)}"(va_abs[{i}])' for i in range(4)]
    ins += [f'"{pin(R.THR)}"(thr)', '[kb] "s"(kb)', '[vb] "s"(vb)']
    if mq == 2:
        ins += ['[kb2] "s"(kb2)', '[vb2] "s"(vb2)']
    ins += ['[lds] "s"(lds_piece)']
    scratch = []
    for par in range(2):
        for qb in range(mq):
            scratch += list(range(R.S[par][qb], R.S[par][qb] + 16))
    for qb in range(mq):
        scratch += list(range(R.P[0][qb], R.P[0][qb] + 8))
    for b in R.KF + R.VF:
        scratch += list(range(b, b + 4))
    scratch += R.PS + [R.NINF]
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"v{r}"' for r in sorted(scratch)]

    return _operand_macro(f"AZA{'' if mq == 1 else '2'}_OPERANDS", outs, ins, clob)


def _operand_macro(name, outs, ins, clob):
    def wrap(items, indent="      "):
        lines, cur = [], indent
        for it in items:
            if len(cur) + len(it) + 2 > 150:
                lines.append(cur.rstrip() + " \\")
                cur = indent
            cur += it + ", "
        lines.append(cur.rstrip().rstrip(","))
        return lines
    out = [f"#define {name} \\"]
    o = wrap(outs)
    o[0] = "    : " + o[0].lstrip()
    i = wrap(ins)
    i[0] = "    : " + i[0].lstrip()
    c = wrap(clob)
    c[0] = "    : " + c[0].lstrip()
    for grp in (o, i):
        out += grp[:-1] + [grp[-1] + " \\"]
    out += c
    out.append("")
    return out


# =====================================================================================================================================
# hi + lo operand planes (the parity modes: three MFMAs per product).  Same pipeline, one 32-query block per wave, eight waves, ONE
# workgroup per CU (two waves per SIMD): a ring slot is [K hi][V hi][K lo][V lo] = 32 KiB, four slots = 128 KiB.  Per 32-key block and
# fragment use the three products are, in attention_z.hip's order, lo.hi, hi.lo, hi.hi (K x Q for the scores, V^T x P for O); the
# probabilities are split in registers: hi = bf16(p), lo = bf16(p - hi) (common.h split_bf16x2: v_cvt_pk, shift / mask, two subtractions,
# v_cvt_pk).  The score product's operand format is the template's (bf16 or fp16 hi + lo planes); P and V are bf16 hi + lo planes in
# both (2^S against the fixed reference 0 lives on bf16's exponent range).
# =====================================================================================================================================
SLOT3 = 32768


class Regs3:
    def __init__(self):
        n = 0

        def take(k):
            nonlocal n
            r = n
            n += k
            return r
        self.O = [take(16), take(16)]                                  # O^T accumulators [d-block]                         in/out
        self.QH = [take(4) for _ in range(4)]                          # Q fragments, hi plane [k-step]                     in
        self.QL = [take(4) for _ in range(4)]                          #              lo plane                              in
        self.S = [take(16), take(16)]                                  # score blocks [parity]                              scratch
        self.PH = [take(8), take(8)]                                   # packed probabilities hi [parity]; parity 1 enters as zeros
        self.PL = [take(8), take(8)]                                   #                      lo
        self.KFH = [take(4), take(4)]                                  # K fragment buffers hi / lo                         scratch
        self.KFL = [take(4), take(4)]
        self.VFH = [take(4), take(4)]                                  # V^T fragment buffers hi / lo                       scratch
        self.VFL = [take(4), take(4)]
        self.KA = [take(4), take(4)]                                   # LDS addresses of the K reads [slot pair][k-step]   in
        self.VA = [take(4), take(4)]                                   # LDS addresses of the V^T reads [slot pair][db][h]  in
        self.SOFF = take(1)
        self.PS = take(1)
        self.L = take(1)
        self.THR = take(1)
        self.NINF = take(1)
        self.T = [take(1), take(1)]                                    # the split's temporaries                            scratch
        self.NV = n


class Emitter3(Emitter):
    def __init__(self, qk_op, ablate=0):
        super().__init__(qk_op, 1, 0)
        self.R = Regs3()
        self.abl3 = ablate            # timing-only builds of the hi + lo body: 16 = no tile barrier, 32 = no LDS-DMA in the loop,
                                      # 64 = no exp / add / split (the vector work), 128 = no fragment reads

    def read_k3(self, i, s, slot, half):
        if self.abl3 & 128:
            return
        slot %= NSLOT
        off = (slot & 1) * SLOT3 + half * 4096
        a = vr(self.R.KA[slot >> 1] + s)
        self.raw(f"ds_read_b128 {vr(self.R.KFL[i], 4)}, {a} offset:{off + 16384}")
        self.raw(f"ds_read_b128 {vr(self.R.KFH[i], 4)}, {a} offset:{off}")
        self.q += [self.R.KFL[i], self.R.KFH[i]]

    def read_v3(self, db, ks, slot):
        if self.abl3 & 128:
            return
        slot %= NSLOT
        off = (slot & 1) * SLOT3 + ks * 2048
        a0, a1 = vr(self.R.VA[slot >> 1] + 2 * db), vr(self.R.VA[slot >> 1] + 2 * db + 1)
        for buf, o in ((self.R.VFL[db], off + 16384), (self.R.VFH[db], off)):
            self.raw(f"ds_read_b64_tr_b16 {vr(buf, 2)}, {a0} offset:{o}")
            self.raw(f"ds_read_b64_tr_b16 {vr(buf + 2, 2)}, {a1} offset:{o}")
            self.q += [buf, buf]

    def mfma(self, op, acc, a, b, first=False):
        self.need(a)
        self.raw(f"{op} {vr(acc, 16)}, {vr(a, 4)}, {vr(b, 4)}, {'0' if first else vr(acc, 16)}")

    def dma_tile3(self, slot):
        base = (slot % NSLOT) * SLOT3
        for off, src in ((0, "kb"), (8192, "vb"), (16384, "kbl"), (24576, "vbl")):
            self.raw(f"s_add_u32 m0, %[lds], {base + off}")
            self.raw("s_nop 0")
            self.raw(f"global_load_lds_dwordx4 {vr(self.R.SOFF)}, %[{src}]")


def stage3(e, slot, odd, *, qk=True, pv=True, masked=False, barrier=False, prefetch_k=True, prefetch_v=True, add_l=False):
    R = e.R
    s_cur, s_nxt = R.S[odd], R.S[odd ^ 1]
    ph_cur, pl_cur, ph_prv, pl_prv = R.PH[odd], R.PL[odd], R.PH[odd ^ 1], R.PL[odd ^ 1]
    k_slot, k_half = (slot + 1, 0) if odd else (slot, 1)
    v_slot, v_ks0 = (slot, 0) if odd else (slot - 1, 2)
    nk_slot, nk_half = (slot + 1, 1) if odd else (slot + 1, 0)
    nv_slot, nv_ks0 = (slot, 2) if odd else (slot, 0)
    e.comment(f"---- stage (hi+lo): slot {slot % NSLOT}, {'odd' if odd else 'even'}{' masked' if masked else ''}"
              f"{'' if qk else ' no-QK'}{'' if pv else ' no-PV'}")
    for g in range(8):
        a, b = s_cur + 2 * g, s_cur + 2 * g + 1
        for m in range(3):
            # ---- the MFMA in front of this gap: lo.hi, hi.lo, hi.hi ----
            if g % 2 == 0:
                if qk:
                    i, s = (g // 2) & 1, g // 2
                    kf = R.KFL[i] if m == 0 else R.KFH[i]
                    qf = R.QL[s] if m == 1 else R.QH[s]
                    e.mfma(e.qk_op, s_nxt, kf, qf, first=(g == 0 and m == 0))
            else:
                if pv:
                    db, ksl = (g // 2) & 1, g // 4
                    vf = R.VFL[db] if m == 0 else R.VFH[db]
                    pf = (pl_prv if m == 1 else ph_prv) + 4 * ksl
                    e.mfma(e.pv_op, R.O[db], vf, pf)
            # ---- the gap ----
            vec = not (e.abl3 & 64)
            if m == 0 and not vec:
                if barrier and g == 4:
                    e.raw("s_waitcnt vmcnt(0)")
                    if not (e.abl3 & 16):
                        e.raw("s_barrier")
                    e.raw("s_cmp_lt_u32 %[cnt], 2")
                    e.branch("s_cbranch_scc1", f"NODMA{slot % NSLOT}")
                    if not (e.abl3 & 32):
                        e.dma_tile3(slot + 2)
                    e.label(f"NODMA{slot % NSLOT}")
            elif m == 1 and not vec:
                pass
            elif m == 0:
                if masked:
                    e.mask(a, key_in_tile(2 * g, odd))
                    e.mask(b, key_in_tile(2 * g + 1, odd))
                e.exp(a)
                e.exp(b)
                if barrier and g == 4:
                    e.raw("s_waitcnt vmcnt(0)")
                    if not (e.abl3 & 16):
                        e.raw("s_barrier")
                    e.raw("s_cmp_lt_u32 %[cnt], 2")
                    e.branch("s_cbranch_scc1", f"NODMA{slot % NSLOT}")
                    if not (e.abl3 & 32):
                        e.dma_tile3(slot + 2)
                    e.label(f"NODMA{slot % NSLOT}")
                if g == 0 and not odd:
                    e.add(R.PS, a, b)
                else:
                    e.add(R.PS, R.PS, a)
            elif m == 1:
                if not (g == 0 and not odd):
                    e.add(R.PS, R.PS, b)
                e.pack(ph_cur + g, a, b)
                e.raw(f"v_lshlrev_b32_e32 {vr(R.T[0])}, 16, {vr(ph_cur + g)}")
                e.raw(f"v_and_b32_e32 {vr(R.T[1])}, 0xffff0000, {vr(ph_cur + g)}")
            else:
                if vec:
                    e.raw(f"v_sub_f32_e32 {vr(a)}, {vr(a)}, {vr(R.T[0])}")
                    e.raw(f"v_sub_f32_e32 {vr(b)}, {vr(b)}, {vr(R.T[1])}")
                    e.pack(pl_cur + g, a, b)
                # fragment reads behind the last use of the buffers
                if g == 0 and qk:
                    e.read_k3(0, 2, k_slot, k_half)
                elif g == 1 and pv:
                    e.read_v3(0, v_ks0 + 1, v_slot)
                elif g == 2 and qk:
                    e.read_k3(1, 3, k_slot, k_half)
                elif g == 3 and pv:
                    e.read_v3(1, v_ks0 + 1, v_slot)
                elif g == 4 and prefetch_k:
                    e.read_k3(0, 0, nk_slot, nk_half)
                elif g == 5 and prefetch_v:
                    e.read_v3(0, nv_ks0, nv_slot)
                elif g == 6 and prefetch_k:
                    e.read_k3(1, 1, nk_slot, nk_half)
                elif g == 7 and prefetch_v:
                    e.read_v3(1, nv_ks0, nv_slot)
                if barrier and g == 5:
                    e.advance_soff()
    if add_l:
        e.add(R.L, R.L, R.PS)


def generate3(qk_op, ablate=0):
    e = Emitter3(qk_op, ablate)
    R = e.R
    e.comment("GENERATED by tools/gen_attn_asm.py -- do not edit (hi + lo planes)")
    e.raw("s_mov_b32 %[m0s], m0")
    e.raw("s_waitcnt vmcnt(0)")            # (tiles 0 / 1: issued by the kernel ahead of its Q loads, as in the single-plane bodies)
    e.raw("s_waitcnt lgkmcnt(0)")
    e.raw("s_barrier")
    # head: S(0)
    e.read_k3(0, 0, 0, 0)
    e.read_k3(1, 1, 0, 0)
    for s in range(4):
        i = s & 1
        e.mfma(e.qk_op, R.S[0], R.KFL[i], R.QH[s], first=(s == 0))
        e.mfma(e.qk_op, R.S[0], R.KFH[i], R.QL[s])
        e.mfma(e.qk_op, R.S[0], R.KFH[i], R.QH[s])
        if s < 2:
            e.read_k3(i, s + 2, 0, 0)
    e.read_k3(0, 0, 0, 1)
    e.read_v3(0, 2, -1)
    e.read_k3(1, 1, 0, 1)
    e.read_v3(1, 2, -1)
    e.raw("s_nop 7")
    entry = tuple(e.q)
    e.raw("s_cmp_eq_u32 %[cnt], 0")
    e.branch("s_cbranch_scc1", "FINAL0")
    for s in range(NSLOT):
        e.label(f"PLAIN{s}")
        stage3(e, s, 0, barrier=True)
        e.raw("s_sub_u32 %[cnt], %[cnt], 1")
        stage3(e, s, 1, add_l=True)
        assert tuple(e.q) == entry
        e.raw("s_cmp_eq_u32 %[cnt], 0")
        e.branch("s_cbranch_scc1", f"FINAL{(s + 1) % NSLOT}")
        if s == NSLOT - 1:
            e.branch("s_branch", "PLAIN0")
    for s in range(NSLOT):
        e.set_queue(entry)
        e.label(f"FINAL{s}")
        e.raw(f"v_mov_b32_e32 {vr(R.NINF)}, 0xff800000")
        stage3(e, s, 0, masked=True, prefetch_k=False)
        e.raw("s_nop 7")
        stage3(e, s, 1, masked=True, qk=False, prefetch_k=False, add_l=True)
        e.comment("---- tail")
        for ksl in range(2):
            for db in range(2):
                e.mfma(e.pv_op, R.O[db], R.VFL[db], R.PH[1] + 4 * ksl)
                e.mfma(e.pv_op, R.O[db], R.VFH[db], R.PL[1] + 4 * ksl)
                e.mfma(e.pv_op, R.O[db], R.VFH[db], R.PH[1] + 4 * ksl)
                if ksl == 0:
                    e.read_v3(db, 3, s)
        assert not e.q
        if s != NSLOT - 1:
            e.branch("s_branch", "END")
    e.set_queue(())
    e.label("END")
    e.raw("s_mov_b32 m0, %[m0s]")
    e.raw("s_nop 15")
    e.raw("s_nop 7")
    return e


def operands3():
    """operand lists of the hi + lo body for attention_za.hip's names: o3[db], pz3[i] (zeros: PH[1], PL[1]), soff, l_run[0], cnt, m0s |
    qh[s], ql[s], ka_abs[set*4+s], va_abs[set*4+i], thr, kb, vb, kbl, vbl, lds_piece"""
    R = Regs3()
    pin = lambda base, n=1: "{" + vr(base, n) + "}"
    outs = [f'"+{pin(R.O[db], 16)}"(o[0][{db}])' for db in range(2)]
    outs += [f'"+{pin(R.PH[1] + 4 * h, 4)}"(pz[{h}])' for h in range(2)] + [f'"+{pin(R.PL[1] + 4 * h, 4)}"(pz[{2 + h}])' for h in range(2)]
    outs += [f'"+{pin(R.SOFF)}"(soff)', f'"+{pin(R.L)}"(l_run[0])', '[cnt] "+s"(cnt)', '[m0s] "=&s"(m0s)']
    ins = [f'"{pin(R.QH[s], 4)}"(qf[0][{s}])' for s in range(4)] + [f'"{pin(R.QL[s], 4)}"(qfl[{s}])' for s in range(4)]
    ins += [f'"{pin(R.KA[t] + s)}"(ka_abs[{4 * t + s}])' for t in range(2) for s in range(4)]
    ins += [f'"{pin(R.VA[t] + i)}"(va_abs[{4 * t + i}])' for t in range(2) for i in range(4)]
    ins += [f'"{pin(R.THR)}"(thr)', '[kb] "s"(kb)', '[vb] "s"(vb)', '[kbl] "s"(kbl)', '[vbl] "s"(vbl)', '[lds] "s"(lds_piece)']
    scratch = list(range(R.S[0], R.S[0] + 32)) + list(range(R.PH[0], R.PH[0] + 8)) + list(range(R.PL[0], R.PL[0] + 8))
    for b in R.KFH + R.KFL + R.VFH + R.VFL:
        scratch += list(range(b, b + 4))
    scratch += [R.PS, R.NINF] + R.T
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"v{r}"' for r in sorted(scratch)]
    return _operand_macro("AZA3_OPERANDS", outs, ins, clob)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--dump", action="store_true", help="print one bf16 body as plain text")
    ap.add_argument("--mq", type=int, default=1)
    ap.add_argument("--regs", action="store_true", help="print the register map")
    args = ap.parse_args()
    if args.regs:
        for k, v in vars(Regs(args.mq)).items():
            print(k, v)
        sys.exit(0)
    if args.dump:
        print("\n".join((generate3("v_mfma_f32_32x32x16_bf16") if args.mq == 3 else generate("v_mfma_f32_32x32x16_bf16", args.mq)).lines))
        sys.exit(0)
    text = render()
    if args.check:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}")
