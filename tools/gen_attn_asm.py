#!/usr/bin/env python3
"""Generator of the hand-scheduled tile loop of dino_amd/csrc/attention_za.hip (writes dino_amd/csrc/attention_za_gen.inc).

    python tools/gen_attn_asm.py            # rewrite the .inc
    python tools/gen_attn_asm.py --check    # exit 1 if the committed .inc is not what this script generates

What is generated: ONE inline-asm body per operand format -- the whole K/V tile loop of a wave of attn_fwd_za_kernel (32 queries, head
dimension 64, 64-key tiles in a four-slot LDS ring) as a software pipeline over 32-key blocks.  Stage b of a wave runs, interleaved
instruction by instruction,

    matrix pipe :  S(b+1) = K(b+1) . Q^T   (4 MFMAs, fresh accumulator)   and   O^T += V(b-1)^T . P(b-1)^T   (4 MFMAs)
    vector port :  P(b) = 2^S(b) in place (16 v_exp_f32), the row sum (16 v_add_f32), the pack to bf16 (8 v_cvt_pk_bf16_f32)
    LDS         :  the K fragments of block b+1 / b+2 and the V^T fragments of block b-1 / b, each four MFMAs ahead of its use

so every MFMA is followed by two exponentials, two adds, one pack and one or two fragment reads ("gap"), and no instruction of a gap
depends on the MFMA in front of it.  Arithmetic, operand orientation, LDS image and summation order are those of attention_z.hip's
attn_fwd_z_kernel<1, 4, 8>: the outputs are bit-identical (tests/test_ops_gpu.py).

Why a generator: the waits.  LDS reads return in order, so the wait in front of an MFMA is `s_waitcnt lgkmcnt(N)` with N = the number of
reads issued after the one the MFMA needs; the emitter keeps the queue of outstanding reads and computes every N, and checks at every
label that all paths arrive with the same queue.  Registers are assigned by hand (the map below) and pinned in the asm statement's
constraints; hipcc only moves the inputs in and the accumulators out.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "dino_amd", "csrc", "attention_za_gen.inc")

# ---- register map (VGPRs; the kernel's constraint list in attention_za.hip mirrors it) -------------------------------------------
O0, O1 = 0, 16                 # O^T accumulators, d-blocks 0 / 1               v[0:15], v[16:31]      in/out
Q = 32                         # four Q fragments (B operand of S^T = K Q^T)    v[32:47]               in
SA, SB = 48, 64                # two score blocks (32 keys x 32 queries)        v[48:63], v[64:79]     scratch
PA, PB = 80, 88                # two packed probability blocks (2 fragments)    v[80:87], v[88:95]     PB in (zeros), both scratch
KFA, KFB = 96, 100             # K fragment buffers                             v[96:99], v[100:103]   scratch
VFA, VFB = 104, 108            # V^T fragment buffers                           v[104:107], v[108:111] scratch
KA = 112                       # LDS addresses of the K fragment reads, s = 0..3  v112..v115           in
VA = 116                       # LDS addresses of the V^T reads [db][h]         v116..v119             in
SOFF = 120                     # lane offset of the LDS-DMA source, next tile   v120                   in/out
PS = 121                       # row-sum partial of the current tile            v121                   scratch
L = 122                        # running row sum                                v122                   in/out
THR = 123                      # valid keys of the last tile minus 8 * (lane >> 5)   v123              in
NINF = 124                     # -inf (the last tile's mask value: a literal and vcc do not fit one VOP2)   v124   scratch
NV = 125                       # first VGPR the asm does not touch

SLOT = 16384                   # bytes per ring slot: K slab (8 KiB) + V slab (8 KiB)
NSLOT = 4
TILE_BYTES = 8192              # one tile of K (or V) rows in HBM: 64 rows x 128 B


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


class Emitter:
    def __init__(self, qk_op):
        self.qk_op = qk_op
        self.pv_op = "v_mfma_f32_32x32x16_bf16"
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
        self.raw(f"ds_read_b128 {vr(buf, 4)}, {vr(KA + s)} offset:{off}")
        self.q.append(buf)

    def read_v(self, buf, db, ks, slot):
        off = (slot % NSLOT) * SLOT + ks * 2048          # (the V slab's 8 KiB are part of the address registers)
        self.raw(f"ds_read_b64_tr_b16 {vr(buf, 2)}, {vr(VA + 2 * db)} offset:{off}")
        self.raw(f"ds_read_b64_tr_b16 {vr(buf + 2, 2)}, {vr(VA + 2 * db + 1)} offset:{off}")
        self.q += [buf, buf]

    def need(self, buf):
        assert buf in self.q, f"fragment buffer v{buf} has no read in flight"
        idx = len(self.q) - 1 - self.q[::-1].index(buf)
        n = len(self.q) - 1 - idx
        self.raw(f"s_waitcnt lgkmcnt({n})")
        self.q = self.q[idx + 1:]

    # -- matrix --
    def mfma_qk(self, acc, kbuf, s, first):
        self.need(kbuf)
        c = "0" if first else vr(acc, 16)
        self.raw(f"{self.qk_op} {vr(acc, 16)}, {vr(kbuf, 4)}, {vr(Q + 4 * s, 4)}, {c}")

    def mfma_pv(self, o, vbuf, p, ksl):
        self.need(vbuf)
        self.raw(f"{self.pv_op} {vr(o, 16)}, {vr(vbuf, 4)}, {vr(p + 4 * ksl, 4)}, {vr(o, 16)}")

    # -- vector --
    def mask(self, reg, const):
        """element of the last tile: -inf unless its key exists (key-in-tile `const` + 8 * (lane >> 5) < valid keys)"""
        self.raw(f"v_cmp_lt_i32_e32 vcc, {const}, {vr(THR)}")
        self.raw("s_nop 1")                                # VALU write of vcc -> VALU read as a mask: two wait states
        self.raw(f"v_cndmask_b32_e32 {vr(reg)}, {vr(NINF)}, {vr(reg)}, vcc")

    def exp(self, reg):
        self.raw(f"v_exp_f32_e32 {vr(reg)}, {vr(reg)}", trans_dst=reg)

    def add(self, dst, a, b):
        self.raw(f"v_add_f32_e32 {vr(dst)}, {vr(a)}, {vr(b)}", reads=(a, b))

    def pack(self, dst, a, b):
        self.raw(f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(a)}, {vr(b)}", reads=(a, b))

    # -- LDS-DMA of one tile into a ring slot (this wave's K piece and V piece) --
    def dma_tile(self, slot):
        base = (slot % NSLOT) * SLOT
        self.raw(f"s_add_u32 m0, %[lds], {base}")
        self.raw("s_nop 0")
        self.raw(f"global_load_lds_dwordx4 {vr(SOFF)}, %[kb]")
        self.raw(f"s_add_u32 m0, %[lds], {base + 8192}")
        self.raw("s_nop 0")
        self.raw(f"global_load_lds_dwordx4 {vr(SOFF)}, %[vb]")


def key_in_tile(r, odd):
    """accumulator register r of a score block holds key (r >> 3) * 16 + 8 * (lane >> 5) + (r & 7) of the block"""
    return (r >> 3) * 16 + (r & 7) + 32 * odd


def stage(e, slot, odd, *, qk=True, pv=True, masked=False, barrier=False, prefetch_k=True, prefetch_v=True, next_is_tail=False,
          add_l=False):
    """One pipeline stage of the tile in ring slot `slot`; odd = 0: its first 32 keys are exponentiated, 1: its last 32.

    Entry: reads in flight, oldest first: KFA (s = 0), VFA (d-block 0), KFB (s = 1), VFB (d-block 1) of this stage (those that exist).
    """
    s_cur, s_nxt = (SB, SA) if odd else (SA, SB)
    p_cur, p_prv = (PB, PA) if odd else (PA, PB)
    # operands of this stage: K of the NEXT block (same tile, second half / next tile, first half), V of the PREVIOUS block
    k_slot, k_half = (slot + 1, 0) if odd else (slot, 1)
    v_slot, v_ks0 = (slot, 0) if odd else (slot - 1, 2)
    # operands of the stage after this one
    nk_slot, nk_half = (slot + 1, 1) if odd else (slot + 1, 0)
    nv_slot, nv_ks0 = (slot, 2) if odd else (slot, 0)
    e.comment(f"---- stage: slot {slot % NSLOT}, {'odd' if odd else 'even'}{' masked' if masked else ''}"
              f"{'' if qk else ' no-QK'}{'' if pv else ' no-PV'}")
    kbuf = [KFA, KFB, KFA, KFB]
    for g in range(8):
        # ---- the MFMA in front of gap g ----
        if g % 2 == 0:
            if qk:
                e.mfma_qk(s_nxt, kbuf[g // 2], g // 2, first=(g == 0))
        else:
            if pv:
                db, ksl = (g // 2) & 1, g // 4
                e.mfma_pv(O1 if db else O0, VFB if db else VFA, p_prv, ksl)
        # ---- gap g: elements 2g, 2g + 1 of the current block ----
        a, b = s_cur + 2 * g, s_cur + 2 * g + 1
        if masked:
            e.mask(a, key_in_tile(2 * g, odd))
            e.mask(b, key_in_tile(2 * g + 1, odd))
        e.exp(a)
        e.exp(b)
        if barrier and g == 4:
            # tile t + 1 is about to be read: this wave's pieces of it have landed, then everyone's; tile t + 2 goes into the slot
            # whose tile was last read two barriers ago
            e.raw("s_waitcnt vmcnt(0)")
            e.raw("s_barrier")
            e.raw("s_cmp_lt_u32 %[cnt], 2")
            e.branch("s_cbranch_scc1", f"NODMA{slot % NSLOT}")
            e.dma_tile(slot + 2)
            e.label(f"NODMA{slot % NSLOT}")
        # fragment reads: gaps 0-3 for the second half of this stage, gaps 4-7 for the first half of the next one
        if g == 0 and qk:
            e.read_k(KFA, 2, k_slot, k_half)
        elif g == 1 and pv:
            e.read_v(VFA, 0, v_ks0 + 1, v_slot)
        elif g == 2 and qk:
            e.read_k(KFB, 3, k_slot, k_half)
        elif g == 3 and pv:
            e.read_v(VFB, 1, v_ks0 + 1, v_slot)
        elif g == 4 and prefetch_k:
            e.read_k(KFA, 0, nk_slot, nk_half)
        elif g == 5 and prefetch_v:
            e.read_v(VFA, 0, nv_ks0, nv_slot)
        elif g == 6 and prefetch_k:
            e.read_k(KFB, 1, nk_slot, nk_half)
        elif g == 7 and prefetch_v:
            e.read_v(VFB, 1, nv_ks0, nv_slot)
        if barrier and g == 5:
            e.raw(f"v_add_u32_e32 {vr(SOFF)}, 0x{TILE_BYTES:x}, {vr(SOFF)}")      # (harmless when the DMA was skipped: no tile follows)
        if g == 0 and not odd:
            e.add(PS, a, b)                                 # a fresh partial sum per tile: (0 + a) + b
        else:
            e.add(PS, PS, a)
            e.add(PS, PS, b)
        e.pack(p_cur + g, a, b)
    if add_l:
        e.add(L, L, PS)


def generate(qk_op):
    e = Emitter(qk_op)
    e.comment("GENERATED by tools/gen_attn_asm.py -- do not edit")
    e.raw("s_mov_b32 %[m0s], m0")
    # ---- prologue: tiles 0 and 1 on their way, tile 0 landed ----
    e.dma_tile(0)
    e.raw(f"v_add_u32_e32 {vr(SOFF)}, 0x{TILE_BYTES:x}, {vr(SOFF)}")
    e.raw("s_cmp_lt_u32 %[cnt], 1")
    e.branch("s_cbranch_scc1", "ONE")
    e.dma_tile(1)
    e.raw(f"v_add_u32_e32 {vr(SOFF)}, 0x{TILE_BYTES:x}, {vr(SOFF)}")
    e.raw("s_waitcnt vmcnt(2)")
    e.branch("s_branch", "LANDED")
    e.label("ONE")
    e.raw("s_waitcnt vmcnt(0)")
    e.label("LANDED")
    e.raw("s_waitcnt lgkmcnt(0)")          # (the zero fill of slot 3's last V rows, written by the kernel in front of this statement)
    e.raw("s_barrier")
    # ---- head: S(0) = K(0) Q^T, nothing to overlap it with ----
    e.read_k(KFA, 0, 0, 0)
    e.read_k(KFB, 1, 0, 0)
    e.mfma_qk(SA, KFA, 0, first=True)
    e.read_k(KFA, 2, 0, 0)
    e.mfma_qk(SA, KFB, 1, first=False)
    e.read_k(KFB, 3, 0, 0)
    e.mfma_qk(SA, KFA, 2, first=False)
    e.mfma_qk(SA, KFB, 3, first=False)
    # fragments of the first stage (tile 0, even): K of block 1, V of "block -1" = slot 3's zeroed rows against P = 0
    e.read_k(KFA, 0, 0, 1)
    e.read_v(VFA, 0, 2, -1)
    e.read_k(KFB, 1, 0, 1)
    e.read_v(VFB, 1, 2, -1)
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
        e.raw(f"v_mov_b32_e32 {vr(NINF)}, 0xff800000")
        stage(e, s, 0, masked=True, prefetch_k=False)
        e.raw("s_nop 7")                   # (no score MFMA leads the next stage: cover the latency of the last S block)
        stage(e, s, 1, masked=True, qk=False, prefetch_k=False, add_l=True)
        e.comment("---- tail")
        e.mfma_pv(O0, VFA, PB, 0)
        e.read_v(VFA, 0, 3, s)
        e.mfma_pv(O1, VFB, PB, 0)
        e.read_v(VFB, 1, 3, s)
        e.mfma_pv(O0, VFA, PB, 1)
        e.mfma_pv(O1, VFB, PB, 1)
        assert not e.q
        if s != NSLOT - 1:
            e.branch("s_branch", "END")
    e.set_queue(())
    e.label("END")
    e.raw("s_mov_b32 m0, %[m0s]")
    e.raw("s_nop 15")                      # the accumulators are read by compiler code next
    e.raw("s_nop 7")
    return e


def render():
    out = ["// GENERATED by tools/gen_attn_asm.py -- do not edit (python tools/gen_attn_asm.py rewrites it; --check compares).",
           "// The tile loop of attn_fwd_za_kernel (attention_za.hip) as one inline-asm body per operand format; register map, pipeline",
           "// and the lgkmcnt bookkeeping are described in the generator.", ""]
    for name, op in (("AZA_BODY_BF16", "v_mfma_f32_32x32x16_bf16"), ("AZA_BODY_FP16", "v_mfma_f32_32x32x16_f16")):
        e = generate(op)
        out.append(f"// {e.n_inst} instructions")
        out.append(f"#define {name} \\")
        body = [ln for ln in e.lines]
        for i, ln in enumerate(body):
            esc = ln.replace("\\", "\\\\").replace('"', '\\"')
            out.append(f'    "{esc}\\n\\t"{"" if i == len(body) - 1 else " "}\\')
        out[-1] = out[-1].rstrip("\\").rstrip()
        out.append("")
    out.append(f"#define AZA_FIRST_FREE_VGPR {NV}")
    out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--dump", action="store_true", help="print the bf16 body as plain text")
    args = ap.parse_args()
    if args.dump:
        print("\n".join(generate("v_mfma_f32_32x32x16_bf16").lines))
        sys.exit(0)
    text = render()
    if args.check:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}")
