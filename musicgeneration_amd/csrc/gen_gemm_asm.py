#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 main loop of the weight-gradient ring GEMM (linear.hip: linear_dw_ring4_kernel).

    python musicgeneration_amd/csrc/gen_gemm_asm.py            -> musicgeneration_amd/csrc/linear_dw_ring4_loop.inc

gW[n][k] += sum_m dY[m][n] X[m][k] for one 256 x 256 tile and one M-split: the LDS-DMA ring of linear_dw_ring_kernel (4 stages of
32 rows of dY and X, 32 KB each, images in 64-column sub-tiles with the transposing-read swizzle) driven by FOUR waves, one per SIMD,
each with a 128 x 128 output tile -- 16 accumulator tiles = all 256 AGPRs, owned by the asm block -- instead of eight waves with
128 x 64 tiles: per 16-row k-step a wave reads 8 operand fragments (16 ds_read_b64_tr_b16) for 16 MFMAs, two thirds of the LDS
fragment bytes per MFMA of the eight-wave kernel, and the loop is MFMA-bound by construction: 32 MFMAs per stage with 32 transposing
reads, 8 DMA pieces, one counted wait and one barrier in their shadows.  hipcc cannot hold 256 accumulators for a wave (it moves
them between the register files around every MFMA: rounds 3-4), hence the asm block; prologue / epilogue stay HIP.

Iteration s (stage s in ring slot s % 4):
    MFMAs  1-16  k-step 0 (fragment set F0)   | transposing reads of k-step 1 -> F1; near the end: wait for stage s+1, barrier
    MFMAs 17-32  k-step 1 (F1)                | transposing reads of stage s+1, k-step 0 -> F0; DMA of stage s+3 (8 pieces)
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_sched import COST, Gen, Item, a, chain, regs, s, schedule, v  # noqa: E402

V_L = 16             # lane
V_VOY = 17           # ..20 DMA source offsets of this wave's four pieces of the dY image
V_VOX = 21           # ..24 ... of the X image
V_TA = 25            # ..26 transposing-read addresses in the dY image (column half 0 / 1), slot 0, this wave's first sub-tile
V_TB = 27            # ..28 ... in the X image
V_ONES = 29          # bf16 (1, 1)
V_TA2 = 30           # ..31 the same addresses + 64 KB (ring slots 2, 3: the offset field of a DS instruction has 16 bits)
V_TB2 = 120          # ..121
V_F = (32, 64)       # two fragment sets: X fragments ct = 0..3 (4 regs each), then dY fragments rt = 0..3
V_TMP = 96           # ..119 prologue temporaries
V_FIRST, V_LAST = 16, 121

S_YP, S_XP = 36, 38            # source pointers of the NEXT stage to request (64-bit)
S_YSTEP, S_XSTEP, S_G, S_S, S_LDS, S_W, S_BIAS, S_DY, S_DX = 40, 41, 42, 43, 44, 45, 46, 47, 48
S_T = 50
S_FIRST, S_LAST = 36, 55

RG_STAGE = 32768
NODMA = bool(os.environ.get("MGX_DW4_NODMA"))        # diagnostic builds (wrong results): which resource bounds the loop?
NOMFMA = bool(os.environ.get("MGX_DW4_NOMFMA"))
NOREAD = bool(os.environ.get("MGX_DW4_NOREAD"))
NOSYNC = os.environ.get("MGX_DW4_NOSYNC", "")        # "1": no wait, no barrier; "w": no wait; "b": no barrier


def frag(fs, kind, i):
    return V_F[fs] + (0 if kind == "x" else 16) + 4 * i


def prologue(g: Gen):
    """%20 = LDS address of this wave's parameter block: dwords 0..11 = dY pointer, X pointer (of the split's first rows, 64-bit each),
    bytes per stage of dY, of X, G (stages), lds0, w, do_bias; + 256 + 256 k + 4 lane: voy0..3, vox0..3, ta0..1, tb0..1"""
    g.comment("==== prologue ====")
    g.drain()
    T = V_TMP
    g.valu(f"v_mbcnt_lo_u32_b32 {v(V_L)}, -1, 0", set(), regs("v", V_L))
    g.valu(f"v_mbcnt_hi_u32_b32 {v(V_L)}, -1, {v(V_L)}", regs("v", V_L), regs("v", V_L))
    g.valu(f"v_mov_b32_e32 {v(T + 12)}, %20", set(), regs("v", T + 12))
    for k in range(3):
        g.ds_read(f"ds_read_b128 {v(T + 4 * k, 4)}, {v(T + 12)} offset:{16 * k}", regs("v", T + 12), regs("v", T + 4 * k, 4))
    g.valu(f"v_lshl_add_u32 {v(T + 13)}, {v(V_L)}, 2, {v(T + 12)}", regs("v", V_L) | regs("v", T + 12), regs("v", T + 13))
    tab = [V_VOY, V_VOY + 1, V_VOY + 2, V_VOY + 3, V_VOX, V_VOX + 1, V_VOX + 2, V_VOX + 3, V_TA, V_TA + 1, V_TB, V_TB + 1]
    for k, dst in enumerate(tab):
        g.ds_read(f"ds_read_b32 {v(dst)}, {v(T + 13)} offset:{256 + 256 * k}", regs("v", T + 13), regs("v", dst))
    sc = [S_YP, S_YP + 1, S_XP, S_XP + 1, S_YSTEP, S_XSTEP, S_G, S_LDS, S_W, S_BIAS]
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.nop(1)
    for k, dst in enumerate(sc):
        g.valu(f"v_readfirstlane_b32 {s(dst)}, {v(T + k)}", regs("v", T + k), regs("s", dst))
    g.nop(4)
    g.valu(f"v_mov_b32_e32 {v(V_ONES)}, 0x3f803f80", set(), regs("v", V_ONES))
    for k in range(2):
        g.valu(f"v_add_u32_e32 {v(V_TA2 + k)}, 0x10000, {v(V_TA + k)}", regs("v", V_TA + k), regs("v", V_TA2 + k))
        g.valu(f"v_add_u32_e32 {v(V_TB2 + k)}, 0x10000, {v(V_TB + k)}", regs("v", V_TB + k), regs("v", V_TB2 + k))
    # this wave's pieces q = 4 w .. 4 w + 3 of each image: LDS destination lds0 + slot * 32 KB (+ 16 KB for X) + q * 1 KB
    g.salu(f"s_lshl_b32 {s(S_T)}, {s(S_W)}, 12", regs("s", S_W), regs("s", S_T))
    g.salu(f"s_add_u32 {s(S_DY)}, {s(S_LDS)}, {s(S_T)}", regs("s", S_LDS) | regs("s", S_T), regs("s", S_DY))
    g.salu(f"s_add_u32 {s(S_DX)}, {s(S_DY)}, 16384", regs("s", S_DY), regs("s", S_DX))
    g.salu(f"s_mov_b32 {s(S_S)}, 0", (), regs("s", S_S))
    for st in range(3):
        pieces, adv = dma_items(g, st, st)
        for f in pieces:
            f()
        adv()
    g.wait_vm_tag("dma0")
    g.raw("s_barrier")
    for f in read_items(g, 0, 0, 0):
        f()


def dma_items(g: Gen, slot: int, ahead: int):
    """request the stage the source pointers stand on (stage S_S + ahead, or the last stage again when that is past the end: the ring
    slot it lands in is never read) into ring slot `slot`: one item per piece (M0 + DMA), then the pointers move one stage on
    unless they stand on the last stage"""
    out = []
    for j in range(4):
        for dst, vo, ptr in ((S_DY, V_VOY, S_YP), (S_DX, V_VOX, S_XP)):
            def f(j=j, dst=dst, vo=vo, ptr=ptr):
                g.salu(f"s_add_u32 m0, {s(dst)}, {slot * RG_STAGE + 1024 * j}", regs("s", dst), {"m0"})
                if not NODMA:
                    g.vmem_dma(f"global_load_lds_dwordx4 {v(vo + j)}, {s(ptr, 2)}", f"dma{slot}", regs("v", vo + j) | regs("s", ptr, 2) | {"m0"})
            out.append(f)

    def adv():
        g.salu(f"s_add_u32 {s(S_T)}, {s(S_S)}, {ahead + 1}", regs("s", S_S), regs("s", S_T))
        g.salu(f"s_cmp_lt_u32 {s(S_T)}, {s(S_G)}", regs("s", S_T) | regs("s", S_G), {"scc"})
        g.salu(f"s_cselect_b32 {s(S_T)}, {s(S_YSTEP)}, 0", {"scc"} | regs("s", S_YSTEP), regs("s", S_T))
        g.salu(f"s_cselect_b32 {s(S_T + 1)}, {s(S_XSTEP)}, 0", {"scc"} | regs("s", S_XSTEP), regs("s", S_T + 1))
        g.salu(f"s_add_u32 {s(S_YP)}, {s(S_YP)}, {s(S_T)}", regs("s", S_YP) | regs("s", S_T), regs("s", S_YP))
        g.salu(f"s_addc_u32 {s(S_YP + 1)}, {s(S_YP + 1)}, 0", regs("s", S_YP + 1), regs("s", S_YP + 1))
        g.salu(f"s_add_u32 {s(S_XP)}, {s(S_XP)}, {s(S_T + 1)}", regs("s", S_XP) | regs("s", S_T + 1), regs("s", S_XP))
        g.salu(f"s_addc_u32 {s(S_XP + 1)}, {s(S_XP + 1)}, 0", regs("s", S_XP + 1), regs("s", S_XP + 1))
    return out, adv


def read_items(g: Gen, slot: int, ks: int, fs: int):
    """the eight operand fragments of k-step ks of the stage in `slot` -> fragment set fs (16 transposing reads, in the order the MFMAs
    need them): fragment i of an image = 64-column sub-tile i >> 1 of the wave's two (+ 4 KB), column half i & 1 (address register),
    rows 16 ks + 8 hh + 4 jq + rq"""
    out = []
    for kind, i in (("x", 0), ("y", 0), ("x", 1), ("x", 2), ("x", 3), ("y", 1), ("y", 2), ("y", 3)):
        base = ((V_TB, V_TB2) if kind == "x" else (V_TA, V_TA2))[slot >> 1] + (i & 1)
        for jq in range(2):
            dst = frag(fs, kind, i) + 2 * jq
            off = (slot & 1) * RG_STAGE + 4096 * (i >> 1) + 2048 * ks + 512 * jq
            out.append((lambda: None) if NOREAD else lambda dst=dst, off=off, base=base: g.ds_read(f"ds_read_b64_tr_b16 {v(dst, 2)}, {v(base)} offset:{off}", regs("v", base), regs("v", dst, 2)))
    return out


def body(g: Gen, slot: int, bias: tuple):
    g.comment(f"==== stage in ring slot {slot}{f' (+ bias gradient of fragments {bias})' if bias else ''} ====")
    mf = []
    for fs in range(2):
        for rt in range(4):
            for ct in range(4):
                mf.append((lambda: None) if NOMFMA else lambda fs=fs, rt=rt, ct=ct: g.mfma_op(f"%{4 * rt + ct}", ("v", frag(fs, "x", ct)), ("v", frag(fs, "y", rt))))
    items = []

    def add(fns, cost, spread=0, **kw):
        """spread: the k-th of the n items not before gap earliest + k * spread // n (an even trickle instead of a burst)"""
        out = []
        for k, f in enumerate(fns):
            kw2 = dict(kw)
            kw2["earliest"] = kw.get("earliest", 1) + k * spread // len(fns)
            out.append(Item(f, cost, **kw2))
        items.extend(out)
        return out

    nslot = (slot + 1) & 3
    add(read_items(g, slot, 1, 1), COST["lds"], earliest=1, deadline=13, spread=11, name="rd_k1")          # F1: free since the previous iteration's MFMAs 17-32

    def barrier():
        if NOSYNC not in ("1", "w"):
            g.wait_vm_tag(f"dma{nslot}")                   # this wave's pieces of stage s+1 have landed
        if NOSYNC not in ("1", "b"):
            g.raw("s_barrier")
    bar = add([barrier], COST["sync"], pin=14, name="barrier")
    add(read_items(g, nslot, 0, 0), COST["lds"], earliest=19, deadline=31, spread=11, deps=bar, name="rd_k0")      # F0: read by MFMAs 1-16
    pieces, adv = dma_items(g, (slot + 3) & 3, 3)
    dm = add(pieces, COST["salu"] + COST["vmem"], earliest=15, deadline=28, deps=bar, name="dma")
    add([adv], 8 * COST["salu"], earliest=16, deadline=31, deps=dm, name="adv")
    if bias:
        for fs in range(2):
            fns = []
            for k in range(4):                       # round-robin over the sums: a v_dot2c waits for the previous one into the same register
                for rt in bias:
                    src = frag(fs, "y", rt) + k
                    fns.append(lambda rt=rt, src=src: g.valu(f"v_dot2c_f32_bf16_e32 %{16 + rt}, {v(src)}, {v(V_ONES)}", regs("v", src) | regs("v", V_ONES), set()))
            add(fns, COST["valu"], earliest=(6, 21)[fs], deadline=15 + 16 * fs, spread=(8, 10)[fs], name=f"bias{fs}")
    table, budget = schedule(items, ngaps=32, budget0=8)
    g.comment(f"per-gap issue budget {budget}")
    for gi in range(1, 33):
        mf[gi - 1]()
        for it in table[gi]:
            it.fn()


def loop_tail(g: Gen, slot: int, tag: str):
    g.salu(f"s_add_u32 {s(S_S)}, {s(S_S)}, 1", regs("s", S_S), regs("s", S_S))
    g.salu(f"s_cmp_ge_u32 {s(S_S)}, {s(S_G)}", regs("s", S_S) | regs("s", S_G), {"scc"})
    g.raw("s_cbranch_scc1 L_dw4_end_%=")
    if slot == 3:
        g.raw(f"s_branch L_dw4_{tag}0_%=")


# which of its four dY fragments a wave sums for the bias gradient (parameter word 9): the waves that hold the same dY fragments -- wn = 0, 1
# of every k-tile of a weight's tile row -- share the work (linear.hip: linear_dw_ring4_kernel)
BIAS_VARIANTS = {1: (0, 1, 2, 3), 2: (0, 1), 3: (2, 3), 4: (0,), 5: (1,), 6: (2,), 7: (3,)}


def fixed_point_loop(g: Gen, variant: int):
    bias = BIAS_VARIANTS.get(variant, ())
    tag = f"v{variant}_"
    texts = []
    for rnd in range(3):
        g.out = []
        for slot in range(4):
            g.out.append(f"L_dw4_{tag}{slot}_%=:")
            body(g, slot, bias)
            loop_tail(g, slot, tag)
        texts.append(list(g.out))
    assert NOMFMA or texts[1] == texts[2], "the loop body is not a fixed point of the wait-count / hazard trackers"
    return texts[2]


def generate():
    g = Gen()
    prologue(g)
    g.drain()                                    # (the loops are generated from their own loop-carried state: enter them drained)
    for k in BIAS_VARIANTS:
        g.salu(f"s_cmp_eq_u32 {s(S_BIAS)}, {k}", regs("s", S_BIAS), {"scc"})
        g.raw(f"s_cbranch_scc1 L_dw4_v{k}_0_%=")
    pro = list(g.out)
    # the prologue issued the requests of stages 0..2 and read F0 of stage 0: that is the state the loop body expects at its top, except
    # that everything has been waited for (harmless: counted waits only ever wait longer)
    loops = []
    for k in [0] + list(BIAS_VARIANTS):
        g.drain()
        loops += fixed_point_loop(g, k)
    g.out = ["L_dw4_end_%=:"]
    g.drain()
    g.nop(16)
    return pro + loops + g.out, g


def clobbers():
    c = [f"v{i}" for i in range(V_FIRST, V_LAST + 1)] + [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "m0", "memory"]
    return c


def write(here):
    lines, g = generate()
    path = os.path.join(here, "linear_dw_ring4_loop.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by gen_gemm_asm.py -- do not edit.  The hand-scheduled main loop of linear_dw_ring4_kernel (one asm statement):\n")
        f.write("// operands %0..%15 = acc[rt][ct] (\"+a\": all 256 AGPRs), %16..%19 = the bias-gradient partial sums (\"+v\"), %20 = LDS address of the\n")
        f.write("// wave's parameter block (\"s\").  Register map and schedule: gen_gemm_asm.py.\n")
        f.write("#define MGX_DW4_LOOP_ASM \\\n")
        for ln in lines:
            if ln.startswith(";"):
                f.write(f"    /* {ln[1:].strip()} */ \\\n")
            else:
                f.write(f'    "{ln}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define MGX_DW4_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
    n_ins = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
    print(f"wrote {path}: {n_ins} instructions, s_nop wait states inserted: {g.nops}; counts {g.stats}", file=sys.stderr)


if __name__ == "__main__":
    write(os.path.dirname(os.path.abspath(__file__)))
