#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 main loops of the four-wave ring GEMMs (linear.hip): the weight-gradient kernel
linear_dw_ring4_kernel (first half of this file) and the forward / dX kernel linear_ring4_kernel (second half: ring_tile).

    python musicgeneration_amd/csrc/gen_gemm_asm.py            -> musicgeneration_amd/csrc/linear_dw_ring4_loop.inc, linear_ring4_loop.inc

Diagnostic builds (results are garbage, timing only; tools/dw4_clock.sh, tools/ring4_times.py): MGX_DW4_NODMA / NOREAD / NOMFMA / NOSYNC and
MGX_RING4_DIAG in the environment of the build leave parts of the loops out; such loops are written to linear_*_loop_diag.inc (untracked),
which linear.hip includes instead of the tracked files when built with -DMGX_GEMM_DIAG=1.

Weight gradient:

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
from asm_sched import COST, Gen, Item, a, chain, regs, s, schedule, v, write_if_changed  # noqa: E402

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
RING_DIAG = os.environ.get("MGX_RING4_DIAG", "")     # "noa" / "nob": no DMA of the A / B image; "nora" / "norb": no fragment reads of A / B
NOSYNC = os.environ.get("MGX_DW4_NOSYNC", "")        # "1": no wait, no barrier; "w": no wait; "b": no barrier
# any of the knobs above: the (wrong-result) loops go to linear_*_loop_diag.inc, which linear.hip includes under -DMGX_GEMM_DIAG=1 --
# never into the tracked files a product build reads (an exported knob used to turn the next auto-build into a garbage GEMM library)
DIAG = bool(NODMA or NOMFMA or NOREAD or RING_DIAG or NOSYNC)


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


# =====================================================================================================================================
# Forward / dX ring GEMM with four waves (linear.hip: linear_ring4_kernel<BTRANS, PRE>):  C[256 m][256 n] += A[m][r] B[n][r]  (NT,
# forward) or A[m][r] B[r][n] (NN, dX), one asm statement per TILE: the HIP tile loop around it fills the wave's parameter block and runs
# the epilogue (bias / ReLU / mask / addend, bf16, row-major stores); the LDS-DMA ring runs on across the statement's end -- the requests
# of the next tile's first two stages are in flight during the epilogue -- and its state (source pointers, requests left in the tile)
# lives in the parameter block between two statements.
#
# A stage is 64 reduction columns = 64 KB in one of TWO ring slots: A image [256 m][64 r] in 128-byte rows (rel_attn_common.hpp: image R,
# chunk ^ ((row >> 1) & 7)) at + 0, B image the same (NT) or two [32 r][256 n] images of four 64-column sub-tiles (NN: image T,
# transposing reads as in the dW kernel above) at + 32 KB.  Every DMA instruction fetches 8 rows x 128 bytes = whole cache lines: with
# 32-column stages (64 bytes per row: the eight-wave kernel) a workgroup's L2 -> LDS fill ran at half the rate and set the pace --
# 1,900 cycles per 32 columns against 1,080 with the DMA left out (profiles/r05_ring4_diag.txt).  nd = R / 64 stages per tile, nd even
# and >= 4 (host): a tile starts in slot 0.  Per stage 64 MFMAs in four k-steps on fragment sets F0..F3:
#     k-step 0 (F0)   reads of k-step 1 -> F1 and of the B fragments of k-step 2 -> F2
#     k-step 1 (F1)   reads of the B fragments of k-step 3 -> F3, of the A fragments of k-step 2; barrier 1 (every wave has read B of this
#                     slot): request B of the stage after next into it (8 pieces per wave)
#     k-step 2 (F2)   reads of the A fragments of k-step 3; wait for the other slot's pieces, barrier 2 (... and every wave has read A):
#                     request A of the stage after next
#     k-step 3 (F3)   reads of the next stage's k-step 0 -> F0
# =====================================================================================================================================
RV_L = 16
RV_VOA, RV_VOB = 17, 25          # DMA source offsets of the wave's eight pieces of the A / B image
RV_AA = 33                       # ..40 A fragment addresses [k-step + 4 * slot] (image R: the k-step is an XOR, not an offset)
RV_BA = 41                       # ..48 NT: the same for B; NN: [column half + 2 * slot] (..44)
RV_BIASO = 49
RV_PB = 50                       # ..51 the parameter block's LDS address (kept to the end), + 4 * lane
RV_F = (52, 84, 116, 148)        # fragment sets: A fragments rt = 0..3, then B fragments ct = 0..3
RV_TMP = 180                     # ..203 (entry / exit)
RV_FIRST, RV_LAST = 16, 203
RS_AP, RS_BP, RS_ANEXT, RS_BNEXT = 36, 38, 40, 42
RS_ASTEP, RS_BSTEP, RS_ND, RS_RQ, RS_LDS, RS_W, RS_FIRSTF, RS_AROW8 = 44, 45, 46, 47, 48, 49, 50, 51
RS_BIASP, RS_BROW8, RS_RQB = 52, 54, 55
RS_DA, RS_DB, RS_CNT, RS_PARAM = 56, 57, 58, 59
RS_T = 60                        # ..63
RS_FIRST, RS_LAST = 36, 63
RG_DSTAGE = 65536
EPI_STORES = 32                  # global stores of one wave's epilogue (linear.hip: store_wave_block4): a LOWER bound is what is safe here


def rfrag(fs, kind, i):
    return RV_F[fs] + (0 if kind == "a" else 16) + 4 * i


def ring_b_dst(j, btrans):
    """LDS offset of the wave's B piece j from RS_DB: NT pieces 8 w + j of the [256][64] image; NN piece 4 w + (j & 3) of half j >> 2"""
    return 16384 * (j >> 2) + 1024 * (j & 3) if btrans else 1024 * j


def ring_requests(g: Gen, slot: int, btrans: bool, which: str):
    """request the A ("a") or B ("b") image of the stage that operand's source pointer stands on into ring slot `slot` (one item per
    piece); then the pointer moves one stage on, or to the next tile's first stage when this was the tile's last request"""
    out = []
    dst, vo, ptr, step, nxt, rq = (RS_DA, RV_VOA, RS_AP, RS_ASTEP, RS_ANEXT, RS_RQ) if which == "a" else (RS_DB, RV_VOB, RS_BP, RS_BSTEP, RS_BNEXT, RS_RQB)
    for j in range(8):
        off = 1024 * j if which == "a" else ring_b_dst(j, btrans)

        def f(j=j, off=off):
            g.salu(f"s_add_u32 m0, {s(dst)}, {slot * RG_DSTAGE + off}", regs("s", dst), {"m0"})
            if not (NODMA or (RING_DIAG == "noa" and which == "a") or (RING_DIAG == "nob" and which == "b")):
                g.vmem_dma(f"global_load_lds_dwordx4 {v(vo + j)}, {s(ptr, 2)}", f"dma{which}{slot}", regs("v", vo + j) | regs("s", ptr, 2) | {"m0"})
        out.append(f)

    def adv():
        g.salu(f"s_add_u32 {s(RS_T)}, {s(ptr)}, {s(step)}", regs("s", ptr) | regs("s", step), regs("s", RS_T))
        g.salu(f"s_addc_u32 {s(RS_T + 1)}, {s(ptr + 1)}, 0", regs("s", ptr + 1), regs("s", RS_T + 1))
        g.salu(f"s_sub_u32 {s(rq)}, {s(rq)}, 1", regs("s", rq), regs("s", rq))
        g.salu(f"s_cmp_eq_u32 {s(rq)}, 0", regs("s", rq), {"scc"})
        g.salu(f"s_cselect_b64 {s(ptr, 2)}, {s(nxt, 2)}, {s(RS_T, 2)}", regs("s", nxt, 2) | regs("s", RS_T, 2), regs("s", ptr, 2))
        g.salu(f"s_cselect_b32 {s(rq)}, {s(RS_ND)}, {s(rq)}", regs("s", RS_ND) | regs("s", rq), regs("s", rq))
    return out, adv


def ring_reads(g: Gen, slot: int, ks: int, btrans: bool, kinds="ba"):
    """operand fragments of k-step ks (0..3) of the stage in `slot` -> fragment set ks (kinds: "a", "b" or both, then in the order the
    MFMAs need them)"""
    out = []
    order = (("b", 0), ("a", 0), ("b", 1), ("b", 2), ("b", 3), ("a", 1), ("a", 2), ("a", 3))
    for kind, i in order:
        if kind not in kinds:
            continue
        dst = rfrag(ks, kind, i)
        if NOREAD or (RING_DIAG == "nora" and kind == "a") or (RING_DIAG == "norb" and kind == "b"):
            continue
        if kind == "a" or not btrans:
            base = (RV_AA if kind == "a" else RV_BA) + ks + 4 * slot
            out.append((lambda dst=dst, base=base, off=4096 * i: g.ds_read(f"ds_read_b128 {v(dst, 4)}, {v(base)} offset:{off}", regs("v", base), regs("v", dst, 4)), COST["lds128"]))
        else:
            base = RV_BA + (i & 1) + 2 * slot
            for jq in range(2):
                off = 16384 * (ks >> 1) + 4096 * (i >> 1) + 2048 * (ks & 1) + 512 * jq
                out.append((lambda dst=dst + 2 * jq, base=base, off=off: g.ds_read(f"ds_read_b64_tr_b16 {v(dst, 2)}, {v(base)} offset:{off}", regs("v", base), regs("v", dst, 2)), COST["lds"]))
    return out


def ring_body(g: Gen, slot: int, first: bool, btrans: bool):
    g.comment(f"==== stage in ring slot {slot}{' (first of the tile: C = 0)' if first else ''} ====")
    mf = []
    for ks in range(4):
        for rt in range(4):
            for ct in range(4):
                mf.append(lambda ks=ks, rt=rt, ct=ct: g.mfma_op(f"%{4 * rt + ct}", ("v", rfrag(ks, "b", ct)), ("v", rfrag(ks, "a", rt)), c0=(first and ks == 0)))
    items = []

    def add(fns, spread=0, **kw):
        out = []
        for k, (f, cost) in enumerate(fns):
            kw2 = dict(kw)
            kw2["earliest"] = kw.get("earliest", 1) + k * spread // len(fns)
            out.append(Item(f, cost, **kw2))
        items.extend(out)
        return out

    oslot = slot ^ 1
    # B is read out first (all of its fragments by MFMA 27), A progressively: the two images of a slot are released -- and requested again --
    # half a stage apart, so the workgroup always has a request in flight and each has a stage and more to land (with both requested at
    # once, 64 KB every 64 MFMAs, the fill rate of one CU -- not its latency -- stalled every stage)
    rb1 = add(ring_reads(g, slot, 1, btrans, "b"), earliest=1, deadline=12, spread=6, name="rd_b1")
    ra1 = add(ring_reads(g, slot, 1, btrans, "a"), earliest=2, deadline=13, spread=8, name="rd_a1")
    rb2 = add(ring_reads(g, slot, 2, btrans, "b"), earliest=8, deadline=22, spread=8, name="rd_b2")
    rb3 = add(ring_reads(g, slot, 3, btrans, "b"), earliest=17, deadline=26, spread=8, name="rd_b3")   # F3: read by MFMAs 49-64 of the previous stage
    ra2 = add(ring_reads(g, slot, 2, btrans, "a"), earliest=17, deadline=29, spread=10, name="rd_a2")
    def barrier1():
        # the slot's B image is requested again right after this barrier: this wave's reads of it must have RETURNED, not only been
        # issued (an LDS-DMA piece served by L2 lands 250+ cycles after its issue; a queue of LDS reads can be that long)
        g.prewait(set().union(*[regs("v", rfrag(ks, "b", i), 4) for ks in (1, 2, 3) for i in range(4)]))
        g.raw("s_barrier")
    bar1 = add([(barrier1, COST["sync"])], pin=28, deadline=64, deps=rb1 + rb2 + rb3, name="barrier1")   # every wave has read B
    pb_, advb = ring_requests(g, slot, btrans, "b")
    dmb = add([(f, COST["salu"] + COST["vmem"]) for f in pb_], earliest=29, deadline=40, spread=8, deps=bar1, name="dma_b")
    add([(advb, 6 * COST["salu"])], earliest=37, deadline=63, deps=dmb, name="adv_b")
    ra3 = add(ring_reads(g, slot, 3, btrans, "a"), earliest=30, deadline=41, spread=8, name="rd_a3")

    def barrier2():
        g.wait_vm_tag(f"dmaa{oslot}")                      # this wave's pieces of the next stage have landed (its B image is older)
        g.prewait(set().union(*[regs("v", rfrag(ks, "a", i), 4) for ks in (1, 2, 3) for i in range(4)]))     # ... and its reads of A have returned
        g.raw("s_barrier")
    bar2 = add([(barrier2, COST["sync"])], pin=44, deadline=64, deps=ra1 + ra2 + ra3, name="barrier2")     # ... and every wave has read A
    pa_, adva = ring_requests(g, slot, btrans, "a")
    dma_ = add([(f, COST["salu"] + COST["vmem"]) for f in pa_], earliest=45, deadline=56, spread=8, deps=bar2, name="dma_a")
    add([(adva, 6 * COST["salu"])], earliest=53, deadline=63, deps=dma_, name="adv_a")
    add(ring_reads(g, oslot, 0, btrans), earliest=46, deadline=62, spread=14, deps=bar2, name="rd_k0")   # F0: read by MFMAs 1-16
    table, budget = schedule(items, ngaps=64, budget0=8)
    g.comment(f"per-gap issue budget {budget}")
    for gi in range(1, 65):
        mf[gi - 1]()
        for it in table[gi]:
            it.fn()


def ring_tile(btrans: bool):
    g = Gen()
    T = RV_TMP
    g.comment("==== entry: the ring's state from the parameter block ====")
    # Only LDS / scalar traffic is waited for here (the parameter block the HIP code has just written).  The VMEM queue is NOT
    # drained (round 6; ADVICE r5 -- until then an `s_waitcnt vmcnt(0)` stood here, which made every tile after a workgroup's first
    # wait for its own epilogue's stores and for the next tile's two stages that were left in flight on purpose, and turned the counted
    # waits below into no-ops): what stands in the in-order queue at this point is modelled below, and the first stages wait with
    # counts sized for it.
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.valu(f"v_mbcnt_lo_u32_b32 {v(RV_L)}, -1, 0", set(), regs("v", RV_L))
    g.valu(f"v_mbcnt_hi_u32_b32 {v(RV_L)}, -1, {v(RV_L)}", regs("v", RV_L), regs("v", RV_L))
    g.valu(f"v_mov_b32_e32 {v(RV_PB)}, %16", set(), regs("v", RV_PB))
    g.salu(f"s_mov_b32 {s(RS_PARAM)}, %16", (), regs("s", RS_PARAM))
    for k in range(5):
        g.ds_read(f"ds_read_b128 {v(T + 4 * k, 4)}, {v(RV_PB)} offset:{16 * k}", regs("v", RV_PB), regs("v", T + 4 * k, 4))
    g.valu(f"v_lshl_add_u32 {v(RV_PB + 1)}, {v(RV_L)}, 2, {v(RV_PB)}", regs("v", RV_L) | regs("v", RV_PB), regs("v", RV_PB + 1))
    # lane table: DMA offsets of the wave's pieces 0 and 1 of either image (the others: + j * 8 rows), fragment addresses in slot 0
    tab = [RV_VOA, RV_VOA + 1, RV_VOB, RV_VOB + 1, RV_AA, RV_BA, RV_BA + 1]
    for k, dst in enumerate(tab):
        g.ds_read(f"ds_read_b32 {v(dst)}, {v(RV_PB + 1)} offset:{1024 + 256 * k}", regs("v", RV_PB + 1), regs("v", dst))
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.nop(1)
    sc = [RS_AP, RS_AP + 1, RS_BP, RS_BP + 1, RS_ANEXT, RS_ANEXT + 1, RS_BNEXT, RS_BNEXT + 1, RS_ASTEP, RS_BSTEP, RS_ND, RS_RQ, RS_LDS, RS_W,
          RS_FIRSTF, RS_AROW8, RS_BIASP, RS_BIASP + 1, RS_BROW8, RS_RQB]
    for k, dst in enumerate(sc):
        g.valu(f"v_readfirstlane_b32 {s(dst)}, {v(T + k)}", regs("v", T + k), regs("s", dst))
    g.nop(4)
    # the other pieces' offsets: piece j = piece (j & 1) + (j - (j & 1)) * 8 rows (the images' swizzles repeat every 16 rows);
    # NN B: piece j = piece 0 + (8 (j & 3) + 32 (j >> 2)) rows
    for j in range(7, -1, -1):
        if j >= 2:
            g.salu(f"s_mul_i32 {s(RS_T)}, {s(RS_AROW8)}, {j - (j & 1)}", regs("s", RS_AROW8), regs("s", RS_T))
            g.valu(f"v_add_u32_e32 {v(RV_VOA + j)}, {s(RS_T)}, {v(RV_VOA + (j & 1))}", regs("s", RS_T) | regs("v", RV_VOA + (j & 1)), regs("v", RV_VOA + j))
        if not btrans:
            if j >= 2:
                g.salu(f"s_mul_i32 {s(RS_T)}, {s(RS_BROW8)}, {j - (j & 1)}", regs("s", RS_BROW8), regs("s", RS_T))
                g.valu(f"v_add_u32_e32 {v(RV_VOB + j)}, {s(RS_T)}, {v(RV_VOB + (j & 1))}", regs("s", RS_T) | regs("v", RV_VOB + (j & 1)), regs("v", RV_VOB + j))
        elif j >= 1:
            g.salu(f"s_mul_i32 {s(RS_T)}, {s(RS_BROW8)}, {(j & 3) + 4 * (j >> 2)}", regs("s", RS_BROW8), regs("s", RS_T))
            g.valu(f"v_add_u32_e32 {v(RV_VOB + j)}, {s(RS_T)}, {v(RV_VOB)}", regs("s", RS_T) | regs("v", RV_VOB), regs("v", RV_VOB + j))
    # fragment addresses: k-step ks = address of k-step 0 ^ (32 ks) (the chunk field), slot 1 = + 64 KB
    for ks in range(1, 4):
        g.valu(f"v_xor_b32_e32 {v(RV_AA + ks)}, {32 * ks}, {v(RV_AA)}", regs("v", RV_AA), regs("v", RV_AA + ks))
    for ks in range(4):
        g.valu(f"v_add_u32_e32 {v(RV_AA + 4 + ks)}, 0x10000, {v(RV_AA + ks)}", regs("v", RV_AA + ks), regs("v", RV_AA + 4 + ks))
    if not btrans:
        for ks in range(1, 4):
            g.valu(f"v_xor_b32_e32 {v(RV_BA + ks)}, {32 * ks}, {v(RV_BA)}", regs("v", RV_BA), regs("v", RV_BA + ks))
        for ks in range(4):
            g.valu(f"v_add_u32_e32 {v(RV_BA + 4 + ks)}, 0x10000, {v(RV_BA + ks)}", regs("v", RV_BA + ks), regs("v", RV_BA + 4 + ks))
    else:
        for k in range(2):
            g.valu(f"v_add_u32_e32 {v(RV_BA + 2 + k)}, 0x10000, {v(RV_BA + k)}", regs("v", RV_BA + k), regs("v", RV_BA + 2 + k))
    g.valu(f"v_lshlrev_b32_e32 {v(RV_BIASO)}, 2, {v(RV_L)}", regs("v", RV_L), regs("v", RV_BIASO))
    # LDS destinations of the wave's pieces: A pieces 8 w + j; B: NT the same in the B image, NN pieces 4 w + (j & 3) of either half
    g.salu(f"s_lshl_b32 {s(RS_T)}, {s(RS_W)}, 13", regs("s", RS_W), regs("s", RS_T))
    g.salu(f"s_add_u32 {s(RS_DA)}, {s(RS_LDS)}, {s(RS_T)}", regs("s", RS_LDS) | regs("s", RS_T), regs("s", RS_DA))
    if btrans:
        g.salu(f"s_lshl_b32 {s(RS_T)}, {s(RS_W)}, 12", regs("s", RS_W), regs("s", RS_T))
        g.salu(f"s_add_u32 {s(RS_DB)}, {s(RS_LDS)}, {s(RS_T)}", regs("s", RS_LDS) | regs("s", RS_T), regs("s", RS_DB))
        g.salu(f"s_add_u32 {s(RS_DB)}, {s(RS_DB)}, 32768", regs("s", RS_DB), regs("s", RS_DB))
    else:
        g.salu(f"s_add_u32 {s(RS_DB)}, {s(RS_DA)}, 32768", regs("s", RS_DA), regs("s", RS_DB))
    g.salu(f"s_lshr_b32 {s(RS_CNT)}, {s(RS_ND)}, 1", regs("s", RS_ND), regs("s", RS_CNT))
    g.salu(f"s_sub_u32 {s(RS_CNT)}, {s(RS_CNT)}, 1", regs("s", RS_CNT), regs("s", RS_CNT))       # loop rounds after the tile's first two stages
    g.salu(f"s_cmp_lg_u32 {s(RS_FIRSTF)}, 0", regs("s", RS_FIRSTF), {"scc"})
    g.raw("s_cbranch_scc1 L_r4_first_%=")
    # not the workgroup's first tile: in the in-order queue stand the requests of this tile's stages 0 and 1 (issued by the previous
    # statement's last two stages) and, younger, the epilogue's stores: stage 0 has landed when all but stage 1 and the stores have
    for st in range(2):
        for which in "ba":
            for _ in range(8):
                g.vm.append((f"dma{which}{st}", set()))
    for _ in range(EPI_STORES):
        g.vm.append(("epi", set()))
    g.wait_vm_tag("dmaa0")
    g.raw("s_branch L_r4_join_%=")
    g.raw("L_r4_first_%=:")
    # the workgroup's first tile: request stages 0 and 1 here and wait for both (once per launch); the counted waits of the first
    # stages below, sized for the queue described above, then wait for nothing
    g2 = Gen()
    for st in range(2):
        for which in "ba":
            pieces, adv = ring_requests(g2, st, btrans, which)
            for f in pieces:
                f()
            adv()
    g.out += g2.out
    g.raw("s_waitcnt vmcnt(0)")
    g.raw("L_r4_join_%=:")
    g.m0_w = None
    # this tile's bias (128 floats of this wave's columns) -> parameter block + 512, for the epilogue; no bias (pointer 0): EXEC = 0, the
    # two instructions fetch nothing but still count in vmcnt, which the counted waits below rely on.  (Starting the accumulators from
    # the bias -- C of the tile's first MFMAs -- would save the epilogue 128 v_pk_add per tile, but the sum then rounds differently from
    # the 128 x 128 kernels': the bit-identity between the GEMM paths is worth more than ~3 % of the forward GEMMs)
    g.salu(f"s_add_u32 m0, {s(RS_PARAM)}, 512", regs("s", RS_PARAM), {"m0"})
    g.salu(f"s_cmp_eq_u64 {s(RS_BIASP, 2)}, 0", regs("s", RS_BIASP, 2), {"scc"})
    g.salu("s_cselect_b64 exec, 0, -1", {"scc"}, {"exec"})
    g.vmem_dma(f"global_load_lds_dword {v(RV_BIASO)}, {s(RS_BIASP, 2)}", "bias", regs("v", RV_BIASO) | regs("s", RS_BIASP, 2) | {"m0"})
    g.vmem_dma(f"global_load_lds_dword {v(RV_BIASO)}, {s(RS_BIASP, 2)} offset:256", "bias", regs("v", RV_BIASO) | regs("s", RS_BIASP, 2) | {"m0"})
    g.salu("s_mov_b64 exec, -1", (), {"exec"})
    g.raw("s_barrier")
    for f, _ in ring_reads(g, 0, 0, btrans):
        f()
    for slot in range(2):
        ring_body(g, slot, slot == 0, btrans)
    head = list(g.out)
    texts = []
    for rnd in range(3):
        g.out = ["L_r4_loop_%=:"]
        for slot in range(2):
            ring_body(g, slot, False, btrans)
        g.salu(f"s_sub_u32 {s(RS_CNT)}, {s(RS_CNT)}, 1", regs("s", RS_CNT), regs("s", RS_CNT))
        g.salu(f"s_cmp_lg_u32 {s(RS_CNT)}, 0", regs("s", RS_CNT), {"scc"})
        g.raw("s_cbranch_scc1 L_r4_loop_%=")
        texts.append(list(g.out))
    assert texts[1] == texts[2], "the ring loop is not a fixed point of the wait-count / hazard trackers"
    g.out = []
    g.comment("==== exit: the ring's state back to the parameter block (its requests stay in flight) ====")
    for k, src in enumerate([RS_AP, RS_AP + 1, RS_BP, RS_BP + 1]):
        g.valu(f"v_mov_b32_e32 {v(T + k)}, {s(src)}", regs("s", src), regs("v", T + k))
    g.valu(f"v_mov_b32_e32 {v(T + 4)}, {s(RS_RQ)}", regs("s", RS_RQ), regs("v", T + 4))
    g.valu(f"v_mov_b32_e32 {v(T + 5)}, {s(RS_RQB)}", regs("s", RS_RQB), regs("v", T + 5))
    g.raw(f"ds_write_b128 {v(RV_PB)}, {v(T, 4)}")
    g.raw(f"ds_write_b32 {v(RV_PB)}, {v(T + 4)} offset:44")
    g.raw(f"ds_write_b32 {v(RV_PB)}, {v(T + 5)} offset:76")
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.nop(16)
    return head + texts[2] + g.out, g


def ring_clobbers():
    return [f"v{i}" for i in range(RV_FIRST, RV_LAST + 1)] + [f"s{i}" for i in range(RS_FIRST, RS_LAST + 1)] + ["vcc", "scc", "m0", "memory"]


def write_ring(here):
    path = os.path.join(here, "linear_ring4_loop_diag.inc" if DIAG else "linear_ring4_loop.inc")
    import io
    f = io.StringIO()
    if True:
        f.write("// GENERATED by gen_gemm_asm.py -- do not edit.  One tile of linear_ring4_kernel<BTRANS, PRE> (one asm statement each): operands\n")
        f.write("// %0..%15 = acc[rt][ct] (\"=a\": all 256 AGPRs, written from C = 0), %16 = LDS address of the wave's parameter block (\"s\").\n")
        for btrans, name in ((False, "NT"), (True, "NN")):
            lines, g = ring_tile(btrans)
            f.write(f"#define MGX_RING4_{name}_ASM \\\n")
            for ln in lines:
                if ln.startswith(";"):
                    f.write(f"    /* {ln[1:].strip()} */ \\\n")
                else:
                    f.write(f'    "{ln}\\n\\t" \\\n')
            f.write('    ""\n')
            n_ins = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
            print(f"wrote {path} ({name}): {n_ins} instructions, s_nop wait states inserted: {g.nops}", file=sys.stderr)
        f.write("#define MGX_RING4_CLOBBERS " + ", ".join(f'"{c}"' for c in ring_clobbers()) + "\n")
        f.write(f"#define MGX_RING4_EPI_STORES {EPI_STORES}\n")
    write_if_changed(path, f.getvalue())


def clobbers():
    c = [f"v{i}" for i in range(V_FIRST, V_LAST + 1)] + [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "m0", "memory"]
    return c


def write(here):
    lines, g = generate()
    path = os.path.join(here, "linear_dw_ring4_loop_diag.inc" if DIAG else "linear_dw_ring4_loop.inc")
    import io
    f = io.StringIO()
    if True:
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
    write_if_changed(path, f.getvalue())
    n_ins = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
    print(f"wrote {path}: {n_ins} instructions, s_nop wait states inserted: {g.nops}; counts {g.stats}", file=sys.stderr)


if __name__ == "__main__":
    write(os.path.dirname(os.path.abspath(__file__)))
    write_ring(os.path.dirname(os.path.abspath(__file__)))
