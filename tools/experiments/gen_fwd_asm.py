#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 sweep of the 64-rows-per-wave forward attention kernel (rel_attn_fwd64.hip).
EXPERIMENT (round 5): correct -- ctx and lse bit-identical to the product's 32-row HIP kernel, padding and lazy-softmax redo
included (check_fwd64.py) -- and NOT faster: 0.603 ms against 0.559 at cfg2 / batch 64 on one box (main body 1,870 stamped cycles
per pair of tiles = the HIP kernel's 890 per tile; 11 % of a wave's time in the latency-exposed prologue + pipeline fill, which
three workgroups per CU hide for the HIP kernel and one wave per SIMD cannot).  Built only by `_build.py --experiments`
(MGX_ATTN_FWD64=3 selects it); kept with its stamp tool as the record of that measurement.

    python tools/experiments/gen_fwd_asm.py            -> tools/experiments/rel_attn_fwd64_loop.inc

Math, LDS images and the rotated fp32 band are those of rel_attn_fwd.hip (the 32-rows-per-wave HIP kernel, which stays the kernel
for L % 128 != 0, the no-mask inference call and the weights output); results are bit-identical to it.  A wave owns TWO query
tiles A = I, B = I + 1 (64 rows) of a 128-row block and sweeps the key tiles 0 .. its diagonal; per key tile n it runs, for each
of its query tiles X:
      QE_X = Q_X . Er_chunk^T          4 MFMA  -> band X (16 ds_write_b32) -> Srel^T (4 ds_read_b128) = initial accumulator of
      S_X^T = K_n . Q_X^T + Srel^T     4 MFMA  (K row fragments shared by A and B)
      P_X^T = exp2(S^T log2e - m_X)    lazy softmax reference; a tile whose partial row sums leave the safe range is redone
      O_X^T += V_n^T . P_X^T           4 MFMA  (V^T fragments shared by A and B)
The E chunk B needs at step n is the chunk A used at step n - 1: two AGPR slots, one chunk load per step.

One iteration n of the generated loop is software-pipelined over three key tiles -- 24 MFMAs:
      1-8   QE_A, QE_B of tile n+2     9-16  S_A, S_B of tile n+1     17-24  O_A, O_B += of tile n
with, in their shadows: barrier + DMA of tile n+2 (three LDS buffers), K fragments of tile n+1, V^T fragments of tile n, the
exponentials of tile n+1 (first part) and of tile n (rest, row sums, the redo check, bf16 packs), band stores / reads of tile
n+2, the E chunk of tile n+3.  The body exists in 6 variants (score-register set and E slot: parity of n; K/V buffer: n mod 3),
as branch-free "main" bodies (tile n+1 strictly below both diagonals, no padded key) and "masked" bodies.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "musicgeneration_amd", "csrc"))
from asm_sched import COST, Gen, Item, a, chain, crow, regs, s, salu_items, schedule, v, write_if_changed  # noqa: E402

# ---------------------------------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------------------------------
V_L16 = 16
V_KOFF = 17          # ..18  DMA source offsets of this wave's two pieces of the K image (R)
V_VOFF = 19          # ..20  ... of the V image (T)
V_AK = 21            # ..24  K row-fragment addresses (ks = 0..3), buffer 0
V_ATR = 25           # ..26  V^T fragment base addresses (ct = 0, 1), buffer 0
V_RB = 27            # band read base (tile A; tile B = + BAND_BYTES)
V_X32 = 28           # (lane ^ 32) * 4: the other half of a query row (redo)
V_NEGINF = 29
V_PADA = 30          # LDS address of the pad-word table
V_WCL = (32, 48)     # band store addresses, chunk parity 0 / 1 (16 each; tile B = + BAND_BYTES)
V_QE = 64            # ..95  chunk products qeA | qeB
V_S = (96, 128)      # two sets: sA | sB (scores of tile t in set t & 1)
V_P = (160, 192)     # two sets: pA | pB (exponentials; the bf16 operand fragments are packed in place)
V_MNEG, V_M, V_L, V_LSUM = 224, 226, 228, 230      # (+ X): -m log2e | m | l | partial row sum of the tile
V_TMP = 232          # ..233, 242..245 temporaries (prologue, redo, pad)
V_SUMT = 234         # ..241 the four partial chains of a tile's row sum (A | B)
V_FIRST, V_LAST = 16, 245

A_O = 128            # O^T accumulators: oA0 | oA1 | oB0 | oB1
A_E = 192            # E chunk slots 0 | 1 (16 each)
A_KF = 224           # K row fragments (4 x 4)
A_VF = 240           # V^T fragments (ss, ct) 4 regs each
A_FIRST, A_LAST = 128, 255

S_MASK = 36          # 16 pairs: lanes (queries) for which the key of accumulator register r is not in the future: crow(r, hh) <= a
S_EFA, S_KVB = 68, 70
S_N, S_NTW, S_IA, S_KSTEP, S_LDS, S_NCH, S_W, S_LOG2E, S_LSAFE, S_PADANY, S_RET, S_PW = 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83
S_DK = 84            # lds0 + w * 1024: this wave's pieces of the K / V images
S_T = 86             # ..93 temporaries (pointer pairs at +2, +4)
S_EP = 94            # ..95 E chunk pointer
S_FULL = 18          # ..21 (masked bodies) per query tile: all ones if the tile is strictly below the diagonal
S_KILL = 22          # ..25 (masked bodies) per query tile: all ones if the query tile is already done (key tile beyond its diagonal)
S_TM = 26            # ..27
S_FIRST, S_LAST = 18, 95
S_STAMP = 14

OFF_K, OFF_V, OFF_BAND = 0, 12288, 24576
BAND_STRIDE, BAND_BYTES = 272, 8704
OFF_PAD = OFF_BAND + 4 * BAND_BYTES          # 59,392: pad words of the batch row's key tiles (first 256)
STAMP = False
PEEL = 0


class Step:
    """instruction groups of iteration n in variant b = n mod 6; X = 0 (tile A) / 1 (tile B)"""

    def __init__(self, g: Gen, b: int):
        self.g = g
        self.b = b
        n = b
        self.t0, self.t1, self.t2 = n, n + 1, n + 2

    def sset(self, t, X):
        return V_S[t & 1] + 16 * X

    def pset(self, t, X):
        return V_P[t & 1] + 16 * X

    def eslot(self, t, X):
        return A_E + 16 * ((t + X) & 1)

    # ---- MFMAs -----------------------------------------------------------------------------------------------------
    def mfma_QE(self, X):
        d, e = V_QE + 16 * X, self.eslot(self.t2, X)
        return [lambda ks=ks: self.g.mfma(("v", d), ("%", f"%{8 + 4 * X + ks}"), ("a", e + 4 * ks), c=0 if ks == 0 else None) for ks in range(4)]

    def mfma_S(self, X):
        d = self.sset(self.t1, X)
        return [lambda ks=ks: self.g.mfma(("v", d), ("a", A_KF + 4 * ks), ("%", f"%{8 + 4 * X + ks}")) for ks in range(4)]

    def mfma_PV(self, X):
        p = self.pset(self.t0, X)
        out = []
        for ss in range(2):
            for ct in range(2):
                out.append(lambda ss=ss, ct=ct: self.g.mfma(("a", A_O + 32 * X + 16 * ct), ("a", A_VF + 4 * (2 * ss + ct)), ("v", p + 8 * ss)))
        return out

    # ---- LDS -------------------------------------------------------------------------------------------------------
    def rd_K(self):
        buf = self.t1 % 3
        return [lambda ks=ks: self.g.ds_read(f"ds_read_b128 {a(A_KF + 4 * ks, 4)}, {v(V_AK + ks)} offset:{OFF_K + 4096 * buf}",
                                             regs("v", V_AK + ks), regs("a", A_KF + 4 * ks, 4)) for ks in range(4)]

    def rd_V(self):
        buf = self.t0 % 3
        out = []
        for ss in range(2):
            for ct in range(2):
                for jq in range(2):
                    dst = A_VF + 4 * (2 * ss + ct) + 2 * jq
                    off = OFF_V + 4096 * buf + (16 * ss + 8 * jq) * 128
                    out.append(lambda dst=dst, off=off, ct=ct: self.g.ds_read(f"ds_read_b64_tr_b16 {a(dst, 2)}, {v(V_ATR + ct)} offset:{off}",
                                                                             regs("v", V_ATR + ct), regs("a", dst, 2)))
        return out

    def band_put(self, X):
        """chunk product of tile t2 -> band X at the parity of its chunk index (dq - 1 = I_X - t2 - 1: parity (X + t2 + 1) & 1, I_A even)"""
        par = (X + self.t2 + 1) & 1
        q = V_QE + 16 * X
        return [lambda r=r: self.g.emit("lds", f"ds_write_b32 {v(V_WCL[par] + r)}, {v(q + r)} offset:{r * BAND_STRIDE + X * BAND_BYTES}",
                                        regs("v", V_WCL[par] + r) | regs("v", q + r), set()) for r in range(16)]

    def band_get(self, X):
        """Srel^T of tile t2 (D / 32 = dq = I_X - t2: parity (X + t2) & 1) -> the score registers of tile t2"""
        par = (X + self.t2) & 1
        d = self.sset(self.t2, X)
        return [lambda g4=g4: self.g.ds_read(f"ds_read_b128 {v(d + 4 * g4, 4)}, {v(V_RB)} offset:{X * BAND_BYTES + 128 * par + 32 * g4}",
                                             regs("v", V_RB), regs("v", d + 4 * g4, 4)) for g4 in range(4)]

    # ---- exponentials ------------------------------------------------------------------------------------------------
    def exp_items(self, t, X, lo, hi):
        """(fma, exp) for elements lo..hi-1 of tile t"""
        sr, p = self.sset(t, X), self.pset(t, X)
        out = []
        for r in range(lo, hi):
            out.append(lambda r=r: self.g.valu(f"v_fma_f32 {v(p + r)}, {v(sr + r)}, {s(S_LOG2E)}, {v(V_MNEG + X)}", regs("v", sr + r) | regs("v", V_MNEG + X), regs("v", p + r)))
            out.append(lambda r=r: self.g.valu(f"v_exp_f32_e32 {v(p + r)}, {v(p + r)}", regs("v", p + r), regs("v", p + r), trans=True))
        return out

    def sum_items(self, t, X):
        """partial row sum of the tile, in the order of rel_attn_fwd.hip (exp_tile): four interleaved chains c_j = p[j] + p[j+4] + p[j+8] +
        p[j+12] (independent: a single chain of 15 dependent adds stalls a lone wave), then (c0 + c1) + (c2 + c3).  Returns (items, deps):
        deps[k] = the elements item k reads"""
        p, ls, c = self.pset(t, X), V_LSUM + X, V_SUMT + 4 * X
        out, need = [], []
        for j in range(4):
            out.append(lambda j=j: self.g.valu(f"v_add_f32_e32 {v(c + j)}, {v(p + j)}, {v(p + j + 4)}", regs("v", p + j) | regs("v", p + j + 4), regs("v", c + j)))
            need.append((j, j + 4))
        for k in (8, 12):
            for j in range(4):
                out.append(lambda j=j, k=k: self.g.valu(f"v_add_f32_e32 {v(c + j)}, {v(c + j)}, {v(p + j + k)}", regs("v", c + j) | regs("v", p + j + k), regs("v", c + j)))
                need.append((j + k,))
        out.append(lambda: self.g.valu(f"v_add_f32_e32 {v(c)}, {v(c)}, {v(c + 1)}", regs("v", c, 2), regs("v", c)))
        out.append(lambda: self.g.valu(f"v_add_f32_e32 {v(c + 2)}, {v(c + 2)}, {v(c + 3)}", regs("v", c + 2, 2), regs("v", c + 2)))
        out.append(lambda: self.g.valu(f"v_add_f32_e32 {v(ls)}, {v(c)}, {v(c + 2)}", regs("v", c) | regs("v", c + 2), regs("v", ls)))
        need += [(), (), ()]
        return out, need

    def check_item(self, t, X, site):
        """!(lsum <= L_SAFE) in any lane -> redo the tile against its true maximum (out of line), then l += lsum"""
        g = self.g
        ls = V_LSUM + X

        def f():
            g.valu(f"v_cmp_nle_f32_e64 vcc, {v(ls)}, {s(S_LSAFE)}", regs("v", ls), {"vcc"})
            g.salu(f"s_mov_b32 {s(S_RET)}, {site}", (), regs("s", S_RET))
            g.nop(4)
            g.raw(f"s_cbranch_vccnz L_fwd_redo{X}{t & 1}_%=")
            g.out.append(f"L_fwd_ret{site}_%=:")
            g.valu(f"v_add_f32_e32 {v(V_L + X)}, {v(V_L + X)}, {v(ls)}", regs("v", V_L + X) | regs("v", ls), regs("v", V_L + X))
        return [f]

    def pack_items(self, t, X):
        p = self.pset(t, X)
        out = []
        for ss in range(2):
            for j in range(4):
                lo, hi, dst = p + 8 * ss + 2 * j, p + 8 * ss + 2 * j + 1, p + 8 * ss + j
                out.append(lambda lo=lo, hi=hi, dst=dst: self.g.valu(f"v_cvt_pk_bf16_f32 {v(dst)}, {v(lo)}, {v(hi)}", regs("v", lo) | regs("v", hi), regs("v", dst)))
        return out

    # ---- DMA / E / scalar bookkeeping ------------------------------------------------------------------------------------
    def dma_items(self, tag):
        """key tile min(t2, ntw - 1) -> buffer t2 % 3 (this wave's two pieces of each image)"""
        g = self.g
        buf = self.t2 % 3
        ops = [(f"s_add_u32 {s(S_T)}, {s(S_N)}, 2", regs("s", S_N), regs("s", S_T)),
               (f"s_sub_u32 {s(S_T + 1)}, {s(S_NTW)}, 1", regs("s", S_NTW), regs("s", S_T + 1)),
               (f"s_min_i32 {s(S_T)}, {s(S_T)}, {s(S_T + 1)}", regs("s", S_T, 2), regs("s", S_T)),
               (f"s_max_i32 {s(S_T)}, {s(S_T)}, 0", regs("s", S_T), regs("s", S_T)),
               (f"s_mul_i32 {s(S_T + 1)}, {s(S_T)}, {s(S_KSTEP)}", regs("s", S_T) | regs("s", S_KSTEP), regs("s", S_T + 1)),
               (f"s_add_u32 {s(S_T + 2)}, {s(S_KVB)}, {s(S_T + 1)}", regs("s", S_KVB) | regs("s", S_T + 1), regs("s", S_T + 2)),
               (f"s_addc_u32 {s(S_T + 3)}, {s(S_KVB + 1)}, 0", regs("s", S_KVB + 1), regs("s", S_T + 3))]
        out = salu_items(g, ops)
        for i in range(2):
            for img, voff in ((OFF_K, V_KOFF + i), (OFF_V, V_VOFF + i)):
                def piece(i=i, img=img, voff=voff):
                    g.salu(f"s_add_u32 m0, {s(S_DK)}, {img + 4096 * buf + 2048 * i}", regs("s", S_DK), {"m0"})
                    g.vmem_dma(f"global_load_lds_dwordx4 {v(voff)}, {s(S_T + 2, 2)}", tag, regs("v", voff) | regs("s", S_T + 2, 2) | {"m0"})
                out.append(piece)
        return out

    def ldE_items(self, tag):
        """chunk clamp(I_A - t2 - 2, 0, nchunk - 1) -> slot (t2 + 1) & 1 (B's slot at t2 = A's slot at t2 + 1)"""
        g = self.g
        slot = A_E + 16 * ((self.t2 + 1) & 1)
        ops = [(f"s_sub_u32 {s(S_T + 4)}, {s(S_IA)}, {s(S_N)}", regs("s", S_IA) | regs("s", S_N), regs("s", S_T + 4)),
               (f"s_sub_u32 {s(S_T + 4)}, {s(S_T + 4)}, 4", regs("s", S_T + 4), regs("s", S_T + 4)),          # I_A - (n + 2) - 2
               (f"s_max_i32 {s(S_T + 4)}, {s(S_T + 4)}, 0", regs("s", S_T + 4), regs("s", S_T + 4)),
               (f"s_lshl_b32 {s(S_T + 4)}, {s(S_T + 4)}, 12", regs("s", S_T + 4), regs("s", S_T + 4)),
               (f"s_add_u32 {s(S_EP)}, {s(S_EFA)}, {s(S_T + 4)}", regs("s", S_EFA) | regs("s", S_T + 4), regs("s", S_EP)),
               (f"s_addc_u32 {s(S_EP + 1)}, {s(S_EFA + 1)}, 0", regs("s", S_EFA + 1), regs("s", S_EP + 1))]
        out = salu_items(g, ops)
        for ks in range(4):
            out.append(lambda ks=ks: g.vmem_load(f"global_load_dwordx4 {a(slot + 4 * ks, 4)}, {v(V_L16)}, {s(S_EP, 2)} offset:{1024 * ks}", tag,
                                                 regs("v", V_L16) | regs("s", S_EP, 2), regs("a", slot + 4 * ks, 4)))
        return out

    def barrier_item(self, dma_tag):
        g = self.g

        def f():
            g.wait_vm_tag(dma_tag)                         # this wave's pieces of tile n+1 have landed
            g.raw("s_barrier")
        return [f]

    # ---- masked bodies ---------------------------------------------------------------------------------------------------
    def mask_setup(self):
        """tile t1 = n + 1, query tile X: dq = I_A + X - n - 1;  FULL_X = dq > 0 ? ~0 : 0, KILL_X = dq < 0 ? ~0 : 0; pad word of key tile t1"""
        g = self.g

        def f():
            for X in range(2):
                g.salu(f"s_sub_u32 {s(S_TM)}, {s(S_IA)}, {s(S_N)}", regs("s", S_IA) | regs("s", S_N), regs("s", S_TM))
                if X == 0:
                    g.salu(f"s_sub_u32 {s(S_TM)}, {s(S_TM)}, 1", regs("s", S_TM), regs("s", S_TM))
                g.salu(f"s_cmp_gt_i32 {s(S_TM)}, 0", regs("s", S_TM), {"scc"})
                g.salu(f"s_cselect_b64 {s(S_FULL + 2 * X, 2)}, -1, 0", {"scc"}, regs("s", S_FULL + 2 * X, 2))
                g.salu(f"s_cmp_lt_i32 {s(S_TM)}, 0", regs("s", S_TM), {"scc"})
                g.salu(f"s_cselect_b64 {s(S_KILL + 2 * X, 2)}, -1, 0", {"scc"}, regs("s", S_KILL + 2 * X, 2))
        return [f]

    def padword_items(self):
        """pad word of key tile min(t1, 255) from the LDS table -> S_PW (0 when the batch row has no padded key)"""
        g = self.g
        T = 242

        def rd():
            g.salu(f"s_add_u32 {s(S_TM)}, {s(S_N)}, 1", regs("s", S_N), regs("s", S_TM))
            g.salu(f"s_min_u32 {s(S_TM)}, {s(S_TM)}, 255", regs("s", S_TM), regs("s", S_TM))
            g.salu(f"s_lshl_b32 {s(S_TM)}, {s(S_TM)}, 2", regs("s", S_TM), regs("s", S_TM))
            g.valu(f"v_add_u32_e32 {v(T)}, {s(S_TM)}, {v(V_PADA)}", regs("s", S_TM) | regs("v", V_PADA), regs("v", T))
            g.ds_read(f"ds_read_b32 {v(T + 1)}, {v(T)}", regs("v", T), regs("v", T + 1))

        def fin():
            g.valu(f"v_readfirstlane_b32 {s(S_PW)}, {v(T + 1)}", regs("v", T + 1), regs("s", S_PW))
            g.nop(1)
            g.salu(f"s_cmp_lg_u32 {s(S_PADANY)}, 0", regs("s", S_PADANY), {"scc"})
            g.salu(f"s_cselect_b32 {s(S_PW)}, {s(S_PW)}, 0", regs("s", S_PW) | {"scc"}, regs("s", S_PW))
        return [rd, fin]

    def mask_apply(self, X):
        """s_X[r] = keep ? s_X[r] : -inf, keep = (key of register r <= query | FULL_X) & ~KILL_X"""
        g = self.g
        c = self.sset(self.t1, X)
        out = []
        for r in range(16):
            def f(r=r):
                g.salu(f"s_or_b64 {s(S_TM, 2)}, {s(S_MASK + 2 * r, 2)}, {s(S_FULL + 2 * X, 2)}", regs("s", S_MASK + 2 * r, 2) | regs("s", S_FULL + 2 * X, 2), regs("s", S_TM, 2))
                g.salu(f"s_andn2_b64 {s(S_TM, 2)}, {s(S_TM, 2)}, {s(S_KILL + 2 * X, 2)}", regs("s", S_TM, 2) | regs("s", S_KILL + 2 * X, 2), regs("s", S_TM, 2))
                g.valu(f"v_cndmask_b32_e64 {v(c + r)}, {v(V_NEGINF)}, {v(c + r)}, {s(S_TM, 2)}", regs("v", V_NEGINF) | regs("v", c + r) | regs("s", S_TM, 2), regs("v", c + r))
            out.append(f)
        return out

    def pad_apply(self, site):
        """padded keys of tile t1 (pad word != 0): s[r] = min(s[r], -1e9) in the lane half whose key is padded (the reference's additive
        mask; future keys stay -inf).  Key of register r in half hh: crow(r, 0) + 4 hh.  ONE item with its own skip branch."""
        g = self.g
        T = 244

        def f():
            g.salu(f"s_cmp_eq_u32 {s(S_PW)}, 0", regs("s", S_PW), {"scc"})
            g.raw(f"s_cbranch_scc1 L_fwd_nopad{site}_%=")
            for r in range(16):
                k = crow(r)
                g.salu(f"s_bitcmp1_b32 {s(S_PW)}, {k}", regs("s", S_PW), {"scc"})
                g.salu(f"s_cselect_b32 {s(S_TM)}, -1, 0", {"scc"}, regs("s", S_TM))
                g.salu(f"s_bitcmp1_b32 {s(S_PW)}, {k + 4}", regs("s", S_PW), {"scc"})
                g.salu(f"s_cselect_b32 {s(S_TM + 1)}, -1, 0", {"scc"}, regs("s", S_TM + 1))
                for X in range(2):
                    c = self.sset(self.t1, X)
                    g.valu(f"v_min_f32_e32 {v(T)}, 0xce6e6b28, {v(c + r)}", regs("v", c + r), regs("v", T))
                    g.valu(f"v_cndmask_b32_e64 {v(c + r)}, {v(c + r)}, {v(T)}, {s(S_TM, 2)}", regs("v", c + r) | regs("v", T) | regs("s", S_TM, 2), regs("v", c + r))
            g.out.append(f"L_fwd_nopad{site}_%=:")
        return [f]


def body(g: Gen, b: int, n_min: int = 0, masked: bool = False, site_base: int = 0):
    """iteration variant b; n_min: tiles with index < n_min do not exist yet (pipeline fill: n = -3, -2, -1 -> n_min = -n)"""
    st = Step(g, b)
    has0, has1, has2 = n_min <= 0, n_min <= 1, n_min <= 3      # tile n / n+1 exist; QE runs from tile -1 on (n + 2 >= -1)
    dma_ok = n_min <= 2                                          # n + 2 >= 0
    prev_b = (b + 5) % 6
    g.comment(f"==== {'masked' if masked else 'main'} body {b}: tiles n={b} (O), n+1 (S), n+2 (QE)  [fill level {n_min}] ====")
    none4 = [None] * 4
    mf = (st.mfma_QE(0) + st.mfma_QE(1)) if has2 else none4 * 2
    mf += (st.mfma_S(0) + st.mfma_S(1)) if has1 else none4 * 2
    mf += (st.mfma_PV(0) + st.mfma_PV(1)) if has0 else none4 * 2
    items = []

    def add(fns, cost, **kw):
        out = [Item(f, cost, name=kw.get("name", ""), **{k: v_ for k, v_ in kw.items() if k != "name"}) for f in fns]
        items.extend(out)
        return out

    bar = add(st.barrier_item(f"dma{prev_b}"), COST["sync"], pin=1, name="barrier")
    if dma_ok:
        dm = chain(add(st.dma_items(f"dma{b}"), COST["salu"] + COST["vmem"], earliest=1, deadline=8, deps=bar, name="dma"))
    chk = [None, None]
    if has0:
        # tile n: the rest of its exponentials, row sums, check, packs -- before its PV (MFMAs 17-20 / 21-24)
        exB = add(st.exp_items(st.t0, 1, 8, 16), 0, earliest=1, deadline=5, name="expB_head")
        for k, it in enumerate(exB):
            it.cost = COST["trans"] if k & 1 else COST["valu"]
            if k & 1:
                it.deps.append(exB[k - 1])
        for X in range(2):
            fns, need = st.sum_items(st.t0, X)
            sm = add(fns, COST["valu"], earliest=1, deadline=8 + 4 * X, name=f"sum{X}")
            for k in range(4, 15):                          # program order inside each chain / the final tree
                sm[k].deps.append(sm[k - 4] if k < 12 else sm[8 + 2 * (k - 12)] if k < 14 else sm[12])
            sm[12].deps.append(sm[9]); sm[13].deps += [sm[10], sm[11]]; sm[14].deps.append(sm[13])
            if X == 1:
                for k, els in enumerate(need):
                    sm[k].deps += [exB[2 * (r - 8) + 1] for r in els if r >= 8]
            ck = add(st.check_item(st.t0, X, site_base + 2 * b + X), 3 * COST["valu"], earliest=3, deadline=9 + 4 * X, deps=sm[-1:], name=f"check{X}")
            pk = add(st.pack_items(st.t0, X), COST["valu"], earliest=3, deadline=14 + 2 * X, deps=ck, name=f"pack{X}")
            chk[X] = ck[0]
        rv = add(st.rd_V(), COST["lds"], earliest=2, deadline=13, name="rd_V")
    if has1:
        rk = add(st.rd_K(), COST["lds128"], earliest=1, deadline=6, deps=bar, name="rd_K")
    if has2:
        for X in range(2):
            put = add(st.band_put(X), COST["lds"], earliest=7 + 4 * X, deadline=18 + 3 * X, name=f"put{X}")
            get = add(st.band_get(X), COST["lds128"], earliest=8 + 4 * X, deadline=20 + 3 * X, deps=put + ([chk[X]] if chk[X] else []), name=f"get{X}")
        le = chain(add(st.ldE_items(f"e{b}"), 0, earliest=1, deadline=8, name="ld_E"))
        for it in le[:-4]:
            it.cost = COST["salu"]
        for it in le[-4:]:
            it.cost = COST["vmem"]
            it.earliest, it.deadline = 8, 11                 # after QE_B (MFMAs 5-8) has been issued; early: the next body's first MFMA reads the slot
    if has1:
        # tile n+1: masks (masked bodies), then the first part of its exponentials
        mk = [None, None]
        if masked:
            ms = add(st.mask_setup(), 10 * COST["salu"], earliest=1, deadline=12, name="mask_setup")
            pw = chain(add(st.padword_items(), 4 * COST["salu"], earliest=1, deadline=14, name="padword"))
            for X in range(2):
                mk[X] = add(st.mask_apply(X), 2 * COST["salu"] + COST["valu"], earliest=15 + 4 * X, deadline=21 + 2 * X, deps=ms, name=f"mask{X}")
            pad = add(st.pad_apply(site_base + b), 8 * COST["valu"], earliest=19, deadline=23, deps=pw + mk[0] + mk[1], name="pad")
        exA = add(st.exp_items(st.t1, 0, 0, 16), 0, earliest=15, deadline=24, name="expA_tail")
        exB1 = add(st.exp_items(st.t1, 1, 0, 8), 0, earliest=19, deadline=24, name="expB_tail")
        for lst, X in ((exA, 0), (exB1, 1)):
            for k, it in enumerate(lst):
                it.cost = COST["trans"] if k & 1 else COST["valu"]
                if k & 1:
                    it.deps.append(lst[k - 1])
                elif masked:
                    it.deps += [mk[X][k >> 1]] + pad
        if masked:
            # (the pad item rewrites every score register of B: its elements 8..15, exponentiated in the next body, are final by then)
            pass
    table, budget = schedule(items, ngaps=24)
    g.comment(f"per-gap issue budget {budget}")
    chains = {}
    if has2:
        chains[1] = regs("a", st.eslot(st.t2, 0), 16)
        chains[5] = regs("a", st.eslot(st.t2, 1), 16)
    if has1:
        chains[9] = regs("a", A_KF, 16) | regs("v", st.sset(st.t1, 0), 16)
        chains[13] = regs("v", st.sset(st.t1, 1), 16)
    if has0:
        chains[17] = regs("a", A_VF, 16)
    for gi in range(1, 25):
        m = mf[gi - 1]
        if gi in chains:
            g.prewait(chains[gi])
        if m is not None:
            m()
        for it in table[gi]:
            it.fn()
        if STAMP and n_min == 0 and gi in (8, 16, 24):
            g.stamp({8: 0, 16: 1, 24: 2}[gi] + (3 if masked else 0))


def redo_block(X: int, par: int, nsites: int):
    """out of line: tile (score set `par`, query tile X) left the safe exponent range -> redo it against the true row maximum: m_new =
    max(m, row max), alpha = exp2((m - m_new) log2e), O_X *= alpha, l_X *= alpha, exponentials and row sum again (rel_attn_fwd.hip:
    softmax_pv).  Own tracker: everything it touches is complete at the call sites (see body), it waits for its own LDS operation."""
    g = Gen()
    sr, p = V_S[par] + 16 * X, V_P[par] + 16 * X
    T, T2, AL = V_TMP, V_TMP + 1, 245
    g.out.append(f"L_fwd_redo{X}{par}_%=:")
    g.nop(4)
    g.valu(f"v_max_f32_e32 {v(T)}, {v(sr)}, {v(sr + 1)}", regs("v", sr, 2), regs("v", T))
    for r in range(2, 16):
        g.valu(f"v_max_f32_e32 {v(T)}, {v(T)}, {v(sr + r)}", regs("v", T) | regs("v", sr + r), regs("v", T))
    g.emit("lds", f"ds_bpermute_b32 {v(T2)}, {v(V_X32)}, {v(T)}", regs("v", V_X32) | regs("v", T), regs("v", T2))
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.valu(f"v_max_f32_e32 {v(T)}, {v(T)}, {v(T2)}", regs("v", T, 2), regs("v", T))
    g.valu(f"v_max_f32_e32 {v(T)}, {v(V_M + X)}, {v(T)}", regs("v", V_M + X) | regs("v", T), regs("v", T))                     # m_new
    g.valu(f"v_sub_f32_e32 {v(AL)}, {v(V_M + X)}, {v(T)}", regs("v", V_M + X) | regs("v", T), regs("v", AL))
    g.valu(f"v_mul_f32_e64 {v(AL)}, {v(AL)}, {s(S_LOG2E)}", regs("v", AL), regs("v", AL))
    g.valu(f"v_exp_f32_e32 {v(AL)}, {v(AL)}", regs("v", AL), regs("v", AL), trans=True)                                      # alpha
    g.valu(f"v_mov_b32_e32 {v(V_M + X)}, {v(T)}", regs("v", T), regs("v", V_M + X))
    g.valu(f"v_mul_f32_e64 {v(V_MNEG + X)}, -{v(T)}, {s(S_LOG2E)}", regs("v", T), regs("v", V_MNEG + X))                       # -m_new * log2e
    for r in range(16):
        g.valu(f"v_fma_f32 {v(p + r)}, {v(sr + r)}, {s(S_LOG2E)}, {v(V_MNEG + X)}", regs("v", sr + r) | regs("v", V_MNEG + X), regs("v", p + r))
        g.valu(f"v_exp_f32_e32 {v(p + r)}, {v(p + r)}", regs("v", p + r), regs("v", p + r), trans=True)
    ls, c = V_LSUM + X, V_SUMT + 4 * X
    g.nop(2)
    for j in range(4):
        g.valu(f"v_add_f32_e32 {v(c + j)}, {v(p + j)}, {v(p + j + 4)}", regs("v", p + j) | regs("v", p + j + 4), regs("v", c + j))
    for k in (8, 12):
        for j in range(4):
            g.valu(f"v_add_f32_e32 {v(c + j)}, {v(c + j)}, {v(p + j + k)}", regs("v", c + j) | regs("v", p + j + k), regs("v", c + j))
    g.valu(f"v_add_f32_e32 {v(c)}, {v(c)}, {v(c + 1)}", regs("v", c, 2), regs("v", c))
    g.valu(f"v_add_f32_e32 {v(c + 2)}, {v(c + 2)}, {v(c + 3)}", regs("v", c + 2, 2), regs("v", c + 2))
    g.valu(f"v_add_f32_e32 {v(ls)}, {v(c)}, {v(c + 2)}", regs("v", c) | regs("v", c + 2), regs("v", ls))
    g.valu(f"v_mul_f32_e32 {v(V_L + X)}, {v(V_L + X)}, {v(AL)}", regs("v", V_L + X) | regs("v", AL), regs("v", V_L + X))
    for k in range(32):
        ar = A_O + 32 * X + k
        g.raw(f"v_accvgpr_read_b32 {v(T2)}, {a(ar)}")
        g.raw(f"v_mul_f32_e32 {v(T2)}, {v(T2)}, {v(AL)}")
        g.raw(f"v_accvgpr_write_b32 {a(ar)}, {v(T2)}")
    g.nop(4)
    # return to the call site
    for site in range(nsites):
        g.raw(f"s_cmp_eq_u32 {s(S_RET)}, {site}")
        g.raw(f"s_cbranch_scc1 L_fwd_ret{site}_%=")
    g.raw("s_trap 2")                                    # unreachable
    return g.out


LANE_TAB = [V_KOFF, V_KOFF + 1, V_VOFF, V_VOFF + 1, V_AK, V_AK + 1, V_AK + 2, V_AK + 3, V_ATR, V_ATR + 1, V_RB] + [V_WCL[0] + r for r in range(16)]


def prologue(g: Gen):
    """%16 = LDS address of this wave's parameter block (written by the HIP code just before):
         dwords 0..15: EfA, kv_base (64-bit each) | ntw, I_A, kstep, lds0, nchunk, w, anypad, 0
         + 256 + 256 k + 4 lane: lane table k = koff0..1, voff0..1, ak0..3, atr0..1, rb, wcl[0..15]"""
    g.comment("==== prologue ====")
    g.drain()
    T = V_QE
    g.valu(f"v_mbcnt_lo_u32_b32 {v(V_TMP)}, -1, 0", set(), regs("v", V_TMP))
    g.valu(f"v_mbcnt_hi_u32_b32 {v(V_TMP)}, -1, {v(V_TMP)}", regs("v", V_TMP), regs("v", V_TMP))          # lane
    g.valu(f"v_lshlrev_b32_e32 {v(V_L16)}, 4, {v(V_TMP)}", regs("v", V_TMP), regs("v", V_L16))
    g.valu(f"v_mov_b32_e32 {v(V_TMP + 1)}, %16", set(), regs("v", V_TMP + 1))
    for k in range(3):
        g.ds_read(f"ds_read_b128 {v(T + 4 * k, 4)}, {v(V_TMP + 1)} offset:{16 * k}", regs("v", V_TMP + 1), regs("v", T + 4 * k, 4))
    g.valu(f"v_lshl_add_u32 {v(V_TMP + 2)}, {v(V_TMP)}, 2, {v(V_TMP + 1)}", regs("v", V_TMP, 2), regs("v", V_TMP + 2))
    for k, dst in enumerate(LANE_TAB):
        g.ds_read(f"ds_read_b32 {v(dst)}, {v(V_TMP + 2)} offset:{256 + 256 * k}", regs("v", V_TMP + 2), regs("v", dst))
    sc = [S_EFA, S_EFA + 1, S_KVB, S_KVB + 1, S_NTW, S_IA, S_KSTEP, S_LDS, S_NCH, S_W, S_PADANY]
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.nop(1)
    for k, dst in enumerate(sc):
        g.valu(f"v_readfirstlane_b32 {s(dst)}, {v(T + k)}", regs("v", T + k), regs("s", dst))
    g.nop(4)
    g.salu(f"s_lshl_b32 {s(S_T)}, {s(S_W)}, 10", regs("s", S_W), regs("s", S_T))
    g.salu(f"s_add_u32 {s(S_DK)}, {s(S_LDS)}, {s(S_T)}", regs("s", S_LDS) | regs("s", S_T), regs("s", S_DK))
    g.salu(f"s_mov_b32 {s(S_LOG2E)}, 0x3fb8aa3b", (), regs("s", S_LOG2E))
    g.salu(f"s_mov_b32 {s(S_LSAFE)}, 0x6753c21c", (), regs("s", S_LSAFE))           # L_SAFE = 1.0e24f
    g.salu(f"s_mov_b32 {s(S_PW)}, 0", (), regs("s", S_PW))
    # E chunks of the first products (tile -1): A: chunk I_A in slot 1, B: chunk I_A + 1 in slot 0
    for slot, add in ((1, 0), (0, 1)):
        g.salu(f"s_add_u32 {s(S_T + 4)}, {s(S_IA)}, {add}", regs("s", S_IA), regs("s", S_T + 4))
        g.salu(f"s_lshl_b32 {s(S_T + 4)}, {s(S_T + 4)}, 12", regs("s", S_T + 4), regs("s", S_T + 4))
        ptr = S_T + 2 if slot else S_EP
        g.salu(f"s_add_u32 {s(ptr)}, {s(S_EFA)}, {s(S_T + 4)}", regs("s", S_EFA) | regs("s", S_T + 4), regs("s", ptr))
        g.salu(f"s_addc_u32 {s(ptr + 1)}, {s(S_EFA + 1)}, 0", regs("s", S_EFA + 1), regs("s", ptr + 1))
        for ks in range(4):
            dst = A_E + 16 * slot + 4 * ks
            g.vmem_load(f"global_load_dwordx4 {a(dst, 4)}, {v(V_L16)}, {s(ptr, 2)} offset:{1024 * ks}", "e0", regs("v", V_L16) | regs("s", ptr, 2), regs("a", dst, 4))
    # constants, state
    for r in range(16):
        g.valu(f"v_xor_b32_e32 {v(V_WCL[1] + r)}, 0x80, {v(V_WCL[0] + r)}", regs("v", V_WCL[0] + r), regs("v", V_WCL[1] + r))
    g.valu(f"v_mov_b32_e32 {v(V_NEGINF)}, 0xff800000", set(), regs("v", V_NEGINF))
    g.valu(f"v_xor_b32_e32 {v(V_X32)}, 32, {v(V_TMP)}", regs("v", V_TMP), regs("v", V_X32))
    g.valu(f"v_lshlrev_b32_e32 {v(V_X32)}, 2, {v(V_X32)}", regs("v", V_X32), regs("v", V_X32))
    g.valu(f"v_mov_b32_e32 {v(V_TMP + 3)}, {OFF_PAD}", set(), regs("v", V_TMP + 3))
    g.valu(f"v_add_u32_e32 {v(V_PADA)}, {s(S_LDS)}, {v(V_TMP + 3)}", regs("v", V_TMP + 3) | regs("s", S_LDS), regs("v", V_PADA))
    for X in range(2):
        g.valu(f"v_mov_b32_e32 {v(V_M + X)}, 0xfcf0bdc2", set(), regs("v", V_M + X))               # M_INIT = -1.0e37f
        g.valu(f"v_mov_b32_e32 {v(V_L + X)}, 0", set(), regs("v", V_L + X))
        g.valu(f"v_mul_f32_e64 {v(V_MNEG + X)}, -{v(V_M + X)}, {s(S_LOG2E)}", regs("v", V_M + X), regs("v", V_MNEG + X))
    for k in range(64):
        g.raw(f"v_accvgpr_write_b32 {a(A_O + k)}, 0")
    # masks: key crow(r, hh) <= query a  <=>  crow(r, 0) <= a - 4 hh
    AM, X_ = V_TMP + 4, V_TMP + 5
    g.valu(f"v_and_b32_e32 {v(AM)}, 31, {v(V_TMP)}", regs("v", V_TMP), regs("v", AM))
    g.valu(f"v_lshrrev_b32_e32 {v(X_)}, 5, {v(V_TMP)}", regs("v", V_TMP), regs("v", X_))
    g.valu(f"v_lshlrev_b32_e32 {v(X_)}, 2, {v(X_)}", regs("v", X_), regs("v", X_))
    g.valu(f"v_sub_u32_e32 {v(AM)}, {v(AM)}, {v(X_)}", regs("v", AM) | regs("v", X_), regs("v", AM))          # am (may be negative)
    for r in range(16):
        g.valu(f"v_cmp_ge_i32_e64 {s(S_MASK + 2 * r, 2)}, {v(AM)}, {crow(r)}", regs("v", AM), regs("s", S_MASK + 2 * r, 2))
    g.salu(f"s_mov_b32 {s(S_N)}, -3", (), regs("s", S_N))
    g.drain()
    g.raw("s_barrier")


def loop_tail(g: Gen, b: int, masked: bool):
    nb = (b + 1) % 6
    g.salu(f"s_add_u32 {s(S_N)}, {s(S_N)}, 1", regs("s", S_N), regs("s", S_N))
    g.salu(f"s_cmp_ge_i32 {s(S_N)}, {s(S_NTW)}", regs("s", S_N) | regs("s", S_NTW), {"scc"})
    g.raw("s_cbranch_scc1 L_fwd_end_%=")
    if masked:
        if b == 5:
            g.raw("s_branch L_fwd_m0_%=")
        return
    # a wave stays in the main bodies while tile n + 1 is strictly below the diagonal of its first query tile: n + 1 < I_A
    g.salu(f"s_add_u32 {s(S_TM)}, {s(S_N)}, 1", regs("s", S_N), regs("s", S_TM))
    g.salu(f"s_cmp_lt_i32 {s(S_TM)}, {s(S_IA)}", regs("s", S_TM) | regs("s", S_IA), {"scc"})
    g.raw(f"s_cbranch_scc1 L_fwd_u{nb}_%=")
    g.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")              # the masked bodies' counted waits assume their own history
    g.raw("s_nop 7")
    g.raw("s_nop 7")
    g.raw(f"s_branch L_fwd_m{nb}_%=")


def fixed_point_loop(g: Gen, masked: bool):
    texts = []
    for rnd in range(3):
        g.out = []
        for b in range(6):
            g.out.append(f"L_fwd_{'m' if masked else 'u'}{b}_%=:")
            body(g, b, masked=masked, site_base=12 if masked else 0)
            if STAMP:
                g.raw(f"v_add_u32_e32 {v(246 + (6 if masked else 7))}, 1, {v(246 + (6 if masked else 7))}")      # iterations: [6] masked, [7] main
            loop_tail(g, b, masked)
        texts.append(list(g.out))
    assert texts[1] == texts[2], "the loop body is not a fixed point of the wait-count / hazard trackers"
    return texts[2]


def generate():
    g = Gen()
    g.s_stamp = S_STAMP
    prologue(g)
    # pipeline fill: n = -3 (QE of tile -1: the diagonal chunks), -2 (QE of tile 0, DMA of tile 0), -1 (QE 1, S 0, DMA 1)
    for n in (-3, -2, -1):
        body(g, n % 6, n_min=-n, masked=True, site_base=24 + 12 * (n + 3))      # masked: tile 0 may be a diagonal tile (first query block)
        g.salu(f"s_add_u32 {s(S_N)}, {s(S_N)}, 1", regs("s", S_N), regs("s", S_N))
    g.drain()
    g.nop(16)
    if STAMP:
        for k in range(8):
            g.raw(f"v_mov_b32_e32 {v(246 + k)}, 0")
        g.stamp(None)
    # n = 0: main bodies if tile 1 is strictly below the wave's first diagonal and no key of the batch row is padded
    g.salu(f"s_cmp_lg_u32 {s(S_PADANY)}, 0", regs("s", S_PADANY), {"scc"})
    g.raw("s_cbranch_scc1 L_fwd_m0_%=")
    g.salu(f"s_cmp_gt_i32 {s(S_IA)}, 1", regs("s", S_IA), {"scc"})
    g.raw("s_cbranch_scc1 L_fwd_u0_%=")
    g.raw("s_branch L_fwd_m0_%=")
    pro_lines = list(g.out)
    u_lines = fixed_point_loop(g, masked=False)
    g.drain()
    m_lines = fixed_point_loop(g, masked=True)
    g.out = ["L_fwd_end_%=:"]
    g.drain()
    g.nop(16)
    # results -> the output operands: O through an MFMA with zero A / B operands (D = 0 * 0 + C copies a 16-register tile), m and l by moves
    Z = V_QE
    for k in range(4):
        g.raw(f"v_mov_b32_e32 {v(Z + k)}, 0")
    g.nop(4)
    for k in range(4):
        g.raw(f"v_mfma_f32_32x32x16_bf16 %{k}, {v(Z, 4)}, {v(Z, 4)}, {a(A_O + 16 * k, 16)}")
    g.raw(f"v_mov_b32_e32 %4, {v(V_M)}")
    g.raw(f"v_mov_b32_e32 %5, {v(V_L)}")
    g.raw(f"v_mov_b32_e32 %6, {v(V_M + 1)}")
    g.raw(f"v_mov_b32_e32 %7, {v(V_L + 1)}")
    if STAMP:                                    # sums -> LDS: OFF_PAD + 512 + 64 w + 4 k (the pad table of a short sequence ends before that)
        g.raw(f"s_lshl_b32 {s(S_TM)}, {s(S_W)}, 6")
        g.raw(f"s_add_u32 {s(S_TM)}, {s(S_TM)}, {OFF_PAD + 512}")
        g.raw(f"s_add_u32 {s(S_TM)}, {s(S_TM)}, {s(S_LDS)}")
        g.raw(f"v_mov_b32_e32 {v(V_TMP)}, {s(S_TM)}")
        for k in range(8):
            g.raw(f"ds_write_b32 {v(V_TMP)}, {v(246 + k)} offset:{4 * k}")
        g.raw("s_waitcnt lgkmcnt(0)")
    g.nop(16)
    g.raw("s_barrier")
    g.raw("s_branch L_fwd_done_%=")
    tail = list(g.out)
    redo = []
    for X in range(2):
        for par in range(2):
            redo += redo_block(X, par, 24)
    return pro_lines + u_lines + m_lines + tail + redo + ["L_fwd_done_%=:"], g


def clobbers():
    c = [f"v{i}" for i in range(V_FIRST, V_LAST + 1)] + [f"a{i}" for i in range(A_FIRST, A_LAST + 1)]
    c += [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "m0", "memory"]
    if STAMP:
        c += [f"v{i}" for i in range(246, 254)] + [f"s{i}" for i in range(S_STAMP, S_STAMP + 4)]
    return c


def write(here):
    lines, g = generate()
    path = os.path.join(here, "rel_attn_fwd64_loop_stamp.inc" if STAMP else "rel_attn_fwd64_loop.inc")
    import io
    f = io.StringIO()
    if True:
        f.write("// GENERATED by gen_fwd_asm.py -- do not edit.  The hand-scheduled sweep of rel_attn_fwd64_kernel (one asm statement):\n")
        f.write("// operands %0..%3 = oA0, oA1, oB0, oB1 (\"=&a\"), %4..%7 = m_A, l_A, m_B, l_B (\"=&v\"), %8..%15 = the scaled q fragments of tiles A, B\n")
        f.write("// (\"a\"), %16 = LDS address of the wave's parameter block (\"s\").  Register map, schedule and hazard rules: gen_fwd_asm.py.\n")
        f.write("#define MGX_FWD64_LOOP_ASM \\\n")
        for ln in lines:
            if ln.startswith(";"):
                f.write(f"    /* {ln[1:].strip()} */ \\\n")
            else:
                f.write(f'    "{ln}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define MGX_FWD64_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
    write_if_changed(path, f.getvalue())
    n_ins = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
    print(f"wrote {path}: {n_ins} instructions, s_nop wait states inserted: {g.nops}; counts {g.stats}", file=sys.stderr)


if __name__ == "__main__":
    for STAMP in (False, True):
        write(os.path.dirname(os.path.abspath(__file__)))
