#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 main loop of the 64-keys-per-wave dK/dV kernel (rel_attn_dkv64.hip).

    python musicgeneration_amd/csrc/gen_dkv_asm.py            -> musicgeneration_amd/csrc/rel_attn_dkv64_loop.inc

The kernel's address set-up and its epilogue (the dk / dv stores) stay HIP; the whole sweep over the query tiles is ONE
`asm volatile` block emitted by this script: a fixed register map, every MFMA followed by the VALU / LDS / VMEM instructions
assigned to its shadow, counted waits, and the hazards hipcc does not pad inside an asm statement checked (and padded) here.
Two loops of six body variants each: "masked" bodies for the steps of the diagonal block (a sub-tile not started / on its
diagonal) and for key blocks with padded keys, and the branch-free main bodies a wave switches to once both its sub-tiles are
full (query tile n >= wk + 2).  What a step computes, and the layouts, are those of rel_attn_bwd.hip
(dkv_kernel) / tools/experiments/rel_attn_bwd64.hip; results are bit-identical to both.

Structure of one iteration n (query tile n against the wave's key tiles J, J+1 = sub-tiles 0, 1), 44 MFMAs:
    A  1-16   S0, S1 (c_u += q k_u^T on top of the skewed Q.Er^T term), dP0, dP1 (dp_u = -delta + dO v_u^T)      [tile n]
    B  17-28  Q.Er^T chunk products t2, t1, t0 of tile n+1                                                         [tile n+1]
    C  29-36  dV0 += dO^T P0, dK0 += q^T dS0                                                                       [tile n]
    D  37-44  dV1, dK1                                                                                             [tile n]
In the shadows: the transposed q / dO fragments and the statistics of tile n (LDS), barrier, DMA of tile n+2, fragments
of tile n+1, exponentials / dS / bf16 packs of tile n, the E chunk of tile n+2 (global), merge + lane-permutation skew of
tile n+1, the four dS stores of tile n.  One barrier per iteration; two LDS buffers; E chunks rotate through three AGPR
slots, so the body exists in 6 variants (buffer parity x E rotation).
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_sched import COST, Gen, Item, a, chain, crow, regs, s, salu_items, schedule, v, write_if_changed  # noqa: E402

# ---------------------------------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------------------------------
V_L16 = 16
V_AQ = 17            # ..20  q row-fragment addresses (ks = 0..3), LDS buffer 0 of image QR
V_ATR = 21           # transposed-fragment base address (image QR, buffer 0)
V_AST = 22           # statistics address (lds0 + OFF_ST + this wave's 256-byte block + 16 hh), buffer 0
V_QOFF = 23          # ..24  DMA source offsets of q (two 1 KB pieces)
V_OOFF = 25          # ..26  ... of dO
V_STOFF = 27         # ... of the statistics
V_KOFF = 28          # K / V fragment offset (prologue only)
V_NEGINF = 29        # -inf in every lane (masked bodies)
V_TMP = 238          # ..245 temporaries (prologue)
V_RD = 30            # ..45  ds_bpermute source-lane addresses of the skew
V_QA = 46            # ..61  q row fragments (4 x 4)
V_OF = 62            # ..77  dO row fragments
V_T = (78, 126)      # two sets of 48: t0 | t1 | t2 chunk products; c0 = t0, c1 = t2 after the skew, nl in t1
V_DP = 174           # ..205 dp0 | dp1
V_TR = 206           # ..237 transposed fragments: dO (ss, ct) 4 regs each, then q
V_FIRST, V_LAST = 16, 245

A_KF = 144           # kf[u][ks] at A_KF + 4 (4u + ks)
A_VF = 176
A_E = 208            # E slot s, ks at A_E + 4 (4s + ks)
A_FIRST, A_LAST = 144, 255

S_MASK = 36          # 16 pairs: lanes with key <= query for accumulator register r
S_EFA, S_QB, S_OB, S_STB, S_DSB, S_KVB = 68, 70, 72, 74, 76, 78      # 64-bit pointers (even-aligned pairs)
S_N, S_NT, S_WK, S_I, S_DSOFF, S_QSTEP, S_OSTEP, S_LDS, S_NCH, S_KEXP, S_W = 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 98
S_T = 90             # ..97 temporaries (even: pointer pairs at +2, +4, +6)
S_DQ, S_DST = 99, 100  # LDS destinations of this wave's DMA pieces (images / statistics), buffer 0
S_DV = 101           # d * 2 (byte offset of V behind K)
S_FIRST, S_LAST = 36, 101
S_ET, S_EP, S_SP = 32, 34, 30   # E-load temporaries (2), E chunk pointer (2), dS store pointer (2): s30..s35
S_ST = 29
S_FIRST = 14
S_STAMP = 10         # ..13 (stamp builds): s_memtime pair, previous stamp, difference
S_PADM = 14          # ..17: lanes whose key is padded, sub-tile 0 / 1
S_FULL = 18          # ..21 (masked bodies): all ones where sub-tile u is below its diagonal (no causal mask), else 0
S_KILL = 22          # ..25 (masked bodies): lanes to mask whatever the row: padded keys; everything while sub-tile u has not started
S_TM = 26            # ..27 (masked bodies): keep-mask of one accumulator register
S_PADANY = 28        # != 0: a key of this wave is padded (every step runs the masked body)
V_STAMP = 246        # ..253 (stamp builds): cycle sums
STAMP = False
ST_CACHE = os.environ.get("MGX_DKV64_ST", "nt")          # experiment: cache policy of the dS stores (nt | plain | sc1 | sc0sc1)
ST_PINS = [int(x) for x in os.environ.get("MGX_DKV64_STPIN", "").split(",") if x]      # experiment: the MFMA shadows of the four dS stores
# experiment (round 6): the two waves of a workgroup run the same instruction stream in lock step (one barrier per iteration), so their
# dS stores -- 1 KB each through the CU's one store path -- always collide; "a,b,c,d:e,f,g,h" gives wave 0 and wave 1 their OWN shadows
# for the four stores (every store is emitted twice, under EXEC = (w == 0) / (w == 1); an EXEC = 0 store moves no data)
STAMP_WAIT = os.environ.get("MGX_DKV64_STAMP_WAIT", "") == "1"      # stamp builds: time the DMA wait and the barrier at the top of a main body on their own
STAMP_PEEL = int(os.environ.get("MGX_DKV64_STAMP_PEEL", "0"))         # stamp builds: PEEL bits applied to the stamped loop (e.g. 1: no dS stores)
STAGGER = [[int(x) for x in part.split(",")] for part in os.environ.get("MGX_DKV64_STAGGER", "").split(":") if part]
S_W0M, S_W1M = 18, 20   # (main bodies, STAGGER) lane masks of wave 0 / wave 1: all ones or zero -- the registers of S_FULL, which only the masked bodies use
PEEL = 0             # timing experiments (results wrong): 1 no dS stores | 2 no E loads | 4 no DMA | 8 no skew | 16 no exp | 32 no merge | 64 no stats reads | 128 no tr reads | 256 only wave 0 stores dS | 512 control of 256 | 1024 no barrier in the loop bodies

OFF_QR, OFF_OR, OFF_ST = 0, 12288, 24576      # three LDS buffers per image: query tile t in buffer t % 3
ST_BYTES = 512




# ---------------------------------------------------------------------------------------------------------------------
# pieces of a step
# ---------------------------------------------------------------------------------------------------------------------
ST_SUFFIX = {"nt": " nt", "plain": "", "sc1": " sc1", "sc0sc1": " sc0 sc1"}[ST_CACHE]


class Step:
    """instruction groups of iteration n in variant `b` = n mod 6"""

    def __init__(self, g: Gen, b: int):
        self.g = g
        n = b
        self.cur, self.nxt, self.dst = n % 3, (n + 1) % 3, (n + 2) % 3      # LDS buffers of tile n, of tile n+1, and of the DMA of tile n+2
        self.T = V_T[n & 1]                    # c0 / nl / c1 of tile n
        self.Tn = V_T[(n + 1) & 1]             # chunk products of tile n+1
        self.slot_t0 = (n + 1) % 3             # E slots of the products of tile n+1
        self.slot_t1 = n % 3
        self.slot_t2 = (n + 2) % 3             # ... and the slot the chunk of tile n+2 is loaded into, once t2 has been issued
        self.tag = f"it{b}"

    # -- A: S and dP of tile n ------------------------------------------------------------------------------------
    def mfma_S(self, u):
        c = self.T + (0 if u == 0 else 32)
        return [lambda ks=ks: self.g.mfma(("v", c), ("v", V_QA + 4 * ks), ("a", A_KF + 4 * (4 * u + ks))) for ks in range(4)]

    def mfma_dP(self, u):
        d = V_DP + 16 * u
        return [lambda ks=ks: self.g.mfma(("v", d), ("v", V_OF + 4 * ks), ("a", A_VF + 4 * (4 * u + ks))) for ks in range(4)]

    # -- B: chunk products of tile n+1 ----------------------------------------------------------------------------
    def mfma_QE(self, k):
        """k = 0: t0 (hi chunk of sub-tile 0), 1: t1, 2: t2"""
        d = self.Tn + 16 * k
        slot = (self.slot_t0, self.slot_t1, self.slot_t2)[k]
        return [lambda ks=ks: self.g.mfma(("v", d), ("v", V_QA + 4 * ks), ("a", A_E + 4 * (4 * slot + ks)), c=0 if ks == 0 else None)
                for ks in range(4)]

    # -- C / D: dV, dK of tile n -------------------------------------------------------------------------------------
    def mfma_dVdK(self, u):
        c = self.T + (0 if u == 0 else 32)       # pf_u[ss] = c_u[8ss .. 8ss+3]
        d = V_DP + 16 * u                       # df_u[ss]
        out = []
        for ss in range(2):
            for ct in range(2):
                out.append(lambda ss=ss, ct=ct: self.g.mfma_op(f"%{4 * u + 2 + ct}", ("v", V_TR + 4 * (2 * ss + ct)), ("v", c + 8 * ss)))      # dv[u][ct]
            for ct in range(2):
                out.append(lambda ss=ss, ct=ct: self.g.mfma_op(f"%{4 * u + ct}", ("v", V_TR + 16 + 4 * (2 * ss + ct)), ("v", d + 8 * ss)))   # dk[u][ct]
        return out

    # -- LDS reads of tile n (before the barrier) ---------------------------------------------------------------------
    def rd_tr(self):
        """transposed fragments of dO (V_TR + 0..15) and q (V_TR + 16..31) of tile n: X[kappa][32 ct + lane&31]"""
        out = []
        for img, base in ((OFF_OR, V_TR), (OFF_QR, V_TR + 16)):
            for ss in range(2):
                for ct in range(2):
                    for jq in range(2):
                        dst = base + 4 * (2 * ss + ct) + 2 * jq
                        off = img + self.cur * 4096 + (16 * ss + 8 * jq) * 128 + ((ct ^ jq) << 6)
                        out.append(lambda dst=dst, off=off: self.g.ds_read(f"ds_read_b64_tr_b16 {v(dst, 2)}, {v(V_ATR)} offset:{off}",
                                                                           regs("v", V_ATR), regs("v", dst, 2)))
        return out

    def rd_nd(self, u, nxt=False):
        """-delta of the accumulator rows: initial value of dp_u (nxt: of tile n+1, from the other buffer)"""
        out = []
        buf = self.nxt if nxt else self.cur
        for g4 in range(4):
            dst = V_DP + 16 * u + 4 * g4
            off = buf * ST_BYTES + 128 + 32 * g4         # (V_AST includes OFF_ST)
            out.append(lambda dst=dst, off=off: self.g.ds_read(f"ds_read_b128 {v(dst, 4)}, {v(V_AST)} offset:{off}", regs("v", V_AST), regs("v", dst, 4)))
        return out

    def rd_nl(self):
        """-lse log2(e) of the accumulator rows -> the t1 registers of this tile's set (free since the merge)"""
        out = []
        for g4 in range(4):
            dst = self.T + 16 + 4 * g4
            off = self.cur * ST_BYTES + 32 * g4
            out.append(lambda dst=dst, off=off: self.g.ds_read(f"ds_read_b128 {v(dst, 4)}, {v(V_AST)} offset:{off}", regs("v", V_AST), regs("v", dst, 4)))
        return out

    # -- barrier, DMA of tile n+2, row fragments of tile n+1 ---------------------------------------------------------------
    def barrier(self, dma_tag):
        g = self.g

        def f():
            if STAMP and STAMP_WAIT:                       # stamp builds with MGX_DKV64_STAMP_WAIT=1: sums 5 / 6 = the counted wait / the barrier alone
                g.stamp(None)
            g.wait_vm_tag(dma_tag)                         # this wave's pieces of tile n+1 have landed
            if STAMP and STAMP_WAIT:
                g.stamp(5)
            if not (PEEL & 1024):                          # PEEL 1024 (round 6, timing only): what the per-iteration barrier of the two waves costs
                g.raw("s_barrier")
            if STAMP and STAMP_WAIT:
                g.stamp(6)
        return [f]

    def dma_addr(self):
        """source pointers of tile min(n + 2, nT - 1): q -> S_T+2, dO -> S_T+4, statistics -> S_T+6 (one SALU instruction per item)"""
        g = self.g
        ops = [(f"s_add_u32 {s(S_T)}, {s(S_N)}, 2", regs("s", S_N), regs("s", S_T)),
               (f"s_sub_u32 {s(S_T + 1)}, {s(S_NT)}, 1", regs("s", S_NT), regs("s", S_T + 1)),
               (f"s_min_u32 {s(S_T)}, {s(S_T)}, {s(S_T + 1)}", regs("s", S_T, 2), regs("s", S_T)),
               (f"s_mul_i32 {s(S_T + 1)}, {s(S_T)}, {s(S_QSTEP)}", regs("s", S_T) | regs("s", S_QSTEP), regs("s", S_T + 1)),
               (f"s_add_u32 {s(S_T + 2)}, {s(S_QB)}, {s(S_T + 1)}", regs("s", S_QB) | regs("s", S_T + 1), regs("s", S_T + 2)),
               (f"s_addc_u32 {s(S_T + 3)}, {s(S_QB + 1)}, 0", regs("s", S_QB + 1), regs("s", S_T + 3)),
               (f"s_mul_i32 {s(S_T + 1)}, {s(S_T)}, {s(S_OSTEP)}", regs("s", S_T) | regs("s", S_OSTEP), regs("s", S_T + 1)),
               (f"s_add_u32 {s(S_T + 4)}, {s(S_OB)}, {s(S_T + 1)}", regs("s", S_OB) | regs("s", S_T + 1), regs("s", S_T + 4)),
               (f"s_addc_u32 {s(S_T + 5)}, {s(S_OB + 1)}, 0", regs("s", S_OB + 1), regs("s", S_T + 5)),
               (f"s_lshl_b32 {s(S_T + 1)}, {s(S_T)}, 7", regs("s", S_T), regs("s", S_T + 1)),
               (f"s_add_u32 {s(S_T + 6)}, {s(S_STB)}, {s(S_T + 1)}", regs("s", S_STB) | regs("s", S_T + 1), regs("s", S_T + 6)),
               (f"s_addc_u32 {s(S_T + 7)}, {s(S_STB + 1)}, 0", regs("s", S_STB + 1), regs("s", S_T + 7))]
        return salu_items(g, ops)

    def dma(self, tag):
        """tile min(n + 2, nT - 1) -> buffer (n + 2) % 3 (pointers from dma_addr)"""
        g = self.g
        out = []
        for i in range(2):
            for img, ptr, voff in ((OFF_QR, S_T + 2, V_QOFF + i), (OFF_OR, S_T + 4, V_OOFF + i)):
                def piece(i=i, img=img, ptr=ptr, voff=voff):
                    g.salu(f"s_add_u32 m0, {s(S_DQ)}, {img + self.dst * 4096 + 2048 * i}", regs("s", S_DQ), {"m0"})
                    g.vmem_dma(f"global_load_lds_dwordx4 {v(voff)}, {s(ptr, 2)}", tag, regs("v", voff) | regs("s", ptr, 2) | {"m0"})
                out.append(piece)

        def stat():
            g.salu(f"s_add_u32 m0, {s(S_DST)}, {self.dst * ST_BYTES}", regs("s", S_DST), {"m0"})
            g.vmem_dma(f"global_load_lds_dword {v(V_STOFF)}, {s(S_T + 6, 2)}", tag, regs("v", V_STOFF) | regs("s", S_T + 6, 2) | {"m0"})
        out.append(stat)
        return out

    def rd_rows(self, which):
        """row fragments of tile n+1: q -> V_QA, dO -> V_OF"""
        img, base = (OFF_QR, V_QA) if which == "q" else (OFF_OR, V_OF)
        out = []
        for ks in range(4):
            off = img + self.nxt * 4096
            out.append(lambda ks=ks, off=off: self.g.ds_read(f"ds_read_b128 {v(base + 4 * ks, 4)}, {v(V_AQ + ks)} offset:{off}",
                                                              regs("v", V_AQ + ks), regs("v", base + 4 * ks, 4)))
        return out

    # -- exponentials, dS, packs of tile n ---------------------------------------------------------------------------------
    def soft_exp(self, u):
        c, nl = self.T + (0 if u == 0 else 32), self.T + 16
        out = []
        for r in range(16):
            out.append(lambda r=r: self.g.valu(f"v_fma_f32 {v(c + r)}, {v(c + r)}, {s(S_KEXP)}, {v(nl + r)}", regs("v", c + r) | regs("v", nl + r), regs("v", c + r)))
            out.append(lambda r=r: self.g.valu(f"v_exp_f32_e32 {v(c + r)}, {v(c + r)}", regs("v", c + r), regs("v", c + r), trans=True))
        return out

    def soft_ds(self, u):
        c, d = self.T + (0 if u == 0 else 32), V_DP + 16 * u
        out = []
        for r in range(16):
            out.append(lambda r=r: self.g.valu(f"v_mul_f32_e32 {v(d + r)}, {v(c + r)}, {v(d + r)}", regs("v", c + r) | regs("v", d + r), regs("v", d + r)))
        # packs: registers 8ss + 2j, 8ss + 2j + 1 -> 8ss + j (ascending j: a destination has been consumed by then)
        for base in (d, c):                       # dS first: its store and its product come first
            for ss in range(2):
                for j in range(4):
                    lo, hi, dst = base + 8 * ss + 2 * j, base + 8 * ss + 2 * j + 1, base + 8 * ss + j
                    out.append(lambda lo=lo, hi=hi, dst=dst: self.g.valu(f"v_cvt_pk_bf16_f32 {v(dst)}, {v(lo)}, {v(hi)}",
                                                                         regs("v", lo) | regs("v", hi), regs("v", dst)))
        return out

    # -- masked bodies: causal / padding / not-started masks of tile n ---------------------------------------------------------
    def mask_setup(self):
        """per sub-tile u (dq_u = n - wk - u): FULL_u = dq_u > 0 ? ~0 : 0;  KILL_u = PADM_u | (dq_u < 0 ? ~0 : 0)"""
        g = self.g
        ops = []
        for u in range(2):
            ops += [(f"s_sub_u32 {s(S_TM)}, {s(S_N)}, {s(S_WK)}", regs("s", S_N) | regs("s", S_WK), regs("s", S_TM))]
            if u:
                ops += [(f"s_sub_u32 {s(S_TM)}, {s(S_TM)}, 1", regs("s", S_TM), regs("s", S_TM))]
            ops += [(f"s_cmp_gt_i32 {s(S_TM)}, 0", regs("s", S_TM), {"scc"}),
                    (f"s_cselect_b64 {s(S_FULL + 2 * u, 2)}, -1, 0", {"scc"}, regs("s", S_FULL + 2 * u, 2)),
                    (f"s_cmp_lt_i32 {s(S_TM)}, 0", regs("s", S_TM), {"scc"}),
                    (f"s_cselect_b64 {s(S_KILL + 2 * u, 2)}, -1, 0", {"scc"}, regs("s", S_KILL + 2 * u, 2)),
                    (f"s_or_b64 {s(S_KILL + 2 * u, 2)}, {s(S_KILL + 2 * u, 2)}, {s(S_PADM + 2 * u, 2)}", regs("s", S_KILL + 2 * u, 2) | regs("s", S_PADM + 2 * u, 2),
                     regs("s", S_KILL + 2 * u, 2))]
        return [lambda: [g.salu(*o) for o in ops]]          # ONE item: the compare / select pairs pass through SCC, S_TM is a temporary

    def mask_apply(self, u):
        """c_u[r] = keep ? c_u[r] : -inf, keep = (lanes with key <= query of register r | FULL_u) & ~KILL_u"""
        g = self.g
        c = self.T + (0 if u == 0 else 32)
        out = []
        for r in range(16):
            def f(r=r):
                g.salu(f"s_or_b64 {s(S_TM, 2)}, {s(S_MASK + 2 * r, 2)}, {s(S_FULL + 2 * u, 2)}", regs("s", S_MASK + 2 * r, 2) | regs("s", S_FULL + 2 * u, 2), regs("s", S_TM, 2))
                g.salu(f"s_andn2_b64 {s(S_TM, 2)}, {s(S_TM, 2)}, {s(S_KILL + 2 * u, 2)}", regs("s", S_TM, 2) | regs("s", S_KILL + 2 * u, 2), regs("s", S_TM, 2))
                g.valu(f"v_cndmask_b32_e64 {v(c + r)}, {v(V_NEGINF)}, {v(c + r)}, {s(S_TM, 2)}", regs("v", V_NEGINF) | regs("v", c + r) | regs("s", S_TM, 2), regs("v", c + r))
            out.append(f)
        return out

    def st_dS_masked(self, tag):
        """the dS stores of sub-tile u exist only for key tile <= query tile (dq_u >= 0): EXEC = 0 otherwise -- a store with no active
        lane writes nothing and still counts in vmcnt, so the counted waits hold.  One item per sub-tile: nothing else may run under the
        cleared EXEC"""
        g = self.g
        out = []
        for u in range(2):
            def f(u=u):
                g.salu(f"s_sub_u32 {s(S_TM)}, {s(S_N)}, {s(S_WK)}", regs("s", S_N) | regs("s", S_WK), regs("s", S_TM))
                g.salu(f"s_cmp_ge_i32 {s(S_TM)}, {u}", regs("s", S_TM), {"scc"})
                g.salu(f"s_cselect_b64 exec, -1, 0", {"scc"}, {"exec"})
                for ss in range(2):
                    src = V_DP + 16 * u + 8 * ss
                    g.vmem_store(f"global_store_dwordx4 {v(V_L16)}, {v(src, 4)}, {s(S_SP, 2)} offset:{2048 * u + 1024 * ss}{ST_SUFFIX}", tag,
                                 regs("v", V_L16) | regs("v", src, 4) | regs("s", S_SP, 2))
                g.salu("s_mov_b64 exec, -1", (), {"exec"})
            out.append(f)
        return out

    # -- E chunk of tile n+2 -------------------------------------------------------------------------------------------------
    def ldE_addr(self):
        """pointer of chunk clamp(n + 2 - wk, 0, nchunk - 1) of the fragment-ordered E copy -> S_EP"""
        g = self.g
        ops = [(f"s_add_u32 {s(S_ET)}, {s(S_N)}, 2", regs("s", S_N), regs("s", S_ET)),
               (f"s_sub_u32 {s(S_ET)}, {s(S_ET)}, {s(S_WK)}", regs("s", S_ET) | regs("s", S_WK), regs("s", S_ET)),
               (f"s_sub_u32 {s(S_ET + 1)}, {s(S_NCH)}, 1", regs("s", S_NCH), regs("s", S_ET + 1)),
               (f"s_min_i32 {s(S_ET)}, {s(S_ET)}, {s(S_ET + 1)}", regs("s", S_ET, 2), regs("s", S_ET)),
               (f"s_max_i32 {s(S_ET)}, {s(S_ET)}, 0", regs("s", S_ET), regs("s", S_ET)),        # (wave 1's first steps: chunk < 0, product unused)
               (f"s_lshl_b32 {s(S_ET)}, {s(S_ET)}, 12", regs("s", S_ET), regs("s", S_ET)),
               (f"s_add_u32 {s(S_EP)}, {s(S_EFA)}, {s(S_ET)}", regs("s", S_EFA) | regs("s", S_ET), regs("s", S_EP)),
               (f"s_addc_u32 {s(S_EP + 1)}, {s(S_EFA + 1)}, 0", regs("s", S_EFA + 1), regs("s", S_EP + 1))]
        return salu_items(g, ops)

    def ld_E(self, tag):
        g = self.g
        out = []
        for ks in range(4):
            dst = A_E + 4 * (4 * self.slot_t2 + ks)
            out.append(lambda ks=ks, dst=dst: g.vmem_load(f"global_load_dwordx4 {a(dst, 4)}, {v(V_L16)}, {s(S_EP, 2)} offset:{1024 * ks}", tag,
                                                          regs("v", V_L16) | regs("s", S_EP, 2), regs("a", dst, 4)))
        return out

    # -- merge + skew of tile n+1 ------------------------------------------------------------------------------------------
    def merge(self, u):
        """sub-tile 0: lanes with key <= query take t0, the others t1 -> t0;  sub-tile 1: t1 / t2 -> t2"""
        t0, t1, t2 = self.Tn, self.Tn + 16, self.Tn + 32
        hi, lo, dst = (t0, t1, t0) if u == 0 else (t1, t2, t2)
        return [lambda r=r: self.g.valu(f"v_cndmask_b32_e64 {v(dst + r)}, {v(lo + r)}, {v(hi + r)}, {s(S_MASK + 2 * r, 2)}",
                                        regs("v", lo + r) | regs("v", hi + r) | regs("s", S_MASK + 2 * r, 2), regs("v", dst + r)) for r in range(16)]

    def skew(self, u):
        c = self.Tn + (0 if u == 0 else 32)
        return [lambda r=r: self.g.emit("lds", f"ds_bpermute_b32 {v(c + r)}, {v(V_RD + r)}, {v(c + r)}", regs("v", V_RD + r) | regs("v", c + r), regs("v", c + r))
                for r in range(16)]

    # -- dS stores of tile n -----------------------------------------------------------------------------------------------
    def st_addr(self):
        """pointer of this step's dS tiles -> S_SP; then the offset of the next tile row: I += 1, offset += I << 11"""
        g = self.g
        ops = [(f"s_add_u32 {s(S_SP)}, {s(S_DSB)}, {s(S_DSOFF)}", regs("s", S_DSB) | regs("s", S_DSOFF), regs("s", S_SP)),
               (f"s_addc_u32 {s(S_SP + 1)}, {s(S_DSB + 1)}, 0", regs("s", S_DSB + 1), regs("s", S_SP + 1)),
               (f"s_add_u32 {s(S_I)}, {s(S_I)}, 1", regs("s", S_I), regs("s", S_I)),
               (f"s_lshl_b32 {s(S_ST)}, {s(S_I)}, 11", regs("s", S_I), regs("s", S_ST)),
               (f"s_add_u32 {s(S_DSOFF)}, {s(S_DSOFF)}, {s(S_ST)}", regs("s", S_DSOFF) | regs("s", S_ST), regs("s", S_DSOFF))]
        return salu_items(g, ops)

    def st_dS(self, tag):
        g = self.g
        out = []
        for u in range(2):
            for ss in range(2):
                src = V_DP + 16 * u + 8 * ss

                def f(u=u, ss=ss, src=src, wave=None):
                    # PEEL 256 / 512 (round 6, timing only): is a store's ~90 cycles the two waves of a workgroup -- and the other
                    # workgroup of the CU -- pushing 1 KB each through the CU's store path AT THE SAME POINT of the body?  256: only
                    # wave 0's stores move data (wave 1: EXEC = 0, still counted in vmcnt); 512: the same three SALU instructions around
                    # every store with all waves storing (what the wrapper itself costs)
                    if PEEL & 768:
                        g.salu(f"s_cmp_eq_u32 {s(S_W)}, 0", regs("s", S_W), {"scc"})
                        g.salu(f"s_cselect_b64 exec, -1, {'0' if PEEL & 256 else '-1'}", {"scc"}, {"exec"})
                    if wave is not None:
                        g.salu(f"s_mov_b64 exec, {s((S_W0M, S_W1M)[wave], 2)}", regs("s", (S_W0M, S_W1M)[wave], 2), {"exec"})
                    g.vmem_store(f"global_store_dwordx4 {v(V_L16)}, {v(src, 4)}, {s(S_SP, 2)} offset:{2048 * u + 1024 * ss}{ST_SUFFIX}", tag,
                                 regs("v", V_L16) | regs("v", src, 4) | regs("s", S_SP, 2))
                    if PEEL & 768 or wave is not None:
                        g.salu("s_mov_b64 exec, -1", (), {"exec"})
                if STAGGER:
                    out.append(lambda f=f: f(wave=0))
                    out.append(lambda f=f: f(wave=1))
                else:
                    out.append(f)
        return out


STAMP_GAPS = {10: 0, 16: 1, 28: 2, 36: 3, 44: 4}      # stamp after the shadow of MFMA g -> sum index


def body(g: Gen, b: int, do_cur: bool = True, masked: bool = False, listing=None):
    st = Step(g, b)
    prev_b = (b + 5) % 6
    g.comment(f"==== {'masked' if masked else 'main'} body {b}: buffer {st.cur}, E slots t0/t1/t2 = {st.slot_t0}/{st.slot_t1}/{st.slot_t2}{'' if do_cur else '  (pipeline fill: tile n+1 part only)'} ====")
    # ---- the MFMA sequence ----
    none4 = [None] * 4
    mf = (st.mfma_S(0) + st.mfma_S(1) + st.mfma_dP(0) + st.mfma_dP(1)) if do_cur else none4 * 4
    mf += st.mfma_QE(0) + st.mfma_QE(1) + st.mfma_QE(2)
    mf += (st.mfma_dVdK(0) + st.mfma_dVdK(1)) if do_cur else none4 * 4
    # ---- the fillers ----
    I = Item
    items = []

    def add(fns, cost, **kw):
        out = [I(f, cost, name=kw.get("name", ""), **{k: v_ for k, v_ in kw.items() if k != "name"}) for f in fns]
        items.extend(out)
        return out

    # Three LDS buffers: the barrier at the top of iteration n tells every wave that (a) all pieces of tile n+1 have landed (each wave
    # waited for its own, requested a whole iteration ago) and (b) everybody has left iteration n-1, so the buffer of tile n-1 is free
    # for the DMA of tile n+2.  No LDS read of the iteration is tied to a position by the barrier: the scheduler spreads them.
    bar = add(st.barrier(f"dma{prev_b}"), COST["sync"], pin=1, name="barrier")
    dma_a = chain(add(st.dma_addr(), COST["salu"], earliest=1, deadline=4, name="dma_addr"))
    dma = chain(add(st.dma(f"dma{b}"), COST["salu"] + COST["vmem"], earliest=1, deadline=12, deps=bar + dma_a[-1:], name="dma"))
    if do_cur:
        nd0 = add(st.rd_nd(0), COST["lds128"], earliest=1, deadline=7, name="nd0")                     # dP0 = MFMA 9
        nd1 = add(st.rd_nd(1), COST["lds128"], earliest=1, deadline=11, name="nd1")                    # dP1 = MFMA 13
        nl = add(st.rd_nl(), COST["lds128"], earliest=1, deadline=6, name="nl")
        tr = add(st.rd_tr(), COST["lds"], earliest=1, deadline=26, name="tr")                          # dV / dK = MFMAs 29..
    rq = add(st.rd_rows("q"), COST["lds128"], earliest=9, deadline=14, deps=bar, name="rows_q")        # S of tile n = MFMAs 1-8; QE of tile n+1 = 17..
    ro = add(st.rd_rows("o"), COST["lds128"], earliest=16, deadline=40, deps=bar, name="rows_o")       # dP of tile n = MFMAs 9-16
    if do_cur:
        cvds = []
        if masked:
            ms = add(st.mask_setup(), 12 * COST["salu"], earliest=1, deadline=6, name="mask_setup")
        for u, (e_c, e_dp, dl) in enumerate(((7, 15, 27), (11, 19, 35))):
            ex = add(st.soft_exp(u), 0, earliest=e_c, deadline=dl - 4, name=f"exp{u}")
            if masked:
                mk = add(st.mask_apply(u), 2 * COST["salu"] + COST["valu"], earliest=e_c, deadline=dl - 5, deps=ms, name=f"mask{u}")
            for r in range(16):
                ex[2 * r].cost, ex[2 * r + 1].cost = COST["valu"], COST["trans"]
                ex[2 * r + 1].deps.append(ex[2 * r])
                if masked:
                    ex[2 * r].deps.append(mk[r])
            sd = add(st.soft_ds(u), COST["valu"], earliest=max(e_c, e_dp), deadline=dl, name=f"ds{u}")
            mul, cvd, cvc = sd[:16], sd[16:24], sd[24:32]
            for r in range(16):
                mul[r].deps.append(ex[2 * r + 1])
            for k, (ss, j) in enumerate((ss, j) for ss in range(2) for j in range(4)):
                cvd[k].deps += [mul[8 * ss + 2 * j], mul[8 * ss + 2 * j + 1], mul[8 * ss + j]] + ([cvd[k - 1]] if j else [])
                cvc[k].deps += [ex[2 * (8 * ss + 2 * j) + 1], ex[2 * (8 * ss + 2 * j + 1) + 1], mul[8 * ss + j]] + ([cvc[k - 1]] if j else [])
            cvds.append(cvd)
        sa = chain(add(st.st_addr(), COST["salu"], earliest=1, deadline=30, name="st_addr"))
        if masked:
            sts = chain(add(st.st_dS_masked(f"st{b}"), 2 * COST["store"] + 4 * COST["salu"], earliest=20, deadline=44, deps=sa[-1:], name="st_dS"))
            for u, it in enumerate(sts):
                it.deps += cvds[u]
        else:
            if STAGGER:
                # items (store k, wave 0), (store k, wave 1), k = 0..3: no chain -- each at its own pin, after the packs of its data
                sts = add(st.st_dS(f"st{b}"), COST["store"] + 2 * COST["salu"], earliest=18, deadline=44, deps=sa[-1:], name="st_dS")
                for j, it in enumerate(sts):
                    k, wave = j >> 1, j & 1
                    it.deps += cvds[k >> 1][4 * (k & 1): 4 * (k & 1) + 4]
                    it.pin = STAGGER[wave][k]
            else:
                sts = chain(add(st.st_dS(f"st{b}"), COST["store"], earliest=20, deadline=44, deps=sa[-1:], name="st_dS"))
                for k, it in enumerate(sts):
                    it.deps += cvds[k >> 1][4 * (k & 1): 4 * (k & 1) + 4]
                    if ST_PINS:
                        it.pin = ST_PINS[k]
    ea = chain(add(st.ldE_addr(), COST["salu"], earliest=1, deadline=28, name="ldE_addr"))
    le = chain(add(st.ld_E(f"e{b}"), COST["vmem"], earliest=28, deadline=40, deps=ea[-1:], name="ld_E"))   # t2 = MFMAs 25-28
    m0 = add(st.merge(0), COST["valu"], earliest=27, deadline=38, name="merge0")                        # t0, t1 done after MFMA 24
    m1 = add(st.merge(1), COST["valu"], earliest=31, deadline=42, name="merge1")
    k0 = add(st.skew(0), COST["lds"], earliest=27, deadline=40, name="skew0")
    k1 = add(st.skew(1), COST["lds"], earliest=31, deadline=44, name="skew1")
    for r in range(16):
        k0[r].deps.append(m0[r])
        k1[r].deps.append(m1[r])
    if PEEL and not masked and do_cur:
        drop = set()
        for bit, names in ((1, ("st_dS",)), (2, ("ld_E",)), (4, ("dma",)), (8, ("skew0", "skew1")), (32, ("merge0", "merge1")),
                           (64, ("nd0", "nd1", "nl")), (128, ("tr",))):
            if PEEL & bit:
                drop |= set(names)
        for it in items:
            if it.name in drop:
                it.fn = lambda: None
        if PEEL & 16:
            for it in items:
                if it.name.startswith("exp") and it.cost == COST["trans"]:
                    it.fn = lambda: None
    table, budget = schedule(items)
    g.comment(f"per-gap issue budget {budget}")
    chains = {17: regs("v", V_QA, 16) | regs("a", A_E + 16 * st.slot_t0, 16)}      # MFMA index -> registers its chain of four reads
    if do_cur:
        chains[9] = regs("v", V_OF, 16) | regs("v", V_DP, 16)
        chains[29] = regs("v", V_TR, 32)
    for gi in range(1, 45):
        m = mf[gi - 1]
        if gi in chains:
            g.prewait(chains[gi])
        if m is not None:
            m()
        if listing is not None:
            listing.append((gi, len(g.out)))
        for it in table[gi]:
            it.fn()
        if STAMP and do_cur and not masked and gi in STAMP_GAPS:
            g.stamp(STAMP_GAPS[gi])


# ---------------------------------------------------------------------------------------------------------------------
# prologue / epilogue
# ---------------------------------------------------------------------------------------------------------------------
def prologue(g: Gen):
    """%8 = LDS address of this wave's parameter block (written by the HIP code just before):
         dwords 0..23: EfA, q_base, o_base, st_base, ds_col, kv_base (64-bit each) | 0, nT, wk, I0, q_step, o_step, lds0, nchunk,
                       w, d*2, pad word of key tile 0, of key tile 1
         + 256 + 256 k + 4 lane: lane table k = aq0..3, atr, ast, qoff0..1, ooff0..1, stoff, koff"""
    g.comment("==== prologue ====")
    g.drain()
    T = V_T[0]                                   # 24 temporaries for the scalar block
    g.valu(f"v_mbcnt_lo_u32_b32 {v(V_TMP)}, -1, 0", set(), regs("v", V_TMP))
    g.valu(f"v_mbcnt_hi_u32_b32 {v(V_TMP)}, -1, {v(V_TMP)}", regs("v", V_TMP), regs("v", V_TMP))          # lane
    g.valu(f"v_lshlrev_b32_e32 {v(V_L16)}, 4, {v(V_TMP)}", regs("v", V_TMP), regs("v", V_L16))
    g.valu(f"v_mov_b32_e32 {v(V_TMP + 1)}, %8", set(), regs("v", V_TMP + 1))
    for k in range(6):
        g.ds_read(f"ds_read_b128 {v(T + 4 * k, 4)}, {v(V_TMP + 1)} offset:{16 * k}", regs("v", V_TMP + 1), regs("v", T + 4 * k, 4))
    g.valu(f"v_lshl_add_u32 {v(V_TMP + 2)}, {v(V_TMP)}, 2, {v(V_TMP + 1)}", regs("v", V_TMP, 2), regs("v", V_TMP + 2))
    lane_tab = [V_AQ, V_AQ + 1, V_AQ + 2, V_AQ + 3, V_ATR, V_AST, V_QOFF, V_QOFF + 1, V_OOFF, V_OOFF + 1, V_STOFF, V_KOFF]
    for k, dst in enumerate(lane_tab):
        g.ds_read(f"ds_read_b32 {v(dst)}, {v(V_TMP + 2)} offset:{256 + 256 * k}", regs("v", V_TMP + 2), regs("v", dst))
    S_PW = S_TM                                  # the two pad words (temporaries until the masks exist)
    sc = [S_EFA, S_EFA + 1, S_QB, S_QB + 1, S_OB, S_OB + 1, S_STB, S_STB + 1, S_DSB, S_DSB + 1, S_KVB, S_KVB + 1,
          S_N, S_NT, S_WK, S_I, S_QSTEP, S_OSTEP, S_LDS, S_NCH, S_W, S_DV, S_PW, S_PW + 1]
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.nop(1)
    for k, dst in enumerate(sc):
        g.valu(f"v_readfirstlane_b32 {s(dst)}, {v(T + k)}", regs("v", T + k), regs("s", dst))
    g.nop(4)                                     # VALU write of an SGPR -> VMEM / SALU users
    # derived scalars
    g.salu(f"s_lshl_b32 {s(S_T)}, {s(S_W)}, 10", regs("s", S_W), regs("s", S_T))
    g.salu(f"s_add_u32 {s(S_DQ)}, {s(S_LDS)}, {s(S_T)}", regs("s", S_LDS) | regs("s", S_T), regs("s", S_DQ))                  # lds0 + w * 1024
    g.salu(f"s_lshl_b32 {s(S_T)}, {s(S_W)}, 8", regs("s", S_W), regs("s", S_T))
    g.salu(f"s_add_u32 {s(S_DST)}, {s(S_LDS)}, {s(S_T)}", regs("s", S_LDS) | regs("s", S_T), regs("s", S_DST))
    g.salu(f"s_add_u32 {s(S_DST)}, {s(S_DST)}, {OFF_ST}", regs("s", S_DST), regs("s", S_DST))                              # lds0 + OFF_ST + w * 256
    # every global request of the prologue goes out now (distinct pointer pairs: no wait in between): the K / V row fragments of the
    # wave's two key tiles (HBM: the longest latency), query tile 0 -> buffer 0, E chunk 0 into the slot of the first step's hi chunk
    # (its other two products use chunks < 0: sub-tiles that have not started or lanes beyond the diagonal, masked whatever they hold)
    P_V0, P_K1, P_V1 = S_T + 2, S_T + 4, S_T + 6
    g.salu(f"s_add_u32 {s(P_V0)}, {s(S_KVB)}, {s(S_DV)}", regs("s", S_KVB) | regs("s", S_DV), regs("s", P_V0))
    g.salu(f"s_addc_u32 {s(P_V0 + 1)}, {s(S_KVB + 1)}, 0", regs("s", S_KVB + 1), regs("s", P_V0 + 1))
    g.salu(f"s_add_u32 {s(P_K1)}, {s(S_KVB)}, {s(S_QSTEP)}", regs("s", S_KVB) | regs("s", S_QSTEP), regs("s", P_K1))
    g.salu(f"s_addc_u32 {s(P_K1 + 1)}, {s(S_KVB + 1)}, 0", regs("s", S_KVB + 1), regs("s", P_K1 + 1))
    g.salu(f"s_add_u32 {s(P_V1)}, {s(P_K1)}, {s(S_DV)}", regs("s", P_K1) | regs("s", S_DV), regs("s", P_V1))
    g.salu(f"s_addc_u32 {s(P_V1 + 1)}, {s(P_K1 + 1)}, 0", regs("s", P_K1 + 1), regs("s", P_V1 + 1))
    for u, (pk, pv) in enumerate(((S_KVB, P_V0), (P_K1, P_V1))):
        for ks in range(4):
            dk, dv_ = A_KF + 4 * (4 * u + ks), A_VF + 4 * (4 * u + ks)
            g.vmem_load(f"global_load_dwordx4 {a(dk, 4)}, {v(V_KOFF)}, {s(pk, 2)} offset:{32 * ks}", "kv", regs("v", V_KOFF) | regs("s", pk, 2), regs("a", dk, 4))
            g.vmem_load(f"global_load_dwordx4 {a(dv_, 4)}, {v(V_KOFF)}, {s(pv, 2)} offset:{32 * ks}", "kv", regs("v", V_KOFF) | regs("s", pv, 2), regs("a", dv_, 4))
    for i in range(2):
        for img, ptr, voff in ((OFF_QR, S_QB, V_QOFF + i), (OFF_OR, S_OB, V_OOFF + i)):
            g.salu(f"s_add_u32 m0, {s(S_DQ)}, {img + 2048 * i}", regs("s", S_DQ), {"m0"})
            g.vmem_dma(f"global_load_lds_dwordx4 {v(voff)}, {s(ptr, 2)}", "dma_first", regs("v", voff) | regs("s", ptr, 2) | {"m0"})
    g.salu(f"s_add_u32 m0, {s(S_DST)}, 0", regs("s", S_DST), {"m0"})
    g.vmem_dma(f"global_load_lds_dword {v(V_STOFF)}, {s(S_STB, 2)}", "dma_first", regs("v", V_STOFF) | regs("s", S_STB, 2) | {"m0"})
    for ks in range(4):
        dst = A_E + 4 * (4 * 0 + ks)             # slot 0 = the hi-chunk slot of the fill iteration (n = -1: (n + 1) % 3)
        g.vmem_load(f"global_load_dwordx4 {a(dst, 4)}, {v(V_L16)}, {s(S_EFA, 2)} offset:{1024 * ks}", "e0", regs("v", V_L16) | regs("s", S_EFA, 2), regs("a", dst, 4))
    # ... and the lane-dependent constants are formed while they are in flight
    g.salu(f"s_mov_b32 {s(S_KEXP)}, 0x3e38aa3b", (), regs("s", S_KEXP))         # 0.125 * log2(e) = 0x3fb8aa3b / 8, exact
    g.salu(f"s_add_u32 {s(S_T)}, {s(S_I)}, 1", regs("s", S_I), regs("s", S_T))                                              # dS offset of tile row I = I0:
    g.salu(f"s_mul_i32 {s(S_T)}, {s(S_T)}, {s(S_I)}", regs("s", S_T) | regs("s", S_I), regs("s", S_T))                      #   I (I + 1) / 2 * 2048
    g.salu(f"s_lshl_b32 {s(S_DSOFF)}, {s(S_T)}, 10", regs("s", S_T), regs("s", S_DSOFF))
    g.salu(f"s_or_b32 {s(S_PADANY)}, {s(S_PW)}, {s(S_PW + 1)}", regs("s", S_PW, 2), regs("s", S_PADANY))
    g.valu(f"v_mov_b32_e32 {v(V_NEGINF)}, 0xff800000", set(), regs("v", V_NEGINF))
    # masks: lanes whose key (lane & 31) <= query crow(r, hh) ; skew source lanes rd[r] = (32 hh + ((crow - bl) & 31)) * 4
    BL, HH4, HH128, X = V_TMP + 3, V_TMP + 4, V_TMP + 5, V_TMP + 6
    g.valu(f"v_and_b32_e32 {v(BL)}, 31, {v(V_TMP)}", regs("v", V_TMP), regs("v", BL))
    g.valu(f"v_lshrrev_b32_e32 {v(HH4)}, 5, {v(V_TMP)}", regs("v", V_TMP), regs("v", HH4))
    g.valu(f"v_lshlrev_b32_e32 {v(HH128)}, 7, {v(HH4)}", regs("v", HH4), regs("v", HH128))       # 32 hh * 4
    g.valu(f"v_lshlrev_b32_e32 {v(HH4)}, 2, {v(HH4)}", regs("v", HH4), regs("v", HH4))           # 4 hh
    for u in range(2):                           # lanes whose key is padded: bit (lane & 31) of the key tile's pad word
        g.valu(f"v_lshrrev_b32_e64 {v(X)}, {v(BL)}, {s(S_PW + u)}", regs("v", BL) | regs("s", S_PW + u), regs("v", X))
        g.valu(f"v_and_b32_e32 {v(X)}, 1, {v(X)}", regs("v", X), regs("v", X))
        g.valu(f"v_cmp_ne_u32_e64 {s(S_PADM + 2 * u, 2)}, 0, {v(X)}", regs("v", X), regs("s", S_PADM + 2 * u, 2))
    for r in range(16):
        g.valu(f"v_add_u32_e32 {v(X)}, {crow(r)}, {v(HH4)}", regs("v", HH4), regs("v", X))                             # crow(r, hh)
        g.valu(f"v_cmp_le_u32_e64 {s(S_MASK + 2 * r, 2)}, {v(BL)}, {v(X)}", regs("v", BL) | regs("v", X), regs("s", S_MASK + 2 * r, 2))
        g.valu(f"v_sub_u32_e32 {v(X)}, {v(X)}, {v(BL)}", regs("v", X) | regs("v", BL), regs("v", X))
        g.valu(f"v_and_b32_e32 {v(X)}, 31, {v(X)}", regs("v", X), regs("v", X))
        g.valu(f"v_lshl_add_u32 {v(V_RD + r)}, {v(X)}, 2, {v(HH128)}", regs("v", X) | regs("v", HH128), regs("v", V_RD + r))
    g.salu(f"s_mov_b32 {s(S_N)}, -1", (), regs("s", S_N))                            # the fill iteration: "tile -1"
    g.raw("s_waitcnt lgkmcnt(0)")
    g.lgkm = []
    g.wait_vm_tag("dma_first")                   # tile 0 has landed; the K / V and E fragments may still be in flight (first used by MFMAs)
    g.raw("s_barrier")


def loop_tail(g: Gen, b: int, masked: bool):
    nb = (b + 1) % 6
    g.salu(f"s_add_u32 {s(S_N)}, {s(S_N)}, 1", regs("s", S_N), regs("s", S_N))
    g.salu(f"s_cmp_ge_u32 {s(S_N)}, {s(S_NT)}", regs("s", S_N) | regs("s", S_NT), {"scc"})
    g.raw("s_cbranch_scc1 L_dkv_end_%=")
    if not masked:
        if b == 5:
            g.raw("s_branch L_dkv_u0_%=")
        return
    # a wave stays in the masked bodies while one of its keys is padded or a sub-tile is not below its diagonal yet (n - wk < 2)
    g.salu(f"s_cmp_lg_u32 {s(S_PADANY)}, 0", regs("s", S_PADANY), {"scc"})
    g.raw(f"s_cbranch_scc1 L_dkv_m{nb}_%=")
    g.salu(f"s_sub_u32 {s(S_TM)}, {s(S_N)}, {s(S_WK)}", regs("s", S_N) | regs("s", S_WK), regs("s", S_TM))
    g.salu(f"s_cmp_lt_i32 {s(S_TM)}, 2", regs("s", S_TM), {"scc"})
    g.raw(f"s_cbranch_scc1 L_dkv_m{nb}_%=")
    # switch to the main bodies: their counted waits assume the main loop's own history, so start them from a drained state
    if STAGGER:
        g.salu(f"s_cmp_eq_u32 {s(S_W)}, 0", regs("s", S_W), {"scc"})
        g.salu(f"s_cselect_b64 {s(S_W0M, 2)}, -1, 0", {"scc"}, regs("s", S_W0M, 2))
        g.salu(f"s_cselect_b64 {s(S_W1M, 2)}, 0, -1", {"scc"}, regs("s", S_W1M, 2))
    g.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    g.raw("s_nop 7")
    g.raw("s_nop 7")
    if STAMP:
        g.stamp(None)
    g.raw(f"s_branch L_dkv_u{nb}_%=")


def fixed_point_loop(g: Gen, masked: bool):
    """three rounds over the six bodies: the second reaches the loop-carried state of the wait / hazard trackers, the third is emitted"""
    texts = []
    for rnd in range(3):
        g.out = []
        for b in range(6):
            g.out.append(f"L_dkv_{'m' if masked else 'u'}{b}_%=:")
            body(g, b, masked=masked)
            if STAMP and not masked:
                g.raw(f"v_add_u32_e32 {v(V_STAMP + 7)}, 1, {v(V_STAMP + 7)}")
            loop_tail(g, b, masked)
        texts.append(list(g.out))
    assert texts[1] == texts[2], "the loop body is not a fixed point of the wait-count / hazard trackers"
    return texts[2]


def generate():
    g = Gen()
    prologue(g)
    # pipeline fill = variant 5 (n = -1: buffer 1, E rotation of n = 5) without the tile-n part
    body(g, 5, do_cur=False)
    g.drain()
    g.nop(16)
    g.salu(f"s_mov_b32 {s(S_N)}, 0", (), regs("s", S_N))
    if STAMP:
        for k in range(8):
            g.raw(f"v_mov_b32_e32 {v(V_STAMP + k)}, 0")
    pro_lines = list(g.out)
    m_lines = fixed_point_loop(g, masked=True)       # entered at n = 0 (dq0 = -wk <= 0: always a masked step)
    g.drain()                                        # (tracker state only: the main loop is entered through the drains of loop_tail)
    u_lines = fixed_point_loop(g, masked=False)
    g.out = []
    g.out.append("L_dkv_end_%=:")
    g.drain()
    if STAMP:                                    # sums -> parameter block + 3584 (+ 4 k); v246.. hold the same value in every lane
        g.raw(f"v_mov_b32_e32 {v(V_TMP)}, %8")
        for k in range(8):
            g.raw(f"ds_write_b32 {v(V_TMP)}, {v(V_STAMP + k)} offset:{3584 + 4 * k}")
        g.raw("s_waitcnt lgkmcnt(0)")
    g.nop(16)                                    # the accumulators are read by compiler code behind the block
    g.raw("s_barrier")
    return pro_lines + m_lines + u_lines + g.out, g


def clobbers():
    c = [f"v{i}" for i in range(V_FIRST, V_LAST + 1)] + [f"a{i}" for i in range(A_FIRST, A_LAST + 1)]
    c += [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "m0", "memory"]       # (EXEC is restored to all ones by every item that clears it)
    if STAMP:
        c += [f"v{i}" for i in range(V_STAMP, V_STAMP + 8)] + [f"s{i}" for i in range(S_STAMP, S_STAMP + 4)]
    return c


def main():
    global STAMP, PEEL
    here = os.path.dirname(os.path.abspath(__file__))
    if not (ST_PINS or STAGGER or ST_CACHE != "nt"):
        for STAMP in (False, True):
            PEEL = STAMP_PEEL if STAMP else 0
            write(here)
        PEEL = 0
    STAMP = False
    PEEL = int(os.environ.get("MGX_DKV64_PEEL", "0"))
    if PEEL or ST_PINS or STAGGER or ST_CACHE != "nt":          # experiment builds: a loop of their own, never the tracked one
        PEEL = PEEL or 1 << 20
        write(here)


def write(here):
    lines, g = generate()
    path = os.path.join(here, "rel_attn_dkv64_loop_stamp.inc" if STAMP else "rel_attn_dkv64_loop_peel.inc" if PEEL else "rel_attn_dkv64_loop.inc")
    import io
    f = io.StringIO()
    if True:
        f.write("// GENERATED by gen_dkv_asm.py -- do not edit.  The hand-scheduled main loop of rel_attn_dkv64_kernel (one asm statement):\n")
        f.write("// operands %0..%7 = dk[0][0], dk[0][1], dv[0][0], dv[0][1], dk[1][0], dk[1][1], dv[1][0], dv[1][1] (\"+a\"), %8 = LDS address of the\n")
        f.write("// wave's parameter block (\"s\").  Register map, schedule and hazard rules: gen_dkv_asm.py.\n")
        f.write("#define MGX_DKV64_LOOP_ASM \\\n")
        for ln in lines:
            if ln.startswith(";"):
                f.write(f"    /* {ln[1:].strip()} */ \\\n")
            else:
                f.write(f'    "{ln}\\n\\t" \\\n')
        f.write('    ""\n')
        f.write("#define MGX_DKV64_LOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
    write_if_changed(path, f.getvalue())
    n_loop = sum(1 for ln in lines if not ln.startswith(";") and not ln.endswith(":"))
    print(f"wrote {path}: {n_loop} instructions, s_nop wait states inserted: {g.nops}; counts {g.stats}", file=sys.stderr)


if __name__ == "__main__":
    main()
