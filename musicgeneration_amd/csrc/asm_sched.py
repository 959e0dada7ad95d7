#!/usr/bin/env python3
"""Shared machinery of the generated gfx950 main loops (gen_dkv_asm.py, gen_fwd_asm.py): instruction emission with the bookkeeping
hipcc does not do inside an asm statement -- counted s_waitcnt from the in-order LDS / VMEM queues, the wait states between an MFMA
and the readers of its result (and the other hazards of such loops) padded with s_nop -- and the list scheduler that assigns the
non-MFMA instructions of a loop body ("fillers") to the shadows of its MFMAs."""
from __future__ import annotations

MFMA_TO_VALU = 13      # wait states between an MFMA and any other instruction touching its result (hipcc pads s_nop 11)
VALU_TO_MFMA = 3       # a VALU-written register as an MFMA operand
TRANS_TO_VALU = 2


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def s(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


def regs(prefix, i, n=1):
    return {f"{prefix}{k}" for k in range(i, i + n)}


class Gen:
    def __init__(self):
        self.out = []                 # text lines
        self.pos = 0                  # wait-state clock (instructions issued; s_nop N counts N+1)
        self.lgkm = []                # outstanding LDS operations: sets of destination registers, oldest first
        self.vm = []                  # every VMEM operation issued so far: (tag, destination registers)
        self.vm_done = 0              # operations [0, vm_done) are known complete
        self.mfma_d = {}              # register -> clock of the last MFMA that wrote it
        self.mfma_c = {}              # register -> clock of the last MFMA that read it as srcC / A / B (WAR)
        self.valu_w = {}              # register -> clock of the last VALU write
        self.trans_w = {}             # register -> clock of the last transcendental write
        self.store_r = {}             # register -> clock of the last VMEM store that reads it
        self.nops = 0
        self.m0_w = None
        self.stats = {}
        self.s_stamp, self.v_stamp = 10, 246      # stamp builds: SGPRs s_stamp..+3, cycle sums in VGPRs v_stamp..

    # ---- low level -------------------------------------------------------------------------------------------------
    def raw(self, text, ws=1):
        self.out.append(text)
        self.pos += ws

    def comment(self, text):
        self.out.append(f"; {text}")

    def nop(self, states):
        while states > 0:
            k = min(states, 8)
            self.raw(f"s_nop {k - 1}", k)
            self.nops += k
            states -= k

    def _wait_lgkm(self, touched):
        idx = -1
        for i, d in enumerate(self.lgkm):
            if d & touched:
                idx = i
        if idx >= 0:
            k = len(self.lgkm) - 1 - idx
            self.raw(f"s_waitcnt lgkmcnt({min(k, 15)})")
            self.lgkm = self.lgkm[len(self.lgkm) - min(k, 15):] if k > 0 else []

    def _wait_vm(self, touched):
        idx = -1
        for i in range(self.vm_done, len(self.vm)):
            if self.vm[i][1] & touched:
                idx = i
        if idx >= 0:
            self.wait_vm_index(idx)

    def wait_vm_index(self, idx):
        k = len(self.vm) - 1 - idx
        self.raw(f"s_waitcnt vmcnt({min(k, 63)})")
        self.vm_done = max(self.vm_done, idx + 1)

    def wait_vm_tag(self, tag):
        idx = max((i for i in range(len(self.vm)) if self.vm[i][0] == tag), default=-1)
        if idx >= self.vm_done:
            self.wait_vm_index(idx)

    def stamp(self, k):
        """diagnostic builds: add the cycles since the previous stamp to sum k (v246 + k; every lane holds the same value).  Reading
        the clock waits for lgkmcnt(0): a stamped build is slower, read its SHARES"""
        self.raw(f"s_memtime {s(self.s_stamp, 2)}")
        self.raw("s_waitcnt lgkmcnt(0)")
        self.lgkm = []
        self.raw(f"s_sub_u32 {s(self.s_stamp + 3)}, {s(self.s_stamp)}, {s(self.s_stamp + 2)}")
        self.raw(f"s_mov_b32 {s(self.s_stamp + 2)}, {s(self.s_stamp)}")
        if k is not None:
            self.raw(f"v_add_u32_e32 {v(self.v_stamp + k)}, {s(self.s_stamp + 3)}, {v(self.v_stamp + k)}")

    def drain(self):
        self.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        self.lgkm = []
        self.vm_done = len(self.vm)

    def _hazards(self, kind, reads, writes):
        need = 0
        touched = reads | writes
        for r in touched:
            if r in self.mfma_d and kind != "mfma_acc_same":
                need = max(need, self.mfma_d[r] + MFMA_TO_VALU - self.pos)
        if kind.startswith("mfma"):
            for r in reads:
                if r in self.valu_w:
                    need = max(need, self.valu_w[r] + VALU_TO_MFMA - self.pos)
        else:
            for r in writes:
                if r in self.mfma_c:                       # WAR on an operand of an MFMA in flight
                    need = max(need, self.mfma_c[r] + 6 - self.pos)
                if r in self.store_r:
                    need = max(need, self.store_r[r] + 3 - self.pos)
        if kind == "valu":
            for r in reads:
                if r in self.trans_w:
                    need = max(need, self.trans_w[r] + TRANS_TO_VALU - self.pos)
        if "m0" in reads and kind == "vmem" and self.m0_w is not None:
            need = max(need, self.m0_w + 2 - self.pos)      # SALU write of M0 -> LDS-DMA: one wait state
        if need > 0:
            self.nop(need)

    def emit(self, kind, text, reads=(), writes=()):
        reads, writes = set(reads), set(writes)
        self._wait_lgkm(reads | writes)
        self._wait_vm(reads | writes)
        self._hazards(kind, reads, writes)
        self.raw(text)
        self.stats[kind] = self.stats.get(kind, 0) + 1
        at = self.pos - 1
        if kind.startswith("mfma"):
            for r in writes:
                self.mfma_d[r] = at
            for r in reads:
                self.mfma_c[r] = at
        else:
            for r in writes:
                self.mfma_d.pop(r, None)
                if kind in ("valu", "trans"):
                    self.valu_w[r] = at
                if kind == "trans":
                    self.trans_w[r] = at
                elif r in self.trans_w:
                    del self.trans_w[r]
        if "m0" in writes:
            self.m0_w = at
        if kind == "lds":
            self.lgkm.append(writes)
        if kind == "store":
            for r in reads:
                self.store_r[r] = at

    def prewait(self, touched):
        """wait now for every outstanding load that writes a register of `touched` (the operands of a whole MFMA chain): one
        s_waitcnt pair instead of a descending ladder in front of every MFMA of the chain"""
        self._wait_lgkm(set(touched))
        self._wait_vm(set(touched))

    # ---- instructions ----------------------------------------------------------------------------------------------
    def mfma(self, d, a_, b_, c=None, dn=16):
        """d (+)= a_ * b_; registers given as (prefix, index), or ("%", "%N") for a whole asm operand (compiler-allocated tuple)"""
        def opnd(x, n):
            if x[0] == "%":
                return x[1], {x[1]}
            return f"{x[0]}[{x[1]}:{x[1] + n - 1}]", regs(x[0], x[1], n)
        dt, dr = opnd(d, dn)
        at, ar = opnd(a_, 4)
        bt, br = opnd(b_, 4)
        if c == 0:
            self.emit("mfma", f"v_mfma_f32_32x32x16_bf16 {dt}, {at}, {bt}, 0", ar | br, dr)
        else:
            # accumulate chain on the same registers: no wait states needed after the previous MFMA of the chain
            same = all(r in self.mfma_d for r in dr)
            self.emit("mfma_acc_same" if same else "mfma", f"v_mfma_f32_32x32x16_bf16 {dt}, {at}, {bt}, {dt}", ar | br | dr, dr)

    def mfma_op(self, opnd, a_, b_, c0=False, cv=None):
        """accumulator given as an asm operand (%N, compiler-allocated AGPRs): never touched by anything else in the block; c0: C = 0;
        cv: C = the 16 VGPRs from cv on (D = C + A B written to the operand)"""
        ar, br = regs(a_[0], a_[1], 4), regs(b_[0], b_[1], 4)
        at = f"{a_[0]}[{a_[1]}:{a_[1] + 3}]"
        bt = f"{b_[0]}[{b_[1]}:{b_[1] + 3}]"
        ct = f"v[{cv}:{cv + 15}]" if cv is not None else ("0" if c0 else opnd)
        self.emit("mfma", f"v_mfma_f32_32x32x16_bf16 {opnd}, {at}, {bt}, {ct}", ar | br | (regs("v", cv, 16) if cv is not None else set()), set())

    def valu(self, text, reads, writes, trans=False):
        self.emit("trans" if trans else "valu", text, reads, writes)

    def salu(self, text, reads=(), writes=()):
        self.emit("salu", text, reads, writes)

    def ds_read(self, text, addr, dst):
        self.emit("lds", text, addr, dst)

    def vmem_load(self, text, tag, reads, dst):
        self.emit("vmem", text, reads, set())
        self.vm.append((tag, set(dst)))

    def vmem_dma(self, text, tag, reads):
        self.emit("vmem", text, reads, set())
        self.vm.append((tag, set()))

    def vmem_store(self, text, tag, reads):
        self.emit("store", text, reads, set())
        self.vm.append((tag, set()))


def crow(r, hh=0):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def salu_items(g, ops):
    """one item per SALU instruction, except that an s_addc_u32 stays glued to the s_add_u32 whose carry (SCC) it consumes:
    the scheduler interleaves items of different chains, and nearly every SALU instruction rewrites SCC"""
    out = []
    i = 0
    while i < len(ops):
        grp = [ops[i]]
        while i + 1 < len(ops) and ops[i + 1][0].startswith("s_addc_u32"):
            i += 1
            grp.append(ops[i])
        out.append(lambda grp=grp: [g.salu(*o) for o in grp])
        i += 1
    return out


class Item:
    """one filler (a function that emits one or a few instructions) with its scheduling constraints: it may be placed in the
    shadow of MFMA `earliest` .. `deadline` (gap g = after MFMA g, before MFMA g+1), after every item in `deps`"""

    def __init__(self, fn, cost, earliest=1, deadline=44, deps=(), pin=None, name=""):
        self.fn, self.cost, self.earliest, self.deadline, self.deps, self.pin, self.name = fn, cost, earliest, deadline, list(deps), pin, name
        self.gap = None


COST = {"valu": 4, "trans": 8, "lds": 4, "lds128": 8, "salu": 3, "vmem": 8, "store": 24, "sync": 4}
GAP_BUDGET = 22        # issue cycles of fillers an MFMA (32 cycles, 8 of them its own issue) is asked to hide


def chain(items):
    """items must keep their order"""
    for x, y in zip(items, items[1:]):
        y.deps.append(x)
    return items


def try_schedule(items, ngaps, budget):
    for it in items:
        it.gap = None
    table = {g: [] for g in range(1, ngaps + 1)}
    todo = list(items)
    over = 0
    for g in range(1, ngaps + 1):
        used = 0
        progress = True
        while progress:
            progress = False
            ready = [it for it in todo if it.earliest <= g and all(d.gap is not None for d in it.deps) and (it.pin is None or it.pin == g)]
            ready.sort(key=lambda it: (it.deadline, it.earliest))
            for it in ready:
                must = it.deadline <= g or it.pin == g
                if must or used + it.cost <= budget:
                    it.gap = g
                    table[g].append(it)
                    todo.remove(it)
                    used += it.cost
                    progress = True
                    break
        over = max(over, used - budget)
    return table, todo, over


def schedule(items, ngaps=44, budget0=None):
    """earliest-deadline-first list scheduling with the smallest uniform per-gap issue budget that places every filler inside its
    window (the fillers of a step cost more than 44 MFMA shadows hide: spread the excess evenly instead of piling it up)"""
    budget = GAP_BUDGET if budget0 is None else budget0
    while True:
        table, todo, over = try_schedule(items, ngaps, budget)
        if not todo and over <= 8:
            break
        budget += 1
        assert budget < 200, f"unschedulable: {[it.name for it in todo][:8]}"
    late = [it for g in table for it in table[g] if g > it.deadline]
    assert not late, f"fillers placed after their deadline: {[(it.name, it.gap, it.deadline) for it in late][:8]}"
    return table, budget




def write_if_changed(path, text):
    """write a generated file only when its content differs: an unchanged schedule keeps its mtime, so a variant build (which
    runs the generators too) does not make the product library look stale (musicgeneration_amd/_build.py: _stale)"""
    try:
        with open(path) as f:
            if f.read() == text:
                return False
    except OSError:
        pass
    with open(path, "w") as f:
        f.write(text)
    return True
