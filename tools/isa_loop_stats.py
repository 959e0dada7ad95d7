"""Instruction mix of the innermost loops of a kernel in a hipcc -save-temps .s file (no GPU needed).
usage: python tools/isa_loop_stats.py file.s [kernel-substring] [--dump N]   (loops listed with their MFMA counts)"""
import collections, re, sys
path = sys.argv[1]
ksub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else None
lines = open(path).read().split("\n")
# kernel extents
starts = [i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", l) and not l.startswith(".L")]
kern = [(i, lines[i].split(":")[0]) for i in starts if ksub in lines[i]]
for k0, name in kern:
    k1 = min([i for i in starts if i > k0] + [len(lines)])
    labels = {}
    for i in range(k0, k1):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m: labels[m.group(1)] = i
    loops = []
    for i in range(k0, k1):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(f"== {name}: {len(loops)} backward branches")
    for n, (a, b) in enumerate(loops):
        body = [l.strip() for l in lines[a:b + 1] if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
        ops = collections.Counter(re.sub(r"_e32|_e64|_sdwa|_dpp", "", l.split()[0]) for l in body)
        mf = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
        if mf == 0: continue
        cat = collections.Counter()
        for k, v in ops.items():
            c = ("mfma" if k.startswith("v_mfma") else "accvgpr" if "accvgpr" in k else "valu" if k.startswith("v_") else
                 "ds_w" if k.startswith("ds_write") else "ds_r" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "scratch_")) else
                 "wait/nop" if k in ("s_waitcnt", "s_nop") else "salu")
            cat[c] += v
        print(f"  loop {n} lines {a+1}-{b+1}: {len(body)} instr, {mf} MFMA ({len(body)/mf:.1f}/MFMA)  " + "  ".join(f"{k} {v}" for k, v in sorted(cat.items())))
        print("     scratch ops:", sum(v for k, v in ops.items() if k.startswith("scratch_")), " vmcnt waits:",
              collections.Counter(l for l in body if l.startswith("s_waitcnt") and "vmcnt" in l).most_common(6))
        if dump == n:
            print("\n".join(lines[a:b + 1]))
