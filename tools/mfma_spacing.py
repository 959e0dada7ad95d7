"""MFMA spacing of the dK/dV main loops, before / after (no GPU needed):
   (a) the hipcc-scheduled two-step main loop of the 32-key kernel (rel_attn_bwd.hip, from `hipcc -save-temps`),
   (b) one main body of the generated 64-key loop (csrc/rel_attn_dkv64_loop.inc).
For each: instructions per MFMA, the histogram of gap lengths (non-MFMA instructions between consecutive MFMAs), runs of back-to-back
MFMAs, and the first gaps verbatim.   python tools/mfma_spacing.py > profiles/r05_isa_mfma_spacing.txt"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(name, body, show=3):
    body = [l for l in body if l and not l.startswith(("/*", ";", ".")) and not l.endswith(":")]
    gaps, cur = [], []
    seen = False
    for l in body:
        if l.startswith("v_mfma"):
            if seen:
                gaps.append(cur)
            cur, seen = [], True
        elif seen:
            cur.append(l)
    gaps.append(cur)
    n_mf = sum(1 for l in body if l.startswith("v_mfma"))
    h = collections.Counter(len(g) for g in gaps)
    print(f"== {name}: {len(body)} instructions, {n_mf} MFMAs = {len(body) / n_mf:.1f} per MFMA")
    print("   gap length (non-MFMA instructions after an MFMA) -> number of gaps: " + "  ".join(f"{k}:{h[k]}" for k in sorted(h)))
    runs, r = [], 1
    for g in gaps[:-1]:
        if len(g) == 0:
            r += 1
        else:
            runs.append(r); r = 1
    runs.append(r)
    print(f"   runs of back-to-back MFMAs: {dict(sorted(collections.Counter(runs).items()))};  longest stretch without an MFMA: {max(len(g) for g in gaps)} instructions")
    kinds = collections.Counter()
    for g in gaps:
        for l in g:
            op = l.split()[0]
            kinds["valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_")) else
                  "wait/nop" if op in ("s_waitcnt", "s_nop") else "salu"] += 1
    print("   non-MFMA mix: " + "  ".join(f"{k} {v}" for k, v in sorted(kinds.items())))
    print(f"   first {show} gaps:")
    i = 0
    for l in body:
        if l.startswith("v_mfma"):
            i += 1
            if i > show + 1:
                break
        if i >= 1:
            print("      " + l)


# (a) hipcc
with tempfile.TemporaryDirectory() as td:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/musicgeneration_amd/csrc", "-Wno-unused-value",
           "-Wno-unused-result", "-c", "-x", "hip", f"{ROOT}/musicgeneration_amd/csrc/rel_attn_bwd.hip", "-o", "bwd.o", "-save-temps"]
    subprocess.run(cmd, cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(td, "rel_attn_bwd-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
k0 = next(i for i, l in enumerate(lines) if l.startswith("_Z19rel_attn_dkv_kernel"))
k1 = next((i for i in range(k0 + 1, len(lines)) if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", lines[i]) and not lines[i].startswith(".L")), len(lines))
labels = {m.group(1): i for i in range(k0, k1) if (m := re.match(r"^(\.LBB\d+_\d+):", lines[i]))}
loops = []
for i in range(k0, k1):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = [l.strip() for l in lines[labels[m.group(1)]:i + 1]]
        loops.append((sum(1 for l in body if l.startswith("v_mfma")), body))
main = [b for n, b in loops if n == 48]
report("hipcc, 32-key kernel rel_attn_dkv_kernel<true>: branch-free main loop, two steps (48 MFMAs) per trip", min(main, key=len))
print()
# (b) generated
inc = open(f"{ROOT}/musicgeneration_amd/csrc/rel_attn_dkv64_loop.inc").read().split("\n")
txt = [re.sub(r'^"|\\n\\t" \\$', "", l.strip()) for l in inc]
a = next(i for i, l in enumerate(txt) if "L_dkv_u0_%=:" in l)
b = next(i for i, l in enumerate(txt) if "L_dkv_u1_%=:" in l)
report("generated (gen_dkv_asm.py), 64-key kernel rel_attn_dkv64_kernel: main body 0 of 6, one step = two key tiles (44 MFMAs)", txt[a:b])
