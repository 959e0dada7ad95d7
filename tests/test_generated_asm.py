"""The hand-scheduled main loops are generated code (csrc/gen_dkv_asm.py, csrc/gen_gemm_asm.py -> csrc/*_loop.inc, tracked so that
the schedules can be read and diffed): the generators must be deterministic, reproduce the tracked files, and every loop they emit
must be a fixed point of their wait-count / hazard trackers (asserted inside the generators).  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "musicgeneration_amd", "csrc")
FILES = {"gen_dkv_asm.py": ["rel_attn_dkv64_loop.inc"], "gen_gemm_asm.py": ["linear_dw_ring4_loop.inc", "linear_ring4_loop.inc"]}


def test_generators_reproduce_the_tracked_loops(tmp_path):
    # run copies of the generators in a scratch directory (they write beside themselves)
    for name in ("asm_sched.py", *FILES):
        with open(os.path.join(CSRC, name)) as f, open(tmp_path / name, "w") as g:
            g.write(f.read())
    env = {k: v for k, v in os.environ.items() if not k.startswith("MGX_")}        # (the diagnostic switches of the generators)
    for gen, outs in FILES.items():
        r = subprocess.run([sys.executable, str(tmp_path / gen)], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        for o in outs:
            with open(tmp_path / o) as f, open(os.path.join(CSRC, o)) as g:
                assert f.read() == g.read(), f"{o} is not what {gen} generates: rebuild (python -m musicgeneration_amd._build) and commit it"


def test_every_mfma_gap_of_the_gemm_loops_is_short():
    """a schedule property the kernels were designed for: no long run of non-MFMA instructions between two MFMAs of a loop body"""
    for name, limit in (("linear_dw_ring4_loop.inc", 24), ("linear_ring4_loop.inc", 24)):
        gap, worst, in_loop = 0, 0, False
        for line in open(os.path.join(CSRC, name)):
            s = line.strip().strip('"\\ ').replace("\\n\\t", "")
            if "L_dw4_v0_0" in s or "L_r4_loop" in s:
                in_loop = True
            if "L_dw4_end" in s or "==== exit" in s:
                in_loop = False
            if not in_loop or s.startswith("/*") or s.endswith(":") or not s:
                continue
            if s.startswith("v_mfma"):
                worst, gap = max(worst, gap), 0
            else:
                gap += 1
        assert 0 < worst <= limit, f"{name}: {worst} instructions between two MFMAs of a loop"


def test_ring4_tile_statement_waits_are_sized_for_the_epilogue_hipcc_emits(tmp_path):
    """the tile statement of linear_ring4_kernel is entered with VMEM traffic in flight -- the next tile's first two
    stages, requested by the previous statement, and, younger, the epilogue's global stores -- and its first counted waits
    (`s_waitcnt vmcnt(16 + MGX_RING4_EPI_STORES)` for stage 0) are only right if the epilogue hipcc compiled issues AT LEAST
    MGX_RING4_EPI_STORES VMEM operations per wave (more is the safe direction: the wait then covers some of them too).
    Compile linear.hip to gfx950 assembly and count: every instantiation must hold >= EPI_STORES global stores outside the
    statement, and the statement itself must no longer open with a full drain (which made those counted waits dead code)."""
    import re
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    inc = open(os.path.join(CSRC, "linear_ring4_loop.inc")).read()
    epi = int(re.search(r"#define MGX_RING4_EPI_STORES (\d+)", inc).group(1))
    for name in ("MGX_RING4_NT_ASM", "MGX_RING4_NN_ASM"):
        first = re.search(name + r" \\\n(?:\s*/\*.*?\*/ \\\n)*\s*\"([^\"\\]+)", inc).group(1)
        assert first.strip() == "s_waitcnt lgkmcnt(0)", f"{name} opens with `{first}`: a VMEM drain at entry defeats the counted waits"
        assert f"s_waitcnt vmcnt({16 + epi})" in inc
    out = tmp_path / "linear.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                        "--cuda-device-only", "-w", os.path.join(CSRC, "linear.hip"), "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    found = 0
    for m in re.finditer(r"^(_Z19linear_ring4_kernelILb[01]ELi\dEE\w+):.*?^\.Lfunc_end\d+:", asm, re.S | re.M):
        body = m.group(0)
        stores = len(re.findall(r"\bglobal_store_dwordx4\b", body))          # (the statement stores nothing: all of them are the epilogue's)
        assert stores >= epi, f"{m.group(1)}: {stores} epilogue stores < MGX_RING4_EPI_STORES = {epi}"
        found += 1
    assert found == 4, "linear_ring4_kernel<false,0>, <true,0>, <true,1>, <true,2>"


def test_every_register_the_asm_statements_name_is_in_their_clobber_lists():
    """the asm statements use FIXED registers (everything but the accumulators, which are compiler-allocated
    operands) and tell hipcc so through their clobber lists; a generator edit that reaches for a register outside its list would
    silently corrupt a value the compiler keeps there.  Static check of the tracked loops: every vN / sN / aN (and every range
    v[a:b]) in the instruction text is named by the statement's MGX_*_CLOBBERS."""
    import re
    for name, cl_macro, asm_macros in (("rel_attn_dkv64_loop.inc", "MGX_DKV64_LOOP_CLOBBERS", ["MGX_DKV64_LOOP_ASM"]),
                                       ("linear_dw_ring4_loop.inc", "MGX_DW4_LOOP_CLOBBERS", ["MGX_DW4_LOOP_ASM"]),
                                       ("linear_ring4_loop.inc", "MGX_RING4_CLOBBERS", ["MGX_RING4_NT_ASM", "MGX_RING4_NN_ASM"])):
        text = open(os.path.join(CSRC, name)).read()
        cl = re.search(r"#define " + cl_macro + r" (.*)", text).group(1)
        clobbered = set(re.findall(r'"([vsa]\d+)"', cl))
        assert {"vcc", "scc", "m0", "memory"} <= set(re.findall(r'"(\w+)"', cl))
        used = set()
        n_lines = 0
        for line in text.split("\n"):
            m = re.match(r'\s*"([^"]*?)(?:\\n\\t)?" \\?$', line)
            if not m:
                continue
            ins = m.group(1).split(";")[0]
            if not ins or ins.endswith(":"):
                continue
            n_lines += 1
            for kind, a, b in re.findall(r"\b([vsa])\[(\d+):(\d+)\]", ins):
                used |= {f"{kind}{i}" for i in range(int(a), int(b) + 1)}
            ins_wo_ranges = re.sub(r"\b[vsa]\[\d+:\d+\]", "", ins)
            # (operand tokens only: skip the mnemonic, which contains things like 'b32' / 'f32' but never a bare vN / sN / aN)
            for tok in re.findall(r"(?<![\w.])([vsa]\d+)(?![\w\[])", ins_wo_ranges.split(None, 1)[1] if " " in ins_wo_ranges else ""):
                used.add(tok)
        assert n_lines > 500, f"{name}: parsed only {n_lines} instruction lines"
        missing = sorted(used - clobbered, key=lambda r: (r[0], int(r[1:])))
        assert not missing, f"{name}: registers used by the asm text but absent from {cl_macro}: {missing[:20]}"
