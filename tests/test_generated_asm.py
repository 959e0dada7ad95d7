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
