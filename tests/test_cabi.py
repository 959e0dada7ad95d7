"""CPU-side checks of the C-ABI boundary: libmgx.so builds/loads, exports every symbol declared in
include/mgx.h, the ctypes table covers the header, and the product path fails loudly without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from musicgeneration_amd import _build, _lib
    path = _build.build()
    lib = ctypes.CDLL(path)
    names = _header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mgx.h but not exported by libmgx.so"
    # the ctypes signature table binds exactly the header's functions (mgx_last_error is bound separately)
    assert set(_lib.SIGNATURES) | {"mgx_last_error"} == set(names)


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "mgx.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)          # declarations only (comments cite torch ops)
    assert "torch" not in code.lower() and "at::" not in code and "#include <hip" not in code
    assert "hipStream_t" not in code                            # the stream crosses the ABI as void*


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_fails_loudly_without_gpu():
    from musicgeneration_amd import _lib, ops
    from musicgeneration_amd.network import MusicTransformer
    lib = _lib.load()
    assert lib.mgx_abi_version() == _lib.EXPECTED_ABI
    assert lib.mgx_device_count() < 0 and b"hipGetDeviceCount" in lib.mgx_last_error()
    with pytest.raises(_lib.MgxError):
        ops.pad_bitmap(torch.zeros(2, 32, dtype=torch.int32), 5)          # CPU tensor: no fallback
    mt = MusicTransformer(embedding_dim=128, vocab_size=309, num_layer=1, max_seq=32)
    with pytest.raises(_lib.MgxError):
        mt(torch.zeros(1, 32, dtype=torch.int32))


def test_shape_errors_have_messages():
    """argument validation happens on the host before any launch, so it is testable without a GPU"""
    from musicgeneration_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    aligned = ctypes.c_void_p(4096)
    rc = lib.mgx_rel_attn_fwd(one, one, None, one, one, aligned, 1 << 20, 1, 33, 64, 64, None)
    assert rc == -1 and b"L%32==0" in lib.mgx_last_error()
    rc = lib.mgx_rel_attn_fwd(None, one, None, one, one, aligned, 1 << 20, 1, 32, 64, 64, None)
    assert rc == -2
    assert lib.mgx_rel_attn_fwd_workspace(2048) == 2048 * 128
    rc = lib.mgx_rel_attn_fwd(one, one, None, one, one, aligned, 64, 1, 32, 64, 64, None)      # workspace too small
    assert rc == -1 and b"workspace" in lib.mgx_last_error()
    rc = lib.mgx_linear_fwd(one, one, None, one, 4, 4, 48, 0, None)
    assert rc == -1 and b"K%64==0" in lib.mgx_last_error()
    rc = lib.mgx_add_ln_fwd(one, one, one, one, one, one, one, 4, 4100, 1e-6, 0.0, 0, None)
    assert rc == -1
    # three f32 [B,h,L] statistics (delta, -lse log2e, -delta) + two fragment-ordered bf16 copies of E [L,64] + the causal half
    # of dS by (query tile, key tile): 64*65/2 tiles of 32x32 bf16 per (b,h)
    assert lib.mgx_rel_attn_bwd_workspace(8, 2048, 512) == 3 * 8 * 8 * 2048 * 4 + 2 * 64 * 2048 * 2 + 8 * 8 * (64 * 65 // 2) * 2048
    # deterministic mode: argument validation (no GPU needed)
    assert lib.mgx_deterministic() == 0
    assert lib.mgx_set_deterministic(ctypes.c_void_p(12), 1 << 20) == -1 and b"aligned" in lib.mgx_last_error()
    assert lib.mgx_set_deterministic(ctypes.c_void_p(64 << 20), 1 << 20) == -1 and b"32 MiB" in lib.mgx_last_error()   # aligned, too small
    assert lib.mgx_set_deterministic_stream(ctypes.c_void_p(1), ctypes.c_void_p(64 << 20), 32 << 20) == -1 and b"first" in lib.mgx_last_error()
    assert lib.mgx_set_deterministic(None, 0) == 0 and lib.mgx_deterministic() == 0
    # stream registry (no stream is created or touched: the library only remembers the number)
    assert lib.mgx_stream_set_cus(ctypes.c_void_p(0x1000), 64) == 0 and lib.mgx_stream_cus(ctypes.c_void_p(0x1000)) == 64
    assert lib.mgx_stream_set_cus(ctypes.c_void_p(0x1000), 0) == 0
    assert lib.mgx_add_ln_bwd_workspace(100, 512) == 512 * 3 * 512 * 4       # 512 blocks of column partials
    rc = lib.mgx_add_ln_bwd(one, one, one, one, one, one, one, one, one, one, None, one, 16, 8, 512, 0.0, 0, None)
    assert rc == -1 and b"workspace" in lib.mgx_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "musicgeneration_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """The boundary is a C ABI: include/mgx.h must compile as strict C99 (no C++/torch types) and a C program must link
    against libmgx.so and get the documented error codes back -- no Python, no GPU needed for argument validation."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    from musicgeneration_amd import _lib
    _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "mgx.h"\n#include <stdio.h>\n'
        'int main(void) {\n'
        '    mgx_dw_problem p = {0};\n'
        '    int rc = mgx_linear_dw_grouped(&p, 1, 128, NULL, 0, NULL);\n'
        '    printf("%d %d %s\\n", mgx_abi_version(), rc, mgx_last_error());\n'
        '    return 0;\n}\n')
    exe = str(tmp_path / "abi")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                    "-o", exe, "-L", libdir, "-l:libmgx.so", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split(None, 2)
    assert int(out[0]) >= 8 and int(out[1]) == -2 and "NULL pointer" in out[2]


def test_stale_library_abi_is_rejected(monkeypatch):
    """a left-over libmgx.so of another ABI exports the same names: load() must refuse it instead of calling it with
    shifted arguments"""
    from musicgeneration_amd import _lib
    _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "EXPECTED_ABI", _lib.EXPECTED_ABI + 1)
    with pytest.raises(_lib.MgxError, match="ABI version"):
        _lib.load()
    monkeypatch.setattr(_lib, "EXPECTED_ABI", _lib.EXPECTED_ABI - 1)
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.load().mgx_abi_version() == _lib.EXPECTED_ABI
    # the header's constant, the library and the binding agree
    hdr = open(os.path.join(ROOT, "include", "mgx.h")).read()
    assert f"#define MGX_ABI_VERSION {_lib.EXPECTED_ABI}" in hdr


def test_bench_refuses_a_world_size_mismatch():
    """`bench.py --gpus N` must never print a line for another N: launched with a WORLD_SIZE that differs from --gpus it
    exits non-zero before touching the GPU; without a launcher and without N devices the self-launch refuses too."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"metric"' not in r.stdout
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and '"metric"' not in r.stdout and "HIP device" in r.stderr


def test_dw_grouped_workspace_says_which_weights_take_the_ring_path():
    """mgx_linear_dw_grouped_workspace is host logic (the split-then-fix-up plan): > 0 exactly for the groups the ring kernel takes --
    many rows, weights that fill their 256 x 256 tiles to 60 % (ragged last tile row / column allowed) -- and it sizes one fp32
    partial tile per (tile, M-split).  Without a GPU the plan assumes 256 CUs."""
    from musicgeneration_amd import _lib, ops
    lib = _lib.load()

    def need(shapes, M):
        arr = (ops._DwProblem * len(shapes))()
        for i, (N, K) in enumerate(shapes):
            arr[i] = ops._DwProblem(16, 16, 16, None, N, K)
        return lib.mgx_linear_dw_grouped_workspace(ctypes.cast(arr, ctypes.c_void_p), len(shapes), M)

    block = [(1536, 512), (512, 512), (256, 512), (512, 256)]           # cfg2's encoder block: 20 tiles
    n = need(block, 131072)
    assert n > 0 and n % (20 * 65536 * 4) == 0 and 1 <= n // (20 * 65536 * 4) <= 256 // 20
    assert need([(448, 512)], 131072) > 0                                # the vocabulary projection: second tile row 192 of 256
    assert need([(384, 768), (768, 384)], 16384) > 0                     # cfg4's FFN weights: 75 % of their tiles
    assert need([(128, 768)], 16384) == 0                                # half a tile row: the 128 x 128 kernel
    assert need(block, 1024) == 0 and need(block, 131072 + 8) == 0       # few rows / M % 32 != 0: the 128 x 128 kernels
    # a mixed group is sized for its ring-shaped weights only
    assert need(block + [(128, 768)], 131072) == n
