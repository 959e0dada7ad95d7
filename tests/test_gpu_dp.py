"""Data-parallel parity with the REAL kernels: 2 ranks x batch B/2 (gloo all-reduce of the flat gradient buffer, issued
bucket by bucket from inside the backward; both ranks share the one MI355X of the test box) must follow the same
loss / parameter trajectory as 1 process x batch B.  (RCCL itself needs one GPU per rank; the driver's 8-GPU run
covers it.  The collective semantics, bucket callbacks, broadcast and 1/world scaling are what this test pins.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(world, out, port, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    worker = os.path.join(HERE, "_dp_gpu_worker.py")
    for attempt in range(3):
        if world == 1:
            cmd = [sys.executable, worker, out]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), worker, out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        # the rendezvous port was free when _free_port() looked and taken when the launcher bound it (a socket of the previous
        # test still closing): a failure of the launcher before any rank started, retried on another port
        if r.returncode != 0 and "EADDRINUSE" in r.stderr and attempt < 2:
            port = _free_port()
            if "MGX_TEST_PORT" in env:
                env["MGX_TEST_PORT"] = str(_free_port())
            continue
        break
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(out))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_ranks_match_one_process(tmp_path):
    one = _run(1, str(tmp_path / "one.json"), 0)
    two = _run(2, str(tmp_path / "two.json"), _free_port())
    assert two["buckets"] >= 3 and two["bytes_reduced"] > 0 and one["bytes_reduced"] == 0
    for a, b in zip(one["losses"], two["losses"]):
        assert abs(a - b) <= 2e-3 * abs(a), (one["losses"], two["losses"])     # bf16 kernels, atomics order
    assert abs(one["param_sum"] - two["param_sum"]) <= 1e-3 * one["param_abs"]
    assert one["losses"][-1] < one["losses"][0]


def test_rccl_world_of_one_matches_no_dp(tmp_path):
    """RCCL itself on the one GPU this box has: `init_process_group("nccl", world_size=1)` and a DataParallel that is
    forced through its whole collective path -- rank-0 broadcast of the flat parameter buffer, one asynchronous
    all-reduce per bucket on views of the flat gradient buffer issued from inside backward, wait_all before the fused
    Adam, the 4-byte loss-weight all-reduce.  A sum over one rank is the identity, so the run must follow the run
    without data parallelism to the run-to-run noise of the kernels' fp32 atomics (measured here by running the
    plain configuration twice); a mis-ordering between the RCCL stream and the backward kernels (an all-reduce reading a
    bucket before its last gradient kernel finished, Adam running before the reduce) shows up orders of magnitude above it."""
    one = _run(1, str(tmp_path / "one.json"), 0)
    again = _run(1, str(tmp_path / "again.json"), 0)
    rccl = _run(1, str(tmp_path / "rccl.json"), 0, MGX_TEST_RCCL1="1", MGX_TEST_PORT=str(_free_port()))
    assert rccl["describe"]["backend"] == "nccl" and rccl["describe"]["rccl_version"], rccl["describe"]
    print("RCCL version", rccl["describe"]["rccl_version"])
    assert rccl["buckets"] >= 3 and rccl["bytes_reduced"] > 0 and one["bytes_reduced"] == 0
    noise = max(abs(a - b) / abs(a) for a, b in zip(one["losses"], again["losses"]))
    # one pair of plain runs can agree by chance far better than the typical noise: the fp32 atomics (dE, dW splits, the embedding
    # gradient) differ in the last bits, and once such a bit flips the bf16 rounding of a shadow weight the loss of the NEXT step
    # moves by 1e-4 .. 5e-4 relative (seen: five steps equal to 4e-6, the sixth 2.9e-4 apart).  The floor sits above that and
    # 10x below what a mis-ordered bucket does (> 1e-2).
    tol = max(10 * noise, 1e-3)
    for a, b in zip(one["losses"], rccl["losses"]):
        assert abs(a - b) <= tol * abs(a), (one["losses"], rccl["losses"], noise)
    pnoise = abs(one["param_sum"] - again["param_sum"]) / one["param_abs"]
    # (floor 3e-5 of the parameters' L1 norm: one pair of plain runs measured 1.06e-5 against a third -- round 6 -- while the losses
    #  agreed to 1e-4; a bucket reduced before its last gradient kernel had finished loses whole gradient terms, >= 1e-3.  The
    #  deterministic-mode twin below asserts bit-identity, which is the sharp form of this test)
    assert abs(one["param_sum"] - rccl["param_sum"]) <= max(10 * pnoise, 3e-5) * one["param_abs"]


def test_deterministic_mode_repeats_bit_for_bit_and_rccl_world_of_one_equals_no_dp(tmp_path):
    """MGX_DETERMINISTIC=1 (include/mgx.h: mgx_set_deterministic): the cross-workgroup sums become order-independent integer
    atomics, so (a) two runs of the same six training steps end in the SAME parameter bytes and the same losses, and (b) the run
    whose every collective goes through RCCL (a world of one: the sum over one rank is the identity) equals them bit for bit --
    the noise floor test_rccl_world_of_one_matches_no_dp has to measure is gone."""
    one = _run(1, str(tmp_path / "one.json"), 0, MGX_DETERMINISTIC="1")
    again = _run(1, str(tmp_path / "again.json"), 0, MGX_DETERMINISTIC="1")
    rccl = _run(1, str(tmp_path / "rccl.json"), 0, MGX_DETERMINISTIC="1", MGX_TEST_RCCL1="1", MGX_TEST_PORT=str(_free_port()))
    assert one["deterministic"] and again["deterministic"] and rccl["deterministic"]
    assert one["losses"] == again["losses"] and one["param_hash"] == again["param_hash"]
    assert rccl["bytes_reduced"] > 0 and rccl["describe"]["backend"] == "nccl"
    assert one["losses"] == rccl["losses"] and one["param_hash"] == rccl["param_hash"]
    # ... and the mode changes the numbers only by the fixed-point quantisation of the partial sums
    plain = _run(1, str(tmp_path / "plain.json"), 0)
    assert not plain["deterministic"]
    assert abs(plain["losses"][0] - one["losses"][0]) <= 1e-6 * abs(one["losses"][0])
    assert abs(plain["grad_l2"] - one["grad_l2"]) <= 1e-5 * one["grad_l2"]


def test_deterministic_two_ranks_match_one_process_to_fp32_rounding(tmp_path):
    """data-parallel N ranks vs one process on the same global batch, loss rtol 1e-5 (fp32 reduce).  In deterministic
    mode nothing but the grouping of the sums differs between the two (each rank sums its rows, the all-reduce adds the ranks):
    the first step's loss and its gradient agree to fp32 rounding; later steps stay within the bf16-shadow flip noise."""
    one = _run(1, str(tmp_path / "one.json"), 0, MGX_DETERMINISTIC="1")
    two = _run(2, str(tmp_path / "two.json"), _free_port(), MGX_DETERMINISTIC="1")
    assert two["deterministic"] and two["bytes_reduced"] > 0
    assert abs(one["losses"][0] - two["losses"][0]) <= 1e-5 * abs(one["losses"][0])
    assert abs(one["grad_l2"] - two["grad_l2"]) <= 1e-5 * one["grad_l2"]
    assert abs(one["grad_sum"] - two["grad_sum"]) <= 1e-5 * one["grad_abs"]
    for a, b in zip(one["losses"], two["losses"]):
        assert abs(a - b) <= 1e-3 * abs(a), (one["losses"], two["losses"])


def _train_hash(side_cus, side_work=3, steps=3):
    """a few deterministic-mode optimiser steps of a small model; -> (losses, sha256 of the parameter buffer)"""
    import hashlib
    import torch
    from musicgeneration_amd import ops
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    V, d, nl, L, B = 90, 256, 3, 256, 16                  # M = 4096: the blocks' weight gradients take the ring path
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.1).cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, warmup_steps=20, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    g = torch.Generator().manual_seed(3)
    plan = ops.configure_streams(side_cus, 0, side_work=side_work) if side_cus else None
    losses = []
    try:
        with torch.cuda.stream(ops.main_stream()):
            for _ in range(steps):
                xf = torch.randint(0, V - 1, (B, L + 1), generator=g)
                xf[0, L - 20:] = V - 1
                x, y = xf[:, :-1].to(torch.int32).cuda(), xf[:, 1:].to(torch.int32).cuda()
                loss = lossf(mt(x), y)
                loss.backward()
                sch.step()
                opt.zero_grad()
                losses.append(loss.item())
        torch.cuda.synchronize()
    finally:
        if plan is not None:
            ops.configure_streams(0, 0)
    return losses, hashlib.sha256(mt.store().param.detach().cpu().numpy().tobytes()).hexdigest()


def test_two_stream_backward_is_bit_identical_in_deterministic_mode():
    """with the off-critical-path kernels of the backward (dE, the weight gradients) on a CU-masked
    side stream (ops.configure_streams) the parameters after three optimiser steps equal the one-stream run's BIT FOR BIT in
    deterministic mode -- the side stream has its own fixed-point scratch (mgx_set_deterministic_stream, ABI 18), the
    optimiser joins it before the update.  Both splits of the work (dE only; dE + weight gradients) and an uneven CU split."""
    from musicgeneration_amd import ops
    ops.set_deterministic(True)
    try:
        base = _train_hash(0)
        assert _train_hash(0) == base, "the one-stream run must repeat itself first"
        assert _train_hash(64) == base
        assert _train_hash(32, side_work=1) == base
        assert _train_hash(96, side_work=2) == base
    finally:
        ops.set_deterministic(False)


@pytest.mark.parametrize("knobs", [
    dict(MGX_TEST_RCCL_CUS="16"),                                        # bench.py --rccl-cus 16: compute stream masked to 240 CUs
    dict(MGX_TEST_NCCL_CHANNELS="2"),                                    # --nccl-channels 2
    dict(MGX_TEST_BUCKETS="2"),                                          # --buckets 2: two merged all-reduces
    dict(MGX_TEST_RCCL_CUS="8", MGX_TEST_SIDE_CUS="64", MGX_TEST_BUCKETS="2", MGX_TEST_NCCL_CHANNELS="4"),   # all of them + the side stream
])
def test_rccl_world_of_one_with_the_co_residency_mitigations_is_bit_identical(tmp_path, knobs):
    """the three mitigations DESIGN.md section 4 lists for RCCL's kernels sharing CUs with the backward's --
    a CU mask on the compute stream(s), few RCCL channels, merged buckets -- are flags (bench.py --rccl-cus / --nccl-channels /
    --buckets; this worker's MGX_TEST_* twins), and each leaves the deterministic-mode run through RCCL (a world of one: every
    collective issued, the sum over one rank is the identity) BIT-IDENTICAL to the run without data parallelism: masked
    streams, the side stream's bucket callbacks and the merged slices change no ordering that matters."""
    base = _run(1, str(tmp_path / "one.json"), 0, MGX_DETERMINISTIC="1")
    got = _run(1, str(tmp_path / "rccl.json"), 0, MGX_DETERMINISTIC="1", MGX_TEST_RCCL1="1", MGX_TEST_PORT=str(_free_port()), **knobs)
    assert got["describe"]["backend"] == "nccl" and got["bytes_reduced"] > 0
    if "MGX_TEST_BUCKETS" in knobs:
        assert len(got["allreduce_units"]) == 2, got["allreduce_units"]
    if "MGX_TEST_RCCL_CUS" in knobs:
        assert got["streams"]["reserved"] == int(knobs["MGX_TEST_RCCL_CUS"])
        assert got["streams"]["main"] == 256 - got["streams"]["reserved"] - got["streams"]["side"]
    if "MGX_TEST_NCCL_CHANNELS" in knobs:
        assert got["nccl_channels"] == knobs["MGX_TEST_NCCL_CHANNELS"]
    assert got["losses"] == base["losses"] and got["param_hash"] == base["param_hash"]


def test_bench_two_rank_gloo_rehearsal_on_one_device(tmp_path):
    """the whole N > 1 path of bench.py -- self-launch of the ranks, DataParallel with merged buckets, a masked compute stream,
    the `dp` block of the JSON line -- rehearsed with two gloo ranks sharing the one GPU of the box (a small model: this is
    about the plumbing; profiles/r05_bench_dp2_gloo_one_device.json is the same run at the bench shape)."""
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--batch", "4", "--seq-len", "512", "--layers", "2", "--d-model", "256", "--steps", "3", "--warmup", "2",
                        "--buckets", "2", "--rccl-cus", "16", "--no-cpu-baseline", "--no-decode", "--no-kernel-timing", "--no-cfg4"],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0
    dp = line["dp"]
    assert dp["world"] == 2 and dp["backend"] == "gloo" and dp["buckets"] == 2 and dp["rccl_cus"] == 16
    assert line["config"]["streams"]["main_cus"] == 240 and line["config"]["streams"]["reserved_for_rccl"] == 16
    assert dp["allreduce_bytes_per_step"] > 0 and dp["exposed_allreduce_ms_per_step"] is not None
