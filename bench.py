#!/usr/bin/env python3
"""bench.py -- training events/sec of the Music Transformer hot path on MI355X.

Metric (BASELINE.json): training events/sec, whole node, REMI seq_len=2048.  A "step" is one pass of
the hot path over one synthetic batch: forward + label-smoothed CE/accuracy + backward + Adam/Noam
step (+ gradient all-reduce when N > 1), dropout 0.2 as in the reference config.  Inputs are
device-resident before the timed region.  Workload = BASELINE.json configs[1] (cfg2):
REMI vocabulary V=337 (336 + pad), 6 layers, d_model=512 (8 heads x 64), L = max_seq = 2048, bf16
kernels with fp32 master weights / statistics / accumulation, per-GPU batch 128 (weak scaling; the
reference's own default is 6, config.py:35 -- larger batches only help both sides; 32 until round 3, 64 in rounds
3-5: the same tree reads 2.5 % higher at 128 than at 64 on one box, profiles/README.md round 6; the line carries
`batch64` and `batch8` blocks beside the main figure).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the driver contract): value, roofline (dominant kernel, timed
live with HIP events on the launch stream) and cpu_baseline (the oracle's eager fp32 CPU restatement
of the same step on a bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MI355X peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense bf16 MFMA; HBM3E spec)
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0

CFG2 = dict(vocab=337, layers=6, d_model=512, seq_len=2048, batch=128)
# BASELINE configs[3]: MuMIDI_EventSeq multi-track (V = 485 + pad), 12 layers, d_model 768 (12 heads), seq_len 4096, DP=8; its
# single-GPU share is per-GPU batch 4
CFG4 = dict(vocab=486, layers=12, d_model=768, seq_len=4096, batch=4)


def train_flops_per_event(nl, d, L, V):
    """ALGORITHMIC training FLOPs per event: 3 x [nl (10 d^2 + 3 L d) + 2 d V];
    causal half only, no credit for recompute / band over-compute / padding."""
    return 3.0 * (nl * (10.0 * d * d + 3.0 * L * d) + 2.0 * d * V)


def attn_flops_per_launch(B, L, d, units):
    """relative attention: one 'unit' = one L x L/2 x 64 product per head = B * L^2 * d flops
    (2*64 flops x L(L)/2 pairs x heads = L^2 d); forward = 3 units, backward = 6 units."""
    return units * float(B) * L * L * d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("cfg2", "cfg4"), default="cfg2", help="cfg2 (default): BASELINE configs[1], the metric's "
                    "workload.  cfg4: BASELINE configs[3] at its single-GPU share (MuMIDI V=486, 12 layers, d=768, L=4096, batch 4) as "
                    "the main line -- for profiling; the default run already reports it in its `cfg4` block")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4 block (N=1 only)")
    ap.add_argument("--deterministic", action="store_true", help="run with mgx_set_deterministic (order-independent integer "
                    "atomics for the cross-workgroup sums; also MGX_DETERMINISTIC=1): reported in config.deterministic")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling; default 128 for cfg2); rounds 1-3 were quoted at "
                    "32 and rounds 3-5 at 64, which read about 4.5 %% and 2.5 %% lower on the same box (profiles/README.md)")
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--d-model", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for timed CPU steps beyond the first three")
    ap.add_argument("--no-decode", action="store_true", help="skip the cfg5 decode block (N=1 only)")
    ap.add_argument("--decode-len", type=int, default=8192)
    ap.add_argument("--decode-batch", type=int, default=32)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "rehearsing the data-parallel path with several ranks on one GPU)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: issue the gradient all-reduces AFTER backward instead of bucket "
                    "by bucket from inside it (same results).  The step-time difference against the default run is what the overlap "
                    "buys; with the per-bucket times of the `dp` block it tells \"RCCL slow\" from \"overlap lost\" on a first multi-GPU run")
    ap.add_argument("--one-device", action="store_true", help="all ranks use cuda:0 (rehearsal with --backend gloo)")
    ap.add_argument("--side-cus", type=int, default=0, help="run the backward's off-critical-path, HBM-bound kernels (dE from the stored dS "
                    "tiles, the blocks' weight gradients) on a side stream restricted to this many CUs (multiple of 8 = whole-XCD-balanced "
                    "mask) beside the critical path, which gets the other CUs (ops.configure_streams; DESIGN.md 2.8).  0 = one stream")
    ap.add_argument("--side-work", choices=("de", "dw", "both"), default="both", help="with --side-cus: what the side stream runs -- the dE "
                    "kernel, the weight gradients, or both")
    ap.add_argument("--side-shared", action="store_true", help="with --side-cus: the main stream keeps the whole chip (only the side "
                    "stream is masked) instead of the complement of the side stream's CUs")
    ap.add_argument("--rccl-cus", type=int, default=0, help="N > 1: keep this many CUs (multiple of 8) free of the compute streams for "
                    "RCCL's kernels (CU-masked compute stream; co-residency mitigation 2 of DESIGN.md 4)")
    ap.add_argument("--nccl-channels", type=int, default=0, help="N > 1: NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = K for the ranks "
                    "(K channels = K workgroups of RCCL's ring kernels; mitigation 1 of DESIGN.md 4).  0 = RCCL's default")
    ap.add_argument("--buckets", type=int, default=0, help="N > 1: merge the per-layer gradient buckets into this many contiguous "
                    "groups (2: one all-reduce mid-backward, one at its end; mitigation 3 of DESIGN.md 4).  0 = one bucket per layer")
    a = ap.parse_args()
    base = CFG4 if a.workload == "cfg4" else CFG2
    for k, dflt in (("batch", base["batch"]), ("seq_len", base["seq_len"]), ("d_model", base["d_model"]),
                    ("layers", base["layers"]), ("vocab", base["vocab"])):
        if getattr(a, k) is None:
            setattr(a, k, dflt)
    return a


def time_kernels(B, L, d, M, reps=10):
    """Per-kernel durations (ms) of the attention kernels at the bench shape, HIP events on the
    current stream (the stream libmgx launches on)."""
    from musicgeneration_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    E = (torch.randn(M, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dctx = torch.randn(B, L, d, generator=g).to(torch.bfloat16).to(dev)
    dE = torch.zeros(M, 64, device=dev)
    ctx, lse = ops.rel_attn_fwd(qkv, E, None)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(ops._lib.load().mgx_rel_attn_bwd_workspace(*qkv.shape[:2], qkv.shape[2] // 3), dtype=torch.uint8, device=dev)
    ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, 15, dqkv, ws)
    torch.cuda.synchronize()

    # The four kernels are timed the way a layer runs them -- interleaved, one after the other -- not as ten back-to-back
    # launches of the same MFMA-dense kernel (the chip then holds a lower clock than inside the training step and the
    # per-kernel times read 15-30 % long against the rocprofv3 trace of the step).
    calls = [("rel_attn_fwd_kernel", lambda: ops.rel_attn_fwd(qkv, E, None))]
    # parts bits of mgx_rel_attn_bwd_parts; the dQ and dE kernels read the dS tiles the dK/dV kernel left in `ws`
    # (the pre-pass -- delta + E re-layout, two small kernels -- is its own entry, so that every other entry brackets exactly ONE
    #  kernel and can be compared with the rocprofv3 average of that kernel)
    for name, bit in (("attn_prepass_kernels", 1), ("rel_attn_dkv_kernel", 4), ("rel_attn_dq_lite_kernel", 2),
                      ("rel_attn_de_tiles_kernel", 8)):
        calls.append((name, lambda bit=bit: ops.rel_attn_bwd(qkv, E, None, ctx, dctx, lse, dE, bit, dqkv, ws)))
    for _, fn in calls:
        fn()
    evs = {name: [] for name, _ in calls}
    for _ in range(reps):
        for name, fn in calls:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs[name].append((e0, e1))
    torch.cuda.synchronize()
    out = {name: sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) for name, pairs in evs.items()}
    return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _time_cpu_trainer(V, d, nl, L, Bc, dropout, seconds, min_steps, max_steps):
    """one untimed step (allocator / thread-pool warm-up), then at least `min_steps` timed ones (>= 3) and more while
    the budget of `seconds` lasts"""
    from oracle import ref_cpu as R
    p = R.init_params(V, d, nl, L, seed=0)
    tr = R.CpuTrainer(p, pad=V - 1, d_cfg=d, dropout=dropout, accum=1)
    gen = torch.Generator().manual_seed(1234)
    nsteps, t_used = -1, 0.0
    while nsteps < min_steps or (t_used < seconds and nsteps < max_steps):
        xf = torch.randint(0, V - 1, (Bc, L + 1), generator=gen)
        t0 = time.time()
        tr.step(xf[:, :-1].to(torch.int32), xf[:, 1:].to(torch.int32))
        if nsteps >= 0:
            t_used += time.time() - t0
        nsteps += 1
    return Bc * L * nsteps / t_used, nsteps, t_used


def cpu_baseline(args):
    """The oracle's eager-PyTorch fp32 CPU restatement of the same training step (reference semantics:
    materialised L x L attention, dropout 0.2, Adam + Noam), bounded to ~10-30 s of CPU work: the bench workload
    (cfg2) at batch 2, and BASELINE cfg1 (the reference's own CPU-runnable case: MIDI-like V=309, 2 layers, d=256,
    L=512, batch 8) beside it."""
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    # threads: a 1-GPU box is handed a 16-core share of its host (gpurun: "size worker pools to the box's CPU share (16
    # for one GPU)"); more threads than that share only oversubscribe it and slow the eager ops down
    ncores = max(1, min(16, avail))
    torch.set_num_threads(ncores)
    V, d, nl, L = args.vocab, args.d_model, args.layers, args.seq_len
    Bc = args.cpu_batch
    v2, n2, t2 = _time_cpu_trainer(V, d, nl, L, Bc, args.dropout, args.cpu_seconds, 3, 3)
    v1, n1, t1 = _time_cpu_trainer(309, 256, 2, 512, 8, args.dropout, 4.0, 3, 20)
    return {"value": v2, "unit": "events/s", "cores": ncores, "kind": "port", "cpu_model": _cpu_model(),
            "host_cpus_visible": avail,
            "sample": f"{n2} timed steps (after one untimed) of the same cfg2 training step at batch {Bc} x L {L} "
                      f"(oracle/ref_cpu.py CpuTrainer, eager PyTorch fp32, {ncores} threads, {t2:.1f} s)",
            "cfg1": {"value": v1, "unit": "events/s",
                     "sample": f"{n1} timed steps (after one untimed) of BASELINE cfg1 (V=309, 2 layers, d=256, L=512, batch 8, fp32) in {t1:.1f} s, "
                               f"{ncores} threads"}}


def decode_bench(args):
    """BASELINE cfg5: autoregressive sampling to seq_len 8192 at batch 32, top-p 0.9, hipGraph-captured KV-cache decode
    (prior length 1, cfg2-shaped model).  HBM-bound: per step the K and V caches of every layer are read once."""
    from musicgeneration_amd.network import MusicTransformer
    Ld, Bd, d, nl, V = args.decode_len, args.decode_batch, args.d_model, args.layers, args.vocab
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=Ld, dropout=0.0).cuda().eval()
    prior = torch.randint(0, V - 1, (Bd, 1), device="cuda")
    mt.generate_cached(prior, 64, top_p=0.9, seed=1)          # warm-up (graph capture path)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mt.generate_cached(prior, Ld - 1, top_p=0.9, seed=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = Ld - 1
    assert out.shape == (Bd, Ld) and int(out.max()) < V
    # prompt handling: a prior of half the window, prefilled in one pass of the full-sequence kernels vs teacher-forced
    # step by step (the first decode step after the prompt is included in both)
    Pp = min(Ld // 2, 2048)
    prompt = torch.randint(0, V - 1, (Bd, Pp), device="cuda")
    pre = {}
    for mode in ("batched", "token"):
        mt.generate_cached(prompt[:, :65], 1, top_p=0.9, seed=1, prefill=mode)      # warm-up of that path
        torch.cuda.synchronize()
        tp = time.perf_counter()
        mt.generate_cached(prompt, 1, top_p=0.9, seed=0, prefill=mode)
        torch.cuda.synchronize()
        pre[mode] = time.perf_counter() - tp
    # cfg5's second reading: the Event_Melody family's own sampler, Event_Melody_RNN.generate (3 x GRU(512),
    # 308 events) for the same number of steps and batch: one captured graph per step, 9.4 MB of bf16 weights per step
    from musicgeneration_amd.melody_rnn import Event_Melody_RNN
    gru = Event_Melody_RNN(init_dim=32, event_dim=308, hidden_dim=512, rnn_layers=3, dropout=0.0).cuda().eval()
    init = torch.randn(Bd, 32, device="cuda")
    gru.generate(init, 64, greedy=0.0, seed=1)               # warm-up / capture path
    torch.cuda.synchronize()
    tg = time.perf_counter()
    gout = gru.generate(init, steps, greedy=0.0, seed=0)
    torch.cuda.synchronize()
    dtg = time.perf_counter() - tg
    assert tuple(gout.shape) == (steps, Bd) and int(gout.max()) < 308
    gru_w_bytes = 2 * sum(p.numel() for p in gru.parameters())
    kv_bytes = nl * 2 * Bd * d * 2 * (steps * (steps + 1) / 2)          # K and V rows read over the whole run
    w_bytes = steps * 2 * sum(p.numel() for p in mt.parameters())       # bf16 weights once per step
    last_step_bytes = nl * 2 * Bd * d * 2 * Ld + 2 * sum(p.numel() for p in mt.parameters())
    return {"workload": f"cfg5: sampling to seq_len {Ld}, batch {Bd}, top-p 0.9, graph-captured KV-cache decode, "
                        f"V={V} layers={nl} d_model={d}",
            "tokens_per_s": Bd * steps / dt, "ms_per_step": 1e3 * dt / steps, "seconds": dt,
            "algorithmic_bytes_per_step_mean": (kv_bytes + w_bytes) / steps,
            "algorithmic_bytes_last_step": last_step_bytes,
            "achieved_gbs": (kv_bytes + w_bytes) / dt / 1e9, "peak_gbs": PEAK_HBM_GBS,
            "frac": (kv_bytes + w_bytes) / dt / 1e9 / PEAK_HBM_GBS, "bound": "hbm",
            "prefill": {"prompt_tokens": Pp, "batch": Bd, "batched_ms": 1e3 * pre["batched"], "token_by_token_ms": 1e3 * pre["token"],
                        "batched_prompt_tokens_per_s": Bd * Pp / pre["batched"]},
            "event_melody_rnn": {"workload": f"Event_Melody_RNN.generate: {steps} steps, batch {Bd}, 3 x GRU(512), 308 events, "
                                             "temperature sampling, one captured graph per step",
                                 "tokens_per_s": Bd * steps / dtg, "ms_per_step": 1e3 * dtg / steps,
                                 "weight_bytes_per_step": gru_w_bytes, "achieved_gbs": gru_w_bytes * steps / dtg / 1e9,
                                 "bound": "launch latency (weights are 9.4 MB: 1.2 us at the HBM peak)"}}


def side_block(args, cfg, label, steps=6, warmup=3, kernel_timing=True):
    """A second configuration beside the main line: the same training step, a short run.  Events/s, the model-level fraction of
    the bf16 MFMA peak in algorithmic FLOPs, and (optionally) the attention kernels' launch times at this shape."""
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.metrics import CategoricalAccuracy, LogitsBucketting, MetricsSet
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    V, nl, d, L, B = cfg["vocab"], cfg["layers"], cfg["d_model"], cfg["seq_len"], cfg["batch"]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=args.dropout).to(dev).train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, optimizer=opt)
    ms = MetricsSet({"accuracy": CategoricalAccuracy(), "loss": SmoothCrossEntropyLoss(0.1, V, V - 1), "bucket": LogitsBucketting(V)})
    gen = torch.Generator().manual_seed(4321)
    ring = []
    for _ in range(2):
        xf = torch.randint(0, V - 1, (B, L + 1), generator=gen)
        ring.append((xf[:, :-1].to(torch.int32).contiguous().to(dev), xf[:, 1:].to(torch.int32).contiguous().to(dev)))
    last = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        x, y = ring[i % 2]
        last = ms(mt(x), y)
        last["loss"].backward()
        sch.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mt.check_no_leading_pads()
    value = B * L * steps / dt
    out = {"workload": f"{label}: MusicTransformer V={V} layers={nl} d_model={d} seq_len={L} bf16, "
                       f"per-GPU batch {B}, fwd+smoothed-CE+bwd+Adam/Noam, dropout {args.dropout}",
           "value": value, "unit": "events/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
           "final_loss": float(last["loss"].item()), "train_mflop_per_event": train_flops_per_event(nl, d, L, V) / 1e6,
           "model_mfma_frac": value * train_flops_per_event(nl, d, L, V) / (PEAK_BF16_TFLOPS * 1e12)}
    del mt, opt, sch, ring, last
    torch.cuda.empty_cache()
    if kernel_timing and not args.no_kernel_timing:
        kt = time_kernels(B, L, d, L)
        out["kernel_ms"] = kt
        out["attention_all_kernels"] = {"ms_per_layer": sum(kt.values()),
                                        "achieved_tflops": attn_flops_per_launch(B, L, d, 9.0) / (sum(kt.values()) * 1e-3) / 1e12}
    return out


def cfg4_block(args):
    """BASELINE configs[3] at its single-GPU share (MuMIDI V=486, 12 layers, d_model=768 = 12 heads, L=4096, per-GPU
    batch 4)"""
    return side_block(args, CFG4, "cfg4 at its single-GPU share, MuMIDI_EventSeq")


def batch8_block(args):
    """"B per GPU = 8 default, also report the largest that fits".  The main line is the large batch (64: twice the
    workgroups per launch, shorter tails); this is the same cfg2 step at per-GPU batch 8."""
    return side_block(args, dict(CFG2, batch=8), "cfg2 REMI_EventSeq at per-GPU batch 8 (SURVEY 8d default)", steps=10, warmup=3,
                      kernel_timing=False)


def batch64_block(args):
    """the cfg2 step at per-GPU batch 64, the bench default of rounds 3-5 (continuity with their lines and profiles)"""
    return side_block(args, dict(CFG2, batch=64), "cfg2 REMI_EventSeq at per-GPU batch 64 (the default of rounds 3-5)", steps=10, warmup=3,
                      kernel_timing=False)


def pmc_traffic(kernel, B, L, d):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950).
    Only reported when this run has a shape the counters were collected on (cfg2 at the per-GPU batch in the file name)."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((os.path.join(prof, f) for f in (f"r06_traffic_cfg2_b{B}.json", f"r05_traffic_cfg2_b{B}.json", f"r04_traffic_cfg2_b{B}.json", f"r03_traffic_cfg2_b{B}.json", f"r02_traffic_cfg2_b{B}.json", f"r01_traffic_cfg2_b{B}.json")
                 if os.path.exists(os.path.join(prof, f))), None)
    if path is None or (L, d) != (2048, 512):
        return {"traffic": None}
    k = json.load(open(path))["kernels"]
    sym = "rel_attn_dkv64_kernel" if (kernel == "rel_attn_dkv_kernel" and L % 128 == 0) else kernel      # the symbol this shape launches
    match = [n for n in k if sym in n]
    if not match:
        return {"traffic": None}
    heads = d // 64
    ds_half = B * heads * (L // 32) * (L // 32 + 1) // 2 * 2048          # causal half of dS, bf16 32x32 tiles
    io = B * L * d * 2                                                   # one bf16 [B,L,d] tensor
    # ALGORITHMIC bytes: the op's bf16 inputs and outputs, each once.  DESIGN bytes: the bf16 dS tiles the backward stores once (dK/dV)
    # and reads twice (dQ, dE) instead of recomputing -- traffic this design chose, not traffic the operation needs
    algo = {"rel_attn_fwd_kernel": 4 * io,                               # q,k,v in, ctx out
            "rel_attn_dkv_kernel": 6 * io,                               # q,k,v,dO in, dk,dv out
            "rel_attn_dq_lite_kernel": 2 * io,                           # k in, dq out
            "rel_attn_de_tiles_kernel": io}[kernel]                      # q in (dE out is 128 KB)
    design = 0 if kernel == "rel_attn_fwd_kernel" else ds_half
    tr = sum(k[n]["hbm_bytes_per_launch"] for n in match)
    return {"traffic": tr, "traffic_unit": f"bytes/launch (PMC, {os.path.basename(path)})",
            "algorithmic_bytes_per_launch": algo, "design_bytes_per_launch": design,
            "design_bytes_note": "the causal half of dS as bf16 32x32 tiles, written by dK/dV and read by dQ and dE (DESIGN.md 2.2)",
            "traffic_over_algorithmic": tr / algo, "traffic_over_algorithmic_plus_design": tr / (algo + design)}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kfd_gpu_count():
    """GPU agents of this node as the kernel driver lists them (CPU agents have simd_count 0); None if unreadable."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except (OSError, ValueError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh ranks as CHILD processes
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  The parent
    never touches the GPU (no HIP call before or after the spawn) and never exec()s."""
    import subprocess
    if not args.one_device:
        # count GPU agents from the KFD topology in sysfs: no HIP call in this process (torch.cuda.device_count() opens
        # the HIP runtime); when sysfs says nothing, let the child ranks fail on their own set_device
        n_dev = _kfd_gpu_count()
        if n_dev is not None and n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) in /sys/class/kfd/kfd/topology", file=sys.stderr)
            return 2
    # build libmgx.so ONCE here (hipcc only, no GPU call): N ranks importing a stale tree would otherwise all compile it
    from musicgeneration_amd import _build
    if _build.have_hipcc():
        _build.build()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if r.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (rc={r.returncode})", file=sys.stderr)
        return r.returncode or 3
    if json.loads(line).get("n_gpus") != args.gpus:
        print(f"bench.py: ranks reported n_gpus={json.loads(line).get('n_gpus')} for --gpus {args.gpus}", file=sys.stderr)
        return 4
    print(line, flush=True)
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a silent 1-GPU run labelled as N (or the reverse) is worse than no number
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.nccl_channels > 0:                         # before the communicator exists: RCCL reads them at init
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_channels)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.dp import DataParallel
    from musicgeneration_amd.metrics import CategoricalAccuracy, LogitsBucketting, MetricsSet
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam

    V, d, nl, L, B = args.vocab, args.d_model, args.layers, args.seq_len, args.batch
    from musicgeneration_amd import ops as _ops
    if args.deterministic:
        _ops.set_deterministic(True, dev)
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=args.dropout).to(dev)
    mt.train()
    # stream plan: a CU-masked side stream for the off-critical-path half of the backward and / or CUs kept free for RCCL
    plan = _ops.configure_streams(args.side_cus, args.rccl_cus, partition=not args.side_shared, device=dev,
                                  side_work={"de": 1, "dw": 2, "both": 3}[args.side_work]) \
        if (args.side_cus or args.rccl_cus) else None
    run_stream = _ops.main_stream(dev)
    dp = DataParallel(mt, groups=args.buckets or None)
    dp.measure_overlap = world > 1
    dp.overlap = not args.no_overlap
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9, grad_scale=dp.grad_scale)
    sch = CustomSchedule(d, optimizer=opt)
    ms = MetricsSet({"accuracy": CategoricalAccuracy(), "loss": SmoothCrossEntropyLoss(0.1, V, V - 1),
                     "bucket": LogitsBucketting(V)})
    # device-resident synthetic batch ring (uniform over real tokens; pad id V-1 never drawn)
    gen = torch.Generator().manual_seed(1234 + rank)
    ring = []
    for _ in range(4):
        xf = torch.randint(0, V - 1, (B, L + 1), generator=gen)
        ring.append((xf[:, :-1].to(torch.int32).contiguous().to(dev), xf[:, 1:].to(torch.int32).contiguous().to(dev)))

    def step(i):
        x, y = ring[i % len(ring)]
        m = ms(mt(x), y)
        loss = m["loss"]
        if world > 1:
            loss = loss * dp.loss_weight(ms.last_nonpad)      # the 4-byte non-pad-count all-reduce of the real train loop
        loss.backward()
        sch.step()
        opt.zero_grad()
        return m

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    with torch.cuda.stream(run_stream):
        for i in range(args.warmup):
            last = step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = step(args.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    mt.check_no_leading_pads()      # device-side no-leading-pads record of every forward above (read after the timed region)
    streams_cfg = ({"side_cus": plan.side.cus if plan.side else 0, "main_cus": plan.main.cus if plan.main else "all",
                    "reserved_for_rccl": plan.reserved} if plan is not None else "one stream, whole chip")
    if plan is not None:
        _ops.configure_streams(0, 0, device=dev)           # the side blocks below (kernel timing, cfg4, batch 8, decode) run on one stream
        plan = None
    events = float(world) * B * L * args.steps
    value = events / dt
    loss_val = float(last["loss"].item())

    out = {
        "metric": "training events/sec (whole node), REMI seq_len=2048" if args.workload == "cfg2"
                  else "training events/sec (whole node), MuMIDI seq_len=4096 (BASELINE configs[3]; NOT the headline metric)",
        "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{'cfg2 REMI_EventSeq' if args.workload == 'cfg2' else 'cfg4 MuMIDI_EventSeq'} MusicTransformer V={V} layers={nl} d_model={d} seq_len={L} "
                               f"bf16, fwd+smoothed-CE+bwd+Adam/Noam, dropout {args.dropout}",
                   "global_batch": world * B, "per_gpu_batch": B, "seq_len": L, "parallelism": f"dp{world}",
                   "final_loss": loss_val, "deterministic": _ops.deterministic(),
                   "streams": streams_cfg},
        "model_mfma_frac": value * train_flops_per_event(nl, d, L, V) / (world * PEAK_BF16_TFLOPS * 1e12),
    }
    if world > 1:
        exposed = dp.exposed_ms()
        bms = dp.bucket_ms()
        out["dp"] = dict(dp.describe(), buckets=len(dp.bucket_names()), bucket_names=dp.bucket_names(), overlap=dp.overlap,
                         rccl_cus=args.rccl_cus, nccl_channels=(args.nccl_channels or "default"), side_cus=args.side_cus,
                         allreduce_bytes_per_step=dp.bytes_reduced // max(1, args.warmup + args.steps),
                         exposed_allreduce_ms_per_step=exposed,
                         exposed_note="compute-stream stall between the end of backward and the Adam kernel (HIP events "
                                      "around the bucket waits); the rest of the all-reduce ran under backward",
                         bucket_issue_to_complete_ms=bms,
                         bucket_note="per bucket, in issue order (fc first, embedding last): mean time from the moment its gradients "
                                     "are complete (event on the compute stream) to the completion of its all-reduce (event on a side "
                                     "stream that depends on the collective; host clock with gloo).  Large while `exposed` is small: the "
                                     "collectives queue behind backward's kernels (co-residency, DESIGN 4) but still finish in time; "
                                     "large AND exposed large: RCCL itself is slow; compare with a --no-overlap run")
    if rank == 0 and not args.no_kernel_timing:
        kt = time_kernels(B, L, d, L)
        # credited (algorithmic) and executed product-units per kernel; 1 unit = B*L^2*d FLOPs
        # backward: dP, dV, dK (the dK/dV kernel, which also forms the dS everybody else reads), dq = dS K + dS_rel Er (dq_lite),
        # dE (de_tiles) = 6 credited; executed adds the recomputed S and two Q.Er^T chunks in the dK/dV kernel and the fifth
        # (half-empty) chunk product of de_tiles
        credited = {"rel_attn_fwd_kernel": 3.0, "rel_attn_dkv_kernel": 3.0, "rel_attn_dq_lite_kernel": 2.0,
                    "rel_attn_de_tiles_kernel": 1.0}
        # (the 64-key dK/dV kernel runs three Q.Er^T chunk products per pair of key tiles instead of four: 5.5 units; the 32-key one 6)
        executed = {"rel_attn_fwd_kernel": 3.0, "rel_attn_dkv_kernel": 5.5 if L % 128 == 0 else 6.0, "rel_attn_dq_lite_kernel": 2.0,
                    "rel_attn_de_tiles_kernel": 1.25}
        per_kernel = {k: {"ms": kt[k], "credited_tflops": attn_flops_per_launch(B, L, d, credited[k]) / (kt[k] * 1e-3) / 1e12,
                          "executed_tflops": attn_flops_per_launch(B, L, d, executed[k]) / (kt[k] * 1e-3) / 1e12}
                      for k in credited}
        # roofline: the single dominant KERNEL of the step by time (6 launches per step each).  Credited units are the
        # algorithmic share of the products it computes (no credit for recomputing S/P/dP or for the second Q.Er^T
        # chunk); executed units count every MFMA product it runs.  The op it belongs to (ONE C-ABI call,
        # mgx_rel_attn_bwd = pre-pass + dK/dV + dQ + dE kernels, 6 credited units) is reported beside it.
        kernel_symbol = {"rel_attn_fwd_kernel": "rel_attn_fwd_kernel<false>",
                         "rel_attn_dkv_kernel": "rel_attn_dkv64_kernel" if L % 128 == 0 else "rel_attn_dkv_kernel<true>",
                         "rel_attn_dq_lite_kernel": "rel_attn_dq_lite_kernel", "rel_attn_de_tiles_kernel": "rel_attn_de_tiles_kernel"}
        dom_k = max(credited, key=lambda k: kt[k])
        dom_ms = kt[dom_k]
        ach = attn_flops_per_launch(B, L, d, credited[dom_k]) / (dom_ms * 1e-3) / 1e12
        exe = attn_flops_per_launch(B, L, d, executed[dom_k]) / (dom_ms * 1e-3) / 1e12
        bwd_ms = (kt["attn_prepass_kernels"] + kt["rel_attn_dkv_kernel"] + kt["rel_attn_dq_lite_kernel"]
                  + kt["rel_attn_de_tiles_kernel"])
        out["roofline"] = {"bound": "mfma", "kernel": kernel_symbol[dom_k], "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": None,
                           "launch_ms": dom_ms, "credited_units": credited[dom_k], "executed_units": executed[dom_k],
                           "unit_flops": attn_flops_per_launch(B, L, d, 1.0),
                           "algorithmic_flops_per_launch": attn_flops_per_launch(B, L, d, credited[dom_k]),
                           "executed_tflops": exe, "executed_frac": exe / PEAK_BF16_TFLOPS,
                           "op": {"name": "mgx_rel_attn_bwd (pre-pass + dkv + dq_lite + de_tiles kernels)", "launch_ms": bwd_ms,
                                  "credited_units": 6.0, "executed_units": executed["rel_attn_dkv_kernel"] + 3.25,
                                  "achieved": attn_flops_per_launch(B, L, d, 6.0) / (bwd_ms * 1e-3) / 1e12,
                                  "frac": attn_flops_per_launch(B, L, d, 6.0) / (bwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}}
        out["roofline"].update(pmc_traffic(dom_k, B, L, d))
        # MFMA-busy of the attention kernels from the committed PMC passes of this shape (tools/pmc_attn.sh: SQ_VALU_MFMA_BUSY_CYCLES
        # / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)); like `traffic`, a committed measurement of this command, not of this run
        pmc_path = next((p_ for p_ in (os.path.join(ROOT, "profiles", f"r0{r_}_pmc_attn_b{B}.json") for r_ in (6, 5, 4)) if os.path.exists(p_)), "")
        if (L, d) == (2048, 512) and pmc_path:
            pk = json.load(open(pmc_path))["kernels"]
            for k in per_kernel:
                m = [v for n, v in pk.items() if kernel_symbol[k].split("<")[0] in n]
                if m:
                    per_kernel[k]["mfma_busy_pmc"] = m[0]["mfma_busy_frac"]
            tw = sum(per_kernel[k]["ms"] * per_kernel[k].get("mfma_busy_pmc", 0.0) for k in per_kernel) / sum(per_kernel[k]["ms"] for k in per_kernel)
            out["attention_mfma_busy_pmc"] = {"time_weighted": tw, "source": os.path.basename(pmc_path),
                                              "note": "cycle ratio; the chip runs these kernels at 1.5-2.0 GHz (DESIGN.md 2.6)"}
        out["kernel_ms"] = kt
        out["attention_kernels"] = per_kernel
        out["attention_all_kernels"] = {
            "ms_per_layer": sum(kt.values()),
            "achieved_tflops": attn_flops_per_launch(B, L, d, 9.0) / (sum(kt.values()) * 1e-3) / 1e12,
            "executed_tflops": attn_flops_per_launch(B, L, d, sum(executed.values())) / (sum(kt.values()) * 1e-3) / 1e12}
    if rank == 0 and world == 1 and not args.no_cfg4 and args.workload == "cfg2":
        del mt, opt, sch, ring, last
        torch.cuda.empty_cache()
        out["cfg4"] = cfg4_block(args)
        out["batch8"] = batch8_block(args)
        if B != 64:
            out["batch64"] = batch64_block(args)
    if rank == 0 and world == 1 and not args.no_decode:
        out["decode"] = decode_bench(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # rank 0 was busy for a second or two more (kernel timing, JSON): the ranks leave the process group together, so that no
        # communicator is torn down under a peer that has already exited
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
