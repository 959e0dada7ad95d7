"""Worker of tests/test_gpu_dp.py: trains the real kernel-backed model for a few steps, either as ONE process on the
full batch or as a rank of a 2-process data-parallel job (gloo all-reduce of the flat gradient buffer, both ranks on
cuda:0), and writes the losses and a parameter checksum to a JSON file."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_path):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    rccl1 = os.environ.get("MGX_TEST_RCCL1") == "1"    # one rank, backend "nccl" (= RCCL), every collective forced
    # the co-residency mitigations of DESIGN.md section 4 (bench.py --rccl-cus / --side-cus / --buckets / --nccl-channels)
    rccl_cus, side_cus = int(os.environ.get("MGX_TEST_RCCL_CUS", "0")), int(os.environ.get("MGX_TEST_SIDE_CUS", "0"))
    groups = int(os.environ.get("MGX_TEST_BUCKETS", "0")) or None
    if os.environ.get("MGX_TEST_NCCL_CHANNELS"):
        os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = os.environ["MGX_TEST_NCCL_CHANNELS"]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif rccl1:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{os.environ['MGX_TEST_PORT']}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
    from musicgeneration_amd.dp import DataParallel
    from musicgeneration_amd.network import MusicTransformer
    from musicgeneration_amd.optim import FusedAdam
    V, d, nl, L, B = 90, 128, 2, 128, 8
    torch.manual_seed(100 + rank)                      # ranks start DIFFERENT: the broadcast must fix that
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0).cuda().train()
    if world == 1:
        torch.manual_seed(100)
        mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0).cuda().train()
    from musicgeneration_amd import ops
    plan = ops.configure_streams(side_cus, rccl_cus) if (side_cus or rccl_cus) else None
    dp = DataParallel(mt, force_collectives=rccl1, groups=groups)
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9, grad_scale=dp.grad_scale)
    sch = CustomSchedule(d, warmup_steps=20, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    g = torch.Generator().manual_seed(9)
    losses = []
    with torch.cuda.stream(ops.main_stream()):     # the plan's masked main stream, or the current stream
        for it in range(6):
            xf = torch.randint(0, V - 1, (B, L + 1), generator=g)          # the GLOBAL batch, pad-free
            if world > 1:
                xf = xf[rank * (B // world):(rank + 1) * (B // world)]
            x, y = xf[:, :-1].to(torch.int32).cuda(), xf[:, 1:].to(torch.int32).cuda()
            loss = lossf(mt(x), y)
            if rccl1:                                                       # the loss-weight all-reduce too (== 1 here)
                loss = loss * dp.loss_weight((y != V - 1).sum())
            loss.backward()
            if it == 0:                                                     # the (all-reduced) gradient of the first step, before Adam
                dp.wait_all()
                torch.cuda.synchronize()
                gr = mt.store().grad.double() * dp.grad_scale
                first = {"grad_sum": float(gr.sum()), "grad_abs": float(gr.abs().sum()), "grad_l2": float(gr.norm())}
            sch.step()                                                      # waits for the bucket all-reduces, then Adam
            opt.zero_grad()
            losses.append(float(dp.all_reduce_scalar_mean(loss.detach())))
    st = mt.store()
    import hashlib
    res = {"losses": losses, "param_sum": float(st.param.double().sum()), "param_abs": float(st.param.double().abs().sum()),
           "param_hash": hashlib.sha256(st.param.detach().cpu().numpy().tobytes()).hexdigest(), "deterministic": ops.deterministic(),
           "buckets": len(st.buckets), "bytes_reduced": dp.bytes_reduced, "describe": dp.describe(), **first,
           "allreduce_units": dp.bucket_names(),
           "streams": None if plan is None else {"side": plan.side.cus if plan.side else 0, "main": plan.main.cus if plan.main else None,
                                                 "reserved": plan.reserved},
           "nccl_channels": os.environ.get("NCCL_MAX_NCHANNELS")}
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    if world > 1 or rccl1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
