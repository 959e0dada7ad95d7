"""cfg4-shaped model (12 layers, d=768, L=4096, batch 4) on the learnable task next = prev + 1 (mod V-1): loss per step for a few
Noam warm-up lengths (GPU box).  Picks the schedule of tests/test_gpu_fullsize.py::test_cfg4_shaped_model_step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd.criterion import CustomSchedule, SmoothCrossEntropyLoss
from musicgeneration_amd.network import MusicTransformer
from musicgeneration_amd.optim import FusedAdam
V, d, L, B = 486, 768, 4096, 4
for warm in [int(a) for a in sys.argv[1:]] or [60, 200]:
    torch.manual_seed(0)
    mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=12, max_seq=L, dropout=0.0).cuda().train()
    opt = FusedAdam(mt, lr=0.0, betas=(0.9, 0.98), eps=1e-9)
    sch = CustomSchedule(d, warmup_steps=warm, optimizer=opt)
    lossf = SmoothCrossEntropyLoss(0.1, V, V - 1)
    g = torch.Generator().manual_seed(5)
    losses = []
    for it in range(16):
        start = torch.randint(0, V - 1, (B, 1), generator=g)
        seq = ((start + torch.arange(L + 1)[None, :]) % (V - 1)).cuda()
        x, y = seq[:, :-1].to(torch.int32).contiguous(), seq[:, 1:].to(torch.int32).contiguous()
        loss = lossf(mt(x), y)
        loss.backward()
        sch.step()
        opt.zero_grad()
        losses.append(round(loss.item(), 3))
    print("warmup", warm, losses, flush=True)
    del mt, opt
    torch.cuda.empty_cache()
