import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import ref_cpu as R
from musicgeneration_amd.network import MusicTransformer
from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
def cos(a,b): a,b=a.float().flatten(),b.float().flatten(); return (a@b/(a.norm()*b.norm()+1e-30)).item()
def rel(a,b): a,b=a.float().flatten(),b.float().flatten(); return ((a-b).norm()/(b.norm()+1e-30)).item()
g = dict(np.load("/root/repo/tests/golden/g12_model_d128_tamed.npz"))
V,d,nl,L,B = (int(v) for v in g["shape"])
p = R.init_params(V,d,nl,L,seed=int(g["seed"]))
for k in p:
    if k.endswith("embedding.weight") or k.endswith("rga.E"): p[k] = p[k]*float(g["scale"])
mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0); mt.load_state_dict(p); mt=mt.cuda().train()
x = torch.from_numpy(g["x"]).cuda()
lg = mt(x[:,:-1].to(torch.int32)); loss = SmoothCrossEntropyLoss(0.1,V,V-1)(lg, x[:,1:].to(torch.int32)); loss.backward()
print("G12 loss", loss.item(), float(g["loss"]))
w = sorted(((cos(q.grad.cpu(), torch.from_numpy(g["g."+n])), rel(q.grad.cpu(), torch.from_numpy(g["g."+n])), n) for n,q in mt.named_parameters()))
for c,r,n in w[:8]: print(f"  {c:.5f} {r:.4f} {n}")
# G9b movement cosines
g9 = dict(np.load("/root/repo/tests/golden/g9b_optim_d128.npz"))
from musicgeneration_amd.criterion import CustomSchedule
from musicgeneration_amd.optim import FusedAdam
V,d,nl,L,B = (int(v) for v in g9["shape"])
p0 = R.init_params(V,d,nl,L,seed=int(g9["seed"]))
mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0); mt.load_state_dict(p0); mt=mt.cuda().train()
opt = FusedAdam(mt, lr=0.0, betas=(0.9,0.98), eps=1e-9); sch = CustomSchedule(d, optimizer=opt); lossf = SmoothCrossEntropyLoss(0.1,V,V-1)
opt.zero_grad()
for it in range(6):
    xf = torch.from_numpy(g9["xs"][it]).cuda()
    l = lossf(mt(xf[:,:-1].to(torch.int32)), xf[:,1:].to(torch.int32))/2; l.backward()
    if (it+1)%2==0: sch.step(); opt.zero_grad()
sd = {k:v.detach().float().cpu() for k,v in mt.state_dict().items()}
for k in [n[len("delta."):] for n in g9 if n.startswith("delta.")]:
    print("G9b delta cos", k, cos(sd[k]-p0[k], torch.from_numpy(g9["delta."+k])))
# layers stand-alone test: cosines vs fp32 oracle and vs bf16-emulating oracle at taming 0.3
from musicgeneration_amd.layers import Encoder
from musicgeneration_amd import utils
V, d, nl, L, B = 90, 128, 2, 64, 2; pad = V-1
for scale in (0.3, 0.15):
    p = R.init_params(V, d, nl, L, seed=2)
    for k in p:
        if k.endswith("embedding.weight") or k.endswith("rga.E"): p[k] = p[k]*scale
    enc = Encoder(num_layers=nl, d_model=d, input_vocab_size=V, rate=0.0, max_len=L)
    enc.load_state_dict({k[len("Decoder."):]: v for k,v in p.items() if k.startswith("Decoder.")}, strict=True); enc = enc.cuda().train()
    gg = torch.Generator().manual_seed(3); tok = torch.randint(0, V-1, (B,L), generator=gg); tok[1,-7:] = pad
    _,_,lam = utils.get_masked_with_pad_tensor(L, tok, tok, pad)
    hid, ws = enc(tok.cuda(), lam.cuda())
    wsum = torch.linspace(-1,1,hid.numel()).reshape(hid.shape)
    (hid*wsum.cuda()).sum().backward()
    res = {}
    for emu in (False, True):
        R.EMULATE_BF16 = emu
        pr = {k: v.clone().requires_grad_(True) for k,v in p.items()}
        ref, wref = R.decoder_stack(pr, tok, R.look_ahead_mask(tok, pad))
        (ref*wsum).sum().backward()
        R.EMULATE_BF16 = False
        res[emu] = sorted((cos(q.grad.cpu(), pr["Decoder."+n].grad), n) for n,q in enc.named_parameters() if not n.endswith("Wk.bias"))
    print("layers scale", scale, "worst vs fp32:", [(round(c,4), n) for c,n in res[False][:4]], " worst vs bf16-emu:", [(round(c,4), n) for c,n in res[True][:4]])
