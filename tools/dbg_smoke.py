import torch, sys
sys.path.insert(0, '.')
from musicgeneration_amd.network import MusicTransformer
from oracle import ref_cpu as R
V, d, nl, L, B = 337, 128, 2, 64, 2
p0 = R.init_params(V, d, nl, L, seed=0)
mt = MusicTransformer(embedding_dim=d, vocab_size=V, num_layer=nl, max_seq=L, dropout=0.0)
mt.load_state_dict(p0); mt = mt.to("cuda:0").train()
g = torch.Generator().manual_seed(0)
xf = torch.randint(0, V - 1, (B, L + 1), generator=g)
x = xf[:, :-1].to(torch.int32)
lg = mt(x.cuda()).float().cpu()
ref, _ = R.model_forward(p0, x, V-1)
e = (lg-ref).abs()
print("max|ref|", ref.abs().max().item(), "max err", e.max().item(), "mean err", e.mean().item(), "rel l2", ((lg-ref).norm()/ref.norm()).item())
idx = e.reshape(B, L, V).amax(-1)
print("per-position max err (b0):", [round(v,3) for v in idx[0].tolist()])
