import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from musicgeneration_amd.network import MusicTransformer
from musicgeneration_amd.criterion import SmoothCrossEntropyLoss
g = dict(np.load('tests/golden/g2_model.npz'))
sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('p.')}
V = sd['fc.weight'].shape[0]
mt = MusicTransformer(embedding_dim=128, vocab_size=V, num_layer=2, max_seq=32, dropout=0.0)
mt.load_state_dict(sd); mt = mt.cuda().train()
x = torch.from_numpy(g['x']).cuda(); y = torch.from_numpy(g['y']).cuda()
loss = SmoothCrossEntropyLoss(0.1, V, V-1)(mt(x), y); loss.backward(); torch.cuda.synchronize()
got = mt.Decoder.embedding.weight.grad.cpu(); ref = torch.from_numpy(g['g.Decoder.embedding.weight'])
rn = ref.norm(dim=1); gn = got.norm(dim=1)
rows = (rn > 0).nonzero().flatten().tolist()
xs = g['x']
for r in rows[:100]:
    c = (got[r] @ ref[r] / (gn[r]*rn[r] + 1e-30)).item()
    cnt = int((xs == r).sum())
    pos = np.argwhere(xs == r).tolist()
    if c < 0.995 or abs(gn[r]/rn[r]-1) > 0.02: print("row", r, "count", cnt, "cos", round(c,4), "norm ratio", round((gn[r]/rn[r]).item(),4), pos)
print("rows with grad in got but not ref:", ((gn>0)&(rn==0)).sum().item(), " total cos", (got.flatten()@ref.flatten()/(got.norm()*ref.norm())).item())
