"""Parity of the experimental 64-rows-per-wave forward kernel with the generated asm sweep (tools/experiments/rel_attn_fwd64.hip,
MGX_ATTN_FWD64=3 in an experiment build) with the product's 32-row HIP kernel: ctx and lse must be bit-identical, without and with
padded keys, and on inputs that force the lazy-softmax redo late in the sweep.
    python -m musicgeneration_amd._build --variant exp --experiments          (here)
    MGX_LIB_PATH=musicgeneration_amd/libmgx_exp.so python tools/experiments/check_fwd64.py        (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from musicgeneration_amd import ops


def fwd(q, E, bits, rows64):
    os.environ["MGX_ATTN_FWD64"] = "3" if rows64 else "0"
    try:
        return ops.rel_attn_fwd(q, E, bits)
    finally:
        os.environ.pop("MGX_ATTN_FWD64", None)


dev = torch.device("cuda")
ok = True
for (B, L, d) in ((1, 128, 64), (1, 256, 64), (2, 256, 128), (1, 1024, 64), (3, 512, 192), (2, 2048, 128), (1, 640, 64)):
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 0.7).to(torch.bfloat16)
    E = (torch.randn(L, 64, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    tok = torch.zeros(B, L, dtype=torch.int32); tok[0, L - 37:] = 5
    for case in ("nopad", "pad", "spike"):
        q = qkv.clone()
        if case == "spike" and L >= 256:        # a score that jumps by > 55 nats late in the sweep: the redo path in a main body
            q[0, L - 3, :64] = 4.0; q[0, L // 2 + 5, d:d + 64] = 4.0
            q[B - 1, L - 70, :64] = 3.0; q[B - 1, 40, d:d + 64] = 5.0
        q = q.to(dev)
        bits = ops.pad_bitmap(tok.to(dev), 5) if case == "pad" else None
        c1, l1 = fwd(q, E, bits, False)
        c2, l2 = fwd(q, E, bits, True)
        torch.cuda.synchronize()
        e1, e2 = torch.equal(c1, c2), torch.equal(l1, l2)
        ok &= e1 and e2
        print(B, L, d, case, "ctx equal:", e1, " lse equal:", e2, " max|dctx|", (c1.float() - c2.float()).abs().max().item(),
              " max|dlse|", (l1 - l2).abs().max().item(), " finite:", bool(torch.isfinite(c2.float()).all()))
        if not (e1 and e2) and L <= 256:
            bad = (c1 != c2).any(-1)[0].nonzero().flatten().tolist()
            print("   rows of batch 0 whose ctx differs:", bad[:40], "..." if len(bad) > 40 else "")
            badl = (l1 != l2)[0].nonzero().tolist()
            print("   (head,row) whose lse differs:", badl[:20])
print("ALL EQUAL" if ok else "MISMATCH")
