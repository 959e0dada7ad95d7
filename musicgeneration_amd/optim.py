"""Fused Adam over the model's flat buffers (replaces torch.optim.Adam of train.py:143).

One kernel launch updates every parameter, both moments and the bf16 shadow.  ``state_dict`` /
``load_state_dict`` use torch.optim.Adam's layout (per-parameter ``step``, ``exp_avg``,
``exp_avg_sq``) so the reference's checkpoints (train.py:201-207) round-trip."""
from __future__ import annotations

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=0.0, betas=(0.9, 0.98), eps=1e-9, grad_scale: float = 1.0):
        self.model = model
        store = model.store()
        self.store = store
        params = [store.params[n] for n in store.names]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.m = torch.zeros_like(store.param)
        self.v = torch.zeros_like(store.param)
        self._t = 0
        self.grad_scale = grad_scale

    def zero_grad(self, set_to_none: bool = False):
        self.store.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        self._t += 1
        dp = getattr(self.model, "_dp", None)
        ops.join_side_stream(self.store.param.device)      # weight gradients / dE issued on the CU-masked side stream (ops.configure_streams)
        if dp is not None:
            dp.wait_all()                 # gradients of every bucket reduced (sum) before the update
        ops.adam_step(self.store.param, self.store.grad, self.m, self.v, self.store.shadow, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], self._t, self.grad_scale)

    # ---- torch.optim.Adam-compatible checkpoint format --------------------------------------------
    # torch (and the reference's ``Adam(mt.parameters())``, train.py:143) number the per-parameter state by
    # ``model.parameters()`` order (Wq.weight, Wq.bias, Wk.weight, ...), NOT by the flat store's order
    # (Wq.weight, Wk.weight, Wv.weight, Wq.bias, ...): emit and read that numbering, so a checkpoint written here loads
    # into ``torch.optim.Adam(mt.parameters())`` and vice versa.  ``param_names`` (extra key) records the order used.
    def _torch_order(self):
        return [n for n, _ in self.model.named_parameters()]

    def state_dict(self):
        st = self.store
        names = self._torch_order()
        state = {}
        for i, n in enumerate(names):
            state[i] = {"step": torch.tensor(float(self._t)), "exp_avg": st.view(self.m, n).clone(),
                        "exp_avg_sq": st.view(self.v, n).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": g["betas"], "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "param_names": names}

    def load_state_dict(self, sd):
        st = self.store
        names = sd.get("param_names")
        if names is None:
            names = self._torch_order()        # a torch.optim.Adam checkpoint
        if len(names) != len(st.names) or set(names) != set(st.names):
            raise ValueError("optimizer state_dict does not match this model's parameters")
        for i, n in enumerate(names):
            s = sd["state"].get(i)
            if s is None:
                continue
            k = st.offsets[n][1]
            if s["exp_avg"].numel() != k:
                raise ValueError(f"optimizer state {i} has {s['exp_avg'].numel()} elements, parameter {n} has {k}")
            st.view(self.m, n).copy_(s["exp_avg"].reshape(st.shapes[n]))
            st.view(self.v, n).copy_(s["exp_avg_sq"].reshape(st.shapes[n]))
            self._t = int(float(s["step"]))
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
