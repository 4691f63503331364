"""Optimizer tail of the train step on the device:

    nn.utils.clip_grad_norm_(model.parameters(), max_norm=10.0, norm_type=2)    # runner/trainer/trainer.py:136
    optimizer.step()                                                            # Adam, utils/learning_utils.py:245-248
    [ema.update() for ema in ema_helper]                                        # ema_pytorch.EMA x 5, :253-274

as two multi-tensor HIP launches (swg_grad_sqnorm + swg_adam_ema_step) instead of ~12 parameter-sized passes.
``FusedAdamEMA`` is a ``torch.optim.Optimizer`` (``param_groups`` / ``zero_grad`` / lr schedulers work as usual);
with ``max_norm=None`` and no EMA it is a plain Adam.  Parameters stay ordinary nn.Parameters; the moments and the
EMA copies live in flat fp32 buffers owned by the optimizer.

EMA semantics follow ema_pytorch's ``EMA(beta, update_every=1, update_after_step=0, inv_gamma=1, power=1)`` as the
reference constructs it: updates 1 and 2 copy the parameters (step <= update_after_step, then the `initted` copy),
update number n >= 3 uses decay = clamp(1 - (1 + (n - 1) / inv_gamma)^-power, 0, beta).  ema_pytorch is not installed in the build container, so this rule
is restated from its documentation and is PARITY-UNPINNED; the Adam and clipping arithmetic is checked
against torch.optim.Adam / clip_grad_norm_ in tests/test_gpu_train.py.
"""
import ctypes as C

import torch

from . import _lib as L
from .engine import stream_ptr


class FusedAdamEMA(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=None, ema_betas=(),
                 ema_inv_gamma=1.0, ema_power=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamEMA: one parameter group (the reference optimises model.parameters())")
        self.max_norm = max_norm
        self.ema_betas = tuple(sorted(float(b) for b in ema_betas))
        if len(self.ema_betas) > 8:
            raise NotImplementedError("at most 8 EMA copies")
        self.ema_inv_gamma, self.ema_power = float(ema_inv_gamma), float(ema_power)
        self.params = [p for p in self.param_groups[0]["params"]]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("swingnn_amd runs on an MI355X only; there is no CPU fallback")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise NotImplementedError("FusedAdamEMA: contiguous fp32 parameters on one device")
        chunk = int(L.lib.swg_optim_chunk())
        numel, flat_off, ct, co = [], [], [], []
        off = 0
        for i, p in enumerate(self.params):
            n = p.numel()
            numel.append(n)
            flat_off.append(off)
            for c in range(0, n, chunk):
                ct.append(i)
                co.append(c)
            off += (n + 63) // 64 * 64
        self.total = off
        i32 = dict(dtype=torch.int32, device=dev)
        self._numel = torch.tensor(numel, **i32)
        self._flat_off = torch.tensor(flat_off, dtype=torch.int64, device=dev)
        self._ct = torch.tensor(ct, **i32)
        self._co = torch.tensor(co, **i32)
        self._offs = flat_off
        self.n_chunks = len(ct)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.ema = torch.zeros((max(len(self.ema_betas), 1), off), dtype=torch.float32, device=dev)
        self._p_key = tuple(p.data_ptr() for p in self.params)
        self._p_ptrs = torch.tensor(self._p_key, dtype=torch.int64, device=dev)
        self._g_ptrs = torch.zeros(len(self.params), dtype=torch.int64, device=dev)
        self._g_key = None
        self._sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.ema_updates = 0

    # ---- EMA bookkeeping (host scalars only) ----
    def _ema_decays(self):
        n = self.ema_updates + 1  # this is update number n
        if n <= 2:
            return [0.0] * len(self.ema_betas)  # the first two updates copy the parameters
        epoch = float(n - 1)
        value = 1.0 - (1.0 + epoch / self.ema_inv_gamma) ** (-self.ema_power)
        return [min(max(value, 0.0), b) for b in self.ema_betas]

    def ema_parameters(self, k):
        """Views of EMA copy k (sorted by beta) shaped like the parameters, in parameter order."""
        return [self.ema[k, o:o + p.numel()].view(p.shape) for o, p in zip(self._offs, self.params)]

    def ema_state_dict(self, k, model):
        """EMA copy k (sorted by beta) as a state dict of ``model`` -- what the reference checkpoints under
        ``model_ema_beta_{beta:.4f}`` (``ema.ema_model.state_dict()``, runner/trainer/trainer_utils.py:176-179):
        parameters come from the EMA buffer, registered buffers (relative_position_index, attn_mask) are copied from
        the model.  Tensors are clones, safe to ``torch.save``."""
        by_id = {id(p): i for i, p in enumerate(self.params)}
        ema = self.ema_parameters(k)
        out = {}
        named = dict(model.named_parameters(remove_duplicate=False))
        for name, t in model.state_dict().items():
            p = named.get(name)
            if p is not None and id(p) in by_id:
                out[name] = ema[by_id[id(p)]].detach().clone()
            else:
                out[name] = t.detach().clone()
        return out

    def ema_checkpoint_entries(self, model):
        """``{'model_ema_beta_{beta:.4f}': state_dict}`` for every EMA copy: the entries get_ckpt_data adds
        (trainer_utils.py:176-179) and eval.py:14-36 selects with ``--use_ema``."""
        return {"model_ema_beta_{:.4f}".format(b): self.ema_state_dict(k, model) for k, b in enumerate(self.ema_betas)}

    # ---- checkpoint / resume of the optimizer itself (moments, step counters, EMA copies) ----
    def state_dict(self):
        sd = super().state_dict()  # param_groups (lr, betas, ...); per-parameter state is kept flat, below
        sd["fused"] = {
            "m": self.m.detach().clone(), "v": self.v.detach().clone(), "ema": self.ema.detach().clone(),
            "step_count": self.step_count, "ema_updates": self.ema_updates, "ema_betas": list(self.ema_betas),
            "numel": [p.numel() for p in self.params],
        }
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        if fused is None:
            raise KeyError("FusedAdamEMA.load_state_dict: no 'fused' entry (not a FusedAdamEMA checkpoint)")
        if list(fused["numel"]) != [p.numel() for p in self.params]:
            raise ValueError("FusedAdamEMA.load_state_dict: parameter list differs from the checkpoint's")
        if [float(b) for b in fused["ema_betas"]] != list(self.ema_betas):
            raise ValueError("FusedAdamEMA.load_state_dict: EMA betas differ from the checkpoint's")
        super().load_state_dict(state_dict)
        self.m.copy_(fused["m"])
        self.v.copy_(fused["v"])
        self.ema.copy_(fused["ema"])
        self.step_count = int(fused["step_count"])
        self.ema_updates = int(fused["ema_updates"])

    def grad_norm(self):
        """Total L2 gradient norm measured by the last step() (before clipping), as a 0-dim device tensor."""
        return self._sqnorm.sqrt().squeeze(0)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        grads = []
        for p in self.params:
            g = p.grad
            if g is None:
                raise RuntimeError("FusedAdamEMA.step: every parameter needs a gradient (the denoiser uses all of them)")
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.to(torch.float32).contiguous()
            grads.append(g)
        key = tuple(g.data_ptr() for g in grads)
        if key != self._g_key:  # zero_grad(set_to_none=True) gives new gradient tensors every step
            self._g_ptrs.copy_(torch.tensor(key, dtype=torch.int64), non_blocking=False)
            self._g_key = key
        pkey = tuple(p.data_ptr() for p in self.params)
        if pkey != self._p_key:  # parameter storage moved (model.to(), p.data = ..., load_state_dict(assign=True))
            self._p_ptrs.copy_(torch.tensor(pkey, dtype=torch.int64), non_blocking=False)
            self._p_key = pkey
        grp = self.param_groups[0]
        st = stream_ptr()
        sq = None
        if self.max_norm is not None:
            L.check(L.lib.swg_grad_sqnorm(self._g_ptrs.data_ptr(), self._numel.data_ptr(), self._ct.data_ptr(),
                                          self._co.data_ptr(), self.n_chunks, self._sqnorm.data_ptr(), st),
                    "swg_grad_sqnorm")
            sq = self._sqnorm.data_ptr()
        self.step_count += 1
        decays = self._ema_decays()
        n_ema = len(self.ema_betas)
        dec = (C.c_float * 8)(*(decays + [0.0] * (8 - n_ema)))
        b1, b2 = grp["betas"]
        L.check(L.lib.swg_adam_ema_step(self._p_ptrs.data_ptr(), self._g_ptrs.data_ptr(), self._numel.data_ptr(),
                                        self._flat_off.data_ptr(), self._ct.data_ptr(), self._co.data_ptr(),
                                        self.n_chunks, self.m.data_ptr(), self.v.data_ptr(), self.ema.data_ptr(),
                                        self.total, n_ema, dec, sq, float(self.max_norm or 0.0), float(grp["lr"]),
                                        float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                                        self.step_count, st), "swg_adam_ema_step")
        if n_ema:
            self.ema_updates += 1
        self._bump_versions()
        return None

    def _bump_versions(self):
        """Parameters were updated through raw pointers; bump their version counters so derived weight caches
        (engine.NetCache) rebuild: one foreach call on zero-size views (no element is touched, the counters move)."""
        torch._foreach_add_([p.view(-1)[:0] for p in self.params], 0.0)
