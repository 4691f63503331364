"""Drop-in forward halves of the EDM training objective and loss:
``EDMObjectiveGenerator`` (runner/objectives/edm.py:132-212) and ``RainbowLoss`` (loss/rainbow_loss.py:38-111),
same constructor keywords, method names, return tuples and quirks; the arithmetic runs in libswingnn_hip.so
(swg_edm_objective / swg_edm_loss).  torch draws the random numbers (``torch.randn`` on the device in the reference's
order: sigmas first, then the [B,N,N] noise) so seeds keep their meaning.  The loss is differentiable w.r.t. the
prediction (reduction 'none', the trainer's call) through swg_edm_loss_bwd.
"""
import torch
import torch.nn as nn

from . import _lib as L
from .engine import _as_f32, stream_ptr


class EDMObjectiveGenerator:
    """Training objective generator in adjacency space, 'edm' preconditioning and sigma distribution only."""

    def __init__(self, precond, sigma_dist, *, other_params=None, dev=None, objective="edm"):
        if precond != "edm" or sigma_dist != "edm":
            raise NotImplementedError("swingnn_amd builds the shipped 'edm' preconditioning / sigma distribution only")
        self.objective = objective
        self.dev = dev
        self.precond = precond
        self.sigma_dist = sigma_dist
        self.other_params = other_params

    def _run(self, clean_adjs, node_flags):
        clean = _as_f32(clean_adjs)
        flags = _as_f32(node_flags)
        if clean.device.type != "cuda":
            raise RuntimeError("swingnn_amd runs on an MI355X only; there is no CPU fallback")
        B, N = clean.shape[0], clean.shape[-1]
        rnd = torch.randn(B, device=clean.device)             # edm.py:174
        raw = torch.randn_like(clean)                         # graph_utils.py:115 (inside add_sym_normal_noise)
        noisy, noise = torch.empty_like(clean), torch.empty_like(clean)
        vec = torch.empty(6, B, device=clean.device, dtype=torch.float32)
        L.check(L.lib.swg_edm_objective(B, N, clean.data_ptr(), flags.data_ptr(), rnd.data_ptr(), raw.data_ptr(),
                                        noisy.data_ptr(), noise.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(),
                                        vec[2].data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(),
                                        stream_ptr()), "swg_edm_objective")
        sigmas, weights, c_skip, c_out, c_in, c_noise = vec.unbind(0)
        return noisy, noise, sigmas, weights, (c_skip, c_out, c_in, c_noise)

    def get_input_output(self, clean_adjs, node_flags, *args, **kwargs):
        """-> net_input, net_cond, net_target, (c_skip, c_out, c_in, c_noise, sigmas, weights)   (edm.py:188-212)"""
        noisy, _, sigmas, weights, (c_skip, c_out, c_in, c_noise) = self._run(clean_adjs, node_flags)
        return noisy, sigmas, clean_adjs, (c_skip, c_out, c_in, c_noise, sigmas, weights)


class _EdmLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, flags, w, rlw, mean):
        B, N = pred.shape[0], pred.shape[-1]
        loss = torch.empty(B, device=pred.device, dtype=torch.float32)
        scratch = torch.empty(2 * B, device=pred.device, dtype=torch.float32)
        L.check(L.lib.swg_edm_loss(B, N, pred.data_ptr(), tgt.data_ptr(), flags.data_ptr(),
                                   None if w is None else w.data_ptr(), float(rlw), int(mean), loss.data_ptr(),
                                   scratch.data_ptr(), stream_ptr()), "swg_edm_loss")
        ctx.save_for_backward(pred, tgt, flags, w)
        ctx.rlw, ctx.mean = float(rlw), mean
        return loss

    @staticmethod
    def backward(ctx, gloss):
        if ctx.mean:
            raise NotImplementedError("swingnn_amd: backward of reduction='mean' (the trainer uses 'none' + .mean())")
        pred, tgt, flags, w = ctx.saved_tensors
        B, N = pred.shape[0], pred.shape[-1]
        gloss = _as_f32(gloss)
        dpred = torch.empty_like(pred)
        L.check(L.lib.swg_edm_loss_bwd(B, N, pred.data_ptr(), tgt.data_ptr(), flags.data_ptr(),
                                       None if w is None else w.data_ptr(), gloss.data_ptr(), ctx.rlw,
                                       dpred.data_ptr(), stream_ptr()), "swg_edm_loss_bwd")
        return dpred, None, None, None, None, None


class RainbowLoss(nn.Module):
    """Regression loss of the 'edm' / 'diffusion' objectives without re-weighting (rainbow_loss.py:57-111)."""

    def __init__(self, regression_loss_weight, flag_reweight, objective):
        super().__init__()
        if flag_reweight or objective not in ("edm", "diffusion"):
            raise NotImplementedError("swingnn_amd builds the shipped configuration: objective 'edm', no re-weighting")
        self.regression_loss_weight = regression_loss_weight
        self.flag_reweight = flag_reweight
        self.objective = objective

    def forward(self, net_pred, net_target, net_cond, adjs_perturbed, adjs_gt, node_flags, loss_weight=None,
                cond_val=None, flag_matching=False, reduction='mean'):
        if flag_matching:
            raise NotImplementedError("graph matching of the target is a debug path of the reference")
        return self.get_regression_loss(net_pred, net_target, net_cond, node_flags, None, loss_weight, cond_val,
                                        reduction)

    def get_regression_loss(self, predictions, targets, conditions, node_flags, reweight_coef, loss_weight,
                            condition_true_values, reduction):
        if reweight_coef is not None:
            raise NotImplementedError("re-weighting is off in the shipped configuration")
        if reduction not in ('mean', 'none', None):
            raise NotImplementedError(reduction)
        pred, tgt, flags = _as_f32(predictions), _as_f32(targets), _as_f32(node_flags)
        assert len(flags.shape) == 2
        if pred.device.type != "cuda":
            raise RuntimeError("swingnn_amd runs on an MI355X only; there is no CPU fallback")
        B, N = pred.shape[0], pred.shape[-1]
        w = None if loss_weight is None else _as_f32(loss_weight)
        return _EdmLossFn.apply(pred, tgt.reshape(pred.shape), flags, w, self.regression_loss_weight, reduction == 'mean')


# ---- molecule path: node features + (multi-channel) adjacency ------------------------------------------------------
class NodeAdjEDMObjectiveGenerator(EDMObjectiveGenerator):
    """runner/objectives/edm.py:215-276.  Draw order of the reference: sigmas, adjacency noise, node noise.  The
    adjacency planes [B,Ca,N,N] go through swg_edm_objective as B*Ca graphs (sigma repeated per plane), the node
    features through swg_nodes_objective."""

    def __init__(self, precond, sigma_dist, *, other_params=None, dev=None, objective="edm", symmetric_noise=True):
        super().__init__(precond, sigma_dist, other_params=other_params, dev=dev, objective=objective)
        if not symmetric_noise:
            raise NotImplementedError("swingnn_amd: symmetric_noise=True (utils/learning_utils.py:48)")
        self.symmetric_noise = symmetric_noise

    def _draw(self, clean, cx):
        """The reference's draws in its order (overridable so that parity tests can inject CPU-generator noise)."""
        rnd = torch.randn(clean.shape[0], device=clean.device)   # edm.py:174
        raw = torch.randn_like(clean)                              # graph_utils.py:117
        raw_x = torch.randn_like(cx)                               # edm.py:244
        return rnd, raw, raw_x

    def get_input_output(self, clean_adjs, clean_x=None, node_flags=None, *args, **kwargs):
        """-> net_input_a, net_input_x, net_cond, net_target_a, net_target_x,
              (c_skip, c_out, c_in, c_noise, sigmas, weights)   (edm.py:251-276)"""
        clean = _as_f32(clean_adjs)
        cx = _as_f32(clean_x)
        flags = _as_f32(node_flags)
        if clean.device.type != "cuda":
            raise RuntimeError("swingnn_amd runs on an MI355X only; there is no CPU fallback")
        B, N = clean.shape[0], clean.shape[-1]
        Ca = 1 if clean.dim() == 3 else clean.shape[1]
        Cn = 1 if cx.dim() == 2 else cx.shape[-1]
        rnd, raw, raw_x = self._draw(clean, cx)
        rnd_p = rnd.repeat_interleave(Ca).contiguous() if Ca > 1 else rnd
        flags_p = flags.repeat_interleave(Ca, dim=0).contiguous() if Ca > 1 else flags
        noisy = torch.empty_like(clean)
        vec = torch.empty(6, B * Ca, device=clean.device, dtype=torch.float32)
        L.check(L.lib.swg_edm_objective(B * Ca, N, clean.data_ptr(), flags_p.data_ptr(), rnd_p.data_ptr(), raw.data_ptr(),
                                        noisy.data_ptr(), None, vec[0].data_ptr(), vec[1].data_ptr(),
                                        vec[2].data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), vec[5].data_ptr(),
                                        stream_ptr()), "swg_edm_objective")
        sigmas, weights, c_skip, c_out, c_in, c_noise = (v[::Ca].contiguous() for v in vec.unbind(0))
        noisy_x = torch.empty_like(cx)
        L.check(L.lib.swg_nodes_objective(B, N, Cn, cx.data_ptr(), raw_x.data_ptr(), flags.data_ptr(), sigmas.data_ptr(),
                                          noisy_x.data_ptr(), stream_ptr()), "swg_nodes_objective")
        return noisy, noisy_x, sigmas, clean_adjs, clean_x, (c_skip, c_out, c_in, c_noise, sigmas, weights)


class _NodeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, flags, w, lw):
        B, N = flags.shape
        Cn = pred.numel() // (B * N)
        loss = torch.empty(B, device=pred.device, dtype=torch.float32)
        L.check(L.lib.swg_nodes_loss(B, N, Cn, pred.data_ptr(), tgt.data_ptr(), flags.data_ptr(),
                                     None if w is None else w.data_ptr(), float(lw), loss.data_ptr(), stream_ptr()),
                "swg_nodes_loss")
        ctx.save_for_backward(pred, tgt, flags, w)
        ctx.lw, ctx.Cn = float(lw), Cn
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, tgt, flags, w = ctx.saved_tensors
        B, N = flags.shape
        dpred = torch.empty_like(pred)
        gloss = _as_f32(gloss)
        L.check(L.lib.swg_nodes_loss_bwd(B, N, ctx.Cn, pred.data_ptr(), tgt.data_ptr(), flags.data_ptr(),
                                         None if w is None else w.data_ptr(), gloss.data_ptr(), ctx.lw,
                                         dpred.data_ptr(), stream_ptr()), "swg_nodes_loss_bwd")
        return dpred, None, None, None, None


class NodeAdjRainbowLoss(nn.Module):
    """loss/rainbow_loss.py:155-251, objective 'edm', reduction 'none' (the trainer's call,
    runner/trainer/trainer_node_adj.py:106-119) -> (loss_adj [B], loss_node [B])."""

    def __init__(self, edge_loss_weight, node_loss_weight, objective, flag_reweight=False):
        super().__init__()
        if flag_reweight or objective not in ("edm", "diffusion"):
            raise NotImplementedError("swingnn_amd builds the shipped configuration: objective 'edm', no re-weighting")
        self.edge_loss_weight, self.node_loss_weight = edge_loss_weight, node_loss_weight
        self.flag_reweight, self.objective = flag_reweight, objective

    def forward(self, net_pred_a, net_pred_x, net_target_a, net_target_x, net_cond, adjs_perturbed, adjs_gt,
                x_perturbed, x_gt, node_flags, loss_weight=None, cond_val=None, flag_matching=False, reduction='mean'):
        if flag_matching:
            raise ValueError("Graph matching is not supported for node-adj loss!")
        return self.get_regression_loss(net_pred_a, net_pred_x, net_target_a, net_target_x, net_cond, node_flags, None,
                                        loss_weight, cond_val, reduction)

    def get_regression_loss(self, pred_adj, pred_node, target_adj, target_node, net_cond, node_flags, reweight_coef,
                            loss_weight, condition_true_values, reduction):
        if reweight_coef is not None:
            raise NotImplementedError("re-weighting is off in the shipped configuration")
        if reduction not in ('none', None):
            raise NotImplementedError("swingnn_amd.NodeAdjRainbowLoss: reduction='none' (the trainer's call)")
        pa, ta = _as_f32(pred_adj), _as_f32(target_adj)
        pn, tn = _as_f32(pred_node), _as_f32(target_node)
        flags = _as_f32(node_flags)
        assert len(flags.shape) == 2
        if pa.device.type != "cuda":
            raise RuntimeError("swingnn_amd runs on an MI355X only; there is no CPU fallback")
        B, N = flags.shape
        Ca = 1 if pa.dim() == 3 else pa.shape[1]
        w = None if loss_weight is None else _as_f32(loss_weight.view(-1))
        flags_p = flags.repeat_interleave(Ca, dim=0).contiguous() if Ca > 1 else flags
        w_p = None if w is None else (w.repeat_interleave(Ca).contiguous() if Ca > 1 else w)
        per_plane = _EdmLossFn.apply(pa.reshape(B * Ca, N, N), ta.reshape(B * Ca, N, N), flags_p, w_p, 1.0, False)
        loss_adj = per_plane.view(B, Ca).sum(dim=1) / Ca * self.edge_loss_weight      # :238-242
        loss_node = _NodeLossFn.apply(pn, tn.reshape(pn.shape), flags, w, self.node_loss_weight)   # :244-248
        return loss_adj, loss_node
