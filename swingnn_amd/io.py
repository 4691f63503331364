"""Host-side data formats either side of the hot path: checkpoint loading, dataset tensorisation and
the post-sampling quantisation, with the reference's semantics.  Pure host code (numpy / torch CPU ops on data that the
reference also handles on the host); nothing here is on the timed path.
"""
import numpy as np
import torch


def load_model(ckp_data, model, weight_keyword):
    """utils/sampling_utils.py:87-113: strict load of ``ckp_data[weight_keyword]`` into ``model``; a ``module.`` prefix
    on either side (DDP / DataParallel checkpoints, runner/trainer/trainer_utils.py:164-181) is reconciled.  Same
    contract as the reference (strict load, KeyError-free for the two wrappings), written on canonical names: both key
    sets are reduced to their un-prefixed form, compared, and the checkpoint tensors are re-keyed to the model's
    spelling."""
    if weight_keyword not in ckp_data:
        # the reference asserts here (sampling_utils.py:92); same exception type for callers that catch it
        raise AssertionError(f"checkpoint has no entry {weight_keyword!r}; entries: {sorted(ckp_data.keys())}")
    weights = ckp_data[weight_keyword]
    prefix = "module."

    def canonical(name):
        return name[len(prefix):] if name.startswith(prefix) else name

    by_canonical = {canonical(k): v for k, v in weights.items()}
    if len(by_canonical) != len(weights):
        raise ValueError("checkpoint holds both a wrapped and an unwrapped copy of some tensor")
    wanted = list(model.state_dict().keys())
    missing = [k for k in wanted if canonical(k) not in by_canonical]
    extra = set(by_canonical) - {canonical(k) for k in wanted}
    if missing or extra:
        raise RuntimeError(f"state-dict mismatch: missing {missing[:5]}, unexpected {sorted(extra)[:5]}")
    model.load_state_dict({k: by_canonical[canonical(k)] for k in wanted}, strict=True)
    return model


def pad_adjs(adj, node_number):
    """utils/graph_utils.py:97-110."""
    a = np.asarray(adj)
    n = a.shape[-1]
    if n == node_number:
        return a
    if n > node_number:
        raise ValueError(f'ori_len {n} > node_number {node_number}')
    out = np.zeros((node_number, node_number), dtype=a.dtype)
    out[:n, :n] = a
    return out


def adjs_to_dataset_tensors(adjs, max_node_num, ddpm_scale=True):
    """utils/dataloader.py:219-247 (``_nx_graphs_to_dataset`` after ``nx.to_numpy_matrix``): list of 0/1 adjacency
    matrices -> (adjs [B,N,N] float32, dummy node features [B,N,1], node_flags [B,N]).  ``node_flags`` marks rows
    with at least one edge BEFORE the rescale (:238, isolated nodes count as padding); with ``ddpm_scale`` the entries
    of valid rows / columns go {0,1} -> {-1,+1} and padding stays 0 (utils/attribute_code.py, 2 attribute types)."""
    a = torch.tensor(np.asarray([pad_adjs(x, max_node_num) for x in adjs]), dtype=torch.float32)
    node_feat = torch.zeros(a.shape[:2], dtype=torch.float32).unsqueeze(-1)
    flags = a.sum(-1).gt(1e-5).to(torch.float32)
    if ddpm_scale:
        keep = flags[:, :, None] * flags[:, None, :]
        a = (a * 2.0 - 1.0) * keep
    return a, node_feat, flags


def quantise_samples(final_samples, node_flags):
    """Post-sampling quantisation of runner/sampler/sampler.py:131-135: clamp to [-1, 1], 2-type ddpm -> int
    (``x > 0``), masked by the node flags (utils/attribute_code.py:133-159)."""
    x = final_samples.clamp(min=-1.0, max=1.0)
    keep = node_flags.to(torch.bool)
    keep = keep[..., :, None] & keep[..., None, :]
    return ((x > 0) & keep).to(final_samples.dtype)


def tile_batch(data_tuple, dataloader):
    """Batch initialisation of the train loop, runner/trainer/trainer.py:29-47: a 2-tuple gets its node flags from the
    adjacency (:31); a batch smaller than ``dataloader.batch_size`` that divides it is stacked up to the full batch
    size, and the stacked tensors are cached on the dataloader (``dataloader.repeated_data``) and returned for every
    later short batch, exactly as the reference does (small datasets: ego-small, community-small)."""
    if len(data_tuple) == 2:
        adjs_gt, x_gt = data_tuple
        node_flags = adjs_gt.abs().sum(-1).gt(1e-5).to(dtype=torch.float32)
    elif len(data_tuple) == 3:
        adjs_gt, x_gt, node_flags = data_tuple
    else:
        raise NotImplementedError
    if len(adjs_gt) < dataloader.batch_size and dataloader.batch_size % len(adjs_gt) == 0:
        cached = getattr(dataloader, "repeated_data", None)
        if cached is None:
            rep = dataloader.batch_size // len(adjs_gt)
            cached = [adjs_gt.repeat(rep, 1, 1), x_gt.repeat(rep, 1, 1), node_flags.repeat(rep, 1)]
            dataloader.repeated_data = cached
        adjs_gt, x_gt, node_flags = cached
    return adjs_gt, x_gt, node_flags
