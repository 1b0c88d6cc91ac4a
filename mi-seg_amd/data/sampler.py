"""Rank sharding of the interleaved CT+MR stream.

Reproduces what the reference gets from ``DistributedSampler(ConcatDataset([CT, MR]))`` (data/multi_modal.py:282-292,
tune.py:164 ``set_epoch``): epoch-seeded permutation of the concatenated index range, padded by wrap-around to a
multiple of the world size, rank r takes ``indices[r::world]``."""
import math

import torch


def rank_indices(n_items: int, world_size: int, rank: int, epoch: int = 0, seed: int = 0, shuffle: bool = True, drop_last: bool = False):
    if not (0 <= rank < world_size):
        raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {world_size - 1}]")
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        indices = torch.randperm(n_items, generator=g).tolist()
    else:
        indices = list(range(n_items))
    if drop_last and n_items % world_size != 0:
        num_samples = math.ceil((n_items - world_size) / world_size)
    else:
        num_samples = math.ceil(n_items / world_size)
    total = num_samples * world_size
    if not drop_last:
        pad = total - len(indices)
        if pad <= len(indices):
            indices += indices[:pad]
        else:
            indices += (indices * math.ceil(pad / len(indices)))[:pad]
    else:
        indices = indices[:total]
    return indices[rank:total:world_size]


def concat_modalities(n_ct: int, n_mr: int):
    """modality id of every item of ConcatDataset([CT, MR]) (data/utils.py:51-52: CT = 0, MR = 1)."""
    return [0] * n_ct + [1] * n_mr
