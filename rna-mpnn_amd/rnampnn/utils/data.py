"""Host-side input contract of the hot path: the batch layout the reference's collate produces and
``separate`` (``rnampnn/utils/data.py:110-142, 594-604``).  Dataset / PDB parsing / plotting of that
file are host I/O outside the hot path and are not mirrored."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import torch
from torch.nn.utils.rnn import pad_sequence

Item = Dict[str, Union[str, torch.Tensor]]


def featurize(batch: Sequence[Item]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[str]]:
    """Collate {'sequence' (L,4) one-hot, 'coordinates' (L,7,3), 'id'} items to the batch max length ->
    (sequences (B,T,4), coords (B,T,7,3), mask (B,T), ids); zero padding, mask = prefix of ones
    (the layout every kernel of this package assumes)."""
    seqs = pad_sequence([it['sequence'].to(torch.float32) for it in batch], batch_first=True)
    xyz = pad_sequence([it['coordinates'].to(torch.float32) for it in batch], batch_first=True)
    lengths = torch.tensor([it['sequence'].shape[0] for it in batch])
    mask = (torch.arange(seqs.shape[1]).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.float32)
    return seqs, xyz, mask, [it['id'] for it in batch]


def separate(concat: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Inverse of boolean-mask selection: a concatenation of per-RNA vectors -> zero-padded (B, max_len)."""
    pieces = torch.split(concat, [int(n) for n in lengths.tolist()])
    return pad_sequence(list(pieces), batch_first=True) if pieces else concat.new_zeros((0, 0))


def check_prefix_mask(mask: torch.Tensor) -> None:
    """The kernels assume the collate's prefix masks (ones then zeros per row)."""
    m = mask.detach().cpu() != 0
    if bool((m[:, 1:] & ~m[:, :-1]).any()):
        raise ValueError("mask rows must be prefixes of ones (as produced by featurize)")


def pack_batch(coords_list: Sequence[torch.Tensor], pin: bool = True):
    """Var-len collate: a list of (L_i,7,3) tensors -> (coords_packed (N,7,3) f32,
    cu_seqlens (B+1) int32, max_len), in pinned host memory so ``.to(device, non_blocking=True)`` overlaps
    the copy with compute.  No padding is materialised or copied (the padded collate moves B*T rows)."""
    lengths = torch.tensor([int(c.shape[0]) for c in coords_list], dtype=torch.int32)
    cu = torch.zeros(len(coords_list) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lengths, 0)
    packed = torch.empty((int(cu[-1]), 7, 3), dtype=torch.float32, pin_memory=pin and torch.cuda.is_available())
    torch.cat([c.to(torch.float32) for c in coords_list], dim=0, out=packed)
    if pin and torch.cuda.is_available():
        cu = cu.pin_memory()
    return packed, cu, int(lengths.max()) if len(coords_list) else 0
