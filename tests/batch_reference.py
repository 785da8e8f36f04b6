"""Tensor-library reference for the index structures of a stochastic batch (tests only): the listed rows of a
SparseSide gathered as COO triplets, from which svi.BatchSide builds the grouped forms with torch sorts.  The product
builds the same structures on the device with hpf_hip_svi_batch_prepare (svi.BatchWorkspace); the tests compare."""
import torch


def gather_rows(side, rows):
    """COO triplets (row, col, y) of the listed rows of a SparseSide (ascending `rows`: the triplets come grouped)."""
    st = side.indptr[rows]
    deg = side.indptr[rows + 1] - st
    total = int(deg.sum().item())
    offs = torch.cumsum(deg, 0) - deg
    pos = torch.repeat_interleave(st - offs, deg, output_size=total) + torch.arange(total, device=rows.device)
    return (torch.repeat_interleave(rows, deg, output_size=total), side.idx[pos].to(torch.int64), side.y[pos])
