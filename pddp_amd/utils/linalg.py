"""Small dense solves the host side needs (GP conditioning, the torch BoxQP)."""
import torch


def cholesky_solve(B, L, upper=False):
    """(L L^T)^-1 B (or (U^T U)^-1 B with `upper`) by two triangular solves.

    Same contract as torch.cholesky_solve, which is NOT used: on the
    PyTorch 2.10 / ROCm 7.0 build of this image its batched MAGMA path writes
    outside its outputs - `cholesky_solve(I, L)` followed by
    `cholesky_solve(y, L)` on a [6, 24, 24] batch clobbers neighbouring blocks
    of the caching allocator (tools/dbg/torch_potrs_canary.py reproduces it
    with torch alone), which showed up as corrupted index tensors and a GPU
    exception many launches later.  Triangular solves go through hipBLAS."""
    if upper:
        w = torch.linalg.solve_triangular(L.transpose(-1, -2), B, upper=False)
        return torch.linalg.solve_triangular(L, w, upper=True)
    w = torch.linalg.solve_triangular(L, B, upper=False)
    return torch.linalg.solve_triangular(L.transpose(-1, -2), w, upper=True)
