"""Blur kernel table (reference: src/physics/kernels.py:3-28). float64, sums to 1."""
import torch

_table = {f"{fam}_R{lvl}": (kind, lvl)
          for fam, kind, lvls in (("Gaussian", "gaussian", (1, 2, 3)), ("Box", "box", (2, 3, 4)))
          for lvl in lvls}


def taps_1d(name, dtype=torch.float64):
    """The 1-D factor t with kernel = outer(t, t): every table kernel is exactly rank 1."""
    assert name in _table, f"Unsupported kernel: {name}"
    kind, level = _table[name]
    if kind == "gaussian":
        n = 6 * level + 1
        u = torch.arange(n, dtype=dtype) - (n - 1) / 2
        t = torch.exp(-(u**2) / (2 * level**2))
    else:
        n = 2 * level + 1
        t = torch.ones(n, dtype=dtype)
    return t


def get_kernel(name, dtype=torch.float64):
    """(k,k) kernel: Gaussian of std R on a (6R+1)^2 grid, or a (2R+1)^2 box; normalised to sum 1.
    Computed as the reference does (2-D exp / 2-D sum) so the float64 values match it bit for bit."""
    assert name in _table, f"Unsupported kernel: {name}"
    kind, level = _table[name]
    if kind == "gaussian":
        n = 6 * level + 1
        u = torch.arange(n, dtype=dtype) - (n - 1) / 2
        sq = u.view(-1, 1) ** 2 + u.view(1, -1) ** 2
        k = torch.exp(-sq / (2 * level**2))
    else:
        n = 2 * level + 1
        k = torch.ones(n, n, dtype=dtype)
    return k / k.sum()
