"""RNG forking helper (reference: src/rng.py)."""
import torch


def fork_rng(enabled):
    return torch.random.fork_rng(enabled=enabled)
