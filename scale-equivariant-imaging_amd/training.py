"""Checkpoint helpers (reference call surface: src/training.py). File format unchanged:
{"epoch", "params" (backbone state_dict), "optimizer", "scheduler"}."""
import os

import torch


def save_training_state(epoch, model, optimizer, scheduler, state_path):
    folder = os.path.dirname(state_path)
    if folder:
        os.makedirs(folder, exist_ok=True)
    print(f"writing the training state to the file {state_path}")
    state = {"epoch": epoch, "params": model.get_weights(), "optimizer": optimizer.state_dict(),
             "scheduler": scheduler.state_dict()}
    torch.save(state, state_path)


def get_weights(weights_name, device):
    """A local path, or the name of published weights (needs network access, as upstream)."""
    if os.path.exists(weights_name):
        weights = torch.load(weights_name, map_location=device)
    else:
        url = f"https://huggingface.co/jscanvic/scale-equivariant-imaging/resolve/main/{weights_name}.pt?download=true"
        weights = torch.hub.load_state_dict_from_url(url, map_location=device)
    return weights["params"] if "params" in weights else weights
