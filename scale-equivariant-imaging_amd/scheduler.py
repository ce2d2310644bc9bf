"""Learning-rate schedules (reference call surface: src/scheduler.py). Built from the same torch
scheduler classes as the reference so that `scheduler.state_dict()` inside checkpoints interchanges."""
from torch.optim.lr_scheduler import LinearLR, MultiStepLR, SequentialLR


def get_lr_scheduler(optimizer, epochs, lr_scheduler_kind):
    if lr_scheduler_kind == "multi_step_decay":
        return MultiStepLR(optimizer, milestones=[epochs * pct // 100 for pct in (50, 80, 90, 95)], gamma=0.5)
    if lr_scheduler_kind == "delayed_linear_decay":
        half = epochs // 2
        hold = LinearLR(optimizer, start_factor=1, end_factor=1, total_iters=half)
        decay = LinearLR(optimizer, start_factor=1, end_factor=1e-2, total_iters=half - 1)
        return SequentialLR(optimizer, [hold, decay], [half])
    raise ValueError(f"Unknown lr_scheduler_kind: {lr_scheduler_kind}")
