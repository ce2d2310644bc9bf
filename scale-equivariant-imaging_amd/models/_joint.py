"""One backward pass for the step's two model calls.

ProposedLoss evaluates the network twice per step: on the 2B crops [y, y + tau b] of the SURE term and on the B
measurements of the equivariance term (reference src/losses/__init__.py:133-142, src/losses/sure.py:7-32, deepinv's
EILoss). The forward passes depend on each other (the second input is built from the first output), the BACKWARD passes
do not: with the reference's default `stop_gradient` the second input is a constant. Autograd nevertheless runs them
one after the other, layer by layer -- every kernel of the backward pass twice, on 2B and on B images, the second time at
half the rows per launch of kernels that are latency- or weight-bandwidth-bound at these sizes.

Here the backward pass of both calls runs ONCE, on 3B images:

  * while a model call is recorded (`Recorder.call`), every activation the layers allocate comes from an ARENA: the first
    call allocates 3B-row buffers and uses their first 2B rows, the second call takes the tails of the same buffers, so
    that afterwards every saved tensor of a layer exists as ONE contiguous 3B-row tensor; every layer function appends
    (its class, its autograd ctx) to the call's tape;
  * the model output is DETACHED and re-enters autograd through `_Top`: the first of the two to receive its gradient parks
    it, the second starts `walk`, which plays the tape backwards -- one call of each layer's own backward function with a
    joint ctx (the 3B-row saved tensors) and a 3B-row gradient, skip gradients routed by a stack as the U-Net nests them
    -- while autograd never visits the layer nodes underneath (they live on through the tape; the network inputs need no
    gradient: otherwise nothing is recorded);
  * a call whose partner never shows up in the same backward pass (only one term differentiated) is walked alone when
    the pass ends (engine callback); a call in which something has no joint form (a float32-path layer) or does not
    mirror the first call (a buffer of another shape) is not cut out: it stays an ordinary autograd graph.

Weight gradients: a layer's backward runs once, so the operands of its weight gradient hold both calls' rows; they are
handed to models/_ops.weight_grad16 as the two row segments it otherwise merges from two backward functions (same
launches: the stored / Adam-epilogue / streamed forms all see "two pairs of the step's two calls").

SEI_NO_JOINT_BACKWARD=1 switches the mechanism off (A/B runs; the tests compare both).
"""
import os

import torch

ENABLED = os.environ.get("SEI_NO_JOINT_BACKWARD") != "1"
KEEP_ARENA = os.environ.get("SEI_KEEP_ARENA") == "1"       # bench.py's roofline leg re-issues a step's launches afterwards: their operands must stay allocated


class JointCtx:
    """What a layer's backward reads from its ctx, for both calls at once."""

    def __init__(self, ctx, saved, needs_input_grad=None, **override):
        for k, v in vars(ctx).items():
            if not k.startswith("_"):
                setattr(self, k, v)
        self.saved_tensors = tuple(saved)
        self.needs_input_grad = tuple(ctx.needs_input_grad) if needs_input_grad is None else tuple(needs_input_grad)
        for k, v in override.items():
            setattr(self, k, v)


class Pair:
    """The step's (up to) two recorded model calls: arena, tapes, and what their backward passes have done so far. A graph
    keeps the Pair it was recorded into (through `_Top`), whatever the recorder starts next."""

    def __init__(self):
        self.calls = 0                  # recorded calls
        self.bases = []                 # arena: 3B-row buffers in allocation order
        self.base_of = {}               # data_ptr of a first-call view -> its base
        self.tapes = [[], []]
        self.outputs = [None, None]     # the calls' outputs inside autograd: they keep the layer nodes alive
        self.broken = False
        self.parked = {}                # call index -> gradient of that call's output, waiting for the partner
        self.done = [False, False]
        self.batch = [0, 0]
        self.in_shape = None

    def joint(self, t1, t2):
        """The 3B-row tensor whose first rows are t1 (first call) and whose tail is t2 (second call)."""
        base = self.base_of.get(t1.data_ptr())
        if base is None or t2.numel() * 2 != t1.numel() or t1.dtype != t2.dtype or \
                t2.data_ptr() != base.data_ptr() + t1.numel() * t1.element_size() or base.numel() != 3 * t2.numel():
            raise _NotJoint()
        return base.view((t1.shape[0] // 2 * 3,) + tuple(t1.shape[1:]))


class Recorder:
    """Per backbone: records the step's first two differentiated model calls into a Pair."""

    def __init__(self):
        self.reset()

    def reset(self):
        """A new step (zero_grad): a fresh Pair; graphs recorded earlier keep theirs."""
        self.armed = False              # the loss has announced a 2B + B pair of calls for this step (expect_pair)
        self.pair = Pair()
        self.cursor = 0
        self.current = None             # call being recorded (0 / 1) or None

    # -- forward --------------------------------------------------------------------------------------------
    def expect_pair(self):
        """The loss is about to call the model on 2B images and then on B images whose gradient nobody needs."""
        if ENABLED and self.pair.calls == 0:
            self.armed = True

    def wants(self, y):
        """Record this model call? The step's first differentiated call (2B images), or its second with exactly half
        as many images of the same size; inputs that need no gradient."""
        p = self.pair
        if not ENABLED or p.broken or not torch.is_grad_enabled() or y.requires_grad or not y.is_cuda:
            return False
        if p.calls == 0:
            return y.shape[0] % 2 == 0
        return p.calls == 1 and 2 * y.shape[0] == p.batch[0] and tuple(y.shape[1:]) == p.in_shape

    def begin(self, y):
        p = self.pair
        self.current = p.calls
        p.batch[self.current] = y.shape[0]
        if self.current == 0:
            p.in_shape = tuple(y.shape[1:])
        self.cursor = 0

    def end(self):
        """True: the call is on its tape and may be cut out of autograd's graph (`_Top`); False: something in it has no
        joint form or does not mirror the first call -- it stays an ordinary autograd graph and is not counted."""
        p = self.pair
        if self.current == 1 and self.cursor != len(p.bases):
            p.broken = True             # the second call allocated fewer buffers than the first: the tapes do not mirror
        ok = not p.broken
        if ok:
            p.calls += 1
        else:
            p.tapes[self.current] = []
        self.current = None
        return ok

    def alloc(self, shape, dtype, device):
        p = self.pair
        shape = tuple(int(v) for v in shape)
        if p.broken:
            return torch.empty(shape, dtype=dtype, device=device)
        if self.current == 0:
            if shape[0] % 2:
                p.broken = True
                return torch.empty(shape, dtype=dtype, device=device)
            base = torch.empty((shape[0] // 2 * 3,) + shape[1:], dtype=dtype, device=device)
            p.bases.append(base)
            # (.data: an alias with its OWN version counter -- the second call's writes into the tail of this buffer must
            # not look like in-place modifications of what the first call saved for its backward)
            view = base[:shape[0]].data
            p.base_of[view.data_ptr()] = base
            return view
        if self.cursor >= len(p.bases):
            p.broken = True
            return torch.empty(shape, dtype=dtype, device=device)
        base = p.bases[self.cursor]
        self.cursor += 1
        if base.dtype != dtype or tuple(base.shape[1:]) != shape[1:] or base.shape[0] != 3 * shape[0]:
            p.broken = True
            return torch.empty(shape, dtype=dtype, device=device)
        return base[2 * shape[0]:].data

    def record(self, fn, ctx):
        if self.current is not None:
            self.pair.tapes[self.current].append((fn, ctx))

    def unsupported(self):
        self.pair.broken = True


class _NotJoint(Exception):
    pass


def recorder_of(backbone):
    rec = backbone.__dict__.get("_sei_joint")
    if rec is None:
        rec = backbone.__dict__["_sei_joint"] = Recorder()
    return rec


class _Top(torch.autograd.Function):
    """The recorded call's output re-enters autograd here: `x_hat` arrives DETACHED (autograd never descends into the
    layer nodes, which live on through the tape), `anchor` is a parameter that only makes the output require a gradient.
    The gradient that arrives starts, or waits for, the joint backward pass."""

    @staticmethod
    def forward(ctx, x_hat, anchor, pair, call, backbone):
        ctx.pair, ctx.call, ctx.backbone = pair, call, backbone
        return x_hat.view_as(x_hat)

    @staticmethod
    def backward(ctx, go):
        rec, call = ctx.pair, ctx.call
        if rec.done[call]:
            # the tapes (and the layer nodes' saved activations) were released by the first walk: autograd's own graph
            # raises here too ("backward through the graph a second time"); silently returning would lose gradients
            raise RuntimeError("this recorded model call has already been walked backward (its saved activations are "
                               "freed); a second backward pass through it needs SEI_NO_JOINT_BACKWARD=1")
        rec.parked[call] = go.contiguous()
        other = 1 - call
        if rec.calls == 2 and other in rec.parked and not rec.done[other]:
            _walk(rec, ctx.backbone, (0, 1))
        elif not _queue_finish(rec, ctx.backbone):
            _finish(rec, ctx.backbone)                       # no engine callback available: walk this call alone now
        return None, None, None, None, None


def _queue_finish(rec, backbone):
    try:
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _finish(rec, backbone))
    except RuntimeError:
        return False
    return True


def _finish(rec, backbone):
    """End of the backward pass: calls whose partner never received a gradient are walked alone."""
    for call in (0, 1):
        if call in rec.parked and not rec.done[call]:
            _walk(rec, backbone, (call,))


def _walk(rec, backbone, which):
    from . import _ops
    joint = len(which) == 2
    tapes = [rec.tapes[c] for c in which]
    if joint and (len(tapes[0]) != len(tapes[1]) or any(a[0] is not b[0] for a, b in zip(*tapes))):
        joint = False
    if not joint and len(which) == 2:                        # tapes that do not mirror each other: one after the other
        _walk(rec, backbone, (which[0],))
        _walk(rec, backbone, (which[1],))
        return
    for c in which:
        rec.done[c] = True
    if joint:
        try:
            ctxs = [_ops.joint_ctx(fn, c1, c2, rec) for (fn, c1), (_, c2) in zip(*tapes)]
        except _NotJoint:
            rec.done[which[0]] = rec.done[which[1]] = False
            _walk(rec, backbone, (which[0],))
            _walk(rec, backbone, (which[1],))
            return
        g1, g2 = rec.parked[0], rec.parked[1]
        import _native as N
        go = N.copy_into(torch.empty((g1.shape[0] + g2.shape[0],) + tuple(g1.shape[1:]), dtype=g1.dtype, device=g1.device),
                         g1.contiguous(), g2.contiguous())
        fns = [fn for fn, _ in tapes[0]]
    else:
        ctxs = [c for _, c in tapes[0]]
        go = rec.parked[which[0]]
        fns = [fn for fn, _ in tapes[0]]
    with _ops.joint_rows(backbone, (rec.batch[0], rec.batch[1]) if joint else None):
        _ops.walk_backward(fns, ctxs, go)
    for c in which:                                          # the layer nodes and their saved activations may go
        rec.outputs[c] = None
        rec.tapes[c] = []
        rec.parked.pop(c, None)
    if not KEEP_ARENA and all(rec.done[c] or c >= rec.calls for c in (0, 1)):   # every recorded call is walked: the 3B-row arena may go too
        rec.bases, rec.base_of = [], {}
