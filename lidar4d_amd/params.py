"""Flat parameter / gradient / fp16-compute arenas.

The reference keeps ~70 separate parameter tensors (24 hex-planes, 1 + 24 hash tables, flow grid + 3 Linear
weights, 3 tcnn networks) and tiny-cuda-nn re-casts each fp32 master copy to fp16 on every
forward.  Here every trainable tensor is a view into ONE fp32 buffer (same state-dict keys and shapes), gradients
are views into one fp32 buffer and the fp16 compute copies are views into one fp16 buffer, so that per step the
cast is one launch, Adam is one launch per lr group and the data-parallel gradient all-reduce is one collective
over one contiguous buffer.  Layout follows LiDAR4D.get_params (model/lidar4d.py:226-237): group 0 (lr) =
planes + hash encoders, group 1 (0.1 lr) = flow / sigma / intensity / raydrop networks.
"""
import torch

from . import ops

ALIGN = 8  # elements: 32-byte fp32 / 16-byte fp16 alignment for vector loads
GATE_TAIL = 32  # floats after the last parameter's slot in the GRADIENT arena: "this range received a gradient" gates

_EPOCH = [0]


def bump_epoch():
    """Called by code that rewrites parameters behind torch's version counters (the HIP Adam kernel)."""
    _EPOCH[0] += 1


class ParamStore:
    def __init__(self, groups, pad_to=None):
        """groups: list of lists of (name, nn.Parameter); order defines the arena layout.
        pad_to: {name: numel} reserves a larger zero-filled slot (e.g. the flow MLP's [6,64] output layer is read
        by the MFMA kernel as a zero-padded [16,64] matrix)."""
        self.groups = groups
        pad_to = pad_to or {}
        self.entries = []  # (name, param, offset, numel, group)
        off = 0
        self.group_ranges = []
        for gi, g in enumerate(groups):
            start = off
            for name, p in g:
                n = p.numel()
                self.entries.append((name, p, off, n, gi))
                slot = max(n, pad_to.get(name, 0))
                off += (slot + ALIGN - 1) // ALIGN * ALIGN
            self.group_ranges.append((start, off))
        self.numel = off
        self.by_param = {id(p): (o, n) for _, p, o, n, _ in self.entries}
        self.grad_numel = off + GATE_TAIL  # the gradient arena carries the optimiser's gates behind the parameters' slots
        self.flat = None
        self.flat_grad = None
        self.flat16 = None
        self._key16 = None
        self.version16 = 0  # bumped whenever the contents of the fp16 compute copy change (derived copies key on it)
        self.build()

    def build(self):
        """(Re)allocate the fp32 arena on the parameters' current device and re-point every parameter at it."""
        live = [p for _, p, _, n, _ in self.entries if n > 0]
        device = live[0].device if live else torch.device("cpu")
        flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        with torch.no_grad():
            for _, p, off, n, _ in self.entries:
                if n == 0:
                    continue
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.flat = flat
        self.flat_grad = None
        self.flat16 = None
        self._key16 = None

    # -- fp16 compute copies -------------------------------------------------------------------------
    def _key(self):
        return (self.flat.data_ptr(), _EPOCH[0], sum(p._version for _, p, _, _, _ in self.entries))

    def refresh16(self):
        key = self._key()
        if key != self._key16:
            if self.flat16 is None or self.flat16.device != self.flat.device:
                self.flat16 = torch.empty(self.numel, dtype=torch.float16, device=self.flat.device)
            ops.cast_f32_to_f16(self.flat, self.flat16)
            self._key16 = key
            self.version16 += 1
        return self.flat16

    def mark16_current(self):
        """The caller has just written flat16 itself (Adam kernel emits the fp16 copy)."""
        self._key16 = self._key()
        self.version16 += 1

    def half(self, param):
        off, n = self.by_param[id(param)]
        return self.refresh16()[off:off + n]

    # -- gradients ------------------------------------------------------------------------------------
    def grad_view(self, param):
        off, n = self.by_param[id(param)]
        return self.flat_grad[off:off + n]

    @property
    def gates(self):
        """[GATE_TAIL] fp32 view behind the gradients: gates[s] != 0 <=> HashGridT time slice s received a gradient since
        the last zero_grad (the reference leaves the other slices' .grad at None and torch.optim.Adam skips them).  Being
        part of the gradient arena they are zeroed with it and merged across ranks by the same SUM all-reduce."""
        return self.flat_grad[self.numel:self.numel + GATE_TAIL]

    def prepare_grads(self):
        """Make every parameter's .grad a view into the flat gradient arena; zero it if no gradient was held.
        Returns the arena.  Foreign .grad tensors (not our views) are folded in and replaced."""
        if self.flat_grad is None or self.flat_grad.device != self.flat.device:
            self.flat_grad = torch.zeros(self.grad_numel, dtype=torch.float32, device=self.flat.device)
            fresh = True
        else:
            fresh = False
        base = self.flat_grad.data_ptr()
        all_none = all(p.grad is None for _, p, _, n, _ in self.entries if n > 0)
        if all_none and not fresh:
            self.flat_grad.zero_()
        for _, p, off, n, _ in self.entries:
            if n == 0:
                continue
            view = self.flat_grad[off:off + n].view(p.shape)
            if p.grad is None:
                if not all_none and not fresh:
                    view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != base + 4 * off:
                view.copy_(p.grad)
                p.grad = view
        return self.flat_grad

    def zero_grad(self):
        if self.flat_grad is not None:
            self.flat_grad.zero_()
            self.prepare_grads()
