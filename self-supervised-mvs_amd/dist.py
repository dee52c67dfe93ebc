"""Data parallelism for the path: one process per GPU, independent (ref, src[], depth-range)
samples per rank, and ONE RCCL all-reduce per optimiser step over a single flat fp32 gradient
bucket (1.35 MB for MVSNet, 2.21 MB for CVP-MVSNet) -- the MI355X-native counterpart of the
reference's single-process nn.DataParallel (jdacs/train.py:65, jdacs-ms/train.py:91), whose
replicate/broadcast + ReduceAddCoalesced traffic it replaces.

xGMI is point-to-point (7 links/GPU); at this message size the collective is latency bound, so one
bucket issued once after backward is the right shape -- no bucketing/overlap machinery.
BatchNorm statistics stay per replica, exactly like DataParallel without SyncBN.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun env).
    Returns (rank, world, local_rank).  backend: "nccl" (== RCCL on ROCm) on GPUs, "gloo" on CPU.
    force: create the process group at world size 1 too (bench.py --force-collective: the collective's code path -- RCCL
    communicator set-up and one all-reduce per step -- then runs on a single-GPU box; a free loopback port is chosen if
    MASTER_PORT is unset)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force and world == 1 and "MASTER_PORT" not in os.environ:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradBucket:
    """One contiguous fp32 gradient buffer for the whole model; ``all_reduce()`` is a single collective
    (sum) followed by 1/world (loss = mean over the global batch).

    Per step: ``zero()`` drops the parameters' .grad (so autograd WRITES fresh gradients instead of
    launching one accumulate-add per parameter), ``gather()`` packs them into the flat buffer with one
    concatenation, ``all_reduce()`` runs the collective on it.  With ``flatten_params=True`` the parameters
    themselves become views of one flat tensor (``flat_param``, whose .grad is the bucket), so the
    optimiser can run as ONE fused update over one tensor instead of 100+ small launches."""

    def __init__(self, params: Iterable[torch.nn.Parameter], flatten_params: bool = False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_param = None
        if flatten_params:
            # every parameter keeps its PHYSICAL layout (a channels-last conv weight stays channels-last): the chunk of the
            # store holds the parameter's elements in storage order and the view re-applies its strides
            for p in self.params:
                if not self._is_dense(p):
                    raise ValueError("FlatGradBucket(flatten_params=True) needs dense parameters, got strides %s for shape %s"
                                     % (p.stride(), tuple(p.shape)))
            store = torch.cat([self._storage_order(p.detach()) for p in self.params])
            off = 0
            for p in self.params:
                n = p.numel()
                p.data = store[off:off + n].as_strided(p.shape, p.stride())
                off += n
            self.flat_param = torch.nn.Parameter(store)
            self.flat_param.grad = self.flat

    def optimizer_params(self, slices: int = 32):
        """flatten_params: the flat store as ``slices`` equal 1-D Parameter views, each with its .grad view of the bucket, for a
        fused multi-tensor optimiser.  Such an optimiser gives every (tensor, 65536-element chunk) one workgroup: the whole model
        as ONE 338k-element tensor is 6 workgroups (98 us per Adam step on an MI355X, measured), 32 slices are 32 workgroups in
        the same single launch.  The update is element-wise, so the result is the same as for the one tensor."""
        if self.flat_param is None:
            raise ValueError("optimizer_params() needs FlatGradBucket(flatten_params=True)")
        n = self.flat_param.numel()
        step = max(4, -(-n // max(1, slices)))
        step += (-step) % 4
        out = []
        self._opt_slices = []
        for i in range(0, n, step):
            p = torch.nn.Parameter(self.flat_param.data[i:i + step])
            p.grad = self.flat[i:i + step]
            out.append(p)
            self._opt_slices.append((p, i, i + step))
        return out

    @staticmethod
    def _is_dense(t: torch.Tensor) -> bool:
        """True if the strides are a permutation layout without gaps or overlaps (contiguous, channels-last, ...)."""
        expect = 1
        for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz > 1), key=lambda x: x[1]):
            if stride != expect:
                return False
            expect *= size
        return True

    @staticmethod
    def _storage_order(t: torch.Tensor) -> torch.Tensor:
        """The elements of a dense tensor in the order they lie in memory, as a 1-D tensor (a view)."""
        return t.as_strided((t.numel(),), (1,))

    def check_aliasing(self) -> None:
        """flatten_params: every parameter must still be a view of the flat store (something that re-allocates parameter
        storage after the bucket was built -- module.to(memory_format=...), .double(), ... -- would detach it from the
        optimiser silently)."""
        if self.flat_param is None:
            return
        lo = self.flat_param.data_ptr()
        hi = lo + self.flat_param.numel() * 4
        for p in self.params:
            if not (lo <= p.data_ptr() < hi):
                raise RuntimeError("FlatGradBucket: a parameter of shape %s no longer aliases the flat store; build the "
                                   "bucket after every layout / dtype conversion of the model" % (tuple(p.shape),))

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def zero(self) -> None:
        for p in self.params:
            p.grad = None

    def gather(self) -> None:
        """Pack the fresh gradients into the flat buffer: ONE multi-tensor concatenation launch.  (Gradients as permanent
        views of the bucket would instead cost one accumulate-add launch per parameter and step: autograd adds into an
        existing .grad, it only hands over ownership when .grad is None.)  With flatten_params the gradient of a
        parameter is laid out like the parameter (same strides), so bucket[i] is the gradient of flat_param[i]."""
        # every step: one tuple of the parameters' storage addresses against the one seen last (55 data_ptr() calls ~ 15 us of launch-thread
        # time; round 5 walked only every 32nd step and could skip a re-allocated parameter for 31 optimiser steps -- ADVICE r5).  On a
        # change: the aliasing check (raises if a parameter left the flat store) and fresh gradient views (strides may have changed).
        ptrs = tuple(p.data_ptr() for p in self.params)
        if ptrs != getattr(self, "_param_ptrs", None):
            self.check_aliasing()
            self._param_ptrs = ptrs
            self._grad_views = None
        fast = self.flat_param is not None and hasattr(torch, "_foreach_copy_")
        if fast:
            # the common step: every parameter has a gradient in the parameter's own layout -> one multi-tensor copy into cached
            # views of the bucket that carry those strides (no per-tensor view construction on the launch thread)
            views = getattr(self, "_grad_views", None)
            if views is None:
                views, off = [], 0
                for p in self.params:
                    n = p.numel()
                    views.append(self.flat[off:off + n].as_strided(p.shape, p.stride()))
                    off += n
                self._grad_views = views
            grads = [p.grad for p in self.params]
            for g, v in zip(grads, views):
                if g is None or g.dtype != v.dtype or g.stride() != v.stride():
                    fast = False
                    break
            if fast:
                torch._foreach_copy_(views, grads)
        if not fast:
            grads = []
            for p in self.params:
                g = p.grad
                if g is None:
                    g = torch.zeros(p.numel(), dtype=torch.float32, device=self.flat.device)
                elif self.flat_param is not None and g.stride() != p.stride():
                    # autograd handed over a gradient in another layout than the parameter's: re-lay it out (one small copy)
                    g = self._storage_order(torch.empty_strided(p.shape, p.stride(), dtype=g.dtype, device=g.device).copy_(g))
                else:
                    g = self._storage_order(g) if self.flat_param is not None else g.reshape(-1)
                grads.append(g)
            torch.cat(grads, out=self.flat)
        # an optimizer.zero_grad() (set_to_none=True is torch's default) drops the slices' .grad views of the bucket and the
        # next opt.step() would then skip every slice without an error: re-attach them (no launch, views only)
        for p, lo, hi in getattr(self, "_opt_slices", ()):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * lo:
                p.grad = self.flat[lo:hi]

    force_collective = False    # True: run the collective at world size 1 as well (bench.py --force-collective)

    def all_reduce(self) -> None:
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective):
            _all_reduce_sum(self.flat)
            if dist.get_world_size() > 1:
                self.flat.div_(dist.get_world_size())


def _host_staged(t: torch.Tensor) -> bool:
    """gloo (the CPU backend used by the tests, also for ranks that share ONE GPU) moves host memory: device tensors are
    staged through a host copy.  nccl (== RCCL, the product path: one rank per GPU over xGMI) takes the device tensor as is."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce_sum(t: torch.Tensor) -> None:
    if _host_staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """One-off broadcast of initial weights + buffers so every rank starts from rank `src`'s model."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            if _host_staged(t.data):
                h = t.data.cpu()
                dist.broadcast(h, src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src)
