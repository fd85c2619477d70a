"""Molecule-batch data parallelism: one process per GPU, RCCL over xGMI.

Molecules never interact inside the encoder (all index tables are batch-local
concatenations, fast_jtnn/mpn.py:129-143, fast_jtnn/datautils.py:166-172), so a
global batch is cut by molecule *before* tensorization and each rank builds and
encodes its own shard.  The only exchange step of a training step is one
all-reduce of the gradients, done on ONE flat fp32 bucket (about 20 MB for the
whole model: a single ring pass per step; many small per-tensor collectives
would be latency-bound on point-to-point xGMI links).  Bulk encoding has no
exchange step at all.

Losses in the reference are normalised by the local batch size
(fast_jtnn/jtnn_vae.py:74,137), so with equal shards the average of the shard
gradients equals the gradient of the concatenated batch.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = None, device_index: int = None) -> Tuple[int, int, int]:
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank if device_index is None else device_index)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def launch_command(script: str, script_args: Sequence[str], n_procs: int, port: int = None) -> List[str]:
    """Command line that starts ``script`` as ``n_procs`` ranks of one node, one per GPU."""
    import sys

    if port is None:
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_procs),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(script_args)


def self_launch_if_needed(script: str, script_args: Sequence[str], n_gpus: int, json_only: bool = False) -> None:
    """``python script --gpus N`` with N > 1 and no launcher environment: start the N ranks as child
    processes (before this process has touched the GPU), pass their output through and exit with their
    return code.  Returns at once when N == 1 or when already running under a launcher.

    ``json_only``: the caller's contract is ONE JSON line on stdout; whatever else the ranks or the
    communication library print there (connection banners) is passed to stderr instead."""
    import subprocess
    import sys

    if n_gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes here
    cmd = launch_command(script, script_args, n_gpus)
    if not json_only:
        sys.exit(subprocess.run(cmd, env=env).returncode)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        is_json = line.lstrip().startswith("{") and line.rstrip().endswith("}")
        (sys.stdout if is_json else sys.stderr).write(line)
        (sys.stdout if is_json else sys.stderr).flush()
    sys.exit(proc.wait())


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split: the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank %d of world %d" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Balanced split of items with unequal cost (e.g. bonds + 20 x tree messages per molecule):
    largest first onto the currently lightest rank.  Deterministic for a given input."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


class FlatGradBucket:
    """The gradients of a module, joined in one contiguous buffer for the collective.

    ``zero()`` before backward drops the gradients (``.grad = None``): autograd then hands every parameter the
    tensor its backward produced, instead of adding it into a zeroed buffer - one add kernel per parameter and
    the fill are saved.  ``all_reduce_mean()`` after backward: with more than one rank the gradients are packed
    into the flat buffer by one copy launch, reduced by ONE collective, and ``.grad`` of every parameter becomes
    a view of the reduced buffer; with one rank nothing is copied at all.
    """

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n

    def zero(self) -> None:
        for p in self.params:
            p.grad = None

    def pack(self) -> None:
        """Copy every gradient into its slot of the flat buffer (zeros where a parameter got none) and point
        ``.grad`` at the slots."""
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            self.flat.zero_()
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None and g.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, force: bool = False) -> None:
        """``force``: run the pack + collective + mean even in a process group of one rank (a rehearsal of the
        communication path on a single GPU; the result is the step without it)."""
        if dist.is_initialized() and (dist.get_world_size() > 1 or force):
            self.pack()
            if self.flat.is_cuda and dist.get_backend() != "nccl":
                # gloo stages a device tensor through host memory on a stream of its own.  With two ranks sharing
                # one GPU on this ROCm stack that all-reduce never completes when the launch stream still has work
                # queued - with stock torch kernels alone, with one launch stream or four, on the null stream or an
                # explicit one (scratch/gloo_stall_probe.py; both ranks park in work.wait()).  Draining the launch
                # stream first (blocking or polling) is the one thing that avoids it, so host-staged backends get
                # this wait.  RCCL runs the collective as a kernel on the launch stream and takes no host wait.
                torch.cuda.current_stream(self.flat.device).synchronize()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()


def all_ranks_ok(ok: bool, device=None, finite: Optional[torch.Tensor] = None) -> bool:
    """Skip-bad-batch semantics of the reference trainer (fast_molvae/vae_train.py:99-108) under
    data parallelism: a step is skipped on every rank if it failed on any.

    ``finite`` (a 0-d bool tensor on the device, e.g. ``isfinite(loss)``) joins the host-side flag on the
    device, so that the decision costs ONE device-to-host copy per step at any world size."""
    multi = dist.is_initialized() and dist.get_world_size() > 1
    if not multi and finite is None:
        return ok
    bad = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device if finite is None else finite.device)
    if finite is not None:
        bad = bad + (~finite).to(torch.int32)
    if multi:
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
    return int(bad.item()) == 0
