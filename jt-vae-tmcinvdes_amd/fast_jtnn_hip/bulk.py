"""Bulk latent generation: encode a large set of molecular graphs without gradients.

The reference does this one small batch at a time, collects every batch on the host and writes a
text file at the end (fast_bo/gen_latent.py:88-99, through ``encode_from_smiles``,
fast_jtnn/jtnn_vae.py:50-54).  Here every rank owns a contiguous range of the items, streams its
batches through the encoder with the device-to-host copy of batch k overlapping the encoding of batch
k+1, and writes its rows straight into a binary ``.npy`` shard (memory-mapped, so 10 M x 900 floats
never sit in host memory at once).  There is no exchange step between ranks.  ``shards_to_text``
produces the reference's ``latent_features.txt`` for consumers that read it with ``np.loadtxt``
(fast_bo/run_bo.py:81).
"""
from __future__ import annotations

import os
from typing import Callable, Sequence, Tuple

import numpy as np
import torch

from .parallel import shard_bounds


def shard_path(prefix: str, rank: int, world: int) -> str:
    return "%s.%05d-of-%05d.npy" % (prefix, rank, world)


def encode_bulk(model, n_items: int, make_holders: Callable[[int, int], Tuple], batch_size: int, prefix: str,
                what: str = "vectors", rank: int = 0, world: int = 1, prefetch: int = 2, shard=None) -> str:
    """Encode items ``[lo, hi)`` of this rank and write them to its shard; returns the shard's path.

    ``make_holders(lo, hi)`` returns ``(jtenc_holder, mpn_holder)`` of items ``lo..hi-1`` (the
    reference's ``tensorize(..., assm=False)`` output, fast_jtnn/datautils.py:140-163).  With ``shard``
    (a ``flat.FlatShard`` holding the items) ``make_holders`` is not used: the holders of a batch are built
    from the flat arrays by the native builders, in the prefetch threads.
    ``what``: ``"vectors"`` stores ``cat(tree_vecs, mol_vecs)`` ``[n, 2H]`` (what ``encode_from_smiles``
    returns), ``"mean"`` the latent means ``[n, latent]`` of ``encode_latent``.
    ``prefetch``: batches that a background thread tensorizes, prepares and uploads ahead of the encoder
    (0: everything in the calling thread).
    """
    if what not in ("vectors", "mean"):
        raise ValueError("what must be 'vectors' or 'mean'")
    if batch_size < 1:
        raise ValueError("batch_size must be positive")
    lo, hi = shard_bounds(n_items, world, rank)
    dev = next(model.parameters()).device
    width = 2 * model.hidden_size if what == "vectors" else 2 * model.latent_size
    path = shard_path(prefix, rank, world)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(hi - lo, width))
    staging = [torch.empty((batch_size, width), dtype=torch.float32).pin_memory() for _ in range(2)]
    events = [torch.cuda.Event(), torch.cuda.Event()]
    pending = None  # (buffer index, first row, rows) of the batch whose copy is in flight

    def drain():
        nonlocal pending
        if pending is not None:
            i, row, n = pending
            events[i].synchronize()
            out[row:row + n] = staging[i][:n].numpy()
            pending = None

    def from_flat(b_lo, b_hi):
        from . import flatprep

        return flatprep.encoder_batch(model, shard, np.arange(b_lo, b_hi), train=False)

    def holders():
        for b_lo in range(lo, hi, batch_size):
            b_hi = min(b_lo + batch_size, hi)
            if shard is not None and prefetch > 0:
                yield (b_lo, b_hi)  # built in a prefetch thread
            else:
                yield (b_lo, b_hi) + tuple((from_flat if shard is not None else make_holders)(b_lo, b_hi))

    source = holders()
    if prefetch > 0:
        from .prefetch import Prefetcher, warm_encoder_holders

        if shard is not None:
            prepare = lambda it: it + tuple(from_flat(*it))  # noqa: E731
        else:
            prepare = lambda it: warm_encoder_holders(model, it[2], it[3], train=False)  # noqa: E731
        source = Prefetcher(source, prepare, dev, depth=prefetch)
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            for k, (b_lo, b_hi, jt, mp) in enumerate(source):
                if what == "vectors":
                    res = model.encode_vectors(jt, mp)
                else:
                    res, _ = model.encode_latent(jt, mp)
                if res.shape != (b_hi - b_lo, width):
                    raise RuntimeError("encoder returned %s for %d items" % (tuple(res.shape), b_hi - b_lo))
                i = k & 1
                if pending is not None and pending[0] == i:
                    drain()
                staging[i][:b_hi - b_lo].copy_(res, non_blocking=True)
                events[i].record()
                if pending is not None:
                    drain()  # the previous batch: its copy ran while this batch was being encoded
                pending = (i, b_lo - lo, b_hi - b_lo)
            drain()
    finally:
        model.train(was_training)
        if prefetch > 0:
            source.close()
    out.flush()
    del out
    return path


def load_shards(prefix: str, world: int) -> np.ndarray:
    """All ranks' rows in item order."""
    return np.concatenate([np.load(shard_path(prefix, r, world), mmap_mode="r") for r in range(world)], axis=0)


def shards_to_text(prefix: str, world: int, txt_path: str, rows_per_chunk: int = 65536) -> int:
    """Write the shards as the reference's ``latent_features.txt`` (``np.savetxt`` defaults: '%.18e',
    space separated, one item per line; fast_bo/gen_latent.py:99).  Streams chunk by chunk; returns
    the number of rows written."""
    n = 0
    with open(txt_path, "w") as f:
        for r in range(world):
            a = np.load(shard_path(prefix, r, world), mmap_mode="r")
            for s in range(0, a.shape[0], rows_per_chunk):
                np.savetxt(f, np.asarray(a[s:s + rows_per_chunk], dtype=np.float64))
            n += a.shape[0]
    return n
