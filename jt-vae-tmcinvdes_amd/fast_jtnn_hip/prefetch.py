"""Holder preparation off the critical path.

The reference builds the index tables of a batch on the host and copies them to the GPU inside every
``forward`` (``create_var``: fast_jtnn/nnutils.py:6-10, called from fast_jtnn/mpn.py:82-85,
jtnn_enc.py:21-25, jtmpn.py:84-87), so the GPU waits for the host once per step.  Here the device-side
form of a batch (CSR adjacency and its transpose, tree-GRU schedules for forward and backward, decoder
trace tables, padded feature matrices) is built and uploaded by a background thread on its own copy
stream while the GPU runs the previous batch; the modules find it attached to the holder tensors.

    for x_batch in Prefetcher(loader, lambda b: warm_training_batch(model, b), device):
        trainer.step(x_batch)
"""
from __future__ import annotations

import queue
import threading

import torch

from . import graphprep

_END = object()


def warm_encoder_holders(model, jtenc_holder, mpn_holder, train=True, messages_used=False):
    """Builds (and attaches to the holder tensors) what ``model.encode`` will look up: the same builders under
    the same keys as JTNNEncoder.forward / MPN.forward use.  Returns None: the holders themselves carry the result."""
    from . import jtnn_enc
    from .mpn import _holder_prep

    dev = model.mpn.W_h.weight.device
    V = model.jtnn.embedding.weight.shape[0]

    def warm_tree(holder):
        build = lambda: graphprep.prepare_tree(holder, dev, V, train)  # noqa: E731
        prep = graphprep.cached_prep(holder[3], ("tree", str(dev), V), build)
        if train:
            prep.ensure_training()
        sched = graphprep.tree_schedule(prep, model.jtnn.depth, mode=jtnn_enc.SCHEDULE)
        if train:
            graphprep.backward_schedule(prep, sched, all_messages=messages_used)
        return prep

    split = None
    if jtnn_enc.SCHEDULE == "once":
        # the deep-tree check JTNNEncoder.forward makes first, under the key it looks it up by
        split = graphprep.cached_prep(jtenc_holder[3], ("deep_split", model.jtnn.depth),
                                      lambda: graphprep.split_deep_trees(jtenc_holder, model.jtnn.depth))
    if split is None:
        warm_tree(jtenc_holder)
    else:
        for sub, _, _ in split:
            warm_tree(sub)
    _holder_prep(*mpn_holder, dev, 0, train)


def warm_training_batch(model, x_batch):
    """The whole x_batch of JTNNVAE.forward / JTpropVAE.forward: encoders, candidate graphs, decoder traces."""
    from .jtnn_dec import JTNNDecoder, decoder_plan_from_holder

    if len(x_batch) == 5:
        _, _, jt, mp, (cj, _) = x_batch
    else:
        _, jt, mp, (cj, _) = x_batch
    dev = model.mpn.W_h.weight.device
    warm_encoder_holders(model, jt, mp, train=True, messages_used=True)
    n_tree = int(jt[3].shape[0])
    build = lambda: graphprep.prepare_mpn(cj, dev, n_tree)  # noqa: E731
    graphprep.cached_prep(cj[3], ("jtmpn", str(dev), n_tree), build)
    if isinstance(getattr(model, "decoder", None), JTNNDecoder):
        graphprep.cached_prep(jt[3], ("decoder_plan",), lambda: decoder_plan_from_holder(jt))


class Prefetcher:
    """Iterates ``source`` in order, with ``prepare(item)`` run up to ``depth`` items ahead by ``workers``
    background threads.

    A feeder thread walks ``source`` (so the loader's own work is off the stepping thread too) and hands every
    item to the pool; ``prepare`` runs under the worker's own copy stream, so its host-to-device copies overlap
    the kernels of the batch in flight.  ``prepare`` either attaches its results to the item's holder tensors and
    returns None (the item itself is handed on), or returns what is to be handed on in the item's place (holders
    built from flat storage).  The consumer's stream waits for the copies (an event per item) when the item is
    handed over.  Device memory that ``prepare`` allocates is marked as read by the stream that was current when
    the Prefetcher was made; a consumer that iterates under another stream is refused rather than left to race
    with the allocator.  An exception raised by the source or by ``prepare`` is re-raised in the consumer at the
    position of the item that caused it.
    """

    def __init__(self, source, prepare, device, depth: int = 4, workers: int = None):
        from concurrent.futures import ThreadPoolExecutor

        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Prefetcher uploads to an MI355X; got device %s" % self.device)
        if workers is None:
            workers = default_workers()
        self.source, self.prepare = source, prepare
        self.q = queue.Queue(maxsize=max(1, depth))
        self.consumer = torch.cuda.current_stream(self.device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="jtvae-prep")
        self.tls = threading.local()
        self.stop = False
        self.thread = threading.Thread(target=self._feed, name="jtvae-prefetch", daemon=True)
        self.thread.start()

    def _prepare(self, item):
        if getattr(self.tls, "stream", None) is None:
            torch.cuda.set_device(self.device)
            self.tls.stream = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self.tls.stream), graphprep.consumer_stream(self.consumer):
            out = self.prepare(item)  # attaches its results to the item's holder tensors, or returns new holders
            ev = torch.cuda.Event()
            ev.record(self.tls.stream)
        return ev, item if out is None else out

    def _feed(self):
        try:
            for item in self.source:
                if self.stop:
                    break
                self.q.put((self.pool.submit(self._prepare, item), None))
            self.q.put((None, _END))
        except BaseException as e:  # handed to the consumer
            self.q.put((None, e))

    def __iter__(self):
        while True:
            fut, tag = self.q.get()
            if tag is _END:
                return
            if tag is not None:
                raise tag
            ev, item = fut.result()
            cur = torch.cuda.current_stream(self.device)
            if cur != self.consumer:
                raise RuntimeError("Prefetcher was created under stream %s and is iterated under %s: the uploaded "
                                   "tables are only protected for the former" % (self.consumer, cur))
            cur.wait_event(ev)
            yield item

    def close(self):
        self.stop = True
        while self.thread.is_alive():
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
        self.pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def default_workers() -> int:
    """Background preparation threads per rank: 3 where the host has the cores for every rank of the node to run
    that many (a worker also spends part of its time inside multi-threaded native builders), fewer otherwise."""
    import os

    env = os.environ.get("JTVAE_PREFETCH_WORKERS")
    if env:
        return max(1, int(env))
    world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    return max(1, min(3, (os.cpu_count() or 1) // (2 * world)))
