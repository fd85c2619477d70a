"""Training-step semantics of the reference's two trainers, for one process per GPU.

Restates ``fast_molvae/vae_train.py:54-58, 73-74, 99-108, 134-144`` (unconditional model) and
``fast_molopt/vae_train_prop.py:74-78, 97-98, 120-150`` (conditional model): parameter
initialisation, Adam + exponential learning-rate decay, gradient clipping at ``clip_norm``, the
"print the failure and skip the batch" rule, the KL-weight schedule and the ``model.iter-N`` /
``model.epoch-N`` state-dict checkpoints (same keys as the reference, so either side loads the other's).

What data parallelism adds (the reference is single-GPU): every rank steps its own shard of the
global batch; one flat-bucket mean all-reduce joins the gradients, and it runs BEFORE the clip so
that all ranks clip the same gradient and take the same optimizer step without another exchange; a
batch that failed on any rank is skipped on all of them.  One rule is added at every world size: a batch whose
loss is not finite is skipped like a failed one (the reference would step NaN gradients into the weights).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import parallel


def init_params(model: nn.Module) -> None:
    """vae_train.py:54-58: vectors to zero, matrices Xavier-normal."""
    for param in model.parameters():
        if param.dim() == 1:
            nn.init.constant_(param, 0)
        else:
            nn.init.xavier_normal_(param)


def param_norm(m: nn.Module) -> float:
    return math.sqrt(sum(float(p.detach().double().pow(2).sum()) for p in m.parameters()))


def grad_norm(m: nn.Module) -> float:
    return math.sqrt(sum(float(p.grad.detach().double().pow(2).sum()) for p in m.parameters() if p.grad is not None))


@dataclass
class Schedule:
    """Knobs of the reference's loops, under its flag names (vae_train.py:157-175)."""

    lr: float = 1e-3
    clip_norm: float = 50.0
    beta: float = 0.0
    step_beta: float = 0.002
    max_beta: float = 1.0
    warmup: int = 500
    anneal_rate: float = 0.9
    anneal_iter: int = 1000
    kl_anneal_iter: int = 300
    print_iter: int = 50
    save_iter: int = 1000


@dataclass
class StepResult:
    ok: bool  # False: the batch failed on some rank and was skipped everywhere
    loss: Optional[float] = None
    metrics: dict = field(default_factory=dict)
    error: Optional[str] = None


class Trainer:
    """One model replica on one GPU (or the CPU oracle in tests): ``step(batch)`` is one iteration of the
    reference's inner loop."""

    def __init__(self, model: nn.Module, sched: Schedule = None, save_dir: str = None, conditional: bool = False,
                 total_step: int = 0, log=print):
        self.model, self.sched, self.save_dir, self.conditional = model, sched or Schedule(), save_dir, conditional
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.bucket = parallel.FlatGradBucket(self.params)  # one buffer for the gradient all-reduce
        self.optimizer = torch.optim.Adam(self.params, lr=self.sched.lr)  # vae_train.py:73-74
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, self.sched.anneal_rate)
        self.beta = self.sched.beta
        self.total_step = total_step
        self.meters = np.zeros(4)
        self.log = log
        self.rank, _, self.world = parallel.dist_env()
        self.device = self.params[0].device
        self.force_collective = False  # True: the gradient all-reduce runs even with one rank (rehearsal)

    def forward_backward(self, batch, epsilons=None):
        """model(batch, beta) + backward into the bucket; returns (loss tensor, metrics dict)."""
        out = self.model(batch, self.beta, epsilons) if epsilons is not None else self.model(batch, self.beta)
        if self.conditional:  # jtprop_vae.py:441 returns (loss, log_metrics)
            loss, metrics = out
        else:  # jtnn_vae.py:97-103 returns (loss, kl, word acc, topo acc, assm acc)
            loss, kl, wacc, tacc, sacc = out
            metrics = {"kl_div": kl, "word_acc": wacc * 100, "topo_acc": tacc * 100, "assm_acc": sacc * 100}
        loss.backward()
        return loss, metrics

    def step(self, batch, epsilons=None) -> StepResult:
        self.total_step += 1
        self.bucket.zero()  # model.zero_grad() of the reference
        loss, metrics, err, finite = None, {}, None, None
        try:
            loss, metrics = self.forward_backward(batch, epsilons)
            ok = True
            finite = torch.isfinite(loss.detach())
        except Exception as e:  # vae_train.py:105-108: report and go on with the next batch
            if self.conditional and self.world == 1:
                raise  # vae_train_prop.py:120-126 has no such guard
            ok, err = False, "%s: %s" % (type(e).__name__, e)
        # A non-finite loss skips the batch at every world size - an addition to the reference, which would step
        # NaN gradients into the weights; the flag rides along with the all-rank failure flag (one host wait).
        if not parallel.all_ranks_ok(ok, self.device if self.device.type == "cuda" else None, finite):
            if err is None and ok:
                err = "non-finite loss"
            self.log("Batch %d failed%s" % (self.total_step, "" if err is None else "\n" + err))
            return StepResult(False, error=err)
        self.bucket.all_reduce_mean(force=self.force_collective)
        nn.utils.clip_grad_norm_(self.params, self.sched.clip_norm)
        self.optimizer.step()
        res = StepResult(True, float(loss.detach()), metrics)
        self._after_step(res)
        return res

    def _after_step(self, res: StepResult) -> None:
        s, m = self.sched, res.metrics
        if not self.conditional:
            self.meters += np.array([m["kl_div"], m["word_acc"], m["topo_acc"], m["assm_acc"]])
        if self.total_step % s.print_iter == 0 and self.rank == 0:
            if self.conditional:  # vae_train_prop.py:129-139
                items = ["Loss: %.2f" % res.loss, " Beta: %.5f" % self.beta] + \
                        [" %s: %.3f" % (k, v) for k, v in m.items()] + \
                        ["PNorm: %.3f" % param_norm(self.model), "GNorm: %.3f" % grad_norm(self.model)]
                self.log(",".join(items))
            else:  # vae_train.py:112-132
                mt = self.meters / s.print_iter
                self.log("[%d] Loss: %.3f,Beta: %.3f,KL: %.2f, Word: %.2f, Topo: %.2f, Assm: %.2f, PNorm: %.2f, "
                         "GNorm: %.2f" % (self.total_step, res.loss, self.beta, mt[0], mt[1], mt[2], mt[3],
                                          param_norm(self.model), grad_norm(self.model)))
        if self.total_step % s.print_iter == 0:
            self.meters *= 0
        if self.total_step % s.save_iter == 0:
            self.save("model.iter-%d" % self.total_step)
        if self.total_step % s.anneal_iter == 0:
            self.scheduler.step()
            if self.rank == 0:
                self.log("learning rate: %.6f" % self.scheduler.get_last_lr()[0])
        if self.total_step % s.kl_anneal_iter == 0 and self.total_step >= s.warmup:
            self.beta = min(s.max_beta, self.beta + s.step_beta)

    def save(self, name: str) -> Optional[str]:
        """``torch.save(model.state_dict(), save_dir/name)`` on rank 0 (vae_train.py:134-137, 146)."""
        if self.save_dir is None or self.rank != 0:
            return None
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, name)
        torch.save(self.model.state_dict(), path)
        return path
