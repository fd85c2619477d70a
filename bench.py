#!/usr/bin/env python3
"""Encoder fwd+bwd throughput on synthetic molecular graphs (1024 molecules of ~40 atoms, <=6 neighbours,
MPN depth 3 + tree GRU depth 20, hidden 450, latent 56, per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = MPN + JTNNEncoder forward (through the latent heads, loss = sum of latent means and
log-variances), backward to every parameter and, for N > 1, one flat-bucket gradient all-reduce
over RCCL.  Every rank owns its own batch (weak scaling); index tables and features are
device-resident before the timed region starts.  One JSON line is printed by rank 0.
``python bench.py --gpus N`` without a launcher environment starts the N ranks itself.

Other modes (their own metric names): ``--mode encode`` no-grad bulk encoding (the latent-generation
job's per-GPU step), ``--mode train_assm`` encoders + candidate scorer + assembly loss,
``--mode train_full [--batch 4096]`` the whole JTNNVAE training step (teacher-forced decoder, clip,
Adam), ``--mode train_prop`` the same for the conditional JTpropVAE.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jt-vae-tmcinvdes_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense, spec
HBM_COPY_PEAK_GBS = 6290.0  # what a streaming copy reaches on this part (measured): the practical ceiling
MFMA_KERNELS = ("gemm_nt", "gemm_tn")
KT_STEPS = 5  # serialized steps the per-kernel HIP-event timings are averaged over
# HBM bytes per launch from the PMC counters cannot be collected from inside this process: they come from the
# committed summary of two rocprofv3 --pmc passes over this same command (see the file's "source" field).
# The summary records the kernel sources it was taken with and the launch count of every kernel; it is used
# only while both still match, otherwise "traffic" is null.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
CSRC = os.path.join(ROOT, "jt-vae-tmcinvdes_amd", "csrc")


def csrc_digest():
    """sha256 over the kernel sources: ties a committed counter summary to the kernels it was taken with."""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()


def pmc_traffic(mode, batch, launches):
    """kernel -> HBM bytes per launch replayed from the committed counter summary, or {} if it does not belong
    to these kernels, this workload or these launch counts."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            doc = json.load(f)
        if doc.get("csrc_sha256") != csrc_digest():
            return {}
        mine = [w for w in doc["workloads"] if w["mode"] == mode and w["batch"] == batch]
        if not mine:
            return {}
        out = {}
        for k, v in mine[0]["kernels"].items():
            if k in launches and v.get("launches_per_step") == launches[k]:
                out[k] = v["hbm_bytes_per_launch"]
        return out
    except (OSError, ValueError, KeyError):
        return {}


def build_problem(seed, batch, vocab, device, mode="train"):
    from fast_jtnn_hip import JTNNVAE, JTpropVAE, synthetic
    from fast_jtnn_hip import tensorize as T

    with_cands = mode in ("train_assm", "train_full", "train_prop")
    trees = synthetic.make_batch(seed, batch, vocab_size=vocab, mean_atoms=40.0, with_cands=with_cands)
    jt, mess_dict = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    fake = synthetic.FakeVocab(vocab)
    if mode == "train_prop":
        model = JTpropVAE(fake, 450, 56, 20, 3, ["denticity", "isomer"])
    else:  # the tree decoder takes part in the full training step only
        model = JTNNVAE(fake, 450, 56, 20, 3, decoder=None if mode == "train_full" else False)
    synthetic.fill_params(list(model.named_parameters()), seed=0, bias_scale=0.0)  # trainer init rule
    model = model.to(device)
    x_batch = None
    if with_cands:
        cands, batch_idx = synthetic.collect_candidates(trees)
        cj = (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(batch_idx))
        x_batch = (trees, jt, mp, cj)
        if mode == "train_prop":
            rs = np.random.RandomState(seed)
            props = np.stack([rs.standard_normal(batch), rs.standard_normal(batch), rs.randint(0, 2, batch),
                              rs.randint(0, 3, batch)], axis=1).astype(np.float32)
            x_batch = (trees, torch.from_numpy(props), jt, mp, cj)
    else:
        # only the encoder path is stepped: candidate scorer and assembly head take no part
        for n, p in model.named_parameters():
            if n.startswith("jtmpn.") or n.startswith("A_assm."):
                p.requires_grad_(False)
    stats = dict(atoms=int(mp[0].shape[0]), directed_bonds=int(mp[1].shape[0]) - 1,
                 tree_nodes=int(jt[0].shape[0]), tree_messages=int(jt[1].shape[0]) - 1,
                 nnz_bgraph=int((mp[3] != 0).sum()), nnz_agraph=int((mp[2] != 0).sum()),
                 nnz_mess_graph=int((jt[3] != 0).sum()))
    if with_cands:
        stats["candidates"] = int(batch_idx.shape[0])
    stats.update(tree_shape_stats(jt, 20, all_messages=with_cands))
    return model, jt, mp, x_batch, stats


def tree_shape_stats(jt, depth_t, all_messages):
    """What the tree-GRU schedules make of this batch: the step's cost depends on the depth distribution of the
    junction trees (a message is computed once, in the iteration in which everything it reads has settled)."""
    from fast_jtnn_hip import _capi as K
    from fast_jtnn_hip import graphprep

    mg, ng = jt[3].numpy(), jt[2].numpy()
    nm = mg.shape[0]
    rp, col = graphprep.padded_to_csr(mg, nm)
    sf = K.prep_stable_from(rp, col, 2 ** 30)
    per_level = np.bincount(sf[1:], minlength=2)[1:]  # messages that settle in iteration 1, 2, ...
    roots = np.asarray([st for st, _ in jt[4]], dtype=np.int64)
    seeds = None if all_messages else ng[roots][ng[roots] != 0]
    plan = K.prep_backward_once(rp, col, seeds)
    return {"tree_levels": int(per_level.shape[0]), "messages_settling_per_level": [int(v) for v in per_level],
            "row_updates": int(nm + per_level[1:].sum()),  # iteration 0 touches every row, later ones their level
            "row_updates_dense_schedule": int(nm * depth_t),
            "grad_carrying_messages": int(plan["act"].sum()),
            "grad_carrying_messages_what": "all messages (the candidate scorer reads them)" if all_messages else
            "the messages that flow towards a root (tree_vecs is the only consumer of the tree encoder in this step)",
            "backward_levels": [int(v) for v in np.diff(plan["lev_off"])]}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


ROW_CLASSES = ((40000, ">= 40000 rows"), (2000, "2000 .. 39999 rows"), (0, "< 2000 rows"))
NARROW = "feature-row products (inner or output extent < 128: bound by the rows they move, not by MFMA)"


def rows_breakdown(detail, steps):
    """MFMA kernels by the row count of the launch (from the shape tag the kernel timer files a call under):
    per class the launches per step, their time and the rate they reach."""
    import re

    out = {}
    for (name, tag), d in detail.items():
        if name not in MFMA_KERNELS:
            continue
        m = re.search(r"M=(\d+)(?:\+(\d+))?", tag)
        if not m:
            continue
        rows = max(int(g) for g in m.groups() if g)
        label = next(lab for lo, lab in ROW_CLASSES if rows >= lo)
        nk = re.search(r"N=(\d+) K=(\d+)", tag)
        if nk and min(int(nk.group(1)), int(nk.group(2))) < 128:
            label = NARROW
        c = out.setdefault(name, {}).setdefault(label, dict(launches=0, ms=0.0, flops=0))
        c["launches"] += d["launches"]
        c["ms"] += d["ms"]
        c["flops"] += d["flops"]
    res = {}
    for name, classes in out.items():
        res[name] = [dict(rows=label, launches=c["launches"] // steps, ms_per_step=round(c["ms"] / steps, 4),
                          achieved=round(c["flops"] / max(c["ms"], 1e-9) / 1e9, 2), unit="TFLOP/s")
                     for label in [lab for _, lab in ROW_CLASSES] + [NARROW] for c in [classes.get(label)] if c]
    return res


def cpu_baseline(seed, vocab, sample_b=256, reps=2, grad=True, keep=None):
    """The CPU oracle (the reference's op sequence in PyTorch CPU ops) on a bounded sample of the
    same workload, timed on this host's cores.  ``keep`` (a dict) receives the sample's holders and the
    oracle's results, for the parity check of the HIP path on the same sample."""
    from fast_jtnn_hip import synthetic
    from fast_jtnn_hip import tensorize as T
    from oracle.ref_encoder import OracleEncoder

    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    trees = synthetic.make_batch(seed + 1000, sample_b, vocab_size=vocab, mean_atoms=40.0)
    jt, _ = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    m = OracleEncoder(vocab, 450, 56, 20, 3)
    synthetic.fill_params(list(m.named_parameters()), seed=0, bias_scale=0.0)
    res = {}

    def step():
        if not grad:
            with torch.no_grad():
                res["mean"], res["logvar"] = m.encode_latent(jt, mp)
            return
        m.zero_grad(set_to_none=True)
        res["mean"], res["logvar"] = m.encode_latent(jt, mp)
        (res["mean"].sum() + res["logvar"].sum()).backward()

    step()  # warm-up
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    if keep is not None:
        keep.update(jt=jt, mp=mp, mean=res["mean"].detach().numpy(), logvar=res["logvar"].detach().numpy(),
                    grads={n: p.grad.numpy() for n, p in m.named_parameters() if p.grad is not None})
    return {"value": sample_b / best, "unit": "molecules/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": "CPU oracle (reference op sequence, torch CPU fp32), %s on %d molecules of the same "
                      "generator, best of %d after 1 warm-up" % ("fwd+bwd" if grad else "forward only (no grad)",
                                                                 sample_b, reps)}


def parity_on_sample(keep, vocab, device, grad=True):
    """The HIP path on the CPU baseline's sample against the oracle's results there: scaled max error of
    the latent means / log-variances, Frobenius error of every gradient tensor; bound 1e-4."""
    from fast_jtnn_hip import JTNNVAE, synthetic

    m = JTNNVAE(synthetic.FakeVocab(vocab), 450, 56, 20, 3, decoder=False)
    synthetic.fill_params(list(m.named_parameters()), seed=0, bias_scale=0.0)
    m = m.to(device)

    def rel(a, b):
        return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))

    if grad:
        mean, logvar = m.encode_latent(keep["jt"], keep["mp"])
        (mean.sum() + logvar.sum()).backward()
    else:
        with torch.no_grad():
            mean, logvar = m.encode_latent(keep["jt"], keep["mp"])
    max_rel = max(rel(mean.detach().cpu().numpy(), keep["mean"]), rel(logvar.detach().cpu().numpy(), keep["logvar"]))
    out = {"max_rel": max_rel, "tol": 1e-4}
    ok = max_rel < 1e-4
    if grad:
        worst = 0.0
        for n, p in m.named_parameters():
            if n in keep["grads"] and p.grad is not None:
                ref = keep["grads"][n].astype(np.float64)
                worst = max(worst, float(np.linalg.norm(p.grad.cpu().numpy() - ref) / max(np.linalg.norm(ref), 1e-30)))
        out["grad_max_fro"] = worst
        ok = ok and worst < 1e-4
    out["ok"] = bool(ok)
    out["what"] = ("HIP path vs CPU oracle on the CPU baseline's sample: latent mean / log-var (max error over the "
                   "tensor's largest entry)" + (", every gradient tensor (Frobenius)" if grad else ""))
    return out


def fresh_holder_rate(model, args, device, rank, step_on):
    """Throughput when every step gets holders it has never seen (what a data loader hands over): the
    index preparation (CSR, transposes, GRU schedules) and the host-to-device copies are then inside the
    step.  The loader's own work (tensorize) is outside, as it is in the reference."""
    from fast_jtnn_hip import synthetic
    from fast_jtnn_hip import tensorize as T

    from fast_jtnn_hip.prefetch import Prefetcher, warm_encoder_holders

    k = int(os.environ.get("JTVAE_FRESH_BATCHES", "12"))
    batches, tree_sets = [], []
    for i in range(k):
        trees = synthetic.make_batch(args.seed + 500 + 13 * i + rank, args.batch, vocab_size=args.vocab, mean_atoms=40.0)
        batches.append((T.jtenc_tensorize(trees)[0], T.mpn_tensorize([t.mol for t in trees])))
        tree_sets.append(trees)

    def unseen(b):  # new tensor objects: nothing prepared is attached to them
        return tuple(tuple(x.clone() if isinstance(x, torch.Tensor) else x for x in h) for h in b)

    issued = [0.0, 0.0]

    def run(source):
        it = iter(source)
        step_on(*next(it))  # allocator warm-up on these shapes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        waited = 0.0
        while True:
            w0 = time.perf_counter()
            try:
                jt, mp = next(it)
            except StopIteration:
                break
            waited += time.perf_counter() - w0
            step_on(jt, mp)
            n += 1
        issued[0] = (time.perf_counter() - t0) / n  # the loop alone: drawing batches and issuing their steps
        issued[1] = waited / n                      # of which: waiting for the next batch
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    dt = run([unseen(b) for b in batches])
    train = step_on.__name__ == "train_on"
    pf = Prefetcher([unseen(b) for b in batches * 2], lambda b: warm_encoder_holders(model, b[0], b[1], train=train),
                    device)
    dt_pf = run(pf)
    pf.close()
    # The same batches drawn from flat storage (the data set at rest: atoms, bonds, feature bytes, tree
    # neighbour lists): now the tensorize work itself - the reference's per-batch datautils.tensorize - is
    # inside the clock as well, done by the native builders straight into upload staging memory.
    from fast_jtnn_hip import flat, flatprep
    from fast_jtnn_hip.prefetch import default_workers

    all_trees = [t for ts in tree_sets for t in ts]
    synthetic.set_batch_nodeID(all_trees)  # node ids = positions in the whole set
    shard = flat.shard_from_trees(all_trees)
    ids = [np.arange(i * args.batch, (i + 1) * args.batch) for i in range(k)]
    threads = flat.host_threads()
    build = lambda sel: flatprep.encoder_batch(model, shard, sel, train=train, threads=threads)  # noqa: E731
    build(ids[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for sel in ids[:4]:
        build(sel)
    host_ms = 1e3 * (time.perf_counter() - t0) / 4  # host time of tensorize + preparation + enqueueing the copy
    torch.cuda.synchronize()
    dt_flat = run(build(sel) for sel in ids)
    workers = default_workers()
    pf = Prefetcher(ids * 2, build, device, workers=workers)
    dt_flat_pf = run(pf)
    issue_flat_pf, wait_flat_pf = issued
    pf.close()
    return {"value_fresh_holders": round(args.batch / dt, 1), "ms_per_step_fresh_holders": round(1e3 * dt, 3),
            "value_fresh_holders_prefetched": round(args.batch / dt_pf, 1),
            "ms_per_step_fresh_holders_prefetched": round(1e3 * dt_pf, 3),
            "value_from_flat_storage": round(args.batch / dt_flat, 1),
            "value_from_flat_storage_prefetched": round(args.batch / dt_flat_pf, 1),
            "ms_per_step_from_flat_storage_prefetched": round(1e3 * dt_flat_pf, 3),
            "host_loop_ms_per_step_from_flat_storage_prefetched": round(1e3 * issue_flat_pf, 3),
            "host_wait_for_batch_ms_per_step_from_flat_storage_prefetched": round(1e3 * wait_flat_pf, 3),
            "tensorize_and_prepare_host_ms_per_batch": round(host_ms, 3),
            "host_threads": {"native_builder_threads": threads, "prefetch_workers": workers,
                             "cpu_count": os.cpu_count()},
            "what": "same step on batches never seen before. fresh_holders: holders come as the reference's tables "
                    "(tensorize excluded), per-batch index preparation and host-to-device copies included - in the "
                    "stepping thread, and prepared ahead by the background threads of fast_jtnn_hip.prefetch. "
                    "from_flat_storage: batches are drawn from flat arrays (fast_jtnn_hip.flat) and TENSORIZE IS "
                    "INCLUDED: native builders + preparation + copies, in the stepping thread and prefetched"}


METRIC = {"train": "encoder fwd+bwd molecules/sec", "encode": "encoder no-grad encode molecules/sec",
          "train_assm": "encoder + candidate scorer + assembly loss fwd+bwd molecules/sec",
          "train_full": "JTNNVAE training step molecules/sec",
          "train_prop": "JTpropVAE training step molecules/sec"}
WHAT = {"train": "forward + backward", "encode": "forward only (no grad)",
        "train_assm": "with JTMPN on the candidate graphs and the assembly loss, forward + backward",
        "train_full": "whole JTNNVAE.forward (JTMPN on the candidate graphs, assembly loss, teacher-forced tree "
                      "decoder, KL) + backward + gradient clip + Adam step",
        "train_prop": "whole JTpropVAE.forward (as JTNNVAE plus property, denticity and isomer heads) + backward + "
                      "gradient clip + Adam step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU")
    ap.add_argument("--vocab", type=int, default=385)
    ap.add_argument("--seed", type=int, default=1234, help="seed of the synthetic batch (rank r uses seed + r)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256, help="molecules in the CPU baseline's sample")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fresh-holders", action="store_true")
    ap.add_argument("--detail", action="store_true", help="print a per-shape GEMM table to stderr")
    ap.add_argument("--mode", default="train", choices=tuple(METRIC),
                    help="train: encoder forward + backward (the headline metric); the other modes report "
                         "under their own metric names (see the module docstring)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the tree encoder and the graph encoder back to back on one stream instead of on two "
                         "overlapping ones (results agree to rounding; bit for bit with JTNNEncoder.crowded pinned)")
    ap.add_argument("--backend", default="nccl", help="collective backend; gloo lets several ranks rehearse the "
                                                      "N > 1 path on one GPU")
    args = ap.parse_args()

    from fast_jtnn_hip import parallel

    # N > 1 without a launcher: start the ranks as children before this process touches the GPU
    parallel.self_launch_if_needed(os.path.abspath(__file__), sys.argv[1:], args.gpus, json_only=True)
    if os.environ.get("JTVAE_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["JTVAE_BENCH_WATCHDOG"]), exit=True)

    from fast_jtnn_hip import _capi as K
    from fast_jtnn_hip import trainer

    rank, local_rank, world = parallel.dist_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in the product")
    dev_index = local_rank % torch.cuda.device_count()  # identity on a node with one GPU per rank
    if args.backend == "nccl" and dev_index != local_rank:
        raise SystemExit("RCCL needs one GPU per rank: %d ranks, %d GPUs" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    parallel.init_process_group(args.backend, device_index=dev_index)

    mode = args.mode
    model, jt, mp, x_batch, stats = build_problem(seed=args.seed + rank, batch=args.batch, vocab=args.vocab,
                                                  device=device, mode=mode)
    overlap = not args.no_overlap
    model.overlap_streams = overlap
    tr = None
    if mode in ("train_full", "train_prop"):
        tr = trainer.Trainer(model, trainer.Schedule(beta=0.005, print_iter=10 ** 9, save_iter=10 ** 9,
                                                     anneal_iter=10 ** 9, kl_anneal_iter=10 ** 9),
                             conditional=mode == "train_prop", log=lambda *a: None)
        bucket = tr.bucket
        torch.manual_seed(7 + rank)
    else:
        bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])

    def train_on(jt_, mp_):
        bucket.zero()
        mean, logvar = model.encode_latent(jt_, mp_)
        (mean.sum() + logvar.sum()).backward()
        bucket.all_reduce_mean()

    def encode_on(jt_, mp_):
        with torch.no_grad():
            model.encode_vectors(jt_, mp_)

    def train_assm_step():  # jtnn_vae.py:85-103 without the decoder terms
        bucket.zero()
        trees, _, _, assm_holder = x_batch
        tree_vecs, tree_mess, mol_vecs = model.encode(jt, mp)
        _, tree_kl = model.rsample(tree_vecs, model.T_mean, model.T_var, eps[0])
        z_mol, mol_kl = model.rsample(mol_vecs, model.G_mean, model.G_var, eps[1])
        assm_loss, _ = model.assm(trees, assm_holder, z_mol, tree_mess)
        (assm_loss + 0.005 * (tree_kl + mol_kl)).backward()
        bucket.all_reduce_mean()

    def full_step():  # one iteration of the reference's training loop (vae_train.py:99-104)
        res = tr.step(x_batch)
        if not res.ok:
            raise RuntimeError("training step failed: %s" % res.error)

    if mode == "train_assm":
        g = torch.Generator(device="cpu").manual_seed(7)
        eps = [torch.randn((args.batch, 28), generator=g).to(device) for _ in range(2)]
    step = {"train": lambda: train_on(jt, mp), "encode": lambda: encode_on(jt, mp), "train_assm": train_assm_step,
            "train_full": full_step, "train_prop": full_step}[mode]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):  # the first call also uploads and caches the index tables
        step()
    # EXACTLY --steps steps per timed window, barrier + synchronize on both sides, MAX over ranks; the window is
    # short (a tenth of a second), so three windows are taken and the median one is reported, with the spread
    windows, issues = [], []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        issued = time.perf_counter() - t0  # the host has queued every launch; the GPU may still be running them
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        windows.append(elapsed)
        issues.append(issued)
    mid = sorted(range(3), key=lambda i: windows[i])[1]
    elapsed, issued = windows[mid], issues[mid]
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.batch * world * args.steps / elapsed

    out = {
        "metric": METRIC[mode], "value": round(value, 1), "unit": "molecules/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "timed_windows": {"what": "three windows of --steps steps each; value / ms_per_step are the median window",
                          "ms_per_step": [round(1e3 * w / args.steps, 3) for w in windows],
                          "value_min": round(args.batch * world * args.steps / max(windows), 1),
                          "value_max": round(args.batch * world * args.steps / min(windows), 1)},
        "host_issue_ms_per_step": round(1e3 * issued / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic batch of %d molecular graphs per GPU (~40 atoms, <=6 neighbours): "
                               "MPN depthG=3 + JTNNEncoder/GraphGRU depthT=20, hidden 450, latent 56, %s%s"
                               % (args.batch, WHAT[mode],
                                  " + flat-bucket %s gradient all-reduce" % ("RCCL" if args.backend == "nccl"
                                                                             else args.backend)
                                  if world > 1 and mode != "encode" else ""),
                   "molecules_per_gpu": args.batch, "global_batch": args.batch * world, "vocab": args.vocab,
                   "parallelism": "dp%d" % world, "streams": 2 if overlap else 1,
                   "grad_bucket_bytes": bucket.nbytes(), **stats},
    }

    if not args.no_kernel_timing:
        # Extra, untimed steps with every C-ABI launch bracketed by HIP events on its stream, the two
        # encoders back to back on one stream so that no kernel's duration includes a co-running one.
        # Every rank runs it (the collective inside must stay matched); rank 0 reports.
        model.overlap_streams = False
        step()  # one serialized step unmeasured: the allocator and the clocks settle into this schedule
        with K.KernelTimer() as kt:
            for _ in range(KT_STEPS):
                step()
        model.overlap_streams = overlap
    if rank == 0 and not args.no_kernel_timing:
        if args.detail:
            for (name, tag), d in sorted(kt.detail().items(), key=lambda kv: -kv[1]["ms"])[:200]:
                tf = d["flops"] / max(d["ms"], 1e-9) / 1e9
                print("%-18s %-34s x%-3d %8.3f ms %7.1f TF" % (name, tag, d["launches"] // KT_STEPS,
                                                              d["ms"] / KT_STEPS, tf), file=sys.stderr)
        kernels = []
        summary = kt.summary()
        by_rows = rows_breakdown(kt.detail(), KT_STEPS)
        for d in summary.values():  # per step: every step issues the same launches
            assert d["launches"] % KT_STEPS == 0, summary
            d["launches"] //= KT_STEPS
            d["ms"] /= KT_STEPS
            d["bytes"] //= KT_STEPS
            d["flops"] //= KT_STEPS
        traffic = pmc_traffic(mode, args.batch, {k: d["launches"] for k, d in summary.items()})
        for name, d in summary.items():
            sec = d["ms"] * 1e-3
            row = {"kernel": name, "launches": d["launches"], "ms_per_step": round(d["ms"], 4),
                   "avg_us": round(1e3 * d["ms"] / d["launches"], 2)}
            if name in MFMA_KERNELS:
                ach = d["flops"] / sec / 1e12
                row.update(bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                           frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                           algorithmic_flops_per_launch=round(d["flops"] / d["launches"]))
            else:
                ach = d["bytes"] / sec / 1e9
                row.update(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=round(ach / HBM_PEAK_GBS, 4), frac_of_copy_peak=round(ach / HBM_COPY_PEAK_GBS, 4))
            row["algorithmic_bytes_per_launch"] = round(d["bytes"] / d["launches"])
            if name in by_rows:
                row["by_rows"] = by_rows[name]  # the average above is over launches of very different sizes
            row["traffic"] = round(traffic[name]) if name in traffic else None
            kernels.append(row)
        kernels.sort(key=lambda k: -k["ms_per_step"])
        dom = kernels[0]
        out["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        out["roofline"]["kernel"] = dom["kernel"]
        out["roofline"]["traffic_source"] = ("replayed from %s (rocprofv3 --pmc passes over this command; kernel "
                                             "sources and launch counts match)" % os.path.relpath(PMC_TRAFFIC_FILE, ROOT)
                                             if dom["traffic"] is not None else None)
        if dom["bound"] == "mfma":
            from fast_jtnn_hip import _capi as K
            split = bool(K.lib.jtvae_set_split_products(-1))
            out["roofline"]["peak_what"] = (
                "dense peak of the fp32 matrix instruction, the form every product below ~10 000 rows runs in"
                + ("; the larger launches run as three-way bf16 splits of the fp32 operands on the bf16 matrix "
                   "instruction with fp32 accumulation (fp32-accurate; ceiling 2500 / 6 = 417 TFLOP/s "
                   "fp32-equivalent, 323 measured with nothing but matrix instructions), so a launch class in "
                   "by_rows can exceed this figure" if split else ""))
            out["roofline"]["large_products"] = "split-bf16" if split else "fp32"
        out["kernels"] = kernels
        out["launches_per_step"] = sum(k["launches"] for k in kernels)
        # HIP-event pairs around back-to-back launches overlap by the event latency, so this sum can exceed
        # ms_per_step slightly; with two overlapping streams in the timed steps it exceeds it by design
        out["kernel_ms_sum_serialized"] = round(sum(k["ms_per_step"] for k in kernels), 3)
    if mode in ("train", "encode") and not args.no_fresh_holders:
        fresh = fresh_holder_rate(model, args, device, rank, train_on if mode == "train" else encode_on)
        if rank == 0:
            out["fresh_holders"] = fresh
    if rank == 0 and world == 1 and not args.no_cpu_baseline and mode in ("train", "encode"):
        keep = {}
        out["cpu_baseline"] = cpu_baseline(1234, args.vocab, sample_b=args.cpu_sample, grad=mode == "train", keep=keep)
        out["parity"] = parity_on_sample(keep, args.vocab, device, grad=mode == "train")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
