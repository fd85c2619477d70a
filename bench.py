#!/usr/bin/env python3
"""Encoder fwd+bwd throughput on synthetic molecular graphs (the second configuration of
BASELINE.json: 1024 molecules of ~40 atoms, <=6 neighbours, MPN depth 3 + tree GRU depth 20,
hidden 450, latent 56, per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = MPN + JTNNEncoder forward (through the latent heads, loss = sum of latent means and
log-variances), backward to every parameter and, for N > 1, one flat-bucket gradient all-reduce
over RCCL.  Every rank owns its own batch (weak scaling); index tables and features are
device-resident before the timed region starts.  One JSON line is printed by rank 0.
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
HBM_KERNELS = ("gru_gate_gather", "gru_bwd_gather", "gru_bwd_pointwise", "csr_gather_sum", "slab_sum",
               "segment_mean", "segment_mean_bwd", "gru_iter0", "copy_rows", "colsum", "transpose", "pack_batch",
               "select_slab_rows", "row_dot")
MFMA_KERNELS = ("gemm_nt", "gemm_tn")
# HBM bytes per launch from the PMC counters cannot be collected from inside this process: they come from the
# committed summary of two rocprofv3 --pmc passes over this same command (see the file's "source" field)
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")


def pmc_traffic():
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            return {k: v["hbm_bytes_per_launch"] for k, v in json.load(f)["kernels"].items()}
    except (OSError, ValueError, KeyError):
        return {}


def build_problem(seed, batch, vocab, device, with_cands=False):
    from fast_jtnn_hip import JTNNVAE, synthetic
    from fast_jtnn_hip import tensorize as T

    trees = synthetic.make_batch(seed, batch, vocab_size=vocab, mean_atoms=40.0, with_cands=with_cands)
    jt, mess_dict = T.jtenc_tensorize(trees)
    mp = T.mpn_tensorize([t.mol for t in trees])
    model = JTNNVAE(synthetic.FakeVocab(vocab), 450, 56, 20, 3)
    synthetic.fill_params(list(model.named_parameters()), seed=0, bias_scale=0.0)  # trainer init rule
    model = model.to(device)
    if with_cands:
        cands, batch_idx = synthetic.collect_candidates(trees)
        model.bench_assm = (trees, (T.jtmpn_tensorize(cands, mess_dict), torch.from_numpy(batch_idx)))
    else:
        # only the encoder path is stepped: candidate scorer and assembly head take no part
        for n, p in model.named_parameters():
            if n.startswith("jtmpn.") or n.startswith("A_assm."):
                p.requires_grad_(False)
    stats = dict(atoms=int(mp[0].shape[0]), directed_bonds=int(mp[1].shape[0]) - 1,
                 tree_nodes=int(jt[0].shape[0]), tree_messages=int(jt[1].shape[0]) - 1,
                 nnz_bgraph=int((mp[3] != 0).sum()), nnz_agraph=int((mp[2] != 0).sum()),
                 nnz_mess_graph=int((jt[3] != 0).sum()))
    return model, jt, mp, stats


def cpu_baseline(seed, vocab, sample_b=256, reps=2):
    """The CPU oracle (the reference's op sequence in PyTorch CPU ops) on a bounded sample of the
    same workload, timed on this host's cores."""
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

    def step():
        m.zero_grad(set_to_none=True)
        mean, logvar = m.encode_latent(jt, mp)
        (mean.sum() + logvar.sum()).backward()

    step()  # warm-up
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        step()
        best = min(best, time.perf_counter() - t0)
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": sample_b / best, "unit": "molecules/s", "cores": cores, "kind": "port", "cpu": cpu_model,
            "sample": "CPU oracle (reference op sequence, torch CPU fp32), fwd+bwd on %d molecules of the same "
                      "generator, best of %d after 1 warm-up" % (sample_b, reps)}


def main():
    if os.environ.get("JTVAE_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["JTVAE_BENCH_WATCHDOG"]), exit=True)
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
    ap.add_argument("--detail", action="store_true", help="print a per-shape GEMM table to stderr")
    ap.add_argument("--mode", default="train", choices=("train", "encode", "train_assm"),
                    help="train: forward + backward (the headline metric); encode: no-grad bulk encoding to "
                         "[B, 900] vectors (latent generation), reported under its own metric name; train_assm: encoders + candidate-graph JTMPN + "
                         "assembly loss + KL, forward + backward (a training step without the tree decoder)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the tree encoder and the graph encoder on two overlapping streams instead of back to "
                         "back on one (faster step; per-kernel durations then include co-running kernels)")
    ap.add_argument("--backend", default="nccl", help="collective backend; gloo lets several ranks rehearse the "
                                                      "N > 1 path on one GPU")
    args = ap.parse_args()

    from fast_jtnn_hip import _capi as K
    from fast_jtnn_hip import parallel

    rank, local_rank, world = parallel.dist_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path in the product")
    dev_index = local_rank % torch.cuda.device_count()  # identity on a node with one GPU per rank
    if args.backend == "nccl" and dev_index != local_rank:
        raise SystemExit("RCCL needs one GPU per rank: %d ranks, %d GPUs" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    parallel.init_process_group(args.backend, device_index=dev_index)

    model, jt, mp, stats = build_problem(seed=args.seed + rank, batch=args.batch, vocab=args.vocab, device=device,
                                        with_cands=args.mode == "train_assm")
    model.overlap_streams = bool(args.overlap)
    bucket = parallel.FlatGradBucket([p for p in model.parameters() if p.requires_grad])

    def train_step():
        bucket.zero()
        mean, logvar = model.encode_latent(jt, mp)
        (mean.sum() + logvar.sum()).backward()
        bucket.all_reduce_mean()

    def encode_step():
        with torch.no_grad():
            model.encode_vectors(jt, mp)

    def train_assm_step():  # jtnn_vae.py:85-103 without the decoder terms
        bucket.zero()
        trees, assm_holder = model.bench_assm
        tree_vecs, tree_mess, mol_vecs = model.encode(jt, mp)
        _, tree_kl = model.rsample(tree_vecs, model.T_mean, model.T_var, eps[0])
        z_mol, mol_kl = model.rsample(mol_vecs, model.G_mean, model.G_var, eps[1])
        assm_loss, _ = model.assm(trees, assm_holder, z_mol, tree_mess)
        (assm_loss + 0.005 * (tree_kl + mol_kl)).backward()
        bucket.all_reduce_mean()

    if args.mode == "train_assm":
        g = torch.Generator(device="cpu").manual_seed(7)
        eps = [torch.randn((args.batch, 28), generator=g).to(device) for _ in range(2)]
        stats["candidates"] = int(model.bench_assm[1][1].shape[0])
    step = {"train": train_step, "encode": encode_step, "train_assm": train_assm_step}[args.mode]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):  # the first call also uploads and caches the index tables
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = args.batch * world * args.steps / elapsed

    out = {
        "metric": {"train": "encoder fwd+bwd molecules/sec", "encode": "encoder no-grad encode molecules/sec",
                   "train_assm": "encoder + candidate scorer + assembly loss fwd+bwd molecules/sec"}[args.mode],
        "value": round(value, 1), "unit": "molecules/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic batch of %d molecular graphs per GPU (~40 atoms, <=6 neighbours): "
                               "MPN depthG=3 + JTNNEncoder/GraphGRU depthT=20, hidden 450, latent 56, "
                               "%s%s" % (args.batch, {"train": "forward + backward", "encode": "forward only (no grad)",
                                                "train_assm": "with JTMPN on the candidate graphs and the assembly "
                                                              "loss, forward + backward"}[args.mode],
                                               " + flat-bucket %s gradient all-reduce" %
                                                         ("RCCL" if args.backend == "nccl" else args.backend)
                                                         if world > 1 and args.mode != "encode" else ""),
                   "molecules_per_gpu": args.batch, "global_batch": args.batch * world, "vocab": args.vocab,
                   "parallelism": "dp%d" % world, "streams": 2 if args.overlap else 1, "grad_bucket_bytes": bucket.nbytes(), **stats},
    }

    if not args.no_kernel_timing and rank != 0:
        step()  # keeps the collective of rank 0's instrumented step matched
    if rank == 0 and not args.no_kernel_timing:
        # one extra, untimed step with every C-ABI launch bracketed by HIP events on its stream
        with K.KernelTimer() as kt:
            step()
        if args.detail:
            for (name, tag), d in sorted(kt.detail().items(), key=lambda kv: -kv[1]["ms"])[:200]:
                tf = d["flops"] / max(d["ms"], 1e-9) / 1e9
                print("%-18s %-34s x%-3d %8.3f ms %7.1f TF" % (name, tag, d["launches"], d["ms"], tf), file=sys.stderr)
        kernels = []
        traffic = pmc_traffic() if args.batch == 1024 and args.mode == "train" else {}  # counters were taken on the default workload
        for name, d in kt.summary().items():
            sec = d["ms"] * 1e-3
            if name in MFMA_KERNELS:
                ach, peak, unit, bound = d["flops"] / sec / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", "mfma"
            else:
                ach, peak, unit, bound = d["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
            kernels.append({"kernel": name, "bound": bound, "launches": d["launches"],
                            "ms_per_step": round(d["ms"], 4), "avg_us": round(1e3 * d["ms"] / d["launches"], 2),
                            "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                            "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                            "traffic": round(traffic[name]) if name in traffic else None})
        kernels.sort(key=lambda k: -k["ms_per_step"])
        dom = kernels[0]
        out["roofline"] = {k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        out["roofline"]["kernel"] = dom["kernel"]
        out["kernels"] = kernels
        out["kernel_ms_sum"] = round(sum(k["ms_per_step"] for k in kernels), 3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(1234, args.vocab, sample_b=args.cpu_sample)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
