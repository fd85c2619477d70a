#!/usr/bin/env python3
"""Trainer for the unconditional (JTNNVAE) and conditional (JTpropVAE) model with the reference's flag
names and step semantics (fast_molvae/vae_train.py:150-202, fast_molopt/vae_train_prop.py:156-199), one
process per GPU, gradients joined by one RCCL all-reduce per step.

    python train.py --train synthetic:20000 --vocab 385 --save_dir out --batch_size 4096 --epoch 1
    python train.py --train synthetic:20000 --vocab 385 --save_dir out --conditional --train_mode denticity isomer --gpus 8

``--batch_size`` is the GLOBAL batch; every rank takes an equal share of each batch by molecule and builds
the holders of its own share with the native tensorizers.  ``--train synthetic:N[:seed]`` draws N abstract
molecules from the in-package generator; ``--train flat:<file.npz>`` reads a data set in flat form
(``fast_jtnn_hip.flat.save_shard``: atoms, bonds, feature bytes, tree neighbour lists, candidates - turning SMILES
into those arrays needs RDKit and stays with the caller).  ``--vocab`` is a vocabulary file (one entry per line; only its
length is used for synthetic data) or an integer.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jt-vae-tmcinvdes_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", required=True)
    ap.add_argument("--vocab", required=True)
    ap.add_argument("--save_dir", required=True)
    ap.add_argument("--load_epoch", type=int, default=0)
    ap.add_argument("--load_previous_model", action="store_true")
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--conditional", action="store_true", help="JTpropVAE (fast_molopt) instead of JTNNVAE")
    ap.add_argument("--train_mode", nargs="*", default=[], choices=["denticity", "isomer"])
    ap.add_argument("--hidden_size", type=int, default=450)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--latent_size", type=int, default=56)
    ap.add_argument("--depthT", type=int, default=20)
    ap.add_argument("--depthG", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--clip_norm", type=float, default=50.0)
    ap.add_argument("--beta", type=float, default=0.0)
    ap.add_argument("--step_beta", type=float, default=0.002)
    ap.add_argument("--max_beta", type=float, default=1.0)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--epoch", type=int, default=100)
    ap.add_argument("--anneal_rate", type=float, default=0.9)
    ap.add_argument("--anneal_iter", type=int, default=1000)
    ap.add_argument("--kl_anneal_iter", type=int, default=300)
    ap.add_argument("--print_iter", type=int, default=50)
    ap.add_argument("--save_iter", type=int, default=1000)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    return ap.parse_args(argv)


def vocab_size(spec: str) -> int:
    if os.path.exists(spec):
        with open(spec) as f:
            return sum(1 for line in f if line.strip("\r\n "))
    return int(spec)


def flat_batches(spec: str, vocab: int, batch_size: int, rank: int, world: int, epoch: int, conditional: bool):
    """Yields this rank's share of every global batch as (flat shard, molecule positions, property rows).

    ``synthetic:N[:seed]``: every global batch is drawn from the generator (one seed per (batch, rank): a rank never
    materialises another rank's molecules).  ``flat:<file.npz>``: a data set written by ``flat.save_shard``, shuffled
    per epoch with the same permutation on every rank, each rank taking its slice of every global batch."""
    import numpy as np

    from fast_jtnn_hip import flat, parallel, synthetic

    lo, hi = parallel.shard_bounds(batch_size, world, rank)
    if spec.startswith("flat:"):
        shard, props = flat.load_shard(spec[5:])
        if conditional and props is None:
            raise SystemExit("--conditional needs a property table in %s" % spec[5:])
        order = np.random.RandomState(1000 + epoch).permutation(shard.n)
        for b0 in range(0, shard.n - batch_size + 1, batch_size):
            ids = np.sort(order[b0 + lo:b0 + hi])
            yield shard, ids, (props[ids] if conditional else None)
        return
    parts = spec.split(":")
    n_total, seed = int(parts[1]), int(parts[2]) if len(parts) > 2 else 0
    for b0 in range(0, n_total - batch_size + 1, batch_size):
        trees = synthetic.make_batch(seed * 7919 + epoch * 104729 + b0 * 31 + rank, hi - lo, vocab_size=vocab,
                                     with_cands=True)
        props = None
        if conditional:
            rs = np.random.RandomState(b0 + rank)
            props = np.stack([rs.standard_normal(hi - lo), rs.standard_normal(hi - lo), rs.randint(0, 2, hi - lo),
                              rs.randint(0, 3, hi - lo)], axis=1).astype(np.float32)
        yield flat.shard_from_trees(trees, with_cands=True), None, props


def main(argv=None):
    args = parse_args(argv)
    from fast_jtnn_hip import parallel

    parallel.self_launch_if_needed(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    import torch

    from fast_jtnn_hip import JTNNVAE, JTpropVAE, synthetic, trainer

    rank, local_rank, world = parallel.dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: there is no CPU path in the product")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    device = torch.device("cuda", torch.cuda.current_device())
    parallel.init_process_group(args.backend, device_index=device.index)
    if args.batch_size % world:
        raise SystemExit("--batch_size %d is not a multiple of %d ranks" % (args.batch_size, world))
    if not args.train.startswith(("synthetic:", "flat:")):
        raise SystemExit("--train: synthetic:N[:seed] or flat:<file.npz> (fast_jtnn_hip.flat.save_shard; produce it "
                         "from SMILES on a host that has RDKit)")
    V = vocab_size(args.vocab)
    vocab = synthetic.FakeVocab(V)
    if args.conditional:
        model = JTpropVAE(vocab, args.hidden_size, args.latent_size, args.depthT, args.depthG, args.train_mode)
    else:
        model = JTNNVAE(vocab, args.hidden_size, args.latent_size, args.depthT, args.depthG)
    torch.manual_seed(0)  # every rank starts from the same weights
    trainer.init_params(model)
    if args.load_previous_model and args.model_path:
        model.load_state_dict(torch.load(args.model_path, map_location="cpu", weights_only=True))
    elif args.load_epoch > 0:
        model.load_state_dict(torch.load(os.path.join(args.save_dir, "model.epoch-%d" % args.load_epoch),
                                         map_location="cpu", weights_only=True))
    model = model.to(device)
    model.overlap_streams = device.type == "cuda"  # tree and graph encoder on two streams: same results to rounding, shorter step
    if rank == 0:
        print("Model #Params: %dK" % (sum(x.nelement() for x in model.parameters()) / 1000,))
    sched = trainer.Schedule(lr=args.lr, clip_norm=args.clip_norm, beta=args.beta, step_beta=args.step_beta,
                             max_beta=args.max_beta, warmup=args.warmup, anneal_rate=args.anneal_rate,
                             anneal_iter=args.anneal_iter, kl_anneal_iter=args.kl_anneal_iter,
                             print_iter=args.print_iter, save_iter=args.save_iter)
    tr = trainer.Trainer(model, sched, save_dir=args.save_dir, conditional=args.conditional,
                         total_step=args.load_epoch)
    torch.manual_seed(1234 + rank)  # every rank draws its own rsample noise
    import torch as _t

    from fast_jtnn_hip import flatprep
    from fast_jtnn_hip.prefetch import Prefetcher

    def prepare(item):  # native tensorize + index preparation + upload, in a background thread
        shard, ids, props = item
        return flatprep.training_batch(model, shard, ids, props=None if props is None else _t.from_numpy(props))

    for epoch in range(args.epoch):
        # the next batches are tensorized, prepared and uploaded by background threads while this one steps
        with Prefetcher(flat_batches(args.train, V, args.batch_size, rank, world, epoch, args.conditional), prepare,
                        device) as loader:  # closed on any exit
            for batch in loader:
                tr.step(batch)
    tr.save("model.epoch-%d" % tr.total_step)  # vae_train.py:146
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
