"""Per-kernel counter summaries from rocprofv3 --pmc passes over bench.py.

  python3 scratch/pmc_summarise.py traffic ROOT MODE BATCH STEPS_TOTAL OUT.json
      ROOT/pmc_FETCH_SIZE, ROOT/pmc_WRITE_SIZE -> HBM bytes per launch of every C-ABI entry point, merged
      into OUT.json under (mode, batch).  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
      the 128-byte requests of wide coalesced reads at 64 bytes, so the read side is doubled.
  python3 scratch/pmc_summarise.py sq ROOT MODE BATCH STEPS_TOTAL OUT.json
      ROOT/pmc_SQ -> wave-cycle split, MFMA busy cycles and LDS bank conflicts per kernel.

The file written carries the sha256 of the kernel sources it was taken with (bench.csrc_digest); bench.py
replays the traffic figures only while that, the workload and the launch counts still match.
"""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

# entry point (the name bench.py's kernel timer files a call under) -> kernels it launches; a call runs exactly
# one of the kernels before "|" (they count the calls), and possibly the ones after it
ENTRY = [("gemm_nt", ("gemm_nt_kernel", "gemm_nt_split_kernel", "gemm_nt_small_kernel", "gemm_nt_pair_kernel", "gemm_nt_small_pair_kernel",
                      "gru_update_kernel", "gru_update_small_kernel", "|")),
         ("gemm_tn", ("gemm_tn_kernel", "|", "slab_reduce_kernel")),
         ("csr_gather_sum", ("csr_gather_sum_kernel",)), ("gru_gate_gather", ("gru_gate_gather_kernel",)),
         ("gru_bwd_gather", ("gru_bwd_gather_kernel",)), ("gru_bwd_pointwise", ("gru_bwd_pointwise_kernel",)),
         ("gru_iter0", ("gru_iter0_kernel",)), ("segment_mean", ("segment_mean_kernel",)),
         ("segment_mean_bwd", ("segment_mean_bwd_kernel",)), ("select_slab_rows", ("select_slab_rows_kernel",)),
         ("pack_batch", ("pack_batch_kernel",)), ("copy_rows", ("copy_rows_kernel",)),
         ("segment_ce", ("segment_ce_kernel",)), ("row_dot", ("row_dot_kernel",)),
         ("row_dot_bwd", ("row_dot_bwd_kernel",))]


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].split("<")[0].replace("void ", "").strip()


def load(pattern):
    """counter -> kernel -> (sum of values, dispatches)"""
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            n = short(r["Kernel_Name"])
            tot[r["Counter_Name"]][n] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]][n] += 1
    return tot, cnt


def merge(path, sha, mode, batch, payload, source):
    doc = {}
    if os.path.exists(path):
        with open(path) as f:
            doc = json.load(f)
    if doc.get("csrc_sha256") != sha:  # figures from other kernel sources do not mix with these
        doc = {"csrc_sha256": sha, "source": source, "workloads": []}
    doc["workloads"] = [w for w in doc["workloads"] if (w["mode"], w["batch"]) != (mode, batch)]
    doc["workloads"].append({"mode": mode, "batch": batch, "kernels": payload})
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")


def traffic(root, steps_total):
    F, Fc = load(root + "/pmc_FETCH_SIZE/*/*_counter_collection.csv")
    W, _ = load(root + "/pmc_WRITE_SIZE/*/*_counter_collection.csv")
    F, Fc, W = F["FETCH_SIZE"], Fc["FETCH_SIZE"], W["WRITE_SIZE"]
    out = {}
    for entry, kernels in ENTRY:
        heads = kernels[:kernels.index("|")] if "|" in kernels else kernels[:1]
        kernels = [k for k in kernels if k != "|"]
        launches = sum(Fc.get(k, 0) for k in heads)
        if not launches:
            continue
        fetch = sum(F.get(k, 0.0) for k in kernels) * 1024 * 2
        write = sum(W.get(k, 0.0) for k in kernels) * 1024
        per_step = launches / steps_total
        out[entry] = {"launches_per_step": int(per_step) if per_step == int(per_step) else per_step,
                      "read_bytes_per_launch": fetch / launches, "write_bytes_per_launch": write / launches,
                      "hbm_bytes_per_launch": (fetch + write) / launches}
    return out, ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --mode MODE "
                 "--batch BATCH --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-fresh-holders; read "
                 "side = 2 x FETCH_SIZE KiB (gfx950 tallies 128-byte read requests at 64 bytes), write side = "
                 "WRITE_SIZE KiB; averaged over all launches of an entry point's kernels")


def sq(root, steps_total):
    T, C = load(root + "/pmc_SQ/*/*_counter_collection.csv")
    ours = {k for _, ks in ENTRY for k in ks if k != "|"}  # the library's kernels; torch's own are left out
    kernels = sorted({k for c in T.values() for k in c} & ours)
    out = {}
    for k in kernels:
        v = {c: T[c].get(k, 0.0) for c in T}
        wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        row = {"dispatches_per_step": C[next(iter(C))].get(k, 0) / steps_total, "counters": v,
               "wait_frac": v.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_frac": v.get("SQ_WAIT_INST_ANY", 0.0) / wc,
               "active_frac": v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        if v.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
        if v.get("SQ_BUSY_CYCLES"):
            row["mfma_busy_per_sq_busy_cycle"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / v["SQ_BUSY_CYCLES"]
        out[k] = row
    return out, ("rocprofv3 --pmc <8 SQ counters> -- python3 bench.py (one pass, same flags as the traffic passes); "
                 "sums over all dispatches of a kernel; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count "
                 "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles; wait + issue_stall + active ~ 1")


if __name__ == "__main__":
    kind, root, mode, batch, steps_total, dest = sys.argv[1:7]
    import bench

    payload, source = (traffic if kind == "traffic" else sq)(root, float(steps_total))
    merge(dest, bench.csrc_digest(), mode, int(batch), payload, source)
    print(kind, mode, batch, "->", dest, ":", len(payload), "kernels")
