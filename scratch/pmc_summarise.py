"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so the read side is doubled.  Output: bytes per launch, averaged over
the launches of each C-ABI entry point (all template variants of its kernels together).
"""
import collections, csv, glob, json, re, sys

# entry point -> kernels it launches; a call runs exactly one of the kernels before "|" (they count the calls)
ENTRY = [("gemm_nt", ("gemm_nt_kernel", "gemm_nt_small_kernel", "gemm_nt_pair_kernel", "gemm_nt_small_pair_kernel", "|")), ("gemm_tn", ("gemm_tn_kernel", "|", "slab_reduce_kernel")),
         ("csr_gather_sum", ("csr_gather_sum_kernel",)), ("gru_gate_gather", ("gru_gate_gather_kernel",)),
         ("gru_bwd_gather", ("gru_bwd_gather_kernel",)), ("gru_bwd_pointwise", ("gru_bwd_pointwise_kernel",)),
         ("gru_iter0", ("gru_iter0_kernel",)), ("segment_mean", ("segment_mean_kernel",)),
         ("segment_mean_bwd", ("segment_mean_bwd_kernel",)), ("select_slab_rows", ("select_slab_rows_kernel",)),
         ("pack_batch", ("pack_batch_kernel",)), ("copy_rows", ("copy_rows_kernel",)), ("colsum", ("colsum_partial_kernel", "colsum_final_kernel"))]


def load(pattern):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].split("<")[0].replace("void ", "").strip()
            tot[n] += float(r["Counter_Value"])
            cnt[n] += 1
    return tot, cnt


def main(root, steps_total):
    F, Fc = load(root + "/pmc_bench_FETCH_SIZE/*/*_counter_collection.csv")
    W, Wc = load(root + "/pmc_bench_WRITE_SIZE/*/*_counter_collection.csv")
    out = {}
    for entry, kernels in ENTRY:
        heads = kernels[:kernels.index("|")] if "|" in kernels else kernels[:1]
        kernels = [k for k in kernels if k != "|"]
        launches = sum(Fc.get(k, 0) for k in heads)
        if not launches:
            continue
        fetch = sum(F.get(k, 0.0) for k in kernels) * 1024 * 2
        write = sum(W.get(k, 0.0) for k in kernels) * 1024
        out[entry] = {"launches_per_step": launches / steps_total, "read_bytes_per_launch": fetch / launches,
                      "write_bytes_per_launch": write / launches, "hbm_bytes_per_launch": (fetch + write) / launches}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                         "--warmup 1 --no-cpu-baseline --no-kernel-timing; read side = 2 x FETCH_SIZE KiB (gfx950 "
                         "tallies 128-byte read requests at 64 bytes), write side = WRITE_SIZE KiB",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]))
