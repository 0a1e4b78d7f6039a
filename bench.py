#!/usr/bin/env python3
"""Benchmark of the hot path: source queries/s over fit + kneighbors (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c2_mp|c3s]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one full `fit(source, target)` + `kneighbors(k)` over synthetic embeddings that are already resident in
HBM (the reference's `rng.rand` data, float32).  Default workload = BASELINE.json configs[1]:
100k x 100k, d=128, euclidean, k=10, hubness=None.  With N > 1 every rank owns a 100k-row source shard (weak
scaling); the target lives on rank 0 and is RCCL-broadcast inside `fit`.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("KIEZ_AMD_WITH_TORCH", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)

WORKLOADS = {
    # name: (n_source_per_gpu, n_target, d, metric, K, k, hubness, hubness_kwargs, description)
    "c1": (100_000, 100_000, 128, "euclidean", 10, 10, None, {},
           "C1: 100k x 100k, d=128, euclidean, k=10, hubness=None"),
    "c2": (100_000, 100_000, 128, "euclidean", 10, 10, "CSLS", {},
           "C2: 100k x 100k, d=128, euclidean, k=10, hubness=CSLS"),
    "c3s": (100_000, 100_000, 200, "cosine", 50, 50, "MutualProximity", {"method": "empiric"},
            "C3 (scaled to 100k x 100k): d=200, cosine, k=50, MutualProximity empiric"),
    "c3": (500_000, 500_000, 200, "cosine", 50, 50, "MutualProximity", {"method": "empiric"},
           "C3: 500k x 500k, d=200, cosine, k=50, MutualProximity empiric"),
    "c4s": (250_000, 1_000_000, 300, "euclidean", 10, 10, "CSLS", {},
            "C4 per-GPU share: 250k source rows x 1M target, d=300, k=10, CSLS"),
    "ns": (250_000, 1_000_000, 200, "euclidean", 10, 10, "CSLS", {},
           "north-star target shape, per-GPU share: 250k source rows x 1M target, d=200, k=10, CSLS"),
    "c1g": (100_000, 100_000, 128, "euclidean", 10, 10, None, {},
            "C1 on gaussian data (rng.randn) for contrast with uniform: 100k x 100k, d=128, euclidean, k=10, hubness=None"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, 'Peak FP32 (matrix)'
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, 'Peak BF16/FP16 MFMA ~2.5 PF dense'


def cpu_baseline(source, target, metric, k, budget_rows=10_000):
    """CPU leg, timed on this host's cores on a bounded query-row sample against the FULL index:
    value       = scikit-learn's brute-force kNN, i.e. the third-party routine the reference's SklearnNN._kneighbors
                  delegates to (kiez/neighbors/exact/sklearn_nearest_neighbors.py:98-101) — the reference's real CPU path;
    port_value  = the oracle's numpy restatement of the same algorithm (oracle/kiez_oracle.py: knn_exact)."""
    from oracle import kiez_oracle as O
    rows = min(budget_rows, len(source))
    q = source[:rows]
    t0 = time.perf_counter()
    O.knn_exact(q, target, k, metric)
    t_oracle = time.perf_counter() - t0
    out = {"value": rows / t_oracle, "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"hubness=None forward pass: first {rows} source rows x all {len(target)} target rows, d={source.shape[1]}, "
                     f"{metric}, k={k}",
           "port_value": rows / t_oracle, "port_seconds": t_oracle}
    try:
        from sklearn.neighbors import NearestNeighbors
        nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric=metric).fit(target)
        nn.kneighbors(q[:256])  # thread-pool warm-up
        t0 = time.perf_counter()
        nn.kneighbors(q)
        t_sk = time.perf_counter() - t0
        out.update(value=rows / t_sk, kind="reference", seconds=t_sk,
                   note="value = sklearn.neighbors.NearestNeighbors(algorithm='brute').kneighbors (what kiez's SklearnNN calls); "
                        "port_value = oracle/kiez_oracle.py knn_exact (numpy, mostly single-threaded selection)")
    except Exception as e:  # pragma: no cover
        out["note"] = f"sklearn timing failed ({e}); value = oracle port"
    return out


def main():
    # Keep stdout clean for the ONE JSON line: librccl prints a start-up banner to fd 1 when the communicator is created.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c1", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify 2000 rows of the result against the oracle (default: 256 rows, hubness=None only)")
    ap.add_argument("--no-check", action="store_true", help="skip the default oracle spot check")
    ap.add_argument("--opt", action="append", default=[], help="context option name=value (tuning experiments)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    import torch.distributed as dist
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # started through torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from kiez_amd.distributed import Comm, HipEngine, ShardedKiez
    n_s, n_t, d, metric, K, k, hub, hub_kw, desc = WORKLOADS[args.workload]
    eng = HipEngine(local_rank)
    comm = Comm()
    for o in args.opt:
        name, val = o.split("=")
        eng.ctx.set_option(name, float(val))

    # synthetic data in the reference's style (kiez/kiez.py:50-52): rng.rand, source first, then target
    rng = np.random.RandomState(0 if rank == 0 else 1000 + rank)
    gen = rng.randn if args.workload.endswith("g") else rng.rand
    source_h = gen(n_s, d).astype(np.float32)
    target_h = gen(n_t, d).astype(np.float32) if rank == 0 else None
    source = eng.to_engine(source_h)
    target = eng.to_engine(target_h) if rank == 0 else None
    eng.sync()

    sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=hub_kw,
                     engine=eng, comm=comm)
    knn_log = []
    orig_knn = eng.knn

    def logged_knn(qm, q_begin, q_count, im, kk, exclude_self):
        out = orig_knn(qm, q_begin, q_count, im, kk, exclude_self)
        knn_log.append((q_count, im.shape[0], dict(eng.last_stats)))
        return out

    eng.knn = logged_knn

    def step():
        sk.fit(source, target)
        return sk.kneighbors(k)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    knn_log.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.cpu()[0])

    # roofline of the dominant kernel (fused distance + top-k): algorithmic flops 2*n_q*n_i*d per launch
    flops = sum(2.0 * q * n * d for q, n, _ in knn_log)
    kernel_s = sum(st["main_kernel_ms"] for _, _, st in knn_log) * 1e-3
    n_launch = len(knn_log)
    achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
    # which fused kernel did (most of) the work: split-bf16 first pass (3 bf16 MFMA products per multiply-add) or float32 MFMA
    tier_ms = {t: sum(st["main_kernel_ms"] for _, _, st in knn_log if st.get("first_pass") == t) for t in (0, 1, 2)}
    tier = max(tier_ms, key=tier_ms.get)       # 0 float32 operands, 1 split-bf16 (3 products), 2 fp16 (1 product)
    tier_bf = tier != 0                        # a 16-bit MFMA tier did the work: price against the bf16/fp16 dense peak
    products = {0: 1, 1: 3, 2: 1}[tier]
    tier_name = {0: "f32", 1: "bf16x2", 2: "f16"}[tier]
    peak = PEAK_BF16_MFMA_TFLOPS if tier_bf else PEAK_F32_MFMA_TFLOPS
    traffic = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"   # HBM bytes per launch from separate rocprofv3 --pmc passes
    if pmc.exists():
        try:
            key = args.workload + "_" + tier_name
            traffic = json.loads(pmc.read_text()).get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    fallback_rows = sum(st["n_fallback_rows"] for _, _, st in knn_log)

    check = None
    do_check = args.check or (not args.no_check and hub is None and world == 1 and n_t <= 200_000)
    if do_check and rank == 0:
        from oracle import kiez_oracle as O
        rows = 2000 if args.check else 256
        dd, ii = res
        od, oi = O.kiez_pipeline(source_h, target_h, K, k, metric, 2, hub, hub_kw, query_rows=rows)
        got_i = ii[:len(oi)].cpu().numpy()
        check = {"rows": int(len(oi)), "index_rows_identical": int((got_i == oi).all(axis=1).sum()),
                 "recall_at_k": float(np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(got_i, oi)])),
                 "max_rel_dist_err": float(np.max(np.abs(dd[:len(od)].cpu().numpy() - od) / np.maximum(np.abs(od), 1e-12)))}

    if rank == 0:
        total_queries = n_s * world * args.steps
        line = {
            "metric": "source queries/sec (fit+kneighbors) + recall@k vs reference, 1/2/4/8 GPU",  # BASELINE.json's metric
            "recall_at_k": (check or {}).get("recall_at_k"),   # against the oracle on a row sample (None if not checked)
            "value": total_queries / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": tier_name,
            "data": "synthetic",
            "config": {"workload": desc, "n_source_per_gpu": n_s, "n_target": n_t, "d": d, "metric": metric,
                       "n_candidates": K, "k": k, "hubness": hub, "hubness_kwargs": hub_kw,
                       "inputs": "float32 rng.rand, resident in HBM; results left in HBM",
                       "parallelism": f"source row-sharded x{world}, target replicated (RCCL broadcast)"},
            "roofline": {"bound": "mfma",
                         "kernel": {2: "kz_knn_cand_h_kernel (fp16 MFMA 32x32x16 on centred operands, 1 product per multiply-add, fused distance+top-k)",
                                    1: "kz_knn_cand_bf_kernel (split-bf16 MFMA 32x32x16, 3 products per multiply-add, fused distance+top-k)",
                                    0: "kz_knn_cand_kernel (fp32 MFMA 32x32x2 fused distance+top-k)"}[tier],
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "launches": n_launch, "avg_launch_ms": kernel_s / max(n_launch, 1) * 1e3,
                         "algorithmic_flop_per_launch": flops / max(n_launch, 1),
                         "mfma_products_per_mac": products,
                         "executed_mfma_frac": achieved * products / peak,
                         "vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS},
            "certification_fallback_rows": int(fallback_rows),
            "escalated_rows": int(sum(st.get("n_escalated_rows", 0) for _, _, st in knn_log)),
            "rounding_bound_self_check": {"max_err_over_eps": max((st.get("max_err_ratio", 0.0) for _, _, st in knn_log), default=0.0),
                                          "note": "max |approximate key - exact key| / eps over all re-ranked candidates; the certification needs < 1"},
            "other_kernels_ms": {"finalize_avg": sum(st["finalize_ms"] for _, _, st in knn_log) / max(n_launch, 1),
                                 "fallback_total": sum(st["fallback_ms"] for _, _, st in knn_log)},
        }
        if check is not None:
            line["check"] = check
        if world == 1:
            # PCIe-inclusive rate of the drop-in API (numpy in -> numpy out); reported beside `value`, never as `value`
            import warnings
            from kiez_amd import Kiez
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kz = Kiez(n_candidates=K, algorithm="SklearnNN", algorithm_kwargs={"metric": metric}, hubness=hub,
                          hubness_kwargs=dict(hub_kw))
                kz.fit(source_h, target_h).kneighbors(k)  # warm-up
                t0 = time.perf_counter()
                kz.fit(source_h, target_h).kneighbors(k)
                t_host = time.perf_counter() - t0
            line["host_api"] = {"value": n_s / t_host, "unit": "queries/s", "ms": t_host * 1e3,
                                "note": "Kiez(...).fit(numpy, numpy).kneighbors(k) -> numpy: includes H2D of both matrices and D2H of the result"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(source_h, target_h, metric, k)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
