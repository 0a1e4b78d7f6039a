#!/usr/bin/env python3
"""Benchmark of the hot path: source queries/s over fit + kneighbors (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--workload ns|c1|c2|c3|c3s|c4s|c4|c1g|hard|cliff|gmm|ea15k] [--no-others]
    python bench.py --openea EMB_DIR KG_DIR [--steps K] [--warmup W]      (real entity-alignment embeddings)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with N > 1 and no torch.distributed environment LAUNCHES ITSELF: the parent (which never touches a
GPU) starts `python -m torch.distributed.run ... bench.py <same arguments>` as a child process, one rank per GPU, forwards
rank 0's JSON line and exits with the child's code.
`--scaling weak` (default): every rank owns a 250k-row source shard of the workload ("ns": 8 GPUs = 2M x 1M).
`--scaling strong`: the TOTAL source is fixed -- default workload BASELINE.json configuration 4 at its stated size (2M x 1M,
d = 300, k = 10, CSLS), rank r owns rows row_slice(2M, r, N); N = 1 is the whole configuration on one GPU.

A "step" is one full `fit(source, target)` + `kneighbors(k)` over synthetic embeddings that are already resident in
HBM (the reference's `rng.rand` data, float32).  Default workload = the shape BASELINE.json quotes its target on, one
GPU's share of it: 250k source rows x 1M targets, d=200, k=10, CSLS ("ns"); with N > 1 every rank owns a 250k-row source
shard (weak scaling, 8 GPUs = 2M x 1M); the target lives on rank 0 and is RCCL-broadcast inside `fit`.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects; at N = 1 the
other BASELINE configurations (C1, C2, C3 full size, C4's per-GPU share) are run briefly too and reported under
`other_workloads` (each with step time, main-kernel time, roofline fraction, fallback counts and a result check).
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
    "c4": (2_000_000, 1_000_000, 300, "euclidean", 10, 10, "CSLS", {},
           "C4 at its stated size: 2M source rows x 1M target, d=300, k=10, CSLS (strong scaling: the 2M rows are split over the ranks)"),
    "ns": (250_000, 1_000_000, 200, "euclidean", 10, 10, "CSLS", {},
           "north-star target shape (BASELINE.json), per-GPU share: 250k source rows x 1M target, d=200, k=10, CSLS"),
    "c1g": (100_000, 100_000, 128, "euclidean", 10, 10, None, {},
            "C1 on gaussian data (rng.randn) for contrast with uniform: 100k x 100k, d=128, euclidean, k=10, hubness=None"),
    "hard": (300_000, 301_000, 64, "cosine", 50, 50, "MutualProximity", {"method": "empiric"},
             "data that is hard for the fp16 first pass (40 tight gaussian clusters far from the centre, rows stored cluster by "
             "cluster; tools/short_route_stress.py): 300k x 301k, d=64, cosine, k=50, MutualProximity empiric"),
    "cliff": (200_000, 201_000, 200, "euclidean", 10, 10, "CSLS", {},
              "clusters of very different density (40 gaussian clusters, spreads 0.05 .. 1.6 at centres ~42 from the origin, rows shuffled; "
              "tools/cliff_probe.py, last kind): a third of the rows is beyond every operand precision and is answered by the range "
              "re-search on the exact float64 kernels (csrc/kz_range.h): 200k x 201k, d=200, euclidean, k=10, CSLS"),
    "ea15k": (15_000, 15_000, 300, "euclidean", 10, 10, "CSLS", {},
              "the size the reference's users actually run (OpenEA 15K entity-alignment sets, kiez/io/data_loading.py:75-99): 15k x 15k, "
              "d=300, euclidean, k=10, CSLS, L2-normalised gaussian mixture -- a launch- and latency-bound step, not a throughput one"),
    "gmm": (200_000, 200_000, 300, "euclidean", 10, 10, "CSLS", {},
            "entity-alignment-like embeddings (the reference's real workload, kiez/io/data_loading.py:75-99, has no synthetic stand-in): "
            "L2-normalised gaussian mixture, 256 clusters shared by both sides, rows in random order: 200k x 200k, d=300, euclidean, k=10, CSLS"),
}
METRIC = "source queries/sec (fit+kneighbors) + recall@k vs reference, 1/2/4/8 GPU"   # BASELINE.json's metric
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, 'Peak FP32 (matrix)'
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, 'Peak BF16/FP16 MFMA ~2.5 PF dense'
TIER_NAME = {0: "f32", 1: "bf16x2", 2: "f16"}
TIER_PRODUCTS = {0: 1, 1: 3, 2: 1}
TIER_KERNEL = {
    2: "kz_knn_cand_h_kernel (fp16 MFMA 32x32x16 on centred operands, 1 product per multiply-add, fused distance+top-k)",
    1: "kz_knn_cand_bf_kernel (split-bf16 MFMA 32x32x16, 3 products per multiply-add, fused distance+top-k)",
    0: "kz_knn_cand_kernel (fp32 MFMA 32x32x2 fused distance+top-k)"}


LINE_LIMIT = 4096     # bytes of the ONE stdout line (the driver keeps a bounded tail of stdout: round 5's 20 KB line did not parse)
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms",
                 "algorithmic_flop_per_launch", "effective_clock_ghz", "mfma_pipe_busy", "shared_sweeps")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "fit_seconds_extrapolated", "kneighbors_seconds_extrapolated")
CHECK_KEYS = ("rows", "ranks", "fit_state_rows", "fit_state_max_rel_err", "fit_state_rows_identical", "index_rows_identical",
              "knife_edge_rows", "knife_edge_rows_identical", "rows_not_knife_edge", "recall_at_k", "max_rel_dist_err",
              "hits_reference_formula")
CONFIG_DROP = ("inputs", "parallelism", "engine")


def _r(x, digits=9):
    """Floats of the compact line at 9 significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_line(line, detail_path):
    """The ONE stdout line: exactly what the contract reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step, scaling,
    dtype, data, config, roofline, cpu_baseline) + check, recall_at_k, the per-workload `summary` rows and the name of the side
    file that holds everything else.  Never longer than LINE_LIMIT bytes: the optional parts are dropped, last first, until it fits."""
    c = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    if line.get("launch_check"):
        c["launch_check"] = True
    cfg = dict(line.get("config") or {})
    par = cfg.get("parallelism")
    for k in CONFIG_DROP:
        cfg.pop(k, None)
    if par:
        cfg["parallelism"] = par.split(" (")[0]
    c["config"] = cfg
    if "roofline" in line:
        roof = {k: line["roofline"].get(k) for k in ROOFLINE_KEYS}
        roof["kernel"] = (roof.get("kernel") or "").split(" (")[0]
        c["roofline"] = roof
    if "cpu_baseline" in line:
        c["cpu_baseline"] = {k: line["cpu_baseline"][k] for k in CPU_KEYS if k in line["cpu_baseline"]}
    if "check" in line:
        c["check"] = {k: line["check"][k] for k in CHECK_KEYS if k in line["check"]}
    c["recall_at_k"] = line.get("recall_at_k")
    optional = []          # dropped from the END of this list first when the line is too long
    for k in ("summary", "host_api", "collective_ms_per_step", "hits", "certification_fallback_rows", "escalated_rows"):
        if line.get(k) is not None:
            v = line[k]
            if k == "host_api":
                v = {"value": v["value"], "unit": v["unit"], "ms": v["ms"]}
            c[k] = v
            optional.append(k)
    c["detail"] = detail_path
    c = _r(c)
    if c.get("roofline") and c["roofline"].get("achieved") is not None and c["roofline"].get("peak"):
        c["roofline"]["frac"] = c["roofline"]["achieved"] / c["roofline"]["peak"]     # (exactly the quotient of the two numbers printed)
    text = json.dumps(c, separators=(",", ":"))
    while len(text) > LINE_LIMIT and optional:
        c.pop(optional.pop())
        text = json.dumps(c, separators=(",", ":"))
    if len(text) > LINE_LIMIT and "sample" in c.get("cpu_baseline", {}):
        c["cpu_baseline"]["sample"] = c["cpu_baseline"]["sample"][:160]
        text = json.dumps(c, separators=(",", ":"))
    if len(text) > LINE_LIMIT:      # (cannot happen with the fields above; the contract fields alone are < 2 KB)
        raise RuntimeError(f"bench.py: the result line is {len(text)} bytes (> {LINE_LIMIT})")
    return text


def emit(line, json_fd, args):
    """Full record -> bench_detail.json next to this script (and gpurun_out/ when that exists) and stderr; compact record ->
    the ONE stdout line."""
    full = json.dumps(line)
    name = getattr(args, "detail", None) or "bench_detail.json"
    written = None
    targets = [Path(name)] if Path(name).is_absolute() else [ROOT / name, ROOT / "gpurun_out" / name]
    for t in targets:
        try:
            if t.parent.is_dir():
                t.write_text(full + "\n")
                written = written or name
        except OSError:
            pass
    sys.stderr.write("bench.py full record: " + full + "\n")
    sys.stderr.flush()
    os.write(json_fd, (compact_line(line, written) + "\n").encode())


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(source, target, metric, K, k, hub, hub_kw, budget_flop=6e11, n_s_total=None):
    """CPU leg on this host's cores: the reference's own CPU path = scikit-learn's brute-force kNN
    (what SklearnNN._kneighbors delegates to, kiez/neighbors/exact/sklearn_nearest_neighbors.py:98-101) for both distance
    passes + the oracle's restatement of the rescale / final sort (oracle/kiez_oracle.py), timed on a bounded ROW SAMPLE
    against the full index, one discarded warm-up, median of 3, and extrapolated linearly in rows (every stage is
    row-independent given the fit state).  value = n_source / (t_fit + t_kneighbors).
    N > 1 (rank 0, after the timed region): `source` = the source rows rank 0 holds on the host -- all shards gathered when the
    workload has a reverse pass (its index is the WHOLE source), rank 0's own shard otherwise -- and `n_s_total` the job's rows:
    the baseline is that of the whole job, as `value` is."""
    from oracle import kiez_oracle as O
    n_held, d = source.shape
    n_s = int(n_s_total) if n_s_total else n_held
    n_t = target.shape[0]
    rows_f = int(max(min(256, n_held), min(n_held, budget_flop / (2.0 * n_t * d))))   # forward sample: source rows vs ALL targets
    rows_r = int(max(min(256, n_t), min(n_t, budget_flop / (2.0 * n_s * d))))          # reverse sample: target rows vs ALL source rows
    if hub is not None and n_held != n_s:
        raise ValueError("cpu_baseline: a reverse pass is timed against the whole source")
    out = {"unit": "queries/s", "cores": os.cpu_count(), "kind": "port"}
    metric_c = O.canonical_metric(metric)
    try:
        from sklearn.neighbors import NearestNeighbors

        def knn(q, index, kk):
            nn = NearestNeighbors(n_neighbors=kk, algorithm="brute", metric=metric_c).fit(index)
            nn.kneighbors(q[:128])  # thread-pool warm-up, discarded
            ts, res = [], None
            for _ in range(3):
                t0 = time.perf_counter()
                res = nn.kneighbors(q)
                ts.append(time.perf_counter() - t0)
            return _median(ts), res
        out["kind"] = "reference"
    except Exception as e:  # pragma: no cover
        out["note_sklearn"] = f"sklearn unavailable ({e}); distance passes timed with the oracle's numpy search"

        def knn(q, index, kk):
            O.knn_exact(q[:64], index, kk, metric_c)
            ts, res = [], None
            for _ in range(3):
                t0 = time.perf_counter()
                res = O.knn_exact(q, index, kk, metric_c)
                ts.append(time.perf_counter() - t0)
            return _median(ts), res
    t_rev = t_resc = 0.0
    if hub is None:
        t_fwd, _ = knn(source[:rows_f], target, k)
        sample = f"forward pass: first {rows_f} source rows x all {n_t} target rows"
    else:
        t_rev_s, (rd, ri) = knn(target[:rows_r], source, K)
        t_fwd, (fd, fi) = knn(source[:rows_f], target, K)
        t_rev = t_rev_s * n_t / rows_r
        # rescale + final sort (oracle) on the forward sample; the fit state of the sampled candidates comes from a reverse
        # pass of exactly those target rows (timed above per row), here replaced by the sampled reverse rows tiled
        reps = -(-n_t // rows_r)
        rd_full = np.tile(rd, (reps, 1))[:n_t]
        ri_full = np.tile(ri, (reps, 1))[:n_t]
        rows_x = min(rows_f, 2000 if hub.lower().startswith("mutual") and hub_kw.get("method") == "empiric" else rows_f)
        h = hub.lower()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            if h == "csls":
                hr = O.csls_transform(fd[:rows_x], fi[:rows_x], rd_full)
            elif h in ("localscaling", "ls"):
                hr = O.ls_transform(fd[:rows_x], fi[:rows_x], rd_full, hub_kw.get("method", "standard"))
            elif h in ("mutualproximity", "mp") and hub_kw.get("method", "normal") in ("exact", "empiric"):
                hr = O.mp_empiric_transform(fd[:rows_x], fi[:rows_x], rd_full, ri_full)
            elif h in ("mutualproximity", "mp"):
                hr = O.mp_normal_transform(fd[:rows_x], fi[:rows_x], rd_full)
            else:
                hr = fd[:rows_x]
            O.sort_topk(hr, fi[:rows_x], k)
            ts.append(time.perf_counter() - t0)
        t_resc = _median(ts) * n_s / rows_x
        sample = (f"reverse pass: first {rows_r} target rows x all {n_s} source rows; forward pass: first {rows_f} source rows x all "
                  f"{n_t} target rows; rescale + sort (oracle): {rows_x} rows; each extrapolated linearly in rows")
    t_fwd_full = t_fwd * n_s / rows_f
    t_fit, t_kn = t_rev, t_fwd_full + t_resc
    out.update(value=n_s / (t_fit + t_kn), sample=sample + f", d={d}, {metric}, K={K}, k={k}",
               fit_seconds_extrapolated=t_fit, kneighbors_seconds_extrapolated=t_kn,
               measured_seconds={"reverse_sample": t_rev * rows_r / n_t if hub else 0.0, "forward_sample": t_fwd},
               protocol="one discarded warm-up, median of 3 per stage",
               note="distance passes = sklearn.neighbors.NearestNeighbors(algorithm='brute') (the reference's CPU path); "
                    "rescale/sort = oracle/kiez_oracle.py")
    return out


def strided_rows(n, rows):
    """The row sample of every check: `rows` rows of n at a fixed stride (deterministic, spans the whole range)."""
    return np.arange(0, n, max(1, n // max(rows, 1)))[:rows]


def sample_check(res, sk, eng, comm, source_dev, target_h, K, k, metric, hub, hub_kw, rows=1024, rows_per_rank=128):
    """Result check that also works where the oracle cannot run the whole fit (1M-row reverse pass) and on ANY number of ranks:
    a row sample of BOTH kNN passes against the oracle's exact float64 search, then the oracle's rescale + final sort on the
    sampled source rows fed with the (sample-verified) fit state of this run.

    Collective (every rank calls it, OUTSIDE the timed region): each rank contributes `rows` sampled rows of its shard's result
    (world 1) or `rows_per_rank` of them (world > 1), with the source rows they belong to and their global ids; where the fit has
    a reverse pass the source shards are gathered so that rank 0 can check `rows` sampled TARGET rows of the fit state against
    the whole source.  Rank 0 (the holder of `target_h`) runs the oracle and returns the record; the other ranks return None."""
    from oracle import kiez_oracle as O
    torch_ = torch
    metric_c = O.canonical_metric(metric)
    world, rank = comm.world, comm.rank
    counts = list(sk.counts)
    per_rank = rows if world == 1 else rows_per_rank
    sel_counts = [len(strided_rows(c, min(per_rank, c))) for c in counts]
    sel_local = strided_rows(counts[rank], min(per_rank, counts[rank]))
    sel_t = torch_.from_numpy(sel_local).to(res[0].device)
    g_d = comm.all_gather_rows(res[0][sel_t].contiguous(), sel_counts)
    g_i = comm.all_gather_rows(res[1][sel_t].contiguous(), sel_counts)
    g_src = comm.all_gather_rows(source_dev[sel_t].contiguous(), sel_counts)
    src_full = comm.gather_rows_to0(source_dev, counts) if hub is not None else None     # (the reverse pass' index: every shard, on rank 0 only)
    if rank != 0:
        return None
    begins = np.concatenate([[0], np.cumsum(counts)])[:-1]
    gid = np.concatenate([begins[r] + strided_rows(counts[r], min(per_rank, counts[r])) for r in range(world)])
    owner = np.repeat(np.arange(world), sel_counts)
    got_d, got_i, s_rows = eng.to_numpy(g_d), eng.to_numpy(g_i), eng.to_numpy(g_src)
    n_s, n_t = int(sum(counts)), len(target_h)
    s64 = s_rows.astype(np.float64) if metric_c == "cosine" else s_rows
    t_all = target_h.astype(np.float64) if metric_c == "cosine" else target_h
    out = {"rows": int(len(gid)), "ranks": world, "rows_per_rank": [int(c) for c in sel_counts]}
    h = hub.lower() if hub else None
    empiric = h in ("mutualproximity", "mp") and hub_kw.get("method") in ("exact", "empiric")
    if hub is None:
        od, oi = O.knn_exact(s64, t_all, k, metric_c)
        fi = oi
    else:
        fd, fi = O.knn_exact(s64, t_all, K, metric_c)
        st = sk.state
        # verify the fit state on a sample of target rows against ALL source rows (every shard)
        sel_tg = strided_rows(n_t, min(rows, n_t))
        s_all = eng.to_numpy(src_full)
        if metric_c == "cosine":
            s_all = s_all.astype(np.float64)
        rd, ri = O.knn_exact(t_all[sel_tg], s_all, min(K, n_s), metric_c)
        out["fit_state_rows"] = int(len(sel_tg))

        def rel(got, want):
            return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
        if h == "csls":
            r_t = eng.to_numpy(st["r_t"])
            out["fit_state_max_rel_err"] = rel(r_t[sel_tg], rd.mean(axis=1))
            hr = 2.0 * fd - fd.mean(axis=1)[:, None] - r_t[fi]
        elif h in ("localscaling", "ls"):
            r_t = eng.to_numpy(st["r_t"])
            nicdm = str(hub_kw.get("method", "standard")).lower() == "nicdm"
            out["fit_state_max_rel_err"] = rel(r_t[sel_tg], rd.mean(axis=1) if nicdm else rd[:, -1])
            hr = fd / np.sqrt(fd.mean(axis=1)[:, None] * r_t[fi]) if nicdm else 1.0 - np.exp(-1 * fd**2 / (fd[:, -1][:, None] * r_t[fi]))
        elif empiric:
            d_t2s, i_t2s = eng.to_numpy(st["dist_t2s"]), eng.to_numpy(st["ind_t2s"])
            out["fit_state_rows_identical"] = int((i_t2s[sel_tg] == ri).all(axis=1).sum())
            if world == 1:   # (the transform is the oracle's slow part: 256 rows of one shard; on > 1 rank every rank's 128)
                sub = min(len(gid), 256)
                fd, fi, got_d, got_i, gid, owner = fd[:sub], fi[:sub], got_d[:sub], got_i[:sub], gid[:sub], owner[:sub]
                out["rows"] = int(sub)
            hr = O.mp_empiric_transform(fd, fi, d_t2s, i_t2s)
        elif h in ("mutualproximity", "mp"):
            mu_t, sd_t = eng.to_numpy(st["mu_t"]), eng.to_numpy(st["sd_t"])
            out["fit_state_max_rel_err"] = max(rel(mu_t[sel_tg], np.nanmean(rd, axis=1)), rel(sd_t[sel_tg], np.nanstd(rd, axis=1)))
            mu, sd = np.nanmean(fd, axis=1)[:, None], np.nanstd(fd, axis=1)[:, None]
            hr = 1 - O._norm_sf(fd, mu, sd) * O._norm_sf(fd, mu_t[fi], sd_t[fi])
        else:
            return None     # DisSimLocal shifts by a minimum over the WHOLE result: no row sample decides it
        od, oi = O.sort_topk(hr, fi, k)
    same = (got_i == oi).all(axis=1)
    if empiric:
        # MP-empiric values are multiples of 1/K; a row whose candidate list contains the query's own id is a knife edge
        # (the reference compares a pair's forward and reverse BLAS value with a strict '>').  Those rows are
        # counted apart, never ORed into "identical": index_rows_identical counts the other rows only.
        knife = (fi == gid[:, None]).any(axis=1)
        out.update(knife_edge_rows=int(knife.sum()), knife_edge_rows_identical=int((same & knife).sum()),
                   rows_not_knife_edge=int((~knife).sum()))
        same &= ~knife
    plain = np.ones(len(got_d), dtype=bool) if "knife_edge_rows" not in out else ~knife     # (a knife-edge row may differ by one count, 1 / K)
    out.update(index_rows_identical=int(same.sum()),
               index_rows_identical_per_rank=[int(same[owner == r].sum()) for r in range(world)],
               recall_at_k=float(np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(got_i, oi)])),
               max_rel_dist_err=float(np.max(np.abs(got_d[plain] - od[plain]) / np.maximum(np.abs(od[plain]), 1e-12))) if plain.any() else 0.0)
    return out


def synth_rows(name, seed, rows, d):
    """Synthetic float32 embeddings of one workload, generated in blocks (no [rows, d] float64 transient).  Default: the
    reference's `rng.rand` (kiez/kiez.py:50-52); names ending in "g": `rng.randn`; "hard": 40 tight gaussian clusters far from
    the centre, stored cluster by cluster (tools/short_route_stress.py)."""
    rng = np.random.RandomState(seed)
    out = np.empty((rows, d), dtype=np.float32)
    if name in ("gmm", "ea15k"):
        # 256 cluster centres common to both embedding spaces (aligned KGs), within-cluster spread a third of the centres' own,
        # every row L2-normalised, rows in random order
        centres = np.random.RandomState(6).standard_normal((256, d)).astype(np.float32)
        for b in range(0, rows, 100_000):
            m = min(100_000, rows - b)
            x = centres[rng.randint(0, 256, m)] + np.float32(0.35) * rng.standard_normal((m, d)).astype(np.float32)
            out[b:b + m] = x / np.sqrt((x * x).sum(axis=1, keepdims=True))
        return out
    if name == "cliff":
        centres = (np.random.RandomState(5).standard_normal((40, d)) * 3).astype(np.float32)     # the same clusters on both sides
        spread = (0.05 * 2.0 ** np.random.RandomState(6).randint(0, 6, 40)).astype(np.float32)
        for b in range(0, rows, 100_000):
            m = min(100_000, rows - b)
            c = rng.randint(0, 40, m)
            out[b:b + m] = centres[c] + spread[c, None] * rng.standard_normal((m, d)).astype(np.float32)
        return out
    if name == "hard":
        centres = np.random.RandomState(5).standard_normal((40, d)) * 3     # the same centres on both sides
        sizes = rng.multinomial(rows, np.ones(40) / 40)
        b = 0
        for c in range(40):
            out[b:b + sizes[c]] = centres[c] + 0.4 * rng.standard_normal((sizes[c], d))
            b += sizes[c]
        return out
    gen = rng.randn if name.endswith("g") else rng.rand
    for b in range(0, rows, 250_000):
        out[b:b + 250_000] = gen(min(250_000, rows - b), d)
    return out


def run_workload(name, eng, comm, dist, rank, world, steps, warmup, check=True, scaling="weak", check_rows=1024,
                 target_upload="broadcast"):
    """Generate the data, run warm-up + timed steps, return (summary dict, host arrays, result, source tensor) on every rank.
    scaling = "weak": every rank owns WORKLOADS[name][0] source rows; "strong": that many rows in total, split over the ranks.
    target_upload = "broadcast": the target lives on rank 0 and is RCCL-broadcast inside every fit (north_star's partitioning);
    "local": every rank holds the target itself (generated from the same seed, uploaded over its own PCIe link before the timed
    region; SURVEY 8e's alternative) -- `fit` then runs no broadcast."""
    from kiez_amd.distributed import ShardedKiez, row_slice
    n_s, n_t, d, metric, K, k, hub, hub_kw, desc = WORKLOADS[name]
    n_s_total = n_s * world if scaling == "weak" else n_s
    if scaling == "strong":
        n_s = row_slice(n_s_total, rank, world)[1]
    source_h = synth_rows(name, 0 if rank == 0 else 1000 + rank, n_s, d)
    local_target = target_upload == "local"
    target_h = synth_rows(name, 77, n_t, d) if (rank == 0 or local_target) else None
    source = eng.to_engine(source_h)
    target = eng.to_engine(target_h) if target_h is not None else None
    eng.sync()
    sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=hub_kw,
                     engine=eng, comm=comm, cache_target=target_upload == "cached")
    knn_log = []
    orig_knn = eng.knn

    def logged_knn(qm, q_begin, q_count, im, kk, exclude_self):
        out = orig_knn(qm, q_begin, q_count, im, kk, exclude_self)
        knn_log.append((q_count, im.shape[0], dict(eng.last_stats)))
        return out

    eng.knn = logged_knn
    rev_log = []
    orig_dual = getattr(eng, "knn_dual", None)

    def logged_dual(am, bm, kk):
        # one sweep of the distance matrix serves both directions (kz_knn_dual): ONE launch of the dominant kernel, whose
        # algorithmic work is the a x b distance matrix once; what the reverse direction costs besides is kept apart
        out = orig_dual(am, bm, kk)
        knn_log.append((am.shape[0], bm.shape[0], dict(eng.last_stats)))
        rev = dict(eng.last_stats_reverse)
        if rev.get("dual"):
            rev_log.append(rev)
        else:   # the library fell back to two ordinary searches: the reverse one is a launch of its own
            knn_log.append((bm.shape[0], am.shape[0], rev))
        return out

    if orig_dual is not None:
        eng.knn_dual = logged_dual

    def step():
        sk.fit(source, target, target_from_rank0=not local_target)
        return sk.kneighbors(k)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        eng.sync()

    try:
        res = None
        for _ in range(warmup):
            res = step()
        knn_log.clear()
        rev_log.clear()
        comm.reset_timers()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        fence()
        elapsed = time.perf_counter() - t0
    finally:
        eng.knn = orig_knn
        if orig_dual is not None:
            eng.knn_dual = orig_dual
    if dist.is_initialized():
        tt = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.cpu()[0])
    # roofline of the dominant kernel (fused distance + top-k): algorithmic flops 2*n_q*n_i*d per launch
    flops = sum(2.0 * q * n * d for q, n, _ in knn_log)
    kernel_s = sum(st["main_kernel_ms"] for _, _, st in knn_log) * 1e-3
    n_launch = len(knn_log)
    achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
    tier_ms = {t: sum(st["main_kernel_ms"] for _, _, st in knn_log if st.get("first_pass") == t) for t in (0, 1, 2)}
    tier = max(tier_ms, key=tier_ms.get)       # 0 float32 operands, 1 split-bf16 (3 products), 2 fp16 (1 product)
    peak = PEAK_F32_MFMA_TFLOPS if tier == 0 else PEAK_BF16_MFMA_TFLOPS
    summary = {
        "name": name, "desc": desc, "n_s": n_s, "n_t": n_t, "d": d, "metric": metric, "K": K, "k": k, "hub": hub, "hub_kw": hub_kw,
        "n_s_total": n_s_total, "scaling": scaling, "shard_rows": [int(c) for c in sk.counts],
        "elapsed": elapsed, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "value": n_s_total * steps / elapsed,
        "tier": tier, "peak": peak, "achieved": achieved, "n_launch": n_launch, "kernel_s": kernel_s, "flops": flops,
        "fallback_rows": int(sum(st["n_fallback_rows"] for _, _, st in knn_log)),
        "escalated_rows": int(sum(st.get("n_escalated_rows", 0) for _, _, st in knn_log)),
        # of fallback_rows: the handful of rows per search answered by the exact kernels launched speculatively behind the finalize
        "spec_rows": int(sum(st.get("n_spec_rows", 0) for _, _, st in knn_log)),
        "range_rows": int(sum(st.get("n_range_rows", 0) for _, _, st in knn_log)),
        "range_group_rows": int(sum(st.get("n_range_group_rows", 0) for _, _, st in knn_log)),
        "max_err_ratio": max((st.get("max_err_ratio", 0.0) for _, _, st in knn_log), default=0.0),
        "finalize_avg_ms": sum(st["finalize_ms"] for _, _, st in knn_log) / max(n_launch, 1),
        "fallback_total_ms": sum(st["fallback_ms"] for _, _, st in knn_log),
        "probe_ms_per_step": sum(st.get("probe_ms", 0.0) for _, _, st in knn_log) / max(steps, 1),
        # > 0: launches that ran the fp16 tier's WIDE route (that many lists of 16 per query: dense keys around the k-th neighbour)
        "wide_lists": max((st.get("wide_lists", 0) for _, _, st in knn_log), default=0),
        "first_pass_fail_rows": int(sum(st.get("n_first_pass_fail", 0) for _, _, st in knn_log)),
        "collective_ms_per_step": comm.timers_ms(steps),
        "collective_traffic_per_step": comm.traffic(steps),   # per kind: calls and payload bytes this rank handed over
        # shared sweeps (kz_knn_dual): how many of the launches served both directions, and what the reverse direction cost
        # on top of the sweep (sample sweep + scatter + select, its finalize, rows searched again)
        "shared_sweeps": len(rev_log),
        "reverse_extra_ms_per_step": sum(r["main_kernel_ms"] + r["finalize_ms"] + r["fallback_ms"] for r in rev_log) / max(steps, 1),
        "reverse_events_per_row": (sum(r["n_events"] for r in rev_log) / max(len(rev_log), 1) / max(min(n_s, n_t), 1)) if rev_log else 0.0,
        # rows of the reverse direction searched again (an overflowing event buffer, fewer than k events, an uncertified list)
        "reverse_escalated_rows": int(sum(r.get("n_escalated_rows", 0) for r in rev_log)),
    }
    summary["target_upload"] = target_upload
    # the oracle check is a collective of its own, outside the timed region: at any world size rank 0 verifies a sample of the
    # fit state and sampled rows of EVERY rank's shard (None on the other ranks)
    summary["check"] = sample_check(res, sk, eng, comm, source, target_h, K, k, metric, hub, hub_kw, rows=check_rows) if check else None
    return summary, (source_h, target_h), res, source


def pmc_traffic(workload, s):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json,
    tools/pmc_derive.py).  The counters are per KERNEL DISPATCH; a launch here (one kz_knn / kz_knn_dual call) sweeps its query
    rows in chunks of 524288, one dispatch each -- scaled by the ratio of the two durations so that `traffic` refers to the
    same launch as `achieved`.  (None, None) where no PMC pass was taken."""
    rec = pmc_record(workload, s)
    if not rec or not s["n_launch"]:
        return None, None
    try:
        per_dispatch = rec.get("hbm_bytes_per_launch")
        if per_dispatch is None:
            return None, None
        avg_launch_ms = s["kernel_s"] / s["n_launch"] * 1e3
        dispatches = max(1, round(avg_launch_ms / rec["avg_ms_under_pmc"]))
        return per_dispatch * dispatches, dispatches
    except Exception:
        return None, None


def pmc_record(workload, s):
    """This workload's record of the committed PMC passes (profiles/pmc_traffic.json), {} where none was taken."""
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    try:
        return json.loads(pmc.read_text()).get(workload + "_" + TIER_NAME[s["tier"]], {}) if pmc.exists() else {}
    except Exception:
        return {}


def roofline_of(workload, s):
    """The `roofline` object of one workload's dominant kernel (contract in the task statement)."""
    tier = s["tier"]
    traffic, dispatches = pmc_traffic(workload, s)
    rec = pmc_record(workload, s)
    clock, busy = rec.get("clock_ghz"), rec.get("mfma_pipe_busy")
    return {"bound": "mfma", "kernel": TIER_KERNEL[tier], "achieved": s["achieved"], "peak": s["peak"], "unit": "TFLOP/s",
            "frac": s["achieved"] / s["peak"], "traffic": traffic, "traffic_kernel_dispatches_per_launch": dispatches,
            "launches": s["n_launch"], "avg_launch_ms": s["kernel_s"] / max(s["n_launch"], 1) * 1e3,
            "algorithmic_flop_per_launch": s["flops"] / max(s["n_launch"], 1), "mfma_products_per_mac": TIER_PRODUCTS[tier],
            "shared_sweeps": s["shared_sweeps"],
            # from the committed PMC pass of this workload's dominant kernel (profiles/pmc_traffic.json): the clock the chip held
            # (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the share of those cycles the matrix pipe was busy; their product over the
            # nominal 2.4 GHz is the executed-MFMA fraction of peak -- frac = busy x clock / 2.4 x (d / d_pad)
            "effective_clock_ghz": clock, "mfma_pipe_busy": busy,
            "busy_x_clock_over_nominal": (busy * clock / 2.4) if (clock and busy) else None, "pmc_source": rec.get("source")}


def short(summary):
    """Compact per-workload record for `other_workloads`."""
    s = summary
    return {"workload": s["desc"], "roofline": roofline_of(s["name"], s), "warmup": s["warmup"], "ms_per_step": s["ms_per_step"], "value": s["value"], "unit": "queries/s",
            "main_kernel_avg_ms": s["kernel_s"] / max(s["n_launch"], 1) * 1e3, "dtype": TIER_NAME[s["tier"]],
            "roofline_frac": s["achieved"] / s["peak"], "achieved_tflops": s["achieved"],
            "finalize_avg_ms": s["finalize_avg_ms"], "certification_fallback_rows": s["fallback_rows"],
            "escalated_rows": s["escalated_rows"], "speculative_rescue_rows": s["spec_rows"], "range_research_rows": s["range_rows"], "range_group_rows": s["range_group_rows"], "fallback_total_ms": s["fallback_total_ms"], "probe_ms_per_step": s["probe_ms_per_step"],
            "wide_lists": s["wide_lists"], "first_pass_fail_rows": s["first_pass_fail_rows"],
            "max_err_over_eps": s["max_err_ratio"], "steps": s["steps"], "shared_sweeps": s["shared_sweeps"],
            "reverse_extra_ms_per_step": s["reverse_extra_ms_per_step"], "reverse_escalated_rows": s["reverse_escalated_rows"], "check": s["check"]}


def run_openea(args):
    """Real entity-alignment embeddings: OpenEA directory -> kiez_amd.io.from_openea -> Kiez.fit/kneighbors
    on the GPU -> kiez_amd.evaluate.hits on the device.  A step = fit + kneighbors with both matrices resident in HBM (torch
    tensors, zero-copy) and the result left in HBM; hits@k is evaluated once, outside the timed region."""
    import warnings
    from kiez_amd import Kiez
    from kiez_amd.evaluate import hits
    from kiez_amd.io import from_openea
    emb_dir, kg_dir = args.openea
    t0 = time.perf_counter()
    emb1, emb2, _, _, links = from_openea(emb_dir, kg_dir)
    t_load = time.perf_counter() - t0
    K = k = args.openea_k
    hub = None if args.openea_hubness.lower() in ("none", "no") else args.openea_hubness
    dev = torch.device("cuda", 0)
    s_dev, t_dev = torch.from_numpy(np.ascontiguousarray(emb1)).to(dev), torch.from_numpy(np.ascontiguousarray(emb2)).to(dev)
    torch.cuda.synchronize()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=K, algorithm="SklearnNN", algorithm_kwargs={"metric": args.openea_metric}, hubness=hub)

        def step():
            kz.fit(s_dev, t_dev)
            return kz.kneighbors_device(k)
        res = None
        for _ in range(args.warmup):
            res = step()
        kz.algorithm.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step()
        kz.algorithm.ctx.sync()
        elapsed = time.perf_counter() - t0
        dist_dev, ind_dev = res
        t0 = time.perf_counter()
        h = hits(ind_dev, links, k=[1, 5, 10])
        t_hits = time.perf_counter() - t0
    n_s, d = emb1.shape
    line = {"metric": "source queries/sec (fit+kneighbors) + recall@k vs reference, 1/2/4/8 GPU", "value": n_s * args.steps / elapsed,
            "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": TIER_NAME.get((kz.algorithm.last_stats or {}).get("first_pass", 2), "f16"),
            "data": f"OpenEA layout: {emb_dir} + {kg_dir}",
            "config": {"workload": "OpenEA dataset", "n_source": int(n_s), "n_target": int(emb2.shape[0]), "d": int(d), "metric": args.openea_metric,
                       "n_candidates": K, "k": k, "hubness": hub, "gold_links": len(links), "inputs": f"{emb1.dtype}, resident in HBM"},
            "hits": {str(kk): v for kk, v in h.items()}, "load_seconds": t_load, "hits_ms": t_hits * 1e3}
    if not args.no_check:
        # the neighbour matrix against the oracle pipeline on a row sample + hits@k by the reference's formula on the full result
        from oracle import kiez_oracle as O
        ind = ind_dev.numpy()
        rows = np.arange(0, n_s, max(1, n_s // 256))[:256]
        chk = {"rows": int(len(rows)), "hits_reference_formula":
               {kk: sum(1 for i in range(n_s) if i in links and links[i] in ind[i][:kk]) / len(links) for kk in (1, 5, 10)}}
        if n_s * emb2.shape[0] <= 4e8:   # the oracle pipeline needs the full reverse pass: small datasets only
            metric_c = O.canonical_metric(args.openea_metric)
            s64 = emb1.astype(np.float64) if metric_c == "cosine" else emb1
            t64 = emb2.astype(np.float64) if metric_c == "cosine" else emb2
            od, oi = O.kiez_pipeline(s64, t64, K, k, metric_c, 2, hub, {})
            chk["index_rows_identical"] = int((ind[rows] == oi[rows]).all(axis=1).sum())
        line["check"] = chk
    return line


def job_cpu_baseline(args, comm, eng, s, source_dev, source_h, target_h):
    """`cpu_baseline` of the line at ANY world size: rank 0 times the reference's CPU path on a bounded sample of the WHOLE
    job's workload after the timed region.  A collective when the workload has a reverse pass on more than one rank (its index is the
    whole source: the shards are gathered to rank 0's host); returns the record on rank 0, None elsewhere."""
    if args.no_cpu_baseline:
        return None
    n_s, n_t, d, metric, K, k, hub, hub_kw, _ = WORKLOADS[s["name"]]
    held = source_h
    if comm.world > 1 and hub is not None:
        full = comm.gather_rows_to0(source_dev, [row for row in s["shard_rows"]])      # (rank 0 only holds the gathered shards)
        held = eng.to_numpy(full) if comm.rank == 0 else None
        del full
    if comm.rank != 0:
        return None
    out = cpu_baseline(held, target_h, metric, K, k, hub, hub_kw, n_s_total=s["n_s_total"])
    out["workload_rows"] = {"n_source_total": s["n_s_total"], "n_target": n_t}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: ns (weak scaling), c4 (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank owns the workload's source rows (default); strong: the workload's source rows are split "
                         "over the ranks (default workload: BASELINE.json configuration 4 at its stated size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", default="bench_detail.json", help="file name of the full record (written next to bench.py and into gpurun_out/; an absolute path: only there)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle sample check")
    ap.add_argument("--no-others", action="store_true", help="do not run the other BASELINE configurations after the main workload")
    ap.add_argument("--opt", action="append", default=[], help="context option name=value (tuning experiments)")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each secondary workload (other_workloads)")
    ap.add_argument("--other-warmup", type=int, default=2)
    ap.add_argument("--openea", nargs=2, metavar=("EMB_DIR", "KG_DIR"), default=None,
                    help="benchmark an OpenEA-layout dataset (kiez/io/data_loading.py:75-99) instead of a synthetic workload")
    ap.add_argument("--openea-hubness", default="CSLS")
    ap.add_argument("--openea-metric", default="euclidean")
    ap.add_argument("--openea-k", type=int, default=10)
    ap.add_argument("--target-upload", default="broadcast", choices=("broadcast", "local", "cached"),
                    help="N > 1: broadcast = the target lives on rank 0 and is RCCL-broadcast over xGMI inside every fit (default, "
                         "north_star's partitioning); local = every rank uploads the target itself before the timed region and fit "
                         "runs no broadcast (ShardedKiez.fit(target_from_rank0=False)) -- the A/B of the 0.8-1.2 GB transfer per step; "
                         "cached = broadcast by the first fit only (ShardedKiez(cache_target=True): the same target tensor in later fits "
                         "reuses every rank's replica -- the serving pattern: one index, many query batches)")
    ap.add_argument("--launch-check", action="store_true",
                    help="NOT a measurement: run the launch / rendezvous / sharding / collective / reporting path of this script on "
                         "the CPU test engine (tests/cpu_engine.py) over gloo with a tiny shape; `value` is null.  For machines "
                         "without a GPU (tests/test_bench_launch.py)")
    args = ap.parse_args(argv)
    if args.workload is None:
        args.workload = "c4" if args.scaling == "strong" else "ns"
    return args


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process tree (one process per GPU,
    `python -m torch.distributed.run`), forward rank 0's JSON line (the child inherits stdout) and return the child's exit code.
    Called before anything in this process has touched a GPU; this process never does."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # started through torch.distributed.run
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    # Keep stdout clean for the ONE JSON line: librccl prints a start-up banner to fd 1 when the communicator is created.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch.distributed as dist
    if args.launch_check:
        # launch-path check on the CPU test engine (no GPU in this process): gloo, tiny shape, no measurement
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1 or launched:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from kiez_amd.distributed import Comm
        from tests.cpu_engine import OracleEngine
        eng = OracleEngine()
        eng.last_stats = {"main_kernel_ms": 0.0, "n_fallback_rows": 0, "finalize_ms": 0.0, "fallback_ms": 0.0}
        eng.last_stats_reverse = dict(eng.last_stats)
        comm = Comm(time_collectives=True)
        WORKLOADS["launch-check"] = (2000 if args.scaling == "weak" else 4001, 1500, 16, "euclidean", 5, 5, "CSLS", {},
                                     "launch check (CPU test engine over gloo; NOT a measurement)")
        s, (src_h, tgt_h), _, src_dev = run_workload("launch-check", eng, comm, dist, rank, world, args.steps, args.warmup,
                                                     check=not args.no_check, scaling=args.scaling, target_upload=args.target_upload)
        cpu = job_cpu_baseline(args, comm, eng, s, src_dev, src_h, tgt_h)
        if rank == 0:
            line = {"metric": METRIC, "recall_at_k": (s["check"] or {}).get("recall_at_k"),
                    "value": None, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": s["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
                    "dtype": None, "data": "synthetic", "launch_check": True,
                    "config": {"workload": s["desc"], "n_source_total": s["n_s_total"], "n_source_this_rank": s["n_s"], "n_target": s["n_t"],
                               "target_upload": s["target_upload"],
                               "engine": "tests/cpu_engine.py over gloo -- launch path only, value deliberately null"},
                    "collective_ms_per_step": s["collective_ms_per_step"], "collective_traffic_per_step": s["collective_traffic_per_step"]}
            if s["check"] is not None:
                line["check"] = s["check"]
            if cpu is not None:
                line["cpu_baseline"] = cpu
            emit(line, json_fd, args)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # (rank 0 runs the oracle check and the CPU baseline between collectives, tens of seconds during which the other ranks
        #  wait inside the next one: a generous watchdog timeout)
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(minutes=30))

    if args.openea is not None:
        if world != 1:
            raise SystemExit("--openea runs on one GPU")
        line = run_openea(args)
        sys.stdout.flush()
        emit(line, json_fd, args)
        return

    from kiez_amd.distributed import Comm, HipEngine
    eng = HipEngine(local_rank)
    comm = Comm(time_collectives=True)
    for o in args.opt:
        name, val = o.split("=")
        eng.ctx.set_option(name, float(val))

    main_s, (source_h, target_h), _, source_dev = run_workload(args.workload, eng, comm, dist, rank, world, args.steps, args.warmup,
                                                               check=not args.no_check, scaling=args.scaling,
                                                               target_upload=args.target_upload)
    cpu = job_cpu_baseline(args, comm, eng, main_s, source_dev, source_h, target_h)      # (collective: every rank calls it)
    del source_dev
    n_s, n_t, d, metric, K, k, hub, hub_kw, desc = WORKLOADS[args.workload]

    line = None
    if rank == 0:
        s = main_s
        tier = s["tier"]
        line = {
            "metric": METRIC,
            "recall_at_k": (s["check"] or {}).get("recall_at_k"),   # against the oracle on a row sample (None if not checked)
            "value": s["value"],
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": s["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": TIER_NAME[tier],
            "data": "synthetic",
            "config": {"workload": desc, "n_source_total": s["n_s_total"], "n_source_per_gpu": s["n_s_total"] // world, "n_target": n_t, "d": d, "metric": metric,
                       "n_candidates": K, "k": k, "hubness": hub, "hubness_kwargs": hub_kw,
                       "inputs": "float32 rng.rand, resident in HBM; results left in HBM",
                       "target_upload": s["target_upload"],
                       "parallelism": f"source row-sharded x{world}, target replicated"
                                      + ((" (per step: " + ("1 RCCL broadcast of the target, " if s["target_upload"] == "broadcast" else
                                                            "target broadcast by the first fit only, replica reused while the tensor is unchanged; " if s["target_upload"] == "cached" else
                                                            "target uploaded by every rank before the timed region, no broadcast; ")
                                          + "1 all-to-all of the per-shard reverse lists, 1 all-gather"
                                          " of the per-target fit state; measured times and bytes: collective_ms_per_step, collective_traffic_per_step)")
                                         if world > 1 else " (single rank: no collective runs)")},
            "roofline": dict(roofline_of(args.workload, s),
                             executed_mfma_frac=s["achieved"] * TIER_PRODUCTS[tier] / s["peak"],
                             vs_fp32_mfma_peak=s["achieved"] / PEAK_F32_MFMA_TFLOPS,
                             # SURVEY.md section 8(d) prices a hubness-reduced step at TWO passes (4 n_s n_t d flop, what the
                             # reference executes); that figure over the same kernel time, for comparison only -- `achieved` /
                             # `frac` count what this kernel executes
                             reference_work_tflops=(s["achieved"] * (1 + s["shared_sweeps"] / max(s["n_launch"], 1))),
                             note=("algorithmic flop = 2 n_q n_i d per launch, counted ONCE for a launch that serves both search "
                                   "directions (kz_knn_dual): the reference evaluates that distance matrix twice; with the nested sample "
                                   "(large shared sweeps) a launch is TWO dispatches that together visit every pair once -- the sample "
                                   "sweep (kz_knn_cand_h_kernel<..,true,..>, 1 / stride of the query rows) and the main sweep "
                                   "(kz_knn_cand_h64_kernel or kz_knn_cand_h_kernel, the other rows): avg_launch_ms is the sum of their "
                                   "rocprof averages; traffic / clock / pipe-busy are the main sweep's")),
            "shared_sweep": {"launches": s["shared_sweeps"], "reverse_extra_ms_per_step": s["reverse_extra_ms_per_step"],
                             "reverse_events_per_row": s["reverse_events_per_row"], "reverse_escalated_rows": s["reverse_escalated_rows"]},
            "certification_fallback_rows": s["fallback_rows"],
            "escalated_rows": s["escalated_rows"],
            "speculative_rescue_rows": s["spec_rows"],
            "range_research_rows": s["range_rows"],
            "range_group_rows": s["range_group_rows"],
            "rounding_bound_self_check": {"max_err_over_eps": s["max_err_ratio"],
                                          "note": "max |approximate key - exact key| / eps over all re-ranked candidates; the certification needs < 1"},
            "other_kernels_ms": {"finalize_avg": s["finalize_avg_ms"], "fallback_total": s["fallback_total_ms"]},
            "collective_ms_per_step": s["collective_ms_per_step"],
            "collective_traffic_per_step": s["collective_traffic_per_step"],
        }
        if s["check"] is not None:
            line["check"] = s["check"]
    if world == 1 and rank == 0:
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
            del kz
        line["host_api"] = {"value": n_s / t_host, "unit": "queries/s", "ms": t_host * 1e3,
                            "note": "Kiez(...).fit(numpy, numpy).kneighbors(k) -> numpy: includes H2D of both matrices and D2H of the result"}
    if rank == 0 and cpu is not None:
        line["cpu_baseline"] = cpu
    del source_h, target_h
    if world == 1 and not args.no_others:
        others = {}
        # c4 = configuration 4 at its stated size on this one GPU (the N = 1 anchor of `--scaling strong`; ~1.3 s per step: fewer
        # steps); c1g / hard = the same kernels on gaussian and on clustered data (how often the tier chain runs is data dependent)
        for name in ("c1", "c2", "c3", "c4s", "c4", "ns", "c1g", "hard", "cliff", "gmm", "ea15k"):
            if name == args.workload:
                continue
            try:
                # (each workload starts from an empty buffer cache: what the previous one left behind is of the wrong sizes and
                #  only turns this one's first releases into hipFree calls)
                eng.ctx.trim()
                o_steps, o_warm = (min(args.other_steps, 3), min(args.other_warmup, 1)) if name == "c4" else (args.other_steps, args.other_warmup)
                osum, _, _, _ = run_workload(name, eng, comm, dist, rank, world, o_steps, o_warm, check=not args.no_check)
                others[name] = short(osum)
            except Exception as e:  # pragma: no cover  (a secondary workload must never cost the main line)
                others[name] = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            line["other_workloads"] = others
            # one short object for all workloads, moved to the front of the line below: [ms per step, main-kernel ms per launch,
            # roofline fraction, M source queries/s, rows of the check identical / rows checked]
            def brief(ms, r, value, chk):
                return [round(ms, 3), round(r["avg_launch_ms"], 3), round(r["frac"], 4), round(value / 1e6, 3),
                        f'{chk["index_rows_identical"]}/{chk.get("rows_not_knife_edge", chk["rows"])}' if chk else None]
            summ = {args.workload: brief(line["ms_per_step"], line["roofline"], line["value"], line.get("check"))}
            for name, o in others.items():
                summ[name] = brief(o["ms_per_step"], o["roofline"], o["value"], o.get("check")) if "error" not in o else "error"
            line["summary"] = {"columns": ["ms_per_step", "main_kernel_ms_per_launch", "roofline_frac", "M_queries_per_s", "check_rows_identical"], **summ}
            try:
                # float64 near-ties: pairs of index rows 1/64 .. 16 ulps apart, the reference's own order on them as the golden
                # (tests/near_ties.py; DESIGN.md section 5: how often the device's float64 order differs from sklearn's, by gap)
                from tests.near_ties import run_probe
                line["fp64_order_probe"] = run_probe(eng.ctx)
            except Exception as e:  # pragma: no cover
                line["fp64_order_probe"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                # cosine + float32 inputs: rows ordered as the REFERENCE RUN ON THE FLOAT32 INPUTS (sgemm) orders them, rows that
                # differ only inside near-tie groups float32 cannot resolve, rows that differ otherwise (tests/cosine_f32.py)
                from tests.cosine_f32 import run_probe as cosine_probe
                line["cosine_f32_probe"] = cosine_probe(eng.ctx)
            except Exception as e:  # pragma: no cover
                line["cosine_f32_probe"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        sys.stdout.flush()
        emit(line, json_fd, args)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
