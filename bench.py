#!/usr/bin/env python3
"""Headline benchmark: train samples/s, DeepFM Criteo-shape (26 sparse x 1M vocab, 13 dense,
embed_dim 16), batch 4096 per GPU, data-parallel over N GPUs of one node.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch already resident in HBM:
fused embedding gather -> FM + DNN forward -> BCE(+L2) -> backward -> row gradients ->
(DP exchange) -> clip + row-wise Adam on touched rows + dense Adam.
Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
`roofline` for the embedding gather kernel (HIP events around every launch of it inside
the timed region), `cpu_baseline` (the numpy oracle's same step on the host cores) and, at
N = 1, `extra_configs`: BASELINE.json configurations 3 (xDeepFM, CIN [128,128,128]) and 4
(AttentionDeepFM, embed_dim 32, 4 heads) timed the same way after the headline, each with the
roofline of its own dominant layer (CIN: algorithmic TFLOP/s against the dense bf16 MFMA peak).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The host driver of this pool supports dmabuf IPC only: without this RCCL (N > 1 ranks) fails in
# hipIpcGetMemHandle.  Must be in the environment before the first HIP call of every rank.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0   # same guide: what a streaming kernel reaches
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (never the 2:1-sparsity figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly")
    ap.add_argument("--autograd", action="store_true", help="torch.autograd step instead of the fused tower kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-timing", action="store_true",
                    help="do not attach HIP events to the gather dispatches (roofline fields become null)")
    ap.add_argument("--dp-mode", choices=("sharded", "replicated"), default="sharded",
                    help="N > 1 ranks: embedding tables sharded by field over the ranks (rows and gradients of the batch "
                         "travel by all-to-all, every row update is local) or replicated on every rank (row lists "
                         "all-gathered, every replica applies every rank's row updates)")
    ap.add_argument("--ids", choices=("uniform", "zipf"), default="uniform",
                    help="id distribution of the synthetic batches: uniform over [1, V) (the headline: worst case for "
                         "caches) or Zipf(1.05) clipped to [1, V) (SURVEY.md 8d's secondary, Criteo-like skew)")
    ap.add_argument("--vocab-profile", choices=("equal", "criteo"), default="equal",
                    help="equal: --vocab ids in every SPARSE field (BASELINE.json's shape); criteo: the 26 vocabulary sizes "
                         "of the public Criteo Kaggle set (3 ... 10 M ids, 33.8 M rows in all), uniform ids inside a field")
    ap.add_argument("--unpacked", action="store_true", help="keep the tables as separate contiguous tensors")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip BASELINE.json configurations 3 and 4 (xDeepFM / AttentionDeepFM) after the headline")
    ap.add_argument("--extra-steps", type=int, default=60, help="timed steps of each extra configuration")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="consecutive training steps captured in one HIP graph (1 under data parallelism); 0 = automatic: "
                         "4..8, chosen so that the timed region is whole graphs plus two timed single steps")
    ap.add_argument("--gather-shape", type=int, default=0,
                    help="tuning aid: force a launch shape of the gather (dfm_gather_set_shape); 0 = automatic")
    ap.add_argument("--rowplan-inline", action="store_true",
                    help="A/B: row plan behind the gather from the staged ids (round 2's order) instead of in front of it on "
                         "the batch record with row-touch workgroups")
    ap.add_argument("--no-plan-lookahead", action="store_true",
                    help="A/B: every step of a multi-step graph builds its own row plan (no dfm_step_apply_plan)")
    ap.add_argument("--tower-mode", type=int, default=-1, choices=(-1, 0, 1, 2),
                    help="arithmetic of the DNN tower's GEMMs (dfm_tower_set_mode): 0 exact fp32 matrix pipe, 1 fp32 forward + "
                         "bf16 x 3 backward, 2 bf16 x 6 (fp32-faithful) forward and backward; -1 = the package default "
                         "(training/step.py::TOWER_MODE_DEFAULT)")
    ap.add_argument("--region-order", choices=("groups-first", "singles-first"), default="groups-first",
                    help="A/B: where the event-timed single steps sit in the timed region")
    ap.add_argument("--gather-samples", type=int, default=32,
                    help="timed gather dispatches wanted for the roofline: those of the timed region plus single steps "
                         "run after it (outside `value`) until this many are collected")
    ap.add_argument("--no-gather-sweep", action="store_true", help="skip the isolated gather batch sweep (roofline.sweep)")
    ap.add_argument("--h2d", action="store_true",
                    help="batches start in HOST memory and go through the packed H2D ring (PCIe-inclusive rate; "
                         "reported in DESIGN.md, never the headline value)")
    return ap.parse_args()


def gather_bytes_per_sample(n_sparse: int, n_dense: int, dim: int) -> int:
    """Algorithmic HBM bytes of the embedding gather: rows + first-order
    scalars + ids + dense values read; field_embeddings (aliased as flat) + first_order written."""
    reads = n_sparse * (dim * 4 + 4 + 8) + n_dense * 4
    writes = (n_sparse + n_dense) * dim * 4 + 4
    return reads + writes


def make_pool(n_batches, n_sparse, n_dense, B, V, seed, device, dist_name="uniform"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if isinstance(V, (list, tuple)):          # one vocabulary per field: uniform ids inside each
        hi = torch.tensor(V, device=device, dtype=torch.float64).view(1, n_sparse, 1)
        u = torch.rand((n_batches, n_sparse, B), generator=g, device=device, dtype=torch.float64)
        ids = (1 + (u * (hi - 1)).floor()).to(torch.int64).clamp_(max=int(max(V)) - 1)
        ids = torch.minimum(ids, (hi - 1).to(torch.int64))
    elif dist_name == "zipf":
        # SURVEY.md section 8(d), secondary distribution: Zipf(s = 1.05) clipped to [1, V) — Criteo-like skew,
        # many duplicates inside a batch (the row plan's skewed-bucket path, cache-friendly gathers)
        z = np.random.default_rng(seed).zipf(1.05, size=(n_batches, n_sparse, B))
        ids = torch.from_numpy(np.clip(z, 1, V - 1).astype(np.int64)).to(device)
    else:
        ids = torch.randint(1, V, (n_batches, n_sparse, B), generator=g, device=device, dtype=torch.int64)
    pad = torch.rand((n_batches, n_sparse, B), generator=g, device=device) < 0.01
    ids.masked_fill_(pad, 0)                                       # 1 % padding ids
    dense = torch.rand((n_batches, n_dense, B), generator=g, device=device)
    labels = (torch.rand((n_batches, B), generator=g, device=device) < 0.25).float()
    return ids, dense, labels


def gather_sweep(model, n_sparse, n_dense, V, D, dev, lib, batches=(4096, 8192, 16384, 32768, 65536), iters=12):
    """The product gather alone on the step's own tables over a batch sweep: where one launch stops being
    latency-structured (ids -> rows -> stores are three dependent HBM round trips per launch) and the
    algorithmic rate reaches its asymptote.  HIP events on every dispatch, 8 rotating batches, uniform ids."""
    import ctypes as C
    from deepfm_amd import _lib
    emb = model.embedding
    g = torch.Generator(device=dev).manual_seed(4242)
    NB, Bmax = 8, max(batches)
    ids = torch.randint(1, V, (NB, n_sparse, Bmax), generator=g, device=dev, dtype=torch.int64)
    dense = torch.rand((NB, n_dense, Bmax), generator=g, device=dev)
    out = []
    for Bs in batches:
        fo = torch.empty(Bs, 1, device=dev)
        fe = torch.empty(Bs, n_sparse + n_dense, D, device=dev)
        fm = torch.empty(Bs, device=dev)
        fsum = torch.empty(Bs, D, device=dev)

        def launch(i):
            nb = i % NB
            inputs = [ids[nb, s, :Bs] for s in range(n_sparse)] + [dense[nb, j, :Bs] for j in range(n_dense)]
            emb.forward_into(inputs, Bs, fo, fe, fm_out=fm, fm_sum=fsum)
        for i in range(3):
            launch(i)
        torch.cuda.synchronize()
        _lib.check(lib.dfm_gather_timing_begin(iters))
        for i in range(iters):
            launch(3 + i)
        us = (C.c_float * iters)()
        got = C.c_int(0)
        _lib.check(lib.dfm_gather_timing_end(us, iters, C.byref(got)))
        ts = [float(us[i]) for i in range(got.value)]
        if not ts:
            continue
        avg = sum(ts) / len(ts)
        algo = gather_bytes_per_sample(n_sparse, n_dense, D) * Bs
        out.append({"batch": Bs, "avg_launch_us": avg, "min_launch_us": min(ts), "launches": len(ts),
                    "achieved_GBps": algo / (avg * 1e-6) / 1e9, "frac": algo / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS})
    return out


def cpu_baseline(model, fields, cfg, hp, ids, dense, labels, seconds):
    """The oracle's row-sparse DeepFM step on the host cores, on a bounded sample of the
    same synthetic batches (kind 'port': numpy fp32; BLAS threads = cores reported)."""
    from oracle import ctr_oracle as O
    params = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()
              if not k.endswith("num_batches_tracked")}
    state = {}
    for k, v in params.items():
        if "running_" in k:
            continue
        state["m/" + k] = np.zeros_like(v)
        state["v/" + k] = np.zeros_like(v)
    ocfg = dict(fm_dim=cfg.feature.fm_embed_dim, hidden_units=cfg.dnn.hidden_units)
    ids_h, dense_h, labels_h = ids.cpu().numpy(), dense.cpu().numpy(), labels.cpu().numpy()
    names_s = [f["name"] for f in fields if f["type"] == "sparse"]
    names_d = [f["name"] for f in fields if f["type"] == "dense"]
    done, t0 = 0, time.perf_counter()
    while True:
        i = done % ids_h.shape[0]
        batch = {n: ids_h[i, j] for j, n in enumerate(names_s)}
        batch.update({n: dense_h[i, j] for j, n in enumerate(names_d)})
        O.deepfm_train_step_rowsparse(fields, params, state, batch, labels_h[i], ocfg, hp, done + 1)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 64:
            break
    B = ids_h.shape[2]
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": done * B / el, "unit": "samples/s", "cores": int(threads), "kind": "port",
            "formulation": "row-sparse numpy port of this package's step (NOT the reference's dense step: dense V x d "
                           "gradients + full-table L2 + dense Adam over 442 M parameters; reference-here: 1.6 K samples/s "
                           "on 8 cores, BASELINE.md section 2)",
            "sample": f"{done} steps of batch {B} (numpy oracle, row-sparse step, {el:.1f} s)",
            "host_cpus": os.cpu_count()}


def build_step(name, V, D, B, dev, args, cin_sizes=None):
    """Model of BASELINE.json's shape + row-sparse optimizer + the fastest step class that takes it."""
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.fused_step import fused_step_class
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    from deepfm_amd.data.synthetic import CRITEO_KAGGLE_CARDINALITIES, criteo_fields, schema_from_fields
    fields = criteo_fields([c + 1 for c in CRITEO_KAGGLE_CARDINALITIES] if args.vocab_profile == "criteo" else V, D)
    cfg = ExperimentConfig()
    cfg.feature.fm_embed_dim = D
    if cin_sizes:
        cfg.cin.layer_sizes = list(cin_sizes)
    torch.manual_seed(0)                      # identical replicas on every rank
    with torch.device(dev):
        model = create_model(name, schema_from_fields(fields), cfg)
    model.train()
    if not args.unpacked:
        model.embedding.pack_tables_()        # 256-B row records: [w2 | w1 m1 v1 | m2 | v2]
    model.embedding.set_grad_mode("rowsparse")
    hp = dict(lr=cfg.training.lr, l2=cfg.feature.embedding_l2_reg, max_grad_norm=cfg.training.gradient_clip_norm)
    cls = None if args.autograd else fused_step_class(model)
    fused = cls is not None
    if dist.is_initialized() and args.dp_mode == "sharded" and fused and not args.unpacked:
        # N ranks (or DFM_FORCE_DP_PATH=1: one rank over RCCL): tables sharded by field, three all-to-alls per step
        from deepfm_amd.training.sharded import make_sharded_step
        step, opt, _ = make_sharded_step(model, B, use_graph=not args.no_graph, **hp)
        return model, opt, step, fields, cfg, hp, fused
    opt = RowSparseAdam(model, lr=hp["lr"], l2=hp["l2"], max_grad_norm=hp["max_grad_norm"])
    step = (cls or RowSparseTrainStep)(model, opt, B, use_graph=not args.no_graph)
    return model, opt, step, fields, cfg, hp, fused


def event_ms(fn, iters, warm=3):
    """Average duration of fn() from HIP events on the current stream (one pair around each call)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in pairs]
    return sum(ts) / len(ts), min(ts)


def extra_config(name, args, dev, lib, ids_dist="uniform"):
    """One of BASELINE.json's configurations 3 / 4 on one GPU: the same timed loop as the headline
    (batches resident in HBM, packed records, HIP graph), then the roofline of the model's own
    interaction layer from HIP events around isolated launches of it at the same shapes.  name "deepfm":
    the headline model again on another id distribution (SURVEY.md 8d's secondary, Zipf), step time only."""
    B, V = args.batch, args.vocab
    D = 32 if name == "attention_deepfm" else (args.dim if name == "deepfm" else 16)
    cin_sizes = [128, 128, 128] if name == "xdeepfm" else None
    model, opt, step, fields, cfg, hp, fused = build_step(name, V, D, B, dev, args, cin_sizes)
    n_sparse, n_dense = 26, 13
    G = 1 if (args.no_graph or (opt.split and not step.exchange_in_body)) else (args.steps_per_graph or 6)
    warm, steps = -(-10 // G) * G, max(args.extra_steps // G, 1) * G                   # whole graph launches
    ids, dense, labels = make_pool(warm + steps, n_sparse, n_dense, B, V, 101, dev, ids_dist)     # every batch used once
    records = step.pack_batches(ids, dense, labels)
    step.load_packed(records[0])
    step.capture(steps_per_graph=G)
    total = warm + steps

    def group(i):
        # every launch announces the record its successor starts with (the row plan of that step is built by this
        # launch's last optimizer kernel: training/step.py), as the headline's loop does
        return [records[i + k] for k in range(G)], (records[i + G] if (G > 1 and i + G < total) else None)

    for i in range(0, warm, G):
        recs, nxt = group(i)
        if G > 1:
            step.run_group(recs, next_record=nxt)
        else:
            step.run_group(recs)
    prepared = G > 1 and bool(getattr(step, "slots", None))
    if prepared:                     # host half of the first timed launch ahead of the synchronisation (see run())
        recs, nxt = group(warm)
        step.prepare_group(recs, next_record=nxt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, total, G):
        if prepared and i == warm:
            step.launch_prepared()
            continue
        recs, nxt = group(i)
        if G > 1:
            step.run_group(recs, next_record=nxt)
        else:
            step.run_group(recs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {
        "workload": (f"{dict(xdeepfm='xDeepFM', attention_deepfm='AttentionDeepFM', deepfm='DeepFM')[name]} synthetic "
                     f"Criteo-shape: {n_sparse} sparse x {V} vocab, {n_dense} dense, embed_dim {D}, batch {B}"
                     + (f", cin.layer_sizes={cin_sizes}" if cin_sizes else
                        (f", attention.num_heads={cfg.attention.num_heads}" if name == "attention_deepfm" else ""))),
        "ids": ids_dist,
        "ms_per_step": el / steps * 1e3, "samples_per_s": steps * B / el, "steps": steps, "warmup": warm,
        "step": type(step).__name__, "hip_graph": not args.no_graph, "steps_per_graph": G,
        "final_loss": float(step.loss.item()),
    }
    if name == "deepfm":
        return out
    # ---- the interaction layer alone, at the step's shapes, on this stream
    fe = step.fe.detach().clone().requires_grad_()
    if name == "xdeepfm":
        layer = model.cin
        flops_f = 2.0 * D * sum(c * h * 39 for c, h in zip(layer.layer_sizes, [39] + list(layer.next_sizes[:-1]))) * B
        flops_b = 2.0 * flops_f
    else:
        layer = model.attention
    g = torch.randn_like(layer(fe).detach())

    def fwd():
        with torch.no_grad():
            layer(fe)

    def fwd_bwd():
        fe.grad = None
        layer(fe).backward(g)
    f_ms, f_min = event_ms(fwd, 20)
    t_ms, t_min = event_ms(fwd_bwd, 20)
    if name == "xdeepfm":
        out["roofline"] = {
            "kernel": "cin_fwd_mfma (whole stack, one launch) / + cin_dgrad_mfma + cin_wgrad_mfma x3 (fwd+bwd)",
            "bound": "mfma", "unit": "TFLOP/s", "peak": BF16_MFMA_PEAK_TFLOPS,
            "mode": {0: "bf16x3 split (parity: 1e-4)", 1: "plain bf16", 2: "fp32 VALU"}[lib.dfm_cin_get_mode()],
            "algorithmic_flops_fwd": flops_f, "algorithmic_flops_fwd_bwd": flops_f + flops_b,
            "fwd_ms": f_ms, "fwd_ms_min": f_min, "fwd_bwd_ms": t_ms, "fwd_bwd_ms_min": t_min,
            "achieved_fwd": flops_f / (f_ms * 1e-3) / 1e12, "frac_fwd": flops_f / (f_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "achieved": (flops_f + flops_b) / (t_ms * 1e-3) / 1e12,
            "frac": (flops_f + flops_b) / (t_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "issued_over_algorithmic": 3.0 if lib.dfm_cin_get_mode() == 0 else 1.0,
            "timer": "HIP events around 20 isolated launches of the layer (forward: repack + 1 kernel) at the step's shapes",
        }
    else:
        io_bytes = 2.0 * 39 * D * 4 * B           # x in + out per sample
        out["roofline"] = {
            "kernel": "field self-attention block (attn_qkv_mfma_fwd: Q|K|V projection + softmax(QK^T)V on the fp32 matrix cores, out GEMM, LayerNorm; + backward)",
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "algorithmic_bytes_fwd": io_bytes, "fwd_ms": f_ms, "fwd_ms_min": f_min, "fwd_bwd_ms": t_ms,
            "fwd_bwd_ms_min": t_min, "achieved": io_bytes / (f_ms * 1e-3) / 1e9,
            "frac": io_bytes / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "timer": "HIP events around 20 isolated launches of the layer at the step's shapes",
        }
    del step, opt, model, records, ids, dense, labels, fe, g
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # DFM_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend — exercises the N>1 code
    # path (two graphs + eager exchange) on a one-GPU box; its timings mean nothing
    rehearsal = os.environ.get("DFM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    json_out = sys.stdout
    if world > 1 or os.environ.get("DFM_FORCE_DP_PATH") == "1":
        # RCCL prints its version banner on stdout when a communicator comes up; the contract is ONE JSON
        # line there, so everything else written to fd 1 goes to stderr and the result to a saved copy
        json_out = os.fdopen(os.dup(1), "w")
        sys.stdout.flush()
        os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif os.environ.get("DFM_FORCE_DP_PATH") == "1":
        # one rank, real RCCL: the N > 1 step structure (graph A -> eager collectives -> graph B) on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    import ctypes as C
    from deepfm_amd import _lib
    from deepfm_amd.training.step import RowSparseTrainStep
    lib = _lib.load()
    from deepfm_amd.training import step as step_mod
    _lib.check(lib.dfm_tower_set_mode(step_mod.TOWER_MODE_DEFAULT if args.tower_mode < 0 else args.tower_mode))
    RowSparseTrainStep.rowplan_first_default = not args.rowplan_inline
    RowSparseTrainStep.plan_lookahead_default = not args.no_plan_lookahead
    _lib.check(lib.dfm_gather_set_shape(args.gather_shape))        # before any capture: the graphs keep the kernel
    B, V, D = args.batch, args.vocab, args.dim
    n_sparse, n_dense = 26, 13
    switches = {k: v for k, v in os.environ.items() if k.startswith("DFM_") or k.startswith("DEEPFM_AMD_")}
    model, opt, step, fields, cfg, hp, fused = build_step("deepfm", V, D, B, dev, args)

    total = args.steps + args.warmup
    pool_v = [f["vocab"] for f in fields if f["type"] == "sparse"] if args.vocab_profile == "criteo" else V
    ids, dense, labels = make_pool(total, n_sparse, n_dense, B, pool_v, 1 + rank, dev, args.ids)
    records = step.pack_batches(ids, dense, labels)      # one record per batch, resident in HBM
    step.load_packed(records[0])
    # several steps per graph need the whole step inside ONE graph: one rank without the split exchange path
    def choose_graph_shape(steps):
        """(steps per graph, timed single steps inside the timed region): the fewest timed singles >= 2 such that the
        other steps fill whole graphs of 4..8 steps (a single step costs ~55 us more than its share of a graph: at the
        driver's --steps 20, 2 + 3 x 6 instead of 4 + 4 x 4 is 6 us per step)."""
        if args.steps_per_graph > 0:
            g = args.steps_per_graph
            t = 2 + (steps - 2) % g if steps >= 2 else 0
            return g, t
        for t in range(2, 10):
            for g in (4, 5, 6, 7, 8):
                if steps - t >= g and (steps - t) % g == 0:
                    return g, t
        return 4, 2 + (steps - 2) % 4 if steps >= 2 else 0
    auto_g, n_timed_region = choose_graph_shape(args.steps)
    spg = 1 if ((opt.split and not step.exchange_in_body) or args.no_graph) else auto_g
    if spg == 1:
        n_timed_region = max(args.steps // 3, 1)
    # Graph capture must succeed on EVERY rank or on none: ranks that replay a graph and ranks that launch eagerly
    # would issue different collective sequences.  A failed capture leaves the step's state restored
    # (RowSparseTrainStep.capture), so the eager fallback starts from the same parameters.
    capture_error = None
    try:
        step.capture(timed_variant=True, steps_per_graph=spg)
    except Exception as exc:      # e.g. a runtime that refuses to capture the collectives
        if not dist.is_initialized():
            raise
        capture_error = f"{type(exc).__name__}: {exc}"
    capture_fallback = None
    if dist.is_initialized():
        ok = torch.tensor([0 if capture_error else 1], device=dev, dtype=torch.int32)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            print(f"[bench rank {rank}] graph capture failed on at least one rank "
                  f"({capture_error or 'not on this one'}); every rank runs the step eagerly", file=sys.stderr, flush=True)
            step.release_graphs()
            step.use_graph = False
            capture_fallback = "eager: " + (capture_error or "capture failed on another rank")

    feed = None
    if args.h2d:
        from deepfm_amd.data.packed import DeviceBatchRing, PackedBatchLoader, PackedColumns
        names_s = [f["name"] for f in fields if f["type"] == "sparse"]
        names_d = [f["name"] for f in fields if f["type"] == "dense"]
        feats = {n: ids[:, j].reshape(-1).cpu().numpy() for j, n in enumerate(names_s)}
        feats.update({n: dense[:, j].reshape(-1).cpu().numpy() for j, n in enumerate(names_d)})
        loader = PackedBatchLoader(PackedColumns(model.schema, feats, labels.reshape(-1).cpu().numpy()), B)
        feed = iter(DeviceBatchRing(loader, dev, depth=4))

    G = step.steps_per_graph if (not args.no_graph and step.use_graph) else 1

    def rec(i):
        return next(feed) if feed is not None else records[i]

    def plan(lo, hi, n_timed):
        """How steps lo..hi-1 are launched: ('group', i) = steps i..i+G-1 in ONE graph launch; ('timed', i) /
        ('single', i) = one step with the row plan and the gather launched eagerly in front of the gather-less copy
        of the graph (the only place HIP events can be attached to the gather's dispatch; ~55 us more than a step
        inside a group).  ``n_timed`` timed single steps come first; steps that do not fill a
        group run as single steps (none when (hi - lo - n_timed) is a multiple of G: choose_graph_shape)."""
        n = hi - lo
        if args.no_graph:
            return [("timed" if k < n_timed else "single", lo + k) for k in range(n)]
        n_timed = min(n_timed, n)
        groups = (n - n_timed) // G
        # groups first, the timed single steps CLOSE the region: the host enqueues their eager launches while the last
        # group is still on the device.  (Round 3 had them open the region, where their ~0.5 ms of device time covered
        # the host-side preparation of the first group launch — three node updates per step + the launch of ~100
        # nodes, ~0.4 ms during which the device sat idle behind the opening synchronisation.  That preparation now
        # happens BEFORE the synchronisation, RowSparseTrainStep.prepare_group: as in a training loop, where launch
        # k + 1 is prepared while launch k runs.)
        out, i = [], lo
        if args.region_order == "singles-first":
            for _ in range(n_timed):
                out.append(("timed", i)); i += 1
        for _ in range(groups):
            out.append(("group", i)); i += G
        if args.region_order != "singles-first":
            for _ in range(n_timed):
                out.append(("timed", i)); i += 1
        while i < hi:
            out.append(("single", i)); i += 1
        return out

    def group_args(p, j, after):
        i = p[j][1]
        nxt = p[j + 1][1] if j + 1 < len(p) else after
        # the launch's last optimizer kernel also sorts the NEXT launch's first batch (training/step.py)
        return [rec(i + k) for k in range(G)], (records[nxt] if (nxt is not None and feed is None) else None)

    def execute(p, after=None, prepared_at=-1):
        """``after``: record index the launch that follows this plan starts with (None: unknown).
        ``prepared_at``: index of the plan's group whose host half ran already (prepare_first)."""
        for j, (kind, i) in enumerate(p):
            if kind == "group":
                if G > 1:
                    if j == prepared_at:
                        step.launch_prepared()
                    else:
                        recs, nxt = group_args(p, j, after)
                        step.run_group(recs, next_record=nxt)
                else:
                    step.run_from(rec(i))
            else:
                step.run_from(rec(i), eager_gather=not args.no_graph)

    def prepare_first(p, after=None):
        """Host half of the plan's first GROUP launch (multi-step graph on resident records); returns its index in the
        plan or -1."""
        if G > 1 and feed is None and getattr(step, "slots", None):
            for j, (kind, _) in enumerate(p):
                if kind == "group":
                    recs, nxt = group_args(p, j, after)
                    step.prepare_group(recs, next_record=nxt)
                    return j
        return -1

    execute(plan(0, args.warmup, 0), after=args.warmup if args.warmup < total else None)
    if G > 1 and args.warmup < 2 * G and feed is None:
        # both instantiated copies of the step graph get one untimed launch (their first launch uploads the exec): W can
        # be smaller than two graphs (the driver's --warmup 5); these steps come on top of the W warm-up steps.  The
        # second one announces the timed region's first record, as every launch announces its successor's.
        first_timed = records[args.warmup] if args.warmup < total else None
        for j in range(2):
            step.run_group([records[k % total] for k in range(G)], next_record=first_timed if j == 1 else None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The gather is a node of the step's graph.  A few steps of the timed region (choose_graph_shape) launch
    # launches it eagerly instead, in front of a gather-less copy of the graph, with HIP start/stop events
    # attached to the dispatch (hipExtLaunchKernel): the same kernel, arguments and position in the step,
    # timed like rocprofv3's kernel trace times it.  (--no-graph: every step is eager and timed.)
    timing = not args.no_gather_timing
    timed_plan = plan(args.warmup, total, n_timed_region if timing else 0)
    n_timed = sum(1 for kind, _ in timed_plan if kind == "timed")
    if n_timed:
        _lib.check(lib.dfm_gather_timing_begin(n_timed))
    prepared_at = prepare_first(timed_plan)     # node updates of the first group launch: host work, nothing enqueued
    prepared = prepared_at >= 0
    if prepared:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    execute(timed_plan, prepared_at=prepared_at)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(step.loss.item())

    us = (C.c_float * max(n_timed, 1))()
    got = C.c_int(0)
    if n_timed:
        _lib.check(lib.dfm_gather_timing_end(us, n_timed, C.byref(got)))
    gather_us = [float(us[i]) for i in range(got.value)]
    n_in_region = len(gather_us)
    # More samples of the same dispatch, AFTER the timed region (not part of `value`): the step goes on on fresh
    # records as single steps whose gather is launched eagerly with events on the dispatch — same kernel, same
    # arguments, same place in the step (behind the previous step's optimizer tail).
    n_after = 0
    if timing and world == 1 and not args.no_graph and step.use_graph and args.gather_samples > n_in_region:
        n_after = args.gather_samples - n_in_region
        ids2, dense2, labels2 = make_pool(n_after, n_sparse, n_dense, B, pool_v, 977 + rank, dev, args.ids)
        rec2 = step.pack_batches(ids2, dense2, labels2)
        _lib.check(lib.dfm_gather_timing_begin(n_after))
        for i in range(n_after):
            if G > 1 and getattr(step, "cont_slots", None):
                # a graph launch whose last apply plans for the timed step: its gather then follows the same launch
                # (step_apply_plan) as every graph-node gather does
                step.run_group([records[(i * G + k) % total] for k in range(G)], next_record=rec2[i])
            step.run_from(rec2[i], eager_gather=True)
        torch.cuda.synchronize()
        us2 = (C.c_float * n_after)()
        _lib.check(lib.dfm_gather_timing_end(us2, n_after, C.byref(got)))
        gather_us += [float(us2[i]) for i in range(got.value)]
        del ids2, dense2, labels2, rec2
    gather_avg_s = (sum(gather_us) / len(gather_us)) * 1e-6 if gather_us else float("nan")
    algo_bytes = gather_bytes_per_sample(n_sparse, n_dense, D) * B
    achieved = algo_bytes / gather_avg_s / 1e9 if gather_us else None
    sweep = gather_sweep(model, n_sparse, n_dense, V, D, dev, lib) if (
        world == 1 and timing and args.vocab_profile == "equal" and not args.no_gather_sweep) else None

    pmc = None
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_gather_pmc.json") for r in (3, 2)) if os.path.exists(q)), "")
    if pmc_path and B == 4096 and V == 1_000_000 and D == 16:
        with open(pmc_path) as fh:
            pmc = json.load(fh)
    if rank == 0:
        out = {
            "metric": "train samples/sec DeepFM Criteo-shape bs4096",
            "value": args.steps * B * world / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "ids": args.ids,
                "vocab_profile": args.vocab_profile,
                "workload": f"DeepFM synthetic Criteo-shape: {n_sparse} sparse x {V} vocab, {n_dense} dense, "
                            f"embed_dim {D}, batch {B}/GPU; step = fwd + BCE + L2 + bwd + clip + "
                            "row-wise Adam on touched rows (lazy L2) + dense Adam",
                "global_batch": B * world,
                "parallelism": f"dp{world}" + ("" if not dist.is_initialized() else
                                                f" ({'field-sharded tables, 3 all-to-alls + 1 small all-gather per step' if step.exchange_in_body else 'replicated tables, one grouped all-gather per step'})"),
                "hip_graph": bool(step.use_graph and not args.no_graph),
                "steps_per_graph": G,
                "region": ("graph launches of %d steps, then %d single step(s) with the gather dispatch event-timed" % (G, n_timed)
                           if G > 1 else "single steps") +
                          ("; the node updates of the first launch (host work) precede the opening synchronisation, as "
                           "every later launch's overlap the launch before it" if prepared else ""),
                "capture_fallback": capture_fallback,
                "rowplan": ("inside the previous step's apply launch for steps 2.. of a graph; first step: "
                            if getattr(step, "_plan_sets", None) else "") +
                           ("in front of the gather, on the batch record, + row touch" if getattr(step, "rowplan_first", False)
                            else "behind the gather, from the staged ids"),
                **({"dp_layout": "field-sharded tables (a deviation from north_star's replicated tables + all-reduce: "
                                 "DESIGN.md section 6 — replicated tables top out near 3.5x at 8 GPUs)"}
                   if dist.is_initialized() and step.exchange_in_body else {}),
                "input": "host memory -> pinned staging -> H2D ring (PCIe-inclusive)" if args.h2d else "resident in HBM",
                "step": "fused tower kernels (no autograd)" if fused else "torch.autograd over the HIP ops",
                **({"rehearsal": "gloo, all ranks on cuda:0 (not a measurement)"} if rehearsal else {}),
                "table_layout": "separate tensors" if args.unpacked else "packed 256-B row records",
                "final_loss": loss,
            },
            "roofline": {
                "kernel": f"emb_fwd_pair<{D},{n_sparse},{n_dense}> (fused gather of all {n_sparse + n_dense} fields, staged inputs)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                "frac_of_achievable_6p3": achieved / HBM_ACHIEVABLE_GBS if achieved else None,
                # counter bytes (static profile, see `rocprof`) over the time measured in THIS run
                "frac_traffic": (pmc["traffic_bytes_per_launch"] / gather_avg_s / 1e9 / HBM_PEAK_GBS) if (pmc and gather_us) else None,
                "target_frac": 0.60, "target_met": bool(achieved and achieved / HBM_PEAK_GBS >= 0.60),
                "floor_note": "one B=4096 launch is a chain of three dependent HBM round trips (ids -> rows -> stores) plus "
                              "~3 us of launch/completion: measured floor ~8 us = 0.28 (DESIGN.md 7r2); the sweep shows the asymptote",
                "sweep": sweep,
                # NOT measured in this run: PMC counters need their own rocprofv3 passes (see `rocprof`)
                "traffic": None,
                "traffic_from_profile": pmc["traffic_bytes_per_launch"] if pmc else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "avg_launch_us": gather_avg_s * 1e6 if gather_us else None,
                "min_launch_us": min(gather_us) if gather_us else None,
                "launches_timed": len(gather_us),
                "launches_timed_in_region": n_in_region, "launches_timed_after_region": n_after,
                "timer": f"HIP start/stop events attached to the gather dispatch (hipExtLaunchKernel, on the launch stream): "
                         f"{n_in_region} step(s) of the timed region are single steps whose gather is launched eagerly in front of a "
                         f"gather-less copy of the step's graph (the other launches are graphs of {G} step(s) with the gather as a "
                         f"node), {n_after} more such steps follow the region",
                "rocprof": pmc,
            },
        }
        out["config"]["env_switches"] = switches          # every DFM_* variable seen at run time
        out["config"]["cin_mode"] = lib.dfm_cin_get_mode()
        out["config"]["tower_mode"] = {0: "exact fp32 MFMA (forward and backward)",
                                       1: "forward exact fp32 MFMA; backward GEMMs bf16 x 3 split on the bf16 MFMA pipe",
                                       2: "bf16 x 6 on the bf16 MFMA pipe, forward and backward: every fp32 operand split "
                                          "exactly into three bf16 values, six partial products, fp32 accumulate "
                                          "(2^-23 per product: fp32-faithful)"}[lib.dfm_tower_get_mode()]
        out["config"]["tower_mode_id"] = int(lib.dfm_tower_get_mode())
        out["config"]["tower_planes"] = bool(getattr(step, "x6", False))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, fields, cfg, hp, ids[:8], dense[:8], labels[:8],
                                               args.cpu_seconds)
        if world == 1 and not args.no_extra_configs and not args.h2d:
            del step, opt, model, records
            torch.cuda.empty_cache()
            out["extra_configs"] = [extra_config(n, args, dev, lib) for n in ("xdeepfm", "attention_deepfm")]
            if args.ids == "uniform":        # the headline model on the secondary id distribution
                out["extra_configs"].append(extra_config("deepfm", args, dev, lib, ids_dist="zipf"))
        print(json.dumps(out), file=json_out, flush=True)


def run():
    """main() with a teardown that cannot hang: graphs that hold captured RCCL kernels are dropped BEFORE the
    communicator goes, whatever main() raised (training/step.py::release_all_graphs)."""
    try:
        main()
    finally:
        if dist.is_available() and dist.is_initialized():
            from deepfm_amd.training.step import release_all_graphs
            release_all_graphs()
            dist.destroy_process_group()


if __name__ == "__main__":
    run()
