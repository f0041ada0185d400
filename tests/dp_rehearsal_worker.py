"""Worker of tests/test_gpu_dp_rehearsal.py: one data-parallel rank of the row-sparse step.

Every rank runs on cuda:0 with the gloo backend (a one-GPU box cannot host two RCCL ranks);
the step code is the one bench.py runs under RCCL: graph A -> eager exchange -> graph B.
Rank 0 writes a JSON line with the per-rank parameter digests."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def trace(msg: str) -> None:
    if os.environ.get("DFM_WORKER_TRACE") == "1":
        print(f"[worker {os.environ.get('RANK')}] {msg}", file=sys.stderr, flush=True)


def main():
    use_graph = sys.argv[1] == "graph"
    steps = int(sys.argv[2])
    fused = len(sys.argv) > 3 and sys.argv[3] in ("fused", "sharded")
    sharded = len(sys.argv) > 3 and sys.argv[3] == "sharded"      # field-sharded tables (training/sharded.py)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # DFM_WORKER_ONE_GPU_PER_RANK=1 (multi-GPU boxes): rank r on cuda:r, the layout bench.py --gpus N runs
    local = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("DFM_WORKER_ONE_GPU_PER_RANK") == "1" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # a fourth argument "nccl" (single rank only): the RCCL backend, so that collectives can be captured
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)
    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.rowsparse import RowSparseAdam
    from deepfm_amd.training.step import RowSparseTrainStep
    from tests.helpers import schema_from_fields
    from tools_shared import criteo_fields

    B, V, D = 512, 300, 16          # small vocabulary: the ranks' row lists overlap heavily
    fields = criteo_fields(V, D)
    cfg = ExperimentConfig()
    torch.manual_seed(0)
    with torch.device(dev):
        model = create_model("deepfm", schema_from_fields(fields), cfg)
    model.train()
    model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    shard = None
    hyper = dict(lr=1e-2, l2=1e-5, max_grad_norm=1.0)
    if sharded:
        from deepfm_amd.training.sharded import make_sharded_step
        step, opt, shard = make_sharded_step(model, B, use_graph=use_graph, **hyper)
    elif fused:
        from deepfm_amd.training.fused_step import FusedDeepFMStep
        opt = RowSparseAdam(model, **hyper)
        step = FusedDeepFMStep(model, opt, B, use_graph=use_graph)        # what bench.py runs
    else:
        opt = RowSparseAdam(model, **hyper)
        step = RowSparseTrainStep(model, opt, B, use_graph=use_graph)
    g = torch.Generator(device=dev).manual_seed(100 + rank)          # a different shard per rank
    ids = torch.randint(0, V, (steps + 1, 26, B), generator=g, device=dev, dtype=torch.int64)
    dense = torch.rand((steps + 1, 13, B), generator=g, device=dev)
    labels = (torch.rand((steps + 1, B), generator=g, device=dev) < 0.3).float()
    trace("step built")
    step.load_batch(ids[0], dense[0], labels[0])
    step.capture()
    trace("captured")
    for i in range(steps):
        step.load_batch(ids[i + 1], dense[i + 1], labels[i + 1])
        step.run()
    torch.cuda.synchronize()
    trace("steps done")
    if shard is not None:
        shard.restore_tables()          # every rank gets every owner's tables back
    tables = torch.cat([p.detach().reshape(-1) for p in model.embedding.table_parameters()])
    mine = {"flat": digest(opt.flat_param), "tables": digest(tables), "loss": float(step.loss),
            "moved": float((tables != 0).float().mean())}
    dump = os.environ.get("DFM_REHEARSAL_DUMP")
    if dump:                            # tensors for tolerance comparisons between the DP layouts
        import numpy as np
        sd = opt.state_dict()                      # sharded: every table's moments (after restore_tables)
        moments = torch.cat([sd["state"][n][k].reshape(-1) for n in sorted(sd["state"]) for k in ("exp_avg", "exp_avg_sq")])
        np.savez(f"{dump}.rank{rank}.npz", flat=opt.flat_param.detach().cpu().numpy(),
                 tables=tables.detach().cpu().numpy(), moments=moments.cpu().numpy(), step=sd["step"],
                 loss=float(step.loss))
    out = [None] * world
    dist.all_gather_object(out, mine)
    if rank == 0:
        print("RESULT " + json.dumps(out), flush=True)
    step.release_graphs()
    trace("graphs released")
    dist.destroy_process_group()
    trace("process group destroyed")


if __name__ == "__main__":
    main()
