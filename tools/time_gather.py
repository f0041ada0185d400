#!/usr/bin/env python3
"""Times the product gather (dfm_embedding_forward / _staged through the module) for every launch shape
and a batch sweep, on the headline tables (26 x 10^6 packed 256-B records + 13 dense fields, D = 16).
Run under rocprofv3 --kernel-trace and read the per-kernel durations with tools/ktrace_groups.py:
    rocprofv3 --kernel-trace --stats -d gpurun_out/tg -- python3 tools/time_gather.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepfm_amd import _lib  # noqa: E402
from deepfm_amd.config import ExperimentConfig  # noqa: E402
from deepfm_amd.models import create_model  # noqa: E402
from deepfm_amd.data.synthetic import schema_from_fields  # noqa: E402
from tools_shared import criteo_fields  # noqa: E402


def main():
    V, D, S, ND = 1_000_000, 16, 26, 13
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    sweep = [4096, 8192, 16384, 32768, 65536]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    with torch.device(dev):
        model = create_model("deepfm", schema_from_fields(criteo_fields(V, D)), ExperimentConfig())
    emb = model.embedding
    emb.pack_tables_()
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(1)
    NB, Bmax = 8, max(sweep)
    ids = torch.randint(1, V, (NB, S, Bmax), generator=g, device=dev, dtype=torch.int64)
    dense = torch.rand((NB, ND, Bmax), generator=g, device=dev)
    labels = torch.rand((NB, Bmax), generator=g, device=dev)
    st_ids = torch.zeros(S, Bmax, dtype=torch.int64, device=dev)
    st_dense = torch.zeros(ND, Bmax, device=dev)
    st_lab = torch.zeros(Bmax, device=dev)
    for B in sweep:
        fo = torch.empty(B, 1, device=dev)
        fe = torch.empty(B, S + ND, D, device=dev)
        fm = torch.empty(B, device=dev)
        fsum = torch.empty(B, D, device=dev)
        stage = [st_ids[i, :B] for i in range(S)] + [st_dense[j, :B] for j in range(ND)]
        for shape in ([2, 3, 4, 5] if B == 4096 else [4, 5]):
            _lib.check(lib.dfm_gather_set_shape(shape))
            for staged in (False, True):
                for i in range(iters):
                    nb = i % NB
                    if staged:
                        src = [ids[nb, s].data_ptr() for s in range(S)] + [dense[nb, j].data_ptr() for j in range(ND)]
                        emb.forward_staged(src, stage, B, fo, fe, fm_out=fm, fm_sum=fsum,
                                           extra_src_ptr=labels[nb].data_ptr(), extra_dst=st_lab[:B])
                    else:
                        inputs = [ids[nb, s, :B] for s in range(S)] + [dense[nb, j, :B] for j in range(ND)]
                        emb.forward_into(inputs, B, fo, fe, fm_out=fm, fm_sum=fsum)
                torch.cuda.synchronize()
                # a marker so that the trace separates plain / staged runs of the same kernel
                torch.zeros(3 if staged else 2, device=dev)
    _lib.check(lib.dfm_gather_set_shape(0))
    print("ok")


if __name__ == "__main__":
    main()
