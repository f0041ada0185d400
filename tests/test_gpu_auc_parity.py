"""GPU: "AUC vs ref" of BASELINE.json — DeepFM trained by this package's fast step (row-sparse Adam,
fused tower, HIP graphs) on a learnable synthetic task against the REFERENCE trained on the same task
(tests/golden/auc_reference.json, generated in the build container by tools/make_auc_golden.py from
the reference's own layer classes, dense Adam, CPU).  Optimizer trajectories differ by construction
(lazy row updates, a different dropout stream, another batch-order RNG is NOT used: the epoch
permutations are shared), so this is statistical parity: the test-set AUC after every epoch within
0.02 of the reference's, and the final log-loss within 0.03."""
import json
import os

import numpy as np
import pytest
import torch

import tools_shared_auc as T
from tests.helpers import GOLDEN, schema_from_fields
from tools_shared import criteo_fields

pytestmark = pytest.mark.gpu


def test_deepfm_auc_matches_the_reference_run():
    from sklearn.metrics import log_loss, roc_auc_score

    from deepfm_amd.config import ExperimentConfig
    from deepfm_amd.models import create_model
    from deepfm_amd.training.fused_step import FusedDeepFMStep
    from deepfm_amd.training.rowsparse import RowSparseAdam
    with open(os.path.join(GOLDEN, "auc_reference.json")) as fh:
        ref = json.load(fh)
    assert (ref["n_train"], ref["n_test"], ref["batch"], ref["epochs"]) == (T.N_TRAIN, T.N_TEST, T.BATCH, T.EPOCHS)
    ids, dense, labels = T.make_task()
    assert abs(float(labels.mean()) - ref["label_rate"]) < 1e-9          # same task as the reference saw
    fields = criteo_fields(T.VOCAB, T.DIM)
    cfg = ExperimentConfig()                                              # reference defaults (config.py)
    torch.manual_seed(0)
    model = create_model("deepfm", schema_from_fields(fields), cfg).cuda().train()
    model.embedding.pack_tables_()
    model.embedding.set_grad_mode("rowsparse")
    opt = RowSparseAdam(model, lr=cfg.training.lr, l2=cfg.feature.embedding_l2_reg,
                        max_grad_norm=cfg.training.gradient_clip_norm)
    step = FusedDeepFMStep(model, opt, T.BATCH, use_graph=True)
    d_ids = torch.from_numpy(ids).cuda().t().contiguous()               # (26, N)
    d_dense = torch.from_numpy(dense).cuda().t().contiguous()            # (13, N)
    d_labels = torch.from_numpy(labels).cuda()
    step.load_batch(d_ids[:, :T.BATCH], d_dense[:, :T.BATCH], d_labels[:T.BATCH])
    start = {k: v.clone() for k, v in model.state_dict().items()}
    step.capture()                                                       # side-effect free
    for k, v in model.state_dict().items():
        assert torch.equal(v, start[k]), k

    def evaluate():
        model.eval()
        scores = []
        with torch.no_grad():
            for s in range(T.N_TRAIN, T.N_TRAIN + T.N_TEST, T.BATCH):
                batch = {f"C{j + 1}": d_ids[j, s:s + T.BATCH].contiguous() for j in range(T.N_SPARSE)}
                batch.update({f"I{j + 1}": d_dense[j, s:s + T.BATCH].contiguous() for j in range(T.N_DENSE)})
                scores.append(model.predict(batch).view(-1).cpu().numpy())
        model.train()
        sc = np.concatenate(scores)
        y = labels[T.N_TRAIN:]
        return float(roc_auc_score(y, sc)), float(log_loss(y, np.clip(sc, 1e-7, 1 - 1e-7)))

    history = []
    for epoch in range(T.EPOCHS):
        order = torch.from_numpy(T.epoch_order(epoch)).cuda()
        for k in range(T.N_TRAIN // T.BATCH):
            idx = order[k * T.BATCH:(k + 1) * T.BATCH]
            step.load_batch(d_ids[:, idx], d_dense[:, idx], d_labels[idx])
            step.run()
        history.append(evaluate())
    print("auc/logloss per epoch:", history, "reference:", [(h["auc"], h["logloss"]) for h in ref["history"]])
    for (auc, ll), h in zip(history, ref["history"]):
        assert abs(auc - h["auc"]) < 0.02, (auc, h)
    assert abs(history[-1][1] - ref["history"][-1]["logloss"]) < 0.03
    assert history[-1][0] > history[0][0] > 0.65
