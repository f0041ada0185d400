#!/usr/bin/env python3
"""AUC of the REFERENCE on a learnable synthetic task -> tests/golden/auc_reference.json.

Run in the build container only (needs /root/reference).  The reference's layer classes are imported
as in tools/make_golden.py (by file path; deepfm/models/__init__ needs the uninstalled `dacite`), the
DeepFM model is composed with the reference's attribute names and formula (deepfm.py:20-42), and the
training loop is the body of Trainer._train_epoch / evaluate (trainer.py:59-70, 212-240, 262-285):
BCEWithLogitsLoss + get_l2_reg_loss (base.py:78-83), clip_grad_norm_, Adam(lr), eval-mode sigmoid
scores, sklearn AUC / log-loss (metrics.py:9-18).  Hyper-parameters are the reference defaults
(config.py: lr 1e-3, embedding_l2_reg 1e-5, gradient_clip_norm 1.0, hidden [256,128,64], dropout 0.1).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as G  # noqa: E402  (reference layer classes + schema helpers)
import tools_shared_auc as T  # noqa: E402
from sklearn.metrics import log_loss, roc_auc_score  # noqa: E402


class RefDeepFM(nn.Module):
    """deepfm.py:13-42 + base.py:27-83 composed from the reference's layer classes."""

    def __init__(self, schema):
        super().__init__()
        self.embedding = G.RefEmbedding(schema, fm_embed_dim=T.DIM)
        self.fm = G.RefFM()
        self.dnn = G.RefDNN(schema.total_embedding_dim, [256, 128, 64], "relu", 0.1, True)
        self.output_linear = nn.Linear(self.dnn.output_dim, 1)

    def forward(self, batch):
        fo, fe, fl = self.embedding(batch)
        return fo + self.fm(fe) + self.output_linear(self.dnn(fl))

    def l2(self, lam):
        return lam * sum(p.pow(2).sum() for p in self.embedding.parameters())


def main():
    torch.set_num_threads(8)
    fields = G.criteo_fields(T.VOCAB, T.DIM)
    schema = G.to_schema(fields)
    ids, dense, labels = T.make_task()
    torch.manual_seed(0)
    model = RefDeepFM(schema)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.BCEWithLogitsLoss()

    def batch_of(idx):
        b = {f"C{j + 1}": torch.from_numpy(ids[idx, j]) for j in range(T.N_SPARSE)}
        b.update({f"I{j + 1}": torch.from_numpy(dense[idx, j]) for j in range(T.N_DENSE)})
        return b

    def evaluate():
        model.eval()
        scores = []
        with torch.no_grad():
            for s in range(T.N_TRAIN, T.N_TRAIN + T.N_TEST, T.BATCH):
                idx = np.arange(s, s + T.BATCH)
                scores.append(torch.sigmoid(model(batch_of(idx))).squeeze(1).numpy())
        sc = np.concatenate(scores)
        y = labels[T.N_TRAIN:]
        return float(roc_auc_score(y, sc)), float(log_loss(y, np.clip(sc, 1e-7, 1 - 1e-7)))

    history = []
    for epoch in range(T.EPOCHS):
        model.train()
        order = T.epoch_order(epoch)
        tot = 0.0
        for k in range(T.N_TRAIN // T.BATCH):
            idx = order[k * T.BATCH:(k + 1) * T.BATCH]
            logits = model(batch_of(idx)).squeeze(1)
            loss = crit(logits, torch.from_numpy(labels[idx])) + model.l2(1e-5)
            opt.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            tot += float(loss)
        auc, ll = evaluate()
        history.append(dict(epoch=epoch, train_loss=tot / (T.N_TRAIN // T.BATCH), auc=auc, logloss=ll))
        print(history[-1], flush=True)
    out = dict(task="tools_shared_auc.make_task(seed=2024)", n_train=T.N_TRAIN, n_test=T.N_TEST, batch=T.BATCH,
               epochs=T.EPOCHS, vocab=T.VOCAB, dim=T.DIM, label_rate=float(labels.mean()), history=history,
               generator="tools/make_auc_golden.py (reference layer classes, dense Adam, CPU fp32)")
    with open(os.path.join(ROOT, "tests", "golden", "auc_reference.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
