import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deepfm_amd.config import ExperimentConfig
from deepfm_amd.models import create_model
from deepfm_amd.training.rowsparse import RowSparseAdam
from deepfm_amd.training.fused_step import FusedDeepFMStep
from tests.helpers import schema_from_fields
from tools_shared import criteo_fields
import bench
B,V,D=4096,1_000_000,16
fields=criteo_fields(V,D); cfg=ExperimentConfig()
torch.manual_seed(0)
with torch.device("cuda"):
    model=create_model("deepfm", schema_from_fields(fields), cfg)
model.train(); model.embedding.pack_tables_(); model.embedding.set_grad_mode("rowsparse")
opt=RowSparseAdam(model, lr=1e-3, l2=1e-5, max_grad_norm=1.0)
step=FusedDeepFMStep(model,opt,B,use_graph=True)
ids,dense,labels=bench.make_pool(64,26,13,B,V,1,torch.device("cuda"))
recs=step.pack_batches(ids,dense,labels)
step.load_packed(recs[0]); step.capture()
for i in range(20): step.run_from(recs[i%64])
torch.cuda.synchronize()
N=300
t0=time.perf_counter()
for i in range(N): step.run_from(recs[i%64])
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print(f"host loop {1e6*(t1-t0)/N:.1f} us/step, with sync {1e6*(t2-t0)/N:.1f} us/step")
tg=tr=0.0
for i in range(N):
    a=time.perf_counter(); step._record=recs[i%64]; step._gather(); b=time.perf_counter(); step.graph_a.replay(); c=time.perf_counter()
    tg+=b-a; tr+=c-b
torch.cuda.synchronize()
print(f"host: gather launch {1e6*tg/N:.1f} us, graph replay {1e6*tr/N:.1f} us")
# host cost with an idle GPU queue: sync before every call
tg=tr=0.0
for i in range(100):
    torch.cuda.synchronize(); a=time.perf_counter(); step._record=recs[i%64]; step._gather(); b=time.perf_counter(); step.graph_a.replay(); c=time.perf_counter()
    tg+=b-a; tr+=c-b
print(f"host (idle queue): gather launch {1e6*tg/100:.1f} us, graph replay {1e6*tr/100:.1f} us")
