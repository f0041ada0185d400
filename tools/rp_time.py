import os, sys, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deepfm_amd import _lib
lib=_lib.load()
B,S,V=4096,26,1_000_000
ids=[torch.randint(1,V,(B,),device="cuda",dtype=torch.int64) for _ in range(S)]
ch=4096; i32=dict(dtype=torch.int32,device="cuda")
sp,u,seg,num,err=torch.empty(1,S,ch,**i32),torch.empty(1,S,ch,**i32),torch.empty(1,S,ch+1,**i32),torch.zeros(1,S,**i32),torch.zeros(1,**i32)
ptrs=(C.c_void_p*S)(*[t.data_ptr() for t in ids]); voc=(C.c_int32*S)(*([V]*S))
for _ in range(50):
    lib.dfm_rowplan_build(ptrs,voc,S,B,sp.data_ptr(),u.data_ptr(),seg.data_ptr(),num.data_ptr(),err.data_ptr(),_lib.stream_handle())
torch.cuda.synchronize()
