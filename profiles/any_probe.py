import sys, json
sys.path.insert(0, '.')
import numpy as np, ctypes as C
from indelminer_amd import capi, synth
L=300
refs, rd = synth.simulate(seed=8, ref_len=1_000_000, coverage=30, read_len=L, isize_mean=900, isize_sd=50, isize_min=700, isize_max=1100)
cand = synth.candidates(rd); n=len(cand["index"])
ctx=capi.Context(0); ctx.set_reference([refs[0].tobytes()]); ctx.expect_read_length(L)
stride=(L+3)//4*4
bases=np.zeros((n,stride),np.uint8); bases[:,:L]=cand["bases"]
flat=np.concatenate([bases.reshape(-1),np.zeros(16,np.uint8)])
d_bases=capi.DevBuf(ctx,flat.nbytes).upload(flat); d_off=capi.DevBuf(ctx,8*n).upload(np.arange(n,dtype=np.int64)*stride)
d_len=capi.DevBuf(ctx,4*n).upload(np.full(n,L,np.int32)); d_tid=capi.DevBuf(ctx,4*n).upload(np.zeros(n,np.int32))
d_anchor=capi.DevBuf(ctx,4*n).upload(cand["anchor"].astype(np.int32)); d_range=capi.DevBuf(ctx,4*n).upload(cand["range_max"].astype(np.int32))
d_res=capi.DevBuf(ctx,512*n)
batch=capi.DevBatch(n,d_bases.ptr,d_off.ptr,d_len.ptr,d_tid.ptr,d_anchor.ptr,d_range.ptr,d_res.ptr,None,None,None)
Lb=capi.lib(); tm=capi.Timer(ctx)
for g in (2, 12, 61):
    P=capi.params(numgaps=g); ts=[]
    for _ in range(4):
        tm.start(ctx.stream); ctx._check(Lb.im_dev_realign(ctx.h,C.byref(P),C.byref(batch),ctx.stream)); tm.stop(ctx.stream); ts.append(tm.elapsed_ms())
    print(sys.argv[1] if len(sys.argv)>1 else "full", "g", g, "n", n, "ms", round(float(np.median(ts[1:])),3), "range_max", int(cand["range_max"][0]), flush=True)
