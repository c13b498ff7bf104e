#!/usr/bin/env python3
"""Garbage through the host-side entry points that take caller arrays (no GPU): lmx_cluster_matches, lmx_merge_raw, lmx_merge_gathered,
lmx_bank_add_class, lmx_bank_set_normal_lut -- within the ABI contract (array lengths as stated), values arbitrary (NaN, huge, negative,
out-of-range ids).  Every call must return a status; run under scripts/sanitize_host.sh's library (LD_PRELOAD the ASan runtime,
LMX_SO_PATH=build/asan/liblmx.so) so that a bad read or an overflow is a report.  usage: python scripts/fuzz_abi_inputs.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
from linemod_pose_estimation_amd import _lib
from linemod_pose_estimation_amd.detector import cluster_matches, merge_raw, MATCH_DTYPE, RAW_MATCH_DTYPE
rng=np.random.default_rng(1)
L=_lib.lib()
st={}
for it in range(3000):
    n=int(rng.integers(0,200)); nt=int(rng.integers(1,50))
    m=np.zeros(n, MATCH_DTYPE)
    m["x"]=rng.integers(-3000,3000,n); m["y"]=rng.integers(-3000,3000,n)
    m["similarity"]=rng.uniform(0,100,n).astype(np.float32)
    if it%5==0: m["similarity"][:n//2]=np.nan
    lo,hi=(-5,nt+5) if it%3==0 else (0,nt)
    m["template_id"]=rng.integers(lo,hi,n) if n else 0
    if it%7==0 and n: m["template_id"][0]=2**31-1
    dists=rng.uniform(0.3,1.2,nt); 
    if it%11==0: dists[0]=np.nan
    if it%13==0: dists[0]=1e30
    rects=rng.integers(-10,400,(nt,4)).astype(np.int32)
    if it%17==0: rects[0]=[2**31-1,2**31-1,2**31-1,2**31-1]
    step=int(rng.choice([0,1,8,10,-3,2**30])) if it%4==0 else 10
    try:
        c,mem=cluster_matches(m,dists,rects,step,float(rng.choice([0.5,0.0,-1.0])),float(rng.choice([0.1,0.0,1e-30])),int(rng.integers(-1,4)))
        st["ok"]=st.get("ok",0)+1
    except _lib.LmxError as e:
        st[e.status]=st.get(e.status,0)+1
print("cluster_matches", st)
st={}
for it in range(2000):
    n=int(rng.integers(0,300))
    r=np.zeros(n, RAW_MATCH_DTYPE)
    for k in r.dtype.names:
        if r.dtype[k].kind=='f': r[k]=rng.uniform(-10,110,n).astype(np.float32)
        else: r[k]=rng.integers(-5,50,n)
    nf=int(rng.integers(1,6))
    try:
        out=merge_raw(r)
        st["ok"]=st.get("ok",0)+1
    except _lib.LmxError as e:
        st[e.status]=st.get(e.status,0)+1
    except TypeError as e:
        print("sig", e); break
print("merge_raw", st)

from linemod_pose_estimation_amd import synth, NativeBank  # noqa: E402
from linemod_pose_estimation_amd.detector import merge_gathered  # noqa: E402
# 1. merge_gathered on garbage blocks
st={}
rec=np.dtype(RAW_MATCH_DTYPE).itemsize
for it in range(3000):
    R=int(rng.integers(1,5)); K=int(rng.integers(0,40)); nf=int(rng.integers(1,5))
    bb=64+K*rec
    blocks=rng.integers(0,256,R*bb,dtype=np.uint8)
    hdr=blocks.view(np.uint8)
    for r in range(R):
        h=np.frombuffer(blocks[r*bb:r*bb+64].tobytes(), np.uint32).copy()
        mode=it%4
        if mode==0: h[0]=rng.integers(0,50); h[1]=rng.integers(0,K+3); h[2]=rng.integers(0,100)
        elif mode==1: h[1]=2**32-1
        blocks[r*bb:r*bb+64]=np.frombuffer(h.tobytes(),np.uint8)
    try:
        merge_gathered(blocks,R,bb,K,nf,cap_total=int(rng.integers(0,300)))
        st['ok']=st.get('ok',0)+1
    except _lib.LmxError as e:
        st[e.status]=st.get(e.status,0)+1
print("merge_gathered",st)
# 2. bank_add_class with garbage arrays
st={}
bank=synth.make_bank(3, seed=4, size_range=(24.0,40.0))
for it in range(3000):
    nb=NativeBank.create(bank.T, bank.modalities)
    cid,t,f=bank.classes[0]
    t=t.copy(); f=f.copy()
    k=it%6
    if k==0: t[rng.integers(0,len(t)), rng.integers(0,5)]=rng.choice([-1,0,2**31-1,-2**31,1000000])
    elif k==1: f[rng.integers(0,len(f)), rng.integers(0,3)]=rng.choice([-1,8,2**31-1,-2**31,70000])
    elif k==2: t[:,3]=rng.integers(-5,len(f)+5,len(t))
    elif k==3: t[:,4]=rng.integers(-5,200,len(t))
    elif k==4: t=t[:rng.integers(0,len(t))]
    n_pyr=int(rng.choice([len(t)//(2*2), 0, min(1,len(t)//4), -3]))
    tt=np.ascontiguousarray(t,np.int32); ff=np.ascontiguousarray(f,np.int32)
    st_=L.lmx_bank_add_class(nb.h, b"x", n_pyr, tt.ctypes.data_as(C.POINTER(C.c_int32)), ff.ctypes.data_as(C.POINTER(C.c_int32)), len(ff)) if k!=5 else L.lmx_bank_add_class(nb.h, b"x", n_pyr, tt.ctypes.data_as(C.POINTER(C.c_int32)), ff.ctypes.data_as(C.POINTER(C.c_int32)), int(rng.integers(0,len(ff)+1)))
    st[st_]=st.get(st_,0)+1
print("bank_add_class",st)
# 3. normal lut garbage
st={}
for it in range(300):
    nb=NativeBank.from_bank(bank)
    lut=rng.integers(0,256,8000,dtype=np.uint8) if it%2 else rng.choice(np.array([0,1,2,4,8,16,32,64,128],np.uint8),8000)
    s_=L.lmx_bank_set_normal_lut(nb.h, lut.ctypes.data)
    st[s_]=st.get(s_,0)+1
print("set_normal_lut",st)
print("abi input fuzz ok")
