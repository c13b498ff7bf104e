"""Loads per wave of the coarse scoring kernel under three exact pruning schemes, simulated on the oracle's linear memories (CPU, numpy):
  current   k_score_coarse_sb: chunks of 504 placements (63 lanes x 8 nibbles), bound test after every block of 15 features
  pooled    VERDICT r2 item 4, variant B: an upper-bound pre-pass on memories max-pooled over 7 placements at stride 4 (one lookup
            per feature bounds 4 placements whatever the feature's alignment; a dword covers 32 placements, one chunk covers all),
            then the current kernel on the chunks that still hold a live group
  loss2     variant A: the first block of 15 features read from a 2-bit "loss" map (min(4 - response, 3): a dword covers 16 placements,
            chunks of 1008), exact because a capped loss under-estimates the true loss; survivors continue with the nibble memories
The model reproduces the measured kernel (PMC: 63 wave loads per wave on the default scenes; simulated: see profiles/r03_score_prefilter_sim.txt).
usage: python scripts/sim_score_prefilter.py synth|mesh [threshold] [texture]"""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from linemod_pose_estimation_amd import synth
from oracle import oracle as o
kind = sys.argv[1] if len(sys.argv)>1 else "synth"
thr = float(sys.argv[2]) if len(sys.argv)>2 else 92.0
tex = float(sys.argv[3]) if len(sys.argv)>3 else 0.6
W,H,T=640,480,8
if kind=="mesh":
    from linemod_pose_estimation_amd import meshsynth as ms
    bank,_,_,_=ms.load_bank("memoryChip2"); chip=ms.load_mesh("memoryChip2"); cpu=ms.load_mesh("cpu_binary"); views=ms.view_grid()
    frames=[ms.make_scene(chip,views,seed=7000+f,n_instances=3,other_tri=cpu,n_other=2,texture=tex)[0] for f in range(3)]
else:
    bank=synth.make_bank(3000,seed=20250215)
    frames=[synth.make_scene(bank,W,H,seed=3000+f,texture=tex)[0] for f in range(3)]
od=o.OracleDetector(bank)
cid,tarr,farr=bank.classes[0]
L,M=2,2
Wc,Hc=W//2//T,H//2//T; cells=Wc*Hc
rng=np.random.default_rng(0)
tsel=rng.choice(bank.num_templates(),300,replace=False)
tot_cur=tot_A=tot_B=tot_l2=0; n=0; surv_t=0; blocksA=[]
for fr in frames:
    od.match(fr,thr)
    lm=[od.linear_memory(1,m,(H//2,W//2)).astype(np.int32) for m in range(M)]   # [8][T*T][cells]
    flat=[np.concatenate([x.reshape(8,-1), np.zeros((8,cells+64),np.int32)],1) for x in lm]  # zero pad
    for t in tsel:
        feats=[]
        for m in range(M):
            w,h,lv,fb,fc=tarr[(t*L+1)*M+m]
            f=farr[fb:fb+fc]
            e0=((f[:,1]%T)*T+(f[:,0]%T))*cells+(f[:,1]//T)*Wc+(f[:,0]//T)
            feats.append([(m,int(l),int(e)) for (l,e) in zip(f[:,2],e0)])
        w,h=tarr[(t*L+1)*M][0],tarr[(t*L+1)*M][1]
        wf,hf=(w-1)//T+1,(h-1)//T+1
        pos=max(0,min((Hc-hf)*Wc+(Wc-wf)+1,cells))
        if pos==0: continue
        # interleave modalities in groups of 3
        order=[]; i=[0,0]
        while i[0]<len(feats[0]) or i[1]<len(feats[1]):
            for m in range(M):
                for _ in range(3):
                    if i[m]<len(feats[m]): order.append(feats[m][i[m]]); i[m]+=1
        nf=len(order); raw_thr=int(2*nf+thr/100*2*nf+0.5)
        rows=np.stack([flat[m][l][e:e+pos] for (m,l,e) in order])   # [nf][pos]
        # current scheme
        loads=0
        for c0 in range(0,pos,504):
            S=np.zeros(min(504,pos-c0),np.int32)
            for b in range(0,nf,15):
                S+=rows[b:b+15,c0:c0+504].sum(0); loads+=min(15,nf-b)
                rem=nf-min(nf,b+15)
                if not (S>=raw_thr+1-4*rem).any(): break
        tot_cur+=loads
        # loss2: first block from the 2-bit loss map in chunks of 1008 placements; chunks with a survivor continue on the nibble memories
        # (the first block's exact sums are then re-read: the 2-bit values cannot be widened back)
        l2=0
        for c0 in range(0,pos,1008):
            Lb=np.minimum(4-rows[0:15,c0:c0+1008],3).sum(0); l2+=min(15,nf)
            rem=nf-min(nf,15)
            if not ((4*min(15,nf)-Lb)>=raw_thr+1-4*rem).any(): continue
            for c1 in range(c0,min(pos,c0+1008),504):
                S=np.zeros(min(504,pos-c1),np.int32)
                for b in range(0,nf,15):
                    S+=rows[b:b+15,c1:c1+504].sum(0); l2+=min(15,nf-b)
                    rem=nf-min(nf,b+15)
                    if not (S>=raw_thr+1-4*rem).any(): break
        tot_l2+=l2
        # pooled scheme: window 7 stride 4 on flat index from each feature's own e0 aligned down to 4
        G=(pos+3)//4
        B=np.zeros(G,np.int32); la=0; nb=0
        full=[]
        for k,(m,l,e) in enumerate(order):
            q=e//4
            arr=flat[m][l]
            idx=4*(q+np.arange(G))[:,None]+np.arange(7)[None,:]
            full.append(arr[idx].max(1))
        full=np.stack(full)
        alive=True
        for b in range(0,nf,15):
            B+=full[b:b+15].sum(0); la+=min(15,nf-b); nb+=1
            rem=nf-min(nf,b+15)
            if not (B>=raw_thr+1-4*rem).any(): alive=False; break
        tot_A+=la; blocksA.append(nb)
        if alive:
            sg=np.nonzero(B>raw_thr)[0]
            if len(sg):
                surv_t+=1
                chunks=set((4*g)//504 for g in sg)|set(min(pos-1,4*g+3)//504 for g in sg)
                for c in chunks:
                    c0=c*504
                    S=np.zeros(min(504,pos-c0),np.int32)
                    for b in range(0,nf,15):
                        S+=rows[b:b+15,c0:c0+504].sum(0); tot_B+=min(15,nf-b)
                        rem=nf-min(nf,b+15)
                        if not (S>=raw_thr+1-4*rem).any(): break
        n+=1
print(kind,"thr",thr,"tex",tex,"templates",n,"loads/wave current %.1f"%(tot_cur/n),"pooled A %.1f + exact B %.1f = %.1f"%(tot_A/n,tot_B/n,(tot_A+tot_B)/n),"| loss2 %.1f"%(tot_l2/n),"| survivor templates (pooled) %.3f"%(surv_t/n),"blocksA hist",np.bincount(blocksA))
