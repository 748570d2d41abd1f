import numpy as np, itertools
g='gpurun_out/probe_'
def ld(n,dt): return np.fromfile(g+n+'.bin',dtype=dt)
# ---- f32 MFMA chain check
for tag,MN in (('f32a',16),('f32b',32)):
    K=64
    A=ld(tag+'_A',np.float32).reshape(-1,MN,K); B=ld(tag+'_B',np.float32).reshape(-1,K,MN); C=ld(tag+'_C',np.float32).reshape(-1,MN,MN); D=ld(tag+'_D',np.float32).reshape(-1,MN,MN)
    acc=C.copy()
    for k in range(K):
        # fma: compute in float64 exactly? product of two f32 has 48 bits; sum with acc needs more than 53 -> use np.longdouble(64-bit mantissa) not exact either. use python fractions on a sample instead
        pass
    # exact fma emulation via float64: a*b exact in f64 (48 bits), acc+prod in f64 may round (double rounding risk is tiny but nonzero) -> use as approximation and count mismatches
    acc=C.astype(np.float64)
    for k in range(K):
        prod=A[:,:,k,None].astype(np.float64)*B[:,k,None,:].astype(np.float64)
        acc=(acc+prod).astype(np.float32).astype(np.float64)
    mism=(acc.astype(np.float32)!=D).sum()
    print(tag,'seq fma chain mismatches',mism,'of',D.size)
    # no-fma variant: round product first
    acc=C.copy()
    for k in range(K):
        prod=(A[:,:,k,None]*B[:,k,None,:]).astype(np.float32)
        acc=(acc+prod).astype(np.float32)
    print(tag,'seq mul+add mismatches',(acc!=D).sum())
# ---- f16 MFMA
def hyp_eval(A,B,C,order_groups,mode):
    # A:[P,M,K] f64, B:[P,K,N], C:[P,M,N] ; order_groups: list of lists of k; each group summed exactly (f64) then added to acc with one rounding to f32
    acc=C.astype(np.float64)
    for grp in order_groups:
        s=np.zeros_like(acc)
        for k in grp: s=s+A[:,:,k,None]*B[:,k,None,:]
        if mode=='rn': acc=(acc+s).astype(np.float32).astype(np.float64)
        elif mode=='grp_then_add':  # round group sum to f32 first, then add
            acc=(acc+s.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    return acc.astype(np.float32)
for tag,M,K in (('h16',16,32),('h32',32,16)):
    A=ld(tag+'_A',np.float16).reshape(-1,M,K).astype(np.float64); B=ld(tag+'_B',np.float16).reshape(-1,K,M).astype(np.float64)
    C=ld(tag+'_C',np.float32).reshape(-1,M,M); D=ld(tag+'_D',np.float32).reshape(-1,M,M)
    hyps={}
    for bs in (1,2,4,8,16,32):
        if bs>K: continue
        hyps['blk%d_korder'%bs]=[list(range(i,i+bs)) for i in range(0,K,bs)]
    hyps['all_exact']=[list(range(K))]
    # C added last: sum all exact then add C? same as all_exact in exact arithmetic with single rounding.
    for name,grps in hyps.items():
        for mode in ('rn','grp_then_add'):
            r=hyp_eval(A,B,C,grps,mode)
            print(tag,name,mode,'mismatch',(r!=D).sum(),'of',D.size)
    # truncation variant: exact then round toward zero
    ex=C.astype(np.float64)+np.einsum('pmk,pkn->pmn',A,B)
    rn=ex.astype(np.float32)
    # toward zero
    rz=rn.copy(); over=np.abs(rz.astype(np.float64))>np.abs(ex); rz[over]=np.nextafter(rz[over],np.float32(0))
    print(tag,'all_exact RZ mismatch',(rz!=D).sum())
    d=(D.astype(np.float64)-ex); ulp=np.spacing(np.abs(rn)).astype(np.float64)
    print(tag,'err vs exact in ulps: min %.3f max %.3f mean %.3f'%((d/ulp).min(),(d/ulp).max(),(d/ulp).mean()))
# ---- order tests
D=ld('ord16_D',np.float32).reshape(-1,16,16)[:,:,0].reshape(32,32,32)  # [i,j,m] -> result
vals,counts=np.unique(D,return_counts=True); print('ord16 values',dict(zip(vals.tolist(),counts.tolist())))
# for which (i,j,m) is the 1 retained? 
keep=(D==1.0)
print('fraction kept',keep.mean())
# group structure: for fixed j, set of (i,m) that keep
import collections
def grp(k): return k//8
tab=collections.Counter()
for i in range(32):
  for j in range(32):
    for m in range(32):
      if len({i,j,m})<3: continue
      tab[(grp(i)==grp(j),grp(j)==grp(m),grp(i)==grp(m), bool(keep[i,j,m]))]+=1
for k,v in sorted(tab.items()): print(k,v)
Dc=ld('ordc16_D',np.float32).reshape(-1,16,16)[:,:,0].reshape(32,32)
print('ordc16 unique',np.unique(Dc,return_counts=True))
