import numpy as np
a=np.fromfile("gpurun_out/stamps.bin",dtype=np.uint64)
nl=a.size//64
MG=a[3*nl*16:].reshape(-1,16).astype(np.int64)
P=a[:2*nl*16].reshape(2,nl,16).astype(np.int64)
for ph,n in ((0,5),(1,4)):
    s=P[ph]
    start = MG[:,0] if ph==0 else MG[:,13]
    idx=[2+j for j in range(2*n)]
    ok=(s[:,2]>start)&(s[:,3+2*(n-1)]-start<100000)&np.all(np.diff(s[:,idx],axis=1)>=0,axis=1)
    s2=s[ok]; st=start[ok]
    print("phase",ph,"DE waves",ok.sum(),"of",nl)
    print("  start->rows of try0 arrived: %d"%(s2[:,2]-st).mean())
    for i in range(n):
        print("  try %d arith %d then wait %s"%(i,(s2[:,3+2*i]-s2[:,2+2*i]).mean(), ("%d"%(s2[:,4+2*i]-s2[:,3+2*i]).mean()) if i+1<n else "-"))
