import numpy as np, struct, sys
def load(fn):
    raw=open(fn,"rb").read()
    nsamp,dw,nsteps,every=struct.unpack("<4I",raw[:16])
    a=np.frombuffer(raw[16:],dtype=np.uint32).reshape(nsamp,3,dw)
    t0=a[:,:,2].astype(np.uint64)|(a[:,:,3].astype(np.uint64)<<32)
    t1=a[:,:,4].astype(np.uint64)|(a[:,:,5].astype(np.uint64)<<32)
    ns=int(np.median(a[:,:,9]))
    l0,l1=t0.min(),t1.max()
    steady=(t0.min(axis=1)>=l0+0.1*(l1-l0))&(t1.max(axis=1)<=l0+0.9*(l1-l0))
    st=a[:,:,16:16+4*ns].reshape(nsamp,3,ns,4)
    role=a[:,:,6]
    d=lambda x,y:(x-y).astype(np.uint32).astype(np.float64)
    c=st[steady[:,None]&(role==2)]
    p=st[steady[:,None]&(role<2)]
    return dict(ns=ns,cwA=d(c[:,:,1],c[:,:,0]),cmid=d(c[:,:,2],c[:,:,1]),cwB=d(c[:,:,3],c[:,:,2]),cwork=d(c[:,1:,0],c[:,:-1,3]),
      pwork=d(p[:,1:,0],p[:,:-1,3]),pwA=d(p[:,:,1],p[:,:,0]),pwB=d(p[:,:,3],p[:,:,2]),pst=d(p[:,:,2],p[:,:,1]), per=d(c[:,1:,3],c[:,:-1,3]))
for fn in sys.argv[1:]:
    r=load(fn); ns=r['ns']
    def seg(x,a,b): return x[:,a:b].mean()/1e3
    print(fn)
    for (a,b) in ((1,20),(20,45),(45,ns-1)):
        print("  steps %2d-%2d: cons fold %6.1f waitA %5.1f sqr %5.1f waitB %4.1f period %6.1f | prod step %6.1f waitA %5.1f store %4.1f waitB %5.1f"%(a,b,seg(r['cwork'],a,b),seg(r['cwA'],a,b),seg(r['cmid'],a,b),seg(r['cwB'],a,b),seg(r['per'],a,b),seg(r['pwork'],a,b),seg(r['pwA'],a,b),seg(r['pst'],a,b),seg(r['pwB'],a,b)))
