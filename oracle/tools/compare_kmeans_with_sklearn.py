import os, sys
os.environ["OPENBLAS_CORETYPE"]=os.environ.get("CT","Haswell")
os.environ["OMP_NUM_THREADS"]="1"
import numpy as np, ctypes, warnings
warnings.filterwarnings("ignore")
import sklearn.cluster._kmeans as skm
from sklearn.cluster import KMeans
L=ctypes.CDLL("/root/repo/oracle/_build/libkmeans_oracle.so")
L.mprg_oracle_kmeans_fit_predict.argtypes=[ctypes.c_void_p,ctypes.c_int,ctypes.c_int,ctypes.c_int,ctypes.c_int,ctypes.c_uint32,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_void_p]
rec=[]
_orig=skm._kmeans_plusplus
def hook(*a,**k):
    c,i=_orig(*a,**k); rec.append(i.copy()); return c,i
skm._kmeans_plusplus=hook
def oracle(X,k,n_init=10):
    X=np.ascontiguousarray(X,dtype=np.float64); D,V=X.shape
    lab=np.zeros(D,np.int32); fl=np.zeros(D,np.int32); pp=np.zeros(n_init*k,np.int32); info=np.zeros(4)
    r=L.mprg_oracle_kmeans_fit_predict(X.ctypes.data,D,V,k,n_init,2,lab.ctypes.data,fl.ctypes.data,pp.ctypes.data,info.ctypes.data)
    assert r==0
    return lab,fl,pp.reshape(n_init,k),info
def sk(X,k):
    rec.clear()
    km=KMeans(n_clusters=k,random_state=2,algorithm="elkan",n_init=10).fit(X)
    return km.predict(X).astype(np.int32), km.labels_.astype(np.int32), np.array(rec), km.inertia_, km.n_iter_
def kmer_counts(seqs,ksz):
    d={}
    for s in seqs:
        for i in range(len(s)-ksz+1):
            d.setdefault(s[i:i+ksz],len(d))
    M=np.zeros((len(seqs),len(d)))
    for j,s in enumerate(seqs):
        for i in range(len(s)-ksz+1): M[j,d[s[i:i+ksz]]]+=1
    return M
def synth(rng,D,Lh,ksz=7):
    root=rng.integers(0,4,Lh); nc=int(rng.integers(2,7))
    cl=[]
    for _ in range(nc):
        y=root.copy(); m=rng.random(Lh)<0.05; y[m]=rng.integers(0,4,m.sum()); cl.append(y)
    seqs=set()
    while len(seqs)<D:
        y=cl[rng.integers(0,nc)].copy(); m=rng.random(Lh)<0.01; y[m]=rng.integers(0,4,m.sum())
        seqs.add("".join("ACGT"[c] for c in y))
    return kmer_counts(sorted(seqs,key=lambda s:hash(s)),ksz)
if __name__=="__main__":
    rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
    n=int(sys.argv[2]) if len(sys.argv)>2 else 60
    stats=dict(n=0,pred=0,fit=0,pp=0,inertia=0,reloc=0)
    for rep in range(n):
        D=int(rng.integers(4,60)); Lh=int(rng.integers(8,120))
        X=synth(rng,D,Lh)
        for k in range(2,min(10,D-1)+1):
            p1,f1,pp1,in1,it1=sk(X,k)
            p2,f2,pp2,info=oracle(X,k)
            stats["n"]+=1
            okp=np.array_equal(p1,p2); okf=np.array_equal(f1,f2); okpp=np.array_equal(pp1,pp2); oki=(in1==info[0])
            stats["pred"]+=okp; stats["fit"]+=okf; stats["pp"]+=okpp; stats["inertia"]+=oki; stats["reloc"]+=int(info[3])&1
            if not (okp and okf and okpp and oki):
                print("MISMATCH D=%d V=%d k=%d pred=%s fit=%s pp=%s inertia=%s (%r vs %r) it %d/%d reloc=%d"%(X.shape[0],X.shape[1],k,okp,okf,okpp,oki,in1,info[0],it1,info[1],int(info[3])))
                if not okpp:
                    bad=[r for r in range(10) if not np.array_equal(pp1[r],pp2[r])]
                    print("   first bad restart",bad[0],pp1[bad[0]],pp2[bad[0]])
    print(stats)
