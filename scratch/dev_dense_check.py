# throwaway first-light check of the dense path on a GPU (before the PQ symbols exist)
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_c as oc
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "longbow_amd", "liblongbow_gpu.so"))
vp=C.c_void_p
lib.lb_gpu_index_new.restype=vp; lib.lb_gpu_index_new.argtypes=[C.c_int]*3+[C.POINTER(C.c_int)]
lib.lb_gpu_index_add.argtypes=[vp,C.c_int64,vp,vp]
lib.lb_gpu_index_search.argtypes=[vp,C.c_int64,vp,C.c_int,vp,vp]
lib.lb_gpu_last_error.restype=C.c_char_p; lib.lb_gpu_last_error.argtypes=[vp]
lib.lb_gpu_index_set_order.argtypes=[vp,C.c_int]
lib.lb_gpu_index_last_fallbacks.restype=C.c_int64; lib.lb_gpu_index_last_fallbacks.argtypes=[vp]
lib.lb_simd_distance_batch_flat.argtypes=[C.c_int]*3+[vp,vp,C.c_int64,C.c_int,vp]
lib.lb_gpu_index_free.argtypes=[vp]
print("devices", lib.lb_gpu_device_count(), flush=True)
ok=True
def run(metric, order, n, d, nq, k, seed=0):
    global ok
    rng=np.random.default_rng(seed)
    X=rng.random((n,d),dtype=np.float32); Q=rng.random((nq,d),dtype=np.float32)
    st=C.c_int(0)
    h=vp(lib.lb_gpu_index_new(0,d,metric,C.byref(st)))
    assert h, st.value
    lib.lb_gpu_index_set_order(h,order)
    rc=lib.lb_gpu_index_add(h,n,X.ctypes.data,None); assert rc==0,(rc,lib.lb_gpu_last_error(h))
    dist=np.empty((nq,k),np.float32); lab=np.empty((nq,k),np.int64)
    t0=time.time()
    rc=lib.lb_gpu_index_search(h,nq,Q.ctypes.data,k,dist.ctypes.data,lab.ctypes.data); assert rc==0,(rc,lib.lb_gpu_last_error(h))
    t1=time.time()
    fb=lib.lb_gpu_index_last_fallbacks(h)
    oi,od=oc.search_batch(metric,Q,X,k,order=order,nthreads=8)
    same_i=np.array_equal(oi,lab); same_d=np.array_equal(od,dist)
    print(f"metric={metric} order={order} n={n} d={d} nq={nq} k={k}: ids_equal={same_i} dist_equal={same_d} fallbacks={fb} t={t1-t0:.3f}s",flush=True)
    if not (same_i and same_d):
        ok=False
        bad=np.argwhere(oi!=lab)
        print("  first mismatches", bad[:5], flush=True)
        for (a,b) in bad[:3]:
            print("   q",a,"r",b,"oracle",oi[a,b],od[a,b],"gpu",lab[a,b],dist[a,b])
    lib.lb_gpu_index_free(h)
for metric in (0,1,2):
    for order in (0,1):
        run(metric,order,5000,64,3,10)
        run(metric,order,20000,128,64,10)
run(0,0,100,4,1,10)
run(1,0,3000,33,5,7)
run(1,0,3000,33,40,7)
run(2,1,70000,96,32,100)
run(1,0,200000,768,256,100)
# simd batch flat
rng=np.random.default_rng(5)
for d in (3,8,64,768):
  for metric in (0,1,2):
    for order in (0,1):
        X=rng.random((1000,d),dtype=np.float32); q=rng.random(d,dtype=np.float32)
        r=np.empty(1000,np.float32)
        rc=lib.lb_simd_distance_batch_flat(0,metric,order,q.ctypes.data,X.ctypes.data,1000,d,r.ctypes.data); assert rc==0
        e=oc.batch_flat(metric,q,X,order)
        if metric==2: e=-e
        if not np.array_equal(e,r): ok=False; print("simd mismatch",d,metric,order,np.abs(e-r).max())
print("ALL OK" if ok else "FAILURES")
