import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cumf_als_amd import als, datagen
from oracle import pyoracle as po
for (m,n,nnz,nt,f) in [(300,200,6000,700,20),(400,300,40000,3000,20),(400,300,40000,3000,100)]:
    r = datagen.synth_ratings(m,n,nnz,nt,seed=1); d=r.numpy(); lam=0.05
    th0,x0 = po.init_factors(m,n,f)
    for solver in ("cg","lu"):
        a,b=th0.copy(),x0.copy(); rm_o,log_o = po.do_als(d,a,b,m,n,f,lam,5,solver=solver)
        a64,b64=th0.copy(),x0.copy(); rm_64,log_64 = po.do_als(d,a64,b64,m,n,f,lam,5,solver=solver,dtype=np.float64)
        th,x,rm,log = als.do_als(d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indices"],
                                    d["csc_indptr"], d["csc_data"], d["coo_row"], d["test_row"], d["test_col"],
                                    d["test_data"], m, n, f, r.nnz, r.nnz_test, lam, 5, 1, 1, 0,
                                    thetat_init=th0, xt_init=x0, solver=solver, return_log=True)
        print(m,n,nnz,f,solver,"gpu-o32 log diff", np.abs(log-log_o).max(0), "o64-o32", np.abs(log_64-log_o).max(0),
              "theta diff gpu-o32", np.abs(th-a).max(), "o64-o32", np.abs(a64-a).max())
