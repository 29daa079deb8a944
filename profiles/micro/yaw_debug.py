import math, os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import motion_primitive_library_amd as m
from oracle import oracle as O
from test_gpu_yaw_pin import oracle_of, threshold_world, make_env
wd = threshold_world(m, n_each=20000, seed=9)
nodes=wd["nodes"]; U=wd["U"]; n=nodes.shape[1]; nU=U.shape[0]
ref = O.expand(oracle_of(wd), nodes, threads=os.cpu_count(), want_state=False)
ref2 = O.expand(oracle_of(wd), nodes, threads=os.cpu_count(), want_state=False, ref=True)
print("oracle vs _ref differ:", int(np.count_nonzero(ref["status"]!=ref2["status"])))
e = make_env(m, wd)
got = e.expand(nodes, want_state=False)
print("host-path dense diff:", int(np.count_nonzero(got["status"]!=ref["status"])), e.yaw_pin_stats())
fr = e.upload_frontier(nodes); slots = e.alloc_slots(n, want_state=False, want_iters=False)
e.expand_resident(fr, slots); e.synchronize(); res = slots.download()
bad = np.nonzero(res["status"]!=ref["status"])[0]
print("resident dense diff:", bad.size, e.yaw_pin_stats())
L = e.expand_lists(nodes, want_state=False, stride=32)
cnt_ref = ((ref["status"].reshape(n,nU)==1)|(ref["status"].reshape(n,nU)==2)).sum(axis=1)
print("lists count diff nodes:", int(np.count_nonzero(L["count"]!=cnt_ref)))
for s in bad[:10]:
    k, ci = divmod(int(s), nU)
    vx, vy = nodes[2,k], nodes[3,k]; yaw=nodes[8,k]; u=U[ci]
    def dec(vx,vy,y):
        sn=math.sqrt(vx*vx+vy*vy); return vx/sn*math.cos(y)+vy/sn*math.sin(y)
    y0=yaw; yT=(0.0+u[2]*1.0)+yaw
    d0=dec(vx,vy,y0); dT=dec(vx+u[0],vy+u[1],yT); cl=math.cos(0.5)
    print(k, ci, u, "dev", res["status"][s], "ref", ref["status"][s], "d0-cl %.3e dT-cl %.3e" % (d0-cl, dT-cl), "half", k>=n//2)
e.close()
