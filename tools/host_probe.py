import importlib, sys, time, os
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0,ROOT)
pt=importlib.import_module('path-tracing_amd')
s=pt.Scene.load_obj(ROOT+'/models/','Tor.obj',device=0)
W,H,spp=1920,1080,int(sys.argv[1]) if len(sys.argv)>1 else 256
n=W*H
acc=(np.zeros((n,3),np.float32),np.zeros((n,3),np.float32),np.zeros(n,np.int32))
for k in range(3):
    t=time.perf_counter(); s.render_host(W,H,spp,8,accum=acc,want_stats=False); print('render_host', round((time.perf_counter()-t)*1e3,2),'ms',flush=True)
