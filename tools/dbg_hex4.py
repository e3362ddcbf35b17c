import sys,os
sys.path.insert(0,os.getcwd()); sys.path.insert(0,os.path.join(os.getcwd(),'oracle')); sys.path.insert(0,os.path.join(os.getcwd(),'tests'))
import numpy as np, ninpol_amd, ninpol_oracle as O
from ninpol_amd import mesh as M
m = M.hex_mesh(4); M.attach_fields(m,"u",perm="ALH",neumann_plane=(2,0.0),seed=11)
I=ninpol_amd.Interpolator(device=0); I.load_mesh(mesh_obj=m)
o=O.OracleInterpolator("port",threads=4); o.load_mesh(m)
W,_=I.interpolate("u","gls"); Wo,_=o.interpolate("u","gls")
W=W.toarray(); Wo=Wo.toarray()
bp=np.asarray(I.grid.boundary_points)
for p in range(W.shape[0]):
    e=np.abs(W[p]-Wo[p]).max()
    if e>1e-10: print(p, bool(bp[p]), e, W[p][W[p]!=0][:8], Wo[p][Wo[p]!=0][:8])
