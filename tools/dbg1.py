import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import fast_solver_lippmann_schwinger_amd as lsfc
from oracle import lsfc_oracle as o
import cases
n=16; x,h=cases.grid(n,False); k=8*np.pi
X,Y,Z=o.grid3d(x,x,x)
for flags in (0,4):
    M=lsfc.buildFastConvolution3D(x,x,x,X,Y,Z,h,k,o.gaussian_bump,flags=flags)
    s=M.working_symbol()
    print(flags, M.pipeline, np.isfinite(s).sum(), s.size, np.abs(s[np.isfinite(s)]).max())
    y=M*o.random_vector(n**3); print(np.isfinite(y).sum())
