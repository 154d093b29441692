import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import __graft_entry__ as ge
qn = ge.load_package()
from oracle import qn_oracle as qo
for (m,n) in [(1030,513),(1040,528),(1026,600),(1100,700),(2000,515),(1030,1030),(1025,513),(1030,515),(1030,515)]:
    rng=np.random.default_rng(3)
    a=rng.standard_normal((m,n))/np.sqrt(n); c=rng.standard_normal(m); x0=rng.standard_normal(n)
    obj=qn.LogSumExp(a,c,0.05); ev=obj(x0); f,g=qo.LogSumExpOracle(a,c,0.05)(x0)
    print(m,n, abs(ev.f()-f), np.linalg.norm(ev.g()-g))
