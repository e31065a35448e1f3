import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
import numpy as np
from permon_amd import problems as P
g = 2000
rp, ci, va = P.laplace2d_csr(g, g); n = g*g
x = np.ones(n)
for nt in (1, 4, 8, 16, 32, 64, 128):
    code = "import os,sys;sys.path.insert(0,%r);import numpy as np;from oracle import oracle as O;from permon_amd import problems as P;rp,ci,va=P.laplace2d_csr(%d,%d);A=O.Csr(%d,%d,rp,ci,va);x=np.ones(%d);t=O.time_spmv(A,x,reps=5,omp=True);print('threads',os.environ['OMP_NUM_THREADS'],'spmv GB/s %%.1f'%%((12*va.size+20*%d)/t/1e9))" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), g, g, n, n, n, n)
    env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PROC_BIND="spread")
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
