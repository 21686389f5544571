import numpy as np, os, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import ref
import synth
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e: print(f, "n/a")
L=ref.lib()
n_ch=4096
sig,_=synth.dtmf_channels(n_ch,160*20,seed=3)
frames=np.ascontiguousarray(sig.reshape(n_ch,20,160).transpose(1,0,2))
states=[L.glue_dtmf_rx_new(None,0,0) for _ in range(n_ch)]
for th in (1,2,4,8,16,32,64,128,256):
    r,loops,dt=ref.timed_baseline(lambda l: ref.mt_rx(ref.MT_DTMF,states,frames,l,th), frames.size, 0.7)
    print(th, "threads: %.1f Msamples/s (%d loops, %.2f s) = %.1f per thread"%(r/1e6,loops,dt,r/1e6/th))
