import os, time, torch
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cpu.max', e)
x = torch.randn(4096, 4096); 
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    x @ x
    t=time.time(); 
    for _ in range(3): x @ x
    print(n, 'threads', round(3*2*4096**3/ (time.time()-t)/1e9,1), 'GFLOP/s')
