import os, time, threading
N=6*1024**3; T=8; path='/tmp/fswrite.bin'
buf=os.urandom(1<<20)*64   # 64 MiB
fd=os.open(path, os.O_WRONLY|os.O_CREAT|os.O_TRUNC)
def w(t):
    per=N//T; off=t*per; done=0
    while done<per:
        k=os.pwrite(fd, buf[:min(len(buf),per-done)], off+done); done+=k
t0=time.time(); th=[threading.Thread(target=w,args=(t,)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
dt=time.time()-t0; os.close(fd); print("pwrite x%d: %.2f GB/s (%.1f s for %.1f GB)"%(T, N/dt/1e9, dt, N/1e9)); os.remove(path)
