import torch, time
A,B=torch.cuda.Stream(),torch.cuda.Stream()
def run(two):
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.cuda.stream(A): torch.cuda._sleep(4_000_000)
    with torch.cuda.stream(B if two else A): torch.cuda._sleep(4_000_000)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)*1e3
run(True)
print("one stream %.2f ms, two streams %.2f ms" % (run(False), run(True)))
import os
print({k:v for k,v in os.environ.items() if 'HIP' in k or 'HSA' in k or 'GPU_' in k or 'AMD' in k})
