"""How fast does the command processor get through chains of tiny dependent kernels?  K graphs of N tiny kernels each,
one stream per graph - the dispatch floor under the pipelined runner (105 launches per yolov8n step)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ultralytics_pro_amd.engine import runtime as R

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 105
for nf in (1, 2, 4, 8):
    ts = [torch.zeros(256, device=dev) for _ in range(nf)]
    graphs = []
    for t in ts:
        def body(t=t):
            for _ in range(N):
                t.add_(1.0)
        body()
        torch.cuda.synchronize()
        g = R.HipGraph()
        g.capture(body, device=dev)
        graphs.append(g)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(nf)]
    def step(i):
        with torch.cuda.stream(lanes[i % nf]):
            graphs[i % nf].replay(dev)
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"in_flight={nf}: {dt * 1e3:.3f} ms per graph of {N} tiny kernels = {dt * 1e6 / N:.2f} us per kernel", flush=True)
