import sys, time, torch
sys.path.insert(0, "/root/repo")
from vtc_amd.host import model as HM
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.manual_seed(1023)
m = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
m.compute_dtype = torch.bfloat16
from vtc_amd.host.datasets import synth_tokens
g = torch.Generator().manual_seed(123)
B = 256
vis = torch.randn(B, 3, 224, 224, generator=g).to(dev).bfloat16()
title = synth_tokens(B, 77, g).to(dev)
comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
for _ in range(3): m(vis, title, comments)
torch.cuda.synchronize()
def run(nstreams, K=12):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    models = [m]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for i in range(K):
        if nstreams == 1:
            outs.append(m(vis, title, comments))
        else:
            st = streams[i % nstreams]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(m(vis, title, comments))
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for ns in (1, 2, 3, 1, 2):
    print(ns, "streams:", round(run(ns), 3), "ms/step", flush=True)
