"""Probe: pieces of the config-3 forward at small batch replayed from a captured HIP graph vs eager launches.
usage: python tools/graph_probe.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.manual_seed(1023)
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
for blk in m.model.visual.transformer.resblocks:
    torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
m = m.eval().to(dev)
g = torch.Generator().manual_seed(123)


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def try_graph(name, fn, n):
    eager = timeit(fn, n)
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        graph = timeit(gr.replay, n)
        print(f"  {name}: eager {eager:.3f} ms, graph {graph:.3f} ms", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"  {name}: eager {eager:.3f} ms, capture FAILED: {str(e)[:160]!r}", flush=True)
        torch.cuda.synchronize()


for B in [int(a) for a in sys.argv[1:]] or [1, 8, 50]:
    vid = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev).bfloat16()
    title = synth_tokens(B, 77, g).to(dev)
    comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
    for _ in range(3):
        m(vid, title, comments)
    pk = m._pack()
    n = 200 if B <= 8 else 60
    print(f"B={B}", flush=True)
    which = os.environ.get("PIECE", "vision")
    if which == "vision":
        try_graph("vision tower", lambda: pk["visual"].forward(vid), n)
    elif which == "text":
        try_graph("text tower (titles + comments, ragged)", lambda: pk["text"].forward(title, ids_b=comments.reshape(-1, 77)), n)
    elif which == "cam":
        fv = torch.randn(B, 512, device=dev); fc = torch.randn(B * 5, 512, device=dev)
        try_graph("CAM", lambda: pk["cam"].forward(fv, fc, comments), n)
    else:
        m.overlap_towers = which == "full_overlap"
        try_graph(f"whole forward ({which})", lambda: m(vid, title, comments), n)
