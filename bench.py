"""bench.py -- headline benchmark of the VTC retrieval forward/eval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one forward pass of BASELINE.json configs[1] over one batch resident in HBM:
PretrainedCLIP_finaltf (CLIP ViT-B/32 image tower + CLIP text tower over 1 title + 5 comments +
Context Adapter Module + batch similarity), B = 256 pairs per GPU, bf16 operands / fp32 accumulate,
synthetic random pixels and tokens, random-init weights of the real architecture.
`value` = pairs encoded per second over the whole job (all ranks; weak scaling: B per GPU fixed).

Extra objects on the same JSON line:
  roofline      the dominant kernel class (the bf16 MFMA GEMM: gemm_phased_kernel and gemm_kernel<bf16>):
                achieved = sum(2MNK) / sum(kernel time), both measured live with HIP events on the
                launch stream inside the timed region (vtc_prof_*); peak = 2.5 PFLOP/s dense bf16;
                traffic = HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json).
  cpu_baseline  the oracle (plain PyTorch fp32 restatement of the reference) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only).
  extra         secondary measurements of the same path: config 3 (8-frame TimeSformer + CAM)
                pairs/s, and the N x N sweep (sim + R@1/5/10 both directions) in ms at N = 10k and 50k,
                sharded over the ranks with one RCCL all-gather + one all-reduce when N > 1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def synth_tokens(n, ctx, gen, empty_frac=0.0):
    """[SOT, t_1..t_L, EOT, 0...] rows, L ~ U{1..75}; a fraction is the empty string."""
    out = torch.zeros(n, ctx, dtype=torch.int64)
    lens = torch.randint(1, ctx - 1, (n,), generator=gen)
    empty = torch.rand(n, generator=gen) < empty_frac
    lens[empty] = 0
    toks = torch.randint(1, 49406, (n, ctx), generator=gen)
    pos = torch.arange(ctx)[None]
    out = torch.where((pos >= 1) & (pos <= lens[:, None]), toks, out)
    out[:, 0] = 49406
    out[torch.arange(n), lens + 1] = 49407
    return out


def prof_run(fn, stream_ptr):
    from vtc_amd import _lib as L
    lib = L.lib()
    lib.vtc_prof_begin()
    fn()
    n = len(L.PROF_CLASSES)
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    L.check(lib.vtc_prof_end(stream_ptr, ms, cnt, work), "vtc_prof_end")
    return {name: dict(ms=ms[i], launches=cnt[i], work=work[i]) for i, name in enumerate(L.PROF_CLASSES)}


def barrier_sync(world):
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, world, device):
    import torch.distributed as dist
    if world == 1:
        return t
    x = torch.tensor([t], dtype=torch.float64, device=device)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x.item())


def cpu_baseline(kind_pairs=96):
    """Oracle (kind 'port') on the host cores: config 2 forward at B = kind_pairs, best of 2."""
    from oracle import arch as A
    from oracle import model_ref as M
    # a 1-GPU box exposes every host core but grants ~16 of them: oversubscribing is far slower
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    a = A.VIT_B32
    sd = A.synth_model(a, 1, "clip_finaltf")
    B = kind_pairs
    vis = A.synth_pixels((B, 3, 224, 224), 2)
    title = A.synth_tokens(B, a, 3)
    comments = A.synth_tokens(B * 5, a, 4, empty_frac=0.1).reshape(B, 5, -1)
    best = 1e30
    with torch.no_grad():
        for _ in range(2):
            t0 = time.perf_counter()
            M.pretrained_clip_finaltf(vis, title, comments, sd, a, "text")
            best = min(best, time.perf_counter() - t0)
    out = dict(value=round(B / best, 3), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
               sample=f"oracle fp32 forward of config 2 at B={B} (1 title + 5 comments per pair), best of 2, {best:.2f} s")
    # the second half of the metric: 10k x 10k sim + R@1/5/10, both directions, on the same cores (the oracle's
    # literal restatement of RecallAtK.compute, model/metric.py:137-161, fp32 numpy)
    import numpy as np
    from oracle import eval_ref as E
    rng = np.random.default_rng(123)
    n = 10000
    va = rng.standard_normal((n, 512)).astype(np.float32)
    va /= np.linalg.norm(va, axis=1, keepdims=True)
    tb = va + 0.05 * rng.standard_normal((n, 512)).astype(np.float32)
    tb /= np.linalg.norm(tb, axis=1, keepdims=True)
    t0 = time.perf_counter()
    E.recall_at_k(va, tb, [1, 5, 10])
    E.recall_at_k(tb, va, [1, 5, 10])
    out["sweep_10000_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel class, from the committed rocprofv3 PMC passes of this
    same command (profiles/*_traffic.json, written by tools/summarize_profile.py; collected in separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs with the gfx950 correction).  None when no such file exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        return round(float(json.load(open(files[-1]))["traffic_bytes_per_launch"]), 1)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-extra", action="store_true", help="skip config 3 and the sweep")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--sweep-n", type=int, default=10000)
    ap.add_argument("--stress-n", type=int, default=50000, help="second sweep size (0 = skip)")
    ap.add_argument("--hipgraph", action="store_true", help="also time the step replayed from a captured HIP graph (last, opt-in)")
    args = ap.parse_args()

    from vtc_amd import dist as vdist
    rank, local, world = vdist.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.set_grad_enabled(False)

    from vtc_amd import _lib as L
    from vtc_amd import ops
    from vtc_amd.host import model as HM
    from vtc_amd.host.metric import RecallAtK

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B = args.batch
    gen = torch.Generator().manual_seed(123 + rank)           # data seed (tests/test_pretrained_clip.py:46)
    torch.manual_seed(1023)                                    # weight seed (train.py:34)

    # ---- config 2: image + title + 5 comments (CAM) ---------------------------------------
    m2 = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt="text", branch_to_adapt_val="text",
                                   init_from_avg=True)
    # a trained adapter is not an identity: give the CAM's zero-initialised projections small weights
    for blk in m2.final_transformer.resblocks:
        torch.nn.init.normal_(blk.attn.out_proj.weight, std=0.02)
        torch.nn.init.normal_(blk.mlp.c_proj.weight, std=0.02)
    m2 = m2.eval().to(device)
    m2.compute_dtype = cdt
    vis = torch.randn(B, 3, 224, 224, generator=gen).to(device).to(cdt)   # BASELINE: pixels cast to bf16 for bf16 runs
    title = synth_tokens(B, 77, gen).to(device)
    comments = synth_tokens(B * 5, 77, gen, empty_frac=0.1).reshape(B, 5, 77).to(device)

    def step2():
        return m2(vis, title, comments)

    for _ in range(args.warmup):
        step2()
    stream_ptr = torch.cuda.current_stream().cuda_stream
    barrier_sync(world)
    lib = L.lib()
    # timed region 1: K steps, nothing but the forward passes -> `value`
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step2()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device)
    # timed region 2: the same K steps with every kernel launch bracketed by HIP events on the launch
    # stream (vtc_prof_*; ~8 % slower because of the 2 x 300 event records per step) -> `roofline`
    # (towers run back to back here, not on two streams: kernels that share the chip would each look slower)
    m2.overlap_towers = False
    lib.vtc_prof_begin()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step2()
    m2.overlap_towers = type(m2).overlap_towers
    n = len(L.PROF_CLASSES)
    pms, pcnt, pwork = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    L.check(lib.vtc_prof_end(stream_ptr, pms, pcnt, pwork), "vtc_prof_end")     # synchronises the stream
    barrier_sync(world)
    dt_instr = max_over_ranks(time.perf_counter() - t1, world, device)
    prof = {name: dict(ms=pms[i], launches=int(pcnt[i]), work=pwork[i]) for i, name in enumerate(L.PROF_CLASSES)}
    assert torch.isfinite(out[2]).all()
    value = world * B * args.steps / dt

    gk = "gemm_bf16" if args.dtype == "bf16" else "gemm_f32"
    g = prof[gk]
    achieved = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
    roofline = dict(bound="mfma", achieved=round(achieved, 1), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                    traffic=pmc_traffic(), kernel=f"{args.dtype} GEMM: gemm_phased_kernel + gemm_kernel<{args.dtype}> (all epilogues)",
                    launches_per_step=g["launches"] // args.steps, avg_launch_us=round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                    flop_per_launch=round(g["work"] / max(1, g["launches"]) / 1e9, 3))
    breakdown = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}

    result = {
        "metric": "video-text pairs encoded/sec (config 2: image+title+5 comments, CAM, ViT-B/32)", "value": round(value, 1),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs/pretrained_clip_comments_attention.jsonc PretrainedCLIP_finaltf forward: "
                               f"{B} pairs/GPU/step = {B} images 224x224 + {B} titles + {5 * B} comments (77 tokens) + CAM + sim",
                   "pairs_per_gpu": B, "parallelism": f"dp{world} (one process per GPU, no collective in the encode path)"},
        "roofline": roofline,
        "kernel_ms_per_step": breakdown,
        "ms_per_step_with_events": round(1e3 * dt_instr / args.steps, 3),
    }

    extra = {}
    if not args.no_extra:
        try:   # the extras never cost the main line: an exception is recorded in extra["error"]
            # ---- config 2 again with the ragged text tower (tokens after EOT are not computed; identical
            # outputs -- tests/test_gpu_towers.py::test_ragged_text_tower_equals_dense).  Reported apart from
            # `value`, which does exactly the reference's work (all 77 positions of every sequence).
            kr = max(2, args.steps // 2)
            from vtc_amd import towers as _tw
            _tw.TEXT_RAGGED = True
            for _ in range(2):
                step2()
            barrier_sync(world)
            t0 = time.perf_counter()
            kr = max(2, args.steps // 2)
            for _ in range(kr):
                step2()
            barrier_sync(world)
            dtr = max_over_ranks(time.perf_counter() - t0, world, device)
            _tw.TEXT_RAGGED = False
            n_tok = int((torch.cat([title, comments.reshape(-1, 77)]).argmax(-1) + 1).sum().item())
            extra["config2_ragged_text_pairs_per_s"] = round(world * B * kr / dtr, 1)
            extra["config2_ragged_text_ms_per_step"] = round(1e3 * dtr / kr, 3)
            extra["config2_ragged_text_tokens_computed_frac"] = round(n_tok / (6 * B * 77), 4)
            # ---- config 3: 8-frame TimeSformer video + title + 5 comments ------------------------
            del m2
            torch.cuda.empty_cache()
            m3 = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
            for blk in m3.model.visual.transformer.resblocks:       # trained temporal_fc is not zero
                torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
            m3 = m3.eval().to(device)
            m3.compute_dtype = cdt
            B3 = min(B, 256)      # SURVEY 8d C3 at B = 256 (the config's own batch_size of 50 under-fills the chip)
            vid = torch.randn(B3, 8, 3, 224, 224, generator=gen).to(device).to(cdt)
            t3, c3 = title[:B3].contiguous(), comments[:B3].contiguous()
            for _ in range(2):
                m3(vid, t3, c3)
            barrier_sync(world)
            k3 = max(2, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(k3):
                o3 = m3(vid, t3, c3)
            barrier_sync(world)
            dt3 = max_over_ranks(time.perf_counter() - t0, world, device)
            extra["config3_timesformer_pairs_per_s"] = round(world * B3 * k3 / dt3, 1)
            extra["config3_ms_per_step"] = round(1e3 * dt3 / k3, 2)
            extra["config3_pairs_per_gpu"] = B3
            m3.overlap_towers = False
            p3 = prof_run(lambda: m3(vid, t3, c3), stream_ptr)
            g3 = p3[gk]
            extra["config3_gemm_tflops"] = round(g3["work"] / (g3["ms"] * 1e-3) / 1e12, 1)
            extra["config3_kernel_ms"] = {k: round(v["ms"], 3) for k, v in p3.items() if v["launches"]}
            adapter_sd = {k: v.detach().clone() for k, v in m3.state_dict().items()
                          if k.startswith("final_transformer.") or k in ("mask_embedding", "model.logit_scale")}
            del m3, vid
            torch.cuda.empty_cache()
            # ---- the stress config's encoder (BASELINE configs[4]): 16-frame TimeSformer + title + 5 comments ------
            class _TSF16(HM.PretrainedCLIP_TimeSformer_finaltf):
                nframes = 16
            m16 = _TSF16(model_type="ViT-B/32", branch_to_adapt_val="text")
            for blk in m16.model.visual.transformer.resblocks:
                torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
            m16 = m16.eval().to(device)
            m16.compute_dtype = cdt
            B16 = min(B, 128)
            vid16 = torch.randn(B16, 16, 3, 224, 224, generator=gen).to(device).to(cdt)
            t16, c16 = title[:B16].contiguous(), comments[:B16].contiguous()
            for _ in range(2):
                m16(vid16, t16, c16)
            barrier_sync(world)
            k16 = max(2, args.steps // 3)
            t0 = time.perf_counter()
            for _ in range(k16):
                m16(vid16, t16, c16)
            barrier_sync(world)
            dt16 = max_over_ranks(time.perf_counter() - t0, world, device)
            extra["stress_timesformer16_pairs_per_s"] = round(world * B16 * k16 / dt16, 1)
            extra["stress_timesformer16_pairs_per_gpu"] = B16
            del m16, vid16
            torch.cuda.empty_cache()

            # ---- adapter-only training step (configs/pretrained_clip_comments_attn_frozen.jsonc:
            # batch 128, frozen towers, clip_loss, Adam amsgrad): forward + backward + update of the CAM on the HIP path
            from vtc_amd.host.adapter_train import AdapterTrainer
            trn = AdapterTrainer(adapter_sd)
            Bt = 128
            gt = torch.Generator().manual_seed(7)
            tfv, tft = torch.randn(Bt, 512, generator=gt).to(device), torch.randn(Bt, 512, generator=gt).to(device)
            tfc = torch.randn(5, Bt, 512, generator=gt).to(device)
            temp = (torch.rand(Bt, 5, generator=gt) < 0.1).to(device)
            tskip = (torch.rand(Bt, generator=gt) > 0.5).to(device)
            for _ in range(3):
                trn.step(tfv, tft, tfc, temp, tskip)
            barrier_sync(world)
            t0 = time.perf_counter()
            kt = 20
            for _ in range(kt):
                tl = trn.step(tfv, tft, tfc, temp, tskip)
            barrier_sync(world)
            dtt = max_over_ranks(time.perf_counter() - t0, world, device)
            extra["adapter_train_step_ms"] = round(1e3 * dtt / kt, 3)
            extra["adapter_train_batch"] = Bt
            extra["adapter_train_loss_after"] = round(float(tl), 4)
            del trn

            # ---- sweep: N x N sim + R@1/5/10 both directions, sharded by query rows ---------------
            # N = 10k (BASELINE configs[3]) and the 50k stress size (configs[4]); embeddings drawn directly
            # (SURVEY 8d), planted positives so that R@K < 1.
            for N in [n for n in (args.sweep_n, args.stress_n) if n > 0]:
                lo, hi = vdist.shard_bounds(N, rank, world)
                g2 = torch.Generator().manual_seed(123)
                va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
                noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
                tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2), dim=-1)
                va_l, tb_l = va[lo:hi].to(device), tb[lo:hi].to(device)
                precs = [("exact", L.SWEEP_EXACT), ("f32", L.SWEEP_F32), ("bf16x3", L.SWEEP_BF16X3)] + ([("bf16", L.SWEEP_BF16)] if N == args.stress_n else [])
                for prec_name, prec in precs:
                    # caller-owned workspace, as a serving loop would hold it (a fresh multi-GiB torch allocation per call
                    # can land on a hipMalloc / cache flush: seen as one 45 ms call in three)
                    lib = L.lib()
                    need = max(lib.vtc_l2_topk_workspace_bytes(N, hi - lo, 512, prec, 0), lib.vtc_l2_topk_workspace_bytes(N, N, 512, prec, 0) if world == 1 else 0,
                               lib.vtc_l2_topk_bidir_workspace_bytes(N, N, 512, prec, 0) if world == 1 else 0)
                    sweep_ws = ops.workspace(need, device)
                    vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=sweep_ws)      # warm-up
                    barrier_sync(world)
                    t0 = time.perf_counter()
                    reps = 3
                    for _ in range(reps):
                        r_ab, r_ba = vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=sweep_ws)
                    barrier_sync(world)
                    dts = max_over_ranks(time.perf_counter() - t0, world, device) / reps
                    del sweep_ws
                    extra[f"sweep_{N}_{prec_name}_ms"] = round(1e3 * dts, 3)
                    extra[f"sweep_{N}_{prec_name}_recall"] = {"t_from_v": r_ab, "v_from_t": r_ba}
                    one_matrix = world == 1 and N >= (vdist.BIDIR_MIN_ROWS_F32 if prec == L.SWEEP_F32 else vdist.BIDIR_MIN_ROWS)
                    extra[f"sweep_{N}_{prec_name}_path"] = "one distance matrix, row + column top-k" if one_matrix else "two searches"
                    # algorithmic HBM bytes, materialised fp32 matrix: 8 N^2 per direction, whole job
                    extra[f"sweep_{N}_{prec_name}_algorithmic_GBps"] = round(2 * 8.0 * N * N / dts / 1e9, 1)
                del va, noise, tb, va_l, tb_l
                torch.cuda.empty_cache()
            if args.hipgraph:
                # opt-in and last: in this ROCm build everything that ran after a capture was 1.5-6x slower
                try:
                    m2g = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(device)
                    m2g.compute_dtype = cdt
                    visg = torch.randn(B, 3, 224, 224, generator=gen).to(device).to(cdt)
                    m2g(visg, title, comments)
                    gstream = torch.cuda.Stream()
                    gstream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(gstream):
                        m2g(visg, title, comments)
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=gstream):
                            gout = m2g(visg, title, comments)
                    torch.cuda.current_stream().wait_stream(gstream)
                    graph.replay()
                    barrier_sync(world)
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        graph.replay()
                    barrier_sync(world)
                    extra["config2_ms_per_step_hipgraph"] = round(1e3 * max_over_ranks(time.perf_counter() - t0, world, device) / args.steps, 3)
                except Exception as e:
                    extra["config2_hipgraph_error"] = repr(e)[:200]
        except Exception as e:   # noqa: BLE001
            extra["error"] = repr(e)[:300]
        result["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
