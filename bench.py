"""bench.py -- headline benchmark of the VTC retrieval forward/eval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one forward pass of BASELINE.json configs[2] -- the largest single-GPU configuration -- over one batch
resident in HBM: PretrainedCLIP_TimeSformer_finaltf (8-frame 224x224 TimeSformer video tower + CLIP text tower over
1 title + 5 comments + Context Adapter Module + batch similarity), `--batch` pairs per GPU (default 1024), bf16-class
operands / fp32 accumulate, synthetic random pixels and tokens, random-init weights of the real architecture.
`value` = pairs encoded per second over the whole job (all ranks; weak scaling: pairs per GPU fixed).

Top-level objects beside the contract's fields:
  roofline      the dominant kernel class of the step (16-bit-operand MFMA GEMMs): achieved = sum(2MNK) / sum(kernel
                time), measured live with HIP events on the launch stream (vtc_prof_*), peak 2.5 PFLOP/s dense bf16;
                traffic = HBM bytes per launch from this round's PMC passes of the same command (profiles/), with their
                provenance, or null.
  timesformer_attention_mfma_frac
                BASELINE.md section 2: the video tower's time + space attention branches (LayerNorm + QKV + attention core +
                out-proj + temporal_fc + cls bookkeeping launches) at the reference's 4.27 GFLOP per layer per video,
                over their summed kernel time, as a fraction of 2.5 PFLOP/s.
  headline_batch_independence[_max_err]
                items [0:16] and [B-16:B] of the timed batch re-encoded as two B = 16 forwards (the shape the tests and the
                cpu_baseline leg hold to the oracle): max abs error of the embeddings and of the cosine block; > 1e-3 fails the run.
  sweep_10000_ms / sweep_50000_ms (+ _roofline, _phases_ms_per_rank)
                the second half of BASELINE's metric: N x N similarity + R@1/5/10 in both directions, parity mode
                (EXACT), sharded over the ranks; sweep_N_roofline: the distance GEMM (and the whole sweep) against the
                bf16 MFMA peak + PMC bytes per sweep (profiles/); the fused sweep never writes the matrix, so no HBM
                "fraction" on the materialised-matrix convention is printed any more.
  cpu_baseline  the oracle (plain PyTorch fp32 restatement of the reference, kind "port") on this box's host cores,
                BASELINE.md section 4 protocol on a bounded sample (rank 0, N = 1 only).
  extra         secondary measurements of the same path (config 2, dense-text variants, 16-frame stress encoder,
                adapter training step, other sweep precisions).
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 / f16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
TSF_ATTN_GFLOP_PER_LAYER_VIDEO = 4.27   # BASELINE.md section 3 / SURVEY 8d: QKV + core + out-proj + temporal_fc, F = 8


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


def prof_regions(fn, stream_ptr):
    """Run fn() with every launch bracketed by HIP events; {class: {region: {ms, launches, work}}}."""
    from vtc_amd import _lib as L
    lib = L.lib()
    nc, nr = len(L.PROF_CLASSES), len(L.PROF_REGIONS)
    lib.vtc_prof_begin()
    fn()
    ms, cnt, work = (C.c_double * (nc * nr))(), (C.c_longlong * (nc * nr))(), (C.c_double * (nc * nr))()
    L.check(lib.vtc_prof_end_regions(stream_ptr, ms, cnt, work), "vtc_prof_end_regions")
    return {cn: {rn: dict(ms=ms[i * nr + j], launches=int(cnt[i * nr + j]), work=work[i * nr + j])
                 for j, rn in enumerate(L.PROF_REGIONS)} for i, cn in enumerate(L.PROF_CLASSES)}


def prof_records(fn, stream_ptr, max_rec=20000):
    """Run fn() with every launch bracketed by HIP events; one record per launch: (class, region, ms, work, tag) -- tag of a
    GEMM launch = (epilogue mode, N, K)."""
    from vtc_amd import _lib as L
    lib = L.lib()
    lib.vtc_prof_begin()
    fn()
    n = C.c_int(0)
    cls, reg, tag = (C.c_int * max_rec)(), (C.c_int * max_rec)(), (C.c_int * (3 * max_rec))()
    ms, work = (C.c_double * max_rec)(), (C.c_double * max_rec)()
    L.check(lib.vtc_prof_end_records(stream_ptr, max_rec, C.byref(n), cls, reg, ms, work, tag), "vtc_prof_end_records")
    return [dict(cls=L.PROF_CLASSES[cls[i]], region=L.PROF_REGIONS[reg[i]], ms=ms[i], work=work[i], tag=(tag[3 * i], tag[3 * i + 1], tag[3 * i + 2]))
            for i in range(n.value)]


def records_to_regions(recs):
    from vtc_amd import _lib as L
    out = {c: {r: dict(ms=0.0, launches=0, work=0.0) for r in L.PROF_REGIONS} for c in L.PROF_CLASSES}
    for x in recs:
        e = out[x["cls"]][x["region"]]
        e["ms"] += x["ms"]; e["launches"] += 1; e["work"] += x["work"]
    return out


GEMM_MODE_NAMES = {0: "store", 1: "QuickGELU", 2: "residual (fp32 stream)", 3: "patch embed", 4: "L2 distance", 5: "scaled similarity",
                   6: "L2 block minima", 7: "residual + LayerNorm tail", 8: "store, folded LayerNorm", 9: "QuickGELU, folded LayerNorm",
                   10: "residual on the (hi, lo) stream", 11: "residual on the (hi, lo) stream, re-centring", 12: "L2 block minima, two planes", 13: "L2 block minima, three planes"}


def dominant_gemm(recs, cls, n_steps, peak):
    """The single GEMM instantiation x shape with the largest summed time: its own roofline line."""
    groups = {}
    for x in recs:
        if x["cls"] != cls:
            continue
        g = groups.setdefault(x["tag"], dict(ms=0.0, launches=0, work=0.0))
        g["ms"] += x["ms"]; g["launches"] += 1; g["work"] += x["work"]
    if not groups:
        return None
    tag, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    tf = g["work"] / (g["ms"] * 1e-3) / 1e12
    return dict(kernel=f"gemm_phased_kernel / gemm_kernel, epilogue '{GEMM_MODE_NAMES.get(tag[0], tag[0])}', N={tag[1]}, K={tag[2]}",
                launches_per_step=g["launches"] // n_steps, avg_launch_us=round(1e3 * g["ms"] / g["launches"], 2),
                flop_per_launch=round(g["work"] / g["launches"] / 1e9, 3), achieved=round(tf, 1), frac=round(tf / peak, 4),
                share_of_class_time=round(g["ms"] / sum(v["ms"] for v in groups.values()), 4))


def class_totals(p):
    return {c: dict(ms=sum(r["ms"] for r in v.values()), launches=sum(r["launches"] for r in v.values()),
                    work=sum(r["work"] for r in v.values())) for c, v in p.items()}


_T0 = time.perf_counter()


def log(msg):
    """progress to stderr (the JSON line is the only thing on stdout)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def gpu_randn(shape, seed, device, dtype):
    """synthetic pixels drawn on the GPU (a 1024-video batch is 1.2e9 values: minutes on the host cores)"""
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(shape, generator=g, device=device, dtype=torch.float32).to(dtype)


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


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut to the cgroup CPU quota when there is one (a 1-GPU
    box exposes every host core but grants about 16 of them; oversubscribing those is far slower than 16 threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        return max(1, min(n, int(quota)))
    return min(n, 16)


def best_of(fn, warm=2, runs=5, budget_s=40.0):
    """`warm` warm-ups, best of `runs` (cut short -- never below 3 -- once `budget_s` is spent)."""
    best, n = 1e30, 0
    t_leg = time.perf_counter()
    for i in range(warm + runs):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        if i >= warm:
            best, n = min(best, dt), n + 1
            if n >= 3 and time.perf_counter() - t_leg > budget_s:
                break
    return best, n, out


def cgroup_throttle():
    """(nr_throttled, throttled_usec) of this process's cgroup: CPU-bandwidth throttling stalls the launching thread"""
    try:
        kv = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except (OSError, ValueError):
        return 0, 0


def cpu_baseline(device=None):
    """BASELINE.md section 4 on a bounded sample: the oracle (kind "port": plain PyTorch fp32 restatement of the reference,
    pinned by the reference-generated golden vectors) on the host cores this process may use.  With `device`, the same
    inputs and weights also go through the HIP path and the embeddings are compared with the oracle's (the oracle as the
    checker): `gpu_vs_oracle_max_err` for the default 16-bit mode and for the text tower with bf16 blocks."""
    from oracle import arch as A
    from oracle import eval_ref as E
    from oracle import model_ref as M
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    a = A.VIT_B32
    # config 3 at B = 16, per pair
    sd = A.synth_model(a, 1023, "timesformer_finaltf", nframes=8)
    B = 16
    vis = A.synth_pixels((B, 8, 3, 224, 224), 123)
    title = A.synth_tokens(B, a, 124)
    comments = A.synth_tokens(B * 5, a, 125, empty_frac=0.1).reshape(B, 5, -1)
    with torch.no_grad():
        best, runs, ref3 = best_of(lambda: M.pretrained_clip_timesformer_finaltf(vis, title, comments, sd, a, "text"))
    out = dict(value=round(B / best, 3), unit="pairs/s", cores=cores, cpu=cpu_model_name(), kind="port",
               sample=f"oracle fp32 forward of config 3 (8-frame TimeSformer + title + 5 comments + CAM) at B={B}, "
                      f"{torch.get_num_threads()} threads, 2 warm-ups, best of {runs}: {best:.2f} s")
    if device is not None:
        try:
            from vtc_amd import towers as TW
            from vtc_amd.host import model as HM
            m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
            m.load_state_dict(sd, strict=True)
            m = m.eval().to(device)
            errs = {}
            was = TW.TEXT_HALF_LAYERS
            try:
                for name, hl in (("default (f16 text blocks)", was), ("text_bf16_blocks", 0)):
                    TW.TEXT_HALF_LAYERS = hl
                    m._packed = {}
                    got = m(vis.to(device).bfloat16(), title.to(device), comments.to(device))
                    errs[name] = {"feats_vis": float((got[0].cpu() - ref3[0]).abs().max()), "feats_text": float((got[1].cpu() - ref3[1]).abs().max()),
                                  "cosine_sim": float((got[0].cpu() @ got[1].cpu().T - ref3[0] @ ref3[1].T).abs().max())}
            finally:
                TW.TEXT_HALF_LAYERS = was
            out["gpu_vs_oracle_max_err"] = errs
            out["gpu_vs_oracle_tolerance"] = 1e-3
            del m
        except Exception as e:   # noqa: BLE001
            out["gpu_vs_oracle_error"] = repr(e)[:200]
    # config 1 (image + title, PretrainedCLIP) at B = 256 in full
    sd1 = A.synth_model(a, 1023, "clip")
    B1 = 256
    img = A.synth_pixels((B1, 3, 224, 224), 126)
    t1 = A.synth_tokens(B1, a, 127)
    with torch.no_grad():
        best1, runs1, _ = best_of(lambda: M.pretrained_clip(img, t1, sd1, a))
    out["config1_B256"] = dict(value=round(B1 / best1, 2), unit="pairs/s", sample=f"oracle fp32 forward of config 1 (ViT-B/32 image + title) at B={B1}, "
                               f"2 warm-ups, best of {runs1}: {best1:.2f} s")
    # sweep: 10k x 10k in full (literal restatement of RecallAtK.compute, model/metric.py:137-161, fp32 numpy), both
    # directions; 50k x 50k by row tiles on a sample of the query rows (2 x 1024 of 2 x 50000), scaled
    rng = np.random.default_rng(123)

    def planted(n):
        va = rng.standard_normal((n, 512)).astype(np.float32)
        va /= np.linalg.norm(va, axis=1, keepdims=True)
        tb = va + 0.05 * rng.standard_normal((n, 512)).astype(np.float32)
        tb /= np.linalg.norm(tb, axis=1, keepdims=True)
        return va, tb
    va, tb = planted(10000)
    t0 = time.perf_counter()
    E.recall_at_k(va, tb, [1, 5, 10])
    E.recall_at_k(tb, va, [1, 5, 10])
    out["sweep_10000_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    va, tb = planted(50000)
    rows = 1024
    t0 = time.perf_counter()
    E.l2_topk(va, tb[:rows], 11, np.float32, row_block=1024)
    E.l2_topk(tb, va[:rows], 11, np.float32, row_block=1024)
    out["sweep_50000_ms"] = round(1e3 * (time.perf_counter() - t0) * 50000 / rows, 1)
    out["sweep_50000_sample"] = f"{rows} of 50000 query rows per direction (row tiles of 1024), scaled by 50000/{rows}"
    return out


def pmc_traffic(workload_tag):
    """HBM bytes per launch of the dominant kernel class from this round's rocprofv3 PMC passes of this same command
    (profiles/*_traffic.json, written by tools/summarize_profile.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE
    runs with the gfx950 FETCH_SIZE x 2 correction).  Returned with its provenance; None when no file matches the
    workload (a number from another workload or kernel build would be stale)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("workload") == workload_tag:
            return round(float(j["traffic_bytes_per_launch"]), 1), ({"stored": True, "note": "PMC passes cannot run inside the timed run: a stored figure "
                                                                     "from the rocprofv3 --pmc passes of this command at the named commit"}
                                                                    | {k: j.get(k) for k in ("commit", "date", "source")} | {"file": "profiles/" + os.path.basename(f)})
    return None, None


def sweep_pmc_traffic(n):
    """HBM bytes of ONE whole sweep at N = n (every launch of it: prologue, distance GEMM, selection, re-rank, hit counting) from
    this round's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/sweep_profile.py (profiles/*_sweep_traffic.json, FETCH_SIZE
    x 2 on gfx950), beside the bytes a materialised fp32 matrix would cost; None when no pass of this round exists for that N."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sweep_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        e = j.get("sweeps", {}).get(str(n))
        if e:
            return dict(e, stored=True, file="profiles/" + os.path.basename(f), commit=j.get("commit"), date=j.get("date"))
    return None


def launches():
    from vtc_amd import _lib as L
    return int(L.lib().vtc_debug_launch_count())


def timed(fn, reps, world, device, warm=2):
    for _ in range(warm):
        fn()
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    barrier_sync(world)
    return max_over_ranks(time.perf_counter() - t0, world, device) / reps


def secondary_points(m3, vid, title, comments, B, world, device, extra, k2):
    """The headline model at the reference's own operating points and in the as-stated precisions (all in `extra`):
      * dense text at the headline batch (all 77 positions of every sequence: exactly the reference's work);
      * the text tower with bf16 blocks (VTC_TEXT_HALF_LAYERS=0: BASELINE's config says bf16; the default runs IEEE half);
      * B = 50, the batch_size of configs/pretrained_clip_timesformer_comments_attention.jsonc:4;
      * the batch-1 loop of evaluation/retrieval_evaluation.py:136,174-233 (one video = one 8-frame chunk + caption + 5 comments
        per forward) and the same videos through the ragged-chunk path (vtc_amd/host/retrieval_evaluation.py);
    each with the library's kernel launches per forward (at small batch the launch count is the cost)."""
    from vtc_amd import _lib as L
    from vtc_amd import towers as TW
    from vtc_amd.host import retrieval_evaluation as RE
    was_ragged, was_half = TW.TEXT_RAGGED, TW.TEXT_HALF_LAYERS
    try:
        log("extras: dense text at the headline batch")
        TW.TEXT_RAGGED = False
        d = timed(lambda: m3(vid, title, comments), k2, world, device, warm=1)
        extra[f"config3_B{B}_dense_text_pairs_per_s"] = round(world * B / d, 1)
        TW.TEXT_RAGGED = was_ragged
        log("extras: bf16 text blocks (VTC_TEXT_HALF_LAYERS=0)")
        TW.TEXT_HALF_LAYERS = 0
        m3._packed = {}
        d = timed(lambda: m3(vid, title, comments), k2, world, device, warm=1)
        extra[f"config3_B{B}_text_bf16_blocks_pairs_per_s"] = round(world * B / d, 1)
    finally:
        TW.TEXT_RAGGED, TW.TEXT_HALF_LAYERS = was_ragged, was_half
        m3._packed = {}
    # every row of the last block computed, as the reference does (default: its out_proj + MLP on the rows that reach the output)
    log("extras: full last block (VTC_TOWER_FULL_LAST_LAYER)")
    was_flags = TW.DEFAULT_FLAGS
    try:
        TW.DEFAULT_FLAGS = was_flags | L.TOWER_FULL_LAST_LAYER
        m3._packed = {}
        d = timed(lambda: m3(vid, title, comments), k2, world, device, warm=1)
        extra[f"config3_B{B}_full_last_block_pairs_per_s"] = round(world * B / d, 1)
    finally:
        TW.DEFAULT_FLAGS = was_flags
        m3._packed = {}
    # IEEE-half operands in BOTH towers (round 6: compute_dtype = torch.float16; the bf16 mode already runs the text blocks on half): same
    # MFMA rate and bytes, 11 significant bits instead of 8 -- the accuracy-side alternative to the headline's bf16 video tower
    log("extras: IEEE-half operands in both towers (--dtype f16)")
    was_dt = m3.compute_dtype
    try:
        m3.compute_dtype = torch.float16
        vh = vid.to(torch.float16)
        o_h = m3(vh, title, comments)
        d = timed(lambda: m3(vh, title, comments), k2, world, device, warm=1)
        m3.compute_dtype = was_dt
        o_b = m3(vid, title, comments)
        extra[f"config3_B{B}_f16_pairs_per_s"] = round(world * B / d, 1)
        extra[f"config3_B{B}_f16"] = {"pairs_per_s": round(world * B / d, 1), "ms_per_step": round(1e3 * d, 3),
                                      "max_abs_diff_vs_bf16_mode": {"feats_vis": round(float((o_h[0] - o_b[0]).abs().max()), 6),
                                                                    "feats_text": round(float((o_h[1] - o_b[1]).abs().max()), 6)},
                                      "what": "PretrainedCLIP_TimeSformer_finaltf with compute_dtype = torch.float16: IEEE-half GEMM / attention operands and (hi, lo) "
                                              "residual stream in the video tower too (tests: 5 - 8x closer to the fp32 oracle than the bf16 mode)"}
        del vh, o_h, o_b
    finally:
        m3.compute_dtype = was_dt
        m3._packed = {}
    # ... and BOTH at once: dense text + every row of the last block = exactly the reference's arithmetic work, in 16-bit
    log("extras: all the reference's work (dense text + full last block)")
    try:
        TW.DEFAULT_FLAGS = was_flags | L.TOWER_FULL_LAST_LAYER
        TW.TEXT_RAGGED = False
        m3._packed = {}
        d = timed(lambda: m3(vid, title, comments), k2, world, device, warm=1)
        extra[f"config3_B{B}_all_work_pairs_per_s"] = round(world * B / d, 1)
    finally:
        TW.DEFAULT_FLAGS, TW.TEXT_RAGGED = was_flags, was_ragged
        m3._packed = {}
    for b in (50, 1):
        v, t, c = vid[:b].contiguous(), title[:b].contiguous(), comments[:b].contiguous()
        m3(v, t, c)
        n0 = launches()
        m3(v, t, c)
        nl = launches() - n0
        d = timed(lambda: m3(v, t, c), 50 if b == 1 else 20, world, device)
        extra[f"config3_B{b}"] = {"pairs_per_s": round(world * b / d, 1), "ms_per_forward": round(1e3 * d, 3), "launches_per_forward": nl,
                                  "what": ("configs/pretrained_clip_timesformer_comments_attention.jsonc:4 batch_size" if b == 50 else
                                           "the batch-1 loop of evaluation/retrieval_evaluation.py:136,174-233, one 8-frame chunk per video")}
    # the same eval loop as ragged batches: 64 videos of 1-4 chunks each, captions + dummy comments, R@K included
    g = torch.Generator().manual_seed(5)
    vids = []
    for i in range(64):
        nfr = [8, 13, 24, 30][i % 4]
        vids.append((vid[i % vid.shape[0], :1].expand(nfr, -1, -1, -1).contiguous() + 0.01 * i, title[i % B].cpu()))
    RE.retrieval_evaluation(m3, vids, "full-test", device, frame_stride=1)
    n0 = launches()
    t0 = time.perf_counter()
    RE.retrieval_evaluation(m3, vids, "full-test", device, frame_stride=1)
    torch.cuda.synchronize()
    d = time.perf_counter() - t0
    extra["chunked_eval_64_videos"] = {"videos_per_s": round(64 / d, 1), "ms_total": round(1e3 * d, 3), "launches": launches() - n0,
                                       "chunks": sum((n + 7) // 8 for n in [8, 13, 24, 30]) * 16,
                                       "what": "vtc_amd/host/retrieval_evaluation.py: ragged chunk batches + segment mean + R@K, instead of 64 batch-1 forwards"}


def e2e_eval(m3, B, title, comments, world, device, extra):
    """End to end through the loop of evaluation/eval.py:101-126 -- host batches in, embeddings stacked, R@1/5/10 both
    directions out -- with the inputs where a loader leaves them: raw uint8 frames (before CLIP_TRANSFORM's ToTensor + Normalize,
    dataset_loaders/dataset_loaders.py:40-49, which the vision tower applies itself) and int64 tokens in PINNED host memory.
    Batch i + 1 is copied on a second stream while batch i is encoded (two device staging buffers); never part of `value`."""
    from vtc_amd.host.metric import RecallAtK
    Be, nb = min(B, 256), 8
    log(f"extras: end-to-end eval loop, {nb} host batches of {Be} uint8 videos")
    g = torch.Generator().manual_seed(77)
    hv = [torch.randint(0, 256, (Be, 8, 3, 224, 224), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
    ht, hc = title[:Be].cpu().pin_memory(), comments[:Be].cpu().pin_memory()
    dv = [torch.empty_like(hv[0], device=device) for _ in range(2)]
    dt_, dc = [torch.empty_like(ht, device=device) for _ in range(2)], [torch.empty_like(hc, device=device) for _ in range(2)]
    copy = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        k = i % 2
        with torch.cuda.stream(copy):
            copy.wait_event(done[k])                 # the forward that last read this staging buffer
            dv[k].copy_(hv[k], non_blocking=True)
            dt_[k].copy_(ht, non_blocking=True)
            dc[k].copy_(hc, non_blocking=True)
            ready[k].record(copy)

    def run(overlap):
        fv, ft = [], []
        for k in range(2):
            done[k].record(main)
        upload(0)
        for i in range(nb):
            k = i % 2
            if overlap and i + 1 < nb:
                upload(i + 1)
            main.wait_event(ready[k])
            out = m3(dv[k], dt_[k], dc[k])
            done[k].record(main)
            fv.append(out[0]); ft.append(out[1])
            if not overlap and i + 1 < nb:
                torch.cuda.synchronize()
                upload(i + 1)
                torch.cuda.synchronize()
        r = RecallAtK("videos", "titles", [1, 5, 10]).compute_both(torch.cat(fv), torch.cat(ft))
        torch.cuda.synchronize()
        return r

    run(True)
    res = {}
    for name, overlap in (("overlapped", True), ("serialised", False)):
        barrier_sync(world)
        t0 = time.perf_counter()
        run(overlap)
        d = max_over_ranks(time.perf_counter() - t0, world, device)
        res[name + "_pairs_per_s"] = round(world * nb * Be / d, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(copy)
    with torch.cuda.stream(copy):
        dv[0].copy_(hv[0], non_blocking=True)
    e1.record(copy)
    torch.cuda.synchronize()
    res["h2d_GBps_uint8_frames"] = round(hv[0].numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    res["what"] = (f"{nb} batches of {Be} pairs: pinned host uint8 frames [B, 8, 3, 224, 224] + int64 tokens -> H2D on a second stream -> "
                   "forward (ToTensor + Normalize inside the vision tower) -> embeddings kept on the GPU -> R@1/5/10 both directions")
    extra["eval_e2e_pairs_per_s"] = res["overlapped_pairs_per_s"]
    extra["eval_e2e"] = res


def eval_pipeline(m3, vid, N, B, rank, world, device, extra):
    """BASELINE configs[3] end to end: a gallery of N synthetic pairs, rank r encodes its contiguous shard [lo_r, hi_r) batch by batch (no
    collective), then ONE sharded sweep over the stacked embeddings (vtc_amd/dist.py sharded_recall: all-gather, one [N/G, N] distance GEMM
    per rank, all-to-all of column planes, all-reduce of the counters) -- the code path of `torch.distributed.run evaluation/eval.py`
    (vtc_amd/host/eval.py) with the inputs already in HBM.  Reported per phase, max over ranks; never part of `value`."""
    from vtc_amd import dist as vdist
    lo, hi = vdist.shard_bounds(N, rank, world)
    log(f"extras: eval pipeline N={N}: rank {rank} encodes pairs [{lo}, {hi})")
    gen = torch.Generator().manual_seed(991)                      # the same N titles / comments on every rank; a rank uses its slice
    titles = synth_tokens(N, 77, gen)[lo:hi].to(device)
    comms = synth_tokens(N * 5, 77, gen, empty_frac=0.1).reshape(N, 5, 77)[lo:hi].to(device)
    gd = torch.Generator(device=device).manual_seed(4242 + rank)

    def encode():
        fv, ft = [], []
        for b0 in range(0, hi - lo, B):
            n = min(B, hi - lo - b0)
            vid[:n].normal_(generator=gd)                          # fresh pixels per batch (identical videos would be exact duplicates in the sweep)
            o = m3(vid[:n], titles[b0:b0 + n], comms[b0:b0 + n])
            fv.append(o[0]); ft.append(o[1])
        return torch.cat(fv), torch.cat(ft)

    encode()                                                       # warm-up (workspaces of the last, shorter batch)
    barrier_sync(world)
    t0 = time.perf_counter()
    fv, ft = encode()
    barrier_sync(world)
    t_enc = max_over_ranks(time.perf_counter() - t0, world, device)
    vdist.sharded_recall(fv, ft, N, [1, 5, 10], rank, world)      # warm-up
    ph = {}
    barrier_sync(world)
    t1 = time.perf_counter()
    r_ab, r_ba = vdist.sharded_recall(fv, ft, N, [1, 5, 10], rank, world, phases=ph)
    barrier_sync(world)
    t_sw = max_over_ranks(time.perf_counter() - t1, world, device)
    phases = [ph]
    if world > 1:
        import torch.distributed as dist
        phases = [None] * world
        dist.all_gather_object(phases, ph)
    extra[f"eval_pipeline_{N}"] = {
        "pairs": N, "pairs_per_rank": hi - lo, "batch": B, "encode_ms": round(1e3 * t_enc, 2), "sweep_ms": round(1e3 * t_sw, 3),
        "total_ms": round(1e3 * (t_enc + t_sw), 2), "pairs_per_s_end_to_end": round(N / (t_enc + t_sw), 1),
        "sweep_phases_ms_per_rank": phases, "recall_t_from_v": r_ab, "recall_v_from_t": r_ba,
        "what": "encode (collective-free, per rank) + exchange + sharded sweep: the multi-GPU eval entry's pipeline (evaluation/eval.py under "
                "torch.distributed.run), inputs resident in HBM; random-init towers on random pixels: the recall values are chance level"}


def headline_independence(m3, vid, title, comments, B):
    """the headline batch held to what the tests hold to the oracle.  Items [0:16] and [B-16:B] of the B-pair
    forward against the same items encoded as two B = 16 batches (the shape `cpu_baseline.gpu_vs_oracle_max_err` and
    tests/test_gpu_towers.py compare with the oracle): max abs error of the unit-norm embeddings and of the cosine similarity
    block (sim / exp(logit_scale)), for the default forward and with every row of the last block computed.  No op of the path
    mixes items (model/model.py:596-623), so these must agree within the 16-bit tolerance 1e-3; the run FAILS otherwise."""
    from vtc_amd import _lib as L
    from vtc_amd import towers as TW
    scale = float(m3.model.logit_scale.detach().exp())
    n = min(16, B)
    res = {}
    was = TW.DEFAULT_FLAGS
    try:
        for name, flags in (("default", was), ("full_last_block", was | L.TOWER_FULL_LAST_LAYER)):
            TW.DEFAULT_FLAGS = flags
            m3._packed = {}
            big = m3(vid, title, comments)
            e = {"feats_vis": 0.0, "feats_text": 0.0, "cosine_sim_block": 0.0}
            for lo in sorted({0, B - n}):
                sl = slice(lo, lo + n)
                small = m3(vid[sl].contiguous(), title[sl].contiguous(), comments[sl].contiguous())
                e["feats_vis"] = max(e["feats_vis"], float((big[0][sl] - small[0]).abs().max()))
                e["feats_text"] = max(e["feats_text"], float((big[1][sl] - small[1]).abs().max()))
                e["cosine_sim_block"] = max(e["cosine_sim_block"], float((big[2][sl, sl] - small[2]).abs().max()) / scale)
            # every row of the batch similarity against the batch's own embeddings in fp64
            e["sim_vs_own_embeddings_fp64"] = float((big[2].double() / scale - big[0].double() @ big[1].double().T).abs().max())
            e["finite"] = bool(all(torch.isfinite(t).all() for t in big))
            res[name] = {k: (round(v, 9) if isinstance(v, float) else v) for k, v in e.items()}
    finally:
        TW.DEFAULT_FLAGS = was
        m3._packed = {}
    worst = max(v for d in res.values() for k, v in d.items() if k in ("feats_vis", "feats_text", "cosine_sim_block"))
    res.update(items=f"[0:{n}] and [{B - n}:{B}] of the B={B} forward vs the same items at B={n}", tolerance=1e-3, max_err=round(worst, 9),
               ok=bool(worst <= 1e-3 and all(d["finite"] and d["sim_vs_own_embeddings_fp64"] <= 1e-5 for d in res.values() if isinstance(d, dict))))
    return res


def spawn_ranks(n):
    """One process per GPU through torch.distributed.run, as the driver itself launches an N > 1 run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n}: starting {n} ranks: {' '.join(cmd[1:9])} ...")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        log("the ranks exited 0 without a JSON line")
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="operand arithmetic: bf16 (BASELINE's; text blocks on IEEE half), f16 (IEEE half in both towers), f32 (the reference's)")
    ap.add_argument("--no-extra", action="store_true", help="skip config 2, the stress encoder and the extra sweep precisions")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-sweep", action="store_true", help="skip the N x N sweeps")
    ap.add_argument("--no-independence", action="store_true",
                    help="skip the headline batch-independence check (profiling passes only: its B = 16 forwards would mix into the per-kernel averages)")
    ap.add_argument("--sweep-n", type=int, default=10000)
    ap.add_argument("--stress-n", type=int, default=50000, help="second sweep size (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a CHILD torchrun (never an exec; nothing in this process
        # has touched the GPU yet), relay rank 0's JSON line and the exit code
        raise SystemExit(spawn_ranks(args.gpus))
    from vtc_amd import dist as vdist
    rank, local, world = vdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report n_gpus={world} for a --gpus {args.gpus} run")
    if os.environ.get("VTC_BENCH_RENDEZVOUS_ONLY") == "1":
        # tests (CPU, gloo): prove the self-launch + rendezvous + collective, stop before anything needs the card
        import torch.distributed as dist
        one = torch.ones(1)
        if world > 1:
            dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({"rendezvous": True, "n_gpus": world, "backend": dist.get_backend() if world > 1 else None,
                              "allreduce_check": float(one.item())}))
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.set_grad_enabled(False)
    # Host threads: torch sizes its intra-op pool by the cores it SEES (128 on a 1-GPU box) while the box's cgroup grants ~16;
    # a CPU op then burns the CFS quota with spinning pool threads and the kernel throttles the whole process -- the launching
    # thread included -- for the rest of the 100 ms period: whatever HIP call is in flight "takes" 5-90 ms.  That was the
    # 12 ms `sweep_10000_ms` of BENCH_r02 (profiles/r03_sweep_stall_rootcause.md).  Bound the pool to the quota.
    torch.set_num_threads(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    import warnings
    warnings.filterwarnings("ignore", message=".*no pretrained weights.*")     # random-init weights are the stated workload

    from vtc_amd import _lib as L
    from vtc_amd import ops
    from vtc_amd import towers as TW
    from vtc_amd.host import model as HM

    rccl = None
    if world > 1:
        import torch.distributed as dist
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        rccl = dict(backend=dist.get_backend(), ranks=dist.get_world_size(), allreduce_check=float(one.item()))
        assert int(one.item()) == world

    cdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    B = args.batch
    gen = torch.Generator().manual_seed(123 + rank)           # data seed (tests/test_pretrained_clip.py:46)
    torch.manual_seed(1023)                                    # weight seed (train.py:34)
    stream_ptr = torch.cuda.current_stream().cuda_stream
    gk = "gemm_bf16" if args.dtype != "f32" else "gemm_f32"
    peak = PEAK_BF16_TFLOPS if args.dtype != "f32" else 157.3

    # ---- config 3: 8-frame TimeSformer video + title + 5 comments (CAM) -- the headline ---------------------
    m3 = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt="text", branch_to_adapt_val="text",
                                               init_from_avg=True)
    # trained weights are not the init's zeros: temporal_fc (timesformer_clip_alt.py:246-250) and the CAM's projections
    for blk in m3.model.visual.transformer.resblocks:
        torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    for blk in m3.final_transformer.resblocks:
        torch.nn.init.normal_(blk.attn.out_proj.weight, std=0.02)
        torch.nn.init.normal_(blk.mlp.c_proj.weight, std=0.02)
    m3 = m3.eval().to(device)
    m3.compute_dtype = cdt
    log(f"config 3: model built, drawing {B} videos")
    vid = gpu_randn((B, 8, 3, 224, 224), 123 + rank, device, cdt)
    title = synth_tokens(B, 77, gen).to(device)
    comments = synth_tokens(B * 5, 77, gen, empty_frac=0.1).reshape(B, 5, 77).to(device)

    def step3():
        return m3(vid, title, comments)

    for _ in range(args.warmup):
        step3()
    barrier_sync(world)
    log("warm-up done, timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step3()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, device)
    log(f"timed region: {dt:.2f} s for {args.steps} steps")
    assert torch.isfinite(out[2]).all()
    value = world * B * args.steps / dt
    if args.no_independence:
        indep = {"max_err": None, "ok": True, "skipped": "--no-independence"}
    else:
        log("headline batch independence (items of the timed batch vs B=16 forwards)")
        indep = headline_independence(m3, vid, title, comments, B)
    if not indep["ok"]:
        raise SystemExit(f"bench.py: the headline batch does not reproduce its own items at B=16 within 1e-3: {json.dumps(indep)}")

    # instrumented steps: every launch bracketed by HIP events on the launch stream, towers back to back on ONE stream
    # (two kernels sharing the chip would each look slower)
    n_prof = min(args.steps, 3)
    m3.overlap_towers = False
    t1 = time.perf_counter()
    log("instrumented steps")
    recs = prof_records(lambda: [step3() for _ in range(n_prof)], stream_ptr)
    p_all = records_to_regions(recs)
    dt_instr = time.perf_counter() - t1
    tot = class_totals(p_all)
    g = tot[gk]
    achieved = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    # the video tower alone: its ATTN region = the time + space attention branches.  Measured with EVERY row of the last block
    # computed (VTC_TOWER_FULL_LAST_LAYER), so that the time covers exactly the work the algorithmic count below prices
    pk_v = m3._pack()["visual"]
    was_v = pk_v.w.flags
    pk_v.w.flags = was_v | L.TOWER_FULL_LAST_LAYER
    try:
        pk_v.forward(vid)
        pv = prof_regions(lambda: pk_v.forward(vid), stream_ptr)
    finally:
        pk_v.w.flags = was_v
    m3.overlap_towers = type(m3).overlap_towers
    attn_ms = sum(v["attn"]["ms"] for v in pv.values())
    attn_launches = sum(v["attn"]["launches"] for v in pv.values())
    attn_flop = TSF_ATTN_GFLOP_PER_LAYER_VIDEO * 1e9 * 12 * B
    attn_frac = attn_flop / (attn_ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12) if attn_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic({"bf16": "config3", "f16": "config3_f16", "f32": "config3_f32"}[args.dtype])
    n_tok = int((torch.cat([title, comments.reshape(-1, 77)]).argmax(-1) + 1).sum().item())
    roofline = dict(bound="mfma", achieved=round(achieved, 1), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                    traffic=traffic, traffic_source=traffic_src,
                    kernel=("16-bit-operand MFMA GEMMs of the step (gemm_phased_kernel / gemm_kernel), all epilogues" if args.dtype != "f32"
                            else "fp32 MFMA GEMMs of the step (gemm_kernel<float, ...>: v_mfma_f32_16x16x4_f32), all epilogues"),
                    launches_per_step=g["launches"] // n_prof, avg_launch_us=round(1e3 * g["ms"] / max(1, g["launches"]), 2),
                    flop_per_launch=round(g["work"] / max(1, g["launches"]) / 1e9, 3),
                    dominant_instantiation=dominant_gemm(recs, gk, n_prof, peak))
    result = {
        "metric": "video-text pairs encoded/sec (config 3: 8-frame TimeSformer video + title + 5 comments, CAM) + 10k x 10k sim+R@K ms",
        "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16 (video tower) + f16 (text tower blocks), fp32 accumulate" if args.dtype == "bf16" and TW.TEXT_HALF_LAYERS > 0 else
                  "f16 (IEEE-half operands in both towers), fp32 accumulate" if args.dtype == "f16" else args.dtype),
        "data": "synthetic",
        "config": {"workload": "configs/pretrained_clip_timesformer_comments_attention.jsonc PretrainedCLIP_TimeSformer_finaltf forward: "
                               f"{B} pairs/GPU/step = {B} videos (8 x 3 x 224 x 224) + {B} titles + {5 * B} comments (77 tokens) + CAM + sim",
                   "pairs_per_gpu": B, "frames": 8,
                   "text_tower": ("ragged: tokens after EOT are not computed, identical outputs" if TW.TEXT_RAGGED else "dense: all 77 positions"),
                   "text_tokens_computed_frac": round(n_tok / (6 * B * 77), 4) if TW.TEXT_RAGGED else 1.0,
                   "last_block": ("queries, out_proj + MLP of each tower's LAST block on the rows that reach the output only (x[:, 0] / the EOT row; keys and values for every row; the other rows "
                                  "of that block are read by nothing) -- identical embeddings; extra.config3_B*_full_last_block_pairs_per_s computes them anyway"
                                  if not (TW.DEFAULT_FLAGS & L.TOWER_FULL_LAST_LAYER) else "every row"),
                   "parallelism": f"dp{world} (one process per GPU, no collective in the encode path)"},
        "roofline": roofline,
        "timesformer_attention_mfma_frac": round(attn_frac, 4),
        "timesformer_attention": {"ms_per_step": round(attn_ms, 3), "launches": attn_launches,
                                  "algorithmic_gflop_per_layer_video": TSF_ATTN_GFLOP_PER_LAYER_VIDEO,
                                  "measured_on": "the video tower with every row of the last block computed (VTC_TOWER_FULL_LAST_LAYER)",
                                  "tflops": round(attn_flop / (attn_ms * 1e-3) / 1e12, 1) if attn_ms > 0 else 0.0},
        "kernel_ms_per_step": {k: round(v["ms"] / n_prof, 3) for k, v in tot.items() if v["launches"]},
        "region_ms_per_step": {r: round(sum(v[r]["ms"] for v in p_all.values()) / n_prof, 3) for r in L.PROF_REGIONS},
        "ms_per_step_with_events": round(1e3 * dt_instr / n_prof, 3),
        "headline_batch_independence_max_err": indep["max_err"],
        "headline_batch_independence": indep,
    }
    if rccl:
        result["rccl"] = rccl
    extra = {}
    if not args.no_extra:
        try:
            secondary_points(m3, vid, title, comments, B, world, device, extra, max(2, args.steps // 4))
        except Exception as e:   # noqa: BLE001
            extra["secondary_error"] = repr(e)[:300]
        try:
            e2e_eval(m3, B, title, comments, world, device, extra)
        except Exception as e:   # noqa: BLE001
            extra["eval_e2e_error"] = repr(e)[:300]
        try:
            eval_pipeline(m3, vid, args.sweep_n, B, rank, world, device, extra)
        except Exception as e:   # noqa: BLE001
            extra["eval_pipeline_error"] = repr(e)[:300]
    # the ragged headline and the dense-text figure (all 77 positions: exactly the reference's work) side by side
    if f"config3_B{B}_dense_text_pairs_per_s" in extra:
        result["value_dense_text"] = extra[f"config3_B{B}_dense_text_pairs_per_s"]
    if f"config3_B{B}_all_work_pairs_per_s" in extra:
        # no output-dead work skipped: all 77 text positions AND every row of each tower's last block -- the reference's FLOPs exactly
        result["value_all_work"] = extra[f"config3_B{B}_all_work_pairs_per_s"]
    adapter_sd = {k: v.detach().clone() for k, v in m3.state_dict().items()
                  if k.startswith("final_transformer.") or k in ("mask_embedding", "model.logit_scale")}
    del m3, vid, out
    torch.cuda.empty_cache()

    # ---- sweep: N x N sim + R@1/5/10 both directions, sharded by query rows (one all-gather + one all-reduce when
    # N > 1).  Embeddings drawn directly, planted positives so that R@K < 1.

    def rep_stats(ts_ms):
        """median / min / max / p90 of the per-repetition times and the count of outliers (> 2 x median)"""
        a = np.sort(np.asarray(ts_ms, dtype=np.float64))
        med = float(np.median(a))
        return dict(median=round(med, 4), min=round(float(a[0]), 4), max=round(float(a[-1]), 4),
                    p90=round(float(a[max(0, int(np.ceil(0.9 * len(a))) - 1)]), 4), reps=int(len(a)),
                    outliers=int((a > 2.0 * med).sum()))

    def run_sweep(N, prec, reps, profile=False):
        """EVERY repetition timed on its own: a HIP event pair on the launch stream (the GPU-side time of the sweep's launches)
        and a host clock around the whole call (what a caller of RecallAtK sees: launches + the D2H of the six hit counters).
        Per repetition the max over ranks; reported = the MEDIAN, with min / max / p90 / outliers beside it (one mean over
        three repetitions let a single 30 ms stall own the number in round 2)."""
        lo, hi = vdist.shard_bounds(N, rank, world)
        # synthetic embeddings with planted positives, drawn ON THE GPU (same on every rank: seeded device generator); no host
        # tensor work next to the timed region
        g2 = torch.Generator(device=device).manual_seed(123)
        va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=device), dim=-1)
        noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=device), dim=-1)
        tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2, device=device), dim=-1)
        va_l, tb_l = va[lo:hi].contiguous(), tb[lo:hi].contiguous()
        del va, tb, noise
        th0 = cgroup_throttle()
        # caller-owned workspace, as a serving loop would hold it (a fresh multi-GiB allocation per call can land on a hipMalloc)
        need = vdist.sweep_workspace_bytes(N, hi - lo, 512, prec, world)
        ws = ops.workspace(need, device)
        for _ in range(3):
            vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=ws)      # warm-up
        host_ms, gpu_ms = [], []
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i in range(reps):
            barrier_sync(world)
            t0 = time.perf_counter()
            evs[i][0].record()
            r_ab, r_ba = vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=ws)
            evs[i][1].record()
            torch.cuda.synchronize()
            host_ms.append(1e3 * (time.perf_counter() - t0))
        gpu_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
        mine = float(np.median(host_ms))
        if world > 1:
            import torch.distributed as dist
            x = torch.tensor([host_ms, gpu_ms], dtype=torch.float64, device=device)
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            host_ms, gpu_ms = x[0].tolist(), x[1].tolist()
        th1 = cgroup_throttle()
        out = dict(host=rep_stats(host_ms), gpu=rep_stats(gpu_ms), mine=mine, r_ab=r_ab, r_ba=r_ba)
        out["host"]["cgroup_throttled_periods"] = th1[0] - th0[0]       # host-side CFS throttling inside the timed repetitions (0 = clean)
        # one more repetition with HIP events between the phases (all-gather / distance GEMM + selection / exchange / column
        # selection / all-reduce), per rank: what a scaling line is read from
        ph = {}
        vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=ws, phases=ph)
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            allph = [None] * world
            dist.all_gather_object(allph, ph)
            out["phases"] = allph
        else:
            out["phases"] = [ph]
        if profile:
            # kernel classes of one more repetition (HIP events around every launch): the distance GEMM against the rest
            pt = class_totals(prof_regions(lambda: vdist.sharded_recall(va_l, tb_l, N, [1, 5, 10], rank, world, precision=prec, ws=ws),
                                           stream_ptr))
            out["classes"] = pt
        return out

    if not args.no_sweep:
        try:
            for N in [n for n in (args.sweep_n, args.stress_n) if n > 0]:
                log(f"sweep N={N}")
                reps = 40 if N <= 20000 else 12
                try:
                    sw = run_sweep(N, L.SWEEP_EXACT, reps, profile=True)
                except Exception as e:   # noqa: BLE001  -- N > 1 only: the exchange failed on this fabric; say so and time the two-search path
                    if world == 1 or os.environ.get("VTC_SWEEP_SHARD_TWO") == "1":
                        raise
                    result["sweep_one_matrix_error"] = repr(e)[:300]
                    os.environ["VTC_SWEEP_SHARD_TWO"] = "1"
                    sw = run_sweep(N, L.SWEEP_EXACT, reps, profile=True)
                dts = sw["host"]["median"] * 1e-3
                result[f"sweep_{N}_ms"] = round(sw["host"]["median"], 3)
                result[f"sweep_{N}_ms_stats"] = sw["host"]
                result[f"sweep_{N}_gpu_ms_stats"] = sw["gpu"]
                result[f"sweep_{N}_phases_ms_per_rank"] = sw["phases"]
                # (rounds 1-4 printed sweep_N_hbm_frac = 2 x 8 N^2 bytes / time / 8 TB/s here -- the bytes of a MATERIALISED fp32 matrix,
                # which this sweep never writes: a "fraction" above 1 is not a measurement, so the figure is gone.
                # What bounds the sweep is its distance GEMM (bf16 MFMA, 2 N^2 512 FLOP) + that GEMM's VALU epilogue: sweep_N_roofline.)
                cl = sw.get("classes", {})
                gk_ = cl.get("gemm_bf16", {"ms": 0.0, "work": 0.0, "launches": 0})
                other_ms = sum(v["ms"] for k, v in cl.items() if k != "gemm_bf16")
                if gk_["ms"] > 0:
                    result[f"sweep_{N}_roofline"] = {
                        "bound": "mfma+valu",
                        "kernel": ("distance GEMM with the block-minima epilogue (gemm_phased_kernel<" +
                                   {2: "EPI_L2MIN2: two key planes, half-distance keys", 3: "EPI_L2MIN3: rows two / columns three key planes",
                                    4: "EPI_L2MIN: four key planes"}[ops.recall_planes([1, 5, 10], N)] + ">)"),
                        "gemm_ms": round(gk_["ms"], 4), "gemm_launches": gk_["launches"], "gemm_tflops": round(gk_["work"] / gk_["ms"] / 1e9, 1),
                        "frac": round(gk_["work"] / (gk_["ms"] * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        # the whole job's distance FLOPs (one [N/G, N] GEMM per rank) over the WHOLE sweep's wall time and all ranks' peak
                        "frac_whole_sweep": round(2.0 * N * N * 512 / dts / (PEAK_BF16_TFLOPS * 1e12 * world), 4),
                        "select_rerank_prologue_ms": round(other_ms, 4),
                        "kernel_ms_sum": round(gk_["ms"] + other_ms, 4),
                        "traffic": sweep_pmc_traffic(N)}
                result[f"sweep_{N}"] = {"mode": "EXACT (fp64-certified ranks: the parity mode)", "recall_t_from_v": sw["r_ab"], "recall_v_from_t": sw["r_ba"],
                                        "materialised_matrix_equivalent_GBps": round(2 * 8.0 * N * N / dts / 1e9, 1),      # NOT bytes moved: what writing + reading 2 fp32 matrices in this time would take
                                        "this_rank_ms": round(sw["mine"], 3),
                                        "path": vdist.sweep_path(N, L.SWEEP_EXACT, world)}
                if not args.no_extra:
                    for name, prec in (("f32", L.SWEEP_F32), ("bf16x3", L.SWEEP_BF16X3), ("bf16", L.SWEEP_BF16)):
                        if prec == L.SWEEP_BF16 and N != args.stress_n:
                            continue
                        s2 = run_sweep(N, prec, 20 if N <= 20000 else 6)
                        extra[f"sweep_{N}_{name}_ms"] = round(s2["host"]["median"], 3)
                        extra[f"sweep_{N}_{name}_ms_stats"] = s2["host"]
                torch.cuda.empty_cache()
        except Exception as e:   # noqa: BLE001
            result["sweep_error"] = repr(e)[:300]

    if not args.no_extra:
        try:   # the extras never cost the main line: an exception is recorded in extra["error"]
            k2 = max(2, args.steps // 4)
            # ---- config 3 with the DENSE text tower (all 77 positions of every sequence, exactly the reference's work)
            was_ragged = TW.TEXT_RAGGED
            m3 = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(device)
            m3.compute_dtype = cdt
            B3 = min(B, 256)
            log("extras: config 3 at B=256, ragged vs dense text")
            v3 = gpu_randn((B3, 8, 3, 224, 224), 223 + rank, device, torch.bfloat16)
            t3, c3 = title[:B3].contiguous(), comments[:B3].contiguous()
            for ragged in (True, False):
                TW.TEXT_RAGGED = ragged
                for _ in range(2):
                    m3(v3, t3, c3)
                barrier_sync(world)
                t0 = time.perf_counter()
                for _ in range(k2):
                    m3(v3, t3, c3)
                barrier_sync(world)
                d3 = max_over_ranks(time.perf_counter() - t0, world, device)
                extra[f"config3_B{B3}_{'ragged' if ragged else 'dense'}_text_pairs_per_s"] = round(world * B3 * k2 / d3, 1)
            del m3, v3
            torch.cuda.empty_cache()
            # ---- config 2: image + title + 5 comments (CAM), B = 256 -------------------------------------------
            m2 = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt="text", branch_to_adapt_val="text").eval().to(device)
            m2.compute_dtype = cdt
            B2 = min(B, 256)
            log("extras: config 2")
            img = gpu_randn((B2, 3, 224, 224), 323 + rank, device, torch.bfloat16)
            t2, c2 = title[:B2].contiguous(), comments[:B2].contiguous()
            for ragged in (True, False):
                TW.TEXT_RAGGED = ragged
                for _ in range(2):
                    m2(img, t2, c2)
                barrier_sync(world)
                t0 = time.perf_counter()
                for _ in range(2 * k2):
                    m2(img, t2, c2)
                barrier_sync(world)
                d2 = max_over_ranks(time.perf_counter() - t0, world, device)
                extra[f"config2_B{B2}_{'ragged' if ragged else 'dense'}_text_pairs_per_s"] = round(world * B2 * 2 * k2 / d2, 1)
                extra[f"config2_B{B2}_{'ragged' if ragged else 'dense'}_text_ms_per_step"] = round(1e3 * d2 / (2 * k2), 3)
            TW.TEXT_RAGGED = was_ragged
            m2.overlap_towers = False
            p2 = class_totals(prof_regions(lambda: m2(img, t2, c2), stream_ptr))
            extra["config2_gemm_tflops"] = round(p2[gk]["work"] / (p2[gk]["ms"] * 1e-3) / 1e12, 1)
            extra["config2_kernel_ms"] = {k: round(v["ms"], 3) for k, v in p2.items() if v["launches"]}
            del m2, img
            torch.cuda.empty_cache()
            # ---- config 3 in the REFERENCE's own arithmetic: fp32 end to end (model/model.py:318 `self.model.float()`), fp32 MFMA
            # (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense peak), every operand and the residual stream fp32, parity 1e-5 ---------
            m3f = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(device)
            for blk in m3f.model.visual.transformer.resblocks:
                torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
            m3f.compute_dtype = torch.float32
            Bf = min(B, 256)
            log("extras: config 3 in fp32 (the reference's arithmetic)")
            vf = gpu_randn((Bf, 8, 3, 224, 224), 523 + rank, device, torch.float32)
            tf_, cf_ = title[:Bf].contiguous(), comments[:Bf].contiguous()
            for _ in range(2):
                m3f(vf, tf_, cf_)
            barrier_sync(world)
            t0 = time.perf_counter()
            kf = max(2, k2 // 2)
            for _ in range(kf):
                m3f(vf, tf_, cf_)
            barrier_sync(world)
            df = max_over_ranks(time.perf_counter() - t0, world, device)
            m3f.overlap_towers = False
            pf = class_totals(prof_regions(lambda: m3f(vf, tf_, cf_), stream_ptr))
            gf = pf["gemm_f32"]
            extra["config3_f32_pairs_per_s"] = round(world * Bf * kf / df, 1)
            extra["config3_f32"] = {"batch_per_gpu": Bf, "ms_per_step": round(1e3 * df / kf, 3),
                                    "gemm_tflops": round(gf["work"] / (gf["ms"] * 1e-3) / 1e12, 1) if gf["ms"] > 0 else 0.0,
                                    "gemm_frac_of_fp32_mfma_peak": round(gf["work"] / (gf["ms"] * 1e-3) / 157.3e12, 4) if gf["ms"] > 0 else 0.0,
                                    "peak_tflops": 157.3, "kernel_ms": {k: round(v["ms"], 3) for k, v in pf.items() if v["launches"]},
                                    "what": "the drop-in's --dtype f32 / VTC_COMPUTE_DTYPE=f32 mode: fp32 operands + fp32 MFMA everywhere, "
                                            "embeddings within 1e-5 of the reference's fp32 CPU path (tests: fp32 2.5e-7)"}
            del m3f, vf
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
            log("extras: 16-frame stress encoder")
            vid16 = gpu_randn((B16, 16, 3, 224, 224), 423 + rank, device, torch.bfloat16)
            t16, c16 = title[:B16].contiguous(), comments[:B16].contiguous()
            for _ in range(2):
                m16(vid16, t16, c16)
            barrier_sync(world)
            t0 = time.perf_counter()
            for _ in range(k2):
                m16(vid16, t16, c16)
            barrier_sync(world)
            dt16 = max_over_ranks(time.perf_counter() - t0, world, device)
            extra["stress_timesformer16_pairs_per_s"] = round(world * B16 * k2 / dt16, 1)
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
            extra["adapter_train_step_ms"] = round(1e3 * max_over_ranks(time.perf_counter() - t0, world, device) / kt, 3)
            extra["adapter_train_batch"] = Bt
            extra["adapter_train_loss_after"] = round(float(tl), 4)
            del trn
        except Exception as e:   # noqa: BLE001
            extra["error"] = repr(e)[:300]
    result["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu:
        log("cpu baseline leg")
        result["cpu_baseline"] = cpu_baseline(device)
    log("done")
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
