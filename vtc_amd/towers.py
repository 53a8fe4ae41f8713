"""Weight packing and tower launches for the HIP path.

A state dict (the reference's key names, tensors on the GPU) is packed ONCE into the C
structs of include/vtc_hip.h: matrices cast to the compute dtype in their PyTorch
``[out, in]`` layout, projections transposed, and -- optionally -- the two back-to-back linear
maps of the temporal branch (``timeattn.out_proj`` then ``temporal_fc``,
model/timesformer_clip_alt.py:65,148) multiplied together on the host so that the branch costs
one GEMM instead of two.  The forward functions then only size a workspace and call the library.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops

SD = Dict[str, torch.Tensor]

# items per vision-tower launch; 0 = the whole batch in one pass
VISION_CHUNK = 0
# sequences per text-tower launch; 0 = the whole batch in one pass.  A chunk whose activations
# (x fp32 + h + qkv/hidden, ~8 KB per token at W = 512) stay inside the 256 MiB Infinity Cache turns the
# HBM-bound stages (LayerNorm, attention, residual epilogues) into cache-bound ones.
TEXT_CHUNK = int(__import__("os").environ.get("VTC_TEXT_CHUNK", "0"))
# Text tower on ragged batches: compute only tokens 0..EOT of every sequence.  Identical outputs: under the causal
# mask no token after EOT can reach the EOT feature, and every other op of the tower is per-row
# (tests/test_gpu_towers.py::test_ragged_text_tower_equals_dense).  On by default (VTC_TEXT_RAGGED=0: the dense path,
# which computes all 77 positions of every sequence as the reference does).
TEXT_RAGGED = __import__("os").environ.get("VTC_TEXT_RAGGED", "1") != "0"
# bf16 mode of the TEXT tower: the first TEXT_HALF_LAYERS blocks run with IEEE-half operands instead of bf16 (same MFMA
# rate, same bytes, 11 significant bits instead of 8; |values| up to 65504 -- the format upstream CLIP itself runs in
# on a GPU).  The operand-rounding floor of an all-bf16 text tower is rms 3.2e-4 / max 1.1-1.4e-3 on the unit-norm
# embedding (tests/bf16_floor_study.py), above BASELINE's 1e-3 budget; layer 0 alone is 42 % of that variance.
# 0 = plain bf16 everywhere.
TEXT_HALF_LAYERS = int(__import__("os").environ.get("VTC_TEXT_HALF_LAYERS", "12"))
# 16-bit modes: pack the gamma-scaled projection weights of the folded LayerNorm next to the plain ones (include/vtc_hip.h,
# vtc_block_w *_wf / *_s / *_c); whether a forward uses them is the packed model's own `flags` (VTC_TOWER_*).
LN_FOLD_PACK = __import__("os").environ.get("VTC_LN_FOLD_PACK", "1") != "0"


def tower_flags(ln_fold: bool = True, full_last_layer: bool = False, splitk: bool = True) -> int:
    """vtc_vision_w.flags / vtc_text_w.flags: ln_fold False = the LayerNorm kernels instead of the folded LayerNorms;
    full_last_layer = compute the last block's out_proj / MLP on every row instead of the output rows only (the rows nothing
    reads; same embeddings); splitk False = the MLP's c_proj as one GEMM also at batch 1 - 2 (ABI 7; default: split over K there).
    (The fused QKV + attention flags of rounds 1-4 are gone with the kernel: ABI 6.)"""
    return (0 if ln_fold else L.TOWER_NO_LN_FOLD) | (L.TOWER_FULL_LAST_LAYER if full_last_layer else 0) | (0 if splitk else L.TOWER_NO_SPLITK)


# defaults of newly packed towers (env VTC_LN_FOLD=0 / VTC_FULL_LAST_LAYER=1: A/B runs); a packed tower's `w.flags` may be set per model
DEFAULT_FLAGS = tower_flags(__import__("os").environ.get("VTC_LN_FOLD", "1") != "0",
                            __import__("os").environ.get("VTC_FULL_LAST_LAYER", "0") == "1",
                            __import__("os").environ.get("VTC_SPLITK", "1") != "0")
_WS: Dict[tuple, torch.Tensor] = {}


def _ws(nbytes: int, device) -> torch.Tensor:
    """One growing workspace per (device, stream): towers running concurrently on two streams must not share scratch."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    w = _WS.get(key)
    if w is None or w.numel() < nbytes:
        _WS[key] = w = None  # drop the old one before growing
        _WS[key] = w = ops.workspace(nbytes, device)
    return w


def _host_sd(sd: SD, prefix: str, skip: str = ""):
    """(the sub-dict under `prefix` with its tensors copied to the HOST, the device they lived on).  Packing -- casts,
    transposes, the fp64 products of the fused temporal map and of the folded LayerNorm -- is format conversion done once per
    set of weights: it runs on the CPU, and the GPU only receives the packed arrays (no torch kernel, no rocBLAS launch on
    the card on behalf of this package)."""
    dev = None
    out = {}
    for k, v in sd.items():
        if not k.startswith(prefix) or (skip and k.startswith(skip)):
            continue
        if v.is_cuda and dev is None:
            dev = v.device
        out[k[len(prefix):]] = v.detach().cpu()
    if dev is None:
        raise RuntimeError("vtc_amd: weights must live on the GPU the model runs on (call .to('cuda') first); no CPU path")
    return out, dev


class _Keep:
    """Owns the packed device arrays a struct points into; converts on the host, uploads once."""

    def __init__(self, device, arena_bytes: int = 0):
        """arena_bytes > 0: every packed array is carved out of ONE device allocation (256-byte aligned pieces).  The one-launch
        CAM reads each weight matrix once per call, in 32 KB slices, with nothing to amortise an address translation over:
        dozens of separate small allocations cost it TLB misses on every phase."""
        self.device = device
        self.t: List[torch.Tensor] = []
        self.arena = torch.empty(arena_bytes, dtype=torch.uint8, device=device) if arena_bytes > 0 else None
        self.used = 0

    def f32(self, t: torch.Tensor) -> int:
        return self.mat(t, torch.float32)

    def mat(self, t: torch.Tensor, dtype) -> int:
        t = t.detach().cpu().to(dtype).contiguous()
        nbytes = t.numel() * t.element_size()
        if self.arena is not None and self.used + nbytes <= self.arena.numel():
            dst = self.arena[self.used:self.used + nbytes].view(dtype).view(t.shape)
            dst.copy_(t)
            self.used = (self.used + nbytes + 255) // 256 * 256
            self.t.append(dst)
            return dst.data_ptr()
        t = t.to(self.device)
        self.t.append(t)
        return t.data_ptr()


def _n_layers(sd: SD, p: str) -> int:
    pre = f"{p}.resblocks."
    return 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))


def _fold_ln(keep: "_Keep", w: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype):
    """LayerNorm folded into the projection behind it (include/vtc_hip.h, vtc_block_w): W' = gamma . W rounded to the operand
    format, s = row sums of the ROUNDED W' (what the GEMM multiplies the mean by), c = bias + W beta -- sums in fp64."""
    wf = (w.double() * gamma.double()[None, :]).float().to(dtype)
    s = wf.double().sum(dim=1).float()
    c = (bias.double() + w.double() @ beta.double()).float()
    return keep.mat(wf, dtype), keep.f32(s), keep.f32(c)


def _pack_blocks(sd: SD, p: str, layers: int, dtype, keep: _Keep, timesformer: bool, fuse_temporal: bool,
                 half_layers: int = 0, fold_ln: bool = False, fold_fp32: bool = False):
    arr = (L.BlockW * layers)()
    base_dtype = dtype
    # (fp32: the one-launch CAM folds its LayerNorms too -- cam.hip -- so that a token row streams past the weights once)
    fold_ln = fold_ln and (base_dtype in (torch.bfloat16, torch.float16) or (fold_fp32 and base_dtype == torch.float32))
    for i in range(layers):
        q, b = f"{p}.resblocks.{i}", arr[i]
        dtype = torch.float16 if (base_dtype == torch.bfloat16 and i < half_layers) else base_dtype
        if fold_ln:
            b.qkv_wf, b.qkv_s, b.qkv_c = _fold_ln(keep, sd[f"{q}.attn.in_proj_weight"], sd[f"{q}.attn.in_proj_bias"],
                                                  sd[f"{q}.ln_1.weight"], sd[f"{q}.ln_1.bias"], dtype)
            b.fc_wf, b.fc_s, b.fc_c = _fold_ln(keep, sd[f"{q}.mlp.c_fc.weight"], sd[f"{q}.mlp.c_fc.bias"],
                                               sd[f"{q}.ln_2.weight"], sd[f"{q}.ln_2.bias"], dtype)
            if timesformer:
                b.tqkv_wf, b.tqkv_s, b.tqkv_c = _fold_ln(keep, sd[f"{q}.timeattn.in_proj_weight"], sd[f"{q}.timeattn.in_proj_bias"],
                                                         sd[f"{q}.ln_time.weight"], sd[f"{q}.ln_time.bias"], dtype)
        b.ln1_g, b.ln1_b = keep.f32(sd[f"{q}.ln_1.weight"]), keep.f32(sd[f"{q}.ln_1.bias"])
        b.qkv_w, b.qkv_b = keep.mat(sd[f"{q}.attn.in_proj_weight"], dtype), keep.f32(sd[f"{q}.attn.in_proj_bias"])
        b.out_w, b.out_b = keep.mat(sd[f"{q}.attn.out_proj.weight"], dtype), keep.f32(sd[f"{q}.attn.out_proj.bias"])
        b.ln2_g, b.ln2_b = keep.f32(sd[f"{q}.ln_2.weight"]), keep.f32(sd[f"{q}.ln_2.bias"])
        b.fc_w, b.fc_b = keep.mat(sd[f"{q}.mlp.c_fc.weight"], dtype), keep.f32(sd[f"{q}.mlp.c_fc.bias"])
        b.proj_w, b.proj_b = keep.mat(sd[f"{q}.mlp.c_proj.weight"], dtype), keep.f32(sd[f"{q}.mlp.c_proj.bias"])
        if timesformer:
            b.lnt_g, b.lnt_b = keep.f32(sd[f"{q}.ln_time.weight"]), keep.f32(sd[f"{q}.ln_time.bias"])
            b.tqkv_w, b.tqkv_b = keep.mat(sd[f"{q}.timeattn.in_proj_weight"], dtype), keep.f32(sd[f"{q}.timeattn.in_proj_bias"])
            wo, bo = sd[f"{q}.timeattn.out_proj.weight"], sd[f"{q}.timeattn.out_proj.bias"]
            if f"{q}.temporal_fc.weight" not in sd:      # model/timesformer_clip.py: no temporal_fc
                b.tout_w, b.tout_b = keep.mat(wo, dtype), keep.f32(bo)
                b.tfc_w, b.tfc_b = None, None
                continue
            wf, bf = sd[f"{q}.temporal_fc.weight"], sd[f"{q}.temporal_fc.bias"]
            if fuse_temporal:
                # temporal_fc(out_proj(a)) = (Wf Wo) a + (Wf bo + bf): one GEMM instead of two
                w64 = wf.double() @ wo.double()
                b64 = wf.double() @ bo.double() + bf.double()
                b.tout_w, b.tout_b = None, None
                b.tfc_w, b.tfc_b = keep.mat(w64.float(), dtype), keep.f32(b64.float())
            else:
                b.tout_w, b.tout_b = keep.mat(wo, dtype), keep.f32(bo)
                b.tfc_w, b.tfc_b = keep.mat(wf, dtype), keep.f32(bf)
    return arr


class PackedVision:
    def __init__(self, sd: SD, prefix: str, dtype, fuse_temporal: bool = True):
        self._src = (sd, prefix, fuse_temporal)          # references to the caller's tensors: the bf16 re-pack of the half mode's range guard
        self._flag_dev = self._flag_host = None
        self._calibrated = False
        self.range_fallbacks = 0
        self._build(dtype)

    def _build(self, dtype):
        sd, prefix, fuse_temporal = self._src
        sd, dev = _host_sd(sd, prefix)
        self.dtype, self.code, self.keep = dtype, ops.dtype_code(dtype), _Keep(dev)
        if dtype == torch.float16 and self._flag_dev is None:
            self._flag_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        k = self.keep
        conv = sd["conv1.weight"]
        w = L.VisionW()
        w.width, w.patch = conv.shape[0], conv.shape[-1]
        w.heads = w.width // 64
        w.layers = _n_layers(sd, "transformer")
        w.grid = int(round(math.sqrt(sd["positional_embedding"].shape[0] - 1)))
        w.embed_dim = sd["proj"].shape[1]
        w.nframes = sd["temporal_embed"].shape[0] if "temporal_embed" in sd else 0
        # variant 1 = model/timesformer_clip.py (frames-major tokens, cls attends globally, no temporal_fc)
        w.variant = 1 if (w.nframes and not any("temporal_fc" in k for k in sd)) else 0
        w.flags = DEFAULT_FLAGS
        # raw uint8 pixels: ToTensor + Normalize of CLIP_TRANSFORM (dataset_loaders/dataset_loaders.py:40-49)
        w.pix_mean = (C.c_float * 3)(0.48145466, 0.4578275, 0.40821073)
        w.pix_std = (C.c_float * 3)(0.26862954, 0.26130258, 0.27577711)
        cw = conv.reshape(w.width, -1)
        kp = (cw.shape[1] + 63) // 64 * 64          # K of the patch GEMM in whole 128-byte rows (ViT-L/14: 588 -> 640, zero columns)
        if kp != cw.shape[1]:
            cw = torch.cat([cw, cw.new_zeros(w.width, kp - cw.shape[1])], dim=1)
        w.conv_w = k.mat(cw, dtype)
        w.class_embedding, w.pos = k.f32(sd["class_embedding"]), k.f32(sd["positional_embedding"])
        w.temporal = k.f32(sd["temporal_embed"]) if w.nframes else None
        w.ln_pre_g, w.ln_pre_b = k.f32(sd["ln_pre.weight"]), k.f32(sd["ln_pre.bias"])
        w.ln_post_g, w.ln_post_b = k.f32(sd["ln_post.weight"]), k.f32(sd["ln_post.bias"])
        w.proj_t = k.mat(sd["proj"].t(), torch.float32)
        self.blocks = _pack_blocks(sd, "transformer", w.layers, dtype, k, bool(w.nframes), fuse_temporal, fold_ln=LN_FOLD_PACK)
        w.blocks = self.blocks
        self.w = w
        self.res = w.grid * w.patch

    # ---- range guard of the IEEE-half mode (round 6; as PackedText's) ------------------------------------------------------------
    # compute_dtype = torch.float16 runs the tower on half operands with a half (hi, lo) residual stream: 11 significant bits, range
    # +-65504.  A checkpoint whose activations leave that range shows as inf / NaN in the output: the FIRST forward after packing checks
    # synchronously, re-packs the tower as bf16 and recomputes; later forwards carry the flag to pinned memory asynchronously and the
    # next call switches (that call's NaN rows have been returned by then -- the wrappers' watchdog raises on them).
    def _range_switch(self, why: str):
        import warnings
        warnings.warn(f"vtc_amd vision tower: {why}: a value left the IEEE-half range (+-65504) in the half-operand mode; re-packing the "
                      "tower as bf16 operands (wider range, 8 significant bits instead of 11)", RuntimeWarning, stacklevel=3)
        self.range_fallbacks += 1
        self._flag_dev.zero_()
        self._flag_host.zero_()
        self._build(torch.bfloat16)

    @ops.on_device
    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels [N,3,H,W] (image tower) or [N,F,3,H,W] (TimeSformer), fp32 / bf16 / half / uint8 -> [N, embed] fp32."""
        if self.dtype == torch.float16 and self._flag_host is not None and int(self._flag_host[0]) != 0:
            self._range_switch("an earlier forward")         # pinned host memory: no synchronisation
        out = self._forward(pixels)
        if self.dtype != torch.float16:
            return out
        L.check(L.lib().vtc_nonfinite_flag(out.data_ptr(), out.numel(), self._flag_dev.data_ptr(), ops._stream()), "vtc_nonfinite_flag")
        if not self._calibrated:
            self._calibrated = True
            if int(self._flag_dev.item()) != 0:               # first call after packing: synchronous
                self._range_switch("first forward after packing")
                return self._forward(pixels)
            return out
        self._flag_host.copy_(self._flag_dev, non_blocking=True)
        return out

    def _forward(self, pixels: torch.Tensor) -> torch.Tensor:
        w = self.w
        if pixels.dim() == 4:
            pixels = pixels.unsqueeze(1)
        pixels = ops._gpu(pixels, name="pixels")
        if pixels.dtype not in (torch.float32, torch.bfloat16, torch.uint8, torch.float16):
            pixels = pixels.float()
        n, F = pixels.shape[0], pixels.shape[1]
        if tuple(pixels.shape[2:]) != (3, self.res, self.res):
            raise ValueError(f"expected [...,3,{self.res},{self.res}] pixels, got {tuple(pixels.shape)}")
        out = torch.empty(n, w.embed_dim, dtype=torch.float32, device=pixels.device)
        chunk = VISION_CHUNK if VISION_CHUNK > 0 else n
        lib = L.lib()
        ws = _ws(lib.vtc_vision_workspace_bytes(C.byref(w), min(chunk, n), F, self.code), pixels.device)
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            pcode = L.VTC_U8 if pixels.dtype == torch.uint8 else ops.dtype_code(pixels.dtype)
            L.check(lib.vtc_vision_forward(C.byref(w), pixels[i0:i0 + m].data_ptr(), pcode, m, F,
                                           out[i0:i0 + m].data_ptr(), ws.data_ptr(), ws.numel(), self.code, ops._stream()),
                    "vtc_vision_forward")
        return out


class PackedText:
    def __init__(self, sd: SD, prefix: str, dtype, heads: Optional[int] = None, half_layers: Optional[int] = None):
        self._src = (sd, prefix, dtype, heads)          # references to the caller's tensors: the bf16 re-pack of the range guard
        self._flag_dev = self._flag_host = None
        self._calibrated = False
        self.range_fallbacks = 0
        self._build(half_layers)

    def _build(self, half_layers):
        sd, prefix, dtype, heads = self._src
        sd, dev = _host_sd(sd, prefix, skip=prefix + "visual.")
        self.dtype, self.code, self.keep = dtype, ops.dtype_code(dtype), _Keep(dev)
        k = self.keep
        w = L.TextW()
        w.vocab, w.width = sd["token_embedding.weight"].shape
        w.heads = heads or w.width // 64
        w.layers = _n_layers(sd, "transformer")
        w.ctx = sd["positional_embedding"].shape[0]
        w.embed_dim = sd["text_projection"].shape[1]
        w.tok_emb, w.pos = k.f32(sd["token_embedding.weight"]), k.f32(sd["positional_embedding"])
        w.ln_final_g, w.ln_final_b = k.f32(sd["ln_final.weight"]), k.f32(sd["ln_final.bias"])
        w.proj_t = k.mat(sd["text_projection"].t(), torch.float32)
        w.half_layers = min(w.layers, TEXT_HALF_LAYERS if half_layers is None else int(half_layers)) if dtype == torch.bfloat16 else 0
        w.flags = DEFAULT_FLAGS
        self.blocks = _pack_blocks(sd, "transformer", w.layers, dtype, k, False, False, half_layers=w.half_layers,
                                   fold_ln=LN_FOLD_PACK)
        w.blocks = self.blocks
        self.w = w
        if w.half_layers > 0 and self._flag_dev is None:
            self._flag_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()

    # ---- range guard of the IEEE-half blocks ---------------------------------------------------------------------------------
    # bf16 mode runs the text blocks on IEEE-half operands (the bf16 rounding floor of this tower is above 1e-3).
    # Half has 5 exponent bits: a residual-stream or MLP-hidden value beyond +-65504 becomes inf and the embedding NaN.  Synthetic
    # and typical CLIP weights stay orders of magnitude inside (tests push the stream to 1e3-1e4), but nothing in a checkpoint
    # promises it, so: every forward ends with vtc_nonfinite_flag on the tower's output (one tiny launch) and an ASYNC copy of
    # the flag to pinned host memory -- no sync.  The FIRST forward after packing (which synchronises anyway: the weights were
    # just uploaded) checks the flag at once and, if set, re-packs the blocks as bf16 (more range, the same speed, a higher
    # rounding floor) and recomputes; later forwards read the pinned flag on entry (the previous call's verdict) and do the
    # same switch with a warning -- the overflowed call's NaN embeddings have been returned by then and say so themselves.
    def _range_switch(self, why: str):
        import warnings
        warnings.warn(f"vtc_amd text tower: {why}: a value left the IEEE-half range (+-65504) in the half-operand blocks; "
                      "re-packing the text blocks as bf16 operands (VTC_TEXT_HALF_LAYERS=0 semantics: wider range, rounding floor "
                      "~1e-3 instead of ~1.5e-4)", RuntimeWarning, stacklevel=3)
        self.range_fallbacks += 1
        self._flag_dev.zero_()
        self._flag_host.zero_()
        self._build(0)

    def _range_guard(self, out: torch.Tensor, run):
        if self.w.half_layers <= 0 or self._flag_dev is None:
            return out
        L.check(L.lib().vtc_nonfinite_flag(out.data_ptr(), out.numel(), self._flag_dev.data_ptr(), ops._stream()), "vtc_nonfinite_flag")
        if not self._calibrated:
            self._calibrated = True
            if int(self._flag_dev.item()) != 0:            # first call after packing: synchronous
                self._range_switch("first forward after packing")
                return run()
            return out
        self._flag_host.copy_(self._flag_dev, non_blocking=True)
        return out

    @ops.on_device
    def forward(self, ids: torch.Tensor, ragged: Optional[bool] = None, ids_b: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids [S, ctx] int64 (+ ids_b [S2, ctx]: more sequences of the same call, e.g. titles + comments, without a
        concatenated copy) -> [S (+ S2), embed] fp32.  One library call, no torch compute and no host sync: on the ragged
        path the EOT positions, their prefix sums and the row count are computed on the device (vtc_text_forward2).

        Memory: because the row count of a ragged batch is known on the device only, the workspace is sized for the
        DENSE bound (S + S2) * ctx rows -- about 2x the rows actually computed at the synthetic length distribution (0.48 of the
        tokens) -- and the 256-vs-128 tile choice of the GEMMs is made on that bound too (the kernels themselves read the true
        count on the device and walk only its tiles).  `forward_host_offsets` is the exact-size alternative at the price of one
        D2H sync; TEXT_CHUNK bounds the workspace of the DENSE path (both id arrays are walked in chunks), not of the ragged one."""
        if self._flag_host is not None and self.w.half_layers > 0 and int(self._flag_host[0]) != 0:
            self._range_switch("an earlier forward")       # pinned host memory: no synchronisation
        w = self.w
        ids = ops._gpu(ids, torch.int64, "token ids")
        if ids.dim() != 2 or ids.shape[1] != w.ctx:
            raise ValueError(f"expected [S,{w.ctx}] token ids, got {tuple(ids.shape)}")
        S, Sb = ids.shape[0], 0
        if ids_b is not None:
            ids_b = ops._gpu(ids_b, torch.int64, "token ids (second array)")
            if ids_b.dim() != 2 or ids_b.shape[1] != w.ctx:
                raise ValueError(f"expected [S,{w.ctx}] token ids, got {tuple(ids_b.shape)}")
            Sb = ids_b.shape[0]
        out = torch.empty(S + Sb, w.embed_dim, dtype=torch.float32, device=ids.device)
        lib = L.lib()
        rag = TEXT_RAGGED if ragged is None else ragged
        chunk = TEXT_CHUNK if (TEXT_CHUNK > 0 and not rag) else S + Sb

        def run():
            w = self.w                                     # (re-read: the range guard may have re-packed)
            ws = _ws(lib.vtc_text_workspace_bytes(C.byref(w), min(chunk, S + Sb), self.code), ids.device)
            if chunk >= S + Sb:
                L.check(lib.vtc_text_forward2(C.byref(w), ids.data_ptr(), S, ids_b.data_ptr() if Sb else None, Sb, int(bool(rag)),
                                              out.data_ptr(), ws.data_ptr(), ws.numel(), self.code, ops._stream()), "vtc_text_forward2")
                return out
            for arr, n_arr, base in ((ids, S, 0), (ids_b, Sb, S)):      # dense, chunked: the first id array, then the second
                for s0 in range(0, n_arr, chunk):
                    n = min(chunk, n_arr - s0)
                    L.check(lib.vtc_text_forward(C.byref(w), arr[s0:s0 + n].data_ptr(), n, out[base + s0:base + s0 + n].data_ptr(),
                                                 ws.data_ptr(), ws.numel(), self.code, ops._stream()), "vtc_text_forward")
            return out

        return self._range_guard(run(), run)

    @ops.on_device
    def forward_host_offsets(self, ids: torch.Tensor) -> torch.Tensor:
        """The ragged tower with the prefix sums computed by the HOST (vtc_text_forward_ragged: exact-size workspace and grids,
        at the price of torch compute + one D2H sync before the launch) -- round 2's path, kept as an entry point and as the
        cross-check of the device-side bookkeeping."""
        w = self.w
        ids = ops._gpu(ids, torch.int64, "token ids")
        S = ids.shape[0]
        out = torch.empty(S, w.embed_dim, dtype=torch.float32, device=ids.device)
        lib = L.lib()
        lens = ids.argmax(dim=-1).to(torch.int32) + 1
        offsets = torch.zeros(S + 1, dtype=torch.int32, device=ids.device)
        offsets[1:] = torch.cumsum(lens, 0)
        total = int(offsets[-1].item())
        ws = _ws(lib.vtc_text_ragged_workspace_bytes(C.byref(w), S, total, self.code), ids.device)
        L.check(lib.vtc_text_forward_ragged(C.byref(w), ids.data_ptr(), S, offsets.data_ptr(), total, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), self.code, ops._stream()), "vtc_text_forward_ragged")
        return out


_ACTS = {None: (L.ACT_NONE, 1.0), "none": (L.ACT_NONE, 1.0), "normalize": (L.ACT_NORMALIZE, 1.0),
         "squash": (L.ACT_SQUASH, 1.0), "squash10": (L.ACT_SQUASH, 10.0), "squash1p2": (L.ACT_SQUASH, 1.2),
         "squash1p5": (L.ACT_SQUASH, 1.5), "squash1p8": (L.ACT_SQUASH, 1.8), "tanh": (L.ACT_TANH, 1.0),
         # eval-mode statistics of mean_center_bn = BatchNorm1d(D, affine=False) (model/model.py:42-61)
         "sub_mean": (L.ACT_SUB_MEAN, 1.0), "bn": (L.ACT_BN, 1.0)}


#: True when another rank of the job computes on this process's card (dist.mark_shared_cards): the one-launch CAM is off for
#: every CAM forward from then on (VTC_CAM_NO_FUSED in vtc_cam_w.flags -- a per-call model flag, so it also reaches modules
#: packed before the process group existed)
_CAM_SHARED_CARD = False


def set_cam_shared_card(shared: bool):
    global _CAM_SHARED_CARD
    _CAM_SHARED_CARD = bool(shared)


class PackedCam:
    def __init__(self, sd: SD, dtype, heads: int, init_from_avg: bool, residual_activation):
        if residual_activation not in _ACTS:
            raise ValueError(f"unknown residual_activation {residual_activation!r} (model/model.py:30-80)")
        sd, dev = _host_sd({k_: v for k_, v in sd.items() if k_.startswith(("final_transformer.", "final_linear.", "mask_embedding",
                                                                               "mean_center_bn."))}, "")
        # one arena for the module's weights (plain + folded copies of the projections): see _Keep
        arena = int(2.1 * sum(v.numel() for v in sd.values()) * 4) + (1 << 20)
        self.dtype, self.code, self.keep = dtype, ops.dtype_code(dtype), _Keep(dev, arena_bytes=arena)
        k = self.keep
        w = L.CamW()
        w.width = sd["final_linear.weight"].shape[0]
        w.heads = heads
        if w.width % heads or w.width // heads > 128:
            raise NotImplementedError(f"CAM head_dim must be an integer <= 128 on the HIP path (width {w.width}, heads {heads})")
        w.layers = _n_layers(sd, "final_transformer")
        w.init_from_avg = int(bool(init_from_avg))
        w.residual_activation, w.squash_scale = _ACTS[residual_activation]
        w.final_linear = k.mat(sd["final_linear.weight"], dtype)
        w.mask_embedding = k.f32(sd["mask_embedding"].reshape(-1))
        self.blocks = _pack_blocks(sd, "final_transformer", w.layers, dtype, k, False, False, fold_ln=True, fold_fp32=True)
        w.blocks = self.blocks
        if residual_activation in ("sub_mean", "bn"):
            if "mean_center_bn.running_mean" not in sd:
                raise KeyError("residual_activation %r needs mean_center_bn.running_mean / running_var in the state dict "
                               "(model/model.py:42-61)" % residual_activation)
            w.bn_mean = k.f32(sd["mean_center_bn.running_mean"].reshape(-1))
            w.bn_var = k.f32(sd["mean_center_bn.running_var"].reshape(-1))
        self.w = w

    @ops.on_device
    def forward(self, main: torch.Tensor, comm_feats: torch.Tensor, comments: torch.Tensor, fused: Optional[bool] = None) -> torch.Tensor:
        """main [B,D], comm_feats [B*nc,D] fp32, comments [B,nc,ctx] int64 -> adapted [B,D].
        fused=False: the multi-launch path for THIS call (the caller has another tower's kernels in flight on a second stream: the
        one-launch form's grid barrier needs every CU to itself and would serialise with them -- or give up; the small launches of the
        multi-launch form interleave)."""
        w = self.w
        if _CAM_SHARED_CARD:
            w.flags |= L.CAM_NO_FUSED
        if fused is False and not (w.flags & L.CAM_NO_FUSED):
            w.flags |= L.CAM_NO_FUSED
            try:
                return self.forward(main, comm_feats, comments)
            finally:
                w.flags &= ~L.CAM_NO_FUSED
        main, comm_feats = ops._gpu(main, torch.float32, "main"), ops._gpu(comm_feats, torch.float32, "comm_feats")
        comments = ops._gpu(comments, torch.int64, "comments")
        B, nc, ctx = comments.shape
        assert main.shape == (B, w.width) and comm_feats.shape == (B * nc, w.width)
        out = torch.empty(B, w.width, dtype=torch.float32, device=main.device)
        lib = L.lib()
        ws = _ws(lib.vtc_cam_workspace_bytes(C.byref(w), B, nc, self.code), main.device)
        L.check(lib.vtc_cam_forward(C.byref(w), main.data_ptr(), comm_feats.data_ptr(), comments.data_ptr(), ctx, B, nc,
                                    out.data_ptr(), ws.data_ptr(), ws.numel(), self.code, ops._stream()), "vtc_cam_forward")
        return out
