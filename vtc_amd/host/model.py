"""Drop-in for the reference's ``model.model`` (model/model.py) on MI355X.

Same class names, constructor kwargs, state-dict keys, ``forward`` signatures and return
triples ``(feats_vis, feats_text, sim)`` as the reference, so that
``config.init_obj("arch", module_arch)`` (utils/parse_config.py:97-112, evaluation/eval.py:88)
and strict ``load_state_dict`` (evaluation/eval.py:90-91) work unchanged:

    PretrainedCLIP                      model/model.py:308-371
    PretrainedCLIP_finaltf              model/model.py:374-480
    PretrainedCLIP_TimeSformer          model/model.py:483-506
    PretrainedCLIP_TimeSformer_finaltf  model/model.py:539-623

The forward pass is forward/eval only and runs entirely in libvtc_hip.so: inputs must be on
a ROCm GPU and the module in ``eval()`` mode, anything else raises (there is deliberately no
CPU or PyTorch fallback -- the CPU restatement lives in ``oracle/`` and is test-only).
Out of scope (raise): training-mode branches (random comment masking / skip adapter),
the audio branch, feature-MLP baselines.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops, towers
from . import clip_arch

__all__ = ["PretrainedCLIP", "PretrainedCLIP_finaltf", "PretrainedCLIP_TimeSformer",
           "PretrainedCLIP_TimeSformer_finaltf", "PretrainedCLIPBase"]


def normalize(x):
    return ops.normalize_rows(x)



# Any registration of a parameter / buffer / submodule on any nn.Module invalidates the cached tensor lists of the wrappers
# (PretrainedCLIPBase._signature): a counter, bumped by torch's global registration hooks.
_REG_EPOCH = [0]


def _bump_reg_epoch(*_args):
    _REG_EPOCH[0] += 1
    return None


for _reg in ("register_module_parameter_registration_hook", "register_module_buffer_registration_hook",
             "register_module_module_registration_hook"):
    getattr(torch.nn.modules.module, _reg)(_bump_reg_epoch)

_DTYPE_NAMES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "16": torch.bfloat16,
                "f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "32": torch.float32,
                # round 6: IEEE-half operands in BOTH towers (the bf16 mode already runs the text blocks on half): 11 significant bits
                # instead of 8 at the same MFMA rate and bytes, range +-65504 (the non-finite watchdog says when a checkpoint leaves it)
                "f16": torch.float16, "fp16": torch.float16, "half": torch.float16, "float16": torch.float16}


def parse_compute_dtype(name):
    """"bf16" | "f32" (and spellings) -> torch dtype; None stays None."""
    if name is None or isinstance(name, torch.dtype):
        return name
    try:
        return _DTYPE_NAMES[str(name).strip().lower()]
    except KeyError:
        raise ValueError(f"vtc_amd: unknown compute dtype {name!r}; known: bf16, f16, f32") from None


def default_compute_dtype():
    """The arithmetic of a freshly constructed wrapper: ``VTC_COMPUTE_DTYPE`` (bf16 | f32), else 16-bit operands with fp32
    accumulation (BASELINE configs[1..2]).  NOTE for drop-in users: the reference computes in fp32 end to end
    (model/model.py:318 ``self.model.float()``); 16-bit mode reproduces its embeddings to 1e-3, fp32 mode to 1e-5."""
    return parse_compute_dtype(__import__("os").environ.get("VTC_COMPUTE_DTYPE")) or torch.bfloat16


nonfinite_bits = ops.nonfinite_bits


def nonfinite_cause(device) -> str:
    """What this build knows can put a NaN into an embedding, for the error message."""
    from .. import _lib as L
    idx = torch.device(device).index
    causes = []
    if L.lib().vtc_cam_fused_gave_up(idx if idx is not None else torch.cuda.current_device()):
        causes.append("a one-launch CAM gave up at a grid barrier (another process or a collective held the card's CUs); the "
                      "multi-launch CAM is selected from now on")
    causes.append("an IEEE-half overflow in the text tower's half-operand blocks (the tower re-packs itself as bf16 at its next forward "
                  "and warns); with compute_dtype = torch.float16 an overflow of +-65504 in the vision tower (choose bfloat16: wider range); "
                  "or non-finite weights / inputs")
    return "; or ".join(causes)


def raise_if_nonfinite(where: str, feats_a: torch.Tensor, feats_b: torch.Tensor):
    """The unconditional finite check of the eval entry points (synchronises): NaN embeddings never reach a recall figure."""
    bits = nonfinite_bits(feats_a, feats_b)
    if bits:
        which = " and ".join(n for n, m in (("visual", 1), ("text", 2)) if bits & m)
        raise RuntimeError(f"vtc_amd {where}: non-finite values in the {which} embeddings -- refusing to rank them.  Possible cause: "
                           f"{nonfinite_cause(feats_a.device)}.  Re-run (the fallbacks are now active) or use --dtype f32")


class PretrainedCLIPBase(nn.Module):
    #: arithmetic of the GEMM/attention operands on the HIP path (fp32 everywhere else); per instance from
    #: default_compute_dtype() at construction, assignable afterwards (a change re-packs the weights)
    compute_dtype = torch.bfloat16
    #: multiply temporal_fc and timeattn.out_proj together at pack time (one GEMM instead of two)
    fuse_temporal = True
    nframes = 8  # model/model.py:488,557

    def _common_init(self):
        self.compute_dtype = default_compute_dtype()
        if getattr(self, "residual_activation", None) in ["sub_mean", "bn"]:
            # model/model.py:134-139: running statistics only (forward is eval-mode here, see _check_eval)
            self.mean_center_bn = nn.BatchNorm1d(self.feature_dim, affine=False, momentum=0.2)
        self._packed = {}

    # ---- packed-weight cache ---------------------------------------------------------------
    def _signature(self):
        """(storage, version) of every parameter and buffer: a changed weight re-packs.  The list of tensors itself is cached
        (walking ~200 modules per forward cost 0.66 ms; 0.07 ms from the list) together with WHERE each tensor is registered
        (the owner module's `_parameters` / `_buffers` dict and its key): the list is rebuilt when any module anywhere registers
        a parameter, buffer or submodule (global registration hooks below bump `_REG_EPOCH`) and ALSO when a registered slot no
        longer holds the cached object -- paths that write `module._parameters[name]` directly (`.to()` under
        torch.__future__.set_overwrite_module_params_on_conversion(True), parametrize / pruning utilities) fire no hook."""
        ent = self.__dict__.get("_sig_list")
        fresh = self.__dict__.get("_sig_epoch") == _REG_EPOCH[0] and ent is not None
        if fresh:
            for d, k, t in ent:
                if d.get(k) is not t:
                    fresh = False
                    break
        if not fresh:
            ent = []
            for mod in self.modules():
                ent += [(mod._parameters, k, t) for k, t in mod._parameters.items() if t is not None]
                ent += [(mod._buffers, k, t) for k, t in mod._buffers.items() if t is not None]
            self.__dict__["_sig_list"] = ent
            self.__dict__["_sig_epoch"] = _REG_EPOCH[0]
        return tuple((t.data_ptr(), t._version) for _, _, t in ent) + (self.compute_dtype, self.fuse_temporal)

    def _pack(self):
        sig = self._signature()
        if self._packed.get("sig") != sig:
            sd = {k: v for k, v in self.state_dict().items()}
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("vtc_amd models run on an MI355X only: call .to('cuda') first "
                                   "(the CPU restatement lives in oracle/ and is test infrastructure)")
            p = {"sig": sig}
            p["visual"] = towers.PackedVision(sd, "model.visual.", self.compute_dtype, self.fuse_temporal)
            p["text"] = towers.PackedText(sd, "model.", self.compute_dtype, heads=self.model.transformer.heads)
            if hasattr(self, "final_transformer"):
                # the CAM sees B*(1+nc) tokens of width 512 -- negligible work -- so it always runs in fp32
                p["cam"] = towers.PackedCam(sd, torch.float32, self.final_transformer.heads, self.init_from_avg,
                                            self.residual_activation)
            self._packed = p
        return self._packed

    #: run the visual tower on a side HIP stream while the text tower runs on the caller's stream: both
    #: towers are chains of persistent-grid kernels, and one tower's kernels fill the other's launch gaps and tails
    overlap_towers = __import__("os").environ.get("VTC_OVERLAP", "1") != "0"
    _video_tower = False      # True: model.visual takes [B,F,3,H,W] (TimeSformer wrappers)

    def _encode_both(self, vis, title, texts_b=None, text_tail=None):
        """(visual features, text features); the two towers are independent until the CAM / similarity.  texts_b: a second
        id array encoded in the same text-tower call (rows after the titles').  text_tail(ft, overlapped) -> ft': what follows the
        text tower and does not need the visual features (the CAM on the text branch) -- enqueued on the caller's stream BEFORE the
        join, so that it runs under the visual tower, which is the longer chain (round 6)."""
        pk = self._pack()        # ONE signature walk per forward; the towers below use the packed structs directly
        enc = pk["visual"].forward if self._video_tower else (lambda v: self._encode_vis(v, pk))
        if not self.overlap_towers or len(vis.shape) == 2:
            # (visual tower first: on ONE stream the towers share a workspace, and the profiler reads the ragged text tower's device-side
            # row count back after the forward -- it must still stand there)
            fv = enc(vis)
            ft = pk["text"].forward(title, ids_b=texts_b)
            return fv, (text_tail(ft, False) if text_tail is not None else ft)
        # The weights were packed (converted / transposed / fused) above, on the CALLER's stream, before the fork: the side stream inherits
        # the dependency through wait_stream, and the packed tensors belong to the caller stream's allocator pool.
        # (Packed lazily inside the fork, the conversions would be enqueued on the side stream only, and the text
        # tower on the caller's stream could read half-converted weights.)
        dev = vis.device
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            side = getattr(self, "_side", None)
            if side is None or side.device != dev:
                side = self._side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                fv = enc(vis)
            ft = pk["text"].forward(title, ids_b=texts_b)
            if text_tail is not None:
                ft = text_tail(ft, True)
            cur.wait_stream(side)
            fv.record_stream(cur)
        return fv, ft

    # ---- non-finite watchdog -----------------------------------------------------------------
    # Two paths of this build can hand back NaN rows with rc 0: a text tower whose IEEE-half blocks overflowed in a batch after the
    # first (towers.PackedText._range_guard switches to bf16 at the NEXT call), and a one-launch CAM whose grid barrier gave up
    # (cam.hip).  So every forward ends with ONE vtc_nonfinite_flag2 launch over the two embedding sets it returns and an ASYNC copy
    # of the word to pinned host memory -- no synchronisation -- and the word is read (a) at the model's next forward and (b) by
    # check_finite(), which synchronises; both raise with the cause.  RecallAtK and the eval entry points check their inputs too.
    #: False switches the per-forward launch off (the checks of RecallAtK / eval.py remain)
    nonfinite_watchdog = True

    def _finish(self, fv, ft):
        """The two `normalize` calls that end every forward (model/model.py:263-264, 366-367) and the watchdog, in ONE launch."""
        if not (self.nonfinite_watchdog and fv.is_cuda and ft.is_cuda and fv.dim() == 2 and ft.dim() == 2 and fv.shape[1] == ft.shape[1]):
            fv, ft = normalize(fv), normalize(ft)
            self._watch(fv, ft)
            return fv, ft
        st = self.__dict__.get("_nf")
        if st is None or st[0].device != fv.device:
            st = self.__dict__["_nf"] = (torch.zeros(1, dtype=torch.int32, device=fv.device), torch.zeros(1, dtype=torch.int32).pin_memory())
        fv, ft = ops.normalize_rows2(fv, ft, st[0])
        with torch.cuda.device(fv.device):
            st[1].copy_(st[0], non_blocking=True)
        return fv, ft

    def _watch(self, fv, ft):
        if not self.nonfinite_watchdog or not (fv.is_cuda and ft.is_cuda) or fv.dtype != torch.float32 or ft.dtype != torch.float32:
            return
        st = self.__dict__.get("_nf")
        if st is None or st[0].device != fv.device:
            st = self.__dict__["_nf"] = (torch.zeros(1, dtype=torch.int32, device=fv.device), torch.zeros(1, dtype=torch.int32).pin_memory())
        ops.nonfinite_flag2(fv, ft, st[0])
        with torch.cuda.device(fv.device):
            st[1].copy_(st[0], non_blocking=True)

    def _raise_watch(self, bits, when):
        st = self.__dict__["_nf"]
        st[0].zero_()
        st[1].zero_()
        which = " and ".join(n for n, m in (("visual", 1), ("text", 2)) if bits & m)
        raise RuntimeError(f"vtc_amd {type(self).__name__}: {when} returned non-finite values in its {which} embeddings.  Possible cause: "
                           f"{nonfinite_cause(st[0].device)}.  The flag is cleared: the next forward runs with the fallbacks active")

    def check_finite(self):
        """Synchronise the device and raise if any forward of this model since the last check returned NaN / inf embeddings."""
        st = self.__dict__.get("_nf")
        if st is None:
            return
        torch.cuda.synchronize(st[0].device)
        bits = int(st[1][0]) | int(st[0].item())
        if bits:
            self._raise_watch(bits, "a forward since the last check")

    def _check_mode(self, *tensors):
        st = self.__dict__.get("_nf")
        if st is not None and int(st[1][0]) != 0:         # pinned host word: the verdict of forwards whose copy has landed, no synchronisation
            self._raise_watch(int(st[1][0]), "an earlier forward")
        if self.training:
            raise RuntimeError("vtc_amd implements the forward/eval path only: call .eval() "
                               "(training branches model/model.py:199-201,236-246 are out of scope)")
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError("vtc_amd: inputs must be on the GPU (no CPU fallback)")

    # ---- towers ----------------------------------------------------------------------------
    def encode_image(self, image):
        return self._pack()["visual"].forward(image)

    def encode_text(self, text):
        return self._pack()["text"].forward(text)

    def _encode_vis(self, vis, pk=None):
        """vis.ndim dispatch of model/model.py:327-338 / :459-470."""
        shp = vis.shape
        if len(shp) == 2 and shp[1] == self.feature_dim:
            return vis.float()                                   # precomputed feature
        tower = (pk or self._pack())["visual"]
        if len(shp) == 4:
            return tower.forward(vis)
        if len(shp) == 5:                                        # frames -> mean over time
            f = tower.forward(vis.reshape(shp[0] * shp[1], shp[2], shp[3], shp[4]))
            return ops.mean_groups(f, shp[1])
        raise ValueError(f"unsupported visual input shape {tuple(shp)}")

    def _forward_with_cam(self, vis, title, comments):
        """forward() of the two *_finaltf wrappers up to the normalised pair: model/model.py:458-478 / :596-621 = towers (:464-472) +
        _encode_with_comments (:216-266, eval path).  Same arithmetic as _encode_all + _encode_with_comments below; what differs is the
        ORDER OF ENQUEUEING: with the text branch adapted, the CAM needs the title and comment features only, so it is enqueued behind
        the text tower on the caller's stream while the visual tower still runs on the side stream (multi-launch form: see
        PackedCam.forward) -- at small batch the CAM's ~0.2 ms left the critical path."""
        branch = self.branch_to_adapt_val
        if branch not in ("text", "image", "skip"):
            raise Exception("Unknown branch_to_adapt")
        if branch == "skip" or comments is None:
            fv, ft = self._encode_both(vis, title)
        else:
            b, ncomms, ntoks = comments.shape
            ids_c = comments.reshape(b * ncomms, ntoks)
            if branch == "text":
                def tail(ft_all, overlapped):
                    return self._packed["cam"].forward(ft_all[:b], ft_all[b:], comments, fused=False if overlapped else None)
                fv, ft = self._encode_both(vis, title, ids_c, text_tail=tail)
            else:
                fv, ft_all = self._encode_both(vis, title, ids_c)
                fv, ft = self._packed["cam"].forward(fv, ft_all[b:], comments), ft_all[:b]
        return self._finish(fv, ft)

    def _encode_all(self, vis, title, comments):
        """Both towers for the *_finaltf wrappers.  The reference encodes the titles (model/model.py:472) and the
        comments (:210) in two text-tower calls; the sequences are independent, so here they go through ONE call
        (1 + nc sequences per pair: bigger GEMMs, 7 launches per layer instead of 14; the library reads the two id arrays
        where they lie -- no concatenated copy) and are split afterwards -- bit-identical per sequence
        (tests/test_gpu_towers.py batch-independence test)."""
        if self.branch_to_adapt_val == "skip" or comments is None:
            fv, ft = self._encode_both(vis, title)
            return fv, ft, None
        b, ncomms, ntoks = comments.shape
        fv, ft_all = self._encode_both(vis, title, comments.reshape(b * ncomms, ntoks))
        return fv, ft_all[:b], ft_all[b:]

    def _encode_with_comments(self, feats_vis, feats_title, comments, feats_comm=None):
        """model/model.py:216-266, eval path."""
        branch = self.branch_to_adapt_val
        if branch not in ("text", "image", "skip"):
            raise Exception("Unknown branch_to_adapt")
        if branch != "skip":
            b, ncomms, ntoks = comments.shape
            if feats_comm is None:
                feats_comm = self.encode_text(comments.reshape(b * ncomms, ntoks))
            cam = self._packed["cam"] if self._packed.get("cam") is not None and feats_comm is not None else self._pack()["cam"]
            if branch == "text":
                feats_title = cam.forward(feats_title, feats_comm, comments)
            else:
                feats_vis = cam.forward(feats_vis, feats_comm, comments)
        return normalize(feats_vis), normalize(feats_title)

    def _sim(self, fv, ft):
        return ops.similarity(fv, ft, self.model.logit_scale)

    def _freeze(self, branch_to_freeze):
        """model/model.py:268-305 (requires_grad bookkeeping only; kept for ctor compatibility)."""
        self.branch_to_freeze = branch_to_freeze
        if branch_to_freeze is False or branch_to_freeze == "none":
            return
        did = False
        if "visual" in branch_to_freeze:
            did = True
            for p in self.model.visual.parameters():
                p.requires_grad = False
        if "text" in branch_to_freeze:
            did = True
            for p in self.model.transformer.parameters():
                p.requires_grad = False
        if "all" in branch_to_freeze:
            did = True
            for p in self.model.parameters():
                p.requires_grad = False
        if "finaltf" in branch_to_freeze:
            did = True
            if hasattr(self, "final_transformer"):
                for p in list(self.final_transformer.parameters()) + list(self.final_linear.parameters()):
                    p.requires_grad = False
                self.mask_embedding.requires_grad = False
        if not did:
            raise Exception("Unknown branch_to_freeze")

    def _init_cam(self, n_layers, n_heads, init_from_avg):
        """model/model.py:396-400, :440-452."""
        hd, rem = divmod(self.feature_dim, int(n_heads))
        if rem or hd > 128:
            # known at construction, so said at construction (never at the first forward): the CAM's attention core covers head_dim 64
            # (every reference config: 512 / 8; the MFMA kernels) and any head_dim <= 128 (ViT-L/14's 768 / 8 = 96: a small
            # generic kernel, round 4)
            raise NotImplementedError(f"Context Adapter Module: head_dim = feature_dim / n_heads must be an integer <= 128 on the HIP path "
                                      f"(feature_dim {self.feature_dim}, n_heads {n_heads})")
        self.final_transformer = clip_arch.Transformer(width=self.feature_dim, layers=int(n_layers), heads=int(n_heads))
        self.final_linear = nn.Linear(self.feature_dim, self.feature_dim, bias=False)
        self.mask_embedding = nn.Parameter(torch.randn(1, self.feature_dim))
        if init_from_avg:
            for blk in self.final_transformer.resblocks:
                blk.mlp.c_proj.weight.data.zero_()
                blk.mlp.c_proj.bias.data.zero_()
                blk.attn.out_proj.weight.data.zero_()
        nn.init.constant_(self.final_linear.weight, 0.0)


class PretrainedCLIP(PretrainedCLIPBase):
    def __init__(self, model_type="ViT-B/32", freeze=False, residual_activation=None, comment_fusion=None):
        super().__init__()
        self.model = clip_arch.load(model_type, device="cpu")
        self.feature_dim = self.model.ln_final.normalized_shape[0]
        self.residual_activation = residual_activation
        self.comment_fusion = comment_fusion
        self._common_init()
        self._freeze(freeze)

    def forward(self, vis, title, comments=None):
        self._check_mode(vis, title, comments)
        if comments is None or self.comment_fusion is None or self.comment_fusion == "None":
            feats_vis, feats_text = self._encode_both(vis, title)
        else:
            if self.comment_fusion != "averaging":
                raise Exception("Comment fusion method not specified.")
            b, ncomms, ntoks = comments.shape
            # titles and comments in ONE text-tower call (two id arrays, no concatenated copy), then the mean of a title's
            # embedding and its comments' (model/model.py:357-362)
            feats_vis, ft_all = self._encode_both(vis, title, comments.reshape(b * ncomms, ntoks))
            feats_text = ops.mean_head_groups(ft_all[:b], ft_all[b:], ncomms)
        feats_vis, feats_text = self._finish(feats_vis, feats_text)
        return feats_vis, feats_text, self._sim(feats_vis, feats_text)


class PretrainedCLIP_finaltf(PretrainedCLIPBase):
    def __init__(self, model_type="ViT-B/32", freeze=False, branch_to_adapt="text", branch_to_adapt_val="text",
                 residual_activation=None, n_layers=2, n_heads=8, init_from_avg=True, random_comment_masking=False,
                 random_skip_adapter=True, init_audio_model=False, audio_model_ckpt=None, clip_audio_ckpt=None):
        super().__init__()
        if init_audio_model:
            raise NotImplementedError("the audio branch (model/model.py:409-438) needs the external GDT repository: out of scope")
        self.model = clip_arch.load(model_type, device="cpu")
        self.feature_dim = self.model.ln_final.normalized_shape[0]
        self.branch_to_adapt, self.branch_to_adapt_val = branch_to_adapt, branch_to_adapt_val
        self.residual_activation = residual_activation
        self.init_from_avg = init_from_avg
        self.random_comment_masking, self.random_skip_adapter = random_comment_masking, random_skip_adapter
        self.init_audio_model = False
        self._init_cam(n_layers, n_heads, init_from_avg)
        self._common_init()
        self._freeze(freeze)

    def forward(self, vis, title, comments):
        self._check_mode(vis, title, comments)
        feats_vis, feats_text = self._forward_with_cam(vis, title, comments)
        return feats_vis, feats_text, self._sim(feats_vis, feats_text)


class PretrainedCLIP_TimeSformer(PretrainedCLIPBase):
    _video_tower = True

    def __init__(self, model_type="ViT-B/32", freeze=False, residual_activation=None):
        super().__init__()
        self.model = clip_arch.load(model_type, device="cpu")
        self.model.visual = clip_arch.make_timesformer_clip_vit_alt(nframes=self.nframes, model=model_type, clip_model=self.model)
        self.feature_dim = self.model.ln_final.normalized_shape[0]
        self.residual_activation = residual_activation
        self._common_init()
        self._freeze(freeze)

    def forward(self, im, text, comments=None):
        self._check_mode(im, text)
        feats_im, feats_text = self._encode_both(im, text)       # model.visual(im), model/model.py:497
        feats_im, feats_text = self._finish(feats_im, feats_text)
        return feats_im, feats_text, self._sim(feats_im, feats_text)


class PretrainedCLIP_TimeSformer_finaltf(PretrainedCLIPBase):
    _video_tower = True

    def __init__(self, model_type="ViT-B/32", freeze=False, branch_to_adapt="text", branch_to_adapt_val="text",
                 residual_activation=None, visual_device=None, n_layers=2, n_heads=8, init_from_avg=True,
                 random_comment_masking=False, random_skip_adapter=True):
        super().__init__()
        self.model = clip_arch.load(model_type, device="cpu")
        self.model.visual = clip_arch.make_timesformer_clip_vit_alt(nframes=self.nframes, model=model_type, clip_model=self.model)
        self.feature_dim = self.model.ln_final.normalized_shape[0]
        self.branch_to_adapt, self.branch_to_adapt_val = branch_to_adapt, branch_to_adapt_val
        self.residual_activation = residual_activation
        self.init_from_avg = init_from_avg
        self.random_comment_masking, self.random_skip_adapter = random_comment_masking, random_skip_adapter
        self._init_cam(n_layers, n_heads, init_from_avg)
        self._common_init()
        self._freeze(freeze)
        # model/model.py:590-611 splits the towers over two GPUs; both fit one MI355X, so the
        # argument is accepted and ignored (scale-out is one process per GPU).
        self.multigpu = False

    def forward(self, vis, title, comments):
        self._check_mode(vis, title, comments)
        feats_vis, feats_text = self._forward_with_cam(vis, title, comments)
        return feats_vis, feats_text, self._sim(feats_vis, feats_text)
