"""GPU: the one-launch CAM's give-up path.  A grid barrier that cannot complete sets a device-visible error word; from
then on every CAM forward on that device must take the multi-launch path -- same results, one line on stderr, rc 0 -- instead of failing
for the rest of the process, and `vtc_cam_fused_gave_up` must say so.  The word is sticky per process, so the scenario runs in a child
process on the TEST build of the library (vtc_amd/lib/libvtc_hip_testhooks.so = the product objects + cam.hip compiled with
-D VTC_TEST_HOOKS, where VTC_CAM_TEST_GAVE_UP=1 makes the word start set; the product library has no such switch)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %r)
from dataclasses import asdict
import torch
from oracle import arch as A
from vtc_amd import _lib as L
from vtc_amd.host import model as HM
from vtc_amd.host.clip_arch import ClipConfig
torch.set_grad_enabled(False)
a = A.TINY
sd = A.synth_model(a, 7, "clip_finaltf")
m = HM.PretrainedCLIP_finaltf(model_type=ClipConfig(**asdict(a)), branch_to_adapt_val="text", n_heads=2)
m.load_state_dict(sd, strict=True)
m = m.eval().cuda()
m.compute_dtype = torch.float32
m.overlap_towers = False      # (with the towers on two streams the text-branch CAM is enqueued under the visual tower in its multi-launch form and
                              #  never tries the one-launch path: this scenario is about the one-launch path's fallback)
B = 6
vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), 8).cuda()
title = A.synth_tokens(B, a, 9).cuda()
comments = A.synth_tokens(B * 5, a, 10, empty_frac=0.3).reshape(B, 5, -1).cuda()
lib = L.lib()
n0 = lib.vtc_debug_launch_count()
out1 = [o.clone() for o in m(vis, title, comments)]
n1 = lib.vtc_debug_launch_count()
out2 = [o.clone() for o in m(vis, title, comments)]
torch.cuda.synchronize()
assert lib.vtc_cam_fused_gave_up(0) == 1
# the reference: the same model with the one-launch path switched off by its flag
m._pack()["cam"].w.flags |= L.CAM_NO_FUSED
ref = m(vis, title, comments)
for o, r in zip(out1 + out2, list(ref) * 2):
    assert torch.isfinite(o).all() and torch.equal(o, r)
print("LAUNCHES", n1 - n0)
print("CHILD_OK")
'''


def test_cam_falls_back_to_the_multi_launch_path_after_a_barrier_gave_up():
    testlib = os.path.join(ROOT, "vtc_amd", "lib", "libvtc_hip_testhooks.so")
    assert os.path.exists(testlib), "build it: make -C vtc_amd/csrc testhooks (or __graft_entry__.build())"
    env = dict(os.environ, VTC_CAM_TEST_GAVE_UP="1", VTC_HIP_LIB=testlib)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stderr.count("the one-launch path is now OFF for this device") == 1          # said once, not per call
    launches = int(r.stdout.split("LAUNCHES")[1].split()[0])
    assert launches > 20                                                                  # towers + the 16-launch CAM, not the 1-launch one
