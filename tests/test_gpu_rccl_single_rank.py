"""RCCL itself, on the one GPU a test box has: a process group of world size 1 with backend "nccl" (= RCCL on ROCm).

The multi-rank logic of drtk_amd/dist.py is covered on CPU over gloo (tests/test_dist_gloo.py) and, two ranks on one GPU,
through bench.py over gloo (tests/test_gpu_bench_contract.py); neither executes RCCL.  A one-rank communicator does: it
loads librccl, creates the communicator on the device, and runs the reducer's real call sequence -- all-reduces issued
from autograd hooks on the side stream with async_op, work.wait() on that stream, the main stream waiting at finish() --
against the backend the 8-GPU job uses.  The sum over one rank is the identity, so the loss must equal that of a step
without any reducer exactly and the gradients to the summation-order noise of two runs of the atomically accumulating
backward kernels, and the collectives' events must have run.  Runs in a child process: the process group must not leak
into the test session.
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["DRTK_ROOT"])
import torch as th
import torch.distributed as dist
import drtk_amd
from drtk_amd import dist as ddist, synthetic as S

dev = th.device("cuda", 0)
# the product's own initialisation (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torch.distributed.run sets them), nccl branch
assert ddist.init_from_env(single_rank_group=True) == (0, 1, 0)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1

N, H, W, C = 3, 96, 128, 8
v_world, vi = S.uv_sphere(20, 24)
cams = [t.to(dev) for t in S.ring_cameras(N, W, H)]
vi = vi.to(dev)
attr0 = S.random_attributes(1, v_world.shape[0], C, seed=5)[:1].contiguous().to(dev)
v0 = v_world.to(dev)


def step(v, a_in, cast):
    campos, camrot, focal, princpt = cams
    v_pix = drtk_amd.transform(v[None].expand(N, -1, -1), campos, camrot, focal, princpt)
    a = cast(a_in).expand(N, -1, -1)
    index_img = drtk_amd.rasterize(v_pix, vi, H, W)
    depth_img, bary_img = drtk_amd.render(v_pix, vi, index_img)
    img = drtk_amd.interpolate(a, vi, index_img, bary_img)
    img = th.where((index_img != -1)[:, None], img, 0.0)
    img = drtk_amd.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
    loss = (img * img).sum() + depth_img.sum()
    loss.backward()
    return float(loss)


# (1) plain step, no reducer: the expected gradients (fp16 leaf: autograd's own .half() round trip of the f32 gradient)
v = v0.clone().requires_grad_(True)
a16 = attr0.half().requires_grad_(True)
loss_ref = step(v, a16, lambda p: p.float())
gv_ref, ga_ref = v.grad.clone(), a16.grad.clone()

# (2) the same step through the reducer over RCCL: vertices in their own collective, the fp16 leaf staged in float32
v = v0.clone().requires_grad_(True)
a16 = attr0.half().requires_grad_(True)
red = ddist.SharedGradReducer([v, [a16]], dtype=th.float32)
red.run_single_rank = True
red.record_timings = True
assert red.collectives_per_step() == 2 and red.flat.dtype == th.float32
for it in range(2):  # a second step on the same reducer: buffers re-armed, events reused
    red.zero_grad()
    loss = step(v, a16, red.upcast)
    assert set(red._pending) == {0, 1}, red._pending  # both collectives were launched from inside the backward pass
    red.finish()
    th.cuda.synchronize()
    assert loss == loss_ref, (loss, loss_ref)
    # atomically accumulated gradients: two runs differ by summation order, not by more
    tol = 1e-5 + 1e-5 * float(gv_ref.abs().max())
    assert float((v.grad - gv_ref).abs().max()) <= tol, float((v.grad - gv_ref).abs().max())
    assert a16.grad.dtype == th.float16
    assert float((a16.grad.float() - ga_ref.float()).abs().max()) <= 2e-3 * float(ga_ref.float().abs().max()) + 1e-6
    t = red.timings_ms()  # (launch -> completion of the collectives on the side stream, main stream's wait in finish())
    assert t is not None and t[0] > 0 and t[1] >= 0, t

# (3) a bare all-reduce of the flat buffer: RCCL's sum over one rank leaves it unchanged
before = red.flat.clone()
dist.all_reduce(red.flat)
th.cuda.synchronize()
assert th.equal(before, red.flat)
ddist.barrier_and_sync()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK", th.cuda.get_device_name(0))
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_reducer_over_rccl_world_size_one():
    env = dict(os.environ, DRTK_ROOT=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DRTK_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT, timeout=550)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
