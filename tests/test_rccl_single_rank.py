"""GPU: the RCCL ("nccl" backend) code path on real hardware with the one rank a 1-GPU box offers.  The multi-rank logic is covered
on gloo (tests/test_dp_gloo.py); what gloo cannot show is that RCCL initialises on this image / GPU and that the stream-ordered
asynchronous collectives the training exchange is built from (parallel/dp.py: BucketedAllReduce - comm stream waits for an event of
the compute stream, async all-reduce of disjoint ranges of the flat buffer, stream-level wait) and the validation gather run on
device tensors.  Runs in a child process (its own process group; 127.0.0.1 rendezvous)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["UPA_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from ultralytics_pro_amd.parallel.dp import BucketedAllReduce, gather_ragged, max_over_ranks
dev = torch.device("cuda:0")
n = 3 * (1 << 20) + 77
flat = torch.arange(n, dtype=torch.float32, device=dev) * 0.5
want = flat.clone()
# the bucketed exchange, collectives issued on the comm stream behind an event of the compute stream
b = BucketedAllReduce(flat, [[(0, 1 << 20), (2 << 20, n)], [(1 << 20, 2 << 20)]])
assert b.covered() == n
b.active = lambda: True            # world 1: drive the path the N > 1 step takes
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
b.issue(1, [ev]); b.issue(0, [ev])
assert b.wait() == 2
torch.cuda.synchronize()
assert torch.equal(flat, want)     # SUM over one rank
ones = torch.ones(1, device=dev); dist.all_reduce(ones); assert int(ones.item()) == 1   # bench.py's rccl_ranks_seen
rows = torch.arange(5 * 6, dtype=torch.float32, device=dev).reshape(5, 6)
g_rows, g_counts = gather_ragged(rows, torch.tensor([2, 3], dtype=torch.int32, device=dev))
assert g_rows.shape[0] >= 5 and torch.equal(g_rows[:5].cpu(), rows.cpu()) and g_counts.cpu().tolist() == [2, 3]
assert abs(max_over_ranks(0.25, dev) - 0.25) < 1e-12
out = torch.empty(4, device=dev); dist.all_gather_into_tensor(out, torch.full((4,), 7.0, device=dev)); assert float(out.sum()) == 28.0
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", torch.version.hip)
"""


def test_rccl_initialises_and_runs_the_exchange_primitives_on_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", UPA_ROOT=str(ROOT))
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL initialisation did not complete within 240 s on this box")
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


TRAIN_CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["UPA_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
from ultralytics_pro_amd.engine.trainer import DetectionTrainer
from ultralytics_pro_amd.nn.tasks import DetectionModel
from ultralytics_pro_amd.utils import procedural as P
dev = torch.device("cuda:0")
x = P.synthetic_images(4, h=256, w=256).to(dev)
lab = P.synthetic_labels(4)

def run(bucketed):
    m = DetectionModel("yolov8n.yaml")
    P.apply_procedural_weights(m)
    tr = DetectionTrainer(m, dtype=torch.float32, device=dev, world_size=2)   # the N > 1 code path on the one rank there is
    table = None
    if bucketed:
        table = tr.enable_overlapped_allreduce(target_bytes=2 << 20)           # several buckets on yolov8n's 12.6 MB of gradients
        tr._buckets.active = lambda: True
        assert len(table) >= 3
    items = [tr.step(x, lab).clone() for _ in range(3)]
    torch.cuda.synchronize()
    # the replicate steps of the multi-rank loop (SURVEY 8e) as real RCCL collectives on the one rank there is: rank 0's EMA / BatchNorm
    # buffers, the stop flag, the rank-averaged validation loss - none may change rank 0's own values
    import ultralytics_pro_amd.parallel.dp as dp
    from ultralytics_pro_amd.engine.validator import DetectionValidator
    saved, dp._multi_rank = dp._multi_rank, (lambda: True)
    try:
        rb, erb = tr.RB.clone(), tr.ERB.clone()
        tr.sync_ema_buffers()
        assert tr.broadcast_stop(True) is True and tr.broadcast_stop(False) is False
        v = DetectionValidator()
        v.add_loss(items[0]); v.add_loss(items[1])
        got = v.reduce_loss()
        torch.cuda.synchronize()
        assert torch.equal(tr.RB, rb) and torch.equal(tr.ERB, erb)
        assert torch.allclose(got, (items[0] + items[1]) / 2)
    finally:
        dp._multi_rank = saved
    exposed = tr.allreduce_exposed_ms()
    n = tr.groups[-1][0] + tr.groups[-1][1]
    return torch.stack(items).cpu(), tr.P[:n].clone().cpu(), exposed, table

it_b, p_b, exposed, table = run(True)
it_s, p_s, none_, _ = run(False)
assert none_ is None                                   # nothing is recorded without buckets
assert exposed is not None and exposed >= 0.0, exposed  # the eager bucketed mode reports the wait it exposed (bench: allreduce_exposed_ms)
assert torch.equal(it_b, it_s) and torch.equal(p_b, p_s), "bucketed exchange (SUM over one rank) must leave the step bit-identical"
# every layer span was issued from the backward walk or by the late path; none is left pending
dist.barrier(); dist.destroy_process_group()
print("TRAIN_RCCL_OK buckets", [t[0] for t in table], "exposed_ms", round(exposed, 4))
"""


def test_trainer_bucketed_exchange_over_rccl_reports_exposed_time_and_equals_single_allreduce():
    """engine/trainer.py on RCCL with the one rank of this box: eager steps with `enable_overlapped_allreduce` (buckets issued from the
    backward walk on the communication stream behind events of the main and weight-gradient streams, late spans behind the same
    events) give bit-identical loss items and weights to the single all-reduce, and `allreduce_exposed_ms()` - the figure
    `bench.py --workload train --gpus N` prints - is populated in that mode."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", UPA_ROOT=str(ROOT))
    try:
        r = subprocess.run([sys.executable, "-c", TRAIN_CHILD], env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL initialisation / the training child did not complete within 420 s on this box")
    assert r.returncode == 0 and "TRAIN_RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
