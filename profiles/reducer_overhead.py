"""What does an ACTIVE SharedGradReducer cost on the step, and which part of it is c10d / RCCL?  Runs bench.py's default
workload as a one-rank process group (DRTK_SINGLE_RANK_GROUP) with torch.distributed.all_reduce optionally replaced by a
stub that returns a finished work handle: the difference is the collective call itself, the rest is the reducer's stream
choreography (side stream, events, waits).
usage (GPU box): python profiles/reducer_overhead.py [stub]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.update(DRTK_SINGLE_RANK_GROUP="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
stub = len(sys.argv) > 1 and sys.argv[1] == "stub"
import torch.distributed as dist  # noqa: E402

if stub:
    real = dist.all_reduce

    class _Done:
        def wait(self):
            return True

    def fake(t, op=None, group=None, async_op=False):
        if t.numel() <= 2:  # the timing's MAX over ranks: keep it real
            return real(t, op=op, group=group, async_op=async_op)
        return _Done() if async_op else None

    dist.all_reduce = fake
sys.argv = ["bench.py", "--no-graph", "--cpu-sample-views", "0", "--kernel-steps", "1"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
