"""GPU: the SMIN HIP path under DistributedDataParallel (BASELINE configs[3]).

Two jobs of fresh rank processes are started by conftest.py before this process initialises the GPU:
  gloo2 -- 2 ranks sharing cuda:0 under gloo: averaged shard gradients == single-process full-batch gradients (<= 1e-5
           relative) and both ranks hold bitwise-equal gradients after the all-reduce;
  rccl1 -- DDP over RCCL at world size 1: the two-stream step (boundary unit on the second HIP stream) under DDP's hooks
           reproduces the plain step's gradients.
An 8-rank RCCL run needs an 8-GPU node; the driver's scaling bench is the only place that exists."""
import json
import os

import pytest

from tests import conftest as C

pytestmark = pytest.mark.gpu


def _collect(mode):
    job = C.DDP_JOBS.get(mode)
    if job is None:
        pytest.skip("rank processes were not started (session was not selected with -m gpu, or SMIN_SKIP_DDP_TEST is set)")
    for p, log in zip(job["procs"], job["logs"]):
        try:
            rc = p.wait(timeout=900)
        except Exception:
            p.kill()
            raise AssertionError(f"{mode}: rank process hung\n" + open(log).read()[-3000:])
        assert rc == 0, f"{mode}: rank exited with {rc}\n" + open(log).read()[-3000:]
    return [json.load(open(os.path.join(job["outdir"], f"{mode}_rank{r}.json"))) for r in range(job["world"])]


@pytest.mark.timeout(1200)
def test_two_ranks_gloo_shard_gradients_equal_full_batch():
    res = _collect("gloo2")
    assert [r["rank"] for r in res] == [0, 1]
    for r in res:
        print(r)
        assert r["backend"] == "gloo" and r["world"] == 2
        assert r["native_so"], "libsmin_hip.so was not mapped in the rank process"
        assert r["worst_rel"] <= 1e-5, (r["worst_name"], r["worst_rel"])
        assert r["bitwise_equal_across_ranks"]
        assert abs(r["loss_full"] - r["loss_shard_mean"]) <= 1e-5 * abs(r["loss_full"])
        # wrap() reports what it switched off for gloo on device tensors
        assert r["overrides"] == {"overlap_boundary": False, "overlap_prep": False} and r["flags_after"] == [False, False]


@pytest.mark.timeout(1200)
def test_rccl_world1_two_stream_step_under_ddp():
    (r,) = _collect("rccl1")
    print(r)
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["native_so"]
    assert r["flags_after"][0] is True, "the boundary unit must stay on the second stream under RCCL"
    assert r["overrides"] == {}, "the one-node step keeps both overlaps under RCCL"
    assert r["worst_rel"] <= 1e-6, (r["worst_name"], r["worst_rel"])
