"""`python bench.py --gpus N` must start N ranks by itself. The launch path - re-exec under
torch.distributed.run on 127.0.0.1, rank bookkeeping, barrier / max-over-ranks timing and the packed single all-gather of the
top-k blocks - is driven here without GPUs: gloo backend, stub searcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_self_launches_n_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest-launch"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == n and r["rank_mask"] == (1 << n) - 1 and r["collectives_per_step"] == 1 and r["gathered_equals_expected"]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_db_sharded_launch_merges_to_the_global_topk(n):
    """`bench.py --gpus N --shard db` (the per-shard loop + merge of /root/reference/scripts/multiple_search.sh:71-107): the database split
    over the ranks, every rank answers all queries on its shard with local ids, one all-gather of the shard-major blocks, merge under (distance, id) with
    the shards' offsets - every rank ends with the global top-k (gloo, stub searcher; the GPU run uses gs_topk_merge_dev for the same merge)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest-launch", "--shard", "db"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == n and r["gathered_equals_expected"] and r["db_sharded_merge_equals_global_topk_on_every_rank"] is True


def test_single_rank_needs_no_launcher():
    from gsearch_amd import sharding as S
    env = dict(os.environ)
    env.pop("RANK", None)
    assert S.ensure_launched(1, "x.py", [], env) is None
    os.environ["RANK"] = "0"
    try:
        assert S.ensure_launched(8, "x.py", []) is None           # already a rank of somebody's launch: carry on
    finally:
        del os.environ["RANK"]


def test_packed_exchange_layout_single_rank():
    import torch
    from gsearch_amd import sharding as S
    ex = S.TopkExchange(4, 3, 1, torch.device("cpu"))
    ex.ids.copy_(torch.arange(12).view(4, 3))
    ex.dist.copy_(torch.arange(12, dtype=torch.float32).view(4, 3) / 8)
    ex.exchange()
    ids, dist = ex.gathered()
    assert torch.equal(ids, torch.arange(12).view(4, 3)) and torch.equal(dist, torch.arange(12, dtype=torch.float32).view(4, 3) / 8)
    assert ex.send.numel() == 4 * 3 * 12                          # 12 bytes per neighbour, SURVEY 8d
