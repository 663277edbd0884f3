"""N>1 path on CPU: world_size-2 gloo processes exercising batch sharding, the optional
scatter/gather of row blocks and the max-over-ranks timing reduction that bench.py uses."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    from tiny_ntt_amd import dist as tdist
    rank, local_rank, world = tdist.init_process_group("gloo")
    assert world == 2 and dist.get_world_size() == 2
    batch, n = 37, 64
    full = (torch.arange(batch * n, dtype=torch.int64).reshape(batch, n) * 7919) if rank == 0 else None
    mine = tdist.scatter_rows(full, batch, n, torch.int64, "cpu")
    start, count = tdist.shard_rows(batch, world, rank)
    expect = torch.arange(batch * n, dtype=torch.int64).reshape(batch, n)[start:start + count] * 7919
    assert mine.shape == (count, n) and torch.equal(mine, expect), "scatter block wrong"
    out = mine + rank                      # stand-in for the per-rank kernel launch: rows stay independent
    got = tdist.gather_rows(out, batch, n)
    if rank == 0:
        ref = torch.arange(batch * n, dtype=torch.int64).reshape(batch, n) * 7919
        s1, c1 = tdist.shard_rows(batch, world, 1)
        ref[s1:s1 + c1] += 1
        assert torch.equal(got, ref), "gather wrong"
    else:
        assert got is None
    assert tdist.max_over_ranks(1.0 + rank) == 2.0
    assert tdist.sum_over_ranks(10.0 * (rank + 1)) == 30.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_shard_scatter_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rank} ok" in o
