"""CPU, world_size 2 (and 3) over gloo: the candidate-sharded step's host protocol.
Each rank solves its K/G slice (here with the oracle standing in for the local GPU solve),
packs its record into its row of the [world][R] int64 slot buffer, ONE all-reduce(min), select.
The result must equal a single np.argmin over the un-sharded candidate set, including the
lowest-index tie-break across ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, K_total, N, tie, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rovmpc_oracle as orc
        from rovmpc.sharded import ShardedMPC, shard_bounds
        from rovmpc.mpc import synthetic_problem
        from rovmpc.model import default_model
        m = default_model()
        om = orc.DynamicsModel(m.mean, m.scale, orc.SymbolicModel(m.expr_theta), orc.SymbolicModel(m.expr_gamma))
        cfg = orc.MPCConfig(N=N, n_shape_pts=6)
        state, U = synthetic_problem(K_total, N, seed=11)
        if tie:                                  # identical best candidates on different ranks
            J0, _, _ = orc.rollout_vec(cfg, om, orc.MPCState.from_array(state), U)
            U[K_total - 2] = U[int(np.argmin(J0))]
        lo, hi = shard_bounds(K_total, rank, world)

        def local_solver():
            J, traj, _ = orc.rollout_vec(cfg, om, orc.MPCState.from_array(state), U[lo:hi])
            k = int(np.argmin(J))
            rec = np.concatenate([[J[k], lo + k], U[lo + k, 0], traj[k].reshape(-1)])
            return torch.tensor(rec, dtype=torch.float64)

        smpc = ShardedMPC(local_solver=local_solver, rank=rank, world=world, K_total=K_total)
        rec = smpc.step_host().numpy()
        if rank == 0:
            J, traj, _ = orc.rollout_vec(cfg, om, orc.MPCState.from_array(state), U)
            k = int(np.argmin(J))
            want = np.concatenate([[J[k], k], U[k, 0], traj[k].reshape(-1)])
            out_q.put((rec.tolist(), want.tolist()))
        # every rank must hold the same record
        t = torch.tensor(rec); ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(t, ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K_total,tie", [(2, 64, False), (2, 64, True), (3, 50, False)])
def test_sharded_step_equals_global_argmin(world, K_total, tie):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K_total, 8, tie, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    np.testing.assert_array_equal(np.array(got), np.array(want))


def test_ordered_keys_roundtrip_and_order():
    from rovmpc.sharded import ordered_keys, ordered_values, pack_record, select_record, shard_bounds, INT64_MAX
    x = torch.tensor([-np.inf, -3.5, -1e-300, -0.0, 0.0, 1e-300, 2.0, 1e300, np.inf], dtype=torch.float64)
    k = ordered_keys(x)
    assert torch.equal(ordered_values(k), x)
    assert torch.all(k[1:] >= k[:-1]) and k[1] < k[2] < k[3] and k[5] < k[6] < k[7] < k[8]
    assert (k < INT64_MAX).all()
    r1 = torch.tensor([0.5, 7.0, 1.0, 2.0, 3.0, -0.1, -0.2], dtype=torch.float64)
    r2 = torch.tensor([0.5, 3.0, 9.0, 9.0, 9.0, 9.0, 9.0], dtype=torch.float64)
    r0 = torch.tensor([0.7, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float64)
    slots = torch.stack([pack_record(r1, 1, 3), pack_record(r2, 2, 3), pack_record(r0, 0, 3)]).min(dim=0).values
    best = select_record(slots)
    assert best[0] == 0.5 and best[1] == 3.0             # cost tie -> lower global index wins
    assert torch.equal(best, r2)
    assert [shard_bounds(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def test_bench_launches_its_own_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2` is a plain command: the parent starts the two ranks as children (torch.distributed.run),
    relays rank 0's single JSON line and exits with their status.  Protocol-only ranks (gloo, no GPU, fabricated
    records) exercise launcher + rendezvous + pack / all-reduce(min) / select + the relay."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--backend", "gloo", "--protocol-only", "--K", "64", "--N", "5"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["ranks_agree"] is True and d["records_are_the_global_min"] is True
    assert d["config"]["collective_fallback_reason"] is None


def test_bench_parent_reports_a_failed_rank():
    """A rank that dies must surface as a non-zero exit of the plain command, not as a hang or a silent success."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
                        "--backend", "gloo", "--protocol-only", "--K", "64", "--N", "5"],
                       capture_output=True, text=True, timeout=300, env=dict(env, ROVMPC_BENCH_TEST_FAIL_RANK="1"))
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
