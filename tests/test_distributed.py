"""N>1 control path on CPU: world_size-2 gloo.  Each rank plans its own seed block with the CPU oracle standing in
for the device (checker only), then the same reductions bench.py uses combine the results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from reak_amd import dist_utils, scenarios


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, results):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scn = scenarios.make_c2(world_seed=1)
    osc = oracle_lib.OracleScene(scn, fast=True)
    seeds = dist_utils.seeds_for_rank(0, rank, world, 2)
    nodes = edges = 0
    best = float("inf")
    for s in seeds:
        rc, out, _ = osc.rrt_dyn(scn.rrt_params(seed=s, max_vertices=60))
        assert rc == 0
        nodes += int(out.num_vertices) - 1
        edges += int(out.edges_checked)
        best = min(best, float(out.best_cost))
    elapsed = 1.0 + rank  # max over ranks must be world
    red = dist_utils.reduce_results(dist, elapsed, nodes, edges, edges, best + rank, torch.device("cpu"))
    results[rank] = (seeds, nodes, edges, red)
    dist.barrier()
    dist.destroy_process_group()


def test_seed_blocks_are_disjoint():
    seen = set()
    for step in range(3):
        for rank in range(8):
            s = dist_utils.seeds_for_rank(step, rank, 8, 4)
            assert not (seen & set(s))
            seen |= set(s)
    assert min(seen) == 1 and len(seen) == 3 * 8 * 4


def test_two_rank_gloo_reduction(oracle):
    oracle.build()
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    (s0, n0, e0, r0), (s1, n1, e1, r1) = results[0], results[1]
    assert s0 == [1, 2] and s1 == [3, 4]
    assert r0 == r1  # every rank sees the same reduced line
    elapsed, nodes, edges, spec, best = r0
    assert elapsed == 2.0 and nodes == n0 + n1 == 4 * 60 and edges == e0 + e1


def _progress_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r "plans" for 3 + 2 r intervals, adding 100 (r + 1) vertices per interval; its best cost improves at interval 2
    my_intervals = 3 + 2 * rank
    calls, log, nodes, best = 0, [], 0, float("inf")
    while True:
        done = calls >= my_intervals
        if not done:
            nodes += 100 * (rank + 1)
            if calls == 2:
                best = 10.0 - rank
        all_done, n_all, e_all, b_all = dist_utils.progress_reduce(dist, done, nodes, 2 * nodes, best, torch.device("cpu"))
        calls += 1
        log.append((all_done, n_all, e_all, b_all))
        if all_done:
            break
    results[rank] = (calls, log)
    dist.barrier()
    dist.destroy_process_group()


def test_report_interval_reductions_with_ranks_that_finish_at_different_times():
    """progress_reduce (bench.py's per-interval collectives): ranks run different numbers of rounds, so a
    rank that is done keeps taking part until every rank is -- all ranks make the same number of calls (no hang), see the
    same sums / minimum at every interval, and stop together."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_progress_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    (c0, l0), (c1, l1) = results[0], results[1]
    assert c0 == c1 == 6 and l0 == l1          # rank 1 needs 5 intervals of work + the one that finds everybody done
    assert [x[0] for x in l0] == [False] * 5 + [True]
    assert l0[-1][1] == 3 * 100 + 5 * 200 and l0[-1][2] == 2 * l0[-1][1] and l0[-1][3] == 9.0
    assert l0[1][3] == float("inf") and l0[2][3] == 9.0  # the minimum appears at the interval both improved
