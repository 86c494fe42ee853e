"""Multi-GPU sharding of independent planning problems.

The path does not shard within one problem (generate_rrt is a strict recurrence on one tree); independent
seeds / queries shard across ranks with no data-path collective.  The only collectives are the reductions of
the best solution cost (min) and of the vertex / edge counters (sum) -- a few bytes, latency-bound -- per report
interval while the ranks plan (progress_reduce) and once more after the run (reduce_results)."""
import torch


def seeds_for_rank(step_index, rank, world, problems_per_rank, base_seed=1):
    """Disjoint seed blocks: every (step, rank) gets its own `problems_per_rank` consecutive seeds."""
    first = base_seed + (step_index * world + rank) * problems_per_rank
    return [first + i for i in range(problems_per_rank)]


def reduce_results(dist, elapsed, nodes, edges, spec, best_cost, device):
    """max over ranks of the elapsed time, sum of the counters, min of the best solution cost."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed, nodes, edges, spec, best_cost
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([nodes, edges, spec], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    b = torch.tensor([best_cost], dtype=torch.float64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.MIN)
    n, e, s = (int(v) for v in c.tolist())
    return float(t.item()), n, e, s, float(b.item())


def progress_reduce(dist, done, nodes, edges, best_cost, device):
    """The report-interval reduction of SURVEY.md 8(e) (what a timing / least-cost reporter prints per interval,
    basic_sbmp_reporters.hpp:318-352, over all ranks): sum of {vertices, edges checked}, min of the best solution cost,
    and whether EVERY rank is done.  Ranks run different numbers of rounds (different seeds); a rank that is done keeps
    taking part at the interval of the others until all are, so every rank makes the same number of calls.
    Returns (all_done, nodes, edges, best)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(done), nodes, edges, best_cost
    c = torch.tensor([nodes, edges], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    b = torch.tensor([best_cost, 1.0 if done else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(b, op=dist.ReduceOp.MIN)
    n, e = (int(v) for v in c.tolist())
    bc, dn = (float(v) for v in b.tolist())
    return dn > 0.5, n, e, bc
