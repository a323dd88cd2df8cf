"""N>1 path on CPU: two processes (gloo) each type their shard with the host-side reduction on recorded hit tables and
rank 0 gathers the TSV rows; the result must equal the single-process order."""

import os
import socket
import sys
from pathlib import Path

import torch.multiprocessing as mp

from kaptive_amd.shard import shard_bounds

ROOT = Path(__file__).resolve().parent.parent


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _rows_for(names):
    from kaptive_amd.core.pairwise import PairwiseAlignments
    from kaptive_amd.serotyping.core import Serotyper
    from kaptive_amd.serotyping.io import KaptiveRow
    from oracle import oracle as O
    from tests.golden_util import hits_to_alignments, load_case, load_db

    def proteins(q, t):
        return PairwiseAlignments.from_table(O.protein_align(q.seqs, q.offsets, q.lengths, t.seqs, t.offsets, t.lengths))

    rows = []
    for name in names:
        key, genome, hits, *_ = load_case(name)
        db = load_db(key)
        typer = Serotyper(db, aligner=lambda g, d=db, h=hits: hits_to_alignments(d, g, h), protein_aligner=proteins)
        rows.append(bytes(KaptiveRow.from_result(typer(genome))))
    return rows


CASES = ["k_plain1", "k_split", "k_nolocus", "o_extra2", "k_is"]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    from kaptive_amd.shard import gather_rows, shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = gather_rows(_rows_for(shard(CASES, rank, world)))
    if rank == 0:
        q.put(rows)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_typing_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rows == _rows_for(CASES)


def _bench_partition_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench._load_dbs("kpsc")
    # bench.py's own partition: every rank generates `assemblies` of its own from rank_seed0
    ids, packed = bench.build_workload(3, bench.rank_seed0(rank, 3), 60_000.0, workers=1)
    mine = [(i, int(p.words.sum())) for i, p in zip(ids, packed)]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        q.put(everyone)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_ranks_hold_disjoint_consecutive_assemblies():
    """bench.py --gpus N: rank r types assemblies seeded rank_seed0(r, A) .. + A; the ranks of a 2-process gloo job hold
    what one process generating all 2 * A holds, split in rank order, with nothing shared."""
    import bench

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_partition_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    per_rank = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bench._load_dbs("kpsc")
    ids, packed = bench.build_workload(6, bench.rank_seed0(0, 3), 60_000.0, workers=1)
    whole = [(i, int(p.words.sum())) for i, p in zip(ids, packed)]
    assert per_rank[0] + per_rank[1] == whole
    assert len({i for i, _ in whole}) == 6


def test_numa_affinity_plan_partitions_the_granted_cpus():
    """kaptive_amd/affinity.py: eight devices on a two-socket node (four per socket) get disjoint shares of the granted CPUs,
    each on its device's node; CPUs of a node without a device are dealt out; devices of unknown node share what is left; a
    one-node box keeps its mask; fewer CPUs than devices means everybody shares."""
    from kaptive_amd import affinity as A

    cpu_node = {c: (0 if c < 64 else 1) for c in range(128)}
    granted = list(range(8, 120))  # a container that was given 112 of the 128
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    shares = A.plan(granted, nodes, cpu_node)
    assert len(shares) == 8 and sorted(c for s in shares for c in s) == granted  # a partition
    assert all(len(set(a) & set(b)) == 0 for i, a in enumerate(shares) for b in shares[i + 1 :])
    for s, node in zip(shares, nodes):
        assert s and all(cpu_node[c] == node for c in s)
    assert [len(s) for s in shares] == [14] * 8
    # a third node with CPUs but no device: its CPUs are dealt out, still a partition
    cpu3 = {**cpu_node, **{c: 2 for c in range(128, 144)}}
    shares = A.plan(list(range(144)), nodes, cpu3)
    assert sorted(c for s in shares for c in s) == list(range(144)) and all(len(s) == 18 for s in shares)
    # unknown nodes: even shares of what nobody claimed
    shares = A.plan(list(range(16)), [None, None], {})
    assert shares == [list(range(8)), list(range(8, 16))]
    shares = A.plan(list(range(128)), [0, None, 1], cpu_node)
    assert sorted(c for s in shares for c in s) == list(range(128)) and all(shares)
    # one node, one device: the mask itself; fewer CPUs than devices: shared
    assert A.plan([3, 4, 5], [0], {3: 0, 4: 0, 5: 0}) == [[3, 4, 5]]
    assert A.plan([0, 1], [0, 0, 1, 1], {0: 0, 1: 1}) == [[0, 1]] * 4
    assert A._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    info = A.place(0)  # one device: the plan is the whole mask, nothing is asked or changed
    assert info["applied"] is False
    assert A.device_numa_nodes([0, 1]) == [None, None]  # (no device in the build container: unknown, not an error)
