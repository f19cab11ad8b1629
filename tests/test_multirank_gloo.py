"""N > 1 path on CPU: world_size-2 (and 4) gloo jobs through the same host layer and transport
interface the RCCL build uses.  Operator results do not depend on how the domain is cut into
ranks, so every rank must print the single-rank reference golden numbers."""
import json
import os
import socket
import subprocess
import sys

import pytest

from hpgmg_testlib import ROOT, build_oracle, load_golden

GOLD = load_golden("fcycle_norms.json")


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run_job(world, variant, log2, per_rank, backend="oracle", gather_dim=0, extra_env=None):
    """gather_dim=0: the reference's rank map on every level; > 0: levels of <= gather_dim^3 cells on rank 0 (the product default is 64)."""
    if backend == "oracle":
        build_oracle()
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1", HPGMG_GATHER_DIM=str(gather_dim), HPGMG_GRAPH="1")   # graphs on: capture / replay / empty-capture paths get exercised
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "multirank_worker.py"), variant, str(log2), str(per_rank), backend]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    dec, res, pos = json.JSONDecoder(), [], 0       # ranks share stdout: two records can land on one line
    while True:
        pos = out.stdout.find("RESULT ", pos)
        if pos < 0:
            break
        obj, end = dec.raw_decode(out.stdout, pos + len("RESULT "))
        res.append(obj); pos = end
    assert len(res) == world
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key", [
    (2, "7pt-cheby", 4, 4, "7pt-cheby 4 8"),          # 2 ranks x 4 boxes of 16^3 = the single-rank `4 8` domain (32^3)
    (2, "7pt-gsrb", 4, 4, "7pt-gsrb 4 8"),
    (2, "7pt-cheby-helm", 4, 4, "7pt-cheby-helm 4 8"),
    (4, "7pt-cheby", 4, 2, "7pt-cheby 4 8"),          # 4 ranks x 2 boxes
    (4, "7pt-cheby", 4, 8, "7pt-cheby 4 27"),         # bench.py --gpus 4 in small: 4 x 8 requested -> 3^3 = 27 boxes, 7/7/7/6 per rank, 48^3
])
def test_multirank_matches_single_rank_reference(world, variant, log2, per_rank, gold_key):
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank)
    # rank 0 owns a box on every level and is the rank whose numbers the reference prints
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:   # every rank sees the same reduced norm on the levels where it is active (h and 2h here)
        assert r["norms"][:2] == gold["norms"][:2], r
        assert r["err"] == gold["richardson_error"]
    # the fine level is really distributed, halo messages really flowed, coarse levels collapse onto rank 0
    total = int(gold_key.split()[-1])
    assert sum(r["levels"][0]["my_boxes"] for r in res) == total and all(r["levels"][0]["my_boxes"] >= total // world for r in res)
    assert all(r["stats"]["messages"] > 50 and r["stats"]["allreduces"] > 5 for r in res)
    last = [r["levels"][-1] for r in res]
    assert last[0]["my_boxes"] == 1 and all(l["my_boxes"] == 0 and l["active"] == 0 for l in last[1:])


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key,gather", [
    (2, "7pt-cheby-helm", 4, 4, "7pt-cheby-helm 4 8", 16),     # 32^3 distributed, 16^3 and below gathered on rank 0
    (4, "7pt-cheby", 4, 8, "7pt-cheby 4 27", 24),              # 48^3 over 4 ranks (7/7/7/6), 24^3 and below on rank 0
    (4, "7pt-cheby", 4, 2, "7pt-cheby 4 8", 64),               # the default value: every coarse level of this small problem is on rank 0
])
def test_gathered_coarse_levels(world, variant, log2, per_rank, gold_key, gather):
    """The MI355X rank map (levels <= gather^3 owned by rank 0, hpgmg_set_gather_dim) changes who owns a box,
    never a number: rank 0 prints the reference's golden values."""
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, gather_dim=gather)
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for lv in range(1, len(res[0]["levels"])):          # the fine level is created by the caller with the reference's map
        owners = [r["levels"][lv]["my_boxes"] for r in res]
        if res[0]["levels"][lv]["dim"] <= gather:
            assert all(n == 0 for n in owners[1:]) and owners[0] > 0, (lv, owners)
    assert all(r["levels"][0]["my_boxes"] > 0 for r in res)


def test_transport_selftest_names_the_rank_pair_of_a_damaged_message():
    """bench.py's first contact (hpgmg_transport_selftest, host/level.c): every multi-rank test above runs it and needs 0; here rank 1's messages
    arrive with one wrong element, and every receiver must say which pair failed."""
    res = run_job(3, "7pt-cheby", 4, 1, extra_env={"HPGMG_TEST_CORRUPT_FROM": "1"})
    assert res[1]["stats"]["selftest"] == -2 and "another rank" in res[1]["stats"]["selftest_message"], res[1]      # its own messages were intact; it learns of the others' through the first reduction
    for r in (res[0], res[2]):
        assert r["stats"]["selftest"] == -1 and ("from rank 1 to rank %d" % r["rank"]) in r["stats"]["selftest_message"], r


def test_bench_supervisor_starts_fresh_ranks_with_the_other_transport_when_the_first_fails():
    """bench.py --gpus 2 on a machine without a GPU: both attempts must fail, but in the right way -- the supervisors (torch.distributed.run workers that
    never touch the GPU) notice the failed rccl ranks, end the attempt, start FRESH rank processes with the ipc transport, and report that no attempt
    produced a result (exit code 1, no JSON line).  The GPU suite runs the same flow to a result (tests/test_gpu_multirank.py)."""
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--watchdog", "60"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
    # (which rank is named depends on timing: a rank still starting up when its peer fails is ended by its supervisor and not listed)
    import re
    assert re.search(r"rccl attempt: rank \d exited with code", out.stderr) and "starting fresh rank processes with the ipc transport" in out.stderr, out.stderr[-2000:]
    assert re.search(r"ipc attempt: rank \d exited with code", out.stderr) and "no attempt produced a result" in out.stderr, out.stderr[-2000:]


def test_subset_reductions_are_one_collective_on_a_sub_communicator():
    """The reference reduces on a per-level communicator of the ranks that own boxes there (MPI_Comm_split, mg.c:985-993).  MGBuild announces every such set that is
    a proper subset of the job on EVERY rank (hpgmg_transport.prepare_subset), the transport builds a sub-communicator -- collective over the whole job, like
    ncclCommSplit in kernels/comm_rccl.hip; here torch.distributed.new_group -- and a reduction over the set is then ONE collective on it, partials added in member
    order.  (On one node with the reference's Z-Morton map such sets have one member unless the rank count exceeds the boxes of an agglomerated level, so the set
    {0, 1} of the transport self-test stands in: HPGMG_SELFTEST_SUBCOMM=1 announces it like a level's.)  Same golden numbers either way."""
    gold = GOLD["7pt-cheby 4 27"]
    res = run_job(4, "7pt-cheby", 4, 8, gather_dim=0, extra_env={"HPGMG_SELFTEST_SUBCOMM": "1"})
    assert res[0]["norms"] == gold["norms"] and res[0]["err"] == gold["richardson_error"], res[0]
    announced = [r["stats"].get("subsets_announced", 0) for r in res]
    assert announced[0] > 0 and len(set(announced)) == 1, announced          # every rank was told about the set, members or not
    assert [r["stats"].get("subset_collectives", 0) for r in res] == [1, 1, 0, 0] and all(r["stats"].get("subset_alltoalls", 0) == 0 for r in res), [r["stats"] for r in res]
    off = run_job(4, "7pt-cheby", 4, 8, gather_dim=0)                        # not announced: the all-to-all among the members, same sum
    assert off[0]["norms"] == gold["norms"] and off[0]["err"] == gold["richardson_error"]
    assert [r["stats"].get("subset_alltoalls", 0) for r in off] == [1, 1, 0, 0] and all(r["stats"].get("subset_collectives", 0) == 0 for r in off)
