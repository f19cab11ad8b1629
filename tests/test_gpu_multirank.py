"""N > 1 on the HIP path: two (and four) processes share the box's single MI355X, each owning part of the
boxes, with the message buffers in device memory.  The transport here stages those buffers through the host
and moves them with gloo (RCCL refuses two ranks on one device); everything else -- pack / unpack kernels,
ghost-free stencils reading remote faces from the ghost zone, active-rank sets on coarse levels, host-driven
BiCGStab with all-reduced dot products -- is the code the RCCL transport drives on a multi-GPU node.
Results must equal the single-rank reference golden numbers."""
import json
import os
import subprocess
import sys

import pytest

from hpgmg_testlib import ROOT, load_golden
from test_multirank_gloo import run_job

pytestmark = pytest.mark.gpu
GOLD = load_golden("fcycle_norms.json")


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key", [
    (2, "7pt-cheby-helm", 4, 4, "7pt-cheby-helm 4 8"),
    (2, "7pt-gsrb", 4, 4, "7pt-gsrb 4 8"),
    (4, "7pt-cheby", 4, 2, "7pt-cheby 4 8"),
    (2, "7pt-cheby-helm", 7, 4, "7pt-cheby-helm 7 8"),    # bench.py --gpus 2 at full size: 2 x 4 boxes of 128^3 (wide kernel + shell launches)
    (4, "7pt-cheby", 4, 8, "7pt-cheby 4 27"),          # bench.py --gpus 4 in small: 27 boxes over 4 ranks (7/7/7/6), 48^3
    (2, "27pt-cheby", 4, 4, "27pt-cheby 4 8"),
    (2, "fv4-gsrb", 4, 4, "fv4-gsrb 4 8"),
    (2, "27pt-gsrb", 7, 4, "27pt-gsrb 7 8"),           # boxes of 128^3 with a remote k face: tiled kernels on exchanged ghost zones, boundary-condition
    (2, "fv4-gsrb", 7, 4, "fv4-gsrb 7 8"),             # entries that read local neighbours directly and remote ones from the ghost zone; gathered levels on rank 0
])
def test_hip_multirank_matches_single_rank_reference(world, variant, log2, per_rank, gold_key):
    if gold_key not in GOLD:
        pytest.skip("no golden record for " + gold_key)
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, backend="hip")
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:
        assert r["norms"][:2] == gold["norms"][:2], r
    total = int(gold_key.split()[-1])
    assert sum(r["levels"][0]["my_boxes"] for r in res) == total
    assert all(r["stats"]["messages"] > 50 for r in res)
    if variant.startswith("7pt") and os.environ.get("HPGMG_OVERLAP", "1") != "0":     # the 7-point path overlaps its halo exchanges with the stencil launches
        assert all(r["stats"]["overlapped_exchanges"] > 20 for r in res), [r["stats"] for r in res]


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key", [
    (2, "7pt-cheby-helm", 7, 4, "7pt-cheby-helm 7 8"),     # bench.py --gpus 2: bricks of 2 x 2 x 1 boxes, one remote k face
    (4, "7pt-cheby-helm", 7, 2, "7pt-cheby-helm 7 8"),     # bench.py --gpus 4: bricks of 2 x 1 x 1 boxes, remote j and k faces + the edge between them
    # (bench.py --gpus 8 -- one box per rank, three remote faces and three edges each -- runs in test_ipc_peer_copy_transport, on device-ordered messages, with the same assertions)
    (2, "7pt-gsrb", 7, 4, "7pt-gsrb 7 8"),
    (4, "7pt-cheby", 7, 2, "7pt-cheby 7 8"),
    # SURVEY 8(e)'s strong series `6 64/N`: the same 256^3 in boxes of 64^3 -- two boxes per 128-cell row of the pair kernel (its NARROW form) AND faces on other ranks
    (2, "7pt-cheby-helm", 6, 32, "7pt-cheby-helm 7 8"),
    (4, "7pt-gsrb", 6, 16, "7pt-gsrb 7 8"),
])
def test_sweep_pairs_across_rank_boundaries(world, variant, log2, per_rank, gold_key):
    """north_star's strong-scaling series (256^3 on 2 / 4 / 8 ranks): the fine-level smoother must stay the two-sweeps-per-pass
    kernel when faces belong to other ranks -- x0 exchanged two cells deep ONCE per sweep pair (chebyshev.c:45-46 exchanges once per
    sweep), x1 on the ghost layer recomputed locally -- and still reproduce the single-rank reference numbers to the last digit."""
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, backend="hip")
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:
        # 4 timed f-cycles at h (three_sizes, richardson, 3 repeats - 1) x 2 smooth() calls on the fine level, at least
        assert r["stats"]["pair_remote_smooths"] >= 8, r["stats"]
        assert r["stats"]["pair_remote_launches"] == 2 * r["stats"]["pair_remote_smooths"], r["stats"]
        if os.environ.get("HPGMG_OVERLAP", "1") != "0":       # each pair's halo exchange runs on the exchange stream under the workgroups that touch no remote face
            assert r["stats"]["overlapped_exchanges"] >= 2 * r["stats"]["pair_remote_smooths"], r["stats"]
        if variant.startswith("7pt-cheby") and log2 == 7:      # the up leg's interpolation_vcycle is folded into the first pair across rank boundaries too (each owner adds the parents to what it sends; whole-row boxes only)
            assert r["stats"]["interp_folded_remote"] >= 4, r["stats"]
        # residual + restriction and residual + norm stay ONE pass each on a bandwidth-bound level with faces on other ranks (x crosses them first)
        if log2 == 7 and per_rank * 128 ** 3 >= 4000000:
            assert r["stats"]["fused_residuals_remote"] >= 8, r["stats"]


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key", [
    (2, "fv4-gsrb", 7, 4, "fv4-gsrb 7 8"),        # bricks of 2 x 2 x 1 boxes of 128^3: one remote k face
    (2, "27pt-gsrb", 7, 4, "27pt-gsrb 7 8"),
    (4, "fv4-gsrb", 7, 2, "fv4-gsrb 7 8"),        # bricks of 2 x 1 x 1: remote j and k faces and the edge between them
    # (one box per rank -- three remote faces, three edges, a corner: `(8, 27pt-gsrb, 7, 1)` -- and
    # (BASELINE config 3 as stated -- 512^3, eight ranks of 2 x 2 x 2 boxes, `(8, fv4-gsrb, 7, 8)` -- run in test_ipc_peer_copy_transport below, on device-ordered messages)
    (2, "fv4-cheby", 5, 4, "fv4-cheby 5 8"),      # the tiled kernels (Chebyshev sweeps, residual) on the images, reference rank map down to boxes of 8^3
    (2, "27pt-cheby", 7, 4, "27pt-cheby 7 8"),
])
def test_config3_kernels_across_rank_boundaries(world, variant, log2, per_rank, gold_key):
    """BASELINE config 3 is an 8-rank configuration: the one-pass red + black kernels of the fv4 and 27-point GSRB smoothers, the LDS-tiled
    kernels and the fused residual passes must keep running when faces belong to other ranks -- on images of the neighbouring ranks' boxes,
    refreshed by ONE message per neighbour and sweep (gsrb.c:30-33 exchanges twice per sweep) -- and reproduce the single-rank reference
    numbers to the last digit."""
    if gold_key not in GOLD:
        pytest.skip("no golden record for " + gold_key)
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, backend="hip")
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:
        assert r["stats"]["image_exchanges"] > 20, r["stats"]
        if variant == "fv4-gsrb":
            assert r["stats"]["fv4_rb_smooths"] >= 8, r["stats"]
        if variant == "27pt-gsrb":
            assert r["stats"]["rb27_passes"] >= 16, r["stats"]
        if variant.endswith("gsrb") and os.environ.get("HPGMG_OVERLAP", "1") != "0":     # the refresh of a red + black pass runs under the tiles that read no image
            assert r["stats"]["overlapped_exchanges"] > 20, r["stats"]


@pytest.mark.parametrize("variant,gold_key,env", [
    ("fv4-gsrb", "fv4-gsrb 7 8", {"HPGMG_OVERLAP": "0"}),       # the refresh in line with the launch stream: one whole launch per pass
    ("27pt-gsrb", "27pt-gsrb 7 8", {"HPGMG_OVERLAP": "0"}),
    ("fv4-gsrb", "fv4-gsrb 7 8", {"HPGMG_IMAGES": "0"}),        # without the images: exchange_boundary + two half sweeps, what round 3 ran
])
def test_config3_multirank_switches(variant, gold_key, env):
    gold = GOLD[gold_key]
    res = run_job(2, variant, 7, 4, backend="hip", extra_env=env)
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:
        if env.get("HPGMG_IMAGES") == "0":
            assert r["stats"]["image_exchanges"] == 0 and r["stats"]["fv4_rb_smooths"] == 0, r["stats"]
        else:
            assert r["stats"]["image_exchanges"] > 20 and r["stats"]["overlapped_exchanges"] == 0, r["stats"]


def test_7pt_multirank_without_overlap():
    """HPGMG_OVERLAP=0 on the 7-point path: the sweep pairs' two-deep exchange and the exchange in front of the fused residual passes run in line on the
    launch stream (whole launches instead of two parts) -- the same numbers, and no exchange may have been counted as overlapped."""
    gold = GOLD["7pt-cheby-helm 7 8"]
    res = run_job(2, "7pt-cheby-helm", 7, 4, backend="hip", extra_env={"HPGMG_OVERLAP": "0"})
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    for r in res:
        assert r["stats"]["pair_remote_smooths"] >= 8 and r["stats"]["fused_residuals_remote"] >= 8 and r["stats"]["overlapped_exchanges"] == 0, r["stats"]


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key,gather", [
    (2, "7pt-cheby-helm", 4, 4, "7pt-cheby-helm 4 8", 16),
    (4, "7pt-cheby", 4, 8, "7pt-cheby 4 27", 24),
    (2, "fv4-gsrb", 4, 4, "fv4-gsrb 4 8", 16),
    (4, "7pt-cheby-helm", 4, 2, "7pt-cheby-helm 4 8", 64),     # product default: hipGraph segments on rank 0, empty segments elsewhere
    (2, "7pt-gsrb", 5, 4, "7pt-gsrb 5 8", 64),
])
def test_hip_gathered_coarse_levels(world, variant, log2, per_rank, gold_key, gather):
    """Default product rank map (coarse levels gathered on rank 0, where the fused tail kernel runs them) on the HIP path."""
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, backend="hip", gather_dim=gather)
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]


@pytest.mark.parametrize("world,variant,log2,per_rank,gold_key,gather", [
    (2, "7pt-cheby-helm", 7, 4, "7pt-cheby-helm 7 8", 64),     # sweep pairs with a two-deep halo + overlapped shell launches, product rank map
    (4, "7pt-cheby", 4, 8, "7pt-cheby 4 27", 0),               # 27 boxes over 4 ranks (7/7/7/6): uneven message plans, reference rank map on every level
    (8, "7pt-cheby-helm", 7, 1, "7pt-cheby-helm 7 8", 64),     # one box per rank: three remote faces and their edges
    (2, "7pt-gsrb", 5, 4, "7pt-gsrb 5 8", 0),
    (2, "fv4-gsrb", 7, 4, "fv4-gsrb 7 8", 64),                 # images of the neighbouring boxes, refreshed on the exchange stream
    (8, "27pt-gsrb", 7, 1, "27pt-gsrb 7 8", 64),
    (8, "fv4-gsrb", 7, 8, "fv4-gsrb 7 64", 64),                # BASELINE config 3 as stated
])
def test_ipc_peer_copy_transport(world, variant, log2, per_rank, gold_key, gather):
    """The node-local transport of the product (kernels/comm_ipc.hip): every message is a device-to-device copy from the sender's buffer into the
    receiver's (hipIpc memory handles), ordered against both ranks' launch streams by stream-ordered host functions on shared counters -- nothing
    staged through the host, no stream synchronisation -- and every scalar reduction goes through shared memory in rank order.  Unlike RCCL it runs with several ranks on ONE
    GPU, so here the multi-rank product path runs end to end on device-ordered messages; the numbers must be the single-rank reference's."""
    gold = GOLD[gold_key]
    res = run_job(world, variant, log2, per_rank, backend="hip", gather_dim=gather, extra_env={"HPGMG_TEST_TRANSPORT": "ipc"})
    assert res[0]["norms"] == gold["norms"], res[0]
    assert res[0]["err"] == gold["richardson_error"] and res[0]["order"] == gold["order"]
    assert res[0]["repeat"] == [gold["norms"][0]] * 3, res[0]["repeat"]
    for r in res:
        assert r["stats"].get("transport") == "ipc" and r["stats"]["messages"] > 50, r["stats"]
        assert r["stats"]["overlapped_exchanges"] > 0, r["stats"]
        if variant == "fv4-gsrb":            # the one-pass red + black kernel on the images, every rank
            assert r["stats"]["fv4_rb_smooths"] >= 8 and r["stats"]["image_exchanges"] > 20, r["stats"]
        if variant == "27pt-gsrb":
            assert r["stats"]["rb27_passes"] >= 16 and r["stats"]["image_exchanges"] > 20, r["stats"]
        if variant.startswith("7pt-cheby") and log2 == 7:       # the fine-level smoother stays the sweep-pair kernel across rank boundaries, one exchange per pair
            assert r["stats"]["pair_remote_smooths"] >= 8 and r["stats"]["pair_remote_launches"] == 2 * r["stats"]["pair_remote_smooths"], r["stats"]
            assert r["stats"]["interp_folded_remote"] >= 4, r["stats"]


def bench_line(argv, extra_env=None, expect_code=0):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == expect_code, (out.returncode, out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout[-1000:], out.stderr[-2000:])
    return json.loads(lines[0]), out.stderr


def test_bench_two_ranks_end_to_end_on_the_ipc_transport():
    """`bench.py --gpus 2` as the driver will call it, on the one GPU of this box: supervisors -> rank processes -> transport self-test -> the strong-scaling
    problem (2 x 4 boxes of 128^3) -> ONE JSON line whose norm is the reference's string."""
    d, _ = bench_line(["--gpus", "2", "--transport", "ipc", "--share-gpu", "--steps", "3", "--warmup", "2"])
    assert d["n_gpus"] == 2 and d["config"]["transport"].startswith("ipc") and "fallback" not in d["config"]["transport"], d["config"]
    assert d["config"]["transport_selftest"] == "passed" and d["config"]["parity_ok"] is True, d["config"]
    assert "%1.15e" % d["config"]["fcycle_residual_norm"] == GOLD["7pt-cheby-helm 7 8"]["norms"][0]
    assert d["config"]["halo"]["smooths_as_sweep_pairs_with_remote_faces"] > 0 and d["roofline"]["frac"] < 1


@pytest.mark.parametrize("how", ["rccl", "rccl:hang"])
def test_bench_falls_back_to_fresh_ipc_ranks_when_the_first_transport_fails(how):
    """The first transport is made to fail (every rank of the rccl attempt exits 97) or to hang (every rank sleeps until its watchdog): the supervisors end the
    attempt and start FRESH rank processes with the ipc transport; the line of that attempt says what happened."""
    d, err = bench_line(["--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--watchdog", "25" if how.endswith("hang") else "300"],
                        extra_env={"HPGMG_TEST_FAIL_TRANSPORT": how})
    assert d["config"]["transport"].startswith("ipc (fallback: rccl attempt: "), d["config"]["transport"]
    assert ("exited with code 124" if how.endswith("hang") else "exited with code 97") in d["config"]["transport"], d["config"]["transport"]
    assert d["config"]["rccl_ranks"] == 0 and d["config"]["parity_ok"] is True and d["n_gpus"] == 2
    assert "starting fresh rank processes with the ipc transport" in err
