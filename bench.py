#!/usr/bin/env python3
"""bench.py -- the headline measurement: DOF/s of one HPGMG-FV FMG F-cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: zero_vector(U) +
one FMG F-cycle (reference bench_hpgmg, finite-volume/source/hpgmg-fv.c:50-99) on
BASELINE.json config 2: `hpgmg-fv 7 8`, fp64 7-point variable-coefficient Helmholtz,
Chebyshev smoother, 8 boxes of 128^3 per GPU (256^3 on one GPU).  Inputs (beta, alpha, F) are
the reference's analytic problem (problem.p6.c), resident in HBM before timing starts.

N > 1: one process per GPU, boxes partitioned over the ranks exactly like the reference
partitions them over MPI ranks (Z-Morton), ghost zones exchanged with RCCL send/recv over xGMI.
The torch.distributed.run workers (started by the driver, or by bench.py itself when it is called
bare) are SUPERVISORS that never touch the GPU: each starts its rank as a child process, with a time
limit per attempt; if the RCCL attempt fails or hangs on any rank, every child is ended and fresh
children run the node-local hipIpc transport (`"transport": "ipc (fallback: <reason>)"`).  Every
attempt begins with a transport self-test (a known pattern to and from every rank, a maximum, a
rank-ordered sum), and the line carries `parity_ok`: the F-cycle norm equals the reference's string.
One GPU, default workload: the line also carries `also` -- configs 3 (fv4, 27-pt), 4 and 1 with
3 timed solves each, under the same clock, each in a fresh child process (a failure there cannot
take the headline measurement with it; the headline line is also written to stderr first).
Default series = north_star's STRONG scaling: the same 256^3 problem on 1, 2, 4, 8 GPUs
(`hpgmg-fv 7 8/N`, i.e. 8/4/2/1 boxes of 128^3 per GPU; "scaling": "strong").  --series weak =
the reference CLI's `7 8` with N ranks (256^3, 256^3, 384^3, 512^3).
value = fine-grid DOF of the whole job / max-over-ranks time.

Extra objects on the JSON line:
  roofline     (frac = the fused-form reading, frac_survey_8d = SURVEY 8(d)'s bytes per sweep; `basis` says which is which)
               the fine-level smoother kernel, timed with hipEvents on the launch stream inside the timed region (every 5th launch: an
               event pair idles the GPU ~10 us, launches_timed says how many were timed).
               achieved / frac = the bytes ONE LAUNCH of that kernel has to move in the form it really has (a kernel that does two
               sweeps per pass is charged its own stream count once, not twice the single-sweep figure) / launch time / 8 TB/s;
               unfused_equivalent_GBs = SURVEY 8(d)'s bytes per cell per sweep x the sweeps the launch performs / launch time: what
               separate sweeps would have had to move in that time (a speed-up figure, not a bandwidth -- it may exceed the peak);
               traffic = HBM bytes per launch from the rocprofv3 PMC passes (traffic_source names the committed summary file),
               dram_GBs_from_pmc = traffic / launch time.  frac > 1 is refused.
  cpu_baseline the REFERENCE binary (oracle/_ref, built from /root/reference by oracle/Makefile)
               run on this box's host cores; falls back to the CPU restatement ("port").
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOG2_BOX_DIM, BOXES_PER_RANK = 7, 8


def host_cores():
    """CPU threads this container may really use (cgroup quota, not the 256 the box reports)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def cpu_baseline():
    """Reference (or port) on the host cores, bounded: the benchmark's own 10+10 solves at 256^3, 128^3, 64^3."""
    cores = host_cores()
    # the CPU code at its best on these cores: one thread per core of the quota, threads that spin between the many small parallel
    # regions of the coarse levels (measured on the GPU box: 1.48e8 DOF/s; with OMP_WAIT_POLICY=passive 1.21e8)
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_WAIT_POLICY", None)
    ref = os.path.join(ROOT, "oracle", "_ref", "hpgmg-7pt-cheby-helm")
    port = os.path.join(ROOT, "oracle", "hpgmg-fv-oracle")
    if os.path.exists(ref):
        cmd, kind, sample = [ref, "7", "8"], "reference", "reference binary `hpgmg-fv 7 8` (7pt VC Helmholtz, Chebyshev): its own protocol, 10 warm-up + 10 timed F-cycles at 256^3 (and 128^3, 64^3)"
    elif os.path.exists(port):
        cmd, kind, sample = [port, "--helmholtz", "--warmup", "2", "--solves", "5", "7", "8"], "port", "CPU restatement `--helmholtz 7 8`: 2 warm-up + 5 timed F-cycles at 256^3 (and 128^3, 64^3)"
    else:
        return None
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900).stdout
        m = re.search(r"h=\S+\s+DOF=\S+\s+time=(\S+)\s+DOF/s=(\S+)", out)
        return {"value": float(m.group(2)), "unit": "DOF/s", "cores": cores, "kind": kind, "sample": sample,
                "seconds_per_solve": float(m.group(1))}
    except Exception as exc:  # report, never fake
        return {"value": None, "unit": "DOF/s", "cores": cores, "kind": kind, "sample": sample, "error": repr(exc)}


def pmc_traffic(workload="config2", fine_cells=None):
    """(HBM bytes per fine-level smoother launch, file it comes from): the committed rocprofv3 --pmc summary of this workload, if any (the
    newest: files sort by round tag).  config 2: the sweep-pair launch with its pre-pass; config 3: the one-pass kernel at the same size.
    The number was measured on ANOTHER run of the same command (profiles/README.md); it is not a measurement of this run."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    want = {"config2": ("_pmc_summary.json", "hbm_bytes_per_launch_cheby_fine"), "config3-fv4": ("_fv4_pmc_summary.json", "hbm_bytes_per_launch_smoother_fine"),
            "config3-27pt": ("_27pt_pmc_summary.json", "hbm_bytes_per_launch_smoother_fine")}.get(workload)
    if want and os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if not f.endswith(want[0]) or (workload == "config2" and ("_fv4_" in f or "_27pt_" in f)):
                continue
            try:
                d = json.load(open(os.path.join(pdir, f)))
                got = d.get(want[1])
                if got and (fine_cells is None or workload == "config2" or d.get("cells") == fine_cells):
                    best = (got, "profiles/" + f)
            except Exception:
                pass
    return best if best else (None, None)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=list(WORKLOADS), default="config2",
                    help="BASELINE.json configs; default config2 = the one the metric is quoted on.  config3 (`7 64`) and config4 (`8 8`) are the 8-GPU "
                         "configurations: with --gpus 1 they run their single-rank reading (512^3 on one GPU).  config5 = config2 with --precision fp32")
    ap.add_argument("--no-also", action="store_true",
                    help="one GPU, config2: do NOT append the short runs of the other BASELINE configurations (`also` on the JSON line: config3-fv4, config3-27pt, "
                         "config4, config1, 3 timed solves each).  The profiling scripts pass this so that a kernel trace holds config 2 only")
    ap.add_argument("--transport", choices=["rccl", "ipc"], default="rccl",
                    help="N > 1: rccl = grouped ncclSend / ncclRecv over xGMI (default, what the driver measures); ipc = the node-local peer-copy transport "
                         "(hipIpc handles, stream-ordered host functions: hpgmg_amd/csrc/kernels/comm_ipc.hip).  If the rccl job fails or hangs, a FRESH set of rank "
                         "processes is started with ipc and the line says so (--no-fallback: do not)")
    ap.add_argument("--no-fallback", action="store_true", help="N > 1: report the failure of the chosen transport instead of trying the other one")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank on device 0 (several ranks on ONE GPU; only the ipc transport can: RCCL refuses two ranks on one device, so the rccl attempt fails and the "
                         "ipc fallback runs).  Exercises and times the N > 1 code path where only one GPU exists; the line says so and is NOT a scaling measurement")
    ap.add_argument("--force-transport", action="store_true", help="initialise torch.distributed + the RCCL transport even with one rank (smoke test of the N>1 bootstrap)")
    ap.add_argument("--precision", choices=["fp64", "fp32"], default="fp64",
                    help="fp32 = BASELINE.json config 5: mixed-precision Chebyshev smoother (fp32 coefficient streams), tolerance-gated; default fp64 = config 2, bit-exact")
    ap.add_argument("--series", choices=["strong", "weak"], default="strong",
                    help="what N > 1 runs (config2): strong = the north_star series, the SAME 256^3 problem cut over N GPUs (`hpgmg-fv 7 8/N`: 8/4/2/1 boxes of "
                         "128^3 per GPU); weak = the reference CLI's own `7 8` with N ranks (256^3, 256^3, 384^3, 512^3 for N = 1, 2, 4, 8; hpgmg-fv.c:184-197)")
    ap.add_argument("--route-b", action="store_true",
                    help="measure INTEGRATION.md Route B instead: the REFERENCE's own driver (mg.c, solvers.c, hpgmg-fv.c, level.c -- oracle/_ref/routeb-7pt-cheby-helm, "
                         "built from /root/reference by oracle/Makefile) running on this repository's operator plugin, config 2, its own 10 + 10 solve protocol")
    ap.add_argument("--watchdog", type=int, default=None,
                    help="seconds after which a rank that has not finished reports where it is stuck and exits 124 (0 = off).  Default 900 for one GPU, 300 per "
                         "attempt for N > 1 (two attempts -- rccl, then the ipc fallback -- fit the driver's limit)")
    ap.add_argument("--as-rank", action="store_true", help=argparse.SUPPRESS)      # internal: this process IS one rank of an N > 1 attempt (started by the supervisor below)
    args = ap.parse_args(argv)
    if args.watchdog is None:
        args.watchdog = 300 if args.gpus > 1 else 900
    return args


# (operator, smoother, helmholtz, variable coefficients, log2 box dim, boxes per rank, description, golden key in tests/golden/fcycle_norms.json by fine-grid dim)
def _workloads():
    import hpgmg_amd as H
    return {"config1": (H.OP_7PT, H.SMOOTH_CHEBY, 0, 0, 5, 8, "7-pt constant-coefficient Poisson, Chebyshev"),
            "config2": (H.OP_7PT, H.SMOOTH_CHEBY, 1, 1, 7, 8, "7-pt variable-coefficient Helmholtz, Chebyshev"),
            "config3-fv4": (H.OP_FV4, H.SMOOTH_GSRB, 0, 1, 7, 64, "4th-order fv4 variable-coefficient Poisson, GSRB"),
            "config3-27pt": (H.OP_27PT, H.SMOOTH_GSRB, 0, 0, 7, 64, "27-pt constant-coefficient Poisson, GSRB"),
            "config4": (H.OP_7PT, H.SMOOTH_CHEBY, 1, 1, 8, 8, "7-pt variable-coefficient Helmholtz, Chebyshev"),
            "config5": (H.OP_7PT, H.SMOOTH_CHEBY, 1, 1, 7, 8, "7-pt variable-coefficient Helmholtz, Chebyshev")}


WORKLOADS = ("config1", "config2", "config3-fv4", "config3-27pt", "config4", "config5")
# the reference's own F-cycle residual norm at level h for (workload, fine-grid dim): the operators' results do not depend on how the domain is cut into boxes
# (SURVEY 8c), so a strong-scaling run over N ranks must print the single-rank string.  Keys of tests/golden/fcycle_norms.json (made from oracle/_ref binaries).
GOLDEN_KEY = {("config1", 64): "7ptcc-cheby 5 8", ("config2", 256): "7pt-cheby-helm 7 8", ("config2", 512): "7pt-cheby-helm 8 8", ("config4", 512): "7pt-cheby-helm 8 8",
              ("config3-fv4", 512): "fv4-gsrb 7 64", ("config3-fv4", 256): "fv4-gsrb 7 8", ("config3-27pt", 512): "27pt-gsrb 7 64", ("config3-27pt", 256): "27pt-gsrb 7 8"}
# The fine-level smoother kernel each workload spends most of its time in: (bytes per cell per sweep when every sweep is a pass of its
# own -- SURVEY 8(d) --, bytes per cell ONE LAUNCH moves when it performs two sweeps in one pass, description)
SMOOTHER = {"config1": (40, 40, "7-pt constant-coefficient Chebyshev sweep (stencil7_kernel): x_n, x_nm1, rhs, Dinv read + x_np1 written"),
            "config2": (72, 80, "hpgmg::cheby_pair_kernel<VC Helmholtz> (+ its edge-column pre-pass): one launch = TWO Chebyshev sweeps in one pass: x0, x_nm1, rhs, Dinv, alpha, beta_i/j/k read once, x1 and x2 written"),
            "config4": (72, 80, "hpgmg::cheby_pair_kernel<VC Helmholtz> (+ its edge-column pre-pass): one launch = TWO Chebyshev sweeps in one pass: x0, x_nm1, rhs, Dinv, alpha, beta_i/j/k read once, x1 and x2 written"),
            "config5": (52, 60, "hpgmg::cheby_pair_kernel<VC Helmholtz, fp32 coefficient streams> (+ pre-pass): one launch = TWO Chebyshev sweeps: x0, x_nm1, rhs (fp64), five fp32 coefficient streams read, x1 and x2 written"),
            "config3-fv4": (56, 56, "hpgmg::fv4_rb_kernel<VC Poisson> (+ its pre-pass): one launch = BOTH coloured half sweeps of an out-of-place GSRB sweep in one pass: x, rhs, Dinv, beta_i/j/k read once, x' written (the intermediate vector stays in LDS); "
                                    "with HPGMG_TUNE_FV4_NO_RB=1 hpgmg::fv4_tile_kernel, one half sweep per launch"),
            "config3-27pt": (32, 32, "hpgmg::stencil27_rb_kernel: one launch = BOTH coloured half sweeps of an out-of-place GSRB sweep in one pass: x, rhs, Dinv read once, x' written (the intermediate vector stays in LDS)")}


def golden_norm(workload, dim):
    """The reference's printed f-cycle norm for this workload at this fine-grid size, or None (tests/golden/fcycle_norms.json: data, not the oracle)."""
    key = GOLDEN_KEY.get(("config2" if workload == "config5" else workload, int(dim)))
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "fcycle_norms.json")))[key]["norms"][0] if key else None
    except Exception:
        return None


def route_b():
    # Nothing of this repository's driver runs here: the reference's FMGSolve calls the plugin through operators.h only, so none of the
    # optional fused hooks are used.  The reference times its own solves (hpgmg-fv.c:77-99) and prints DOF/s.
    binary = os.path.join(ROOT, "oracle", "_ref", "routeb-7pt-cheby-helm")
    if not os.path.exists(binary):
        raise SystemExit("bench.py --route-b: oracle/_ref/routeb-7pt-cheby-helm is not built (make -C oracle ref, needs /root/reference)")
    env = dict(os.environ, OMP_NUM_THREADS=str(min(host_cores(), 8)))
    out = subprocess.run([binary, str(LOG2_BOX_DIM), str(BOXES_PER_RANK)], capture_output=True, text=True, env=env, timeout=900)
    m = re.search(r"h=\S+\s+DOF=(\S+)\s+time=(\S+)\s+DOF/s=(\S+)", out.stdout)
    norm = re.search(r"f-cycle\s+norm=(\S+)", out.stdout)
    if out.returncode or not m:
        sys.stderr.write(out.stdout[-2000:] + out.stderr[-2000:])
        raise SystemExit("bench.py --route-b: the reference driver did not finish")
    print(json.dumps({"metric": "DOF/s (fine-grid) for FMG F-cycle", "value": float(m.group(3)), "unit": "DOF/s", "n_gpus": 1, "steps": 10, "warmup": 10,
                      "ms_per_step": float(m.group(2)) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"hpgmg-fv {LOG2_BOX_DIM} {BOXES_PER_RANK}: 256^3 fp64 7-pt variable-coefficient Helmholtz, Chebyshev smoother -- Route B: the reference's "
                                             "unmodified mg.c / solvers.c / hpgmg-fv.c (level.c with the storage lines patched) on this repository's operator plugin",
                                 "route": "B", "fine_grid_dof": float(m.group(1)), "fcycle_residual_norm": float(norm.group(1)) if norm else None,
                                 "protocol": "the reference's own: 10 warm-up + 10 timed FMGSolve, host clock around each (hpgmg-fv.c:77-99)"},
                      "roofline": None}), flush=True)


# ---------------------------------------------------------------------------------------------------------------- N > 1: the supervisor
# `bench.py --gpus N` is ALWAYS N supervisor processes (torch.distributed.run workers: started by the driver, or by the bare call below) that never
# touch the GPU; each starts its rank of an attempt as a CHILD process (`--as-rank`).  So a transport that fails or hangs costs one attempt, not the
# measurement: the children are ended, and FRESH children run the node-local ipc transport (never a re-exec of a process that has initialised the GPU).
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _die_with_parent():
    """preexec_fn of a rank child: SIGKILL when the supervisor that started it dies (prctl PR_SET_PDEATHSIG), so no rank outlives a killed job on the GPU."""
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, 9, 0, 0, 0)      # PR_SET_PDEATHSIG = 1, SIGKILL = 9
    except Exception:
        pass


def _unlink_segment(name):
    try:
        os.unlink("/dev/shm" + name)
    except OSError:
        pass


def supervise(args):
    import signal
    import tempfile
    import torch.distributed as dist      # gloo between the supervisors only: CPU tensors, no device is touched
    running = [None]      # the rank child of the attempt in progress

    def _ended(signum, _frame):      # the launcher (or the driver's time-out) ends this supervisor: its rank goes first
        if running[0] is not None and running[0].poll() is None:
            running[0].kill()
        os._exit(128 + signum)
    signal.signal(signal.SIGTERM, _ended)
    signal.signal(signal.SIGINT, _ended)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    dist.init_process_group(backend="gloo")
    import torch
    attempts = [args.transport] + (["ipc"] if (args.transport == "rccl" and not args.no_fallback) else [])
    passthrough = [a for a in sys.argv[1:] if a not in ("--as-rank",)]
    why, line, code = None, None, 1
    for number, transport in enumerate(attempts):
        box = [(_free_port(), "%d_%d" % (os.getpid(), number)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        port, nonce = box[0]
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "GROUP_RANK", "ROLE_RANK")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=os.environ.get("LOCAL_RANK", str(rank)),
                   HPGMG_BENCH_NONCE=nonce, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = tempfile.NamedTemporaryFile(prefix="hpgmg_bench_rank%d_" % rank, suffix=".out", delete=False)
        cmd = [sys.executable, os.path.abspath(__file__)] + passthrough + ["--as-rank", "--transport", transport]
        child = subprocess.Popen(cmd, env=env, stdout=out, stdin=subprocess.DEVNULL, preexec_fn=_die_with_parent)
        running[0] = child
        # lock-step polling: once a second every supervisor says whether its child is running / done / failed; one failure (or the
        # attempt's time limit, which all supervisors reach in the same iteration) ends every child of the attempt
        limit, waited, state = (args.watchdog + 30) if args.watchdog > 0 else 10 ** 9, 0, None
        while True:
            rc = child.poll()
            # exit code 3 = rank 0 printed its line and flagged it (roofline fraction > 1): a finished measurement, not a transport failure
            t = torch.tensor([1.0 if (rc is not None and rc not in (0, 3)) else 0.0, 1.0 if rc in (0, 3) else 0.0])
            dist.all_reduce(t)
            if t[0].item() > 0:
                state = "failed"
            elif int(t[1].item()) == world:
                state = "ok"
            elif waited >= limit:
                state = "timeout"
            if state:
                break
            time.sleep(1.0)
            waited += 1
        if child.poll() is None:
            child.kill()       # exactly the process this supervisor started
        child.wait()
        running[0] = None
        codes = [None] * world
        dist.all_gather_object(codes, child.returncode)
        out.close()
        text = open(out.name).read()
        os.unlink(out.name)
        lines = [l for l in text.splitlines() if l.strip()]
        found = next((l for l in reversed(lines) if l.startswith("{") and '"metric"' in l), None)
        for l in lines:
            if l is not found:
                print(l, file=sys.stderr)
        if state == "ok":
            # rank 0's line and rank 0's exit code (0, or 3 with the line's `error` field) are the job's; "ok" without a line is a failure of its own
            line, code = found, (codes[0] if found is not None else 1)
            if rank == 0 and found is None:
                print("bench.py: every rank of the %s attempt ended normally but rank 0 printed no result line" % transport, file=sys.stderr, flush=True)
            break
        if rank == 0:          # ranks that were killed never reached hpgmg_transport_finalize_ipc: remove the attempt's shared-memory segment
            _unlink_segment("/hpgmg_bench_%s_%s" % (port, nonce))
        bad = [(r, c) for r, c in enumerate(codes) if c not in (0, 3, -9)]
        why = (f"{transport} attempt: " + ("no rank finished within %d s" % limit if state == "timeout" else
                                           ", ".join("rank %d exited with code %s" % rc for rc in bad) or "a rank was killed"))
        if rank == 0:
            print("bench.py: " + why + (" -- starting fresh rank processes with the ipc transport" if number + 1 < len(attempts) else ""), file=sys.stderr, flush=True)
    if rank == 0:
        if line is not None:
            d = json.loads(line)
            if why is not None:      # the line of the fallback attempt says what it is
                d["config"]["transport"] = (d["config"].get("transport") or "ipc").replace("ipc", "ipc (fallback: %s)" % why, 1)
                d["config"]["rccl_ranks"] = 0
            print(json.dumps(d), flush=True)
        else:
            print("bench.py: no attempt produced a result (" + (why or "?") + ")", file=sys.stderr, flush=True)
    box = [code]
    dist.broadcast_object_list(box, src=0)      # every supervisor leaves with rank 0's verdict
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(box[0])


# ---------------------------------------------------------------------------------------------------------------- one rank (or the single GPU)
class Job:
    """What a rank process sets up once: device, libraries, (N > 1) torch.distributed + the solver's transport."""
    pass


def open_job(args):
    import torch
    J = Job()
    J.args = args
    J.rank = int(os.environ.get("RANK", "0")) if args.as_rank or args.force_transport else 0
    J.world = int(os.environ.get("WORLD_SIZE", "1")) if args.as_rank else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.as_rank else 0
    J.stage = ["start"]
    rank, world = J.rank, J.world
    if args.watchdog > 0:      # a multi-rank job that hangs (a peer died, a mismatched exchange) must fail loudly, not sit until the driver's limit
        import threading

        def _expired():
            sys.stderr.write(f"bench.py: rank {rank}/{world} still in stage '{J.stage[0]}' after {args.watchdog} s -- giving up "
                             f"(HPGMG_OVERLAP=0 serialises the halo exchange, HPGMG_PAIR_REMOTE=0 exchanges once per sweep)\n")
            sys.stderr.flush()
            os._exit(124)
        dog = threading.Timer(args.watchdog, _expired)
        dog.daemon = True
        dog.start()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    fail = os.environ.get("HPGMG_TEST_FAIL_TRANSPORT", "")      # tests: "rccl" = every rank of an rccl attempt exits 97 here; "rccl:hang" = no rank gets further
    if world > 1 and fail.split(":")[0] == args.transport:
        if fail.endswith(":hang"):
            time.sleep(10 ** 6)      # until this rank's watchdog fires (exit code 124)
        else:
            raise SystemExit(97)
    if world > 1 and args.share_gpu and args.transport != "ipc":
        raise SystemExit("--share-gpu: RCCL refuses two ranks on one device (only the ipc transport can share a GPU)")
    if world > 1 and torch.cuda.device_count() < world and not args.share_gpu:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP operator path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    J.torch = torch

    import hpgmg_amd as H
    J.H, J.K, J.lib = H, H.load_kernels(), H.load_driver()
    K, lib = J.K, J.lib
    assert K.hpgmg_hip_set_device(local_rank) == 0
    lib.hpgmg_set_verbose(0)

    J.dist = None
    J.stage[0] = "transport bootstrap (torch.distributed + the solver's transport)"
    if world > 1 or args.force_transport:
        import torch.distributed as dist
        J.dist = dist
        if world == 1 and "RANK" not in os.environ:      # --force-transport outside torchrun: a one-rank job on this machine
            import tempfile      # a one-rank job needs no network rendezvous at all: a file store
            store = "file://" + os.path.join(tempfile.mkdtemp(prefix="hpgmg_bench_"), "store")
            dist.init_process_group(backend="nccl", init_method=store, rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        if args.transport == "ipc":      # torch.distributed only starts the job and carries the barriers (gloo); every message of the solver is a peer copy
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo")
            token = "/hpgmg_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("HPGMG_BENCH_NONCE", "0"))      # the supervisors' nonce: one segment per attempt
            lib.hpgmg_transport_init_ipc.restype = ctypes.c_int
            lib.hpgmg_transport_init_ipc.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
            assert lib.hpgmg_transport_init_ipc(token.encode(), rank, world) == 0
        if args.transport == "rccl" and not dist.is_initialized():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if args.transport == "rccl":
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                assert K.hpgmg_hip_rccl_unique_id(buf) == 0
                ident = torch.tensor(list(buf.raw), dtype=torch.uint8, device="cuda")
            dist.broadcast(ident, src=0)
            lib.hpgmg_transport_init_rccl.restype = ctypes.c_int
            lib.hpgmg_transport_init_rccl.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
            assert lib.hpgmg_transport_init_rccl(bytes(ident.cpu().tolist()), rank, world) == 0
        ctypes.CDLL(None).fflush(None)      # RCCL prints a version banner through C stdio: get it out now, not after the JSON line
        # first contact: a known pattern to and from every other rank, a maximum, a rank-ordered sum -- a wrong byte ends the attempt with the pair named
        J.stage[0] = "transport self-test (a known pattern to and from every rank, one maximum, one ordered sum)"
        msg = ctypes.create_string_buffer(512)
        lib.hpgmg_transport_selftest.restype = ctypes.c_int
        lib.hpgmg_transport_selftest.argtypes = [ctypes.c_char_p, ctypes.c_int]
        if lib.hpgmg_transport_selftest(msg, 512) != 0:
            sys.stderr.write("bench.py: rank %d: %s\n" % (rank, msg.value.decode()))
            sys.stderr.flush()
            os._exit(98)
        J.selftest = "passed"
    return J


def barrier(J):
    J.torch.cuda.synchronize()
    J.K.hpgmg_hip_sync()
    if J.dist is not None:
        J.dist.barrier()
    J.torch.cuda.synchronize()


def run_workload(J, workload, steps, warmup, precision="fp64"):
    """One BASELINE configuration on this job's ranks: set-up, `warmup` untimed + `steps` timed F-cycles between barriers, max over ranks.
    Returns (line dict on rank 0 | None, error text | None)."""
    H, K, lib, dist, args, rank, world = J.H, J.K, J.lib, J.dist, J.args, J.rank, J.world
    torch = J.torch
    if workload == "config5":
        precision = "fp32"
    mixed = precision == "fp32"
    w_op, w_sm, w_helm, w_vc, w_log2, w_boxes, w_text = _workloads()[workload]
    # N > 1.  strong (default): the SAME problem on N GPUs -- north_star's "256^3 at 1, 2, 4 and 8 GPUs" = `hpgmg-fv 7 8/N` (SURVEY 8e:
    # `7 8`, `7 4`, `7 2`, `7 1`); weak: the reference CLI's reading of `7 8` with N ranks (the domain grows: hpgmg-fv.c:184-197).
    boxes_per_rank = w_boxes
    if world > 1 and args.series == "strong":
        if w_boxes % world:
            raise SystemExit(f"strong scaling of `{w_log2} {w_boxes}` needs a rank count that divides {w_boxes}, got {world}")
        boxes_per_rank = w_boxes // world
    lib.hpgmg_set_smoother_precision.argtypes = [ctypes.c_int]
    lib.hpgmg_set_smoother_precision(32 if mixed else 64)
    cfg = H.Config(w_op, w_sm, w_helm, w_vc)
    assert lib.hpgmg_configure(ctypes.byref(cfg)) == 0
    J.stage[0] = f"{workload}: problem setup (levels, operators, MGBuild)"
    solver = lib.hpgmg_solver_create(w_log2, boxes_per_rank, H.BC_DIRICHLET, rank, world)
    assert solver, "no acceptable problem size"
    info = (ctypes.c_int * H.INFO_COUNT)()
    lib.hpgmg_level_info(lib.hpgmg_solver_level(solver, 0), info)
    dim, box_dim, my_boxes = info[H.INFO_DIM], info[H.INFO_BOX_DIM], info[H.INFO_NUM_MY_BOXES]
    dof = float(dim) ** 3

    J.stage[0] = f"{workload}: warm-up solves"
    for _ in range(warmup):
        lib.hpgmg_solver_fmg(solver, 0)
    J.stage[0] = f"{workload}: timed solves"

    # time only the fine-level smoother launches with hipEvents on the launch stream
    fine_cells = my_boxes * box_dim ** 3
    K.hpgmg_hip_profile_smoother_min_cells(max(fine_cells, 1))
    # an event pair idles the GPU ~10 us per timed launch (1.2 % of a config-2 solve when every launch is timed): every 5th fine-level smoother launch of the
    # timed region is timed -- 5 is coprime with the 4 / 6 / 8 launches per solve, so every position of the cycle is visited in turn
    K.hpgmg_hip_profile_smoother_stride(5)
    K.hpgmg_hip_profile_smoother(1)
    barrier(J)
    t0 = time.perf_counter()
    norm = 0.0
    for _ in range(steps):
        norm = lib.hpgmg_solver_fmg(solver, 0)      # returns ||F - A u||_inf -> synchronises
    barrier(J)
    elapsed = time.perf_counter() - t0
    K.hpgmg_hip_profile_smoother(0)
    ms, launches, cells = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_longlong()
    K.hpgmg_hip_profile_smoother_read(ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(cells))

    lib.hpgmg_pair_remote_smooths.restype = ctypes.c_longlong
    lib.hpgmg_overlap_count.restype = ctypes.c_longlong
    remote_smooths, overlapped = lib.hpgmg_pair_remote_smooths(), lib.hpgmg_overlap_count()
    lib.hpgmg_image_exchanges.restype = ctypes.c_longlong
    image_refreshes = lib.hpgmg_image_exchanges()
    J.stage[0] = f"{workload}: result reduction"
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.transport == "rccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctypes.CDLL(None).fflush(None)

    line, roof_error = None, None
    if rank == 0:
        sec_per_step = elapsed / steps
        roof = None
        smoother = SMOOTHER[workload]
        if launches.value > 0 and ms.value > 0:
            avg_s = ms.value * 1e-3 / launches.value
            # cells.value counts cell-sweeps: a kernel that does two sweeps per pass reports two per launch
            sweeps_per_launch = cells.value / launches.value / fine_cells
            fused = sweeps_per_launch > 1.5
            bytes_per_launch = (smoother[1] if fused else smoother[0] * sweeps_per_launch) * fine_cells
            achieved = bytes_per_launch / avg_s / 1e9
            unfused = smoother[0] * sweeps_per_launch * fine_cells / avg_s / 1e9
            scaled = world > 1 or workload == "config4"
            traffic, traffic_source = pmc_traffic(workload, int(fine_cells)) if not scaled else (None, None)
            if scaled:         # the single-GPU counter summary of the same kernel, scaled to the cells this launch covers (the launch has the same structure per cell): N > 1, and config 4 = config 2's kernel on 512^3
                whole, traffic_source = pmc_traffic("config2" if workload == "config4" else workload, None)
                one_gpu_cells = {"config2": 256 ** 3, "config4": 256 ** 3, "config5": 256 ** 3, "config3-fv4": 512 ** 3, "config3-27pt": 512 ** 3}.get(workload)
                traffic = whole * fine_cells / one_gpu_cells if (whole and one_gpu_cells) else None
                if traffic is None:
                    traffic_source = None
            if achieved > HBM_PEAK_GBS:      # the line is still printed (the measurement is done), with roofline null and an error field; exit code 3
                roof_error = f"roofline fraction {achieved / HBM_PEAK_GBS:.3f} > 1 -- the byte model of this kernel is wrong"
            else:
                roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4),
                        # SURVEY 8(d)'s reading: its bytes per cell per sweep x the sweeps the launch performs / the launch time / peak
                        "frac_survey_8d": round(unfused / HBM_PEAK_GBS, 4),
                        "basis": (f"frac = the bytes ONE launch has to move in the form it runs in ({smoother[1]} B/cell for {sweeps_per_launch:.0f} sweep(s) in one pass: every stream once) "
                                  f"/ launch time / peak -- the stricter reading of HBM use; frac_survey_8d = SURVEY.md 8(d)'s {smoother[0]} B/cell per sweep x {sweeps_per_launch:.0f} sweep(s) "
                                  "/ the same time / peak (= unfused_equivalent_GBs / peak): what separate sweeps would have had to move -- it rewards fusion and may approach or exceed 1"),
                        "traffic": traffic, "traffic_source": traffic_source,
                        "traffic_note": ("PMC bytes per launch from the committed summary named in traffic_source: ANOTHER run (and possibly build) of the same command, not this one"
                                         + ("; a single-GPU figure scaled to the cells this rank owns" if world > 1 else "")
                                         + ("; the 256^3 figure of the same kernel scaled to 512^3" if (world == 1 and workload == "config4") else "")) if traffic else None,
                        "kernel": smoother[2] + f" over the {my_boxes} finest-level boxes of {box_dim}^3 of this rank",
                        "sweeps_per_launch": round(sweeps_per_launch, 3),
                        "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": round(avg_s * 1e6, 2),
                        "launches_timed": launches.value,
                        # what separate sweeps would have had to move in the same time: a speed-up figure, not a bandwidth
                        "unfused_equivalent_GBs": round(unfused, 1),
                        # what the launch really moves (rocprofv3 PMC, the file named in traffic_source): DRAM-level rate
                        "dram_GBs_from_pmc": round(traffic / avg_s / 1e9, 1) if traffic else None}
        gold = golden_norm(workload, dim) if not mixed else None
        line = {
            "metric": "DOF/s (fine-grid) for FMG F-cycle", "value": dof / sec_per_step, "unit": "DOF/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True, "scaling": args.series if world > 1 else "strong", "vs_baseline": None,
            "dtype": "f64 arithmetic and iterate, f32 coefficient streams in the smoother" if mixed else "f64", "data": "synthetic",
            "config": {"workload": f"hpgmg-fv {w_log2} {boxes_per_rank}{f' x {world} ranks' if world > 1 else ''}: {dim}^3 fp64 {w_text} smoother"
                                   f"{' (mixed precision, BASELINE config 5)' if mixed else ''}, {my_boxes} boxes of {box_dim}^3 per GPU, BiCGStab bottom, Dirichlet",
                       "series": (f"{args.series} scaling: " + ("same problem on every N (north_star series)" if args.series == "strong" else "reference CLI `7 8` with N ranks (domain grows with N)")) if world > 1 else "single GPU",
                       "rccl_ranks": world if (dist is not None and args.transport == "rccl") else 0,
                       "transport": (args.transport + (": ALL RANKS SHARE ONE GPU -- a functional run of the N > 1 path, not a scaling measurement" if args.share_gpu else "")) if dist is not None else None,
                       "transport_selftest": getattr(J, "selftest", None),
                       "halo": ({"smooths_as_sweep_pairs_with_remote_faces": remote_smooths, "exchanges_overlapped_with_stencil_launches": overlapped, "refreshes_of_neighbour_box_images": image_refreshes,
                                 "note": "7-point: ONE two-cell-deep halo exchange per sweep pair, residual and coarser sweeps overlapped with the interior launch; 27-point / fv4: images of the neighbouring ranks' boxes, one refresh per red + black pass, run on the exchange stream under the tiles that read no image"}
                                if world > 1 else None),
                       "baseline_config": workload,
                       "fine_grid_dof": dof, "fcycle_residual_norm": norm,
                       # the reference's own printed norm for this problem (tests/golden/fcycle_norms.json, generated from the reference binaries): the same string or not
                       "parity_ok": (("%1.15e" % norm) == gold) if gold else None, "reference_norm": gold,
                       "parallelism": f"boxes over {world} GPU(s), " + (f"{args.transport} halo exchange" if world > 1 else "one rank")},
            "roofline": roof,
        }
        if roof_error:
            line["error"] = roof_error
    lib.hpgmg_solver_destroy(solver)
    return line, roof_error


def rank_main(args):
    J = open_job(args)
    line, roof_error = run_workload(J, args.workload, args.steps, args.warmup, args.precision)
    exit_code = 0
    if J.dist is not None:
        J.dist.barrier()                       # every rank has flushed its C-level output before rank 0 prints the result
    if J.rank == 0:
        if J.world == 1 and args.workload == "config2" and args.precision == "fp64" and not args.no_also:
            # the other BASELINE configurations under the same clock: 1 warm-up + 3 timed solves each (the 8-GPU ones in their single-rank reading), each in a FRESH
            # child process (a new process, not a re-exec) with its own time limit -- whatever ends one of them (an abort() from a failed allocation, its watchdog)
            # cannot take the headline measurement, which is already in hand, with it.  The headline line goes to stderr first, for the same reason.
            sys.stderr.write("bench.py: headline line before the `also` workloads: " + json.dumps(line) + "\n")
            sys.stderr.flush()
            also = []
            for w, k_steps, k_warm in (("config3-fv4", 3, 1), ("config3-27pt", 3, 1), ("config4", 3, 1), ("config1", 10, 3)):      # config 1 is a third of a millisecond per solve
                J.stage[0] = f"also: {w} (child process)"
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--no-also", "--no-cpu-baseline", "--steps", str(k_steps), "--warmup", str(k_warm), "--watchdog", "150"]
                try:
                    child = subprocess.run(cmd, capture_output=True, text=True, timeout=180, stdin=subprocess.DEVNULL)
                    found = next((x for x in reversed(child.stdout.splitlines()) if x.startswith("{") and '"metric"' in x), None)
                    if found is None:
                        also.append({"workload": w, "error": "exit code %s, no result line: %s" % (child.returncode, child.stderr.strip()[-300:])})
                        continue
                    l = json.loads(found)
                    also.append({"workload": w, "description": l["config"]["workload"], "steps": k_steps, "warmup": k_warm, "ms_per_step": l["ms_per_step"], "value": l["value"],
                                 "fcycle_residual_norm": l["config"]["fcycle_residual_norm"], "parity_ok": l["config"]["parity_ok"],
                                 "roofline": ({k: l["roofline"][k] for k in ("frac", "frac_survey_8d", "achieved", "avg_launch_us", "sweeps_per_launch", "algorithmic_bytes_per_launch", "basis")} if l["roofline"] else None),
                                 "error": l.get("error")})
                except Exception as exc:       # the headline measurement is done: report, never drop the line
                    also.append({"workload": w, "error": repr(exc)})
            line["also"] = also
        if J.world == 1 and not args.no_cpu_baseline and args.workload in ("config2", "config5"):
            J.stage[0] = "cpu baseline (the reference binary on the host cores)"
            line["cpu_baseline"] = cpu_baseline()
        ctypes.CDLL(None).fflush(None)      # anything C code buffered on stdout goes first: the JSON line is the last line
        print(json.dumps(line), flush=True)
        if roof_error:
            exit_code = 3
    if J.dist is not None:
        if args.transport == "ipc":
            J.dist.barrier()
            J.lib.hpgmg_transport_finalize_ipc()
        else:
            J.lib.hpgmg_transport_finalize_rccl()
        J.dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


def main():
    args = parse_args()
    if args.route_b:
        return route_b()
    if args.gpus > 1 and not args.as_rank:
        if "WORLD_SIZE" in os.environ:      # a torch.distributed.run worker (the driver's launch line, or the one below): supervise this rank's attempts
            return supervise(args)
        # `python bench.py --gpus N` outside torchrun: start the N supervisors ourselves (this parent never touches the GPU), relay their JSON line
        # and exit with their code.  Never measure one GPU and call it N.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus))))
        limit = (2 * (args.watchdog + 60) + 120) if args.watchdog > 0 else None      # two attempts and the start-up of the launcher
        try:
            child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=limit)
        except subprocess.TimeoutExpired:
            raise SystemExit(f"bench.py: the {args.gpus}-rank job did not finish within {limit} s")
        lines = [l for l in child.stdout.splitlines() if l.strip()]
        result = next((l for l in reversed(lines) if l.startswith("{") and '"metric"' in l), None)
        for l in lines:
            if l is not result:
                print(l, file=sys.stderr)
        if result is not None:
            print(result, flush=True)
        raise SystemExit(child.returncode if child.returncode or result is not None else 1)
    rank_main(args)


if __name__ == "__main__":
    main()
