#!/usr/bin/env python3
"""Benchmark of the halo2 create_proof hot path on MI355X.

A step is one pass of the synthetic BN254 kernel configuration the benchmark is quoted on: one 2^19-point G1 MSM plus
one 2^19 -> 2^21 coset-extension NTT, inputs resident in HBM.  With N > 1 every rank runs the same step on its own
columns (columns are the independent unit of the prover: no data-path collective), so scaling is weak and `value`
is the sum over ranks.  After the timed steps the run also measures evaluate_h on a synthetic constraint system
(k = 19, e = 2, A = 64, F = 16, 256 gate polynomials, 16 lookups, 8 permutation columns), an iNTT, and
(rank 0, N = 1) the restated CPU baseline on the host cores.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAUNCH_LIMIT_SECONDS = 1500                     # a self-launched job that has not ended by then is killed (exit 124)


def requested_gpus(argv):
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    return ap.parse_known_args(argv)[0].gpus


def visible_devices():
    """GPUs this process could use, counted by a child process so that the launcher itself never touches the GPU"""
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                         capture_output=True, text=True, timeout=600)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise RuntimeError(f"could not count the GPUs: {out.stderr[-300:]}")


def launch_ranks(n, argv):
    """`bench.py --gpus N` without a launcher's environment: N child processes of this script, one rank per GPU, rendezvous on
    127.0.0.1 (the same environment `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would set).  The
    parent makes no GPU call and loads neither torch nor the library; it relays rank 0's line (the children share its
    stdout, the other ranks print nothing there) and exits with the first non-zero status of a rank, after ending the
    others.  ZKHIP_BENCH_BACKEND=gloo lets the ranks share fewer GPUs than ranks (a rehearsal of the multi-rank path)."""
    import signal
    import socket
    import subprocess
    backend = os.environ.get("ZKHIP_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        have = visible_devices()
        if n > have:
            print(f"bench.py: --gpus {n} but {have} GPU(s) are visible; one rank per GPU over RCCL needs {n} "
                  f"(ZKHIP_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKHIP_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + LAUNCH_LIMIT_SECONDS
    status = 0
    live = list(procs)
    while live and not status:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                status = rc if rc > 0 else 128 - rc
                print(f"bench.py: rank {procs.index(p)} ended with status {rc}", file=sys.stderr)
        if time.monotonic() > deadline and live:
            status = 124
            print(f"bench.py: ranks still running after {LAUNCH_LIMIT_SECONDS} s", file=sys.stderr)
    for p in live:                                  # a rank failed or the limit passed: end the others, by their own PIDs
        p.send_signal(signal.SIGTERM)
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return status


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and requested_gpus(sys.argv[1:]) > 1:
    sys.exit(launch_ranks(requested_gpus(sys.argv[1:]), sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import zkhip_pkg  # noqa: E402

zk = zkhip_pkg.load()
from zkevm_circuits_amd import evalh_host, ffi, plonk, sharding, synthetic  # noqa: E402

K = 19
LOG_EXT = 21
MADS_PER_MUL, MADS_PER_SQR = 171, 135          # 81 + 81 + 9 and 45 + 81 + 9 v_mad_{u64_u32,i64_i32} / v_mul_lo_u32
# what this library's mixed addition issues (nine 29-bit digits): six products, two squares and y3 = a b - c d, whose two
# products share one reduction (81 + 81 + 81 + 9)
MADS_PER_MADD = 6 * MADS_PER_MUL + 2 * MADS_PER_SQR + 252
# Algorithmic price of one mixed addition: about 10 field multiplications of 8 x 8 + 8 x 8 + 8 = 136 32-bit multiply-adds
# (the minimum over 32-bit limbs).  The roofline fraction is quoted on this count; the kernel's own count is reported beside it.
ALG_MADS_PER_MADD = 10 * 136
VALU_MAD_PEAK = 256 * 4 * 16 * 2.4e9           # nominal: a quarter-rate 32x32+64 multiply-add, 16 lanes per SIMD per clock,
                                               # 256 CUs x 4 SIMDs x 2.4 GHz = 39.3 T/s
HBM_PEAK = 8000.0                              # GB/s
EXTRAS_LIMIT_SECONDS = 420                      # multi-rank runs: the extras are abandoned after this long (see main)
PREHEAT_SECONDS = 0.1                          # untimed steps before the warm-up: clocks ramp for about 30 ms under load
PROFILES = os.path.join(ROOT, "profiles")


def load_profile_json(name):
    """a summary that an earlier profiling run wrote under profiles/ (None when it is absent)"""
    try:
        with open(os.path.join(PROFILES, name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def rand_fr_device(n, gen):
    """n canonical Montgomery residues (< 2^252 < r) as an (n, 4) int64 CUDA tensor."""
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen)
    t[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return t


R2_FR = 0x0216D0B17F4E44A58C49833D53BB808553FE3AB1E35C59E31BB8E645AE216DA7   # 2^512 mod r


def scalars_by_distribution(ctx, n, which, seed):
    """(n, 4) uint64 Montgomery scalars.  W: 60 % zero, 20 % one, 10 % < 2^8, 5 % < 2^16, 5 % uniform; B: all < 2^16.
    Canonical values are built with numpy and moved to Montgomery form on the device (multiplication by R^2)."""
    rng = np.random.default_rng(seed)
    canon = np.zeros((n, 4), dtype=np.uint64)
    if which == "B":
        canon[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    else:
        u = rng.random(n)
        canon[(u >= 0.6) & (u < 0.8), 0] = 1
        m = (u >= 0.8) & (u < 0.9)
        canon[m, 0] = rng.integers(0, 1 << 8, size=int(m.sum()), dtype=np.uint64)
        m = (u >= 0.9) & (u < 0.95)
        canon[m, 0] = rng.integers(0, 1 << 16, size=int(m.sum()), dtype=np.uint64)
        m = u >= 0.95
        full = rng.integers(0, 2 ** 64, size=(int(m.sum()), 4), dtype=np.uint64)
        full[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        canon[m] = full
    r2 = np.array([[(R2_FR >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64)
    return ctx.field_binop(0, 2, canon, np.repeat(r2, n, axis=0))   # field 0 = Fr, op 2 = mul


def cpu_baseline(steps_hint):
    """Restated CPU prover (oracle/oracle.c, OpenMP) on the same workload: full 2^19 MSM, 2^19 -> 2^21 coset NTT, and
    evaluate_h on a bounded sample of rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as co
    threads = min(os.cpu_count() or 1, 16)
    rng = np.random.default_rng(20250523)
    n = 1 << K
    sc = rng.integers(0, 2 ** 64, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    bases = co.g1_bases(n)
    t0 = time.time()
    co.best_multiexp(sc, bases, threads=threads)
    t_msm = time.time() - t0
    dom = co.Domain(5, K, threads=threads)
    t0 = time.time()
    dom.coeff_to_extended(sc)
    t_ntt = time.time() - t0
    # evaluate_h: the bench's constraint system on a domain of 2^15 rows (a sixteenth of the 2^21 extended rows of the
    # GPU measurement; the cost per extended row does not depend on k)
    ke = 15
    cs = bench_cs()
    od = co.Domain(cs.degree(), ke, threads=threads)
    en_e = 1 << od.extended_k

    def col():
        a = rng.integers(0, 2 ** 64, size=(en_e, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        return a
    call = evalh_host.EvalhCall(cs)
    call.fixed = [col() for _ in range(cs.num_fixed_columns)]
    call.advice = [col() for _ in range(cs.num_advice_columns)]
    call.instance = [col() for _ in range(cs.num_instance_columns)]
    call.challenges = col()[:cs.num_challenges]
    call.y, call.beta, call.gamma, call.theta = (r for r in col()[:4])
    call.l0, call.l_last, call.l_active = col(), col(), col()
    call.lookups = [(col(), col(), col()) for _ in cs.lookups]
    call.perm_sigma = [col() for _ in cs.permutation_columns]
    call.perm_z = [col() for _ in range(cs.num_permutation_sets())]
    desc = call.build(ffi.MEM_HOST, ffi.FORM_EXTENDED)
    t0 = time.time()
    od.evaluate_h(desc)
    t_evalh = time.time() - t0
    return {"value": n / t_msm / 1e6, "unit": "Mpts/s", "cores": threads, "kind": "port",
            "sample": "one full 2^19 MSM (uniform scalars), one 2^19->2^21 coset NTT and evaluate_h of the bench's constraint "
                      "system over 2^17 extended rows (k = 15), restated C/OpenMP prover",
            "msm_s": t_msm, "coset_ntt_s": t_ntt, "coset_ntt_GBps": 32.0 * (n + (1 << LOG_EXT)) / t_ntt / 1e9,
            "evaluate_h_sample_s": t_evalh, "evaluate_h_extended_rows_per_s": en_e / t_evalh,
            "evaluate_h_s_at_2p21_rows_extrapolated": t_evalh * (1 << LOG_EXT) / en_e}


def ntt_product_count(k, log_ext):
    """Field products `ntt_tile_kernel` executes for one coeff_to_extended 2^k -> 2^log_ext, as csrc/ntt.hip runs it:
    passes of radix 2^r (the fewest passes of radix <= 2^8, as even as possible); in a pass the butterflies of stage 0
    have the twiddle 1 and take no product, half of stage 1's likewise; the first log_ext - k stages of the first pass
    see a zero partner (the padding is never materialised) and are copies; every pass but the last multiplies each
    element once more on store (the twiddle the next pass would apply on load, one lookup in a data-sized table); two
    thirds of the inputs take the zeta^(i mod 3) scaling."""
    en = 1 << log_ext
    npass = (log_ext + 7) // 8
    left, radices = log_ext, []
    for p in range(npass):
        r = (left + (npass - p) - 1) // (npass - p)
        radices.append(r)
        left -= r
    butterflies = 0.0
    for p, r in enumerate(radices):
        zskip = min(log_ext - k, r) if p == 0 else 0
        for s in range(r):
            if s < zskip:
                continue
            butterflies += (en / 2) * (0.0 if s == 0 else 0.5 if s == 1 else 1.0)
    boundary = (npass - 1) * float(en)
    scaling = 2.0 * (1 << k) / 3
    return {"radices": radices, "butterfly": butterflies, "boundary": boundary, "zeta_scaling": scaling,
            "total": butterflies + boundary + scaling}


def bench_cs():
    return synthetic.synthetic_cs(num_advice=64, num_fixed=16, num_instance=1, num_gate_polys=256, max_degree=5,
                                  rotations=(-2, -1, 0, 1), num_lookups=16, lookup_width=3, num_perm_cols=8)


def build_evalh(ctx, dom, gen, k, cosets=None):
    cs = bench_cs()
    assert cs.degree() == 5
    # all cosets: the whole extended domain; a subset: the compact layout of those cosets (2^k rows each)
    en = dom.extended_len if cosets is None else dom.n * len(cosets)
    keep = []

    def col():
        t = rand_fr_device(en, gen)
        keep.append(t)
        return t.data_ptr()

    call = evalh_host.EvalhCall(cs)
    call.cosets = cosets
    call.fixed = [col() for _ in range(cs.num_fixed_columns)]
    call.advice = [col() for _ in range(cs.num_advice_columns)]
    call.instance = [col() for _ in range(cs.num_instance_columns)]
    call.challenges = rand_fr_device(cs.num_challenges, gen).cpu().numpy().view(np.uint64)
    sc = rand_fr_device(4, gen).cpu().numpy().view(np.uint64)
    call.y, call.beta, call.gamma, call.theta = sc[0], sc[1], sc[2], sc[3]
    call.l0, call.l_last, call.l_active = col(), col(), col()
    call.lookups = [(col(), col(), col()) for _ in cs.lookups]
    call.perm_sigma = [col() for _ in cs.permutation_columns]
    call.perm_z = [col() for _ in range(cs.num_permutation_sets())]
    out = torch.empty((en, 4), dtype=torch.int64, device="cuda")
    keep.append(out)
    desc = call.build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED)
    c_read = (cs.num_advice_columns + cs.num_fixed_columns + cs.num_instance_columns + len(cs.permutation_columns)
              + cs.num_permutation_sets() + 3 + 3 * len(cs.lookups))
    n_calc = len(call.ev.custom_gates.calculations) + sum(len(g.calculations) for g in call.ev.lookups)
    return desc, out, keep, call, c_read, n_calc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--skip-evalh", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-extras", action="store_true",
                    help="only the timed steps: the kernel averages of a rocprofv3 run then describe exactly the timed region")
    ap.add_argument("--skip-pmc", action="store_true", help="quote HBM traffic from profiles/ instead of collecting the counters now")
    ap.add_argument("--skip-proof", action="store_true", help="leave out the whole-proof measurement (create_proof_k19)")
    ap.add_argument("--skip-replay", action="store_true", help="leave out the replay of the SuperCircuit's operator multiset")
    ap.add_argument("--evalh-k", type=int, default=K)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # --gpus N > 1 without a launcher's environment never gets here (launch_ranks above); what is left is a launcher
        # whose world size contradicts the flag
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # the replay is a child process with 166 GB of columns of its own: it runs before this process touches the GPU
    replay = None
    if not args.skip_extras and not args.skip_replay and world == 1:
        try:
            replay = run_replay()
        except Exception as exc:                              # reported, never allowed to lose the bench line
            replay = {"error": f"{type(exc).__name__}: {exc}"}
    # one rank per GPU; ZKHIP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("ZKHIP_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    ranks_seen = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                print(f"bench.py: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible)", file=sys.stderr)
                sys.exit(2)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()                 # what the line reports is what the process group holds
        ranks_seen = gather_rank_ids(rank, device_index)

    ctx = zk.Context(device_index)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20250523 + rank)

    n, en = 1 << K, 1 << LOG_EXT
    bases = ctx.g1_multiples(n)                 # synthetic SRS [1]G .. [n]G, computed on the device
    bid = ctx.register_bases(bases)
    dom = zk.Domain(ctx, 5, K)                  # degree 5 -> quotient degree 4 -> extended_k = 21
    assert dom.extended_k == LOG_EXT
    with torch.cuda.stream(stream):
        scalars = rand_fr_device(n, gen)
        coeff = rand_fr_device(n, gen)
        ext = torch.empty((en, 4), dtype=torch.int64, device="cuda")
    stream.synchronize()

    def step():
        jac = ctx.msm_range_dev(bid, 0, scalars.data_ptr(), n)
        dom.coeff_to_extended_dev(coeff.data_ptr(), ext.data_ptr())
        return jac

    # the chip needs about 30 ms of sustained load to reach its steady clocks (the accumulate kernel of the first MSM after
    # an idle period takes 615 us, from the thirtieth on 555 us): the same step, untimed, before the warm-up steps
    # (the first steps after set-up are timed too and reported as `cold_start`, so that the line shows both states)
    cold_n = 5
    step()
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for _ in range(cold_n):
        step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t_cold) * 1e3 / cold_n
    preheat_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PREHEAT_SECONDS:
        step()
        preheat_steps += 1
    for _ in range(args.warmup):
        step()
    # HIP events around the two kernels the rooflines are quoted for only: bracketing all four stages of the MSM costs the
    # step about 30 us of idle queue; the stage figures come from a separate pass below
    ctx.timing_enable(2)
    ctx.timing_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps + 1)]
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev[0].record(stream)
        for s in range(args.steps):
            ctx.msm_range_dev(bid, 0, scalars.data_ptr(), n)
            ev[2 * s + 1].record(stream)
            dom.coeff_to_extended_dev(coeff.data_ptr(), ext.data_ptr())
            ev[2 * s + 2].record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    msm_ms = sum(ev[2 * s].elapsed_time(ev[2 * s + 1]) for s in range(args.steps)) / args.steps
    ntt_ms = sum(ev[2 * s + 1].elapsed_time(ev[2 * s + 2]) for s in range(args.steps)) / args.steps
    acc_us, acc_launches = ctx.timing_get("msm_accumulate")
    ntt_us, _ = ctx.timing_get("ntt")
    # the stages of the MSM: the same MSM again, outside the timed steps, with every stage bracketed
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(args.steps):
        ctx.msm_range_dev(bid, 0, scalars.data_ptr(), n)
    torch.cuda.synchronize()
    sort_us, _ = ctx.timing_get("msm_sort")
    merge_us, _ = ctx.timing_get("msm_merge")
    reduce_us, _ = ctx.timing_get("msm_reduce")
    fold_us, _ = ctx.timing_get("msm_host_fold")
    acc_bracketed_us, _ = ctx.timing_get("msm_accumulate")
    ctx.timing_enable(False)
    extra_phases = {"sort": sort_us, "accumulate": acc_bracketed_us, "merge": merge_us, "reduce": reduce_us, "host_fold": fold_us,
                    "what": "separate pass after the timed steps with every stage bracketed by HIP events (about 30 us more per MSM); "
                            "merge = the final merge level fused with level 1 of the reduce, reduce = its level 2"}

    if world > 1:
        t = torch.tensor([elapsed, msm_ms, ntt_ms], dtype=torch.float64, device=group_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, msm_ms, ntt_ms = (float(x) for x in t.cpu())

    # ---- extra measurements outside the timed steps (rank-local) ---------------------------------------
    extra = {"ranks_seen": ranks_seen,
             "launcher": "self" if os.environ.get("ZKHIP_BENCH_SELF_LAUNCHED") else ("external" if world > 1 else None),
             "backend": backend if world > 1 else None,
             "preheat": {"steps": preheat_steps, "seconds": PREHEAT_SECONDS,
                         "what": "untimed steps before the warm-up steps, until the chip holds its steady clocks"},
             "cold_start": {"ms_per_step": cold_ms, "steps": cold_n,
                            "what": "the same step timed right after set-up (one step run before), before the preheat: the chip "
                                    "has not reached its steady clocks yet"},
             "msm_phases_us": extra_phases}
    # With several ranks some extras below are collective (split evaluate_h, sharded proof, split MSM): a rank that fails or
    # hangs inside one would leave the others waiting until the collective library aborts the job, and the measured line
    # would be lost with it.  Every rank therefore arms a timer: when it fires, rank 0 prints the line of the timed steps
    # (complete at this point) without the extras and every rank leaves.
    watchdog = None
    if world > 1:
        def abandon_extras():
            try:
                if rank == 0:
                    finish(args, ctx, dist, world, rank, device_index, bid, scalars, n, en, elapsed, msm_ms, ntt_ms, acc_us,
                           acc_launches, ntt_us, dict(extra, extras=f"abandoned after {EXTRAS_LIMIT_SECONDS} s: a rank failed or hung in "
                                                       "a multi-rank extra measurement; the timed steps are complete; exit status 3"),
                           None, minimal=True)
                    sys.stdout.flush()
                else:
                    time.sleep(5)                     # rank 0 gets its line out before the launcher sees a rank end
            finally:
                os._exit(3)                           # a hung job is a failed job: the partial line is out, the status says so
        watchdog = threading.Timer(EXTRAS_LIMIT_SECONDS, abandon_extras)
        watchdog.daemon = True
        watchdog.start()

    if not args.skip_extras:
        measure_extras(ctx, dom, bid, stream, gen, n, en, extra)
    evalh = None
    if not args.skip_evalh and not args.skip_extras:
        evalh = measure_evalh(ctx, dom, stream, gen, args.evalh_k)
        if world > 1:
            try:
                split = measure_evalh_split(ctx, dom, stream, gen, world, rank)
            except Exception as exc:                      # reported, never allowed to lose the bench line
                split = {"error": f"{type(exc).__name__}: {exc}"}
            if evalh is not None:
                evalh["split_by_coset"] = split
    if not args.skip_extras and not args.skip_proof:
        try:
            extra["create_proof_k19"] = measure_create_proof(ctx, bid, dist if world > 1 else None)
        except Exception as exc:                          # reported, never allowed to lose the line above it
            extra["create_proof_k19"] = {"error": f"{type(exc).__name__}: {exc}"}
    if replay is not None:
        extra["replay_supercircuit_shape"] = replay
    finish(args, ctx, dist, world, rank, device_index, bid, scalars, n, en, elapsed, msm_ms, ntt_ms, acc_us, acc_launches, ntt_us,
           extra, evalh, watchdog=watchdog)


def group_device():
    """where the small tensors of a reduction live: on the GPU under RCCL (it moves nothing else), in host memory otherwise"""
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def gather_rank_ids(rank, device_index):
    """[{rank, device, pci_bus}] of every rank, gathered over the process group (device tensors: RCCL moves nothing else)"""
    props = torch.cuda.get_device_properties(device_index)
    mine = torch.tensor([rank, device_index, int(getattr(props, "pci_bus_id", -1)), os.getpid()], dtype=torch.int64, device="cuda")
    got = sharding.all_gather_rows(mine, dist)
    return [dict(zip(("rank", "device", "pci_bus", "pid"), (int(v) for v in g.cpu()))) for g in got]


def measure_create_proof(ctx, bid, dist_=None):
    """One whole `create_proof` (zkevm_circuits_amd/prover.py) at k = 19 for the SuperCircuit STAND-IN of
    `synthetic.supercircuit_standin`: a satisfied circuit with the advice, lookup and permutation column counts that the
    committed static census of the reference's sources gives for the SuperCircuit (tests/golden/supercircuit_census.json)
    and its hard features together -- about 360 advice columns in two phases, three challenges, about 200 lookups of which
    eight are ten columns wide into advice tables, keccak-like rotations up to +204 (32 blinding factors), about 2100 gate
    polynomials of degree up to nine (extended domain 2^22).  The SRS here is the benchmark's unstructured table, so this proof is timed, not
    verified; tests/test_gpu_configs.py runs the same circuit over a structured SRS and verifies the bytes.  With several
    ranks the commitments are column-sharded and the quotient numerator coset-sharded; seconds = the slowest rank."""
    from zkevm_circuits_amd import prover
    k = K
    cs, shape = synthetic.supercircuit_standin("standard")
    dom = zk.Domain(ctx, cs.degree(), k)
    params = prover.ParamsKZG.from_registered(ctx, dom, bid, bid)
    # the witness columns live in page-locked host memory (what the library asks of its caller): the upload is a DMA
    advice, instance, fixed, asm = synthetic.standin_witness(cs, dom.n, k, active_rows=3000, zeros=ctx.host_zeros)
    t0 = time.perf_counter()
    ctx.reset_peak()
    resident0 = ctx.allocated_bytes
    pk = prover.keygen(ctx, dom, params, cs, fixed, asm, dist=dist_)
    ctx.sync()
    t1 = time.perf_counter()
    phases = {}
    keygen_seconds = t1 - t0
    if dist_ is not None:
        dist_.barrier()
        t1 = time.perf_counter()
    # the timed proof runs without phase timings (they synchronise the device at every phase boundary, which keeps the host
    # from preparing a phase while the device finishes the one before); the phases come from a further proof below
    proof = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(),
                                None if dist_ is None else phases, dist=dist_)
    seconds = time.perf_counter() - t1
    # proofs in a row, as a prover serves them: the large blocks stay in a workspace between the calls (single rank)
    repeat = None
    if dist_ is None:
        ws = prover.ProofWorkspace(ctx)
        times = []
        for _ in range(3):
            ctx.sync()
            t2 = time.perf_counter()
            again = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(),
                                        workspace=ws)
            times.append(time.perf_counter() - t2)
            assert again == proof
        ctx.sync()
        t2 = time.perf_counter()
        again = prover.create_proof(ctx, dom, params, pk, advice, instance, prover.FieldRng(k), prover.Blake2bWrite(), phases,
                                    workspace=ws)
        phases["total_with_phase_synchronisation"] = time.perf_counter() - t2
        assert again == proof
        repeat = {"seconds_each": times, "what": "the same proof three more times with the large device blocks kept in a "
                                                  "ProofWorkspace between the calls (the first of them allocates the blocks)"}
        # the first-phase witness already in device memory (a caller that uploads columns while it produces them, or the
        # next proof's witness sent while this one runs): what is left of a proof when the host link is out of the way
        first_phase = list(cs.phases()[0][1])
        host_cols = advice(0, [0] * cs.num_challenges)
        resident = {c: prover.DeviceColumn(ctx.upload(host_cols[c]), len(host_cols[c])) for c in first_phase}
        mixed = lambda phase, challenges: [resident.get(c, col) for c, col in enumerate(advice(phase, challenges))]   # noqa: E731
        ctx.sync()
        t2 = time.perf_counter()
        again = prover.create_proof(ctx, dom, params, pk, mixed, instance, prover.FieldRng(k), prover.Blake2bWrite(), workspace=ws)
        repeat["witness_resident_seconds"] = time.perf_counter() - t2
        assert again == proof
        for col in resident.values():
            ctx.free(col.ptr)
        ws.close()
    for p in pk.pointers():
        ctx.free(p)
    dom.close()
    ranks = 1
    if dist_ is not None:
        ranks = dist_.get_world_size()
        t = torch.tensor([seconds], dtype=torch.float64, device=group_device())
        dist_.all_reduce(t, op=dist_.ReduceOp.MAX)
        seconds = float(t.cpu()[0])
    return {"seconds": seconds, "ranks": ranks, "keygen_seconds": keygen_seconds, "phases_s": phases, "proof_bytes": len(proof),
            "peak_device_GB_this_rank": (ctx.peak_allocated_bytes - resident0) / 1e9,
            "evaluate_h_low_degree_split": {"used": bool(pk.polys.split), "multiplications_per_row": pk.split_programs},
            "repeat_with_workspace": repeat,
            "shape": {"k": k, "extended_k": dom.extended_k, "advice": cs.num_advice_columns, "fixed": cs.num_fixed_columns,
                      "lookups": len(cs.lookups), "widest_lookup": max(len(lk.input_expressions) for lk in cs.lookups),
                      "permutation_columns": len(cs.permutation_columns), "blinding_factors": cs.blinding_factors(),
                      "phases": len(cs.phases()), "challenges": cs.num_challenges,
                      "gate_polynomials": sum(len(g) for g in cs.gates), "degree": cs.degree(),
                      "census": shape["census"]},
            "note": "STAND-IN, not the SuperCircuit (which cannot be instantiated without the reference's Rust halo2 fork): wall "
                    "time of one proof (slowest rank) including the Python host side and the upload of the witness; with "
                    "several ranks only the commitments and the quotient numerator are sharded, the rest is replicated; timed "
                    "here, verified in tests/test_gpu_configs.py"}


def run_replay():
    """`replay.py` (the operator multiset of one create_proof at the SuperCircuit's estimated shape, synthetic columns) as
    a child process; returns its JSON line"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "replay.py")], capture_output=True, text=True, timeout=600)
    for ln in reversed(out.stdout.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    raise RuntimeError(f"replay.py printed no JSON line (exit {out.returncode}): {out.stderr[-300:]}")


def live_traffic(static):
    """HBM bytes per launch of the two kernels of the roofline objects, from counters collected NOW: two child runs of this
    file's timed region under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE cannot share a pass; the program itself stands
    behind `--`), each under a hard time limit.  The counter-to-bytes factors of the access shapes (a streamed 32-byte read
    is tallied at half its size on gfx950, a 64-byte gather at its size) are the ones `tools/gather_probe` measured, read
    from `static` (profiles/pmc_traffic.json).  Returns None when the profiler is absent, already wraps this process, or a
    pass fails: the caller then quotes the committed file and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if not rocprof or not static or any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES")):
        return None
    calib = static.get("calibration", {})
    gather = calib.get("gather64", {}).get("counter_over_true")
    stream32 = calib.get("stream32", {}).get("counter_over_true")
    if not gather or not stream32:
        return None
    tmp = tempfile.mkdtemp(prefix="zkhip_pmc_", dir="/tmp")
    per_kernel = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["timeout", "-s", "KILL", "150", rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--skip-cpu", "--skip-extras", "--steps", "6", "--warmup", "1"]
            env = dict(os.environ, TMPDIR="/tmp")
            for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(v, None)
            done = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200)
            if done.returncode != 0:
                return None
            total, count = {}, {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] != counter:
                            continue
                        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("zkhip::", "").strip()
                        total[name] = total.get(name, 0.0) + float(r["Counter_Value"])
                        count[name] = count.get(name, 0) + 1
            per_kernel[counter] = {k: (total[k] / count[k] * 1024.0, count[k]) for k in total}      # the counters are in KiB
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = per_kernel.get("FETCH_SIZE", {}), per_kernel.get("WRITE_SIZE", {})
    if "accumulate_kernel" not in fetch or "ntt_tile_kernel" not in fetch:
        return None
    how = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this command's timed region, collected by this run (two child runs)"
    f_b, w_b = fetch["accumulate_kernel"][0], write.get("accumulate_kernel", (0.0, 0))[0]
    acc = {"fetch_counter_bytes": f_b, "write_counter_bytes": w_b, "fetch_correction": 1.0 / gather,
           "fetch_correction_from": "gather64 (64-byte gathers of table points dominate its reads)",
           "bytes_per_launch": f_b / gather + w_b, "launches_counted": fetch["accumulate_kernel"][1], "source": how}
    f_b, w_b = fetch["ntt_tile_kernel"][0], write.get("ntt_tile_kernel", (0.0, 0))[0]
    ntt = {"fetch_counter_bytes_per_launch": f_b, "write_counter_bytes_per_launch": w_b, "launches_per_transform": 3,
           "fetch_correction": 1.0 / stream32, "fetch_correction_from": "stream32 (32 contiguous bytes per lane)",
           "bytes_per_transform": 3 * (f_b / stream32 + w_b), "launches_counted": fetch["ntt_tile_kernel"][1], "source": how}
    return {"accumulate_kernel": acc, "ntt_tile_kernel": ntt}


def measure_extras(ctx, dom, bid, stream, gen, n, en, extra):
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lag = rand_fr_device(n, gen)
        dom.lagrange_to_coeff_dev(lag.data_ptr())
        e0.record(stream)
        for _ in range(5):
            dom.lagrange_to_coeff_dev(lag.data_ptr())
        e1.record(stream)
    torch.cuda.synchronize()
    intt_ms = e0.elapsed_time(e1) / 5
    extra["intt_2p19_ms"] = intt_ms
    extra["intt_2p19_GBps"] = 32.0 * 2 * n / (intt_ms * 1e-3) / 1e9
    # 64-column batch against the same bases (8 distinct scalar vectors cycled): the prover's commit phase
    with torch.cuda.stream(stream):
        cols = [rand_fr_device(n, gen) for _ in range(8)]
    stream.synchronize()
    ptrs = [cols[i % 8].data_ptr() for i in range(64)]
    ctx.msm_batch_dev(bid, ptrs[:16], n)     # warm-up: every workspace of the pipeline (up to ten) gets its buffers here
    torch.cuda.synchronize()
    tb0 = time.perf_counter()
    ctx.msm_batch_dev(bid, ptrs, n)
    torch.cuda.synchronize()
    tb = time.perf_counter() - tb0
    extra["msm_batch64_ms_per_msm"] = tb * 1e3 / 64
    extra["msm_batch64_Mpts_per_s"] = 64 * n / tb / 1e6
    del cols
    # witness-like and 16-bit scalar distributions (single MSM each)
    for which in ("W", "B"):
        d_sc = ctx.upload(scalars_by_distribution(ctx, n, which, 20250523))
        ctx.msm_range_dev(bid, 0, d_sc, n)
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        for _ in range(3):
            ctx.msm_range_dev(bid, 0, d_sc, n)
        torch.cuda.synchronize()
        td = (time.perf_counter() - td0) / 3
        extra[f"msm_{which}_ms"] = td * 1e3
        extra[f"msm_{which}_Mpts_per_s"] = n / td / 1e6
        ctx.free(d_sc)
    # 64 columns of coset extension in one call
    with torch.cuda.stream(stream):
        bcoef = rand_fr_device(64 * n, gen)
        bext = torch.empty((64 * en, 4), dtype=torch.int64, device="cuda")
    stream.synchronize()
    dom.coeff_to_extended_dev(bcoef.data_ptr(), bext.data_ptr(), batch=64)
    torch.cuda.synchronize()
    tn0 = time.perf_counter()
    dom.coeff_to_extended_dev(bcoef.data_ptr(), bext.data_ptr(), batch=64)
    torch.cuda.synchronize()
    tn = time.perf_counter() - tn0
    extra["coset_ntt_batch64_ms_per_column"] = tn * 1e3 / 64
    extra["coset_ntt_batch64_GBps"] = 64 * 32.0 * (n + en) / tn / 1e9
    del bcoef, bext
    mads, muls = ctx.probe_mad_rate(1024)
    extra["fr_mul_probe_Gps"] = muls / 1e9
    # the prover steps next to the hot path, n = 2^19: batch inversion, one permutation set of 3 columns, one lookup
    # product, one polynomial evaluation
    with torch.cuda.stream(stream):
        cols = [rand_fr_device(n, gen) for _ in range(7)]
    stream.synchronize()
    sc = rand_fr_device(4, gen).cpu().numpy().view(np.uint64)
    ptr = [c.data_ptr() for c in cols]

    def timed(fn, reps=5):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    extra["batch_invert_2p19_ms"] = timed(lambda: ctx.batch_invert_dev(0, ptr[6], n))
    extra["permutation_product_3cols_2p19_ms"] = timed(
        lambda: dom.permutation_product_dev(ptr[0:3], ptr[3:6], sc[0], sc[1], sc[2], sc[3], ptr[6]))
    extra["lookup_product_2p19_ms"] = timed(lambda: dom.lookup_product_dev(ptr[0], ptr[1], ptr[2], ptr[3], sc[0], sc[1], ptr[6]))
    extra["eval_polynomial_2p19_ms"] = timed(lambda: ctx.eval_polynomial_dev(ptr[0], n, sc[0]))
    # SHPLONK building blocks: a 64-term combination (2 GiB read) and one division by (X - z)
    lens = [n] * 64
    cf = rand_fr_device(64, gen).cpu().numpy().view(np.uint64)
    lc_ptrs = [ptr[i % len(ptr)] for i in range(64)]
    extra["poly_lincomb_64x2p19_ms"] = timed(lambda: ctx.poly_lincomb_dev(lc_ptrs, lens, cf, n, ptr[-1]))
    # seven distinct polynomials are cycled, so most of this traffic is served by the 256 MB Infinity Cache, not HBM
    extra["poly_lincomb_cached_GBps"] = 65 * n * 32 / (extra["poly_lincomb_64x2p19_ms"] * 1e-3) / 1e9
    extra["kate_division_2p19_ms"] = timed(lambda: ctx.kate_division_dev(ptr[0], n, sc[0], ptr[-1]))
    # the sort behind the permuted lookup columns: table = random column, input = values drawn from it
    usable = n - 6
    with torch.cuda.stream(stream):
        table = rand_fr_device(n, gen)
        inputs = table[torch.randint(0, usable, (n,), device="cuda", generator=gen)].contiguous()
        pa, ps = torch.empty_like(table), torch.empty_like(table)
    stream.synchronize()
    extra["lookup_permute_2p19_ms"] = timed(lambda: dom.lookup_permute_dev(inputs.data_ptr(), table.data_ptr(), usable,
                                                                          pa.data_ptr(), ps.data_ptr()))
    del cols


def measure_evalh(ctx, dom, stream, gen, evalh_k):
    kdom = dom if evalh_k == K else zk.Domain(ctx, 5, evalh_k)
    with torch.cuda.stream(stream):
        desc, out, keep, call, c_read, n_calc = build_evalh(ctx, kdom, gen, evalh_k)
    stream.synchronize()
    kdom.evaluate_h(desc, out.data_ptr())           # warm-up (includes program upload)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t1 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        kdom.evaluate_h(desc, out.data_ptr())
    torch.cuda.synchronize()
    ev_s = (time.perf_counter() - t1) / reps
    k_us, k_n = ctx.timing_get("evalh")
    ctx.timing_enable(False)
    bytes_alg = 32.0 * kdom.extended_len * (c_read + 1)
    prog = ctx.evaluate_h_last_stats()
    mads = prog["multiplications"] * MADS_PER_MUL * float(kdom.extended_len)
    counters = None
    e = (load_profile_json("r03_pmc_valu.json") or {}).get("kernels", {}).get("evalh_interp_kernel")
    if e and evalh_k == K:                       # the profiled run (tools/evalh_bench.py) is this configuration at k = 19
        insts = e["counters_per_launch"].get("SQ_INSTS_VALU")
        counters = {"file": "profiles/r03_pmc_valu.json", "avg_us_in_that_run": e["avg_us"], "SQ_INSTS_VALU_per_launch": insts,
                    "valu_issue_frac": e["valu_issue_frac"],
                    "multiply_add_share_of_valu_insts": mads / (insts * 64.0) if insts else None,
                    "mad_frac_from_counters": mads / (e["avg_us"] * 1e-6) / VALU_MAD_PEAK,
                    "hbm_bytes_per_launch": e.get("hbm_bytes_per_launch")}
    return {"program": prog, "valu_mads": mads, "valu_frac": mads / (k_us * 1e-6) / VALU_MAD_PEAK if k_us else 0.0, "counters": counters,
            "traffic": e.get("hbm_bytes_per_launch") if e and evalh_k == K else None, "k": evalh_k, "extended_k": kdom.extended_k, "seconds": ev_s, "algorithmic_GB": bytes_alg / 1e9,
            "GBps": bytes_alg / ev_s / 1e9, "frac_of_hbm_peak": bytes_alg / ev_s / 1e9 / HBM_PEAK,
            "calculations": n_calc, "kernel_ms": k_us / 1e3, "columns_read": c_read}


def measure_evalh_split(ctx, dom, stream, gen, world, rank):
    """evaluate_h of the bench configuration split by coset over the ranks (strong scaling of ONE evaluation): every
    owner evaluates its cosets on compact columns, one all-gather moves the pieces of h.  Reported next to the
    per-rank figures; the headline value does not include it."""
    parts = sharding.coset_partition(dom.extended_k, dom.k, world)
    owners = sum(p is not None for p in parts)
    mine = parts[rank]
    built = None
    if mine is not None:
        with torch.cuda.stream(stream):
            built = build_evalh(ctx, dom, gen, dom.k, mine[0])
        stream.synchronize()
        if mine[2] != dom.n * len(mine[0]):               # a row range of a shared coset
            built[3].rows = (mine[1], mine[2])
            built = (built[3].build(ffi.MEM_DEVICE, ffi.FORM_EXTENDED),) + tuple(built[1:])

    def evaluate(cs_, row_first, row_count):
        if not cs_:
            return torch.empty((0, 4), dtype=torch.int64, device="cuda")
        dom.evaluate_h(built[0], built[1].data_ptr())
        return built[1][:row_count]
    sharding.evaluate_h_by_cosets(dom.extended_k, dom.k, evaluate, dist)      # warm-up
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    pieces = sharding.evaluate_h_by_cosets(dom.extended_k, dom.k, evaluate, dist)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=group_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {"ranks": world, "ranks_with_rows": owners, "rows_per_owner": int(mine[2]) if mine is not None else 0,
            "cosets": 1 << (dom.extended_k - dom.k),
            "seconds_including_all_gather": float(t.cpu()[0]), "rows_gathered": int(sum(p.shape[0] for _, _, p in pieces))}


def finish(args, ctx, dist, world, rank, local_rank, bid, scalars, n, en, elapsed, msm_ms, ntt_ms, acc_us, acc_launches, ntt_us,
           extra, evalh, watchdog=None, minimal=False):
    """Builds and prints the line.  minimal: called by the watchdog of a multi-rank run, no collective and no GPU call."""
    split_ok = None
    if world > 1 and not minimal:
        # one MSM split by base range over the ranks: every rank takes rank 0's scalars, commits its own range, the partial
        # sums are all-gathered and folded; the fold must be the point a single GPU computes over the whole range
        try:
            shared = scalars.clone()
            sharding.broadcast_(shared, 0, dist)
            torch.cuda.synchronize()
            full = sharding.msm_base_split(
                n, lambda first, cnt: ctx.msm_range_dev(bid, first, shared.data_ptr() + 32 * first, cnt),
                ctx.g1_sum_jacobian, dist, device=torch.device("cuda", local_rank))
            single = ctx.msm_range_dev(bid, 0, shared.data_ptr(), n)
            split_ok = bool(np.array_equal(ctx.g1_to_affine(full), ctx.g1_to_affine(single)) and np.asarray(single).any())
            flag = torch.tensor([1 if split_ok else 0], dtype=torch.int64, device=group_device())
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            split_ok = bool(int(flag.cpu()[0]))
        except Exception as exc:                          # reported as a failed check; the measured line is still printed
            split_ok = f"error: {type(exc).__name__}: {exc}"

    cpu = None
    if rank == 0 and world == 1 and not args.skip_cpu:
        cpu = cpu_baseline(args.steps)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        msm_mpts = world * n / (msm_ms * 1e-3) / 1e6
        ntt_bytes = 32.0 * (n + en)
        ntt_products = ntt_product_count(K, LOG_EXT)
        ntt_mads = ntt_products["total"] * MADS_PER_MUL
        W = 16
        adds = float(n) * W                                    # mixed additions of one accumulate launch
        alg_mads, own_mads = adds * ALG_MADS_PER_MADD, adds * MADS_PER_MADD
        achieved = alg_mads / (acc_us * 1e-6) if acc_us else 0.0
        own = own_mads / (acc_us * 1e-6) if acc_us else 0.0
        probe = load_profile_json("valu_probe.json")            # tools/valu_probe on this chip (profiles/)
        mad_peak_measured = None
        if probe:
            rows = [r for r in probe["rows"] if r["inst"] == "v_mad_u64_u32"]
            mad_peak_measured = max(r["t_lane_ops_per_s"] for r in rows) * 1e12 if rows else None
        pmc = load_profile_json("pmc_traffic.json") or {}      # tools/collect_pmc.sh + tools/pmc_to_json.py (profiles/)
        live = None
        if world == 1 and not args.skip_extras and not args.skip_pmc:
            live = live_traffic(pmc)
        if live:
            pmc = live
        else:
            for k in ("accumulate_kernel", "ntt_tile_kernel"):
                if isinstance(pmc.get(k), dict):
                    pmc[k]["source"] = "profiles/pmc_traffic.json: an earlier profiling run of this command (no live counter pass in this run)"
        acc_pmc, ntt_pmc = pmc.get("accumulate_kernel"), pmc.get("ntt_tile_kernel")
        # instruction counters of the same kernels (tools/collect_pmc_valu.sh + tools/pmc_valu_to_json.py): an earlier
        # profiling run of this command on another box, so each entry carries the duration of ITS run; the fractions
        # quoted from it are counter / that duration / peak, the live fractions use this run's own durations
        valu_pmc = (load_profile_json("r03_pmc_valu.json") or {}).get("kernels", {})

        def counter_view(kernel, own_mads_per_launch):
            e = valu_pmc.get(kernel)
            if not e:
                return None
            insts = e["counters_per_launch"].get("SQ_INSTS_VALU")
            out = {"file": "profiles/r03_pmc_valu.json", "avg_us_in_that_run": e["avg_us"], "SQ_INSTS_VALU_per_launch": insts,
                   "valu_issue_frac": e["valu_issue_frac"],
                   "valu_issue_frac_is": "SQ_INSTS_VALU / (1024 SIMDs x avg_us x 2.4 GHz / 4): share of the chip's quarter-rate "
                                         "issue slots that held a vector instruction",
                   "lds_bank_conflict_share_of_lds_cycles": e.get("lds_bank_conflict_share_of_lds_cycles")}
            if insts and own_mads_per_launch:
                out["multiply_add_share_of_valu_insts"] = own_mads_per_launch / (insts * 64.0)
                out["mad_frac_from_counters"] = own_mads_per_launch / (e["avg_us"] * 1e-6) / VALU_MAD_PEAK
            return out
        standin = extra.get("create_proof_k19") or {}
        replay = extra.get("replay_supercircuit_shape") or {}
        line = {
            "metric": "super_circuit k=19 proof time (s) at 1/8 GPU; MSM Mpts/s, NTT GB/s",
            "value": msm_mpts, "unit": "Mpts/s (2^19 G1 MSM; NTT GB/s and per-phase figures in the extra keys)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (254-bit Fr/Fq as nine 29-bit digits, 32x32+64 multiply-adds)",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 2^19 G1 MSM (uniform scalars, bases [i]G) + 2^19->2^21 coset NTT per step",
                       "k": K, "extended_k": LOG_EXT, "parallelism": f"columns x{world}" if world > 1 else "single GPU"},
            # the metric's first quantity: NOT measured on the SuperCircuit (it needs the reference's Rust halo2 fork); two
            # stand-ins on this GPU bracket it -- a verified-shape proof of a satisfied stand-in circuit, and the replay of
            # the operator multiset of the estimated SuperCircuit shape on random columns
            "proof_time_s_standin": {"create_proof_k19_standin_s": standin.get("seconds"),
                                     "replay_supercircuit_shape_s": replay.get("total_s"),
                                     "what": "stand-in circuits, see create_proof_k19 / replay_supercircuit_shape below"},
            "msm_ms": msm_ms, "msm_Mpts_per_s_per_gpu": n / (msm_ms * 1e-3) / 1e6,
            "coset_ntt_ms": ntt_ms, "coset_ntt_GBps": world * ntt_bytes / (ntt_ms * 1e-3) / 1e9,
            "coset_ntt_frac_of_hbm_peak": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK,
            "roofline": {"bound": "valu", "bound_note": "32x32+64 integer multiply-add issue rate of the vector ALUs: the kernel is no dense contraction (no MFMA) and moves 0.75 GB per launch (not HBM-bound)", "kernel": "accumulate_kernel",
                         # algorithmic count: 1360 multiply-adds per mixed addition (10 field products of 136)
                         "achieved": achieved / 1e12, "peak": VALU_MAD_PEAK / 1e12, "unit": "Tmad/s", "frac": achieved / VALU_MAD_PEAK,
                         "algorithmic": f"{n}*{W} mixed adds * {ALG_MADS_PER_MADD} multiply-adds",
                         # the kernel's own instruction count (nine 29-bit digits: 1638 per addition), and both against the
                         # v_mad_u64_u32 rate measured on this chip with independent chains
                         "achieved_own_count": own / 1e12, "frac_own_count": own / VALU_MAD_PEAK,
                         "peak_measured": mad_peak_measured / 1e12 if mad_peak_measured else None,
                         "frac_of_measured_peak": achieved / mad_peak_measured if mad_peak_measured else None,
                         "frac_own_count_of_measured_peak": own / mad_peak_measured if mad_peak_measured else None,
                         "avg_launch_us": acc_us, "launches": acc_launches,
                         # HBM bytes per launch from separate rocprofv3 --pmc passes over this command's timed region
                         "traffic": acc_pmc["bytes_per_launch"] if acc_pmc else None,
                         "traffic_detail": acc_pmc,
                         "algorithmic_bytes": adds * 64 + adds * 4,
                         "counters": counter_view("accumulate_kernel", own_mads)},
            "roofline_ntt": {"bound": "hbm", "kernel": "ntt_tile_kernel (3 passes)", "achieved": ntt_bytes / (ntt_us * 1e-6) / 1e9 if ntt_us else 0.0,
                             "peak": HBM_PEAK, "unit": "GB/s", "frac": (ntt_bytes / (ntt_us * 1e-6) / 1e9 / HBM_PEAK) if ntt_us else 0.0,
                             "avg_launch_us": ntt_us,
                             "traffic": ntt_pmc["bytes_per_transform"] if ntt_pmc else None,
                             "traffic_detail": ntt_pmc,
                             "algorithmic_bytes": ntt_bytes,
                             # the transform is multiplier-bound, not HBM-bound: (M/2) log2 M butterfly products, two
                             # twiddle products per element between passes, the zeta scaling of 2/3 of the inputs
                             "valu_mads": ntt_mads, "valu_frac": (ntt_mads / (ntt_us * 1e-6) / VALU_MAD_PEAK) if ntt_us else 0.0,
                             "products": ntt_products,
                             "counters": counter_view("ntt_tile_kernel", ntt_mads / len(ntt_products["radices"]))},
            "evaluate_h": evalh, "cpu_baseline": cpu, "msm_base_split_ok": split_ok,
        }
        line.update(extra)
        if watchdog is not None:
            watchdog.cancel()
        print(json.dumps(line))
    if minimal:
        return
    if watchdog is not None:
        watchdog.cancel()
    ctx.release_bases(bid)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
